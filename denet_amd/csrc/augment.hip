// Shape / stochastic layers of the operator surface that sit around the conv stack, NHWC fp32:
//   B    zero border                         reference denet/layer/border.py:11-35
//   CM   random crop / mirror / flip         reference denet/layer/crop_mirror.py:10-58
//   D    dropout                             reference denet/layer/dropout.py:9-27
//   SKIP "concat" combine mode               reference denet/layer/skip.py:93-96
// All HBM-bound copies: one thread per float4 of channels per output pixel, grid-stride, coalesced along C.
//
// Random numbers: the reference draws from Theano's MRG_RandomStreams (an un-vendored third-party generator,
// layer/__init__.py:5-6) - its stream cannot be reproduced here, so the masks are defined by a counter-based
// generator instead: u = mix64(seed + index * GOLDEN) (the splitmix64 finaliser), no state, no mask tensor kept
// between forward and backward (the backward pass regenerates the same bits from the same counter).
// The CPU checker under tests/ restates the generator bit for bit.
#include "common.h"

namespace {

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// 24 uniform bits
__device__ __forceinline__ uint32_t u24(uint64_t seed, uint64_t idx) { return (uint32_t)(mix64(seed, idx) >> 40); }

int grid_for(long total) {
    long b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

// y[n, oy, ox, :] = x[n, oy - top, ox - left, :] inside the image, 0 in the border
__global__ __launch_bounds__(256) void border_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N,
                                                         int H, int W, int C, int left, int top, int OH, int OW) {
    const int C4 = C / 4;
    const long total = (long)N * OH * OW * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long t = i / C4;
        const int ox = (int)(t % OW);
        t /= OW;
        const int oy = (int)(t % OH);
        const int n = (int)(t / OH);
        const int iy = oy - top, ix = ox - left;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
            v = *(const f32x4*)(x + (((long)n * H + iy) * W + ix) * C + c4 * 4);
        *(f32x4*)(y + i * 4) = v;
    }
}

// dx[n, iy, ix, :] = dy[n, iy + top, ix + left, :]
__global__ __launch_bounds__(256) void border_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N,
                                                         int H, int W, int C, int left, int top, int OH, int OW) {
    const int C4 = C / 4;
    const long total = (long)N * H * W * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long t = i / C4;
        const int ix = (int)(t % W);
        t /= W;
        const int iy = (int)(t % H);
        const int n = (int)(t / H);
        *(f32x4*)(dx + i * 4) = *(const f32x4*)(dy + (((long)n * OH + iy + top) * OW + ix + left) * C + c4 * 4);
    }
}

// Per-image crop geometry of the CM layer. Reference naming (crop_mirror.py:26-53): index_x walks dim 2 (rows,
// crop[0] long), index_y walks dim 3 (columns, crop[1] long); "flip" reverses rows, "mirror" reverses columns;
// the crop offset is uniform in [0, in - crop] while training and the centre (in - crop)//2 otherwise.
struct CropGeom {
    int off_r, off_c, flip, mirror;
};

__device__ __forceinline__ CropGeom crop_geom(int n, int H, int W, int CH, int CW, uint32_t mirror_thr,
                                              uint32_t flip_thr, int train, uint64_t seed) {
    CropGeom g;
    const int dr = H - CH, dc = W - CW;
    if (train) {
        g.mirror = u24(seed, (uint64_t)n * 4 + 0) > mirror_thr;
        g.flip = u24(seed, (uint64_t)n * 4 + 1) > flip_thr;
        g.off_r = (int)(((uint64_t)u24(seed, (uint64_t)n * 4 + 2) * (uint64_t)(dr + 1)) >> 24);
        g.off_c = (int)(((uint64_t)u24(seed, (uint64_t)n * 4 + 3) * (uint64_t)(dc + 1)) >> 24);
    } else {
        g.mirror = g.flip = 0;
        g.off_r = dr / 2;
        g.off_c = dc / 2;
    }
    return g;
}

// y[n, i, j, :] = x[n, off_r + (flip ? CH-1-i : i), off_c + (mirror ? CW-1-j : j), :]
__global__ __launch_bounds__(256) void crop_mirror_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              int N, int H, int W, int C, int CH, int CW,
                                                              uint32_t mirror_thr, uint32_t flip_thr, int train,
                                                              uint64_t seed) {
    const int C4 = C / 4;
    const long total = (long)N * CH * CW * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long t = i / C4;
        const int j = (int)(t % CW);
        t /= CW;
        const int r = (int)(t % CH);
        const int n = (int)(t / CH);
        const CropGeom g = crop_geom(n, H, W, CH, CW, mirror_thr, flip_thr, train, seed);
        const int iy = g.off_r + (g.flip ? CH - 1 - r : r);
        const int ix = g.off_c + (g.mirror ? CW - 1 - j : j);
        *(f32x4*)(y + i * 4) = *(const f32x4*)(x + (((long)n * H + iy) * W + ix) * C + c4 * 4);
    }
}

// adjoint: dx is dy scattered back into the crop window, 0 outside (a gather from the input side: no atomics)
__global__ __launch_bounds__(256) void crop_mirror_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                              int N, int H, int W, int C, int CH, int CW,
                                                              uint32_t mirror_thr, uint32_t flip_thr, int train,
                                                              uint64_t seed) {
    const int C4 = C / 4;
    const long total = (long)N * H * W * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long t = i / C4;
        const int ix = (int)(t % W);
        t /= W;
        const int iy = (int)(t % H);
        const int n = (int)(t / H);
        const CropGeom g = crop_geom(n, H, W, CH, CW, mirror_thr, flip_thr, train, seed);
        int r = iy - g.off_r, j = ix - g.off_c;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)r < (unsigned)CH && (unsigned)j < (unsigned)CW) {
            if (g.flip) r = CH - 1 - r;
            if (g.mirror) j = CW - 1 - j;
            v = *(const f32x4*)(dy + (((long)n * CH + r) * CW + j) * C + c4 * 4);
        }
        *(f32x4*)(dx + i * 4) = v;
    }
}

// y = x * keep / (1 - rate); keep(n,c,h,w) = u24(seed, LOGICAL NCHW index) < keep_thr, so the mask does not depend
// on the channel padding of the device layout. Used for the forward pass and, with dy, for the backward pass.
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int N,
                                                      int HW, int C, int CL, uint32_t keep_thr, float scale,
                                                      uint64_t seed) {
    const int C4 = C / 4;
    const long total = (long)N * HW * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long t = i / C4;
        const int p = (int)(t % HW);
        const int n = (int)(t / HW);
        f32x4 v = *(const f32x4*)(x + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = c4 * 4 + e;
            const uint64_t idx = ((uint64_t)n * CL + c) * (uint64_t)HW + p;
            v[e] = (c < CL && u24(seed, idx) < keep_thr) ? v[e] * scale : 0.f;
        }
        *(f32x4*)(y + i * 4) = v;
    }
}

// y[row, 0:CA] = a[row, 0:CA], y[row, CA:CA+CB] = b[row, 0:CB], zero up to CYP (logical channel concatenation of two
// channel-padded buffers)
__global__ __launch_bounds__(256) void concat_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ y, long rows, int CA, int CAP, int CB,
                                                         int CBP, int CYP) {
    const long total = rows * CYP;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % CYP);
        const long row = i / CYP;
        float v = 0.f;
        if (c < CA)
            v = a[row * CAP + c];
        else if (c < CA + CB)
            v = b[row * CBP + (c - CA)];
        y[i] = v;
    }
}

// da[row, c] = dy[row, c] (c < CA), db[row, c] = dy[row, CA + c] (c < CB); channel padding written as 0
__global__ __launch_bounds__(256) void concat_bwd_kernel(const float* __restrict__ dy, float* __restrict__ da,
                                                         float* __restrict__ db, long rows, int CA, int CAP, int CB,
                                                         int CBP, int CYP) {
    const int CT = CAP + CBP;
    const long total = rows * CT;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % CT);
        const long row = i / CT;
        if (c < CAP)
            da[row * CAP + c] = c < CA ? dy[row * CYP + c] : 0.f;
        else {
            const int cb = c - CAP;
            db[row * CBP + cb] = cb < CB ? dy[row * CYP + CA + cb] : 0.f;
        }
    }
}

// y[row, c] = x[row, c] + bias[c] (DC layer: the data-gradient kernel has no bias epilogue); in place allowed
__global__ __launch_bounds__(256) void add_bias_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                                       float* __restrict__ y, long rows, int C) {
    const int C4 = C / 4;
    const long total = rows * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        *(f32x4*)(y + i * 4) = *(const f32x4*)(x + i * 4) + *(const f32x4*)(bias + c4 * 4);
    }
}

uint32_t thr24(double p) {
    if (p <= 0.0) return 0u;
    if (p >= 1.0) return 1u << 24;
    return (uint32_t)(p * 16777216.0);
}

}  // namespace

extern "C" int denet_border_fwd(const float* x, float* y, int N, int H, int W, int C, int left, int right, int top,
                                int bottom, hipStream_t stream) {
    DENET_CHECK_ARG(x && y, "border_fwd: null pointer");
    DENET_CHECK_ARG(C % 4 == 0 && left >= 0 && right >= 0 && top >= 0 && bottom >= 0, "border_fwd: bad args");
    const int OH = H + top + bottom, OW = W + left + right;
    const long total = (long)N * OH * OW * (C / 4);
    hipLaunchKernelGGL(border_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, y, N, H, W, C, left, top, OH,
                       OW);
    DENET_CHECK_LAUNCH("border_fwd");
    return DENET_OK;
}

extern "C" int denet_border_bwd(const float* dy, float* dx, int N, int H, int W, int C, int left, int right, int top,
                                int bottom, hipStream_t stream) {
    DENET_CHECK_ARG(dy && dx, "border_bwd: null pointer");
    DENET_CHECK_ARG(C % 4 == 0 && left >= 0 && right >= 0 && top >= 0 && bottom >= 0, "border_bwd: bad args");
    const int OH = H + top + bottom, OW = W + left + right;
    const long total = (long)N * H * W * (C / 4);
    hipLaunchKernelGGL(border_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, dy, dx, N, H, W, C, left, top,
                       OH, OW);
    DENET_CHECK_LAUNCH("border_bwd");
    return DENET_OK;
}

extern "C" int denet_crop_mirror_fwd(const float* x, float* y, int N, int H, int W, int C, int crop_h, int crop_w,
                                     float mirror_pr, float flip_pr, int train, uint64_t seed, hipStream_t stream) {
    DENET_CHECK_ARG(x && y, "crop_mirror_fwd: null pointer");
    DENET_CHECK_ARG(C % 4 == 0 && crop_h > 0 && crop_w > 0 && crop_h <= H && crop_w <= W, "crop_mirror_fwd: bad args");
    const long total = (long)N * crop_h * crop_w * (C / 4);
    hipLaunchKernelGGL(crop_mirror_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, y, N, H, W, C, crop_h,
                       crop_w, thr24(1.0 - (double)mirror_pr), thr24(1.0 - (double)flip_pr), train, seed);
    DENET_CHECK_LAUNCH("crop_mirror_fwd");
    return DENET_OK;
}

extern "C" int denet_crop_mirror_bwd(const float* dy, float* dx, int N, int H, int W, int C, int crop_h, int crop_w,
                                     float mirror_pr, float flip_pr, int train, uint64_t seed, hipStream_t stream) {
    DENET_CHECK_ARG(dy && dx, "crop_mirror_bwd: null pointer");
    DENET_CHECK_ARG(C % 4 == 0 && crop_h > 0 && crop_w > 0 && crop_h <= H && crop_w <= W, "crop_mirror_bwd: bad args");
    const long total = (long)N * H * W * (C / 4);
    hipLaunchKernelGGL(crop_mirror_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, dy, dx, N, H, W, C, crop_h,
                       crop_w, thr24(1.0 - (double)mirror_pr), thr24(1.0 - (double)flip_pr), train, seed);
    DENET_CHECK_LAUNCH("crop_mirror_bwd");
    return DENET_OK;
}

extern "C" int denet_dropout(const float* x, float* y, int N, int HW, int C, int C_logical, float rate, uint64_t seed,
                             hipStream_t stream) {
    DENET_CHECK_ARG(x && y, "dropout: null pointer");
    DENET_CHECK_ARG(C % 4 == 0 && C_logical > 0 && C_logical <= C && rate >= 0.f && rate < 1.f, "dropout: bad args");
    const long total = (long)N * HW * (C / 4);
    const float scale = (float)(1.0 / (1.0 - (double)rate));
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, y, N, HW, C, C_logical,
                       thr24(1.0 - (double)rate), scale, seed);
    DENET_CHECK_LAUNCH("dropout");
    return DENET_OK;
}

extern "C" int denet_concat_fwd(const float* a, const float* b, float* y, long rows, int CA, int CAP, int CB, int CBP,
                                int CYP, hipStream_t stream) {
    DENET_CHECK_ARG(a && b && y, "concat_fwd: null pointer");
    DENET_CHECK_ARG(CA > 0 && CB > 0 && CA <= CAP && CB <= CBP && CA + CB <= CYP, "concat_fwd: bad args");
    hipLaunchKernelGGL(concat_fwd_kernel, dim3(grid_for(rows * CYP)), dim3(256), 0, stream, a, b, y, rows, CA, CAP, CB,
                       CBP, CYP);
    DENET_CHECK_LAUNCH("concat_fwd");
    return DENET_OK;
}

extern "C" int denet_concat_bwd(const float* dy, float* da, float* db, long rows, int CA, int CAP, int CB, int CBP,
                                int CYP, hipStream_t stream) {
    DENET_CHECK_ARG(dy && da && db, "concat_bwd: null pointer");
    DENET_CHECK_ARG(CA > 0 && CB > 0 && CA <= CAP && CB <= CBP && CA + CB <= CYP, "concat_bwd: bad args");
    hipLaunchKernelGGL(concat_bwd_kernel, dim3(grid_for(rows * (CAP + CBP))), dim3(256), 0, stream, dy, da, db, rows,
                       CA, CAP, CB, CBP, CYP);
    DENET_CHECK_LAUNCH("concat_bwd");
    return DENET_OK;
}

extern "C" int denet_add_bias(const float* x, const float* bias, float* y, long rows, int C, hipStream_t stream) {
    DENET_CHECK_ARG(x && bias && y, "add_bias: null pointer");
    DENET_CHECK_ARG(C % 4 == 0 && rows >= 0, "add_bias: bad args");
    hipLaunchKernelGGL(add_bias_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, stream, x, bias, y, rows, C);
    DENET_CHECK_LAUNCH("add_bias");
    return DENET_OK;
}
