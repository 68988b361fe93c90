// Pooling layers of the hot path, NHWC fp32.
//   P  : max pooling (cuDNN, -inf padding) and average_inc_pad   reference denet/layer/pool.py:28-40
//   PI : "pool-inv" nearest-neighbour up-sampling and its gradient reference denet/layer/pool_inv.py:21-26,
//        denet/layer/pool_inv_op.py:38-63 (k_pool_inv), :144-169 (k_pool_inv_grad)
// All HBM-bound; one thread per float4 of channels per output pixel, grid-stride.
#include "common.h"

namespace {

// forward max pool; also records the argmax tap (ky*kw + kx, first maximum in scan order) per element
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          unsigned char* __restrict__ arg, int N, int H, int W, int C,
                                                          int OH, int OW, int k, int s, int pad) {
    const int C4 = C / 4;
    const long total = (long)N * OH * OW * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long t = i / C4;
        const int ox = (int)(t % OW);
        t /= OW;
        const int oy = (int)(t % OH);
        const int n = (int)(t / OH);
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
        for (int ky = 0; ky < k; ++ky) {
            const int iy = oy * s - pad + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int ix = ox * s - pad + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const f32x4 v = *(const f32x4*)(x + (((long)n * H + iy) * W + ix) * C + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (v[e] > best[e]) {
                        best[e] = v[e];
                        bi[e] = ky * k + kx;
                    }
                }
            }
        }
        *(f32x4*)(y + i * 4) = best;
        if (arg) {
            uchar4 a = make_uchar4((unsigned char)bi[0], (unsigned char)bi[1], (unsigned char)bi[2],
                                   (unsigned char)bi[3]);
            *(uchar4*)(arg + i * 4) = a;
        }
    }
}

// gather form of the max-pool gradient: every input element sums dy of the windows that selected it
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy,
                                                          const unsigned char* __restrict__ arg,
                                                          float* __restrict__ dx, int N, int H, int W, int C, int OH,
                                                          int OW, int k, int s, int pad) {
    const int C4 = C / 4;
    const long total = (long)N * H * W * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long t = i / C4;
        const int ix = (int)(t % W);
        t /= W;
        const int iy = (int)(t % H);
        const int n = (int)(t / H);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // windows oy with oy*s - pad <= iy <= oy*s - pad + k - 1
        int oy_lo = (iy + pad - k + 1 + s - 1);
        oy_lo = oy_lo < 0 ? 0 : oy_lo / s;
        int oy_hi = (iy + pad) / s;
        if (oy_hi > OH - 1) oy_hi = OH - 1;
        int ox_lo = (ix + pad - k + 1 + s - 1);
        ox_lo = ox_lo < 0 ? 0 : ox_lo / s;
        int ox_hi = (ix + pad) / s;
        if (ox_hi > OW - 1) ox_hi = OW - 1;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            const int ky = iy - (oy * s - pad);
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const int kx = ix - (ox * s - pad);
                const int tap = ky * k + kx;
                const long o = (((long)n * OH + oy) * OW + ox) * C + c4 * 4;
                const uchar4 a = *(const uchar4*)(arg + o);
                const f32x4 g = *(const f32x4*)(dy + o);
                acc[0] += (a.x == tap) ? g[0] : 0.f;
                acc[1] += (a.y == tap) ? g[1] : 0.f;
                acc[2] += (a.z == tap) ? g[2] : 0.f;
                acc[3] += (a.w == tap) ? g[3] : 0.f;
            }
        }
        *(f32x4*)(dx + i * 4) = acc;
    }
}

// average_inc_pad: divisor is always k*k, padded taps contribute 0
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N,
                                                          int H, int W, int C, int OH, int OW, int k, int s, int pad) {
    const int C4 = C / 4;
    const long total = (long)N * OH * OW * C4;
    const float inv = 1.0f / (float)(k * k);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long t = i / C4;
        const int ox = (int)(t % OW);
        t /= OW;
        const int oy = (int)(t % OH);
        const int n = (int)(t / OH);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int ky = 0; ky < k; ++ky) {
            const int iy = oy * s - pad + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int ix = ox * s - pad + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                acc += *(const f32x4*)(x + (((long)n * H + iy) * W + ix) * C + c4 * 4);
            }
        }
        *(f32x4*)(y + i * 4) = acc * inv;
    }
}

__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N,
                                                          int H, int W, int C, int OH, int OW, int k, int s, int pad) {
    const int C4 = C / 4;
    const long total = (long)N * H * W * C4;
    const float inv = 1.0f / (float)(k * k);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long t = i / C4;
        const int ix = (int)(t % W);
        t /= W;
        const int iy = (int)(t % H);
        const int n = (int)(t / H);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int oy_lo = (iy + pad - k + 1 + s - 1);
        oy_lo = oy_lo < 0 ? 0 : oy_lo / s;
        int oy_hi = (iy + pad) / s;
        if (oy_hi > OH - 1) oy_hi = OH - 1;
        int ox_lo = (ix + pad - k + 1 + s - 1);
        ox_lo = ox_lo < 0 ? 0 : ox_lo / s;
        int ox_hi = (ix + pad) / s;
        if (ox_hi > OW - 1) ox_hi = OW - 1;
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
            for (int ox = ox_lo; ox <= ox_hi; ++ox)
                acc += *(const f32x4*)(dy + (((long)n * OH + oy) * OW + ox) * C + c4 * 4);
        *(f32x4*)(dx + i * 4) = acc * inv;
    }
}

// r[n, f*y+dy, f*x+dx, c] = x[n, y, x, c]
__global__ __launch_bounds__(256) void pool_inv_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N,
                                                           int H, int W, int C, int fy, int fx) {
    const int C4 = C / 4;
    const int OH = H * fy, OW = W * fx;
    const long total = (long)N * OH * OW * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long t = i / C4;
        const int ox = (int)(t % OW);
        t /= OW;
        const int oy = (int)(t % OH);
        const int n = (int)(t / OH);
        *(f32x4*)(y + i * 4) = *(const f32x4*)(x + (((long)n * H + oy / fy) * W + ox / fx) * C + c4 * 4);
    }
}

// dx[n, y, x, c] = sum over the fy x fx block of dy
__global__ __launch_bounds__(256) void pool_inv_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N,
                                                           int H, int W, int C, int fy, int fx) {
    const int C4 = C / 4;
    const int OH = H * fy, OW = W * fx;
    const long total = (long)N * H * W * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long t = i / C4;
        const int ix = (int)(t % W);
        t /= W;
        const int iy = (int)(t % H);
        const int n = (int)(t / H);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int a = 0; a < fy; ++a)
            for (int b = 0; b < fx; ++b)
                acc += *(const f32x4*)(dy + (((long)n * OH + iy * fy + a) * OW + ix * fx + b) * C + c4 * 4);
        *(f32x4*)(dx + i * 4) = acc;
    }
}

int grid_for(long total) {
    long b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" int denet_maxpool_fwd(const float* x, float* y, unsigned char* argmax, int N, int H, int W, int C, int OH,
                                 int OW, int k, int stride, int pad, hipStream_t stream) {
    DENET_CHECK_ARG(x && y, "maxpool_fwd: null pointer");
    DENET_CHECK_ARG(C % 4 == 0 && k > 0 && k * k <= 255 && stride > 0 && pad >= 0 && pad < k, "maxpool_fwd: bad args");
    long total = (long)N * OH * OW * (C / 4);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, y, argmax, N, H, W, C, OH,
                       OW, k, stride, pad);
    DENET_CHECK_LAUNCH("maxpool_fwd");
    return DENET_OK;
}

extern "C" int denet_maxpool_bwd(const float* dy, const unsigned char* argmax, float* dx, int N, int H, int W, int C,
                                 int OH, int OW, int k, int stride, int pad, hipStream_t stream) {
    DENET_CHECK_ARG(dy && argmax && dx, "maxpool_bwd: null pointer");
    DENET_CHECK_ARG(C % 4 == 0 && k > 0 && stride > 0 && pad >= 0, "maxpool_bwd: bad args");
    long total = (long)N * H * W * (C / 4);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, dy, argmax, dx, N, H, W, C, OH,
                       OW, k, stride, pad);
    DENET_CHECK_LAUNCH("maxpool_bwd");
    return DENET_OK;
}

extern "C" int denet_avgpool_fwd(const float* x, float* y, int N, int H, int W, int C, int OH, int OW, int k,
                                 int stride, int pad, hipStream_t stream) {
    DENET_CHECK_ARG(x && y, "avgpool_fwd: null pointer");
    DENET_CHECK_ARG(C % 4 == 0 && k > 0 && stride > 0 && pad >= 0, "avgpool_fwd: bad args");
    long total = (long)N * OH * OW * (C / 4);
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, y, N, H, W, C, OH, OW, k,
                       stride, pad);
    DENET_CHECK_LAUNCH("avgpool_fwd");
    return DENET_OK;
}

extern "C" int denet_avgpool_bwd(const float* dy, float* dx, int N, int H, int W, int C, int OH, int OW, int k,
                                 int stride, int pad, hipStream_t stream) {
    DENET_CHECK_ARG(dy && dx, "avgpool_bwd: null pointer");
    DENET_CHECK_ARG(C % 4 == 0 && k > 0 && stride > 0 && pad >= 0, "avgpool_bwd: bad args");
    long total = (long)N * H * W * (C / 4);
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, dy, dx, N, H, W, C, OH, OW, k,
                       stride, pad);
    DENET_CHECK_LAUNCH("avgpool_bwd");
    return DENET_OK;
}

extern "C" int denet_pool_inv_fwd(const float* x, float* y, int N, int H, int W, int C, int fy, int fx,
                                  hipStream_t stream) {
    DENET_CHECK_ARG(x && y, "pool_inv_fwd: null pointer");
    DENET_CHECK_ARG(C % 4 == 0 && fy > 0 && fx > 0, "pool_inv_fwd: bad args");
    long total = (long)N * H * fy * W * fx * (C / 4);
    hipLaunchKernelGGL(pool_inv_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, y, N, H, W, C, fy, fx);
    DENET_CHECK_LAUNCH("pool_inv_fwd");
    return DENET_OK;
}

extern "C" int denet_pool_inv_bwd(const float* dy, float* dx, int N, int H, int W, int C, int fy, int fx,
                                  hipStream_t stream) {
    DENET_CHECK_ARG(dy && dx, "pool_inv_bwd: null pointer");
    DENET_CHECK_ARG(C % 4 == 0 && fy > 0 && fx > 0, "pool_inv_bwd: bad args");
    long total = (long)N * H * W * (C / 4);
    hipLaunchKernelGGL(pool_inv_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, dy, dx, N, H, W, C, fy, fx);
    DENET_CHECK_LAUNCH("pool_inv_bwd");
    return DENET_OK;
}
