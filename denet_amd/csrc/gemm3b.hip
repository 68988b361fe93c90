// OPT-IN variant, not the headline path: fp32 GEMMs of the detection head's 1x1 convolutions (denet/layer/convolution.py:80-83,
// 4736 -> 1536 -> 1024 -> 768 -> 512 at 24x24) with every product evaluated as a 3-term bf16 split on the bf16 matrix cores,
//     a b ~= a_hi b_hi + a_hi b_lo + a_lo b_hi,    x_hi = bf16(x), x_lo = bf16(x - x_hi)        (x_lo b_lo, ~2^-16 a b, is dropped)
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16. This is NOT the exact fp32 FMA chain of igemm.hip: a product carries a relative
// error of ~2^-16 (operands are represented to 16 mantissa bits), a K-term sum ~1e-6 ... 1e-5 of its magnitude (measured, tests) -
// inside the 1e-3 activation budget of the parity contract but reported under its own key (bench.py `split_bf16`), never
// as the fp32 number. The bf16 pipe is 16x the fp32 one, so the 3 products still leave the kernel bound by the operand split
// (VALU) and the operand stream rather than by the matrix cores.
//
//   C[M][N] = A[M][K] B[N][K]^T   (row-major, K contiguous: the forward pass; A = activations, B = filters)
// Tile 128 x 128 x 32, 256 threads = 2 x 2 waves of 64 x 64; fp32 operands are loaded to registers (the next K block while
// this one is multiplied), split, and written to LDS as four bf16 images (A_hi, A_lo, B_hi, B_lo; rows padded to 80 bytes:
// the 16-byte fragment reads of 16 consecutive rows cover all 64 banks).
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int ROWB = 80;                              // bytes per LDS row: 32 bf16 + 16 bytes of padding
constexpr int IMG = BM * ROWB;                        // one bf16 image of a 128 x 32 tile
constexpr int OOBV = (int)0xF0000000u;

struct G3Params {
    const float* a;      // [M][K]
    const float* b;      // [N][K]
    const float* bias;   // [N] or null
    float* c;            // [M][N]
    int M, N, K;
    unsigned a_bytes, b_bytes;
};

// 4 fp32 -> 4 bf16 (hi) + 4 bf16 (lo), packed two per dword
__device__ __forceinline__ void split4(const f32x4 v, unsigned (&hi)[2], unsigned (&lo)[2]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const f32x2 x = {v[2 * p], v[2 * p + 1]};
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));      // v_cvt_pk_bf16_f32, RNE
        const f32x2 back = {__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xFFFF0000u)};
        const f32x2 r = x - back;
        hi[p] = h;
        lo[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    }
}

__global__ __launch_bounds__(256, 3) void gemm3b_nt_kernel(const G3Params p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * IMG];
    char* Ah = smem;
    char* Al = smem + IMG;
    char* Bh = smem + 2 * IMG;
    char* Bl = smem + 3 * IMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    // XCD-aware tile order: neighbouring tiles (same A rows) on one XCD
    const uint32_t tiles_n = p.N / BN;
    const uint32_t tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (int)(tile / tiles_n) * BM, n0 = (int)(tile % tiles_n) * BN;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.b, 0, p.b_bytes, 0x00020000);
    // loader: a wave instruction reads 8 rows x 128 B = 8 whole cache lines (lane = (row l / 8, 16-byte piece l % 8)). Measured
    // against lanes that own 16 consecutive floats of a row: 32 lines per instruction +2 %, 64 lines per instruction +25 % time
    const int lrow = tid >> 3, lk = (tid & 7) * 4;
    const int a_off = ((m0 + lrow) * p.K + lk) * 4, b_off = ((n0 + lrow) * p.K + lk) * 4;
    f32x4 ga[4], gb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ga[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                ra, m0 + lrow + 32 * i < p.M ? a_off + (32 * i * p.K + k0) * 4 : OOBV, 0, 0));
            gb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, b_off + (32 * i * p.K + k0) * 4, 0, 0));
        }
    };
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = (lrow + 32 * i) * ROWB + lk * 2;
            unsigned h[2], l[2];
            split4(ga[i], h, l);
            *(u32x2*)(Ah + o) = u32x2{h[0], h[1]};
            *(u32x2*)(Al + o) = u32x2{l[0], l[1]};
            split4(gb[i], h, l);
            *(u32x2*)(Bh + o) = u32x2{h[0], h[1]};
            *(u32x2*)(Bl + o) = u32x2{l[0], l[1]};
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment addresses: lane = (row l & 31 of a 32-row tile, K group l >> 5 of 8 values)
    const int fa = (64 * wm + (lane & 31)) * ROWB + (lane >> 5) * 16;
    const int fb = (64 * wn + (lane & 31)) * ROWB + (lane >> 5) * 16;

    gload(0);
    stage();
    __syncthreads();
    const int nk = p.K / BK;
    for (int kb = 0; kb < nk; ++kb) {
        if (kb + 1 < nk) gload((kb + 1) * BK);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ah[t] = *(const bf16x8*)(Ah + fa + t * 32 * ROWB + ks * 32);
                al[t] = *(const bf16x8*)(Al + fa + t * 32 * ROWB + ks * 32);
                bh[t] = *(const bf16x8*)(Bh + fb + t * 32 * ROWB + ks * 32);
                bl[t] = *(const bf16x8*)(Bl + fb + t * 32 * ROWB + ks * 32);
            }
            // B as the first operand: the accumulator of a lane then holds 4-element runs along N (16-byte stores)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
        if (kb + 1 < nk) {
            stage();
            __syncthreads();
        }
    }
    // epilogue. With B first the result tile is transposed in the registers: D[row = n][col = m]: lane -> m = lane & 31,
    // n = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): 4 consecutive n per register quad -> one 16-byte store
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + 64 * wm + 32 * i + (lane & 31);
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + 64 * wn + 32 * j + 8 * q + 4 * (lane >> 5);
                f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                if (p.bias) v += *(const f32x4*)(p.bias + n);
                *(f32x4*)(p.c + (long)m * p.N + n) = v;
            }
    }
}

// dst[c][r] = src[r][c] (fp32): brings the operands of the data / filter gradient into the K-contiguous form the kernel above
// takes (w^T for the data gradient; dy^T and x^T, contraction over pixels, for the filter gradient). 64 x 64 tiles through LDS.
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int C) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty + 4 * i, c = c0 + tx;
        tile[ty + 4 * i][tx] = (r < R && c < C) ? src[(long)r * C + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty + 4 * i, r = r0 + tx;
        if (r < R && c < C) dst[(long)c * R + r] = tile[tx][ty + 4 * i];
    }
}

}  // namespace

extern "C" int denet_gemm_bf16x3_ok(int M, int N, int K) {
    return (M > 0 && N > 0 && K > 0 && N % 128 == 0 && K % 32 == 0 && (long)M * K * 4 < 0xF0000000L && (long)N * K * 4 < 0xF0000000L) ? 1 : 0;
}

// C[M][N] = A[M][K] B[N][K]^T (+ bias[N]) with 3-term bf16 split products (see the header of this file): the forward pass of a
// 1x1 stride-1 convolution, A = x [N*H*W][C], B = w [K][C]. OPT-IN: not bit-compatible with denet_conv_fwd.
extern "C" int denet_gemm_bf16x3_nt(const float* a, const float* b, const float* bias, float* c, int M, int N, int K,
                                    hipStream_t stream) {
    DENET_CHECK_ARG(a && b && c, "gemm_bf16x3_nt: null pointer");
    DENET_CHECK_ARG(denet_gemm_bf16x3_ok(M, N, K), "gemm_bf16x3_nt: needs N %% 128 = 0, K %% 32 = 0");
    G3Params p = {a, b, bias, c, M, N, K, (unsigned)((size_t)M * K * 4), (unsigned)((size_t)N * K * 4)};
    const unsigned tiles = (unsigned)(((M + BM - 1) / BM) * (N / BN));
    hipLaunchKernelGGL(gemm3b_nt_kernel, dim3(tiles), dim3(256), 0, stream, p);
    DENET_CHECK_LAUNCH("gemm_bf16x3_nt");
    return DENET_OK;
}

extern "C" int denet_transpose_f32(const float* src, float* dst, int R, int C, hipStream_t stream) {
    DENET_CHECK_ARG(src && dst && R > 0 && C > 0, "transpose_f32: bad arguments");
    hipLaunchKernelGGL(transpose_f32_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, stream, src, dst, R, C);
    DENET_CHECK_LAUNCH("transpose_f32");
    return DENET_OK;
}
