// Winograd F(m x m, 3x3) paths (m = 2 or 4) for the stride-1 3x3 convolutions with many channels: forward, data
// gradient and filter gradient.
// Reference op: the same `C[k,3]` layer (denet/layer/convolution.py:80-83; its gradient model_cnn.py:318); the
// minimal-filtering algorithm computes the identical sums with (m+2)^2 multiplications per m x m output tile and
// channel pair instead of 9 m^2 (2.25x fewer MFMA FLOPs for m = 2, 4x for m = 4), at the price of HBM-bound transforms:
//     V[xi][t][c] = (B^T d B)[xi]      d: 4x4 input patch of tile t (xi = 4*i+j)          wino_input_kernel
//     U[xi][k][c] = (G g G^T)[xi]      g: 3x3 filter (already the correlation taps)        wino_filter_kernel
//     M[xi][t][k] = sum_c V[xi][t][c] * U[xi][k][c]      16 GEMMs, batched                 igemm forward kernel
//     y tile      = A^T M A (+ bias, + add)                                                wino_output_kernel
// fp32 throughout; the result differs from the direct kernel by rounding only (m = 2: ~1e-6, m = 4: ~1e-5 relative).
// Filter gradient: dM = A dy A^T (wino_dout_kernel), dU[xi] = dM[xi]^T V[xi] (batched split-K product through the wgrad
// kernel), dw = G^T dU G (wino_dfilter_kernel).
// The data gradient of a stride-1 pad-1 3x3 convolution is the same convolution of dy with the taps rotated by 180
// degrees and the channel roles swapped: wino_filter_kernel<true> writes U'[xi][c][k] from w[k][2-r][2-s][c].
#include "common.h"
#include <stdlib.h>
#include "../../include/denet_hip.h"

int denet_gemm_batched_nt(const float* a, const float* w, float* out, int batch, int M, int Nc, int Kc, long stride_a,
                          long stride_w, long stride_out, hipStream_t stream);
int denet_gemm_batched_tune(const float* a, const float* w, float* out, int batch, int M, int Nc, int Kc, long stride_a,
                            long stride_w, long stride_out, hipStream_t stream);

int denet_wgrad_batched(const float* x, const float* dy, float* dw, float* workspace, size_t workspace_bytes, int batch,
                        int T, int Cc, int Kr, hipStream_t stream);
int denet_wgrad_batched_tune(const float* x, const float* dy, float* dw, float* workspace, size_t workspace_bytes, int batch,
                             int T, int Cc, int Kr, hipStream_t stream);

// wino4f.hip: F(4x4) products + output transform in one kernel
int denet_wino4f_block(int tile, long T, int C, int K);
int denet_wino4f_stats_rows(int tb, long T);
int denet_wino4f_run(int tb, const float* V, const float* U, const float* bias, const float* add, float* y, double* stats,
                     const float* bs_x, const float* bs_y, const float* bs_gamma, const float* bs_beta, const float* bs_mean,
                     const float* bs_invstd, int bs_relu, int N, int H, int W, int C, int K, int relu, hipStream_t stream);

// wino4g.hip: F(4x4) filter-gradient component products + adjoint filter transform
int denet_wino4g_splits(int tile, long T, int C, int K);
size_t denet_wino4g_workspace_bytes(int splits, int C, int K);
int denet_wino4g_run(int splits, const float* dM, const float* V, float* dU, float* part, size_t part_bytes, long T, int C, int K,
                     hipStream_t stream);

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *(const f32x4*)p; }

// ---- transform matrices (Lavin & Gray, "Fast algorithms for convolutional neural networks") -----------------
// F(m x m, 3x3): TS = m + 2 points. Y = A^T [ (G g G^T) .* (B^T d B) ] A for the CORRELATION taps g.
template <int MO>
struct Wino;
template <>
struct Wino<2> {
    static constexpr int TS = 4;
    static constexpr float BT[4][4] = {{1, 0, -1, 0}, {0, 1, 1, 0}, {0, -1, 1, 0}, {0, 1, 0, -1}};
    static constexpr float G[4][3] = {{1, 0, 0}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0, 0, 1}};
    static constexpr float AT[2][4] = {{1, 1, 1, 0}, {0, 1, -1, -1}};
};
template <>
struct Wino<4> {
    static constexpr int TS = 6;
    static constexpr float BT[6][6] = {{4, 0, -5, 0, 1, 0},  {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0},
                                       {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
    static constexpr float G[6][3] = {{0.25f, 0, 0},
                                      {-1.f / 6, -1.f / 6, -1.f / 6},
                                      {-1.f / 6, 1.f / 6, -1.f / 6},
                                      {1.f / 24, 1.f / 12, 1.f / 6},
                                      {1.f / 24, -1.f / 12, 1.f / 6},
                                      {0, 0, 1}};
    static constexpr float AT[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};
};

// one multiply-add of a transform, written as an explicit fused operation: left to the compiler, an expression like
// 4*d0 - 5*d2 + d4 is contracted as fma(-5, d2, 4*d0) in one kernel and as fma(4, d0, -5*d2) in another (both legal, different
// roundings) - with the explicit form every kernel that evaluates a transform produces the same bits (the transforms that
// evaluate a batch norm on the fly, wino_prep_*, must equal wino_input_kernel / wino_dout_kernel exactly)
__device__ __forceinline__ float wino_fma(float c, float v, float acc) { return __builtin_fmaf(c, v, acc); }
__device__ __forceinline__ f32x4 wino_fma(float c, f32x4 v, f32x4 acc) {
    f32x4 r;
    r[0] = __builtin_fmaf(c, v[0], acc[0]);
    r[1] = __builtin_fmaf(c, v[1], acc[1]);
    r[2] = __builtin_fmaf(c, v[2], acc[2]);
    r[3] = __builtin_fmaf(c, v[3], acc[3]);
    return r;
}

// acc = sum_l coef[l] * v[l] with the compile-time coefficients folded (0 dropped): first product rounded, then one fused
// multiply-add per further term, in index order
#define WINO_DOT(acc, NL, COEF, VAL)                       \
    {                                                      \
        bool first_ = true;                                \
        _Pragma("unroll") for (int l_ = 0; l_ < (NL); ++l_) { \
            const float c_ = (COEF);                       \
            if (c_ != 0.f) {                               \
                if (first_) acc = c_ * (VAL);              \
                else acc = wino_fma(c_, (VAL), acc);       \
                first_ = false;                            \
            }                                              \
        }                                                  \
    }

// V[xi][t][c] = (B^T d B)[xi]: one thread per (tile, 4 channels); d = TS x TS input patch at (MO*ty-1, MO*tx-1).
// up = 1: x is [N][H/2][W/2][C] and the layer's input is its 2 x 2 nearest-neighbour up-sampling (the pool-inverse layer in front,
// pool_inv.py:10-41, never written): pixel (iy, ix) reads (iy / 2, ix / 2)
template <int MO>
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, float* __restrict__ V, int N, int H,
                                                         int W, int C, int TH, int TW, long T, int up) {
    using WT = Wino<MO>;
    constexpr int TS = WT::TS;
    const int c4n = C / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * c4n) return;
    const int c4 = (int)(idx % c4n);
    const long t = idx / c4n;
    const int tx = (int)(t % TW);
    const int ty = (int)((t / TW) % TH);
    const int n = (int)(t / ((long)TW * TH));
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 tt[TS][TS];       // B^T d, built column by column
#pragma unroll
    for (int j = 0; j < TS; ++j) {
        const int ix = MO * tx - 1 + j;
        f32x4 d[TS];
#pragma unroll
        for (int i = 0; i < TS; ++i) {
            const int iy = MO * ty - 1 + i;
            const bool ok = ((unsigned)iy < (unsigned)H) && ((unsigned)ix < (unsigned)W);
            d[i] = ok ? ld4(x + (((long)n * (H >> up) + (iy >> up)) * (W >> up) + (ix >> up)) * C + c4 * 4) : z;
        }
#pragma unroll
        for (int i = 0; i < TS; ++i) {
            f32x4 acc = z;
            WINO_DOT(acc, TS, WT::BT[i][l_], d[l_]);
            tt[i][j] = acc;
        }
    }
#pragma unroll
    for (int i = 0; i < TS; ++i)
#pragma unroll
        for (int j = 0; j < TS; ++j) {
            f32x4 acc = z;
            WINO_DOT(acc, TS, WT::BT[j][l_], tt[i][l_]);
            *(f32x4*)(V + ((long)(TS * i + j) * T + t) * C + c4 * 4) = acc;
        }
}

// ---- transforms whose input is formed on the fly from a batch-norm layer -----------------------------------------------------
// A Winograd pass that reads a tensor a batch norm has just written re-reads what a pointwise kernel produced one launch
// earlier. These kernels evaluate the batch-norm expression while they gather the patch (same operations in the same order as
// bn_apply_kernel / bn_bwd_apply_kernel of bn.hip, so every value is bit-identical to the separate passes) and write the tensor
// the pointwise kernel would have written from the inner MO x MO pixels of each tile (each pixel belongs to exactly one tile):
//   FWD  d = relu?(fma(x, gamma*invstd, beta - mean*gamma*invstd) (+ res));  out = d (the activation);  V = B^T d B
//        replaces bn_apply_kernel + wino_input_kernel of the next convolution (batch_norm_relu.py:34-48 -> convolution.py:80-83)
//   BWD  g = dy masked by the ReLU (y > 0, or recomputed from x), d = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat));
//        out = g (residual-branch gradient, optional);  V = B^T d B (input of the data-gradient products);
//        dM = A d A^T over the inner pixels (input of the filter-gradient products)
//        replaces bn_bwd_apply_kernel + wino_input_kernel + wino_dout_kernel: d (the gradient of the convolution's output) is
//        never written
struct BnFoldDev {
    const float* x;       // FWD: pre-normalisation tensor; BWD: the same (the convolution's output)
    const float* aux;     // FWD: residual input or NULL; BWD: dy (gradient of the batch norm's output)
    const float* y;       // BWD: forward output for the ReLU mask, or NULL (recomputed from x)
    const float* gamma;
    const float* beta;
    const float* mean;
    const float* invstd;
    const float* coef;    // BWD: [2][C] mean(g), mean(g*xhat) (bn_bwd_final_kernel)
    float* out;           // FWD: the activation; BWD: g or NULL
    int relu;
};

template <int MO, bool BWD>
__global__ __launch_bounds__(256) void wino_prep_kernel(BnFoldDev f, float* __restrict__ V, float* __restrict__ dM, int N, int H,
                                                        int W, int C, int TH, int TW, long T) {
    using WT = Wino<MO>;
    constexpr int TS = WT::TS;
    const int c4n = C / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * c4n) return;
    const int c4 = (int)(idx % c4n);
    const long t = idx / c4n;
    const int tx = (int)(t % TW);
    const int ty = (int)((t / TW) % TH);
    const int n = (int)(t / ((long)TW * TH));
    const int c = c4 * 4;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    float mu[4], is[4], sc[4], sh[4], mg[4], mgx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        mu[k] = f.mean[c + k];
        is[k] = f.invstd[c + k];
        sc[k] = f.gamma[c + k] * is[k];
        sh[k] = (f.beta ? f.beta[c + k] : 0.f) - mu[k] * sc[k];
        mg[k] = BWD ? f.coef[c + k] : 0.f;
        mgx[k] = BWD ? f.coef[C + c + k] : 0.f;
    }
    // the value of the folded tensor at pixel (iy, ix); `inner`: this tile owns the pixel and writes `out`
    auto value = [&](int iy, int ix, bool inner) -> f32x4 {
        if (!(((unsigned)iy < (unsigned)H) && ((unsigned)ix < (unsigned)W))) return z;
        const long o = (((long)n * H + iy) * W + ix) * C + c;
        const f32x4 xv = ld4(f.x + o);
        f32x4 d;
        if (!BWD) {
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = fmaf(xv[k], sc[k], sh[k]);
            if (f.aux) d += ld4(f.aux + o);
            if (f.relu) {
#pragma unroll
                for (int k = 0; k < 4; ++k) d[k] = fmaxf(d[k], 0.f);
            }
            if (inner) *(f32x4*)(f.out + o) = d;
        } else {
            f32x4 g = ld4(f.aux + o);
            if (f.relu) {
                if (f.y) {
                    const f32x4 yv = ld4(f.y + o);
#pragma unroll
                    for (int k = 0; k < 4; ++k) g[k] = yv[k] > 0.f ? g[k] : 0.f;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) g[k] = fmaf(xv[k], sc[k], sh[k]) > 0.f ? g[k] : 0.f;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xh = (xv[k] - mu[k]) * is[k];
                d[k] = sc[k] * (g[k] - mg[k] - xh * mgx[k]);
            }
            if (inner && f.out) *(f32x4*)(f.out + o) = g;
        }
        return d;
    };
    {
        f32x4 tt[TS][TS];       // B^T d, built column by column
#pragma unroll
        for (int j = 0; j < TS; ++j) {
            const int ix = MO * tx - 1 + j;
            f32x4 d[TS];
#pragma unroll
            for (int i = 0; i < TS; ++i) d[i] = value(MO * ty - 1 + i, ix, i >= 1 && i <= MO && j >= 1 && j <= MO);
#pragma unroll
            for (int i = 0; i < TS; ++i) {
                f32x4 acc = z;
                WINO_DOT(acc, TS, WT::BT[i][l_], d[l_]);
                tt[i][j] = acc;
            }
        }
#pragma unroll
        for (int i = 0; i < TS; ++i)
#pragma unroll
            for (int j = 0; j < TS; ++j) {
                f32x4 acc = z;
                WINO_DOT(acc, TS, WT::BT[j][l_], tt[i][l_]);
                *(f32x4*)(V + ((long)(TS * i + j) * T + t) * C + c) = acc;
            }
    }
    if (BWD) {
        // dM[xi][t][c] = (A d A^T)[xi] over the tile's own MO x MO pixels (wino_dout_kernel); the values are formed a second
        // time from lines this thread has just read (cache hits) - keeping them through the first transform would spill
        f32x4 a[TS][MO];
#pragma unroll
        for (int j = 0; j < MO; ++j) {
            f32x4 d[MO];
#pragma unroll
            for (int i = 0; i < MO; ++i) d[i] = value(MO * ty + i, MO * tx + j, false);
#pragma unroll
            for (int i = 0; i < TS; ++i) {
                f32x4 acc = z;
                WINO_DOT(acc, MO, WT::AT[l_][i], d[l_]);
                a[i][j] = acc;
            }
        }
#pragma unroll
        for (int i = 0; i < TS; ++i)
#pragma unroll
            for (int j = 0; j < TS; ++j) {
                f32x4 acc = z;
                WINO_DOT(acc, MO, WT::AT[l_][j], a[i][l_]);
                *(f32x4*)(dM + ((long)(TS * i + j) * T + t) * C + c) = acc;
            }
    }
}

// The same passes with the patch staged through LDS: a workgroup owns a block of BT x BT tiles (BT*MO pixels square) and CQ
// channel quads; it evaluates the folded tensor ONCE per pixel of the block + halo ((BT*MO+2)^2 pixels: 1.27x the block for
// F(4x4), instead of the 2.25x of one-thread-per-tile gathers through L2, which made the version above slower than the passes
// it replaces), writes the tensor the pointwise kernel would have written, and the threads then take their tiles from LDS.
template <int MO, bool BWD>
struct PrepLds {
    static constexpr int BT = 4;                    // tiles per block edge
    static constexpr int CQ = 8;                    // channel quads (32 channels) per workgroup
    static constexpr int PB = BT * MO;              // pixels per block edge
    static constexpr int PP = PB + 2;               // with halo
    static constexpr int NT = BT * BT * CQ;         // threads
    static constexpr int PS = CQ + 1;               // LDS pixel stride in float4 (odd: tiles of a wave fall into different banks)
    static constexpr size_t LDS = (size_t)PP * PP * PS * sizeof(f32x4);
};

template <int MO, bool BWD>
__global__ __launch_bounds__(128) void wino_prep_lds_kernel(BnFoldDev f, float* __restrict__ V, float* __restrict__ dM, int N,
                                                            int H, int W, int C, int TH, int TW, long T, int BH, int BW) {
    using WT = Wino<MO>;
    using PL = PrepLds<MO, BWD>;
    static_assert(PL::NT == 128, "launch bounds");
    constexpr int TS = WT::TS, BT = PL::BT, CQ = PL::CQ, PB = PL::PB, PP = PL::PP, NT = PL::NT, PS = PL::PS;
    extern __shared__ __attribute__((aligned(16))) unsigned char prep_smem[];
    f32x4* patch = (f32x4*)prep_smem;               // [PP][PP][PS]
    const int tid = threadIdx.x;
    const int cq = tid % CQ;
    const int cblk = blockIdx.y;                    // channel block of 32
    const int c = (cblk * CQ + cq) * 4;
    int b = blockIdx.x;
    const int bx = b % BW; b /= BW;
    const int by = b % BH;
    const int n = b / BH;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    float mu[4], is[4], sc[4], sh[4], mg[4], mgx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        mu[k] = f.mean[c + k];
        is[k] = f.invstd[c + k];
        sc[k] = f.gamma[c + k] * is[k];
        sh[k] = (f.beta ? f.beta[c + k] : 0.f) - mu[k] * sc[k];
        mg[k] = BWD ? f.coef[c + k] : 0.f;
        mgx[k] = BWD ? f.coef[C + c + k] : 0.f;
    }
    // phase 1: the folded tensor over the block + halo, one pixel x channel quad per thread and round. The loads of a batch of
    // rounds are issued together, unconditionally (a pixel outside the image reads element 0 and is zeroed afterwards):
    // with 1.5 waves per SIMD the latency of a dependent load -> LDS chain per round was the whole kernel time
    const int y0 = by * PB - 1, x0 = bx * PB - 1;
    constexpr int SLOTS = NT / CQ;                                   // pixels per round
    constexpr int ROUNDS = (PP * PP + SLOTS - 1) / SLOTS;
    constexpr int UB = 7;                                            // rounds per batch
    for (int r0 = 0; r0 < ROUNDS; r0 += UB) {
        f32x4 xv[UB], av[UB], yv[UB];
        long off[UB];
        int pix[UB];
        bool ok[UB], inner[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int p = (r0 + u) * SLOTS + tid / CQ;
            const int py = p / PP, px = p - py * PP;
            const int iy = y0 + py, ix = x0 + px;
            pix[u] = p;
            ok[u] = p < PP * PP && ((unsigned)iy < (unsigned)H) && ((unsigned)ix < (unsigned)W);
            inner[u] = py >= 1 && py <= PB && px >= 1 && px <= PB;
            off[u] = ok[u] ? (((long)n * H + iy) * W + ix) * C + c : (long)c;
            xv[u] = ld4(f.x + off[u]);
            if (BWD || f.aux) av[u] = ld4(f.aux + off[u]);
            if (BWD && f.relu && f.y) yv[u] = ld4(f.y + off[u]);
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            f32x4 d = z;
            if (!BWD) {
#pragma unroll
                for (int k = 0; k < 4; ++k) d[k] = fmaf(xv[u][k], sc[k], sh[k]);
                if (f.aux) d += av[u];
                if (f.relu) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) d[k] = fmaxf(d[k], 0.f);
                }
                if (ok[u] && inner[u]) *(f32x4*)(f.out + off[u]) = d;
            } else {
                f32x4 g = av[u];
                if (f.relu) {
                    if (f.y) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) g[k] = yv[u][k] > 0.f ? g[k] : 0.f;
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) g[k] = fmaf(xv[u][k], sc[k], sh[k]) > 0.f ? g[k] : 0.f;
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float xh = (xv[u][k] - mu[k]) * is[k];
                    d[k] = sc[k] * (g[k] - mg[k] - xh * mgx[k]);
                }
                if (ok[u] && inner[u] && f.out) *(f32x4*)(f.out + off[u]) = g;
            }
            if (pix[u] < PP * PP) patch[pix[u] * PS + cq] = ok[u] ? d : z;
        }
    }
    __syncthreads();
    // phase 2: one thread per (tile, channel quad)
    const int tl = tid / CQ;
    const int tyl = tl / BT, txl = tl - tyl * BT;
    const int ty = by * BT + tyl, tx = bx * BT + txl;
    if (ty >= TH || tx >= TW) return;
    const long t = ((long)n * TH + ty) * TW + tx;
    {
        f32x4 tt[TS][TS];
#pragma unroll
        for (int j = 0; j < TS; ++j) {
            f32x4 d[TS];
#pragma unroll
            for (int i = 0; i < TS; ++i) d[i] = patch[((MO * tyl + i) * PP + MO * txl + j) * PS + cq];
#pragma unroll
            for (int i = 0; i < TS; ++i) {
                f32x4 acc = z;
                WINO_DOT(acc, TS, WT::BT[i][l_], d[l_]);
                tt[i][j] = acc;
            }
        }
#pragma unroll
        for (int i = 0; i < TS; ++i)
#pragma unroll
            for (int j = 0; j < TS; ++j) {
                f32x4 acc = z;
                WINO_DOT(acc, TS, WT::BT[j][l_], tt[i][l_]);
                *(f32x4*)(V + ((long)(TS * i + j) * T + t) * C + c) = acc;
            }
    }
    if (BWD) {
        f32x4 a[TS][MO];
#pragma unroll
        for (int j = 0; j < MO; ++j) {
            f32x4 d[MO];
#pragma unroll
            for (int i = 0; i < MO; ++i) d[i] = patch[((MO * tyl + 1 + i) * PP + MO * txl + 1 + j) * PS + cq];
#pragma unroll
            for (int i = 0; i < TS; ++i) {
                f32x4 acc = z;
                WINO_DOT(acc, MO, WT::AT[l_][i], d[l_]);
                a[i][j] = acc;
            }
        }
#pragma unroll
        for (int i = 0; i < TS; ++i)
#pragma unroll
            for (int j = 0; j < TS; ++j) {
                f32x4 acc = z;
                WINO_DOT(acc, MO, WT::AT[l_][j], a[i][l_]);
                *(f32x4*)(dM + ((long)(TS * i + j) * T + t) * C + c) = acc;
            }
    }
}

template <int MO, bool BWD>
int launch_prep_lds(const BnFoldDev& f, float* V, float* dM, int N, int H, int W, int C, int TH, int TW, long T, hipStream_t stream) {
    using PL = PrepLds<MO, BWD>;
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute((const void*)wino_prep_lds_kernel<MO, BWD>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)PL::LDS);
        if (e != hipSuccess) {
            denet_set_error("conv_wino fold: hipFuncSetAttribute(%zu B LDS): %s", PL::LDS, hipGetErrorString(e));
            return -(int)e;
        }
        attr_set = true;
    }
    const int BH = (TH + PL::BT - 1) / PL::BT, BW = (TW + PL::BT - 1) / PL::BT;
    hipLaunchKernelGGL((wino_prep_lds_kernel<MO, BWD>), dim3((unsigned)(N * BH * BW), (unsigned)(C / (4 * PL::CQ))), dim3(PL::NT),
                       PL::LDS, stream, f, V, dM, N, H, W, C, TH, TW, T, BH, BW);
    return DENET_OK;
}

// U[xi][k][c] = (G g G^T)[xi] (DGRAD = false) or U'[xi][c][k] from the rotated taps (DGRAD = true); thread per (k, c)
template <int MO, bool DGRAD>
__global__ __launch_bounds__(256) void wino_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int K, int C) {
    using WT = Wino<MO>;
    constexpr int TS = WT::TS;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)K * C) return;
    const int c = (int)(idx % C);
    const int k = (int)(idx / C);
    float g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) g[r][s] = w[(((long)k * 3 + (DGRAD ? 2 - r : r)) * 3 + (DGRAD ? 2 - s : s)) * C + c];
    float a[TS][3];
#pragma unroll
    for (int i = 0; i < TS; ++i)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            float acc = 0.f;
            WINO_DOT(acc, 3, WT::G[i][l_], g[l_][s]);
            a[i][s] = acc;
        }
    const long KC = (long)K * C;
    const long o = DGRAD ? ((long)c * K + k) : ((long)k * C + c);
#pragma unroll
    for (int i = 0; i < TS; ++i)
#pragma unroll
        for (int j = 0; j < TS; ++j) {
            float acc = 0.f;
            WINO_DOT(acc, 3, WT::G[j][l_], a[i][l_]);
            U[(long)(TS * i + j) * KC + o] = acc;
        }
}

// Data-gradient filters U'[xi][c][k] from the rotated taps of w[k][r][s][c]: the (k, c) -> (c, k) transposition goes through
// LDS so that global reads (16 consecutive c) and writes (16 consecutive k) both move 64-byte segments (the one-thread-per-
// (k, c) form of wino_filter_kernel<MO, true> wrote 4-byte words K floats apart: 4x the time of the forward transform).
// Block = 16 k x 16 c.
template <int MO>
__global__ __launch_bounds__(256) void wino_filter_dgrad_kernel(const float* __restrict__ w, float* __restrict__ U, int K,
                                                                int C) {
    using WT = Wino<MO>;
    constexpr int TS = WT::TS;
    __shared__ float tile[TS * TS][16][17];
    const int kl = threadIdx.x >> 4, cl = threadIdx.x & 15;
    const int k0 = blockIdx.y * 16, c0 = blockIdx.x * 16;
    {
        const int k = k0 + kl, c = c0 + cl;
        float g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) g[r][q] = w[(((long)k * 3 + (2 - r)) * 3 + (2 - q)) * C + c];
        float a[TS][3];
#pragma unroll
        for (int i = 0; i < TS; ++i)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float acc = 0.f;
                WINO_DOT(acc, 3, WT::G[i][l_], g[l_][q]);
                a[i][q] = acc;
            }
#pragma unroll
        for (int i = 0; i < TS; ++i)
#pragma unroll
            for (int j = 0; j < TS; ++j) {
                float acc = 0.f;
                WINO_DOT(acc, 3, WT::G[j][l_], a[i][l_]);
                tile[TS * i + j][cl][kl] = acc;
            }
    }
    __syncthreads();
    const long KC = (long)K * C;
    const int co = threadIdx.x >> 4, ko = threadIdx.x & 15;      // consecutive threads -> consecutive k
#pragma unroll
    for (int xi = 0; xi < TS * TS; ++xi) U[(long)xi * KC + (long)(c0 + co) * K + k0 + ko] = tile[xi][co][ko];
}

// y tile = A^T M A (+ bias) (+ add): one thread per (tile, 4 output channels). stats != NULL (needs 256 % (K/4) == 0): the
// block also writes the per-channel sums of what it stored, stats[blockIdx.x][2][K] doubles (sum | sum of squares): the
// batch-norm statistics of the layer behind this convolution without a pass of their own
template <int MO>
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ Mx, const float* __restrict__ bias,
                                                          const float* __restrict__ add, float* __restrict__ y, int N,
                                                          int H, int W, int K, int TH, int TW, long T, int relu,
                                                          double* __restrict__ stats, BnFoldDev bs) {
    using WT = Wino<MO>;
    constexpr int TS = WT::TS;
    __shared__ double red[2][256][4];
    const int k4n = K / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const bool valid = idx < T * k4n;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    double dsum[4] = {0.0, 0.0, 0.0, 0.0}, dsq[4] = {0.0, 0.0, 0.0, 0.0};      // per output row in fp32, doubles from there
    if (valid) {
        const int k4 = (int)(idx % k4n);
        const long t = idx / k4n;
        const int tx = (int)(t % TW);
        const int ty = (int)((t / TW) % TH);
        const int n = (int)(t / ((long)TW * TH));
        f32x4 s[MO][TS];        // A^T m, column by column
#pragma unroll
        for (int j = 0; j < TS; ++j) {
            f32x4 m[TS];
#pragma unroll
            for (int i = 0; i < TS; ++i) m[i] = ld4(Mx + ((long)(TS * i + j) * T + t) * K + k4 * 4);
#pragma unroll
            for (int i = 0; i < MO; ++i) {
                f32x4 acc = z;
                WINO_DOT(acc, TS, WT::AT[i][l_], m[l_]);
                s[i][j] = acc;
            }
        }
        f32x4 b = z;
        if (bias) b = ld4(bias + k4 * 4);
        float bs_mu[4] = {0.f, 0.f, 0.f, 0.f}, bs_is[4] = {0.f, 0.f, 0.f, 0.f}, bs_sc[4] = {0.f, 0.f, 0.f, 0.f}, bs_sh[4] = {0.f, 0.f, 0.f, 0.f};
        if (bs.x) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                bs_mu[c] = bs.mean[k4 * 4 + c];
                bs_is[c] = bs.invstd[k4 * 4 + c];
                bs_sc[c] = (bs.gamma ? bs.gamma[k4 * 4 + c] : 1.f) * bs_is[c];
                bs_sh[c] = (bs.beta ? bs.beta[k4 * 4 + c] : 0.f) - bs_mu[c] * bs_sc[c];
            }
        }
#pragma unroll
        for (int i = 0; i < MO; ++i) {
            f32x4 ssum = z, ssq = z;
#pragma unroll
            for (int j = 0; j < MO; ++j) {
                f32x4 acc = z;
                WINO_DOT(acc, TS, WT::AT[j][l_], s[i][l_]);
                if (MO * ty + i >= H || MO * tx + j >= W) continue;     // a tile that reaches beyond the map: nothing stored or summed
                const long o = (((long)n * H + MO * ty + i) * W + MO * tx + j) * K + k4 * 4;
                acc += b;
                if (add) acc += ld4(add + o);
                if (relu) {
                    acc[0] = fmaxf(acc[0], 0.f);
                    acc[1] = fmaxf(acc[1], 0.f);
                    acc[2] = fmaxf(acc[2], 0.f);
                    acc[3] = fmaxf(acc[3], 0.f);
                }
                *(f32x4*)(y + o) = acc;
                if (bs.x) {
                    // backward sums of the batch norm whose OUTPUT gradient this pass writes (bn_bwd_partial_kernel): g = the value
                    // masked by that layer's ReLU, sums of g and g * xhat
                    const f32x4 xv = ld4(bs.x + o);
                    f32x4 g = acc;
                    if (bs.relu) {
                        if (bs.y) {
                            const f32x4 yv = ld4(bs.y + o);
#pragma unroll
                            for (int c = 0; c < 4; ++c) g[c] = yv[c] > 0.f ? g[c] : 0.f;
                        } else {
#pragma unroll
                            for (int c = 0; c < 4; ++c) g[c] = fmaf(xv[c], bs_sc[c], bs_sh[c]) > 0.f ? g[c] : 0.f;
                        }
                    }
                    ssum += g;
#pragma unroll
                    for (int c = 0; c < 4; ++c) ssq[c] += g[c] * ((xv[c] - bs_mu[c]) * bs_is[c]);
                } else {
                    ssum += acc;
                    ssq += acc * acc;
                }
            }
            if (stats) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    dsum[c] += (double)ssum[c];
                    dsq[c] += (double)ssq[c];
                }
            }
        }
    }
    if (!stats) return;
    // threads tid, tid + k4n, tid + 2 k4n, ... hold the same channels of different tiles. The MO values of an output row were added
    // in fp32; from there on the sums run in doubles (a sum of squares of 128-256 values kept in fp32 costs
    // var = E[x^2] - mean^2 three digits at |mean| / std = 30)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        red[0][threadIdx.x][c] = dsum[c];
        red[1][threadIdx.x][c] = dsq[c];
    }
    __syncthreads();
    for (int st = 128; st >= k4n; st >>= 1) {
        if ((int)threadIdx.x < st) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                red[0][threadIdx.x][c] += red[0][threadIdx.x + st][c];
                red[1][threadIdx.x][c] += red[1][threadIdx.x + st][c];
            }
        }
        __syncthreads();
    }
    if ((int)threadIdx.x < k4n) {
        double* ps = stats + (long)blockIdx.x * 2 * K + threadIdx.x * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            ps[c] = red[0][threadIdx.x][c];
            ps[K + c] = red[1][threadIdx.x][c];
        }
    }
}

// filter gradient, step 1: dM[xi][t][k] = (A dy_tile A^T)[xi], the adjoint of the output transform
template <int MO>
__global__ __launch_bounds__(256) void wino_dout_kernel(const float* __restrict__ dy, float* __restrict__ dM, int N, int H,
                                                        int W, int K, int TH, int TW, long T) {
    using WT = Wino<MO>;
    constexpr int TS = WT::TS;
    const int k4n = K / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * k4n) return;
    const int k4 = (int)(idx % k4n);
    const long t = idx / k4n;
    const int tx = (int)(t % TW);
    const int ty = (int)((t / TW) % TH);
    const int n = (int)(t / ((long)TW * TH));
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 a[TS][MO];        // A dy: a[i][j] = sum_l AT[l][i] dy[l][j]
#pragma unroll
    for (int j = 0; j < MO; ++j) {
        f32x4 d[MO];
#pragma unroll
        for (int i = 0; i < MO; ++i) {
            // (a tile that reaches beyond the map: those pixels carry no gradient; the address is clamped, the value dropped)
            const int oy = MO * ty + i, ox = MO * tx + j;
            const bool in = oy < H && ox < W;
            const f32x4 v = ld4(dy + (((long)n * H + (in ? oy : 0)) * W + (in ? ox : 0)) * K + k4 * 4);
            d[i] = in ? v : z;
        }
#pragma unroll
        for (int i = 0; i < TS; ++i) {
            f32x4 acc = z;
            WINO_DOT(acc, MO, WT::AT[l_][i], d[l_]);
            a[i][j] = acc;
        }
    }
#pragma unroll
    for (int i = 0; i < TS; ++i)
#pragma unroll
        for (int j = 0; j < TS; ++j) {
            f32x4 acc = z;
            WINO_DOT(acc, MO, WT::AT[l_][j], a[i][l_]);
            *(f32x4*)(dM + ((long)(TS * i + j) * T + t) * K + k4 * 4) = acc;
        }
}

// filter gradient, step 3: dw[k][r][s][c] = (G^T dU G)[r][s], the adjoint of the filter transform
template <int MO>
__global__ __launch_bounds__(256) void wino_dfilter_kernel(const float* __restrict__ dU, float* __restrict__ dw, int K, int C) {
    using WT = Wino<MO>;
    constexpr int TS = WT::TS;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)K * C) return;
    const int c = (int)(idx % C);
    const int k = (int)(idx / C);
    const long KC = (long)K * C;
    float r[3][TS];         // G^T u, column by column
#pragma unroll
    for (int j = 0; j < TS; ++j) {
        float u[TS];
#pragma unroll
        for (int i = 0; i < TS; ++i) u[i] = dU[(long)(TS * i + j) * KC + (long)k * C + c];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float acc = 0.f;
            WINO_DOT(acc, TS, WT::G[l_][i], u[l_]);
            r[i][j] = acc;
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float acc = 0.f;
            WINO_DOT(acc, TS, WT::G[l_][j], r[i][l_]);
            dw[(((long)k * 3 + i) * 3 + j) * C + c] = acc;
        }
}

struct WinoDims {
    int TS, NX, TH, TW;
    long T;
    size_t nU, nV, nM;
};

int wino_dims(int mo, int N, int H, int W, int C, int K, WinoDims* d) {
    DENET_CHECK_ARG(mo == 2 || mo == 4, "conv_wino: output tile must be 2 or 4 (got %d)", mo);
    DENET_CHECK_ARG(H > 0 && W > 0 && C % 32 == 0 && K % 32 == 0, "conv_wino: the channel counts must be multiples of 32");
    // a map that is no multiple of the tile is covered by ceil(H / mo) x ceil(W / mo) tiles: the input side reads zeros beyond the
    // image (it does so for the halo anyway), the output side drops the pixels beyond it (stores, sums, operand reads)
    d->TS = mo + 2;
    d->NX = d->TS * d->TS;
    d->TH = (H + mo - 1) / mo;
    d->TW = (W + mo - 1) / mo;
    d->T = (long)N * d->TH * d->TW;
    d->nU = (size_t)d->NX * C * K;
    d->nV = (size_t)d->NX * d->T * C;
    d->nM = (size_t)d->NX * d->T * K;
    DENET_CHECK_ARG(d->T < (1L << 31) / d->NX, "conv_wino: too many tiles");
    return DENET_OK;
}

#define WINO_LAUNCH(MO_, KERNEL, NTHREADS, ...)                                                                       \
    {                                                                                                               \
        const unsigned blocks_ = (unsigned)(((NTHREADS) + 255) / 256);                                              \
        if ((MO_) == 2) hipLaunchKernelGGL(KERNEL<2>, dim3(blocks_), dim3(256), 0, stream, __VA_ARGS__);             \
        else hipLaunchKernelGGL(KERNEL<4>, dim3(blocks_), dim3(256), 0, stream, __VA_ARGS__);                        \
    }

int wino_filter(int mo, bool dgrad, const float* w, float* U, int K, int C, hipStream_t stream) {
    // w: [K,3,3,C]; forward: U[xi][K][C]; data gradient: U'[xi][C][K]
    const unsigned fb = (unsigned)(((long)K * C + 255) / 256);
    if (dgrad && K % 16 == 0 && C % 16 == 0) {
        if (mo == 2) hipLaunchKernelGGL(wino_filter_dgrad_kernel<2>, dim3(C / 16, K / 16), dim3(256), 0, stream, w, U, K, C);
        else hipLaunchKernelGGL(wino_filter_dgrad_kernel<4>, dim3(C / 16, K / 16), dim3(256), 0, stream, w, U, K, C);
    } else if (dgrad) {
        if (mo == 2) hipLaunchKernelGGL((wino_filter_kernel<2, true>), dim3(fb), dim3(256), 0, stream, w, U, K, C);
        else hipLaunchKernelGGL((wino_filter_kernel<4, true>), dim3(fb), dim3(256), 0, stream, w, U, K, C);
    } else {
        if (mo == 2) hipLaunchKernelGGL((wino_filter_kernel<2, false>), dim3(fb), dim3(256), 0, stream, w, U, K, C);
        else hipLaunchKernelGGL((wino_filter_kernel<4, false>), dim3(fb), dim3(256), 0, stream, w, U, K, C);
    }
    DENET_CHECK_LAUNCH("conv_wino filter transform");
    return DENET_OK;
}

// u_cached: transformed filters prepared by denet_conv_wino_filter (NULL: transform here); v_keep: where the transformed
// input is written (NULL: inside the workspace) - the filter gradient of the same layer can reuse it
int wino_run(int mo, bool dgrad, const float* in, const float* w, const float* u_cached, float* v_keep, const float* bias,
             const float* add, float* out, float* ws, size_t ws_bytes, int N, int H, int W, int Cin, int Cout,
             hipStream_t stream, int relu = 0, double* stats = nullptr, const BnFoldDev* fold = nullptr, float* dm_out = nullptr,
             hipEvent_t transform_done = nullptr, const BnFoldDev* out_sums = nullptr, int in_up = 0) {
    // in: [N,H,W,Cin]   out: [N,H,W,Cout]   w: KRSC with (K,C) = dgrad ? (Cin,Cout) : (Cout,Cin)
    // fold: the input is formed on the fly from a batch-norm layer (wino_prep_kernel; dgrad: backward form, dm_out receives dM)
    DENET_CHECK_ARG((in || fold) && w && out && ws, "conv_wino: null pointer");
    WinoDims d;
    int rc = wino_dims(mo, N, H, W, Cin, Cout, &d);
    if (rc) return rc;
    // workspace: [U | V | M]
    DENET_CHECK_ARG(ws_bytes >= (d.nU + d.nV + d.nM) * sizeof(float), "conv_wino: workspace too small (%zu < %zu)",
                    ws_bytes, (d.nU + d.nV + d.nM) * sizeof(float));
    float* Uw = ws;
    float* V = v_keep ? v_keep : Uw + d.nU;
    float* Mx = Uw + d.nU + d.nV;
    const long kc = (long)Cin * Cout;
    const float* U = u_cached;
    if (!U) {
        rc = dgrad ? wino_filter(mo, true, w, Uw, Cin, Cout, stream) : wino_filter(mo, false, w, Uw, Cout, Cin, stream);
        if (rc) return rc;
        U = Uw;
    }
    static const int prep_lds = [] { const char* e = getenv("DENET_PREP_LDS"); return e ? atoi(e) : 1; }();
    if (fold && prep_lds && Cin % 32 == 0 && (long)N * ((d.TH + 3) / 4) * ((d.TW + 3) / 4) < 0x7FFFFFFFL) {
        DENET_CHECK_ARG(!dgrad || dm_out, "conv_wino: the backward fold needs the dM buffer");
        if (dgrad) rc = mo == 2 ? launch_prep_lds<2, true>(*fold, V, dm_out, N, H, W, Cin, d.TH, d.TW, d.T, stream)
                                : launch_prep_lds<4, true>(*fold, V, dm_out, N, H, W, Cin, d.TH, d.TW, d.T, stream);
        else rc = mo == 2 ? launch_prep_lds<2, false>(*fold, V, nullptr, N, H, W, Cin, d.TH, d.TW, d.T, stream)
                          : launch_prep_lds<4, false>(*fold, V, nullptr, N, H, W, Cin, d.TH, d.TW, d.T, stream);
        if (rc) return rc;
    } else if (fold) {
        const unsigned blocks = (unsigned)((d.T * (Cin / 4) + 255) / 256);
        if (dgrad) {
            DENET_CHECK_ARG(dm_out, "conv_wino: the backward fold needs the dM buffer");
            if (mo == 2) hipLaunchKernelGGL((wino_prep_kernel<2, true>), dim3(blocks), dim3(256), 0, stream, *fold, V, dm_out, N, H, W, Cin, d.TH, d.TW, d.T);
            else hipLaunchKernelGGL((wino_prep_kernel<4, true>), dim3(blocks), dim3(256), 0, stream, *fold, V, dm_out, N, H, W, Cin, d.TH, d.TW, d.T);
        } else {
            if (mo == 2) hipLaunchKernelGGL((wino_prep_kernel<2, false>), dim3(blocks), dim3(256), 0, stream, *fold, V, (float*)nullptr, N, H, W, Cin, d.TH, d.TW, d.T);
            else hipLaunchKernelGGL((wino_prep_kernel<4, false>), dim3(blocks), dim3(256), 0, stream, *fold, V, (float*)nullptr, N, H, W, Cin, d.TH, d.TW, d.T);
        }
    } else {
        WINO_LAUNCH(mo, wino_input_kernel, d.T * (Cin / 4), in, V, N, H, W, Cin, d.TH, d.TW, d.T, in_up);
    }
    DENET_CHECK_LAUNCH("conv_wino transforms");
    if (transform_done) {
        // dm_out is complete here: the filter-gradient chain of another stream may start while this one runs the products
        const hipError_t e = hipEventRecord(transform_done, stream);
        if (e != hipSuccess) {
            denet_set_error("conv_wino: hipEventRecord: %s", hipGetErrorString(e));
            return -(int)e;
        }
    }
    // out_sums (with stats): the sums written are the backward reductions of the batch norm whose output gradient `out` is
    BnFoldDev bs = {};
    if (out_sums && stats) bs = *out_sums;
    // (the fused kernel's epilogue writes whole 4x4 blocks: maps that are multiples of the tile only)
    if (const int tb4 = (H % mo == 0 && W % mo == 0) ? denet_wino4f_block(mo, d.T, Cin, Cout) : 0)      // products + output transform in one kernel: no M
        return denet_wino4f_run(tb4, V, U, bias, add, out, stats, bs.x, bs.y, bs.gamma, bs.beta, bs.mean, bs.invstd, bs.relu, N, H, W,
                                Cin, Cout, relu, stream);
    rc = denet_gemm_batched_nt(V, U, Mx, d.NX, (int)d.T, Cout, Cin, d.T * Cin, kc, d.T * Cout, stream);
    if (rc) return rc;
    WINO_LAUNCH(mo, wino_output_kernel, d.T * (Cout / 4), Mx, bias, add, out, N, H, W, Cout, d.TH, d.TW, d.T, relu, stats, bs);
    DENET_CHECK_LAUNCH("conv_wino output");
    return DENET_OK;
}

// rows of partial column sums the output side of a pass writes ([rows][2][Kout] doubles), 0 = this pass cannot: the fused
// F(4x4) kernel leaves one row per tile block, the output transform one per 256 threads (needs 256 % (Kout / 4) == 0)
int wino_stats_rows(int tile, int N, int H, int W, int Cin, int Kout, size_t stats_bytes) {
    const long T = (long)N * ((H + tile - 1) / tile) * ((W + tile - 1) / tile);
    long rows;
    if (const int tb4 = (H % tile == 0 && W % tile == 0) ? denet_wino4f_block(tile, T, Cin, Kout) : 0) rows = denet_wino4f_stats_rows(tb4, T);
    else {
        const int k4n = Kout / 4;
        if (!(k4n > 0 && k4n <= 256 && 256 % k4n == 0)) return 0;
        rows = (T * k4n + 255) / 256;
    }
    return stats_bytes >= (size_t)rows * 2 * Kout * sizeof(double) ? (int)rows : 0;
}

}  // namespace

// dw = filter gradient of the 3x3 stride-1 pad-1 convolution; x:[N,H,W,C] dy:[N,H,W,K] dw:[K,3,3,C].
// workspace (denet_conv_wino_workspace_bytes): dU | V | dM; split_ws: the split-K slices of the batched product.
static int wino_wgrad_run(const float* x, const float* dy, const float* dm_ready, const float* v_cached, float* dw,
                          float* workspace, size_t workspace_bytes, float* split_ws, size_t split_ws_bytes, int tile, int N,
                          int H, int W, int C, int K, hipStream_t stream) {
    DENET_CHECK_ARG((x || v_cached) && (dy || dm_ready) && dw && workspace, "conv_wino_wgrad: null pointer");
    WinoDims d;
    int rc = wino_dims(tile, N, H, W, C, K, &d);
    if (rc) return rc;
    DENET_CHECK_ARG(workspace_bytes >= (d.nU + d.nV + d.nM) * sizeof(float), "conv_wino_wgrad: workspace too small");
    float* dU = workspace;
    const float* V = v_cached;
    float* dM = dU + d.nU + d.nV;
    if (!V) {      // v_cached: the transformed input the forward pass of this layer kept (same tile)
        float* Vw = dU + d.nU;
        WINO_LAUNCH(tile, wino_input_kernel, d.T * (C / 4), x, Vw, N, H, W, C, d.TH, d.TW, d.T, 0);
        V = Vw;
    }
    const float* dMr = dm_ready;       // already formed by the backward fold of the data-gradient call (wino_prep_kernel)
    if (!dMr) {
        WINO_LAUNCH(tile, wino_dout_kernel, d.T * (K / 4), dy, dM, N, H, W, K, d.TH, d.TW, d.T);
        dMr = dM;
    }
    DENET_CHECK_LAUNCH("conv_wino_wgrad transforms");
    const int sp = denet_wino4g_splits(tile, d.T, C, K);
    if (sp && (sp == 1 || (split_ws && split_ws_bytes >= denet_wino4g_workspace_bytes(sp, C, K))))
        rc = denet_wino4g_run(sp, dMr, V, dU, split_ws, split_ws_bytes, d.T, C, K, stream);     // the operands' rows as they lie
    else
        rc = denet_wgrad_batched(V, dMr, dU, split_ws, split_ws_bytes, d.NX, (int)d.T, C, K, stream);
    if (rc) return rc;
    WINO_LAUNCH(tile, wino_dfilter_kernel, (long)K * C, dU, dw, K, C);
    DENET_CHECK_LAUNCH("conv_wino_wgrad filter");
    return DENET_OK;
}

extern "C" int denet_conv_wino_wgrad(const float* x, const float* dy, const float* v_cached, float* dw, float* workspace,
                                     size_t workspace_bytes, float* split_ws, size_t split_ws_bytes, int tile, int N, int H,
                                     int W, int C, int K, hipStream_t stream) {
    DENET_CHECK_ARG(x && dy, "conv_wino_wgrad: null pointer");
    return wino_wgrad_run(x, dy, nullptr, v_cached, dw, workspace, workspace_bytes, split_ws, split_ws_bytes, tile, N, H, W, C, K,
                          stream);
}

// the filter gradient from dM = A dy A^T already formed by denet_conv_wino_dgrad_fold: dm [(tile+2)^2][T][K]
extern "C" int denet_conv_wino_wgrad_dm(const float* x, const float* dm, const float* v_cached, float* dw, float* workspace,
                                        size_t workspace_bytes, float* split_ws, size_t split_ws_bytes, int tile, int N, int H,
                                        int W, int C, int K, hipStream_t stream) {
    DENET_CHECK_ARG(dm, "conv_wino_wgrad_dm: null pointer");
    return wino_wgrad_run(x, nullptr, dm, v_cached, dw, workspace, workspace_bytes, split_ws, split_ws_bytes, tile, N, H, W, C, K,
                          stream);
}

static int fold_from(const denet_bn_link* bn, bool bwd, BnFoldDev* f) {
    DENET_CHECK_ARG(bn && bn->x && bn->gamma && bn->mean && bn->invstd, "conv_wino fold: null pointer");
    DENET_CHECK_ARG(bwd ? (bn->aux && bn->coef && (!bn->relu || bn->y || bn->beta)) : (bn->beta && bn->out),
                    "conv_wino fold: incomplete batch-norm description");
    f->x = bn->x; f->aux = bn->aux; f->y = bn->y; f->gamma = bn->gamma; f->beta = bn->beta; f->mean = bn->mean;
    f->invstd = bn->invstd; f->coef = bn->coef; f->out = bn->out; f->relu = bn->relu;
    return DENET_OK;
}

// denet_conv_wino_fwd (+ ReLU / batch-norm column sums of the output, as denet_conv_wino_fwd_act / _stats) whose input is the
// output of the batch-norm layer `bn` evaluated on the fly; bn->out receives that activation (what denet_bn_fwd_train_pre
// would have written: bit-identical)
extern "C" int denet_conv_wino_fwd_fold(const denet_bn_link* bn, const float* w, const float* u_cached, float* v_keep,
                                        const float* bias, const float* add, float* y, int relu, double* stats_partial,
                                        size_t stats_bytes, int* stats_rows, float* workspace, size_t workspace_bytes, int tile,
                                        int N, int H, int W, int C, int K, hipStream_t stream) {
    BnFoldDev f;
    int rc = fold_from(bn, false, &f);
    if (rc) return rc;
    double* st = nullptr;
    if (stats_partial) {
        DENET_CHECK_ARG(stats_rows && !relu, "conv_wino_fwd_fold: bad statistics arguments");
        *stats_rows = wino_stats_rows(tile, N, H, W, C, K, stats_bytes);
        st = *stats_rows ? stats_partial : nullptr;
    }
    return wino_run(tile, false, nullptr, w, u_cached, v_keep, bias, add, y, workspace, workspace_bytes, N, H, W, C, K, stream,
                    relu ? 1 : 0, st, &f);
}

// denet_conv_wino_dgrad whose input dy is the gradient a batch-norm layer `bn` hands to this convolution's output, evaluated on
// the fly (bn->x = the convolution's output, bn->aux = gradient of the batch norm's output, bn->coef from denet_bn_bwd_sums);
// that gradient tensor is never written. dm_out [(tile+2)^2][T][K] receives A dy A^T for denet_conv_wino_wgrad_dm, bn->out (if
// not NULL) the masked gradient for the residual branch (the `dres` of denet_bn_bwd)
// statistics request of a data-gradient pass: dx is the gradient of the OUTPUT of the batch norm `sums_of` (x = its input,
// y = its forward output or NULL, gamma / beta / mean / invstd, relu); the output transform then also writes that layer's two
// backward reductions, stats_partial [rows][2][C] doubles (*stats_rows = 0: this channel count is not supported, no sums)
static int sums_request(const denet_bn_link* sums_of, double* stats_partial, size_t stats_bytes, int* stats_rows, int tile, int N,
                        int H, int W, int C, int K, BnFoldDev* bs, double** st) {
    *st = nullptr;
    if (!sums_of) return DENET_OK;
    DENET_CHECK_ARG(stats_partial && stats_rows, "conv_wino dgrad: the backward sums need a buffer");
    DENET_CHECK_ARG(sums_of->x && sums_of->mean && sums_of->invstd && (!sums_of->relu || sums_of->y || (sums_of->gamma && sums_of->beta)),
                    "conv_wino dgrad: incomplete batch-norm description for the backward sums");
    *stats_rows = wino_stats_rows(tile, N, H, W, K, C, stats_bytes);
    if (*stats_rows) {
        *bs = BnFoldDev{};
        bs->x = sums_of->x; bs->y = sums_of->relu ? sums_of->y : nullptr; bs->gamma = sums_of->gamma; bs->beta = sums_of->beta;
        bs->mean = sums_of->mean; bs->invstd = sums_of->invstd; bs->relu = sums_of->relu;
        *st = stats_partial;
    }
    return DENET_OK;
}

extern "C" int denet_conv_wino_dgrad_fold(const denet_bn_link* bn, float* dm_out, const float* w, const float* u_cached,
                                          const float* add, float* dx, const denet_bn_link* sums_of, double* stats_partial,
                                          size_t stats_bytes, int* stats_rows, float* workspace, size_t workspace_bytes, int tile,
                                          int N, int H, int W, int C, int K, void* transform_done_event, hipStream_t stream) {
    BnFoldDev f, bs;
    double* st;
    int rc = fold_from(bn, true, &f);
    if (rc) return rc;
    rc = sums_request(sums_of, stats_partial, stats_bytes, stats_rows, tile, N, H, W, C, K, &bs, &st);
    if (rc) return rc;
    return wino_run(tile, true, nullptr, w, u_cached, nullptr, nullptr, add, dx, workspace, workspace_bytes, N, H, W, K, C, stream,
                    0, st, &f, dm_out, (hipEvent_t)transform_done_event, st ? &bs : nullptr);
}

// denet_conv_wino_dgrad with such a statistics request
extern "C" int denet_conv_wino_dgrad_sums(const float* dy, const float* w, const float* u_cached, const float* add, float* dx,
                                          const denet_bn_link* sums_of, double* stats_partial, size_t stats_bytes, int* stats_rows,
                                          float* workspace, size_t workspace_bytes, int tile, int N, int H, int W, int C, int K,
                                          hipStream_t stream) {
    BnFoldDev bs;
    double* st;
    int rc = sums_request(sums_of, stats_partial, stats_bytes, stats_rows, tile, N, H, W, C, K, &bs, &st);
    if (rc) return rc;
    return wino_run(tile, true, dy, w, u_cached, nullptr, nullptr, add, dx, workspace, workspace_bytes, N, H, W, K, C, stream, 0, st,
                    nullptr, nullptr, nullptr, st ? &bs : nullptr);
}

// measures the launch configuration of the component GEMMs of this geometry (all three passes); synchronises
extern "C" int denet_conv_wino_tune(float* workspace, size_t workspace_bytes, float* split_ws, size_t split_ws_bytes,
                                    int tile, int N, int H, int W, int C, int K, hipStream_t stream) {
    DENET_CHECK_ARG(workspace, "conv_wino_tune: null workspace");
    WinoDims d;
    int rc = wino_dims(tile, N, H, W, C, K, &d);
    if (rc) return rc;
    DENET_CHECK_ARG(workspace_bytes >= (d.nU + d.nV + d.nM) * sizeof(float), "conv_wino_tune: workspace too small");
    float* U = workspace;
    (void)hipMemsetAsync(U, 0, (d.nU + d.nV + d.nM) * sizeof(float), stream);
    const long T = d.T;
    // forward: V [T x C] -> M [T x K];  data gradient: V [T x K] -> M [T x C]
    rc = denet_gemm_batched_tune(U + d.nU, U, U + d.nU + d.nV, d.NX, (int)T, K, C, T * C, (long)C * K, T * K, stream);
    if (rc) return rc;
    rc = denet_gemm_batched_tune(U + d.nU + d.nV, U, U + d.nU, d.NX, (int)T, C, K, T * K, (long)C * K, T * C, stream);
    if (rc || !split_ws) return rc;
    return denet_wgrad_batched_tune(U + d.nU, U + d.nU + d.nV, U, split_ws, split_ws_bytes, d.NX, (int)T, C, K, stream);
}

extern "C" size_t denet_conv_wino_workspace_bytes(int tile, int N, int H, int W, int C, int K) {
    const size_t nx = (size_t)(tile + 2) * (tile + 2);
    const size_t T = (size_t)N * ((H + tile - 1) / tile) * ((W + tile - 1) / tile);
    return (nx * C * K + nx * T * C + nx * T * K) * sizeof(float);
}

// y = conv3x3(x, w) stride 1 pad 1 (+ bias) (+ add); x:[N,H,W,C] w:[K,3,3,C] y:[N,H,W,K]; tile = 2: F(2x2,3x3), 4: F(4x4,3x3)
extern "C" int denet_conv_wino_fwd(const float* x, const float* w, const float* u_cached, float* v_keep, const float* bias,
                                   const float* add, float* y, float* workspace, size_t workspace_bytes, int tile, int N,
                                   int H, int W, int C, int K, hipStream_t stream) {
    return wino_run(tile, false, x, w, u_cached, v_keep, bias, add, y, workspace, workspace_bytes, N, H, W, C, K, stream);
}

// the same with max(., 0) after bias and add (see denet_conv_fwd_act)
extern "C" int denet_conv_wino_fwd_act(const float* x, const float* w, const float* u_cached, float* v_keep,
                                       const float* bias, const float* add, float* y, int relu, float* workspace,
                                       size_t workspace_bytes, int tile, int N, int H, int W, int C, int K,
                                       hipStream_t stream) {
    return wino_run(tile, false, x, w, u_cached, v_keep, bias, add, y, workspace, workspace_bytes, N, H, W, C, K, stream,
                    relu ? 1 : 0);
}

// denet_conv_wino_fwd whose output transform also emits the batch-norm column sums (see denet_conv_fwd_stats):
// stats_partial [rows][2][K] doubles, rows = N*(H/tile)*(W/tile)*(K/4) / 256 written to *stats_rows; *stats_rows = 0 (and no
// sums) when 256 is not a multiple of K/4 - the caller then lets the batch norm compute its own statistics
// denet_conv_wino_fwd_stats on the 2 x 2 nearest-neighbour up-sampling of x_small [N][H/2][W/2][C] (H, W: the layer's input size):
// the pool-inverse layer in front of the convolution (pool_inv.py:10-41) inside the input transform, its output never written
extern "C" int denet_conv_wino_fwd_stats_up(const float* x_small, const float* w, const float* u_cached, float* v_keep,
                                            const float* bias, const float* add, float* y, double* stats_partial,
                                            size_t stats_bytes, int* stats_rows, float* workspace, size_t workspace_bytes,
                                            int tile, int N, int H, int W, int C, int K, hipStream_t stream) {
    DENET_CHECK_ARG((tile == 2 || tile == 4) && H % 2 == 0 && W % 2 == 0, "conv_wino_fwd_stats_up: bad arguments");
    double* st = nullptr;
    if (stats_partial) {
        DENET_CHECK_ARG(stats_rows, "conv_wino_fwd_stats_up: null pointer");
        *stats_rows = wino_stats_rows(tile, N, H, W, C, K, stats_bytes);
        st = *stats_rows ? stats_partial : nullptr;
    }
    return wino_run(tile, false, x_small, w, u_cached, v_keep, bias, add, y, workspace, workspace_bytes, N, H, W, C, K, stream, 0, st,
                    nullptr, nullptr, nullptr, nullptr, 1);
}

extern "C" int denet_conv_wino_fwd_stats(const float* x, const float* w, const float* u_cached, float* v_keep,
                                         const float* bias, const float* add, float* y, double* stats_partial,
                                         size_t stats_bytes, int* stats_rows, float* workspace, size_t workspace_bytes,
                                         int tile, int N, int H, int W, int C, int K, hipStream_t stream) {
    DENET_CHECK_ARG(stats_partial && stats_rows && (tile == 2 || tile == 4), "conv_wino_fwd_stats: bad arguments");
    *stats_rows = wino_stats_rows(tile, N, H, W, C, K, stats_bytes);
    return wino_run(tile, false, x, w, u_cached, v_keep, bias, add, y, workspace, workspace_bytes, N, H, W, C, K, stream, 0,
                    *stats_rows ? stats_partial : nullptr);
}

// transformed filters of a layer, prepared ahead of its passes (e.g. for all layers on a side stream right after the
// solver step): dgrad = 0: U[xi][K][C] for denet_conv_wino_fwd, 1: U'[xi][C][K] for denet_conv_wino_dgrad
extern "C" int denet_conv_wino_filter(const float* w, float* u, int tile, int dgrad, int C, int K, hipStream_t stream) {
    DENET_CHECK_ARG(w && u && (tile == 2 || tile == 4) && C > 0 && K > 0, "conv_wino_filter: bad arguments");
    return wino_filter(tile, dgrad != 0, w, u, K, C, stream);
}

// dx = conv3x3_transposed(dy, w) (+ add); dy:[N,H,W,K] w:[K,3,3,C] dx:[N,H,W,C]
extern "C" int denet_conv_wino_dgrad(const float* dy, const float* w, const float* u_cached, const float* add, float* dx,
                                     float* workspace, size_t workspace_bytes, int tile, int N, int H, int W, int C, int K,
                                     hipStream_t stream) {
    return wino_run(tile, true, dy, w, u_cached, nullptr, nullptr, add, dx, workspace, workspace_bytes, N, H, W, K, C, stream);
}
