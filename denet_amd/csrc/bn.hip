// Spatial batch normalisation (+ fused ReLU / residual add) forward and backward for NHWC fp32.
// Reference: denet/layer/batch_norm.py:50-79 (cuDNN dnn_batch_normalization_train/test, running
// `mean` and `stdinv`), denet/layer/batch_norm_relu.py:34-54 (in-place ReLU after BN, gradient masked
// by xn > 0 then cuDNN BN-grad), denet/layer/resnet.py:109-113 (residual add + ReLU).
//
// All kernels are HBM-bound. Thread mapping: a thread owns one float4 of channels (c4) and walks
// rows; LC = gcd(C/4, 256) lanes cover consecutive channels so a wave reads >= 256 contiguous bytes.
// Per-channel reductions are accumulated in fp64 per thread, combined through LDS, written as
// per-workgroup partials and finished by a second tiny kernel: deterministic, no atomics.
#include "common.h"
#include <stdlib.h>

namespace {

struct BnMap {
    int LC;      // lanes along channels (float4 units)
    int RS;      // rows handled concurrently by one workgroup
    int gx, gy;  // grid
};

BnMap bn_map(long M, int C) {
    BnMap m;
    int c4 = C / 4;
    int lc = 1;
    while (lc < 256 && (c4 % (lc * 2)) == 0) lc *= 2;
    m.LC = lc;
    m.RS = 256 / lc;
    m.gx = c4 / lc;
    long rows_blocks = (M + m.RS - 1) / m.RS;
    int gy = 1024 / m.gx;   // ~4 workgroups per CU; keeps the second-stage reduction short
    if (gy < 1) gy = 1;
    if (gy > rows_blocks) gy = (int)rows_blocks;
    m.gy = gy;
    return m;
}

// ---- a max pool directly behind a BN + ReLU layer (the ResNet stem: convolution.py -> batch_norm_relu.py -> pool.py:38) ----
// Forward: y = relu(bn(x)) is never written - the pooled maximum (and its argmax tap, pool.hip's rule: first maximum in scan
// order) is taken over values formed on the fly. Backward: the gradient of y at an input position is the sum of the pooled
// gradients of the windows that selected it (the gather of pool.hip's maxpool_bwd_kernel, same loop order: bit-identical
// values), evaluated inside the two batch-norm gradient passes instead of being written and read back twice.
struct PoolGeom {
    int H, W, OH, OW, k, s, pad;
    FastDiv fw, fh, fow, foh, fs, fc; // row index -> (n, y, x) without 64-bit divisions (N*H*W < 2^31 is checked by the callers)
};

PoolGeom pool_geom(int H, int W, int OH, int OW, int k, int s, int pad, int C) {
    PoolGeom g = {H, W, OH, OW, k, s, pad, {}, {}, {}, {}, {}, {}};
    g.fw.init(W); g.fh.init(H); g.fow.init(OW); g.foh.init(OH); g.fs.init(s); g.fc.init(C / 4);
    return g;
}

__device__ __forceinline__ f32x4 pool_gather(const float* __restrict__ dyp, const unsigned char* __restrict__ arg, const PoolGeom& g,
                                             long r, int C, int c) {
    const uint32_t t = g.fw.div((uint32_t)r);
    const int ix = (int)((uint32_t)r - t * (uint32_t)g.W);
    const int n = (int)g.fh.div(t);
    const int iy = (int)(t - (uint32_t)n * (uint32_t)g.H);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int oy_lo = (iy + g.pad - g.k + 1 + g.s - 1);
    oy_lo = oy_lo < 0 ? 0 : (int)g.fs.div((uint32_t)oy_lo);
    int oy_hi = (int)g.fs.div((uint32_t)(iy + g.pad));
    if (oy_hi > g.OH - 1) oy_hi = g.OH - 1;
    int ox_lo = (ix + g.pad - g.k + 1 + g.s - 1);
    ox_lo = ox_lo < 0 ? 0 : (int)g.fs.div((uint32_t)ox_lo);
    int ox_hi = (int)g.fs.div((uint32_t)(ix + g.pad));
    if (ox_hi > g.OW - 1) ox_hi = g.OW - 1;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        const int ky = iy - (oy * g.s - g.pad);
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            const int kx = ix - (ox * g.s - g.pad);
            const int tap = ky * g.k + kx;
            const long o = (((long)n * g.OH + oy) * g.OW + ox) * C + c;
            const uchar4 a = *(const uchar4*)(arg + o);
            const f32x4 v = *(const f32x4*)(dyp + o);
            acc[0] += (a.x == tap) ? v[0] : 0.f;
            acc[1] += (a.y == tap) ? v[1] : 0.f;
            acc[2] += (a.z == tap) ? v[2] : 0.f;
            acc[3] += (a.w == tap) ? v[3] : 0.f;
        }
    }
    return acc;
}

// partial[gy][2][C] : sum(x), sum(x*x)
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ x, long M, int C, int LC,
                                                               double* __restrict__ partial) {
    __shared__ double red[256 * 8];
    const int tid = threadIdx.x;
    const int cl = tid % LC, rsub = tid / LC, RS = 256 / LC;
    const int c = (blockIdx.x * LC + cl) * 4;
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    for (long r = (long)blockIdx.y * RS + rsub; r < M; r += (long)gridDim.y * RS) {
        const f32x4 v = *(const f32x4*)(x + r * C + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double d = (double)v[k];
            s[k] += d;
            ss[k] += d * d;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[tid * 8 + k] = s[k];
        red[tid * 8 + 4 + k] = ss[k];
    }
    __syncthreads();
    if (rsub == 0) {
        for (int j = 1; j < RS; ++j) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s[k] += red[(j * LC + cl) * 8 + k];
                ss[k] += red[(j * LC + cl) * 8 + 4 + k];
            }
        }
        double* p = partial + (long)blockIdx.y * 2 * C;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            p[c + k] = s[k];
            p[C + c + k] = ss[k];
        }
    }
}

// second stage: a one-wave workgroup reduces FC channels, FJ = 64/FC lanes stride over the partial rows and a shuffle tree
// adds them. The stage is latency bound (a few hundred KB, a chain of dependent row loads per lane): 32 row lanes keep the
// chain at gy/128 rounds of 4 loads in flight, C/2 workgroups spread it over the chip. NO LDS on purpose: these kernels run
// beside the persistent 64-channel Winograd kernels, whose workgroups hold a CU's whole LDS - a kernel that asks for any
// waits for one of them to retire (measured: 165 us instead of 7 for the six layer-1 finals of a step).
constexpr int FC = 2, FJ = 64 / FC, FINAL_NT = 64;
__device__ __forceinline__ void reduce_row_lanes(double& s, double& ss) {
#pragma unroll
    for (int off = FC; off < 64; off <<= 1) {
        s += __shfl_xor(s, off, 64);
        ss += __shfl_xor(ss, off, 64);
    }
}
__device__ __forceinline__ void reduce_partials(const double* __restrict__ partial, int gy, int C, int c, int jl,
                                                double& s, double& ss) {
    s = 0;
    ss = 0;
    if (c < C) {
        int j = jl;
        // 4 independent row loads in flight per lane
        for (; j + 3 * FJ < gy; j += 4 * FJ) {
            const double a0 = partial[(long)j * 2 * C + c], b0 = partial[(long)j * 2 * C + C + c];
            const double a1 = partial[(long)(j + FJ) * 2 * C + c], b1 = partial[(long)(j + FJ) * 2 * C + C + c];
            const double a2 = partial[(long)(j + 2 * FJ) * 2 * C + c], b2 = partial[(long)(j + 2 * FJ) * 2 * C + C + c];
            const double a3 = partial[(long)(j + 3 * FJ) * 2 * C + c], b3 = partial[(long)(j + 3 * FJ) * 2 * C + C + c];
            s += (a0 + a1) + (a2 + a3);
            ss += (b0 + b1) + (b2 + b3);
        }
        for (; j < gy; j += FJ) {
            s += partial[(long)j * 2 * C + c];
            ss += partial[(long)j * 2 * C + C + c];
        }
    }
}

// first of two stages for long partial lists (a stem convolution leaves 16 384 rows = 16 MB, which the C/8 workgroups of the
// final kernel would walk alone): slice `blockIdx.y` of the rows -> out[slice][2][C]
__global__ __launch_bounds__(FINAL_NT) void bn_stats_fold_kernel(const double* __restrict__ partial, int gy, int per, int C,
                                                                 double* __restrict__ out) {
    const int cl = threadIdx.x % FC, jl = threadIdx.x / FC;
    const int c = blockIdx.x * FC + cl;
    const int j0 = blockIdx.y * per;
    const int n = gy - j0 < per ? gy - j0 : per;
    double s, ss;
    reduce_partials(partial + (long)j0 * 2 * C, n, C, c, jl, s, ss);
    reduce_row_lanes(s, ss);
    if (jl != 0 || c >= C) return;
    out[(long)blockIdx.y * 2 * C + c] = s;
    out[(long)blockIdx.y * 2 * C + C + c] = ss;
}

__global__ __launch_bounds__(FINAL_NT) void bn_stats_final_kernel(const double* __restrict__ partial, int gy, long M, int C,
                                                             float eps, float momentum, float* __restrict__ save_mean,
                                                             float* __restrict__ save_invstd,
                                                             float* __restrict__ run_mean,
                                                             float* __restrict__ run_stdinv) {
    const int cl = threadIdx.x % FC, jl = threadIdx.x / FC;
    const int c = blockIdx.x * FC + cl;
    double s, ss;
    reduce_partials(partial, gy, C, c, jl, s, ss);
    reduce_row_lanes(s, ss);
    if (jl != 0 || c >= C) return;
    const double mean = s / (double)M;
    double var = ss / (double)M - mean * mean;  // biased variance (cuDNN)
    if (var < 0) var = 0;
    const float fmean = (float)mean;
    const float finv = (float)(1.0 / sqrt(var + (double)eps));
    save_mean[c] = fmean;
    save_invstd[c] = finv;
    if (run_mean) {
        // batch_norm.py:75-76: mean <- m*mean + (1-m)*batch_mean ; stdinv <- m*stdinv + (1-m)*batch_invstd
        const float om = (float)(1.0 - (double)momentum);
        run_mean[c] = momentum * run_mean[c] + om * fmean;
        run_stdinv[c] = momentum * run_stdinv[c] + om * finv;
    }
}

// y = relu?( gamma*(x-mean)*invstd + beta (+ res) )
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                       float* __restrict__ y, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, long M, int C, int LC,
                                                       int relu) {
    const int tid = threadIdx.x;
    const int cl = tid % LC, rsub = tid / LC, RS = 256 / LC;
    const int c = (blockIdx.x * LC + cl) * 4;
    float sc[4], sh[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sc[k] = gamma[c + k] * invstd[c + k];
        sh[k] = beta[c + k] - mean[c + k] * sc[k];
    }
    for (long r = (long)blockIdx.y * RS + rsub; r < M; r += (long)gridDim.y * RS) {
        const long o = r * C + c;
        f32x4 v = *(const f32x4*)(x + o);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = fmaf(v[k], sc[k], sh[k]);
        if (res) v += *(const f32x4*)(res + o);
        if (relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        *(f32x4*)(y + o) = v;
    }
}

// pooled[n][oy][ox][c] = max over the window of relu(gamma*(x-mean)*invstd + beta), arg = its tap (first maximum in scan
// order, padded taps skipped): bn_apply_kernel + maxpool_fwd_kernel without the tensor in between
// xh (optional): the normalised input (x - mean) * invstd AT the argmax - with it the two reductions of the layer's backward
// are sums over the pooled tensors (denet_bn_relu_pool_bwd_sums), a quarter of the elements and no window gather
__global__ __launch_bounds__(256) void bn_apply_pool_kernel(const float* __restrict__ x, float* __restrict__ yp,
                                                            unsigned char* __restrict__ arg, float* __restrict__ xh,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, int N, int C, PoolGeom g) {
    const int C4 = C / 4;
    const long total = (long)N * g.OH * g.OW * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const uint32_t q = g.fc.div((uint32_t)i);            // pooled pixel
        const int c = (int)((uint32_t)i - q * (uint32_t)C4) * 4;
        const uint32_t t = g.fow.div(q);
        const int ox = (int)(q - t * (uint32_t)g.OW);
        const int n = (int)g.foh.div(t);
        const int oy = (int)(t - (uint32_t)n * (uint32_t)g.OH);
        float sc[4], sh[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sc[e] = gamma[c + e] * invstd[c + e];
            sh[e] = beta[c + e] - mean[c + e] * sc[e];
        }
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, bx = {0.f, 0.f, 0.f, 0.f};
        int bi[4] = {0, 0, 0, 0};
        for (int ky = 0; ky < g.k; ++ky) {
            const int iy = oy * g.s - g.pad + ky;
            if ((unsigned)iy >= (unsigned)g.H) continue;
            for (int kx = 0; kx < g.k; ++kx) {
                const int ix = ox * g.s - g.pad + kx;
                if ((unsigned)ix >= (unsigned)g.W) continue;
                const f32x4 v = *(const f32x4*)(x + (((long)n * g.H + iy) * g.W + ix) * C + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float yv = fmaxf(fmaf(v[e], sc[e], sh[e]), 0.f);
                    if (yv > best[e]) {
                        best[e] = yv;
                        bx[e] = v[e];
                        bi[e] = ky * g.k + kx;
                    }
                }
            }
        }
        *(f32x4*)(yp + i * 4) = best;
        *(uchar4*)(arg + i * 4) = make_uchar4((unsigned char)bi[0], (unsigned char)bi[1], (unsigned char)bi[2], (unsigned char)bi[3]);
        if (xh) {
            f32x4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = (bx[e] - mean[c + e]) * invstd[c + e];     // bn_bwd_partial_kernel's expression
            *(f32x4*)(xh + i * 4) = h;
        }
    }
}

// partial[gy][2][C] : sum(g), sum(g*xhat)  with g = dy * (relu ? y>0 : 1)
// relu mask: y > 0 when the forward output is given (residual-fused blocks), else recomputed from x with the
// forward's own expression fmaf(x, gamma*invstd, beta - mean*gamma*invstd) > 0 (bit-identical, saves a pass)
// POOL: dy is the gradient of the max pool behind this layer (dy [N,OH,OW,C], arg its taps): g is gathered on the fly
template <bool POOL>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ dy,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, long M, int C, int LC,
                                                             int relu, double* __restrict__ partial,
                                                             const unsigned char* __restrict__ arg, PoolGeom pg) {
    __shared__ double red[256 * 8];
    const int tid = threadIdx.x;
    const int cl = tid % LC, rsub = tid / LC, RS = 256 / LC;
    const int c = (blockIdx.x * LC + cl) * 4;
    float mu[4], is[4], sc[4], sh[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        mu[k] = mean[c + k];
        is[k] = invstd[c + k];
        sc[k] = gamma[c + k] * is[k];
        sh[k] = (beta ? beta[c + k] : 0.f) - mu[k] * sc[k];
    }
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    for (long r = (long)blockIdx.y * RS + rsub; r < M; r += (long)gridDim.y * RS) {
        const long o = r * C + c;
        const f32x4 xv = *(const f32x4*)(x + o);
        f32x4 g = POOL ? pool_gather(dy, arg, pg, r, C, c) : *(const f32x4*)(dy + o);
        if (relu) {
            if (y) {
                const f32x4 yv = *(const f32x4*)(y + o);
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
            s[k] += (double)g[k];
            ss[k] += (double)g[k] * (double)xh;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[tid * 8 + k] = s[k];
        red[tid * 8 + 4 + k] = ss[k];
    }
    __syncthreads();
    if (rsub == 0) {
        for (int j = 1; j < RS; ++j) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s[k] += red[(j * LC + cl) * 8 + k];
                ss[k] += red[(j * LC + cl) * 8 + 4 + k];
            }
        }
        double* p = partial + (long)blockIdx.y * 2 * C;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            p[c + k] = s[k];
            p[C + c + k] = ss[k];
        }
    }
}

// dbeta = sum g ; dgamma = sum g*xhat ; coef[0][c] = dbeta/M ; coef[1][c] = dgamma/M
__global__ __launch_bounds__(FINAL_NT) void bn_bwd_final_kernel(const double* __restrict__ partial, int gy, long M, int C,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           float* __restrict__ coef) {
    const int cl = threadIdx.x % FC, jl = threadIdx.x / FC;
    const int c = blockIdx.x * FC + cl;
    double s, ss;
    reduce_partials(partial, gy, C, c, jl, s, ss);
    reduce_row_lanes(s, ss);
    if (jl != 0 || c >= C) return;
    dbeta[c] = (float)s;
    dgamma[c] = (float)ss;
    coef[c] = (float)(s / (double)M);
    coef[C + c] = (float)(ss / (double)M);
}

// dx = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)) ; optional dres = g
template <bool POOL>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ dy,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ coef, float* __restrict__ dx,
                                                           float* __restrict__ dres, long M, int C, int LC, int relu,
                                                           const unsigned char* __restrict__ arg, PoolGeom pg) {
    const int tid = threadIdx.x;
    const int cl = tid % LC, rsub = tid / LC, RS = 256 / LC;
    const int c = (blockIdx.x * LC + cl) * 4;
    float mu[4], is[4], sc[4], sh[4], mg[4], mgx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        mu[k] = mean[c + k];
        is[k] = invstd[c + k];
        sc[k] = gamma[c + k] * is[k];
        sh[k] = (beta ? beta[c + k] : 0.f) - mu[k] * sc[k];
        mg[k] = coef[c + k];
        mgx[k] = coef[C + c + k];
    }
    for (long r = (long)blockIdx.y * RS + rsub; r < M; r += (long)gridDim.y * RS) {
        const long o = r * C + c;
        const f32x4 xv = *(const f32x4*)(x + o);
        f32x4 g = POOL ? pool_gather(dy, arg, pg, r, C, c) : *(const f32x4*)(dy + o);
        if (relu) {
            if (y) {
                const f32x4 yv = *(const f32x4*)(y + o);
#pragma unroll
                for (int k = 0; k < 4; ++k) g[k] = yv[k] > 0.f ? g[k] : 0.f;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) g[k] = fmaf(xv[k], sc[k], sh[k]) > 0.f ? g[k] : 0.f;
            }
        }
        f32x4 d;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (xv[k] - mu[k]) * is[k];
            d[k] = sc[k] * (g[k] - mg[k] - xh * mgx[k]);
        }
        *(f32x4*)(dx + o) = d;
        if (dres) *(f32x4*)(dres + o) = g;
    }
}

// The pooled form of bn_bwd_apply_kernel for the stem's 3x3 / stride 2 / pad 1 max pool over an even map: a thread owns the
// 2 x 2 input pixels (2 qy + {0, 1}, 2 qx + {0, 1}) of 4 channels. The windows that contain them are (qy, qx), (qy, qx + 1),
// (qy + 1, qx), (qy + 1, qx + 1) - four (gradient, argmax) loads for four pixels instead of nine, no divisions; an even pixel is
// in one window, an odd-even pair in two, the odd-odd pixel in all four. The gathered gradient adds the windows in
// pool_gather's order (window row, then column), so the result is bit-identical to the general kernel.
__global__ __launch_bounds__(256) void bn_bwd_apply_pool_quad_kernel(const float* __restrict__ x, const float* __restrict__ dyp,
                                                                     const unsigned char* __restrict__ arg,
                                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                     const float* __restrict__ coef, float* __restrict__ dx, int N,
                                                                     int H, int W, int C, FastDiv fc, FastDiv fow, FastDiv foh) {
    const int OH = H / 2, OW = W / 2, C4 = C / 4;
    const long total = (long)N * OH * OW * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const uint32_t q = fc.div((uint32_t)i);
        const int c = (int)((uint32_t)i - q * (uint32_t)C4) * 4;
        const uint32_t t = fow.div(q);
        const int qx = (int)(q - t * (uint32_t)OW);
        const int n = (int)foh.div(t);
        const int qy = (int)(t - (uint32_t)n * (uint32_t)OH);
        float mu[4], is[4], sc[4], sh[4], mg[4], mgx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mu[k] = mean[c + k];
            is[k] = invstd[c + k];
            sc[k] = gamma[c + k] * is[k];
            sh[k] = (beta ? beta[c + k] : 0.f) - mu[k] * sc[k];
            mg[k] = coef[c + k];
            mgx[k] = coef[C + c + k];
        }
        // the four windows: w[a][b] = window (qy + a, qx + b); outside the pooled map: no window (tap 255 matches nothing)
        f32x4 wv[2][2];
        uchar4 wa[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const bool in = qy + a < OH && qx + b < OW;
                const long o = (((long)n * OH + (in ? qy + a : qy)) * OW + (in ? qx + b : qx)) * C + c;
                wv[a][b] = *(const f32x4*)(dyp + o);
                const uchar4 av = *(const uchar4*)(arg + o);
                wa[a][b] = in ? av : make_uchar4(255, 255, 255, 255);
            }
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                // window (qy + a, qx + b) holds pixel (2 qy + py, 2 qx + px) at tap (py + 1 - 2 a, px + 1 - 2 b) when that is in 0..2
                f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int a = 0; a <= py; ++a)
#pragma unroll
                    for (int b = 0; b <= px; ++b) {
                        const int tap = (py + 1 - 2 * a) * 3 + (px + 1 - 2 * b);
                        const bool in = qy + a < OH && qx + b < OW;
                        if (in) {                   // (pool_gather skips windows outside the map: no "+ 0" for them either)
                            g[0] += (wa[a][b].x == tap) ? wv[a][b][0] : 0.f;
                            g[1] += (wa[a][b].y == tap) ? wv[a][b][1] : 0.f;
                            g[2] += (wa[a][b].z == tap) ? wv[a][b][2] : 0.f;
                            g[3] += (wa[a][b].w == tap) ? wv[a][b][3] : 0.f;
                        }
                    }
                const long o = (((long)n * H + 2 * qy + py) * W + 2 * qx + px) * C + c;
                const f32x4 xv = *(const f32x4*)(x + o);
#pragma unroll
                for (int k = 0; k < 4; ++k) g[k] = fmaf(xv[k], sc[k], sh[k]) > 0.f ? g[k] : 0.f;
                f32x4 d;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float xh = (xv[k] - mu[k]) * is[k];
                    d[k] = sc[k] * (g[k] - mg[k] - xh * mgx[k]);
                }
                *(f32x4*)(dx + o) = d;
            }
    }
}

// inference transform: batch_norm.py:50-52 feeds var = (1/stdinv)^2 to cuDNN which adds eps again
__global__ void bn_test_coef_kernel(const float* __restrict__ run_mean, const float* __restrict__ run_stdinv, int C,
                                    float eps, float* __restrict__ mean_out, float* __restrict__ invstd_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float si = run_stdinv[c];
    const float var = (1.0f / si) * (1.0f / si);
    mean_out[c] = run_mean[c];
    invstd_out[c] = 1.0f / sqrtf(var + eps);
}

}  // namespace

// inference: batch norm folded into the filters of the convolution in front of it. With the test-mode transform
// y = gamma * (x - mean) * inv + beta, inv = 1 / sqrt((1/stdinv)^2 + eps) (bn_test_coef_kernel: the reference's double
// epsilon), conv(x, w) + b followed by BN equals conv(x, w * s[k]) + (beta - (mean - b) * s), s = gamma * inv.
__global__ __launch_bounds__(256) void bn_fold_kernel(const float* __restrict__ w, const float* __restrict__ conv_bias,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ run_mean, const float* __restrict__ run_stdinv,
                                                      float eps, float* __restrict__ w_out, float* __restrict__ b_out, int K,
                                                      long per_k) {
    const int k = blockIdx.x;
    const float sd = 1.0f / run_stdinv[k];
    const float inv = 1.0f / sqrtf(sd * sd + eps);
    const float s = gamma[k] * inv;
    for (long i = threadIdx.x; i < per_k; i += blockDim.x) w_out[(long)k * per_k + i] = w[(long)k * per_k + i] * s;
    if (threadIdx.x == 0) {
        const float b = conv_bias ? conv_bias[k] : 0.f;
        b_out[k] = beta[k] - (run_mean[k] - b) * s;
    }
}

extern "C" size_t denet_bn_workspace_bytes(long M, int C) {
    if (C <= 0 || C % 4) return 0;
    BnMap m = bn_map(M, C);
    // partials (at least the 64 rows of a folded list) + 2*C floats of coefficients (backward) / test-mode coefficients
    return (size_t)(m.gy > 64 ? m.gy : 64) * 2 * C * sizeof(double) + (size_t)2 * C * sizeof(float);
}

extern "C" int denet_bn_fwd_train(const float* x, const float* res, float* y, const float* gamma, const float* beta,
                                  float* run_mean, float* run_stdinv, float* save_mean, float* save_invstd,
                                  void* workspace, long M, int C, float momentum, float eps, int relu,
                                  hipStream_t stream) {
    DENET_CHECK_ARG(x && y && gamma && beta && save_mean && save_invstd && workspace, "bn_fwd_train: null pointer");
    DENET_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0, "bn_fwd_train: bad shape M=%ld C=%d", M, C);
    BnMap m = bn_map(M, C);
    double* partial = (double*)workspace;
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(m.gx, m.gy), dim3(256), 0, stream, x, M, C, m.LC, partial);
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3((C + FC - 1) / FC), dim3(FINAL_NT), 0, stream, partial, m.gy, M, C, eps,
                       momentum, save_mean, save_invstd, run_mean, run_stdinv);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(m.gx, m.gy), dim3(256), 0, stream, x, res, y, gamma, beta, save_mean,
                       save_invstd, M, C, m.LC, relu);
    DENET_CHECK_LAUNCH("bn_fwd_train");
    return DENET_OK;
}

// batch norm whose per-channel sums were already produced by the convolution in front of it (denet_conv_fwd_stats /
// denet_conv_wino_fwd_stats): partial [rows][2][C] doubles (sum | sum of squares over disjoint row sets covering all M rows)
extern "C" int denet_bn_fwd_train_pre(const float* x, const float* res, float* y, const float* gamma, const float* beta,
                                      float* run_mean, float* run_stdinv, float* save_mean, float* save_invstd,
                                      const double* partial, int rows, long M, int C, float momentum, float eps, int relu,
                                      hipStream_t stream) {
    DENET_CHECK_ARG(x && y && gamma && beta && save_mean && save_invstd && partial && rows > 0, "bn_fwd_train_pre: bad arguments");
    DENET_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0, "bn_fwd_train_pre: bad shape M=%ld C=%d", M, C);
    BnMap m = bn_map(M, C);
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3((C + FC - 1) / FC), dim3(FINAL_NT), 0, stream, partial, rows, M, C, eps,
                       momentum, save_mean, save_invstd, run_mean, run_stdinv);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(m.gx, m.gy), dim3(256), 0, stream, x, res, y, gamma, beta, save_mean,
                       save_invstd, M, C, m.LC, relu);
    DENET_CHECK_LAUNCH("bn_fwd_train_pre");
    return DENET_OK;
}

extern "C" int denet_bn_stats_final(const double* partial, int rows, long M, int C, float momentum, float eps, float* run_mean,
                                    float* run_stdinv, float* save_mean, float* save_invstd, hipStream_t stream) {
    DENET_CHECK_ARG(partial && rows > 0 && save_mean && save_invstd && M > 0 && C > 0, "bn_stats_final: bad arguments");
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3((C + FC - 1) / FC), dim3(FINAL_NT), 0, stream, partial, rows, M, C, eps,
                       momentum, save_mean, save_invstd, run_mean, run_stdinv);
    DENET_CHECK_LAUNCH("bn_stats_final");
    return DENET_OK;
}

extern "C" int denet_bn_apply(const float* x, const float* res, float* y, const float* gamma, const float* beta,
                              const float* save_mean, const float* save_invstd, long M, int C, int relu, hipStream_t stream) {
    DENET_CHECK_ARG(x && y && gamma && beta && save_mean && save_invstd, "bn_apply: null pointer");
    DENET_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0, "bn_apply: bad shape M=%ld C=%d", M, C);
    BnMap m = bn_map(M, C);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(m.gx, m.gy), dim3(256), 0, stream, x, res, y, gamma, beta, save_mean,
                       save_invstd, M, C, m.LC, relu);
    DENET_CHECK_LAUNCH("bn_apply");
    return DENET_OK;
}

extern "C" int denet_bn_bwd_sums(const float* x, const float* y, const float* dy, const float* gamma, const float* beta,
                                 const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, float* coef,
                                 void* workspace, long M, int C, int relu, hipStream_t stream) {
    DENET_CHECK_ARG(x && dy && gamma && save_mean && save_invstd && dgamma && dbeta && coef && workspace, "bn_bwd_sums: null pointer");
    DENET_CHECK_ARG(!relu || y || beta, "bn_bwd_sums: the relu mask needs the forward output y, or beta to recompute it");
    DENET_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0, "bn_bwd_sums: bad shape M=%ld C=%d", M, C);
    BnMap m = bn_map(M, C);
    double* partial = (double*)workspace;
    hipLaunchKernelGGL(bn_bwd_partial_kernel<false>, dim3(m.gx, m.gy), dim3(256), 0, stream, x, y, dy, gamma, beta, save_mean,
                       save_invstd, M, C, m.LC, relu, partial, (const unsigned char*)nullptr, PoolGeom{});
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((C + FC - 1) / FC), dim3(FINAL_NT), 0, stream, partial, m.gy, M, C, dgamma,
                       dbeta, coef);
    DENET_CHECK_LAUNCH("bn_bwd_sums");
    return DENET_OK;
}

// the second stage of the two backward reductions alone: partial [rows][2][C] doubles (sum(g) | sum(g * xhat) over disjoint row
// sets), written by bn_bwd_partial_kernel or by the data-gradient pass that produced the gradient (denet_conv_wino_dgrad_sums,
// denet_conv_wino2f_sums) -> dgamma, dbeta, coef [2][C]
extern "C" int denet_bn_bwd_final(const double* partial, int rows, long M, int C, float* dgamma, float* dbeta, float* coef,
                                  hipStream_t stream) {
    DENET_CHECK_ARG(partial && rows > 0 && dgamma && dbeta && coef && M > 0 && C > 0, "bn_bwd_final: bad arguments");
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((C + FC - 1) / FC), dim3(FINAL_NT), 0, stream, partial, rows, M, C, dgamma, dbeta, coef);
    DENET_CHECK_LAUNCH("bn_bwd_final");
    return DENET_OK;
}

extern "C" int denet_bn_bwd_apply(const float* x, const float* y, const float* dy, const float* gamma, const float* beta,
                                  const float* save_mean, const float* save_invstd, const float* coef, float* dx, float* dres,
                                  long M, int C, int relu, hipStream_t stream) {
    DENET_CHECK_ARG(x && dy && gamma && save_mean && save_invstd && coef && dx, "bn_bwd_apply: null pointer");
    DENET_CHECK_ARG(!relu || y || beta, "bn_bwd_apply: the relu mask needs the forward output y, or beta to recompute it");
    DENET_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0, "bn_bwd_apply: bad shape M=%ld C=%d", M, C);
    BnMap m = bn_map(M, C);
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(m.gx, m.gy), dim3(256), 0, stream, x, y, dy, gamma, beta, save_mean,
                       save_invstd, coef, dx, dres, M, C, m.LC, relu, (const unsigned char*)nullptr, PoolGeom{});
    DENET_CHECK_LAUNCH("bn_bwd_apply");
    return DENET_OK;
}

extern "C" int denet_bn_fwd_test(const float* x, const float* res, float* y, const float* gamma, const float* beta,
                                 const float* run_mean, const float* run_stdinv, float* coef, int coef_ready, long M, int C,
                                 float eps, int relu, hipStream_t stream) {
    // coef: 2*C floats owned by the caller (mean | inverse std of the inference transform). coef_ready != 0: they were
    // written by an earlier call and the running statistics have not changed since (inference: one kernel per layer)
    DENET_CHECK_ARG(x && y && gamma && beta && run_mean && run_stdinv && coef, "bn_fwd_test: null pointer");
    DENET_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0, "bn_fwd_test: bad shape M=%ld C=%d", M, C);
    BnMap m = bn_map(M, C);
    if (!coef_ready)
        hipLaunchKernelGGL(bn_test_coef_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, run_mean, run_stdinv, C, eps,
                           coef, coef + C);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(m.gx, m.gy), dim3(256), 0, stream, x, res, y, gamma, beta, coef, coef + C,
                       M, C, m.LC, relu);
    DENET_CHECK_LAUNCH("bn_fwd_test");
    return DENET_OK;
}

extern "C" int denet_bn_bwd(const float* x, const float* y, const float* dy, const float* gamma, const float* beta,
                            const float* save_mean, const float* save_invstd, float* dx, float* dres, float* dgamma,
                            float* dbeta, void* workspace, long M, int C, int relu, hipStream_t stream) {
    DENET_CHECK_ARG(x && dy && gamma && save_mean && save_invstd && dx && dgamma && dbeta && workspace,
                    "bn_bwd: null pointer");
    DENET_CHECK_ARG(!relu || y || beta, "bn_bwd: the relu mask needs the forward output y, or beta to recompute it");
    DENET_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0, "bn_bwd: bad shape M=%ld C=%d", M, C);
    BnMap m = bn_map(M, C);
    double* partial = (double*)workspace;
    float* coef = (float*)(partial + (size_t)m.gy * 2 * C);
    hipLaunchKernelGGL(bn_bwd_partial_kernel<false>, dim3(m.gx, m.gy), dim3(256), 0, stream, x, y, dy, gamma, beta, save_mean,
                       save_invstd, M, C, m.LC, relu, partial, (const unsigned char*)nullptr, PoolGeom{});
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((C + FC - 1) / FC), dim3(FINAL_NT), 0, stream, partial, m.gy, M, C, dgamma,
                       dbeta, coef);
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(m.gx, m.gy), dim3(256), 0, stream, x, y, dy, gamma, beta, save_mean,
                       save_invstd, coef, dx, dres, M, C, m.LC, relu, (const unsigned char*)nullptr, PoolGeom{});
    DENET_CHECK_LAUNCH("bn_bwd");
    return DENET_OK;
}

// BN + ReLU + max pool in one (training): statistics as denet_bn_fwd_train (partial = NULL: measured here, workspace =
// denet_bn_workspace_bytes(M, C)) or denet_bn_fwd_train_pre (partial / rows from the convolution in front; workspace optional:
// with it a list of more than 2048 rows is folded in two stages); the pooled
// output y_pool [N,OH,OW,C] and its argmax taps are written, relu(bn(x)) itself is not (batch_norm_relu.py:34-48 + pool.py:38)
extern "C" int denet_bn_relu_pool_fwd_train_xhat(const float* x, float* y_pool, unsigned char* argmax, float* xhat_pool,
                                                 const float* gamma, const float* beta, float* run_mean, float* run_stdinv,
                                                 float* save_mean, float* save_invstd, const double* partial, int rows,
                                                 void* workspace, int N, int H, int W, int C, int OH, int OW, int k, int stride,
                                                 int pad, float momentum, float eps, hipStream_t stream);
extern "C" int denet_bn_relu_pool_fwd_train(const float* x, float* y_pool, unsigned char* argmax, const float* gamma,
                                            const float* beta, float* run_mean, float* run_stdinv, float* save_mean,
                                            float* save_invstd, const double* partial, int rows, void* workspace, int N, int H,
                                            int W, int C, int OH, int OW, int k, int stride, int pad, float momentum, float eps,
                                            hipStream_t stream) {
    return denet_bn_relu_pool_fwd_train_xhat(x, y_pool, argmax, nullptr, gamma, beta, run_mean, run_stdinv, save_mean, save_invstd,
                                             partial, rows, workspace, N, H, W, C, OH, OW, k, stride, pad, momentum, eps, stream);
}

// the same; xhat_pool (optional) [N,OH,OW,C]: the normalised input at each window's argmax, for denet_bn_relu_pool_bwd_sums
extern "C" int denet_bn_relu_pool_fwd_train_xhat(const float* x, float* y_pool, unsigned char* argmax, float* xhat_pool,
                                                 const float* gamma, const float* beta, float* run_mean, float* run_stdinv,
                                                 float* save_mean, float* save_invstd, const double* partial, int rows,
                                                 void* workspace, int N, int H, int W, int C, int OH, int OW, int k, int stride,
                                                 int pad, float momentum, float eps, hipStream_t stream) {
    DENET_CHECK_ARG(x && y_pool && argmax && gamma && beta && save_mean && save_invstd && (partial || workspace),
                    "bn_relu_pool_fwd_train: null pointer");
    DENET_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && k > 0 && k * k <= 255 && stride > 0 && pad >= 0 && pad < k &&
                    (long)N * H * W < (1L << 31), "bn_relu_pool_fwd_train: bad arguments");
    const long M = (long)N * H * W;
    BnMap m = bn_map(M, C);
    if (!partial) {
        double* p = (double*)workspace;
        hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(m.gx, m.gy), dim3(256), 0, stream, x, M, C, m.LC, p);
        partial = p;
        rows = m.gy;
    } else if (rows > 2048 && workspace) {
        // a long list from the convolution's epilogue (one row per 128 output pixels): folded to 64 rows by 64 x C/8 workgroups first
        const int per = (rows + 63) / 64;
        const int slices = (rows + per - 1) / per;
        double* folded = (double*)workspace;
        hipLaunchKernelGGL(bn_stats_fold_kernel, dim3((C + FC - 1) / FC, slices), dim3(FINAL_NT), 0, stream, partial, rows, per, C, folded);
        partial = folded;
        rows = slices;
    }
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3((C + FC - 1) / FC), dim3(FINAL_NT), 0, stream, partial, rows, M, C, eps,
                       momentum, save_mean, save_invstd, run_mean, run_stdinv);
    const PoolGeom g = pool_geom(H, W, OH, OW, k, stride, pad, C);
    const long total = (long)N * OH * OW * (C / 4);
    long blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(bn_apply_pool_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, y_pool, argmax, xhat_pool, gamma, beta,
                       save_mean, save_invstd, N, C, g);
    DENET_CHECK_LAUNCH("bn_relu_pool_fwd_train");
    return DENET_OK;
}

// the pointwise pass of the fused layer's backward: the quad kernel where the pool is the stem's (3x3, stride 2, pad 1, even map;
// DENET_POOL_QUAD=0: the general kernel), bit-identical either way
static void launch_pool_apply(const float* x, const float* dy_pool, const unsigned char* argmax, const float* gamma, const float* beta,
                              const float* save_mean, const float* save_invstd, const float* coef, float* dx, int N, int H, int W, int C,
                              int OH, int OW, int k, int stride, int pad, hipStream_t stream) {
    static int quad = -1;
    if (quad < 0) {
        const char* e = getenv("DENET_POOL_QUAD");
        quad = e ? atoi(e) : 1;
    }
    if (quad && k == 3 && stride == 2 && pad == 1 && H % 2 == 0 && W % 2 == 0 && OH == H / 2 && OW == W / 2) {
        FastDiv fc, fow, foh;
        fc.init(C / 4); fow.init(OW); foh.init(OH);
        const long total = (long)N * OH * OW * (C / 4);
        long blocks = (total + 255) / 256;
        if (blocks > 65536) blocks = 65536;
        hipLaunchKernelGGL(bn_bwd_apply_pool_quad_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, dy_pool, argmax, gamma, beta,
                           save_mean, save_invstd, coef, dx, N, H, W, C, fc, fow, foh);
        return;
    }
    const long M = (long)N * H * W;
    BnMap m = bn_map(M, C);
    const PoolGeom g = pool_geom(H, W, OH, OW, k, stride, pad, C);
    hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(m.gx, m.gy), dim3(256), 0, stream, x, (const float*)nullptr, dy_pool,
                       gamma, beta, save_mean, save_invstd, coef, dx, (float*)nullptr, M, C, m.LC, 1, argmax, g);
}

// its gradient: dy_pool [N,OH,OW,C] + argmax -> dx [N,H,W,C], dgamma, dbeta (masked BN gradient of the gathered pool gradient)
extern "C" int denet_bn_relu_pool_bwd(const float* x, const float* dy_pool, const unsigned char* argmax, const float* gamma,
                                      const float* beta, const float* save_mean, const float* save_invstd, float* dx,
                                      float* dgamma, float* dbeta, void* workspace, int N, int H, int W, int C, int OH, int OW,
                                      int k, int stride, int pad, hipStream_t stream) {
    DENET_CHECK_ARG(x && dy_pool && argmax && gamma && beta && save_mean && save_invstd && dx && dgamma && dbeta && workspace,
                    "bn_relu_pool_bwd: null pointer");
    DENET_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && k > 0 && stride > 0 && pad >= 0 && (long)N * H * W < (1L << 31),
                    "bn_relu_pool_bwd: bad arguments");
    const long M = (long)N * H * W;
    BnMap m = bn_map(M, C);
    double* partial = (double*)workspace;
    float* coef = (float*)(partial + (size_t)m.gy * 2 * C);
    const PoolGeom g = pool_geom(H, W, OH, OW, k, stride, pad, C);
    hipLaunchKernelGGL(bn_bwd_partial_kernel<true>, dim3(m.gx, m.gy), dim3(256), 0, stream, x, (const float*)nullptr, dy_pool,
                       gamma, beta, save_mean, save_invstd, M, C, m.LC, 1, partial, argmax, g);
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((C + FC - 1) / FC), dim3(FINAL_NT), 0, stream, partial, m.gy, M, C, dgamma,
                       dbeta, coef);
    launch_pool_apply(x, dy_pool, argmax, gamma, beta, save_mean, save_invstd, coef, dx, N, H, W, C, OH, OW, k, stride, pad, stream);
    DENET_CHECK_LAUNCH("bn_relu_pool_bwd");
    return DENET_OK;
}

// The backward pass of the same layer in two calls, its reductions over the POOLED tensors. With g[p] = the pool gradient routed
// to input pixel p and masked by the ReLU: sum_p g[p] = sum_w dy_pool[w] * [y_pool[w] > 0] and sum_p g[p] * xhat[p] = sum_w
// dy_pool[w] * [y_pool[w] > 0] * xhat_pool[w] (every window w sends its gradient to its argmax pixel, whose ReLU output IS
// y_pool[w]): a quarter of the elements, no window gather. zeros / ones: [C] constant vectors (the kernel shared with
// denet_bn_bwd_sums normalises its "x" - here already normalised - with mean 0, invstd 1). The sums may also come from the
// data-gradient pass that wrote dy_pool (denet_conv_wino2f_sums with sums_of = {x: xhat_pool, y: y_pool, mean: zeros, invstd: ones,
// relu}) and go through denet_bn_bwd_final(partial, rows, N*H*W, ...). coef [2][C]: the two means over the N*H*W INPUT pixels.
extern "C" int denet_bn_relu_pool_bwd_sums(const float* xhat_pool, const float* y_pool, const float* dy_pool, const float* zeros,
                                           const float* ones, float* dgamma, float* dbeta, float* coef, void* workspace, int N,
                                           int H, int W, int C, int OH, int OW, hipStream_t stream) {
    DENET_CHECK_ARG(xhat_pool && y_pool && dy_pool && zeros && ones && dgamma && dbeta && coef && workspace,
                    "bn_relu_pool_bwd_sums: null pointer");
    DENET_CHECK_ARG(N > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && C > 0 && C % 4 == 0, "bn_relu_pool_bwd_sums: bad arguments");
    const long Mp = (long)N * OH * OW;
    BnMap m = bn_map(Mp, C);
    double* partial = (double*)workspace;
    hipLaunchKernelGGL(bn_bwd_partial_kernel<false>, dim3(m.gx, m.gy), dim3(256), 0, stream, xhat_pool, y_pool, dy_pool, ones,
                       (const float*)nullptr, zeros, ones, Mp, C, m.LC, 1, partial, (const unsigned char*)nullptr, PoolGeom{});
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((C + FC - 1) / FC), dim3(FINAL_NT), 0, stream, partial, m.gy, (long)N * H * W, C,
                       dgamma, dbeta, coef);
    DENET_CHECK_LAUNCH("bn_relu_pool_bwd_sums");
    return DENET_OK;
}

// dx [N,H,W,C] from dy_pool + argmax and the two means in coef (denet_bn_relu_pool_bwd's pointwise pass)
extern "C" int denet_bn_relu_pool_bwd_apply(const float* x, const float* dy_pool, const unsigned char* argmax, const float* gamma,
                                            const float* beta, const float* save_mean, const float* save_invstd, const float* coef,
                                            float* dx, int N, int H, int W, int C, int OH, int OW, int k, int stride, int pad,
                                            hipStream_t stream) {
    DENET_CHECK_ARG(x && dy_pool && argmax && gamma && beta && save_mean && save_invstd && coef && dx, "bn_relu_pool_bwd_apply: null pointer");
    DENET_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && k > 0 && stride > 0 && pad >= 0 && (long)N * H * W < (1L << 31),
                    "bn_relu_pool_bwd_apply: bad arguments");
    launch_pool_apply(x, dy_pool, argmax, gamma, beta, save_mean, save_invstd, coef, dx, N, H, W, C, OH, OW, k, stride, pad, stream);
    DENET_CHECK_LAUNCH("bn_relu_pool_bwd_apply");
    return DENET_OK;
}

// w: [K][per_k] filters (KRSC, per_k = R*S*C), conv_bias: [K] or NULL -> w_out, b_out of the folded convolution
extern "C" int denet_bn_fold(const float* w, const float* conv_bias, const float* gamma, const float* beta,
                             const float* run_mean, const float* run_stdinv, float eps, float* w_out, float* b_out, int K,
                             long per_k, hipStream_t stream) {
    DENET_CHECK_ARG(w && gamma && beta && run_mean && run_stdinv && w_out && b_out, "bn_fold: null pointer");
    DENET_CHECK_ARG(K > 0 && per_k > 0, "bn_fold: bad shape");
    hipLaunchKernelGGL(bn_fold_kernel, dim3(K), dim3(256), 0, stream, w, conv_bias, gamma, beta, run_mean, run_stdinv, eps,
                       w_out, b_out, K, per_k);
    DENET_CHECK_LAUNCH("bn_fold");
    return DENET_OK;
}
