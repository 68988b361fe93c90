// Implicit-GEMM convolution for gfx950 (MI355X): forward, data-gradient and weight-gradient of the
// DeNet `C` layer (reference: denet/layer/convolution.py:80-83 + tensor.grad model_cnn.py:318, which
// lower to cuDNN conv fwd / bwd-data / bwd-filter in the reference).
//
// Layout (HBM):   activations NHWC fp32, filters KRSC fp32 (already flipped, so the kernel computes a
//                 correlation; the host converts from the reference's OIHW true-convolution filters).
// Arithmetic:     v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain), accumulators in registers.
// Tiling:         one 256-thread workgroup (4 wave64) computes a BM x BN output tile, K is walked in
//                 chunks of 32; operands are staged global -> VGPR -> LDS (double buffered, one barrier
//                 per chunk). Two LDS layouts:
//                   K-inner  [rows][32+4]  read with ds_read_b128 (4 consecutive k per lane-half)
//                   K-outer  [32][cols]    read with ds_read_b32  (lanes walk the contiguous dim)
//                 fwd:   A = im2col(x) K-inner,  B = w            K-inner
//                 dgrad: A = im2col(dy) K-inner, B = w (per tap)  K-outer   (no transposed filter copy)
//                 wgrad: A = dy^T K-outer,       B = im2col(x)    K-outer   (+ split-K over pixels)
//                 Inside an 8-wide k block lane-half h consumes k = 4h..4h+3 for both operands, so the
//                 reduction order is a fixed permutation of k (legal: the sum is over the same terms).
// Launch:         1-D grid over tiles with an XCD-aware remap (8 XCDs, private L2 each).
#include "common.h"
#include "bn_final.h"
#include "../../include/denet_hip.h"
#include <map>
#include <vector>
#include <type_traits>
#include <stdlib.h>

namespace {

constexpr int BK = 32;
constexpr int LDK = BK + 4;  // padded K-inner row (floats): 144 B rows -> conflict-free ds_read_b128

enum { MODE_FWD = 0, MODE_DGRAD = 1, MODE_WGRAD = 2, MODE_DGRAD_T = 3 };
// MODE_DGRAD_T: the data gradient with the filter given TRANSPOSED, wt [R][S][C][K] (denet_transpose_f32 of w [K][R*S*C]): the
// B operand is then reduction-contiguous like the forward pass's, goes into LDS K-inner and is read back as one ds_read_b128 per
// fragment instead of four scalar reads - the same products in the same order as MODE_DGRAD (bit-identical results).

struct IgemmParams {
    const float* act;   // fwd: x        dgrad: dy       wgrad: x
    const float* wgt;   // fwd: w(KRSC)  dgrad: w(KRSC)  wgrad: dy
    float* out;         // fwd: y        dgrad: dx       wgrad: dw or split workspace
    const float* bias;  // fwd only, [K] or null
    const float* add;   // fwd/dgrad: tensor of the output's shape added in the epilogue, or null
    int relu;           // fwd: max(., 0) after bias and add (inference with batch norm folded into the filters)
    double* stats;      // fwd, optional: per M tile the column sums [tiles_m][2][NC] (sum, sum of squares) of the stored values:
                        // the batch-norm statistics of the layer behind this convolution (batch_norm.py:50-53), no extra pass
    int N, H, W, C;     // x geometry (C = physical channels)
    int OH, OW, K;      // y geometry (K = physical channels)
    int R, S, S_real;   // filter taps (S may be padded; taps s >= S_real carry zero weight)
    int stride, sshift, pad;
    int M;              // GEMM rows
    int NC;             // GEMM cols
    int ksteps;         // number of 32-wide reduction chunks
    int steps_per_split;
    int tiles_m, tiles_n;
    long split_stride;  // wgrad: elements between split slices of the workspace
    int npix;           // N*OH*OW
    FastDiv div_row_hw; // fwd/wgrad: OH*OW    dgrad: Hc*Wc (pixels of one stride-parity class per image)
    FastDiv div_row_w;  // fwd/wgrad: OW       dgrad: Wc
    int Hc, Wc;         // dgrad: H/stride, W/stride
    int wg_split_slow;  // wgrad: 1 = split slice is the slow (XCD-local) index of the linear workgroup id
    // backward sums (bs_x != null, with `stats`): the tensor written is the gradient of the OUTPUT of a batch norm whose input
    // was bs_x; the column sums become that layer's two backward reductions, sum(g) and sum(g * xhat), g = the value masked by
    // the layer's ReLU (bs_y > 0, or recomputed from bs_x) - what bn_bwd_partial_kernel computes in a pass of its own
    const float* bs_x;
    const float* bs_y;
    const float* bs_gamma;
    const float* bs_beta;
    const float* bs_mean;
    const float* bs_invstd;
    int bs_relu;
    long batch_act, batch_wgt, batch_out;  // fwd / wgrad: element strides of blockIdx.y (batched GEMMs of the Winograd path)
    int batch;          // wgrad: number of batch members (grid.y), 0 = 1
    unsigned act_bytes, wgt_bytes;  // extents of `act` / `wgt` (buffer descriptors: out-of-range lanes read 0)
    BnFinalDev fin;     // with stats: the last row tile of a column tile to arrive reduces the rows itself (bn_final.h); null: off
};

// Guarded operand loads are BRANCH-FREE: a lane whose tap falls into the zero padding (or beyond the tensor) gets
// an offset past the descriptor's extent and the buffer unit returns 0. (`ok ? *ptr : 0` compiles to an exec-mask
// branch per load: every branch ends the scheduling region, so nothing can be interleaved with the MFMAs, and the
// waitcnt pass falls back to vmcnt(0) behind it.)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
constexpr int OOB_OFFSET = (int)0xF0000000u;   // >= any extent check_geom admits; cannot wrap when 15 is added
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, int elem_off, bool ok) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, ok ? elem_off * 4 : OOB_OFFSET, 0, 0);
    return __builtin_bit_cast(f32x4, v);
}

template <int MODE, int BM, int BN, int WM, int WN, int NBUF>
__global__ __launch_bounds__(256, (NBUF == 1 ? (BM * BN <= 128 * 64 ? 4 : 3) : (NBUF == 2 ? 2 : 1))) void igemm_kernel(const IgemmParams p) {
    constexpr bool A_KIN = (MODE != MODE_WGRAD);
    constexpr bool B_KIN = (MODE == MODE_FWD || MODE == MODE_DGRAD_T);
    constexpr bool IS_DGRAD = (MODE == MODE_DGRAD || MODE == MODE_DGRAD_T);
    constexpr int TM = BM / (32 * WM);
    constexpr int TN = BN / (32 * WN);
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(TM >= 1 && TN >= 1, "tile too small");
    constexpr int SZA = A_KIN ? BM * LDK : BK * BM;
    constexpr int SZB = B_KIN ? BN * LDK : BK * BN;
    // loader passes (one float4 per thread per pass)
    constexpr int PA = BM / 32;
    constexpr int PB = BN / 32;
    // K-outer thread mapping
    constexpr int LPR_A = BM / 4, RPP_A = 256 / LPR_A;
    constexpr int LPR_B = BN / 4, RPP_B = 256 / LPR_B;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                 // [NBUF][SZA]
    float* sB = smem + NBUF * SZA;    // [NBUF][SZB]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    // XCD-aware linear workgroup id. fwd/dgrad: consecutive ids walk the N tiles of one M tile, then the next M
    // tile, so an XCD's L2 holds a contiguous range of pixels. wgrad: the split slice is the SLOW index — all
    // (K x RSC) tiles of one pixel range run on the same XCD and share its x / dy rows in L2 (with the slice on
    // grid.y every XCD streamed the whole of x and dy: 8x the HBM traffic).
    const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t ntiles = (uint32_t)(p.tiles_m * p.tiles_n);
    uint32_t tile_id = bid;
    int split_id = 0;
    if (MODE == MODE_WGRAD) {
        if (p.wg_split_slow) {
            tile_id = bid % ntiles;
            split_id = (int)(bid / ntiles);
        } else {
            const uint32_t nsplit = gridDim.x / ntiles;
            split_id = (int)(bid % nsplit);
            tile_id = bid / nsplit;
        }
    }
    const int tile_n = tile_id % p.tiles_n;
    const int tile_m = tile_id / p.tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    int step_begin = 0, step_end = p.ksteps;
    if (MODE == MODE_WGRAD) {
        step_begin = split_id * p.steps_per_split;
        step_end = min(p.ksteps, step_begin + p.steps_per_split);
    }
    // dgrad with stride > 1: blockIdx.y enumerates the stride^2 parity classes of input pixels. A pixel of class
    // (py,px) only receives taps r = r0 + stride*r', s = s0 + stride*s' (the others hit "holes" of the strided
    // output), so each class is a dense problem over its own tap subset: no wasted MFMA work.
    int dg_py = 0, dg_px = 0, dg_r0 = 0, dg_s0 = 0, dg_rc = p.R, dg_sc = p.S;
    if (IS_DGRAD && p.stride > 1) {
        dg_py = blockIdx.y >> p.sshift;
        dg_px = blockIdx.y & (p.stride - 1);
        dg_r0 = (dg_py + p.pad) & (p.stride - 1);
        dg_s0 = (dg_px + p.pad) & (p.stride - 1);
        dg_rc = (p.R > dg_r0) ? ((p.R - dg_r0 + p.stride - 1) >> p.sshift) : 0;
        dg_sc = (p.S > dg_s0) ? ((p.S - dg_s0 + p.stride - 1) >> p.sshift) : 0;
        step_end = dg_rc * dg_sc * (p.K / BK);
    }
    const int nsteps = step_end - step_begin;

    // ---------------- per-thread loader state ----------------
    const int q8 = tid & 7;        // K-inner: float4 column inside the 32-wide chunk
    const int row8 = tid >> 3;     // K-inner: row inside a 32-row pass
    const int qa = tid % LPR_A, kra = tid / LPR_A;   // K-outer A
    const int qb = tid % LPR_B, krb = tid / LPR_B;   // K-outer B

    int a_off[PA];   // element offsets
    int a_y[PA];     // fwd: iy0        dgrad: iy+pad
    int a_x[PA];     // fwd: ix0(+qs)   dgrad: ix+pad
    int b_off[PB];
    bool b_ok[PB];
    int wg_r = 0, wg_s = 0, wg_c = 0;  // wgrad: tap / channel of this thread's B column
    bool wg_colok = false;

    // uniform reduction cursor (fwd / dgrad): tap (r,s) and channel base c0
    int cur_r = 0, cur_s = 0, cur_c = 0;

    if (MODE == MODE_FWD) {
        const int qs = (p.C < 32) ? (4 * q8) / p.C : 0;
        const int qc = (p.C < 32) ? (4 * q8) % p.C : 4 * q8;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int m = m0 + row8 + 32 * i;
            if (m < p.M) {
                const uint32_t n = p.div_row_hw.div(m);
                const uint32_t rem = m - n * (p.OH * p.OW);
                const uint32_t oy = p.div_row_w.div(rem);
                const uint32_t ox = rem - oy * p.OW;
                a_y[i] = (int)oy * p.stride - p.pad;
                a_x[i] = (int)ox * p.stride - p.pad + qs;
                a_off[i] = (((int)n * p.H + a_y[i]) * p.W + a_x[i]) * p.C + qc;
            } else {
                a_y[i] = -(1 << 24);
                a_x[i] = 0;
                a_off[i] = 0;
            }
        }
        const int kred = p.R * p.S * p.C;
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int n = n0 + row8 + 32 * i;
            b_ok[i] = n < p.K;
            b_off[i] = n * kred + 4 * q8;
        }
    } else if (IS_DGRAD) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int m = m0 + row8 + 32 * i;
            if (m < p.M) {
                const uint32_t n = p.div_row_hw.div(m);
                const uint32_t rem = m - n * (p.Hc * p.Wc);
                const uint32_t ya = p.div_row_w.div(rem);
                const uint32_t xa = rem - ya * p.Wc;
                a_y[i] = (int)ya * p.stride + dg_py + p.pad - dg_r0;
                a_x[i] = (int)xa * p.stride + dg_px + p.pad - dg_s0;
                a_off[i] = (int)n * (p.OH * p.OW * p.K) + 4 * q8;
            } else {
                a_y[i] = -(1 << 24);
                a_x[i] = 0;
                a_off[i] = 0;
            }
        }
        const int rsc = p.R * p.S * p.C;
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            if (MODE == MODE_DGRAD_T) {          // wt [R][S][C][K]: row = output channel c of the tap, K-inner
                const int n = n0 + row8 + 32 * i;
                b_ok[i] = n < p.C;
                b_off[i] = n * p.K + 4 * q8;
            } else {
                const int col = n0 + 4 * qb;
                b_ok[i] = col < p.C;
                b_off[i] = (krb + RPP_B * i) * rsc + col;
            }
        }
    } else {  // WGRAD
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            a_off[i] = (kra + RPP_A * i) * p.K + m0 + 4 * qa;
            a_y[i] = 0;
            a_x[i] = 0;
        }
        const int jg = n0 + 4 * qb;
        wg_colok = jg < p.NC;
        const int rs = jg / p.C;
        wg_c = jg - rs * p.C;
        wg_r = rs / p.S;
        wg_s = rs - wg_r * p.S;
        wg_colok = wg_colok && (wg_s < p.S_real);
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            b_ok[i] = wg_colok;
            b_off[i] = 0;
        }
    }
    const bool wg_rowok = (MODE == MODE_WGRAD) ? (m0 + 4 * qa < p.K) : true;

    f32x4 ra[PA], rb[PB];
    // fwd: blockIdx.y selects one GEMM of a batch (same shapes, strided operands); the descriptors cover one member
    const long by = (MODE == MODE_FWD || MODE == MODE_WGRAD) ? (long)blockIdx.y : 0;
    const __amdgpu_buffer_rsrc_t r_act =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.act + by * p.batch_act), 0, p.act_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_wgt =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.wgt + by * p.batch_wgt), 0, p.wgt_bytes, 0x00020000);

    // ---- chunk loader, split so that the pipelined main loop can slot single passes behind MFMAs ----------
    // prep(kc): wave-uniform offsets of reduction chunk `kc` (fwd/dgrad walk the cursor, wgrad is absolute);
    // load_a(i) / load_b(i): pass i (one float4 per thread) into ra[i] / rb[i]; advance(): cursor to the next chunk
    int u_off = 0, u_offb = 0, u_ty = 0, u_tx = 0, u_pix0 = 0;
    auto prep = [&](int kc) {
        if (MODE == MODE_FWD) {
            u_off = (cur_r * p.W + cur_s) * p.C + cur_c;
            u_offb = kc * BK;
        } else if (IS_DGRAD) {
            u_ty = cur_r << p.sshift;
            u_tx = cur_s << p.sshift;
            u_off = cur_c;
            if (MODE == MODE_DGRAD_T) u_offb = ((dg_r0 + u_ty) * p.S + dg_s0 + u_tx) * p.C * p.K + cur_c;
            else u_offb = cur_c * (p.R * p.S * p.C) + ((dg_r0 + u_ty) * p.S + dg_s0 + u_tx) * p.C;
        } else {
            u_pix0 = kc * BK;
        }
    };
    auto load_a = [&](int i) {
        if (MODE == MODE_FWD) {
            const bool ok = ((unsigned)(a_y[i] + cur_r) < (unsigned)p.H) && ((unsigned)(a_x[i] + cur_s) < (unsigned)p.W);
            ra[i] = buf_load4(r_act, a_off[i] + u_off, ok);
        } else if (IS_DGRAD) {
            // (iy + pad - r) is a multiple of the stride by construction of the class
            const int ty = a_y[i] - u_ty;
            const int tx = a_x[i] - u_tx;
            const int oy = ty >> p.sshift;
            const int ox = tx >> p.sshift;
            const bool ok = (ty >= 0) && (tx >= 0) && (oy < p.OH) && (ox < p.OW);
            ra[i] = buf_load4(r_act, a_off[i] + (oy * p.OW + ox) * p.K + u_off, ok);
        } else {
            const int pix = u_pix0 + kra + RPP_A * i;
            const bool ok = wg_rowok && (pix < p.npix);
            ra[i] = buf_load4(r_wgt, u_pix0 * p.K + a_off[i], ok);
        }
    };
    auto load_b = [&](int i) {
        if (MODE == MODE_FWD || IS_DGRAD) {
            rb[i] = buf_load4(r_wgt, b_off[i] + u_offb, b_ok[i]);
        } else {
            const int pix = u_pix0 + krb + RPP_B * i;
            const uint32_t n = p.div_row_hw.div(pix);
            const uint32_t rem = pix - n * (p.OH * p.OW);
            const uint32_t oy = p.div_row_w.div(rem);
            const uint32_t ox = rem - oy * p.OW;
            const int iy = (int)oy * p.stride - p.pad + wg_r;
            const int ix = (int)ox * p.stride - p.pad + wg_s;
            const bool ok = wg_colok && (pix < p.npix) && ((unsigned)iy < (unsigned)p.H) && ((unsigned)ix < (unsigned)p.W);
            rb[i] = buf_load4(r_act, (((int)n * p.H + iy) * p.W + ix) * p.C + wg_c, ok);
        }
    };
    auto advance = [&]() {
        if (MODE == MODE_FWD) {
            cur_c += BK;
            if (cur_c >= p.C) {
                cur_c = 0;
                cur_s += (p.C < BK) ? (BK / p.C) : 1;
                if (cur_s >= p.S) {
                    cur_s = 0;
                    cur_r += 1;
                }
            }
        } else if (IS_DGRAD) {
            cur_c += BK;
            if (cur_c >= p.K) {
                cur_c = 0;
                cur_s += 1;
                if (cur_s >= dg_sc) {
                    cur_s = 0;
                    cur_r += 1;
                }
            }
        }
    };
    // loads the reduction chunk `kc` (absolute chunk index) into ra/rb
    auto load_chunk = [&](int kc) {
        prep(kc);
#pragma unroll
        for (int i = 0; i < PA; ++i) load_a(i);
#pragma unroll
        for (int i = 0; i < PB; ++i) load_b(i);
        advance();
    };

    auto write_a = [&](float* dA, int i) {
        if (A_KIN)
            *(f32x4*)(dA + (row8 + 32 * i) * LDK + 4 * q8) = ra[i];
        else
            *(f32x4*)(dA + (kra + RPP_A * i) * BM + 4 * qa) = ra[i];
    };
    auto write_b = [&](float* dB, int i) {
        if (B_KIN)
            *(f32x4*)(dB + (row8 + 32 * i) * LDK + 4 * q8) = rb[i];
        else
            *(f32x4*)(dB + (krb + RPP_B * i) * BN + 4 * qb) = rb[i];
    };

    auto store_chunk = [&](int buf) {
        float* dA = sA + buf * SZA;
        float* dB = sB + buf * SZB;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            if (A_KIN)
                *(f32x4*)(dA + (row8 + 32 * i) * LDK + 4 * q8) = ra[i];
            else
                *(f32x4*)(dA + (kra + RPP_A * i) * BM + 4 * qa) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            if (B_KIN)
                *(f32x4*)(dB + (row8 + 32 * i) * LDK + 4 * q8) = rb[i];
            else
                *(f32x4*)(dB + (krb + RPP_B * i) * BN + 4 * qb) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment base offsets inside the LDS tiles
    const int fa = A_KIN ? (wm * TM * 32 + li) * LDK + 4 * lh : (4 * lh) * BM + wm * TM * 32 + li;
    const int fb = B_KIN ? (wn * TN * 32 + li) * LDK + 4 * lh : (4 * lh) * BN + wn * TN * 32 + li;

    // MFMA phase of one K chunk. Operand fragments are double buffered in registers: the LDS reads of 8-wide k
    // block kb+1 are issued BEFORE the 16 MFMAs of block kb, so the ~100+ cycle LDS latency is always covered by
    // a full block of matrix work (the compiler otherwise places the reads 1-4 MFMAs ahead of their use).
    auto load_frags = [&](const float* cA, const float* cB, int kb, float (&av)[TM][4], float (&bv)[TN][4]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (A_KIN) {
                const f32x4 t = *(const f32x4*)(cA + i * 32 * LDK + kb * 8);
                av[i][0] = t[0]; av[i][1] = t[1]; av[i][2] = t[2]; av[i][3] = t[3];
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) av[i][t] = cA[(kb * 8 + t) * BM + i * 32];
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (B_KIN) {
                const f32x4 t = *(const f32x4*)(cB + j * 32 * LDK + kb * 8);
                bv[j][0] = t[0]; bv[j][1] = t[1]; bv[j][2] = t[2]; bv[j][3] = t[3];
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) bv[j][t] = cB[(kb * 8 + t) * BN + j * 32];
            }
        }
    };

    // a wave whose 32*TN columns lie entirely in the padding of the last N tile (e.g. wgrad of a 64-channel 3x3:
    // 576 = 4.5 x 128 columns) skips its matrix work; it still helps staging and keeps the barriers
    const bool wave_active = (n0 + wn * TN * 32 < p.NC) && (m0 + wm * TM * 32 < p.M);
    auto compute = [&](int buf) {
        if (!wave_active) return;
        const float* cA = sA + buf * SZA + fa;
        const float* cB = sB + buf * SZB + fb;
        float av[2][TM][4], bv[2][TN][4];
        load_frags(cA, cB, 0, av[0], bv[0]);
#pragma unroll
        for (int kb = 0; kb < BK / 8; ++kb) {
            if (kb + 1 < BK / 8) load_frags(cA, cB, kb + 1, av[(kb + 1) & 1], bv[(kb + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch above the matrix block
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[kb & 1][j][t], av[kb & 1][i][t], acc[i][j],
                                                                         0, 0, 0);
        }
    };

    // single-buffer loop flavour: the interleaved form pays on the narrow tiles (8 MFMAs per block: +4...8 %), costs
    // 2-4 % on 128x128 dgrad and spills on wgrad (its loader state is large) - measured with tools/bench_conv.py
    constexpr bool IL1 = (MODE != MODE_WGRAD) && (BN == 64);
    if (NBUF == 1 && !IL1) {
        if (nsteps > 0) {
            load_chunk(step_begin);
            store_chunk(0);
            __syncthreads();
            for (int s = 0; s < nsteps; ++s) {
                const bool more = (s + 1 < nsteps);
                if (more) load_chunk(step_begin + s + 1);
                compute(0);
                __syncthreads();
                if (more) store_chunk(0);
                __syncthreads();
            }
        }
    } else if (NBUF == 1) {
        if (nsteps > 0) {
            // one LDS buffer: a third of the LDS -> 3-4 workgroups per CU cover each other's barrier bubbles. The
            // global loads of chunk s+1 are slotted pass by pass behind the MFMAs of chunk s (no burst in front of
            // the matrix block); the LDS writes have to wait for the barrier that ends the reads of chunk s.
            constexpr int KB = BK / 8;
            static_assert(PA <= KB && PB <= KB, "one staging pass per MFMA block");
            float av[2][TM][4], bv[2][TN][4];
            load_chunk(step_begin);
            store_chunk(0);
            __syncthreads();
            load_frags(sA + fa, sB + fb, 0, av[0], bv[0]);
            auto body1 = [&](int s, auto LF) {
                constexpr bool do_l = decltype(LF)::value;
                if (do_l) prep(step_begin + s + 1);
                auto step = [&](auto KBI) {
                    constexpr int kb = decltype(KBI)::value;
                    constexpr bool rd = (kb + 1 < KB);
                    if (rd) load_frags(sA + fa, sB + fb, kb + 1, av[(kb + 1) & 1], bv[(kb + 1) & 1]);
                    if constexpr (kb < PA) {
                        if (do_l) load_a(kb);
                    }
                    if constexpr (kb < PB) {
                        if (do_l) load_b(kb);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[kb & 1][j][t], av[kb & 1][i][t],
                                                                                 acc[i][j], 0, 0, 0);
                    constexpr int NM = 4 * TM * TN;
                    constexpr int NR = rd ? (A_KIN ? TM : 4 * TM) + (B_KIN ? TN : 4 * TN) : 0;
                    constexpr int NL = do_l ? ((kb < PA) ? 1 : 0) + ((kb < PB) ? 1 : 0) : 0;
                    constexpr int HALF = (NM / 2 > 0) ? NM / 2 : 1;
                    constexpr int RP = NR ? (NR + HALF - 1) / HALF : 1;
                    constexpr int RS = NR ? (NR + RP - 1) / RP : 0;
                    constexpr int LS = (NL < NM - RS) ? NL : ((NM - RS > 0) ? NM - RS : 0);
#pragma unroll
                    for (int g = 0; g < RS; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, RP, 0);
                    }
#pragma unroll
                    for (int g = 0; g < LS; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, MODE == MODE_WGRAD ? 12 : 6, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                    if constexpr (NM - RS - LS > 0) __builtin_amdgcn_sched_group_barrier(0x008, NM - RS - LS, 0);
                    __builtin_amdgcn_sched_barrier(0);
                };
                step(std::integral_constant<int, 0>{});
                step(std::integral_constant<int, 1>{});
                step(std::integral_constant<int, 2>{});
                step(std::integral_constant<int, 3>{});
                if (do_l) {
                    advance();
                    __syncthreads();
                    store_chunk(0);
                    __syncthreads();
                    load_frags(sA + fa, sB + fb, 0, av[0], bv[0]);
                }
            };
            int s = 0;
            for (; s + 1 < nsteps; ++s) body1(s, std::true_type{});
            body1(s, std::false_type{});
        }
    } else if (nsteps > 0) {
        // ---- pipelined main loop (NBUF = 2 or 3 LDS buffers, ONE register staging set) --------------------
        // Iteration s multiplies chunk s out of LDS buffer s % NBUF. Its MFMAs come in BK/8 blocks; behind the
        // first MFMAs of every block the scheduler is told (sched_group_barrier) to slot, one per MFMA:
        //   - the fragment reads of the NEXT block (register double buffer),
        //   - pass `kb` of the staging: ds_write of chunk s+AH (its loads were issued one iteration ago), then
        //     the global loads of chunk s+AH+1 into the SAME registers (write-then-reload, AH = NBUF-1).
        // A v_mfma_f32_32x32x2 occupies the matrix pipe for 64 cycles; one LDS / VMEM / few VALU instructions
        // issue in its shadow for free, whereas the same instructions in a burst between MFMA blocks drain the
        // pipe (measured: 75 % -> 90 % MFMA utilisation on GEMM-shaped layers). With 3 buffers the first
        // fragments of chunk s+1 are read before the barrier as well (nothing is exposed after it); with 2
        // buffers they are read right after the barrier and two workgroups per CU cover that bubble.
        constexpr int AH = NBUF - 1;
        constexpr int KB = BK / 8;
        static_assert(PA <= KB && PB <= KB, "one staging pass per MFMA block");
        float av[2][TM][4], bv[2][TN][4];
        // prologue: chunks 0..AH-1 -> LDS, chunk AH -> registers
        load_chunk(step_begin);
        store_chunk(0);
        if (NBUF == 3 && nsteps > 1) {
            load_chunk(step_begin + 1);
            store_chunk(1);
        }
        if (nsteps > AH) load_chunk(step_begin + AH);
        __syncthreads();
        load_frags(sA + fa, sB + fb, 0, av[0], bv[0]);
        int cur = 0;
        auto body = [&](int s, auto WF, auto LF, auto NF) {
            constexpr bool do_w = decltype(WF)::value, do_l = decltype(LF)::value, has_next = decltype(NF)::value;
            const int nxt = (cur == NBUF - 1) ? 0 : cur + 1;                                // buffer of chunk s+1
            const int wbuf = (NBUF == 2) ? nxt : ((nxt == NBUF - 1) ? 0 : nxt + 1);         // buffer of chunk s+AH
            const float* cA = sA + cur * SZA + fa;
            const float* cB = sB + cur * SZB + fb;
            float* wA = sA + wbuf * SZA;
            float* wB = sB + wbuf * SZB;
            if (do_l) prep(step_begin + s + AH + 1);
            auto step = [&](auto KBI) {
                constexpr int kb = decltype(KBI)::value;
                constexpr bool rd = (kb + 1 < KB) || (NBUF == 3 && has_next);
                if (kb + 1 < KB) load_frags(cA, cB, kb + 1, av[(kb + 1) & 1], bv[(kb + 1) & 1]);
                else if (NBUF == 3 && has_next) load_frags(sA + nxt * SZA + fa, sB + nxt * SZB + fb, 0, av[0], bv[0]);
                if constexpr (kb < PA) {
                    if (do_w) write_a(wA, kb);
                    if (do_l) load_a(kb);
                }
                if constexpr (kb < PB) {
                    if (do_w) write_b(wB, kb);
                    if (do_l) load_b(kb);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[kb & 1][j][t], av[kb & 1][i][t], acc[i][j],
                                                                             0, 0, 0);
                // issue order of this block
                constexpr int NM = 4 * TM * TN;
                constexpr int NR = rd ? (A_KIN ? TM : 4 * TM) + (B_KIN ? TN : 4 * TN) : 0;
                constexpr int NW = do_w ? ((kb < PA) ? 1 : 0) + ((kb < PB) ? 1 : 0) : 0;
                constexpr int HALF = (NM / 2 > 0) ? NM / 2 : 1;
                constexpr int RP = NR ? (NR + HALF - 1) / HALF : 1;
                constexpr int RS = NR ? (NR + RP - 1) / RP : 0;
                constexpr int WS = (NW < NM - RS) ? NW : ((NM - RS > 0) ? NM - RS : 0);
#pragma unroll
                for (int g = 0; g < RS; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, RP, 0);
                }
#pragma unroll
                for (int g = 0; g < WS; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    if (do_l) {
                        __builtin_amdgcn_sched_group_barrier(0x002, MODE == MODE_WGRAD ? 12 : 6, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
                if constexpr (NM - RS - WS > 0) __builtin_amdgcn_sched_group_barrier(0x008, NM - RS - WS, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{});
            if (do_l) advance();
            __syncthreads();
            if (NBUF == 2 && has_next) load_frags(sA + nxt * SZA + fa, sB + nxt * SZB + fb, 0, av[0], bv[0]);
            cur = nxt;
        };
        {
            using T = std::true_type;
            using F = std::false_type;
            int s = 0;
            for (; s + AH + 1 < nsteps; ++s) body(s, T{}, T{}, T{});
            for (; s + AH < nsteps; ++s) body(s, T{}, F{}, T{});
            for (; s + 1 < nsteps; ++s) body(s, F{}, F{}, T{});
            for (; s < nsteps; ++s) body(s, F{}, F{}, F{});
        }
    }

    // ---------------- epilogue ----------------
    // The MFMAs were issued with the B fragment as the row operand and the A fragment as the column operand, so the
    // accumulators hold the TRANSPOSED 32x32 blocks: lane (li, lh) owns output row m = .. + li and, per register group g,
    // the four consecutive columns n = .. + 8g + 4lh + {0..3}: one 16-byte store per group instead of four scalar ones
    // (the scalar form is store-issue bound: a quarter of the run time of the short-reduction component GEMMs).
    float* out = p.out;
    if (MODE == MODE_WGRAD) out += (long)split_id * p.split_stride;
    if (MODE == MODE_FWD || MODE == MODE_WGRAD) out += by * p.batch_out;
    if ((MODE == MODE_FWD || IS_DGRAD) && p.stats) {
        // ---- store + batch-norm column sums. Lane (li, lh) holds row m and, per (j, g), 4 consecutive columns: the sums
        // over the rows of the tile are a reduction over li (shuffles inside each 32-lane half), then over the two waves
        // stacked along M (LDS), written as doubles: partial[tile_m][0][n] = sum, [1][n] = sum of squares.
        __syncthreads();                       // the operand buffers are reused below: every wave is out of the main loop
        double* redd = (double*)smem;          // [WM][2][BN] (the operand buffers hold at least 2 x 128 x 36 floats)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = (wn * TN + j) * 32 + 8 * g + 4 * lh;      // column inside the tile
                const int n = n0 + nl;
                f32x4 sv = {0.f, 0.f, 0.f, 0.f}, sq = {0.f, 0.f, 0.f, 0.f};
                if (n < p.NC) {
                    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
                    if (p.bias) bias4 = *(const f32x4*)(p.bias + n);
                    f32x4 bmu = {0.f, 0.f, 0.f, 0.f}, bis = bmu, bsc = bmu, bsh = bmu;
                    if (p.bs_x) {
                        bmu = *(const f32x4*)(p.bs_mean + n);
                        bis = *(const f32x4*)(p.bs_invstd + n);
                        if (p.bs_relu && !p.bs_y) {
                            bsc = *(const f32x4*)(p.bs_gamma + n) * bis;
                            bsh = *(const f32x4*)(p.bs_beta + n) - bmu * bsc;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int m = m0 + (wm * TM + i) * 32 + li;
                        if (m < p.M) {
                            long row = (long)m * p.NC;
                            if (IS_DGRAD && p.stride > 1) {      // pixel m of this parity class (see the plain epilogue)
                                const uint32_t ni = p.div_row_hw.div(m);
                                const uint32_t rem = m - ni * (p.Hc * p.Wc);
                                const uint32_t ya = p.div_row_w.div(rem);
                                const uint32_t xa = rem - ya * p.Wc;
                                row = (((long)ni * p.H + (ya * p.stride + dg_py)) * p.W + (xa * p.stride + dg_px)) * p.NC;
                            }
                            f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                            v += bias4;
                            if (p.add) v += *(const f32x4*)(p.add + row + n);
                            if (p.relu) {
                                v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                            }
                            *(f32x4*)(p.out + row + n) = v;
                            if (p.bs_x) {
                                const f32x4 xv = *(const f32x4*)(p.bs_x + row + n);
                                f32x4 gq = v;
                                if (p.bs_relu) {
                                    if (p.bs_y) {
                                        const f32x4 yv = *(const f32x4*)(p.bs_y + row + n);
#pragma unroll
                                        for (int c = 0; c < 4; ++c) gq[c] = yv[c] > 0.f ? gq[c] : 0.f;
                                    } else {
#pragma unroll
                                        for (int c = 0; c < 4; ++c) gq[c] = fmaf(xv[c], bsc[c], bsh[c]) > 0.f ? gq[c] : 0.f;
                                    }
                                }
                                sv += gq;
                                sq += gq * ((xv - bmu) * bis);
                            } else {
                                sv += v;
                                sq += v * v;
                            }
                        }
                    }
                }
                // fp32 while a partial holds at most 2 TM values (one shuffle step), doubles from there on: a sum of squares of
                // 128-256 values kept in fp32 costs var = E[x^2] - mean^2 three digits at |mean| / std = 30
#pragma unroll
                for (int off = 16; off > 8; off >>= 1) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        sv[c] += __shfl_xor(sv[c], off, 64);
                        sq[c] += __shfl_xor(sq[c], off, 64);
                    }
                }
                double dv[4], dq[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    dv[c] = (double)sv[c];
                    dq[c] = (double)sq[c];
                }
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        dv[c] += __shfl_xor(dv[c], off, 64);
                        dq[c] += __shfl_xor(dq[c], off, 64);
                    }
                }
                if (li == 0) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        redd[(wm * 2 + 0) * BN + nl + c] = dv[c];
                        redd[(wm * 2 + 1) * BN + nl + c] = dq[c];
                    }
                }
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < p.NC) {
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int w = 0; w < WM; ++w) {
                a += redd[(w * 2 + 0) * BN + tid];
                b += redd[(w * 2 + 1) * BN + tid];
            }
            // one row per (parity class, row tile): every pixel of dx is in exactly one
            double* ps = p.stats + ((IS_DGRAD ? (long)blockIdx.y * p.tiles_m : 0L) + tile_m) * 2 * p.NC;
            bnf_store(ps + n0 + tid, a);
            bnf_store(ps + p.NC + n0 + tid, b);
        }
        // the last row tile (and parity class) of this column tile to arrive finishes the batch norm's reduction (bn_final.h)
        const int rows_all = p.tiles_m * (IS_DGRAD ? (int)gridDim.y : 1);
        bnf_tail<256>(p.fin, p.stats, rows_all, n0, (p.NC - n0 < BN) ? p.NC - n0 : BN, tile_n, (unsigned)rows_all,
                      (int*)(redd + WM * 2 * BN));
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + (wm * TM + i) * 32 + li;
        if (m >= p.M) continue;
        long row = (long)m * p.NC;
        if (IS_DGRAD && p.stride > 1) {
            const uint32_t ni = p.div_row_hw.div(m);
            const uint32_t rem = m - ni * (p.Hc * p.Wc);
            const uint32_t ya = p.div_row_w.div(rem);
            const uint32_t xa = rem - ya * p.Wc;
            row = (((long)ni * p.H + (ya * p.stride + dg_py)) * p.W + (xa * p.stride + dg_px)) * p.NC;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + (wn * TN + j) * 32 + 8 * g + 4 * lh;
                if (n >= p.NC) continue;                 // NC is a multiple of 4: the group is all inside or all outside
                f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                if (MODE == MODE_FWD && p.bias) v += *(const f32x4*)(p.bias + n);
                if (MODE != MODE_WGRAD && p.add) v += *(const f32x4*)(p.add + row + n);
                if (MODE == MODE_FWD && p.relu) {
                    v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                }
                *(f32x4*)(out + row + n) = v;
            }
        }
    }
}

// sums the split-K slices of a weight-gradient workspace: a workgroup owns 64 float4 columns, its 4 waves take
// every 4th slice (4 loads in flight each) and combine through LDS in a fixed order (deterministic)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                            long n4, int splits, long stride4) {
    __shared__ f32x4 red[4][64];
    const f32x4* w4 = (const f32x4*)ws;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long col = (long)blockIdx.x * 64 + lane;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (col < n4) {
        int z = wv;
        for (; z + 12 < splits; z += 16) {
            const f32x4 a = w4[col + (long)z * stride4], b = w4[col + (long)(z + 4) * stride4];
            const f32x4 c = w4[col + (long)(z + 8) * stride4], d = w4[col + (long)(z + 12) * stride4];
            s += (a + b) + (c + d);
        }
        for (; z < splits; z += 4) s += w4[col + (long)z * stride4];
    }
    red[wv][lane] = s;
    __syncthreads();
    if (wv == 0 && col < n4) ((f32x4*)out)[col] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// the instantiation chosen by the last conv launch of this thread (profiling / roofline bookkeeping)
thread_local int g_last_cfg[5] = {0, 0, 0, 0, 0};   // mode, BM, BN, NBUF, grid.y

// optional live timing of the igemm kernel ALONE (bench.py's roofline leg): one event pair per launch, recorded on
// the stream the kernel runs on, so that the average agrees with rocprofv3's per-kernel duration
struct ProfRec {
    hipEvent_t a, b;
    int cfg[4];
};
std::vector<ProfRec> g_prof;
bool g_prof_on = false;
bool g_prof_events = true;       // false: launch trace only (denet_conv_profile(2)): which instantiation ran, no events, no timing

// opens a record (event on `stream` before the launch); -1 when profiling is off or an event cannot be created
int prof_begin(int mode, int bm, int bn, int nbuf, hipStream_t stream) {
    if (!g_prof_on) return -1;
    ProfRec rec;
    rec.cfg[0] = mode; rec.cfg[1] = bm; rec.cfg[2] = bn; rec.cfg[3] = nbuf;
    rec.a = rec.b = nullptr;
    if (!g_prof_events) {
        g_prof.push_back(rec);
        return -1;                   // nothing to close
    }
    if (hipEventCreate(&rec.a) != hipSuccess) return -1;
    if (hipEventCreate(&rec.b) != hipSuccess) {
        (void)hipEventDestroy(rec.a);
        return -1;
    }
    (void)hipEventRecord(rec.a, stream);
    g_prof.push_back(rec);
    return (int)g_prof.size() - 1;
}
void prof_end(int idx, hipStream_t stream) {
    if (idx >= 0) (void)hipEventRecord(g_prof[idx].b, stream);
}

template <int MODE, int BM, int BN, int WM, int WN, int NBUF = 2>
int launch_igemm(const IgemmParams& p, int splits, hipStream_t stream) {
    g_last_cfg[0] = MODE; g_last_cfg[1] = BM; g_last_cfg[2] = BN; g_last_cfg[3] = NBUF; g_last_cfg[4] = splits;
    constexpr bool A_KIN = (MODE != MODE_WGRAD);
    constexpr bool B_KIN = (MODE == MODE_FWD || MODE == MODE_DGRAD_T);
    constexpr int SZA = A_KIN ? BM * LDK : BK * BM;
    constexpr int SZB = B_KIN ? BN * LDK : BK * BN;
    constexpr size_t lds = NBUF * (size_t)(SZA + SZB) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)igemm_kernel<MODE, BM, BN, WM, WN, NBUF>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            denet_set_error("igemm: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
            return -(int)e;
        }
        attr_set = true;
    }
    // wgrad folds the split slices into grid.x (see the id decode in the kernel); dgrad: y = stride parity classes
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n) * (MODE == MODE_WGRAD ? (unsigned)splits : 1u),
              MODE == MODE_WGRAD ? (unsigned)(p.batch > 1 ? p.batch : 1) : (unsigned)splits, 1);
    const int prof = prof_begin(MODE, BM, BN, NBUF, stream);
    hipLaunchKernelGGL((igemm_kernel<MODE, BM, BN, WM, WN, NBUF>), grid, dim3(256), lds, stream, p);
    prof_end(prof, stream);
    DENET_CHECK_LAUNCH("igemm");
    return DENET_OK;
}

#define LAUNCH_NBUF(MODE_, BM_, BN_, nbuf_, p_, splits_, stream_)                                        \
    ((nbuf_) == 1   ? launch_igemm<MODE_, BM_, BN_, 2, 2, 1>(p_, splits_, stream_)                       \
     : (nbuf_) == 2 ? launch_igemm<MODE_, BM_, BN_, 2, 2, 2>(p_, splits_, stream_)                       \
                    : launch_igemm<MODE_, BM_, BN_, 2, 2, 3>(p_, splits_, stream_))

// Launch-shape selection. The chip runs 256 CUs x `cap` resident workgroups (cap = 2 with the double LDS buffer,
// 3 with the single buffer); a grid that needs a partial extra round wastes up to a whole round. The launchers
// price each candidate (tile, buffering) by rounds x tile area / relative tile efficiency and take the cheapest.
// DENET_IGEMM_NBUF=1|2 forces the buffering (experiments).
// the batch norm whose sums this launch writes, if the caller armed it (bn_final.h): the last row tile of a column tile then reads
// rows x bn columns x 16 bytes - taken over only while that stays below 1 MB (the head: 144 rows; a 64x64 map at batch 32 leaves
// 1024 rows, which the separate kernel's C / 2 workgroups reduce faster than one workgroup could)
void take_final(IgemmParams& p, int bn, int classes) {
    if (!p.stats) return;
    const long rows = (long)p.tiles_m * classes;
    if (rows * bn * 16 > (1L << 20)) return;
    p.fin = denet_bn_final_take(p.bs_x ? 2 : 1, p.NC, p.tiles_n);
}

int forced_nbuf() {
    static int forced = -1;
    if (forced < 0) {
        const char* e = getenv("DENET_IGEMM_NBUF");
        forced = e ? atoi(e) : 0;
    }
    return forced;
}

struct Choice {
    double cost;
    int tile;   // index of the tile candidate
    int nbuf;
};

// nblocks[t]: grid size with tile candidate t; area[t]: BM*BN; eff[t]: relative MFMA efficiency of the tile.
// Measured behaviour behind the rules: with >= 4 rounds of work the single-buffer variant streams ~12 % faster
// (3 resident workgroups cover each other's barrier bubbles); for small grids what counts is the number of
// rounds, and a half-size tile pays off only when the full tile cannot fill the chip once.
// `want_nbuf`: the loop structure measured fastest for the mode / reduction length (see the callers); the tile shape
// is then chosen for that structure's occupancy.
// ---- measured launch configurations (denet_conv_tune) ---------------------------------------------------
// key: mode + geometry; value: tile index, loop structure (NBUF) and, for wgrad, the round count that fixes the split
struct TuneKey {
    int v[11];
    bool operator<(const TuneKey& o) const {
        for (int i = 0; i < 11; ++i)
            if (v[i] != o.v[i]) return v[i] < o.v[i];
        return false;
    }
};
struct TuneVal {
    int tile, nbuf, rounds;
};
std::map<TuneKey, TuneVal> g_tuned;
thread_local TuneVal t_try = {-1, 0, 0};     // candidate being timed by denet_conv_tune (tile < 0: none)

TuneKey tune_key(int mode, int N, int H, int W, int C, int K, int R, int S, int S_real, int stride, int pad) {
    return TuneKey{{mode, N, H, W, C, K, R, S, S_real, stride, pad}};
}

// the configuration to use: the candidate under test, else a measured entry, else nothing (heuristics apply)
bool tuned_choice(const TuneKey& k, TuneVal* out) {
    if (t_try.tile >= 0) {
        *out = t_try;
        return true;
    }
    auto it = g_tuned.find(k);
    if (it == g_tuned.end()) return false;
    *out = it->second;
    return true;
}

int forced_tile() {      // DENET_IGEMM_TILE: 1 = first candidate (128x128), 2 = second (128x64); tuning aid
    static int forced = -1;
    if (forced < 0) {
        const char* e = getenv("DENET_IGEMM_TILE");
        forced = e ? atoi(e) : 0;
    }
    return forced;
}

Choice choose_launch(const long* nblocks, const double* area, const double* eff, int ntiles, int want_nbuf) {
    Choice best = {1e300, 0, 2};
    for (int t = 0; t < ntiles; ++t) {
        if (forced_tile() && ntiles > 1 && forced_tile() - 1 != t) continue;
        for (int nbuf = 1; nbuf <= 3; ++nbuf) {
            if ((forced_nbuf() ? forced_nbuf() : want_nbuf) != nbuf) continue;
            const long cap = (nbuf == 1) ? (area[t] <= 128.0 * 64.0 ? 4 : 3) : (nbuf == 2 ? 2 : 1);
            const long slots = 256 * cap;
            const long nb = nblocks[t];
            double rounds;
            if (nb >= 4 * slots) rounds = (double)nb / slots;          // streaming regime: no quantisation
            else rounds = (double)((nb + slots - 1) / slots);
            const double per_round = area[t] / eff[t] * (nbuf == 1 ? (cap / 2.0) / 1.12 : 1.0);
            const double cost = rounds * per_round;
            if (cost < best.cost) best = {cost, t, nbuf};
        }
    }
    return best;
}

int ilog2_exact(int v) {
    int s = 0;
    while ((1 << s) < v) s++;
    return ((1 << s) == v) ? s : -1;
}

int check_geom(int N, int H, int W, int C, int K, int R, int S, int S_real, int stride, int pad, int OH, int OW) {
    DENET_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && K > 0 && R > 0 && S > 0, "conv: non-positive dimension");
    DENET_CHECK_ARG(S_real > 0 && S_real <= S, "conv: S_real out of range");
    DENET_CHECK_ARG(ilog2_exact(stride) >= 0, "conv: stride must be a power of two (got %d)", stride);
    DENET_CHECK_ARG(pad >= 0, "conv: negative pad");
    DENET_CHECK_ARG(K % 32 == 0, "conv: physical K (%d) must be a multiple of 32", K);
    if (C >= 32) {
        DENET_CHECK_ARG(C % 32 == 0, "conv: physical C (%d) must be a multiple of 32", C);
    } else {
        DENET_CHECK_ARG(C == 4 || C == 8 || C == 16, "conv: small C must be 4, 8 or 16 (got %d)", C);
        DENET_CHECK_ARG((S * C) % 32 == 0, "conv: S*C (%d) must be a multiple of 32 for small C", S * C);
    }
    DENET_CHECK_ARG((H + 2 * pad - R) / stride + 1 >= OH && OH > 0, "conv: OH=%d inconsistent", OH);
    DENET_CHECK_ARG((W + 2 * pad - S_real) / stride + 1 >= OW && OW > 0, "conv: OW=%d inconsistent", OW);
    // operands are addressed through 32-bit buffer descriptors (byte offsets); 0xF0000000 is the out-of-range marker
    DENET_CHECK_ARG((long)N * H * W * C * 4 < 0xF0000000L && (long)N * OH * OW * K * 4 < 0xF0000000L &&
                        (long)K * R * S * C * 4 < 0xF0000000L,
                    "conv: tensor exceeds the 32-bit buffer extent (3.75 GiB)");
    return DENET_OK;
}

// split-K selection, launch and second-stage reduction of a weight-gradient problem described by `p` (p.M = K rows,
// p.NC columns, p.npix reduction length, p.batch members sharing one launch)
int wgrad_dispatch(IgemmParams& p, int K, float* dw, float* workspace, size_t workspace_bytes, const TuneKey& key,
                   hipStream_t stream) {
    p.ksteps = ceil_div(p.npix, BK);
    const int batch = p.batch > 1 ? p.batch : 1;
    int rc = DENET_OK;
    const bool big_m = (K >= 128);
    const int bm = big_m ? 128 : 64;
    // 64 x 64 tiles when the product has at most 64 columns (the batched filter-gradient products of the 64-channel Winograd
    // layers: dU[64][64]): with 128-wide tiles half of the waves would own nothing but padding
    const bool narrow = !big_m && p.NC <= 64;
    p.tiles_m = ceil_div(K, bm); p.tiles_n = ceil_div(p.NC, narrow ? 64 : 128);
    const long wsize = (long)K * p.NC * batch;     // all batch members of one split slice
    p.split_stride = wsize;
    // Split-K selection. Every candidate (rounds r of a full chip, LDS buffering) fixes the number of slices so
    // that tiles x slices just fills r rounds; its price is r x (chunks per workgroup + fixed overhead) x chunk
    // time, plus the second-stage reduction that has to read slices x |dw| bytes. Measured constants: a 128x128
    // chunk takes ~4.2 us with 2 resident workgroups per CU and ~5.6 us with 3; HBM-side reduce ~3 TB/s.
    const int tiles = p.tiles_m * p.tiles_n * batch;
    const size_t max_by_ws = workspace ? workspace_bytes / ((size_t)wsize * sizeof(float)) : 0;
    static int forced_blocks = -1;
    if (forced_blocks < 0) {
        const char* e = getenv("DENET_WGRAD_BLOCKS");
        forced_blocks = e ? atoi(e) : 0;
    }
    int splits = 1, wg_nbuf = 2;
    {
        const double chunk_us2 = (big_m ? 4.2 : 2.4), chunk_us3 = chunk_us2 * 1.5 / 1.12;
        double best = 1e300;
        TuneVal tv = {0, 0, 0};
        const bool have_tv = tuned_choice(key, &tv);
        for (int nbuf = 1; nbuf <= 3; ++nbuf) {
            if ((have_tv ? tv.nbuf : (forced_nbuf() ? forced_nbuf() : 1)) != nbuf) continue;
            const int slots = 256 * (nbuf == 1 ? 3 : (nbuf == 2 ? 2 : 1));
            for (int r = 1; r <= 6; ++r) {
                if (have_tv && tv.rounds != r) continue;
                int sp = forced_blocks ? ceil_div(forced_blocks, tiles) : (r * slots) / tiles;
                if (sp < 1) sp = 1;
                if (sp > p.ksteps / 4) sp = p.ksteps / 4 > 0 ? p.ksteps / 4 : 1;
                if (sp > 512) sp = 512;
                if ((size_t)sp > max_by_ws && sp > 1) sp = max_by_ws > 0 ? (int)max_by_ws : 1;
                const int per = ceil_div(p.ksteps, sp);
                sp = ceil_div(p.ksteps, per);
                const long nb = (long)tiles * sp;
                const double rounds = (double)((nb + slots - 1) / slots);
                const double t_main = rounds * (per + 2.0) * (nbuf == 1 ? chunk_us3 : chunk_us2);
                const double t_red = sp > 1 ? 3.0 + (double)sp * wsize * 4.0 / 3.0e6 : 0.0;
                if (t_main + t_red < best) {
                    best = t_main + t_red;
                    splits = sp;
                    wg_nbuf = nbuf;
                }
            }
        }
    }
    if (splits <= 1) {
        splits = 1;
        p.out = dw;
    } else {
        p.out = workspace;
    }
    p.steps_per_split = ceil_div(p.ksteps, splits);
    splits = ceil_div(p.ksteps, p.steps_per_split);
    if (big_m)
        rc = LAUNCH_NBUF(MODE_WGRAD, 128, 128, wg_nbuf, p, splits, stream);
    else if (narrow)
        rc = LAUNCH_NBUF(MODE_WGRAD, 64, 64, wg_nbuf, p, splits, stream);
    else
        rc = LAUNCH_NBUF(MODE_WGRAD, 64, 128, wg_nbuf, p, splits, stream);
    if (rc) return rc;
    if (splits > 1) {
        DENET_CHECK_ARG(wsize % 4 == 0, "conv_wgrad: weight size not a multiple of 4");
        long n4 = wsize / 4;
        int blocks = (int)((n4 + 63) / 64);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, workspace, dw, n4, splits, n4);
        DENET_CHECK_LAUNCH("splitk_reduce");
    }
    return DENET_OK;
}

}  // namespace

// the same live timing for the kernels of other files (wino2f.hip: mode 10 = wino2f_ws_kernel, 11 = wino2f_wgrad_kernel)
int denet_prof_begin(int mode, int bm, int bn, int nbuf, hipStream_t stream) { return prof_begin(mode, bm, bn, nbuf, stream); }
void denet_prof_end(int idx, hipStream_t stream) { prof_end(idx, stream); }

extern "C" int denet_conv_profile(int enable) {
    for (auto& r : g_prof) {
        if (r.a) (void)hipEventDestroy(r.a);
        if (r.b) (void)hipEventDestroy(r.b);
    }
    g_prof.clear();
    g_prof_on = enable != 0;
    g_prof_events = enable != 2;
    return DENET_OK;
}

extern "C" int denet_conv_profile_count(void) { return (int)g_prof.size(); }

extern "C" int denet_conv_profile_read(int i, float* ms, int* mode, int* bm, int* bn, int* nbuf) {
    DENET_CHECK_ARG(i >= 0 && i < (int)g_prof.size() && ms, "conv_profile_read: index %d out of range", i);
    *ms = 0.f;
    if (g_prof[i].a) {               // (a launch-trace record has no events: duration 0)
        hipError_t e = hipEventSynchronize(g_prof[i].b);
        if (e == hipSuccess) e = hipEventElapsedTime(ms, g_prof[i].a, g_prof[i].b);
        if (e != hipSuccess) {
            denet_set_error("conv_profile_read: %s", hipGetErrorString(e));
            return -(int)e;
        }
    }
    if (mode) *mode = g_prof[i].cfg[0];
    if (bm) *bm = g_prof[i].cfg[1];
    if (bn) *bn = g_prof[i].cfg[2];
    if (nbuf) *nbuf = g_prof[i].cfg[3];
    return DENET_OK;
}

extern "C" int denet_conv_fwd(const float* x, const float* w, const float* bias, const float* add, float* y, int N,
                              int H, int W, int C, int K, int R, int S, int S_real, int stride, int pad, int OH,
                              int OW, hipStream_t stream);
extern "C" int denet_conv_fwd_act(const float* x, const float* w, const float* bias, const float* add, float* y, int relu,
                                  int N, int H, int W, int C, int K, int R, int S, int S_real, int stride, int pad, int OH,
                                  int OW, hipStream_t stream);
extern "C" int denet_conv_dgrad(const float* dy, const float* w, const float* add, float* dx, int N, int H, int W,
                                int C, int K, int R, int S, int S_real, int stride, int pad, int OH, int OW,
                                hipStream_t stream);
extern "C" int denet_conv_wgrad(const float* x, const float* dy, float* dw, float* workspace, size_t workspace_bytes,
                                int N, int H, int W, int C, int K, int R, int S, int S_real, int stride, int pad,
                                int OH, int OW, hipStream_t stream);

// Batch of `batch` independent GEMMs out_b[M,Nc] = a_b[M,Kc] * w_b[Nc,Kc]^T (all row-major, K contiguous): the component
// products of the Winograd path (winograd.hip). Strides in elements. A measured choice (denet_gemm_batched_tune) is
// TuneVal{tile, nbuf, 0}: tile 0 = 128 x 128, 1 = 128 x 64; nbuf = loop structure. (Round 2's persistent stream-K kernel for
// these products - 10-17 % faster alone, neutral inside the training step in every measurement incl. round 3's - is gone.)
int denet_gemm_batched_nt(const float* a, const float* w, float* out, int batch, int M, int Nc, int Kc, long stride_a,
                          long stride_w, long stride_out, hipStream_t stream) {
    DENET_CHECK_ARG(a && w && out && batch > 0 && batch <= 65535 && M > 0, "gemm_batched: bad arguments");
    DENET_CHECK_ARG(Nc % 32 == 0 && Kc % 32 == 0, "gemm_batched: Nc, Kc must be multiples of 32");
    DENET_CHECK_ARG((long)M * Kc * 4 < 0xF0000000L && (long)Nc * Kc * 4 < 0xF0000000L, "gemm_batched: operand too large");
    IgemmParams p = {};
    p.act = a; p.wgt = w; p.out = out;
    p.N = 1; p.H = 1; p.W = M; p.C = Kc; p.OH = 1; p.OW = M; p.K = Nc;
    p.R = 1; p.S = 1; p.S_real = 1; p.stride = 1; p.sshift = 0; p.pad = 0;
    p.act_bytes = (unsigned)((size_t)M * Kc * 4); p.wgt_bytes = (unsigned)((size_t)Nc * Kc * 4);
    p.batch_act = stride_a; p.batch_wgt = stride_w; p.batch_out = stride_out;
    p.M = M; p.NC = Nc; p.ksteps = Kc / BK; p.steps_per_split = p.ksteps; p.npix = M;
    p.div_row_hw.init(M); p.div_row_w.init(M);
    p.tiles_m = ceil_div(M, 128);
    // measured choice (denet_gemm_batched_tune) or: single-buffer loop for short reductions, 128x64 when N is small
    int nbuf = (p.ksteps >= 24) ? 2 : 1;
    bool big = Nc >= 128 && (long)p.tiles_m * ceil_div(Nc, 128) * batch >= 1024;
    TuneVal tv;
    if (tuned_choice(tune_key(3, batch, 1, M, Kc, Nc, 1, 1, 1, 1, 0), &tv) && tv.nbuf < 10) {      // (>= 10: a stream-K record of round 2)
        nbuf = tv.nbuf;
        big = (tv.tile == 0) && Nc >= 128;
    }
    if (big) {
        p.tiles_n = ceil_div(Nc, 128);
        return LAUNCH_NBUF(MODE_FWD, 128, 128, nbuf, p, batch, stream);
    }
    p.tiles_n = ceil_div(Nc, 64);
    return LAUNCH_NBUF(MODE_FWD, 128, 64, nbuf, p, batch, stream);
}

// Batch of `batch` independent products dw_b[Kr,Cc] = sum_t dy_b[t,Kr] * x_b[t,Cc] (row-major, t = reduction) through
// the weight-gradient kernel (1x1 geometry, split-K over t): the filter-gradient products of the Winograd path.
int denet_wgrad_batched(const float* x, const float* dy, float* dw, float* workspace, size_t workspace_bytes, int batch,
                        int T, int Cc, int Kr, hipStream_t stream) {
    DENET_CHECK_ARG(x && dy && dw && batch > 0 && batch <= 65535 && T > 0, "wgrad_batched: bad arguments");
    DENET_CHECK_ARG(Cc % 32 == 0 && Kr % 32 == 0, "wgrad_batched: channel counts must be multiples of 32");
    DENET_CHECK_ARG((long)T * Cc * 4 < 0xF0000000L && (long)T * Kr * 4 < 0xF0000000L, "wgrad_batched: operand too large");
    IgemmParams p = {};
    p.act = x; p.wgt = dy;
    p.N = 1; p.H = 1; p.W = T; p.C = Cc; p.OH = 1; p.OW = T; p.K = Kr;
    p.R = 1; p.S = 1; p.S_real = 1; p.stride = 1; p.sshift = 0; p.pad = 0;
    p.act_bytes = (unsigned)((size_t)T * Cc * 4); p.wgt_bytes = (unsigned)((size_t)T * Kr * 4);
    p.batch = batch; p.batch_act = (long)T * Cc; p.batch_wgt = (long)T * Kr; p.batch_out = (long)Kr * Cc;
    p.M = Kr; p.NC = Cc; p.npix = T;
    p.wg_split_slow = 1;
    p.div_row_hw.init(T); p.div_row_w.init(T);
    return wgrad_dispatch(p, Kr, dw, workspace, workspace_bytes, tune_key(4, batch, 1, T, Cc, Kr, 1, 1, 1, 1, 0), stream);
}

// measures the split / loop structure of the batched weight-gradient product (synchronises the stream)
int denet_wgrad_batched_tune(const float* x, const float* dy, float* dw, float* workspace, size_t workspace_bytes, int batch,
                             int T, int Cc, int Kr, hipStream_t stream) {
    const TuneKey key = tune_key(4, batch, 1, T, Cc, Kr, 1, 1, 1, 1, 0);
    if (g_tuned.count(key)) return DENET_OK;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        denet_set_error("wgrad_batched_tune: hipEventCreate failed");
        return DENET_ERR_ARG;
    }
    const bool prof_was_on = g_prof_on;
    g_prof_on = false;
    float best_ms = 1e30f;
    TuneVal best = {0, 1, 1};
    int rc = DENET_OK;
    for (int nbuf = 1; nbuf <= 2 && !rc; ++nbuf)
        for (int r = 1; r <= (nbuf == 1 ? 6 : 3) && !rc; ++r) {
            t_try = TuneVal{0, nbuf, r};
            rc = denet_wgrad_batched(x, dy, dw, workspace, workspace_bytes, batch, T, Cc, Kr, stream);
            float ms_min = 1e30f;
            for (int rep = 0; rep < 3 && !rc; ++rep) {
                (void)hipEventRecord(e0, stream);
                rc = denet_wgrad_batched(x, dy, dw, workspace, workspace_bytes, batch, T, Cc, Kr, stream);
                (void)hipEventRecord(e1, stream);
                if (hipEventSynchronize(e1) != hipSuccess) rc = DENET_ERR_ARG;
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < ms_min) ms_min = ms;
            }
            if (!rc && ms_min < best_ms) {
                best_ms = ms_min;
                best = t_try;
            }
        }
    t_try = TuneVal{-1, 0, 0};
    g_prof_on = prof_was_on;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) return rc;
    g_tuned[key] = best;
    return DENET_OK;
}

// measures the launch configuration of the batched GEMM for these sizes (synchronises the stream)
int denet_gemm_batched_tune(const float* a, const float* w, float* out, int batch, int M, int Nc, int Kc, long stride_a,
                            long stride_w, long stride_out, hipStream_t stream) {
    const TuneKey key = tune_key(3, batch, 1, M, Kc, Nc, 1, 1, 1, 1, 0);
    if (g_tuned.count(key)) return DENET_OK;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        denet_set_error("gemm_batched_tune: hipEventCreate failed");
        return DENET_ERR_ARG;
    }
    const bool prof_was_on = g_prof_on;
    g_prof_on = false;
    float best_ms = 1e30f;
    TuneVal best = {1, 1, 0};
    int rc = DENET_OK;
    std::vector<TuneVal> cand;
    for (int t = (Nc >= 128 ? 0 : 1); t < 2; ++t) {
        for (int nbuf = 1; nbuf <= 2; ++nbuf) cand.push_back(TuneVal{t, nbuf, 0});
    }
    for (const TuneVal& c : cand) {
        if (rc) break;
        t_try = c;
        rc = denet_gemm_batched_nt(a, w, out, batch, M, Nc, Kc, stride_a, stride_w, stride_out, stream);
        float ms_min = 1e30f;
        for (int rep = 0; rep < 3 && !rc; ++rep) {
            (void)hipEventRecord(e0, stream);
            rc = denet_gemm_batched_nt(a, w, out, batch, M, Nc, Kc, stride_a, stride_w, stride_out, stream);
            (void)hipEventRecord(e1, stream);
            if (hipEventSynchronize(e1) != hipSuccess) rc = DENET_ERR_ARG;
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < ms_min) ms_min = ms;
        }
        if (!rc && ms_min < best_ms) {
            best_ms = ms_min;
            best = t_try;
        }
    }
    t_try = TuneVal{-1, 0, 0};
    g_prof_on = prof_was_on;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) return rc;
    g_tuned[key] = best;
    return DENET_OK;
}

// Times the candidate launch configurations of one convolution pass on the caller's own buffers and remembers the
// fastest for this geometry (every candidate computes the same result, so `out` is valid afterwards). mode 0 = fwd
// (a = x, b = w, bias/add as in denet_conv_fwd), 1 = dgrad (a = dy, b = w, add), 2 = wgrad (a = x, b = dy, out = dw,
// workspace). The ONLY entry point that synchronises the stream. fwd/dgrad candidates accumulate in the same order
// (bit-identical results); wgrad candidates differ in the split-K grouping, the choice is fixed for the process.
extern "C" int denet_conv_tune(int mode, const float* a, const float* b, const float* bias, const float* add, float* out,
                               float* workspace, size_t workspace_bytes, int N, int H, int W, int C, int K, int R, int S,
                               int S_real, int stride, int pad, int OH, int OW, hipStream_t stream) {
    DENET_CHECK_ARG(mode >= 0 && mode <= 2, "conv_tune: mode must be 0, 1 or 2");
    const TuneKey key = tune_key(mode, N, H, W, C, K, R, S, S_real, stride, pad);
    auto run = [&]() -> int {
        if (mode == MODE_FWD) return denet_conv_fwd(a, b, bias, add, out, N, H, W, C, K, R, S, S_real, stride, pad, OH, OW, stream);
        if (mode == MODE_DGRAD) return denet_conv_dgrad(a, b, add, out, N, H, W, C, K, R, S, S_real, stride, pad, OH, OW, stream);
        return denet_conv_wgrad(a, b, out, workspace, workspace_bytes, N, H, W, C, K, R, S, S_real, stride, pad, OH, OW, stream);
    };
    std::vector<TuneVal> cand;
    if (mode == MODE_WGRAD) {
        for (int r = 1; r <= 6; ++r) cand.push_back(TuneVal{0, 1, r});
        for (int r = 1; r <= 3; ++r) cand.push_back(TuneVal{0, 2, r});
    } else {
        const int ndim = (mode == MODE_FWD) ? K : C;
        for (int t = (ndim >= 128 ? 0 : 1); t < 2; ++t)
            for (int nbuf = 1; nbuf <= 2; ++nbuf) cand.push_back(TuneVal{t, nbuf, 0});
    }
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        denet_set_error("conv_tune: hipEventCreate failed");
        return DENET_ERR_ARG;
    }
    const bool prof_was_on = g_prof_on;
    g_prof_on = false;
    float best_ms = 1e30f;
    TuneVal best = cand[0];
    int rc = DENET_OK;
    for (const TuneVal& c : cand) {
        t_try = c;
        rc = run();                                  // warm-up (also sets the LDS attribute of the instantiation)
        if (rc) break;
        float ms_min = 1e30f;
        for (int rep = 0; rep < 3 && !rc; ++rep) {
            (void)hipEventRecord(e0, stream);
            rc = run();
            (void)hipEventRecord(e1, stream);
            if (hipEventSynchronize(e1) != hipSuccess) rc = DENET_ERR_ARG;
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < ms_min) ms_min = ms;
        }
        if (rc) break;
        if (ms_min < best_ms) {
            best_ms = ms_min;
            best = c;
        }
    }
    t_try = TuneVal{-1, 0, 0};
    g_prof_on = prof_was_on;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) return rc;
    g_tuned[key] = best;
    return run();                                    // leave the result of the chosen configuration in `out`
}

extern "C" int denet_conv_tuned(int mode, int N, int H, int W, int C, int K, int R, int S, int S_real, int stride,
                                int pad, int* tile, int* nbuf, int* rounds) {
    auto it = g_tuned.find(tune_key(mode, N, H, W, C, K, R, S, S_real, stride, pad));
    if (it == g_tuned.end()) return 1;
    if (tile) *tile = it->second.tile;
    if (nbuf) *nbuf = it->second.nbuf;
    if (rounds) *rounds = it->second.rounds;
    return DENET_OK;
}

// ---- persistence of the measured launch configurations (reproducible runs: the same geometry always runs the same kernels)
// a record is 14 ints: the 11 key fields (mode, N, H, W, C, K, R, S, S_real, stride, pad; modes 3 / 4 = batched component
// GEMM / batched filter-gradient product of the Winograd passes, see tune_key call sites) + tile, nbuf, rounds
extern "C" int denet_tune_export(int* records, int capacity) {
    int n = 0;
    for (const auto& kv : g_tuned) {
        if (records && n < capacity) {
            int* r = records + (size_t)n * 14;
            for (int i = 0; i < 11; ++i) r[i] = kv.first.v[i];
            r[11] = kv.second.tile; r[12] = kv.second.nbuf; r[13] = kv.second.rounds;
        }
        ++n;
    }
    return n;
}

extern "C" int denet_tune_import(const int* records, int count) {
    DENET_CHECK_ARG(records || count == 0, "tune_import: null records");
    for (int n = 0; n < count; ++n) {
        const int* r = records + (size_t)n * 14;
        TuneKey k;
        for (int i = 0; i < 11; ++i) k.v[i] = r[i];
        DENET_CHECK_ARG(r[0] >= 0 && r[0] <= 4 && r[11] >= 0 && r[11] <= 1 && r[12] >= 1 && r[12] <= 12 && r[13] >= 0 && r[13] <= 6,
                        "tune_import: record %d is not a launch configuration", n);
        g_tuned[k] = TuneVal{r[11], r[12], r[13]};
    }
    return DENET_OK;
}

extern "C" int denet_tune_clear(void) {
    g_tuned.clear();
    return DENET_OK;
}

extern "C" int denet_conv_last_config(int* mode, int* bm, int* bn, int* nbuf, int* grid_y) {
    if (mode) *mode = g_last_cfg[0];
    if (bm) *bm = g_last_cfg[1];
    if (bn) *bn = g_last_cfg[2];
    if (nbuf) *nbuf = g_last_cfg[3];
    if (grid_y) *grid_y = g_last_cfg[4];
    return DENET_OK;
}

extern "C" int denet_conv_fwd(const float* x, const float* w, const float* bias, const float* add, float* y, int N,
                              int H, int W, int C, int K, int R, int S, int S_real, int stride, int pad, int OH,
                              int OW, hipStream_t stream) {
    return denet_conv_fwd_act(x, w, bias, add, y, 0, N, H, W, C, K, R, S, S_real, stride, pad, OH, OW, stream);
}

static int conv_fwd_impl(const float* x, const float* w, const float* bias, const float* add, float* y, int relu,
                         double* stats, int N, int H, int W, int C, int K, int R, int S, int S_real, int stride, int pad,
                         int OH, int OW, hipStream_t stream, const denet_bn_link* sums_of = nullptr);

// forward convolution with an activation in the epilogue: y = act(conv(x, w) + bias + add), relu != 0: max(., 0)
extern "C" int denet_conv_fwd_act(const float* x, const float* w, const float* bias, const float* add, float* y, int relu,
                                  int N, int H, int W, int C, int K, int R, int S, int S_real, int stride, int pad, int OH,
                                  int OW, hipStream_t stream) {
    return conv_fwd_impl(x, w, bias, add, y, relu, nullptr, N, H, W, C, K, R, S, S_real, stride, pad, OH, OW, stream);
}

// the same, and the epilogue also emits the per-channel sums of y for the batch norm that follows (batch_norm.py:50-53;
// batch_norm_relu.py:34-54): stats_partial [ceil(N*OH*OW / 128)][2][K] doubles (sum | sum of squares per row tile);
// *stats_rows receives the row count. Feed both to denet_bn_fwd_train_pre: the normalisation then needs no statistics pass.
extern "C" int denet_conv_fwd_stats(const float* x, const float* w, const float* bias, const float* add, float* y,
                                    double* stats_partial, size_t stats_bytes, int* stats_rows, int N, int H, int W, int C,
                                    int K, int R, int S, int S_real, int stride, int pad, int OH, int OW, hipStream_t stream) {
    DENET_CHECK_ARG(stats_partial && stats_rows, "conv_fwd_stats: null pointer");
    if (!add && denet_conv_stem_ok(0, N, H, W, C, K, R, S, S_real, stride, pad, OH, OW))
        return denet_conv_stem_fwd(x, w, bias, y, stats_partial, stats_bytes, stats_rows, N, H, W, stream);
    const long rows = ((long)N * OH * OW + 127) / 128;
    DENET_CHECK_ARG(stats_bytes >= (size_t)rows * 2 * K * sizeof(double), "conv_fwd_stats: statistics buffer too small");
    *stats_rows = (int)rows;
    return conv_fwd_impl(x, w, bias, add, y, 0, stats_partial, N, H, W, C, K, R, S, S_real, stride, pad, OH, OW, stream);
}

static int conv_fwd_impl(const float* x, const float* w, const float* bias, const float* add, float* y, int relu,
                         double* stats, int N, int H, int W, int C, int K, int R, int S, int S_real, int stride, int pad,
                         int OH, int OW, hipStream_t stream, const denet_bn_link* sums_of) {
    int rc = check_geom(N, H, W, C, K, R, S, S_real, stride, pad, OH, OW);
    if (rc) return rc;
    DENET_CHECK_ARG(x && w && y, "conv_fwd: null pointer");
    if (!add && !stats && denet_conv_stem_ok(0, N, H, W, C, K, R, S, S_real, stride, pad, OH, OW))
        return denet_conv_stem_fwd_act(x, 0, w, bias, y, relu, nullptr, 0, nullptr, N, H, W, stream);
    IgemmParams p = {};
    p.act = x; p.wgt = w; p.out = y; p.bias = bias; p.add = add; p.relu = relu ? 1 : 0; p.stats = stats;
    if (sums_of && stats) {          // the column sums of the epilogue become a batch norm's backward reductions (denet_conv_dgrad_1x1t)
        p.bs_x = sums_of->x; p.bs_y = sums_of->relu ? sums_of->y : nullptr; p.bs_gamma = sums_of->gamma; p.bs_beta = sums_of->beta;
        p.bs_mean = sums_of->mean; p.bs_invstd = sums_of->invstd; p.bs_relu = sums_of->relu;
    }
    p.N = N; p.H = H; p.W = W; p.C = C; p.OH = OH; p.OW = OW; p.K = K;
    p.R = R; p.S = S; p.S_real = S_real; p.stride = stride; p.sshift = ilog2_exact(stride); p.pad = pad;
    p.act_bytes = (unsigned)((size_t)N * H * W * C * 4); p.wgt_bytes = (unsigned)((size_t)K * R * S * C * 4);
    p.M = N * OH * OW; p.NC = K; p.ksteps = R * S * C / BK; p.steps_per_split = p.ksteps;
    p.npix = p.M;
    p.div_row_hw.init(OH * OW); p.div_row_w.init(OW);
    {
        const long tm = ceil_div(p.M, 128);
        const long nb[2] = {tm * ceil_div(K, 128), tm * ceil_div(K, 64)};
        const double area[2] = {128.0 * 128.0, 128.0 * 64.0}, eff[2] = {1.0, 0.85};
        // measured (tools/bench_conv.py): the pipelined 2-buffer loop wins on forward once the reduction has >= 24
        // chunks (+4...14 %); shorter reductions (stem, 64-channel 3x3, 1x1 on 128 channels) amortise its longer
        // prologue badly and run faster on the single-buffer loop at 3-4 workgroups per CU
        const int want = (p.ksteps >= 24) ? 2 : 1;
        Choice c = (K >= 128) ? choose_launch(nb, area, eff, 2, want) : choose_launch(nb + 1, area + 1, eff + 1, 1, want);
        int tile = (K >= 128) ? c.tile : 1;
        TuneVal tv;
        if (tuned_choice(tune_key(MODE_FWD, N, H, W, C, K, R, S, S_real, stride, pad), &tv)) {
            tile = (K >= 128) ? tv.tile : 1;
            c.nbuf = tv.nbuf;
        }
        p.tiles_m = (int)tm;
        if (tile == 0) {
            p.tiles_n = ceil_div(K, 128);
            take_final(p, 128, 1);
            return LAUNCH_NBUF(MODE_FWD, 128, 128, c.nbuf, p, 1, stream);
        }
        p.tiles_n = ceil_div(K, 64);
        take_final(p, 64, 1);
        return LAUNCH_NBUF(MODE_FWD, 128, 64, c.nbuf, p, 1, stream);
    }
}

static int conv_dgrad_impl(const float* dy, const float* w, const float* add, float* dx, const denet_bn_link* sums_of,
                           double* stats, int N, int H, int W, int C, int K, int R, int S, int S_real, int stride, int pad, int OH,
                           int OW, hipStream_t stream, bool transposed = false);

extern "C" int denet_conv_dgrad(const float* dy, const float* w, const float* add, float* dx, int N, int H, int W,
                                int C, int K, int R, int S, int S_real, int stride, int pad, int OH, int OW,
                                hipStream_t stream) {
    return conv_dgrad_impl(dy, w, add, dx, nullptr, nullptr, N, H, W, C, K, R, S, S_real, stride, pad, OH, OW, stream);
}

// denet_conv_dgrad of a layer whose dx is the gradient of the OUTPUT of the batch norm `sums_of` (see
// denet_conv_wino_dgrad_sums): the epilogue also writes that layer's backward reductions, stats_partial [rows][2][C] doubles
// with rows = stride^2 * ceil(N*(H/stride)*(W/stride) / 128) (a row per parity class of input pixels and row tile; *stats_rows
// receives it, 0 and no sums when the buffer is too small)
extern "C" int denet_conv_dgrad_sums(const float* dy, const float* w, const float* add, float* dx, const denet_bn_link* sums_of,
                                     double* stats_partial, size_t stats_bytes, int* stats_rows, int N, int H, int W, int C, int K,
                                     int R, int S, int S_real, int stride, int pad, int OH, int OW, hipStream_t stream) {
    DENET_CHECK_ARG(sums_of && stats_partial && stats_rows, "conv_dgrad_sums: null pointer");
    DENET_CHECK_ARG(sums_of->x && sums_of->mean && sums_of->invstd && (!sums_of->relu || sums_of->y || (sums_of->gamma && sums_of->beta)),
                    "conv_dgrad_sums: incomplete batch-norm description");
    const bool geom = stride >= 1 && H % stride == 0 && W % stride == 0;
    const long rows = geom ? (long)stride * stride * (((long)N * (H / stride) * (W / stride) + 127) / 128) : 0;
    const bool ok = geom && stats_bytes >= (size_t)rows * 2 * C * sizeof(double);
    *stats_rows = ok ? (int)rows : 0;
    return conv_dgrad_impl(dy, w, add, dx, ok ? sums_of : nullptr, ok ? stats_partial : nullptr, N, H, W, C, K, R, S, S_real, stride,
                           pad, OH, OW, stream);
}

// denet_conv_dgrad / denet_conv_dgrad_sums with the filter given transposed: wt [R][S][C][K] = denet_transpose_f32 of w [K][R*S*C]
// (K rows, R*S*C columns). The filter operand is then reduction-contiguous like the forward pass's (one 16-byte LDS read per
// fragment instead of four scalar ones); the products and their order are those of denet_conv_dgrad: bit-identical results.
// sums_of / stats_partial / stats_rows as in denet_conv_dgrad_sums, or all null.
extern "C" int denet_conv_dgrad_t(const float* dy, const float* wt, const float* add, float* dx, const denet_bn_link* sums_of,
                                  double* stats_partial, size_t stats_bytes, int* stats_rows, int N, int H, int W, int C, int K,
                                  int R, int S, int S_real, int stride, int pad, int OH, int OW, hipStream_t stream) {
    bool ok = false;
    if (sums_of) {
        DENET_CHECK_ARG(stats_partial && stats_rows, "conv_dgrad_t: null pointer");
        DENET_CHECK_ARG(sums_of->x && sums_of->mean && sums_of->invstd && (!sums_of->relu || sums_of->y || (sums_of->gamma && sums_of->beta)),
                        "conv_dgrad_t: incomplete batch-norm description");
        const bool geom = stride >= 1 && H % stride == 0 && W % stride == 0;
        const long rows = geom ? (long)stride * stride * (((long)N * (H / stride) * (W / stride) + 127) / 128) : 0;
        ok = geom && stats_bytes >= (size_t)rows * 2 * C * sizeof(double);
        *stats_rows = ok ? (int)rows : 0;
    }
    return conv_dgrad_impl(dy, wt, add, dx, ok ? sums_of : nullptr, ok ? stats_partial : nullptr, N, H, W, C, K, R, S, S_real, stride,
                           pad, OH, OW, stream, true);
}

// the data gradient of a 1x1 stride-1 convolution as a FORWARD product over the transposed filter wt [C][K]
// (dx[pix][c] = sum_k dy[pix][k] wt[c][k]): the forward loop reads both operands reduction-contiguous and runs the pipelined
// 128 x 128 tile (the head layers: 118 -> 137, 100 -> 124 TFLOP/s against the k-major filter reads of the data-gradient mode),
// same products in the same order - bit-identical to denet_conv_dgrad. sums_of / stats_partial as in denet_conv_dgrad_sums
// (both null: no sums), rows = ceil(N*H*W / 128).
extern "C" int denet_conv_dgrad_1x1t(const float* dy, const float* wt, const float* add, float* dx, const denet_bn_link* sums_of,
                                     double* stats_partial, size_t stats_bytes, int* stats_rows, int N, int H, int W, int C, int K,
                                     hipStream_t stream) {
    DENET_CHECK_ARG(dy && wt && dx, "conv_dgrad_1x1t: null pointer");
    bool ok = false;
    if (sums_of) {
        DENET_CHECK_ARG(stats_partial && stats_rows, "conv_dgrad_1x1t: null pointer");
        DENET_CHECK_ARG(sums_of->x && sums_of->mean && sums_of->invstd && (!sums_of->relu || sums_of->y || (sums_of->gamma && sums_of->beta)),
                        "conv_dgrad_1x1t: incomplete batch-norm description");
        const long rows = ((long)N * H * W + 127) / 128;
        ok = stats_bytes >= (size_t)rows * 2 * C * sizeof(double);
        *stats_rows = ok ? (int)rows : 0;
    }
    return conv_fwd_impl(dy, wt, nullptr, add, dx, 0, ok ? stats_partial : nullptr, N, H, W, K, C, 1, 1, 1, 1, 0, H, W, stream,
                         ok ? sums_of : nullptr);
}

static int conv_dgrad_impl(const float* dy, const float* w, const float* add, float* dx, const denet_bn_link* sums_of,
                           double* stats, int N, int H, int W, int C, int K, int R, int S, int S_real, int stride, int pad, int OH,
                           int OW, hipStream_t stream, bool transposed) {
    int rc = check_geom(N, H, W, C, K, R, S, S_real, stride, pad, OH, OW);
    if (rc) return rc;
    DENET_CHECK_ARG(dy && w && dx, "conv_dgrad: null pointer");
    DENET_CHECK_ARG(C >= 32, "conv_dgrad: C < 32 not supported (first layer needs no data gradient)");
    IgemmParams p = {};
    p.act = dy; p.wgt = w; p.out = dx; p.bias = nullptr; p.add = add;
    if (sums_of && stats) {
        p.stats = stats;
        p.bs_x = sums_of->x; p.bs_y = sums_of->relu ? sums_of->y : nullptr; p.bs_gamma = sums_of->gamma; p.bs_beta = sums_of->beta;
        p.bs_mean = sums_of->mean; p.bs_invstd = sums_of->invstd; p.bs_relu = sums_of->relu;
    }
    p.N = N; p.H = H; p.W = W; p.C = C; p.OH = OH; p.OW = OW; p.K = K;
    p.R = R; p.S = S; p.S_real = S_real; p.stride = stride; p.sshift = ilog2_exact(stride); p.pad = pad;
    DENET_CHECK_ARG(H % stride == 0 && W % stride == 0, "conv_dgrad: H, W must be multiples of the stride");
    p.act_bytes = (unsigned)((size_t)N * OH * OW * K * 4); p.wgt_bytes = (unsigned)((size_t)K * R * S * C * 4);
    const int classes = stride * stride;
    p.Hc = H / stride; p.Wc = W / stride;
    p.M = N * p.Hc * p.Wc; p.NC = C; p.ksteps = R * S * K / BK; p.steps_per_split = p.ksteps;
    p.npix = N * OH * OW;
    p.div_row_hw.init(p.Hc * p.Wc); p.div_row_w.init(p.Wc);
    {
        const long tm = ceil_div(p.M, 128);
        const long nb[2] = {tm * ceil_div(C, 128) * classes, tm * ceil_div(C, 64) * classes};
        const double area[2] = {128.0 * 128.0, 128.0 * 64.0}, eff[2] = {1.0, 0.85};
        // dgrad / wgrad read K-outer LDS tiles (4x the fragment-read instructions): the single-buffer loop at 3-4
        // workgroups per CU measured faster than the pipelined one on every DeNet-34 layer but two (within 3 %)
        Choice c = (C >= 128) ? choose_launch(nb, area, eff, 2, 1) : choose_launch(nb + 1, area + 1, eff + 1, 1, 1);
        int tile = (C >= 128) ? c.tile : 1;
        TuneVal tv;
        if (tuned_choice(tune_key(MODE_DGRAD, N, H, W, C, K, R, S, S_real, stride, pad), &tv)) {
            tile = (C >= 128) ? tv.tile : 1;
            c.nbuf = tv.nbuf;
        }
        p.tiles_m = (int)tm;
        if (transposed) {
            // w is wt [R][S][C][K] (MODE_DGRAD_T): both operands reduction-contiguous, as in the forward pass - whose loop choice
            // applies (the pipelined two-buffer loop from 24 chunks per parity class on); DENET_DGRAD_T_NBUF / _TILE: experiments
            static const int env_nbuf = [] { const char* e = getenv("DENET_DGRAD_T_NBUF"); return e ? atoi(e) : 0; }();
            static const int env_tile = [] { const char* e = getenv("DENET_DGRAD_T_TILE"); return e ? atoi(e) : 0; }();
            const int per_class = p.ksteps / classes;
            const int nbuf = env_nbuf ? env_nbuf : (per_class >= 24 ? 2 : 1);
            const int wide = env_tile ? (env_tile == 1) : (C >= 128 && nbuf == 2 && nb[0] >= 512);
            if (wide) {
                p.tiles_n = ceil_div(C, 128);
                take_final(p, 128, classes);
                return LAUNCH_NBUF(MODE_DGRAD_T, 128, 128, nbuf, p, classes, stream);
            }
            p.tiles_n = ceil_div(C, 64);
            take_final(p, 64, classes);
            return LAUNCH_NBUF(MODE_DGRAD_T, 128, 64, nbuf, p, classes, stream);
        }
        if (tile == 0) {
            p.tiles_n = ceil_div(C, 128);
            take_final(p, 128, classes);
            return LAUNCH_NBUF(MODE_DGRAD, 128, 128, c.nbuf, p, classes, stream);
        }
        p.tiles_n = ceil_div(C, 64);
        take_final(p, 64, classes);
        return LAUNCH_NBUF(MODE_DGRAD, 128, 64, c.nbuf, p, classes, stream);
    }
}

extern "C" size_t denet_conv_wgrad_workspace_bytes(int N, int C, int K, int R, int S, int OH, int OW) {
    // upper bound used by the launcher below: at most 512 split slices
    return (size_t)512 * K * R * S * C * sizeof(float);
}

extern "C" int denet_conv_wgrad(const float* x, const float* dy, float* dw, float* workspace, size_t workspace_bytes,
                                int N, int H, int W, int C, int K, int R, int S, int S_real, int stride, int pad,
                                int OH, int OW, hipStream_t stream) {
    int rc = check_geom(N, H, W, C, K, R, S, S_real, stride, pad, OH, OW);
    if (rc) return rc;
    DENET_CHECK_ARG(x && dy && dw, "conv_wgrad: null pointer");
    if (workspace && workspace_bytes >= denet_conv_stem_wgrad_workspace_bytes() &&
        denet_conv_stem_ok(1, N, H, W, C, K, R, S, S_real, stride, pad, OH, OW))
        return denet_conv_stem_wgrad(x, dy, dw, workspace, workspace_bytes, N, H, W, stream);
    IgemmParams p = {};
    p.act = x; p.wgt = dy; p.bias = nullptr; p.add = nullptr;
    p.N = N; p.H = H; p.W = W; p.C = C; p.OH = OH; p.OW = OW; p.K = K;
    p.R = R; p.S = S; p.S_real = S_real; p.stride = stride; p.sshift = ilog2_exact(stride); p.pad = pad;
    p.act_bytes = (unsigned)((size_t)N * H * W * C * 4); p.wgt_bytes = (unsigned)((size_t)N * OH * OW * K * 4);
    p.M = K; p.NC = R * S * C; p.npix = N * OH * OW;
    {
        static int v = -1;
        if (v < 0) {
            const char* e = getenv("DENET_WGRAD_SPLIT_SLOW");
            v = e ? atoi(e) : 1;
        }
        p.wg_split_slow = v;
    }
    p.div_row_hw.init(OH * OW); p.div_row_w.init(OW);
    return wgrad_dispatch(p, K, dw, workspace, workspace_bytes, tune_key(MODE_WGRAD, N, H, W, C, K, R, S, S_real, stride, pad),
                          stream);
}
