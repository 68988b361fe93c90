// Fused Winograd F(2x2,3x3) convolution for the 64-input-channel 3x3 stride-1 layers (the first ResNet stage:
// denet/layer/convolution.py:80-83 forward; its data gradient, model_cnn.py:318, is the same operation on dy with the
// rotated, channel-swapped filters): input transform, the 16 component products and the output transform in ONE kernel.
// Nothing but x (or dy) is read and y (or dx) written - the un-fused passes move the 4x (F2) / 2.25x (F4) expanded V and M
// tensors through HBM, which is what bounds them at 64 channels (winograd.hip: 0.33 ms per pass; the direct kernel 0.34).
//
// Three kernels below share the LDS image of the input (18x18-pixel patch of a 16x16-pixel block as channel-quad planes, even
// and odd columns apart, filled by LDS-DMA with the image border zeroed by the buffer bounds check) and the persistent-
// workgroup scheme (one workgroup per CU, work items strided over the grid, the next item's patch streaming in while this one
// is multiplied): wino2f_ws_kernel (forward / data gradient) and wino2f_wgrad_kernel (filter gradient) + its two reducers.
// Exact fp32 FMA chains; the association differs from the direct kernel (F(2x2): ~1e-6 relative).
#include "common.h"
#include "bn_final.h"
#include "../../include/denet_hip.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

struct W2Params {
    const float* x;      // [N,H,W,64]
    const float* U;      // [16][Co][64] transformed filters (denet_conv_wino_filter, tile 2)
    const float* bias;   // [Co] or null
    const float* add;    // [N,H,W,Co] or null
    float* y;            // [N,H,W,Co]
    double* stats;       // Co = 64: [grid][2][64], a row per workgroup (the sums of all its blocks); else [blocks][2][Co]; or null
    // stats of the BACKWARD kind (bs_x != null; the kernel then computes a data gradient): the tensor written is the gradient of
    // the output of a batch-norm layer whose input was bs_x; the sums are that layer's two reductions, sum(g) and
    // sum(g * xhat) with g = y masked by the layer's ReLU (bs_y > 0, or recomputed from bs_x) - bn_bwd_partial_kernel's
    const float* bs_x;
    const float* bs_y;
    const float* bs_gamma;
    const float* bs_beta;
    const float* bs_mean;
    const float* bs_invstd;
    int bs_relu;
    int N, H, W, Co;
    int by, bx;          // 16x16 output blocks per image
    int nco;             // Co / 64
    int items;           // N * by * bx * nco work items: (block, 64 output channels)
    int relu;            // y = max(y, 0) (the inference fold of a ReLU layer)
    unsigned x_bytes, y_bytes;
    BnFinalDev fin;      // Co = 64 with stats: the last workgroup reduces the rows itself (bn_final.h); counter null: off
};

constexpr int CI = 64;
constexpr int P_ROW = 20, P_PAR = 10, P_PLANE = 368;        // 16-byte slots: row / parity / plane strides of the patch
constexpr int P_BYTES = 16 * P_PLANE * 16;                  // 16 channel-quad planes
constexpr int OOB = (int)0xF0000000u;

// s_waitcnt vmcnt(vm) lgkmcnt(0) [gfx9 encoding: vmcnt = bits 15:14 | 3:0, expcnt 6:4, lgkmcnt 11:8] + s_barrier. The raw
// barrier leaves the newest `vm` vector-memory operations of the wave in flight (LDS-DMA pieces that are not needed yet,
// result stores): __syncthreads() would drain them all.
#define W2_BARRIER(vm)                                                                           \
    {                                                                                            \
        __builtin_amdgcn_s_waitcnt(((vm) & 15) | ((((vm) >> 4) & 3) << 14) | (7 << 4));          \
        __builtin_amdgcn_s_barrier();                                                            \
    }

struct W2Item {
    int n, oy0, ox0, co0, block;
};
__device__ __forceinline__ W2Item w2_item(int bx, int by, int nco, int item) {
    W2Item it;
    it.co0 = (item % nco) * 64;
    it.block = item / nco;
    int b = it.block;
    it.ox0 = (b % bx) * 16;
    b /= bx;
    it.oy0 = (b % by) * 16;
    it.n = b / by;
    return it;
}

// ================================================================================================================
// Forward / data-gradient kernel: the transformed filters stay in REGISTERS. A wave owns two of the 16
// Winograd components (row I = wave / 2, columns J0, J0 + 1, J0 = 2 (wave & 1)) for ALL tiles and ALL output channels:
//   * its filters U[xi][co][ci] for the two components are 2 x 64 x 64 values = 128 registers per lane, loaded once per kernel
//     (row operand of v_mfma_f32_16x16x4_f32: lane = (co mod 16, ci quad)); no filter traffic through LDS, no DMA for it;
//   * per group of 16 tiles (two tile rows) and per 16 input channels a lane reads 2 patch rows x 3 columns of its tile and
//     forms its two components of B^T d B with 5 additions; 32 products follow (2 components x 4 output-channel blocks x 4);
//   * the products of a tile group, M[xi][tile][co], meet in LDS (64 KB, XOR-swizzled by tile): every wave writes its two
//     components, one barrier, then a lane reads the 12 components one output row of its (tile, 4 channels) needs, applies
//     A^T . A, bias / add / ReLU, stores 16 bytes (a wave instruction covers whole 256-byte pixel rows of the output);
//   * the input patch is refilled by LDS-DMA in three row bands, each one tile group after its last reader (below).
constexpr int S_MB_SLOTS = 16 * 16 * 16;                     // M of a tile group: [xi][tile][co quad ^ tile] float4
constexpr int S_LDS_BYTES = P_BYTES + S_MB_SLOTS * 16;       // 159 744
// The patch is refilled for the next item in three row bands, each as soon as no tile group reads it any more and at least one
// whole tile group before it is read again (tile group g reads patch rows 4 g .. 4 g + 5):
//   band A = rows 0..7   (slots   0..159: pieces at 0, 64, 96)    last read by group 1, refilled during group 2
//   band B = rows 8..9   (slots 160..199: one piece at 136)       last read by group 2, refilled during group 3
//   band C = rows 10..17 (slots 200..359: pieces at 200, 264, 296) last read by group 3, refilled during the next item's group 0
// (pieces 2, 3 and 6 overlap their neighbours with the same values of the same item)
__host__ __device__ constexpr int sp_start(int k) { return k == 0 ? 0 : k == 1 ? 64 : k == 2 ? 96 : k == 3 ? 136 : k == 4 ? 200 : k == 5 ? 264 : 296; }

// PF: the launch has output operands to read (an `add`, the tensors of the backward sums): they are prefetched (below)
template <int I, int J0, bool PF>
__device__ __forceinline__ void w2s_run(const W2Params& p, char* smem) {
    f32x4* P = (f32x4*)smem;
    f32x4* MB = (f32x4*)(smem + P_BYTES);
    double* red = (double*)MB;                   // the batch-norm sums of the 8 waves meet in the M buffer (between two items)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = lane & 15, kk = lane >> 4;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    // the output phase's global operands (residual / accumulated gradient `add`, the batch-norm input and output of the backward
    // sums) come through buffer descriptors of their own: an absent tensor is a descriptor of 0 bytes, whose loads return zeros
    // without touching memory - the loads are issued unconditionally, a whole matrix round ahead of their use (below)
    const __amdgpu_buffer_rsrc_t radd = __builtin_amdgcn_make_buffer_rsrc((void*)p.add, 0, p.add ? p.y_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbx = __builtin_amdgcn_make_buffer_rsrc((void*)p.bs_x, 0, p.bs_x ? p.y_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rby = __builtin_amdgcn_make_buffer_rsrc((void*)p.bs_y, 0, (p.bs_x && p.bs_relu && p.bs_y) ? p.y_bytes : 0u, 0x00020000);

    // piece n = 0..13 of this wave: plane 2 w + n / 7, piece n % 7. The slot -> pixel arithmetic is redone per piece (a dozen
    // integer instructions, 14 pieces per item) rather than kept in registers: the filters leave none to spare.
    auto piece = [&](int n, const W2Item& it) {
        const int plane = 2 * wave + n / 7, k = n % 7;
        int ln = lane;
        asm volatile("" : "+v"(ln));                // opaque: keeps this arithmetic here instead of hoisted into 14 register sets
        const int sl = sp_start(k) + ln;
        const int row = (sl * 3277) >> 16;           // sl / 20 for sl < 1 << 14
        const int r = sl - row * P_ROW;
        const int par = r >= P_PAR ? 1 : 0, col = r - par * P_PAR;
        const int iy = it.oy0 - 1 + row, ix = it.ox0 - 1 + 2 * col + par;
        const bool ok = col < 9 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const int off = (((it.n * p.H + iy) * p.W + ix) * CI + plane * 4) * 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(P + plane * P_PLANE + sp_start(k)), 16, ok ? off : OOB, 0, 0, 0);
    };

    int item = blockIdx.x;
    if (item >= p.items) return;
    W2Item cur = w2_item(p.bx, p.by, p.nco, item);
#pragma unroll
    for (int n = 0; n < 14; ++n) piece(n, cur);

    // the two components' filters: row operand, lane = (co = 16 cb + t, ci = 16 r + 4 kk + j)
    f32x4 Ur[2][4][4];
    int u_co0 = -1;
    auto load_u = [&](int co0) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Ur[a][cb][r] = *(const f32x4*)(p.U + ((long)(4 * I + J0 + a) * p.Co + co0 + 16 * cb + t) * CI + 16 * r + 4 * kk);
        u_co0 = co0;
    };
    load_u(cur.co0);

    constexpr int RA = I == 0 ? 0 : I == 2 ? 2 : 1, RB = I == 0 ? 2 : I == 1 ? 2 : I == 2 ? 1 : 3;
    constexpr float SG = I == 1 ? 1.f : -1.f;                  // t[I] = d[RA] + SG d[RB]
    constexpr int C0 = J0 == 0 ? 0 : 1;                        // columns C0, C0 + 1, C0 + 2 of t
    // the lane's tile of a group: tile row 2 g + (t >> 3), column t & 7; patch reads of round r: plane 4 r + kk
    const f32x4* Pl = P + kk * P_PLANE + (2 * (t >> 3)) * P_ROW + (t & 7);
    auto load_d = [&](int g, int r, f32x4 (&d)[2][3]) {
        const f32x4* c = Pl + 4 * r * P_PLANE + 4 * g * P_ROW;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int bb = C0 + b;
            d[0][b] = c[RA * P_ROW + (bb & 1) * P_PAR + (bb >> 1)];
            d[1][b] = c[RB * P_ROW + (bb & 1) * P_PAR + (bb >> 1)];
        }
    };
    // output phase: lane = (4 output channels tq = t, tile 2 w + (kk & 1) of the group, output row oi = kk >> 1)
    const int otile = 2 * wave + (kk & 1), oi = kk >> 1;
    const f32x4* Mo = MB + (4 * oi * 16 + otile) * 16 + (t ^ otile);
    const float osg = oi ? -1.f : 1.f;
    f32x4* const Mw0 = MB + ((4 * I + J0) * 16 + t) * 16;       // + a * 256 + ((4 cb + kk) ^ t)

    W2_BARRIER(0);
    f32x4 d[2][3];
    load_d(0, 0, d);
    f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = {0.f, 0.f, 0.f, 0.f};
    double row_a = 0.0, row_b = 0.0;      // Co = 64: the workgroup's row of column sums (lane tid < 64 owns channel tid), stored at the end
    while (true) {
        const int next = item + gridDim.x;
        const bool has_next = next < p.items;
        const W2Item nxt = w2_item(p.bx, p.by, p.nco, has_next ? next : item);
        if (cur.co0 != u_co0) load_u(cur.co0);
#pragma unroll 1
        for (int g = 0; g < 4; ++g) {
            f32x4 acc[2][4];
            f32x4 pf[6];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x4 V[2];
                {
                    f32x4 tc[3];
#pragma unroll
                    for (int b = 0; b < 3; ++b) tc[b] = d[0][b] + SG * d[1][b];
                    if (J0 == 0) {
                        V[0] = tc[0] - tc[2];
                        V[1] = tc[1] + tc[2];
                    } else {
                        V[0] = tc[1] - tc[0];
                        V[1] = tc[0] - tc[2];
                    }
                }
                // the next round's patch values (after the last round of the item: the next item's first, whose band landed
                // before the barrier of tile group 3)
                if (r < 3) load_d(g, r + 1, d);
                // last round: the patch registers are free until the next group - they carry this group's output operands, asked
                // for now and used a matrix round and two barriers later
                if (PF && r == 3) {
                    const int ty = 2 * g + (otile >> 3), tx = otile & 7;
                    const int oy = cur.oy0 + 2 * ty + oi, ox = cur.ox0 + 2 * tx;
                    const int ob = (oy < p.H && ox < p.W) ? ((((cur.n * p.H + oy) * p.W + ox) * p.Co + cur.co0 + 4 * t) * 4) : OOB;
                    pf[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(radd, ob, 0, 0));
                    pf[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(radd, ob, p.Co * 4, 0));
                    pf[2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbx, ob, 0, 0));
                    pf[3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbx, ob, p.Co * 4, 0));
                    pf[4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rby, ob, 0, 0));
                    pf[5] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rby, ob, p.Co * 4, 0));
                }
                // patch refill, two pieces per round (the wave's two planes): band C of THIS item during its tile group 0 (the first
                // item came complete), bands A and B of the NEXT item during tile groups 2 and 3
                if (g == 0 && r < 3 && item != (int)blockIdx.x) {
                    piece(4 + r, cur);
                    piece(7 + 4 + r, cur);
                }
                if (g == 2 && r < 3 && has_next) {
                    piece(r, nxt);
                    piece(7 + r, nxt);
                }
                if (g == 3 && r == 0 && has_next) {
                    piece(3, nxt);
                    piece(7 + 3, nxt);
                }
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb)
                            acc[a][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ur[a][cb][r][j], V[a][j], (r == 0 && j == 0) ? zero : acc[a][cb], 0, 0, 0);
            }
            // every wave has read the previous group's M; this group's goes in
            W2_BARRIER(63);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) Mw0[a * 256 + ((4 * cb + kk) ^ t)] = acc[a][cb];
            // all 16 components are there. The band pieces issued during this group (6 in groups 0 and 2, 2 in group 3) may still
            // fly: a later barrier waits for them, when they are more than a tile group old and before their rows are read (band
            // A, issued in group 2, by the barrier of group 3; band B by that of the next item's group 0; band C by that of group 1)
            // (+ 6: the output operands asked for in the last round, younger than every piece)
            if (g == 0 || g == 2) W2_BARRIER(PF ? 12 : 6)
            else if (g == 3) W2_BARRIER(PF ? 8 : 2)
            else W2_BARRIER(PF ? 6 : 0)
            // the next group's first patch values (group 3: the next item's); with output operands in flight they sit in the
            // patch registers until the output phase has used them, and the patch values are read after it
            if (!PF) load_d((g + 1) & 3, 0, d);
            // ---- output phase: Y row oi of the lane's tile = A^T M A, + bias, + add, ReLU; 2 pixels x 4 channels ---------------
            {
                f32x4 sv[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const f32x4 m0 = Mo[(0 * 4 + jj) * 256], m1 = Mo[(1 * 4 + jj) * 256], m2 = Mo[(2 * 4 + jj) * 256];
                    sv[jj] = m0 + osg * (m1 + m2);
                }
                f32x4 y0 = sv[0] + sv[1] + sv[2];
                f32x4 y1 = sv[1] - sv[2] - sv[3];
                if (p.bias) {
                    const f32x4 bias4 = *(const f32x4*)(p.bias + cur.co0 + 4 * t);
                    y0 += bias4;
                    y1 += bias4;
                }
                const int ty = 2 * g + (otile >> 3), tx = otile & 7;
                const int oy = cur.oy0 + 2 * ty + oi, ox = cur.ox0 + 2 * tx;
                if (oy < p.H && ox < p.W) {          // blocks at the right / bottom edge of a map that is no multiple of 16 (W even)
                    const long o = (((long)cur.n * p.H + oy) * p.W + ox) * p.Co + cur.co0 + 4 * t;
                    if (PF) {
                        y0 += pf[0];           // zeros without an `add`
                        y1 += pf[1];
                    }
                    if (p.relu) {
                        y0 = __builtin_elementwise_max(y0, f32x4{0.f, 0.f, 0.f, 0.f});
                        y1 = __builtin_elementwise_max(y1, f32x4{0.f, 0.f, 0.f, 0.f});
                    }
                    *(f32x4*)(p.y + o) = y0;
                    *(f32x4*)(p.y + o + p.Co) = y1;
                    if (PF && p.bs_x) {
                        const int cq = cur.co0 + 4 * t;
                        const f32x4 mu = *(const f32x4*)(p.bs_mean + cq), is = *(const f32x4*)(p.bs_invstd + cq);
                        const f32x4 x0 = pf[2], x1 = pf[3];
                        f32x4 g0 = y0, g1 = y1;
                        if (p.bs_relu) {
                            if (p.bs_y) {
                                const f32x4 v0 = pf[4], v1 = pf[5];
#pragma unroll
                                for (int c = 0; c < 4; ++c) {
                                    g0[c] = v0[c] > 0.f ? g0[c] : 0.f;
                                    g1[c] = v1[c] > 0.f ? g1[c] : 0.f;
                                }
                            } else {
                                const f32x4 ga = *(const f32x4*)(p.bs_gamma + cq), be = *(const f32x4*)(p.bs_beta + cq);
#pragma unroll
                                for (int c = 0; c < 4; ++c) {
                                    const float sc = ga[c] * is[c], sh = be[c] - mu[c] * sc;
                                    g0[c] = fmaf(x0[c], sc, sh) > 0.f ? g0[c] : 0.f;
                                    g1[c] = fmaf(x1[c], sc, sh) > 0.f ? g1[c] : 0.f;
                                }
                            }
                        }
                        ssum += g0 + g1;
                        ssq += g0 * ((x0 - mu) * is) + g1 * ((x1 - mu) * is);
                    } else {
                        ssum += y0 + y1;
                        ssq += y0 * y0 + y1 * y1;
                    }
                }
            }
            if (PF) load_d((g + 1) & 3, 0, d);
        }
        if (p.stats) {
            // batch-norm column sums of this block. A lane's fp32 sums cover the 8 values of its four tile groups; from there
            // doubles: over the lanes of a wave that share the channels (kk), then over the waves through the M buffer, which
            // every wave has finished reading once the barrier below is passed
            double ds[4], dq[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                ds[c] = (double)ssum[c];
                dq[c] = (double)ssq[c];
            }
#pragma unroll
            for (int off = 16; off < 64; off <<= 1)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    ds[c] += __shfl_xor(ds[c], off, 64);
                    dq[c] += __shfl_xor(dq[c], off, 64);
                }
            W2_BARRIER(63);
            if (kk == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    red[(wave * 2 + 0) * 64 + 4 * t + c] = ds[c];
                    red[(wave * 2 + 1) * 64 + 4 * t + c] = dq[c];
                }
            }
            W2_BARRIER(63);
            if (tid < 64) {
                double a = 0.0, bq = 0.0;
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    a += red[(w * 2 + 0) * 64 + tid];
                    bq += red[(w * 2 + 1) * 64 + tid];
                }
                if (p.nco == 1) {
                    // one row per WORKGROUP: lane tid owns its channel's two sums over all items (blocks added in item order)
                    row_a += a;
                    row_b += bq;
                } else {
                    // several output-channel chunks per block: one row per block, each item writes its chunk of it
                    double* ps = p.stats + (long)cur.block * 2 * p.Co;
                    ps[cur.co0 + tid] = a;
                    ps[p.Co + cur.co0 + tid] = bq;
                }
            }
            ssum = f32x4{0.f, 0.f, 0.f, 0.f};
            ssq = f32x4{0.f, 0.f, 0.f, 0.f};
            W2_BARRIER(63);                          // `red` is free again
        }
        if (!has_next) break;
        item = next;
        cur = nxt;
    }
    __builtin_amdgcn_s_waitcnt(0);
    if (p.stats && p.nco == 1) {
        if (tid < 64) {
            double* ps = p.stats + (long)blockIdx.x * 2 * p.Co;
            bnf_store(ps + tid, row_a);
            bnf_store(ps + p.Co + tid, row_b);
        }
        // the last workgroup to arrive finishes the batch norm's reduction over the rows of all workgroups (bn_final.h)
        bnf_tail<512>(p.fin, p.stats, (int)gridDim.x, 0, 64, 0, gridDim.x, (int*)red);
    }
}

template <bool PF>
__global__ __launch_bounds__(512, 2) void wino2f_ws_kernel(const W2Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {
        case 0: w2s_run<0, 0, PF>(p, smem); break;
        case 1: w2s_run<0, 2, PF>(p, smem); break;
        case 2: w2s_run<1, 0, PF>(p, smem); break;
        case 3: w2s_run<1, 2, PF>(p, smem); break;
        case 4: w2s_run<2, 0, PF>(p, smem); break;
        case 5: w2s_run<2, 2, PF>(p, smem); break;
        case 6: w2s_run<3, 0, PF>(p, smem); break;
        default: w2s_run<3, 2, PF>(p, smem); break;
    }
}

// ================================================================================================================
// Filter gradient of the same layers (64 -> 64 channels), fused: dw = G^T dU G with dU[xi][k][c] = sum over all 2x2 tiles of
// (A dy A^T)[xi][k] * (B^T d B)[xi][c]  (the adjoint of the forward pass; winograd.hip computes it as wino_input + wino_dout
// + a batched product). Here the two transforms feed the matrix cores straight from LDS:
//   * persistent workgroups; a work item = a 16x16-pixel block of one image: its 18x18x64 input patch and its 16x16x64 block
//     of dy live in LDS as channel-quad planes (the layout of the forward kernel, plane stride = 1 mod 16 slots so that
//     16 lanes reading the same pixel of 16 planes fall into 16 bank groups);
//   * the contraction runs over TILES: one v_mfma_f32_16x16x4_f32 consumes 4 tiles (lane = (channel quad, tile)); with a
//     float4 of 4 consecutive channels per lane on both sides, the 16 pairs (j, j') of components give the full 64 x 64
//     (k = 4 m + j, c = 4 n + j') product of one Winograd component: 16 accumulators per component;
//   * a wave owns 2 of the 16 components (row I = wave / 2, columns J0, J0 + 1 with J0 = 2 (wave & 1)) for ALL tiles: 128
//     accumulator registers that live through the whole kernel; it needs 2 patch rows x 3 columns and the 2x2 dy values per
//     tile and a handful of additions - no transformed tensor ever exists in memory;
//   * the LDS is refilled for the next item while this one is multiplied, in two halves: rows that no wave reads again
//     (three barriers per item);
//   * at the end every workgroup stores its 16 x 64 x 64 partial sums; w2g_reduce_kernel adds them in a fixed order and
//     applies G^T . G.
constexpr int GP_PLANE = 369, GD_PLANE = 257;                // slots per patch plane (360 used) / dy plane (256 used)
constexpr int GP_BYTES = 16 * GP_PLANE * 16;
constexpr int GD_BYTES = 16 * GD_PLANE * 16;
constexpr int G_LDS_BYTES = GP_BYTES + GD_BYTES;             // 160 256

__host__ __device__ constexpr int gp_start(int k) { return k < 3 ? 64 * k : k == 3 ? 136 : k == 4 ? 200 : k == 5 ? 264 : 296; }

struct W2GParams {
    const float* x;      // [N,H,W,64]
    const float* dy;     // [N,H,W,64]
    float* part;         // [grid][16][64][64] partial dU
    int N, H, W;
    int by, bx;
    int items;
    unsigned x_bytes;
};

template <int I, int J0>
__device__ __forceinline__ void w2g_run(const W2GParams& p, char* smem) {
    f32x4* P = (f32x4*)smem;
    f32x4* D = (f32x4*)(smem + GP_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int idx = lane & 15, kk = lane >> 4;       // channel quad (of c for the patch, of k for dy); tile inside a step
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.x_bytes, 0x00020000);

    // ---- LDS-DMA pieces: 64 consecutive slots of one plane per wave instruction -----------------------------------------
    // patch: plane c = 6 pieces (the last one overlaps the fifth: slots 296..359); wave w moves planes 2w, 2w+1
    // dy:    plane k = 4 pieces of 4 pixel rows;                                   wave w moves planes 2w, 2w+1
    // piece numbers 0..11 = patch (plane 2w + n / 6, piece n % 6), 12..19 = dy (plane 2w + (n - 12) / 4, piece (n - 12) % 4)
    // per-lane constants of the pieces (kept small: the compiler would otherwise keep 20 hoisted address sets in registers)
    int pcl[4];                                      // patch pieces 2j, 2j+1: row | column << 8 (column 255: padding slot)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        pcl[j] = 0;
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            const int k = 2 * j + hlf;
            const int sl = gp_start(k < 7 ? k : 6) + lane;
            const int row = sl / P_ROW, r = sl - row * P_ROW, par = r / P_PAR, col = r - par * P_PAR;
            pcl[j] |= (row | ((col < 9 ? 2 * col + par : 255) << 8)) << (16 * hlf);
        }
    }
    const int dy_row = lane >> 4, dy_col = 2 * (lane & 7) + ((lane >> 3) & 1);             // dy pieces: the lane's pixel of a 4-row piece
    // piece numbers 0..13 = patch (plane 2w + n / 7, piece n % 7), 14..21 = dy (plane 2w + (n - 14) / 4, piece (n - 14) % 4).
    // The 7 patch pieces of a plane start at slots 0, 64, 128, 136 (rows 0..9 = slots 0..199: the FIRST HALF) and 200, 264, 296
    // (rows 10..17: the second half); pieces 3 and 6 overlap their predecessors with the same values.
    auto piece = [&](int n, const W2Item& it) {
        if (n < 14) {
            const int plane = 2 * wave + n / 7, k = n % 7;
            const int e = (pcl[k >> 1] >> (16 * (k & 1))) & 0xFFFF;
            const int iy = it.oy0 - 1 + (e & 255), ix = it.ox0 - 1 + (e >> 8);
            const bool ok = (e >> 8) != 255 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const int off = (((it.n * p.H + iy) * p.W + ix) * CI + plane * 4) * 4;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(P + plane * GP_PLANE + gp_start(k)), 16, ok ? off : OOB, 0, 0, 0);
        } else {
            const int m = n - 14;
            const int plane = 2 * wave + m / 4, k = m % 4;
            const int iy = it.oy0 + 4 * k + dy_row, ix = it.ox0 + dy_col;
            const int off = (((it.n * p.H + iy) * p.W + ix) * CI + plane * 4) * 4;
            // pixels past the right / bottom edge (maps that are no multiple of 16) read as zeros: they add nothing to dU
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_ptr_t)(D + plane * GD_PLANE + 64 * k), 16,
                                                     iy < p.H && ix < p.W ? off : OOB, 0, 0, 0);
        }
    };
    // The first half of the LDS image (patch rows 0..9, dy rows 0..7) serves tile rows 0..3 and the top of tile row 4; it is
    // refilled for the next item once tile row 4 is done (barrier Z), the rest when the item is done (barrier X) - in flight
    // during the next item's tile rows 0..3 (barrier Y before tile row 4 is read).
    auto in_first_half = [](int n) { return n < 14 ? (n % 7) < 4 : ((n - 14) % 4) < 2; };

    f32x4 acc[2][4][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[a][j][k] = f32x4{0.f, 0.f, 0.f, 0.f};

    int item = blockIdx.x;
    if (item < p.items) {
        W2Item cur = w2_item(p.bx, p.by, 1, item);
#pragma unroll
        for (int n = 0; n < 22; ++n) piece(n, cur);
        W2_BARRIER(0);

        // the two patch rows / three patch columns / dy rows this wave's components need
        constexpr int RA = I == 0 ? 0 : I == 2 ? 2 : 1, RB = I == 0 ? 2 : I == 1 ? 2 : I == 2 ? 1 : 3;
        constexpr float SG = I == 1 ? 1.f : -1.f;                  // t[I] = d[RA] + SG d[RB]
        constexpr int C0 = J0 == 0 ? 0 : 1;                        // columns C0, C0 + 1, C0 + 2 of t
        const f32x4* Pl = P + idx * GP_PLANE + kk;
        const f32x4* Dl = D + idx * GD_PLANE + kk;

        // operands of one step: the lane's tile = (tile row s / 2, tile column 4 (s & 1) + kk)
        auto load = [&](int s, f32x4 (&d)[2][3], f32x4 (&g)[2][2]) {
            const int ty = s >> 1, txb = 4 * (s & 1);
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int bb = C0 + b;
                d[0][b] = Pl[(2 * ty + RA) * P_ROW + (bb & 1) * P_PAR + (bb >> 1) + txb];
                d[1][b] = Pl[(2 * ty + RB) * P_ROW + (bb & 1) * P_PAR + (bb >> 1) + txb];
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    if ((I != 0 || a == 0) && (I != 3 || a == 1)) g[a][b] = Dl[(2 * ty + a) * 16 + b * 8 + txb];
        };

        f32x4 d[2][3], g[2][2];
        load(0, d, g);
        while (true) {
            const int next = item + gridDim.x;
            const bool has_next = next < p.items;
            const W2Item nxt = w2_item(p.bx, p.by, 1, has_next ? next : item);
#pragma unroll 2
            for (int s = 0; s < 16; ++s) {
                // this step's operands from the values read one step ago
                f32x4 V[2], M[2];
                {
                    f32x4 t[3];
#pragma unroll
                    for (int b = 0; b < 3; ++b) t[b] = d[0][b] + SG * d[1][b];
                    if (J0 == 0) {
                        V[0] = t[0] - t[2];          // (B^T d B)[I][0]
                        V[1] = t[1] + t[2];          //            [I][1]
                    } else {
                        V[0] = t[1] - t[0];          //            [I][2]   (t = columns 1, 2, 3)
                        V[1] = t[0] - t[2];          //            [I][3]
                    }
                    f32x4 r[2];                      // (A dy)[I][b]
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        r[b] = I == 0 ? g[0][b] : I == 1 ? g[0][b] + g[1][b] : I == 2 ? g[0][b] - g[1][b] : -g[1][b];
                    if (J0 == 0) {
                        M[0] = r[0];                 // (A dy A^T)[I][0]
                        M[1] = r[0] + r[1];          //            [I][1]
                    } else {
                        M[0] = r[0] - r[1];          //            [I][2]
                        M[1] = -r[1];                //            [I][3]
                    }
                }
                // X (step 15): every wave has consumed its last values of this item, and the first-half refill (steps 10..12)
                // has landed: the next item's first tile can be read, its second half refilled.
                // Y (step 7): the second-half refill (steps 0..3) has landed before tile row 4 is read.
                if (s == 15 || s == 7) W2_BARRIER(0);
                // the next step's values, read one step ahead (past the last step: the next item's first tile)
                load((s + 1) & 15, d, g);
                // refills, a few pieces per step: the second half of THIS item's LDS image during its steps 0..3 (the first item
                // came complete), the first half of the NEXT item's during steps 10..12
                if (s < 4 && item != (int)blockIdx.x) {
                    int ord = 0;
#pragma unroll
                    for (int n = 0; n < 22; ++n)
                        if (!in_first_half(n)) {                         // 10 per wave: 3, 3, 2, 2
                            if ((ord < 6 ? ord / 3 : 2 + (ord - 6) / 2) == s) piece(n, cur);
                            ++ord;
                        }
                }
                if (has_next && s >= 10 && s < 13) {
                    int ord = 0;
#pragma unroll
                    for (int n = 0; n < 22; ++n)
                        if (in_first_half(n)) {                          // 12 per wave: 4, 4, 4
                            if (ord / 4 == s - 10) piece(n, nxt);
                            ++ord;
                        }
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            acc[a][j][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(M[a][j], V[a][k], acc[a][j][k], 0, 0, 0);
                if (s == 9) W2_BARRIER(0);           // Z: tile rows 0..4 are done everywhere: the first half may be refilled
            }
            if (!has_next) break;
            item = next;
            cur = nxt;
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    // partial sums: component xi = 4 I + J0 + a; lane (n = idx, q = kk) holds k = 16 q + 4 r + j, c = 4 n + j'
    float* out = p.part + (long)blockIdx.x * 16 * 4096;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 16 * kk + 4 * r + j;
                const f32x4 v = {acc[a][j][0][r], acc[a][j][1][r], acc[a][j][2][r], acc[a][j][3][r]};
                *(f32x4*)(out + (4 * I + J0 + a) * 4096 + k * 64 + 4 * idx) = v;
            }
}

__global__ __launch_bounds__(512, 2) void wino2f_wgrad_kernel(const W2GParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {
        case 0: w2g_run<0, 0>(p, smem); break;
        case 1: w2g_run<0, 2>(p, smem); break;
        case 2: w2g_run<1, 0>(p, smem); break;
        case 3: w2g_run<1, 2>(p, smem); break;
        case 4: w2g_run<2, 0>(p, smem); break;
        case 5: w2g_run<2, 2>(p, smem); break;
        case 6: w2g_run<3, 0>(p, smem); break;
        default: w2g_run<3, 2>(p, smem); break;
    }
}

// dU[xi][k][c] = sum over the workgroups' partial sums, in workgroup order (deterministic); block = (xi, k), 64 c x 4 slices
__global__ __launch_bounds__(256) void w2g_reduce_kernel(const float* __restrict__ part, int parts, float* __restrict__ dU) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long o = (long)blockIdx.x * 64 + c;            // (xi * 64 + k) * 64 + c
    const int per = (parts + 3) / 4;
    float a = 0.f;
    for (int w = sl * per; w < parts && w < (sl + 1) * per; ++w) a += part[(long)w * 16 * 4096 + o];
    red[sl][c] = a;
    __syncthreads();
    if (sl == 0) dU[o] = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
}

// dw[k][r][s][c] = (G^T dU G)[r][s]: the adjoint of the filter transform (G of F(2x2,3x3))
__global__ __launch_bounds__(256) void w2g_dfilter_kernel(const float* __restrict__ dU, float* __restrict__ dw) {
    const int t = blockIdx.x * 256 + threadIdx.x;        // k * 64 + c
    const int c = t & 63, k = t >> 6;
    float u[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) u[i][j] = dU[(4 * i + j) * 4096 + t];
    float r[3][4];                                       // G^T u: rows (u0 + (u1 + u2) / 2, (u1 - u2) / 2, (u1 + u2) / 2 + u3)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[0][j] = u[0][j] + 0.5f * u[1][j] + 0.5f * u[2][j];
        r[1][j] = 0.5f * u[1][j] - 0.5f * u[2][j];
        r[2][j] = 0.5f * u[1][j] + 0.5f * u[2][j] + u[3][j];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        dw[((k * 3 + i) * 3 + 0) * 64 + c] = r[i][0] + 0.5f * r[i][1] + 0.5f * r[i][2];
        dw[((k * 3 + i) * 3 + 1) * 64 + c] = 0.5f * r[i][1] - 0.5f * r[i][2];
        dw[((k * 3 + i) * 3 + 2) * 64 + c] = 0.5f * r[i][1] + 0.5f * r[i][2] + r[i][3];
    }
}

}  // namespace

// geometry this kernel covers
extern "C" int denet_conv_wino2f_ok(int N, int H, int W, int Ci, int Co) {
    return (Ci == 64 && Co > 0 && Co % 64 == 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && N > 0 &&
            (long)N * H * W * 64 * 4 < 0xF0000000L && (long)N * H * W * Co * 4 < 0x7FFFFFFFL) ? 1 : 0;
}

// y = conv3x3(x) stride 1 pad 1 (+ bias) (+ add) from the F(2x2) transformed filters u = [16][Co][64]
// (denet_conv_wino_filter with tile 2: dgrad = 0 for the forward pass, 1 for the data gradient, where x = dy, Co = C).
// stats_partial (optional): [rows][2][Co] doubles, the batch-norm column sums of y (see denet_conv_fwd_stats); rows = the launch's
// workgroups for Co = 64, N*ceil(H/16)*ceil(W/16) otherwise (the most the buffer must hold); *stats_rows receives rows.
extern "C" int denet_conv_wino2f_sums(const float* x, const float* u, const float* bias, const float* add, float* y, int relu,
                                      double* stats_partial, size_t stats_bytes, int* stats_rows, const denet_bn_link* sums_of,
                                      int N, int H, int W, int Ci, int Co, hipStream_t stream);

extern "C" int denet_conv_wino2f(const float* x, const float* u, const float* bias, const float* add, float* y, int relu,
                                 double* stats_partial, size_t stats_bytes, int* stats_rows, int N, int H, int W, int Ci,
                                 int Co, hipStream_t stream) {
    return denet_conv_wino2f_sums(x, u, bias, add, y, relu, stats_partial, stats_bytes, stats_rows, nullptr, N, H, W, Ci, Co, stream);
}

// the same; sums_of != NULL (a data-gradient call): the tensor written is the gradient of the OUTPUT of the batch-norm layer
// sums_of describes (x = its input, y = its forward output or NULL, gamma / beta / mean / invstd, relu) and stats_partial
// receives that layer's backward reductions per block: [rows][2][Co] doubles = sum(g), sum(g * xhat) - the input of
// denet_bn_bwd_final instead of a pass of its own over three tensors (bn_bwd_partial_kernel)
extern "C" int denet_conv_wino2f_sums(const float* x, const float* u, const float* bias, const float* add, float* y, int relu,
                                      double* stats_partial, size_t stats_bytes, int* stats_rows, const denet_bn_link* sums_of,
                                      int N, int H, int W, int Ci, int Co, hipStream_t stream) {
    DENET_CHECK_ARG(x && u && y, "conv_wino2f: null pointer");
    DENET_CHECK_ARG(denet_conv_wino2f_ok(N, H, W, Ci, Co), "conv_wino2f: needs Ci = 64, Co %% 64 = 0, even H and W");
    W2Params p = {};
    p.x = x; p.U = u; p.bias = bias; p.add = add; p.y = y;
    p.N = N; p.H = H; p.W = W; p.Co = Co;
    p.by = (H + 15) / 16; p.bx = (W + 15) / 16;
    p.nco = Co / 64;
    p.relu = relu;
    p.x_bytes = (unsigned)((size_t)N * H * W * 64 * 4);
    p.y_bytes = (unsigned)((size_t)N * H * W * Co * 4);
    const long blocks = (long)N * p.by * p.bx;
    p.items = (int)(blocks * p.nco);
    // the LDS footprint allows one workgroup per CU: a persistent grid, work items strided over it
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            denet_set_error("conv_wino2f: cannot query the device");
            return DENET_ERR_ARG;
        }
        cus = prop.multiProcessorCount;
    }
    const int grid = p.items < cus ? p.items : cus;
    if (stats_partial) {
        const long rows = p.nco == 1 ? grid : blocks;
        DENET_CHECK_ARG(stats_rows && stats_bytes >= (size_t)rows * 2 * Co * sizeof(double), "conv_wino2f: statistics buffer too small");
        *stats_rows = (int)rows;
        p.stats = stats_partial;
        if (p.nco == 1) p.fin = denet_bn_final_take(sums_of ? 2 : 1, Co, 1);     // (armed by the caller: bn_final.h)
        if (sums_of) {
            DENET_CHECK_ARG(sums_of->x && sums_of->mean && sums_of->invstd && (!sums_of->relu || sums_of->y || (sums_of->gamma && sums_of->beta)),
                            "conv_wino2f: incomplete batch-norm description for the backward sums");
            p.bs_x = sums_of->x; p.bs_y = sums_of->relu ? sums_of->y : nullptr; p.bs_gamma = sums_of->gamma; p.bs_beta = sums_of->beta;
            p.bs_mean = sums_of->mean; p.bs_invstd = sums_of->invstd; p.bs_relu = sums_of->relu;
        }
    }
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)wino2f_ws_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)wino2f_ws_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES);
        if (e != hipSuccess) {
            denet_set_error("conv_wino2f: hipFuncSetAttribute(%d B LDS): %s", S_LDS_BYTES, hipGetErrorString(e));
            return -(int)e;
        }
        attr_set = true;
    }
    const int prof = denet_prof_begin(10, (p.add || p.bs_x) ? 1 : 0, 0, 0, stream);      // which instantiation runs
    // the variant that prefetches the output phase's operands only where there are any (it costs the plain pass 4 %)
    if (p.add || p.bs_x) hipLaunchKernelGGL(wino2f_ws_kernel<true>, dim3((unsigned)grid), dim3(512), S_LDS_BYTES, stream, p);
    else hipLaunchKernelGGL(wino2f_ws_kernel<false>, dim3((unsigned)grid), dim3(512), S_LDS_BYTES, stream, p);
    denet_prof_end(prof, stream);
    DENET_CHECK_LAUNCH("conv_wino2f");
    return DENET_OK;
}

// geometry the fused filter-gradient kernel covers
extern "C" int denet_conv_wino2f_wgrad_ok(int N, int H, int W, int C, int K) {
    return (C == 64 && K == 64 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && N > 0 &&
            (long)N * H * W * 64 * 4 < 0xF0000000L) ? 1 : 0;
}

static int w2g_grid(int N, int H, int W) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
        cus = prop.multiProcessorCount;
    }
    const long items = (long)N * ((H + 15) / 16) * ((W + 15) / 16);
    return (int)(items < cus ? items : cus);
}

extern "C" size_t denet_conv_wino2f_wgrad_workspace_bytes(int N, int H, int W) {
    const int grid = w2g_grid(N, H, W);
    return grid < 0 ? 0 : ((size_t)grid + 1) * 16 * 4096 * sizeof(float);
}

// dw[64][3][3][64] = the filter gradient of a 3x3 stride-1 pad-1 convolution x [N,H,W,64] -> y [N,H,W,64] for dy
// (denet/model/model_cnn.py:318, tensor.grad of convolution.py:80-83), F(2x2,3x3) with the transforms and the products fused
extern "C" int denet_conv_wino2f_wgrad(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes, int N,
                                       int H, int W, int C, int K, hipStream_t stream) {
    DENET_CHECK_ARG(x && dy && dw && workspace, "conv_wino2f_wgrad: null pointer");
    DENET_CHECK_ARG(denet_conv_wino2f_wgrad_ok(N, H, W, C, K), "conv_wino2f_wgrad: needs C = K = 64, even H and W");
    const int grid = w2g_grid(N, H, W);
    DENET_CHECK_ARG(grid > 0, "conv_wino2f_wgrad: cannot query the device");
    DENET_CHECK_ARG(workspace_bytes >= denet_conv_wino2f_wgrad_workspace_bytes(N, H, W), "conv_wino2f_wgrad: workspace too small");
    W2GParams p = {};
    p.x = x; p.dy = dy; p.part = (float*)workspace;
    p.N = N; p.H = H; p.W = W;
    p.by = (H + 15) / 16; p.bx = (W + 15) / 16;
    p.items = N * p.by * p.bx;
    p.x_bytes = (unsigned)((size_t)N * H * W * 64 * 4);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)wino2f_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS_BYTES);
        if (e != hipSuccess) {
            denet_set_error("conv_wino2f_wgrad: hipFuncSetAttribute(%d B LDS): %s", G_LDS_BYTES, hipGetErrorString(e));
            return -(int)e;
        }
        attr_set = true;
    }
    float* dU = p.part + (size_t)grid * 16 * 4096;
    const int prof = denet_prof_begin(11, 0, 0, 0, stream);
    hipLaunchKernelGGL(wino2f_wgrad_kernel, dim3(grid), dim3(512), G_LDS_BYTES, stream, p);
    denet_prof_end(prof, stream);
    hipLaunchKernelGGL(w2g_reduce_kernel, dim3(16 * 64), dim3(256), 0, stream, p.part, grid, dU);
    hipLaunchKernelGGL(w2g_dfilter_kernel, dim3(16), dim3(256), 0, stream, dU, dw);
    DENET_CHECK_LAUNCH("conv_wino2f_wgrad");
    return DENET_OK;
}
