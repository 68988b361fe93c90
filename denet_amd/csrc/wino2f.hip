// Fused Winograd F(2x2,3x3) convolution for the 64-input-channel 3x3 stride-1 layers (the first ResNet stage:
// denet/layer/convolution.py:80-83 forward; its data gradient, model_cnn.py:318, is the same operation on dy with the
// rotated, channel-swapped filters): input transform, the 16 component products and the output transform in ONE kernel.
// Nothing but x (or dy) is read and y (or dx) written - the un-fused passes move the 4x (F2) / 2.25x (F4) expanded V and M
// tensors through HBM, which is what bounds them at 64 channels (winograd.hip: 0.33 ms per pass; the direct kernel 0.34).
//
// Workgroup = 8 waves, one 16x16-pixel output block (8x8 tiles of 2x2) x 64 output channels, all 64 input channels:
//   * the 18x18-pixel input patch lives in LDS for the whole workgroup (channel-quad planes, even / odd columns apart: the
//     16-byte reads of the 16 tiles of a wave fall into 16 different bank groups);
//   * the transformed filters U[xi][co][ci] stream through LDS in 16-channel groups, as two halves (components 0-7, 8-15)
//     filled by LDS-DMA (global_load_lds: no staging registers) while the other half is being multiplied;
//   * v_mfma_f32_16x16x4_f32 with the filter as the row operand: a lane ends up with ALL 16 components of its tile for 4
//     consecutive output channels, so the output transform A^T M A is register-local and the result is stored 16 bytes
//     at a time; a wave = 16 tiles x 32 output channels x 16 components = 128 accumulator registers;
//   * the epilogue adds bias / the accumulated gradient and can emit the batch-norm column sums (batch_norm.py:50-53).
// Exact fp32 FMA chains; the association differs from the direct kernel (F(2x2): ~1e-6 relative).
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

struct W2Params {
    const float* x;      // [N,H,W,64]
    const float* U;      // [16][Co][64] transformed filters (denet_conv_wino_filter, tile 2)
    const float* bias;   // [Co] or null
    const float* add;    // [N,H,W,Co] or null
    float* y;            // [N,H,W,Co]
    double* stats;       // [blocks][2][Co] or null
    int N, H, W, Co;
    int by, bx;          // 16x16 output blocks per image
    int nco;             // Co / 64
    int items;           // N * by * bx * nco work items: (block, 64 output channels)
    int relu;            // y = max(y, 0) (the inference fold of a ReLU layer)
    unsigned x_bytes;
};

constexpr int CI = 64;
constexpr int P_ROW = 20, P_PAR = 10, P_PLANE = 368;        // 16-byte slots: row / parity / plane strides of the patch
constexpr int P_USED = 18 * P_ROW;                          // slots of a plane that hold pixels (or in-row padding)
constexpr int P_BYTES = 16 * P_PLANE * 16;                  // 16 channel-quad planes
constexpr int UH_SLOTS = 8 * 4 * 64;                        // one half of a filter group: [8 xi][4 q][64 co] float4
constexpr int RED_BYTES = 4 * 2 * 64 * 4;                   // batch-norm sums of the 4 tile-row waves
constexpr int LDS_BYTES = P_BYTES + 2 * UH_SLOTS * 16 + RED_BYTES;      // 161 792 of 163 840
constexpr int OOB = (int)0xF0000000u;
#ifdef W2F_TRACE
constexpr int LDS_ALLOC = LDS_BYTES + 1024;
#else
constexpr int LDS_ALLOC = LDS_BYTES;
#endif

// s_waitcnt vmcnt(vm) lgkmcnt(0) [gfx9 encoding: vmcnt = bits 15:14 | 3:0, expcnt 6:4, lgkmcnt 11:8] + s_barrier. The raw
// barrier leaves the newest `vm` vector-memory operations of the wave in flight (LDS-DMA pieces that are not needed yet,
// result stores): __syncthreads() would drain them all.
#define W2_BARRIER(vm)                                                                           \
    {                                                                                            \
        __builtin_amdgcn_s_waitcnt(((vm) & 15) | ((((vm) >> 4) & 3) << 14) | (7 << 4));          \
        __builtin_amdgcn_s_barrier();                                                            \
    }

#ifdef W2F_TRACE
__device__ unsigned g_w2f_trace[8 * 32];
#define W2_MARK(idx)                                                      \
    if (trace_on) {                                                       \
        const unsigned t_ = (unsigned)__builtin_readcyclecounter();       \
        if (lane == 0) trc[wave * 32 + (idx)] = t_;                       \
    }
#else
#define W2_MARK(idx)
#endif

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

// Persistent workgroups: 8 waves, one work item = a 16x16-pixel output block (8x8 tiles of 2x2) x 64 output channels after
// the other. While item i is being multiplied the input patch and the first filter group of item i + 1 stream into LDS.
__global__ __launch_bounds__(512, 2) void wino2f_kernel(const W2Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* P = (f32x4*)smem;
    f32x4* UA = (f32x4*)(smem + P_BYTES);
    f32x4* UB = UA + UH_SLOTS;
    float* red = (float*)(smem + P_BYTES + 2 * UH_SLOTS * 16);
#ifdef W2F_TRACE
    unsigned* trc = (unsigned*)(smem + LDS_BYTES);
    int iter = 0;
    bool trace_on = false;
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform: addresses derived from it stay scalar
    const int t = lane & 15, q = lane >> 4;          // tile inside the wave's 16, channel quad inside a group of 16
    const int tg = wave & 3, nh = wave >> 2;         // tile rows 2tg, 2tg+1 of the block; output-channel half
    const int ty = 2 * tg + (t >> 3), tx = t & 7;
    const int ucol = 32 * nh + t;                    // + 16 nb: the lane's row of the filter operand
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);

    // ---- LDS-DMA pieces (one wave instruction = 64 lanes x 16 B into 64 consecutive slots) -------------------------------
    // filters: a half = 32 rows (xi8, q) of 64 output channels; wave w moves rows 4w .. 4w+3, piece k = row 4w + k
    auto u_piece = [&](f32x4* dst, int h, int g, int co0, int k) {
        const int r = wave * 4 + k;
        const int xi8 = r >> 2, qq = r & 3;
        const float* src = p.U + ((long)(8 * h + xi8) * p.Co + co0 + lane) * CI + 16 * g + 4 * qq;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(dst + (xi8 * 4 + qq) * 64), 16, 0, 0);
    };
    // input patch: plane (channel quad) c of the 18 x 18 pixels = 6 pieces of 64 slots; the 24 pieces of channel group g
    // are spread over the waves: wave w moves plane 4g + w/2, pieces 3 (w & 1) + j, j = 0..2. A lane's slot -> (row, column)
    // does not depend on the item; image borders and the padding slots come back as zeros from the buffer bounds check.
    // Piece 5 of a plane starts at slot 296 = 360 - 64 (it rewrites 24 slots of piece 4 with the same values): every piece is
    // a full wave instruction, no lane mask, no branch.
    int pc[3];                                       // row | column << 8; column 255: not a pixel
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int k = 3 * (wave & 1) + j;
        const int sl = (k == 5 ? P_USED - 64 : 64 * k) + lane;
        const int row = sl / P_ROW, r = sl - row * P_ROW, par = r / P_PAR, col = r - par * P_PAR;
        pc[j] = row | ((col < 9 ? 2 * col + par : 255) << 8);
    }
    auto patch_piece = [&](int g, int j, const W2Item& it) {
        const int plane = 4 * g + (wave >> 1);
        const int k = 3 * (wave & 1) + j;
        const int iy = it.oy0 - 1 + (pc[j] & 255), ix = it.ox0 - 1 + (pc[j] >> 8);
        const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && (pc[j] >> 8) != 255;
        const int off = (((it.n * p.H + iy) * p.W + ix) * CI + plane * 4) * 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(P + plane * P_PLANE + (k == 5 ? P_USED - 64 : 64 * k)), 16,
                                                 ok ? off : OOB, 0, 0, 0);
    };

    // t = B^T d, row i of the lane's tile for its 4 channels of group g: a combination of two patch rows
    //   row 0 = d0 - d2, row 1 = d1 + d2, row 2 = d2 - d1, row 3 = d1 - d3
    auto load_tt = [&](f32x4 (&tt)[4], int g, int i) {
        const int ra = i == 0 ? 0 : 1, rb = i == 3 ? 3 : 2;
        const f32x4* Pp = P + (4 * g + q) * P_PLANE + (2 * ty) * P_ROW + tx;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const f32x4* c = Pp + (bb & 1) * P_PAR + (bb >> 1);
            const f32x4 ea = c[ra * P_ROW], eb = c[rb * P_ROW];
            tt[bb] = i == 0 ? ea - eb : i == 1 ? ea + eb : i == 2 ? eb - ea : ea - eb;
        }
    };

    int item = blockIdx.x;
    if (item >= p.items) return;
#ifdef W2F_TRACE
    trace_on = blockIdx.x == 0;
    W2_MARK(26);
#endif
    W2Item cur = w2_item(p.bx, p.by, p.nco, item);
    // prologue: channel groups 0..2 of the first patch (group 3 comes with the first half, like every later one) and the
    // first filter half
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int j = 0; j < 3; ++j) patch_piece(g, j, cur);
#pragma unroll
    for (int k = 0; k < 4; ++k) u_piece(UA, 0, 0, cur.co0, k);
    W2_BARRIER(0);

    // Software pipeline over the halves h = (item, group g, A | B); half h multiplies components 8 (h & 1) .. + 7 of group g
    // out of buffer UA (A) / UB (B) in 8 steps of 8 products:
    //   steps 0-1  the 4 DMA pieces of the NEXT half's filters go into the other buffer
    //   steps 2-4  (A halves) 3 pieces refill the patch planes of the previous channel group for the next item
    //   step  5    the filter operands of steps 6-7 are read into registers: no wave reads this buffer after the barrier
    //   step  6    wait for the own pieces, ONE barrier: the other buffer is complete, this one may be refilled next half
    //   steps 6-7  the first row of t = B^T d and the first filter operands of the next half are read: it starts without
    //              an LDS round trip (its second row is read during its first steps)
    f32x4 tt[4], uc0, uc1;                           // first row + operands: carried from half to half (across the epilogue)
    load_tt(tt, 0, 0);
    uc0 = UA[q * 64 + ucol];
    uc1 = UA[q * 64 + ucol + 16];

    while (true) {
        const int next = item + gridDim.x;
        const bool has_next = next < p.items;
        const W2Item nxt = w2_item(p.bx, p.by, p.nco, has_next ? next : item);
#ifdef W2F_TRACE
        trace_on = blockIdx.x == 0 && iter == 2;
        ++iter;
#endif
        W2_MARK(0);

        f32x4 acc[16][2];                            // started by the first product of channel group 0 (C operand 0)

        auto half = [&](auto G, auto HB) {
            constexpr int g = decltype(G)::value;
            constexpr int hb = decltype(HB)::value;              // 0: half A, 1: half B
            constexpr int i0 = 2 * hb;
            const f32x4* up = (hb ? UB : UA) + q * 64 + ucol;
            f32x4* const other = hb ? UA : UB;
            // what the next half is. The pipeline has no branches: past the last item the pieces and reads repeat this item's
            // (`nxt` = `cur`), into buffers that nobody reads again.
            constexpr int ng = hb ? (g + 1) & 3 : g;
            const int nco0 = (hb && g == 3) ? nxt.co0 : cur.co0;
            f32x4 U0[8], U1[8], t2[4], tn[4], un0, un1;  // filter operands per step: read one step ahead
            U0[0] = uc0;
            U1[0] = uc1;
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = i0 + ii;
                const f32x4 (&tr)[4] = ii ? t2 : tt;
                f32x4 V[4];
                V[0] = tr[0] - tr[2];
                V[1] = tr[1] + tr[2];
                V[2] = tr[2] - tr[1];
                V[3] = tr[1] - tr[3];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int st = ii * 4 + j;
#ifdef W2F_PRIO
                    // the two waves of a SIMD (nh = 0, 1) take turns at the higher issue priority, step by step: the older one
                    // would otherwise run ahead and leave the other to finish the half alone
                    if ((st + nh) & 1) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
#endif
                    if (st < 5) {
                        U0[st + 1] = up[(st + 1) * 256];
                        U1[st + 1] = up[(st + 1) * 256 + 16];
                    } else if (st == 5) {            // the last reads of this buffer, before the barrier of step 6
                        U0[6] = up[6 * 256];
                        U1[6] = up[6 * 256 + 16];
                        U0[7] = up[7 * 256];
                        U1[7] = up[7 * 256 + 16];
                    }
                    if (st == 1) load_tt(t2, g, i0 + 1);         // the second row of this half, used from step 4 on
                    const f32x4 u0 = U0[st], u1 = U1[st];
                    if (st < 2) {
                        u_piece(other, hb ^ 1, ng, nco0, 2 * st);
                        u_piece(other, hb ^ 1, ng, nco0, 2 * st + 1);
                    } else if (st < 5 && !hb) {
                        // patch refill in A halves: group g - 1 for the next item; g = 0: group 3 for THIS item (its planes
                        // were last read in the previous item's half B of group 3)
                        patch_piece((g + 3) & 3, st - 2, g == 0 ? cur : nxt);
                    }
                    if (st == 6) {
                        W2_MARK(1 + (2 * g + hb) * 3);
                        if (!hb) W2_BARRIER(3)       // the 3 patch pieces, issued after the filter pieces, may still fly
                        else W2_BARRIER(0)
                        W2_MARK(2 + (2 * g + hb) * 3);
                        load_tt(tn, ng, hb ? 0 : 2); // the first row of the next half
                        un0 = other[q * 64 + ucol];
                        un1 = other[q * 64 + ucol + 16];
                    }
                    // the two accumulators alternate: a dependent v_mfma_f32_16x16x4_f32 issues after 40 cycles, an
                    // independent one after 32
                    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const bool start = g == 0 && c == 0;
                        acc[4 * i + j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u0[c], V[j][c], start ? zero : acc[4 * i + j][0], 0, 0, 0);
                        acc[4 * i + j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u1[c], V[j][c], start ? zero : acc[4 * i + j][1], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);           // the steps stay in this order
                }
            }
            W2_MARK(3 + (2 * g + hb) * 3);
#pragma unroll
            for (int b = 0; b < 4; ++b) tt[b] = tn[b];
            uc0 = un0;
            uc1 = un1;
        };
        half(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        half(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        half(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
        half(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        half(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
        half(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});
        half(std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{});
        half(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{});

        // ---- epilogue: Y = A^T M A per (tile, 4 output channels), + bias, + add; 16-byte stores -------------------------
        const long pix = ((long)cur.n * p.H + cur.oy0 + 2 * ty) * p.W + cur.ox0 + 2 * tx;
        f32x4 ssum[2], ssq[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int ch = cur.co0 + 32 * nh + 16 * nb + 4 * q;
            f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) bias4 = *(const f32x4*)(p.bias + ch);
            f32x4 addv[4];
            if (p.add) {
#pragma unroll
                for (int ij = 0; ij < 4; ++ij) addv[ij] = *(const f32x4*)(p.add + (pix + (ij >> 1) * p.W + (ij & 1)) * p.Co + ch);
            }
            f32x4 s[2][4];
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                s[0][bb] = acc[0 + bb][nb] + acc[4 + bb][nb] + acc[8 + bb][nb];
                s[1][bb] = acc[4 + bb][nb] - acc[8 + bb][nb] - acc[12 + bb][nb];
            }
            ssum[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
            ssq[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ij = 0; ij < 4; ++ij) {
                const int i = ij >> 1;
                f32x4 yv = (ij & 1) ? (s[i][1] - s[i][2] - s[i][3]) : (s[i][0] + s[i][1] + s[i][2]);
                yv += bias4;
                if (p.add) yv += addv[ij];
                if (p.relu) yv = __builtin_elementwise_max(yv, f32x4{0.f, 0.f, 0.f, 0.f});
                *(f32x4*)(p.y + (pix + i * p.W + (ij & 1)) * p.Co + ch) = yv;
                ssum[nb] += yv;
                ssq[nb] += yv * yv;
            }
        }
        if (p.stats) {
            // batch-norm column sums of this block: over the 16 tiles of a wave (lanes t), then over the 4 tile-row waves
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
                for (int off = 8; off > 0; off >>= 1)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        ssum[nb][c] += __shfl_xor(ssum[nb][c], off, 64);
                        ssq[nb][c] += __shfl_xor(ssq[nb][c], off, 64);
                    }
                if (t == 0) {
                    const int cl = 32 * nh + 16 * nb + 4 * q;
                    *(f32x4*)(red + (tg * 2 + 0) * 64 + cl) = ssum[nb];
                    *(f32x4*)(red + (tg * 2 + 1) * 64 + cl) = ssq[nb];
                }
            }
            W2_BARRIER(63);                          // LDS only: nothing in flight is waited for
            if (tid < 64) {
                double a = 0.0, bq = 0.0;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    a += (double)red[(w * 2 + 0) * 64 + tid];
                    bq += (double)red[(w * 2 + 1) * 64 + tid];
                }
                double* ps = p.stats + (long)cur.block * 2 * p.Co;
                ps[cur.co0 + tid] = a;
                ps[p.Co + cur.co0 + tid] = bq;
            }
        }
        W2_MARK(25);
        if (!has_next) break;
        item = next;
        cur = nxt;
    }
    __builtin_amdgcn_s_waitcnt(0);                   // no LDS-DMA piece may land after the workgroup has released its LDS
#ifdef W2F_TRACE
    trace_on = blockIdx.x == 0;
    W2_MARK(27);
    __syncthreads();
    if (blockIdx.x == 0 && tid < 256) g_w2f_trace[tid] = trc[tid];
#endif
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
    const int dyl = ((lane >> 4) * p.W + 2 * (lane & 7) + ((lane >> 3) & 1)) * CI * 4;   // dy pieces: lane part of the offset
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
            const int base = (((it.n * p.H + it.oy0 + 4 * k) * p.W + it.ox0) * CI + plane * 4) * 4;      // wave-uniform
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_ptr_t)(D + plane * GD_PLANE + 64 * k), 16, dyl, base, 0, 0);
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
#pragma unroll 1
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

#ifdef W2F_TRACE
extern "C" int denet_w2f_trace(unsigned* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_w2f_trace), sizeof(unsigned) * 256); }
#endif

// geometry this kernel covers
extern "C" int denet_conv_wino2f_ok(int N, int H, int W, int Ci, int Co) {
    return (Ci == 64 && Co > 0 && Co % 64 == 0 && H > 0 && W > 0 && H % 16 == 0 && W % 16 == 0 && N > 0 &&
            (long)N * H * W * 64 * 4 < 0xF0000000L) ? 1 : 0;
}

// y = conv3x3(x) stride 1 pad 1 (+ bias) (+ add) from the F(2x2) transformed filters u = [16][Co][64]
// (denet_conv_wino_filter with tile 2: dgrad = 0 for the forward pass, 1 for the data gradient, where x = dy, Co = C).
// stats_partial (optional): [N*(H/16)*(W/16)][2][Co] doubles, the batch-norm column sums of y (see denet_conv_fwd_stats).
extern "C" int denet_conv_wino2f(const float* x, const float* u, const float* bias, const float* add, float* y, int relu,
                                 double* stats_partial, size_t stats_bytes, int* stats_rows, int N, int H, int W, int Ci,
                                 int Co, hipStream_t stream) {
    DENET_CHECK_ARG(x && u && y, "conv_wino2f: null pointer");
    DENET_CHECK_ARG(denet_conv_wino2f_ok(N, H, W, Ci, Co), "conv_wino2f: needs Ci = 64, Co %% 64 = 0, H, W multiples of 16");
    W2Params p = {};
    p.x = x; p.U = u; p.bias = bias; p.add = add; p.y = y;
    p.N = N; p.H = H; p.W = W; p.Co = Co;
    p.by = H / 16; p.bx = W / 16;
    p.nco = Co / 64;
    p.relu = relu;
    p.x_bytes = (unsigned)((size_t)N * H * W * 64 * 4);
    const long blocks = (long)N * p.by * p.bx;
    p.items = (int)(blocks * p.nco);
    if (stats_partial) {
        DENET_CHECK_ARG(stats_rows && stats_bytes >= (size_t)blocks * 2 * Co * sizeof(double), "conv_wino2f: statistics buffer too small");
        *stats_rows = (int)blocks;
        p.stats = stats_partial;
    }
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)wino2f_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_ALLOC);
        if (e != hipSuccess) {
            denet_set_error("conv_wino2f: hipFuncSetAttribute(%d B LDS): %s", LDS_BYTES, hipGetErrorString(e));
            return -(int)e;
        }
        attr_set = true;
    }
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
    hipLaunchKernelGGL(wino2f_kernel, dim3((unsigned)grid), dim3(512), LDS_ALLOC, stream, p);
    DENET_CHECK_LAUNCH("conv_wino2f");
    return DENET_OK;
}

// geometry the fused filter-gradient kernel covers
extern "C" int denet_conv_wino2f_wgrad_ok(int N, int H, int W, int C, int K) {
    return (C == 64 && K == 64 && H > 0 && W > 0 && H % 16 == 0 && W % 16 == 0 && N > 0 &&
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
    const long items = (long)N * (H / 16) * (W / 16);
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
    DENET_CHECK_ARG(denet_conv_wino2f_wgrad_ok(N, H, W, C, K), "conv_wino2f_wgrad: needs C = K = 64, H, W multiples of 16");
    const int grid = w2g_grid(N, H, W);
    DENET_CHECK_ARG(grid > 0, "conv_wino2f_wgrad: cannot query the device");
    DENET_CHECK_ARG(workspace_bytes >= denet_conv_wino2f_wgrad_workspace_bytes(N, H, W), "conv_wino2f_wgrad: workspace too small");
    W2GParams p = {};
    p.x = x; p.dy = dy; p.part = (float*)workspace;
    p.N = N; p.H = H; p.W = W;
    p.by = H / 16; p.bx = W / 16;
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
    hipLaunchKernelGGL(wino2f_wgrad_kernel, dim3(grid), dim3(512), G_LDS_BYTES, stream, p);
    hipLaunchKernelGGL(w2g_reduce_kernel, dim3(16 * 64), dim3(256), 0, stream, p.part, grid, dU);
    hipLaunchKernelGGL(w2g_dfilter_kernel, dim3(16), dim3(256), 0, stream, dU, dw);
    DENET_CHECK_LAUNCH("conv_wino2f_wgrad");
    return DENET_OK;
}
