// Winograd F(4x4,3x3), TILE-PARALLEL: input transform, the 36 component products and the output transform in ONE kernel - neither
// V nor M ever reaches HBM (verdict item of rounds 4-5: "a tile-parallel fused F(4x4) kernel with the input transform inside").
// Reference op: the `C[k,3]` layer, denet/layer/convolution.py:80-83 (forward) and its data gradient, model_cnn.py:318 (the same
// pipeline on dy with the rotated, channel-swapped filters) - here for the layers whose x is a plain tensor (the 64-channel stage,
// where the fused F(2x2) kernels of wino2f.hip execute 1.78 times the products of this one).
//
// The component-walk kernel (wino4f.hip) keeps 16 output positions per value and reads the 2.25x expanded V from HBM. This one
// turns the loop nest round: a workgroup of FOUR waves owns 16 tiles (2 x 8 tiles = 8 x 32 output pixels) x 64 output channels
// and keeps ALL 36 components of its products as accumulators (a wave: 16 tiles x 16 channels x 36 components = 144 registers per
// lane); the reduction runs over chunks of 16 input channels:
//     patch chunk (10 x 34 pixels x 16 channels, 24 KB)  --LDS-DMA-->  LDS (one buffer; the next chunk's pieces are issued in the
//                 middle of this chunk's products, behind the transform that read the buffer)
//     transform:  every thread forms half the components of (one tile, two channels): 30 x ds_read_b64, 72 packed fp32
//                 operations, 18 x ds_write_b64 -> V[36][4 channel quads][16 tiles][4] in LDS (36 KB)
//     products:   per component one ds_read_b128 (V fragment), one buffer_load_dwordx4 (U fragment, straight from L2: the
//                 packed filters [C/16][36][K][16] make a wave's fragment one contiguous KB) and four v_mfma_f32_16x16x4_f32
// and the epilogue is the output transform in registers (the lane holds the 6 x 6 components of its (tile, 4 channels)) followed
// by the stores of wino4f.hip's epilogue (bias / add / ReLU, batch-norm column sums, backward sums).
// TWO workgroups share a CU (61 KB of LDS, 256 registers per lane each): while one transforms, waits for its patch or streams its
// epilogue, the other multiplies. (Measured on the way, tools/exp/w4t_check.py with the -DT_EXP ablation builds: ONE 8-wave
// workgroup per CU on 32 tiles ran 64 -> 64 channels on a 128x128 map at batch 32 in 170 us = products 89 + epilogue 45 + transform
// 24 + prologue, nothing overlapping - every CU reaches its epilogue at the same time, and the 33 MB store burst holds the next
// item's fragment loads back; a persistent grid with the LDS-DMA look-ahead running across items changed nothing.)
// LDS layouts are chosen so that every access is conflict-free without padding the DMA's linear writes:
//   patch plane (one channel quad): [10 rows][38 slots of 16 B]: slot q = v * 9 + u holds patch column 4 u + v (q = 36, 37 unused):
//       the 8 tiles of a tile row read column 4 tx + b = slots 16 B apart, the next tile row lies 4 x 608 B = 128 B (mod 256)
//       further: 32 lanes x 8 B cover the 64 banks once (ds_read_b64);
//   V: [xi][channel quad][tile][4]: a transform wave's 16 consecutive lanes write 128 contiguous bytes, a product wave's
//       ds_read_b128 reads lane r + 16 g at g * 256 + r * 16.
// vmcnt discipline: LDS-DMA pieces and U fragment loads share one in-order counter, so a fragment load issued behind a piece
// cannot be consumed before the piece has landed. The next chunk's pieces are issued after component pair T_ISSUE of this
// chunk's products: the fragment loads behind them are consumed T_D (three) pairs later at the earliest, and the consumption of the
// last ones implies that the pieces have landed when the next transform starts.
#include "common.h"
#include "bn_final.h"
#include "../../include/denet_hip.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));

// buffer_store_dwordx4 with two wait states behind it (the hazard described in wino4f.hip: a register soffset exempts the store
// from the compiler's wait state, a VALU write of the data registers in the next cycle corrupts it)
__device__ __forceinline__ void t_store_b128(const f32x4& v, const i32x4_t& rsrc, int voff, int soff) {
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

struct W4TParams {
    const float* x;      // [N,H,W,C]
    const float* U;      // packed [C/16][36][K][16] (denet_conv_wino4t_pack)
    const float* bias;   // [K] or null
    const float* add;    // [N,H,W,K] or null
    float* y;            // [N,H,W,K]
    double* stats;       // [tile blocks][2][K] or null
    const float* bs_x;   // backward sums (wino4f.hip / winograd.hip wino_output_kernel)
    const float* bs_y;
    const float* bs_gamma;
    const float* bs_beta;
    const float* bs_mean;
    const float* bs_invstd;
    int bs_relu;
    int N, H, W, C, K;
    int bh, bw;          // tile blocks per image (rows of 2 tiles, columns of 8 tiles)
    int tiles_k;         // K / 64
    int chunks;          // C / 16
    int relu;
    unsigned x_bytes, y_bytes, u_bytes;
    unsigned long long* dbg;   // -DT_TRACE builds: s_memtime stamps of one workgroup per 64 (tools/exp/w4t_trace.py)
};

constexpr int T_OOB = (int)0xF0000000u;
constexpr int T_ROWS = 10;                        // patch rows of a block of 2 x 8 tiles
constexpr int T_QS = 38;                          // 16-byte slots per patch row (36 used)
constexpr int T_PLANE_SLOTS = T_ROWS * T_QS;      // 380
constexpr int T_PLANE_B = T_PLANE_SLOTS * 16;     // 6 080
constexpr int T_PATCH_SLOTS = 4 * T_PLANE_SLOTS;  // 1 520
constexpr int T_PATCH_B = 24 * 1024;              // 24 pieces of 64 slots: 6 per wave
constexpr int T_V_B = 36 * 1024;                  // 36 864
constexpr int T_LDS = T_PATCH_B + T_V_B;          // 61 440: two workgroups per CU
#ifndef T_STAG
#define T_STAG 0
#endif
#ifndef T_STAG_N
#define T_STAG_N 1
#endif
#ifndef T_D_
#define T_D_ 3
#endif
#ifndef T_ISSUE_
#define T_ISSUE_ 9
#endif
constexpr int T_D = T_D_;                         // component pairs whose filter fragments are loaded ahead
constexpr int T_R = 12;                           // ring of filter fragments (components)
constexpr int T_ISSUE = T_ISSUE_;                      // the next chunk's pieces leave behind this component pair
static_assert(36 % T_R == 0 && T_R >= 2 * T_D + 2, "the ring index must run on across chunks and cover the fragments in flight");

#define T_WAITCNT(vm) __builtin_amdgcn_s_waitcnt(((vm) & 15) | ((((vm) >> 4) & 3) << 14) | (7 << 4))              /* + lgkmcnt(0) */
#define T_WAIT_VM(vm) __builtin_amdgcn_s_waitcnt(((vm) & 15) | ((((vm) >> 4) & 3) << 14) | (7 << 4) | (15 << 8))  /* vmcnt only */
#define T_BARRIER()                        \
    {                                      \
        asm volatile("" ::: "memory");     \
        __builtin_amdgcn_s_barrier();      \
        asm volatile("" ::: "memory");     \
    }

// B^T of F(4x4,3x3) applied to a 6-vector, three of its six outputs (LH = 0: rows 0..2 from d0..d4; LH = 1: rows 3..5 from d1..d5)
template <int LH>
__device__ __forceinline__ void t_bt3(const f32x2 (&d)[6], f32x2 (&o)[3]) {
    if (LH == 0) {
        o[0] = __builtin_elementwise_fma(f32x2{-5.f, -5.f}, d[2], __builtin_elementwise_fma(f32x2{4.f, 4.f}, d[0], d[4]));
        const f32x2 p = __builtin_elementwise_fma(f32x2{-4.f, -4.f}, d[2], d[4]);
        const f32x2 q = __builtin_elementwise_fma(f32x2{-4.f, -4.f}, d[1], d[3]);
        o[1] = p + q;
        o[2] = p - q;
    } else {
        const f32x2 p = d[4] - d[2];
        const f32x2 q = d[3] - d[1];
        o[0] = __builtin_elementwise_fma(f32x2{2.f, 2.f}, q, p);
        o[1] = __builtin_elementwise_fma(f32x2{-2.f, -2.f}, q, p);
        o[2] = __builtin_elementwise_fma(f32x2{-5.f, -5.f}, d[3], __builtin_elementwise_fma(f32x2{4.f, 4.f}, d[1], d[5]));
    }
}
// all six outputs
__device__ __forceinline__ void t_bt6(const f32x2 (&t)[6], f32x2 (&o)[6]) {
    o[0] = __builtin_elementwise_fma(f32x2{-5.f, -5.f}, t[2], __builtin_elementwise_fma(f32x2{4.f, 4.f}, t[0], t[4]));
    const f32x2 p = __builtin_elementwise_fma(f32x2{-4.f, -4.f}, t[2], t[4]);
    const f32x2 q = __builtin_elementwise_fma(f32x2{-4.f, -4.f}, t[1], t[3]);
    o[1] = p + q;
    o[2] = p - q;
    const f32x2 p2 = t[4] - t[2];
    const f32x2 q2 = t[3] - t[1];
    o[3] = __builtin_elementwise_fma(f32x2{2.f, 2.f}, q2, p2);
    o[4] = __builtin_elementwise_fma(f32x2{-2.f, -2.f}, q2, p2);
    o[5] = __builtin_elementwise_fma(f32x2{-5.f, -5.f}, t[3], __builtin_elementwise_fma(f32x2{4.f, 4.f}, t[1], t[5]));
}

// the input transform of one thread: components (l, m), l = 3 LH .. 3 LH + 2, m = 0 .. 5 of its (tile, channel pair)
template <int LH>
__device__ __forceinline__ void t_transform(const char* prd, char* vwr) {
    f32x2 t[3][6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        f32x2 d[6];
#pragma unroll
        for (int a = LH; a < LH + 5; ++a) d[a] = *(const f32x2*)(prd + (a * T_QS + (b & 3) * 9 + (b >> 2)) * 16);
        if (LH == 0) d[5] = f32x2{0.f, 0.f};
        else d[0] = f32x2{0.f, 0.f};
        f32x2 o[3];
        t_bt3<LH>(d, o);
        t[0][b] = o[0];
        t[1][b] = o[1];
        t[2][b] = o[2];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        f32x2 o[6];
        t_bt6(t[i], o);
#pragma unroll
        for (int m = 0; m < 6; ++m) *(f32x2*)(vwr + (6 * (3 * LH + i) + m) * 1024) = o[m];
    }
}

// a double rotated right by N lanes inside its row of 16 lanes (two 32-bit DPP moves)
template <int N>
__device__ __forceinline__ double t_row_ror(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x120 + N, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x120 + N, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

constexpr float T_AT[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};

#ifdef T_TRACE
#define T_STAMP() if (p.dbg && (blockIdx.x & 63) == 0 && tid == 0 && dbg_n < 32) p.dbg[(blockIdx.x >> 6) * 32 + dbg_n++] = __builtin_amdgcn_s_memtime()
#else
#define T_STAMP()
#endif
#ifndef T_EXP
#define T_EXP 0       // experiment builds (tools/exp/w4t_variants.sh -DT_EXP=bits): 1 no input transform, 2 no products, 4 no epilogue,
                      // 8 no LDS-DMA, 16 stores into a 1 MB window, 32 no filter fragment loads, 64 no V fragment reads, 128 every second filter fragment pair only - wrong results, the time that is left tells what each phase costs
#endif

// EP: 0 = store only, 1 = + batch-norm column sums of what is stored, 2 = + backward sums of the batch norm in front
template <int EP>
__global__ __launch_bounds__(256, 2) void wino4t_kernel(const W4TParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const vbuf = smem + T_PATCH_B;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);       // neighbouring items (channel blocks of one tile block) share an L2
    const int kblk = (int)(bid % (uint32_t)p.tiles_k);
    const int brow = (int)(bid / (uint32_t)p.tiles_k);           // tile block = row of the statistics
    int blk = brow;
    const int bx = blk % p.bw;
    blk /= p.bw;
    const int by = blk % p.bh;
    const int n = blk / p.bh;
    const int k0 = kblk * 64;
    const int y0 = by * 8, x0 = bx * 32;            // first output pixel of the block
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);

    // ---- LDS-DMA pieces of this wave: piece = wave + 4 j covers patch slots 64 piece .. 64 piece + 63 ----
    int pc_off[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int L = (wave + 4 * j) * 64 + lane;
        const int pl = L / T_PLANE_SLOTS, rem = L - pl * T_PLANE_SLOTS;
        const int row = rem / T_QS, q = rem - row * T_QS;
        const int v = q / 9, u = q - v * 9;
        const int pcx = 4 * u + v;
        const int iy = y0 - 1 + row, ix = x0 - 1 + pcx;
        const bool ok = L < T_PATCH_SLOTS && q < 36 && pcx < 34 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        pc_off[j] = ok ? (((n * p.H + iy) * p.W + ix) * p.C + pl * 4) * 4 : T_OOB;
    }
    int d_chunk = 0;          // the chunk the next issue fetches
    auto issue = [&]() {
        if (T_EXP & 8) return;
        const bool live = d_chunk < p.chunks;
#pragma unroll
        for (int j = 0; j < 6; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_ptr_t)(smem + (wave + 4 * j) * 1024), 16, live ? pc_off[j] : T_OOB,
                                                     d_chunk * 64, 0, 0);
        d_chunk += 1;
    };

    // ---- transform role: lane = (half, tile column, tile row, low bit of the channel quad); wave = (high bit, component half) ----
    const int t_half = lane & 1, t_tx = (lane >> 1) & 7, t_ty = (lane >> 4) & 1;
    const int t_cq = ((lane >> 5) & 1) | ((wave & 1) << 1), t_lh = wave >> 1;
    const int t_rd = t_cq * T_PLANE_B + (4 * t_ty * T_QS + t_tx) * 16 + t_half * 8;
    char* const t_wr = vbuf + t_cq * 256 + (t_ty * 8 + t_tx) * 16 + t_half * 8;

    // ---- product role: wave = channel block kw (16 channels), all 16 tiles; lane = (r, g) ----
    const int r15 = lane & 15, g = lane >> 4;
    const int kw = wave;
    const char* const v_rd = vbuf + g * 256 + r15 * 16;
    // the filter fragments come through a buffer descriptor: one lane offset, the (chunk, component) offset is scalar
    const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void*)p.U, 0, p.u_bytes, 0x00020000);
    // (T_EXP 256: waves 2, 3 fetch the fragments of waves 0, 1 - the same L1 requests, half of them L2 hits... or L1 hits)
    const int u_voff = ((k0 + 16 * ((T_EXP & 256) ? (kw & 1) : kw) + r15) * 16 + 4 * g) * 4;
    const int u_xi = p.K * 64;                       // bytes per component
    const int u_chunk = 36 * u_xi;                   // bytes per chunk

    f32x4 acc[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // filter fragments 2 T_D components (T_D pairs) ahead: a ring indexed by component % T_R - T_R divides 36, so the index runs
    // on across chunks, and exceeds the 2 T_D + 2 components in flight (everything is unrolled; the registers follow liveness)
    f32x4 fu[T_R];
    if (T_EXP & 32) {
#pragma unroll
        for (int i = 0; i < T_R; ++i) asm volatile("" : "=v"(fu[i]));
    }
    auto load_u = [&](int chunk, int pair) {
        if (T_EXP & 32) return;
        if ((T_EXP & 128) && (pair & 1)) {          // half the filter traffic: odd pairs reuse the fragments of the pair before
            fu[(2 * pair) % T_R] = fu[(2 * pair - 2) % T_R];
            fu[(2 * pair + 1) % T_R] = fu[(2 * pair - 1) % T_R];
            return;
        }
        const int so = chunk * u_chunk + (2 * pair) * u_xi;
        fu[(2 * pair) % T_R] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rU, u_voff, so, 0));
        fu[(2 * pair + 1) % T_R] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rU, u_voff, so + u_xi, 0));
    };
    f32x4 fv[2][2];
    if (T_EXP & 64) asm volatile("" : "=v"(fv[0][0]), "=v"(fv[0][1]), "=v"(fv[1][0]), "=v"(fv[1][1]));
    auto read_v = [&](int pair) {
        if (T_EXP & 64) return;
        fv[pair & 1][0] = *(const f32x4*)(v_rd + (2 * pair) * 1024);
        fv[pair & 1][1] = *(const f32x4*)(v_rd + (2 * pair + 1) * 1024);
    };

#if T_STAG
    {
        const bool late = T_STAG == 1 ? ((blockIdx.x >> 8) & 1) && blockIdx.x < 512 : (blockIdx.x & 1) && blockIdx.x < 512;
        if (late) {
#pragma unroll
            for (int i = 0; i < T_STAG_N; ++i) __builtin_amdgcn_s_sleep(115);
        }
    }
#endif
    int dbg_n = 0;
    (void)dbg_n;
    T_STAMP();
    issue();
    if (!(T_EXP & 2)) {
#pragma unroll
        for (int pr = 0; pr < T_D; ++pr) load_u(0, pr);
    }

    for (int s = 0; s < p.chunks; ++s) {
        // this chunk's pieces have landed: only the 2 T_D look-ahead fragment loads behind them may still fly (behind the first
        // chunk that is implied: fragment loads issued behind the pieces have been consumed). Every wave is done with the
        // products of chunk s - 1: V is free
        T_WAIT_VM(2 * T_D);
        T_STAMP();
        T_BARRIER();
        T_STAMP();
        if (!(T_EXP & 1)) {
            const char* prd = smem + t_rd;
            if (t_lh == 0) t_transform<0>(prd, t_wr);
            else t_transform<1>(prd, t_wr);
        }
        T_WAITCNT(63);                // lgkmcnt(0): this wave's V rows are written (vmcnt left alone)
        T_STAMP();
        T_BARRIER();                  // V is complete, the patch buffer is free
        T_STAMP();
        if (!(T_EXP & 2)) {
            const int sn = s + 1 < p.chunks ? s + 1 : s;      // (the last chunk's look-ahead loads re-read its own fragments)
            read_v(0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pr = 0; pr < 18; ++pr) {
                // the fragment loads T_D pairs ahead (behind pair 17 - T_D: the next chunk's first pairs), the V fragments one pair
                // ahead
                if (pr + T_D < 18) load_u(s, pr + T_D);
                else load_u(sn, pr + T_D - 18);
                if (pr + 1 < 18) read_v(pr + 1);
                const f32x4 ua = fu[(2 * pr) % T_R], ub = fu[(2 * pr + 1) % T_R];
                const f32x4 va = fv[pr & 1][0], vb = fv[pr & 1][1];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[2 * pr] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[j], va[j], acc[2 * pr], 0, 0, 0);
                    acc[2 * pr + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ub[j], vb[j], acc[2 * pr + 1], 0, 0, 0);
                }
                // issue order of the pair: the loads slotted behind the first products (left to the compiler they sink to their uses)
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if (pr + 1 < 18) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (pr == T_ISSUE && s + 1 < p.chunks) {
                    issue();                 // the next chunk's pieces, into the buffer this chunk was transformed from
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else if (s + 1 < p.chunks) {
            issue();
        }
    }
    // (no piece is in flight: the last chunk issued none; the look-ahead loads behind its last pairs are never consumed)
    T_STAMP();
    if (T_EXP & 4) {
#pragma unroll
        for (int i = 0; i < 36; ++i) asm volatile("" ::"v"(acc[i]));
        return;
    }

    // ---- output transform in registers: Y = A^T M A, M[l][m] = acc[6 l + m] (4 output channels per lane) ----
    f32x4 Y[16];
    {
        f32x4 Z[6][4];
#pragma unroll
        for (int l = 0; l < 6; ++l)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
                bool fst = true;
#pragma unroll
                for (int m = 0; m < 6; ++m) {
                    const float c = T_AT[j][m];
                    if (c == 0.f) continue;
                    if (fst) a = c == 1.f ? acc[6 * l + m] : acc[6 * l + m] * c;
                    else if (c == 1.f) a += acc[6 * l + m];
                    else if (c == -1.f) a -= acc[6 * l + m];
                    else a = __builtin_elementwise_fma(f32x4{c, c, c, c}, acc[6 * l + m], a);
                    fst = false;
                }
                Z[l][j] = a;
            }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
                bool fst = true;
#pragma unroll
                for (int l = 0; l < 6; ++l) {
                    const float c = T_AT[i][l];
                    if (c == 0.f) continue;
                    if (fst) a = c == 1.f ? Z[l][j] : Z[l][j] * c;
                    else if (c == 1.f) a += Z[l][j];
                    else if (c == -1.f) a -= Z[l][j];
                    else a = __builtin_elementwise_fma(f32x4{c, c, c, c}, Z[l][j], a);
                    fst = false;
                }
                Y[4 * i + j] = a;
            }
    }

    T_STAMP();
    // ---- epilogue (wino4f.hip's): lane = (tile r15 of the block, channels k0 + 16 kw + 4 g .. + 3) ----
    const int oy = y0 + 4 * (r15 >> 3), ox = x0 + 4 * (r15 & 7);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const i32x4_t rsY = {(int)(unsigned)(unsigned long long)p.y, (int)(((unsigned long long)p.y >> 32) & 0xffffu), (int)p.y_bytes, 0x00020000};
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.add, 0, p.add ? p.y_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rBX = __builtin_amdgcn_make_buffer_rsrc((void*)p.bs_x, 0, p.bs_x ? p.y_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rBY = __builtin_amdgcn_make_buffer_rsrc((void*)p.bs_y, 0, p.bs_y ? p.y_bytes : 0u, 0x00020000);
    const float floor_ = p.relu ? 0.f : -__builtin_inff();
    const bool has_add = p.add != nullptr;          // (uniform: a pass without an add issues no loads and waits for none)
    const bool mask_y = p.bs_relu && p.bs_y, mask_x = p.bs_relu && !p.bs_y;
    double ds[8];
    const int kc = k0 + 16 * kw + 4 * g;
    const bool valid = oy < p.H && ox < p.W && kc < p.K;
    {
#pragma unroll
        for (int c = 0; c < 8; ++c) ds[c] = 0.0;
        const int voff = valid ? (((n * p.H + oy) * p.W + ox) * p.K + kc) * 4 : T_OOB;
        const int voff_st = (T_EXP & 16) ? (valid ? (voff & 0xFFFFF) : T_OOB) : voff;
        const int kcs = valid ? kc : 0;
        f32x4 b = z;
        if (p.bias) b = *(const f32x4*)(p.bias + kcs);
        f32x4 bmu = z, bis = z, bsc = z, bsh = z;
        if (EP == 2) {
            bmu = *(const f32x4*)(p.bs_mean + kcs);
            bis = *(const f32x4*)(p.bs_invstd + kcs);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                bsc[c] = (p.bs_gamma ? p.bs_gamma[kcs + c] : 1.f) * bis[c];
                bsh[c] = (p.bs_beta ? p.bs_beta[kcs + c] : 0.f) - bmu[c] * bsc[c];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 ssum = z, ssq = z;          // the four values of an output row in fp32, doubles from there
            f32x4 av[4], xv[4], yv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int soff = (i * p.W + j) * p.K * 4;
                av[j] = z;
                if (has_add) av[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, voff, soff, 0));
                if (EP == 2) {
                    xv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rBX, voff, soff, 0));
                    yv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rBY, voff, soff, 0));
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int soff = (i * p.W + j) * p.K * 4;
                f32x4 o = (Y[4 * i + j] + b) + av[j];
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = fmaxf(o[c], floor_);
                t_store_b128(o, rsY, voff_st, soff);
                if (EP == 2) {
                    f32x4 gq;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float mk = mask_y ? yv[j][c] : (mask_x ? fmaf(xv[j][c], bsc[c], bsh[c]) : 1.f);
                        gq[c] = mk > 0.f ? o[c] : 0.f;
                        ssq[c] += gq[c] * ((xv[j][c] - bmu[c]) * bis[c]);
                    }
                    ssum += gq;
                } else if (EP == 1) {
                    ssum += o;
                    ssq += o * o;
                }
            }
            if (EP != 0 && valid) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    ds[c] += (double)ssum[c];
                    ds[4 + c] += (double)ssq[c];
                }
            }
        }
    }
    T_STAMP();
    if (EP == 0) return;
    // over the 16 tiles of the wave = the 16 lanes of a DPP row (row_ror 8, 4, 2, 1: every lane ends with the row's sum; the
    // additions pair the same lanes as xor-shuffles would): the wave owns its 16 channels' column sums
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        ds[c] += t_row_ror<8>(ds[c]);
        ds[c] += t_row_ror<4>(ds[c]);
        ds[c] += t_row_ror<2>(ds[c]);
        ds[c] += t_row_ror<1>(ds[c]);
    }
    if (r15 == 0 && kc < p.K) {
        double* row = p.stats + (long)brow * 2 * p.K;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            row[kc + c] = ds[c];
            row[p.K + kc + c] = ds[4 + c];
        }
    }
    T_STAMP();
}

// U [36][K][C] (denet_conv_wino_filter, tile 4; K = output channels of the pass, C = its reduction) -> [C/16][36][K][16]
__global__ __launch_bounds__(256) void wino4t_pack_kernel(const float* __restrict__ U, float* __restrict__ P, int K, int C) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;          // one 16-byte quad of the destination
    const long total = 36L * K * C / 4;
    if (idx >= total) return;
    const int q = (int)(idx & 3);
    long rest = idx >> 2;
    const int k = (int)(rest % K);
    rest /= K;
    const int xi = (int)(rest % 36);
    const int chunk = (int)(rest / 36);
    *(f32x4*)(P + idx * 4) = *(const f32x4*)(U + ((long)xi * K + k) * C + chunk * 16 + q * 4);
}

unsigned long long* g_w4t_dbg = nullptr;

}  // namespace

#ifdef T_TRACE
extern "C" int denet_conv_wino4t_debug(unsigned long long* buf) { g_w4t_dbg = buf; return 0; }       // tools/exp/w4t_trace.py
#endif

// geometry the kernel covers: 3x3 stride 1 pad 1 (the caller's business), H and W multiples of 4, C a multiple of 16, K of 64
extern "C" int denet_conv_wino4t_ok(int N, int H, int W, int C, int K) {
    return (N > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0 && C > 0 && C % 16 == 0 && K > 0 && K % 64 == 0 &&
            (long)N * H * W * C * 4 < 0x7FFFFFFFL && (long)N * H * W * K * 4 < 0x7FFFFFFFL && 36L * K * C * 4 < 0x7FFFFFFFL) ? 1 : 0;
}

// rows of partial statistics a launch writes: one per block of 2 x 8 tiles
extern "C" int denet_conv_wino4t_stats_rows(int N, int H, int W) {
    return N * ((H + 7) / 8) * ((W + 31) / 32);
}

extern "C" int denet_conv_wino4t_pack(const float* u, float* packed, int C, int K, hipStream_t stream) {
    DENET_CHECK_ARG(u && packed && C > 0 && C % 16 == 0 && K > 0, "conv_wino4t_pack: bad arguments");
    const long total = 36L * K * C / 4;
    hipLaunchKernelGGL(wino4t_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, u, packed, K, C);
    DENET_CHECK_LAUNCH("conv_wino4t_pack");
    return DENET_OK;
}

// y = conv3x3(x) stride 1 pad 1 (+ bias) (+ add) (ReLU) from the packed F(4x4) filters; statistics / backward sums as
// denet_conv_wino2f_sums (stats_partial [rows][2][K] doubles, rows = denet_conv_wino4t_stats_rows)
extern "C" int denet_conv_wino4t_sums(const float* x, const float* u_packed, const float* bias, const float* add, float* y, int relu,
                                      double* stats_partial, size_t stats_bytes, int* stats_rows, const denet_bn_link* sums_of, int N,
                                      int H, int W, int C, int K, hipStream_t stream) {
    DENET_CHECK_ARG(x && u_packed && y, "conv_wino4t: null pointer");
    DENET_CHECK_ARG(denet_conv_wino4t_ok(N, H, W, C, K), "conv_wino4t: needs H, W %% 4 = 0, C %% 16 = 0, K %% 64 = 0");
    W4TParams p = {};
    p.x = x; p.U = u_packed; p.bias = bias; p.add = add; p.y = y;
    p.N = N; p.H = H; p.W = W; p.C = C; p.K = K;
    p.bh = (H + 7) / 8; p.bw = (W + 31) / 32;
    p.tiles_k = K / 64; p.chunks = C / 16; p.relu = relu;
    p.x_bytes = (unsigned)((size_t)N * H * W * C * 4);
    p.y_bytes = (unsigned)((size_t)N * H * W * K * 4);
    p.u_bytes = (unsigned)((size_t)36 * K * C * 4);
    p.dbg = g_w4t_dbg;
    const long blocks = (long)N * p.bh * p.bw;
    int ep = 0;
    if (stats_partial) {
        DENET_CHECK_ARG(stats_rows && stats_bytes >= (size_t)blocks * 2 * K * sizeof(double), "conv_wino4t: statistics buffer too small");
        *stats_rows = (int)blocks;
        p.stats = stats_partial;
        ep = 1;
        if (sums_of) {
            DENET_CHECK_ARG(sums_of->x && sums_of->mean && sums_of->invstd && (!sums_of->relu || sums_of->y || (sums_of->gamma && sums_of->beta)),
                            "conv_wino4t: incomplete batch-norm description for the backward sums");
            p.bs_x = sums_of->x; p.bs_y = sums_of->relu ? sums_of->y : nullptr; p.bs_gamma = sums_of->gamma; p.bs_beta = sums_of->beta;
            p.bs_mean = sums_of->mean; p.bs_invstd = sums_of->invstd; p.bs_relu = sums_of->relu;
            ep = 2;
        }
    }
    typedef void (*kern_t)(const W4TParams);
    static const kern_t kerns[3] = {wino4t_kernel<0>, wino4t_kernel<1>, wino4t_kernel<2>};
    static bool attr_done[3] = {};
    const kern_t fn = kerns[ep];
    if (!attr_done[ep]) {
        const hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        if (e != hipSuccess) {
            denet_set_error("conv_wino4t: hipFuncSetAttribute(%d B LDS): %s", T_LDS, hipGetErrorString(e));
            return -(int)e;
        }
        attr_done[ep] = true;
    }
    // (experiments: DENET_W4T_LDS = LDS bytes requested per workgroup, e.g. 100000 leaves room for ONE workgroup per CU)
    static const int lds_req = [] { const char* e = getenv("DENET_W4T_LDS"); const int v = e ? atoi(e) : 0; return v > T_LDS && v <= 163840 ? v : T_LDS; }();
    const int prof = denet_prof_begin(16, ep, 0, 0, stream);
    hipLaunchKernelGGL(fn, dim3((unsigned)(blocks * p.tiles_k)), dim3(256), lds_req, stream, p);
    denet_prof_end(prof, stream);
    DENET_CHECK_LAUNCH("conv_wino4t");
    return DENET_OK;
}
