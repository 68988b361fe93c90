// Winograd F(4x4,3x3): the 36 component products AND the output transform in one kernel - M never reaches HBM.
// Reference op: the `C[k,3]` layer, denet/layer/convolution.py:80-83 (forward) and its data gradient, model_cnn.py:318 (the
// same pipeline on dy with the rotated, channel-swapped filters). The un-fused path of winograd.hip runs the products as 36
// batched GEMMs that write M[36][T][K] and a transform kernel that reads it back: for a 128-channel layer on a 64x64 map that
// is 2 x 151 MB of the pass's 740 MB, and the K = 128 products need the HBM stream and the matrix pipe at the same time.
//
// Here a workgroup owns TB tiles x 64 output channels and walks the 36 components ITSELF:
//     for xi = (l, m):   M = V[xi][tiles][:] . U[xi][channels][:]^T        (v_mfma_f32_16x16x4_f32, 4 registers per lane)
//                        Y[i][j] += AT[i][l] * AT[j][m] * M                 (<= 16 fused multiply-adds per value, VALU)
// so only the 16 output positions of a (tile, channel) live in registers (64 per lane) - 16 waves of one 16 x 16 block each,
// i.e. the whole register file of a CU holds the outputs of 64 tiles x 64 channels (TB = 32: 8 waves, two workgroups per CU).
// Operands: V / U rows of 64 reduction channels (256 B) arrive by LDS-DMA (buffer_load ... lds, 16 B per lane), four chunks
// of (TB + 64) rows in flight; the 16-byte slots of a row are XOR-swizzled with the row index at the SOURCE address (the DMA
// writes LDS linearly), which makes the ds_read_b128 fragment reads conflict-free. One barrier per chunk, in the MIDDLE of the
// chunk's products: it publishes chunk s+1 (every wave has waited for its own pieces) and frees the buffer of chunk s-1 for
// the pieces of chunk s+3 - the matrix pipe never waits for it, the fragments of the next 16 channels are in registers.
// The epilogue is the output transform's (winograd.hip wino_output_kernel): bias / add / ReLU, the batch-norm column sums of
// what is stored, or the backward sums of the batch norm whose output gradient is written.
// Association: Y is accumulated component by component instead of A^T (M A) column by column: same sums, other rounding
// (measured against fp64 in tests/test_conv_fullsize_gpu.py like every other pass).
#include "common.h"
#include "bn_final.h"
#include "../../include/denet_hip.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));

typedef int i32x4_t __attribute__((ext_vector_type(4)));

// buffer_store_dwordx4 through inline assembly, with two wait states behind it. A store of more than 64 bits reads its data
// registers over several cycles; a vector instruction that overwrites them in the next cycle changes what is stored. The
// compiler's hazard table inserts the wait state for global / flat stores and for buffer stores with an IMMEDIATE offset,
// but exempts buffer stores whose soffset is a register - and on gfx950 that combination does corrupt the store: with the
// intrinsic, an epilogue in which `v_or_b32 v80, ...` directly followed `buffer_store_dwordx4 v[80:83], .., s11 offen` wrote wrong
// first dwords in lanes 12-15 of every row of 16 as soon as a second stream kept the memory system busy (found with a
// re-ordered epilogue, tools/exp/w4_repro.py; the order below happened to keep a few instructions between the two).
// tests/test_kernels_gpu.py::test_fused_winograd_f4_kernel_under_memory_pressure is the test that sees it.
__device__ __forceinline__ void store_b128(const f32x4& v, const i32x4_t& rsrc, int voff, int soff) {
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

struct W4Params {
    const float* V;      // [36][T][C]
    const float* U;      // [36][K][C]
    const float* bias;   // [K] or null
    const float* add;    // [N,H,W,K] or null
    float* y;            // [N,H,W,K]
    double* stats;       // [tile blocks][2][K] or null
    const float* bs_x;   // backward sums (see winograd.hip BnFoldDev / wino_output_kernel)
    const float* bs_y;
    const float* bs_gamma;
    const float* bs_beta;
    const float* bs_mean;
    const float* bs_invstd;
    int bs_relu;
    int N, H, W, C, K, TH, TW;
    int T;
    int relu;
    int tiles_k;         // channel blocks (64 or 32 channels per workgroup)
    int chunks;          // C / 64
    unsigned v_bytes, u_bytes, y_bytes;
    unsigned long long* dbg;   // -DW4_TRACE builds: s_memtime stamps of workgroup 0 (tools/exp/w4_trace.py)
    BnFinalDev fin;      // with stats: the last workgroup of a channel block reduces the rows itself (bn_final.h); counter null: off
    int tiles_t;         // tile blocks = rows of `stats` = workgroups per channel block
};

constexpr int W4_OOB = (int)0xF0000000u;
constexpr float W4_AT[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};

// s_waitcnt vmcnt(vm) only (lgkmcnt / expcnt left alone) and the compiler kept from moving memory operations across
#define W4_WAIT_VM(vm) __builtin_amdgcn_s_waitcnt(((vm) & 15) | ((((vm) >> 4) & 3) << 14) | (7 << 4) | (15 << 8))
#ifndef W4_EXP
#define W4_EXP 0      // experiment builds (tools/exp/w4_trace_build.sh -DW4_EXP=bits): 1 no fragment reads, 2 no updates (one chain over all
                      // components), 4 no barrier, 8 no DMA, 16 no products (the fragments are still read)
#endif
#ifdef W4_TRACE
#define W4_STAMP() if (p.dbg && blockIdx.x == 0 && lane == 0 && dbg_n < 600) p.dbg[wave * 640 + dbg_n++] = __builtin_amdgcn_s_memtime()
#else
#define W4_STAMP()
#endif
#define W4_BARRIER()                       \
    {                                      \
        asm volatile("" ::: "memory");     \
        __builtin_amdgcn_s_barrier();      \
        asm volatile("" ::: "memory");     \
    }

// TB tiles x 64 channels per workgroup; a wave owns 16 tiles x NB blocks of 16 channels (NB = 2: one tile fragment serves two
// products, three fragment reads per eight products instead of four, and a 16-channel block of the wave's own products
// covers 256 cycles of LDS latency), (TB / 16) x (4 / NB) waves.
// EP: 0 = store only, 1 = + batch-norm column sums of what is stored, 2 = + backward sums of the batch norm in front
// NBUF = 4: four 32 KB chunk buffers, the whole LDS of a CU minus the epilogue's scratch - one workgroup per CU. NBUF = 3 (32-tile
// blocks, 24 KB per chunk): 72 KB, TWO workgroups per CU whose barriers and DMA waits cover each other.
template <int TB, int KB, int NB, int EP, int NBUF>
__device__ __forceinline__ void wino4f_body(const W4Params& p) {
    constexpr int W4_SLOT = NBUF == 4 ? 32768 : TB * 256 + KB * 256;      // LDS bytes per chunk buffer (V rows, then U rows)
    constexpr int NW = (TB / 16) * (KB / 16 / NB);      // waves
    constexpr int V_BYTES = TB * 256;             // a chunk of V rows
    constexpr int VP = TB / 4;                    // 1 KB pieces of the V chunk (4 rows each)
    constexpr int PIECES = VP + KB / 4;
    constexpr int PPW = PIECES / NW;              // pieces per wave and chunk
    static_assert(PIECES % NW == 0, "pieces divide over the waves");
    static_assert((NBUF == 4 || NBUF == 3) && V_BYTES + KB * 256 <= W4_SLOT, "chunk buffer holds the V and the U rows");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tw = wave % (TB / 16), kw = wave / (TB / 16);
    const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
    const int kblk = (int)(bid % (uint32_t)p.tiles_k), tblk = (int)(bid / (uint32_t)p.tiles_k);
    const int t0 = tblk * TB, k0 = kblk * KB;
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void*)p.V, 0, p.v_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void*)p.U, 0, p.u_bytes, 0x00020000);

    // ---- DMA pieces of this wave: piece q = wave + NW * j; q < VP: rows 4q..4q+3 of the V chunk, else of the U chunk ----
    int pv_off[PPW];          // byte offset of this lane's 16 bytes inside component 0, chunk 0
    const int prow = lane >> 4, pslot = lane & 15;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int q = wave + NW * j;
        const bool isv = q < VP;
        const int row = 4 * (isv ? q : q - VP) + prow;
        const int chunk16 = pslot ^ (row & 15);
        // a V row beyond the last tile (T % TB != 0, last tile block) is zero-filled explicitly: out-of-range offset, like the
        // trailing pieces below - not left to whatever lies behind the component (for component 35: behind V)
        pv_off[j] = (!isv || t0 + row < p.T) ? ((isv ? t0 : k0) + row) * p.C * 4 + chunk16 * 16 : W4_OOB;
    }
    const int compV = p.T * p.C * 4, compU = p.K * p.C * 4;     // bytes per component
    int d_cc = 0, d_sV = 0, d_sU = 0, d_left = 36 * p.chunks, d_buf = 0;
    auto issue = [&]() {
        const bool live = d_left > 0;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int q = wave + NW * j;
            const bool isv = q < VP;
            char* dst = smem + d_buf * W4_SLOT + (isv ? q * 1024 : V_BYTES + (q - VP) * 1024);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(isv ? rV : rU, (lds_ptr_t)dst, 16, live ? pv_off[j] : W4_OOB, isv ? d_sV : d_sU,
                                                     0, 0);
        }
        d_left -= 1;
        d_buf = (d_buf + 1 == NBUF) ? 0 : d_buf + 1;
        d_cc += 1;
        if (d_cc == p.chunks) {
            d_cc = 0;
            d_sV += compV - (p.chunks - 1) * 256;
            d_sU += compU - (p.chunks - 1) * 256;
        } else {
            d_sV += 256;
            d_sU += 256;
        }
    };

    // ---- fragment addresses: lane (r = lane % 16, g = lane / 16) reads row 16 grp + r, 16-byte slot (4 kb + g) ^ r of the
    // chunk; one register per 16-channel block and operand, the chunk's buffer is the instruction's immediate offset (0 or one
    // slot) and the PAIR of buffers a flip of bit 16 in the registers every second chunk
    const int r15 = lane & 15, g = lane >> 4;
    const int lo = ((g ^ (r15 & 3)) << 4), h2 = (r15 >> 2) & 3;
    int aV[4], aU[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        aV[kb] = (16 * tw + r15) * 256 + lo + ((kb ^ h2) << 6);
        aU[kb] = V_BYTES + (16 * NB * kw + r15) * 256 + lo + ((kb ^ h2) << 6);
    }

    f32x4 Y[NB][16];
    f32x4 fu[2][NB], fv[2];
    if (W4_EXP & 1) asm volatile("" : "=v"(fu[0][0]), "=v"(fu[1][0]), "=v"(fu[0][NB - 1]), "=v"(fu[1][NB - 1]), "=v"(fv[0]), "=v"(fv[1]));
    // NBUF = 4: `par` selects the buffer of the current pair (an immediate offset), the pair is a flip of bit 16 in the address
    // registers; NBUF = 3: par = 0 reads the current chunk's buffer (byte offset boff), 1 the next chunk's
    int boff = 0;
    auto frags = [&](int par, int kb, int slot) {
        if (W4_EXP & 1) return;
        const int off = NBUF == 4 ? par * W4_SLOT : (par == 0 ? boff : (boff + W4_SLOT == NBUF * W4_SLOT ? 0 : boff + W4_SLOT));
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) fu[slot][nb] = *(const f32x4*)(smem + aU[kb] + nb * 4096 + off);
        fv[slot] = *(const f32x4*)(smem + aV[kb] + off);
    };
    // two accumulator chains per block (even / odd products of a 16-channel block): a product never waits for the one issued
    // before it. ZC: the block starts a component (the chains start from zero: no clearing pass)
    f32x4 Mc[NB][2];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) Mc[nb][0] = Mc[nb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mm = [&](int slot, auto ZC) {
        constexpr bool zc = decltype(ZC)::value && !(W4_EXP & 2);
        if (W4_EXP & 16) {
            asm volatile("" ::"v"(fu[slot][0]), "v"(fv[slot]));
            return;
        }
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                Mc[nb][t & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fu[slot][nb][t], fv[slot][t], (zc && t < 2) ? z4 : Mc[nb][t & 1], 0, 0, 0);
    };
    // The output transform in two stages, every multiply-add packed (v_pk_fma_f32).
    // Row stage, per component (l, m):  Z[j] += AT[j][m] M          (<= 4 per value)
    // column stage, per finished row l: Y[i][j] += AT[i][l] Z[j]    (<= 16 per value, six times)
    // = 180 multiply-adds per value instead of the 324 of a one-stage update. Z / Y pass through an opaque asm after every
    // update: left alone, the compiler sinks the updates to the epilogue and keeps every component's M alive until then (spilled).
    f32x4 Z[NB][4];
    auto fma4 = [&](f32x4& acc, float c, const f32x4& v, bool first) {
        f32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
        f32x2 alo = {acc[0], acc[1]}, ahi = {acc[2], acc[3]};
        const f32x2 cc = {c, c};
        if (first) {
            if (c != 1.f) { lo = lo * cc; hi = hi * cc; }
            alo = lo; ahi = hi;
        } else if (c == 1.f) { alo += lo; ahi += hi; }
        else { alo = __builtin_elementwise_fma(lo, cc, alo); ahi = __builtin_elementwise_fma(hi, cc, ahi); }
        acc = f32x4{alo[0], alo[1], ahi[0], ahi[1]};
    };
    auto update = [&](auto LI, auto MI) {
        constexpr int l = decltype(LI)::value, m = decltype(MI)::value;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (W4_EXP & 2) {
                if (l == 5 && m == 5) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) Y[nb][i] = Mc[nb][0] + Mc[nb][1];
                }
                continue;
            }
            const f32x4 M = Mc[nb][0] + Mc[nb][1];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float c = W4_AT[j][m];
                // the first contribution to Z[j] of a row: m = 0 for j = 0, m = 1 for the others (AT[j][0] = 0)
                const bool first = (j == 0) ? m == 0 : m == 1;
                if (c != 0.f) {
                    fma4(Z[nb][j], c, M, first);
                    asm volatile("" : "+v"(Z[nb][j]));
                }
            }
            if (m == 5) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float c = W4_AT[i][l];
                    // the first contribution to Y[i][.]: row 0 for i = 0, row 1 for the others
                    const bool first = (i == 0) ? l == 0 : l == 1;
                    if (c != 0.f) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            fma4(Y[nb][4 * i + j], c, Z[nb][j], first);
                            asm volatile("" : "+v"(Y[nb][4 * i + j]));
                        }
                    }
                }
            }
        }
    };
    // issue order of a 16-channel block: the two fragment reads of the NEXT block go out behind the first two products of this
    // one (left to the compiler they are sunk below the products that free their registers: the LDS latency fully exposed)
    auto order = [&]() {
        // fragment reads of the NEXT block behind the first products of this one, one read per product
#pragma unroll
        for (int r = 0; r < NB + 1; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NB - (NB + 1), 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    int dbg_n = 0;
    (void)dbg_n;
    // NBUF - 1 chunks in flight
    issue();
    issue();
    if (NBUF == 4) issue();
    W4_WAIT_VM((NBUF - 2) * PPW);
    W4_BARRIER();
    frags(0, 0, 0);
    using T_ = std::true_type;
    using F_ = std::false_type;
    // one chunk of 64 reduction channels out of buffer `par` of the current pair; the first fragments are in slot 0.
    // FIRST: the chunk starts a component
    auto chunk = [&](auto PAR, auto FIRST) {
        constexpr int par = decltype(PAR)::value;
        frags(NBUF == 4 ? par : 0, 1, 1);
        mm(0, FIRST);
        order();
        frags(NBUF == 4 ? par : 0, 2, 0);
        mm(1, F_{});
        order();
        // chunk s+1 published, the buffer of chunk s-1 free: its pieces (chunk s+3) leave now
        W4_STAMP();
        W4_WAIT_VM((NBUF - 3) * PPW);
        W4_STAMP();
        if (!(W4_EXP & 4)) W4_BARRIER();
        W4_STAMP();
        if (!(W4_EXP & 8)) issue();
        frags(NBUF == 4 ? par : 0, 3, 1);
        mm(0, F_{});
        order();
        if (NBUF == 4 && par == 1) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                aV[kb] ^= 2 * W4_SLOT;
                aU[kb] ^= 2 * W4_SLOT;
            }
        }
        frags(NBUF == 4 ? (par ^ 1) : 1, 0, 0);
        mm(1, F_{});
        order();
        if (NBUF != 4) boff = (boff + W4_SLOT == NBUF * W4_SLOT) ? 0 : boff + W4_SLOT;
    };
    auto component = [&](auto LI, auto MI) {
        chunk(std::integral_constant<int, 0>{}, T_{});
        chunk(std::integral_constant<int, 1>{}, F_{});
        for (int cc = 2; cc < p.chunks; cc += 2) {
            chunk(std::integral_constant<int, 0>{}, F_{});
            chunk(std::integral_constant<int, 1>{}, F_{});
        }
        update(LI, MI);
    };
    auto row = [&](auto LI) {
        component(LI, std::integral_constant<int, 0>{});
        component(LI, std::integral_constant<int, 1>{});
        component(LI, std::integral_constant<int, 2>{});
        component(LI, std::integral_constant<int, 3>{});
        component(LI, std::integral_constant<int, 4>{});
        component(LI, std::integral_constant<int, 5>{});
    };
    row(std::integral_constant<int, 0>{});
    row(std::integral_constant<int, 1>{});
    row(std::integral_constant<int, 2>{});
    row(std::integral_constant<int, 3>{});
    row(std::integral_constant<int, 4>{});
    row(std::integral_constant<int, 5>{});
    W4_STAMP();
    __builtin_amdgcn_s_waitcnt(0);       // the trailing (out-of-range) pieces have landed before the buffers are reused

    // ---- epilogue: lane = (tile t0 + 16 tw + r15, channels k0 + 16 (NB kw + nb) + 4 g .. + 3, nb = 0 .. NB-1) ----
    // every tensor of the output's shape goes through a buffer descriptor: one 32-bit lane offset (out of range for a lane
    // without a tile / channels: its stores are dropped, its loads return 0; an absent tensor is a descriptor of 0 bytes) +
    // a wave-uniform offset per output position. The operands of the four positions of an output row are loaded together.
    const int t = t0 + 16 * tw + r15;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const int tx = t % p.TW;
    const int ty = (t / p.TW) % p.TH;
    const int n = t / (p.TW * p.TH);
    const i32x4_t rsY = {(int)(unsigned)(unsigned long long)p.y, (int)(((unsigned long long)p.y >> 32) & 0xffffu), (int)p.y_bytes, 0x00020000};
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.add, 0, p.add ? p.y_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)p.bs_x, 0, p.bs_x ? p.y_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void*)p.bs_y, 0, p.bs_y ? p.y_bytes : 0u, 0x00020000);
    const float floor_ = p.relu ? 0.f : -__builtin_inff();
    // the ReLU mask of the backward sums: from the forward output (bs_y), recomputed from x, or none
    const bool mask_y = p.bs_relu && p.bs_y, mask_x = p.bs_relu && !p.bs_y;
    double ds[NB][8];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int kc = k0 + 16 * (NB * kw + nb) + 4 * g;
        const bool valid = t < p.T && kc < p.K;
#pragma unroll
        for (int c = 0; c < 8; ++c) ds[nb][c] = 0.0;
        const int voff = valid ? (((n * p.H + 4 * ty) * p.W + 4 * tx) * p.K + kc) * 4 : W4_OOB;
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
                av[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, voff, soff, 0));
                if (EP == 2) {
                    xv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rX, voff, soff, 0));
                    yv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rR, voff, soff, 0));
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int soff = (i * p.W + j) * p.K * 4;
                f32x4 acc = (Y[nb][4 * i + j] + b) + av[j];
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = fmaxf(acc[c], floor_);
                store_b128(acc, rsY, voff, soff);
                if (EP == 2) {
                    f32x4 gq;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float mk = mask_y ? yv[j][c] : (mask_x ? fmaf(xv[j][c], bsc[c], bsh[c]) : 1.f);
                        gq[c] = mk > 0.f ? acc[c] : 0.f;
                        ssq[c] += gq[c] * ((xv[j][c] - bmu[c]) * bis[c]);
                    }
                    ssum += gq;
                } else if (EP == 1) {
                    ssum += acc;
                    ssq += acc * acc;
                }
            }
            if (EP != 0 && valid) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    ds[nb][c] += (double)ssum[c];
                    ds[nb][4 + c] += (double)ssq[c];
                }
            }
        }
    }
    W4_STAMP();
    if (EP == 0) return;
    // over the 16 tiles of the wave (shuffles inside each group of 16 lanes), then over the tile waves through LDS in wave order
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int off = 8; off > 0; off >>= 1)
#pragma unroll
            for (int c = 0; c < 8; ++c) ds[nb][c] += __shfl_xor(ds[nb][c], off, 64);
    __syncthreads();
    double* red = (double*)smem;                  // [tile wave][2][KB channels]
    if (r15 == 0) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                red[(tw * 2 + 0) * KB + 16 * (NB * kw + nb) + 4 * g + c] = ds[nb][c];
                red[(tw * 2 + 1) * KB + 16 * (NB * kw + nb) + 4 * g + c] = ds[nb][4 + c];
            }
    }
    __syncthreads();
    if (tid < 2 * KB) {
        const int which = tid / KB, ch = tid % KB;
        double a = 0.0;
#pragma unroll
        for (int w = 0; w < TB / 16; ++w) a += red[(w * 2 + which) * KB + ch];
        if (k0 + ch < p.K) bnf_store(p.stats + ((long)tblk * 2 + which) * p.K + k0 + ch, a);
    }
    // the last tile block of this channel block to arrive finishes the batch norm's reduction over all rows (bn_final.h)
    bnf_tail<NW * 64>(p.fin, p.stats, p.tiles_t, k0, KB, kblk, (unsigned)p.tiles_t, (int*)(smem + 16384));
}

// 8 waves, up to 256 registers each (two waves per SIMD), one workgroup per CU ...
template <int EP>
__global__ __launch_bounds__(512, 2) void wino4f_kernel_64(const W4Params p) {
    wino4f_body<64, 64, 2, EP, 4>(p);
}
template <int EP>
__global__ __launch_bounds__(512, 2) void wino4f_kernel_32(const W4Params p) {
    wino4f_body<32, 64, 1, EP, 4>(p);
}
// ... or 4 waves on 32 tiles x 64 channels (a wave: 16 tiles x 32 channels), three chunk buffers: two workgroups per CU
template <int EP>
__global__ __launch_bounds__(256, 2) void wino4f_kernel_32x2(const W4Params p) {
    wino4f_body<32, 64, 2, EP, 3>(p);
}
// ... or 4 waves on 32 tiles x 32 channels (a wave: 16 tiles x 16 channels), 16 KB chunk buffers: two to three workgroups per CU where
// 32 x 64 blocks would leave one 4-wave workgroup per CU (the 32x32 maps: 64 x 8 workgroups)
template <int EP>
__global__ __launch_bounds__(256, 2) void wino4f_kernel_32k(const W4Params p) {
    wino4f_body<32, 32, 1, EP, 3>(p);
}

int g_w4_mode = -1;
unsigned long long* g_w4_dbg = nullptr;       // -1: DENET_WINO4F / DENET_WINO4F_TB from the environment; 0: off; 32 / 64: that tile block wherever it fits

}  // namespace

// tests / experiments: overrides the environment's choice of the fused F(4x4) kernel (-1 restores it); returns the old value
#ifdef W4_TRACE
extern "C" int denet_conv_wino4f_debug(unsigned long long* buf) { g_w4_dbg = buf; return 0; }       // tools/exp/w4_trace.py
#endif
extern "C" int denet_conv_wino4f_mode(int mode) {
    const int old = g_w4_mode;
    g_w4_mode = (mode == 0 || mode == 32 || mode == 33 || mode == 34 || mode == 64) ? mode : -1;
    return old;
}

// the tile-block size the fused kernel would use for this problem (0: the un-fused path runs): a launch has to fill the chip
// with ONE round of workgroups, each of which walks all 36 components
int denet_wino4f_block(int tile, long T, int C, int K) {
    if (tile != 4 || C % 128 != 0 || K % 64 != 0 || T <= 0 || T * 64 * K >= (1L << 31)) return 0;
    // V / U go through 32-bit buffer descriptors and the kernel steps through the components with a signed 32-bit byte offset
    // (d_sV reaches 36 * T * C * 4): a problem beyond that takes the un-fused path instead of failing in denet_wino4f_run
    if (36L * T * C * 4 >= (1L << 31) || 36L * K * C * 4 >= (1L << 31)) return 0;
    static const int env_on = [] { const char* e = getenv("DENET_WINO4F"); return e ? atoi(e) : 1; }();
    static const int env_tb = [] { const char* e = getenv("DENET_WINO4F_TB"); return e ? atoi(e) : 0; }();
    const int mode = g_w4_mode >= 0 ? g_w4_mode : (env_on ? env_tb : 0);
    if (g_w4_mode < 0 && !env_on) return 0;
    if (g_w4_mode == 0) return 0;
    if (mode == 32 || mode == 33 || mode == 34 || mode == 64) return mode;
    // The workgroups of a launch have to fill the 256 CUs in ONE round, and two small workgroups per CU beat one large one: their
    // barriers and DMA waits cover each other. Kernel alone, epilogues as a training step runs them (statistics / backward sums +
    // add), l2 = 64x64 map 128 -> 128, l3 = 32x32 map 256 -> 256 (tools/exp/w4_time.py, W4_STEP_LIKE=1):
    //   shape (mode)                            workgroups l2 / l3    l2 fwd  l2 dgrad   l3 fwd  l3 dgrad  [us]
    //   64 tiles x 64 ch, 8 waves (64)              256 / 128           118     186       170     200
    //   32 x 64, 8 waves (32)                       512 / 256           122     170       105     119
    //   32 x 64, 4 waves, two per CU (33)           512 / 256           103     163       110     141
    //   32 x 32, 4 waves, up to three per CU (34)  1024 / 512           118     161       101     119
    // -> 32 x 64 as two 4-wave workgroups per CU where that gives >= 448 workgroups, the 8-wave shapes for smaller grids. The
    // 32 x 32 shape (mode 34, forced only) wins the table on the 32x32 maps but LOSES inside the training step (same box, A / B:
    // 1 030-1 044 against 1 054-1 056 img/s, this kernel 5.20 against 5.00 ms per step): more operand traffic per product, and its
    // third workgroup per CU takes slots from the other stream's kernels. A 16x16 map (T = 512) has too few tiles either way.
    const long kb = K / 64;
    if (((T + 31) / 32) * kb >= 448) return 33;
    if (((T + 63) / 64) * kb >= 224) return 64;
    if (((T + 31) / 32) * kb >= 224) return 32;
    return 0;
}

// rows of partial statistics the fused kernel writes for this problem
int denet_wino4f_stats_rows(int tb, long T) {
    const int tiles = tb == 64 ? 64 : 32;       // 32, 33, 34: 32-tile blocks
    return (int)((T + tiles - 1) / tiles);
}

int denet_wino4f_run(int tb, const float* V, const float* U, const float* bias, const float* add, float* y, double* stats,
                     const float* bs_x, const float* bs_y, const float* bs_gamma, const float* bs_beta, const float* bs_mean,
                     const float* bs_invstd, int bs_relu, int N, int H, int W, int C, int K, int relu, hipStream_t stream) {
    DENET_CHECK_ARG(V && U && y && (tb == 32 || tb == 33 || tb == 34 || tb == 64), "conv_wino4f: bad arguments");
    DENET_CHECK_ARG(H % 4 == 0 && W % 4 == 0 && C % 128 == 0 && K % 64 == 0, "conv_wino4f: unsupported geometry");
    W4Params p = {};
    p.V = V; p.U = U; p.bias = bias; p.add = add; p.y = y; p.stats = stats;
    p.bs_x = bs_x; p.bs_y = bs_y; p.bs_gamma = bs_gamma; p.bs_beta = bs_beta; p.bs_mean = bs_mean; p.bs_invstd = bs_invstd;
    p.bs_relu = bs_relu;
    p.N = N; p.H = H; p.W = W; p.C = C; p.K = K; p.TH = H / 4; p.TW = W / 4;
    const long T = (long)N * p.TH * p.TW;
    const size_t vb = (size_t)36 * T * C * 4, ub = (size_t)36 * K * C * 4;
    DENET_CHECK_ARG(vb < 0xE0000000ul && ub < 0xE0000000ul, "conv_wino4f: operand too large for a buffer descriptor");
    const int kbw = tb == 34 ? 32 : 64;           // output channels per workgroup
    p.T = (int)T; p.relu = relu; p.tiles_k = K / kbw; p.chunks = C / 64;
    p.dbg = g_w4_dbg;
    p.v_bytes = (unsigned)vb; p.u_bytes = (unsigned)ub; p.y_bytes = (unsigned)((size_t)T * 16 * K * 4);
    const int which = tb == 64 ? 1 : (tb == 33 ? 2 : (tb == 34 ? 3 : 0));
    const int tiles = tb == 64 ? 64 : 32;
    const int tiles_t = (int)((T + tiles - 1) / tiles);
    const int lds = which == 2 ? 3 * (32 * 256 + 16384) : (which == 3 ? 3 * (32 * 256 + 32 * 256) : 4 * 32768);
    const int ep = !stats ? 0 : (bs_x ? 2 : 1);
    p.tiles_t = tiles_t;
    // the batch norm these sums belong to, if the caller armed it (one counter per channel block; a last workgroup reads
    // tiles_t rows of its 32 / 64 columns: a few hundred KB at most)
    if (ep && (long)tiles_t * kbw * 16 <= (1L << 20)) p.fin = denet_bn_final_take(ep, K, p.tiles_k);
    typedef void (*kern_t)(const W4Params);
    static const kern_t kerns[4][3] = {{wino4f_kernel_32<0>, wino4f_kernel_32<1>, wino4f_kernel_32<2>},
                                       {wino4f_kernel_64<0>, wino4f_kernel_64<1>, wino4f_kernel_64<2>},
                                       {wino4f_kernel_32x2<0>, wino4f_kernel_32x2<1>, wino4f_kernel_32x2<2>},
                                       {wino4f_kernel_32k<0>, wino4f_kernel_32k<1>, wino4f_kernel_32k<2>}};
    static bool attr_done[4][3] = {};
    const kern_t fn = kerns[which][ep];
    if (!attr_done[which][ep]) {
        const hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            denet_set_error("conv_wino4f: hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
            return -(int)e;
        }
        attr_done[which][ep] = true;
    }
    const int prof = denet_prof_begin(14, tb, ep, which >= 2 ? 3 : 4, stream);      // (tile block, epilogue) = the instantiation
    hipLaunchKernelGGL(fn, dim3((unsigned)(tiles_t * p.tiles_k)), dim3(which >= 2 ? 256 : 512), lds, stream, p);
    denet_prof_end(prof, stream);
    DENET_CHECK_LAUNCH("conv_wino4f");
    return DENET_OK;
}
