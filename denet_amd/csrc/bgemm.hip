// Batched short-reduction GEMM for gfx950:  C_b[M,N] = A_b[M,K] * B_b[N,K]^T  (row-major, K contiguous, fp32),
// b = 0 .. batch-1.  These are the component products of the Winograd passes (winograd.hip: 16 or 36 members,
// K = channel count = 64 .. 512, i.e. only 2 .. 16 reduction chunks per output tile), which replace the cuDNN
// convolution of the reference's `C[k,3]` layers (denet/layer/convolution.py:80-83; gradient model_cnn.py:318).
//
// STATUS: opt-in (DENET_BGEMM=1 adds it to the measured candidates of denet_gemm_batched_tune). Alone it beats the
// implicit-GEMM kernel on these shapes by 10-17 %; inside a training step it measured neutral to -2 % (persistent
// workgroups do not interleave with the second stream's kernels), see DESIGN.md section 3.
//
// Why a kernel of its own (the implicit-GEMM kernel of igemm.hip ran these at 55-65 % of the fp32 MFMA rate):
//   * a tile lives for 2-16 chunks, so per-tile overheads (first-chunk latency, result stores) are a third of its life;
//   * tiles x members rarely fill the 256 CUs a whole number of times: the last, partial round ran at 1/8 .. 1/2 load.
// Design: PERSISTENT workgroups with a STREAM-K partition. The work is the linear sequence of (tile, chunk) units, tile
// major; workgroup g owns the contiguous units [g U/G, (g+1) U/G): every workgroup multiplies the same number of chunks
// (+-1), whatever the tile count. Its units form ONE software-pipelined stream: the operand loads of the next tile's
// first chunks are in flight while the current tile's results are stored, so nothing is exposed at a tile boundary.
// A tile cut by a range boundary is finished by the workgroup that owns its FIRST chunk: the owner of the tail part
// (which reaches it first thing in its range) publishes its partial accumulators (write-through stores + flag), the
// owner of the head part (which reaches it last thing in its range) adds them and writes C: the sum is
// head chunks (in order) + tail chunks (in order), a fixed association for a given (shape, G) - deterministic.
// G and the tile shape are fixed per geometry by the host, so results are reproducible run to run.
//
// Main loop: the pipelined loop of igemm.hip (two LDS buffers, one register staging set, every LDS write / global load
// / fragment read slotted behind an MFMA with sched_group_barrier), v_mfma_f32_32x32x2_f32, accumulators transposed so
// that the epilogue stores 16 bytes per lane.
#include "common.h"
#include <type_traits>
#include <stdlib.h>

namespace {

constexpr int BK = 32;
constexpr int LDK = BK + 4;

struct BgemmParams {
    const float* A;
    const float* B;
    float* C;
    int M, N, K, batch;
    long sA, sB, sC;        // member strides (elements)
    int tiles_m, tiles_n, nk;
    long units;             // batch * tiles_m * tiles_n * nk
    int G;                  // persistent workgroups
    float* partial;         // [G][BM*BN] fix-up slabs (tail parts of cut tiles)
    unsigned* flags;        // [G] epoch of the slab's last publication
    unsigned* err;          // set to 1 if a wait timed out
    unsigned epoch;
};

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, unsigned bytes) {
    // the descriptor must be provably wave-uniform, else every buffer op is wrapped in a waterfall loop
    const unsigned long long a = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const unsigned n = __builtin_amdgcn_readfirstlane(bytes);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, n, 0x00020000);
}

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void bgemm_kernel(const BgemmParams p) {
    constexpr int WM = 2, WN = 2;
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    constexpr int SZA = BM * LDK, SZB = BN * LDK;
    constexpr int PA = BM / 32, PB = BN / 32;
    constexpr int KB = BK / 8;
    static_assert(PA <= KB && PB <= KB, "one staging pass per MFMA block");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;             // [2][SZA]
    float* sB = smem + 2 * SZA;   // [2][SZB]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int q8 = tid & 7, row8 = tid >> 3;

    // XCD-aware logical id: the workgroups of one XCD own a contiguous eighth of the unit sequence (a few members: their
    // B matrices stay in that XCD's L2, A is streamed once)
    const int g = (int)xcd_remap(blockIdx.x, gridDim.x);
    const long u0 = (long)g * p.units / p.G;
    const long u1 = (long)(g + 1) * p.units / p.G;
    const int n = (int)(u1 - u0);
    if (n <= 0) return;
    const int nk = p.nk;

    // per-thread byte offsets inside a tile's operand panels (constant for the whole kernel)
    int voa[PA], vob[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) voa[i] = ((row8 + 32 * i) * p.K + 4 * q8) * 4;
#pragma unroll
    for (int i = 0; i < PB; ++i) vob[i] = ((row8 + 32 * i) * p.K + 4 * q8) * 4;

    // ---- loader cursor: tile (member, tm, tn) and chunk of the NEXT chunk to fetch ----
    const int tiles_mn = p.tiles_m * p.tiles_n;
    int ld_kc, ld_b, ld_tm, ld_tn;
    {
        const long t = u0 / nk;
        ld_kc = (int)(u0 - t * nk);
        ld_b = (int)(t / tiles_mn);
        const int r = (int)(t - (long)ld_b * tiles_mn);
        ld_tm = r / p.tiles_n;
        ld_tn = r - ld_tm * p.tiles_n;
    }
    // compute cursor starts at the same place
    int cp_kc = ld_kc, cp_b = ld_b, cp_tm = ld_tm, cp_tn = ld_tn;
    const bool first_is_tail = (ld_kc != 0);      // the first tile began in the previous workgroup's range

    __amdgpu_buffer_rsrc_t rA, rB;
    int soff = 0;
    auto set_tile = [&]() {
        const int m0 = ld_tm * BM, n0 = ld_tn * BN;
        const int rows_a = min(BM, p.M - m0), rows_b = min(BN, p.N - n0);
        rA = make_rsrc(p.A + (long)ld_b * p.sA + (long)m0 * p.K, (unsigned)(rows_a * p.K * 4));
        rB = make_rsrc(p.B + (long)ld_b * p.sB + (long)n0 * p.K, (unsigned)(rows_b * p.K * 4));
    };
    set_tile();
    auto prep = [&]() { soff = ld_kc * (BK * 4); };
    auto advance = [&]() {
        if (++ld_kc == nk) {
            ld_kc = 0;
            if (++ld_tn == p.tiles_n) {
                ld_tn = 0;
                if (++ld_tm == p.tiles_m) {
                    ld_tm = 0;
                    ++ld_b;
                }
            }
            if (ld_b < p.batch) set_tile();
        }
    };
    f32x4 ra[PA], rb[PB];
    auto load_a = [&](int i) {
        ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, voa[i], soff, 0));
    };
    auto load_b = [&](int i) {
        rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB, vob[i], soff, 0));
    };
    auto write_a = [&](float* dA, int i) { *(f32x4*)(dA + (row8 + 32 * i) * LDK + 4 * q8) = ra[i]; };
    auto write_b = [&](float* dB, int i) { *(f32x4*)(dB + (row8 + 32 * i) * LDK + 4 * q8) = rb[i]; };
    auto load_chunk = [&]() {
        prep();
#pragma unroll
        for (int i = 0; i < PA; ++i) load_a(i);
#pragma unroll
        for (int i = 0; i < PB; ++i) load_b(i);
        advance();
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PA; ++i) write_a(sA + buf * SZA, i);
#pragma unroll
        for (int i = 0; i < PB; ++i) write_b(sB + buf * SZB, i);
    };

    f32x16 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    zero_acc();

    const int fa = (wm * TM * 32 + li) * LDK + 4 * lh;
    const int fb = (wn * TN * 32 + li) * LDK + 4 * lh;
    auto load_frags = [&](const float* cA, const float* cB, int kb, float (&av)[TM][4], float (&bv)[TN][4]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const f32x4 t = *(const f32x4*)(cA + i * 32 * LDK + kb * 8);
            av[i][0] = t[0]; av[i][1] = t[1]; av[i][2] = t[2]; av[i][3] = t[3];
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const f32x4 t = *(const f32x4*)(cB + j * 32 * LDK + kb * 8);
            bv[j][0] = t[0]; bv[j][1] = t[1]; bv[j][2] = t[2]; bv[j][3] = t[3];
        }
    };

    // ---- tile end: results to C, or the fix-up hand-off --------------------------------------------------------
    // slab element order = the accumulator registers themselves: word ((i*TN + j)*4 + grp)*256 + tid holds 16 bytes, so
    // producer and consumer (same fragment layout) move whole 1 KiB wave rows
    auto slab_rsrc = [&](int owner) { return make_rsrc(p.partial + (long)owner * (BM * BN), (unsigned)(BM * BN * 4)); };
    auto publish_partial = [&]() {
        const __amdgpu_buffer_rsrc_t rs = slab_rsrc(g);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs,
                                                           ((((i * TN + j) * 4 + q) * 256) + tid) * 16, 0, 16 /* sc1 */);
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores
        __syncthreads();
        if (tid == 0)
            __hip_atomic_store((__attribute__((address_space(1))) unsigned*)(p.flags + g), p.epoch, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    };
    auto add_partial = [&]() {
        // the tail part of this tile belongs to workgroup g + 1, which produced it at the very start of its range
        if (wave == 0) {
            bool ok = false;
            for (unsigned spins = 0; spins < (1u << 22); ++spins) {
                const unsigned v = __hip_atomic_load((__attribute__((address_space(1))) unsigned*)(p.flags + g + 1),
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v == p.epoch) {
                    ok = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
            if (!ok && lane == 0) *p.err = 1u;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        const __amdgpu_buffer_rsrc_t rs = slab_rsrc(g + 1);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = __builtin_bit_cast(
                        f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ((((i * TN + j) * 4 + q) * 256) + tid) * 16, 0,
                                                                     16 /* sc1: served by L2, never a stale L1 line */));
                    acc[i][j][4 * q] += v[0]; acc[i][j][4 * q + 1] += v[1];
                    acc[i][j][4 * q + 2] += v[2]; acc[i][j][4 * q + 3] += v[3];
                }
    };
    auto store_tile = [&]() {
        float* out = p.C + (long)cp_b * p.sC;
        const int m0 = cp_tm * BM, n0 = cp_tn * BN;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + (wm * TM + i) * 32 + li;
            if (m >= p.M) continue;
            const long row = (long)m * p.N;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int nn = n0 + (wn * TN + j) * 32 + 8 * q + 4 * lh;
                    if (nn >= p.N) continue;
                    const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                    *(f32x4*)(out + row + nn) = v;
                }
        }
    };
    // called after the MFMAs of unit s: closes the tile if this was its last chunk, or if the range ends inside it
    bool in_first_tile = true;
    auto tile_boundary = [&](bool last_unit) {
        const bool tile_done = (cp_kc == nk - 1);
        if (!tile_done && !last_unit) {
            ++cp_kc;
            return;
        }
        if (tile_done && in_first_tile && first_is_tail) {
            publish_partial();
        } else {
            if (!tile_done) add_partial();       // range ends inside the tile: we hold the head part
            store_tile();
        }
        in_first_tile = false;
        zero_acc();
        cp_kc = 0;
        if (++cp_tn == p.tiles_n) {
            cp_tn = 0;
            if (++cp_tm == p.tiles_m) {
                cp_tm = 0;
                ++cp_b;
            }
        }
    };

    // ---- pipelined stream over the n units of this workgroup -----------------------------------------------------
    float av[2][TM][4], bv[2][TN][4];
    load_chunk();                  // unit 0 -> registers -> LDS buffer 0
    store_chunk(0);
    if (n > 1) load_chunk();       // unit 1 -> registers
    __syncthreads();
    load_frags(sA + fa, sB + fb, 0, av[0], bv[0]);
    int cur = 0;
    auto body = [&](auto WF, auto LF, auto NF) {
        constexpr bool do_w = decltype(WF)::value, do_l = decltype(LF)::value, has_next = decltype(NF)::value;
        const int nxt = cur ^ 1;
        const float* cA = sA + cur * SZA + fa;
        const float* cB = sB + cur * SZB + fb;
        float* wA = sA + nxt * SZA;
        float* wB = sB + nxt * SZB;
        if (do_l) prep();
        auto step = [&](auto KBI) {
            constexpr int kb = decltype(KBI)::value;
            constexpr bool rd = (kb + 1 < KB);
            if (rd) load_frags(cA, cB, kb + 1, av[(kb + 1) & 1], bv[(kb + 1) & 1]);
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
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[kb & 1][j][t], av[kb & 1][i][t], acc[i][j], 0, 0, 0);
            constexpr int NM = 4 * TM * TN;
            constexpr int NR = rd ? TM + TN : 0;
            constexpr int NW = do_w ? ((kb < PA) ? 1 : 0) + ((kb < PB) ? 1 : 0) : 0;
            constexpr int HALF = (NM / 2 > 0) ? NM / 2 : 1;
            constexpr int RP = NR ? (NR + HALF - 1) / HALF : 1;
            constexpr int RS = NR ? (NR + RP - 1) / RP : 0;
            constexpr int WS = (NW < NM - RS) ? NW : ((NM - RS > 0) ? NM - RS : 0);
#pragma unroll
            for (int q = 0; q < RS; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, RP, 0);
            }
#pragma unroll
            for (int q = 0; q < WS; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                if (do_l) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
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
        if (has_next) load_frags(sA + nxt * SZA + fa, sB + nxt * SZB + fb, 0, av[0], bv[0]);
        cur = nxt;
    };
    {
        using T = std::true_type;
        using F = std::false_type;
        int s = 0;
        for (; s + 2 < n; ++s) {
            body(T{}, T{}, T{});
            tile_boundary(false);
        }
        for (; s + 1 < n; ++s) {
            body(T{}, F{}, T{});
            tile_boundary(false);
        }
        for (; s < n; ++s) {
            body(F{}, F{}, F{});
            tile_boundary(true);
        }
    }
}

int g_occ[2] = {0, 0};       // resident workgroups per CU of the two instantiations
int g_cus = 0;
unsigned g_epoch = 0;

template <int BM, int BN>
int launch_bgemm(BgemmParams& p, int slot, int wg_per_cu, hipStream_t stream) {
    constexpr size_t lds = 2 * (size_t)(BM + BN) * LDK * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)bgemm_kernel<BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            denet_set_error("bgemm: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
            return -(int)e;
        }
        int occ = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)bgemm_kernel<BM, BN>, 256, lds) != hipSuccess || occ < 1) {
            denet_set_error("bgemm: occupancy query failed");
            return DENET_ERR_ARG;
        }
        g_occ[slot] = occ < 2 ? occ : 2;
        g_cus = prop.multiProcessorCount;
        attr_set = true;
    }
    const int occ = wg_per_cu > 0 && wg_per_cu < g_occ[slot] ? wg_per_cu : g_occ[slot];
    const long tiles = (long)p.batch * p.tiles_m * p.tiles_n;
    long G = (long)g_cus * occ;
    if (G > tiles) G = tiles;            // at least one whole tile of chunks per workgroup: a tile is cut at most once
    p.G = (int)G;
    p.units = tiles * p.nk;
    hipLaunchKernelGGL((bgemm_kernel<BM, BN>), dim3((unsigned)G), dim3(256), lds, stream, p);
    DENET_CHECK_LAUNCH("bgemm");
    return DENET_OK;
}

}  // namespace

// workspace of denet_bgemm: flags [4095] u32 (zeroed ONCE by the caller when the buffer is created), error word [4095] | 512
// fix-up slabs of 64 KiB. A multiple of 4 KiB: what the caller places behind it keeps its alignment.
size_t denet_bgemm_workspace_bytes() { return (size_t)16384 + (size_t)512 * 128 * 128 * sizeof(float); }

// tile: 0 = 128x128, 1 = 128x64; wg_per_cu: 0 = as many as fit (2)
int denet_bgemm(const float* a, const float* b, float* c, int batch, int M, int N, int K, long stride_a, long stride_b,
                long stride_c, void* workspace, size_t workspace_bytes, int tile, int wg_per_cu, hipStream_t stream) {
    DENET_CHECK_ARG(a && b && c && workspace && batch > 0 && M > 0, "bgemm: bad arguments");
    DENET_CHECK_ARG(N % 32 == 0 && K % 32 == 0 && N > 0 && K > 0, "bgemm: N, K must be multiples of 32");
    DENET_CHECK_ARG((long)128 * K * 4 < (1L << 31), "bgemm: K too large");
    DENET_CHECK_ARG(workspace_bytes >= denet_bgemm_workspace_bytes(), "bgemm: workspace too small");
    BgemmParams p = {};
    p.A = a; p.B = b; p.C = c;
    p.M = M; p.N = N; p.K = K; p.batch = batch;
    p.sA = stride_a; p.sB = stride_b; p.sC = stride_c;
    p.nk = K / BK;
    p.tiles_m = ceil_div(M, 128);
    p.flags = (unsigned*)workspace;
    p.err = p.flags + 4095;
    p.partial = (float*)((char*)workspace + 16384);
    p.epoch = ++g_epoch;
    if (p.epoch == 0) p.epoch = ++g_epoch;
    if (tile == 0 && N >= 128) {
        p.tiles_n = ceil_div(N, 128);
        return launch_bgemm<128, 128>(p, 0, wg_per_cu, stream);
    }
    p.tiles_n = ceil_div(N, 64);
    return launch_bgemm<128, 64>(p, 1, wg_per_cu, stream);
}
