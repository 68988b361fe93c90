// Experimental harness (development aid, not part of the product): C[M,N] = A[M,K] * B[N,K]^T, fp32 MFMA.
// Ablation variants of the igemm main loop to find what bounds it.  hipcc --offload-arch=gfx950 -O3 gemm_exp.hip -o gemm_exp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int BK = 32, LDK = 36;

// VAR: 0 baseline, 1 no global loads in loop, 2 also no LDS writes, 3 also no barriers, 4 MFMA only
// 5: 4 without the prologue chunk load, 6: 4 without the epilogue stores, 7: neither (pure MFMA + launch)
template <int BM, int BN, int WM, int WN, int NBUF, int VAR>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                      float* __restrict__ C, int M, int N, int K) {
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int SZA = BM * LDK, SZB = BN * LDK;
    constexpr int PA = BM / 32, PB = BN / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;
    float* sB = smem + NBUF * SZA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;
    const int tiles_n = N / BN;
    const int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int q8 = tid & 7, row8 = tid >> 3;
    const float* pa[PA];
    const float* pb[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) pa[i] = A + (long)(m0 + row8 + 32 * i) * K + 4 * q8;
#pragma unroll
    for (int i = 0; i < PB; ++i) pb[i] = B + (long)(n0 + row8 + 32 * i) * K + 4 * q8;
    f32x4 ra[PA], rb[PB];
    auto load_chunk = [&](int kc) {
#pragma unroll
        for (int i = 0; i < PA; ++i) ra[i] = *(const f32x4*)(pa[i] + kc * BK);
#pragma unroll
        for (int i = 0; i < PB; ++i) rb[i] = *(const f32x4*)(pb[i] + kc * BK);
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PA; ++i) *(f32x4*)(sA + buf * SZA + (row8 + 32 * i) * LDK + 4 * q8) = ra[i];
#pragma unroll
        for (int i = 0; i < PB; ++i) *(f32x4*)(sB + buf * SZB + (row8 + 32 * i) * LDK + 4 * q8) = rb[i];
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
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
    float cav[TM][4], cbv[TN][4];
    if (VAR >= 4) {
#pragma unroll
        for (int i = 0; i < TM; ++i) for (int t = 0; t < 4; ++t) cav[i][t] = A[lane + i + t];
#pragma unroll
        for (int j = 0; j < TN; ++j) for (int t = 0; t < 4; ++t) cbv[j][t] = B[lane + j + t];
    }
    auto compute = [&](int buf) {
        const float* cA = sA + buf * SZA + fa;
        const float* cB = sB + buf * SZB + fb;
        float av[2][TM][4], bv[2][TN][4];
        if (VAR < 4) load_frags(cA, cB, 0, av[0], bv[0]);
#pragma unroll
        for (int kb = 0; kb < BK / 8; ++kb) {
            if (VAR < 4) {
                if (kb + 1 < BK / 8) load_frags(cA, cB, kb + 1, av[(kb + 1) & 1], bv[(kb + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(VAR >= 4 ? cav[i][t] : av[kb & 1][i][t],
                                                                         VAR >= 4 ? cbv[j][t] : bv[kb & 1][j][t], acc[i][j], 0, 0, 0);
        }
    };
    const int nsteps = K / BK;
    if (VAR != 5 && VAR != 7) {
        load_chunk(0);
        store_chunk(0);
        __syncthreads();
    }
    for (int s = 0; s < nsteps; ++s) {
        const bool more = (s + 1 < nsteps);
        if (VAR == 0 && more) load_chunk(s + 1);
        if (NBUF == 2) {
            compute(s & 1);
            if (VAR <= 1 && more) store_chunk((s + 1) & 1);
            if (VAR <= 2) __syncthreads();
        } else {
            compute(0);
            if (VAR <= 2) __syncthreads();
            if (VAR <= 1 && more) store_chunk(0);
            if (VAR <= 2) __syncthreads();
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if ((VAR != 6 && VAR != 7) || acc[i][j][r] == 123.456f) C[(long)m * N + n] = acc[i][j][r];
            }
        }
}


// ---- pipelined variant: 3 LDS buffers, write-after-barrier (chunk s+2), global loads re-issued right after the
// write (chunk s+3), fragment prefetch of chunk s+1 issued before the last MFMA block so nothing is exposed at the barrier
template <int BM, int BN, int WM, int WN, int OCC, int PD>
__global__ __launch_bounds__(256, OCC) void gemm_pipe(const float* __restrict__ A, const float* __restrict__ B,
                                                       float* __restrict__ C, int M, int N, int K) {
    constexpr int NB = 3;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int SZA = BM * LDK, SZB = BN * LDK;
    constexpr int PA = BM / 32, PB = BN / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;
    float* sB = smem + NB * SZA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;
    const int tiles_n = N / BN;
    const int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int q8 = tid & 7, row8 = tid >> 3;
    const float* pa[PA];
    const float* pb[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) pa[i] = A + (long)(m0 + row8 + 32 * i) * K + 4 * q8;
#pragma unroll
    for (int i = 0; i < PB; ++i) pb[i] = B + (long)(n0 + row8 + 32 * i) * K + 4 * q8;
    f32x4 ra[PD][PA], rb[PD][PB];
    auto load_chunk = [&](int kc, int set) {
#pragma unroll
        for (int i = 0; i < PA; ++i) ra[set][i] = *(const f32x4*)(pa[i] + kc * BK);
#pragma unroll
        for (int i = 0; i < PB; ++i) rb[set][i] = *(const f32x4*)(pb[i] + kc * BK);
    };
    auto store_chunk = [&](int buf, int set) {
#pragma unroll
        for (int i = 0; i < PA; ++i) *(f32x4*)(sA + buf * SZA + (row8 + 32 * i) * LDK + 4 * q8) = ra[set][i];
#pragma unroll
        for (int i = 0; i < PB; ++i) *(f32x4*)(sB + buf * SZB + (row8 + 32 * i) * LDK + 4 * q8) = rb[set][i];
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int fa = (wm * TM * 32 + li) * LDK + 4 * lh;
    const int fb = (wn * TN * 32 + li) * LDK + 4 * lh;
    auto load_frags = [&](int buf, int kb, float (&av)[TM][4], float (&bv)[TN][4]) {
        const float* cA = sA + buf * SZA + fa;
        const float* cB = sB + buf * SZB + fb;
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
    const int nsteps = K / BK;
    float av[2][TM][4], bv[2][TN][4];
    // prologue: chunks 0,1 -> LDS; chunk 2 (and 3 when PD == 2) in flight in registers
    load_chunk(0, 0);
    store_chunk(0, 0);
    if (nsteps > 1) { load_chunk(1, 0); store_chunk(1, 0); }
    if (nsteps > 2) load_chunk(2, 0);
    if (PD == 2 && nsteps > 3) load_chunk(3, 1);
    __syncthreads();
    load_frags(0, 0, av[0], bv[0]);
    int cur = 0;   // buffer of chunk s
    auto body = [&](int s, int set) {
        const int nxt = (cur == 2) ? 0 : cur + 1;       // chunk s+1
        const int nx2 = (nxt == 2) ? 0 : nxt + 1;       // chunk s+2
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            if (kb < 3) load_frags(cur, kb + 1, av[(kb + 1) & 1], bv[(kb + 1) & 1]);
            else if (s + 1 < nsteps) load_frags(nxt, 0, av[0], bv[0]);
            if (kb == 0 && s + 2 < nsteps) store_chunk(nx2, set);               // chunk s+2 (register set `set`)
            if (kb == 1 && s + 2 + PD < nsteps) load_chunk(s + 2 + PD, set);    // refill the same set
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kb & 1][i][t], bv[kb & 1][j][t], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        cur = nxt;
    };
    if (PD == 1) {
        for (int s = 0; s < nsteps; ++s) body(s, 0);
    } else {
        int s = 0;
        for (; s + 1 < nsteps; s += 2) { body(s, 0); body(s + 1, 1); }
        if (s < nsteps) body(s, 0);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                C[(long)m * N + n] = acc[i][j][r];
            }
        }
}

template <int BM, int BN, int OCC, int PD>
float run_pipe(const float* A, const float* B, float* C, int M, int N, int K, int iters) {
    constexpr size_t lds = 3 * (size_t)(BM + BN) * LDK * sizeof(float);
    CK(hipFuncSetAttribute((const void*)gemm_pipe<BM, BN, 2, 2, OCC, PD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((M / BM) * (N / BN));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    gemm_pipe<BM, BN, 2, 2, OCC, PD><<<grid, 256, lds>>>(A, B, C, M, N, K);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) gemm_pipe<BM, BN, 2, 2, OCC, PD><<<grid, 256, lds>>>(A, B, C, M, N, K);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

// ---- fine-interleave variant: NB LDS buffers (2 or 3), one register set, write-then-reload per quarter (T14 order),
// every non-MFMA instruction is slotted behind an MFMA with sched_group_barrier so the matrix pipe never drains
template <int BM, int BN, int WM, int WN, int OCC, int NB, int IL>
__global__ __launch_bounds__(256, OCC) void gemm_il(const float* __restrict__ A, const float* __restrict__ B,
                                                     float* __restrict__ C, int M, int N, int K) {
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int SZA = BM * LDK, SZB = BN * LDK;
    constexpr int PA = BM / 32, PB = BN / 32;
    static_assert(PA <= 4 && PB <= 4, "quarters");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;
    float* sB = smem + NB * SZA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;
    const int tiles_n = N / BN;
    const int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int q8 = tid & 7, row8 = tid >> 3;
    const float* pa[PA];
    const float* pb[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) pa[i] = A + (long)(m0 + row8 + 32 * i) * K + 4 * q8;
#pragma unroll
    for (int i = 0; i < PB; ++i) pb[i] = B + (long)(n0 + row8 + 32 * i) * K + 4 * q8;
    f32x4 ra[PA], rb[PB];
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int fa = (wm * TM * 32 + li) * LDK + 4 * lh;
    const int fb = (wn * TN * 32 + li) * LDK + 4 * lh;
    auto load_frags = [&](int buf, int kb, float (&av)[TM][4], float (&bv)[TN][4]) {
        const float* cA = sA + buf * SZA + fa;
        const float* cB = sB + buf * SZB + fb;
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
    const int nsteps = K / BK;
    float av[2][TM][4], bv[2][TN][4];
    // prologue
#pragma unroll
    for (int i = 0; i < PA; ++i) ra[i] = *(const f32x4*)(pa[i]);
#pragma unroll
    for (int i = 0; i < PB; ++i) rb[i] = *(const f32x4*)(pb[i]);
#pragma unroll
    for (int i = 0; i < PA; ++i) *(f32x4*)(sA + (row8 + 32 * i) * LDK + 4 * q8) = ra[i];
#pragma unroll
    for (int i = 0; i < PB; ++i) *(f32x4*)(sB + (row8 + 32 * i) * LDK + 4 * q8) = rb[i];
    if (NB == 3 && nsteps > 1) {
#pragma unroll
        for (int i = 0; i < PA; ++i) ra[i] = *(const f32x4*)(pa[i] + BK);
#pragma unroll
        for (int i = 0; i < PB; ++i) rb[i] = *(const f32x4*)(pb[i] + BK);
#pragma unroll
        for (int i = 0; i < PA; ++i) *(f32x4*)(sA + SZA + (row8 + 32 * i) * LDK + 4 * q8) = ra[i];
#pragma unroll
        for (int i = 0; i < PB; ++i) *(f32x4*)(sB + SZB + (row8 + 32 * i) * LDK + 4 * q8) = rb[i];
    }
    constexpr int AH = NB - 1;     // chunk written during iteration s: s + AH
    if (nsteps > AH) {
#pragma unroll
        for (int i = 0; i < PA; ++i) ra[i] = *(const f32x4*)(pa[i] + AH * BK);
#pragma unroll
        for (int i = 0; i < PB; ++i) rb[i] = *(const f32x4*)(pb[i] + AH * BK);
    }
    __syncthreads();
    load_frags(0, 0, av[0], bv[0]);
    int cur = 0;
    auto body = [&](int s, auto WF, auto LF, auto NF) {
        constexpr bool do_w = decltype(WF)::value, do_l = decltype(LF)::value, has_next = decltype(NF)::value;
        const int nxt = (cur == NB - 1) ? 0 : cur + 1;            // buffer of chunk s+1
        const int wbuf = (NB == 2) ? nxt : ((nxt == NB - 1) ? 0 : nxt + 1);   // buffer of chunk s+AH
        float* wA = sA + wbuf * SZA + row8 * LDK + 4 * q8;
        float* wB = sB + wbuf * SZB + row8 * LDK + 4 * q8;
        const int koff = (s + AH + 1) * BK;
        auto step = [&](auto KB) {
            constexpr int kb = decltype(KB)::value;
            if (kb < 3) load_frags(cur, kb + 1, av[(kb + 1) & 1], bv[(kb + 1) & 1]);
            else if (NB == 3 && has_next) load_frags(nxt, 0, av[0], bv[0]);
            if (kb < PA) {
                if (do_w) *(f32x4*)(wA + 32 * kb * LDK) = ra[kb];
                if (do_l) ra[kb] = *(const f32x4*)(pa[kb] + koff);
            }
            if (kb < PB) {
                if (do_w) *(f32x4*)(wB + 32 * kb * LDK) = rb[kb];
                if (do_l) rb[kb] = *(const f32x4*)(pb[kb] + koff);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kb & 1][i][t], bv[kb & 1][j][t], acc[i][j], 0, 0, 0);
            if (IL >= 1) {
                constexpr int NR = (kb < 3 || (NB == 3 && has_next)) ? TM + TN : 0;
#pragma unroll
                for (int g = 0; g < NR; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                constexpr int NW = do_w ? 2 : 0;
#pragma unroll
                for (int g = 0; g < NW; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    if (do_l) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 16 - NR - NW * (do_l ? 2 : 1), 0);
            }
            if (IL == 2) __builtin_amdgcn_sched_barrier(0);
        };
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
        __syncthreads();
        if (NB == 2 && has_next) load_frags(nxt, 0, av[0], bv[0]);
        cur = nxt;
    };
    {
        using T = std::true_type; using F = std::false_type;
        int s = 0;
        for (; s + AH + 1 < nsteps; ++s) body(s, T{}, T{}, T{});
        for (; s + AH < nsteps; ++s) body(s, T{}, F{}, T{});
        for (; s + 1 < nsteps; ++s) body(s, F{}, F{}, T{});
        for (; s < nsteps; ++s) body(s, F{}, F{}, F{});
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                C[(long)m * N + n] = acc[i][j][r];
            }
        }
}

template <int BM, int BN, int OCC, int NB, int IL>
float run_il(const float* A, const float* B, float* C, int M, int N, int K, int iters) {
    constexpr size_t lds = NB * (size_t)(BM + BN) * LDK * sizeof(float);
    CK(hipFuncSetAttribute((const void*)gemm_il<BM, BN, 2, 2, OCC, NB, IL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((M / BM) * (N / BN));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    gemm_il<BM, BN, 2, 2, OCC, NB, IL><<<grid, 256, lds>>>(A, B, C, M, N, K);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) gemm_il<BM, BN, 2, 2, OCC, NB, IL><<<grid, 256, lds>>>(A, B, C, M, N, K);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

// ---- persistent variant of gemm_il: a workgroup walks tiles w, w+G, w+2G, ... and treats their K chunks as ONE stream:
// the staging pipeline (LDS ring, register prefetch) never drains at a tile boundary, only the accumulators are stored
template <int BM, int BN, int WM, int WN, int OCC, int NB>
__global__ __launch_bounds__(256, OCC) void gemm_pers(const float* __restrict__ A, const float* __restrict__ B,
                                                       float* __restrict__ C, int M, int N, int K) {
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int SZA = BM * LDK, SZB = BN * LDK;
    constexpr int PA = BM / 32, PB = BN / 32;
    static_assert(PA <= 4 && PB <= 4, "quarters");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;
    float* sB = smem + NB * SZA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;
    const int tiles_n = N / BN, ntiles = (M / BM) * tiles_n;
    const int ksteps = K / BK;
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    if (my_tiles <= 0) return;
    const int total = my_tiles * ksteps;
    const int q8 = tid & 7, row8 = tid >> 3;
    // load cursor
    int l_tile = blockIdx.x, l_k = 0;
    const float* pa[PA];
    const float* pb[PB];
    auto set_ptrs = [&](int tile) {
        const int tn = tile % tiles_n, tmm = tile / tiles_n;
#pragma unroll
        for (int i = 0; i < PA; ++i) pa[i] = A + (long)(tmm * BM + row8 + 32 * i) * K + 4 * q8;
#pragma unroll
        for (int i = 0; i < PB; ++i) pb[i] = B + (long)(tn * BN + row8 + 32 * i) * K + 4 * q8;
    };
    set_ptrs(l_tile);
    auto advance_load = [&]() {
        if (++l_k == ksteps) {
            l_k = 0;
            l_tile += gridDim.x;
            if (l_tile < ntiles) set_ptrs(l_tile);
        }
    };
    f32x4 ra[PA], rb[PB];
    auto load_all = [&]() {
#pragma unroll
        for (int i = 0; i < PA; ++i) ra[i] = *(const f32x4*)(pa[i] + l_k * BK);
#pragma unroll
        for (int i = 0; i < PB; ++i) rb[i] = *(const f32x4*)(pb[i] + l_k * BK);
    };
    auto store_all = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PA; ++i) *(f32x4*)(sA + buf * SZA + (row8 + 32 * i) * LDK + 4 * q8) = ra[i];
#pragma unroll
        for (int i = 0; i < PB; ++i) *(f32x4*)(sB + buf * SZB + (row8 + 32 * i) * LDK + 4 * q8) = rb[i];
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
    auto load_frags = [&](int buf, int kb, float (&av)[TM][4], float (&bv)[TN][4]) {
        const float* cA = sA + buf * SZA + fa;
        const float* cB = sB + buf * SZB + fb;
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
    float av[2][TM][4], bv[2][TN][4];
    constexpr int AH = NB - 1;
    // prologue: chunks 0..AH-1 -> LDS, chunk AH -> registers
    load_all(); advance_load(); store_all(0);
    if (NB == 3 && total > 1) { load_all(); advance_load(); store_all(1); }
    if (total > AH) { load_all(); advance_load(); }
    __syncthreads();
    load_frags(0, 0, av[0], bv[0]);
    int cur = 0, c_tile = blockIdx.x, c_k = 0;
    auto body = [&](auto WF, auto LF, auto NF) {
        constexpr bool do_w = decltype(WF)::value, do_l = decltype(LF)::value, has_next = decltype(NF)::value;
        const int nxt = (cur == NB - 1) ? 0 : cur + 1;
        const int wbuf = (NB == 2) ? nxt : ((nxt == NB - 1) ? 0 : nxt + 1);
        float* wA = sA + wbuf * SZA + row8 * LDK + 4 * q8;
        float* wB = sB + wbuf * SZB + row8 * LDK + 4 * q8;
        auto step = [&](auto KB) {
            constexpr int kb = decltype(KB)::value;
            if (kb < 3) load_frags(cur, kb + 1, av[(kb + 1) & 1], bv[(kb + 1) & 1]);
            else if (NB == 3 && has_next) load_frags(nxt, 0, av[0], bv[0]);
            if constexpr (kb < PA) {
                if (do_w) *(f32x4*)(wA + 32 * kb * LDK) = ra[kb];
                if (do_l) ra[kb] = *(const f32x4*)(pa[kb] + l_k * BK);
            }
            if constexpr (kb < PB) {
                if (do_w) *(f32x4*)(wB + 32 * kb * LDK) = rb[kb];
                if (do_l) rb[kb] = *(const f32x4*)(pb[kb] + l_k * BK);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kb & 1][i][t], bv[kb & 1][j][t], acc[i][j], 0, 0, 0);
            constexpr int NM = 4 * TM * TN;
            constexpr int NR = (kb < 3 || (NB == 3 && has_next)) ? TM + TN : 0;
            constexpr int NW = do_w ? ((kb < PA) ? 1 : 0) + ((kb < PB) ? 1 : 0) : 0;
#pragma unroll
            for (int g = 0; g < NR; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int g = 0; g < NW; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                if (do_l) {
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            }
            if constexpr (NM - NR - NW > 0) __builtin_amdgcn_sched_group_barrier(0x008, NM - NR - NW, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
        if (do_l) advance_load();
        if (++c_k == ksteps) {          // tile finished: store and clear the accumulators, the staging keeps streaming
            const int tn = c_tile % tiles_n, tmm = c_tile / tiles_n;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = tn * BN + (wn * TN + j) * 32 + li;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = tmm * BM + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        C[(long)m * N + n] = acc[i][j][r];
                    }
                }
            zero_acc();
            c_k = 0;
            c_tile += gridDim.x;
        }
        __syncthreads();
        if (NB == 2 && has_next) load_frags(nxt, 0, av[0], bv[0]);
        cur = nxt;
    };
    {
        using T = std::true_type; using F = std::false_type;
        int s = 0;
        for (; s + AH + 1 < total; ++s) body(T{}, T{}, T{});
        for (; s + AH < total; ++s) body(T{}, F{}, T{});
        for (; s + 1 < total; ++s) body(F{}, F{}, T{});
        for (; s < total; ++s) body(F{}, F{}, F{});
    }
}

template <int BM, int BN, int OCC, int NB>
float run_pers(const float* A, const float* B, float* C, int M, int N, int K, int iters, int wgs_per_cu) {
    constexpr size_t lds = NB * (size_t)(BM + BN) * LDK * sizeof(float);
    CK(hipFuncSetAttribute((const void*)gemm_pers<BM, BN, 2, 2, OCC, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int ntiles = (M / BM) * (N / BN);
    int g = 256 * wgs_per_cu; if (g > ntiles) g = ntiles;
    dim3 grid(g);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    gemm_pers<BM, BN, 2, 2, OCC, NB><<<grid, 256, lds>>>(A, B, C, M, N, K);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) gemm_pers<BM, BN, 2, 2, OCC, NB><<<grid, 256, lds>>>(A, B, C, M, N, K);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

template <int BM, int BN, int WM, int WN, int NBUF, int VAR>
float run(const float* A, const float* B, float* C, int M, int N, int K, int iters) {
    constexpr size_t lds = NBUF * (size_t)(BM + BN) * LDK * sizeof(float);
    CK(hipFuncSetAttribute((const void*)gemm_kernel<BM, BN, WM, WN, NBUF, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((M / BM) * (N / BN));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    gemm_kernel<BM, BN, WM, WN, NBUF, VAR><<<grid, 256, lds>>>(A, B, C, M, N, K);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) gemm_kernel<BM, BN, WM, WN, NBUF, VAR><<<grid, 256, lds>>>(A, B, C, M, N, K);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

// ---- direct-to-LDS variant (gfx950 global_load_lds_dwordx4): no staging VGPRs, no ds_write. LDS rows are 128 B, unpadded
// (the DMA writes wave-linear: 64 lanes x 16 B = 8 rows); bank conflicts are avoided by an XOR swizzle of the 16-byte chunk
// index with (row & 7), applied to the per-lane GLOBAL source address and to the fragment read address.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
template <int BM, int BN, int OCC, int NB>
__global__ __launch_bounds__(256, OCC) void gemm_glds(const float* __restrict__ A, const float* __restrict__ B,
                                                       float* __restrict__ C, int M, int N, int K) {
    constexpr int WM = 2, WN = 2;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int SZA = BM * 32, SZB = BN * 32, SZ = SZA + SZB;
    constexpr int IA = BM / 32, IB = BN / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;
    const int tiles_n = N / BN;
    const int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int lr = lane >> 3, lj = lane & 7;
    const float* ga[IA];
    const float* gb[IB];
#pragma unroll
    for (int i = 0; i < IA; ++i) {
        const int row = wave * (BM / 4) + i * 8 + lr;
        ga[i] = A + (long)(m0 + row) * K + ((lj ^ (row & 7)) * 4);
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
        const int row = wave * (BN / 4) + i * 8 + lr;
        gb[i] = B + (long)(n0 + row) * K + ((lj ^ (row & 7)) * 4);
    }
    auto stage = [&](int buf, int kc) {
        float* sa = smem + buf * SZ;
        float* sb = sa + SZA;
#pragma unroll
        for (int i = 0; i < IA; ++i)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(ga[i] + kc * 32), (lds_ptr_t)(sa + (wave * (BM / 4) + i * 8) * 32), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < IB; ++i)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(gb[i] + kc * 32), (lds_ptr_t)(sb + (wave * (BN / 4) + i * 8) * 32), 16, 0, 0);
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int swz = li & 7;
    const int fa = (wm * TM * 32 + li) * 32, fb = (wn * TN * 32 + li) * 32;
    auto load_frags = [&](const float* sa, const float* sb, int kb, f32x4 (&av)[TM], f32x4 (&bv)[TN]) {
        const int off = ((kb * 2 + lh) ^ swz) * 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) av[i] = *(const f32x4*)(sa + fa + i * 32 * 32 + off);
#pragma unroll
        for (int j = 0; j < TN; ++j) bv[j] = *(const f32x4*)(sb + fb + j * 32 * 32 + off);
    };
    auto compute = [&](int buf) {
        const float* sa = smem + buf * SZ;
        const float* sb = sa + SZA;
        f32x4 av[2][TM], bv[2][TN];
        load_frags(sa, sb, 0, av[0], bv[0]);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            if (kb + 1 < 4) load_frags(sa, sb, kb + 1, av[(kb + 1) & 1], bv[(kb + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kb & 1][i][t], bv[kb & 1][j][t], acc[i][j], 0, 0, 0);
        }
    };
    const int nsteps = K / 32;
    if (NB == 2) {
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int s = 0; s < nsteps; ++s) {
            if (s + 1 < nsteps) stage((s + 1) & 1, s + 1);
            compute(s & 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else {
        stage(0, 0);
        if (nsteps > 1) stage(1, 1);
        int cur = 0, nxt = 2;
        for (int s = 0; s < nsteps; ++s) {
            if (s + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IA + IB) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (s + 2 < nsteps) stage(nxt, s + 2);
            compute(cur);
            cur = cur == 2 ? 0 : cur + 1;
            nxt = nxt == 2 ? 0 : nxt + 1;
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                C[(long)m * N + n] = acc[i][j][r];
            }
        }
}

template <int BM, int BN, int OCC, int NB>
float run_glds(const float* A, const float* B, float* C, int M, int N, int K, int iters) {
    constexpr size_t lds = NB * (size_t)(BM + BN) * 32 * sizeof(float);
    CK(hipFuncSetAttribute((const void*)gemm_glds<BM, BN, OCC, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((M / BM) * (N / BN));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    gemm_glds<BM, BN, OCC, NB><<<grid, 256, lds>>>(A, B, C, M, N, K);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) gemm_glds<BM, BN, OCC, NB><<<grid, 256, lds>>>(A, B, C, M, N, K);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 18432, K = argc > 2 ? atoi(argv[2]) : 1536, N = argc > 3 ? atoi(argv[3]) : 1024;
    float *A, *B, *C;
    CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&B, (size_t)N * K * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
    std::vector<float> h((size_t)M * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    CK(hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice));
    const double flop = 2.0 * M * N * K;
#define RUNG(BM, BN, OCC, NB) { float ms = run_glds<BM, BN, OCC, NB>(A, B, C, M, N, K, 10); printf("glds %dx%d occ %d nb %d : %.3f ms %.1f TF\n", BM, BN, OCC, NB, ms, flop / ms / 1e9); fflush(stdout); }
    if (N % 128 == 0) { RUNG(128, 128, 2, 2) RUNG(128, 128, 1, 3) RUNG(128, 128, 1, 2) }
    RUNG(128, 64, 2, 2) RUNG(128, 64, 3, 2) RUNG(128, 64, 2, 3)
    if (N % 128 == 0) {
        std::vector<float> c1((size_t)256 * N), c2((size_t)256 * N);
        run<128, 128, 2, 2, 1, 0>(A, B, C, M, N, K, 1);
        CK(hipMemcpy(c1.data(), C + (size_t)(M - 256) * N, c1.size() * 4, hipMemcpyDeviceToHost));
        for (int v = 0; v < 2; ++v) {
            CK(hipMemset(C, 0, (size_t)M * N * 4));
            if (v == 0) run_glds<128, 128, 2, 2>(A, B, C, M, N, K, 1); else run_glds<128, 64, 2, 3>(A, B, C, M, N, K, 1);
            CK(hipMemcpy(c2.data(), C + (size_t)(M - 256) * N, c2.size() * 4, hipMemcpyDeviceToHost));
            double md = 0; for (size_t i = 0; i < c1.size(); ++i) md = fmax(md, fabs((double)c1[i] - c2[i]));
            printf("max |glds%d - base| (last rows) = %g (ref %g)\n", v, md, (double)c1[7]);
        }
    }
    if (getenv("GLDS_ONLY")) return 0;
#define RUNG(BM, BN, OCC, NB) { float ms = run_glds<BM, BN, OCC, NB>(A, B, C, M, N, K, 10); printf("glds %dx%d occ %d nb %d : %.3f ms %.1f TF\n", BM, BN, OCC, NB, ms, flop / ms / 1e9); fflush(stdout); }
    if (N % 128 == 0) { RUNG(128, 128, 2, 2) RUNG(128, 128, 1, 3) RUNG(128, 128, 1, 2) }
    RUNG(128, 64, 2, 2) RUNG(128, 64, 3, 2) RUNG(128, 64, 2, 3)
    if (N % 128 == 0) {
        std::vector<float> c1((size_t)256 * N), c2((size_t)256 * N);
        run<128, 128, 2, 2, 1, 0>(A, B, C, M, N, K, 1);
        CK(hipMemcpy(c1.data(), C + (size_t)(M - 256) * N, c1.size() * 4, hipMemcpyDeviceToHost));
        for (int v = 0; v < 2; ++v) {
            CK(hipMemset(C, 0, (size_t)M * N * 4));
            if (v == 0) run_glds<128, 128, 2, 2>(A, B, C, M, N, K, 1); else run_glds<128, 64, 2, 3>(A, B, C, M, N, K, 1);
            CK(hipMemcpy(c2.data(), C + (size_t)(M - 256) * N, c2.size() * 4, hipMemcpyDeviceToHost));
            double md = 0; for (size_t i = 0; i < c1.size(); ++i) md = fmax(md, fabs((double)c1[i] - c2[i]));
            printf("max |glds%d - base| (last rows) = %g (ref %g)\n", v, md, (double)c1[7]);
        }
    }
    if (getenv("GLDS_ONLY")) return 0;
#define RUN(BM, BN, NB, V) { float ms = run<BM, BN, 2, 2, NB, V>(A, B, C, M, N, K, 10); printf("tile %dx%d nbuf %d var %d : %.3f ms %.1f TF\n", BM, BN, NB, V, ms, flop / ms / 1e9); fflush(stdout); }
    { float ms = run_pipe<128, 128, 1, 1>(A, B, C, M, N, K, 10); printf("pipe3 128x128 pd1: %.3f ms %.1f TF\n", ms, flop / ms / 1e9); }
    { float ms = run_pipe<128, 128, 1, 2>(A, B, C, M, N, K, 10); printf("pipe3 128x128 pd2: %.3f ms %.1f TF\n", ms, flop / ms / 1e9); }
    { float ms = run_pipe<128, 64, 1, 2>(A, B, C, M, N, K, 10); printf("pipe3 128x64 pd2: %.3f ms %.1f TF\n", ms, flop / ms / 1e9); }
#define RUNP(BM, BN, OCC, NB, W) { float ms = run_pers<BM, BN, OCC, NB>(A, B, C, M, N, K, 10, W); printf("pers %dx%d occ %d nb %d wg/cu %d : %.3f ms %.1f TF\n", BM, BN, OCC, NB, W, ms, flop / ms / 1e9); fflush(stdout); }
    RUNP(128, 128, 2, 2, 2) RUNP(128, 128, 1, 3, 1) RUNP(128, 64, 2, 2, 2) RUNP(128, 64, 3, 2, 3) RUNP(128, 64, 2, 3, 2)
    {
        std::vector<float> c1((size_t)256 * N), c2((size_t)256 * N);
        run<128, 128, 2, 2, 2, 0>(A, B, C, M, N, K, 1);
        CK(hipMemcpy(c1.data(), C + (size_t)(M - 256) * N, c1.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemset(C, 0, (size_t)M * N * 4));
        run_pers<128, 64, 2, 2>(A, B, C, M, N, K, 1, 2);
        CK(hipMemcpy(c2.data(), C + (size_t)(M - 256) * N, c2.size() * 4, hipMemcpyDeviceToHost));
        double md = 0; for (size_t i = 0; i < c1.size(); ++i) md = fmax(md, fabs((double)c1[i] - c2[i]));
        printf("max |pers - base| (last rows) = %g\n", md);
    }
#define RUNIL(BM, BN, OCC, NB, IL) { float ms = run_il<BM, BN, OCC, NB, IL>(A, B, C, M, N, K, 10); printf("il %dx%d occ %d nb %d il %d : %.3f ms %.1f TF\n", BM, BN, OCC, NB, IL, ms, flop / ms / 1e9); fflush(stdout); }
    RUNIL(128, 128, 2, 2, 1) RUNIL(128, 128, 2, 2, 2) RUNIL(128, 128, 1, 3, 1) RUNIL(128, 128, 1, 3, 2) RUNIL(128, 64, 2, 2, 2) RUNIL(128, 64, 3, 2, 2)
    {
        std::vector<float> c1((size_t)256 * N), c2((size_t)256 * N);
        run<128, 128, 2, 2, 2, 0>(A, B, C, M, N, K, 1);
        CK(hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost));
        for (int v = 0; v < 2; ++v) {
            if (v == 0) run_il<128, 128, 2, 2, 1>(A, B, C, M, N, K, 1); else run_il<128, 128, 1, 3, 1>(A, B, C, M, N, K, 1);
            CK(hipMemcpy(c2.data(), C, c2.size() * 4, hipMemcpyDeviceToHost));
            double md = 0; for (size_t i = 0; i < c1.size(); ++i) md = fmax(md, fabs((double)c1[i] - c2[i]));
            printf("max |il%d - base| = %g\n", v, md);
        }
    }
    // correctness spot check vs the baseline kernel
    {
        std::vector<float> c1((size_t)256 * N), c2((size_t)256 * N);
        run<128, 128, 2, 2, 2, 0>(A, B, C, M, N, K, 1);
        CK(hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost));
        run_pipe<128, 128, 1, 2>(A, B, C, M, N, K, 1);
        CK(hipMemcpy(c2.data(), C, c2.size() * 4, hipMemcpyDeviceToHost));
        double md = 0; for (size_t i = 0; i < c1.size(); ++i) md = fmax(md, fabs((double)c1[i] - c2[i]));
        printf("max |pipe - base| = %g (ref %g)\n", md, (double)c1[5]);
    }
    RUN(128, 128, 2, 0) RUN(128, 128, 2, 1) RUN(128, 128, 2, 2) RUN(128, 128, 2, 3) RUN(128, 128, 2, 4)
    RUN(128, 128, 2, 5) RUN(128, 128, 2, 6) RUN(128, 128, 2, 7) RUN(128, 64, 2, 4) RUN(128, 64, 2, 5) RUN(128, 64, 2, 6) RUN(128, 64, 2, 7)
    RUN(128, 128, 1, 0) RUN(128, 128, 1, 1) RUN(128, 128, 1, 2)
    return 0;
}
