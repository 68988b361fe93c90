// The second stage of a batch norm's cross-workgroup reductions INSIDE the launch that produces the partial sums.
//
// A convolution pass leaves, per workgroup, a row of partial column sums [rows][2][C] doubles (forward: sum / sum of squares of
// what it stored - batch_norm.py:50-53; backward: sum(g) / sum(g * xhat) of the gradient it wrote - what tensor.grad of the
// batch norm needs, model_cnn.py:318). Until round 4 a kernel of its own reduced the rows (bn_stats_final_kernel /
// bn_bwd_final_kernel, 84 launches per DeNet-34 step): 5 us of work each, but on the critical chain of the backward sweep such a
// launch waits for a free CU slot beside the other stream's matrix kernels (measured 25-37 us in the step).
// Here the LAST workgroup of a column group to finish does that reduction itself:
//   * every workgroup stores its row with write-through 8-byte stores (bnf_store: global_store_dwordx2 sc1), every storing wave
//     drains them (s_waitcnt vmcnt(0)), the workgroup synchronises, ONE lane takes a ticket with a relaxed agent-scope
//     fetch_add on the group's counter (cdna_hip_programming.md, Guideline 16 / the split-K seam recipe in its sc1 form: no
//     release fence, hence no write-back of the XCD's whole L2 per workgroup);
//   * the workgroup that draws the last ticket reads ALL rows of its columns with sc1 loads (past its own L1 / the XCD's L2) in the
//     order of the separate kernels - one wave per FC channels, 64 / FC row lanes striding over the rows with four loads in
//     flight, a shuffle tree - so the result is BIT-IDENTICAL to what bn_stats_final_kernel / bn_bwd_final_kernel compute from
//     the same rows (tests compare the two forms exactly: a stale row would show as a different bit);
//   * it leaves the counter at zero again: the buffers are at rest between launches (allocated zeroed by the host).
// Which batch norm the sums belong to arrives through an "armed" descriptor (denet_bn_final_arm, include/denet_hip.h): the
// C-ABI of the producing passes is unchanged.
#pragma once
#include "common.h"

struct BnFinalDev {
    unsigned* counter;      // [column groups], zero at rest; null: the rows are reduced by a later launch
    int kind;               // 1: forward statistics, 2: backward sums
    int C;                  // channels of the batch norm = row stride / 2 of the partial buffer
    long M;                 // values per channel
    float eps, momentum;
    float* o0;              // kind 1: save_mean      kind 2: dgamma
    float* o1;              //         save_invstd            dbeta
    float* o2;              //         run_mean / null        coef [2][C]
    float* o3;              //         run_stdinv / null      -
};

// host side (runtime.hip): the descriptor a caller armed for the NEXT producing pass of this thread; take returns a zero
// descriptor (counter == null) unless one is armed for this kind, channel count and at most the armed number of column groups
BnFinalDev denet_bn_final_take(int kind, int C, int groups);

constexpr int BNF_FC = 2, BNF_FJ = 64 / BNF_FC;      // channels per wave, row lanes (bn.hip: FC, FJ)

typedef __attribute__((address_space(1))) unsigned long long bnf_gu64;
typedef __attribute__((address_space(1))) unsigned bnf_gu32;

// a partial sum: write-through (sc1) 8-byte store / load past L1 and the XCD's L2
__device__ __forceinline__ void bnf_store(double* p, double v) {
    __hip_atomic_store((bnf_gu64*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool SC1>
__device__ __forceinline__ double bnf_load(const double* p) {
    if (SC1) return __longlong_as_double((long long)__hip_atomic_load((const bnf_gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return *p;
}

__device__ __forceinline__ void bnf_reduce_row_lanes(double& s, double& ss) {
#pragma unroll
    for (int off = BNF_FC; off < 64; off <<= 1) {
        s += __shfl_xor(s, off, 64);
        ss += __shfl_xor(ss, off, 64);
    }
}

// lane (c, jl) of a wave: the sums over rows jl, jl + FJ, ... of column c (4 independent row loads in flight)
template <bool SC1>
__device__ __forceinline__ void bnf_reduce_partials(const double* __restrict__ partial, int gy, int C, int c, int jl, double& s,
                                                    double& ss) {
    s = 0;
    ss = 0;
    if (c < C) {
        int j = jl;
        for (; j + 3 * BNF_FJ < gy; j += 4 * BNF_FJ) {
            const double a0 = bnf_load<SC1>(partial + (long)j * 2 * C + c), b0 = bnf_load<SC1>(partial + (long)j * 2 * C + C + c);
            const double a1 = bnf_load<SC1>(partial + (long)(j + BNF_FJ) * 2 * C + c),
                         b1 = bnf_load<SC1>(partial + (long)(j + BNF_FJ) * 2 * C + C + c);
            const double a2 = bnf_load<SC1>(partial + (long)(j + 2 * BNF_FJ) * 2 * C + c),
                         b2 = bnf_load<SC1>(partial + (long)(j + 2 * BNF_FJ) * 2 * C + C + c);
            const double a3 = bnf_load<SC1>(partial + (long)(j + 3 * BNF_FJ) * 2 * C + c),
                         b3 = bnf_load<SC1>(partial + (long)(j + 3 * BNF_FJ) * 2 * C + C + c);
            s += (a0 + a1) + (a2 + a3);
            ss += (b0 + b1) + (b2 + b3);
        }
        for (; j < gy; j += BNF_FJ) {
            s += bnf_load<SC1>(partial + (long)j * 2 * C + c);
            ss += bnf_load<SC1>(partial + (long)j * 2 * C + C + c);
        }
    }
}

// the same sums for U columns c[0..U-1] of one lane at once: the loads of all U columns of a round are issued together (a wave
// that walks its column pairs one after the other is latency-bound: two dependent round trips per pair), the additions of every
// column are those of bnf_reduce_partials in the same order
template <bool SC1, int U>
__device__ __forceinline__ void bnf_reduce_partials_multi(const double* __restrict__ partial, int gy, int C, const int (&c)[U], int jl,
                                                          double (&s)[U], double (&ss)[U]) {
    int cs[U];          // (a column outside the tensor: a valid address is loaded - no exec-masked load, which would end the
                        // scheduling region and drain the loads in flight - and the caller drops the result)
#pragma unroll
    for (int u = 0; u < U; ++u) {
        s[u] = ss[u] = 0;
        cs[u] = c[u] < C ? c[u] : 0;
    }
    int j = jl;
    // two groups of four rows per lane at a time (the loads of 2 x 4 x 2 x U values leave together: 256 rows - what a fused
    // kernel's launch leaves - are ONE round trip for the last workgroup, on whose latency the end of the launch waits)
    for (; j + 7 * BNF_FJ < gy; j += 8 * BNF_FJ) {
        double a[U][8], b[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const double* row = partial + (long)(j + r * BNF_FJ) * 2 * C;
                a[u][r] = bnf_load<SC1>(row + cs[u]);
                b[u][r] = bnf_load<SC1>(row + C + cs[u]);
            }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            s[u] += (a[u][0] + a[u][1]) + (a[u][2] + a[u][3]);
            ss[u] += (b[u][0] + b[u][1]) + (b[u][2] + b[u][3]);
            s[u] += (a[u][4] + a[u][5]) + (a[u][6] + a[u][7]);
            ss[u] += (b[u][4] + b[u][5]) + (b[u][6] + b[u][7]);
        }
    }
    for (; j + 3 * BNF_FJ < gy; j += 4 * BNF_FJ) {
        double a[U][4], b[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double* row = partial + (long)(j + r * BNF_FJ) * 2 * C;
                a[u][r] = bnf_load<SC1>(row + cs[u]);
                b[u][r] = bnf_load<SC1>(row + C + cs[u]);
            }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            s[u] += (a[u][0] + a[u][1]) + (a[u][2] + a[u][3]);
            ss[u] += (b[u][0] + b[u][1]) + (b[u][2] + b[u][3]);
        }
    }
    for (; j < gy; j += BNF_FJ) {
        double a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double* row = partial + (long)j * 2 * C;
            a[u] = bnf_load<SC1>(row + cs[u]);
            b[u] = bnf_load<SC1>(row + C + cs[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            s[u] += a[u];
            ss[u] += b[u];
        }
    }
}

// what the lane with jl == 0 does with the sums of channel c (the bodies of bn_stats_final_kernel / bn_bwd_final_kernel)
__device__ __forceinline__ void bnf_finish_stats(double s, double ss, long M, float eps, float momentum, int c,
                                                 float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                 float* __restrict__ run_mean, float* __restrict__ run_stdinv) {
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
__device__ __forceinline__ void bnf_finish_sums(double s, double ss, long M, int C, int c, float* __restrict__ dgamma,
                                                float* __restrict__ dbeta, float* __restrict__ coef) {
    dbeta[c] = (float)s;
    dgamma[c] = (float)ss;
    coef[c] = (float)(s / (double)M);
    coef[C + c] = (float)(ss / (double)M);
}

// Called by EVERY thread of a workgroup of NT threads (a multiple of 64) right after its threads have written the workgroup's
// partial row(s) with bnf_store. group: the column group [c0, c0 + cn) this workgroup contributed to; arrivals: the number of
// workgroups that contribute to it; rows: the rows of the partial buffer that hold its sums; s_flag: 4 bytes of LDS nobody else
// uses at this point (a word of the kernel's own LDS block - no second __shared__ object beside an LDS-DMA pipeline).
template <int NT>
__device__ __forceinline__ void bnf_tail(const BnFinalDev& f, const double* partial, int rows, int c0, int cn, int group,
                                         unsigned arrivals, int* s_flag) {
    if (!f.counter) return;                                  // uniform: the rows are reduced by a later launch
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // EVERY storing wave drains its write-through stores ...
    __syncthreads();                                         // ... before ONE lane takes the workgroup's ticket
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add((bnf_gu32*)(f.counter + group), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == arrivals - 1u) ? 1 : 0;
        if (last) __hip_atomic_store((bnf_gu32*)(f.counter + group), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // at rest again
        *s_flag = last;
    }
    __syncthreads();
    if (!*s_flag) return;
    // every other workgroup of the group has drained its rows to memory before its ticket: read them past the caches
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cl = lane % BNF_FC, jl = lane / BNF_FC;
    constexpr int U = 4, NW = NT / 64;
    for (int cc = wave * BNF_FC; cc < cn; cc += U * NW * BNF_FC) {
        int c[U];
        double s[U], ss[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = cc + u * NW * BNF_FC + cl;
            c[u] = k < cn ? c0 + k : f.C;              // (a column outside the group: skipped like one outside the tensor)
        }
        bnf_reduce_partials_multi<true, U>(partial, rows, f.C, c, jl, s, ss);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bnf_reduce_row_lanes(s[u], ss[u]);
            if (jl != 0 || c[u] >= f.C) continue;
            if (f.kind == 1) bnf_finish_stats(s[u], ss[u], f.M, f.eps, f.momentum, c[u], f.o0, f.o1, f.o2, f.o3);
            else bnf_finish_sums(s[u], ss[u], f.M, f.C, c[u], f.o0, f.o1, f.o2);
        }
    }
}
