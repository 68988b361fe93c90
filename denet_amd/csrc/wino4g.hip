// Winograd F(4x4,3x3) FILTER gradient: the 36 component products dU[xi] = dM[xi]^T V[xi] (contraction over the tiles) and the
// adjoint filter transform dw = G^T dU G. Reference op: tensor.grad of the `C[k,3]` layer with respect to its filters
// (denet/layer/convolution.py:80-83, denet/model/model_cnn.py:318; cuDNN bwd-filter in the reference).
//
// The generic path (winograd.hip: denet_wgrad_batched through the implicit-GEMM filter-gradient kernel + splitk_reduce +
// wino_dfilter) treats a component as a 1x1 convolution's filter gradient: both operands are CONTRACTION-major (a row per tile),
// which that kernel stages as K-outer LDS tiles and reads with ds_read_b32 - four times the fragment-read instructions of a
// reduction-contiguous layout - on its single-buffer loop. Here the operand rows are used as they lie:
//   * a workgroup (4 waves) owns one component, a 128 x 128 block of dU and a slice of the tiles; a wave owns 64 x 64 = 16 blocks of
//     v_mfma_f32_16x16x4_f32 (64 accumulator registers), the four tiles of an MFMA's reduction depth are the four lane groups;
//   * lane (a, g) reads 16 bytes = channels 4a..4a+3 of tile g from a row of dM and a row of V: ONE ds_read_b128 per operand serves
//     16 products (product (r, s) takes element r of the dM fragment and element s of the V fragment: its output rows are the
//     channels 4a+r, its columns 4a'+s - a permutation that the 16-byte result stores undo for free). Rows of 512 B need no
//     swizzle: the 16 lanes of an LDS access group fall into disjoint bank ranges;
//   * rows arrive by LDS-DMA (buffer_load ... lds), 16 KB chunk buffers of 16 tiles, one barrier per chunk; four buffers (64 KB,
//     two workgroups per CU) where a component has few blocks, three (48 KB, three workgroups per CU) otherwise: the co-resident
//     workgroups cover each other's barriers;
//   * split over the tiles so that 2-3 workgroups per CU exist; every workgroup writes its partial block once, a reduction kernel
//     adds the slices in slice order (deterministic), wino_dfilter_kernel (winograd.hip) applies G^T . G.
// Exact fp32 FMA chains; the association differs from the generic path (sums over 8192 tiles in other groupings).
#include "common.h"
#include "../../include/denet_hip.h"
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct G4Params {
    const float* dM;     // [36][T][K]
    const float* V;      // [36][T][C]
    float* part;         // [splits][36][K][C]
    int T, K, C;
    int kblocks, cblocks, splits;
    int tiles_per_split; // multiple of 16
    unsigned dm_bytes, v_bytes;
};

constexpr int G4_CHUNK = 16;                 // tiles per chunk
constexpr int G4_ROW = 512;                  // bytes of a 128-channel row slice
constexpr int G4_OP = G4_CHUNK * G4_ROW;     // 8 KB per operand and chunk
constexpr int G4_SLOT = 2 * G4_OP;           // 16 KB
constexpr int G4_OOB = (int)0xF0000000u;

#define G4_WAIT_VM(vm) __builtin_amdgcn_s_waitcnt(((vm) & 15) | ((((vm) >> 4) & 3) << 14) | (7 << 4) | (15 << 8))
#define G4_BARRIER()                       \
    {                                      \
        asm volatile("" ::: "memory");     \
        __builtin_amdgcn_s_barrier();      \
        asm volatile("" ::: "memory");     \
    }

// NBUF = 4: 64 KB of LDS, two workgroups per CU, chunks issued two ahead (the 64x64 maps: every operand row is read once, from HBM);
// NBUF = 3: 48 KB, three workgroups per CU (many blocks per component: the rows come from L2)
template <int G4_NBUF>
__global__ __launch_bounds__(256, G4_NBUF == 4 ? 2 : 3) void wino4g_kernel(const G4Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave >> 1, wc = wave & 1;
    // workgroup id -> (split, component, k block, c block): the blocks of one component and slice are neighbours (they share rows)
    uint32_t b = xcd_remap(blockIdx.x, gridDim.x);
    const int cb = (int)(b % (uint32_t)p.cblocks);
    b /= (uint32_t)p.cblocks;
    const int kb = (int)(b % (uint32_t)p.kblocks);
    b /= (uint32_t)p.kblocks;
    const int xi = (int)(b % 36u);
    const int split = (int)(b / 36u);
    const int t_begin = split * p.tiles_per_split;
    const int t_end = min(p.T, t_begin + p.tiles_per_split);
    const int nchunks = (t_end - t_begin + G4_CHUNK - 1) / G4_CHUNK;

    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.dM, 0, p.dm_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.V, 0, p.v_bytes, 0x00020000);
    // DMA pieces: a chunk is 16 rows x 512 B per operand = 8 pieces of 1 KB (2 rows) each; wave w takes pieces 2w, 2w+1 of both
    // operands. Lane l of a piece: row 2 piece + l / 32, 16-byte slot l % 32
    int offA[2], offB[2], rowp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int piece = 2 * wave + j;
        rowp[j] = 2 * piece + (lane >> 5);
        const int slot = lane & 31;
        offA[j] = (((xi * p.T + t_begin + rowp[j]) * p.K) + kb * 128) * 4 + slot * 16;
        offB[j] = (((xi * p.T + t_begin + rowp[j]) * p.C) + cb * 128) * 4 + slot * 16;
    }
    const int stepA = G4_CHUNK * p.K * 4, stepB = G4_CHUNK * p.C * 4;
    int d_chunk = 0, d_buf = 0, d_sA = 0, d_sB = 0;
    auto issue = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            // rows beyond the slice (the last chunk of a ragged slice, the dummy chunks behind the end) read nothing: zeros
            const bool live = d_chunk < nchunks && t_begin + d_chunk * G4_CHUNK + rowp[j] < t_end;
            char* dst = smem + d_buf * G4_SLOT + (2 * wave + j) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)dst, 16, live ? offA[j] : G4_OOB, d_sA, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_ptr_t)(dst + G4_OP), 16, live ? offB[j] : G4_OOB, d_sB, 0, 0);
        }
        d_chunk += 1;
        d_buf = (d_buf + 1 == G4_NBUF) ? 0 : d_buf + 1;
        d_sA += stepA;
        d_sB += stepB;
    };

    // fragment address of lane (a = lane % 16, g = lane / 16) for reduction step ks of a chunk: row 4 ks + g, channels 4a..4a+3 of
    // this wave's 64-channel half
    const int a16 = lane & 15, g = lane >> 4;
    const int fA = g * G4_ROW + wk * 256 + a16 * 16;
    const int fB = G4_OP + g * G4_ROW + wc * 256 + a16 * 16;

    f32x4 acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[r][s] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 fa[2], fb[2];
    auto frags = [&](int buf, int ks, int slot) {
        fa[slot] = *(const f32x4*)(smem + buf * G4_SLOT + fA + ks * 4 * G4_ROW);
        fb[slot] = *(const f32x4*)(smem + buf * G4_SLOT + fB + ks * 4 * G4_ROW);
    };
    auto mm = [&](int slot) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[r][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[slot][r], fb[slot][s], acc[r][s], 0, 0, 0);
    };
    auto order = [&]() {
        // the two fragment reads of the next reduction step behind the first products of this one
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

    issue();
    issue();
    if (G4_NBUF == 4) issue();
    G4_WAIT_VM(4 * (G4_NBUF - 2));      // chunk 0 has landed (4 DMA instructions per chunk and wave)
    G4_BARRIER();
    frags(0, 0, 0);
    int buf = 0;
    for (int c = 0; c < nchunks; ++c) {
        const int nbuf = (buf + 1 == G4_NBUF) ? 0 : buf + 1;
        frags(buf, 1, 1);
        mm(0);
        order();
        frags(buf, 2, 0);
        mm(1);
        order();
        // chunk c+1 published, the buffer of chunk c-1 free: the pieces of chunk c + NBUF - 1 leave now
        G4_WAIT_VM(4 * (G4_NBUF - 3));
        G4_BARRIER();
        issue();
        frags(buf, 3, 1);
        mm(0);
        order();
        frags(nbuf, 0, 0);
        mm(1);
        order();
        buf = nbuf;
    }
    __builtin_amdgcn_s_waitcnt(0);

    // lane holds, per product (r, s) and register q: row k = 64 wk + 16 g + 4 q + r, column c = 64 wc + 4 a16 + s
    float* out = p.part + ((((long)split * 36 + xi) * p.K + kb * 128) * p.C + cb * 128);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 64 * wk + 16 * g + 4 * q + r;
            const f32x4 v = {acc[r][0][q], acc[r][1][q], acc[r][2][q], acc[r][3][q]};
            *(f32x4*)(out + (long)k * p.C + 64 * wc + 4 * a16) = v;
        }
}

// dU[i] = sum over the slices of part[s][i], in slice order (deterministic); one float4 per thread
__global__ __launch_bounds__(256) void wino4g_reduce_kernel(const float* __restrict__ part, float* __restrict__ dU, long n4, int splits) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4* p4 = (const f32x4*)part;
    f32x4 a = p4[i];
    for (int s = 1; s < splits; ++s) a += p4[i + (long)s * n4];
    ((f32x4*)dU)[i] = a;
}

int g_g4_mode = -1;       // -1: DENET_WINO4G from the environment (default on); 0: off; 1: on wherever the geometry allows

}  // namespace

// tests / experiments: overrides the environment's choice of the F(4x4) filter-gradient kernel (-1 restores it); returns the old value
extern "C" int denet_conv_wino4g_mode(int mode) {
    const int old = g_g4_mode;
    g_g4_mode = (mode == 0 || mode == 1) ? mode : -1;
    return old;
}

// number of tile slices this kernel would use (0: the generic path runs). Few blocks per component (the 64x64 maps: one or two):
// every operand row is streamed once from HBM - four chunk buffers, two workgroups per CU; many blocks: three buffers, three
// workgroups per CU. The slices fill those slots once.
static int g4_nbuf(int C, int K) { return 36L * (K / 128) * (C / 128) <= 72 ? 4 : 3; }
int denet_wino4g_splits(int tile, long T, int C, int K) {
    if (tile != 4 || C % 128 != 0 || K % 128 != 0 || T < 64) return 0;
    static const int env_on = [] { const char* e = getenv("DENET_WINO4G"); return e ? atoi(e) : 1; }();
    if (g_g4_mode == 0 || (g_g4_mode < 0 && !env_on)) return 0;
    if ((long)36 * T * (K > C ? K : C) * 4 >= 0x7FFFFFFFL) return 0;      // 32-bit byte offsets into the operands
    const long blocks = 36L * (K / 128) * (C / 128);
    long s = (g4_nbuf(C, K) == 4 ? 512 : 768) / blocks;
    if (s < 1) s = 1;
    const long max_s = T / 64;                           // at least four chunks per slice
    if (s > max_s) s = max_s;
    if (s > 16) s = 16;
    return (int)s;
}

size_t denet_wino4g_workspace_bytes(int splits, int C, int K) { return (size_t)splits * 36 * K * C * sizeof(float); }

// dU [36][K][C] = dM[xi]^T V[xi] from dM [36][T][K] and V [36][T][C]; part: denet_wino4g_workspace_bytes(splits, C, K) (not needed,
// and dU written directly, when splits == 1)
int denet_wino4g_run(int splits, const float* dM, const float* V, float* dU, float* part, size_t part_bytes, long T, int C, int K,
                     hipStream_t stream) {
    DENET_CHECK_ARG(dM && V && dU && splits >= 1 && T > 0 && C % 128 == 0 && K % 128 == 0, "conv_wino4g: bad arguments");
    DENET_CHECK_ARG(splits == 1 || (part && part_bytes >= denet_wino4g_workspace_bytes(splits, C, K)),
                    "conv_wino4g: workspace too small (%zu < %zu)", part_bytes, denet_wino4g_workspace_bytes(splits, C, K));
    G4Params p = {};
    p.dM = dM; p.V = V; p.part = splits == 1 ? dU : part;
    p.T = (int)T; p.K = K; p.C = C;
    p.kblocks = K / 128; p.cblocks = C / 128;
    const long per = (T + splits - 1) / splits;
    p.tiles_per_split = (int)((per + G4_CHUNK - 1) / G4_CHUNK * G4_CHUNK);
    // rounding the slice up to whole chunks can leave the last slices without tiles: they are not launched
    splits = (int)((T + p.tiles_per_split - 1) / p.tiles_per_split);
    if (splits == 1) p.part = dU;
    p.splits = splits;
    p.dm_bytes = (unsigned)((size_t)36 * T * K * 4);
    p.v_bytes = (unsigned)((size_t)36 * T * C * 4);
    const int nbuf = g4_nbuf(C, K);
    const int lds = nbuf * G4_SLOT;
    const unsigned grid = (unsigned)(splits * 36 * p.kblocks * p.cblocks);
    const int prof = denet_prof_begin(15, 128, 128, nbuf, stream);
    if (nbuf == 4) hipLaunchKernelGGL(wino4g_kernel<4>, dim3(grid), dim3(256), lds, stream, p);
    else hipLaunchKernelGGL(wino4g_kernel<3>, dim3(grid), dim3(256), lds, stream, p);
    denet_prof_end(prof, stream);
    if (splits > 1) {
        const long n4 = 36L * K * C / 4;
        hipLaunchKernelGGL(wino4g_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, part, dU, n4, splits);
    }
    DENET_CHECK_LAUNCH("conv_wino4g");
    return DENET_OK;
}
