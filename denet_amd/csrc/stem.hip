// The first layer of the network, `C.B[64,7,2]` on the 3-channel image (reference: denet/layer/convolution.py:80-83, desc
// examples/resnet34-imagenet.sh:7; its filter gradient: tensor.grad in model_cnn.py:318): forward pass and filter gradient as
// kernels of their own (the layer has no data gradient). The generic implicit-GEMM kernel (igemm.hip) walks this layer as
// 7 x 8 x 4 = 224 padded taps per filter - a third of its matrix work multiplies zeros (the 4th channel of the padded image,
// the 8th filter column). Here the reduction is the 147 real taps padded to 160:
//   * a persistent 512-thread workgroup per CU walks 8 x 64 blocks of OUTPUT pixels; the 21 x 133 input pixels a block reads
//     are staged once in LDS as [row][column][3] floats (the padding channel dropped; image borders zero-filled), double
//     buffered: the next block's patch is loaded in pieces while this block is multiplied;
//   * wave w owns output row w of the block. A tap (r, s, c) of output pixel px sits at  (2w + r) * RS + (2 px + s) * 3 + c
//     = pixel base + tap offset, and the 21 taps of a filter row are CONTIGUOUS: tap t -> (t / 21) * RS + t % 21. Operands
//     go from LDS straight into v_mfma_f32_32x32x2_f32 (one ds_read_b32 per fragment);
//   * forward: D^T[filter][pixel] += w[tap][filter] * patch[pixel][tap], the 160 x 64 filter matrix resident in LDS; epilogue
//     bias + whole-line stores (each 32 x 32 block transposed through a per-wave LDS staging area) + the batch-norm column
//     sums of y (every value enters as a double: one row per WORKGROUP);
//   * filter gradient: D[filter][tap] += dy[pixel][filter] * patch[pixel][tap]; here a wave owns 32 filters x all 160 taps over
//     TWO rows of the block, its 80 accumulator registers live over ALL its blocks; the waves add theirs in a fixed order
//     through LDS, one partial per workgroup goes to the workspace and stem_wgrad_reduce_kernel adds the partials in workgroup
//     order (bit-reproducible) into KRSC.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int TH = 8, TW = 64;                     // output pixels of a block
constexpr int IR = 2 * TH + 5, IC = 2 * TW + 5;    // input pixels it reads: 21 x 133
constexpr int RS = 405;                            // floats per staged row (3 * 133 = 399 used; 405 % 32 = 21: a filter row's
                                                   // taps continue on the banks the previous row's left off)
constexpr int REGION = IR * RS;                    // 8505 floats = 34 KB
constexpr int TAPS = 147, KP = 160;                // 7 * 7 * 3 real taps, padded to 5 x 32
constexpr int NTH = 512;
constexpr int PIECES = (IR * IC + NTH - 1) / NTH;  // 6 rounds of one pixel per thread
constexpr int WLS = 65;                            // floats per tap row of the resident filter matrix (64 filters + 1: the fill
                                                   // walks taps at a fixed filter, conflict-free at an odd stride)
constexpr int STG = 36;                            // floats per pixel row of a wave's store staging (32 filters + 4: conflict-free
                                                   // 16-byte writes down a column and reads along a row)
constexpr int F_LDS_BYTES = (KP * WLS + 2 * REGION + 64 + 8 * 32 * STG) * 4;   // 146 760 (+ bias, + staging)
constexpr int G_LDS_BYTES = 2 * REGION * 4;                //  68 040 (the wave sums reuse it: 64 * 160 floats)
constexpr int PART = 64 * KP;                              // floats of one workgroup's partial filter gradient

// s_waitcnt lgkmcnt(0) (vmcnt untouched: result stores and prefetches stay in flight; gfx9 encoding vmcnt = 15:14 | 3:0, expcnt
// 6:4, lgkmcnt 11:8) + s_barrier: __syncthreads() would drain every vector-memory operation of the wave
#define STEM_BARRIER()                                              \
    {                                                               \
        __builtin_amdgcn_s_waitcnt(15 | (3 << 14) | (7 << 4));      \
        __builtin_amdgcn_s_barrier();                               \
    }

struct StemParams {
    const float* x;      // [N][H][W][4], or planar [N][3][H][W] (the reference's own input layout)
    const float* w;      // [64][7][8][4] (forward)
    const float* bias;   // [64] or null
    float* y;            // [N][OH][OW][64] (forward)
    double* stats;       // [grid][2][64] or null
    const float* dy;     // [N][OH][OW][64] (filter gradient)
    float* part;         // [grid][PART]
    int N, H, W, OH, OW, tiles_y, tiles_x, tiles, planar, relu;
};

struct Block {
    int n, oy0, ox0;
};

__device__ __forceinline__ Block block_of(const StemParams& p, int tile) {
    Block b;
    const int tx = tile % p.tiles_x, r = tile / p.tiles_x;
    b.ox0 = tx * TW;
    b.oy0 = (r % p.tiles_y) * TH;
    b.n = r / p.tiles_y;
    return b;
}

// offset of tap t = (r * 7 + s) * 3 + c from a pixel's base; the padding taps read the base itself (their filter rows are zero
// in the forward pass, their gradient columns are dropped)
__device__ __forceinline__ constexpr int tap_off(int t) { return t < TAPS ? (t / 21) * RS + t % 21 : 0; }

typedef float f32x3 __attribute__((ext_vector_type(3)));      // (a 16-byte load would leave a dead register the compiler reuses at once)

// piece k of a block's patch: pixel (tid + 512 k) of the 21 x 133, image -> registers, registers -> LDS. Branch-free: a clamped
// address is always loaded and nothing touches the registers until the piece is stored (so nothing waits on the load before);
// pixels outside the image become zeros there
__device__ __forceinline__ f32x3 piece_load(const StemParams& p, const Block& b, int k, int tid) {
    const int idx = tid + NTH * k;
    const int rr = idx / IC, cc = idx - rr * IC;
    const int iy = 2 * b.oy0 - 3 + rr, ix = 2 * b.ox0 - 3 + cc;
    const int iyc = iy < 0 ? 0 : iy >= p.H ? p.H - 1 : iy, ixc = ix < 0 ? 0 : ix >= p.W ? p.W - 1 : ix;
    f32x3 v;
    if (p.planar) {
        const float* const src = p.x + ((long)b.n * 3 * p.H + iyc) * p.W + ixc;
        const long plane = (long)p.H * p.W;
        v[0] = src[0];
        v[1] = src[plane];
        v[2] = src[2 * plane];
    } else {
        __builtin_memcpy(&v, p.x + (((long)b.n * p.H + iyc) * p.W + ixc) * 4, 12);
    }
    return v;
}
__device__ __forceinline__ void piece_store(const StemParams& p, const Block& b, float* buf, int k, int tid, const f32x3 v) {
    const int idx = tid + NTH * k;
    if (idx < IR * IC) {
        const int rr = idx / IC, cc = idx - rr * IC;
        const int iy = 2 * b.oy0 - 3 + rr, ix = 2 * b.ox0 - 3 + cc;
        const bool in = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        float* d = buf + rr * RS + cc * 3;
        d[0] = in ? v[0] : 0.f;
        d[1] = in ? v[1] : 0.f;
        d[2] = in ? v[2] : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTH, 2) void stem_fwd_kernel(const StemParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const wl = smem;                      // [160][WLS]: wl[t][f] = w[f][r][s][c]
    float* const reg0 = smem + KP * WLS;
    float* const bl = reg0 + 2 * REGION;         // [64] bias (zeros without one): the epilogue reads it from LDS - a global load
                                                 // there would wait for the result stores issued before it (one counter)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, lh = lane >> 5;
    float* const stg = bl + 64 + wave * 32 * STG;    // the wave's own [32 pixels][STG]

    // the filters, read in memory order (coalesced) and scattered to [tap][filter]; the 13 padding taps are zero rows
    // (28 loads per thread, all in flight at once)
    {
        float wv[64 * 224 / NTH];
#pragma unroll
        for (int k = 0; k < 64 * 224 / NTH; ++k) wv[k] = p.w[tid + NTH * k];
#pragma unroll
        for (int k = 0; k < 64 * 224 / NTH; ++k) {
            const int i = tid + NTH * k;
            const int f = i / 224, q = i - 224 * f, r = q >> 5, s = (q >> 2) & 7, c = q & 3;
            if (s < 7 && c < 3) wl[(r * 21 + s * 3 + c) * WLS + f] = wv[k];
        }
    }
    for (int i = tid; i < (KP - TAPS) * 64; i += NTH) wl[(TAPS + (i >> 6)) * WLS + (i & 63)] = 0.f;
    if (tid < 64) bl[tid] = p.bias ? p.bias[tid] : 0.f;
    int tile = blockIdx.x;
    Block cur = block_of(p, tile);
#pragma unroll
    for (int k = 0; k < PIECES; ++k) piece_store(p, cur, reg0, k, tid, piece_load(p, cur, k, tid));
    __syncthreads();

    // batch-norm sums of the filters the lane STORES (32 j + 4 (lane & 7) + {0..3}, see the epilogue) over all its pixels
    // (doubles throughout)
    double dsum[2][4], dsq[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) dsum[j][q] = dsq[j][q] = 0.0;

    const float* const Wl = wl + lh * WLS + li;
    int it = 0;
    while (true) {
        float* const buf = reg0 + (it & 1) * REGION;
        float* const nbuf = reg0 + ((it + 1) & 1) * REGION;
        const int next = tile + gridDim.x;
        const bool has_next = next < p.tiles;
        const Block nxt = block_of(p, has_next ? next : tile);
        const float* const A0 = buf + 2 * wave * RS + 6 * li;      // pixel li of the row (pixel 32 + li: + 192 floats)
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        // 80 reduction steps of two taps, in 4 chunks; the next block's patch comes in beside chunks 0..3 (pieces 0-1, 2-3, 4, 5).
        // The fragments of step j + 1 are read BEFORE the matrix instructions of step j are issued (two register sets; the
        // scheduler is pinned: 2 LDS reads, 4 MFMAs), so an LDS round trip hides behind 256 cycles of the matrix pipe
        auto frag = [&](int j, float (&a)[2], float (&b)[2]) {
            const int o = lh ? tap_off(2 * j + 1) : tap_off(2 * j);
            a[0] = A0[o];
            a[1] = A0[o + 192];
            b[0] = Wl[2 * j * WLS];
            b[1] = Wl[2 * j * WLS + 32];
        };
        float fa[2][2], fb[2][2];
        frag(0, fa[0], fb[0]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x3 pv[2];
            const int k0 = c < 2 ? 2 * c : 2 + c, nk = c < 2 ? 2 : 1;
            if (has_next) {
#pragma unroll
                for (int k = 0; k < nk; ++k) pv[k] = piece_load(p, nxt, k0 + k, tid);
            }
#pragma unroll
            for (int jj = 0; jj < 20; ++jj) {
                const int j = 20 * c + jj, cs = j & 1;
                if (j + 1 < 80) frag(j + 1, fa[cs ^ 1], fb[cs ^ 1]);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[cs][0], fa[cs][0], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[cs][1], fa[cs][0], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[cs][0], fa[cs][1], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[cs][1], fa[cs][1], acc[1][1], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // DS reads (ds_read2_b32 pairs)
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);      // MFMA
            }
            if (has_next) {
#pragma unroll
                for (int k = 0; k < nk; ++k) piece_store(p, nxt, nbuf, k0 + k, tid, pv[k]);
            }
        }
        // the next patch is complete and everybody is done with this one. The barrier stands BEFORE the epilogue: its stores
        // drain behind the next block's matrix work, and the waves of a SIMD drift apart (one's epilogue under the other's MFMAs)
        STEM_BARRIER();
        // epilogue: lane (li, lh) holds pixel li (+ 32 i) of the row and, per (j, g), filters 32 j + 8 g + 4 lh + {0..3} - stored
        // from there a 16-byte store touches 32 lines. Each 32 x 32 block goes through the wave's LDS staging instead and leaves
        // as [pixel][32 filters]: lane L writes filters 4 (L & 7).. of pixel 8 r + (L >> 3), 8 whole 128-byte rows per store
        const int oy = cur.oy0 + wave;
        float* const yrow = p.y + (((long)cur.n * p.OH + oy) * p.OW + cur.ox0) * 64 + 4 * (lane & 7);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *(f32x4*)(stg + li * STG + 8 * g + 4 * lh) =
                        f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                const f32x4 bias4 = *(const f32x4*)(bl + 32 * j + 4 * (lane & 7));
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int px = 32 * i + 8 * r + (lane >> 3);
                    f32x4 v = *(const f32x4*)(stg + (8 * r + (lane >> 3)) * STG + 4 * (lane & 7));
                    v += bias4;
                    if (p.relu) v = __builtin_elementwise_max(v, f32x4{0.f, 0.f, 0.f, 0.f});      // (the inference fold of BN + ReLU)
                    if (oy < p.OH && cur.ox0 + px < p.OW) {
                        *(f32x4*)(yrow + px * 64 + 32 * j) = v;
                        // the one biased convolution in front of a batch norm (C.B[64,7,2] BN): every value and its square
                        // enter the sums as doubles (128 double FMAs per lane and tile: nothing beside the 160-deep products)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const double dv = (double)v[q];
                            dsum[j][q] += dv;
                            dsq[j][q] = __builtin_fma(dv, dv, dsq[j][q]);
                        }
                    }
                }
            }
        if (!has_next) break;
        tile = next;
        cur = nxt;
        ++it;
    }
    if (p.stats) {
        // over the 8 pixel lanes that share the filters (doubles from here), then over the waves in a fixed order
        double* const red = (double*)reg0;           // [8][2][64]
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                double a = dsum[j][q], b = dsq[j][q];
#pragma unroll
                for (int off = 32; off >= 8; off >>= 1) {
                    a += __shfl_xor(a, off, 64);
                    b += __shfl_xor(b, off, 64);
                }
                if (lane < 8) {
                    const int f = 32 * j + 4 * lane + q;
                    red[(wave * 2 + 0) * 64 + f] = a;
                    red[(wave * 2 + 1) * 64 + f] = b;
                }
            }
        __syncthreads();
        if (tid < 128) {
            double a = 0.0;
#pragma unroll
            for (int wv = 0; wv < 8; ++wv) a += red[wv * 128 + tid];
            p.stats[(long)blockIdx.x * 128 + tid] = a;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTH, 2) void stem_wgrad_kernel(const StemParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const reg0 = smem;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, lh = lane >> 5;

    int tile = blockIdx.x;
    Block cur = block_of(p, tile);
#pragma unroll
    for (int k = 0; k < PIECES; ++k) piece_store(p, cur, reg0, k, tid, piece_load(p, cur, k, tid));

    // wave (mt, rp) = (wave >> 2, wave & 3): filters 32 mt .. 32 mt + 31 over output rows 2 rp and 2 rp + 1 of the block.
    // The lane's tap of each of the five 32-tap tiles, as an offset from a pixel's base (+ the lane half's pixel of a pair)
    const int mt = wave >> 2, rp = wave & 3;
    int toff[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int t = 32 * j + li;
        toff[j] = (t < TAPS ? (t / 21) * RS + t % 21 : 0) + 4 * rp * RS + 6 * lh;
    }
    f32x16 acc[5];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    // dy of 8 pixel pairs (chunk c of 8: row c >> 2, pairs 8 (c & 3) ..) for the lane: pixel 2 kk + lh, filter 32 mt + li.
    // Branch-free: clamped addresses; bit kk of the returned mask says whether the pixel exists (ragged blocks) - the zeroing
    // happens when the values are USED, a chunk later, so that nothing waits on the loads here
    auto load_dy = [&](const Block& b, int c, float (&a)[8]) -> unsigned {
        const int oy = b.oy0 + 2 * rp + (c >> 2), oyc = oy < p.OH ? oy : p.OH - 1;
        const float* const r = p.dy + ((long)b.n * p.OH + oyc) * p.OW * 64 + 32 * mt + li;
        unsigned m = 0;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int ox = b.ox0 + 2 * (8 * (c & 3) + kk) + lh, oxc = ox < p.OW ? ox : p.OW - 1;
            a[kk] = r[oxc * 64];
            m |= (unsigned)(oy == oyc && ox == oxc) << kk;
        }
        return m;
    };
    float an[8];
    unsigned am = load_dy(cur, 0, an);
    __syncthreads();

    int it = 0;
    while (true) {
        const float* const buf = reg0 + (it & 1) * REGION;
        float* const nbuf = reg0 + ((it + 1) & 1) * REGION;
        const int next = tile + gridDim.x;
        const bool has_next = next < p.tiles;
        const Block nxt = block_of(p, has_next ? next : tile);
        // the next block's patch comes in beside chunks 0..5, a piece each
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float ac[8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) ac[kk] = ((am >> kk) & 1) ? an[kk] : 0.f;
            // the following chunk's dy (the next block's first after the last)
            if (c < 7) am = load_dy(cur, c + 1, an);
            else if (has_next) am = load_dy(nxt, 0, an);
            f32x3 pv = {0.f, 0.f, 0.f};
            if (c < PIECES && has_next) pv = piece_load(p, nxt, c, tid);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                // row c >> 2 of the wave's two (2 staged rows down), pair 8 (c & 3) + kk (2 pixels x 2 columns x 3 floats each)
                const int o = (c >> 2) * 2 * RS + 12 * (8 * (c & 3) + kk);
                float b[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) b[j] = buf[toff[j] + o];
#pragma unroll
                for (int j = 0; j < 5; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[kk], b[j], acc[j], 0, 0, 0);
            }
            if (c < PIECES && has_next) piece_store(p, nxt, nbuf, c, tid, pv);
        }
        STEM_BARRIER();             // (the dy prefetch of the next block stays in flight)
        if (!has_next) break;
        tile = next;
        cur = nxt;
        ++it;
    }
    // the waves' sums, added in wave order (four per filter half): element (j, e) of lane l at ((5 mt + j) * 16 + e) * 64 + l
    float* const red = reg0;
#pragma unroll 1
    for (int wv = 0; wv < 4; ++wv) {
        if (rp == wv) {
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float* const d = red + ((5 * mt + j) * 16 + e) * 64 + lane;
                    *d = wv ? *d + acc[j][e] : acc[j][e];
                }
        }
        __syncthreads();
    }
    float* const out = p.part + (long)blockIdx.x * PART;
    for (int i = tid; i < PART; i += NTH) out[i] = red[i];
}

// dw[f][r][s][c] (KRSC with the padded 8th column and 4th channel zero) = sum over the workgroups' partials in a FIXED order: a
// workgroup takes 64 consecutive elements of the partial layout (one accumulator register of a wave's 32 x 32 block: element e of
// block (i, j), lanes (lh, li)); its four waves sum a quarter of the partials each, one after the other, and the four quarter sums
// are added in wave order
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ part, int grid, float* __restrict__ dw) {
    __shared__ float q[4][64];
    const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int row = blockIdx.x;                      // (5 i + j) * 16 + e
    const int per = (grid + 3) / 4;
    const int k0 = seg * per, k1 = k0 + per < grid ? k0 + per : grid;
    const float* src = part + row * 64 + lane;
    float v = 0.f;
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = src[(long)(k + u) * PART];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; k < k1; ++k) v += src[(long)k * PART];
    q[seg][lane] = v;
    __syncthreads();
    if (seg == 0) {
        v = ((q[0][lane] + q[1][lane]) + q[2][lane]) + q[3][lane];
        const int e = row & 15, blk = row >> 4, i = blk / 5, j = blk - 5 * i;
        const int lh = lane >> 5, li = lane & 31;
        const int f = 32 * i + 8 * (e >> 2) + 4 * lh + (e & 3), t = 32 * j + li;
        if (t < TAPS) {
            const int r = t / 21, w = t - 21 * r, sx = w / 3, c = w - 3 * sx;
            dw[((f * 7 + r) * 8 + sx) * 4 + c] = v;
        }
    }
    // the padding entries (8th column, 4th channel) are zero
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o < 64 * 224 && (((o >> 2) & 7) == 7 || (o & 3) == 3)) dw[o] = 0.f;
}

int stem_grid(int tiles) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
        cus = prop.multiProcessorCount;
    }
    return tiles < cus ? tiles : cus;
}

int fill(StemParams& p, int N, int H, int W) {
    p.N = N; p.H = H; p.W = W; p.OH = H / 2; p.OW = W / 2;
    p.tiles_y = (p.OH + TH - 1) / TH;
    p.tiles_x = (p.OW + TW - 1) / TW;
    p.tiles = N * p.tiles_y * p.tiles_x;
    return stem_grid(p.tiles);
}

}  // namespace

// the geometry these kernels cover: the physical layout of `C.B[64,7,2]` on a 3-channel image (C = 4, S padded 7 -> 8, pad 3).
// DENET_STEM: bit 0 allows the forward kernel, bit 1 the filter gradient (default 3)
extern "C" int denet_conv_stem_ok(int pass, int N, int H, int W, int C, int K, int R, int S, int S_real, int stride, int pad, int OH,
                                  int OW) {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("DENET_STEM");
        v = e ? atoi(e) : 3;
    }
    return ((v >> (pass ? 1 : 0)) & 1) && C == 4 && K == 64 && R == 7 && S == 8 && S_real == 7 && stride == 2 && pad == 3 && H % 2 == 0 &&
           W % 2 == 0 && OH == H / 2 && OW == W / 2 && N > 0 && (long)N * H * W * 4 < (1L << 31);
}

extern "C" int denet_conv_stem_fwd_from(const float* x, int x_nchw, const float* w, const float* bias, float* y, double* stats_partial,
                                        size_t stats_bytes, int* stats_rows, int N, int H, int W, hipStream_t stream);
extern "C" int denet_conv_stem_wgrad_from(const float* x, int x_nchw, const float* dy, float* dw, float* workspace,
                                          size_t workspace_bytes, int N, int H, int W, hipStream_t stream);

// y = conv7x7/2(x) (+ bias) for x [N][H][W][4] (4th channel ignored), w [64][7][8][4]; stats_partial (optional): the batch-norm
// column sums of y, [rows][2][64] doubles with rows = the launch's workgroups (<= 256 on this chip; *stats_rows receives it)
extern "C" int denet_conv_stem_fwd(const float* x, const float* w, const float* bias, float* y, double* stats_partial,
                                   size_t stats_bytes, int* stats_rows, int N, int H, int W, hipStream_t stream) {
    return denet_conv_stem_fwd_from(x, 0, w, bias, y, stats_partial, stats_bytes, stats_rows, N, H, W, stream);
}

// the same; x_nchw != 0: x is the image batch as the reference holds it, [N][3][H][W] (dataset/__init__.py:359, the layout
// model_cnn.py feeds the first layer) - no NHWC copy of the input has to exist
extern "C" int denet_conv_stem_fwd_act(const float* x, int x_nchw, const float* w, const float* bias, float* y, int relu,
                                       double* stats_partial, size_t stats_bytes, int* stats_rows, int N, int H, int W,
                                       hipStream_t stream);
extern "C" int denet_conv_stem_fwd_from(const float* x, int x_nchw, const float* w, const float* bias, float* y, double* stats_partial,
                                        size_t stats_bytes, int* stats_rows, int N, int H, int W, hipStream_t stream) {
    return denet_conv_stem_fwd_act(x, x_nchw, w, bias, y, 0, stats_partial, stats_bytes, stats_rows, N, H, W, stream);
}

// the same with y = max(y, 0) when relu != 0 (inference: the batch norm behind the layer folded into w / bias, its ReLU here)
extern "C" int denet_conv_stem_fwd_act(const float* x, int x_nchw, const float* w, const float* bias, float* y, int relu,
                                       double* stats_partial, size_t stats_bytes, int* stats_rows, int N, int H, int W,
                                       hipStream_t stream) {
    DENET_CHECK_ARG(x && w && y, "conv_stem_fwd: null pointer");
    DENET_CHECK_ARG(N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "conv_stem_fwd: needs even H and W");
    StemParams p = {};
    p.x = x; p.w = w; p.bias = bias; p.y = y; p.planar = x_nchw ? 1 : 0; p.relu = relu ? 1 : 0;
    const int grid = fill(p, N, H, W);
    DENET_CHECK_ARG(grid > 0, "conv_stem_fwd: cannot query the device");
    if (stats_partial) {
        DENET_CHECK_ARG(stats_rows && stats_bytes >= (size_t)grid * 128 * sizeof(double), "conv_stem_fwd: statistics buffer too small");
        *stats_rows = grid;
        p.stats = stats_partial;
    }
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)stem_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS_BYTES);
        if (e != hipSuccess) {
            denet_set_error("conv_stem_fwd: hipFuncSetAttribute(%d B LDS): %s", F_LDS_BYTES, hipGetErrorString(e));
            return -(int)e;
        }
        attr_set = true;
    }
    const int prof = denet_prof_begin(12, 0, 0, 0, stream);
    hipLaunchKernelGGL(stem_fwd_kernel, dim3((unsigned)grid), dim3(NTH), F_LDS_BYTES, stream, p);
    denet_prof_end(prof, stream);
    DENET_CHECK_LAUNCH("conv_stem_fwd");
    return DENET_OK;
}

extern "C" size_t denet_conv_stem_wgrad_workspace_bytes(void) {
    const int grid = stem_grid(1 << 30);
    return grid > 0 ? (size_t)grid * PART * sizeof(float) : 0;
}

// dw [64][7][8][4] = the filter gradient of the same layer from x and dy [N][H/2][W/2][64]; workspace: one partial per workgroup
extern "C" int denet_conv_stem_wgrad(const float* x, const float* dy, float* dw, float* workspace, size_t workspace_bytes, int N,
                                     int H, int W, hipStream_t stream) {
    return denet_conv_stem_wgrad_from(x, 0, dy, dw, workspace, workspace_bytes, N, H, W, stream);
}

extern "C" int denet_conv_stem_wgrad_from(const float* x, int x_nchw, const float* dy, float* dw, float* workspace,
                                          size_t workspace_bytes, int N, int H, int W, hipStream_t stream) {
    DENET_CHECK_ARG(x && dy && dw && workspace, "conv_stem_wgrad: null pointer");
    DENET_CHECK_ARG(N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "conv_stem_wgrad: needs even H and W");
    StemParams p = {};
    p.x = x; p.dy = dy; p.part = workspace; p.planar = x_nchw ? 1 : 0;
    const int grid = fill(p, N, H, W);
    DENET_CHECK_ARG(grid > 0, "conv_stem_wgrad: cannot query the device");
    DENET_CHECK_ARG(workspace_bytes >= (size_t)grid * PART * sizeof(float), "conv_stem_wgrad: workspace too small");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)stem_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS_BYTES);
        if (e != hipSuccess) {
            denet_set_error("conv_stem_wgrad: hipFuncSetAttribute(%d B LDS): %s", G_LDS_BYTES, hipGetErrorString(e));
            return -(int)e;
        }
        attr_set = true;
    }
    const int prof = denet_prof_begin(13, 0, 0, 0, stream);
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3((unsigned)grid), dim3(NTH), G_LDS_BYTES, stream, p);
    denet_prof_end(prof, stream);
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(PART / 64), dim3(256), 0, stream, workspace, grid, dw);
    DENET_CHECK_LAUNCH("conv_stem_wgrad");
    return DENET_OK;
}
