// Data gradient of the 3x3 STRIDE-2 (pad 1) convolutions - the first convolution of every ResNet stage (denet/layer/convolution.py:
// 80-83 under tensor.grad, model_cnn.py:318 -> cuDNN bwd-data) - with the FOUR parity classes of input pixels in ONE workgroup.
// dx[2i+py][2j+px] = sum over the taps (r, s) with r = py + 1 - 2 di, s = px + 1 - 2 dj (di, dj in {0, 1}) of dy[i+di][j+dj] . w[r][s]:
// class (0,0) has one tap, (0,1) and (1,0) two, (1,1) four. The implicit-GEMM kernel (igemm.hip) runs every class as a problem of
// its own: 4 096 workgroups whose reductions are 4-16 chunks long, 55 TFLOP/s on the 64 <- 128 layer (EXPERIMENTS.md, rounds 4-5;
// DESIGN.md section 8 item 3). Here a workgroup owns 8 x 8 positions of dy-space = 16 x 16 pixels of dx for 64 channels: the four
// classes SHARE one dy patch (9 x 9 positions x 64 reduction channels per chunk, 20 KB, LDS-DMA, double-buffered) and one
// reduction of 9 taps x K channels:
//     wave = 16 of the 64 channels, all four classes x four blocks of 16 positions: 16 accumulator blocks = 64 registers;
//     per 16 reduction channels: 9 filter fragments straight from L2 (the filters packed [K/16][9][C][16]: a wave's fragment is one
//     contiguous KB), 16 dy fragments from LDS (4 shifts x 4 position blocks, ds_read_b128, the 16-byte slots of a position's 256
//     bytes XOR-swizzled at the DMA's source address with a key that is distinct over any 2 rows x 8 columns), 144 products;
//     epilogue: lane = (position, 4 channels): 16-byte stores of the four classes' pixels, optional add, optional backward sums of
//     the batch norm in front (wino4f.hip's epilogue).
// Two workgroups per CU (41 KB of LDS each). vmcnt discipline as in wino4t.hip: pieces and filter fragment loads share one in-order
// counter; a chunk's pieces are issued in the last third of the chunk before it.
#include "common.h"
#include "../../include/denet_hip.h"
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef int i32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void s2_store_b128(const f32x4& v, const i32x4_t& rsrc, int voff, int soff) {
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

struct S2Params {
    const float* dy;     // [N, OH, OW, K]
    const float* wp;     // packed [K/16][9][C][16] (denet_conv_dgrad_s2_pack)
    const float* add;    // [N, H, W, C] or null
    float* dx;           // [N, H, W, C]
    double* stats;       // [position blocks][2][C] or null
    const float* bs_x;   // backward sums (wino4f.hip)
    const float* bs_y;
    const float* bs_gamma;
    const float* bs_beta;
    const float* bs_mean;
    const float* bs_invstd;
    int bs_relu;
    int N, H, W, C, K, OH, OW;
    int bh, bw;          // blocks of 8 x 8 positions per image
    int tiles_c;         // C / 64
    int chunks;          // K / 64
    unsigned dy_bytes, dx_bytes, w_bytes;
};

constexpr int S2_OOB = (int)0xF0000000u;
constexpr int S2_SLOTS = 81 * 16;                 // 16-byte slots of a patch chunk: 81 positions x 64 channels
constexpr int S2_PIECES = 21;                     // of 64 slots (20.25)
constexpr int S2_BUF = S2_PIECES * 1024;          // 21 504
constexpr int S2_LDS = 2 * S2_BUF + 1024;         // two buffers + 1 KB that the surplus pieces write
constexpr int S2_D = 9;                           // filter fragments this many taps (one 16-channel step) ahead
constexpr int S2_R = 18;                          // ring (taps): divides the 36 taps of a chunk, > S2_D

#define S2_WAIT_VM(vm) __builtin_amdgcn_s_waitcnt(((vm) & 15) | ((((vm) >> 4) & 3) << 14) | (7 << 4) | (15 << 8))
#define S2_BARRIER()                       \
    {                                      \
        asm volatile("" ::: "memory");     \
        __builtin_amdgcn_s_barrier();      \
        asm volatile("" ::: "memory");     \
    }

// the swizzle key of a patch position (row pi, column pj): distinct over any two consecutive rows x eight consecutive columns
__device__ __forceinline__ int s2_key(int pi, int pj) { return ((pi & 1) << 3) | (pj & 7); }

// EP: 0 = store (+ add), 2 = + backward sums of the batch norm in front
template <int EP>
__global__ __launch_bounds__(256, 2) void dgrad_s2_kernel(const S2Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
    const int cblk = (int)(bid % (uint32_t)p.tiles_c);
    const int brow = (int)(bid / (uint32_t)p.tiles_c);
    int blk = brow;
    const int bx = blk % p.bw;
    blk /= p.bw;
    const int by = blk % p.bh;
    const int n = blk / p.bh;
    const int c0 = cblk * 64;
    const int i0 = by * 8, j0 = bx * 8;             // first dy position of the block
    const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, p.w_bytes, 0x00020000);

    // ---- LDS-DMA pieces: piece = wave + 4 j covers slots 64 piece .. + 63; slot = position * 16 + 16-byte slot of its 64 channels;
    // the lane fetches the channel quad (slot ^ key) of its position: the LDS image is swizzled, the DMA writes it linearly ----
    int pc_off[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int L = (wave + 4 * j) * 64 + lane;
        const int pos = L >> 4, sl = L & 15;
        const int pi = pos / 9, pj = pos - pi * 9;
        const int i = i0 + pi, jj = j0 + pj;
        const bool ok = L < S2_SLOTS && i < p.OH && jj < p.OW;
        pc_off[j] = ok ? (((n * p.OH + i) * p.OW + jj) * p.K + 4 * (sl ^ s2_key(pi, pj))) * 4 : S2_OOB;
    }
    int d_chunk = 0;
    auto issue = [&]() {
        const bool live = d_chunk < p.chunks;
        char* const dstb = smem + (d_chunk & 1) * S2_BUF;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int piece = wave + 4 * j;
            char* const dst = piece < S2_PIECES ? dstb + piece * 1024 : smem + 2 * S2_BUF;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rD, (lds_ptr_t)dst, 16, live ? pc_off[j] : S2_OOB, d_chunk * 256, 0, 0);
        }
        d_chunk += 1;
    };

    // ---- product role: wave = 16 channels; lane = (r15, g): filter fragment row = channel r15, dy fragment column = position r15
    // of a block of 2 rows x 8 columns, both with the reduction channels 4 g .. 4 g + 3 of a 16-channel step ----
    const int r15 = lane & 15, g = lane >> 4;
    const int w_voff = ((c0 + 16 * wave + r15) * 16 + 4 * g) * 4;
    const int w_tap = p.C * 64;                      // bytes per tap
    const int w_step = 9 * w_tap;                    // bytes per 16 reduction channels
    // dy fragment of (shift (di, dj), position block pb, 16-channel step q of the chunk): position (2 pb + (r15 >> 3) + di, (r15 & 7) + dj)
    const int pr = r15 >> 3, pq = r15 & 7;
    int v_base[2][2];                                // [di][dj]: byte offset of the position's 256 bytes, and its key
    int v_key[2][2];
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {
            v_base[di][dj] = ((pr + di) * 9 + pq + dj) * 256;
            v_key[di][dj] = s2_key(pr + di, pq + dj);     // (2 pb is even: it does not change the key's row bit)
        }

    f32x4 acc[4][4];                                 // [class 2 py + px][position block]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // filter fragments S2_D taps ahead (a whole 16-channel step: a fragment load issued behind a chunk's pieces is consumed 144
    // products later, when the pieces have landed); ring by the tap's number within the chunk (36 taps, S2_R divides 36)
    f32x4 fw[S2_R];
    const int steps_total = p.chunks * 4;
    auto load_w = [&](int s, int gt) {               // gt: tap number within chunk s (>= 36: the next chunk's)
        int st = 4 * s + gt / 9;
        st = st < steps_total ? st : steps_total - 1;      // (behind the last chunk the look-ahead re-reads its last step)
        fw[gt % S2_R] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, w_voff, st * w_step + (gt % 9) * w_tap, 0));
    };

    issue();
    issue();
#pragma unroll
    for (int t = 0; t < S2_D; ++t) load_w(0, t);

    for (int s = 0; s < p.chunks; ++s) {
        // this chunk's pieces have landed: only the S2_D look-ahead fragment loads are younger (the first chunk also waits for
        // the second chunk's pieces, once)
        S2_WAIT_VM(S2_D);
        S2_BARRIER();
        // every wave is done with chunk s - 1: its buffer takes chunk s + 1 (the first two chunks came with the prologue)
        if (s >= 1) issue();
        const char* const buf = smem + (s & 1) * S2_BUF;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // the four shifts' fragments of the four position blocks: slot (4 q + g) ^ key of the lane's position
            f32x4 dv[2][2][4];
#pragma unroll
            for (int di = 0; di < 2; ++di)
#pragma unroll
                for (int dj = 0; dj < 2; ++dj)
#pragma unroll
                    for (int pb = 0; pb < 4; ++pb)
                        dv[di][dj][pb] = *(const f32x4*)(buf + v_base[di][dj] + pb * (2 * 9 * 256) + (((4 * q + g) ^ v_key[di][dj]) << 4));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int gt = 9 * q + tap;
                load_w(s, gt + S2_D);
                const int r = tap / 3, sx = tap % 3;
                // tap (r, sx) feeds class (py, px) with r = py + 1 - 2 di: r = 1 -> (py 0, di 0); r = 0 -> (1, 1); r = 2 -> (1, 0)
                const int py = r == 1 ? 0 : 1, di = r == 0 ? 1 : 0;
                const int px = sx == 1 ? 0 : 1, dj = sx == 0 ? 1 : 0;
                const f32x4 wv = fw[gt % S2_R];
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int pb = 0; pb < 4; ++pb)
                        acc[2 * py + px][pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[k], dv[di][dj][pb][k], acc[2 * py + px][pb], 0, 0, 0);
                // issue order of the tap: its look-ahead fragment load behind the first product (left to the compiler the loads
                // sink to their uses and every tap waits for its own fragment)
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 15, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0);

    // ---- epilogue: lane = (position 2 pb + pr, pq of the block, channels c0 + 16 wave + 4 g .. + 3), four classes ----
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const i32x4_t rsX = {(int)(unsigned)(unsigned long long)p.dx, (int)(((unsigned long long)p.dx >> 32) & 0xffffu), (int)p.dx_bytes, 0x00020000};
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.add, 0, p.add ? p.dx_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rBX = __builtin_amdgcn_make_buffer_rsrc((void*)p.bs_x, 0, p.bs_x ? p.dx_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rBY = __builtin_amdgcn_make_buffer_rsrc((void*)p.bs_y, 0, p.bs_y ? p.dx_bytes : 0u, 0x00020000);
    const bool has_add = p.add != nullptr;
    const bool mask_y = p.bs_relu && p.bs_y, mask_x = p.bs_relu && !p.bs_y;
    const int kc = c0 + 16 * wave + 4 * g;
    double ds[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) ds[c] = 0.0;
    f32x4 bmu = z, bis = z, bsc = z, bsh = z;
    if (EP == 2) {
        bmu = *(const f32x4*)(p.bs_mean + kc);
        bis = *(const f32x4*)(p.bs_invstd + kc);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bsc[c] = (p.bs_gamma ? p.bs_gamma[kc + c] : 1.f) * bis[c];
            bsh[c] = (p.bs_beta ? p.bs_beta[kc + c] : 0.f) - bmu[c] * bsc[c];
        }
    }
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) {
        const int i = i0 + 2 * pb + pr, j = j0 + pq;
        const bool valid = i < p.OH && j < p.OW;
        const int voff = valid ? (((n * p.H + 2 * i) * p.W + 2 * j) * p.C + kc) * 4 : S2_OOB;
        f32x4 ssum = z, ssq = z;                     // the four classes of a position in fp32, doubles from there
        f32x4 av[4], xv[4], yv[4];
#pragma unroll
        for (int cl = 0; cl < 4; ++cl) {
            const int soff = ((cl >> 1) * p.W + (cl & 1)) * p.C * 4;
            av[cl] = z;
            if (has_add) av[cl] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, voff, soff, 0));
            if (EP == 2) {
                xv[cl] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rBX, voff, soff, 0));
                yv[cl] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rBY, voff, soff, 0));
            }
        }
#pragma unroll
        for (int cl = 0; cl < 4; ++cl) {
            const int soff = ((cl >> 1) * p.W + (cl & 1)) * p.C * 4;
            const f32x4 o = acc[cl][pb] + av[cl];
            s2_store_b128(o, rsX, voff, soff);
            if (EP == 2) {
                f32x4 gq;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float mk = mask_y ? yv[cl][c] : (mask_x ? fmaf(xv[cl][c], bsc[c], bsh[c]) : 1.f);
                    gq[c] = mk > 0.f ? o[c] : 0.f;
                    ssq[c] += gq[c] * ((xv[cl][c] - bmu[c]) * bis[c]);
                }
                ssum += gq;
            }
        }
        if (EP == 2 && valid) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                ds[c] += (double)ssum[c];
                ds[4 + c] += (double)ssq[c];
            }
        }
    }
    if (EP != 2) return;
    // over the 16 positions of the lane's row of 16 lanes (DPP row rotations), the wave owns its 16 channels' sums
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int sh = 8; sh > 0; sh >>= 1) {
            const long long b = __double_as_longlong(ds[c]);
            int lo, hi;
            if (sh == 8) { lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x128, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x128, 0xF, 0xF, false); }
            else if (sh == 4) { lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x124, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x124, 0xF, 0xF, false); }
            else if (sh == 2) { lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x122, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x122, 0xF, 0xF, false); }
            else { lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x121, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x121, 0xF, 0xF, false); }
            ds[c] += __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
        }
    }
    if (r15 == 0 && kc < p.C) {
        double* row = p.stats + (long)brow * 2 * p.C;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            row[kc + c] = ds[c];
            row[p.C + kc + c] = ds[4 + c];
        }
    }
}

// w [K][3][3][C] -> [K/16][9][C][16]
__global__ __launch_bounds__(256) void dgrad_s2_pack_kernel(const float* __restrict__ w, float* __restrict__ P, int K, int C) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;          // one destination float
    const long total = 9L * K * C;
    if (idx >= total) return;
    const int kk = (int)(idx & 15);
    long rest = idx >> 4;
    const int c = (int)(rest % C);
    rest /= C;
    const int tap = (int)(rest % 9);
    const int kc = (int)(rest / 9);
    P[idx] = w[((long)(kc * 16 + kk) * 9 + tap) * C + c];
}

}  // namespace

// geometry: a 3x3 stride-2 pad-1 convolution x [N,H,W,C] -> y [N,H/2,W/2,K] with even H and W; C and K multiples of 64
extern "C" int denet_conv_dgrad_s2_ok(int N, int H, int W, int C, int K) {
    return (N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 64 == 0 && K > 0 && K % 64 == 0 &&
            (long)N * H * W * C * 4 < 0x7FFFFFFFL && (long)N * (H / 2) * (W / 2) * K * 4 < 0x7FFFFFFFL && 9L * K * C * 4 < 0x7FFFFFFFL) ? 1 : 0;
}

extern "C" int denet_conv_dgrad_s2_stats_rows(int N, int H, int W) {
    return N * ((H / 2 + 7) / 8) * ((W / 2 + 7) / 8);
}

extern "C" int denet_conv_dgrad_s2_pack(const float* w, float* packed, int C, int K, hipStream_t stream) {
    DENET_CHECK_ARG(w && packed && C > 0 && K > 0 && K % 16 == 0, "conv_dgrad_s2_pack: bad arguments");
    const long total = 9L * K * C;
    hipLaunchKernelGGL(dgrad_s2_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, packed, K, C);
    DENET_CHECK_LAUNCH("conv_dgrad_s2_pack");
    return DENET_OK;
}

// dx = the data gradient of that convolution for dy (+ add); sums_of / stats_partial / stats_rows as denet_conv_dgrad_sums
// (rows = denet_conv_dgrad_s2_stats_rows), or all null
extern "C" int denet_conv_dgrad_s2(const float* dy, const float* w_packed, const float* add, float* dx, const denet_bn_link* sums_of,
                                   double* stats_partial, size_t stats_bytes, int* stats_rows, int N, int H, int W, int C, int K,
                                   hipStream_t stream) {
    DENET_CHECK_ARG(dy && w_packed && dx, "conv_dgrad_s2: null pointer");
    DENET_CHECK_ARG(denet_conv_dgrad_s2_ok(N, H, W, C, K), "conv_dgrad_s2: needs even H and W, C %% 64 = 0, K %% 64 = 0");
    S2Params p = {};
    p.dy = dy; p.wp = w_packed; p.add = add; p.dx = dx;
    p.N = N; p.H = H; p.W = W; p.C = C; p.K = K; p.OH = H / 2; p.OW = W / 2;
    p.bh = (p.OH + 7) / 8; p.bw = (p.OW + 7) / 8;
    p.tiles_c = C / 64; p.chunks = K / 64;
    p.dy_bytes = (unsigned)((size_t)N * p.OH * p.OW * K * 4);
    p.dx_bytes = (unsigned)((size_t)N * H * W * C * 4);
    p.w_bytes = (unsigned)((size_t)9 * K * C * 4);
    const long blocks = (long)N * p.bh * p.bw;
    int ep = 0;
    if (sums_of) {
        DENET_CHECK_ARG(stats_partial && stats_rows && stats_bytes >= (size_t)blocks * 2 * C * sizeof(double), "conv_dgrad_s2: statistics buffer too small");
        DENET_CHECK_ARG(sums_of->x && sums_of->mean && sums_of->invstd && (!sums_of->relu || sums_of->y || (sums_of->gamma && sums_of->beta)),
                        "conv_dgrad_s2: incomplete batch-norm description for the backward sums");
        *stats_rows = (int)blocks;
        p.stats = stats_partial;
        p.bs_x = sums_of->x; p.bs_y = sums_of->relu ? sums_of->y : nullptr; p.bs_gamma = sums_of->gamma; p.bs_beta = sums_of->beta;
        p.bs_mean = sums_of->mean; p.bs_invstd = sums_of->invstd; p.bs_relu = sums_of->relu;
        ep = 2;
    } else if (stats_rows) {
        *stats_rows = 0;
    }
    typedef void (*kern_t)(const S2Params);
    const kern_t fn = ep == 2 ? dgrad_s2_kernel<2> : dgrad_s2_kernel<0>;
    const int prof = denet_prof_begin(17, ep, 0, 0, stream);
    hipLaunchKernelGGL(fn, dim3((unsigned)(blocks * p.tiles_c)), dim3(256), S2_LDS, stream, p);
    denet_prof_end(prof, stream);
    DENET_CHECK_LAUNCH("conv_dgrad_s2");
    return DENET_OK;
}
