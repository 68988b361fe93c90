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
__device__ __forceinline__ W2Item w2_item(const W2Params& p, int item) {
    W2Item it;
    it.co0 = (item % p.nco) * 64;
    it.block = item / p.nco;
    int b = it.block;
    it.ox0 = (b % p.bx) * 16;
    b /= p.bx;
    it.oy0 = (b % p.by) * 16;
    it.n = b / p.by;
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
    int pc[3];                                       // row | column << 8; column 255: not a pixel
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int sl = 64 * (3 * (wave & 1) + j) + lane;
        const int row = sl / P_ROW, r = sl - row * P_ROW, par = r / P_PAR, col = r - par * P_PAR;
        pc[j] = row | ((col < 9 && sl < P_USED ? 2 * col + par : 255) << 8);
    }
    auto patch_piece = [&](int g, int j, const W2Item& it) {
        const int plane = 4 * g + (wave >> 1);
        const int k = 3 * (wave & 1) + j;
        const int iy = it.oy0 - 1 + (pc[j] & 255), ix = it.ox0 - 1 + (pc[j] >> 8);
        const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && (pc[j] >> 8) != 255;
        const int off = (((it.n * p.H + iy) * p.W + ix) * CI + plane * 4) * 4;
        if (j < 2 || 64 * k + lane < P_USED)          // the last piece of a plane is 40 slots
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(P + plane * P_PLANE + 64 * k), 16, ok ? off : OOB, 0, 0, 0);
    };

    int item = blockIdx.x;
    if (item >= p.items) return;
    W2Item cur = w2_item(p, item);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 3; ++j) patch_piece(g, j, cur);
#pragma unroll
    for (int k = 0; k < 4; ++k) u_piece(UA, 0, 0, cur.co0, k);
#pragma unroll
    for (int k = 0; k < 4; ++k) u_piece(UB, 1, 0, cur.co0, k);
    W2_BARRIER(0);

    while (true) {
        const int next = item + gridDim.x;
        const bool has_next = next < p.items;
        const W2Item nxt = w2_item(p, has_next ? next : item);

        f32x4 acc[16][2];
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // components xi = 4 i + j, i = i0, i0 + 1, out of the filter half Uh. t = B^T d for these two rows comes from three
        // patch rows (two transformed rows at a time halve the registers held through the products); `issue(s)` is called
        // in step s = 0..7 and puts the DMA pieces for the other buffers between the products (a piece costs 60 - 180 issue
        // cycles: back to back after a barrier both waves of a SIMD would leave the matrix pipe idle for all of them).
        auto half = [&](const f32x4* Uh, int g, auto I0, auto issue) {
            constexpr int i0 = decltype(I0)::value;
            f32x4 tt[2][4];
            {
                const f32x4* Pp = P + (4 * g + q) * P_PLANE + (2 * ty + (i0 ? 1 : 0)) * P_ROW + tx;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const f32x4* c = Pp + (bb & 1) * P_PAR + (bb >> 1);
                    const f32x4 e0 = c[0], e1 = c[P_ROW], e2 = c[2 * P_ROW];
                    if (i0 == 0) {                   // rows 0, 1 of B^T d from d0, d1, d2
                        tt[0][bb] = e0 - e2;
                        tt[1][bb] = e1 + e2;
                    } else {                         // rows 2, 3 from d1, d2, d3
                        tt[0][bb] = e1 - e0;
                        tt[1][bb] = e0 - e2;
                    }
                }
            }
            const f32x4* up = Uh + q * 64 + ucol;
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = i0 + ii;
                f32x4 V[4];
                V[0] = tt[ii][0] - tt[ii][2];
                V[1] = tt[ii][1] + tt[ii][2];
                V[2] = tt[ii][2] - tt[ii][1];
                V[3] = tt[ii][1] - tt[ii][3];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int xi8 = ii * 4 + j;
                    const f32x4 u0 = up[xi8 * 256], u1 = up[xi8 * 256 + 16];
                    issue(xi8);
                    // the two accumulators alternate: a dependent v_mfma_f32_16x16x4_f32 issues after 40 cycles, an
                    // independent one after 32
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        acc[4 * i + j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u0[c], V[j][c], acc[4 * i + j][0], 0, 0, 0);
                        acc[4 * i + j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u1[c], V[j][c], acc[4 * i + j][1], 0, 0, 0);
                    }
                }
            }
        };

#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // half A; meanwhile the B half of this group lands (g = 0: it came with the previous item / the prologue) and
            // the patch planes of the previous channel group, which no wave reads again, are refilled for the next item
            half(UA, g, std::integral_constant<int, 0>{}, [&](int s) {
                if (g == 0) return;
                if (s < 4) u_piece(UB, 1, g, cur.co0, s);
                else if (s < 7 && has_next) patch_piece(g - 1, s - 4, nxt);
            });
            if (g > 0 && has_next) W2_BARRIER(3)     // every wave is done with A; B has landed; the patch pieces may fly
            else W2_BARRIER(0)
            // half B; meanwhile the A half of the next group (or of the next item) lands
            half(UB, g, std::integral_constant<int, 2>{}, [&](int s) {
                if (s >= 4) return;
                if (g < 3) u_piece(UA, 0, g + 1, cur.co0, s);
                else if (has_next) u_piece(UA, 0, 0, nxt.co0, s);
            });
            W2_BARRIER(0);
        }
        if (has_next) {
#pragma unroll
            for (int k = 0; k < 4; ++k) u_piece(UB, 1, 0, nxt.co0, k);
#pragma unroll
            for (int j = 0; j < 3; ++j) patch_piece(3, j, nxt);
        }

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
        if (!has_next) break;
        item = next;
        cur = nxt;
        // the next item's patch and first filter group have landed: they were issued before this item's result stores (16
        // per lane; wave 0 adds the 2 statistics stores), and vector memory operations complete in order
        W2_BARRIER(16);
    }
}

}  // namespace

// geometry this kernel covers
extern "C" int denet_conv_wino2f_ok(int N, int H, int W, int Ci, int Co) {
    return (Ci == 64 && Co > 0 && Co % 64 == 0 && H > 0 && W > 0 && H % 16 == 0 && W % 16 == 0 && N > 0 &&
            (long)N * H * W * 64 * 4 < 0xF0000000L) ? 1 : 0;
}

// y = conv3x3(x) stride 1 pad 1 (+ bias) (+ add) from the F(2x2) transformed filters u = [16][Co][64]
// (denet_conv_wino_filter with tile 2: dgrad = 0 for the forward pass, 1 for the data gradient, where x = dy, Co = C).
// stats_partial (optional): [N*(H/16)*(W/16)][2][Co] doubles, the batch-norm column sums of y (see denet_conv_fwd_stats).
extern "C" int denet_conv_wino2f(const float* x, const float* u, const float* bias, const float* add, float* y,
                                 double* stats_partial, size_t stats_bytes, int* stats_rows, int N, int H, int W, int Ci,
                                 int Co, hipStream_t stream) {
    DENET_CHECK_ARG(x && u && y, "conv_wino2f: null pointer");
    DENET_CHECK_ARG(denet_conv_wino2f_ok(N, H, W, Ci, Co), "conv_wino2f: needs Ci = 64, Co %% 64 = 0, H, W multiples of 16");
    W2Params p = {};
    p.x = x; p.U = u; p.bias = bias; p.add = add; p.y = y;
    p.N = N; p.H = H; p.W = W; p.Co = Co;
    p.by = H / 16; p.bx = W / 16;
    p.nco = Co / 64;
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
        hipError_t e = hipFuncSetAttribute((const void*)wino2f_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
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
    hipLaunchKernelGGL(wino2f_kernel, dim3((unsigned)grid), dim3(512), LDS_BYTES, stream, p);
    DENET_CHECK_LAUNCH("conv_wino2f");
    return DENET_OK;
}
