// Directed-sparse-sampling kernels of the DeNet hot path (NHWC fp32, gfx950).
//   corner log-softmax + corner NLL cost/gradient   reference denet/layer/denet_corner.py:50-53, 126-134
//   sparse RoI feature gather (k_sparse_sample)      reference denet/layer/denet_sparse_op.py:42-85 and the
//                                                    Theano fallback denet/layer/denet_sparse.py:70-96
//   its gradient (k_sparse_sample_grad, atomicAdd)   reference denet/layer/denet_sparse_op.py:171-212
//   detection cost/gradient                          reference denet/layer/denet_detect.py:238-313,
//                                                    denet/common/theano_util.py:27-34
// This file is compiled with -ffp-contract=off: tap indices must be reproducible bit for bit.
#include "common.h"
#include <stdlib.h>
#include <math.h>

namespace {

constexpr double LN2 = 0.6931471805599453;

int grid_for(long total) {
    long b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

// ---------------------------------------------------------------------------------------------
// corner map: conv[b,y,x,ci] -> corner_pr[b,k,ci,y,x], k=0: log P(no corner) from logit +x,
// k=1: log P(corner) from logit -x   (denet_corner.py:52-53: lh = [x, -x], log_softmax(axis=1))
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void corner_fwd_kernel(const float* __restrict__ conv, float* __restrict__ pr, int B,
                                                         int H, int W, int CP, int Cn) {
    const long total = (long)B * H * W;
    const long plane = (long)H * W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / plane, yx = i - b * plane;
        for (int ci = 0; ci < Cn; ++ci) {
            const float x = conv[i * CP + ci];
            const float mx = fmaxf(x, -x);
            const float d0 = x - mx, d1 = -x - mx;
            const float ls = logf(expf(d0) + expf(d1));
            pr[((b * 2 + 0) * Cn + ci) * plane + yx] = d0 - ls;
            pr[((b * 2 + 1) * Cn + ci) * plane + yx] = d1 - ls;
        }
    }
}

// cost = -sum(target*logpr)/B/ln2*cost_factor ; dconv[b,y,x,ci] = scale*((t0+t1)(p0-p1) - (t0-t1))
__global__ __launch_bounds__(256) void corner_loss_kernel(const float* __restrict__ pr, const float* __restrict__ tgt,
                                                          float* __restrict__ dconv, double* __restrict__ partial,
                                                          int B, int H, int W, int CP, int Cn, float scale) {
    __shared__ double red[256];
    const long total = (long)B * H * W;
    const long plane = (long)H * W;
    double acc = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / plane, yx = i - b * plane;
        for (int ci = 0; ci < Cn; ++ci) {
            const long o0 = ((b * 2 + 0) * Cn + ci) * plane + yx;
            const long o1 = ((b * 2 + 1) * Cn + ci) * plane + yx;
            const float l0 = pr[o0], l1 = pr[o1];
            const float t0 = tgt[o0], t1 = tgt[o1];
            acc += (double)t0 * (double)l0 + (double)t1 * (double)l1;
            if (dconv) {
                const float p0 = expf(l0), p1 = expf(l1);
                dconv[i * CP + ci] = scale * ((t0 + t1) * (p0 - p1) - (t0 - t1));
            }
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// out[slot] = factor * sum(partial[0..n))
__global__ void finish_sum_kernel(const double* __restrict__ partial, int n, double factor, float* __restrict__ out) {
    __shared__ double red[256];
    double acc = 0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = (float)(factor * red[0]);
}

// ---------------------------------------------------------------------------------------------
// sparse RoI gather
// ---------------------------------------------------------------------------------------------
// tap rule 0 ("theano"): denet_sparse.py:72-84  p = p0 + (i*extent)/(gs-1) ; round half to even
// tap rule 1 ("cuda")  : denet_sparse_op.py:65-71 p = p0 + i*extent*(1/(gs-1)) ; lroundf (half away)
__device__ __forceinline__ int tap_index(float p0, float extent, int i, int gs, int size, int rule) {
    float p;
    if (rule == 0) {
        p = p0 + ((float)i * extent) / (float)(gs - 1);
    } else {
        const float k = 1.0f / (float)(gs - 1);
        p = p0 + ((float)i * extent) * k;
    }
    float f = p * (float)size;
    f = fmaxf(0.0f, fminf(f, (float)size - 1.0f));
    return (rule == 0) ? (int)rintf(f) : (int)lroundf(f);
}

template <bool VEC>
__global__ __launch_bounds__(256) void sparse_fwd_kernel(const float* __restrict__ fmap, const float* __restrict__ bbox,
                                                         float* __restrict__ out, int* __restrict__ taps, int H, int W,
                                                         int CP, int coff, int F, int rois_per_image, int gs, int KP,
                                                         int rule) {
    __shared__ int s_cell[256];
    const int m = blockIdx.x;
    const int b = m / rois_per_image;
    const int ntap = gs * gs;
    const float x0 = bbox[m * 4 + 0], y0 = bbox[m * 4 + 1], x1 = bbox[m * 4 + 2], y1 = bbox[m * 4 + 3];
    const float bw = x1 - x0, bh = y1 - y0;
    if ((int)threadIdx.x < ntap) {
        const int yi = threadIdx.x / gs, xi = threadIdx.x - yi * gs;
        const int ys = tap_index(y0, bh, yi, gs, H, rule);
        const int xs = tap_index(x0, bw, xi, gs, W, rule);
        const int cell = ys * W + xs;
        s_cell[threadIdx.x] = cell;
        if (taps) taps[(long)m * ntap + threadIdx.x] = cell;
    }
    __syncthreads();
    float* o = out + (long)m * KP;
    const float* fb = fmap + (long)b * H * W * CP + coff;
    if (VEC) {
        const int F4 = F / 4;
        for (int idx = threadIdx.x; idx < ntap * F4; idx += 256) {
            const int tap = idx / F4, f4 = idx - tap * F4;
            *(f32x4*)(o + (long)idx * 4) = *(const f32x4*)(fb + (long)s_cell[tap] * CP + f4 * 4);
        }
    } else {
        for (int idx = threadIdx.x; idx < ntap * F; idx += 256) {
            const int tap = idx / F, f = idx - tap * F;
            o[idx] = fb[(long)s_cell[tap] * CP + f];
        }
    }
    // denet_sparse_op.py:83-84: channel gs*gs*F = box height, +1 = box width; the rest is K padding
    for (int k = ntap * F + threadIdx.x; k < KP; k += 256)
        o[k] = (k == ntap * F) ? bh : (k == ntap * F + 1) ? bw : 0.f;
}

// ---- gather gradient, step 1: the tap list of every image grouped by feature-map cell (counting sort) --------------
// The reference scatters with atomicAdd (denet_sparse_op.py:171-212); here the (roi, tap) slots that sampled a cell are
// listed per cell in ASCENDING SLOT ORDER and summed in that order: deterministic, no floating-point atomics.
// A slot is s = roi * ntap + tap (n = rois_per_image * ntap per image); slots are cut into chunks of SORT_CHUNK; one
// workgroup owns one chunk (B x nchunk of them fill the chip - the per-image LDS bitonic sort this replaces ran on B workgroups):
//   sparse_count    chunk histogram over the cells (LDS integer atomics)         -> table[b][chunk][cell]
//   sparse_offsets  thread per (image, cell): the cell's slots in the chunks in front of each chunk -> table_pos, its total
//   sparse_scan     workgroup per image: exclusive scan of the cell totals -> cell_start[b][0..HW]
//   sparse_scatter  one wave walks the chunk in slot order, 64 slots at a time; the lanes holding the same cell are
//                   found with one ballot per key bit (no match-any instruction on gfx950), a lane's rank among them is
//                   a popcount below its lane id, the lowest lane advances the cell's LDS cursor: stable by construction
constexpr int SORT_CHUNK = 2048;

// workgroup (chunk, image), four waves: the histogram's 4 HW bytes of LDS are cleared and written out by all of them (one wave
// took 2 x HW / 64 dependent rounds for it: 1.2 ms on the 128 x 128 map of DeNet-101 wide)
__global__ __launch_bounds__(256) void sparse_count_kernel(const int* __restrict__ taps, int* __restrict__ table, int n,
                                                           int HW, int nchunk) {
    extern __shared__ int s_hist[];
    const int b = blockIdx.y, chunk = blockIdx.x;
    for (int c = threadIdx.x; c < HW; c += 256) s_hist[c] = 0;
    __syncthreads();
    const int lo = chunk * SORT_CHUNK, hi = min(n, lo + SORT_CHUNK);
    const int* t = taps + (long)b * n;
    for (int i = lo + threadIdx.x; i < hi; i += 256) atomicAdd(&s_hist[t[i]], 1);
    __syncthreads();
    int* row = table + ((long)b * nchunk + chunk) * HW;
    for (int c = threadIdx.x; c < HW; c += 256) row[c] = s_hist[c];
}

// thread (image, cell): the cell's slots in the chunks in front of chunk k -> table_pos[k][cell]; the cell's total -> cell_start
// (sparse_scan_kernel turns the totals into starts)
__global__ __launch_bounds__(256) void sparse_offsets_kernel(const int* __restrict__ table, int* __restrict__ table_pos,
                                                            int* __restrict__ cell_start, int HW, int nchunk) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= HW) return;
    const int* tb = table + (long)b * nchunk * HW + c;
    int* tp = table_pos + (long)b * nchunk * HW + c;
    int run = 0;
    for (int k = 0; k < nchunk; ++k) {
        const int cnt = tb[(long)k * HW];
        tp[(long)k * HW] = run;
        run += cnt;
    }
    cell_start[(long)b * (HW + 1) + c] = run;
}

// workgroup per image: cell_start[0..HW) from totals to their exclusive prefix sums, cell_start[HW] = n
__global__ __launch_bounds__(1024) void sparse_scan_kernel(int* __restrict__ cell_start, int HW, int n) {
    __shared__ int s_wave[16];
    __shared__ int s_carry;
    int* cs = cell_start + (long)blockIdx.x * (HW + 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int c0 = 0; c0 < HW; c0 += 1024) {
        const int c = c0 + tid;
        const int v = c < HW ? cs[c] : 0;
        int inc = v;                                    // inclusive scan over the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int u = __shfl_up(inc, off, 64);
            if (lane >= off) inc += u;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        int before = s_carry;
        for (int w = 0; w < wave; ++w) before += s_wave[w];
        if (c < HW) cs[c] = before + inc - v;
        __syncthreads();
        if (tid == 1023) s_carry = before + inc;
        __syncthreads();
    }
    if (tid == 0) cs[HW] = n;
}

// workgroup (chunk, image): all four waves fill the chunk's cursors (first output position of (chunk, cell)), wave 0 then walks
// the chunk in slot order
__global__ __launch_bounds__(256) void sparse_scatter_kernel(const int* __restrict__ taps, const int* __restrict__ table,
                                                             const int* __restrict__ cell_start, int* __restrict__ order,
                                                             int n, int HW, int nchunk, int key_bits) {
    extern __shared__ int s_cur[];
    const int b = blockIdx.y, chunk = blockIdx.x, lane = threadIdx.x & 63;
    const int lo = chunk * SORT_CHUNK, hi = min(n, lo + SORT_CHUNK);
    const int* t = taps + (long)b * n;
    // the whole chunk's cells first: SORT_CHUNK / 64 independent loads in flight per lane (the walk below is serial)
    int cells[SORT_CHUNK / 64];
    if (threadIdx.x < 64) {
#pragma unroll
        for (int r = 0; r < SORT_CHUNK / 64; ++r) {
            const int i = lo + r * 64 + lane;
            cells[r] = (i < hi) ? t[i] : 0;
        }
    }
    const int* row = table + ((long)b * nchunk + chunk) * HW;
    const int* cs = cell_start + (long)b * (HW + 1);
    for (int c = threadIdx.x; c < HW; c += 256) s_cur[c] = cs[c] + row[c];
    __syncthreads();
    if (threadIdx.x >= 64) return;
    int* o = order + (long)b * n;
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < SORT_CHUNK / 64; ++r) {
        const int i = lo + r * 64 + lane;
        const bool live = i < hi;
        const int cell = cells[r];
        unsigned long long same = __ballot(live);
        for (int bit = 0; bit < key_bits; ++bit) {
            const unsigned long long set = __ballot((cell >> bit) & 1);
            same &= ((cell >> bit) & 1) ? set : ~set;
        }
        if (!live) same = 0;
        const int rank = __popcll(same & below);
        int start = 0;
        if (live && rank == 0) {                        // lowest lane of its group: distinct cells, no conflict
            start = s_cur[cell];
            s_cur[cell] = start + __popcll(same);
        }
        const int leader = live ? (__ffsll((long long)same) - 1) : lane;
        start = __shfl(start, leader, 64);
        if (live) o[start + rank] = i;
    }
}

// gather gradient, step 2: one wave per feature-map cell sums the dy rows of every (roi, tap) that sampled this cell, in
// ascending slot order (deterministic replacement for the reference's atomicAdd scatter)
__global__ __launch_bounds__(256) void sparse_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ order,
                                                         const int* __restrict__ cell_start,
                                                         float* __restrict__ dfmap, int HW, int CP, int coff, int F,
                                                         int rois_per_image, int ntap, int KP, int zero_from,
                                                         long ncell_total) {
    const int lane = threadIdx.x & 63;
    const long cellg = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cellg >= ncell_total) return;
    const int b = (int)(cellg / HW);
    const int cell = (int)(cellg - (long)b * HW);
    const int n = rois_per_image * ntap;
    const int F4 = F / 4;
    const int epi = 64 / F4;  // entries per iteration
    const int e = lane / F4, f4 = lane - e * F4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int* cs = cell_start + (long)b * (HW + 1);
    const int start = cs[cell], end = cs[cell + 1];
    const int* ord = order + (long)b * n;
    // entry j of the cell belongs to lane group j % epi: a fixed left-to-right chain per group. The slots of up to CH entries come
    // in with ONE coalesced load and are handed out by shuffles, and a group has the rows of four entries in flight before it
    // adds them (in order): slot -> address -> row was two dependent loads per entry, 3-4 times in a row per wave - the kernel
    // ran at the latency of that chain (281 us alone, 0.84 ms beside the first head layer's filter gradient)
    const int CH = (64 / epi) * epi;
    const bool grp = e < epi;
    const float* dyb = dy + (long)b * rois_per_image * KP + f4 * 4;
    for (int base = start; base < end; base += CH) {
        const int cnt = min(CH, end - base);
        const int mine = (lane < cnt) ? ord[base + lane] : 0;
        for (int j0 = 0; j0 < cnt; j0 += 4 * epi) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * epi + e;
                const int slot = __shfl(mine, j & 63, 64);
                const int roi = slot / ntap, tap = slot - roi * ntap;
                v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (grp && j < cnt) v[u] = *(const f32x4*)(dyb + (long)roi * KP + (long)tap * F);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (grp && j0 + u * epi + e < cnt) acc += v[u];
        }
    }
    f32x4 tot = acc;
    for (int k = 1; k < epi; ++k) {
#pragma unroll
        for (int c = 0; c < 4; ++c) tot[c] += __shfl(acc[c], f4 + k * F4, 64);
    }
    float* o = dfmap + cellg * CP;
    if (lane < F4) {
        if ((coff & 3) == 0) {
            *(f32x4*)(o + coff + lane * 4) = tot;
        } else {      // DNC.C: five corner channels in front of the features, the slice is not 16-byte aligned
            float* q = o + coff + lane * 4;
            q[0] = tot[0]; q[1] = tot[1]; q[2] = tot[2]; q[3] = tot[3];
        }
    }
    // zero the physical padding channels of this cell
    for (int c = zero_from + lane; c < CP; c += 64) o[c] = 0.f;
}

// ---------------------------------------------------------------------------------------------
// detection cost: one wave per RoI
//   det_err  = -sum_c t_c*logsoftmax(z)_c / ln(ncls)                       denet_detect.py:257
//   bbox_err = bbox_factor*valid*sum_k smoothL1(t_k - reg_k)                 denet_detect.py:289-295
//   cost     = cost_factor*sum(det_err)/B + bbox_factor*sum(bbox_err)/B      denet_detect.py:304-313
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void detect_loss_kernel(const float* __restrict__ logits, const float* __restrict__ det_t,
                                                          const float* __restrict__ bbox_valid,
                                                          const float* __restrict__ bbox_t, float* __restrict__ dlogits,
                                                          double* __restrict__ partial, int M, int CP, int ncls,
                                                          int nreg, float det_scale, float bbox_factor,
                                                          float bbox_scale, int bounded_iou,
                                                          const float* __restrict__ roi_bbox,
                                                          const float* __restrict__ fit_t, int nfit, float fit_factor,
                                                          float fit_scale) {
    __shared__ double red[2][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int m = blockIdx.x * 4 + wv;
    double derr = 0, berr = 0;
    if (m < M) {
        const float* z = logits + (long)m * CP;
        float* dz = dlogits ? dlogits + (long)m * CP : nullptr;
        const float* t = det_t + (long)m * ncls;
        float mx = -INFINITY;
        for (int c = lane; c < ncls; c += 64) mx = fmaxf(mx, z[c]);
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float se = 0.f, tsum = 0.f;
        for (int c = lane; c < ncls; c += 64) {
            se += expf(z[c] - mx);
            tsum += t[c];
        }
        for (int o = 32; o > 0; o >>= 1) {
            se += __shfl_xor(se, o, 64);
            tsum += __shfl_xor(tsum, o, 64);
        }
        const float lse = logf(se);
        const float inv_logn = (float)(1.0 / log((double)ncls));
        float e = 0.f;
        for (int c = lane; c < ncls; c += 64) {
            const float lp = (z[c] - mx) - lse;
            e += t[c] * lp;
            if (dz) dz[c] = det_scale * inv_logn * (tsum * expf(lp) - t[c]);
        }
        for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o, 64);
        derr = -(double)e * (double)inv_logn;
        if (nreg > 0) {
            float sl = 0.f;
            if (lane < 4) {
                const float valid = bbox_valid[m];
                const float* bt = bbox_t + (long)m * 8;
                const float reg = z[ncls + lane];
                float d = 0.f, dd_dreg = -1.f;   // d = residual fed to smooth L1 ; dd_dreg = d(d)/d(reg)
                if (!bounded_iou) {
                    // Fast R-CNN targets: (tcx-scx)/sw, (tcy-scy)/sh, ln(tw/sw), ln(th/sh)
                    float tk;
                    if (lane < 2) tk = (bt[lane] - bt[4 + lane]) / bt[6 + lane];
                    else tk = logf(bt[lane] / bt[4 + lane]);
                    d = tk - reg;
                    dd_dreg = -1.f;
                } else {
                    // bounded IoU (denet_detect.py:266-286); prediction decoded from the RoI box
                    const float* rb = roi_bbox + (long)m * 4;
                    const float scx = 0.5f * (rb[0] + rb[2]), scy = 0.5f * (rb[1] + rb[3]);
                    const float sw = rb[2] - rb[0], sh = rb[3] - rb[1];
                    const float eps = 0.001f;
                    if (lane < 2) {
                        const float sc = lane == 0 ? scx : scy, se_ = lane == 0 ? sw : sh;
                        // predict_x = 0.5*((cx - w/2) + (cx + w/2)) -> cx = reg*extent + centre
                        const float pc = reg * se_ + sc;
                        const float dxy = bt[lane] - pc;
                        const float tw = bt[2 + lane];
                        if (dxy >= 0.f) {
                            const float den = tw + dxy + eps;
                            d = 2.f * dxy / den;
                            // d/d(dxy) = 2*(tw+eps)/den^2 ; d(dxy)/dreg = -extent
                            dd_dreg = 2.f * (tw + eps) / (den * den) * (-se_);
                        } else {
                            const float den = tw - dxy + eps;
                            d = -2.f * dxy / den;
                            dd_dreg = -2.f * (tw + eps) / (den * den) * (-se_);
                        }
                    } else {
                        const float se_ = lane == 2 ? sw : sh;
                        const float pw = expf(reg) * se_;
                        const float tw = bt[lane];
                        const float a = tw / (pw + eps), c2 = pw / (tw + eps);
                        if (a <= c2) {
                            d = 1.f - a;
                            dd_dreg = tw / ((pw + eps) * (pw + eps)) * pw;   // d(1-a)/dpw * dpw/dreg
                        } else {
                            d = 1.f - c2;
                            dd_dreg = -pw / (tw + eps);
                        }
                    }
                }
                const float ad = fabsf(d);
                sl = (ad < 1.f) ? 0.5f * d * d : ad - 0.5f;
                const float dsl = (ad < 1.f) ? d : (d > 0.f ? 1.f : -1.f);
                if (dz) dz[ncls + lane] = bbox_scale * valid * dsl * dd_dreg;
                sl *= valid;
            }
            for (int o = 2; o > 0; o >>= 1) sl += __shfl_xor(sl, o, 64);
            berr = (double)bbox_factor * (double)bbox_factor * (double)sl;     // factor of get_errors :295 and of cost :310
        }
        if (nfit > 0) {
            // independent fitness distribution (denet_detect.py:103-108, 298-301): a second soft-target cross entropy
            // over the nfit logits behind the box regressors
            const float* zf = z + ncls + nreg;
            const float* tf = fit_t + (long)m * nfit;
            const float zv = lane < nfit ? zf[lane] : -INFINITY;
            float fm = zv;
            for (int o = 32; o > 0; o >>= 1) fm = fmaxf(fm, __shfl_xor(fm, o, 64));
            float fe = lane < nfit ? expf(zv - fm) : 0.f, ft = lane < nfit ? tf[lane] : 0.f;
            float fse = fe, fts = ft;
            for (int o = 32; o > 0; o >>= 1) {
                fse += __shfl_xor(fse, o, 64);
                fts += __shfl_xor(fts, o, 64);
            }
            const float flse = logf(fse);
            const float finv = (float)(1.0 / log((double)nfit));
            const float flp = lane < nfit ? (zv - fm) - flse : 0.f;
            float fsum = ft * flp;
            if (dz && lane < nfit) dz[ncls + nreg + lane] = fit_scale * finv * (fts * expf(flp) - ft);
            for (int o = 32; o > 0; o >>= 1) fsum += __shfl_xor(fsum, o, 64);
            berr += (double)fit_factor * (-(double)fsum * (double)finv);
        }
        if (dz) {
            for (int c = ncls + nreg + nfit + lane; c < CP; c += 64) dz[c] = 0.f;
        }
    }
    if (lane == 0) {
        red[0][wv] = derr;
        red[1][wv] = berr;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        partial[gridDim.x + blockIdx.x] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

}  // namespace

extern "C" int denet_corner_fwd(const float* conv, float* corner_pr, int B, int H, int W, int CP, int Cn,
                                hipStream_t stream) {
    DENET_CHECK_ARG(conv && corner_pr && Cn > 0 && Cn <= CP, "corner_fwd: bad args");
    hipLaunchKernelGGL(corner_fwd_kernel, dim3(grid_for((long)B * H * W)), dim3(256), 0, stream, conv, corner_pr, B, H,
                       W, CP, Cn);
    DENET_CHECK_LAUNCH("corner_fwd");
    return DENET_OK;
}

extern "C" size_t denet_loss_workspace_bytes(void) { return (size_t)2 * 65536 * sizeof(double); }

extern "C" int denet_corner_loss(const float* corner_pr, const float* target, float* dconv, float* cost,
                                 void* workspace, int B, int H, int W, int CP, int Cn, float cost_factor,
                                 hipStream_t stream) {
    DENET_CHECK_ARG(corner_pr && target && cost && workspace && Cn > 0, "corner_loss: bad args");
    int g = grid_for((long)B * H * W);
    if (g > 1024) g = 1024;
    const float scale = (float)((double)cost_factor / ((double)B * LN2));
    hipLaunchKernelGGL(corner_loss_kernel, dim3(g), dim3(256), 0, stream, corner_pr, target, dconv, (double*)workspace,
                       B, H, W, CP, Cn, scale);
    hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, stream, (const double*)workspace, g,
                       -(double)cost_factor / ((double)B * LN2), cost);
    DENET_CHECK_LAUNCH("corner_loss");
    return DENET_OK;
}

extern "C" int denet_sparse_fwd(const float* fmap, const float* bbox, float* out, int* taps, int B, int H, int W,
                                int CP, int coff, int F, int rois_per_image, int gs, int KP, int tap_rule,
                                hipStream_t stream) {
    DENET_CHECK_ARG(fmap && bbox && out, "sparse_fwd: null pointer");
    DENET_CHECK_ARG(gs >= 2 && gs * gs <= 256, "sparse_fwd: grid size %d unsupported", gs);
    DENET_CHECK_ARG(coff + F <= CP && KP >= gs * gs * F + 2, "sparse_fwd: channel layout inconsistent");
    DENET_CHECK_ARG(tap_rule == 0 || tap_rule == 1, "sparse_fwd: tap_rule must be 0 (theano) or 1 (cuda)");
    const int M = B * rois_per_image;
    const bool vec = (coff % 4 == 0) && (F % 4 == 0) && (CP % 4 == 0) && (KP % 4 == 0);
    if (vec)
        hipLaunchKernelGGL(sparse_fwd_kernel<true>, dim3(M), dim3(256), 0, stream, fmap, bbox, out, taps, H, W, CP, coff,
                           F, rois_per_image, gs, KP, tap_rule);
    else
        hipLaunchKernelGGL(sparse_fwd_kernel<false>, dim3(M), dim3(256), 0, stream, fmap, bbox, out, taps, H, W, CP,
                           coff, F, rois_per_image, gs, KP, tap_rule);
    DENET_CHECK_LAUNCH("sparse_fwd");
    return DENET_OK;
}

// The three kernels above as ONE, a 1024-thread workgroup per image, for feature maps of up to 4096 cells and up to 65 535 slots
// per image (DeNet-34: 4096 cells, 28 224 slots): the (chunk, cell) table of a whole image - 16 chunks, one per wave, 16-bit
// counts / positions - is 128 KB of LDS, so counting, the scan and the stable scatter never leave the CU and the 32 images of a
// batch are 32 workgroups instead of 2 x 448 single-wave ones + 512 (which, queued beside the head's matrix kernels, held slots
// for over a millisecond). Same lists, same order: chunk-major, slot order inside a chunk.
constexpr int SORT1_WAVES = 16;
__global__ __launch_bounds__(1024) void sparse_sort_image_kernel(const int* __restrict__ taps, int* __restrict__ order,
                                                                 int* __restrict__ cell_start, int n, int HW, int key_bits) {
    extern __shared__ unsigned s_tab32[];                    // [16][HWP / 2]: two 16-bit entries per word
    __shared__ int s_scan[1024];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int HWP = (HW + 1) & ~1;
    unsigned short* tab = (unsigned short*)s_tab32;
    for (int i = tid; i < SORT1_WAVES * HWP / 2; i += 1024) s_tab32[i] = 0u;
    __syncthreads();
    const int chunk = (((n + SORT1_WAVES - 1) / SORT1_WAVES) + 63) & ~63;
    const int lo = wave * chunk, hi = min(n, lo + chunk);
    const int* t = taps + (long)b * n;
    for (int i = lo + lane; i < hi; i += 64) {
        const int cell = t[i];
        atomicAdd(&s_tab32[(wave * HWP + cell) >> 1], 1u << (((wave * HWP + cell) & 1) * 16));
    }
    __syncthreads();
    // cell totals over the chunks, exclusive scan over the cells (thread = CPT consecutive cells), positions back into the table
    const int CPT = (HW + 1023) / 1024;
    int tot = 0;
    for (int k = 0; k < CPT; ++k) {
        const int c = tid * CPT + k;
        if (c < HW)
            for (int w = 0; w < SORT1_WAVES; ++w) tot += tab[w * HWP + c];
    }
    s_scan[tid] = tot;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = (tid >= off) ? s_scan[tid - off] : 0;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    int pos = s_scan[tid] - tot;
    int* cs = cell_start + (long)b * (HW + 1);
    for (int k = 0; k < CPT; ++k) {
        const int c = tid * CPT + k;
        if (c < HW) {
            cs[c] = pos;
            for (int w = 0; w < SORT1_WAVES; ++w) {
                const int cnt = tab[w * HWP + c];
                tab[w * HWP + c] = (unsigned short)pos;
                pos += cnt;
            }
        }
    }
    if (tid == 0) cs[HW] = n;
    __syncthreads();
    // stable scatter of this wave's chunk: 64 slots at a time, lanes of the same cell found with one ballot per key bit
    int* o = order + (long)b * n;
    unsigned short* cur = tab + wave * HWP;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int i0 = lo; i0 < hi; i0 += 64) {
        const int i = i0 + lane;
        const bool live = i < hi;
        const int cell = live ? t[i] : 0;
        unsigned long long same = __ballot(live);
        for (int bit = 0; bit < key_bits; ++bit) {
            const unsigned long long set = __ballot((cell >> bit) & 1);
            same &= ((cell >> bit) & 1) ? set : ~set;
        }
        if (!live) same = 0;
        const int rank = __popcll(same & below);
        int start = 0;
        if (live && rank == 0) {                        // lowest lane of its group: distinct cells, no conflict
            start = cur[cell];
            cur[cell] = (unsigned short)(start + __popcll(same));
        }
        const int leader = live ? (__ffsll((long long)same) - 1) : lane;
        start = __shfl(start, leader, 64);
        if (live) o[start + rank] = i;
    }
}

// Workspace of denet_sparse_sort / denet_sparse_bwd: order [B][n] | cell_start [B][HW+1] | counts [B][nchunk][HW] |
// positions [B][nchunk][HW] (int32)
namespace {
struct SortLayout {
    int n, HW, nchunk, key_bits;
    size_t off_start, off_table, total;   // in ints
};
int sort_layout(int B, int H, int W, int rois_per_image, int gs, SortLayout* L) {
    DENET_CHECK_ARG(B > 0 && H > 0 && W > 0 && rois_per_image > 0 && gs > 0, "sparse_sort: bad arguments");
    DENET_CHECK_ARG((long)H * W <= 32768, "sparse_sort: feature map of %d x %d cells exceeds the 32768-cell LDS cursor", H, W);
    L->n = rois_per_image * gs * gs;
    L->HW = H * W;
    L->nchunk = (L->n + SORT_CHUNK - 1) / SORT_CHUNK;
    L->key_bits = 1;
    while ((1 << L->key_bits) < L->HW) L->key_bits++;
    L->off_start = (size_t)B * L->n;
    L->off_table = L->off_start + (size_t)B * (L->HW + 1);
    L->total = L->off_table + (size_t)2 * B * L->nchunk * L->HW;      // counts | positions
    return DENET_OK;
}
}  // namespace

extern "C" size_t denet_sparse_sort_workspace_bytes(int B, int H, int W, int rois_per_image, int gs) {
    SortLayout L;
    if (sort_layout(B, H, W, rois_per_image, gs, &L)) return 0;
    return L.total * sizeof(int);
}

namespace {
// the one-kernel form (sparse_sort_image_kernel: one 1024-thread workgroup per image) serves this problem
bool sort_is_single(const SortLayout& L) {
    static const int one_kernel = [] { const char* e = getenv("DENET_SORT_ONE_KERNEL"); return e ? atoi(e) : 1; }();
    return one_kernel && L.HW <= 4096 && L.n <= 65535;
}
}  // namespace

// 1 if denet_sparse_sort runs as ONE kernel for this problem (the host then keeps it on the compute stream: alone it takes
// ~50 us, on a side stream its workgroups starve for LDS beside the head's matrix kernels), 0 for the three-kernel form
extern "C" int denet_sparse_sort_is_single(int B, int H, int W, int rois_per_image, int gs) {
    SortLayout L;
    if (sort_layout(B, H, W, rois_per_image, gs, &L)) return 0;
    return sort_is_single(L) ? 1 : 0;
}

// groups the tap list of every image by cell (see sparse_count_kernel); depends only on `taps`, so the host may queue it
// on a side stream right after the forward gather, off the critical path of the backward sweep
extern "C" int denet_sparse_sort(const int* taps, void* sort_ws, size_t sort_ws_bytes, int B, int H, int W,
                                 int rois_per_image, int gs, hipStream_t stream) {
    DENET_CHECK_ARG(taps && sort_ws, "sparse_sort: null pointer");
    SortLayout L;
    int rc = sort_layout(B, H, W, rois_per_image, gs, &L);
    if (rc) return rc;
    DENET_CHECK_ARG(sort_ws_bytes >= L.total * sizeof(int), "sparse_sort: workspace too small (%zu < %zu)", sort_ws_bytes,
                    L.total * sizeof(int));
    int* order = (int*)sort_ws;
    int* cell_start = order + L.off_start;
    int* table = order + L.off_table;
    const size_t lds = (size_t)L.HW * sizeof(int);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)sparse_count_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           32768 * 4);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)sparse_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4);
        if (e != hipSuccess) {
            denet_set_error("sparse_sort: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return -(int)e;
        }
        attr_set = true;
    }
    if (sort_is_single(L)) {
        const size_t lds1 = (size_t)SORT1_WAVES * ((L.HW + 1) & ~1) * sizeof(unsigned short);
        static bool attr1 = false;
        if (!attr1) {
            const hipError_t e = hipFuncSetAttribute((const void*)sparse_sort_image_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     SORT1_WAVES * 4096 * 2);
            if (e != hipSuccess) {
                denet_set_error("sparse_sort: hipFuncSetAttribute: %s", hipGetErrorString(e));
                return -(int)e;
            }
            attr1 = true;
        }
        hipLaunchKernelGGL(sparse_sort_image_kernel, dim3(B), dim3(1024), lds1, stream, taps, order, cell_start, L.n, L.HW, L.key_bits);
        DENET_CHECK_LAUNCH("sparse_sort");
        return DENET_OK;
    }
    hipLaunchKernelGGL(sparse_count_kernel, dim3(L.nchunk, B), dim3(256), lds, stream, taps, table, L.n, L.HW, L.nchunk);
    int* table_pos = table + (size_t)B * L.nchunk * L.HW;
    hipLaunchKernelGGL(sparse_offsets_kernel, dim3((L.HW + 255) / 256, B), dim3(256), 0, stream, table, table_pos, cell_start,
                       L.HW, L.nchunk);
    hipLaunchKernelGGL(sparse_scan_kernel, dim3(B), dim3(1024), 0, stream, cell_start, L.HW, L.n);
    hipLaunchKernelGGL(sparse_scatter_kernel, dim3(L.nchunk, B), dim3(256), lds, stream, taps, table_pos, cell_start, order, L.n,
                       L.HW, L.nchunk, L.key_bits);
    DENET_CHECK_LAUNCH("sparse_sort");
    return DENET_OK;
}

// taps == NULL: `sort_ws` already holds the lists written by denet_sparse_sort for the same tap list
extern "C" int denet_sparse_bwd(const float* dy, const int* taps, void* sort_ws, size_t sort_ws_bytes, float* dfmap, int B,
                                int H, int W, int CP, int coff, int F, int rois_per_image, int gs, int KP, int zero_from,
                                hipStream_t stream) {
    DENET_CHECK_ARG(dy && sort_ws && dfmap, "sparse_bwd: null pointer");
    const int ntap = gs * gs;
    DENET_CHECK_ARG(F % 4 == 0 && F / 4 <= 64 && CP % 4 == 0 && KP % 4 == 0,
                    "sparse_bwd: F/CP/KP must be multiples of 4 and F <= 256");
    DENET_CHECK_ARG(zero_from >= coff + F && zero_from <= CP, "sparse_bwd: zero_from out of range");
    SortLayout L;
    int rc = sort_layout(B, H, W, rois_per_image, gs, &L);
    if (rc) return rc;
    DENET_CHECK_ARG(sort_ws_bytes >= L.total * sizeof(int), "sparse_bwd: workspace too small");
    if (taps) {
        rc = denet_sparse_sort(taps, sort_ws, sort_ws_bytes, B, H, W, rois_per_image, gs, stream);
        if (rc) return rc;
    }
    const int* order = (const int*)sort_ws;
    const long ncell = (long)B * H * W;
    hipLaunchKernelGGL(sparse_bwd_kernel, dim3((unsigned)((ncell + 3) / 4)), dim3(256), 0, stream, dy, order,
                       order + L.off_start, dfmap, H * W, CP, coff, F, rois_per_image, ntap, KP, zero_from, ncell);
    DENET_CHECK_LAUNCH("sparse_bwd");
    return DENET_OK;
}

extern "C" int denet_detect_loss(const float* logits, const float* det_target, const float* bbox_valid,
                                 const float* bbox_target, const float* roi_bbox, const float* fit_target, float* dlogits,
                                 float* costs, void* workspace, int M, int batch, int CP, int ncls, int nreg, int nfit,
                                 float cost_factor, float bbox_factor, float fit_factor, int bounded_iou,
                                 hipStream_t stream) {
    DENET_CHECK_ARG(logits && det_target && costs && workspace, "detect_loss: null pointer");
    DENET_CHECK_ARG(nreg == 0 || nreg == 4, "detect_loss: nreg must be 0 or 4");
    DENET_CHECK_ARG(nreg == 0 || (bbox_valid && bbox_target), "detect_loss: bbox targets missing");
    DENET_CHECK_ARG(!bounded_iou || roi_bbox, "detect_loss: bounded IoU needs the RoI boxes");
    DENET_CHECK_ARG(nfit >= 0 && nfit <= 64 && (nfit == 0 || fit_target), "detect_loss: fitness targets missing / nfit > 64");
    DENET_CHECK_ARG(ncls + nreg + nfit <= CP, "detect_loss: CP too small");
    const int g = (M + 3) / 4;
    DENET_CHECK_ARG(g <= 65536, "detect_loss: too many RoIs (%d)", M);
    const float det_scale = cost_factor / (float)batch;
    // bbox_factor is applied twice in the reference: get_errors :295 and cost :310
    const float bbox_scale = bbox_factor * bbox_factor / (float)batch;
    hipLaunchKernelGGL(detect_loss_kernel, dim3(g), dim3(256), 0, stream, logits, det_target, bbox_valid, bbox_target,
                       dlogits, (double*)workspace, M, CP, ncls, nreg, det_scale, bbox_factor, bbox_scale, bounded_iou,
                       roi_bbox, fit_target, nfit, fit_factor, fit_factor / (float)batch);
    hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, stream, (const double*)workspace, g,
                       (double)cost_factor / (double)batch, costs);
    // costs[1] = box cost + independent-fitness cost (both already carry their factors)
    hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, stream, (const double*)workspace + g, g,
                       1.0 / (double)batch, costs + 1);
    DENET_CHECK_LAUNCH("detect_loss");
    return DENET_OK;
}
