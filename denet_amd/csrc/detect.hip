// Inference tail of the DeNet detector: decoding of the detection head and per-class NMS.
//   denet_detect_decode      reference denet/layer/denet_detect.py:76-100 (log-softmax over the class logits, box
//                            decoding from the regressors) and :330-349 (joint-fitness marginalisation)
//   denet_detect_nms         reference denet/layer/denet_detect.cc:99-173 build_detections_nms with hard NMS
//                            (:73-97 perform_nms): an instance is dropped iff a strictly better-scored instance of
//                            the same class overlaps it by more than the threshold
//   denet_soft_nms_host      reference denet/layer/denet_detect.cc:35-71 perform_soft_nms (Gaussian, log-domain
//                            scores, sequential by construction) — host code on the candidates kept by the GPU
// Compiled with -ffp-contract=off: IoU comparisons against a threshold must not depend on FMA contraction.
#include "common.h"
#include <math.h>
#include <string.h>
#include <vector>

namespace {

// one wave per RoI. logits [M,CP]: s0 class(-fitness) logits then 4 regressors.
// det_pr [M, class_num+1] log-probabilities, fitness [M, class_num+1], bbox [M,4] decoded boxes (or the RoI itself)
__global__ __launch_bounds__(256) void detect_decode_kernel(const float* __restrict__ logits,
                                                            const float* __restrict__ roi, float* __restrict__ det_pr,
                                                            float* __restrict__ fitness, float* __restrict__ bbox, int M,
                                                            int CP, int class_num, int fit_num, int jointfit, int nreg,
                                                            float t0, int nfit) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int s0 = jointfit ? class_num * fit_num + 1 : class_num + 1;
    const float* z = logits + (long)m * CP;
    float mx = -INFINITY;
    for (int c = lane; c < s0; c += 64) mx = fmaxf(mx, z[c]);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float se = 0.f;
    for (int c = lane; c < s0; c += 64) se += expf(z[c] - mx);
    for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o, 64);
    const float lse = logf(se);
    float* dp = det_pr + (long)m * (class_num + 1);
    float* fp = fitness + (long)m * (class_num + 1);
    if (!jointfit) {
        for (int c = lane; c < s0; c += 64) {
            const float lp = (z[c] - mx) - lse;
            dp[c] = lp;
            fp[c] = lp;      // denet_detect.py:388: fitness = copy(det_pr)
        }
    } else {
        // det_fit (class, f) log-probabilities -> det_pr[c] = logsumexp_f, fitness[c] = log sum_f exp(lp) * val[f],
        // val[f] = t0 + f*(1-t0)/fit_num (denet_detect.py:333-349)
        for (int c = lane; c < class_num; c += 64) {
            float lmax = -INFINITY;
            for (int f = 0; f < fit_num; ++f) lmax = fmaxf(lmax, (z[c * fit_num + f] - mx) - lse);
            float s = 0.f, sv = 0.f;
            for (int f = 0; f < fit_num; ++f) {
                const float lp = (z[c * fit_num + f] - mx) - lse;
                s += expf(lp - lmax);
                sv += expf(lp) * (t0 + (float)f * (1.0f - t0) / (float)fit_num);
            }
            dp[c] = lmax + logf(s);
            fp[c] = logf(sv);
        }
        if (lane == 0) {
            const float lp = (z[s0 - 1] - mx) - lse;
            dp[class_num] = lp;
            fp[class_num] = lp;
        }
    }
    if (nfit > 0) {
        // independent fitness head (denet_detect.py:396-401): fitness += log( sum_f P(f) * val[f] ),
        // val = [0, t0 + i*(1-t0)/(nfit-1)], the expectation in double, cast to float32, log in float32
        const float* zf = z + s0 + nreg;
        const float zv = lane < nfit ? zf[lane] : -INFINITY;
        float fm = zv;
        for (int o = 32; o > 0; o >>= 1) fm = fmaxf(fm, __shfl_xor(fm, o, 64));
        float fe = lane < nfit ? expf(zv - fm) : 0.f;
        float fse = fe;
        for (int o = 32; o > 0; o >>= 1) fse += __shfl_xor(fse, o, 64);
        const float lp = (zv - fm) - logf(fse);
        double term = 0.0;
        if (lane >= 1 && lane < nfit)
            term = (double)expf(lp) * ((double)t0 + (double)(lane - 1) * (1.0 - (double)t0) / (double)(nfit - 1));
        for (int o = 32; o > 0; o >>= 1) term += __shfl_xor(term, o, 64);
        const float add = logf((float)term);
        for (int c = lane; c < class_num + 1; c += 64) fp[c] += add;
    }
    if (lane == 0) {
        const float x0 = roi[m * 4 + 0], y0 = roi[m * 4 + 1], x1 = roi[m * 4 + 2], y1 = roi[m * 4 + 3];
        float* b = bbox + (long)m * 4;
        if (nreg == 4) {
            // denet_detect.py:87-100
            const float scx = 0.5f * (x0 + x1), scy = 0.5f * (y0 + y1), sw = x1 - x0, sh = y1 - y0;
            const float* r = z + s0;
            const float pcx = r[0] * sw + scx, pcy = r[1] * sh + scy;
            const float pw = expf(r[2]) * sw, ph = expf(r[3]) * sh;
            b[0] = pcx - pw * 0.5f;
            b[1] = pcy - ph * 0.5f;
            b[2] = pcx + pw * 0.5f;
            b[3] = pcy + ph * 0.5f;
        } else {
            b[0] = x0; b[1] = y0; b[2] = x1; b[3] = y1;
        }
    }
}

__device__ __forceinline__ float box_iou(float ax0, float ay0, float ax1, float ay1, float bx0, float by0, float bx1,
                                         float by1) {
    // denet_detect.cc:13-31, operation for operation in fp32
    const float dx = fmaxf(0.0f, fminf(ax1, bx1) - fmaxf(ax0, bx0));
    const float dy = fmaxf(0.0f, fminf(ay1, by1) - fmaxf(ay0, by0));
    const float ai = dx * dy;
    const float aa = (ax1 - ax0) * (ay1 - ay0);
    const float ab = (bx1 - bx0) * (by1 - by0);
    const float au = aa + ab - ai;
    return ai / au;
}

// one workgroup per (class, image): keep[b, cls, i] = 1 iff RoI i is a surviving detection of class cls
__global__ __launch_bounds__(256) void detect_nms_kernel(const float* __restrict__ det_pr,
                                                         const float* __restrict__ fitness,
                                                         const float* __restrict__ bbox, const int* __restrict__ count,
                                                         unsigned char* __restrict__ keep, int S, int C1,
                                                         float log_thr, float nms_thr, int do_nms) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* s_score = sm;              // [S]
    float* s_box = sm + S;            // [S][4]
    int* s_idx = (int*)(sm + 5 * S);  // [S]
    __shared__ int s_w[4];
    const int cls = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = min(count[b], S);
    unsigned char* kp = keep + ((long)b * (C1 - 1) + cls) * S;
    for (int i = tid; i < S; i += 256) kp[i] = 0;
    // candidates in RoI order (ballot compaction keeps the reference's j,i scan order)
    int base = 0;
    for (int chunk = 0; chunk < nb; chunk += 256) {
        const int i = chunk + tid;
        const long m = (long)b * S + i;
        const bool pred = (i < nb) && (det_pr[m * C1 + cls] >= log_thr);
        const unsigned long long mask = __ballot(pred);
        if (lane == 0) s_w[wave] = __popcll(mask);
        __syncthreads();
        int wb = 0, tot = 0;
        for (int w = 0; w < 4; ++w) {
            if (w < wave) wb += s_w[w];
            tot += s_w[w];
        }
        if (pred) {
            const int k = base + wb + __popcll(mask & ((1ull << lane) - 1ull));
            s_score[k] = fitness[m * C1 + cls];
            s_idx[k] = i;
            s_box[k * 4 + 0] = bbox[m * 4 + 0];
            s_box[k * 4 + 1] = bbox[m * 4 + 1];
            s_box[k * 4 + 2] = bbox[m * 4 + 2];
            s_box[k * 4 + 3] = bbox[m * 4 + 3];
        }
        base += tot;
        __syncthreads();
    }
    const int n = base;
    for (int a = tid; a < n; a += 256) {
        bool unique = true;
        if (do_nms) {
            const float sa = s_score[a];
            const float ax0 = s_box[a * 4], ay0 = s_box[a * 4 + 1], ax1 = s_box[a * 4 + 2], ay1 = s_box[a * 4 + 3];
            for (int o = 0; o < n; ++o) {
                if (sa < s_score[o] &&
                    box_iou(ax0, ay0, ax1, ay1, s_box[o * 4], s_box[o * 4 + 1], s_box[o * 4 + 2], s_box[o * 4 + 3]) > nms_thr) {
                    unique = false;
                    break;
                }
            }
        }
        if (unique) kp[s_idx[a]] = 1;
    }
}


// Gaussian soft-NMS (denet_detect.cc:35-71) of one (class, image) per WAVE. The method is sequential in its selections (every
// selected instance rescales the scores the next selection is made from), but the 80 x 32 (class, image) pairs of a batch are
// independent, and inside a pair both steps of an iteration are parallel over the candidates: the arg-max over the live
// scores (ties: the earliest list position - the reference's strict `>` scan over a std::list that keeps the scan order) and
// the rescoring s -= iou^2 / threshold, discard below -6.9. The candidates of the pair sit in LDS ([S] x score, box, RoI);
// a lane owns positions lane, lane + 64, ...; no barriers (one wave), the arg-max is a butterfly over (score, position).
// Output, dense per pair: sel_roi / sel_score [pair][S] in selection order, sel_n[pair]; pair = b * class_num + cls.
__global__ __launch_bounds__(64) void soft_nms_pair_kernel(const float* __restrict__ det_pr, const float* __restrict__ fitness,
                                                           const float* __restrict__ bbox, const int* __restrict__ count,
                                                           int* __restrict__ sel_roi, float* __restrict__ sel_score,
                                                           int* __restrict__ sel_n, int S, int C1, float log_thr, float nms_thr,
                                                           int do_nms) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* s_score = sm;              // [S]; a dead candidate holds -inf ... (its `alive` bit is what counts)
    float* s_box = sm + S;            // [S][4]
    int* s_idx = (int*)(sm + 5 * S);  // [S]
    unsigned char* s_alive = (unsigned char*)(sm + 6 * S);   // [S]
    const int cls = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int pair = b * (C1 - 1) + cls;
    const int nb = min(count[b], S);
    // candidates in RoI order (the reference's j, i scan)
    int n = 0;
    for (int chunk = 0; chunk < nb; chunk += 64) {
        const int i = chunk + lane;
        const long m = (long)b * S + i;
        const bool pred = (i < nb) && (det_pr[m * C1 + cls] >= log_thr);
        const unsigned long long mask = __ballot(pred);
        if (pred) {
            const int k = n + __popcll(mask & ((1ull << lane) - 1ull));
            s_score[k] = fitness[m * C1 + cls];
            s_idx[k] = i;
            s_box[k * 4 + 0] = bbox[m * 4 + 0];
            s_box[k * 4 + 1] = bbox[m * 4 + 1];
            s_box[k * 4 + 2] = bbox[m * 4 + 2];
            s_box[k * 4 + 3] = bbox[m * 4 + 3];
            s_alive[k] = 1;
        }
        n += __popcll(mask);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    int* o_roi = sel_roi + (long)pair * S;
    float* o_sc = sel_score + (long)pair * S;
    if (!do_nms) {                    // denet_detect.cc:76: the threshold outside (0, 1) keeps every candidate, in list order
        for (int k = lane; k < n; k += 64) {
            o_roi[k] = s_idx[k];
            o_sc[k] = s_score[k];
        }
        if (lane == 0) sel_n[pair] = n;
        return;
    }
    const float discard = -6.9f;
    int kept = 0;
    while (true) {
        // the reference's scan (denet_detect.cc:44-52): best = the list's head, a later element replaces it only if strictly
        // greater. Without NaNs that is the largest score, the earliest position among equals. A NaN never wins a `>` and,
        // as the running best, is never beaten: the scan returns the HEAD if the head is NaN, else the largest non-NaN score.
        // The lanes own interleaved positions, so the butterfly runs over the non-NaN candidates only and carries the
        // earliest live position beside it; a NaN head is then selected explicitly ([3, NaN, 5] -> 5, [NaN, 3] -> NaN).
        float best = 0.f;
        int pos = 0x7fffffff, first = 0x7fffffff;
        for (int k = lane; k < n; k += 64) {
            if (!s_alive[k]) continue;
            if (first == 0x7fffffff) first = k;
            const float v = s_score[k];
            if (v != v) continue;
            if (pos == 0x7fffffff || v > best) {
                best = v;
                pos = k;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ob = __shfl_xor(best, off, 64);
            const int op = __shfl_xor(pos, off, 64);
            const int of = __shfl_xor(first, off, 64);
            first = min(first, of);
            // the reference's scan keeps the EARLIER element unless a later one is strictly greater
            const bool take = (op != 0x7fffffff) && (pos == 0x7fffffff || (op < pos ? !(best > ob) : (ob > best)));
            if (take) {
                best = ob;
                pos = op;
            }
        }
        if (first != 0x7fffffff) {
            const float hv = s_score[first];
            if (hv != hv || pos == 0x7fffffff) {      // the head is NaN (or every live score is): the head is the selection
                best = hv;
                pos = first;
            }
        }
        if (pos == 0x7fffffff) break;
        const float mx0 = s_box[pos * 4], my0 = s_box[pos * 4 + 1], mx1 = s_box[pos * 4 + 2], my1 = s_box[pos * 4 + 3];
        if (lane == 0) {
            o_roi[kept] = s_idx[pos];
            o_sc[kept] = best;
            s_alive[pos] = 0;
        }
        ++kept;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        for (int k = lane; k < n; k += 64) {
            if (!s_alive[k]) continue;
            const float iou = box_iou(mx0, my0, mx1, my1, s_box[k * 4], s_box[k * 4 + 1], s_box[k * 4 + 2], s_box[k * 4 + 3]);
            const float v = s_score[k] - iou * iou / nms_thr;
            s_score[k] = v;
            if (v < discard) s_alive[k] = 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) sel_n[pair] = kept;
}

// offsets of the pairs' selections in the flat output (image, class, selection order), the per-image totals and the total
__global__ __launch_bounds__(1024) void soft_nms_scan_kernel(const int* __restrict__ sel_n, int* __restrict__ pair_off,
                                                             int* __restrict__ per_image, int* __restrict__ total, int B, int CN) {
    __shared__ int s_part[1024];
    const int tid = threadIdx.x, P = B * CN;
    int carry = 0;
    for (int base = 0; base < P; base += 1024) {
        const int v = base + tid < P ? sel_n[base + tid] : 0;
        s_part[tid] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int a = tid >= off ? s_part[tid - off] : 0;
            __syncthreads();
            s_part[tid] += a;
            __syncthreads();
        }
        if (base + tid < P) pair_off[base + tid] = carry + s_part[tid] - v;
        carry += s_part[1023];
        __syncthreads();
    }
    if (tid == 0) *total = carry;
    for (int b = tid; b < B; b += 1024) {
        int a = 0;
        for (int c = 0; c < CN; ++c) a += sel_n[b * CN + c];
        per_image[b] = a;
    }
}

__global__ __launch_bounds__(64) void soft_nms_gather_kernel(const int* __restrict__ sel_roi, const float* __restrict__ sel_score,
                                                             const int* __restrict__ sel_n, const int* __restrict__ pair_off,
                                                             float* __restrict__ out_score, int* __restrict__ out_cls,
                                                             int* __restrict__ out_row, int S, int CN) {
    const int cls = blockIdx.x, b = blockIdx.y, pair = b * CN + cls;
    const int n = sel_n[pair], o = pair_off[pair];
    for (int k = threadIdx.x; k < n; k += 64) {
        out_score[o + k] = sel_score[(long)pair * S + k];
        out_cls[o + k] = cls;
        out_row[o + k] = b * S + sel_roi[(long)pair * S + k];
    }
}

}  // namespace

extern "C" int denet_detect_decode(const float* logits, const float* roi_bbox, float* det_pr, float* fitness,
                                   float* bbox, int M, int CP, int class_num, int jointfit, int nreg, int nfit,
                                   float overlap_threshold, hipStream_t stream) {
    DENET_CHECK_ARG(logits && roi_bbox && det_pr && fitness && bbox, "detect_decode: null pointer");
    DENET_CHECK_ARG(nreg == 0 || nreg == 4, "detect_decode: nreg must be 0 or 4");
    const int fit_num = 5;
    const int s0 = jointfit ? class_num * fit_num + 1 : class_num + 1;
    DENET_CHECK_ARG(nfit >= 0 && nfit <= 64 && !(nfit > 0 && jointfit), "detect_decode: bad independent-fitness size");
    DENET_CHECK_ARG(s0 + nreg + nfit <= CP, "detect_decode: CP too small");
    hipLaunchKernelGGL(detect_decode_kernel, dim3((M + 3) / 4), dim3(256), 0, stream, logits, roi_bbox, det_pr, fitness,
                       bbox, M, CP, class_num, fit_num, jointfit, nreg, overlap_threshold, nfit);
    DENET_CHECK_LAUNCH("detect_decode");
    return DENET_OK;
}

extern "C" int denet_detect_nms(const float* det_pr, const float* fitness, const float* bbox, const int* count,
                                unsigned char* keep, int B, int S, int class_num, float pr_threshold,
                                float nms_threshold, hipStream_t stream) {
    DENET_CHECK_ARG(det_pr && fitness && bbox && count && keep, "detect_nms: null pointer");
    DENET_CHECK_ARG(S > 0 && S <= 4096 && class_num > 0, "detect_nms: bad sizes");
    const int do_nms = (nms_threshold > 0.0f && nms_threshold < 1.0f) ? 1 : 0;   // denet_detect.cc:76
    const size_t lds = (size_t)S * 6 * sizeof(float);
    hipLaunchKernelGGL(detect_nms_kernel, dim3(class_num, B), dim3(256), lds, stream, det_pr, fitness, bbox, count, keep,
                       S, class_num + 1, logf(pr_threshold), nms_threshold, do_nms);
    DENET_CHECK_LAUNCH("detect_nms");
    return DENET_OK;
}

// Gaussian soft-NMS of ONE class on the host (denet_detect.cc:35-71): scores are log-probabilities, every selected
// instance lowers the others by iou^2/threshold, instances below -6.9 are discarded; output in selection order.
// score/box: n candidates in RoI order; out_order[k] = index into the candidate list, out_score[k] its final score.
extern "C" int denet_soft_nms_host(const float* score, const float* box, int n, float nms_threshold,
                                   int* out_order, float* out_score, int* out_n) {
    DENET_CHECK_ARG((n == 0) || (score && box), "soft_nms_host: null pointer");
    DENET_CHECK_ARG(out_order && out_score && out_n, "soft_nms_host: null output");
    std::vector<float> s(score, score + n);
    std::vector<int> alive(n);
    for (int i = 0; i < n; ++i) alive[i] = i;
    const float discard = -6.9f;
    int k = 0;
    while (!alive.empty()) {
        size_t mi = 0;
        for (size_t j = 0; j < alive.size(); ++j)
            if (s[alive[j]] > s[alive[mi]]) mi = j;
        const int M = alive[mi];
        out_order[k] = M;
        out_score[k] = s[M];
        ++k;
        alive.erase(alive.begin() + mi);
        const float* bm = box + (size_t)M * 4;
        std::vector<int> next;
        for (int j : alive) {
            const float* bj = box + (size_t)j * 4;
            const float dx = fmaxf(0.0f, fminf(bm[2], bj[2]) - fmaxf(bm[0], bj[0]));
            const float dy = fmaxf(0.0f, fminf(bm[3], bj[3]) - fmaxf(bm[1], bj[1]));
            const float ai = dx * dy;
            const float au = (bm[2] - bm[0]) * (bm[3] - bm[1]) + (bj[2] - bj[0]) * (bj[3] - bj[1]) - ai;
            const float iou = ai / au;
            s[j] -= iou * iou / nms_threshold;
            if (!(s[j] < discard)) next.push_back(j);
        }
        alive.swap(next);
    }
    *out_n = k;
    return DENET_OK;
}

// The whole soft-NMS tail of a batch in one host call (denet_detect.cc:99-173 with use_soft_nms): for every image and
// class, the RoIs with det_pr >= log(pr_threshold) go through denet_soft_nms_host (skipped when the threshold is outside
// (0, 1), :129) and come out in selection order, classes ascending.
//   det_pr / fitness: [B*S, C1] host copies; bbox: [B*S, 4]; counts[b]: valid RoIs of image b.
//   out_score[k] = final log-domain score (the caller exponentiates), out_cls[k], out_row[k] = b*S + RoI index; out_count[b] detections of image b.
//   capacity: entries available in the out_* arrays; returns the total number of detections, or a negative error code.
extern "C" long denet_soft_nms_batch_host(const float* det_pr, const float* fitness, const float* bbox, const int* counts,
                                          int B, int S, int class_num, float pr_threshold, float nms_threshold,
                                          float* out_score, int* out_cls, int* out_row, int* out_count, long capacity) {
    DENET_CHECK_ARG(det_pr && fitness && bbox && counts && out_score && out_cls && out_row && out_count,
                    "soft_nms_batch_host: null pointer");
    const int C1 = class_num + 1;
    const float log_thr = logf(pr_threshold);
    const bool do_nms = nms_threshold > 0.0f && nms_threshold < 1.0f;
    std::vector<int> cand, order;
    std::vector<float> sc, bx, fin;
    long total = 0;
    for (int b = 0; b < B; ++b) {
        int nb = 0;
        for (int cls = 0; cls < class_num; ++cls) {
            cand.clear();
            for (int i = 0; i < counts[b]; ++i)
                if (det_pr[((size_t)b * S + i) * C1 + cls] >= log_thr) cand.push_back(b * S + i);
            const int n = (int)cand.size();
            if (n == 0) continue;
            sc.resize(n);
            bx.resize((size_t)n * 4);
            for (int k = 0; k < n; ++k) {
                sc[k] = fitness[(size_t)cand[k] * C1 + cls];
                memcpy(&bx[(size_t)k * 4], bbox + (size_t)cand[k] * 4, 4 * sizeof(float));
            }
            order.resize(n);
            fin.resize(n);
            int kept = n;
            if (do_nms) {
                const int rc = denet_soft_nms_host(sc.data(), bx.data(), n, nms_threshold, order.data(), fin.data(), &kept);
                if (rc != DENET_OK) return rc;
            } else {
                for (int k = 0; k < n; ++k) { order[k] = k; fin[k] = sc[k]; }
            }
            if (total + kept > capacity) { denet_set_error("soft_nms_batch_host: output capacity %ld exceeded", capacity); return DENET_ERR_ARG; }
            for (int k = 0; k < kept; ++k) {
                out_score[total] = fin[k];
                out_cls[total] = cls;
                out_row[total] = cand[order[k]];
                ++total;
            }
            nb += kept;
        }
        out_count[b] = nb;
    }
    return total;
}

// The soft-NMS tail of a batch on the DEVICE (denet_detect.cc:99-173 with use_soft_nms): the same result as
// denet_soft_nms_batch_host, bit for bit - out_score[k] (final log-domain score), out_cls[k], out_row[k] = b*S + RoI index in
// the reference's output order (image, class ascending, selection order), out_count[b], *out_total. All pointers are device
// memory; out_* hold B*class_num*S entries at most (the caller reads *out_total first); workspace: denet_soft_nms_workspace_bytes.
extern "C" size_t denet_soft_nms_workspace_bytes(int B, int S, int class_num) {
    const size_t P = (size_t)B * class_num;
    return P * S * (sizeof(int) + sizeof(float)) + 2 * P * sizeof(int);
}

extern "C" int denet_soft_nms_batch(const float* det_pr, const float* fitness, const float* bbox, const int* count, int B, int S,
                                    int class_num, float pr_threshold, float nms_threshold, float* out_score, int* out_cls,
                                    int* out_row, int* out_count, int* out_total, void* workspace, size_t workspace_bytes,
                                    hipStream_t stream) {
    DENET_CHECK_ARG(det_pr && fitness && bbox && count && out_score && out_cls && out_row && out_count && out_total && workspace,
                    "soft_nms_batch: null pointer");
    DENET_CHECK_ARG(B > 0 && S > 0 && S <= 4096 && class_num > 0, "soft_nms_batch: bad sizes");
    DENET_CHECK_ARG(workspace_bytes >= denet_soft_nms_workspace_bytes(B, S, class_num), "soft_nms_batch: workspace too small");
    const size_t P = (size_t)B * class_num;
    int* sel_roi = (int*)workspace;
    float* sel_score = (float*)(sel_roi + P * S);
    int* sel_n = (int*)(sel_score + P * S);
    int* pair_off = sel_n + P;
    const int do_nms = (nms_threshold > 0.0f && nms_threshold < 1.0f) ? 1 : 0;   // denet_detect.cc:76
    const size_t lds = (size_t)S * 6 * sizeof(float) + (size_t)((S + 3) & ~3);
    static size_t lds_set = 0;
    if (lds > 65536 && lds > lds_set) {
        const hipError_t e = hipFuncSetAttribute((const void*)soft_nms_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            denet_set_error("soft_nms_batch: hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
            return -(int)e;
        }
        lds_set = lds;
    }
    hipLaunchKernelGGL(soft_nms_pair_kernel, dim3(class_num, B), dim3(64), lds, stream, det_pr, fitness, bbox, count, sel_roi,
                       sel_score, sel_n, S, class_num + 1, logf(pr_threshold), nms_threshold, do_nms);
    hipLaunchKernelGGL(soft_nms_scan_kernel, dim3(1), dim3(1024), 0, stream, sel_n, pair_off, out_count, out_total, B, class_num);
    hipLaunchKernelGGL(soft_nms_gather_kernel, dim3(class_num, B), dim3(64), 0, stream, sel_roi, sel_score, sel_n, pair_off,
                       out_score, out_cls, out_row, S, class_num);
    DENET_CHECK_LAUNCH("soft_nms_batch");
    return DENET_OK;
}
