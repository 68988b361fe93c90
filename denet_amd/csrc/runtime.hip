// Error reporting for the C-ABI (include/denet_hip.h). The reference's GpuOps report failures through
// PyErr_Format + %(fail)s (denet_sparse_op.py:137-142); here every entry point returns an int status and
// leaves a thread-local message behind.
#include "common.h"
#include <type_traits>
#include <stdlib.h>
#include <stdarg.h>
#include <math.h>
#include <string.h>
#include <vector>

static thread_local char g_err[512] = "";

void denet_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- the batch norm whose reductions the next producing pass finishes itself (bn_final.h) ------------------------------------------
#include "bn_final.h"
static thread_local BnFinalDev t_bnf = {};
static thread_local int t_bnf_groups = 0;       // counters the armed buffer holds
static thread_local int t_bnf_state = 0;        // 0 nothing armed, 1 armed, 2 taken by a pass

static int bnf_arm(int kind, long M, int C, float momentum, float eps, float* o0, float* o1, float* o2, float* o3, unsigned* counters,
                   int ncounters) {
    DENET_CHECK_ARG(M > 0 && C > 0 && o0 && o1 && counters && ncounters > 0, "bn_final_arm: bad arguments");
    t_bnf.counter = counters; t_bnf.kind = kind; t_bnf.C = C; t_bnf.M = M; t_bnf.eps = eps; t_bnf.momentum = momentum;
    t_bnf.o0 = o0; t_bnf.o1 = o1; t_bnf.o2 = o2; t_bnf.o3 = o3;
    t_bnf_groups = ncounters;
    t_bnf_state = 1;
    return DENET_OK;
}

extern "C" int denet_bn_final_arm_stats(long M, int C, float momentum, float eps, float* run_mean, float* run_stdinv, float* save_mean,
                                        float* save_invstd, unsigned* counters, int ncounters) {
    return bnf_arm(1, M, C, momentum, eps, save_mean, save_invstd, run_mean, run_stdinv, counters, ncounters);
}

extern "C" int denet_bn_final_arm_sums(long M, int C, float* dgamma, float* dbeta, float* coef, unsigned* counters, int ncounters) {
    DENET_CHECK_ARG(coef, "bn_final_arm_sums: null pointer");
    return bnf_arm(2, M, C, 0.f, 0.f, dgamma, dbeta, coef, nullptr, counters, ncounters);
}

extern "C" int denet_bn_final_disarm(void) {
    const int taken = t_bnf_state == 2 ? 1 : 0;
    t_bnf_state = 0;
    t_bnf = BnFinalDev{};
    return taken;
}

// which reductions a producing pass may finish itself: bit 0 the forward statistics, bit 1 the backward sums. OFF by default
// (DENET_BN_FINAL_FOLD, denet_bn_final_mode): measured on MI355X the last workgroup's serial tail (ticket round trip under 256
// simultaneous arrivals + one or two dependent round trips of row loads, with the launch's own store burst on the fabric) costs
// the producing kernels +5...23 us each, more than the 5 us launches it replaces - 1 025 against 1 050 img/s (EXPERIMENTS.md).
static int g_bnf_mode = -1;
static int bnf_mode() {
    if (g_bnf_mode < 0) {
        const char* e = getenv("DENET_BN_FINAL_FOLD");
        g_bnf_mode = e ? (atoi(e) & 3) : 0;
    }
    return g_bnf_mode;
}
extern "C" int denet_bn_final_mode(int bits) {
    const int old = bnf_mode();
    if (bits >= 0) g_bnf_mode = bits & 3;
    return old;
}

BnFinalDev denet_bn_final_take(int kind, int C, int groups) {
    const int env_on = bnf_mode();
    if (!((env_on >> (kind - 1)) & 1) || t_bnf_state != 1 || t_bnf.kind != kind || t_bnf.C != C || groups <= 0 || groups > t_bnf_groups) return BnFinalDev{};
    t_bnf_state = 2;
    return t_bnf;
}

extern "C" const char* denet_last_error(void) { return g_err; }

extern "C" int denet_abi_version(void) { return 1; }

// device properties the host side needs to size grids / report rooflines
extern "C" int denet_device_info(int device, int* cu_count, int* clock_khz, char* arch, int arch_len) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        denet_set_error("hipGetDeviceProperties: %s", hipGetErrorString(e));
        return -(int)e;
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (clock_khz) *clock_khz = prop.clockRate;
    if (arch && arch_len > 0) snprintf(arch, arch_len, "%s", prop.gcnArchName);
    return DENET_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Host helper for the training-time RoI list editing (denet/layer/denet_sparse.py:184-187): the reference
// trims an over-full RoI list with Python's `random.sample(list, n)`. To stay call-for-call on the stdlib
// generator without a 16k-iteration Python loop per step, this function advances a copy of CPython's MT19937
// state exactly like `random.sample(range(n), k)` does for the "pool" branch (n <= setsize):
//     for i in range(k): j = randbelow(n - i); result[i] = pool[j]; pool[j] = pool[n - i - 1]
//     randbelow(m): k = m.bit_length(); r = getrandbits(k); while r >= m: r = getrandbits(k)
//     getrandbits(k <= 32) = genrand_uint32() >> (32 - k)
// mt: the 624 state words, *pos: the index word of random.getstate()[1]. Pure host code.
// ---------------------------------------------------------------------------------------------------------
static inline uint32_t mt_next(uint32_t* mt, int* pos) {
    const int N = 624, M = 397;
    if (*pos >= N) {
        int kk;
        uint32_t y;
        for (kk = 0; kk < N - M; kk++) {
            y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
            mt[kk] = mt[kk + M] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
        }
        for (; kk < N - 1; kk++) {
            y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
            mt[kk] = mt[kk + (M - N)] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
        }
        y = (mt[N - 1] & 0x80000000U) | (mt[0] & 0x7fffffffU);
        mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
        *pos = 0;
    }
    uint32_t y = mt[(*pos)++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680U;
    y ^= (y << 15) & 0xefc60000U;
    y ^= (y >> 18);
    return y;
}

// where the 32-bit outputs come from: the live generator state, or a stretch of its outputs drawn ahead of time
// (denet_host_mt_prefetch) that is consumed through a cursor
struct MtLive {
    uint32_t* mt;
    int* pos;
    bool ok() const { return true; }
    uint32_t next() { return mt_next(mt, pos); }
};
struct MtStream {
    const uint32_t* out;
    long n, cursor;
    bool good;
    bool ok() const { return good; }
    uint32_t next() {
        if (cursor >= n) {       // ran dry (a long rejection run of random.sample): the caller redoes the batch on the live generator
            good = false;
            return 0u;
        }
        return out[cursor++];
    }
};

template <class Src>
static int py_random_sample_from(Src& src, int n, int k, int* pool_ws, int* out);

extern "C" int denet_host_py_random_sample(uint32_t* mt, int* pos, int n, int k, int* pool_ws, int* out) {
    DENET_CHECK_ARG(mt && pos && pool_ws && out, "py_random_sample: null pointer");
    MtLive src{mt, pos};
    return py_random_sample_from(src, n, k, pool_ws, out);
}

// the same on a prefetched stretch, written without a data-dependent branch: _randbelow's rejection loop ("draw again while the
// value is not below m") is taken or not with a probability of up to one half, and every wrong guess of the branch predictor costs
// more than the rest of the draw - the RoI hand-off spends most of its host time here while the device stands idle. One iteration
// per OUTPUT: the value is taken (pool entry out, last entry in, i + 1) or not (a dummy slot is written, i stays) by selects.
static int py_random_sample_stream(MtStream& src, int n, int k, int* pool_ws, int* out);

template <class Src>
static int py_random_sample_from(Src& src, int n, int k, int* pool_ws, int* out) {
    DENET_CHECK_ARG(n > 0 && k >= 0 && k <= n, "py_random_sample: need 0 <= k <= n");
    if constexpr (std::is_same<Src, MtStream>::value) {
        if (src.ok()) return py_random_sample_stream(src, n, k, pool_ws, out);
    }
    // setsize of random.sample (CPython Lib/random.py): 21, plus 4**ceil(log(3k, 4)) when k > 5
    long setsize = 21;
    if (k > 5) {
        long p = 1;
        while (p < 3L * k) p *= 4;
        setsize += p;
    }
    DENET_CHECK_ARG(n <= setsize, "py_random_sample: the set-based branch of random.sample is not provided (n=%d k=%d)",
                    n, k);
    for (int i = 0; i < n; ++i) pool_ws[i] = i;
    for (int i = 0; i < k; ++i) {
        const uint32_t m = (uint32_t)(n - i);
        const int bits = 32 - __builtin_clz(m);          // m.bit_length() (m >= 1)
        uint32_t r = src.next() >> (32 - bits);
        while (r >= m && src.ok()) r = src.next() >> (32 - bits);
        if (!src.ok()) return DENET_OK;          // the caller checks the source
        out[i] = pool_ws[r];
        pool_ws[r] = pool_ws[n - i - 1];
    }
    return DENET_OK;
}

static int py_random_sample_stream(MtStream& src, int n, int k, int* pool_ws, int* out) {
    long setsize = 21;
    if (k > 5) {
        long p = 1;
        while (p < 3L * k) p *= 4;
        setsize += p;
    }
    DENET_CHECK_ARG(n <= setsize, "py_random_sample: the set-based branch of random.sample is not provided (n=%d k=%d)", n, k);
    for (int i = 0; i < n; ++i) pool_ws[i] = i;
    const uint32_t* const o = src.out;
    long cur = src.cursor;
    const long end = src.n;
    int dummy = 0;
    int i = 0;
    while (i < k && cur < end) {
        // m = n - i keeps its bit length while it stays >= 2^(bits - 1): the shift is constant over the segment
        const int bits = 32 - __builtin_clz((uint32_t)(n - i));
        const int sh = 32 - bits;
        const long seg = (long)n - (1L << (bits - 1)) + 1;
        const int seg_end = seg < k ? (int)seg : k;
        while (i < seg_end && cur < end) {
            const uint32_t r = o[cur++] >> sh;
            const uint32_t m = (uint32_t)(n - i);
            const bool take = r < m;
            int* const slot = take ? pool_ws + r : &dummy;
            out[i] = *slot;                     // (not taken: overwritten by the draw that is)
            *slot = pool_ws[m - 1];
            i += take ? 1 : 0;
        }
    }
    src.cursor = cur;
    if (i < k) src.good = false;          // the stretch ran dry inside the selection: the caller redoes the batch on the live generator
    return DENET_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Training-time RoI list editing of a whole batch (denet/layer/denet_sparse.py:184-201), call for call on a copy
// of the stdlib generator's MT19937 state. Per image, in order:
//   n_det > n_keep : keep the entries random.sample(list, n_keep) picks (pool branch, see above)
//   while len < S  : append (0.0, (x0, y0, x1, y1)), x0 = uniform(0,1), y0 = uniform(0,1), x1 = uniform(x0,1),
//                    y1 = uniform(y0,1); uniform(a,b) = a + (b-a)*random(); random() = genrand_res53
//   sample_gt      : list[-(k+1)] = (1.0, gt_k)
// det: [B,S,5] float (pr,x0,y0,x1,y1) rows of denet_samples_finish_host; gt: concatenated [n,4] doubles with
// gt_off[B+1]. Outputs: out_pr [B,S] and out_box [B,S,4] doubles (the values the reference's Python list would
// hold) and out_box_f32 [B,S,4] (what build_bbox_array would upload). Pure host code.
// ---------------------------------------------------------------------------------------------------------
template <class Src>
static inline double mt_random(Src& src) {
    const uint32_t a = src.next() >> 5, b = src.next() >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}

template <class Src>
static int edit_samples_from(Src& src, const float* det, const int* count, int B, int S, int n_keep, const double* gt,
                             const int* gt_off, int sample_gt, int* ws, double* out_pr, double* out_box, float* out_box_f32) {
#pragma clang fp contract(off)
    DENET_CHECK_ARG(det && count && ws && out_pr && out_box && out_box_f32, "edit_samples: null pointer");
    DENET_CHECK_ARG(B > 0 && S > 0 && n_keep >= 0 && n_keep <= S, "edit_samples: bad sizes");
    DENET_CHECK_ARG(!sample_gt || (gt_off && (gt || gt_off[B] == 0)), "edit_samples: ground truth missing");
    int* pool = ws;          // [S]
    int* pick = ws + S;      // [S]
    for (int b = 0; b < B; ++b) {
        const float* d = det + (size_t)b * S * 5;
        double* pr = out_pr + (size_t)b * S;
        double* bx = out_box + (size_t)b * S * 4;
        int n = count[b];
        DENET_CHECK_ARG(n >= 0 && n <= S, "edit_samples: count[%d] = %d out of range", b, n);
        if (n > n_keep) {
            int rc = py_random_sample_from(src, n, n_keep, pool, pick);
            if (rc != DENET_OK) return rc;
            if (!src.ok()) return DENET_OK;
            n = n_keep;
            for (int i = 0; i < n; ++i) {
                const float* r = d + (size_t)pick[i] * 5;
                pr[i] = r[0];
                for (int c = 0; c < 4; ++c) bx[i * 4 + c] = r[1 + c];
            }
        } else {
            for (int i = 0; i < n; ++i) {
                pr[i] = d[i * 5];
                for (int c = 0; c < 4; ++c) bx[i * 4 + c] = d[i * 5 + 1 + c];
            }
        }
        for (int i = n; i < S; ++i) {
            const double x0 = 0.0 + (1.0 - 0.0) * mt_random(src);
            const double y0 = 0.0 + (1.0 - 0.0) * mt_random(src);
            const double x1 = x0 + (1.0 - x0) * mt_random(src);
            const double y1 = y0 + (1.0 - y0) * mt_random(src);
            pr[i] = 0.0;
            bx[i * 4 + 0] = x0; bx[i * 4 + 1] = y0; bx[i * 4 + 2] = x1; bx[i * 4 + 3] = y1;
        }
        if (sample_gt) {
            const int g0 = gt_off[b], ng = gt_off[b + 1] - g0;
            DENET_CHECK_ARG(ng >= 0 && ng <= S, "edit_samples: image %d has %d ground-truth boxes (> %d RoIs)", b, ng, S);
            for (int k = 0; k < ng; ++k) {
                const int i = S - 1 - k;
                pr[i] = 1.0;
                for (int c = 0; c < 4; ++c) bx[i * 4 + c] = gt[(size_t)(g0 + k) * 4 + c];
            }
        }
        float* f = out_box_f32 + (size_t)b * S * 4;
        for (int i = 0; i < S * 4; ++i) f[i] = (float)bx[i];
        if (!src.ok()) return DENET_OK;
    }
    return DENET_OK;
}

extern "C" int denet_host_edit_samples(uint32_t* mt, int* pos, const float* det, const int* count, int B, int S,
                                       int n_keep, const double* gt, const int* gt_off, int sample_gt, int* ws,
                                       double* out_pr, double* out_box, float* out_box_f32) {
    DENET_CHECK_ARG(mt && pos, "edit_samples: null generator state");
    MtLive src{mt, pos};
    return edit_samples_from(src, det, count, B, S, n_keep, gt, gt_off, sample_gt, ws, out_pr, out_box, out_box_f32);
}

// The generator's outputs drawn AHEAD of the hand-off (the RoI list editing waits for the device's proposal, but the numbers it
// will draw do not): advances a COPY of the state (mt_host 624 words, *pos_host) by n outputs into out_host[n] and records the
// state words after every refill: snaps_host[j][624], snap_first_host[j] = index in out_host of the first output drawn from
// snapshot j (snapshot 0 = the state at entry, snap_first 0 = 0, its position word is the entry position; later snapshots
// start at position 0). n_snaps_host receives the count (<= max_snaps, else an argument error).
extern "C" int denet_host_mt_prefetch(uint32_t* mt, int* pos, long n, uint32_t* out, uint32_t* snaps, long* snap_first,
                                      int max_snaps, int* n_snaps) {
    DENET_CHECK_ARG(mt && pos && out && snaps && snap_first && n_snaps && n >= 0 && max_snaps >= 1, "mt_prefetch: bad arguments");
    memcpy(snaps, mt, 624 * sizeof(uint32_t));
    snap_first[0] = 0;
    int ns = 1;
    for (long i = 0; i < n; ++i) {
        const bool refill = *pos >= 624;
        out[i] = mt_next(mt, pos);
        if (refill) {
            DENET_CHECK_ARG(ns < max_snaps, "mt_prefetch: more than %d refills", max_snaps);
            memcpy(snaps + (size_t)ns * 624, mt, 624 * sizeof(uint32_t));
            snap_first[ns] = i;
            ++ns;
        }
    }
    *n_snaps = ns;
    return DENET_OK;
}

// denet_host_edit_samples drawing from such a stretch: stream_host[n_stream], *cursor_host = outputs consumed (in: where to
// start, out: where it stopped). *exhausted_host = 1 when the stretch ran dry: the outputs are then incomplete and the caller
// repeats the batch with denet_host_edit_samples on the live generator (which has not been touched).
extern "C" int denet_host_edit_samples_stream(const uint32_t* stream, long n_stream, long* cursor, int* exhausted, const float* det,
                                              const int* count, int B, int S, int n_keep, const double* gt, const int* gt_off,
                                              int sample_gt, int* ws, double* out_pr, double* out_box, float* out_box_f32) {
    DENET_CHECK_ARG(stream && cursor && exhausted && *cursor >= 0 && *cursor <= n_stream, "edit_samples_stream: bad stream arguments");
    MtStream src{stream, n_stream, *cursor, true};
    const int rc = edit_samples_from(src, det, count, B, S, n_keep, gt, gt_off, sample_gt, ws, out_pr, out_box, out_box_f32);
    *cursor = src.cursor;
    *exhausted = src.good ? 0 : 1;
    return rc;
}

// The host's whole share of the RoI hand-off in ONE call, for the moment the device stands idle between its proposal and the
// gather: denet_samples_finish_host (sample tuples from the packed proposal: det_out [B][S][5]) followed by
// denet_host_edit_samples_stream on them. Same outputs as the two calls.
extern "C" int denet_samples_finish_host(const int* box_host, const float* absd_host, const int* count_host, int B,
                                         int sample_count, int H, int W, float* samples_host);
extern "C" int denet_host_handoff_stream(const uint32_t* stream, long n_stream, long* cursor, int* exhausted, const int* box_host,
                                         const float* absd_host, const int* count_host, int H, int W, int B, int S, int n_keep,
                                         const double* gt, const int* gt_off, int sample_gt, int* ws, float* det_out, double* out_pr,
                                         double* out_box, float* out_box_f32) {
    DENET_CHECK_ARG(det_out, "handoff_stream: null pointer");
    int rc = denet_samples_finish_host(box_host, absd_host, count_host, B, S, H, W, det_out);
    if (rc != DENET_OK) return rc;
    return denet_host_edit_samples_stream(stream, n_stream, cursor, exhausted, det_out, count_host, B, S, n_keep, gt, gt_off, sample_gt,
                                          ws, out_pr, out_box, out_box_f32);
}

// ... and the part of it the device is actually waiting for: the bbox array alone (out_box_f32 [B][S][4]), straight from the
// packed proposal's integer boxes - the same selection (random.sample on the same outputs), the same random boxes, the same
// float32 values as denet_host_handoff_stream writes, without the score arithmetic (expf) and the double-precision lists, which
// the caller produces later with that call from the same cursor (nothing has been consumed for good: *cursor is its own copy).
// uniforms_host (may be null): uniforms_host[p] = the double random.random() returns when it starts at output p of the stretch
// (denet_host_mt_uniforms, computed ahead like the stretch itself): the random boxes then cost four table reads each
extern "C" int denet_host_handoff_boxes_stream_u(const uint32_t* stream, long n_stream, long* cursor, int* exhausted,
                                                 const int* box_host, const int* count_host, int H, int W, int B, int S, int n_keep,
                                                 const double* gt, const int* gt_off, int sample_gt, int* ws, float* out_box_f32,
                                                 const double* uniforms_host) {
#pragma clang fp contract(off)
    DENET_CHECK_ARG(stream && cursor && exhausted && *cursor >= 0 && *cursor <= n_stream, "handoff_boxes_stream: bad stream arguments");
    DENET_CHECK_ARG(box_host && count_host && ws && out_box_f32, "handoff_boxes_stream: null pointer");
    DENET_CHECK_ARG(B > 0 && S > 0 && n_keep >= 0 && n_keep <= S && H > 0 && W > 0, "handoff_boxes_stream: bad sizes");
    DENET_CHECK_ARG(!sample_gt || (gt_off && (gt || gt_off[B] == 0)), "handoff_boxes_stream: ground truth missing");
    MtStream src{stream, n_stream, *cursor, true};
    int* pool = ws;          // [S]
    int* pick = ws + S;      // [S]
    // (float)((double)v / W) for every cell coordinate 0 .. W (and H): 2 x 70 000 divisions per batch become table reads
    float tw[258], th[258];
    DENET_CHECK_ARG(H <= 256 && W <= 256, "handoff_boxes_stream: map %dx%d unsupported", H, W);
    for (int v = 0; v <= W + 1; ++v) tw[v] = (float)((double)v / W);
    for (int v = 0; v <= H + 1; ++v) th[v] = (float)((double)v / H);
    // box_host is what the device has just written (a pinned buffer: its lines are in nobody's cache) and random.sample's picks
    // walk an image's rows in random order - a DRAM latency per row. The rows of an image are copied front to back first (the
    // prefetcher's pattern), the picks then read the copy
    static thread_local std::vector<int> rows;
    rows.resize((size_t)S * 4);
    for (int b = 0; b < B && src.ok(); ++b) {
        float* f = out_box_f32 + (size_t)b * S * 4;
        int n = count_host[b];
        DENET_CHECK_ARG(n >= 0 && n <= S, "handoff_boxes_stream: count[%d] = %d out of range", b, n);
        memcpy(rows.data(), box_host + (size_t)b * S * 4, (size_t)n * 4 * sizeof(int));
        const int* bx = rows.data();
        const bool trim = n > n_keep;
        if (trim) {
            int rc = py_random_sample_from(src, n, n_keep, pool, pick);
            if (rc != DENET_OK) return rc;
            if (!src.ok()) break;
            n = n_keep;
        }
        unsigned bad = 0;
        for (int i = 0; i < n; ++i) {
            const int* r = bx + (size_t)(trim ? pick[i] : i) * 4;
            bad |= (unsigned)((unsigned)r[0] > (unsigned)W) | (unsigned)((unsigned)r[2] > (unsigned)W) |
                   (unsigned)((unsigned)r[1] > (unsigned)H) | (unsigned)((unsigned)r[3] > (unsigned)H);
            // (an index beyond the tables is clamped here and reported behind the loop)
            f[i * 4 + 0] = tw[(unsigned)r[0] > 256u ? 256 : r[0]];
            f[i * 4 + 1] = th[(unsigned)r[1] > 256u ? 256 : r[1]];
            f[i * 4 + 2] = tw[((unsigned)r[2] > 256u ? 256 : r[2]) + 1];
            f[i * 4 + 3] = th[((unsigned)r[3] > 256u ? 256 : r[3]) + 1];
        }
        DENET_CHECK_ARG(!bad, "handoff_boxes_stream: box outside the map");
        const long need = 8L * (S - n);
        if (src.cursor + need <= src.n) {
            // the whole image's random boxes lie inside the stretch: no per-output bounds test (and table reads with `uniforms_host`)
            const long c0 = src.cursor;
            if (uniforms_host) {
                const double* u = uniforms_host + c0;
                for (int i = n; i < S; ++i, u += 8) {
                    const double x0 = 0.0 + (1.0 - 0.0) * u[0];
                    const double y0 = 0.0 + (1.0 - 0.0) * u[2];
                    const double x1 = x0 + (1.0 - x0) * u[4];
                    const double y1 = y0 + (1.0 - y0) * u[6];
                    f[i * 4 + 0] = (float)x0; f[i * 4 + 1] = (float)y0; f[i * 4 + 2] = (float)x1; f[i * 4 + 3] = (float)y1;
                }
            } else {
                const uint32_t* o = stream + c0;
                for (int i = n; i < S; ++i, o += 8) {
                    double u[4];
                    for (int q = 0; q < 4; ++q) u[q] = ((o[2 * q] >> 5) * 67108864.0 + (o[2 * q + 1] >> 6)) * (1.0 / 9007199254740992.0);
                    const double x0 = 0.0 + (1.0 - 0.0) * u[0];
                    const double y0 = 0.0 + (1.0 - 0.0) * u[1];
                    const double x1 = x0 + (1.0 - x0) * u[2];
                    const double y1 = y0 + (1.0 - y0) * u[3];
                    f[i * 4 + 0] = (float)x0; f[i * 4 + 1] = (float)y0; f[i * 4 + 2] = (float)x1; f[i * 4 + 3] = (float)y1;
                }
            }
            src.cursor = c0 + need;
        } else {
            for (int i = n; i < S; ++i) {
                const double x0 = 0.0 + (1.0 - 0.0) * mt_random(src);
                const double y0 = 0.0 + (1.0 - 0.0) * mt_random(src);
                const double x1 = x0 + (1.0 - x0) * mt_random(src);
                const double y1 = y0 + (1.0 - y0) * mt_random(src);
                f[i * 4 + 0] = (float)x0; f[i * 4 + 1] = (float)y0; f[i * 4 + 2] = (float)x1; f[i * 4 + 3] = (float)y1;
            }
        }
        if (sample_gt) {
            const int g0 = gt_off[b], ng = gt_off[b + 1] - g0;
            DENET_CHECK_ARG(ng >= 0 && ng <= S, "handoff_boxes_stream: image %d has %d ground-truth boxes (> %d RoIs)", b, ng, S);
            for (int k = 0; k < ng; ++k)
                for (int c = 0; c < 4; ++c) f[(S - 1 - k) * 4 + c] = (float)gt[(size_t)(g0 + k) * 4 + c];
        }
    }
    *cursor = src.cursor;
    *exhausted = src.good ? 0 : 1;
    return DENET_OK;
}

extern "C" int denet_host_handoff_boxes_stream(const uint32_t* stream, long n_stream, long* cursor, int* exhausted,
                                               const int* box_host, const int* count_host, int H, int W, int B, int S, int n_keep,
                                               const double* gt, const int* gt_off, int sample_gt, int* ws, float* out_box_f32) {
    return denet_host_handoff_boxes_stream_u(stream, n_stream, cursor, exhausted, box_host, count_host, H, W, B, S, n_keep, gt, gt_off,
                                             sample_gt, ws, out_box_f32, nullptr);
}

// uniforms_host[p] = genrand_res53 of the outputs p, p + 1 of a prefetched stretch (the value random.random() returns when the
// generator stands at output p), for every p < n - 1; uniforms_host[n - 1] = 0 (never read: a box needs 8 outputs)
extern "C" int denet_host_mt_uniforms(const uint32_t* stream, long n, double* uniforms_host) {
#pragma clang fp contract(off)
    DENET_CHECK_ARG(stream && uniforms_host && n >= 1, "mt_uniforms: bad arguments");
    for (long p = 0; p + 1 < n; ++p)
        uniforms_host[p] = ((stream[p] >> 5) * 67108864.0 + (stream[p + 1] >> 6)) * (1.0 / 9007199254740992.0);
    uniforms_host[n - 1] = 0.0;
    return DENET_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Detection targets of a batch (denet/layer/denet_detect.py:147-235), RoI-major: row m = b*S + index.
// IoU matrix in float32 with the operation order of the compiled Theano function (common/theano_util.py:38-59);
// every GT/RoI pair with IoU > t0 sets its class (or class x fitness-bin, :180-183, double arithmetic on the
// float32 IoU) and clears the null class; per RoI the arg-max GT gives the box-regression target if IoU > t1
// (:194-213, double arithmetic, stored as float32); rows are normalised to sum 1 and divided by S (:216-226).
// gt: concatenated [n,4] doubles, gt_off [B+1], gt_class [n]; roi [B,S,4] doubles (the edited RoI list).
// det [B*S,s0], valid [B*S] (or null), reg [B*S,8] (or null), indfit [B*S,fitness_num] (or null: no independent
// fitness head). Pure host code.
// ---------------------------------------------------------------------------------------------------------
extern "C" int denet_host_detect_targets(const double* gt, const int* gt_off, const int* gt_class, const double* roi,
                                         int B, int S, int s0, int null_class, int fitness_num, int jointfit,
                                         double t0, double t1, float* det, float* valid, float* reg, float* indfit) {
#pragma clang fp contract(off)
    DENET_CHECK_ARG(gt_off && roi && det && B > 0 && S > 0 && s0 > 0, "detect_targets: bad arguments");
    DENET_CHECK_ARG(null_class >= 0 && null_class < s0, "detect_targets: null class out of range");
    DENET_CHECK_ARG((valid == nullptr) == (reg == nullptr), "detect_targets: valid and reg go together");
    const float t0f = (float)t0, t1f = (float)t1, Sf = (float)S;
    const float inv_s = 1.0f / Sf;
    std::vector<float> gx, garea, ov;
    for (int b = 0; b < B; ++b) {
        const int g0 = gt_off[b], ng = gt_off[b + 1] - g0;
        DENET_CHECK_ARG(ng >= 0 && (ng == 0 || (gt && gt_class)), "detect_targets: ground truth of image %d missing", b);
        gx.resize((size_t)ng * 4);
        garea.resize(ng);
        ov.resize(ng);
        for (int k = 0; k < ng; ++k) {
            for (int c = 0; c < 4; ++c) gx[k * 4 + c] = (float)gt[(size_t)(g0 + k) * 4 + c];
            garea[k] = (gx[k * 4 + 2] - gx[k * 4 + 0]) * (gx[k * 4 + 3] - gx[k * 4 + 1]);
        }
        for (int i = 0; i < S; ++i) {
            const size_t row = (size_t)b * S + i;
            float* d = det + row * s0;
            for (int c = 0; c < s0; ++c) d[c] = 0.f;
            d[null_class] = 1.f;
            float* r = reg ? reg + row * 8 : nullptr;
            if (r) {
                r[0] = r[1] = r[4] = r[5] = 0.f;
                r[2] = r[3] = r[6] = r[7] = 1.f;
                valid[row] = 0.f;
            }
            // independent fitness target (:188-192, :219-226): bin 0 = background, matched RoIs mark bins 1..n-1
            float* fi = indfit ? indfit + row * fitness_num : nullptr;
            if (fi) {
                for (int c = 0; c < fitness_num; ++c) fi[c] = 0.f;
                fi[0] = 1.f;
            }
            if (ng == 0) {
                d[null_class] = inv_s;      // (1 / 1) / S
                if (fi) fi[0] = inv_s;
                continue;
            }
            const double* yd = roi + row * 4;
            const float y0 = (float)yd[0], y1 = (float)yd[1], y2 = (float)yd[2], y3 = (float)yd[3];
            const float yarea = (y2 - y0) * (y3 - y1);
            int best = 0;
            bool best_nan = false;
            for (int k = 0; k < ng; ++k) {
                const float* x = &gx[k * 4];
                const float dx = fmaxf(fminf(x[2], y2) - fmaxf(x[0], y0), 0.f);
                const float dy = fmaxf(fminf(x[3], y3) - fmaxf(x[1], y1), 0.f);
                const float inter = dx * dy;
                const float uni = (garea[k] + yarea) - inter;
                const float v = inter / uni;
                ov[k] = v;
                // numpy.argmax: the first maximum, a NaN counts as the maximum
                if (!best_nan) {
                    if (v != v) { best = k; best_nan = true; }
                    else if (k > 0 && v > ov[best]) best = k;
                }
                if (v > t0f) {
                    int col = gt_class[g0 + k];
                    if (jointfit) {
                        const double sf = ((double)v - t0) / (1.0 - t0);
                        long f = (long)((double)fitness_num * sf);
                        f = f < 0 ? 0 : (f > fitness_num - 1 ? fitness_num - 1 : f);
                        col = col * fitness_num + (int)f;
                    }
                    DENET_CHECK_ARG(col >= 0 && col < s0, "detect_targets: class column %d out of range", col);
                    d[col] = 1.f;
                    d[null_class] = 0.f;
                    if (fi) {
                        const double sf = ((double)v - t0) / (1.0 - t0);
                        long f = 1 + (long)floor((double)(fitness_num - 1) * sf);
                        f = f < 1 ? 1 : (f > fitness_num - 1 ? fitness_num - 1 : f);
                        fi[0] = 0.f;
                        fi[f] = 1.f;
                    }
                }
            }
            if (fi) {
                float fs = 0.f;
                for (int c = 0; c < fitness_num; ++c) fs += fi[c];
                for (int c = 0; c < fitness_num; ++c)
                    if (fi[c] != 0.f) fi[c] = (fi[c] / fs) / Sf;
            }
            float sum = 0.f;
            for (int c = 0; c < s0; ++c) sum += d[c];
            for (int c = 0; c < s0; ++c)
                if (d[c] != 0.f) d[c] = (d[c] / sum) / Sf;
            if (r && ov[best] > t1f) {
                const double* t = gt + (size_t)(g0 + best) * 4;
                valid[row] = inv_s;
                r[0] = (float)(0.5 * (t[0] + t[2]));
                r[1] = (float)(0.5 * (t[1] + t[3]));
                r[2] = (float)(t[2] - t[0]);
                r[3] = (float)(t[3] - t[1]);
                r[4] = (float)(0.5 * (yd[0] + yd[2]));
                r[5] = (float)(0.5 * (yd[1] + yd[3]));
                r[6] = (float)(yd[2] - yd[0]);
                r[7] = (float)(yd[3] - yd[1]);
            }
        }
    }
    return DENET_OK;
}

// ---- stream placement probe -------------------------------------------------------------------------------------------
// One wavefront that does nothing for `cycles` shader clocks. Two of them on two streams finish in ~1x the time when the
// streams sit on different hardware queues and in ~2x when the runtime multiplexed both onto one queue (HIP gives a process
// GPU_MAX_HW_QUEUES = 4 queues; torch's stream pool and RCCL create dozens of streams). The host picks, for the
// filter-gradient chain, a stream that really runs beside the compute stream (ops.init_streams).
namespace {
__global__ void spin_kernel(long long cycles, int* sink) {
    const long long t0 = wall_clock64();
    long long t = t0;
    while (t - t0 < cycles) {
        __builtin_amdgcn_s_sleep(32);
        t = wall_clock64();
    }
    if (sink && t == 0) *sink = 1;
}
}  // namespace

// wall_clock64 ticks at 100 MHz on gfx950: `microseconds` x 100 ticks
extern "C" int denet_spin(int microseconds, hipStream_t stream) {
    DENET_CHECK_ARG(microseconds > 0 && microseconds <= 100000, "spin: duration out of range");
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, stream, (long long)microseconds * 100, (int*)nullptr);
    DENET_CHECK_LAUNCH("spin");
    return DENET_OK;
}
