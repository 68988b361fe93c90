// Sanitizer driver for the HOST-native functions of libdenet_hip (tests/test_sanitizers.py builds runtime.hip, samples.hip,
// detect.hip and image.hip with -Xarch_host -fsanitize=address,undefined and links this file against them).
// These functions do raw pointer arithmetic on caller buffers (MT19937 emulation, RoI list editing, detection targets, RoI
// clustering, soft-NMS, Pillow coefficient tables); every buffer here is heap-allocated at EXACTLY the size the C ABI
// (include/denet_hip.h) documents, so any out-of-bounds access or undefined operation aborts the run. Results are
// checked for the structural properties the reference guarantees (denet/layer/denet_sparse.py:184-201,
// denet/layer/denet_detect.py:147-235, denet/layer/denet_sparse.cc:165-242, denet/layer/denet_detect.cc:35-71).
// No GPU is touched. SURVEY.md section 5 (race / sanitizer row).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include <hip/hip_runtime_api.h>
#include "../../include/denet_hip.h"

#define CHECK(cond)                                                                     \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            fprintf(stderr, "host_sanitize: %s:%d: %s\n", __FILE__, __LINE__, #cond);   \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

// exact-size heap buffers (ASAN red zones right behind the last element)
template <typename T>
struct Buf {
    T* p;
    size_t n;
    explicit Buf(size_t n_) : p((T*)malloc(n_ * sizeof(T) + (n_ == 0))), n(n_) { memset(p, 0, n_ * sizeof(T)); }
    ~Buf() { free(p); }
    T& operator[](size_t i) { return p[i]; }
};

static void mt_seed(uint32_t* mt, int* pos, uint32_t s) {      // init_genrand of MT19937
    mt[0] = s;
    for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    *pos = 624;
}

static uint32_t lcg = 12345u;
static double urand() { lcg = lcg * 1664525u + 1013904223u; return (lcg >> 8) / 16777216.0; }

static void random_box(double* b) {
    double x0 = urand(), x1 = urand(), y0 = urand(), y1 = urand();
    if (x1 < x0) { double t = x0; x0 = x1; x1 = t; }
    if (y1 < y0) { double t = y0; y0 = y1; y1 = t; }
    b[0] = x0; b[1] = y0; b[2] = x1; b[3] = y1;
}

static void test_random_sample() {
    Buf<uint32_t> mt(624);
    Buf<int> pos(1);
    mt_seed(mt.p, pos.p, 5489u);
    // {n, k, served}: random.sample's pool branch (n <= 21 + 4**ceil(log4(3k))) is provided, the set branch is refused
    const int cases[][3] = {{1, 1, 1}, {5, 5, 1}, {21, 3, 1}, {22, 6, 1}, {577, 576, 1}, {576, 519, 1}, {4117, 576, 1}, {20, 0, 1},
                            {1000, 1, 0}, {7936, 576, 0}, {64, 0, 0}};
    for (auto& c : cases) {
        const int n = c[0], k = c[1];
        Buf<int> pool(n), out(k);
        const int rc = denet_host_py_random_sample(mt.p, pos.p, n, k, pool.p, out.p);
        CHECK((rc == 0) == (c[2] == 1));
        if (rc != 0) continue;
        std::vector<char> seen(n, 0);
        for (int i = 0; i < k; ++i) {
            CHECK(out[i] >= 0 && out[i] < n && !seen[out[i]]);       // k distinct indices of range(n)
            seen[out[i]] = 1;
        }
        CHECK(pos[0] >= 0 && pos[0] <= 624);
    }
    Buf<int> pool(4), out(4);
    CHECK(denet_host_py_random_sample(mt.p, pos.p, 3, 4, pool.p, out.p) != 0);     // k > n: an error, not an overrun
}

static void test_edit_samples() {
    Buf<uint32_t> mt(624);
    Buf<int> pos(1);
    mt_seed(mt.p, pos.p, 19650218u);
    struct Case { int B, S, n_keep, sample_gt, gt_max; };
    const Case cases[] = {{3, 16, 16, 1, 4}, {4, 16, 12, 1, 5}, {2, 576, 519, 1, 39}, {2, 9, 5, 0, 3}, {1, 4, 0, 1, 4}, {2, 25, 25, 1, 0}};
    for (const Case& c : cases) {
        const int B = c.B, S = c.S;
        Buf<float> det((size_t)B * S * 5);
        Buf<int> count(B), off(B + 1), ws(2 * S);
        std::vector<double> gtv;
        for (int b = 0; b < B; ++b) {
            // trim (count > n_keep), no trim, empty, full
            const int choices[4] = {S, c.n_keep > 1 ? c.n_keep - 1 : 0, 0, c.n_keep < S ? c.n_keep + 1 : S};
            count[b] = choices[b % 4];
            for (int i = 0; i < count[b]; ++i) {
                double bx[4];
                random_box(bx);
                float* r = det.p + ((size_t)b * S + i) * 5;
                r[0] = (float)(0.5 * urand());
                for (int k = 0; k < 4; ++k) r[1 + k] = (float)bx[k];
            }
            const int ng = c.gt_max == 0 ? 0 : (b == 0 ? c.gt_max : (int)(urand() * (c.gt_max + 1)));   // image 0: the maximum (39 boxes)
            off[b + 1] = off[b] + ng;
            for (int g = 0; g < ng; ++g) {
                double bx[4];
                random_box(bx);
                gtv.insert(gtv.end(), bx, bx + 4);
            }
        }
        Buf<double> gt(gtv.size());
        if (!gtv.empty()) memcpy(gt.p, gtv.data(), gtv.size() * sizeof(double));
        Buf<double> out_pr((size_t)B * S), out_box((size_t)B * S * 4);
        Buf<float> out_f32((size_t)B * S * 4);
        CHECK(denet_host_edit_samples(mt.p, pos.p, det.p, count.p, B, S, c.n_keep, gt.p, off.p, c.sample_gt, ws.p, out_pr.p,
                                      out_box.p, out_f32.p) == 0);
        for (int b = 0; b < B; ++b) {
            const int ng = c.sample_gt ? off[b + 1] - off[b] : 0;
            for (int k = 0; k < ng; ++k) {                        // ground truth k sits at position -(k+1) with score 1.0
                const size_t i = (size_t)b * S + (S - 1 - k);
                CHECK(out_pr[i] == 1.0);
                for (int q = 0; q < 4; ++q) CHECK(out_box[i * 4 + q] == gt[(size_t)(off[b] + k) * 4 + q]);
            }
            for (int i = 0; i < S; ++i) {
                const double* bx = out_box.p + ((size_t)b * S + i) * 4;
                CHECK(bx[0] >= 0.0 && bx[2] <= 1.0 && bx[0] <= bx[2] && bx[1] <= bx[3]);
                for (int q = 0; q < 4; ++q) CHECK(out_f32[((size_t)b * S + i) * 4 + q] == (float)bx[q]);
            }
        }
    }
    // more ground truth than RoIs: refused, nothing written out of bounds
    Buf<float> det(4 * 5);
    Buf<int> count(1), off(2), ws(8);
    off[1] = 5;
    Buf<double> gt(20), pr(4), bx(16);
    Buf<float> f32(16);
    CHECK(denet_host_edit_samples(mt.p, pos.p, det.p, count.p, 1, 4, 4, gt.p, off.p, 1, ws.p, pr.p, bx.p, f32.p) != 0);
    count[0] = 5;       // a count beyond S
    off[1] = 0;
    CHECK(denet_host_edit_samples(mt.p, pos.p, det.p, count.p, 1, 4, 4, gt.p, off.p, 1, ws.p, pr.p, bx.p, f32.p) != 0);
}

static void test_prefetch_stream() {
    // outputs drawn ahead + the editing that walks through them == the editing on the live generator, state included
    const int B = 3, S = 16, n_keep = 12;
    for (int round = 0; round < 2; ++round) {
        Buf<uint32_t> mt(624), mt2(624);
        Buf<int> pos(1), pos2(1);
        mt_seed(mt.p, pos.p, 777u + round);
        pos[0] = round ? 620 : 624;                    // a refill right away / a few outputs before one
        memcpy(mt2.p, mt.p, 624 * sizeof(uint32_t));
        pos2[0] = pos[0];
        const long n = round ? 40 : 8L * B * S + 64;   // round 1: far too short a stretch -> must report "dry", not overrun
        const int max_snaps = (int)(n / 624) + 3;
        Buf<uint32_t> out(n), snaps((size_t)max_snaps * 624);
        Buf<long> first(max_snaps);
        Buf<int> ns(1);
        CHECK(denet_host_mt_prefetch(mt2.p, pos2.p, n, out.p, snaps.p, first.p, max_snaps, ns.p) == 0);
        CHECK(ns[0] >= 1 && ns[0] <= max_snaps && first[0] == 0);
        Buf<float> det((size_t)B * S * 5);
        Buf<int> count(B), off(B + 1), ws(2 * S), dry(1);
        count[0] = S; count[1] = 0; count[2] = 5;       // trim, empty, short
        for (size_t i = 0; i < det.n; ++i) det[i] = (float)urand();
        off[1] = 2; off[2] = 2; off[3] = 5;
        Buf<double> gt(5 * 4), pr_a((size_t)B * S), bx_a((size_t)B * S * 4), pr_b((size_t)B * S), bx_b((size_t)B * S * 4);
        for (size_t i = 0; i < gt.n; ++i) gt[i] = urand();
        Buf<float> f_a((size_t)B * S * 4), f_b((size_t)B * S * 4);
        Buf<long> cursor(1);
        CHECK(denet_host_edit_samples_stream(out.p, n, cursor.p, dry.p, det.p, count.p, B, S, n_keep, gt.p, off.p, 1, ws.p, pr_a.p, bx_a.p, f_a.p) == 0);
        CHECK(cursor[0] >= 0 && cursor[0] <= n);
        if (round) { CHECK(dry[0] == 1); continue; }
        CHECK(dry[0] == 0);
        CHECK(denet_host_edit_samples(mt.p, pos.p, det.p, count.p, B, S, n_keep, gt.p, off.p, 1, ws.p, pr_b.p, bx_b.p, f_b.p) == 0);
        CHECK(memcmp(pr_a.p, pr_b.p, pr_a.n * sizeof(double)) == 0 && memcmp(bx_a.p, bx_b.p, bx_a.n * sizeof(double)) == 0);
        CHECK(memcmp(f_a.p, f_b.p, f_a.n * sizeof(float)) == 0);
        // the live state after the batch == the snapshot the cursor ended in
        int j = 0;
        while (j + 1 < ns[0] && first[j + 1] < cursor[0]) ++j;
        const long p = (j == 0 ? 624 : 0) + (cursor[0] - first[j]);
        CHECK(p == pos[0] && memcmp(snaps.p + (size_t)j * 624, mt.p, 624 * sizeof(uint32_t)) == 0);
        Buf<uint32_t> big(5000), two((size_t)2 * 624);
        Buf<long> first2(2);
        CHECK(denet_host_mt_prefetch(mt2.p, pos2.p, 5000, big.p, two.p, first2.p, 2, ns.p) != 0);     // too few snapshot slots: refused
    }
}

static void test_detect_targets() {
    struct Case { int B, S, ncls, jointfit, reg, indfit; };
    const Case cases[] = {{3, 16, 5, 0, 1, 0}, {2, 576, 80, 0, 1, 0}, {2, 25, 3, 1, 1, 0}, {2, 16, 4, 0, 0, 1}, {1, 9, 6, 1, 0, 0}};
    for (const Case& c : cases) {
        const int B = c.B, S = c.S, fnum = c.jointfit ? 5 : 6;
        const int null_class = c.jointfit ? c.ncls * fnum : c.ncls, s0 = null_class + 1;
        Buf<int> off(B + 1);
        std::vector<double> gtv;
        std::vector<int> clsv;
        Buf<double> roi((size_t)B * S * 4);
        for (int b = 0; b < B; ++b) {
            const int ng = b == 1 ? 0 : (S >= 576 ? 39 : 3);                 // image 1 has no ground truth
            off[b + 1] = off[b] + ng;
            for (int g = 0; g < ng; ++g) {
                double bx[4];
                random_box(bx);
                gtv.insert(gtv.end(), bx, bx + 4);
                clsv.push_back((int)(urand() * c.ncls) % c.ncls);
            }
            for (int i = 0; i < S; ++i) {
                double* r = roi.p + ((size_t)b * S + i) * 4;
                if (ng && i < 3 * ng) {              // graded copies of the ground truth: every fitness bin, exact matches
                    const double* g = &gtv[(size_t)(off[b] + i / 3) * 4];
                    const double sh = (i % 3) * 0.15;
                    r[0] = g[0] + sh * (g[2] - g[0]); r[1] = g[1]; r[2] = g[2]; r[3] = g[3] - 0.5 * sh * (g[3] - g[1]);
                } else if (i == S - 1) {
                    r[0] = r[1] = r[2] = r[3] = 0.5;                        // zero-area RoI
                } else {
                    random_box(r);
                }
            }
        }
        Buf<double> gt(gtv.size() ? gtv.size() : 4);
        Buf<int> cls(clsv.size() ? clsv.size() : 1);
        if (!gtv.empty()) { memcpy(gt.p, gtv.data(), gtv.size() * sizeof(double)); memcpy(cls.p, clsv.data(), clsv.size() * sizeof(int)); }
        Buf<float> det((size_t)B * S * s0), valid((size_t)B * S), reg((size_t)B * S * 8), fit((size_t)B * S * fnum);
        CHECK(denet_host_detect_targets(gt.p, off.p, cls.p, roi.p, B, S, s0, null_class, fnum, c.jointfit, 0.5, 0.5, det.p,
                                        c.reg ? valid.p : nullptr, c.reg ? reg.p : nullptr, c.indfit ? fit.p : nullptr) == 0);
        for (size_t m = 0; m < (size_t)B * S; ++m) {           // every row is a distribution scaled by 1/S
            double s = 0;
            for (int k = 0; k < s0; ++k) { CHECK(det[m * s0 + k] >= 0.f); s += det[m * s0 + k]; }
            CHECK(fabs(s * S - 1.0) < 1e-5);
            if (c.indfit) {
                double f = 0;
                for (int k = 0; k < fnum; ++k) f += fit[m * fnum + k];
                CHECK(fabs(f * S - 1.0) < 1e-5);
            }
        }
    }
}

static void test_samples_and_cluster() {
    const int B = 2, S = 40, H = 16, W = 20;
    Buf<int> box((size_t)B * S * 4), count(B);
    Buf<float> absd((size_t)B * S), samples((size_t)B * S * 5);
    count[0] = S; count[1] = 7;
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < count[b]; ++i) {
            int* bx = box.p + ((size_t)b * S + i) * 4;
            const int x0 = (int)(urand() * (W - 1)), y0 = (int)(urand() * (H - 1));
            bx[0] = x0; bx[1] = y0; bx[2] = x0 + (int)(urand() * (W - x0)); bx[3] = y0 + (int)(urand() * (H - y0));
            absd[(size_t)b * S + i] = (float)(i * 0.01);
        }
    CHECK(denet_samples_finish_host(box.p, absd.p, count.p, B, S, H, W, samples.p) == 0);
    for (int i = 0; i < S; ++i) CHECK(samples[(size_t)i * 5] > 0.f && samples[(size_t)i * 5] <= 0.5f);
    for (int i = count[1]; i < S; ++i) CHECK(samples[((size_t)S + i) * 5] == 0.f);
    const float thresholds[] = {0.0f, 0.3f, 0.5f, 0.7f, 0.99f};
    for (float thr : thresholds)
        for (int out_n : {1, 9, 39}) {
            Buf<float> out((size_t)out_n * 5);
            Buf<int> n_out(1);
            CHECK(denet_host_cluster_samples(samples.p, S, thr, out_n, out.p, n_out.p) == 0);
            CHECK(n_out[0] >= 1 && n_out[0] <= out_n);
            for (int i = 1; i < n_out[0]; ++i) CHECK(out[(size_t)i * 5] <= out[(size_t)(i - 1) * 5]);     // ranked
        }
    Buf<float> out(5 * (size_t)S);
    Buf<int> n_out(1);
    CHECK(denet_host_cluster_samples(samples.p, S, 0.5f, S, out.p, n_out.p) != 0);    // needs more candidates than outputs
}

static void test_soft_nms() {
    for (int n : {1, 2, 17, 300}) {
        Buf<float> score(n), box((size_t)n * 4), out_score(n);
        Buf<int> order(n), out_n(1);
        for (int i = 0; i < n; ++i) {
            double b[4];
            random_box(b);
            for (int k = 0; k < 4; ++k) box[(size_t)i * 4 + k] = (float)b[k];
            score[i] = (float)log(0.01 + 0.99 * urand());
        }
        if (n > 1) memcpy(box.p + 4, box.p, 4 * sizeof(float));      // an exact duplicate
        for (float thr : {0.1f, 0.5f, 0.9f}) {
            CHECK(denet_soft_nms_host(score.p, box.p, n, thr, order.p, out_score.p, out_n.p) == 0);
            CHECK(out_n[0] >= 1 && out_n[0] <= n);
            for (int i = 0; i < out_n[0]; ++i) CHECK(order[i] >= 0 && order[i] < n);
        }
    }
    const int B = 2, S = 12, ncls = 3, C1 = ncls + 1;
    Buf<float> det_pr((size_t)B * S * C1), fitness((size_t)B * S * C1), bbox((size_t)B * S * 4);
    Buf<int> counts(B);
    counts[0] = S; counts[1] = 5;
    for (size_t i = 0; i < det_pr.n; ++i) { det_pr[i] = (float)log(0.001 + urand()); fitness[i] = det_pr[i]; }
    for (size_t i = 0; i < (size_t)B * S; ++i) {
        double b[4];
        random_box(b);
        for (int k = 0; k < 4; ++k) bbox[i * 4 + k] = (float)b[k];
    }
    for (float nms : {0.3f, 1.0f}) {
        const long cap = (long)B * S * ncls;
        Buf<float> out_score(cap);
        Buf<int> out_cls(cap), out_row(cap), out_count(B);
        const long total = denet_soft_nms_batch_host(det_pr.p, fitness.p, bbox.p, counts.p, B, S, ncls, 0.05f, nms, out_score.p, out_cls.p,
                                                     out_row.p, out_count.p, cap);
        CHECK(total >= 0 && total <= cap && out_count[0] + out_count[1] == total);
        for (long k = 0; k < total; ++k) CHECK(out_cls[k] >= 0 && out_cls[k] < ncls && out_row[k] >= 0 && out_row[k] < B * S);
        Buf<float> small(1);
        Buf<int> c1(1), r1(1), cnt(B);
        if (total > 1)      // too small an output: refused before anything is written past the end
            CHECK(denet_soft_nms_batch_host(det_pr.p, fitness.p, bbox.p, counts.p, B, S, ncls, 0.05f, nms, small.p, c1.p, r1.p, cnt.p, 1) < 0);
    }
}

static void test_resample_coeffs() {
    struct Case { int in_size; double in0, in1; int out_size, filter; };
    // Pillow filter ids: 1 = LANCZOS (support 3), 2 = BILINEAR, 3 = BICUBIC
    const Case cases[] = {{640, 0.0, 640.0, 512, 1}, {480, 13.5, 470.25, 512, 1}, {37, 0.0, 37.0, 512, 3}, {2000, 100.0, 1900.0, 64, 2},
                          {5, 0.0, 5.0, 1, 1}, {1, 0.0, 1.0, 9, 3}, {4000, 0.0, 4000.0, 48, 1}};
    for (const Case& c : cases) {
        const double scale = (c.in1 - c.in0) / c.out_size;
        const double support = (c.filter == 1 ? 3.0 : c.filter == 3 ? 2.0 : 1.0) * (scale < 1.0 ? 1.0 : scale);
        const long ksize = (long)ceil(support) * 2 + 1;
        Buf<int> bounds((size_t)c.out_size * 2), kk((size_t)c.out_size * ksize);
        const int k = denet_host_resample_coeffs(c.in_size, c.in0, c.in1, c.out_size, c.filter, bounds.p, kk.p, (long)kk.n);
        CHECK(k == ksize);
        for (int i = 0; i < c.out_size; ++i) CHECK(bounds[2 * i] >= 0 && bounds[2 * i] + bounds[2 * i + 1] <= c.in_size && bounds[2 * i + 1] <= k);
        CHECK(denet_host_resample_coeffs(c.in_size, c.in0, c.in1, c.out_size, c.filter, bounds.p, kk.p, (long)kk.n - 1) < 0);   // table too small
    }
}

int main() {
    test_random_sample();
    test_edit_samples();
    test_prefetch_stream();
    test_detect_targets();
    test_samples_and_cluster();
    test_soft_nms();
    test_resample_coeffs();
    printf("host_sanitize: ok\n");
    return 0;
}
