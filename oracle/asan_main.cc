// TEST INFRASTRUCTURE (see the header of build_samples.cc). `make -C oracle asan`: the C++ checker of the RoI proposal and of
// the inference tail under AddressSanitizer + UndefinedBehaviorSanitizer, on random corner maps - sparse, dense, more than
// max_corners per type (the truncation branch, denet_sparse.cc:526-530), local maximum windows, the 5-map centre variant,
// RoI clustering (denet_sparse.cc:165-242) and both NMS variants (denet_detect.cc:35-97) - with every caller buffer
// heap-allocated at exactly its documented size.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

extern "C" int oracle_build_samples(const float* corner_pr, int B, int Cn, int H, int W, float corner_threshold, int sample_num,
                                    int max_corners, int local_max_r, float cluster_threshold, float* out_samples, int* out_box,
                                    float* out_absd, int* out_count);
extern "C" int oracle_count_corners(const float* corner_pr, int B, int Cn, int H, int W, float corner_threshold, int max_corners,
                                    int local_max_r, int* out_counts);
extern "C" int oracle_build_detections_nms(float pr_threshold, float nms_threshold, int use_soft_nms, const float* det_pr,
                                           const float* fitness, const float* bbox, const int* bbox_num, int B, int C1, int sn,
                                           int max_out, float* out, int* out_count);

static uint32_t state = 2463534242u;
static double urand() { state ^= state << 13; state ^= state >> 17; state ^= state << 5; return (state >> 8) / 16777216.0; }

template <typename T>
static T* alloc(size_t n) { return (T*)calloc(n ? n : 1, sizeof(T)); }

int main() {
    int cases = 0;
    for (int c = 0; c < 60; ++c) {
        const int Cn = c % 5 == 4 ? 5 : 4, B = 1 + c % 2, H = 5 + (int)(urand() * 20), W = 5 + (int)(urand() * 20);
        const double mu = -3.5 + (c % 4);
        const int max_corners = (c / 3) % 3 == 2 ? 7 : 1024, sample_num = (c / 2) % 3 == 0 ? 24 : 4, local_max = c % 3;
        const float cluster = c % 6 == 5 ? 0.5f : 1.0f;
        const size_t cells = (size_t)B * Cn * H * W, S = (size_t)sample_num * sample_num;
        float* pr = alloc<float>(cells * 2);
        for (int b = 0; b < B; ++b)
            for (size_t i = 0; i < (size_t)Cn * H * W; ++i) {
                const double z = mu + 4.0 * (urand() + urand() + urand() - 1.5), p = 1.0 / (1.0 + exp(-z));
                pr[((size_t)b * 2 + 0) * Cn * H * W + i] = (float)log1p(-p);
                pr[((size_t)b * 2 + 1) * Cn * H * W + i] = (float)log(p);
            }
        float* samples = alloc<float>(B * S * 5);
        int* box = alloc<int>(B * S * 4);
        float* absd = alloc<float>(B * S);
        int* count = alloc<int>(B);
        int* ncorner = alloc<int>((size_t)B * Cn);
        if (oracle_build_samples(pr, B, Cn, H, W, 0.05f, sample_num, max_corners, local_max, cluster, samples, box, absd, count)) return 1;
        if (oracle_count_corners(pr, B, Cn, H, W, 0.05f, max_corners, local_max, ncorner)) return 1;
        for (int b = 0; b < B; ++b) {
            if (count[b] < 0 || (size_t)count[b] > S) return 2;
            for (int i = 1; i < count[b]; ++i)
                if (samples[((size_t)b * S + i) * 5] > samples[((size_t)b * S + i - 1) * 5]) return 3;      // ranked by score
        }
        free(pr); free(samples); free(box); free(absd); free(count); free(ncorner);
        ++cases;
    }
    for (int c = 0; c < 12; ++c) {
        const int B = 2, sn = 3 + c % 4, ncls = 1 + c % 5, C1 = ncls + 1, max_out = sn * sn * ncls;
        float* det = alloc<float>((size_t)B * C1 * sn * sn);
        float* fit = alloc<float>((size_t)B * C1 * sn * sn);
        float* bbox = alloc<float>((size_t)B * sn * sn * 4);
        int* num = alloc<int>(B);
        num[0] = sn * sn; num[1] = c % (sn * sn);
        for (size_t i = 0; i < (size_t)B * C1 * sn * sn; ++i) { det[i] = (float)log(0.001 + urand()); fit[i] = det[i] + (float)(0.1 * urand()); }
        for (size_t i = 0; i < (size_t)B * sn * sn; ++i) {
            const double x0 = 0.6 * urand(), y0 = 0.6 * urand();
            bbox[i * 4] = (float)x0; bbox[i * 4 + 1] = (float)y0; bbox[i * 4 + 2] = (float)(x0 + 0.4 * urand()); bbox[i * 4 + 3] = (float)(y0 + 0.4 * urand());
        }
        float* out = alloc<float>((size_t)B * max_out * 6);
        int* out_count = alloc<int>(B);
        for (int soft = 0; soft < 2; ++soft)
            if (oracle_build_detections_nms(0.05f, c % 3 == 2 ? 1.0f : 0.4f, soft, det, fit, bbox, num, B, C1, sn, max_out, out, out_count)) return 4;
        if (out_count[0] < 0 || out_count[0] > max_out) return 5;
        free(det); free(fit); free(bbox); free(num); free(out); free(out_count);
        ++cases;
    }
    printf("oracle_asan: ok (%d cases)\n", cases);
    return 0;
}
