// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under denet_amd/ may import, link or call this file; only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the checker.
//
// CPU restatement of the reference's corner selection + RoI proposal, written from the behaviour of
// /root/reference/denet/layer/denet_sparse.cc (run_build_samples :489-557, search_corners :321-471,
// get_sample :271-308, get_bbox_hash :311-318, get_local_max :474-487, apply_cluster :165-242) and of
// /root/reference/denet/layer/denet_detect.cc (perform_nms :73-97, perform_soft_nms :35-71,
// build_detections_nms :99-173).  Plain C ABI (no Python.h), flat arrays in, flat arrays out.
//
// PARITY PIN: the reference file itself cannot be built here — it includes "theano_mod_helper.h" from
// the un-vendored Theano tree, and writing a stand-in header is not allowed — so this restatement is pinned
// by the known answers recorded in SURVEY.md §8(c) (tests/golden/build_samples_kat.json) and by
// property tests; everything else about it is "parity unpinned".
//
// Tie behaviour: like the reference this file ranks with std::partial_sort on the score alone, so the
// order inside a group of exactly equal scores is whatever libstdc++'s heap-select produces.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <list>
#include <unordered_map>
#include <vector>

namespace {

struct Corner {
    int x, y;
    float logpr;
};

struct Sample {
    float pr, x0, y0, x1, y1;
    int ix0, iy0, ix1, iy1;
    float absd;
    bool operator<(const Sample& rhs) const { return pr > rhs.pr; }   // denet_sparse.cc:78
};

inline float at(const float* pr, int Cn, int H, int W, int b, int k, int ci, int y, int x) {
    return pr[((((size_t)b * 2 + k) * Cn + ci) * H + y) * W + x];
}

// denet_sparse.cc:474-487 — note the EXCLUSIVE upper bounds after clipping to size-1
float local_max(const float* pr, int Cn, int H, int W, int b, int ci, int y, int x, int l) {
    int x0 = std::max(0, x - l), y0 = std::max(0, y - l);
    int x1 = std::min(W - 1, x + l), y1 = std::min(H - 1, y + l);
    float m = -100000.f;
    for (int yy = y0; yy < y1; ++yy)
        for (int xx = x0; xx < x1; ++xx) m = std::max(m, at(pr, Cn, H, W, b, 1, ci, yy, xx));
    return m;
}

// denet_sparse.cc:271-308
void score_box(const float* pr, int Cn, int H, int W, int b, int x0, int y0, int x1, int y1, std::vector<Sample>& out) {
    float pf = 0, pt = 0;
    pf += at(pr, Cn, H, W, b, 0, 0, y0, x0);
    pf += at(pr, Cn, H, W, b, 0, 1, y0, x1);
    pf += at(pr, Cn, H, W, b, 0, 2, y1, x0);
    pf += at(pr, Cn, H, W, b, 0, 3, y1, x1);
    pt += at(pr, Cn, H, W, b, 1, 0, y0, x0);
    pt += at(pr, Cn, H, W, b, 1, 1, y0, x1);
    pt += at(pr, Cn, H, W, b, 1, 2, y1, x0);
    pt += at(pr, Cn, H, W, b, 1, 3, y1, x1);
    if (Cn == 5) {
        int cx = (x0 + x1) / 2, cy = (y0 + y1) / 2;
        pf += at(pr, Cn, H, W, b, 0, 4, cy, cx);
        pt += at(pr, Cn, H, W, b, 1, 4, cy, cx);
    }
    Sample s;
    s.absd = std::fabs(pf - pt);                          // float overload
    s.pr = (float)(1.0 / (1.0 + std::exp(s.absd)));       // exp(float) -> expf, then double arithmetic
    s.x0 = (float)((double)x0 / W);
    s.y0 = (float)((double)y0 / H);
    s.x1 = (float)((double)(x1 + 1) / W);
    s.y1 = (float)((double)(y1 + 1) / H);
    s.ix0 = x0; s.iy0 = y0; s.ix1 = x1; s.iy1 = y1;
    out.push_back(s);
}

inline uint64_t box_key(int x0, int y0, int x1, int y1) {
    return ((uint64_t)x0 << 48) | ((uint64_t)y0 << 32) | ((uint64_t)x1 << 16) | (uint64_t)y1;
}

struct Seen {
    std::unordered_map<uint64_t, bool> m;
    bool first(int x0, int y0, int x1, int y1) {
        uint64_t k = box_key(x0, y0, x1, y1);
        if (m.count(k)) return false;
        m[k] = true;
        return true;
    }
};

// denet_sparse.cc:321-471
std::vector<Sample> search(const float* pr, int Cn, int H, int W, int b, const std::vector<std::vector<Corner>>& cl) {
    std::vector<Sample> out;
    Seen seen;
    const std::vector<Corner>&TL = cl[0], &TR = cl[1], &BL = cl[2], &BR = cl[3];
    for (const Corner& tl : TL)
        for (const Corner& br : BR) {
            if (br.x <= tl.x || br.y <= tl.y) continue;
            if (seen.first(tl.x, tl.y, br.x, br.y)) score_box(pr, Cn, H, W, b, tl.x, tl.y, br.x, br.y, out);
        }
    for (const Corner& tr : TR)
        for (const Corner& bl : BL) {
            int x1 = tr.x, y0 = tr.y, x0 = bl.x, y1 = bl.y;
            if (x1 <= x0 || y1 <= y0) continue;
            if (seen.first(x0, y0, x1, y1)) score_box(pr, Cn, H, W, b, x0, y0, x1, y1, out);
        }
    if (cl.size() == 5) {
        for (const Corner& c : cl[4]) {
            int cx = c.x, cy = c.y;
            auto consider = [&](int x0, int y0, int x1, int y1) {
                if (x0 < 0 || y0 < 0 || x1 >= W || y1 >= H || x1 <= x0 || y1 <= y0) return;
                if (seen.first(x0, y0, x1, y1)) score_box(pr, Cn, H, W, b, x0, y0, x1, y1, out);
            };
            for (const Corner& tl : TL) consider(tl.x, tl.y, tl.x + 2 * (cx - tl.x), tl.y + 2 * (cy - tl.y));
            for (const Corner& tr : TR) consider(tr.x - 2 * (tr.x - cx), tr.y, tr.x, tr.y + 2 * (cy - tr.y));
            for (const Corner& bl : BL) consider(bl.x, bl.y - 2 * (bl.y - cy), bl.x + 2 * (cx - bl.x), bl.y);
            for (const Corner& br : BR) consider(br.x - 2 * (br.x - cx), br.y - 2 * (br.y - cy), br.x, br.y);
        }
    }
    return out;
}

// ---- apply_cluster, denet_sparse.cc:165-242 (with SampleType::overlap / overlap_iou :86-102 in fp32 and ClusterType
// :105-163). Samples are visited in ranked order; a sample joins every cluster that holds a member with IoU > threshold
// (cheap reject on the cluster's bounding box first): it is appended to the LAST such cluster in list order and the other
// overlapping clusters are merged into that one. If more than output_num clusters remain the largest are kept (stable
// list sort by member count). Every cluster then contributes its 1 + floor(size * ratio) best members,
// ratio = (output_num - #clusters) / (#samples - #clusters) in double.
float s_overlap(const Sample& a, const Sample& b) {
    float dx = std::max(0.0f, std::min(a.x1, b.x1) - std::max(a.x0, b.x0));
    float dy = std::max(0.0f, std::min(a.y1, b.y1) - std::max(a.y0, b.y0));
    return dx * dy;
}
float s_area(const Sample& a) { return (a.x1 - a.x0) * (a.y1 - a.y0); }
float s_iou(const Sample& a, const Sample& b) {
    float ai = s_overlap(a, b);
    float au = s_area(a) + s_area(b) - ai;
    return ai / au;
}
struct Cluster {
    Sample bbox;
    std::vector<std::vector<Sample>> sv;      // sv[0] grows by add_sample; merged clusters append their vectors
    size_t count() const {
        size_t n = 0;
        for (const auto& v : sv) n += v.size();
        return n;
    }
    void bounds(const Sample& s) {
        bbox.pr = std::max(s.pr, bbox.pr);
        bbox.x0 = std::min(s.x0, bbox.x0);
        bbox.y0 = std::min(s.y0, bbox.y0);
        bbox.x1 = std::max(s.x1, bbox.x1);
        bbox.y1 = std::max(s.y1, bbox.y1);
    }
    bool overlaps(const Sample& s, float thr) const {
        if (s_overlap(s, bbox) == 0) return false;
        for (const auto& v : sv)
            for (const Sample& m : v)
                if (s_iou(s, m) > thr) return true;
        return false;
    }
};
void apply_cluster(std::vector<Sample>& samples, float threshold, size_t input_num, size_t output_num) {
    if (samples.size() > input_num) {
        std::partial_sort(samples.begin(), samples.begin() + input_num, samples.end());
        samples.resize(input_num);
    }
    std::list<Cluster> clusters;
    for (const Sample& s : samples) {
        std::vector<std::list<Cluster>::iterator> hit;
        for (auto it = clusters.begin(); it != clusters.end(); ++it)
            if (it->overlaps(s, threshold)) hit.push_back(it);
        if (!hit.empty()) {
            auto tgt = hit.back();
            hit.pop_back();
            tgt->sv[0].push_back(s);
            tgt->bounds(s);
            for (auto it : hit) {
                tgt->bounds(it->bbox);
                for (auto& v : it->sv) tgt->sv.push_back(std::move(v));
                clusters.erase(it);
            }
        } else {
            Cluster c;
            c.bbox = s;
            c.sv.push_back(std::vector<Sample>(1, s));
            clusters.push_back(std::move(c));
        }
    }
    if (clusters.size() > output_num) {
        clusters.sort([](const Cluster& a, const Cluster& b) { return a.count() > b.count(); });
        clusters.resize(output_num);
    }
    const double ratio = (double)(output_num - clusters.size()) / (double)(samples.size() - clusters.size());
    samples.resize(0);
    for (const Cluster& c : clusters) {
        std::vector<Sample> all;
        for (const auto& v : c.sv) all.insert(all.end(), v.begin(), v.end());
        const size_t n = 1 + (size_t)std::floor(all.size() * ratio);
        std::partial_sort(all.begin(), all.begin() + n, all.end());
        samples.insert(samples.end(), all.begin(), all.begin() + n);
    }
}

}  // namespace

// corner_pr: [B,2,Cn,H,W] fp32 C-contiguous.  Outputs (caller allocated):
//   out_samples [B,sample_count,5] = pr,x0,y0,x1,y1 ; out_box [B,sample_count,4] integer cells ;
//   out_absd [B,sample_count] ; out_count [B].  cluster_threshold >= 1 disables clustering (the default path).
extern "C" int oracle_build_samples(const float* corner_pr, int B, int Cn, int H, int W, float corner_threshold,
                                    int sample_num, int max_corners, int local_max_r, float cluster_threshold,
                                    float* out_samples, int* out_box, float* out_absd, int* out_count) {
    const size_t sample_count = (size_t)sample_num * sample_num;
    const float thr = std::log(corner_threshold);   // float overload, denet_sparse.cc:503
    for (int b = 0; b < B; ++b) {
        std::vector<std::vector<Corner>> cl(Cn);
        for (int ci = 0; ci < Cn; ++ci) {
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    float lp = at(corner_pr, Cn, H, W, b, 1, ci, y, x);
                    if (lp > thr) {
                        if (local_max_r > 0 && lp < local_max(corner_pr, Cn, H, W, b, ci, y, x, local_max_r)) continue;
                        cl[ci].push_back(Corner{x, y, lp});
                    }
                }
            if ((int)cl[ci].size() > max_corners) {
                std::partial_sort(cl[ci].begin(), cl[ci].begin() + max_corners, cl[ci].end(),
                                  [](const Corner& a, const Corner& c) { return a.logpr > c.logpr; });
                cl[ci].resize(max_corners);
            }
        }
        std::vector<Sample> s = search(corner_pr, Cn, H, W, b, cl);
        if (s.size() > sample_count && cluster_threshold < 1.0f)          // denet_sparse.cc:541-542
            apply_cluster(s, cluster_threshold, 10 * sample_count, sample_count);
        std::partial_sort(s.begin(), s.begin() + std::min(s.size(), sample_count), s.end());
        if (s.size() > sample_count) s.resize(sample_count);
        out_count[b] = (int)s.size();
        for (size_t i = 0; i < sample_count; ++i) {
            float* o = out_samples + ((size_t)b * sample_count + i) * 5;
            int* ob = out_box + ((size_t)b * sample_count + i) * 4;
            if (i < s.size()) {
                o[0] = s[i].pr; o[1] = s[i].x0; o[2] = s[i].y0; o[3] = s[i].x1; o[4] = s[i].y1;
                ob[0] = s[i].ix0; ob[1] = s[i].iy0; ob[2] = s[i].ix1; ob[3] = s[i].iy1;
                out_absd[(size_t)b * sample_count + i] = s[i].absd;
            } else {
                o[0] = o[1] = o[2] = o[3] = o[4] = 0.f;
                ob[0] = ob[1] = ob[2] = ob[3] = 0;
                out_absd[(size_t)b * sample_count + i] = 0.f;
            }
        }
    }
    return 0;
}

// apply_cluster (denet_sparse.cc:165-242) + the final ranking (:543-545) on a candidate list that is ALREADY ranked (the n <=
// 10 * output_num best of an image, rows of pr, x0, y0, x1, y1 in ranked order): the grouping walks the candidates in the order
// given. Tests hand over the PRODUCT's ranked list, whose order inside a group of exactly equal scores the reference leaves to
// std::partial_sort (README: ties) - the clustered result is then comparable exactly although the inputs hold ties.
extern "C" int oracle_cluster_ranked(const float* ranked, int n, float cluster_threshold, int output_num, float* out_samples,
                                     int* out_count) {
    std::vector<Sample> s((size_t)n);
    for (int i = 0; i < n; ++i) {
        const float* r = ranked + (size_t)i * 5;
        Sample t = Sample();
        t.pr = r[0]; t.x0 = r[1]; t.y0 = r[2]; t.x1 = r[3]; t.y1 = r[4];
        s[(size_t)i] = t;
    }
    if ((size_t)n > (size_t)output_num && cluster_threshold < 1.0f) apply_cluster(s, cluster_threshold, (size_t)n, (size_t)output_num);
    std::stable_sort(s.begin(), s.end());          // (stable: equal scores keep the clustered order; tests compare such groups as sets)
    if (s.size() > (size_t)output_num) s.resize((size_t)output_num);
    *out_count = (int)s.size();
    for (size_t i = 0; i < s.size(); ++i) {
        float* o = out_samples + i * 5;
        o[0] = s[i].pr; o[1] = s[i].x0; o[2] = s[i].y0; o[3] = s[i].x1; o[4] = s[i].y1;
    }
    return 0;
}

// number of corners per type after thresholding / local max / truncation (diagnostics for tests)
extern "C" int oracle_count_corners(const float* corner_pr, int B, int Cn, int H, int W, float corner_threshold,
                                    int max_corners, int local_max_r, int* out_counts) {
    const float thr = std::log(corner_threshold);
    for (int b = 0; b < B; ++b)
        for (int ci = 0; ci < Cn; ++ci) {
            int n = 0;
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    float lp = at(corner_pr, Cn, H, W, b, 1, ci, y, x);
                    if (lp > thr) {
                        if (local_max_r > 0 && lp < local_max(corner_pr, Cn, H, W, b, ci, y, x, local_max_r)) continue;
                        n++;
                    }
                }
            out_counts[b * Cn + ci] = std::min(n, max_corners);
        }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// detection list building: denet_detect.cc:99-173 build_detections_nms, :73-97 perform_nms (an instance is
// dropped iff a strictly better-scored instance overlaps it by more than the threshold; input order is kept;
// NMS is skipped when the threshold is outside (0,1)), :35-71 perform_soft_nms (Gaussian, scores stay in the
// log domain: score -= iou^2/threshold, instances below -6.9 are discarded, output in selection order).
// det_pr / fitness: [B,C+1,sn,sn] log-probabilities ; bbox: [B,sn,sn,4] ; bbox_num[B] = valid RoIs per image.
// out: [B,max_out,6] = exp(score), cls, x0,y0,x1,y1 ; out_count[B] (total found, may exceed max_out).
// ---------------------------------------------------------------------------------------------------------
namespace {
struct Inst {
    float score, x0, y0, x1, y1;
    int cls;
};
float inst_iou(const Inst& a, const Inst& b) {
    float dx = std::max(0.0f, std::min(a.x1, b.x1) - std::max(a.x0, b.x0));
    float dy = std::max(0.0f, std::min(a.y1, b.y1) - std::max(a.y0, b.y0));
    float ai = dx * dy;
    float aa = (a.x1 - a.x0) * (a.y1 - a.y0);
    float ab = (b.x1 - b.x0) * (b.y1 - b.y0);
    float au = aa + ab - ai;
    return ai / au;
}
std::vector<Inst> soft_nms(const std::vector<Inst>& in, float thr, float discard = -6.9f) {
    std::vector<Inst> D;
    std::vector<Inst> Bv(in);
    while (!Bv.empty()) {
        size_t m = 0;
        for (size_t k = 0; k < Bv.size(); ++k)
            if (Bv[k].score > Bv[m].score) m = k;
        Inst M = Bv[m];
        D.push_back(M);
        Bv.erase(Bv.begin() + m);
        for (Inst& e : Bv) {
            float iou = inst_iou(M, e);
            e.score -= iou * iou / thr;
        }
        std::vector<Inst> keep;
        for (const Inst& e : Bv)
            if (!(e.score < discard)) keep.push_back(e);
        Bv.swap(keep);
    }
    return D;
}
std::vector<Inst> nms(const std::vector<Inst>& in, float thr, bool soft) {
    if (thr <= 0.0f || thr >= 1.0f || in.empty()) return in;
    if (soft) return soft_nms(in, thr);
    std::vector<Inst> out;
    for (const Inst& a : in) {
        bool unique = true;
        for (const Inst& b : in)
            if (a.score < b.score && inst_iou(a, b) > thr) { unique = false; break; }
        if (unique) out.push_back(a);
    }
    return out;
}
}  // namespace

extern "C" int oracle_build_detections_nms(float pr_threshold, float nms_threshold, int use_soft_nms, const float* det_pr,
                                           const float* fitness, const float* bbox, const int* bbox_num, int B, int C1,
                                           int sn, int max_out, float* out, int* out_count) {
    const int class_num = C1 - 1;
    const float log_thr = std::log(pr_threshold);
    for (int b = 0; b < B; ++b) {
        std::vector<Inst> all;
        for (int cls = 0; cls < class_num; ++cls) {
            std::vector<Inst> inst;
            for (int j = 0; j < sn && j * sn < bbox_num[b]; ++j)
                for (int i = 0; i < sn && j * sn + i < bbox_num[b]; ++i) {
                    size_t o = (((size_t)b * C1 + cls) * sn + j) * sn + i;
                    if (det_pr[o] >= log_thr) {
                        const float* bx = bbox + (((size_t)b * sn + j) * sn + i) * 4;
                        inst.push_back(Inst{fitness[o], bx[0], bx[1], bx[2], bx[3], cls});
                    }
                }
            std::vector<Inst> kept = nms(inst, nms_threshold, use_soft_nms != 0);
            all.insert(all.end(), kept.begin(), kept.end());
        }
        out_count[b] = (int)all.size();
        int n = std::min((int)all.size(), max_out);
        for (int k = 0; k < n; ++k) {
            float* o = out + ((size_t)b * max_out + k) * 6;
            o[0] = std::exp(all[k].score); o[1] = (float)all[k].cls;
            o[2] = all[k].x0; o[3] = all[k].y0; o[4] = all[k].x1; o[5] = all[k].y1;
        }
    }
    return 0;
}
