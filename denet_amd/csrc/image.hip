// Device-side rendering of the data pipeline's augmentation plan (the per-step host work §8e names as
// the scaling risk): the host decodes an image to u8 and draws the random crop / colour decisions; the pixels - border,
// crop, resampling to the network size, /255, photometric + PCA colour jitter, mean/std normalisation, mirror, NHWC
// layout - are produced here, straight into the training batch in HBM.
//
// Reference path being replaced: denet/dataset/augment.py (add_border :51-61, crop, scale :21-47 -> Pillow
// Image.thumbnail / Image.resize, image_to_array :9-17, photometric :271-285, colorspace :288-293) and
// denet/dataset/image_loader.py:71-105. The resampling arithmetic is Pillow's (third party, version 12.2.0 in this image,
// src/libImaging/Resample.c: separable two-pass convolution, horizontal pass first, each pass rounded to u8; coefficients
// computed in double, normalised, converted to 22-bit fixed point) and is reproduced BIT FOR BIT: the coefficient tables
// are built on the host with the same libm calls (denet_host_resample_coeffs), the passes are integer arithmetic.
// All kernels are tiny and HBM/latency bound: one thread per output pixel, RGBX (4 bytes) pixels for aligned access.
#include <math.h>
#include <string.h>

#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

// ---- Pillow's filters (Resample.c: bilinear_filter, bicubic_filter, lanczos_filter) ------------------------------------
double sinc_filter(double x) {
    if (x == 0.0) return 1.0;
    x = x * M_PI;
    return sin(x) / x;
}
double lanczos_filter(double x) {
    if (-3.0 <= x && x < 3.0) return sinc_filter(x) * sinc_filter(x / 3);
    return 0.0;
}
double bilinear_filter(double x) {
    if (x < 0.0) x = -x;
    if (x < 1.0) return 1.0 - x;
    return 0.0;
}
double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// dst[cy][cx] = canvas[y0 + cy][x0 + cx]; the canvas is black with the source pasted at (px, py)
__global__ __launch_bounds__(256) void image_crop_kernel(const unsigned char* __restrict__ src, uchar4* __restrict__ dst,
                                                         int sw, int sh, int bpp, int px, int py, int x0, int y0, int w,
                                                         int h) {
    const long total = (long)w * h;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cx = (int)(i % w), cy = (int)(i / w);
        const int sx = x0 + cx - px, sy = y0 + cy - py;
        uchar4 v = make_uchar4(0, 0, 0, 0);
        if ((unsigned)sx < (unsigned)sw && (unsigned)sy < (unsigned)sh) {
            const unsigned char* p = src + ((long)sy * sw + sx) * bpp;
            v = make_uchar4(p[0], p[1], p[2], 0);
        }
        dst[i] = v;
    }
}

__device__ __forceinline__ unsigned char clip8(int v) {
    v >>= PRECISION_BITS;       // arithmetic shift, like the lookup Pillow indexes with (in >> PRECISION_BITS)
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// one pass of the separable convolution: HORIZONTAL: out[y][x] = sum_k in[y][xmin(x)+k] * kk[x][k]; else along y
template <bool HORIZONTAL>
__global__ __launch_bounds__(256) void image_resample_kernel(const uchar4* __restrict__ in, uchar4* __restrict__ out,
                                                             int in_w, int out_w, int out_h,
                                                             const int* __restrict__ bounds, const int* __restrict__ kk,
                                                             int ksize) {
    const long total = (long)out_w * out_h;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % out_w), y = (int)(i / out_w);
        const int o = HORIZONTAL ? x : y;
        const int lo = bounds[2 * o], n = bounds[2 * o + 1];
        const int* k = kk + (long)o * ksize;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int t = 0; t < n; ++t) {
            const uchar4 p = HORIZONTAL ? in[(long)y * in_w + lo + t] : in[(long)(lo + t) * in_w + x];
            const int c = k[t];
            s0 += (int)p.x * c;
            s1 += (int)p.y * c;
            s2 += (int)p.z * c;
        }
        out[i] = make_uchar4(clip8(s0), clip8(s1), clip8(s2), 0);
    }
}

// Pillow's Image.reduce((fx, fy)) for 8-bit pixels (src/libImaging/Reduce.c): box average in fixed point,
// out = ((sum + n/2) * (2^24 / n)) >> 24 with n the pixels of the block (partial blocks at the right / bottom edge are
// averaged over the pixels they have); output size = ceil(in / f)
__global__ __launch_bounds__(256) void image_reduce_kernel(const uchar4* __restrict__ in, uchar4* __restrict__ out, int in_w,
                                                           int in_h, int out_w, int out_h, int fx, int fy) {
    const long total = (long)out_w * out_h;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % out_w), y = (int)(i / out_w);
        const int x1 = min((x + 1) * fx, in_w), y1 = min((y + 1) * fy, in_h);
        unsigned s0 = 0, s1 = 0, s2 = 0;
        for (int yy = y * fy; yy < y1; ++yy)
            for (int xx = x * fx; xx < x1; ++xx) {
                const uchar4 p = in[(long)yy * in_w + xx];
                s0 += p.x;
                s1 += p.y;
                s2 += p.z;
            }
        const unsigned n = (unsigned)((x1 - x * fx) * (y1 - y * fy));
        const unsigned long long m = (1u << 24) / n, amend = n / 2;
        out[i] = make_uchar4((unsigned char)(((s0 + amend) * m) >> 24), (unsigned char)(((s1 + amend) * m) >> 24),
                             (unsigned char)(((s2 + amend) * m) >> 24), 0);
    }
}

// exact per-channel sums of an RGBX image (for the grey mean of the "contrast" jitter): integer atomics, order free
__global__ __launch_bounds__(256) void image_sum_kernel(const uchar4* __restrict__ img, long n,
                                                        unsigned long long* __restrict__ sums) {
    __shared__ unsigned long long red[3][256];
    unsigned long long a = 0, b = 0, c = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const uchar4 p = img[i];
        a += p.x;
        b += p.y;
        c += p.z;
    }
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = b;
    red[2][threadIdx.x] = c;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s)
            for (int ch = 0; ch < 3; ++ch) red[ch][threadIdx.x] += red[ch][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 3) atomicAdd(&sums[threadIdx.x], red[threadIdx.x][0]);
}

struct FinishParams {
    int n_ops;            // photometric ops in application order
    int op[3];            // 0 brightness, 1 contrast, 2 saturation
    float alpha[3];       // fp32(alpha) and fp32(1 - alpha) (numpy multiplies the fp32 image by the python scalar)
    float one_minus[3];
    double alpha_d[3];    // the same factors in double, for the running mean of the grey image
    int use_noise;        // PCA colour noise (added in double, rounded once, like `im_x += noise`)
    double noise[3];
    int subtract_mean;
    float mean[3], std[3];
    int mirror;
};

// u8 RGBX -> fp32 NHWC slot of the batch: /255, photometric ops applied one after the other in fp32 in the reference's
// expression order (no FMA contraction in this file), colour noise, mean/std, horizontal mirror
__global__ __launch_bounds__(256) void image_finish_kernel(const uchar4* __restrict__ img, float* __restrict__ out, int w,
                                                           int h, int cp, FinishParams fp,
                                                           const unsigned long long* __restrict__ sums) {
    // mean of the grey image before each contrast op: grey is linear in RGB and every op is affine with the same
    // coefficients for all pixels, so it follows from the exact RGB sums of the u8 image
    float contrast_mean[3] = {0.f, 0.f, 0.f};
    if (fp.n_ops > 0) {
        const double npix = (double)w * (double)h;
        double m[3];
        for (int c = 0; c < 3; ++c) m[c] = sums ? (double)sums[c] / npix / 255.0 : 0.0;
        for (int o = 0; o < fp.n_ops; ++o) {
            const double a = fp.alpha_d[o];
            const double g = 0.299 * m[0] + 0.587 * m[1] + 0.114 * m[2];
            if (fp.op[o] == 0) {
                for (int c = 0; c < 3; ++c) m[c] *= a;
            } else if (fp.op[o] == 1) {
                contrast_mean[o] = (float)g;
                for (int c = 0; c < 3; ++c) m[c] = m[c] * a + (1.0 - a) * g;
            } else {
                for (int c = 0; c < 3; ++c) m[c] = m[c] * a + (1.0 - a) * g;   // mean of grey[None] is g again
            }
        }
    }
    const long total = (long)w * h;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % w), y = (int)(i / w);
        const uchar4 p = img[i];
        float v[3] = {(float)p.x / 255.0f, (float)p.y / 255.0f, (float)p.z / 255.0f};
        for (int o = 0; o < fp.n_ops; ++o) {
            const float a = fp.alpha[o], b = fp.one_minus[o];
            if (fp.op[o] == 0) {
                for (int c = 0; c < 3; ++c) v[c] = v[c] * a;
            } else if (fp.op[o] == 1) {
                const float t = b * contrast_mean[o];
                for (int c = 0; c < 3; ++c) v[c] = v[c] * a + t;
            } else {
                const float grey = (0.299f * v[0] + 0.587f * v[1]) + 0.114f * v[2];
                const float t = b * grey;
                for (int c = 0; c < 3; ++c) v[c] = v[c] * a + t;
            }
        }
        if (fp.use_noise)
            for (int c = 0; c < 3; ++c) v[c] = (float)((double)v[c] + fp.noise[c]);
        if (fp.subtract_mean)
            for (int c = 0; c < 3; ++c) v[c] = (v[c] - fp.mean[c]) / fp.std[c];
        const int ox = fp.mirror ? w - 1 - x : x;
        float* q = out + ((long)y * w + ox) * cp;
        q[0] = v[0];
        q[1] = v[1];
        q[2] = v[2];
        for (int c = 3; c < cp; ++c) q[c] = 0.f;
    }
}

int grid_for(long total) {
    long b = (total + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

// Pillow precompute_coeffs + normalize_coeffs_8bpc (Resample.c) for one axis: filter 1 = LANCZOS, 2 = BILINEAR, 3 = BICUBIC
// (Pillow's Resampling enum); 4 = NEAREST (not Pillow's number for it, which is 0: a nearest-neighbour index table, below). bounds_host: [out_size][2] = (first input index, tap count); kk_host: [out_size][ksize]
// fixed-point taps (22 fractional bits), zero padded. Returns ksize (> 0), or a negative error code.
extern "C" int denet_host_resample_coeffs(int in_size, double in0, double in1, int out_size, int filter, int* bounds_host,
                                          int* kk_host, long kk_capacity) {
    DENET_CHECK_ARG(bounds_host && kk_host, "resample_coeffs: null pointer");
    DENET_CHECK_ARG(in_size > 0 && out_size > 0 && in1 > in0, "resample_coeffs: bad sizes");
    if (filter == 4) {
        // NEAREST is no convolution in Pillow: Image.resize hands it to ImagingScaleAffine (Geometry.c), which steps a double
        // through the source, xo = in0 + a/2, += a with a = (in1 - in0) / out_size, and copies pixel (int)xo (COORD: -1 below
        // zero). The same additions in the same order here; as a table it is one tap of weight 1 (the pass kernel's rounding
        // term and shift return the byte unchanged), no tap where Pillow leaves the fill colour (0).
        DENET_CHECK_ARG(out_size <= kk_capacity, "resample_coeffs: table needs %d ints, capacity %ld", out_size, kk_capacity);
        const double a = (in1 - in0) / out_size;
        double xo = in0 + a * 0.5;
        for (int xx = 0; xx < out_size; ++xx) {
            const int xin = xo < 0.0 ? -1 : (int)xo;
            const bool inside = xin >= 0 && xin < in_size;
            bounds_host[2 * xx] = inside ? xin : 0;
            bounds_host[2 * xx + 1] = inside ? 1 : 0;
            kk_host[xx] = inside ? (1 << PRECISION_BITS) : 0;
            xo += a;
        }
        return 1;
    }
    double (*f)(double) = nullptr;
    double support = 0.0;
    if (filter == 1) { f = lanczos_filter; support = 3.0; }
    else if (filter == 2) { f = bilinear_filter; support = 1.0; }
    else if (filter == 3) { f = bicubic_filter; support = 2.0; }
    DENET_CHECK_ARG(f != nullptr, "resample_coeffs: filter %d is none of 1 lanczos, 2 bilinear, 3 bicubic, 4 nearest", filter);
    double scale, filterscale;
    filterscale = scale = (in1 - in0) / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    support = support * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    DENET_CHECK_ARG((long)out_size * ksize <= kk_capacity, "resample_coeffs: table needs %ld ints, capacity %ld",
                    (long)out_size * ksize, kk_capacity);
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = in0 + (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double w[64];
        double* wp = w;
        double* big = nullptr;
        if (ksize > 64) wp = big = new double[ksize];
        int x;
        for (x = 0; x < xmax; ++x) {
            const double v = f((x + xmin - center + 0.5) * ss);
            wp[x] = v;
            ww += v;
        }
        int* k = kk_host + (long)xx * ksize;
        for (x = 0; x < xmax; ++x) {
            double v = wp[x];
            if (ww != 0.0) v /= ww;
            k[x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
        }
        for (; x < ksize; ++x) k[x] = 0;
        bounds_host[2 * xx] = xmin;
        bounds_host[2 * xx + 1] = xmax;
        delete[] big;
    }
    return ksize;
}

// dst (RGBX, w x h) = window (x0, y0, w, h) of the black canvas on which the source (sw x sh, src_bpp = 3: packed RGB as
// decoded, 4: RGBX from a previous step) is pasted at (px, py)
extern "C" int denet_image_crop(const unsigned char* src, unsigned char* dst_rgbx, int sw, int sh, int src_bpp, int px, int py,
                                int x0, int y0, int w, int h, hipStream_t stream) {
    DENET_CHECK_ARG(src && dst_rgbx, "image_crop: null pointer");
    DENET_CHECK_ARG(sw > 0 && sh > 0 && w > 0 && h > 0 && (src_bpp == 3 || src_bpp == 4), "image_crop: bad sizes");
    hipLaunchKernelGGL(image_crop_kernel, dim3(grid_for((long)w * h)), dim3(256), 0, stream, src, (uchar4*)dst_rgbx, sw, sh,
                       src_bpp, px, py, x0, y0, w, h);
    DENET_CHECK_LAUNCH("image_crop");
    return DENET_OK;
}

// Image.reduce((fx, fy)) of an RGBX image: (in_w x in_h) -> (ceil(in_w / fx) x ceil(in_h / fy)); the box pre-pass
// Image.thumbnail / resize(reducing_gap=2) insert before the convolution when the shrink factor reaches 4
extern "C" int denet_image_reduce(const unsigned char* in_rgbx, unsigned char* out_rgbx, int in_w, int in_h, int fx, int fy,
                                  hipStream_t stream) {
    DENET_CHECK_ARG(in_rgbx && out_rgbx, "image_reduce: null pointer");
    DENET_CHECK_ARG(in_w > 0 && in_h > 0 && fx > 0 && fy > 0 && fx * fy <= 65536, "image_reduce: bad sizes");
    const int ow = (in_w + fx - 1) / fx, oh = (in_h + fy - 1) / fy;
    hipLaunchKernelGGL(image_reduce_kernel, dim3(grid_for((long)ow * oh)), dim3(256), 0, stream, (const uchar4*)in_rgbx,
                       (uchar4*)out_rgbx, in_w, in_h, ow, oh, fx, fy);
    DENET_CHECK_LAUNCH("image_reduce");
    return DENET_OK;
}

// one pass of Pillow's separable resampling on an RGBX image; horizontal: (in_w x in_h) -> (out_n x in_h), else -> (in_w x out_n)
extern "C" int denet_image_resample_pass(const unsigned char* in_rgbx, unsigned char* out_rgbx, int in_w, int in_h, int out_n,
                                         int horizontal, const int* bounds_dev, const int* kk_dev, int ksize,
                                         hipStream_t stream) {
    DENET_CHECK_ARG(in_rgbx && out_rgbx && bounds_dev && kk_dev, "image_resample_pass: null pointer");
    DENET_CHECK_ARG(in_w > 0 && in_h > 0 && out_n > 0 && ksize > 0, "image_resample_pass: bad sizes");
    if (horizontal)
        hipLaunchKernelGGL(image_resample_kernel<true>, dim3(grid_for((long)out_n * in_h)), dim3(256), 0, stream,
                           (const uchar4*)in_rgbx, (uchar4*)out_rgbx, in_w, out_n, in_h, bounds_dev, kk_dev, ksize);
    else
        hipLaunchKernelGGL(image_resample_kernel<false>, dim3(grid_for((long)in_w * out_n)), dim3(256), 0, stream,
                           (const uchar4*)in_rgbx, (uchar4*)out_rgbx, in_w, in_w, out_n, bounds_dev, kk_dev, ksize);
    DENET_CHECK_LAUNCH("image_resample_pass");
    return DENET_OK;
}

// RGBX u8 (w x h) -> one fp32 NHWC image of the batch (cp channels per pixel, channels >= 3 zero).
//   ops / alphas: the photometric jitter in application order (0 brightness, 1 contrast, 2 saturation; n_ops 0..3)
//   noise: PCA colour offsets (3 doubles) or NULL; mean_std: 6 floats (mean rgb, std rgb) or NULL; sums_ws: 3 device
//   uint64 used for the grey mean of the contrast op (required when n_ops > 0)
extern "C" int denet_image_finish(const unsigned char* img_rgbx, float* out_nhwc, int w, int h, int cp, int n_ops,
                                  const int* ops_host, const double* alphas_host, const double* noise_host,
                                  const float* mean_std_host, int mirror, unsigned long long* sums_ws,
                                  hipStream_t stream) {
    DENET_CHECK_ARG(img_rgbx && out_nhwc, "image_finish: null pointer");
    DENET_CHECK_ARG(w > 0 && h > 0 && cp >= 3 && n_ops >= 0 && n_ops <= 3, "image_finish: bad arguments");
    DENET_CHECK_ARG(n_ops == 0 || (ops_host && alphas_host && sums_ws), "image_finish: photometric ops need ops, alphas and sums_ws");
    FinishParams fp = {};
    fp.n_ops = n_ops;
    for (int o = 0; o < n_ops; ++o) {
        DENET_CHECK_ARG(ops_host[o] >= 0 && ops_host[o] <= 2, "image_finish: unknown photometric op %d", ops_host[o]);
        fp.op[o] = ops_host[o];
        fp.alpha_d[o] = alphas_host[o];
        fp.alpha[o] = (float)alphas_host[o];
        fp.one_minus[o] = (float)(1.0 - alphas_host[o]);
    }
    if (noise_host) {
        fp.use_noise = 1;
        for (int c = 0; c < 3; ++c) fp.noise[c] = noise_host[c];
    }
    if (mean_std_host) {
        fp.subtract_mean = 1;
        for (int c = 0; c < 3; ++c) {
            fp.mean[c] = mean_std_host[c];
            fp.std[c] = mean_std_host[3 + c];
        }
    }
    fp.mirror = mirror ? 1 : 0;
    if (n_ops > 0) {
        (void)hipMemsetAsync(sums_ws, 0, 3 * sizeof(unsigned long long), stream);
        hipLaunchKernelGGL(image_sum_kernel, dim3(grid_for((long)w * h / 4)), dim3(256), 0, stream, (const uchar4*)img_rgbx,
                           (long)w * h, sums_ws);
    }
    hipLaunchKernelGGL(image_finish_kernel, dim3(grid_for((long)w * h)), dim3(256), 0, stream, (const uchar4*)img_rgbx, out_nhwc,
                       w, h, cp, fp, n_ops > 0 ? sums_ws : nullptr);
    DENET_CHECK_LAUNCH("image_finish");
    return DENET_OK;
}

// A whole batch in one call (the Python loader thread holds no interpreter lock meanwhile): copies the decoded images and
// the coefficient tables of every resampling pass into the pinned staging buffer, issues ONE host-to-device copy and then
// the kernels of every image's program, writing image b to out_dev + b * crop * crop * cp.
//   ops: [n_ops][8] ints, per image the range op_off[b] .. op_off[b+1]:
//        kind 0 crop    : sw, sh, px, py, x0, y0 ; the window size is (aux[0], aux[1]) = ops_wh
//        kind 1 reduce  : in_w, in_h, fx, fy
//        kind 2 pass    : horizontal, in_w, in_h, out_n, filter ; ops_in1: the upper end of the source box (in0 = 0)
//   photo_ops / photo_alpha: [B][3] (photo_n[b] valid entries); noise: [B][3] doubles, used where has_noise[b];
//   mean_std: 6 floats or NULL; the caller guarantees the previous batch's copy has finished reading `pinned_host`.
extern "C" int denet_image_render_batch(int B, const unsigned char* const* src_host, const int* src_wh, const int* op_off,
                                        const int* ops, const int* ops_wh, const double* ops_in1, const int* photo_n,
                                        const int* photo_ops, const double* photo_alpha, const double* noise,
                                        const unsigned char* has_noise, const float* mean_std, const unsigned char* mirror,
                                        int crop, int cp, float* out_dev, unsigned char* pinned_host, size_t pinned_bytes,
                                        unsigned char* staging_dev, unsigned char* scratch0, unsigned char* scratch1,
                                        size_t scratch_bytes, unsigned long long* sums_dev, hipStream_t stream) {
    DENET_CHECK_ARG(B > 0 && src_host && src_wh && op_off && ops && ops_wh && ops_in1 && photo_n && photo_ops && photo_alpha,
                    "image_render_batch: null pointer");
    DENET_CHECK_ARG(out_dev && pinned_host && staging_dev && scratch0 && scratch1 && sums_dev && mirror,
                    "image_render_batch: null buffer");
    auto align16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const int n_ops = op_off[B];
    struct Tab { size_t bounds, kk; int ksize; };
    Tab* tabs = new Tab[n_ops > 0 ? n_ops : 1];
    size_t* img_off = new size_t[B];
    size_t off = 0;
    int rc = DENET_OK;
    for (int b = 0; b < B && rc == DENET_OK; ++b) {
        const size_t bytes = (size_t)src_wh[2 * b] * src_wh[2 * b + 1] * 3;
        if (off + bytes > pinned_bytes) { denet_set_error("image_render_batch: staging buffer too small"); rc = DENET_ERR_ARG; break; }
        memcpy(pinned_host + off, src_host[b], bytes);
        img_off[b] = off;
        off = align16(off + bytes);
        for (int o = op_off[b]; o < op_off[b + 1]; ++o) {
            const int* q = ops + 8 * o;
            if (q[0] != 2) continue;
            const int in_size = q[1] ? q[2] : q[3], out_n = q[4];
            const size_t bb = (size_t)2 * out_n * sizeof(int);
            if (off + bb > pinned_bytes) { denet_set_error("image_render_batch: staging buffer too small"); rc = DENET_ERR_ARG; break; }
            const long cap = (long)((pinned_bytes - off - bb) / sizeof(int));
            const int ks = denet_host_resample_coeffs(in_size, 0.0, ops_in1[o], out_n, q[5], (int*)(pinned_host + off),
                                                      (int*)(pinned_host + off + bb), cap);
            if (ks <= 0) { rc = ks < 0 ? ks : DENET_ERR_ARG; break; }
            tabs[o].bounds = off;
            tabs[o].kk = off + bb;
            tabs[o].ksize = ks;
            off = align16(off + bb + (size_t)out_n * ks * sizeof(int));
        }
    }
    if (rc == DENET_OK) {
        hipError_t e = hipMemcpyAsync(staging_dev, pinned_host, off, hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) { denet_set_error("image_render_batch: %s", hipGetErrorString(e)); rc = -(int)e; }
    }
    for (int b = 0; b < B && rc == DENET_OK; ++b) {
        const unsigned char* cur = staging_dev + img_off[b];
        int bpp = 3, cw = src_wh[2 * b], chh = src_wh[2 * b + 1], flip = 0;
        auto to_rgbx = [&]() {      // a program that starts with a resampling step: unpack the decoded image first
            unsigned char* dst = flip ? scratch1 : scratch0;
            int r = denet_image_crop(cur, dst, cw, chh, 3, 0, 0, 0, 0, cw, chh, stream);
            cur = dst; bpp = 4; flip ^= 1;
            return r;
        };
        for (int o = op_off[b]; o < op_off[b + 1] && rc == DENET_OK; ++o) {
            const int* q = ops + 8 * o;
            if (q[0] != 0 && bpp == 3) rc = to_rgbx();
            if (rc != DENET_OK) break;
            unsigned char* dst = flip ? scratch1 : scratch0;
            int ow = 0, oh = 0;
            if (q[0] == 0) {
                ow = ops_wh[2 * o]; oh = ops_wh[2 * o + 1];
                if ((size_t)ow * oh * 4 > scratch_bytes) { denet_set_error("image_render_batch: scratch too small"); rc = DENET_ERR_ARG; break; }
                rc = denet_image_crop(cur, dst, q[1], q[2], bpp, q[3], q[4], q[5], q[6], ow, oh, stream);
            } else if (q[0] == 1) {
                ow = (q[1] + q[3] - 1) / q[3]; oh = (q[2] + q[4] - 1) / q[4];
                rc = denet_image_reduce(cur, dst, q[1], q[2], q[3], q[4], stream);
            } else if (q[0] == 2) {
                ow = q[1] ? q[4] : q[2]; oh = q[1] ? q[3] : q[4];
                if ((size_t)ow * oh * 4 > scratch_bytes) { denet_set_error("image_render_batch: scratch too small"); rc = DENET_ERR_ARG; break; }
                rc = denet_image_resample_pass(cur, dst, q[2], q[3], q[4], q[1], (const int*)(staging_dev + tabs[o].bounds),
                                               (const int*)(staging_dev + tabs[o].kk), tabs[o].ksize, stream);
            } else {
                denet_set_error("image_render_batch: unknown op kind %d", q[0]);
                rc = DENET_ERR_ARG;
            }
            cur = dst; bpp = 4; flip ^= 1; cw = ow; chh = oh;
        }
        if (rc == DENET_OK && bpp == 3) rc = to_rgbx();
        if (rc == DENET_OK && (cw != crop || chh != crop)) {
            denet_set_error("image_render_batch: image %d renders %dx%d, expected %dx%d", b, cw, chh, crop, crop);
            rc = DENET_ERR_ARG;
        }
        if (rc == DENET_OK)
            rc = denet_image_finish(cur, out_dev + (size_t)b * crop * crop * cp, crop, crop, cp, photo_n[b], photo_ops + 3 * b,
                                    photo_alpha + 3 * b, (has_noise && has_noise[b]) ? noise + 3 * b : nullptr, mean_std,
                                    mirror[b], sums_dev, stream);
    }
    delete[] tabs;
    delete[] img_off;
    return rc;
}
