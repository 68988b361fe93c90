// Internal helpers shared by the HIP kernels of the DeNet hot path (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define DENET_OK 0
#define DENET_ERR_ARG (-1000)

// thread-local last error string (see include/denet_hip.h: denet_last_error)
void denet_set_error(const char* fmt, ...);

#define DENET_CHECK_ARG(cond, ...)            \
    do {                                      \
        if (!(cond)) {                        \
            denet_set_error(__VA_ARGS__);     \
            return DENET_ERR_ARG;             \
        }                                     \
    } while (0)

// map a launch failure to a negative hipError_t
#define DENET_CHECK_LAUNCH(name)                                                   \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            denet_set_error("%s: %s", name, hipGetErrorString(e__));               \
            return -(int)e__;                                                      \
        }                                                                          \
    } while (0)

// live per-launch timing (bench.py's roofline leg, denet_conv_profile; records live in igemm.hip): bracket a launch with
// begin / end on its stream. begin returns -1 while profiling is off.
int denet_prof_begin(int mode, int bm, int bn, int nbuf, hipStream_t stream);
void denet_prof_end(int idx, hipStream_t stream);

// the first layer's own kernels (stem.hip); the generic entry points of igemm.hip hand the geometry over when *_ok says so
extern "C" int denet_conv_stem_ok(int pass, int N, int H, int W, int C, int K, int R, int S, int S_real, int stride, int pad, int OH,
                                  int OW);
extern "C" int denet_conv_stem_fwd(const float* x, const float* w, const float* bias, float* y, double* stats_partial,
                                   size_t stats_bytes, int* stats_rows, int N, int H, int W, hipStream_t stream);
extern "C" int denet_conv_stem_fwd_act(const float* x, int x_nchw, const float* w, const float* bias, float* y, int relu,
                                       double* stats_partial, size_t stats_bytes, int* stats_rows, int N, int H, int W,
                                       hipStream_t stream);
extern "C" size_t denet_conv_stem_wgrad_workspace_bytes(void);
extern "C" int denet_conv_stem_wgrad(const float* x, const float* dy, float* dw, float* workspace, size_t workspace_bytes, int N,
                                     int H, int W, hipStream_t stream);

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// n / d for n < 2^31 with a precomputed multiplier (d >= 1)
struct FastDiv {
    uint32_t d, magic, shift;
    __host__ void init(uint32_t d_) {
        d = d_;
        shift = 0;
        while ((1u << shift) < d_) shift++;
        magic = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << shift) - d_)) / d_ + 1);
    }
    __device__ __forceinline__ uint32_t div(uint32_t n) const {
        return (__umulhi(n, magic) + n) >> shift;
    }
};

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// XCD-aware bijective remap of a linear workgroup id: workgroup b runs on XCD b%8; give each
// XCD a contiguous chunk of the tile space so neighbouring tiles share one L2.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t nwg) {
    const uint32_t NX = 8;
    uint32_t q = nwg / NX, r = nwg % NX;
    uint32_t xcd = b % NX, idx = b / NX;
    uint32_t base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
