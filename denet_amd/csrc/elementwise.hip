// Small HBM-bound helpers of the hot path (NHWC fp32):
//   layout conversion at the boundary (the reference hands NCHW float32 batches to train_step,
//   denet/dataset/__init__.py:349-366; model_cnn.py:407), residual / skip adds (denet/layer/skip.py:81-86),
//   the plain `A` relu layer (denet/layer/activation.py:31-34), conv-bias gradient (column sums),
//   and the fused solver update (denet/model/model_cnn.py:282-294, 321-331).
#include "common.h"
#include <math.h>

namespace {

int grid_for(long total) {
    long b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

// x NCHW [N][C][H][W] -> y NHWC [N][H][W][CP], channels >= C zero filled
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int N,
                                                           int C, int H, int W, int CP) {
    const long total = (long)N * H * W * CP;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % CP);
        long t = i / CP;
        const int w = (int)(t % W);
        t /= W;
        const int h = (int)(t % H);
        const int n = (int)(t / H);
        y[i] = (c < C) ? x[(((long)n * C + c) * H + h) * W + w] : 0.f;
    }
}

// x NHWC [N][H][W][CP] -> y NCHW [N][C][H][W] (first C channels)
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int N,
                                                           int C, int H, int W, int CP) {
    const long total = (long)N * C * H * W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        long t = i / W;
        const int h = (int)(t % H);
        t /= H;
        const int c = (int)(t % C);
        const int n = (int)(t / C);
        y[i] = x[(((long)n * H + h) * W + w) * CP + c];
    }
}

// y = a + b  (optionally relu)
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ y, long n4, int relu) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 v = ((const f32x4*)a)[i] + ((const f32x4*)b)[i];
        if (relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        ((f32x4*)y)[i] = v;
    }
}

__global__ __launch_bounds__(256) void relu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 v = ((const f32x4*)x)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
        ((f32x4*)y)[i] = v;
    }
}

// dx = dy * (y > 0)
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                       float* __restrict__ dx, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 yv = ((const f32x4*)y)[i];
        f32x4 g = ((const f32x4*)dy)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = yv[k] > 0.f ? g[k] : 0.f;
        ((f32x4*)dx)[i] = g;
    }
}

// column sums of a [M][C] matrix in two deterministic stages (conv bias gradient)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, long M, int C, int LC,
                                                             double* __restrict__ partial) {
    __shared__ double red[256 * 4];
    const int tid = threadIdx.x;
    const int cl = tid % LC, rsub = tid / LC, RS = 256 / LC;
    const int c = (blockIdx.x * LC + cl) * 4;
    double s[4] = {0, 0, 0, 0};
    for (long r = (long)blockIdx.y * RS + rsub; r < M; r += (long)gridDim.y * RS) {
        const f32x4 v = *(const f32x4*)(x + r * C + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += (double)v[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[tid * 4 + k] = s[k];
    __syncthreads();
    if (rsub == 0) {
        for (int j = 1; j < RS; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] += red[(j * LC + cl) * 4 + k];
        double* p = partial + (long)blockIdx.y * C;
#pragma unroll
        for (int k = 0; k < 4; ++k) p[c + k] = s[k];
    }
}

__global__ __launch_bounds__(256) void colsum_final_kernel(const double* __restrict__ partial, int gy, int C,
                                                           float* __restrict__ out) {
    __shared__ double red[8][32];
    const int cl = threadIdx.x & 31, jl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double s = 0;
    if (c < C)
        for (int j = jl; j < gy; j += 8) s += partial[(long)j * C + c];
    red[jl][cl] = s;
    __syncthreads();
    if (jl != 0 || c >= C) return;
    for (int j = 1; j < 8; ++j) s += red[j][cl];
    out[c] = (float)s;
}

// model_cnn.py:282-294,321-331.  mode 0 = sgd, 1 = torch / nesterov.
//   g += decay*p for the first n_decay elements (the `weights()`; biases / BN gamma,beta follow)
//   rho = it > 0 ? mu : 0
//   sgd:      m' = rho*m + (1-rho)*g ; p' = p - lr*m'
//   nesterov: m' = rho*m + g         ; p' = p - lr*(g + mu*m')
__global__ __launch_bounds__(256) void solver_kernel(float* __restrict__ p, float* __restrict__ m,
                                                     const float* __restrict__ g, long n, long n_decay, float lr,
                                                     float mu, float rho, float decay, float gscale, int mode) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float pv = p[i];
        float gv = g[i] * gscale;
        if (i < n_decay) gv += decay * pv;
        float mv;
        if (mode == 1) {
            mv = rho * m[i] + gv;
            p[i] = pv - lr * (gv + mu * mv);
        } else {
            mv = rho * m[i] + (1.0f - rho) * gv;
            p[i] = pv - lr * mv;
        }
        m[i] = mv;
    }
}

// adam (model_cnn.py:296-305): m' = b1*m + (1-b1)*g ; v' = b2*v + (1-b2)*g*g ; p' = p - lr*(m'*c1)/(sqrt(v'*c2) + 1e-8)
// with c1 = 1/(1-b1^(it+1)), c2 = 1/(1-b2^(it+1)) evaluated on the host in double
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                   const float* __restrict__ g, long n, long n_decay, float lr, float b1,
                                                   float b2, float c1, float c2, float decay, float gscale) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float pv = p[i];
        float gv = g[i] * gscale;
        if (i < n_decay) gv += decay * pv;
        const float mv = b1 * m[i] + (1.0f - b1) * gv;
        const float vv = b2 * v[i] + (1.0f - b2) * (gv * gv);
        p[i] = pv - lr * (mv * c1) / (sqrtf(vv * c2) + 1e-8f);
        m[i] = mv;
        v[i] = vv;
    }
}

__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, long n, float s) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] *= s;
}

}  // namespace

extern "C" int denet_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, int CP, hipStream_t stream) {
    DENET_CHECK_ARG(x && y && CP >= C && C > 0, "nchw_to_nhwc: bad args");
    long total = (long)N * H * W * CP;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, y, N, C, H, W, CP);
    DENET_CHECK_LAUNCH("nchw_to_nhwc");
    return DENET_OK;
}

extern "C" int denet_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, int CP, hipStream_t stream) {
    DENET_CHECK_ARG(x && y && CP >= C && C > 0, "nhwc_to_nchw: bad args");
    long total = (long)N * C * H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, y, N, C, H, W, CP);
    DENET_CHECK_LAUNCH("nhwc_to_nchw");
    return DENET_OK;
}

extern "C" int denet_add(const float* a, const float* b, float* y, long n, int relu, hipStream_t stream) {
    DENET_CHECK_ARG(a && b && y && n % 4 == 0, "add: bad args");
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, a, b, y, n / 4, relu);
    DENET_CHECK_LAUNCH("add");
    return DENET_OK;
}

extern "C" int denet_relu_fwd(const float* x, float* y, long n, hipStream_t stream) {
    DENET_CHECK_ARG(x && y && n % 4 == 0, "relu_fwd: bad args");
    hipLaunchKernelGGL(relu_fwd_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, x, y, n / 4);
    DENET_CHECK_LAUNCH("relu_fwd");
    return DENET_OK;
}

extern "C" int denet_relu_bwd(const float* y, const float* dy, float* dx, long n, hipStream_t stream) {
    DENET_CHECK_ARG(y && dy && dx && n % 4 == 0, "relu_bwd: bad args");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, y, dy, dx, n / 4);
    DENET_CHECK_LAUNCH("relu_bwd");
    return DENET_OK;
}

extern "C" size_t denet_colsum_workspace_bytes(long M, int C) { return (size_t)2048 * C * sizeof(double); }

extern "C" int denet_colsum(const float* x, float* out, void* workspace, long M, int C, hipStream_t stream) {
    DENET_CHECK_ARG(x && out && workspace && C % 4 == 0 && M > 0, "colsum: bad args");
    int c4 = C / 4, lc = 1;
    while (lc < 256 && (c4 % (lc * 2)) == 0) lc *= 2;
    int rs = 256 / lc, gx = c4 / lc;
    long rb = (M + rs - 1) / rs;
    int gy = 1024 / gx;
    if (gy < 1) gy = 1;
    if (gy > rb) gy = (int)rb;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(gx, gy), dim3(256), 0, stream, x, M, C, lc, (double*)workspace);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 31) / 32), dim3(256), 0, stream, (const double*)workspace, gy,
                       C, out);
    DENET_CHECK_LAUNCH("colsum");
    return DENET_OK;
}

extern "C" int denet_solver_step(float* params, float* moments, const float* grads, long n, long n_decay, float lr,
                                 float momentum, int iteration, float decay, float grad_scale, int mode,
                                 hipStream_t stream) {
    DENET_CHECK_ARG(params && moments && grads && n > 0 && n_decay >= 0 && n_decay <= n, "solver_step: bad args");
    DENET_CHECK_ARG(mode == 0 || mode == 1, "solver_step: mode must be 0 (sgd) or 1 (nesterov/torch)");
    const float rho = iteration > 0 ? momentum : 0.0f;
    hipLaunchKernelGGL(solver_kernel, dim3(grid_for(n)), dim3(256), 0, stream, params, moments, grads, n, n_decay, lr,
                       momentum, rho, decay, grad_scale, mode);
    DENET_CHECK_LAUNCH("solver_step");
    return DENET_OK;
}

extern "C" int denet_solver_adam(float* params, float* m, float* v, const float* grads, long n, long n_decay, float lr,
                                 float beta1, float beta2, int iteration, float decay, float grad_scale,
                                 hipStream_t stream) {
    DENET_CHECK_ARG(params && m && v && grads && n > 0 && n_decay >= 0 && n_decay <= n, "solver_adam: bad args");
    DENET_CHECK_ARG(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && iteration >= 0, "solver_adam: bad betas");
    const float c1 = (float)(1.0 / (1.0 - pow((double)beta1, (double)iteration + 1.0)));
    const float c2 = (float)(1.0 / (1.0 - pow((double)beta2, (double)iteration + 1.0)));
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, stream, params, m, v, grads, n, n_decay, lr, beta1,
                       beta2, c1, c2, decay, grad_scale);
    DENET_CHECK_LAUNCH("solver_adam");
    return DENET_OK;
}

extern "C" int denet_scale(float* x, long n, float s, hipStream_t stream) {
    DENET_CHECK_ARG(x && n > 0, "scale: bad args");
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, n, s);
    DENET_CHECK_LAUNCH("scale");
    return DENET_OK;
}
