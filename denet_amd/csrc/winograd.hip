// Winograd F(2x2,3x3) path for the stride-1 3x3 convolutions with many channels (forward and data gradient).
// Reference op: the same `C[k,3]` layer (denet/layer/convolution.py:80-83; its gradient model_cnn.py:318); the
// minimal-filtering algorithm computes the identical sums with 16 multiplications per 2x2 output tile and channel
// pair instead of 36 (2.25x fewer MFMA FLOPs), at the price of three HBM-bound transforms:
//     V[xi][t][c] = (B^T d B)[xi]      d: 4x4 input patch of tile t (xi = 4*i+j)          wino_input_kernel
//     U[xi][k][c] = (G g G^T)[xi]      g: 3x3 filter (already the correlation taps)        wino_filter_kernel
//     M[xi][t][k] = sum_c V[xi][t][c] * U[xi][k][c]      16 GEMMs, batched                 igemm forward kernel
//     y tile      = A^T M A (+ bias, + add)                                                wino_output_kernel
// fp32 throughout; the result differs from the direct kernel by rounding only (~1e-6 relative).
// The data gradient of a stride-1 pad-1 3x3 convolution is the same convolution of dy with the taps rotated by 180
// degrees and the channel roles swapped: wino_filter_kernel<true> writes U'[xi][c][k] from w[k][2-r][2-s][c].
#include "common.h"

int denet_gemm_batched_nt(const float* a, const float* w, float* out, int batch, int M, int Nc, int Kc, long stride_a,
                          long stride_w, long stride_out, hipStream_t stream);
int denet_gemm_batched_tune(const float* a, const float* w, float* out, int batch, int M, int Nc, int Kc, long stride_a,
                            long stride_w, long stride_out, hipStream_t stream);

int denet_wgrad_batched(const float* x, const float* dy, float* dw, float* workspace, size_t workspace_bytes, int batch,
                        int T, int Cc, int Kr, hipStream_t stream);
int denet_wgrad_batched_tune(const float* x, const float* dy, float* dw, float* workspace, size_t workspace_bytes, int batch,
                             int T, int Cc, int Kr, hipStream_t stream);

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *(const f32x4*)p; }

// one thread per (tile, 4 channels)
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, float* __restrict__ V, int N, int H,
                                                         int W, int C, int TH, int TW, long T) {
    const int c4n = C / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * c4n) return;
    const int c4 = (int)(idx % c4n);
    const long t = idx / c4n;
    const int tx = (int)(t % TW);
    const int ty = (int)((t / TW) % TH);
    const int n = (int)(t / ((long)TW * TH));
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int iy = 2 * ty - 1 + i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ix = 2 * tx - 1 + j;
            const bool ok = ((unsigned)iy < (unsigned)H) && ((unsigned)ix < (unsigned)W);
            d[i][j] = ok ? ld4(x + (((long)n * H + iy) * W + ix) * C + c4 * 4) : z;
        }
    }
    f32x4 tt[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {       // B^T d
        tt[0][j] = d[0][j] - d[2][j];
        tt[1][j] = d[1][j] + d[2][j];
        tt[2][j] = d[2][j] - d[1][j];
        tt[3][j] = d[1][j] - d[3][j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {       // (B^T d) B
        const f32x4 v0 = tt[i][0] - tt[i][2];
        const f32x4 v1 = tt[i][1] + tt[i][2];
        const f32x4 v2 = tt[i][2] - tt[i][1];
        const f32x4 v3 = tt[i][1] - tt[i][3];
        float* o = V + ((long)(4 * i) * T + t) * C + c4 * 4;
        *(f32x4*)(o) = v0;
        *(f32x4*)(o + T * C) = v1;
        *(f32x4*)(o + 2 * T * C) = v2;
        *(f32x4*)(o + 3 * T * C) = v3;
    }
}

// one thread per (k, c): U[xi][k][c] (DGRAD = false) or U'[xi][c][k] from the rotated taps (DGRAD = true)
template <bool DGRAD>
__global__ __launch_bounds__(256) void wino_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int K, int C) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)K * C) return;
    const int c = (int)(idx % C);
    const int k = (int)(idx / C);
    float g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) g[r][s] = w[(((long)k * 3 + (DGRAD ? 2 - r : r)) * 3 + (DGRAD ? 2 - s : s)) * C + c];
    float a[4][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {       // G g
        a[0][s] = g[0][s];
        a[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]);
        a[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]);
        a[3][s] = g[2][s];
    }
    const long KC = (long)K * C;
    const long o = DGRAD ? ((long)c * K + k) : ((long)k * C + c);
#pragma unroll
    for (int i = 0; i < 4; ++i) {       // (G g) G^T
        U[(4 * i + 0) * KC + o] = a[i][0];
        U[(4 * i + 1) * KC + o] = 0.5f * (a[i][0] + a[i][1] + a[i][2]);
        U[(4 * i + 2) * KC + o] = 0.5f * (a[i][0] - a[i][1] + a[i][2]);
        U[(4 * i + 3) * KC + o] = a[i][2];
    }
}

// one thread per (tile, 4 output channels): y tile = A^T M A (+ bias) (+ add)
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ Mx, const float* __restrict__ bias,
                                                          const float* __restrict__ add, float* __restrict__ y, int N,
                                                          int H, int W, int K, int TH, int TW, long T) {
    const int k4n = K / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * k4n) return;
    const int k4 = (int)(idx % k4n);
    const long t = idx / k4n;
    const int tx = (int)(t % TW);
    const int ty = (int)((t / TW) % TH);
    const int n = (int)(t / ((long)TW * TH));
    f32x4 m[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) m[i][j] = ld4(Mx + ((long)(4 * i + j) * T + t) * K + k4 * 4);
    f32x4 s[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {       // A^T m
        s[0][j] = m[0][j] + m[1][j] + m[2][j];
        s[1][j] = m[1][j] - m[2][j] - m[3][j];
    }
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (bias) b = ld4(bias + k4 * 4);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const f32x4 y0 = s[i][0] + s[i][1] + s[i][2] + b;
        const f32x4 y1 = s[i][1] - s[i][2] - s[i][3] + b;
        const long o = (((long)n * H + 2 * ty + i) * W + 2 * tx) * K + k4 * 4;
        if (add) {
            *(f32x4*)(y + o) = y0 + ld4(add + o);
            *(f32x4*)(y + o + K) = y1 + ld4(add + o + K);
        } else {
            *(f32x4*)(y + o) = y0;
            *(f32x4*)(y + o + K) = y1;
        }
    }
}

// filter gradient, step 1: dM[xi][t][k] = (A dy_tile A^T)[xi], the adjoint of the output transform
__global__ __launch_bounds__(256) void wino_dout_kernel(const float* __restrict__ dy, float* __restrict__ dM, int N, int H,
                                                        int W, int K, int TH, int TW, long T) {
    const int k4n = K / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * k4n) return;
    const int k4 = (int)(idx % k4n);
    const long t = idx / k4n;
    const int tx = (int)(t % TW);
    const int ty = (int)((t / TW) % TH);
    const int n = (int)(t / ((long)TW * TH));
    const long o = (((long)n * H + 2 * ty) * W + 2 * tx) * K + k4 * 4;
    const f32x4 y00 = ld4(dy + o), y01 = ld4(dy + o + K);
    const f32x4 y10 = ld4(dy + o + (long)W * K), y11 = ld4(dy + o + (long)W * K + K);
    // A = [[1,0],[1,1],[1,-1],[0,-1]]:  rows of A dy
    const f32x4 a[4][2] = {{y00, y01}, {y00 + y10, y01 + y11}, {y00 - y10, y01 - y11}, {-y10, -y11}};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float* d = dM + ((long)(4 * i) * T + t) * K + k4 * 4;
        *(f32x4*)(d) = a[i][0];
        *(f32x4*)(d + T * K) = a[i][0] + a[i][1];
        *(f32x4*)(d + 2 * T * K) = a[i][0] - a[i][1];
        *(f32x4*)(d + 3 * T * K) = -a[i][1];
    }
}

// filter gradient, step 3: dw[k][r][s][c] = (G^T dU G)[r][s], the adjoint of the filter transform
__global__ __launch_bounds__(256) void wino_dfilter_kernel(const float* __restrict__ dU, float* __restrict__ dw, int K, int C) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)K * C) return;
    const int c = (int)(idx % C);
    const int k = (int)(idx / C);
    const long KC = (long)K * C;
    float u[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) u[i][j] = dU[(4 * i + j) * KC + (long)k * C + c];
    float r[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {       // G^T u
        r[0][j] = u[0][j] + 0.5f * (u[1][j] + u[2][j]);
        r[1][j] = 0.5f * (u[1][j] - u[2][j]);
        r[2][j] = 0.5f * (u[1][j] + u[2][j]) + u[3][j];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {       // (G^T u) G
        float* o = dw + (((long)k * 3 + i) * 3) * C + c;
        o[0] = r[i][0] + 0.5f * (r[i][1] + r[i][2]);
        o[C] = 0.5f * (r[i][1] - r[i][2]);
        o[2 * C] = 0.5f * (r[i][1] + r[i][2]) + r[i][3];
    }
}

int wino_run(bool dgrad, const float* in, const float* w, const float* bias, const float* add, float* out, float* ws,
             size_t ws_bytes, int N, int H, int W, int Cin, int Cout, hipStream_t stream) {
    // in: [N,H,W,Cin]   out: [N,H,W,Cout]   w: KRSC with (K,C) = dgrad ? (Cin,Cout) : (Cout,Cin)
    DENET_CHECK_ARG(in && w && out && ws, "conv_wino: null pointer");
    DENET_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && Cin % 32 == 0 && Cout % 32 == 0, "conv_wino: H, W must be even and the channel counts multiples of 32");
    const int TH = H / 2, TW = W / 2;
    const long T = (long)N * TH * TW;
    const size_t nU = (size_t)16 * Cin * Cout, nV = (size_t)16 * T * Cin, nM = (size_t)16 * T * Cout;
    DENET_CHECK_ARG(ws_bytes >= (nU + nV + nM) * sizeof(float), "conv_wino: workspace too small (%zu < %zu)", ws_bytes,
                    (nU + nV + nM) * sizeof(float));
    DENET_CHECK_ARG(T < (1L << 31) / 16, "conv_wino: too many tiles");
    float* U = ws;
    float* V = U + nU;
    float* Mx = V + nV;
    const long kc = (long)Cin * Cout;
    if (dgrad)
        hipLaunchKernelGGL(wino_filter_kernel<true>, dim3((unsigned)((kc + 255) / 256)), dim3(256), 0, stream, w, U, Cin, Cout);
    else
        hipLaunchKernelGGL(wino_filter_kernel<false>, dim3((unsigned)((kc + 255) / 256)), dim3(256), 0, stream, w, U, Cout, Cin);
    const long ni = T * (Cin / 4);
    hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)((ni + 255) / 256)), dim3(256), 0, stream, in, V, N, H, W, Cin, TH,
                       TW, T);
    DENET_CHECK_LAUNCH("conv_wino transforms");
    int rc = denet_gemm_batched_nt(V, U, Mx, 16, (int)T, Cout, Cin, T * Cin, kc, T * Cout, stream);
    if (rc) return rc;
    const long no = T * (Cout / 4);
    hipLaunchKernelGGL(wino_output_kernel, dim3((unsigned)((no + 255) / 256)), dim3(256), 0, stream, Mx, bias, add, out, N, H,
                       W, Cout, TH, TW, T);
    DENET_CHECK_LAUNCH("conv_wino output");
    return DENET_OK;
}

}  // namespace

// dw = filter gradient of the 3x3 stride-1 pad-1 convolution; x:[N,H,W,C] dy:[N,H,W,K] dw:[K,3,3,C].
// workspace (denet_conv_wino_workspace_bytes): dU | V | dM; split_ws: the split-K slices of the batched product.
extern "C" int denet_conv_wino_wgrad(const float* x, const float* dy, float* dw, float* workspace, size_t workspace_bytes,
                                     float* split_ws, size_t split_ws_bytes, int N, int H, int W, int C, int K,
                                     hipStream_t stream) {
    DENET_CHECK_ARG(x && dy && dw && workspace, "conv_wino_wgrad: null pointer");
    DENET_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && C % 32 == 0 && K % 32 == 0, "conv_wino_wgrad: H, W must be even and the channel counts multiples of 32");
    const int TH = H / 2, TW = W / 2;
    const long T = (long)N * TH * TW;
    const size_t nU = (size_t)16 * C * K, nV = (size_t)16 * T * C, nM = (size_t)16 * T * K;
    DENET_CHECK_ARG(workspace_bytes >= (nU + nV + nM) * sizeof(float), "conv_wino_wgrad: workspace too small");
    DENET_CHECK_ARG(T < (1L << 31) / 16, "conv_wino_wgrad: too many tiles");
    float* dU = workspace;
    float* V = dU + nU;
    float* dM = V + nV;
    const long ni = T * (C / 4), no = T * (K / 4);
    hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)((ni + 255) / 256)), dim3(256), 0, stream, x, V, N, H, W, C, TH, TW, T);
    hipLaunchKernelGGL(wino_dout_kernel, dim3((unsigned)((no + 255) / 256)), dim3(256), 0, stream, dy, dM, N, H, W, K, TH, TW, T);
    DENET_CHECK_LAUNCH("conv_wino_wgrad transforms");
    int rc = denet_wgrad_batched(V, dM, dU, split_ws, split_ws_bytes, 16, (int)T, C, K, stream);
    if (rc) return rc;
    const long kc = (long)K * C;
    hipLaunchKernelGGL(wino_dfilter_kernel, dim3((unsigned)((kc + 255) / 256)), dim3(256), 0, stream, dU, dw, K, C);
    DENET_CHECK_LAUNCH("conv_wino_wgrad filter");
    return DENET_OK;
}

// measures the launch configuration of the component GEMMs of this geometry (both directions); synchronises
extern "C" int denet_conv_wino_tune(float* workspace, size_t workspace_bytes, float* split_ws, size_t split_ws_bytes, int N,
                                    int H, int W, int C, int K, hipStream_t stream) {
    DENET_CHECK_ARG(workspace && H % 2 == 0 && W % 2 == 0 && C % 32 == 0 && K % 32 == 0, "conv_wino_tune: bad arguments");
    const long T = (long)N * (H / 2) * (W / 2);
    const size_t nU = (size_t)16 * C * K, nV = (size_t)16 * T * C, nM = (size_t)16 * T * K;
    DENET_CHECK_ARG(workspace_bytes >= (nU + nV + nM) * sizeof(float), "conv_wino_tune: workspace too small");
    (void)hipMemsetAsync(workspace, 0, (nU + nV + nM) * sizeof(float), stream);
    float* U = workspace;
    // forward: V [T x C] -> M [T x K];  data gradient: V [T x K] -> M [T x C]  (the larger of V / M regions is reused)
    int rc = denet_gemm_batched_tune(U + nU, U, U + nU + nV, 16, (int)T, K, C, T * C, (long)C * K, T * K, stream);
    if (rc) return rc;
    rc = denet_gemm_batched_tune(U + nU + nV, U, U + nU, 16, (int)T, C, K, T * K, (long)C * K, T * C, stream);
    if (rc || !split_ws) return rc;
    return denet_wgrad_batched_tune(U + nU, U + nU + nV, U, split_ws, split_ws_bytes, 16, (int)T, C, K, stream);
}

extern "C" size_t denet_conv_wino_workspace_bytes(int N, int H, int W, int C, int K) {
    const size_t T = (size_t)N * (H / 2) * (W / 2);
    return ((size_t)16 * C * K + (size_t)16 * T * C + (size_t)16 * T * K) * sizeof(float);
}

// y = conv3x3(x, w) stride 1 pad 1 (+ bias) (+ add); x:[N,H,W,C] w:[K,3,3,C] y:[N,H,W,K]
extern "C" int denet_conv_wino_fwd(const float* x, const float* w, const float* bias, const float* add, float* y,
                                   float* workspace, size_t workspace_bytes, int N, int H, int W, int C, int K,
                                   hipStream_t stream) {
    return wino_run(false, x, w, bias, add, y, workspace, workspace_bytes, N, H, W, C, K, stream);
}

// dx = conv3x3_transposed(dy, w) (+ add); dy:[N,H,W,K] w:[K,3,3,C] dx:[N,H,W,C]
extern "C" int denet_conv_wino_dgrad(const float* dy, const float* w, const float* add, float* dx, float* workspace,
                                     size_t workspace_bytes, int N, int H, int W, int C, int K, hipStream_t stream) {
    return wino_run(true, dy, w, nullptr, add, dx, workspace, workspace_bytes, N, H, W, K, C, stream);
}
