// Kernels of the PixelCNN prior over VQ code grids (reference models/pixelcnn_prior.py):
// embedding gather (nn.Embedding forward), the gated residual combine, the in-place weight mask of MaskedConv2d and the
// categorical cross-entropy over the code axis.  All HBM-bound single passes; the convolutions of the prior run on the
// implicit-GEMM kernels of conv_igemm.hip.  (The embedding's gradient is a segmented sum over sorted codes: vq.hip.)
#include "common.h"

namespace {

inline int grid_for(long n) {
    long g = (n + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

// y[r][:] = w[idx[r]][:]   (d % 4 == 0: 16-byte rows chunks; else scalar)
__global__ void embedding_fwd_k(const float* __restrict__ w, const int64_t* __restrict__ idx, float* __restrict__ y, long rows, int d,
                                int k) {
    const long stride = (long)gridDim.x * blockDim.x;
    if ((d & 3) == 0) {
        const int dq = d >> 2;
        const long nv = rows * dq;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
            const long r = i / dq;
            const int c = (int)(i - r * dq);
            long code = idx[r];
            code = code < 0 ? 0 : (code >= k ? k - 1 : code);  // (the host checks the range in debug runs; never read outside w)
            reinterpret_cast<f32x4*>(y)[i] = reinterpret_cast<const f32x4*>(w + code * d)[c];
        }
    } else {
        const long n = rows * d;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
            const long r = i / d;
            long code = idx[r];
            code = code < 0 ? 0 : (code >= k ? k - 1 : code);
            y[i] = w[code * d + (i - r * d)];
        }
    }
}

// out = res + sigmoid-gate * tanh-feature, where `gate` and `feat` already hold the activated values
__global__ void gated_fwd_k(const float* __restrict__ res, const float* __restrict__ gate, const float* __restrict__ feat,
                            float* __restrict__ out, long n4) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f32x4 r = reinterpret_cast<const f32x4*>(res)[i], g = reinterpret_cast<const f32x4*>(gate)[i],
                    f = reinterpret_cast<const f32x4*>(feat)[i];
        reinterpret_cast<f32x4*>(out)[i] = r + g * f;
    }
}

__global__ void gated_bwd_k(const float* __restrict__ dout, const float* __restrict__ gate, const float* __restrict__ feat,
                            float* __restrict__ dgate, float* __restrict__ dfeat, long n4) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f32x4 d = reinterpret_cast<const f32x4*>(dout)[i], g = reinterpret_cast<const f32x4*>(gate)[i],
                    f = reinterpret_cast<const f32x4*>(feat)[i];
        reinterpret_cast<f32x4*>(dgate)[i] = d * f;
        reinterpret_cast<f32x4*>(dfeat)[i] = d * g;
    }
}

__global__ void mul_k(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = a[i] * b[i];
}

// One wave per row of logits[rows][k]: lse = log sum exp, nll = lse - logit[target]; block partial of nll in fp64.
__global__ __launch_bounds__(256) void xent_fwd_k(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                  float* __restrict__ lse_out, double* __restrict__ part, long rows, int k) {
    __shared__ double sh[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double acc = 0.0;
    for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
        const float* row = logits + r * k;
        float m = -INFINITY;
        for (int c = lane; c < k; c += 64) m = fmaxf(m, row[c]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float s = 0.f;
        for (int c = lane; c < k; c += 64) s += expf(row[c] - m);
        s = wave_sum(s);
        const float lse = m + logf(s);
        if (lane == 0) {
            lse_out[r] = lse;
            long t = target[r];
            t = t < 0 ? 0 : (t >= k ? k - 1 : t);
            acc += (double)(lse - row[t]);
        }
    }
    if (lane == 0) sh[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void xent_final_k(const double* __restrict__ part, int nblk, double factor, float* __restrict__ out) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) s += part[i];
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) out[0] = (float)(s * factor);
}

// dlogits[r][c] = g / rows * (softmax(r)[c] - [c == target[r]])
__global__ void xent_bwd_k(const float* __restrict__ logits, const int64_t* __restrict__ target, const float* __restrict__ lse,
                           const float* __restrict__ gs, float* __restrict__ dlogits, long rows, int k) {
    const float f = (gs ? gs[0] : 1.f) / (float)rows;
    const long n = rows * k, stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const long r = i / k;
        const int c = (int)(i - r * k);
        const float p = expf(logits[i] - lse[r]);
        long t = target[r];
        t = t < 0 ? 0 : (t >= k ? k - 1 : t);  // the same clamp as xent_fwd_k: the gradient is the gradient of the loss that was reported
        dlogits[i] = f * (p - (t == c ? 1.f : 0.f));
    }
}

}  // namespace

extern "C" {

int movae_embedding_fwd(const float* weight, const int64_t* idx, float* y, size_t rows, int k, int d, movae_stream_t stream) {
    MOVAE_CHECK_ARG(weight && idx && y && rows > 0 && k > 0 && d > 0, "movae_embedding_fwd: bad argument");
    hipLaunchKernelGGL(embedding_fwd_k, dim3(grid_for((long)rows * d / 4 + 1)), dim3(256), 0, (hipStream_t)stream, weight, idx, y,
                       (long)rows, d, k);
    MOVAE_CHECK_LAUNCH("embedding_fwd");
    return MOVAE_OK;
}

int movae_gated_residual_fwd(const float* res, const float* gate, const float* feat, float* out, size_t n, movae_stream_t stream) {
    MOVAE_CHECK_ARG(res && gate && feat && out && n > 0 && n % 4 == 0, "movae_gated_residual_fwd: bad argument (n %% 4 == 0)");
    hipLaunchKernelGGL(gated_fwd_k, dim3(grid_for((long)n / 4)), dim3(256), 0, (hipStream_t)stream, res, gate, feat, out, (long)n / 4);
    MOVAE_CHECK_LAUNCH("gated_residual_fwd");
    return MOVAE_OK;
}

int movae_gated_residual_bwd(const float* dout, const float* gate, const float* feat, float* dgate, float* dfeat, size_t n,
                             movae_stream_t stream) {
    MOVAE_CHECK_ARG(dout && gate && feat && dgate && dfeat && n > 0 && n % 4 == 0, "movae_gated_residual_bwd: bad argument");
    hipLaunchKernelGGL(gated_bwd_k, dim3(grid_for((long)n / 4)), dim3(256), 0, (hipStream_t)stream, dout, gate, feat, dgate, dfeat,
                       (long)n / 4);
    MOVAE_CHECK_LAUNCH("gated_residual_bwd");
    return MOVAE_OK;
}

int movae_mul(const float* a, const float* b, float* y, size_t n, movae_stream_t stream) {
    MOVAE_CHECK_ARG(a && b && y && n > 0, "movae_mul: bad argument");
    hipLaunchKernelGGL(mul_k, dim3(grid_for((long)n)), dim3(256), 0, (hipStream_t)stream, a, b, y, (long)n);
    MOVAE_CHECK_LAUNCH("mul");
    return MOVAE_OK;
}

size_t movae_cross_entropy_ws_bytes(size_t rows) {
    const long nb = (long)(rows + 3) / 4;
    return MOVAE_WS_HEADER_BYTES + (size_t)(nb > 2048 ? 2048 : nb) * sizeof(double);
}

int movae_cross_entropy_fwd(const float* logits, const int64_t* target, float* loss, float* lse, size_t rows, int k, void* ws,
                            size_t ws_bytes, movae_stream_t stream) {
    MOVAE_WS_SCRATCH(ws, ws_bytes);
    MOVAE_CHECK_ARG(logits && target && loss && lse && rows > 0 && k > 0, "movae_cross_entropy_fwd: bad argument");
    long nb = (long)(rows + 3) / 4;
    if (nb > 2048) nb = 2048;
    MOVAE_CHECK_ARG(ws && ws_bytes >= (size_t)nb * sizeof(double), "movae_cross_entropy_fwd: workspace too small");
    double* part = static_cast<double*>(ws);
    hipLaunchKernelGGL(xent_fwd_k, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, logits, target, lse, part, (long)rows, k);
    MOVAE_CHECK_LAUNCH("xent_fwd");
    hipLaunchKernelGGL(xent_final_k, dim3(1), dim3(256), 0, (hipStream_t)stream, part, (int)nb, 1.0 / (double)rows, loss);
    MOVAE_CHECK_LAUNCH("xent_final");
    return MOVAE_OK;
}

int movae_cross_entropy_bwd(const float* logits, const int64_t* target, const float* lse, const float* gscale_dev, float* dlogits,
                            size_t rows, int k, movae_stream_t stream) {
    MOVAE_CHECK_ARG(logits && target && lse && dlogits && rows > 0 && k > 0, "movae_cross_entropy_bwd: bad argument");
    hipLaunchKernelGGL(xent_bwd_k, dim3(grid_for((long)rows * k)), dim3(256), 0, (hipStream_t)stream, logits, target, lse, gscale_dev,
                       dlogits, (long)rows, k);
    MOVAE_CHECK_LAUNCH("xent_bwd");
    return MOVAE_OK;
}

}  // extern "C"
