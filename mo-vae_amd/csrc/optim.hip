// Optimizer tail over one flat fp32 arena (parameters / gradients / moments contiguous):
// torch.optim.Adam / AdamW arithmetic (main.py:1169-1178, step at main.py:214) and the
// clip_grad_norm_ pieces (main.py:211-212).  One HBM pass: 4 reads + 3 writes per element.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void adam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                              float wd, int decoupled, float bc1, float bc2_sqrt) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float gi = g[i], pi = p[i];
        if (wd != 0.f) {
            if (decoupled) pi *= 1.f - lr * wd;
            else gi += wd * pi;
        }
        // torch: exp_avg.lerp_(grad, 1 - beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        const float mi = m[i] + (gi - m[i]) * (1.f - b1);
        const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}

__global__ __launch_bounds__(256) void sumsq_partial(const float* __restrict__ x, long n, double* __restrict__ part) {
    __shared__ double sh[4];
    double s = 0.0;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) s += (double)x[i] * x[i];
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void sumsq_final(const double* __restrict__ part, int nblk, float* __restrict__ out) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) s += part[i];
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) out[0] = (float)s;
}

__global__ void clip_scale_k(float* __restrict__ g, long n, const float* __restrict__ sumsq, float max_norm) {
    // clip_coef = max_norm / (total_norm + 1e-6), clamped to 1 (torch.nn.utils.clip_grad_norm_)
    const float coef = fminf(max_norm / (sqrtf(sumsq[0]) + 1e-6f), 1.f);
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) g[i] *= coef;
}

inline int blocks_for(size_t n) {
    size_t b = (n + 1023) / 1024;
    return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" {

int movae_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int decoupled_wd, int step, movae_stream_t stream) {
    MOVAE_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "movae_adam_step: bad argument");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_k, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, lr, beta1, beta2, eps,
                       weight_decay, decoupled_wd, (float)bc1, (float)sqrt(bc2));
    MOVAE_CHECK_LAUNCH("adam");
    return MOVAE_OK;
}

int movae_sumsq(const float* x, size_t n, float* out, void* ws, size_t ws_bytes, movae_stream_t stream) {
    MOVAE_CHECK_ARG(x && out && n > 0, "movae_sumsq: bad argument");
    const int nb = blocks_for(n);
    MOVAE_CHECK_ARG(ws && ws_bytes >= (size_t)nb * sizeof(double), "movae_sumsq: workspace too small");
    double* part = static_cast<double*>(ws);
    hipLaunchKernelGGL(sumsq_partial, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, (long)n, part);
    MOVAE_CHECK_LAUNCH("sumsq_partial");
    hipLaunchKernelGGL(sumsq_final, dim3(1), dim3(256), 0, (hipStream_t)stream, part, nb, out);
    MOVAE_CHECK_LAUNCH("sumsq_final");
    return MOVAE_OK;
}

int movae_scale_by_clip(float* g, size_t n, const float* sumsq_dev, float max_norm, movae_stream_t stream) {
    MOVAE_CHECK_ARG(g && sumsq_dev && n > 0 && max_norm > 0.f, "movae_scale_by_clip: bad argument");
    hipLaunchKernelGGL(clip_scale_k, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, g, (long)n, sumsq_dev, max_norm);
    MOVAE_CHECK_LAUNCH("clip_scale");
    return MOVAE_OK;
}

}  // extern "C"
