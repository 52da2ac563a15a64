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

// ---- multi-tensor Adam: ONE launch for a whole parameter list (the step's ~50 tensors range from 32 to 1.2M elements; a
// launch per tensor -- or torch's capturable foreach path, ~110 launches -- costs more than the arithmetic) ----------
constexpr int ADAM_MT = 64;  // tensors per launch: 64 * (4 pointers + count) = 2.3 KB of kernel arguments
struct AdamTable {
    float* p[ADAM_MT];
    const float* g[ADAM_MT];
    float* m[ADAM_MT];
    float* v[ADAM_MT];
    unsigned n[ADAM_MT];
    unsigned first[ADAM_MT + 1];  // first block of tensor t in the flat grid: a block owns ADAM_BLK consecutive elements of ONE tensor
};
constexpr unsigned ADAM_BLK = 4096;  // four 16-byte pieces per thread and array: sixteen loads in flight per thread

__device__ __forceinline__ void adam_one(float& pi, float gi, float& mi, float& vi, float lr, float b1, float b2, float eps,
                                         float wd, int decoupled, float step_size, float bc2_sqrt) {
    if (wd != 0.f) {
        if (decoupled) pi *= 1.f - lr * wd;
        else gi += wd * pi;
    }
    mi = mi + (gi - mi) * (1.f - b1);
    vi = vi * b2 + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi = pi - step_size * (mi / denom);
}

// A flat grid of ceil(n_t / ADAM_BLK) blocks per tensor (a 2-D grid of <= 256 blocks x tensors left the two 4-8 M element tensors of
// C5 to 256 blocks -- one per CU, 3.4 TB/s -- and launched mostly empty blocks for the small ones).  hyper (optional) = device
// {step, lr}: the step counter has already been incremented by adam_tick on the same stream, so a captured hipGraph replays with a
// live counter / learning rate.
__device__ unsigned g_adam_done = 0;  // blocks of the running adam_multi_k launch that have read the step counter and finished

__global__ __launch_bounds__(256) void adam_multi_k(AdamTable tab, int cnt, float lr, float b1, float b2, float eps, float wd,
                                                    int decoupled, float bc1, float bc2_sqrt, float* __restrict__ hyper, int tick) {
    int t = 0;
    while (t + 1 < cnt && blockIdx.x >= tab.first[t + 1]) ++t;  // (uniform: <= 64 scalar compares)
    const unsigned n = tab.n[t], base = (blockIdx.x - tab.first[t]) * ADAM_BLK;
    float step_f = 0.f;
    if (hyper) {
        // the counter holds the number of steps made so far; this one is step + 1.  It is advanced by the block that finishes LAST
        // (below) -- every block has read it by then -- instead of by a one-thread launch in front of this one
        step_f = hyper[0] + 1.f;
        const double step = (double)step_f;
        lr = hyper[1];
        bc1 = (float)(1.0 - pow((double)b1, step));
        bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, step));
    }
    const float step_size = lr / bc1;
    float* __restrict__ p = tab.p[t];
    const float* __restrict__ g = tab.g[t];
    float* __restrict__ m = tab.m[t];
    float* __restrict__ v = tab.v[t];
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                       reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    if (vec && base + ADAM_BLK <= n) {  // a whole block of 16-byte pieces: every load ahead of the arithmetic
        f32x4 pv[4], mv[4], vv[4], gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned i = base + u * 1024u + threadIdx.x * 4;
            pv[u] = *reinterpret_cast<f32x4*>(p + i), mv[u] = *reinterpret_cast<f32x4*>(m + i), vv[u] = *reinterpret_cast<f32x4*>(v + i);
            gv[u] = *reinterpret_cast<const f32x4*>(g + i);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned i = base + u * 1024u + threadIdx.x * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float pj = pv[u][j], mj = mv[u][j], vj = vv[u][j];
                adam_one(pj, gv[u][j], mj, vj, lr, b1, b2, eps, wd, decoupled, step_size, bc2_sqrt);
                pv[u][j] = pj; mv[u][j] = mj; vv[u][j] = vj;
            }
            *reinterpret_cast<f32x4*>(p + i) = pv[u];
            *reinterpret_cast<f32x4*>(m + i) = mv[u];
            *reinterpret_cast<f32x4*>(v + i) = vv[u];
        }
    } else
    for (unsigned u = 0; u < 4; ++u) {
        const unsigned i = base + u * 1024u + threadIdx.x * 4;
        if (i >= n) break;
        if (vec && i + 4 <= n) {
            f32x4 pv = *reinterpret_cast<f32x4*>(p + i), mv = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
            const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float pj = pv[j], mj = mv[j], vj = vv[j];
                adam_one(pj, gv[j], mj, vj, lr, b1, b2, eps, wd, decoupled, step_size, bc2_sqrt);
                pv[j] = pj; mv[j] = mj; vv[j] = vj;
            }
            *reinterpret_cast<f32x4*>(p + i) = pv;
            *reinterpret_cast<f32x4*>(m + i) = mv;
            *reinterpret_cast<f32x4*>(v + i) = vv;
        } else {
            const unsigned e = min(i + 4, n);
            for (unsigned k = i; k < e; ++k) {
                float pj = p[k], mj = m[k], vj = v[k];
                adam_one(pj, g[k], mj, vj, lr, b1, b2, eps, wd, decoupled, step_size, bc2_sqrt);
                p[k] = pj; m[k] = mj; v[k] = vj;
            }
        }
    }
    if (hyper) {  // (every path of this kernel falls through to here)
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned done = atomicAdd(&g_adam_done, 1u);
            if (done == gridDim.x - 1) {
                g_adam_done = 0;
                if (tick) hyper[0] = step_f;
            }
        }
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

// ---- clip_grad_norm_ over a tensor list: 3 launches (partials, final, scale) whatever the list length ------------------
struct ClipTable {
    float* g[ADAM_MT];
    unsigned n[ADAM_MT];
};

// grid = (blocks per tensor, tensors); part[tensor * gridDim.x + block] = sum of squares of that block's slice (fp64)
__global__ __launch_bounds__(256) void sumsq_multi_k(ClipTable tab, double* __restrict__ part) {
    __shared__ double sh[4];
    const float* __restrict__ g = tab.g[blockIdx.y];
    const unsigned n = tab.n[blockIdx.y];
    double s = 0.0;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) s += (double)g[i] * g[i];
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) part[blockIdx.y * gridDim.x + blockIdx.x] = s;
}

// sumsq_acc (+)= sum of the partials; then, once every chunk of the list has been added, norm = sqrt
__global__ __launch_bounds__(256) void sumsq_multi_final(const double* __restrict__ part, int nparts, float* __restrict__ sumsq,
                                                         int accumulate) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) sumsq[0] = accumulate ? sumsq[0] + (float)s : (float)s;
}

__global__ __launch_bounds__(256) void clip_multi_k(ClipTable tab, const float* __restrict__ sumsq, float max_norm) {
    const float coef = fminf(max_norm / (sqrtf(sumsq[0]) + 1e-6f), 1.f);  // torch.nn.utils.clip_grad_norm_
    if (coef >= 1.f) return;                                                // torch multiplies by 1.0: same values
    float* __restrict__ g = tab.g[blockIdx.y];
    const unsigned n = tab.n[blockIdx.y];
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) g[i] *= coef;
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

int movae_adam_multi(int n_tensors, float* const* p, const float* const* g, float* const* m, float* const* v, const size_t* numel,
                     float lr, float beta1, float beta2, float eps, float weight_decay, int decoupled_wd, int step,
                     float* hyper_dev, movae_stream_t stream) {
    MOVAE_CHECK_ARG(n_tensors >= 0 && (n_tensors == 0 || (p && g && m && v && numel)), "movae_adam_multi: null table");
    MOVAE_CHECK_ARG(hyper_dev || step >= 1, "movae_adam_multi: step must be >= 1 when no device counter is given");
    float bc1 = 1.f, bc2_sqrt = 1.f;
    if (!hyper_dev) {
        bc1 = (float)(1.0 - pow((double)beta1, step));
        bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, step));
    }
    for (int t0 = 0; t0 < n_tensors; t0 += ADAM_MT) {
        AdamTable tab;
        const int cnt = n_tensors - t0 < ADAM_MT ? n_tensors - t0 : ADAM_MT;
        unsigned nblk = 0;
        for (int i = 0; i < cnt; ++i) {
            MOVAE_CHECK_ARG(p[t0 + i] && g[t0 + i] && m[t0 + i] && v[t0 + i], "movae_adam_multi: null tensor");
            MOVAE_CHECK_ARG(numel[t0 + i] > 0 && numel[t0 + i] < 0xffffffffUL, "movae_adam_multi: tensor size out of range");
            tab.p[i] = p[t0 + i]; tab.g[i] = g[t0 + i]; tab.m[i] = m[t0 + i]; tab.v[i] = v[t0 + i];
            tab.n[i] = (unsigned)numel[t0 + i];
            tab.first[i] = nblk;
            nblk += (unsigned)((numel[t0 + i] + ADAM_BLK - 1) / ADAM_BLK);
        }
        for (int i = cnt; i < ADAM_MT; ++i) { tab.p[i] = nullptr; tab.g[i] = nullptr; tab.m[i] = nullptr; tab.v[i] = nullptr; tab.n[i] = 0; }
        for (int i = cnt; i <= ADAM_MT; ++i) tab.first[i] = nblk;
        hipLaunchKernelGGL(adam_multi_k, dim3(nblk), dim3(256), 0, (hipStream_t)stream, tab, cnt, lr, beta1, beta2, eps, weight_decay,
                           decoupled_wd, bc1, bc2_sqrt, hyper_dev, t0 + ADAM_MT >= n_tensors ? 1 : 0);
        MOVAE_CHECK_LAUNCH("adam_multi");
    }
    return MOVAE_OK;
}

int movae_clip_grad_norm_multi(int n_tensors, float* const* g, const size_t* numel, float max_norm, float* total_sumsq_dev, void* ws,
                               size_t ws_bytes, movae_stream_t stream) {
    MOVAE_WS_SCRATCH(ws, ws_bytes);
    MOVAE_CHECK_ARG(n_tensors >= 1 && g && numel && total_sumsq_dev && max_norm > 0.f, "movae_clip_grad_norm_multi: bad argument");
    constexpr int BX = 32;  // blocks per tensor
    MOVAE_CHECK_ARG(ws && ws_bytes >= (size_t)ADAM_MT * BX * sizeof(double), "movae_clip_grad_norm_multi: workspace too small");
    double* part = static_cast<double*>(ws);
    hipStream_t st = (hipStream_t)stream;
    for (int pass = 0; pass < 2; ++pass)  // pass 0: total norm over all chunks; pass 1: scale
        for (int t0 = 0; t0 < n_tensors; t0 += ADAM_MT) {
            ClipTable tab;
            const int cnt = n_tensors - t0 < ADAM_MT ? n_tensors - t0 : ADAM_MT;
            for (int i = 0; i < ADAM_MT; ++i) {
                tab.g[i] = i < cnt ? g[t0 + i] : nullptr;
                tab.n[i] = i < cnt ? (unsigned)numel[t0 + i] : 0;
                if (i < cnt) MOVAE_CHECK_ARG(g[t0 + i] && numel[t0 + i] < 0xffffffffUL, "movae_clip_grad_norm_multi: bad tensor");
            }
            if (pass == 0) {
                hipLaunchKernelGGL(sumsq_multi_k, dim3(BX, cnt), dim3(256), 0, st, tab, part);
                MOVAE_CHECK_LAUNCH("sumsq_multi");
                hipLaunchKernelGGL(sumsq_multi_final, dim3(1), dim3(256), 0, st, part, BX * cnt, total_sumsq_dev, t0 > 0 ? 1 : 0);
                MOVAE_CHECK_LAUNCH("sumsq_multi_final");
            } else {
                hipLaunchKernelGGL(clip_multi_k, dim3(BX, cnt), dim3(256), 0, st, tab, total_sumsq_dev, max_norm);
                MOVAE_CHECK_LAUNCH("clip_multi");
            }
        }
    return MOVAE_OK;
}

int movae_sumsq(const float* x, size_t n, float* out, void* ws, size_t ws_bytes, movae_stream_t stream) {
    MOVAE_WS_SCRATCH(ws, ws_bytes);
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
