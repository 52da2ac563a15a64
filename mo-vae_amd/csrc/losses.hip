// Loss terms of the ELBO family: reconstruction objectives, KL to N(0, I), and the Beta-TC
// decomposition (mutual information / total correlation / dimension-wise KL).
// Reductions: per-thread fp32 term -> fp64 wave shuffle reduction -> fp64 block partial ->
// single-block final (deterministic).
#include "common.h"

namespace {

constexpr int RED_BLOCKS = 1024;

inline int red_blocks(size_t n) {
    size_t g = (n + 1023) / 1024;
    return (int)(g > RED_BLOCKS ? RED_BLOCKS : (g < 1 ? 1 : g));
}

__device__ __forceinline__ float recon_term(float r, float x, int kind) {
    switch (kind) {
        case MOVAE_RECON_BCE: {
            const float l1 = fmaxf(logf(r), -100.f), l2 = fmaxf(logf(1.f - r), -100.f);
            return -(x * l1 + (1.f - x) * l2);
        }
        case MOVAE_RECON_L1: return fabsf(r - x);
        case MOVAE_RECON_SMOOTH_L1: {
            const float d = fabsf(r - x);
            return d < 1.f ? 0.5f * d * d : d - 0.5f;
        }
        default: {
            const float d = r - x;
            return d * d;
        }
    }
}

__device__ __forceinline__ float recon_dterm(float r, float x, int kind) {
    switch (kind) {
        case MOVAE_RECON_BCE: return (r - x) / fmaxf((1.f - r) * r, 1e-12f);  // ATen binary_cross_entropy_backward
        case MOVAE_RECON_L1: {
            const float d = r - x;
            return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        }
        case MOVAE_RECON_SMOOTH_L1: {
            const float d = r - x;
            return fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f);
        }
        default: return 2.f * (r - x);
    }
}

__global__ __launch_bounds__(256) void recon_partial(const float* __restrict__ r, const float* __restrict__ x,
                                                     double* __restrict__ part, long n, int kind) {
    __shared__ double sh[4];
    double s = 0.0;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) s += recon_term(r[i], x[i], kind);
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// out[0] = factor * sum(part)
__global__ __launch_bounds__(256) void final_sum(const double* __restrict__ part, int nblk, double factor,
                                                 float* __restrict__ out) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) s += part[i];
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) out[0] = (float)(s * factor);
}

// recon_partial on blocks [0, nb1), kl_partial on blocks [nb1, nb1 + nb2): the two partial passes of VAE.loss_function in one
// launch, each block doing exactly what its stand-alone kernel's block does (same strides, same partials)
__global__ __launch_bounds__(256) void recon_kl_partial(const float* __restrict__ r, const float* __restrict__ x, long n, int kind,
                                                        const float* __restrict__ mu, const float* __restrict__ lv, long nk, int nb1,
                                                        int nb2, double* __restrict__ part) {
    __shared__ double sh[4];
    double s = 0.0;
    if ((int)blockIdx.x < nb1) {
        const long stride = (long)nb1 * 256;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) s += recon_term(r[i], x[i], kind);
    } else {
        const long stride = (long)nb2 * 256;
        for (long i = (long)((int)blockIdx.x - nb1) * 256 + threadIdx.x; i < nk; i += stride) {
            const float m = mu[i], l = lv[i];
            s += 1.f + l - m * m - expf(l);
        }
    }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// two loss terms and their sum in one launch: out[0] = f1 * sum(part[0 .. n1)), out[1] = f2 * sum(part[n1 .. n1 + n2)),
// out[2] = out[0] + out[1] in fp32 (the total_loss of models/vae.py:226: a tensor add of the two fp32 scalars)
__global__ __launch_bounds__(256) void final_sum2(const double* __restrict__ part, int n1, int n2, double f1, double f2,
                                                  float* __restrict__ out) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n1; i += 256) s += part[i];
    s = block_sum_256(s, sh);
    const float a = (float)(s * f1);
    double q = 0.0;
    for (int i = threadIdx.x; i < n2; i += 256) q += part[n1 + i];
    q = block_sum_256(q, sh);
    const float b = (float)(q * f2);
    if (threadIdx.x == 0) out[0] = a, out[1] = b, out[2] = a + b;
}

__global__ void recon_bwd_k(const float* __restrict__ r, const float* __restrict__ x, const float* __restrict__ gs,
                            float* __restrict__ dr, long n, int kind, float factor) {
    const float f = factor * (gs ? gs[0] : 1.f);
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dr[i] = f * recon_dterm(r[i], x[i], kind);
}

// ... times act'(pre) where recons = act(pre) is the output of the decoder's last activation (tanh / sigmoid: the derivative is a
// function of the output): the gradient leaves this kernel as the PRE-activation gradient, no activation-backward pass
__global__ void recon_bwd_act_k(const float* __restrict__ r, const float* __restrict__ x, const float* __restrict__ gs,
                                float* __restrict__ dr, long n, int kind, float factor, int act, float slope) {
    const float f = factor * (gs ? gs[0] : 1.f);
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dr[i] = (f * recon_dterm(r[i], x[i], kind)) * act_grad_from_out(r[i], act, slope);
}

__global__ __launch_bounds__(256) void kl_partial(const float* __restrict__ mu, const float* __restrict__ lv,
                                                  double* __restrict__ part, long n) {
    __shared__ double sh[4];
    double s = 0.0;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float m = mu[i], l = lv[i];
        s += 1.f + l - m * m - expf(l);
    }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ void kl_bwd_k(const float* __restrict__ mu, const float* __restrict__ lv, const float* __restrict__ gs,
                         float* __restrict__ dmu, float* __restrict__ dlv, long n, float factor) {
    const float f = factor * (gs ? gs[0] : 1.f);
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        dmu[i] = f * mu[i];
        dlv[i] = f * 0.5f * (expf(lv[i]) - 1.f);
    }
}

// ------------------------------------------------------------------------------------------------
// Beta-TC decomposition.  One block per sample i; wave w sweeps j = w, w+4, ...; lanes sweep d.
// ------------------------------------------------------------------------------------------------
constexpr int TC_DU = 8;  // D <= 64 * TC_DU
constexpr float LOG_2PI = 1.8378770664093453f;

struct Lse {
    float m, s;
    __device__ void init() { m = -INFINITY; s = 0.f; }
    __device__ void add(float v) {
        if (v > m) {
            s = s * expf(m - v) + 1.f;
            m = v;
        } else {
            s += expf(v - m);
        }
    }
    __device__ void merge(float m2, float s2) {
        if (m2 == -INFINITY) return;
        if (m2 > m) {
            s = s * expf(m - m2) + s2;
            m = m2;
        } else {
            s += s2 * expf(m2 - m);
        }
    }
    __device__ float value() const { return m + logf(s); }
};

__global__ __launch_bounds__(256) void tc_fwd_k(const float* __restrict__ z, const float* __restrict__ mu,
                                                const float* __restrict__ lv, const float* __restrict__ liw,
                                                float* __restrict__ lse_joint /* [B + B*B] */,
                                                float* __restrict__ lse_marg /* [B][D] */, float* __restrict__ rows /* [B][3] */,
                                                int B, int D) {
    extern __shared__ float smem[];  // 4 waves x (2 + 2*D) floats
    const int i = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float zi[TC_DU];
#pragma unroll
    for (int u = 0; u < TC_DU; ++u) {
        const int d = lane + 64 * u;
        zi[u] = d < D ? z[(long)i * D + d] : 0.f;
    }
    Lse joint;
    joint.init();
    Lse marg[TC_DU];
#pragma unroll
    for (int u = 0; u < TC_DU; ++u) marg[u].init();
    float own = 0.f;  // sum_d m_iid, accumulated by the wave that owns j == i
    for (int j = w; j < B; j += 4) {
        const float L = liw[(long)i * B + j];
        float part = 0.f;
#pragma unroll
        for (int u = 0; u < TC_DU; ++u) {
            const int d = lane + 64 * u;
            if (d < D) {
                const float l = lv[(long)j * D + d];
                const float df = zi[u] - mu[(long)j * D + d];
                const float m = -0.5f * (LOG_2PI + l) - 0.5f * (df * df * expf(-l));
                part += m;
                marg[u].add(m + L);
            }
        }
        const float sum = wave_sum(part);
        if (j == i) own = sum;
        const float a = sum + (float)D * L;
        joint.add(a);
        if (lane == 0) lse_joint[B + (long)i * B + j] = a;
    }
    // merge the four waves
    float* sj = smem;                 // [4][2]
    float* sm = smem + 8;             // [4][D][2]
    float* so = smem + 8 + 8 * D;     // [4]
    if (lane == 0) {
        sj[2 * w] = joint.m;
        sj[2 * w + 1] = joint.s;
        so[w] = own;
    }
#pragma unroll
    for (int u = 0; u < TC_DU; ++u) {
        const int d = lane + 64 * u;
        if (d < D) {
            sm[((long)w * D + d) * 2] = marg[u].m;
            sm[((long)w * D + d) * 2 + 1] = marg[u].s;
        }
    }
    __syncthreads();
    if (w == 0) {
        Lse J;
        J.init();
        for (int q = 0; q < 4; ++q) J.merge(sj[2 * q], sj[2 * q + 1]);
        const float lqz = J.value();
        float lp = 0.f, lpz = 0.f;
#pragma unroll
        for (int u = 0; u < TC_DU; ++u) {
            const int d = lane + 64 * u;
            if (d < D) {
                Lse Mg;
                Mg.init();
                for (int q = 0; q < 4; ++q) Mg.merge(sm[((long)q * D + d) * 2], sm[((long)q * D + d) * 2 + 1]);
                const float v = Mg.value();
                lse_marg[(long)i * D + d] = v;
                lp += v;
                lpz += -0.5f * (LOG_2PI + zi[u] * zi[u]);
            }
        }
        lp = wave_sum(lp);
        lpz = wave_sum(lpz);
        if (lane == 0) {
            const float lqzx = so[0] + so[1] + so[2] + so[3];
            lse_joint[i] = lqz;
            rows[i * 3 + 0] = lqzx - lqz;
            rows[i * 3 + 1] = lqz - lp;
            rows[i * 3 + 2] = lp - lpz;
        }
    }
}

__global__ void tc_mean_k(const float* __restrict__ rows, float* __restrict__ out, int B) {  // one wave
    // (three threads walking the B rows one dependent load at a time took ~10 us of L2 round trips for 96 numbers)
    const int lane = threadIdx.x;
    double s[3] = {0.0, 0.0, 0.0};
    for (int i = lane; i < B; i += 64)
#pragma unroll
        for (int q = 0; q < 3; ++q) s[q] += (double)rows[i * 3 + q];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s[q] += __shfl_xor(s[q], off, 64);
        if (lane == 0) out[q] = (float)(s[q] / B);
    }
}

// dz[i][d] : block per i; thread (d, jl): the B rows of the inner sum are dealt to JL = blockDim.x / DT row lanes (a thread per d
// alone walks B x ~80 dependent VALU instructions with 64 waves on the whole chip: 8 us for 0.5 MFLOP), folded through LDS in
// row-lane order.  DT = threads along d (host: a power of two <= 256).
__global__ void tc_bwd_dz(const float* __restrict__ z, const float* __restrict__ mu, const float* __restrict__ lv,
                          const float* __restrict__ liw, const float* __restrict__ lse_joint,
                          const float* __restrict__ lse_marg, const float* __restrict__ g, float* __restrict__ dz, int B,
                          int D, int DT) {
    extern __shared__ float red[];  // [JL][DT]
    const int i = blockIdx.x, dl = threadIdx.x % DT, jl = threadIdx.x / DT, JL = blockDim.x / DT;
    const float gmi = g[0] / B, cq = (g[1] - g[0]) / B, cp = (g[2] - g[1]) / B, gk = g[2] / B;
    const float lqz = lse_joint[i];
    for (int d0 = 0; d0 < D; d0 += DT) {
        const int d = d0 + dl;
        float acc = 0.f;
        if (d < D) {
            const float zi = z[(long)i * D + d], lm = lse_marg[(long)i * D + d];
            for (int j = jl; j < B; j += JL) {
                const float l = lv[(long)j * D + d];
                const float iv = expf(-l), df = zi - mu[(long)j * D + d];
                const float m = -0.5f * (LOG_2PI + l) - 0.5f * (df * df * iv);
                const float P = expf(lse_joint[B + (long)i * B + j] - lqz);
                const float Q = expf(m + liw[(long)i * B + j] - lm);
                const float wgt = cq * P + cp * Q + (j == i ? gmi : 0.f);
                acc += wgt * (-df * iv);
            }
        }
        red[jl * DT + dl] = acc;
        __syncthreads();
        if (jl == 0 && d < D) {
            float s = 0.f;
            for (int q = 0; q < JL; ++q) s += red[q * DT + dl];
            dz[(long)i * D + d] = s + gk * z[(long)i * D + d];
        }
        __syncthreads();
    }
}

// dmu[j][d], dlv[j][d] : block per j; thread (d, il) as above
__global__ void tc_bwd_dparams(const float* __restrict__ z, const float* __restrict__ mu, const float* __restrict__ lv,
                               const float* __restrict__ liw, const float* __restrict__ lse_joint,
                               const float* __restrict__ lse_marg, const float* __restrict__ g, float* __restrict__ dmu,
                               float* __restrict__ dlv, int B, int D, int DT) {
    extern __shared__ float red[];  // [2][IL][DT]
    const int j = blockIdx.x, dl = threadIdx.x % DT, il = threadIdx.x / DT, IL = blockDim.x / DT;
    const float gmi = g[0] / B, cq = (g[1] - g[0]) / B, cp = (g[2] - g[1]) / B;
    for (int d0 = 0; d0 < D; d0 += DT) {
        const int d = d0 + dl;
        float am = 0.f, al = 0.f;
        if (d < D) {
            const float l = lv[(long)j * D + d], mj = mu[(long)j * D + d];
            const float iv = expf(-l);
            for (int i = il; i < B; i += IL) {
                const float df = z[(long)i * D + d] - mj;
                const float m = -0.5f * (LOG_2PI + l) - 0.5f * (df * df * iv);
                const float P = expf(lse_joint[B + (long)i * B + j] - lse_joint[i]);
                const float Q = expf(m + liw[(long)i * B + j] - lse_marg[(long)i * D + d]);
                const float wgt = cq * P + cp * Q + (j == i ? gmi : 0.f);
                am += wgt * (df * iv);
                al += wgt * (-0.5f + 0.5f * df * df * iv);
            }
        }
        red[il * DT + dl] = am;
        red[(IL + il) * DT + dl] = al;
        __syncthreads();
        if (il == 0 && d < D) {
            float s1 = 0.f, s2 = 0.f;
            for (int q = 0; q < IL; ++q) s1 += red[q * DT + dl], s2 += red[(IL + q) * DT + dl];
            dmu[(long)j * D + d] = s1;
            dlv[(long)j * D + d] = s2;
        }
        __syncthreads();
    }
}

inline int grid_for(long total) {
    long gq = (total + 255) / 256;
    return (int)(gq > 4096 ? 4096 : (gq < 1 ? 1 : gq));
}

}  // namespace

extern "C" {

size_t movae_reduce_ws_bytes(size_t n) { return MOVAE_WS_HEADER_BYTES + (size_t)red_blocks(n) * sizeof(double); }

int movae_recon_loss_fwd(const float* recons, const float* inputs, float* out, size_t n, int kind, float scale, void* ws,
                         size_t ws_bytes, movae_stream_t stream) {
    MOVAE_WS_SCRATCH(ws, ws_bytes);
    MOVAE_CHECK_ARG(recons && inputs && out && n > 0, "movae_recon_loss_fwd: bad argument");
    MOVAE_CHECK_ARG(kind >= 0 && kind <= 3, "movae_recon_loss_fwd: unknown objective %d", kind);
    MOVAE_CHECK_ARG(ws && ws_bytes >= movae_reduce_ws_bytes(n), "movae_recon_loss_fwd: workspace too small");
    const int nb = red_blocks(n);
    double* part = static_cast<double*>(ws);
    hipLaunchKernelGGL(recon_partial, dim3(nb), dim3(256), 0, (hipStream_t)stream, recons, inputs, part, (long)n, kind);
    MOVAE_CHECK_LAUNCH("recon_partial");
    hipLaunchKernelGGL(final_sum, dim3(1), dim3(256), 0, (hipStream_t)stream, part, nb, (double)scale / (double)n, out);
    MOVAE_CHECK_LAUNCH("final_sum");
    return MOVAE_OK;
}

int movae_recon_loss_bwd(const float* recons, const float* inputs, const float* gscale_dev, float* drecons, size_t n, int kind,
                         float scale, movae_stream_t stream) {
    MOVAE_CHECK_ARG(recons && inputs && drecons && n > 0, "movae_recon_loss_bwd: bad argument");
    MOVAE_CHECK_ARG(kind >= 0 && kind <= 3, "movae_recon_loss_bwd: unknown objective %d", kind);
    hipLaunchKernelGGL(recon_bwd_k, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, recons, inputs, gscale_dev, drecons,
                       (long)n, kind, scale / (float)n);
    MOVAE_CHECK_LAUNCH("recon_bwd");
    return MOVAE_OK;
}

int movae_recon_loss_bwd_act(const float* recons, const float* inputs, const float* gscale_dev, float* dpre, size_t n, int kind,
                             float scale, int act, float slope, movae_stream_t stream) {
    MOVAE_CHECK_ARG(recons && inputs && dpre && n > 0, "movae_recon_loss_bwd_act: bad argument");
    MOVAE_CHECK_ARG(kind >= 0 && kind <= 3, "movae_recon_loss_bwd_act: unknown objective %d", kind);
    MOVAE_CHECK_ARG(act >= MOVAE_ACT_NONE && act <= MOVAE_ACT_SIGMOID, "movae_recon_loss_bwd_act: unknown activation %d", act);
    hipLaunchKernelGGL(recon_bwd_act_k, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, recons, inputs, gscale_dev, dpre, (long)n,
                       kind, scale / (float)n, act, slope);
    MOVAE_CHECK_LAUNCH("recon_bwd_act");
    return MOVAE_OK;
}

int movae_kl_fwd(const float* mu, const float* log_var, float* out, int b, int d, float scale, void* ws, size_t ws_bytes,
                 movae_stream_t stream) {
    MOVAE_WS_SCRATCH(ws, ws_bytes);
    MOVAE_CHECK_ARG(mu && log_var && out && b > 0 && d > 0, "movae_kl_fwd: bad argument");
    const size_t n = (size_t)b * d;
    MOVAE_CHECK_ARG(ws && ws_bytes >= movae_reduce_ws_bytes(n), "movae_kl_fwd: workspace too small");
    const int nb = red_blocks(n);
    double* part = static_cast<double*>(ws);
    hipLaunchKernelGGL(kl_partial, dim3(nb), dim3(256), 0, (hipStream_t)stream, mu, log_var, part, (long)n);
    MOVAE_CHECK_LAUNCH("kl_partial");
    hipLaunchKernelGGL(final_sum, dim3(1), dim3(256), 0, (hipStream_t)stream, part, nb, -0.5 * (double)scale / (double)b, out);
    MOVAE_CHECK_LAUNCH("final_sum");
    return MOVAE_OK;
}

// models/vae.py:211-228 in two launches instead of five: the two partial-sum passes of movae_recon_loss_fwd / movae_kl_fwd as one
// launch and ONE final kernel that also forms total_loss.  out[3] = (reconstruction_loss, kld_loss, total_loss), each value bit-identical to what
// the separate entry points + a tensor add give.
int movae_vae_losses_fwd(const float* recons, const float* inputs, size_t n, int kind, float rec_scale, const float* mu,
                         const float* log_var, int b, int d, float kl_scale, float* out, void* ws, size_t ws_bytes,
                         movae_stream_t stream) {
    MOVAE_WS_SCRATCH(ws, ws_bytes);
    MOVAE_CHECK_ARG(recons && inputs && mu && log_var && out && n > 0 && b > 0 && d > 0, "movae_vae_losses_fwd: bad argument");
    MOVAE_CHECK_ARG(kind >= 0 && kind <= 3, "movae_vae_losses_fwd: unknown objective %d", kind);
    const size_t nk = (size_t)b * d;
    MOVAE_CHECK_ARG(ws && ws_bytes >= movae_reduce_ws_bytes(n) + movae_reduce_ws_bytes(nk), "movae_vae_losses_fwd: workspace too small");
    const int nb1 = red_blocks(n), nb2 = red_blocks(nk);
    double* part = static_cast<double*>(ws);
    hipLaunchKernelGGL(recon_kl_partial, dim3(nb1 + nb2), dim3(256), 0, (hipStream_t)stream, recons, inputs, (long)n, kind, mu, log_var,
                       (long)nk, nb1, nb2, part);
    MOVAE_CHECK_LAUNCH("recon_kl_partial");
    hipLaunchKernelGGL(final_sum2, dim3(1), dim3(256), 0, (hipStream_t)stream, part, nb1, nb2, (double)rec_scale / (double)n,
                       -0.5 * (double)kl_scale / (double)b, out);
    MOVAE_CHECK_LAUNCH("final_sum2");
    return MOVAE_OK;
}

// ---- loss_function arithmetic in one launch --------------------------------------------------------------------------------------
// The models' loss_function methods weight and add scalar terms (models/vq_vae.py:381-391, vq_vae2.py:313-334,
// betatc_vae.py:298-324): as tensor arithmetic that is one launch per `*` and `+`, forward and backward.  Here: T scalar terms,
// K outputs out[k] = f_k * sum_t coef[k][t] * term[t] (f_k = the annealing factor for k == anneal_row, else 1), out[K] = their sum.
// BetaTC's annealing counter (a device float) is advanced and clamped in the same kernel.
struct CLTerms {
    const float* p[8];
};
struct CLCoef {
    float c[64];  // [K][T] row-major, K, T <= 8
};
__global__ void combine_losses_k(CLTerms terms, CLCoef coef, int T, int K, float* __restrict__ iter_dev, float anneal_steps,
                                 int anneal_row, int training, float* __restrict__ out, float* __restrict__ anneal_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float anneal = 1.f;
    if (anneal_row >= 0 && iter_dev) {
        float it = iter_dev[0];
        if (training) {
            it += 1.f;  // models/betatc_vae.py:299: num_iter += 1 per training call
            iter_dev[0] = it;
            anneal = fminf(it / anneal_steps, 1.f);
        }
    }
    if (anneal_out) anneal_out[0] = anneal;
    float tv[8];
    for (int t = 0; t < T; ++t) tv[t] = terms.p[t][0];
    float total = 0.f;
    for (int k = 0; k < K; ++k) {
        // the reference's order: the row's terms are added first (vq_vae2.py: top + bottom), then weight (* annealing factor) * sum
        float v = 0.f;
        bool first = true;
        float w = 0.f;
        for (int t = 0; t < T; ++t) {
            const float c = coef.c[k * T + t];
            if (c == 0.f) continue;
            v = first ? tv[t] : v + tv[t];
            w = c;
            first = false;
        }
        v = (k == anneal_row ? w * anneal : w) * v;  // betatc_vae.py:321: kld_weight * 1 * anneal * kld
        out[k] = v;
        total = k == 0 ? v : total + v;
    }
    out[K] = total;
}

// gterms[t] = sum_k (g[k] + g[K]) * f_k * coef[k][t]   (absent cotangents: null pointers)
__global__ void combine_losses_bwd_k(CLTerms g, CLCoef coef, int T, int K, const float* __restrict__ anneal_dev, int anneal_row,
                                     float* __restrict__ gterms) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float gt = g.p[K] ? g.p[K][0] : 0.f;
    for (int t = 0; t < T; ++t) {
        float acc = 0.f;
        for (int k = 0; k < K; ++k) {
            const float c = coef.c[k * T + t];
            if (c == 0.f) continue;
            float gk = (g.p[k] ? g.p[k][0] : 0.f) + gt;
            if (k == anneal_row && anneal_dev) gk *= anneal_dev[0];
            acc += gk * c;
        }
        gterms[t] = acc;
    }
}

int movae_combine_losses_fwd(int nterms, const float* const* terms, int nout, const float* coef, float* iter_dev, float anneal_steps,
                             int anneal_row, int training, float* out, float* anneal_out, movae_stream_t stream) {
    MOVAE_CHECK_ARG(terms && coef && out && nterms >= 1 && nterms <= 8 && nout >= 1 && nout <= 8, "movae_combine_losses_fwd: bad argument");
    CLTerms tp{};
    CLCoef cc{};
    for (int t = 0; t < nterms; ++t) {
        MOVAE_CHECK_ARG(terms[t], "movae_combine_losses_fwd: null term");
        tp.p[t] = terms[t];
    }
    for (int i = 0; i < nout * nterms; ++i) cc.c[i] = coef[i];
    for (int k = 0; k < nout; ++k) {  // a row's non-zero coefficients must be equal: out[k] = w_k * (sum of its terms)
        float w = 0.f;
        for (int t = 0; t < nterms; ++t) {
            const float c = coef[k * nterms + t];
            MOVAE_CHECK_ARG(c == 0.f || w == 0.f || c == w, "movae_combine_losses_fwd: row %d mixes different weights", k);
            if (c != 0.f) w = c;
        }
    }
    hipLaunchKernelGGL(combine_losses_k, dim3(1), dim3(64), 0, (hipStream_t)stream, tp, cc, nterms, nout, iter_dev, anneal_steps,
                       anneal_row, training, out, anneal_out);
    MOVAE_CHECK_LAUNCH("combine_losses");
    return MOVAE_OK;
}

int movae_combine_losses_bwd(int nterms, int nout, const float* const* g, const float* coef, const float* anneal_dev, int anneal_row,
                             float* gterms, movae_stream_t stream) {
    MOVAE_CHECK_ARG(g && coef && gterms && nterms >= 1 && nterms <= 8 && nout >= 1 && nout <= 7, "movae_combine_losses_bwd: bad argument");
    CLTerms gp{};
    CLCoef cc{};
    for (int k = 0; k <= nout; ++k) gp.p[k] = g[k];
    for (int i = 0; i < nout * nterms; ++i) cc.c[i] = coef[i];
    hipLaunchKernelGGL(combine_losses_bwd_k, dim3(1), dim3(64), 0, (hipStream_t)stream, gp, cc, nterms, nout, anneal_dev, anneal_row, gterms);
    MOVAE_CHECK_LAUNCH("combine_losses_bwd");
    return MOVAE_OK;
}

int movae_kl_bwd(const float* mu, const float* log_var, const float* gscale_dev, float* dmu, float* dlog_var, int b, int d,
                 float scale, movae_stream_t stream) {
    MOVAE_CHECK_ARG(mu && log_var && dmu && dlog_var && b > 0 && d > 0, "movae_kl_bwd: bad argument");
    const long n = (long)b * d;
    hipLaunchKernelGGL(kl_bwd_k, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, mu, log_var, gscale_dev, dmu, dlog_var, n,
                       scale / (float)b);
    MOVAE_CHECK_LAUNCH("kl_bwd");
    return MOVAE_OK;
}

int movae_tc_decomp_fwd(const float* z, const float* mu, const float* log_var, const float* log_iw, float* out,
                        float* lse_joint, float* lse_marg, int b, int d, void* ws, size_t ws_bytes, movae_stream_t stream) {
    MOVAE_WS_SCRATCH(ws, ws_bytes);
    MOVAE_CHECK_ARG(z && mu && log_var && log_iw && out && lse_joint && lse_marg, "movae_tc_decomp_fwd: null pointer");
    MOVAE_CHECK_ARG(b > 1 && d > 0 && d <= 64 * TC_DU, "movae_tc_decomp_fwd: need b > 1 and d <= %d (got b=%d d=%d)", 64 * TC_DU, b, d);
    MOVAE_CHECK_ARG(ws && ws_bytes >= (size_t)b * 3 * sizeof(float), "movae_tc_decomp_fwd: workspace too small");
    float* rows = static_cast<float*>(ws);
    const size_t sh = (size_t)(8 + 8 * d + 4) * sizeof(float);
    hipLaunchKernelGGL(tc_fwd_k, dim3(b), dim3(256), sh, (hipStream_t)stream, z, mu, log_var, log_iw, lse_joint, lse_marg, rows, b, d);
    MOVAE_CHECK_LAUNCH("tc_fwd");
    hipLaunchKernelGGL(tc_mean_k, dim3(1), dim3(64), 0, (hipStream_t)stream, rows, out, b);
    MOVAE_CHECK_LAUNCH("tc_mean");
    return MOVAE_OK;
}

int movae_tc_decomp_bwd(const float* z, const float* mu, const float* log_var, const float* log_iw, const float* lse_joint,
                        const float* lse_marg, const float* g, float* dz, float* dmu, float* dlog_var, int b, int d,
                        movae_stream_t stream) {
    MOVAE_CHECK_ARG(z && mu && log_var && log_iw && lse_joint && lse_marg && g && dz && dmu && dlog_var,
                    "movae_tc_decomp_bwd: null pointer");
    MOVAE_CHECK_ARG(b > 1 && d > 0, "movae_tc_decomp_bwd: bad shape");
    // DT threads along d, 1024 / DT row lanes (at most one per row)
    int dt = 32;
    while (dt < d && dt < 256) dt *= 2;
    int lanes = 1024 / dt;
    while (lanes > 1 && lanes > b) lanes /= 2;
    const int th = dt * lanes;
    hipLaunchKernelGGL(tc_bwd_dz, dim3(b), dim3(th), (size_t)th * sizeof(float), (hipStream_t)stream, z, mu, log_var, log_iw, lse_joint,
                       lse_marg, g, dz, b, d, dt);
    MOVAE_CHECK_LAUNCH("tc_bwd_dz");
    hipLaunchKernelGGL(tc_bwd_dparams, dim3(b), dim3(th), (size_t)2 * th * sizeof(float), (hipStream_t)stream, z, mu, log_var, log_iw,
                       lse_joint, lse_marg, g, dmu, dlog_var, b, d, dt);
    MOVAE_CHECK_LAUNCH("tc_bwd_dparams");
    return MOVAE_OK;
}

}  // extern "C"
