// BatchNorm2d (training statistics) + activation over NHWC activations viewed as [rows][C].
// HBM-bound: the statistics pass reads y once (fp64 accumulation, deterministic two-stage
// reduction), the apply pass reads y and writes out once.  Backward mirrors it.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int MAX_PARTS = 256;
constexpr int MAX_GROUPS = 8;  // == MOVAE_MAX_K: cotangent groups of one batched backward (blockIdx.y)

struct BnOut {  // per-group destinations of dgamma / dbeta (rows of the Jacobian arena), by value in the kernel arguments
    float* dgamma[MAX_GROUPS];
    float* dbeta[MAX_GROUPS];
};

__host__ __device__ inline int pow2_ge(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

struct Split {
    int CB;          // columns handled per pass (power of two <= 256)
    int RG;          // row groups per block = 256 / CB
    int rows_per_block;
    int nblk;
};

inline Split make_split(int rows, int c) {
    Split s;
    s.CB = pow2_ge(c) < 256 ? pow2_ge(c) : 256;
    s.RG = 256 / s.CB;
    int rpb = ceil_div(rows, MAX_PARTS);
    rpb = ceil_div(rpb, s.RG) * s.RG;
    if (rpb < s.RG * 4) rpb = s.RG * 4;
    s.rows_per_block = rpb;
    s.nblk = ceil_div(rows, rpb);
    return s;
}

// partial[blk][c][2] = (sum y, sum y^2) over the block's rows
__global__ __launch_bounds__(256) void bn_stats_partial(const float* __restrict__ y, double* __restrict__ part, int rows,
                                                        int C, int CB, int rows_per_block) {
    __shared__ double sh[2][256];
    const int t = threadIdx.x;
    const int RG = 256 / CB;
    const int cl = t % CB, rg = t / CB;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min((long)rows, r0 + rows_per_block);
    for (int cb = 0; cb < C; cb += CB) {
        const int c = cb + cl;
        double s = 0.0, q = 0.0;
        if (c < C)
            for (long r = r0 + rg; r < r1; r += RG) {
                const double v = y[r * C + c];
                s += v;
                q += v * v;
            }
        sh[0][t] = s;
        sh[1][t] = q;
        __syncthreads();
        if (rg == 0 && c < C) {
            for (int i = 1; i < RG; ++i) {
                s += sh[0][i * CB + cl];
                q += sh[1][i * CB + cl];
            }
            part[((long)blockIdx.x * C + c) * 2 + 0] = s;
            part[((long)blockIdx.x * C + c) * 2 + 1] = q;
        }
        __syncthreads();
    }
}

// one wave per channel: lanes stride over the block partials, fp64 shuffle reduction
__global__ __launch_bounds__(64) void bn_stats_final(const double* __restrict__ part, int nblk, int rows, int C, float eps,
                                                     float momentum, float* __restrict__ save_mean,
                                                     float* __restrict__ save_rstd, float* __restrict__ running_mean,
                                                     float* __restrict__ running_var, long long* __restrict__ nbt) {
    const int c = blockIdx.x;
    if (nbt && c == 0 && threadIdx.x == 0) nbt[0] += 1;  // nn.BatchNorm2d.num_batches_tracked, no launch of its own
    double s = 0.0, q = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) {
        s += part[((long)b * C + c) * 2 + 0];
        q += part[((long)b * C + c) * 2 + 1];
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if (threadIdx.x != 0) return;
    const double mean = s / rows;
    double var = q / rows - mean * mean;
    if (var < 0.0) var = 0.0;
    save_mean[c] = (float)mean;
    save_rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unb = rows > 1 ? var * ((double)rows / (rows - 1)) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
    }
}

__global__ void bn_eval_prepare(const float* __restrict__ running_mean, const float* __restrict__ running_var, float eps,
                                float* __restrict__ save_mean, float* __restrict__ save_rstd, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    save_mean[c] = running_mean[c];
    save_rstd[c] = 1.f / sqrtf(running_var[c] + eps);
}

template <bool VEC>
__global__ __launch_bounds__(256) void bn_apply(const float* __restrict__ y, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, const float* __restrict__ mean,
                                                const float* __restrict__ rstd, float* __restrict__ out, long total, int C,
                                                int act, float slope) {
    const long stride = (long)gridDim.x * blockDim.x;
    if (VEC) {  // 4 independent 16-byte loads in flight per thread; the per-channel parameters come as float4 too
        const long nv = total / 4;
        for (long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x; i0 < nv; i0 += 4 * stride) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long i = i0 + u * stride;
                if (i < nv) v[u] = reinterpret_cast<const f32x4*>(y)[i];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long i = i0 + u * stride;
                if (i < nv) {
                    const int c = (int)((i * 4) % C);
                    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), rs = *reinterpret_cast<const f32x4*>(rstd + c);
                    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
                    f32x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = apply_act((v[u][j] - mu[j]) * rs[j] * ga[j] + be[j], act, slope);
                    reinterpret_cast<f32x4*>(out)[i] = o;
                }
            }
        }
    } else {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
            const int c = (int)(i % C);
            out[i] = apply_act((y[i] - mean[c]) * rstd[c] * gamma[c] + beta[c], act, slope);
        }
    }
}

// backward partials: (sum dz, sum dz * xhat)
__global__ __launch_bounds__(256) void bn_bwd_partial(const float* __restrict__ dout, const float* __restrict__ y,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      double* __restrict__ part, int rows, int C, int CB,
                                                      int rows_per_block, int act, float slope) {
    __shared__ double sh[2][256];
    const int t = threadIdx.x;
    const int RG = 256 / CB;
    const int cl = t % CB, rg = t / CB;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min((long)rows, r0 + rows_per_block);
    for (int cb = 0; cb < C; cb += CB) {
        const int c = cb + cl;
        double s = 0.0, q = 0.0;
        if (c < C) {
            const float mu = mean[c], rs = rstd[c], ga = gamma[c], be = beta[c];
            for (long r = r0 + rg; r < r1; r += RG) {
                const float xh = (y[r * C + c] - mu) * rs;
                const float o = apply_act(xh * ga + be, act, slope);
                const float dz = dout[r * C + c] * act_grad_from_out(o, act, slope);
                s += dz;
                q += (double)dz * xh;
            }
        }
        sh[0][t] = s;
        sh[1][t] = q;
        __syncthreads();
        if (rg == 0 && c < C) {
            for (int i = 1; i < RG; ++i) {
                s += sh[0][i * CB + cl];
                q += sh[1][i * CB + cl];
            }
            part[((long)blockIdx.x * C + c) * 2 + 0] = s;
            part[((long)blockIdx.x * C + c) * 2 + 1] = q;
        }
        __syncthreads();
    }
}

// sums[c] = (mean dz, mean dz*xhat) ; dgamma / dbeta written.  One wave per channel.
__global__ __launch_bounds__(64) void bn_bwd_final(const double* __restrict__ part, int nblk, int rows, int C,
                                                   float* __restrict__ sums, BnOut tab, int accumulate) {
    const int c = blockIdx.x, grp = blockIdx.y;
    part += (long)grp * nblk * C * 2;
    sums += (long)grp * 2 * C;
    float* __restrict__ dgamma = tab.dgamma[grp];
    float* __restrict__ dbeta = tab.dbeta[grp];
    double s = 0.0, q = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) {
        s += part[((long)b * C + c) * 2 + 0];
        q += part[((long)b * C + c) * 2 + 1];
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if (threadIdx.x != 0) return;
    sums[2 * c + 0] = (float)(s / rows);
    sums[2 * c + 1] = (float)(q / rows);
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)q : (float)q;
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)s : (float)s;
}

__global__ __launch_bounds__(256) void bn_bwd_apply(const float* __restrict__ dout, const float* __restrict__ y,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                    const float* __restrict__ sums, float* __restrict__ dy, long total,
                                                    int C, int act, float slope) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % C);
        const float xh = (y[i] - mean[c]) * rstd[c];
        const float o = apply_act(xh * gamma[c] + beta[c], act, slope);
        const float dz = dout[i] * act_grad_from_out(o, act, slope);
        dy[i] = gamma[c] * rstd[c] * (dz - sums[2 * c] - xh * sums[2 * c + 1]);
    }
}

// ---- 16-byte variants (C % 4 == 0): a thread owns one 4-channel quad, a wave reads 1 KiB contiguous ----
struct Split4 {
    int CQB, RG, rows_per_block, nblk;
};

// max_parts = most partial blocks: measured best ~256 for the forward statistics and for grouped backward passes,
// ~1024 for a single-cotangent backward (two input streams per thread)
inline Split4 make_split4(int rows, int c, int parts = 256) {
    Split4 s;
    const int cq = c / 4;
    s.CQB = pow2_ge(cq) < 256 ? pow2_ge(cq) : 256;
    s.RG = 256 / s.CQB;
    static const int env_parts = getenv("MOVAE_BN_PARTS") ? atoi(getenv("MOVAE_BN_PARTS")) : 0;
    static const int min_iter = getenv("MOVAE_BN_MINITER") ? atoi(getenv("MOVAE_BN_MINITER")) : 4;
    const int max_parts = env_parts > 0 ? env_parts : parts;
    int rpb = ceil_div(rows, max_parts);  // up to 4 partial blocks per CU keep enough 16-byte loads in flight
    rpb = ceil_div(rpb, s.RG) * s.RG;
    if (rpb < s.RG * min_iter) rpb = s.RG * min_iter;
    s.rows_per_block = rpb;
    s.nblk = ceil_div(rows, rpb);
    return s;
}

__global__ __launch_bounds__(256) void bn_stats_partial4(const float* __restrict__ y, double* __restrict__ part, int rows,
                                                         int C, int CQB, int rows_per_block, unsigned* __restrict__ counter,
                                                         float eps, float momentum, float* __restrict__ save_mean,
                                                         float* __restrict__ save_rstd, float* __restrict__ running_mean,
                                                         float* __restrict__ running_var, long long* __restrict__ nbt) {
    __shared__ double sh[8 * 256];
    const int t = threadIdx.x, RG = 256 / CQB, cl = t % CQB, rg = t / CQB, CQ = C / 4;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min((long)rows, r0 + rows_per_block);
    for (int cb = 0; cb < CQ; cb += CQB) {
        const int cq = cb + cl;
        double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (cq < CQ)
            for (long r = r0 + rg; r < r1; r += 4 * RG) {  // 4 independent 16-byte loads in flight per lane
                f32x4 x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long rr = r + (long)u * RG;
                    x[u] = rr < r1 ? *reinterpret_cast<const f32x4*>(y + rr * C + cq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] += (double)x[u][j];
                        v[4 + j] += (double)x[u][j] * (double)x[u][j];
                    }
            }
        fold_columns_256<8>(v, sh, CQB);
        if (rg == 0 && cq < CQ)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                part[((long)blockIdx.x * C + cq * 4 + j) * 2 + 0] = v[j];
                part[((long)blockIdx.x * C + cq * 4 + j) * 2 + 1] = v[4 + j];
            }
    }
    if (counter == nullptr || !arrive_last(counter, gridDim.x)) return;
    // last block: one wave per channel folds the block partials (fp64) and publishes mean / rstd / running stats
    if (nbt && t == 0) nbt[0] += 1;
    const int lane = t & 63, nblk = gridDim.x;
    for (int c = t >> 6; c < C; c += 4) {
        double s = 0.0, q = 0.0;
        for (int b = lane; b < nblk; b += 64) {
            s += part[((long)b * C + c) * 2 + 0];
            q += part[((long)b * C + c) * 2 + 1];
        }
        s = wave_sum(s);
        q = wave_sum(q);
        if (lane == 0) {
            const double mean = s / rows;
            double var = q / rows - mean * mean;
            if (var < 0.0) var = 0.0;
            save_mean[c] = (float)mean;
            save_rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
            if (running_mean) {
                const double unb = rows > 1 ? var * ((double)rows / (rows - 1)) : var;
                running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
                running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
            }
        }
    }
}

__global__ __launch_bounds__(256) void bn_bwd_partial4(const float* __restrict__ dout, const float* __restrict__ y,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       double* __restrict__ part, int rows, int C, int CQB,
                                                       int rows_per_block, int act, float slope,
                                                       unsigned* __restrict__ counter, float* __restrict__ sums,
                                                       BnOut tab, int accumulate, long gstride) {
    __shared__ double sh[8 * 256];
    dout += (long)blockIdx.y * gstride;                  // cotangent group; y and the statistics are shared
    part += (long)blockIdx.y * gridDim.x * C * 2;
    float* __restrict__ dgamma = tab.dgamma[blockIdx.y];
    float* __restrict__ dbeta = tab.dbeta[blockIdx.y];
    const int t = threadIdx.x, RG = 256 / CQB, cl = t % CQB, rg = t / CQB, CQ = C / 4;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min((long)rows, r0 + rows_per_block);
    for (int cb = 0; cb < CQ; cb += CQB) {
        const int cq = cb + cl;
        double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (cq < CQ) {
            const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + cq * 4), rs = *reinterpret_cast<const f32x4*>(rstd + cq * 4);
            const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + cq * 4), be = *reinterpret_cast<const f32x4*>(beta + cq * 4);
            for (long r = r0 + rg; r < r1; r += 4 * RG) {  // 8 independent 16-byte loads in flight per lane
                f32x4 yy[4], go[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long rr = r + (long)u * RG;
                    const bool ok = rr < r1;
                    yy[u] = ok ? *reinterpret_cast<const f32x4*>(y + rr * C + cq * 4) : mu;
                    go[u] = ok ? *reinterpret_cast<const f32x4*>(dout + rr * C + cq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xh = (yy[u][j] - mu[j]) * rs[j];
                        const float o = apply_act(xh * ga[j] + be[j], act, slope);
                        const float dz = go[u][j] * act_grad_from_out(o, act, slope);
                        v[j] += (double)dz;
                        v[4 + j] += (double)dz * (double)xh;
                    }
            }
        }
        fold_columns_256<8>(v, sh, CQB);
        if (rg == 0 && cq < CQ)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                part[((long)blockIdx.x * C + cq * 4 + j) * 2 + 0] = v[j];
                part[((long)blockIdx.x * C + cq * 4 + j) * 2 + 1] = v[4 + j];
            }
    }
    if (counter == nullptr || !arrive_last(counter, gridDim.x)) return;
    const int lane = t & 63, nblk = gridDim.x;
    for (int c = t >> 6; c < C; c += 4) {
        double s = 0.0, q = 0.0;
        for (int b = lane; b < nblk; b += 64) {
            s += part[((long)b * C + c) * 2 + 0];
            q += part[((long)b * C + c) * 2 + 1];
        }
        s = wave_sum(s);
        q = wave_sum(q);
        if (lane == 0) {
            sums[2 * c + 0] = (float)(s / rows);
            sums[2 * c + 1] = (float)(q / rows);
            if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)q : (float)q;
            if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)s : (float)s;
        }
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply4(const float* __restrict__ dout, const float* __restrict__ y,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ sums, float* __restrict__ dy, long total, int C,
                                                     int act, float slope) {
    dout += (long)blockIdx.y * total;  // cotangent group (stacked [G][rows][C]); y is shared
    dy += (long)blockIdx.y * total;
    sums += (long)blockIdx.y * 2 * C;
    const long stride = (long)gridDim.x * blockDim.x, nv = total / 4;
    for (long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x; i0 < nv; i0 += 4 * stride) {
        f32x4 yy[4], go[4];  // 8 independent 16-byte loads in flight per thread
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = i0 + u * stride;
            if (i < nv) {
                yy[u] = reinterpret_cast<const f32x4*>(y)[i];
                go[u] = reinterpret_cast<const f32x4*>(dout)[i];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = i0 + u * stride;
            if (i < nv) {
                const int c = (int)((i * 4) % C);
                const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), rs = *reinterpret_cast<const f32x4*>(rstd + c);
                const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
                f32x4 o4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = (yy[u][j] - mu[j]) * rs[j];
                    const float o = apply_act(xh * ga[j] + be[j], act, slope);
                    const float dz = go[u][j] * act_grad_from_out(o, act, slope);
                    o4[j] = ga[j] * rs[j] * (dz - sums[2 * (c + j)] - xh * sums[2 * (c + j) + 1]);
                }
                reinterpret_cast<f32x4*>(dy)[i] = o4;
            }
        }
    }
}

// ---- one-launch BatchNorm for short tensors ----------------------------------------------------------------------
// rows <= small_rows(): a block owns 4 adjacent channels (one 16-byte column), its 256 threads stride over the rows
// (at most 4 rows per thread), twice -- statistics, then apply; the second walk hits L2 -- and needs no partial buffer
// and no second / third launch.  These layers (rows = batch * 4 ... batch) are launch-latency bound: a dependent launch
// costs ~1.7 us inside a replayed graph, more than the whole tensor takes to stream.  Longer tensors lose here (one
// block per column cannot keep enough loads in flight: measured 31 us vs ~9 us for rows = 4096) and take the
// three-launch path.
// v[8] per thread -> (s[4], q[4]) of the block, returned to every thread through LDS
__device__ __forceinline__ void small_fold(double (&v)[8], double (*sh)[8], int t) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = wave_sum(v[j]);
    if ((t & 63) == 0)
#pragma unroll
        for (int j = 0; j < 8; ++j) sh[t >> 6][j] = v[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = sh[0][j] + sh[1][j] + sh[2][j] + sh[3][j];
}

__global__ __launch_bounds__(256) void bn_fwd_small(const float* __restrict__ y, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float* __restrict__ out,
                                                    float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                                    float* __restrict__ running_mean, float* __restrict__ running_var,
                                                    long long* __restrict__ nbt, int rows, int C, float eps, float momentum,
                                                    int act, float slope) {
    __shared__ double sh[4][8];
    const int t = threadIdx.x;
    const int c0 = blockIdx.x * 4;
    f32x4 x[4];
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // rows <= 1024: every row of the column is held in registers for the apply pass
        const int r = t + u * 256;
        x[u] = r < rows ? *reinterpret_cast<const f32x4*>(y + (long)r * C + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] += (double)x[u][j];
            v[4 + j] += (double)x[u][j] * (double)x[u][j];
        }
    }
    small_fold(v, sh, t);
    f32x4 mu, rs;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double mean = v[j] / rows;
        double var = v[4 + j] / rows - mean * mean;
        if (var < 0.0) var = 0.0;
        mu[j] = (float)mean;
        rs[j] = (float)(1.0 / sqrt(var + (double)eps));
        if (t == j) {
            const int c = c0 + j;
            save_mean[c] = mu[j];
            save_rstd[c] = rs[j];
            if (running_mean) {
                const double unb = rows > 1 ? var * ((double)rows / (rows - 1)) : var;
                running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
                running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
            }
        }
    }
    if (nbt && blockIdx.x == 0 && t == 0) nbt[0] += 1;
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0), be = *reinterpret_cast<const f32x4*>(beta + c0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = t + u * 256;
        if (r < rows) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = apply_act((x[u][j] - mu[j]) * rs[j] * ga[j] + be[j], act, slope);
            *reinterpret_cast<f32x4*>(out + (long)r * C + c0) = o;
        }
    }
}

__global__ __launch_bounds__(256) void bn_bwd_small(const float* __restrict__ dout, const float* __restrict__ y,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                    float* __restrict__ dy, BnOut tab, int rows, int C, int act, float slope,
                                                    int accumulate) {
    __shared__ double sh[4][8];
    const int t = threadIdx.x;
    const int c0 = blockIdx.x * 4;
    dout += (long)blockIdx.y * rows * C;  // cotangent group (stacked [G][rows][C]); y is shared
    dy += (long)blockIdx.y * rows * C;
    float* __restrict__ dgamma = tab.dgamma[blockIdx.y];
    float* __restrict__ dbeta = tab.dbeta[blockIdx.y];
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c0), rs = *reinterpret_cast<const f32x4*>(rstd + c0);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0), be = *reinterpret_cast<const f32x4*>(beta + c0);
    f32x4 xh[4], dz[4];  // kept in registers for the second pass
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = t + u * 256;
        const bool in = r < rows;
        const f32x4 yy = in ? *reinterpret_cast<const f32x4*>(y + (long)r * C + c0) : mu;
        const f32x4 go = in ? *reinterpret_cast<const f32x4*>(dout + (long)r * C + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xh[u][j] = (yy[j] - mu[j]) * rs[j];
            const float o = apply_act(xh[u][j] * ga[j] + be[j], act, slope);
            dz[u][j] = go[j] * act_grad_from_out(o, act, slope);
            v[j] += (double)dz[u][j];
            v[4 + j] += (double)dz[u][j] * (double)xh[u][j];
        }
    }
    small_fold(v, sh, t);
    f32x4 m1, m2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        m1[j] = (float)(v[j] / rows);
        m2[j] = (float)(v[4 + j] / rows);
        if (t == j) {
            const int c = c0 + j;
            if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)v[4 + j] : (float)v[4 + j];
            if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)v[j] : (float)v[j];
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = t + u * 256;
        if (r < rows) {
            f32x4 o4;
#pragma unroll
            for (int j = 0; j < 4; ++j) o4[j] = ga[j] * rs[j] * (dz[u][j] - m1[j] - xh[u][j] * m2[j]);
            *reinterpret_cast<f32x4*>(dy + (long)r * C + c0) = o4;
        }
    }
}


// ---- fused BatchNorm: the producer conv's epilogue (or its split-K reduce) wrote per-column partial
// sums; this finishes them.  One wave per channel: lanes stride over the partials, fp64 shuffle fold; then everything the
// consumers need: the saved statistics for the backward, the folded affine map x_hat*gamma+beta = scale*y + shift that the next
// conv applies while it loads y, and nn.BatchNorm2d's running statistics / num_batches_tracked.
__global__ __launch_bounds__(256) void bn_finalize_k(const float* __restrict__ part, int parts, int rows, int C,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                    float momentum, float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                                    float* __restrict__ scale, float* __restrict__ shift,
                                                    float* __restrict__ running_mean, float* __restrict__ running_var,
                                                    long long* __restrict__ nbt) {
    const int c = blockIdx.x;
    if (nbt && c == 0 && threadIdx.x == 0) nbt[0] += 1;
    __shared__ double sh[8];
    double s = 0.0, q = 0.0;
    for (int p = threadIdx.x; p < parts; p += 256) {  // (a block per channel: up to FOLD_ABOVE partials without a fold launch)
        s += (double)part[((long)p * 2 + 0) * C + c];
        q += (double)part[((long)p * 2 + 1) * C + c];
    }
    s = block_sum_256(s, sh);
    q = block_sum_256(q, sh + 4);
    if (threadIdx.x != 0) return;
    const double mean = s / rows;
    double var = q / rows - mean * mean;
    if (var < 0.0) var = 0.0;
    const float m = (float)mean, r = (float)(1.0 / sqrt(var + (double)eps));
    save_mean[c] = m;
    save_rstd[c] = r;
    // the same fp32 arithmetic as the stand-alone apply kernel: (y - mean) * rstd * gamma + beta, folded
    const float a = r * gamma[c];
    scale[c] = a;
    shift[c] = fmaf(-m, a, beta[c]);
    if (running_mean) {
        const double unb = rows > 1 ? var * ((double)rows / (rows - 1)) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
    }
}

// out = act(scale[c] * y + shift[c]): materialises a fused BatchNorm output where a consumer cannot apply it on load
__global__ __launch_bounds__(256) void scale_shift_act_k(const float* __restrict__ y, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float* __restrict__ out, long nv, int C,
                                                         float slope) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
        const int c = (int)((i * 4) % C);
        const f32x4 v = reinterpret_cast<const f32x4*>(y)[i];
        const f32x4 a = *reinterpret_cast<const f32x4*>(scale + c), b = *reinterpret_cast<const f32x4*>(shift + c);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float z = fmaf(v[j], a[j], b[j]);
            o[j] = z > 0.f ? z : z * slope;
        }
        reinterpret_cast<f32x4*>(out)[i] = o;
    }
}

// Many partials (a 262144-row layer leaves 8192 per channel): folding them one wave per channel is a long strided read.  This
// first stage cuts the partial rows into gridDim.x slices; block b sums its slice for every column (reads coalesced across the
// 2C columns, fixed order) into row b of `dst` -- the finalize kernels then fold gridDim.x rows.  blockIdx.y = cotangent group.
__global__ __launch_bounds__(256) void bn_partials_fold_k(const float* __restrict__ part, int P, int E, float* __restrict__ dst) {
    __shared__ float sh[256];
    const int t = threadIdx.x, nb = gridDim.x, b = blockIdx.x;
    part += (long)blockIdx.y * P * E;
    dst += (long)blockIdx.y * nb * E;
    const int chunk = (P + nb - 1) / nb, p0 = b * chunk, p1 = min(P, p0 + chunk);
    const int EL = E < 256 ? E : 256, PL = 256 / EL;  // column lanes x partial-row lanes
    const int el = t % EL, pl = t / EL;
    for (int e0 = 0; e0 < E; e0 += EL) {
        const int e = e0 + el;
        float acc = 0.f;
        if (e < E && pl < PL)
            for (int p = p0 + pl; p < p1; p += PL) acc += part[(long)p * E + e];
        sh[t] = acc;
        __syncthreads();
        if (pl == 0 && e < E) {
            float s = acc;
            for (int i = 1; i < PL; ++i) s += sh[i * EL + el];
            dst[(long)b * E + e] = s;
        }
        __syncthreads();
    }
}

// -> (partials to finalise, their count): folds in the free tail of the caller's buffer when there are many
inline const float* fold_partials(const float* part, size_t cap_floats, int* parts, int groups, int c, hipStream_t st) {
    const int P = *parts, E = 2 * c;
    const int nb = 64;
    // (the finalize kernels take a 256-thread block per channel: up to 2048 partials -- 8 strided pairs per thread -- cost less than
    // a fold launch and its dependent kernel boundary)
    static const int fold_above = getenv("MOVAE_BN_FOLD_ABOVE") ? atoi(getenv("MOVAE_BN_FOLD_ABOVE")) : 2048;
    if (P <= fold_above || (size_t)groups * P * E + (size_t)groups * nb * E > cap_floats) return part;
    float* dst = const_cast<float*>(part) + (size_t)groups * P * E;
    hipLaunchKernelGGL(bn_partials_fold_k, dim3(nb, groups), dim3(256), 0, st, part, P, E, dst);
    *parts = nb;
    return dst;
}

// ---- fused BatchNorm backward: the input-gradient pass that produced dout already emitted, per cotangent
// group, the partial sums S1 = sum d, S2 = sum d * y with d = dout * act'(scale * y + shift).  One wave per (channel, group):
//   dbeta = S1,  dgamma = sum d * x_hat = rstd * (S2 - mean * S1)
//   dy = gamma * rstd * (d - S1/M - x_hat * dgamma/M) = k1 * d + c2 * y + c3,
//   k1 = gamma * rstd,  c2 = -k1 * rstd * dgamma / M,  c3 = -k1 * S1 / M - c2 * mean          coef[g][0..2][C]
__global__ __launch_bounds__(256) void bn_bwd_finalize_k(const float* __restrict__ part, int ppg, int rows, int C,
                                                        const float* __restrict__ gamma, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, BnOut out, float* __restrict__ coef,
                                                        int accumulate) {
    const int c = blockIdx.x, g = blockIdx.y;
    __shared__ double sh[8];
    double s1 = 0.0, s2 = 0.0;
    for (int p = threadIdx.x; p < ppg; p += 256) {
        const long q = (long)g * ppg + p;
        s1 += (double)part[(q * 2 + 0) * C + c];
        s2 += (double)part[(q * 2 + 1) * C + c];
    }
    s1 = block_sum_256(s1, sh);
    s2 = block_sum_256(s2, sh + 4);
    if (threadIdx.x != 0) return;
    const double m = mean[c], r = rstd[c], ga = gamma[c];
    const double dgam = r * (s2 - m * s1);
    if (out.dbeta[g]) out.dbeta[g][c] = (float)(accumulate ? out.dbeta[g][c] + s1 : s1);
    if (out.dgamma[g]) out.dgamma[g][c] = (float)(accumulate ? out.dgamma[g][c] + dgam : dgam);
    const double k1 = ga * r, c2 = -k1 * r * dgam / rows, c3 = -k1 * s1 / rows - c2 * m;
    float* cf = coef + (long)g * 3 * C;
    cf[c] = (float)k1;
    cf[C + c] = (float)c2;
    cf[2 * C + c] = (float)c3;
}

// dy[g][i] = k1[c] * dout[g][i] * act'(scale[c] * y[i] + shift[c]) + c2[c] * y[i] + c3[c]
__global__ __launch_bounds__(256) void bn_bwd_apply_coef_k(const float* __restrict__ dout, const float* __restrict__ y,
                                                           const float* __restrict__ scale, const float* __restrict__ shift, float slope,
                                                           const float* __restrict__ coef, float* __restrict__ dy, long nv, int C) {
    const int g = blockIdx.y;
    const float* cf = coef + (long)g * 3 * C;
    const long off = (long)g * nv;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
        const int c = (int)((i * 4) % C);
        const f32x4 d4 = reinterpret_cast<const f32x4*>(dout)[off + i], y4 = reinterpret_cast<const f32x4*>(y)[i];
        const f32x4 a = *reinterpret_cast<const f32x4*>(scale + c), b = *reinterpret_cast<const f32x4*>(shift + c);
        const f32x4 k1 = *reinterpret_cast<const f32x4*>(cf + c), c2 = *reinterpret_cast<const f32x4*>(cf + C + c),
                    c3 = *reinterpret_cast<const f32x4*>(cf + 2 * C + c);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float z = fmaf(y4[j], a[j], b[j]);
            const float d = d4[j] * (z > 0.f ? 1.f : slope);
            o[j] = fmaf(k1[j], d, fmaf(c2[j], y4[j], c3[j]));
        }
        reinterpret_cast<f32x4*>(dy)[off + i] = o;
    }
}

// Both steps in one launch for the layers whose sums arrive in few partials (the deep, latency-bound ones: <= 256 per group).  A block
// owns 32 channels x a chunk of rows: it first folds the partials of its 32 channels itself (<= 64 KiB out of L2, eight pairs in flight per lane, doubles, the
// arithmetic of bn_bwd_finalize_k), keeps the three coefficients in LDS and then forms dy for its rows -- eight lanes cover the 128
// bytes of a row's channel slice.  The blocks of the first row chunk also write dgamma / dbeta.
__global__ __launch_bounds__(256) void bn_bwd_fused_k(const float* __restrict__ part, int ppg, int rows, int C,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, BnOut out, int accumulate,
                                                     const float* __restrict__ dout, const float* __restrict__ y,
                                                     const float* __restrict__ scale, const float* __restrict__ shift, float slope,
                                                     float* __restrict__ dy, int rows_per_chunk) {
    const int g = blockIdx.z, c0 = blockIdx.y * 32, t = threadIdx.x;
    __shared__ double red[2][8][32];
    __shared__ __attribute__((aligned(16))) float cf[3][32];
    {
        const int cl = t & 31, c = c0 + cl, pl = t >> 5;
        double s1 = 0.0, s2 = 0.0;
        if (c < C)
            for (int p0 = pl; p0 < ppg; p0 += 64) {  // eight partial pairs in flight per lane
                float v1[8], v2[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int p = p0 + 8 * u;
                    const long q = (long)g * ppg + (p < ppg ? p : pl);
                    v1[u] = part[(q * 2 + 0) * C + c], v2[u] = part[(q * 2 + 1) * C + c];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (p0 + 8 * u < ppg) s1 += (double)v1[u], s2 += (double)v2[u];
            }
        red[0][pl][cl] = s1;
        red[1][pl][cl] = s2;
        __syncthreads();
        if (t < 32 && c < C) {
            s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i) s1 += red[0][i][cl], s2 += red[1][i][cl];
            const double m = mean[c], r = rstd[c], ga = gamma[c];
            const double dgam = r * (s2 - m * s1);
            if (blockIdx.x == 0) {
                if (out.dbeta[g]) out.dbeta[g][c] = (float)(accumulate ? out.dbeta[g][c] + s1 : s1);
                if (out.dgamma[g]) out.dgamma[g][c] = (float)(accumulate ? out.dgamma[g][c] + dgam : dgam);
            }
            const double k1 = ga * r, c2 = -k1 * r * dgam / rows, c3 = -k1 * s1 / rows - c2 * m;
            cf[0][cl] = (float)k1, cf[1][cl] = (float)c2, cf[2][cl] = (float)c3;
        }
        __syncthreads();
    }
    const int q = t & 7, c = c0 + 4 * q;
    if (c >= C) return;
    const f32x4 a = *reinterpret_cast<const f32x4*>(scale + c), b = *reinterpret_cast<const f32x4*>(shift + c);
    const f32x4 k1 = *reinterpret_cast<const f32x4*>(&cf[0][4 * q]), c2 = *reinterpret_cast<const f32x4*>(&cf[1][4 * q]),
                c3 = *reinterpret_cast<const f32x4*>(&cf[2][4 * q]);
    const long goff = (long)g * rows * C;
    const int row0 = blockIdx.x * rows_per_chunk, row1 = min(rows, row0 + rows_per_chunk);
    for (int row = row0 + (t >> 3); row < row1; row += 4 * 32) {
        f32x4 d4[4], y4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = row + 32 * u;
            const bool ok = rr < row1;
            d4[u] = ok ? *reinterpret_cast<const f32x4*>(dout + goff + (long)rr * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
            y4[u] = ok ? *reinterpret_cast<const f32x4*>(y + (long)rr * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = row + 32 * u;
            if (rr >= row1) break;
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z = fmaf(y4[u][j], a[j], b[j]);
                const float d = d4[u][j] * (z > 0.f ? 1.f : slope);
                o[j] = fmaf(k1[j], d, fmaf(c2[j], y4[u][j], c3[j]));
            }
            *reinterpret_cast<f32x4*>(dy + goff + (long)rr * C + c) = o;
        }
    }
}

inline int small_rows() {  // MOVAE_BN_SMALL_ROWS: largest row count served by the one-launch kernels (0 disables them)
    static const int v = getenv("MOVAE_BN_SMALL_ROWS") ? atoi(getenv("MOVAE_BN_SMALL_ROWS")) : 1024;
    return v < 1024 ? v : 1024;  // the kernels hold the whole column in registers: 4 rows per thread
}

inline int grid_for(long total) {
    long g = (total + 255) / 256;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

// The "last block folds the partials" variant (arrive_last) is kept for experiments only: measured on MI355X
// it is SLOWER than a second tiny launch for these kernels (C2 step 3.2 ms vs 2.25 ms) because every block pays
// an agent-scope release (L2 write-back) while the conv outputs are still dirty in L2.  MOVAE_INLAUNCH=1 enables it.
inline bool in_launch_final() {
    static const bool v = getenv("MOVAE_INLAUNCH") != nullptr;
    return v;
}

inline bool al16(const void* a, const void* b = nullptr, const void* c = nullptr) {
    return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
}

}  // namespace

extern "C" {

size_t movae_bn_ws_bytes(int rows, int c) {
    if (rows <= 0 || c <= 0) return 0;
    const int nblk = (c % 4 == 0) ? make_split4(rows, c, 1024).nblk : make_split(rows, c).nblk;
    return MOVAE_WS_HEADER_BYTES + (size_t)nblk * c * 2 * sizeof(double) + (size_t)c * 2 * sizeof(float) + 64;
}

int movae_bn_act_fwd(const float* y, const float* gamma, const float* beta, float* out, float* save_mean, float* save_rstd,
                     float* running_mean, float* running_var, long long* num_batches_tracked, int rows, int c, float eps,
                     float momentum, int training, int act, float slope, void* ws, size_t ws_bytes, movae_stream_t stream) {
    MOVAE_CHECK_ARG(y && gamma && beta && out && save_mean && save_rstd, "movae_bn_act_fwd: null pointer");
    MOVAE_CHECK_ARG(rows > 0 && c > 0, "movae_bn_act_fwd: bad shape rows=%d c=%d", rows, c);
    hipStream_t st = (hipStream_t)stream;
    long long* nbt = training ? num_batches_tracked : nullptr;
    if (training && rows <= small_rows() && c % 4 == 0 && al16(y, out) && al16(gamma, beta)) {
        hipLaunchKernelGGL(bn_fwd_small, dim3(c / 4), dim3(256), 0, st, y, gamma, beta, out, save_mean, save_rstd,
                           running_mean, running_var, nbt, rows, c, eps, momentum, act, slope);
        MOVAE_CHECK_LAUNCH("bn_fwd_small");
        return MOVAE_OK;
    }
    if (training) {
        MOVAE_CHECK_ARG(ws && ws_bytes >= movae_bn_ws_bytes(rows, c), "movae_bn_act_fwd: workspace too small");
        unsigned* counter = static_cast<unsigned*>(ws);
        double* part = reinterpret_cast<double*>(static_cast<char*>(ws) + MOVAE_WS_HEADER_BYTES);
        int nblk;
        if (c % 4 == 0 && al16(y)) {
            const Split4 s = make_split4(rows, c);
            hipLaunchKernelGGL(bn_stats_partial4, dim3(s.nblk), dim3(256), 0, st, y, part, rows, c, s.CQB, s.rows_per_block,
                               in_launch_final() ? counter : nullptr, eps, momentum, save_mean, save_rstd, running_mean,
                               running_var, nbt);
            MOVAE_CHECK_LAUNCH("bn_stats_partial4");
            nblk = in_launch_final() ? 0 : s.nblk;  // 0: statistics finished in-launch by the last block
        } else {
            const Split s = make_split(rows, c);
            nblk = s.nblk;
            hipLaunchKernelGGL(bn_stats_partial, dim3(s.nblk), dim3(256), 0, st, y, part, rows, c, s.CB, s.rows_per_block);
        }
        MOVAE_CHECK_LAUNCH("bn_stats_partial");
        if (nblk > 0) {
            hipLaunchKernelGGL(bn_stats_final, dim3(c), dim3(64), 0, st, part, nblk, rows, c, eps, momentum, save_mean,
                               save_rstd, running_mean, running_var, nbt);
            MOVAE_CHECK_LAUNCH("bn_stats_final");
        }
    } else {
        MOVAE_CHECK_ARG(running_mean && running_var, "movae_bn_act_fwd: eval mode needs running statistics");
        hipLaunchKernelGGL(bn_eval_prepare, dim3(ceil_div(c, 128)), dim3(128), 0, st, running_mean, running_var, eps,
                           save_mean, save_rstd, c);
        MOVAE_CHECK_LAUNCH("bn_eval_prepare");
    }
    const long total = (long)rows * c;
    const bool vec = (c % 4 == 0) && al16(y, out) && al16(gamma, beta) && al16(save_mean, save_rstd);
    if (vec)
        hipLaunchKernelGGL(bn_apply<true>, dim3(grid_for(total / 16)), dim3(256), 0, st, y, gamma, beta, save_mean, save_rstd,
                           out, total, c, act, slope);
    else
        hipLaunchKernelGGL(bn_apply<false>, dim3(grid_for(total)), dim3(256), 0, st, y, gamma, beta, save_mean, save_rstd,
                           out, total, c, act, slope);
    MOVAE_CHECK_LAUNCH("bn_apply");
    return MOVAE_OK;
}

int movae_bn_act_bwd_grouped(int groups, const float* dout, const float* y, const float* gamma, const float* beta,
                             const float* save_mean, const float* save_rstd, float* dy, float* const* dgamma,
                             float* const* dbeta, int rows, int c, int act, float slope, int accumulate, void* ws,
                             size_t ws_bytes, movae_stream_t stream) {
    MOVAE_CHECK_ARG(dout && y && gamma && beta && save_mean && save_rstd && dy, "movae_bn_act_bwd: null pointer");
    MOVAE_CHECK_ARG(rows > 0 && c > 0, "movae_bn_act_bwd: bad shape rows=%d c=%d", rows, c);
    MOVAE_CHECK_ARG(groups >= 1 && groups <= MAX_GROUPS, "movae_bn_act_bwd: groups=%d out of range", groups);
    hipStream_t st = (hipStream_t)stream;
    BnOut tab;
    for (int g = 0; g < MAX_GROUPS; ++g) {
        tab.dgamma[g] = (g < groups && dgamma) ? dgamma[g] : nullptr;
        tab.dbeta[g] = (g < groups && dbeta) ? dbeta[g] : nullptr;
    }
    const long total = (long)rows * c;
    const bool vec = c % 4 == 0 && al16(dout, y, dy) && al16(gamma, beta) && al16(save_mean, save_rstd);
    if (vec && rows <= small_rows()) {
        hipLaunchKernelGGL(bn_bwd_small, dim3(c / 4, groups), dim3(256), 0, st, dout, y, gamma, beta, save_mean, save_rstd, dy, tab,
                           rows, c, act, slope, accumulate);
        MOVAE_CHECK_LAUNCH("bn_bwd_small");
        return MOVAE_OK;
    }
    MOVAE_CHECK_ARG(ws && ws_bytes >= MOVAE_WS_HEADER_BYTES + (size_t)groups * (movae_bn_ws_bytes(rows, c) - MOVAE_WS_HEADER_BYTES),
                    "movae_bn_act_bwd: workspace too small");
    unsigned* counter = static_cast<unsigned*>(ws);
    double* part = reinterpret_cast<double*>(static_cast<char*>(ws) + MOVAE_WS_HEADER_BYTES);
    if (vec) {
        const Split4 s = make_split4(rows, c, groups == 1 ? 1024 : 512);
        const bool fold = in_launch_final() && groups == 1;
        float* sums = reinterpret_cast<float*>(part + (size_t)groups * s.nblk * c * 2);
        hipLaunchKernelGGL(bn_bwd_partial4, dim3(s.nblk, groups), dim3(256), 0, st, dout, y, gamma, beta, save_mean, save_rstd, part,
                           rows, c, s.CQB, s.rows_per_block, act, slope, fold ? counter : nullptr, sums, tab, accumulate, total);
        MOVAE_CHECK_LAUNCH("bn_bwd_partial");
        if (!fold) {
            hipLaunchKernelGGL(bn_bwd_final, dim3(c, groups), dim3(64), 0, st, part, s.nblk, rows, c, sums, tab, accumulate);
            MOVAE_CHECK_LAUNCH("bn_bwd_final");
        }
        hipLaunchKernelGGL(bn_bwd_apply4, dim3(grid_for(total / 16), groups), dim3(256), 0, st, dout, y, gamma, beta, save_mean,
                           save_rstd, sums, dy, total, c, act, slope);
        MOVAE_CHECK_LAUNCH("bn_bwd_apply");
        return MOVAE_OK;
    }
    // generic path (C % 4 != 0 or unaligned operands): one group at a time
    const Split s = make_split(rows, c);
    float* sums = reinterpret_cast<float*>(part + (size_t)s.nblk * c * 2);
    for (int g = 0; g < groups; ++g) {
        BnOut one = tab;
        one.dgamma[0] = tab.dgamma[g];
        one.dbeta[0] = tab.dbeta[g];
        hipLaunchKernelGGL(bn_bwd_partial, dim3(s.nblk), dim3(256), 0, st, dout + g * total, y, gamma, beta, save_mean, save_rstd,
                           part, rows, c, s.CB, s.rows_per_block, act, slope);
        MOVAE_CHECK_LAUNCH("bn_bwd_partial");
        hipLaunchKernelGGL(bn_bwd_final, dim3(c, 1), dim3(64), 0, st, part, s.nblk, rows, c, sums, one, accumulate);
        MOVAE_CHECK_LAUNCH("bn_bwd_final");
        hipLaunchKernelGGL(bn_bwd_apply, dim3(grid_for(total)), dim3(256), 0, st, dout + g * total, y, gamma, beta, save_mean,
                           save_rstd, sums, dy + g * total, total, c, act, slope);
        MOVAE_CHECK_LAUNCH("bn_bwd_apply");
    }
    return MOVAE_OK;
}

int movae_bn_act_bwd(const float* dout, const float* y, const float* gamma, const float* beta, const float* save_mean,
                     const float* save_rstd, float* dy, float* dgamma, float* dbeta, int rows, int c, int act, float slope,
                     int accumulate, void* ws, size_t ws_bytes, movae_stream_t stream) {
    float* dg[1] = {dgamma};
    float* db[1] = {dbeta};
    return movae_bn_act_bwd_grouped(1, dout, y, gamma, beta, save_mean, save_rstd, dy, dg, db, rows, c, act, slope, accumulate, ws,
                                    ws_bytes, stream);
}

int movae_bn_finalize(const float* stats, size_t stats_cap, int parts, int rows, int c, const float* gamma, const float* beta, float eps,
                      float momentum, float* save_mean, float* save_rstd, float* scale, float* shift, float* running_mean,
                      float* running_var, long long* num_batches_tracked, movae_stream_t stream) {
    MOVAE_CHECK_ARG(stats && gamma && beta && save_mean && save_rstd && scale && shift, "movae_bn_finalize: null pointer");
    MOVAE_CHECK_ARG(parts > 0 && rows > 0 && c > 0, "movae_bn_finalize: bad shape parts=%d rows=%d c=%d", parts, rows, c);
    stats = fold_partials(stats, stats_cap, &parts, 1, c, (hipStream_t)stream);
    hipLaunchKernelGGL(bn_finalize_k, dim3(c), dim3(256), 0, (hipStream_t)stream, stats, parts, rows, c, gamma, beta, eps, momentum,
                       save_mean, save_rstd, scale, shift, running_mean, running_var, num_batches_tracked);
    MOVAE_CHECK_LAUNCH("bn_finalize");
    return MOVAE_OK;
}

int movae_scale_shift_act(const float* y, const float* scale, const float* shift, float* out, size_t rows, int c, float slope,
                          movae_stream_t stream) {
    MOVAE_CHECK_ARG(y && scale && shift && out && rows > 0 && c > 0, "movae_scale_shift_act: bad argument");
    MOVAE_CHECK_ARG(c % 4 == 0 && al16(y, out) && al16(scale, shift), "movae_scale_shift_act: needs c %% 4 == 0 and 16-byte aligned operands");
    const long nv = (long)rows * c / 4;
    hipLaunchKernelGGL(scale_shift_act_k, dim3(grid_for(nv / 4 + 1)), dim3(256), 0, (hipStream_t)stream, y, scale, shift, out, nv, c, slope);
    MOVAE_CHECK_LAUNCH("scale_shift_act");
    return MOVAE_OK;
}

int movae_bn_bwd_finalize(const float* bn_part, size_t bn_cap, int ppg, int groups, int rows, int c, const float* gamma,
                          const float* save_mean, const float* save_rstd, float* const* dgamma, float* const* dbeta, float* coef,
                          int accumulate, movae_stream_t stream) {
    MOVAE_CHECK_ARG(bn_part && gamma && save_mean && save_rstd && coef, "movae_bn_bwd_finalize: null pointer");
    MOVAE_CHECK_ARG(ppg > 0 && rows > 0 && c > 0 && groups >= 1 && groups <= MAX_GROUPS, "movae_bn_bwd_finalize: bad shape");
    bn_part = fold_partials(bn_part, bn_cap, &ppg, groups, c, (hipStream_t)stream);
    BnOut tab;
    for (int g = 0; g < MAX_GROUPS; ++g) {
        tab.dgamma[g] = (g < groups && dgamma) ? dgamma[g] : nullptr;
        tab.dbeta[g] = (g < groups && dbeta) ? dbeta[g] : nullptr;
    }
    hipLaunchKernelGGL(bn_bwd_finalize_k, dim3(c, groups), dim3(256), 0, (hipStream_t)stream, bn_part, ppg, rows, c, gamma, save_mean,
                       save_rstd, tab, coef, accumulate);
    MOVAE_CHECK_LAUNCH("bn_bwd_finalize");
    return MOVAE_OK;
}

int movae_bn_bwd_apply(const float* dout, const float* y, const float* scale, const float* shift, float slope, const float* coef,
                       float* dy, int groups, size_t rows, int c, movae_stream_t stream) {
    MOVAE_CHECK_ARG(dout && y && scale && shift && coef && dy && rows > 0 && c > 0 && groups >= 1, "movae_bn_bwd_apply: bad argument");
    MOVAE_CHECK_ARG(c % 4 == 0 && al16(dout, y, dy) && al16(scale, shift, coef), "movae_bn_bwd_apply: needs c %% 4 == 0 and aligned operands");
    const long nv = (long)rows * c / 4;
    hipLaunchKernelGGL(bn_bwd_apply_coef_k, dim3(grid_for(nv / 4 + 1), groups), dim3(256), 0, (hipStream_t)stream, dout, y, scale, shift, slope,
                       coef, dy, nv, c);
    MOVAE_CHECK_LAUNCH("bn_bwd_apply");
    return MOVAE_OK;
}

int movae_bn_bwd_finalize_apply(const float* bn_part, size_t bn_cap, int ppg, int groups, size_t rows, int c, const float* gamma,
                                const float* save_mean, const float* save_rstd, float* const* dgamma, float* const* dbeta, float* coef,
                                int accumulate, const float* dout, const float* y, const float* scale, const float* shift, float slope,
                                float* dy, movae_stream_t stream) {
    MOVAE_CHECK_ARG(bn_part && gamma && save_mean && save_rstd && coef && dout && y && scale && shift && dy,
                    "movae_bn_bwd_finalize_apply: null pointer");
    MOVAE_CHECK_ARG(ppg > 0 && rows > 0 && rows <= 0x7fffffffUL && c > 0 && groups >= 1 && groups <= MAX_GROUPS,
                    "movae_bn_bwd_finalize_apply: bad shape");
    MOVAE_CHECK_ARG(c % 4 == 0 && al16(dout, y, dy) && al16(scale, shift, coef), "movae_bn_bwd_finalize_apply: needs c %% 4 == 0 and aligned operands");
    static const int fuse_below = getenv("MOVAE_BN_BWD_FUSE_PARTS") ? atoi(getenv("MOVAE_BN_BWD_FUSE_PARTS")) : 256;
    if (ppg > fuse_below) {  // many partials: every block folding them again costs more than the finalize launch
        if (int rc = movae_bn_bwd_finalize(bn_part, bn_cap, ppg, groups, (int)rows, c, gamma, save_mean, save_rstd, dgamma, dbeta, coef,
                                           accumulate, stream))
            return rc;
        return movae_bn_bwd_apply(dout, y, scale, shift, slope, coef, dy, groups, rows, c, stream);
    }
    BnOut tab;
    for (int g = 0; g < MAX_GROUPS; ++g) {
        tab.dgamma[g] = (g < groups && dgamma) ? dgamma[g] : nullptr;
        tab.dbeta[g] = (g < groups && dbeta) ? dbeta[g] : nullptr;
    }
    const int slices = (c + 31) / 32;
    // about 2048 blocks over (row chunks, channel slices, groups), chunks of at least 128 rows (four per lane group)
    long chunks = 2048 / ((long)slices * groups);
    if (chunks < 1) chunks = 1;
    long rpc = ((long)rows + chunks - 1) / chunks;
    if (rpc < 128) rpc = 128;
    chunks = ((long)rows + rpc - 1) / rpc;
    hipLaunchKernelGGL(bn_bwd_fused_k, dim3((unsigned)chunks, slices, groups), dim3(256), 0, (hipStream_t)stream, bn_part, ppg, (int)rows, c,
                       gamma, save_mean, save_rstd, tab, accumulate, dout, y, scale, shift, slope, dy, (int)rpc);
    MOVAE_CHECK_LAUNCH("bn_bwd_finalize_apply");
    return MOVAE_OK;
}

}  // extern "C"
