// Sobel edge losses of the gradient-guided VAE (models/gg_vae.py:42-53,125-156): NHWC images [n][h][w][c].
//   edge-weighted pixel loss  mean( wgt[n][h][w] * (recons - inputs)^2 ),  wgt = max_c |sobel(inputs)| / (global max + EPS)
//   edge matching losses      smooth_l1( |sobel(recons)|, |sobel(inputs)| ) and its variants (enum movae_edge_match),
//                             |sobel| = sqrt(gx^2 + gy^2 + EPS)
// The reference runs four depthwise F.conv2d calls plus ~10 element-wise ATen kernels per loss; here every pass is one
// or two fused HBM-bound launches over the 3-channel images (fp64 block partials, deterministic).
#include "common.h"

namespace {

constexpr float EDGE_EPS = 1e-8f;  // models/gg_vae.py:8

inline int red_blocks(size_t n) {
    size_t b = (n + 1023) / 1024;
    return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}

// depthwise 3x3 cross-correlation with zero padding 1 (F.conv2d(x, sobel, padding=1, groups=C)) at (y, x, c)
__device__ __forceinline__ void sobel_at(const float* __restrict__ img, int H, int W, int C, int y, int x, int c, float& gx, float& gy) {
    float v[3][3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int yy = y + dy - 1, xx = x + dx - 1;
            v[dy][dx] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? img[((long)yy * W + xx) * C + c] : 0.f;
        }
    // explicit fma: the same rounding in every kernel that calls this (see mag())
    gx = fmaf(2.f, v[1][2] - v[1][0], v[0][2] - v[0][0]) + (v[2][2] - v[2][0]);
    gy = fmaf(2.f, v[2][1] - v[0][1], v[2][0] - v[0][0]) + (v[2][2] - v[0][2]);
}

__device__ __forceinline__ double block_max_256(double v, double* sh) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    v = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
    __syncthreads();
    return v;
}

// w_raw[pixel] = max_c sqrt(gx^2 + gy^2 + EPS); part[block] = max over the block's pixels
__global__ __launch_bounds__(256) void edge_weights_k(const float* __restrict__ x, float* __restrict__ w_raw, double* __restrict__ part,
                                                      int N, int H, int W, int C) {
    __shared__ double sh[4];
    const long npx = (long)N * H * W, stride = (long)gridDim.x * blockDim.x;
    double mx = 0.0;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npx; p += stride) {
        const int xx = (int)(p % W), yy = (int)((p / W) % H);
        const float* img = x + (p / ((long)H * W)) * (long)H * W * C;
        float m = 0.f;
        for (int c = 0; c < C; ++c) {
            float gx, gy;
            sobel_at(img, H, W, C, yy, xx, c, gx, gy);
            m = fmaxf(m, sqrtf(gx * gx + gy * gy + EDGE_EPS));
        }
        w_raw[p] = m;
        mx = fmax(mx, (double)m);
    }
    mx = block_max_256(mx, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = mx;
}

__global__ __launch_bounds__(256) void final_max(const double* __restrict__ part, int nblk, float* __restrict__ out) {
    __shared__ double sh[4];
    double m = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) m = fmax(m, part[i]);
    m = block_max_256(m, sh);
    if (threadIdx.x == 0) out[0] = (float)m;
}

__global__ __launch_bounds__(256) void final_sum_e(const double* __restrict__ part, int nblk, double factor, float* __restrict__ out) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) s += part[i];
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) out[0] = (float)(s * factor);
}

// part[block] = sum over elements of wgt * (r - x)^2, wgt = w_raw[pixel] / (wmax + EPS)
__global__ __launch_bounds__(256) void edge_wmse_partial(const float* __restrict__ r, const float* __restrict__ x,
                                                         const float* __restrict__ w_raw, const float* __restrict__ wmax,
                                                         double* __restrict__ part, long n, int C) {
    __shared__ double sh[4];
    const float inv = 1.f / (wmax[0] + EDGE_EPS);
    const long stride = (long)gridDim.x * blockDim.x;
    double s = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float d = r[i] - x[i], wg = w_raw[i / C] * inv;
        s += (double)(wg * (d * d));
    }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ void edge_wmse_bwd_k(const float* __restrict__ r, const float* __restrict__ x, const float* __restrict__ w_raw,
                                const float* __restrict__ wmax, const float* __restrict__ gs, float* __restrict__ dr, long n, int C,
                                float factor) {
    const float f = factor * (gs ? gs[0] : 1.f) / (wmax[0] + EDGE_EPS);
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dr[i] = f * w_raw[i / C] * 2.f * (r[i] - x[i]);
}

__device__ __forceinline__ void elem_coords(long i, int H, int W, int C, long& img_off, int& y, int& x, int& c) {
    c = (int)(i % C);
    const long p = i / C;
    x = (int)(p % W);
    y = (int)((p / W) % H);
    img_off = (p / ((long)H * W)) * (long)H * W * C;
}

// ---- edge matching losses: every variant is a point-wise function of the four Sobel responses ------------------
// (rx, ry) = sobel(recons), (tx, ty) = sobel(inputs) at one element, plus (for two variants) batch-global scalars:
//   st[0] = max |sobel recons|, st[1] = max |sobel inputs|, st[2] = mean |sobel inputs|     (stats pre-pass)
//   st[3] = sum_i s_i * gp_i,   st[4] = number of elements attaining st[0]                   (forward main pass, MAXNORM)
enum {
    EM_MAG = MOVAE_EDGE_MAG, EM_SIGNED_MSE = MOVAE_EDGE_SIGNED_MSE, EM_MAXNORM = MOVAE_EDGE_MAXNORM, EM_ANGLE = MOVAE_EDGE_ANGLE,
    EM_MASKED = MOVAE_EDGE_MASKED, EM_COSINE = MOVAE_EDGE_COSINE
};

// one evaluation order for the magnitude in every kernel: MAXNORM compares gp == max bit-for-bit across launches
__device__ __forceinline__ float mag(float gx, float gy) { return sqrtf(fmaf(gx, gx, fmaf(gy, gy, EDGE_EPS))); }
__device__ __forceinline__ float sl1(float d) {
    const float ad = fabsf(d);
    return ad < 1.f ? 0.5f * d * d : ad - 0.5f;
}
__device__ __forceinline__ float sl1_grad(float d) { return fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f); }

// F.normalize(p=2, eps=1e-12) followed by F.cosine_similarity's own clamp (eps=1e-8) on a 2-vector
__device__ __forceinline__ void unit2(float x, float y, float& ux, float& uy, float& n1, float& d1, float& n2, float& d2) {
    n1 = sqrtf(x * x + y * y);
    d1 = fmaxf(n1, 1e-12f);
    ux = x / d1;
    uy = y / d1;
    n2 = sqrtf(ux * ux + uy * uy);
    d2 = fmaxf(n2, 1e-8f);
}

template <int MODE>
__device__ __forceinline__ float em_value(float rx, float ry, float tx, float ty, const float* __restrict__ st, double& dot, double& ties) {
    if (MODE == EM_MAG) return sl1(mag(rx, ry) - mag(tx, ty));
    if (MODE == EM_SIGNED_MSE) return (rx - tx) * (rx - tx) + (ry - ty) * (ry - ty);
    if (MODE == EM_MAXNORM) {
        const float gp = mag(rx, ry), d = gp / (st[0] + EDGE_EPS) - mag(tx, ty) / (st[1] + EDGE_EPS);
        dot += (double)(sl1_grad(d) * gp);
        ties += gp == st[0] ? 1.0 : 0.0;
        return sl1(d);
    }
    if (MODE == EM_ANGLE) return sl1(atan2f(ry, rx) - atan2f(ty, tx));
    if (MODE == EM_MASKED) {
        const float gt = mag(tx, ty);
        return gt > st[2] ? sl1(mag(rx, ry) - gt) : 0.f;
    }
    float ux, uy, vx, vy, n1, d1, n2, d2, m1, e1, m2, e2;  // EM_COSINE: the cosine itself; the launcher turns the mean into 1 - mean
    unit2(rx, ry, ux, uy, n1, d1, n2, d2);
    unit2(tx, ty, vx, vy, m1, e1, m2, e2);
    return (ux / d2) * (vx / e2) + (uy / d2) * (vy / e2);
}

// (a, b) = f * d value / d (rx, ry)
template <int MODE>
__device__ __forceinline__ void em_grad(float rx, float ry, float tx, float ty, const float* __restrict__ st, float f, float& a, float& b) {
    if (MODE == EM_MAG || MODE == EM_MASKED || MODE == EM_MAXNORM) {
        const float gp = mag(rx, ry), gt = mag(tx, ty);
        float dgp;
        if (MODE == EM_MAG) dgp = f * sl1_grad(gp - gt);
        if (MODE == EM_MASKED) dgp = gt > st[2] ? f * sl1_grad(gp - gt) : 0.f;
        if (MODE == EM_MAXNORM) {
            const float ip = 1.f / (st[0] + EDGE_EPS);
            dgp = f * sl1_grad(gp * ip - gt / (st[1] + EDGE_EPS)) * ip;
            if (gp == st[0]) dgp -= f * st[3] * ip * ip / st[4];  // Tensor.max() spreads its gradient evenly over ties
        }
        a = dgp * rx / gp;
        b = dgp * ry / gp;
    } else if (MODE == EM_SIGNED_MSE) {
        a = f * 2.f * (rx - tx);
        b = f * 2.f * (ry - ty);
    } else if (MODE == EM_ANGLE) {
        const float g = f * sl1_grad(atan2f(ry, rx) - atan2f(ty, tx));
        const float recip = 1.f / (rx * rx + ry * ry);  // atan2_backward: NaN at (0, 0), as in the reference
        a = g * -ry * recip;
        b = g * rx * recip;
    } else {  // EM_COSINE
        float ux, uy, vx, vy, n1, d1, n2, d2, m1, e1, m2, e2;
        unit2(rx, ry, ux, uy, n1, d1, n2, d2);
        unit2(tx, ty, vx, vy, m1, e1, m2, e2);
        const float hx = f * vx / e2, hy = f * vy / e2;                       // d / d u_hat
        const float k2 = n2 > 1e-8f ? (hx * ux + hy * uy) / (d2 * d2 * n2) : 0.f;
        const float gux = hx / d2 - k2 * ux, guy = hy / d2 - k2 * uy;       // d / d u
        const float k1 = n1 > 1e-12f ? (gux * rx + guy * ry) / (d1 * d1 * n1) : 0.f;
        a = gux / d1 - k1 * rx;
        b = guy / d1 - k1 * ry;
    }
}

// stats pre-pass: part[3 * block + {0, 1, 2}] = {max gp, max gt, sum gt}
__global__ __launch_bounds__(256) void edge_match_stats_k(const float* __restrict__ r, const float* __restrict__ x,
                                                          double* __restrict__ part, long n, int H, int W, int C) {
    __shared__ double sh[4];
    const long stride = (long)gridDim.x * blockDim.x;
    double mp = 0.0, mt = 0.0, sum = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        long off;
        int yy, xx, c;
        elem_coords(i, H, W, C, off, yy, xx, c);
        float rx, ry, tx, ty;
        sobel_at(r + off, H, W, C, yy, xx, c, rx, ry);
        sobel_at(x + off, H, W, C, yy, xx, c, tx, ty);
        const float gt = mag(tx, ty);
        mp = fmax(mp, (double)mag(rx, ry));
        mt = fmax(mt, (double)gt);
        sum += (double)gt;
    }
    mp = block_max_256(mp, sh);
    mt = block_max_256(mt, sh);
    sum = block_sum_256(sum, sh);
    if (threadIdx.x == 0) {
        part[3 * blockIdx.x] = mp;
        part[3 * blockIdx.x + 1] = mt;
        part[3 * blockIdx.x + 2] = sum;
    }
}

__global__ __launch_bounds__(256) void edge_match_stats_final(const double* __restrict__ part, int nblk, double inv_n, float* __restrict__ st) {
    __shared__ double sh[4];
    double mp = 0.0, mt = 0.0, sum = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) {
        mp = fmax(mp, part[3 * i]);
        mt = fmax(mt, part[3 * i + 1]);
        sum += part[3 * i + 2];
    }
    mp = block_max_256(mp, sh);
    mt = block_max_256(mt, sh);
    sum = block_sum_256(sum, sh);
    if (threadIdx.x == 0) {
        st[0] = (float)mp;
        st[1] = (float)mt;
        st[2] = (float)(sum * inv_n);
    }
}

// part[3 * block + {0, 1, 2}] = {sum value, sum s * gp, ties}
template <int MODE>
__global__ __launch_bounds__(256) void edge_match_partial(const float* __restrict__ r, const float* __restrict__ x,
                                                          const float* __restrict__ st, double* __restrict__ part, long n, int H, int W,
                                                          int C) {
    __shared__ double sh[4];
    const long stride = (long)gridDim.x * blockDim.x;
    double s = 0.0, dot = 0.0, ties = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        long off;
        int yy, xx, c;
        elem_coords(i, H, W, C, off, yy, xx, c);
        float rx, ry, tx, ty;
        sobel_at(r + off, H, W, C, yy, xx, c, rx, ry);
        sobel_at(x + off, H, W, C, yy, xx, c, tx, ty);
        s += (double)em_value<MODE>(rx, ry, tx, ty, st, dot, ties);
    }
    s = block_sum_256(s, sh);
    if (MODE == EM_MAXNORM) {
        dot = block_sum_256(dot, sh);
        ties = block_sum_256(ties, sh);
    }
    if (threadIdx.x == 0) {
        part[3 * blockIdx.x] = s;
        part[3 * blockIdx.x + 1] = dot;
        part[3 * blockIdx.x + 2] = ties;
    }
}

// out = scale * (offset + factor * sum value); st[3], st[4] = the MAXNORM backward's two sums
__global__ __launch_bounds__(256) void edge_match_final(const double* __restrict__ part, int nblk, double offset, double factor, double scale,
                                                        float* __restrict__ out, float* __restrict__ st) {
    __shared__ double sh[4];
    double s = 0.0, dot = 0.0, ties = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) {
        s += part[3 * i];
        dot += part[3 * i + 1];
        ties += part[3 * i + 2];
    }
    s = block_sum_256(s, sh);
    dot = block_sum_256(dot, sh);
    ties = block_sum_256(ties, sh);
    if (threadIdx.x == 0) {
        out[0] = (float)(scale * (offset + factor * s));
        if (st) {
            st[3] = (float)dot;
            st[4] = (float)ties;
        }
    }
}

// stage 1: a[i] = dL/d gx(i), b[i] = dL/d gy(i) of the recons-side Sobel responses
template <int MODE>
__global__ void edge_match_bwd1(const float* __restrict__ r, const float* __restrict__ x, const float* __restrict__ gs,
                                const float* __restrict__ st, float* __restrict__ a, float* __restrict__ b, long n, int H, int W, int C,
                                float factor) {
    const float f = factor * (gs ? gs[0] : 1.f);
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        long off;
        int yy, xx, c;
        elem_coords(i, H, W, C, off, yy, xx, c);
        float rx, ry, tx, ty;
        sobel_at(r + off, H, W, C, yy, xx, c, rx, ry);
        sobel_at(x + off, H, W, C, yy, xx, c, tx, ty);
        float ga, gb;
        em_grad<MODE>(rx, ry, tx, ty, st, f, ga, gb);
        a[i] = ga;
        b[i] = gb;
    }
}

// stage 2: d recons[q] = sum_p a[p] * sx[q - p] + b[p] * sy[q - p]  (the transpose of the depthwise correlation)
__global__ void edge_match_bwd2(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ dr, long n, int H, int W,
                                int C) {
    const float sx[3][3] = {{-1.f, 0.f, 1.f}, {-2.f, 0.f, 2.f}, {-1.f, 0.f, 1.f}};
    const float sy[3][3] = {{-1.f, -2.f, -1.f}, {0.f, 0.f, 0.f}, {1.f, 2.f, 1.f}};
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        long off;
        int yy, xx, c;
        elem_coords(i, H, W, C, off, yy, xx, c);
        float acc = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                // response at p = q - (dy - 1, dx - 1) read input q with tap (dy, dx)
                const int py = yy - (dy - 1), px = xx - (dx - 1);
                if (py >= 0 && py < H && px >= 0 && px < W) {
                    const long j = off + ((long)py * W + px) * C + c;
                    acc += a[j] * sx[dy][dx] + b[j] * sy[dy][dx];
                }
            }
        dr[i] = acc;
    }
}

inline int grid_for(long total) {
    long g = (total + 255) / 256;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" {

int movae_edge_weights(const float* inputs, float* w_raw, float* wmax, int n, int h, int w, int c, void* ws, size_t ws_bytes,
                       movae_stream_t stream) {
    MOVAE_WS_SCRATCH(ws, ws_bytes);
    MOVAE_CHECK_ARG(inputs && w_raw && wmax && n > 0 && h > 0 && w > 0 && c > 0, "movae_edge_weights: bad argument");
    const long npx = (long)n * h * w;
    const int nb = red_blocks(npx);
    MOVAE_CHECK_ARG(ws && ws_bytes >= (size_t)nb * sizeof(double), "movae_edge_weights: workspace too small");
    double* part = static_cast<double*>(ws);
    hipLaunchKernelGGL(edge_weights_k, dim3(nb), dim3(256), 0, (hipStream_t)stream, inputs, w_raw, part, n, h, w, c);
    MOVAE_CHECK_LAUNCH("edge_weights");
    hipLaunchKernelGGL(final_max, dim3(1), dim3(256), 0, (hipStream_t)stream, part, nb, wmax);
    MOVAE_CHECK_LAUNCH("final_max");
    return MOVAE_OK;
}

int movae_edge_weighted_mse_fwd(const float* recons, const float* inputs, const float* w_raw, const float* wmax, float* out, int n,
                                int h, int w, int c, float scale, void* ws, size_t ws_bytes, movae_stream_t stream) {
    MOVAE_WS_SCRATCH(ws, ws_bytes);
    MOVAE_CHECK_ARG(recons && inputs && w_raw && wmax && out && n > 0 && h > 0 && w > 0 && c > 0, "movae_edge_weighted_mse_fwd: bad argument");
    const long total = (long)n * h * w * c;
    const int nb = red_blocks(total);
    MOVAE_CHECK_ARG(ws && ws_bytes >= (size_t)nb * sizeof(double), "movae_edge_weighted_mse_fwd: workspace too small");
    double* part = static_cast<double*>(ws);
    hipLaunchKernelGGL(edge_wmse_partial, dim3(nb), dim3(256), 0, (hipStream_t)stream, recons, inputs, w_raw, wmax, part, total, c);
    MOVAE_CHECK_LAUNCH("edge_wmse_partial");
    hipLaunchKernelGGL(final_sum_e, dim3(1), dim3(256), 0, (hipStream_t)stream, part, nb, (double)scale / (double)total, out);
    MOVAE_CHECK_LAUNCH("final_sum");
    return MOVAE_OK;
}

int movae_edge_weighted_mse_bwd(const float* recons, const float* inputs, const float* w_raw, const float* wmax,
                                const float* gscale_dev, float* drecons, int n, int h, int w, int c, float scale,
                                movae_stream_t stream) {
    MOVAE_CHECK_ARG(recons && inputs && w_raw && wmax && drecons && n > 0 && h > 0 && w > 0 && c > 0, "movae_edge_weighted_mse_bwd: bad argument");
    const long total = (long)n * h * w * c;
    hipLaunchKernelGGL(edge_wmse_bwd_k, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, recons, inputs, w_raw, wmax,
                       gscale_dev, drecons, total, c, scale / (float)total);
    MOVAE_CHECK_LAUNCH("edge_wmse_bwd");
    return MOVAE_OK;
}

#define EM_DISPATCH(mode, CALL)                \
    switch (mode) {                            \
        case EM_MAG: CALL(EM_MAG); break;      \
        case EM_SIGNED_MSE: CALL(EM_SIGNED_MSE); break; \
        case EM_MAXNORM: CALL(EM_MAXNORM); break;       \
        case EM_ANGLE: CALL(EM_ANGLE); break;  \
        case EM_MASKED: CALL(EM_MASKED); break;\
        default: CALL(EM_COSINE); break;       \
    }

int movae_edge_match_fwd(const float* recons, const float* inputs, float* out, int n, int h, int w, int c, float scale, int mode,
                         float* stats, void* ws, size_t ws_bytes, movae_stream_t stream) {
    MOVAE_WS_SCRATCH(ws, ws_bytes);
    MOVAE_CHECK_ARG(recons && inputs && out && n > 0 && h > 0 && w > 0 && c > 0, "movae_edge_match_fwd: bad argument");
    MOVAE_CHECK_ARG(mode >= EM_MAG && mode <= EM_COSINE, "movae_edge_match_fwd: unknown mode");
    const bool need_stats = mode == EM_MAXNORM || mode == EM_MASKED;
    MOVAE_CHECK_ARG(!need_stats || stats, "movae_edge_match_fwd: this mode needs the stats[8] buffer");
    const long total = (long)n * h * w * c;
    const int nb = red_blocks(total);
    MOVAE_CHECK_ARG(ws && ws_bytes >= (size_t)nb * 3 * sizeof(double), "movae_edge_match_fwd: workspace too small");
    double* part = static_cast<double*>(ws);
    hipStream_t st = (hipStream_t)stream;
    if (need_stats) {
        hipLaunchKernelGGL(edge_match_stats_k, dim3(nb), dim3(256), 0, st, recons, inputs, part, total, h, w, c);
        MOVAE_CHECK_LAUNCH("edge_match_stats");
        hipLaunchKernelGGL(edge_match_stats_final, dim3(1), dim3(256), 0, st, part, nb, 1.0 / (double)total, stats);
        MOVAE_CHECK_LAUNCH("edge_match_stats_final");
    }
#define EM_FWD(M) hipLaunchKernelGGL(edge_match_partial<M>, dim3(nb), dim3(256), 0, st, recons, inputs, stats, part, total, h, w, c)
    EM_DISPATCH(mode, EM_FWD)
#undef EM_FWD
    MOVAE_CHECK_LAUNCH("edge_match_partial");
    const double inv = 1.0 / (double)total;  // EM_COSINE: 1 - mean(cos)
    hipLaunchKernelGGL(edge_match_final, dim3(1), dim3(256), 0, st, part, nb, mode == EM_COSINE ? 1.0 : 0.0,
                       mode == EM_COSINE ? -inv : inv, (double)scale, out, mode == EM_MAXNORM ? stats : nullptr);
    MOVAE_CHECK_LAUNCH("edge_match_final");
    return MOVAE_OK;
}

int movae_edge_match_bwd(const float* recons, const float* inputs, const float* gscale_dev, float* drecons, float* tmp_a,
                         float* tmp_b, int n, int h, int w, int c, float scale, int mode, const float* stats, movae_stream_t stream) {
    MOVAE_CHECK_ARG(recons && inputs && drecons && tmp_a && tmp_b && n > 0 && h > 0 && w > 0 && c > 0, "movae_edge_match_bwd: bad argument");
    MOVAE_CHECK_ARG(mode >= EM_MAG && mode <= EM_COSINE, "movae_edge_match_bwd: unknown mode");
    MOVAE_CHECK_ARG(!(mode == EM_MAXNORM || mode == EM_MASKED) || stats, "movae_edge_match_bwd: this mode needs the forward's stats[8]");
    const long total = (long)n * h * w * c;
    const float factor = (mode == EM_COSINE ? -scale : scale) / (float)total;
    hipStream_t st = (hipStream_t)stream;
#define EM_BWD(M) \
    hipLaunchKernelGGL(edge_match_bwd1<M>, dim3(grid_for(total)), dim3(256), 0, st, recons, inputs, gscale_dev, stats, tmp_a, tmp_b, total, h, w, c, factor)
    EM_DISPATCH(mode, EM_BWD)
#undef EM_BWD
    MOVAE_CHECK_LAUNCH("edge_match_bwd1");
    hipLaunchKernelGGL(edge_match_bwd2, dim3(grid_for(total)), dim3(256), 0, st, tmp_a, tmp_b, drecons, total, h, w, c);
    MOVAE_CHECK_LAUNCH("edge_match_bwd2");
    return MOVAE_OK;
}

}  // extern "C"
