// Sobel edge losses of the gradient-guided VAE (models/gg_vae.py:42-53,125-156): NHWC images [n][h][w][c].
//   edge-weighted pixel loss  mean( wgt[n][h][w] * (recons - inputs)^2 ),  wgt = max_c |sobel(inputs)| / (global max + EPS)
//   edge matching loss (v1)   smooth_l1( |sobel(recons)|, |sobel(inputs)| )        |sobel| = sqrt(gx^2 + gy^2 + EPS)
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
    gx = (v[0][2] - v[0][0]) + 2.f * (v[1][2] - v[1][0]) + (v[2][2] - v[2][0]);
    gy = (v[2][0] - v[0][0]) + 2.f * (v[2][1] - v[0][1]) + (v[2][2] - v[0][2]);
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

// part[block] = sum smooth_l1(|sobel r| - |sobel x|), beta = 1 (F.smooth_l1_loss default)
__global__ __launch_bounds__(256) void edge_match_partial(const float* __restrict__ r, const float* __restrict__ x,
                                                          double* __restrict__ part, long n, int H, int W, int C) {
    __shared__ double sh[4];
    const long stride = (long)gridDim.x * blockDim.x;
    double s = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        long off;
        int yy, xx, c;
        elem_coords(i, H, W, C, off, yy, xx, c);
        float rx, ry, tx, ty;
        sobel_at(r + off, H, W, C, yy, xx, c, rx, ry);
        sobel_at(x + off, H, W, C, yy, xx, c, tx, ty);
        const float d = sqrtf(rx * rx + ry * ry + EDGE_EPS) - sqrtf(tx * tx + ty * ty + EDGE_EPS);
        const float ad = fabsf(d);
        s += (double)(ad < 1.f ? 0.5f * d * d : ad - 0.5f);
    }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// stage 1: a[i] = dL/d gx(i), b[i] = dL/d gy(i) of the recons-side Sobel responses
__global__ void edge_match_bwd1(const float* __restrict__ r, const float* __restrict__ x, const float* __restrict__ gs,
                                float* __restrict__ a, float* __restrict__ b, long n, int H, int W, int C, float factor) {
    const float f = factor * (gs ? gs[0] : 1.f);
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        long off;
        int yy, xx, c;
        elem_coords(i, H, W, C, off, yy, xx, c);
        float rx, ry, tx, ty;
        sobel_at(r + off, H, W, C, yy, xx, c, rx, ry);
        sobel_at(x + off, H, W, C, yy, xx, c, tx, ty);
        const float gp = sqrtf(rx * rx + ry * ry + EDGE_EPS);
        const float d = gp - sqrtf(tx * tx + ty * ty + EDGE_EPS);
        const float ds = fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f);
        const float coef = f * ds / gp;
        a[i] = coef * rx;
        b[i] = coef * ry;
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

int movae_edge_match_fwd(const float* recons, const float* inputs, float* out, int n, int h, int w, int c, float scale, void* ws,
                         size_t ws_bytes, movae_stream_t stream) {
    MOVAE_WS_SCRATCH(ws, ws_bytes);
    MOVAE_CHECK_ARG(recons && inputs && out && n > 0 && h > 0 && w > 0 && c > 0, "movae_edge_match_fwd: bad argument");
    const long total = (long)n * h * w * c;
    const int nb = red_blocks(total);
    MOVAE_CHECK_ARG(ws && ws_bytes >= (size_t)nb * sizeof(double), "movae_edge_match_fwd: workspace too small");
    double* part = static_cast<double*>(ws);
    hipLaunchKernelGGL(edge_match_partial, dim3(nb), dim3(256), 0, (hipStream_t)stream, recons, inputs, part, total, h, w, c);
    MOVAE_CHECK_LAUNCH("edge_match_partial");
    hipLaunchKernelGGL(final_sum_e, dim3(1), dim3(256), 0, (hipStream_t)stream, part, nb, (double)scale / (double)total, out);
    MOVAE_CHECK_LAUNCH("final_sum");
    return MOVAE_OK;
}

int movae_edge_match_bwd(const float* recons, const float* inputs, const float* gscale_dev, float* drecons, float* tmp_a,
                         float* tmp_b, int n, int h, int w, int c, float scale, movae_stream_t stream) {
    MOVAE_CHECK_ARG(recons && inputs && drecons && tmp_a && tmp_b && n > 0 && h > 0 && w > 0 && c > 0, "movae_edge_match_bwd: bad argument");
    const long total = (long)n * h * w * c;
    hipLaunchKernelGGL(edge_match_bwd1, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, recons, inputs, gscale_dev, tmp_a,
                       tmp_b, total, h, w, c, scale / (float)total);
    MOVAE_CHECK_LAUNCH("edge_match_bwd1");
    hipLaunchKernelGGL(edge_match_bwd2, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, tmp_a, tmp_b, drecons, total, h, w, c);
    MOVAE_CHECK_LAUNCH("edge_match_bwd2");
    return MOVAE_OK;
}

}  // extern "C"
