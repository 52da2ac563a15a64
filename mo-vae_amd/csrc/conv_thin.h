// conv_thin.h -- direct (VALU) kernels for the 3-channel image ends of the networks (included inside
// conv_igemm.hip's anonymous namespace).
//
// A 32-wide MFMA tile wastes 29/32 of its rows or columns when one side of the contraction has 3
// channels (first encoder conv, last decoder conv and their gradients), and those layers touch the largest
// activation tensors of the step, so they are HBM-bound: each kernel below streams its big operand once
// with one output pixel per thread, keeps the tiny weight matrix in LDS (broadcast ds_read_b128) and the
// per-pixel accumulators in registers.
//   thin_in  : reduction channels Cr <= 4, NN (32|64) outputs per thread         conv1 fwd, last-conv dgrad
//   thin_out : Nn <= 4 outputs, reduction channels Cr % 4 == 0                    last-conv fwd
//   thin_wgrad: one side of dW has <= 4 channels; (tap, wide-channel) per thread, pixels split over
//               blocks, deterministic slab reduce                                 conv1 / last-conv wgrad
#pragma once

namespace thin {

// ---- thin reduction side ---------------------------------------------------------------------------
template <int NN, bool BWD>
__global__ __launch_bounds__(256) void thin_in_k(const float* __restrict__ X, const float* __restrict__ W,
                                                 const float* __restrict__ bias, float* __restrict__ Y, Geom g, int M, int act,
                                                 float slope) {
    extern __shared__ __attribute__((aligned(16))) float Wl[];  // [K][NN]
    const int t = threadIdx.x;
    const int taps = g.KH * g.KW, K = taps * g.Cr, N = g.Nn;
    const int n0 = blockIdx.y * NN;
    for (int idx = t; idx < K * NN; idx += 256) {
        const int k = idx / NN, nl = idx - k * NN, n = n0 + nl;
        const int tap = k / g.Cr, c = k - tap * g.Cr;
        float v = 0.f;
        if (n < N) v = BWD ? W[((long)c * taps + tap) * N + n] : W[((long)n * taps + tap) * g.Cr + c];
        Wl[idx] = v;
    }
    __syncthreads();
    const int p = blockIdx.x * 256 + t;
    if (p >= M) return;
    const int hw = g.Ho * g.Wo;
    const int img = p / hw, rem = p - img * hw;
    const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
    float acc[NN];
#pragma unroll
    for (int n = 0; n < NN; ++n) acc[n] = (bias && n0 + n < N) ? bias[n0 + n] : 0.f;
    const float* xb = X + (long)img * g.Hi * g.Wi * g.Cr;
    for (int kh = 0; kh < g.KH; ++kh) {
        int h;
        if (BWD) {
            const int hh = ho + g.pad - kh;
            if (hh < 0 || hh % g.stride) continue;
            h = hh / g.stride;
        } else {
            h = ho * g.stride - g.pad + kh;
        }
        if (h < 0 || h >= g.Hi) continue;
        for (int kw = 0; kw < g.KW; ++kw) {
            int w;
            if (BWD) {
                const int ww = wo + g.pad - kw;
                if (ww < 0 || ww % g.stride) continue;
                w = ww / g.stride;
            } else {
                w = wo * g.stride - g.pad + kw;
            }
            if (w < 0 || w >= g.Wi) continue;
            const float* px = xb + ((long)h * g.Wi + w) * g.Cr;
            const float* wl = Wl + (kh * g.KW + kw) * g.Cr * NN;
            for (int c = 0; c < g.Cr; ++c) {
                const float v = px[c];
#pragma unroll
                for (int q = 0; q < NN / 4; ++q) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wl + c * NN + q * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[q * 4 + e] += v * w4[e];
                }
            }
        }
    }
    float* yo = Y + (long)p * N + n0;
    if (N % 4 == 0 && n0 + NN <= N) {
#pragma unroll
        for (int q = 0; q < NN / 4; ++q) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = apply_act(acc[q * 4 + e], act, slope);
            *reinterpret_cast<f32x4*>(yo + q * 4) = o;
        }
    } else {
#pragma unroll
        for (int n = 0; n < NN; ++n)
            if (n0 + n < N) yo[n] = apply_act(acc[n], act, slope);
    }
}

// ---- thin output side (FWD gather) -----------------------------------------------------------------
template <int NO>
__global__ __launch_bounds__(256) void thin_out_fwd_k(const float* __restrict__ X, const float* __restrict__ W,
                                                      const float* __restrict__ bias, float* __restrict__ Y, Geom g, int M,
                                                      int act, float slope) {
    extern __shared__ __attribute__((aligned(16))) float Wl[];  // [NO][K]
    const int t = threadIdx.x;
    const int K = g.KH * g.KW * g.Cr, N = g.Nn;
    for (int idx = t; idx < NO * K; idx += 256) {
        const int n = idx / K;
        Wl[idx] = n < N ? W[idx] : 0.f;  // W is [n][tap][c] == [n][K]
    }
    __syncthreads();
    const int p = blockIdx.x * 256 + t;
    if (p >= M) return;
    const int hw = g.Ho * g.Wo;
    const int img = p / hw, rem = p - img * hw;
    const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
    float acc[NO];
#pragma unroll
    for (int n = 0; n < NO; ++n) acc[n] = (bias && n < N) ? bias[n] : 0.f;
    const float* xb = X + (long)img * g.Hi * g.Wi * g.Cr;
    const int cq = g.Cr / 4;
    for (int kh = 0; kh < g.KH; ++kh) {
        const int h = ho * g.stride - g.pad + kh;
        if (h < 0 || h >= g.Hi) continue;
        for (int kw = 0; kw < g.KW; ++kw) {
            const int w = wo * g.stride - g.pad + kw;
            if (w < 0 || w >= g.Wi) continue;
            const f32x4* px = reinterpret_cast<const f32x4*>(xb + ((long)h * g.Wi + w) * g.Cr);
            const int kb = (kh * g.KW + kw) * g.Cr;
#pragma unroll 4
            for (int q = 0; q < cq; ++q) {
                const f32x4 x4 = px[q];
#pragma unroll
                for (int n = 0; n < NO; ++n) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(Wl + n * K + kb + q * 4);
                    acc[n] += x4[0] * w4[0] + x4[1] * w4[1] + x4[2] * w4[2] + x4[3] * w4[3];
                }
            }
        }
    }
#pragma unroll
    for (int n = 0; n < NO; ++n)
        if (n < N) Y[(long)p * N + n] = apply_act(acc[n], act, slope);
}

// ---- thin weight gradient ----------------------------------------------------------------------------
// dW[a][tap][b] = sum_p S[p][a] * Bg[p*s - pad + tap][b].  THIN_SMALL: Cs <= 4 (accumulators over a, lanes
// over b); else Cb <= 4 (accumulators over b, lanes over a).  grid = (pixel chunks, taps); block = WL wide
// lanes x PL pixel lanes.
template <bool THIN_SMALL>
__global__ __launch_bounds__(256) void thin_wgrad_k(const float* __restrict__ S, const float* __restrict__ Bg,
                                                    float* __restrict__ slab, WGeom g, int K, int chunk, int WL) {
    __shared__ float sh[4 * 256];
    const int t = threadIdx.x;
    const int PL = 256 / WL;
    const int wl = t % WL, pl = t / WL;
    const int tap = blockIdx.y, kh = tap / g.KW, kw = tap - kh * g.KW;
    const int wide = THIN_SMALL ? g.Cb : g.Cs, thinc = THIN_SMALL ? g.Cs : g.Cb;
    const int hw = g.Hs * g.Ws;
    const int p0 = blockIdx.x * chunk, p1 = min(K, p0 + chunk);
    const int N = g.KH * g.KW * g.Cb;
    for (int wb = 0; wb < wide; wb += WL) {
        const int wc = wb + wl;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (wc < wide) {
            for (int p = p0 + pl; p < p1; p += PL) {
                const int img = p / hw, rem = p - img * hw;
                const int hs = rem / g.Ws, ws = rem - hs * g.Ws;
                const int h = hs * g.stride - g.pad + kh, w = ws * g.stride - g.pad + kw;
                if (h < 0 || h >= g.Hb || w < 0 || w >= g.Wb) continue;
                const float* sp = S + (long)p * g.Cs;
                const float* bp = Bg + (((long)img * g.Hb + h) * g.Wb + w) * g.Cb;
                if (THIN_SMALL) {
                    const float bv = bp[wc];
                    _Pragma("unroll") for (int a = 0; a < 4; ++a) if (a < thinc) acc[a] += sp[a] * bv;
                } else {
                    const float sv = sp[wc];
                    _Pragma("unroll") for (int a = 0; a < 4; ++a) if (a < thinc) acc[a] += sv * bp[a];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) sh[q * 256 + t] = acc[q];
        __syncthreads();
        if (pl == 0 && wc < wide) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q >= thinc) break;
                float s = acc[q];
                for (int i = 1; i < PL; ++i) s += sh[q * 256 + i * WL + wl];
                const long o = THIN_SMALL ? ((long)q * N + tap * g.Cb + wc) : ((long)wc * N + tap * g.Cb + q);
                slab[(long)blockIdx.x * g.Cs * N + o] = s;
            }
        }
        __syncthreads();
    }
}

// LDS-tiled variant: one block = one TH x TW tile of small-side pixels of one image; both operand tiles
// (small tile, big tile with its kernel halo, zero filled outside the image) are staged once and every
// (tap, wide-channel) work item sweeps the tile from LDS, so the big operand is read from HBM once instead
// of once per tap.  Output: one slab row per block, reduced deterministically afterwards.
template <bool THIN_SMALL, int TC>  // TC = channel count of the thin side (1..4)
__global__ void thin_wgrad_tiled_k(const float* __restrict__ S, const float* __restrict__ Bg, float* __restrict__ slab,
                                   WGeom g, int TH, int TW, int tiles_h, int tiles_w) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x, nth = blockDim.x;
    const int BH = (TH - 1) * g.stride + g.KH, BW = (TW - 1) * g.stride + g.KW;
    float* smallT = lds;                        // [TH*TW][Cs]
    float* bigT = lds + TH * TW * g.Cs;         // [BH*BW][Cb]
    int b = blockIdx.x;
    const int tw = b % tiles_w;
    b /= tiles_w;
    const int th = b % tiles_h, img = b / tiles_h;
    const int hs0 = th * TH, ws0 = tw * TW;
    const float* Sb = S + (long)img * g.Hs * g.Ws * g.Cs;
    const float* Bb = Bg + (long)img * g.Hb * g.Wb * g.Cb;
    for (int idx = t; idx < TH * TW * g.Cs; idx += nth) {
        const int pix = idx / g.Cs, c = idx - pix * g.Cs;
        const int py = pix / TW, px = pix - py * TW;
        const int hs = hs0 + py, ws = ws0 + px;
        smallT[idx] = (hs < g.Hs && ws < g.Ws) ? Sb[((long)hs * g.Ws + ws) * g.Cs + c] : 0.f;
    }
    const int hb0 = hs0 * g.stride - g.pad, wb0 = ws0 * g.stride - g.pad;
    for (int idx = t; idx < BH * BW * g.Cb; idx += nth) {
        const int pix = idx / g.Cb, c = idx - pix * g.Cb;
        const int r = pix / BW, cc = pix - r * BW;
        const int h = hb0 + r, w = wb0 + cc;
        bigT[idx] = (h >= 0 && h < g.Hb && w >= 0 && w < g.Wb) ? Bb[((long)h * g.Wb + w) * g.Cb + c] : 0.f;
    }
    __syncthreads();
    const int wide = THIN_SMALL ? g.Cb : g.Cs;
    const int taps = g.KH * g.KW, N = taps * g.Cb;
    float* out = slab + (long)blockIdx.x * g.Cs * N;
    for (int item = t; item < taps * wide; item += nth) {
        const int tap = item / wide, wl = item - tap * wide;
        const int kh = tap / g.KW, kw = tap - kh * g.KW;
        float acc[TC];
#pragma unroll
        for (int a = 0; a < TC; ++a) acc[a] = 0.f;
        const int sstep = TC, bstep = g.stride * g.Cb;
        for (int py = 0; py < TH; ++py) {
            const float* sp = smallT + py * TW * g.Cs;
            const float* bp = bigT + ((py * g.stride + kh) * BW + kw) * g.Cb;
            int px = 0;
            for (; px + 4 <= TW; px += 4) {  // 4 pixels per trip: all LDS reads issued before the FMAs
                float wv[4], tv[4][TC];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (THIN_SMALL) {
                        wv[u] = bp[(px + u) * bstep + wl];
#pragma unroll
                        for (int a = 0; a < TC; ++a) tv[u][a] = sp[(px + u) * sstep + a];
                    } else {
                        wv[u] = sp[(px + u) * g.Cs + wl];
#pragma unroll
                        for (int a = 0; a < TC; ++a) tv[u][a] = bp[(px + u) * bstep + a];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int a = 0; a < TC; ++a) acc[a] += tv[u][a] * wv[u];
            }
            for (; px < TW; ++px) {
                if (THIN_SMALL) {
                    const float xv = bp[px * bstep + wl];
#pragma unroll
                    for (int a = 0; a < TC; ++a) acc[a] += sp[px * sstep + a] * xv;
                } else {
                    const float sv = sp[px * g.Cs + wl];
#pragma unroll
                    for (int a = 0; a < TC; ++a) acc[a] += sv * bp[px * bstep + a];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < TC; ++q) {
            const long o = THIN_SMALL ? ((long)q * N + tap * g.Cb + wl) : ((long)wl * N + tap * g.Cb + q);
            out[o] = acc[q];
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------
inline bool thin_in_ok(const Geom& g) { return g.Cr <= 4 && g.Nn >= 8 && g.KH * g.KW * g.Cr * 64 * 4 <= 48 * 1024; }

template <bool BWD>
int launch_thin_in(const float* X, const float* W, float* Y, const Geom& g, const Epilogue& ep, hipStream_t st) {
    const long Ml = (long)g.Nimg * g.Ho * g.Wo;
    const int M = (int)Ml, K = g.KH * g.KW * g.Cr;
    if (g.Nn > 32) {
        dim3 grid(ceil_div(M, 256), ceil_div(g.Nn, 64));
        hipLaunchKernelGGL((thin_in_k<64, BWD>), grid, dim3(256), (size_t)K * 64 * sizeof(float), st, X, W, ep.bias, Y, g, M,
                           ep.act, ep.slope);
    } else {
        dim3 grid(ceil_div(M, 256), 1);
        hipLaunchKernelGGL((thin_in_k<32, BWD>), grid, dim3(256), (size_t)K * 32 * sizeof(float), st, X, W, ep.bias, Y, g, M,
                           ep.act, ep.slope);
    }
    MOVAE_CHECK_LAUNCH("thin_in");
    return MOVAE_OK;
}

inline bool thin_out_ok(const Geom& g, const float* X) {
    return g.Nn <= 4 && g.Cr % 4 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 &&
           (size_t)4 * g.KH * g.KW * g.Cr * sizeof(float) <= 60 * 1024;
}

int launch_thin_out_fwd(const float* X, const float* W, float* Y, const Geom& g, const Epilogue& ep, hipStream_t st) {
    const int M = g.Nimg * g.Ho * g.Wo, K = g.KH * g.KW * g.Cr;
    hipLaunchKernelGGL((thin_out_fwd_k<4>), dim3(ceil_div(M, 256)), dim3(256), (size_t)4 * K * sizeof(float), st, X, W, ep.bias,
                       Y, g, M, ep.act, ep.slope);
    MOVAE_CHECK_LAUNCH("thin_out_fwd");
    return MOVAE_OK;
}

inline bool thin_wgrad_ok(const WGeom& g) { return g.Cs <= 4 || g.Cb <= 4; }

int launch_thin_wgrad(const float* S, const float* Bg, float* dW, const WGeom& g, int K, int accumulate, void* ws,
                      size_t ws_bytes, hipStream_t st) {
    const int M = g.Cs, N = g.KH * g.KW * g.Cb;
    const bool thin_small = g.Cs <= 4;
    const int wide = thin_small ? g.Cb : g.Cs;
    {   // preferred: LDS-tiled kernel (tile sized to <= 48 KiB of LDS)
        int TW = g.Ws < 32 ? g.Ws : 32, TH = g.Hs < 16 ? g.Hs : 16;
        auto lds_floats = [&](int th, int tw) {
            return (long)th * tw * g.Cs + (long)((th - 1) * g.stride + g.KH) * ((tw - 1) * g.stride + g.KW) * g.Cb;
        };
        while (lds_floats(TH, TW) * 4 > 48 * 1024 && (TH > 1 || TW > 1)) {
            if (TH >= TW && TH > 1) TH = (TH + 1) / 2;
            else TW = (TW + 1) / 2;
        }
        const int tiles_h = ceil_div(g.Hs, TH), tiles_w = ceil_div(g.Ws, TW);
        const long nblk = (long)g.Nimg * tiles_h * tiles_w;
        const size_t per1 = (size_t)M * N * sizeof(float);
        if (lds_floats(TH, TW) * 4 <= 48 * 1024 && nblk <= 65535L * 16 && ws && per1 * (size_t)nblk <= ws_bytes) {
            const int items = g.KH * g.KW * wide;
            int nth = ceil_div(items, 64) * 64;
            if (nth > 1024) nth = 1024;
            float* slab = static_cast<float*>(ws);
            const size_t shb = (size_t)lds_floats(TH, TW) * sizeof(float);
            const int tc = thin_small ? g.Cs : g.Cb;
#define MOVAE_TW(SM, TCV)                                                                                               \
    hipLaunchKernelGGL((thin_wgrad_tiled_k<SM, TCV>), dim3((unsigned)nblk), dim3(nth), shb, st, S, Bg, slab, g, TH, TW, \
                       tiles_h, tiles_w)
            if (thin_small) {
                if (tc == 1) MOVAE_TW(true, 1); else if (tc == 2) MOVAE_TW(true, 2); else if (tc == 3) MOVAE_TW(true, 3); else MOVAE_TW(true, 4);
            } else {
                if (tc == 1) MOVAE_TW(false, 1); else if (tc == 2) MOVAE_TW(false, 2); else if (tc == 3) MOVAE_TW(false, 3); else MOVAE_TW(false, 4);
            }
#undef MOVAE_TW
            MOVAE_CHECK_LAUNCH("thin_wgrad_tiled");
            return launch_reduce(slab, dW, (long)M * N, (int)nblk, N, nullptr, 0, 0.f, accumulate, st);
        }
    }
    int WL = 1;  // smallest power of two >= wide, capped at the block size
    while (WL < wide && WL < 256) WL *= 2;
    const int PL = 256 / WL;
    int nch = ceil_div(K, PL * 16);
    if (nch > 1024) nch = 1024;
    const size_t per = (size_t)M * N * sizeof(float);
    while (nch > 1 && per * nch > ws_bytes) nch /= 2;
    if (!ws || per * nch > ws_bytes) {
        movae_set_error("thin wgrad: workspace too small");
        return MOVAE_EINVAL;
    }
    const int chunk = ceil_div(K, nch);
    nch = ceil_div(K, chunk);
    float* slab = static_cast<float*>(ws);
    dim3 grid(nch, g.KH * g.KW);
    if (thin_small)
        hipLaunchKernelGGL((thin_wgrad_k<true>), grid, dim3(256), 0, st, S, Bg, slab, g, K, chunk, WL);
    else
        hipLaunchKernelGGL((thin_wgrad_k<false>), grid, dim3(256), 0, st, S, Bg, slab, g, K, chunk, WL);
    MOVAE_CHECK_LAUNCH("thin_wgrad");
    return launch_reduce(slab, dW, (long)M * N, nch, N, nullptr, 0, 0.f, accumulate, st);
}

}  // namespace thin
