// conv_thin.h -- kernels for the 3-channel image ends of the networks (included inside conv_igemm.hip's anonymous namespace).
//
// Used as a plain GEMM dimension, 3 channels waste 29/32 of a 32-wide MFMA tile (first encoder conv, last decoder conv / transposed
// conv and their gradients), and those layers touch the largest activation tensors of the step, so they are HBM-bound.  Two
// generations live here:
//  * MFMA forms (round 2, preferred): the thin side becomes a GEMM dimension of (tap, channel) pairs -- 27 or 48 columns --
//      thin_in_mfma_k    3 reduction channels -> 32 k outputs: K = taps * 3, input rows staged once in LDS     conv1 fwd, last-conv dgrad
//      thin_out_mfma_k   C -> 3, 3x3 stride 1: P[input pixel][tap, co] = X W, then a 27-term gather per output pixel   last-conv fwd
//      thin_outT_mfma_k  C -> 3, 4x4 stride 2 transposed: the same with 48 columns and a 2 x 2 parity gather   VQ last layer fwd
//      thin_wgrad_mfma_k dW[c][tap, j] = sum over pixels, two pixels per MFMA, persistent blocks              conv1 / last-conv wgrad
//  * VALU forms (round 1; the shapes the MFMA forms do not take): one output pixel per thread, the tiny weight matrix in LDS
//    (broadcast ds_read_b128 -- whose return bandwidth bounds them), per-pixel accumulators in registers --
//      thin_in_k, thin_out_fwd_k / thin_out_tile_k, thin_wgrad_k / thin_wgrad_tiled_k / thin_wgrad_sweep_k
#pragma once

namespace thin {

// Output side of the 32-output thin-input kernels: the block's 256 pixels x 32 outputs sit in LDS as Wl[pixel * 33 + n] (activation
// applied) and are one contiguous 32 KiB run of Y.  BatchNorm statistics / backward sums of the tile, the ActMul factor, then stores of
// 1 KiB per wave instead of 64 scattered 16-byte pieces.
__device__ __forceinline__ void thin_in_tile_out(float* __restrict__ Wl, float* __restrict__ Y, int M, int t, float* __restrict__ stats,
                                                 const BnBwd& bb, const ActMul& am, int N, int n0, int bx) {
    // bx: the block's index among the 256-pixel blocks (bx, or its place in a paired launch)
    // N > 32: the tile is columns n0 .. n0 + 31 of 256 rows of N floats (128-byte row segments; statistics / backward sums: N == 32 only)
    float* yb = Y + (long)bx * 256 * N + n0;
    const long rows_left = (long)M - (long)bx * 256;
    if (stats) {
        // column statistics of the block's 256 x 32 tile for the BatchNorm that follows (the tile sits in LDS anyway):
        // thread t sums channel t & 31 over rows 32 * (t >> 5) ..+31 (stride 33: conflict-free), the two row groups of a
        // wave fold with one shuffle; one partial pair per wave: stats[((4 * block + wave) * 2 + {0,1}) * 32 + c]
        const int c = t & 31, r0 = (t >> 5) * 32;
        float sm = 0.f, sq = 0.f;
        for (int r = 0; r < 32; ++r)
            if (r0 + r < rows_left) {
                const float v = Wl[(r0 + r) * 33 + c];
                sm += v;
                sq = fmaf(v, v, sq);
            }
        sm += __shfl_xor(sm, 32, 64);
        sq += __shfl_xor(sq, 32, 64);
        if ((t & 63) < 32) {
            const long pidx = (long)bx * 4 + (t >> 6);
            stats[(pidx * 2 + 0) * 32 + c] = sm;
            stats[(pidx * 2 + 1) * 32 + c] = sq;
        }
    }
    if (bb.y) {
        // the tile is `dout` of a fused BatchNorm over y (same shape as Y): its backward sums, same thread mapping; a
        // block's 256 rows never straddle two cotangent groups (host).  Each load instruction covers two whole 128-byte rows.
        const int c = t & 31, r0 = (t >> 5) * 32;
        const float sc = bb.scale[c], sh_ = bb.shift[c];
        const float* yb_ = bb.y + ((long)bx * 256 % bb.rows_per_group) * 32;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int rb = 0; rb < 32; rb += 16) {  // sixteen loads of y in flight (one per trip paid an L2 round trip per row: 13 us at C2)
            float yv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) yv[u] = r0 + rb + u < rows_left ? yb_[(r0 + rb + u) * 32 + c] : 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (r0 + rb + u < rows_left) {
                    const float z = fmaf(yv[u], sc, sh_);
                    const float d = Wl[(r0 + rb + u) * 33 + c] * (z > 0.f ? 1.f : bb.slope);
                    s1 += d;
                    s2 = fmaf(d, yv[u], s2);
                }
        }
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if ((t & 63) < 32) {
            const long pidx = (long)bx * 4 + (t >> 6);
            bb.part[(pidx * 2 + 0) * 32 + c] = s1;
            bb.part[(pidx * 2 + 1) * 32 + c] = s2;
        }
    }
    if (actmul_on(am)) {
        // the result is the cotangent of an activation output / a residual block's branch (ActMul): factor act'(y) and the
        // identity cotangent on the way out, 16-byte pieces, all loads of a thread ahead of its stores
        // (two halves of four pieces: sixteen 16-byte registers for y and res at once would cost the kernels a wave of occupancy)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 y4[4], r4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = h * 4 + j;
                const int idx = i * 256 + t, px = idx >> 3, q = idx & 7;
                const long o = ((long)bx * 256 + px) * N + n0 + q * 4;
                const bool ok = px < rows_left;
                y4[j] = (ok && am.y) ? *reinterpret_cast<const f32x4*>(am.y + o % am.per_group) : f32x4{0.f, 0.f, 0.f, 0.f};
                r4[j] = (ok && am.res) ? *reinterpret_cast<const f32x4*>(am.res + o) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int idx = (h * 4 + j) * 256 + t, px = idx >> 3, q = idx & 7;
                if (px < rows_left) {
                    const float* src = Wl + px * 33 + q * 4;
                    f32x4 o4;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        o4[e] = src[e] * (am.y ? act_grad_from_out(y4[j][e], am.act, am.slope) : 1.f) + r4[j][e];
                    *reinterpret_cast<f32x4*>(yb + (long)px * N + q * 4) = o4;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = i * 256 + t, px = idx >> 3, q = idx & 7;
        if (px < rows_left) {
            const float* src = Wl + px * 33 + q * 4;
            *reinterpret_cast<f32x4*>(yb + (long)px * N + q * 4) = f32x4{src[0], src[1], src[2], src[3]};
        }
    }
}

// ---- thin reduction side ---------------------------------------------------------------------------
template <int NN, bool BWD>
__global__ __launch_bounds__(256) void thin_in_k(const float* __restrict__ X, const float* __restrict__ W,
                                                 const float* __restrict__ bias, float* __restrict__ Y, Geom g, int M, int act,
                                                 float slope, float* __restrict__ stats, BnBwd bb, ActMul am) {
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
    const bool live = p < M;
    const int pp = live ? p : 0;
    const int hw = g.Ho * g.Wo;
    const int img = pp / hw, rem = pp - img * hw;
    const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
    float acc[NN];
#pragma unroll
    for (int n = 0; n < NN; ++n) acc[n] = (bias && n0 + n < N) ? bias[n0 + n] : 0.f;
    const float* xb = X + (long)img * g.Hi * g.Wi * g.Cr;
    for (int kh = 0; live && kh < g.KH; ++kh) {
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
    if (NN == 32 && N == 32) {  // (launch_thin_in sizes the dynamic LDS for the 256 x 33 transpose tile in this case)
        // the block's 256 pixels x 32 outputs are one contiguous 32 KiB run of Y: transpose through LDS (the weight tile is
        // dead by now) so that every wave stores 1 KiB contiguous instead of 64 scattered 16-byte pieces
        __syncthreads();
#pragma unroll
        for (int n = 0; n < NN; ++n) Wl[t * 33 + n] = apply_act(acc[n], act, slope);
        __syncthreads();
        thin_in_tile_out(Wl, Y, M, t, stats, bb, am, 32, 0, blockIdx.x);
        return;
    }
    if (!live) return;
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

// ---- thin reduction side on the MFMA pipe (3 reduction channels, 32 outputs) ------------------------------------
// thin_in_k is bound by the LDS return bandwidth of its broadcast weight reads (one ds_read_b128 per four FMAs: 47 us of LDS time
// for the first layer of C5, measured 50).  As a GEMM the layer is out[pixel][n] = sum_k patch[pixel][k] * W[k][n] with K = taps * 3
// = 27 / 48: a v_mfma_f32_32x32x2_f32 step takes two k for 32 pixels, with B = W[2j + half][n = lane & 31] held in registers for the
// whole block and A = one ds_read_b32 per lane from the block's input region, staged in LDS as a plain copy of the image rows
// (3 floats per pixel: lanes are consecutive pixels, stride 3 or 6 floats): address = pixel corner + a lane-half constant per step.
// ~30x less LDS traffic than the VALU form; K / 2 MFMAs per 32 pixels.  A block owns 256 consecutive output pixels of one image --
// whole rows or a row segment (host) -- and leaves through thin_in_tile_out like thin_in_k.  BWD: stride 1 only.
struct ThinInArgs {  // thin_in_mfma_k's arguments (a paired launch carries them as one struct)
    const float* X;
    const float* W;
    const float* bias;
    float* Y;
    Geom g;
    int M, act;
    float slope;
    float* stats;
    BnBwd bb;
    ActMul am;
    int ncols, RH, RW;
    FastDiv fd_row, fd3, fd_ncols, fd_hw, fd_wo;
    int gx, gy;        // the launch's grid
    unsigned lds_bytes;
};

template <int KH, int KW, bool BWD>
__device__ __forceinline__ void thin_in_mfma_body(const float* __restrict__ X, const float* __restrict__ W,
                                                  const float* __restrict__ bias, float* __restrict__ Y, const Geom& g, int M, int act,
                                                  float slope, float* __restrict__ stats, const BnBwd& bb, const ActMul& am, int ncols, int RH,
                                                  int RW, FastDiv fd_row, FastDiv fd3, FastDiv fd_ncols, FastDiv fd_hw, FastDiv fd_wo, int bx,
                                                  int by) {
    constexpr int TC = 3, TAPS = KH * KW, K = TAPS * TC, KS = (K + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) float Wl[];  // input region [RH][RW * 3], then the output tile [256][33]
    const int t = threadIdx.x, lane = t & 63, half = lane >> 5, l31 = lane & 31, wave = t >> 6;
    const int N = g.Nn, n0 = by * 32;  // N % 32 == 0 (host): a block computes 32 of the N outputs of its 256 pixels
    const int p0 = bx * 256, hw = g.Ho * g.Wo;
    const int img = fdiv(p0, fd_hw), rem = p0 - img * hw;
    const int ho0 = fdiv(rem, fd_wo), wo0 = rem - ho0 * g.Wo;
    const int s = g.stride;
    const int ih0 = BWD ? ho0 + g.pad - (KH - 1) : ho0 * s - g.pad, iw0 = BWD ? wo0 + g.pad - (KW - 1) : wo0 * s - g.pad;
    constexpr int KP = K | 1;  // odd LDS row pitch of the staged weights: lanes are outputs, conflict-free
    float bw[KS];  // B operand: W[k = 2j + half][n = lane & 31]
    if (BWD) {     // W[c][tap][n]: the lanes of a load are neighbours in memory
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const int k = 2 * j + half, kk = k < K ? k : K - 1;
            const int tap = kk / TC, c = kk - tap * TC;
            const float w = W[((long)c * TAPS + tap) * N + n0 + l31];
            bw[j] = k < K ? w : 0.f;
        }
    }
    float* Ww = Wl + RH * RW * 3;  // FWD: [32][KP] copy of W[n][tap][c] (the lanes' own rows lie 4 * K bytes apart in memory)
    {   // the region: rows ih0 .. ih0 + RH - 1, columns iw0 .. iw0 + RW - 1 of image img, zero outside; nine loads in flight per thread
        const float* Xb = X + (long)img * g.Hi * g.Wi * TC;
        const int row_f = RW * 3, total = RH * row_f;
        float wv[BWD ? 1 : (32 * K + 255) / 256];
        if (!BWD) {
#pragma unroll
            for (int u = 0; u < (32 * K + 255) / 256; ++u) wv[u] = t + u * 256 < 32 * K ? W[(long)n0 * K + t + u * 256] : 0.f;
        }
        for (int base = t; base < total; base += 9 * 256) {
            float v[9];
#pragma unroll
            for (int u = 0; u < 9; ++u) {
                const int f = base + u * 256;
                const int r = fdiv(f, fd_row), o = f - r * row_f;
                const int ih = ih0 + r, iw = iw0 + fdiv(o, fd3);
                const bool ok = f < total && ih >= 0 && ih < g.Hi && iw >= 0 && iw < g.Wi;
                v[u] = ok ? Xb[((long)ih * g.Wi + iw0) * TC + o] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 9; ++u)
                if (base + u * 256 < total) Wl[base + u * 256] = v[u];
        }
        if (!BWD) {
#pragma unroll
            for (int u = 0; u < (32 * K + 255) / 256; ++u) {
                const int f = t + u * 256, n = f / K;
                if (f < 32 * K) Ww[n * KP + f - n * K] = wv[u];
            }
        }
    }
    __syncthreads();
    int dl[KS];
#pragma unroll
    for (int j = 0; j < KS; ++j) {
        const int k = 2 * j + half, kk = k < K ? k : K - 1;
        const int tap = kk / TC, c = kk - tap * TC, kh = tap / KW, kw = tap - kh * KW;
        if (!BWD) bw[j] = k < K ? Ww[l31 * KP + kk] : 0.f;
        dl[j] = (BWD ? -(kh * RW + kw) : kh * RW + kw) * 3 + c;
    }
    f32x16 acc[2];
    int cb[2];  // float offset of this lane's pixel corner in the region, per 32-pixel tile of the wave
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
        const int pix = wave * 64 + u * 32 + l31;
        const int py = fdiv(pix, fd_ncols), px = pix - py * ncols;
        cb[u] = BWD ? ((py + KH - 1) * RW + px + KW - 1) * 3 : (py * s * RW + px * s) * 3;
    }
#pragma unroll
    for (int j = 0; j < KS; ++j)
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(Wl[cb[u] + dl[j]], bw[j], acc[u], 0, 0, 0);
    __syncthreads();  // the region is dead: the output tile takes its place
    const float bv = bias ? bias[n0 + l31] : 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
            Wl[(wave * 64 + u * 32 + m) * 33 + l31] = apply_act(acc[u][r] + bv, act, slope);
        }
    __syncthreads();
    // (Loading the ActMul factor's source ahead of the staging / MFMA phase was measured slower: 32 more registers, three waves per SIMD,
    // 211 vs 176 us on the C5 layer.)
    thin_in_tile_out(Wl, Y, M, t, stats, bb, am, N, n0, bx);
}

template <int KH, int KW, bool BWD>
__global__ __launch_bounds__(256) void thin_in_mfma_k(const float* __restrict__ X, const float* __restrict__ W,
                                                      const float* __restrict__ bias, float* __restrict__ Y, Geom g, int M, int act,
                                                      float slope, float* __restrict__ stats, BnBwd bb, ActMul am, int ncols, int RH,
                                                      int RW, FastDiv fd_row, FastDiv fd3, FastDiv fd_ncols, FastDiv fd_hw, FastDiv fd_wo) {
    thin_in_mfma_body<KH, KW, BWD>(X, W, bias, Y, g, M, act, slope, stats, bb, am, ncols, RH, RW, fd_row, fd3, fd_ncols, fd_hw, fd_wo,
                                   blockIdx.x, blockIdx.y);
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

// ---- thin output side, LDS-tiled (FWD and BWD gather) -----------------------------------------------------------
// N <= 4 outputs per pixel from many reduction channels (last decoder layer: Conv2d -> 3 or ConvTranspose2d -> 3).  The
// layer streams the step's largest activation once; one output pixel per thread reading its taps straight from global
// memory (thin_out_fwd_k) re-fetches every input pixel KH*KW times through the L1 in 16-byte pieces.  Here a block owns a
// TH x TW tile of output pixels of one image and walks the reduction channels in chunks of CC: the input region of the
// tile (with halo, zero outside the image) and the chunk's weights are staged in LDS with coalesced 16-byte loads, then
// every thread accumulates PPT pixels x 3..4 outputs from LDS (pixel stride CC + 4 floats => conflict-free ds_read_b128;
// weight reads are wave-uniform broadcasts).  BWD (transposed conv, stride 2): wave w owns output-parity class w, so
// the tap set -- and with it every weight address -- stays uniform inside a wave.
template <bool BWD, int CC, int PPT>
__global__ __launch_bounds__(256) void thin_out_tile_k(const float* __restrict__ X, const float* __restrict__ W,
                                                       const float* __restrict__ bias, float* __restrict__ Y, Geom g, int TH,
                                                       int TW, int tiles_h, int tiles_w, int IH, int IW, int act, float slope,
                                                       Norm nrm) {
    constexpr int XS = CC + 4;  // LDS pixel stride
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xT = lds;                    // [IH*IW][XS]
    float* wT = lds + IH * IW * XS;     // [taps][4][CC]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int s = g.stride, taps = g.KH * g.KW, N = g.Nn;
    int b = blockIdx.x;
    const int tw = b % tiles_w;
    b /= tiles_w;
    const int th = b % tiles_h, img = b / tiles_h;
    const int y0 = th * TH, x0 = tw * TW;
    // input tile origin
    int iy0, ix0;
    if (BWD) {
        const int ny = y0 + g.pad - g.KH + 1, nx = x0 + g.pad - g.KW + 1;
        iy0 = ny >= 0 ? ny / s : -((-ny + s - 1) / s);
        ix0 = nx >= 0 ? nx / s : -((-nx + s - 1) / s);
    } else {
        iy0 = y0 * s - g.pad;
        ix0 = x0 * s - g.pad;
    }
    // this thread's pixels (tile-local) and, for BWD, its parity class
    int py[PPT], px[PPT];
    int kh0 = 0, kw0 = 0, nA = g.KH, nB = g.KW;
    if (BWD && s == 2) {
        const int ph = wave >> 1, pw = wave & 1;  // TH, TW even; class grid (TH/2) x (TW/2)
        const int cw = TW / 2;
#pragma unroll
        for (int u = 0; u < PPT; ++u) {
            const int j = lane + u * 64;
            py[u] = (j / cw) * 2 + ph;
            px[u] = (j % cw) * 2 + pw;
        }
        kh0 = (y0 + ph + g.pad) % 2;  // y0, x0 are even (tile sizes are even)
        kw0 = (x0 + pw + g.pad) % 2;
        nA = kh0 < g.KH ? (g.KH - kh0 + 1) / 2 : 0;
        nB = kw0 < g.KW ? (g.KW - kw0 + 1) / 2 : 0;
    } else {
#pragma unroll
        for (int u = 0; u < PPT; ++u) {
            const int j = t + u * 256;
            py[u] = j / TW;
            px[u] = j % TW;
        }
    }
    float acc[PPT][4];
#pragma unroll
    for (int u = 0; u < PPT; ++u)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[u][n] = 0.f;
    const float* Xb = X + (long)img * g.Hi * g.Wi * g.Cr;
    constexpr int QC = CC / 4;
    for (int c0 = 0; c0 < g.Cr; c0 += CC) {
        __syncthreads();
        // 4 independent 16-byte loads in flight per thread before the first LDS store: a load-store loop with one
        // load per trip pays the full memory latency every trip
        for (int base = t; base < IH * IW * QC; base += 4 * 256) {
            f32x4 v[4];
            int dst[4], nch[4];  // nch: first channel of the chunk when it holds real data under a fused input transform, else -1
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * 256;
                const int pix = idx / QC, q = idx - pix * QC;
                const int r = pix / IW, c = pix - r * IW;
                const int h = iy0 + r, w = ix0 + c, ch = c0 + q * 4;
                const bool ok = idx < IH * IW * QC && h >= 0 && h < g.Hi && w >= 0 && w < g.Wi && ch < g.Cr;
                dst[u] = idx < IH * IW * QC ? pix * XS + q * 4 : -1;
                nch[u] = (ok && nrm.scale) ? ch : -1;  // padding stays exactly zero under the transform
                v[u] = ok ? *reinterpret_cast<const f32x4*>(Xb + ((long)h * g.Wi + w) * g.Cr + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (nrm.scale) {  // virtual input (fused BatchNorm + activation of the producer): transform between load and LDS store
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (nch[u] >= 0)
                        v[u] = norm_apply(v[u], *reinterpret_cast<const f32x4*>(nrm.scale + nch[u]),
                                          *reinterpret_cast<const f32x4*>(nrm.shift + nch[u]), nrm.slope);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (dst[u] >= 0) *reinterpret_cast<f32x4*>(xT + dst[u]) = v[u];
        }
        for (int base = t; base < taps * 4 * CC; base += 4 * 256) {  // same batching for the chunk's weights
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * 256;
                const int tap = idx / (4 * CC), rem = idx - tap * 4 * CC;
                const int n = rem / CC, cl = rem - n * CC, c = c0 + cl;
                const bool ok = idx < taps * 4 * CC && n < N && c < g.Cr;
                v[u] = ok ? (BWD ? W[((long)c * taps + tap) * N + n] : W[((long)n * taps + tap) * g.Cr + c]) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (base + u * 256 < taps * 4 * CC) wT[base + u * 256] = v[u];
        }
        __syncthreads();
        for (int a = 0; a < nA; ++a)
            for (int bb = 0; bb < nB; ++bb) {
                const int kh = BWD ? kh0 + s * a : a, kw = BWD ? kw0 + s * bb : bb;
                const float* wp = wT + (kh * g.KW + kw) * 4 * CC;
                const float* xp[PPT];
#pragma unroll
                for (int u = 0; u < PPT; ++u) {
                    int ih, iw;
                    if (BWD) {
                        ih = (y0 + py[u] + g.pad - kh) / s - iy0;  // exact division by construction of the class
                        iw = (x0 + px[u] + g.pad - kw) / s - ix0;
                    } else {
                        ih = py[u] * s + kh;
                        iw = px[u] * s + kw;
                    }
                    xp[u] = xT + (ih * IW + iw) * XS;
                }
#pragma unroll
                for (int q = 0; q < QC; ++q) {
                    f32x4 x4[PPT];
#pragma unroll
                    for (int u = 0; u < PPT; ++u) x4[u] = *reinterpret_cast<const f32x4*>(xp[u] + q * 4);
#pragma unroll
                    for (int n = 0; n < 3; ++n) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wp + n * CC + q * 4);
#pragma unroll
                        for (int u = 0; u < PPT; ++u)
                            acc[u][n] += x4[u][0] * w4[0] + x4[u][1] * w4[1] + x4[u][2] * w4[2] + x4[u][3] * w4[3];
                    }
                    if (N == 4) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wp + 3 * CC + q * 4);
#pragma unroll
                        for (int u = 0; u < PPT; ++u)
                            acc[u][3] += x4[u][0] * w4[0] + x4[u][1] * w4[1] + x4[u][2] * w4[2] + x4[u][3] * w4[3];
                    }
                }
            }
    }
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
        const int ho = y0 + py[u], wo = x0 + px[u];
        if (py[u] < TH && ho < g.Ho && wo < g.Wo) {
            float* yo = Y + (((long)img * g.Ho + ho) * g.Wo + wo) * N;
#pragma unroll
            for (int n = 0; n < 4; ++n)
                if (n < N) yo[n] = apply_act(acc[u][n] + (bias ? bias[n] : 0.f), act, slope);
        }
    }
}

// ---- thin output side on the MFMA pipe (3 outputs, stride-1 KH x KW conv, reduction channels a multiple of 32) ----------------
// thin_out_tile_k is bound by the LDS return bandwidth of its operand reads (weights as broadcast ds_read_b128, 5 reads per 24 FMAs:
// ~88 us of LDS time for the last conv of C5, measured 153).  The same arithmetic as a GEMM over INPUT pixels,
//   P[q][tap * 3 + co] = sum_c X[q][c] * W[co][tap][c]                       (M = input pixels, K = Cr, N = taps * 3 = 27 <= 32)
// followed by out[y][x][co] = sum_tap P[(y - pad + kh, x - pad + kw)][tap * 3 + co]: the big operand goes from global memory straight
// into the A registers of v_mfma_f32_32x32x2_f32 (lane = input pixel, its half of the 32-channel chunk as four 16-byte loads; step j
// pairs channel j of the lower half with channel 16 + j of the upper one, B = the weights under the same pairing), P passes through
// LDS once (27 floats per input pixel) and every output pixel gathers its 27 terms: ~10x less LDS traffic.  A block owns an 8 x 32
// output tile and computes P for the tile's (8 + KH - 1) x (32 + KW - 1) input pixels in 32-pixel groups dealt to its four waves.
template <int KH, int KW, bool ONE>  // ONE: 32 reduction channels (a single chunk)
__global__ __launch_bounds__(256) void thin_out_mfma_k(const float* __restrict__ X, const float* __restrict__ W,
                                                       const float* __restrict__ bias, float* __restrict__ Y, Geom g, int tiles_h,
                                                       int tiles_w, int act, float slope, Norm nrm) {
    constexpr int TH = 8, TW = 32, RH = TH + KH - 1, RW = TW + KW - 1, NPX = RH * RW, NG = (NPX + 31) / 32;
    constexpr int TAPS = KH * KW, NC = TAPS * 3, PS = NC | 1;  // odd pitch: the gather's lanes are consecutive input pixels
    static_assert(NC <= 32, "taps * 3 columns in one MFMA tile");
    extern __shared__ __attribute__((aligned(16))) float Pl[];  // [NG * 32][PS]
    const int t = threadIdx.x, lane = t & 63, half = lane >> 5, l31 = lane & 31, wave = t >> 6;
    int b = blockIdx.x;
    const int tw = b % tiles_w;
    b /= tiles_w;
    const int th = b % tiles_h, img = b / tiles_h;
    const int y0 = th * TH, x0 = tw * TW;
    const float* Xb = X + (long)img * g.Hi * g.Wi * g.Cr + 16 * half;
    // this lane's weight row: column n = l31 -> (tap, co); W[co][tap][c]
    const int ntap = l31 / 3, nco = l31 - ntap * 3;
    const float* Wn = W + ((long)nco * TAPS + ntap) * g.Cr + 16 * half;
    const bool ncol = l31 < NC;
    if (ONE) {
        // one chunk: the wave's (up to) three groups are loaded back to back before the first MFMA -- one memory latency per block
        // instead of one per group -- and share the weight registers
        constexpr int GW = (NG + 3) / 4;
        f32x4 a[GW][4], bw[4];
        bool ok[GW];
#pragma unroll
        for (int u = 0; u < GW; ++u) {
            const int ri = (wave + 4 * u) * 32 + l31;
            const int ry = ri / RW, rx = ri - ry * RW;
            const int iy = y0 - g.pad + ry, ix = x0 - g.pad + rx;
            ok[u] = wave + 4 * u < NG && ri < NPX && iy >= 0 && iy < g.Hi && ix >= 0 && ix < g.Wi;
            const float* xp = Xb + ((long)iy * g.Wi + ix) * g.Cr;
#pragma unroll
            for (int i = 0; i < 4; ++i) a[u][i] = ok[u] ? *reinterpret_cast<const f32x4*>(xp + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) bw[i] = ncol ? *reinterpret_cast<const f32x4*>(Wn + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < GW; ++u) {
            const int grp = wave + 4 * u;
            if (grp >= NG) break;
            if (nrm.scale) {  // virtual input (fused BatchNorm + activation of the producer); padding stays exactly zero
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    a[u][i] = norm_apply_if(ok[u], a[u][i], *reinterpret_cast<const f32x4*>(nrm.scale + 16 * half + 4 * i),
                                            *reinterpret_cast<const f32x4*>(nrm.shift + 16 * half + 4 * i), nrm.slope);
            }
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i][e], bw[i][e], acc, 0, 0, 0);
            if (ncol) {
#pragma unroll
                for (int r = 0; r < 16; ++r) Pl[(grp * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * PS + l31] = acc[r];
            }
        }
    }
    // (Several chunks: loading the operands of the next item ahead of an item's MFMAs in a rolling two-deep pipeline was measured
    // slower -- 152 VGPRs, three waves per SIMD: 126 vs 110 us on the C5 layer.)
    for (int grp = wave; !ONE && grp < NG; grp += 4) {
        const int ri = grp * 32 + l31;
        const int ry = ri / RW, rx = ri - ry * RW;
        const int iy = y0 - g.pad + ry, ix = x0 - g.pad + rx;
        const bool ok = ri < NPX && iy >= 0 && iy < g.Hi && ix >= 0 && ix < g.Wi;
        const float* xp = Xb + ((long)iy * g.Wi + ix) * g.Cr;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int c0 = 0; c0 < g.Cr; c0 += 32) {
            f32x4 a[4], bw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = ok ? *reinterpret_cast<const f32x4*>(xp + c0 + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
                bw[i] = ncol ? *reinterpret_cast<const f32x4*>(Wn + c0 + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (nrm.scale) {  // virtual input (fused BatchNorm + activation of the producer); padding stays exactly zero
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    a[i] = norm_apply_if(ok, a[i], *reinterpret_cast<const f32x4*>(nrm.scale + c0 + 16 * half + 4 * i),
                                         *reinterpret_cast<const f32x4*>(nrm.shift + c0 + 16 * half + 4 * i), nrm.slope);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], bw[i][e], acc, 0, 0, 0);
        }
        if (ncol) {
#pragma unroll
            for (int r = 0; r < 16; ++r) Pl[(grp * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * PS + l31] = acc[r];
        }
    }
    __syncthreads();
    const int py = t >> 5, px = t & 31;
    const int oy = y0 + py, ox = x0 + px;
    if (oy < g.Ho && ox < g.Wo) {
        float v[3];
#pragma unroll
        for (int co = 0; co < 3; ++co) v[co] = bias ? bias[co] : 0.f;
#pragma unroll
        for (int kh = 0; kh < KH; ++kh)
#pragma unroll
            for (int kw = 0; kw < KW; ++kw) {
                const float* pp = Pl + ((py + kh) * RW + px + kw) * PS + (kh * KW + kw) * 3;
#pragma unroll
                for (int co = 0; co < 3; ++co) v[co] += pp[co];
            }
        float* yo = Y + (((long)img * g.Ho + oy) * g.Wo + ox) * 3;
#pragma unroll
        for (int co = 0; co < 3; ++co) yo[co] = apply_act(v[co], act, slope);
    }
}

// The transposed form (last layer of the VQ decoders: ConvTranspose2d C -> 3, 4x4 taps, stride 2, pad 1): the same two steps,
//   P[q][tap * 3 + co] = sum_c X[q][c] * W[c][tap][co]                      (48 columns: two MFMA column tiles, the second half used)
//   out[y][x][co] = sum over the 2 x 2 taps of y's / x's parity of P[((y + 1 - kh) / 2, (x + 1 - kw) / 2)][tap * 3 + co].
// A block owns 16 x 64 output pixels = 8 x 32 input pixels plus a one-pixel halo (10 x 34, eleven groups of 32, <= 3 per wave).  The
// weights pass through LDS once, transposed to [column][channel] (the lanes of B are columns: their channels lie 192 bytes apart in
// memory); every 32-channel chunk loads its B registers for both column tiles from there and the A registers of the wave's groups from
// global memory, then runs 2 x 16 MFMAs per group into accumulators that stay in registers over all chunks; P takes the place of the
// weights in LDS after the last chunk and each thread gathers four output pixels.
__global__ __launch_bounds__(256, 2) void thin_outT_mfma_k(const float* __restrict__ X, const float* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ Y, Geom g, int tiles_h,
                                                        int tiles_w, int act, float slope, Norm nrm) {
    constexpr int ITH = 8, ITW = 32, RH = ITH + 2, RW = ITW + 2, NPX = RH * RW, NG = (NPX + 31) / 32, GW = (NG + 3) / 4;
    constexpr int TAPS = 16, NC = TAPS * 3, PS = NC + 1;  // 48 columns, odd pitch
    extern __shared__ __attribute__((aligned(16))) float Pl[];  // W^T [NC][Cr + 4] during the MFMA phase, then P [NG * 32][PS]
    const int t = threadIdx.x, lane = t & 63, half = lane >> 5, l31 = lane & 31, wave = t >> 6;
    int b = blockIdx.x;
    const int tw = b % tiles_w;
    b /= tiles_w;
    const int th = b % tiles_h, img = b / tiles_h;
    const int a0 = th * ITH, b0 = tw * ITW;  // input tile origin; output tile origin (2 * a0, 2 * b0)
    const int Cr = g.Cr, WP = Cr + 4;
    for (int f = t; f < Cr * NC; f += 256) {  // W[c][n] -> W^T[n][c]
        const int c = f / NC, n = f - c * NC;
        Pl[n * WP + c] = W[f];
    }
    // this lane's pixels: group u of the wave -> region pixel -> image pixel
    const float* xp[GW];
    bool ok[GW];
#pragma unroll
    for (int u = 0; u < GW; ++u) {
        const int ri = (wave + 4 * u) * 32 + l31;
        const int ry = ri / RW, rx = ri - ry * RW;
        const int iy = a0 - 1 + ry, ix = b0 - 1 + rx;
        ok[u] = wave + 4 * u < NG && ri < NPX && iy >= 0 && iy < g.Hi && ix >= 0 && ix < g.Wi;
        xp[u] = X + (((long)img * g.Hi + iy) * g.Wi + ix) * Cr + 16 * half;
    }
    f32x16 acc[GW][2];
#pragma unroll
    for (int u = 0; u < GW; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][v][r] = 0.f;
    __syncthreads();
    const bool col1 = l31 < NC - 32;  // second column tile: 16 of its 32 columns exist
    for (int c0 = 0; c0 < Cr; c0 += 32) {
        f32x4 bw[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bw[0][i] = *reinterpret_cast<const f32x4*>(Pl + l31 * WP + c0 + 16 * half + 4 * i);
            bw[1][i] = col1 ? *reinterpret_cast<const f32x4*>(Pl + (32 + l31) * WP + c0 + 16 * half + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        f32x4 a[2][4];  // this group's operand and the next one's, loaded one group ahead
#pragma unroll
        for (int i = 0; i < 4; ++i) a[0][i] = ok[0] ? *reinterpret_cast<const f32x4*>(xp[0] + c0 + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < GW; ++u) {
            if (wave + 4 * u >= NG) break;
            if (u + 1 < GW) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    a[(u + 1) & 1][i] = ok[u + 1] ? *reinterpret_cast<const f32x4*>(xp[u + 1] + c0 + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (nrm.scale) {  // virtual input (fused BatchNorm + activation of the producer); padding stays exactly zero
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    a[u & 1][i] = norm_apply_if(ok[u], a[u & 1][i], *reinterpret_cast<const f32x4*>(nrm.scale + c0 + 16 * half + 4 * i),
                                                *reinterpret_cast<const f32x4*>(nrm.shift + c0 + 16 * half + 4 * i), nrm.slope);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[u][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u & 1][i][e], bw[0][i][e], acc[u][0], 0, 0, 0);
                    acc[u][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u & 1][i][e], bw[1][i][e], acc[u][1], 0, 0, 0);
                }
        }
    }
    __syncthreads();  // every wave is done with the weights: P takes their place
#pragma unroll
    for (int u = 0; u < GW; ++u) {
        const int grp = wave + 4 * u;
        if (grp >= NG) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float* row = Pl + (grp * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * PS;
            row[l31] = acc[u][0][r];
            if (col1) row[32 + l31] = acc[u][1][r];
        }
    }
    __syncthreads();
    const float b0v = bias ? bias[0] : 0.f, b1v = bias ? bias[1] : 0.f, b2v = bias ? bias[2] : 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int op = t + 256 * u, oy = op >> 6, ox = op & 63;
        const int y = 2 * a0 + oy, x = 2 * b0 + ox;
        if (y >= g.Ho || x >= g.Wo) continue;
        float v0 = b0v, v1 = b1v, v2 = b2v;
        const int ph = (oy + 1) & 1, pw = (ox + 1) & 1;  // (2 * a0, 2 * b0 are even; pad = 1)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int kh = ph + 2 * i, rr = ((oy + 1 - kh) >> 1) + 1;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int kw = pw + 2 * j, rc = ((ox + 1 - kw) >> 1) + 1;
                const float* pp = Pl + (rr * RW + rc) * PS + (kh * 4 + kw) * 3;
                v0 += pp[0], v1 += pp[1], v2 += pp[2];
            }
        }
        float* yo = Y + (((long)img * g.Ho + y) * g.Wo + x) * 3;
        yo[0] = apply_act(v0, act, slope), yo[1] = apply_act(v1, act, slope), yo[2] = apply_act(v2, act, slope);
    }
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

// Sweep variant (preferred): a thread owns ONE wide channel and ALL taps.  Per wide-side pixel it reads its channel once
// (conflict-free ds_read_b32) and, per tap, the matching thin-side pixel as one broadcast ds_read_b128 (thin pixels are
// padded to 4 floats in LDS), i.e. (1 + taps) LDS reads per taps*TC FMAs -- the tiled kernel above issues 1 + TC reads per
// TC FMAs and is LDS-issue bound.  Block = 32 channel lanes x 8 row groups over one TH x TW tile of wide-side pixels of one
// image; the row groups fold through LDS and the block writes one slab row (deterministic reduce afterwards).
//   REV = false: wide = S (small side), thin = Bg:  acc[tap][b] += S[p][a]  * Bg[p*s - pad + tap][b],  thread a
//   REV = true : wide = Bg, thin = S, stride 1:      acc[tap][a] += Bg[q][b] * S[q + pad - tap][a],     thread b
template <int TC, int KH, int KW, bool REV>
__global__ __launch_bounds__(256) void thin_wgrad_sweep_k(const float* __restrict__ Wide, const float* __restrict__ Thin,
                                                          float* __restrict__ slab, int Hw, int Ww, int Cw, int Ht, int Wt,
                                                          int stride, int pad, int TH, int TW, int tiles_h, int tiles_w,
                                                          int Cs, int Cb, long wide_gs, long thin_gs, long slab_gs, Norm nrm) {
    constexpr int TAPS = KH * KW, NA = TAPS * TC;
    Wide += blockIdx.z * wide_gs;  // cotangent group (batched pull-back): 0 for the operand the groups share
    Thin += blockIdx.z * thin_gs;
    slab += blockIdx.z * slab_gs;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x, cw = t & 31, rg = t >> 5;
    const int BH = REV ? TH + KH - 1 : (TH - 1) * stride + KH, BW = REV ? TW + KW - 1 : (TW - 1) * stride + KW;
    float* wideT = lds;                                // [TH*TW][32]; reused as the row-group fold buffer
    const int wide_floats = max(TH * TW * 32, 8 * 32 * ((NA + 1) / 2));
    f32x4* thinT = reinterpret_cast<f32x4*>(lds + wide_floats);  // [BH*BW] pixels padded to 4 channels
    int b = blockIdx.x;
    const int tw = b % tiles_w;
    b /= tiles_w;
    const int th = b % tiles_h, img = b / tiles_h;
    const int c0 = blockIdx.y * 32;
    const int y0 = th * TH, x0 = tw * TW;
    // stage the wide tile: 8 threads x 16 B cover one pixel's 32-channel slice
    {
        const float* Wb = Wide + (long)img * Hw * Ww * Cw + c0;
        const int q = t & 7;
        for (int pix = t >> 3; pix < TH * TW; pix += 32) {
            const int py = pix / TW, px = pix - py * TW;
            const int y = y0 + py, x = x0 + px;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (y < Hw && x < Ww) {
                v = *reinterpret_cast<const f32x4*>(Wb + ((long)y * Ww + x) * Cw + q * 4);
                if (nrm.scale)  // the wide operand is a virtual activation (fused BatchNorm + activation of its producer)
                    v = norm_apply(v, *reinterpret_cast<const f32x4*>(nrm.scale + c0 + q * 4),
                                   *reinterpret_cast<const f32x4*>(nrm.shift + c0 + q * 4), nrm.slope);
            }
            *reinterpret_cast<f32x4*>(wideT + pix * 32 + q * 4) = v;
        }
        // thin tile with its halo, zero outside the image
        const float* Tb = Thin + (long)img * Ht * Wt * TC;
        const int ty0 = REV ? y0 + pad - (KH - 1) : y0 * stride - pad, tx0 = REV ? x0 + pad - (KW - 1) : x0 * stride - pad;
        for (int pix = t; pix < BH * BW; pix += 256) {
            const int r = pix / BW, c = pix - r * BW;
            const int y = ty0 + r, x = tx0 + c;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (y >= 0 && y < Ht && x >= 0 && x < Wt) {
                const float* src = Tb + ((long)y * Wt + x) * TC;
#pragma unroll
                for (int j = 0; j < TC; ++j) v[j] = src[j];
            }
            thinT[pix] = v;
        }
    }
    __syncthreads();
    float acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = 0.f;
    for (int py = rg; py < TH; py += 8) {
        const float* wrow = wideT + py * TW * 32 + cw;
        const f32x4* trow = thinT + (REV ? py : py * stride) * BW;
        for (int px = 0; px < TW; ++px) {
            const float wv = wrow[px * 32];
            const f32x4* tp = trow + (REV ? px : px * stride);
#pragma unroll
            for (int kh = 0; kh < KH; ++kh)
#pragma unroll
                for (int kw = 0; kw < KW; ++kw) {
                    const f32x4 tv = tp[(REV ? KH - 1 - kh : kh) * BW + (REV ? KW - 1 - kw : kw)];
#pragma unroll
                    for (int j = 0; j < TC; ++j) acc[(kh * KW + kw) * TC + j] += wv * tv[j];
                }
        }
    }
    // fold the 8 row groups through LDS (two passes keep the buffer inside the wide tile's footprint), then one slab row
    const int N = TAPS * Cb;
    float* out = slab + (long)blockIdx.x * Cs * N;
    constexpr int HALF = (NA + 1) / 2;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < HALF; ++i)
            if (pass * HALF + i < NA) wideT[(rg * HALF + i) * 32 + cw] = acc[pass * HALF + i];
        __syncthreads();
        for (int i = rg; i < HALF && pass * HALF + i < NA; i += 8) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) v += wideT[(r * HALF + i) * 32 + cw];
            const int ai = pass * HALF + i, tap = ai / TC, j = ai - tap * TC;
            const int c = c0 + cw;
            // dW[a][tap][b]: REV -> a = j (thin), b = c (wide);  else a = c (wide), b = j (thin)
            const long o = REV ? ((long)j * TAPS + tap) * Cb + c : ((long)c * TAPS + tap) * Cb + j;
            out[o] = v;
        }
    }
}

// MFMA variant of the sweep (preferred where it applies): the weight gradient of a thin-channel layer is the GEMM
//   dW[c][tap * TC + j] = sum_pixels Wide[p][c] * ThinPatch[p][tap * TC + j]            (M = 32 wide channels, N = TAPS * TC <= 64)
// and a v_mfma_f32_32x32x2_f32 step consumes two pixels: lane (half h, l) supplies A = Wide[p + h][c = l] (conflict-free
// ds_read_b32 of the staged wide tile) and, per 32-column tile, B = the thin pixel's channel j under tap (kh, kw) with
// (tap, j) = column l -- one gathered ds_read_b32 at a lane-constant offset from the pixel's corner.  Three LDS reads per two
// MFMAs where the sweep kernel issued (1 + TAPS) reads per TAPS * TC FMAs and was bound by the LDS return bandwidth of its
// broadcast ds_read_b128 (C5 first layer, 4x4 taps: 217 us for 92 MB of operands).  Same tiling, staging, slab row per block and
// deterministic reduce as the sweep kernel; the four waves take every fourth pixel pair and fold through LDS.
struct ThinWgArgs {  // thin_wgrad_mfma_k's arguments (a paired launch carries them as one struct)
    const float* Wide;
    const float* Thin;
    float* slab;
    int Hw, Ww, Cw, Ht, Wt, stride, pad, TH, TW, tiles_h, tiles_w, Cs, Cb;
    long wide_gs, thin_gs, slab_gs;
    Norm nrm;
    int ntiles;
    FastDiv fd_tw, fd_bw;
    int cs_on;
    long row_stride;
    int gx, gy, gz;    // the launch's grid
    unsigned lds_bytes;
};

template <int TC, int KH, int KW, bool REV>
__device__ __forceinline__ void thin_wgrad_mfma_body(const float* __restrict__ Wide, const float* __restrict__ Thin,
                                                     float* __restrict__ slab, int Hw, int Ww, int Cw, int Ht, int Wt,
                                                     int stride, int pad, int TH, int TW, int tiles_h, int tiles_w,
                                                     int Cs, int Cb, long wide_gs, long thin_gs, long slab_gs, const Norm& nrm, int ntiles,
                                                     FastDiv fd_tw, FastDiv fd_bw, int cs_on, long row_stride, int bx, int by, int bz,
                                                     int gdx) {
    // cs_on (REV, the two images equally large: host): the column sums of the thin operand -- the bias gradient when it is dy --
    // ride along: the blockIdx.y == 0 blocks add up the pixels their tiles own while they stage them; TC floats behind the slab row
    constexpr int TAPS = KH * KW, NA = TAPS * TC, NT = (NA + 31) / 32;
    static_assert(NT <= 2, "at most 64 columns");
    Wide += bz * wide_gs;
    Thin += bz * thin_gs;
    slab += bz * slab_gs;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x;
    const int BH = REV ? TH + KH - 1 : (TH - 1) * stride + KH, BW = REV ? TW + KW - 1 : (TW - 1) * stride + KW;
    const int npix = TH * TW;
    float* wideT = lds;                                 // [npix + 1][32] (one zero pixel behind the tile); reused for the fold
    const int wide_floats = max((npix + 1) * 32, 4 * NT * 1024);
    float* thinF = lds + wide_floats;                   // [BH * BW] pixels padded to 4 channels
    const int c0 = by * 32;
    const int lane = t & 63, half = lane >> 5, l31 = lane & 31, wave = t >> 6;
    // column n = tile * 32 + l31 -> (tap, j): the float offset of that element from the thin pixel under the wide pixel's corner
    int delta[NT];
    float keep[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int n = u * 32 + l31, nn = n < NA ? n : 0;
        const int tap = nn / TC, j = nn - tap * TC, kh = tap / KW, kw = tap - kh * KW;
        delta[u] = ((REV ? KH - 1 - kh : kh) * BW + (REV ? KW - 1 - kw : kw)) * 4 + j;
        keep[u] = n < NA ? 1.f : 0.f;
    }
    f32x16 acc[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    const int step = REV ? 1 : stride;
    const int tw_shift = __ffs(TW) - 1;
    // Operands of a tile travel global -> registers -> LDS, and the registers of tile i+1 are loaded BEFORE the MFMA loop of tile i:
    // a block keeps WPF 16-byte pieces of the wide tile and TPF thin pixels in flight while it computes (a load-store loop with one
    // piece per trip had ~16 KB per CU in flight, i.e. ~1.3 TB/s of the ~5 the HBM delivers).  Host: TH * TW <= 256, BH * BW <= 1280.
    constexpr int WPF = 8, TPF = 5;
    f32x4 wv[WPF];
    float tv[TPF][TC];
    unsigned wok = 0;  // bit u: piece u of the wide tile lies inside the image (padding stays exactly zero under a fused transform)
    const int q = t & 7;
    auto fetch = [&](int tile) {
        int b = tile;
        const int tw = b % tiles_w;
        b /= tiles_w;
        const int th = b % tiles_h, img = b / tiles_h;
        const int y0 = th * TH, x0 = tw * TW;
        const float* Wb = Wide + (long)img * Hw * Ww * Cw + c0 + q * 4;
        wok = 0;
#pragma unroll
        for (int u = 0; u < WPF; ++u) {
            const int pix = (t >> 3) + 32 * u;
            const int py = fdiv(pix, fd_tw), px = pix - py * TW;
            const int y = y0 + py, x = x0 + px;
            const bool ok = pix < npix && y < Hw && x < Ww;
            wv[u] = ok ? *reinterpret_cast<const f32x4*>(Wb + ((long)y * Ww + x) * Cw) : f32x4{0.f, 0.f, 0.f, 0.f};
            wok |= ok ? 1u << u : 0u;
        }
        const float* Tb = Thin + (long)img * Ht * Wt * TC;
        const int ty0 = REV ? y0 + pad - (KH - 1) : y0 * stride - pad, tx0 = REV ? x0 + pad - (KW - 1) : x0 * stride - pad;
#pragma unroll
        for (int u = 0; u < TPF; ++u) {
            const int pix = t + 256 * u;
            const int r = fdiv(pix, fd_bw), c = pix - r * BW;
            const int y = ty0 + r, x = tx0 + c;
            const bool ok = pix < BH * BW && y >= 0 && y < Ht && x >= 0 && x < Wt;
            const float* src = Tb + ((long)y * Wt + x) * TC;
#pragma unroll
            for (int j = 0; j < TC; ++j) tv[u][j] = ok ? src[j] : 0.f;
        }
    };
    f32x4 nsc = f32x4{1.f, 1.f, 1.f, 1.f}, nsh = f32x4{0.f, 0.f, 0.f, 0.f};  // this thread's channel quad under a fused input transform
    if (nrm.scale) {
        nsc = *reinterpret_cast<const f32x4*>(nrm.scale + c0 + q * 4);
        nsh = *reinterpret_cast<const f32x4*>(nrm.shift + c0 + q * 4);
    }
    if (t < 8) *reinterpret_cast<f32x4*>(wideT + npix * 32 + t * 4) = f32x4{0.f, 0.f, 0.f, 0.f};  // the zero pixel behind the tile
    float csum[TC];
#pragma unroll
    for (int j = 0; j < TC; ++j) csum[j] = 0.f;
    const bool cs_here = REV && cs_on && by == 0;
    if (bx < ntiles) fetch(bx);
    // persistent: the block walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... and keeps ONE accumulator set -- one fold, one slab
    // row and one reduce input per block instead of per tile (the per-tile epilogue and the 8192-slab reduce were the cost)
    for (int tile = bx; tile < ntiles; tile += gdx) {
        __syncthreads();  // the previous tile's readers are done
#pragma unroll
        for (int u = 0; u < WPF; ++u) {
            const int pix = (t >> 3) + 32 * u;
            f32x4 v = wv[u];
            if (nrm.scale && (wok >> u & 1)) v = norm_apply(v, nsc, nsh, nrm.slope);
            if (pix < npix) *reinterpret_cast<f32x4*>(wideT + pix * 32 + q * 4) = v;
        }
        int oy0 = 0, ox0 = 0;  // cs_here: the tile's first owned thin pixel inside the staged region (rows / columns KH - 1 - pad on)
        if (cs_here) {
            int b = tile;
            const int tw = b % tiles_w;
            b /= tiles_w;
            const int th = b % tiles_h;
            oy0 = min(TH, Hw - th * TH), ox0 = min(TW, Ww - tw * TW);  // owned extent (the image may end inside the tile)
        }
#pragma unroll
        for (int u = 0; u < TPF; ++u) {
            const int pix = t + 256 * u;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < TC; ++j) v[j] = tv[u][j];
            if (pix < BH * BW) *reinterpret_cast<f32x4*>(thinF + pix * 4) = v;
            if (cs_here) {
                const int r = fdiv(pix, fd_bw), c = pix - r * BW;
                const int ry = r - (KH - 1 - pad), rx = c - (KW - 1 - pad);  // position inside the tile
                if (pix < BH * BW && ry >= 0 && ry < oy0 && rx >= 0 && rx < ox0) {
#pragma unroll
                    for (int j = 0; j < TC; ++j) csum[j] += tv[u][j];
                }
            }
        }
        if (tile + gdx < ntiles) fetch(tile + gdx);  // in flight under this tile's MFMA loop
        __syncthreads();
        // this lane's pixel of the wave's first pair; every step moves on by 8 pixels (4 waves x 2).
        // Host: tile width a power of two, tile a multiple of 32 pixels.  Four pairs per trip -- their 4 * (1 + NT) LDS reads go out back
        // to back ahead of the 4 * NT MFMAs, addresses by shift / mask / 24-bit multiply-add.  (One pair per trip with a wrap loop for
        // the row change spent ~500 cycles of issue and LDS latency per 64-cycle MFMA.)
        for (int p = 2 * wave + half; p < npix; p += 32) {
            float a[4], bv[4][NT];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int pk = p + 8 * k;
                a[k] = wideT[pk * 32 + l31];
                const int base = __mul24(__mul24(pk >> tw_shift, step), BW) + __mul24(pk & (TW - 1), step);
#pragma unroll
                for (int u = 0; u < NT; ++u) bv[k][u] = thinF[base * 4 + delta[u]] * keep[u];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int u = 0; u < NT; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], bv[k][u], acc[u], 0, 0, 0);
        }
    }
    // fold the four waves through LDS (the wide tile is dead), then this block's slab row
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) wideT[((wave * NT + u) * 16 + r) * 64 + lane] = acc[u][r];
    __syncthreads();
    const int N = TAPS * Cb;
    float* out = slab + (long)bx * row_stride;
    if (cs_here) {  // wave sums by shuffle, the four waves in fixed order
        __shared__ float csr[4][TC];
#pragma unroll
        for (int j = 0; j < TC; ++j) {
            float v = csum[j];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if (lane == 0) csr[wave][j] = v;
        }
        __syncthreads();
        if (t < TC) out[(long)Cs * N + t] = (csr[0][t] + csr[1][t]) + (csr[2][t] + csr[3][t]);
    }
    for (int idx = t; idx < NT * 1024; idx += 256) {
        const int u = idx >> 10, r = (idx >> 6) & 15, ln = idx & 63;
        const int n = u * 32 + (ln & 31);
        if (n >= NA) continue;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) v += wideT[((w * NT + u) * 16 + r) * 64 + ln];
        const int m = (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
        const int tap = n / TC, j = n - tap * TC, c = c0 + m;
        const long o = REV ? ((long)j * TAPS + tap) * Cb + c : ((long)c * TAPS + tap) * Cb + j;
        out[o] = v;
    }
}

template <int TC, int KH, int KW, bool REV>
__global__ __launch_bounds__(256) void thin_wgrad_mfma_k(const float* __restrict__ Wide, const float* __restrict__ Thin,
                                                         float* __restrict__ slab, int Hw, int Ww, int Cw, int Ht, int Wt,
                                                         int stride, int pad, int TH, int TW, int tiles_h, int tiles_w,
                                                         int Cs, int Cb, long wide_gs, long thin_gs, long slab_gs, Norm nrm, int ntiles,
                                                         FastDiv fd_tw, FastDiv fd_bw, int cs_on, long row_stride) {
    thin_wgrad_mfma_body<TC, KH, KW, REV>(Wide, Thin, slab, Hw, Ww, Cw, Ht, Wt, stride, pad, TH, TW, tiles_h, tiles_w, Cs, Cb, wide_gs,
                                          thin_gs, slab_gs, nrm, ntiles, fd_tw, fd_bw, cs_on, row_stride, blockIdx.x, blockIdx.y,
                                          blockIdx.z, gridDim.x);
}

// The last conv's backward in ONE launch (a 3-channel output: its input gradient is thin_in_mfma_k's BWD form, its weight gradient
// thin_wgrad_mfma_k): blocks [0, nw) the weight gradient's persistent blocks (the longer ones: dispatched first), the rest the
// input gradient's.  The two share read-only operands only; one dynamic LDS allocation, the larger of the two needs.
template <int K, bool REVV>
__global__ __launch_bounds__(256) void thin_pair_k(ThinInArgs d, ThinWgArgs w, int nw) {
    int b = blockIdx.x;
    if (b < nw) {
        const int bx = b % w.gx, r = b / w.gx;
        thin_wgrad_mfma_body<3, K, K, REVV>(w.Wide, w.Thin, w.slab, w.Hw, w.Ww, w.Cw, w.Ht, w.Wt, w.stride, w.pad, w.TH, w.TW, w.tiles_h,
                                            w.tiles_w, w.Cs, w.Cb, w.wide_gs, w.thin_gs, w.slab_gs, w.nrm, w.ntiles, w.fd_tw, w.fd_bw,
                                            w.cs_on, w.row_stride, bx, r % w.gy, r / w.gy, w.gx);
    } else {
        b -= nw;
        thin_in_mfma_body<K, K, true>(d.X, d.W, d.bias, d.Y, d.g, d.M, d.act, d.slope, d.stats, d.bb, d.am, d.ncols, d.RH, d.RW, d.fd_row,
                                      d.fd3, d.fd_ncols, d.fd_hw, d.fd_wo, b % d.gx, b / d.gx);
    }
}

// an input gradient of the thin MFMA form planned inside a dgrad + wgrad call (v2::g_pair_collect), waiting for its weight gradient
struct ThinPending {
    bool active = false;
    int k = 3;
    ThinInArgs d;
};
static thread_local ThinPending g_thin_pend;

// ---- host side ---------------------------------------------------------------------------------------
inline bool thin_in_ok(const Geom& g) { return g.Cr <= 4 && g.Nn >= 8 && g.KH * g.KW * g.Cr * 64 * 4 <= 48 * 1024; }

// MOVAE_THIN_PAIR=1: the last conv's input gradient and weight gradient in one launch (thin_pair_k).  OFF by default: measured level
// at C1-C4 (C2 0.766 vs 0.762 ms, C1 0.576 vs 0.579) and 1.3 % slower at C5 (5.49 vs 5.42 ms: the pair takes the larger of the two
// LDS needs for every block, and these two HBM-bound kernels do not overlap the way two latency-bound ones do).
inline bool thin_pair_on() {
    static const bool v = getenv("MOVAE_THIN_PAIR") && atoi(getenv("MOVAE_THIN_PAIR")) != 0;
    return v;
}

inline int thin_flush(hipStream_t st) {  // the weight gradient took another kernel: the planned input gradient goes alone
    ThinPending& p = g_thin_pend;
    if (!p.active) return MOVAE_OK;
    p.active = false;
    const ThinInArgs& d = p.d;
    const dim3 grid(d.gx, d.gy);
    if (p.k == 3)
        hipLaunchKernelGGL((thin_in_mfma_k<3, 3, true>), grid, dim3(256), d.lds_bytes, st, d.X, d.W, d.bias, d.Y, d.g, d.M, d.act, d.slope,
                           d.stats, d.bb, d.am, d.ncols, d.RH, d.RW, d.fd_row, d.fd3, d.fd_ncols, d.fd_hw, d.fd_wo);
    else
        hipLaunchKernelGGL((thin_in_mfma_k<4, 4, true>), grid, dim3(256), d.lds_bytes, st, d.X, d.W, d.bias, d.Y, d.g, d.M, d.act, d.slope,
                           d.stats, d.bb, d.am, d.ncols, d.RH, d.RW, d.fd_row, d.fd3, d.fd_ncols, d.fd_hw, d.fd_wo);
    MOVAE_CHECK_LAUNCH("thin_in_mfma (unpaired input gradient)");
    return MOVAE_OK;
}

template <bool BWD>
int launch_thin_in(const float* X, const float* W, float* Y, const Geom& g, const Epilogue& ep, hipStream_t st) {
    const long Ml = (long)g.Nimg * g.Ho * g.Wo;
    const int M = (int)Ml, K = g.KH * g.KW * g.Cr;
    const bool n32 = g.Nn == 32;
    dim3 grid(ceil_div(M, 256), 1);
    float* stats = nullptr;  // the 32-output kernels hold their tile in LDS: statistics for a following BatchNorm come for free
    if (!BWD && n32 && ep.act == MOVAE_ACT_NONE) stats = fuse_stats_claim((long)grid.x * 4, 32);
    BnBwd bb{};  // input gradient of the last conv = output gradient of the fused BatchNorm in front of it
    if (BWD && n32 && g_fuse.bn_y && ep.act == MOVAE_ACT_NONE && !ep.bias && M % g_fuse.bn_groups == 0 && (M / g_fuse.bn_groups) % 256 == 0) {
        const long rpg = M / g_fuse.bn_groups;
        if (float* part = fuse_bn_claim(rpg / 256 * 4, 32))
            bb = BnBwd{g_fuse.bn_y, g_fuse.bn_scale, g_fuse.bn_shift, g_fuse.bn_slope, part, (int)rpg, (int)(rpg / 256 * 4)};
    }
    // MFMA form: 3 reduction channels, a multiple of 32 outputs (32 per block: grid.y column tiles), 3x3 / 4x4 taps, blocks of whole
    // rows or of a row segment of one image
    static const bool valu_only = getenv("MOVAE_THIN_IN_VALU") != nullptr;  // (A/B knob)
    const int hw = g.Ho * g.Wo;
    const bool k33 = g.KH == 3 && g.KW == 3, k44 = g.KH == 4 && g.KW == 4;
    bool mfma = !valu_only && g.Nn % 32 == 0 && g.Cr == 3 && (k33 || k44) && (!BWD || g.stride == 1) && hw % 256 == 0 &&
                (256 % g.Wo == 0 || g.Wo % 256 == 0) && g.wrow == 0 && (reinterpret_cast<uintptr_t>(Y) & 15) == 0;
    const int ncols = g.Wo < 256 ? g.Wo : 256, nrows = 256 / ncols;
    const int RH = BWD ? nrows + g.KH - 1 : (nrows - 1) * g.stride + g.KH;
    const int RW = BWD ? ncols + g.KW - 1 : (ncols - 1) * g.stride + g.KW;
    size_t lf = (size_t)RH * RW * 3 + (BWD ? 0 : 32 * (g.KH * g.KW * 3 | 1));
    if (lf < 256 * 33) lf = 256 * 33;
    mfma = mfma && lf * sizeof(float) <= 60 * 1024;
    // the previous layer's activation derivative / a residual block's identity cotangent on the result (kernels with the LDS output tile)
    ActMul am{nullptr, 0, 0.f, 0, 0, nullptr};
    if ((g_fuse.am.y || g_fuse.am.res) && (n32 || mfma) && ep.act == MOVAE_ACT_NONE && (!ep.bias || !g_fuse.am.y) && !bb.y && !stats &&
        M % g_fuse.am_groups == 0 && (reinterpret_cast<uintptr_t>(Y) & 15) == 0) {
        am = g_fuse.am;
        am.per_group = (long)(M / g_fuse.am_groups) * g.Nn;
        g_fuse.am_done = true;
    }
    if (mfma) {
        grid.y = g.Nn / 32;
        if (BWD && v2::g_pair_collect && thin_pair_on()) {  // inside a dgrad + wgrad call: planned, launched with the weight gradient (thin_pair_k)
            ThinPending& p = g_thin_pend;
            p.active = true, p.k = k33 ? 3 : 4;
            p.d = ThinInArgs{X, W, ep.bias, Y, g, M, ep.act, ep.slope, stats, bb, am, ncols, RH, RW, fastdiv_make(RW * 3), fastdiv_make(3),
                             fastdiv_make(ncols), fastdiv_make(hw), fastdiv_make(g.Wo), (int)grid.x, (int)grid.y,
                             (unsigned)(lf * sizeof(float))};
            return MOVAE_OK;
        }
#define MOVAE_TI(K)                                                                                                                \
    hipLaunchKernelGGL((thin_in_mfma_k<K, K, BWD>), grid, dim3(256), lf * sizeof(float), st, X, W, ep.bias, Y, g, M, ep.act, ep.slope,  \
                       stats, bb, am, ncols, RH, RW, fastdiv_make(RW * 3), fastdiv_make(3), fastdiv_make(ncols), fastdiv_make(hw),  \
                       fastdiv_make(g.Wo))
        if (k33) MOVAE_TI(3); else MOVAE_TI(4);
#undef MOVAE_TI
        MOVAE_CHECK_LAUNCH("thin_in_mfma");
        return MOVAE_OK;
    }
    if (g.Nn > 32) {
        grid.y = ceil_div(g.Nn, 64);
        hipLaunchKernelGGL((thin_in_k<64, BWD>), grid, dim3(256), (size_t)K * 64 * sizeof(float), st, X, W, ep.bias, Y, g, M,
                           ep.act, ep.slope, (float*)nullptr, BnBwd{}, ActMul{nullptr, 0, 0.f, 0, 0, nullptr});
    } else {
        size_t lds_floats = (size_t)K * 32;
        if (n32 && lds_floats < 256 * 33) lds_floats = 256 * 33;  // room for the coalescing transpose of the outputs
        hipLaunchKernelGGL((thin_in_k<32, BWD>), grid, dim3(256), lds_floats * sizeof(float), st, X, W, ep.bias, Y, g, M,
                           ep.act, ep.slope, stats, bb, am);
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

// tile geometry of thin_out_tile_k; returns false when the layer does not fit the kernel
template <bool BWD>
inline bool thin_out_tile_plan(const Geom& g, const float* X, int& TH, int& TW, int& IH, int& IW, int& CC, int& PPT) {
    if (g.Nn > 4 || g.Cr % 4 != 0 || (reinterpret_cast<uintptr_t>(X) & 15) != 0) return false;
    if (BWD && g.stride != 2) return false;  // stride-1 transposed convs take the generic path
    if (g.KH * g.KW > 16) return false;
    TW = g.Wo < 32 ? (BWD ? (g.Wo + 1) / 2 * 2 : g.Wo) : 32;
    if (TW < 1) return false;
    // two pixels per thread halve the weight reads per FMA but also the block count (measured: better from ~200k pixels)
    const long px_total = (long)g.Nimg * g.Ho * g.Wo;
    static const int force_ppt = getenv("MOVAE_THIN_PPT") ? atoi(getenv("MOVAE_THIN_PPT")) : 0;  // tuning knob
    for (int ppt = force_ppt ? force_ppt : (px_total >= 200000L ? 2 : 1); ppt >= 1; --ppt) {
        int th = 256 * ppt / TW;
        if (BWD) th = th / 2 * 2;
        if (th < (BWD ? 2 : 1)) continue;
        if (BWD && (256 * ppt) % (TW * 2) != 0) continue;          // whole class rows per wave pass
        if (!BWD && (256 * ppt) % TW != 0) continue;
        for (int cc = 32; cc >= 16; cc -= 16) {
            const int ih = BWD ? (th + g.KH - 2) / 2 + 2 : (th - 1) * g.stride + g.KH;
            const int iw = BWD ? (TW + g.KW - 2) / 2 + 2 : (TW - 1) * g.stride + g.KW;
            const long bytes = ((long)ih * iw * (cc + 4) + (long)g.KH * g.KW * 4 * cc) * 4;
            if (bytes <= 62 * 1024) {
                TH = th; IH = ih; IW = iw; CC = cc; PPT = ppt;
                return true;
            }
        }
    }
    return false;
}

template <bool BWD>
int launch_thin_out_tile(const float* X, const float* W, float* Y, const Geom& g, const Epilogue& ep, hipStream_t st, bool* handled) {
    static const bool valu_only = getenv("MOVAE_THIN_OUT_VALU") != nullptr;  // (A/B knob)
    if (!BWD && !valu_only && g.Nn == 3 && g.Cr % 32 == 0 && g.stride == 1 && g.KH == 3 && g.KW == 3 && g.wrow == 0 &&
        g.Ho == g.Hi + 2 * g.pad - 2 && g.Wo == g.Wi + 2 * g.pad - 2 &&
        ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(g_fuse.nrm.scale) |
          reinterpret_cast<uintptr_t>(g_fuse.nrm.shift)) & 15) == 0) {
        const int tiles_h = ceil_div(g.Ho, 8), tiles_w = ceil_div(g.Wo, 32);
        const long nblk = (long)g.Nimg * tiles_h * tiles_w;
        if (nblk <= 0x7fffffffL) {
            constexpr int NG = (10 * 34 + 31) / 32;
            if (g.Cr == 32)
                hipLaunchKernelGGL((thin_out_mfma_k<3, 3, true>), dim3((unsigned)nblk), dim3(256), (size_t)NG * 32 * 27 * sizeof(float), st,
                                   X, W, ep.bias, Y, g, tiles_h, tiles_w, ep.act, ep.slope, g_fuse.nrm);
            else
                hipLaunchKernelGGL((thin_out_mfma_k<3, 3, false>), dim3((unsigned)nblk), dim3(256), (size_t)NG * 32 * 27 * sizeof(float), st,
                                   X, W, ep.bias, Y, g, tiles_h, tiles_w, ep.act, ep.slope, g_fuse.nrm);
            MOVAE_CHECK_LAUNCH("thin_out_mfma");
            *handled = true;
            return MOVAE_OK;
        }
    }
    if (BWD && !valu_only && g.Nn == 3 && g.Cr % 32 == 0 && g.stride == 2 && g.KH == 4 && g.KW == 4 && g.pad == 1 && g.Ho == 2 * g.Hi &&
        g.Wo == 2 * g.Wi && ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(g_fuse.nrm.scale) |
                              reinterpret_cast<uintptr_t>(g_fuse.nrm.shift)) & 15) == 0) {
        const int tiles_h = ceil_div(g.Hi, 8), tiles_w = ceil_div(g.Wi, 32);
        const long nblk = (long)g.Nimg * tiles_h * tiles_w;
        const size_t wt = (size_t)48 * (g.Cr + 4), pt = (size_t)11 * 32 * 49;
        const size_t shb = (wt > pt ? wt : pt) * sizeof(float);
        if (nblk <= 0x7fffffffL && shb <= 80 * 1024) {
            hipLaunchKernelGGL(thin_outT_mfma_k, dim3((unsigned)nblk), dim3(256), shb, st, X, W, ep.bias, Y, g, tiles_h, tiles_w, ep.act,
                               ep.slope, g_fuse.nrm);
            MOVAE_CHECK_LAUNCH("thin_outT_mfma");
            *handled = true;
            return MOVAE_OK;
        }
    }
    int TH, TW, IH, IW, CC, PPT;
    *handled = thin_out_tile_plan<BWD>(g, X, TH, TW, IH, IW, CC, PPT);
    if (!*handled) return MOVAE_OK;
    const int tiles_h = ceil_div(g.Ho, TH), tiles_w = ceil_div(g.Wo, TW);
    const long nblk = (long)g.Nimg * tiles_h * tiles_w;
    if (nblk > 0x7fffffffL) {
        *handled = false;
        return MOVAE_OK;
    }
    const size_t shb = ((size_t)IH * IW * (CC + 4) + (size_t)g.KH * g.KW * 4 * CC) * sizeof(float);
#define MOVAE_TO(CCV, PPTV)                                                                                                      \
    hipLaunchKernelGGL((thin_out_tile_k<BWD, CCV, PPTV>), dim3((unsigned)nblk), dim3(256), shb, st, X, W, ep.bias, Y, g, TH, TW, tiles_h, \
                       tiles_w, IH, IW, ep.act, ep.slope, g_fuse.nrm)
    if (CC == 32) {
        if (PPT == 2) MOVAE_TO(32, 2); else MOVAE_TO(32, 1);
    } else {
        if (PPT == 2) MOVAE_TO(16, 2); else MOVAE_TO(16, 1);
    }
#undef MOVAE_TO
    MOVAE_CHECK_LAUNCH("thin_out_tile");
    return MOVAE_OK;
}

inline bool thin_wgrad_ok(const WGeom& g) { return g.Cs <= 4 || g.Cb <= 4; }

int launch_thin_wgrad(const float* S, const float* Bg, float* dW, const WGeom& g, int K, int accumulate, void* ws,
                      size_t ws_bytes, hipStream_t st);

// G cotangent groups: one sweep launch (blockIdx.z = group) when the sweep kernel applies, else group by group
int launch_thin_wgrad_grouped(const float* S, const float* Bg, float* const* dW, int G, long s_gs, long b_gs, const WGeom& g, int K,
                              int accumulate, void* ws, size_t ws_bytes, hipStream_t st, float* const* colsum = nullptr,
                              bool* colsum_done = nullptr);

int launch_thin_wgrad_impl(const float* S, const float* Bg, float* const* dW, int G, long s_gs, long b_gs, const WGeom& g, int K,
                           int accumulate, void* ws, size_t ws_bytes, hipStream_t st, bool* handled, float* const* colsum = nullptr,
                           bool* colsum_done = nullptr) {
    const int M = g.Cs, N = g.KH * g.KW * g.Cb;
    *handled = true;
    const bool thin_small = g.Cs <= 4;
    const int wide = thin_small ? g.Cb : g.Cs;
    {   // preferred: sweep kernel (3x3 / 4x4 taps, wide side a multiple of 32 channels; thin = S needs stride 1)
        const int tc = thin_small ? g.Cs : g.Cb;
        const bool k33 = g.KH == 3 && g.KW == 3, k44 = g.KH == 4 && g.KW == 4;
        const bool aligned = (reinterpret_cast<uintptr_t>(thin_small ? Bg : S) & 15) == 0;
        if ((k33 || k44) && wide % 32 == 0 && tc == 3 && aligned && (!thin_small || g.stride == 1) && ws) {
            // wide side: REV (thin = S) sweeps the big-side image, else the small-side image
            const int Hw = thin_small ? g.Hb : g.Hs, Ww = thin_small ? g.Wb : g.Ws;
            const int Ht = thin_small ? g.Hs : g.Hb, Wt = thin_small ? g.Ws : g.Wb;
            const int TW = Ww < 32 ? Ww : 32;
            static const int tile_px = getenv("MOVAE_THIN_TILE_PX") ? atoi(getenv("MOVAE_THIN_TILE_PX")) : 256;
            int TH = tile_px / TW;
            if (TH < 1) TH = 1;
            if (TH > Hw) TH = Hw;
            const int TH0 = TH;
            const int taps = g.KH * g.KW, na = taps * 3;
            static const bool mfma_env = !getenv("MOVAE_THIN_WGRAD_SWEEP");  // (A/B knob: the VALU sweep kernel)
            bool use_mfma = mfma_env;
            auto lds_bytes = [&](int th, int tw) {
                const int bh = thin_small ? th + g.KH - 1 : (th - 1) * g.stride + g.KH;
                const int bw = thin_small ? tw + g.KW - 1 : (tw - 1) * g.stride + g.KW;
                long tile_f = (long)th * tw * 32, fold_f = 8L * 32 * ((na + 1) / 2);
                if (use_mfma) tile_f += 32, fold_f = 4L * ((na + 31) / 32) * 1024;
                const long wide_f = tile_f > fold_f ? tile_f : fold_f;
                return (wide_f + (long)bh * bw * 4) * 4;
            };
            // the MFMA kernel prefetches a tile into registers: <= 256 wide pixels (8 pieces per thread), <= 1280 thin pixels (5 per thread)
            auto thin_px = [&](int th, int tw) {
                return (thin_small ? th + g.KH - 1 : (th - 1) * g.stride + g.KH) * (thin_small ? tw + g.KW - 1 : (tw - 1) * g.stride + g.KW);
            };
            while ((lds_bytes(TH, TW) > 60 * 1024 || (use_mfma && (TH * TW > 256 || thin_px(TH, TW) > 1280))) && TH > 1) TH = (TH + 1) / 2;
            // ... and steps through it with shifts: tile width a power of two, tile a multiple of 32 pixels; the VALU sweep kernel otherwise
            if (use_mfma && !(TH * TW <= 256 && thin_px(TH, TW) <= 1280 && (TW & (TW - 1)) == 0 && (TH * TW) % 32 == 0)) {
                use_mfma = false;
                TH = TH0;
                while (lds_bytes(TH, TW) > 60 * 1024 && TH > 1) TH = (TH + 1) / 2;
            }
            const int tiles_h = ceil_div(Hw, TH), tiles_w = ceil_div(Ww, TW);
            const long ntiles = (long)g.Nimg * tiles_h * tiles_w;
            // MFMA kernel: persistent blocks (one slab row each), about four per CU and (wide slice, group)
            // (three blocks of the MFMA kernel fit a CU -- 136 VGPRs + 16 accumulators per lane: 768 persistent blocks fill the chip once)
            static const int persist = getenv("MOVAE_THIN_PERSIST") ? atoi(getenv("MOVAE_THIN_PERSIST")) : 768;
            long nblk = ntiles;
            if (use_mfma) {
                const long cap = persist / ((long)(wide / 32) * G) > 64 ? persist / ((long)(wide / 32) * G) : 64;
                if (nblk > cap) nblk = cap;
            }
            // column sums of S (the bias gradient when S is dy) from the MFMA kernel: S is its thin operand, the images equally large
            // (from 256 k pixels: the three extra floats per slab row take the reduce off its 16-byte path, +5 us at C2's last conv,
            // but the stand-alone column sum it saves is two launches there -- C2 0.822 -> 0.801 ms; C5's last conv: 34 us saved;
            // at C1's 128 k pixels the two are level)
            static const long cs_min = getenv("MOVAE_THIN_CS_MIN") ? atol(getenv("MOVAE_THIN_CS_MIN")) : (1L << 18);
            bool cs_on = use_mfma && thin_small && colsum && g.Hs == g.Hb && g.Ws == g.Wb && (long)K >= cs_min;
            for (int i = 0; cs_on && i < G; ++i) cs_on = colsum[i] != nullptr;
            const long row = (long)M * N + (cs_on ? M : 0);  // floats per slab row
            const size_t per1 = (size_t)row * sizeof(float);
            if (lds_bytes(TH, TW) <= 60 * 1024 && ntiles <= 0x7fffffffL && per1 * (size_t)nblk * G <= ws_bytes) {
                float* slab = static_cast<float*>(ws);
                const dim3 grid((unsigned)nblk, wide / 32, G);
                const long slab_gs = (long)nblk * row;
                const long wide_gs = thin_small ? b_gs : s_gs, thin_gs = thin_small ? s_gs : b_gs;
                const size_t shb = (size_t)lds_bytes(TH, TW);
                const float* Wd = thin_small ? Bg : S;
                const float* Tn = thin_small ? S : Bg;
                // fused input transform: only when the activation is the WIDE operand (the thin one would be the image side)
                Norm wide_nrm{nullptr, nullptr, 1.f};
                if (fuse_norm()) {
                    if (g_fuse.nrm_side != (thin_small ? 2 : 1)) {
                        movae_set_error("thin wgrad: the fused input transform applies to the wide operand only");
                        return MOVAE_EUNSUPPORTED;
                    }
                    wide_nrm = g_fuse.nrm;
                }
#define MOVAE_SW(K, REVV)                                                                                                      \
    hipLaunchKernelGGL((thin_wgrad_sweep_k<3, K, K, REVV>), grid, dim3(256), shb, st, Wd, Tn, slab, Hw, Ww, wide, Ht, Wt, g.stride, \
                       g.pad, TH, TW, tiles_h, tiles_w, g.Cs, g.Cb, wide_gs, thin_gs, slab_gs, wide_nrm)
#define MOVAE_MF(K, REVV)                                                                                                      \
    hipLaunchKernelGGL((thin_wgrad_mfma_k<3, K, K, REVV>), grid, dim3(256), shb, st, Wd, Tn, slab, Hw, Ww, wide, Ht, Wt, g.stride, \
                       g.pad, TH, TW, tiles_h, tiles_w, g.Cs, g.Cb, wide_gs, thin_gs, slab_gs, wide_nrm, (int)ntiles, fastdiv_make(TW),    \
                       fastdiv_make(thin_small ? TW + g.KW - 1 : (TW - 1) * g.stride + g.KW), cs_on ? 1 : 0, row)
                if (use_mfma && thin_small && g_thin_pend.active && g_thin_pend.k == (k33 ? 3 : 4)) {
                    // the layer's input gradient waits: one launch for both (thin_pair_k)
                    g_thin_pend.active = false;
                    const ThinInArgs& d = g_thin_pend.d;
                    const ThinWgArgs w{Wd, Tn, slab, Hw, Ww, wide, Ht, Wt, g.stride, g.pad, TH, TW, tiles_h, tiles_w, g.Cs, g.Cb, wide_gs, thin_gs,
                                       slab_gs, wide_nrm, (int)ntiles, fastdiv_make(TW), fastdiv_make(TW + g.KW - 1), cs_on ? 1 : 0, row,
                                       (int)grid.x, (int)grid.y, (int)grid.z, (unsigned)shb};
                    const int nw = (int)(grid.x * grid.y * grid.z), nd = d.gx * d.gy;
                    const unsigned lds = d.lds_bytes > (unsigned)shb ? d.lds_bytes : (unsigned)shb;
                    if (k33) hipLaunchKernelGGL((thin_pair_k<3, true>), dim3(nw + nd), dim3(256), lds, st, d, w, nw);
                    else hipLaunchKernelGGL((thin_pair_k<4, true>), dim3(nw + nd), dim3(256), lds, st, d, w, nw);
                    g_last_kernel = k33 ? "thin_pair_k<3,true>" : "thin_pair_k<4,true>";
                } else if (use_mfma) {
                    if (thin_small) {
                        if (k33) MOVAE_MF(3, true); else MOVAE_MF(4, true);
                    } else {
                        if (k33) MOVAE_MF(3, false); else MOVAE_MF(4, false);
                    }
                } else if (thin_small) {
                    if (k33) MOVAE_SW(3, true); else MOVAE_SW(4, true);
                } else {
                    if (k33) MOVAE_SW(3, false); else MOVAE_SW(4, false);
                }
#undef MOVAE_SW
#undef MOVAE_MF
                MOVAE_CHECK_LAUNCH("thin_wgrad_sweep");
                {  // ONE reduce launch for all groups (blockIdx.y = group); it may wait for a later launch to carry it (RSide)
                    RGroups rg{};
                    for (int i = 0; i < G; ++i) rg.out[i] = dW[i], rg.out2[i] = cs_on ? colsum[i] : nullptr;
                    rg.slab_gs = slab_gs;
                    if (int rc = launch_reduce_groups(slab, rg, G, (long)M * N, cs_on ? M : 0, (int)nblk, N, nullptr, 0, 0.f, accumulate, st,
                                                      ActMul{nullptr, 0, 0.f, 0, 0, nullptr}, true))
                        return rc;
                }
                if (colsum_done) *colsum_done = cs_on;
                return MOVAE_OK;
            }
        }
    }
    *handled = false;
    MOVAE_NO_NORM("thin wgrad (non-sweep kernels)");
    return MOVAE_OK;
}

int launch_thin_wgrad_grouped(const float* S, const float* Bg, float* const* dW, int G, long s_gs, long b_gs, const WGeom& g, int K,
                              int accumulate, void* ws, size_t ws_bytes, hipStream_t st, float* const* colsum, bool* colsum_done) {
    bool handled = false;
    if (int rc = launch_thin_wgrad_impl(S, Bg, dW, G, s_gs, b_gs, g, K, accumulate, ws, ws_bytes, st, &handled, colsum, colsum_done))
        return rc;
    if (handled) return MOVAE_OK;
    for (int i = 0; i < G; ++i)
        if (int rc = launch_thin_wgrad(S + i * s_gs, Bg + i * b_gs, dW[i], g, K, accumulate, ws, ws_bytes, st)) return rc;
    return MOVAE_OK;
}

int launch_thin_wgrad(const float* S, const float* Bg, float* dW, const WGeom& g, int K, int accumulate, void* ws,
                      size_t ws_bytes, hipStream_t st) {
    const int M = g.Cs, N = g.KH * g.KW * g.Cb;
    const bool thin_small = g.Cs <= 4;
    const int wide = thin_small ? g.Cb : g.Cs;
    {
        float* one[1] = {dW};
        bool handled = false;
        if (int rc = launch_thin_wgrad_impl(S, Bg, one, 1, 0, 0, g, K, accumulate, ws, ws_bytes, st, &handled)) return rc;
        if (handled) return MOVAE_OK;
    }
    {   // LDS-tiled kernel (tile sized to <= 48 KiB of LDS)
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
