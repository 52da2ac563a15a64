// Implicit-GEMM convolution family for gfx950 (MI355X), fp32 in / fp32 accumulate on
// v_mfma_f32_32x32x2_f32 (exact fp32 fma chains -- the reference path is fp32 end to end).
//
// Three gather forms cover every conv / transposed-conv / linear pass of the hot path:
//   FWD   y[p][n]  = sum_{tap,c} x[p*s - pad + tap][c] * W[n][tap][c]        conv fwd, convT dgrad, linear fwd
//   BWD   y[p][n]  = sum_{tap,c} x[(p + pad - tap)/s][c] * W[c][tap][n]      conv dgrad, convT fwd, linear dgrad
//                    (decomposed into s*s output-parity classes so no zero taps are multiplied)
//   WGRAD dW[a][tap][b] = sum_p S[p][a] * Bg[p*s - pad + tap][b]             conv/convT/linear wgrad (split-K)
// Activations are NHWC so the reduction channel is the contiguous axis of every gather; weights
// are kept in the channels_last image of the reference's parameter shapes so no repacking pass
// exists.  Block = 256 threads (4 waves), tile BM x BN x 16, operands staged k-major in LDS
// (conflict-free ds_read_b32 per MFMA operand), global loads for tile t+1 issued before the
// MFMAs of tile t.
#include <cstring>
#include <cstdio>
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int BK = 16;

struct Geom {
    int Nimg;
    int Hi, Wi, Cr;  // gathered tensor: spatial dims, reduction channels
    int Ho, Wo, Nn;  // output pixel grid, output channels
    int KH, KW, stride, pad;
    // FWD gather only, 0 = dense: the taps form a KH x KW window of a larger stored kernel -- the weight row of one output
    // channel has wrow floats, starts woff floats in, and every wlen floats (one window row) jump to the next stored row
    // wstride floats on (tap_window(): 1x1 outputs whose other taps only ever meet padding)
    int wrow, wlen, wstride, woff;
};

struct Epilogue {
    const float* bias;  // [Nn] or null
    int act;
    float slope;
};

// The result is the gradient w.r.t. the OUTPUT of an activation -- the epilogue activation of the layer before, whose stored output
// is y (same layout as the result; shared by the cotangent groups, per_group floats each): multiply by act'(.) taken from that
// output, so the layer before starts its backward from the pre-activation gradient and runs no activation-backward pass.
struct ActMul {
    const float* y;  // null: no derivative factor
    int act;
    float slope;
    long per_group;  // floats of y
    int gx_per_group;  // BWD-form kernel: row blocks per group and class (0: one group)
    // ... and / or `res` (layout and size of the result, cotangent groups included) is ADDED to it: the identity branch of a
    // residual block, so the block's input gradient leaves this kernel complete (no accumulation launch by the autograd engine)
    const float* res;
};
__device__ __forceinline__ bool actmul_on(const ActMul& am) { return am.y != nullptr || am.res != nullptr; }
__device__ __forceinline__ float actmul_apply(const ActMul& am, float v, long i) {  // i: index into the result
    if (am.y) v *= act_grad_from_out(am.y[i % am.per_group], am.act, am.slope);
    if (am.res) v += am.res[i];
    return v;
}

// Division by a launch-invariant divisor inside the k loops (tap / pixel decoding of the gathers): n / d for 0 <= n < 2^31 as
// one v_mul_hi + one shift instead of the ~35-instruction software division (the 128x128 kernels issued 4-8 VALU instructions
// per MFMA, most of them these divisions; PMC: profiles/r02_pmc_igemm128.json).  m = ceil(2^(31+s) / d), s = ceil(log2 d):
// exact because the rounding error of m, < d <= 2^s, times n < 2^31 stays below 2^(31+s).
// (d = 1 has no such m: mul = 0 and the all-ones `one` mask passes n through -- branch-free in the kernel.)
struct FastDiv {
    unsigned mul, shift, one;
};
inline FastDiv fastdiv_make(int d) {
    FastDiv f{0u, 0u, d <= 1 ? 0xffffffffu : 0u};
    if (d <= 1) return f;
    unsigned s = 0;
    while ((1u << s) < (unsigned)d) ++s;
    const unsigned long long p = 31ull + s;
    f.mul = (unsigned)(((1ull << p) + (unsigned long long)d - 1ull) / (unsigned long long)d);
    f.shift = s - 1u;
    return f;
}
__device__ __forceinline__ int fdiv(int n, const FastDiv& f) {
    return (int)((__umulhi((unsigned)n, f.mul) >> f.shift) | ((unsigned)n & f.one));
}

// Fusion of a training-mode BatchNorm (+ LeakyReLU / ReLU) into its neighbours:
//  * Norm -- the gathered activation operand is VIRTUAL: the kernel reads the producer's raw conv output y and applies
//    x = act(scale[c] * y + shift[c]) (scale = gamma * rstd, shift = beta - mean * scale, act = leaky-ReLU with `slope`;
//    slope 1 = none, 0 = ReLU) between the global load and the LDS store; padding stays exactly zero.  The normalised
//    activation is never written to memory.
//  * stats -- the epilogue also emits, per column (output channel), the partial sums (sum v, sum v^2) of the values it
//    stores, one pair per (row block, wave): the statistics pass of the BatchNorm that follows reads no activation.
struct Norm {
    const float* scale;  // [C] or null: plain operand
    const float* shift;
    float slope;
};

__device__ __forceinline__ f32x4 norm_apply(f32x4 v, const f32x4 sc, const f32x4 sh, float slope) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float z = fmaf(v[e], sc[e], sh[e]);
        v[e] = z > 0.f ? z : z * slope;
    }
    return v;
}
// branch-free form for the staging loops: padding / out-of-range chunks (valid == false) stay exactly zero
__device__ __forceinline__ float norm1(float v, float sc, float sh, float slope, float keep) {
    const float z = fmaf(v, sc, sh);
    return keep * (fmaxf(z, 0.f) + slope * fminf(z, 0.f));
}
__device__ __forceinline__ f32x4 norm_apply_if(bool valid, const f32x4 v, const f32x4 sc, const f32x4 sh, float slope) {
    const float keep = valid ? 1.f : 0.f;
    return f32x4{norm1(v[0], sc[0], sh[0], slope, keep), norm1(v[1], sc[1], sh[1], slope, keep), norm1(v[2], sc[2], sh[2], slope, keep),
                 norm1(v[3], sc[3], sh[3], slope, keep)};
}

// The result of an input-gradient pass is `dout`, the gradient w.r.t. the (never materialised) output of a fused BatchNorm +
// activation over the raw conv output y: the epilogue also emits that BatchNorm's backward sums, per column and cotangent
// group, d = dout * act'(scale * y + shift):   (sum d, sum d * y)   -- the BatchNorm backward then needs no reduction pass.
// y is shared by the cotangent groups (rows_per_group rows each); the partials of group g are rows [g * ppg, (g + 1) * ppg).
struct BnBwd {
    const float* y;      // null: none
    const float* scale;
    const float* shift;
    float slope;
    float* part;         // [groups * ppg][2][N]
    int rows_per_group;
    int ppg;             // partials per group
};

// What the *_f entry points ask of the next conv-family dispatch of the calling thread (cleared by the entry point on return).
// The launchers that can honour a request mark it: an operand transform that the dispatched kernel cannot apply is an
// error raised BEFORE anything is launched (MOVAE_EUNSUPPORTED: the caller materialises the activation and calls the plain
// entry point); statistics that the dispatched kernel cannot emit leave stats_parts at 0 (the caller runs the stand-alone
// statistics pass instead).
// "finish the BatchNorm that follows inside this launch" (movae_fuse_t::fin_*): what the finalizing block needs
struct BnFin {
    const float* gamma = nullptr;  // null: no request
    const float* beta = nullptr;
    float* out = nullptr;           // [4][N]: save_mean, save_rstd, scale, shift
    float* running_mean = nullptr;
    float* running_var = nullptr;
    long long* nbt = nullptr;
    unsigned* counter = nullptr;    // arrival counters, one per column tile (workspace header, zero between launches)
    float eps = 0.f, momentum = 0.f;
    int rows = 0, parts = 0, group = 0;  // rows of the result, partial pairs per column, blocks per column tile
};

struct FuseCtx {
    BnFin fin;                        // in: the request (gamma .. nbt, eps, momentum); fin_done: a kernel honoured it
    bool fin_done = false;
    Norm nrm{nullptr, nullptr, 1.f};  // virtual activation operand (x of fwd / wgrad)
    int nrm_side = 0;                 // wgrad: which operand is the activation -- 1 small side Sm, 2 gathered side Bg
    float* stats = nullptr;           // column partial sums of the forward result
    size_t stats_cap = 0;             // floats available at `stats`
    int stats_parts = 0;              // out: partial pairs written per column (0 = none)
    // backward sums of the fused BatchNorm whose output gradient the dispatched input-gradient pass produces (BnBwd)
    const float* bn_y = nullptr;
    const float* bn_scale = nullptr;
    const float* bn_shift = nullptr;
    float bn_slope = 1.f;
    float* bn_part = nullptr;
    size_t bn_cap = 0;
    int bn_groups = 1;
    int bn_ppg = 0;                   // out: partial pairs per group and column (0 = none)
    // activation derivative of the layer before, applied by the input-gradient pass's epilogue / reduce (ActMul)
    ActMul am{nullptr, 0, 0.f, 0, 0, nullptr};
    int am_groups = 1;
    bool am_done = false;             // out
};
static thread_local FuseCtx g_fuse;

inline bool fuse_norm() { return g_fuse.nrm.scale != nullptr; }
// claims the statistics slot for `parts` partials of `n` columns; false (and no statistics) when they do not fit
inline float* fuse_stats_claim(long parts, int n) {
    if (!g_fuse.stats || parts <= 0 || (size_t)parts * 2 * (size_t)n > g_fuse.stats_cap) return nullptr;
    g_fuse.stats_parts = (int)parts;
    return g_fuse.stats;
}
// claims room for `groups * ppg` backward-sum partials of `n` columns; null when absent / no room
inline float* fuse_bn_claim(long ppg, int n) {
    if (!g_fuse.bn_part || !g_fuse.bn_y || ppg <= 0 || (size_t)ppg * g_fuse.bn_groups * 2 * (size_t)n > g_fuse.bn_cap) return nullptr;
    g_fuse.bn_ppg = (int)ppg;
    return g_fuse.bn_part;
}
#define MOVAE_NO_NORM(what)                                                                                 \
    do {                                                                                                    \
        if (fuse_norm()) {                                                                                  \
            movae_set_error("%s: this shape dispatches to a kernel without the fused input transform", what); \
            return MOVAE_EUNSUPPORTED;                                                                      \
        }                                                                                                   \
    } while (0)

template <int TM>
__device__ __forceinline__ void mma_tile(const float* __restrict__ As, const float* __restrict__ Bs, int lda, int ldb,
                                         int a_off, int b_off, f32x16 (&acc)[TM]) {
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
        const float b = Bs[(2 * kk + half) * ldb + b_off + l31];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const float a = As[(2 * kk + half) * lda + a_off + tm * 32 + l31];
            acc[tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[tm], 0, 0, 0);
        }
    }
}

template <int BM, int BN>
struct Tile {
    static constexpr int WN = BN / 32;        // waves along N
    static constexpr int WM = 4 / WN;         // waves along M
    static constexpr int TM = BM / (WM * 32); // 32-row MFMA tiles per wave
    static constexpr int LDA = BM + 2;        // k-major leading dims (+2 keeps the transposing store conflict-free)
    static constexpr int LDB = BN + 2;
    static_assert(TM >= 1, "tile");
};

// ------------------------------------------------------------------------------------------------
// FWD form
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, bool VEC>
__global__ __launch_bounds__(256) void igemm_fwd(const float* __restrict__ X, const float* __restrict__ W,
                                                 float* __restrict__ Y, Geom g, Epilogue ep, int M, int K,
                                                 int ktiles_per_split, float* __restrict__ slab) {
    using T = Tile<BM, BN>;
    __shared__ float As[BK * T::LDA];
    __shared__ float Bs[BK * T::LDB];
    const int t = threadIdx.x;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int N = g.Nn;
    constexpr int AR = BM / 64;
    constexpr int BR = (BN + 63) / 64;
    const int kq = t & 3, r4 = t >> 2;

    const float* a_base[AR];
    int a_h0[AR], a_w0[AR];
    bool a_ok[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + r4 + i * 64;
        a_ok[i] = m < M;
        const int mm = a_ok[i] ? m : 0;
        const int hw = g.Ho * g.Wo;
        const int img = mm / hw, rem = mm - img * hw;
        const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
        a_h0[i] = ho * g.stride - g.pad;
        a_w0[i] = wo * g.stride - g.pad;
        a_base[i] = X + (long)img * g.Hi * g.Wi * g.Cr;
    }
    const bool b_thread = (BN >= 64) || (r4 < BN);
    bool b_ok[BR];
    const float* b_base[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) {
        const int n = n0 + r4 + i * 64;
        b_ok[i] = b_thread && n < N;
        b_base[i] = W + (long)(b_ok[i] ? n : 0) * K;
    }

    const int nk_total = (K + BK - 1) / BK;
    const int kt_begin = blockIdx.z * ktiles_per_split;
    const int kt_end = min(nk_total, kt_begin + ktiles_per_split);

    f32x4 ra[AR], rb[BR];
    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        if (VEC) {
            const int tap = k0 / g.Cr, c0 = k0 - tap * g.Cr;
            const int kh = tap / g.KW, kw = tap - kh * g.KW;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const int h = a_h0[i] + kh, w = a_w0[i] + kw;
                const bool v = a_ok[i] && h >= 0 && h < g.Hi && w >= 0 && w < g.Wi;
                ra[i] = v ? *reinterpret_cast<const f32x4*>(a_base[i] + ((long)h * g.Wi + w) * g.Cr + c0 + kq * 4)
                          : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int i = 0; i < BR; ++i)
                rb[i] = b_ok[i] ? *reinterpret_cast<const f32x4*>(b_base[i] + k0 + kq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + kq * 4 + j;
                const bool kv = k < K;
                const int kk = kv ? k : 0;
                const int tap = kk / g.Cr, c = kk - tap * g.Cr;
                const int kh = tap / g.KW, kw = tap - kh * g.KW;
#pragma unroll
                for (int i = 0; i < AR; ++i) {
                    const int h = a_h0[i] + kh, w = a_w0[i] + kw;
                    const bool v = kv && a_ok[i] && h >= 0 && h < g.Hi && w >= 0 && w < g.Wi;
                    ra[i][j] = v ? a_base[i][((long)h * g.Wi + w) * g.Cr + c] : 0.f;
                }
#pragma unroll
                for (int i = 0; i < BR; ++i) rb[i][j] = (kv && b_ok[i]) ? b_base[i][kk] : 0.f;
            }
        }
    };

    f32x16 acc[T::TM];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int wave = t >> 6;
    const int wm = wave / T::WN, wn = wave % T::WN;
    if (kt_begin < kt_end) load_tile(kt_begin);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < AR; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) As[(kq * 4 + j) * T::LDA + r4 + i * 64] = ra[i][j];
        if (b_thread) {
#pragma unroll
            for (int i = 0; i < BR; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Bs[(kq * 4 + j) * T::LDB + r4 + i * 64] = rb[i][j];
        }
        __syncthreads();
        if (kt + 1 < kt_end) load_tile(kt + 1);
        mma_tile<T::TM>(As, Bs, T::LDA, T::LDB, wm * T::TM * 32, wn * 32, acc);
    }

    const int lane = t & 63, half = lane >> 5, l31 = lane & 31;
    const int n = n0 + wn * 32 + l31;
    if (n >= N) return;
    const bool to_slab = slab != nullptr;
    const float bv = (!to_slab && ep.bias) ? ep.bias[n] : 0.f;
    float* out = to_slab ? slab + (long)blockIdx.z * M * N : Y;
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * T::TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m < M) {
                float v = acc[tm][r];
                if (!to_slab) v = apply_act(v + bv, ep.act, ep.slope);
                out[(long)m * N + n] = v;
            }
        }
}

// ------------------------------------------------------------------------------------------------
// BWD form (one output-parity class per blockIdx.z)
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, bool VEC>
__global__ __launch_bounds__(256) void igemm_bwd(const float* __restrict__ X, const float* __restrict__ W,
                                                 float* __restrict__ Y, Geom g, Epilogue ep, int S, int ktiles_per_split,
                                                 float* __restrict__ slab, long total) {
    using T = Tile<BM, BN>;
    __shared__ float As[BK * T::LDA];
    __shared__ float Bs[BK * T::LDB];
    const int t = threadIdx.x;
    const int s = g.stride;
    const int cls = blockIdx.z / S, split = blockIdx.z - cls * S;
    const int ph = cls / s, pw = cls % s;
    const int Hoc = (g.Ho - ph + s - 1) / s, Woc = (g.Wo - pw + s - 1) / s;
    const int M = g.Nimg * Hoc * Woc;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    if (m0 >= M) return;
    const int N = g.Nn;
    const int kh0 = (ph + g.pad) % s, kw0 = (pw + g.pad) % s;
    const int nA = kh0 < g.KH ? (g.KH - kh0 + s - 1) / s : 0;
    const int nB = kw0 < g.KW ? (g.KW - kw0 + s - 1) / s : 0;
    const int qh = (ph + g.pad - kh0) / s, qw = (pw + g.pad - kw0) / s;
    const int K = nA * nB * g.Cr;
    const int taps = g.KH * g.KW;

    constexpr int AR = BM / 64;
    const int kq = t & 3, r4 = t >> 2;
    const float* a_base[AR];
    int a_h0[AR], a_w0[AR];
    bool a_ok[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + r4 + i * 64;
        a_ok[i] = m < M;
        const int mm = a_ok[i] ? m : 0;
        const int hw = Hoc * Woc;
        const int img = mm / hw, rem = mm - img * hw;
        const int hc = rem / Woc, wc = rem - hc * Woc;
        a_h0[i] = hc + qh;
        a_w0[i] = wc + qw;
        a_base[i] = X + (long)img * g.Hi * g.Wi * g.Cr;
    }
    // B tile: BK rows x BN cols, n contiguous
    constexpr int BQ = BN / 4;  // float4 per k-row
    const int bk = t / BQ, bnq = t % BQ;
    const bool b_thread = bk < BK;

    f32x4 ra[AR], rb;
    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        if (VEC) {
            const int tt = k0 / g.Cr, c0 = k0 - tt * g.Cr;
            const int a = tt / nB, b = tt - a * nB;
            const int kh = kh0 + s * a, kw = kw0 + s * b;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const int h = a_h0[i] - a, w = a_w0[i] - b;
                const bool v = a_ok[i] && h >= 0 && h < g.Hi && w >= 0 && w < g.Wi;
                ra[i] = v ? *reinterpret_cast<const f32x4*>(a_base[i] + ((long)h * g.Wi + w) * g.Cr + c0 + kq * 4)
                          : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (b_thread) {
                const int c = c0 + bk;
                const int n = n0 + bnq * 4;
                const float* src = W + ((long)c * taps + kh * g.KW + kw) * N + n;
                if (n + 3 < N)
                    rb = *reinterpret_cast<const f32x4*>(src);
                else
                    for (int j = 0; j < 4; ++j) rb[j] = (n + j < N) ? src[j] : 0.f;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + kq * 4 + j;
                const bool kv = k < K;
                const int kk = kv ? k : 0;
                const int tt = kk / g.Cr, c = kk - tt * g.Cr;
                const int a = nB > 0 ? tt / nB : 0, b = tt - a * nB;
#pragma unroll
                for (int i = 0; i < AR; ++i) {
                    const int h = a_h0[i] - a, w = a_w0[i] - b;
                    const bool v = kv && a_ok[i] && h >= 0 && h < g.Hi && w >= 0 && w < g.Wi;
                    ra[i][j] = v ? a_base[i][((long)h * g.Wi + w) * g.Cr + c] : 0.f;
                }
            }
            if (b_thread) {
                const int k = k0 + bk;
                const bool kv = k < K;
                const int kk = kv ? k : 0;
                const int tt = kk / g.Cr, c = kk - tt * g.Cr;
                const int a = nB > 0 ? tt / nB : 0, b = tt - a * nB;
                const int kh = kh0 + s * a, kw = kw0 + s * b;
                const float* src = W + ((long)c * taps + kh * g.KW + kw) * N;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + bnq * 4 + j;
                    rb[j] = (kv && n < N) ? src[n] : 0.f;
                }
            }
        }
    };

    f32x16 acc[T::TM];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int wave = t >> 6;
    const int wm = wave / T::WN, wn = wave % T::WN;
    const int nk_total = (K + BK - 1) / BK;
    const int kt_begin = split * ktiles_per_split;
    const int nk = min(nk_total, kt_begin + ktiles_per_split);
    if (kt_begin < nk) load_tile(kt_begin);
    for (int kt = kt_begin; kt < nk; ++kt) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < AR; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) As[(kq * 4 + j) * T::LDA + r4 + i * 64] = ra[i][j];
        if (b_thread) {
#pragma unroll
            for (int j = 0; j < 4; ++j) Bs[bk * T::LDB + bnq * 4 + j] = rb[j];
        }
        __syncthreads();
        if (kt + 1 < nk) load_tile(kt + 1);
        mma_tile<T::TM>(As, Bs, T::LDA, T::LDB, wm * T::TM * 32, wn * 32, acc);
    }

    const int lane = t & 63, half = lane >> 5, l31 = lane & 31;
    const int n = n0 + wn * 32 + l31;
    if (n >= N) return;
    const bool to_slab = slab != nullptr;
    const float bv = (!to_slab && ep.bias) ? ep.bias[n] : 0.f;
    float* out = to_slab ? slab + (long)split * total : Y;
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * T::TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m < M) {
                const int hw = Hoc * Woc;
                const int img = m / hw, rem = m - img * hw;
                const int hc = rem / Woc, wc = rem - hc * Woc;
                const long p = ((long)img * g.Ho + (hc * s + ph)) * g.Wo + (wc * s + pw);
                out[p * N + n] = to_slab ? acc[tm][r] : apply_act(acc[tm][r] + bv, ep.act, ep.slope);
            }
        }
}

// ------------------------------------------------------------------------------------------------
// WGRAD form: dW[a][tap][b] = sum_p S[p][a] * Bg[p*s - pad + tap][b]   (M = Cs, N = taps*Cb, K = pixels)
// ------------------------------------------------------------------------------------------------
struct WGeom {
    int Nimg;
    int Hs, Ws, Cs;  // small-side tensor (conv: dy; convT: x)
    int Hb, Wb, Cb;  // big-side tensor   (conv: x;  convT: dy)
    int KH, KW, stride, pad;
};

template <int BM, int BN, bool VECA, bool VECB>
__global__ __launch_bounds__(256) void igemm_wgrad(const float* __restrict__ S, const float* __restrict__ Bg,
                                                   float* __restrict__ out, WGeom g, int K, int kchunk, int to_slab) {
    using T = Tile<BM, BN>;
    __shared__ float As[BK * T::LDA];
    __shared__ float Bs[BK * T::LDB];
    const int t = threadIdx.x;
    const int M = g.Cs, N = g.KH * g.KW * g.Cb;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int k_begin = blockIdx.z * kchunk, k_end = min(K, k_begin + kchunk);

    constexpr int AQ = BM / 4, BQ = BN / 4;  // float4 per k-row
    constexpr int APASS = (BK * AQ + 255) / 256;
    const int hw = g.Hs * g.Ws;

    f32x4 ra[APASS], rb;
    const int bk = t / BQ, bnq = t % BQ;
    const bool b_thread = bk < BK;
    // column (n) decode is loop invariant
    int b_tap[4], b_c[4];
    bool b_nv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + bnq * 4 + j;
        b_nv[j] = n < N;
        const int nn = b_nv[j] ? n : 0;
        b_tap[j] = nn / g.Cb;
        b_c[j] = nn - b_tap[j] * g.Cb;
    }
    const int tap0_kh = b_tap[0] / g.KW, tap0_kw = b_tap[0] - tap0_kh * g.KW;

    auto load_tile = [&](int k0) {
#pragma unroll
        for (int pss = 0; pss < APASS; ++pss) {
            const int id = t + pss * 256;
            const int ak = id / AQ, amq = id % AQ;
            const int k = k0 + ak, m = m0 + amq * 4;
            if (ak < BK) {
                if (VECA) {
                    ra[pss] = (k < k_end && m < M) ? *reinterpret_cast<const f32x4*>(S + (long)k * M + m)
                                                 : f32x4{0.f, 0.f, 0.f, 0.f};
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) ra[pss][j] = (k < k_end && m + j < M) ? S[(long)k * M + m + j] : 0.f;
                }
            }
        }
        if (b_thread) {
            const int k = k0 + bk;
            const bool kv = k < k_end;
            const int kk = kv ? k : 0;
            const int img = kk / hw, rem = kk - img * hw;
            const int hs = rem / g.Ws, ws = rem - hs * g.Ws;
            const float* base = Bg + (long)img * g.Hb * g.Wb * g.Cb;
            if (VECB) {
                const int h = hs * g.stride - g.pad + tap0_kh, w = ws * g.stride - g.pad + tap0_kw;
                const bool v = kv && b_nv[0] && h >= 0 && h < g.Hb && w >= 0 && w < g.Wb;
                rb = v ? *reinterpret_cast<const f32x4*>(base + ((long)h * g.Wb + w) * g.Cb + b_c[0])
                       : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kh = b_tap[j] / g.KW, kw = b_tap[j] - kh * g.KW;
                    const int h = hs * g.stride - g.pad + kh, w = ws * g.stride - g.pad + kw;
                    const bool v = kv && b_nv[j] && h >= 0 && h < g.Hb && w >= 0 && w < g.Wb;
                    rb[j] = v ? base[((long)h * g.Wb + w) * g.Cb + b_c[j]] : 0.f;
                }
            }
        }
    };

    f32x16 acc[T::TM];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int wave = t >> 6;
    const int wm = wave / T::WN, wn = wave % T::WN;
    if (k_begin < k_end) load_tile(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        __syncthreads();
#pragma unroll
        for (int pss = 0; pss < APASS; ++pss) {
            const int id = t + pss * 256;
            const int ak = id / AQ, amq = id % AQ;
            if (ak < BK) {
#pragma unroll
                for (int j = 0; j < 4; ++j) As[ak * T::LDA + amq * 4 + j] = ra[pss][j];
            }
        }
        if (b_thread) {
#pragma unroll
            for (int j = 0; j < 4; ++j) Bs[bk * T::LDB + bnq * 4 + j] = rb[j];
        }
        __syncthreads();
        if (k0 + BK < k_end) load_tile(k0 + BK);
        mma_tile<T::TM>(As, Bs, T::LDA, T::LDB, wm * T::TM * 32, wn * 32, acc);
    }

    const int lane = t & 63, half = lane >> 5, l31 = lane & 31;
    const int n = n0 + wn * 32 + l31;
    if (n >= N) return;
    float* dst = to_slab ? out + (long)blockIdx.z * M * N : out;
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * T::TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m < M) dst[(long)m * N + n] = acc[tm][r];
        }
}

// out[i] = (accumulate ? out[i] : 0) + sum_z slab[z][i], then optional bias (per column n = i % N) + activation
// Block = 16 split-lanes x 16 consecutive outputs: lane (sl, il) sums slabs sl, sl+16, ... of output i0+il
// (64-byte coalesced rows), then the 16 split-lanes are folded with 4 shuffles.  Deterministic.
// (all three plain reduces: outputs [0, n1) go to `out`, the tail [n1, total) to `out2` -- the bias-gradient partials a weight-
// gradient slab carries behind its M * N floats; n1 == total, out2 == null otherwise)
// (all four plain reduces: blockIdx.y = cotangent group -- the group's slabs start rg.slab_gs floats further, its destinations are
// rg.out / rg.out2[group]; one launch for the G weight gradients of a grouped call instead of G)
struct RGroups {
    float* out[8];
    float* out2[8];
    long slab_gs;
};
__device__ __forceinline__ void splitk_reduce_body(const float* __restrict__ slab, const RGroups& rg, long total, int S, int N,
                                                   const float* __restrict__ bias, int act, float slope, int accumulate, long n1,
                                                   const ActMul& am, int bx, int by, float* lds) {
    slab += by * rg.slab_gs;
    float* __restrict__ out = rg.out[by];
    float* __restrict__ out2 = rg.out2[by];
    const int il = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const long i = (long)bx * 16 + il;
    float v = 0.f;
    if (i < total)
        for (int z = sl; z < S; z += 16) v += slab[(long)z * total + i];
    // lanes of one wave hold sl in {4w..4w+3}; fold those with shuffles, then the 4 waves through LDS
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    float(*sh)[16] = reinterpret_cast<float(*)[16]>(lds);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < 16) sh[w][il] = v;
    __syncthreads();
    if (threadIdx.x < 16 && i < total) {
        v = (sh[0][il] + sh[1][il]) + (sh[2][il] + sh[3][il]);
        if (bias) v += bias[i % N];
        v = actmul_on(am) ? actmul_apply(am, v, i) : apply_act(v, act, slope);
        float* o = i < n1 ? out + i : out2 + (i - n1);
        *o = accumulate ? *o + v : v;
    }
}
__global__ __launch_bounds__(256) void splitk_reduce(const float* __restrict__ slab, RGroups rg, long total,
                                                     int S, int N, const float* __restrict__ bias, int act, float slope,
                                                     int accumulate, long n1, ActMul am) {
    __shared__ __attribute__((aligned(16))) float lds[64];
    splitk_reduce_body(slab, rg, total, S, N, bias, act, slope, accumulate, n1, am, blockIdx.x, blockIdx.y, lds);
}

// many slabs (S >= 64) over few outputs: 64 split-lanes x 4 float4 columns per block, 4 independent 16-byte loads in
// flight per lane; the 16 split-lanes of a wave fold with shuffles, the 4 waves through LDS.  Deterministic.
__device__ __forceinline__ void splitk_reduce_wide_body(const float* __restrict__ slab, const RGroups& rg, long total, int S, int N,
        const float* __restrict__ bias, int act, float slope, int accumulate, long n1, const ActMul& am, int bx, int by, float* lds) {
    slab += by * rg.slab_gs;
    float* __restrict__ out = rg.out[by];
    float* __restrict__ out2 = rg.out2[by];
    const int il = threadIdx.x & 3, sl = threadIdx.x >> 2;
    const long i = ((long)bx * 4 + il) * 4;
    f32x4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (i < total)
        for (int z = sl; z < S; z += 256) {
            f32x4 x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int zz = z + u * 64;
                x[u] = zz < S ? *reinterpret_cast<const f32x4*>(slab + (long)zz * total + i) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] += x[u];
        }
    f32x4 v = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float s = v[j];
        s += __shfl_xor(s, 4, 64);
        s += __shfl_xor(s, 8, 64);
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        v[j] = s;
    }
    f32x4(*sh)[4] = reinterpret_cast<f32x4(*)[4]>(lds);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < 4) sh[w][il] = v;
    __syncthreads();
    if (threadIdx.x < 4 && i < total) {
        v = (sh[0][il] + sh[1][il]) + (sh[2][il] + sh[3][il]);
        float* dst = i < n1 ? out + i : out2 + (i - n1);  // (n1 % 4 == 0: a quad never straddles the two)
        f32x4 o = accumulate ? *reinterpret_cast<const f32x4*>(dst) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float r = v[j];
            if (bias) r += bias[(i + j) % N];
            o[j] += actmul_on(am) ? actmul_apply(am, r, i + j) : apply_act(r, act, slope);
        }
        *reinterpret_cast<f32x4*>(dst) = o;
    }
}
__global__ __launch_bounds__(256) void splitk_reduce_wide(const float* __restrict__ slab, RGroups rg, long total, int S, int N,
        const float* __restrict__ bias, int act, float slope, int accumulate, long n1, ActMul am) {
    __shared__ __attribute__((aligned(16))) float lds[64];
    splitk_reduce_wide_body(slab, rg, total, S, N, bias, act, slope, accumulate, n1, am, blockIdx.x, blockIdx.y, lds);
}

// 8 <= S < 64 slabs over MANY outputs (>= 2^19): a thread owns four consecutive outputs and walks the slabs with four 16-byte loads
// in flight -- the 16-split-lane kernel above reads 4-byte pieces in 64-byte runs, which is what a few hundred outputs per CU need
// to fill the chip but half the achievable rate on the 2-10 MB results of the C3-C5 weight gradients (14.4 us for 33 MB of slabs).
__device__ __forceinline__ void splitk_reduce_vec_body(const float* __restrict__ slab, const RGroups& rg, long total, int S, int N,
        const float* __restrict__ bias, int act, float slope, int accumulate, long n1, const ActMul& am, int bx, int by) {
    slab += by * rg.slab_gs;
    float* __restrict__ out = rg.out[by];
    float* __restrict__ out2 = rg.out2[by];
    const long i = ((long)bx * 256 + threadIdx.x) * 4;
    if (i >= total) return;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    int z = 0;
    for (; z + 3 < S; z += 4) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(slab + (long)z * total + i),
                    a1 = *reinterpret_cast<const f32x4*>(slab + (long)(z + 1) * total + i),
                    a2 = *reinterpret_cast<const f32x4*>(slab + (long)(z + 2) * total + i),
                    a3 = *reinterpret_cast<const f32x4*>(slab + (long)(z + 3) * total + i);
        v += (a0 + a1) + (a2 + a3);
    }
    for (; z < S; ++z) v += *reinterpret_cast<const f32x4*>(slab + (long)z * total + i);
    float* dst = i < n1 ? out + i : out2 + (i - n1);  // (n1 % 4 == 0: a quad never straddles the two)
    f32x4 o = accumulate ? *reinterpret_cast<const f32x4*>(dst) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float r = v[j];
        if (bias) r += bias[(i + j) % N];
        o[j] += actmul_on(am) ? actmul_apply(am, r, i + j) : apply_act(r, act, slope);
    }
    *reinterpret_cast<f32x4*>(dst) = o;
}
__global__ __launch_bounds__(256) void splitk_reduce_vec(const float* __restrict__ slab, RGroups rg, long total, int S, int N,
        const float* __restrict__ bias, int act, float slope, int accumulate, long n1, ActMul am) {
    splitk_reduce_vec_body(slab, rg, total, S, N, bias, act, slope, accumulate, n1, am, blockIdx.x, blockIdx.y);
}

// few slabs over many outputs: one thread per output, grid-stride
__device__ __forceinline__ void splitk_reduce_flat_body(const float* __restrict__ slab, const RGroups& rg, long total, int S, int N,
        const float* __restrict__ bias, int act, float slope, int accumulate, long n1, const ActMul& am, int bx, int by, int nbx) {
    slab += by * rg.slab_gs;
    float* __restrict__ out = rg.out[by];
    float* __restrict__ out2 = rg.out2[by];
    const long stride = (long)nbx * 256;
    for (long i = (long)bx * 256 + threadIdx.x; i < total; i += stride) {
        float v = 0.f;
        for (int z = 0; z < S; ++z) v += slab[(long)z * total + i];
        if (bias) v += bias[i % N];
        v = actmul_on(am) ? actmul_apply(am, v, i) : apply_act(v, act, slope);
        float* o = i < n1 ? out + i : out2 + (i - n1);
        *o = accumulate ? *o + v : v;
    }
}
__global__ __launch_bounds__(256) void splitk_reduce_flat(const float* __restrict__ slab, RGroups rg, long total, int S, int N,
        const float* __restrict__ bias, int act, float slope, int accumulate, long n1, ActMul am) {
    splitk_reduce_flat_body(slab, rg, total, S, N, bias, act, slope, accumulate, n1, am, blockIdx.x, blockIdx.y, gridDim.x);
}

// ---- a weight-gradient reduce riding on a LATER launch ---------------------------------------------------
// Nobody reads a weight gradient before the aggregation / the optimizer, so the reduce of layer i's slabs need not be a launch of
// its own on the backward's dependency chain: it is parked (host side: g_defer) and the next implicit-GEMM launch -- layer i-1's
// input gradient / pair -- carries it as extra blocks behind its own.  Same arithmetic, same order as the stand-alone kernels.
struct RSide {
    const float* slab;
    RGroups rg;
    long total, n1;
    int S, N, accumulate;
    int kind;  // 0 wide, 1 vec, 2 plain (16 split lanes), 3 flat
    int nbx;   // blocks per group (the stand-alone launch's grid x); blocks in all: nbx * groups
    int nblk;  // 0: nothing rides on this launch
};
__device__ __forceinline__ void side_reduce(const RSide& r, int bid, float* lds) {
    const int bx = bid % r.nbx, by = bid / r.nbx;
    const ActMul none{nullptr, 0, 0.f, 0, 0, nullptr};
    if (r.kind == 0) splitk_reduce_wide_body(r.slab, r.rg, r.total, r.S, r.N, nullptr, 0, 0.f, r.accumulate, r.n1, none, bx, by, lds);
    else if (r.kind == 1) splitk_reduce_vec_body(r.slab, r.rg, r.total, r.S, r.N, nullptr, 0, 0.f, r.accumulate, r.n1, none, bx, by);
    else if (r.kind == 2) splitk_reduce_body(r.slab, r.rg, r.total, r.S, r.N, nullptr, 0, 0.f, r.accumulate, r.n1, none, bx, by, lds);
    else splitk_reduce_flat_body(r.slab, r.rg, r.total, r.S, r.N, nullptr, 0, 0.f, r.accumulate, r.n1, none, bx, by, r.nbx);
}

// BWD form with per-class split factors: pixel (ho, wo) belongs to output-parity class (ho % s) * s + (wo % s) and only the
// first scls[class] slabs hold its partial sums (the others were never written for that pixel).  One float4 per thread.
struct ClsSplit {
    int s[4];
};
__global__ __launch_bounds__(256) void splitk_reduce_cls(const float* __restrict__ slab, float* __restrict__ out, long total, int N,
                                                         int Ho, int Wo, int stride, ClsSplit scls, const float* __restrict__ bias,
                                                         int act, float slope, ActMul am) {
    const long nv = total / 4, gstride = (long)gridDim.x * blockDim.x;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nv; q += gstride) {
        const long i = q * 4, p = i / N;
        const int n = (int)(i - p * N);
        const int wo = (int)(p % Wo), ho = (int)((p / Wo) % Ho);
        const int S = scls.s[(ho % stride) * stride + (wo % stride)];
        f32x4 v = *reinterpret_cast<const f32x4*>(slab + i);
        for (int z = 1; z < S; ++z) v += *reinterpret_cast<const f32x4*>(slab + (long)z * total + i);
        if (actmul_on(am)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = actmul_apply(am, v[j] + (bias ? bias[n + j] : 0.f), i + j);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j] + (bias ? bias[n + j] : 0.f), act, slope);
        }
        *reinterpret_cast<f32x4*>(out + i) = v;
    }
}

// Split-K reduction that also emits the column statistics of its result (the BatchNorm that follows a split-K layer needs no
// pass of its own): out[m][n] = sum_z slab[z][m][n] + bias[n]; part[(blk * 2 + {0,1}) * N + n] = sum / sum of squares over the
// block's rows.  Thread t owns column quad t %% NQ (NQ = N / 4 divides 256) and every (256 / NQ)-th row of the block's
// rows_per_block rows.  stride > 0: BWD-form slabs with per-class split counts (splitk_reduce_cls); else S slabs everywhere.
__global__ __launch_bounds__(256) void splitk_reduce_stats(const float* __restrict__ slab, float* __restrict__ out, int M, int N, int S,
                                                           ClsSplit scls, int Ho, int Wo, int stride, const float* __restrict__ bias,
                                                           float* __restrict__ part, int rows_per_block, BnBwd bb) {
    __shared__ float sh[2][4][256];
    const int t = threadIdx.x, NQ = N >> 2, RG = 256 / NQ;
    const int cq = t % NQ, rg = t / NQ;
    const long total = (long)M * N;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    const f32x4 b4 = bias ? *reinterpret_cast<const f32x4*>(bias + cq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    // BnBwd mode (bb.y): the reduced value is dout; the sums are (d, d * y) with d = dout * act'(scale * y + shift)
    f32x4 sc4 = f32x4{0.f, 0.f, 0.f, 0.f}, sh4 = sc4;
    if (bb.y) {
        sc4 = *reinterpret_cast<const f32x4*>(bb.scale + cq * 4);
        sh4 = *reinterpret_cast<const f32x4*>(bb.shift + cq * 4);
        part = bb.part;
    }
    f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f}, q4 = s4;
    for (int r = r0 + rg; r < r1; r += RG) {
        int Sr = S;
        if (stride > 0) {
            const int wo = r % Wo, ho = (r / Wo) % Ho;
            Sr = scls.s[(ho % stride) * stride + (wo % stride)];
        }
        const long i = (long)r * N + cq * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(slab + i);
        int z = 1;
        for (; z + 3 < Sr; z += 4) {  // four slabs in flight per trip (a dependent chain of S loads is what made this kernel slow)
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(slab + (long)z * total + i),
                        a1 = *reinterpret_cast<const f32x4*>(slab + (long)(z + 1) * total + i),
                        a2 = *reinterpret_cast<const f32x4*>(slab + (long)(z + 2) * total + i),
                        a3 = *reinterpret_cast<const f32x4*>(slab + (long)(z + 3) * total + i);
            v += (a0 + a1) + (a2 + a3);
        }
        for (; z < Sr; ++z) v += *reinterpret_cast<const f32x4*>(slab + (long)z * total + i);
        v += b4;
        if (out) *reinterpret_cast<f32x4*>(out + i) = v;
        if (bb.y) {
            const f32x4 y4 = *reinterpret_cast<const f32x4*>(bb.y + (long)(r % bb.rows_per_group) * N + cq * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z = fmaf(y4[j], sc4[j], sh4[j]);
                const float d = v[j] * (z > 0.f ? 1.f : bb.slope);
                s4[j] += d;
                q4[j] = fmaf(d, y4[j], q4[j]);
            }
        } else {
            s4 += v;
#pragma unroll
            for (int j = 0; j < 4; ++j) q4[j] = fmaf(v[j], v[j], q4[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) sh[0][j][t] = s4[j], sh[1][j][t] = q4[j];
    __syncthreads();
    if (t < NQ) {  // fixed-order fold over the row groups
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = 0.f, q = 0.f;
            for (int i = 0; i < RG; ++i) s += sh[0][j][i * NQ + t], q += sh[1][j][i * NQ + t];
            s4[j] = s, q4[j] = q;
        }
        *reinterpret_cast<f32x4*>(part + ((long)blockIdx.x * 2 + 0) * N + t * 4) = s4;
        *reinterpret_cast<f32x4*>(part + ((long)blockIdx.x * 2 + 1) * N + t * 4) = q4;
    }
}

const char* g_last_kernel = "";  // movae_bench_last_kernel(): main kernel chosen by the most recent conv-family dispatch
int g_force_split = 0;           // movae_bench_force_split(): > 0 pins the split-K factor (tuning sweeps)
bool g_bench_main_only = false;  // movae_bench_main_kernel_only(): time the MFMA kernel without its epilogue launches
int g_compute_bf16 = 0;          // movae_set_compute_dtype(): 1 = bf16 operands (fp32 accumulate) in the 128x128 implicit-GEMM kernels
int g_force_kgemm = 0;           // movae_bench_force_kgemm(): 1 = every shape kgemm.h can serve takes it, -1 = none does (tests, A/B)

inline long reduce_vec_min() {  // outputs from which the 16-byte reduce serves 8 <= S < 64 (tuning knob)
    static const long v = getenv("MOVAE_REDUCE_VEC_MIN") ? atol(getenv("MOVAE_REDUCE_VEC_MIN")) : (1L << 19);
    return v;
}

// ---- host side of the parked reduce ------------------------------------------------------------------------------------------
// movae_reduce_defer(1) (ops.py: the destination is a gradient sink nobody reads before the aggregation, and the call's scratch
// arena is not touched by the calls that follow) arms the NEXT weight-gradient call: its split-K reduce is parked instead of
// launched.  The next implicit-GEMM launch on the same stream takes it along (defer_take -> extra blocks); movae_reduce_flush(),
// any second parked reduce, or a call that cannot carry it launch it stand-alone.  At most one is parked.
struct DeferredReduce {
    size_t bytes = 0;     // of the parked reduce's slabs
    bool armed = false, pending = false;
    unsigned serial = 0;  // counts parked reduces (DeferScope: was the one parked at entry carried?)
    RSide r{};
    hipStream_t st = nullptr;
};
// PROCESS-wide, not thread_local like the per-call plans (g_pending, g_kpend): the reduce is parked by one library call and taken
// by a later one, and torch runs a backward's nodes on its autograd thread while the caller that flushes sits on another.  The
// calls themselves are sequential (one stream, one step); the arming flag travels with the thread that makes the armed call.
static DeferredReduce g_defer;
static long g_defer_stats[3] = {0, 0, 0};  // parked, carried by a later launch, launched stand-alone after all

struct DeferArmScope {  // a weight-gradient entry point: remembers what was parked at entry, disarms at exit
    unsigned serial = g_defer.serial;
    ~DeferArmScope() { g_defer.armed = false; }
};

inline int reduce_launch(const RSide& r, const float* bias, int act, float slope, const ActMul& am, hipStream_t st) {
    const dim3 grid((unsigned)r.nbx, (unsigned)(r.nblk / r.nbx));
    if (r.kind == 0)
        hipLaunchKernelGGL(splitk_reduce_wide, grid, dim3(256), 0, st, r.slab, r.rg, r.total, r.S, r.N, bias, act, slope, r.accumulate, r.n1, am);
    else if (r.kind == 1)
        hipLaunchKernelGGL(splitk_reduce_vec, grid, dim3(256), 0, st, r.slab, r.rg, r.total, r.S, r.N, bias, act, slope, r.accumulate, r.n1, am);
    else if (r.kind == 2)
        hipLaunchKernelGGL(splitk_reduce, grid, dim3(256), 0, st, r.slab, r.rg, r.total, r.S, r.N, bias, act, slope, r.accumulate, r.n1, am);
    else
        hipLaunchKernelGGL(splitk_reduce_flat, grid, dim3(256), 0, st, r.slab, r.rg, r.total, r.S, r.N, bias, act, slope, r.accumulate, r.n1, am);
    MOVAE_CHECK_LAUNCH("splitk_reduce");
    return MOVAE_OK;
}

inline int defer_flush() {
    if (!g_defer.pending) return MOVAE_OK;
    g_defer.pending = false;
    ++g_defer_stats[2];
    return reduce_launch(g_defer.r, nullptr, 0, 0.f, ActMul{nullptr, 0, 0.f, 0, 0, nullptr}, g_defer.st);
}

// a parked reduce is a few blocks' work in a launch built for something else (its occupancy, not the reduce's): the size gates.
// Paired launches (100-130 registers, 3-4 blocks per CU) carry more than the tiled 3-D ones (33 MB of slabs behind a 185 us
// igemm2_bwd<128,64> cost it 49 us).  A reduce its next carrier refuses is launched stand-alone by that carrier's call.
static size_t g_defer_max_bytes = getenv("MOVAE_DEFER_MAX_BYTES") ? (size_t)atol(getenv("MOVAE_DEFER_MAX_BYTES")) : (size_t)(12u << 20);
static size_t g_defer_max_bytes_pair = getenv("MOVAE_DEFER_MAX_BYTES_PAIR") ? (size_t)atol(getenv("MOVAE_DEFER_MAX_BYTES_PAIR")) : (size_t)(12u << 20);
inline size_t defer_max_bytes() { return g_defer_max_bytes; }
inline size_t defer_max_bytes_pair() { return g_defer_max_bytes_pair; }

// the parked reduce, if this launch (on stream st; `pair`: a paired dgrad + wgrad launch) can carry it; r.nblk == 0 otherwise
inline RSide defer_take(hipStream_t st, bool pair = true) {
    RSide none{};
    if (!g_defer.pending || g_defer.st != st || g_bench_main_only) return none;
    if (g_defer.bytes > (pair ? defer_max_bytes_pair() : defer_max_bytes())) {
        (void)defer_flush();  // too large for this carrier: stand-alone, in front of it
        return none;
    }
    g_defer.pending = false;
    ++g_defer_stats[1];
    return g_defer.r;
}

// a 3-D launch takes the parked reduce as whole extra z layers BEHIND its own (*gz = the main problem's z extent, g.z grows).
// (In front -- the reduce's blocks dispatched first -- measured worse: C2 0.871 vs 0.823 ms, C4 4.72 vs 4.49.)  The hardware's z
// limit can refuse (a reduce of many blocks behind a one-tile layer): the reduce stays parked then
inline RSide defer_take_3d(hipStream_t st, dim3* g, int* gz) {
    *gz = (int)g->z;
    RSide none{};
    if (!g_defer.pending || g_defer.st != st || g_bench_main_only) return none;
    const long layers = ((long)g_defer.r.nblk + (long)g->x * g->y - 1) / ((long)g->x * g->y);
    if ((long)g->z + layers > 65535) return none;
    if (g_defer.bytes > defer_max_bytes()) {
        (void)defer_flush();
        return none;
    }
    g->z += (unsigned)layers;
    return defer_take(st, false);
}

inline int launch_reduce_groups(const float* slab, const RGroups& rg, int G, long n1, long n2, int S, int N, const float* bias, int act,
                                float slope, int accumulate, hipStream_t st, ActMul am = ActMul{nullptr, 0, 0.f, 0, 0, nullptr},
                                bool deferrable = false) {
    if (g_bench_main_only) return MOVAE_OK;
    const long total = n1 + n2;  // floats per slab: n1 outputs for `out`, then n2 for `out2`
    uintptr_t al = reinterpret_cast<uintptr_t>(slab) | (uintptr_t)(rg.slab_gs * 4);
    for (int i = 0; i < G; ++i) al |= reinterpret_cast<uintptr_t>(rg.out[i]) | reinterpret_cast<uintptr_t>(rg.out2[i]);
    const bool al16 = (al & 15) == 0;
    RSide r{slab, rg, total, n1, S, N, accumulate, 0, 0, 0};
    if (S >= 64 && total % 4 == 0 && n1 % 4 == 0 && total <= (1L << 20) && al16) {
        r.kind = 0, r.nbx = (int)ceil_div(total, 16);
    } else if (S >= 8 && total >= reduce_vec_min() && total % 4 == 0 && n1 % 4 == 0 && al16) {
        r.kind = 1, r.nbx = (int)ceil_div(total / 4, 256);
    } else if (S >= 8 && total <= (1L << 20)) {
        r.kind = 2, r.nbx = (int)ceil_div(total, 16);
    } else {
        long gq = (total + 255) / 256;
        if (gq > 4096) gq = 4096;
        r.kind = 3, r.nbx = (int)gq;
    }
    r.nblk = r.nbx * G;
    if (deferrable && g_defer.armed && !bias && act == MOVAE_ACT_NONE && !am.y && !am.res &&
        (size_t)total * S * G * sizeof(float) <= (defer_max_bytes() > defer_max_bytes_pair() ? defer_max_bytes() : defer_max_bytes_pair())) {
        if (int rc = defer_flush()) return rc;  // (one slot)
        g_defer.r = r, g_defer.st = st, g_defer.pending = true, g_defer.bytes = (size_t)total * S * G * sizeof(float);
        ++g_defer.serial;
        ++g_defer_stats[0];
        return MOVAE_OK;
    }
    return reduce_launch(r, bias, act, slope, am, st);
}

inline int launch_reduce(const float* slab, float* out, long n1, int S, int N, const float* bias, int act, float slope,
                         int accumulate, hipStream_t st, float* out2 = nullptr, long n2 = 0, ActMul am = ActMul{nullptr, 0, 0.f, 0, 0, nullptr}) {
    if (!out2) n2 = 0;
    RGroups rg{};
    rg.out[0] = out, rg.out2[0] = out2, rg.slab_gs = 0;
    return launch_reduce_groups(slab, rg, 1, n1, n2, S, N, bias, act, slope, accumulate, st, am);
}

// The reduce of a split-K forward whose result feeds a BatchNorm: sums, adds the bias and emits the column statistics.
// Returns false when the shape does not fit the kernel (the caller then reduces plainly and no statistics are produced).
inline bool reduce_stats_shape_ok(const void* slab, const void* out, const void* part, const void* bias, long M, int N) {
    return N % 4 == 0 && N <= 1024 && 256 % (N / 4) == 0 && M <= 0x7fffffffL &&
           ((reinterpret_cast<uintptr_t>(slab) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(part)) & 15) == 0 &&
           (!bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0);
}

inline bool launch_reduce_stats(const float* slab, float* out, long M, int N, int S, const ClsSplit* scls, int Ho, int Wo, int stride,
                                const float* bias, hipStream_t st) {
    if (!g_fuse.stats || !reduce_stats_shape_ok(slab, out, g_fuse.stats, bias, M, N)) return false;
    const int RG = 256 / (N / 4);
    long rpb = 2L * RG;                                  // two rows per thread ...
    if ((M + rpb - 1) / rpb > 1024) rpb = ((M + 1023) / 1024 + RG - 1) / RG * RG;  // ... at most 1024 partials
    const long nblk = (M + rpb - 1) / rpb;
    float* part = fuse_stats_claim(nblk, N);
    if (!part) return false;
    ClsSplit one{{S, S, S, S}};
    hipLaunchKernelGGL(splitk_reduce_stats, dim3((unsigned)nblk), dim3(256), 0, st, slab, out, (int)M, N, S, scls ? *scls : one, Ho, Wo,
                       scls ? stride : 0, bias, part, (int)rpb, BnBwd{});
    return true;
}

// Plan of the same reduce in BnBwd mode (the reduced tensor is a fused BatchNorm's output gradient): rows per block such that
// no block straddles two cotangent groups.  Fills `bb` (part / rows_per_group / ppg) and claims the partials; false = not possible.
inline bool plan_reduce_bnbwd(long M, int N, BnBwd* bb, int* rows_per_block) {
    if (!g_fuse.bn_y || !g_fuse.bn_part || N % 4 != 0 || N > 1024 || 256 % (N / 4) != 0 || M > 0x7fffffffL || M % g_fuse.bn_groups != 0)
        return false;
    const long rpg = M / g_fuse.bn_groups;
    const int RG = 256 / (N / 4);
    long rpb = 2L * RG;
    while (rpb > RG && rpg % rpb != 0) rpb -= RG;
    if (rpg % rpb != 0) return false;
    while (rpg / rpb * g_fuse.bn_groups > 2048 && rpg % (rpb * 2) == 0) rpb *= 2;
    float* part = fuse_bn_claim(rpg / rpb, N);
    if (!part || ((reinterpret_cast<uintptr_t>(part) | reinterpret_cast<uintptr_t>(g_fuse.bn_y) | reinterpret_cast<uintptr_t>(g_fuse.bn_scale) |
                   reinterpret_cast<uintptr_t>(g_fuse.bn_shift)) & 15) != 0) {
        g_fuse.bn_ppg = 0;
        return false;
    }
    *bb = BnBwd{g_fuse.bn_y, g_fuse.bn_scale, g_fuse.bn_shift, g_fuse.bn_slope, part, (int)rpg, (int)(rpg / rpb)};
    *rows_per_block = (int)rpb;
    return true;
}

inline void launch_reduce_bnbwd(const float* slab, float* out, long M, int N, int S, const ClsSplit* scls, int Ho, int Wo, int stride,
                                const BnBwd& bb, int rows_per_block, hipStream_t st) {
    ClsSplit one{{S, S, S, S}};
    hipLaunchKernelGGL(splitk_reduce_stats, dim3((unsigned)((M + rows_per_block - 1) / rows_per_block)), dim3(256), 0, st, slab, out, (int)M,
                       N, S, scls ? *scls : one, Ho, Wo, scls ? stride : 0, (const float*)nullptr, (float*)nullptr, rows_per_block, bb);
}

// split-K factor from a small cost model fitted to tools/conv_microbench.py --sweep-split on MI355X (tools/split_model.py
// holds the same rule and the fit):   T(S) = rounds * (t0 + k-tiles-per-split * tk * share) + [S>1] * (t_reduce + S * out * c)
//   * 256 CUs hold two 256-thread blocks each; a CU holding two runs each ~1.6x slower, so 256 < blocks < 512 is the worst
//     place to be and blocks <= 256 the best unless the k loop per block stays long;
//   * every split writes and re-reads the whole output once (slab), ~0.87 us per MB at the rate these short kernels reach;
//   * tk = time of one 32-deep k-tile for a 64x64 block (~1 us: latency bound, one block per CU cannot hide the
//     global->LDS->MFMA chain), scaled by tile area.
enum { FORM_FWD = 0, FORM_BWD = 1, FORM_WGRAD = 2 };
inline int choose_split(int form, int tile_area, int bk, long tiles, int nk, size_t per_slab_bytes, size_t ws_bytes, bool have_ws) {
    if (!have_ws || nk < 2) return 1;
    long smax = nk < 256 ? nk : 256;
    while (smax > 1 && per_slab_bytes * (size_t)smax > ws_bytes) --smax;
    if (g_force_split > 0) return (int)(g_force_split < smax ? g_force_split : smax);  // movae_bench_force_split()
    static const int legacy_tiles = getenv("MOVAE_SPLIT_TILES") ? atoi(getenv("MOVAE_SPLIT_TILES")) : -1;
    if (legacy_tiles >= 0 && tiles > legacy_tiles) return 1;
    const bool big = tile_area >= 128 * 128;
    const double form_tk = form == FORM_FWD ? 1.0 : (form == FORM_BWD ? 1.25 : 0.9);
    const double tk = form_tk * (tile_area / 4096.0) * (big ? 0.83 : 1.0) * (bk / 32.0);
    const double out_mb = (double)per_slab_bytes * 1e-6;
    int best = 1;
    double best_t = 0.0;
    for (long S = 1; S <= smax; ++S) {
        const long kps = (nk + S - 1) / S;
        if ((nk + kps - 1) / kps != S) continue;  // the launchers round S to this value anyway
        const long blocks = tiles * S;
        const long rounds = (blocks + 511) / 512;
        const double share = blocks <= 256 ? 1.0 : 1.6;
        double t = rounds * (2.0 + kps * tk * share);
        if (S > 1) t += 2.5 + S * out_mb * 0.87;
        if (S == 1 || t < best_t * 0.97) {  // fewer splits unless the gain is clear
            best = (int)S;
            best_t = t;
        }
    }
    return best;
}

// ------------------------------------------------------------------------------------------------
// host dispatch
// ------------------------------------------------------------------------------------------------
#include "igemm_v2.h"
#include "conv_thin.h"
#include "linear_small.h"
#include "kgemm.h"
namespace kg {
inline bool g_force_kgemm_on() { return g_force_kgemm > 0; }
}

inline bool is_linear(const Geom& g) {
    return g.KH == 1 && g.KW == 1 && g.Hi == 1 && g.Wi == 1 && g.Ho == 1 && g.Wo == 1 && g.stride == 1 && g.pad == 0 && g.wlen == 0;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline long big_tile_min() {  // minimum number of 128x128 work items before the big tile is preferred
    static const long v = getenv("MOVAE_BIG_TILE_MIN") ? atol(getenv("MOVAE_BIG_TILE_MIN")) : 256;
    return v;
}

template <int BM, int BN>
int launch_fwd_t(const float* X, const float* W, float* Y, const Geom& g, const Epilogue& ep, int M, int K, bool vec,
                 void* ws, size_t ws_bytes, hipStream_t st) {
    const int gx = ceil_div(M, BM), gy = ceil_div(g.Nn, BN);
    const int nk = ceil_div(K, BK);
    const long tiles = (long)gx * gy;
    int S = choose_split(FORM_FWD, BM * BN, BK, tiles, nk, (size_t)M * g.Nn * sizeof(float), ws_bytes, ws != nullptr);
    const int per_split = ceil_div(nk, S);
    S = ceil_div(nk, per_split);
    float* slab = S > 1 ? static_cast<float*>(ws) : nullptr;
    dim3 grid(gx, gy, S);
    if (vec)
        hipLaunchKernelGGL((igemm_fwd<BM, BN, true>), grid, dim3(256), 0, st, X, W, Y, g, ep, M, K, per_split, slab);
    else
        hipLaunchKernelGGL((igemm_fwd<BM, BN, false>), grid, dim3(256), 0, st, X, W, Y, g, ep, M, K, per_split, slab);
    MOVAE_CHECK_LAUNCH("igemm_fwd");
    if (S > 1) {
        const long total = (long)M * g.Nn;
        if (int rc = launch_reduce(slab, Y, total, S, g.Nn, ep.bias, ep.act, ep.slope, 0, st)) return rc;
    }
    return MOVAE_OK;
}

// A 1x1 output grid reads tap (kh, kw) at input (kh - pad, kw - pad) for every image alike: taps outside the image multiply
// padding only.  The window of taps that can hit the image becomes the kernel (pad 0) and the stored weights are addressed
// through Geom::wrow/wlen/wstride/woff.  conv 2x2x256 -> 1x1x512 (k3 s2 p1): K 2304 -> 1024.
inline bool tap_window(const Geom& g, Geom* out) {
    static const bool enabled = !getenv("MOVAE_NO_TAPWIN");
    if (!enabled || g.Ho != 1 || g.Wo != 1 || g.pad >= g.KH || g.pad >= g.KW) return false;
    const int nkh = g.KH - g.pad < g.Hi ? g.KH - g.pad : g.Hi, nkw = g.KW - g.pad < g.Wi ? g.KW - g.pad : g.Wi;
    if (nkh < 1 || nkw < 1 || nkh * nkw == g.KH * g.KW) return false;
    *out = g;
    out->KH = nkh, out->KW = nkw, out->pad = 0;
    out->wrow = g.KH * g.KW * g.Cr;
    out->wlen = nkw * g.Cr, out->wstride = g.KW * g.Cr, out->woff = (g.pad * g.KW + g.pad) * g.Cr;
    return true;
}

int launch_fwd(const float* X, const float* W, float* Y, const Geom& g_in, const Epilogue& ep, void* ws, size_t ws_bytes,
               hipStream_t st) {
    Geom g = g_in;
    {
        const long Kfull = (long)g.KH * g.KW * g.Cr;
        Geom gw;
        // (fast MFMA path only: the other kernels address dense weights)
        if (g.Cr % 4 == 0 && aligned16(X) && aligned16(W) && !is_linear(g) && !thin::thin_in_ok(g) && g.Nn > 4 && Kfull <= 0x7fffffffL &&
            tap_window(g, &gw))
            g = gw;
    }
    const long Ml = (long)g.Nimg * g.Ho * g.Wo;
    const long Kl = (long)g.KH * g.KW * g.Cr;
    if (Ml <= 0 || g.Nn <= 0 || Kl <= 0 || Ml > 0x7fffffffL) {
        movae_set_error("conv fwd-form: bad shape M=%ld N=%d K=%ld", Ml, g.Nn, Kl);
        return MOVAE_EINVAL;
    }
    const int M = (int)Ml, K = (int)Kl;
    if (is_linear(g) && lin::linear_small_ok(Ml, g.Nn, Kl) && aligned16(X) && aligned16(W)) {
        MOVAE_NO_NORM("linear (fwd)");
        float* ys[1] = {Y};
        return (g_last_kernel = "linear_small_k<NT>",
                lin::launch_linear_small<0>(X, W, ys, nullptr, 1, 0, ep.bias, M, g.Nn, K, ep.act, ep.slope, 0, st));
    }
    if (thin::thin_in_ok(g)) {
        MOVAE_NO_NORM("thin-channel input conv");
        return (g_last_kernel = "thin_in_k<fwd>", thin::launch_thin_in<false>(X, W, Y, g, ep, st));
    }
    if (g.Nn <= 4) {  // 3-channel output: LDS-tiled direct kernel (applies the fused input transform while staging)
        bool handled = false;
        if (int rc = thin::launch_thin_out_tile<false>(X, W, Y, g, ep, st, &handled)) return rc;
        if (handled) return (g_last_kernel = "thin_out_tile_k<fwd>", MOVAE_OK);
    }
    if (thin::thin_out_ok(g, X)) {
        MOVAE_NO_NORM("thin-channel output conv");
        return (g_last_kernel = "thin_out_fwd_k", thin::launch_thin_out_fwd(X, W, Y, g, ep, st));
    }
    // fast path (igemm_v2.h); its 32-bit buffer offsets need every row block's images and the weights within 2 GiB of their bases
    const bool span_ok = v2::buf_span_ok((128L / ((long)g.Ho * g.Wo) + 2) * g.Hi * g.Wi * g.Cr) &&
                         v2::buf_span_ok((long)g.Nn * (g.wlen ? g.wrow : Kl) + g.woff);
    if (g_force_kgemm >= 0) {  // small, latency-bound problems: the reduction split inside the block (kgemm.h)
        bool handled = false;
        if (int rc = kg::launch_kfwd(X, W, Y, g, ep, M, K, st, &handled)) return rc;
        if (handled) return MOVAE_OK;
    }
    if (g.Cr % 4 == 0 && aligned16(X) && aligned16(W) && span_ok) {
        if (g.Nn <= 32) return (g_last_kernel = "igemm2_fwd<128,32>", v2::launch_fwd2<128, 32>(X, W, Y, g, ep, M, K, ws, ws_bytes, st));
        // 2x2 register tiling (64x64 per wave) once the 128x128 grid alone fills the chip
        if (g.Nn >= 128 && (Ml / 128) * (g.Nn / 128) >= big_tile_min()) return (g_last_kernel = g_compute_bf16 ? "igemm2_fwd<128,128,true>" : "igemm2_fwd<128,128>", v2::launch_fwd2<128, 128>(X, W, Y, g, ep, M, K, ws, ws_bytes, st));
        if (Ml >= 128 * 512) return (g_last_kernel = "igemm2_fwd<128,64>", v2::launch_fwd2<128, 64>(X, W, Y, g, ep, M, K, ws, ws_bytes, st));
        return (g_last_kernel = "igemm2_fwd<64,64>", v2::launch_fwd2<64, 64>(X, W, Y, g, ep, M, K, ws, ws_bytes, st));
    }
    MOVAE_NO_NORM("generic conv (fwd form)");
    const bool vec = (g.Cr % BK == 0) && aligned16(X) && aligned16(W);
    if (g.Nn <= 32) return (g_last_kernel = "igemm_fwd<128,32>", launch_fwd_t<128, 32>(X, W, Y, g, ep, M, K, vec, ws, ws_bytes, st));
    if (Ml >= 128 * 512) return (g_last_kernel = "igemm_fwd<128,64>", launch_fwd_t<128, 64>(X, W, Y, g, ep, M, K, vec, ws, ws_bytes, st));
    return (g_last_kernel = "igemm_fwd<64,64>", launch_fwd_t<64, 64>(X, W, Y, g, ep, M, K, vec, ws, ws_bytes, st));
}

template <int BM, int BN>
int launch_bwd_t(const float* X, const float* W, float* Y, const Geom& g, const Epilogue& ep, bool vec, void* ws,
                 size_t ws_bytes, hipStream_t st) {
    const int s = g.stride;
    const long Mmax = (long)g.Nimg * ceil_div(g.Ho, s) * ceil_div(g.Wo, s);
    const int gx = ceil_div(Mmax, BM), gy = ceil_div(g.Nn, BN);
    const int nk_max = ceil_div((long)ceil_div(g.KH, s) * ceil_div(g.KW, s) * g.Cr, BK);
    const long total = (long)g.Nimg * g.Ho * g.Wo * g.Nn;
    int S = choose_split(FORM_BWD, BM * BN, BK, (long)gx * gy * s * s, nk_max, (size_t)total * sizeof(float), ws_bytes, ws != nullptr);
    const int per_split = ceil_div(nk_max, S);
    S = ceil_div(nk_max, per_split);
    float* slab = S > 1 ? static_cast<float*>(ws) : nullptr;
    dim3 grid(gx, gy, s * s * S);
    if (vec)
        hipLaunchKernelGGL((igemm_bwd<BM, BN, true>), grid, dim3(256), 0, st, X, W, Y, g, ep, S, per_split, slab, total);
    else
        hipLaunchKernelGGL((igemm_bwd<BM, BN, false>), grid, dim3(256), 0, st, X, W, Y, g, ep, S, per_split, slab, total);
    MOVAE_CHECK_LAUNCH("igemm_bwd");
    if (S > 1) {
        if (int rc = launch_reduce(slab, Y, total, S, g.Nn, ep.bias, ep.act, ep.slope, 0, st)) return rc;
    }
    return MOVAE_OK;
}

int launch_bwd(const float* X, const float* W, float* Y, const Geom& g, const Epilogue& ep, void* ws, size_t ws_bytes,
               hipStream_t st) {
    const long Ml = (long)g.Nimg * g.Ho * g.Wo;
    if (Ml <= 0 || g.Nn <= 0 || g.Cr <= 0 || Ml > 0x7fffffffL) {
        movae_set_error("conv bwd-form: bad shape M=%ld N=%d Cr=%d", Ml, g.Nn, g.Cr);
        return MOVAE_EINVAL;
    }
    const long Mc = Ml / (g.stride * g.stride);
    if (is_linear(g) && lin::linear_small_ok(Ml, g.Nn, g.Cr) && g.Nn % 4 == 0 && aligned16(X) && aligned16(W)) {
        MOVAE_NO_NORM("linear (bwd form)");
        float* ys[1] = {Y};
        if (v2::g_pair_collect) {  // inside a dgrad + wgrad call: planned, launched together with the weight gradient (linear_bwd_k)
            lin::g_lin_pend.d = lin::make_prob<1>(X, nullptr, W, nullptr, ys, nullptr, 1, 1, 0, ep.bias, nullptr, (int)Ml, g.Nn, g.Cr, ep.act,
                                                  ep.slope, 0);
            lin::g_lin_pend.active = true;
            g_last_kernel = "linear_small_k<NN>";
            return MOVAE_OK;
        }
        return (g_last_kernel = "linear_small_k<NN>",
                lin::launch_linear_small<1>(X, W, ys, nullptr, 1, 0, ep.bias, (int)Ml, g.Nn, g.Cr, ep.act, ep.slope, 0, st));
    }
    if (thin::thin_in_ok(g)) {
        MOVAE_NO_NORM("thin-channel input (bwd form)");
        return (g_last_kernel = "thin_in_k<bwd>", thin::launch_thin_in<true>(X, W, Y, g, ep, st));
    }
    if (g.Nn <= 4) {  // 3-channel output of a transposed conv: LDS-tiled direct kernel
        bool handled = false;
        if (int rc = thin::launch_thin_out_tile<true>(X, W, Y, g, ep, st, &handled)) return rc;
        if (handled) return (g_last_kernel = "thin_out_tile_k<bwd>", MOVAE_OK);
    }
    // fast path (igemm_v2.h); 32-bit buffer offsets: a row block's images (smallest output-parity class) and the weights within 2 GiB
    const long hwc_min = (long)(g.Ho / g.stride > 0 ? g.Ho / g.stride : 1) * (g.Wo / g.stride > 0 ? g.Wo / g.stride : 1);
    const bool span_ok = v2::buf_span_ok((128L / hwc_min + 2) * g.Hi * g.Wi * g.Cr) && v2::buf_span_ok((long)g.Cr * g.KH * g.KW * g.Nn);
    if (g_force_kgemm >= 0) {
        bool handled = false;
        if (int rc = kg::launch_kbwd(X, W, Y, g, ep, st, &handled)) return rc;
        if (handled) return MOVAE_OK;
    }
    if (g.Cr % 4 == 0 && g.Nn % 4 == 0 && g.stride <= 2 && aligned16(X) && aligned16(W) && aligned16(Y) && span_ok) {
        if (g.Nn <= 32) return (g_last_kernel = "igemm2_bwd<128,32>", v2::launch_bwd2<128, 32>(X, W, Y, g, ep, ws, ws_bytes, st));
        if (g.Nn >= 128 && (Mc / 128) * (g.Nn / 128) * g.stride * g.stride >= big_tile_min())
            return (g_last_kernel = g_compute_bf16 ? "igemm2_bwd<128,128,true>" : "igemm2_bwd<128,128>", v2::launch_bwd2<128, 128>(X, W, Y, g, ep, ws, ws_bytes, st));
        if (Mc >= 128 * 512) return (g_last_kernel = "igemm2_bwd<128,64>", v2::launch_bwd2<128, 64>(X, W, Y, g, ep, ws, ws_bytes, st));
        return (g_last_kernel = "igemm2_bwd<64,64>", v2::launch_bwd2<64, 64>(X, W, Y, g, ep, ws, ws_bytes, st));
    }
    MOVAE_NO_NORM("generic conv (bwd form)");
    const bool vec = (g.Cr % BK == 0) && (g.Nn % 4 == 0) && aligned16(X) && aligned16(W);
    if (g.Nn <= 32) return (g_last_kernel = "igemm_bwd<128,32>", launch_bwd_t<128, 32>(X, W, Y, g, ep, vec, ws, ws_bytes, st));
    if (Mc >= 128 * 512) return (g_last_kernel = "igemm_bwd<128,64>", launch_bwd_t<128, 64>(X, W, Y, g, ep, vec, ws, ws_bytes, st));
    return (g_last_kernel = "igemm_bwd<64,64>", launch_bwd_t<64, 64>(X, W, Y, g, ep, vec, ws, ws_bytes, st));
}

template <int BM, int BN>
int launch_wgrad_t(const float* S, const float* Bg, float* dW, const WGeom& g, int K, int vec, int accumulate, void* ws,
                   size_t ws_bytes, hipStream_t st) {
    const int M = g.Cs, N = g.KH * g.KW * g.Cb;
    const int gx = ceil_div(M, BM), gy = ceil_div(N, BN);
    const long tiles = (long)gx * gy;
    int Sp = choose_split(FORM_WGRAD, BM * BN, BK, tiles, ceil_div(K, BK), (size_t)M * N * sizeof(float), ws_bytes, ws != nullptr);
    int kchunk = ceil_div(ceil_div(K, Sp), BK) * BK;
    Sp = ceil_div(K, kchunk);
    const bool slab = Sp > 1 || accumulate;
    if (slab && (!ws || (size_t)M * N * sizeof(float) * Sp > ws_bytes)) {
        movae_set_error("wgrad: workspace too small (%zu bytes) for %d splits of %dx%d", ws_bytes, Sp, M, N);
        return MOVAE_EINVAL;
    }
    float* out = slab ? static_cast<float*>(ws) : dW;
    dim3 grid(gx, gy, Sp);
    const bool va = (vec & 1) != 0, vb = (vec & 2) != 0;
    if (va && vb)
        hipLaunchKernelGGL((igemm_wgrad<BM, BN, true, true>), grid, dim3(256), 0, st, S, Bg, out, g, K, kchunk, slab ? 1 : 0);
    else if (va)
        hipLaunchKernelGGL((igemm_wgrad<BM, BN, true, false>), grid, dim3(256), 0, st, S, Bg, out, g, K, kchunk, slab ? 1 : 0);
    else if (vb)
        hipLaunchKernelGGL((igemm_wgrad<BM, BN, false, true>), grid, dim3(256), 0, st, S, Bg, out, g, K, kchunk, slab ? 1 : 0);
    else
        hipLaunchKernelGGL((igemm_wgrad<BM, BN, false, false>), grid, dim3(256), 0, st, S, Bg, out, g, K, kchunk, slab ? 1 : 0);
    MOVAE_CHECK_LAUNCH("igemm_wgrad");
    if (slab) {
        const long total = (long)M * N;
        if (int rc = launch_reduce(out, dW, total, Sp, N, nullptr, 0, 0.f, accumulate, st)) return rc;
    }
    return MOVAE_OK;
}

int launch_wgrad1(const float* S, const float* Bg, float* dW, const WGeom& g, int accumulate, void* ws, size_t ws_bytes,
                  hipStream_t st) {  // kernels without a group dimension (thin-channel ends, generic gather path)
    const long Kl = (long)g.Nimg * g.Hs * g.Ws;
    const int vec = ((g.Cs % 4 == 0 && aligned16(S)) ? 1 : 0) | ((g.Cb % 4 == 0 && aligned16(Bg)) ? 2 : 0);
    const int N = g.KH * g.KW * g.Cb;
    if (thin::thin_wgrad_ok(g) && ws) return (g_last_kernel = "thin_wgrad", thin::launch_thin_wgrad(S, Bg, dW, g, (int)Kl, accumulate, ws, ws_bytes, st));
    if (N <= 32) return (g_last_kernel = "igemm_wgrad<128,32>", launch_wgrad_t<128, 32>(S, Bg, dW, g, (int)Kl, vec, accumulate, ws, ws_bytes, st));
    return (g_last_kernel = "igemm_wgrad<64,64>", launch_wgrad_t<64, 64>(S, Bg, dW, g, (int)Kl, vec, accumulate, ws, ws_bytes, st));
}

// G cotangent groups of one layer: group i reads S + i * s_gs and Bg + i * b_gs (0 = shared operand), writes dW[i]
// colsum_S (optional): per-group destinations of sum_p S[p][a] (the bias gradient when S is dy); *colsum_done reports whether
// the chosen kernel produced it
int launch_wgrad(const float* S, const float* Bg, float* const* dW, int G, long s_gs, long b_gs, const WGeom& g, int accumulate,
                 void* ws, size_t ws_bytes, hipStream_t st, float* const* colsum_S = nullptr, bool* colsum_done = nullptr) {
    const long Kl = (long)g.Nimg * g.Hs * g.Ws;
    if (Kl <= 0 || Kl > 0x7fffffffL || g.Cs <= 0 || g.Cb <= 0) {
        movae_set_error("wgrad: bad shape K=%ld Cs=%d Cb=%d", Kl, g.Cs, g.Cb);
        return MOVAE_EINVAL;
    }
    if (colsum_done) *colsum_done = false;
    if (g.KH == 1 && g.KW == 1 && g.Hs == 1 && g.Ws == 1 && g.Hb == 1 && g.Wb == 1 && b_gs == 0 &&
        lin::linear_small_ok(g.Cs, g.Cb, (Kl + 3) / 4 * 4) && !g_bench_main_only) {  // reduction = batch rows, any count
        MOVAE_NO_NORM("linear (wgrad)");
        if (colsum_done) *colsum_done = colsum_S != nullptr;
        if (lin::g_lin_pend.active) {  // the layer's input gradient waits: one launch for both
            lin::g_lin_pend.active = false;
            const lin::LinProb wp = lin::make_prob<2>(S, nullptr, Bg, nullptr, dW, colsum_S, G, 1, s_gs, nullptr, nullptr, g.Cs, g.Cb, (int)Kl, 0,
                                                      0.f, accumulate);
            g_last_kernel = "linear_bwd_k<false>";
            return lin::launch_linear_bwd<false>(lin::g_lin_pend.d, wp, st);
        }
        return (g_last_kernel = "linear_small_k<TN>",
                lin::launch_linear_small<2>(S, Bg, dW, colsum_S, G, s_gs, nullptr, g.Cs, g.Cb, (int)Kl, 0, 0.f, accumulate, st));
    }
    // (the fast path's 32-bit buffer offsets: each group's two tensors below 2 GiB)
    const bool vec = g.Cs % 4 == 0 && g.Cb % 4 == 0 && aligned16(S) && aligned16(Bg) && s_gs % 4 == 0 && b_gs % 4 == 0 &&
                     v2::buf_span_ok(Kl * g.Cs) && v2::buf_span_ok((long)g.Nimg * g.Hb * g.Wb * g.Cb);
    const int N = g.KH * g.KW * g.Cb;
    if (g_force_kgemm >= 0 && !colsum_S && !thin::thin_wgrad_ok(g)) {  // kgemm.h (no bias-gradient column sums there)
        bool handled = false;
        if (int rc = kg::launch_kwgrad(S, Bg, dW, G, s_gs, b_gs, g, (int)Kl, accumulate, ws, ws_bytes, st, &handled)) return rc;
        if (handled) return MOVAE_OK;
    }
    if (vec && !(thin::thin_wgrad_ok(g) && ws)) {  // fast path (igemm_v2.h)
        // the column sums of S (bias gradient) ride along: the by == 0 blocks stage every S value of their split anyway
        float* const* cs = nullptr;
        if (colsum_S && !(g_fuse.nrm.scale && g_fuse.nrm_side == 1)) {
            cs = colsum_S;
            for (int i = 0; i < G; ++i)
                if (!colsum_S[i]) cs = nullptr;
        }
        if (colsum_done) *colsum_done = cs != nullptr;
        if (N <= 32) return (g_last_kernel = "igemm2_wgrad<128,32>", v2::launch_wgrad2<128, 32>(S, Bg, dW, G, s_gs, b_gs, g, (int)Kl, accumulate, ws, ws_bytes, st, cs));
        // <= 32 rows of dW (32-channel layers): a 64-row tile would multiply half a tile of padding
        if (g.Cs <= 32 && N >= 128) return (g_last_kernel = "igemm2_wgrad<32,128>", v2::launch_wgrad2<32, 128>(S, Bg, dW, G, s_gs, b_gs, g, (int)Kl, accumulate, ws, ws_bytes, st, cs));
        if (g.Cs >= 128 && (long)(g.Cs / 128) * (N / 128) * (Kl / 512) * G >= big_tile_min())
            return (g_last_kernel = g_compute_bf16 ? "igemm2_wgrad<128,128,true>" : "igemm2_wgrad<128,128>", v2::launch_wgrad2<128, 128>(S, Bg, dW, G, s_gs, b_gs, g, (int)Kl, accumulate, ws, ws_bytes, st, cs));
        // 64 rows of dW with a long reduction (C5: 32 -> 64 channels on 64 x 64 images, four cotangent groups): a 64-wide tile moves
        // 16 KiB of operands per 64 x 64 x 32 MACs -- 16 flop per byte from L2, bandwidth-bound near 0.6 of the MFMA peak; twice the
        // width reads the S operand half as often.  Unpaired (work of this size fills the chip on its own).
        static const bool wide64 = !getenv("MOVAE_NO_WGRAD_64x128");
        if (wide64 && g.Cs > 32 && g.Cs <= 64 && N >= 256 && (long)(N / 128) * (Kl / 512) * G >= big_tile_min())
            return (g_last_kernel = "igemm2_wgrad<64,128>", v2::launch_wgrad2<64, 128>(S, Bg, dW, G, s_gs, b_gs, g, (int)Kl, accumulate, ws, ws_bytes, st, cs));
        return (g_last_kernel = "igemm2_wgrad<64,64>", v2::launch_wgrad2<64, 64>(S, Bg, dW, G, s_gs, b_gs, g, (int)Kl, accumulate, ws, ws_bytes, st, cs));
    }
    if (thin::thin_wgrad_ok(g) && ws)
        return (g_last_kernel = "thin_wgrad",
                thin::launch_thin_wgrad_grouped(S, Bg, dW, G, s_gs, b_gs, g, (int)Kl, accumulate, ws, ws_bytes, st, colsum_S, colsum_done));
    MOVAE_NO_NORM("generic wgrad");
    for (int i = 0; i < G; ++i)
        if (int rc = launch_wgrad1(S + i * s_gs, Bg + i * b_gs, dW[i], g, accumulate, ws, ws_bytes, st)) return rc;
    return MOVAE_OK;
}

int check_conv_shape(const char* who, int n, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride,
                     int pad, bool transposed) {
    if (n <= 0 || hi <= 0 || wi <= 0 || ci <= 0 || ho <= 0 || wo <= 0 || co <= 0 || kh <= 0 || kw <= 0 || stride <= 0 ||
        pad < 0) {
        movae_set_error("%s: non-positive dimension", who);
        return MOVAE_EINVAL;
    }
    if (!transposed) {
        if ((hi + 2 * pad - kh) / stride + 1 != ho || (wi + 2 * pad - kw) / stride + 1 != wo) {
            movae_set_error("%s: output %dx%d inconsistent with input %dx%d k%dx%d s%d p%d", who, ho, wo, hi, wi, kh, kw,
                            stride, pad);
            return MOVAE_EINVAL;
        }
    } else {
        const int op_h = ho - ((hi - 1) * stride - 2 * pad + kh), op_w = wo - ((wi - 1) * stride - 2 * pad + kw);
        if (op_h < 0 || op_h >= stride || op_w < 0 || op_w >= stride) {
            movae_set_error("%s: transposed output %dx%d inconsistent with input %dx%d k%dx%d s%d p%d", who, ho, wo, hi, wi,
                            kh, kw, stride, pad);
            return MOVAE_EINVAL;
        }
    }
    return MOVAE_OK;
}

}  // namespace

// installs a movae_fuse_t for the dispatch made inside the scope; on exit reports the statistics partial count and clears it
struct FuseScope {
    movae_fuse_t* f;
    bool fwd;  // a forward pass (nrm_side 0): owns ep_res / ep_act_done (the backward entry points set them around their dgrad)
    explicit FuseScope(movae_fuse_t* fuse, int nrm_side) : f(fuse), fwd(nrm_side == 0) {
        g_fuse = FuseCtx();
        if (!f) return;
        f->stats_parts = 0;
        if (f->in_scale && f->in_shift) g_fuse.nrm = Norm{f->in_scale, f->in_shift, f->in_slope}, g_fuse.nrm_side = nrm_side;
        if (f->stats && f->stats_cap > 0) g_fuse.stats = f->stats, g_fuse.stats_cap = f->stats_cap;
        f->fin_done = 0;
        if (fwd && g_fuse.stats && f->fin_gamma && f->fin_beta && f->fin_out) {
            g_fuse.fin.gamma = f->fin_gamma, g_fuse.fin.beta = f->fin_beta, g_fuse.fin.out = f->fin_out;
            g_fuse.fin.running_mean = f->fin_running_mean, g_fuse.fin.running_var = f->fin_running_var, g_fuse.fin.nbt = f->fin_nbt;
            g_fuse.fin.eps = f->fin_eps, g_fuse.fin.momentum = f->fin_momentum;
        }
        // forward passes: out = conv(x) (+ bias) + ep_res -- the last conv of a residual branch adds the block's input itself
        if (fwd) {
            f->ep_act_done = 0;
            if (f->ep_res && (reinterpret_cast<uintptr_t>(f->ep_res) & 15) == 0)
                g_fuse.am = ActMul{nullptr, 0, 0.f, 0, 0, f->ep_res}, g_fuse.am_groups = 1;
        }
    }
    ~FuseScope() {
        if (f) f->stats_parts = g_fuse.stats_parts;
        if (f && fwd) f->ep_act_done = g_fuse.am_done ? 1 : 0, f->fin_done = g_fuse.fin_done ? 1 : 0;
        g_fuse = FuseCtx();
    }
};

// the BatchNorm-backward request of an input-gradient pass (movae_fuse_t::bn_*): installed for the dgrad dispatch only
inline void fuse_bn_install(movae_fuse_t* f, int groups) {
    if (!f) return;
    f->bn_ppg = 0;
    f->ep_act_done = 0;
    const bool want_act = f->ep_act_y && f->ep_act != MOVAE_ACT_NONE;
    if ((want_act || f->ep_res) && groups >= 1 &&
        ((reinterpret_cast<uintptr_t>(f->ep_act_y) | reinterpret_cast<uintptr_t>(f->ep_res)) & 15) == 0)
        g_fuse.am = ActMul{want_act ? f->ep_act_y : nullptr, f->ep_act, f->ep_slope, 0, 0, f->ep_res}, g_fuse.am_groups = groups,
        g_fuse.am_done = false;
    if (f->bn_y && f->bn_scale && f->bn_shift && f->bn_part && f->bn_cap > 0 && groups >= 1) {
        g_fuse.bn_y = f->bn_y, g_fuse.bn_scale = f->bn_scale, g_fuse.bn_shift = f->bn_shift, g_fuse.bn_slope = f->bn_slope;
        g_fuse.bn_part = f->bn_part, g_fuse.bn_cap = f->bn_cap, g_fuse.bn_groups = groups;
    }
}
inline void fuse_bn_collect(movae_fuse_t* f) {
    if (f) f->bn_ppg = g_fuse.bn_ppg, f->ep_act_done = g_fuse.am_done ? 1 : 0;
    g_fuse.bn_y = nullptr, g_fuse.bn_part = nullptr, g_fuse.bn_ppg = 0, g_fuse.bn_groups = 1;
    g_fuse.am.y = nullptr, g_fuse.am.res = nullptr, g_fuse.am_groups = 1, g_fuse.am_done = false;
}
#define MOVAE_CHECK_FUSE(f, c)                                                                                                   \
    MOVAE_CHECK_ARG(!(f) || !(f)->in_scale ||                                                                                    \
                        ((f)->in_shift && (c) % 4 == 0 && ((reinterpret_cast<uintptr_t>((f)->in_scale) | reinterpret_cast<uintptr_t>((f)->in_shift)) & 15) == 0), \
                    "fused input transform: needs scale AND shift, 16-byte aligned, channels %% 4 == 0")

extern "C" {

const char* movae_bench_last_kernel(void) { return g_last_kernel; }

int movae_reduce_defer(int on) {
    const int prev = g_defer.armed ? 1 : 0;
    g_defer.armed = on != 0;
    return prev;
}

int movae_reduce_flush(void) { return defer_flush(); }

long long movae_reduce_defer_max_bytes(long long bytes) {
    const long long prev = (long long)g_defer_max_bytes;
    if (bytes >= 0) g_defer_max_bytes = g_defer_max_bytes_pair = (size_t)bytes;  // (both carrier kinds)
    return prev;
}

int movae_reduce_defer_stats(long long* out3, int reset) {
    if (out3)
        for (int i = 0; i < 3; ++i) out3[i] = g_defer_stats[i];
    if (reset) g_defer_stats[0] = g_defer_stats[1] = g_defer_stats[2] = 0;
    return g_defer.pending ? 1 : 0;
}

int movae_bench_force_split(int s) {
    const int prev = g_force_split;
    g_force_split = s > 0 ? s : 0;
    return prev;
}

int movae_set_compute_dtype(int dtype) {
    const int prev = g_compute_bf16 ? MOVAE_DTYPE_BF16 : MOVAE_DTYPE_F32;
    g_compute_bf16 = dtype == MOVAE_DTYPE_BF16 ? 1 : 0;
    return prev;
}

int movae_bench_force_kgemm(int mode) {
    const int prev = g_force_kgemm;
    g_force_kgemm = mode > 0 ? 1 : (mode < 0 ? -1 : 0);
    return prev;
}

int movae_bench_kgemm_bn_fin(int mode) {  // 1 / 0: kgemm forwards do / do not finish the following BatchNorm in-launch; -1: the environment's choice
    const int prev = kg::g_kgemm_bn_fin;
    kg::g_kgemm_bn_fin = mode;
    return prev;
}

int movae_bench_main_kernel_only(int on) {
    const int prev = g_bench_main_only ? 1 : 0;
    g_bench_main_only = on != 0;
    return prev;
}

int movae_conv2d_fwd_f(const float* x, const float* w, const float* bias, float* y, int n, int hi, int wi, int ci, int ho,
                       int wo, int co, int kh, int kw, int stride, int pad, int act, float slope, void* ws, size_t ws_bytes,
                       movae_stream_t stream, movae_fuse_t* fuse) {
    unsigned* const ws_header = ws && ws_bytes > (size_t)MOVAE_WS_HEADER_BYTES ? static_cast<unsigned*>(ws) : nullptr;
    MOVAE_WS_SCRATCH(ws, ws_bytes);
    MOVAE_CHECK_ARG(x && w && y, "movae_conv2d_fwd: null pointer");
    if (int rc = check_conv_shape("movae_conv2d_fwd", n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad, false)) return rc;
    MOVAE_CHECK_FUSE(fuse, ci);
    FuseScope scope(fuse, 0);
    g_fuse.fin.counter = ws_header ? ws_header + 64 : nullptr;  // words 64 .. 127 of the header: one arrival counter per column tile
    Geom g{n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad};
    return launch_fwd(x, w, y, g, Epilogue{bias, act, slope}, ws, ws_bytes, (hipStream_t)stream);
}

int movae_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, int n, int hi, int wi, int ci, int ho,
                     int wo, int co, int kh, int kw, int stride, int pad, int act, float slope, void* ws, size_t ws_bytes,
                     movae_stream_t stream) {
    return movae_conv2d_fwd_f(x, w, bias, y, n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad, act, slope, ws, ws_bytes, stream, nullptr);
}

int movae_conv2d_dgrad(const float* dy, const float* w, float* dx, int n, int hi, int wi, int ci, int ho, int wo, int co,
                       int kh, int kw, int stride, int pad, void* ws, size_t ws_bytes, movae_stream_t stream) {
    MOVAE_WS_SCRATCH(ws, ws_bytes);
    MOVAE_CHECK_ARG(dy && w && dx, "movae_conv2d_dgrad: null pointer");
    if (int rc = check_conv_shape("movae_conv2d_dgrad", n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad, false)) return rc;
    // gathered tensor = dy (ho x wo x co), output grid = dx (hi x wi x ci); W[co][tap][ci] is the [Cr][tap][Nn] image
    Geom g{n, ho, wo, co, hi, wi, ci, kh, kw, stride, pad};
    return launch_bwd(dy, w, dx, g, Epilogue{nullptr, MOVAE_ACT_NONE, 0.f}, ws, ws_bytes, (hipStream_t)stream);
}

// n counts the images of ALL `groups` cotangents (dy / dx stacked); fuse->bn_* as in include/movae.h
int movae_conv2d_dgrad_f(const float* dy, const float* w, float* dx, int n, int hi, int wi, int ci, int ho, int wo, int co,
                         int kh, int kw, int stride, int pad, void* ws, size_t ws_bytes, movae_stream_t stream, movae_fuse_t* fuse,
                         int groups) {
    g_fuse = FuseCtx();
    fuse_bn_install(fuse, groups);
    const int rc = movae_conv2d_dgrad(dy, w, dx, n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad, ws, ws_bytes, stream);
    fuse_bn_collect(fuse);
    return rc;
}

int movae_conv2d_wgrad_grouped_f(int groups, const float* dy, const float* x, float* const* dw, float* const* dbias, int n, int hi,
                               int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int pad, int accumulate,
                               void* ws, size_t ws_bytes, movae_stream_t stream, const movae_fuse_t* fuse) {
    DeferArmScope defer_scope;  // (movae_reduce_defer arms ONE weight-gradient call)
    MOVAE_CHECK_FUSE(fuse, ci);
    FuseScope scope(const_cast<movae_fuse_t*>(fuse), 2);
    void* ws_full = ws;
    const size_t ws_full_bytes = ws_bytes;
    MOVAE_WS_SCRATCH(ws, ws_bytes);
    MOVAE_CHECK_ARG(dy && x && dw && groups >= 1 && groups <= 8, "movae_conv2d_wgrad: null pointer / bad group count");
    for (int i = 0; i < groups; ++i) MOVAE_CHECK_ARG(dw[i], "movae_conv2d_wgrad: null dw");
    if (int rc = check_conv_shape("movae_conv2d_wgrad", n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad, false)) return rc;
    WGeom g{n, ho, wo, co, hi, wi, ci, kh, kw, stride, pad};
    const long dy_gs = (long)n * ho * wo * co;  // dy is stacked [groups][n][ho][wo][co]; x is shared
    bool bias_done = false;
    if (int rc = launch_wgrad(dy, x, dw, groups, dy_gs, 0, g, accumulate, ws, ws_bytes, (hipStream_t)stream, dbias, &bias_done)) return rc;
    if (dbias && !bias_done && !g_bench_main_only && g_defer.pending && g_defer.serial != defer_scope.serial)
        if (int rc = defer_flush()) return rc;  // the column sums below use the arena the parked reduce's slabs lie in
    if (dbias && !bias_done && !g_bench_main_only)
        for (int i = 0; i < groups; ++i)
            if (dbias[i])
                if (int rc = movae_colsum(dy + i * dy_gs, dbias[i], n * ho * wo, co, accumulate, ws_full, ws_full_bytes, stream)) return rc;
    return MOVAE_OK;
}

int movae_conv2d_wgrad_grouped(int groups, const float* dy, const float* x, float* const* dw, float* const* dbias, int n, int hi,
                               int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int pad, int accumulate,
                               void* ws, size_t ws_bytes, movae_stream_t stream) {
    return movae_conv2d_wgrad_grouped_f(groups, dy, x, dw, dbias, n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad, accumulate, ws, ws_bytes, stream, nullptr);
}

int movae_conv2d_wgrad(const float* dy, const float* x, float* dw, float* dbias, int n, int hi, int wi, int ci, int ho,
                       int wo, int co, int kh, int kw, int stride, int pad, int accumulate, void* ws, size_t ws_bytes,
                       movae_stream_t stream) {
    float* dws[1] = {dw};
    float* dbs[1] = {dbias};
    return movae_conv2d_wgrad_grouped(1, dy, x, dws, dbias ? dbs : nullptr, n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad, accumulate,
                                      ws, ws_bytes, stream);
}

int movae_convT2d_fwd_f(const float* x, const float* w, const float* bias, float* y, int n, int hi, int wi, int ci, int ho,
                        int wo, int co, int kh, int kw, int stride, int pad, int act, float slope, void* ws, size_t ws_bytes,
                        movae_stream_t stream, movae_fuse_t* fuse) {
    unsigned* const ws_header = ws && ws_bytes > (size_t)MOVAE_WS_HEADER_BYTES ? static_cast<unsigned*>(ws) : nullptr;
    MOVAE_WS_SCRATCH(ws, ws_bytes);
    MOVAE_CHECK_ARG(x && w && y, "movae_convT2d_fwd: null pointer");
    if (int rc = check_conv_shape("movae_convT2d_fwd", n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad, true)) return rc;
    MOVAE_CHECK_FUSE(fuse, ci);
    FuseScope scope(fuse, 0);
    g_fuse.fin.counter = ws_header ? ws_header + 64 : nullptr;  // words 64 .. 127 of the header: one arrival counter per column tile
    Geom g{n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad};
    return launch_bwd(x, w, y, g, Epilogue{bias, act, slope}, ws, ws_bytes, (hipStream_t)stream);
}

int movae_convT2d_fwd(const float* x, const float* w, const float* bias, float* y, int n, int hi, int wi, int ci, int ho,
                      int wo, int co, int kh, int kw, int stride, int pad, int act, float slope, void* ws, size_t ws_bytes,
                      movae_stream_t stream) {
    return movae_convT2d_fwd_f(x, w, bias, y, n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad, act, slope, ws, ws_bytes, stream, nullptr);
}

int movae_convT2d_dgrad(const float* dy, const float* w, float* dx, int n, int hi, int wi, int ci, int ho, int wo, int co,
                        int kh, int kw, int stride, int pad, void* ws, size_t ws_bytes, movae_stream_t stream) {
    MOVAE_WS_SCRATCH(ws, ws_bytes);
    MOVAE_CHECK_ARG(dy && w && dx, "movae_convT2d_dgrad: null pointer");
    if (int rc = check_conv_shape("movae_convT2d_dgrad", n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad, true)) return rc;
    // dx[p][ci] = sum dy[p*s - pad + tap][co] * W[ci][tap][co]  : FWD form over dy with W as [Nn=ci][tap][Cr=co]
    Geom g{n, ho, wo, co, hi, wi, ci, kh, kw, stride, pad};
    return launch_fwd(dy, w, dx, g, Epilogue{nullptr, MOVAE_ACT_NONE, 0.f}, ws, ws_bytes, (hipStream_t)stream);
}

int movae_convT2d_dgrad_f(const float* dy, const float* w, float* dx, int n, int hi, int wi, int ci, int ho, int wo, int co,
                          int kh, int kw, int stride, int pad, void* ws, size_t ws_bytes, movae_stream_t stream, movae_fuse_t* fuse,
                          int groups) {
    g_fuse = FuseCtx();
    fuse_bn_install(fuse, groups);
    const int rc = movae_convT2d_dgrad(dy, w, dx, n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad, ws, ws_bytes, stream);
    fuse_bn_collect(fuse);
    return rc;
}

int movae_convT2d_wgrad_grouped_f(int groups, const float* dy, const float* x, float* const* dw, float* const* dbias, int n, int hi,
                                int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int pad, int accumulate,
                                void* ws, size_t ws_bytes, movae_stream_t stream, const movae_fuse_t* fuse) {
    DeferArmScope defer_scope;  // (movae_reduce_defer arms ONE weight-gradient call)
    MOVAE_CHECK_FUSE(fuse, ci);
    FuseScope scope(const_cast<movae_fuse_t*>(fuse), 1);
    void* ws_full = ws;
    const size_t ws_full_bytes = ws_bytes;
    MOVAE_WS_SCRATCH(ws, ws_bytes);
    MOVAE_CHECK_ARG(dy && x && dw && groups >= 1 && groups <= 8, "movae_convT2d_wgrad: null pointer / bad group count");
    for (int i = 0; i < groups; ++i) MOVAE_CHECK_ARG(dw[i], "movae_convT2d_wgrad: null dw");
    if (int rc = check_conv_shape("movae_convT2d_wgrad", n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad, true)) return rc;
    WGeom g{n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad};
    const long dy_gs = (long)n * ho * wo * co;  // the small side (x) is shared, the gathered side (dy) is per group
    if (int rc = launch_wgrad(x, dy, dw, groups, 0, dy_gs, g, accumulate, ws, ws_bytes, (hipStream_t)stream)) return rc;
    if (dbias && !g_bench_main_only && g_defer.pending && g_defer.serial != defer_scope.serial)
        if (int rc = defer_flush()) return rc;  // the column sums below use the arena the parked reduce's slabs lie in
    if (dbias && !g_bench_main_only)
        for (int i = 0; i < groups; ++i)
            if (dbias[i])
                if (int rc = movae_colsum(dy + i * dy_gs, dbias[i], n * ho * wo, co, accumulate, ws_full, ws_full_bytes, stream)) return rc;
    return MOVAE_OK;
}

int movae_convT2d_wgrad_grouped(int groups, const float* dy, const float* x, float* const* dw, float* const* dbias, int n, int hi,
                                int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int pad, int accumulate,
                                void* ws, size_t ws_bytes, movae_stream_t stream) {
    return movae_convT2d_wgrad_grouped_f(groups, dy, x, dw, dbias, n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad, accumulate, ws, ws_bytes, stream, nullptr);
}

int movae_convT2d_wgrad(const float* dy, const float* x, float* dw, float* dbias, int n, int hi, int wi, int ci, int ho,
                        int wo, int co, int kh, int kw, int stride, int pad, int accumulate, void* ws, size_t ws_bytes,
                        movae_stream_t stream) {
    float* dws[1] = {dw};
    float* dbs[1] = {dbias};
    return movae_convT2d_wgrad_grouped(1, dy, x, dws, dbias ? dbs : nullptr, n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad, accumulate,
                                       ws, ws_bytes, stream);
}

// dgrad (over groups * n images) and the grouped wgrad of one layer through ONE main launch when both land on the small-tile
// MFMA kernels (igemm2_pair); otherwise exactly the two calls above, in that order.
static int pair_calls(bool transposed, int groups, const float* dy, const float* w, const float* x, float* dx, float* const* dw,
                      float* const* dbias, int n, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw, int stride, int pad,
                      int accumulate, void* ws, size_t ws_bytes, movae_stream_t stream, const movae_fuse_t* fuse) {
    static const bool enabled = !getenv("MOVAE_NO_PAIR");
    v2::g_pending.active = false;
    kg::g_kpend.active = false;
    lin::g_lin_pend.active = false;
    thin::g_thin_pend.active = false;
    v2::g_pair_collect = enabled;
    g_fuse = FuseCtx();
    fuse_bn_install(const_cast<movae_fuse_t*>(fuse), groups);  // (the dgrad's plan -- also a stashed one -- keeps what it claimed)
    int rc = transposed ? movae_convT2d_dgrad(dy, w, dx, groups * n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad, ws, ws_bytes, stream)
                        : movae_conv2d_dgrad(dy, w, dx, groups * n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad, ws, ws_bytes, stream);
    fuse_bn_collect(const_cast<movae_fuse_t*>(fuse));
    v2::g_pair_collect = false;
    const char* dgrad_kernel = g_last_kernel;
    if (rc) {
        v2::g_pending.active = false;
        kg::g_kpend.active = false;
        lin::g_lin_pend.active = false;
        thin::g_thin_pend.active = false;
        return rc;
    }
    size_t used = v2::g_pending.active ? (v2::g_pending.ws_used + 255) / 256 * 256 : 0;
    if (v2::g_pending.active && (!ws || used + (32u << 20) > ws_bytes)) {  // no room left for the wgrad's slabs next to the dgrad's
        if ((rc = v2::flush_pending((hipStream_t)stream))) return rc;
        used = 0;
    }
    void* ws2 = ws ? static_cast<char*>(ws) + used : nullptr;
    // (`fuse` describes the activation operand x, which only the weight gradient reads)
    rc = transposed ? movae_convT2d_wgrad_grouped_f(groups, dy, x, dw, dbias, n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad, accumulate, ws2,
                                                    ws_bytes - used, stream, fuse)
                    : movae_conv2d_wgrad_grouped_f(groups, dy, x, dw, dbias, n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad, accumulate, ws2,
                                                   ws_bytes - used, stream, fuse);
    if (v2::g_pending.active || kg::g_kpend.active) {  // the wgrad took a kernel that does not pair (thin / linear / generic)
        const int rc2 = v2::flush_pending((hipStream_t)stream);
        if (!rc) rc = rc2;
    }
    if (lin::g_lin_pend.active) {  // a linear input gradient whose weight gradient took another kernel
        const int rc2 = lin::lin_flush((hipStream_t)stream);
        if (!rc) rc = rc2;
    }
    if (thin::g_thin_pend.active) {  // likewise a thin-channel input gradient
        const int rc2 = thin::thin_flush((hipStream_t)stream);
        if (!rc) rc = rc2;
    }
    if (strncmp(g_last_kernel, "igemm2_pair", 11) != 0 && strncmp(g_last_kernel, "kpair_k", 7) != 0 && strncmp(g_last_kernel, "linear_bwd_k", 12) != 0 && strncmp(g_last_kernel, "thin_pair_k", 11) != 0) {  // two main launches: movae_bench_last_kernel() names both
        static thread_local char both[128];
        snprintf(both, sizeof(both), "%s + %s", dgrad_kernel, g_last_kernel);
        g_last_kernel = both;
    }
    return rc;
}

int movae_conv2d_dgrad_wgrad_grouped_f(int groups, const float* dy, const float* w, const float* x, float* dx, float* const* dw,
                                       float* const* dbias, int n, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw,
                                       int stride, int pad, int accumulate, void* ws, size_t ws_bytes, movae_stream_t stream,
                                       const movae_fuse_t* fuse) {
    MOVAE_CHECK_ARG(dx && w, "movae_conv2d_dgrad_wgrad: null pointer");
    return pair_calls(false, groups, dy, w, x, dx, dw, dbias, n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad, accumulate, ws, ws_bytes,
                      stream, fuse);
}

int movae_conv2d_dgrad_wgrad_grouped(int groups, const float* dy, const float* w, const float* x, float* dx, float* const* dw,
                                     float* const* dbias, int n, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw,
                                     int stride, int pad, int accumulate, void* ws, size_t ws_bytes, movae_stream_t stream) {
    return movae_conv2d_dgrad_wgrad_grouped_f(groups, dy, w, x, dx, dw, dbias, n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad, accumulate,
                                              ws, ws_bytes, stream, nullptr);
}

int movae_convT2d_dgrad_wgrad_grouped_f(int groups, const float* dy, const float* w, const float* x, float* dx, float* const* dw,
                                        float* const* dbias, int n, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw,
                                        int stride, int pad, int accumulate, void* ws, size_t ws_bytes, movae_stream_t stream,
                                        const movae_fuse_t* fuse) {
    MOVAE_CHECK_ARG(dx && w, "movae_convT2d_dgrad_wgrad: null pointer");
    return pair_calls(true, groups, dy, w, x, dx, dw, dbias, n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad, accumulate, ws, ws_bytes,
                      stream, fuse);
}

int movae_convT2d_dgrad_wgrad_grouped(int groups, const float* dy, const float* w, const float* x, float* dx, float* const* dw,
                                      float* const* dbias, int n, int hi, int wi, int ci, int ho, int wo, int co, int kh, int kw,
                                      int stride, int pad, int accumulate, void* ws, size_t ws_bytes, movae_stream_t stream) {
    return movae_convT2d_dgrad_wgrad_grouped_f(groups, dy, w, x, dx, dw, dbias, n, hi, wi, ci, ho, wo, co, kh, kw, stride, pad, accumulate,
                                               ws, ws_bytes, stream, nullptr);
}

// Two fully connected layers on one input (fc_mu || fc_var) -- see lin::LinPair.  MOVAE_EUNSUPPORTED (nothing launched) for shapes
// outside the one-launch GEMM kernels: the caller then makes the two ordinary calls.
static bool linear_pair_ok(int groups, int m, int n, int k, const void* a, const void* b, const void* c, const void* d) {
    return groups >= 1 && groups <= 4 && m > 0 && n > 0 && k > 0 && n % 4 == 0 && k % 4 == 0 && lin::linear_small_ok(m, n, k) &&
           lin::linear_small_ok(m, k, n) && lin::linear_small_ok(n, k, (m + 3) / 4 * 4) &&
           ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) | reinterpret_cast<uintptr_t>(d)) & 15) == 0;
}

int movae_linear_pair_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* y1, float* y2,
                          int m, int n, int k, movae_stream_t stream) {
    MOVAE_CHECK_ARG(x && w1 && w2 && y1 && y2, "movae_linear_pair_fwd: null pointer");
    if (!linear_pair_ok(1, m, n, k, x, w1, w2, nullptr)) return MOVAE_EUNSUPPORTED;
    float* ys[2] = {y1, y2};
    g_last_kernel = "linear_small_k<NT pair>";
    return lin::launch_linear_pair<0>(x, nullptr, w1, w2, ys, nullptr, 1, 0, b1, b2, m, n, k, (hipStream_t)stream);
}

int movae_linear_pair_bwd(int groups, const float* dy1, const float* dy2, const float* w1, const float* w2, const float* x, float* dx,
                          float* const* dw1, float* const* dw2, float* const* db1, float* const* db2, int m, int n, int k,
                          movae_stream_t stream) {
    MOVAE_CHECK_ARG(dy1 && dy2 && w1 && w2 && x, "movae_linear_pair_bwd: null pointer");
    if (!linear_pair_ok(groups, m, n, k, dy1, dy2, w1, w2) || (reinterpret_cast<uintptr_t>(x) & 15) != 0) return MOVAE_EUNSUPPORTED;
    const long gs = (long)m * n;  // one cotangent group of dy
    if (dx && dw1 && dw2 && !g_bench_main_only) {  // both gradients: ONE launch (linear_bwd_k<true>)
        float* yd[8];
        float* yw[8];
        float* cs[8];
        for (int g = 0; g < groups; ++g) {
            yd[g] = dx + (long)g * m * k;
            yw[g] = dw1[g], yw[groups + g] = dw2[g];
            cs[g] = db1 ? db1[g] : nullptr, cs[groups + g] = db2 ? db2[g] : nullptr;
            MOVAE_CHECK_ARG(yw[g] && yw[groups + g], "movae_linear_pair_bwd: null weight-gradient destination");
        }
        const lin::LinProb dp = lin::make_prob<1>(dy1, dy2, w1, w2, yd, nullptr, groups, groups, gs, nullptr, nullptr, m, k, n, 0, 0.f, 0);
        const lin::LinProb wp = lin::make_prob<2>(dy1, dy2, x, nullptr, yw, cs, 2 * groups, groups, gs, nullptr, nullptr, n, k, m, 0, 0.f, 0);
        g_last_kernel = "linear_bwd_k<true>";
        return lin::launch_linear_bwd<true>(dp, wp, (hipStream_t)stream);
    }
    if (dx) {  // dx[g] = dy1[g] W1 + dy2[g] W2: [m][n] x [n][k]
        float* ys[8];
        for (int g = 0; g < groups; ++g) ys[g] = dx + (long)g * m * k;
        if (int rc = lin::launch_linear_pair<1>(dy1, dy2, w1, w2, ys, nullptr, groups, gs, nullptr, nullptr, m, k, n, (hipStream_t)stream))
            return rc;
    }
    if (dw1 && dw2) {  // dW[g] = dy[g]^T x: [n][k], reduction over the m rows; the bias gradients are the column sums of dy
        float* ys[8];
        float* cs[8];
        for (int g = 0; g < groups; ++g) {
            ys[g] = dw1[g], ys[groups + g] = dw2[g];
            cs[g] = db1 ? db1[g] : nullptr, cs[groups + g] = db2 ? db2[g] : nullptr;
            MOVAE_CHECK_ARG(ys[g] && ys[groups + g], "movae_linear_pair_bwd: null weight-gradient destination");
        }
        if (int rc = lin::launch_linear_pair<2>(dy1, dy2, x, nullptr, ys, cs, groups, gs, nullptr, nullptr, n, k, m, (hipStream_t)stream))
            return rc;
    }
    g_last_kernel = "linear_small_k<NN pair> + linear_small_k<TN pair>";
    return MOVAE_OK;
}

// Stand-alone statistics pass in the partial-sum format of the fused epilogues, for producers that cannot emit them
// (a thin-channel or generic kernel): one read of y.
int movae_bn_stats(const float* y, int rows, int c, float* stats, size_t stats_cap, int* parts_out, movae_stream_t stream) {
    MOVAE_CHECK_ARG(y && stats && parts_out && rows > 0 && c > 0, "movae_bn_stats: bad argument");
    g_fuse = FuseCtx();
    g_fuse.stats = stats, g_fuse.stats_cap = stats_cap;
    const bool ok = launch_reduce_stats(y, nullptr, rows, c, 1, nullptr, 0, 0, 0, nullptr, (hipStream_t)stream);
    *parts_out = g_fuse.stats_parts;
    g_fuse = FuseCtx();
    if (!ok) {
        movae_set_error("movae_bn_stats: unsupported shape rows=%d c=%d (needs c %% 4 == 0, 256 %% (c / 4) == 0, aligned operands, room for the partials)", rows, c);
        return MOVAE_EUNSUPPORTED;
    }
    MOVAE_CHECK_LAUNCH("bn_stats");
    return MOVAE_OK;
}

}  // extern "C"
