// linear_small.h -- one-launch GEMMs for the fully connected layers of the hot path (included inside conv_igemm.hip's
// anonymous namespace).
//
// fc_mu / fc_var / decoder_input are [batch x 512] x [512 x 128]-sized problems: 34 MFLOP, a few microseconds of MFMA
// time, for which the tiled implicit-GEMM kernels need split-K plus a reduce launch (and a column-sum launch pair for the
// bias gradient) -- 8-22 us per call, ~120 us of a 1.5 ms step.  Here a block owns ONE 32x32 output tile, its four waves
// split the reduction four ways, operands go global -> registers directly in MFMA lane order (no LDS staging, the whole
// k-slice of a wave is in flight at once), the four partial accumulators fold through LDS and the epilogue (bias,
// activation, bias gradient) happens in the same launch.
//   NT  (linear forward)  Y[m][n]  = sum_k A[m][k] * B[n][k]          A = x [M][K],  B = W [N][K]
//   NN  (linear dgrad)    Y[m][n]  = sum_k A[m][k] * B[k][n]          A = dy [M][K], B = W [K][N]
//   TN  (linear wgrad)    Y[m][n]  = sum_k A[k][m] * B[k][n]          A = dy [K][M], B = x [K][N];  colsum[m] = sum_k A[k][m]
// MFMA k assignment: within a group of 8 reduction indices lane half h consumes k = 4h + j at step j (both operands
// agree, so the contraction is unchanged).
#pragma once

namespace lin {

constexpr int LU = 16;  // 8-wide reduction groups a wave keeps in flight (16 * 8 = 128 reduction indices)

struct LinOut {
    float* y[8];       // per cotangent group (blockIdx.z)
    float* colsum[8];  // TN only: bias gradient per group (may be null)
};

// PAIR: two layers that read the same input in one launch (fc_mu || fc_var, models/vae.py:128-129,187-192: each is ~6 us of latency
// for ~0.2 us of arithmetic).  Second problem's operands in LinPair; the first one's in the ordinary arguments.
//   NT (forward):  blockIdx.z = layer;  A = x shared, B / bias / Y per layer
//   TN (wgrad):    blockIdx.z = layer * G + group;  A = dy of the layer and group, B = x shared
//   NN (dgrad):    blockIdx.z = group;  dx = dy1 W1 + dy2 W2 -- waves 0-1 reduce over layer 1, waves 2-3 over layer 2, the fold adds them
struct LinPair {
    const float* A2;
    const float* B2;
    const float* bias2;
    int G;  // TN: cotangent groups per layer
};

// one problem of the family, as a launch argument (the backward launches two: linear_bwd_k)
struct LinProb {
    const float* A;
    const float* B;
    LinOut out;
    const float* bias;
    int M, N, K, act;
    float slope;
    long a_gs;
    int accumulate;
    LinPair pr;
    int gx, gy, gz;  // the problem's grid
};

template <int FORM, bool PAIR>  // 0 = NT, 1 = NN, 2 = TN
__device__ __forceinline__ void linear_small_body(const float* __restrict__ A, const float* __restrict__ B, const LinOut& out,
                                                  const float* __restrict__ bias, int M, int N, int K, int act, float slope, long a_gs,
                                                  int accumulate, const LinPair& pr, int bx, int by, int bz, float (*red)[16][64],
                                                  float (*csum)[32]) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
    const int m0 = bx * 32, n0 = by * 32;
    int slot = wave, nslots = 4;  // this wave's share of the reduction
    if (PAIR) {
        if (FORM == 0) {
            if (bz) B = pr.B2, bias = pr.bias2;
        } else if (FORM == 2) {
            const int layer = bz / pr.G;
            A = (layer ? pr.A2 : A) + (bz - layer * pr.G) * a_gs;
        } else {
            const int layer = wave >> 1;
            A = (layer ? pr.A2 : A) + bz * a_gs;
            if (layer) B = pr.B2;
            slot = wave & 1, nslots = 2;
        }
    } else {
        A += bz * a_gs;  // cotangent group: only the A operand (dy) is per group
    }
    // reduction slice of this wave, in groups of 8
    const int groups = (K + 7) / 8, gper = (groups + nslots - 1) / nslots;
    const int g_begin = slot * gper, g_end = min(groups, g_begin + gper);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float cs = 0.f;
    const int am = m0 + r, bn = n0 + r;
    for (int gb = g_begin; gb < g_end; gb += LU) {
        f32x4 a4[LU], b4[LU];
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            const int k = (gb + u) * 8 + 4 * h;
            const bool kv = gb + u < g_end && k < K;  // K % 4 == 0
            if (FORM == 0) {
                a4[u] = (kv && am < M) ? *reinterpret_cast<const f32x4*>(A + (long)am * K + k) : f32x4{0.f, 0.f, 0.f, 0.f};
                b4[u] = (kv && bn < N) ? *reinterpret_cast<const f32x4*>(B + (long)bn * K + k) : f32x4{0.f, 0.f, 0.f, 0.f};
            } else if (FORM == 1) {
                a4[u] = (kv && am < M) ? *reinterpret_cast<const f32x4*>(A + (long)am * K + k) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) b4[u][j] = (kv && bn < N) ? B[(long)(k + j) * N + bn] : 0.f;
            } else {  // the reduction index is the batch row here: any count, checked per row
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool kj = gb + u < g_end && k + j < K;
                    a4[u][j] = (kj && am < M) ? A[(long)(k + j) * M + am] : 0.f;
                    b4[u][j] = (kj && bn < N) ? B[(long)(k + j) * N + bn] : 0.f;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < LU; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u][j], b4[u][j], acc, 0, 0, 0);
                if (FORM == 2) cs += a4[u][j];
            }
    }
    // fold the four waves' partial tiles; thread t then owns 4 consecutive outputs of one row
#pragma unroll
    for (int i = 0; i < 16; ++i) red[wave][i][lane] = acc[i];
    if (FORM == 2) {
        cs += __shfl_xor(cs, 32, 64);
        if (lane < 32) csum[wave][lane] = cs;
    }
    __syncthreads();
    // accumulator register i of lane (r, h) is C[row = (i & 3) + 8 * (i >> 2) + 4 * h][col = r]
    {
        const int row = t >> 3, c4 = (t & 7) * 4;  // 32 rows x 8 column quads
        const int i = (row & 3) + 4 * (row >> 3), hh = (row >> 2) & 1;
        const int m = m0 + row;
        float* Y = out.y[bz];
        if (m < M) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + c4 + j;
                if (n < N) {
                    const int ln = hh * 32 + c4 + j;
                    float v = (red[0][i][ln] + red[1][i][ln]) + (red[2][i][ln] + red[3][i][ln]);
                    if (FORM != 2) v = apply_act(v + (bias ? bias[n] : 0.f), act, slope);
                    float* dst = Y + (long)m * N + n;
                    *dst = accumulate ? *dst + v : v;
                }
            }
        }
    }
    if (FORM == 2 && by == 0 && t < 32 && m0 + t < M) {
        float* dbias = out.colsum[bz];
        if (dbias) {
            const float v = (csum[0][t] + csum[1][t]) + (csum[2][t] + csum[3][t]);
            dbias[m0 + t] = accumulate ? dbias[m0 + t] + v : v;
        }
    }
}

template <int FORM, bool PAIR = false>
__global__ __launch_bounds__(256) void linear_small_k(const float* __restrict__ A, const float* __restrict__ B, LinOut out,
                                                      const float* __restrict__ bias, int M, int N, int K, int act, float slope,
                                                      long a_gs, int accumulate, LinPair pr = LinPair{nullptr, nullptr, nullptr, 1}) {
    __shared__ float red[4][16][64];
    __shared__ float csum[4][32];
    linear_small_body<FORM, PAIR>(A, B, out, bias, M, N, K, act, slope, a_gs, accumulate, pr, blockIdx.x, blockIdx.y, blockIdx.z, red, csum);
}

// A linear layer's backward in ONE launch: blocks [0, nd) the input gradient (NN), the rest the weight gradient (TN) -- the two
// only share read-only operands, and each alone is a few blocks of latency (the dgrad || wgrad pairing of igemm2_pair at the
// fully connected layers' scale).
template <bool PAIR>
__global__ __launch_bounds__(256) void linear_bwd_k(LinProb d, LinProb w, int nd) {
    __shared__ float red[4][16][64];
    __shared__ float csum[4][32];
    int b = blockIdx.x;
    if (b < nd) {
        const int bx = b % d.gx, r = b / d.gx;
        linear_small_body<1, PAIR>(d.A, d.B, d.out, d.bias, d.M, d.N, d.K, d.act, d.slope, d.a_gs, d.accumulate, d.pr, bx, r % d.gy, r / d.gy, red, csum);
    } else {
        b -= nd;
        const int bx = b % w.gx, r = b / w.gx;
        linear_small_body<2, PAIR>(w.A, w.B, w.out, w.bias, w.M, w.N, w.K, w.act, w.slope, w.a_gs, w.accumulate, w.pr, bx, r % w.gy, r / w.gy, red, csum);
    }
}

// shapes served: 1x1 spatial, reduction short enough for one in-flight slice chain, few enough tiles to be latency bound
inline bool linear_small_ok(long M, long N, long K) {
    static const bool off = getenv("MOVAE_NO_LINEAR_SMALL") != nullptr;
    return !off && K % 4 == 0 && K <= 2048 && M * N <= 1024L * 1024 && M > 0 && N > 0;
}

template <int FORM>
int launch_linear_small(const float* A, const float* B, float* const* Y, float* const* colsum, int G, long a_gs, const float* bias,
                        int M, int N, int K, int act, float slope, int accumulate, hipStream_t st) {
    LinOut out;
    for (int i = 0; i < 8; ++i) {
        out.y[i] = i < G ? Y[i] : nullptr;
        out.colsum[i] = (i < G && colsum) ? colsum[i] : nullptr;
    }
    hipLaunchKernelGGL((linear_small_k<FORM>), dim3(ceil_div(M, 32), ceil_div(N, 32), G), dim3(256), 0, st, A, B, out, bias, M, N, K, act,
                       slope, a_gs, accumulate, LinPair{nullptr, nullptr, nullptr, 1});
    MOVAE_CHECK_LAUNCH("linear_small");
    return MOVAE_OK;
}

// the pair forms (see LinPair).  Y / colsum: NT two entries (layer), TN 2 * G entries (layer-major), NN G entries (group)
template <int FORM>
int launch_linear_pair(const float* A, const float* A2, const float* B, const float* B2, float* const* Y, float* const* colsum, int G,
                       long a_gs, const float* bias, const float* bias2, int M, int N, int K, hipStream_t st) {
    const int nz = FORM == 0 ? 2 : (FORM == 2 ? 2 * G : G);
    LinOut out;
    for (int i = 0; i < 8; ++i) {
        out.y[i] = i < nz ? Y[i] : nullptr;
        out.colsum[i] = (i < nz && colsum) ? colsum[i] : nullptr;
    }
    hipLaunchKernelGGL((linear_small_k<FORM, true>), dim3(ceil_div(M, 32), ceil_div(N, 32), nz), dim3(256), 0, st, A, B, out, bias, M, N, K,
                       0, 0.f, a_gs, 0, LinPair{A2, B2, bias2, G});
    MOVAE_CHECK_LAUNCH("linear_pair");
    return MOVAE_OK;
}

template <int FORM>
inline LinProb make_prob(const float* A, const float* A2, const float* B, const float* B2, float* const* Y, float* const* colsum, int nz,
                         int G, long a_gs, const float* bias, const float* bias2, int M, int N, int K, int act, float slope, int accumulate) {
    LinProb p{};
    p.A = A, p.B = B, p.bias = bias, p.M = M, p.N = N, p.K = K, p.act = act, p.slope = slope, p.a_gs = a_gs, p.accumulate = accumulate;
    p.pr = LinPair{A2, B2, bias2, G};
    for (int i = 0; i < 8; ++i) {
        p.out.y[i] = i < nz ? Y[i] : nullptr;
        p.out.colsum[i] = (i < nz && colsum) ? colsum[i] : nullptr;
    }
    p.gx = ceil_div(M, 32), p.gy = ceil_div(N, 32), p.gz = nz;
    return p;
}

template <bool PAIR>
inline int launch_linear_bwd(const LinProb& d, const LinProb& w, hipStream_t st) {
    const int nd = d.gx * d.gy * d.gz, nw = w.gx * w.gy * w.gz;
    hipLaunchKernelGGL((linear_bwd_k<PAIR>), dim3(nd + nw), dim3(256), 0, st, d, w, nd);
    MOVAE_CHECK_LAUNCH("linear_bwd");
    return MOVAE_OK;
}

// a linear layer's input gradient planned inside a dgrad + wgrad call (v2::g_pair_collect), waiting for its weight gradient
struct LinPending {
    bool active = false;
    LinProb d;
};
static thread_local LinPending g_lin_pend;
inline int lin_flush(hipStream_t st) {  // the weight gradient took another kernel: the input gradient goes alone
    if (!g_lin_pend.active) return MOVAE_OK;
    g_lin_pend.active = false;
    const LinProb& d = g_lin_pend.d;
    hipLaunchKernelGGL((linear_small_k<1>), dim3(d.gx, d.gy, d.gz), dim3(256), 0, st, d.A, d.B, d.out, d.bias, d.M, d.N, d.K, d.act, d.slope,
                       d.a_gs, d.accumulate, d.pr);
    MOVAE_CHECK_LAUNCH("linear_small (unpaired input gradient)");
    return MOVAE_OK;
}

}  // namespace lin
