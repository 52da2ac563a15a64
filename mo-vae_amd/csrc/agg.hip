// K-loss gradient aggregation on device: Gramian of the K x m Jacobian, the K x K weight solves
// (UPGrad dual-cone projection, MGDA Frank-Wolfe, Aligned-MTL eigen balance) and the combine.
// Gram / combine / similarity are pure HBM streams over J (K <= 8 rows, so no MFMA: the
// arithmetic intensity is K/4 flop per byte); the solves are single-wave kernels in fp64 so the
// reference's host round trip (GPU -> numpy float64 -> quadprog -> GPU) never happens.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int MAXK = MOVAE_MAX_K;
// blocks of the streaming passes over J (gram_partial, combine, similarity): one per 2048 columns, at most MOVAE_GRAM_BLOCKS
inline int gram_blocks(size_t m) {
    static const size_t cap = getenv("MOVAE_GRAM_BLOCKS") ? (size_t)atol(getenv("MOVAE_GRAM_BLOCKS")) : 1024;
    size_t g = (m + 2047) / 2048;
    return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

template <int K, bool VEC>
__global__ __launch_bounds__(256) void gram_partial(const float* __restrict__ J, long ldj, long m, double* __restrict__ part) {
    constexpr int NP = K * (K + 1) / 2;
    double acc[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) acc[q] = 0.0;
    const long stride = (long)gridDim.x * blockDim.x;
    if (VEC) {
        const long nv = m / 4;
        for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < nv; c += stride) {
            f32x4 v[K];
#pragma unroll
            for (int i = 0; i < K; ++i) v[i] = reinterpret_cast<const f32x4*>(J + i * ldj)[c];
            int q = 0;
#pragma unroll
            for (int i = 0; i < K; ++i)
#pragma unroll
                for (int j = i; j < K; ++j) {
                    const f32x4 p = v[i] * v[j];
                    acc[q] += (double)p[0] + (double)p[1] + (double)p[2] + (double)p[3];
                    ++q;
                }
        }
        // tail (m % 4) handled by thread 0 of block 0
        if (blockIdx.x == 0 && threadIdx.x == 0)
            for (long c = nv * 4; c < m; ++c) {
                int q = 0;
                for (int i = 0; i < K; ++i)
                    for (int j = i; j < K; ++j) acc[q++] += (double)(J[i * ldj + c] * J[j * ldj + c]);
            }
    } else {
        for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < m; c += stride) {
            float v[K];
#pragma unroll
            for (int i = 0; i < K; ++i) v[i] = J[i * ldj + c];
            int q = 0;
#pragma unroll
            for (int i = 0; i < K; ++i)
#pragma unroll
                for (int j = i; j < K; ++j) acc[q++] += (double)(v[i] * v[j]);
        }
    }
    __shared__ double sh[4];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const double s = block_sum_256(acc[q], sh);
        if (threadIdx.x == 0) part[(long)blockIdx.x * NP + q] = s;
    }
}

// one wave per Gramian entry (upper triangle), lanes stride over the block partials
__global__ __launch_bounds__(64) void gram_final(const double* __restrict__ part, int nblk, int K, float* __restrict__ G) {
    const int NP = K * (K + 1) / 2;
    const int q = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) s += part[(long)b * NP + q];
    s = wave_sum(s);
    if (threadIdx.x != 0) return;
    int i = 0, rem = q;
    while (rem >= K - i) {
        rem -= K - i;
        ++i;
    }
    const int j = i + rem;
    G[i * K + j] = (float)s;
    G[j * K + i] = (float)s;
}

template <int K, bool VEC>
__global__ __launch_bounds__(256) void combine_k(const float* __restrict__ J, long ldj, long m, const float* __restrict__ w,
                                                 float* __restrict__ g, int accumulate) {
    float wv[K];
#pragma unroll
    for (int i = 0; i < K; ++i) wv[i] = w[i];
    const long stride = (long)gridDim.x * blockDim.x;
    if (VEC) {
        const long nv = m / 4;
        for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < nv; c += stride) {
            f32x4 s = accumulate ? reinterpret_cast<const f32x4*>(g)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < K; ++i) s += wv[i] * reinterpret_cast<const f32x4*>(J + i * ldj)[c];
            reinterpret_cast<f32x4*>(g)[c] = s;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0)
            for (long c = nv * 4; c < m; ++c) {
                float s = accumulate ? g[c] : 0.f;
                for (int i = 0; i < K; ++i) s += wv[i] * J[i * ldj + c];
                g[c] = s;
            }
    } else {
        for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < m; c += stride) {
            float s = accumulate ? g[c] : 0.f;
#pragma unroll
            for (int i = 0; i < K; ++i) s += wv[i] * J[i * ldj + c];
            g[c] = s;
        }
    }
}

__global__ __launch_bounds__(256) void similarity_partial(const float* __restrict__ J, long ldj, int K, long m,
                                                          const float* __restrict__ w, double* __restrict__ part) {
    __shared__ double sh[4];
    double ab = 0.0, aa = 0.0, bb = 0.0;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < m; c += stride) {
        float a = 0.f, b = 0.f;
        for (int i = 0; i < K; ++i) {
            const float v = J[i * ldj + c];
            a += w[i] * v;
            b += v;
        }
        b /= (float)K;
        ab += (double)a * b;
        aa += (double)a * a;
        bb += (double)b * b;
    }
    ab = block_sum_256(ab, sh);
    aa = block_sum_256(aa, sh);
    bb = block_sum_256(bb, sh);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 3 + 0] = ab;
        part[blockIdx.x * 3 + 1] = aa;
        part[blockIdx.x * 3 + 2] = bb;
    }
}

__global__ void similarity_final(const double* __restrict__ part, int nblk, float* __restrict__ out) {
    if (threadIdx.x != 0) return;
    double ab = 0, aa = 0, bb = 0;
    for (int b = 0; b < nblk; ++b) {
        ab += part[b * 3];
        aa += part[b * 3 + 1];
        bb += part[b * 3 + 2];
    }
    const double na = fmax(sqrt(aa), 1e-8), nb = fmax(sqrt(bb), 1e-8);  // F.cosine_similarity eps
    out[0] = (float)(ab / (na * nb));
}

// ---- UPGrad --------------------------------------------------------------------------------------
// One wave; lane s evaluates the active-set candidate whose free set is the bit mask s (s += 64
// until 2^K), solves the free block by Cholesky in fp64 and scores the KKT violation; the wave
// keeps the least-violating candidate (the unique KKT point of the strictly convex QP).
// norm_mode: how the Gramian is normalised before the projection --
//   0 trace (torchjd UPGrad), 1 min-L2-norm scaling (NUPGrad, utils/torchmoo/nupgrad.py:122-158),
//   2 cosine (PNUPGrad's `normalize`, utils/torchmoo/pnupgrad.py:13-24)
// dual != 0: torchjd DualProj -- ONE projection, of the whole preference vector u (default: the mean weights 1/K), instead of
// one per row: w = u + argmin_{v >= 0} 1/2 v'Gv + (Gu)'v
// KT = K at compile time (every loop unrolls, every array index is a constant, so the fp64 work lives in registers; with a
// run-time K the 8x8 arrays sit in scratch memory and the K = 2 solve took 12 us instead of ~2).  An active set is handled
// without compaction: rows / columns outside the free set are replaced by the identity and their right-hand side by 0, which
// leaves the free block's Cholesky arithmetic untouched and yields v = 0 on the active coordinates.
template <int KT>
__global__ __launch_bounds__(64) void upgrad_k(const float* __restrict__ Gptr, int norm_mode, float norm_eps, float reg_eps,
                                               const float* __restrict__ pref, float* __restrict__ wout, int dual,
                                               const double* __restrict__ gpart, int gnb, float* __restrict__ Gout) {
    constexpr int K = KT;
    const int lane = threadIdx.x;
    // gpart: the Gramian has not been formed yet -- gram_partial's block partials are folded HERE, entry by entry, in gram_final's
    // order (lanes stride over the blocks, one wave fold), so the values are gram_final's to the bit; Gout receives them
    float Gin[K * K];
    if (gpart) {
        constexpr int NP = K * (K + 1) / 2;
        int q = 0;
#pragma unroll
        for (int i = 0; i < K; ++i)
#pragma unroll
            for (int j = i; j < K; ++j, ++q) {
                double s = 0.0;
                for (int b = lane; b < gnb; b += 64) s += gpart[(long)b * NP + q];
                s = wave_sum(s);
                Gin[i * K + j] = Gin[j * K + i] = (float)s;
            }
        if (lane == 0 && Gout)
#pragma unroll
            for (int i = 0; i < K * K; ++i) Gout[i] = Gin[i];
    } else {
#pragma unroll
        for (int i = 0; i < K * K; ++i) Gin[i] = Gptr[i];
    }
    double G[K][K];
    if (norm_mode == 0) {
        double tr = 0.0;
#pragma unroll
        for (int i = 0; i < K; ++i) tr += (double)Gin[i * K + i];
        const bool zero = tr < (double)norm_eps;
#pragma unroll
        for (int i = 0; i < K; ++i)
#pragma unroll
            for (int j = 0; j < K; ++j) G[i][j] = zero ? 0.0 : (double)Gin[i * K + j] / tr;
    } else {
        // the reference normalises with float32 tensor ops and only then hands the matrix to the fp64 QP; the same
        // roundings are kept here because the QP of a rank-deficient Gramian amplifies them by ~1 / reg_eps
        float l2[K], sf[K];
        float amin = 0.f;
        bool any = false;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            l2[i] = sqrtf(fmaxf(Gin[i * K + i], norm_eps));
            if (l2[i] > norm_eps && (!any || l2[i] < amin)) {
                amin = l2[i];
                any = true;
            }
        }
#pragma unroll
        for (int i = 0; i < K; ++i) sf[i] = (any && l2[i] > norm_eps) ? amin / l2[i] : 0.f;
#pragma unroll
        for (int i = 0; i < K; ++i)
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const float g = Gin[i * K + j];
                G[i][j] = norm_mode == 1 ? (double)(g * (sf[i] * sf[j])) : (double)(g / (l2[i] * l2[j]));
            }
    }
#pragma unroll
    for (int i = 0; i < K; ++i) G[i][i] += (double)reg_eps;
    double u[K], wsum[K];
#pragma unroll
    for (int i = 0; i < K; ++i) {
        u[i] = pref ? (double)pref[i] : 1.0 / K;
        wsum[i] = 0.0;
    }
    constexpr int nsub = 1 << K;
    for (int row = 0; row < (dual ? 1 : K); ++row) {
        double c[K];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < K; ++j) acc += G[i][j] * ((dual || j == row) ? u[j] : 0.0);
            c[i] = acc;  // one projection per row of diag(u): c = G[:, row] * u_row; DualProj: c = G u
        }
        double best_viol = 1e300;
        int best_mask = 0;
        double best_v[K];
#pragma unroll
        for (int i = 0; i < K; ++i) best_v[i] = 0.0;
        for (int mask = lane; mask < nsub; mask += 64) {
            double L[K][K], y[K], v[K];
            bool ok = true;
#pragma unroll
            for (int a = 0; a < K; ++a) {
                const bool fa = mask >> a & 1;
#pragma unroll
                for (int b = 0; b <= a; ++b) {
                    const bool fb = mask >> b & 1;
                    double s = (fa && fb) ? G[a][b] : (a == b ? 1.0 : 0.0);
#pragma unroll
                    for (int q = 0; q < b; ++q) s -= L[a][q] * L[b][q];
                    if (a == b) {
                        if (s <= 0.0) {
                            ok = false;
                            s = 1.0;
                        }
                        L[a][a] = sqrt(s);
                    } else {
                        L[a][b] = s / L[b][b];
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < K; ++a) {
                double s = (mask >> a & 1) ? -c[a] : 0.0;
#pragma unroll
                for (int q = 0; q < a; ++q) s -= L[a][q] * y[q];
                y[a] = s / L[a][a];
            }
#pragma unroll
            for (int a = K - 1; a >= 0; --a) {
                double s = y[a];
#pragma unroll
                for (int q = a + 1; q < K; ++q) s -= L[q][a] * v[q];
                v[a] = s / L[a][a];
            }
            double viol = ok ? 0.0 : 1e200;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                if (mask >> i & 1) {
                    viol = fmax(viol, -v[i]);
                } else {
                    v[i] = 0.0;  // exactly zero on the active set (the identity rows give +-0)
                    double gr = c[i];
#pragma unroll
                    for (int j = 0; j < K; ++j) gr += G[i][j] * ((mask >> j & 1) ? v[j] : 0.0);
                    viol = fmax(viol, -gr);
                }
            }
            if (viol < best_viol) {
                best_viol = viol;
                best_mask = mask;
#pragma unroll
                for (int i = 0; i < K; ++i) best_v[i] = v[i];
            }
        }
        // wave arg-min on (violation, mask)
        for (int o = 32; o > 0; o >>= 1) {
            const double ov = __shfl_xor(best_viol, o, 64);
            const int om = __shfl_xor(best_mask, o, 64);
            double tmp[K];
#pragma unroll
            for (int i = 0; i < K; ++i) tmp[i] = __shfl_xor(best_v[i], o, 64);
            if (ov < best_viol || (ov == best_viol && om < best_mask)) {
                best_viol = ov;
                best_mask = om;
#pragma unroll
                for (int i = 0; i < K; ++i) best_v[i] = tmp[i];
            }
        }
#pragma unroll
        for (int i = 0; i < K; ++i) wsum[i] += best_v[i] + ((dual || i == row) ? u[i] : 0.0);
    }
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < K; ++i) wout[i] = (float)wsum[i];
}

// ---- torchjd CAGrad (Liu et al. 2021) ---------------------------------------------------------------------------------
// u = 1/K, b = G u, kappa = c sqrt(u'Gu):  w* = argmin_{w in simplex} b'w + kappa sqrt(w'Gw),  weights = u + kappa / sqrt(w*'Gw*) w*
// (the mean weights when kappa or sqrt(w*'Gw*) is <= norm_eps).  On a support S the KKT system is closed-form:
//   G_S w = (t / kappa)(lambda 1 - b_S),  p = G_S^-1 1, q = G_S^-1 b_S,  A = 1'p, B = 1'q, C = b_S'q,
//   kappa^2 = A lambda^2 - 2 B lambda + C  (larger root),  w = (lambda p - q) / (lambda A - B)
// one lane per support, masked-identity Cholesky in fp64 as in upgrad_k, wave arg-min on the KKT violation.  torchjd solves the
// same cone problem with cvxpy / CLARABEL; Gramians with a null direction inside the simplex (singular supports are skipped
// here) are where the two can differ.  Raw Gramian: CAGrad neither normalises nor regularises.
template <int KT>
__global__ __launch_bounds__(64) void cagrad_k(const float* __restrict__ Gin, float c, float norm_eps, float* __restrict__ wout) {
    constexpr int K = KT;
    const int lane = threadIdx.x;
    double G[K][K], b[K];
    double tr = 0.0, uGu = 0.0;
#pragma unroll
    for (int i = 0; i < K; ++i) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            G[i][j] = 0.5 * ((double)Gin[i * K + j] + (double)Gin[j * K + i]);
            uGu += G[i][j];
        }
        tr += G[i][i];
    }
    uGu /= (double)K * K;
    const double kappa = (double)c * sqrt(fmax(uGu, 0.0)), jitter = 1e-14 * tr;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < K; ++j) acc += G[i][j];
        b[i] = acc / K;
    }
    double best_viol = 1e300;
    int best_mask = 0;
    double best_w[K];
#pragma unroll
    for (int i = 0; i < K; ++i) best_w[i] = 0.0;
    for (int mask = lane + 1; mask < (1 << K); mask += 64) {
        double L[K][K], yp[K], yq[K], p[K], q[K], w[K];
        bool ok = true;
#pragma unroll
        for (int a = 0; a < K; ++a) {
            const bool fa = mask >> a & 1;
#pragma unroll
            for (int bb = 0; bb <= a; ++bb) {
                const bool fb = mask >> bb & 1;
                double s = (fa && fb) ? G[a][bb] + (a == bb ? jitter : 0.0) : (a == bb ? 1.0 : 0.0);
#pragma unroll
                for (int r = 0; r < bb; ++r) s -= L[a][r] * L[bb][r];
                if (a == bb) {
                    if (s <= 0.0) {
                        ok = false;
                        s = 1.0;
                    }
                    L[a][a] = sqrt(s);
                } else {
                    L[a][bb] = s / L[bb][bb];
                }
            }
        }
#pragma unroll
        for (int a = 0; a < K; ++a) {
            const bool fa = mask >> a & 1;
            double sp = fa ? 1.0 : 0.0, sq = fa ? b[a] : 0.0;
#pragma unroll
            for (int r = 0; r < a; ++r) {
                sp -= L[a][r] * yp[r];
                sq -= L[a][r] * yq[r];
            }
            yp[a] = sp / L[a][a];
            yq[a] = sq / L[a][a];
        }
#pragma unroll
        for (int a = K - 1; a >= 0; --a) {
            double sp = yp[a], sq = yq[a];
#pragma unroll
            for (int r = a + 1; r < K; ++r) {
                sp -= L[r][a] * p[r];
                sq -= L[r][a] * q[r];
            }
            p[a] = sp / L[a][a];
            q[a] = sq / L[a][a];
        }
        double A = 0.0, B = 0.0, C = 0.0;
#pragma unroll
        for (int i = 0; i < K; ++i)
            if (mask >> i & 1) {
                A += p[i];
                B += q[i];
                C += b[i] * q[i];
            }
        const double disc = B * B - A * (C - kappa * kappa);
        ok = ok && disc >= 0.0 && A > 0.0;
        const double lam = ok ? (B + sqrt(disc)) / A : 0.0, den = lam * A - B;
        ok = ok && den > 0.0;
#pragma unroll
        for (int i = 0; i < K; ++i) w[i] = (ok && (mask >> i & 1)) ? (lam * p[i] - q[i]) / den : 0.0;
        double Gw[K], wGw = 0.0;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < K; ++j) acc += G[i][j] * w[j];
            Gw[i] = acc;
            wGw += w[i] * acc;
        }
        const double t = sqrt(fmax(wGw, 0.0));
        ok = ok && t > 0.0;
        double viol = ok ? 0.0 : 1e200;
        if (ok) {
#pragma unroll
            for (int i = 0; i < K; ++i) {
                if (mask >> i & 1) viol = fmax(viol, -w[i]);
                else viol = fmax(viol, -(b[i] + kappa * Gw[i] / t - lam));
            }
        }
        if (viol < best_viol) {
            best_viol = viol;
            best_mask = mask;
#pragma unroll
            for (int i = 0; i < K; ++i) best_w[i] = w[i];
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(best_viol, o, 64);
        const int om = __shfl_xor(best_mask, o, 64);
        double tmp[K];
#pragma unroll
        for (int i = 0; i < K; ++i) tmp[i] = __shfl_xor(best_w[i], o, 64);
        if (ov < best_viol || (ov == best_viol && om < best_mask)) {
            best_viol = ov;
            best_mask = om;
#pragma unroll
            for (int i = 0; i < K; ++i) best_w[i] = tmp[i];
        }
    }
    if (lane == 0) {
        double wGw = 0.0;
#pragma unroll
        for (int i = 0; i < K; ++i)
#pragma unroll
            for (int j = 0; j < K; ++j) wGw += best_w[i] * G[i][j] * best_w[j];
        const double gw = sqrt(fmax(wGw, 0.0));
        const bool mean = kappa <= (double)norm_eps || best_viol >= 1e199 || gw <= (double)norm_eps;
#pragma unroll
        for (int i = 0; i < K; ++i) wout[i] = (float)(mean ? 1.0 / K : 1.0 / K + kappa / gw * best_w[i]);
    }
}

inline void launch_cagrad(const float* G, int k, float c, float norm_eps, float* w, hipStream_t st) {
#define MOVAE_CAG(KV) hipLaunchKernelGGL((cagrad_k<KV>), dim3(1), dim3(64), 0, st, G, c, norm_eps, w)
    switch (k) {
        case 1: MOVAE_CAG(1); break;
        case 2: MOVAE_CAG(2); break;
        case 3: MOVAE_CAG(3); break;
        case 4: MOVAE_CAG(4); break;
        case 5: MOVAE_CAG(5); break;
        case 6: MOVAE_CAG(6); break;
        case 7: MOVAE_CAG(7); break;
        default: MOVAE_CAG(8); break;
    }
#undef MOVAE_CAG
}

inline void launch_upgrad(const float* G, int k, int norm_mode, float norm_eps, float reg_eps, const float* pref, float* w, int dual,
                          hipStream_t st, const double* gpart = nullptr, int gnb = 0, float* Gout = nullptr) {
#define MOVAE_UPG(KV) \
    hipLaunchKernelGGL((upgrad_k<KV>), dim3(1), dim3(64), 0, st, G, norm_mode, norm_eps, reg_eps, pref, w, dual, gpart, gnb, Gout)
    switch (k) {
        case 1: MOVAE_UPG(1); break;
        case 2: MOVAE_UPG(2); break;
        case 3: MOVAE_UPG(3); break;
        case 4: MOVAE_UPG(4); break;
        case 5: MOVAE_UPG(5); break;
        case 6: MOVAE_UPG(6); break;
        case 7: MOVAE_UPG(7); break;
        default: MOVAE_UPG(8); break;
    }
#undef MOVAE_UPG
}

// ---- MGDA Frank-Wolfe (fp32, op order of utils/torchmoo/mgda.py:241-265) ------------------------------
// cyclic Jacobi eigen-decomposition of a symmetric K x K matrix in fp64: on return A is (numerically) diagonal with the
// eigenvalues and the columns of V are the eigenvectors (V must enter as the identity)
__device__ void jacobi_eigh(double (&A)[MAXK][MAXK], double (&V)[MAXK][MAXK], int K) {
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < K; ++i)
            for (int j = 0; j < K; ++j) (i == j ? diag : off) += A[i][j] * A[i][j];
        if (off <= 1e-30 * (diag + 1e-300)) break;
        for (int p = 0; p < K - 1; ++p)
            for (int q = p + 1; q < K; ++q) {
                if (A[p][q] == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(tt * tt + 1.0), sn = tt * cs;
                for (int k = 0; k < K; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = cs * akp - sn * akq;
                    A[k][q] = sn * akp + cs * akq;
                }
                for (int k = 0; k < K; ++k) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = cs * apk - sn * aqk;
                    A[q][k] = sn * apk + cs * aqk;
                }
                for (int k = 0; k < K; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = cs * vkp - sn * vkq;
                    V[k][q] = sn * vkp + cs * vkq;
                }
            }
    }
}

__global__ void mgda_k(const float* __restrict__ Gin, int K, int norm, const float* __restrict__ losses, float epsilon,
                       int max_iters, float* __restrict__ wout, int* __restrict__ info, int stable, float min_eig) {
    if (threadIdx.x != 0) return;
    float G[MAXK][MAXK], nrm[MAXK], ls[MAXK];
    for (int i = 0; i < K; ++i) {
        nrm[i] = sqrtf(fmaxf(Gin[i * K + i], 1e-20f));
        ls[i] = losses ? fmaxf(losses[i], 1e-20f) : 1.f;
    }
    for (int i = 0; i < K; ++i)
        for (int j = 0; j < K; ++j) {
            float d = 1.f;
            if (norm == MOVAE_MGDA_L2) d = nrm[i] * nrm[j];
            else if (norm == MOVAE_MGDA_LOSS) d = ls[i] * ls[j];
            else if (norm == MOVAE_MGDA_LOSS_PLUS) d = (ls[i] * nrm[i]) * (ls[j] * nrm[j]);
            G[i][j] = Gin[i * K + j] / d;
        }
    if (stable) {  // StableMGDA (utils/torchmoo/mgda.py:286-317): clamp the eigenvalues from below, reconstruct
        double A[MAXK][MAXK], V[MAXK][MAXK];
        for (int i = 0; i < K; ++i)
            for (int j = 0; j < K; ++j) {
                A[i][j] = (double)(i <= j ? G[i][j] : G[j][i]);  // eigh reads one triangle
                V[i][j] = i == j ? 1.0 : 0.0;
            }
        jacobi_eigh(A, V, K);
        for (int i = 0; i < K; ++i)
            for (int j = 0; j < K; ++j) {
                double sacc = 0.0;
                for (int e = 0; e < K; ++e) sacc += V[i][e] * fmax(A[e][e], (double)min_eig) * V[j][e];
                G[i][j] = (float)sacc;
            }
    }
    float alpha[MAXK], Ga[MAXK];
    for (int i = 0; i < K; ++i) alpha[i] = 1.f / K;
    int it = 0;
    for (it = 0; it < max_iters; ++it) {
        int t = 0;
        for (int i = 0; i < K; ++i) {
            float s = 0.f;
            for (int j = 0; j < K; ++j) s += G[i][j] * alpha[j];
            Ga[i] = s;
            if (s < Ga[t]) t = i;  // first minimum
        }
        float a = 0.f, b = 0.f;
        for (int i = 0; i < K; ++i) {
            a += alpha[i] * G[i][t];
            b += alpha[i] * Ga[i];
        }
        const float c = G[t][t];
        float gamma;
        if (c <= a) gamma = 1.f;
        else if (b <= a) gamma = 0.f;
        else gamma = (b - a) / (b + c - 2.f * a);
        for (int i = 0; i < K; ++i) alpha[i] = (1.f - gamma) * alpha[i];
        alpha[t] += gamma;
        if (gamma < epsilon) { ++it; break; }
    }
    for (int i = 0; i < K; ++i) wout[i] = alpha[i];
    if (info) info[0] = it > max_iters ? max_iters : it;
}

// ---- Aligned-MTL: cyclic Jacobi eigen-decomposition in fp64 + balance transformation ------------------------
__global__ void amtl_k(const float* __restrict__ Gin, int K, int scale_mode, const float* __restrict__ pref,
                       float* __restrict__ wout) {
    if (threadIdx.x != 0) return;
    double A[MAXK][MAXK], V[MAXK][MAXK];
    for (int i = 0; i < K; ++i)
        for (int j = 0; j < K; ++j) {
            // eigh(UPLO="U") reads the upper triangle only
            A[i][j] = (double)(i <= j ? Gin[i * K + j] : Gin[j * K + i]);
            V[i][j] = i == j ? 1.0 : 0.0;
        }
    jacobi_eigh(A, V, K);
    double lam[MAXK];
    int order[MAXK];
    double lmax = -1e300;
    for (int i = 0; i < K; ++i) {
        lam[i] = A[i][i];
        order[i] = i;
        lmax = fmax(lmax, lam[i]);
    }
    double w0[MAXK];
    for (int i = 0; i < K; ++i) w0[i] = pref ? (double)pref[i] : 1.0 / K;
    const double tol = lmax * K * 1.1920928955078125e-07;  // torch.finfo(float32).eps
    int rank = 0;
    for (int i = 0; i < K; ++i) rank += lam[i] > tol;
    if (rank == 0) {
        for (int i = 0; i < K; ++i) wout[i] = (float)w0[i];
        return;
    }
    for (int i = 1; i < K; ++i) {  // descending insertion sort of indices
        const int oi = order[i];
        int j = i - 1;
        while (j >= 0 && lam[order[j]] < lam[oi]) {
            order[j + 1] = order[j];
            --j;
        }
        order[j + 1] = oi;
    }
    double scale;
    if (scale_mode == MOVAE_AMTL_MEDIAN) {
        scale = lam[order[rank - 1 - (rank - 1) / 2]];  // lower middle of the kept (ascending) values
    } else if (scale_mode == MOVAE_AMTL_RMSE) {
        scale = 0.0;
        for (int r = 0; r < rank; ++r) scale += lam[order[r]];
        scale /= rank;
    } else {
        scale = lam[order[rank - 1]];
    }
    // alpha = sqrt(scale) * V diag(lam^-1/2) V^T w0   over the kept eigen-pairs
    double out[MAXK];
    for (int i = 0; i < K; ++i) out[i] = 0.0;
    for (int r = 0; r < rank; ++r) {
        const int e = order[r];
        double proj = 0.0;
        for (int k = 0; k < K; ++k) proj += V[k][e] * w0[k];
        proj /= sqrt(lam[e]);
        for (int i = 0; i < K; ++i) out[i] += V[i][e] * proj;
    }
    const double sc = sqrt(scale);
    for (int i = 0; i < K; ++i) wout[i] = (float)(sc * out[i]);
}

__global__ void const_k(int K, float value, float* __restrict__ w) {
    if (threadIdx.x < K) w[threadIdx.x] = value;
}

template <int K>
int launch_gram_k(const float* J, size_t ldj, size_t m, double* part, int nb, bool vec, hipStream_t st) {
    if (vec)
        hipLaunchKernelGGL((gram_partial<K, true>), dim3(nb), dim3(256), 0, st, J, (long)ldj, (long)m, part);
    else
        hipLaunchKernelGGL((gram_partial<K, false>), dim3(nb), dim3(256), 0, st, J, (long)ldj, (long)m, part);
    MOVAE_CHECK_LAUNCH("gram_partial");
    return MOVAE_OK;
}

template <int K>
int launch_combine_k(const float* J, size_t ldj, size_t m, const float* w, float* g, int acc, bool vec, hipStream_t st) {
    const int nb = gram_blocks(m);
    if (vec)
        hipLaunchKernelGGL((combine_k<K, true>), dim3(nb), dim3(256), 0, st, J, (long)ldj, (long)m, w, g, acc);
    else
        hipLaunchKernelGGL((combine_k<K, false>), dim3(nb), dim3(256), 0, st, J, (long)ldj, (long)m, w, g, acc);
    MOVAE_CHECK_LAUNCH("combine");
    return MOVAE_OK;
}

#define DISPATCH_K(k, CALL)                                      \
    switch (k) {                                                 \
        case 1: return CALL(1);                                  \
        case 2: return CALL(2);                                  \
        case 3: return CALL(3);                                  \
        case 4: return CALL(4);                                  \
        case 5: return CALL(5);                                  \
        case 6: return CALL(6);                                  \
        case 7: return CALL(7);                                  \
        case 8: return CALL(8);                                  \
        default: break;                                          \
    }

// ---- torchjd PCGrad weighting (Yu et al. 2020, "Gradient Surgery"), on the Gramian --------------------------------
// for each task i: walk the other tasks in the order perm[i][:], and whenever the running combination conflicts with
// task j (gramian[j] . current < 0) subtract its projection: current[j] -= that inner product / G[j][j]; w = sum_i current.
// The permutations come from the caller (the reference draws them with torch.randperm on the host).  K <= 8: one lane.
__global__ __launch_bounds__(64) void pcgrad_k(const float* __restrict__ G, int K, const int32_t* __restrict__ perm,
                                               float* __restrict__ wout) {
    if (threadIdx.x != 0) return;
    float w[MAXK];
    for (int i = 0; i < K; ++i) w[i] = 0.f;
    for (int i = 0; i < K; ++i) {
        float cur[MAXK];
        for (int q = 0; q < K; ++q) cur[q] = q == i ? 1.f : 0.f;
        for (int t = 0; t < K; ++t) {
            const int j = perm[i * K + t];
            if (j == i || j < 0 || j >= K) continue;
            float ip = 0.f;
            for (int q = 0; q < K; ++q) ip += G[j * K + q] * cur[q];
            if (ip < 0.f) cur[j] -= ip / G[j * K + j];
        }
        for (int q = 0; q < K; ++q) w[q] += cur[q];
    }
    for (int i = 0; i < K; ++i) wout[i] = w[i];
}

// ---- torchjd IMTL-G weighting (Liu et al. 2021) --------------------------------------------------------------------
// v = pinv(G) d with d_i = ||row i|| = sqrt(G_ii); w = v / sum(v), or zeros when |sum(v)| < 1e-12.  The pseudo-inverse
// goes through the Jacobi eigen-decomposition with torch.linalg.pinv's default cut-off (K * eps_f32 * largest).
__global__ __launch_bounds__(64) void imtlg_k(const float* __restrict__ Gin, int K, float* __restrict__ wout) {
    if (threadIdx.x != 0) return;
    double A[MAXK][MAXK], V[MAXK][MAXK], d[MAXK], v[MAXK];
    for (int i = 0; i < K; ++i)
        for (int j = 0; j < K; ++j) {
            A[i][j] = 0.5 * ((double)Gin[i * K + j] + (double)Gin[j * K + i]);
            V[i][j] = i == j ? 1.0 : 0.0;
        }
    for (int i = 0; i < K; ++i) d[i] = sqrt(fmax((double)Gin[i * K + i], 0.0));
    jacobi_eigh(A, V, K);
    double lmax = 0.0;
    for (int i = 0; i < K; ++i) lmax = fmax(lmax, fabs(A[i][i]));
    const double cut = (double)K * 1.1920928955078125e-07 * lmax;
    for (int i = 0; i < K; ++i) v[i] = 0.0;
    for (int e = 0; e < K; ++e) {
        if (fabs(A[e][e]) <= cut) continue;
        double proj = 0.0;
        for (int i = 0; i < K; ++i) proj += V[i][e] * d[i];
        proj /= A[e][e];
        for (int i = 0; i < K; ++i) v[i] += V[i][e] * proj;
    }
    double sum = 0.0;
    for (int i = 0; i < K; ++i) sum += v[i];
    for (int i = 0; i < K; ++i) wout[i] = fabs(sum) < 1e-12 ? 0.f : (float)(v[i] / sum);
}


}  // namespace

extern "C" {

size_t movae_gram_ws_bytes(int k, size_t m) {
    if (k <= 0) return 0;
    return MOVAE_WS_HEADER_BYTES + (size_t)gram_blocks(m) * (size_t)(k * (k + 1) / 2) * sizeof(double);
}

int movae_gram(const float* J, size_t ldj, int k, size_t m, float* G, void* ws, size_t ws_bytes, movae_stream_t stream) {
    MOVAE_WS_SCRATCH(ws, ws_bytes);
    MOVAE_CHECK_ARG(J && G && m > 0 && ldj >= m, "movae_gram: bad argument");
    MOVAE_CHECK_ARG(k >= 1 && k <= MAXK, "movae_gram: k=%d outside 1..%d", k, MAXK);
    MOVAE_CHECK_ARG(ws && ws_bytes >= movae_gram_ws_bytes(k, m), "movae_gram: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    double* part = static_cast<double*>(ws);
    const int nb = gram_blocks(m);
    const bool vec = (ldj % 4 == 0) && (reinterpret_cast<uintptr_t>(J) & 15) == 0;
    int rc = MOVAE_EINVAL;
#define CALL_GRAM(KK) launch_gram_k<KK>(J, ldj, m, part, nb, vec, st)
    rc = [&]() -> int { DISPATCH_K(k, CALL_GRAM) return MOVAE_EINVAL; }();
#undef CALL_GRAM
    if (rc) return rc;
    hipLaunchKernelGGL(gram_final, dim3(k * (k + 1) / 2), dim3(64), 0, st, part, nb, k, G);
    MOVAE_CHECK_LAUNCH("gram_final");
    return MOVAE_OK;
}

int movae_combine(const float* J, size_t ldj, int k, size_t m, const float* w, float* g, int accumulate, movae_stream_t stream) {
    MOVAE_CHECK_ARG(J && w && g && m > 0 && ldj >= m, "movae_combine: bad argument");
    MOVAE_CHECK_ARG(k >= 1 && k <= MAXK, "movae_combine: k=%d outside 1..%d", k, MAXK);
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (ldj % 4 == 0) && ((reinterpret_cast<uintptr_t>(J) | reinterpret_cast<uintptr_t>(g)) & 15) == 0;
#define CALL_COMB(KK) launch_combine_k<KK>(J, ldj, m, w, g, accumulate, vec, st)
    return [&]() -> int { DISPATCH_K(k, CALL_COMB) return MOVAE_EINVAL; }();
#undef CALL_COMB
}

int movae_gd_similarity(const float* J, size_t ldj, int k, size_t m, const float* w, float* out, void* ws, size_t ws_bytes,
                        movae_stream_t stream) {
    MOVAE_WS_SCRATCH(ws, ws_bytes);
    MOVAE_CHECK_ARG(J && w && out && m > 0 && k >= 1 && k <= MAXK, "movae_gd_similarity: bad argument");
    const int nb = gram_blocks(m);
    MOVAE_CHECK_ARG(ws && ws_bytes >= (size_t)nb * 3 * sizeof(double), "movae_gd_similarity: workspace too small");
    double* part = static_cast<double*>(ws);
    hipLaunchKernelGGL(similarity_partial, dim3(nb), dim3(256), 0, (hipStream_t)stream, J, (long)ldj, k, (long)m, w, part);
    MOVAE_CHECK_LAUNCH("similarity_partial");
    hipLaunchKernelGGL(similarity_final, dim3(1), dim3(64), 0, (hipStream_t)stream, part, nb, out);
    MOVAE_CHECK_LAUNCH("similarity_final");
    return MOVAE_OK;
}

int movae_weights_upgrad_norm(const float* G, int k, int norm_mode, float norm_eps, float reg_eps, const float* pref, float* w,
                              movae_stream_t stream) {
    MOVAE_CHECK_ARG(G && w, "movae_weights_upgrad: null pointer");
    MOVAE_CHECK_ARG(k >= 1 && k <= MAXK, "movae_weights_upgrad: k=%d outside 1..%d", k, MAXK);
    MOVAE_CHECK_ARG(norm_mode >= 0 && norm_mode <= 2, "movae_weights_upgrad: unknown normalisation %d", norm_mode);
    launch_upgrad(G, k, norm_mode, norm_eps, reg_eps, pref, w, 0, (hipStream_t)stream);
    MOVAE_CHECK_LAUNCH("upgrad");
    return MOVAE_OK;
}

// movae_gram + movae_weights_upgrad_norm / movae_weights_dualproj in two launches instead of three: the solver kernel folds the
// Gramian's block partials itself (gram_final's arithmetic) and writes G too.
int movae_gram_upgrad(const float* J, size_t ldj, int k, size_t m, float* G, int norm_mode, float norm_eps, float reg_eps,
                      const float* pref, float* w, int dual, void* ws, size_t ws_bytes, movae_stream_t stream) {
    MOVAE_WS_SCRATCH(ws, ws_bytes);
    MOVAE_CHECK_ARG(J && G && w && m > 0 && ldj >= m, "movae_gram_upgrad: bad argument");
    MOVAE_CHECK_ARG(k >= 1 && k <= MAXK, "movae_gram_upgrad: k=%d outside 1..%d", k, MAXK);
    MOVAE_CHECK_ARG(norm_mode >= 0 && norm_mode <= 2, "movae_gram_upgrad: unknown normalisation %d", norm_mode);
    MOVAE_CHECK_ARG(ws && ws_bytes >= movae_gram_ws_bytes(k, m), "movae_gram_upgrad: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    double* part = static_cast<double*>(ws);
    const int nb = gram_blocks(m);
    const bool vec = (ldj % 4 == 0) && (reinterpret_cast<uintptr_t>(J) & 15) == 0;
    int rc = MOVAE_EINVAL;
#define CALL_GRAM(KK) launch_gram_k<KK>(J, ldj, m, part, nb, vec, st)
    rc = [&]() -> int { DISPATCH_K(k, CALL_GRAM) return MOVAE_EINVAL; }();
#undef CALL_GRAM
    if (rc) return rc;
    launch_upgrad(nullptr, k, norm_mode, norm_eps, reg_eps, pref, w, dual ? 1 : 0, st, part, nb, G);
    MOVAE_CHECK_LAUNCH("upgrad (with the Gramian's final fold)");
    return MOVAE_OK;
}

int movae_weights_dualproj(const float* G, int k, float norm_eps, float reg_eps, const float* pref, float* w, movae_stream_t stream) {
    MOVAE_CHECK_ARG(G && w, "movae_weights_dualproj: null pointer");
    MOVAE_CHECK_ARG(k >= 1 && k <= MAXK, "movae_weights_dualproj: k=%d outside 1..%d", k, MAXK);
    launch_upgrad(G, k, 0, norm_eps, reg_eps, pref, w, 1, (hipStream_t)stream);
    MOVAE_CHECK_LAUNCH("dualproj");
    return MOVAE_OK;
}

int movae_weights_cagrad(const float* G, int k, float c, float norm_eps, float* w, movae_stream_t stream) {
    MOVAE_CHECK_ARG(G && w, "movae_weights_cagrad: null pointer");
    MOVAE_CHECK_ARG(k >= 1 && k <= MAXK, "movae_weights_cagrad: k=%d outside 1..%d", k, MAXK);
    MOVAE_CHECK_ARG(c >= 0.f, "Parameter `c` should be a non-negative float. Found `c = %g`.", (double)c);
    launch_cagrad(G, k, c, norm_eps, w, (hipStream_t)stream);
    MOVAE_CHECK_LAUNCH("cagrad");
    return MOVAE_OK;
}

int movae_weights_pcgrad(const float* G, int k, const int32_t* perm, float* w, movae_stream_t stream) {
    MOVAE_CHECK_ARG(G && w && perm, "movae_weights_pcgrad: null pointer");
    MOVAE_CHECK_ARG(k >= 1 && k <= MAXK, "movae_weights_pcgrad: k=%d outside 1..%d", k, MAXK);
    hipLaunchKernelGGL(pcgrad_k, dim3(1), dim3(64), 0, (hipStream_t)stream, G, k, perm, w);
    MOVAE_CHECK_LAUNCH("pcgrad");
    return MOVAE_OK;
}

int movae_weights_imtlg(const float* G, int k, float* w, movae_stream_t stream) {
    MOVAE_CHECK_ARG(G && w, "movae_weights_imtlg: null pointer");
    MOVAE_CHECK_ARG(k >= 1 && k <= MAXK, "movae_weights_imtlg: k=%d outside 1..%d", k, MAXK);
    hipLaunchKernelGGL(imtlg_k, dim3(1), dim3(64), 0, (hipStream_t)stream, G, k, w);
    MOVAE_CHECK_LAUNCH("imtlg");
    return MOVAE_OK;
}

int movae_weights_upgrad(const float* G, int k, float norm_eps, float reg_eps, const float* pref, float* w, movae_stream_t stream) {
    return movae_weights_upgrad_norm(G, k, MOVAE_UPGRAD_TRACE, norm_eps, reg_eps, pref, w, stream);
}

int movae_weights_mgda_stable(const float* G, int k, int norm, const float* losses, float epsilon, int max_iters, float min_eigenvalue,
                              float* w, int32_t* info, movae_stream_t stream) {
    MOVAE_CHECK_ARG(G && w, "movae_weights_mgda: null pointer");
    MOVAE_CHECK_ARG(k >= 1 && k <= MAXK, "movae_weights_mgda: k=%d outside 1..%d", k, MAXK);
    MOVAE_CHECK_ARG(norm >= 0 && norm <= 3, "movae_weights_mgda: unknown norm %d", norm);
    MOVAE_CHECK_ARG(!(norm >= MOVAE_MGDA_LOSS && !losses),
                    "Losses must be set before calling forward() when using norm_type='loss'/'loss+'");
    hipLaunchKernelGGL(mgda_k, dim3(1), dim3(64), 0, (hipStream_t)stream, G, k, norm, losses, epsilon, max_iters, w, info, 1,
                       min_eigenvalue);
    MOVAE_CHECK_LAUNCH("mgda");
    return MOVAE_OK;
}

int movae_weights_mgda(const float* G, int k, int norm, const float* losses, float epsilon, int max_iters, float* w,
                       int32_t* info, movae_stream_t stream) {
    MOVAE_CHECK_ARG(G && w, "movae_weights_mgda: null pointer");
    MOVAE_CHECK_ARG(k >= 1 && k <= MAXK, "movae_weights_mgda: k=%d outside 1..%d", k, MAXK);
    MOVAE_CHECK_ARG(norm >= 0 && norm <= 3, "movae_weights_mgda: unknown norm %d", norm);
    MOVAE_CHECK_ARG(!(norm >= MOVAE_MGDA_LOSS && !losses),
                    "Losses must be set before calling forward() when using norm_type='loss'/'loss+'");
    hipLaunchKernelGGL(mgda_k, dim3(1), dim3(64), 0, (hipStream_t)stream, G, k, norm, losses, epsilon, max_iters, w, info, 0, 0.f);
    MOVAE_CHECK_LAUNCH("mgda");
    return MOVAE_OK;
}

int movae_weights_amtl(const float* G, int k, int scale_mode, const float* pref, float* w, movae_stream_t stream) {
    MOVAE_CHECK_ARG(G && w, "movae_weights_amtl: null pointer");
    MOVAE_CHECK_ARG(k >= 1 && k <= MAXK, "movae_weights_amtl: k=%d outside 1..%d", k, MAXK);
    MOVAE_CHECK_ARG(scale_mode >= 0 && scale_mode <= 2, "Invalid scale_mode=%d. Expected 'min', 'median', or 'rmse'.", scale_mode);
    hipLaunchKernelGGL(amtl_k, dim3(1), dim3(64), 0, (hipStream_t)stream, G, k, scale_mode, pref, w);
    MOVAE_CHECK_LAUNCH("amtl");
    return MOVAE_OK;
}

int movae_weights_const(int k, float value, float* w, movae_stream_t stream) {
    MOVAE_CHECK_ARG(w && k >= 1 && k <= MAXK, "movae_weights_const: bad argument");
    hipLaunchKernelGGL(const_k, dim3(1), dim3(64), 0, (hipStream_t)stream, k, value, w);
    MOVAE_CHECK_LAUNCH("const_w");
    return MOVAE_OK;
}

}  // extern "C"
