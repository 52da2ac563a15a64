// Vector-quantiser codebook lookup (models/vq_vae.py:27-64) for gfx950.
// Distances are the contraction  E[K x D] . X^T[D x rows]  on v_mfma_f32_32x32x2_f32 with the
// CODE index on the accumulator rows and the LATENT row on the lane, so the arg-min over codes is
// lane-local (16 registers + one cross-half exchange) -- no K-wide distance matrix, no one-hot
// matrix and no one-hot GEMM ever touch HBM (the reference materialises two [rows x K] fp32
// temporaries).  The codebook chunk is staged once per block in LDS with a +1 padded row.
#include "common.h"
#include <hipcub/hipcub.hpp>

namespace {

constexpr int CH = 128;        // codes staged per chunk
// rows per block: 64, or 32 when there are few rows (a block per CU at least; measured: 8192 rows 41.6 -> 33.6 us with 32)
inline int vq_rows_per_block(int rows) { return rows <= 16384 ? 32 : 64; }

// Block = ROWS_PER_BLOCK (32 | 64) latent rows x all K codes.  Waves (rw, cw): rw = which 32 rows, cw = which share of every
// staged chunk's four 32-code tiles -- the waves of a row group hold disjoint code sets and merge their arg-min through LDS
// (lower distance, then lower index: torch.argmin's first-index rule).  64 rows per block put two blocks on a CU even at C3's 32768 rows
// (one 128-row block per CU left the chunk staging and the gather epilogue -- 75 of 91 us -- with nothing to overlap with);
// the chunk is staged with 16-byte loads and its squared norms are folded from the values in flight.
template <int D, int ROWS_PER_BLOCK>
__global__ __launch_bounds__(256) void vq_nearest_mfma(const float* __restrict__ x, const float* __restrict__ e,
                                                       float* __restrict__ q, int64_t* __restrict__ idx,
                                                       double* __restrict__ sse_part, int* __restrict__ used, int rows,
                                                       int K) {
    constexpr int NRW = ROWS_PER_BLOCK / 32, NCW = 4 / NRW, TPW = 4 / NCW;  // row groups, code splits, 32-code tiles per wave and chunk
    constexpr int LD = D + 1;
    constexpr int QR = D / 4;  // 16-byte pieces (threads) per code row while staging: 2 .. 32, a power of two
    __shared__ float Es[CH * LD];
    __shared__ float ee[CH];
    __shared__ double shd[4];
    __shared__ float mbest[NCW][ROWS_PER_BLOCK];
    __shared__ int mbesti[NCW][ROWS_PER_BLOCK];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int rw = wave % NRW, cw = wave / NRW;
    const int rowbase = blockIdx.x * ROWS_PER_BLOCK + rw * 32;
    const int row = rowbase + l31;
    const bool rv = row < rows;
    float xf[D / 2];
    float xx = 0.f;
#pragma unroll
    for (int s = 0; s < D / 2; ++s) {
        xf[s] = rv ? x[(long)row * D + 2 * s + half] : 0.f;
        xx += xf[s] * xf[s];
    }
    xx += __shfl_xor(xx, 32, 64);

    float best = INFINITY;
    int besti = 0;
    for (int c0 = 0; c0 < K; c0 += CH) {
        __syncthreads();
        for (int i = t; i < CH * QR; i += 256) {
            const int r = i / QR, qd = i - r * QR;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c0 + r < K) v = *reinterpret_cast<const f32x4*>(e + (long)(c0 + r) * D + qd * 4);
            float* dst = Es + r * LD + qd * 4;
            dst[0] = v[0], dst[1] = v[1], dst[2] = v[2], dst[3] = v[3];
            float sq = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
#pragma unroll
            for (int o = 1; o < QR; o <<= 1) sq += __shfl_xor(sq, o, 64);  // the QR lanes of one code row are neighbours
            if (qd == 0) ee[r] = sq;
        }
        __syncthreads();
        const int ntile = min(CH, K - c0);
#pragma unroll
        for (int cj = 0; cj < TPW; ++cj) {
            const int ct = cw * TPW + cj;
            if (ct * 32 >= ntile) break;  // (wave-uniform)
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* ep = Es + (ct * 32 + l31) * LD + half;
#pragma unroll
            for (int s = 0; s < D / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ep[2 * s], xf[s], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cl = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int code = c0 + cl;
                if (code < K) {
                    const float dist = (xx + ee[cl]) - 2.f * acc[r];
                    if (dist < best) {
                        best = dist;
                        besti = code;
                    }
                }
            }
        }
    }
    {   // merge the two halves of the wave (same latent row, disjoint code sets) ...
        const float ob = __shfl_xor(best, 32, 64);
        const int oi = __shfl_xor(besti, 32, 64);
        if (ob < best || (ob == best && oi < besti)) {
            best = ob;
            besti = oi;
        }
    }
    // ... and the NCW waves of the row group
    if (half == 0) mbest[cw][rw * 32 + l31] = best, mbesti[cw][rw * 32 + l31] = besti;
    __syncthreads();
    if (cw == 0 && half == 0) {
#pragma unroll
        for (int o = 1; o < NCW; ++o) {
            const float ob = mbest[o][rw * 32 + l31];
            const int oi = mbesti[o][rw * 32 + l31];
            if (ob < best || (ob == best && oi < besti)) best = ob, besti = oi;
        }
        mbesti[0][rw * 32 + l31] = besti;
        if (rv) {
            idx[row] = besti;
            used[besti] = 1;
        }
    }
    __syncthreads();
    // gather q = E[idx] and accumulate sum (q - x)^2: every wave takes a quarter of the block's rows, four rows in flight
    double sse = 0.0;
    constexpr int RPW = ROWS_PER_BLOCK / 4;
    const int r0 = blockIdx.x * ROWS_PER_BLOCK + wave * RPW;
#pragma unroll 4
    for (int r = 0; r < RPW; ++r) {
        const int rr = r0 + r;
        const int bi = mbesti[0][wave * RPW + r];
        if (rr < rows)
            for (int d = lane; d < D; d += 64) {
                const float qv = e[(long)bi * D + d];
                const float dv = qv - x[(long)rr * D + d];
                q[(long)rr * D + d] = qv;
                sse += (double)(dv * dv);
            }
    }
    sse = block_sum_256(sse, shd);
    if (t == 0) sse_part[blockIdx.x] = sse;
}

// generic fallback (any D): one thread per latent row
__global__ __launch_bounds__(256) void vq_nearest_generic(const float* __restrict__ x, const float* __restrict__ e,
                                                          float* __restrict__ q, int64_t* __restrict__ idx,
                                                          double* __restrict__ sse_part, int* __restrict__ used, int rows,
                                                          int K, int D) {
    __shared__ double shd[4];
    const int row = blockIdx.x * 256 + threadIdx.x;
    double sse = 0.0;
    if (row < rows) {
        float xx = 0.f;
        for (int d = 0; d < D; ++d) xx += x[(long)row * D + d] * x[(long)row * D + d];
        float best = INFINITY;
        int besti = 0;
        for (int c = 0; c < K; ++c) {
            float ee = 0.f, dot = 0.f;
            for (int d = 0; d < D; ++d) {
                const float ev = e[(long)c * D + d];
                ee += ev * ev;
                dot += ev * x[(long)row * D + d];
            }
            const float dist = (xx + ee) - 2.f * dot;
            if (dist < best) {
                best = dist;
                besti = c;
            }
        }
        idx[row] = besti;
        used[besti] = 1;
        for (int d = 0; d < D; ++d) {
            const float qv = e[(long)besti * D + d];
            const float dv = qv - x[(long)row * D + d];
            q[(long)row * D + d] = qv;
            sse += (double)(dv * dv);
        }
    }
    sse = block_sum_256(sse, shd);
    if (threadIdx.x == 0) sse_part[blockIdx.x] = sse;
}

// The "code i was chosen" flags of the nearest-code kernels.  They must read zero when a lookup starts: kept in a device array of the
// library's own (not in the caller's scratch arena, which other calls overwrite) they are cleared by vq_finalize right after it has
// counted them -- no memset launch in front of every lookup.  (Up to VQ_FLAGS_MAX codes; one lookup at a time per process.)
constexpr int VQ_FLAGS_MAX = 16384;
__device__ int g_vq_used[VQ_FLAGS_MAX];

__global__ __launch_bounds__(256) void vq_finalize(const double* __restrict__ part, int nblk, int* __restrict__ used,
                                                   int K, float* __restrict__ sse, int* __restrict__ used_count,
                                                   float* __restrict__ mse2, float numel, int clear) {
    __shared__ double shd[4];
    double s = 0.0, u = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) s += part[i];
    for (int i = threadIdx.x; i < K; i += 256) {
        u += used[i] ? 1.0 : 0.0;
        if (clear) used[i] = 0;
    }
    s = block_sum_256(s, shd);
    u = block_sum_256(u, shd);
    if (threadIdx.x == 0) {
        sse[0] = (float)s;
        if (mse2) mse2[0] = mse2[1] = (float)s / numel;  // commitment and embedding terms: mse(q, x), the fp32 division torch would do
        if (used_count) used_count[0] = (int)(u + 0.5);
    }
}

// dx = dq + gc * 2 (x - q) / numel   (straight-through estimator + commitment term)
__global__ void vq_bwd_dx_k(const float* __restrict__ x, const float* __restrict__ q, const float* __restrict__ dq,
                            const float* __restrict__ gc, float* __restrict__ dx, long total, float inv_numel) {
    const float fc = gc ? gc[0] * 2.f * inv_numel : 0.f;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
        dx[i] = (dq ? dq[i] : 0.f) + fc * (x[i] - q[i]);
}

// ---- codebook gradient: de[k] = ge * 2 / numel * sum_{rows r with idx[r] == k} (q[r] - x[r]) -------------------------------
// A scatter-add with float atomics (what ATen's index_add / embedding backward does) is neither deterministic nor fast
// when few codes are in use (every row hits the same 64 addresses).  Here the rows are sorted by (code, row) with one
// radix sort of 64-bit keys, each code's contiguous segment is cut into work items of VQ_CHUNK rows that are summed in
// ascending row order, and the item partials are folded in fixed order: bit-reproducible, no atomics, no memset.
constexpr int VQ_CHUNK = 128;  // rows per work item of the segmented sum

__global__ void vq_keys_k(const int64_t* __restrict__ idx, unsigned long long* __restrict__ keys, int rows, int K) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) {
        // clamped like the forward lookups (embedding_fwd_k, vq gather): a code outside [0, K) must not index the segment tables
        long c = idx[r];
        c = c < 0 ? 0 : (c >= K ? K - 1 : c);
        keys[r] = ((unsigned long long)c << 32) | (unsigned)r;
    }
}

__device__ __forceinline__ int lower_bound_key(const unsigned long long* __restrict__ keys, int n, unsigned long long key) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// One block: seg_lo[k] = first sorted position of code k (seg_lo[K] = rows), item_base[k] = exclusive prefix sum of
// ceil(count_k / VQ_CHUNK) -- code k's segment is cut into that many work items so that a code holding most of the rows
// (early training: a handful of codes in use) is summed by many blocks instead of one.
__global__ __launch_bounds__(1024) void vq_segments_k(const unsigned long long* __restrict__ sorted, int rows, int K,
                                                       int* __restrict__ seg_lo, int* __restrict__ item_base) {
    __shared__ int sh[1024];
    __shared__ int carry;
    const int t = threadIdx.x;
    if (t == 0) carry = 0;
    for (int k = t; k <= K; k += 1024) seg_lo[k] = lower_bound_key(sorted, rows, (unsigned long long)k << 32);
    __syncthreads();
    for (int k0 = 0; k0 < K; k0 += 1024) {
        const int k = k0 + t;
        const int items = k < K ? (seg_lo[k + 1] - seg_lo[k] + VQ_CHUNK - 1) / VQ_CHUNK : 0;
        sh[t] = items;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {  // inclusive Hillis-Steele scan
            const int v = t >= off ? sh[t - off] : 0;
            __syncthreads();
            sh[t] += v;
            __syncthreads();
        }
        if (k < K) item_base[k] = carry + sh[t] - items;
        __syncthreads();
        if (t == 1023) carry += sh[1023];
        __syncthreads();
    }
    if (t == 0) item_base[K] = carry;
}

// block b = work item b: VQ_CHUNK consecutive sorted rows of one code; part[b][d] = sum_rows (q - x) in ascending row order
__global__ __launch_bounds__(256) void vq_embed_part_k(const float* __restrict__ x, const float* __restrict__ q,
                                                       const unsigned long long* __restrict__ sorted, const int* __restrict__ seg_lo,
                                                       const int* __restrict__ item_base, float* __restrict__ part, int K, int D) {
    __shared__ float sh[256];
    __shared__ int code;
    const int b = blockIdx.x, t = threadIdx.x;
    if (b >= item_base[K]) return;
    if (t == 0) {  // largest k with item_base[k] <= b
        int lo = 0, hi = K;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (item_base[mid] <= b) lo = mid;
            else hi = mid;
        }
        code = lo;
    }
    __syncthreads();
    const int k = code;
    const int lo = seg_lo[k] + (b - item_base[k]) * VQ_CHUNK, hi = min(seg_lo[k + 1], lo + VQ_CHUNK);
    int DL = 1;
    while (DL < D && DL < 256) DL <<= 1;
    const int slots = 256 / DL, dl = t % DL, slot = t / DL;
    for (int d0 = 0; d0 < D; d0 += DL) {
        const int d = d0 + dl;
        float acc = 0.f;
        if (d < D)
            for (int i = lo + slot; i < hi; i += slots) {
                const long r = (long)(sorted[i] & 0xffffffffu);
                acc += x ? q[r * D + d] - x[r * D + d] : q[r * D + d];  // x == nullptr: a plain segmented sum of q's rows
            }
        sh[t] = acc;
        __syncthreads();
        if (slot == 0 && d < D) {
            float s = 0.f;
            for (int j = 0; j < slots; ++j) s += sh[j * DL + dl];  // fixed order
            part[(long)b * D + d] = s;
        }
        __syncthreads();
    }
}

// block k: de[k][d] = ge * 2 / numel * (sum of code k's item partials, in item order); zero for unused codes
__global__ __launch_bounds__(256) void vq_embed_final_k(const float* __restrict__ part, const int* __restrict__ item_base,
                                                        const float* __restrict__ ge, float* __restrict__ de, int D, float inv_numel) {
    __shared__ float sh[256];
    const int k = blockIdx.x, t = threadIdx.x;
    const float fe = ge ? ge[0] * 2.f * inv_numel : 1.f;
    const int b0 = item_base[k], b1 = item_base[k + 1];
    int DL = 1;
    while (DL < D && DL < 256) DL <<= 1;
    const int slots = 256 / DL, dl = t % DL, slot = t / DL;
    for (int d0 = 0; d0 < D; d0 += DL) {
        const int d = d0 + dl;
        float acc = 0.f;
        if (d < D)
            for (int b = b0 + slot; b < b1; b += slots) acc += part[(long)b * D + d];
        sh[t] = acc;
        __syncthreads();
        if (slot == 0 && d < D) {
            float s = 0.f;
            for (int j = 0; j < slots; ++j) s += sh[j * DL + dl];
            de[(long)k * D + d] = fe * s;
        }
        __syncthreads();
    }
}

// ---- the (code, row) order without a sort ------------------------------------------------------------------------------------------
// The radix sort above is seven launches of rocprim for 8-32 k keys.  The order it produces -- codes ascending, rows ascending within a
// code -- is a counting placement: (1) every block of 256 consecutive rows counts its codes (integer LDS atomics: the counts do not
// depend on the order of the adds), (2) one block turns the per-block counts into each block's starting offset inside every code's
// segment and the counts' totals into seg_lo / item_base (what vq_segments_k computes from the sorted keys), (3) every block places
// its rows: position = seg_lo[code] + offset of the block in that code + number of EARLIER rows of the block with the same code.
// Three launches, the same `sorted` array to the last element (so the sums below are bit-identical).
constexpr int VQ_PLACE_MAX_K = 4096;  // codes whose per-block histogram fits the kernels' LDS

__device__ __forceinline__ int vq_code_of(const int64_t* __restrict__ idx, int r, int K) {
    long c = idx[r];
    return (int)(c < 0 ? 0 : (c >= K ? K - 1 : c));  // clamped like vq_keys_k
}

__global__ __launch_bounds__(256) void vq_hist_k(const int64_t* __restrict__ idx, int rows, int K, int* __restrict__ blk_hist) {
    extern __shared__ int hist[];
    const int t = threadIdx.x, r = blockIdx.x * 256 + t;
    for (int k = t; k < K; k += 256) hist[k] = 0;
    __syncthreads();
    if (r < rows) atomicAdd(&hist[vq_code_of(idx, r, K)], 1);
    __syncthreads();
    for (int k = t; k < K; k += 256) blk_hist[(long)blockIdx.x * K + k] = hist[k];
}

// one block: blk_hist[b][k] -> blk_off[b][k] = rows of code k in blocks < b; seg_lo / item_base as vq_segments_k
__global__ __launch_bounds__(1024) void vq_scan_k(const int* __restrict__ blk_hist, int nblk, int rows, int K, int* __restrict__ blk_off,
                                                   int* __restrict__ seg_lo, int* __restrict__ item_base) {
    __shared__ int sh[1024];
    __shared__ int carry_s, carry_i;
    const int t = threadIdx.x;
    if (t == 0) carry_s = 0, carry_i = 0;
    __syncthreads();
    for (int k0 = 0; k0 < K; k0 += 1024) {
        const int k = k0 + t;
        int count = 0;
        if (k < K)
            for (int b = 0; b < nblk; ++b) {  // (coalesced across the codes)
                blk_off[(long)b * K + k] = count;
                count += blk_hist[(long)b * K + k];
            }
        const int items = (count + VQ_CHUNK - 1) / VQ_CHUNK;
        // two inclusive Hillis-Steele scans over the 1024 codes of this round: rows, then work items
        sh[t] = count;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int v = t >= off ? sh[t - off] : 0;
            __syncthreads();
            sh[t] += v;
            __syncthreads();
        }
        if (k < K) seg_lo[k] = carry_s + sh[t] - count;
        __syncthreads();
        const int tot_s = sh[1023];
        __syncthreads();
        sh[t] = items;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int v = t >= off ? sh[t - off] : 0;
            __syncthreads();
            sh[t] += v;
            __syncthreads();
        }
        if (k < K) item_base[k] = carry_i + sh[t] - items;
        __syncthreads();
        if (t == 0) carry_s += tot_s, carry_i += sh[1023];
        __syncthreads();
    }
    if (t == 0) seg_lo[K] = rows, item_base[K] = carry_i;
}

__global__ __launch_bounds__(256) void vq_place_k(const int64_t* __restrict__ idx, int rows, int K, const int* __restrict__ seg_lo,
                                                  const int* __restrict__ blk_off, unsigned long long* __restrict__ sorted) {
    __shared__ int codes[256];
    const int t = threadIdx.x, r = blockIdx.x * 256 + t;
    const int c = r < rows ? vq_code_of(idx, r, K) : -1;
    codes[t] = c;
    __syncthreads();
    if (r >= rows) return;
    int rank = 0;
    for (int j = 0; j < t; ++j) rank += codes[j] == c ? 1 : 0;  // earlier rows of this block with the same code
    const int pos = seg_lo[c] + blk_off[(long)blockIdx.x * K + c] + rank;
    sorted[pos] = ((unsigned long long)c << 32) | (unsigned)r;
}

template <int D>
void launch_mfma(const float* x, const float* e, float* q, int64_t* idx, double* part, int* used, int rows, int K, int nblk,
                 hipStream_t st) {
    if (vq_rows_per_block(rows) == 32)
        hipLaunchKernelGGL((vq_nearest_mfma<D, 32>), dim3(nblk), dim3(256), 0, st, x, e, q, idx, part, used, rows, K);
    else
        hipLaunchKernelGGL((vq_nearest_mfma<D, 64>), dim3(nblk), dim3(256), 0, st, x, e, q, idx, part, used, rows, K);
}

// de[k] = f * sum_{rows r with idx[r] == k} (q[r] - x[r])   (x may be null: plain sum of q's rows; ge null: f = 1)
static int segmented_code_sum(const float* x, const float* q, const int64_t* idx, const float* ge, float* de, int rows, int k, int d,
                              void* ws, hipStream_t st) {
    const long total = (long)rows * d;
    char* base = static_cast<char*>(ws);
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(base);
    unsigned long long* sorted = keys + rows;
    const int max_items = rows / VQ_CHUNK + k + 1;
    float* part = reinterpret_cast<float*>(sorted + rows);
    int* seg_lo = reinterpret_cast<int*>(part + (size_t)max_items * d);
    int* item_base = seg_lo + k + 2;
    void* temp = reinterpret_cast<void*>(item_base + k + 2);
    temp = reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(temp) + 255) & ~(uintptr_t)255);
    static const bool no_place = getenv("MOVAE_VQ_RADIX") && atoi(getenv("MOVAE_VQ_RADIX")) != 0;  // (A/B knob: the radix-sort path)
    if (k <= VQ_PLACE_MAX_K && !no_place) {  // counting placement: three launches instead of the sort's nine, the same order
        const int nblk = ceil_div(rows, 256);
        int* blk_hist = reinterpret_cast<int*>(temp);
        int* blk_off = blk_hist + (size_t)nblk * k;
        hipLaunchKernelGGL(vq_hist_k, dim3(nblk), dim3(256), (size_t)k * sizeof(int), st, idx, rows, k, blk_hist);
        MOVAE_CHECK_LAUNCH("vq_hist");
        hipLaunchKernelGGL(vq_scan_k, dim3(1), dim3(1024), 0, st, blk_hist, nblk, rows, k, blk_off, seg_lo, item_base);
        MOVAE_CHECK_LAUNCH("vq_scan");
        hipLaunchKernelGGL(vq_place_k, dim3(nblk), dim3(256), 0, st, idx, rows, k, seg_lo, blk_off, sorted);
        MOVAE_CHECK_LAUNCH("vq_place");
        hipLaunchKernelGGL(vq_embed_part_k, dim3(max_items), dim3(256), 0, st, x, q, sorted, seg_lo, item_base, part, k, d);
        MOVAE_CHECK_LAUNCH("vq_embed_part");
        hipLaunchKernelGGL(vq_embed_final_k, dim3(k), dim3(256), 0, st, part, item_base, ge, de, d, 1.f / (float)total);
        MOVAE_CHECK_LAUNCH("vq_embed_final");
        return MOVAE_OK;
    }
    size_t temp_bytes = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, temp_bytes, keys, sorted, rows, 0, 64);
    hipLaunchKernelGGL(vq_keys_k, dim3(ceil_div(rows, 256)), dim3(256), 0, st, idx, keys, rows, k);
    MOVAE_CHECK_LAUNCH("vq_keys");
    int kbits = 1;
    while ((1 << kbits) < k) ++kbits;
    // keys are (code << 32 | row) in row order: a STABLE sort on the code bits alone leaves every code's rows ascending -- the order of
    // the full 41-bit sort at a quarter of its digit passes
    if (hipcub::DeviceRadixSort::SortKeys(temp, temp_bytes, keys, sorted, rows, 32, 32 + kbits, st) != hipSuccess) {
        movae_set_error("segmented_code_sum: radix sort failed");
        return MOVAE_ELAUNCH;
    }
    hipLaunchKernelGGL(vq_segments_k, dim3(1), dim3(1024), 0, st, sorted, rows, k, seg_lo, item_base);
    MOVAE_CHECK_LAUNCH("vq_segments");
    hipLaunchKernelGGL(vq_embed_part_k, dim3(max_items), dim3(256), 0, st, x, q, sorted, seg_lo, item_base, part, k, d);
    MOVAE_CHECK_LAUNCH("vq_embed_part");
    hipLaunchKernelGGL(vq_embed_final_k, dim3(k), dim3(256), 0, st, part, item_base, ge, de, d, 1.f / (float)total);
    MOVAE_CHECK_LAUNCH("vq_embed_final");
    return MOVAE_OK;
}

}  // namespace

extern "C" {

static int vq_nearest_impl(const float* x, const float* e, float* q, int64_t* idx, float* sse, int32_t* used_count, float* mse2, int rows,
                           int k, int d, void* ws, size_t ws_bytes, movae_stream_t stream) {
    MOVAE_WS_SCRATCH(ws, ws_bytes);
    MOVAE_CHECK_ARG(x && e && q && idx && sse, "movae_vq_nearest_fwd: null pointer");
    MOVAE_CHECK_ARG(rows > 0 && k > 0 && d > 0, "movae_vq_nearest_fwd: bad shape rows=%d k=%d d=%d", rows, k, d);
    hipStream_t st = (hipStream_t)stream;
    const bool mfma = d == 8 || d == 16 || d == 32 || d == 64 || d == 128;
    const int nblk = mfma ? ceil_div(rows, vq_rows_per_block(rows)) : ceil_div(rows, 256);
    const size_t need = (size_t)nblk * sizeof(double) + (size_t)k * sizeof(int);
    MOVAE_CHECK_ARG(ws && ws_bytes >= need, "movae_vq_nearest_fwd: workspace too small (%zu < %zu)", ws_bytes, need);
    double* part = static_cast<double*>(ws);
    int* used = reinterpret_cast<int*>(part + nblk);
    static int* own_flags = nullptr;  // g_vq_used's device address (zero-initialised with the code object, self-cleaning afterwards)
    if (!own_flags && hipGetSymbolAddress(reinterpret_cast<void**>(&own_flags), HIP_SYMBOL(g_vq_used)) != hipSuccess) own_flags = nullptr;
    const bool self_clean = own_flags != nullptr && k <= VQ_FLAGS_MAX;
    if (self_clean) {
        used = own_flags;
    } else if (hipMemsetAsync(used, 0, (size_t)k * sizeof(int), st) != hipSuccess) {
        movae_set_error("movae_vq_nearest_fwd: memset failed");
        return MOVAE_ELAUNCH;
    }
    switch (mfma ? d : 0) {
        case 8: launch_mfma<8>(x, e, q, idx, part, used, rows, k, nblk, st); break;
        case 16: launch_mfma<16>(x, e, q, idx, part, used, rows, k, nblk, st); break;
        case 32: launch_mfma<32>(x, e, q, idx, part, used, rows, k, nblk, st); break;
        case 64: launch_mfma<64>(x, e, q, idx, part, used, rows, k, nblk, st); break;
        case 128: launch_mfma<128>(x, e, q, idx, part, used, rows, k, nblk, st); break;
        default:
            hipLaunchKernelGGL(vq_nearest_generic, dim3(nblk), dim3(256), 0, st, x, e, q, idx, part, used, rows, k, d);
    }
    MOVAE_CHECK_LAUNCH("vq_nearest");
    hipLaunchKernelGGL(vq_finalize, dim3(1), dim3(256), 0, st, part, nblk, used, k, sse, used_count, mse2, (float)rows * (float)d,
                       self_clean ? 1 : 0);
    MOVAE_CHECK_LAUNCH("vq_finalize");
    return MOVAE_OK;
}

int movae_vq_nearest_fwd(const float* x, const float* e, float* q, int64_t* idx, float* sse, int32_t* used_count, int rows,
                         int k, int d, void* ws, size_t ws_bytes, movae_stream_t stream) {
    return vq_nearest_impl(x, e, q, idx, sse, used_count, nullptr, rows, k, d, ws, ws_bytes, stream);
}

// ... and the two loss terms of models/vq_vae.py:51-52 from the same finalize kernel: mse2[0] = mse2[1] = sse / (rows * d)
// (commitment = mse(q.detach(), x), embedding = mse(q, x.detach()): one value, two tape nodes) -- no division / clone launches
int movae_vq_nearest_fwd_mse(const float* x, const float* e, float* q, int64_t* idx, float* sse, int32_t* used_count, float* mse2,
                             int rows, int k, int d, void* ws, size_t ws_bytes, movae_stream_t stream) {
    MOVAE_CHECK_ARG(mse2, "movae_vq_nearest_fwd_mse: null pointer");
    return vq_nearest_impl(x, e, q, idx, sse, used_count, mse2, rows, k, d, ws, ws_bytes, stream);
}

size_t movae_vq_bwd_ws_bytes(int rows, int k, int d) {
    if (rows <= 0 || k <= 0 || d <= 0) return 0;
    size_t temp = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, temp, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, rows, 0, 64);
    const size_t items = (size_t)rows / VQ_CHUNK + k + 1;
    // (the counting placement keeps two per-block histograms where the sort keeps its temporary storage)
    const size_t place = k <= VQ_PLACE_MAX_K ? 2 * (size_t)ceil_div(rows, 256) * k * sizeof(int) : 0;
    return MOVAE_WS_HEADER_BYTES + 2 * (size_t)rows * sizeof(unsigned long long) + items * d * sizeof(float) +
           2 * ((size_t)k + 2) * sizeof(int) + (temp > place ? temp : place) + 1024;
}

int movae_vq_bwd(const float* x, const float* q, const int64_t* idx, const float* dq, const float* gc, const float* ge,
                 float* dx, float* de, int rows, int k, int d, void* ws, size_t ws_bytes, movae_stream_t stream) {
    MOVAE_WS_SCRATCH(ws, ws_bytes);
    MOVAE_CHECK_ARG(x && q && idx, "movae_vq_bwd: null pointer");
    MOVAE_CHECK_ARG(rows > 0 && k > 0 && d > 0, "movae_vq_bwd: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const long total = (long)rows * d;
    if (dx) {
        long gq = (total + 255) / 256;
        if (gq > 4096) gq = 4096;
        hipLaunchKernelGGL(vq_bwd_dx_k, dim3((int)gq), dim3(256), 0, st, x, q, dq, gc, dx, total, 1.f / (float)total);
        MOVAE_CHECK_LAUNCH("vq_bwd_dx");
    }
    if (de && ge) {
        MOVAE_CHECK_ARG(ws && ws_bytes + MOVAE_WS_HEADER_BYTES >= movae_vq_bwd_ws_bytes(rows, k, d), "movae_vq_bwd: workspace too small");
        if (int rc = segmented_code_sum(x, q, idx, ge, de, rows, k, d, ws, st)) return rc;
    } else if (de) {
        if (hipMemsetAsync(de, 0, (size_t)k * d * sizeof(float), st) != hipSuccess) {
            movae_set_error("movae_vq_bwd: memset failed");
            return MOVAE_ELAUNCH;
        }
    }
    return MOVAE_OK;
}

int movae_embedding_bwd(const float* dy, const int64_t* idx, float* dweight, int rows, int k, int d, void* ws, size_t ws_bytes,
                        movae_stream_t stream) {
    MOVAE_WS_SCRATCH(ws, ws_bytes);
    MOVAE_CHECK_ARG(dy && idx && dweight && rows > 0 && k > 0 && d > 0, "movae_embedding_bwd: bad argument");
    MOVAE_CHECK_ARG(ws && ws_bytes + MOVAE_WS_HEADER_BYTES >= movae_vq_bwd_ws_bytes(rows, k, d), "movae_embedding_bwd: workspace too small");
    return segmented_code_sum(nullptr, dy, idx, nullptr, dweight, rows, k, d, ws, (hipStream_t)stream);
}

}  // extern "C"
