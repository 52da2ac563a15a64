// igemm_v2.h -- the fast implicit-GEMM path (included inside conv_igemm.hip's anonymous namespace).
//
// Same three gather forms as the v1 kernels, restricted to 4-float-aligned reduction / output
// channels (every layer of the hot path except the 3-channel image ends), and built for latency:
//   * BK = 32 per stage: 16 v_mfma_f32_32x32x2_f32 per wave tile between barriers (1024 MFMA cycles to
//     cover the next stage's global loads, which are issued before the MFMAs);
//   * every global access is a 16-byte load along the operand's contiguous axis (k for gathered
//     activations and [n][k] weights, n for [k][n] weights / wgrad operands) and goes to LDS with one
//     ds_write_b128 -- no transposing stores;
//   * k-contiguous operands live row-major [row][BK+4] in LDS and are read with ds_read_b128 (4 MFMA
//     operands per read, stride 36 floats => conflict-free 16-lane groups); n-contiguous operands live
//     k-major [BK][cols+4] and are read with conflict-free ds_read_b32.
//   MFMA k assignment: lane half h consumes k = 16h + j at step j (both operands agree, so the
//   contraction is unchanged).
#pragma once
#include <type_traits>

namespace v2 {

#ifndef MOVAE_IGEMM_KB
#define MOVAE_IGEMM_KB 32
#endif
constexpr int BK2 = MOVAE_IGEMM_KB;   // k per stage
constexpr int HK = BK2 / 2;           // k per lane half
constexpr int LDR = BK2 + 4;          // row-major leading dimension (16-byte rows, conflict-free b128 reads)
constexpr int CPR = BK2 / 4;          // 16-byte chunks per row-major row
constexpr int RPP = 256 / CPR;        // rows staged per pass of the 256 threads

template <int BM, int BN>
struct T2 {
    static constexpr int WN = BM == 32 ? 4 : (BN >= 64 ? 2 : 1);  // waves along N (32-row tile: all four side by side)
    static constexpr int WM = 4 / WN;            // waves along M
    static constexpr int TM = BM / (WM * 32);    // 32x32 accumulators per wave along M ...
    static constexpr int TN = BN / (WN * 32);    // ... and along N (2x2 = 64x64 per wave for the 128x128 tile)
    static constexpr int LDKA = BM + 4;          // k-major leading dims
    static constexpr int LDKB = BN + 4;
    static constexpr bool DB = true;             // double-buffered LDS stages (128x128: 74 KiB of the CU's 160 KiB, two blocks resident)
    static_assert(TM >= 1 && TN >= 1, "tile");
};

// A row-major, B row-major (FWD)
template <int TM, int TN, int PART>  // PART 0/1: first / second half of the stage's k range
__device__ __forceinline__ void mma_rr(const float* __restrict__ As, const float* __restrict__ Bs, int a_row, int b_row,
                                       f32x16 (&acc)[TM * TN]) {
    const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int jj = PART * (HK / 8); jj < (PART + 1) * (HK / 8); ++jj) {
        f32x4 a4[TM], b4[TN];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
            b4[tn] = *reinterpret_cast<const f32x4*>(Bs + (b_row + tn * 32 + l31) * LDR + half * HK + jj * 4);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
            a4[tm] = *reinterpret_cast<const f32x4*>(As + (a_row + tm * 32 + l31) * LDR + half * HK + jj * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm * TN + tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[tm][e], b4[tn][e], acc[tm * TN + tn], 0, 0, 0);
    }
}

// A row-major, B k-major (BWD)
template <int TM, int TN, int PART>
__device__ __forceinline__ void mma_rk(const float* __restrict__ As, const float* __restrict__ Bs, int ldb, int a_row,
                                       int b_col, f32x16 (&acc)[TM * TN]) {
    const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int jj = PART * (HK / 8); jj < (PART + 1) * (HK / 8); ++jj) {
        f32x4 a4[TM];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
            a4[tm] = *reinterpret_cast<const f32x4*>(As + (a_row + tm * 32 + l31) * LDR + half * HK + jj * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float b[TN];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) b[tn] = Bs[(half * HK + jj * 4 + e) * ldb + b_col + tn * 32 + l31];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm * TN + tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[tm][e], b[tn], acc[tm * TN + tn], 0, 0, 0);
        }
    }
}

// A k-major, B k-major (WGRAD)
template <int TM, int TN, int PART>
__device__ __forceinline__ void mma_kk(const float* __restrict__ As, const float* __restrict__ Bs, int lda, int ldb,
                                       int a_col, int b_col, f32x16 (&acc)[TM * TN]) {
    const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int j = PART * (HK / 2); j < (PART + 1) * (HK / 2); ++j) {
        float a[TM], b[TN];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) b[tn] = Bs[(half * HK + j) * ldb + b_col + tn * 32 + l31];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) a[tm] = As[(half * HK + j) * lda + a_col + tm * 32 + l31];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm * TN + tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm * TN + tn], 0, 0, 0);
    }
}

// ---- opt-in bf16 operands (movae_set_compute_dtype(1); fp32 stays the default and the parity path) --------------------------
// The SAME kernels with bf16 operand tiles in LDS and v_mfma_f32_32x32x16_bf16 (16x the fp32 matrix rate): global loads, the
// virtual-operand transform, every epilogue and the fp32 accumulators are unchanged -- operands are rounded to bf16 (RNE,
// v_cvt_pk_bf16_f32) on their way into LDS, where EVERY operand lives row-major [row][k] (k-major operands are transposed by
// their 2-byte stores), 32 k = 64 bytes per row + 16 bytes of padding (80-byte rows: conflict-free 16-lane groups for the
// ds_read_b128 that fetches a lane's 8 consecutive k).  Lane (r, h) of a 32x32x16 MFMA holds A[r][8h + j], B[8h + j][r].
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int LDH = BK2 + 8;  // halfs per LDS row
static_assert(BK2 == 32, "the bf16 stage is two 16-deep MFMA steps");

__device__ __forceinline__ void st_row4(__bf16* __restrict__ base, int row, int k4, f32x4 v) {  // 4 consecutive k of one row
    *reinterpret_cast<bf16x4*>(base + row * LDH + k4) = __builtin_convertvector(v, bf16x4);
}
__device__ __forceinline__ void st_col4(__bf16* __restrict__ base, int row4, int k, f32x4 v) {  // one k of 4 consecutive rows
#pragma unroll
    for (int j = 0; j < 4; ++j) base[(row4 + j) * LDH + k] = (__bf16)v[j];
}

template <int TM, int TN, int PART>  // PART 0 / 1: the stage's first / second 16 reduction indices
__device__ __forceinline__ void mma_bf(const __bf16* __restrict__ As, const __bf16* __restrict__ Bs, int a_row, int b_row,
                                       f32x16 (&acc)[TM * TN]) {
    const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
    bf16x8 a[TM], b[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) b[tn] = *reinterpret_cast<const bf16x8*>(Bs + (b_row + tn * 32 + l31) * LDH + 16 * PART + 8 * half);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) a[tm] = *reinterpret_cast<const bf16x8*>(As + (a_row + tm * 32 + l31) * LDH + 16 * PART + 8 * half);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
            acc[tm * TN + tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], b[tn], acc[tm * TN + tn], 0, 0, 0);
}

#define ZERO4 (f32x4{0.f, 0.f, 0.f, 0.f})

// Stage order inside the (single basic block) k loop: first MFMA half | LDS stores of the prefetched stage | address arithmetic and
// buffer loads of the stage after, free to spread under the second MFMA half.  Without the two fences the scheduler hoists the LDS
// stores to the top of the stage and sinks the loads to its end -- eight MFMAs between a load and its use instead of sixty-four.
#ifndef MOVAE_SCHED_PIN_OFF
#define MOVAE_SCHED_PIN() __builtin_amdgcn_sched_barrier(0)
#else
#define MOVAE_SCHED_PIN() ((void)0)
#endif

// Gathers go through raw buffer loads: the descriptor spans 2 GiB from a block-uniform base, a lane whose element is padding /
// past the tile's edge gets the offset BUF_OOB (outside the descriptor: the load returns zeros without touching memory).  No
// branch around any load, 32-bit offsets instead of 64-bit pointers -- the k loop is one basic block the scheduler can spread
// under the MFMAs.  (Host: buf_span_ok -- every valid offset stays below 2 GiB.)
using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr int BUF_OOB = (int)0x80000000;
__device__ __forceinline__ rsrc_t buf_rsrc(const float* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)0x80000000, 0x00020000);
}
__device__ __forceinline__ rsrc_t buf_rsrc_if(const float* p, bool on) {  // on == false: an empty descriptor, every load returns zeros
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, on ? (int)0x80000000 : 0, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(rsrc_t r, int byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
inline bool buf_span_ok(long floats) { return floats * 4 < 0x7fffff00L; }

// ------------------------------------------------------------------------------------------------
// Kernel arguments as plain structs and kernel bodies as device functions of an explicit block index (bx, by, bz) and an
// LDS base: the stand-alone kernels below pass blockIdx, the paired kernel (igemm2_pair) runs a dgrad body and a wgrad
// body side by side in ONE launch, sharing one LDS allocation.
struct FwdArgs {
    const float* X;
    const float* W;
    float* Y;
    Geom g;
    Epilogue ep;
    int M, K, ktiles_per_split;
    float* slab;
    Norm nrm;      // virtual gathered operand (zero-initialised: plain)
    float* stats;  // column partial sums of the stored result [gx * WM][2][N], or null (only without split-K)
    BnBwd bb;      // the result is a fused BatchNorm's output gradient: its backward sums (only without split-K)
    int side_lds;  // host: the side products leave through the LDS tile epilogue -- ONE partial pair per block (else one per wave row)
    FastDiv fd_cr, fd_kw, fd_wlen;  // divisions of the k loop: by g.Cr, g.KW, g.wlen
    ActMul am;     // the stored result is multiplied by the previous layer's activation derivative (only without split-K: else the reduce)
};

template <int BM, int BN>
struct FwdSmem {
    static constexpr int ASZ = BM * LDR, BSZ = BN * LDR;
    static constexpr int FLOATS = (T2<BM, BN>::DB ? 2 : 1) * (ASZ + BSZ);
};

template <int BM, int BN, bool BF = false>
__device__ __forceinline__ void igemm2_fwd_body(const FwdArgs& a, float* __restrict__ smem, int bx, int by, int bz) {
    using T = T2<BM, BN>;
    constexpr int ASZ = BM * LDR, BSZ = BN * LDR;
    float* As = smem;                                   // double buffered when it fits
    float* Bs = smem + (T::DB ? 2 : 1) * ASZ;
    constexpr int ASZH = BM * LDH, BSZH = BN * LDH;    // BF: bf16 tiles (in halfs), both operands [row][k]
    __bf16* AsH = reinterpret_cast<__bf16*>(smem);
    __bf16* BsH = AsH + 2 * ASZH;
    const float* __restrict__ X = a.X;
    const float* __restrict__ W = a.W;
    float* __restrict__ Y = a.Y;
    float* __restrict__ slab = a.slab;
    const Geom g = a.g;
    const Epilogue ep = a.ep;
    const int M = a.M, K = a.K, ktiles_per_split = a.ktiles_per_split;
    const int t = threadIdx.x;
    const int m0 = bx * BM, n0 = by * BN;
    const int N = g.Nn;
    constexpr int AC = BM / RPP, BC = BN / RPP;  // 16-byte chunks per thread
    const int kq = t % CPR, r8 = t / CPR;

    // row i of this thread: byte offset of its (h0, w0) corner from the block's first image, and the corner itself for the
    // padding test (a row past M gets a corner no tap brings back inside)
    const int hw = g.Ho * g.Wo;
    const int img0 = m0 / hw;
    const rsrc_t xr = buf_rsrc(X + (long)img0 * g.Hi * g.Wi * g.Cr);
    int a_off[AC], a_h0[AC], a_w0[AC];
#pragma unroll
    for (int i = 0; i < AC; ++i) {
        const int m = m0 + r8 + RPP * i;
        const bool ok = m < M;
        const int mm = ok ? m : m0;
        const int img = mm / hw, rem = mm - img * hw;
        const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
        const int h0 = ho * g.stride - g.pad, w0 = wo * g.stride - g.pad;
        a_off[i] = ((((img - img0) * g.Hi + h0) * g.Wi + w0) * g.Cr) * 4;
        a_h0[i] = ok ? h0 : -(1 << 20);
        a_w0[i] = w0;
    }
    const int wrow = g.wlen ? g.wrow : K;  // floats per output channel in the stored weights (tap window: Geom)
    const rsrc_t wr = buf_rsrc(W + g.woff);
    int b_off[BC];
#pragma unroll
    for (int i = 0; i < BC; ++i) {
        const int n = n0 + r8 + RPP * i;
        b_off[i] = n < N ? n * wrow * 4 : BUF_OOB;
    }
    const int nk_total = (K + BK2 - 1) / BK2;
    const int kt_begin = bz * ktiles_per_split;
    const int kt_end = min(nk_total, kt_begin + ktiles_per_split);

    f32x4 ra[AC], rb[BC];
    // virtual operand (Norm): the transform is applied in store_tile, i.e. where the loaded values are first consumed -- applied
    // at the load it would put the load's wait in front of the MFMAs that are meant to hide it
    const float* __restrict__ nsc = a.nrm.scale;
    const float* __restrict__ nsh = a.nrm.shift;
    const float nslope = a.nrm.slope;
    f32x4 nsa = ZERO4, nsb = ZERO4;
    unsigned amask = 0;
    f32x16 acc[T::TM * T::TN];
#pragma unroll
    for (int i = 0; i < T::TM * T::TN; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int wave = t >> 6, wm = wave / T::WN, wn = wave % T::WN;
    // The k loop exists twice (PLAIN: no virtual operand, no tap window -- the common case, and free of their uniform branches, so
    // each stage is one basic block; otherwise the general form): unswitched by hand, the compiler does not.
    // KAL (PLAIN and Cr % BK2 == 0, every layer of 32+ channels): a k-stage lies inside ONE tap, so the tap decode is scalar
    // arithmetic on the stage index and a lane's offsets are loop constants plus one scalar -- ~6 VALU per gathered row, 1 per
    // weight row, instead of two divisions and five multiplications per lane and stage.
    auto pipeline = [&](auto plain_c, auto kal_c) {
        constexpr bool PLAIN = decltype(plain_c)::value, KAL = decltype(kal_c)::value;
        int a_offq[AC], b_offq[BC];  // KAL: the lane's own k column folded in
    #pragma unroll
        for (int i = 0; i < AC; ++i) a_offq[i] = a_off[i] + kq * 16;
    #pragma unroll
        for (int i = 0; i < BC; ++i) b_offq[i] = b_off[i] + kq * 16;  // (BUF_OOB + small stays out of range)
        auto load_tile = [&](int kt) {  // (a k-tile at or past kt_end loads nothing: every offset is BUF_OOB)
            if constexpr (KAL) {
                const bool kv = kt < kt_end;  // K % BK2 == 0: a stage is whole or past the end (then: empty descriptors)
                const int k0 = kv ? kt * BK2 : 0;
                const int tap = fdiv(k0, a.fd_cr), c0 = k0 - tap * g.Cr;
                const int kh = fdiv(tap, a.fd_kw), kw = tap - kh * g.KW;
                const int soff = ((kh * g.Wi + kw) * g.Cr + c0) * 4;
                const rsrc_t xs = buf_rsrc_if(X + (long)img0 * g.Hi * g.Wi * g.Cr, kv), ws_ = buf_rsrc_if(W + g.woff, kv);
    #pragma unroll
                for (int i = 0; i < AC; ++i) {
                    const int h = a_h0[i] + kh, w = a_w0[i] + kw;
                    const bool v = (unsigned)h < (unsigned)g.Hi && (unsigned)w < (unsigned)g.Wi;
                    ra[i] = buf_load4(xs, v ? a_offq[i] + soff : BUF_OOB);
                }
    #pragma unroll
                for (int i = 0; i < BC; ++i) rb[i] = buf_load4(ws_, b_offq[i] + k0 * 4);
                return;
            }
            const int k = kt * BK2 + kq * 4;
            const bool kv = k < K && kt < kt_end;
            const int kk = kv ? k : 0;
            const int tap = fdiv(kk, a.fd_cr), c = kk - tap * g.Cr;
            const int kh = fdiv(tap, a.fd_kw), kw = tap - kh * g.KW;
            const int khv = kv ? kh : -(1 << 20);
            const int toff = ((kh * g.Wi + kw) * g.Cr + c) * 4;
            amask = 0;
    #pragma unroll
            for (int i = 0; i < AC; ++i) {
                const int h = a_h0[i] + khv, w = a_w0[i] + kw;
                const bool v = (unsigned)h < (unsigned)g.Hi && (unsigned)w < (unsigned)g.Wi;
                ra[i] = buf_load4(xr, v ? a_off[i] + toff : BUF_OOB);
                amask |= (v ? 1u : 0u) << i;
            }
            if (!PLAIN && nsc) {
                nsa = *reinterpret_cast<const f32x4*>(nsc + c);
                nsb = *reinterpret_cast<const f32x4*>(nsh + c);
            }
            int kb = kk;  // window row -> stored kernel row
            if (!PLAIN && g.wlen) {
                const int wr_ = fdiv(kk, a.fd_wlen);
                kb = wr_ * g.wstride + (kk - wr_ * g.wlen);
            }
    #pragma unroll
            for (int i = 0; i < BC; ++i) rb[i] = buf_load4(wr, kv ? b_off[i] + kb * 4 : BUF_OOB);
        };

        auto store_tile = [&](int buf) {
            if (!PLAIN && nsc) {
    #pragma unroll
                for (int i = 0; i < AC; ++i)
                    ra[i] = norm_apply_if((amask >> i) & 1u, ra[i], nsa, nsb, nslope);
            }
            if constexpr (BF) {
    #pragma unroll
                for (int i = 0; i < AC; ++i) st_row4(AsH + buf * ASZH, r8 + RPP * i, kq * 4, ra[i]);
    #pragma unroll
                for (int i = 0; i < BC; ++i) st_row4(BsH + buf * BSZH, r8 + RPP * i, kq * 4, rb[i]);
                return;
            }
    #pragma unroll
            for (int i = 0; i < AC; ++i) *reinterpret_cast<f32x4*>(As + buf * ASZ + (r8 + RPP * i) * LDR + kq * 4) = ra[i];
    #pragma unroll
            for (int i = 0; i < BC; ++i) *reinterpret_cast<f32x4*>(Bs + buf * BSZ + (r8 + RPP * i) * LDR + kq * 4) = rb[i];
        };
        auto mma = [&](auto part_c, int cur) {
            constexpr int PART = decltype(part_c)::value;
            if constexpr (BF) mma_bf<T::TM, T::TN, PART>(AsH + cur * ASZH, BsH + cur * BSZH, wm * T::TM * 32, wn * T::TN * 32, acc);
            else mma_rr<T::TM, T::TN, PART>(As + cur * ASZ, Bs + cur * BSZ, wm * T::TM * 32, wn * T::TN * 32, acc);
        };
        // software pipeline: stage t is multiplied out of LDS buffer t&1 while the registers of stage t+1 are
        // written to the other buffer between the two MFMA halves and the loads of stage t+2 are issued; one
        // barrier per stage.
        // (store and prefetch are unconditional -- past the end they move zeros / load nothing -- so the loop body is ONE basic block)
        const int nkt = kt_end - kt_begin;
        if (nkt > 0) {
            load_tile(kt_begin);
            store_tile(0);
            __syncthreads();
            load_tile(kt_begin + 1);
        }
        for (int it = 0; it < nkt; ++it) {
            const int cur = it & 1;
            mma(std::integral_constant<int, 0>{}, cur);
            MOVAE_SCHED_PIN();
            store_tile(cur ^ 1);
            MOVAE_SCHED_PIN();
            load_tile(kt_begin + it + 2);
            mma(std::integral_constant<int, 1>{}, cur);
            __syncthreads();
        }
    };
    if (!nsc && !g.wlen) {
        if (g.Cr % BK2 == 0) pipeline(std::true_type{}, std::true_type{});
        else pipeline(std::true_type{}, std::false_type{});
    } else {
        pipeline(std::false_type{}, std::false_type{});
    }

    const int lane = t & 63, half = lane >> 5, l31 = lane & 31;
    const bool to_slab = slab != nullptr;
    float* out = to_slab ? slab + (long)bz * M * N : Y;
    // Fused BatchNorm side products of the stores (never both): `stats` = column sums (v, v^2) of what is stored (the host asks
    // only with act == none); `bb` = the stored value is dout of a fused BatchNorm over bb.y: sums (d, d * y).  One partial pair
    // per wave and column; the two lane halves fold with one shuffle.  (No row block straddles two cotangent groups: host.)
    const bool st_on = a.stats && !to_slab, bb_on = a.bb.y && !to_slab, am_on = actmul_on(a.am) && !to_slab;
    const long pidx = (long)bx * T::WM + wm;  // (per-wave partials: the scalar epilogue, side_lds == 0)
    // the block's first row inside its cotangent group (bb / am: the auxiliary tensor is shared by the groups)
    const int yrow0 = bb_on ? m0 % a.bb.rows_per_group : ((am_on && a.am.y) ? (int)(m0 % (a.am.per_group / N)) : 0);
    // Every epilogue that emits no BatchNorm side product moves its tile through LDS once (free after the loop's last barrier)
    // and then works in 16-byte pieces along n: in accumulator layout a lane holds one COLUMN -- 64 four-byte stores per lane
    // (and as many loads per auxiliary operand of the ActMul form, issued in batches as registers allow).  Measured: plain
    // stores / slab writes -3 % (C3 layers) to -6 % (C2 layers) per call; the ActMul form was 6 % SLOWER than the element-wise
    // pass it replaces before this.  (ActMul: the host guarantees N % 4 == 0 and 16-byte aligned operands.)
    const bool lds_ep = am_on || (!st_on && !bb_on && (N & 3) == 0 &&
                                  ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(ep.bias)) & 15) == 0);
    if (lds_ep) {
        constexpr int LDT = BN + 4, QPR = BN / 4;
        static_assert(BM * LDT <= FwdSmem<BM, BN>::FLOATS, "tile fits the stage buffers");
        float* Ts = smem;
#pragma unroll
        for (int tn = 0; tn < T::TN; ++tn)
#pragma unroll
            for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Ts[(wm * T::TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * LDT + wn * T::TN * 32 + tn * 32 + l31] =
                        acc[tm * T::TN + tn][r];
        __syncthreads();
        const float* __restrict__ yp = a.am.y;
        const float* __restrict__ rp = a.am.res;
        for (int q = t; q < BM * QPR; q += 256) {
            const int row = q / QPR, n = n0 + (q - row * QPR) * 4, m = m0 + row;
            if (m >= M || n >= N) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(Ts + row * LDT + (n - n0));
            if (!to_slab) {
                if (ep.bias) v += *reinterpret_cast<const f32x4*>(ep.bias + n);
                if (am_on) {
                    if (yp) {
                        const f32x4 y4 = *reinterpret_cast<const f32x4*>(yp + (long)(yrow0 + row) * N + n);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] *= act_grad_from_out(y4[j], a.am.act, a.am.slope);
                    }
                    if (rp) v += *reinterpret_cast<const f32x4*>(rp + (long)m * N + n);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], ep.act, ep.slope);
                }
            }
            *reinterpret_cast<f32x4*>(out + (long)m * N + n) = v;
        }
        return;
    }
    if ((st_on || bb_on) && a.side_lds) {
        // the same through-LDS epilogue with a BatchNorm side product: a thread owns one column quad and every (256 / QPR)-th row
        // of the tile, adds up its part of the column sums, the row lanes fold through LDS in fixed order.  ONE partial pair per
        // block, and the host sized the partial layout for that (side_lds: N % 4 == 0, 16-byte aligned operands).
        constexpr int LDT = BN + 4, QPR = BN / 4, RSTEP = 256 / QPR;
        float* Ts = smem;
#pragma unroll
        for (int tn = 0; tn < T::TN; ++tn)
#pragma unroll
            for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Ts[(wm * T::TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * LDT + wn * T::TN * 32 + tn * 32 + l31] =
                        acc[tm * T::TN + tn][r];
        __syncthreads();
        const int cq = t % QPR, rowi = t / QPR, n = n0 + cq * 4;
        f32x4 s1 = f32x4{0.f, 0.f, 0.f, 0.f}, s2 = s1;
        if (n < N) {
            const f32x4 b4 = ep.bias ? *reinterpret_cast<const f32x4*>(ep.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 sc4 = bb_on ? *reinterpret_cast<const f32x4*>(a.bb.scale + n) : b4;
            const f32x4 sh4 = bb_on ? *reinterpret_cast<const f32x4*>(a.bb.shift + n) : b4;
            const float* __restrict__ yp = a.bb.y;
            for (int rb = rowi; rb < BM; rb += 4 * RSTEP) {
                f32x4 y4[4];  // (the y loads of four rows ahead of their stores: see the note on aliasing below)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int row = rb + u * RSTEP;
                    y4[u] = (bb_on && row < BM && m0 + row < M) ? *reinterpret_cast<const f32x4*>(yp + (long)(yrow0 + row) * N + n)
                                                                : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int row = rb + u * RSTEP, m = m0 + row;
                    if (row >= BM || m >= M) continue;
                    f32x4 v = *reinterpret_cast<const f32x4*>(Ts + row * LDT + cq * 4);
                    if (st_on) {
                        v += b4;
                        s1 += v;
#pragma unroll
                        for (int j = 0; j < 4; ++j) s2[j] = fmaf(v[j], v[j], s2[j]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float d = v[j] * (fmaf(y4[u][j], sc4[j], sh4[j]) > 0.f ? 1.f : a.bb.slope);
                            s1[j] += d;
                            s2[j] = fmaf(d, y4[u][j], s2[j]);
                        }
                    }
                    *reinterpret_cast<f32x4*>(out + (long)m * N + n) = v;
                }
            }
        }
        __syncthreads();
        *reinterpret_cast<f32x4*>(Ts + rowi * BN + cq * 4) = s1;
        *reinterpret_cast<f32x4*>(Ts + (RSTEP + rowi) * BN + cq * 4) = s2;
        __syncthreads();
        if (t < BN && n0 + t < N) {
            float c1 = 0.f, c2 = 0.f;
            for (int i = 0; i < RSTEP; ++i) c1 += Ts[i * BN + t], c2 += Ts[(RSTEP + i) * BN + t];
            float* P = st_on ? a.stats : a.bb.part;
            P[((long)bx * 2 + 0) * N + n0 + t] = c1;
            P[((long)bx * 2 + 1) * N + n0 + t] = c2;
        }
        return;
    }
#pragma unroll
    for (int tn = 0; tn < T::TN; ++tn) {
        const int n = n0 + wn * T::TN * 32 + tn * 32 + l31;
        if (n >= N) continue;
        const float bv = (!to_slab && ep.bias) ? ep.bias[n] : 0.f;
        const float sc = bb_on ? a.bb.scale[n] : 0.f, sh = bb_on ? a.bb.shift[n] : 0.f;
        float s1 = 0.f, s2 = 0.f;
        // all of the column's y values are requested BEFORE the first store: `out` and `y` may alias as far as the compiler
        // knows, so a load placed after a store waits for it -- sixteen serial round trips per tile otherwise
        float yv[T::TM * 16];
        if (bb_on) {
            const float* __restrict__ yp = a.bb.y;
#pragma unroll
            for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = wm * T::TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    yv[tm * 16 + r] = (m0 + ml < M) ? yp[(long)(yrow0 + ml) * N + n] : 0.f;
                }
        }
#pragma unroll
        for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = wm * T::TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;  // row inside the block
                const int m = m0 + ml;
                const float v = acc[tm * T::TN + tn][r];
                if (m < M) {
                    const float o = to_slab ? v : apply_act(v + bv, ep.act, ep.slope);
                    out[(long)m * N + n] = o;
                    if (st_on) {
                        s1 += o;
                        s2 = fmaf(o, o, s2);
                    } else if (bb_on) {
                        const float y1 = yv[tm * 16 + r];
                        const float d = v * (fmaf(y1, sc, sh) > 0.f ? 1.f : a.bb.slope);
                        s1 += d;
                        s2 = fmaf(d, y1, s2);
                    }
                }
            }
        if (st_on || bb_on) {
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (half == 0) {
                float* P = st_on ? a.stats : a.bb.part;
                P[(pidx * 2 + 0) * N + n] = s1;
                P[(pidx * 2 + 1) * N + n] = s2;
            }
        }
    }
}

// XCD-aware block -> tile map.  The hardware deals consecutive workgroups to the 8 XCDs in turn (linear id % 8), each with an L2 of
// its own: with tile = block id, the row tiles an XCD works on are every 8th one -- neighbours' halo rows and the column tiles that
// re-read the same activation rows land in other XCDs' L2s, and every XCD streams (nearly) the whole activation (PMC: 541 MB per
// launch of igemm2_bwd<128,128> at C3 for ~75 MB algorithmic).  Here XCD x takes a CONTIGUOUS run of tiles in (row tile, column
// tile) order -- column tiles of one row tile adjacent -- so an activation row is fetched by one XCD.  A bijection on [0, gx * gy).
__device__ __forceinline__ void xcd_tile(int xr, int& bx, int& by) {
    if (!xr) return;
    const int gx = (int)gridDim.x, gy = (int)gridDim.y, T = gx * gy;
    const int lin = bx + gx * by, x = lin & 7, j = lin >> 3;
    const int per = T >> 3, rem = T & 7;
    const int t = x * per + (x < rem ? x : rem) + j;
    bx = t / gy, by = t - bx * gy;
}

template <int BM, int BN, bool BF = false>
__global__ __launch_bounds__(256) void igemm2_fwd(FwdArgs a, RSide sd, int gz, int xr) {
    __shared__ __attribute__((aligned(16))) float smem[FwdSmem<BM, BN>::FLOATS];
    if ((int)blockIdx.z >= gz) {  // a parked weight-gradient reduce rides BEHIND this launch's own blocks (conv_igemm.hip: RSide)
        const int bid = (((int)blockIdx.z - gz) * (int)gridDim.y + (int)blockIdx.y) * (int)gridDim.x + (int)blockIdx.x;
        if (bid < sd.nblk) side_reduce(sd, bid, smem);
        return;
    }
    int bx = blockIdx.x, by = blockIdx.y;
    xcd_tile(xr & 1, bx, by);
    igemm2_fwd_body<BM, BN, BF>(a, smem, bx, by, blockIdx.z);
}

// ------------------------------------------------------------------------------------------------
struct BwdArgs {
    const float* X;
    const float* W;
    float* Y;
    Geom g;
    Epilogue ep;
    ClsSplit scls, kps;
    float* slab;
    long total;
    Norm nrm;
    float* stats;  // [classes * gx * WM][2][N] or null (only without split-K)
    int stats_gx;  // row blocks per class (the launch's grid x) -- the partial index is (class * gx + bx) * WM + wave row
    BnBwd bb;      // backward sums of a fused BatchNorm (only without split-K; ppg = classes * (gx / groups) * slots)
    int side_lds;  // host: side products through the LDS tile epilogue, one partial pair (slot) per block; else WM slots per block
    FastDiv fd_cr;         // k loop: by g.Cr ...
    FastDiv fd_nb[4];      // ... and, per output-parity class, by the class's tap columns nB
    FastDiv fd_hw[4], fd_w[4];  // epilogue, per output-parity class: by Hoc * Woc and by Woc
    ActMul am;             // see FwdArgs
};

template <int BM, int BN>
struct BwdSmem {
    static constexpr int ASZ = BM * LDR, BSZ = BK2 * T2<BM, BN>::LDKB;
    static constexpr int FLOATS = (T2<BM, BN>::DB ? 2 : 1) * (ASZ + BSZ);
};

template <int BM, int BN, bool BF = false>
__device__ __forceinline__ void igemm2_bwd_body(const BwdArgs& a, float* __restrict__ smem, int bx, int by, int bz) {
    using T = T2<BM, BN>;
    const BwdArgs& a_ = a;  // (the tap loops below name a local `a`)
    constexpr int ASZ = BM * LDR, BSZ = BK2 * T::LDKB;
    float* As = smem;
    float* Bs = smem + (T::DB ? 2 : 1) * ASZ;
    constexpr int ASZH = BM * LDH, BSZH = BN * LDH;    // BF: bf16 tiles, both [row][k] (the k-major weights are transposed by their stores)
    __bf16* AsH = reinterpret_cast<__bf16*>(smem);
    __bf16* BsH = AsH + 2 * ASZH;
    const float* __restrict__ X = a.X;
    const float* __restrict__ W = a.W;
    float* __restrict__ Y = a.Y;
    float* __restrict__ slab = a.slab;
    const Geom g = a.g;
    const Epilogue ep = a.ep;
    const ClsSplit& scls = a.scls;  // indexed by a run-time class: stays in the kernel-argument segment (a local copy
    const ClsSplit& kps = a.kps;    // would be an alloca the compiler parks in LDS, 4 KiB per block)
    const long total = a.total;
    const int t = threadIdx.x;
    const int s = g.stride;
    // blockIdx.z enumerates (class, split) pairs; class c owns scls.s[c] consecutive z values and kps.s[c] k-tiles per split
    // (the s*s output-parity classes of a 3x3 stride-2 layer carry 1/2/2/4 taps: one split factor for all would let the
    // 4-tap class set the kernel's duration)
    int cls = 0, split = bz;
    while (cls < s * s - 1 && split >= scls.s[cls]) split -= scls.s[cls++];
    const int ktiles_per_split = kps.s[cls];
    const int ph = cls / s, pw = cls % s;
    const int Hoc = (g.Ho - ph + s - 1) / s, Woc = (g.Wo - pw + s - 1) / s;
    const int M = g.Nimg * Hoc * Woc;
    const int m0 = bx * BM, n0 = by * BN;
    const int N = g.Nn;
    if (m0 >= M) {  // a class with fewer row blocks than the grid: nothing to compute, but its statistics slots must read zero
        if (a.stats && a.slab == nullptr) {
            const int wv = t >> 6, wm_ = wv / T2<BM, BN>::WN, wn_ = wv % T2<BM, BN>::WN, l = t & 63;
            const long pidx = ((long)cls * a.stats_gx + bx) * T2<BM, BN>::WM + wm_;
            if (l < 32)
                for (int tn = 0; tn < T2<BM, BN>::TN; ++tn) {
                    const int n = n0 + wn_ * T2<BM, BN>::TN * 32 + tn * 32 + l;
                    if (n < N) a.stats[(pidx * 2 + 0) * N + n] = 0.f, a.stats[(pidx * 2 + 1) * N + n] = 0.f;
                }
        }
        return;
    }
    const int kh0 = (ph + g.pad) % s, kw0 = (pw + g.pad) % s;
    const int nA = kh0 < g.KH ? (g.KH - kh0 + s - 1) / s : 0;
    const int nB = kw0 < g.KW ? (g.KW - kw0 + s - 1) / s : 0;
    const int qh = (ph + g.pad - kh0) / s, qw = (pw + g.pad - kw0) / s;
    const int K = nA * nB * g.Cr;
    const int taps = g.KH * g.KW;
    const int nBd = nB > 0 ? nB : 1;

    constexpr int AC = BM / RPP;
    const int kq = t % CPR, r8 = t / CPR;
    // gathered rows through a buffer descriptor based at the block's first image (see buf_rsrc)
    const int hwc = Hoc * Woc;
    const int img0 = m0 / hwc;
    const rsrc_t xr = buf_rsrc(X + (long)img0 * g.Hi * g.Wi * g.Cr);
    int a_off[AC], a_h0[AC], a_w0[AC];
#pragma unroll
    for (int i = 0; i < AC; ++i) {
        const int m = m0 + r8 + RPP * i;
        const bool ok = m < M;
        const int mm = ok ? m : m0;
        const int img = mm / hwc, rem = mm - img * hwc;
        const int hc = rem / Woc, wc = rem - hc * Woc;
        const int h0 = hc + qh, w0 = wc + qw;
        a_off[i] = ((((img - img0) * g.Hi + h0) * g.Wi + w0) * g.Cr) * 4;
        a_h0[i] = ok ? h0 : -(1 << 20);
        a_w0[i] = w0;
    }
    constexpr int BQ = BN / 4;              // 16-byte chunks per k-row
    constexpr int BC = BK2 * BQ / 256;      // chunks per thread (2 for BN=64, 1 for BN=32)
    constexpr int KSTEP = 256 / BQ;         // k-rows covered per pass
    const int bq = t % BQ, bk = t / BQ;
    const int bn = n0 + bq * 4;
    const rsrc_t wr = buf_rsrc(W);
    const int b_col = bn < N ? bn * 4 : BUF_OOB;  // N % 4 == 0 on this path
    const FastDiv fd_nb = a.fd_nb[cls];

    f32x4 ra[AC], rb[BC];
    const float* __restrict__ nsc = a.nrm.scale;  // virtual operand: see igemm2_fwd_body
    const float* __restrict__ nsh = a.nrm.shift;
    const float nslope = a.nrm.slope;
    f32x4 nsa = ZERO4, nsb = ZERO4;
    unsigned amask = 0;
    const int nk_total = (K + BK2 - 1) / BK2;
    const int kt_begin = split * ktiles_per_split;
    const int kt_end = min(nk_total, kt_begin + ktiles_per_split);
    f32x16 acc[T::TM * T::TN];
#pragma unroll
    for (int i = 0; i < T::TM * T::TN; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int wave = t >> 6, wm = wave / T::WN, wn = wave % T::WN;
    auto pipeline = [&](auto plain_c, auto kal_c) {  // (the k loop: general / plain / plain and k-aligned -- see igemm2_fwd_body)
        constexpr bool PLAIN = decltype(plain_c)::value, KAL = decltype(kal_c)::value;
        int a_offq[AC], b_thr[BC];  // KAL: loop-constant parts of the lane's offsets
    #pragma unroll
        for (int i = 0; i < AC; ++i) a_offq[i] = a_off[i] + kq * 16;
    #pragma unroll
        for (int i = 0; i < BC; ++i) b_thr[i] = (bk + KSTEP * i) * taps * N * 4 + b_col;  // (BUF_OOB + a valid offset stays out of range)
        auto load_tile = [&](int kt) {  // (a k-tile at or past kt_end loads nothing)
            if constexpr (KAL) {
                const bool kv = kt < kt_end;
                const int k0 = kv ? kt * BK2 : 0;
                const int tt = fdiv(k0, a_.fd_cr), c0 = k0 - tt * g.Cr;
                const int ta = fdiv(tt, fd_nb), tb = tt - ta * nBd;
                const int soff = (c0 - (ta * g.Wi + tb) * g.Cr) * 4;
                const int kh = kh0 + s * ta, kw = kw0 + s * tb;
                const int wbase = ((c0 * taps + kh * g.KW + kw) * N) * 4;
                const rsrc_t xs = buf_rsrc_if(X + (long)img0 * g.Hi * g.Wi * g.Cr, kv), ws_ = buf_rsrc_if(W, kv);
    #pragma unroll
                for (int i = 0; i < AC; ++i) {
                    const int h = a_h0[i] - ta, w = a_w0[i] - tb;
                    const bool v = (unsigned)h < (unsigned)g.Hi && (unsigned)w < (unsigned)g.Wi;
                    ra[i] = buf_load4(xs, v ? a_offq[i] + soff : BUF_OOB);
                }
    #pragma unroll
                for (int i = 0; i < BC; ++i) rb[i] = buf_load4(ws_, b_thr[i] + wbase);
                return;
            }
            {
                const int k = kt * BK2 + kq * 4;
                const bool kv = k < K && kt < kt_end;
                const int kk = kv ? k : 0;
                const int tt = fdiv(kk, a_.fd_cr), c = kk - tt * g.Cr;
                const int ta = fdiv(tt, fd_nb), tb = tt - ta * nBd;
                const int tav = kv ? ta : (1 << 20);
                const int toff = (c - (ta * g.Wi + tb) * g.Cr) * 4;
                amask = 0;
    #pragma unroll
                for (int i = 0; i < AC; ++i) {
                    const int h = a_h0[i] - tav, w = a_w0[i] - tb;
                    const bool v = (unsigned)h < (unsigned)g.Hi && (unsigned)w < (unsigned)g.Wi;
                    ra[i] = buf_load4(xr, v ? a_off[i] + toff : BUF_OOB);
                    amask |= (v ? 1u : 0u) << i;
                }
                if (!PLAIN && nsc) {
                    nsa = *reinterpret_cast<const f32x4*>(nsc + c);
                    nsb = *reinterpret_cast<const f32x4*>(nsh + c);
                }
            }
    #pragma unroll
            for (int i = 0; i < BC; ++i) {
                const int k = kt * BK2 + bk + KSTEP * i;
                const bool kv = k < K && kt < kt_end;
                const int kk = kv ? k : 0;
                const int tt = fdiv(kk, a_.fd_cr), c = kk - tt * g.Cr;
                const int ta = fdiv(tt, fd_nb), tb = tt - ta * nBd;
                const int kh = kh0 + s * ta, kw = kw0 + s * tb;
                rb[i] = buf_load4(wr, kv ? ((c * taps + kh * g.KW + kw) * N) * 4 + b_col : BUF_OOB);
            }
        };

        auto store_tile = [&](int buf) {
            if (!PLAIN && nsc) {
    #pragma unroll
                for (int i = 0; i < AC; ++i)
                    ra[i] = norm_apply_if((amask >> i) & 1u, ra[i], nsa, nsb, nslope);
            }
            if constexpr (BF) {
    #pragma unroll
                for (int i = 0; i < AC; ++i) st_row4(AsH + buf * ASZH, r8 + RPP * i, kq * 4, ra[i]);
    #pragma unroll
                for (int i = 0; i < BC; ++i) st_col4(BsH + buf * BSZH, bq * 4, bk + KSTEP * i, rb[i]);
                return;
            }
    #pragma unroll
            for (int i = 0; i < AC; ++i) *reinterpret_cast<f32x4*>(As + buf * ASZ + (r8 + RPP * i) * LDR + kq * 4) = ra[i];
    #pragma unroll
            for (int i = 0; i < BC; ++i) *reinterpret_cast<f32x4*>(Bs + buf * BSZ + (bk + KSTEP * i) * T::LDKB + bq * 4) = rb[i];
        };
        auto mma = [&](auto part_c, int cur) {
            constexpr int PART = decltype(part_c)::value;
            if constexpr (BF) mma_bf<T::TM, T::TN, PART>(AsH + cur * ASZH, BsH + cur * BSZH, wm * T::TM * 32, wn * T::TN * 32, acc);
            else mma_rk<T::TM, T::TN, PART>(As + cur * ASZ, Bs + cur * BSZ, T::LDKB, wm * T::TM * 32, wn * T::TN * 32, acc);
        };
        const int nkt = kt_end - kt_begin;  // (one basic block per stage: see igemm2_fwd_body)
        if (nkt > 0) {
            load_tile(kt_begin);
            store_tile(0);
            __syncthreads();
            load_tile(kt_begin + 1);
        }
        for (int it = 0; it < nkt; ++it) {
            const int cur = it & 1;
            mma(std::integral_constant<int, 0>{}, cur);
            MOVAE_SCHED_PIN();
            store_tile(cur ^ 1);
            MOVAE_SCHED_PIN();
            load_tile(kt_begin + it + 2);
            MOVAE_SCHED_PIN();  // (measured: the gather issued BEFORE the second half beats spreading it under it, this form only)
            mma(std::integral_constant<int, 1>{}, cur);
            __syncthreads();
        }
    };
    if (!nsc) {
        if (g.Cr % BK2 == 0) pipeline(std::true_type{}, std::true_type{});
        else pipeline(std::true_type{}, std::false_type{});
    } else {
        pipeline(std::false_type{}, std::false_type{});
    }

    const int lane = t & 63, half = lane >> 5, l31 = lane & 31;
    const bool to_slab = slab != nullptr;
    float* out = to_slab ? slab + (long)split * total : Y;
    // fused BatchNorm side products of the stores (see igemm2_fwd_body); the output pixel p is computed once for both
    const bool st_on = a.stats && !to_slab, bb_on = a.bb.y && !to_slab, am_on = actmul_on(a.am) && !to_slab;
    const int SL = a.side_lds ? 1 : T::WM, sw = a.side_lds ? 0 : wm;  // partial slots per block, this wave's slot
    long pidx = ((long)cls * a.stats_gx + bx) * SL + sw;
    long ybase = 0;  // first pixel of the block's cotangent group
    if (am_on && a.am.y && a.am.gx_per_group > 0) ybase = (long)(bx / a.am.gx_per_group) * (a.am.per_group / N);
    if (bb_on) {
        // classes are equally large and no row block straddles two cotangent groups (host): group gi owns bpg row blocks of
        // every class; its partials are [gi * ppg, (gi + 1) * ppg), ordered (class, block in group, wave row)
        const int bpg = a.bb.ppg / (s * s * SL), gi = bx / bpg;
        pidx = (long)gi * a.bb.ppg + ((long)cls * bpg + (bx - gi * bpg)) * SL + sw;
        ybase = (long)gi * a.bb.rows_per_group;
    }
    const bool lds_ep = am_on || (!st_on && !bb_on && (N & 3) == 0 &&
                                  ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(ep.bias)) & 15) == 0);
    if (lds_ep) {  // epilogue through LDS in 16-byte pieces: see igemm2_fwd_body (rows are the class's pixels here)
        constexpr int LDT = BN + 4, QPR = BN / 4;
        static_assert(BM * LDT <= BwdSmem<BM, BN>::FLOATS, "tile fits the stage buffers");
        float* Ts = smem;
#pragma unroll
        for (int tn = 0; tn < T::TN; ++tn)
#pragma unroll
            for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Ts[(wm * T::TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * LDT + wn * T::TN * 32 + tn * 32 + l31] =
                        acc[tm * T::TN + tn][r];
        __syncthreads();
        const float* __restrict__ yp = a.am.y;
        const float* __restrict__ rp = a.am.res;
        const int hwc = Hoc * Woc;
        for (int q = t; q < BM * QPR; q += 256) {
            const int row = q / QPR, n = n0 + (q - row * QPR) * 4, m = m0 + row;
            if (m >= M || n >= N) continue;
            const int img = fdiv(m, a.fd_hw[cls]), rem = m - img * hwc;
            const int hc = fdiv(rem, a.fd_w[cls]), wc = rem - hc * Woc;
            const long p = (long)(img * g.Ho + (hc * s + ph)) * g.Wo + (wc * s + pw);
            f32x4 v = *reinterpret_cast<const f32x4*>(Ts + row * LDT + (n - n0));
            if (!to_slab) {
                if (ep.bias) v += *reinterpret_cast<const f32x4*>(ep.bias + n);
                if (am_on) {
                    if (yp) {
                        const f32x4 y4 = *reinterpret_cast<const f32x4*>(yp + (p - ybase) * N + n);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] *= act_grad_from_out(y4[j], a.am.act, a.am.slope);
                    }
                    if (rp) v += *reinterpret_cast<const f32x4*>(rp + p * N + n);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], ep.act, ep.slope);
                }
            }
            *reinterpret_cast<f32x4*>(out + p * N + n) = v;
        }
        return;
    }
    if ((st_on || bb_on) && a.side_lds) {
        // through-LDS epilogue with a BatchNorm side product: see igemm2_fwd_body (one partial pair per block)
        constexpr int LDT = BN + 4, QPR = BN / 4, RSTEP = 256 / QPR;
        float* Ts = smem;
#pragma unroll
        for (int tn = 0; tn < T::TN; ++tn)
#pragma unroll
            for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Ts[(wm * T::TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * LDT + wn * T::TN * 32 + tn * 32 + l31] =
                        acc[tm * T::TN + tn][r];
        __syncthreads();
        const int cq = t % QPR, rowi = t / QPR, n = n0 + cq * 4;
        const int hwc = Hoc * Woc;
        f32x4 s1 = f32x4{0.f, 0.f, 0.f, 0.f}, s2 = s1;
        if (n < N) {
            const f32x4 b4 = ep.bias ? *reinterpret_cast<const f32x4*>(ep.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 sc4 = bb_on ? *reinterpret_cast<const f32x4*>(a.bb.scale + n) : b4;
            const f32x4 sh4 = bb_on ? *reinterpret_cast<const f32x4*>(a.bb.shift + n) : b4;
            const float* __restrict__ yp = a.bb.y;
            for (int rb = rowi; rb < BM; rb += 4 * RSTEP) {
                f32x4 y4[4];
                long px[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int row = rb + u * RSTEP, m = m0 + row;
                    px[u] = -1;
                    y4[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (row < BM && m < M) {
                        const int img = fdiv(m, a.fd_hw[cls]), rem = m - img * hwc;
                        const int hc = fdiv(rem, a.fd_w[cls]), wc = rem - hc * Woc;
                        px[u] = (long)(img * g.Ho + (hc * s + ph)) * g.Wo + (wc * s + pw);
                        if (bb_on) y4[u] = *reinterpret_cast<const f32x4*>(yp + (px[u] - ybase) * N + n);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (px[u] < 0) continue;
                    f32x4 v = *reinterpret_cast<const f32x4*>(Ts + (rb + u * RSTEP) * LDT + cq * 4);
                    if (st_on) {
                        v += b4;
                        s1 += v;
#pragma unroll
                        for (int j = 0; j < 4; ++j) s2[j] = fmaf(v[j], v[j], s2[j]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float d = v[j] * (fmaf(y4[u][j], sc4[j], sh4[j]) > 0.f ? 1.f : a.bb.slope);
                            s1[j] += d;
                            s2[j] = fmaf(d, y4[u][j], s2[j]);
                        }
                    }
                    *reinterpret_cast<f32x4*>(out + px[u] * N + n) = v;
                }
            }
        }
        __syncthreads();
        *reinterpret_cast<f32x4*>(Ts + rowi * BN + cq * 4) = s1;
        *reinterpret_cast<f32x4*>(Ts + (RSTEP + rowi) * BN + cq * 4) = s2;
        __syncthreads();
        if (t < BN && n0 + t < N) {
            float c1 = 0.f, c2 = 0.f;
            for (int i = 0; i < RSTEP; ++i) c1 += Ts[i * BN + t], c2 += Ts[(RSTEP + i) * BN + t];
            float* P = st_on ? a.stats : a.bb.part;
            P[(pidx * 2 + 0) * N + n0 + t] = c1;  // (side_lds: pidx is the block's one slot)
            P[(pidx * 2 + 1) * N + n0 + t] = c2;
        }
        return;
    }
    float sc[T::TN], sh[T::TN], s1[T::TN], s2[T::TN];
#pragma unroll
    for (int tn = 0; tn < T::TN; ++tn) {
        const int n = n0 + wn * T::TN * 32 + tn * 32 + l31;
        sc[tn] = (bb_on && n < N) ? a.bb.scale[n] : 0.f;
        sh[tn] = (bb_on && n < N) ? a.bb.shift[n] : 0.f;
        s1[tn] = s2[tn] = 0.f;
    }
    // output pixel of every accumulator row, computed once (three integer divisions each); -1 = past the class's rows
    int pix[T::TM * 16];
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * T::TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int hw = Hoc * Woc;
            const int img = fdiv(m, a.fd_hw[cls]), rem = m - img * hw;
            const int hc = fdiv(rem, a.fd_w[cls]), wc = rem - hc * Woc;
            pix[tm * 16 + r] = m < M ? (img * g.Ho + (hc * s + ph)) * g.Wo + (wc * s + pw) : -1;
        }
    // BnBwd: every y value is requested before the first store (see igemm2_fwd_body)
    float yv[T::TM * 16 * T::TN];
    if (bb_on) {
        const float* __restrict__ yp = a.bb.y;
#pragma unroll
        for (int i = 0; i < T::TM * 16; ++i)
#pragma unroll
            for (int tn = 0; tn < T::TN; ++tn) {
                const int n = n0 + wn * T::TN * 32 + tn * 32 + l31;
                yv[i * T::TN + tn] = (pix[i] >= 0 && n < N) ? yp[((long)pix[i] - ybase) * N + n] : 0.f;
            }
    }
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pi = pix[tm * 16 + r];
            if (pi < 0) continue;
            const long p = pi;
#pragma unroll
            for (int tn = 0; tn < T::TN; ++tn) {
                const int n = n0 + wn * T::TN * 32 + tn * 32 + l31;
                if (n >= N) continue;
                const float bv = (!to_slab && ep.bias) ? ep.bias[n] : 0.f;
                const float v = acc[tm * T::TN + tn][r];
                const float o = to_slab ? v : apply_act(v + bv, ep.act, ep.slope);
                out[p * N + n] = o;
                if (st_on) {
                    s1[tn] += o;
                    s2[tn] = fmaf(o, o, s2[tn]);
                } else if (bb_on) {
                    const float y1 = yv[(tm * 16 + r) * T::TN + tn];
                    const float d = v * (fmaf(y1, sc[tn], sh[tn]) > 0.f ? 1.f : a.bb.slope);
                    s1[tn] += d;
                    s2[tn] = fmaf(d, y1, s2[tn]);
                }
            }
        }
    if (st_on || bb_on) {
        float* P = st_on ? a.stats : a.bb.part;
#pragma unroll
        for (int tn = 0; tn < T::TN; ++tn) {
            const int n = n0 + wn * T::TN * 32 + tn * 32 + l31;
            const float t1 = s1[tn] + __shfl_xor(s1[tn], 32, 64), t2 = s2[tn] + __shfl_xor(s2[tn], 32, 64);
            if (half == 0 && n < N) {
                P[(pidx * 2 + 0) * N + n] = t1;
                P[(pidx * 2 + 1) * N + n] = t2;
            }
        }
    }
}

template <int BM, int BN, bool BF = false>
__global__ __launch_bounds__(256) void igemm2_bwd(BwdArgs a, RSide sd, int gz, int xr) {
    __shared__ __attribute__((aligned(16))) float smem[BwdSmem<BM, BN>::FLOATS];
    if ((int)blockIdx.z >= gz) {  // a parked weight-gradient reduce rides BEHIND this launch's own blocks (conv_igemm.hip: RSide)
        const int bid = (((int)blockIdx.z - gz) * (int)gridDim.y + (int)blockIdx.y) * (int)gridDim.x + (int)blockIdx.x;
        if (bid < sd.nblk) side_reduce(sd, bid, smem);
        return;
    }
    int bx = blockIdx.x, by = blockIdx.y;
    xcd_tile(xr & 1, bx, by);
    // (xr & 2: the (class, split) pairs in REVERSE order -- the last output-parity class of a 3x3 stride-2 layer carries 4 of the 9
    // taps, the first one 1: blocks are dispatched in index order, and the long ones should not be the tail; cls_order())
    igemm2_bwd_body<BM, BN, BF>(a, smem, bx, by, (xr & 2) ? gz - 1 - (int)blockIdx.z : (int)blockIdx.z);
}

// ------------------------------------------------------------------------------------------------
// Cotangent groups (batched pull-back): blockIdx.z = group * Sp + split.  A group adds s_gs / b_gs floats to the
// operand bases (0 for the operand the groups share) and owns its own Sp slabs / its own destination.
struct WOut {
    float* p[8];
};

struct WgArgs {
    const float* Sm;
    const float* Bg;
    float* out;
    WGeom g;
    int K, kchunk, to_slab, Sp;
    long s_gs, b_gs;
    WOut tab;
    Norm nrm;      // virtual activation operand ...
    int nrm_side;  // ... 1: the small-side operand Sm (transposed-conv x), 2: the gathered operand Bg (conv x); 0: none
    FastDiv fd_hw, fd_ws;  // k loop (pixel -> image, row, column): by g.Hs * g.Ws and by g.Ws
    // Column sums of the small-side operand, sum_k Sm[k][m] (the bias gradient when Sm is dy), from the by == 0 blocks, which stage
    // every Sm value of their split once anyway: cs.p[group] (or null) are the destinations; with slabs the partial goes behind the
    // slab's M * N weight-gradient floats (slab_stride = M * N + M) and the split-K reduce folds it like the rest.
    WOut cs;
    int cs_on;
    long slab_stride;
};

template <int BM, int BN>
struct WgSmem {
    static constexpr int ASZ = BK2 * T2<BM, BN>::LDKA, BSZ = BK2 * T2<BM, BN>::LDKB;
    static constexpr int FLOATS = (T2<BM, BN>::DB ? 2 : 1) * (ASZ + BSZ);
};

template <int BM, int BN, bool BF = false>
__device__ __forceinline__ void igemm2_wgrad_body(const WgArgs& a, float* __restrict__ smem, int bx, int by, int bz) {
    using T = T2<BM, BN>;
    constexpr int ASZ = BK2 * T::LDKA, BSZ = BK2 * T::LDKB;
    float* As = smem;
    float* Bs = smem + (T::DB ? 2 : 1) * ASZ;
    constexpr int ASZH = BM * LDH, BSZH = BN * LDH;    // BF: bf16 tiles [row][k] -- both operands are k-major in memory: transposing stores
    __bf16* AsH = reinterpret_cast<__bf16*>(smem);
    __bf16* BsH = AsH + 2 * ASZH;
    const WGeom g = a.g;
    const int K = a.K, kchunk = a.kchunk, to_slab = a.to_slab, Sp = a.Sp;
    float* __restrict__ out = a.out;
    const int t = threadIdx.x;
    const int M = g.Cs, N = g.KH * g.KW * g.Cb;
    const int m0 = bx * BM, n0 = by * BN;
    const int grp = bz / Sp, split = bz - grp * Sp;
    // both operands through buffer descriptors based at the group's tensors (see buf_rsrc; host: whole tensors below 2 GiB)
    const rsrc_t sr = buf_rsrc(a.Sm + grp * a.s_gs);
    const rsrc_t br = buf_rsrc(a.Bg + grp * a.b_gs);
    const int k_begin = split * kchunk, k_end = min(K, k_begin + kchunk);
    if (g.Hs == 1 && g.Ws == 1) {
        // a 1x1 small side meets tap (kh, kw) at big-side pixel (kh - pad, kw - pad) only: column tiles whose taps all fall
        // outside the image are identically zero -- store the zeros, skip the reduction (5 of 9 taps for 2x2 <-> 1x1, k3 p1)
        const int t0 = n0 / g.Cb, t1 = (min(n0 + BN, N) - 1) / g.Cb;
        bool any = false;
        for (int tp = t0; tp <= t1; ++tp) {
            const int h = tp / g.KW - g.pad, w = tp % g.KW - g.pad;
            any = any || (h >= 0 && h < g.Hb && w >= 0 && w < g.Wb);
        }
        if (!any && !(a.cs_on && by == 0)) {
            float* dst0 = to_slab ? out + (long)bz * a.slab_stride : a.tab.p[grp];
            for (int i = t; i < BM * BN; i += 256) {  // scalar stores: a Jacobian-row destination is only 4-byte aligned
                const int m = m0 + i / BN, n = n0 + i % BN;
                if (m < M && n < N) dst0[(long)m * N + n] = 0.f;
            }
            return;
        }
    }
    constexpr int AQ = BM / 4, BQ = BN / 4;
    constexpr int ACH = BK2 * AQ / 256, BCH = BK2 * BQ / 256;
    constexpr int AKS = 256 / AQ, BKS = 256 / BQ;
    const int aq = t % AQ, ak = t / AQ;
    const int bq = t % BQ, bk = t / BQ;
    const int am = m0 + aq * 4;
    const bool am_ok = am < M;  // M % 4 == 0 on this path
    const int bn = n0 + bq * 4;
    const bool bn_ok = bn < N;  // Cb % 4 == 0 => the 4 columns share one tap
    const int b_tap = (bn_ok ? bn : 0) / g.Cb, b_c = (bn_ok ? bn : 0) - b_tap * g.Cb;
    const int b_kh = b_tap / g.KW, b_kw = b_tap - b_kh * g.KW;
    const int hw = g.Hs * g.Ws;
    const int a_col = am_ok ? am * 4 : BUF_OOB;
    // big-side element of small-side pixel (img, hs, ws) under this thread's tap: byte offset
    // img * s_img + hs * s_h + ws * s_w + b_const; rows / columns tested against the padding as unsigned compares
    const int s_img = g.Hb * g.Wb * g.Cb * 4, s_h = g.stride * g.Wb * g.Cb * 4, s_w = g.stride * g.Cb * 4;
    const int b_h0 = bn_ok ? b_kh - g.pad : -(1 << 20), b_w0 = b_kw - g.pad;
    const int b_const = (((b_kh - g.pad) * g.Wb + (b_kw - g.pad)) * g.Cb + b_c) * 4;

    f32x4 ra[ACH], rb[BCH];
    // virtual activation operand (Norm): this thread's four channels are the same for every k, so scale / shift are loaded once;
    // applied in store_tile (see igemm2_fwd_body)
    const int nside = a.nrm.scale ? a.nrm_side : 0;
    const float nslope = a.nrm.slope;
    f32x4 nsa = ZERO4, nsb = ZERO4;
    if (nside == 1 && am_ok) {
        nsa = *reinterpret_cast<const f32x4*>(a.nrm.scale + am);
        nsb = *reinterpret_cast<const f32x4*>(a.nrm.shift + am);
    } else if (nside == 2 && bn_ok) {
        nsa = *reinterpret_cast<const f32x4*>(a.nrm.scale + b_c);
        nsb = *reinterpret_cast<const f32x4*>(a.nrm.shift + b_c);
    }
    unsigned vmask = 0;
    f32x16 acc[T::TM * T::TN];
#pragma unroll
    for (int i = 0; i < T::TM * T::TN; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int wave = t >> 6, wm = wave / T::WN, wn = wave % T::WN;
    f32x4 csum = ZERO4;  // CS: this lane's share of the column sums of Sm (its 4 columns, its k rows)
    auto pipeline = [&](auto plain_c, auto cs_c) {  // (the k loop with and without the virtual operand: see igemm2_fwd_body)
        constexpr bool PLAIN = decltype(plain_c)::value, CS = decltype(cs_c)::value;
        // PLAIN stages the gathered operand one PIXEL per lane and stage (8 lanes share a pixel, each with BCH column quads, i.e.
        // BCH taps / channel groups that are loop constants): the pixel -> (image, row, column) decode -- two divisions, three
        // multiplications -- happens once per lane and stage instead of once per 16-byte chunk.
        constexpr int PL = 8;                       // lanes per pixel row
        const int pq = t % PL, pk = t / PL;         // column-quad phase, pixel row of the stage (256 / 8 = BK2 rows)
        static_assert(256 / PL == BK2 && BQ % PL == 0 && BQ / PL == BCH, "one pixel row per lane");
        int p_h0[BCH], p_w0[BCH], p_const[BCH];
        int a_thr[ACH];
        if constexpr (PLAIN) {
    #pragma unroll
            for (int i = 0; i < BCH; ++i) {
                const int n = n0 + (pq + PL * i) * 4;
                const bool ok = n < N;
                const int tap = (ok ? n : 0) / g.Cb, c = (ok ? n : 0) - tap * g.Cb;
                const int kh = tap / g.KW, kw = tap - kh * g.KW;
                p_h0[i] = ok ? kh - g.pad : -(1 << 20);
                p_w0[i] = kw - g.pad;
                p_const[i] = (((kh - g.pad) * g.Wb + (kw - g.pad)) * g.Cb + c) * 4;
            }
    #pragma unroll
            for (int i = 0; i < ACH; ++i) a_thr[i] = (ak + AKS * i) * (M * 4) + a_col;  // (BUF_OOB + a valid offset stays out of range)
        }
        auto load_tile = [&](int k0) {  // (a stage at or past k_end loads nothing)
            if constexpr (PLAIN) {
                const int kM = k0 * (M * 4);
    #pragma unroll
                for (int i = 0; i < ACH; ++i) ra[i] = buf_load4(sr, k0 + ak + AKS * i < k_end ? a_thr[i] + kM : BUF_OOB);
                const int k = k0 + pk;
                const bool kv = k < k_end;
                const int kk = kv ? k : k_begin;
                const int img = fdiv(kk, a.fd_hw), rem = kk - img * hw;
                const int hs = fdiv(rem, a.fd_ws), ws = rem - hs * g.Ws;
                const int hS = kv ? hs * g.stride : -(1 << 20), wS = ws * g.stride;
                const int base = img * s_img + hs * s_h + ws * s_w;
    #pragma unroll
                for (int i = 0; i < BCH; ++i) {
                    const int h = hS + p_h0[i], w = wS + p_w0[i];
                    const bool v = (unsigned)h < (unsigned)g.Hb && (unsigned)w < (unsigned)g.Wb;
                    rb[i] = buf_load4(br, v ? base + p_const[i] : BUF_OOB);
                }
                return;
            }
            unsigned va = 0, vb = 0;
    #pragma unroll
            for (int i = 0; i < ACH; ++i) {
                const int k = k0 + ak + AKS * i;
                const bool v = k < k_end;
                ra[i] = buf_load4(sr, v ? k * (M * 4) + a_col : BUF_OOB);
                va |= ((v && am_ok) ? 1u : 0u) << i;
            }
    #pragma unroll
            for (int i = 0; i < BCH; ++i) {
                const int k = k0 + bk + BKS * i;
                const bool kv = k < k_end;
                const int kk = kv ? k : k_begin;
                const int img = fdiv(kk, a.fd_hw), rem = kk - img * hw;
                const int hs = fdiv(rem, a.fd_ws), ws = rem - hs * g.Ws;
                const int h = hs * g.stride + b_h0, w = ws * g.stride + b_w0;
                const bool v = kv && (unsigned)h < (unsigned)g.Hb && (unsigned)w < (unsigned)g.Wb;
                rb[i] = buf_load4(br, v ? img * s_img + hs * s_h + ws * s_w + b_const : BUF_OOB);
                vb |= (v ? 1u : 0u) << i;
            }
            if (!PLAIN) vmask = nside == 1 ? va : vb;
        };

        auto store_tile = [&](int buf) {
            // (two separate ifs with a compiler barrier between them: as an if / else over equally long arrays the two arms are merged
            // into one loop over a run-time selected array, which moves ra / rb to scratch memory)
            if (!PLAIN && nside == 1) {
    #pragma unroll
                for (int i = 0; i < ACH; ++i) ra[i] = norm_apply_if((vmask >> i) & 1u, ra[i], nsa, nsb, nslope);
            }
            asm volatile("");
            if (!PLAIN && nside == 2) {
    #pragma unroll
                for (int i = 0; i < BCH; ++i) rb[i] = norm_apply_if((vmask >> i) & 1u, rb[i], nsa, nsb, nslope);
            }
            if constexpr (CS) {  // (rows past k_end and columns past M were loaded as zeros)
    #pragma unroll
                for (int i = 0; i < ACH; ++i) csum += ra[i];
            }
            if constexpr (BF) {
    #pragma unroll
                for (int i = 0; i < ACH; ++i) st_col4(AsH + buf * ASZH, aq * 4, ak + AKS * i, ra[i]);
                if constexpr (PLAIN) {
    #pragma unroll
                    for (int i = 0; i < BCH; ++i) st_col4(BsH + buf * BSZH, (pq + PL * i) * 4, pk, rb[i]);
                } else {
    #pragma unroll
                    for (int i = 0; i < BCH; ++i) st_col4(BsH + buf * BSZH, bq * 4, bk + BKS * i, rb[i]);
                }
                return;
            }
    #pragma unroll
            for (int i = 0; i < ACH; ++i) *reinterpret_cast<f32x4*>(As + buf * ASZ + (ak + AKS * i) * T::LDKA + aq * 4) = ra[i];
            if constexpr (PLAIN) {
    #pragma unroll
                for (int i = 0; i < BCH; ++i) *reinterpret_cast<f32x4*>(Bs + buf * BSZ + pk * T::LDKB + (pq + PL * i) * 4) = rb[i];
                return;
            }
    #pragma unroll
            for (int i = 0; i < BCH; ++i) *reinterpret_cast<f32x4*>(Bs + buf * BSZ + (bk + BKS * i) * T::LDKB + bq * 4) = rb[i];
        };
        auto mma = [&](auto part_c, int cur) {
            constexpr int PART = decltype(part_c)::value;
            if constexpr (BF) mma_bf<T::TM, T::TN, PART>(AsH + cur * ASZH, BsH + cur * BSZH, wm * T::TM * 32, wn * T::TN * 32, acc);
            else mma_kk<T::TM, T::TN, PART>(As + cur * ASZ, Bs + cur * BSZ, T::LDKA, T::LDKB, wm * T::TM * 32, wn * T::TN * 32, acc);
        };
        const int nkt = k_begin < k_end ? (k_end - k_begin + BK2 - 1) / BK2 : 0;  // (one basic block per stage: see igemm2_fwd_body)
        if (nkt > 0) {
            load_tile(k_begin);
            store_tile(0);
            __syncthreads();
            load_tile(k_begin + BK2);
        }
        for (int it = 0; it < nkt; ++it) {
            const int cur = it & 1;
            mma(std::integral_constant<int, 0>{}, cur);
            MOVAE_SCHED_PIN();
            store_tile(cur ^ 1);
            MOVAE_SCHED_PIN();
            load_tile(k_begin + (it + 2) * BK2);
            mma(std::integral_constant<int, 1>{}, cur);
            __syncthreads();
        }
    };
    const bool cs_blk = a.cs_on && by == 0;  // (host: only with a plain Sm operand, i.e. nside != 1)
    if (nside == 0) {
        if (cs_blk) pipeline(std::true_type{}, std::true_type{});
        else pipeline(std::true_type{}, std::false_type{});
    } else {
        if (cs_blk) pipeline(std::false_type{}, std::true_type{});
        else pipeline(std::false_type{}, std::false_type{});
    }
    if (cs_blk) {  // fold the AKS k-lanes of every column through LDS (free after the loop's last barrier), fixed order
        // (the prologue's stage-0 store ran once per stage like all others: every staged value was added exactly once)
        float* fs = smem;
        *reinterpret_cast<f32x4*>(fs + ak * BM + aq * 4) = csum;
        __syncthreads();
        if (t < BM && m0 + t < M) {
            float v = 0.f;
    #pragma unroll
            for (int i = 0; i < AKS; ++i) v += fs[i * BM + t];
            if (to_slab) out[(long)bz * a.slab_stride + (long)M * N + m0 + t] = v;
            else if (a.cs.p[grp]) a.cs.p[grp][m0 + t] = v;
        }
    }

    const int lane = t & 63, half = lane >> 5, l31 = lane & 31;
    float* dst = to_slab ? out + (long)bz * a.slab_stride : a.tab.p[grp];
    if ((N & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {  // through LDS in 16-byte pieces (see igemm2_fwd_body)
        constexpr int LDT = BN + 4, QPR = BN / 4;
        static_assert(BM * LDT <= WgSmem<BM, BN>::FLOATS, "tile fits the stage buffers");
        __syncthreads();  // (the column-sum fold above may still be reading LDS)
        float* Ts = smem;
#pragma unroll
        for (int tn = 0; tn < T::TN; ++tn)
#pragma unroll
            for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Ts[(wm * T::TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * LDT + wn * T::TN * 32 + tn * 32 + l31] =
                        acc[tm * T::TN + tn][r];
        __syncthreads();
        for (int q = t; q < BM * QPR; q += 256) {
            const int row = q / QPR, n = n0 + (q - row * QPR) * 4, m = m0 + row;
            if (m < M && n < N) *reinterpret_cast<f32x4*>(dst + (long)m * N + n) = *reinterpret_cast<const f32x4*>(Ts + row * LDT + (n - n0));
        }
        return;
    }
#pragma unroll
    for (int tn = 0; tn < T::TN; ++tn) {
        const int n = n0 + wn * T::TN * 32 + tn * 32 + l31;
        if (n >= N) continue;
#pragma unroll
        for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * T::TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < M) dst[(long)m * N + n] = acc[tm * T::TN + tn][r];
            }
    }
}

template <int BM, int BN, bool BF = false>
__global__ __launch_bounds__(256) void igemm2_wgrad(WgArgs a, RSide sd, int gz, int xr) {
    __shared__ __attribute__((aligned(16))) float smem[WgSmem<BM, BN>::FLOATS];
    if ((int)blockIdx.z >= gz) {  // a parked weight-gradient reduce rides BEHIND this launch's own blocks (conv_igemm.hip: RSide)
        const int bid = (((int)blockIdx.z - gz) * (int)gridDim.y + (int)blockIdx.y) * (int)gridDim.x + (int)blockIdx.x;
        if (bid < sd.nblk) side_reduce(sd, bid, smem);
        return;
    }
    int bx = blockIdx.x, by = blockIdx.y;
    xcd_tile(xr & 1, bx, by);
    igemm2_wgrad_body<BM, BN, BF>(a, smem, bx, by, blockIdx.z);
}

// ---- one launch, two problems: the input gradient (FWD or BWD gather form) and the weight gradient of one layer ----------
// blocks [0, nd) run the dgrad body on its (dgx, dgy, *) grid, the rest the wgrad body on (wgx, wgy, *).  The two only
// share read-only operands.  Each alone is a latency-bound launch that cannot fill 256 CUs (C2: <= 2 blocks per CU, one
// 32-deep k-stage chain each); together the CU's wave slots and the HBM queue stay busy.  LDS: one allocation, the larger
// of the two bodies' needs.
template <int FORM, int ABM, int ABN, int WBM, int WBN>
__global__ __launch_bounds__(256) void igemm2_pair(FwdArgs fa, BwdArgs ba, WgArgs wa, int nd, int dgx, int dgy, int wgx, int wgy,
                                                   int nw, int inter, RSide sd) {
    constexpr int DF = FORM == 0 ? FwdSmem<ABM, ABN>::FLOATS : BwdSmem<ABM, ABN>::FLOATS;
    constexpr int WF = WgSmem<WBM, WBN>::FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[DF > WF ? DF : WF];
    int b = blockIdx.x;
    if (b >= nd + nw) {  // ... and, behind the two, the previous layer's parked weight-gradient reduce (conv_igemm.hip: RSide)
        side_reduce(sd, b - nd - nw, smem);
        return;
    }
    const int rev = inter >> 2;  // (bit 2: cls_order())
    inter &= 3;
    if (inter == 2) {  // weight-gradient blocks first
        b = b < nw ? nd + b : b - nw;
    } else if (inter) {  // alternate the two problems' blocks while both last (the dispatcher hands out blocks in index order)
        const int m = nd < nw ? nd : nw;
        if (b < 2 * m) b = (b & 1) ? nd + (b >> 1) : (b >> 1);
        else b = nd < nw ? b : b - m;  // tail: the longer problem's remaining blocks (dgrad tail keeps indices m.., wgrad tail nd + m..)
    }
    if (b < nd) {
        const int bx = b % dgx, r = b / dgx;
        if (FORM == 0)
            igemm2_fwd_body<ABM, ABN>(fa, smem, bx, r % dgy, r / dgy);
        else
            igemm2_bwd_body<ABM, ABN>(ba, smem, bx, r % dgy, rev ? nd / (dgx * dgy) - 1 - r / dgy : r / dgy);  // (cls_order())
    } else {
        b -= nd;
        const int bx = b % wgx, r = b / wgx;
        igemm2_wgrad_body<WBM, WBN>(wa, smem, bx, r % wgy, r / wgy);
    }
}

#undef ZERO4

// ---- host side ------------------------------------------------------------------------------------
// Pairing (movae_conv*_dgrad_wgrad*): while g_pair_collect is set, a dgrad that lands on one of the small-tile kernels is
// PLANNED but not launched -- its arguments wait in g_pending -- and the wgrad that follows launches both through
// igemm2_pair when it lands on a small-tile kernel too.  Anything else flushes the pending dgrad as an ordinary launch.
struct PendingDgrad {
    bool active = false;
    int form = 0, bm = 0, bn = 0;  // 0 = FWD gather (transposed-conv dgrad), 1 = BWD gather (conv dgrad)
    FwdArgs fa;
    BwdArgs ba;
    int gx = 0, gy = 0, gz = 0;
    size_t ws_used = 0;            // bytes of the scratch arena taken by the dgrad's slabs
    // split-K epilogue of the dgrad, issued after the (paired or plain) main launch
    bool reduce = false;
    int S = 1;
    long total = 0;
    BnBwd rbb{};          // rbb.y != null: the reduce also emits the fused BatchNorm's backward sums (rows per block: rbb_rpb)
    int rbb_rpb = 0;
    ActMul ram{nullptr, 0, 0.f, 0, 0, nullptr};  // set: the reduce applies the ActMul (activation derivative / residual) instead of the epilogue
};
static thread_local PendingDgrad g_pending;
static thread_local bool g_pair_collect = false;

// a paired launch brings the partner's blocks onto the chip as well: the split-K cost model is asked about this many
// times the own tiles (tuning knob MOVAE_PAIR_TILES, percent)
inline long pair_tiles(long tiles, bool paired) {
    static const int pct = getenv("MOVAE_PAIR_TILES") ? atoi(getenv("MOVAE_PAIR_TILES")) : 100;
    return paired ? (tiles * pct + 99) / 100 : tiles;
}

template <int BM, int BN>
constexpr bool pair_dgrad_tile() { return (BM == 64 && BN == 64) || (BM == 128 && BN == 32); }
template <int BM, int BN>
constexpr bool pair_wgrad_tile() { return (BM == 64 && BN == 64) || (BM == 32 && BN == 128); }

// The block-internal split-K kernels (kgemm.h, included after this file) stash their input-gradient launch the same way; the
// weight-gradient launchers below reach it through these hooks: pending? / launch it together with this weight gradient (one
// kernel: kpair_k) / launch it on its own.
struct KPairHooks {
    bool (*pending)();
    int (*pair)(const WgArgs& wa, int wgx, int wgy, int wgz, bool w64, hipStream_t st);
    int (*flush)(hipStream_t st);
};
static KPairHooks g_kpair{nullptr, nullptr, nullptr};

inline int finish_pending(hipStream_t st) {  // the stashed dgrad's reduce
    PendingDgrad& p = g_pending;
    if (!p.reduce || g_bench_main_only) return MOVAE_OK;
    if (p.form == 0) {
        const FwdArgs& a = p.fa;
        if (p.rbb.y) {
            launch_reduce_bnbwd(a.slab, a.Y, a.M, a.g.Nn, p.S, nullptr, 0, 0, 0, p.rbb, p.rbb_rpb, st);
            MOVAE_CHECK_LAUNCH("splitk_reduce_stats (bn bwd)");
            return MOVAE_OK;
        }
        return launch_reduce(a.slab, a.Y, (long)a.M * a.g.Nn, p.S, a.g.Nn, a.ep.bias, a.ep.act, a.ep.slope, 0, st, nullptr, 0, p.ram);
    }
    const BwdArgs& a = p.ba;
    if (p.rbb.y) {
        launch_reduce_bnbwd(a.slab, a.Y, (long)a.g.Nimg * a.g.Ho * a.g.Wo, a.g.Nn, p.S, &a.scls, a.g.Ho, a.g.Wo, a.g.stride, p.rbb, p.rbb_rpb, st);
        MOVAE_CHECK_LAUNCH("splitk_reduce_stats (bn bwd)");
        return MOVAE_OK;
    }
    long gq = (a.total / 4 + 255) / 256;
    if (gq > 4096) gq = 4096;
    hipLaunchKernelGGL(splitk_reduce_cls, dim3((unsigned)gq), dim3(256), 0, st, a.slab, a.Y, a.total, a.g.Nn, a.g.Ho, a.g.Wo, a.g.stride,
                       a.scls, a.ep.bias, a.ep.act, a.ep.slope, p.ram);
    MOVAE_CHECK_LAUNCH("splitk_reduce_cls");
    return MOVAE_OK;
}

// MOVAE_CLS_ORDER=1: a BWD-form launch enumerates its (class, split) pairs last class first (bit 1 of the kernels' `xr`).  Measured
// neutral (C2 0.759 vs 0.768 ms, inside the noise; C1, C5 level: the per-class split factors already balance the classes): off.
inline int cls_order() {
    static const int v = getenv("MOVAE_CLS_ORDER") ? atoi(getenv("MOVAE_CLS_ORDER")) : 0;
    return v ? 2 : 0;
}

// MOVAE_XCD_REMAP: 1 = XCD-aware tile map (xcd_tile) for launches of at least 64 tiles per z slice, 0 = tile = block id
inline int xcd_remap(const dim3& grid) {
    static const int mode = getenv("MOVAE_XCD_REMAP") ? atoi(getenv("MOVAE_XCD_REMAP")) : 0;
    return mode && (long)grid.x * grid.y >= 64 ? 1 : 0;
}

inline int flush_pending(hipStream_t st) {  // launch the stashed dgrad on its own
    if (g_kpair.flush)
        if (int rc = g_kpair.flush(st)) return rc;
    PendingDgrad& p = g_pending;
    if (!p.active) return MOVAE_OK;
    p.active = false;
    dim3 grid(p.gx, p.gy, p.gz);
    int gz;
    const RSide sd = defer_take_3d(st, &grid, &gz);
    if (p.form == 0) {
        if (p.bm == 64) hipLaunchKernelGGL((igemm2_fwd<64, 64>), grid, dim3(256), 0, st, p.fa, sd, gz, xcd_remap(grid));
        else hipLaunchKernelGGL((igemm2_fwd<128, 32>), grid, dim3(256), 0, st, p.fa, sd, gz, xcd_remap(grid));
    } else {
        if (p.bm == 64) hipLaunchKernelGGL((igemm2_bwd<64, 64>), grid, dim3(256), 0, st, p.ba, sd, gz, xcd_remap(grid) | cls_order());
        else hipLaunchKernelGGL((igemm2_bwd<128, 32>), grid, dim3(256), 0, st, p.ba, sd, gz, xcd_remap(grid) | cls_order());
    }
    MOVAE_CHECK_LAUNCH("igemm2 dgrad (unpaired)");
    return finish_pending(st);
}

template <int BM, int BN>
int launch_fwd2(const float* X, const float* W, float* Y, const Geom& g, const Epilogue& ep, int M, int K, void* ws,
                size_t ws_bytes, hipStream_t st) {
    const int gx = ceil_div(M, BM), gy = ceil_div(g.Nn, BN);
    const int nk = ceil_div(K, BK2);
    const bool pairing = g_pair_collect && pair_dgrad_tile<BM, BN>();
    int S = choose_split(FORM_FWD, BM * BN, BK2, pair_tiles((long)gx * gy, pairing), nk, (size_t)M * g.Nn * sizeof(float), ws_bytes,
                         ws != nullptr);
    const int per_split = ceil_div(nk, S);
    S = ceil_div(nk, per_split);
    float* slab = S > 1 ? static_cast<float*>(ws) : nullptr;
    FwdArgs a{X, W, Y, g, ep, M, K, per_split, slab};
    a.fd_cr = fastdiv_make(g.Cr), a.fd_kw = fastdiv_make(g.KW), a.fd_wlen = fastdiv_make(g.wlen > 0 ? g.wlen : 1);
    a.nrm = g_fuse.nrm;
    // statistics of the result for the BatchNorm that follows (never on a backward pass: those are the paired / collected ones)
    const bool want_stats = g_fuse.stats && ep.act == MOVAE_ACT_NONE && !g_pair_collect;
    // BatchNorm side products through the LDS tile epilogue (one partial pair per block) where its 16-byte pieces apply
    a.side_lds = (g.Nn % 4 == 0 && ((reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(ep.bias) |
                                     reinterpret_cast<uintptr_t>(g_fuse.bn_y)) & 15) == 0) ? 1 : 0;
    const int slots = a.side_lds ? 1 : T2<BM, BN>::WM;
    if (want_stats && S == 1) a.stats = fuse_stats_claim((long)gx * slots, g.Nn);
    // the result is a fused BatchNorm's output gradient: its backward sums from the epilogue (unsplit) or from the reduce
    BnBwd rbb{};
    int rbb_rpb = 0;
    if (g_fuse.bn_y && ep.act == MOVAE_ACT_NONE && !ep.bias) {
        const long rpg = M / g_fuse.bn_groups;
        if (S == 1) {
            if (M % g_fuse.bn_groups == 0 && rpg % BM == 0) {
                const long ppg = rpg / BM * slots;
                if (float* part = fuse_bn_claim(ppg, g.Nn))
                    a.bb = BnBwd{g_fuse.bn_y, g_fuse.bn_scale, g_fuse.bn_shift, g_fuse.bn_slope, part, (int)rpg, (int)ppg};
            }
        } else {
            plan_reduce_bnbwd(M, g.Nn, &rbb, &rbb_rpb);
        }
    }
    // the previous layer's activation derivative on the result: in the epilogue (unsplit) or in the reduce
    ActMul ram{nullptr, 0, 0.f, 0, 0, nullptr};
    if ((g_fuse.am.y || g_fuse.am.res) && ep.act == MOVAE_ACT_NONE && (!ep.bias || !g_fuse.am.y) && !rbb.y && !a.bb.y &&
        M % g_fuse.am_groups == 0 && g.Nn % 4 == 0 && ((reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(ep.bias)) & 15) == 0 &&
        !want_stats) {
        const long rpg = M / g_fuse.am_groups;
        if (g_fuse.am_groups == 1 || rpg % BM == 0 || S > 1 || !g_fuse.am.y) {
            ActMul am = g_fuse.am;
            am.per_group = rpg * g.Nn;
            if (S == 1) a.am = am;
            else ram = am;
            g_fuse.am_done = true;
        }
    }
    if (g_pair_collect && pair_dgrad_tile<BM, BN>()) {
        PendingDgrad& p = g_pending;
        p.active = true;
        p.form = 0, p.bm = BM, p.bn = BN, p.fa = a, p.gx = gx, p.gy = gy, p.gz = S;
        p.rbb = rbb, p.rbb_rpb = rbb_rpb, p.ram = ram;
        p.reduce = S > 1, p.S = S, p.total = (long)M * g.Nn;
        p.ws_used = S > 1 ? (size_t)M * g.Nn * sizeof(float) * S : 0;
        return MOVAE_OK;
    }
    dim3 grid(gx, gy, S);
    int gz;
    const RSide sd = defer_take_3d(st, &grid, &gz);
    if (BM == 128 && BN == 128 && g_compute_bf16) hipLaunchKernelGGL((igemm2_fwd<BM, BN, BM == 128 && BN == 128>), grid, dim3(256), 0, st, a, sd, gz, xcd_remap(grid));
    else hipLaunchKernelGGL((igemm2_fwd<BM, BN>), grid, dim3(256), 0, st, a, sd, gz, xcd_remap(grid));
    MOVAE_CHECK_LAUNCH("igemm2_fwd");
    if (S > 1) {
        if (rbb.y && !g_bench_main_only) {
            launch_reduce_bnbwd(slab, Y, M, g.Nn, S, nullptr, 0, 0, 0, rbb, rbb_rpb, st);
            MOVAE_CHECK_LAUNCH("splitk_reduce_stats (bn bwd)");
            return MOVAE_OK;
        }
        if (want_stats && !g_bench_main_only && launch_reduce_stats(slab, Y, M, g.Nn, S, nullptr, 0, 0, 0, ep.bias, st)) {
            MOVAE_CHECK_LAUNCH("splitk_reduce_stats");
            return MOVAE_OK;
        }
        return launch_reduce(slab, Y, (long)M * g.Nn, S, g.Nn, ep.bias, ep.act, ep.slope, 0, st, nullptr, 0, ram);
    }
    return MOVAE_OK;
}

template <int BM, int BN>
int launch_bwd2(const float* X, const float* W, float* Y, const Geom& g, const Epilogue& ep, void* ws, size_t ws_bytes,
                hipStream_t st) {
    const int s = g.stride;
    const long Mmax = (long)g.Nimg * ceil_div(g.Ho, s) * ceil_div(g.Wo, s);
    const int gx = ceil_div(Mmax, BM), gy = ceil_div(g.Nn, BN);
    const long total = (long)g.Nimg * g.Ho * g.Wo * g.Nn;
    // k-tiles of every output-parity class (same arithmetic as the kernel)
    int nk_c[4] = {0, 0, 0, 0}, nk_max = 0;
    long nk_sum = 0;
    const int ncls = s * s <= 4 ? s * s : 0;
    for (int c = 0; c < ncls; ++c) {
        const int ph = c / s, pw = c % s;
        const int kh0 = (ph + g.pad) % s, kw0 = (pw + g.pad) % s;
        const int nA = kh0 < g.KH ? (g.KH - kh0 + s - 1) / s : 0, nB = kw0 < g.KW ? (g.KW - kw0 + s - 1) / s : 0;
        nk_c[c] = ceil_div((long)nA * nB * g.Cr, BK2);
        nk_max = nk_c[c] > nk_max ? nk_c[c] : nk_max;
        nk_sum += nk_c[c];
    }
    if (ncls == 0 || nk_max == 0) {
        movae_set_error("conv bwd-form: stride %d unsupported by the fast path", s);
        return MOVAE_EUNSUPPORTED;
    }
    // split factor of the heaviest class from the cost model, on the block count the balanced grid will have
    const long tiles_eff = pair_tiles(ceil_div((long)gx * gy * nk_sum, nk_max), g_pair_collect && pair_dgrad_tile<BM, BN>());
    int Smax = choose_split(FORM_BWD, BM * BN, BK2, tiles_eff, nk_max, (size_t)total * sizeof(float), ws_bytes, ws != nullptr);
    // the cost model sees balanced blocks; unsplit but unbalanced (heaviest class >= 2x the lightest) it is better to split
    int nk_min = nk_max;
    for (int c = 0; c < ncls; ++c)
        if (nk_c[c] > 0 && nk_c[c] < nk_min) nk_min = nk_c[c];
    static const int balance = getenv("MOVAE_BWD_BALANCE") ? atoi(getenv("MOVAE_BWD_BALANCE")) : 1;
    // (measured: pays while the output is below ~5 MB -- every extra slab writes and re-reads it once)
    if (balance && Smax == 1 && ws && nk_max >= 2 * nk_min && nk_min >= 4 && (size_t)total * sizeof(float) <= (5u << 20) &&
        (size_t)total * sizeof(float) * (nk_max / nk_min) <= ws_bytes && g.Nn % 4 == 0)
        Smax = nk_max / nk_min;
    ClsSplit scls, kps;
    int zsum = 0, Sreal = 1;
    for (int c = 0; c < 4; ++c) {
        int Sc = 1, per = 1;
        if (c < ncls && nk_c[c] > 0) {
            Sc = balance ? (int)(((long)Smax * nk_c[c] + nk_max - 1) / nk_max) : Smax;
            if (Sc < 1) Sc = 1;
            per = ceil_div(nk_c[c], Sc);
            Sc = ceil_div(nk_c[c], per);
        }
        scls.s[c] = c < ncls ? Sc : 0;
        kps.s[c] = per;
        if (c < ncls) {
            zsum += Sc;
            Sreal = Sc > Sreal ? Sc : Sreal;
        }
    }
    float* slab = Sreal > 1 ? static_cast<float*>(ws) : nullptr;
    if (slab && ((size_t)total * sizeof(float) * Sreal > ws_bytes || g.Nn % 4 != 0)) {
        movae_set_error("conv bwd-form: workspace too small for %d slabs", Sreal);
        return MOVAE_EINVAL;
    }
    BwdArgs a{X, W, Y, g, ep, scls, kps, slab, total};
    a.fd_cr = fastdiv_make(g.Cr);
    for (int c = 0; c < 4; ++c) {
        const int ph = c / s, pw = c % s;
        const int Hoc = c < ncls ? (g.Ho - ph + s - 1) / s : 1, Woc = c < ncls ? (g.Wo - pw + s - 1) / s : 1;
        a.fd_hw[c] = fastdiv_make(Hoc * Woc > 0 ? Hoc * Woc : 1), a.fd_w[c] = fastdiv_make(Woc > 0 ? Woc : 1);
        const int kw0 = (pw + g.pad) % s, nB = kw0 < g.KW ? (g.KW - kw0 + s - 1) / s : 0;
        a.fd_nb[c] = fastdiv_make(nB > 0 ? nB : 1);
    }
    a.nrm = g_fuse.nrm;
    a.stats_gx = gx;
    const bool want_stats = g_fuse.stats && ep.act == MOVAE_ACT_NONE && !g_pair_collect;
    a.side_lds = (g.Nn % 4 == 0 && ((reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(ep.bias) |
                                     reinterpret_cast<uintptr_t>(g_fuse.bn_y)) & 15) == 0) ? 1 : 0;  // (see launch_fwd2)
    const int slots = a.side_lds ? 1 : T2<BM, BN>::WM;
    if (want_stats && Sreal == 1) a.stats = fuse_stats_claim((long)ncls * gx * slots, g.Nn);
    BnBwd rbb{};
    int rbb_rpb = 0;
    if (g_fuse.bn_y && ep.act == MOVAE_ACT_NONE && !ep.bias) {
        const long pix = (long)g.Nimg * g.Ho * g.Wo;  // output pixels over all cotangent groups
        if (Sreal == 1) {
            // equally large classes (even output grid) and whole row blocks per group and class
            const long rows_c = pix / ncls / g_fuse.bn_groups;
            if (g.Ho % s == 0 && g.Wo % s == 0 && pix % ((long)ncls * g_fuse.bn_groups) == 0 && rows_c % BM == 0) {
                const long ppg = (long)ncls * (rows_c / BM) * slots;
                if (float* part = fuse_bn_claim(ppg, g.Nn))
                    a.bb = BnBwd{g_fuse.bn_y, g_fuse.bn_scale, g_fuse.bn_shift, g_fuse.bn_slope, part, (int)(pix / g_fuse.bn_groups), (int)ppg};
            }
        } else {
            plan_reduce_bnbwd(pix, g.Nn, &rbb, &rbb_rpb);
        }
    }
    ActMul ram{nullptr, 0, 0.f, 0, 0, nullptr};  // (see launch_fwd2)
    if ((g_fuse.am.y || g_fuse.am.res) && ep.act == MOVAE_ACT_NONE && (!ep.bias || !g_fuse.am.y) && !rbb.y && !a.bb.y && g.Nn % 4 == 0 &&
        ((reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(ep.bias)) & 15) == 0 && !want_stats) {
        const long pix = (long)g.Nimg * g.Ho * g.Wo;
        const int G = g_fuse.am_groups;
        if (pix % G == 0) {
            ActMul am = g_fuse.am;
            am.per_group = pix / G * g.Nn;
            bool ok = true;
            if (Sreal == 1 && G > 1 && am.y) {  // epilogue: whole row blocks per group and class (equally large classes)
                const long rows_c = pix / ncls / G;
                ok = g.Ho % s == 0 && g.Wo % s == 0 && pix % ((long)ncls * G) == 0 && rows_c % BM == 0;
                am.gx_per_group = ok ? (int)(rows_c / BM) : 0;
            }
            if (ok) {
                if (Sreal == 1) a.am = am;
                else ram = am;
                g_fuse.am_done = true;
            }
        }
    }
    if (g_pair_collect && pair_dgrad_tile<BM, BN>()) {
        PendingDgrad& p = g_pending;
        p.active = true;
        p.form = 1, p.bm = BM, p.bn = BN, p.ba = a, p.gx = gx, p.gy = gy, p.gz = zsum;
        p.rbb = rbb, p.rbb_rpb = rbb_rpb, p.ram = ram;
        p.reduce = Sreal > 1, p.S = Sreal, p.total = total;
        p.ws_used = Sreal > 1 ? (size_t)total * sizeof(float) * Sreal : 0;
        return MOVAE_OK;
    }
    dim3 grid(gx, gy, zsum);
    int gz;
    const RSide sd = defer_take_3d(st, &grid, &gz);
    if (BM == 128 && BN == 128 && g_compute_bf16) hipLaunchKernelGGL((igemm2_bwd<BM, BN, BM == 128 && BN == 128>), grid, dim3(256), 0, st, a, sd, gz, xcd_remap(grid) | cls_order());
    else hipLaunchKernelGGL((igemm2_bwd<BM, BN>), grid, dim3(256), 0, st, a, sd, gz, xcd_remap(grid) | cls_order());
    MOVAE_CHECK_LAUNCH("igemm2_bwd");
    if (Sreal > 1 && !g_bench_main_only) {
        if (rbb.y) {
            launch_reduce_bnbwd(slab, Y, (long)g.Nimg * g.Ho * g.Wo, g.Nn, Sreal, &scls, g.Ho, g.Wo, s, rbb, rbb_rpb, st);
            MOVAE_CHECK_LAUNCH("splitk_reduce_stats (bn bwd)");
            return MOVAE_OK;
        }
        if (want_stats && launch_reduce_stats(slab, Y, (long)g.Nimg * g.Ho * g.Wo, g.Nn, Sreal, &scls, g.Ho, g.Wo, s, ep.bias, st)) {
            MOVAE_CHECK_LAUNCH("splitk_reduce_stats");
            return MOVAE_OK;
        }
        long gq = (total / 4 + 255) / 256;
        if (gq > 4096) gq = 4096;
        hipLaunchKernelGGL(splitk_reduce_cls, dim3((unsigned)gq), dim3(256), 0, st, slab, Y, total, g.Nn, g.Ho, g.Wo, s, scls, ep.bias,
                           ep.act, ep.slope, ram);
        MOVAE_CHECK_LAUNCH("splitk_reduce_cls");
    }
    return MOVAE_OK;
}

// Order of a paired launch's blocks (MOVAE_PAIR_INTERLEAVE): 2 = the weight gradient's first (default: they are the longer ones --
// a deep reduction slice each -- and the dispatcher hands blocks out in index order, so the short input-gradient blocks fill in
// behind them; C2 0.793 -> 0.780 ms, C4 level), 1 = alternating, 0 = the input gradient's first.
inline int pair_order() {
    static const int v = getenv("MOVAE_PAIR_INTERLEAVE") ? atoi(getenv("MOVAE_PAIR_INTERLEAVE")) : 2;
    return v;
}

template <int FORM, int ABM, int ABN, int WBM, int WBN>
inline void launch_pair(const PendingDgrad& p, const WgArgs& wa, int wgx, int wgy, int wgz, hipStream_t st) {
    const int nd = p.gx * p.gy * p.gz, nw = wgx * wgy * wgz;
    const int inter = pair_order() | (cls_order() ? 4 : 0);
    const RSide sd = defer_take(st);  // the previous layer's parked weight-gradient reduce rides behind the two problems
    hipLaunchKernelGGL((igemm2_pair<FORM, ABM, ABN, WBM, WBN>), dim3(nd + nw + sd.nblk), dim3(256), 0, st, p.fa, p.ba, wa, nd, p.gx, p.gy,
                       wgx, wgy, nw, inter, sd);
}

template <int BM, int BN>
int launch_wgrad2(const float* Sm, const float* Bg, float* const* dW, int G, long s_gs, long b_gs, const WGeom& g, int K,
                  int accumulate, void* ws, size_t ws_bytes, hipStream_t st, float* const* colsum = nullptr) {
    const int M = g.Cs, N = g.KH * g.KW * g.Cb;
    const long stride = (long)M * N + (colsum ? M : 0);  // floats per slab
    const int gx = ceil_div(M, BM), gy = ceil_div(N, BN);
    // the G groups run side by side, so the split factor is chosen for G times the tiles
    int Sp = choose_split(FORM_WGRAD, BM * BN, BK2, pair_tiles((long)gx * gy * G, (g_pending.active || (g_kpair.pending && g_kpair.pending())) && pair_wgrad_tile<BM, BN>()),
                          ceil_div(K, BK2), (size_t)stride * sizeof(float) * G, ws_bytes, ws != nullptr);
    const int kchunk = ceil_div(ceil_div(K, Sp), BK2) * BK2;
    Sp = ceil_div(K, kchunk);
    const bool slab = Sp > 1 || accumulate;
    if (slab && (!ws || (size_t)stride * sizeof(float) * Sp * G > ws_bytes)) {
        movae_set_error("wgrad: workspace too small (%zu bytes) for %d x %d splits of %dx%d", ws_bytes, G, Sp, M, N);
        return MOVAE_EINVAL;
    }
    WOut tab;
    for (int i = 0; i < 8; ++i) tab.p[i] = i < G ? dW[i] : nullptr;
    float* out = slab ? static_cast<float*>(ws) : nullptr;
    WgArgs a{Sm, Bg, out, g, K, kchunk, slab ? 1 : 0, Sp, s_gs, b_gs, tab};
    a.fd_hw = fastdiv_make(g.Hs * g.Ws), a.fd_ws = fastdiv_make(g.Ws);
    a.nrm = g_fuse.nrm;  // virtual activation operand: Sm (transposed-conv x) or Bg (conv x), as the entry point says
    a.nrm_side = g_fuse.nrm_side;
    a.cs_on = colsum ? 1 : 0;
    a.slab_stride = stride;
    for (int i = 0; i < 8; ++i) a.cs.p[i] = (colsum && i < G) ? colsum[i] : nullptr;
    PendingDgrad& p = g_pending;
    if (g_kpair.pending && g_kpair.pending() && pair_wgrad_tile<BM, BN>()) {  // a kgemm.h input gradient waits: one launch for both
        if (int rc = g_kpair.pair(a, gx, gy, Sp * G, BM == 64, st)) return rc;
    } else if (p.active && pair_wgrad_tile<BM, BN>() && (long)p.gx * p.gy * p.gz + (long)gx * gy * Sp * G < 0x7fffffffL) {
        p.active = false;
        constexpr int W64 = BM == 64 ? 1 : 0;  // wgrad tile: <64,64> or <32,128>
        // names as rocprofv3 prints the instantiations: <form, dgrad tile, wgrad tile>
        if (p.form == 0 && p.bm == 64) {
            launch_pair<0, 64, 64, W64 ? 64 : 32, W64 ? 64 : 128>(p, a, gx, gy, Sp * G, st);
            g_last_kernel = W64 ? "igemm2_pair<0,64,64,64,64>" : "igemm2_pair<0,64,64,32,128>";
        } else if (p.form == 0) {
            launch_pair<0, 128, 32, W64 ? 64 : 32, W64 ? 64 : 128>(p, a, gx, gy, Sp * G, st);
            g_last_kernel = W64 ? "igemm2_pair<0,128,32,64,64>" : "igemm2_pair<0,128,32,32,128>";
        } else if (p.bm == 64) {
            launch_pair<1, 64, 64, W64 ? 64 : 32, W64 ? 64 : 128>(p, a, gx, gy, Sp * G, st);
            g_last_kernel = W64 ? "igemm2_pair<1,64,64,64,64>" : "igemm2_pair<1,64,64,32,128>";
        } else {
            launch_pair<1, 128, 32, W64 ? 64 : 32, W64 ? 64 : 128>(p, a, gx, gy, Sp * G, st);
            g_last_kernel = W64 ? "igemm2_pair<1,128,32,64,64>" : "igemm2_pair<1,128,32,32,128>";
        }
        MOVAE_CHECK_LAUNCH("igemm2_pair");
        if (int rc = finish_pending(st)) return rc;
    } else {
        if (int rc = flush_pending(st)) return rc;
        dim3 grid(gx, gy, Sp * G);
        int gz;
        const RSide sd = defer_take_3d(st, &grid, &gz);
        if (BM == 128 && BN == 128 && g_compute_bf16) hipLaunchKernelGGL((igemm2_wgrad<BM, BN, BM == 128 && BN == 128>), grid, dim3(256), 0, st, a, sd, gz, xcd_remap(grid));
        else hipLaunchKernelGGL((igemm2_wgrad<BM, BN>), grid, dim3(256), 0, st, a, sd, gz, xcd_remap(grid));
        MOVAE_CHECK_LAUNCH("igemm2_wgrad");
    }
    if (slab) {  // ONE reduce launch for all groups (blockIdx.y = group)
        RGroups rg{};
        for (int i = 0; i < G; ++i) rg.out[i] = dW[i], rg.out2[i] = colsum ? colsum[i] : nullptr;
        rg.slab_gs = (long)Sp * stride;
        // (deferrable: movae_reduce_defer armed this call -- the reduce waits for the next launch that can carry it)
        return launch_reduce_groups(out, rg, G, (long)M * N, colsum ? M : 0, Sp, N, nullptr, 0, 0.f, accumulate, st,
                                    ActMul{nullptr, 0, 0.f, 0, 0, nullptr}, true);
    }
    return MOVAE_OK;
}

}  // namespace v2
