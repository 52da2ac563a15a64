// kgemm.h -- register-direct implicit GEMM with the reduction split INSIDE the block (included inside conv_igemm.hip's
// anonymous namespace, after igemm_v2.h).
//
// The 64..512-channel layers of the CIFAR VAE on 1x1..4x4 images (models/vae.py:119-126,147-158) are GEMMs of 256..4096 rows
// whose 64x64 tiles cannot fill 256 CUs: the tiled kernels split K across BLOCKS, write partial slabs and run a reduce launch
// (plus the BatchNorm finalize) -- ~10 us of latency per pass where the arithmetic needs 2-4 us, and 3.4x the algorithmic bytes.
// Here a wave owns ONE 32x32 output tile and a slice of the reduction; the KS waves that share a tile fold their accumulators
// through LDS (fixed order: bit-reproducible, no slabs, no reduce launch).  Operands go global -> registers in MFMA lane order
// (v_mfma_f32_32x32x2_f32: lane l holds A[row l & 31][k = l >> 5], B[k = l >> 5][col l & 31]) with one chunk of the slice in
// flight ahead of the chunk being multiplied; nothing is staged in LDS, so a block's only barriers are the fold's.
//   KS = 4: block = one 32x32 tile, K split four ways        (tiles <= ~512: the deep layers)
//   KS = 2: block = two row tiles, K split two ways
//   KS = 1: block = four row tiles, every wave its whole K   (B is shared through the L1)
// k assignment inside a group of 8 reduction indices: lane half h consumes k = 8g + 4h + j at step j (both operands agree).
// The three gather forms, the fused-BatchNorm epilogues (statistics / backward sums, one partial pair per 32-row tile), the
// normalise-on-load operand and the ActMul epilogue are those of igemm_v2.h (DESIGN.md sections 3.1, 3.5, 3.6).
#pragma once

namespace kg {

using v2::buf_load4;
using v2::buf_rsrc;
using v2::BUF_OOB;
using v2::rsrc_t;

#define ZERO4_ (f32x4{0.f, 0.f, 0.f, 0.f})

constexpr int CG = 2;    // 8-wide reduction groups per chunk, FWD / BWD forms (16 reduction indices per wave and chunk) ...
constexpr int NB = 6;    // ... and chunks in the register ring: five chunks of loads are in flight ahead of the chunk being multiplied
constexpr int NBW = 8;   // WGRAD form: one group per chunk (every operand element is its own 4-byte load: 8 loads per chunk and lane)
constexpr int NORM_MAX_C = 1024;  // channels of a virtual operand whose scale / shift are staged in LDS

__device__ __forceinline__ float buf_load1(rsrc_t r, int byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}

// BWD form, per output-parity class: the window of taps (ta, tb) that meets the image for at least one row of the class -- on
// 1x1 / 2x2 images most of a class's taps only ever multiply padding (1x1 -> 2x2, k3 s2: one tap of 1 / 2 / 2 / 4)
struct ClsTaps {
    int ta_lo[4], tb_lo[4], nAw[4], nBw[4];
};

struct KArgs {
    const float* X;
    const float* W;
    float* Y;
    Geom g;
    Epilogue ep;
    int M;          // FWD: output rows; BWD: rows of ONE output-parity class (the host only dispatches equally large classes)
    int K;          // FWD: reduction length (window taps * Cr)
    Norm nrm;       // virtual gathered operand (scale == null: plain)
    float* stats;   // [tiles][2][N] column partial sums of the stored result, or null
    BnBwd bb;       // the result is a fused BatchNorm's output gradient: its backward sums, [groups * ppg][2][N]
    ActMul am;      // activation derivative of the layer before / residual cotangent on the result
    int tiles_c;    // row tiles per class (BWD) / in all (FWD)
    int tpg;        // row tiles per cotangent group (and class); == tiles_c with one group
    FastDiv fd_cr, fd_kw, fd_wlen, fd_hw, fd_w;
    FastDiv fd_nb[4];  // BWD: by the class's window width nBw
    ClsTaps ct;
    BnFin fin;      // with stats: the block arriving last at its column tile finishes the BatchNorm that follows (gamma == null: off)
};

constexpr int SIDE_FLOATS = 2 * 32 * 32;      // per-thread column sums of one tile, before the row fold
template <int NW>
constexpr int smem_floats() { return NW * 16 * 64 + SIDE_FLOATS + 2 * NORM_MAX_C; }

// acc register i of lane (r, h) is C[row = (i & 3) + 8 * (i >> 2) + 4 * h][col = r]: thread t (< 256) of the block reads back row
// t >> 3, columns 4 * (t & 7) .. + 3 of tile `w`, summed over the KS waves that split its reduction (fixed order: pairwise tree)
template <int KS>
__device__ __forceinline__ f32x4 fold_tile(const float* __restrict__ red, int w, int t) {
    const int row = t >> 3, c4 = (t & 7) * 4;
    const int i = (row & 3) + 4 * (row >> 3), hh = (row >> 2) & 1;
    const float* p = red + (w * KS * 16 + i) * 64 + hh * 32 + c4;
    auto at = [&](int s) { return *reinterpret_cast<const f32x4*>(p + s * 16 * 64); };
    if (KS == 1) return at(0);
    if (KS == 2) return at(0) + at(1);
    if (KS == 4) return (at(0) + at(1)) + (at(2) + at(3));
    return ((at(0) + at(1)) + (at(2) + at(3))) + ((at(4) + at(5)) + (at(6) + at(7)));
}

// FORM 0 = FWD gather (conv fwd, convT dgrad), 1 = BWD gather (conv dgrad, convT fwd; blockIdx.z = output-parity class)
// NW waves per block, KS of them split one tile's reduction, NW / KS row tiles per block.  NRM: the gathered operand is virtual
// (a.nrm).  The k loop is straight-line code: the groups are visited in order by a SCALAR iterator (tap / channel counters
// advanced with compare-and-select, no division per group), loads of groups past the slice's end go through offsets outside
// the buffer descriptor (they return zeros without touching memory).
template <int FORM, int NW, int KS, bool NRM>
__device__ __forceinline__ void kgemm_body(const KArgs& a, float* __restrict__ smem, int bx, int by, int bz) {
    constexpr int WT = NW / KS;
    float* red = smem;
    float* side = smem + NW * 16 * 64;
    float* nsc_s = side + SIDE_FLOATS;
    float* nsh_s = nsc_s + NORM_MAX_C;
    const Geom g = a.g;
    const int t = threadIdx.x, lane = t & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wt = wave / KS, slot = wave - wt * KS;
    const int N = g.Nn, M = a.M, s = g.stride;
    const int n0 = by * 32;
    const int T = bx * WT + wt;          // this wave's row tile (class-local in the BWD form)
    const int m = T * 32 + r;
    const bool row_ok = m < M;

    // ---- the virtual operand's per-channel map into LDS (applied between load and MFMA) ----
    if (NRM) {
        for (int c = t; c < g.Cr; c += NW * 64) nsc_s[c] = a.nrm.scale[c], nsh_s[c] = a.nrm.shift[c];
        __syncthreads();
    }
    const float nslope = a.nrm.slope;

    // ---- per-form geometry ----
    int cls = 0, ph = 0, pw = 0, kh0 = 0, kw0 = 0, nBw = 1, ta_lo = 0, tb_lo = 0, qh = 0, qw = 0, K = a.K;
    int Hr = g.Ho, Wr = g.Wo;  // the row index decodes over this grid
    FastDiv fd_nb = a.fd_kw;
    if (FORM == 1) {
        cls = s * s - 1 - bz;  // heaviest class first
        ph = cls / s, pw = cls - ph * s;
        kh0 = (ph + g.pad) % s, kw0 = (pw + g.pad) % s;
        qh = (ph + g.pad - kh0) / s, qw = (pw + g.pad - kw0) / s;
        ta_lo = a.ct.ta_lo[cls], tb_lo = a.ct.tb_lo[cls], nBw = a.ct.nBw[cls];
        K = a.ct.nAw[cls] * nBw * g.Cr;
        fd_nb = a.fd_nb[cls];
        Hr = g.Ho / s, Wr = g.Wo / s;
    }
    const int hw = Hr * Wr;
    const int mm = row_ok ? m : 0;
    const int img = fdiv(mm, a.fd_hw), rem = mm - img * hw;
    const int hr = fdiv(rem, a.fd_w), wr_ = rem - hr * Wr;
    const int h0 = FORM == 0 ? hr * s - g.pad : hr + qh - ta_lo;
    const int w0 = FORM == 0 ? wr_ * s - g.pad : wr_ + qw - tb_lo;
    const int a_off = (((img * g.Hi + h0) * g.Wi + w0) * g.Cr + 4 * h) * 4;
    const int a_h0 = row_ok ? h0 : -(1 << 20);
    const rsrc_t xr = buf_rsrc(a.X);
    const rsrc_t wr = buf_rsrc(FORM == 0 ? a.W + g.woff : a.W);
    const int n = n0 + r;
    const int taps = g.KH * g.KW;
    const int wrow = g.wlen ? g.wrow : K;
    const int wlen = g.wlen ? g.wlen : (K > 0 ? K : 1), wstride = g.wlen ? g.wstride : 0;
    const int wjump = g.wlen ? g.wstride - g.wlen : 0;  // FWD: floats skipped in the stored kernel at the end of a window row
    // FWD: W[n][k], the lane's 16 bytes at k = 8g + 4h;  BWD: W[c][tap][n], four 4-byte loads at c = c0 + 4h + j
    const int b_off = n < N ? (FORM == 0 ? (n * wrow + 4 * h) * 4
                                         : ((4 * h * taps + (kh0 + s * ta_lo) * g.KW + kw0 + s * tb_lo) * N + n) * 4)
                            : BUF_OOB;
    const int b_cstep = taps * N * 4;  // BWD: one channel further

    const int G8 = K >> 3, gper = (G8 + KS - 1) / KS;
    const int g_begin = slot * gper, g_end = min(G8, g_begin + gper);

    // ---- scalar iterator over the slice's groups: (c0, ka, kb) = channel offset, tap row, tap column of the NEXT group to load
    // (FWD: ka = kh, kb = kw of the window; BWD: ka = ta, kb = tb of the class's tap window) ----
    int it_left = g_end - g_begin, it_c0, it_ka, it_kb, it_sa = 0, it_sb = 0;
    {
        const int k0 = (it_left > 0 ? g_begin : 0) * 8;
        const int tp = fdiv(k0, a.fd_cr);
        it_c0 = k0 - tp * g.Cr;
        if (FORM == 0) {
            it_ka = fdiv(tp, a.fd_kw), it_kb = tp - it_ka * g.KW;
            const int wr2 = fdiv(k0, a.fd_wlen);
            it_sa = ((it_ka * g.Wi + it_kb) * g.Cr + it_c0) * 4;
            it_sb = (wr2 * wstride + (k0 - wr2 * wlen)) * 4;
        } else {
            it_ka = fdiv(tp, fd_nb), it_kb = tp - it_ka * nBw;
        }
    }
    struct Chunk {
        f32x4 a[CG], b[CG];
        unsigned vm;   // validity of the lane's A row per group (virtual operand: padding stays zero)
        int c0[CG];    // (scalar) channel offset of the group: where its scale / shift live
    };
    auto load = [&](Chunk& c) {
        c.vm = 0;
#pragma unroll
        for (int u = 0; u < CG; ++u) {
            const bool gv = it_left > 0;
            c.c0[u] = it_c0;
            bool v;
            if (FORM == 0) {
                const int hh = a_h0 + it_ka, ww = w0 + it_kb;
                v = (int)gv & (int)((unsigned)hh < (unsigned)g.Hi) & (int)((unsigned)ww < (unsigned)g.Wi);  // (no short circuit: no branch)
                c.a[u] = buf_load4(xr, v ? a_off + it_sa : BUF_OOB);
                c.b[u] = buf_load4(wr, gv ? b_off + it_sb : BUF_OOB);
            } else {
                const int hh = a_h0 - it_ka, ww = w0 - it_kb;
                v = (int)gv & (int)((unsigned)hh < (unsigned)g.Hi) & (int)((unsigned)ww < (unsigned)g.Wi);
                const int sa = (it_c0 - (it_ka * g.Wi + it_kb) * g.Cr) * 4;
                const int sb = ((it_c0 * taps + s * (it_ka * g.KW + it_kb)) * N) * 4;
                c.a[u] = buf_load4(xr, v ? a_off + sa : BUF_OOB);
                const int boff = gv ? b_off + sb : BUF_OOB;
#pragma unroll
                for (int j = 0; j < 4; ++j) c.b[u][j] = buf_load1(wr, boff + j * b_cstep);
            }
            c.vm |= (v ? 1u : 0u) << u;
            // advance to the next group: eight channels on; at the end of the tap's channels the next tap (column, then row)
            --it_left;
            it_c0 += 8;
            const bool cw = it_c0 >= g.Cr;
            it_c0 = cw ? 0 : it_c0;
            if (FORM == 0) {
                it_sa += 32, it_sb += 32;
                it_kb += cw ? 1 : 0;
                const bool rw = it_kb >= g.KW;  // end of a window row: next input row, next stored kernel row
                it_kb = rw ? 0 : it_kb;
                it_ka += rw ? 1 : 0;
                it_sa += rw ? (g.Wi - g.KW) * g.Cr * 4 : 0;
                it_sb += rw ? wjump * 4 : 0;
            } else {
                it_kb += cw ? 1 : 0;
                const bool rw = it_kb >= nBw;
                it_kb = rw ? 0 : it_kb;
                it_ka += rw ? 1 : 0;
            }
        }
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    auto compute = [&](const Chunk& c) {
#pragma unroll
        for (int u = 0; u < CG; ++u) {
            f32x4 av = c.a[u];
            if (NRM) {
                const f32x4 sc = *reinterpret_cast<const f32x4*>(nsc_s + c.c0[u] + 4 * h), sh = *reinterpret_cast<const f32x4*>(nsh_s + c.c0[u] + 4 * h);
                av = norm_apply_if((c.vm >> u) & 1u, av, sc, sh, nslope);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], c.b[u][j], acc, 0, 0, 0);
        }
    };
    // a ring of NB chunks of registers: the loads of chunk i + NB - 1 are issued before the MFMAs of chunk i.
    // Whole trips round the ring are one basic block with ONE back edge (an exit from the middle of a trip becomes, after CFG
    // structurisation, a path back to the loop header a few loads long, and the compiler then drains the whole ring at the top
    // of every trip); the last nch % NB chunks sit in ring slots 0 .. already loaded, or in flight.
    {
        Chunk ring[NB];
        const int ng = g_end - g_begin;
        const int nch = ng > 0 ? (ng + CG - 1) / CG : 0;
        const int ntrip = nch / NB, rem = nch - ntrip * NB;
#pragma unroll
        for (int i = 0; i < NB - 1; ++i) load(ring[i]);
        for (int trip = 0; trip < ntrip; ++trip) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                load(ring[(i + NB - 1) % NB]);  // (past the slice's end: nothing is loaded)
                compute(ring[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < NB - 1; ++i)
            if (i < rem) compute(ring[i]);
    }

    // ---- fold the KS partial tiles, then the epilogue: thread t < 256 owns row t >> 3, columns 4 (t & 7) .. + 3 of each tile ----
#pragma unroll
    for (int i = 0; i < 16; ++i) red[(wave * 16 + i) * 64 + lane] = acc[i];
    __syncthreads();
    const Epilogue ep = a.ep;
    const bool st_on = a.stats != nullptr, bb_on = a.bb.y != nullptr, am_on = actmul_on(a.am);
    const bool ep_thread = NW == 4 || t < 256;
    const int row = (t >> 3) & 31, c4 = (t & 7) * 4, nq = n0 + c4;
    const bool col_ok = nq < N && ep_thread;  // N % 4 == 0
    f32x4 b4 = ZERO4_, sc4 = ZERO4_, sh4 = ZERO4_;
    if (col_ok) {
        if (ep.bias) b4 = *reinterpret_cast<const f32x4*>(ep.bias + nq);
        if (bb_on) sc4 = *reinterpret_cast<const f32x4*>(a.bb.scale + nq), sh4 = *reinterpret_cast<const f32x4*>(a.bb.shift + nq);
    }
#pragma unroll
    for (int w = 0; w < WT; ++w) {
        const int Tw = bx * WT + w, me = Tw * 32 + row;
        f32x4 v = ep_thread ? fold_tile<KS>(red, w, t) : ZERO4_;
        const bool ok = me < M && col_ok;
        f32x4 s1 = ZERO4_, s2 = ZERO4_;
        const int gi = Tw / a.tpg;  // cotangent group of the tile
        if (ok) {
            long oidx, yrow;  // output row (pixel), row of the auxiliary tensor shared by the groups
            if (FORM == 0) {
                oidx = me;
                yrow = me - (long)gi * a.tpg * 32;
            } else {
                const int im = fdiv(me, a.fd_hw), rm = me - im * hw;
                const int hc = fdiv(rm, a.fd_w), wc = rm - hc * Wr;
                oidx = (long)(im * g.Ho + (hc * s + ph)) * g.Wo + (wc * s + pw);
                yrow = oidx - (long)gi * a.tpg * 32 * (s * s);
            }
            v += b4;
            if (st_on) {
                s1 = v;
#pragma unroll
                for (int j = 0; j < 4; ++j) s2[j] = v[j] * v[j];
            } else if (bb_on) {
                const f32x4 y4 = *reinterpret_cast<const f32x4*>(a.bb.y + yrow * N + nq);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = v[j] * (fmaf(y4[j], sc4[j], sh4[j]) > 0.f ? 1.f : a.bb.slope);
                    s1[j] = d;
                    s2[j] = d * y4[j];
                }
            } else if (am_on) {
                if (a.am.y) {
                    const f32x4 y4 = *reinterpret_cast<const f32x4*>(a.am.y + yrow * N + nq);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] *= act_grad_from_out(y4[j], a.am.act, a.am.slope);
                }
                if (a.am.res) v += *reinterpret_cast<const f32x4*>(a.am.res + oidx * N + nq);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], ep.act, ep.slope);
            }
            *reinterpret_cast<f32x4*>(a.Y + oidx * N + nq) = v;
        }
        if (st_on || bb_on) {  // (block-uniform) column sums over the tile's 32 rows, fixed order
            if (ep_thread) {
                *reinterpret_cast<f32x4*>(side + row * 32 + c4) = s1;
                *reinterpret_cast<f32x4*>(side + 1024 + row * 32 + c4) = s2;
            }
            __syncthreads();
            if (t < 64) {
                const int col = t & 31, which = t >> 5;
                float c = 0.f;
#pragma unroll 8
                for (int i = 0; i < 32; ++i) c += side[which * 1024 + i * 32 + col];
                if (n0 + col < N && Tw < a.tiles_c) {
                    long pidx;
                    if (st_on) pidx = (long)cls * a.tiles_c + Tw;
                    else pidx = FORM == 0 ? (long)Tw : (long)gi * a.bb.ppg + (long)cls * a.tpg + (Tw - gi * a.tpg);
                    float* P = st_on ? a.stats : a.bb.part;
                    // (handed to another block inside this launch: write-through stores, read back with sc1 loads -- no fences)
                    if (st_on && a.fin.gamma) __hip_atomic_store(P + (pidx * 2 + which) * N + n0 + col, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else P[(pidx * 2 + which) * N + n0 + col] = c;
                }
            }
            if (w + 1 < WT) __syncthreads();
        }
    }
    // ---- the BatchNorm that follows, finished here (movae_fuse_t::fin_*): the block that arrives LAST at this column tile folds the
    // tile's partial pairs of every row tile / class and writes what bn_finalize_k would -- save_mean, save_rstd, the folded affine
    // map and the running statistics of its 32 channels.  Hand-off in the guide's write-through form: the partials were stored sc1,
    // the storing wave drains them, one lane takes an agent-scope ticket, the last arriver reads them with sc1 loads (no release /
    // acquire fence: the conv output of this block is still dirty in L2, a release here is what made the fence form lose).
    if (st_on && a.fin.gamma) {  // (block-uniform)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* flag = reinterpret_cast<unsigned*>(side);
        if (t == 0) {
            const unsigned ticket = __hip_atomic_fetch_add(a.fin.counter + by, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned last = ticket == (unsigned)a.fin.group - 1u ? 1u : 0u;
            if (last) __hip_atomic_store(a.fin.counter + by, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next launch
            flag[0] = last;
        }
        __syncthreads();
        const bool last = flag[0] != 0u;
        __syncthreads();
        if (last) {
            double* sd = reinterpret_cast<double*>(red);  // [2][32 part lanes][32 columns] doubles = 16 KB: the fold buffer is free now
            if (ep_thread) {
                // 16-byte sc1 buffer loads (aux 16), all independent: a dependent chain of 4-byte agent-scope atomic loads made this
                // tail 8-25 us long
                const int r = t >> 3;  // part lane 0..31; c4 = 4 (t & 7) as in the epilogue
                const v2::rsrc_t rs = v2::buf_rsrc_if(a.stats, col_ok);
                double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
                for (int p = r; p < a.fin.parts; p += 32) {
                    const f32x4 u = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ((p * 2 + 0) * N + nq) * 4, 0, 16));
                    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ((p * 2 + 1) * N + nq) * 4, 0, 16));
#pragma unroll
                    for (int j = 0; j < 4; ++j) s1[j] += (double)u[j], s2[j] += (double)v[j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) sd[r * 32 + c4 + j] = s1[j], sd[1024 + r * 32 + c4 + j] = s2[j];
            }
            __syncthreads();
            if (t < 32 && n0 + t < N) {
                const int n = n0 + t;
                double s1 = 0.0, s2 = 0.0;
#pragma unroll 8
                for (int r = 0; r < 32; ++r) s1 += sd[r * 32 + t], s2 += sd[1024 + r * 32 + t];  // fixed order
                const double rows = (double)a.fin.rows, mean = s1 / rows;
                double var = s2 / rows - mean * mean;
                if (var < 0.0) var = 0.0;
                const float m = (float)mean, rs = (float)(1.0 / sqrt(var + (double)a.fin.eps));
                float* o = a.fin.out;
                o[n] = m, o[N + n] = rs;
                const float sc = rs * a.fin.gamma[n];  // the same fp32 arithmetic as bn_finalize_k
                o[2 * N + n] = sc, o[3 * N + n] = fmaf(-m, sc, a.fin.beta[n]);
                if (a.fin.running_mean) {
                    const double unb = a.fin.rows > 1 ? var * (rows / (rows - 1.0)) : var, mom = (double)a.fin.momentum;
                    a.fin.running_mean[n] = (float)((1.0 - mom) * a.fin.running_mean[n] + mom * mean);
                    a.fin.running_var[n] = (float)((1.0 - mom) * a.fin.running_var[n] + mom * unb);
                }
            }
            if (by == 0 && t == 0 && a.fin.nbt) a.fin.nbt[0] += 1;
        }
    }
}

template <int FORM, int NW, int KS, bool NRM>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 2 : 2) void kgemm_k(KArgs a, RSide sd, int gz, int rev) {
    __shared__ __attribute__((aligned(16))) float smem[smem_floats<NW>()];
    if (NW == 4 && (int)blockIdx.z >= gz) {  // a parked weight-gradient reduce rides behind this launch's own blocks (conv_igemm.hip: RSide)
        const int bid = (((int)blockIdx.z - gz) * (int)gridDim.y + (int)blockIdx.y) * (int)gridDim.x + (int)blockIdx.x;
        if (bid < sd.nblk) side_reduce(sd, bid, smem);
        return;
    }
    kgemm_body<FORM, NW, KS, NRM>(a, smem, blockIdx.x, blockIdx.y, (FORM == 1 && rev) ? gz - 1 - (int)blockIdx.z : (int)blockIdx.z);  // (v2::cls_order())
}

// ---- weight gradient: dW[a][tap][b] = sum_p S[p][a] * Bg[p * s - pad + tap][b]   (M = Cs, N = taps * Cb) ------------------------
// Cb % 32 == 0: a 32-column tile lies inside one tap (kh, kw).  The reduction over the small side's pixels runs POSITION-major:
// for (hs, ws) in the window of positions whose tap meets the image { for img } -- a position's validity and its two offsets are
// then wave-uniform scalars (no per-lane pixel decode: ~6 VALU per reduction index instead of ~28), and positions that only meet
// padding are never multiplied (4x4 -> 2x2, k3 s2 p1: the corner taps skip 3 of 4 positions).  The images are what is split:
// over the KS waves of a tile and, where a layer has too few tiles for its batch, over Sp blocks (blockIdx.z = group * Sp + split;
// partial slabs, reduced by launch_reduce like the tiled kernels').
struct KWArgs {
    const float* Sm;
    const float* Bg;
    float* slab;     // Sp > 1: partial results, else null
    WGeom g;
    int ichunk;      // images per split (a multiple of 8)
    int Sp, accumulate;
    long s_gs, b_gs, slab_stride;
    v2::WOut tab;
    Norm nrm;
    int nrm_side;
};

template <int NW, int KS, int NSIDE>  // NSIDE: which operand is the virtual activation (0 none, 1 Sm, 2 Bg)
__device__ __forceinline__ void kwgrad_body(const KWArgs& a, float* __restrict__ smem, int bx, int by, int bz) {
    constexpr int WT = NW / KS;
    float* red = smem;
    const WGeom g = a.g;
    const int t = threadIdx.x, lane = t & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wt = wave / KS, slot = wave - wt * KS;
    const int M = g.Cs, N = g.KH * g.KW * g.Cb;
    const int grp = bz / a.Sp, split = bz - grp * a.Sp;
    const int n0 = by * 32, m0w = (bx * WT + wt) * 32;
    const int tap = n0 / g.Cb, b0 = n0 - tap * g.Cb;
    const int kh = tap / g.KW, kw = tap - kh * g.KW;
    float* dst = a.slab ? a.slab + (long)bz * a.slab_stride : a.tab.p[grp];
    const bool ep_thread = NW == 4 || t < 256;
    const int row = (t >> 3) & 31, c4 = (t & 7) * 4;
    // small-side positions whose tap (kh, kw) meets the big side: hb = hs * stride - pad + kh in [0, Hb)  (a window: hb is monotone)
    const int st = g.stride;
    int hs_lo = g.pad - kh > 0 ? (g.pad - kh + st - 1) / st : 0, ws_lo = g.pad - kw > 0 ? (g.pad - kw + st - 1) / st : 0;
    int hs_hi = (g.Hb - 1 + g.pad - kh) >= 0 ? (g.Hb - 1 + g.pad - kh) / st : -1, ws_hi = (g.Wb - 1 + g.pad - kw) >= 0 ? (g.Wb - 1 + g.pad - kw) / st : -1;
    hs_hi = hs_hi > g.Hs - 1 ? g.Hs - 1 : hs_hi, ws_hi = ws_hi > g.Ws - 1 ? g.Ws - 1 : ws_hi;
    const int nh = hs_hi - hs_lo + 1, nw = ws_hi - ws_lo + 1;
    if (nh <= 0 || nw <= 0) {  // the tap never meets the image: the tile is identically zero
        if (!(a.accumulate && !a.slab) && ep_thread) {
            for (int w = 0; w < WT; ++w) {
                const int me = (bx * WT + w) * 32 + row;
                if (me < M)
#pragma unroll
                    for (int j = 0; j < 4; ++j) dst[(long)me * N + n0 + c4 + j] = 0.f;
            }
        }
        return;
    }
    // this wave's images, in groups of 8 (lane half h takes images 8g + 4h + j)
    const int i_begin = split * a.ichunk, i_end = min(g.Nimg, i_begin + a.ichunk);
    const int G8 = (i_end - i_begin + 7) >> 3, gper = (G8 + KS - 1) / KS;
    const int g_begin = slot * gper, g_end = min(G8, g_begin + gper);
    const int ng = g_end - g_begin;
    const rsrc_t sr = buf_rsrc(a.Sm + grp * a.s_gs);
    const rsrc_t br = buf_rsrc(a.Bg + grp * a.b_gs);
    const int am = m0w + r;
    const bool am_ok = am < M;
    const int hw = g.Hs * g.Ws;
    const int sA_img = hw * M * 4;                  // bytes from one image to the next, small side ...
    const int sB_img = g.Hb * g.Wb * g.Cb * 4;      // ... and big side
    const int s_h = st * g.Wb * g.Cb * 4, s_w = st * g.Cb * 4;
    const int a_lane = am_ok ? (i_begin + 4 * h) * sA_img + am * 4 : BUF_OOB;
    const int b_lane = (i_begin + 4 * h) * sB_img + (((kh - g.pad) * g.Wb + (kw - g.pad)) * g.Cb + b0 + r) * 4;
    const int img_lane = i_begin + 4 * h;           // + 8 g + j: the image of this lane's element
    float nsa = 0.f, nsb = 0.f;
    if (NSIDE == 1 && am_ok) nsa = a.nrm.scale[am], nsb = a.nrm.shift[am];
    if (NSIDE == 2) nsa = a.nrm.scale[b0 + r], nsb = a.nrm.shift[b0 + r];
    const float nslope = a.nrm.slope;

    // scalar iterator: (hs, ws) position, image group g, and the two scalar offsets of (position, 8 g)
    int it_left = ng > 0 ? ng * nh * nw : 0, it_g = g_begin, it_hs = hs_lo, it_ws = ws_lo;
    auto offs = [&](int hs, int ws, int gg, int& sa, int& sb) {
        sa = (hs * g.Ws + ws) * (M * 4) + gg * 8 * sA_img;
        sb = hs * s_h + ws * s_w + gg * 8 * sB_img;
    };
    int it_sa, it_sb;
    offs(it_hs, it_ws, it_g, it_sa, it_sb);
    struct Chunk {
        float a[4], b[4];
        unsigned vm;
    };
    auto load = [&](Chunk& c) {  // one group of one position: this lane's images 8 g + 4 h + j
        const bool gv = it_left > 0;
        const int lim = gv ? i_end - it_g * 8 : -(1 << 20);  // image `img_lane + j` is valid iff img_lane + j < lim + i_begin ...
        c.vm = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool kv = img_lane + j < lim;  // (i_begin is inside img_lane; lim counts from 0)
            c.a[j] = buf_load1(sr, kv ? a_lane + it_sa + j * sA_img : BUF_OOB);
            c.b[j] = buf_load1(br, kv ? b_lane + it_sb + j * sB_img : BUF_OOB);
            c.vm |= (unsigned)((int)kv & (int)am_ok) << j | (unsigned)kv << (4 + j);
        }
        --it_left;
        ++it_g;
        it_sa += 8 * sA_img, it_sb += 8 * sB_img;
        const bool gw = it_g >= g_end;  // this position's images are done: next position of the window
        it_g = gw ? g_begin : it_g;
        it_ws += gw ? 1 : 0;
        const bool ww = it_ws > ws_hi;
        it_ws = ww ? ws_lo : it_ws;
        it_hs += ww ? 1 : 0;
        if (gw) offs(it_hs, it_ws, it_g, it_sa, it_sb);
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    auto compute = [&](const Chunk& c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float av = c.a[j], bv = c.b[j];
            if (NSIDE == 1) av = norm1(av, nsa, nsb, nslope, ((c.vm >> j) & 1u) ? 1.f : 0.f);
            if (NSIDE == 2) bv = norm1(bv, nsa, nsb, nslope, ((c.vm >> (4 + j)) & 1u) ? 1.f : 0.f);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
    };
    {
        Chunk ring[NBW];  // (see kgemm_body)
        const int nch = it_left;
        const int ntrip = nch / NBW, rem = nch - ntrip * NBW;
#pragma unroll
        for (int i = 0; i < NBW - 1; ++i) load(ring[i]);
        for (int trip = 0; trip < ntrip; ++trip) {
#pragma unroll
            for (int i = 0; i < NBW; ++i) {
                load(ring[(i + NBW - 1) % NBW]);
                compute(ring[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < NBW - 1; ++i)
            if (i < rem) compute(ring[i]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) red[(wave * 16 + i) * 64 + lane] = acc[i];
    __syncthreads();
    if (!ep_thread) return;
    const bool vec = (N & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
    const bool acc_in_place = a.accumulate && !a.slab;
#pragma unroll
    for (int w = 0; w < WT; ++w) {
        const int me = (bx * WT + w) * 32 + row;
        if (me >= M) continue;
        f32x4 v = fold_tile<KS>(red, w, t);
        float* o = dst + (long)me * N + n0 + c4;
        if (vec) {
            if (acc_in_place) v += *reinterpret_cast<const f32x4*>(o);
            *reinterpret_cast<f32x4*>(o) = v;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = acc_in_place ? o[j] + v[j] : v[j];
        }
    }
}

template <int NW, int KS, int NSIDE>
__global__ __launch_bounds__(NW * 64, 2) void kwgrad_k(KWArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[NW * 16 * 64];
    kwgrad_body<NW, KS, NSIDE>(a, smem, blockIdx.x, blockIdx.y, blockIdx.z);
}

#undef ZERO4_

// ---- host side ----------------------------------------------------------------------------------------------------------------
// MOVAE_KGEMM=0 switches the family off (A/B); movae_bench_force_kgemm(1) makes every supported shape take it (tests).
inline bool kgemm_enabled() {
    static const bool on = !(getenv("MOVAE_KGEMM") && atoi(getenv("MOVAE_KGEMM")) == 0);
    return on;
}

// reduction split inside the block: KS waves per tile.  Few tiles: eight waves on one tile (512-thread blocks: two waves per SIMD
// even at one block per CU, so one wave's loads and scalar work hide behind the other's MFMAs); many tiles: fewer slices per tile.
struct KSplit {
    int nw, ks;
};
inline KSplit choose_ks(long tiles) {
    static const int forced = getenv("MOVAE_KGEMM_KS") ? atoi(getenv("MOVAE_KGEMM_KS")) : 0;
    if (forced == 8) return KSplit{8, 8};
    if (forced == 1 || forced == 2 || forced == 4) return KSplit{4, forced};
    // (measured on the CIFAR VAE's middle layers, 256 tiles: eight waves per tile 12.1-14.0 us, four waves 11.4-12.9 us per call --
    // the second fold level and the larger block cost more than the second wave per SIMD hides; KS = 8 stays selectable by env)
    if (tiles <= 1024) return KSplit{4, 4};
    if (tiles <= 2048) return KSplit{4, 2};
    return KSplit{4, 1};
}
inline const char* ks_name(const char* const (&names)[4], KSplit k) { return names[k.ks == 8 ? 0 : (k.ks == 4 ? 1 : (k.ks == 2 ? 2 : 3))]; }

inline bool g_force_kgemm_on();  // movae_bench_force_kgemm(): every supported shape takes the family (tests)

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// shapes the family is MEANT for (the launchers below decide whether they CAN serve a call): too few 64x64 tiles to fill the chip
// without a split across blocks, and little enough work that latency, not the MFMA pipe, sets the time
inline bool small_problem(long rows, long cols, long k, long copies) {
    const long tiles64 = ((rows + 63) / 64) * ((cols + 63) / 64) * copies;
    const double flop = 2.0 * (double)rows * (double)cols * (double)k * (double)copies;
    static const long tmax = getenv("MOVAE_KGEMM_TILES64") ? atol(getenv("MOVAE_KGEMM_TILES64")) : 384;
    static const double fmax = getenv("MOVAE_KGEMM_GFLOP") ? atof(getenv("MOVAE_KGEMM_GFLOP")) * 1e9 : 4e9;
    // a short reduction (K = 288: 32 -> 64 channels on 16x16 images) leaves a wave's slice nine groups long -- the fold and the
    // epilogue then outweigh the slabs they replace (measured 18.6 vs 13.5 us per call)
    static const long kmin = getenv("MOVAE_KGEMM_KMIN") ? atol(getenv("MOVAE_KGEMM_KMIN")) : 512;
    // ... and enough 32x32 tiles to occupy half the chip: the reduction is only split INSIDE a block, so a problem with a handful
    // of tiles and a very long reduction (BetaTC-VAE's fc 32768 -> 256 at batch 32: 8 tiles) belongs to the split-K kernels
    // (measured at C5: 309 us here against 54 us there)
    const long tiles32 = ((rows + 31) / 32) * ((cols + 31) / 32) * copies;
    return tiles64 <= tmax && tiles32 >= 128 && flop <= fmax && k >= kmin;
}

template <int FORM, bool NRM>
inline void launch_k2(const KArgs& a, KSplit k, dim3 tiles, hipStream_t st) {  // tiles: (row tiles, column tiles, classes)
    if (k.ks == 8) {
        KArgs b8 = a;
        b8.fin.group = (int)(tiles.x * tiles.z);
        hipLaunchKernelGGL((kgemm_k<FORM, 8, 8, NRM>), tiles, dim3(512), 0, st, b8, RSide{}, (int)tiles.z, v2::cls_order());
        return;
    }
    dim3 grid = k.ks == 4 ? tiles : k.ks == 2 ? dim3((tiles.x + 1) / 2, tiles.y, tiles.z) : dim3((tiles.x + 3) / 4, tiles.y, tiles.z);
    KArgs b = a;
    b.fin.group = (int)(grid.x * grid.z);  // blocks per column tile (before any carried reduce's extra z layers)
    int gz;
    const RSide sd = defer_take_3d(st, &grid, &gz);
    if (k.ks == 4) hipLaunchKernelGGL((kgemm_k<FORM, 4, 4, NRM>), grid, dim3(256), 0, st, b, sd, gz, v2::cls_order());
    else if (k.ks == 2) hipLaunchKernelGGL((kgemm_k<FORM, 4, 2, NRM>), grid, dim3(256), 0, st, b, sd, gz, v2::cls_order());
    else hipLaunchKernelGGL((kgemm_k<FORM, 4, 1, NRM>), grid, dim3(256), 0, st, b, sd, gz, v2::cls_order());
}
template <int FORM>
inline void launch_k(const KArgs& a, KSplit k, dim3 tiles, hipStream_t st) {
    if (a.nrm.scale) launch_k2<FORM, true>(a, k, tiles, st);
    else launch_k2<FORM, false>(a, k, tiles, st);
}

// ---- one launch, two problems (see v2::igemm2_pair): blocks [0, nd) run a stashed kgemm input gradient (four waves, KS = 4),
// the rest the tiled weight gradient of the same layer.  The two share read-only operands only.
template <int FORM, int WBM, int WBN>
__global__ __launch_bounds__(256) void kpair_k(KArgs ka, v2::WgArgs wa, int nd, int dgx, int dgy, int wgx, int wgy, int nw, RSide sd,
                                               int wfirst) {
    constexpr int DF = smem_floats<4>(), WF = v2::WgSmem<WBM, WBN>::FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[DF > WF ? DF : WF];
    int b = blockIdx.x;
    if (b >= nd + nw) {  // behind the two: the previous layer's parked weight-gradient reduce (conv_igemm.hip: RSide)
        side_reduce(sd, b - nd - nw, smem);
        return;
    }
    if (wfirst & 1) b = b < nw ? nd + b : b - nw;  // the weight gradient's (longer) blocks are dispatched first (v2::pair_order; bit 1: cls_order)
    if (b < nd) {
        const int bx = b % dgx, r = b / dgx;
        kgemm_body<FORM, 4, 4, false>(ka, smem, bx, r % dgy, (FORM == 1 && (wfirst & 2)) ? nd / (dgx * dgy) - 1 - r / dgy : r / dgy);
    } else {
        b -= nd;
        const int bx = b % wgx, r = b / wgx;
        v2::igemm2_wgrad_body<WBM, WBN>(wa, smem, bx, r % wgy, r / wgy);
    }
}

struct KPending {
    bool active = false;
    int form = 0;
    KArgs a;
    dim3 tiles;
};
static thread_local KPending g_kpend;

template <int FORM>
inline void launch_k(const KArgs& a, KSplit k, dim3 tiles, hipStream_t st);

inline bool kpend_active() { return g_kpend.active; }
inline int kpend_flush(hipStream_t st) {
    if (!g_kpend.active) return MOVAE_OK;
    g_kpend.active = false;
    if (g_kpend.form == 0) launch_k<0>(g_kpend.a, KSplit{4, 4}, g_kpend.tiles, st);
    else launch_k<1>(g_kpend.a, KSplit{4, 4}, g_kpend.tiles, st);
    MOVAE_CHECK_LAUNCH("kgemm_k (unpaired input gradient)");
    return MOVAE_OK;
}
inline int kpend_pair(const v2::WgArgs& wa, int wgx, int wgy, int wgz, bool w64, hipStream_t st) {
    KPending& p = g_kpend;
    p.active = false;
    const int nd = p.tiles.x * p.tiles.y * p.tiles.z, nw = wgx * wgy * wgz;
    const RSide sd = defer_take(st);  // the previous layer's parked weight-gradient reduce rides behind the two problems
    const dim3 grid(nd + nw + sd.nblk);
#define MOVAE_KP(F_, BM_, BN_) hipLaunchKernelGGL((kpair_k<F_, BM_, BN_>), grid, dim3(256), 0, st, p.a, wa, nd, (int)p.tiles.x, (int)p.tiles.y, wgx, wgy, nw, sd, (v2::pair_order() == 2 ? 1 : 0) | v2::cls_order())
    if (p.form == 0 && w64) {
        MOVAE_KP(0, 64, 64);
        g_last_kernel = "kpair_k<0,64,64>";
    } else if (p.form == 0) {
        MOVAE_KP(0, 32, 128);
        g_last_kernel = "kpair_k<0,32,128>";
    } else if (w64) {
        MOVAE_KP(1, 64, 64);
        g_last_kernel = "kpair_k<1,64,64>";
    } else {
        MOVAE_KP(1, 32, 128);
        g_last_kernel = "kpair_k<1,32,128>";
    }
#undef MOVAE_KP
    MOVAE_CHECK_LAUNCH("kpair_k");
    return MOVAE_OK;
}
struct KPairInstall {
    KPairInstall() { v2::g_kpair = v2::KPairHooks{kpend_active, kpend_pair, kpend_flush}; }
};
static KPairInstall g_kpair_install;

// MOVAE_KGEMM_BN_FIN=1 (or movae_bench_kgemm_bn_fin(1)): a kgemm forward finishes the BatchNorm that follows inside its own launch
// (the tail of kgemm_body).  OFF by default: measured at C2 the six layers it applies to cost 7-9 us more per kernel under the
// profiler (the agent-scope ticket, the write-through partials read back over the fabric, the barriers) against the 4.6 us
// bn_finalize launch each replaces -- 0.818 vs 0.821 ms per step, inside the noise, C1 0.602 vs 0.599.
static int g_kgemm_bn_fin = -1;
inline bool kgemm_bn_fin() {
    static const bool env = getenv("MOVAE_KGEMM_BN_FIN") && atoi(getenv("MOVAE_KGEMM_BN_FIN")) != 0;
    return g_kgemm_bn_fin < 0 ? env : g_kgemm_bn_fin != 0;
}

// The side products / epilogue requests of the calling entry point (g_fuse), as launch_fwd2 / launch_bwd2 honour them.
// rows_all: output rows over all cotangent groups (FWD: M; BWD: pixels), ncls: output-parity classes.  False: a requested
// BatchNorm-backward epilogue cannot be served at this shape (the caller falls back to the tiled kernels, nothing claimed).
inline bool plan_side(KArgs& a, const Epilogue& ep, long rows_all, int ncls, int N) {
    const long rows_c = rows_all / ncls;
    a.tiles_c = (int)((rows_c + 31) / 32);
    a.tpg = a.tiles_c;
    const bool want_stats = g_fuse.stats && ep.act == MOVAE_ACT_NONE && !v2::g_pair_collect;
    if (g_fuse.bn_y && ep.act == MOVAE_ACT_NONE && !ep.bias) {
        const int G = g_fuse.bn_groups;
        if (rows_all % ((long)ncls * G) != 0 || (rows_c / G) % 32 != 0 || !al16(g_fuse.bn_y) || !al16(g_fuse.bn_scale) || !al16(g_fuse.bn_shift))
            return false;
        const long tpg = rows_c / G / 32, ppg = tpg * ncls;
        float* part = fuse_bn_claim(ppg, N);
        if (!part) return false;
        a.tpg = (int)tpg;
        a.bb = BnBwd{g_fuse.bn_y, g_fuse.bn_scale, g_fuse.bn_shift, g_fuse.bn_slope, part, (int)(rows_all / G), (int)ppg};
        return true;
    }
    if (want_stats) a.stats = fuse_stats_claim((long)ncls * a.tiles_c, N);
    if (a.stats && g_fuse.fin.gamma && g_fuse.fin.counter && kgemm_bn_fin() && ceil_div(N, 32) <= 64 && rows_all <= 0x7fffffffL) {
        a.fin = g_fuse.fin;  // (group: set at the launch, where the grid is known)
        a.fin.rows = (int)rows_all, a.fin.parts = ncls * a.tiles_c;
        g_fuse.fin_done = true;
    }
    if ((g_fuse.am.y || g_fuse.am.res) && ep.act == MOVAE_ACT_NONE && (!ep.bias || !g_fuse.am.y) && !a.stats && al16(g_fuse.am.y) &&
        al16(g_fuse.am.res)) {
        const int G = g_fuse.am_groups;
        const bool grouped_y = g_fuse.am.y && G > 1;
        if (rows_all % ((long)ncls * G) == 0 && (!grouped_y || (rows_c / G) % 32 == 0)) {
            a.am = g_fuse.am;
            a.am.per_group = rows_all / G * N;
            if (grouped_y) a.tpg = (int)(rows_c / G / 32);
            g_fuse.am_done = true;
        }
    }
    return true;
}

inline int launch_kfwd(const float* X, const float* W, float* Y, const Geom& g, const Epilogue& ep, int M, int K, hipStream_t st,
                       bool* handled) {
    *handled = false;
    const bool forced = g_force_kgemm_on();
    if (!(kgemm_enabled() || forced)) return MOVAE_OK;
    const long wrow = g.wlen ? g.wrow : K;
    if (g.Cr % 8 != 0 || g.Nn % 4 != 0 || (g.wlen && g.wlen % 8 != 0) || !al16(X) || !al16(W) || !al16(Y) || !al16(ep.bias) ||
        !v2::buf_span_ok((long)g.Nimg * g.Hi * g.Wi * g.Cr) || !v2::buf_span_ok((long)g.Nn * wrow + g.woff))
        return MOVAE_OK;
    if (fuse_norm() && g.Cr > NORM_MAX_C) return MOVAE_OK;
    if (!forced && !small_problem(M, g.Nn, K, 1)) return MOVAE_OK;
    KArgs a{};
    a.X = X, a.W = W, a.Y = Y, a.g = g, a.ep = ep, a.M = M, a.K = K;
    a.nrm = g_fuse.nrm;
    a.am = ActMul{nullptr, 0, 0.f, 0, 0, nullptr};
    a.fd_cr = fastdiv_make(g.Cr), a.fd_kw = fastdiv_make(g.KW), a.fd_wlen = fastdiv_make(g.wlen > 0 ? g.wlen : (K > 0 ? K : 1));
    a.fd_hw = fastdiv_make(g.Ho * g.Wo), a.fd_w = fastdiv_make(g.Wo);
    for (int c = 0; c < 4; ++c) a.fd_nb[c] = fastdiv_make(1);
    if (!plan_side(a, ep, M, 1, g.Nn)) return MOVAE_OK;
    const int tx = ceil_div(M, 32), ty = ceil_div(g.Nn, 32);
    const KSplit ks = choose_ks((long)tx * ty);
    static const char* const names[4] = {"kgemm_k<0,8,8,false>", "kgemm_k<0,4,4,false>", "kgemm_k<0,4,2,false>", "kgemm_k<0,4,1,false>"};
    static const char* const names_n[4] = {"kgemm_k<0,8,8,true>", "kgemm_k<0,4,4,true>", "kgemm_k<0,4,2,true>", "kgemm_k<0,4,1,true>"};
    g_last_kernel = a.nrm.scale ? ks_name(names_n, ks) : ks_name(names, ks);  // (as rocprofv3 prints the instantiation)
    if (v2::g_pair_collect && ks.ks == 4 && !a.nrm.scale) {  // an input gradient with its layer's weight gradient to follow: stash it
        g_kpend.active = true, g_kpend.form = 0, g_kpend.a = a, g_kpend.tiles = dim3(tx, ty, 1);
        *handled = true;
        return MOVAE_OK;
    }
    launch_k<0>(a, ks, dim3(tx, ty, 1), st);
    MOVAE_CHECK_LAUNCH("kgemm_k (fwd form)");
    *handled = true;
    return MOVAE_OK;
}

inline int launch_kbwd(const float* X, const float* W, float* Y, const Geom& g, const Epilogue& ep, hipStream_t st, bool* handled) {
    *handled = false;
    const bool forced = g_force_kgemm_on();
    if (!(kgemm_enabled() || forced)) return MOVAE_OK;
    const int s = g.stride;
    if (s < 1 || s > 2 || g.Ho % s != 0 || g.Wo % s != 0 || g.Cr % 8 != 0 || g.Nn % 4 != 0 || !al16(X) || !al16(W) || !al16(Y) ||
        !al16(ep.bias) || !v2::buf_span_ok((long)g.Nimg * g.Hi * g.Wi * g.Cr) || !v2::buf_span_ok((long)g.Cr * g.KH * g.KW * g.Nn))
        return MOVAE_OK;
    if (fuse_norm() && g.Cr > NORM_MAX_C) return MOVAE_OK;
    const int ncls = s * s;
    const long pix = (long)g.Nimg * g.Ho * g.Wo, Mc = pix / ncls;
    const long kmax = (long)ceil_div(g.KH, s) * ceil_div(g.KW, s) * g.Cr;
    if (!forced && !small_problem(Mc, g.Nn, kmax, ncls)) return MOVAE_OK;
    KArgs a{};
    a.X = X, a.W = W, a.Y = Y, a.g = g, a.ep = ep, a.M = (int)Mc, a.K = 0;
    a.nrm = g_fuse.nrm;
    a.am = ActMul{nullptr, 0, 0.f, 0, 0, nullptr};
    a.fd_cr = fastdiv_make(g.Cr), a.fd_kw = fastdiv_make(g.KW), a.fd_wlen = fastdiv_make(1);
    a.fd_hw = fastdiv_make((g.Ho / s) * (g.Wo / s)), a.fd_w = fastdiv_make(g.Wo / s);
    for (int c = 0; c < 4; ++c) {
        // taps (ta, tb) of class c: input row h = hc + qh - ta for hc in [0, Ho / s): some row meets the image for
        // ta in [qh - (Hi - 1), qh + Ho / s - 1], likewise tb -- the class's reduction runs over that window only
        const int ph = c / s, pw = c % s;
        const int kh0 = (ph + g.pad) % s, kw0 = (pw + g.pad) % s;
        const int nA = (c < ncls && kh0 < g.KH) ? (g.KH - kh0 + s - 1) / s : 0, nB = (c < ncls && kw0 < g.KW) ? (g.KW - kw0 + s - 1) / s : 0;
        const int qh = (ph + g.pad - kh0) / s, qw = (pw + g.pad - kw0) / s;
        int alo = qh - (g.Hi - 1), ahi = qh + g.Ho / s - 1, blo = qw - (g.Wi - 1), bhi = qw + g.Wo / s - 1;
        alo = alo < 0 ? 0 : alo, blo = blo < 0 ? 0 : blo;
        ahi = ahi > nA - 1 ? nA - 1 : ahi, bhi = bhi > nB - 1 ? nB - 1 : bhi;
        a.ct.ta_lo[c] = alo, a.ct.tb_lo[c] = blo;
        a.ct.nAw[c] = ahi >= alo ? ahi - alo + 1 : 0, a.ct.nBw[c] = bhi >= blo ? bhi - blo + 1 : 0;
        a.fd_nb[c] = fastdiv_make(a.ct.nBw[c] > 0 ? a.ct.nBw[c] : 1);
        if (a.ct.nBw[c] == 0) a.ct.nBw[c] = 1, a.ct.nAw[c] = 0;
    }
    if (!plan_side(a, ep, pix, ncls, g.Nn)) return MOVAE_OK;
    const int tx = ceil_div(Mc, 32), ty = ceil_div(g.Nn, 32);
    const KSplit ks = choose_ks((long)tx * ty * ncls);
    static const char* const names[4] = {"kgemm_k<1,8,8,false>", "kgemm_k<1,4,4,false>", "kgemm_k<1,4,2,false>", "kgemm_k<1,4,1,false>"};
    static const char* const names_n[4] = {"kgemm_k<1,8,8,true>", "kgemm_k<1,4,4,true>", "kgemm_k<1,4,2,true>", "kgemm_k<1,4,1,true>"};
    g_last_kernel = a.nrm.scale ? ks_name(names_n, ks) : ks_name(names, ks);
    if (v2::g_pair_collect && ks.ks == 4 && !a.nrm.scale) {
        g_kpend.active = true, g_kpend.form = 1, g_kpend.a = a, g_kpend.tiles = dim3(tx, ty, ncls);
        *handled = true;
        return MOVAE_OK;
    }
    launch_k<1>(a, ks, dim3(tx, ty, ncls), st);
    MOVAE_CHECK_LAUNCH("kgemm_k (bwd form)");
    *handled = true;
    return MOVAE_OK;
}

// weight gradient of G cotangent groups (see launch_wgrad2); colsum requests are left to the tiled kernels
inline int launch_kwgrad(const float* Sm, const float* Bg, float* const* dW, int G, long s_gs, long b_gs, const WGeom& g, int K,
                         int accumulate, void* ws, size_t ws_bytes, hipStream_t st, bool* handled) {
    *handled = false;
    const bool forced = g_force_kgemm_on();
    if (!(kgemm_enabled() || forced)) return MOVAE_OK;
    const int M = g.Cs, N = g.KH * g.KW * g.Cb;
    if (g.Cb % 32 != 0 || G > 8 || !v2::buf_span_ok((long)K * g.Cs) || !v2::buf_span_ok((long)g.Nimg * g.Hb * g.Wb * g.Cb)) return MOVAE_OK;
    // Measured against the tiled weight-gradient kernels + their reduce (C2 layers, us per call): 64->128 @8x8 26.0 vs 17.5,
    // 128->256 @4x4 16.6 vs 16.4, 256->512 @2x2 14.2 vs 10.6 -- every operand element is its own 4-byte load here (the reduction
    // index is the slow axis of both operands), eight times the load instructions of the LDS-staged 16-byte path.  Not selected by
    // the size heuristic; MOVAE_KWGRAD=1 or movae_bench_force_kgemm(1) take it.
    static const bool kw_on = getenv("MOVAE_KWGRAD") && atoi(getenv("MOVAE_KWGRAD")) != 0;
    if (!forced && !(kw_on && small_problem(M, N, K, G))) return MOVAE_OK;
    const long tiles = (long)ceil_div(M, 32) * (N / 32) * G;
    const KSplit ks = choose_ks(tiles);
    const int wt = ks.nw / ks.ks;
    // a split across blocks only where the tiles alone leave most of the chip idle AND a wave's slice would be long
    // (the reduction per tile: images x positions; a wave multiplies 2 MFMAs per 4 reduction indices ...)
    const long red = (long)K;  // pixels of the small side
    int Sp = 1;
    while (tiles * Sp / wt < 384 && red / ((long)ks.ks * Sp) > 256 && Sp < 64 && g.Nimg / (Sp * 2) >= 8) Sp *= 2;
    if (g_force_split > 0) Sp = g_force_split;
    const long stride = (long)M * N;
    if (Sp > 1 && (!ws || (size_t)stride * sizeof(float) * Sp * G > ws_bytes)) Sp = 1;
    const int ichunk = ceil_div(ceil_div(g.Nimg, Sp), 8) * 8;
    Sp = ceil_div(g.Nimg, ichunk);
    KWArgs a{};
    a.Sm = Sm, a.Bg = Bg, a.slab = Sp > 1 ? static_cast<float*>(ws) : nullptr, a.g = g, a.ichunk = ichunk, a.Sp = Sp;
    a.accumulate = accumulate, a.s_gs = s_gs, a.b_gs = b_gs, a.slab_stride = stride;
    for (int i = 0; i < 8; ++i) a.tab.p[i] = i < G ? dW[i] : nullptr;
    a.nrm = g_fuse.nrm, a.nrm_side = g_fuse.nrm_side;
    if (int rc = v2::flush_pending(st)) return rc;  // (a stashed dgrad of the paired path goes first, on its own)
    const dim3 grid(ceil_div(ceil_div(M, 32), wt), N / 32, G * Sp);
    const int nside = a.nrm.scale ? a.nrm_side : 0;
#define MOVAE_KW(NW_, KS_)                                                                               \
    do {                                                                                                 \
        if (nside == 1) hipLaunchKernelGGL((kwgrad_k<NW_, KS_, 1>), grid, dim3(NW_ * 64), 0, st, a);      \
        else if (nside == 2) hipLaunchKernelGGL((kwgrad_k<NW_, KS_, 2>), grid, dim3(NW_ * 64), 0, st, a); \
        else hipLaunchKernelGGL((kwgrad_k<NW_, KS_, 0>), grid, dim3(NW_ * 64), 0, st, a);                 \
    } while (0)
    if (ks.ks == 8) MOVAE_KW(8, 8);
    else if (ks.ks == 4) MOVAE_KW(4, 4);
    else if (ks.ks == 2) MOVAE_KW(4, 2);
    else MOVAE_KW(4, 1);
#undef MOVAE_KW
    MOVAE_CHECK_LAUNCH("kwgrad_k");
    static const char* const names[3][4] = {{"kwgrad_k<8,8,0>", "kwgrad_k<4,4,0>", "kwgrad_k<4,2,0>", "kwgrad_k<4,1,0>"},
                                            {"kwgrad_k<8,8,1>", "kwgrad_k<4,4,1>", "kwgrad_k<4,2,1>", "kwgrad_k<4,1,1>"},
                                            {"kwgrad_k<8,8,2>", "kwgrad_k<4,4,2>", "kwgrad_k<4,2,2>", "kwgrad_k<4,1,2>"}};
    g_last_kernel = ks_name(names[nside], ks);
    if (Sp > 1)
        for (int i = 0; i < G; ++i)
            if (int rc = launch_reduce(a.slab + (long)i * Sp * stride, dW[i], stride, Sp, N, nullptr, 0, 0.f, accumulate, st)) return rc;
    *handled = true;
    return MOVAE_OK;
}

}  // namespace kg
