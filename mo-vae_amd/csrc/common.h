// Shared helpers for the gfx950 kernels (wave64, fp32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>

#include "../../include/movae.h"

#define MOVAE_OK 0
#define MOVAE_EINVAL (-1)
#define MOVAE_EUNSUPPORTED (-2)
#define MOVAE_ELAUNCH (-3)

void movae_set_error(const char* fmt, ...);

#define MOVAE_CHECK_ARG(cond, ...)            \
    do {                                      \
        if (!(cond)) {                        \
            movae_set_error(__VA_ARGS__);     \
            return MOVAE_EINVAL;              \
        }                                     \
    } while (0)

#define MOVAE_CHECK_LAUNCH(name)                                                     \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) {                                                      \
            movae_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
            return MOVAE_ELAUNCH;                                                    \
        }                                                                            \
    } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    switch (act) {
        case MOVAE_ACT_LRELU: return v > 0.f ? v : v * slope;
        case MOVAE_ACT_RELU: return v > 0.f ? v : 0.f;
        case MOVAE_ACT_TANH: return tanhf(v);
        case MOVAE_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        default: return v;
    }
}

// derivative of the activation expressed through its OUTPUT value
__device__ __forceinline__ float act_grad_from_out(float o, int act, float slope) {
    switch (act) {
        case MOVAE_ACT_LRELU: return o > 0.f ? 1.f : slope;
        case MOVAE_ACT_RELU: return o > 0.f ? 1.f : 0.f;
        case MOVAE_ACT_TANH: return 1.f - o * o;
        case MOVAE_ACT_SIGMOID: return o * (1.f - o);
        default: return 1.f;
    }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- in-launch "last block finishes" hand-off ---------------------------------------------------------
// Two-stage reductions write per-block partials and let the LAST block to arrive fold them, instead of a
// second launch (a dependent kernel boundary costs ~1.5-2 us plus the tiny kernel itself).  Placement
// independent: every storing wave drains its stores, one lane publishes with an agent-scope release before
// the ticket, the last arriver takes an agent-scope acquire before any wave of its block reads the partials
// (cdna_hip_programming.md Guideline 16, counter form).  The counter word lives in the caller's workspace
// header (MOVAE_WS_HEADER_BYTES, zero on first use) and is re-armed to zero by the last block.
#define MOVAE_WS_HEADER_BYTES 4096

// plain-scratch users skip the counter header so it stays zero between the launches that use it
#define MOVAE_WS_SCRATCH(ws, ws_bytes)                                         \
    do {                                                                       \
        if ((ws) && (ws_bytes) > (size_t)MOVAE_WS_HEADER_BYTES) {              \
            (ws) = static_cast<char*>(ws) + MOVAE_WS_HEADER_BYTES;             \
            (ws_bytes) -= MOVAE_WS_HEADER_BYTES;                               \
        } else {                                                               \
            (ws) = nullptr;                                                    \
            (ws_bytes) = 0;                                                    \
        }                                                                      \
    } while (0)

__device__ __forceinline__ bool arrive_last(unsigned* counter, unsigned nblk) {
    __shared__ unsigned s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned last = ticket == nblk - 1 ? 1u : 0u;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_last = last;
    }
    __syncthreads();
    return s_last != 0;
}

// block-wide sum for 256-thread blocks; result valid in every thread
// Column fold of a 256-thread block whose thread t owns column cl = t % CQB of row group rg = t / CQB (CQB a power of two):
// on return the threads with rg == 0 hold, in v[], the sums over all row groups of their column.  Row groups that share a
// wave are folded with shuffles first, so the serial part is at most 4 LDS reads per value (the naive form made the
// 256 / CQB row groups a serial loop on CQB threads -- 32 deep for a 32-channel tensor).  sh: NV * 256 doubles.
template <int NV>
__device__ __forceinline__ void fold_columns_256(double (&v)[NV], double* sh, int CQB) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (CQB < 64) {
        for (int off = CQB; off < 64; off <<= 1)
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q] += __shfl_xor(v[q], off, 64);
        if (lane < CQB)
#pragma unroll
            for (int q = 0; q < NV; ++q) sh[q * 256 + wave * 64 + lane] = v[q];
        __syncthreads();
        if (t < CQB)
#pragma unroll
            for (int q = 0; q < NV; ++q)
                v[q] = (sh[q * 256 + t] + sh[q * 256 + 64 + t]) + (sh[q * 256 + 128 + t] + sh[q * 256 + 192 + t]);
    } else {
#pragma unroll
        for (int q = 0; q < NV; ++q) sh[q * 256 + t] = v[q];
        __syncthreads();
        if (t < CQB) {
            const int RG = 256 / CQB;
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                double s = v[q];
                for (int i = 1; i < RG; ++i) s += sh[q * 256 + i * CQB + t];
                v[q] = s;
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ double block_sum_256(double v, double* sh /* >= 4 doubles */) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
