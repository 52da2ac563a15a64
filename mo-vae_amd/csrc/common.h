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

// block-wide sum for 256-thread blocks; result valid in every thread
__device__ __forceinline__ double block_sum_256(double v, double* sh /* >= 4 doubles */) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
