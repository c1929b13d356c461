// Shared device/host helpers for the parrot_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define PH_CHECK(x)                                   \
    do {                                              \
        hipError_t e__ = (x);                         \
        if (e__ != hipSuccess) return (int)e__;       \
    } while (0)

// Error codes returned by the C-ABI on bad arguments (hipError_t values are
// returned unchanged for runtime failures).
#define PH_ERR_BADARG 10001
#define PH_ERR_UNSUPPORTED 10002

__device__ __forceinline__ float ph_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
