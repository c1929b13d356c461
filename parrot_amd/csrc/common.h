// Shared device/host helpers for the parrot_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define PH_CHECK(x)                                   \
    do {                                              \
        hipError_t e__ = (x);                         \
        if (e__ != hipSuccess) return (int)e__;       \
    } while (0)

// First statement of every extern "C" entry point: the launch wrappers report `hipGetLastError()`, which is sticky
// per thread across ALL users of the runtime in the process -- an error left behind by somebody else's call (seen
// on the GPU box: 100 after PyTorch's own start-up probing) must not be blamed on our first launch.
#define PH_ENTRY() (void)hipGetLastError()

// Error codes returned by the C-ABI on bad arguments (hipError_t values are
// returned unchanged for runtime failures).
#define PH_ERR_BADARG 10001
#define PH_ERR_UNSUPPORTED 10002

// Eight f32 -> eight bf16, round to nearest even (v_cvt_pk_bf16_f32 on gfx950): the operand rounding of the
// bf16 compute mode (weights and activations are rounded where they enter an MFMA, accumulation stays f32).
__device__ __forceinline__ bf16x8 ph_bf16x8(const f32x4& lo, const f32x4& hi) {
    const f32x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_convertvector(v, bf16x8);
}

__device__ __forceinline__ float ph_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// Sum over the 64 lanes of a wave, result broadcast to every lane.  Uses DPP row shifts / row
// broadcasts (a few cycles each) instead of ds_bpermute-based shuffles (an LDS round trip each): the
// attention kernels are made of dozens of such reductions.
__device__ __forceinline__ float wave_sum(float v) {
#define PH_DPP_ADD(ctrl, rmask)                                                                      \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xf, true))
    PH_DPP_ADD(0x111, 0xf);  // row_shr:1   (inclusive prefix sums inside each row of 16 lanes)
    PH_DPP_ADD(0x112, 0xf);  // row_shr:2
    PH_DPP_ADD(0x114, 0xf);  // row_shr:4
    PH_DPP_ADD(0x118, 0xf);  // row_shr:8   -> lane 15 of every row holds the row total
#undef PH_DPP_ADD
    // row_bcast:15 into rows 1 and 3, then row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
