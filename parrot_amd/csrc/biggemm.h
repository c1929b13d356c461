// Internal interface of the LDS-tiled f32 MFMA GEMM used for everything that is batched over
// time: readouts / output projection (reference model.py:739-755), teacher-forcing feedback and
// speaker projections (model.py:580-627), the deferred weight gradients of the scan, and the
// SampleRNN training MLP (three_tier.py:452-515).
#pragma once
#include "common.h"

struct BgArgs {
    const float* A;  // element (m,k) at A[m*sam + k*sak]; one of sam/sak must be 1
    const float* B;  // element (k,n) at B[k*sbk + n*sbn]; one of sbk/sbn must be 1
    float* C;        // row-major [M,N], leading dimension ldc
    const float* bias;  // [N] or null, added once
    int M, N, K;
    long long sam, sak, sbk, sbn;
    int ldc;
    long long batchA, batchB, batchC;  // element strides between batch entries
    int nbatch;
    int splitk;      // >1: K is split over grid.y; the slices' partial tiles go to `ws` and are summed in slice
                     // order by a second kernel (deterministic)
    float* ws;       // [nbatch * splitk, M, N] partial sums (required when splitk > 1)
    int accumulate;  // C += result (C must hold valid data; with splitk>1 always accumulates)
    float alpha;
    int act;         // SkAct-compatible activation, only when splitk == 1
    const float* gate;  // optional [M, N] (leading dimension ldg): the result is kept where gate > 0 and zeroed elsewhere --
    int ldg;            // ReLU backward through the saved activation, fused into the dx product (nbatch == 1 only)
    int bf16;        // 1: operands rounded to bf16 on their way into LDS, v_mfma_f32_32x32x16_bf16, f32 accumulation
                     // 2: A and B point at bf16 data (strides in bf16 elements), any "one stride is 1" layout: bgh_kernel
                     // 3: f32 operands split into three bf16 terms each inside the kernel, six bf16 MFMAs per block,
                     //    f32-grade result (bgs_kernel; set by gemm_impl for PARROT_PRECISION_BF16X3)
};

int bg_launch(const BgArgs& a, hipStream_t stream);
void bg_tile_shape(int bf16, int& bm, int& bn);  // macro tile bg_launch will use (split-K heuristics)
int bg_to_bf16_launch(const float* x, void* y, long long n, hipStream_t stream);  // f32 -> bf16 copy (RNE), n % 8 == 0
int bg_reduce_launch(const BgArgs& a, hipStream_t stream);  // second pass of the deterministic split-K

// Operand precision of parrot_gemm's batched path: the process-wide mode (parrot_set_gemm_precision) unless a scan
// plan running on this thread pins its own (a plan built for bf16 operands keeps them whatever the caller's mode is).
struct BgPrecisionScope {
    explicit BgPrecisionScope(int bf16);  // 1 / 0: pin bf16 / f32 operands; < 0: keep what is in force
    ~BgPrecisionScope();
    int saved;
};
