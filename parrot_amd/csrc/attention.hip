// Graves GMM-window location attention, forward and backward, one decoder timestep per launch.
// Restates reference model.py:664-690 (training) and :931-958 (sampling: sharpening / timing
// coefficients).  attention_type 0 = "graves", 1 = "softmax" (model.py:666-669, 680-687).
//
//   p = h1 @ Watt + batt  (Watt is stored transposed, WattT [3A,H], so both dot-product operands are
//                            contiguous along H)
//                  p = [alpha_hat | beta_hat | kappa_hat], each [A]
//   a = exp(alpha_hat) + eps   (graves)  |  softmax(alpha_hat) + eps   (softmax)
//   b = exp(beta_hat) * sharpening + eps
//   kappa = kappa_prev + alignment * exp(kappa_hat) / timing
//   phi[u] = sum_A a * exp(-b (kappa - u)^2)                               (graves)
//          = 0.3989422917366028 * sum_A a * sqrt(b) * exp(-b/2 (kappa - u)^2)   (softmax)
//   w[e]   = sum_u phi[u] * ctx[b, u, e]
//
// HBM-bound part: the per-step read of ctx[B,U,E] (13.1 MB at B=64,U=200,E=256).  Forward runs
// ESPLIT workgroups per batch row, each reducing over all U for a slice of E with coalesced
// row-segment reads; the tiny projection and the window are recomputed per slice (L2-resident
// inputs).  Reductions over A happen in-lane, reductions over U / H use wave shuffles + LDS.
#include "attention.h"
#include "att_fwd_body.h"
#include "att_bwd_body.h"
#include "elementwise.h"

#include <stdlib.h>
#include <type_traits>
#include <string.h>

namespace {

__global__ __launch_bounds__(ATT_THREADS) void att_fwd_kernel(const AttFwdArgs g) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    att_fwd_block<ATT_THREADS, ATT_PROJ_UNROLL>(g, blockIdx.x, blockIdx.y, sm);
}

__global__ __launch_bounds__(ATTB_THREADS) void att_bwd_kernel(const AttBwdArgs g) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    att_bwd_row(g, blockIdx.x, sm);
}

template <class SA>
__global__ __launch_bounds__(ATTB_THREADS) void att_state_bwd_kernel(const AttBwdArgs g, const SA sa, int att_rows,
                                                                      int l0_chain) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    att_state_bwd_block(g, sa, att_rows, l0_chain, blockIdx.x, sm);
}

}  // namespace


int att_fwd_check(AttFwdArgs& g) {
    {   // read per launch (launches happen once, at graph capture): tests toggle it between two plans
        const char* e = getenv("PARROT_ATT_DENSE");
        g.dense = e ? atoi(e) : 0;
    }
    if (g.A < 1 || g.A > ATT_MAXA || g.B < 1 || g.U < 1 || g.E < 1 || g.esplit < 1) return PH_ERR_BADARG;
    if (att_fwd_lds(g.U) > 160 * 1024) return PH_ERR_UNSUPPORTED;
    return 0;
}

int att_fwd_launch(const AttFwdArgs& gin, hipStream_t stream) {
    AttFwdArgs g = gin;
    const int rc = att_fwd_check(g);
    if (rc != 0) return rc;
    const size_t lds = att_fwd_lds(g.U);
    hipLaunchKernelGGL(att_fwd_kernel, dim3(g.B, g.esplit), dim3(ATT_THREADS), lds, stream, g);
    return (int)hipGetLastError();
}

int att_bwd_launch(const AttBwdArgs& gin, hipStream_t stream) {
    AttBwdArgs g = gin;
    if (g.A < 1 || g.A > ATT_MAXA || g.B < 1 || g.U < 1 || g.E < 1) return PH_ERR_BADARG;
    const size_t lds = att_bwd_lds(g.U, g.E);
    if (lds > 160 * 1024) return PH_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(att_bwd_kernel, dim3(g.B), dim3(ATTB_THREADS), lds, stream, g);
    return (int)hipGetLastError();
}

template <class SA>
static int att_state_bwd_launch_t(const AttBwdArgs* gin, const SA& sa, int l0_chain, hipStream_t stream) {
    AttBwdArgs g{};
    int att_rows = 0;
    size_t lds = 0;
    if (gin) {
        g = *gin;
        if (g.A < 1 || g.A > ATT_MAXA || g.B < 1 || g.U < 1 || g.E < 1 || g.B != sa.B) return PH_ERR_BADARG;
        lds = att_bwd_lds(g.U, g.E);
        if (lds > 160 * 1024) return PH_ERR_UNSUPPORTED;
        att_rows = g.B;
    } else {
        memset(&g, 0, sizeof(g));
        l0_chain = -1;
    }
    if (sa.nchain < 0 || sa.nchain > 4 || l0_chain >= sa.nchain) return PH_ERR_BADARG;
    const int others = sa.nchain - ((att_rows > 0 && l0_chain >= 0) ? 1 : 0);
    const int blocks = att_rows + others * sa.B;
    if (blocks < 1) return 0;
    hipLaunchKernelGGL((att_state_bwd_kernel<SA>), dim3(blocks), dim3(ATTB_THREADS), lds, stream, g, sa, att_rows,
                       l0_chain);
    return (int)hipGetLastError();
}

int att_state_bwd_launch(const AttBwdArgs* gin, const GruStateBwdArgs& sa, int l0_chain, hipStream_t stream) {
    return att_state_bwd_launch_t(gin, sa, l0_chain, stream);
}
int att_state_bwd_launch(const AttBwdArgs* gin, const LstmStateBwdArgs& sa, int l0_chain, hipStream_t stream) {
    return att_state_bwd_launch_t(gin, sa, l0_chain, stream);
}

int att_default_esplit(int B, int E) {
    // aim for >= 256 workgroups while keeping slices >= 32 columns
    // every slice recomputes the projection (reads WattT, 3A*H floats), so more slices = more L2
    // traffic; fewer = fewer busy CUs.
    int es = 1;
    while (B * es < 128 && E / (es * 2) >= 32) es *= 2;
    return es;
}
