// Internal interface of the GMM-window attention step kernels (reference model.py:664-690, 931-958).
#pragma once
#include "common.h"

// Development (-DSK_TIMERS, `python -m parrot_amd.build --timers`, tools/att_timing.py): every wave of an attention row block
// stamps its phases (100 MHz wall clock) into att_timer_buf[dir][row][wave][16]; dir 0 = forward, 1 = backward.
#ifdef SK_TIMERS
static __constant__ unsigned long long* att_timer_buf = nullptr;  // one copy per translation unit (set by sk_debug_set_timers)
#define ATT_STAMP(dir, row, i)                                                                                       \
    do {                                                                                                             \
        if (att_timer_buf && (threadIdx.x & 63) == 0 && (row) < 64)                                                  \
            att_timer_buf[((((dir) * 64 + (row)) * 16) + (threadIdx.x >> 6)) * 16 + (i)] = wall_clock64();           \
    } while (0)
#else
#define ATT_STAMP(dir, row, i) do { } while (0)
#endif

struct AttFwdArgs {
    const float* h1;  int ldh;     // [B,H] layer-1 state of this step
    const float* WattT;            // [3A,H] (alpha | beta | kappa rows): h1_to_att weights, transposed
    const float* batt;             // [3A] or null
    const float* kappa_prev;       // [B,A]
    const float* ctx;              // [B,U,E] encoder output * labels_mask
    float* a_out; float* b_out; float* kappa_out;  // [B,A]
    float* phi_out;                // [B,U]
    float* w_out; int ldw;         // [B,E]
    int B, H, A, U, E, esplit, att_type;
    float eps, alignment, sharpening, timing;
    int* sup_out;  // [B,2] or null: first / last context position with phi != 0 (saved for the backward step)
    int dense;  // set by att_fwd_launch (PARROT_ATT_DENSE=1): read all U context rows, also those with phi == 0
    unsigned* flag;  // null, or an arrival counter: w_out is stored write-through (sc1) and every workgroup adds 1 once
                     // its slice is out, so that workgroups of the SAME launch can consume w (skinny.hip, SkJob::wait_flag)
};

struct AttBwdArgs {
    float* dw; int lddw;           // [B,E] gradient wrt w_t (in; the total is written back when dw2 != null)
    const float* dw2;              // [B,E] optional second share of the gradient (layer-0 path) or null
    const float* dw3;              // [B,E] optional further shares (the K parts of the split backward products) or null
    const float* dw4;
    const float* dw5;
    const float* dw6;
    const float* ctx;              // [B,U,E]
    const float* a; const float* b; const float* kappa; const float* kappa_prev;  // [B,A]
    const float* WattT;            // [3A,H]
    float* dkappa;                 // [B,A] in: carry from step t+1, out: carry to step t-1
    float* dp_out;                 // [B,3A] gradient wrt the projection (for deferred dWatt)
    float* dh1; int lddh;          // [B,H] accumulated (+=)
    int B, H, A, U, E, att_type;
    float eps;
    const int* sup;  // [B,2] or null: the forward step's window support
};

int att_fwd_launch(const AttFwdArgs& g, hipStream_t stream);
// Validates the arguments and sets `dense` from PARROT_ATT_DENSE (what att_fwd_launch does before it launches).
int att_fwd_check(AttFwdArgs& g);
int att_bwd_launch(const AttBwdArgs& g, hipStream_t stream);
struct GruStateBwdArgs;
// att (or null) + the GRU state backward of all chains in one launch; l0_chain = index of layer 0's chain or -1.
int att_state_bwd_launch(const AttBwdArgs* g, const GruStateBwdArgs& sa, int l0_chain, hipStream_t stream);
struct LstmStateBwdArgs;
int att_state_bwd_launch(const AttBwdArgs* g, const LstmStateBwdArgs& sa, int l0_chain, hipStream_t stream);
int att_default_esplit(int B, int E);
