// Persistent-thread sample-level kernel of the SampleRNN generation loop (three_tier.py:452-515, 809-832): all
// FRAME_SIZE sample steps between two frame-tier steps in ONE launch.  See sr_persist.hip.
#pragma once
#include "common.h"

struct SrpArgs {
    const int* tbase; int toff;        // first sample index of this launch: tbase[0] + toff
    int* samples; int len;             // [B][len] sample history (read: the FS samples before t0; written: t0 .. t0+nsteps-1)
    int B, D, Q, FS, nsteps;
    // Everything in front of L2's ReLU composed through W2 by the plan (samplernn.hip, SrPlan::make_composed):
    const float* t2tbl;                // [FS][Q][D] = emb_tbl[pos] . W2 (Embedding folded with L1_PrevSamples, then with L2)
    const float* frame_out; int ldf;   // [B][ldf]; step i adds columns [i*D, (i+1)*D) of h . (Wout . W2) + bout . W2 + b2
    const float* W3; const float* b3;  // [D][D], [D]
    const float* W4; const float* b4;  // [D][Q], [Q]
    float* logits;                     // [B][Q]: logits of the launch's last step (or null)
    // Optional: the frame tier's input for the NEXT frame (three_tier.py:398-411), computed by every team for its own
    // streams from the samples it has just produced: next_in[b][d] = next_bias[d] + next_add[b][d] + sum_i xf_i * next_Win[i][d],
    // xf_i = (sample / (Q/2) - 1) * 2 of the launch's last FS samples.  next_in == null: not computed.
    // Round 5: next_n (0 = D) is the width of that product -- with next_Win = Win . U ([FS, 3D], composed by the plan) and
    // next_add = the big-tier share of the pre-activations, the kernel leaves the frame tier's additive gate / candidate
    // inputs themselves ([B, 3D]) and the tier's two step GEMMs walk K = D instead of 2 D.  next_bias may be null.
    float* next_in; const float* next_Win; const float* next_bias; const float* next_add; int next_ld_add, next_n;
    // With next_n = 3 D and next_gpre = h . Wg of the next frame ([B, 2D], update | reset: made beside the output projection,
    // it does not depend on the new samples) the kernel also finishes that frame's gates: next_z = sigm(gpre_z + in_z),
    // next_rh = sigm(gpre_r + in_r) * next_h; only the candidate's third of next_in is written.  Null: not computed.
    const float* next_gpre; const float* next_h; float* next_z; float* next_rh;
    float* ws;                         // srp_ws_floats() floats, prepared once by srp_init_ws
    float temperature; int pad;
    unsigned long long seed;
};

bool srp_eligible(int B, int D, int Q, int FS);
long long srp_ws_floats(int D, int Q);
int srp_init_ws(float* ws, int D, int Q);  // once per plan: sync words zero, hand-off slots EMPTY
int srp_prepare(int D);  // once per process and width, outside stream capture (raises the kernel's LDS limit)
int srp_launch(const SrpArgs& a, hipStream_t stream);
// 0 when no launch on this workspace has timed out / mis-teamed so far
int srp_status(const float* ws);
