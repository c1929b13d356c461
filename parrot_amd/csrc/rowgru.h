// Row-owning GRU scan for narrow layers (rowgru.hip): one launch per direction for the whole sequence.
#pragma once
#include "common.h"

struct RowGruChain {
    const float* Wg_f; const float* Wc_f;  // fragment-major copies for x . W      (sk_tile_weights mode 0)
    const float* Wg_r; const float* Wc_r;  // fragment-major copies for dy . W^T   (mode 1)
    const float* inputs;       // [T,B,H]  or null
    const float* gate_inputs;  // [T,B,2H] or null
    float* h;                  // [T+1,B,H]
    float* z; float* r; float* rh; float* c;  // [T,B,H] saved activations (indexed by input time t)
    float* dh;                 // [T+1,B,H] in: gradient per slot from the consumers, out: total
    float* dG; float* dC;      // [T,B,2H], [T,B,H]
    int reverse, pad;
};

struct RowGruArgs {
    RowGruChain chain[4];
    const float* mask;  // [T,B] or null
    int T, B, H, nchain;
};

bool rowgru_supported(int T, int B, int H, int nchain);
int rowgru_fwd_launch(const RowGruArgs& g, hipStream_t stream);
int rowgru_bwd_launch(const RowGruArgs& g, hipStream_t stream);
