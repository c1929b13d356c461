// Persistent phase machine: the decoder scan as ONE resident kernel (gfx950).
//
// Replaces the thousands of per-step launches of the wavefront schedules (plans.hip) for the scan of
// Parrot.compute_cost (reference model.py:651-737).  One workgroup per CU stays resident for the whole window; a
// tick = n_slots phases separated by grid barriers; in a phase every workgroup runs up to maxu work units from a
// host-built table.  A unit is a 16-column tile of a recurrent-step GEMM over all batch rows with the GRU gate math
// fused (the same algebra as sk_kernel's epilogues, Blocks GatedRecurrent / sampleRNN/lib/ops.py:364-393), or the
// GMM-window attention of one batch row (model.py:664-690).  What the launches could not do:
//   * weights are STATIONARY: a unit's [K,16] weight slab is copied into the CU's LDS once per window (up to 144 KB per
//     CU) and every step reads it from there; only what does not fit is streamed from the fragment-major copies;
//   * activations travel between workgroups in fragment-major slabs (1 KB blocks in MFMA operand order), stored
//     write-through (sc1) by the producing epilogue and read with sc1 loads by the consumers, so the only inter-phase
//     cost is the barrier (XCD-hierarchical, ~2.4 us measured, tools/persist_probe.hip) -- no launch, no cold start;
//   * the projections of lower-layer outputs into a layer (Fork h{j}_to_h{l}, inp_to_h{l}; model.py:692-722) run as
//     separate units one tick ahead of the layer's recurrent units, so every phase's longest dependent GEMM has
//     K = H instead of K = (l+1) H + E.
#pragma once
#include "common.h"

enum { PM_MAXENT = 12,   // unit descriptors per workgroup: n_slots * maxu <= PM_MAXENT (training 3 x 3, decode 9 x 1)
       PM_MAXSLOTS = 9, PM_THREADS = 512, PM_MAXINIT = 12, PM_MAXDST = 6, PM_MAXWDST = 8, PM_MAXFILL = 16 };
enum { PM_NONE = 0, PM_GEMM = 1, PM_ATT = 2 };
enum { PM_EPI_LINEAR = 0, PM_EPI_GATES = 1, PM_EPI_CAND = 2 };

// LDS map (floats): resident weights | scratch shared by the split-K reduction (8 partial 16x16 tiles, rows padded
// to 20 floats) and the attention row (never live at the same time) | the workgroup's unit table | misc.
enum {
    PM_LDS_W = 36864,                 // 144 KB = 2304 K-rows of a 16-column tile
    PM_LDS_RED = 8 * 16 * 20,         // 10 KB
    PM_LDS_UNITS = 1280,              // 5 KB: PM_MAXENT unit descriptors
    PM_LDS_MISC = 64,                 // census, flags, phase timers
    PM_LDS_FLOATS = PM_LDS_W + PM_LDS_RED + PM_LDS_UNITS + PM_LDS_MISC,
    PM_ATT_MAXU = 800,                // context length limit of the in-kernel attention (208 + U + 512 + 512 floats <= PM_LDS_RED)
    PM_ATT_MAXA = 32,
};

// Row-major operand at step t: p + t * st (floats), leading dimension ld; p already points at the unit's first column.
struct PmRM {
    float* p;
    long long st;
    int ld, pad;
};

// Fragment-major activation slab of one step: [MB row blocks][K/16 chunks][256 floats]; block (rb, c) holds, for lane
// l = kk*16 + r, the four values X[16 rb + r][16 c + 4 kk + 0..3] -- the A operand of four 16x16x4 MFMAs.
// Every unit reads ONE slab (its whole K range, e.g. [h0[t] ; w[t]] for the layer-0 gates): the producers write their
// tiles into every slab that contains them (PmDst), so the consumer's K loop needs no segment lookup.
// Slabs are addressed as BYTE offsets from PmProgram::fm_base (one buffer resource for the whole region, < 4 GB).
struct PmDst {
    unsigned off, st;   // slab of step t = fm_base + off + t * st
    int nch, chunk;     // chunks per row block of that slab / chunk of the producer's first column
};

struct PmUnit {
    int kind, lag, K, w_lds;     // w_lds >= 0: float offset of the resident weight slab in LDS; -1: stream from W
    unsigned a_off, a_st;        // the activation slab the unit reads
    int a_nch, a_c0;             // chunks per row block of that slab / the unit's first chunk in it (K/16 chunks are read)
    const float* W;              // fragment-major weights of the unit: [K/16][256] floats
    int epi, M, rtile, row;      // rtile: GATES tile of the reset gate; row: batch row of an ATT unit
    const float* bias;           // 16 floats (the tile's columns) or null
    PmRM add[4];                 // additive pre-activation inputs (p == null: unused)
    PmRM e0, e1;                 // GATES (r tile): e0 = h_prev;  CAND: e0 = h_prev, e1 = z
    PmRM out, o1, o2;            // LINEAR: out;  GATES: o1 = z | o2 = r, out = r*h_prev;  CAND: o1 = c, out = h_new
    int ndst, pad2;              // fragment-major copies of `out` for the consuming units
    PmDst dst[PM_MAXDST];
    // Round 5 (decode, batch <= 16): the attention projection folded into layer 0's candidate units.  pw[jt] = the
    // fragment-major block of the padded projection matrix Watt [H, 32] for this unit's 16 state columns and output
    // column tile jt; the finalising wave multiplies the h_new tile it holds by them (8 MFMAs) and publishes the
    // [B, 32] partial sums at pp (row-major, write-through) -- the attention row then adds H / 16 partials of 32 values
    // instead of reading its state row and the 120 KB projection matrix.  pw[0] == null: no fold.
    const float* pw[2];
    PmRM pp;
};

struct PmAtt {
    PmRM h1;                      // layer-0 state history [T+1,B,H]: row-major, p = slot 0
    const float* WattT; const float* batt; const float* ctx;
    float* kappa;                 // [T+1,B,A]
    float* a; float* b;           // [T,B,A]
    float* phi;                   // [T,B,U]
    float* w;                     // [T+1,B,E] row-major
    int nwdst, pad3;              // fragment-major copies of w[t+1] (slab of step t+1 for layer 0, step t above)
    PmDst wdst[PM_MAXWDST];       // off already points at the slab the value of step t goes to (t * st is added)
    int* sup;                     // [T,B,2] or null
    const float* pp;              // (round 5) per-tile partial sums of the projection, [T][H / 16][B][32], or null
    long long pp_st;              // floats per step
    int B, H, A, U, E, att_type, dense, pad;
    float eps, alignment, sharpening, timing;
};

struct PmInit {  // prologue: row-major [M,K] (ld) -> chunks [chunk, chunk + K/16) of a fragment-major slab
    const float* src;
    unsigned dst_off;
    int nch, chunk, ld, K, pad;
};

// Dataflow mode (PmProgram::dataflow): no grid barriers.  Every buffer a unit reads from another workgroup is write-once
// per launch; pm_launch fills those buffers (PmFill) with PM_EMPTY, a NaN payload arithmetic never produces, producers
// overwrite 16-byte slots with single write-through stores, and consumers re-read a slot until it is full.  A unit's
// inputs always come from units at earlier (tick, slot) positions and every workgroup walks its units in (tick, slot)
// order, so some unit can always run.  A hand-off then costs one store-to-load latency instead of store acknowledgement
// + barrier + load.
struct PmFill {
    void* p;
    long long bytes;
};

struct PmProgram {
    int T, n_ticks, nwg, MB, M, ninit, n_slots, maxu;  // a tick = n_slots phases of up to maxu units per workgroup
    int dataflow, nfill;
    PmFill fill[PM_MAXFILL];
    const PmUnit* units;  // device: [n_slots][nwg][maxu]
    unsigned* sync;       // device: PM_SYNC_WORDS + PM_DBG_WORDS unsigned, zeroed before every launch
    float* fm_base;       // start of the fragment-major slab region (all PmDst / a_off offsets are relative to it)
    PmAtt att;
    PmInit init[PM_MAXINIT];
};

enum { PM_SYNC_WORDS = 1024, PM_DBG_WORDS = 256 * 24 * 2 };  // dbg: per workgroup 24 x u64: work[9], wait[9] per slot, 4 gemm stages (100 MHz ticks)
// word offsets inside `sync` (128 B apart)
enum { PM_S_XCNT = 0, PM_S_XGEN = 256, PM_S_TOP = 512, PM_S_CENSUS = 544, PM_S_TOTAL = 800, PM_S_ABORT = 832,
       PM_S_STICKY = 992 };  // words >= PM_S_STICKY survive pm_launch's clearing: [PM_S_STICKY] != 0 = some launch gave up
#define PM_EMPTY 0x7FC0DEADu

int pm_launch(const PmProgram& prog, hipStream_t stream);
int pm_status(const PmProgram& prog);  // synchronises; 0, or non-zero when a launch on this program's workspace gave up
int pm_max_workgroups();  // number of workgroups the machine runs with on this device (one per CU, <= 256)
