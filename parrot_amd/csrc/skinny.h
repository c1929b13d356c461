// Internal interface of the "skinny" recurrent-step GEMM: out[M<=64.., N] = sum over K-segments
// of A_s[M,K_s] * B_s, with the GRU/LSTM gate math fused into the epilogue.
//
// This is the kernel that runs once per (layer, phase, timestep) inside the decoder scan, i.e. the
// replacement for one Blocks GatedRecurrent.apply(..., iterate=False) call
// (reference model.py:659-662, 707-710, 719-722) plus the Fork/Linear applies that feed it
// (model.py:655, 692-700, 712).
#pragma once
#include "common.h"

#ifndef SK_NWAVES
#define SK_NWAVES 8
#endif
enum { SK_MAXSEG = 5, SK_MAXJOB = 9, SK_NW = SK_NWAVES, SK_THREADS = SK_NW * 64 };

enum SkEpi {
    SK_EPI_LINEAR = 0,     // out = act(acc + bias + add) (optionally accumulated into out)
    SK_EPI_GRU_GATES = 1,  // N = 2H: z -> o1, r -> o2, r*h_prev -> out
    SK_EPI_GRU_CAND = 2,   // N = H : c = tanh(pre) -> o1, h' = z*c + (1-z)*h_prev -> out
    SK_EPI_BWD_RH = 3,     // N = H : acc = d(r*h_prev); writes dG_r and accumulates dh_prev
    SK_EPI_LSTM = 4,       // N = 4H with gate-interleaved columns (see lstm notes in skinny.hip)
};

enum SkAct { SK_ACT_NONE = 0, SK_ACT_RELU = 1, SK_ACT_TANH = 2, SK_ACT_SIGMOID = 3 };

struct SkSeg {
    const float* A;  // [M, K] row-major, leading dimension lda (K contiguous)
    const float* B;  // b_kcontig: 0: B[k * ldb + n]   1: B[n * ldb + k]
                     //            2: fragment-major copy (sk_tile_weights): B = block of column tile 0 / this
                     //               segment's first chunk, ldb = floats between consecutive column tiles
                     //            3: bf16 fragment-major copy (sk_tile_weights_bf16; 32-deep chunks): same addressing,
                     //               B / ldb still counted in 4-byte units; the activations are rounded to bf16 too
    int lda, ldb, K, b_kcontig;
};

struct SkJob {
    SkSeg seg[SK_MAXSEG];
    int nseg, M, N, epi;
    int act, accumulate, H, aligned;  // accumulate: 0 store, 1 out += (the job is the exclusive owner of its tiles)
                                      // aligned: every segment allows the branch-free 16-byte fetch
    const float* bias;  // [N] or null
    const float* add;   // [M, N] additive pre-activation input or null
    float* out;
    const float* e0;  // epilogue input 0 (h_prev)
    const float* e1;  // epilogue input 1 (z for CAND, r for BWD_RH, c_prev for LSTM)
    float* o1;
    float* o2;
    const float* mask;  // [M] optional step mask (GRU_CAND): h' = m*h' + (1-m)*h_prev
    int ld_add, ldo, lde0, lde1, ldo1, ldo2, wait_all, ksplit;
    // Optional in-launch dependency: the A operand of the LAST segment is produced by other workgroups of the same
    // launch (the attention step of a heterogeneous launch, ska_kernel).  The workgroup multiplies all other segments
    // first, then waits until *wait_flag >= wait_target and takes the last segment with sc1 (L1-bypassing) loads.
    // Fragment-major weights only; the producers must be dispatched BEFORE the waiting workgroups (sk_launch_att
    // puts them first), the wait is bounded (~1 s, then the kernel traps).
    // wait_all = 1 / 2 (wide bf16 kernel only, single-segment jobs): the WHOLE A operand is produced inside the launch (the
    // state-backward rows at the head of the fused backward tick, wkb_kernel): the workgroup waits before its first load.
    // 2: the producer is the slow one of the launch (the rows behind the attention backward): such jobs go last in the grid.
    // ksplit = 2..4 (wide bf16 kernel only; single-segment LINEAR jobs): the K range is cut in equal parts handled by different
    // workgroups; the first half follows `accumulate` into `out`, the second half goes into o1 (ldo1): stored, or added when
    // ldo2 != 0 (a destination several jobs of a window add to; the field is otherwise unused by LINEAR jobs) -- the consumer
    // adds the two.  A wide workgroup's time is the time to stream its [M, K] operand: halving K halves the launch.
    float* kout2;  // ksplit = 3 / 4: the third / fourth K part's sums are STORED here (leading dimension ldo1)
    float* kout3;
    const unsigned* wait_flag;
    unsigned wait_target;
    int colmode;  // 1: a LINEAR job over the gate-interleaved column order of an LSTM matrix (N = 4H; the fragment-major
                  // copies of such matrices hold their column tiles in that order): output column of (tile, j) as in
                  // SK_EPI_LSTM.  Used by the input projections of LSTM layers (plans.hip, schedule 5).
};

struct SkLaunch {
    SkJob job[SK_MAXJOB];
    int njobs, zmode;  // zmode: grid.z = job index (all jobs have the same number of workgroups)
    int tile_end[SK_MAXJOB];  // 16-column tiles per job; sk_launch turns it into the prefix of workgroups
    int force_tile;  // host-side hint: 10 * MB + NB to use where legal (0 = the heuristic of sk_prepare)
    int force_wide;  // host-side hint: the wide bf16 kernel takes the launch whenever it is legal (no minimum size)
    int full_wgs;  // host-side hint: workgroups at which the launch counts as filling its share of the chip
                   // (0 = the whole chip, 224); plans that run several launches side by side pass less
};

// Enqueue one launch on `stream`. Returns hipError_t / PH_ERR_*.
int sk_launch(const SkLaunch& L, hipStream_t stream);
// Tile shape, grid, dynamic LDS bytes and the finished descriptor of a launch (mbnb = 10 * MB + NB).
void sk_prepare(const SkLaunch& Lin, SkLaunch& L, dim3& grid, size_t& lds, int& mbnb);
struct AttFwdArgs;
// The attention forward step and the jobs of L in ONE launch (attention workgroups first, see ska_kernel).
int sk_launch_att(const SkLaunch& L, const AttFwdArgs& att, hipStream_t stream);
struct AttBwdArgs;
struct GruStateBwdArgs;
// Attention backward (or null) + the GRU state backward of all chains as row blocks + the jobs of L in ONE launch
// (skinny.hip skb_kernel; the jobs must not depend on the row blocks): plans.hip bwd8.
int sk_launch_bwd_hetero(const SkLaunch& L, const AttBwdArgs* att, const GruStateBwdArgs& sa, int l0_chain, hipStream_t stream);
struct LstmStateBwdArgs;
// The fused backward tick of LSTM layers with bf16 operands (plans.hip schedule 7): attention backward (or null) + the
// state backward of every chain as 1024-thread row blocks at the head of the grid, each chain publishing its dP rows
// write-through and arriving on flags[chain]; behind them the wide step workgroups of L, whose jobs wait (wait_all) on
// the flag of the chain that writes their operand.  Returns PH_ERR_UNSUPPORTED when the wide kernel does not take the
// launch (the caller then runs the two launches one after the other).
int sk_launch_bwd_fused(const SkLaunch& L, const AttBwdArgs* att, const LstmStateBwdArgs& sa, int l0_chain,
                        unsigned* const* flags, hipStream_t stream);
int sk_zero_words_launch(unsigned* p, int n, hipStream_t stream);
void sk_profile_begin();
long long sk_profile_end(double* total_us, double* flops, double* bytes);
long long sk_profile_end2(double* total_us, double* flops, double* bytes, double* plain);  // + the plain (non-heterogeneous) launches alone

// Helpers to build jobs.
static inline SkSeg sk_seg(const float* A, int lda, const float* B, int ldb, int K, int b_kcontig) {
    SkSeg s;
    s.A = A; s.B = B; s.lda = lda; s.ldb = ldb; s.K = K; s.b_kcontig = b_kcontig;
    return s;
}
int sk_tile_weights_launch(const float* W, int rows, int cols, int ld, float* out, int mode, int lstm_H,
                           hipStream_t stream);
int sk_tile_weights_bf16_launch(const float* W, int rows, int cols, int ld, void* out, int mode, int lstm_H,
                                hipStream_t stream);
void sk_job_init(SkJob& j);
void sk_finalize_job(SkJob& j);  // computes `aligned`
int sk_make_launch(SkLaunch& L, const SkJob* jobs, int njobs);

// bf16 launches: would the wide step kernel (wk_kernel) take jobs of M rows, ncols columns in total, K segments of H / E?
bool sk_wide_takes(int M, int ncols, int H, int E);
