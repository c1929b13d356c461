/*
 * parrot_hip.h -- C ABI of libparrot_hip.so, the MI355X (gfx950) implementation of the
 * acoustic-frame generation hot path of sotelo/parrot (Char2Wav).
 *
 * The reference has no FFI boundary of its own: the hot path sits behind Python brick/operator
 * calls that Theano compiles at run time.  Each entry point below names
 * the reference call it replaces (file:line relative to the reference checkout).  The Python host
 * layer (parrot_amd/) binds these symbols with ctypes; INTEGRATION.md shows the stub a maintainer
 * of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - matrices are row-major float32 with explicit leading dimensions, "x . W" convention of the
 *     reference (W is [in, out]);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued asynchronously on it, the
 *     library never synchronises and never allocates device memory;
 *   - return value 0 = success; otherwise a hipError_t value, or PARROT_ERR_* for bad arguments.
 *     Plans additionally keep the first error (parrot_plan_last_error).
 */
#ifndef PARROT_HIP_H
#define PARROT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PARROT_ERR_BADARG 10001
#define PARROT_ERR_UNSUPPORTED 10002

#define PARROT_NORM_EPS 1e-5f /* _simple_norm's eps default, model.py:24 */
#define PARROT_MAX_LAYERS 3

/* Library / build identification ("parrot_hip <ver> gfx950"). */
const char* parrot_hip_version(void);

/* Measurement aid (bench.py roofline leg): between begin and end every dispatch of the recurrent-step
 * kernel is timed with HIP events attached to the dispatch (hipExtLaunchKernelGGL start/stop events)
 * and its algorithmic flops / bytes are accumulated.  Use only with eager plans (use_graph = 0) and
 * call parrot_profile_end after synchronising the stream; it returns the number of dispatches and the
 * summed kernel time [us], flops and bytes. */
int parrot_profile_begin(void);
long long parrot_profile_end(double* total_us, double* flops, double* bytes);
/* As parrot_profile_end; plain4 (or NULL) receives {dispatch time in us, flops, bytes, launch count} of the PLAIN step-GEMM
 * launches alone (no attention / state row blocks in the grid): the dominant kernel without the latency chains that the
 * heterogeneous launches carry beside their GEMM workgroups. */
long long parrot_profile_end2(double* total_us, double* flops, double* bytes, double* plain4);

/* ------------------------------------------------------------------------------------------
 * Dense projections (Blocks Linear / Fork applies, model.py:580-627, 739-755; lib.ops.Linear,
 * sampleRNN/lib/ops.py:32-128).  C[M,N] (+)= alpha * opA(A) * opB(B) + bias, f32 MFMA.
 *   transA = 0: A is [M,K] (lda), 1: A is stored [K,M] (lda)      (same for B / transB, [K,N])
 *   batched with element strides; split_k > 1 splits K over workgroups, the partial products are summed in
 *   slice order by a second kernel (deterministic; inside a stream capture the product runs unsplit);
 *   split_k = 0 picks a split automatically (long reductions with few output tiles, e.g. deferred weight gradients).
 * M <= 64 with transA = 0 dispatches to the weight-streaming recurrent-step kernel.
 * ------------------------------------------------------------------------------------------ */
#define PARROT_PRECISION_F32 0
#define PARROT_PRECISION_BF16 1
#define PARROT_PRECISION_BF16X3 2
/* Operand precision of the batched path of parrot_gemm (M > 64, or transA / batched / split-K calls), process-wide:
 * PARROT_PRECISION_F32 (default; the reference computes in floatX = float32, model.py:21) or PARROT_PRECISION_BF16:
 * A and B are read as f32 and rounded to bf16 (nearest even) on their way into the matrix cores, products are
 * accumulated in f32, C / bias / activation stay f32 (BASELINE configs[3]).  Not a per-stream setting: change it only
 * between calls.  A decoder plan created with bf16 = 1 applies the mode to its own batched projections regardless. */
int parrot_set_gemm_precision(int mode);
int parrot_get_gemm_precision(void);
int parrot_gemm(const float* A, int lda, int transA, const float* B, int ldb, int transB, float* C, int ldc,
                int M, int N, int K, const float* bias, float alpha, int accumulate, int act, int nbatch,
                long long strideA, long long strideB, long long strideC, int split_k, void* stream);

/* C[M,N] = A . B kept where gate[m,n] > 0 and zeroed elsewhere (gate: row-major, leading dimension ldg >= N): the
 * backward of y = relu(x . W + b) through the saved activation y, fused into the product that makes the gradient wrt
 * relu's output (three_tier.py:504-509: the two ReLU layers of sample_level_predictor).  Operand layouts as parrot_gemm. */
int parrot_gemm_gated(const float* A, int lda, int transA, const float* B, int ldb, int transB, float* C, int ldc,
                      int M, int N, int K, const float* gate, int ldg, void* stream);

/* bf16-IN weight-gradient product (round 4): C[M,N] (+)= A^T . B with A [K, M] and B [K, N] ALREADY bf16 in device
 * memory (row-major, leading dimensions in elements, both multiples of 8; M, N multiples of 8; 16-byte aligned),
 * f32 accumulation, f32 C; deterministic split-K as parrot_gemm (split_k = 0: automatic).  parrot_to_bf16 makes the
 * operand copies (round to nearest even, n a multiple of 8): what a bf16-operand decoder's deferred weight gradients
 * dW = X^T . dPre (the Theano gradient of the Fork / recurrent weights inside model.py:651-724) run on. */
int parrot_to_bf16(const float* x, void* y, long long n, void* stream);
int parrot_gemm_bf16in(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K,
                       int accumulate, int split_k, void* stream);
/* Round 5: the same bf16-in product for every operand layout parrot_gemm takes (A(m,k) at A[m*lda + k], or A[k*lda + m]
 * when transA; B(k,n) at B[k*ldb + n], or B[n*ldb + k] when transB; leading dimensions in bf16 elements, multiples of 8;
 * K % 8 == 0; the contiguous extent of an operand a multiple of 8), plus an optional f32 bias [N]: the readout products
 * h . Wr and dread . Wr^T of a bf16-operand decoder (model.py:739-755) on bf16 copies instead of f32 operands rounded
 * inside the product.  parrot_gemm_bf16in(A, ...) == parrot_gemm_bf16in_ex(A, lda, 1, B, ldb, 0, ..., NULL, ...). */
int parrot_gemm_bf16in_ex(const void* A, int lda, int transA, const void* B, int ldb, int transB, float* C, int ldc, int M,
                          int N, int K, const float* bias, int accumulate, int split_k, void* stream);

/* out[n] (+)= sum_m x[m, n]  -- bias gradients. */
int parrot_colsum(const float* x, long long M, int N, int ld, float* out, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * One GatedRecurrent step (Blocks GatedRecurrent.apply(inputs, gate_inputs, states,
 * iterate=False), model.py:659-662, 707-710, 719-722; same algebra ops.py:364-393):
 *   g = sigmoid(h . Wg + gate_inputs); z = g[:, :H]; r = g[:, H:]
 *   c = tanh((r*h) . Wc + inputs);     h' = z*c + (1-z)*h;  h' = m*h' + (1-m)*h if mask
 * Wg = state_to_gates [H,2H], Wc = state_to_state [H,H].  z, r, rh, c ([B,H] each) are the saved
 * activations the backward step needs.
 * ------------------------------------------------------------------------------------------ */
int parrot_gru_step_fwd(const float* h, const float* inputs, const float* gate_inputs, const float* mask,
                        const float* Wg, const float* Wc, float* h_out, float* z, float* r, float* rh,
                        float* c, int B, int H, void* stream);

/* Backward of the step: given dh_out [B,H] produces d_inputs (=dC) [B,H], d_gate_inputs (=dG)
 * [B,2H] and dh [B,H] (overwritten).  Weight gradients are left to the caller:
 * dWg += h^T dG, dWc += rh^T dC (parrot_gemm with transA = 1). */
int parrot_gru_step_bwd(const float* dh_out, const float* h, const float* mask, const float* Wg,
                        const float* Wc, const float* z, const float* r, const float* c, float* dh,
                        float* d_inputs, float* d_gate_inputs, int B, int H, void* stream);

/* ------------------------------------------------------------------------------------------
 * Plain GRU scans (Blocks GatedRecurrent.apply over a sequence -- the bidirectional encoder,
 * model.py:226-231, 245; lib.ops.LowMemGRU / stackedGRU, ops.py:395-440, 612-777).
 * Up to 4 independent chains (e.g. forward + backward direction) advance in the same launches.
 * ------------------------------------------------------------------------------------------ */
typedef struct ParrotGruSeqDesc {
    int T, B, H, nchain, use_graph, reserved;
    int reverse[4];              /* chain consumes its inputs from t = T-1 down to 0 */
    const float* Wg[4];          /* [H,2H] */
    const float* Wc[4];          /* [H,H]  */
    const float* inputs[4];      /* [T,B,H]  pre-projected candidate inputs (biases included) */
    const float* gate_inputs[4]; /* [T,B,2H] */
    const float* mask;           /* [T,B] or NULL (shared by all chains) */
    float* h[4];                 /* [T+1,B,H]; slot 0 = initial state (caller), slot s+1 = state after
                                    the chain's s-th processed step */
    float* z[4]; float* r[4]; float* rh[4]; float* c[4]; /* [T,B,H] saved activations (per step s) */
    /* backward */
    float* dh[4];                /* [T+1,B,H] in: dL/dh[slot] from consumers (slots 1..T), slot 0 zero;
                                    out: total gradient per slot (slot 0 = grad of the initial state) */
    float* dG[4];                /* [T,B,2H] out: gradient wrt gate_inputs (per step s) */
    float* dC[4];                /* [T,B,H]  out: gradient wrt inputs      (per step s) */
} ParrotGruSeqDesc;

int parrot_gru_seq_create(const ParrotGruSeqDesc* desc, void** plan);
int parrot_gru_seq_fwd(void* plan, void* stream);
int parrot_gru_seq_bwd(void* plan, void* stream);
int parrot_gru_seq_destroy(void* plan);

/* ------------------------------------------------------------------------------------------
 * LSTM scans (lib.ops.__LSTMStep / LowMemLSTM / stackedLSTM, sampleRNN/lib/ops.py:461-610, 823-989):
 *   pre = s_{t-1} . W + pre_in[t]   (pre_in = x_t . U + b, gate order i | f | o | g, each H wide)
 *   i,f,o = sigmoid, g = tanh;  c_t = c_{t-1}*f + g*i;  s_t = tanh(c_t)*o
 * W = Recurrent_Gates [H,4H].  s / c histories have T+1 slots (slot 0 = initial state).
 * Backward: in dS [T+1,B,H] (gradient wrt s slots from the consumers), in/out dc [B,H] (carry: gradient
 * wrt the final cell on entry, wrt the initial cell on return); out dP [T,B,4H] (gradient wrt pre_in) and
 * dS slot 0 (gradient wrt the initial state).  dW = s[0:T]^T dP is left to the caller (parrot_gemm).
 * ------------------------------------------------------------------------------------------ */
typedef struct ParrotLstmSeqDesc {
    int T, B, H, use_graph;
    const float* W;        /* [H,4H] */
    const float* pre_in;   /* [T,B,4H] */
    float* s; float* c;    /* [T+1,B,H] */
    float* gates;          /* [T,B,4H] saved activations (i|f|o|g after the nonlinearity) */
    float* dS;             /* [T+1,B,H] */
    float* dc;             /* [B,H] */
    float* dP;             /* [T,B,4H] */
} ParrotLstmSeqDesc;

int parrot_lstm_seq_create(const ParrotLstmSeqDesc* desc, void** plan);
int parrot_lstm_seq_fwd(void* plan, void* stream);
int parrot_lstm_seq_bwd(void* plan, void* stream);
int parrot_lstm_seq_destroy(void* plan);

/* ------------------------------------------------------------------------------------------
 * GMM-window attention step (model.py:664-690; sampling variant :931-958).
 * att_type 0 = graves, 1 = softmax.  WattT = h1_to_att [alpha;beta;kappa] weights stored
 * TRANSPOSED, [3A,H] (row j = output j), so the per-row dot products read contiguous memory.
 * ------------------------------------------------------------------------------------------ */
int parrot_gmm_attention_fwd(const float* h1, const float* WattT, const float* batt, const float* kappa_prev,
                             const float* ctx, float* a, float* b, float* kappa, float* phi, float* w, int B,
                             int H, int A, int U, int E, int att_type, float eps, float alignment,
                             float sharpening, float timing, void* stream);

/* dw [B,E] in; dkappa [B,A] in/out carry; dp [B,3A] out; dh1 [B,H] accumulated. */
int parrot_gmm_attention_bwd(const float* dw, const float* ctx, const float* a, const float* b,
                             const float* kappa, const float* kappa_prev, const float* WattT, float* dkappa,
                             float* dp, float* dh1, int B, int H, int A, int U, int E, int att_type, float eps,
                             void* stream);

/* ------------------------------------------------------------------------------------------
 * Whole-window decoder scan for training: the theano.scan over `step` of Parrot.compute_cost
 * (model.py:651-737) and its reverse-mode gradient.  L in {1,2,3} GRU layers with the reference's
 * dense skip topology (h_j -> h_l for j < l), attention after layer 1 (0-based layer 0).
 *
 * Packed per-layer weights, rows in this order (K_l = H + E + l*H, l 0-based):
 *     [ h_l (recurrent) ; w (attention context) ; h_0 .. h_{l-1} ]
 *   Wg[l] [K_l, 2H]: rnn{l+1}.state_to_gates ; inp_to_h{l+1}.gates ; h{j+1}_to_h{l+1}.gates
 *   Wc[l] [K_l,  H]: rnn{l+1}.state_to_state ; inp_to_h{l+1}.inputs ; h{j+1}_to_h{l+1}.inputs
 *   bg[l] [2H], bc[l] [H]: sums of the biases of the Forks feeding the layer.
 * Layer 0 consumes w_{t-1}; layers >= 1 consume w_t (model.py:655, 692-693).
 * History buffers have T+1 slots: slot 0 = state entering the window (learned initial state or the
 * carried last_* state, model.py:633-643), slot t+1 = value after step t.
 * ------------------------------------------------------------------------------------------ */
typedef struct ParrotDecoderDesc {
    int T, B, H, E, A, U, L, att_type, use_graph;
    int reserved; /* 0 (rounds 3-4: independent row "strands" on their own streams; measured slower, removed in round 5) */
    float eps, alignment, sharpening, timing;
    const float* Wg[PARROT_MAX_LAYERS];
    const float* Wc[PARROT_MAX_LAYERS];
    const float* bg[PARROT_MAX_LAYERS];
    const float* bc[PARROT_MAX_LAYERS];
    const float* WattT; /* [3A,H] */
    const float* batt;  /* [3A]   */
    const float* ctx;   /* [B,U,E] encoder output * labels_mask (model.py:645-646) */
    /* Per-step additive inputs (teacher-forcing feedback + speaker, model.py:562-627), or NULL.
     * For layers l >= 1 the buffers double as the destination of the chunk-batched projections of the lower
     * layers' outputs (see plans.hip, "chunked layer pipeline"): when every layer l >= 1 has them the scan
     * runs layer-pipelined and ADDS those projections in place (bit l of seq_init tells whether the buffer
     * holds caller data or is scratch to be overwritten).  The caller must refill them before each seq_fwd. */
    float* seq_c[PARROT_MAX_LAYERS]; /* [T,B,H]  */
    float* seq_g[PARROT_MAX_LAYERS]; /* [T,B,2H] */
    /* forward state / saved activations */
    float* h[PARROT_MAX_LAYERS];   /* [T+1,B,H] */
    float* w;                      /* [T+1,B,E] */
    float* kappa;                  /* [T+1,B,A] */
    float* z[PARROT_MAX_LAYERS]; float* r[PARROT_MAX_LAYERS];
    float* rh[PARROT_MAX_LAYERS]; float* c[PARROT_MAX_LAYERS]; /* [T,B,H] */
    float* a; float* b;            /* [T,B,A] */
    float* phi;                    /* [T,B,U] */
    /* backward */
    float* dh[PARROT_MAX_LAYERS];  /* [T+1,B,H] in: gradient from the readouts per slot; out: total */
    float* dw;                     /* [T+1,B,E] in: gradient from att_to_readout per slot; out: total */
    float* dw0;                    /* [T+1,B,E] scratch, zero-filled by the caller: layer-0 share of dw */
    float* dhup[PARROT_MAX_LAYERS];/* [T+1,B,H] scratch, zero-filled by the caller: share of dh[l] that comes from
                                      the layers above (needed for l < L-1) */
    float* dkappa;                 /* [B,A] in: gradient wrt final kappa (0); out: wrt initial kappa */
    float* dG[PARROT_MAX_LAYERS];  /* [T,B,2H] out: gradient wrt gate pre-activations */
    float* dC[PARROT_MAX_LAYERS];  /* [T,B,H]  out: gradient wrt candidate pre-activations */
    float* dp;                     /* [T,B,3A] out: gradient wrt attention projection */
    /* LSTM variant of the layers (cell = 1; generalisation for BASELINE configs[3], cell algebra of
     * sampleRNN/lib/ops.py:505-553, gate order i|f|o|g): one packed matrix per layer, passed in Wg
     * ([K_l,4H], rows as above), bias in bg ([4H]), per-step inputs in seq_g ([T,B,4H]), gradient wrt the
     * pre-activations out in dG ([T,B,4H]); Wc / bc / seq_c / z / r / rh / c / dC are unused. */
    int cell;
    int seq_init;                     /* bit l set: seq_g[l]/seq_c[l] hold caller data (else scratch), see below */
    float* cst[PARROT_MAX_LAYERS];    /* [T+1,B,H] cell-state history (slot 0 = entering the window) */
    float* gate4[PARROT_MAX_LAYERS];  /* [T,B,4H] saved gate activations */
    float* dcell[PARROT_MAX_LAYERS];  /* [B,H] in: gradient wrt the final cell (0), out: wrt the initial cell */
    /* layer_norm = 1 (model.py:24-34, 703-722): the projections of the lower layers' outputs into layer l
     * (Fork h{j}_to_h{l}, index [l*PARROT_MAX_LAYERS + j], j < l) are normalised row-wise before they are
     * summed, so they cannot ride in the packed GEMM; the scan then runs on the chunked layer pipeline and
     * needs, per (l, j) pair and group: the Fork bias (ln_b*, taken out of bg/bc by the caller), a buffer for
     * the normalised projection (ln_y*, [T,B,width]; seq_bwd overwrites it with the gradient wrt the
     * PRE-norm projection, which the caller turns into the weight/bias gradients) and the row std (ln_s*,
     * [T,B]).  Rows of Wg/Wc keep the packed layout. */
    /* bf16 = 1 (BASELINE configs[3]): Wg_f / Wc_f / Wg_r / Wc_r are bf16 copies made by parrot_tile_weights_bf16
     * (mandatory then; H and E multiples of 32), the step GEMMs round the activations they read to bf16 and run on
     * v_mfma_f32_16x16x32_bf16 with f32 accumulation, the batched lower-layer projections of the chunked schedule take
     * the bf16 path of parrot_gemm.  States, gate activations, attention and all gradients buffers stay f32. */
    int layer_norm, bf16;
    const float* ln_bg[PARROT_MAX_LAYERS * PARROT_MAX_LAYERS];
    const float* ln_bc[PARROT_MAX_LAYERS * PARROT_MAX_LAYERS];
    float* ln_yg[PARROT_MAX_LAYERS * PARROT_MAX_LAYERS];
    float* ln_yc[PARROT_MAX_LAYERS * PARROT_MAX_LAYERS];
    float* ln_sg[PARROT_MAX_LAYERS * PARROT_MAX_LAYERS];
    float* ln_sc[PARROT_MAX_LAYERS * PARROT_MAX_LAYERS];
    /* Optional fragment-major copies of the packed layer matrices (parrot_tile_weights), same element count as
     * Wg[l] / Wc[l]: *_f for the forward products (mode 0; LSTM: lstm_H = H), *_r for the backward products
     * with the transpose (mode 1).  When all are given the scan reads the weights through them (contiguous
     * 1 KB wave loads); the caller refreshes them whenever the weights change.  NULL: plain matrices. */
    const float* Wg_f[PARROT_MAX_LAYERS];
    const float* Wc_f[PARROT_MAX_LAYERS];
    const float* Wg_r[PARROT_MAX_LAYERS];
    const float* Wc_r[PARROT_MAX_LAYERS];
    /* Optional [T,B,2] int32 scratch: per step and row the first / last context position whose window weight phi is
     * not exactly 0.0f (written by seq_fwd, read by seq_bwd).  Rows outside it are multiplied by exact zeros in
     * model.py:675-690 and its gradient, so the scan does not read them; NULL: all U rows are read. */
    int* att_sup;
    /* Optional workspace of the persistent forward scan (PARROT_SCHEDULE=4): at least
     * parrot_decoder_persist_floats(desc) floats, ZERO-FILLED by the caller once.  It holds the machine's
     * fragment-major activation slabs, pre-activation buffers, unit table and barrier words.  With it (and GRU layers,
     * B <= 64, H and E multiples of 16, U <= 800, fragment-major weight copies present) seq_fwd runs the whole window as
     * one resident kernel: weights stationary in LDS, phases separated by grid barriers (parrot_amd/csrc/persist.h).
     * NULL, too small or a non-qualifying configuration: the launch schedules are used. */
    float* persist_ws;
    long long persist_ws_floats;
    /* Optional second accumulators of the backward scan (round 4; LSTM layers with bf16 operands): [T+1,B,H] per layer
     * for dh and dhup, [T+1,B,E] for dw and dw0, ZERO-FILLED by the caller once.  With all of them given the transposed
     * products of a backward tick (dP . W^T, K = 4H) are cut into two K halves handled by different workgroups -- a wide
     * workgroup's time is the time to stream its [B, K] operand -- and the second half's sums land here instead of in
     * dh / dhup / dw / dw0: stored into dh_b and dw0_b (one writer per slot), ADDED into dhup_b and dw_b (every layer above
     * adds its share: the caller zero-fills these two before every seq_bwd, as it does dhup and dw0); the state backward
     * and the attention backward of the next tick add both parts (so does the caller for slot 0).  NULL: one accumulator
     * per gradient, as before. */
    float* dh_b[PARROT_MAX_LAYERS];
    float* dhup_b[PARROT_MAX_LAYERS];
    float* dw_b;
    float* dw0_b;
    /* Third accumulators (same shapes and rules as dhup_b / dw_b / dw0_b; dw0_c is stored, the other two are added into
     * and zero-filled by the caller before every seq_bwd).  With all second AND third accumulators given, a 2-layer f32
     * GRU decoder runs the K-balanced backward tick (plans.hip bwd8): every K = 2H product of a tick is cut into its
     * update-gate and reset-gate halves writing separate buffers, layer 1 runs two ticks ahead of layer 0, and the
     * downward products of layer 1 ride in the NEXT tick's attention launch beside the attention backward blocks. */
    float* dhup_c[PARROT_MAX_LAYERS];
    float* dw_c;
    float* dw0_c;
    /* Optional (round 5; bf16 LSTM decoders whose backward tick is one launch): dG16[l], [T,B,4H] bf16 -- the backward scan
     * also leaves every pre-activation gradient row rounded to bf16 (nearest even: exactly what parrot_to_bf16 would make
     * of dG[l]), so the deferred weight-gradient products (parrot_gemm_bf16in) need no conversion pass over 3 x [T,B,4H]
     * floats.  parrot_decoder_writes_bf16_grads(plan) says whether the plan honours them (all given, fused backward). */
    void* dG16[PARROT_MAX_LAYERS];
} ParrotDecoderDesc;

/* Floats of persist_ws a plan for this descriptor needs; 0 when the configuration does not qualify for the persistent
 * scan (see above).  Only T, B, H, E, A, U, L, cell, layer_norm and the Wg_f / Wc_f pointers are read. */
long long parrot_decoder_persist_floats(const ParrotDecoderDesc* desc);
/* 1 when the plan's forward scan runs on the persistent phase machine. */
int parrot_decoder_is_persistent(void* plan);
/* The launch schedule the plan's scan runs on (PARROT_SCHEDULE, after the fall-backs for configurations a schedule does
 * not cover): 0 merged wavefront, 2 / 3 chunked pipelines, 5 balanced wavefront (attention beside the upper layers'
 * input projections), 6 two launches per tick (attention inside the gate launch). */
int parrot_decoder_schedule(void* plan);
/* Which backward tick the plan runs: 8 = the K-balanced tick with the downward products of the upper layers inside the
 * attention launch (GRU stacks of 2 or 3 layers, f32: reference depth model.py:312-347), 7 = the fused LSTM tick of
 * bf16-operand decoders, 0 = three launches per tick (attention + state backward, X, Y). */
int parrot_decoder_backward_tick(void* plan);
/* Schedule tracing (test infrastructure of the library itself; runs without a GPU): instead of launching, the plan
 * records for every launch of direction `which` (0 forward, 1 backward) and every job in it the byte ranges of the
 * descriptor's buffers the job reads / writes.  Records are 5 x int64: launch index, job id (0..8 step-GEMM jobs, 100 the
 * attention (+ the state backward fused behind it), 200 + c state-backward chain c), kind (0 read, 1 write, 2 read +
 * write, 3 read behind an in-launch flag), lo, hi (byte addresses).  Returns the number of records (fills min(n, cap)
 * of them into `out`), or -(error code); launch schedules 0, 5 and 7 only.  PARROT_TRACE_ONLY=1 lets schedule 7 be
 * created on a box without a GPU. */
long long parrot_decoder_trace(void* plan, int which, long long* out, long long cap);
/* The same dry run, one record per job: 6 x int64 = launch index, job id, M (rows), N (output columns), K (sum over the
 * job's segments), epilogue code (SkEpi; -1 attention forward, -2 attention backward: then M, N, K = B, E, H).  What
 * tools/tick_model.py prices with the measured launch cost model. */
long long parrot_decoder_trace_jobs(void* plan, int which, long long* out, long long cap);
/* Waits for the device; 0, or non-zero when a persistent launch of this plan gave up (a workgroup waited ~1 s for a
 * rendezvous or for an operand that never arrived): the results of that window are invalid.  0 on the launch schedules. */
int parrot_decoder_status(void* plan);

int parrot_decoder_create(const ParrotDecoderDesc* desc, void** plan);
int parrot_decoder_writes_bf16_grads(void* plan);
int parrot_decoder_seq_fwd(void* plan, void* stream);
int parrot_decoder_seq_bwd(void* plan, void* stream);
int parrot_decoder_destroy(void* plan);
/* ------------------------------------------------------------------------------------------
 * Autoregressive decode: the theano.scan over `sample_step` of Parrot.sample_model_fun
 * (model.py:882-1057) for the MSE ("greedy") head: x_t = readout_to_output(readouts_t).
 * Same packed weights as above plus, per layer, the fed-back-output rows (out_to_h*, present
 * when weak/full feedback is on; Wfg[l] [O,2H], Wfc[l] [O,H], NULL otherwise), and the readout
 * stack Wr = [h1_to_readout ; .. ; hL_to_readout ; att_to_readout] [L*H+E, R], br = summed bias,
 * Wo [R,O], bo [O].  x is stored with leading dimension ldx >= O (padded rows are zero).
 * Whole sequences are captured into one hipGraph when use_graph = 1.
 * ------------------------------------------------------------------------------------------ */
typedef struct ParrotSampleDesc {
    int S, B, H, E, A, U, L, O, R, ldx, att_type, use_graph;
    float eps, alignment, sharpening, timing;
    const float* Wg[PARROT_MAX_LAYERS];
    const float* Wc[PARROT_MAX_LAYERS];
    const float* bg[PARROT_MAX_LAYERS];
    const float* bc[PARROT_MAX_LAYERS];
    const float* Wfg[PARROT_MAX_LAYERS];
    const float* Wfc[PARROT_MAX_LAYERS];
    const float* seq_c[PARROT_MAX_LAYERS]; /* [B,H]  constant additive inputs (speaker) or NULL */
    const float* seq_g[PARROT_MAX_LAYERS]; /* [B,2H] */
    const float* WattT; const float* batt;
    const float* Wr; const float* br;      /* [L*H+E, R], [R] (+ speaker readout folded in by caller) */
    const float* radd;                     /* [B,R] constant additive readout term or NULL */
    const float* Wo; const float* bo;      /* [R,O], [O] */
    const float* oadd;                     /* [B,O] constant additive output term or NULL */
    const float* ctx;                      /* [B,U,E] */
    float* x;      /* [S+1,B,ldx] slot 0 = initial_x (zeros, model.py:834-835) */
    float* h[PARROT_MAX_LAYERS]; /* [2,B,H] ping-pong, slot 0 = initial state */
    float* w;      /* [S+1,B,E]  */
    float* kappa;  /* [S+1,B,A]  */
    float* a;      /* [S,B,A] (pi_att) */
    float* bwork;  /* [B,A] scratch */
    float* phi;    /* [S,B,U] */
    float* zwork; float* rwork; float* rhwork; float* readout; /* [B,H],[B,H],[B,H],[B,R] scratch */
    /* GMM head (which_cost = 'GMM', model.py:1017-1033 + sample_gmm :94-118); gmm_K = 0 selects the MSE head.
     * Randomness is supplied by the caller: unif [S,B] in [0,1) picks the component exactly like Theano's
     * multinomial (first k with cumsum(pi) > u), noise [S,B,O] ~ N(0,1) scales sigma. */
    int gmm_K, reserved2;
    float sampling_bias, reserved3;
    const float* Wmu; const float* bmu; const float* Wsig; const float* bsig; const float* Wco; const float* bco;
                                 /* [R,O*K],[O*K] x2 (column o*K + k), [R,K],[K] */
    const float* add_mu; const float* add_sig; const float* add_co; /* [B,O*K] x2, [B,K] speaker terms or NULL */
    const float* unif; const float* noise;
    float* gmm_mu; float* gmm_sig; float* gmm_co;   /* scratch [B,O*K] x2, [B,K] */
    float* pi_out;                                   /* [S,B,K] mixture weights per step (the reference's `pi`) */
    /* LSTM variant (cell = 1): Wg/bg/Wfg/seq_g are 4H wide, Wc/bc/Wfc/seq_c unused; cwork [2,B,H] ping-pong
     * cell state (slot 0 = initial cell), gwork [B,4H] scratch. */
    int cell, reserved5;
    float* cwork[PARROT_MAX_LAYERS];
    float* gwork;
    /* layer_norm = 1 (model.py:899-1006 with _apply_norm active): the feedback Fork out_to_h{l}, the Forks
     * h{j}_to_h{l} and the h{l}_to_readout projections are taken as separate small GEMMs, normalised row-wise
     * and summed; their biases therefore come separately (bfg/bfc, ln_bg/ln_bc [l*PARROT_MAX_LAYERS + j],
     * br_l) and must NOT be folded into bg/bc/br.  seq_g/seq_c hold the already normalised speaker terms.
     * ln_scratch: at least B * (5*Ng + 5*Nc + (L+1)*R) floats (Ng/Nc = widths of the two groups). */
    int layer_norm, reserved7;
    const float* bfg[PARROT_MAX_LAYERS];
    const float* bfc[PARROT_MAX_LAYERS];
    const float* ln_bg[PARROT_MAX_LAYERS * PARROT_MAX_LAYERS];
    const float* ln_bc[PARROT_MAX_LAYERS * PARROT_MAX_LAYERS];
    const float* br_l[PARROT_MAX_LAYERS];
    float* ln_scratch;
    long long ln_scratch_floats;
    /* Optional: decode on the persistent phase machine (parrot_amd/csrc/persist.h) -- the whole S-step loop as one
     * resident kernel, 2L + 3 barrier-separated phases per step instead of 2L + 3 launches.  GRU layers, MSE head, no
     * layer_norm, B <= 64, O <= 64 <= ldx.  The caller provides fragment-major copies (parrot_tile_weights, mode 0) of
     *   Wg_t[l] / Wc_t[l]: [K_l + F_l, 2H] / [K_l + F_l, H] = the packed layer matrix with the feedback rows Wfg / Wfc
     *                      appended and zero-padded to F_l = 64 rows (F_l = 0 without feedback into the layer),
     *   Wr_t: [L*H + E, R],   Wo_t: [R, 64] (columns >= O zero),
     * bo_pad [64] (zeros beyond O), oadd_pad [B, 64] when oadd is used, and a ZERO-FILLED workspace of
     * parrot_sample_persist_floats(desc) floats.  Anything missing or non-qualifying: the per-step launches run.
     * The h ping-pong buffers keep the initial state (slot 0) only; PARROT_SAMPLE_PERSIST=0 disables the machine. */
    const float* Wg_t[PARROT_MAX_LAYERS];
    const float* Wc_t[PARROT_MAX_LAYERS];
    const float* Wr_t; const float* Wo_t; const float* bo_pad; const float* oadd_pad;
    float* persist_ws;
    long long persist_ws_floats;
    /* Optional (round 4): with the MSE head and no layer norm, readout -> output is linear in the readout's operands
     * (model.py:992-1013).  Given
     *   Wro_t:    fragment-major copy of Wr . Wo, [L*H + E, 64] (columns >= O zero), and
     *   ro_const: [B, 64] row-major = (br + radd) . Wo + bo + oadd (columns >= O zero),
     * the machine computes x[t+1] straight from [h_0 .. h_{L-1} ; w] in ONE phase and cuts every product of the step
     * along K by the age of its operands (2L + 2 phases per step, each walking only the rows of its newest operand;
     * plans.hip, build_persist_pieces).  NULL, or PARROT_PM_PIECES=0: the 2L + 3 whole-K phases above. */
    const float* Wro_t;
    const float* ro_const;
    /* Optional (round 5), on top of Wro_t / ro_const: the fed-back frame out of the step's dependency chain.  With feedback
     * into layer 0 only (weak feedback, model.py:899-911) and L >= 2, x[t+1] = x_pre + h_{L-1}[t+1] . A with
     * A = Wro[(L-1)H : L H] and x_pre (the other rows' shares plus ro_const) known two phases before h_{L-1}, so layer 0's
     * next gates take  x_pre . Wfg  (K = 64, early)  +  h_{L-1} . (A . Wfg)  (K = H, the critical piece) and the output
     * product leaves the chain: 2L + 1 phases per step.  Given, for l = 0, fragment-major copies of
     *   Wgx_t[0] / Wcx_t[0]: [H + E + 64 + H, 2H] / [.., H] = the rows of Wg_t[0] / Wc_t[0] followed by A . Wfg / A . Wfc
     *                        (composed in double precision, rounded once; Wf zero-padded to 64 rows)
     * the machine plans that way; NULL, other feedback patterns, L = 1 or PARROT_PM_FBC=0: the 2L + 2 phases above. */
    const float* Wgx_t[PARROT_MAX_LAYERS];
    const float* Wcx_t[PARROT_MAX_LAYERS];
    /* Optional (round 5, B <= 16, 3A <= 32): Watt_t = fragment-major copy (parrot_tile_weights, mode 0) of the attention
     * projection as an [H, 32] matrix (column j < 3A = row j of WattT, the rest zero).  Layer 0's candidate units then also
     * multiply the h_1 tile they have just computed by their 16 rows of it and publish [B, 32] partial sums; the attention
     * row adds H / 16 of them instead of reading its state row and the whole projection matrix (model.py:926-930 is the
     * same sum, associated tile by tile).  NULL or PARROT_PM_ATTFOLD=0: the row computes the projection itself. */
    const float* Watt_t;
} ParrotSampleDesc;

long long parrot_sample_persist_floats(const ParrotSampleDesc* desc);
/* 0: per-step launches; 1: the machine with 2L + 3 whole-K phases; 2: the machine with the step cut along K (Wro_t given);
 * 3: the same with the fed-back frame out of the chain (Wgx_t / Wcx_t given, 2L + 1 phases) */
int parrot_sample_is_persistent(void* plan);
/* Plans the decode machine for `desc` with `nwg` workgroups WITHOUT touching device memory (pointers are only used for
 * address arithmetic) and replays the unit table symbolically: every read must find its value written in an earlier
 * phase, every buffer element is written once.  info16: [0] phases per step, [1] partial-sum buffers, [2] checker
 * verdict (0 ok), [3] units per step, [4 + s] units in phase s, [14] units that stream their weights, [15] bit 0: the
 * fed-back frame is out of the chain (Wgx_t / Wcx_t), bit 1: the attention projection is folded into the candidate units (Watt_t).
 * 0, or PARROT_ERR_UNSUPPORTED when the configuration does not take this plan.  Used by the CPU tests. */
int parrot_sample_plan_pieces_dry(const ParrotSampleDesc* desc, int nwg, int* info16);
/* Like parrot_decoder_status, for a decode plan. */
int parrot_sample_status(void* plan);

int parrot_sample_create(const ParrotSampleDesc* desc, void** plan);
int parrot_sample_run(void* plan, void* stream);
int parrot_sample_destroy(void* plan);

int parrot_plan_last_error(void* plan);

/* ------------------------------------------------------------------------------------------
 * Optimiser step next to the path (train.py:100-108): StepClipping(threshold) o Adam on the flat
 * parameter buffer.  gnorm_sq is a device scalar filled by parrot_sumsq (after the gradient
 * all-reduce in data-parallel runs); grad_scale rescales the raw gradient first (1/world_size).
 * ------------------------------------------------------------------------------------------ */
int parrot_sumsq(const float* x, size_t n, float* out, void* stream);

/* Fragment-major copy of a row-major weight matrix W [rows, cols] (leading dimension ld; rows, cols multiples
 * of 16) for the recurrent-step kernel: 256-float blocks ordered [column tile][16-deep chunk], each holding the
 * 64 lanes' MFMA operand quads.  mode 0: for products x . W (tiles over columns, chunks over rows; lstm_H > 0
 * applies the gate-interleaved column order of the fused LSTM step, cols = 4*lstm_H); mode 1: for products
 * x . W^T (tiles over rows, chunks over columns).  out: rows*cols floats. */
int parrot_tile_weights(const float* W, int rows, int cols, int ld, float* out, int mode, int lstm_H, void* stream);
/* The same copy rounded to bf16 (round to nearest even) for the bf16 operand mode of the decoder scan
 * (ParrotDecoderDesc::bf16): 1 KB blocks of 16 columns x 32 K-rows in v_mfma_f32_16x16x32_bf16 operand order.  The
 * K extent (rows in mode 0, cols in mode 1) must be a multiple of 32, the other one of 16.  out: rows*cols bf16. */
int parrot_tile_weights_bf16(const float* W, int rows, int cols, int ld, void* out, int mode, int lstm_H, void* stream);

/* _simple_norm / _apply_norm of the reference (model.py:24-34; used when layer_norm=True on the Fork outputs
 * and readout projections, model.py:585-620, 703-722, 746): y = (x - mean) / (eps + std) over the last axis
 * (population std, no affine).  x [R,N] (leading dimension ldx); y [R,N] may alias x; sigma [R] receives the
 * row std (saved for the backward); add_dst (or NULL) [R,N] gets += y. */
int parrot_simple_norm_fwd(const float* x, int ldx, float* y, int ldy, float* sigma, long long R, int N, float eps,
                           float* add_dst, int ld_add, void* stream);
/* Backward of the above: dx [R,N] (may alias dy or y) = J^T dy, from the saved y and sigma. */
int parrot_simple_norm_bwd(const float* dy, int lddy, const float* y, int ldy, const float* sigma, float* dx,
                           int lddx, long long R, int N, float eps, int accumulate, void* stream);
int parrot_adam_clip_step(float* param, const float* grad, float* m, float* v, size_t n,
                          const float* gnorm_sq, float grad_scale, float clip_threshold, float lr,
                          float beta1, float beta2, float eps, int step, void* stream);

/* ------------------------------------------------------------------------------------------
 * Audio quantisers (quantize.py:14-99).  x is [rows, n] float32; every row is min-max normalised
 * in float64 exactly as quantize.__batch_quantize does.  mode 0: mu-law -> int16 out,
 * mode 1: linear(q_levels) -> int32 out.  ws = device scratch of 2*rows doubles.
 * parrot_mu2linear: int32 class indices -> float32 amplitudes (quantize.py:68-78).
 * ------------------------------------------------------------------------------------------ */
int parrot_batch_quantize(const float* x, int rows, int n, int ld, double* ws, void* out, int ldo, int mode,
                          int q_levels, void* stream);
int parrot_mu2linear(const int32_t* q, size_t n, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training-side operators of the SampleRNN sample-level tier (round 5; parrot_amd/csrc/trainops.hip).
 *
 * Embedding (sampleRNN/lib/ops.py:252-266) followed by SampleLevel.L1_PrevSamples (three_tier.py:486-497: a
 * [FS*EMB, DIM] Linear without bias over the FS concatenated embeddings) as a gather-sum over the folded table
 * tbl[j][q][:] = Embedding[q] . W1[j*EMB:(j+1)*EMB, :] (the form the generation loop uses, SampleRnnGenDesc::emb_tbl):
 *   parrot_gather_sum_fwd:  y[i,:] = add[i,:] + sum_{j<J} tbl[j][idx[i*J + j]][:]     (add may be NULL; summed in j order)
 *   parrot_gather_sum_bwd:  dtbl[j][q][:] (+)= sum over the rows i with idx[i*J + j] == q of dy[i,:]
 * The backward is a SEGMENTED sum, no scatter-add: perm [J,N] lists, per j, the rows sorted by (idx[.,j], row) (a stable
 * sort, so the order of the addends depends on the indices only) and offs [J,Q+1] the first position of every bin in that
 * order (offs[j][Q] = N).  ws: parrot_gather_sum_bwd_ws_floats(N,J,Q,D) floats of scratch.  D % 4 == 0, 16-byte aligned.
 *
 * Softmax cross-entropy with integer targets (three_tier.py:565-584, T.nnet.categorical_crossentropy(softmax(.), target)):
 *   parrot_softmax_ce_fwd:  lse[i] = log sum_q exp(logits[i,q]),  ce[i] = lse[i] - logits[i, target[i]]
 *   parrot_softmax_ce_bwd:  dlogits[i,q] = rowscale[i] * (exp(logits[i,q] - lse[i]) - [q == target[i]])
 * parrot_relu_gate: out = dy where gate > 0, else 0 (n % 4 == 0; ReLU backward where no product can carry it).
 * ------------------------------------------------------------------------------------------ */
int parrot_gather_sum_fwd(const float* tbl, const int* idx, const float* add, int ldadd, float* y, int ldy, long long N,
                          int J, int Q, int D, void* stream);
long long parrot_gather_sum_bwd_ws_floats(long long N, int J, int Q, int D);
int parrot_gather_sum_bwd(const float* dy, int lddy, const int* perm, const int* offs, float* dtbl, float* ws,
                          long long ws_floats, long long N, int J, int Q, int D, int accumulate, void* stream);
int parrot_softmax_ce_fwd(const float* logits, int ld, const int* target, long long rows, int Q, float* lse, float* ce,
                          void* stream);
int parrot_softmax_ce_bwd(const float* logits, int ld, const int* target, const float* lse, const float* rowscale,
                          long long rows, int Q, float* dlogits, int ldd, void* stream);
int parrot_relu_gate(const float* dy, const float* gate, float* out, long long n, void* stream);

/* Weight-norm fold of a SampleRNN Linear (sampleRNN/lib/ops.py:101-110: `W * (g / W.norm(2, axis=0))`), SURVEY 8b's K9:
 *   samplernn_weightnorm_fold:      W_eff[k][n] = W[k][n] * g[n] / ||W[:, n]||_2;  norm[n] = ||W[:, n]||_2 (may be NULL)
 *   samplernn_weightnorm_fold_bwd:  dg[n] (+)= sum_k dW_eff[k][n] W[k][n] / norm[n]
 *                                   dW[k][n] (+)= g[n] / norm[n] * (dW_eff[k][n] - W[k][n] * sum_k' dW_eff[k'][n] W[k'][n] / norm[n]^2)
 * Row-major [K, N] matrices with leading dimensions ld / ldo / ldd / lddw >= N; ws: samplernn_weightnorm_ws_floats(N)
 * floats of scratch (partial column sums, added in a fixed order: no atomics). */
long long samplernn_weightnorm_ws_floats(int N);
int samplernn_weightnorm_fold(const float* W, int ld, const float* g, float* W_eff, int ldo, float* norm, float* ws, int K, int N,
                              void* stream);
int samplernn_weightnorm_fold_bwd(const float* W, int ld, const float* g, const float* norm, const float* dW_eff, int ldd,
                                  float* dW, int lddw, float* dg, float* ws, int K, int N, int accumulate_w, int accumulate_g,
                                  void* stream);

/* ------------------------------------------------------------------------------------------
 * Conditional three-tier SampleRNN generation: the per-sample loop of generate_and_save_samples
 * (sampleRNN/models/conditional/three_tier.py:794-832) and the three Theano functions it calls
 * (getting_generation_functions, :703-734) as one device-resident launch sequence per 80-sample
 * period, replayed from a hipGraph.  All weights are the EFFECTIVE weights (weight norm,
 * ops.py:101-110, folded by the caller), row-major [in, out].  emb_tbl = Embedding folded with
 * SampleLevel.L1_PrevSamples: emb_tbl[pos][q][:] = Embedding[q] . W1[pos*EMB:(pos+1)*EMB, :].
 * temperature = 0 -> argmax (softmax_and_argmax, ops.py:296-297; lowest index on ties);
 * temperature > 0 -> multinomial draw from a seeded counter-based generator (not Theano's MRG stream).
 * samples: [B, BFS*T] int32, first BFS entries = Q/2 set by the caller; big_h / frm_h hold the
 * initial states (learned h0) on entry and the final states on return.  The weight buffers must not change during a
 * plan's life: create makes fragment-major copies of the tiers' matrices for the step kernel (single-GRU tiers) and, on the
 * persistent path, composes everything in front of L2's ReLU through W2 once (emb_tbl[pos] . W2, frm_Wout_i . W2,
 * frm_bout_i . W2 + b2; round 5) -- the sample kernel then has no L2 product.
 * ------------------------------------------------------------------------------------------ */
typedef struct SampleRnnGenDesc {
    int B, D, T, Q, FS, BFS, feat_dim, use_graph;
    float temperature; int reserved;
    unsigned long long seed;
    const float* big_Win_frames; const float* big_Win_feats; const float* big_bin; /* [BFS,D],[feat,D],[D] */
    const float* big_U; const float* big_bU; const float* big_Wg; const float* big_Wc; /* [D,3D],[3D],[D,2D],[D,D] */
    const float* big_Wout; const float* big_bout;                                  /* [D,(BFS/FS)*D] */
    const float* frm_Win; const float* frm_bin;                                    /* [FS,D],[D] */
    const float* frm_U; const float* frm_bU; const float* frm_Wg; const float* frm_Wc;
    const float* frm_Wout; const float* frm_bout;                                  /* [D,FS*D] */
    const float* emb_tbl;                                                          /* [FS,Q,D] */
    const float* W2; const float* b2; const float* W3; const float* b3; const float* W4; const float* b4;
    const float* features;   /* [T,B,feat_dim] time-major */
    int* samples;            /* [B,BFS*T] */
    float* big_h; float* frm_h;            /* [B,D] */
    /* scratch */
    float* xf_big; float* xf_frm; float* feat_cur; float* gru_in; float* P; float* z; float* r; float* rh;
    float* big_out; float* frame_out; float* o1; float* o2; float* o3; float* logits;
    int* tbase;
    /* Stacked / LSTM tiers (three_tier.py:147-169 allows RNN_TYPE = 'LSTM' and N_RNN in 1..5; stackedGRU / stackedLSTM,
     * sampleRNN/lib/ops.py:612-777, 823-989, without skip connections).  n_rnn = 0 keeps the single-GRU fields above.
     * Otherwise, per tier and layer k < n_rnn, four pointers:
     *   GRU  (lstm = 0): { Input.W [D,3D], Input.b [3D], Recurrent_Gates [D,2D], Recurrent_Candidate [D,D] }
     *   LSTM (lstm = 1): { Input.W [D,4D], b [4D], Recurrent_Gates [D,4D], NULL }   (gate order i | f | o | g)
     * and the layer states big_hs / frm_hs [k] ([B,D]; initial state on entry, final on return) plus, for LSTM, the cell
     * states big_cs / frm_cs [k].  P must then hold B*4D floats, gate_ws B*4D floats (LSTM gate scratch), and
     * layer_tmp B*D floats. */
    int n_rnn, lstm;
    const float* big_L[5][4];
    const float* frm_L[5][4];
    float* big_hs[5]; float* big_cs[5];
    float* frm_hs[5]; float* frm_cs[5];
    float* gate_ws; float* layer_tmp;
    /* Optional workspace of the persistent-thread sample kernel (parrot_amd/csrc/sr_persist.hip): at least
     * samplernn_persist_floats(desc) floats (create initialises it; it belongs to ONE plan).  With it (B <= 32, D in {256, 512, 1024},
     * Q = 256, a 256-CU device) the FS sample steps between two frame-tier steps run as ONE launch: each XCD takes four
     * streams through the whole sample-level MLP with its weights held in LDS / registers and hand-offs that stay inside
     * the XCD's L2.  NULL or a non-qualifying configuration: five launches per sample as before. */
    float* persist_ws;
    long long persist_ws_floats;
} SampleRnnGenDesc;

/* Floats of persist_ws a plan for this descriptor needs; 0 when the configuration does not qualify. */
long long samplernn_persist_floats(const SampleRnnGenDesc* desc);
int samplernn_generate_create(const SampleRnnGenDesc* desc, void** plan);
int samplernn_generate_run(void* plan, void* stream);
/* 1 when the plan's sample steps run on the persistent-thread kernel. */
int samplernn_generate_is_persistent(void* plan);
/* Waits for the device and returns 0, or a non-zero fault code when a persistent sample kernel gave up (a team of
 * workgroups was incomplete or timed out): the samples of that run are invalid.  Always 0 on the launch path. */
int samplernn_generate_status(void* plan);
int samplernn_generate_destroy(void* plan);

#ifdef __cplusplus
}
#endif
#endif /* PARROT_HIP_H */
