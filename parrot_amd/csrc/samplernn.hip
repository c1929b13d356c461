// Conditional three-tier SampleRNN generation loop on the device (gfx950).
//
// Replaces the per-sample Python loop of reference three_tier.py:809-832, which crosses the
// host/device boundary three times per audio sample (big_frame_level_generate_fn every 80 samples,
// frame_level_generate_fn every 10, sample_level_generate_fn every sample).  Here the whole loop is a
// fixed launch sequence per 80-sample period (1 big-tier step, 8 frame-tier steps, 80 sample-MLP
// steps) captured once into a hipGraph and replayed per period; the running sample index lives in
// device memory, so no value ever returns to the host inside the loop.
//
// Arithmetic restated from sampleRNN/lib/ops.py:329-393 (GRU step with the Input linear inside the
// step) and three_tier.py:291-515; weight norm (ops.py:101-110) is folded by the caller, and the
// sample-level Embedding (ops.py:252-266) is folded with SampleLevel.L1_PrevSamples into a
// [FS, Q, D] table, which turns the K = FS*EMB GEMM into a 10-row gather-sum.
#include <new>
#include <stdlib.h>

#include "../../include/parrot_hip.h"
#include "biggemm.h"
#include "skinny.h"
#include "sr_persist.h"

namespace {


// xf[b][i] = (s / (Q/2) - 1) * 2 for the `n` samples before t (three_tier.py:309-310, 398-399);
// optionally copies the conditioning frame of the current big frame.
__global__ __launch_bounds__(256) void sr_prep_kernel(const int* __restrict__ samples, int len, const int* __restrict__ tbase,
                                                      int toff, int n, float half_q, float* __restrict__ xf, int B,
                                                      const float* __restrict__ feats, float* __restrict__ feat_cur,
                                                      int feat_dim, int bfs) {
    const int t = tbase[0] + toff;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < B * n) {
        const int b = idx / n, i = idx % n;
        xf[idx] = ((float)samples[(size_t)b * len + t - n + i] / half_q - 1.0f) * 2.0f;
    }
    if (feats && idx < B * feat_dim) {
        const int b = idx / feat_dim, f = idx % feat_dim;
        feat_cur[idx] = feats[((size_t)(t / bfs) * B + b) * feat_dim + f];
    }
}

// o1[b][d] = sum_pos tbl[pos][samples[b][t-FS+pos]][d] + frame_out[b][d]   (frame_out row already offset)
__global__ __launch_bounds__(256) void sr_embed_sum_kernel(const int* __restrict__ samples, int len,
                                                           const int* __restrict__ tbase, int toff, int FS, int Q, int D,
                                                           const float* __restrict__ tbl, const float* __restrict__ frame_out,
                                                           int ldf, float* __restrict__ o1) {
    const int t = tbase[0] + toff;
    const int b = blockIdx.y;
    const int d4 = blockIdx.x * 256 + threadIdx.x;  // float4 index
    if (d4 * 4 >= D) return;
    f32x4 acc = *reinterpret_cast<const f32x4*>(frame_out + (size_t)b * ldf + d4 * 4);
    for (int pos = 0; pos < FS; ++pos) {
        const int q = samples[(size_t)b * len + t - FS + pos];
        acc += *reinterpret_cast<const f32x4*>(tbl + ((size_t)pos * Q + q) * D + d4 * 4);
    }
    *reinterpret_cast<f32x4*>(o1 + (size_t)b * D + d4 * 4) = acc;
}

// samples[b][t] = argmax_q logits[b][q] (lowest index on ties, like numpy / Theano argmax), or a
// temperature-scaled multinomial draw from a counter-based generator when temperature > 0.
__global__ __launch_bounds__(256) void sr_pick_kernel(const float* __restrict__ logits, int Q, int* __restrict__ samples,
                                                      int len, const int* __restrict__ tbase, int toff, float temperature,
                                                      unsigned long long seed) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const int t = tbase[0] + toff;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* lg = logits + (size_t)b * Q;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int q = tid; q < Q; q += 256) {
        const float v = lg[q];
        if (v > best || (v == best && q < bi)) { best = v; bi = q; }
    }
    sv[tid] = best; si[tid] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            const float v = sv[tid + s];
            const int i = si[tid + s];
            if (v > sv[tid] || (v == sv[tid] && i < si[tid])) { sv[tid] = v; si[tid] = i; }
        }
        __syncthreads();
    }
    if (temperature <= 0.f) {
        if (tid == 0) samples[(size_t)b * len + t] = si[0];
        return;
    }
    // inverse-CDF sampling of softmax(logits / temperature) by one thread (Q = 256)
    if (tid == 0) {
        const float mx = sv[0];
        float tot = 0.f;
        for (int q = 0; q < Q; ++q) tot += expf((lg[q] - mx) / temperature);
        unsigned long long x = seed ^ (0x9E3779B97F4A7C15ull * (unsigned long long)(t + 1)) ^
                               (0xBF58476D1CE4E5B9ull * (unsigned long long)(b + 1));
        x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;  // splitmix64
        const float u = (float)((x >> 40) + 0.5) * (1.0f / 16777216.0f) * tot;
        float c = 0.f;
        int pick = Q - 1;
        for (int q = 0; q < Q; ++q) {
            c += expf((lg[q] - mx) / temperature);
            if (u < c) { pick = q; break; }
        }
        samples[(size_t)b * len + t] = pick;
    }
}

// Frame-tier input (three_tier.py:398-411): out[b][d] = bias[d] + add[b][d] + sum_i xf(b, i) * Win[i][d] with
// xf = (sample / (Q/2) - 1) * 2 of the FS samples before t: the K = FS product is ten FMAs per output, taken
// straight from the integer samples.
__global__ __launch_bounds__(256) void sr_frame_in_kernel(const int* __restrict__ samples, int len, const int* __restrict__ tbase,
                                                          int toff, int FS, float half_q, const float* __restrict__ Win,
                                                          const float* __restrict__ bias, const float* __restrict__ add,
                                                          int ld_add, float* __restrict__ out, int D,
                                                          const float* __restrict__ gpre = nullptr, const float* __restrict__ h = nullptr,
                                                          float* __restrict__ z = nullptr, float* __restrict__ rh = nullptr, int Dh = 0) {
    const int t = tbase[0] + toff;
    const int b = blockIdx.y, dd = blockIdx.x * 256 + threadIdx.x;
    if (dd >= D) return;
    float acc = 0.f;
    for (int i = 0; i < FS; ++i) {
        const float xf = ((float)samples[(size_t)b * len + t - FS + i] / half_q - 1.0f) * 2.0f;
        acc = fmaf(xf, Win[(size_t)i * D + dd], acc);
    }
    acc += (bias ? bias[dd] : 0.f) + add[(size_t)b * ld_add + dd];
    if (gpre && dd < 2 * Dh) {
        // (D = 3 Dh: gates | candidate) the recurrent product h . Wg of this frame's gates is there already (the previous
        // frame's projection launch made it): the gates are finished here, as the sample kernel's tail does for the frames
        // inside a period -- the frame costs one launch less
        const float gt = ph_sigmoid(gpre[(size_t)b * 2 * Dh + dd] + acc);
        if (dd < Dh) z[(size_t)b * Dh + dd] = gt;
        else rh[(size_t)b * Dh + dd - Dh] = gt * h[(size_t)b * Dh + dd - Dh];
        return;
    }
    out[(size_t)b * D + dd] = acc;
}

__global__ void sr_tick_kernel(int* tbase, int inc, int set) {
    if (threadIdx.x == 0 && blockIdx.x == 0) tbase[0] = set ? inc : tbase[0] + inc;
}

#define SR_TRY(x)                 \
    do {                          \
        const int rc__ = (x);     \
        if (rc__ != 0) return rc__; \
    } while (0)

struct SrPlan {
    SampleRnnGenDesc d;
    bool persist = false;  // sample steps on the persistent-thread kernel (sr_persist.hip)
    // Fragment-major copies (sk_tile_weights, mode 0) of the single-GRU tiers' matrices, owned by the plan and made once
    // at create time: the step kernel then reads its weights as contiguous 1 KB wave loads instead of 64-byte row pieces.
    struct Tiled { float* U = nullptr; float* Wg = nullptr; float* Wc = nullptr; float* Wout = nullptr; };
    Tiled t_big, t_frm;
    float* tiled_slab = nullptr;
    // Persistent sample kernel (round 5): everything in front of L2's ReLU is linear, so it is composed through W2 once --
    //   t2tbl[pos] = emb_tbl[pos] . W2  [FS, Q, D],   Pout_i = Wout_i . W2  [D, FS*D],   cb_i = bout_i . W2 + b2  [FS*D]
    // -- the frame tier's projection launch makes h . Pout + cb (same shape and cost as h . Wout + bout) and the sample
    // kernel's L2 pre-activation is a gather-sum: one product and one hand-off fewer per audio sample.
    float* t2tbl = nullptr; float* Pout = nullptr; float* cb = nullptr;
    // Round 5 (single-GRU tiers): the frame tier's input xf . Win + bin + big_out only enters the step through x . U, so
    //   x . U + bU = xf . (Win . U) + [(bin + big_out) . U + bU].
    // winu = Win . U [FS, 3D] and pbias = bin . U + bU [3D] are composed once (the weights are fixed for the plan's life);
    // per 80-sample period ONE product makes pbig[b][f] = big_out[b, f] . U + pbias for all BFS / FS frames, and the two step
    // GEMMs of a frame walk K = D (their own recurrent product) instead of 2 D, with pin [B, 3D] = xf . winu + pbig[f] as
    // their additive input -- written by the previous frame's sample kernel (SrpArgs::next_n) or, for the first frame of a
    // period, by sr_frame_in_kernel at width 3D.
    float* winu = nullptr; float* pbias = nullptr; float* pbig = nullptr; float* pin = nullptr;
    float* gpre = nullptr;  // [B, 2D] h . Wg of the next frame, made beside the projection (persistent path)
    int make_winu() {
        const size_t D = d.D, nfr = d.BFS / d.FS;
        if (d.n_rnn != 0) return 0;
        if (hipMalloc(&winu, ((size_t)d.FS * 3 * D + 3 * D + (size_t)d.B * nfr * 3 * D + (size_t)d.B * 3 * D + (size_t)d.B * 2 * D) * sizeof(float)) != hipSuccess) {
            winu = nullptr;
            return 0;
        }
        pbias = winu + (size_t)d.FS * 3 * D;
        pbig = pbias + 3 * D;
        pin = pbig + (size_t)d.B * nfr * 3 * D;
        gpre = pin + (size_t)d.B * 3 * D;
        BgPrecisionScope f32_only(0);
        int rc = parrot_gemm(d.frm_Win, (int)D, 0, d.frm_U, 3 * (int)D, 0, winu, 3 * (int)D, d.FS, 3 * (int)D, (int)D, nullptr, 1.f, 0, 0,
                             1, 0, 0, 0, 1, nullptr);
        if (rc == 0)
            rc = parrot_gemm(d.frm_bin, (int)D, 0, d.frm_U, 3 * (int)D, 0, pbias, 3 * (int)D, 1, 3 * (int)D, (int)D, d.frm_bU, 1.f, 0, 0,
                             1, 0, 0, 0, 1, nullptr);
        if (rc == 0) rc = (int)hipDeviceSynchronize();
        if (rc != 0) { (void)hipFree(winu); winu = nullptr; }
        return 0;
    }
    int make_composed() {
        const size_t D = d.D, Q = d.Q, FS = d.FS;
        if (hipMalloc(&t2tbl, (FS * Q * D + D * FS * D + FS * D) * sizeof(float)) != hipSuccess) { t2tbl = nullptr; return 1; }
        Pout = t2tbl + FS * Q * D;
        cb = Pout + D * FS * D;
        BgPrecisionScope f32_only(0);  // whatever the process-wide GEMM precision is: these tables feed an f32 path
        int rc = parrot_gemm(d.emb_tbl, (int)D, 0, d.W2, (int)D, 0, t2tbl, (int)D, (int)(FS * Q), (int)D, (int)D, nullptr, 1.f, 0, 0, 1,
                             0, 0, 0, 1, nullptr);
        for (size_t i = 0; i < FS && rc == 0; ++i) {
            rc = parrot_gemm(d.frm_Wout + i * D, (int)(FS * D), 0, d.W2, (int)D, 0, Pout + i * D, (int)(FS * D), (int)D, (int)D, (int)D,
                             nullptr, 1.f, 0, 0, 1, 0, 0, 0, 1, nullptr);
            if (rc == 0)
                rc = parrot_gemm(d.frm_bout + i * D, (int)D, 0, d.W2, (int)D, 0, cb + i * D, (int)D, 1, (int)D, (int)D, d.b2, 1.f, 0, 0,
                                 1, 0, 0, 0, 1, nullptr);
        }
        if (rc == 0) rc = (int)hipDeviceSynchronize();
        if (rc != 0) { (void)hipFree(t2tbl); t2tbl = nullptr; Pout = cb = nullptr; }
        return rc;
    }
    int make_tiled() {
        const size_t D = d.D, nfr = d.BFS / d.FS;
        if (d.n_rnn != 0 || (d.D & 15)) return 0;
        const size_t per = 3 * D * D + 2 * D * D + D * D;
        const size_t total = 2 * per + D * nfr * D + D * d.FS * D;
        if (hipMalloc(&tiled_slab, total * sizeof(float)) != hipSuccess) { tiled_slab = nullptr; return 0; }
        float* p = tiled_slab;
        auto tile = [&](const float* W, int rows, int cols, float*& dst) -> int {
            dst = p;
            p += (size_t)rows * cols;
            return sk_tile_weights_launch(W, rows, cols, cols, dst, 0, 0, nullptr);
        };
        int rc = 0;
        rc |= tile(d.big_U, d.D, 3 * d.D, t_big.U); rc |= tile(d.big_Wg, d.D, 2 * d.D, t_big.Wg);
        rc |= tile(d.big_Wc, d.D, d.D, t_big.Wc); rc |= tile(d.big_Wout, d.D, (int)nfr * d.D, t_big.Wout);
        rc |= tile(d.frm_U, d.D, 3 * d.D, t_frm.U); rc |= tile(d.frm_Wg, d.D, 2 * d.D, t_frm.Wg);
        rc |= tile(d.frm_Wc, d.D, d.D, t_frm.Wc); rc |= tile(persist ? Pout : d.frm_Wout, d.D, d.FS * d.D, t_frm.Wout);
        if (rc != 0 || hipDeviceSynchronize() != hipSuccess) {
            (void)hipFree(tiled_slab);
            tiled_slab = nullptr;
            t_big = Tiled(); t_frm = Tiled();
        }
        return 0;
    }
    enum { SR_CHUNK = 8 };
    hipGraphExec_t exec = nullptr, exec_chunk = nullptr;
    int periods_left_single = 0;
    hipStream_t cap = nullptr;
    int last_error = 0;
    ~SrPlan() {
        if (exec) hipGraphExecDestroy(exec);
        if (exec_chunk) hipGraphExecDestroy(exec_chunk);
        if (cap) hipStreamDestroy(cap);
        if (tiled_slab) (void)hipFree(tiled_slab);
        if (t2tbl) (void)hipFree(t2tbl);
        if (winu) (void)hipFree(winu);
    }

    int linear(const float* A, int lda, const float* W, int ldw, int K, int N, const float* bias, const float* add,
               int ld_add, float* out, int ldo, int act, hipStream_t st, const float* A2 = nullptr, int lda2 = 0,
               const float* W2 = nullptr, int K2 = 0, const float* Wt = nullptr) {
        SkJob j;
        sk_job_init(j);
        j.nseg = 1;
        j.seg[0] = Wt ? sk_seg(A, lda, Wt, (K >> 4) * 256, K, 2) : sk_seg(A, lda, W, ldw, K, 0);
        if (A2) { j.seg[1] = sk_seg(A2, lda2, W2, ldw, K2, 0); j.nseg = 2; }
        j.M = d.B; j.N = N; j.H = N; j.epi = SK_EPI_LINEAR; j.act = act;
        j.bias = bias; j.add = add; j.ld_add = ld_add;
        j.out = out; j.ldo = ldo;
        SkLaunch L;
        SR_TRY(sk_make_launch(L, &j, 1));
        return sk_launch(L, st);
    }

    // GRU step of a tier (ops.py:356-393): gates = sigm(h.Wg + x.U[:, :2D] + b[:2D]); cand = tanh((r*h).Wc + x.U[:, 2D:] +
    // b[2D:]); in-place h.  The Input linear rides in the step GEMMs as a second K segment: two launches, not three.
    int gru(const float* x, const float* U, const float* bU, const float* Wg, const float* Wc, float* h, hipStream_t st,
            const Tiled* t = nullptr, const float* pre = nullptr, bool gates_done = false) {
        const int D = d.D;
        const int blk = (D >> 4) * 256;  // floats per column tile of a [D, *] fragment-major copy
        const bool tl = t && t->U;
        SkJob j;
        SkLaunch L;
        sk_job_init(j);
        if (pre) {  // x . U + bU arrives precomputed ([B, 3D]: gates | candidate): only the recurrent product is left
            if (!gates_done) {  // (gates_done: the previous frame's sample kernel has left z and r*h already)
                j.nseg = 1;
                j.seg[0] = tl ? sk_seg(h, D, t->Wg, blk, D, 2) : sk_seg(h, D, Wg, 2 * D, D, 0);
                j.M = d.B; j.N = 2 * D; j.H = D; j.epi = SK_EPI_GRU_GATES;
                j.add = pre; j.ld_add = 3 * D;
                j.e0 = h; j.lde0 = D;
                j.o1 = d.z; j.ldo1 = D; j.o2 = d.r; j.ldo2 = D; j.out = d.rh; j.ldo = D;
                SR_TRY(sk_make_launch(L, &j, 1));
                SR_TRY(sk_launch(L, st));
            }
            sk_job_init(j);
            j.nseg = 1;
            j.seg[0] = tl ? sk_seg(d.rh, D, t->Wc, blk, D, 2) : sk_seg(d.rh, D, Wc, D, D, 0);
            j.M = d.B; j.N = D; j.H = D; j.epi = SK_EPI_GRU_CAND;
            j.add = pre + 2 * D; j.ld_add = 3 * D;
            j.e0 = h; j.lde0 = D; j.e1 = d.z; j.lde1 = D;
            j.o1 = nullptr; j.out = h; j.ldo = D;
            SR_TRY(sk_make_launch(L, &j, 1));
            return sk_launch(L, st);
        }
        j.nseg = 2;
        j.seg[0] = tl ? sk_seg(h, D, t->Wg, blk, D, 2) : sk_seg(h, D, Wg, 2 * D, D, 0);
        j.seg[1] = tl ? sk_seg(x, D, t->U, blk, D, 2) : sk_seg(x, D, U, 3 * D, D, 0);
        j.M = d.B; j.N = 2 * D; j.H = D; j.epi = SK_EPI_GRU_GATES;
        j.bias = bU;
        j.e0 = h; j.lde0 = D;
        j.o1 = d.z; j.ldo1 = D; j.o2 = d.r; j.ldo2 = D; j.out = d.rh; j.ldo = D;
        SR_TRY(sk_make_launch(L, &j, 1));
        SR_TRY(sk_launch(L, st));
        sk_job_init(j);
        j.nseg = 2;
        j.seg[0] = tl ? sk_seg(d.rh, D, t->Wc, blk, D, 2) : sk_seg(d.rh, D, Wc, D, D, 0);
        j.seg[1] = tl ? sk_seg(x, D, t->U + (size_t)(2 * D >> 4) * blk, blk, D, 2) : sk_seg(x, D, U + 2 * D, 3 * D, D, 0);
        j.M = d.B; j.N = D; j.H = D; j.epi = SK_EPI_GRU_CAND;
        j.bias = bU + 2 * D;
        j.e0 = h; j.lde0 = D; j.e1 = d.z; j.lde1 = D;
        j.o1 = nullptr; j.out = h; j.ldo = D;
        SR_TRY(sk_make_launch(L, &j, 1));
        return sk_launch(L, st);
    }

    // LSTM step of a tier layer (ops.py:505-553): pre = x . Win + b; gates = s . Wrec + pre (i | f | o | g);
    // c' = c*f + g*i; s' = tanh(c')*o.  s is the A operand of the step GEMM, so s' goes to a temporary first.
    int lstm_layer(const float* x, const float* Win, const float* b, const float* Wrec, float* s, float* c, hipStream_t st) {
        const int D = d.D;
        SR_TRY(linear(x, D, Win, 4 * D, D, 4 * D, b, nullptr, 0, d.P, 4 * D, 0, st));
        SkJob j;
        SkLaunch L;
        sk_job_init(j);
        j.nseg = 1;
        j.seg[0] = sk_seg(s, D, Wrec, 4 * D, D, 0);
        j.M = d.B; j.N = 4 * D; j.H = D; j.epi = SK_EPI_LSTM;
        j.add = d.P; j.ld_add = 4 * D;
        j.e1 = c; j.lde1 = D;
        j.o1 = c; j.ldo1 = D;
        j.o2 = d.gate_ws; j.ldo2 = 4 * D;
        j.out = d.layer_tmp; j.ldo = D;
        SR_TRY(sk_make_launch(L, &j, 1));
        SR_TRY(sk_launch(L, st));
        return (int)hipMemcpyAsync(s, d.layer_tmp, (size_t)d.B * D * sizeof(float), hipMemcpyDeviceToDevice, st);
    }

    // One step of a tier's RNN stack on input x [B,D]; returns the top layer's output (state) pointer.
    int stack_step(bool big, const float* x, const float** top, hipStream_t st) {
        if (d.n_rnn == 0) {  // single GRU layer, original fields
            if (big) SR_TRY(gru(x, d.big_U, d.big_bU, d.big_Wg, d.big_Wc, d.big_h, st, &t_big));
            else SR_TRY(gru(x, d.frm_U, d.frm_bU, d.frm_Wg, d.frm_Wc, d.frm_h, st, &t_frm));
            *top = big ? d.big_h : d.frm_h;
            return 0;
        }
        for (int k = 0; k < d.n_rnn; ++k) {
            const float* const* Lk = big ? d.big_L[k] : d.frm_L[k];
            float* hs = big ? d.big_hs[k] : d.frm_hs[k];
            if (d.lstm) SR_TRY(lstm_layer(x, Lk[0], Lk[1], Lk[2], hs, big ? d.big_cs[k] : d.frm_cs[k], st));
            else SR_TRY(gru(x, Lk[0], Lk[1], Lk[2], Lk[3], hs, st));
            x = hs;
        }
        *top = x;
        return 0;
    }

    int period(hipStream_t st) {
        const int B = d.B, D = d.D, FS = d.FS, BFS = d.BFS, len = BFS * d.T;
        const float half_q = (float)(d.Q / 2);
        const int nfr = BFS / FS;
        // ---- big-frame tier (three_tier.py:291-380), consumes samples[t-80:t] and features[t/80]
        {
            const int n = B * (BFS > d.feat_dim ? BFS : d.feat_dim);
            hipLaunchKernelGGL(sr_prep_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, d.samples, len, d.tbase, 0, BFS,
                               half_q, d.xf_big, B, d.features, d.feat_cur, d.feat_dim, BFS);
            SR_TRY(linear(d.xf_big, BFS, d.big_Win_frames, D, BFS, D, d.big_bin, nullptr, 0, d.gru_in, D, 0, st,
                          d.feat_cur, d.feat_dim, d.big_Win_feats, d.feat_dim));
            const float* top = nullptr;
            SR_TRY(stack_step(true, d.gru_in, &top, st));
            SR_TRY(linear(top, D, d.big_Wout, nfr * D, D, nfr * D, d.big_bout, nullptr, 0, d.big_out, nfr * D, 0, st,
                          nullptr, 0, nullptr, 0, t_big.Wout));
            if (winu) {  // the big tier's share of every frame's pre-activations: pbig[(b, f)] = big_out[b, f*D : (f+1)*D] . U +
                         // (bin . U + bU), one step-kernel launch with one job per frame (the LDS-tiled GEMM would put the
                         // [B nfr, 3D] product on 48 workgroups of 128 x 128 x K: measured 75 us per period)
                // big_out [B, nfr D] is a [B nfr, D] matrix of (stream, frame) rows and pbig [B nfr, 3D] likewise: jobs of up
                // to 64 consecutive rows each (the step kernel's tallest tile), every job streaming U once -- 12 MB x
                // B nfr / 64 per period where one job per frame (32 rows) read it nfr times
                SkJob jobs[SK_MAXJOB];
                const int blk = (D >> 4) * 256, rows = B * nfr, RJ = 64;
                for (int r0 = 0; r0 < rows; r0 += RJ * SK_MAXJOB) {
                    int nj = 0;
                    for (int r = r0; r < rows && nj < SK_MAXJOB; r += RJ, ++nj) {
                        SkJob& j = jobs[nj];
                        sk_job_init(j);
                        j.nseg = 1;
                        j.seg[0] = t_frm.U ? sk_seg(d.big_out + (size_t)r * D, D, t_frm.U, blk, D, 2)
                                           : sk_seg(d.big_out + (size_t)r * D, D, d.frm_U, 3 * D, D, 0);
                        j.M = rows - r < RJ ? rows - r : RJ; j.N = 3 * D; j.H = 3 * D; j.epi = SK_EPI_LINEAR;
                        j.bias = pbias;
                        j.out = pbig + (size_t)r * 3 * D; j.ldo = 3 * D;
                    }
                    SkLaunch L;
                    SR_TRY(sk_make_launch(L, jobs, nj));
                    SR_TRY(sk_launch(L, st));
                }
            }
        }
        for (int f = 0; f < nfr; ++f) {
            // ---- frame tier (three_tier.py:382-450), consumes samples[t-10:t] and big_out[:, (t/10)%8]
            const int toff = f * FS;
            // (on the persistent path the previous frame's sample kernel has already left this frame's input in gru_in)
            const float* ftop = nullptr;
            if (winu) {  // composed input: pin = xf . (Win . U) + pbig[f]  ([B, 3D]); the step GEMMs walk K = D
                // (persistent path: h . Wg of this frame's gates was made beside the previous frame's projection -- the last
                // frame of the previous period, or run()'s launch in front of the first one -- so the input kernel of a
                // period's first frame finishes the gates as the sample kernel does for the other frames)
                if (!persist)
                    hipLaunchKernelGGL(sr_frame_in_kernel, dim3(ceil_div(3 * D, 256), B), dim3(256), 0, st, d.samples, len,
                                       d.tbase, toff, FS, half_q, winu, (const float*)nullptr, pbig + (size_t)f * 3 * D,
                                       nfr * 3 * D, pin, 3 * D);
                else if (f == 0)
                    hipLaunchKernelGGL(sr_frame_in_kernel, dim3(ceil_div(3 * D, 256), B), dim3(256), 0, st, d.samples, len,
                                       d.tbase, toff, FS, half_q, winu, (const float*)nullptr, pbig + (size_t)f * 3 * D,
                                       nfr * 3 * D, pin, 3 * D, (const float*)gpre, (const float*)d.frm_h, d.z, d.rh, D);
                SR_TRY(gru(nullptr, d.frm_U, d.frm_bU, d.frm_Wg, d.frm_Wc, d.frm_h, st, &t_frm, pin, persist));
                ftop = d.frm_h;
            } else {
                if (!persist || f == 0)
                    hipLaunchKernelGGL(sr_frame_in_kernel, dim3(ceil_div(D, 256), B), dim3(256), 0, st, d.samples, len, d.tbase,
                                       toff, FS, half_q, d.frm_Win, d.frm_bin, d.big_out + (size_t)f * D, nfr * D, d.gru_in, D);
                SR_TRY(stack_step(false, d.gru_in, &ftop, st));
            }
            if (persist && winu) {  // (last frame of the period: the gates' product of the next period's first frame)
                // the projection and, beside it, the recurrent product of the NEXT frame's gates (h' . Wg: it does not wait for
                // the frame's samples); the sample kernel finishes those gates at its end
                SkJob jobs[2];
                const int blk = (D >> 4) * 256;
                sk_job_init(jobs[0]);
                jobs[0].nseg = 1;
                jobs[0].seg[0] = t_frm.Wout ? sk_seg(ftop, D, t_frm.Wout, blk, D, 2) : sk_seg(ftop, D, Pout, FS * D, D, 0);
                jobs[0].M = B; jobs[0].N = FS * D; jobs[0].H = FS * D; jobs[0].epi = SK_EPI_LINEAR;
                jobs[0].bias = cb; jobs[0].out = d.frame_out; jobs[0].ldo = FS * D;
                sk_job_init(jobs[1]);
                jobs[1].nseg = 1;
                jobs[1].seg[0] = t_frm.Wg ? sk_seg(ftop, D, t_frm.Wg, blk, D, 2) : sk_seg(ftop, D, d.frm_Wg, 2 * D, D, 0);
                jobs[1].M = B; jobs[1].N = 2 * D; jobs[1].H = 2 * D; jobs[1].epi = SK_EPI_LINEAR;
                jobs[1].out = gpre; jobs[1].ldo = 2 * D;
                SkLaunch L;
                SR_TRY(sk_make_launch(L, jobs, 2));
                // (FS + 2) D columns = 768 column tiles at D = 1024: 16-column workgroups are three per CU; the heuristic's
                // 32-column ones (384) leave half the CUs with two streams and half with one (8.17 vs 8.07 us per sample)
                L.force_tile = 21;
                SR_TRY(sk_launch(L, st));
            } else {
                SR_TRY(linear(ftop, D, persist ? Pout : d.frm_Wout, FS * D, D, FS * D, persist ? cb : d.frm_bout, nullptr, 0,
                              d.frame_out, FS * D, 0, st, nullptr, 0, nullptr, 0, t_frm.Wout));
            }
            if (persist) {
                // ---- all FS sample steps of this frame in one launch: XCD-local persistent-thread kernel
                SrpArgs sa{};
                sa.tbase = d.tbase; sa.toff = toff; sa.samples = d.samples; sa.len = len;
                sa.B = B; sa.D = D; sa.Q = d.Q; sa.FS = FS; sa.nsteps = FS;
                sa.t2tbl = t2tbl; sa.frame_out = d.frame_out; sa.ldf = FS * D;
                sa.W3 = d.W3; sa.b3 = d.b3; sa.W4 = d.W4; sa.b4 = d.b4;
                sa.logits = d.logits; sa.ws = d.persist_ws; sa.temperature = d.temperature; sa.seed = d.seed;
                if (f + 1 < nfr && winu) {  // the next frame of this period: its big-tier share is already known
                    sa.next_in = pin; sa.next_Win = winu; sa.next_bias = nullptr;
                    sa.next_add = pbig + (size_t)(f + 1) * 3 * D; sa.next_ld_add = nfr * 3 * D; sa.next_n = 3 * D;
                    sa.next_gpre = gpre; sa.next_h = d.frm_h; sa.next_z = d.z; sa.next_rh = d.rh;
                } else if (f + 1 < nfr) {
                    sa.next_in = d.gru_in; sa.next_Win = d.frm_Win; sa.next_bias = d.frm_bin;
                    sa.next_add = d.big_out + (size_t)(f + 1) * D; sa.next_ld_add = nfr * D;
                }
                {
                    static const int timing = getenv("PARROT_SR_TIMING") ? atoi(getenv("PARROT_SR_TIMING")) : 0;
                    sa.pad = timing;
                }
                SR_TRY(srp_launch(sa, st));
                continue;
            }
            for (int i = 0; i < FS; ++i) {
                // ---- sample-level MLP (three_tier.py:452-515) + pick (ops.py:268-297)
                const int to = toff + i;
                hipLaunchKernelGGL(sr_embed_sum_kernel, dim3(ceil_div(D / 4, 256), B), dim3(256), 0, st, d.samples, len,
                                   d.tbase, to, FS, d.Q, D, d.emb_tbl, d.frame_out + (size_t)i * D, FS * D, d.o1);
                SR_TRY(linear(d.o1, D, d.W2, D, D, D, d.b2, nullptr, 0, d.o2, D, SK_ACT_RELU, st));
                SR_TRY(linear(d.o2, D, d.W3, D, D, D, d.b3, nullptr, 0, d.o3, D, SK_ACT_RELU, st));
                SR_TRY(linear(d.o3, D, d.W4, d.Q, D, d.Q, d.b4, nullptr, 0, d.logits, d.Q, 0, st));
                hipLaunchKernelGGL(sr_pick_kernel, dim3(B), dim3(256), 0, st, d.logits, d.Q, d.samples, len, d.tbase, to,
                                   d.temperature, d.seed);
            }
        }
        hipLaunchKernelGGL(sr_tick_kernel, dim3(1), dim3(64), 0, st, d.tbase, BFS, 0);
        return (int)hipGetLastError();
    }

    int run(hipStream_t st) {
        // tbase starts at BFS (first 80 samples are Q_ZERO, set by the caller); T-1 periods follow.
        hipLaunchKernelGGL(sr_tick_kernel, dim3(1), dim3(64), 0, st, d.tbase, d.BFS, 1);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
        const int periods = d.T - 1;
        if (persist && winu)  // h0 . Wg: the gates' recurrent product of the very first frame (see period())
            SR_TRY(linear(d.frm_h, d.D, d.frm_Wg, 2 * d.D, d.D, 2 * d.D, nullptr, nullptr, 0, gpre, 2 * d.D, 0, st, nullptr, 0,
                          nullptr, 0, t_frm.Wg));
        if (!d.use_graph) {
            for (int p = 0; p < periods; ++p) SR_TRY(period(st));
            return 0;
        }
        // Two graphs: one period, and SR_CHUNK periods back to back (a graph launch costs ~9 us of idle GPU between two
        // periods -- 0.1 us per sample; the chunk graph pays it once per SR_CHUNK periods).  tbase advances on the device.
        auto capture = [&](int n, hipGraphExec_t* out) -> int {
            if (!cap) {
                hipError_t ce = hipStreamCreateWithFlags(&cap, hipStreamNonBlocking);
                if (ce != hipSuccess) return (int)ce;
            }
            hipError_t ce = hipStreamBeginCapture(cap, hipStreamCaptureModeRelaxed);
            if (ce != hipSuccess) return (int)ce;
            int rc = 0;
            for (int q = 0; q < n && rc == 0; ++q) rc = period(cap);
            hipGraph_t graph = nullptr;
            ce = hipStreamEndCapture(cap, &graph);
            if (rc != 0) { if (graph) hipGraphDestroy(graph); return rc; }
            if (ce != hipSuccess) return (int)ce;
            ce = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
            hipGraphDestroy(graph);
            if (ce != hipSuccess) { *out = nullptr; return (int)ce; }
            return 0;
        };
        if (!exec) SR_TRY(capture(1, &exec));
        if (periods >= SR_CHUNK && !exec_chunk) SR_TRY(capture(SR_CHUNK, &exec_chunk));
        int left = periods;
        for (; exec_chunk && left >= SR_CHUNK; left -= SR_CHUNK) {
            e = hipGraphLaunch(exec_chunk, st);
            if (e != hipSuccess) return (int)e;
        }
        periods_left_single = left;
        for (int p = 0; p < periods_left_single; ++p) {
            e = hipGraphLaunch(exec, st);
            if (e != hipSuccess) return (int)e;
        }
        return 0;
    }
};

}  // namespace

extern "C" {

int samplernn_generate_create(const SampleRnnGenDesc* desc, void** plan) { PH_ENTRY();
    if (!desc || !plan || desc->B < 1 || desc->T < 2 || desc->D < 4 || (desc->D & 3) || desc->FS < 1 ||
        desc->BFS % desc->FS != 0 || desc->Q < 2 || desc->n_rnn < 0 || desc->n_rnn > 5)
        return PARROT_ERR_BADARG;
    if (desc->n_rnn > 0) {
        if (desc->lstm && (!desc->gate_ws || !desc->layer_tmp)) return PARROT_ERR_BADARG;
        for (int k = 0; k < desc->n_rnn; ++k) {
            if (!desc->big_hs[k] || !desc->frm_hs[k] || (desc->lstm && (!desc->big_cs[k] || !desc->frm_cs[k])))
                return PARROT_ERR_BADARG;
            for (int q = 0; q < (desc->lstm ? 3 : 4); ++q)
                if (!desc->big_L[k][q] || !desc->frm_L[k][q]) return PARROT_ERR_BADARG;
        }
    }
    SrPlan* p = new (std::nothrow) SrPlan();
    if (!p) return PARROT_ERR_BADARG;
    p->d = *desc;
    p->persist = desc->persist_ws && srp_eligible(desc->B, desc->D, desc->Q, desc->FS) &&
                 desc->persist_ws_floats >= srp_ws_floats(desc->D, desc->Q) && srp_prepare(desc->D) == 0 &&
                 srp_init_ws(desc->persist_ws, desc->D, desc->Q) == 0 && p->make_composed() == 0;
    p->make_tiled();  // (after the decision: the persistent path tiles the composed projection)
    p->make_winu();
    *plan = p;
    return 0;
}

long long samplernn_persist_floats(const SampleRnnGenDesc* desc) { PH_ENTRY();
    if (!desc || !srp_eligible(desc->B, desc->D, desc->Q, desc->FS)) return 0;
    return srp_ws_floats(desc->D, desc->Q);
}

int samplernn_generate_is_persistent(void* plan) { PH_ENTRY();
    return plan && static_cast<SrPlan*>(plan)->persist ? 1 : 0;
}

int samplernn_generate_status(void* plan) { PH_ENTRY();
    SrPlan* p = static_cast<SrPlan*>(plan);
    if (!p) return PARROT_ERR_BADARG;
    const hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return (int)e;
    if (p->last_error) return p->last_error;
    return p->persist ? srp_status(p->d.persist_ws) : 0;
}

int samplernn_generate_run(void* plan, void* stream) { PH_ENTRY();
    SrPlan* p = static_cast<SrPlan*>(plan);
    const int rc = p->run((hipStream_t)stream);
    if (rc != 0 && p->last_error == 0) p->last_error = rc;
    return rc;
}

int samplernn_generate_destroy(void* plan) { PH_ENTRY();
    delete static_cast<SrPlan*>(plan);
    return 0;
}

}  // extern "C"
