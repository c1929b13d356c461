// extern "C" entry points that are single operations (the scan plans live in plans.hip).
#include "../../include/parrot_hip.h"
#include "attention.h"
#include "biggemm.h"
#include "elementwise.h"
#include "quantize.h"
#include "skinny.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <map>
#include <mutex>

static int gemm_target_wgs() {
    return 1024;  // 8-wave kernel: 1024..4096 measure the same; fewer slices = less partial-sum traffic
}

// parrot_set_gemm_precision (process-wide; plans override per thread).  Default since round 6: the split-bf16 products
// (PARROT_PRECISION_BF16X3: f32 operands, f32-grade results, 1.7-1.9 x the f32-input MFMA kernel's rate);
// PARROT_GEMM_PRECISION=f32 in the environment restores the f32-input matrix instructions for everything.
static int gemm_default_mode() {
    const char* e = getenv("PARROT_GEMM_PRECISION");
    if (e && (!strcmp(e, "f32") || !strcmp(e, "0"))) return PARROT_PRECISION_F32;
    if (e && (!strcmp(e, "bf16") || !strcmp(e, "1"))) return PARROT_PRECISION_BF16;
    return PARROT_PRECISION_BF16X3;
}
static std::atomic<int> g_gemm_bf16{gemm_default_mode()};
static thread_local int t_gemm_bf16 = -1;   // BgPrecisionScope (-1: follow the process-wide mode)
BgPrecisionScope::BgPrecisionScope(int bf16) : saved(t_gemm_bf16) { t_gemm_bf16 = bf16 < 0 ? saved : (bf16 ? 1 : 0); }
BgPrecisionScope::~BgPrecisionScope() { t_gemm_bf16 = saved; }

extern "C" {

int parrot_set_gemm_precision(int mode) { PH_ENTRY();
    if (mode != PARROT_PRECISION_F32 && mode != PARROT_PRECISION_BF16 && mode != PARROT_PRECISION_BF16X3) return PARROT_ERR_BADARG;
    g_gemm_bf16.store(mode, std::memory_order_relaxed);
    return 0;
}

int parrot_get_gemm_precision(void) { PH_ENTRY(); return g_gemm_bf16.load(std::memory_order_relaxed); }

const char* parrot_hip_version(void) { PH_ENTRY(); return "parrot_hip 0.5.0 gfx950"; }

int parrot_profile_begin(void) { PH_ENTRY();
    sk_profile_begin();
    return 0;
}

long long parrot_profile_end(double* total_us, double* flops, double* bytes) { PH_ENTRY();
    return sk_profile_end(total_us, flops, bytes);
}

long long parrot_profile_end2(double* total_us, double* flops, double* bytes, double* plain4) { PH_ENTRY();
    return sk_profile_end2(total_us, flops, bytes, plain4);
}

// Split-K workspace, one per stream: products on different streams (the weight-gradient GEMMs that run beside the
// backward scan, model.py's _backward) never share partial tiles.  Grows on demand, reused by later calls in stream
// order.  Under stream capture (or when the memory cannot be had) there is none and the product runs unsplit, so no
// graph ever holds a pointer into a workspace that a later, larger call may replace.
static float* bg_workspace(hipStream_t st, size_t need) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cs);
    if (cs != hipStreamCaptureStatusNone) return nullptr;
    struct Ws { float* p; size_t floats; };
    static std::mutex ws_mu;
    static std::map<hipStream_t, Ws> ws_of;
    std::lock_guard<std::mutex> lock(ws_mu);
    Ws& w = ws_of[st];
    if (need > w.floats) {
        if (w.p) {
            (void)hipStreamSynchronize(st);
            (void)hipFree(w.p);
            w.p = nullptr;
            w.floats = 0;
        }
        if (hipMalloc(&w.p, need * sizeof(float)) == hipSuccess) w.floats = need;
        else w.p = nullptr;
    }
    return (w.p && need <= w.floats) ? w.p : nullptr;
}

// Auto split-K, the deterministic split-K workspace and the launch of one batched product (shared by parrot_gemm and
// parrot_gemm_bf16in).
static int bg_run(BgArgs a, int split_k, hipStream_t st) {
    const int M = a.M, N = a.N, K = a.K, nbatch = a.nbatch, act = a.act;
    const float* bias = a.bias;
    int split = split_k;
    if (split <= 0) {
        // auto: few output tiles but a long reduction (deferred weight gradients: K = T*B rows)
        // -> spread K over enough workgroups to fill 256 CUs.
        int bm, bn;
        bg_tile_shape(a.bf16, bm, bn);
        const long long tiles = (long long)ceil_div(M, bm) * ceil_div(N, bn) * nbatch;
        split = 1;
        if (a.bf16 == 3) {
            // 256 x 256 tiles, one workgroup per CU, slices dealt to XCDs: a multiple of 8 slices whose workgroups fill
            // whole rounds of 256 best, at least 512 K rows per slice
            if (act == 0 && tiles < 256 && K >= 4096) {
                double best = -1.0;
                for (int sp = 8; sp <= 64 && K / sp >= 512; sp += 8) {
                    const long long wgs = tiles * sp;
                    const double eff = (double)wgs / (double)((wgs + 255) / 256 * 256);
                    const double score = eff - 0.001 * sp;
                    if (score > best) { best = score; split = sp; }
                }
            }
        } else if (a.bf16 == 2 && K >= 512 && tiles < 256) {
            // (tiles >= 256: the output tiles fill the chip by themselves -- the M = T*B readout products -- and a split
            // would only add a partial-sum pass larger than the operands: ADVICE r05)
            // 256 x 256 tiles, one workgroup per CU: the slice count whose workgroups fill whole rounds of 256 best
            // (at least two rounds, at most 16 slices, at least 1024 K rows per slice)
            double best = -1.0;
            for (int sp = 1; sp <= 16 && K / sp >= 1024; ++sp) {
                const long long wgs = tiles * sp;
                const double eff = (double)wgs / (double)((wgs + 255) / 256 * 256);
                const double score = eff - (wgs < 512 ? 0.25 : 0.0) - 0.002 * sp;
                if (score > best) { best = score; split = sp; }
            }
        } else if (act == 0 && tiles < 512 && K >= 512) {
            split = (int)((gemm_target_wgs() + tiles - 1) / tiles);
            const int maxs = K / 256 > 0 ? K / 256 : 1;
            if (split > maxs) split = maxs;
            if (split > 64) split = 64;
        }
    }
    a.splitk = split;
    a.ws = nullptr;
    if (a.splitk > 1) {
        if (act != 0) return PARROT_ERR_BADARG;
        // Deterministic split-K: the slices write partial tiles to a library-owned workspace and a second kernel adds
        // them in slice order (results do not depend on scheduling).  The workspace grows on demand and is reused by
        // later calls in stream order.  Under stream capture or when the workspace cannot be had, the product runs
        // unsplit instead: there is no float-atomic combine any more.
        float* ws = bg_workspace(st, (size_t)nbatch * a.splitk * M * N);
        if (ws) {
            a.ws = ws;
            a.bias = nullptr;  // the reducer adds it
            int rc = bg_launch(a, st);
            if (rc) return rc;
            a.bias = bias;
            return bg_reduce_launch(a, st);
        }
        a.splitk = 1;
    }
    return bg_launch(a, st);
}

static int gemm_impl(const float* A, int lda, int transA, const float* B, int ldb, int transB, float* C, int ldc,
                     int M, int N, int K, const float* bias, float alpha, int accumulate, int act, int nbatch,
                     long long strideA, long long strideB, long long strideC, int split_k, const float* gate, int ldg,
                     void* stream) {
    if (!A || !B || !C || M < 1 || N < 1 || K < 1 || nbatch < 1) return PARROT_ERR_BADARG;
    if (gate && (nbatch != 1 || ldg < N)) return PARROT_ERR_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (M <= 64 && !transA && nbatch == 1 && split_k <= 1 && alpha == 1.0f && !gate) {
        SkJob j;
        sk_job_init(j);
        j.nseg = 1;
        j.seg[0] = sk_seg(A, lda, B, ldb, K, transB ? 1 : 0);
        j.M = M; j.N = N; j.H = N; j.epi = SK_EPI_LINEAR; j.act = act; j.accumulate = accumulate;
        j.bias = bias;
        j.out = C; j.ldo = ldc;
        SkLaunch L;
        int rc = sk_make_launch(L, &j, 1);
        if (rc) return rc;
        return sk_launch(L, st);
    }
    BgArgs a;
    a.A = A; a.B = B; a.C = C; a.bias = bias;
    a.M = M; a.N = N; a.K = K;
    a.sam = transA ? 1 : lda; a.sak = transA ? lda : 1;
    a.sbk = transB ? 1 : ldb; a.sbn = transB ? ldb : 1;
    a.ldc = ldc;
    a.batchA = strideA; a.batchB = strideB; a.batchC = strideC;
    a.nbatch = nbatch;
    a.accumulate = accumulate; a.alpha = alpha; a.act = act;
    a.gate = gate; a.ldg = ldg;
    a.bf16 = t_gemm_bf16 >= 0 ? t_gemm_bf16 : g_gemm_bf16.load(std::memory_order_relaxed);
    if (a.bf16 == PARROT_PRECISION_BF16X3) {
        // split-bf16 products (bgs_kernel): f32 operands and f32-grade results on the bf16 matrix pipe.  Taken where the
        // 256 x 256 tile pays: both operands fetched as aligned 16-byte vectors, enough work; the rest stays on the
        // f32-input MFMA kernel.
        auto al = [](const float* p, long long ld) { return (((uintptr_t)p & 15) == 0) && (ld % 4 == 0); };
        const bool ok = act <= 1 && al(A, lda) && al(B, ldb) && (strideA % 4 == 0) && (strideB % 4 == 0) && M >= 128 && N >= 128 &&
                        K >= 64 && (!transA || M % 4 == 0) && (transA || K % 4 == 0) && (transB || N % 4 == 0) && (!transB || K % 4 == 0);
        a.bf16 = ok ? 3 : 0;
    }
    return bg_run(a, split_k, st);
}

int parrot_gemm(const float* A, int lda, int transA, const float* B, int ldb, int transB, float* C, int ldc,
                int M, int N, int K, const float* bias, float alpha, int accumulate, int act, int nbatch,
                long long strideA, long long strideB, long long strideC, int split_k, void* stream) { PH_ENTRY();
    return gemm_impl(A, lda, transA, B, ldb, transB, C, ldc, M, N, K, bias, alpha, accumulate, act, nbatch, strideA, strideB,
                     strideC, split_k, nullptr, 0, stream);
}

int parrot_gemm_gated(const float* A, int lda, int transA, const float* B, int ldb, int transB, float* C, int ldc,
                      int M, int N, int K, const float* gate, int ldg, void* stream) { PH_ENTRY();
    if (!gate) return PARROT_ERR_BADARG;
    return gemm_impl(A, lda, transA, B, ldb, transB, C, ldc, M, N, K, nullptr, 1.0f, 0, 0, 1, 0, 0, 0, 0, gate, ldg, stream);
}

int parrot_to_bf16(const float* x, void* y, long long n, void* stream) { PH_ENTRY();
    if (!x || !y || n < 0) return PARROT_ERR_BADARG;
    return bg_to_bf16_launch(x, y, n, (hipStream_t)stream);
}

int parrot_gemm_bf16in_ex(const void* A, int lda, int transA, const void* B, int ldb, int transB, float* C, int ldc, int M,
                          int N, int K, const float* bias, int accumulate, int split_k, void* stream) { PH_ENTRY();
    if (!A || !B || !C || M < 1 || N < 1 || K < 1) return PARROT_ERR_BADARG;
    BgArgs a;
    a.A = reinterpret_cast<const float*>(A); a.B = reinterpret_cast<const float*>(B); a.C = C; a.bias = bias;
    a.M = M; a.N = N; a.K = K;
    a.sam = transA ? 1 : lda; a.sak = transA ? lda : 1;   // as parrot_gemm, in bf16 elements
    a.sbk = transB ? 1 : ldb; a.sbn = transB ? ldb : 1;
    a.ldc = ldc;
    a.batchA = a.batchB = a.batchC = 0;
    a.nbatch = 1;
    a.accumulate = accumulate; a.alpha = 1.0f; a.act = 0;
    a.gate = nullptr; a.ldg = 0;
    a.bf16 = 2;
    return bg_run(a, split_k, (hipStream_t)stream);
}

int parrot_gemm_bf16in(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K,
                       int accumulate, int split_k, void* stream) {
    // A(m, k) at A[k * lda + m]: the caller's [K, M] matrix read as its transpose
    return parrot_gemm_bf16in_ex(A, lda, 1, B, ldb, 0, C, ldc, M, N, K, nullptr, accumulate, split_k, stream);
}

int parrot_tile_weights(const float* W, int rows, int cols, int ld, float* out, int mode, int lstm_H, void* stream) { PH_ENTRY();
    if (mode != 0 && mode != 1) return PARROT_ERR_BADARG;
    return sk_tile_weights_launch(W, rows, cols, ld, out, mode, lstm_H, (hipStream_t)stream);
}

int parrot_tile_weights_bf16(const float* W, int rows, int cols, int ld, void* out, int mode, int lstm_H, void* stream) { PH_ENTRY();
    if (mode != 0 && mode != 1) return PARROT_ERR_BADARG;
    return sk_tile_weights_bf16_launch(W, rows, cols, ld, out, mode, lstm_H, (hipStream_t)stream);
}

int parrot_simple_norm_fwd(const float* x, int ldx, float* y, int ldy, float* sigma, long long R, int N, float eps,
                           float* add_dst, int ld_add, void* stream) { PH_ENTRY();
    if (!x || !y || !sigma) return PARROT_ERR_BADARG;
    return simple_norm_fwd_launch(x, ldx, y, ldy, sigma, R, N, eps, add_dst, ld_add, (hipStream_t)stream);
}

int parrot_simple_norm_bwd(const float* dy, int lddy, const float* y, int ldy, const float* sigma, float* dx,
                           int lddx, long long R, int N, float eps, int accumulate, void* stream) { PH_ENTRY();
    if (!dy || !y || !sigma || !dx) return PARROT_ERR_BADARG;
    return simple_norm_bwd_launch(dy, lddy, y, ldy, sigma, dx, lddx, R, N, eps, accumulate, (hipStream_t)stream);
}

int parrot_colsum(const float* x, long long M, int N, int ld, float* out, int accumulate, void* stream) { PH_ENTRY();
    return colsum_launch(x, M, N, ld, out, accumulate, (hipStream_t)stream);
}

int parrot_gru_step_fwd(const float* h, const float* inputs, const float* gate_inputs, const float* mask,
                        const float* Wg, const float* Wc, float* h_out, float* z, float* r, float* rh,
                        float* c, int B, int H, void* stream) { PH_ENTRY();
    if (!h || !Wg || !Wc || !h_out || !z || !r || !rh || B < 1 || H < 1) return PARROT_ERR_BADARG;
    hipStream_t st = (hipStream_t)stream;
    SkJob j;
    SkLaunch L;
    sk_job_init(j);
    j.nseg = 1;
    j.seg[0] = sk_seg(h, H, Wg, 2 * H, H, 0);
    j.M = B; j.N = 2 * H; j.H = H; j.epi = SK_EPI_GRU_GATES;
    j.add = gate_inputs; j.ld_add = 2 * H;
    j.e0 = h; j.lde0 = H;
    j.o1 = z; j.ldo1 = H; j.o2 = r; j.ldo2 = H; j.out = rh; j.ldo = H;
    int rc = sk_make_launch(L, &j, 1);
    if (rc) return rc;
    rc = sk_launch(L, st);
    if (rc) return rc;
    sk_job_init(j);
    j.nseg = 1;
    j.seg[0] = sk_seg(rh, H, Wc, H, H, 0);
    j.M = B; j.N = H; j.H = H; j.epi = SK_EPI_GRU_CAND;
    j.add = inputs; j.ld_add = H;
    j.e0 = h; j.lde0 = H; j.e1 = z; j.lde1 = H;
    j.o1 = c; j.ldo1 = H; j.out = h_out; j.ldo = H; j.mask = mask;
    rc = sk_make_launch(L, &j, 1);
    if (rc) return rc;
    return sk_launch(L, st);
}

int parrot_gru_step_bwd(const float* dh_out, const float* h, const float* mask, const float* Wg,
                        const float* Wc, const float* z, const float* r, const float* c, float* dh,
                        float* d_inputs, float* d_gate_inputs, int B, int H, void* stream) { PH_ENTRY();
    if (!dh_out || !h || !Wg || !Wc || !z || !r || !c || !dh || !d_inputs || !d_gate_inputs)
        return PARROT_ERR_BADARG;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(dh, 0, sizeof(float) * (size_t)B * H, st);
    if (e != hipSuccess) return (int)e;
    GruStateBwdArgs ga;
    ga.nchain = 1; ga.B = B; ga.H = H;
    ga.chain[0].dh = dh_out; ga.chain[0].dh2 = nullptr; ga.chain[0].hprev = h; ga.chain[0].z = z; ga.chain[0].c = c;
    ga.chain[0].mask = mask; ga.chain[0].dC = d_inputs; ga.chain[0].dG = d_gate_inputs; ga.chain[0].dhprev = dh;
    int rc = gru_state_bwd_launch(ga, st);
    if (rc) return rc;
    SkJob j;
    SkLaunch L;
    sk_job_init(j);
    j.nseg = 1;
    j.seg[0] = sk_seg(d_inputs, H, Wc, H, H, 1);
    j.M = B; j.N = H; j.H = H; j.epi = SK_EPI_BWD_RH;
    j.e0 = h; j.lde0 = H; j.e1 = r; j.lde1 = H;
    j.out = d_gate_inputs + H; j.ldo = 2 * H; j.o1 = dh; j.ldo1 = H;
    rc = sk_make_launch(L, &j, 1);
    if (rc) return rc;
    rc = sk_launch(L, st);
    if (rc) return rc;
    sk_job_init(j);
    j.nseg = 1;
    j.seg[0] = sk_seg(d_gate_inputs, 2 * H, Wg, 2 * H, 2 * H, 1);
    j.M = B; j.N = H; j.H = H; j.epi = SK_EPI_LINEAR; j.accumulate = 1;
    j.out = dh; j.ldo = H;
    rc = sk_make_launch(L, &j, 1);
    if (rc) return rc;
    return sk_launch(L, st);
}

int parrot_gmm_attention_fwd(const float* h1, const float* WattT, const float* batt, const float* kappa_prev,
                             const float* ctx, float* a, float* b, float* kappa, float* phi, float* w, int B,
                             int H, int A, int U, int E, int att_type, float eps, float alignment,
                             float sharpening, float timing, void* stream) { PH_ENTRY();
    if (!h1 || !WattT || !kappa_prev || !ctx || !a || !b || !kappa || !phi || !w) return PARROT_ERR_BADARG;
    AttFwdArgs g{};
    g.h1 = h1; g.ldh = H; g.WattT = WattT; g.batt = batt; g.kappa_prev = kappa_prev; g.ctx = ctx;
    g.a_out = a; g.b_out = b; g.kappa_out = kappa; g.phi_out = phi; g.w_out = w; g.ldw = E;
    g.B = B; g.H = H; g.A = A; g.U = U; g.E = E; g.esplit = att_default_esplit(B, E);
    g.att_type = att_type; g.eps = eps; g.alignment = alignment; g.sharpening = sharpening; g.timing = timing;
    return att_fwd_launch(g, (hipStream_t)stream);
}

int parrot_gmm_attention_bwd(const float* dw, const float* ctx, const float* a, const float* b,
                             const float* kappa, const float* kappa_prev, const float* WattT, float* dkappa,
                             float* dp, float* dh1, int B, int H, int A, int U, int E, int att_type, float eps,
                             void* stream) { PH_ENTRY();
    if (!dw || !ctx || !a || !b || !kappa || !kappa_prev || !WattT || !dkappa || !dp || !dh1)
        return PARROT_ERR_BADARG;
    AttBwdArgs g{};
    g.dw = const_cast<float*>(dw); g.dw2 = nullptr; g.lddw = E; g.ctx = ctx; g.a = a; g.b = b; g.kappa = kappa; g.kappa_prev = kappa_prev;
    g.WattT = WattT; g.dkappa = dkappa; g.dp_out = dp; g.dh1 = dh1; g.lddh = H;
    g.B = B; g.H = H; g.A = A; g.U = U; g.E = E; g.att_type = att_type; g.eps = eps;
    return att_bwd_launch(g, (hipStream_t)stream);
}

int parrot_sumsq(const float* x, size_t n, float* out, void* stream) { PH_ENTRY();
    return sumsq_launch(x, n, out, (hipStream_t)stream);
}

int parrot_adam_clip_step(float* param, const float* grad, float* m, float* v, size_t n,
                          const float* gnorm_sq, float grad_scale, float clip_threshold, float lr,
                          float beta1, float beta2, float eps, int step, void* stream) { PH_ENTRY();
    if (!param || !grad || !m || !v || step < 1) return PARROT_ERR_BADARG;
    if (clip_threshold > 0.f && !gnorm_sq) return PARROT_ERR_BADARG;
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, step)) / (1.0 - pow((double)beta1, step));
    return adam_clip_launch(param, grad, m, v, n, gnorm_sq, grad_scale, clip_threshold, (float)lr_t, beta1,
                            beta2, eps, (hipStream_t)stream);
}

int parrot_batch_quantize(const float* x, int rows, int n, int ld, double* ws, void* out, int ldo, int mode,
                          int q_levels, void* stream) { PH_ENTRY();
    if (!x || !ws || !out) return PARROT_ERR_BADARG;
    return quantize_launch(x, rows, n, ld, ws, out, ldo, mode, q_levels, (hipStream_t)stream);
}

int parrot_mu2linear(const int32_t* q, size_t n, float* out, void* stream) { PH_ENTRY();
    if (!q || !out) return PARROT_ERR_BADARG;
    return mu2linear_launch(q, n, out, (hipStream_t)stream);
}

}  // extern "C"
