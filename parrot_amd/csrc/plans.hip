// Host-side scan orchestration: the loops that theano.scan ran inside one theano.function call in
// the reference (model.py:726-737, 1038-1057; ops.py:299-327) become fixed launch sequences of the
// fused step kernels, captured once into a hipGraph and replayed per window (all pointers in a
// plan are fixed, so a replay costs one hipGraphLaunch instead of thousands of host launches).
#include <stdlib.h>

#include <string.h>

#include <algorithm>
#include <array>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/parrot_hip.h"
#include "attention.h"
#include "biggemm.h"
#include "elementwise.h"
#include "persist.h"
#include "rowgru.h"
#include "skinny.h"

namespace {

struct PlanBase {
    int last_error = 0;
    int use_graph = 0;
    hipGraphExec_t exec[2] = {nullptr, nullptr};
    hipStream_t cap_stream = nullptr;
    virtual ~PlanBase() {
        for (int i = 0; i < 2; ++i)
            if (exec[i]) hipGraphExecDestroy(exec[i]);
        if (cap_stream) hipStreamDestroy(cap_stream);
    }
    virtual int enqueue(int which, hipStream_t s) = 0;

    virtual int run(int which, hipStream_t s) {
        if (!use_graph) return note(enqueue(which, s));
        if (!exec[which]) {
            if (!cap_stream) {
                hipError_t e = hipStreamCreateWithFlags(&cap_stream, hipStreamNonBlocking);
                if (e != hipSuccess) return note((int)e);
            }
            hipError_t e = hipStreamBeginCapture(cap_stream, hipStreamCaptureModeRelaxed);
            if (e != hipSuccess) return note((int)e);
            const int rc = enqueue(which, cap_stream);
            hipGraph_t graph = nullptr;
            e = hipStreamEndCapture(cap_stream, &graph);
            if (rc != 0) {
                if (graph) hipGraphDestroy(graph);
                return note(rc);
            }
            if (e != hipSuccess) return note((int)e);
            e = hipGraphInstantiate(&exec[which], graph, nullptr, nullptr, 0);
            hipGraphDestroy(graph);
            if (e != hipSuccess) {
                exec[which] = nullptr;
                return note((int)e);
            }
        }
        return note((int)hipGraphLaunch(exec[which], s));
    }
    int note(int rc) {
        if (rc != 0 && last_error == 0) last_error = rc;
        return rc;
    }
};

#define PL_TRY(x)                \
    do {                         \
        const int rc__ = (x);    \
        if (rc__ != 0) return rc__; \
    } while (0)

// ---- schedule tracing (parrot_decoder_trace) ---------------------------------------------------------------------
// With a tracer installed the launch helpers below do not launch anything: they record, per launch and per job, the
// byte ranges the job reads, writes, reads-and-writes, or reads behind an in-launch flag.  tests/test_schedule_cpu.py
// runs every launch schedule through it on fake device addresses (no GPU) and checks the orderings a schedule must keep:
// no job of a launch touches what another job of the same launch writes, write-once buffers are read only after their
// writer's launch, accumulators are not written after their consumer has read them.
struct TraceRec { long long launch, job, kind, lo, hi; };  // kind: 0 read, 1 write, 2 read+write, 3 read behind a flag
struct TraceJob { long long launch, job, M, N, Kmax_seg_sum, epi; };  // shape of a step-GEMM job (job 100: the attention)
struct Tracer {
    std::vector<TraceRec> recs;
    std::vector<TraceJob> jobs;
    long long launch = -1;
    void begin() { ++launch; }
    void mat(const void* p, long long rows, long long cols, long long ld, int kind, int job, int elt = 4) {
        if (!p || rows < 1 || cols < 1) return;
        const long long base = (long long)(uintptr_t)p;
        if (ld == cols) { recs.push_back({launch, job, kind, base, base + rows * cols * elt}); return; }
        for (long long r = 0; r < rows; ++r) recs.push_back({launch, job, kind, base + r * ld * elt, base + (r * ld + cols) * elt});
    }
    void sk_job(const SkJob& j, int id) {
        long long ks = 0;
        for (int q = 0; q < j.nseg; ++q) ks += j.seg[q].K;
        jobs.push_back({launch, id, j.M, j.N, ks, j.epi});
        for (int q = 0; q < j.nseg; ++q)
            mat(j.seg[q].A, j.M, j.seg[q].K, j.seg[q].lda, (j.wait_flag && (j.wait_all || q == j.nseg - 1)) ? 3 : 0, id);
        mat(j.add, j.M, j.N, j.ld_add, 0, id);
        const int H = j.H;
        switch (j.epi) {
            case SK_EPI_LINEAR:
                mat(j.out, j.M, j.N, j.ldo, j.accumulate ? 2 : 1, id);
                if (j.ksplit > 1) mat(j.o1, j.M, j.N, j.ldo1, j.ldo2 ? 2 : 1, id);  // the second K part's sums
                if (j.ksplit > 2) mat(j.kout2, j.M, j.N, j.ldo1, 1, id);
                if (j.ksplit > 3) mat(j.kout3, j.M, j.N, j.ldo1, 1, id);
                break;
            case SK_EPI_GRU_GATES:
                mat(j.e0, j.M, H, j.lde0, 0, id); mat(j.o1, j.M, H, j.ldo1, 1, id); mat(j.o2, j.M, H, j.ldo2, 1, id);
                mat(j.out, j.M, H, j.ldo, 1, id);
                break;
            case SK_EPI_GRU_CAND:
                mat(j.e0, j.M, H, j.lde0, 0, id); mat(j.e1, j.M, H, j.lde1, 0, id); mat(j.mask, j.M, 1, 1, 0, id);
                mat(j.o1, j.M, H, j.ldo1, 1, id); mat(j.out, j.M, H, j.ldo, 1, id);
                break;
            case SK_EPI_BWD_RH:
                mat(j.e0, j.M, H, j.lde0, 0, id); mat(j.e1, j.M, H, j.lde1, 0, id); mat(j.out, j.M, H, j.ldo, 1, id);
                mat(j.o1, j.M, H, j.ldo1, 2, id);
                break;
            case SK_EPI_LSTM:
                mat(j.e1, j.M, H, j.lde1, 0, id); mat(j.o1, j.M, H, j.ldo1, 1, id); mat(j.o2, j.M, 4 * H, j.ldo2, 1, id);
                mat(j.out, j.M, H, j.ldo, 1, id);
                break;
            default: break;
        }
    }
    void att_fwd(const AttFwdArgs& g, int id) {
        jobs.push_back({launch, id, g.B, g.E, g.H, -1});
        mat(g.h1, g.B, g.H, g.ldh, 0, id); mat(g.kappa_prev, g.B, g.A, g.A, 0, id);
        mat(g.a_out, g.B, g.A, g.A, 1, id); mat(g.b_out, g.B, g.A, g.A, 1, id); mat(g.kappa_out, g.B, g.A, g.A, 1, id);
        mat(g.phi_out, g.B, g.U, g.U, 1, id); mat(g.w_out, g.B, g.E, g.ldw, 1, id); mat(g.sup_out, g.B, 2, 2, 1, id);
    }
    void att_bwd(const AttBwdArgs& g, int id) {
        jobs.push_back({launch, id, g.B, g.E, g.H, -2});
        mat(g.dw, g.B, g.E, g.lddw, g.dw2 ? 2 : 0, id); mat(g.dw2, g.B, g.E, g.lddw, 0, id);
        mat(g.dw3, g.B, g.E, g.lddw, 0, id); mat(g.dw4, g.B, g.E, g.lddw, 0, id);
        mat(g.dw5, g.B, g.E, g.lddw, 0, id); mat(g.dw6, g.B, g.E, g.lddw, 0, id);
        mat(g.a, g.B, g.A, g.A, 0, id); mat(g.b, g.B, g.A, g.A, 0, id); mat(g.kappa, g.B, g.A, g.A, 0, id);
        mat(g.kappa_prev, g.B, g.A, g.A, 0, id); mat(g.sup, g.B, 2, 2, 0, id);
        mat(g.dkappa, g.B, g.A, g.A, 2, id); mat(g.dp_out, g.B, 3 * g.A, 3 * g.A, 1, id); mat(g.dh1, g.B, g.H, g.lddh, 2, id);
    }
    void chain(const GruStateBwdChain& c, int B, int H, int id) {
        mat(c.dh, B, H, H, 0, id); mat(c.dh2, B, H, H, 0, id); mat(c.hprev, B, H, H, 0, id); mat(c.z, B, H, H, 0, id);
        for (int q = 0; q < 3; ++q) mat(c.dhx[q], B, H, H, 0, id);
        mat(c.c, B, H, H, 0, id); mat(c.mask, B, 1, 1, 0, id);
        mat(c.dC, B, H, H, 1, id); mat(c.dG, B, H, 2 * H, 1, id); mat(c.dhprev, B, H, H, 2, id);
    }
    void chain(const LstmStateBwdChain& c, int B, int H, int id) {
        mat(c.dh, B, H, H, 0, id); mat(c.dh2, B, H, H, 0, id); mat(c.dh3, B, H, H, 0, id); mat(c.dh4, B, H, H, 0, id);
        mat(c.dh5, B, H, H, 0, id); mat(c.dh6, B, H, H, 0, id);
        mat(c.gates, B, 4 * H, 4 * H, 0, id);
        mat(c.c_prev, B, H, H, 0, id); mat(c.c_new, B, H, H, 0, id);
        mat(c.dc, B, H, H, 2, id); mat(c.dP, B, 4 * H, 4 * H, 1, id);
    }
};
thread_local Tracer* g_tracer = nullptr;
enum { TRACE_JOB_ATT = 100, TRACE_JOB_CHAIN = 200 };

int launch_jobs(const SkJob* jobs, int n, hipStream_t s, int full_wgs = 0, int force_tile = 0, int force_wide = 0) {
    SkLaunch L;
    PL_TRY(sk_make_launch(L, jobs, n));
    L.force_wide = force_wide;
    if (g_tracer) {
        g_tracer->begin();
        for (int q = 0; q < n; ++q) {
            if (jobs[q].wait_flag) return PARROT_ERR_BADARG;  // (a flag needs its producers in the launch)
            g_tracer->sk_job(jobs[q], q);
        }
        return 0;
    }
    L.full_wgs = full_wgs;
    L.force_tile = force_tile;
    return sk_launch(L, s);
}

int traced_att_fwd_launch(const AttFwdArgs& att, hipStream_t s) {
    if (g_tracer) {
        g_tracer->begin();
        g_tracer->att_fwd(att, TRACE_JOB_ATT);
        return 0;
    }
    return att_fwd_launch(att, s);
}

// The attention forward step and n step-GEMM jobs in one heterogeneous launch (skinny.hip: ska_kernel).
int launch_jobs_att(const SkJob* jobs, int n, const AttFwdArgs& att, hipStream_t s, int full_wgs = 0) {
    if (n < 1) return traced_att_fwd_launch(att, s);
    SkLaunch L;
    PL_TRY(sk_make_launch(L, jobs, n));
    if (g_tracer) {
        g_tracer->begin();
        g_tracer->att_fwd(att, TRACE_JOB_ATT);
        for (int q = 0; q < n; ++q) g_tracer->sk_job(jobs[q], q);
        return 0;
    }
    L.full_wgs = full_wgs;
    return sk_launch_att(L, att, s);
}

// attention backward (or null) + the state backward of all chains in one launch; layer 0's chain is fused behind the
// attention in the same workgroups (one job as far as ordering goes)
template <class SA>
int traced_att_state_bwd_launch(const AttBwdArgs* g, const SA& sa, int l0_chain, hipStream_t s) {
    if (g_tracer) {
        g_tracer->begin();
        if (g) g_tracer->att_bwd(*g, TRACE_JOB_ATT);
        for (int q = 0; q < sa.nchain; ++q)
            g_tracer->chain(sa.chain[q], sa.B, sa.H, (g && q == l0_chain) ? (int)TRACE_JOB_ATT : TRACE_JOB_CHAIN + q);
        return 0;
    }
    return att_state_bwd_launch(g, sa, l0_chain, s);
}

// Attention backward (or null) + GRU state backward of all chains + step-GEMM jobs nothing in the launch feeds, in ONE
// heterogeneous launch (skinny.hip skb_kernel): the attention launch of the K-balanced backward tick (bwd8).
int traced_bwd_hetero_launch(const AttBwdArgs* g, const GruStateBwdArgs& sa, int l0_chain, const SkJob* jobs, int n,
                             hipStream_t s) {
    SkLaunch L;
    PL_TRY(sk_make_launch(L, jobs, n));
    if (g_tracer) {
        g_tracer->begin();
        if (g) g_tracer->att_bwd(*g, TRACE_JOB_ATT);
        for (int q = 0; q < sa.nchain; ++q)
            g_tracer->chain(sa.chain[q], sa.B, sa.H, (g && q == l0_chain) ? (int)TRACE_JOB_ATT : TRACE_JOB_CHAIN + q);
        for (int q = 0; q < n; ++q) g_tracer->sk_job(jobs[q], q);
        return 0;
    }
    L.full_wgs = 128;  // beside 64 + 64 row blocks: keep the 32 x 32 tiles (160 workgroups at cfg2), one workgroup per CU
    return sk_launch_bwd_hetero(L, g, sa, l0_chain, s);
}

// The fused backward tick of schedule 7 (skinny.hip wkb_kernel): attention backward (or null) + the LSTM state backward of
// all chains as row blocks at the head of ONE launch, the transposed products `jobs` behind them, each waiting (wait_all)
// on the flag of the chain that writes its dP operand.  PH_ERR_UNSUPPORTED: the wide kernel does not take these jobs.
int traced_bwd_fused_launch(const AttBwdArgs* g, const LstmStateBwdArgs& sa, int l0_chain, const SkJob* jobs, int n,
                            unsigned* const* flags, hipStream_t s) {
    SkLaunch L;
    PL_TRY(sk_make_launch(L, jobs, n));
    L.force_wide = 1;
    if (g_tracer) {
        g_tracer->begin();
        if (g) g_tracer->att_bwd(*g, TRACE_JOB_ATT);
        for (int q = 0; q < sa.nchain; ++q)
            g_tracer->chain(sa.chain[q], sa.B, sa.H, (g && q == l0_chain) ? (int)TRACE_JOB_ATT : TRACE_JOB_CHAIN + q);
        for (int q = 0; q < n; ++q) g_tracer->sk_job(jobs[q], q);
        return 0;
    }
    return sk_launch_bwd_fused(L, g, sa, l0_chain, flags, s);
}


// ----------------------------------------------------------------------------- GRU scan
struct GruSeqPlan : PlanBase {
    ParrotGruSeqDesc d;

    // Narrow layers (H <= 256: the encoder) run the whole sequence as ONE launch per direction on the row-owning
    // kernels of rowgru.hip (PARROT_GRU_ROWWISE=0: the per-step launches below).  The plan owns the fragment-major
    // weight copies and refreshes them at the head of every forward scan (the weights change between steps).
    bool rowwise = false;
    float* tiled = nullptr;  // per chain: Wg_f, Wc_f, Wg_r, Wc_r
    ~GruSeqPlan() override {
        if (tiled) (void)hipFree(tiled);
    }
    int setup_rowwise() {
        const char* e = getenv("PARROT_GRU_ROWWISE");
        rowwise = rowgru_supported(d.T, d.B, d.H, d.nchain) && !(e && atoi(e) == 0);
        for (int ch = 0; ch < d.nchain && rowwise; ++ch)  // (the tiling kernel wants 16-byte aligned matrices)
            if (!d.Wg[ch] || !d.Wc[ch] || ((uintptr_t)d.Wg[ch] & 15) || ((uintptr_t)d.Wc[ch] & 15)) rowwise = false;
        if (!rowwise) return 0;
        const size_t per = (size_t)6 * d.H * d.H;  // 2 x (H x 2H + H x H) floats
        if (hipMalloc(&tiled, sizeof(float) * per * d.nchain) != hipSuccess) {
            tiled = nullptr;
            rowwise = false;
        }
        return 0;
    }
    RowGruArgs row_args() const {
        RowGruArgs g;
        memset(&g, 0, sizeof(g));
        const size_t HH = (size_t)d.H * d.H;
        for (int ch = 0; ch < d.nchain; ++ch) {
            RowGruChain& c = g.chain[ch];
            float* base = tiled + (size_t)ch * 6 * HH;
            c.Wg_f = base; c.Wc_f = base + 2 * HH; c.Wg_r = base + 3 * HH; c.Wc_r = base + 5 * HH;
            c.inputs = d.inputs[ch]; c.gate_inputs = d.gate_inputs[ch];
            c.h = d.h[ch]; c.z = d.z[ch]; c.r = d.r[ch]; c.rh = d.rh[ch]; c.c = d.c[ch];
            c.dh = d.dh[ch]; c.dG = d.dG[ch]; c.dC = d.dC[ch];
            c.reverse = d.reverse[ch];
        }
        g.mask = d.mask; g.T = d.T; g.B = d.B; g.H = d.H; g.nchain = d.nchain;
        return g;
    }
    int fwd_rowwise(hipStream_t st) {
        const RowGruArgs g = row_args();
        for (int ch = 0; ch < d.nchain; ++ch) {
            const RowGruChain& c = g.chain[ch];
            PL_TRY(sk_tile_weights_launch(d.Wg[ch], d.H, 2 * d.H, 2 * d.H, const_cast<float*>(c.Wg_f), 0, 0, st));
            PL_TRY(sk_tile_weights_launch(d.Wc[ch], d.H, d.H, d.H, const_cast<float*>(c.Wc_f), 0, 0, st));
            PL_TRY(sk_tile_weights_launch(d.Wg[ch], d.H, 2 * d.H, 2 * d.H, const_cast<float*>(c.Wg_r), 1, 0, st));
            PL_TRY(sk_tile_weights_launch(d.Wc[ch], d.H, d.H, d.H, const_cast<float*>(c.Wc_r), 1, 0, st));
        }
        return rowgru_fwd_launch(g, st);
    }

    int enqueue(int which, hipStream_t s) override {
        if (rowwise) return which == 0 ? fwd_rowwise(s) : rowgru_bwd_launch(row_args(), s);
        return which == 0 ? fwd(s) : bwd(s);
    }

    int fwd(hipStream_t st) {
        const size_t BH = (size_t)d.B * d.H;
        for (int s = 0; s < d.T; ++s) {
            SkJob jobs[4];
            for (int ch = 0; ch < d.nchain; ++ch) {
                const int t = d.reverse[ch] ? d.T - 1 - s : s;
                SkJob& j = jobs[ch];
                sk_job_init(j);
                j.nseg = 1;
                j.seg[0] = sk_seg(d.h[ch] + s * BH, d.H, d.Wg[ch], 2 * d.H, d.H, 0);
                j.M = d.B; j.N = 2 * d.H; j.H = d.H; j.epi = SK_EPI_GRU_GATES;
                j.add = d.gate_inputs[ch] ? d.gate_inputs[ch] + t * 2 * BH : nullptr; j.ld_add = 2 * d.H;
                j.e0 = d.h[ch] + s * BH; j.lde0 = d.H;
                j.o1 = d.z[ch] + t * BH; j.ldo1 = d.H;
                j.o2 = d.r[ch] + t * BH; j.ldo2 = d.H;
                j.out = d.rh[ch] + t * BH; j.ldo = d.H;
            }
            PL_TRY(launch_jobs(jobs, d.nchain, st));
            for (int ch = 0; ch < d.nchain; ++ch) {
                const int t = d.reverse[ch] ? d.T - 1 - s : s;
                SkJob& j = jobs[ch];
                sk_job_init(j);
                j.nseg = 1;
                j.seg[0] = sk_seg(d.rh[ch] + t * BH, d.H, d.Wc[ch], d.H, d.H, 0);
                j.M = d.B; j.N = d.H; j.H = d.H; j.epi = SK_EPI_GRU_CAND;
                j.add = d.inputs[ch] ? d.inputs[ch] + t * BH : nullptr; j.ld_add = d.H;
                j.e0 = d.h[ch] + s * BH; j.lde0 = d.H;
                j.e1 = d.z[ch] + t * BH; j.lde1 = d.H;
                j.o1 = d.c[ch] + t * BH; j.ldo1 = d.H;
                j.out = d.h[ch] + (s + 1) * BH; j.ldo = d.H;
                j.mask = d.mask ? d.mask + (size_t)t * d.B : nullptr;
            }
            PL_TRY(launch_jobs(jobs, d.nchain, st));
        }
        return 0;
    }

    int bwd(hipStream_t st) {
        const size_t BH = (size_t)d.B * d.H;
        for (int s = d.T - 1; s >= 0; --s) {
            GruStateBwdArgs ga;
            ga.nchain = d.nchain; ga.B = d.B; ga.H = d.H;
            SkJob jx[4], jy[4];
            for (int ch = 0; ch < d.nchain; ++ch) {
                const int t = d.reverse[ch] ? d.T - 1 - s : s;
                GruStateBwdChain& c = ga.chain[ch];
                c.dh = d.dh[ch] + (s + 1) * BH;
                c.dh2 = nullptr;
                c.hprev = d.h[ch] + s * BH;
                c.z = d.z[ch] + t * BH;
                c.c = d.c[ch] + t * BH;
                c.mask = d.mask ? d.mask + (size_t)t * d.B : nullptr;
                c.dC = d.dC[ch] + t * BH;
                c.dG = d.dG[ch] + t * 2 * BH;
                c.dhprev = d.dh[ch] + s * BH;

                SkJob& x = jx[ch];
                sk_job_init(x);
                x.nseg = 1;
                x.seg[0] = sk_seg(d.dC[ch] + t * BH, d.H, d.Wc[ch], d.H, d.H, 1);
                x.M = d.B; x.N = d.H; x.H = d.H; x.epi = SK_EPI_BWD_RH;
                x.e0 = d.h[ch] + s * BH; x.lde0 = d.H;
                x.e1 = d.r[ch] + t * BH; x.lde1 = d.H;
                x.out = d.dG[ch] + t * 2 * BH + d.H; x.ldo = 2 * d.H;
                x.o1 = d.dh[ch] + s * BH; x.ldo1 = d.H;

                SkJob& y = jy[ch];
                sk_job_init(y);
                y.nseg = 1;
                y.seg[0] = sk_seg(d.dG[ch] + t * 2 * BH, 2 * d.H, d.Wg[ch], 2 * d.H, 2 * d.H, 1);
                y.M = d.B; y.N = d.H; y.H = d.H; y.epi = SK_EPI_LINEAR; y.accumulate = 1;
                y.out = d.dh[ch] + s * BH; y.ldo = d.H;
            }
            PL_TRY(gru_state_bwd_launch(ga, st));
            PL_TRY(launch_jobs(jx, d.nchain, st));
            PL_TRY(launch_jobs(jy, d.nchain, st));
        }
        return 0;
    }
};

// ----------------------------------------------------------------------------- LSTM scan
struct LstmSeqPlan : PlanBase {
    ParrotLstmSeqDesc d;
    int enqueue(int which, hipStream_t s) override { return which == 0 ? fwd(s) : bwd(s); }

    int fwd(hipStream_t st) {
        const size_t BH = (size_t)d.B * d.H;
        for (int t = 0; t < d.T; ++t) {
            SkJob j;
            sk_job_init(j);
            j.nseg = 1;
            j.seg[0] = sk_seg(d.s + t * BH, d.H, d.W, 4 * d.H, d.H, 0);
            j.M = d.B; j.N = 4 * d.H; j.H = d.H; j.epi = SK_EPI_LSTM;
            j.add = d.pre_in + (size_t)t * 4 * BH; j.ld_add = 4 * d.H;
            j.e1 = d.c + t * BH; j.lde1 = d.H;
            j.o1 = d.c + (t + 1) * BH; j.ldo1 = d.H;
            j.o2 = d.gates + (size_t)t * 4 * BH; j.ldo2 = 4 * d.H;
            j.out = d.s + (t + 1) * BH; j.ldo = d.H;
            PL_TRY(launch_jobs(&j, 1, st));
        }
        return 0;
    }

    int bwd(hipStream_t st) {
        const size_t BH = (size_t)d.B * d.H;
        for (int t = d.T - 1; t >= 0; --t) {
            PL_TRY(lstm_state_bwd_launch(d.dS + (t + 1) * BH, nullptr, d.dc, d.gates + (size_t)t * 4 * BH, d.c + t * BH,
                                         d.c + (t + 1) * BH, d.dP + (size_t)t * 4 * BH, d.B, d.H, st));
            SkJob y;
            sk_job_init(y);
            y.nseg = 1;
            y.seg[0] = sk_seg(d.dP + (size_t)t * 4 * BH, 4 * d.H, d.W, 4 * d.H, 4 * d.H, 1);
            y.M = d.B; y.N = d.H; y.H = d.H; y.epi = SK_EPI_LINEAR; y.accumulate = 1;
            y.out = d.dS + t * BH; y.ldo = d.H;
            PL_TRY(launch_jobs(&y, 1, st));
        }
        return 0;
    }
};

// ----------------------------------------------------------------------------- persistent machine: unit placement
struct PmReq { PmUnit u; int slot, krows, crit; };

// Greedy placement of the units of one tick on the workgroups: per slot, critical and long units first; a unit goes
// to the workgroup whose slot would end earliest (measured unit cost: ~3.5 us fixed + ~3.5 us per 1024 K-rows from
// LDS, ~2x the K term when streamed; an attention row ~9 us), ties broken towards the least loaded workgroup.  A
// unit's weight slab becomes LDS-resident when the workgroup still has room for it.
bool pm_place(std::vector<PmReq>& reqs, int n_slots, int maxu, int nwg, std::vector<PmUnit>& table) {
    table.assign((size_t)n_slots * nwg * maxu, PmUnit());
    memset(table.data(), 0, table.size() * sizeof(PmUnit));
    std::vector<int> lds_used(nwg, 0);
    std::vector<int> cnt((size_t)n_slots * nwg, 0);
    std::vector<double> busy((size_t)n_slots * nwg, 0.0), load(nwg, 0.0);
    std::vector<int> order(reqs.size());
    for (size_t i = 0; i < reqs.size(); ++i) order[i] = (int)i;
    // critical units first, then the longest K first across ALL slots: LDS residency is handed out in this order, and
    // it pays most where the weight slab is largest
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        if (reqs[a].crit != reqs[b].crit) return reqs[a].crit > reqs[b].crit;
        if (reqs[a].krows != reqs[b].krows) return reqs[a].krows > reqs[b].krows;
        return reqs[a].slot < reqs[b].slot;
    });
    for (int idx : order) {
        PmReq& q = reqs[idx];
        const int need = q.krows * 16;
        const bool is_att = q.u.kind == PM_ATT;
        int best = -1;
        double best_key = 0;
        for (int w = 0; w < nwg; ++w) {
            if (cnt[(size_t)q.slot * nwg + w] >= maxu) continue;
            const bool fits = lds_used[w] + need <= PM_LDS_W;
            const double cost = is_att ? 9.0 : 3.5 + (fits ? 3.5 : 7.0) * q.krows / 1024.0;
            const double key = (busy[(size_t)q.slot * nwg + w] + cost) * 1e3 + load[w];
            if (best < 0 || key < best_key) { best = w; best_key = key; }
        }
        if (best < 0) return false;  // more units than places
        const bool fits = lds_used[best] + need <= PM_LDS_W;
        if (need > 0 && fits) {
            q.u.w_lds = lds_used[best];
            lds_used[best] += need;
        }
        const double cost = is_att ? 9.0 : 3.5 + (fits ? 3.5 : 7.0) * q.krows / 1024.0;
        busy[(size_t)q.slot * nwg + best] += cost;
        load[best] += cost;
        int& c = cnt[(size_t)q.slot * nwg + best];
        table[((size_t)q.slot * nwg + best) * maxu + c] = q.u;
        ++c;
    }
    return true;
}

// ----------------------------------------------------------------------------- decoder (training)
struct DecoderPlan : PlanBase {
    ParrotDecoderDesc d;
    int esplit = 1;

    // Schedules (PARROT_SCHEDULE; 2 and 3 need seq buffers for the upper layers).  cfg2 (T=800, B=64, H=1024, L=2)
    // on MI355X, fwd/bwd ms at the time each was measured against schedule 0:
    //   0 merged wavefront launches, one stream (default)                         46 / 60
    //   1 stream per layer with per-step events in one graph (experiment)         69 / 105  (0: 56 / 78)
    //   2 chunked layer pipeline, one graph per (layer, chunk) piece, 2 streams   57 / 94   (0: 56 / 78)
    //   3 chunk-skewed merged wavefront with hoisted projections, one stream      49 / 70   (0: 46 / 60)
    // The per-step kernels are latency-bound: a merged launch costs ~9 us + ~7.7 us per 1024 of its longest K, so
    // moving K from the step kernels to batched GEMMs (2, 3) or splitting layers over streams (1, 2) buys less than
    // the extra kernels / GEMMs cost.  2 and 3 are what layer_norm needs (the projections that must be normalised
    // are the hoisted ones); layer_norm with L >= 2 therefore runs on 3.
    int schedule = 0, chunk = 50;
    bool try_persist = false;

    int full_wgs = 0;  // (launch_jobs: 0 = the step kernel's own tile heuristic)

    // Records what the plan's launches of direction `which` read and write (see Tracer); schedules 0, 5, 6 and 7 only.
    int trace(int which, std::vector<TraceRec>& out, std::vector<TraceJob>* jobs_out = nullptr) {
        if (persist_ok || (schedule != 0 && schedule < 5) || d.layer_norm) return PARROT_ERR_UNSUPPORTED;
        Tracer tr;
        g_tracer = &tr;
        const int rc = enqueue(which, nullptr);
        g_tracer = nullptr;
        out.swap(tr.recs);
        if (jobs_out) jobs_out->swap(tr.jobs);
        return rc;
    }

    int enqueue(int which, hipStream_t s) override {
        BgPrecisionScope precision(d.bf16);  // the hoisted projections follow the plan's operand mode
        if (which == 0 && persist_ok) return pm_launch(pm_prog, s);  // schedule 4: the persistent phase machine
        if (schedule == 3) return which == 0 ? fwd_skew(s) : bwd_skew(s);
        if (which == 1 && bwd_hetero && (schedule == 0 || schedule == 5)) return bwd8(s);
        if (schedule == 5) return which == 0 ? fwd5(s) : bwd(s);
        if (schedule == 7) return which == 0 ? fwd7(s) : bwd(s);
        return which == 0 ? fwd(s) : bwd(s);
    }

    void choose_schedule() {
        const char* e = getenv("PARROT_SCHEDULE");
        int want = e ? atoi(e) : -1;
        bool pipe_ok = d.L >= 2;
        for (int l = 1; l < d.L; ++l)
            if (!d.seq_g[l] || (d.cell == 0 && !d.seq_c[l])) pipe_ok = false;
        // 4 = persistent forward scan (resolved in parrot_decoder_create; everything it does not cover -- the backward
        // scan, LSTM layers, layer_norm, B > 64 -- runs on the launch schedules chosen below)
        const bool want_persist = want == 4;  // opt-in: measured at cfg2 it only matches the launch schedules
        if (want_persist) want = -1;
        if (want < 0) {
            // default: the balanced wavefront (5) where it pays -- GRU layers, L >= 2, no layer_norm; measured at cfg2:
            // forward scan 29.2 -> 25.2 ms
            // GRU stacks, f32 or bf16 operands (cfg2: 82.7 -> 74.6 ms f32, 59.8 -> 54.0 ms bf16; 3 layers 122.5 -> 115.0).
            // LSTM stacks stay on schedule 0 (5 covers them, opt-in): their tick is ONE fused launch of > 1000 workgroups,
            // bound by total work rather than by its longest K, and cutting it in two only adds fixed cost -- cfg4 bf16
            // 118.9 vs 127.5 ms (the wide kernel has ~10 us of fixed cost per launch), cfg4 f32 256.5 vs 265.4 ms.
            want = (pipe_ok && d.cell == 0 && !d.layer_norm) ? 5 : 0;
            // LSTM stacks with bf16 operands (BASELINE configs[3]): the attention inside the tick's one launch (7), where
            // the wide step kernel takes the launch (checked in parrot_decoder_create)
            if (d.cell == 1 && d.bf16 && !d.layer_norm) want = 7;
        }
        if (d.layer_norm && d.L >= 2 && want < 2) want = 3;  // the in-scan normalisations need the hoisted projections
        if (want != 0 && want != 3 && want != 5 && want != 7) want = 0;      // (schedules 1, 2 and 6 of rounds 1-3 were removed:
                                                                             //  measured losers, numbers in DESIGN.md 3.2)
        if (want >= 2 && want != 7 && !pipe_ok) want = 0;
        if (want == 7 && d.cell != 1) want = pipe_ok ? 5 : 0;                // one launch per tick: LSTM layers
        if (want >= 5 && d.layer_norm) want = 0;
        schedule = want;
        try_persist = want_persist && d.cell == 0 && !d.layer_norm && !d.bf16;
        const char* c = getenv("PARROT_CHUNK");  // (tests: several chunks and a ragged last one on short windows)
        if (c && atoi(c) > 0) chunk = atoi(c);
    }


    // ---- schedule 4: persistent phase machine for the forward scan (persist.h) ---------------------------------
    // Units per step t: layer 0: G0 = gates over [h0[t]; w[t]], C0 = candidate over [r*h0; w[t]], ATT (one per batch
    // row); layer l >= 1: IG_l / IC_l = the projections of [w[t+1]; h_0[t+1] .. h_{l-1}[t+1]] into the layer's gates /
    // candidate (written to pre-activation buffers), G_l / C_l = the recurrent products over h_l[t] / r*h_l.
    // Tick q runs slot 0: G0(q), G_l(q - 2l);  slot 1: C0(q), C_l(q - 2l), IC_l(q - 2l + 1);  slot 2: ATT(q),
    // IG_l(q - 2l + 1).  Every unit's inputs were published at least one barrier earlier (see the lag arithmetic in
    // DESIGN.md).  Units are spread over the workgroups greedily; a unit's weight slab stays in the workgroup's LDS
    // for the whole window when it fits (critical recurrent units first), otherwise it is streamed.
    bool persist_ok = false;
    enum { PERSIST_MAXPIECES = 4, TR_SLOTS = 3, TR_MAXU = 3 };
    PmProgram pm_prog;
    static long long persist_floats(const ParrotDecoderDesc& d, int nwg) {
        const int MB = d.B <= 16 ? 1 : (d.B <= 32 ? 2 : 4);
        const long long rows = (long long)MB * 16;
        long long n = PM_SYNC_WORDS + PM_DBG_WORDS;
        n += ((long long)TR_SLOTS * nwg * TR_MAXU * sizeof(PmUnit) + 3) / 4 + 64;
        n += 2 * (long long)(d.T + 1) * rows * (d.H + d.E);                       // XG0, XC0
        for (int l = 1; l < d.L; ++l)
            n += 2 * (long long)(d.T + 1) * rows * d.H + (long long)d.T * rows * (d.E + l * d.H);  // XG_l, XC_l, XI_l
        for (int l = 1; l < d.L; ++l) n += (long long)d.T * d.B * 3 * d.H * PERSIST_MAXPIECES / (l == 1 ? 2 : 1);  // partials
        return n + 1024;
    }
    static bool persist_eligible(const ParrotDecoderDesc& d) {
        if (d.cell != 0 || d.layer_norm || d.bf16 || d.B > 64 || (d.H % 16) || (d.E % 16) || d.U > PM_ATT_MAXU ||
            d.A > PM_ATT_MAXA || d.T < 1)
            return false;
        for (int l = 0; l < d.L; ++l)
            if (!d.Wg_f[l] || !d.Wc_f[l]) return false;
        return pm_max_workgroups() >= 64;
    }

    int build_persist() {
        persist_ok = false;
        if (!persist_eligible(d) || !tiled || !d.persist_ws) return 0;
        const int nwg = pm_max_workgroups();
        if (d.persist_ws_floats < persist_floats(d, nwg)) return 0;
        const int H = d.H, E = d.E, B = d.B, L = d.L, T = d.T;
        const int MB = B <= 16 ? 1 : (B <= 32 ? 2 : 4);
        const long long rows = (long long)MB * 16;
        const long long BH = (long long)B * H;
        // carve the workspace (16-byte aligned pieces)
        float* ws = d.persist_ws;
        auto take = [&](long long n) { float* p = ws; ws += (n + 3) / 4 * 4; return p; };
        unsigned* sync = reinterpret_cast<unsigned*>(take(PM_SYNC_WORDS + PM_DBG_WORDS));
        const size_t unit_bytes = (size_t)TR_SLOTS * nwg * TR_MAXU * sizeof(PmUnit);
        PmUnit* units_dev = reinterpret_cast<PmUnit*>(take((long long)(unit_bytes + 3) / 4 + 16));
        // activation slabs (fragment-major, one per consumer kind and step):
        //   XG[l][t] = A operand of G_l(t):  l = 0: [h_0[t] ; w[t]],  l >= 1: [h_l[t]]
        //   XC[l][t] = A operand of C_l(t):  l = 0: [r*h_0 ; w[t]],   l >= 1: [r*h_l]
        //   XI[l][t] = A operand of IG_l(t) / IC_l(t), l >= 1: [w[t+1] ; h_0[t+1] ; .. ; h_{l-1}[t+1]]
        float* XG[PARROT_MAX_LAYERS];
        float* XC[PARROT_MAX_LAYERS];
        float* XI[PARROT_MAX_LAYERS] = {nullptr, nullptr, nullptr};
        long long kx[PARROT_MAX_LAYERS], ki[PARROT_MAX_LAYERS] = {0, 0, 0};
        float* fm_base = ws;
        for (int l = 0; l < L; ++l) {
            kx[l] = l == 0 ? H + E : H;
            XG[l] = take((long long)(T + 1) * rows * kx[l]);
            XC[l] = take((long long)(T + 1) * rows * kx[l]);
            if (l >= 1) {
                ki[l] = E + (long long)l * H;
                XI[l] = take((long long)T * rows * ki[l]);
            }
        }
        const long long fm_bytes = (long long)(ws - fm_base) * 4;
        if (fm_bytes >= 0xfff00000ll) return 0;  // one 32-bit buffer resource addresses the slab region
        // Input projections of the layers l >= 1 are cut into pieces (K ranges of the XI slab) that write separate
        // partial pre-activation buffers; the consuming recurrent unit adds them.  Pieces: the w rows, then per lower
        // layer its H rows -- whole when the slab can stay LDS-resident, otherwise (streamed) in two halves so that
        // no streamed unit is longer than a resident recurrent one.
        float* pre[PARROT_MAX_LAYERS][2][PERSIST_MAXPIECES];
        memset(pre, 0, sizeof(pre));
        float* const pre_begin = ws;
        for (int l = 1; l < L; ++l)
            for (int g = 0; g < 2; ++g)
                for (int q = 0; q < (l == 1 ? PERSIST_MAXPIECES / 2 : PERSIST_MAXPIECES); ++q)
                    pre[l][g][q] = take((long long)T * B * (g == 0 ? 2 * H : H));
        float* const pre_end = ws;
        auto boff = [&](const float* p) { return (unsigned)((p - fm_base) * 4); };
        auto mkdst = [&](float* slab, long long step0, long long kslab, int chunk) {
            PmDst q;
            q.off = boff(slab + step0 * rows * kslab);
            q.st = (unsigned)(rows * kslab * 4);
            q.nch = (int)(kslab / 16);
            q.chunk = chunk;
            return q;
        };

        // When the chip's LDS (nwg x PM_LDS_W) cannot hold all weights, some input-projection pieces are streamed:
        // cut the H-row pieces in halves then (see above).
        long long all_rows = 0;
        for (int l = 0; l < L; ++l) all_rows += (long long)krows(l) * 3 * H / 16;
        // (measured at cfg2: halves do not pay -- every unit carries ~3.5 us of fixed latency -- so off unless asked for)
        const bool stream_split = getenv("PARROT_PM_SPLIT") && atoi(getenv("PARROT_PM_SPLIT")) &&
                                  all_rows * 16 > (long long)nwg * PM_LDS_W;
        std::vector<PmReq> reqs;
        typedef PmReq Req;
        auto rm = [](float* p, long long st, int ld) { PmRM r; r.p = p; r.st = st; r.ld = ld; r.pad = 0; return r; };
        auto seq_on = [&](int l, const float* p) { return p && ((d.seq_init >> l) & 1); };
        for (int l = 0; l < L; ++l) {
            const int Kl = krows(l), nchK = Kl / 16;
            for (int g = 0; g < 2; ++g) {                      // g = 0: gates (2H wide), 1: candidate (H wide)
                const int wd = g == 0 ? 2 * H : H;
                const float* Wf = g == 0 ? d.Wg_f[l] : d.Wc_f[l];
                const float* bias = g == 0 ? d.bg[l] : d.bc[l];
                float* sq = g == 0 ? d.seq_g[l] : d.seq_c[l];
                for (int ct = 0; ct < wd / 16; ++ct) {
                    // recurrent unit (layer 0: the whole product)
                    Req q;
                    memset(&q, 0, sizeof(q));
                    PmUnit& u = q.u;
                    u.kind = PM_GEMM; u.lag = 2 * l; u.M = B; u.w_lds = -1;
                    float* slab = g == 0 ? XG[l] : XC[l];
                    u.a_off = boff(slab); u.a_st = (unsigned)(rows * kx[l] * 4); u.K = (int)kx[l];
                    u.a_nch = (int)(kx[l] / 16); u.a_c0 = 0;
                    u.W = Wf + (size_t)ct * nchK * 256;
                    if (l == 0) {
                        u.bias = bias ? bias + 16 * ct : nullptr;
                        if (seq_on(l, sq)) u.add[0] = rm(sq + 16 * ct, (long long)B * wd, wd);
                    }
                    if (g == 0) {
                        u.epi = PM_EPI_GATES;
                        u.rtile = 16 * ct >= H;
                        if (!u.rtile) {
                            u.o1 = rm(d.z[l] + 16 * ct, BH, H);
                        } else {
                            const int j0 = 16 * ct - H;
                            u.o2 = rm(d.r[l] + j0, BH, H);
                            u.e0 = rm(d.h[l] + j0, BH, H);
                            u.out = rm(d.rh[l] + j0, BH, H);
                            u.dst[u.ndst++] = mkdst(XC[l], 0, kx[l], j0 / 16);
                        }
                    } else {
                        u.epi = PM_EPI_CAND;
                        u.e0 = rm(d.h[l] + 16 * ct, BH, H);
                        u.e1 = rm(d.z[l] + 16 * ct, BH, H);
                        u.o1 = rm(d.c[l] + 16 * ct, BH, H);
                        u.out = rm(d.h[l] + BH + 16 * ct, BH, H);
                        u.dst[u.ndst++] = mkdst(XG[l], 1, kx[l], ct);               // h_l[t+1] for G_l(t+1)
                        for (int m2 = l + 1; m2 < L; ++m2)                            // ... and for the layers above
                            u.dst[u.ndst++] = mkdst(XI[m2], 0, ki[m2], E / 16 + l * (H / 16) + ct);
                    }
                    q.slot = g; q.crit = 1;
                    q.krows = (int)kx[l];
                    if (l == 0) { reqs.push_back(q); continue; }
                    // input projection of the layer, one tick ahead of its consumer, in pieces
                    struct Piece { int c0, K; };
                    std::vector<Piece> pieces;
                    pieces.push_back({0, E});
                    for (int j = 0; j < l; ++j) {
                        const int cj = E / 16 + j * (H / 16);
                        const bool split = stream_split && l >= 2 && (H % 32 == 0) &&
                                           (int)pieces.size() + 2 <= PERSIST_MAXPIECES - (l - 1 - j);
                        if (split) { pieces.push_back({cj, H / 2}); pieces.push_back({cj + H / 32, H / 2}); }
                        else pieces.push_back({cj, H});
                    }
                    int na = 0;
                    for (size_t pi = 0; pi < pieces.size(); ++pi) {
                        Req qi;
                        memset(&qi, 0, sizeof(qi));
                        PmUnit& v = qi.u;
                        v.kind = PM_GEMM; v.lag = 2 * l - 1; v.M = B; v.w_lds = -1;
                        v.a_off = boff(XI[l]); v.a_st = (unsigned)(rows * ki[l] * 4);
                        v.a_nch = (int)(ki[l] / 16); v.a_c0 = pieces[pi].c0; v.K = pieces[pi].K;
                        v.W = Wf + ((size_t)ct * nchK + H / 16 + pieces[pi].c0) * 256;
                        if (pi == 0) {
                            v.bias = bias ? bias + 16 * ct : nullptr;
                            if (seq_on(l, sq)) v.add[0] = rm(sq + 16 * ct, (long long)B * wd, wd);
                        }
                        v.epi = PM_EPI_LINEAR;
                        v.out = rm(pre[l][g][pi] + 16 * ct, (long long)B * wd, wd);
                        u.add[na++] = rm(pre[l][g][pi] + 16 * ct, (long long)B * wd, wd);
                        qi.slot = g == 0 ? 2 : 1; qi.crit = 0;
                        qi.krows = pieces[pi].K;
                        reqs.push_back(qi);
                    }
                    reqs.push_back(q);
                }
            }
        }
        for (int b = 0; b < B; ++b) {
            Req q;
            memset(&q, 0, sizeof(q));
            q.u.kind = PM_ATT; q.u.lag = 0; q.u.row = b; q.u.w_lds = -1;
            q.slot = 2; q.crit = 1; q.krows = 0;
            reqs.push_back(q);
        }
        std::vector<PmUnit> table;
        if (!pm_place(reqs, TR_SLOTS, TR_MAXU, nwg, table)) return 0;
        if (hipMemcpy(units_dev, table.data(), unit_bytes, hipMemcpyHostToDevice) != hipSuccess) return 0;

        PmProgram& P = pm_prog;
        memset(&P, 0, sizeof(P));
        P.T = T; P.n_ticks = T + 2 * (L - 1); P.nwg = nwg; P.MB = MB; P.M = B; P.n_slots = TR_SLOTS; P.maxu = TR_MAXU;
        P.units = units_dev; P.sync = sync; P.fm_base = fm_base;
        PmAtt& a = P.att;
        a.h1 = rm(d.h[0], BH, H);
        a.WattT = d.WattT; a.batt = d.batt; a.ctx = d.ctx;
        a.kappa = d.kappa; a.a = d.a; a.b = d.b; a.phi = d.phi; a.w = d.w;
        a.sup = d.att_sup;
        a.wdst[a.nwdst++] = mkdst(XG[0], 1, kx[0], H / 16);   // w[t+1] for G_0(t+1) and C_0(t+1) ...
        a.wdst[a.nwdst++] = mkdst(XC[0], 1, kx[0], H / 16);
        for (int l = 1; l < L; ++l) a.wdst[a.nwdst++] = mkdst(XI[l], 0, ki[l], 0);  // ... and for the layers above
        a.B = B; a.H = H; a.A = d.A; a.U = d.U; a.E = E; a.att_type = d.att_type;
        {
            const char* e = getenv("PARROT_ATT_DENSE");
            a.dense = e ? atoi(e) : 0;
        }
        a.eps = d.eps; a.alignment = d.alignment; a.sharpening = d.sharpening; a.timing = d.timing;
        int ni = 0;
        auto add_init = [&](const float* src, int ld, int K, float* slab, long long kslab, int chunk) {
            PmInit& in = P.init[ni++];
            in.src = src; in.ld = ld; in.K = K; in.dst_off = boff(slab); in.nch = (int)(kslab / 16); in.chunk = chunk;
            in.pad = 0;
        };
        for (int l = 0; l < L; ++l) add_init(d.h[l], H, H, XG[l], kx[l], 0);  // states entering the window (slot 0)
        add_init(d.w, E, E, XG[0], kx[0], H / 16);
        add_init(d.w, E, E, XC[0], kx[0], H / 16);
        P.ninit = ni;
        // dataflow mode: everything a unit reads from another workgroup starts EMPTY (slot 0 of the histories is the
        // caller's: the states entering the window)
        {
            const char* e = getenv("PARROT_PM_DATAFLOW");
            P.dataflow = e ? atoi(e) : 0;
        }
        auto add_fill = [&](void* q, long long nfloats) {
            if (nfloats > 0) { P.fill[P.nfill].p = q; P.fill[P.nfill].bytes = nfloats * 4; ++P.nfill; }
        };
        add_fill(fm_base, fm_bytes / 4);
        add_fill(pre_begin, (long long)(pre_end - pre_begin));
        for (int l = 0; l < L; ++l) {
            add_fill(d.h[l] + BH, (long long)T * BH);
            add_fill(d.z[l], (long long)T * BH);
        }
        persist_ok = true;
        return 0;
    }
    int persist_status() const { return persist_ok ? pm_status(pm_prog) : 0; }

    // ---- weight operands: plain packed matrices, or their fragment-major copies when the caller gave them
    bool tiled = false;
    int krows(int l) const { return d.H + d.E + l * d.H; }
    // forward product x . W[r0 : r0+K, :] of layer l's matrix g (0: Wg, 1: Wc), width ldw
    SkSeg fseg(const float* A, int lda, int l, int g, int r0, int K, int ldw) const {
        if (tiled) {
            const float* Wt = g == 0 ? d.Wg_f[l] : d.Wc_f[l];
            if (d.bf16) return sk_seg(A, lda, Wt + (size_t)(r0 >> 5) * 256, (krows(l) >> 5) * 256, K, 3);
            return sk_seg(A, lda, Wt + (size_t)(r0 >> 4) * 256, (krows(l) >> 4) * 256, K, 2);
        }
        const float* W = g == 0 ? d.Wg[l] : d.Wc[l];
        return sk_seg(A, lda, W + (size_t)r0 * ldw, ldw, K, 0);
    }
    // backward product dP . W[r0 : r0+N, :]^T (K = ldw = width of the matrix)
    SkSeg rseg(const float* A, int l, int g, int r0, int ldw) const {
        if (tiled) {
            const float* Wt = g == 0 ? d.Wg_r[l] : d.Wc_r[l];
            if (d.bf16) return sk_seg(A, ldw, Wt + (size_t)(r0 >> 4) * (ldw >> 5) * 256, (ldw >> 5) * 256, ldw, 3);
            return sk_seg(A, ldw, Wt + (size_t)(r0 >> 4) * (ldw >> 4) * 256, (ldw >> 4) * 256, ldw, 2);
        }
        const float* W = g == 0 ? d.Wg[l] : d.Wc[l];
        return sk_seg(A, ldw, W + (size_t)r0 * ldw, ldw, ldw, 1);
    }

    // Adds the K-segments [h_l ; w ; h_0..h_{l-1}] against layer l's matrix g (row-major [K_l, ldw]).
    void layer_segs(SkJob& j, int l, int t, const float* first, int g, int ldw) const {
        const size_t BH = (size_t)d.B * d.H, BE = (size_t)d.B * d.E;
        int n = 0;
        j.seg[n++] = fseg(first, d.H, l, g, 0, d.H, ldw);
        const float* wsrc = d.w + (size_t)(l == 0 ? t : t + 1) * BE;
        j.seg[n++] = fseg(wsrc, d.E, l, g, d.H, d.E, ldw);
        for (int q = 0; q < l; ++q)
            j.seg[n++] = fseg(d.h[q] + (size_t)(t + 1) * BH, d.H, l, g, d.H + d.E + q * d.H, d.H, ldw);
        j.nseg = n;
    }

    // The additive-input buffer of layer l is live when the caller filled it (seq_init bit) or when the
    // pipeline schedule batches the lower layers' projections into it.
    bool has_seq(int l, const float* p) const { return p && (((d.seq_init >> l) & 1) || (schedule >= 2 && schedule != 7 && l > 0)); }

    void gates_job(SkJob& j, int l, int t) const {
        const size_t BH = (size_t)d.B * d.H;
        sk_job_init(j);
        layer_segs(j, l, t, d.h[l] + t * BH, 0, 2 * d.H);
        j.M = d.B; j.N = 2 * d.H; j.H = d.H; j.epi = SK_EPI_GRU_GATES;
        j.bias = d.bg[l];
        j.add = has_seq(l, d.seq_g[l]) ? d.seq_g[l] + t * 2 * BH : nullptr; j.ld_add = 2 * d.H;
        j.e0 = d.h[l] + t * BH; j.lde0 = d.H;
        j.o1 = d.z[l] + t * BH; j.ldo1 = d.H;
        j.o2 = d.r[l] + t * BH; j.ldo2 = d.H;
        j.out = d.rh[l] + t * BH; j.ldo = d.H;
    }

    void cand_job(SkJob& j, int l, int t) const {
        const size_t BH = (size_t)d.B * d.H;
        sk_job_init(j);
        layer_segs(j, l, t, d.rh[l] + t * BH, 1, d.H);
        j.M = d.B; j.N = d.H; j.H = d.H; j.epi = SK_EPI_GRU_CAND;
        j.bias = d.bc[l];
        j.add = has_seq(l, d.seq_c[l]) ? d.seq_c[l] + t * BH : nullptr; j.ld_add = d.H;
        j.e0 = d.h[l] + t * BH; j.lde0 = d.H;
        j.e1 = d.z[l] + t * BH; j.lde1 = d.H;
        j.o1 = d.c[l] + t * BH; j.ldo1 = d.H;
        j.out = d.h[l] + (t + 1) * BH; j.ldo = d.H;
    }

    void lstm_job(SkJob& j, int l, int t) const {
        const size_t BH = (size_t)d.B * d.H;
        sk_job_init(j);
        layer_segs(j, l, t, d.h[l] + t * BH, 0, 4 * d.H);
        j.M = d.B; j.N = 4 * d.H; j.H = d.H; j.epi = SK_EPI_LSTM;
        j.bias = d.bg[l];
        j.add = has_seq(l, d.seq_g[l]) ? d.seq_g[l] + t * 4 * BH : nullptr; j.ld_add = 4 * d.H;
        j.e1 = d.cst[l] + t * BH; j.lde1 = d.H;
        j.o1 = d.cst[l] + (t + 1) * BH; j.ldo1 = d.H;
        j.o2 = d.gate4[l] + t * 4 * BH; j.ldo2 = 4 * d.H;
        j.out = d.h[l] + (t + 1) * BH; j.ldo = d.H;
    }

    AttFwdArgs att_fwd_args(int t) const {
        const size_t BH = (size_t)d.B * d.H, BE = (size_t)d.B * d.E, BA = (size_t)d.B * d.A;
        AttFwdArgs g{};
        g.h1 = d.h[0] + (t + 1) * BH; g.ldh = d.H;
        g.WattT = d.WattT; g.batt = d.batt;
        g.kappa_prev = d.kappa + t * BA;
        g.ctx = d.ctx;
        g.a_out = d.a + t * BA; g.b_out = d.b + t * BA; g.kappa_out = d.kappa + (t + 1) * BA;
        g.phi_out = d.phi + (size_t)t * d.B * d.U;
        g.w_out = d.w + (t + 1) * BE; g.ldw = d.E;
        g.B = d.B; g.H = d.H; g.A = d.A; g.U = d.U; g.E = d.E; g.esplit = esplit;
        g.att_type = d.att_type; g.eps = d.eps; g.alignment = d.alignment;
        g.sharpening = d.sharpening; g.timing = d.timing;
        g.sup_out = d.att_sup ? d.att_sup + (size_t)t * d.B * 2 : nullptr;
        return g;
    }
    int att_fwd_step(int t, hipStream_t st) const { return traced_att_fwd_launch(att_fwd_args(t), st); }

    // Forward wavefront: at tick q layer l advances step t = q - l, so the gate GEMMs of all layers share
    // one launch, the candidate GEMMs a second one, and the attention of step q is the third.  Layer
    // l >= 1 needs h_j(t) (j < l) and w_t, both produced in earlier ticks; layer 0 needs w_{t-1}.
    int nticks() const { return d.T + d.L - 1; }
    int fwd(hipStream_t st) { return fwd(st, 0, nticks()); }
    int fwd(hipStream_t st, int q0, int q1) {
        for (int q = q0; q < q1; ++q) {
            SkJob jobs[PARROT_MAX_LAYERS];
            int n = 0;
            if (d.cell == 1) {  // LSTM layers: a single fused GEMM + cell update per layer-step
                for (int l = 0; l < d.L; ++l) {
                    const int t = q - l;
                    if (t >= 0 && t < d.T) lstm_job(jobs[n++], l, t);
                }
                PL_TRY(launch_jobs(jobs, n, st, full_wgs));
                if (q < d.T) PL_TRY(att_fwd_step(q, st));
                continue;
            }
            for (int l = 0; l < d.L; ++l) {
                const int t = q - l;
                if (t >= 0 && t < d.T) gates_job(jobs[n++], l, t);
            }
            PL_TRY(launch_jobs(jobs, n, st, full_wgs));
            n = 0;
            for (int l = 0; l < d.L; ++l) {
                const int t = q - l;
                if (t >= 0 && t < d.T) cand_job(jobs[n++], l, t);
            }
            PL_TRY(launch_jobs(jobs, n, st, full_wgs));
            if (q < d.T) PL_TRY(att_fwd_step(q, st));
        }
        return 0;
    }

    // ---- schedule 5: balanced wavefront (GRU layers, L >= 2) --------------------------------------------------------
    // A step launch costs ~4.7 us + ~4.8 us per 1024 of the LONGEST K among its workgroups (tools/skbench4.hip,
    // profiles/r03_launch_cost_model.txt): in schedule 0 the upper layers' workgroups (K = 2H + E and more) set the
    // pace of both GEMM launches while layer 0's finish early, and the attention launch leaves the chip idle.  Here
    // the products of layer l >= 1 are cut in two: the INPUT projection [w_t ; h_0 .. h_{l-1}] . W[H:, :] (everything
    // that comes from below, ready one tick before it is needed) is a separate linear job that writes the layer's
    // additive-input buffer seq_g / seq_c and rides in the launch of the ATTENTION step (heterogeneous launch,
    // skinny.hip ska_kernel); the gate / candidate launches keep only the recurrent K = H of the upper layers next to
    // layer 0's K = H + E.  Layer l >= 1 lags l + 1 ticks.  Per tick q:
    //   A: gates(l0, q), gates_rec(l, q - l - 1)     B: cand(l0, q), cand_rec(l, q - l - 1)
    //   C: attention(q) + input projections of layer l for step q - l (all l >= 1)
    // The pre-activation of an upper layer is now (recurrent sum) + (input sum) instead of one running sum over the
    // concatenated K: same terms, other rounding (not bit-identical to schedule 0; the oracle tests cover both).
    int esplit5 = 1;
    bool s5_split = true;  // PARROT_S5_SPLIT=0: LSTM input projections of l >= 2 as one job
    int lag5(int l) const { return l == 0 ? 0 : l + 1; }
    int nticks5() const { return d.T + lag5(d.L - 1); }
    // part 0: the whole projection; 1: the rows of w and h_0 .. h_{l-2} (ready a tick earlier); 2: the rows of h_{l-1},
    // accumulated onto part 1 (LSTM stacks with l >= 2: two jobs of about the recurrent K instead of one of K = E + l H)
    void input_job(SkJob& j, int l, int t, int g, int part = 0) const {
        const size_t BH = (size_t)d.B * d.H, BE = (size_t)d.B * d.E;
        const int wd = d.cell == 1 ? 4 * d.H : (g == 0 ? 2 * d.H : d.H);  // LSTM layers: one 4H-wide matrix (g = 0)
        sk_job_init(j);
        j.colmode = d.cell == 1 && tiled ? 1 : 0;  // (the tiled copies of LSTM matrices keep the gate-interleaved tile order)
        int n = 0;
        if (part != 2) j.seg[n++] = fseg(d.w + (size_t)(t + 1) * BE, d.E, l, g, d.H, d.E, wd);
        for (int q = 0; q < l; ++q) {
            if ((part == 1 && q == l - 1) || (part == 2 && q != l - 1)) continue;
            j.seg[n++] = fseg(d.h[q] + (size_t)(t + 1) * BH, d.H, l, g, d.H + d.E + q * d.H, d.H, wd);
        }
        j.nseg = n;
        j.M = d.B; j.N = wd; j.H = d.H; j.epi = SK_EPI_LINEAR;
        float* sq = (g == 0 ? d.seq_g[l] : d.seq_c[l]) + (size_t)t * d.B * wd;
        j.out = sq; j.ldo = wd;
        j.accumulate = part == 2 ? 1 : ((d.seq_init >> l) & 1);  // caller data (feedback / speaker terms) already there
    }
    int fwd5(hipStream_t st) {
        const int Q = nticks5();
        const char* fe = getenv("PARROT_S5_FULL");
        const int cfull = fe ? atoi(fe) : 160;
        // launches of a tick: <= L gate jobs, <= L candidate jobs, <= 3 input-projection jobs per upper layer
        static_assert(3 * (PARROT_MAX_LAYERS - 1) <= SK_MAXJOB && PARROT_MAX_LAYERS <= SK_MAXJOB, "fwd5: jobs[] too short");
        for (int q = 0; q < Q; ++q) {
            SkJob jobs[SK_MAXJOB];
            int n = 0;
            for (int l = 0; l < d.L; ++l) {
                const int t = q - lag5(l);
                if (t < 0 || t >= d.T) continue;
                SkJob& j = jobs[n++];
                if (d.cell == 1) lstm_job(j, l, t);  // LSTM layers: one fused product + cell update per layer-step
                else gates_job(j, l, t);
                if (l > 0) j.nseg = 1;  // recurrent block only; the rest arrives through seq_g (has_seq)
            }
            if (n > 0) PL_TRY(launch_jobs(jobs, n, st, full_wgs));
            n = 0;
            if (d.cell == 0) {
                for (int l = 0; l < d.L; ++l) {
                    const int t = q - lag5(l);
                    if (t < 0 || t >= d.T) continue;
                    SkJob& j = jobs[n++];
                    cand_job(j, l, t);
                    if (l > 0) j.nseg = 1;
                }
                if (n > 0) PL_TRY(launch_jobs(jobs, n, st, full_wgs));
            }
            n = 0;
            for (int l = 1; l < d.L; ++l) {
                const int t = q - lag5(l) + 1;
                const bool split = d.cell == 1 && l >= 2 && s5_split;
                if (t >= 0 && t < d.T) {
                    input_job(jobs[n++], l, t, 0, split ? 2 : 0);
                    if (d.cell == 0) input_job(jobs[n++], l, t, 1);
                }
                if (split) {  // the rows that were ready a tick earlier
                    const int ta = t + 1;
                    if (ta >= 0 && ta < d.T) input_job(jobs[n++], l, ta, 0, 1);
                }
            }
            if (q < d.T) {
                AttFwdArgs ag = att_fwd_args(q);
                if (n > 0) ag.esplit = esplit5;  // beside GEMM workgroups: one attention workgroup per batch row (measured)
                PL_TRY(launch_jobs_att(jobs, n, ag, st, cfull));
            } else if (n > 0) PL_TRY(launch_jobs(jobs, n, st, full_wgs));
        }
        return 0;
    }


    // ---- schedule 7: ONE launch per tick for LSTM layers (round 4) ---------------------------------------------------
    // Schedule 0 runs an LSTM tick as the fused launch of all layers (wk_kernel at cfg4: ~35 us, bound by its total work)
    // followed by the attention step ALONE (~12.6 us: a chain of dependent round trips on 64 CUs, the other 192 idle).
    // The attention of step q-1 only feeds the LAST K = E rows of layer 0's product at step q (and the upper layers a
    // tick later), and an LSTM launch is three times as long as the attention chain.  So the attention rides at the head
    // of the next tick's launch: its blocks are dispatched first and publish w write-through plus an arrival count
    // (att_fwd_body.h, as in schedule 6); layer 0's workgroups -- the shortest K of the launch -- come LAST in the grid,
    // start on the CUs the attention blocks free, walk their h rows and take the w rows behind the flag (wk_body's tail;
    // sk_body's for f32 operands).  The upper layers lag one tick more than in schedule 0 so that the w they read was
    // published by an EARLIER launch:  tick q:  attention(q-1) || lstm(l0, q) [w rows flagged], lstm(l, q - lag7(l)),
    // lag7 = 0, 2, 3.  Same terms per output element as schedule 0 (the attention runs one block per batch row here, so
    // its sums differ from schedule 0's column-sliced blocks in the last bits).
    int lag7(int l) const { return l == 0 ? 0 : l + 1; }
    int nticks7() const { return d.T + std::max(1, lag7(d.L - 1)); }
    int fwd7(hipStream_t st) {
        if (!att_flags) return PARROT_ERR_BADARG;  // (allocated by parrot_decoder_create, outside any stream capture)
        if (!g_tracer) PL_TRY(sk_zero_words_launch(att_flags, d.T + 2, st));
        const int Q = nticks7();
        for (int q = 0; q < Q; ++q) {
            SkJob jobs[PARROT_MAX_LAYERS];
            int n = 0;
            const bool att_on = q >= 1 && q - 1 < d.T;
            AttFwdArgs ag{};
            if (att_on) {
                ag = att_fwd_args(q - 1);
                ag.esplit = 1;  // beside GEMM workgroups: one attention workgroup per batch row
            }
            for (int l = 0; l < d.L; ++l) {
                const int t = q - lag7(l);
                if (t < 0 || t >= d.T) continue;
                SkJob& j = jobs[n++];
                lstm_job(j, l, t);
                if (l == 0 && att_on) {  // w_{q-1} arrives inside this launch: its segment goes last and waits
                    if (j.nseg != 2) return PARROT_ERR_BADARG;
                    j.wait_flag = att_flags + q;
                    j.wait_target = (unsigned)(ag.B * ag.esplit);
                    ag.flag = att_flags + q;
                }
            }
            if (att_on) PL_TRY(launch_jobs_att(jobs, n, ag, st, 0));
            else if (n > 0) PL_TRY(launch_jobs(jobs, n, st, full_wgs));
        }
        return 0;
    }

    // Backward wavefront: at tick q layer l (upper layers first) handles step t = T-1-(q-(L-1-l)).
    // Per tick: attention backward of layer 0's step, then one elementwise launch, one launch of the
    // d(r*h) GEMMs and one launch of the input-gradient GEMMs for all active layers.
    // Gradient contributions that cross layers land in separate buffers (dhup[l] for the state, dw0
    // for layer 0's share of dw), so no two jobs of a launch update the same element: no atomics, and
    // the result is deterministic.  The consumers add the parts when they read.
    bool bwd_split = true;  // PARROT_BWD_SPLIT=0: the dC products stay in the Y launch (K = 3H jobs), as before round 3
    // schedule 7 with bf16 operands: the backward tick of LSTM layers as ONE launch (skinny.hip wkb_kernel); bwd_flags =
    // [ticks x 4 chains] arrival counters, plan-owned, zeroed at the head of the backward scan
    unsigned* att_flags = nullptr;  // [T + 2] arrival counters, one per tick (plan-owned, zeroed at the head of the scan)
    bool flags_fake = false;        // (placeholder for CPU-only schedule tracing: never dereferenced, never freed)
    bool bwd_fused = false;
    unsigned* bwd_flags = nullptr;
    // LSTM layers, bf16 operands, second accumulators given (ParrotDecoderDesc::dh_b ...): the backward products in two K
    // halves.  A wide workgroup streams its whole [B, 4H] operand: 156 workgroups of ~40 us each at cfg4, whatever
    // their width, and 100 idle CUs; two K halves = 312 workgroups of ~20 us (PARROT_BWD_KSPLIT=0: one part).
    bool bwd_ksplit = false;
    // (Layer 0's products in FOUR K parts were built and measured in round 4: cfg4 94.6 vs 91.4 ms -- 112 narrow workgroups
    // with a ring fill each cost more than the shorter stream returns; removed in round 5.)
    int bwd(hipStream_t st) { return bwd(st, 0, nticks()); }
    int bwd(hipStream_t st, int q0, int q1) {
        const size_t BH = (size_t)d.B * d.H, BE = (size_t)d.B * d.E;
        const int H = d.H, E = d.E;
        if (bwd_fused && q0 == 0 && !g_tracer) PL_TRY(sk_zero_words_launch(bwd_flags, 4 * nticks(), st));
        for (int q = q0; q < q1; ++q) {
            int tl[PARROT_MAX_LAYERS];
            for (int l = 0; l < d.L; ++l) tl[l] = d.T - 1 - (q - (d.L - 1 - l));
            const int t0 = tl[0];
            const bool att_on = t0 >= 0 && t0 < d.T;
            AttBwdArgs g{};
            if (att_on) g = att_bwd_args(t0);
            if (att_on && bwd_ksplit) {  // (LSTM layers) the second K halves' shares of dw
                g.dw3 = d.dw_b + (size_t)(t0 + 1) * BE;
                g.dw4 = d.dw0_b + (size_t)(t0 + 1) * BE;
            }
            if (d.cell == 1) {
                SkJob jl[SK_MAXJOB];
                int nl = 0;
                LstmStateBwdArgs la;
                la.nchain = 0; la.B = d.B; la.H = H;
                int chain_of[PARROT_MAX_LAYERS];
                for (int l = d.L - 1; l >= 0; --l) {  // state updates of all active layers + attention
                    chain_of[l] = -1;
                    const int t = tl[l];
                    if (t < 0 || t >= d.T) continue;
                    chain_of[l] = la.nchain;
                    LstmStateBwdChain& c = la.chain[la.nchain++];
                    c.dh = d.dh[l] + (t + 1) * BH;
                    c.dh2 = (l + 1 < d.L) ? d.dhup[l] + (t + 1) * BH : nullptr;
                    c.dh3 = bwd_ksplit ? d.dh_b[l] + (t + 1) * BH : nullptr;  // the second K halves' sums (below)
                    c.dh4 = (bwd_ksplit && l + 1 < d.L) ? d.dhup_b[l] + (t + 1) * BH : nullptr;
                    c.dh5 = c.dh6 = nullptr;
                    c.dc = d.dcell[l];
                    c.gates = d.gate4[l] + (size_t)t * 4 * BH;
                    c.c_prev = d.cst[l] + t * BH;
                    c.c_new = d.cst[l] + (t + 1) * BH;
                    c.dP = d.dG[l] + (size_t)t * 4 * BH;
                }
                for (int l = d.L - 1; l >= 0; --l) {
                    const int t = tl[l];
                    if (t < 0 || t >= d.T) continue;
                    float* dP = d.dG[l] + (size_t)t * 4 * BH;
                    const int first = nl;
                    {   // previous state of this layer
                        SkJob& j = jl[nl++];
                        sk_job_init(j);
                        j.nseg = 1;
                        j.seg[0] = rseg(dP, l, 0, 0, 4 * H);
                        j.M = d.B; j.N = H; j.H = H; j.epi = SK_EPI_LINEAR; j.accumulate = 1;
                        j.out = d.dh[l] + t * BH; j.ldo = H;
                        if (bwd_ksplit) { j.ksplit = 2; j.o1 = d.dh_b[l] + t * BH; j.ldo1 = H; }
                    }
                    {   // attention context
                        SkJob& j = jl[nl++];
                        sk_job_init(j);
                        j.nseg = 1;
                        j.seg[0] = rseg(dP, l, 0, H, 4 * H);
                        j.M = d.B; j.N = E; j.H = H; j.epi = SK_EPI_LINEAR; j.accumulate = 1;
                        j.out = (l == 0 ? d.dw0 + (size_t)t * BE : d.dw + (size_t)(t + 1) * BE); j.ldo = E;
                        if (bwd_ksplit) {
                            j.ksplit = 2; j.ldo1 = E;
                            j.o1 = (l == 0 ? d.dw0_b + (size_t)t * BE : d.dw_b + (size_t)(t + 1) * BE);
                            j.ldo2 = l == 0 ? 0 : 1;  // dw_b[t + 1] collects every upper layer's share: added (caller-zeroed)
                        }
                    }
                    for (int p = 0; p < l; ++p) {
                        SkJob& j = jl[nl++];
                        sk_job_init(j);
                        j.nseg = 1;
                        j.seg[0] = rseg(dP, l, 0, H + E + p * H, 4 * H);
                        j.M = d.B; j.N = H; j.H = H; j.epi = SK_EPI_LINEAR; j.accumulate = 1;
                        j.out = d.dhup[p] + (t + 1) * BH; j.ldo = H;
                        if (bwd_ksplit) { j.ksplit = 2; j.o1 = d.dhup_b[p] + (t + 1) * BH; j.ldo1 = H; j.ldo2 = 1; }  // (added: all layers above p)
                    }
                    if (bwd_fused)  // the products of a layer read what its chain's rows publish inside the launch
                        for (int q2 = first; q2 < nl; ++q2) {
                            jl[q2].wait_flag = bwd_flags + (size_t)q * 4 + chain_of[l];
                            jl[q2].wait_target = (unsigned)d.B;
                            jl[q2].wait_all = (l == 0 && att_on) ? 2 : 1;  // (2: behind the attention rows, last in the grid)
                        }
                }
                const int l0c = att_on ? la.nchain - 1 : -1;
                if (bwd_fused && la.nchain > 0 && nl > 0) {
                    unsigned* fl[4] = {nullptr, nullptr, nullptr, nullptr};
                    for (int c2 = 0; c2 < la.nchain; ++c2) fl[c2] = bwd_flags + (size_t)q * 4 + c2;
                    const int rc = traced_bwd_fused_launch(att_on ? &g : nullptr, la, l0c, jl, nl, fl, st);
                    if (rc != PARROT_ERR_UNSUPPORTED) {
                        PL_TRY(rc);
                        continue;
                    }
                    for (int q2 = 0; q2 < nl; ++q2) { jl[q2].wait_flag = nullptr; jl[q2].wait_target = 0; jl[q2].wait_all = 0; }
                }
                if (la.nchain > 0) PL_TRY(traced_att_state_bwd_launch(att_on ? &g : nullptr, la, l0c, st));
                if (nl > 0) PL_TRY(launch_jobs(jl, nl, st, full_wgs, 0, bwd_ksplit ? 1 : 0));
                continue;
            }
            GruStateBwdArgs ga;
            ga.nchain = 0; ga.B = d.B; ga.H = H;
            // the split backward tick carries 1 + 1 + l jobs per layer in each of the X and Y launches
            static_assert(PARROT_MAX_LAYERS * (PARROT_MAX_LAYERS + 3) / 2 <= SK_MAXJOB,
                          "backward tick: jx / jy cannot hold every layer's jobs");
            SkJob jx[SK_MAXJOB], jy[SK_MAXJOB];
            int nx = 0, ny = 0;
            for (int l = d.L - 1; l >= 0; --l) {
                const int t = tl[l];
                if (t < 0 || t >= d.T) continue;
                GruStateBwdChain& c = ga.chain[ga.nchain++];
                c.dh = d.dh[l] + (t + 1) * BH;
                c.dh2 = (l + 1 < d.L) ? d.dhup[l] + (t + 1) * BH : nullptr;
                c.hprev = d.h[l] + t * BH;
                c.z = d.z[l] + t * BH;
                c.c = d.c[l] + t * BH;
                c.mask = nullptr;
                c.dC = d.dC[l] + t * BH;
                c.dG = d.dG[l] + t * 2 * BH;
                c.dhprev = d.dh[l] + t * BH;

                // X: d(r*h_prev) = dC . Wc[0:H,:]^T ; epilogue -> dG_r, dh_prev += d(rh) * r
                SkJob& x = jx[nx++];
                sk_job_init(x);
                x.nseg = 1;
                x.seg[0] = rseg(d.dC[l] + t * BH, l, 1, 0, H);
                x.M = d.B; x.N = H; x.H = H; x.epi = SK_EPI_BWD_RH;
                x.e0 = d.h[l] + t * BH; x.lde0 = H;
                x.e1 = d.r[l] + t * BH; x.lde1 = H;
                x.out = d.dG[l] + t * 2 * BH + H; x.ldo = 2 * H;
                x.o1 = d.dh[l] + t * BH; x.ldo1 = H;

                // Y: gradients flowing to the layer's inputs, one job per destination.
                const float* dG = d.dG[l] + t * 2 * BH;
                const float* dC = d.dC[l] + t * BH;
                {   // previous state of this layer: only the gate GEMM (rh part handled by X)
                    SkJob& j = jy[ny++];
                    sk_job_init(j);
                    j.nseg = 1;
                    j.seg[0] = rseg(dG, l, 0, 0, 2 * H);
                    j.M = d.B; j.N = H; j.H = H; j.epi = SK_EPI_LINEAR; j.accumulate = 1;
                    j.out = d.dh[l] + t * BH; j.ldo = H;
                }
                // The products with dC (K = H) do not need the X launch's dG_r: with bwd_split they ride in the X launch
                // and the Y launch keeps the dG products (K = 2H) only, so no workgroup of a tick walks K = 3H any more
                // (a launch costs ~4.7 us + ~4.8 us per 1024 of its LONGEST K: 12.0 + 7.9 + 19.9 -> 12.0 + 9.5 + 14.3 us).
                {   // attention context
                    float* out = (l == 0 ? d.dw0 + (size_t)t * BE : d.dw + (size_t)(t + 1) * BE);
                    SkJob& j = jy[ny++];
                    sk_job_init(j);
                    j.nseg = bwd_split ? 1 : 2;
                    j.seg[0] = rseg(dG, l, 0, H, 2 * H);
                    j.seg[1] = rseg(dC, l, 1, H, H);
                    j.M = d.B; j.N = E; j.H = H; j.epi = SK_EPI_LINEAR; j.accumulate = 1;
                    j.out = out; j.ldo = E;
                    if (bwd_split) {
                        SkJob& k = jx[nx++];
                        sk_job_init(k);
                        k.nseg = 1;
                        k.seg[0] = rseg(dC, l, 1, H, H);
                        k.M = d.B; k.N = E; k.H = H; k.epi = SK_EPI_LINEAR; k.accumulate = 1;
                        k.out = out; k.ldo = E;
                    }
                }
                for (int p = 0; p < l; ++p) {  // lower layers' states of the same step
                    SkJob& j = jy[ny++];
                    sk_job_init(j);
                    j.nseg = bwd_split ? 1 : 2;
                    j.seg[0] = rseg(dG, l, 0, H + E + p * H, 2 * H);
                    j.seg[1] = rseg(dC, l, 1, H + E + p * H, H);
                    j.M = d.B; j.N = H; j.H = H; j.epi = SK_EPI_LINEAR; j.accumulate = 1;
                    j.out = d.dhup[p] + (t + 1) * BH; j.ldo = H;  // separate buffer: no two jobs share a tile
                    if (bwd_split) {
                        SkJob& k = jx[nx++];
                        sk_job_init(k);
                        k.nseg = 1;
                        k.seg[0] = rseg(dC, l, 1, H + E + p * H, H);
                        k.M = d.B; k.N = H; k.H = H; k.epi = SK_EPI_LINEAR; k.accumulate = 1;
                        k.out = d.dhup[p] + (t + 1) * BH; k.ldo = H;
                    }
                }
            }
            if (ga.nchain == 0) continue;
            // layer 0's chain (if active) is the last one added; attention + all state updates in one launch
            PL_TRY(traced_att_state_bwd_launch(att_on ? &g : nullptr, ga, att_on ? ga.nchain - 1 : -1, st));
            PL_TRY(launch_jobs(jx, nx, st, full_wgs));
            PL_TRY(launch_jobs(jy, ny, st, full_wgs));
        }
        return 0;
    }

    // ---- bwd8: the K-balanced backward tick (2-layer f32 GRU decoders; round 4) ---------------------------------------
    // The tick of bwd() is three dependent launches: attention + state backward (11.1 us at cfg2: a chain of dependent
    // round trips on 64-128 CUs, the rest idle), X (d(rh) and the dC products, K = H: 9.5 us) and Y (the dG products,
    // K = 2H: 14.3 us -- a launch costs ~4.7 us + ~4.8 us per 1024 of its LONGEST K, tools/tick_model.py).  A tick is 672
    // units of 32 x 32 x K1024 work, 2.6 rounds of 256 CUs, so three launches are the minimum -- but they need not be
    // one idle launch, one short and one long.  Here
    //   * every K = 2H product is cut into its update-gate (z) and reset-gate (r) halves, K = H each, writing SEPARATE
    //     buffers that the consumer adds (second / third accumulators of ParrotDecoderDesc): dG_z exists after the state
    //     backward, dG_r only after X, so the z halves move up a launch;
    //   * layer 1 runs TWO ticks ahead of layer 0, so its downward products (into dhup_0 and dw) have a tick of slack;
    //   * those with dG operands ride in the NEXT tick's attention launch as step-GEMM workgroups beside the attention
    //     backward blocks (skinny.hip skb_kernel), the dC ones in Y.
    // Per tick (L = 2, cfg2): S' = 64 attention rows + 64 state rows + 160 GEMM workgroups, X = 256, Y = 256, every K = H:
    // predicted 12.1 + 9.5 + 9.5 = 31.1 us against 34.9 (profiles/r04_tick_model_whatif.txt).
    bool bwd_hetero = false;
    int lag8(int l) const { return 2 * (d.L - 1 - l); }
    int nticks8() const { return d.T + 2 * (d.L - 1); }
    // backward product dP[:, k0 : k0 + K] . W[r0 : r0 + N, k0 : k0 + K]^T over the fragment-major reverse copies
    SkSeg rseg_k(const float* A, int l, int g, int r0, int ldw, int k0, int K) const {
        const float* Wt = g == 0 ? d.Wg_r[l] : d.Wc_r[l];
        return sk_seg(A + k0, ldw, Wt + ((size_t)(r0 >> 4) * (ldw >> 4) + (k0 >> 4)) * 256, (ldw >> 4) * 256, K, 2);
    }
    static void lin_job(SkJob& j, const SkSeg& sg, int M, int N, int H, float* out, int ldo, int accumulate) {
        sk_job_init(j);
        j.nseg = 1;
        j.seg[0] = sg;
        j.M = M; j.N = N; j.H = H; j.epi = SK_EPI_LINEAR; j.accumulate = accumulate;
        j.out = out; j.ldo = ldo;
    }
    int bwd8(hipStream_t st) {
        const size_t BH = (size_t)d.B * d.H, BE = (size_t)d.B * d.E;
        const int H = d.H, E = d.E, Q = nticks8();
        for (int q = 0; q < Q; ++q) {
            int tl[PARROT_MAX_LAYERS];
            for (int l = 0; l < d.L; ++l) tl[l] = d.T - 1 - (q - lag8(l));
            const int t0 = tl[0];
            const bool att_on = t0 >= 0 && t0 < d.T;
            AttBwdArgs g{};
            if (att_on) {
                g = att_bwd_args(t0);
                g.dw3 = d.dw0_b + (size_t)(t0 + 1) * BE; g.dw4 = d.dw0_c + (size_t)(t0 + 1) * BE;
                g.dw5 = d.dw_b + (size_t)(t0 + 1) * BE;  g.dw6 = d.dw_c + (size_t)(t0 + 1) * BE;
            }
            // ---- S': state backward of every active layer (+ attention), and the deferred downward products
            GruStateBwdArgs ga;
            ga.nchain = 0; ga.B = d.B; ga.H = H;
            for (int l = d.L - 1; l >= 0; --l) {
                const int t = tl[l];
                if (t < 0 || t >= d.T) continue;
                GruStateBwdChain& c = ga.chain[ga.nchain++];
                c.dh = d.dh[l] + (t + 1) * BH;
                c.dh2 = (l + 1 < d.L) ? d.dhup[l] + (t + 1) * BH : nullptr;
                c.dhx[0] = d.dh_b[l] + (t + 1) * BH;
                c.dhx[1] = (l + 1 < d.L) ? d.dhup_b[l] + (t + 1) * BH : nullptr;
                c.dhx[2] = (l + 1 < d.L) ? d.dhup_c[l] + (t + 1) * BH : nullptr;
                c.hprev = d.h[l] + t * BH;
                c.z = d.z[l] + t * BH;
                c.c = d.c[l] + t * BH;
                c.mask = nullptr;
                c.dC = d.dC[l] + t * BH;
                c.dG = d.dG[l] + t * 2 * BH;
                c.dhprev = d.dh[l] + t * BH;
            }
            SkJob js[SK_MAXJOB], jx[SK_MAXJOB], jy[SK_MAXJOB];
            int ns = 0, nx = 0, ny = 0;
            for (int l = d.L - 1; l >= 1; --l) {  // the dG halves of the step layer l handled one tick ago
                const int s = tl[l] + 1;
                if (s < 0 || s >= d.T) continue;
                const float* dG = d.dG[l] + (size_t)s * 2 * BH;
                for (int p = 0; p < l; ++p) {
                    lin_job(js[ns++], rseg_k(dG, l, 0, H + E + p * H, 2 * H, 0, H), d.B, H, H, d.dhup_b[p] + (s + 1) * BH, H, 1);
                    lin_job(js[ns++], rseg_k(dG, l, 0, H + E + p * H, 2 * H, H, H), d.B, H, H, d.dhup_c[p] + (s + 1) * BH, H, 1);
                }
                lin_job(js[ns++], rseg_k(dG, l, 0, H, 2 * H, 0, H), d.B, E, H, d.dw_b + (size_t)(s + 1) * BE, E, 1);
                lin_job(js[ns++], rseg_k(dG, l, 0, H, 2 * H, H, H), d.B, E, H, d.dw_c + (size_t)(s + 1) * BE, E, 1);
            }
            // ---- X: d(r*h) (epilogue: dG_r, dh_prev += d(rh) * r) and the update-gate half of dG -> dh_prev (own buffer)
            // ---- Y: the reset-gate half of dG -> dh_prev, layer 0's context shares, the upper layers' dC shares downward
            for (int l = d.L - 1; l >= 0; --l) {
                const int t = tl[l];
                if (t < 0 || t >= d.T) continue;
                const float* dG = d.dG[l] + (size_t)t * 2 * BH;
                const float* dC = d.dC[l] + t * BH;
                SkJob& x = jx[nx++];
                sk_job_init(x);
                x.nseg = 1;
                x.seg[0] = rseg(dC, l, 1, 0, H);
                x.M = d.B; x.N = H; x.H = H; x.epi = SK_EPI_BWD_RH;
                x.e0 = d.h[l] + t * BH; x.lde0 = H;
                x.e1 = d.r[l] + t * BH; x.lde1 = H;
                x.out = d.dG[l] + t * 2 * BH + H; x.ldo = 2 * H;
                x.o1 = d.dh[l] + t * BH; x.ldo1 = H;
                lin_job(jx[nx++], rseg_k(dG, l, 0, 0, 2 * H, 0, H), d.B, H, H, d.dh_b[l] + t * BH, H, 0);
                lin_job(jy[ny++], rseg_k(dG, l, 0, 0, 2 * H, H, H), d.B, H, H, d.dh[l] + t * BH, H, 1);
                if (l == 0) {
                    lin_job(jy[ny++], rseg(dC, 0, 1, H, H), d.B, E, H, d.dw0 + (size_t)t * BE, E, 1);
                    lin_job(jy[ny++], rseg_k(dG, 0, 0, H, 2 * H, 0, H), d.B, E, H, d.dw0_b + (size_t)t * BE, E, 0);
                    lin_job(jy[ny++], rseg_k(dG, 0, 0, H, 2 * H, H, H), d.B, E, H, d.dw0_c + (size_t)t * BE, E, 0);
                } else {
                    lin_job(jy[ny++], rseg(dC, l, 1, H, H), d.B, E, H, d.dw + (size_t)(t + 1) * BE, E, 1);
                    for (int p = 0; p < l; ++p)
                        lin_job(jy[ny++], rseg(dC, l, 1, H + E + p * H, H), d.B, H, H, d.dhup[p] + (t + 1) * BH, H, 1);
                }
            }
            if (ns > SK_MAXJOB || nx > SK_MAXJOB || ny > SK_MAXJOB) return PARROT_ERR_BADARG;
            if (ga.nchain > 0) {
                const int l0c = att_on ? ga.nchain - 1 : -1;
                if (ns > 0) PL_TRY(traced_bwd_hetero_launch(att_on ? &g : nullptr, ga, l0c, js, ns, st));
                else PL_TRY(traced_att_state_bwd_launch(att_on ? &g : nullptr, ga, l0c, st));
            } else if (ns > 0) {
                PL_TRY(launch_jobs(js, ns, st, full_wgs));
            }
            if (nx > 0) PL_TRY(launch_jobs(jx, nx, st, full_wgs));
            if (ny > 0) PL_TRY(launch_jobs(jy, ny, st, full_wgs));
        }
        return 0;
    }

    // ---- chunked layer pipeline (default for L >= 2) ----------------------------------------------
    // Layer l >= 1 only consumes finished outputs of the layers below (h_j(t), w_t), never the other way
    // round.  So the scan is run layer by layer over chunks of `chunk` steps: once layer l-1 has finished
    // a chunk, the projections of its outputs into layer l (the Fork bricks h{j}_to_h{l} / inp_to_h{l},
    // model.py:692-722) are taken for the whole chunk by the LDS-tiled GEMM (M = chunk*B rows instead of
    // B) into the layer's additive-input buffer seq_g/seq_c, and the sequential part of layer l shrinks to
    // its own recurrent block h_l . W[0:H].  Each layer runs on its own stream; the only cross-stream
    // edges are one event per (layer, chunk), so layer 0's latency-bound chain (GEMM -> attention per
    // step) overlaps with the upper layers' work instead of adding to it.
    int hoist_fwd(int l, int t0, int t1, hipStream_t st) const {
        const int H = d.H, E = d.E, R = (t1 - t0) * d.B;
        const size_t BH = (size_t)d.B * H, BE = (size_t)d.B * E;
        const int ng = d.cell == 1 ? 1 : 2;
        for (int g = 0; g < ng; ++g) {
            const int wd = d.cell == 1 ? 4 * H : (g == 0 ? 2 * H : H);
            const float* W = g == 0 ? d.Wg[l] : d.Wc[l];
            float* out = (g == 0 ? d.seq_g[l] : d.seq_c[l]) + (size_t)t0 * d.B * wd;
            int acc = (d.seq_init >> l) & 1;
            PL_TRY(parrot_gemm(d.w + (size_t)(t0 + 1) * BE, E, 0, W + (size_t)H * wd, wd, 0, out, wd, R, wd, E,
                               nullptr, 1.f, acc, 0, 1, 0, 0, 0, 1, st));
            for (int j = 0; j < l; ++j) {
                const float* A = d.h[j] + (size_t)(t0 + 1) * BH;
                const float* Wj = W + (size_t)(H + E + j * H) * wd;
                if (!d.layer_norm) {
                    PL_TRY(parrot_gemm(A, H, 0, Wj, wd, 0, out, wd, R, wd, H, nullptr, 1.f, 1, 0, 1, 0, 0, 0, 1, st));
                    continue;
                }
                // layer_norm: project (with the Fork's own bias), normalise each row, then add (model.py:703-722)
                const int pj = l * PARROT_MAX_LAYERS + j;
                float* y = (g == 0 ? d.ln_yg[pj] : d.ln_yc[pj]) + (size_t)t0 * d.B * wd;
                float* sg = (g == 0 ? d.ln_sg[pj] : d.ln_sc[pj]) + (size_t)t0 * d.B;
                PL_TRY(parrot_gemm(A, H, 0, Wj, wd, 0, y, wd, R, wd, H, g == 0 ? d.ln_bg[pj] : d.ln_bc[pj], 1.f, 0, 0,
                                   1, 0, 0, 0, 1, st));
                PL_TRY(simple_norm_fwd_launch(y, wd, y, wd, sg, R, wd, PARROT_NORM_EPS, out, wd, st));
            }
        }
        return 0;
    }

    // Gradients of the hoisted projections for one chunk: dw[t+1] and dhup[p][t+1] += dPre . W^T.
    int hoist_bwd(int l, int t0, int t1, hipStream_t st) const {
        const int H = d.H, E = d.E, R = (t1 - t0) * d.B;
        const size_t BH = (size_t)d.B * H, BE = (size_t)d.B * E;
        const int ng = d.cell == 1 ? 1 : 2;
        for (int g = 0; g < ng; ++g) {
            const int wd = d.cell == 1 ? 4 * H : (g == 0 ? 2 * H : H);
            const float* W = g == 0 ? d.Wg[l] : d.Wc[l];
            const float* dP = (g == 0 ? d.dG[l] : d.dC[l]) + (size_t)t0 * d.B * wd;
            PL_TRY(parrot_gemm(dP, wd, 0, W + (size_t)H * wd, wd, 1, d.dw + (size_t)(t0 + 1) * BE, E, R, E, wd,
                               nullptr, 1.f, 1, 0, 1, 0, 0, 0, 1, st));
            for (int p = 0; p < l; ++p) {
                const float* dsrc = dP;
                if (d.layer_norm) {  // back through the row normalisation; the pre-norm gradient replaces y
                    const int pj = l * PARROT_MAX_LAYERS + p;
                    float* y = (g == 0 ? d.ln_yg[pj] : d.ln_yc[pj]) + (size_t)t0 * d.B * wd;
                    const float* sg = (g == 0 ? d.ln_sg[pj] : d.ln_sc[pj]) + (size_t)t0 * d.B;
                    PL_TRY(simple_norm_bwd_launch(dP, wd, y, wd, sg, y, wd, R, wd, PARROT_NORM_EPS, 0, st));
                    dsrc = y;
                }
                PL_TRY(parrot_gemm(dsrc, wd, 0, W + (size_t)(H + E + p * H) * wd, wd, 1,
                                   d.dhup[p] + (size_t)(t0 + 1) * BH, H, R, H, wd, nullptr, 1.f, 1, 0, 1, 0, 0, 0, 1,
                                   st));
            }
        }
        return 0;
    }

    void own_segs(SkJob& j, int l, int t, const float* first, int g, int ldw) const {
        const size_t BE = (size_t)d.B * d.E;
        j.seg[0] = fseg(first, d.H, l, g, 0, d.H, ldw);
        j.nseg = 1;
        if (l == 0) j.seg[j.nseg++] = fseg(d.w + (size_t)t * BE, d.E, l, g, d.H, d.E, ldw);
    }

    // Kernels of layer l for the steps of chunk c (forward): batched projections from below, then the
    // layer's own sequential chain.

    // Arguments of the attention backward of step t0 (one place: four schedules use it).
    AttBwdArgs att_bwd_args(int t0) const {
        const size_t BH = (size_t)d.B * d.H, BE = (size_t)d.B * d.E, BA = (size_t)d.B * d.A;
        AttBwdArgs g{};
        g.dw = d.dw + (t0 + 1) * BE; g.dw2 = d.dw0 + (t0 + 1) * BE; g.lddw = d.E;
        g.ctx = d.ctx;
        g.a = d.a + t0 * BA; g.b = d.b + t0 * BA;
        g.kappa = d.kappa + (t0 + 1) * BA; g.kappa_prev = d.kappa + t0 * BA;
        g.WattT = d.WattT;
        g.dkappa = d.dkappa;
        g.dp_out = d.dp + (size_t)t0 * d.B * 3 * d.A;
        g.sup = d.att_sup ? d.att_sup + (size_t)t0 * d.B * 2 : nullptr;
        g.dh1 = d.dh[0] + (t0 + 1) * BH; g.lddh = d.H;
        g.B = d.B; g.H = d.H; g.A = d.A; g.U = d.U; g.E = d.E; g.att_type = d.att_type; g.eps = d.eps;
        return g;
    }
    int att_bwd_step(int t0, hipStream_t st) const { return att_bwd_launch(att_bwd_args(t0), st); }


    // ---- schedule 3: skewed wavefront with hoisting -------------------------------------------------
    // Merged launches as in schedule 0, but layer l lags layer l-1 by one CHUNK of steps instead of one step.
    // When layer l-1 has finished a chunk, the Fork projections of its outputs (and of w) into layer l are taken
    // for the whole chunk by the LDS-tiled GEMM (hoist_fwd), so every per-step job keeps only the layer's own
    // recurrent block (+ w_{t-1} for layer 0): the slowest workgroups of a merged launch shrink from
    // K = H+E+lH to K <= H+E, and ~36 % of the step-kernel flops move to a kernel that runs at 115+ TFLOP/s.
    // One stream, one graph; costs (L-1) extra chunks of (light) ticks at the ends.
    int fwd_skew(hipStream_t st) {
        const size_t BH = (size_t)d.B * d.H;
        const int C = ceil_div(d.T, chunk);
        for (int sc = 0; sc < C + d.L - 1; ++sc) {
            for (int l = 1; l < d.L; ++l) {
                const int c = sc - l;
                if (c >= 0 && c < C) PL_TRY(hoist_fwd(l, c * chunk, (c + 1) * chunk < d.T ? (c + 1) * chunk : d.T, st));
            }
            for (int s = 0; s < chunk; ++s) {
                SkJob jobs[PARROT_MAX_LAYERS];
                int n = 0, t0 = -1;
                for (int l = 0; l < d.L; ++l) {
                    const int c = sc - l, t = c * chunk + s;
                    if (c < 0 || c >= C || t >= d.T) continue;
                    if (l == 0) t0 = t;
                    if (d.cell == 1) {
                        lstm_job(jobs[n], l, t);
                        own_segs(jobs[n], l, t, d.h[l] + t * BH, 0, 4 * d.H);
                    } else {
                        gates_job(jobs[n], l, t);
                        own_segs(jobs[n], l, t, d.h[l] + t * BH, 0, 2 * d.H);
                    }
                    ++n;
                }
                if (n == 0) continue;
                PL_TRY(launch_jobs(jobs, n, st));
                if (d.cell == 0) {
                    n = 0;
                    for (int l = 0; l < d.L; ++l) {
                        const int c = sc - l, t = c * chunk + s;
                        if (c < 0 || c >= C || t >= d.T) continue;
                        cand_job(jobs[n], l, t);
                        own_segs(jobs[n], l, t, d.rh[l] + t * BH, 1, d.H);
                        ++n;
                    }
                    PL_TRY(launch_jobs(jobs, n, st));
                }
                if (t0 >= 0) PL_TRY(att_fwd_step(t0, st));
            }
        }
        return 0;
    }

    int bwd_skew(hipStream_t st) {
        const size_t BH = (size_t)d.B * d.H, BE = (size_t)d.B * d.E, BA = (size_t)d.B * d.A;
        const int H = d.H, E = d.E;
        const int C = ceil_div(d.T, chunk);
        for (int sc = 0; sc < C + d.L - 1; ++sc) {
            int cl[PARROT_MAX_LAYERS];
            for (int l = 0; l < d.L; ++l) cl[l] = C - 1 - (sc - (d.L - 1 - l));  // upper layers lead
            for (int s = chunk - 1; s >= 0; --s) {
                int tl[PARROT_MAX_LAYERS];
                for (int l = 0; l < d.L; ++l) {
                    const int t = cl[l] * chunk + s;
                    tl[l] = (cl[l] >= 0 && cl[l] < C && t < d.T) ? t : -1;
                }
                const int t0 = tl[0];
                AttBwdArgs g{};
                if (t0 >= 0) {
                    g = att_bwd_args(t0);
                    if (d.cell == 1) PL_TRY(att_bwd_launch(g, st));
                }
                GruStateBwdArgs ga;
                ga.nchain = 0; ga.B = d.B; ga.H = H;
                SkJob jx[PARROT_MAX_LAYERS], jy[2 * PARROT_MAX_LAYERS];
                int nx = 0, ny = 0;
                for (int l = d.L - 1; l >= 0; --l) {
                    const int t = tl[l];
                    if (t < 0) continue;
                    const float* dh2 = (l + 1 < d.L) ? d.dhup[l] + (t + 1) * BH : nullptr;
                    if (d.cell == 1) {
                        float* dP = d.dG[l] + (size_t)t * 4 * BH;
                        PL_TRY(lstm_state_bwd_launch(d.dh[l] + (t + 1) * BH, dh2, d.dcell[l],
                                                     d.gate4[l] + (size_t)t * 4 * BH, d.cst[l] + t * BH,
                                                     d.cst[l] + (t + 1) * BH, dP, d.B, H, st));
                        SkJob& j = jy[ny++];
                        sk_job_init(j);
                        j.nseg = 1;
                        j.seg[0] = rseg(dP, l, 0, 0, 4 * H);
                        j.M = d.B; j.N = H; j.H = H; j.epi = SK_EPI_LINEAR; j.accumulate = 1;
                        j.out = d.dh[l] + t * BH; j.ldo = H;
                        if (l == 0) {
                            SkJob& k = jy[ny++];
                            sk_job_init(k);
                            k.nseg = 1;
                            k.seg[0] = rseg(dP, 0, 0, H, 4 * H);
                            k.M = d.B; k.N = E; k.H = H; k.epi = SK_EPI_LINEAR; k.accumulate = 1;
                            k.out = d.dw0 + (size_t)t * BE; k.ldo = E;
                        }
                        continue;
                    }
                    GruStateBwdChain& ch = ga.chain[ga.nchain++];
                    ch.dh = d.dh[l] + (t + 1) * BH;
                    ch.dh2 = dh2;
                    ch.hprev = d.h[l] + t * BH;
                    ch.z = d.z[l] + t * BH;
                    ch.c = d.c[l] + t * BH;
                    ch.mask = nullptr;
                    ch.dC = d.dC[l] + t * BH;
                    ch.dG = d.dG[l] + t * 2 * BH;
                    ch.dhprev = d.dh[l] + t * BH;
                    SkJob& x = jx[nx++];
                    sk_job_init(x);
                    x.nseg = 1;
                    x.seg[0] = rseg(d.dC[l] + t * BH, l, 1, 0, H);
                    x.M = d.B; x.N = H; x.H = H; x.epi = SK_EPI_BWD_RH;
                    x.e0 = d.h[l] + t * BH; x.lde0 = H;
                    x.e1 = d.r[l] + t * BH; x.lde1 = H;
                    x.out = d.dG[l] + t * 2 * BH + H; x.ldo = 2 * H;
                    x.o1 = d.dh[l] + t * BH; x.ldo1 = H;
                    const float* dG = d.dG[l] + t * 2 * BH;
                    const float* dC = d.dC[l] + t * BH;
                    {
                        SkJob& j = jy[ny++];
                        sk_job_init(j);
                        j.nseg = 1;
                        j.seg[0] = rseg(dG, l, 0, 0, 2 * H);
                        j.M = d.B; j.N = H; j.H = H; j.epi = SK_EPI_LINEAR; j.accumulate = 1;
                        j.out = d.dh[l] + t * BH; j.ldo = H;
                    }
                    if (l == 0) {
                        SkJob& j = jy[ny++];
                        sk_job_init(j);
                        j.nseg = 2;
                        j.seg[0] = rseg(dG, 0, 0, H, 2 * H);
                        j.seg[1] = rseg(dC, 0, 1, H, H);
                        j.M = d.B; j.N = E; j.H = H; j.epi = SK_EPI_LINEAR; j.accumulate = 1;
                        j.out = d.dw0 + (size_t)t * BE; j.ldo = E;
                    }
                }
                if (d.cell == 0) {
                    if (ga.nchain == 0) continue;
                    PL_TRY(att_state_bwd_launch(t0 >= 0 ? &g : nullptr, ga, t0 >= 0 ? ga.nchain - 1 : -1, st));
                    PL_TRY(launch_jobs(jx, nx, st));
                }
                if (ny > 0) PL_TRY(launch_jobs(jy, ny, st));
            }
            for (int l = 1; l < d.L; ++l) {
                const int c = cl[l];
                if (c >= 0 && c < C) PL_TRY(hoist_bwd(l, c * chunk, (c + 1) * chunk < d.T ? (c + 1) * chunk : d.T, st));
            }
        }
        return 0;
    }


    int run(int which, hipStream_t s) override {
        BgPrecisionScope precision(d.bf16);
        return PlanBase::run(which, s);
    }

    ~DecoderPlan() override {
        if (att_flags && !flags_fake) (void)hipFree(att_flags);
        if (bwd_flags && !flags_fake) (void)hipFree(bwd_flags);
    }
};

// ----------------------------------------------------------------------------- decoder (sampling)
struct SamplePlan : PlanBase {
    ParrotSampleDesc d;
    int esplit = 1;

    int enqueue(int, hipStream_t s) override { return persist_ok ? run_persist(s) : run_all(s); }

    // ---- persistent phase machine for the decode loop (persist.h) ------------------------------------------------
    // One resident kernel runs all S steps; a step = 2L + 3 phases: G_0, C_0, ATT, (G_l, C_l for l >= 1), readout,
    // output.  With output feedback (x_{t-1} -> layer inputs, model.py:899-924) the whole step is one dependency chain,
    // so every phase is on the critical path and costs its fixed latency (~4 us) instead of a launch (~16 us at
    // M = 16).  Each unit reads ONE fragment-major slab assembled by its producers:
    //   XG[l][t] / XC[l][t] = [h_l[t] or r*h_l ; w ; h_0[t+1] .. h_{l-1}[t+1] ; x[t] (64 columns, zero padded)]
    //   XR[t] = [h_0[t+1] .. h_{L-1}[t+1] ; w[t+1]]      XO[t] = readout[t]
    // Weights: fragment-major copies prepared by the caller (Wg_t / Wc_t: packed layer matrix with the feedback rows
    // appended and padded to 64; Wr_t; Wo_t with the columns padded to 64).
    bool persist_ok = false;
    PmProgram pm_prog;
    float* hist_h[PARROT_MAX_LAYERS] = {nullptr, nullptr, nullptr};

    static bool persist_eligible_shape(const ParrotSampleDesc& d) {  // (no device query: the CPU tests plan too)
        if (d.cell != 0 || d.layer_norm || d.gmm_K > 0 || d.B > 64 || (d.H % 16) || (d.E % 16) || (d.R % 16) ||
            d.U > PM_ATT_MAXU || d.A > PM_ATT_MAXA || d.S < 1 || d.O > 64 || d.ldx < 64 || (d.ldx % 4))
            return false;
        for (int l = 0; l < d.L; ++l)
            if (!d.Wg_t[l] || !d.Wc_t[l]) return false;
        return true;
    }
    static bool legacy_eligible(const ParrotSampleDesc& d) {
        if (2 * d.L + 3 > PM_MAXSLOTS || !d.Wr_t || !d.Wo_t || !d.bo_pad) return false;
        return (d.oadd != nullptr) == (d.oadd_pad != nullptr);
    }
    static bool persist_eligible(const ParrotSampleDesc& d) {
        if (!persist_eligible_shape(d) || !(legacy_eligible(d) || pieces_wanted(d))) return false;
        return pm_max_workgroups() >= 64;
    }
    static int fb_rows(const ParrotSampleDesc& d, int l) { return d.Wfg[l] ? 64 : 0; }
    static long long kslab(const ParrotSampleDesc& d, int l) { return d.H + d.E + (long long)l * d.H + fb_rows(d, l); }
    static long long persist_floats(const ParrotSampleDesc& d, int nwg) {
        const int MB = d.B <= 16 ? 1 : (d.B <= 32 ? 2 : 4);
        const long long rows = (long long)MB * 16, S = d.S;
        long long n = PM_SYNC_WORDS + PM_DBG_WORDS;
        n += ((long long)(2 * d.L + 3) * nwg * sizeof(PmUnit) + 3) / 4 + 64;
        for (int l = 0; l < d.L; ++l) n += 2 * (S + 1) * rows * kslab(d, l);
        n += S * rows * ((long long)d.L * d.H + d.E) + S * rows * d.R;
        n += (long long)d.L * (S + 1) * d.B * d.H + (long long)d.L * S * d.B * d.H;   // h and z histories (row-major)
        n += S * d.B * d.R + S * d.B * d.A;
        return n + piece_floats(d) + 4096;
    }

    int build_persist() {
        persist_ok = false;
        const char* e = getenv("PARROT_SAMPLE_PERSIST");
        if (e && atoi(e) == 0) return 0;
        if (!persist_eligible(d) || !d.persist_ws) return 0;
        if (pieces_wanted(d)) {  // the step cut along K by the age of its operands (below); else the 2L + 3 whole-K phases
            build_persist_pieces(false, 0);
            if (persist_ok) return 0;
        }
        if (!legacy_eligible(d)) return 0;
        const int nwg = pm_max_workgroups();
        if (d.persist_ws_floats < persist_floats(d, nwg)) return 0;
        const int H = d.H, E = d.E, B = d.B, L = d.L, S = d.S, R = d.R;
        const int MB = B <= 16 ? 1 : (B <= 32 ? 2 : 4);
        const long long rows = (long long)MB * 16, BH = (long long)B * H;
        const int n_slots = 2 * L + 3;
        float* ws = d.persist_ws;
        auto take = [&](long long n) { float* p = ws; ws += (n + 3) / 4 * 4; return p; };
        unsigned* sync = reinterpret_cast<unsigned*>(take(PM_SYNC_WORDS + PM_DBG_WORDS));
        const size_t unit_bytes = (size_t)n_slots * nwg * sizeof(PmUnit);
        PmUnit* units_dev = reinterpret_cast<PmUnit*>(take((long long)(unit_bytes + 3) / 4 + 16));
        float* fm_base = ws;
        float* XG[PARROT_MAX_LAYERS];
        float* XC[PARROT_MAX_LAYERS];
        long long kx[PARROT_MAX_LAYERS];
        for (int l = 0; l < L; ++l) {
            kx[l] = kslab(d, l);
            XG[l] = take((S + 1) * rows * kx[l]);
            XC[l] = take((S + 1) * rows * kx[l]);
        }
        const long long kr = (long long)L * H + E;
        float* XR = take(S * rows * kr);
        float* XO = take(S * rows * R);
        if ((long long)(ws - fm_base) * 4 >= 0xfff00000ll) return 0;
        float* const fm_end = ws;
        float* zh[PARROT_MAX_LAYERS];
        for (int l = 0; l < L; ++l) hist_h[l] = take((S + 1) * BH);
        for (int l = 0; l < L; ++l) zh[l] = take(S * BH);
        float* ro_hist = take((long long)S * B * R);
        float* b_hist = take((long long)S * B * d.A);

        auto boff = [&](const float* p) { return (unsigned)((p - fm_base) * 4); };
        auto mkdst = [&](float* slab, long long step0, long long ks, int chunk) {
            PmDst q;
            q.off = boff(slab + step0 * rows * ks); q.st = (unsigned)(rows * ks * 4); q.nch = (int)(ks / 16); q.chunk = chunk;
            return q;
        };
        auto rm = [](const float* p, long long st, int ld) { PmRM r; r.p = const_cast<float*>(p); r.st = st; r.ld = ld; r.pad = 0; return r; };
        std::vector<PmReq> reqs;
        auto gemm_unit = [&](int slot, float* slab, long long ks) {
            PmReq q;
            memset(&q, 0, sizeof(q));
            q.u.kind = PM_GEMM; q.u.M = B; q.u.w_lds = -1;
            q.u.a_off = boff(slab); q.u.a_st = (unsigned)(rows * ks * 4); q.u.a_nch = (int)(ks / 16); q.u.K = (int)ks;
            q.slot = slot; q.crit = 1; q.krows = (int)ks;
            return q;
        };
        for (int l = 0; l < L; ++l) {
            const int sg = l == 0 ? 0 : 2 * l + 1, sc = sg + 1;
            const int nch = (int)(kx[l] / 16);
            for (int ct = 0; ct < 2 * H / 16; ++ct) {     // gates
                PmReq q = gemm_unit(sg, XG[l], kx[l]);
                PmUnit& u = q.u;
                u.W = d.Wg_t[l] + (size_t)ct * nch * 256;
                u.bias = d.bg[l] ? d.bg[l] + 16 * ct : nullptr;
                if (d.seq_g[l]) u.add[0] = rm(d.seq_g[l] + 16 * ct, 0, 2 * H);
                u.epi = PM_EPI_GATES;
                u.rtile = 16 * ct >= H;
                if (!u.rtile) {
                    u.o1 = rm(zh[l] + 16 * ct, BH, H);
                } else {
                    const int j0 = 16 * ct - H;
                    u.e0 = rm(hist_h[l] + j0, BH, H);
                    u.dst[u.ndst++] = mkdst(XC[l], 0, kx[l], j0 / 16);
                }
                reqs.push_back(q);
            }
            for (int ct = 0; ct < H / 16; ++ct) {         // candidate -> h_l[t+1]
                PmReq q = gemm_unit(sc, XC[l], kx[l]);
                PmUnit& u = q.u;
                u.W = d.Wc_t[l] + (size_t)ct * nch * 256;
                u.bias = d.bc[l] ? d.bc[l] + 16 * ct : nullptr;
                if (d.seq_c[l]) u.add[0] = rm(d.seq_c[l] + 16 * ct, 0, H);
                u.epi = PM_EPI_CAND;
                u.e0 = rm(hist_h[l] + 16 * ct, BH, H);
                u.e1 = rm(zh[l] + 16 * ct, BH, H);
                u.out = rm(hist_h[l] + BH + 16 * ct, BH, H);
                u.dst[u.ndst++] = mkdst(XG[l], 1, kx[l], ct);
                for (int m2 = l + 1; m2 < L; ++m2) {
                    const int ch = (H + E) / 16 + l * (H / 16) + ct;
                    u.dst[u.ndst++] = mkdst(XG[m2], 0, kx[m2], ch);
                    u.dst[u.ndst++] = mkdst(XC[m2], 0, kx[m2], ch);
                }
                u.dst[u.ndst++] = mkdst(XR, 0, kr, l * (H / 16) + ct);
                if (u.ndst > PM_MAXDST) return 0;
                reqs.push_back(q);
            }
        }
        for (int b = 0; b < B; ++b) {
            PmReq q;
            memset(&q, 0, sizeof(q));
            q.u.kind = PM_ATT; q.u.row = b; q.u.w_lds = -1;
            q.slot = 2; q.crit = 1; q.krows = 0;
            reqs.push_back(q);
        }
        for (int ct = 0; ct < R / 16; ++ct) {             // readout
            PmReq q = gemm_unit(2 * L + 1, XR, kr);
            PmUnit& u = q.u;
            u.W = d.Wr_t + (size_t)ct * (kr / 16) * 256;
            u.bias = d.br ? d.br + 16 * ct : nullptr;
            if (d.radd) u.add[0] = rm(d.radd + 16 * ct, 0, R);
            u.epi = PM_EPI_LINEAR;
            u.out = rm(ro_hist + 16 * ct, (long long)B * R, R);
            u.dst[u.ndst++] = mkdst(XO, 0, R, ct);
            reqs.push_back(q);
        }
        for (int ct = 0; ct < 4; ++ct) {                  // output frame x[t+1] (63 columns, padded to 64)
            PmReq q = gemm_unit(2 * L + 2, XO, R);
            PmUnit& u = q.u;
            u.W = d.Wo_t + (size_t)ct * (R / 16) * 256;
            u.bias = d.bo_pad + 16 * ct;
            if (d.oadd_pad) u.add[0] = rm(d.oadd_pad + 16 * ct, 0, 64);
            u.epi = PM_EPI_LINEAR;
            u.out = rm(d.x + (size_t)B * d.ldx + 16 * ct, (long long)B * d.ldx, d.ldx);
            for (int l = 0; l < L; ++l) {
                if (!fb_rows(d, l)) continue;
                const int ch = (int)((kx[l] - 64) / 16) + ct;
                u.dst[u.ndst++] = mkdst(XG[l], 1, kx[l], ch);
                u.dst[u.ndst++] = mkdst(XC[l], 1, kx[l], ch);
            }
            if (u.ndst > PM_MAXDST) return 0;
            reqs.push_back(q);
        }
        std::vector<PmUnit> table;
        if (!pm_place(reqs, n_slots, 1, nwg, table)) return 0;
        if (hipMemcpy(units_dev, table.data(), unit_bytes, hipMemcpyHostToDevice) != hipSuccess) return 0;

        PmProgram& P = pm_prog;
        memset(&P, 0, sizeof(P));
        P.T = S; P.n_ticks = S; P.nwg = nwg; P.MB = MB; P.M = B; P.n_slots = n_slots; P.maxu = 1;
        P.units = units_dev; P.sync = sync; P.fm_base = fm_base;
        PmAtt& a = P.att;
        a.h1 = rm(hist_h[0], BH, H);
        a.WattT = d.WattT; a.batt = d.batt; a.ctx = d.ctx;
        a.kappa = d.kappa; a.a = d.a; a.b = b_hist; a.phi = d.phi; a.w = d.w; a.sup = nullptr;
        a.B = B; a.H = H; a.A = d.A; a.U = d.U; a.E = E; a.att_type = d.att_type; a.dense = 0;
        a.eps = d.eps; a.alignment = d.alignment; a.sharpening = d.sharpening; a.timing = d.timing;
        a.wdst[a.nwdst++] = mkdst(XG[0], 1, kx[0], H / 16);
        a.wdst[a.nwdst++] = mkdst(XC[0], 1, kx[0], H / 16);
        for (int l = 1; l < L; ++l) {
            a.wdst[a.nwdst++] = mkdst(XG[l], 0, kx[l], H / 16);
            a.wdst[a.nwdst++] = mkdst(XC[l], 0, kx[l], H / 16);
        }
        a.wdst[a.nwdst++] = mkdst(XR, 0, kr, L * (H / 16));
        if (a.nwdst > PM_MAXWDST) return 0;
        int ni = 0;
        auto add_init = [&](const float* src, int ld, int K, float* slab, long long ks, int chunk) {
            PmInit& in = P.init[ni++];
            in.src = src; in.ld = ld; in.K = K; in.dst_off = boff(slab); in.nch = (int)(ks / 16); in.chunk = chunk; in.pad = 0;
        };
        for (int l = 0; l < L; ++l) add_init(d.h[l], H, H, XG[l], kx[l], 0);   // initial states (slot 0 of the ping-pong)
        add_init(d.w, E, E, XG[0], kx[0], H / 16);
        add_init(d.w, E, E, XC[0], kx[0], H / 16);
        // x[0] = 0 (model.py:834-835): slot 0 of d.x, converted like the other entering states (the slabs start EMPTY in
        // dataflow mode, so "stays at the zero fill" is not enough)
        for (int l = 0; l < L; ++l) {
            if (!fb_rows(d, l)) continue;
            if (ni + 2 > PM_MAXINIT) return 0;
            add_init(d.x, d.ldx, 64, XG[l], kx[l], (int)((kx[l] - 64) / 16));
            add_init(d.x, d.ldx, 64, XC[l], kx[l], (int)((kx[l] - 64) / 16));
        }
        P.ninit = ni;
        {
            const char* e2 = getenv("PARROT_PM_DATAFLOW");
            P.dataflow = e2 ? atoi(e2) : 0;
        }
        auto add_fill = [&](void* q, long long nfloats) {
            if (nfloats > 0) { P.fill[P.nfill].p = q; P.fill[P.nfill].bytes = nfloats * 4; ++P.nfill; }
        };
        add_fill(fm_base, (long long)(fm_end - fm_base));
        for (int l = 0; l < L; ++l) {
            add_fill(hist_h[l] + BH, (long long)S * BH);
            add_fill(zh[l], (long long)S * BH);
        }
        persist_ok = true;
        return 0;
    }
    int persist_status() const { return persist_ok ? pm_status(pm_prog) : 0; }

    int run_persist(hipStream_t st) {
        const size_t BH = (size_t)d.B * d.H;
        for (int l = 0; l < d.L; ++l)  // row-major initial state for the epilogues (r * h_prev, state blend)
            PL_TRY((int)hipMemcpyAsync(hist_h[l], d.h[l], BH * sizeof(float), hipMemcpyDeviceToDevice, st));
        return pm_launch(pm_prog, st);
    }

    // ---- round 4: the decode step cut along K by the AGE of its operands ------------------------------------------
    // With the output fed back (model.py:899-924) a step is one dependency chain x[t] -> G_0 -> C_0 -> attention ->
    // G_1 -> C_1 .. -> readout -> output -> x[t+1], and above every phase walked the whole K of its product (1600 ..
    // 2624 rows at configs[2]) although only ONE operand of each product is new when the phase starts.  Here
    //   * readout and output are one phase: without GMM head and layer norm, x = (XR . Wr + br + radd) . Wo + bo + oadd is
    //     linear in XR (model.py:992-1013), so the caller hands over Wro = Wr . Wo ([L H + E, 64], fragment-major) and
    //     ro_const = (br + radd) . Wo + bo + oadd ([B, 64]); the readout itself is not an output of sample_model;
    //   * every product is cut into pieces along K, one per operand ([h_l ; w ; h_0 .. h_{l-1} ; x] are chunk ranges of
    //     the unit's slab).  The piece whose operand is produced by the phase just before the product's own is the
    //     CRITICAL unit (K = 64 for G_0, E for G_1, H for the candidates and the output); every other piece runs as a
    //     LINEAR unit on workgroups that are idle anyway (the decode loop keeps < 130 of 256 busy per phase), in a phase
    //     between its operand's and the product's, and leaves its [B, N] partial sums row-major and write-through; the
    //     critical unit adds them in its epilogue (PmUnit::add, up to 4).
    // A tick has 2L + 2 phases; main units run step (tick - 1), pieces whose operand dates from the previous step may run
    // in the previous tick (lag 0), so the launch has S + 1 ticks.  check_pieces() replays the table symbolically (every
    // read satisfied by a write of a strictly earlier phase, every buffer element written once) before it is used.
    struct PmPiece {
        int c0, nch, gp;  // chunk range of the slab; position (phase index over two ticks) after which the operand exists
        bool crit;
        int lag, slot, pbuf;
    };
    struct PmGroup {
        int kind, l, slot, N, res;  // kind 0 gates, 1 candidate, 2 output, 3 x_pre (fbc); res = checker resource id of the slab
        int glag;                   // the group's critical unit runs `glag` ticks after the step's other main units
        long long ks;
        std::vector<PmPiece> pc;
    };
    struct PmAccess { int res, dstep, c0, nch; };
    struct PmMeta { int lag, slot; std::vector<PmAccess> rd, wr; };
    enum { RES_XG = 10, RES_XC = 20, RES_XR = 30, RES_H = 40, RES_Z = 50, RES_X = 60, RES_KAPPA = 61, RES_XPRE = 62,
           RES_PART = 100 };
    bool pieces_ok = false;
    bool fbc_on = false;
    // Round 5 ("fbc"): the fed-back frame out of the chain.  x[t+1] = x_pre + h_{L-1}[t+1] . A (A = the last layer's rows of
    // Wr . Wo), so layer 0's next gates need  x_pre . Wfg  (x_pre is complete two phases before h_{L-1}) and
    // h_{L-1} . (A . Wfg)  -- the caller composes A . Wfg / A . Wfc and appends them to layer 0's matrices (Wgx_t / Wcx_t).
    // The output product then feeds nothing inside the loop: it runs beside the next step's gate phase, and a step is
    // 2L + 1 dependent phases (G_0 with K = H critical instead of K = 64, but one phase of ~5 us less).
    static bool fbc_wanted(const ParrotSampleDesc& d) {
        const char* e = getenv("PARROT_PM_FBC");
        if (e && atoi(e) == 0) return false;
        if (d.L < 2 || !d.Wgx_t[0] || !d.Wcx_t[0] || !fb_rows(d, 0)) return false;
        for (int l = 1; l < d.L; ++l)
            if (fb_rows(d, l)) return false;
        return 2 * d.L + 1 <= PM_MAXSLOTS;
    }
    static long long kslab_p(const ParrotSampleDesc& d, int l, bool fbc) { return kslab(d, l) + ((fbc && l == 0) ? d.H : 0); }
    static int n_phases(const ParrotSampleDesc& d, bool fbc) { return fbc ? 2 * d.L + 1 : 2 * d.L + 2; }
    static int slot_pre(const ParrotSampleDesc& d) { return std::max(slotC(d.L - 2), 2) + 1; }  // (fbc) after x_pre's last operand
    int pieces_info[16] = {0};

    static int slotG(int l) { return l == 0 ? 0 : 2 * l + 1; }
    static int slotC(int l) { return slotG(l) + 1; }
    static bool pieces_wanted(const ParrotSampleDesc& d) {
        const char* e = getenv("PARROT_PM_PIECES");
        return d.Wro_t && d.ro_const && !(e && atoi(e) == 0) && 2 * d.L + 2 <= PM_MAXSLOTS;
    }
    static bool piece_groups(const ParrotSampleDesc& d, std::vector<PmGroup>& gs, bool fbc) {
        const int H = d.H, E = d.E, L = d.L, n = n_phases(d, fbc), sATT = 2, sOUT = fbc ? 0 : 2 * L + 1;
        const int hc = H / 16, ec = E / 16;
        auto pos = [&](int delta, int slot) { return (1 + delta) * n + slot; };
        auto piece = [](int c0, int nch, int gp) { PmPiece p; p.c0 = c0; p.nch = nch; p.gp = gp; p.crit = false; p.lag = 1; p.slot = -1; p.pbuf = -1; return p; };
        gs.clear();
        for (int l = 0; l < L; ++l)
            for (int kind = 0; kind < 2; ++kind) {
                PmGroup g;
                g.kind = kind; g.l = l; g.slot = kind == 0 ? slotG(l) : slotC(l); g.N = kind == 0 ? 2 * H : H; g.glag = 0;
                g.ks = kslab_p(d, l, fbc); g.res = (kind == 0 ? RES_XG : RES_XC) + l;
                g.pc.push_back(piece(0, hc, kind == 0 ? pos(-1, slotC(l)) : pos(0, slotG(l))));  // h_l[t] | r * h_l[t]
                g.pc.push_back(piece(hc, ec, l == 0 ? pos(-1, sATT) : pos(0, sATT)));              // w[t] | w[t+1]
                for (int j = 0; j < l; ++j) g.pc.push_back(piece(hc + ec + j * hc, hc, pos(0, slotC(j))));  // h_j[t+1]
                if (fb_rows(d, l) && !fbc) g.pc.push_back(piece((int)(g.ks / 16) - 4, 4, pos(-1, sOUT)));  // x[t]
                if (fb_rows(d, l) && fbc) {  // (l == 0) x_pre of the previous step, and the last layer's state . (A . Wf)
                    g.pc.push_back(piece(hc + ec, 4, pos(-1, slot_pre(d))));
                    g.pc.push_back(piece(hc + ec + 4, hc, pos(-1, slotC(L - 1))));
                }
                gs.push_back(g);
            }
        {
            PmGroup o;
            o.kind = 2; o.l = 0; o.slot = sOUT; o.N = 64; o.ks = (long long)L * H + E; o.res = RES_XR; o.glag = fbc ? 1 : 0;
            if (!fbc) {
                for (int j = 0; j < L; ++j) o.pc.push_back(piece(j * hc, hc, pos(0, slotC(j))));
                o.pc.push_back(piece(L * hc, ec, pos(0, sATT)));
            } else {  // x = x_pre (added in the epilogue) + h_{L-1} . A, beside the NEXT step's gate phase
                o.pc.push_back(piece((L - 1) * hc, hc, pos(0, slotC(L - 1))));
            }
            gs.push_back(o);
        }
        if (fbc) {
            PmGroup o;
            o.kind = 3; o.l = 0; o.slot = slot_pre(d); o.N = 64; o.ks = (long long)L * H + E; o.res = RES_XR; o.glag = 0;
            for (int j = 0; j + 1 < L; ++j) o.pc.push_back(piece(j * hc, hc, pos(0, slotC(j))));
            o.pc.push_back(piece(L * hc, ec, pos(0, sATT)));
            gs.push_back(o);
        }
        for (PmGroup& g : gs) {
            size_t ci = 0;
            for (size_t i = 1; i < g.pc.size(); ++i)
                if (g.pc[i].gp > g.pc[ci].gp) ci = i;
            if (g.pc[ci].gp >= (1 + g.glag) * n + g.slot) return false;
            g.pc[ci].crit = true;
            g.pc[ci].slot = g.slot;
            g.pc[ci].lag = 1 + g.glag;
            const int fixed = g.kind >= 2 ? 1 : ((g.kind == 0 ? d.seq_g[g.l] : d.seq_c[g.l]) ? 1 : 0);
            while ((int)g.pc.size() - 1 + fixed > 4) {  // more partial sums than a unit can add: join neighbours
                int best = -1, bestd = 1 << 30;
                for (size_t i = 0; i + 1 < g.pc.size(); ++i) {
                    const PmPiece &a = g.pc[i], &b = g.pc[i + 1];
                    if (a.crit || b.crit || a.c0 + a.nch != b.c0) continue;
                    const int dd = a.gp > b.gp ? a.gp - b.gp : b.gp - a.gp;
                    if (dd < bestd) { bestd = dd; best = (int)i; }
                }
                if (best < 0) return false;
                g.pc[best].nch += g.pc[best + 1].nch;
                g.pc[best].gp = std::max(g.pc[best].gp, g.pc[best + 1].gp);
                g.pc.erase(g.pc.begin() + best + 1);
            }
        }
        return true;
    }
    // the phase of every non-critical piece: most constrained first; a phase may not take more units than workgroups, and
    // a piece should not outlast the critical units of its phase (unit cost as in pm_place)
    static bool piece_slots(const ParrotSampleDesc& d, std::vector<PmGroup>& gs, int nwg, bool fbc) {
        const int n = n_phases(d, fbc);
        auto cost = [](int K) { return 3.5 + 3.5 * K / 1024.0; };
        std::vector<int> cnt(n, 0);
        std::vector<double> tcrit(n, 0.0);
        cnt[2] = d.B; tcrit[2] = 9.0;
        struct Ref { int g, p, ncand, K, tiles; };
        std::vector<Ref> refs;
        for (size_t gi = 0; gi < gs.size(); ++gi)
            for (size_t pi = 0; pi < gs[gi].pc.size(); ++pi) {
                PmGroup& g = gs[gi];
                PmPiece& p = g.pc[pi];
                if (p.crit) {
                    cnt[g.slot] += g.N / 16;
                    tcrit[g.slot] = std::max(tcrit[g.slot], cost(p.nch * 16));
                } else {
                    refs.push_back({(int)gi, (int)pi, (1 + g.glag) * n + g.slot - 1 - p.gp, p.nch * 16, g.N / 16});
                }
            }
        for (int s = 0; s < n; ++s)
            if (cnt[s] > nwg) return false;
        std::stable_sort(refs.begin(), refs.end(), [](const Ref& a, const Ref& b) {
            if (a.ncand != b.ncand) return a.ncand < b.ncand;
            if (a.K != b.K) return a.K > b.K;
            return a.tiles > b.tiles;
        });
        for (const Ref& r : refs) {
            PmGroup& g = gs[r.g];
            PmPiece& p = g.pc[r.p];
            int best = -1;
            double best_score = 0;
            for (int q = p.gp + 1; q < (1 + g.glag) * n + g.slot; ++q) {
                const int s = q % n;
                if (cnt[s] + r.tiles > nwg) continue;
                const double late = cost(r.K) - tcrit[s];
                const double score = (late > 0 ? late : 0) * 1e3 + cnt[s] + r.tiles;
                if (best < 0 || score < best_score) { best = q; best_score = score; }
            }
            if (best < 0) return false;
            p.slot = best % n;
            p.lag = best / n;
            cnt[p.slot] += r.tiles;
        }
        return true;
    }
    // Joins the two neighbouring non-critical pieces (same product, adjacent chunk ranges) whose operands appear closest in
    // time -- ties: the widest product first, it frees the most units -- into one piece that waits for the later operand.
    // The plan with the fed-back frame out of the chain has one phase less to spread its pieces over (round 5).
    static bool join_closest_pieces(std::vector<PmGroup>& gs) {
        int bg = -1, bi = -1, bd = 1 << 30, bn = 0;
        for (size_t gi = 0; gi < gs.size(); ++gi) {
            const PmGroup& g = gs[gi];
            for (size_t i = 0; i + 1 < g.pc.size(); ++i) {
                const PmPiece &a = g.pc[i], &b = g.pc[i + 1];
                if (a.crit || b.crit || a.c0 + a.nch != b.c0) continue;
                const int dd = a.gp > b.gp ? a.gp - b.gp : b.gp - a.gp;
                if (dd < bd || (dd == bd && g.N > bn)) { bd = dd; bn = g.N; bg = (int)gi; bi = (int)i; }
            }
        }
        if (bg < 0) return false;
        PmGroup& g = gs[bg];
        g.pc[bi].nch += g.pc[bi + 1].nch;
        g.pc[bi].gp = std::max(g.pc[bi].gp, g.pc[bi + 1].gp);
        g.pc.erase(g.pc.begin() + bi + 1);
        return true;
    }
    static long long piece_floats(const ParrotSampleDesc& d) {  // the partial-sum buffers of the pieces (an upper bound:
        std::vector<PmGroup> gs;                                 // before any capacity-driven joins)
        const bool fbc = fbc_wanted(d);
        if (!pieces_wanted(d) || !piece_groups(d, gs, fbc)) return 0;
        long long n = 0;
        for (const PmGroup& g : gs)
            for (const PmPiece& p : g.pc)
                if (!p.crit) n += (long long)(d.S + 1) * d.B * g.N + 16;
        if (fbc) {  // the longer layer-0 slabs, x_pre row-major, the zero rows behind x[0]
            const long long rows = d.B <= 16 ? 16 : (d.B <= 32 ? 32 : 64);
            n += 2 * (long long)(d.S + 1) * rows * d.H + (long long)(d.S + 2) * d.B * 64 + (long long)d.B * d.H + 64;
        }
        return n;
    }
    // symbolic replay over S steps: 0 = every read finds its value written in an earlier phase and nothing is written twice
    static int check_pieces(const std::vector<PmMeta>& metas, const std::vector<PmAccess>& init, int n_slots, int S,
                            int n_ticks) {
        std::vector<std::array<int, 3>> written;  // (res, step, chunk), kept sorted
        auto has = [&](int r, int t, int c) {
            const std::array<int, 3> k = {r, t, c};
            return std::binary_search(written.begin(), written.end(), k);
        };
        auto put = [&](int r, int t, int c) {
            const std::array<int, 3> k = {r, t, c};
            auto it = std::lower_bound(written.begin(), written.end(), k);
            if (it != written.end() && *it == k) return false;
            written.insert(it, k);
            return true;
        };
        for (const PmAccess& a : init)
            for (int c = a.c0; c < a.c0 + a.nch; ++c)
                if (!put(a.res, a.dstep, c)) return 1;
        for (int tick = 0; tick < n_ticks; ++tick)
            for (int s = 0; s < n_slots; ++s) {
                for (const PmMeta& m : metas) {
                    const int t = tick - m.lag;
                    if (m.slot != s || t < 0 || t >= S) continue;
                    for (const PmAccess& a : m.rd)
                        for (int c = a.c0; c < a.c0 + a.nch; ++c)
                            if (!has(a.res, t + a.dstep, c)) return 2;
                }
                for (const PmMeta& m : metas) {
                    const int t = tick - m.lag;
                    if (m.slot != s || t < 0 || t >= S) continue;
                    for (const PmAccess& a : m.wr)
                        for (int c = a.c0; c < a.c0 + a.nch; ++c)
                            if (!put(a.res, t + a.dstep, c)) return 3;
                }
            }
        for (int t = 1; t <= S; ++t)
            if (!has(RES_X, t, 0)) return 4;
        return 0;
    }

    // dry = true: plan, place and check only (no device memory is touched; nwg given by the caller) -- the CPU tests
    int build_persist_pieces(bool dry, int nwg_dry) {
        pieces_ok = false;
        if (!pieces_wanted(d) || !persist_eligible_shape(d)) return 0;
        const int nwg = dry ? nwg_dry : pm_max_workgroups();
        if (nwg < 64) return 0;
        if (!dry && (!d.persist_ws || d.persist_ws_floats < persist_floats(d, nwg))) return 0;
        std::vector<PmGroup> gs;
        const bool fbc = fbc_wanted(d);
        fbc_on = false;
        if (!piece_groups(d, gs, fbc)) return 0;
        while (!piece_slots(d, gs, nwg, fbc))   // more units than one per workgroup and phase: join two pieces and try again
            if (!join_closest_pieces(gs)) return 0;
        const int H = d.H, E = d.E, B = d.B, L = d.L, S = d.S;
        const int MB = B <= 16 ? 1 : (B <= 32 ? 2 : 4);
        const long long rows = (long long)MB * 16, BH = (long long)B * H;
        const int n_slots = n_phases(d, fbc), sATT = 2, hc = H / 16, ec = E / 16;
        const int fbx = hc + ec, fbh = hc + ec + 4;  // (fbc) layer 0's x_pre chunks / the last layer's state chunks
        float* ws = dry ? reinterpret_cast<float*>((uintptr_t)0x10000000) : d.persist_ws;
        auto take = [&](long long n) { float* p = ws; ws += (n + 3) / 4 * 4; return p; };
        unsigned* sync = reinterpret_cast<unsigned*>(take(PM_SYNC_WORDS + PM_DBG_WORDS));
        const size_t unit_bytes = (size_t)n_slots * nwg * sizeof(PmUnit);
        PmUnit* units_dev = reinterpret_cast<PmUnit*>(take((long long)(unit_bytes + 3) / 4 + 16));
        float* fm_base = ws;
        float* XG[PARROT_MAX_LAYERS];
        float* XC[PARROT_MAX_LAYERS];
        long long kx[PARROT_MAX_LAYERS];
        for (int l = 0; l < L; ++l) {
            kx[l] = kslab_p(d, l, fbc);
            XG[l] = take((S + 1) * rows * kx[l]);
            XC[l] = take((S + 1) * rows * kx[l]);
        }
        const long long kr = (long long)L * H + E;
        float* XR = take(S * rows * kr);
        if ((long long)(ws - fm_base) * 4 >= 0xfff00000ll) return 0;
        float* const fm_end = ws;
        float* zh[PARROT_MAX_LAYERS];
        for (int l = 0; l < L; ++l) hist_h[l] = take((S + 1) * BH);
        for (int l = 0; l < L; ++l) zh[l] = take(S * BH);
        float* b_hist = take((long long)S * B * d.A);
        float* zero_rows = fbc ? take(BH) : nullptr;               // never written: the workspace arrives zero-filled
        float* xpre_rm = fbc ? take((long long)(S + 1) * B * 64) : nullptr;  // x_pre of step t, row-major (the output unit adds it)
        float* const part_base = ws;
        int npart = 0;
        std::vector<float*> pbuf;
        for (PmGroup& g : gs)
            for (PmPiece& p : g.pc)
                if (!p.crit) {
                    p.pbuf = npart++;
                    pbuf.push_back(take((long long)(S + 1) * B * g.N + 16));
                }
        float* const part_end = ws;

        auto boff = [&](const float* p) { return (unsigned)((p - fm_base) * 4); };
        auto mkdst = [&](float* slab, long long step0, long long ks, int chunk) {
            PmDst q;
            q.off = boff(slab + step0 * rows * ks); q.st = (unsigned)(rows * ks * 4); q.nch = (int)(ks / 16); q.chunk = chunk;
            return q;
        };
        auto rm = [](const float* p, long long st, int ld) { PmRM r; r.p = const_cast<float*>(p); r.st = st; r.ld = ld; r.pad = 0; return r; };
        std::vector<PmReq> reqs;
        std::vector<PmMeta> metas;
        auto acc = [](int res, int dstep, int c0, int nch) { PmAccess a; a.res = res; a.dstep = dstep; a.c0 = c0; a.nch = nch; return a; };
        for (const PmGroup& g : gs) {
            const int l = g.l, N = g.N, nch_all = (int)(g.ks / 16);
            float* slab = g.kind == 0 ? XG[l] : (g.kind == 1 ? XC[l] : XR);
            const float* Wt = g.kind == 0 ? ((fbc && l == 0) ? d.Wgx_t[0] : d.Wg_t[l])
                                          : (g.kind == 1 ? ((fbc && l == 0) ? d.Wcx_t[0] : d.Wc_t[l]) : d.Wro_t);
            for (const PmPiece& p : g.pc) {
                PmMeta m;
                m.lag = p.lag; m.slot = p.slot;
                m.rd.push_back(acc(g.res, 0, p.c0, p.nch));
                if (!p.crit) {
                    m.wr.push_back(acc(RES_PART + p.pbuf, 0, 0, 1));
                } else {
                    for (const PmPiece& o : g.pc)
                        if (!o.crit) m.rd.push_back(acc(RES_PART + o.pbuf, 0, 0, 1));
                    if (g.kind == 0) {
                        m.rd.push_back(acc(RES_H + l, 0, 0, 1));
                        m.wr.push_back(acc(RES_Z + l, 0, 0, 1));
                        m.wr.push_back(acc(RES_XC + l, 0, 0, hc));
                    } else if (g.kind == 1) {
                        m.rd.push_back(acc(RES_H + l, 0, 0, 1));
                        m.rd.push_back(acc(RES_Z + l, 0, 0, 1));
                        m.wr.push_back(acc(RES_H + l, 1, 0, 1));
                        m.wr.push_back(acc(RES_XG + l, 1, 0, hc));
                        for (int m2 = l + 1; m2 < L; ++m2) {
                            m.wr.push_back(acc(RES_XG + m2, 0, hc + ec + l * hc, hc));
                            m.wr.push_back(acc(RES_XC + m2, 0, hc + ec + l * hc, hc));
                        }
                        m.wr.push_back(acc(RES_XR, 0, l * hc, hc));
                        if (fbc && l == L - 1) {
                            m.wr.push_back(acc(RES_XG, 1, fbh, hc));
                            m.wr.push_back(acc(RES_XC, 1, fbh, hc));
                        }
                    } else if (g.kind == 2) {
                        m.wr.push_back(acc(RES_X, 1, 0, 1));
                        if (fbc) m.rd.push_back(acc(RES_XPRE, 0, 0, 1));
                        for (int q = 0; q < L && !fbc; ++q)
                            if (fb_rows(d, q)) {
                                m.wr.push_back(acc(RES_XG + q, 1, (int)(kx[q] / 16) - 4, 4));
                                m.wr.push_back(acc(RES_XC + q, 1, (int)(kx[q] / 16) - 4, 4));
                            }
                    } else {  // x_pre
                        m.wr.push_back(acc(RES_XPRE, 0, 0, 1));
                        m.wr.push_back(acc(RES_XG, 1, fbx, 4));
                        m.wr.push_back(acc(RES_XC, 1, fbx, 4));
                    }
                }
                metas.push_back(m);
                for (int ct = 0; ct < N / 16; ++ct) {
                    PmReq q;
                    memset(&q, 0, sizeof(q));
                    PmUnit& u = q.u;
                    u.kind = PM_GEMM; u.M = B; u.w_lds = -1; u.lag = p.lag;
                    u.a_off = boff(slab); u.a_st = (unsigned)(rows * g.ks * 4); u.a_nch = nch_all; u.a_c0 = p.c0; u.K = p.nch * 16;
                    u.W = Wt + ((size_t)ct * nch_all + p.c0) * 256;
                    q.slot = p.slot; q.crit = p.crit ? 1 : 0; q.krows = p.nch * 16;
                    if (!p.crit) {
                        u.epi = PM_EPI_LINEAR;
                        u.out = rm(pbuf[p.pbuf] + 16 * ct, (long long)B * N, N);
                        reqs.push_back(q);
                        continue;
                    }
                    int na = 0;
                    for (const PmPiece& o : g.pc)
                        if (!o.crit) u.add[na++] = rm(pbuf[o.pbuf] + 16 * ct, (long long)B * N, N);
                    if (g.kind == 0) {
                        u.bias = d.bg[l] ? d.bg[l] + 16 * ct : nullptr;
                        if (d.seq_g[l]) u.add[na++] = rm(d.seq_g[l] + 16 * ct, 0, 2 * H);
                        u.epi = PM_EPI_GATES;
                        u.rtile = 16 * ct >= H;
                        if (!u.rtile) {
                            u.o1 = rm(zh[l] + 16 * ct, BH, H);
                        } else {
                            const int j0 = 16 * ct - H;
                            u.e0 = rm(hist_h[l] + j0, BH, H);
                            u.dst[u.ndst++] = mkdst(XC[l], 0, kx[l], j0 / 16);
                        }
                    } else if (g.kind == 1) {
                        u.bias = d.bc[l] ? d.bc[l] + 16 * ct : nullptr;
                        if (d.seq_c[l]) u.add[na++] = rm(d.seq_c[l] + 16 * ct, 0, H);
                        u.epi = PM_EPI_CAND;
                        u.e0 = rm(hist_h[l] + 16 * ct, BH, H);
                        u.e1 = rm(zh[l] + 16 * ct, BH, H);
                        u.out = rm(hist_h[l] + BH + 16 * ct, BH, H);
                        u.dst[u.ndst++] = mkdst(XG[l], 1, kx[l], ct);
                        for (int m2 = l + 1; m2 < L; ++m2) {
                            const int ch = hc + ec + l * hc + ct;
                            u.dst[u.ndst++] = mkdst(XG[m2], 0, kx[m2], ch);
                            u.dst[u.ndst++] = mkdst(XC[m2], 0, kx[m2], ch);
                        }
                        u.dst[u.ndst++] = mkdst(XR, 0, kr, l * hc + ct);
                        if (fbc && l == L - 1) {  // ... and the operand of layer 0's composed feedback rows, next step
                            if (u.ndst + 2 > PM_MAXDST) return 0;
                            u.dst[u.ndst++] = mkdst(XG[0], 1, kx[0], fbh + ct);
                            u.dst[u.ndst++] = mkdst(XC[0], 1, kx[0], fbh + ct);
                        }
                    } else if (g.kind == 3) {  // x_pre = ro_const + the shares of every operand but the last layer's state
                        u.add[na++] = rm(d.ro_const + 16 * ct, 0, 64);
                        u.epi = PM_EPI_LINEAR;
                        u.out = rm(xpre_rm + 16 * ct, (long long)B * 64, 64);
                        u.dst[u.ndst++] = mkdst(XG[0], 1, kx[0], fbx + ct);
                        u.dst[u.ndst++] = mkdst(XC[0], 1, kx[0], fbx + ct);
                    } else {
                        u.add[na++] = fbc ? rm(xpre_rm + 16 * ct, (long long)B * 64, 64) : rm(d.ro_const + 16 * ct, 0, 64);
                        u.epi = PM_EPI_LINEAR;
                        u.out = rm(d.x + (size_t)B * d.ldx + 16 * ct, (long long)B * d.ldx, d.ldx);
                        for (int q2 = 0; q2 < L && !fbc; ++q2) {
                            if (!fb_rows(d, q2)) continue;
                            const int ch = (int)(kx[q2] / 16) - 4 + ct;
                            u.dst[u.ndst++] = mkdst(XG[q2], 1, kx[q2], ch);
                            u.dst[u.ndst++] = mkdst(XC[q2], 1, kx[q2], ch);
                        }
                    }
                    if (na > 4 || u.ndst > PM_MAXDST) return 0;
                    reqs.push_back(q);
                }
            }
        }
        {
            PmMeta m;
            m.lag = 1; m.slot = sATT;
            m.rd.push_back(acc(RES_H, 1, 0, 1));
            m.rd.push_back(acc(RES_KAPPA, 0, 0, 1));
            m.wr.push_back(acc(RES_KAPPA, 1, 0, 1));
            m.wr.push_back(acc(RES_XG, 1, hc, ec));
            m.wr.push_back(acc(RES_XC, 1, hc, ec));
            for (int l = 1; l < L; ++l) {
                m.wr.push_back(acc(RES_XG + l, 0, hc, ec));
                m.wr.push_back(acc(RES_XC + l, 0, hc, ec));
            }
            m.wr.push_back(acc(RES_XR, 0, L * hc, ec));
            metas.push_back(m);
        }
        for (int b = 0; b < B; ++b) {
            PmReq q;
            memset(&q, 0, sizeof(q));
            q.u.kind = PM_ATT; q.u.row = b; q.u.w_lds = -1; q.u.lag = 1;
            q.slot = sATT; q.crit = 1; q.krows = 0;
            reqs.push_back(q);
        }
        std::vector<PmAccess> init;
        for (int l = 0; l < L; ++l) {
            init.push_back(acc(RES_XG + l, 0, 0, hc));
            init.push_back(acc(RES_H + l, 0, 0, 1));
            if (fb_rows(d, l) && !fbc) {
                init.push_back(acc(RES_XG + l, 0, (int)(kx[l] / 16) - 4, 4));
                init.push_back(acc(RES_XC + l, 0, (int)(kx[l] / 16) - 4, 4));
            }
        }
        if (fbc) {  // step 0: x[0] in the x_pre chunks, zero rows where the last layer's state would go
            init.push_back(acc(RES_XG, 0, fbx, 4 + hc));
            init.push_back(acc(RES_XC, 0, fbx, 4 + hc));
        }
        init.push_back(acc(RES_XG, 0, hc, ec));
        init.push_back(acc(RES_XC, 0, hc, ec));
        init.push_back(acc(RES_KAPPA, 0, 0, 1));
        if (getenv("PARROT_PM_DUMP_PLAN"))
            for (const PmGroup& g : gs)
                for (const PmPiece& p : g.pc)
                    fprintf(stderr, "[pieces] %s%d phase %d: chunks %d..%d (K %d) %s phase %d lag %d\n",
                            g.kind == 0 ? "G" : (g.kind == 1 ? "C" : (g.kind == 2 ? "OUT" : "XPRE")), g.l, g.slot, p.c0, p.c0 + p.nch, p.nch * 16,
                            p.crit ? "CRITICAL" : "piece", p.slot, p.lag);
        const int n_extra = fbc ? 2 : 1;  // ticks beyond S: main units lag one tick, the output unit of the fbc plan two
        const int chk = check_pieces(metas, init, n_slots, 4, 4 + n_extra);
        memset(pieces_info, 0, sizeof(pieces_info));
        pieces_info[0] = n_slots; pieces_info[1] = npart; pieces_info[2] = chk; pieces_info[3] = (int)reqs.size();
        pieces_info[15] = fbc ? 1 : 0;
        for (const PmReq& q : reqs) pieces_info[4 + q.slot] += 1;
        if (chk != 0) return 0;
        std::vector<PmUnit> table;
        if (!pm_place(reqs, n_slots, 1, nwg, table)) return 0;
        for (const PmUnit& u : table)
            if (u.kind == PM_GEMM && u.w_lds < 0) pieces_info[14] += 1;  // units that stream their weights
        pieces_ok = true;
        fbc_on = fbc;
        if (dry) return 0;
        if (hipMemcpy(units_dev, table.data(), unit_bytes, hipMemcpyHostToDevice) != hipSuccess) { pieces_ok = false; return 0; }

        PmProgram& P = pm_prog;
        memset(&P, 0, sizeof(P));
        P.T = S; P.n_ticks = S + n_extra; P.nwg = nwg; P.MB = MB; P.M = B; P.n_slots = n_slots; P.maxu = 1;
        P.units = units_dev; P.sync = sync; P.fm_base = fm_base;
        PmAtt& a = P.att;
        a.h1 = rm(hist_h[0], BH, H);
        a.WattT = d.WattT; a.batt = d.batt; a.ctx = d.ctx;
        a.kappa = d.kappa; a.a = d.a; a.b = b_hist; a.phi = d.phi; a.w = d.w; a.sup = nullptr;
        a.B = B; a.H = H; a.A = d.A; a.U = d.U; a.E = E; a.att_type = d.att_type; a.dense = 0;
        a.eps = d.eps; a.alignment = d.alignment; a.sharpening = d.sharpening; a.timing = d.timing;
        a.wdst[a.nwdst++] = mkdst(XG[0], 1, kx[0], hc);
        a.wdst[a.nwdst++] = mkdst(XC[0], 1, kx[0], hc);
        for (int l = 1; l < L; ++l) {
            a.wdst[a.nwdst++] = mkdst(XG[l], 0, kx[l], hc);
            a.wdst[a.nwdst++] = mkdst(XC[l], 0, kx[l], hc);
        }
        a.wdst[a.nwdst++] = mkdst(XR, 0, kr, L * hc);
        if (a.nwdst > PM_MAXWDST) { pieces_ok = false; return 0; }
        int ni = 0;
        auto add_init = [&](const float* src, int ld, int K, float* slabp, long long ks, int chunk) {
            PmInit& in = P.init[ni++];
            in.src = src; in.ld = ld; in.K = K; in.dst_off = boff(slabp); in.nch = (int)(ks / 16); in.chunk = chunk; in.pad = 0;
        };
        for (int l = 0; l < L; ++l) add_init(d.h[l], H, H, XG[l], kx[l], 0);
        add_init(d.w, E, E, XG[0], kx[0], hc);
        add_init(d.w, E, E, XC[0], kx[0], hc);
        for (int l = 0; l < L && !fbc; ++l) {
            if (!fb_rows(d, l)) continue;
            if (ni + 2 > PM_MAXINIT) { pieces_ok = false; return 0; }
            add_init(d.x, d.ldx, 64, XG[l], kx[l], (int)(kx[l] / 16) - 4);
            add_init(d.x, d.ldx, 64, XC[l], kx[l], (int)(kx[l] / 16) - 4);
        }
        if (fbc) {
            if (ni + 4 > PM_MAXINIT) { pieces_ok = false; return 0; }
            add_init(d.x, d.ldx, 64, XG[0], kx[0], fbx);
            add_init(d.x, d.ldx, 64, XC[0], kx[0], fbx);
            add_init(zero_rows, H, H, XG[0], kx[0], fbh);
            add_init(zero_rows, H, H, XC[0], kx[0], fbh);
        }
        P.ninit = ni;
        {
            // no grid barriers by default: with the step cut into pieces a phase is ~3 us of fixed latency + a short K
            // walk, and the barrier was a quarter of it (43.0 -> 36.1 us per step at configs[2]); the whole-K plan above
            // measured no gain (57.6 either way).  Bit-identical to the barrier mode (tests/test_gpu_persist.py)
            const char* e2 = getenv("PARROT_PM_DATAFLOW");
            P.dataflow = e2 ? atoi(e2) : 1;
        }
        auto add_fill = [&](void* q, long long nfloats) {
            if (nfloats > 0) { P.fill[P.nfill].p = q; P.fill[P.nfill].bytes = nfloats * 4; ++P.nfill; }
        };
        add_fill(fm_base, (long long)(fm_end - fm_base));
        for (int l = 0; l < L; ++l) {
            add_fill(hist_h[l] + BH, (long long)S * BH);
            add_fill(zh[l], (long long)S * BH);
        }
        add_fill(part_base, (long long)(part_end - part_base));
        if (fbc) add_fill(xpre_rm, (long long)(S + 1) * B * 64);
        persist_ok = true;
        return 0;
    }


    void layer_segs(SkJob& j, int l, int t, const float* first, const float* W, int ldw, const float* Wf) const {
        const size_t BH = (size_t)d.B * d.H, BE = (size_t)d.B * d.E;
        const int nxt = (t + 1) & 1;
        int n = 0;
        j.seg[n++] = sk_seg(first, d.H, W, ldw, d.H, 0);
        const float* wsrc = d.w + (size_t)(l == 0 ? t : t + 1) * BE;
        j.seg[n++] = sk_seg(wsrc, d.E, W + (size_t)d.H * ldw, ldw, d.E, 0);
        if (!d.layer_norm) {  // with layer_norm these arrive normalised through the additive input instead
            for (int q = 0; q < l; ++q)
                j.seg[n++] = sk_seg(d.h[q] + nxt * BH, d.H, W + (size_t)(d.H + d.E + q * d.H) * ldw, ldw, d.H, 0);
            if (Wf) j.seg[n++] = sk_seg(d.x + (size_t)t * d.B * d.ldx, d.ldx, Wf, ldw, d.O, 0);
        }
        j.nseg = n;
    }

    // ---- layer_norm: scratch layout and the per-step normalised sums --------------------------------
    int ngrp() const { return d.cell == 1 ? 1 : 2; }
    int gw(int g) const { return d.cell == 1 ? 4 * d.H : (g == 0 ? 2 * d.H : d.H); }
    float* tmp(int g, int k) const {  // k = 0..3 projections, k = 4: the summed additive input
        float* p = d.ln_scratch;
        if (g == 1) p += (size_t)5 * d.B * gw(0);
        return p + (size_t)k * d.B * gw(g);
    }
    float* rtmp(int k) const {  // k = 0..L-1 readout projections, k = L: un-normalised base
        size_t off = (size_t)5 * d.B * gw(0) + (ngrp() == 2 ? (size_t)5 * d.B * gw(1) : 0);
        return d.ln_scratch + off + (size_t)k * d.B * d.R;
    }
    long long scratch_need() const {
        return (long long)5 * d.B * gw(0) + (ngrp() == 2 ? (long long)5 * d.B * gw(1) : 0) +
               (long long)(d.L + 1) * d.B * d.R;
    }

    // Additive inputs of layer l at step t = speaker term + norm(feedback Fork) + sum_j norm(h_j Fork).
    int ln_layer_inputs(int l, int t, hipStream_t st, const float*& addg, const float*& addc) const {
        const size_t BH = (size_t)d.B * d.H;
        const int nxt = (t + 1) & 1;
        SkJob jobs[SK_MAXJOB];
        NormSumGroup grp[2];
        int nj = 0;
        for (int g = 0; g < ngrp(); ++g) {
            const int wd = gw(g);
            const float* W = g == 0 ? d.Wg[l] : d.Wc[l];
            const float* Wf = g == 0 ? d.Wfg[l] : d.Wfc[l];
            NormSumGroup& G = grp[g];
            G.nsrc = 0; G.N = wd;
            G.base = g == 0 ? d.seq_g[l] : d.seq_c[l];
            G.dst = tmp(g, 4);
            if (Wf) {
                SkJob& j = jobs[nj++];
                sk_job_init(j);
                j.nseg = 1;
                j.seg[0] = sk_seg(d.x + (size_t)t * d.B * d.ldx, d.ldx, Wf, wd, d.O, 0);
                j.M = d.B; j.N = wd; j.H = d.H; j.epi = SK_EPI_LINEAR;
                j.bias = g == 0 ? d.bfg[l] : d.bfc[l];
                j.out = tmp(g, G.nsrc); j.ldo = wd;
                G.src[G.nsrc++] = j.out;
            }
            for (int q = 0; q < l; ++q) {
                SkJob& j = jobs[nj++];
                sk_job_init(j);
                j.nseg = 1;
                j.seg[0] = sk_seg(d.h[q] + nxt * BH, d.H, W + (size_t)(d.H + d.E + q * d.H) * wd, wd, d.H, 0);
                j.M = d.B; j.N = wd; j.H = d.H; j.epi = SK_EPI_LINEAR;
                const int pj = l * PARROT_MAX_LAYERS + q;
                j.bias = g == 0 ? d.ln_bg[pj] : d.ln_bc[pj];
                j.out = tmp(g, G.nsrc); j.ldo = wd;
                G.src[G.nsrc++] = j.out;
            }
        }
        if (nj == 0) return 0;
        PL_TRY(launch_jobs(jobs, nj, st));
        PL_TRY(norm_sum_launch(grp, ngrp(), d.B, PARROT_NORM_EPS, st));
        addg = tmp(0, 4);
        if (ngrp() == 2) addc = tmp(1, 4);
        return 0;
    }

    // readout = att_to_readout(w) + speaker term + sum_l norm(h{l}_to_readout(h_l))   (model.py:992-1006)
    int ln_readout(int t, hipStream_t st) const {
        const size_t BH = (size_t)d.B * d.H, BE = (size_t)d.B * d.E;
        const int nxt = (t + 1) & 1;
        SkJob jobs[PARROT_MAX_LAYERS + 1];
        NormSumGroup G;
        G.nsrc = 0; G.N = d.R; G.base = rtmp(d.L); G.dst = d.readout;
        for (int l = 0; l < d.L; ++l) {
            SkJob& j = jobs[l];
            sk_job_init(j);
            j.nseg = 1;
            j.seg[0] = sk_seg(d.h[l] + nxt * BH, d.H, d.Wr + (size_t)l * d.H * d.R, d.R, d.H, 0);
            j.M = d.B; j.N = d.R; j.H = d.H; j.epi = SK_EPI_LINEAR;
            j.bias = d.br_l[l];
            j.out = rtmp(l); j.ldo = d.R;
            G.src[G.nsrc++] = j.out;
        }
        SkJob& b = jobs[d.L];
        sk_job_init(b);
        b.nseg = 1;
        b.seg[0] = sk_seg(d.w + (t + 1) * BE, d.E, d.Wr + (size_t)d.L * d.H * d.R, d.R, d.E, 0);
        b.M = d.B; b.N = d.R; b.H = d.H; b.epi = SK_EPI_LINEAR;
        b.bias = d.br; b.add = d.radd; b.ld_add = d.R;
        b.out = rtmp(d.L); b.ldo = d.R;
        PL_TRY(launch_jobs(jobs, d.L + 1, st));
        return norm_sum_launch(&G, 1, d.B, PARROT_NORM_EPS, st);
    }

    int run_all(hipStream_t st) {
        const size_t BH = (size_t)d.B * d.H, BE = (size_t)d.B * d.E, BA = (size_t)d.B * d.A;
        const int H = d.H;
        for (int t = 0; t < d.S; ++t) {
            const int cur = t & 1, nxt = (t + 1) & 1;
            for (int l = 0; l < d.L; ++l) {
                SkJob j;
                const float* addg = d.seq_g[l];
                const float* addc = d.seq_c[l];
                if (d.layer_norm) PL_TRY(ln_layer_inputs(l, t, st, addg, addc));
                if (d.cell == 1) {
                    sk_job_init(j);
                    layer_segs(j, l, t, d.h[l] + cur * BH, d.Wg[l], 4 * H, d.Wfg[l]);
                    j.M = d.B; j.N = 4 * H; j.H = H; j.epi = SK_EPI_LSTM;
                    j.bias = d.bg[l];
                    j.add = addg; j.ld_add = 4 * H;
                    j.e1 = d.cwork[l] + cur * BH; j.lde1 = H;
                    j.o1 = d.cwork[l] + nxt * BH; j.ldo1 = H;
                    j.o2 = d.gwork; j.ldo2 = 4 * H;
                    j.out = d.h[l] + nxt * BH; j.ldo = H;
                    PL_TRY(launch_jobs(&j, 1, st));
                } else {
                sk_job_init(j);
                layer_segs(j, l, t, d.h[l] + cur * BH, d.Wg[l], 2 * H, d.Wfg[l]);
                j.M = d.B; j.N = 2 * H; j.H = H; j.epi = SK_EPI_GRU_GATES;
                j.bias = d.bg[l];
                j.add = addg; j.ld_add = 2 * H;
                j.e0 = d.h[l] + cur * BH; j.lde0 = H;
                j.o1 = d.zwork; j.ldo1 = H;
                j.o2 = d.rwork; j.ldo2 = H;
                j.out = d.rhwork; j.ldo = H;
                PL_TRY(launch_jobs(&j, 1, st));

                sk_job_init(j);
                layer_segs(j, l, t, d.rhwork, d.Wc[l], H, d.Wfc[l]);
                j.M = d.B; j.N = H; j.H = H; j.epi = SK_EPI_GRU_CAND;
                j.bias = d.bc[l];
                j.add = addc; j.ld_add = H;
                j.e0 = d.h[l] + cur * BH; j.lde0 = H;
                j.e1 = d.zwork; j.lde1 = H;
                j.o1 = nullptr;
                j.out = d.h[l] + nxt * BH; j.ldo = H;
                PL_TRY(launch_jobs(&j, 1, st));
                }

                if (l == 0) {
                    AttFwdArgs g{};
                    g.h1 = d.h[0] + nxt * BH; g.ldh = H;
                    g.WattT = d.WattT; g.batt = d.batt;
                    g.kappa_prev = d.kappa + t * BA;
                    g.ctx = d.ctx;
                    g.a_out = d.a + t * BA; g.b_out = d.bwork; g.kappa_out = d.kappa + (t + 1) * BA;
                    g.phi_out = d.phi + (size_t)t * d.B * d.U;
                    g.w_out = d.w + (t + 1) * BE; g.ldw = d.E;
                    g.B = d.B; g.H = H; g.A = d.A; g.U = d.U; g.E = d.E; g.esplit = esplit;
                    g.att_type = d.att_type; g.eps = d.eps; g.alignment = d.alignment;
                    g.sharpening = d.sharpening; g.timing = d.timing;
                    PL_TRY(att_fwd_launch(g, st));
                }
            }
            // readouts (model.py:992-1006) and output (model.py:1008-1013)
            SkJob j;
            if (d.layer_norm) {
                PL_TRY(ln_readout(t, st));
            } else {
                sk_job_init(j);
                int n = 0;
                for (int l = 0; l < d.L; ++l)
                    j.seg[n++] = sk_seg(d.h[l] + nxt * BH, H, d.Wr + (size_t)l * H * d.R, d.R, H, 0);
                j.seg[n++] = sk_seg(d.w + (t + 1) * BE, d.E, d.Wr + (size_t)d.L * H * d.R, d.R, d.E, 0);
                j.nseg = n;
                j.M = d.B; j.N = d.R; j.H = H; j.epi = SK_EPI_LINEAR;
                j.bias = d.br; j.add = d.radd; j.ld_add = d.R;
                j.out = d.readout; j.ldo = d.R;
                PL_TRY(launch_jobs(&j, 1, st));
            }

            if (d.gmm_K > 0) {
                // GMM head: three projections of the readout in one launch, then the sampling kernel.
                const int OK = d.O * d.gmm_K;
                SkJob g3[3];
                const float* Ws[3] = {d.Wmu, d.Wsig, d.Wco};
                const float* bs[3] = {d.bmu, d.bsig, d.bco};
                const float* as[3] = {d.add_mu, d.add_sig, d.add_co};
                float* os[3] = {d.gmm_mu, d.gmm_sig, d.gmm_co};
                const int ns[3] = {OK, OK, d.gmm_K};
                for (int q = 0; q < 3; ++q) {
                    sk_job_init(g3[q]);
                    g3[q].nseg = 1;
                    g3[q].seg[0] = sk_seg(d.readout, d.R, Ws[q], ns[q], d.R, 0);
                    g3[q].M = d.B; g3[q].N = ns[q]; g3[q].H = H; g3[q].epi = SK_EPI_LINEAR;
                    g3[q].bias = bs[q]; g3[q].add = as[q]; g3[q].ld_add = ns[q];
                    g3[q].out = os[q]; g3[q].ldo = ns[q];
                }
                PL_TRY(launch_jobs(g3, 3, st));
                PL_TRY(gmm_sample_launch(d.gmm_mu, d.gmm_sig, d.gmm_co, d.B, d.O, d.gmm_K, d.sampling_bias, d.eps,
                                         d.unif + (size_t)t * d.B, d.noise + (size_t)t * d.B * d.O,
                                         d.x + (size_t)(t + 1) * d.B * d.ldx, d.ldx,
                                         d.pi_out ? d.pi_out + (size_t)t * d.B * d.gmm_K : nullptr, st));
                continue;
            }
            sk_job_init(j);
            j.nseg = 1;
            j.seg[0] = sk_seg(d.readout, d.R, d.Wo, d.O, d.R, 0);
            j.M = d.B; j.N = d.O; j.H = H; j.epi = SK_EPI_LINEAR;
            j.bias = d.bo; j.add = d.oadd; j.ld_add = d.O;
            j.out = d.x + (size_t)(t + 1) * d.B * d.ldx; j.ldo = d.ldx;
            PL_TRY(launch_jobs(&j, 1, st));
        }
        return 0;
    }
};

bool bad_dims(int L) { return L < 1 || L > PARROT_MAX_LAYERS; }

}  // namespace

extern "C" {

int parrot_gru_seq_create(const ParrotGruSeqDesc* desc, void** plan) { PH_ENTRY();
    if (!desc || !plan || desc->T < 1 || desc->B < 1 || desc->H < 1 || desc->nchain < 1 || desc->nchain > 4)
        return PARROT_ERR_BADARG;
    GruSeqPlan* p = new (std::nothrow) GruSeqPlan();
    if (!p) return PARROT_ERR_BADARG;
    p->d = *desc;
    p->use_graph = desc->use_graph;
    p->setup_rowwise();
    *plan = p;
    return 0;
}
int parrot_gru_seq_fwd(void* plan, void* stream) { PH_ENTRY(); return static_cast<PlanBase*>(plan)->run(0, (hipStream_t)stream); }
int parrot_gru_seq_bwd(void* plan, void* stream) { PH_ENTRY(); return static_cast<PlanBase*>(plan)->run(1, (hipStream_t)stream); }
int parrot_gru_seq_destroy(void* plan) { PH_ENTRY();
    delete static_cast<PlanBase*>(plan);
    return 0;
}

int parrot_lstm_seq_create(const ParrotLstmSeqDesc* desc, void** plan) { PH_ENTRY();
    if (!desc || !plan || desc->T < 1 || desc->B < 1 || desc->H < 4 || (desc->H & 3)) return PARROT_ERR_BADARG;
    LstmSeqPlan* p = new (std::nothrow) LstmSeqPlan();
    if (!p) return PARROT_ERR_BADARG;
    p->d = *desc;
    p->use_graph = desc->use_graph;
    *plan = p;
    return 0;
}
int parrot_lstm_seq_fwd(void* plan, void* stream) { PH_ENTRY(); return static_cast<PlanBase*>(plan)->run(0, (hipStream_t)stream); }
int parrot_lstm_seq_bwd(void* plan, void* stream) { PH_ENTRY(); return static_cast<PlanBase*>(plan)->run(1, (hipStream_t)stream); }
int parrot_lstm_seq_destroy(void* plan) { PH_ENTRY();
    delete static_cast<PlanBase*>(plan);
    return 0;
}

int parrot_decoder_create(const ParrotDecoderDesc* desc, void** plan) { PH_ENTRY();
    if (!desc || !plan || desc->T < 1 || desc->B < 1 || bad_dims(desc->L)) return PARROT_ERR_BADARG;
    DecoderPlan* p = new (std::nothrow) DecoderPlan();
    if (!p) return PARROT_ERR_BADARG;
    p->d = *desc;
    p->use_graph = desc->use_graph;
    p->esplit = att_default_esplit(desc->B, desc->E);
    p->choose_schedule();
    {
        bool all = (desc->H % 16 == 0) && (desc->E % 16 == 0);
        for (int l = 0; l < desc->L; ++l) {
            if (!desc->Wg_f[l] || !desc->Wg_r[l]) all = false;
            if (desc->cell == 0 && (!desc->Wc_f[l] || !desc->Wc_r[l])) all = false;
        }
        const char* e = getenv("PARROT_TILED_WEIGHTS");
        p->tiled = all && !(e && atoi(e) == 0);
        if (desc->bf16) {  // bf16 operands exist only as fragment-major copies; 32-deep K chunks
            if (!all || desc->layer_norm || (desc->H % 32) || (desc->E % 32)) {
                delete p;
                return PARROT_ERR_BADARG;
            }
            p->tiled = true;
        }
    }
    if (p->schedule == 7) {
        // bf16 operands: only the wide step kernel takes a launch with a waiting job (skinny.hip wk_try_launch): every
        // launch of the scan must qualify, the first tick's (layer 0 alone) included.  f32 operands run on ska_kernel.
        const bool ok = p->tiled && desc->B <= 64 && (desc->bf16 ? sk_wide_takes(desc->B, 4 * desc->H, desc->H, desc->E)
                                                                 : getenv("PARROT_SCHEDULE") != nullptr);
        if (!ok) p->schedule = 0;
    }
    if (p->schedule == 5 && desc->L < 2) p->schedule = 0;
    {
        const char* e = getenv("PARROT_BWD_SPLIT");
        p->bwd_split = e ? atoi(e) != 0 : true;
    }
    if (p->schedule == 5) {
        const char* e = getenv("PARROT_S5_ESPLIT");
        p->esplit5 = e && atoi(e) > 0 ? atoi(e) : 1;
        e = getenv("PARROT_S5_SPLIT");
        p->s5_split = e ? atoi(e) != 0 : true;
    }
    if (p->schedule == 7) {
        if (hipMalloc(&p->att_flags, sizeof(unsigned) * (size_t)(desc->T + 2)) != hipSuccess) {
            if (!getenv("PARROT_TRACE_ONLY")) {  // (schedule tracing on a box without a GPU: a placeholder address)
                delete p;
                return PARROT_ERR_BADARG;
            }
            (void)hipGetLastError();
            p->att_flags = reinterpret_cast<unsigned*>((uintptr_t)0x1000);
            p->flags_fake = true;
        }
        if (p->schedule == 7 && desc->bf16 && !(getenv("PARROT_BWD_FUSED") && atoi(getenv("PARROT_BWD_FUSED")) == 0)) {
            // the backward tick as one launch too (wkb_kernel); PARROT_BWD_FUSED=0 keeps the two launches of schedule 0
            const size_t words = (size_t)4 * (desc->T + desc->L);
            if (p->flags_fake) p->bwd_flags = reinterpret_cast<unsigned*>((uintptr_t)0x100000);
            else if (hipMalloc(&p->bwd_flags, sizeof(unsigned) * words) != hipSuccess) { delete p; return PARROT_ERR_BADARG; }
            p->bwd_fused = true;
        }
    }
    if (desc->cell == 1 && desc->bf16 && p->tiled && !(getenv("PARROT_BWD_KSPLIT") && atoi(getenv("PARROT_BWD_KSPLIT")) == 0)) {
        bool have = desc->dw_b && desc->dw0_b && desc->B <= 64 && sk_wide_takes(desc->B, 4096, desc->H, desc->E) &&
                    (4 * desc->H) % 128 == 0;
        for (int l = 0; l < desc->L; ++l)
            if (!desc->dh_b[l] || (l + 1 < desc->L && !desc->dhup_b[l])) have = false;
        p->bwd_ksplit = have;
    }
    if (p->try_persist) p->build_persist();  // persist_ok stays false when the shape / workspace does not qualify
    {   // the K-balanced backward tick (bwd8): 2-layer f32 GRU decoders with fragment-major weights and all accumulators
        const char* e = getenv("PARROT_BWD_HETERO");
        bool ok = desc->cell == 0 && desc->L == 2 && !desc->bf16 && !desc->layer_norm && p->tiled && desc->B <= 64 &&
                  desc->dw_b && desc->dw_c && desc->dw0_b && desc->dw0_c &&
                  desc->dhup_b[0] && desc->dhup_c[0] && (e ? atoi(e) != 0 : true);
        for (int l = 0; l < desc->L; ++l)
            if (!desc->dh_b[l]) ok = false;
        p->bwd_hetero = ok;
    }
    if (desc->layer_norm && desc->L >= 2) {
        bool ok = p->schedule >= 2 && p->schedule != 7;
        for (int l = 1; l < desc->L && ok; ++l)
            for (int j = 0; j < l; ++j) {
                const int pj = l * PARROT_MAX_LAYERS + j;
                if (!desc->ln_yg[pj] || !desc->ln_sg[pj] || !desc->ln_bg[pj]) ok = false;
                if (desc->cell == 0 && (!desc->ln_yc[pj] || !desc->ln_sc[pj] || !desc->ln_bc[pj])) ok = false;
            }
        if (!ok) {
            delete p;
            return PARROT_ERR_BADARG;
        }
    }
    *plan = p;
    return 0;
}
long long parrot_decoder_persist_floats(const ParrotDecoderDesc* desc) { PH_ENTRY();
    if (!desc || !DecoderPlan::persist_eligible(*desc)) return 0;
    return DecoderPlan::persist_floats(*desc, pm_max_workgroups());
}

long long parrot_sample_persist_floats(const ParrotSampleDesc* desc) { PH_ENTRY();
    if (!desc || !SamplePlan::persist_eligible(*desc)) return 0;
    return SamplePlan::persist_floats(*desc, pm_max_workgroups());
}
int parrot_sample_is_persistent(void* plan) {
    const SamplePlan* p = static_cast<SamplePlan*>(plan);
    return p->persist_ok ? (p->pieces_ok ? (p->fbc_on ? 3 : 2) : 1) : 0;
}
int parrot_sample_status(void* plan) { PH_ENTRY(); return plan ? static_cast<SamplePlan*>(plan)->persist_status() : PARROT_ERR_BADARG; }
int parrot_decoder_status(void* plan) { PH_ENTRY(); return plan ? static_cast<DecoderPlan*>(plan)->persist_status() : PARROT_ERR_BADARG; }

int parrot_decoder_is_persistent(void* plan) { return static_cast<DecoderPlan*>(plan)->persist_ok ? 1 : 0; }
int parrot_decoder_schedule(void* plan) { return plan ? static_cast<DecoderPlan*>(plan)->schedule : -1; }
long long parrot_decoder_trace_jobs(void* plan, int which, long long* out, long long cap) { PH_ENTRY();
    if (!plan || which < 0 || which > 1) return -PARROT_ERR_BADARG;
    std::vector<TraceRec> recs;
    std::vector<TraceJob> jobs;
    const int rc = static_cast<DecoderPlan*>(plan)->trace(which, recs, &jobs);
    if (rc != 0) return -(long long)rc;
    const long long n = (long long)jobs.size();
    if (out)
        for (long long i = 0; i < n && i < cap; ++i) {
            out[6 * i] = jobs[i].launch; out[6 * i + 1] = jobs[i].job; out[6 * i + 2] = jobs[i].M;
            out[6 * i + 3] = jobs[i].N; out[6 * i + 4] = jobs[i].Kmax_seg_sum; out[6 * i + 5] = jobs[i].epi;
        }
    return n;
}
long long parrot_decoder_trace(void* plan, int which, long long* out, long long cap) { PH_ENTRY();
    if (!plan || which < 0 || which > 1) return -PARROT_ERR_BADARG;
    std::vector<TraceRec> recs;
    const int rc = static_cast<DecoderPlan*>(plan)->trace(which, recs);
    if (rc != 0) return -(long long)rc;
    const long long n = (long long)recs.size();
    if (out)
        for (long long i = 0; i < n && i < cap; ++i) {
            out[5 * i] = recs[i].launch; out[5 * i + 1] = recs[i].job; out[5 * i + 2] = recs[i].kind;
            out[5 * i + 3] = recs[i].lo; out[5 * i + 4] = recs[i].hi;
        }
    return n;
}

int parrot_decoder_seq_fwd(void* plan, void* stream) { PH_ENTRY(); return static_cast<PlanBase*>(plan)->run(0, (hipStream_t)stream); }
int parrot_decoder_seq_bwd(void* plan, void* stream) { PH_ENTRY(); return static_cast<PlanBase*>(plan)->run(1, (hipStream_t)stream); }
int parrot_decoder_destroy(void* plan) { PH_ENTRY();
    delete static_cast<PlanBase*>(plan);
    return 0;
}

int parrot_sample_create(const ParrotSampleDesc* desc, void** plan) { PH_ENTRY();
    if (!desc || !plan || desc->S < 1 || desc->B < 1 || bad_dims(desc->L) || desc->ldx < desc->O)
        return PARROT_ERR_BADARG;
    SamplePlan* p = new (std::nothrow) SamplePlan();
    if (!p) return PARROT_ERR_BADARG;
    p->d = *desc;
    p->use_graph = desc->use_graph;
    p->esplit = att_default_esplit(desc->B, desc->E);
    if (desc->layer_norm) {
        bool ok = desc->ln_scratch && desc->ln_scratch_floats >= p->scratch_need();
        for (int l = 0; l < desc->L; ++l) ok = ok && desc->br_l[l];
        if (!ok) {
            delete p;
            return PARROT_ERR_BADARG;
        }
    }
    p->build_persist();  // decode on the persistent phase machine when the configuration qualifies
    *plan = p;
    return 0;
}
int parrot_sample_plan_pieces_dry(const ParrotSampleDesc* desc, int nwg, int* info16) { PH_ENTRY();
    if (!desc || !info16 || nwg < 1 || desc->S < 1 || desc->B < 1 || bad_dims(desc->L)) return PARROT_ERR_BADARG;
    std::unique_ptr<SamplePlan> p(new (std::nothrow) SamplePlan());
    if (!p) return PARROT_ERR_BADARG;
    p->d = *desc;
    p->build_persist_pieces(true, nwg);
    for (int i = 0; i < 16; ++i) info16[i] = p->pieces_info[i];
    return p->pieces_ok ? 0 : PARROT_ERR_UNSUPPORTED;
}
int parrot_sample_run(void* plan, void* stream) { PH_ENTRY(); return static_cast<PlanBase*>(plan)->run(0, (hipStream_t)stream); }
int parrot_sample_destroy(void* plan) { PH_ENTRY();
    delete static_cast<PlanBase*>(plan);
    return 0;
}

int parrot_plan_last_error(void* plan) { PH_ENTRY(); return plan ? static_cast<PlanBase*>(plan)->last_error : PARROT_ERR_BADARG; }

}  // extern "C"
