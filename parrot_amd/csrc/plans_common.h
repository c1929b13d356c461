// Shared by the scan plans (plans.hip: training) and the decode planner (plans_decode.hip): the plan base class with its
// hipGraph capture / replay, the schedule tracer behind parrot_decoder_trace, the traced launch helpers, and the unit
// placement of the persistent phase machine.  Everything has internal linkage (one copy per translation unit).
#pragma once
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <array>
#include <memory>
#include <new>
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
        mat(c.dP16, B, 4 * H, 4 * H, 1, id, 2);
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

bool bad_dims(int L) { return L < 1 || L > PARROT_MAX_LAYERS; }

}  // namespace
