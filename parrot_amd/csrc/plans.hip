// Host-side scan orchestration: the loops that theano.scan ran inside one theano.function call in
// the reference (model.py:726-737, 1038-1057; ops.py:299-327) become fixed launch sequences of the
// fused step kernels, captured once into a hipGraph and replayed per window (all pointers in a
// plan are fixed, so a replay costs one hipGraphLaunch instead of thousands of host launches).
#include "plans_common.h"

namespace {

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
        BgPrecisionScope precision(d.bf16 ? 1 : -1);  // the hoisted projections follow the plan's operand mode (f32 plans: the process-wide one)
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
        const bool stream_split = false;
        (void)all_rows;
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
    bool s5_split = true;  // LSTM input projections of l >= 2 as two K-balanced jobs (false: one job; measured slower)
    int lag5(int l) const { return l == 0 ? 0 : l + 1; }
    int nticks5() const { return d.T + lag5(d.L - 1); }
    // part 0: the whole projection; 1: the rows of w and h_0 .. h_{l-2} (ready a tick earlier); 2: the rows of h_{l-1},
    // accumulated onto part 1 (LSTM stacks with l >= 2: two jobs of about the recurrent K instead of one of K = E + l H)
    void input_job(SkJob& j, int l, int t, int g, int part = 0, bool no_w = false) const {
        const size_t BH = (size_t)d.B * d.H, BE = (size_t)d.B * d.E;
        const int wd = d.cell == 1 ? 4 * d.H : (g == 0 ? 2 * d.H : d.H);  // LSTM layers: one 4H-wide matrix (g = 0)
        sk_job_init(j);
        j.colmode = d.cell == 1 && tiled ? 1 : 0;  // (the tiled copies of LSTM matrices keep the gate-interleaved tile order)
        int n = 0;
        if (part != 2 && !no_w) j.seg[n++] = fseg(d.w + (size_t)(t + 1) * BE, d.E, l, g, d.H, d.E, wd);
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
    // Round 6, GRU layers l >= 2 (the reference's own depth: model.py:312-347): the input projection walks K = E + l H
    // (2304 at l = 2) -- the longest K of its launch by far, in a launch that already holds 1.75 rounds of workgroups
    // (25.8 us at three layers).  It is cut like the LSTM one: part 1 = the rows of w and h_0 .. h_{l-2}, ready two ticks
    // before the rows of h_{l-1}, rides in the gate launch (the gate block) and the candidate launch (the candidate block) of
    // the tick in between -- both are 1.5 rounds there and take the extra half round for nothing --, part 2 = the rows of
    // h_{l-1} stays in the attention launch and accumulates.  No job of a tick walks more than K = H + E.
    bool s5_gru_split = true;
    // Round 6: the rows of w of an upper layer's input projection (K = E) ride in that layer's OWN gate / candidate job (a
    // second segment behind the recurrent block: w_{t+1} is two ticks old by then) instead of the attention launch's
    // projection jobs.  At two layers the gate launch holds K = H + E (layer 0) beside K = H (layer 1) workgroups, one per
    // CU, and the candidate launch likewise: the upper layers' workgroups walk the extra E rows while layer 0's are still
    // busy, and the attention launch -- bound by its GEMM workgroups since the attention chain shrank -- walks K = l H
    // instead of E + l H.
    bool s5_w_in_step = getenv("PARROT_S5_WSTEP") ? atoi(getenv("PARROT_S5_WSTEP")) != 0 : true;
    int fwd5(hipStream_t st) {
        const int Q = nticks5();
        const int cfull = 160;  // workgroup count at which the heterogeneous launch keeps 32 x 32 tiles (measured)
        // launches of a tick: <= L gate jobs (+ part 1 of the upper layers' gate projections), <= L candidate jobs (+ part 1
        // of the candidate projections), <= 3 input-projection jobs per upper layer
        static_assert(3 * (PARROT_MAX_LAYERS - 1) <= SK_MAXJOB && 2 * PARROT_MAX_LAYERS - 2 <= SK_MAXJOB, "fwd5: jobs[] too short");
        const bool gsplit = d.cell == 0 && s5_gru_split && d.L >= 3;
        const bool wstep = s5_w_in_step;
        for (int q = 0; q < Q; ++q) {
            SkJob jobs[SK_MAXJOB];
            int n = 0;
            for (int l = 0; l < d.L; ++l) {
                const int t = q - lag5(l);
                if (t < 0 || t >= d.T) continue;
                SkJob& j = jobs[n++];
                if (d.cell == 1) lstm_job(j, l, t);  // LSTM layers: one fused product + cell update per layer-step
                else gates_job(j, l, t);
                if (l > 0) j.nseg = wstep ? 2 : 1;  // recurrent block (+ the rows of w); the rest arrives through seq_g (has_seq)
            }
            if (gsplit)
                for (int l = 2; l < d.L; ++l) {  // part 1 of the step whose part 2 the attention launch of THIS tick adds
                    const int tp = q - lag5(l) + 1;
                    if (tp >= 0 && tp < d.T) input_job(jobs[n++], l, tp, 0, 1, wstep);
                }
            if (n > 0) PL_TRY(launch_jobs(jobs, n, st, full_wgs));
            n = 0;
            if (d.cell == 0) {
                for (int l = 0; l < d.L; ++l) {
                    const int t = q - lag5(l);
                    if (t < 0 || t >= d.T) continue;
                    SkJob& j = jobs[n++];
                    cand_job(j, l, t);
                    if (l > 0) j.nseg = wstep ? 2 : 1;
                }
                if (gsplit)
                    for (int l = 2; l < d.L; ++l) {
                        const int tp = q - lag5(l) + 1;
                        if (tp >= 0 && tp < d.T) input_job(jobs[n++], l, tp, 1, 1, wstep);
                    }
                if (n > 0) PL_TRY(launch_jobs(jobs, n, st, full_wgs));
            }
            n = 0;
            for (int l = 1; l < d.L; ++l) {
                const int t = q - lag5(l) + 1;
                const bool split = d.cell == 1 && l >= 2 && s5_split;
                const bool gs = gsplit && l >= 2;  // (part 1 was written by the gate / candidate launches of this tick)
                if (t >= 0 && t < d.T) {
                    input_job(jobs[n++], l, t, 0, (split || gs) ? 2 : 0, wstep);
                    if (d.cell == 0) input_job(jobs[n++], l, t, 1, gs ? 2 : 0, wstep);
                }
                if (split) {  // the rows that were ready a tick earlier
                    const int ta = t + 1;
                    if (ta >= 0 && ta < d.T) input_job(jobs[n++], l, ta, 0, 1, wstep);
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
    // (att_fwd_body.h; the hand-off first built for round 3's schedule 6); layer 0's workgroups -- the shortest K of the launch -- come LAST in the grid,
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
    bool bwd_split = true;  // false: the dC products stay in the Y launch (K = 3H jobs), as before round 3
    // schedule 7 with bf16 operands: the backward tick of LSTM layers as ONE launch (skinny.hip wkb_kernel); bwd_flags =
    // [ticks x 4 chains] arrival counters, plan-owned, zeroed at the head of the backward scan
    unsigned* att_flags = nullptr;  // [T + 2] arrival counters, one per tick (plan-owned, zeroed at the head of the scan)
    bool flags_fake = false;        // (placeholder for CPU-only schedule tracing: never dereferenced, never freed)
    bool bwd_fused = false;
    unsigned* bwd_flags = nullptr;
    // LSTM layers, bf16 operands, second accumulators given (ParrotDecoderDesc::dh_b ...): the backward products in two K
    // halves.  A wide workgroup streams its whole [B, 4H] operand: 156 workgroups of ~40 us each at cfg4, whatever
    // their width, and 100 idle CUs; two K halves = 312 workgroups of ~20 us.
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
                    c.dP16 = (bwd_fused && d.dG16[l]) ? static_cast<char*>(d.dG16[l]) + (size_t)t * 4 * BH * 2 : nullptr;
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
                // Only the fused tick's row blocks write the bf16 copies of the pre-activation gradients (dG16): on this
                // fall-back path they are made here, so that parrot_decoder_writes_bf16_grads() stays true for every tick
                // (the weight-gradient products would otherwise read rows nobody wrote).
                for (int c2 = 0; c2 < la.nchain; ++c2)
                    if (la.chain[c2].dP16)
                        PL_TRY(bg_to_bf16_launch(la.chain[c2].dP, la.chain[c2].dP16, (long long)d.B * 4 * H, st));
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
            // (round 6, L = 3: a tick has 10 / 6 / 11 jobs for S' / X / Y and a launch carries SK_MAXJOB = 9 -- its
            // descriptors travel by value in the 4 KB kernel-argument block --, so the lists are built generously and
            // the overflow is moved: S' -> Y (everything S' multiplies is a tick old), Y -> X for the jobs that do not
            // read what X writes: the dC products and the update-gate halves, `ymov`.)
            constexpr int JCAP = 2 * SK_MAXJOB;
            SkJob js[JCAP], jx[JCAP], jy[JCAP];
            bool ymov[JCAP];
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
                ymov[ny] = false; lin_job(jy[ny++], rseg_k(dG, l, 0, 0, 2 * H, H, H), d.B, H, H, d.dh[l] + t * BH, H, 1);
                if (l == 0) {
                    ymov[ny] = true;  lin_job(jy[ny++], rseg(dC, 0, 1, H, H), d.B, E, H, d.dw0 + (size_t)t * BE, E, 1);
                    ymov[ny] = true;  lin_job(jy[ny++], rseg_k(dG, 0, 0, H, 2 * H, 0, H), d.B, E, H, d.dw0_b + (size_t)t * BE, E, 0);
                    ymov[ny] = false; lin_job(jy[ny++], rseg_k(dG, 0, 0, H, 2 * H, H, H), d.B, E, H, d.dw0_c + (size_t)t * BE, E, 0);
                } else {
                    ymov[ny] = true;  lin_job(jy[ny++], rseg(dC, l, 1, H, H), d.B, E, H, d.dw + (size_t)(t + 1) * BE, E, 1);
                    for (int p = 0; p < l; ++p) {
                        ymov[ny] = true;
                        lin_job(jy[ny++], rseg(dC, l, 1, H + E + p * H, H), d.B, H, H, d.dhup[p] + (t + 1) * BH, H, 1);
                    }
                }
            }
            while (ns > SK_MAXJOB) { ymov[ny] = true; jy[ny++] = js[--ns]; }  // (a tick old: movable further, too)
            for (int i = ny - 1; i >= 0 && ny > SK_MAXJOB && nx < SK_MAXJOB; --i)
                if (ymov[i]) {
                    jx[nx++] = jy[i];
                    for (int k2 = i; k2 + 1 < ny; ++k2) { jy[k2] = jy[k2 + 1]; ymov[k2] = ymov[k2 + 1]; }
                    --ny;
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
        BgPrecisionScope precision(d.bf16 ? 1 : -1);
        return PlanBase::run(which, s);
    }

    ~DecoderPlan() override {
        if (att_flags && !flags_fake) (void)hipFree(att_flags);
        if (bwd_flags && !flags_fake) (void)hipFree(bwd_flags);
    }
};

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
        p->tiled = all;
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
    p->bwd_split = true;                      // the dC products ride in the X launch of the backward tick (round 3)
    p->esplit5 = 1; p->s5_split = true;       // schedule 5: one attention block per row, layer >= 2 projections in two jobs
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
        if (p->schedule == 7 && desc->bf16) {
            // the backward tick as one launch too (wkb_kernel)
            const size_t words = (size_t)4 * (desc->T + desc->L);
            if (p->flags_fake) p->bwd_flags = reinterpret_cast<unsigned*>((uintptr_t)0x100000);
            else if (hipMalloc(&p->bwd_flags, sizeof(unsigned) * words) != hipSuccess) { delete p; return PARROT_ERR_BADARG; }
            p->bwd_fused = true;
        }
    }
    if (desc->cell == 1 && desc->bf16 && p->tiled) {
        bool have = desc->dw_b && desc->dw0_b && desc->B <= 64 && sk_wide_takes(desc->B, 4096, desc->H, desc->E) &&
                    (4 * desc->H) % 128 == 0;
        for (int l = 0; l < desc->L; ++l)
            if (!desc->dh_b[l] || (l + 1 < desc->L && !desc->dhup_b[l])) have = false;
        p->bwd_ksplit = have;
    }
    if (p->try_persist) p->build_persist();  // persist_ok stays false when the shape / workspace does not qualify
    {   // the K-balanced backward tick (bwd8): 2-layer f32 GRU decoders with fragment-major weights and all accumulators
        const char* e = getenv("PARROT_BWD_HETERO");
        bool ok = desc->cell == 0 && (desc->L == 2 || desc->L == 3) && !desc->bf16 && !desc->layer_norm && p->tiled && desc->B <= 64 &&
                  desc->dw_b && desc->dw_c && desc->dw0_b && desc->dw0_c && (e ? atoi(e) != 0 : true);
        for (int l = 0; l < desc->L; ++l)
            if (!desc->dh_b[l] || (l + 1 < desc->L && (!desc->dhup_b[l] || !desc->dhup_c[l]))) ok = false;
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
int parrot_decoder_status(void* plan) { PH_ENTRY(); return plan ? static_cast<DecoderPlan*>(plan)->persist_status() : PARROT_ERR_BADARG; }

int parrot_decoder_is_persistent(void* plan) { return static_cast<DecoderPlan*>(plan)->persist_ok ? 1 : 0; }
int parrot_decoder_schedule(void* plan) { return plan ? static_cast<DecoderPlan*>(plan)->schedule : -1; }
int parrot_decoder_backward_tick(void* plan) { return plan ? (static_cast<DecoderPlan*>(plan)->bwd_hetero ? 8 : (static_cast<DecoderPlan*>(plan)->bwd_fused ? 7 : 0)) : -1; }
int parrot_decoder_writes_bf16_grads(void* plan) {
    const DecoderPlan* p = static_cast<const DecoderPlan*>(plan);
    if (!p || !p->bwd_fused) return 0;
    for (int l = 0; l < p->d.L; ++l)
        if (!p->d.dG16[l]) return 0;
    return 1;
}
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

int parrot_plan_last_error(void* plan) { PH_ENTRY(); return plan ? static_cast<PlanBase*>(plan)->last_error : PARROT_ERR_BADARG; }

}  // extern "C"
