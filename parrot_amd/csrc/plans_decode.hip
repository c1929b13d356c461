// The decode loop of Parrot.sample_model_fun (model.py:882-1057) as a plan: per-step launches in one hipGraph, or the
// whole loop as ONE resident kernel on the persistent phase machine (persist.h) -- 2L + 3 whole-K phases (round 2),
// 2L + 2 phases with every product cut along K by the age of its operands (round 4), 2L + 1 with the fed-back frame out
// of the step's dependency chain (round 5).  The planner, its symbolic checker and the parrot_sample_* entry points.
#include "plans_common.h"

namespace {

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
    enum { RES_XG = 10, RES_XC = 20, RES_XR = 30, RES_H = 40, RES_Z = 50, RES_X = 60, RES_KAPPA = 61, RES_XPRE = 62, RES_PP = 63,
           RES_PART = 100 };
    // Round 5: the attention projection folded into layer 0's candidate units (PmUnit::pw / pp, persist.hip)
    static bool attfold_wanted(const ParrotSampleDesc& d) {
        const char* e = getenv("PARROT_PM_ATTFOLD");
        return d.Watt_t && d.B <= 16 && 3 * d.A <= 32 && !(e && atoi(e) == 0);
    }
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
        if (attfold_wanted(d)) n += (long long)d.S * (d.H / 16) * d.B * 32 + 64;  // the projection's partial sums
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
        const bool attfold = attfold_wanted(d) && MB == 1;
        float* pp = attfold ? take((long long)S * hc * B * 32) : nullptr;  // [S][H / 16][B][32] partial projections
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
                        if (attfold && l == 0) m.wr.push_back(acc(RES_PP, 0, 0, 1));
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
                        if (attfold && l == 0) {  // this tile's share of the attention projection h_1 . Watt
                            u.pw[0] = d.Watt_t + (size_t)ct * 256;
                            u.pw[1] = d.Watt_t + (size_t)(hc + ct) * 256;
                            u.pp = rm(pp + (size_t)ct * B * 32, (long long)hc * B * 32, 32);
                        }
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
            if (attfold) m.rd.push_back(acc(RES_PP, 0, 0, 1));
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
        pieces_info[15] = (fbc ? 1 : 0) + (attfold ? 2 : 0);
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
        a.pp = pp; a.pp_st = (long long)hc * B * 32;
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
        if (attfold) add_fill(pp, (long long)S * hc * B * 32);
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

}  // namespace

extern "C" {


long long parrot_sample_persist_floats(const ParrotSampleDesc* desc) { PH_ENTRY();
    if (!desc || !SamplePlan::persist_eligible(*desc)) return 0;
    return SamplePlan::persist_floats(*desc, pm_max_workgroups());
}
int parrot_sample_is_persistent(void* plan) {
    const SamplePlan* p = static_cast<SamplePlan*>(plan);
    return p->persist_ok ? (p->pieces_ok ? (p->fbc_on ? 3 : 2) : 1) : 0;
}
int parrot_sample_status(void* plan) { PH_ENTRY(); return plan ? static_cast<SamplePlan*>(plan)->persist_status() : PARROT_ERR_BADARG; }

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

}  // extern "C"
