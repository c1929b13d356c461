// Phase timers of the recurrent-step kernel inside a replayed graph (what the scan's launches look like from within):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSK_TIMERS -Iparrot_amd/csrc tools/sktimers.hip parrot_amd/csrc/attention.hip -o tools/probe_bin/sktimers
// A graph of 2 x NL launches alternates the forward tick's gate launch (A: l0 K = H + E, l1 K = H; N = 2H; 256 workgroups of
// 32 x 32) and candidate launch (B: N = H; 256 workgroups of 32 x 16) over rotating operand sets; every wave stamps its phases
// (100 MHz wall clock) and the stamps of the LAST launch of the graph are read back: per phase the quartiles over all waves,
// relative to the launch's earliest entry.  Beside them: wall time per launch of the same graph (events around 10 replays).
#include "../parrot_amd/csrc/skinny.hip"

#include <algorithm>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
static float* dalloc(size_t n, float val) {
    float* p;
    CK(hipMalloc(&p, n * sizeof(float)));
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = val * (float)((i * 2654435761u) % 1000) / 1000.f - val * 0.5f;
    CK(hipMemcpy(p, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    return p;
}
int main(int argc, char** argv) {
    const int B = 64, H = 1024, E = 256, L = 2, NSETS = 4, NL = argc > 1 ? atoi(argv[1]) : 40;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Set { float *h[2], *rh[2], *w, *z[2], *r[2], *c[2], *hn[2], *Wg_f[2], *Wc_f[2]; };
    std::vector<Set> S(NSETS);
    float* bias = dalloc(2 * H, 0.1f);
    for (auto& s : S) {
        s.w = dalloc((size_t)B * E, 1.f);
        for (int l = 0; l < L; ++l) {
            const int K = H + E;  // schedule 5: the launches keep K = H (+ E for layer 0); the lower inputs are separate jobs
            s.h[l] = dalloc((size_t)B * H, 1.f); s.rh[l] = dalloc((size_t)B * H, 1.f);
            s.z[l] = dalloc((size_t)B * H, 0.5f); s.r[l] = dalloc((size_t)B * H, 0.5f);
            s.c[l] = dalloc((size_t)B * H, 0.5f); s.hn[l] = dalloc((size_t)B * H, 0.5f);
            float* Wg = dalloc((size_t)K * 2 * H, 0.05f);
            float* Wc = dalloc((size_t)K * H, 0.05f);
            s.Wg_f[l] = dalloc((size_t)K * 2 * H, 0.f); s.Wc_f[l] = dalloc((size_t)K * H, 0.f);
            sk_tile_weights_launch(Wg, K, 2 * H, 2 * H, s.Wg_f[l], 0, 0, st);
            sk_tile_weights_launch(Wc, K, H, H, s.Wc_f[l], 0, 0, st);
            CK(hipStreamSynchronize(st));
            CK(hipFree(Wg)); CK(hipFree(Wc));
        }
    }
    auto fseg = [&](const float* A, int lda, const float* Wt, int Krows, int r0, int K) {
        return sk_seg(A, lda, Wt + (size_t)(r0 >> 4) * 256, (Krows >> 4) * 256, K, 2);
    };
    std::vector<SkLaunch> LA(NSETS), LB(NSETS);
    for (int si = 0; si < NSETS; ++si)
        for (int kind = 0; kind < 2; ++kind) {
            Set& s = S[si];
            SkJob jobs[SK_MAXJOB];
            int n = 0;
            for (int l = 0; l < L; ++l) {
                const int K = H + E;
                SkJob& j = jobs[n++];
                sk_job_init(j);
                const float* Wt = kind == 0 ? s.Wg_f[l] : s.Wc_f[l];
                j.seg[0] = fseg(kind == 0 ? s.h[l] : s.rh[l], H, Wt, K, 0, H);
                j.nseg = 1;
                if (l == 0) { j.seg[1] = fseg(s.w, E, Wt, K, H, E); j.nseg = 2; }
                j.M = B; j.N = kind == 0 ? 2 * H : H; j.H = H; j.bias = bias;
                j.epi = kind == 0 ? SK_EPI_GRU_GATES : SK_EPI_GRU_CAND;
                j.e0 = s.h[l]; j.lde0 = H; j.e1 = s.z[l]; j.lde1 = H;
                j.o1 = kind == 0 ? s.z[l] : s.c[l]; j.ldo1 = H; j.o2 = s.r[l]; j.ldo2 = H;
                j.out = kind == 0 ? s.rh[l] : s.hn[l]; j.ldo = H;
            }
            if (sk_make_launch(kind == 0 ? LA[si] : LB[si], jobs, n) != 0) { printf("make_launch failed\n"); return 1; }
        }
    const size_t NW = 1024 * 16 * 8;
    unsigned long long* tb; CK(hipMalloc(&tb, NW * 8));
    for (int last = 0; last < 2; ++last) {  // which launch ends the graph: 0 = gate launch (A), 1 = candidate launch (B)
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 2 * NL; ++i) {
            const bool isA = ((i & 1) == 0) == (last == 1);
            if (sk_launch(isA ? LA[(i / 2) % NSETS] : LB[(i / 2) % NSETS], st) != 0) { printf("launch failed\n"); return 1; }
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        unsigned long long* nullp = nullptr;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(sk_timer_buf), &nullp, sizeof(nullp)));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("graph of %d alternating gate / candidate launches, no stamps: %.2f us per launch (pair %.2f us)\n", 2 * NL, ms * 1000.0 / (20 * NL), ms * 1000.0 / (10 * NL));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(sk_timer_buf), &tb, sizeof(tb)));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        CK(hipMemset(tb, 0, NW * 8));
        // one more replay: stamps of waves that do not reach a phase in the last launch stay 0
        {   hipGraph_t g1; hipGraphExec_t ge1;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            for (int i = 0; i < 2 * NL; ++i) {
                const bool isA = ((i & 1) == 0) == (last == 1);
                if (i == 2 * NL - 1) CK(hipMemsetAsync(tb, 0, NW * 8, st));
                sk_launch(isA ? LA[(i / 2) % NSETS] : LB[(i / 2) % NSETS], st);
            }
            CK(hipStreamEndCapture(st, &g1));
            CK(hipGraphInstantiate(&ge1, g1, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge1, st)); CK(hipStreamSynchronize(st));
            CK(hipGraphExecDestroy(ge1)); CK(hipGraphDestroy(g1)); }
        std::vector<unsigned long long> hb(NW);
        CK(hipMemcpy(hb.data(), tb, NW * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        int nwg = 0;
        for (size_t w = 0; w < NW / 8; ++w) if (hb[w * 8]) { t0 = std::min(t0, hb[w * 8]); nwg = std::max(nwg, (int)(w / 16) + 1); }
        const char* names[7] = {"entry", "segments set up, K loop starts", "K loop done", "past the LDS meeting point",
                                "epilogue stores issued", "stores acknowledged", "epilogue operands requested"};
        printf("last launch = %s: %d workgroups; stamps in us after the launch's earliest entry (min / quartile / median / quartile / max over waves)\n",
               last == 0 ? "gates (A, 32 x 32 tiles)" : "candidates (B, 32 x 16 tiles)", nwg);
        for (int ph : {0, 6, 1, 2, 3, 4, 5}) {
            std::vector<double> v;
            for (size_t w = 0; w < NW / 8; ++w) if (hb[w * 8 + ph] && hb[w * 8] && hb[w * 8 + ph] >= t0 && hb[w * 8 + ph] - t0 < 100000) v.push_back((double)(hb[w * 8 + ph] - t0) * 0.01);
            if (v.empty()) continue;
            std::sort(v.begin(), v.end());
            printf("  [%d] %-36s n=%5zu  %6.2f %6.2f %6.2f %6.2f %6.2f", ph, names[ph], v.size(), v.front(), v[v.size() / 4], v[v.size() / 2], v[3 * v.size() / 4], v.back());
            for (int wv = 0; wv < 8; ++wv) {  // median per wave index
                std::vector<double> u;
                for (size_t w = wv; w < NW / 8; w += 16) if (hb[w * 8 + ph] && hb[w * 8] && hb[w * 8 + ph] >= t0 && hb[w * 8 + ph] - t0 < 100000) u.push_back((double)(hb[w * 8 + ph] - t0) * 0.01);
                std::sort(u.begin(), u.end());
                if (wv == 0) printf("   | median by wave:");
                if (!u.empty()) printf(" %5.2f", u[u.size() / 2]);
            }
            printf("\n");
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
