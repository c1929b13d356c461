// skbench3 + (a) the launch descriptor read through a device pointer instead of the kernarg segment (SKB_PTR=1),
// (b) a K sweep of a single N = 4096 job on 256 workgroups (intercept = fixed cost of a launch, slope = cost per K),
// (c) SKB_NSETS = operand sets rotated through (1: weights stay cache-resident), (d) PARROT_SK_LDS_PAD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DSK_A_FRAG_PROBE=1] tools/skbench4.hip parrot_amd/csrc/attention.hip -o tools/probe_bin/skbench4
#include "../parrot_amd/csrc/skinny.hip"

#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int MB, int NB>
__global__ __launch_bounds__(SK_THREADS) void sk_kernel_p(const SkLaunch* __restrict__ Lp) {
    extern __shared__ __attribute__((aligned(16))) char sk_smem[];
    f32x4* red = reinterpret_cast<f32x4*>(sk_smem);
    const SkLaunch& L = *Lp;
    int j = blockIdx.z, bx = blockIdx.x;
    if (!L.zmode) {
        j = 0;
#pragma unroll
        for (int q = 0; q < SK_MAXJOB - 1; ++q)
            if (q < L.njobs - 1 && bx >= L.tile_end[q]) j = q + 1;
        bx -= (j > 0 ? L.tile_end[j - 1] : 0);
    }
    const SkJob& job = L.job[j];
    sk_body<MB, NB, true>(job, bx * NB, red);
}
static int g_ptr = 0;
static SkLaunch* g_dev[64];
static int g_ndev = 0;
static void launch_any(const SkLaunch& Lin, int slot, hipStream_t st) {
    if (!g_ptr) { sk_launch(Lin, st); return; }
    SkLaunch L; dim3 grid; size_t lds; int mbnb;
    sk_prepare(Lin, L, grid, lds, mbnb);
    if (!g_dev[slot]) {
        (void)hipMalloc(&g_dev[slot], sizeof(SkLaunch));
        (void)hipMemcpy(g_dev[slot], &L, sizeof(SkLaunch), hipMemcpyHostToDevice);
    }
    if (mbnb == 22) hipLaunchKernelGGL((sk_kernel_p<2, 2>), grid, dim3(SK_THREADS), lds, st, g_dev[slot]);
    else if (mbnb == 21) hipLaunchKernelGGL((sk_kernel_p<2, 1>), grid, dim3(SK_THREADS), lds, st, g_dev[slot]);
    else { printf("unexpected tile %d\n", mbnb); exit(1); }
}
static void reset_dev() { for (int i = 0; i < 64; ++i) if (g_dev[i]) { (void)hipFree(g_dev[i]); g_dev[i] = nullptr; } }

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
    const int B = argc > 1 ? atoi(argv[1]) : 64, H = 1024, E = 256, L = 2;
    const int NSETS = getenv("SKB_NSETS") ? atoi(getenv("SKB_NSETS")) : 4, iters = 400;
    g_ptr = getenv("SKB_PTR") ? atoi(getenv("SKB_PTR")) : 0;
    {
        const int big = 160 * 1024;
        (void)hipFuncSetAttribute((const void*)sk_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
        (void)hipFuncSetAttribute((const void*)sk_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
        (void)hipFuncSetAttribute((const void*)sk_kernel_p<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
        (void)hipFuncSetAttribute((const void*)sk_kernel_p<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    }
    printf("# nsets %d  descriptor %s  lds_pad %s\n", NSETS, g_ptr ? "device pointer" : "kernarg", getenv("PARROT_SK_LDS_PAD") ? getenv("PARROT_SK_LDS_PAD") : "0");
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // per set: states, outputs, tiled weights (forward copies of Wg [K_l,2H], Wc [K_l,H]; reverse copies for the backward)
    struct Set { float *h[2], *rh[2], *w, *z[2], *r[2], *c[2], *hn[2], *Wg_f[2], *Wc_f[2], *Wg_r[2], *Wc_r[2], *dC[2], *dG[2], *dh[2], *dw; };
    std::vector<Set> S(NSETS);
    float* bias = dalloc(2 * H, 0.1f);
    for (auto& s : S) {
        s.w = dalloc((size_t)B * E, 1.f);
        s.dw = dalloc((size_t)B * E, 1.f);
        for (int l = 0; l < L; ++l) {
            const int K = H + E + l * H;
            s.h[l] = dalloc((size_t)B * H, 1.f); s.rh[l] = dalloc((size_t)B * H, 1.f);
            s.z[l] = dalloc((size_t)B * H, 0.5f); s.r[l] = dalloc((size_t)B * H, 0.5f);
            s.c[l] = dalloc((size_t)B * H, 0.5f); s.hn[l] = dalloc((size_t)B * H, 0.5f);
            s.dC[l] = dalloc((size_t)B * H, 0.5f); s.dG[l] = dalloc((size_t)B * 2 * H, 0.5f); s.dh[l] = dalloc((size_t)B * H, 0.5f);
            float* Wg = dalloc((size_t)K * 2 * H, 0.05f);
            float* Wc = dalloc((size_t)K * H, 0.05f);
            s.Wg_f[l] = dalloc((size_t)K * 2 * H, 0.f); s.Wc_f[l] = dalloc((size_t)K * H, 0.f);
            s.Wg_r[l] = dalloc((size_t)K * 2 * H, 0.f); s.Wc_r[l] = dalloc((size_t)K * H, 0.f);
            sk_tile_weights_launch(Wg, K, 2 * H, 2 * H, s.Wg_f[l], 0, 0, st);
            sk_tile_weights_launch(Wc, K, H, H, s.Wc_f[l], 0, 0, st);
            sk_tile_weights_launch(Wg, K, 2 * H, 2 * H, s.Wg_r[l], 1, 0, st);
            sk_tile_weights_launch(Wc, K, H, H, s.Wc_r[l], 1, 0, st);
            CK(hipStreamSynchronize(st));
            CK(hipFree(Wg)); CK(hipFree(Wc));
        }
    }
    auto fseg = [&](const float* A, int lda, const float* Wt, int Krows, int r0, int K) {
        return sk_seg(A, lda, Wt + (size_t)(r0 >> 4) * 256, (Krows >> 4) * 256, K, 2);
    };
    auto rseg = [&](const float* A, const float* Wt, int r0, int ldw) {
        return sk_seg(A, ldw, Wt + (size_t)(r0 >> 4) * (ldw >> 4) * 256, (ldw >> 4) * 256, ldw, 2);
    };
    const char* names[4] = {"fwd gates (l0 K=1280, l1 K=2304; N=2048)", "fwd cand  (l0 K=1280, l1 K=2304; N=1024)",
                            "bwd X d(r.h) (2 x K=1024, N=1024)", "bwd Y (5 jobs: dh, dw, dhup)"};
    for (int kind = 0; kind < 4; ++kind) {
        std::vector<SkLaunch> Ls(NSETS);
        double flops = 0;
        for (int si = 0; si < NSETS; ++si) {
            Set& s = S[si];
            SkJob jobs[SK_MAXJOB];
            int n = 0;
            flops = 0;
            for (int l = 0; l < L; ++l) {
                const int K = H + E + l * H;
                if (kind <= 1) {
                    SkJob& j = jobs[n++];
                    sk_job_init(j);
                    const float* Wt = kind == 0 ? s.Wg_f[l] : s.Wc_f[l];
                    j.seg[0] = fseg(kind == 0 ? s.h[l] : s.rh[l], H, Wt, K, 0, H);
                    j.seg[1] = fseg(s.w, E, Wt, K, H, E);
                    j.nseg = 2;
                    if (l == 1) { j.seg[2] = fseg(s.h[0], H, Wt, K, H + E, H); j.nseg = 3; }
                    j.M = B; j.N = kind == 0 ? 2 * H : H; j.H = H; j.bias = bias;
                    j.epi = kind == 0 ? SK_EPI_GRU_GATES : SK_EPI_GRU_CAND;
                    j.e0 = s.h[l]; j.lde0 = H; j.e1 = s.z[l]; j.lde1 = H;
                    j.o1 = kind == 0 ? s.z[l] : s.c[l]; j.ldo1 = H; j.o2 = s.r[l]; j.ldo2 = H;
                    j.out = kind == 0 ? s.rh[l] : s.hn[l]; j.ldo = H;
                    flops += 2.0 * B * K * j.N;
                } else if (kind == 2) {
                    SkJob& x = jobs[n++];
                    sk_job_init(x);
                    x.nseg = 1; x.seg[0] = rseg(s.dC[l], s.Wc_r[l], 0, H);
                    x.M = B; x.N = H; x.H = H; x.epi = SK_EPI_BWD_RH;
                    x.e0 = s.h[l]; x.lde0 = H; x.e1 = s.r[l]; x.lde1 = H;
                    x.out = s.dG[l] + H; x.ldo = 2 * H; x.o1 = s.dh[l]; x.ldo1 = H;
                    flops += 2.0 * B * H * H;
                } else {
                    {   SkJob& j = jobs[n++]; sk_job_init(j); j.nseg = 1; j.seg[0] = rseg(s.dG[l], s.Wg_r[l], 0, 2 * H);
                        j.M = B; j.N = H; j.H = H; j.epi = SK_EPI_LINEAR; j.accumulate = 1; j.out = s.dh[l]; j.ldo = H;
                        flops += 2.0 * B * 2 * H * H; }
                    {   SkJob& j = jobs[n++]; sk_job_init(j); j.nseg = 2; j.seg[0] = rseg(s.dG[l], s.Wg_r[l], H, 2 * H);
                        j.seg[1] = rseg(s.dC[l], s.Wc_r[l], H, H);
                        j.M = B; j.N = E; j.H = H; j.epi = SK_EPI_LINEAR; j.accumulate = 1; j.out = s.dw; j.ldo = E;
                        flops += 2.0 * B * 3 * H * E; }
                    if (l == 1) {
                        SkJob& j = jobs[n++]; sk_job_init(j); j.nseg = 2; j.seg[0] = rseg(s.dG[l], s.Wg_r[l], H + E, 2 * H);
                        j.seg[1] = rseg(s.dC[l], s.Wc_r[l], H + E, H);
                        j.M = B; j.N = H; j.H = H; j.epi = SK_EPI_LINEAR; j.accumulate = 1; j.out = s.hn[0]; j.ldo = H;
                        flops += 2.0 * B * 3 * H * H;
                    }
                }
            }
            if (sk_make_launch(Ls[si], jobs, n) != 0) { printf("make_launch failed\n"); return 1; }
        }
        reset_dev();
        for (int i = 0; i < 20; ++i) launch_any(Ls[i % NSETS], i % NSETS, st);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) launch_any(Ls[i % NSETS], i % NSETS, st);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1000.0 / iters;
        printf("%-46s B=%d: %7.2f us/launch  %6.1f TFLOP/s (MFMA floor %.2f us)\n", names[kind], B, us, flops / us * 1e-6, flops / 157.3e6);
    }
    // K sweep: one job, N = 4096 (256 workgroups of 32 x 32), plain linear epilogue
    {
        const int N = 4096, KMAX = 4096;
        std::vector<float*> W(NSETS), A(NSETS), O(NSETS);
        for (int si = 0; si < NSETS; ++si) { W[si] = dalloc((size_t)KMAX * N, 0.05f); A[si] = dalloc((size_t)B * KMAX, 1.f); O[si] = dalloc((size_t)B * N, 1.f); }
        const int Ks[] = {16, 128, 256, 512, 1024, 2048, 4096};
        for (int K : Ks) {
            std::vector<SkLaunch> Ls(NSETS);
            for (int si = 0; si < NSETS; ++si) {
                SkJob j; sk_job_init(j);
                j.nseg = 1; j.seg[0] = sk_seg(A[si], KMAX, W[si], (KMAX >> 4) * 256, K, 2);
                j.M = B; j.N = N; j.H = N; j.epi = SK_EPI_LINEAR; j.out = O[si]; j.ldo = N;
                if (sk_make_launch(Ls[si], &j, 1) != 0) { printf("make_launch failed\n"); return 1; }
            }
            reset_dev();
            for (int i = 0; i < 20; ++i) launch_any(Ls[i % NSETS], i % NSETS, st);
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) launch_any(Ls[i % NSETS], i % NSETS, st);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1000.0 / iters, flops = 2.0 * B * K * N;
            printf("sweep N=4096 K=%-5d B=%d: %7.2f us/launch  %6.1f TFLOP/s (MFMA floor %.2f us, weights %.1f MB)\n", K, B, us, flops / us * 1e-6, flops / 157.3e6, 4e-6 * K * N);
        }
    }
    return 0;
}
