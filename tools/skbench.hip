// Micro-benchmark of the recurrent-step kernel at BASELINE cfg2 shapes (development aid).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/skbench.hip -o /tmp/skbench
#include "../parrot_amd/csrc/skinny.hip"

#include <stdio.h>
#include <stdlib.h>
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
    const int B = argc > 1 ? atoi(argv[1]) : 64, H = 1024, E = 256;
    const int NSETS = 6;  // rotate weight sets so the working set (~6 x 18 MB) behaves like the real scan
    const int iters = 300;
    float* h = dalloc((size_t)B * H, 1.f);
    float* w = dalloc((size_t)B * E, 1.f);
    float* h1 = dalloc((size_t)B * H, 1.f);
    float* z = dalloc((size_t)B * H, 0.f);
    float* r = dalloc((size_t)B * H, 0.f);
    float* rh = dalloc((size_t)B * H, 0.f);
    float* bias = dalloc(2 * H, 0.1f);
    std::vector<float*> Wg(NSETS);
    const int K = H + E + H;
    for (int s = 0; s < NSETS; ++s) Wg[s] = dalloc((size_t)K * 2 * H, 0.05f);
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    struct Case { const char* name; int nseg; int N; int kcontig; };
    Case cases[] = {{"gates L2 (K=2304,N=2048) NN", 3, 2 * H, 0}, {"gates L1 (K=1280,N=2048) NN", 2, 2 * H, 0},
                    {"cand  L2 (K=2304,N=1024) NN", 3, H, 0},     {"bwdY   (K=2048,N=1024) NT", 1, H, 1}};
    for (auto& c : cases) {
        std::vector<SkLaunch> Ls(NSETS);
        for (int s = 0; s < NSETS; ++s) {
            SkJob j;
            sk_job_init(j);
            if (c.kcontig) {
                j.nseg = 1;
                j.seg[0] = sk_seg(Wg[(s + 1) % NSETS], 2 * H, Wg[s], 2 * H, 2 * H, 1);  // A = [B,2H] slab
            } else {
                j.nseg = c.nseg;
                j.seg[0] = sk_seg(h, H, Wg[s], c.N, H, 0);
                j.seg[1] = sk_seg(w, E, Wg[s] + (size_t)H * c.N, c.N, E, 0);
                if (c.nseg > 2) j.seg[2] = sk_seg(h1, H, Wg[s] + (size_t)(H + E) * c.N, c.N, H, 0);
            }
            j.M = B; j.N = c.N; j.H = H;
            if (c.N == 2 * H && !c.kcontig) {
                j.epi = SK_EPI_GRU_GATES; j.bias = bias; j.e0 = h; j.lde0 = H;
                j.o1 = z; j.ldo1 = H; j.o2 = r; j.ldo2 = H; j.out = rh; j.ldo = H;
            } else {
                j.epi = SK_EPI_LINEAR; j.out = z; j.ldo = H; j.bias = bias;
            }
            if (sk_make_launch(Ls[s], &j, 1) != 0) { printf("make_launch failed\n"); return 1; }
        }
        for (int i = 0; i < 20; ++i) sk_launch(Ls[i % NSETS], st);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) sk_launch(Ls[i % NSETS], st);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1000.0 / iters;
        const double kk = c.kcontig ? 2.0 * H : (c.nseg > 2 ? K : H + E);
        const double flops = 2.0 * B * kk * c.N;
        printf("%-32s B=%d: %8.2f us/launch  %7.1f TFLOP/s  weights %6.2f TB/s\n", c.name, B, us,
               flops / us * 1e-6, kk * c.N * 4.0 / us * 1e-6);
    }
    return 0;
}
