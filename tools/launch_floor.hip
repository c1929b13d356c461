// What does a step launch cost before it does any work?  Back-to-back launches on one stream (as the scan's graph replays them):
//   empty kernel, 256 x 512 threads                                  -> the kernel boundary itself
//   + a 3.4 KB by-value argument (sizeof(SkLaunch))                  -> kernarg size
//   + 32 KB of dynamic LDS                                           -> LDS allocation
//   + every wave reads 340 B of its descriptor (scalar loads)        -> descriptor fetch
//   + one dependent global load + one store per thread               -> first round trip + write-back
//   + LDS reduce (8 waves, one barrier)                              -> the split-K meeting point
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/launch_floor.hip -o tools/probe_bin/launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
struct Big { int w[850]; const float* src; float* dst; };  // 3.4 KB
struct Small { const float* src; float* dst; int n; };

__global__ __launch_bounds__(512) void k_empty(Small a) {}
__global__ __launch_bounds__(512) void k_bigarg(Big a) {}
__global__ __launch_bounds__(512) void k_lds(Big a) { extern __shared__ float sm[]; if (a.w[0] == 12345) sm[threadIdx.x] = 1.f; }
__global__ __launch_bounds__(512) void k_desc(Big a) {
    extern __shared__ float sm[];
    int s = 0;
    const int j = blockIdx.x % 9;
#pragma unroll
    for (int i = 0; i < 85; ++i) s += a.w[j * 85 + i];
    if (s == 12345) sm[threadIdx.x] = 1.f;
}
template <int MODE>
__global__ __launch_bounds__(512) void k_work(Big a) {
    extern __shared__ float sm[];
    int s = 0;
    const int j = blockIdx.x % 9;
#pragma unroll
    for (int i = 0; i < 85; ++i) s += a.w[j * 85 + i];
    const int t = threadIdx.x;
    float v = a.src[(size_t)blockIdx.x * 512 + t + (s & 1)];
    if (MODE >= 2) {
        sm[t] = v;
        __syncthreads();
        if (t >= 256) return;
        v = sm[t] + sm[t + 256];
    }
    if (MODE >= 3) v += a.src[(size_t)(blockIdx.x * 512 + t) + (size_t)(((int)v) & 1) * 131072];  // a second, dependent round trip
    a.dst[(size_t)blockIdx.x * 512 + t] = v;
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float *src, *dst; CK(hipMalloc(&src, 4 << 20)); CK(hipMalloc(&dst, 4 << 20)); CK(hipMemset(src, 0, 4 << 20));
    Big b; for (int i = 0; i < 850; ++i) b.w[i] = i & 1; b.src = src; b.dst = dst;
    Small s{src, dst, 0};
    const int iters = 2000;
    auto time_it = [&](const char* name, auto launch) {
        for (int i = 0; i < 50; ++i) launch();
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) launch();
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-72s %6.2f us per launch\n", name, ms * 1000.0 / iters);
    };
    for (int grid : {256, 512}) {
        printf("# grid %d workgroups of 512 threads\n", grid);
        time_it("empty kernel, 24-byte argument", [&] { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(512), 0, st, s); });
        time_it("empty kernel, 3.4 KB by-value argument", [&] { hipLaunchKernelGGL(k_bigarg, dim3(grid), dim3(512), 0, st, b); });
        time_it("+ 32 KB dynamic LDS", [&] { hipLaunchKernelGGL(k_lds, dim3(grid), dim3(512), 32768, st, b); });
        time_it("+ each wave reads its 340-byte descriptor", [&] { hipLaunchKernelGGL(k_desc, dim3(grid), dim3(512), 32768, st, b); });
        time_it("+ one global load and one store per thread", [&] { hipLaunchKernelGGL(k_work<1>, dim3(grid), dim3(512), 32768, st, b); });
        time_it("+ LDS meeting point (barrier, half the threads go on)", [&] { hipLaunchKernelGGL(k_work<2>, dim3(grid), dim3(512), 32768, st, b); });
        time_it("+ a second, dependent global round trip", [&] { hipLaunchKernelGGL(k_work<3>, dim3(grid), dim3(512), 32768, st, b); });
    }
    // the same through a graph of 200 kernel nodes (how the scan replays its launches)
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_work<2>, dim3(256), dim3(512), 32768, st, b);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-72s %6.2f us per launch\n", "graph of 200 x (load + LDS meeting point + store), 256 workgroups", ms * 1000.0 / 2000);
    }
    return 0;
}
