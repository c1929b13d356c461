// Does hipGraph replay overlap independent branches captured from two streams?  (development probe)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void spin(float* p, int iters) {
    float v = p[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.000001f + 0.5f;
    p[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
int main() {
    float *a, *b;
    CK(hipMalloc(&a, 1 << 22)); CK(hipMalloc(&b, 1 << 22));
    hipStream_t s0, s1, s2; CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    hipEvent_t e0, e1, ef, ej; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    const int blocks = 64, reps = 200; const int iters = getenv("ITERS") ? atoi(getenv("ITERS")) : 20000;  // 64 blocks: a quarter of the chip per kernel
    for (int mode = 0; mode < 3; ++mode) {  // 0: serial one stream, 1: two streams eager, 2: graph with two branches
        hipGraphExec_t exec = nullptr;
        if (mode == 2) {
            hipGraph_t g;
            CK(hipStreamBeginCapture(s0, hipStreamCaptureModeRelaxed));
            for (int r = 0; r < reps; ++r) {
                CK(hipEventRecord(ef, s0)); CK(hipStreamWaitEvent(s1, ef, 0));
                hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s0, a, iters);
                hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s1, b, iters);
                CK(hipEventRecord(ej, s1)); CK(hipStreamWaitEvent(s0, ej, 0));
            }
            CK(hipStreamEndCapture(s0, &g));
            CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
        }
        for (int w = 0; w < 2; ++w) {
            CK(hipEventRecord(e0, s2));
            if (mode == 0) {
                for (int r = 0; r < reps; ++r) {
                    hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s2, a, iters);
                    hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s2, b, iters);
                }
            } else if (mode == 1) {
                for (int r = 0; r < reps; ++r) {
                    CK(hipEventRecord(ef, s2)); CK(hipStreamWaitEvent(s1, ef, 0));
                    hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s2, a, iters);
                    hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s1, b, iters);
                    CK(hipEventRecord(ej, s1)); CK(hipStreamWaitEvent(s2, ej, 0));
                }
            } else {
                CK(hipGraphLaunch(exec, s2));
            }
            CK(hipEventRecord(e1, s2));
            CK(hipStreamSynchronize(s2)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (w == 1) printf("mode %d: %.1f us per pair\n", mode, ms * 1000 / reps);
        }
    }
    return 0;
}
