// Do two chains of dependent kernels run side by side on this runtime?  Replays of two hipGraphs on two streams, one
// hipGraph with two captured branches, and eager launches from two host threads -- each against the serial time.
// Also the host cost of a graph replay per kernel node (0.01-0.02 us: not a bottleneck).  (development probe, round 3)
//   hipcc --offload-arch=gfx950 -O2 -o tools/probe_bin/graph_host_probe tools/graph_host_probe.hip -lpthread
//   run under DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 / DEBUG_HIP_FORCE_GRAPH_QUEUES=4 / ... to compare runtime modes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
struct Big { float* p; int iters; int pad[13]; char fat[2560]; };  // ~2.6 KB, like SkLaunch
__global__ void spin(const Big a) {
    float v = a.p[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < a.iters; ++i) v = v * 1.000001f + 0.5f;
    a.p[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static hipGraphExec_t build(hipStream_t cap, float* buf, int nodes, int iters, int blocks) {
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(cap, hipStreamCaptureModeRelaxed));
    Big a; a.p = buf; a.iters = iters;
    for (int i = 0; i < nodes; ++i) hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, cap, a);
    CK(hipStreamEndCapture(cap, &g));
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    CK(hipGraphDestroy(g));
    return ex;
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 600;   // ~10 us kernels
    const int nodes = argc > 2 ? atoi(argv[2]) : 1000;
    const int blocks = argc > 3 ? atoi(argv[3]) : 128;  // half the chip per kernel
    float *a, *b;
    CK(hipMalloc(&a, 1 << 24)); CK(hipMalloc(&b, 1 << 24));
    CK(hipMemset(a, 0, 1 << 24)); CK(hipMemset(b, 0, 1 << 24));
    hipStream_t cap, cap2, s0, s1;
    CK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&cap2, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipGraphExec_t e0 = build(cap, a, nodes, iters, blocks), e1 = build(cap, b, nodes, iters, blocks);
    // one graph with two branches
    hipGraphExec_t e2;
    {
        hipGraph_t g;
        hipEvent_t ef, ej;
        CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
        Big x; x.p = a; x.iters = iters;
        Big y; y.p = b; y.iters = iters;
        CK(hipStreamBeginCapture(cap, hipStreamCaptureModeRelaxed));
        CK(hipEventRecord(ef, cap)); CK(hipStreamWaitEvent(cap2, ef, 0));
        for (int i = 0; i < nodes; ++i) {
            hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, cap, x);
            hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, cap2, y);
        }
        CK(hipEventRecord(ej, cap2)); CK(hipStreamWaitEvent(cap, ej, 0));
        CK(hipStreamEndCapture(cap, &g));
        CK(hipGraphInstantiate(&e2, g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
    }
    double serial = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        const double t0 = now();
        CK(hipGraphLaunch(e0, s0));
        const double t1 = now();
        CK(hipStreamSynchronize(s0));
        serial = now() - t0;
        if (rep == 2) printf("one chain (graph replay): host %.3f us/node, device %.2f us/node\n", 1e6 * (t1 - t0) / nodes, 1e6 * serial / nodes);
    }
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        const double t0 = now();
        CK(hipGraphLaunch(e0, s0));
        CK(hipGraphLaunch(e1, s1));
        CK(hipDeviceSynchronize());
        if (rep == 2) printf("two graph replays on two streams : %.2f x the time of one chain\n", (now() - t0) / serial);
    }
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        const double t0 = now();
        CK(hipGraphLaunch(e2, s0));
        CK(hipDeviceSynchronize());
        if (rep == 2) printf("one graph with two branches      : %.2f x\n", (now() - t0) / serial);
    }
    Big x; x.p = a; x.iters = iters;
    Big y; y.p = b; y.iters = iters;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        const double t0 = now();
        for (int i = 0; i < nodes; ++i) hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s0, x);
        const double t1 = now();
        CK(hipDeviceSynchronize());
        if (rep == 2) printf("one chain, eager launches        : %.2f x (host %.2f us per launch)\n", (now() - t0) / serial, 1e6 * (t1 - t0) / nodes);
    }
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        const double t0 = now();
        std::thread th([&] { for (int i = 0; i < nodes; ++i) hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s1, y); });
        for (int i = 0; i < nodes; ++i) hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s0, x);
        th.join();
        CK(hipDeviceSynchronize());
        if (rep == 2) printf("two chains, eager, two host threads, two streams: %.2f x\n", (now() - t0) / serial);
    }
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        const double t0 = now();
        for (int i = 0; i < nodes; ++i) {
            hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s0, x);
            hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s1, y);
        }
        CK(hipDeviceSynchronize());
        if (rep == 2) printf("two chains, eager, one host thread, two streams : %.2f x\n", (now() - t0) / serial);
    }
    return 0;
}
