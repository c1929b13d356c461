// xcc_map_probe: does workgroup i of a 256-workgroup launch run on XCD (i + c) % 8 with ONE c per launch?  (sr_persist.hip
// numbers the workgroups of an XCD by blockIdx / 8 on that assumption; c itself depends on where the previous kernel's grid
// ended -- a 1-workgroup kernel in front rotates it.)  Launches the grid many times, behind kernels of even and odd sizes,
// plain and inside a replayed graph; counts the launches in which more than one rotation was seen, and the workgroups
// off the c = 0 mapping (for the record: not zero once odd grids run in front).
//   build: make -C tools xcc_map_probe ; run: tools/probe_bin/xcc_map_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); exit(1); } } while (0)

__global__ __launch_bounds__(512) void who(unsigned* bad, unsigned* hist, int spin, unsigned* rot) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    if (threadIdx.x == 0) {
        if ((v & 7) != (blockIdx.x & 7)) atomicAdd(bad, 1u);
        atomicAdd(hist + (v & 7), 1u);
        atomicOr(rot, 1u << ((v - blockIdx.x) & 7));  // the rotation this workgroup sees
    }
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(32);
}
__global__ void filler(float* p, int n) {  // a kernel of many short workgroups in front: the sample kernel never starts on an idle chip
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}

int main() {
    unsigned *bad, *hist, *rot;
    float* buf;
    const int NL = 2000 + 16;
    CHECK(hipMalloc(&bad, 4)); CHECK(hipMalloc(&hist, 32)); CHECK(hipMalloc(&buf, 64 << 20)); CHECK(hipMalloc(&rot, NL * 4));
    CHECK(hipMemset(rot, 0, NL * 4));
    CHECK(hipMemset(bad, 0, 4)); CHECK(hipMemset(hist, 0, 32)); CHECK(hipMemset(buf, 0, 64 << 20));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    const int lds = 62 * 1024;
    for (int rep = 0; rep < 2000; ++rep) {
        if (rep & 1) hipLaunchKernelGGL(filler, dim3(16384 + rep % 5), dim3(256), 0, st, buf, 16 << 20);
        if (rep % 3 == 0) hipLaunchKernelGGL(filler, dim3(1), dim3(64), 0, st, buf, 64);
        hipLaunchKernelGGL(who, dim3(256), dim3(512), lds, st, bad, hist, rep % 7, rot + rep);
    }
    CHECK(hipStreamSynchronize(st));
    unsigned hb = 0, hh[8];
    CHECK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hh, hist, 32, hipMemcpyDeviceToHost));
    printf("eager: 2000 launches x 256 workgroups: %u workgroups off blockIdx %% 8; per XCD:", hb);
    for (int i = 0; i < 8; ++i) printf(" %u", hh[i]);
    printf("\n");
    CHECK(hipMemset(bad, 0, 4));
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    for (int rep = 0; rep < 16; ++rep) {
        hipLaunchKernelGGL(filler, dim3(4096 + rep), dim3(256), 0, st, buf, 1 << 20);
        hipLaunchKernelGGL(who, dim3(256), dim3(512), lds, st, bad, hist, rep % 5, rot + 2000 + rep);
    }
    CHECK(hipStreamEndCapture(st, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 200; ++rep) CHECK(hipGraphLaunch(ge, st));
    CHECK(hipStreamSynchronize(st));
    CHECK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    printf("graph: 3200 launches x 256 workgroups: %u workgroups off blockIdx %% 8\n", hb);
    static unsigned hr[2016];
    CHECK(hipMemcpy(hr, rot, NL * 4, hipMemcpyDeviceToHost));
    int mixed = 0, seen[8] = {0};
    for (int i = 0; i < NL; ++i) {
        if (hr[i] & (hr[i] - 1)) ++mixed;
        for (int c = 0; c < 8; ++c) if (hr[i] >> c & 1) ++seen[c];
    }
    printf("launch slots with more than one rotation (blockIdx %% 8 classes split over XCDs): %d of %d; rotations seen:", mixed, NL);
    for (int c = 0; c < 8; ++c) printf(" %d", seen[c]);
    printf("\n");
    return 0;
}
