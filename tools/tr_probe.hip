// tr_probe: lane mapping of ds_read_b64_tr_b16 (gfx950) -- LDS holds sm[i] = i; pattern 0: lane l points at elements 4l .. 4l+3.
// Result (profiles/r02_ds_read_tr16_probe.txt): within a 16-lane group, lane i receives elements i, 16+i, 32+i, 48+i of the
// 64 the group points at.  build: hipcc -O2 --offload-arch=gfx950 -o tr_probe tools/tr_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned short* out, const int* addr) {
    __shared__ unsigned short sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) sm[i] = in[i];
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sm + addr[threadIdx.x]));
    for (int q = 0; q < 4; ++q) out[threadIdx.x * 4 + q] = (unsigned short)v[q];
}
int main() {
    std::vector<unsigned short> in(4096); for (int i = 0; i < 4096; ++i) in[i] = i;
    unsigned short *din, *dout; int* daddr;
    hipMalloc(&din, 8192); hipMalloc(&dout, 512); hipMalloc(&daddr, 256);
    hipMemcpy(din, in.data(), 8192, hipMemcpyHostToDevice);
    for (int pat = 0; pat < 2; ++pat) {
        std::vector<int> addr(64);
        for (int l = 0; l < 64; ++l) addr[l] = pat == 0 ? 4 * l : ((l % 16) * 64 + (l / 16) * 4);  // pat 1: row l%16 (pitch 64 elems), col group l/16
        hipMemcpy(daddr, addr.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, daddr);
        std::vector<unsigned short> out(256);
        hipMemcpy(out.data(), dout, 512, hipMemcpyDeviceToHost);
        printf("pattern %d\n", pat);
        for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int q = 0; q < 4; ++q) printf(" %5d", out[l * 4 + q]); printf("\n"); }
    }
    return 0;
}
