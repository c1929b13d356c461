// stream_probe: how fast can every XCD stream the SAME weight slices (the SampleRNN frame tier's 54 MB) while its 32 CUs
// each take their own 32 columns -- the access pattern of a frame tier that stays XCD-local inside the resident sample
// kernel.  256 workgroups of 512 threads (one per CU), teams by HW_REG_XCC_ID as in sr_persist.hip.
//
//   mode 0: all 8 waves load a slice straight into registers (16 x 16 B per thread), one slice after the other
//   mode 1: waves 4-7 stream the slice into LDS with global_load_lds_dwordx4 (1 KB per instruction, 16 in flight per
//           wave, wait for all, consume) while waves 0-3 idle -- also checks the lane -> LDS address mapping of the b128
//           LDS-DMA on gfx950 (lane l lands at base + 16 l)
//   mode 2: as 1 with two batches of 8 in flight (vmcnt(8) before a batch is consumed)
// Every variant sums what it loaded and the host checks the sums.
//   build: hipcc -O3 --offload-arch=gfx950 -o stream_probe tools/stream_probe.hip ; run: ./stream_probe [reps=20]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); exit(1); } } while (0)

constexpr int D = 1024, FS = 10, LD = FS * D, DC = 32;

__device__ __forceinline__ int xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return (int)(v & 7);
}
__device__ __forceinline__ unsigned long long clk() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

struct Out {
    unsigned census[8];
    unsigned pad[8];
    unsigned long long ticks[256];
    float sums[256][512];
};

template <int MODE>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ W, Out* o, int reps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int sh_rank;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, x = xcc_id();
    if (tid == 0) sh_rank = (int)atomicAdd(&o->census[x], 1u) % 32;
    __syncthreads();
    const int cu = sh_rank;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = clk();
    if (MODE == 0) {
        const int g = (tid >> 3) % 8, s = 8 * (tid / 64) + (tid & 7);
        for (int rep = 0; rep < reps; ++rep)
            for (int i = 0; i < FS; ++i) {
                f32x4 w[16];
#pragma unroll
                for (int kk = 0; kk < 16; ++kk)
                    w[kk] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(W + (size_t)(kk * 64 + s) * LD + i * D + cu * DC + 4 * g));
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) acc += w[kk];
            }
    } else if (wave >= 4) {
        const int v = wave - 4;
        f32x4* ring = reinterpret_cast<f32x4*>(smem) + v * 16 * 64;  // 16 slots of 1 KB per wave
        constexpr int BATCH = MODE == 1 ? 16 : 8;
        for (int rep = 0; rep < reps; ++rep)
            for (int i = 0; i < FS; ++i) {
                auto issue = [&](int j) {  // chunk j of this wave's K quarter: 8 k-rows x 32 columns
                    const float* src = W + (size_t)(v * 256 + j * 8 + lane / 8) * LD + i * D + cu * DC + 4 * (lane % 8);
                    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(ring + (j % 16) * 64), 16, 0, 0);
                };
                if (MODE == 1) {
                    for (int b = 0; b < 2; ++b) {
#pragma unroll
                        for (int j = 0; j < 16; ++j) issue(b * 16 + j);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                        for (int j = 0; j < 16; ++j) acc += ring[j * 64 + lane];
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) issue(j);
                    for (int b = 0; b < 4; ++b) {
                        if (b < 3) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) issue((b + 1) * 8 + j);
                            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                        } else {
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        }
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc += ring[((b * 8 + j) % 16) * 64 + lane];
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                }
            }
    }
    const unsigned long long t1 = clk();
    o->sums[x * 32 + cu][tid] = acc[0] + acc[1] + acc[2] + acc[3];
    if (tid == (MODE == 0 ? 0 : 256)) o->ticks[x * 32 + cu] = t1 - t0;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    std::vector<float> hW((size_t)D * LD);
    for (int k = 0; k < D; ++k)
        for (int n = 0; n < LD; ++n) hW[(size_t)k * LD + n] = (float)((k * 7 + n * 3) % 13);
    float* W;
    Out* o;
    CHECK(hipMalloc(&W, hW.size() * 4));
    CHECK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&o, sizeof(Out)));
    std::vector<Out> ho(1);
    for (int mode = 0; mode < 3; ++mode) {
        for (int pass = 0; pass < 2; ++pass) {
            CHECK(hipMemset(o, 0, sizeof(Out)));
            const size_t lds = 64 * 1024;
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), lds, 0, W, o, reps);
            if (mode == 1) {
                CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(probe<1>, dim3(256), dim3(512), lds, 0, W, o, reps);
            }
            if (mode == 2) {
                CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(probe<2>, dim3(256), dim3(512), lds, 0, W, o, reps);
            }
            CHECK(hipDeviceSynchronize());
        }
        CHECK(hipMemcpy(ho.data(), o, sizeof(Out), hipMemcpyDeviceToHost));
        // check: per-thread sums
        long long bad = 0;
        for (int x = 0; x < 8; ++x) {
            if (ho[0].census[x] != 32) printf("mode %d: XCD %d has %u workgroups\n", mode, x, ho[0].census[x]);
            for (int cu = 0; cu < 32; ++cu)
                for (int tid = 0; tid < 512; ++tid) {
                    double e = 0;
                    if (mode == 0) {
                        const int g = (tid >> 3) % 8, s = 8 * (tid / 64) + (tid & 7);
                        for (int i = 0; i < FS; ++i)
                            for (int kk = 0; kk < 16; ++kk)
                                for (int c = 0; c < 4; ++c) e += hW[(size_t)(kk * 64 + s) * LD + i * D + cu * DC + 4 * g + c];
                    } else if (tid >= 256) {
                        const int v = (tid >> 6) - 4, lane = tid & 63;
                        for (int i = 0; i < FS; ++i)
                            for (int j = 0; j < 32; ++j)
                                for (int c = 0; c < 4; ++c)
                                    e += hW[(size_t)(v * 256 + j * 8 + lane / 8) * LD + i * D + cu * DC + 4 * (lane % 8) + c];
                    }
                    e *= reps;
                    if ((double)ho[0].sums[x * 32 + cu][tid] != e) ++bad;
                }
        }
        std::vector<double> us;
        for (int w = 0; w < 256; ++w) us.push_back(ho[0].ticks[w] / 100.0 / (reps * FS));
        std::sort(us.begin(), us.end());
        printf("mode %d: %lld wrong sums; per 128 KB slice and CU: min %.2f median %.2f max %.2f us  -> %.0f GB/s per XCD, %.2f TB/s chip-wide (median)\n",
               mode, bad, us[0], us[128], us[255], 32 * 131072.0 / us[128] / 1e3, 256 * 131072.0 / us[128] / 1e6);
    }
    return 0;
}
