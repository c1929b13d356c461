// xcd_probe: cost and correctness of a hand-off that never leaves one XCD (MI355X: 8 XCDs x 32 CUs, private L2 each).
//
// 256 workgroups (one per CU) form 8 teams by HW_REG_XCC_ID.  Per iteration every workgroup publishes 512 B into its
// team's (re-used, parity double-buffered) exchange buffer, crosses a TEAM barrier and reads + verifies the team's
// 16 KB.  The SampleRNN sample-level kernel (samplernn.hip) is built on the cheapest variant that verifies clean.
//
//   variant 0: agent-scope protocol (release fence, agent atomics, acquire fence, plain loads) -- the safe baseline
//   variant 1: plain stores + vmcnt(0); arrival / poll = L2-executed atomics without sc1 (inline asm); TCP invalidate
//              (buffer_inv sc1); plain loads
//   variant 2: as 1, but no invalidate: payload loads carry sc1 (miss in TCP, served by the XCD's L2)
//   variant 3: as 1, but payload loads carry sc0 sc1
//
//   build: hipcc -O3 --offload-arch=gfx950 -o xcd_probe tools/xcd_probe.hip ; run: ./xcd_probe [iters=20000]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); exit(1); } } while (0)

struct Sync {
    unsigned census[8][32];
    unsigned arrive[8][32];
    unsigned abort_[32];
    unsigned errors[32];
    unsigned long long ticks[8][4];
};

__device__ __forceinline__ int xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return (int)(v & 7);
}
__device__ __forceinline__ unsigned l2_add(unsigned* p, unsigned v) {  // RMW executed in this XCD's L2, returns the old value
    unsigned old;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(old) : "v"(p), "v"(v) : "memory");
    return old;
}
template <int MODE>
__device__ __forceinline__ f32x4 ld16(const f32x4* p) {
    f32x4 v;
    if (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (MODE == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else v = *p;
    return v;
}

template <int V>
__global__ __launch_bounds__(512) void probe(Sync* s, f32x4* xbuf, int iters, int team_size) {
    __shared__ int sh_rank, sh_ok;
    const int tid = threadIdx.x, x = xcc_id();
    if (tid == 0) {
        sh_rank = (int)__hip_atomic_fetch_add(&s->census[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh_ok = 1;
    }
    __syncthreads();
    const int rank = sh_rank;
    if (rank >= team_size) return;  // (counted by the host through census)
    f32x4* team = xbuf + (size_t)x * 2 * team_size * 32;
    unsigned errs = 0;
    unsigned long long t0 = 0;
    if (tid == 0) t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        f32x4* buf = team + (size_t)(it & 1) * team_size * 32;
        if (tid < 32) {
            const float base = (float)(it * 64 + rank);
            buf[rank * 32 + tid] = (f32x4){base, base + 0.25f, (float)tid, base - 1.f};
        }
        if (V == 0) __threadfence();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const unsigned target = (unsigned)(it + 1) * team_size;
            unsigned spins = 0;
            if (V == 0) {
                __hip_atomic_fetch_add(&s->arrive[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(&s->arrive[x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    if (++spins > 4000000u || (spins % 1024 == 0 && __hip_atomic_load(&s->abort_[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { sh_ok = 0; break; }
                }
            } else {
                l2_add(&s->arrive[x][0], 1u);
                while (l2_add(&s->arrive[x][0], 0u) < target) {
                    if (++spins > 4000000u || (spins % 1024 == 0 && __hip_atomic_load(&s->abort_[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { sh_ok = 0; break; }
                }
            }
            if (!sh_ok) __hip_atomic_store(&s->abort_[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!sh_ok) break;
        if (V == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        else if (V == 1) asm volatile("buffer_inv sc1" ::: "memory");
        // consume: 32 * team_size vectors, two per thread at team_size = 32
        for (int j = tid; j < 32 * team_size; j += 512) {
            const f32x4 v = ld16<V>(buf + j);
            const int r = j >> 5, q = j & 31;
            const float base = (float)(it * 64 + r);
            if (v[0] != base || v[1] != base + 0.25f || v[2] != (float)q || v[3] != base - 1.f) ++errs;
        }
    }
    if (errs) atomicAdd(&s->errors[0], errs);
    if (tid == 0 && rank == 0) s->ticks[x][0] = wall_clock64() - t0;
}

template <int V>
static void run(Sync* s, f32x4* xbuf, int iters, const char* name) {
    CHECK(hipMemset(s, 0, sizeof(Sync)));
    CHECK(hipMemset(xbuf, 0, 8 * 2 * 32 * 32 * sizeof(f32x4)));
    hipLaunchKernelGGL(probe<V>, dim3(256), dim3(512), 0, 0, s, xbuf, iters, 32);
    CHECK(hipDeviceSynchronize());
    Sync h;
    CHECK(hipMemcpy(&h, s, sizeof(h), hipMemcpyDeviceToHost));
    double worst = 0;
    for (int x = 0; x < 8; ++x) worst = h.ticks[x][0] > worst ? (double)h.ticks[x][0] : worst;
    printf("%-70s %7.3f us/iter  errors %u  abort %u  census [", name, worst / 100.0 / iters, h.errors[0], h.abort_[0]);
    for (int x = 0; x < 8; ++x) printf("%u ", h.census[x][0]);
    printf("]\n");
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    Sync* s;
    f32x4* xbuf;
    CHECK(hipMalloc(&s, sizeof(Sync)));
    CHECK(hipMalloc(&xbuf, 8 * 2 * 32 * 32 * sizeof(f32x4)));
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(s, xbuf, iters, "0: agent fences + agent atomics + plain loads");
        run<1>(s, xbuf, iters, "1: vmcnt(0) + L2 atomics (no sc1) + buffer_inv sc1 + plain loads");
        run<2>(s, xbuf, iters, "2: vmcnt(0) + L2 atomics (no sc1) + sc1 loads");
        run<3>(s, xbuf, iters, "3: vmcnt(0) + L2 atomics (no sc1) + sc0 sc1 loads");
    }
    return 0;
}
