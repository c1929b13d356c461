// persist_probe: what does one phase of a persistent, weight-stationary decoder step cost on MI355X?
//
// Skeleton of the kernel VERDICT r01 item 3 asks for: NWG workgroups (one per CU) stay resident for the whole window;
// a phase = [consume the activation slab every workgroup published in the previous phase] -> [MFMA work on weights
// held in VGPRs] -> [publish this workgroup's slice of the new slab, write-through] -> [grid barrier].
// The probe times the pieces separately and together, and VERIFIES every consumed word (hand-offs are wrong-not-slow
// when the protocol is broken: MI355X_MICROARCH.md, inter-workgroup visibility).
//
//   build: hipcc -O3 --offload-arch=gfx950 -o persist_probe tools/persist_probe.hip
//   run:   ./persist_probe [phases=2400] [nwg=256]
//
// Protocol (guide G16, R1 without the fence): payload stores are 16-byte sc1 (write-through) buffer stores, every
// storing wave drains vmcnt before the workgroup barrier, one lane arrives on a monotonic counter (relaxed, agent);
// consumers poll relaxed, then read the payload with sc1 loads (L1 bypass; the slabs are write-once per launch, so no
// stale L2 line can exist).  Barrier = XCD-hierarchical (per-XCC arrival counter, last arriver of an XCC goes to the top
// counter, waits for all XCCs, then bumps the XCC's generation word) or flat (one counter).  Every spin is bounded by
// wall_clock64(); on timeout an abort word makes all workgroups leave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e__ = (x);                                                     \
        if (e__ != hipSuccess) {                                                  \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

struct Sync {
    unsigned xcnt[8][32];   // per-XCC arrival counters (128 B apart)
    unsigned xgen[8][32];   // per-XCC generation words
    unsigned top[32];
    unsigned flat[32];
    unsigned census[8][32];
    unsigned total[32];
    unsigned abort_[32];
    unsigned errors[32];
    unsigned long long t_wait[32];  // summed barrier wait of workgroup 0's lane 0 [100 MHz ticks]
};

#define RLX __ATOMIC_RELAXED
#define AGENT __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ unsigned ld_agent(unsigned* p) { return __hip_atomic_load(p, RLX, AGENT); }
__device__ __forceinline__ void st_agent(unsigned* p, unsigned v) { __hip_atomic_store(p, v, RLX, AGENT); }
__device__ __forceinline__ unsigned add_agent(unsigned* p, unsigned v) { return __hip_atomic_fetch_add(p, v, RLX, AGENT); }

__device__ __forceinline__ int xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return (int)(v & 7);
}

// one lane: spin until *p >= target; false on timeout / abort
__device__ __forceinline__ bool spin_ge(unsigned* p, unsigned target, Sync* s) {
    const unsigned long long t0 = wall_clock64();
    unsigned it = 0;
    while (ld_agent(p) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++it & 255) == 0) {
            if (ld_agent(&s->abort_[0])) return false;
            if (wall_clock64() - t0 > 20000000ull) {  // 200 ms
                st_agent(&s->abort_[0], 1u);
                return false;
            }
        }
    }
    return true;
}

struct BarCtx { int xcc; unsigned n_x, n_xcc, nwg; int hier; };

// All threads call it.  epoch = 1, 2, 3 ... (number of barriers including this one).
__device__ __forceinline__ bool grid_barrier(Sync* s, const BarCtx& c, unsigned epoch, unsigned long long* waited) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its write-through stores
    __syncthreads();
    __shared__ int ok_sh;
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        bool ok;
        if (c.hier) {
            const unsigned old = add_agent(&s->xcnt[c.xcc][0], 1u);
            if (old == c.n_x * epoch - 1u) {  // last arriver of this XCC
                add_agent(&s->top[0], 1u);
                ok = spin_ge(&s->top[0], c.n_xcc * epoch, s);
                st_agent(&s->xgen[c.xcc][0], epoch);
            } else {
                ok = spin_ge(&s->xgen[c.xcc][0], epoch, s);
            }
        } else {
            add_agent(&s->flat[0], 1u);
            ok = spin_ge(&s->flat[0], c.nwg * epoch, s);
        }
        if (waited) *waited += wall_clock64() - t0;
        ok_sh = ok ? 1 : 0;
    }
    __syncthreads();
    return ok_sh != 0;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// mode bits
#define M_HIER 1      // hierarchical barrier (else flat)
#define M_PUBLISH 2   // publish out_f4 x 16 B per thread... (per workgroup: OUTF4 vectors)
#define M_CONSUME 4   // read the whole previous slab (sc1 loads)
#define M_VERIFY 8    // check every consumed word
#define M_MFMA 16     // MFMA work with register-resident weights
#define M_PLAIN 32    // plain (non-sc1) loads + one agent acquire fence per phase instead of sc1 loads

// slab layout: phase p -> slab[p][wg][OUTF4] f32x4 ; value = f(p, wg, i)
template <int NWREG>
__global__ __launch_bounds__(512) void probe_kernel(Sync* s, f32x4* slab, int phases, int outf4, int mode, int nmfma,
                                                    float* sink, int cdiv) {
    const int tid = threadIdx.x, wg = blockIdx.x, nwg = gridDim.x;
    BarCtx c;
    c.xcc = xcc_id(); c.nwg = nwg; c.hier = mode & M_HIER;
    // census: how many workgroups sit on each XCC (placement is not architecturally defined)
    __shared__ unsigned cen[8];
    if (tid == 0) {
        add_agent(&s->census[c.xcc][0], 1u);
        add_agent(&s->total[0], 1u);
        spin_ge(&s->total[0], nwg, s);
        for (int x = 0; x < 8; ++x) cen[x] = ld_agent(&s->census[x][0]);
    }
    __syncthreads();
    c.n_x = cen[c.xcc];
    c.n_xcc = 0;
    for (int x = 0; x < 8; ++x) c.n_xcc += cen[x] ? 1 : 0;
    if (ld_agent(&s->abort_[0])) return;

    // register-resident "weights"
    float w[NWREG];
#pragma unroll
    for (int i = 0; i < NWREG; ++i) w[i] = 0.001f * (float)((tid * 31 + i * 7 + wg) % 97);
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};

    const size_t slab_f4 = (size_t)nwg * outf4;  // f32x4 per phase
    const size_t cons_f4 = slab_f4 / (size_t)cdiv;  // how much of it this workgroup consumes (row-split jobs read less)
    unsigned long long waited = 0;
    unsigned errs = 0;
    const unsigned slab_bytes = (unsigned)(slab_f4 * 16);
    for (int p = 0; p < phases; ++p) {
        f32x4 a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = (f32x4){1.f, 2.f, 3.f, 4.f};
        if ((mode & M_CONSUME) && p > 0) {
            const f32x4* src = slab + (size_t)(p - 1) * slab_f4;
            if (mode & M_PLAIN) {
                if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __syncthreads();
                for (size_t i = tid; i < cons_f4; i += 512 * 8) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const size_t j = i + (size_t)u * 512;
                        if (j < cons_f4) a[u] = src[j];
                    }
                    if (mode & M_VERIFY) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const size_t j = i + (size_t)u * 512;
                            if (j < cons_f4) {
                                const float want = (float)(((p - 1) * 131 + (int)(j / outf4) * 17 + (int)(j % outf4)) & 0xffff);
                                if (a[u][0] != want || a[u][3] != want + 3.f) ++errs;
                            }
                        }
                    }
                }
            } else {
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(src, slab_bytes);
                for (size_t i = tid; i < cons_f4; i += 512 * 8) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const size_t j = i + (size_t)u * 512;
                        const unsigned off = j < cons_f4 ? (unsigned)(j * 16) : 0u;
                        i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);  // aux 16 = sc1
                        a[u] = __builtin_bit_cast(f32x4, v);
                    }
                    if (mode & M_VERIFY) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const size_t j = i + (size_t)u * 512;
                            if (j < cons_f4) {
                                const float want = (float)(((p - 1) * 131 + (int)(j / outf4) * 17 + (int)(j % outf4)) & 0xffff);
                                if (a[u][0] != want || a[u][3] != want + 3.f) ++errs;
                            }
                        }
                    }
                }
            }
        }
        if (mode & M_MFMA) {
            // nmfma groups of (NWREG weights x 4 row blocks) MFMAs; A operands from the consumed data
            for (int r = 0; r < nmfma; ++r) {
#pragma unroll
                for (int i = 0; i < NWREG; ++i) {
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb)
                        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(i + rb) & 7][i & 3], w[i], acc[rb], 0, 0, 0);
                }
            }
        }
        if (mode & M_PUBLISH) {
            f32x4* dst = slab + (size_t)p * slab_f4 + (size_t)wg * outf4;
            const __amdgpu_buffer_rsrc_t rd = make_rsrc(dst, (unsigned)(outf4 * 16));
            for (int i = tid; i < outf4; i += 512) {
                const float base = (float)((p * 131 + wg * 17 + i) & 0xffff);
                f32x4 v = {base, base + 1.f, base + 2.f, base + 3.f};
                if (mode & M_MFMA) v[1] += acc[0][0] * 0.f;  // keep the MFMA chain live
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), rd, (unsigned)(i * 16), 0, 16);
            }
        }
        if (!grid_barrier(s, c, (unsigned)(p + 1), wg == 0 ? &waited : nullptr)) break;
    }
    if (errs) add_agent(&s->errors[0], errs);
    if (wg == 0 && tid == 0) s->t_wait[0] = waited;
    float r = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if (r == 12345.678f) sink[tid] = r;
}

int main(int argc, char** argv) {
    const int phases = argc > 1 ? atoi(argv[1]) : 2400;
    const int nwg = argc > 2 ? atoi(argv[2]) : 256;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    Sync* s;
    CHECK(hipMalloc(&s, sizeof(Sync)));
    float* sink;
    CHECK(hipMalloc(&sink, 4096));
    const int max_outf4 = 256;  // 4 KB per workgroup per phase
    f32x4* slab;
    const size_t slab_bytes = (size_t)phases * nwg * max_outf4 * 16;
    CHECK(hipMalloc(&slab, slab_bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    struct Case { const char* name; int mode; int outf4; int nmfma; int cdiv; };
    // MFMA sizing: a cfg2 decoder tick is 2688 MFMA 16x16x4 per CU; over 3 phases and 8 waves that is 112 per wave and
    // phase = 28 weight registers x 4 row blocks (nmfma = 1) = 3.0 us of f32 matrix-pipe time per phase.
    const std::vector<Case> cases = {
        {"flat barrier only", 0, 0, 0, 1},
        {"xcd barrier only", M_HIER, 0, 0, 1},
        {"xcd + publish 1KB/WG", M_HIER | M_PUBLISH, 64, 0, 1},
        {"xcd + publish 1KB + consume 256KB sc1 + VERIFY", M_HIER | M_PUBLISH | M_CONSUME | M_VERIFY, 64, 0, 1},
        {"xcd + publish 1KB + consume 256KB sc1", M_HIER | M_PUBLISH | M_CONSUME, 64, 0, 1},
        {"xcd + publish 1KB + consume 64KB sc1", M_HIER | M_PUBLISH | M_CONSUME, 64, 0, 4},
        {"xcd + publish 1KB + consume 256KB plain+acquire + VERIFY", M_HIER | M_PUBLISH | M_CONSUME | M_VERIFY | M_PLAIN, 64, 0, 1},
        {"xcd + publish 1KB + consume 256KB plain+acquire", M_HIER | M_PUBLISH | M_CONSUME | M_PLAIN, 64, 0, 1},
        {"xcd + publish 4KB + consume 1MB sc1", M_HIER | M_PUBLISH | M_CONSUME, 256, 0, 1},
        {"flat + publish 1KB + consume 256KB sc1", M_PUBLISH | M_CONSUME, 64, 0, 1},
        {"xcd + mfma 28x4/wave (3.0 us pipe time)", M_HIER | M_MFMA, 0, 1, 1},
        {"xcd + publish + consume 256KB + mfma 28x4", M_HIER | M_PUBLISH | M_CONSUME | M_MFMA, 64, 1, 1},
        {"xcd + publish + consume 64KB + mfma 28x4", M_HIER | M_PUBLISH | M_CONSUME | M_MFMA, 64, 1, 4},
        {"xcd + publish + consume 256KB + mfma 56x4 (6 us)", M_HIER | M_PUBLISH | M_CONSUME | M_MFMA, 64, 2, 1},
    };
    for (const Case& cs : cases) {
        CHECK(hipMemset(s, 0, sizeof(Sync)));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((probe_kernel<28>), dim3(nwg), dim3(512), 0, 0, s, slab, phases, cs.outf4, cs.mode, cs.nmfma,
                           sink, cs.cdiv);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipGetLastError());
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        Sync h;
        CHECK(hipMemcpy(&h, s, sizeof(Sync), hipMemcpyDeviceToHost));
        int nx = 0;
        for (int x = 0; x < 8; ++x) nx += h.census[x][0] ? 1 : 0;
        printf("%-62s %8.3f us/phase  wait(wg0) %6.3f us/phase  errors %u  abort %u  xccs %d [%u %u %u %u %u %u %u %u]\n",
               cs.name, 1000.0 * ms / phases, (double)h.t_wait[0] / 100.0 / phases, h.errors[0], h.abort_[0], nx,
               h.census[0][0], h.census[1][0], h.census[2][0], h.census[3][0], h.census[4][0], h.census[5][0],
               h.census[6][0], h.census[7][0]);
        fflush(stdout);
        if (h.abort_[0]) {
            printf("ABORTED (timeout in a spin): stopping\n");
            break;
        }
    }
    // reference: the same number of trivial dependent kernel launches (kernel boundary cost)
    return 0;
}
