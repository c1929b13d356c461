// Persistent phase machine for the decoder scan (see persist.h).  gfx950 only.
#include "persist.h"

#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

typedef int i32x4 __attribute__((ext_vector_type(4)));

// 16-deep K chunks in flight per wave (1 KB of activations + 1 KB of weights each).  The sc1 activation loads miss the
// XCD's L2 by construction (another XCD wrote them through), ~0.5 us per round trip: with 4 in flight a [64, 1280]
// operand took ~10 round trips; 12 keeps 96 KB per CU in flight.
#ifndef PM_DEPTH
#define PM_DEPTH 16
#endif
#ifndef PM_SDEPTH
#define PM_SDEPTH 12  // streamed units: activation and weight chunks both come from global memory
#endif
#ifndef PM_WDEPTH
#define PM_WDEPTH 4
#endif

namespace {

#define PM_RLX __ATOMIC_RELAXED
#define PM_AGENT __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ unsigned pm_ld(unsigned* p) { return __hip_atomic_load(p, PM_RLX, PM_AGENT); }
__device__ __forceinline__ void pm_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, PM_RLX, PM_AGENT); }
__device__ __forceinline__ unsigned pm_add(unsigned* p, unsigned v) { return __hip_atomic_fetch_add(p, v, PM_RLX, PM_AGENT); }

// 4-byte agent-scope (write-through / L1-bypassing) accesses for the few scalar hand-offs (kappa, w fragments)
__device__ __forceinline__ float pm_ldf(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), PM_RLX, PM_AGENT));
}
__device__ __forceinline__ void pm_stf(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), PM_RLX, PM_AGENT);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t pm_rsrc(const void* p) {
    // raw buffer over [p, p + 2 GB); p is wave-uniform
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* q = (void*)(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, 0x7fffffff, 0x00020000);
}
// 16-byte sc1 (agent-coherent) load / store: aux bit 4 = sc1 on gfx950
__device__ __forceinline__ f32x4 pm_ld16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));
}
__device__ __forceinline__ void pm_st16(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), r, byte_off, 0, 16);
}

// 100 MHz wall clock, not reorderable (the builtin has no side effects as far as the optimiser is concerned)
__device__ __forceinline__ unsigned long long pm_clock() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

__device__ __forceinline__ int pm_xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return (int)(v & 7);
}

// one lane: wait until *p >= target; gives up after ~200 ms (sets the abort word) or when somebody else did
__device__ __forceinline__ bool pm_spin_ge(unsigned* p, unsigned target, unsigned* sync) {
    const unsigned long long t0 = wall_clock64();
    unsigned it = 0;
    while (pm_ld(p) < target) {
        // back off: idle workgroups (decode: most of them in every phase) must not hammer the counter's L2 channel
        if (it < 16) __builtin_amdgcn_s_sleep(1);
        else __builtin_amdgcn_s_sleep(6);
        if ((++it & 255) == 0) {
            if (pm_ld(sync + PM_S_ABORT)) return false;
            if (wall_clock64() - t0 > 20000000ull) {
                pm_st(sync + PM_S_ABORT, 1u);
                pm_st(sync + PM_S_STICKY, 1u);  // the word pm_status reads: survives the next launch's clearing
                                                // (value = the site that gave up: 1 spin, 2 census, 3 first / 4 phase barrier, 5 slot poll)
                return false;
            }
        }
    }
    return true;
}

struct PmBar {
    int xcc;
    unsigned n_x, n_xcc;
};

// XCD-hierarchical grid barrier.  Payload protocol: every wave has drained its write-through (sc1) stores before the
// workgroup barrier; one lane arrives; consumers read the payload with sc1 loads afterwards (no fences needed: the
// payload buffers are write-once per launch, so no cache can hold a stale copy).  epoch = 1, 2, ...
__device__ __forceinline__ bool pm_barrier(unsigned* sync, const PmBar& c, unsigned epoch, int* ok_sh) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        bool ok;
        const unsigned old = pm_add(sync + PM_S_XCNT + c.xcc * 32, 1u);
        if (old == c.n_x * epoch - 1u) {  // last arriver of this XCC: on to the top counter
            pm_add(sync + PM_S_TOP, 1u);
            ok = pm_spin_ge(sync + PM_S_TOP, c.n_xcc * epoch, sync);
            pm_st(sync + PM_S_XGEN + c.xcc * 32, epoch);
        } else {
            ok = pm_spin_ge(sync + PM_S_XGEN + c.xcc * 32, epoch, sync);
        }
        *ok_sh = ok ? 1 : 0;
    }
    __syncthreads();
    return *ok_sh != 0;
}

__device__ __forceinline__ bool pm_is_empty(const f32x4& v) {
    return __float_as_uint(v[0]) == PM_EMPTY || __float_as_uint(v[3]) == PM_EMPTY;
}
// somebody waited ~1 s for a slot that never filled: everybody leaves, the host sees the sticky word
__device__ __forceinline__ void pm_give_up(unsigned* sync) {
    pm_st(sync + PM_S_ABORT, 1u);
    pm_st(sync + PM_S_STICKY, 5u);
}
constexpr unsigned PM_POLL_LIMIT = 1u << 21;

__device__ __forceinline__ f32x4 pm_rm_load(const PmRM& o, int t, int m, int n0) {
    // 16 B of row m at columns n0..n0+3 (sc1: written by another workgroup in an earlier phase)
    const float* p = o.p + (long long)t * o.st + (long long)m * o.ld + n0;
    const unsigned long long a = (unsigned long long)p;
    // per-lane address: use a global sc1 load through the flat-address buffer trick is not possible -> two 8-byte
    // agent-scope loads (L1-bypassing); these operands are a few KB per unit
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(a);
    const unsigned long long x0 = __hip_atomic_load(q, PM_RLX, PM_AGENT);
    const unsigned long long x1 = __hip_atomic_load(q + 1, PM_RLX, PM_AGENT);
    f32x4 v;
    v[0] = __uint_as_float((unsigned)x0); v[1] = __uint_as_float((unsigned)(x0 >> 32));
    v[2] = __uint_as_float((unsigned)x1); v[3] = __uint_as_float((unsigned)(x1 >> 32));
    return v;
}
// dataflow mode: the same operand, re-read until its producer's 16-byte store has landed (a torn pair of 8-byte loads
// shows EMPTY in one of the two checked words)
__device__ __forceinline__ f32x4 pm_rm_take(const PmRM& o, int t, int m, int n0, unsigned* sync) {
    f32x4 v = pm_rm_load(o, t, m, n0);
    unsigned n = 0;
    while (pm_is_empty(v)) {
        if ((++n & 1023u) == 0u) {
            if (pm_ld(sync + PM_S_ABORT)) break;
            if (n > PM_POLL_LIMIT) { pm_give_up(sync); break; }
        }
        __builtin_amdgcn_s_sleep(1);
        v = pm_rm_load(o, t, m, n0);
    }
    return v;
}
// 16-byte row-major stores.  WT = true: write-through (sc1) for values another workgroup reads later in this launch
// (z, h_new, pre-activation partials); false: plain store for values only the backward pass / the host side reads.
// Issued as inline asm (no C++ construct yields a 16-byte sc1 global store); every phase ends with an explicit
// s_waitcnt vmcnt(0) before the barrier, which is what orders them.
template <bool WT>
__device__ __forceinline__ void pm_rm_store(const PmRM& o, int t, int m, int n0, f32x4 v) {
    float* p = o.p + (long long)t * o.st + (long long)m * o.ld + n0;
    if (WT) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else *reinterpret_cast<f32x4*>(p) = v;
}

__device__ __forceinline__ f32x4 pm_sigmoid4(f32x4 x) {
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = ph_sigmoid(x[i]);
    return r;
}

// ------------------------------------------------------------------------------------------------ GEMM unit
// 8 waves = MB row blocks x KS = 8/MB contiguous K ranges.  Wave (rb, ks) owns row block rb of the 16-column tile
// over its K range: per 16-deep chunk one fragment-major A block (1 KB, sc1 global load) and one B block (1 KB from
// LDS, shared by the MB waves of the same ks; or a global load when the slab is streamed) feed 4 MFMA 16x16x4.
template <int MB, bool DF>
__device__ __forceinline__ void pm_gemm(const PmUnit& u, int t, const float* lds_w, float* lds_red,
                                        const __amdgpu_buffer_rsrc_t fm, unsigned long long* stage, unsigned* sync) {
    const unsigned long long ts0 = pm_clock();
    constexpr int KS = 8 / MB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rb = wave % MB, ks = wave / MB;

    const int total = __builtin_amdgcn_readfirstlane(u.K >> 4);  // 16-deep chunks of the unit's slab
    const unsigned abase = __builtin_amdgcn_readfirstlane(u.a_off + (unsigned)t * u.a_st) +
                           ((unsigned)(rb * __builtin_amdgcn_readfirstlane(u.a_nch) + __builtin_amdgcn_readfirstlane(u.a_c0)) << 10);
    const int c0 = (ks * total) / KS, c1 = ((ks + 1) * total) / KS;
    const bool resident = u.w_lds >= 0;
    const float* wl = lds_w + (resident ? u.w_lds : 0);  // stays an LDS (address space 3) pointer
    // global (address space 1) pointer: the descriptor lives in LDS, so the compiler cannot infer it
    typedef const __attribute__((address_space(1))) f32x4* gptr4;
    gptr4 wg = (gptr4)(unsigned long long)u.W;

    // finalising waves (ks == 0) request their epilogue operands now: they were published in earlier phases
    const bool fin = wave < MB;
    const int kk = lane >> 4, r16 = lane & 15;
    const int m = 16 * rb + r16, n0 = 4 * kk;
    const bool row_ok = m < u.M;
    f32x4 p_bias = {0.f, 0.f, 0.f, 0.f}, p_add = {0.f, 0.f, 0.f, 0.f}, p_e0 = {0.f, 0.f, 0.f, 0.f},
          p_e1 = {0.f, 0.f, 0.f, 0.f};
    f32x4 p_in[4];  // the additive inputs, kept apart until the epilogue (dataflow mode may have to ask again)
#pragma unroll
    for (int q = 0; q < 4; ++q) p_in[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // (round 5) layer 0's candidate units of the decode machine: the two projection-matrix blocks of the fold, asked for now
    const bool fold = MB == 1 && fin && __builtin_amdgcn_readfirstlane(u.pw[0] != nullptr);
    f32x4 pwb0 = {0.f, 0.f, 0.f, 0.f}, pwb1 = {0.f, 0.f, 0.f, 0.f};
    if (fold) {
        pwb0 = *reinterpret_cast<const f32x4*>(u.pw[0] + (lane << 2));
        pwb1 = *reinterpret_cast<const f32x4*>(u.pw[1] + (lane << 2));
    }
    if (fin && row_ok) {
        if (u.bias) p_bias = *reinterpret_cast<const f32x4*>(u.bias + n0);
        // requested NOW in both modes: these operands come from phases before the one that produced the unit's A operand,
        // so they have almost always landed; asked for behind the K loop, one poll loop after the other, they cost a
        // memory round trip EACH on the step's critical chain (dataflow mode, round 4)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (u.add[q].p) p_in[q] = pm_rm_load(u.add[q], t, m, n0);
        if (u.e0.p) p_e0 = pm_rm_load(u.e0, t, m, n0);
        if (u.e1.p) p_e1 = pm_rm_load(u.e1, t, m, n0);
    }

    const unsigned long long ts1 = pm_clock();
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    auto loadA = [&](int c) -> f32x4 {  // block (rb, c): scalar offset + lane * 16
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(fm, (unsigned)lane << 4,
                                                                               abase + ((unsigned)c << 10), 16));
    };
    auto mma = [&](const f32x4& a, const f32x4& b, f32x4& acc) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[q], acc, 0, 0, 0);
    };
    // dataflow: block c was requested earlier; while any lane still holds an EMPTY slot, ask again (wave-uniform loop)
    auto takeA = [&](f32x4& a, int c) {
        if (!DF) return;
        unsigned n = 0;
        // (lanes of padding rows are never waited for: the attention units only write the rows that exist)
        while (__builtin_amdgcn_ballot_w64(row_ok && pm_is_empty(a)) != 0ull) {
            if ((++n & 1023u) == 0u) {
                if (pm_ld(sync + PM_S_ABORT)) break;
                if (n > PM_POLL_LIMIT) { pm_give_up(sync); break; }
            }
            __builtin_amdgcn_s_sleep(1);
            a = loadA(c);
        }
    };
    // two copies of the K loop: weights from LDS (ds_read_b128) or streamed from global memory.  One loop with a
    // run-time pointer would turn both into flat loads (conservative waits on both counters).
    auto run = [&](auto res_tag) {
        constexpr bool RES = decltype(res_tag)::value;
        auto loadB = [&](int c) -> f32x4 {
            if (RES) return *reinterpret_cast<const f32x4*>(wl + ((size_t)c << 8) + (lane << 2));
            return __builtin_nontemporal_load(wg + ((size_t)c << 6) + lane);
        };
        // activation ring: PM_DEPTH chunks in flight (global, ~1 us away); weight ring: PM_WDEPTH (LDS is close)
        constexpr int D = RES ? PM_DEPTH : PM_SDEPTH, DB = RES ? PM_WDEPTH : PM_SDEPTH;
        static_assert(D % DB == 0, "");
        f32x4 ra[D], rbv[DB];
        const int last = c1 - 1;
#pragma unroll
        for (int d = 0; d < D; ++d) ra[d] = loadA(min(c0 + d, last));
#pragma unroll
        for (int d = 0; d < DB; ++d) rbv[d] = loadB(min(c0 + d, last));
        if (DF) {
            // A wave that got here before its operand sees EMPTY in every ring entry.  Waiting for them one after the other
            // (takeA below) would cost a round trip per chunk: wait for the first, then ask for all the others again at once
            bool waited = false;
            unsigned n = 0;
            while (__builtin_amdgcn_ballot_w64(row_ok && pm_is_empty(ra[0])) != 0ull) {
                waited = true;
                if ((++n & 1023u) == 0u) {
                    if (pm_ld(sync + PM_S_ABORT)) break;
                    if (n > PM_POLL_LIMIT) { pm_give_up(sync); break; }
                }
                __builtin_amdgcn_s_sleep(1);
                ra[0] = loadA(c0);
            }
            if (waited) {
#pragma unroll
                for (int d = 1; d < D; ++d) ra[d] = loadA(min(c0 + d, last));
            }
        }
        int c = c0;
        for (; c + D <= c1; c += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                takeA(ra[d], c + d);
                mma(ra[d], rbv[d % DB], (d & 1) ? acc1 : acc0);
                ra[d] = loadA(min(c + D + d, last));
                rbv[d % DB] = loadB(min(c + DB + d, last));
            }
        }
#pragma unroll
        for (int d = 0; d < D - 1; ++d)
            if (c + d < c1) {
                takeA(ra[d], c + d);
                mma(ra[d], rbv[d % DB], (d & 1) ? acc1 : acc0);
                rbv[d % DB] = loadB(min(c + DB + d, last));
            }
    };
    if (c1 > c0) {
        if (resident) run(std::true_type{});
        else run(std::false_type{});
    }
    const f32x4 part = acc0 + acc1;
    if (DF && fin && row_ok) {  // whatever had not landed when it was first asked for: poll it now
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (u.add[q].p && pm_is_empty(p_in[q])) p_in[q] = pm_rm_take(u.add[q], t, m, n0, sync);
        if (u.e0.p && pm_is_empty(p_e0)) p_e0 = pm_rm_take(u.e0, t, m, n0, sync);
        if (u.e1.p && pm_is_empty(p_e1)) p_e1 = pm_rm_take(u.e1, t, m, n0, sync);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) p_add += p_in[q];  // (fixed order, both modes)
    const unsigned long long ts2 = pm_clock();

    // split-K reduction through LDS: C layout (col = lane & 15, row = 4 (lane >> 4) + reg) -> [row][col] tiles, rows
    // padded to 20 floats; the finalising wave reads 16-byte row quads = the fragment-major lane order
    {
        float* dst = lds_red + (ks * MB + rb) * 320;
        const int g = lane >> 4, jj = lane & 15;
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[(4 * g + i) * 20 + jj] = part[i];
    }
    __syncthreads();
    const unsigned long long ts3 = pm_clock();
    if (fin) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < KS; ++q)
            v += *reinterpret_cast<const f32x4*>(lds_red + (q * MB + rb) * 320 + r16 * 20 + n0);
        f32x4 fmv = {0.f, 0.f, 0.f, 0.f};  // value published to the consumers (zero in the padding rows)
        if (row_ok) {
            const f32x4 pre = v + p_bias + p_add;
            if (u.epi == PM_EPI_LINEAR) {
                fmv = pre;
                pm_rm_store<true>(u.out, t, m, n0, pre);
            } else if (u.epi == PM_EPI_GATES) {
                const f32x4 gt = pm_sigmoid4(pre);
                if (!u.rtile) {
                    pm_rm_store<true>(u.o1, t, m, n0, gt);  // update gate z (read by the candidate units)
                } else {
                    if (u.o2.p) pm_rm_store<false>(u.o2, t, m, n0, gt);  // reset gate r (backward only)
                    fmv = gt * p_e0;                   // r * h_prev
                    if (u.out.p) pm_rm_store<false>(u.out, t, m, n0, fmv);  // row-major r*h: backward only
                }
            } else {  // PM_EPI_CAND
                f32x4 c;
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = tanhf(pre[i]);
                const f32x4 one = {1.f, 1.f, 1.f, 1.f};
                fmv = p_e1 * c + (one - p_e1) * p_e0;  // z*c + (1-z)*h_prev
                if (u.o1.p) pm_rm_store<false>(u.o1, t, m, n0, c);
                pm_rm_store<true>(u.out, t, m, n0, fmv);  // h_new: next step's epilogues and the attention read it
            }
        }
        const int ndst = __builtin_amdgcn_readfirstlane(u.ndst);
#pragma unroll
        for (int q = 0; q < PM_MAXDST; ++q) {
            if (q < ndst) {
                const PmDst ds = u.dst[q];
                const unsigned so = ds.off + (unsigned)t * ds.st + ((unsigned)(rb * ds.nch + ds.chunk) << 10);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, fmv), fm, (unsigned)lane << 4,
                                                       __builtin_amdgcn_readfirstlane(so), 16);
            }
        }
        if (fold) {
            // fmv IS the A operand of the tile (rows = batch, k = the unit's 16 state columns; zero in the padding rows):
            // partial[b][j] = sum_k h_new[b][16 ct + k] * Watt[16 ct + k][j], j < 32, as two 16 x 16 output tiles
            f32x4 pa0 = {0.f, 0.f, 0.f, 0.f}, pa1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                pa0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fmv[q], pwb0[q], pa0, 0, 0, 0);
                pa1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fmv[q], pwb1[q], pa1, 0, 0, 0);
            }
            // C layout (col = lane & 15, row = 4 (lane >> 4) + i) -> [row][32] through LDS (this wave has read all of
            // lds_red's partial tiles above; LDS operations of a wave execute in order), then 16-byte write-through stores
            float* tr = lds_red;  // 16 rows x 36 floats
            const int g4 = lane >> 4, jc = lane & 15;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                tr[(4 * g4 + i) * 36 + jc] = pa0[i];
                tr[(4 * g4 + i) * 36 + 16 + jc] = pa1[i];
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int qd = lane + 64 * s2, prow = qd >> 3, pc4 = qd & 7;
                if (prow < u.M) {
                    const f32x4 pv = *reinterpret_cast<const f32x4*>(tr + prow * 36 + 4 * pc4);
                    pm_rm_store<true>(u.pp, t, prow, 4 * pc4, pv);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long ts4 = pm_clock();
    stage[0] += ts1 - ts0; stage[1] += ts2 - ts1; stage[2] += ts3 - ts2; stage[3] += ts4 - ts3;
    __syncthreads();  // lds_red is reused by the next unit
}

// ------------------------------------------------------------------------------------------------ attention row
// GMM-window attention of batch row b at step t (model.py:664-690) by the whole workgroup (512 threads):
// projection h1 . Watt, window parameters, phi over the context, w = sum_u phi[u] ctx[b,u,:] over the support of the
// window.  Same formulas as att_fwd_kernel (attention.hip); the summation order over u differs (two row groups).
template <bool DF>
__device__ __forceinline__ void pm_att_row(const PmAtt& g, int b, int t, float* sm, float* fm_base, unsigned* sync) {
    const int A = g.A, U = g.U, E = g.E, H = g.H;
    float* s_p = sm;                        // [3A]
    float* s_a = s_p + 3 * PM_ATT_MAXA;     // [A]
    float* s_b = s_a + PM_ATT_MAXA;
    float* s_k = s_b + PM_ATT_MAXA;
    float* s_red = s_k + PM_ATT_MAXA;       // [16]
    float* s_phi = s_red + 16;              // [U]
    float* s_acc = s_phi + ((U + 3) & ~3);  // [512]
    float* s_stage = s_acc + PM_THREADS;    // [<= 512] one column block of w, staged for 16-byte stores
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* h = g.h1.p + (long long)(t + 1) * g.h1.st + (long long)b * g.h1.ld;
    const size_t BA = (size_t)g.B * A;

    // kappa[t] of this row was written a step ago (by this workgroup, or by the host side for t = 0): ask for it now, it
    // is needed behind the projection
    float kp_pre = 0.f;
    if (tid < A) kp_pre = pm_ldf(g.kappa + (size_t)t * BA + (size_t)b * A + tid);
    // 1) projection.  (round 5) with the fold: the H / 16 partial sums the candidate units of this step published
    if (g.pp) {
        const int nct = H >> 4;
        const float* base = g.pp + (long long)t * g.pp_st + (long long)b * 32;
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        const int c4 = tid & 7;
        for (int ct0 = 0; ct0 < nct; ct0 += PM_THREADS / 8) {  // 64 tiles per pass: one 16-byte slot per thread
            const int ct = ct0 + (tid >> 3);
            if (ct < nct) {
                PmRM o;
                o.p = const_cast<float*>(base) + (long long)ct * g.B * 32; o.st = 0; o.ld = 32; o.pad = 0;
                f32x4 v = pm_rm_load(o, 0, 0, 4 * c4);
                if (DF && pm_is_empty(v)) v = pm_rm_take(o, 0, 0, 4 * c4, sync);
                sum += v;  // (tiles ct0 + k, k fixed per thread: ascending order)
            }
        }
        // lanes of a wave: 8 tiles x 8 column quads -> add the 8 tiles (lane bits 3..5), then the 8 waves through LDS
#pragma unroll
        for (int sh = 8; sh <= 32; sh <<= 1)
#pragma unroll
            for (int i = 0; i < 4; ++i) sum[i] += __shfl_xor(sum[i], sh, 64);
        float* s_w = s_acc;  // [8 waves][32] (s_acc is not live before the weighted sum)
        if (lane < 8) *reinterpret_cast<f32x4*>(s_w + wave * 32 + 4 * lane) = sum;
        __syncthreads();
        if (tid < 3 * A) {
            float r = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < PM_THREADS / 64; ++w2) r += s_w[w2 * 32 + tid];
            s_p[tid] = r + (g.batt ? g.batt[tid] : 0.f);
        }
    } else {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const __amdgpu_buffer_rsrc_t hr = pm_rsrc(h);
        for (int k0 = 4 * lane; k0 < H; k0 += 1024) {
            // four blocks of the row in flight: in dataflow mode every block is polled, and one poll loop after the other
            // would cost a round trip per block
            f32x4 hq[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = k0 + 256 * i;
                hq[i] = k < H ? pm_ld16(hr, (unsigned)k << 2) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            // the projection weights do not depend on h: ask for them before the first poll, not behind it
            f32x4 wq[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k = k0 + 256 * i, j = wave + 8 * q;
                    wq[i][q] = (k < H && j < 3 * A) ? *reinterpret_cast<const f32x4*>(g.WattT + (size_t)j * H + k)
                                                    : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = k0 + 256 * i;
                if (k >= H) continue;
                f32x4 hv = hq[i];
                if (DF) {
                    unsigned n = 0;
                    while (pm_is_empty(hv)) {
                        if ((++n & 1023u) == 0u) {
                            if (pm_ld(sync + PM_S_ABORT)) break;
                            if (n > PM_POLL_LIMIT) { pm_give_up(sync); break; }
                        }
                        __builtin_amdgcn_s_sleep(1);
                        hv = pm_ld16(hr, (unsigned)k << 2);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int j = wave + 8 * q;
                    if (j < 3 * A) {
                        const f32x4 wv = wq[i][q];
                        acc[q] += hv[0] * wv[0] + hv[1] * wv[1] + hv[2] * wv[2] + hv[3] * wv[3];
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float r = wave_sum(acc[q]);
            const int j = wave + 8 * q;
            if (lane == 0 && j < 3 * A) s_p[j] = r + (g.batt ? g.batt[j] : 0.f);
        }
    }
    __syncthreads();
    // 2) window parameters
    if (g.att_type == 1) {
        if (tid == 0) {
            float mx = -INFINITY;
            for (int j = 0; j < A; ++j) mx = fmaxf(mx, s_p[j]);
            float s = 0.f;
            for (int j = 0; j < A; ++j) s += expf(s_p[j] - mx);
            s_red[4] = mx;
            s_red[5] = s;
        }
        __syncthreads();
    }
    if (tid < A) {
        float av;
        if (g.att_type == 1) av = expf(s_p[tid] - s_red[4]) / s_red[5] + g.eps;
        else av = expf(s_p[tid]) + g.eps;
        const float bv = expf(s_p[A + tid]) * g.sharpening + g.eps;
        const float kp = kp_pre;
        const float kv = kp + g.alignment * expf(s_p[2 * A + tid]) / g.timing;
        s_a[tid] = av; s_b[tid] = bv; s_k[tid] = kv;
        g.a[(size_t)t * BA + (size_t)b * A + tid] = av;
        g.b[(size_t)t * BA + (size_t)b * A + tid] = bv;
        pm_stf(g.kappa + (size_t)(t + 1) * BA + (size_t)b * A + tid, kv);
    }
    __syncthreads();
    // 3) phi.  Round 6 (as att_fwd_body.h): the support [lo, hi] = positions whose phi is not exactly zero from wave ballots
    // and one LDS word per wave instead of two LDS atomics per position on one address.
    float* phi_out = g.phi + ((size_t)t * g.B + b) * U;
    int w_lo = U, w_hi = -1;  // this wave's support (wave-uniform)
    for (int base = 0; base < U; base += PM_THREADS) {
        const int u = base + tid;
        float ph = 0.f;
        const float uf = (float)u;
        if (g.att_type == 1) {
            for (int j = 0; j < A; ++j) {
                const float d = s_k[j] - uf;
                ph += s_a[j] * sqrtf(s_b[j]) * expf(-0.5f * s_b[j] * d * d);
            }
            ph *= 0.3989422917366028f;
        } else {
            for (int j = 0; j < A; ++j) {
                const float d = s_k[j] - uf;
                ph += s_a[j] * expf(-s_b[j] * d * d);
            }
        }
        if (u < U) {
            s_phi[u] = ph;
            phi_out[u] = ph;
        }
        const unsigned long long nz = __ballot(u < U && ph != 0.f);
        if (nz) {
            w_lo = min(w_lo, base + wave * 64 + __ffsll((long long)nz) - 1);
            w_hi = max(w_hi, base + wave * 64 + 63 - __clzll((long long)nz));
        }
    }
    if (lane == 0) {
        reinterpret_cast<int*>(s_acc)[wave] = w_lo;
        reinterpret_cast<int*>(s_acc)[PM_THREADS / 64 + wave] = w_hi;
    }
    __syncthreads();
    int u_lo = 0, u_hi = U - 1;
    {
        int s_lo = U, s_hi = -1;
#pragma unroll
        for (int q = 0; q < PM_THREADS / 64; ++q) {
            s_lo = min(s_lo, reinterpret_cast<const int*>(s_acc)[q]);
            s_hi = max(s_hi, reinterpret_cast<const int*>(s_acc)[PM_THREADS / 64 + q]);
        }
        if (!g.dense) {
            u_lo = s_lo;
            u_hi = s_hi;
        }
    }
    if (g.sup && tid == 0) {
        g.sup[((size_t)t * g.B + b) * 2] = u_lo;
        g.sup[((size_t)t * g.B + b) * 2 + 1] = u_hi;
    }
    // 4) w[e] = sum_u phi[u] ctx[b,u,e] over the support: CW columns x G row groups per pass
    int CW = 1;
    while (CW < E && CW < PM_THREADS) CW <<= 1;
    const int G = PM_THREADS / CW;
    const int c = tid % CW, ug = tid / CW;
    const float* ctx = g.ctx + (size_t)b * U * E;
    float* w_rm = g.w + ((size_t)(t + 1) * g.B + b) * E;
    for (int eb = 0; eb < E; eb += CW) {
        const int e = eb + c;
        float acc = 0.f;
        if (e < E && u_lo <= u_hi) {
            // row group ug takes u = ug (mod G), as in the walk over all rows: skipping the rows with phi == 0 then
            // leaves every partial sum bit-identical (PARROT_ATT_DENSE=1 reads them all)
            int u = u_lo + ((ug - u_lo % G + G) % G);
            for (; u + 7 * G <= u_hi; u += 8 * G) {  // eight rows in flight (a decode row group walks the whole support)
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = ctx[(size_t)(u + q * G) * E + e];
#pragma unroll
                for (int q = 0; q < 8; ++q) acc = __builtin_fmaf(s_phi[u + q * G], v[q], acc);
            }
            {   // the remaining (< 8) rows in ONE batch of clamped, unconditional loads (row after row each was a dependent
                // round trip): same terms, same order
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = ctx[(size_t)min(u + q * G, u_hi) * E + e];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (u + q * G <= u_hi) acc = __builtin_fmaf(s_phi[u + q * G], v[q], acc);
            }
        }
        __syncthreads();
        s_acc[tid] = acc;
        __syncthreads();
        if (ug == 0 && e < E) {
            float s = 0.f;
            for (int q = 0; q < G; ++q) s += s_acc[q * CW + c];
            s_stage[c] = s;
        }
        __syncthreads();
        // 16-byte stores of the block's columns: row-major w, and one piece per fragment-major destination
        // (block (b / 16, chunk + e / 16), lane (e % 16) / 4 * 16 + b % 16 holds columns e .. e + 3)
        const int ncol = min(CW, E - eb);
        if (4 * tid < ncol) {
            const int e4 = eb + 4 * tid;
            const f32x4 v = *reinterpret_cast<const f32x4*>(s_stage + 4 * tid);
            *reinterpret_cast<f32x4*>(w_rm + e4) = v;
            for (int q = 0; q < g.nwdst && q < PM_MAXWDST; ++q) {
                const PmDst ds = g.wdst[q];
                float* slab = fm_base + ((size_t)ds.off + (size_t)t * ds.st) / 4;
                float* p = slab + (((size_t)((b >> 4) * ds.nch + ds.chunk + (e4 >> 4))) << 8) +
                           ((((e4 & 15) >> 2) * 16 + (b & 15)) << 2);
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
            }
        }
        __syncthreads();
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------ kernel
template <int MB, bool DF>
__global__ __launch_bounds__(PM_THREADS) void pm_kernel(const PmProgram P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* lds_w = lds;
    float* lds_red = lds + PM_LDS_W;
    float* lds_att = lds_red;  // the attention row reuses the reduction scratch
    // the workgroup's unit descriptors: read once from global memory, then from LDS every phase
    PmUnit* lds_units = reinterpret_cast<PmUnit*>(lds_red + PM_LDS_RED);
    static_assert(sizeof(PmUnit) * PM_MAXENT <= PM_LDS_UNITS * sizeof(float), "unit table does not fit");
    const int n_slots = P.n_slots, maxu = P.maxu;
    // all scratch lives in the dynamic region (a static __shared__ would push the total over the 160 KB limit)
    unsigned* cen = reinterpret_cast<unsigned*>(lds_red + PM_LDS_RED + PM_LDS_UNITS);
    int& ok_sh = *reinterpret_cast<int*>(lds_red + PM_LDS_RED + PM_LDS_UNITS + 16);
    const int tid = threadIdx.x, wg = blockIdx.x, nwg = gridDim.x;
    const int lane = tid & 63, wave = tid >> 6;
    unsigned* sync = P.sync;
    const __amdgpu_buffer_rsrc_t fmr = __builtin_amdgcn_make_buffer_rsrc(P.fm_base, 0, 0xffffffff, 0x00020000);

    // census of the workgroup -> XCC placement (not architecturally defined), then the first rendezvous
    PmBar bar;
    bar.xcc = pm_xcc_id();
    bar.n_x = 0; bar.n_xcc = 0;
    if (!DF) {
        if (tid == 0) {
            pm_add(sync + PM_S_CENSUS + bar.xcc * 32, 1u);
            pm_add(sync + PM_S_TOTAL, 1u);
            const bool ok = pm_spin_ge(sync + PM_S_TOTAL, (unsigned)nwg, sync);
            for (int x = 0; x < 8; ++x) cen[x] = pm_ld(sync + PM_S_CENSUS + x * 32);
            if (!ok) pm_st(sync + PM_S_STICKY, 2u);
            ok_sh = ok ? 1 : 0;
        }
        __syncthreads();
        if (!ok_sh) return;
        bar.n_x = cen[bar.xcc];
        for (int x = 0; x < 8; ++x) bar.n_xcc += cen[x] ? 1u : 0u;
    }

    {
        static_assert(sizeof(PmUnit) % 4 == 0, "");
        constexpr int words = (int)(sizeof(PmUnit) / 4);
        for (int s = 0; s < n_slots; ++s)
            for (int q = 0; q < maxu; ++q) {
                const unsigned* src = reinterpret_cast<const unsigned*>(&P.units[((size_t)s * nwg + wg) * maxu + q]);
                unsigned* dst = reinterpret_cast<unsigned*>(&lds_units[s * maxu + q]);
                for (int i = tid; i < words; i += PM_THREADS) dst[i] = src[i];
            }
    }
    __syncthreads();
    // resident weight slabs -> LDS (once per window)
    for (int s = 0; s < n_slots; ++s)
        for (int q = 0; q < maxu; ++q) {
            const PmUnit& u = lds_units[s * maxu + q];
            if (u.kind != PM_GEMM || u.w_lds < 0) continue;
            const int nch = u.K >> 4;
            const f32x4* src = reinterpret_cast<const f32x4*>(u.W);
            f32x4* dst = reinterpret_cast<f32x4*>(lds_w + u.w_lds);
            for (int i = tid; i < nch * 64; i += PM_THREADS) dst[i] = src[i];
        }
    // prologue: the states entering the window (row-major, written by the host side) -> fragment-major slabs
    {
        int base = 0;
        for (int q = 0; q < P.ninit; ++q) {
            const PmInit in = P.init[q];
            const int nch = in.K >> 4, nblk = MB * nch;
            for (int blk = wg * 8 + wave - base; blk < nblk; blk += nwg * 8) {
                if (blk < 0) continue;
                const int rb = blk / nch, c = blk % nch;
                const int m = 16 * rb + (lane & 15), k = 16 * c + 4 * (lane >> 4);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (m < P.M) v = *reinterpret_cast<const f32x4*>(in.src + (size_t)m * in.ld + k);
                __builtin_amdgcn_raw_buffer_store_b128(
                    __builtin_bit_cast(i32x4, v), fmr, (unsigned)lane << 4,
                    __builtin_amdgcn_readfirstlane(in.dst_off + ((unsigned)(rb * in.nch + in.chunk + c) << 10)), 16);
            }
            base = (base + nblk) % (nwg * 8);
        }
    }
    unsigned epoch = 1;
    if (!DF && !pm_barrier(sync, bar, epoch++, &ok_sh)) {
        if (tid == 0) pm_st(sync + PM_S_STICKY, 3u);
        return;
    }

    // phase timers (work / barrier wait per slot, summed over the ticks): a handful of s_memrealtime reads per phase
    unsigned long long* t_work = reinterpret_cast<unsigned long long*>(lds_red + PM_LDS_RED + PM_LDS_UNITS + 20);
    unsigned long long* t_wait = t_work + PM_MAXSLOTS;  // 20 + 2 * 18 floats <= PM_LDS_MISC
    unsigned long long stage[4] = {0, 0, 0, 0};
    if (tid < 2 * PM_MAXSLOTS) t_work[tid] = 0;
    __syncthreads();
    for (int tick = 0; tick < P.n_ticks; ++tick) {
        for (int s = 0; s < n_slots; ++s) {
            const unsigned long long ta = pm_clock();
            for (int q = 0; q < maxu; ++q) {
                const PmUnit& u = lds_units[s * maxu + q];
                const int kind = __builtin_amdgcn_readfirstlane(u.kind);
                if (kind == PM_NONE) continue;
                const int t = tick - __builtin_amdgcn_readfirstlane(u.lag);
                if (t < 0 || t >= P.T) continue;
                if (kind == PM_GEMM) pm_gemm<MB, DF>(u, t, lds_w, lds_red, fmr, stage, sync);
                else pm_att_row<DF>(P.att, u.row, t, lds_att, P.fm_base, sync);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long tb = pm_clock();
            if (DF) {  // no rendezvous: only look at the abort word now and then
                if ((s == 0) && (tick & 7) == 0) {
                    if (tid == 0) ok_sh = pm_ld(sync + PM_S_ABORT) ? 0 : 1;
                    __syncthreads();
                    if (!ok_sh) return;
                }
            } else if (!pm_barrier(sync, bar, epoch++, &ok_sh)) {
                if (tid == 0) pm_st(sync + PM_S_STICKY, 4u + ((unsigned)tick << 8) + ((unsigned)s << 4));
                return;
            }
            if (tid == 0) {
                t_work[s] += tb - ta;
                t_wait[s] += pm_clock() - tb;
            }
        }
    }
    if (tid == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(sync + PM_SYNC_WORDS) + (size_t)wg * 24;
        for (int s = 0; s < PM_MAXSLOTS; ++s) { dbg[s] = t_work[s]; dbg[PM_MAXSLOTS + s] = t_wait[s]; }
        for (int q = 0; q < 4; ++q) dbg[18 + q] = stage[q];
    }
}

}  // namespace

namespace {
__global__ __launch_bounds__(256) void pm_fill_kernel(unsigned* p, long long n, unsigned v) {
    const long long head = (16 - ((unsigned long long)p & 15)) % 16 / 4;  // words up to the first 16-byte boundary
    const long long step = (long long)gridDim.x * 256, i0 = (long long)blockIdx.x * 256 + threadIdx.x;
    for (long long i = i0; i < (head < n ? head : n); i += step) p[i] = v;
    if (n > head) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4* q = reinterpret_cast<u32x4*>(p + head);
        const long long n4 = (n - head) / 4;
        const u32x4 vv = {v, v, v, v};
        for (long long i = i0; i < n4; i += step) q[i] = vv;
        for (long long i = head + 4 * n4 + i0; i < n; i += step) p[i] = v;
    }
}
__global__ __launch_bounds__(256) void pm_clear_kernel(unsigned* a, int na, unsigned* b, int nb) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < na; i += gridDim.x * 256) a[i] = 0u;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nb; i += gridDim.x * 256) b[i] = 0u;
}
}  // namespace

int pm_max_workgroups() {
    static int n = -1;
    if (n < 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        n = prop.multiProcessorCount > 256 ? 256 : prop.multiProcessorCount;
    }
    return n;
}

int pm_status(const PmProgram& P) {
    PH_CHECK(hipDeviceSynchronize());
    unsigned w = 0;
    PH_CHECK(hipMemcpy(&w, P.sync + PM_S_STICKY, sizeof(w), hipMemcpyDeviceToHost));
    if (w) {
        fprintf(stderr, "[parrot_amd] persistent launch gave up: sticky word 0x%x (site %u)\n", w, w & 15u);
        if (getenv("PARROT_PM_DUMP_PLAN")) {  // development aid: the barrier words of the launch that gave up
            unsigned h[PM_SYNC_WORDS];
            if (hipMemcpy(h, P.sync, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
                fprintf(stderr, "  nwg %d n_ticks %d n_slots %d  TOP %u TOTAL %u ABORT %u\n", P.nwg, P.n_ticks, P.n_slots, h[PM_S_TOP],
                        h[PM_S_TOTAL], h[PM_S_ABORT]);
                for (int x = 0; x < 8; ++x)
                    fprintf(stderr, "  xcc %d: census %u arrivals %u generation %u\n", x, h[PM_S_CENSUS + 32 * x], h[PM_S_XCNT + 32 * x],
                            h[PM_S_XGEN + 32 * x]);
            }
        }
    }
    return w ? PH_ERR_UNSUPPORTED + 100 : 0;
}

int pm_launch(const PmProgram& P, hipStream_t stream) {
    if (P.nwg < 1 || P.nwg > pm_max_workgroups() || !P.units || !P.sync || P.n_slots < 1 || P.n_slots > PM_MAXSLOTS ||
        P.maxu < 1 || P.n_slots * P.maxu > PM_MAXENT || P.nfill < 0 || P.nfill > PM_MAXFILL)
        return PH_ERR_BADARG;
    if (P.att.U > PM_ATT_MAXU || P.att.A > PM_ATT_MAXA) return PH_ERR_UNSUPPORTED;
    // barrier / census / abort words and the timers are cleared; the sticky words at the end of the sync area are not
    // (cleared by a kernel of our own, not by hipMemsetAsync: as graph memset nodes replayed on the default stream the
    // two memsets were seen to leave a non-zero pattern in the words when earlier work was still in flight -- every
    // barrier of the launch then found the abort word set; tests/test_gpu_persist.py, L = 3 / B = 64)
    hipLaunchKernelGGL(pm_clear_kernel, dim3(8), dim3(256), 0, stream, P.sync, (int)PM_S_STICKY, P.sync + PM_SYNC_WORDS,
                       (int)PM_DBG_WORDS);
    PH_CHECK(hipGetLastError());
    if (P.dataflow)
        for (int q = 0; q < P.nfill; ++q) {
            const long long n = P.fill[q].bytes / 4;
            const int blocks = (int)((n / 4 + 255) / 256 > 4096 ? 4096 : (n / 4 + 255) / 256 + 1);
            hipLaunchKernelGGL(pm_fill_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<unsigned*>(P.fill[q].p), n,
                               (unsigned)PM_EMPTY);
            PH_CHECK(hipGetLastError());
        }
    const size_t lds = (size_t)PM_LDS_FLOATS * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        const int l = (int)lds;
        PH_CHECK(hipFuncSetAttribute((const void*)pm_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, l));
        PH_CHECK(hipFuncSetAttribute((const void*)pm_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, l));
        PH_CHECK(hipFuncSetAttribute((const void*)pm_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, l));
        PH_CHECK(hipFuncSetAttribute((const void*)pm_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, l));
        PH_CHECK(hipFuncSetAttribute((const void*)pm_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, l));
        PH_CHECK(hipFuncSetAttribute((const void*)pm_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, l));
        attr_done = true;
    }
    const dim3 grid(P.nwg), block(PM_THREADS);
    switch (P.MB * 2 + (P.dataflow ? 1 : 0)) {
        case 2: hipLaunchKernelGGL((pm_kernel<1, false>), grid, block, lds, stream, P); break;
        case 3: hipLaunchKernelGGL((pm_kernel<1, true>), grid, block, lds, stream, P); break;
        case 4: hipLaunchKernelGGL((pm_kernel<2, false>), grid, block, lds, stream, P); break;
        case 5: hipLaunchKernelGGL((pm_kernel<2, true>), grid, block, lds, stream, P); break;
        case 8: hipLaunchKernelGGL((pm_kernel<4, false>), grid, block, lds, stream, P); break;
        case 9: hipLaunchKernelGGL((pm_kernel<4, true>), grid, block, lds, stream, P); break;
        default: return PH_ERR_BADARG;
    }
    return (int)hipGetLastError();
}
