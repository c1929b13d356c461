// Device helpers of the SampleRNN persistent-thread sample kernel (sr_persist.hip): XCD teams, EMPTY-slot hand-offs that
// stay in the XCD's L2, the register-resident [4 streams x CU columns] product, DPP wave reductions.
#pragma once
#include "common.h"

namespace {

constexpr int SRP_THREADS = 512, SRP_TEAM = 32, SRP_NTEAMS = 8, SRP_ROWS = 4, SRP_Q = 256;
constexpr int SRP_SYNC_WORDS = 1024;  // [256 + 32 team] arrivals per team (counted, not waited for), [512] abort, [513] fault code,
                                      // [600 ..) phase stamps of a timed launch (PARROT_SR_TIMING)
constexpr int SRP_MAXHIST = 64;       // FS + nsteps
constexpr int SRP_GMAX = 12;          // table rows of the gather requested in one batch (FS - 1 <= SRP_GMAX; else a loop)
// f32x4 hand-off slots per team in the workspace: x1, x2 [D] and the logits [Q]
__host__ __device__ constexpr int srp_team_vecs(int D, int Q) { return 2 * D + Q; }
// Carry between consecutive launches, per team, behind the hand-off slots: [0] the sample index it is for (-1: none),
// [16 ..) the team's last FS samples [4][SRP_CARRY_HIST], then per CU the table part of the next first gather [32][4][DC <= 32]
constexpr int SRP_CARRY_HIST = 32;
constexpr int SRP_CARRY_WORDS = 16 + SRP_ROWS * SRP_CARRY_HIST + SRP_TEAM * SRP_ROWS * 32;
__host__ __device__ constexpr long long srp_carry_base(int D, int Q) {
    return SRP_SYNC_WORDS + (long long)SRP_NTEAMS * srp_team_vecs(D, Q) * 4;
}

__device__ __forceinline__ int srp_xcc() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return (int)(v & 7);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t srp_rsrc(const void* p) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* q = (void*)(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, 0x7fffffff, 0x00020000);
}
// sc1 load: misses in the CU's vector cache, served by the XCD's L2 (where the team's stores have landed)
__device__ __forceinline__ f32x4 srp_ld(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));
}

// Hand-off slots are 16 bytes = one value for each of the team's 4 streams.  An empty slot holds a NaN with a payload
// arithmetic never produces; a 16-byte store replaces it atomically in the L2, so a consumer simply re-reads a slot
// until it is full: no store acknowledgement, arrival counter or flag poll sits between producer and consumer.
constexpr unsigned SRP_EMPTY = 0x7FC0DEADu;
__device__ __forceinline__ f32x4 srp_empty() {
    const float e = __uint_as_float(SRP_EMPTY);
    return (f32x4){e, e, e, e};
}
__device__ __forceinline__ bool srp_is_empty(const f32x4& v) {
    return __float_as_uint(v[0]) == SRP_EMPTY || __float_as_uint(v[3]) == SRP_EMPTY;
}

// 100 MHz wall clock (timing aid, PARROT_SR_TIMING=1: workgroup 0 of team 0 stamps the phase boundaries of every step
// into sync words 600..)
__device__ __forceinline__ unsigned long long srp_clock() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

struct SrpShared {
    int ok;
    int hist[SRP_ROWS][SRP_MAXHIST];
    float mx[SRP_ROWS];
};

// Load a hand-off slot, re-reading until it is full (bounded: ~1 s, then the abort word is raised).
__device__ __forceinline__ f32x4 srp_take(__amdgpu_buffer_rsrc_t r, unsigned byte_off, unsigned* abort_, SrpShared* sh) {
    f32x4 v = srp_ld(r, byte_off);
    unsigned n = 0;
    while (srp_is_empty(v)) {
        if ((++n & 1023u) == 0u) {
            if (__hip_atomic_load(abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { sh->ok = 0; break; }
            if (n > (1u << 21)) {
                __hip_atomic_store(abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(abort_ + 1, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sh->ok = 0;
                break;
            }
        }
        __builtin_amdgcn_s_sleep(1);
        v = srp_ld(r, byte_off);
    }
    return v;
}

// f32x4 += the same vector of the lane selected by a DPP control (all four components)
template <int CTRL>
__device__ __forceinline__ f32x4 srp_dpp_add(const f32x4& v) {
    f32x4 r;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        r[c] = v[c] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[c]), CTRL, 0xf, 0xf, true));
    return r;
}

// out[4 rows][CC columns of this CU] = sum_k a[k][row] * W[k][col].  Thread (sl = tid & 7, g = (tid >> 3) % GG,
// sh = tid / (8 GG)) owns K-slice s = 8 sh + sl, i.e. k = kk * SS + s, and columns 4g .. 4g+3 of the CU's slice; its
// weights w(kk) are W[k][4g .. 4g+3] (VGPRs).  The 8 slices of neighbouring lanes
// are added with DPP (quad swaps, then half-row mirror: a fixed tree), the 64 / GG lane groups through LDS in group
// order.  Result: threads tid < CC return the finished f32x4 (4 rows) of CU column 4 * (tid % GG) + tid / GG.
// Wave-wide max / min with every lane receiving the result: quad swaps, half-row and row mirrors on the DPP path (no LDS
// crossbar as __shfl_xor would use: ~100 clocks instead of ~1200 for the six levels), the four row results through
// v_readlane.
template <int CTRL>
__device__ __forceinline__ float srp_dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int srp_dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ float srp_wave_max(float v) {
    v = fmaxf(v, srp_dpp_f<0xB1>(v));
    v = fmaxf(v, srp_dpp_f<0x4E>(v));
    v = fmaxf(v, srp_dpp_f<0x141>(v));
    v = fmaxf(v, srp_dpp_f<0x140>(v));
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
__device__ __forceinline__ int srp_wave_min(int v) {
    v = min(v, srp_dpp_i<0xB1>(v));
    v = min(v, srp_dpp_i<0x4E>(v));
    v = min(v, srp_dpp_i<0x141>(v));
    v = min(v, srp_dpp_i<0x140>(v));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
// argmax of row r of the team's logits lg[Q] (4 streams per vector), lowest index on ties (numpy / Theano argmax):
// every lane returns the index and, through best, the maximum.
template <int Q>
__device__ __forceinline__ int srp_argmax_row(const f32x4* __restrict__ lg, int r, int lane, float& best) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int m = 0; m < Q / 64; ++m) {
        const float v = lg[lane + 64 * m][r];
        if (v > bv) { bv = v; bi = lane + 64 * m; }
    }
    best = srp_wave_max(bv);
    return srp_wave_min(bv == best ? bi : 0x7fffffff);
}

// v + the same vector of the lane 16 (ROWS = 16: neighbouring row) or 32 (other half of the wave) lanes away
template <int ROWS>
__device__ __forceinline__ f32x4 srp_swap_add(const f32x4& v) {
    f32x4 r;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const unsigned u = __float_as_uint(v[c]);
        if (ROWS == 16) {
            const auto q = __builtin_amdgcn_permlane16_swap(u, u, false, false);
            r[c] = __uint_as_float(q[0]) + __uint_as_float(q[1]);
        } else {
            const auto q = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            r[c] = __uint_as_float(q[0]) + __uint_as_float(q[1]);
        }
    }
    return r;
}

// Fold of a product's per-thread partial sums acc[r] (stream r, the thread's 4 columns): the 8 K-slices of neighbouring
// lanes with DPP (quad swaps, then half-row mirror: a fixed tree), the 64 / GG lane groups through LDS in group order.
// Threads tid < 4 GG return the finished f32x4 (4 streams) of CU column 4 * (tid % GG) + tid / GG.
template <int GG>
__device__ __forceinline__ void srp_reduce(f32x4 (&acc)[4], f32x4* __restrict__ red, f32x4& out, int tid,
                                           unsigned long long* tp = nullptr) {
    constexpr int CC = 4 * GG, GROUPS = 64 / GG;
    const int sl = tid & 7, g = (tid >> 3) % GG, shi = tid / (8 * GG);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        acc[r] = srp_dpp_add<0xB1>(acc[r]);   // quad_perm [1,0,3,2]
        acc[r] = srp_dpp_add<0x4E>(acc[r]);   // quad_perm [2,3,0,1]
        acc[r] = srp_dpp_add<0x141>(acc[r]);  // row_half_mirror
    }
    if (tp) tp[1] = srp_clock();
    if (sl == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) red[shi * CC + j * GG + g] = (f32x4){acc[0][j], acc[1][j], acc[2][j], acc[3][j]};
    }
    __syncthreads();
    if (tp) tp[2] = srp_clock();
    // The GROUPS partial sums of each of the CC column vectors: wave 0 takes four of them per lane in ONE batch of LDS
    // reads (lane = part * CC + column) and adds the 64 / CC parts across lanes on the VALU -- row rotate, then the
    // gfx950 row / half-wave swaps.  (Round 5 had threads tid < CC walk the GROUPS partials four at a time: eight
    // dependent LDS round trips for the output layer, 1.1 us of the step.)
    if (tid < 64) {
        static_assert(GROUPS * CC == 256, "64 lanes x 4 partials");
        const int c = tid % CC, part = tid / CC;
        const f32x4 p0 = red[(4 * part + 0) * CC + c], p1 = red[(4 * part + 1) * CC + c];
        const f32x4 p2 = red[(4 * part + 2) * CC + c], p3 = red[(4 * part + 3) * CC + c];
        f32x4 v = (p0 + p1) + (p2 + p3);
        if (CC <= 8) v = srp_dpp_add<0x128>(v);  // row_ror:8
        if (CC <= 16) v = srp_swap_add<16>(v);
        out = srp_swap_add<32>(v);
    }
    if (tp) tp[3] = srp_clock();
}

template <int KPP, int GG>
__device__ __forceinline__ void srp_layer(const f32x4* __restrict__ act, const f32x4 (&w)[KPP], f32x4* __restrict__ red,
                                          f32x4& out, int tid, unsigned long long* tp = nullptr) {
    constexpr int SS = SRP_THREADS / GG;
    const int s = 8 * (tid / (8 * GG)) + (tid & 7);
    f32x4 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // Activation reads run PF iterations ahead of the FMAs that use them, pinned by scheduling barriers: left alone the
    // scheduler (register pressure near the limit) emits read, wait, 8 FMAs, read, wait ... -- an LDS latency per step of K.
    constexpr int PF = KPP < 4 ? KPP : 4;
    f32x4 buf[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) buf[j] = act[j * SS + s];
#pragma unroll
    for (int kk = 0; kk < KPP; ++kk) {
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 a = buf[kk % PF];
        const f32x4 wv = w[kk];
        // acc[r] = the 4 columns of stream r.  The scalar operand is the (transient) activation: broadcasting the
        // loop-invariant weights instead makes the compiler keep a 4-register splat of every weight alive.
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += a[r] * wv;
        if (kk + PF < KPP) {
            __builtin_amdgcn_sched_barrier(0);
            buf[kk % PF] = act[(kk + PF) * SS + s];
        }
    }
    if (tp) tp[0] = srp_clock();
    srp_reduce<GG>(acc, red, out, tid, tp);
}

}  // namespace
