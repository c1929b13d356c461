// SampleRNN sample-level MLP as a persistent-thread kernel (gfx950).
//
// The loop body of three_tier.py:809-832 is a chain of four dependent [B,D] products per audio sample
// (embedding gather-sum -> L2 -> L3 -> Output -> argmax -> next sample).  As separate launches each link costs a
// launch (~6 us: 31.7 us per sample at B = 32, D = 1024); as phases of a chip-wide persistent kernel each link costs a
// cross-XCD hand-off (~4.7 us).  But the streams are independent and the sample-level MLP is small, so the
// chip is split along its XCDs instead: XCD x (32 CUs, one workgroup each, private L2) runs streams 4x .. 4x+3 through
// the complete MLP, and nothing ever crosses an XCD boundary:
//   * the first product of the chain is gone (round 5): everything in front of L2's ReLU is linear, so the caller
//     composes it through W2 -- the embedding tables (t2tbl[pos] = emb_tbl[pos] . W2) and the frame tier's output
//     projection (frame_out = h . (Wout_i . W2) + bout_i . W2 + b2) -- and L2's pre-activation is a ten-row gather-sum:
//     nine rows known a step early (summed in the shadow of the previous step), the newest sample's row from this CU's
//     slice of t2tbl[FS-1] in LDS (32 KB).  Per sample: three hand-offs and two products instead of four and three;
//   * every CU owns D/32 output columns of L3 and Q/32 of the output layer and keeps those weight slices in VGPRs
//     (64 + 16 per thread) for the whole launch: no weight traffic per sample;
//   * a hand-off = 16-byte stores of the CU's [4 streams x columns] slice (write-through into the XCD's L2) and sc1
//     loads (TCP miss, L2 hit) on the consumer side that re-read a slot until it is no longer EMPTY (a NaN payload):
//     one store-to-load latency per link.  The counted variant (store, s_waitcnt, L2 atomic, poll, load) measured
//     1.66 us in isolation (tools/xcd_probe.hip; agent-scope fences: 45 us, buffer_inv: 15 us) and 4 us per link inside
//     this kernel, of which 2.4 us were the four serialised L2 round trips;
//   * the 4 streams of a team ride in the 4 lanes of an f32x4, the products run on the vector ALUs (a 4-row tile wastes
//     3/4 of an MFMA; v_pk_fma_f32 has the same f32 peak as the matrix pipe);
//   * the pick (argmax with lowest-index ties, or the seeded inverse-CDF draw) is recomputed by every CU of the team from
//     the team's logits, so the new sample index is local knowledge everywhere and the embedding gather of the next step
//     needs no further hand-off.
// One launch covers the FRAME_SIZE steps between two frame-tier steps; the tiers above stay on the launch path.
// Every spin is bounded; a team that times out (or a workgroup that finds its XCD's team already complete) raises the
// abort word, everybody leaves, srp_status() reports it and the caller falls back to the per-sample launches.
#include "sr_persist.h"

#include <stdlib.h>

namespace {

constexpr int SRP_THREADS = 512, SRP_TEAM = 32, SRP_NTEAMS = 8, SRP_ROWS = 4, SRP_Q = 256;
constexpr int SRP_SYNC_WORDS = 1024;  // [0..255] arrive (32 words per team), [256..511] census, [512] abort, [513] fault code
constexpr int SRP_MAXHIST = 64;       // FS + nsteps

__device__ __forceinline__ int srp_xcc() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return (int)(v & 7);
}
// RMW executed in the issuing XCD's L2 (no sc1: the line never leaves this XCD), returns the previous value
__device__ __forceinline__ unsigned srp_l2_add(unsigned* p, unsigned v) {
    unsigned old;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(old) : "v"(p), "v"(v) : "memory");
    return old;
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
    int rank, ok, gen;
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
template <int KPP, int GG>
__device__ __forceinline__ void srp_layer(const f32x4* __restrict__ act, const f32x4 (&w)[KPP], f32x4* __restrict__ red,
                                          f32x4& out) {
    constexpr int CC = 4 * GG, SS = SRP_THREADS / GG, GROUPS = 64 / GG;
    const int tid = threadIdx.x, sl = tid & 7, g = (tid >> 3) % GG, shi = tid / (8 * GG), s = 8 * shi + sl;
    f32x4 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KPP; ++kk) {
        const f32x4 a = act[kk * SS + s];
        const f32x4 wv = w[kk];
        // acc[r] = the 4 columns of stream r.  The scalar operand is the (transient) activation: broadcasting the
        // loop-invariant weights instead makes the compiler keep a 4-register splat of every weight alive.
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += a[r] * wv;
        if ((kk & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // do not hoist all KPP operand reads: registers are tight
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        acc[r] = srp_dpp_add<0xB1>(acc[r]);   // quad_perm [1,0,3,2]
        acc[r] = srp_dpp_add<0x4E>(acc[r]);   // quad_perm [2,3,0,1]
        acc[r] = srp_dpp_add<0x141>(acc[r]);  // row_half_mirror
    }
    if (sl == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) red[shi * CC + j * GG + g] = (f32x4){acc[0][j], acc[1][j], acc[2][j], acc[3][j]};
    }
    __syncthreads();
    if (tid < CC) {
        f32x4 v = red[tid];
#pragma unroll 4
        for (int p = 1; p < GROUPS; ++p) v += red[p * CC + tid];
        out = v;
    }
}

template <int D>
__global__ __launch_bounds__(SRP_THREADS) void srp_kernel(const SrpArgs a) {
    constexpr int Q = SRP_Q;
    constexpr int DC = D / 32, G = DC / 4, S = SRP_THREADS / G, KP = D / S;
    constexpr int QC = Q / 32, GQ = QC / 4, SQ = SRP_THREADS / GQ, KQ = D / SQ;
    static_assert(KP >= 1 && KQ >= 1 && S * G == SRP_THREADS && S * KP == D, "unsupported width");
    extern __shared__ __attribute__((aligned(16))) char srp_smem[];
    f32x4* act = reinterpret_cast<f32x4*>(srp_smem);       // [D]
    f32x4* red = act + D;                                   // [256]
    f32x4* lg = red + 256;                                  // [Q]   team logits (4 streams per vector)
    float* ev = reinterpret_cast<float*>(lg + Q);           // [4][Q] exp values of the temperature draw
    float* tmp = ev + 4 * Q;                                // [4 * DC] transposition scratch of the gather phase
    float* t2l = tmp + 4 * DC;                              // [Q][DC] this CU's columns of t2tbl[FS-1] (the newest sample's row)
    SrpShared* sh = reinterpret_cast<SrpShared*>(t2l + Q * DC);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned* sync = reinterpret_cast<unsigned*>(a.ws);
    unsigned* abort_ = sync + 512;
    const int team = srp_xcc();
    if (tid == 0) {
        const unsigned old = srp_l2_add(sync + 256 + team * 32, 1u);
        sh->rank = (int)(old % SRP_TEAM);
        sh->gen = (int)(old / SRP_TEAM);
        sh->ok = 1;
    }
    __syncthreads();
    const int cu = sh->rank;

    // ---- weight slices: L3 and Output -> registers; this CU's columns of the newest sample's table -> LDS
    const int g = (tid >> 3) % G, s = 8 * (tid / (8 * G)) + (tid & 7);
    const int gq = (tid >> 3) % GQ, sq = 8 * (tid / (8 * GQ)) + (tid & 7);
    f32x4 w3[KP], w4[KQ];
#pragma unroll
    for (int kk = 0; kk < KP; ++kk)
        w3[kk] = *reinterpret_cast<const f32x4*>(a.W3 + (size_t)(kk * S + s) * D + cu * DC + 4 * g);
    for (int idx = tid; idx < Q * (DC / 4); idx += SRP_THREADS) {
        const int q = idx / (DC / 4), c4 = idx % (DC / 4);
        reinterpret_cast<f32x4*>(t2l)[idx] =
            *reinterpret_cast<const f32x4*>(a.t2tbl + ((size_t)(a.FS - 1) * Q + q) * D + cu * DC + 4 * c4);
    }
#pragma unroll
    for (int kk = 0; kk < KQ; ++kk)
        w4[kk] = *reinterpret_cast<const f32x4*>(a.W4 + (size_t)(kk * SQ + sq) * Q + cu * QC + 4 * gq);
    // column this thread finishes in the reductions (threads tid < DC / tid < QC)
    const int fin_h = cu * DC + 4 * (tid % G) + tid / G;
    const int fin_q = cu * QC + 4 * (tid % GQ) + tid / GQ;
    const float bias3 = tid < DC ? a.b3[fin_h] : 0.f;
    const float bias4 = tid < QC ? a.b4[fin_q] : 0.f;

    const int t0 = a.tbase[0] + a.toff;
    if (tid < SRP_ROWS * a.FS) {
        const int r = tid / a.FS, pos = tid % a.FS;
        const int b = min(team * SRP_ROWS + r, a.B - 1);
        sh->hist[r][pos] = a.samples[(size_t)b * a.len + t0 - a.FS + pos];
    }
    // team exchange buffers ([D] f32x4 each, 4 streams per vector): x1, x2, and the logits ([Q] f32x4)
    float* xbase = a.ws + SRP_SYNC_WORDS + (size_t)team * (2 * D + Q) * 4;
    const __amdgpu_buffer_rsrc_t xr = srp_rsrc(xbase);
    f32x4* x1 = reinterpret_cast<f32x4*>(xbase);
    f32x4* x2 = x1 + D;
    f32x4* lb = x2 + D;
    __syncthreads();

    // Slot life cycle.  All slots are EMPTY when a plan is created (srp_init_ws) and every launch leaves them EMPTY again,
    // except the logits of its last step, which their new owners empty right here (nobody looks at the logits before
    // having taken a complete x2, and the emptying threads publish part of x1 -- after s_waitcnt vmcnt(0) -- before that).
    // During the launch a buffer is emptied by its owner as soon as the owner has taken the NEXT buffer of the chain from
    // all 32 CUs (which proves that all of them are done with this one), always by threads that later publish, behind an
    // s_waitcnt vmcnt(0), something the readers take before they look at the emptied buffer again.  No counted barrier.
    if (tid < QC) lb[fin_q] = srp_empty();

    // part_i = frame_out[:, i] + sum_{pos < FS-1} t2tbl[pos][sample[t - FS + pos]] for this CU's DC columns: everything of
    // step i's L2 pre-activation that is known one step early (frame_out carries the composed projection and both
    // biases); threads tid < DC keep it as one f32x4 (4 streams) of column fin_h
    f32x4 part = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto make_part = [&](int i) {
        if (tid < SRP_ROWS * DC) {
            const int r = tid / DC, c = tid % DC, col = cu * DC + c;
            const int b = min(team * SRP_ROWS + r, a.B - 1);
            float acc = a.frame_out[(size_t)b * a.ldf + (size_t)i * D + col];
            for (int pos = 0; pos < a.FS - 1; ++pos) {
                const int q = sh->hist[r][i + pos];
                acc += a.t2tbl[((size_t)pos * Q + q) * D + col];
            }
            tmp[c * SRP_ROWS + r] = acc;
        }
        __syncthreads();
        if (tid < DC) part = reinterpret_cast<const f32x4*>(tmp)[fin_h - cu * DC];
    };
    make_part(0);

    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(sync + 600);
    const bool timing = a.pad && team == 0 && cu == 0 && tid == 0;
    auto stamp = [&](int i, int q) { if (timing && i < 10) stamps[i * 8 + q] = srp_clock(); };
    for (int i = 0; i < a.nsteps; ++i) {
        const bool more = i + 1 < a.nsteps;
        stamp(i, 0);
        // ---- x1 = relu(part_i + the newest sample's row): no product, the owner publishes its columns right away
        if (tid < DC) {
            const int c = fin_h - cu * DC;
            f32x4 v = part;
#pragma unroll
            for (int r = 0; r < SRP_ROWS; ++r) v[r] += t2l[sh->hist[r][i + a.FS - 1] * DC + c];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            x1[fin_h] = (f32x4){fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
        }
        stamp(i, 1);
        {
            for (int k = tid; k < D; k += SRP_THREADS) act[k] = srp_take(xr, (unsigned)(k * 16), abort_, sh);
            __syncthreads();
            if (!sh->ok) return;
            stamp(i, 2);
            // x1(i) complete => every CU is done with the logits of step i-1
            if (i > 0 && tid < QC) lb[fin_q] = srp_empty();
            f32x4 v;
            srp_layer<KP, G>(act, w3, red, v);
            if (tid < DC) {
                v += bias3;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                x2[fin_h] = (f32x4){fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
            }
            stamp(i, 3);
            if (more) make_part(i + 1);  // needs nothing from the other workgroups: runs while their x2 slices arrive
            stamp(i, 4);
        }
        {
            for (int k = tid; k < D; k += SRP_THREADS) act[k] = srp_take(xr, (unsigned)((D + k) * 16), abort_, sh);
            __syncthreads();
            if (!sh->ok) return;
            stamp(i, 5);
            if (tid < QC) {  // x2(i) complete => every CU is done with x1(i); emptied by the threads that publish lb
#pragma unroll
                for (int q = 0; q < DC / QC; ++q) x1[cu * DC + tid * (DC / QC) + q] = srp_empty();
            }
            f32x4 v;
            srp_layer<KQ, GQ>(act, w4, red, v);
            if (tid < QC) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                lb[fin_q] = v + bias4;
            }
        }

        // ---- pick (every CU of the team, identical result): argmax with lowest-index ties, or the seeded draw
        if (tid < Q) lg[tid] = srp_take(xr, (unsigned)((2 * D + tid) * 16), abort_, sh);
        __syncthreads();
        if (!sh->ok) return;
        stamp(i, 6);
        if (tid < DC) x2[fin_h] = srp_empty();  // logits(i) complete => every CU is done with x2(i)
        const int t = t0 + i;
        if (wave < SRP_ROWS) {
            const int r = wave;
            float best = -INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int m = 0; m < Q / 64; ++m) {
                const float v = lg[lane + 64 * m][r];
                if (v > best) { best = v; bi = lane + 64 * m; }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float ov = __shfl_xor(best, off, 64);
                const int oi = __shfl_xor(bi, off, 64);
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            int pick = bi;
            if (a.temperature > 0.f) {
#pragma unroll
                for (int m = 0; m < Q / 64; ++m) ev[r * Q + lane + 64 * m] = expf((lg[lane + 64 * m][r] - best) / a.temperature);
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) {
                    const int b = team * SRP_ROWS + r;
                    float tot = 0.f;
                    for (int q = 0; q < Q; ++q) tot += ev[r * Q + q];
                    unsigned long long x = a.seed ^ (0x9E3779B97F4A7C15ull * (unsigned long long)(t + 1)) ^
                                           (0xBF58476D1CE4E5B9ull * (unsigned long long)(b + 1));
                    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
                    const float u = (float)((x >> 40) + 0.5) * (1.0f / 16777216.0f) * tot;
                    float c = 0.f;
                    pick = Q - 1;
                    for (int q = 0; q < Q; ++q) {
                        c += ev[r * Q + q];
                        if (u < c) { pick = q; break; }
                    }
                }
            }
            if (lane == 0) {
                sh->hist[r][a.FS + i] = pick;
                const int b = team * SRP_ROWS + r;
                if (cu == 0 && b < a.B) a.samples[(size_t)b * a.len + t] = pick;
            }
        }
        if (a.logits && i == a.nsteps - 1 && cu == 0 && tid < Q) {
#pragma unroll
            for (int r = 0; r < SRP_ROWS; ++r) {
                const int b = team * SRP_ROWS + r;
                if (b < a.B) a.logits[(size_t)b * Q + tid] = lg[tid][r];
            }
        }
        __syncthreads();
        stamp(i, 7);
    }
    // ---- the next frame's frame-tier input for this team's streams (saves the launch that would compute it)
    if (a.next_in) {
        const float half_q = (float)(Q / 2);
        const int NN = a.next_n > 0 ? a.next_n : D, NC = NN / 32;  // this CU's share: NC of the NN columns
        for (int idx = tid; idx < SRP_ROWS * NC; idx += SRP_THREADS) {
            const int r = idx / NC, col = cu * NC + idx % NC;
            const int b = team * SRP_ROWS + r;
            if (b >= a.B) continue;
            float acc = 0.f;
            for (int p = 0; p < a.FS; ++p) {
                const float xf = ((float)sh->hist[r][a.nsteps + p] / half_q - 1.0f) * 2.0f;
                acc = fmaf(xf, a.next_Win[(size_t)p * NN + col], acc);
            }
            a.next_in[(size_t)b * NN + col] = acc + (a.next_bias ? a.next_bias[col] : 0.f) + a.next_add[(size_t)b * a.next_ld_add + col];
        }
    }
}

size_t srp_lds_bytes(int D) {
    return (size_t)(D + 256 + SRP_Q) * 16 + (size_t)(4 * SRP_Q + 4 * (D / 32) + SRP_Q * (D / 32)) * 4 + sizeof(SrpShared) + 64;
}

}  // namespace

bool srp_eligible(int B, int D, int Q, int FS) {
    static int cus = -1;
    const char* e = getenv("PARROT_SR_PERSIST");
    const int enabled = e ? atoi(e) : 1;
    if (cus < 0) {
        hipDeviceProp_t prop;
        int dev = 0;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 0;
    }
    if (!enabled || cus != SRP_TEAM * SRP_NTEAMS) return false;
    if (B < 1 || B > SRP_ROWS * SRP_NTEAMS || Q != SRP_Q || FS < 1 || 2 * FS > SRP_MAXHIST) return false;
    return D == 256 || D == 512 || D == 1024;
}

long long srp_ws_floats(int D, int Q) { return SRP_SYNC_WORDS + (long long)SRP_NTEAMS * (2 * D + Q) * 4; }

int srp_init_ws(float* ws, int D, int Q) {
    // barrier / census / abort words zero, every hand-off slot EMPTY
    PH_CHECK(hipMemset(ws, 0, SRP_SYNC_WORDS * sizeof(float)));
    PH_CHECK(hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(ws + SRP_SYNC_WORDS), (int)SRP_EMPTY,
                          (size_t)SRP_NTEAMS * (2 * D + Q) * 4));
    return (int)hipDeviceSynchronize();
}

int srp_status(const float* ws) {
    unsigned w[2] = {0, 0};
    if (hipMemcpy(w, reinterpret_cast<const unsigned*>(ws) + 512, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return w[0] ? (int)(w[1] ? w[1] : 1) : 0;
}

int srp_prepare(int D) {
    const int lds = (int)srp_lds_bytes(D);
    switch (D) {
        case 256: PH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(srp_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); break;
        case 512: PH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(srp_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); break;
        case 1024: PH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(srp_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); break;
        default: return PH_ERR_UNSUPPORTED;
    }
    return 0;
}

int srp_launch(const SrpArgs& a, hipStream_t stream) {
    if (!a.ws || a.nsteps < 1 || a.FS + a.nsteps > SRP_MAXHIST) return PH_ERR_BADARG;
    const size_t lds = srp_lds_bytes(a.D);
    const dim3 grid(SRP_TEAM * SRP_NTEAMS), block(SRP_THREADS);
    switch (a.D) {
        case 256: hipLaunchKernelGGL(srp_kernel<256>, grid, block, lds, stream, a); break;
        case 512: hipLaunchKernelGGL(srp_kernel<512>, grid, block, lds, stream, a); break;
        case 1024: hipLaunchKernelGGL(srp_kernel<1024>, grid, block, lds, stream, a); break;
        default: return PH_ERR_UNSUPPORTED;
    }
    return (int)hipGetLastError();
}
