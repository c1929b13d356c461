// Forward step of the GMM-window attention for ONE (batch row, column slice) pair, run by the first NT
// threads of a workgroup (reference model.py:664-690).  Shared by att_fwd_kernel (attention.hip) and by the
// heterogeneous step launch of skinny.hip, whose spare workgroups carry the attention of the tick beside the
// upper layers' input projections (plans.hip, schedule 5).  Everything that calls __syncthreads() here is reached by
// exactly the threads t < NT of the workgroup: callers with larger workgroups retire the other waves first
// (s_barrier only waits for the waves of a workgroup that are still alive).
#pragma once
#include "attention.h"

#ifndef ATT_PROJ_UNROLL
#define ATT_PROJ_UNROLL 8
#endif
#ifndef ATT_FWD_PRELOAD
#define ATT_FWD_PRELOAD 0  // measured: no gain in the forward kernel (the backward preloads pay)
#endif

constexpr int ATT_THREADS = 256;
constexpr int ATT_MAXA = 32;  // attention_size limit (reference default 10)

constexpr int ATT_THREADS_MAX = 512;  // the heterogeneous step launch runs the block on all eight waves of its workgroups
static inline size_t att_fwd_lds(int U) { return sizeof(float) * (6 * ATT_MAXA + 8 + ((U + 3) & ~3) + 2 * ATT_THREADS_MAX); }

template <int NT, int PROJ_UNROLL>
__device__ __forceinline__ void att_fwd_block(const AttFwdArgs& g, const int b, const int es, float* sm) {
    const int A = g.A, U = g.U, E = g.E, H = g.H;
    float* s_p = sm;                 // [3A] projection
    float* s_a = s_p + 3 * ATT_MAXA; // [A]
    float* s_b = s_a + ATT_MAXA;
    float* s_k = s_b + ATT_MAXA;
    float* s_red = s_k + ATT_MAXA;   // [8]
    float* s_phi = s_red + 8;        // [U]
    float* s_acc = s_phi + ((U + 3) & ~3);  // [NT] (+ column loop reuse)
    float* s_out = s_acc + NT;              // [NT] finished w values of one column pass (published hand-off)
    constexpr int NW = NT / 64;             // waves running the block

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const float* h = g.h1 + (size_t)b * g.ldh;
    ATT_STAMP(0, b, 0);

    // Geometry of step 4 (w[e] = sum_u phi[u] ctx[b,u,e] for this workgroup's slice of E), fixed up front so
    // that the context values can be requested before anything else: they do not depend on phi, and their
    // latency then hides behind the projection / window phases instead of following them.
    const int EW = (E + g.esplit - 1) / g.esplit;
    const int e0 = es * EW, e1 = min(E, e0 + EW);
    int CW = 1;
    while (CW < EW && CW < NT) CW <<= 1;  // columns handled per pass (power of two)
    const int G = NT / CW;                // u-groups
    const int c = t % CW, ug = t / CW;
    const float* ctx = g.ctx + (size_t)b * U * E;
    constexpr int NPRE = 32;
    const bool use_pre = ATT_FWD_PRELOAD && (EW <= CW) && ((U + G - 1) / G <= NPRE);
    float pre[NPRE];
    if (use_pre) {
        const int e = e0 + c;
#pragma unroll
        for (int q = 0; q < NPRE; ++q) {
            const int u = ug + q * G;
            pre[q] = (u < U && e < e1) ? ctx[(size_t)u * E + e] : 0.f;
        }
    }

    // kappa_{t-1}: needed two phases later, requested now (one dependent round trip less)
    const int ta_ = t < A ? t : A - 1;
    const float pre_kprev = g.kappa_prev[(size_t)b * A + ta_];

    // 1) projection p[j] = sum_k h[k] * Watt[k][j] + batt[j]; wave w handles j = w, w+NW, ...
    // Round 6: 16-byte loads, every load of a pass requested before the first is used, no conditional loads (outputs past
    // 3A re-read the last row and are dropped): one round trip of 4 + 16 wave loads of 1 KB instead of four rounds of 20
    // wave loads of 256 B -- the phase took 4.7 of the row block's 9.6 us (tools/att_timing.py).  Lane l sums
    // k = 4l .. 4l+3 (+ 256 i), then the wave: another order of the same 1024 terms than rounds 1-5 (k = l + 64 i).
    const bool pvec = !(H & 3) && !(g.ldh & 3) && !((size_t)g.h1 & 15) && !((size_t)g.WattT & 15) && 3 * A <= 4 * NW;
    if (pvec) {
        const int jmax = 3 * A - 1;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const float* wrow[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) wrow[q] = g.WattT + (size_t)min(wave + NW * q, jmax) * H;
        constexpr int KU = 4;  // 256-wide k blocks in flight per pass (registers: KU * 5 vectors)
        for (int k0 = 4 * lane; k0 < H; k0 += 256 * KU) {
            f32x4 hv[KU], wv[KU][4];
#pragma unroll
            for (int i = 0; i < KU; ++i) {
                const int k = min(k0 + 256 * i, H - 4);
                hv[i] = *reinterpret_cast<const f32x4*>(h + k);
#pragma unroll
                for (int q = 0; q < 4; ++q) wv[i][q] = *reinterpret_cast<const f32x4*>(wrow[q] + k);
            }
#pragma unroll
            for (int i = 0; i < KU; ++i) {
                if (k0 + 256 * i >= H) break;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[q] += hv[i][u] * wv[i][q][u];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float r = wave_sum(acc[q]);
            const int j = wave + NW * q;
            if (lane == 0 && j < 3 * A) s_p[j] = r + (g.batt ? g.batt[j] : 0.f);
        }
    } else
    // Wave w owns outputs j = w, w+NW, ... (up to 8 per pass); the loads of all its outputs for one k-slab
    // are issued together (8 rows + h in flight), instead of one output after the other.
    for (int jb = wave; jb < 3 * A; jb += NW * 8) {
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.f;
#pragma unroll PROJ_UNROLL
        for (int k = lane; k < H; k += 64) {
            const float hv = h[k];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int j = jb + NW * q;
                const float wv = (j < 3 * A) ? g.WattT[(size_t)j * H + k] : 0.f;
                acc[q] += hv * wv;
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float r = wave_sum(acc[q]);
            const int j = jb + NW * q;
            if (lane == 0 && j < 3 * A) s_p[j] = r + (g.batt ? g.batt[j] : 0.f);
        }
    }
    ATT_STAMP(0, b, 1);  // projection done (this wave)
    __syncthreads();
    ATT_STAMP(0, b, 2);

    // 2) window parameters
    if (g.att_type == 1) {
        if (t == 0) {
            float mx = -INFINITY;
            for (int j = 0; j < A; ++j) mx = fmaxf(mx, s_p[j]);
            float s = 0.f;
            for (int j = 0; j < A; ++j) s += expf(s_p[j] - mx);
            s_red[4] = mx;
            s_red[5] = s;
        }
        __syncthreads();
    }
    if (t < A) {
        float av;
        if (g.att_type == 1) av = expf(s_p[t] - s_red[4]) / s_red[5] + g.eps;
        else av = expf(s_p[t]) + g.eps;
        const float bv = expf(s_p[A + t]) * g.sharpening + g.eps;
        const float kv = pre_kprev + g.alignment * expf(s_p[2 * A + t]) / g.timing;
        s_a[t] = av;
        s_b[t] = bv;
        s_k[t] = kv;
        if (es == 0) {
            g.a_out[(size_t)b * A + t] = av;
            g.b_out[(size_t)b * A + t] = bv;
            g.kappa_out[(size_t)b * A + t] = kv;
        }
    }
    __syncthreads();
    ATT_STAMP(0, b, 3);  // window parameters in LDS

    // 3) phi[u].  Round 6: two threads per position where the block has them (each sums every second mixture, a lane
    // swap adds the halves: sum of the even terms + sum of the odd ones, another order than rounds 1-5), and the support
    // [lo, hi] = positions whose phi is not exactly zero comes from wave ballots + one LDS word per wave instead of two
    // LDS atomics per position on one address (the phase took 2.3 us).
    const int P = (NT >= 2 * U) ? 2 : 1;
    int w_lo = U, w_hi = -1;  // this wave's support (wave-uniform)
    for (int base = 0; base < U; base += NT / P) {
        const int u = base + (P == 2 ? (t >> 1) : t);
        const int half = P == 2 ? (t & 1) : 0;
        float ph = 0.f;
        const float uf = (float)u;
        if (g.att_type == 1) {
            for (int j = half; j < A; j += P) {
                const float d = s_k[j] - uf;
                ph += s_a[j] * sqrtf(s_b[j]) * expf(-0.5f * s_b[j] * d * d);
            }
        } else {
            for (int j = half; j < A; j += P) {
                const float d = s_k[j] - uf;
                ph += s_a[j] * expf(-s_b[j] * d * d);
            }
        }
        if (P == 2) ph += __shfl_xor(ph, 1, 64);
        if (g.att_type == 1) ph *= 0.3989422917366028f;
        const bool mine = u < U && half == 0;
        if (mine) {
            s_phi[u] = ph;
            if (es == 0) g.phi_out[(size_t)b * U + u] = ph;
        }
        const unsigned long long nz = __ballot(mine && ph != 0.f);
        if (nz) {  // (wave-uniform) positions of the first / last lane with a non-zero weight
            const int l_lo = __ffsll((long long)nz) - 1, l_hi = 63 - __clzll((long long)nz);
            const int u_first = base + (P == 2 ? ((wave * 64 + l_lo) >> 1) : wave * 64 + l_lo);
            const int u_last = base + (P == 2 ? ((wave * 64 + l_hi) >> 1) : wave * 64 + l_hi);
            w_lo = min(w_lo, u_first);
            w_hi = max(w_hi, u_last);
        }
    }
    if (lane == 0) {
        reinterpret_cast<int*>(s_acc)[wave] = w_lo;
        reinterpret_cast<int*>(s_acc)[NW + wave] = w_hi;
    }
    __syncthreads();
    ATT_STAMP(0, b, 4);  // phi and its support known
    // The Gaussian window underflows to exactly 0.0f a few positions away from kappa (exp(-b d^2), fp32), and a
    // zero weight adds exactly nothing to w: rows outside [lo, hi] are not read.  Same sums, same order, minus
    // the +0 terms -- bit-identical to reading all U rows (PARROT_ATT_DENSE=1 reads them all).
    int s_lo = U, s_hi = -1;
#pragma unroll
    for (int q = 0; q < NW; ++q) {
        s_lo = min(s_lo, reinterpret_cast<const int*>(s_acc)[q]);
        s_hi = max(s_hi, reinterpret_cast<const int*>(s_acc)[NW + q]);
    }
    int u_lo = 0, u_hi = U - 1;
    if (!g.dense) {
        u_lo = s_lo;
        u_hi = s_hi;
    }
    if (g.sup_out && es == 0 && t == 0) {  // saved for the backward step (dense mode saves the full range)
        g.sup_out[2 * b] = u_lo;
        g.sup_out[2 * b + 1] = u_hi;
    }

    // (hand-off mode) whole 16-byte groups: every slice starts and ends on a multiple of 4 columns of aligned rows
    const bool wide = g.flag && !(EW & 3) && !(E & 3) && !(g.ldw & 3) && !((size_t)g.w_out & 15) && CW >= 4;
    // 4) w[e] = sum_u phi[u] ctx[b,u,e] for this workgroup's slice of E.
    for (int eb = e0; eb < e1; eb += CW) {
        const int e = eb + c;
        float acc = 0.f;
        if (use_pre) {
#pragma unroll
            for (int q = 0; q < NPRE; ++q) {
                const int u = ug + q * G;
                if (u < U) acc += s_phi[u] * pre[q];
            }
        } else if (e < e1 && u_lo <= u_hi) {
            // rows of this thread: u = ug (mod G), as in the dense walk, starting at the first one >= u_lo
            int u = u_lo + ((ug - u_lo % G + G) % G);
            for (; u + 7 * G <= u_hi; u += 8 * G) {  // 8 independent row reads in flight
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = ctx[(size_t)(u + q * G) * E + e];
#pragma unroll
                for (int q = 0; q < 8; ++q) acc = __builtin_fmaf(s_phi[u + q * G], v[q], acc);
            }
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (u + q * G <= u_hi) ? ctx[(size_t)(u + q * G) * E + e] : 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (u + q * G <= u_hi) acc = __builtin_fmaf(s_phi[u + q * G], v[q], acc);  // same rounding as above
        }
        ATT_STAMP(0, b, 5);  // this wave's context rows multiplied
        __syncthreads();
        s_acc[t] = acc;
        __syncthreads();
        ATT_STAMP(0, b, 6);
        if (ug == 0 && e < e1) {
            float s = 0.f;
            for (int q = 0; q < G; ++q) s += s_acc[q * CW + c];
            if (!g.flag) g.w_out[(size_t)b * g.ldw + e] = s;
            else if (wide) s_out[c] = s;
            else  // consumed inside this launch: write-through, visible to every XCD once vmcnt drains
                __hip_atomic_store(reinterpret_cast<unsigned*>(g.w_out + (size_t)b * g.ldw + e), __float_as_uint(s),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (g.flag && wide) {  // 16-byte write-through stores (a 4-byte sc1 store is one fabric write each)
            __syncthreads();
            if (4 * t < CW && eb + 4 * t < e1) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(s_out + 4 * t);
                float* p = g.w_out + (size_t)b * g.ldw + eb + 4 * t;
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
            }
        }
    }
    ATT_STAMP(0, b, 7);  // w stored (issued)
    if (g.flag) {  // publish: every wave drains its stores, then one lane arrives (protocol of persist.hip's barrier)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) (void)__hip_atomic_fetch_add(g.flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

