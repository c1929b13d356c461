// Device body of the attention BACKWARD step for one batch row (reference model.py:664-690 reversed), shared by the
// stand-alone kernels (attention.hip) and the fused backward tick of plans.hip schedule 7 (skinny.hip wkb_kernel), where
// the row blocks lead a launch whose wide step workgroups consume what they produce.
#pragma once
#include "attention.h"
#include "att_fwd_body.h"
#include "elementwise.h"

#include <type_traits>

constexpr int ATTB_THREADS = 1024;  // backward: one workgroup per batch row, 16 waves

static inline size_t att_bwd_lds(int U, int E) {
    const int dwsz = ((E + 3) & ~3) > 16 * 3 * ATT_MAXA ? ((E + 3) & ~3) : 16 * 3 * ATT_MAXA;  // also holds s_part
    return sizeof(float) * (6 * ATT_MAXA + 24 + dwsz + ((U + 3) & ~3));
}

// Backward of one step for batch row b (one workgroup per row).
// dh_io (optional, used when every thread owns one column of dh1, H <= ATTB_THREADS): on entry the caller's prefetched
// dh1[b][t], on exit the updated value; returns whether that path was taken (else dh1 was updated in memory only).
struct AttBwdNoLate { __device__ __forceinline__ void operator()() const {} };
// late(): called once the context registers are dead (after the dphi phase): the caller requests what it needs behind the
// attention backward there (layer 0's state-backward operands), where 64 VGPRs have just been freed -- requested at entry
// they pushed the 1024-thread block over its 128 registers (84 bytes of scratch per lane).
template <class LATE = AttBwdNoLate>
__device__ __forceinline__ bool att_bwd_row(const AttBwdArgs& g, int b, float* sm, float* dh_io = nullptr, LATE late = LATE()) {
    const int A = g.A, U = g.U, E = g.E, H = g.H;
    float* s_a = sm;                  // [A]
    float* s_b = s_a + ATT_MAXA;
    float* s_k = s_b + ATT_MAXA;
    float* s_dp = s_k + ATT_MAXA;     // [3A]
    float* s_red = s_dp + 3 * ATT_MAXA;  // [24]
    float* s_dw = s_red + 24;          // [E]
    float* s_dphi = s_dw + (((E + 3) & ~3) > 16 * 3 * ATT_MAXA ? ((E + 3) & ~3) : 16 * 3 * ATT_MAXA);  // [U]

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const float* ctx = g.ctx + (size_t)b * U * E;
    ATT_STAMP(1, b, 0);

    // Everything that does not depend on the incoming gradient is requested first: this wave's context rows
    // (for dphi) and this thread's column of the projection matrix (for dh1).  One round trip instead of a
    // chain of four dependent ones.
    constexpr int NWB = ATTB_THREADS / 64;
    constexpr int RPW = 16, SEG = 4;  // rows per wave / 64-float segments per row covered by the preload
    const bool use_pre = (U <= NWB * RPW) && (E <= 64 * SEG);
    // Support of the window saved by the forward step: outside [u_lo, u_hi] every exp(-b (kappa-u)^2) is exactly
    // 0.0f, so dphi[u] is multiplied by zero in all three mixture gradients and its context row is not needed.
    int u_lo = 0, u_hi = U - 1;
    if (g.sup) {
        u_lo = g.sup[2 * b];
        u_hi = g.sup[2 * b + 1];
    }
    // Round 6: every request below is unconditional and branch-free (indices clamped into what exists, absent gradient
    // shares re-read the first one and are dropped by a select later): under `cond ? p[i] : 0` / `if (g.dw3) v += ...`
    // the compiler ended each conditional load with s_waitcnt vmcnt(0) -- the six shares of dw alone were six dependent
    // round trips, 4.3 us from entry to the first barrier (tools/att_timing.py).  The sums keep their order.
    float cpre[RPW][SEG];
    if (use_pre) {
        const int hi_c = u_hi >= u_lo ? u_hi : u_lo;  // (empty support: u_lo = U, u_hi = -1 -> everything clamps to row U - 1 / 0)
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            int u = wave + q * NWB;
            u = u < u_lo ? u_lo : u;
            u = u > hi_c ? hi_c : u;
            u = u > U - 1 ? U - 1 : (u < 0 ? 0 : u);  // rows outside the support re-read a row that is fetched anyway
#pragma unroll
            for (int sg = 0; sg < SEG; ++sg) {
                const int e = lane + 64 * sg;
                cpre[q][sg] = ctx[(size_t)u * E + (e < E ? e : E - 1)];  // (their sums are discarded below)
            }
        }
    }
    constexpr int WPRE = 32;
    const bool use_wpre = (H <= ATTB_THREADS) && (3 * A <= WPRE);
    float wpre[WPRE];
    // the carry of kappa's gradient and kappa_{t-1}: needed four phases later, requested now
    const int ta = t < A ? t : A - 1;
    const float pre_dkappa = g.dkappa[(size_t)b * A + ta], pre_kprev = g.kappa_prev[(size_t)b * A + ta];
    const float pre_a = g.a[(size_t)b * A + ta], pre_b = g.b[(size_t)b * A + ta], pre_k = g.kappa[(size_t)b * A + ta];
    {
        const bool on2 = g.dw2 != nullptr, on3 = on2 && g.dw3, on4 = on2 && g.dw4, on5 = on2 && g.dw5, on6 = on2 && g.dw6;
        const float* p2 = on2 ? g.dw2 : g.dw;
        const float* p3 = on3 ? g.dw3 : g.dw;
        const float* p4 = on4 ? g.dw4 : g.dw;
        const float* p5 = on5 ? g.dw5 : g.dw;
        const float* p6 = on6 ? g.dw6 : g.dw;
        for (int e = t; e < E; e += ATTB_THREADS) {
            const size_t i = (size_t)b * g.lddw + e;
            const float x1 = g.dw[i], x2 = p2[i], x3 = p3[i], x4 = p4[i], x5 = p5[i], x6 = p6[i];
            float v = x1;
            if (on2) {
                v += x2;
                v += on3 ? x3 : 0.f;
                v += on4 ? x4 : 0.f;
                v += on5 ? x5 : 0.f;
                v += on6 ? x6 : 0.f;
                g.dw[i] = v;  // total, needed later for the deferred d(ctx) GEMM
            }
            s_dw[e] = v;
        }
    }
    if (t < A) {
        s_a[t] = pre_a;
        s_b[t] = pre_b;
        s_k[t] = pre_k;
    }
    ATT_STAMP(1, b, 1);  // dw total, window parameters in LDS (this wave)
    __syncthreads();
    ATT_STAMP(1, b, 2);

    // dphi[u] = sum_e dw[e] ctx[u][e]: one wave per u, lanes over e (coalesced row reads).
    if (use_pre) {
        float dseg[SEG];
#pragma unroll
        for (int sg = 0; sg < SEG; ++sg) dseg[sg] = (lane + 64 * sg < E) ? s_dw[lane + 64 * sg] : 0.f;
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int u = wave + q * NWB;
            float acc = 0.f;
#pragma unroll
            for (int sg = 0; sg < SEG; ++sg) acc += dseg[sg] * cpre[q][sg];
            if (u < U) {  // wave-uniform
                const float r = (u >= u_lo && u <= u_hi) ? wave_sum(acc) : 0.f;
                if (lane == 0) s_dphi[u] = r;
            }
        }
    } else
    for (int u0 = wave * 4; u0 < U; u0 += NWB * 4) {  // 4 context rows in flight per wave
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int e = lane; e < E; e += 64) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = (u0 + q < U) ? ctx[(size_t)(u0 + q) * E + e] : 0.f;
            const float d = s_dw[e];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += d * v[q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float r = wave_sum(acc[q]);
            if (lane == 0 && u0 + q < U) s_dphi[u0 + q] = r;
        }
    }
    ATT_STAMP(1, b, 3);  // dphi of this wave's rows
    __syncthreads();
    ATT_STAMP(1, b, 4);

    // The context registers are dead now: request this thread's column of the projection matrix (used by the
    // last phase) so that its latency hides behind the reductions below.
    late();
    if (use_wpre) {
        const int tc = t < H ? t : H - 1, jmax = 3 * A - 1;
#pragma unroll
        for (int j = 0; j < WPRE; ++j) wpre[j] = g.WattT[(size_t)(j < jmax ? j : jmax) * H + tc];  // (rows >= 3A: never used)
    }

    // da, db, dkappa: reduce over the support of the window for every mixture j, one wave per mixture
    // (lanes over u, three wave reductions per mixture, results straight into s_dp).
    {
        constexpr int NWB2 = ATTB_THREADS / 64;
        for (int j = wave; j < A; j += NWB2) {
            const float aj = s_a[j], bj = s_b[j], kj = s_k[j];
            float da = 0.f, db = 0.f, dk = 0.f;
            for (int u = u_lo + lane; u <= u_hi; u += 64) {
                const float d = kj - (float)u;
                const float dph = s_dphi[u];
                if (g.att_type == 1) {
                    const float sq = sqrtf(bj);
                    const float ex = 0.3989422917366028f * expf(-0.5f * bj * d * d);
                    da += dph * sq * ex;
                    // d/db [a sqrt(b) exp(-b d^2/2)] = a ex (1/(2 sqrt b) - sqrt(b) d^2 / 2)
                    db += dph * aj * ex * (0.5f / sq - 0.5f * sq * d * d);
                    dk += dph * aj * sq * ex * (-bj * d);
                } else {
                    const float ex = expf(-bj * d * d);
                    da += dph * ex;
                    db += dph * aj * ex * (-d * d);
                    dk += dph * aj * ex * (-2.f * bj * d);
                }
            }
            da = wave_sum(da);
            db = wave_sum(db);
            dk = wave_sum(dk);
            if (lane == 0) {
                s_dp[j] = da;            // [da | db | dkappa (without carry)]
                s_dp[A + j] = db;
                s_dp[2 * A + j] = dk;
            }
        }
    }
    __syncthreads();
    ATT_STAMP(1, b, 5);  // mixture reductions done

    // chain through the window parameterisation.
    if (g.att_type == 1) {
        if (t == 0) {
            // a = softmax(p) + eps : dp = s * (da - sum(da * s)), s = a - eps
            float dot = 0.f;
            for (int j = 0; j < A; ++j) dot += s_dp[j] * (s_a[j] - g.eps);
            s_red[20] = dot;
        }
        __syncthreads();
    }
    float dpa = 0.f, dpb = 0.f, dpk = 0.f;
    if (t < A) {
        const float sa = s_a[t] - g.eps;
        if (g.att_type == 1) dpa = sa * (s_dp[t] - s_red[20]);
        else dpa = s_dp[t] * sa;
        dpb = s_dp[A + t] * (s_b[t] - g.eps);
        const float dkt = s_dp[2 * A + t] + pre_dkappa;  // + carry from step t+1
        dpk = dkt * (s_k[t] - pre_kprev);
        g.dkappa[(size_t)b * A + t] = dkt;  // kappa_t = kappa_{t-1} + ... : carry to step t-1
    }
    __syncthreads();
    if (t < A) {
        s_dp[t] = dpa;
        s_dp[A + t] = dpb;
        s_dp[2 * A + t] = dpk;
        g.dp_out[(size_t)b * 3 * A + t] = dpa;
        g.dp_out[(size_t)b * 3 * A + A + t] = dpb;
        g.dp_out[(size_t)b * 3 * A + 2 * A + t] = dpk;
    }
    __syncthreads();
    ATT_STAMP(1, b, 6);  // dp in LDS

    // dh1[b][k] += sum_j dp[j] Watt[k][j]
    float* dh = g.dh1 + (size_t)b * g.lddh;
    if (use_wpre) {
        if (t < H) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < WPRE; ++j)
                if (j < 3 * A) acc += s_dp[j] * wpre[j];
            if (dh_io) {
                const float v = *dh_io + acc;
                dh[t] = v;
                *dh_io = v;
            } else {
                dh[t] += acc;
            }
        }
        ATT_STAMP(1, b, 7);  // dh1 updated
        return true;
    }
    for (int k = t; k < H; k += ATTB_THREADS) {
        float acc = 0.f;
#pragma unroll 6
        for (int j = 0; j < 3 * A; ++j) acc += s_dp[j] * g.WattT[(size_t)j * H + k];
        dh[k] += acc;
    }
    return false;
}

// One launch for the attention backward of layer 0's step AND the elementwise GRU state backward of every layer
// active in this tick of the backward wavefront (plans.hip): blocks [0, att_rows) take one attention row each
// and then, with the same threads (thread k updated dh1[b][k] itself), layer 0's state backward for that row;
// the remaining blocks take one (chain, row) pair of the other layers.  Saves a kernel boundary per step.
__device__ __forceinline__ void state_bwd_row(const GruStateBwdChain& c, int m, int H, int tid, int nthr) {
    gru_state_bwd_row(c, m, H, tid, nthr);
}
__device__ __forceinline__ void state_bwd_row(const LstmStateBwdChain& c, int m, int H, int tid, int nthr) {
    lstm_state_bwd_row(c, m, H, tid, nthr);
}
__device__ __forceinline__ void state_bwd_rows(const GruStateBwdChain& c, int m0, int nrows, int H, int tid, int nthr) {
    for (; nrows > 0; m0 += 4, nrows -= 4) gru_state_bwd_rows4(c, m0, nrows < 4 ? nrows : 4, H, tid, nthr);
}
__device__ __forceinline__ void state_bwd_rows(const LstmStateBwdChain& c, int m0, int nrows, int H, int tid, int nthr) {
    for (int m = m0; m < m0 + nrows; ++m) lstm_state_bwd_row(c, m, H, tid, nthr);
}

// rpb: batch rows per block of the chains that are NOT fused behind the attention (1 in the stand-alone kernel; the
// heterogeneous backward launch packs 4 so that the upper layers' elementwise rows take 16 CUs instead of 64 and every
// GEMM workgroup of the launch finds a CU at once).
template <class SA>
__device__ __forceinline__ void att_state_bwd_block(const AttBwdArgs& g, const SA& sa, int att_rows, int l0_chain, int bx,
                                                    float* sm, int rpb = 1) {
    if (bx < att_rows) {
        if constexpr (std::is_same<SA, GruStateBwdArgs>::value) {
            // Layer 0's state backward needs nothing from the attention backward except dh1 itself: its operands (and
            // the old dh1 the attention adds to) are requested up front, so that after the attention's chain of
            // dependent phases only arithmetic and three stores remain (two memory round trips less per tick).
            const int t = threadIdx.x, H = sa.H;
            if (l0_chain >= 0 && H <= ATTB_THREADS && 3 * g.A <= 32) {
                const GruStateBwdChain& c = sa.chain[l0_chain];
                const size_t i = (size_t)bx * H + t;
                // (all requested back to back, unconditionally: a share that is absent re-reads dh and is dropped by a select;
                // integer address selects: see gru_state_bwd_row)
                const size_t ic = (size_t)bx * H + (t < H ? t : 0);
                const unsigned long long a1 = (unsigned long long)c.dh, a2 = (unsigned long long)c.dh2, a3 = (unsigned long long)c.dhx[0],
                                         a4 = (unsigned long long)c.dhx[1], a5 = (unsigned long long)c.dhx[2], am = (unsigned long long)c.mask;
                const float* q2 = reinterpret_cast<const float*>(a2 ? a2 : a1);
                const float* q3 = reinterpret_cast<const float*>(a3 ? a3 : a1);
                const float* q4 = reinterpret_cast<const float*>(a4 ? a4 : a1);
                const float* q5 = reinterpret_cast<const float*>(a5 ? a5 : a1);
                const float* qm = reinterpret_cast<const float*>(am ? am + 4ull * (unsigned)bx : (unsigned long long)c.z);
                float dh = 0.f, y2 = 0.f, y3 = 0.f, y4 = 0.f, y5 = 0.f, hp = 0.f, z = 0.f, cc = 0.f, dhp = 0.f, mkl = 1.f;
                const bool got = att_bwd_row(g, bx, sm, &dh, [&]() {
                    dh = c.dh[ic];  // (= g.dh1[b][t]: the attention backward's accumulation target)
                    y2 = q2[ic]; y3 = q3[ic]; y4 = q4[ic]; y5 = q5[ic];
                    hp = c.hprev[ic]; z = c.z[ic]; cc = c.c[ic]; dhp = c.dhprev[ic]; mkl = *qm;
                });
                if (got) {
                    if (t < H) {
                        float dh2 = c.dh2 ? y2 : 0.f;
                        dh2 += c.dhx[0] ? y3 : 0.f;  // (further shares from above: added to the same late term)
                        dh2 += c.dhx[1] ? y4 : 0.f;
                        dh2 += c.dhx[2] ? y5 : 0.f;
                        const float mk = c.mask ? mkl : 1.f;
                        dh += dh2;  // same order as gru_state_bwd_row: (dh1 + attention share) + share from above
                        float dhp_direct = 0.f;
                        if (c.mask) { dhp_direct = dh * (1.f - mk); dh *= mk; }
                        c.dC[i] = dh * z * (1.f - cc * cc);
                        c.dG[(size_t)bx * 2 * H + t] = dh * (cc - hp) * z * (1.f - z);
                        c.dhprev[i] = dhp + (dh * (1.f - z) + dhp_direct);
                    }
                    ATT_STAMP(1, bx, 8);  // layer 0's state backward stored (issued)
                    return;
                }
                state_bwd_row(c, bx, H, t, ATTB_THREADS);
                return;
            }
        }
        att_bwd_row(g, bx, sm);
        if (l0_chain >= 0) state_bwd_row(sa.chain[l0_chain], bx, sa.H, threadIdx.x, ATTB_THREADS);
        return;
    }
    const int idx = bx - att_rows;
    const int bpc = (sa.B + rpb - 1) / rpb;  // blocks per chain
    int ch = idx / bpc;
    const int m0 = (idx % bpc) * rpb;
    if (att_rows > 0 && l0_chain >= 0 && ch >= l0_chain) ++ch;  // skip the chain fused above
    if (ch < sa.nchain && m0 < sa.B)
        state_bwd_rows(sa.chain[ch], m0, (m0 + rpb <= sa.B ? rpb : sa.B - m0), sa.H, threadIdx.x, ATTB_THREADS);
}
