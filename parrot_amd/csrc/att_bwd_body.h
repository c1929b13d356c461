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
__device__ __forceinline__ bool att_bwd_row(const AttBwdArgs& g, int b, float* sm, float* dh_io = nullptr) {
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
    float cpre[RPW][SEG];
    if (use_pre) {
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int u = wave + q * NWB;
#pragma unroll
            for (int sg = 0; sg < SEG; ++sg) {
                const int e = lane + 64 * sg;
                cpre[q][sg] = (u >= u_lo && u <= u_hi && e < E) ? ctx[(size_t)u * E + e] : 0.f;
            }
        }
    }
    constexpr int WPRE = 32;
    const bool use_wpre = (H <= ATTB_THREADS) && (3 * A <= WPRE);
    float wpre[WPRE];

    for (int e = t; e < E; e += ATTB_THREADS) {
        float v = g.dw[(size_t)b * g.lddw + e];
        if (g.dw2) {
            v += g.dw2[(size_t)b * g.lddw + e];
            if (g.dw3) v += g.dw3[(size_t)b * g.lddw + e];
            if (g.dw4) v += g.dw4[(size_t)b * g.lddw + e];
            if (g.dw5) v += g.dw5[(size_t)b * g.lddw + e];
            if (g.dw6) v += g.dw6[(size_t)b * g.lddw + e];
            g.dw[(size_t)b * g.lddw + e] = v;  // total, needed later for the deferred d(ctx) GEMM
        }
        s_dw[e] = v;
    }
    if (t < A) {
        s_a[t] = g.a[(size_t)b * A + t];
        s_b[t] = g.b[(size_t)b * A + t];
        s_k[t] = g.kappa[(size_t)b * A + t];
    }
    __syncthreads();

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
    __syncthreads();

    // The context registers are dead now: request this thread's column of the projection matrix (used by the
    // last phase) so that its latency hides behind the reductions below.
    if (use_wpre && t < H) {
#pragma unroll
        for (int j = 0; j < WPRE; ++j) wpre[j] = (j < 3 * A) ? g.WattT[(size_t)j * H + t] : 0.f;
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
        const float dkt = s_dp[2 * A + t] + g.dkappa[(size_t)b * A + t];  // + carry from step t+1
        dpk = dkt * (s_k[t] - g.kappa_prev[(size_t)b * A + t]);
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
                float dh = 0.f, dh2 = 0.f, hp = 0.f, z = 0.f, cc = 0.f, dhp = 0.f, mk = 1.f;
                if (t < H) {
                    dh = c.dh[i];  // (= g.dh1[b][t]: the attention backward's accumulation target)
                    if (c.dh2) dh2 = c.dh2[i];
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        if (c.dhx[q]) dh2 += c.dhx[q][i];  // (further shares from above: added to the same late term)
                    hp = c.hprev[i]; z = c.z[i]; cc = c.c[i]; dhp = c.dhprev[i];
                    if (c.mask) mk = c.mask[bx];
                }
                const bool got = att_bwd_row(g, bx, sm, &dh);
                if (got) {
                    if (t < H) {
                        dh += dh2;  // same order as gru_state_bwd_row: (dh1 + attention share) + share from above
                        float dhp_direct = 0.f;
                        if (c.mask) { dhp_direct = dh * (1.f - mk); dh *= mk; }
                        c.dC[i] = dh * z * (1.f - cc * cc);
                        c.dG[(size_t)bx * 2 * H + t] = dh * (cc - hp) * z * (1.f - z);
                        c.dhprev[i] = dhp + (dh * (1.f - z) + dhp_direct);
                    }
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
    if (ch < sa.nchain)
        for (int m = m0; m < m0 + rpb && m < sa.B; ++m) state_bwd_row(sa.chain[ch], m, sa.H, threadIdx.x, ATTB_THREADS);
}
