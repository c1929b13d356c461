// Row-owning GRU scan for narrow layers (H <= 128): the bidirectional encoder of model.py:201-247 and the other
// plain GatedRecurrent scans (parrot_gru_seq_*).
//
// The recurrence of a GRU never mixes batch rows, and with H = 128 the whole weight set of a chain (Wg [H,2H] + Wc
// [H,H] = 192 KB) is re-read from L2 in ~2 us.  So instead of two (forward) / three (backward) latency-bound launches
// per time step -- 320 launches of ~5.8 us for the literal encoder of BASELINE configs[1], 1.8 ms of an 75 ms step --
// ONE launch per direction runs the whole sequence: a workgroup (4 waves) owns 16 rows of one chain, keeps their state
// in LDS, and walks all T steps with two / three workgroup barriers per step and no grid-level synchronisation at all.
// Products on v_mfma_f32_16x16x4_f32 (exact f32), weights from the fragment-major copies of skinny.hip
// (sk_tile_weights: one contiguous 1 KB wave load per 16 x 16 block), activations from LDS.
//
// Algebra = the per-step launch path of plans.hip (GruSeqPlan), i.e. Blocks GatedRecurrent (twin:
// sampleRNN/lib/ops.py:364-393): z|r = sigmoid(h Wg + gate_inputs), c = tanh((r*h) Wc + inputs),
// h' = z c + (1 - z) h, optional step mask.  Same saved activations, same gradient buffers; the K sums run in chunk
// order inside one wave instead of being split over eight, so results agree with the launch path to rounding.
#include "rowgru.h"

#include "skinny.h"

namespace {

constexpr int RG_THREADS = 256, RG_ROWS = 16, RG_MAXH = 128;  // (weights in registers: (3 H / 16 / 4 waves) * H / 4 VGPRs)

__device__ __forceinline__ float rg_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// The weights never change during a scan, and a workgroup of 4 waves has a whole SIMD's register file per wave: every
// wave keeps the B operands of its column tiles in VGPRs for the whole sequence (H = 128: 192 registers), so a step is
// LDS reads + MFMAs only.  (First version: the blocks were re-read from L2 in every step, one dependent round trip per
// 16-deep chunk: 15 us per step, no faster than the launches it replaced.)
// wb[q][c] = block (tile ct0 + 4 q, chunk c) of the fragment-major copy: lane l holds W[16c + 4 (l >> 4) + u][16 ct + (l & 15)].
template <int NT, int NCH>
__device__ __forceinline__ void rg_load_w(const float* __restrict__ Bt, int ct0, int ntiles, int lane, f32x4 (&wb)[NT][NCH]) {
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        const int ct = min(ct0 + 4 * q, ntiles - 1);
#pragma unroll
        for (int c = 0; c < NCH; ++c) wb[q][c] = *reinterpret_cast<const f32x4*>(Bt + ((size_t)ct * NCH + c) * 256 + 4 * lane);
    }
}
// acc[q] = A[16, 16 NCH] . W(tile q); A row-major in LDS (pitch lda floats)
template <int NT, int NCH>
__device__ __forceinline__ void rg_mma(const float* __restrict__ A, int lda, const f32x4 (&wb)[NT][NCH], int lane, f32x4 (&acc)[NT]) {
    const int kk = lane >> 4, i = lane & 15;
#pragma unroll
    for (int q = 0; q < NT; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(A + i * lda + 16 * c + 4 * kk);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < NT; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], wb[q][c][u], acc[q], 0, 0, 0);
    }
}

// grid = (row blocks, chains).  LDS: h [16][H+4], rh [16][H+4], z [16][H+4].
template <int NCH>  // H / 16
__global__ __launch_bounds__(RG_THREADS) void rg_fwd_kernel(const RowGruArgs g) {
    constexpr int NTG = (2 * NCH + 3) / 4, NTC = (NCH + 3) / 4;  // column tiles per wave: gates (2H wide), candidate (H)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int H = g.H, B = g.B, T = g.T, P = H + 4;
    float* s_h = sm;
    float* s_rh = s_h + RG_ROWS * P;
    float* s_z = s_rh + RG_ROWS * P;
    const int ch = blockIdx.y, m0 = blockIdx.x * RG_ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const RowGruChain& c = g.chain[ch];
    const size_t BH = (size_t)B * H;
    const int ntg = (2 * H) >> 4, ntc = H >> 4;
    f32x4 wg[NTG][NCH], wc[NTC][NCH];
    rg_load_w<NTG, NCH>(c.Wg_f, wave, ntg, lane, wg);
    rg_load_w<NTC, NCH>(c.Wc_f, wave, ntc, lane, wc);
    // state entering the sequence
    for (int e = tid; e < RG_ROWS * H; e += RG_THREADS) {
        const int m = e / H, k = e % H;
        s_h[m * P + k] = (m0 + m < B) ? c.h[(size_t)(m0 + m) * H + k] : 0.f;
    }
    __syncthreads();
    const int gq = lane >> 4, jj = lane & 15;
    // The additive inputs of a step do not depend on the recurrence: those of step s + 1 are requested while step s
    // computes (otherwise every phase ends with an exposed HBM round trip: 14 us per step instead of ~6).
    float pg[NTG][4], pc[NTC][4];
    auto prefetch = [&](int s) {
        const int t = c.reverse ? T - 1 - s : s;
#pragma unroll
        for (int q = 0; q < NTG; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 16 * (wave + 4 * q) + jj, mr = m0 + 4 * gq + r;
                pg[q][r] = (c.gate_inputs && s < T && wave + 4 * q < ntg && mr < B)
                               ? c.gate_inputs[(size_t)t * 2 * BH + (size_t)mr * 2 * H + n] : 0.f;
            }
#pragma unroll
        for (int q = 0; q < NTC; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 16 * (wave + 4 * q) + jj, mr = m0 + 4 * gq + r;
                pc[q][r] = (c.inputs && s < T && wave + 4 * q < ntc && mr < B) ? c.inputs[(size_t)t * BH + (size_t)mr * H + n] : 0.f;
            }
    };
    prefetch(0);
    for (int s = 0; s < T; ++s) {
        const int t = c.reverse ? T - 1 - s : s;
        float cg[NTG][4], cc_in[NTC][4];
#pragma unroll
        for (int q = 0; q < NTG; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) cg[q][r] = pg[q][r];
#pragma unroll
        for (int q = 0; q < NTC; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) cc_in[q][r] = pc[q][r];
        prefetch(s + 1);
        // ---- gates: [16, H] . Wg [H, 2H]
        {
            f32x4 acc[NTG];
            rg_mma<NTG, NCH>(s_h, P, wg, lane, acc);
#pragma unroll
            for (int q = 0; q < NTG; ++q) {
                const int ct = wave + 4 * q;
                if (ct >= ntg) continue;
                const int n = 16 * ct + jj;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 4 * gq + r, mr = m0 + m;
                    if (mr >= B) continue;
                    const float pre = acc[q][r] + cg[q][r];
                    const float gt = rg_sigmoid(pre);
                    if (n < H) {
                        c.z[(size_t)t * BH + (size_t)mr * H + n] = gt;
                        s_z[m * P + n] = gt;
                    } else {
                        const int j = n - H;
                        c.r[(size_t)t * BH + (size_t)mr * H + j] = gt;
                        const float rh = gt * s_h[m * P + j];
                        c.rh[(size_t)t * BH + (size_t)mr * H + j] = rh;
                        s_rh[m * P + j] = rh;
                    }
                }
            }
        }
        __syncthreads();
        // ---- candidate: [16, H] (r*h) . Wc [H, H], state blend
        {
            f32x4 acc[NTC];
            rg_mma<NTC, NCH>(s_rh, P, wc, lane, acc);
#pragma unroll
            for (int q = 0; q < NTC; ++q) {
                const int ct = wave + 4 * q;
                if (ct >= ntc) continue;
                const int n = 16 * ct + jj;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 4 * gq + r, mr = m0 + m;
                    if (mr >= B) continue;
                    const float pre = acc[q][r] + cc_in[q][r];
                    const float cc = tanhf(pre);
                    const float z = s_z[m * P + n], hp = s_h[m * P + n];
                    float hn = z * cc + (1.f - z) * hp;
                    if (g.mask) {
                        const float mk = g.mask[(size_t)t * B + mr];
                        hn = mk * hn + (1.f - mk) * hp;
                    }
                    c.c[(size_t)t * BH + (size_t)mr * H + n] = cc;
                    c.h[(size_t)(s + 1) * BH + (size_t)mr * H + n] = hn;
                    s_h[m * P + n] = hn;  // (only this lane reads element (m, n) of h in this phase)
                }
            }
        }
        __syncthreads();
    }
}

// Backward: LDS cur [16][H+4] (gradient wrt the state leaving step s), dC [16][H+4], dG [16][2H+4], dhp [16][H+4].
template <int NCH>
__global__ __launch_bounds__(RG_THREADS) void rg_bwd_kernel(const RowGruArgs g) {
    constexpr int NTC = (NCH + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int H = g.H, B = g.B, T = g.T, P = H + 4, P2 = 2 * H + 4;
    float* s_cur = sm;
    float* s_dC = s_cur + RG_ROWS * P;
    float* s_dhp = s_dC + RG_ROWS * P;
    float* s_dG = s_dhp + RG_ROWS * P;
    const int ch = blockIdx.y, m0 = blockIdx.x * RG_ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const RowGruChain& c = g.chain[ch];
    const size_t BH = (size_t)B * H;
    const int ntc = H >> 4;
    f32x4 wx[NTC][NCH], wy[NTC][2 * NCH];
    rg_load_w<NTC, NCH>(c.Wc_r, wave, ntc, lane, wx);
    rg_load_w<NTC, 2 * NCH>(c.Wg_r, wave, ntc, lane, wy);
    for (int e = tid; e < RG_ROWS * H; e += RG_THREADS) {
        const int m = e / H, k = e % H;
        s_cur[m * P + k] = (m0 + m < B) ? c.dh[(size_t)T * BH + (size_t)(m0 + m) * H + k] : 0.f;
    }
    __syncthreads();
    const int gq = lane >> 4, jj = lane & 15;
    // Saved activations and the consumers' gradients do not depend on the recurrence: those of step s - 1 are requested
    // while step s computes.  Element e = tid + 256 i of the [16, H] block (i < NCH) for the elementwise half, the
    // (tile, row) elements of this lane for the two epilogues.
    float p_hp[NCH], p_z[NCH], p_c[NCH], p_r[NTC][4], p_hp2[NTC][4], p_slot[NTC][4];
    auto prefetch = [&](int s) {
        const int t = c.reverse ? T - 1 - s : s;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int e = tid + RG_THREADS * i, m = e / H, k = e % H, mr = m0 + m;
            const bool ok = s >= 0 && mr < B;
            const size_t ix = ok ? (size_t)t * BH + (size_t)mr * H + k : 0;
            p_hp[i] = ok ? c.h[(size_t)s * BH + (size_t)mr * H + k] : 0.f;
            p_z[i] = ok ? c.z[ix] : 0.f;
            p_c[i] = ok ? c.c[ix] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < NTC; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 16 * (wave + 4 * q) + jj, mr = m0 + 4 * gq + r;
                const bool ok = s >= 0 && wave + 4 * q < ntc && mr < B;
                p_r[q][r] = ok ? c.r[(size_t)t * BH + (size_t)mr * H + n] : 0.f;
                p_hp2[q][r] = ok ? c.h[(size_t)s * BH + (size_t)mr * H + n] : 0.f;
                p_slot[q][r] = ok ? c.dh[(size_t)s * BH + (size_t)mr * H + n] : 0.f;
            }
    };
    prefetch(T - 1);
    for (int s = T - 1; s >= 0; --s) {
        const int t = c.reverse ? T - 1 - s : s;
        float q_hp[NCH], q_z[NCH], q_c[NCH], q_r[NTC][4], q_hp2[NTC][4], q_slot[NTC][4];
#pragma unroll
        for (int i = 0; i < NCH; ++i) { q_hp[i] = p_hp[i]; q_z[i] = p_z[i]; q_c[i] = p_c[i]; }
#pragma unroll
        for (int q = 0; q < NTC; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) { q_r[q][r] = p_r[q][r]; q_hp2[q][r] = p_hp2[q][r]; q_slot[q][r] = p_slot[q][r]; }
        prefetch(s - 1);
        // ---- elementwise half: dC, dG_z, direct share of dh_prev
#pragma unroll
        for (int ei = 0; ei < NCH; ++ei) {
            const int e = tid + RG_THREADS * ei;
            const int m = e / H, k = e % H, mr = m0 + m;
            float dC = 0.f, dGz = 0.f, dhp = 0.f;
            if (mr < B) {
                const size_t i = (size_t)t * BH + (size_t)mr * H + k;
                float dh = s_cur[m * P + k];
                const float hp = q_hp[ei];
                float direct = 0.f;
                if (g.mask) {
                    const float mk = g.mask[(size_t)t * B + mr];
                    direct = dh * (1.f - mk);
                    dh *= mk;
                }
                const float z = q_z[ei], cc = q_c[ei];
                dC = dh * z * (1.f - cc * cc);
                dGz = dh * (cc - hp) * z * (1.f - z);
                dhp = dh * (1.f - z) + direct;
                c.dC[i] = dC;
                c.dG[(size_t)t * 2 * BH + (size_t)mr * 2 * H + k] = dGz;
            }
            s_dC[m * P + k] = dC;
            s_dG[m * P2 + k] = dGz;
            s_dhp[m * P + k] = dhp;
        }
        __syncthreads();
        // ---- X: d(r*h) = dC . Wc^T ; dG_r = d(rh) h r (1 - r) ; dh_prev += d(rh) r
        {
            f32x4 acc[NTC];
            rg_mma<NTC, NCH>(s_dC, P, wx, lane, acc);
#pragma unroll
            for (int q = 0; q < NTC; ++q) {
                const int ct = wave + 4 * q;
                if (ct >= ntc) continue;
                const int n = 16 * ct + jj;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 4 * gq + r, mr = m0 + m;
                    float dGr = 0.f;
                    if (mr < B) {
                        const float x = acc[q][r];
                        const float r_ = q_r[q][r];
                        const float hp = q_hp2[q][r];
                        dGr = x * hp * r_ * (1.f - r_);
                        c.dG[(size_t)t * 2 * BH + (size_t)mr * 2 * H + H + n] = dGr;
                        s_dhp[m * P + n] += x * r_;
                    }
                    s_dG[m * P2 + H + n] = dGr;
                }
            }
        }
        __syncthreads();
        // ---- Y: dh_prev += dG . Wg^T ; plus the gradient the consumers left in slot s
        {
            f32x4 acc[NTC];
            rg_mma<NTC, 2 * NCH>(s_dG, P2, wy, lane, acc);
#pragma unroll
            for (int q = 0; q < NTC; ++q) {
                const int ct = wave + 4 * q;
                if (ct >= ntc) continue;
                const int n = 16 * ct + jj;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 4 * gq + r, mr = m0 + m;
                    if (mr >= B) continue;
                    float* slot = c.dh + (size_t)s * BH + (size_t)mr * H + n;
                    const float tot = q_slot[q][r] + (s_dhp[m * P + n] + acc[q][r]);
                    *slot = tot;
                    s_cur[m * P + n] = tot;
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace

bool rowgru_supported(int T, int B, int H, int nchain) {
    return T >= 1 && B >= 1 && H >= 16 && H <= RG_MAXH && (H % 16) == 0 && nchain >= 1 && nchain <= 4;
}

template <int NCH>
static void rg_launch(int which, const RowGruArgs& g, dim3 grid, size_t lds, hipStream_t stream) {
    if (which == 0) hipLaunchKernelGGL((rg_fwd_kernel<NCH>), grid, dim3(RG_THREADS), lds, stream, g);
    else hipLaunchKernelGGL((rg_bwd_kernel<NCH>), grid, dim3(RG_THREADS), lds, stream, g);
}
static int rg_dispatch(int which, const RowGruArgs& g, hipStream_t stream) {
    if (!rowgru_supported(g.T, g.B, g.H, g.nchain)) return PH_ERR_UNSUPPORTED;
    const dim3 grid(ceil_div(g.B, RG_ROWS), g.nchain);
    const size_t lds = which == 0 ? sizeof(float) * 3 * RG_ROWS * (g.H + 4)
                                  : sizeof(float) * RG_ROWS * (3 * (g.H + 4) + 2 * g.H + 4);
    switch (g.H >> 4) {  // chunk count as a template parameter: the weight registers are indexed by constants only
        case 1: rg_launch<1>(which, g, grid, lds, stream); break;
        case 2: rg_launch<2>(which, g, grid, lds, stream); break;
        case 3: rg_launch<3>(which, g, grid, lds, stream); break;
        case 4: rg_launch<4>(which, g, grid, lds, stream); break;
        case 5: rg_launch<5>(which, g, grid, lds, stream); break;
        case 6: rg_launch<6>(which, g, grid, lds, stream); break;
        case 7: rg_launch<7>(which, g, grid, lds, stream); break;
        default: rg_launch<8>(which, g, grid, lds, stream); break;
    }
    return (int)hipGetLastError();
}

int rowgru_fwd_launch(const RowGruArgs& g, hipStream_t stream) { return rg_dispatch(0, g, stream); }
int rowgru_bwd_launch(const RowGruArgs& g, hipStream_t stream) { return rg_dispatch(1, g, stream); }
