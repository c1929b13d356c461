// Internal interface of the elementwise / reduction kernels.
#pragma once
#include "common.h"

struct GruStateBwdChain {
    const float* dh;     // [B,H] gradient wrt h_t
    const float* dh2;    // [B,H] optional second share of it (from the layers above) or null
    const float* dhx[3] = {nullptr, nullptr, nullptr};  // [B,H] optional further shares (K-split backward products, plans.hip bwd8)
    const float* hprev;  // [B,H]
    const float* z;      // [B,H]
    const float* c;      // [B,H]
    const float* mask;   // [B] or null
    float* dC;           // [B,H]
    float* dG;           // [B,2H]; only columns [0,H) (update gate) are written here
    float* dhprev;       // [B,H] accumulated (+=)
};
struct GruStateBwdArgs {
    GruStateBwdChain chain[4];
    int nchain, B, H;
};

// Elementwise half of the GRU backward step for row m of one chain (h_t = z*c + (1-z)*h_prev):
// dC = dh*z*(1-c^2), dG_z = dh*(c-h_prev)*z*(1-z), dh_prev += dh*(1-z).  Threads tid, tid+nthr, ... of the caller.
// Every operand is requested before anything is used, unconditionally (an optional share that is absent reads the first
// one again and is dropped by a select): written as `if (c.dh2) dh += c.dh2[i]` the compiler closed each conditional
// load with s_waitcnt vmcnt(0), one dependent round trip per share (round 6, tools/att_timing.py).  The sums keep their order.
__device__ __forceinline__ void gru_state_bwd_row(const GruStateBwdChain& c, int m, int H, int tid, int nthr) {
    // (addresses picked as INTEGERS: a select of pointers loaded from a dynamically indexed kernel-argument struct makes the
    // compiler copy the struct to scratch -- the wk_body lesson of round 4, tests/test_build_cpu.py)
    const unsigned long long a1 = (unsigned long long)c.dh, a2 = (unsigned long long)c.dh2, a3 = (unsigned long long)c.dhx[0],
                             a4 = (unsigned long long)c.dhx[1], a5 = (unsigned long long)c.dhx[2];
    const bool on2 = a2 != 0, on3 = a3 != 0, on4 = a4 != 0, on5 = a5 != 0;
    const float* p2 = reinterpret_cast<const float*>(on2 ? a2 : a1);
    const float* p3 = reinterpret_cast<const float*>(on3 ? a3 : a1);
    const float* p4 = reinterpret_cast<const float*>(on4 ? a4 : a1);
    const float* p5 = reinterpret_cast<const float*>(on5 ? a5 : a1);
    const unsigned long long am = (unsigned long long)c.mask;
    const float* pm = reinterpret_cast<const float*>(am ? am + 4ull * (unsigned)m : (unsigned long long)c.z);  // (c.z: any live word)
    for (int k = tid; k < H; k += nthr) {
        const size_t i = (size_t)m * H + k;
        const float x1 = c.dh[i], x2 = p2[i], x3 = p3[i], x4 = p4[i], x5 = p5[i];
        const float hp = c.hprev[i], z = c.z[i], cc = c.c[i], dhp = c.dhprev[i], mkv = *pm;
        float dh = x1;
        dh += on2 ? x2 : 0.f;
        dh += on3 ? x3 : 0.f;
        dh += on4 ? x4 : 0.f;
        dh += on5 ? x5 : 0.f;
        float dhp_direct = 0.f;
        if (c.mask) {
            dhp_direct = dh * (1.f - mkv);
            dh *= mkv;
        }
        c.dC[i] = dh * z * (1.f - cc * cc);
        c.dG[(size_t)m * 2 * H + k] = dh * (cc - hp) * z * (1.f - z);
        c.dhprev[i] = dhp + (dh * (1.f - z) + dhp_direct);
    }
}

// Up to four consecutive rows m0 .. m0 + nrows - 1 of one chain: the operands of ALL rows are in flight before the first
// row's results are stored (row after row, the stores of one row fence the loads of the next: four round trips).
__device__ __forceinline__ void gru_state_bwd_rows4(const GruStateBwdChain& c, int m0, int nrows, int H, int tid, int nthr) {
    // (addresses picked as INTEGERS: a select of pointers loaded from a dynamically indexed kernel-argument struct makes the
    // compiler copy the struct to scratch -- the wk_body lesson of round 4, tests/test_build_cpu.py)
    const unsigned long long a1 = (unsigned long long)c.dh, a2 = (unsigned long long)c.dh2, a3 = (unsigned long long)c.dhx[0],
                             a4 = (unsigned long long)c.dhx[1], a5 = (unsigned long long)c.dhx[2];
    const bool on2 = a2 != 0, on3 = a3 != 0, on4 = a4 != 0, on5 = a5 != 0;
    const float* p2 = reinterpret_cast<const float*>(on2 ? a2 : a1);
    const float* p3 = reinterpret_cast<const float*>(on3 ? a3 : a1);
    const float* p4 = reinterpret_cast<const float*>(on4 ? a4 : a1);
    const float* p5 = reinterpret_cast<const float*>(on5 ? a5 : a1);
    for (int k = tid; k < H; k += nthr) {
        float x1[4], x2[4], x3[4], x4[4], x5[4], hp[4], z[4], cc[4], dhp[4], mkv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + (r < nrows ? r : 0);
            const unsigned i = (unsigned)m * (unsigned)H + (unsigned)k;  // (32-bit offsets from uniform bases: no 64-bit VGPR addresses)
            x1[r] = c.dh[i]; x2[r] = p2[i]; x3[r] = p3[i]; x4[r] = p4[i]; x5[r] = p5[i];
            hp[r] = c.hprev[i]; z[r] = c.z[i]; cc[r] = c.c[i]; dhp[r] = c.dhprev[i];
            mkv[r] = c.mask ? c.mask[m] : 1.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r >= nrows) break;
            const int m = m0 + r;
            const unsigned i = (unsigned)m * (unsigned)H + (unsigned)k;
            float dh = x1[r];
            dh += on2 ? x2[r] : 0.f;
            dh += on3 ? x3[r] : 0.f;
            dh += on4 ? x4[r] : 0.f;
            dh += on5 ? x5[r] : 0.f;
            float dhp_direct = 0.f;
            if (c.mask) {
                dhp_direct = dh * (1.f - mkv[r]);
                dh *= mkv[r];
            }
            c.dC[i] = dh * z[r] * (1.f - cc[r] * cc[r]);
            c.dG[2u * (unsigned)m * (unsigned)H + (unsigned)k] = dh * (cc[r] - hp[r]) * z[r] * (1.f - z[r]);
            c.dhprev[i] = dhp[r] + (dh * (1.f - z[r]) + dhp_direct);
        }
    }
}

int gru_state_bwd_launch(const GruStateBwdArgs& g, hipStream_t stream);
int colsum_launch(const float* x, long long M, int N, int ld, float* out, int accumulate, hipStream_t stream);
int sumsq_launch(const float* x, size_t n, float* out, hipStream_t stream);
int adam_clip_launch(float* p, const float* g, float* m, float* v, size_t n, const float* gnorm_sq,
                     float grad_scale, float threshold, float lr_t, float b1, float b2, float eps,
                     hipStream_t stream);

// GMM output head of the sampling step (model.py:1024-1033, sample_gmm model.py:94-118).
int gmm_sample_launch(const float* mu, const float* sig_hat, const float* co_hat, int B, int O, int K, float bias,
                      float eps, const float* unif, const float* noise, float* x, int ldx, float* pi_out,
                      hipStream_t stream);

struct LstmStateBwdChain {
    const float* dh;      // [B,H] gradient wrt s_t
    const float* dh2;     // [B,H] optional second share of it or null
    const float* dh3 = nullptr;  // [B,H] optional further shares (second K halves of the split backward products) or null
    const float* dh4 = nullptr;
    const float* dh5 = nullptr;
    const float* dh6 = nullptr;
    float* dc;            // [B,H] carry: in gradient wrt c_t (from step t+1), out gradient wrt c_{t-1}
    const float* gates;   // [B,4H] saved activations i|f|o|g
    const float* c_prev;  // [B,H]
    const float* c_new;   // [B,H]
    float* dP;            // [B,4H] out: gradient wrt the pre-activations
    void* dP16 = nullptr; // [B,4H] bf16, optional: the same rows rounded to bf16 (nearest even) for the deferred weight-gradient
                          // products of a bf16-operand decoder -- written by the in-launch variant below (round 5)
};
struct LstmStateBwdArgs {
    LstmStateBwdChain chain[4];
    int nchain, B, H;
};

// Operands of column k of row m, all requested before any is used and unconditionally (a gradient share that is absent
// re-reads dh and is dropped by a select; written as `c.dh2 ? c.dh2[idx] : 0.f` every optional share was a dependent round
// trip of its own: round 6).  The sums keep their order.
struct LstmSbVals { float gi, gf, go, gg, cn, cp, dc, x1, x2, x3, x4, x5, x6; };
__device__ __forceinline__ LstmSbVals lstm_sb_load(const LstmStateBwdChain& c, int m, int H, int k) {
    const size_t idx = (size_t)m * H + k;
    const float* g = c.gates + (size_t)m * 4 * H;
    const float* p2 = c.dh2 ? c.dh2 : c.dh;
    const float* p3 = c.dh3 ? c.dh3 : c.dh;
    const float* p4 = c.dh4 ? c.dh4 : c.dh;
    const float* p5 = c.dh5 ? c.dh5 : c.dh;
    const float* p6 = c.dh6 ? c.dh6 : c.dh;
    LstmSbVals v;
    v.gi = g[k]; v.gf = g[H + k]; v.go = g[2 * H + k]; v.gg = g[3 * H + k];
    v.cn = c.c_new[idx]; v.cp = c.c_prev[idx]; v.dc = c.dc[idx];
    v.x1 = c.dh[idx]; v.x2 = p2[idx]; v.x3 = p3[idx]; v.x4 = p4[idx]; v.x5 = p5[idx]; v.x6 = p6[idx];
    return v;
}
// dP (i | f | o | g) and the carried dc of one column: out[0..3], returns dc_{t-1}
__device__ __forceinline__ float lstm_sb_math(const LstmStateBwdChain& c, const LstmSbVals& v, float (&out)[4]) {
    const float tc = tanhf(v.cn);
    const float dhv = v.x1 + (c.dh2 ? v.x2 : 0.f) + (c.dh3 ? v.x3 : 0.f) + (c.dh4 ? v.x4 : 0.f) + (c.dh5 ? v.x5 : 0.f) +
                      (c.dh6 ? v.x6 : 0.f);
    const float dcv = dhv * v.go * (1.f - tc * tc) + v.dc;
    out[0] = dcv * v.gg * v.gi * (1.f - v.gi);
    out[1] = dcv * v.cp * v.gf * (1.f - v.gf);
    out[2] = dhv * tc * v.go * (1.f - v.go);
    out[3] = dcv * v.gi * (1.f - v.gg * v.gg);
    return dcv * v.gf;
}

// Row m of one chain of the LSTM state backward (same arithmetic as lstm_state_bwd_kernel); two columns per thread in flight.
__device__ __forceinline__ void lstm_state_bwd_row(const LstmStateBwdChain& c, int m, int H, int tid, int nthr) {
    float* o = c.dP + (size_t)m * 4 * H;
    for (int k = tid; k < H; k += 2 * nthr) {
        const int k2 = k + nthr < H ? k + nthr : k;
        const LstmSbVals va = lstm_sb_load(c, m, H, k), vb = lstm_sb_load(c, m, H, k2);
        float ra[4], rb[4];
        const float da = lstm_sb_math(c, va, ra), db = lstm_sb_math(c, vb, rb);
        o[k] = ra[0]; o[H + k] = ra[1]; o[2 * H + k] = ra[2]; o[3 * H + k] = ra[3];
        c.dc[(size_t)m * H + k] = da;
        if (k2 != k) {
            o[k2] = rb[0]; o[H + k2] = rb[1]; o[2 * H + k2] = rb[2]; o[3 * H + k2] = rb[3];
            c.dc[(size_t)m * H + k2] = db;
        }
    }
}

// The same row for consumers INSIDE the launch (skinny.hip wkb_kernel): same arithmetic and thread -> column mapping, but
// the dP row is staged in LDS (`row`, 4H floats) and leaves as 16-byte write-through stores (a 4-byte sc1 store is one
// fabric write each); the caller drains them (s_waitcnt vmcnt(0) + barrier) before it arrives on the chain's flag.
__device__ __forceinline__ void lstm_state_bwd_row_pub(const LstmStateBwdChain& c, int m, int H, int tid, int nthr, float* row) {
    for (int k = tid; k < H; k += 2 * nthr) {
        const int k2 = k + nthr < H ? k + nthr : k;
        const LstmSbVals va = lstm_sb_load(c, m, H, k), vb = lstm_sb_load(c, m, H, k2);
        float ra[4], rb[4];
        const float da = lstm_sb_math(c, va, ra), db = lstm_sb_math(c, vb, rb);
        row[k] = ra[0]; row[H + k] = ra[1]; row[2 * H + k] = ra[2]; row[3 * H + k] = ra[3];
        c.dc[(size_t)m * H + k] = da;
        if (k2 != k) {
            row[k2] = rb[0]; row[H + k2] = rb[1]; row[2 * H + k2] = rb[2]; row[3 * H + k2] = rb[3];
            c.dc[(size_t)m * H + k2] = db;
        }
    }
    __syncthreads();
    float* o = c.dP + (size_t)m * 4 * H;
    for (int i = tid; i < H; i += nthr) {  // (4H / 4 vectors; H % 4 == 0 and 16-byte aligned rows are the launcher's conditions)
        const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * i);
        float* q = o + 4 * i;
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(q), "v"(v) : "memory");
    }
}

// The bf16 copy of the same row for the weight-gradient GEMMs that run after the scan (plain stores: nobody reads it in this
// launch).  Called AFTER the row has arrived on its chain's flag, from the staging row still in LDS, so that the products
// waiting for the row inside the launch do not wait for these stores as well.
__device__ __forceinline__ void lstm_state_bwd_row_copy16(const LstmStateBwdChain& c, int m, int H, int tid, int nthr, const float* row) {
    if (!c.dP16) return;
    __bf16* o16 = reinterpret_cast<__bf16*>(c.dP16) + (size_t)m * 4 * H;
    for (int i = tid; i < (H >> 1); i += nthr)  // 4H / 8 vectors of eight (H % 2 == 0; bf16 decoders have H % 32 == 0)
        *reinterpret_cast<bf16x8*>(o16 + 8 * i) = ph_bf16x8(*reinterpret_cast<const f32x4*>(row + 8 * i),
                                                            *reinterpret_cast<const f32x4*>(row + 8 * i + 4));
}

// Elementwise half of the LSTM backward step (ops.py:505-553 reversed).  dh: gradient wrt s_t; dc: carry
// (in: gradient wrt c_t from step t+1, out: gradient wrt c_{t-1}); gates [B,4H] = i|f|o|g; dP [B,4H] out.
int lstm_state_bwd_launch(const float* dh, const float* dh2, float* dc, const float* gates, const float* c_prev, const float* c_new,
                          float* dP, int B, int H, hipStream_t stream);

// _simple_norm of the reference (model.py:24-27): y = (x - mean) / (eps + std) over the last axis, population
// std, no affine.  fwd: y (may alias x) and sigma [R] are saved for the backward; if add_dst is given the
// normalised rows are also added into it (the Fork outputs are summed after normalisation, model.py:703-722).
int simple_norm_fwd_launch(const float* x, int ldx, float* y, int ldy, float* sigma, long long R, int N, float eps,
                           float* add_dst, int ld_add, hipStream_t stream);
// dx (may alias y or dy) = d/dx of the above given dy, y, sigma; accumulate: dx += instead of =.
int simple_norm_bwd_launch(const float* dy, int lddy, const float* y, int ldy, const float* sigma, float* dx, int lddx,
                           long long R, int N, float eps, int accumulate, hipStream_t stream);

// dst[r,:] = (base ? base[r,:] : 0) + sum_s simple_norm(src[s][r,:]); all matrices [R,N] with leading dimension N
// (the per-step form of the normalised Fork sums in the sampling scan, model.py:899-1006).
struct NormSumGroup {
    const float* src[4];
    int nsrc, N;
    const float* base;
    float* dst;
};
int norm_sum_launch(const NormSumGroup* groups, int ngroups, int R, float eps, hipStream_t stream);
