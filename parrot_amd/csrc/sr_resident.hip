// SampleRNN generation: ONE resident launch per big frame -- the frame tier inside the sample kernel (gfx950).
//
// sr_persist.hip keeps the FS sample steps of a frame in one launch and leaves the frame tier (three_tier.py:382-450: a
// GRU step and the [D, FS*D] output projection, 54 MB of weights) to three launches in front of it: per frame ~63 us of
// sample steps and ~50 us of launches, launch gaps and weight prologue.  Here the kernel stays resident for all
// BFS / FS frames of a period and runs the frame tier itself, XCD-local like the sample steps: team x (one XCD, 32 CUs)
// owns streams 4x .. 4x+3, CU c of a team owns hidden columns [c D/32, (c+1) D/32) of everything.
//
//   * nothing of the tier fits on chip, so its weights are STREAMED, every XCD reading the same 54 MB per frame out of
//     the memory-side cache (tools/stream_probe.hip: 128 KB per CU in 3.7 us with all eight XCDs streaming, 1.1 TB/s per
//     XCD, 9 TB/s chip-wide).  What depends on the frame's last sample is short: r, z (element-wise: the gates'
//     recurrent product h . Wg does not depend on the new samples and the samples' share is xf . (Win . U), K = FS),
//     one hand-off of r*h, the candidate product (one 128 KB slice per CU), one hand-off of h', the first D columns
//     of the composed projection (one slice).  These two products load their slices when they are needed.
//   * the other FS + 1 slices of a frame (projection columns of sample steps 1 .. FS-1, the NEXT frame's h' . Wg) are
//     nobody's critical path: waves 4-7 of every workgroup are BACKGROUND waves.  They take no part in the hand-off
//     polls (waves 0-3 take all slots; a wave's loads return in order, so a wave that streams cannot poll), and in the
//     windows where the foreground waits -- the x2 and logits hand-offs of every sample step -- each of them moves its
//     K quarter of the current slice: global_load_lds_dwordx4 straight into a private LDS ring (1 KB per instruction,
//     up to 18 in flight per wave, issued one window ahead), then 4 x f32x4 FMAs per 16 bytes of weights against h' in
//     LDS; lanes fold with DPP-free xor shuffles, the four K quarters meet through 4 KB of LDS after the next
//     workgroup barrier, where wave 4 adds the bias and leaves the slice's [4 streams x D/32] result in LDS for the
//     foreground.  The products the sample steps wait for (L3, output layer) run on all eight waves as before.
//   * hand-offs: r*h and h' travel like x1 / x2 (16-byte EMPTY slots in the XCD's L2); frame_out never leaves the CU.
// Restates ops.py:356-393 (GRU step), three_tier.py:382-450 (frame tier), :452-515 (sample-level MLP), :809-832 (loop).
#include "sr_persist.h"

#include <stdlib.h>

#include "sr_common.h"

namespace {

constexpr int SRQ_MAXFS = 16;

template <int D>
struct SrqGeom {
    static constexpr int Q = SRP_Q;
    static constexpr int DC = D / 32, G = DC / 4, S = SRP_THREADS / G, KP = D / S;
    static constexpr int QC = Q / 32, GQ = QC / 4, SQ = SRP_THREADS / GQ, KQ = D / SQ;
    // background streaming: a slice = D K-rows x DC columns; background wave v owns K rows [v D/4, (v+1) D/4)
    static constexpr int KPC = 64 / G;          // K rows per 1 KB chunk (one LDS-DMA instruction of a wave)
    static constexpr int NCH = (D / 4) / KPC;   // chunks per slice and wave (1024: 32, 512: 8, 256: 2)
    static constexpr int CW = NCH / 2 + (NCH / 16 > 0 ? NCH / 16 : 1);  // chunks per window (18, 5, 2): two windows per step
    static constexpr int RING = CW;
};

template <int D>
__global__ __launch_bounds__(SRP_THREADS) void srq_kernel(const SrqArgs a) {
    using GE = SrqGeom<D>;
    constexpr int Q = GE::Q, DC = GE::DC, G = GE::G, S = GE::S, KP = GE::KP;
    constexpr int QC = GE::QC, GQ = GE::GQ, SQ = GE::SQ, KQ = GE::KQ;
    constexpr int KPC = GE::KPC, NCH = GE::NCH, CW = GE::CW, RING = GE::RING;
    constexpr int NT = (2 * D + SRP_THREADS - 1) / SRP_THREADS;  // hand-off slots per taking thread (waves 0-3 take)
    extern __shared__ __attribute__((aligned(16))) char srq_smem[];
    f32x4* act = reinterpret_cast<f32x4*>(srq_smem);        // [D]   the hand-off just taken; its first Q vectors double as
    f32x4* lg = act;                                         //       the team's logits (x2 is dead once the output product is summed)
    f32x4* red = act + D;                                    // [256] reduction scratch; doubles as the exp values of the draw
    float* ev = reinterpret_cast<float*>(red);
    f32x4* hact = red + 256;                                 // [D]   h' of the current frame (4 streams per vector)
    f32x4* gpre = hact + D;                                  // [2][DC]  h' . Wg (update | reset) for the next frame
    f32x4* pbl = gpre + 2 * DC;                              // [3][DC]  the big tier's share of the frame's additive inputs
    f32x4* gate = pbl + 3 * DC;                              // [2][DC]  z and the candidate's additive input, across the boundary
    f32x4* bgpart = gate + 2 * DC;                           // [2][4][G][4] K-quarter partial sums of a background slice
    f32x4* ring = bgpart + 2 * 4 * G * 4;                    // [4][RING][64] the background waves' private rings
    f32x4* fo = ring + 4 * RING * 64;                        // [FS][DC] this CU's columns of the composed frame projection
    float* tmp = reinterpret_cast<float*>(fo + a.FS * DC);   // [4 * DC] transposition scratch of the gather phase
    float* t2l = tmp + 4 * DC;                               // [Q][DC] this CU's columns of t2tbl[FS-1]
    float* wul = t2l + Q * DC;                               // [FS][3][DC] this CU's columns of Win . U
    SrpShared* sh = reinterpret_cast<SrpShared*>(wul + a.FS * 3 * DC);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool bgw = wave >= 4;        // background wave
    const int bv = wave & 3;           // its K quarter
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
    const int FS = a.FS, nsteps = a.nfr * FS;

    // ---- resident weight slices: L3 and Output -> registers; newest-sample table and Win . U columns -> LDS
    const int g = (tid >> 3) % G, s = 8 * (tid / (8 * G)) + (tid & 7);
    const int gq = (tid >> 3) % GQ, sq = 8 * (tid / (8 * GQ)) + (tid & 7);
    f32x4 w3[KP], w4[KQ];
#pragma unroll
    for (int kk = 0; kk < KP; ++kk)
        w3[kk] = *reinterpret_cast<const f32x4*>(a.W3 + (size_t)(kk * S + s) * D + cu * DC + 4 * g);
#pragma unroll
    for (int kk = 0; kk < KQ; ++kk)
        w4[kk] = *reinterpret_cast<const f32x4*>(a.W4 + (size_t)(kk * SQ + sq) * Q + cu * QC + 4 * gq);
    for (int idx = tid; idx < Q * (DC / 4); idx += SRP_THREADS) {
        const int q = idx / (DC / 4), c4 = idx % (DC / 4);
        reinterpret_cast<f32x4*>(t2l)[idx] =
            *reinterpret_cast<const f32x4*>(a.t2tbl + ((size_t)(FS - 1) * Q + q) * D + cu * DC + 4 * c4);
    }
    for (int idx = tid; idx < FS * 3 * DC; idx += SRP_THREADS) {
        const int p = idx / (3 * DC), j = (idx / DC) % 3, c = idx % DC;
        wul[idx] = a.winu[(size_t)p * 3 * D + (size_t)j * D + cu * DC + c];
    }
    // column this thread finishes in the reductions (threads tid < DC / tid < QC), and its index inside the CU's slice
    const int fin_h = cu * DC + 4 * (tid % G) + tid / G, own_c = 4 * (tid % G) + tid / G;
    const int fin_q = cu * QC + 4 * (tid % GQ) + tid / GQ;
    const float bias3 = tid < DC ? a.b3[fin_h] : 0.f;
    const float bias4 = tid < QC ? a.b4[fin_q] : 0.f;
    const float cb0 = tid < DC ? a.cb[fin_h] : 0.f;  // bias of the first projection slice (the others: off the path)
    const float half_q = (float)(Q / 2);

    const int t0 = a.tbase[0];
    if (tid < SRP_ROWS * FS) {
        const int r = tid / FS, pos = tid % FS;
        const int b = min(team * SRP_ROWS + r, a.B - 1);
        sh->hist[r][pos] = a.samples[(size_t)b * a.len + t0 - FS + pos];
    }
    // the frame tier's state of this team's streams (left by the previous period's launch, or the learned h0)
    for (int k = tid; k < D; k += SRP_THREADS) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < SRP_ROWS; ++r) v[r] = a.frm_h[(size_t)min(team * SRP_ROWS + r, a.B - 1) * D + k];
        hact[k] = v;
    }
    // team exchange buffers ([D] f32x4 each, 4 streams per vector): x1, x2, r*h, h' (two, by frame parity), logits [Q]
    float* xbase = a.ws + SRP_SYNC_WORDS + (size_t)team * srp_team_vecs(D, Q) * 4;
    const __amdgpu_buffer_rsrc_t xr = srp_rsrc(xbase);
    f32x4* x1 = reinterpret_cast<f32x4*>(xbase);
    f32x4* x2 = x1 + D;
    f32x4* rx = x2 + D;
    f32x4* hx = rx + D;
    f32x4* lb = hx + 2 * D;
    __syncthreads();
    // Slot life cycle as in sr_persist.hip: every launch leaves all slots EMPTY except the logits of its last step, which
    // their new owners empty right here.  The chain of a frame is logits(last step) -> r*h -> h' -> x1 -> x2 -> logits ...;
    // a buffer is emptied by its owner once the owner has taken the NEXT buffer of the chain from all 32 CUs, always by
    // threads that later publish -- behind s_waitcnt vmcnt(0) -- something the readers take before they look again.
    if (tid < QC) lb[fin_q] = srp_empty();
    f32x4 h_own = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (tid < DC) h_own = hact[fin_h];

    // ---- the first frame's gate pre-activations h . Wg (nobody computed them ahead): two streamed slices
    {
        f32x4 v;
        srp_stream_layer<KP, G>(a.Wg + cu * DC, 2 * D, hact, red, v, tid);
        if (tid < DC) gpre[own_c] = v;
        __syncthreads();
        srp_stream_layer<KP, G>(a.Wg + D + cu * DC, 2 * D, hact, red, v, tid);
        if (tid < DC) gpre[DC + own_c] = v;
        __syncthreads();
    }

    // part = sum_{pos < FS-1} t2tbl[pos][sample[t - FS + pos]] for this CU's columns: what of step n's L2 pre-activation is
    // known one step early.  Wave 0 alone (no workgroup barrier: the background waves are busy elsewhere); threads
    // tid < DC keep it as one f32x4 (4 streams) of column fin_h.
    f32x4 part = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto make_part = [&](int n) {
        if (wave == 0) {
            for (int idx = lane; idx < SRP_ROWS * DC; idx += 64) {
                const int r = idx / DC, c = idx % DC, col = cu * DC + c;
                float acc = 0.f;
                for (int pos = 0; pos < FS - 1; ++pos) {
                    const int q = sh->hist[r][n + pos];
                    acc += a.t2tbl[((size_t)pos * Q + q) * D + col];
                }
                tmp[c * SRP_ROWS + r] = acc;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (tid < DC) part = reinterpret_cast<const f32x4*>(tmp)[own_c];
            __builtin_amdgcn_wave_barrier();
        }
    };
    // the big tier's share of frame f's additive inputs (z | r | candidate) for this thread's column, 4 streams each
    auto load_pbig = [&](int f) {
        if (tid < DC) {
            f32x4 pz, pr, pc;
#pragma unroll
            for (int r = 0; r < SRP_ROWS; ++r) {
                const float* pb = a.pbig + (size_t)min(team * SRP_ROWS + r, a.B - 1) * a.ld_pbig + (size_t)f * 3 * D + fin_h;
                pz[r] = pb[0]; pr[r] = pb[D]; pc[r] = pb[2 * D];
            }
            pbl[own_c] = pz; pbl[DC + own_c] = pr; pbl[2 * DC + own_c] = pc;
        }
    };
    make_part(0);
    load_pbig(0);

    // ---- background program (waves 4-7): the frame's off-path slices, one chunk sequence per wave
    f32x4* ring_v = ring + bv * RING * 64;
    const int bg_total = (FS + 1) * NCH;  // projection slices 1 .. FS-1, then Wg (update), Wg (reset)
    int bg_issued = 0, bg_done = 0, bg_complete = -1, bg_finished = -1;
    auto bg_issue = [&](int n) {
#pragma unroll 1
        for (int i = 0; i < n && bg_issued < bg_total; ++i, ++bg_issued) {
            const int j = bg_issued / NCH, c = bg_issued % NCH;
            const int k = bv * (D / 4) + c * KPC + lane / G;
            const float* src = j < FS - 1 ? a.Pout + (size_t)k * FS * D + (size_t)(j + 1) * D
                                          : a.Wg + (size_t)k * 2 * D + (size_t)(j - (FS - 1)) * D;
            src += cu * DC + 4 * (lane % G);
            const int slot = __builtin_amdgcn_readfirstlane(bg_issued % RING);
            __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(ring_v + slot * 64), 16, 0, 0);
        }
    };
    // A window's chunks of ONE slice: products into registers, lanes that share a column group folded with xor shuffles,
    // the wave's partial added to its own entry of bgpart (first == the slice's first chunk: the entry is overwritten).
    auto bg_run = [&](int j, int c0, int c1) {
        f32x4 bacc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bacc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int c = c0; c < c1; ++c) {
            const f32x4 w = ring_v[((j * NCH + c) % RING) * 64 + lane];
            const f32x4 av = hact[bv * (D / 4) + c * KPC + lane / G];
#pragma unroll
            for (int r = 0; r < 4; ++r) bacc[r] += av[r] * w;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int off = G; off < 64; off <<= 1) {
                f32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = __shfl_xor(bacc[r][q], off, 64);
                bacc[r] += o;
            }
        }
        if (lane < G) {
            f32x4* dst = bgpart + (((j & 1) * 4 + bv) * G + lane) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[r] = c0 == 0 ? bacc[r] : dst[r] + bacc[r];
        }
    };
    auto bg_consume = [&]() {  // everything issued so far has had a whole foreground phase to land
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        while (bg_done < bg_issued) {
            const int j = bg_done / NCH, c0 = bg_done % NCH;
            const int c1 = min(NCH, c0 + (bg_issued - bg_done));
            bg_run(j, c0, c1);
            bg_done += c1 - c0;
            if (c1 == NCH) bg_complete = j;  // this wave's K quarter of slice j is complete
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // wave 4, one window (= at least one workgroup barrier) after the four quarters of a slice were left in bgpart
    auto bg_finish = [&]() {
        while (bg_finished < bg_complete) {
            const int j = ++bg_finished;
            if (wave == 4 && lane < DC) {
                const int c = lane, gg = c / 4, cc = c % 4;
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = 0.f;
#pragma unroll
                    for (int vv = 0; vv < 4; ++vv) t += bgpart[(((j & 1) * 4 + vv) * G + gg) * 4 + r][cc];
                    o[r] = t;
                }
                if (j < FS - 1) fo[(j + 1) * DC + c] = o + a.cb[(size_t)(j + 1) * D + cu * DC + c];
                else gpre[(j - (FS - 1)) * DC + c] = o;
            }
        }
    };

    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(sync + 600);
    const bool timing = a.timing && team == 0 && cu == 0 && tid == 0;
    auto stamp = [&](int f, int q) { if (timing && f == 1 && q < 96) stamps[q] = srp_clock(); };
    auto leave = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };  // no LDS-DMA may outlive the workgroup

    for (int f = 0; f < a.nfr; ++f) {
        // =================== frame boundary: the GRU step of the frame tier and its first projection slice ===================
        stamp(f, 0);
        if (tid < DC) {
            // additive inputs: the big tier's share (prefetched) + the samples' share xf . (Win . U), K = FS
            f32x4 pinz = pbl[own_c], pinr = pbl[DC + own_c], pinc = pbl[2 * DC + own_c];
            for (int p = 0; p < FS; ++p) {
                const float wz = wul[(p * 3 + 0) * DC + own_c], wr = wul[(p * 3 + 1) * DC + own_c], wc = wul[(p * 3 + 2) * DC + own_c];
#pragma unroll
                for (int r = 0; r < SRP_ROWS; ++r) {
                    const float xf = ((float)sh->hist[r][f * FS + p] / half_q - 1.0f) * 2.0f;
                    pinz[r] = fmaf(xf, wz, pinz[r]); pinr[r] = fmaf(xf, wr, pinr[r]); pinc[r] = fmaf(xf, wc, pinc[r]);
                }
            }
            const f32x4 pz = gpre[own_c] + pinz, pr = gpre[DC + own_c] + pinr;
            f32x4 rh, zg;
#pragma unroll
            for (int r = 0; r < SRP_ROWS; ++r) {
                zg[r] = ph_sigmoid(pz[r]);
                rh[r] = ph_sigmoid(pr[r]) * h_own[r];
            }
            gate[own_c] = zg;
            gate[DC + own_c] = pinc;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            rx[fin_h] = rh;
        }
        if (!bgw)
            srp_take_n<NT>(xr, (unsigned)(2 * D) * 16u, tid, SRP_THREADS / 2, D, act, abort_, sh);
        __syncthreads();
        if (!sh->ok) { leave(); return; }
        stamp(f, 1);
        // r*h complete => every CU has picked the previous frame's last sample: done with its logits
        if (f > 0 && tid < QC) lb[fin_q] = srp_empty();
        {
            f32x4 v;
            srp_stream_layer<KP, G>(a.Wc + cu * DC, D, act, red, v, tid);
            if (tid < DC) {
                const f32x4 zg = gate[own_c], pinc = gate[DC + own_c];
#pragma unroll
                for (int r = 0; r < SRP_ROWS; ++r) {
                    const float cnd = tanhf(v[r] + pinc[r]);
                    h_own[r] = zg[r] * cnd + (1.f - zg[r]) * h_own[r];
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                hx[(f & 1) * D + fin_h] = h_own;
            }
        }
        stamp(f, 2);
        if (!bgw)
            srp_take_n<NT>(xr, (unsigned)(3 * D + (f & 1) * D) * 16u, tid, SRP_THREADS / 2, D, hact, abort_, sh);
        __syncthreads();
        if (!sh->ok) { leave(); return; }
        stamp(f, 3);
        if (tid < DC) rx[fin_h] = srp_empty();  // h' complete => every CU is done with r*h
        if (bgw) {  // the frame's off-path slices start to move while the foreground streams its own
            bg_issued = bg_done = 0;
            bg_complete = bg_finished = -1;
            bg_issue(CW);
        }
        {
            f32x4 v;
            srp_stream_layer<KP, G>(a.Pout + cu * DC, FS * D, hact, red, v, tid);
            if (tid < DC) fo[own_c] = v + cb0;
        }
        if (f == a.nfr - 1 && tid < DC) {
#pragma unroll
            for (int r = 0; r < SRP_ROWS; ++r) {
                const int b = team * SRP_ROWS + r;
                if (b < a.B) a.frm_h[(size_t)b * D + fin_h] = h_own[r];
            }
        }
        __syncthreads();
        stamp(f, 4);

        // =================== the frame's FS sample steps ===================
        for (int i = 0; i < FS; ++i) {
            const int n = f * FS + i;
            const bool more = n + 1 < nsteps;
            stamp(f, 8 + i * 8);
            // ---- x1 = relu(part + projection + the newest sample's row): no product
            if (tid < DC) {
                f32x4 v = part + fo[i * DC + own_c];
#pragma unroll
                for (int r = 0; r < SRP_ROWS; ++r) v[r] += t2l[sh->hist[r][n + FS - 1] * DC + own_c];
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                x1[fin_h] = (f32x4){fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
            }
            stamp(f, 8 + i * 8 + 1);
            if (!bgw)
                srp_take_n<NT>(xr, 0u, tid, SRP_THREADS / 2, D, act, abort_, sh);
            else
                bg_finish();
            __syncthreads();
            if (!sh->ok) { leave(); return; }
            stamp(f, 8 + i * 8 + 2);
            // x1 complete => every CU is done with the logits of the previous step and (step 0) with this frame's h'
            if (i > 0 && tid < QC) lb[fin_q] = srp_empty();
            if (i == 0 && tid < DC) hx[(f & 1) * D + fin_h] = srp_empty();
            {
                f32x4 v;
                srp_layer<KP, G>(act, w3, red, v, tid);
                if (tid < DC) {
                    v += bias3;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    x2[fin_h] = (f32x4){fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
                }
            }
            stamp(f, 8 + i * 8 + 3);
            // ---- window 1: x2 arrives.  Foreground: next step's partial sum (wave 0), then the take; background: one window
            if (!bgw) {
                if (more) make_part(n + 1);
                if (i == FS - 1 && f + 1 < a.nfr) load_pbig(f + 1);
                srp_take_n<NT>(xr, (unsigned)D * 16u, tid, SRP_THREADS / 2, D, act, abort_, sh);
            } else {
                bg_finish();
                bg_consume();
                bg_issue(CW);
            }
            __syncthreads();
            if (!sh->ok) { leave(); return; }
            stamp(f, 8 + i * 8 + 4);
            if (tid < QC) {  // x2 complete => every CU is done with x1; emptied by the threads that publish the logits
#pragma unroll
                for (int q = 0; q < DC / QC; ++q) x1[cu * DC + tid * (DC / QC) + q] = srp_empty();
            }
            {
                f32x4 v;
                srp_layer<KQ, GQ>(act, w4, red, v, tid);
                if (tid < QC) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    lb[fin_q] = v + bias4;
                }
            }
            stamp(f, 8 + i * 8 + 5);
            // ---- window 2: the logits arrive
            if (!bgw) {
                if (tid < Q) lg[tid] = srp_take(xr, (unsigned)((5 * D + tid) * 16), abort_, sh);
            } else {
                bg_finish();
                bg_consume();
                bg_issue(CW);
            }
            __syncthreads();
            if (!sh->ok) { leave(); return; }
            stamp(f, 8 + i * 8 + 6);
            if (tid < DC) x2[fin_h] = srp_empty();  // logits complete => every CU is done with x2
            // ---- pick (every CU of the team, identical result): argmax with lowest-index ties, or the seeded draw
            const int t = t0 + n;
            if (wave < SRP_ROWS) {
                const int r = wave;
                float best;
                const int bi = srp_argmax_row<Q>(lg, r, lane, best);
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
                    sh->hist[r][FS + n] = pick;
                    const int b = team * SRP_ROWS + r;
                    if (cu == 0 && b < a.B) a.samples[(size_t)b * a.len + t] = pick;
                }
            } else {
                bg_finish();
            }
            if (a.logits && n == nsteps - 1 && cu == 0 && tid < Q) {
#pragma unroll
                for (int r = 0; r < SRP_ROWS; ++r) {
                    const int b = team * SRP_ROWS + r;
                    if (b < a.B) a.logits[(size_t)b * Q + tid] = lg[tid][r];
                }
            }
            __syncthreads();
            stamp(f, 8 + i * 8 + 7);
        }
    }
    leave();
}

size_t srq_lds_bytes(int D, int FS) {
    const int DC = D / 32, G = DC / 4, KPC = 64 / G, NCH = (D / 4) / KPC, CW = NCH / 2 + (NCH / 16 > 0 ? NCH / 16 : 1);
    return (size_t)(D + 256 + D + 7 * DC + 2 * 4 * G * 4 + 4 * CW * 64 + FS * DC) * 16 +
           (size_t)(4 * DC + SRP_Q * DC + FS * 3 * DC) * 4 + sizeof(SrpShared) + 64;
}

}  // namespace

bool srq_eligible(int B, int D, int Q, int FS, int nfr) {
    static const int enabled = getenv("PARROT_SR_RESIDENT") ? atoi(getenv("PARROT_SR_RESIDENT")) : 0;
    if (!enabled || !srp_eligible(B, D, Q, FS)) return false;
    if (FS < 2 || FS > SRQ_MAXFS || nfr < 1 || FS + nfr * FS > SRP_MAXHIST) return false;
    // the background waves get two windows of CW chunks per sample step: the frame's FS + 1 off-path slices must fit
    const int DC = D / 32, G = DC / 4, NCH = (D / 4) / (64 / G), CW = NCH / 2 + (NCH / 16 > 0 ? NCH / 16 : 1);
    if ((FS + 1) * NCH > 2 * FS * CW) return false;
    return srq_lds_bytes(D, FS) <= 160 * 1024;
}

int srq_prepare(int D, int FS) {
    const int lds = (int)srq_lds_bytes(D, FS);
    switch (D) {
        case 256: PH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(srq_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); break;
        case 512: PH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(srq_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); break;
        case 1024: PH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(srq_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); break;
        default: return PH_ERR_UNSUPPORTED;
    }
    return 0;
}

int srq_launch(const SrqArgs& a, hipStream_t stream) {
    if (!a.ws || a.nfr < 1 || a.FS + a.nfr * a.FS > SRP_MAXHIST) return PH_ERR_BADARG;
    const size_t lds = srq_lds_bytes(a.D, a.FS);
    const dim3 grid(SRP_TEAM * SRP_NTEAMS), block(SRP_THREADS);
    switch (a.D) {
        case 256: hipLaunchKernelGGL(srq_kernel<256>, grid, block, lds, stream, a); break;
        case 512: hipLaunchKernelGGL(srq_kernel<512>, grid, block, lds, stream, a); break;
        case 1024: hipLaunchKernelGGL(srq_kernel<1024>, grid, block, lds, stream, a); break;
        default: return PH_ERR_UNSUPPORTED;
    }
    return (int)hipGetLastError();
}
