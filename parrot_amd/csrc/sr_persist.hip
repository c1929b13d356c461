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

#include "sr_common.h"

namespace {

// -DSRP_FINE_TIMERS: step 5 of a timed launch also stamps the inside of its two products (tools/sr_timing.py)
#ifdef SRP_FINE_TIMERS
#define SRP_FINE(slot) (timing && i == 5 ? stamps + (slot) : nullptr)
#else
#define SRP_FINE(slot) nullptr
#endif

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
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(sync + 600);
    // Team = the XCD this workgroup runs on; rank = its place among the XCD's 32 workgroups.  The dispatcher deals
    // workgroups to the XCDs round-robin -- workgroup i runs on XCD (i + c) % 8 with one c per launch --, so the workgroups
    // with equal blockIdx % 8 share an XCD and blockIdx / 8 numbers them 0 .. 31 without the census atomic of round 5 (a
    // cold L2 round trip in front of every load of the prologue; the census word is still counted, without waiting for
    // it).  c is NOT always 0: tools/xcc_map_probe.hip saw c = 0 in 5200 launches (eager, graphs, odd grids in front), but
    // a check `XCC_ID == blockIdx % 8` placed here aborted 9 of the 28 generation tests that this numbering passes (session
    // r06bv) -- so the team is read from the hardware register, never computed.  A placement that splits a blockIdx % 8
    // class over XCDs leaves a team incomplete: its polls time out and raise abort (the caller falls back to launches).
    const int team = srp_xcc();
    const int cu = (int)(blockIdx.x / SRP_NTEAMS);
    const bool timing = a.pad && blockIdx.x == 0 && tid == 0;
    if (timing) stamps[80] = srp_clock();
    if (tid == 0) __hip_atomic_fetch_add(sync + 256 + team * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);

    // ---- Prologue.  Everything the first step needs is requested before the first wait, in the order of need: the
    // carry of the previous launch (below), the newest sample's table slice (-> LDS), then the weight slices of L3 and
    // Output (-> registers), which the first step's products wait for where they use them.
    //
    // Carry: the sample history of this team and the table part of step 0's gather depend on the samples the PREVIOUS
    // launch produced -- through tbase -> samples -> table rows, three dependent cold round trips.  The previous launch
    // knows all of it when it ends, so its tail leaves both in the workspace (history [4][FS]; per CU the sums
    // [4][DC]) together with the sample index they are for; a launch whose t0 matches takes them from fixed
    // addresses (one round trip beside everything else), any other launch (the first of an utterance) walks the chain.
    // (tbase and the carry's sample index through the scalar cache: not in the queue of the vector loads, whose results
    // return in order -- the weight requests need not wait for them)
    typedef const int __attribute__((address_space(4))) * srp_cint;
    const int t0 = *(srp_cint)(unsigned long long)a.tbase + a.toff;
    int* carry = reinterpret_cast<int*>(a.ws + srp_carry_base(D, Q)) + (size_t)team * SRP_CARRY_WORDS;
    float* carry_sum = reinterpret_cast<float*>(carry + 16 + SRP_ROWS * SRP_CARRY_HIST) + (size_t)cu * SRP_ROWS * DC;
    const int carry_t = *(srp_cint)(unsigned long long)carry;
    // (threads past SRP_ROWS * FS repeat the last slot: same address, same value, same LDS word -- no divergent block the
    // compiler could sink the request into)
    const int hu = min(tid, SRP_ROWS * a.FS - 1), hr = hu / a.FS, hpos = hu % a.FS;
    const int hc = carry[16 + hr * SRP_CARRY_HIST + min(hpos, SRP_CARRY_HIST - 1)];
    constexpr int GU = SRP_ROWS * DC, GW0 = (SRP_THREADS - GU) / 64;  // first gathering wave (GU <= 128: waves 6-7 at D = 1024)
    const int gu = min(max(tid - GW0 * 64, 0), GU - 1), gr = gu / DC, gc = gu % DC;
    const int gb = min(team * SRP_ROWS + gr, a.B - 1);
    const float cs = carry_sum[gu];
    const float f0 = a.frame_out[(size_t)gb * a.ldf + cu * DC + gc];
    const int g = (tid >> 3) % G, s = 8 * (tid / (8 * G)) + (tid & 7);
    const int gq = (tid >> 3) % GQ, sq = 8 * (tid / (8 * GQ)) + (tid & 7);
    constexpr int T2V = Q * (DC / 4) / SRP_THREADS;  // f32x4 of the table slice per thread
    static_assert(T2V * SRP_THREADS == Q * (DC / 4), "table slice not a multiple of the workgroup");
    f32x4 tv[T2V];
#pragma unroll
    for (int j = 0; j < T2V; ++j) {
        const int idx = tid + j * SRP_THREADS, q = idx / (DC / 4), c4 = idx % (DC / 4);
        tv[j] = *reinterpret_cast<const f32x4*>(a.t2tbl + ((size_t)(a.FS - 1) * Q + q) * D + cu * DC + 4 * c4);
    }
    // column this thread finishes in the reductions (threads tid < DC / tid < QC)
    const int fin_h = cu * DC + 4 * (tid % G) + tid / G;
    const int fin_q = cu * QC + 4 * (tid % GQ) + tid / GQ;
    const float bias3 = a.b3[min(fin_h, D - 1)];
    const float bias4 = a.b4[min(fin_q, Q - 1)];
    __builtin_amdgcn_sched_barrier(0);
    f32x4 w3[KP], w4[KQ];
#pragma unroll
    for (int kk = 0; kk < KP; ++kk)
        w3[kk] = *reinterpret_cast<const f32x4*>(a.W3 + (size_t)(kk * S + s) * D + cu * DC + 4 * g);
#pragma unroll
    for (int kk = 0; kk < KQ; ++kk)
        w4[kk] = *reinterpret_cast<const f32x4*>(a.W4 + (size_t)(kk * SQ + sq) * Q + cu * QC + 4 * gq);
    __builtin_amdgcn_sched_barrier(0);
    // team exchange buffers ([D] f32x4 each, 4 streams per vector): x1, x2, and the logits ([Q] f32x4)
    float* xbase = a.ws + SRP_SYNC_WORDS + (size_t)team * srp_team_vecs(D, Q) * 4;
    const __amdgpu_buffer_rsrc_t xr = srp_rsrc(xbase);
    f32x4* x1 = reinterpret_cast<f32x4*>(xbase);
    f32x4* x2 = x1 + D;
    f32x4* lb = x2 + D;

    // Slot life cycle.  All slots are EMPTY when a plan is created (srp_init_ws) and every launch leaves them EMPTY again,
    // except the logits of its last step: their owners empty them once they hold the complete x1 of the NEXT launch's
    // first step (every CU that published into it has left the previous launch), like the logits of any other step.
    // A buffer is emptied by its owner as soon as the owner has taken the NEXT buffer of the chain from all 32 CUs (which
    // proves that all of them are done with this one), always by threads that later publish, behind an
    // s_waitcnt vmcnt(0), something the readers take before they look at the emptied buffer again.  No counted barrier.

    // part_i = frame_out[:, i] + sum_{pos < FS-1} t2tbl[pos][sample[t - FS + pos]] for this CU's DC columns: everything of
    // step i's L2 pre-activation that is known one step early (frame_out carries the composed projection and both
    // biases); threads tid < DC keep it as one f32x4 (4 streams) of column fin_h.
    // The gather: SRP_ROWS * DC sums of FS rows each, by the last waves of the workgroup (the first one publishes and
    // finishes the reductions).  Its rows are REQUESTED right after x1 has been taken and SUMMED behind the L3 product --
    // a memory latency in the product's shadow; requests are unconditional (clamped indices, selected sums): a load
    // inside a divergent block is closed by the compiler with s_waitcnt vmcnt(0).  The sum keeps the order of the
    // positions.  Result transposed through tmp; threads tid < DC pick it up behind the next workgroup barrier.
    f32x4 part = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool gwave = __builtin_amdgcn_readfirstlane(wave) >= GW0, gfast = a.FS - 1 <= SRP_GMAX;
    const bool glane = tid - GW0 * 64 >= 0 && tid - GW0 * 64 < GU;
    float gf = 0.f, gv[SRP_GMAX];
    const __amdgpu_buffer_rsrc_t tr = srp_rsrc(a.t2tbl);
    auto gather_rows = [&](int i) {
        int hq[SRP_GMAX];  // the history in one LDS round trip (left to itself the compiler reads, waits, requests: x 12)
#pragma unroll
        for (int pos = 0; pos < SRP_GMAX; ++pos) hq[pos] = sh->hist[gr][i + max(min(pos, a.FS - 2), 0)];
        __builtin_amdgcn_sched_barrier(0);
        // (buffer loads: table base in scalar registers, the position's table as the scalar offset -- no 64-bit address
        // per position kept alive across the sample loop)
#pragma unroll
        for (int pos = 0; pos < SRP_GMAX; ++pos) {
            const int pp = max(min(pos, a.FS - 2), 0);
            gv[pos] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                tr, (unsigned)(hq[pos] * D + cu * DC + gc) * 4u, (unsigned)(pp * Q * D) * 4u, 0));
        }
    };
    auto gather_request = [&](int i) {
        gf = a.frame_out[(size_t)gb * a.ldf + (size_t)i * D + cu * DC + gc];
        gather_rows(i);
    };
    auto rows_sum = [&](int i, float acc) {
        if (gfast) {
#pragma unroll
            for (int pos = 0; pos < SRP_GMAX; ++pos) acc = pos < a.FS - 1 ? acc + gv[pos] : acc;
        } else {
            for (int pos = 0; pos < a.FS - 1; ++pos)
                acc += a.t2tbl[((size_t)pos * Q + sh->hist[gr][i + pos]) * D + cu * DC + gc];
        }
        return acc;
    };
    auto gather_sum = [&](int i) {
        const float acc = rows_sum(i, gf);
        if (glane) tmp[gc * SRP_ROWS + gr] = acc;
    };
    auto take_part = [&]() { if (tid < DC) part = reinterpret_cast<const f32x4*>(tmp)[fin_h - cu * DC]; };
    if (tid == 0) sh->ok = 1;
#pragma unroll
    for (int j = 0; j < T2V; ++j) reinterpret_cast<f32x4*>(t2l)[tid + j * SRP_THREADS] = tv[j];
    if (carry_t == t0) {
        sh->hist[hr][hpos] = hc;
        if (glane) tmp[gc * SRP_ROWS + gr] = f0 + cs;
    } else {
        const int hv = a.samples[(size_t)min(team * SRP_ROWS + hr, a.B - 1) * a.len + t0 - a.FS + hpos];
        sh->hist[hr][hpos] = hv;
        __syncthreads();
        if (gwave) {
            gather_request(0);
            gather_sum(0);
        }
    }
    __syncthreads();
    take_part();
    if (timing) stamps[81] = srp_clock();

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
            // (behind the stores that emptied x2: none in front of step 0, whose wait would be for the weight slices)
            if (i > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            x1[fin_h] = (f32x4){fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
        }
        stamp(i, 1);
        {
            for (int k = tid; k < D; k += SRP_THREADS) act[k] = srp_take(xr, (unsigned)(k * 16), abort_, sh);
            __syncthreads();
            if (!sh->ok) return;
            stamp(i, 2);
            // x1(i) complete => every CU is done with the logits of step i-1 (i = 0: of the previous launch's last step)
            if (tid < QC) lb[fin_q] = srp_empty();
            if (more && gwave) gather_request(i + 1);
            f32x4 v;
            srp_layer<KP, G>(act, w3, red, v, tid, SRP_FINE(84));
            if (tid < DC) {
                v += bias3;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                x2[fin_h] = (f32x4){fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
            }
            stamp(i, 3);
            if (more && gwave) gather_sum(i + 1);
            stamp(i, 4);
        }
        {
            for (int k = tid; k < D; k += SRP_THREADS) act[k] = srp_take(xr, (unsigned)((D + k) * 16), abort_, sh);
            __syncthreads();
            if (!sh->ok) return;
            if (more) take_part();
            stamp(i, 5);
            if (tid < QC) {  // x2(i) complete => every CU is done with x1(i); emptied by the threads that publish lb
#pragma unroll
                for (int q = 0; q < DC / QC; ++q) x1[cu * DC + tid * (DC / QC) + q] = srp_empty();
            }
            f32x4 v;
            srp_layer<KQ, GQ>(act, w4, red, v, tid, SRP_FINE(88));
            if (tid < QC) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                lb[fin_q] = v + bias4;
            }
#ifdef SRP_FINE_TIMERS
            if (timing && i == 5) stamps[92] = srp_clock();
#endif
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
    // ---- the carry for the next launch (see the prologue): this team's last FS samples and, per CU, the table part of
    // the next launch's first gather -- requested here, stored at the very end
    const bool leave = gfast && a.FS <= SRP_CARRY_HIST;
    if (leave && gwave) gather_rows(a.nsteps);
    if (leave && cu == 0) {
        carry[16 + hr * SRP_CARRY_HIST + min(hpos, SRP_CARRY_HIST - 1)] = sh->hist[hr][a.nsteps + hpos];
        if (tid == 0) carry[0] = t0 + a.nsteps;
    }
    if (!leave && cu == 0 && tid == 0) carry[0] = -1;
    // ---- the next frame's frame-tier input for this team's streams (saves the launch that would compute it).  All
    // operands of an item are requested in one batch (unconditional loads, dummy addresses where an operand is absent).
    if (timing) stamps[82] = srp_clock();
    if (a.next_in) {
        constexpr int TW = SRP_GMAX + 1;
        const float half_q = (float)(Q / 2);
        const int NN = a.next_n > 0 ? a.next_n : D, NC = NN / 32;  // this CU's share: NC of the NN columns
        const bool gates = a.next_gpre != nullptr, wfast = a.FS <= TW;
        const float* bias_p = a.next_bias ? a.next_bias : a.next_Win;
        const float* gpre_p = gates ? a.next_gpre : a.next_Win;
        const float* h_p = gates ? a.next_h : a.next_Win;
        for (int idx0 = 0; idx0 < SRP_ROWS * NC; idx0 += SRP_THREADS) {
            const int idx = min(idx0 + tid, SRP_ROWS * NC - 1);
            const int r = idx / NC, col = cu * NC + idx % NC;
            const int b = team * SRP_ROWS + r, bb = min(b, a.B - 1);
            const bool live = idx0 + tid < SRP_ROWS * NC && b < a.B;
            const bool gcol = gates && col < 2 * D;
            const int hc = col < D ? col : min(col - D, D - 1);
            float wv[TW];
#pragma unroll
            for (int p = 0; p < TW; ++p) wv[p] = a.next_Win[(size_t)min(p, a.FS - 1) * NN + col];
            const float bv = bias_p[col];
            const float av = a.next_add[(size_t)bb * a.next_ld_add + col];
            const float gp = gpre_p[gcol ? (size_t)bb * 2 * D + col : 0];
            const float hv2 = h_p[gcol ? (size_t)bb * D + hc : 0];
            float acc = 0.f;
            if (wfast) {
                int hq[TW];
#pragma unroll
                for (int p = 0; p < TW; ++p) hq[p] = sh->hist[r][a.nsteps + min(p, a.FS - 1)];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < TW; ++p) {
                    const float xf = ((float)hq[p] / half_q - 1.0f) * 2.0f;
                    acc = p < a.FS ? fmaf(xf, wv[p], acc) : acc;
                }
            } else {
                for (int p = 0; p < a.FS; ++p) {
                    const float xf = ((float)sh->hist[r][a.nsteps + p] / half_q - 1.0f) * 2.0f;
                    acc = fmaf(xf, a.next_Win[(size_t)p * NN + col], acc);
                }
            }
            acc += (a.next_bias ? bv : 0.f) + av;
            if (!live) continue;
            if (gcol) {
                // the recurrent product h . Wg of the NEXT frame's gates is already there (it does not depend on the new
                // samples: the projection launch made it): finish the gates here -- one launch fewer per frame
                const float gt = ph_sigmoid(gp + acc);
                if (col < D) a.next_z[(size_t)b * D + col] = gt;          // update gate
                else a.next_rh[(size_t)b * D + col - D] = gt * hv2;       // reset gate * h
            } else {
                a.next_in[(size_t)b * NN + col] = acc;
            }
        }
    }
    if (leave && gwave) {
        const float acc = rows_sum(a.nsteps, 0.f);
        if (glane) carry_sum[gu] = acc;
    }
    if (timing) stamps[83] = srp_clock();
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

long long srp_ws_floats(int D, int Q) { return srp_carry_base(D, Q) + (long long)SRP_NTEAMS * SRP_CARRY_WORDS; }

int srp_init_ws(float* ws, int D, int Q) {
    // barrier / census / abort words zero, every hand-off slot EMPTY
    PH_CHECK(hipMemset(ws, 0, SRP_SYNC_WORDS * sizeof(float)));
    PH_CHECK(hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(ws + SRP_SYNC_WORDS), (int)SRP_EMPTY,
                          (size_t)SRP_NTEAMS * srp_team_vecs(D, Q) * 4));
    // no carry yet: the sample index it is for can never match
    PH_CHECK(hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(ws + srp_carry_base(D, Q)), -1, (size_t)SRP_NTEAMS * SRP_CARRY_WORDS));
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
