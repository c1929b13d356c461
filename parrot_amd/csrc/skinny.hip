// Recurrent-step GEMM with fused gate epilogues, hand-written for gfx950 (CDNA4).
//
// Shape regime: M = batch rows of one timestep (<= 64 per workgroup row-chunk), N = 2H / H / 4H
// gate columns, K = concatenation of up to four operand segments (previous state, attention
// context w, lower-layer states, fed-back output frame).  Each weight element is used by only M
// MACs, so the kernel is a weight-streaming kernel: weights go HBM/L2 -> VGPR exactly once per
// launch, no LDS staging (nothing is shared between waves), f32 MFMA 16x16x4 does the math
// (exact f32, the reference computes in floatX = float32, model.py:21).
//
// Decomposition (wave64, 8 waves = 512 threads per workgroup):
//   * one workgroup owns a 64(M) x 16(N) output tile; grid.x enumerates the 16-column tiles of
//     up to four independent jobs (e.g. the gate GEMMs of different layers), grid.y the 64-row
//     chunks of the batch;
//   * the K range is split round-robin in 16-deep chunks over the 8 waves (intra-workgroup
//     split-K); every wave keeps MB (<=4) 16x16 accumulators = all 64 rows of the tile;
//   * per chunk a lane (kk = lane>>4, i = lane&15) loads one 16-byte vector of A per 16-row block
//     (row i, k = kc+4kk..+3) and the matching B values, then issues 4*MB MFMAs, using vector
//     component u as the k = kc+4kk+u step.  Any bijection k <-> (step, kk) is legal because the
//     MFMA sums over kk and we sum over steps;
//   * partial tiles are reduced through LDS (32 KiB), then 256 threads run the fused epilogue:
//     sigmoid / tanh / state blend of the GRU (Blocks GatedRecurrent; twin in
//     sampleRNN/lib/ops.py:364-393), its backward counterpart, or the LSTM cell (ops.py:505-553).
#include "skinny.h"
#include "att_fwd_body.h"
#include "att_bwd_body.h"
#include "elementwise.h"

#include <stdio.h>
#include <stdlib.h>

#include <hip/hip_ext.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>

template <bool AL>
__device__ __forceinline__ f32x4 ld4k(const float* __restrict__ p, int k, int K) {
    // p already points at element k; returns elements k..k+3 with zero fill beyond K.
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (AL) {
        if (k < K) v = *reinterpret_cast<const f32x4*>(p);
    } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k + u < K) v[u] = p[u];
    }
    return v;
}

// Column mapping of MFMA column jj (0..15) of tile `tile` to a real output column.
__device__ __forceinline__ int sk_col(int epi, int H, int tile, int jj) {
    if (epi == SK_EPI_LSTM) return (jj >> 2) * H + tile * 4 + (jj & 3);  // gate-major columns
    return tile * 16 + jj;
}
__device__ __forceinline__ int sk_jcol(const SkJob& job, int tile, int jj) {
    return sk_col(job.colmode == 1 ? (int)SK_EPI_LSTM : job.epi, job.H, tile, jj);
}

template <int MB, bool AL>
__device__ __forceinline__ void sk_fetch(const SkSeg& sg, int kc, int m0, int M, int ncol, bool ncol_ok,
                                         int kk, int i, f32x4 (&a)[MB], f32x4& b) {
    const int k = kc + 4 * kk;
#pragma unroll
    for (int rb = 0; rb < MB; ++rb) {
        const int m = m0 + rb * 16 + i;
        a[rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (m < M) a[rb] = ld4k<AL>(sg.A + (size_t)m * sg.lda + k, k, sg.K);
    }
    b = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (ncol_ok) {
        if (sg.b_kcontig) {
            b = ld4k<AL>(sg.B + (size_t)ncol * sg.ldb + k, k, sg.K);
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (k + u < sg.K) b[u] = sg.B[(size_t)(k + u) * sg.ldb + ncol];
        }
    }
}

// Activation operand of the fast path: 1 = quad-contiguous loads + ds_bpermute into the MFMA lanes (see
// sk_fetch_fast / sk_a_unpermute), 0 = each lane loads its own row directly.
#ifndef SK_A_PERMUTE
#define SK_A_PERMUTE 1
#endif

// Register-buffer ring depth of the fast path (chunks in flight per wave).  Measured on MI355X, cfg2 training
// step fwd/bwd ms: depth 2: 55.9/79.9, 4: 57.4/80.0, 6: 61.9/82.5, 8: 60.7/87.9 -- more loads in flight do
// not help (the clamped tail refills add traffic), so the ping-pong pair stays.
#ifndef SK_EARLY_DESC
#define SK_EARLY_DESC 1
#endif
#ifndef SK_DEPTH
#define SK_DEPTH 2
#endif
#ifndef SK_PIPE
#define SK_PIPE 0
#endif
#ifndef SK_A_FRAG_PROBE
#define SK_A_FRAG_PROBE 0
#endif

template <int MB>
__device__ __forceinline__ void sk_mma(const f32x4 (&a)[MB], const f32x4& b, f32x4 (&acc)[MB]) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int rb = 0; rb < MB; ++rb)
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb][u], b[u], acc[rb], 0, 0, 0);
}

// Branch-free operand fetch of the fast path: every segment has K % 16 == 0 and 16-byte aligned
// rows, out-of-range rows / columns are clamped (their results are discarded by the epilogue).
// KC = weights stored [N][K] (K contiguous: one 16-byte load); otherwise [K][N] (4 dword loads).
template <int MB, int NB, int BM>
__device__ __forceinline__ void sk_fetch_fast(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                              int kc, const int (&mrow)[MB], const int (&ncol)[NB],
                                              const int (&btile)[NB], int kk, f32x4 (&a)[MB], f32x4 (&b)[NB]) {
    const int k = kc + 4 * kk;
#pragma unroll
    for (int rb = 0; rb < MB; ++rb) {
#if SK_A_FRAG_PROBE
        // TIMING PROBE ONLY (values are wrong): the load pattern of a fragment-major activation copy -- one contiguous
        // 1 KB block per (16-row block, 16-deep chunk), lanes in MFMA operand order, no lane permutation afterwards
        a[rb] = *reinterpret_cast<const f32x4*>(A + ((size_t)((mrow[rb] >> 4) * (lda >> 4) + (kc >> 4)) << 8) +
                                                ((threadIdx.x & 63) << 2));
#elif SK_A_PERMUTE
        // quad-contiguous mapping: lane l reads 16 B of row (l >> 2) at k-offset 4 * (l & 3), so every quad of
        // lanes covers one contiguous 64-B segment; sk_a_unpermute moves the quads to the MFMA lanes later.
        a[rb] = *reinterpret_cast<const f32x4*>(A + (size_t)mrow[rb] * lda + kc + 4 * (threadIdx.x & 3));
#else
        a[rb] = *reinterpret_cast<const f32x4*>(A + (size_t)mrow[rb] * lda + k);
#endif
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        if (BM == 2) {
            // fragment-major weights (sk_tile_weights): the 64 lanes read one contiguous 1 KB block
            b[nb] = *reinterpret_cast<const f32x4*>(B + (size_t)btile[nb] * ldb + ((size_t)(kc >> 4) << 8) +
                                                    ((threadIdx.x & 63) << 2));
        } else if (BM == 1) {
            b[nb] = *reinterpret_cast<const f32x4*>(B + (size_t)ncol[nb] * ldb + k);
        } else {
            const float* bp = B + (size_t)k * ldb + ncol[nb];
#pragma unroll
            for (int u = 0; u < 4; ++u) b[nb][u] = bp[(size_t)u * ldb];
        }
    }
}

// MFMA lane (kk, i) = kk * 16 + i takes the quad that lane 4 * i + kk loaded (see sk_fetch_fast).
template <int MB>
__device__ __forceinline__ void sk_a_unpermute(f32x4 (&a)[MB]) {
#if SK_A_PERMUTE && !SK_A_FRAG_PROBE
    const int lane = threadIdx.x & 63;
    const int src = (((lane & 15) << 2) | (lane >> 4)) << 2;  // byte address of the source lane
#pragma unroll
    for (int rb = 0; rb < MB; ++rb)
#pragma unroll
        for (int u = 0; u < 4; ++u)
            a[rb][u] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(a[rb][u])));
#endif
}

template <int MB, int NB>
__device__ __forceinline__ void sk_mma2(const f32x4 (&a)[MB], const f32x4 (&b)[NB], f32x4 (&acc)[MB][NB]) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int rb = 0; rb < MB; ++rb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                acc[rb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb][u], b[nb][u], acc[rb][nb], 0, 0, 0);
}

// ---- bf16 operand mode (b_kcontig == 3) --------------------------------------------------------------
// Weights come from a bf16 fragment-major copy (sk_tile_weights_bf16: one 1 KB block = 16 columns x 32 K-rows in
// v_mfma_f32_16x16x32_bf16 B-operand order), activations stay f32 in HBM and are rounded to bf16 in registers
// (quad-contiguous 32-byte loads, v_cvt_pk_bf16_f32, then the same ds_bpermute move as the f32 path, on 4 dwords
// instead of 8).  Per 32 K-rows a wave issues MB*NB MFMAs of 16 cycles instead of 8*MB*NB of 32 cycles: the f32
// matrix pipe, 55 % busy in sk_kernel<2,2>, drops out of the picture and half the weight bytes move.
template <int MB, int NB>
__device__ __forceinline__ void sk_fetch_bf16(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                              int kc, const int (&mrow)[MB], const int (&btile)[NB],
                                              f32x4 (&a)[MB][2], f32x4 (&b)[NB]) {
#pragma unroll
    for (int rb = 0; rb < MB; ++rb) {
        const float* p = A + (size_t)mrow[rb] * lda + kc + 8 * (threadIdx.x & 3);
        a[rb][0] = *reinterpret_cast<const f32x4*>(p);
        a[rb][1] = *reinterpret_cast<const f32x4*>(p + 4);
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
        b[nb] = *reinterpret_cast<const f32x4*>(B + (size_t)btile[nb] * ldb + ((size_t)(kc >> 5) << 8) +
                                                ((threadIdx.x & 63) << 2));
}

template <int MB, int NB>
__device__ __forceinline__ void sk_mma_bf16(const f32x4 (&a)[MB][2], const f32x4 (&b)[NB], f32x4 (&acc)[MB][NB]) {
    const int lane = threadIdx.x & 63;
    const int src = (((lane & 15) << 2) | (lane >> 4)) << 2;  // MFMA lane (kk, i) takes the quad lane 4 * i + kk loaded
    bf16x8 av[MB];
#pragma unroll
    for (int rb = 0; rb < MB; ++rb) {
        i32x4 q = __builtin_bit_cast(i32x4, ph_bf16x8(a[rb][0], a[rb][1]));
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = __builtin_amdgcn_ds_bpermute(src, q[u]);
        av[rb] = __builtin_bit_cast(bf16x8, q);
    }
#pragma unroll
    for (int rb = 0; rb < MB; ++rb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            acc[rb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[rb], __builtin_bit_cast(bf16x8, b[nb]), acc[rb][nb], 0, 0, 0);
}

// Wait (one lane) until *p >= target; relaxed agent-scope polls with a back-off.  Bounded: a
// producer that never arrives (it can only mean the dispatch order assumption of sk_launch_att broke) ends the kernel
// with a trap, which the host sees as a launch failure -- loud, never a silent wrong result.
__device__ __forceinline__ void sk_wait_flag(const unsigned* p, unsigned target) {
    unsigned it = 0;
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(const_cast<unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(8);
        if ((++it & 1023u) == 0 && wall_clock64() - t0 > 100000000ull) __builtin_trap();
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");  // (compiler ordering: the tail loads stay behind the poll)
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sk_rsrc(const void* p) {  // raw buffer over [p, p + 2 GB), p uniform
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* q = (void*)(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, 0x7fffffff, 0x00020000);
}

// Development: -DSK_TIMERS stamps every wave's phases (100 MHz wall clock) into sk_timer_buf (tools/sktimers.hip):
// [0] entry  [1] operands requested, K loop starts  [2] K loop done  [3] past the LDS meeting point  [4] epilogue stores issued
#ifdef SK_TIMERS
__constant__ unsigned long long* sk_timer_buf = nullptr;  // (constant address space: a scalar load, no vmcnt wait at a stamp)
#define SK_STAMP(i)                                                                                                  \
    do {                                                                                                             \
        if (sk_timer_buf && (threadIdx.x & 63) == 0) {                                                               \
            const size_t wg_ = blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z);       \
            sk_timer_buf[(wg_ * 16 + (threadIdx.x >> 6)) * 8 + (i)] = wall_clock64();                                \
        }                                                                                                            \
    } while (0)
#else
#define SK_STAMP(i) do { } while (0)
#endif

// Element offset m * ld + c of an epilogue operand: rows and leading dimensions are below 2^24 (sk_make_launch checks), so
// one full-rate v_mad_u32_u24 instead of the quarter-rate 64-bit multiply-add `(size_t)m * ld + c` compiles to; the
// epilogue's ~30 addresses per wave sit on the critical path in front of the K loop and behind the LDS meeting point.
__device__ __forceinline__ unsigned sk_off(int m, int ld, int c) { return __umul24((unsigned)m, (unsigned)ld) + (unsigned)c; }

// One workgroup: (16*MB rows) x (16*NB columns) output tile, K split over the SK_NW waves.
template <int MB, int NB, bool FAST>
__device__ __forceinline__ void sk_body(const SkJob& job, int tile0, f32x4* red) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kk = lane >> 4, i = lane & 15;
    const int m0 = blockIdx.y * (16 * MB);  // the launch picks MB / NB
    const int M = job.M, N = job.N;
    SK_STAMP(0);
#if SK_EARLY_DESC  // (default 1: 20.2 -> 19.3 us per gate / candidate launch pair, tools/sktimers.hip)
    // The descriptor fields the prologue needs, pulled into scalar registers in one batch: read where the code first
    // needs them they arrive through five or six dependent s_load / s_waitcnt rounds (a cold kernel-argument line
    // each time) spread over the branches in front of the K loop.  (Invariant loads: the later reads reuse these.)
    asm volatile("" ::"s"(job.M), "s"(job.N), "s"(job.epi), "s"(job.H), "s"(job.nseg), "s"(job.accumulate), "s"(job.colmode),
                 "s"(job.bias), "s"(job.add), "s"(job.out), "s"(job.e0), "s"(job.e1), "s"(job.o1), "s"(job.ld_add),
                 "s"(job.ldo), "s"(job.lde0), "s"(job.lde1), "s"(job.ldo1), "s"(job.wait_flag), "s"(job.seg[0].A),
                 "s"(job.seg[0].B), "s"(job.seg[0].lda), "s"(job.seg[0].ldb), "s"(job.seg[0].K), "s"(job.seg[0].b_kcontig),
                 "s"(job.seg[1].A), "s"(job.seg[1].B), "s"(job.seg[1].lda), "s"(job.seg[1].ldb), "s"(job.seg[1].K));
#endif
    if (m0 >= M) return;

    f32x4 acc[MB][NB];
#pragma unroll
    for (int rb = 0; rb < MB; ++rb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[rb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Epilogue operands (bias, additive input, previous state / gates, accumulate targets) are requested now by
    // the waves that will run the epilogue: nothing in this launch writes them, and their latency then hides
    // behind the K loop instead of adding a dependent memory round trip after the reduction.
    // Branch-free and unconditional: one uniform choice of four source pointers (a slot that is off reads element 0 of a
    // buffer that always exists), clamped indices, 17 loads issued back to back -- nothing here waits for anything.
    // (Rounds 1-5 requested them under `if (job.add)` / `switch (job.epi)`: the compiler closed every conditional block
    // with s_waitcnt vmcnt(0), four dependent round trips = 1.4 - 2.1 us during which the epilogue waves had not started
    // their K loop and for which the other waves then waited at the LDS meeting point: tools/sktimers.hip,
    // profiles/r06_sk_phase_timers.txt.)  Values of slots that are off, or of rows / columns outside the job, are never used.
    // Round 6: the reduction + epilogue is dealt to ALL waves where the tile allows it -- EG row groups per 16 x 16 block, a
    // wave takes RPW = 4 / EG of a lane's four accumulator rows -- instead of MB * NB waves doing four rows each while the
    // rest exit: half the instructions on the tail's critical path (1.1 us from the LDS meeting point to the last store
    // at 32 x 32 tiles, one wave per SIMD), half the request instructions in front of every wave's K loop, all waves alike.
    constexpr int NBLK = MB * NB;
    constexpr int EG = (SK_NW % NBLK == 0) ? (SK_NW / NBLK > 4 ? 4 : SK_NW / NBLK) : 1;
    constexpr int RPW = 4 / EG, EWAVES = NBLK * EG;
    const int eblk = wave % NBLK, er0 = (wave / NBLK) * RPW;  // this wave's block and first row (of a lane's four)
    float p_bias = 0.f, p_add[RPW], p_e0[RPW], p_e1[RPW], p_oc[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) p_add[r] = p_e0[r] = p_e1[r] = p_oc[r] = 0.f;
    const bool has_add = job.add != nullptr;
    if (wave < EWAVES) {
        const int rb_ = eblk / NB, tile_ = tile0 + eblk % NB;
        const int g_ = lane >> 4, jj_ = lane & 15;
        const int epi_ = job.epi, H_ = job.H;
        const int n_ = min(sk_jcol(job, tile_, jj_), N - 1);
        const float* dummy = epi_ == SK_EPI_BWD_RH ? job.o1 : job.out;  // (always a live buffer of this job)
        const float* q_add = has_add ? job.add : dummy;
        const int l_add = has_add ? job.ld_add : 0, c_add = has_add ? n_ : 0;
        const bool on0 = epi_ == SK_EPI_GRU_GATES || epi_ == SK_EPI_GRU_CAND || epi_ == SK_EPI_BWD_RH;
        const float* q_e0 = on0 ? job.e0 : dummy;
        const int l_e0 = on0 ? job.lde0 : 0;
        const int c_e0 = !on0 ? 0 : (epi_ == SK_EPI_GRU_GATES ? max(n_ - H_, 0) : n_);
        const bool on1 = epi_ == SK_EPI_GRU_CAND || epi_ == SK_EPI_BWD_RH || epi_ == SK_EPI_LSTM;
        const float* q_e1 = on1 ? job.e1 : dummy;
        const int l_e1 = on1 ? job.lde1 : 0;
        const int c_e1 = !on1 ? 0 : (epi_ == SK_EPI_LSTM ? ((jj_ >> 2) == 0 ? min(n_, H_ - 1) : 0) : n_);
        const bool onc = (epi_ == SK_EPI_LINEAR && job.accumulate == 1) || epi_ == SK_EPI_BWD_RH;
        const float* q_oc = !onc ? dummy : (epi_ == SK_EPI_BWD_RH ? job.o1 : job.out);
        const int l_oc = !onc ? 0 : (epi_ == SK_EPI_BWD_RH ? job.ldo1 : job.ldo);
        const int c_oc = onc ? n_ : 0;
        const float* q_b = job.bias ? job.bias : dummy;
        p_bias = q_b[job.bias ? n_ : 0];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int m = min(m0 + rb_ * 16 + 4 * g_ + er0 + r, M - 1);
            p_add[r] = q_add[sk_off(m, l_add, c_add)];
            p_e0[r] = q_e0[sk_off(m, l_e0, c_e0)];
            p_e1[r] = q_e1[sk_off(m, l_e1, c_e1)];
            p_oc[r] = q_oc[sk_off(m, l_oc, c_oc)];
        }
    }

    SK_STAMP(6);
    if (FAST) {
        // One flat sequence of 16-deep chunks over all segments, dealt round-robin to the waves.  The
        // loop body is straight-line code: segment descriptors sit in scalar registers and are picked
        // with selects, fetch indices are clamped to the wave's last chunk, and the weight layout is a
        // per-job constant (two copies of the loop).  That lets the compiler keep the next chunk's
        // loads in flight behind counted s_waitcnt vmcnt(N) while the current chunk feeds the MFMAs.
        int cend[SK_MAXSEG], slda[SK_MAXSEG], sldb[SK_MAXSEG];
        const float* sA[SK_MAXSEG];
        const float* sB[SK_MAXSEG];
        int total = 0;
        const int csh = job.seg[0].b_kcontig == 3 ? 5 : 4;  // K rows per chunk: 32 (bf16 operands) or 16
        const bool flagged = job.wait_flag != nullptr;       // (uniform) the last segment waits for its producers
        const int nsm = flagged ? job.nseg - 1 : job.nseg;   // segments of the main ring
#pragma unroll
        for (int s = 0; s < SK_MAXSEG; ++s) {
            const bool on = s < nsm;
            const int ss = on ? s : 0;
            if (on) total += job.seg[ss].K >> csh;
            cend[s] = total;
            sA[s] = job.seg[ss].A; sB[s] = job.seg[ss].B; slda[s] = job.seg[ss].lda; sldb[s] = job.seg[ss].ldb;
        }
        int mrow[MB], ncl[NB], btile[NB];
#pragma unroll
        for (int rb = 0; rb < MB; ++rb) mrow[rb] = min(m0 + rb * 16 + (SK_A_PERMUTE ? ((lane >> 2) & 15) : i), M - 1);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            ncl[nb] = min(sk_jcol(job, tile0 + nb, i), N - 1);
            btile[nb] = min(tile0 + nb, ((N + 15) >> 4) - 1);
        }
#ifdef SK_BLOCKED
        // contiguous range of chunks per wave: consecutive loads of a wave walk along the rows (whole 128-B lines)
        const int base_n = total / SK_NW, extra = total % SK_NW;
        const int mine = base_n + (wave < extra ? 1 : 0);
        const int first = wave * base_n + min(wave, extra);
        const int last = first + mine - 1;
        constexpr int STR = 1;
#else
        const int mine = (total - wave + SK_NW - 1) / SK_NW;  // chunks of this wave (dealt round-robin)
        const int first = wave;
        const int last = wave + (mine - 1) * SK_NW;
        constexpr int STR = SK_NW;
#endif
        auto run = [&](auto kc_tag) {
            constexpr int BM = decltype(kc_tag)::value;
            auto fetch = [&](int g, f32x4 (&a)[MB], f32x4 (&b)[NB]) {
                const float* A = sA[0];
                const float* B = sB[0];
                int lda = slda[0], ldb = sldb[0], beg = 0;
#pragma unroll
                for (int s = 0; s < SK_MAXSEG - 1; ++s) {
                    const bool nx = g >= cend[s];
                    A = nx ? sA[s + 1] : A; B = nx ? sB[s + 1] : B;
                    lda = nx ? slda[s + 1] : lda; ldb = nx ? sldb[s + 1] : ldb; beg = nx ? cend[s] : beg;
                }
                sk_fetch_fast<MB, NB, BM>(A, lda, B, ldb, (g - beg) << 4, mrow, ncl, btile, kk, a, b);
            };
#if SK_PIPE
            // Three-stage software pipeline, stages kept apart with scheduling barriers: [loads of chunk i+SK_DEPTH-1]
            // [lane permutation of chunk i+1 (ds_bpermute into its own registers)] [MFMAs of chunk i].  Left to itself
            // the scheduler parks every ds_bpermute right in front of the MFMA pair that consumes it, so each pair
            // waits out an LDS round trip (s_waitcnt lgkmcnt(0) eight times per chunk) and the matrix pipe starves;
            // it also hoists later chunks' permutations upwards, which drags their vmcnt waits along and collapses
            // the load prefetch distance.  Here the permutation of the next chunk flies while this chunk multiplies.
            {
                f32x4 ra[SK_DEPTH][MB], rb_[SK_DEPTH][NB], pa[2][MB];
#pragma unroll
                for (int dd = 0; dd < SK_DEPTH; ++dd) fetch(min(first + dd * STR, last), ra[dd], rb_[dd]);
#pragma unroll
                for (int rb = 0; rb < MB; ++rb) pa[0][rb] = ra[0][rb];
                sk_a_unpermute<MB>(pa[0]);
                int g = first;
                // one iteration = SK_DEPTH chunks (static slot indices); the tail runs with clamped fetches
                const int ngroups = (mine + SK_DEPTH - 1) / SK_DEPTH;
                for (int gr = 0; gr < ngroups; ++gr) {
#pragma unroll
                    for (int dd = 0; dd < SK_DEPTH; ++dd) {
                        const int nx = (dd + 1) % SK_DEPTH;
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int rb = 0; rb < MB; ++rb) pa[(dd + 1) & 1][rb] = ra[nx][rb];
                        sk_a_unpermute<MB>(pa[(dd + 1) & 1]);   // next chunk's lanes, consumed one stage later
                        __builtin_amdgcn_sched_barrier(0);
                        if (g + dd * STR <= last) sk_mma2<MB, NB>(pa[dd & 1], rb_[dd], acc);
                        __builtin_amdgcn_sched_barrier(0);
                        fetch(min(g + (SK_DEPTH + dd) * STR, last), ra[dd], rb_[dd]);
                    }
                    g += SK_DEPTH * STR;
                }
                return;
            }
#endif
            // Ring of SK_DEPTH register buffers, SK_DEPTH chunks per iteration, each slot refilled right after
            // it has fed the MFMAs: no register copies, so nothing in the body waits for loads it has just
            // issued, and SK_DEPTH chunks stay in flight per wave.
            f32x4 ra[SK_DEPTH][MB], rb_[SK_DEPTH][NB];
#pragma unroll
            for (int dd = 0; dd < SK_DEPTH; ++dd) fetch(min(first + dd * STR, last), ra[dd], rb_[dd]);
            int g = first;
            const int ngroups = mine / SK_DEPTH;
            for (int gr = 0; gr < ngroups; ++gr) {
#pragma unroll
                for (int dd = 0; dd < SK_DEPTH; ++dd) {
                    sk_a_unpermute<MB>(ra[dd]);
                    sk_mma2<MB, NB>(ra[dd], rb_[dd], acc);
                    fetch(min(g + (SK_DEPTH + dd) * STR, last), ra[dd], rb_[dd]);
                }
                g += SK_DEPTH * STR;
            }
            const int rem = mine - ngroups * SK_DEPTH;
#pragma unroll
            for (int dd = 0; dd < SK_DEPTH - 1; ++dd)
                if (dd < rem) {
                    sk_a_unpermute<MB>(ra[dd]);
                    sk_mma2<MB, NB>(ra[dd], rb_[dd], acc);
                }
        };
        auto run_bf16 = [&]() {
            auto fetch = [&](int g, f32x4 (&a)[MB][2], f32x4 (&b)[NB]) {
                const float* A = sA[0];
                const float* B = sB[0];
                int lda = slda[0], ldb = sldb[0], beg = 0;
#pragma unroll
                for (int s = 0; s < SK_MAXSEG - 1; ++s) {
                    const bool nx = g >= cend[s];
                    A = nx ? sA[s + 1] : A; B = nx ? sB[s + 1] : B;
                    lda = nx ? slda[s + 1] : lda; ldb = nx ? sldb[s + 1] : ldb; beg = nx ? cend[s] : beg;
                }
                sk_fetch_bf16<MB, NB>(A, lda, B, ldb, (g - beg) << 5, mrow, btile, a, b);
            };
            f32x4 ra[SK_DEPTH][MB][2], rb_[SK_DEPTH][NB];
#pragma unroll
            for (int dd = 0; dd < SK_DEPTH; ++dd) fetch(min(first + dd * STR, last), ra[dd], rb_[dd]);
            int g = first;
            const int ngroups = mine / SK_DEPTH;
            for (int gr = 0; gr < ngroups; ++gr) {
#pragma unroll
                for (int dd = 0; dd < SK_DEPTH; ++dd) {
                    sk_mma_bf16<MB, NB>(ra[dd], rb_[dd], acc);
                    fetch(min(g + (SK_DEPTH + dd) * STR, last), ra[dd], rb_[dd]);
                }
                g += SK_DEPTH * STR;
            }
            const int rem = mine - ngroups * SK_DEPTH;
#pragma unroll
            for (int dd = 0; dd < SK_DEPTH - 1; ++dd)
                if (dd < rem) sk_mma_bf16<MB, NB>(ra[dd], rb_[dd], acc);
        };
        SK_STAMP(1);
#if SK_PRIO_YOUNG
        if (wave >= SK_NW / 2) __builtin_amdgcn_s_setprio(1);  // the second-dispatched half loses every arbitration otherwise
#endif
        if (mine > 0) {
            if (job.seg[0].b_kcontig == 3) run_bf16();
            else if (job.seg[0].b_kcontig == 2) run(std::integral_constant<int, 2>{});
            else if (job.seg[0].b_kcontig == 1) run(std::integral_constant<int, 1>{});
            else run(std::integral_constant<int, 0>{});
        }
        if (flagged) {
            // Tail segment (fragment-major weights, quad-contiguous activation loads as in sk_fetch_fast): its chunks are
            // dealt round-robin like the main ring's, so with a main ring of a multiple of SK_NW chunks every wave adds
            // exactly the terms, in exactly the order, of the unflagged kernel.
            // ONE poller per workgroup (a thousand waves polling one word saturate its memory channel and slow the
            // producers down: measured 36 ms instead of 26 ms per forward scan), behind a workgroup barrier
            __syncthreads();
            if (tid == 0) sk_wait_flag(job.wait_flag, job.wait_target);
            __syncthreads();
            const SkSeg& sg = job.seg[job.nseg - 1];
            const int ntail = sg.K >> 4;
            const __amdgpu_buffer_rsrc_t rs = sk_rsrc(sg.A);
            for (int c = wave; c < ntail; c += SK_NW) {
                f32x4 a[MB], b[NB];
#pragma unroll
                for (int rb = 0; rb < MB; ++rb)
                    a[rb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        rs, (unsigned)((mrow[rb] * sg.lda + (c << 4) + 4 * (lane & 3)) * 4), 0, 16 /* sc1 */));
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    b[nb] = *reinterpret_cast<const f32x4*>(sg.B + (size_t)btile[nb] * sg.ldb + ((size_t)c << 8) + (lane << 2));
                sk_a_unpermute<MB>(a);
                sk_mma2<MB, NB>(a, b, acc);
            }
        }
    } else {
        // Generic path (NB == 1): per-element masks for K tails / unaligned operands (e.g. the 63-wide
        // fed-back output frame).  Round-robin 16-deep K chunks over the waves, segment by segment.
        const int ncol = sk_jcol(job, tile0, i);
        const bool ncol_ok = ncol < N;
        f32x4 accg[MB];
#pragma unroll
        for (int rb = 0; rb < MB; ++rb) accg[rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int base = 0;
        for (int s = 0; s < job.nseg; ++s) {
            const SkSeg sg = job.seg[s];
            const int nch = (sg.K + 15) >> 4;
            int c = (wave - (base % SK_NW) + SK_NW) % SK_NW;
            f32x4 a_cur[MB], b_cur;
            if (c < nch) sk_fetch<MB, false>(sg, c * 16, m0, M, ncol, ncol_ok, kk, i, a_cur, b_cur);
            while (c < nch) {
                const int cn = c + SK_NW;
                f32x4 a_nxt[MB], b_nxt;
                if (cn < nch) sk_fetch<MB, false>(sg, cn * 16, m0, M, ncol, ncol_ok, kk, i, a_nxt, b_nxt);
                sk_mma<MB>(a_cur, b_cur, accg);
                if (cn < nch) {
#pragma unroll
                    for (int rb = 0; rb < MB; ++rb) a_cur[rb] = a_nxt[rb];
                    b_cur = b_nxt;
                }
                c = cn;
            }
            base += nch;
        }
#pragma unroll
        for (int rb = 0; rb < MB; ++rb) acc[rb][0] = accg[rb];
    }

    // Intra-workgroup split-K reduction through LDS.
#if SK_PRIO_YOUNG
    __builtin_amdgcn_s_setprio(0);
#endif
    SK_STAMP(2);
#pragma unroll
    for (int rb = 0; rb < MB; ++rb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) red[((wave * MB + rb) * NB + nb) * 64 + lane] = acc[rb][nb];
    __syncthreads();
    SK_STAMP(3);
    if (wave >= EWAVES) return;
    const int blk = eblk;
    const int rb = blk / NB, nbi = blk % NB;
    const int tile = tile0 + nbi;
    float v[RPW];  // rows er0 .. er0 + RPW - 1 of this lane's four, summed over the waves in wave order (as ever)
    {
        const float* rp = reinterpret_cast<const float*>(red) + er0;
#pragma unroll
        for (int r = 0; r < RPW; ++r) v[r] = rp[(blk * 64 + lane) * 4 + r];
#pragma unroll
        for (int w = 1; w < SK_NW; ++w)
#pragma unroll
            for (int r = 0; r < RPW; ++r) v[r] += rp[((w * NBLK + blk) * 64 + lane) * 4 + r];
    }

    // Fused epilogue.  MFMA C/D layout (16x16): column = lane & 15, row = (lane >> 4) * 4 + reg.
    const int g = lane >> 4, jj = lane & 15;
    const int n = sk_jcol(job, tile, jj);
    const bool n_ok = n < N;
    const float bias = job.bias ? p_bias : 0.f;
    const int H = job.H;
    if (!has_add) {
#pragma unroll
        for (int r = 0; r < RPW; ++r) p_add[r] = 0.f;
    }

    if (job.epi == SK_EPI_LSTM) {
        // Gates of hidden unit j live in lanes jj = q*4 + (j&3), q = 0..3 (i, f, o, g order,
        // ops.py:523-541).  Gather them with wave shuffles; lanes with q == 0 do the cell update.
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int m = m0 + rb * 16 + 4 * g + er0 + r;
            const bool ok = (m < M) && n_ok;
            const float pre = v[r] + bias + p_add[r];
            const int q = jj >> 2;
            const float gate = (q == 3) ? tanhf(pre) : ph_sigmoid(pre);
            if (job.o2 && ok) job.o2[sk_off(m, job.ldo2, n)] = gate;  // saved activations [M,4H]
            const int src = (lane & ~12);
            const float gi = __shfl(gate, src | 0, 64);
            const float gf = __shfl(gate, src | 4, 64);
            const float go = __shfl(gate, src | 8, 64);
            const float gg = __shfl(gate, src | 12, 64);
            if (q == 0 && ok) {
                const int j = n;  // q == 0 -> n = hidden index
                const float cp = p_e1[r];
                const float cn = cp * gf + gg * gi;
                job.o1[sk_off(m, job.ldo1, j)] = cn;
                job.out[sk_off(m, job.ldo, j)] = tanhf(cn) * go;
            }
        }
        SK_STAMP(4);
        return;
    }

#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int m = m0 + rb * 16 + 4 * g + er0 + r;
        if (m >= M || !n_ok) continue;
        float x = v[r];
        switch (job.epi) {
            case SK_EPI_LINEAR: {
                x += bias + p_add[r];
                if (job.act == SK_ACT_RELU) x = fmaxf(x, 0.f);
                else if (job.act == SK_ACT_TANH) x = tanhf(x);
                else if (job.act == SK_ACT_SIGMOID) x = ph_sigmoid(x);
                float* o = job.out + sk_off(m, job.ldo, n);
                if (job.accumulate == 1) x += p_oc[r];  // exclusive owner of the tile: no two jobs of a launch share one
                *o = x;
            } break;
            case SK_EPI_GRU_GATES: {
                x += bias + p_add[r];
                const float gt = ph_sigmoid(x);
                if (n < H) {
                    job.o1[sk_off(m, job.ldo1, n)] = gt;  // update gate z
                } else {
                    const int j = n - H;
                    job.o2[sk_off(m, job.ldo2, j)] = gt;  // reset gate r
                    job.out[sk_off(m, job.ldo, j)] = gt * p_e0[r];
                }
            } break;
            case SK_EPI_GRU_CAND: {
                x += bias + p_add[r];
                const float c = tanhf(x);
                const float z = p_e1[r];
                const float hp = p_e0[r];
                float hn = z * c + (1.f - z) * hp;
                if (job.mask) {
                    const float mk = job.mask[m];
                    hn = mk * hn + (1.f - mk) * hp;
                }
                if (job.o1) job.o1[sk_off(m, job.ldo1, n)] = c;
                job.out[sk_off(m, job.ldo, n)] = hn;
            } break;
            case SK_EPI_BWD_RH: {
                // x = d(r*h_prev)[m][n]
                const float r_ = p_e1[r];
                const float hp = p_e0[r];
                job.out[sk_off(m, job.ldo, n)] = x * hp * r_ * (1.f - r_);  // dG_r
                job.o1[sk_off(m, job.ldo1, n)] = p_oc[r] + x * r_;          // dh_prev +=
            } break;
            default: break;
        }
    }
    SK_STAMP(4);
#ifdef SK_TIMERS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // [5] the epilogue's stores acknowledged
    SK_STAMP(5);
#endif
}

template <int MB, int NB>
__global__ __launch_bounds__(SK_THREADS) void sk_kernel(const SkLaunch L) {
    extern __shared__ __attribute__((aligned(16))) char sk_smem[];
    f32x4* red = reinterpret_cast<f32x4*>(sk_smem);
    // Launches whose jobs all have the same number of workgroups use grid.z = job: the descriptor address then
    // follows from the block id alone and the wave's first scalar loads fetch the job itself.  Otherwise the
    // workgroups of the jobs are laid out back to back along x and the job is found in the prefix table
    // (a z-grid sized for the largest job was measured slower there: 59 -> 65 ms backward at cfg2).
    int j = blockIdx.z, bx = blockIdx.x;
#if SK_EARLY_DESC  // the launch header in one batch of scalar loads (zmode, njobs and the prefix table sit in one line)
    asm volatile("" ::"s"(L.zmode), "s"(L.njobs), "s"(L.tile_end[0]), "s"(L.tile_end[1]), "s"(L.tile_end[2]), "s"(L.tile_end[3]),
                 "s"(L.tile_end[4]), "s"(L.tile_end[5]), "s"(L.tile_end[6]), "s"(L.tile_end[7]));
#endif
    if (!L.zmode) {
        j = 0;
#pragma unroll
        for (int q = 0; q < SK_MAXJOB - 1; ++q)
            if (q < L.njobs - 1 && bx >= L.tile_end[q]) j = q + 1;
        bx -= (j > 0 ? L.tile_end[j - 1] : 0);
    }
    const SkJob& job = L.job[j];
    const int tile0 = bx * NB;  // first 16-column tile of this workgroup
    if (NB > 1 || job.aligned) sk_body<MB, NB, true>(job, tile0, red);  // fast path: branch-free operand fetch
    else sk_body<MB, 1, false>(job, tile0, red);
}

// Heterogeneous step launch: the first natt_x workgroups of every grid row carry the attention forward step
// (att_fwd_body.h; one (batch row, column slice) pair each, on all eight waves), the others are
// step-GEMM workgroups as in sk_kernel (jobs found through the workgroup prefix table).  The attention of a tick is a
// chain of dependent round trips that leaves the chip idle; here independent GEMM jobs (the upper layers' input
// projections, plans.hip schedule 5) run in its shadow.
#ifndef SKA_PROJ_UNROLL
#define SKA_PROJ_UNROLL 4  // 8 (the stand-alone kernel's) would cost the GEMM side its second workgroup per CU
#endif
template <int MB, int NB>
__global__ __launch_bounds__(SK_THREADS, 4) void ska_kernel(const SkLaunch L, const AttFwdArgs g, const int natt_x,
                                                         const int att_last) {
    extern __shared__ __attribute__((aligned(16))) char sk_smem[];
    int bx = blockIdx.x;
    if (att_last == 1) {  // GEMM workgroups (the long ones) are dispatched first, the attention fills in behind them
        const int ngemm = (int)gridDim.x - natt_x;
        bx = bx >= ngemm ? bx - ngemm : bx + natt_x;
    }
    if (bx < natt_x) {
        // att_last == 2 (a job of this launch waits for the attention): all attention workgroups lead grid row 0
        if (att_last == 2 && blockIdx.y != 0) return;
        const int id = att_last == 2 ? bx : blockIdx.y * natt_x + bx;
        if (id >= g.B * g.esplit) return;
        att_fwd_block<SK_THREADS, SKA_PROJ_UNROLL>(g, id / g.esplit, id % g.esplit, reinterpret_cast<float*>(sk_smem));
        return;
    }
    bx -= natt_x;
    f32x4* red = reinterpret_cast<f32x4*>(sk_smem);
    int j = 0;
#pragma unroll
    for (int q = 0; q < SK_MAXJOB - 1; ++q)
        if (q < L.njobs - 1 && bx >= L.tile_end[q]) j = q + 1;
    bx -= (j > 0 ? L.tile_end[j - 1] : 0);
    const SkJob& job = L.job[j];
    const int tile0 = bx * NB;
    if (NB > 1 || job.aligned) sk_body<MB, NB, true>(job, tile0, red);
    else sk_body<MB, 1, false>(job, tile0, red);
}

// Heterogeneous BACKWARD launch (plans.hip bwd8, GRU layers): the attention backward + state backward row blocks of
// att_state_bwd_kernel (1024 threads) lead the grid, the other workgroups are step-GEMM workgroups as in sk_kernel on
// eight of the block's sixteen waves (the others exit at once).  The row blocks are a chain of dependent round trips on
// 64-128 CUs; the GEMM jobs riding here are products nothing in this launch depends on (the downward products of the
// upper layers' previous tick), so the idle CUs of that chain do work that used to lengthen the next two launches.
template <int MB, int NB>
__global__ __launch_bounds__(ATTB_THREADS) void skb_kernel(const SkLaunch L, const AttBwdArgs g, const GruStateBwdArgs sa,
                                                           const int att_rows, const int l0_chain, const int nlead,
                                                           const int nlead_x, const int rpb) {
    extern __shared__ __attribute__((aligned(16))) char sk_smem[];
    int bx = blockIdx.x;
    if (bx < nlead_x) {
        const int id = blockIdx.y * nlead_x + bx;
        if (id >= nlead) return;
        att_state_bwd_block(g, sa, att_rows, l0_chain, id, reinterpret_cast<float*>(sk_smem), rpb);
        return;
    }
    if (threadIdx.x >= SK_THREADS) return;
    bx -= nlead_x;
    f32x4* red = reinterpret_cast<f32x4*>(sk_smem);
    int j = 0;
#pragma unroll
    for (int q = 0; q < SK_MAXJOB - 1; ++q)
        if (q < L.njobs - 1 && bx >= L.tile_end[q]) j = q + 1;
    bx -= (j > 0 ? L.tile_end[j - 1] : 0);
    const SkJob& job = L.job[j];
    const int tile0 = bx * NB;
    if (NB > 1 || job.aligned) sk_body<MB, NB, true>(job, tile0, red);
    else sk_body<MB, 1, false>(job, tile0, red);
}

namespace {
__global__ __launch_bounds__(256) void sk_zero_words_kernel(unsigned* p, int n) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = 0u;
}
}  // namespace
// Zero-fill of a few flag words as a kernel node of its own (graph memset nodes replayed on the default stream were
// seen to race with work still in flight, see pm_launch in persist.hip).
int sk_zero_words_launch(unsigned* p, int n, hipStream_t stream) {
    if (!p || n < 1) return PH_ERR_BADARG;
    hipLaunchKernelGGL(sk_zero_words_kernel, dim3(n > 4096 ? 16 : 1), dim3(256), 0, stream, p, n);
    return (int)hipGetLastError();
}

void sk_job_init(SkJob& j) { memset(&j, 0, sizeof(j)); }

void sk_finalize_job(SkJob& j) {
    int al = 1;
    for (int s = 0; s < j.nseg; ++s) {
        const SkSeg& g = j.seg[s];
        if (((uintptr_t)g.A & 15) || (g.lda & 3) || (g.K & 15)) al = 0;
        if (g.b_kcontig && (((uintptr_t)g.B & 15) || (g.ldb & 3))) al = 0;
        if (g.b_kcontig >= 2 && (j.N & 15)) al = 0;
        if (g.b_kcontig == 3 && (g.K & 31)) al = 0;
        if (g.b_kcontig != j.seg[0].b_kcontig) al = 0;  // the fast path assumes one weight layout per job
    }
    j.aligned = al;
    if (j.ksplit > 1 && (j.seg[0].b_kcontig != 3 || j.nseg != 1)) j.aligned = -1;  // K parts: wide bf16 kernel only
    // a waiting job: fragment-major weights, f32 (sk_body's tail) or bf16 (wk_body's tail: sk_launch_att refuses the
    // launch if the wide kernel does not take it); anything else is rejected by sk_make_launch
    if (j.wait_flag && (!al || (j.wait_all ? (j.nseg != 1 || j.seg[0].b_kcontig != 3) : j.nseg < 2) || j.seg[0].b_kcontig < 2 || !SK_A_PERMUTE))
        j.aligned = -1;
}

int sk_make_launch(SkLaunch& L, const SkJob* jobs, int njobs) {
    if (njobs < 1 || njobs > SK_MAXJOB) return PH_ERR_BADARG;
    for (int q = 0; q < njobs; ++q) {  // sk_off: 24-bit rows and leading dimensions
        const SkJob& j = jobs[q];
        const int lim = 1 << 24;
        if (j.M >= lim || j.ld_add >= lim || j.ldo >= lim || j.lde0 >= lim || j.lde1 >= lim || j.ldo1 >= lim || j.ldo2 >= lim ||
            j.ld_add < 0 || j.ldo < 0 || j.lde0 < 0 || j.lde1 < 0 || j.ldo1 < 0 || j.ldo2 < 0)
            return PH_ERR_BADARG;
    }
    memset(&L, 0, sizeof(L));
    int t = 0;
    for (int q = 0; q < njobs; ++q) {
        L.job[q] = jobs[q];
        sk_finalize_job(L.job[q]);
        const SkJob& j = L.job[q];
        if (j.nseg < 1 || j.nseg > SK_MAXSEG || j.M < 1 || j.N < 1 || j.aligned < 0) return PH_ERR_BADARG;
        if (j.seg[0].b_kcontig >= 2 && !j.aligned) return PH_ERR_BADARG;  // tiled weights: fast path only
        int tiles;
        if (j.epi == SK_EPI_LSTM || j.colmode == 1) {
            if (j.N != 4 * j.H || (j.H & 3)) return PH_ERR_BADARG;
            if (j.colmode == 1 && j.epi != SK_EPI_LINEAR) return PH_ERR_BADARG;
            tiles = j.H / 4;
        } else {
            tiles = ceil_div(j.N, 16);
        }
        (void)t;
        L.tile_end[q] = tiles;  // 16-column tiles of the job; sk_launch turns this into a prefix of workgroups
    }
    L.njobs = njobs;
    return 0;
}

// ---- fragment-major weight copies ------------------------------------------------------------
// The step kernels are bound by how fast a CU's texture path turns wave loads into cache-line requests:
// a wave load whose 64 lanes hit 16+ different lines (activation rows 4 KB apart, or 4-byte-per-lane
// weight loads) moves 16 B per clock, one contiguous 1 KB block moves 64 B per clock (measured: -20..27 %
// per launch with both operands contiguous).  So the scan reads its weights from copies laid out in the
// exact order the MFMA operand registers want them: block (column tile ct, 16-deep chunk c) holds
// [kk][i][u] = W[16c + 4kk + u][col(ct, i)]  (mode 0, forward: out = A . W), or
// [kk][i][u] = W[16ct + i][16c + 4kk + u]    (mode 1, backward: out = A . W^T),
// 256 floats each, blocks ordered [ct][c].  lstm_H > 0 applies the gate-interleaved column order of
// SK_EPI_LSTM (sk_col) in mode 0.  The copies are refreshed once per training step (44 MB, ~30 us).
namespace {
__global__ __launch_bounds__(256) void sk_tile_weights_kernel(const float* __restrict__ W, int rows, int cols, int ld,
                                                              float* __restrict__ out, int mode, int lstm_H) {
    const int nct = mode == 0 ? cols >> 4 : rows >> 4;
    const int nch = mode == 0 ? rows >> 4 : cols >> 4;
    const size_t total = (size_t)nct * nch * 64;  // one f32x4 per thread-item
    for (size_t it = (size_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (size_t)gridDim.x * 256) {
        const int lane = (int)(it & 63);
        const size_t blk = it >> 6;
        const int c = (int)(blk % nch), ct = (int)(blk / nch);
        const int i = lane & 15, kk = lane >> 4;
        f32x4 v;
        if (mode == 0) {
            const int col = lstm_H > 0 ? (i >> 2) * lstm_H + ct * 4 + (i & 3) : ct * 16 + i;
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = W[(size_t)(c * 16 + kk * 4 + u) * ld + col];
        } else {
            v = *reinterpret_cast<const f32x4*>(W + (size_t)(ct * 16 + i) * ld + c * 16 + kk * 4);
        }
        *reinterpret_cast<f32x4*>(out + it * 4) = v;
    }
}
// bf16 variant: block (column tile ct, 32-deep chunk c) holds, per lane (kk = lane >> 4, i = lane & 15), the 8
// values k = 32c + 8kk .. +7 of column col(ct, i) (mode 0) / of W[16ct + i][k] (mode 1), rounded to nearest even;
// 512 bf16 = 1 KB per block, blocks ordered [ct][c].
__global__ __launch_bounds__(256) void sk_tile_weights_bf16_kernel(const float* __restrict__ W, int rows, int cols, int ld,
                                                                   bf16x8* __restrict__ out, int mode, int lstm_H) {
    const int nct = mode == 0 ? cols >> 4 : rows >> 4;
    const int nch = mode == 0 ? rows >> 5 : cols >> 5;
    const size_t total = (size_t)nct * nch * 64;
    for (size_t it = (size_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (size_t)gridDim.x * 256) {
        const int lane = (int)(it & 63);
        const size_t blk = it >> 6;
        const int c = (int)(blk % nch), ct = (int)(blk / nch);
        const int i = lane & 15, kk = lane >> 4;
        f32x4 lo, hi;
        if (mode == 0) {
            const int col = lstm_H > 0 ? (i >> 2) * lstm_H + ct * 4 + (i & 3) : ct * 16 + i;
            const float* p = W + (size_t)(c * 32 + kk * 8) * ld + col;
#pragma unroll
            for (int u = 0; u < 4; ++u) { lo[u] = p[(size_t)u * ld]; hi[u] = p[(size_t)(u + 4) * ld]; }
        } else {
            const float* p = W + (size_t)(ct * 16 + i) * ld + c * 32 + kk * 8;
            lo = *reinterpret_cast<const f32x4*>(p);
            hi = *reinterpret_cast<const f32x4*>(p + 4);
        }
        out[it] = ph_bf16x8(lo, hi);
    }
}
}  // namespace

int sk_tile_weights_bf16_launch(const float* W, int rows, int cols, int ld, void* out, int mode, int lstm_H,
                                hipStream_t stream) {
    const int K = mode == 0 ? rows : cols, N = mode == 0 ? cols : rows;
    if (!W || !out || K < 32 || N < 16 || (K & 31) || (N & 15) || (ld & 3) || ((uintptr_t)W & 15) ||
        ((uintptr_t)out & 15) || (lstm_H > 0 && (mode != 0 || cols != 4 * lstm_H)))
        return PH_ERR_BADARG;
    const size_t items = (size_t)rows * cols / 8;
    int blocks = (int)((items + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sk_tile_weights_bf16_kernel, dim3(blocks), dim3(256), 0, stream, W, rows, cols, ld,
                       reinterpret_cast<bf16x8*>(out), mode, lstm_H);
    return (int)hipGetLastError();
}

int sk_tile_weights_launch(const float* W, int rows, int cols, int ld, float* out, int mode, int lstm_H,
                           hipStream_t stream) {
    if (!W || !out || rows < 16 || cols < 16 || (rows & 15) || (cols & 15) || (ld & 3) || ((uintptr_t)W & 15) ||
        ((uintptr_t)out & 15) || (lstm_H > 0 && (mode != 0 || cols != 4 * lstm_H)))
        return PH_ERR_BADARG;
    const size_t items = (size_t)rows * cols / 4;
    int blocks = (int)((items + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sk_tile_weights_kernel, dim3(blocks), dim3(256), 0, stream, W, rows, cols, ld, out, mode, lstm_H);
    return (int)hipGetLastError();
}

// ---- optional per-dispatch timing (bench.py roofline leg) ------------------------------------
// When enabled, every launch is bracketed by HIP events attached to the dispatch itself
// (hipExtLaunchKernelGGL start/stop events = the kernel's own begin/end timestamps), and the
// algorithmic flops / bytes of its jobs are recorded next to them.
namespace {
struct SkProfRec { hipEvent_t e0, e1; double flops, bytes; int hetero = 0; };  // hetero: the launch also carries attention / state row blocks
struct SkProf { bool on = false; std::vector<SkProfRec> recs; } g_prof;

void sk_account(const SkLaunch& L, double& flops, double& bytes) {
    flops = 0.0; bytes = 0.0;
    for (int q = 0; q < L.njobs; ++q) {
        const SkJob& j = L.job[q];
        double ksum = 0.0;
        for (int s = 0; s < j.nseg; ++s) ksum += j.seg[s].K;
        const double wb = j.seg[0].b_kcontig == 3 ? 2.0 : 4.0;  // bytes per weight element
        flops += 2.0 * j.M * ksum * j.N;
        double epi = 1.0;  // values moved per output element by the epilogue
        switch (j.epi) {
            case SK_EPI_GRU_GATES: epi = 2.0 + (j.add ? 1.0 : 0.0); break;          // z|r out, rh out + h_prev in (half width each)
            case SK_EPI_GRU_CAND: epi = 4.0 + (j.add ? 1.0 : 0.0); break;           // c, h' out; z, h_prev in
            case SK_EPI_BWD_RH: epi = 5.0; break;                                   // r, h_prev in; dG_r out; dh rmw
            case SK_EPI_LSTM: epi = 2.0; break;
            default: epi = 1.0 + (j.accumulate ? 1.0 : 0.0) + (j.add ? 1.0 : 0.0); break;
        }
        bytes += wb * ksum * j.N + 4.0 * ((double)j.M * ksum + epi * j.M * j.N);
    }
}
}  // namespace

#ifdef SK_TIMERS
// Development builds only (python -m parrot_amd.build --timers): where the step kernels / attention row blocks stamp.
extern "C" int parrot_debug_set_timers(void* sk_buf, void* att_buf) {
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(sk_timer_buf), &sk_buf, sizeof(sk_buf));
    if (e != hipSuccess) return (int)e;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(att_timer_buf), &att_buf, sizeof(att_buf));
}
#endif

void sk_profile_begin() {
    for (auto& r : g_prof.recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_prof.recs.clear();
    g_prof.on = true;
}

// Must be called after the stream has been synchronised.  Returns the number of launches.
long long sk_profile_end(double* total_us, double* flops, double* bytes) { return sk_profile_end2(total_us, flops, bytes, nullptr); }

// plain[0..3]: dispatch time (us), flops, bytes and launch count of the PLAIN step-GEMM launches (sk_kernel / wk_kernel:
// no attention or state row blocks in the grid) -- the dominant kernel on its own; the totals cover the whole family.
long long sk_profile_end2(double* total_us, double* flops, double* bytes, double* plain) {
    g_prof.on = false;
    double us = 0.0, fl = 0.0, by = 0.0, pus = 0.0, pfl = 0.0, pby = 0.0, pn = 0.0;
    for (auto& r : g_prof.recs) {
        float ms = 0.f;
        const bool ok = hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess;
        if (ok) us += 1000.0 * ms;
        fl += r.flops; by += r.bytes;
        if (!r.hetero) {
            if (ok) pus += 1000.0 * ms;
            pfl += r.flops; pby += r.bytes; pn += 1.0;
        }
        (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
    }
    const long long n = (long long)g_prof.recs.size();
    g_prof.recs.clear();
    if (total_us) *total_us = us;
    if (flops) *flops = fl;
    if (bytes) *bytes = by;
    if (plain) { plain[0] = pus; plain[1] = pfl; plain[2] = pby; plain[3] = pn; }
    return n;
}

// ---- wide bf16 step kernel ("wk") ----------------------------------------------------------------------------
// For the big bf16 launches (3 x LSTM-1536: 132 MB of weights per tick) sk_kernel's 32 x 32 tiles are bound by the
// CU's L1 bandwidth: every tile re-reads its [32, K] f32 activation slab, 793 MB of L2 -> CU traffic per tick at 43 %
// of the aggregate L1 peak.  Here a workgroup owns ALL (<= 64) rows of NC = 64 or 128 columns, the eight waves split N
// (and, for NC = 64, the two row halves) instead of K, and the activation stage ([64, 64] f32 -> bf16) goes through
// LDS once per workgroup: 342 MB per tick, no split-K reduction at all (every wave ends with complete sums and runs
// its part of the fused epilogue).  Weights: one 1 KB fragment-major block per wave and 32-deep chunk, straight
// into registers.  One __syncthreads per 64-deep stage, two LDS stage buffers, register rings for both operands.
// Ring depths 2 / 2 (round 4): 121 VGPRs instead of 157, i.e. four waves per SIMD -- TWO workgroups per CU -- which is what
// lets the attention blocks of a heterogeneous launch (wka_kernel, plans.hip schedule 7) run BESIDE the wide workgroups
// instead of in front of them: cfg4 step 104.5 -> 97.9 ms with the attention in the tick's launch, and 106.9 -> 104.6 ms
// on schedule 0 (measured on one box, profiles/r04_cfg4_schedule7.txt); deeper rings never bought anything (below).
#ifndef WK_PA_DEPTH
#define WK_PA_DEPTH 2
#endif
#ifndef WK_PB_DEPTH
#define WK_PB_DEPTH 2
#endif
// K per stage; LDS row pitch in bytes; ring depths in stages.  Measured the same at cfg4 (34.4-36.1 us per launch):
// ring depths 2/3, 4/4 and 8/8, a stage of 128 instead of 64 K rows.  Measured worse: starting every workgroup at a
// different stage of the K range (43 us: the lock-step walk is what makes the shared activation rows hit in L2).
#ifndef WK_STAGE_K
#define WK_STAGE_K 64
#endif
enum { WK_STAGE = WK_STAGE_K, WK_PITCH = 2 * WK_STAGE_K + 16, WK_PA = WK_PA_DEPTH, WK_PB = WK_PB_DEPTH,
       WK_TPR = WK_STAGE_K / 4,          // threads per activation row of a stage (4 k each)
       WK_RPP = SK_THREADS / WK_TPR,     // rows per staging pass
       WK_NP = 64 / WK_RPP,              // staging passes = 16-byte loads per thread and stage
       WK_KS = WK_STAGE_K / 32 };        // MFMA k-steps = weight blocks per wave and stage

struct WkLaunch {
    SkJob job[SK_MAXJOB];
    int njobs;
    int wg_end[SK_MAXJOB];  // prefix of workgroups per job
    int ncw[SK_MAXJOB];     // 16-column tiles per workgroup: 4 or 8
    int wgh[SK_MAXJOB];     // workgroups per K part of the job (SkJob::ksplit parts follow each other in the grid)
};

// WAITALL: the job's whole A operand is produced inside the launch (the fused backward tick, wkb_kernel)
template <int NCW, bool WAITALL = false>
__device__ __forceinline__ void wk_body(const SkJob& job, int wg_in, int wgh, char* smem) {
    // K part of a split job (SkJob::ksplit): workgroups [p * wgh, (p + 1) * wgh) take the p-th part of the K range
    const int nparts = job.ksplit > 1 ? job.ksplit : 1;
    const int kpart = __builtin_amdgcn_readfirstlane(nparts > 1 ? wg_in / wgh : 0);
    const int wg = nparts > 1 ? wg_in - kpart * wgh : wg_in;
    constexpr int MB = NCW == 8 ? 4 : 2;  // row blocks per wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ctl = NCW == 8 ? wave : (wave & 3), rh = NCW == 8 ? 0 : (wave >> 2);
    const int M = job.M, N = job.N;
    const int ntiles = (job.epi == SK_EPI_LSTM || job.colmode == 1) ? (job.H >> 2) : ((N + 15) >> 4);
    const int tile = min(wg * NCW + ctl, ntiles - 1);
    const bool tile_ok = wg * NCW + ctl < ntiles;

    // K stages over the concatenated segments.  The two operand streams run ahead of the MFMAs by different distances,
    // so each has its own (wave-uniform) cursor into the segment list; past the end a cursor stays on the last stage.
    // A job whose LAST segment's activations are produced by the attention blocks of the same launch (SkJob::wait_flag,
    // plans.hip schedule 7) walks its other segments through the pipelined ring and takes that segment afterwards
    // (wk_tail): same terms in the same order as the unflagged job, bit for bit.
    const bool flagged = !WAITALL && job.wait_flag != nullptr;
    const int nseg_main = flagged ? job.nseg - 1 : job.nseg;
    int total = 0;
    for (int q = 0; q < nseg_main; ++q) total += job.seg[q].K / WK_STAGE;
    total /= nparts;  // (split jobs have one segment whose K is a multiple of nparts stages: wk_build checks)
    struct Cursor { const float* A; const float* B; int lda, ldb, left, seg, k; };
    auto cursor_init = [&](Cursor& c) __attribute__((always_inline)) {
        c.seg = 0; c.k = kpart * total * WK_STAGE;
        c.A = job.seg[0].A; c.B = job.seg[0].B; c.lda = job.seg[0].lda; c.ldb = job.seg[0].ldb;
        c.left = nparts > 1 ? total : job.seg[0].K / WK_STAGE;
    };
    auto cursor_next = [&](Cursor& c) __attribute__((always_inline)) {
        if (c.left > 1) { --c.left; c.k += WK_STAGE; return; }
        if (c.seg + 1 < nseg_main) {
            ++c.seg;
            const SkSeg& sg = job.seg[c.seg];
            c.A = sg.A; c.B = sg.B; c.lda = sg.lda; c.ldb = sg.ldb; c.left = sg.K / WK_STAGE; c.k = 0;
        }
    };

    // staging role: rows r0 + p * WK_RPP (p < WK_NP), k = 4 * akq .. +3 of the stage: WK_TPR lanes read one row's
    // contiguous 4 * WK_STAGE bytes
    const int ar0 = tid / WK_TPR, akq = tid % WK_TPR;
    Cursor ca, cb;
    cursor_init(ca);
    cursor_init(cb);
    // WAITALL: one poller per workgroup waits for the rows' arrival count before the first load.  The rows were published
    // write-through (sc1 stores, drained before the arrival) and are written exactly ONCE per launch, before any read of
    // them; a kernel starts with its L2s invalidated, so no cache can hold an older copy of these lines and the ring
    // reads them with ordinary (L2-cached) loads -- sc1 loads, which miss every L2, made every workgroup fetch the whole
    // [64, 4H] operand from the fabric: 63 us per tick instead of 56 for the two launches (profiles/r04_cfg4_schedule7.txt).
    // tests/test_gpu_bf16.py::test_in_launch_handoffs_never_see_stale_rows replays a plan on changing data to pin that.
    if (WAITALL) {
        if (tid == 0) {
            sk_wait_flag(job.wait_flag, job.wait_target);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // drops this CU's L1 (buffer_inv sc1): one lane, once
        }
        __syncthreads();
    }
    auto loadA = [&](f32x4 (&a)[WK_NP]) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < WK_NP; ++p)
            a[p] = *reinterpret_cast<const f32x4*>(ca.A + (size_t)min(ar0 + p * WK_RPP, M - 1) * ca.lda + ca.k + 4 * akq);
        cursor_next(ca);
    };
    auto loadB = [&](f32x4 (&b)[WK_KS]) __attribute__((always_inline)) {
        const float* p = cb.B + (size_t)tile * cb.ldb + ((size_t)(cb.k >> 5) << 8) + (lane << 2);
#pragma unroll
        for (int q = 0; q < WK_KS; ++q) b[q] = *reinterpret_cast<const f32x4*>(p + 256 * q);
        cursor_next(cb);
    };

    f32x4 acc[MB];
#pragma unroll
    for (int rb = 0; rb < MB; ++rb) acc[rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 ra[WK_PA][WK_NP], rbv[WK_PB][WK_KS];
#pragma unroll
    for (int q = 0; q < WK_PA; ++q) loadA(ra[q]);
#pragma unroll
    for (int q = 0; q < WK_PB; ++q) loadB(rbv[q]);

    const int kk = lane >> 4, i16 = lane & 15;
    // Stage st: the MFMA operands of stage st are read from the buffer the previous iteration filled, the NEXT stage is
    // converted and written into the other buffer meanwhile, then the MFMAs run and one barrier closes the stage (it
    // publishes stage st + 1 and retires the reads of stage st before that buffer is refilled).  With write -> barrier ->
    // read -> MFMA in a row the LDS round trip and the barrier sat on every stage's critical path (measured: 875
    // clocks per stage, independent of the prefetch depth; a variant that halved the waves working per stage doubled it).
    auto fill = [&](int st, f32x4 (&a)[WK_NP]) __attribute__((always_inline)) {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        char* buf = smem + (st & 1) * (64 * WK_PITCH);
#pragma unroll
        for (int p = 0; p < WK_NP; ++p)
            *reinterpret_cast<bf16x4*>(buf + (ar0 + p * WK_RPP) * WK_PITCH + 8 * akq) = __builtin_convertvector(a[p], bf16x4);
        loadA(a);
    };
    auto stage = [&](int st, f32x4 (&anext)[WK_NP], f32x4 (&b)[WK_KS]) __attribute__((always_inline)) {
        const char* buf = smem + (st & 1) * (64 * WK_PITCH);
        if (st + 1 < total) fill(st + 1, anext);
#pragma unroll
        for (int ks = 0; ks < WK_KS; ++ks) {
            const bf16x8 bv = __builtin_bit_cast(bf16x8, b[ks]);
#pragma unroll
            for (int rb = 0; rb < MB; ++rb) {
                const bf16x8 av = *reinterpret_cast<const bf16x8*>(buf + (16 * (rh * MB + rb) + i16) * WK_PITCH + ks * 64 + 16 * kk);
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[rb], 0, 0, 0);
            }
        }
        loadB(b);
        __syncthreads();
    };

    // the rings are indexed with compile-time constants: unroll by WK_PB (a multiple of WK_PA); stage st + 1 is filled
    // from ring slot (st + 1) % WK_PA
    static_assert(WK_PB % WK_PA == 0, "");
    fill(0, ra[0]);
    __syncthreads();
    int st = 0;
    for (; st + WK_PB <= total; st += WK_PB) {
#pragma unroll
        for (int q = 0; q < WK_PB; ++q) stage(st + q, ra[(q + 1) % WK_PA], rbv[q]);
    }
#pragma unroll
    for (int q = 0; q < WK_PB - 1; ++q)
        if (st + q < total) stage(st + q, ra[(q + 1) % WK_PA], rbv[q]);
    if (flagged) {
        // Tail segment: one poller per workgroup waits for the producers' arrival count (they publish their rows
        // write-through before they arrive), then the stages of the segment run unpipelined through LDS buffer 0: the
        // activations come through sc1 buffer loads (another XCD's L2 may hold a stale line of a buffer that is
        // rewritten every window), the weights as in the ring.  The workgroups of such a job have the shortest K of
        // their launch and sit last in the grid: the few microseconds of this loop are slack.
        if (tid == 0) sk_wait_flag(job.wait_flag, job.wait_target);
        __syncthreads();
        const SkSeg& sg = job.seg[job.nseg - 1];
        const __amdgpu_buffer_rsrc_t rs = sk_rsrc(sg.A);
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        for (int s2 = 0; s2 < sg.K / WK_STAGE; ++s2) {
            f32x4 b[WK_KS];
            const float* pb = sg.B + (size_t)tile * sg.ldb + ((size_t)((s2 * WK_STAGE) >> 5) << 8) + (lane << 2);
#pragma unroll
            for (int q = 0; q < WK_KS; ++q) b[q] = *reinterpret_cast<const f32x4*>(pb + 256 * q);
#pragma unroll
            for (int p = 0; p < WK_NP; ++p) {
                const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    rs, (unsigned)((min(ar0 + p * WK_RPP, M - 1) * sg.lda + s2 * WK_STAGE + 4 * akq) * 4), 0, 16 /* sc1 */));
                *reinterpret_cast<bf16x4*>(smem + (ar0 + p * WK_RPP) * WK_PITCH + 8 * akq) = __builtin_convertvector(a, bf16x4);
            }
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < WK_KS; ++ks) {
                const bf16x8 bv = __builtin_bit_cast(bf16x8, b[ks]);
#pragma unroll
                for (int rb = 0; rb < MB; ++rb) {
                    const bf16x8 av = *reinterpret_cast<const bf16x8*>(smem + (16 * (rh * MB + rb) + i16) * WK_PITCH + ks * 64 + 16 * kk);
                    acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[rb], 0, 0, 0);
                }
            }
            __syncthreads();
        }
    }
    if (!tile_ok) return;

    // fused epilogue, per wave (complete sums).  C layout: column = lane & 15, row = 4 * (lane >> 4) + reg.

    // Output of this workgroup's K part, picked once with scalar selects over the four (unconditionally loaded) kernel
    // arguments: a select of POINTERS inside the store loop made the compiler index the argument block dynamically and
    // copy all 3 KB of it to scratch (3000 bytes per lane, 190 / 390 us per launch instead of 41 / 46).
    unsigned long long kb_ = (unsigned long long)job.out;
    {
        const unsigned long long b1 = (unsigned long long)job.o1, b2 = (unsigned long long)job.kout2, b3 = (unsigned long long)job.kout3;
        if (kpart == 1) kb_ = b1;
        if (kpart == 2) kb_ = b2;
        if (kpart == 3) kb_ = b3;
    }
    float* const kbase = reinterpret_cast<float*>(kb_);
    const int kld = kpart ? job.ldo1 : job.ldo;
    const int g = lane >> 4, jj = lane & 15;
    const int n = sk_jcol(job, tile, jj);
    const bool n_ok = n < N;
    // Round 6: every epilogue operand of the wave (additive input, previous cell, the sums a split part adds to) is requested
    // up front, unconditionally, before the first is used: read where needed -- `if (job.add && ok) pre += job.add[..]`, then
    // `cp = job.e1[..]` behind the gate shuffles, `x += *o` -- each conditional load ended in s_waitcnt vmcnt(0): up to two
    // dependent round trips per accumulator row, sixteen rows per lane.
    const bool lstm = job.epi == SK_EPI_LSTM;
    const bool has_add = job.add != nullptr && (lstm || !kpart);
    const bool acc_on = !lstm && (kpart == 0 ? job.accumulate != 0 : (kpart == 1 && job.ldo2 != 0));
    const int nc = min(n, N - 1);
    const float* q_add = has_add ? job.add : kbase;
    const int l_add = has_add ? job.ld_add : kld;
    const float* q_e1 = lstm ? job.e1 : kbase;
    const int l_e1 = lstm ? job.lde1 : kld;
    const int c_e1 = lstm ? ((jj >> 2) == 0 ? nc : 0) : nc;
    const float* q_b = job.bias ? job.bias : kbase;
    const float bias_raw = q_b[job.bias ? nc : 0];
    float v_add[MB][4], v_e1[MB][4], v_old[MB][4];
#pragma unroll
    for (int rb = 0; rb < MB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int mc = min(16 * (rh * MB + rb) + 4 * g + r, M - 1);
            v_add[rb][r] = q_add[sk_off(mc, l_add, nc)];
            v_e1[rb][r] = q_e1[sk_off(mc, l_e1, c_e1)];
            v_old[rb][r] = kbase[sk_off(mc, kld, nc)];
        }
    const float bias = (job.bias && n_ok) ? bias_raw : 0.f;
#pragma unroll
    for (int rb = 0; rb < MB; ++rb) {
        const int mb0 = 16 * (rh * MB + rb) + 4 * g;
        if (lstm) {
            const int q = jj >> 2;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mb0 + r;
                const bool ok = m < M && n_ok;
                float pre = acc[rb][r] + bias;
                if (has_add && ok) pre += v_add[rb][r];
                const float gate = (q == 3) ? tanhf(pre) : ph_sigmoid(pre);
                if (job.o2 && ok) job.o2[sk_off(m, job.ldo2, n)] = gate;
                const int src = lane & ~12;
                const float gi = __shfl(gate, src | 0, 64);
                const float gf = __shfl(gate, src | 4, 64);
                const float go = __shfl(gate, src | 8, 64);
                const float gg = __shfl(gate, src | 12, 64);
                if (q == 0 && ok) {
                    const float cp = v_e1[rb][r];
                    const float cn = cp * gf + gg * gi;
                    job.o1[sk_off(m, job.ldo1, n)] = cn;
                    job.out[sk_off(m, job.ldo, n)] = tanhf(cn) * go;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mb0 + r;
                if (m >= M || !n_ok) continue;
                float x = acc[rb][r] + (kpart ? 0.f : bias);
                if (has_add) x += v_add[rb][r];
                if (job.act == SK_ACT_RELU) x = fmaxf(x, 0.f);
                else if (job.act == SK_ACT_TANH) x = tanhf(x);
                else if (job.act == SK_ACT_SIGMOID) x = ph_sigmoid(x);
                float* o = kbase + sk_off(m, kld, n);
                // part 0 follows `accumulate`, part 1 the flag in ldo2 (split LINEAR jobs), parts 2 and 3 are stored
                if (acc_on) x += v_old[rb][r];
                *o = x;
            }
        }
    }
}

__global__ __launch_bounds__(SK_THREADS) void wk_kernel(const WkLaunch L) {
    extern __shared__ __attribute__((aligned(16))) char wk_smem[];
    int j = 0, bx = blockIdx.x;
#pragma unroll
    for (int q = 0; q < SK_MAXJOB - 1; ++q)
        if (q < L.njobs - 1 && bx >= L.wg_end[q]) j = q + 1;
    bx -= (j > 0 ? L.wg_end[j - 1] : 0);
    const SkJob& job = L.job[j];
    if (L.ncw[j] == 8) wk_body<8>(job, bx, L.wgh[j], wk_smem);
    else wk_body<4>(job, bx, L.wgh[j], wk_smem);
}

// Heterogeneous variant of wk_kernel: workgroups [0, natt) carry the attention forward step (one batch row each, all
// eight waves), the others are wide-kernel workgroups (the bf16 input projections of LSTM layers beside the attention).
__global__ __launch_bounds__(SK_THREADS) void wka_kernel(const WkLaunch L, const AttFwdArgs g, const int natt, const int att_last) {
    extern __shared__ __attribute__((aligned(16))) char wk_smem[];
    int bx = blockIdx.x;
    if (att_last) {
        const int ngemm = (int)gridDim.x - natt;
        bx = bx >= ngemm ? bx - ngemm : bx + natt;
    }
    if (bx < natt) {
        att_fwd_block<SK_THREADS, SKA_PROJ_UNROLL>(g, bx / g.esplit, bx % g.esplit, reinterpret_cast<float*>(wk_smem));
        return;
    }
    bx -= natt;
    int j = 0;
#pragma unroll
    for (int q = 0; q < SK_MAXJOB - 1; ++q)
        if (q < L.njobs - 1 && bx >= L.wg_end[q]) j = q + 1;
    bx -= (j > 0 ? L.wg_end[j - 1] : 0);
    const SkJob& job = L.job[j];
    if (L.ncw[j] == 8) wk_body<8>(job, bx, L.wgh[j], wk_smem);
    else wk_body<4>(job, bx, L.wgh[j], wk_smem);
}

// ---- fused backward tick (plans.hip schedule 7, LSTM layers, bf16 operands) ------------------------------------------
// Schedule 0 runs a backward tick as two launches: the attention backward + the elementwise state backward of every
// active layer (att_state_bwd_kernel: a 16 us chain of dependent round trips on 64-192 CUs at cfg4) and then the wide
// launch of all transposed products (~40 us), every one of which reads the dP rows the first launch wrote.  Here the row
// blocks LEAD the grid of ONE launch (1024 threads, as in att_state_bwd_kernel); each chain publishes its dP rows
// write-through and arrives on its flag; the wide workgroups behind them (8 of the block's 16 waves, the others exit at
// once) wait for the flag of the chain that feeds them before their first load and read dP through sc1 loads.  The
// upper layers' chains are elementwise and done in a few microseconds, so their products start almost at once; layer
// 0's products -- the shortest of the launch, last in the grid -- start on the CUs the attention rows free.
struct WkbProducers {
    AttBwdArgs att;
    LstmStateBwdArgs sa;
    int att_rows, l0_chain, nprod;
    int row_off;        // floats: where the publishing state backward's staging row starts in LDS (behind the attention's)
    unsigned* flag[4];  // per chain
};

__global__ __launch_bounds__(ATTB_THREADS) void wkb_kernel(const WkLaunch L, const WkbProducers P) {
    extern __shared__ __attribute__((aligned(16))) char wkb_smem[];
    int bx = blockIdx.x;
    if (bx < P.nprod) {
        float* sm = reinterpret_cast<float*>(wkb_smem);
        float* row = sm + P.row_off;  // [4H] staging row
        int ch, m;
        if (bx < P.att_rows) {
            att_bwd_row(P.att, bx, sm);
            __syncthreads();
            ch = P.l0_chain; m = bx;
        } else {
            const int idx = bx - P.att_rows;
            ch = idx / P.sa.B; m = idx % P.sa.B;
            if (P.att_rows > 0 && P.l0_chain >= 0 && ch >= P.l0_chain) ++ch;  // skip the chain fused behind the attention
        }
        if (ch >= 0 && ch < P.sa.nchain) {
            lstm_state_bwd_row_pub(P.sa.chain[ch], m, P.sa.H, threadIdx.x, ATTB_THREADS, row);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave drains its write-through stores ...
            __syncthreads();
            if (threadIdx.x == 0)                              // ... then one lane arrives for the row
                (void)__hip_atomic_fetch_add(P.flag[ch], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lstm_state_bwd_row_copy16(P.sa.chain[ch], m, P.sa.H, threadIdx.x, ATTB_THREADS, row);  // (behind the hand-off)
        }
        return;
    }
    if (threadIdx.x >= SK_THREADS) return;  // the wide workgroups run on eight waves
    bx -= P.nprod;
    int j = 0;
#pragma unroll
    for (int q = 0; q < SK_MAXJOB - 1; ++q)
        if (q < L.njobs - 1 && bx >= L.wg_end[q]) j = q + 1;
    bx -= (j > 0 ? L.wg_end[j - 1] : 0);
    const SkJob& job = L.job[j];
    if (L.ncw[j] == 8) wk_body<8, true>(job, bx, L.wgh[j], wkb_smem);
    else wk_body<4, true>(job, bx, L.wgh[j], wkb_smem);
}

// Would wk_try_launch take a bf16 launch whose jobs have M rows, ncols output columns in total and K segments of H and E
// rows?  (plans.hip asks before it commits a plan to a schedule only the wide kernel can run.)
bool sk_wide_takes(int M, int ncols, int H, int E) {
    const char* e = getenv("PARROT_WK");
    const int enabled = e ? atoi(e) : 1;
    if (!enabled || M < 1 || M > 64 || (H % WK_STAGE) || (E % WK_STAGE)) return false;
    return ncols >= 4096 || enabled >= 2;
}

// Takes the launch when every job is a bf16-operand LSTM / LINEAR job over <= 64 rows with 64-deep K segments.
// att != null: the attention step rides in the same launch (wka_kernel); `reserve` CUs are left to its blocks.
// Legality + layout of a wide launch: W (jobs that wait go last), its workgroup count t, whether any job waits.
// nlead = workgroups that lead the grid beside / before the wide ones (attention blocks, state-backward rows);
// lead_waiters: some job waits for them (then the non-waiting jobs are sized to fit beside them).
static bool wk_build(const SkLaunch& Lin, int nlead, bool has_lead, WkLaunch& W, int& t, bool& any_flag) {
    const char* e = getenv("PARROT_WK");  // 0: never; 1 (default): launches with >= 4096 output columns; 2: whenever legal
    const int enabled = e ? atoi(e) : 1;
    if (!enabled) return false;
    long long work = 0;
    for (int q = 0; q < Lin.njobs; ++q) {
        const SkJob& j = Lin.job[q];
        if (j.wait_flag && !has_lead) return false;  // a flag needs its producers in the launch
        if (j.wait_flag && (j.wait_all ? j.nseg != 1 : j.nseg < 2)) return false;  // (wait_all: 1, or 2 = behind the attention rows)
        if (j.seg[0].b_kcontig != 3 || !j.aligned || j.M > 64 || j.M < 1) return false;
        if (j.epi != SK_EPI_LSTM && j.epi != SK_EPI_LINEAR) return false;
        if (j.epi == SK_EPI_LINEAR && (j.N & 15)) return false;
        for (int s = 0; s < j.nseg; ++s)
            if (j.seg[s].K % WK_STAGE) return false;
        if (j.ksplit > 1 && (j.ksplit > 4 || j.nseg != 1 || j.epi != SK_EPI_LINEAR || j.act || !j.o1 || (j.ksplit > 2 && !j.kout2) ||
                             (j.ksplit > 3 && !j.kout3) || j.seg[0].K % (j.ksplit * WK_STAGE)))
            return false;
        work += (long long)j.N;
    }
    if (work < 4096 && enabled < 2 && !Lin.force_wide) return false;
    memset(&W, 0, sizeof(W));
    W.njobs = Lin.njobs;
    int ksum[SK_MAXJOB], tiles[SK_MAXJOB];
    int units = 0, units_flagged = 0;
    any_flag = false;
    {   // jobs that wait for the lead blocks go LAST in the grid -- in the caller's order among themselves -- so that the
        // others take the CUs beside the lead blocks and the waiting ones start on the CUs those free, their flag set
        int n = 0;
        for (int pass = 0; pass < 2; ++pass)
            for (int q = 0; q < Lin.njobs; ++q) {
                const bool late = Lin.job[q].wait_flag != nullptr && Lin.job[q].wait_all != 1;
                if (late != (pass == 1)) continue;
                W.job[n] = Lin.job[q];
                tiles[n] = Lin.tile_end[q];
                ++n;
            }
    }
    for (int q = 0; q < Lin.njobs; ++q) {
        W.ncw[q] = 4;
        ksum[q] = 0;
        for (int s = 0; s < W.job[q].nseg; ++s) ksum[q] += W.job[q].seg[s].K;
        const int parts = W.job[q].ksplit > 1 ? W.job[q].ksplit : 1;
        units += ceil_div(tiles[q], 4) * parts;
        if (W.job[q].wait_flag && W.job[q].wait_all != 1) { any_flag = true; units_flagged += ceil_div(tiles[q], 4) * parts; }
    }
    // one workgroup per CU and launch: widen the jobs with the shortest K to 128 columns until the launch fits (with
    // jobs that wait behind their other segments: until the OTHER jobs fit beside the lead blocks; the waiting ones, at
    // most a CU round of their own, follow in the lead blocks' place)
    auto fits = [&]() {
        if (any_flag) return units - units_flagged <= 256 - nlead && units_flagged <= 256;
        return units <= 256 - (nlead < 128 ? nlead : 128);
    };
    while (!fits()) {
        int best = -1;
        for (int q = 0; q < Lin.njobs; ++q)
            if (W.ncw[q] == 4 && (best < 0 || ksum[q] < ksum[best])) best = q;
        if (best < 0) break;
        const int gain = (ceil_div(tiles[best], 4) - ceil_div(tiles[best], 8)) * (W.job[best].ksplit > 1 ? W.job[best].ksplit : 1);
        units -= gain;
        if (W.job[best].wait_flag && W.job[best].wait_all != 1) units_flagged -= gain;
        W.ncw[best] = 8;
    }
    t = 0;
    for (int q = 0; q < Lin.njobs; ++q) {
        W.wgh[q] = ceil_div(tiles[q], W.ncw[q]);
        t += W.wgh[q] * (W.job[q].ksplit > 1 ? W.job[q].ksplit : 1);
        W.wg_end[q] = t;
    }
    return true;
}

static bool wk_try_launch(const SkLaunch& Lin, hipStream_t stream, int* rc, const AttFwdArgs* att = nullptr) {
    for (int q = 0; q < Lin.njobs; ++q)
        if (Lin.job[q].wait_all) return false;  // (the fused backward tick has its own entry point)
    const int natt = att ? att->B * att->esplit : 0;
    WkLaunch W;
    int t;
    bool any_flag;
    if (!wk_build(Lin, natt, att != nullptr, W, t, any_flag)) return false;
    size_t lds = 2 * 64 * WK_PITCH;  // two stage buffers
    if (att) {
        const size_t alds = att_fwd_lds(att->U);
        if (alds > lds) lds = alds;
        const int att_last = any_flag ? 0 : 1;  // producers lead the grid when somebody waits for them
        if (g_prof.on) {
            SkProfRec r;
            (void)hipEventCreate(&r.e0);
            (void)hipEventCreate(&r.e1);
            sk_account(Lin, r.flops, r.bytes);
            hipExtLaunchKernelGGL(wka_kernel, dim3(t + natt), dim3(SK_THREADS), lds, stream, r.e0, r.e1, 0, W, *att, natt, att_last);
            r.hetero = 1;
        g_prof.recs.push_back(r);
        } else {
            hipLaunchKernelGGL(wka_kernel, dim3(t + natt), dim3(SK_THREADS), lds, stream, W, *att, natt, att_last);
        }
        *rc = (int)hipGetLastError();
        return true;
    }
    if (g_prof.on) {
        SkProfRec r;
        (void)hipEventCreate(&r.e0);
        (void)hipEventCreate(&r.e1);
        sk_account(Lin, r.flops, r.bytes);
        hipExtLaunchKernelGGL(wk_kernel, dim3(t), dim3(SK_THREADS), lds, stream, r.e0, r.e1, 0, W);
        g_prof.recs.push_back(r);
    } else {
        hipLaunchKernelGGL(wk_kernel, dim3(t), dim3(SK_THREADS), lds, stream, W);
    }
    *rc = (int)hipGetLastError();
    return true;
}

int sk_launch_bwd_fused(const SkLaunch& Lin, const AttBwdArgs* att, const LstmStateBwdArgs& sa, int l0_chain,
                        unsigned* const* flags, hipStream_t stream) {
    static_assert(sizeof(WkLaunch) + sizeof(WkbProducers) <= 4096, "kernel arguments of wkb_kernel");
    if (sa.nchain < 1 || sa.nchain > 4 || sa.B < 1 || (sa.H & 3)) return PH_ERR_UNSUPPORTED;
    WkbProducers P;
    memset(&P, 0, sizeof(P));
    P.sa = sa;
    P.l0_chain = -1;
    size_t plds = 0;
    if (att) {
        if (att->A < 1 || att->A > ATT_MAXA || att->B != sa.B || att->U < 1 || att->E < 1 || l0_chain < 0 || l0_chain >= sa.nchain)
            return PH_ERR_BADARG;
        P.att = *att;
        P.att_rows = att->B;
        P.l0_chain = l0_chain;
        plds = att_bwd_lds(att->U, att->E);
    }
    P.row_off = (int)(plds / sizeof(float));
    plds += (size_t)4 * sa.H * sizeof(float);  // the staging row of the publishing state backward
    P.nprod = P.att_rows + (sa.nchain - (att ? 1 : 0)) * sa.B;
    for (int q = 0; q < sa.nchain; ++q) {
        if (!flags[q] || ((uintptr_t)sa.chain[q].dP & 15)) return PH_ERR_UNSUPPORTED;
        P.flag[q] = flags[q];
    }
    for (int q = 0; q < Lin.njobs; ++q)
        if (!Lin.job[q].wait_flag || !Lin.job[q].wait_all) return PH_ERR_BADARG;  // every product reads a dP row block

    WkLaunch W;
    int t;
    bool any_flag;
    // The upper layers' row blocks are elementwise and gone in a few microseconds: their products are sized as if they had
    // the chip beside the attention rows (~16 us on one CU each, 1024 threads: nothing shares a CU with them).  The
    // products behind the attention rows (wait_all == 2: layer 0's) go last in the grid, at their narrow width, and
    // start on the CUs those rows free.
    if (!wk_build(Lin, P.att_rows, true, W, t, any_flag)) return PH_ERR_UNSUPPORTED;
    size_t lds = 2 * 64 * WK_PITCH;
    if (plds > lds) lds = plds;
    if (lds > 160 * 1024) return PH_ERR_UNSUPPORTED;
    static bool allowed = false;
    if (!allowed) {
        (void)hipFuncSetAttribute((const void*)wkb_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        allowed = true;
    }
    if (g_prof.on) {
        SkProfRec r;
        (void)hipEventCreate(&r.e0);
        (void)hipEventCreate(&r.e1);
        sk_account(Lin, r.flops, r.bytes);
        hipExtLaunchKernelGGL(wkb_kernel, dim3(P.nprod + t), dim3(ATTB_THREADS), lds, stream, r.e0, r.e1, 0, W, P);
        r.hetero = 1;
        g_prof.recs.push_back(r);
    } else {
        hipLaunchKernelGGL(wkb_kernel, dim3(P.nprod + t), dim3(ATTB_THREADS), lds, stream, W, P);
    }
    return (int)hipGetLastError();
}


template <int MB, int NB>
static void sk_dispatch(const SkLaunch& L, dim3 grid, size_t lds, hipStream_t stream) {
    if (g_prof.on) {
        SkProfRec r;
        (void)hipEventCreate(&r.e0);
        (void)hipEventCreate(&r.e1);
        sk_account(L, r.flops, r.bytes);
        hipExtLaunchKernelGGL((sk_kernel<MB, NB>), grid, dim3(SK_THREADS), lds, stream, r.e0, r.e1, 0, L);
        g_prof.recs.push_back(r);
    } else {
        hipLaunchKernelGGL((sk_kernel<MB, NB>), grid, dim3(SK_THREADS), lds, stream, L);
    }
}

// Tile shape, grid and finished descriptor (workgroup prefix, z-mode) of a launch.
void sk_prepare(const SkLaunch& Lin, SkLaunch& L, dim3& grid_out, size_t& lds_out, int& mbnb_out) {
    L = Lin;
    int maxM = 0;
    bool nb2_ok = true;
    for (int q = 0; q < L.njobs; ++q) {
        maxM = L.job[q].M > maxM ? L.job[q].M : maxM;
        if (!L.job[q].aligned) nb2_ok = false;
    }
    // Tile shape per workgroup: (16*mb rows) x (16*nb columns).  Per-CU load bandwidth (~50 GB/s measured)
    // is what limits this kernel, so prefer the shape with the fewest operand bytes per flop
    // (1/(16 mb) + 1/(16 nb)) among those that still give the chip >= 224 workgroups; if none does,
    // take the shape with the most workgroups.
    const int full_thr = Lin.full_wgs > 0 ? Lin.full_wgs : 224;
    const int mbmax = maxM >= 49 ? 4 : (maxM + 15) / 16;
    int best_mb = 1, best_nb = 1, best_wg = -1;
    double best_cost = 1e30;
    const int mbs[4] = {4, 3, 2, 1};
    for (int a = 0; a < 4; ++a) {
        const int mb = mbs[a];
        if (mb > mbmax) continue;
        // do not pad the batch by more than one 16-row MFMA block (mb = 3 on 64 rows would compute 96)
        if (ceil_div(maxM, 16 * mb) * 16 * mb - maxM >= 16) continue;
        for (int nb = 2; nb >= 1; --nb) {
            if (nb == 2 && !nb2_ok) continue;
            int wgx = 0;
            for (int q = 0; q < L.njobs; ++q) wgx += ceil_div(L.tile_end[q], nb);
            const int wg = wgx * ceil_div(maxM, 16 * mb);
            // operand bytes per flop ~ 1/mb + 1/nb, but 48/64-row tiles measured slower than 32-row ones whenever both
            // fill the chip (cfg2 merged launches, and the 2300-workgroup launches of a 3 x LSTM-1536 decoder:
            // <2,2> 183 k frames/s vs <4,2> 154 k, <4,1> 162 k, <3,2> 152 k), so rows beyond 32 earn no credit
            const double cost = 1.0 / (mb > 2 ? 2 : mb) + 1.0 / nb + (mb > 2 ? 0.1 : 0.0);
            const bool full = wg >= full_thr, best_full = best_wg >= full_thr;
            bool better;
            if (full != best_full) better = full;
            else if (full) better = cost < best_cost - 1e-9;
            else better = wg > best_wg;
            if (better) { best_mb = mb; best_nb = nb; best_wg = wg; best_cost = cost; }
        }
    }
    if (Lin.force_tile > 0) {  // the plan's choice for this launch
        const int fmb = Lin.force_tile / 10, fnb = Lin.force_tile % 10;
        if (fmb >= 1 && fmb <= 4 && (fnb == 1 || (fnb == 2 && nb2_ok)) && 16 * (fmb - 1) < maxM) {
            best_mb = fmb; best_nb = fnb;
        }
    }
    const int mb = best_mb, nb = best_nb;
    int t = 0, wmax = 0;
    bool equal = true;
    for (int q = 0; q < L.njobs; ++q) {
        const int w = ceil_div(L.tile_end[q], nb);
        if (q > 0 && w != wmax) equal = false;
        wmax = w > wmax ? w : wmax;
        t += w;
        L.tile_end[q] = t;  // prefix of workgroups (used when !zmode)
    }
    L.zmode = equal ? 1 : 0;
    grid_out = dim3(equal ? wmax : t, ceil_div(maxM, 16 * mb), equal ? L.njobs : 1);
    lds_out = (size_t)SK_NW * mb * nb * 64 * sizeof(f32x4);
    mbnb_out = mb * 10 + nb;
}

int sk_launch(const SkLaunch& Lin, hipStream_t stream) {
    for (int q = 0; q < Lin.njobs; ++q)
        if (Lin.job[q].wait_flag) return PH_ERR_BADARG;  // a flag needs its producers in the launch: sk_launch_att
    {
        int rc = 0;
        if (wk_try_launch(Lin, stream, &rc)) return rc;
        for (int q = 0; q < Lin.njobs; ++q)
            if (Lin.job[q].ksplit > 1) return PH_ERR_UNSUPPORTED;  // K parts exist in the wide kernel only
    }
    SkLaunch L;
    dim3 grid;
    size_t lds;
    int mbnb;
    sk_prepare(Lin, L, grid, lds, mbnb);
    switch (mbnb) {
        case 11: sk_dispatch<1, 1>(L, grid, lds, stream); break;
        case 12: sk_dispatch<1, 2>(L, grid, lds, stream); break;
        case 21: sk_dispatch<2, 1>(L, grid, lds, stream); break;
        case 22: sk_dispatch<2, 2>(L, grid, lds, stream); break;
        case 31: sk_dispatch<3, 1>(L, grid, lds, stream); break;
        case 32: sk_dispatch<3, 2>(L, grid, lds, stream); break;
        case 41: sk_dispatch<4, 1>(L, grid, lds, stream); break;
        default: sk_dispatch<4, 2>(L, grid, lds, stream); break;
    }
    return (int)hipGetLastError();
}

template <int MB, int NB>
static void ska_allow_lds() {
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute((const void*)ska_kernel<MB, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        done = true;
    }
}

template <int MB, int NB>
static void ska_dispatch(const SkLaunch& L, const AttFwdArgs& g, int natt_x, dim3 grid, size_t lds, hipStream_t stream) {
    // Order of the two kinds of workgroups in the grid.  Without in-launch dependencies the GEMM workgroups (the long
    // ones) go first and the attention fills in behind them (measured: forward scan 28.0 -> 25.3 ms at cfg2); a job that
    // waits for the attention needs its producers dispatched first.
    ska_allow_lds<MB, NB>();
    int att_last = 1;
    for (int q = 0; q < L.njobs; ++q)
        if (L.job[q].wait_flag) att_last = 2;  // producers first, and all of them in grid row 0
    if (g_prof.on) {
        SkProfRec r;
        (void)hipEventCreate(&r.e0);
        (void)hipEventCreate(&r.e1);
        sk_account(L, r.flops, r.bytes);
        hipExtLaunchKernelGGL((ska_kernel<MB, NB>), grid, dim3(SK_THREADS), lds, stream, r.e0, r.e1, 0, L, g, natt_x,
                              att_last);
        r.hetero = 1;
        g_prof.recs.push_back(r);
    } else {
        hipLaunchKernelGGL((ska_kernel<MB, NB>), grid, dim3(SK_THREADS), lds, stream, L, g, natt_x, att_last);
    }
}

// One launch for the attention step `att` and the step-GEMM jobs of L (njobs may be 0: attention alone).
int sk_launch_att(const SkLaunch& Lin, const AttFwdArgs& att, hipStream_t stream) {
    AttFwdArgs g = att;
    {
        const int rc = att_fwd_check(g);
        if (rc != 0) return rc;
    }
    if (Lin.njobs < 1) return att_fwd_launch(att, stream);
    {   // bf16 launches the wide kernel takes (a job that waits for the attention keeps its place behind it there)
        int rc = 0;
        if (wk_try_launch(Lin, stream, &rc, &g)) return rc;
        for (int q = 0; q < Lin.njobs; ++q)  // (sk_body's flagged tail multiplies f32 operands only)
            if (Lin.job[q].wait_flag && Lin.job[q].seg[0].b_kcontig == 3) return PH_ERR_UNSUPPORTED;
    }
    SkLaunch L;
    dim3 grid;
    size_t lds;
    int mbnb;
    sk_prepare(Lin, L, grid, lds, mbnb);
    // sk_prepare may have chosen the z-grid; this kernel always walks the prefix table
    if (L.zmode) {
        const int per = (int)grid.x;
        for (int q = 0; q < L.njobs; ++q) L.tile_end[q] = per * (q + 1);
        grid.x = (unsigned)(per * L.njobs);
        grid.z = 1;
        L.zmode = 0;
    }
    const int natt = g.B * g.esplit;
    bool flagged = false;
    for (int q = 0; q < L.njobs; ++q)
        if (L.job[q].wait_flag) flagged = true;
    // A job that waits for the attention inside the launch needs every producer dispatched BEFORE any waiter.
    // Workgroups are dispatched row by row (blockIdx.x fastest), so with several grid rows the producers must all sit
    // at the head of row 0 (ska_kernel, att_last == 2; rows y > 0 return at once there): spread over the rows they
    // would queue behind all of row 0's waiters, and a chip full of waiters would spin until the 1 s trap.
    const int natt_x = flagged ? natt : ceil_div(natt, (int)grid.y);
    grid.x += (unsigned)natt_x;
    const size_t alds = att_fwd_lds(g.U);
    if (alds > lds) lds = alds;
    switch (mbnb) {
        case 11: ska_dispatch<1, 1>(L, g, natt_x, grid, lds, stream); break;
        case 12: ska_dispatch<1, 2>(L, g, natt_x, grid, lds, stream); break;
        case 21: ska_dispatch<2, 1>(L, g, natt_x, grid, lds, stream); break;
        case 22: ska_dispatch<2, 2>(L, g, natt_x, grid, lds, stream); break;
        case 31: ska_dispatch<3, 1>(L, g, natt_x, grid, lds, stream); break;
        case 32: ska_dispatch<3, 2>(L, g, natt_x, grid, lds, stream); break;
        case 41: ska_dispatch<4, 1>(L, g, natt_x, grid, lds, stream); break;
        default: ska_dispatch<4, 2>(L, g, natt_x, grid, lds, stream); break;
    }
    return (int)hipGetLastError();
}

template <int MB, int NB>
static void skb_dispatch(const SkLaunch& L, const AttBwdArgs& g, const GruStateBwdArgs& sa, int att_rows, int l0_chain,
                         int nlead, int nlead_x, int rpb, dim3 grid, size_t lds, hipStream_t stream) {
    static bool allowed = false;
    if (!allowed) {
        (void)hipFuncSetAttribute((const void*)skb_kernel<MB, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        allowed = true;
    }
    if (g_prof.on) {
        SkProfRec r;
        (void)hipEventCreate(&r.e0);
        (void)hipEventCreate(&r.e1);
        sk_account(L, r.flops, r.bytes);
        hipExtLaunchKernelGGL((skb_kernel<MB, NB>), grid, dim3(ATTB_THREADS), lds, stream, r.e0, r.e1, 0, L, g, sa, att_rows,
                              l0_chain, nlead, nlead_x, rpb);
        r.hetero = 1;
        g_prof.recs.push_back(r);
    } else {
        hipLaunchKernelGGL((skb_kernel<MB, NB>), grid, dim3(ATTB_THREADS), lds, stream, L, g, sa, att_rows, l0_chain, nlead,
                           nlead_x, rpb);
    }
}

// Attention backward (or null) + GRU state backward of all chains + the step-GEMM jobs of L in ONE launch (skb_kernel).
int sk_launch_bwd_hetero(const SkLaunch& Lin, const AttBwdArgs* att, const GruStateBwdArgs& sa, int l0_chain,
                         hipStream_t stream) {
    static_assert(sizeof(SkLaunch) + sizeof(AttBwdArgs) + sizeof(GruStateBwdArgs) + 32 <= 4096, "kernel arguments of skb_kernel");
    if (sa.nchain < 1 || sa.nchain > 4 || Lin.njobs < 1) return PH_ERR_BADARG;
    for (int q = 0; q < Lin.njobs; ++q)
        if (Lin.job[q].wait_flag || Lin.job[q].ksplit > 1) return PH_ERR_BADARG;
    AttBwdArgs g{};
    int att_rows = 0;
    size_t alds = 0;
    if (att) {
        g = *att;
        if (g.A < 1 || g.A > ATT_MAXA || g.B < 1 || g.U < 1 || g.E < 1 || g.B != sa.B || l0_chain < 0 || l0_chain >= sa.nchain)
            return PH_ERR_BADARG;
        alds = att_bwd_lds(g.U, g.E);
        att_rows = g.B;
    } else {
        l0_chain = -1;
    }
    // the chains that are not fused behind the attention take 4 batch rows per block: 16 CUs instead of 64 at B = 64, so
    // that (64 attention rows + 16 + 160 GEMM workgroups at cfg2) every block of the launch finds a CU at once
    const int rpb = 4;
    const int nlead = att_rows + (sa.nchain - (att ? 1 : 0)) * ceil_div(sa.B, rpb);
    SkLaunch L;
    dim3 grid;
    size_t lds;
    int mbnb;
    sk_prepare(Lin, L, grid, lds, mbnb);
    if (L.zmode) {  // this kernel always walks the prefix table
        const int per = (int)grid.x;
        for (int q = 0; q < L.njobs; ++q) L.tile_end[q] = per * (q + 1);
        grid.x = (unsigned)(per * L.njobs);
        grid.z = 1;
        L.zmode = 0;
    }
    const int nlead_x = ceil_div(nlead, (int)grid.y);
    grid.x += (unsigned)nlead_x;
    if (alds > lds) lds = alds;
    if (lds > 160 * 1024) return PH_ERR_UNSUPPORTED;
    switch (mbnb) {
        case 11: skb_dispatch<1, 1>(L, g, sa, att_rows, l0_chain, nlead, nlead_x, rpb, grid, lds, stream); break;
        case 12: skb_dispatch<1, 2>(L, g, sa, att_rows, l0_chain, nlead, nlead_x, rpb, grid, lds, stream); break;
        case 21: skb_dispatch<2, 1>(L, g, sa, att_rows, l0_chain, nlead, nlead_x, rpb, grid, lds, stream); break;
        case 22: skb_dispatch<2, 2>(L, g, sa, att_rows, l0_chain, nlead, nlead_x, rpb, grid, lds, stream); break;
        case 31: skb_dispatch<3, 1>(L, g, sa, att_rows, l0_chain, nlead, nlead_x, rpb, grid, lds, stream); break;
        case 32: skb_dispatch<3, 2>(L, g, sa, att_rows, l0_chain, nlead, nlead_x, rpb, grid, lds, stream); break;
        case 41: skb_dispatch<4, 1>(L, g, sa, att_rows, l0_chain, nlead, nlead_x, rpb, grid, lds, stream); break;
        default: skb_dispatch<4, 2>(L, g, sa, att_rows, l0_chain, nlead, nlead_x, rpb, grid, lds, stream); break;
    }
    return (int)hipGetLastError();
}
