// Training-side operators of the SampleRNN sample-level tier (gfx950): what surrounds the HIP GEMMs of
// three_tier.py:452-515 (sample_level_predictor) and :565-593 (the cross-entropy cost) and used to run as torch kernels.
//
//  * the Embedding (sampleRNN/lib/ops.py:252-266) followed by SampleLevel.L1_PrevSamples (a [FS*EMB, DIM] Linear without
//    bias) is, row by row, a sum of FS table rows: out[i] = sum_j (Emb . W1_j)[idx[i, j]].  The generation loop has always
//    used that form (samplernn.hip sr_embed_sum_kernel); training now does too: a gather-sum forward and, backward, a
//    SEGMENTED sum of the output gradient over the rows that picked table row (j, q) -- no [rows, FS*EMB] activation, no
//    K = FS*EMB product (0.67 of the 1.28 TFLOP of the sample MLP's forward at configs[4]), no scatter-add;
//  * softmax cross-entropy with integer targets (T.nnet.categorical_crossentropy(softmax(logits), target),
//    three_tier.py:565-584): log-sum-exp minus the picked logit per row, and its gradient, one wave per row.
//
// Everything sums in a fixed order (no float atomics): a training step stays reproducible bit for bit.
#include <stdlib.h>

#include "../../include/parrot_hip.h"
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ gather-sum forward
// y[i, :] = add[i, :] + sum_{j < J} tbl[j][idx[i, j]][:], summed in j order (the order of sr_embed_sum_kernel).
// Thread = one 16-byte column group of one row; a 256-thread block covers 256 / (D/4) rows (D <= 1024) or walks the
// columns (D > 1024).
__global__ __launch_bounds__(256) void gs_fwd_kernel(const float* __restrict__ tbl, const int* __restrict__ idx,
                                                     const float* __restrict__ add, int ldadd, float* __restrict__ y,
                                                     int ldy, long long N, int J, int Q, int D) {
    const int D4 = D >> 2;
    const int cpb = D4 < 256 ? D4 : 256;        // column groups a block covers at once
    const int rpb = 256 / cpb;                  // rows per block
    const int c = threadIdx.x % cpb, r = threadIdx.x / cpb;
    if (r >= rpb) return;
    const long long i = (long long)blockIdx.x * rpb + r;
    if (i >= N) return;
    const int* ix = idx + i * J;
    for (int d4 = c; d4 < D4; d4 += cpb) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (add) acc = *reinterpret_cast<const f32x4*>(add + i * ldadd + 4 * d4);
        int j = 0;
        for (; j + 4 <= J; j += 4) {  // four table rows in flight
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                v[u] = *reinterpret_cast<const f32x4*>(tbl + ((size_t)(j + u) * Q + ix[j + u]) * D + 4 * d4);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += v[u];
        }
        for (; j < J; ++j) acc += *reinterpret_cast<const f32x4*>(tbl + ((size_t)j * Q + ix[j]) * D + 4 * d4);
        *reinterpret_cast<f32x4*>(y + i * ldy + 4 * d4) = acc;
    }
}

// ------------------------------------------------------------------------------------------------ segmented sum backward
// dtbl[j][q][:] = sum over the rows i with idx[i, j] == q of dy[i, :].
// perm[j][.] lists the rows sorted by (idx[., j], row) -- a stable sort, so the order of the addends is a function of the
// indices alone --, offs[j][q] is the first position of bin q in that order (offs[j][Q] = N).
// Pass 1 (grid: chunks of GS_CH sorted positions x J): a block walks its positions in order and keeps one running sum per
// thread (a 16-byte column group); where the bin changes, the run's sum goes to slot (q + chunk) of the partial buffer --
// bins are met in ascending order by ascending chunks, so no two runs share a slot and there are at most nchunks + Q.
// Pass 2 (grid: Q x J): adds the slots of a bin's chunks in chunk order.  Load-balanced whatever the histogram is.
constexpr int GS_CH = 128;

__global__ __launch_bounds__(256) void gs_bwd_chunk_kernel(const float* __restrict__ dy, int lddy, const int* __restrict__ perm,
                                                           const int* __restrict__ offs, float* __restrict__ part, long long N,
                                                           int Q, int D, int nslots) {
    const int j = blockIdx.y, c = blockIdx.x;
    const int* pj = perm + (size_t)j * N;
    const int* oj = offs + (size_t)j * (Q + 1);
    const long long p0 = (long long)c * GS_CH, p1 = min(N, p0 + GS_CH);
    // bin of the chunk's first position: the last q with offs[q] <= p0 (uniform over the block)
    int lo = 0, hi = Q;  // invariant: offs[lo] <= p0 < offs[hi]   (offs[0] = 0, offs[Q] = N > p0)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (oj[mid] <= p0) lo = mid; else hi = mid;
    }
    int q = lo;
    const int D4 = D >> 2;
    float* pslab = part + (size_t)j * nslots * D;
    for (int d4 = threadIdx.x; d4 < D4; d4 += 256) {
        long long p = p0;
        int qq = q;
        while (p < p1) {
            while (oj[qq + 1] <= p) ++qq;  // (empty bins)
            const long long end = min(p1, (long long)oj[qq + 1]);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (; p + 8 <= end; p += 8) {  // eight rows in flight, added in position order
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(dy + (size_t)pj[p + u] * lddy + 4 * d4);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u];
            }
            for (; p < end; ++p) acc += *reinterpret_cast<const f32x4*>(dy + (size_t)pj[p] * lddy + 4 * d4);
            *reinterpret_cast<f32x4*>(pslab + (size_t)(qq + c) * D + 4 * d4) = acc;
        }
    }
}

__global__ __launch_bounds__(256) void gs_bwd_reduce_kernel(const float* __restrict__ part, const int* __restrict__ offs,
                                                            float* __restrict__ dtbl, int Q, int D, int nslots,
                                                            int accumulate) {
    const int j = blockIdx.y, q = blockIdx.x;
    const int* oj = offs + (size_t)j * (Q + 1);
    const int a = oj[q], b = oj[q + 1];
    const float* pslab = part + (size_t)j * nslots * D;
    float* out = dtbl + ((size_t)j * Q + q) * D;
    for (int d4 = threadIdx.x; d4 < (D >> 2); d4 += 256) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (b > a) {
            const int c0 = a / GS_CH, c1 = (b - 1) / GS_CH;
            for (int c = c0; c <= c1; ++c) acc += *reinterpret_cast<const f32x4*>(pslab + (size_t)(q + c) * D + 4 * d4);
        }
        f32x4* o = reinterpret_cast<f32x4*>(out + 4 * d4);
        *o = accumulate ? *o + acc : acc;
    }
}

// ------------------------------------------------------------------------------------------------ softmax cross-entropy
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = fmaxf(v, __shfl_xor(v, s, 64));
    return v;
}

// One wave per row: lse[i] = log sum_q exp(x[i, q]) (max-shifted), ce[i] = lse[i] - x[i, target[i]].
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ x, int ld, const int* __restrict__ target,
                                                     long long rows, int Q, float* __restrict__ lse, float* __restrict__ ce) {
    const int lane = threadIdx.x & 63;
    const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= rows) return;
    const float* xi = x + i * ld;
    float m = -INFINITY;
    for (int q = lane; q < Q; q += 64) m = fmaxf(m, xi[q]);
    m = wave_max(m);
    float s = 0.f;
    for (int q = lane; q < Q; q += 64) s += expf(xi[q] - m);
    s = wave_sum(s);
    if (lane == 0) {
        const float l = logf(s) + m;
        lse[i] = l;
        ce[i] = l - xi[target[i]];
    }
}

// dx[i, q] = rowscale[i] * (exp(x[i, q] - lse[i]) - [q == target[i]])
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ x, int ld, const int* __restrict__ target,
                                                     const float* __restrict__ lse, const float* __restrict__ rowscale,
                                                     long long rows, int Q, float* __restrict__ dx, int ldd) {
    const int lane = threadIdx.x & 63;
    const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= rows) return;
    const float* xi = x + i * ld;
    float* di = dx + i * ldd;
    const float l = lse[i], g = rowscale[i];
    const int t = target[i];
    for (int q = lane; q < Q; q += 64) di[q] = g * (expf(xi[q] - l) - (q == t ? 1.f : 0.f));
}

// y = dy where gate > 0, else 0 (ReLU backward through the saved activation), 16 bytes per thread
__global__ __launch_bounds__(256) void relu_gate_kernel(const float* __restrict__ dy, const float* __restrict__ gate,
                                                        float* __restrict__ out, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4 d = reinterpret_cast<const f32x4*>(dy)[i], g = reinterpret_cast<const f32x4*>(gate)[i];
    f32x4 o;
#pragma unroll
    for (int u = 0; u < 4; ++u) o[u] = g[u] > 0.f ? d[u] : 0.f;
    reinterpret_cast<f32x4*>(out)[i] = o;
}


// Weight-norm fold (sampleRNN/lib/ops.py:101-110): W_eff[k][n] = W[k][n] * g[n] / ||W[:, n]||_2.
// Two launches each way, both chip-wide: (1) column sums of W^2 (backward: of dW_eff W) over WN_KS row slices -- workgroup
// (column block of 64, slice): lane = column (256-byte row segments), 4 waves x 4 rows in flight, added across the waves
// through LDS in wave order -> ws[slice][n]; (2) workgroup (column block, 64-row chunk): every lane adds its column's WN_KS partial sums in slice
// order once (L2 hits) and scales its rows.  No atomics: the sums have one order.
// Backward: dg[n] = dot[n] / ||W_n||,  dW[k][n] = (g[n] / ||W_n||) * (dW_eff[k][n] - W[k][n] * dot[n] / ||W_n||^2).
constexpr int WN_KS = 16;

template <bool DOT>
__global__ __launch_bounds__(256) void wn_colsum_kernel(const float* __restrict__ W, int ld, const float* __restrict__ D, int ldd,
                                                       float* __restrict__ ws, int K, int N) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane, nc = n < N ? n : N - 1;
    const int rows = (K + WN_KS - 1) / WN_KS, k0 = blockIdx.y * rows, k1 = min(K, k0 + rows);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = k0 + wave;
    for (; k + 12 < k1; k += 16) {
        const float a = W[(size_t)k * ld + nc], b = W[(size_t)(k + 4) * ld + nc], c = W[(size_t)(k + 8) * ld + nc],
                    d = W[(size_t)(k + 12) * ld + nc];
        if (DOT) {
            s0 = fmaf(D[(size_t)k * ldd + nc], a, s0); s1 = fmaf(D[(size_t)(k + 4) * ldd + nc], b, s1);
            s2 = fmaf(D[(size_t)(k + 8) * ldd + nc], c, s2); s3 = fmaf(D[(size_t)(k + 12) * ldd + nc], d, s3);
        } else {
            s0 = fmaf(a, a, s0); s1 = fmaf(b, b, s1); s2 = fmaf(c, c, s2); s3 = fmaf(d, d, s3);
        }
    }
    for (; k < k1; k += 4) {
        const float a = W[(size_t)k * ld + nc];
        s0 = fmaf(DOT ? D[(size_t)k * ldd + nc] : a, a, s0);
    }
    part[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && n < N) ws[(size_t)blockIdx.y * N + n] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

__device__ __forceinline__ float wn_total(const float* __restrict__ ws, int N, int n) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < WN_KS; ++q) t += ws[(size_t)q * N + n];
    return t;
}

// workgroup (column block of 64, chunk of WN_RC rows): lane = column -- its total once --, the four waves deal the rows
constexpr int WN_RC = 64;
__global__ __launch_bounds__(256) void wn_scale_fwd_kernel(const float* __restrict__ W, int ld, const float* __restrict__ g,
                                                          const float* __restrict__ ws, float* __restrict__ out, int ldo,
                                                          float* __restrict__ norm, int K, int N) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    if (n >= N) return;
    const float nrm = sqrtf(wn_total(ws, N, n));
    if (blockIdx.y == 0 && wave == 0 && norm) norm[n] = nrm;
    const float sc = g[n] / nrm;
    const int k0 = blockIdx.y * WN_RC, k1 = min(K, k0 + WN_RC);
    for (int k = k0 + wave; k < k1; k += 4) out[(size_t)k * ldo + n] = W[(size_t)k * ld + n] * sc;
}

__global__ __launch_bounds__(256) void wn_scale_bwd_kernel(const float* __restrict__ W, int ld, const float* __restrict__ g,
                                                          const float* __restrict__ norm, const float* __restrict__ ws,
                                                          const float* __restrict__ dWe, int ldd, float* __restrict__ dW, int lddw,
                                                          float* __restrict__ dg, int K, int N, int acc_w, int acc_g) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    if (n >= N) return;
    const float dot = wn_total(ws, N, n), nrm = norm[n];
    if (blockIdx.y == 0 && wave == 0) dg[n] = acc_g ? dg[n] + dot / nrm : dot / nrm;
    const float sc = g[n] / nrm, q = dot / (nrm * nrm);
    const int k0 = blockIdx.y * WN_RC, k1 = min(K, k0 + WN_RC);
    for (int k = k0 + wave; k < k1; k += 4) {
        const float v = sc * (dWe[(size_t)k * ldd + n] - W[(size_t)k * ld + n] * q);
        float* o = dW + (size_t)k * lddw + n;
        *o = acc_w ? *o + v : v;
    }
}

}  // namespace

extern "C" {

int parrot_gather_sum_fwd(const float* tbl, const int* idx, const float* add, int ldadd, float* y, int ldy, long long N,
                          int J, int Q, int D, void* stream) { PH_ENTRY();
    if (!tbl || !idx || !y || N < 0 || J < 1 || Q < 1 || D < 4 || (D & 3) || (ldy & 3) || (add && (ldadd & 3)))
        return PH_ERR_BADARG;
    if (N == 0) return 0;
    const int D4 = D >> 2, cpb = D4 < 256 ? D4 : 256, rpb = 256 / cpb;
    const long long blocks = (N + rpb - 1) / rpb;
    if (blocks > 0x7fffffffll) return PH_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(gs_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, tbl, idx, add, ldadd, y, ldy,
                       N, J, Q, D);
    return (int)hipGetLastError();
}

long long parrot_gather_sum_bwd_ws_floats(long long N, int J, int Q, int D) { PH_ENTRY();
    if (N < 0 || J < 1 || Q < 1 || D < 4) return 0;
    const long long nchunks = (N + GS_CH - 1) / GS_CH;
    return (long long)J * (nchunks + Q) * D;
}

int parrot_gather_sum_bwd(const float* dy, int lddy, const int* perm, const int* offs, float* dtbl, float* ws,
                          long long ws_floats, long long N, int J, int Q, int D, int accumulate, void* stream) { PH_ENTRY();
    if (!dy || !perm || !offs || !dtbl || !ws || N < 1 || J < 1 || Q < 1 || D < 4 || (D & 3) || (lddy & 3) || J > 65535)
        return PH_ERR_BADARG;
    if (ws_floats < parrot_gather_sum_bwd_ws_floats(N, J, Q, D)) return PH_ERR_BADARG;
    const long long nchunks = (N + GS_CH - 1) / GS_CH;
    if (nchunks + Q > 0x7fffffffll) return PH_ERR_UNSUPPORTED;
    const int nslots = (int)(nchunks + Q);
    hipLaunchKernelGGL(gs_bwd_chunk_kernel, dim3((unsigned)nchunks, (unsigned)J), dim3(256), 0, (hipStream_t)stream, dy, lddy,
                       perm, offs, ws, N, Q, D, nslots);
    PH_CHECK(hipGetLastError());
    hipLaunchKernelGGL(gs_bwd_reduce_kernel, dim3((unsigned)Q, (unsigned)J), dim3(256), 0, (hipStream_t)stream, ws, offs, dtbl,
                       Q, D, nslots, accumulate);
    return (int)hipGetLastError();
}

int parrot_softmax_ce_fwd(const float* logits, int ld, const int* target, long long rows, int Q, float* lse, float* ce,
                          void* stream) { PH_ENTRY();
    if (!logits || !target || !lse || !ce || rows < 0 || Q < 1 || ld < Q) return PH_ERR_BADARG;
    if (rows == 0) return 0;
    const long long blocks = (rows + 3) / 4;
    if (blocks > 0x7fffffffll) return PH_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(ce_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, logits, ld, target, rows, Q,
                       lse, ce);
    return (int)hipGetLastError();
}

int parrot_softmax_ce_bwd(const float* logits, int ld, const int* target, const float* lse, const float* rowscale,
                          long long rows, int Q, float* dlogits, int ldd, void* stream) { PH_ENTRY();
    if (!logits || !target || !lse || !rowscale || !dlogits || rows < 0 || Q < 1 || ld < Q || ldd < Q) return PH_ERR_BADARG;
    if (rows == 0) return 0;
    const long long blocks = (rows + 3) / 4;
    if (blocks > 0x7fffffffll) return PH_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, logits, ld, target, lse,
                       rowscale, rows, Q, dlogits, ldd);
    return (int)hipGetLastError();
}

int parrot_relu_gate(const float* dy, const float* gate, float* out, long long n, void* stream) { PH_ENTRY();
    if (!dy || !gate || !out || n < 0 || (n & 3)) return PH_ERR_BADARG;
    if (n == 0) return 0;
    const long long n4 = n >> 2, blocks = (n4 + 255) / 256;
    if (blocks > 0x7fffffffll) return PH_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(relu_gate_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, gate, out, n4);
    return (int)hipGetLastError();
}

long long samplernn_weightnorm_ws_floats(int N) { PH_ENTRY(); return N > 0 ? (long long)WN_KS * N : 0; }

int samplernn_weightnorm_fold(const float* W, int ld, const float* g, float* W_eff, int ldo, float* norm, float* ws, int K, int N,
                              void* stream) { PH_ENTRY();
    if (!W || !g || !W_eff || !ws || K < 1 || N < 1 || ld < N || ldo < N) return PH_ERR_BADARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(wn_colsum_kernel<false>, dim3((unsigned)((N + 63) / 64), WN_KS), dim3(256), 0, st, W, ld,
                       (const float*)nullptr, 0, ws, K, N);
    hipLaunchKernelGGL(wn_scale_fwd_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)((K + WN_RC - 1) / WN_RC)), dim3(256), 0, st,
                       W, ld, g, ws, W_eff, ldo, norm, K, N);
    return (int)hipGetLastError();
}

int samplernn_weightnorm_fold_bwd(const float* W, int ld, const float* g, const float* norm, const float* dW_eff, int ldd,
                                  float* dW, int lddw, float* dg, float* ws, int K, int N, int accumulate_w, int accumulate_g,
                                  void* stream) { PH_ENTRY();
    if (!W || !g || !norm || !dW_eff || !dW || !dg || !ws || K < 1 || N < 1 || ld < N || ldd < N || lddw < N) return PH_ERR_BADARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(wn_colsum_kernel<true>, dim3((unsigned)((N + 63) / 64), WN_KS), dim3(256), 0, st, W, ld, dW_eff, ldd, ws, K, N);
    hipLaunchKernelGGL(wn_scale_bwd_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)((K + WN_RC - 1) / WN_RC)), dim3(256), 0, st,
                       W, ld, g, norm, ws, dW_eff, ldd, dW, lddw, dg, K, N, accumulate_w, accumulate_g);
    return (int)hipGetLastError();
}

}  // extern "C"
