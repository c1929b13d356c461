// HBM-bound elementwise / reduction kernels around the recurrent step (gfx950).
//  * GRU state backward (the elementwise half of the GatedRecurrent backward step)
//  * column sums (bias gradients), sum of squares (global-norm StepClipping, train.py:100-101)
//  * fused global-norm clip + Adam over the flat parameter buffer (train.py:100-108,
//    Blocks StepClipping -> Adam, Appendix A.1 of SURVEY.md)
#include "elementwise.h"

#include <map>
#include <mutex>

namespace {

// dh: total gradient wrt h_t. Emits dC = dh*z*(1-c^2), dGz = dh*(c-hp)*z*(1-z),
// and accumulates dh_prev += dh*(1-z).   (h_t = z*c + (1-z)*hp)
__global__ __launch_bounds__(256) void gru_state_bwd_kernel(const GruStateBwdArgs g) {
    const GruStateBwdChain& c = g.chain[blockIdx.y];
    for (int m = blockIdx.x; m < g.B; m += gridDim.x) gru_state_bwd_row(c, m, g.H, threadIdx.x, 256);
}

// out[n] (+)= sum_m x[m*ld + n]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long long M, int N, int ld,
                                                     float* __restrict__ out, int accumulate) {
    // block handles 64 columns; 4 row-groups of 64 lanes
    __shared__ float red[4][64];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rg = threadIdx.x >> 6;
    float acc = 0.f;
    if (n < N) {
        const long long per = (M + gridDim.y - 1) / gridDim.y;
        const long long mb = (long long)blockIdx.y * per, me = min(M, mb + per);
        for (long long m = mb + rg; m < me; m += 4) acc += x[m * ld + n];
    }
    red[rg][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rg == 0 && n < N) {
        const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (gridDim.y > 1) out[(size_t)blockIdx.y * N + n] = s;  // `out` is the partial buffer [ysplit, N] here
        else if (accumulate) out[n] += s;
        else out[n] = s;
    }
}

// The same sums for 16-byte aligned rows with N % 4 == 0: a lane owns 4 adjacent columns (one 1 KB wave load per row
// instead of a 256-byte one) and keeps 8 rows in flight; block = 256 columns x one row slice, the 4 waves take rows
// rg, rg + 4, ...  (the stream of dG [T*B, 2H] for the bias gradients: 1.2 -> ~4 TB/s)
__global__ __launch_bounds__(256) void colsum4_kernel(const float* __restrict__ x, long long M, int N, int ld,
                                                      float* __restrict__ out, int accumulate) {
    __shared__ f32x4 red[4][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int n = blockIdx.x * 256 + 4 * lane;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
        const long long per = (M + gridDim.y - 1) / gridDim.y;
        const long long mb = (long long)blockIdx.y * per, me = min(M, mb + per);
        const float* p = x + n;
        long long m = mb + rg;
        for (; m + 28 < me; m += 32) {
            f32x4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const f32x4*>(p + (m + 4 * q) * ld);
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += v[q];
        }
        for (; m < me; m += 4) acc += *reinterpret_cast<const f32x4*>(p + m * ld);
    }
    red[rg][lane] = acc;
    __syncthreads();
    if (rg == 0 && n < N) {
        const f32x4 s = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
        float* o = gridDim.y > 1 ? out + (size_t)blockIdx.y * N + n : out + n;  // partial buffer [ysplit, N] or the result
        f32x4 r = s;
        if (gridDim.y == 1 && accumulate) r += *reinterpret_cast<const f32x4*>(o);
        *reinterpret_cast<f32x4*>(o) = r;
    }
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, size_t n, float* __restrict__ out) {
    __shared__ float red[4];
    float acc = 0.f;
    const size_t n4 = n / 4;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 v = x4[i];
        acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) acc += x[i] * x[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];  // per-block partial (no atomics)
}

// out[0] = sum of n partials, fixed order: the global norm (and with it every clipped update) is reproducible.
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += part[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = red[0] + red[1] + red[2] + red[3];
}

// out[n] (+)= sum over the row slices' partial column sums, slice order
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ part, int N, int ysplit,
                                                            float* __restrict__ out, int accumulate) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float v = accumulate ? out[n] : 0.f;
    // (eight partials requested per round; the sum keeps the slice order.  One load per dependent add, as the plain loop
    // compiled, made this 8-block kernel a chain of `ysplit` memory round trips: 20 us for 2048 columns.)
    int y = 0;
    for (; y + 8 <= ysplit; y += 8) {
        float p[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) p[q] = part[(size_t)(y + q) * N + n];
#pragma unroll
        for (int q = 0; q < 8; ++q) v += p[q];
    }
    for (; y < ysplit; ++y) v += part[(size_t)y * N + n];
    out[n] = v;
}

// StepClipping(threshold) then Adam (Blocks 0.2 semantics):
//   g' = g * min(1, threshold / ||g||)      (global norm over all parameters, gnorm_sq = ||g||^2)
//   m = b1 m + (1-b1) g' ;  v = b2 v + (1-b2) g'^2
//   lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t) ;  p -= lr_t * m / (sqrt(v) + eps)
__global__ __launch_bounds__(256) void adam_clip_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, size_t n,
                                                        const float* __restrict__ gnorm_sq, float grad_scale,
                                                        float threshold, float lr_t, float b1, float b2, float eps) {
    float scale = grad_scale;
    if (threshold > 0.f) {
        const float nrm = sqrtf(*gnorm_sq) * grad_scale;
        if (nrm > threshold) scale *= threshold / nrm;
        // Non-finite gradient norm: skip the step entirely -- no write to p, m or v (g[i] * 0 would still be NaN and
        // poison the moments).  Blocks' StepClipping has no such guard (the NaN reaches the parameters and the
        // LearningRateSchedule reloads the best checkpoint, extensions.py:126-140); skipping keeps the replicas finite
        // and the schedule still sees the non-finite validation cost.
        if (!(nrm == nrm) || nrm > 3.0e38f) return;
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gi = g[i] * scale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

// One workgroup per batch row: pi = softmax(co_hat (1 + bias)) + eps; component = first k with
// cumsum(pi) > u (Theano multinomial); x[o] = mu[o,k] + (exp(sig_hat[o,k] - bias) + eps) * noise[o].
__global__ __launch_bounds__(64) void gmm_sample_kernel(const float* __restrict__ mu, const float* __restrict__ sig_hat,
                                                        const float* __restrict__ co_hat, int O, int K, float bias,
                                                        float eps, const float* __restrict__ unif,
                                                        const float* __restrict__ noise, float* __restrict__ x, int ldx,
                                                        float* __restrict__ pi_out) {
    __shared__ int s_pick;
    const int b = blockIdx.x, t = threadIdx.x;
    if (t == 0) {
        const float* c = co_hat + (size_t)b * K;
        const float sc = 1.f + bias;
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, c[k] * sc);
        float tot = 0.f;
        for (int k = 0; k < K; ++k) tot += expf(c[k] * sc - mx);
        const float u = unif[b];
        float cum = 0.f;
        int pick = K - 1;
        bool found = false;
        for (int k = 0; k < K; ++k) {
            const float pk = expf(c[k] * sc - mx) / tot + eps;
            if (pi_out) pi_out[(size_t)b * K + k] = pk;
            cum += pk;
            if (!found && cum > u) { pick = k; found = true; }
        }
        s_pick = pick;
    }
    __syncthreads();
    const int k = s_pick;
    for (int o = t; o < O; o += 64) {
        const size_t i = (size_t)b * O * K + (size_t)o * K + k;
        x[(size_t)b * ldx + o] = mu[i] + (expf(sig_hat[i] - bias) + eps) * noise[(size_t)b * O + o];
    }
}

__global__ __launch_bounds__(256) void lstm_state_bwd_kernel(const float* __restrict__ dh,
                                                             const float* __restrict__ dh2, float* __restrict__ dc,
                                                             const float* __restrict__ gates,
                                                             const float* __restrict__ c_prev,
                                                             const float* __restrict__ c_new, float* __restrict__ dP,
                                                             int B, int H) {
    const size_t n = (size_t)B * H;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (size_t)gridDim.x * 256) {
        const int m = (int)(idx / H), k = (int)(idx % H);
        const float* g = gates + (size_t)m * 4 * H;
        const float gi = g[k], gf = g[H + k], go = g[2 * H + k], gg = g[3 * H + k];
        const float tc = tanhf(c_new[idx]);
        const float dhv = dh[idx] + (dh2 ? dh2[idx] : 0.f);
        const float dcv = dhv * go * (1.f - tc * tc) + dc[idx];
        float* o = dP + (size_t)m * 4 * H;
        o[k] = dcv * gg * gi * (1.f - gi);
        o[H + k] = dcv * c_prev[idx] * gf * (1.f - gf);
        o[2 * H + k] = dhv * tc * go * (1.f - go);
        o[3 * H + k] = dcv * gi * (1.f - gg * gg);
        dc[idx] = dcv * gf;
    }
}

// Sum over the 256 threads of a workgroup, result in every thread (red: 4 floats of LDS).
__device__ __forceinline__ float block_sum256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();  // protects red against the previous use
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// One workgroup per row.  Two passes over the row (mean, then centred second moment), like the oracle.
__global__ __launch_bounds__(256) void simple_norm_fwd_kernel(const float* x, int ldx, float* y, int ldy,
                                                              float* __restrict__ sigma, int N, float eps,
                                                              float* add_dst, int ld_add) {
    __shared__ float red[4];
    const size_t r = blockIdx.x;
    const float* xr = x + r * ldx;
    float s = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) s += xr[n];
    const float mean = block_sum256(s, red) / (float)N;
    float q = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) {
        const float d = xr[n] - mean;
        q += d * d;
    }
    const float sd = sqrtf(block_sum256(q, red) / (float)N);
    const float inv = 1.f / (eps + sd);
    for (int n = threadIdx.x; n < N; n += 256) {
        const float v = (xr[n] - mean) * inv;
        y[r * ldy + n] = v;
        if (add_dst) add_dst[r * ld_add + n] += v;
    }
    if (threadIdx.x == 0) sigma[r] = sd;
}

// y = (x - mu) / s, s = eps + sigma  =>  dx = (dy - mean(dy)) / s - y * sum(dy * y) / (N * sigma)
__global__ __launch_bounds__(256) void simple_norm_bwd_kernel(const float* dy, int lddy, const float* y, int ldy,
                                                              const float* __restrict__ sigma, float* dx, int lddx,
                                                              int N, float eps, int accumulate) {
    __shared__ float red[4];
    const size_t r = blockIdx.x;
    const float* dyr = dy + r * lddy;
    const float* yr = y + r * ldy;
    float a = 0.f, b = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) {
        a += dyr[n];
        b += dyr[n] * yr[n];
    }
    const float mdy = block_sum256(a, red) / (float)N;
    const float dot = block_sum256(b, red);
    const float sd = sigma[r];
    const float inv = 1.f / (eps + sd);
    const float k = sd > 0.f ? dot / ((float)N * sd) : 0.f;
    for (int n = threadIdx.x; n < N; n += 256) {
        const float v = (dyr[n] - mdy) * inv - yr[n] * k;
        if (accumulate) dx[r * lddx + n] += v;
        else dx[r * lddx + n] = v;
    }
}

struct NormSumArgs {
    NormSumGroup g[2];
};

__global__ __launch_bounds__(256) void norm_sum_kernel(const NormSumArgs a, float eps) {
    __shared__ float red[4];
    const NormSumGroup& g = a.g[blockIdx.y];
    const size_t r = blockIdx.x;
    const int N = g.N;
    float* dst = g.dst + r * N;
    for (int n = threadIdx.x; n < N; n += 256) dst[n] = g.base ? g.base[r * N + n] : 0.f;
    for (int s = 0; s < g.nsrc; ++s) {
        const float* xr = g.src[s] + r * N;
        float sum = 0.f;
        for (int n = threadIdx.x; n < N; n += 256) sum += xr[n];
        const float mean = block_sum256(sum, red) / (float)N;
        float q = 0.f;
        for (int n = threadIdx.x; n < N; n += 256) {
            const float dlt = xr[n] - mean;
            q += dlt * dlt;
        }
        const float inv = 1.f / (eps + sqrtf(block_sum256(q, red) / (float)N));
        for (int n = threadIdx.x; n < N; n += 256) dst[n] += (xr[n] - mean) * inv;  // same thread owns column n
    }
}

}  // namespace

// Library-owned scratch for the two-stage reductions, ONE PER STREAM (grown on demand; never from inside a captured scan
// plan): since round 4 the encoder's backward runs on a side stream beside the weight-gradient products of the main
// stream, and both take column sums -- a shared buffer would have the two finish passes read each other's partials.
static float* ew_scratch(size_t floats, hipStream_t stream) {
    struct Buf { float* p = nullptr; size_t cap = 0; };
    static std::mutex mu;
    static std::map<hipStream_t, Buf> bufs;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream, &cs);
    if (cs != hipStreamCaptureStatusNone) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    Buf& b = bufs[stream];
    if (floats > b.cap) {
        if (b.p) {
            (void)hipStreamSynchronize(stream);
            (void)hipFree(b.p);
            b.p = nullptr;
            b.cap = 0;
        }
        const size_t want = floats < (1u << 20) ? (1u << 20) : floats;
        if (hipMalloc(&b.p, want * sizeof(float)) != hipSuccess) return nullptr;
        b.cap = want;
    }
    return b.p;
}


int norm_sum_launch(const NormSumGroup* groups, int ngroups, int R, float eps, hipStream_t stream) {
    if (ngroups < 1 || ngroups > 2 || R < 1) return PH_ERR_BADARG;
    NormSumArgs a;
    for (int i = 0; i < ngroups; ++i) a.g[i] = groups[i];
    hipLaunchKernelGGL(norm_sum_kernel, dim3((unsigned)R, (unsigned)ngroups), dim3(256), 0, stream, a, eps);
    return (int)hipGetLastError();
}

int simple_norm_fwd_launch(const float* x, int ldx, float* y, int ldy, float* sigma, long long R, int N, float eps,
                           float* add_dst, int ld_add, hipStream_t stream) {
    if (R < 1 || N < 1) return PH_ERR_BADARG;
    hipLaunchKernelGGL(simple_norm_fwd_kernel, dim3((unsigned)R), dim3(256), 0, stream, x, ldx, y, ldy, sigma, N, eps,
                       add_dst, ld_add);
    return (int)hipGetLastError();
}

int simple_norm_bwd_launch(const float* dy, int lddy, const float* y, int ldy, const float* sigma, float* dx, int lddx,
                           long long R, int N, float eps, int accumulate, hipStream_t stream) {
    if (R < 1 || N < 1) return PH_ERR_BADARG;
    hipLaunchKernelGGL(simple_norm_bwd_kernel, dim3((unsigned)R), dim3(256), 0, stream, dy, lddy, y, ldy, sigma, dx,
                       lddx, N, eps, accumulate);
    return (int)hipGetLastError();
}


int lstm_state_bwd_launch(const float* dh, const float* dh2, float* dc, const float* gates, const float* c_prev,
                          const float* c_new, float* dP, int B, int H, hipStream_t stream) {
    const size_t n = (size_t)B * H;
    int bx = (int)((n + 255) / 256);
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(lstm_state_bwd_kernel, dim3(bx), dim3(256), 0, stream, dh, dh2, dc, gates, c_prev, c_new, dP, B,
                       H);
    return (int)hipGetLastError();
}

int gmm_sample_launch(const float* mu, const float* sig_hat, const float* co_hat, int B, int O, int K, float bias,
                      float eps, const float* unif, const float* noise, float* x, int ldx, float* pi_out,
                      hipStream_t stream) {
    hipLaunchKernelGGL(gmm_sample_kernel, dim3(B), dim3(64), 0, stream, mu, sig_hat, co_hat, O, K, bias, eps, unif, noise,
                       x, ldx, pi_out);
    return (int)hipGetLastError();
}

int gru_state_bwd_launch(const GruStateBwdArgs& g, hipStream_t stream) {
    if (g.nchain < 1 || g.nchain > 4) return PH_ERR_BADARG;
    const int bx = g.B < 1024 ? g.B : 1024;  // one row per block
    hipLaunchKernelGGL(gru_state_bwd_kernel, dim3(bx, g.nchain), dim3(256), 0, stream, g);
    return (int)hipGetLastError();
}

int colsum_launch(const float* x, long long M, int N, int ld, float* out, int accumulate, hipStream_t stream) {
    if (M < 0 || N < 1) return PH_ERR_BADARG;
    if (!(N & 3) && !(ld & 3) && !((uintptr_t)x & 15) && !((uintptr_t)out & 15) && M >= 64) {
        const int bx = ceil_div(N, 256);
        int ysplit = 1;
        // one workgroup per CU: more row slices only lengthen the serial finish pass (8 column blocks x 128 slices
        // measured 19 us + 33 us of finish for [51200, 2048]; 32 slices: the finish reads a quarter)
        while (bx * ysplit < 256 && M / (ysplit * 2) >= 64) ysplit *= 2;
        float* part = ysplit > 1 ? ew_scratch((size_t)ysplit * N, stream) : nullptr;
        if (ysplit == 1 || part) {
            hipLaunchKernelGGL(colsum4_kernel, dim3(bx, ysplit), dim3(256), 0, stream, x, M, N, ld, ysplit > 1 ? part : out,
                               accumulate);
            if (ysplit > 1)
                hipLaunchKernelGGL(colsum_finish_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, stream, part, N, ysplit, out,
                                   accumulate);
            return (int)hipGetLastError();
        }
    }
    int ysplit = 1;
    const int bx = ceil_div(N, 64);
    while (bx * ysplit < 512 && M / (ysplit * 2) >= 256) ysplit *= 2;
    if (ysplit > 1) {  // row slices -> partial sums -> fixed-order finish (deterministic, no float atomics)
        float* part = ew_scratch((size_t)ysplit * N, stream);
        if (!part) ysplit = 1;
        else {
            hipLaunchKernelGGL(colsum_kernel, dim3(bx, ysplit), dim3(256), 0, stream, x, M, N, ld, part, 0);
            hipLaunchKernelGGL(colsum_finish_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, stream, part, N, ysplit, out,
                               accumulate);
            return (int)hipGetLastError();
        }
    }
    hipLaunchKernelGGL(colsum_kernel, dim3(bx, ysplit), dim3(256), 0, stream, x, M, N, ld, out, accumulate);
    return (int)hipGetLastError();
}

int sumsq_launch(const float* x, size_t n, float* out, hipStream_t stream) {
    if (((uintptr_t)x & 15) != 0) return PH_ERR_BADARG;
    int bx = (int)((n / 4 + 255) / 256);
    if (bx < 1) bx = 1;
    if (bx > 2048) bx = 2048;
    float* part = ew_scratch((size_t)bx, stream);
    if (!part) return PH_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(sumsq_kernel, dim3(bx), dim3(256), 0, stream, x, n, part);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, stream, part, bx, out);
    return (int)hipGetLastError();
}

int adam_clip_launch(float* p, const float* g, float* m, float* v, size_t n, const float* gnorm_sq,
                     float grad_scale, float threshold, float lr_t, float b1, float b2, float eps,
                     hipStream_t stream) {
    int bx = (int)((n + 255) / 256);
    if (bx < 1) bx = 1;
    if (bx > 4096) bx = 4096;
    hipLaunchKernelGGL(adam_clip_kernel, dim3(bx), dim3(256), 0, stream, p, g, m, v, n, gnorm_sq, grad_scale,
                       threshold, lr_t, b1, b2, eps);
    return (int)hipGetLastError();
}
