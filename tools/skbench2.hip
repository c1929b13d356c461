// Design-space probe for the recurrent-step GEMM (development aid; timing only, values unchecked).
// out[64, N] = A[64, K] * W[K, N], M = 64 batch rows, f32 MFMA 16x16x4.
//   ALAY 0: A row-major [M][K]           ALAY 1: A transposed [K][64]
//   BLAY 0: W [K][N] (dword loads)       BLAY 1: W^T [N][K] (float4 along k)
//   BLAY 2: W pre-tiled [N/16][K/4][16][4] (one fully coalesced float4 load per 16-deep chunk)
//   ROWS: rows per workgroup (64 or 32)  NW: waves per workgroup (split-K)   PF: prefetch depth
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int ALAY, int BLAY, int ROWS>
struct Frag {
    f32x4 a[ROWS / 16];
    f32x4 b;
};

template <int ALAY, int BLAY, int ROWS>
__device__ __forceinline__ void fetch(Frag<ALAY, BLAY, ROWS>& f, const float* __restrict__ A, const float* __restrict__ W,
                                      int K, int N, int kc, int m0, int tile, int kk, int i) {
    constexpr int MB = ROWS / 16;
    const int k = kc + 4 * kk;
    if (ALAY == 0) {
#pragma unroll
        for (int rb = 0; rb < MB; ++rb) f.a[rb] = *reinterpret_cast<const f32x4*>(A + (size_t)(m0 + rb * 16 + i) * K + k);
    } else {
        // transposed [K][64]: lane (kk,i) takes rows {MB*i + q} for k = kc + 4kk + u  -> a[q][u]
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MB == 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(A + (size_t)(k + u) * 64 + m0 + 4 * i);
                f.a[0][u] = v[0]; f.a[1][u] = v[1]; f.a[2][u] = v[2]; f.a[3][u] = v[3];
            } else {
                const f32x2 v = *reinterpret_cast<const f32x2*>(A + (size_t)(k + u) * 64 + m0 + 2 * i);
                f.a[0][u] = v[0]; f.a[1][u] = v[1];
            }
        }
    }
    const int n = tile * 16 + i;
    if (BLAY == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) f.b[u] = W[(size_t)(k + u) * N + n];
    } else if (BLAY == 1) {
        f.b = *reinterpret_cast<const f32x4*>(W + (size_t)n * K + k);
    } else {
        f.b = *reinterpret_cast<const f32x4*>(W + ((size_t)tile * (K / 4) + (k >> 2)) * 64 + i * 4);
    }
}

template <int ALAY, int BLAY, int ROWS, int NW, int PF, int ROT>
__global__ __launch_bounds__(NW * 64) void probe(const float* __restrict__ A, const float* __restrict__ W,
                                                 float* __restrict__ out, int K, int N) {
    constexpr int MB = ROWS / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* red = reinterpret_cast<f32x4*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kk = lane >> 4, i = lane & 15;
    const int tile = blockIdx.x, m0 = blockIdx.y * ROWS;
    f32x4 acc[MB];
#pragma unroll
    for (int rb = 0; rb < MB; ++rb) acc[rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nch = K / 16;
    const int rot = ROT ? (int)((blockIdx.x * 37u + blockIdx.y * 11u) % (unsigned)nch) : 0;
    auto kof = [&](int c) { int x = c + rot; if (x >= nch) x -= nch; return x * 16; };
    Frag<ALAY, BLAY, ROWS> ring[PF + 1];
    // Branch-free software pipeline: this wave owns chunks wave, wave+NW, ...; fetches beyond the
    // last chunk are clamped to it (harmless re-read), so the loop body is straight-line code and the
    // compiler can keep PF chunks in flight with counted s_waitcnt vmcnt(N).
    const int mine = (nch - wave + NW - 1) / NW;       // chunks of this wave (>= 0)
    const int lastc = wave + (mine - 1) * NW;
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        const int c = min(wave + p * NW, lastc);
        fetch<ALAY, BLAY, ROWS>(ring[p], A, W, K, N, kof(c), m0, tile, kk, i);
    }
    const int rounds = (mine + PF) / (PF + 1);
    int cbase = wave;
    for (int rd = 0; rd < rounds; ++rd) {
#pragma unroll
        for (int p = 0; p <= PF; ++p) {
            const int c = cbase + p * NW;
            const int cn = min(c + PF * NW, lastc);
            fetch<ALAY, BLAY, ROWS>(ring[(p + PF) % (PF + 1)], A, W, K, N, kof(cn), m0, tile, kk, i);
            if (c <= lastc) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int rb = 0; rb < MB; ++rb)
                        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[p].a[rb][u], ring[p].b[u], acc[rb], 0, 0, 0);
            }
        }
        cbase += (PF + 1) * NW;
    }
#pragma unroll
    for (int rb = 0; rb < MB; ++rb) red[(wave * MB + rb) * 64 + lane] = acc[rb];
    __syncthreads();
    if (tid >= MB * 64) return;
    const int rb = tid >> 6;
    f32x4 v = red[rb * 64 + lane];
#pragma unroll
    for (int w = 1; w < NW; ++w) v += red[(w * MB + rb) * 64 + lane];
    const int g = lane >> 4, jj = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(size_t)(m0 + rb * 16 + 4 * g + r) * N + tile * 16 + jj] = 1.f / (1.f + __expf(-v[r]));
}


// ---- probe2: cross-workgroup split-K with the A slice shared through LDS -------------------
// grid (N/BN, S): workgroup = NWC waves, tile 64 rows x BN = 16*NWC columns, K-slice KS = K/S.
// A slice [64][KS] is loaded once (coalesced, row-major source) into LDS; each wave streams the
// weights of its 16 columns.  Partial tile is written to slab[s][64][N] (no reduction here).
template <int BLAY, int NWC>
__global__ __launch_bounds__(NWC * 64) void probe2(const float* __restrict__ A, const float* __restrict__ W,
                                                   float* __restrict__ slab, int K, int N, int KS) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);
    const int P = KS + 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kk = lane >> 4, i = lane & 15;
    const int k0 = blockIdx.y * KS;
    const int tile = blockIdx.x * NWC + wave;
    // stage A: 64 rows x KS floats, float4 per thread, rows contiguous in global
    const int v4_per_row = KS / 4;
    for (int idx = tid; idx < 64 * v4_per_row; idx += NWC * 64) {
        const int m = idx / v4_per_row, q = idx % v4_per_row;
        *reinterpret_cast<f32x4*>(As + m * P + 4 * q) = *reinterpret_cast<const f32x4*>(A + (size_t)m * K + k0 + 4 * q);
    }
    __syncthreads();
    f32x4 acc[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) acc[rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nch = KS / 16;
    f32x4 bcur, bnxt;
    auto loadb = [&](int c) -> f32x4 {
        const int k = k0 + c * 16 + 4 * kk;
        f32x4 b;
        if (BLAY == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) b[u] = W[(size_t)(k + u) * N + tile * 16 + i];
        } else {
            b = *reinterpret_cast<const f32x4*>(W + ((size_t)tile * (K / 4) + (k >> 2)) * 64 + i * 4);
        }
        return b;
    };
    bcur = loadb(0);
    for (int c = 0; c < nch; ++c) {
        if (c + 1 < nch) bnxt = loadb(c + 1);
        f32x4 a[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) a[rb] = *reinterpret_cast<const f32x4*>(As + (rb * 16 + i) * P + c * 16 + 4 * kk);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb][u], bcur[u], acc[rb], 0, 0, 0);
        bcur = bnxt;
    }
    const int g = lane >> 4, jj = lane & 15;
    float* o = slab + (size_t)blockIdx.y * 64 * N;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[(size_t)(rb * 16 + 4 * g + r) * N + tile * 16 + jj] = acc[rb][r];
}

template <int BLAY, int NWC>
void run2(int K, int N, int S, float* A, std::vector<float*>& W, float* slab, hipStream_t st) {
    const int iters = 300, NS = (int)W.size();
    const int KS = K / S;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    dim3 grid(N / (16 * NWC), S), block(NWC * 64);
    const size_t lds = (size_t)64 * (KS + 4) * 4;
    CK(hipFuncSetAttribute((const void*)probe2<BLAY, NWC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int it = 0; it < 20; ++it)
        hipLaunchKernelGGL((probe2<BLAY, NWC>), grid, block, lds, st, A, W[it % NS], slab, K, N, KS);
    CK(hipStreamSynchronize(st));
    CK(hipGetLastError());
    CK(hipEventRecord(e0, st));
    for (int it = 0; it < iters; ++it)
        hipLaunchKernelGGL((probe2<BLAY, NWC>), grid, block, lds, st, A, W[it % NS], slab, K, N, KS);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / iters;
    printf("probe2     B%d nwc%d S=%d (KS=%d, %d WGs) K=%d N=%d: %7.2f us  %6.1f TF  W %5.2f TB/s\n", BLAY, NWC, S, KS,
           (N / (16 * NWC)) * S, K, N, us, 2.0 * 64 * K * N / us * 1e-6, 4.0 * K * N / us * 1e-6);
}

// ---- pure weight streaming calibration -------------------------------------------------------
__global__ __launch_bounds__(256) void stream_read(const f32x4* __restrict__ W, size_t n4, float* __restrict__ out) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n4; idx += (size_t)gridDim.x * 256) acc += W[idx];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = 1.f;
}
void run_stream(size_t nfloat, std::vector<float*>& W, float* out, hipStream_t st, int blocks) {
    const int iters = 300, NS = (int)W.size();
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(stream_read, dim3(blocks), dim3(256), 0, st, (const f32x4*)W[it % NS], nfloat / 4, out);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(stream_read, dim3(blocks), dim3(256), 0, st, (const f32x4*)W[it % NS], nfloat / 4, out);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / iters;
    printf("stream_read %zu MB x%d sets, %d blocks: %7.2f us  %5.2f TB/s\n", nfloat * 4 >> 20, NS, blocks, us, nfloat * 4.0 / us * 1e-6);
}

static float* dalloc(size_t n) {
    float* p;
    CK(hipMalloc(&p, n * sizeof(float)));
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = 0.01f * (float)((i * 2654435761u) % 1000) / 1000.f - 0.005f;
    CK(hipMemcpy(p, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    return p;
}

template <int ALAY, int BLAY, int ROWS, int NW, int PF, int ROT>
void run(const char* tag, int K, int N, float* A, std::vector<float*>& W, float* out, hipStream_t st) {
    const int iters = 300, NS = (int)W.size();
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    dim3 grid(N / 16, 64 / ROWS), block(NW * 64);
    const size_t lds = (size_t)NW * (ROWS / 16) * 64 * 16;
    for (int it = 0; it < 20; ++it)
        hipLaunchKernelGGL((probe<ALAY, BLAY, ROWS, NW, PF, ROT>), grid, block, lds, st, A, W[it % NS], out, K, N);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int it = 0; it < iters; ++it)
        hipLaunchKernelGGL((probe<ALAY, BLAY, ROWS, NW, PF, ROT>), grid, block, lds, st, A, W[it % NS], out, K, N);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / iters;
    printf("%-10s A%d B%d rows%d nw%-2d pf%d rot%d K=%d N=%d: %7.2f us  %6.1f TF\n", tag, ALAY, BLAY, ROWS, NW, PF, ROT, K, N, us,
           2.0 * 64 * K * N / us * 1e-6);
}

int main() {
    const int K = 2304, N = 2048, NS = 6;
    float* A = dalloc((size_t)64 * K);
    std::vector<float*> W(NS);
    for (int s = 0; s < NS; ++s) W[s] = dalloc((size_t)K * N);
    float* out = dalloc((size_t)64 * N);
    hipStream_t st;
    CK(hipStreamCreate(&st));
#define R(a, b, rows, nw, pf, rot) run<a, b, rows, nw, pf, rot>("gatesL2", K, N, A, W, out, st)
    R(0, 0, 32, 8, 1, 0);
    R(0, 0, 32, 8, 2, 0);
    R(0, 0, 32, 8, 3, 0);
    R(0, 0, 64, 8, 1, 0);
    R(0, 0, 64, 8, 2, 0);
    R(0, 0, 64, 8, 3, 0);
    R(0, 2, 32, 8, 2, 0);
    R(0, 2, 64, 8, 2, 0);
    R(1, 2, 32, 8, 2, 0);
    R(1, 2, 64, 8, 2, 0);
    R(1, 2, 64, 8, 3, 0);
    R(1, 2, 64, 16, 2, 0);
    float* slab = dalloc((size_t)16 * 64 * N);
    run2<2, 8>(K, N, 16, A, W, slab, st);
    run_stream((size_t)K * N, W, out, st, 256);
    run_stream((size_t)K * N, W, out, st, 1024);
    run_stream((size_t)K * N, W, out, st, 2048);
    std::vector<float*> W1(1, W[0]);
    run_stream((size_t)K * N, W1, out, st, 1024);
    return 0;
}
