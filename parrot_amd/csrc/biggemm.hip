// LDS-tiled f32 GEMM on the CDNA4 f32-input matrix cores (v_mfma_f32_32x32x2_f32), gfx950 only.
//
// C[M,N] (+)= alpha * A(M,K) * B(K,N) (+ bias) with arbitrary "one stride is 1" operand layouts, so
// the same kernel serves NN (x*W), NT (dy*W^T) and TN (x^T*dy, the deferred weight gradients of
// the decoder scan) without transposed copies in HBM.
//
// Tiling: 128x128x16 per 256-thread workgroup (4 waves as 2x2, each 64x64 = 2x2 MFMA 32x32
// blocks, 64 accumulator VGPRs).  Operands are staged global -> VGPR -> LDS in [k][m] / [k][n]
// images (row pitch 132 floats) so that an MFMA fragment read is 32 consecutive floats per half
// wave (conflict-free ds_read_b32); the next K-tile is prefetched into registers while the
// current one is consumed (double-buffered LDS, one barrier per K-tile).  Workgroup ids are
// remapped so that each XCD (private 4 MiB L2) works on 8 x 8 blocks of tiles (bg_tile_of_block).
#include "biggemm.h"

#include <atomic>

#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int BM = 128, BN = 128, BK = 16, PITCH = 132;

// Workgroup -> output tile.  Block b runs on XCD b % 8 (private 4 MiB L2 each).  Every XCD gets a contiguous range
// of a tile order that walks column groups of 8 tiles row by row, so the ~64 workgroups an XCD runs at a time cover an
// 8 x 8 block of tiles: 16 operand panels feed 64 tiles (8-fold reuse out of L2).  A plain row-major band gave 1 + 48
// panels per 48 tiles on the 48-tile-wide weight gradients of a 3 x LSTM-1536 decoder, i.e. every tile row re-read the
// whole [T*B, 4H] gradient matrix from HBM.
__device__ __forceinline__ void bg_tile_of_block(int bid, int tiles_m, int tiles_n, int& tm, int& tn) {
    const int nwg = tiles_m * tiles_n;
    const int q = nwg / 8, r = nwg % 8;
    const int xcd = bid % 8, idx = bid / 8;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    constexpr int GW = 8;
    const int group = lin / (GW * tiles_m);
    const int first = group * GW;
    const int gsz = min(tiles_n - first, GW);
    const int rem = lin - group * GW * tiles_m;
    tm = rem / gsz;
    tn = first + rem % gsz;
}

// Loads the 128 x 16 slab of an operand whose element (x, k) sits at p[x*sx + k*sk].
// XC = true : x is the contiguous index (sx == 1).  thread -> (xq = t&31, kr = t>>5), 2 passes.
// XC = false: k is the contiguous index (sk == 1).  thread -> (x = t>>1, kq = t&1), 2 vectors.
template <bool XC>
__device__ __forceinline__ void bg_load(const float* __restrict__ p, int x0, int X, int k0, int kend,
                                        long long sx, long long sk, bool vec, int t, f32x4 (&v)[2]) {
    if (XC) {
        const int xq = t & 31, kr = t >> 5;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int k = k0 + kr + 8 * ps;
            const int x = x0 + 4 * xq;
            v[ps] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (k < kend) {
                const float* q = p + (long long)k * sk + x;
                if (vec && x + 3 < X) {
                    v[ps] = *reinterpret_cast<const f32x4*>(q);
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (x + u < X) v[ps][u] = q[u];
                }
            }
        }
    } else {
        const int x = x0 + (t >> 1), kq = t & 1;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int k = k0 + 8 * kq + 4 * ps;
            v[ps] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (x < X) {
                const float* q = p + (long long)x * sx + k;
                if (vec && k + 3 < kend) {
                    v[ps] = *reinterpret_cast<const f32x4*>(q);
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (k + u < kend) v[ps][u] = q[u];
                }
            }
        }
    }
}

template <bool XC>
__device__ __forceinline__ void bg_store(float* __restrict__ s, int t, const f32x4 (&v)[2]) {
    if (XC) {
        const int xq = t & 31, kr = t >> 5;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
            *reinterpret_cast<f32x4*>(s + (kr + 8 * ps) * PITCH + 4 * xq) = v[ps];
    } else {
        const int x = t >> 1, kq = t & 1;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
#pragma unroll
            for (int u = 0; u < 4; ++u) s[(8 * kq + 4 * ps + u) * PITCH + x] = v[ps][u];
    }
}

template <bool AXC, bool BXC>
__global__ __launch_bounds__(256) void bg_kernel(const BgArgs a, int vecA, int vecB, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) float As[2][BK * PITCH];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * PITCH];

    int tm, tn;
    bg_tile_of_block(blockIdx.x, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    const int z = blockIdx.y;
    const int batch = z / a.splitk, ks = z % a.splitk;
    int kchunk = (a.K + a.splitk - 1) / a.splitk;
    kchunk = (kchunk + BK - 1) / BK * BK;
    const int kbeg = ks * kchunk;
    const int kend = min(a.K, kbeg + kchunk);

    const float* A = a.A + (long long)batch * a.batchA;
    const float* B = a.B + (long long)batch * a.batchB;
    float* C = a.C + (long long)batch * a.batchC;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int kk = lane >> 5, li = lane & 31;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    f32x4 ra[2], rb[2];
    const int nk = (kend - kbeg + BK - 1) / BK;
    if (nk > 0) {
        bg_load<AXC>(A, m0, a.M, kbeg, kend, a.sam, a.sak, vecA, t, ra);
        bg_load<BXC>(B, n0, a.N, kbeg, kend, a.sbn, a.sbk, vecB, t, rb);
        bg_store<AXC>(As[0], t, ra);
        bg_store<BXC>(Bs[0], t, rb);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            bg_load<AXC>(A, m0, a.M, kbeg + (kt + 1) * BK, kend, a.sam, a.sak, vecA, t, ra);
            bg_load<BXC>(B, n0, a.N, kbeg + (kt + 1) * BK, kend, a.sbn, a.sbk, vecB, t, rb);
        }
        const float* as = As[cur] + wm * 64 + li;
        const float* bs = Bs[cur] + wn * 64 + li;
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp) {
            const int row = (2 * kp + kk) * PITCH;
            const float a0 = as[row], a1 = as[row + 32];
            const float b0 = bs[row], b1 = bs[row + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            bg_store<AXC>(As[cur ^ 1], t, ra);
            bg_store<BXC>(Bs[cur ^ 1], t, rb);
        }
        __syncthreads();
    }

    // Epilogue.  32x32 C/D layout: col = lane & 31, row = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5).
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + li;
            if (n >= a.N) continue;
            const float bias = (a.bias && ks == 0) ? a.bias[n] : 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int m = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * kk;
                if (m >= a.M) continue;
                float v = a.alpha * acc[i][j][q] + bias;
                float* c = C + (long long)m * a.ldc + n;
                if (a.splitk > 1) {
                    a.ws[((long long)z * a.M + m) * a.N + n] = a.alpha * acc[i][j][q];  // bias added by the reducer
                } else {
                    if (a.accumulate) v += *c;
                    if (a.act == 1) v = fmaxf(v, 0.f);
                    else if (a.act == 2) v = tanhf(v);
                    else if (a.act == 3) v = 1.f / (1.f + expf(-v));
                    if (a.gate && !(a.gate[(long long)m * a.ldg + n] > 0.f)) v = 0.f;
                    *c = v;
                }
            }
        }
}


// ---- 8-wave variant (512 threads, same 128x128x16 tile): waves as 2 x 4, each 64 x 32 (2 MFMA blocks, 32
// accumulator VGPRs), twice the waves per CU for the same LDS: the kernel that runs (the 4-wave
// kernel above is kept for reference).  PMC on the 4-wave kernel: MFMA pipe 56 % busy, waves parked at s_waitcnt/barriers 45 % of
// their cycles, no LDS bank conflicts -- more resident waves hide those waits.
template <bool XC>
__device__ __forceinline__ void bg_load8(const float* __restrict__ p, int x0, int X, int k0, int kend,
                                         long long sx, long long sk, bool vec, int t, f32x4& v) {
    v = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (XC) {
        const int xq = t & 31, kr = t >> 5;
        const int k = k0 + kr, x = x0 + 4 * xq;
        if (k < kend) {
            const float* q = p + (long long)k * sk + x;
            if (vec && x + 3 < X) v = *reinterpret_cast<const f32x4*>(q);
            else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (x + u < X) v[u] = q[u];
            }
        }
    } else {
        const int x = x0 + (t >> 2), k = k0 + 4 * (t & 3);
        if (x < X) {
            const float* q = p + (long long)x * sx + k;
            if (vec && k + 3 < kend) v = *reinterpret_cast<const f32x4*>(q);
            else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (k + u < kend) v[u] = q[u];
            }
        }
    }
}

template <bool XC>
__device__ __forceinline__ void bg_store8(float* __restrict__ s, int t, const f32x4& v) {
    if (XC) {
        *reinterpret_cast<f32x4*>(s + (t >> 5) * PITCH + 4 * (t & 31)) = v;
    } else {
        const int x = t >> 2, kq = t & 3;
#pragma unroll
        for (int u = 0; u < 4; ++u) s[(4 * kq + u) * PITCH + x] = v[u];
    }
}

template <bool AXC, bool BXC>
__global__ __launch_bounds__(512) void bg_kernel8(const BgArgs a, int vecA, int vecB, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) float As[2][BK * PITCH];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * PITCH];
    int tm, tn;
    bg_tile_of_block(blockIdx.x, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = blockIdx.y;
    const int batch = z / a.splitk, ks = z % a.splitk;
    int kchunk = (a.K + a.splitk - 1) / a.splitk;
    kchunk = (kchunk + BK - 1) / BK * BK;
    const int kbeg = ks * kchunk;
    const int kend = min(a.K, kbeg + kchunk);
    const float* A = a.A + (long long)batch * a.batchA;
    const float* B = a.B + (long long)batch * a.batchB;
    float* C = a.C + (long long)batch * a.batchC;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int kk = lane >> 5, li = lane & 31;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    f32x4 ra, rb;
    const int nk = (kend - kbeg + BK - 1) / BK;
    if (nk > 0) {
        bg_load8<AXC>(A, m0, a.M, kbeg, kend, a.sam, a.sak, vecA, t, ra);
        bg_load8<BXC>(B, n0, a.N, kbeg, kend, a.sbn, a.sbk, vecB, t, rb);
        bg_store8<AXC>(As[0], t, ra);
        bg_store8<BXC>(Bs[0], t, rb);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            bg_load8<AXC>(A, m0, a.M, kbeg + (kt + 1) * BK, kend, a.sam, a.sak, vecA, t, ra);
            bg_load8<BXC>(B, n0, a.N, kbeg + (kt + 1) * BK, kend, a.sbn, a.sbk, vecB, t, rb);
        }
        const float* as = As[cur] + wm * 64 + li;
        const float* bs = Bs[cur] + wn * 32 + li;
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp) {
            const int row = (2 * kp + kk) * PITCH;
            const float a0 = as[row], a1 = as[row + 32];
            const float b0 = bs[row];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            bg_store8<AXC>(As[cur ^ 1], t, ra);
            bg_store8<BXC>(Bs[cur ^ 1], t, rb);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int n = n0 + wn * 32 + li;
        if (n >= a.N) continue;
        const float bias = (a.bias && ks == 0) ? a.bias[n] : 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int m = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * kk;
            if (m >= a.M) continue;
            float v = a.alpha * acc[i][q] + bias;
            float* c = C + (long long)m * a.ldc + n;
            if (a.splitk > 1) {
                a.ws[((long long)z * a.M + m) * a.N + n] = a.alpha * acc[i][q];  // bias added by the reducer
            } else {
                if (a.accumulate) v += *c;
                if (a.act == 1) v = fmaxf(v, 0.f);
                else if (a.act == 2) v = tanhf(v);
                else if (a.act == 3) v = 1.f / (1.f + expf(-v));
                if (a.gate && !(a.gate[(long long)m * a.ldg + n] > 0.f)) v = 0.f;
                *c = v;
            }
        }
    }
}

// ---- macro tiles tried in round 4 and dropped (profiles/r04_gemm_tile_variants.txt; the kernels live in commit
// "bf16-in weight-gradient GEMM ..." of the history): the same loop on 128 x 128 x 32, 256 x 128 x 16 / 32,
// 128 x 256 x 16 / 32 and 256 x 256 x 16 tiles (64 x 64 and 64 x 128 wave tiles, fewer LDS reads and barriers per MFMA)
// ran at 86 / 94 / 76 / 97 / 75 / 77 TFLOP/s on the cfg2 products where this kernel runs at 102: what the bigger tiles
// save per MFMA they lose in resident waves (this kernel: 65 VGPRs and 33 KB of LDS per workgroup = four workgroups,
// 32 waves per CU).  And the roof is lower than the data-sheet figure: under this kernel the chip sustains 2.09 GHz
// (GRBM_GUI_ACTIVE / duration, profiles/r04_gemm_clock.txt), where 256 CUs x 4 SIMDs x 64 flop/clk = 136.9 TFLOP/s,
// not the 157.3 of 2.4 GHz: 102-112 TFLOP/s is 75-82 % of what the clock allows (matrix pipe 80 % busy by the SQ counters).

// ---- bf16-operand variant (BgArgs::bf16): same 128 x 128 output tile and 2 x 4 wave grid, K-tile 32 ------------
// The operands stay f32 in HBM (they are the scan's saved activations / gradients and the f32 master weights); each
// thread rounds the 8 values it fetched to bf16 and writes them with one ds_write_b128 into an [x][k] image (row pitch
// 40 bf16 = 80 B: the 16 lanes an LDS cycle serves hit 16 disjoint 4-bank groups, for the writes and for the
// fragment reads).  A wave reads its MFMA fragments as ds_read_b128 (row li, k = 16s + 8kk .. +7) and issues
// 4 v_mfma_f32_32x32x16_bf16 per K-tile (32 cycles each) where the f32 kernel issues 32 v_mfma_f32_32x32x2_f32
// (64 cycles each) for the same K range: the matrix pipe stops being the limit and the kernel becomes bound by
// the L2 -> LDS operand traffic (32 KB of f32 per K-tile and workgroup).
constexpr int BK16 = 32, PITCH16 = 40, PITCHT = 144;  // row pitches in bf16: [x][k] image / [k][x] image

// k-contiguous operand (element (x,k) at p[x*sx + k]): thread -> (x = t >> 2, kg = t & 3), two 16-byte loads, one
// ds_write_b128 into an [x][k] image; MFMA fragments are plain ds_read_b128.
// x-contiguous operand (element (x,k) at p[k*sk + x], e.g. both operands of the weight gradients X^T . dG): thread ->
// (xq = t & 31, k = t >> 5 and + 16): two 16-byte loads of 4 consecutive x (a wave instruction = two whole 512-byte
// rows), one ds_write_b64 each into a [k][x] image (pitch 288 B).  The MFMA operand wants 8 consecutive k per lane:
// ds_read_b64_tr_b16 (gfx950) hands lane i of a 16-lane group the i-th column of the 4 x 16 block the group's lanes
// point at (mapping measured with tools/tr_probe.hip), so two of them give k0 .. k0+7 of row x0 + i without any
// transposition in registers.  The first version fetched these operands with eight 4-byte loads per thread.
template <bool XC>
__device__ __forceinline__ void bg_load16(const float* __restrict__ p, int x0, int X, int k0, int kend,
                                          long long sx, long long sk, bool vec, int t, f32x4 (&v)[2]) {
    v[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    v[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (XC) {
        const int x = x0 + 4 * (t & 31);
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int k = k0 + (t >> 5) + 16 * ps;
            if (k < kend) {
                const float* q = p + (long long)k * sk + x;
                if (vec && x + 3 < X) {
                    v[ps] = *reinterpret_cast<const f32x4*>(q);
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (x + u < X) v[ps][u] = q[u];
                }
            }
        }
    } else {
        const int x = x0 + (t >> 2), k = k0 + 8 * (t & 3);
        if (x < X) {
            const float* q = p + (long long)x * sx + k;
            if (vec && k + 7 < kend) {
                v[0] = *reinterpret_cast<const f32x4*>(q);
                v[1] = *reinterpret_cast<const f32x4*>(q + 4);
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (k + u < kend) v[u >> 2][u & 3] = q[u];
            }
        }
    }
}

template <bool XC>
__device__ __forceinline__ void bg_store16(__bf16* __restrict__ s, int t, const f32x4 (&v)[2]) {
    if (XC) {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
            *reinterpret_cast<bf16x4*>(s + ((t >> 5) + 16 * ps) * PITCHT + 4 * (t & 31)) = __builtin_convertvector(v[ps], bf16x4);
    } else {
        *reinterpret_cast<bf16x8*>(s + (t >> 2) * PITCH16 + 8 * (t & 3)) = ph_bf16x8(v[0], v[1]);
    }
}

// MFMA 32x32x16 operand of lane (kk = lane >> 5, li = lane & 31): row xb + li, k = 16 s + 8 kk .. +7
template <bool XC>
__device__ __forceinline__ bf16x8 bg_frag16(const __bf16* __restrict__ img, int xb, int s, int lane) {
    const int kk = lane >> 5, li = lane & 31;
    if (XC) {
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) s16x4* lds4;
        // 16-lane group: rows (k) k0 + q, q = (li >> 2) & 3; 4 lanes per row, 4 columns (x) each
        const int k0 = 16 * s + 8 * kk, q = (li >> 2) & 3, xg = xb + (li & 16) + 4 * (li & 3);
        const __bf16* p = img + (k0 + q) * PITCHT + xg;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(p));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(p + 4 * PITCHT));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    } else {
        return *reinterpret_cast<const bf16x8*>(img + (xb + li) * PITCH16 + 16 * s + 8 * kk);
    }
}

template <bool AXC, bool BXC>
__global__ __launch_bounds__(512) void bg_kernel_bf16(const BgArgs a, int vecA, int vecB, int tiles_m, int tiles_n) {
    constexpr int ASZ = AXC ? BK16 * PITCHT : BM * PITCH16, BSZ = BXC ? BK16 * PITCHT : BN * PITCH16;
    __shared__ __attribute__((aligned(16))) __bf16 As[2][ASZ];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[2][BSZ];
    int tm, tn;
    bg_tile_of_block(blockIdx.x, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = blockIdx.y;
    const int batch = z / a.splitk, ks = z % a.splitk;
    int kchunk = (a.K + a.splitk - 1) / a.splitk;
    kchunk = (kchunk + BK16 - 1) / BK16 * BK16;
    const int kbeg = ks * kchunk;
    const int kend = min(a.K, kbeg + kchunk);
    const float* A = a.A + (long long)batch * a.batchA;
    const float* B = a.B + (long long)batch * a.batchB;
    float* C = a.C + (long long)batch * a.batchC;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int kk = lane >> 5, li = lane & 31;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    f32x4 ra[2], rb[2];
    const int nk = (kend - kbeg + BK16 - 1) / BK16;
    if (nk > 0) {
        bg_load16<AXC>(A, m0, a.M, kbeg, kend, a.sam, a.sak, vecA, t, ra);
        bg_load16<BXC>(B, n0, a.N, kbeg, kend, a.sbn, a.sbk, vecB, t, rb);
        bg_store16<AXC>(As[0], t, ra);
        bg_store16<BXC>(Bs[0], t, rb);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            bg_load16<AXC>(A, m0, a.M, kbeg + (kt + 1) * BK16, kend, a.sam, a.sak, vecA, t, ra);
            bg_load16<BXC>(B, n0, a.N, kbeg + (kt + 1) * BK16, kend, a.sbn, a.sbk, vecB, t, rb);
        }
#pragma unroll
        for (int s = 0; s < BK16 / 16; ++s) {
            const bf16x8 a0 = bg_frag16<AXC>(As[cur], wm * 64, s, lane);
            const bf16x8 a1 = bg_frag16<AXC>(As[cur], wm * 64 + 32, s, lane);
            const bf16x8 b0 = bg_frag16<BXC>(Bs[cur], wn * 32, s, lane);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            bg_store16<AXC>(As[cur ^ 1], t, ra);
            bg_store16<BXC>(Bs[cur ^ 1], t, rb);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int n = n0 + wn * 32 + li;
        if (n >= a.N) continue;
        const float bias = (a.bias && ks == 0) ? a.bias[n] : 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int m = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * kk;
            if (m >= a.M) continue;
            float v = a.alpha * acc[i][q] + bias;
            float* c = C + (long long)m * a.ldc + n;
            if (a.splitk > 1) {
                a.ws[((long long)z * a.M + m) * a.N + n] = a.alpha * acc[i][q];  // bias added by the reducer
            } else {
                if (a.accumulate) v += *c;
                if (a.act == 1) v = fmaxf(v, 0.f);
                else if (a.act == 2) v = tanhf(v);
                else if (a.act == 3) v = 1.f / (1.f + expf(-v));
                if (a.gate && !(a.gate[(long long)m * a.ldg + n] > 0.f)) v = 0.f;
                *c = v;
            }
        }
    }
}

// ---- bf16-IN variant (BgArgs::bf16 == 2, round 4): both operands ARE bf16 in HBM, x-contiguous (TN products) ----------
// The deferred weight gradients of a bf16-operand decoder (X^T . dG over K = T*B rows, 6.8 TFLOP per cfg4 window) ran
// at ~11 % of the bf16 MFMA peak on bg_kernel_bf16: f32 operands (4 B per element through L2 and the staging
// registers), a 64 x 32 wave tile (1.5 KB of LDS fragment reads per MFMA: LDS-bound at twice the MFMA time).  Here the
// operands are bf16 copies written once per window (parrot_to_bf16; half the bytes, no conversion in the loop), the
// macro tile is 256 x 256 x BKT and a wave owns 128 x 64 (6 fragment reads per 8 MFMAs: 0.75 KB each).  LDS images
// are [k][x] with a row pitch of 272 bf16 (544 B = 32 B mod 256: the four k rows of a ds_read_b64_tr_b16 group fall
// on disjoint bank octets), written with one ds_write_b128 per 8 elements; fragments as in bg_frag16<true>.
constexpr int HBMT = 256, HBNT = 256, HPITCH = 272;
// Round 5: the same kernel for k-CONTIGUOUS bf16 operands (element (x, k) at p[x * ld + k]: the A operand of x . W, both
// operands of dy . W^T), so that the readout products of a bf16-operand decoder (model.py:739-755, 1.5 TFLOP per cfg4
// window) run on bf16 copies too instead of rounding f32 operands inside bg_kernel_bf16 (~270 TFLOP/s).  Such an operand
// is staged into an [x][k] image (row pitch BKT + 8 bf16: 16-byte rows of the 32 lanes of a fragment read fall on
// disjoint bank groups, as PITCH16 above), 16 bytes = 8 consecutive k per thread and vector, and its MFMA fragment is
// one plain ds_read_b128 (row xb + li, k = 16 s + 8 kk .. + 7).  XC = x-contiguous (the round-4 path), else k-contiguous.
template <int BKT, bool XC>
struct BghOperand {
    static constexpr int PK = BKT + 8;                                  // [x][k] row pitch (bf16)
    static constexpr int SZ = XC ? BKT * HPITCH : 256 * PK;             // bf16 per image
    static constexpr int NV = BKT / 16;                                 // 16-byte vectors per thread and K-tile
    // element strides of the operand: ld = elements between consecutive k (XC) or consecutive x (!XC)
    static __device__ __forceinline__ void load(const __bf16* __restrict__ p, long long ld, int x0, int X, int k0, int kend,
                                                int t, bf16x8 (&v)[NV]) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            bf16x8 r;
#pragma unroll
            for (int u = 0; u < 8; ++u) r[u] = (__bf16)0.f;
            if (XC) {  // vector id = t + 512 i -> k row id / 32, x = 8 * (id % 32): a wave reads two whole 512-byte rows
                const int k = k0 + (t >> 5) + 16 * i, x = x0 + 8 * (t & 31);
                if (k < kend && x < X) r = *reinterpret_cast<const bf16x8*>(p + (long long)k * ld + x);  // (X % 8 == 0)
            } else {   // vector id = t + 512 i -> row id / (BKT / 8), k group id % (BKT / 8)
                const int id = t + 512 * i;
                const int x = x0 + id / (BKT / 8), k = k0 + 8 * (id % (BKT / 8));
                if (x < X && k < kend) r = *reinterpret_cast<const bf16x8*>(p + (long long)x * ld + k);  // (K % 8 == 0)
            }
            v[i] = r;
        }
    }
    static __device__ __forceinline__ void store(__bf16* __restrict__ s, int t, const bf16x8 (&v)[NV]) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (XC) {
                *reinterpret_cast<bf16x8*>(s + ((t >> 5) + 16 * i) * HPITCH + 8 * (t & 31)) = v[i];
            } else {
                const int id = t + 512 * i;
                *reinterpret_cast<bf16x8*>(s + (id / (BKT / 8)) * PK + 8 * (id % (BKT / 8))) = v[i];
            }
        }
    }
    // MFMA 32x32x16 operand of lane (kk, li): row xb + li, k = 16 s + 8 kk .. +7
    static __device__ __forceinline__ bf16x8 frag(const __bf16* __restrict__ img, int xb, int s_, int kk, int li) {
        if (XC) {
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            typedef __attribute__((address_space(3))) s16x4* lds4;
            const int k0 = 16 * s_ + 8 * kk, q = (li >> 2) & 3, xg = xb + (li & 16) + 4 * (li & 3);
            const __bf16* p = img + (k0 + q) * HPITCH + xg;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(p));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(p + 4 * HPITCH));
            const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            return __builtin_bit_cast(bf16x8, v);
        } else {
            return *reinterpret_cast<const bf16x8*>(img + (xb + li) * PK + 16 * s_ + 8 * kk);
        }
    }
};

template <int BKT, bool AXC, bool BXC>
__global__ __launch_bounds__(512) void bgh_kernel(const BgArgs a, int tiles_m, int tiles_n) {
    typedef BghOperand<BKT, AXC> OA;
    typedef BghOperand<BKT, BXC> OB;
    constexpr int NV = BKT / 16;
    extern __shared__ __attribute__((aligned(16))) __bf16 bgh_smem[];
    __bf16* As = bgh_smem;                 // [2][OA::SZ]
    __bf16* Bs = bgh_smem + 2 * OA::SZ;    // [2][OB::SZ]
    int tm, tn;
    bg_tile_of_block(blockIdx.x, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * HBMT, n0 = tn * HBNT;
    const int z = blockIdx.y;
    const int batch = z / a.splitk, ks = z % a.splitk;
    int kchunk = (a.K + a.splitk - 1) / a.splitk;
    kchunk = (kchunk + BKT - 1) / BKT * BKT;
    const int kbeg = ks * kchunk;
    const int kend = min(a.K, kbeg + kchunk);
    const __bf16* A = reinterpret_cast<const __bf16*>(a.A) + (long long)batch * a.batchA;
    const __bf16* B = reinterpret_cast<const __bf16*>(a.B) + (long long)batch * a.batchB;
    float* C = a.C + (long long)batch * a.batchC;
    const long long lda = AXC ? a.sak : a.sam, ldb = BXC ? a.sbk : a.sbn;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int kk = lane >> 5, li = lane & 31;
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    bf16x8 ra[NV], rb[NV];
    const int nk = (kend - kbeg + BKT - 1) / BKT;
    if (nk > 0) {
        OA::load(A, lda, m0, a.M, kbeg, kend, t, ra);
        OB::load(B, ldb, n0, a.N, kbeg, kend, t, rb);
        OA::store(As, t, ra);
        OB::store(Bs, t, rb);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            OA::load(A, lda, m0, a.M, kbeg + (kt + 1) * BKT, kend, t, ra);
            OB::load(B, ldb, n0, a.N, kbeg + (kt + 1) * BKT, kend, t, rb);
        }
        const __bf16* as = As + cur * OA::SZ;
        const __bf16* bs = Bs + cur * OB::SZ;
#pragma unroll
        for (int s_ = 0; s_ < BKT / 16; ++s_) {
            bf16x8 fa[4], fb[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = OB::frag(bs, wn * 64 + 32 * j, s_, kk, li);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = OA::frag(as, wm * 128 + 32 * i, s_, kk, li);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            OA::store(As + (cur ^ 1) * OA::SZ, t, ra);
            OB::store(Bs + (cur ^ 1) * OB::SZ, t, rb);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + li;
            if (n >= a.N) continue;
            const float bias = (a.bias && ks == 0) ? a.bias[n] : 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int m = m0 + wm * 128 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * kk;
                if (m >= a.M) continue;
                const float v = a.alpha * acc[i][j][q];
                if (a.splitk > 1) {
                    a.ws[((long long)z * a.M + m) * a.N + n] = v;  // summed in slice order by the reducer (which adds the bias)
                } else {
                    float* c = C + (long long)m * a.ldc + n;
                    *c = a.accumulate ? *c + v + bias : v + bias;
                }
            }
        }
}

// ---- split-bf16 variant (BgArgs::bf16 == 3, round 6): f32 operands, f32-grade results, bf16 matrix pipe --------------
// On gfx950 the f32-input MFMA runs at 1/16 of the bf16 rate and there is no xf32.  An f32 value is EXACTLY the sum of
// three bf16 values (x1 = rn(x), x2 = rn(x - x1), x3 = x - x1 - x2: 8 + 8 + 8 significand bits, the residuals are exact
// in f32 and the last one fits bf16), so a * b = sum of nine bf16 products, each exact in f32.  The six with i + j <= 4
// are kept -- (1,1) (1,2) (2,1) (1,3) (3,1) (2,2); the dropped ones are below 2^-26 |a b|, a quarter of the rounding of
// one f32 product -- and accumulated in f32 by six v_mfma_f32_32x32x16_bf16 per 32 x 32 x 16 block: 192 matrix-pipe
// cycles where v_mfma_f32_32x32x2_f32 needs 512.  The operands stay f32 in HBM (no copies): each thread splits the
// values it fetched (5.5 VALU per element, hidden beside the MFMAs of the SIMD's other wave) and writes three bf16
// planes per operand into LDS; 256 x 256 x 16 macro tile, a wave owns 128 x 64: 18 fragment reads feed 48 MFMAs
// (0.375 per MFMA; bgh_kernel: 0.75).  Inf operands give NaN (inf - inf in the residual); finite data only.
// Split-K slices are mapped to XCDs (z % 8 == XCC): the workgroups that run side by side on one XCD's 32 CUs walk the
// same K range, so each XCD streams its own rows of both operands from HBM exactly once.
constexpr int SBK = 16;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void bgs_split(const f32x4& v, bf16x4& p0, bf16x4& p1, bf16x4& p2) {
    p0 = __builtin_convertvector(v, bf16x4);
    const f32x4 r1 = v - __builtin_convertvector(p0, f32x4);
    p1 = __builtin_convertvector(r1, bf16x4);
    const f32x4 r2 = r1 - __builtin_convertvector(p1, f32x4);
    p2 = __builtin_convertvector(r2, bf16x4);
}

template <bool XC>
struct BgsOperand {
    static constexpr int PK = SBK + 8;                                  // [x][k] row pitch (bf16): 48 B
    static constexpr int PLANE = XC ? SBK * HPITCH : 256 * PK;          // bf16 per plane
    static constexpr int SZ = 3 * PLANE;                                // bf16 per stage
    // 256 x 16 f32 per K tile = two 16-byte vectors per thread.  XC: vector id -> (k = id / 64, x = 4 (id % 64)): a wave
    // load is one whole 1 KB row.  KC: id -> (x = id / 4, k = 4 (id % 4)): 16 rows x 64 B per wave load.
    static __device__ __forceinline__ void load(const float* __restrict__ p, long long ld, int x0, int X, int k0, int kend,
                                                bool vec, int t, f32x4 (&v)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = t + 512 * i;
            v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (XC) {
                const int k = k0 + (id >> 6), x = x0 + 4 * (id & 63);
                if (k < kend && x < X) {
                    const float* q = p + (long long)k * ld + x;
                    if (vec && x + 3 < X) v[i] = *reinterpret_cast<const f32x4*>(q);
                    else {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (x + u < X) v[i][u] = q[u];
                    }
                }
            } else {
                const int x = x0 + (id >> 2), k = k0 + 4 * (id & 3);
                if (x < X && k < kend) {
                    const float* q = p + (long long)x * ld + k;
                    if (vec && k + 3 < kend) v[i] = *reinterpret_cast<const f32x4*>(q);
                    else {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (k + u < kend) v[i][u] = q[u];
                    }
                }
            }
        }
    }
    static __device__ __forceinline__ void store(__bf16* __restrict__ s, int t, const f32x4 (&v)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = t + 512 * i;
            bf16x4 p0, p1, p2;
            bgs_split(v[i], p0, p1, p2);
            __bf16* d = XC ? s + (id >> 6) * HPITCH + 4 * (id & 63) : s + (id >> 2) * PK + 4 * (id & 3);
            *reinterpret_cast<bf16x4*>(d) = p0;
            *reinterpret_cast<bf16x4*>(d + PLANE) = p1;
            *reinterpret_cast<bf16x4*>(d + 2 * PLANE) = p2;
        }
    }
    // MFMA 32x32x16 operand of lane (kk, li): row xb + li, k = 8 kk .. +7 of the tile
    static __device__ __forceinline__ bf16x8 frag(const __bf16* __restrict__ img, int xb, int kk, int li) {
        if (XC) {
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            typedef __attribute__((address_space(3))) s16x4* lds4;
            const int q = (li >> 2) & 3, xg = xb + (li & 16) + 4 * (li & 3);
            const __bf16* p = img + (8 * kk + q) * HPITCH + xg;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(p));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(p + 4 * HPITCH));
            const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            return __builtin_bit_cast(bf16x8, v);
        } else {
            return *reinterpret_cast<const bf16x8*>(img + (xb + li) * PK + 8 * kk);
        }
    }
};

template <bool AXC, bool BXC, int ROLES>
__global__ __launch_bounds__(512) void bgs_kernel(const BgArgs a, int vecA, int vecB, int tiles_m, int tiles_n, int zmap) {
    typedef BgsOperand<AXC> OA;
    typedef BgsOperand<BXC> OB;
    extern __shared__ __attribute__((aligned(16))) __bf16 bgs_smem[];
    __bf16* As = bgs_smem;                 // [2][OA::SZ]
    __bf16* Bs = bgs_smem + 2 * OA::SZ;    // [2][OB::SZ]
    int tm, tn, z;
    if (zmap) {  // 1-d grid, slices dealt to XCDs: block b runs on XCD b % 8 and takes slice 8 * (idx / tiles) + b % 8
        const int tiles = tiles_m * tiles_n, idx = blockIdx.x >> 3;
        z = 8 * (idx / tiles) + (blockIdx.x & 7);
        const int tl = idx % tiles;
        tm = tl / tiles_n;
        tn = tl % tiles_n;
    } else {
        bg_tile_of_block(blockIdx.x, tiles_m, tiles_n, tm, tn);
        z = blockIdx.y;
    }
    const int m0 = tm * HBMT, n0 = tn * HBNT;
    const int batch = z / a.splitk, ks = z % a.splitk;
    int kchunk = (a.K + a.splitk - 1) / a.splitk;
    kchunk = (kchunk + SBK - 1) / SBK * SBK;
    const int kbeg = ks * kchunk;
    const int kend = min(a.K, kbeg + kchunk);
    const float* A = a.A + (long long)batch * a.batchA;
    const float* B = a.B + (long long)batch * a.batchB;
    float* C = a.C + (long long)batch * a.batchC;
    const long long lda = AXC ? a.sak : a.sam, ldb = BXC ? a.sbk : a.sbn;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int kk = lane >> 5, li = lane & 31;
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    f32x4 ra[2], rb[2];
    const int nk = (kend - kbeg + SBK - 1) / SBK;
    if (nk > 0) {
        OA::load(A, lda, m0, a.M, kbeg, kend, vecA, t, ra);
        OB::load(B, ldb, n0, a.N, kbeg, kend, vecB, t, rb);
        OA::store(As, t, ra);
        OB::store(Bs, t, rb);
        if (nk > 1) {
            OA::load(A, lda, m0, a.M, kbeg + SBK, kend, vecA, t, ra);
            OB::load(B, ldb, n0, a.N, kbeg + SBK, kend, vecB, t, rb);
        }
    }
    __syncthreads();
    // Waves w and w + 4 share a SIMD (a workgroup's waves are dealt to the SIMDs cyclically).  ROLES: the upper four
    // split and store the next tile BEFORE their MFMAs, the lower four after, so that on every SIMD one wave's vector /
    // LDS-store work runs beside the other's matrix work instead of all eight meeting in the same phase.
    const bool early = ROLES && (wave >= 4);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const __bf16* as = As + cur * OA::SZ;
        const __bf16* bs = Bs + cur * OB::SZ;
        if (early && kt + 1 < nk) {
            OA::store(As + (cur ^ 1) * OA::SZ, t, ra);
            OB::store(Bs + (cur ^ 1) * OB::SZ, t, rb);
            if (kt + 2 < nk) {
                OA::load(A, lda, m0, a.M, kbeg + (kt + 2) * SBK, kend, vecA, t, ra);
                OB::load(B, ldb, n0, a.N, kbeg + (kt + 2) * SBK, kend, vecB, t, rb);
            }
        }
        bf16x8 fb[2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) fb[j][p] = OB::frag(bs + p * OB::PLANE, wn * 64 + 32 * j, kk, li);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf16x8 fa[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) fa[p] = OA::frag(as + p * OA::PLANE, wm * 128 + 32 * i, kk, li);
            // smallest terms first; the two accumulators alternate so that no MFMA waits for its predecessor
#define BGS_MM(pa, pb)                                                                                    \
    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[pa], fb[0][pb], acc[i][0], 0, 0, 0);          \
    acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[pa], fb[1][pb], acc[i][1], 0, 0, 0)
            BGS_MM(2, 0);
            BGS_MM(0, 2);
            BGS_MM(1, 1);
            BGS_MM(1, 0);
            BGS_MM(0, 1);
            BGS_MM(0, 0);
#undef BGS_MM
        }
        if (!early && kt + 1 < nk) {
            OA::store(As + (cur ^ 1) * OA::SZ, t, ra);
            OB::store(Bs + (cur ^ 1) * OB::SZ, t, rb);
            if (kt + 2 < nk) {
                OA::load(A, lda, m0, a.M, kbeg + (kt + 2) * SBK, kend, vecA, t, ra);
                OB::load(B, ldb, n0, a.N, kbeg + (kt + 2) * SBK, kend, vecB, t, rb);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + li;
            if (n >= a.N) continue;
            const float bias = (a.bias && ks == 0) ? a.bias[n] : 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int m = m0 + wm * 128 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * kk;
                if (m >= a.M) continue;
                const float v = a.alpha * acc[i][j][q];
                if (a.splitk > 1) {
                    a.ws[((long long)z * a.M + m) * a.N + n] = v;  // summed in slice order by the reducer (which adds the bias)
                } else {
                    float* c = C + (long long)m * a.ldc + n;
                    float r = a.accumulate ? *c + v + bias : v + bias;
                    if (a.act == 1) r = fmaxf(r, 0.f);  // (tanh / sigmoid epilogues stay on the f32 kernel: bg_launch)
                    if (a.gate && !(a.gate[(long long)m * a.ldg + n] > 0.f)) r = 0.f;
                    *c = r;
                }
            }
        }
}

// f32 -> bf16 (round to nearest even), 8 elements per thread: the operand copies bgh_kernel reads.
__global__ __launch_bounds__(256) void bg_to_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ y, long long n8) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(x + 8 * i);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(x + 8 * i + 4);
        *reinterpret_cast<bf16x8*>(y + 8 * i) = ph_bf16x8(lo, hi);
    }
}

// C[b][m][n] (+)= bias[n] + sum over the K slices, in slice order (deterministic split-K).
__global__ __launch_bounds__(256) void bg_reduce_kernel(const BgArgs a) {
    const long long mn = (long long)a.M * a.N;
    const long long total = mn * a.nbatch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int batch = (int)(i / mn);
        const long long r = i % mn;
        const int m = (int)(r / a.N), n = (int)(r % a.N);
        float* c = a.C + (long long)batch * a.batchC + (long long)m * a.ldc + n;
        float v = a.accumulate ? *c : 0.f;
        if (a.bias) v += a.bias[n];
        const float* w = a.ws + ((long long)batch * a.splitk) * mn + r;
        int ks = 0;  // (eight slices requested per round, added in slice order)
        for (; ks + 8 <= a.splitk; ks += 8) {
            float p[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) p[q] = w[(long long)(ks + q) * mn];
#pragma unroll
            for (int q = 0; q < 8; ++q) v += p[q];
        }
        for (; ks < a.splitk; ++ks) v += w[(long long)ks * mn];
        if (a.gate && !(a.gate[(long long)m * a.ldg + n] > 0.f)) v = 0.f;
        *c = v;
    }
}

}  // namespace

int bg_reduce_launch(const BgArgs& a, hipStream_t stream) {
    const long long total = (long long)a.M * a.N * a.nbatch;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(bg_reduce_kernel, dim3(blocks), dim3(256), 0, stream, a);
    return (int)hipGetLastError();
}

void bg_tile_shape(int bf16, int& bm, int& bn) {
    bm = bn = 128;
    if (bf16 == 2 || bf16 == 3) { bm = HBMT; bn = HBNT; return; }
}

// One "dynamic LDS above 64 KB" attribute call per kernel instantiation (a function template has one static per
// instantiation; the generic lambda this replaces had ONE flag for all bgh_kernel variants: ADVICE r05).
template <typename K>
static void bg_big_lds_once(K kern, bool& done) {
    if (!done) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        done = true;
    }
}

template <int BKT, bool AX, bool BX>
static void bgh_go(const BgArgs& a, dim3 grid, int tm, int tn, hipStream_t stream) {
    static bool done = false;
    bg_big_lds_once(bgh_kernel<BKT, AX, BX>, done);
    const size_t lds = (size_t)4 * (BghOperand<BKT, AX>::SZ + BghOperand<BKT, BX>::SZ);
    hipLaunchKernelGGL((bgh_kernel<BKT, AX, BX>), grid, dim3(512), lds, stream, a, tm, tn);
}

static int bgs_roles() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("PARROT_GEMM_SPLIT_ROLES");
        v = e ? atoi(e) : 1;
    }
    return v;
}

template <bool AX, bool BX, int ROLES>
static void bgs_go2(const BgArgs& a, dim3 grid, int vecA, int vecB, int tm, int tn, int zmap, hipStream_t stream) {
    static bool done = false;
    bg_big_lds_once(bgs_kernel<AX, BX, ROLES>, done);
    const size_t lds = (size_t)4 * (BgsOperand<AX>::SZ + BgsOperand<BX>::SZ);
    hipLaunchKernelGGL((bgs_kernel<AX, BX, ROLES>), grid, dim3(512), lds, stream, a, vecA, vecB, tm, tn, zmap);
}

template <bool AX, bool BX>
static void bgs_go(const BgArgs& a, dim3 grid, int vecA, int vecB, int tm, int tn, int zmap, hipStream_t stream) {
    if (bgs_roles()) bgs_go2<AX, BX, 1>(a, grid, vecA, vecB, tm, tn, zmap, stream);
    else bgs_go2<AX, BX, 0>(a, grid, vecA, vecB, tm, tn, zmap, stream);
}

int bg_to_bf16_launch(const float* x, void* y, long long n, hipStream_t stream) {
    if (n <= 0) return 0;
    if ((n & 7) || ((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return PH_ERR_BADARG;
    const long long n8 = n / 8;
    int blocks = (int)((n8 + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(bg_to_bf16_kernel, dim3(blocks), dim3(256), 0, stream, x, reinterpret_cast<__bf16*>(y), n8);
    return (int)hipGetLastError();
}

int bg_launch(const BgArgs& a, hipStream_t stream) {
    const size_t pad = 0;
    if (a.M <= 0 || a.N <= 0 || a.K < 0 || a.nbatch < 1 || a.splitk < 1 || (a.splitk > 1 && !a.ws)) return PH_ERR_BADARG;
    const bool axc = (a.sam == 1), bxc = (a.sbn == 1);
    if (!axc && a.sak != 1) return PH_ERR_BADARG;
    if (!bxc && a.sbk != 1) return PH_ERR_BADARG;
    // vector loads need 16-byte aligned addresses for every (tile, k) start.
    auto al = [](const float* p, long long stride, long long batch) {
        return (((uintptr_t)p & 15) == 0) && (stride % 4 == 0) && (batch % 4 == 0);
    };
    const int vecA = axc ? al(a.A, a.sak, a.batchA) : al(a.A, a.sam, a.batchA);
    const int vecB = bxc ? al(a.B, a.sbk, a.batchB) : al(a.B, a.sbn, a.batchB);
    const int tiles_m = ceil_div(a.M, BM), tiles_n = ceil_div(a.N, BN);
    dim3 grid(tiles_m * tiles_n, a.nbatch * a.splitk);
    dim3 block(256);
    if (a.bf16 == 2) {  // operands ARE bf16 in memory (strides in bf16 elements): bgh_kernel, any "one stride is 1" layout
        if (a.act || a.gate || ((uintptr_t)a.A & 15) || ((uintptr_t)a.B & 15) || (a.batchA & 7) || (a.batchB & 7))
            return PH_ERR_UNSUPPORTED;
        // 16-byte vectors: along x for an x-contiguous operand (its extent and its k stride multiples of 8), along k for a
        // k-contiguous one (K and its x stride multiples of 8)
        if (axc ? ((a.M & 7) || (a.sak & 7)) : ((a.K & 7) || (a.sam & 7))) return PH_ERR_UNSUPPORTED;
        if (bxc ? ((a.N & 7) || (a.sbk & 7)) : ((a.K & 7) || (a.sbn & 7))) return PH_ERR_UNSUPPORTED;
        const int tm = ceil_div(a.M, HBMT), tn = ceil_div(a.N, HBNT);
        // K tile: 32 for the TN products (round 4: 833 TFLOP/s; PARROT_GEMM_BF16IN_BK=64 for experiments).  A k-contiguous
        // operand is fetched as 16-byte vectors along k: with a 32-deep tile a wave load touches 16 rows x 64 bytes = 16 half
        // cache lines, which the CU's vector-memory path moves at a quarter of the rate of whole lines (first
        // r05 build: 310 TFLOP/s); with 64 a wave load is 8 rows x one whole 128-byte line.
        const int bkt = (axc && bxc) ? 32 : 64;
        const dim3 g2(tm * tn, a.nbatch * a.splitk);
        if (bkt == 64) {
            if (axc && bxc) bgh_go<64, true, true>(a, g2, tm, tn, stream);
            else if (axc) bgh_go<64, true, false>(a, g2, tm, tn, stream);
            else if (bxc) bgh_go<64, false, true>(a, g2, tm, tn, stream);
            else bgh_go<64, false, false>(a, g2, tm, tn, stream);
        } else {
            bgh_go<32, true, true>(a, g2, tm, tn, stream);
        }
        return (int)hipGetLastError();
    }
    if (a.bf16 == 3) {  // f32 operands split into three bf16 terms each, six bf16 MFMAs per block: bgs_kernel
        if (a.act > 1) return PH_ERR_UNSUPPORTED;
        const int tm = ceil_div(a.M, HBMT), tn = ceil_div(a.N, HBNT);
        const int Z = a.nbatch * a.splitk;
        const int zmap = (Z % 8 == 0) ? 1 : 0;
        const dim3 g = zmap ? dim3(tm * tn * Z) : dim3(tm * tn, Z);
        if (axc && bxc) bgs_go<true, true>(a, g, vecA, vecB, tm, tn, zmap, stream);
        else if (axc) bgs_go<true, false>(a, g, vecA, vecB, tm, tn, zmap, stream);
        else if (bxc) bgs_go<false, true>(a, g, vecA, vecB, tm, tn, zmap, stream);
        else bgs_go<false, false>(a, g, vecA, vecB, tm, tn, zmap, stream);
        return (int)hipGetLastError();
    }
    if (a.bf16) {
        dim3 b8(512);
        // vector loads of the k-contiguous layout take 8 consecutive k: both 16-byte halves must be aligned
        if (axc && bxc) hipLaunchKernelGGL((bg_kernel_bf16<true, true>), grid, b8, pad, stream, a, vecA, vecB, tiles_m, tiles_n);
        else if (axc && !bxc) hipLaunchKernelGGL((bg_kernel_bf16<true, false>), grid, b8, pad, stream, a, vecA, vecB, tiles_m, tiles_n);
        else if (!axc && bxc) hipLaunchKernelGGL((bg_kernel_bf16<false, true>), grid, b8, pad, stream, a, vecA, vecB, tiles_m, tiles_n);
        else hipLaunchKernelGGL((bg_kernel_bf16<false, false>), grid, b8, pad, stream, a, vecA, vecB, tiles_m, tiles_n);
        return (int)hipGetLastError();
    }
    constexpr bool w8 = true;  // measured on MI355X: 113-124 TFLOP/s vs 87-102 for the 4-wave kernel (kept below for reference)
    if (w8) {
        dim3 b8(512);
        if (axc && bxc) hipLaunchKernelGGL((bg_kernel8<true, true>), grid, b8, pad, stream, a, vecA, vecB, tiles_m, tiles_n);
        else if (axc && !bxc) hipLaunchKernelGGL((bg_kernel8<true, false>), grid, b8, pad, stream, a, vecA, vecB, tiles_m, tiles_n);
        else if (!axc && bxc) hipLaunchKernelGGL((bg_kernel8<false, true>), grid, b8, pad, stream, a, vecA, vecB, tiles_m, tiles_n);
        else hipLaunchKernelGGL((bg_kernel8<false, false>), grid, b8, pad, stream, a, vecA, vecB, tiles_m, tiles_n);
        return (int)hipGetLastError();
    }
    if (axc && bxc) hipLaunchKernelGGL((bg_kernel<true, true>), grid, block, pad, stream, a, vecA, vecB, tiles_m, tiles_n);
    else if (axc && !bxc) hipLaunchKernelGGL((bg_kernel<true, false>), grid, block, pad, stream, a, vecA, vecB, tiles_m, tiles_n);
    else if (!axc && bxc) hipLaunchKernelGGL((bg_kernel<false, true>), grid, block, pad, stream, a, vecA, vecB, tiles_m, tiles_n);
    else hipLaunchKernelGGL((bg_kernel<false, false>), grid, block, pad, stream, a, vecA, vecB, tiles_m, tiles_n);
    return (int)hipGetLastError();
}
