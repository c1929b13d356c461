// Audio quantisers of reference quantize.py on the GPU: per-row min-max normalisation in float64
// followed by 8-bit mu-law (quantize.py:44-66, 83-99) or linear (quantize.py:20-36) encoding, and
// mu-law expansion (quantize.py:68-78).  Integer outputs are a bit-exact contract, so the encode
// path does the reference's float64 arithmetic operation by operation (IEEE add/sub/mul/div are
// exact matches; log() is the only library call).  HBM-bound: 4 B in, 2 or 4 B out per sample.
#include "quantize.h"

// The reference does separate IEEE operations; never let the compiler contract them into FMAs.
#pragma clang fp contract(off)

namespace {

// Pass 1: per-row min / max with many workgroups per row (grid.x chunks x grid.y rows).  Floats are
// combined with integer atomics on an order-preserving key (monotone map float -> uint32), so the result is
// exact and independent of the order of arrival.
__device__ __forceinline__ unsigned f2key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ __launch_bounds__(256) void row_minmax_kernel(const float* __restrict__ x, int n, int ld,
                                                         unsigned* __restrict__ keys) {
    __shared__ float smn[4], smx[4];
    const float* row = x + (size_t)blockIdx.y * ld;
    float mn = INFINITY, mx = -INFINITY;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float v = row[i];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o, 64));
        mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        smn[threadIdx.x >> 6] = mn;
        smx[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
        mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
        atomicMin(&keys[2 * blockIdx.y], f2key(mn));
        atomicMax(&keys[2 * blockIdx.y + 1], f2key(mx));
    }
}

__global__ void minmax_init_kernel(unsigned* keys, int rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows) {
        keys[2 * i] = 0xffffffffu;
        keys[2 * i + 1] = 0u;
    }
}

template <int MODE>  // 0 = mu-law -> int16, 1 = linear -> int32
__global__ __launch_bounds__(256) void quantize_kernel(const float* __restrict__ x, int n, int ld,
                                                       const double* __restrict__ mnmx, void* __restrict__ out,
                                                       int ldo, int q_levels) {
    const int r = blockIdx.y;
    const unsigned* keys = reinterpret_cast<const unsigned*>(mnmx);
    const double mn = (double)key2f(keys[2 * r]);
    const double rng = (double)key2f(keys[2 * r + 1]) - mn;  // max of the shifted row (quantize.py:16-17)
    const float* row = x + (size_t)r * ld;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        double d = (double)row[i];
        d -= mn;
        d /= rng;
        if (MODE == 0) {
            d = 2. * d - 1.;
            const double sgn = (d > 0.) ? 1. : ((d < 0.) ? -1. : 0.);
            const double xmu = sgn * log(1. + 255. * fabs(d)) / 5.545177444479562;  // np.log(256.)
            const double q = (xmu + 1.) / 2. * 255.;
            reinterpret_cast<int16_t*>(out)[(size_t)r * ldo + i] = (int16_t)(int)q;  // trunc toward 0
        } else {
            const double eps = 1e-5;
            d *= ((double)q_levels - eps);
            d += eps / 2;
            reinterpret_cast<int32_t*>(out)[(size_t)r * ldo + i] = (int32_t)d;
        }
    }
}

__global__ __launch_bounds__(256) void mu2linear_kernel(const int32_t* __restrict__ q, size_t n,
                                                        float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float x = (float)q[i];
        const float y = 2.f * (x - 128.f) / 256.f;
        const float sgn = (y > 0.f) ? 1.f : ((y < 0.f) ? -1.f : 0.f);
        out[i] = sgn * 0.00392156862745098f * (powf(256.f, fabsf(y)) - 1.f);
    }
}

}  // namespace

int quantize_launch(const float* x, int rows, int n, int ld, double* mnmx_ws, void* out, int ldo, int mode,
                    int q_levels, hipStream_t stream) {
    if (rows < 1 || n < 1 || (mode != 0 && mode != 1)) return PH_ERR_BADARG;
    unsigned* keys = reinterpret_cast<unsigned*>(mnmx_ws);
    hipLaunchKernelGGL(minmax_init_kernel, dim3(ceil_div(rows, 256)), dim3(256), 0, stream, keys, rows);
    int bx = ceil_div(n, 256 * 8);
    if (bx < 1) bx = 1;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(row_minmax_kernel, dim3(bx, rows), dim3(256), 0, stream, x, n, ld, keys);
    bx = ceil_div(n, 256 * 4);
    if (bx < 1) bx = 1;
    if (bx > 256) bx = 256;
    if (mode == 0)
        hipLaunchKernelGGL(quantize_kernel<0>, dim3(bx, rows), dim3(256), 0, stream, x, n, ld, mnmx_ws, out, ldo, q_levels);
    else
        hipLaunchKernelGGL(quantize_kernel<1>, dim3(bx, rows), dim3(256), 0, stream, x, n, ld, mnmx_ws, out, ldo, q_levels);
    return (int)hipGetLastError();
}

int mu2linear_launch(const int32_t* q, size_t n, float* out, hipStream_t stream) {
    int bx = (int)((n + 255) / 256);
    if (bx < 1) bx = 1;
    if (bx > 4096) bx = 4096;
    hipLaunchKernelGGL(mu2linear_kernel, dim3(bx), dim3(256), 0, stream, q, n, out);
    return (int)hipGetLastError();
}
