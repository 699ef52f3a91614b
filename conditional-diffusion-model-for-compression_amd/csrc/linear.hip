// U1: small-M linear (timestep-embedding MLP and the per-ResBlock temb projections, all ResBlocks in
// one call).  out[m][n] = sum_k act(x[m][k]) * w[n][k] + bias[n]; any M (grid.y walks blocks of 16 rows), any K (walked in
// slabs of 1024: 64 KiB of x in LDS per slab; ADVICE r02: K = 4 * base_channels > 1024 used to be refused).
// The workgroup's 16 rows of x (with the optional SiLU applied once) are staged in LDS.  A LANE owns one output column (64 columns per workgroup),
// the 4 waves split K into quarters, every x value is a broadcast LDS read shared by the 64 columns, and the only
// reduction is the fixed-order sum of the four K quarters through LDS at the end.  Weight rows are read 16 bytes per
// lane (a 128-byte line serves 8 consecutive steps from L1/L2).  Measured on the 16 x 512 x 8704 projection call
// (weights 17.8 MB = 4 us of HBM): one column per WAVE with 16 x 6 shuffle steps per column 57 us -> this form 27 us
// (rows split over the waves instead of K, no reduction at all: 43 us -- the kernel is bound by how many weight bytes
// are in flight, and the K split quadruples the waves per column block).
// No reference file exists to cite (reference snapshot is empty); semantics = torch F.linear.
#include "common.h"

using namespace cdx;

namespace {

constexpr int kMB = 16;            // rows per workgroup
constexpr int kMaxK = 1024;        // kMB * K floats of x in LDS (64 KiB) + 16 KiB for the reduction

__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, int x_ld, const float* __restrict__ w,
                                                     const float* __restrict__ bias, int M, int N, int K, int silu_in,
                                                     float* __restrict__ out, int out_ld) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // xs[rows][KS], then red[4][kMB][64]
    const int m0 = blockIdx.y * kMB;
    const int rows = min(kMB, M - m0);                             // block-uniform
    const int KS = min(K, kMaxK);                                  // slab width
    float* xs = smem;
    float* red = smem + (size_t)kMB * KS;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.x * 64 + lane;
    const bool nok = n < N;
    const float* __restrict__ wr = w + (size_t)(nok ? n : 0) * K;
    float acc[kMB];
#pragma unroll
    for (int j = 0; j < kMB; ++j) acc[j] = 0.f;
    for (int ks = 0; ks < K; ks += kMaxK) {                        // one pass for K <= 1024 (every UNet up to 256 base channels)
        const int kw = min(kMaxK, K - ks);                         // this slab's width (multiple of 4)
        if (ks) __syncthreads();                                   // every wave is done with the previous slab
        for (int i = tid * 4; i < rows * kw; i += 1024) {
            const int m = i / kw, k = i - m * kw;
            f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)(m0 + m) * x_ld + ks + k);
            if (silu_in) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] / (1.0f + expf(-v[e]));
            }
            *reinterpret_cast<f32x4*>(xs + (size_t)m * KS + k) = v;
        }
        __syncthreads();
        // this wave's quarter of the slab (multiples of 4); a wave adds its quarters of successive slabs in slab order
        const int kq = ((kw / 4 + 3) / 4) * 4;
        const int k0 = wave * kq, k1 = min(kw, k0 + kq);
        for (int k = k0; k < k1; k += 4) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + ks + k);
#pragma unroll
            for (int j = 0; j < kMB; ++j) {
                if (j < rows) {      // block-uniform
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + (size_t)j * KS + k);   // broadcast
                    acc[j] = fmaf(wv[0], xv[0], acc[j]);
                    acc[j] = fmaf(wv[1], xv[1], acc[j]);
                    acc[j] = fmaf(wv[2], xv[2], acc[j]);
                    acc[j] = fmaf(wv[3], xv[3], acc[j]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < kMB; ++j) red[(wave * kMB + j) * 64 + lane] = acc[j];
    __syncthreads();
    // the rows are split over the 4 waves for the final sum (fixed order: quarter 0, 1, 2, 3 -- a row's bits do not
    // depend on M or on which row block it falls in)
    for (int j = wave; j < kMB; j += 4) {
        if (j < rows && nok) {
            const float v = ((red[(0 * kMB + j) * 64 + lane] + red[(1 * kMB + j) * 64 + lane]) + red[(2 * kMB + j) * 64 + lane]) +
                            red[(3 * kMB + j) * 64 + lane];
            out[(size_t)(m0 + j) * out_ld + n] = v + (bias ? bias[n] : 0.f);
        }
    }
}

}  // namespace

extern "C" size_t cdx_linear_f32_workspace(const cdx_linear_args*) { return 0; }

extern "C" int cdx_linear_f32(const cdx_linear_args* a, void*, size_t, cdx_stream_t stream) {
    CDX_REQUIRE(a && a->x && a->w && a->out);
    CDX_REQUIRE(a->m > 0 && a->n > 0 && a->k > 0 && (a->k % 4) == 0);
    CDX_REQUIRE((a->flags & ~CDX_LINEAR_SILU_IN) == 0);
    CDX_REQUIRE(a->x_ld >= a->k && (a->x_ld % 4) == 0 && a->out_ld >= a->n);
    CDX_REQUIRE(aligned16(a->x) && aligned16(a->w));
    if ((a->m + kMB - 1) / kMB > 65535) return CDX_ENOTSUP;
    const size_t lds_bytes = ((size_t)kMB * (a->k < kMaxK ? a->k : kMaxK) + 4 * kMB * 64) * sizeof(float);      // <= 80 KiB
    if (lds_bytes > 48 * 1024 &&      // more dynamic LDS than the default allowance: declare it (idempotent, no sync)
        hipFuncSetAttribute(reinterpret_cast<const void*>(linear_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
        return CDX_ELAUNCH;
    hipLaunchKernelGGL(linear_kernel, dim3((a->n + 63) / 64, (a->m + kMB - 1) / kMB), dim3(256),
                       lds_bytes, static_cast<hipStream_t>(stream), a->x, a->x_ld, a->w,
                       a->bias, a->m, a->n, a->k, (a->flags & CDX_LINEAR_SILU_IN) ? 1 : 0, a->out, a->out_ld);
    return check_launch();
}
