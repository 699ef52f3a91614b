// U1: small-M linear (timestep-embedding MLP and the per-ResBlock temb projections, all ResBlocks in
// one call).  out[m][n] = sum_k act(x[m][k]) * w[n][k] + bias[n], M <= 64.
// x (with the optional SiLU applied once) is staged in LDS; each wave owns output columns, lanes split
// K with float4 loads of the weight row (coalesced, each weight read once), wave-reduced by shuffles.
// No reference file exists to cite (reference snapshot is empty); semantics = torch F.linear.
#include "common.h"

using namespace cdx;

namespace {

constexpr int kMB = 16;            // rows accumulated per pass
constexpr int kColsPerWave = 8;
constexpr int kMaxXFloats = 16384; // 64 KiB of LDS

__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, int x_ld, const float* __restrict__ w,
                                                     const float* __restrict__ bias, int M, int N, int K, int silu_in,
                                                     float* __restrict__ out, int out_ld) {
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [M][K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid * 4; i < M * K; i += 1024) {
        const int m = i / K, k = i - m * K;
        f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)m * x_ld + k);
        if (silu_in) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] / (1.0f + expf(-v[e]));
        }
        *reinterpret_cast<f32x4*>(xs + i) = v;
    }
    __syncthreads();
    const int n0 = (blockIdx.x * 4 + wave) * kColsPerWave;
    for (int n = n0; n < min(N, n0 + kColsPerWave); ++n) {
        const float* __restrict__ wr = w + (size_t)n * K;
        for (int m0 = 0; m0 < M; m0 += kMB) {
            float acc[kMB];
#pragma unroll
            for (int j = 0; j < kMB; ++j) acc[j] = 0.f;
            for (int k = lane * 4; k < K; k += 256) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + k);
#pragma unroll
                for (int j = 0; j < kMB; ++j) {
                    if (m0 + j < M) {
                        const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + (size_t)(m0 + j) * K + k);
                        acc[j] = fmaf(wv[0], xv[0], acc[j]);
                        acc[j] = fmaf(wv[1], xv[1], acc[j]);
                        acc[j] = fmaf(wv[2], xv[2], acc[j]);
                        acc[j] = fmaf(wv[3], xv[3], acc[j]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < kMB; ++j) {
                float v = acc[j];
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
                if (lane == 0 && m0 + j < M) out[(size_t)(m0 + j) * out_ld + n] = v + (bias ? bias[n] : 0.f);
            }
        }
    }
}

}  // namespace

extern "C" size_t cdx_linear_f32_workspace(const cdx_linear_args*) { return 0; }

extern "C" int cdx_linear_f32(const cdx_linear_args* a, void*, size_t, cdx_stream_t stream) {
    CDX_REQUIRE(a && a->x && a->w && a->out);
    CDX_REQUIRE(a->m > 0 && a->m <= 64 && a->n > 0 && a->k > 0 && (a->k % 4) == 0);
    CDX_REQUIRE(a->x_ld >= a->k && (a->x_ld % 4) == 0 && a->out_ld >= a->n);
    CDX_REQUIRE(aligned16(a->x) && aligned16(a->w));
    if ((int64_t)a->m * a->k > kMaxXFloats) return CDX_ENOTSUP;
    const int cols_per_wg = 4 * kColsPerWave;
    hipLaunchKernelGGL(linear_kernel, dim3((a->n + cols_per_wg - 1) / cols_per_wg), dim3(256),
                       (size_t)a->m * a->k * sizeof(float), static_cast<hipStream_t>(stream), a->x, a->x_ld, a->w,
                       a->bias, a->m, a->n, a->k, (a->flags & CDX_LINEAR_SILU_IN) ? 1 : 0, a->out, a->out_ld);
    return check_launch();
}
