// U2: GroupNorm statistics -> per-(batch, channel) scale / shift, consumed by the convolution's
// tile gather (so GroupNorm+SiLU never makes its own HBM round trip).
//
// Two HBM-bound kernels, no atomics, fixed summation order (bitwise reproducible, and independent of
// the batch size: the pixel split depends on hw only, so an image's statistics do not change with
// sharding):
//   gn_partial : grid (nsplit, batch); each workgroup sums x and x^2 per channel over its pixel
//                range, float4 loads, float64 accumulators -> partial[b][split][c][2]
//   gn_finalize: one wave per (batch, group): sums the partials in float64, mean / rstd, then
//                scale = rstd*gamma in float32 (F.group_norm's own form) and shift = beta - mean*scale evaluated in
//                float64 and rounded once (gn_affine below).
// No reference file exists to cite (reference snapshot is empty); semantics = torch F.group_norm.
#include <math.h>
#include <stdlib.h>

#include "common.h"

using namespace cdx;

namespace {

constexpr int kPixelsPerSplit = 256;

__host__ __device__ inline int gn_nsplit(int hw) {
    int n = hw / kPixelsPerSplit;
    return n < 1 ? 1 : (n > 1024 ? 1024 : n);
}

// scale = rstd * gamma in float32 (times 2^out_exp: exact).  shift = beta - mean * scale is evaluated in FLOAT64 from the float64
// mean and the FLOAT32 scale the consumer multiplies by, and rounded once: x * scale + shift then equals (x - mean) * scale + beta
// up to that one rounding.  The all-float32 form rounds the mean, the product and the sum -- each costs |mean| / sigma ulps of the
// NORMALISED value (a layer whose input is a large constant plus a small signal: conv_in's bias 0.1 + a 1e-4 input is 2000
// sigmas from zero), which round 3 measured as the whole-forward error at input scale 1e-4.
// legacy (tuning build only): the round-3 float32 form, for the before / after record.
__device__ __forceinline__ void gn_affine(double mean, float rstdf, float gamma, float beta, float oscale, bool legacy, float& scale, float& shift) {
    const float sc = rstdf * gamma;
    scale = sc * oscale;
    shift = legacy ? (-sc * (float)mean + beta) * oscale : (float)((double)beta - mean * (double)sc) * oscale;
}

__global__ __launch_bounds__(256) void gn_partial(const float* __restrict__ src, int C, int hw, int nsplit,
                                                  double* __restrict__ part, int coff, int ctot) {
    __shared__ double red[256 * 8];
    const int tid = threadIdx.x;
    const int split = blockIdx.x, b = blockIdx.y;
    const int per = (hw + nsplit - 1) / nsplit;
    const int p0 = split * per;
    const int p1 = min(hw, p0 + per);
    const int nq = C >> 2;
    const int qw = nq < 256 ? nq : 256;     // quads handled side by side
    const int rows = 256 / qw;              // pixel rows handled side by side
    const int qi = tid % qw, r = tid / qw;
    const bool active = r < rows;
    const float* __restrict__ base = src + (size_t)b * hw * C;
    for (int qb = 0; qb < nq; qb += qw) {
        const int q = qb + qi;
        double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
        if (active && q < nq) {
            for (int p = p0 + r; p < p1; p += rows) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)p * C + q * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double d = (double)v[e];
                    s[e] += d;
                    ss[e] = fma(d, d, ss[e]);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[tid * 8 + e] = s[e];
            red[tid * 8 + 4 + e] = ss[e];
        }
        __syncthreads();
        if (r == 0 && q < nq) {
            double* o = part + (((size_t)b * nsplit + split) * ctot + coff + q * 4) * 2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double a = 0, c = 0;
                for (int k = 0; k < rows; ++k) {   // fixed order
                    a += red[(k * qw + qi) * 8 + e];
                    c += red[(k * qw + qi) * 8 + 4 + e];
                }
                o[e * 2] = a;
                o[e * 2 + 1] = c;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void gn_finalize(const double* __restrict__ part, int nsplit, int ctot, int groups,
                                                  int hw, float eps, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, float* __restrict__ scale,
                                                  float* __restrict__ shift, float* __restrict__ mean_out,
                                                  float* __restrict__ rstd_out, float oscale, bool legacy) {
    const int g = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int cpg = ctot / groups;
    const int items = nsplit * cpg;
    double s = 0, ss = 0;
    for (int it = lane; it < items; it += 64) {
        const int split = it / cpg, c = g * cpg + it % cpg;
        const double* q = part + (((size_t)b * nsplit + split) * ctot + c) * 2;
        s += q[0];
        ss += q[1];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        s += __shfl_xor(s, off);
        ss += __shfl_xor(ss, off);
    }
    const double n = (double)hw * cpg;
    const double mean = s / n;
    double var = ss / n - mean * mean;
    var = var < 0 ? 0 : var;
    const float meanf = (float)mean;
    const float rstdf = (float)(1.0 / sqrt(var + (double)eps));
    if (lane == 0) {
        if (mean_out) mean_out[b * groups + g] = meanf;
        if (rstd_out) rstd_out[b * groups + g] = rstdf;
    }
    for (int k = lane; k < cpg; k += 64) {
        const int c = g * cpg + k;
        gn_affine(mean, rstdf, gamma[c], beta[c], oscale, legacy, scale[(size_t)b * ctot + c], shift[(size_t)b * ctot + c]);
    }
}

// Fused form: partial sums were left by the producing convolutions (conv_kernel epilogue), one slot per
// (spatial tile, wave row); the two sources of a channel concat each bring their own slot count.
// One workgroup of 256 threads per (image, group).  A thread owns ONE channel of the group and every tpc-th slot
// (tpc = 256 / channels-per-group): no index arithmetic in the loop, 4x the loads in flight of the one-wave version
// (13 us average, 40 us at 256^2 where a group is 4 channels x 1024 slots = 64 KiB of 16-byte reads at 2 KiB stride).
// Summation order is fixed by (thread, slot) and by the LDS tree below: results do not depend on timing or batch.
__global__ __launch_bounds__(256) void gn_finalize2(const double* __restrict__ part0, int slots0, int c0,
                                                    const double* __restrict__ part1, int slots1, int c1, int groups,
                                                    int hw, float eps, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float* __restrict__ scale,
                                                    float* __restrict__ shift, float* __restrict__ mean_out,
                                                    float* __restrict__ rstd_out, float oscale, bool legacy) {
    __shared__ double red[2][256];
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int ctot = c0 + c1, cpg = ctot / groups;
    const int cbeg = g * cpg;
    const int tpc = cpg <= 256 ? 256 / cpg : 1;          // threads per channel
    double s = 0, ss = 0;
    for (int k0 = 0; k0 < cpg; k0 += 256) {             // one pass unless a group has more than 256 channels
        const int k = k0 + (cpg <= 256 ? tid % cpg : tid), j = cpg <= 256 ? tid / cpg : 0;
        if (k < cpg && j < tpc) {
            const int c = cbeg + k;
            const bool src1 = c >= c0;
            const int slots = src1 ? slots1 : slots0, cs = src1 ? c1 : c0;
            const double* q = (src1 ? part1 : part0) + (((size_t)b * slots + j) * cs + (src1 ? c - c0 : c)) * 2;
            const size_t step = (size_t)tpc * cs * 2;
            for (int slot = j; slot < slots; slot += tpc, q += step) {
                s += q[0];
                ss += q[1];
            }
        }
    }
    red[0][tid] = s;
    red[1][tid] = ss;
    __syncthreads();
#pragma unroll
    for (int w = 128; w >= 1; w >>= 1) {
        if (tid < w) {
            red[0][tid] += red[0][tid + w];
            red[1][tid] += red[1][tid + w];
        }
        __syncthreads();
    }
    const double n = (double)hw * cpg;
    const double mean = red[0][0] / n;
    double var = red[1][0] / n - mean * mean;
    var = var < 0 ? 0 : var;
    const float meanf = (float)mean;
    const float rstdf = (float)(1.0 / sqrt(var + (double)eps));
    if (tid == 0) {
        if (mean_out) mean_out[b * groups + g] = meanf;
        if (rstd_out) rstd_out[b * groups + g] = rstdf;
    }
    for (int k = tid; k < cpg; k += 256) {
        const int c = cbeg + k;
        gn_affine(mean, rstdf, gamma[c], beta[c], oscale, legacy, scale[(size_t)b * ctot + c], shift[(size_t)b * ctot + c]);
    }
}

#ifdef CDX_TUNING
// tuning build: CDX_DIAG bit 4096 restores the round-3 float32 shift (tools/diag_scale.py: per-layer error, before / after)
inline bool gn_legacy() {
    static const bool on = [] { const char* e = getenv("CDX_DIAG"); return e && (atoi(e) & 4096); }();
    return on;
}
#else
constexpr bool gn_legacy() { return false; }
#endif

}  // namespace

extern "C" size_t cdx_gn_finalize_f32_workspace(const cdx_gn_finalize_args*) { return 0; }

extern "C" int cdx_gn_finalize_f32(const cdx_gn_finalize_args* a, void*, size_t, cdx_stream_t stream) {
    CDX_REQUIRE(a && a->part0 && a->gamma && a->beta && a->scale && a->shift);
    CDX_REQUIRE(a->c0 > 0 && a->slots0 > 0 && a->c1 >= 0);
    CDX_REQUIRE((a->c1 == 0) == (a->part1 == nullptr));
    if (a->c1) CDX_REQUIRE(a->slots1 > 0);
    CDX_REQUIRE(a->batch > 0 && a->batch <= 65535 && a->hw > 0 && a->groups > 0 && (a->c0 + a->c1) % a->groups == 0);
    CDX_REQUIRE(a->out_exp >= -60 && a->out_exp <= 60);
    hipLaunchKernelGGL(gn_finalize2, dim3(a->groups, a->batch), dim3(256), 0, static_cast<hipStream_t>(stream), a->part0,
                       a->slots0, a->c0, a->part1, a->slots1, a->c1, a->groups, a->hw, a->eps, a->gamma, a->beta, a->scale,
                       a->shift, a->mean, a->rstd, ldexpf(1.f, a->out_exp), gn_legacy());
    return check_launch();
}

extern "C" size_t cdx_gn_stats_f32_workspace(const cdx_gn_stats_args* a) {
    if (!a || a->batch <= 0 || a->hw <= 0) return 0;
    return (size_t)a->batch * gn_nsplit(a->hw) * (a->c0 + a->c1) * 2 * sizeof(double);
}

extern "C" int cdx_gn_stats_f32(const cdx_gn_stats_args* a, void* ws, size_t ws_bytes, cdx_stream_t stream) {
    CDX_REQUIRE(a && a->src0 && a->gamma && a->beta && a->scale && a->shift);
    CDX_REQUIRE(a->c0 > 0 && a->c1 >= 0 && (a->c0 % 4) == 0 && (a->c1 % 4) == 0);
    CDX_REQUIRE((a->c1 == 0) == (a->src1 == nullptr));
    CDX_REQUIRE(a->batch > 0 && a->batch <= 65535 && a->hw > 0 && a->groups > 0);
    CDX_REQUIRE(a->out_exp >= -60 && a->out_exp <= 60);
    const int ctot = a->c0 + a->c1;
    CDX_REQUIRE(ctot % a->groups == 0);
    CDX_REQUIRE(aligned16(a->src0) && aligned16(a->src1) && aligned16(ws));
    if (!ws || ws_bytes < cdx_gn_stats_f32_workspace(a)) return CDX_ENOSPC;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nsplit = gn_nsplit(a->hw);
    double* part = static_cast<double*>(ws);
    hipLaunchKernelGGL(gn_partial, dim3(nsplit, a->batch), dim3(256), 0, st, a->src0, a->c0, a->hw, nsplit, part, 0, ctot);
    int rc = check_launch();
    if (rc) return rc;
    if (a->c1) {
        hipLaunchKernelGGL(gn_partial, dim3(nsplit, a->batch), dim3(256), 0, st, a->src1, a->c1, a->hw, nsplit, part,
                           a->c0, ctot);
        if ((rc = check_launch())) return rc;
    }
    hipLaunchKernelGGL(gn_finalize, dim3(a->groups, a->batch), dim3(64), 0, st, part, nsplit, ctot, a->groups, a->hw,
                       a->eps, a->gamma, a->beta, a->scale, a->shift, a->mean, a->rstd, ldexpf(1.f, a->out_exp), gn_legacy());
    return check_launch();
}
