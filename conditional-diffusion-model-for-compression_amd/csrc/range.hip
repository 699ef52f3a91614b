// Range bookkeeping of the split-fp16 convolution tiles (cdx.h "RANGE CONTRACT") and the debug checks of the Python host:
//   cdx_amax_f32         per-image max |x| of a tensor no convolution of this library produced (atomic max-combine into the
//                        same [batch] words the convolution epilogues write through cdx_conv_args.amax_out)
//   cdx_fill_u32         zeroes those words (and status words) at the start of a forward pass
//   cdx_check_finite_f32 sets a status word when a tensor holds NaN / Inf (or exceeds a magnitude limit)
//   cdx_gn_act_exp       HOST: the static activation exponent of a GroupNorm-ed convolution input
// HBM-bound, one pass, float4 loads.  No reference file exists to cite (the reference snapshot is empty).
#include <math.h>

#include "common.h"

using namespace cdx;

namespace {


// max |x| over channels [0, channels) of rows [0, n) of image blockIdx.y; NaNs skipped (fmaxf), Inf kept.  Dense rows (x_ld ==
// channels: every tensor of the UNet but the padded x_t | cond buffer) are walked as ONE flat float4 stream -- the row / column
// split of the general case costs a 64-bit division per element (measured 45 us for 25 MB where the flat form takes ~10).
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, int x_ld, int n, int channels,
                                                   unsigned* __restrict__ out) {
    const int b = blockIdx.y;
    const float* __restrict__ base = x + (size_t)b * n * x_ld;
    float am = 0.f;
    auto take = [&](const f32x4 v) { am = fmaxf(fmaxf(am, fabsf(v[0])), fmaxf(fabsf(v[1]), fmaxf(fabsf(v[2]), fabsf(v[3])))); };
    if (x_ld == channels) {
        const size_t total = (size_t)n * (channels >> 2);
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
            take(*reinterpret_cast<const f32x4*>(base + i * 4));
    } else {
        const unsigned nq = (unsigned)channels >> 2;
        const unsigned total = (unsigned)n * nq;                  // (host: n * channels / 4 < 2^32)
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
            const unsigned row = i / nq, qd = i - row * nq;
            take(*reinterpret_cast<const f32x4*>(base + (size_t)row * x_ld + qd * 4));
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) am = fmaxf(am, __shfl_xor(am, off));
    if ((threadIdx.x & 63) == 0 && am != 0.f)      // spread over the image's words (same-address atomics serialise in L2)
        atomicMax(out + (size_t)b * CDX_AMAX_WORDS + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & (CDX_AMAX_WORDS - 1)), __float_as_uint(am));
}

__global__ __launch_bounds__(256) void fill_u32_kernel(unsigned* __restrict__ x, long long n, unsigned value) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) x[i] = value;
}

__global__ __launch_bounds__(256) void check_finite_kernel(const float* __restrict__ x, int x_ld, long long rows, int channels,
                                                           float limit, int* __restrict__ status) {
    int bad = 0;
    const long long total = rows * channels;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / channels;
        const float v = x[(size_t)row * x_ld + (int)(i - row * channels)];
        if (!(fabsf(v) <= 3.402823466e38f)) bad |= 1;
        else if (limit > 0.f && fabsf(v) > limit) bad |= 2;
    }
    if (bad) atomicOr(status, bad);
}

}  // namespace

namespace cdx {
// used by cdx_conv_f32 for the tile shapes whose epilogue does not produce amax_out itself
int amax_launch(const float* x, int x_ld, int batch, int n, int channels, unsigned* out, hipStream_t stream) {
    const long long work = (long long)n * (channels >> 2);
    long long blocks = (work + 256 * 8 - 1) / (256 * 8);      // ~8 float4 per thread
    blocks = blocks < 1 ? 1 : blocks > 4096 ? 4096 : blocks;
    hipLaunchKernelGGL(amax_kernel, dim3((unsigned)blocks, batch), dim3(256), 0, stream, x, x_ld, n, channels, out);
    return check_launch();
}
}  // namespace cdx

extern "C" size_t cdx_amax_f32_workspace(const cdx_amax_args*) { return 0; }
extern "C" int cdx_amax_f32(const cdx_amax_args* a, void*, size_t, cdx_stream_t stream) {
    CDX_REQUIRE(a && a->x && a->out && aligned16(a->x) && (reinterpret_cast<uintptr_t>(a->out) & 63u) == 0);
    CDX_REQUIRE(a->batch > 0 && a->batch <= 65535 && a->n > 0 && a->channels > 0 && (a->channels % 4) == 0);
    CDX_REQUIRE(a->x_ld >= a->channels && (a->x_ld % 4) == 0);
    CDX_REQUIRE((int64_t)a->n * (a->channels / 4) < (1ll << 32));
    return amax_launch(a->x, a->x_ld, a->batch, a->n, a->channels, a->out, static_cast<hipStream_t>(stream));
}

extern "C" size_t cdx_fill_u32_workspace(const cdx_fill_u32_args*) { return 0; }
extern "C" int cdx_fill_u32(const cdx_fill_u32_args* a, void*, size_t, cdx_stream_t stream) {
    CDX_REQUIRE(a && a->x && a->n > 0);
    long long blocks = (a->n + 255) / 256;
    blocks = blocks > 1024 ? 1024 : blocks;
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a->x, (long long)a->n, a->value);
    return check_launch();
}

extern "C" size_t cdx_check_finite_f32_workspace(const cdx_check_finite_args*) { return 0; }
extern "C" int cdx_check_finite_f32(const cdx_check_finite_args* a, void*, size_t, cdx_stream_t stream) {
    CDX_REQUIRE(a && a->x && a->status && a->rows > 0 && a->channels > 0 && a->x_ld >= a->channels);
    CDX_REQUIRE(a->limit >= 0.f);
    long long blocks = (a->rows * a->channels + 256 * 8 - 1) / (256 * 8);
    blocks = blocks < 1 ? 1 : blocks > 4096 ? 4096 : blocks;
    hipLaunchKernelGGL(check_finite_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a->x, a->x_ld,
                       (long long)a->rows, a->channels, a->limit, a->status);
    return check_launch();
}

// HOST.  bound = max |gamma| sqrt(n) + max |beta| >= every value GroupNorm (+ SiLU) can produce; largest e with bound 2^e < 2^15.
extern "C" int32_t cdx_gn_act_exp(const float* gamma, const float* beta, int32_t channels, int32_t groups, int32_t hw) {
    if (!gamma || !beta || channels <= 0 || groups <= 0 || hw <= 0 || channels % groups) return 0;
    double g = 0, bmax = 0;
    for (int c = 0; c < channels; ++c) {
        g = fmax(g, fabs((double)gamma[c]));
        bmax = fmax(bmax, fabs((double)beta[c]));
    }
    const double bound = g * sqrt((double)(channels / groups) * hw) + bmax;
    if (!(bound > 0.0) || !(bound < 1e300)) return 0;
    int ex;
    frexp(bound, &ex);                 // bound = m 2^ex, m in [0.5, 1)  ->  bound < 2^ex
    int e = 15 - ex;
    return e > 60 ? 60 : e < -60 ? -60 : e;
}
