// (f4) Image side of the bitstream path: what happens to the decoded float tensor.
//   cdx_export_u8     [-1, 1] float NHWC (the sampler's state buffer) -> 8-bit interleaved RGB rows (PPM / PNG scanline order)
//   cdx_psnr_f32      per-image PSNR of two NCHW float tensors (float64 sums, fixed order)
//   cdx_msssim_f32    per-image multi-scale SSIM (Wang, Simoncelli, Bovik 2003): 5 scales, 11 x 11 Gaussian window (sigma 1.5,
//                     "valid" support), 2 x 2 average pooling between scales, contrast-structure at every scale and luminance at
//                     the coarsest, mean over channels -- the definition oracle/metrics_ref.py restates with F.conv2d / F.avg_pool2d.
// All HBM-bound reductions: one pass over the images per scale, tile + halo through LDS, separable filter, float64 partial sums
// per workgroup summed in a fixed order by a one-workgroup-per-image finalize (no atomics: bitwise reproducible).
// No reference file exists to cite (the reference snapshot is empty); format and metric definitions are build-defined.
#include <math.h>

#include "common.h"

using namespace cdx;

namespace {

// ---------------------------------------------------------------- 8-bit export
__global__ __launch_bounds__(256) void export_u8_kernel(const float* __restrict__ x, int x_ld, long long pixels, int channels,
                                                        float lo, float hi, uint8_t* __restrict__ out) {
    const float sc = 255.0f / (hi - lo);
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < pixels; p += (long long)gridDim.x * blockDim.x) {
        const float* s = x + (size_t)p * x_ld;
        uint8_t* o = out + (size_t)p * channels;
        for (int c = 0; c < channels; ++c) {
            float v = (fminf(fmaxf(s[c], lo), hi) - lo) * sc;       // NaN -> lo (fmaxf returns the non-NaN operand)
            o[c] = (uint8_t)floorf(v + 0.5f);                       // round half up: the oracle's definition
        }
    }
}

// ---------------------------------------------------------------- PSNR
constexpr int kPsnrBlocks = 64;      // partial sums per image

__global__ __launch_bounds__(256) void sqdiff_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n,
                                                             double* __restrict__ part) {
    __shared__ double red[256];
    const int img = blockIdx.y;
    const float* __restrict__ pa = a + (size_t)img * n;
    const float* __restrict__ pb = b + (size_t)img * n;
    double s = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {      // fixed assignment
        const double d = (double)pa[i] - (double)pb[i];
        s = fma(d, d, s);
    }
    red[threadIdx.x] = s;
    __syncthreads();
#pragma unroll
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[(size_t)img * gridDim.x + blockIdx.x] = red[0];
}

__global__ void psnr_finalize_kernel(const double* __restrict__ part, int nblocks, long long n, float range, float* __restrict__ out) {
    const int img = blockIdx.x;
    if (threadIdx.x != 0) return;
    double s = 0;
    for (int k = 0; k < nblocks; ++k) s += part[(size_t)img * nblocks + k];
    const double mse = s / (double)n;
    out[img] = mse > 0 ? (float)(10.0 * log10((double)range * (double)range / mse)) : INFINITY;
}

// ---------------------------------------------------------------- MS-SSIM
constexpr int kWin = 11, kHalo = kWin - 1, kTile = 16, kIn = kTile + kHalo;      // 16 x 16 outputs from a 26 x 26 input patch
constexpr int kScales = 5;

struct Gauss { float w[kWin]; };

// One scale: per (image-channel plane, 16 x 16 output tile) the sums of ssim and cs over the tile's valid outputs.
__global__ __launch_bounds__(256) void ssim_scale_kernel(const float* __restrict__ x, const float* __restrict__ y, int H, int W, Gauss g,
                                                         float c1, float c2, double* __restrict__ part) {
    __shared__ float sx[kIn][kIn + 1], sy[kIn][kIn + 1];
    __shared__ float hq[5][kIn][kTile + 1];      // horizontally filtered x, y, xx, yy, xy
    __shared__ double red[2][256];
    const int Ho = H - kHalo, Wo = W - kHalo;
    const int tiles_x = (Wo + kTile - 1) / kTile;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int plane = blockIdx.y;
    const float* __restrict__ px = x + (size_t)plane * H * W;
    const float* __restrict__ py = y + (size_t)plane * H * W;
    const int y0 = ty * kTile, x0 = tx * kTile;
    const int tid = threadIdx.x;
    for (int i = tid; i < kIn * kIn; i += 256) {
        const int r = i / kIn, c = i - r * kIn;
        const int yy = y0 + r, xx = x0 + c;
        const bool ok = yy < H && xx < W;
        sx[r][c] = ok ? px[(size_t)yy * W + xx] : 0.f;
        sy[r][c] = ok ? py[(size_t)yy * W + xx] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < kIn * kTile; i += 256) {
        const int r = i / kTile, c = i - r * kTile;
        float a = 0, b = 0, aa = 0, bb = 0, ab = 0;
#pragma unroll
        for (int k = 0; k < kWin; ++k) {
            const float u = sx[r][c + k], v = sy[r][c + k], w = g.w[k];
            a = fmaf(w, u, a);
            b = fmaf(w, v, b);
            aa = fmaf(w, u * u, aa);
            bb = fmaf(w, v * v, bb);
            ab = fmaf(w, u * v, ab);
        }
        hq[0][r][c] = a; hq[1][r][c] = b; hq[2][r][c] = aa; hq[3][r][c] = bb; hq[4][r][c] = ab;
    }
    __syncthreads();
    const int r = tid >> 4, c = tid & 15;
    double ssim = 0, cs = 0;
    if (y0 + r < Ho && x0 + c < Wo) {
        float m[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < kWin; ++k) {
            const float w = g.w[k];
#pragma unroll
            for (int q = 0; q < 5; ++q) m[q] = fmaf(w, hq[q][r + k][c], m[q]);
        }
        const float mx = m[0], my = m[1];
        const float vx = m[2] - mx * mx, vy = m[3] - my * my, cxy = m[4] - mx * my;
        const float csv = (2.f * cxy + c2) / (vx + vy + c2);
        const float lum = (2.f * mx * my + c1) / (mx * mx + my * my + c1);
        cs = (double)csv;
        ssim = (double)(lum * csv);
    }
    red[0][tid] = ssim;
    red[1][tid] = cs;
    __syncthreads();
#pragma unroll
    for (int w = 128; w >= 1; w >>= 1) {
        if (tid < w) {
            red[0][tid] += red[0][tid + w];
            red[1][tid] += red[1][tid + w];
        }
        __syncthreads();
    }
    if (tid == 0) {
        double* o = part + ((size_t)plane * gridDim.x + blockIdx.x) * 2;
        o[0] = red[0][0];
        o[1] = red[1][0];
    }
}

// 2 x 2 average pooling (floor sizes) of both tensors, plane-wise
__global__ __launch_bounds__(256) void pool2_kernel(const float* __restrict__ x, const float* __restrict__ y, int H, int W,
                                                    float* __restrict__ ox, float* __restrict__ oy) {
    const int Ho = H / 2, Wo = W / 2;
    const int plane = blockIdx.y;
    const float* __restrict__ px = x + (size_t)plane * H * W;
    const float* __restrict__ py = y + (size_t)plane * H * W;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < Ho * Wo; i += gridDim.x * 256) {
        const int r = i / Wo, c = i - r * Wo;
        const size_t s = (size_t)(2 * r) * W + 2 * c;
        ox[(size_t)plane * Ho * Wo + i] = 0.25f * ((px[s] + px[s + 1]) + (px[s + W] + px[s + W + 1]));
        oy[(size_t)plane * Ho * Wo + i] = 0.25f * ((py[s] + py[s + 1]) + (py[s + W] + py[s + W + 1]));
    }
}

struct ScaleInfo { long long part_off; int blocks; int outs; };      // per scale: partial-sum offset, blocks and valid outputs per plane
struct MsInfo { ScaleInfo s[kScales]; };

// out[b] = prod_{j < 4} mcs_j^w_j * mssim_4^w_4, every factor the mean over the image's channels and pixels (clamped at 0 as the
// usual implementations do before the fractional power); one thread per image walks the partial sums in a fixed order
__global__ void msssim_finalize_kernel(const double* __restrict__ part, MsInfo info, int channels, float* __restrict__ out,
                                       float* __restrict__ per_scale) {
    const int img = blockIdx.x;
    if (threadIdx.x != 0) return;
    const double wts[kScales] = {0.0448, 0.2856, 0.3001, 0.2363, 0.1333};
    double result = 1.0;
    for (int j = 0; j < kScales; ++j) {
        double ss = 0, cs = 0;
        for (int c = 0; c < channels; ++c) {
            const double* p = part + (info.s[j].part_off + (size_t)(img * channels + c) * info.s[j].blocks) * 2;
            for (int k = 0; k < info.s[j].blocks; ++k) {
                ss += p[2 * k];
                cs += p[2 * k + 1];
            }
        }
        const double n = (double)info.s[j].outs * channels;
        const double v = (j == kScales - 1 ? ss : cs) / n;
        if (per_scale) per_scale[img * kScales + j] = (float)v;
        result *= pow(v > 0 ? v : 0.0, wts[j]);
    }
    out[img] = (float)result;
}

inline size_t align16(size_t n) { return (n + 15) & ~size_t(15); }

struct MsLayout {
    size_t pool_off[kScales];      // float offsets of the pooled x planes of scale j >= 1 (y follows x)
    size_t part_bytes_off;
    MsInfo info;
    size_t total;
    int H[kScales], W[kScales];
};

bool ms_layout(const cdx_msssim_args* a, MsLayout& L) {
    size_t off = 0;
    long long poff = 0;
    const size_t planes = (size_t)a->batch * a->channels;
    for (int j = 0; j < kScales; ++j) {
        L.H[j] = j ? L.H[j - 1] / 2 : a->h;
        L.W[j] = j ? L.W[j - 1] / 2 : a->w;
        if (L.H[j] < kWin || L.W[j] < kWin) return false;
        L.pool_off[j] = off;
        if (j) off += align16(2 * planes * L.H[j] * L.W[j] * sizeof(float));
        const int Ho = L.H[j] - kHalo, Wo = L.W[j] - kHalo;
        const int blocks = ((Ho + kTile - 1) / kTile) * ((Wo + kTile - 1) / kTile);
        L.info.s[j] = ScaleInfo{poff, blocks, Ho * Wo};
        poff += (long long)planes * blocks;
    }
    L.part_bytes_off = off;
    L.total = off + (size_t)poff * 2 * sizeof(double);
    return true;
}

}  // namespace

extern "C" size_t cdx_export_u8_workspace(const cdx_export_u8_args*) { return 0; }
extern "C" int cdx_export_u8(const cdx_export_u8_args* a, void*, size_t, cdx_stream_t stream) {
    CDX_REQUIRE(a && a->x && a->out && a->batch > 0 && a->hw > 0 && a->channels > 0 && a->x_ld >= a->channels && a->hi > a->lo);
    const long long pixels = (long long)a->batch * a->hw;
    long long blocks = (pixels + 255) / 256;
    blocks = blocks > 4096 ? 4096 : blocks;
    hipLaunchKernelGGL(export_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a->x, a->x_ld, pixels,
                       a->channels, a->lo, a->hi, a->out);
    return check_launch();
}

extern "C" size_t cdx_psnr_f32_workspace(const cdx_psnr_args* a) {
    return a && a->batch > 0 ? (size_t)a->batch * kPsnrBlocks * sizeof(double) : 0;
}
extern "C" int cdx_psnr_f32(const cdx_psnr_args* a, void* ws, size_t ws_bytes, cdx_stream_t stream) {
    CDX_REQUIRE(a && a->a && a->b && a->out && a->batch > 0 && a->batch <= 65535 && a->n > 0 && a->range > 0.f);
    if (!ws || ws_bytes < cdx_psnr_f32_workspace(a)) return CDX_ENOSPC;
    hipStream_t st = static_cast<hipStream_t>(stream);
    double* part = static_cast<double*>(ws);
    hipLaunchKernelGGL(sqdiff_partial_kernel, dim3(kPsnrBlocks, a->batch), dim3(256), 0, st, a->a, a->b, (long long)a->n, part);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(psnr_finalize_kernel, dim3(a->batch), dim3(64), 0, st, part, kPsnrBlocks, (long long)a->n, a->range, a->out);
    return check_launch();
}

extern "C" size_t cdx_msssim_f32_workspace(const cdx_msssim_args* a) {
    MsLayout L;
    if (!a || a->batch <= 0 || a->channels <= 0 || !ms_layout(a, L)) return 0;
    return L.total;
}
extern "C" int cdx_msssim_f32(const cdx_msssim_args* a, void* ws, size_t ws_bytes, cdx_stream_t stream) {
    CDX_REQUIRE(a && a->x && a->y && a->out && a->batch > 0 && a->channels > 0 && a->range > 0.f);
    CDX_REQUIRE((int64_t)a->batch * a->channels <= 65535);
    MsLayout L;
    if (!ms_layout(a, L)) return CDX_EINVAL;      // the coarsest scale must hold one 11 x 11 window: h, w >= 176
    if (!ws || ws_bytes < L.total || !aligned16(ws)) return CDX_ENOSPC;
    hipStream_t st = static_cast<hipStream_t>(stream);
    Gauss g;
    double sum = 0, e[kWin];
    for (int k = 0; k < kWin; ++k) {
        const double d = k - (kWin - 1) / 2;
        e[k] = exp(-d * d / (2.0 * 1.5 * 1.5));
        sum += e[k];
    }
    for (int k = 0; k < kWin; ++k) g.w[k] = (float)(e[k] / sum);
    const float c1 = (0.01f * a->range) * (0.01f * a->range), c2 = (0.03f * a->range) * (0.03f * a->range);
    const int planes = a->batch * a->channels;
    char* base = static_cast<char*>(ws);
    double* part = reinterpret_cast<double*>(base + L.part_bytes_off);
    const float *x = a->x, *y = a->y;
    for (int j = 0; j < kScales; ++j) {
        if (j) {
            float* ox = reinterpret_cast<float*>(base + L.pool_off[j]);
            float* oy = ox + (size_t)planes * L.H[j] * L.W[j];
            const int n = L.H[j] * L.W[j];
            hipLaunchKernelGGL(pool2_kernel, dim3((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256, planes), dim3(256), 0, st, x, y,
                               L.H[j - 1], L.W[j - 1], ox, oy);
            int rc = check_launch();
            if (rc) return rc;
            x = ox;
            y = oy;
        }
        hipLaunchKernelGGL(ssim_scale_kernel, dim3(L.info.s[j].blocks, planes), dim3(256), 0, st, x, y, L.H[j], L.W[j], g, c1, c2,
                           part + L.info.s[j].part_off * 2);
        int rc = check_launch();
        if (rc) return rc;
    }
    hipLaunchKernelGGL(msssim_finalize_kernel, dim3(a->batch), dim3(64), 0, st, part, L.info, a->channels, a->out, a->per_scale);
    return check_launch();
}
