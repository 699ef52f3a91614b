// S3/S4/U1/U8 and the sampler's entry / exit copies: HBM-bound pointwise kernels.
// No reference file exists to cite (reference snapshot is empty); definitions: schedule.py, rng.py.
#include "common.h"

// Keep a*b + c as separate IEEE operations so the update matches a plain mul/add evaluation
// (torch elementwise ops) bit for bit.
#pragma clang fp contract(off)

using namespace cdx;

namespace {

// float32(sqrt(-2 ln u1) cos(2 pi u2)) evaluated in float64 (rng.py: normal()).
__device__ __forceinline__ float normal_from_hash(uint64_t h) {
    const double u1 = ((double)(uint32_t)(h >> 32) + 0.5) * (1.0 / 4294967296.0);
    const double u2 = ((double)(uint32_t)(h & 0xFFFFFFFFull) + 0.5) * (1.0 / 4294967296.0);
    return (float)(sqrt(-2.0 * log(u1)) * cos(2.0 * 3.14159265358979323846 * u2));
}

__global__ __launch_bounds__(256) void gauss_fill_kernel(float* __restrict__ x, int x_ld, int hw, int channels,
                                                         uint64_t seed, int64_t first_image, int stream) {
    const int b = blockIdx.y;
    const uint64_t key = stream_key(seed, (uint64_t)(first_image + b), (uint64_t)stream);
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += gridDim.x * blockDim.x) {
        float* o = x + ((size_t)b * hw + p) * x_ld;
        for (int c = 0; c < channels; ++c) {
            const uint64_t idx = (uint64_t)c * hw + p;
            o[c] = normal_from_hash(mix64(key + (idx + 1) * kGold));
        }
    }
}

struct UpdateCoef {
    float ca, cb, cx, c0, ce, sigma;
    int clip;
};

__global__ __launch_bounds__(256) void diffusion_update_kernel(float* __restrict__ x, int x_ld,
                                                               const float* __restrict__ eps, int eps_ld, int hw,
                                                               int channels, UpdateCoef k, uint64_t seed,
                                                               int64_t first_image, int stream) {
    const int b = blockIdx.y;
    const uint64_t key = k.sigma != 0.f ? stream_key(seed, (uint64_t)(first_image + b), (uint64_t)stream) : 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += gridDim.x * blockDim.x) {
        float* xo = x + ((size_t)b * hw + p) * x_ld;
        const float* e = eps + ((size_t)b * hw + p) * eps_ld;
        for (int c = 0; c < channels; ++c) {
            const float xv = xo[c], ev = e[c];
            float x0 = k.ca * xv + k.cb * ev;
            if (k.clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);
            float v = k.cx * xv + k.c0 * x0 + k.ce * ev;
            if (k.sigma != 0.f) {
                const uint64_t idx = (uint64_t)c * hw + p;
                v += k.sigma * normal_from_hash(mix64(key + (idx + 1) * kGold));
            }
            xo[c] = v;
        }
    }
}

__global__ __launch_bounds__(256) void cond_embed_kernel(const float* __restrict__ cond, int cc, int hc, int wc,
                                                         float* __restrict__ x, int x_ld, int c_off, int h, int w) {
    const int b = blockIdx.y;
    const int hw = h * w;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += gridDim.x * blockDim.x) {
        const int y = p / w, xx = p - y * w;
        // F.interpolate(mode="nearest"): src = floor(dst * in / out)
        const int sy = min((int)(((int64_t)y * hc) / h), hc - 1), sx = min((int)(((int64_t)xx * wc) / w), wc - 1);
        float* o = x + ((size_t)b * hw + p) * x_ld;
        for (int c = 0; c < cc; ++c) o[c_off + c] = cond[(((size_t)b * cc + c) * hc + sy) * wc + sx];
        for (int c = c_off + cc; c < x_ld; ++c) o[c] = 0.f;
    }
}

__global__ __launch_bounds__(256) void export_image_kernel(const float* __restrict__ x, int x_ld, int hw, int channels,
                                                           float lo, float hi, float* __restrict__ out) {
    const int b = blockIdx.y;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += gridDim.x * blockDim.x) {
        const float* xi = x + ((size_t)b * hw + p) * x_ld;
        for (int c = 0; c < channels; ++c)
            out[((size_t)b * channels + c) * hw + p] = fminf(fmaxf(xi[c], lo), hi);
    }
}

__global__ __launch_bounds__(256) void timestep_embedding_kernel(const int32_t* __restrict__ t, int batch, int dim,
                                                                 float* __restrict__ out) {
    const int half = dim / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch * half) return;
    const int b = i / half, k = i - b * half;
    const double f = exp(-9.210340371976184 /* ln 1e4 */ * (double)k / (double)(half - 1));
    const double arg = (double)t[b] * f;
    out[(size_t)b * dim + k] = (float)sin(arg);
    out[(size_t)b * dim + half + k] = (float)cos(arg);
}

// S5: per output pixel, the (at most 2 x 2) tiles that cover it, separable linear ramps over the overlaps.
__device__ __forceinline__ float ramp_weight(int u, int tile, int ov_lo, int ov_hi) {
    const float a = fminf(1.f, ((float)u + 0.5f) / (float)ov_lo);
    const float b = fminf(1.f, ((float)(tile - u) - 0.5f) / (float)ov_hi);
    return a * b;
}

__global__ __launch_bounds__(256) void tile_blend_kernel(const float* __restrict__ tiles, int channels, int tile, int ny,
                                                         int nx, const int* __restrict__ y0, const int* __restrict__ x0,
                                                         int h, int w, float* __restrict__ out) {
    const int b = blockIdx.y;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < h * w; p += gridDim.x * blockDim.x) {
        const int y = p / w, x = p - y * w;
        float acc[4] = {0.f, 0.f, 0.f, 0.f}, wsum = 0.f;      // channels <= 4
        for (int iy = 0; iy < ny; ++iy) {
            const int u = y - y0[iy];
            if (u < 0 || u >= tile) continue;
            const int ovl = iy > 0 ? max(1, y0[iy - 1] + tile - y0[iy]) : 1, ovh = iy + 1 < ny ? max(1, y0[iy] + tile - y0[iy + 1]) : 1;
            const float wy = ramp_weight(u, tile, ovl, ovh);
            for (int ix = 0; ix < nx; ++ix) {
                const int v = x - x0[ix];
                if (v < 0 || v >= tile) continue;
                const int oxl = ix > 0 ? max(1, x0[ix - 1] + tile - x0[ix]) : 1, oxh = ix + 1 < nx ? max(1, x0[ix] + tile - x0[ix + 1]) : 1;
                const float wt = wy * ramp_weight(v, tile, oxl, oxh);
                const float* t = tiles + ((((size_t)b * ny + iy) * nx + ix) * channels) * tile * tile + (size_t)u * tile + v;
                for (int c = 0; c < channels; ++c) acc[c] += wt * t[(size_t)c * tile * tile];
                wsum += wt;
            }
        }
        for (int c = 0; c < channels; ++c) out[((size_t)b * channels + c) * h * w + p] = acc[c] / wsum;
    }
}

inline dim3 pixel_grid(int hw, int batch) {
    int gx = (hw + 255) / 256;
    if (gx > 2048) gx = 2048;
    return dim3(gx, batch);
}

}  // namespace

extern "C" size_t cdx_gauss_fill_f32_workspace(const cdx_gauss_fill_args*) { return 0; }
extern "C" int cdx_gauss_fill_f32(const cdx_gauss_fill_args* a, void*, size_t, cdx_stream_t stream) {
    CDX_REQUIRE(a && a->x && a->batch > 0 && a->batch <= 65535 && a->hw > 0 && a->channels > 0 && a->x_ld >= a->channels);
    hipLaunchKernelGGL(gauss_fill_kernel, pixel_grid(a->hw, a->batch), dim3(256), 0, static_cast<hipStream_t>(stream),
                       a->x, a->x_ld, a->hw, a->channels, a->seed, a->first_image, a->noise_stream);
    return check_launch();
}

extern "C" size_t cdx_diffusion_update_f32_workspace(const cdx_diffusion_update_args*) { return 0; }
extern "C" int cdx_diffusion_update_f32(const cdx_diffusion_update_args* a, void*, size_t, cdx_stream_t stream) {
    CDX_REQUIRE(a && a->x && a->eps && a->batch > 0 && a->batch <= 65535 && a->hw > 0 && a->channels > 0);
    CDX_REQUIRE(a->x_ld >= a->channels && a->eps_ld >= a->channels);
    UpdateCoef k{a->ca, a->cb, a->cx, a->c0, a->ce, a->sigma, a->clip_x0};
    hipLaunchKernelGGL(diffusion_update_kernel, pixel_grid(a->hw, a->batch), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a->x, a->x_ld, a->eps, a->eps_ld, a->hw, a->channels, k,
                       a->seed, a->first_image, a->noise_stream);
    return check_launch();
}

extern "C" size_t cdx_cond_embed_f32_workspace(const cdx_cond_embed_args*) { return 0; }
extern "C" int cdx_cond_embed_f32(const cdx_cond_embed_args* a, void*, size_t, cdx_stream_t stream) {
    CDX_REQUIRE(a && a->x && a->batch > 0 && a->batch <= 65535 && a->h > 0 && a->w > 0 && a->cc >= 0);
    CDX_REQUIRE(a->cc == 0 || (a->cond && a->hc > 0 && a->wc > 0));
    CDX_REQUIRE(a->c_off >= 0 && a->c_off + a->cc <= a->x_ld);
    hipLaunchKernelGGL(cond_embed_kernel, pixel_grid(a->h * a->w, a->batch), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a->cond, a->cc, a->hc, a->wc, a->x, a->x_ld, a->c_off, a->h, a->w);
    return check_launch();
}

extern "C" size_t cdx_export_image_f32_workspace(const cdx_export_image_args*) { return 0; }
extern "C" int cdx_export_image_f32(const cdx_export_image_args* a, void*, size_t, cdx_stream_t stream) {
    CDX_REQUIRE(a && a->x && a->out && a->batch > 0 && a->batch <= 65535 && a->hw > 0 && a->channels > 0 && a->x_ld >= a->channels);
    hipLaunchKernelGGL(export_image_kernel, pixel_grid(a->hw, a->batch), dim3(256), 0, static_cast<hipStream_t>(stream),
                       a->x, a->x_ld, a->hw, a->channels, a->lo, a->hi, a->out);
    return check_launch();
}

extern "C" size_t cdx_timestep_embedding_f32_workspace(const cdx_timestep_embedding_args*) { return 0; }
extern "C" int cdx_timestep_embedding_f32(const cdx_timestep_embedding_args* a, void*, size_t, cdx_stream_t stream) {
    CDX_REQUIRE(a && a->t && a->out && a->batch > 0 && a->dim >= 4 && (a->dim % 2) == 0);
    const int n = a->batch * (a->dim / 2);
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       a->t, a->batch, a->dim, a->out);
    return check_launch();
}

extern "C" size_t cdx_tile_blend_f32_workspace(const cdx_tile_blend_args*) { return 0; }
extern "C" int cdx_tile_blend_f32(const cdx_tile_blend_args* a, void*, size_t, cdx_stream_t stream) {
    CDX_REQUIRE(a && a->tiles && a->y0 && a->x0 && a->out);
    CDX_REQUIRE(a->batch > 0 && a->batch <= 65535 && a->channels > 0 && a->channels <= 4 && a->tile > 0);
    CDX_REQUIRE(a->ny > 0 && a->nx > 0 && a->h >= a->tile && a->w >= a->tile);
    hipLaunchKernelGGL(tile_blend_kernel, pixel_grid(a->h * a->w, a->batch), dim3(256), 0, static_cast<hipStream_t>(stream),
                       a->tiles, a->channels, a->tile, a->ny, a->nx, a->y0, a->x0, a->h, a->w, a->out);
    return check_launch();
}
