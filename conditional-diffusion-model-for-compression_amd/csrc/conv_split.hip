// SPLIT mode of the 16-bit convolution kernel (conv16_kernel.h): float32 convolution on the fp16 matrix pipe with hi | lo split
// operands, reached through cdx_conv_f32 (conv.hip selects it as CDX_TILE_SPLIT).  Host packer + launch.
// (A translation unit of its own: the kernel template is instantiated ~50 times here and in conv16.hip.)
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "conv_kpar_kernel.h"
#ifdef CDX_TUNING
#include "conv_wsp_kernel.h"      // the persistent form (a measured negative): tuning build only
#endif

using namespace cdx;

namespace {
inline int chunks_of(int c) { return (c + 31) / 32; }
}  // namespace

// ------------------------------------------------------------------------------------------------------------
// SPLIT mode (conv16_kernel.h): float32 convolution on the fp16 matrix pipe, reached through cdx_conv_f32 (conv.hip).
// Host: fp16 hi/lo fragment image of the weights, pre-scaled by 2^s with s chosen per layer so that max |w| 2^s lies in
// [2^13, 2^14): hi = fp16(w 2^s), lo = fp16(w 2^s - hi) are then normal numbers for every weight above 2^-27 of the
// largest.  Layout [ntile][chunk][tap][j = 0..1][plane = hi, lo][lane][8 halves], + 16 KiB zero pad.
extern "C" size_t cdx_conv_split_packed_halves(int32_t c0, int32_t c1, int32_t cout, int32_t ksize) {
    if (c0 <= 0 || c1 < 0 || cout <= 0 || (ksize != 1 && ksize != 3)) return 0;
    const size_t ntiles = (cout + 31) / 32, nch = chunks_of(c0) + chunks_of(c1);
    return ntiles * nch * ksize * ksize * 2048 + 8192;
}

namespace {
// fp16 hi | lo fragment image of OIHW weights with a k x k kernel (k = 1, 2, 3), scaled by the power of two that puts max |w| in [2^13, 2^14)
int pack_split(const float* w, int c0, int c1, int cout, int ksize, _Float16* o, float* unscale) {
    const int taps = ksize * ksize, ctot = c0 + c1;
    const int nch0 = chunks_of(c0), nch = nch0 + chunks_of(c1), ntiles = (cout + 31) / 32;
    float wmax = 0.f;
    for (size_t i = 0; i < (size_t)cout * ctot * taps; ++i) {
        if (!(w[i] == w[i] && w[i] - w[i] == 0.f)) return CDX_EINVAL;      // finite
        wmax = fmaxf(wmax, fabsf(w[i]));
    }
    int e = 0;
    if (wmax > 0.f) {
        frexpf(wmax, &e);                  // wmax = m 2^e, m in [0.5, 1)
        e = 14 - e;                        // wmax 2^e in [2^13, 2^14)
        if (e > 100) e = 100;
        if (e < -100) e = -100;
    }
    const float sc = ldexpf(1.f, e);
    *unscale = ldexpf(1.f, -e);
    for (int nt = 0; nt < ntiles; ++nt)
        for (int ch = 0; ch < nch; ++ch)
            for (int tap = 0; tap < taps; ++tap)
                for (int j = 0; j < 2; ++j)
                    for (int plane = 0; plane < 2; ++plane)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int k = 0; k < 8; ++k) {
                                const int n = nt * 32 + (lane & 31);
                                const int cl = (ch < nch0 ? ch : ch - nch0) * 32 + 16 * j + 8 * (lane >> 5) + k;
                                const int csrc = ch < nch0 ? c0 : c1;
                                float v = 0.f;
                                if (n < cout && cl < csrc) v = w[((size_t)n * ctot + (ch < nch0 ? 0 : c0) + cl) * taps + tap] * sc;
                                const _Float16 hi = (_Float16)v;
                                *o++ = plane == 0 ? hi : (_Float16)(v - (float)hi);
                            }
    memset(o, 0, 8192 * sizeof(_Float16));
    return CDX_OK;
}
}  // namespace

extern "C" int cdx_conv_pack_weights_split_f16(const float* w, int32_t c0, int32_t c1, int32_t cout, int32_t ksize,
                                               cdx_half* packed, float* unscale) {
    CDX_REQUIRE(w && packed && unscale && c0 > 0 && c1 >= 0 && cout > 0 && (ksize == 1 || ksize == 3));
    return pack_split(w, c0, c1, cout, ksize, reinterpret_cast<_Float16*>(packed), unscale);
}

// ------------------------------------------------------------------------------------------------------------
// 3x3 convolution AFTER nearest-2x upsampling = four 2x2 convolutions on the LOW-resolution source, one per output phase
// (dy, dx): output row 2y + dy reads upsampled rows 2y + dy - 1 + ky, i.e. source rows y - 1, y, y (dy = 0) or y, y, y + 1
// (dy = 1) -- taps that read the same source pixel merge: K(0,0) = {0}, K(0,1) = {1, 2}, K(1,0) = {0, 1}, K(1,1) = {2} per axis.
// 16 tap-passes over N_lo pixels instead of 9 over 4 N_lo: 2.25x fewer MFMAs, and the source is read at low resolution.
// The merged weights are summed in float64 and rounded once to float32 before the hi | lo split (a different rounding of the
// same sum as F.conv2d's: the tolerance does not move).  Image: [phase = 2 dy + dx][ksize-2 split image], unscale per phase.
extern "C" size_t cdx_conv_split_up_packed_halves(int32_t c0, int32_t c1, int32_t cout) {
    if (c0 <= 0 || c1 < 0 || cout <= 0) return 0;
    const size_t ntiles = (cout + 31) / 32, nch = chunks_of(c0) + chunks_of(c1);
    return 4 * (ntiles * nch * 4 * 2048 + 8192);
}

extern "C" int cdx_conv_pack_weights_split_up_f16(const float* w, int32_t c0, int32_t c1, int32_t cout, cdx_half* packed, float* unscale4) {
    CDX_REQUIRE(w && packed && unscale4 && c0 > 0 && c1 >= 0 && cout > 0);
    const int ctot = c0 + c1;
    const size_t per = cdx_conv_split_up_packed_halves(c0, c1, cout) / 4;
    float* w2 = static_cast<float*>(malloc((size_t)cout * ctot * 4 * sizeof(float)));
    CDX_REQUIRE(w2);
    static const int klo[2][2] = {{0, 1}, {0, 2}}, khi[2][2] = {{0, 2}, {1, 2}};      // [d][t]: taps K(d, t) = klo .. khi
    int rc = CDX_OK;
    for (int dy = 0; dy < 2 && rc == CDX_OK; ++dy)
        for (int dx = 0; dx < 2 && rc == CDX_OK; ++dx) {
            for (size_t nc = 0; nc < (size_t)cout * ctot; ++nc)
                for (int ty = 0; ty < 2; ++ty)
                    for (int tx = 0; tx < 2; ++tx) {
                        double acc = 0;
                        for (int ky = klo[dy][ty]; ky <= khi[dy][ty]; ++ky)
                            for (int kx = klo[dx][tx]; kx <= khi[dx][tx]; ++kx) acc += (double)w[nc * 9 + ky * 3 + kx];
                        w2[nc * 4 + ty * 2 + tx] = (float)acc;
                    }
            rc = pack_split(w2, c0, c1, cout, 2, reinterpret_cast<_Float16*>(packed) + (size_t)(2 * dy + dx) * per, unscale4 + 2 * dy + dx);
        }
    free(w2);
    return rc;
}

namespace cdx {
#ifdef CDX_TUNING
// CUs of the current device: the persistent kernel launches one workgroup per CU (tuning build only; queried once)
int wsp_cu_count() {
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        return cus;
    }();
    return n;
}
// CDX_DIAG bit 2048: bias + temb through the accumulator init as in round 3 (tools/diag_scale.py: per-layer error, before / after)
static int diag_bits() {
    static const int v = [] { const char* e = getenv("CDX_DIAG"); return e ? atoi(e) : 0; }();
    return v;
}
#endif

// Is this cdx_conv_f32 launch one the SPLIT kernel is built for?  (conv.hip asks before choosing the tile.)
bool conv_split_ok(const cdx_conv_args* a) {
    if (!a->wpacked_split || !aligned16(a->wpacked_split) || !(a->wsplit_unscale > 0.f)) return false;
    // RANGE CONTRACT (cdx.h): the staged activations need a power-of-two exponent -- static for GroupNorm-ed inputs (gn_exp),
    // per image from the producers' amax words otherwise.  A launch that brings neither runs on the f32-input MFMA kernels.
    if (!(a->flags & CDX_CONV_GN) && (!a->src_amax0 || (a->c1 > 0 && !a->src_amax1))) return false;
    // (... and a GroupNorm-ed launch must STATE its exponent: the plain scale / shift pair -- gn_exp 0 by default -- would be staged at
    // unit scale with no clamp, so a large |gamma| could leave the fp16 range where the f32 tiles are exact: ADVICE r03)
    if ((a->flags & CDX_CONV_GN) && !(a->flags & CDX_CONV_GN_EXP)) return false;
    // (below 8 pixels wide the f32-MFMA split-K tiles stay)
    if (a->wout < 8 || a->cout <= 4) return false;
    if (a->stride == 2 && a->wout < 16) return false;
    if ((a->c0 % 8) != 0 || (a->c1 % 8) != 0) return false;                           // the loader moves 8-channel octets
    if (a->stride == 2 && a->ksize != 3) return false;
    if ((a->out_ld % 4) != 0 || a->out_ld < ((a->cout + 3) & ~3)) return false;      // outputs move as 4-channel vectors
    if ((a->residual || a->stats_out) && (a->cout % 4) != 0) return false;
    // 32-bit byte offsets inside ONE image of a source (the kernel rebases its buffer resource per image)
    const int cmax = a->c0 > a->c1 ? a->c0 : a->c1;
    if ((int64_t)a->hin * a->win * cmax * 4 >= (1ll << 31)) return false;
    return true;
}

// The four-phase form of an upsampled 3x3 layer (above): the phase image is given, the LOW-resolution rows hold a 32-pixel tile
// row, no residual (the phase launches keep the plain accumulator init: an up-sampling layer has none in this model family)
bool conv_split_up_ok(const cdx_conv_args* a) {
    return (a->flags & CDX_CONV_UPSAMPLE2X) && a->ksize == 3 && a->stride == 1 && a->wpacked_split_up && aligned16(a->wpacked_split_up) &&
           a->win >= 32 && !a->residual && a->wsplit_up_unscale[0] > 0.f && a->wsplit_up_unscale[1] > 0.f && a->wsplit_up_unscale[2] > 0.f &&
           a->wsplit_up_unscale[3] > 0.f;
}

// Spatial tiling of a SPLIT launch (also asked by cdx_conv_stats_slots).  3x3 stride-1 layers at >= 32 pixels wide use an 8 x 16-pixel
// tile: its 10 x 18 halo is 180 pixels = 3 staging passes of 64, where the 4 x 32 tile's 6 x 34 halo is 204 pixels = 4 passes with 52
// empty slots -- a quarter less staging work for the producer waves and 12 % fewer halo bytes, bit-identical results (+1.2...2 %
// in-process, tools/conv_bench.py --tiles 11,109 before the switch).  1x1 layers (no halo) and the phase launches keep 4 x 32.
void conv_split_tile_shape(const cdx_conv_args* a, int& tw, int& th) {
    if (a->stride == 2) { tw = a->wout >= 32 ? 32 : 16; th = 64 / tw; return; }
    if (a->wout >= 32) { if (a->ksize == 3) { tw = 16; th = 8; } else { tw = 32; th = 4; } return; }
    if (a->wout >= 16) { tw = 16; th = 4; return; }
    tw = 8; th = 8;
}

// GroupNorm-sum slots per spatial tile of a SPLIT launch: 2 when the 128-pixel tile's last channel block runs 2 x 2
int conv_split_slots_per_tile(const cdx_conv_args* a) {
    if (a->stride == 1 && a->wout < 16) return 4;      // chunk-parallel tile (conv_kpar_kernel.h): one slot per finishing wave
    return conv16_tail_2x2(a->cout, (a->stride == 2 || a->wout < 32) ? 2 : 4) ? 2 : 1;
}

int conv_split_launch(const cdx_conv_args* a, hipStream_t stream, int variant) {
    Conv16Params p;
    p.src[0] = a->src0;
    p.src[1] = a->src1 ? a->src1 : a->src0;
    p.csrc[0] = a->c0;
    p.csrc[1] = a->c1 ? a->c1 : a->c0;
    p.src_f32 = 1;
    p.nchunk0 = chunks_of(a->c0);
    p.nchunks = p.nchunk0 + chunks_of(a->c1);
    p.ctot = a->c0 + a->c1;
    p.B = a->batch; p.Hin = a->hin; p.Win = a->win; p.Hout = a->hout; p.Wout = a->wout; p.Cout = a->cout;
    p.ups = (a->flags & CDX_CONV_UPSAMPLE2X) ? 1 : 0;
    p.gn = (a->flags & CDX_CONV_GN) ? 1 : 0;
    p.silu = (a->flags & CDX_CONV_SILU) ? 1 : 0;
    p.abl = 0;
#ifdef CDX_TUNING
    p.abl = diag_bits() & 2048;
#endif
    p.wunscale = a->wsplit_unscale;
    p.w = a->wpacked_split;
    p.bias = a->bias; p.gscale = a->gn_scale; p.gshift = a->gn_shift; p.temb = a->temb; p.temb_ld = a->temb_ld;
    p.residual = a->residual;
    p.out = a->out; p.out_f32 = 1; p.out_ld = a->out_ld; p.stats = a->stats_out;
    p.stats_wm = conv_split_slots_per_tile(a);
    p.ostep = 1; p.ody = p.odx = 0; p.pady = p.padx = a->ksize / 2; p.slot_base = 0; p.nslots_total = 0;
    p.act_exp = (a->flags & CDX_CONV_GN_EXP) ? a->gn_exp : 0;
    p.amax[0] = (a->flags & CDX_CONV_GN) ? nullptr : a->src_amax0;
    p.amax[1] = (a->flags & CDX_CONV_GN) || a->c1 == 0 ? nullptr : a->src_amax1;
    p.amax_out = a->amax_out;
    // tile: 128 pixels x 128 channels at >= 32 pixels wide; 64 x 128 for stride 2 and at 16 pixels wide; 64 pixels x 32
    // channels with the input chunks split over the waves at 8 pixels wide (conv_kpar_kernel.h: at batch 16 that level has
    // too few output pixels to fill 256 CUs with 128-channel tiles: 1.06 -> 0.49 ms per forward for its 3x3 layers)
    if (conv_split_up_ok(a)) {
        // four phase launches over the LOW-resolution pixel grid (128-pixel x 128-channel wave-specialised tiles, 2 x 2 taps)
        p.ups = 0;
        p.Hout = a->hin; p.Wout = a->win;
        p.tiles_x = ceil_div(a->win, 32);
        p.tiles_y = ceil_div(a->hin, 4);
        CDX_REQUIRE((int64_t)p.tiles_x * p.tiles_y * p.B < (1ll << 31));
        p.ostep = 2;
        p.nslots_total = 4 * p.tiles_x * p.tiles_y * p.stats_wm;
        const size_t per = cdx_conv_split_up_packed_halves(a->c0, a->c1, a->cout) / 4;
        for (int ph = 0; ph < 4; ++ph) {
            p.ody = ph >> 1; p.odx = ph & 1;
            p.pady = 1 - p.ody; p.padx = 1 - p.odx;
            p.slot_base = ph * p.tiles_x * p.tiles_y * p.stats_wm;
            p.w = a->wpacked_split_up + (size_t)ph * per;
            p.wunscale = a->wsplit_up_unscale[ph];
            const int rc = conv16_ws_launch<Conv16Cfg<2, 1, 5, 4, 2, 0, 1, 1, 0, 1>>(p, stream);
            if (rc) return rc;
        }
        return CDX_OK;
    }
    const int logtw = a->wout >= 32 ? 5 : a->wout >= 16 ? 4 : 3;
    int tw, th;
    conv_split_tile_shape(a, tw, th);
    p.tiles_x = ceil_div(a->wout, tw);
    p.tiles_y = ceil_div(a->hout, th);
    CDX_REQUIRE((int64_t)p.tiles_x * p.tiles_y * p.B < (1ll << 31));
    if (a->stride == 2) {      // 64 output pixels x 128 channels, one halo image (DB = 0)
        if (logtw == 5) return conv16_launch<Conv16Cfg<3, 2, 5, 2, 3, 0, 1, 0>>(p, stream);
        return conv16_launch<Conv16Cfg<3, 2, 4, 2, 3, 0, 1, 0>>(p, stream);
    }
    if (logtw == 4) {      // 16 pixels wide: 64 pixels x 128 channels (measured at 16^2 x 512, batch 16: 0.93 ms per forward for the
        // 3x3 layers against 0.98 with the chunk-parallel tile, whose waves each stage a whole chunk for 32 channels)
        // (wave-specialised like the 128-pixel tile: 256 workgroups of a batch-16 level at 16^2 are ONE per CU, so the 4 producer
        // waves are the only thing that can stage under the MFMAs -- same-box A/B of whole steps: cfg2 22.24 -> 22.15 ms)
        if (a->ksize == 3) return conv16_ws_launch<Conv16Cfg<3, 1, 4, 2, 3, 0, 1, 1, 0, 1>>(p, stream);
        return conv16_launch<Conv16Cfg<1, 1, 4, 2, 3, 0, 1>>(p, stream);
    }
    if (logtw == 3) {      // 8 pixels wide: 64 pixels x 32 channels per workgroup, the 4 waves split the input chunks
        if (a->ksize == 3) return conv_kpar_launch<KparCfg<3, 3>>(p, stream);
        return conv_kpar_launch<KparCfg<1, 3>>(p, stream);
    }
#ifdef CDX_TUNING
    if (a->ksize == 3 && variant == 50) return conv16_ws_launch<Conv16Cfg<3, 1, 4, 4, 3, 4096, 1, 1, 0, 1>>(p, stream);      // the shipped tile WITHOUT pipelined operand reads
    if (a->ksize == 3 && variant == 64) return conv16_ws_launch<Conv16Cfg<3, 1, 4, 4, 3, 65536, 1, 1, 0, 1>>(p, stream);     // the shipped tile WITHOUT the producers' prologue priority
    if (a->ksize == 3 && variant == 52) return conv16_ws_launch<Conv16Cfg<3, 1, 4, 4, 3, 8192, 1, 1, 0, 1>>(p, stream);      // the shipped tile in the XCD-contiguous workgroup order (measured, not shipped: conv16_kernel.h)
    if (a->ksize == 3 && variant == 51) return conv16_ws_launch<Conv16Cfg<3, 1, 4, 4, 3, 0, 1, 1, 0, 1>>(p, stream);         // the shipped tile (same code path as tile 11)
    if (a->ksize == 3 && variant >= 59 && variant <= 62) {      // stamped timing ablations of the shipped tile: what does the CLOCK do without ... (tools/ws_stamps.py --tile 119 .. 122)
        if (variant == 59) return conv16_ws_launch<Conv16Cfg<3, 1, 4, 4, 3, 552, 1, 1, 0, 1>>(p, stream);       // ... the LDS operand reads
        if (variant == 60) return conv16_ws_launch<Conv16Cfg<3, 1, 4, 4, 3, 548, 1, 1, 0, 1>>(p, stream);       // ... the weight refills
        if (variant == 61) return conv16_ws_launch<Conv16Cfg<3, 1, 4, 4, 3, 558, 1, 1, 0, 1>>(p, stream);       // ... all three (MFMAs + barriers + epilogue only)
        return conv16_ws_launch<Conv16Cfg<3, 1, 4, 4, 3, 545, 1, 1, 0, 1>>(p, stream);                           // ... the epilogue
    }
    if (a->ksize == 3 && (variant == 48 || variant == 49)) {      // the SHIPPED 8 x 16 tile with per-wave stamps / barrier accounting (tools/ws_stamps.py)
        if (variant == 48) return conv16_ws_launch<Conv16Cfg<3, 1, 4, 4, 3, 544, 1, 1, 0, 1>>(p, stream);
        return conv16_ws_launch<Conv16Cfg<3, 1, 4, 4, 3, 546, 1, 1, 0, 1>>(p, stream);       // ... producers stage only the first chunk
    }
    if (a->ksize == 3 && variant) {      // the tuning variants are 4 x 32-pixel tiles (round 2's geometry)
        p.tiles_x = ceil_div(a->wout, 32);
        p.tiles_y = ceil_div(a->hout, 4);
    }
    if (a->ksize == 1 && variant == 40) return conv16_ws_launch<Conv16Cfg<1, 1, 5, 4, 3, 0, 1, 1, 0, 1>>(p, stream);
    if (a->ksize == 3 && variant == 46) return conv16_launch<Conv16Cfg<3, 1, 5, 4, 3, 0, 1>>(p, stream);      // the 4-wave tile (round 2's product)
    if (a->ksize == 3 && variant) {      // timing ablations / tuning variants (tools/conv_bench.py --tiles 60..)
        if (variant >= 100) {            // + 100: the same variant (100 = the shipped tile) at ONE workgroup per CU (solo waves)
            p.abl |= 1024;
            variant -= 100;
        }
        switch (variant) {
            case 42: return conv_wsp_launch<WspCfg<3>>(p, stream);      // persistent wave-specialised (4 sum slots per tile)
            case 43: return conv_wsp_launch<WspCfg<3, 1>>(p, stream);   // ... MFMA waves' own bound
            case 44: return conv_wsp_launch<WspCfg<3, 2>>(p, stream);   // ... producers' own bound
            case 45: return conv_wsp_launch<WspCfg<3, 5>>(p, stream);   // ... MFMA waves without weight refills
            case 47: return conv_wsp_launch<WspCfg<3, 8>>(p, stream);   // ... producers at s_setprio 3
            case 40: return conv16_ws_launch<Conv16Cfg<3, 1, 5, 4, 3, 0, 1, 1, 0, 1>>(p, stream);      // wave-specialised: 4 MFMA + 4 producer waves
            case 41: return conv16_ws_launch<Conv16Cfg<3, 1, 5, 4, 2, 0, 1, 1, 0, 1>>(p, stream);      // ... ring depth 2
            case 0: return conv16_launch<Conv16Cfg<3, 1, 5, 4, 3, 0, 1>>(p, stream);
            case 1: return conv16_launch<Conv16Cfg<3, 1, 5, 4, 3, 1, 1>>(p, stream);      // no epilogue
            case 2: return conv16_launch<Conv16Cfg<3, 1, 5, 4, 3, 2, 1>>(p, stream);      // stage first chunk only
            case 3: return conv16_launch<Conv16Cfg<3, 1, 5, 4, 3, 3, 1>>(p, stream);
            case 4: return conv16_launch<Conv16Cfg<3, 1, 5, 4, 3, 4, 1>>(p, stream);      // no weight refills
            case 7: return conv16_launch<Conv16Cfg<3, 1, 5, 4, 3, 7, 1>>(p, stream);
            case 8: return conv16_launch<Conv16Cfg<3, 1, 5, 4, 3, 8, 1>>(p, stream);      // no LDS operand reads
            case 15: return conv16_launch<Conv16Cfg<3, 1, 5, 4, 3, 15, 1>>(p, stream);    // MFMA stream only
            case 20: return conv16_launch<Conv16Cfg<3, 1, 5, 4, 3, 16, 1>>(p, stream);    // no residual loads
            case 21: return conv16_launch<Conv16Cfg<3, 1, 5, 4, 3, 32, 1>>(p, stream);    // no GroupNorm sums
            case 22: return conv16_launch<Conv16Cfg<3, 1, 5, 4, 3, 48, 1>>(p, stream);    // neither
            case 23: return conv16_launch<Conv16Cfg<3, 1, 5, 4, 3, 64, 1>>(p, stream);    // halo loads of chunks 0, 1 only
            case 24: return conv16_launch<Conv16Cfg<3, 1, 5, 4, 3, 68, 1>>(p, stream);    // + no weight refills
            case 32: return conv16_launch<Conv16Cfg<3, 1, 5, 4, 3, 544, 1>>(p, stream);   // phase stamps into the stats buffer
            case 25: return conv16_launch<Conv16Cfg<3, 1, 5, 4, 3, 128, 1>>(p, stream);   // units not interleaved
            case 26: return conv16_launch<Conv16Cfg<3, 1, 5, 4, 3, 256, 1>>(p, stream);   // units computed, not stored
            case 27: return conv16_launch<Conv16Cfg<3, 1, 5, 4, 3, 320, 1>>(p, stream);   // ... and no halo loads after chunk 1
            case 16: return conv16_launch<Conv16Cfg<3, 1, 5, 4, 6, 0, 1>>(p, stream);     // ring depth 6
            case 17: return conv16_launch<Conv16Cfg<3, 1, 5, 4, 2, 0, 1>>(p, stream);     // ring depth 2
            default: return CDX_ENOTSUP;
        }
    }
#endif
    // 3x3 at >= 32 pixels wide (79 % of the cfg2 step): the WAVE-SPECIALISED workgroup -- 4 MFMA waves + 4 producer waves
    // (conv16_kernel.h WS): bit-identical to the 4-wave tile, +3...5 % (in-process A/B, profiles/r03_*); the HBM-bound 1x1
    // layers gain nothing from it and keep the 4-wave form
    if (a->ksize == 3) return conv16_ws_launch<Conv16Cfg<3, 1, 4, 4, 3, 0, 1, 1, 0, 1>>(p, stream);      // 8 x 16-pixel tile (conv_split_tile_shape)
    return conv16_launch<Conv16Cfg<1, 1, 5, 4, 3, 0, 1>>(p, stream);
}
}  // namespace cdx
