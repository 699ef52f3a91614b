// Host side + instantiations of the fp16-storage convolution (conv16_kernel.h).
#include <math.h>
#include <string.h>

#include "conv16_kernel.h"
#include "conv_kpar_kernel.h"

using namespace cdx;

namespace {
inline int chunks_of(int c) { return (c + 31) / 32; }

int validate(const cdx_conv_f16_args* a) {
    CDX_REQUIRE(a && a->src0 && a->wpacked && a->out);
    const int cm = a->src_is_f32 ? 4 : 8;
    CDX_REQUIRE(a->c0 > 0 && a->c1 >= 0 && (a->c0 % cm) == 0 && (a->c1 % cm) == 0);
    CDX_REQUIRE((a->c1 == 0) == (a->src1 == nullptr));
    if (a->c1) CDX_REQUIRE((a->c0 % 32) == 0 && (a->c1 % 32) == 0);
    CDX_REQUIRE(a->batch > 0 && a->hin > 0 && a->win > 0 && a->cout > 0);
#ifdef CDX_TUNING
    CDX_REQUIRE((a->flags & ~(CDX_CONV_UPSAMPLE2X | CDX_CONV_GN | CDX_CONV_SILU | CDX_CONV_BF16 | 0xF00)) == 0);
#else
    CDX_REQUIRE((a->flags & ~(CDX_CONV_UPSAMPLE2X | CDX_CONV_GN | CDX_CONV_SILU | CDX_CONV_BF16)) == 0);   // unknown flag bits are an error
#endif
    CDX_REQUIRE(a->ksize == 1 || a->ksize == 3);
    CDX_REQUIRE(a->stride == 1 || (a->stride == 2 && a->ksize == 3));
    const int ups = (a->flags & CDX_CONV_UPSAMPLE2X) ? 1 : 0;
    CDX_REQUIRE(!(ups && a->stride != 1));
    const int hv = a->hin << ups, wv = a->win << ups;
    CDX_REQUIRE(a->hout == (a->stride == 1 ? hv : (hv + 1) / 2) && a->wout == (a->stride == 1 ? wv : (wv + 1) / 2));
    CDX_REQUIRE(a->out_ld >= ((a->cout + 3) & ~3) && (a->out_ld % 4) == 0);   // outputs move as 4-channel vectors
    if (a->residual) CDX_REQUIRE((a->cout % 4) == 0);
    CDX_REQUIRE(aligned16(a->src0) && aligned16(a->src1) && aligned16(a->wpacked));
    if (a->flags & CDX_CONV_GN) CDX_REQUIRE(a->gn_scale && a->gn_shift && aligned16(a->gn_scale) && aligned16(a->gn_shift) && ((a->c0 + a->c1) % 4) == 0);
    if (a->temb) CDX_REQUIRE(a->temb_ld >= a->cout);
    if (a->stats_out) CDX_REQUIRE(!a->out_is_f32);
    CDX_REQUIRE((int64_t)a->batch * a->hin * a->win < (1ll << 31) && (int64_t)a->batch * a->hout * a->wout < (1ll << 31));
    // 32-bit byte offsets inside ONE image of a source (the kernel rebases its buffer resource per image)
    CDX_REQUIRE((int64_t)a->hin * a->win * (a->c0 > a->c1 ? a->c0 : a->c1) * (a->src_is_f32 ? 4 : 2) < (1ll << 31));
    return CDX_OK;
}

// M-tiles per workgroup: 128 pixels at >= 32 pixels wide; 64 for stride 2 and at 16 / 8 pixels wide, where 128-pixel tiles
// of a batch-16 level leave most of the 256 CUs without a workgroup (16^2: 32 tiles x cout / 128)
int f16_mt(const cdx_conv_f16_args* a) { return (a->stride == 2 || (a->wout >= 8 && a->wout < 32)) ? 2 : 4; }

// 8 pixels wide, stride 1, 16-bit in and out: the chunk-parallel tile (conv_kpar_kernel.h; 4 GroupNorm-sum slots per tile)
bool f16_kpar(const cdx_conv_f16_args* a) {
    return a->stride == 1 && a->wout >= 8 && a->wout < 16 && !a->src_is_f32 && !a->out_is_f32 && (a->c0 % 8) == 0 && (a->c1 % 8) == 0 &&
           (a->cout % 4) == 0;
}

// 3x3 stride-1 layers at >= 32 pixels wide: the 8 x 16-pixel wave-specialised tile (180-pixel halo = 3 staging passes instead of
// 204 = 4: conv_split.hip conv_split_tile_shape)
bool f16_tile816(const cdx_conv_f16_args* a) { return a->ksize == 3 && a->stride == 1 && a->wout >= 32; }

void tile_grid(const cdx_conv_f16_args* a, int& logtw, int& tx, int& ty) {
    logtw = a->wout >= 32 ? 5 : a->wout >= 16 ? 4 : a->wout >= 8 ? 3 : 2;
    if (f16_tile816(a)) logtw = 4;
    const int bm = 32 * f16_mt(a), tw = 1 << logtw, th = bm / tw;
    tx = ceil_div(a->wout, tw);
    ty = ceil_div(a->hout, th);
}
}  // namespace

namespace cdx {
int conv16_dispatch_bf16(int ks, int stride, int logtw, int mt, const Conv16Params& p, hipStream_t stream);
int conv16_kpar_dispatch_bf16(int ks, const Conv16Params& p, hipStream_t stream);
int conv16_kpar_dispatch(int ks, bool bf, const Conv16Params& p, hipStream_t stream) {
    if (bf) return conv16_kpar_dispatch_bf16(ks, p, stream);
    if (ks == 3) return conv_kpar_launch<KparCfg<3, 3, 0, 0>>(p, stream);
    return conv_kpar_launch<KparCfg<1, 3, 0, 0>>(p, stream);
}
int conv16_dispatch(int ks, int stride, int logtw, int mt, bool bf, const Conv16Params& p, hipStream_t stream) {
#ifdef CDX_TUNING
    // timing ablations of the dominant shape (libcdx_tune.so only), selected by flag bits 8..10
    if (ks == 3 && stride == 1 && logtw == 4 && mt == 4 && p.abl) {
        switch (p.abl) {
            case 1: return conv16_launch<Conv16Cfg<3, 1, 4, 4, 3, 1>>(p, stream);
            case 4: return conv16_launch<Conv16Cfg<3, 1, 4, 4, 3, 4>>(p, stream);
            case 7: return conv16_launch<Conv16Cfg<3, 1, 4, 4, 3, 7>>(p, stream);
            case 5: return conv16_ws_launch<Conv16Cfg<3, 1, 4, 4, 3, 0, 0, 1, 0, 1>>(p, stream);      // wave-specialised: 4 MFMA + 4 producer waves
            case 6: return conv16_ws_launch<Conv16Cfg<3, 1, 4, 4, 3, 2, 0, 1, 0, 1>>(p, stream);      // ... producers stage only the first chunk (the MFMA waves' bound)
            case 3: return conv16_ws_launch<Conv16Cfg<3, 1, 4, 4, 3, 1, 0, 1, 0, 1>>(p, stream);      // ... no epilogue
            case 2: return conv16_ws_launch<Conv16Cfg<3, 1, 4, 4, 3, 3, 0, 1, 0, 1>>(p, stream);      // ... neither
            case 10: return conv16_ws_launch<Conv16Cfg<3, 1, 4, 4, 3, 544, 0, 1, 0, 1>>(p, stream);   // ... per-wave stamps / barrier accounting instead of GroupNorm sums (tools/ws_stamps.py --fp16)
            case 11: return conv16_ws_launch<Conv16Cfg<3, 1, 4, 4, 3, 65536, 0, 1, 0, 1>>(p, stream);   // ... without the producers' prologue priority
            case 9: return conv16_ws_launch<Conv16Cfg<3, 1, 4, 4, 3, 8192, 0, 1, 0, 1>>(p, stream);   // ... XCD-contiguous workgroup order (measured, not shipped)
            case 8: return conv16_ws_launch<Conv16Cfg<3, 1, 4, 4, 3, 4096, 0, 1, 0, 1>>(p, stream);   // ... operand reads NOT software-pipelined (round 3's MFMA waves)
            default: return CDX_ENOTSUP;
        }
    }
#endif
    if (bf) return conv16_dispatch_bf16(ks, stride, logtw, mt, p, stream);      // conv16_bf16.hip
    // 3x3 at >= 32 pixels wide: the wave-specialised workgroup (4 MFMA + 4 producer waves, conv16_kernel.h WS): with one MFMA per
    // (tap, 16 channels, M-tile) the staging arithmetic was 40 % of the 4-wave kernel -- +7...14 % in-process (profiles/r03_*)
    if (ks == 3 && stride == 1 && logtw == 4 && mt == 4) return conv16_ws_launch<Conv16Cfg<3, 1, 4, 4, 3, 0, 0, 1, 0, 1>>(p, stream);
#define C16(KS, ST, LT, MT) if (ks == KS && stride == ST && logtw == LT) return conv16_launch<Conv16Cfg<KS, ST, LT, MT>>(p, stream);
    C16(3, 1, 2, 4) C16(3, 1, 3, 2) C16(3, 1, 4, 2)
    C16(1, 1, 2, 4) C16(1, 1, 3, 2) C16(1, 1, 4, 2) C16(1, 1, 5, 4)
    C16(3, 2, 2, 2) C16(3, 2, 3, 2) C16(3, 2, 4, 2) C16(3, 2, 5, 2)
#undef C16
    return CDX_ENOTSUP;
}
}  // namespace cdx

extern "C" size_t cdx_conv_f16_packed_halves(int32_t c0, int32_t c1, int32_t cout, int32_t ksize) {
    if (c0 <= 0 || c1 < 0 || cout <= 0 || (ksize != 1 && ksize != 3)) return 0;
    const size_t ntiles = (cout + 31) / 32, nch = chunks_of(c0) + chunks_of(c1);
    return ntiles * nch * ksize * ksize * 1024 + 8192;
}

extern "C" int cdx_conv_pack_weights_f16(const float* w, int32_t c0, int32_t c1, int32_t cout, int32_t ksize, cdx_half* packed) {
    CDX_REQUIRE(w && packed && c0 > 0 && c1 >= 0 && cout > 0 && (ksize == 1 || ksize == 3));
    const int taps = ksize * ksize, ctot = c0 + c1;
    const int nch0 = chunks_of(c0), nch = nch0 + chunks_of(c1), ntiles = (cout + 31) / 32;
    _Float16* o = reinterpret_cast<_Float16*>(packed);
    for (int nt = 0; nt < ntiles; ++nt)
        for (int ch = 0; ch < nch; ++ch)
            for (int tap = 0; tap < taps; ++tap)
                for (int j = 0; j < 2; ++j)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int n = nt * 32 + (lane & 31);
                            const int cl = (ch < nch0 ? ch : ch - nch0) * 32 + 16 * j + 8 * (lane >> 5) + e;
                            const int csrc = ch < nch0 ? c0 : c1;
                            float v = 0.f;
                            if (n < cout && cl < csrc) v = w[((size_t)n * ctot + (ch < nch0 ? 0 : c0) + cl) * taps + tap];
                            *o++ = (_Float16)v;
                        }
    memset(o, 0, 8192 * sizeof(_Float16));
    return CDX_OK;
}

extern "C" int cdx_conv_pack_weights_bf16(const float* w, int32_t c0, int32_t c1, int32_t cout, int32_t ksize, uint16_t* packed) {
    CDX_REQUIRE(w && packed && c0 > 0 && c1 >= 0 && cout > 0 && (ksize == 1 || ksize == 3));
    const int taps = ksize * ksize, ctot = c0 + c1;
    const int nch0 = chunks_of(c0), nch = nch0 + chunks_of(c1), ntiles = (cout + 31) / 32;
    uint16_t* o = packed;
    for (int nt = 0; nt < ntiles; ++nt)
        for (int ch = 0; ch < nch; ++ch)
            for (int tap = 0; tap < taps; ++tap)
                for (int j = 0; j < 2; ++j)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int n = nt * 32 + (lane & 31);
                            const int cl = (ch < nch0 ? ch : ch - nch0) * 32 + 16 * j + 8 * (lane >> 5) + e;
                            const int csrc = ch < nch0 ? c0 : c1;
                            float v = 0.f;
                            if (n < cout && cl < csrc) v = w[((size_t)n * ctot + (ch < nch0 ? 0 : c0) + cl) * taps + tap];
                            uint32_t u;
                            memcpy(&u, &v, 4);
                            u += 0x7FFFu + ((u >> 16) & 1u);      // round to nearest even (finite weights)
                            *o++ = (uint16_t)(u >> 16);
                        }
    memset(o, 0, 8192 * sizeof(uint16_t));
    return CDX_OK;
}

extern "C" size_t cdx_conv_f16_workspace(const cdx_conv_f16_args*) { return 0; }

extern "C" int32_t cdx_conv_f16_stats_slots(const cdx_conv_f16_args* a) {
    if (validate(a)) return 0;
    int logtw, tx, ty;
    tile_grid(a, logtw, tx, ty);
    return tx * ty * (f16_kpar(a) ? 4 : conv16_tail_2x2(a->cout, f16_mt(a)) ? 2 : 1);
}

extern "C" int cdx_conv_f16(const cdx_conv_f16_args* a, void*, size_t, cdx_stream_t stream) {
    const int rc = validate(a);
    if (rc) return rc;
    Conv16Params p;
    p.src[0] = a->src0;
    p.src[1] = a->src1 ? a->src1 : a->src0;
    p.csrc[0] = a->c0;
    p.csrc[1] = a->c1 ? a->c1 : a->c0;
    p.src_f32 = a->src_is_f32 ? 1 : 0;
    p.nchunk0 = chunks_of(a->c0);
    p.nchunks = p.nchunk0 + chunks_of(a->c1);
    p.ctot = a->c0 + a->c1;
    p.B = a->batch; p.Hin = a->hin; p.Win = a->win; p.Hout = a->hout; p.Wout = a->wout; p.Cout = a->cout;
    p.ups = (a->flags & CDX_CONV_UPSAMPLE2X) ? 1 : 0;
    p.gn = (a->flags & CDX_CONV_GN) ? 1 : 0;
    p.silu = (a->flags & CDX_CONV_SILU) ? 1 : 0;
    p.abl = (a->flags >> 8) & 15;
    p.wunscale = 1.f;
    p.w = a->wpacked;
    p.bias = a->bias; p.gscale = a->gn_scale; p.gshift = a->gn_shift; p.temb = a->temb; p.temb_ld = a->temb_ld;
    p.residual = a->residual;
    p.out = a->out; p.out_f32 = a->out_is_f32 ? 1 : 0; p.out_ld = a->out_ld; p.stats = a->stats_out;
    p.stats_wm = conv16_tail_2x2(a->cout, f16_mt(a)) ? 2 : 1;
    p.ostep = 1; p.ody = p.odx = 0; p.pady = p.padx = a->ksize / 2; p.slot_base = 0; p.nslots_total = 0;
    p.act_exp = 0;
    p.amax[0] = p.amax[1] = nullptr;
    p.amax_out = nullptr;
    int logtw;
    tile_grid(a, logtw, p.tiles_x, p.tiles_y);
    CDX_REQUIRE((int64_t)p.tiles_x * p.tiles_y * p.B < (1ll << 31));
    if (f16_kpar(a)) return conv16_kpar_dispatch(a->ksize, (a->flags & CDX_CONV_BF16) != 0, p, static_cast<hipStream_t>(stream));
    return conv16_dispatch(a->ksize, a->stride, logtw, f16_mt(a), (a->flags & CDX_CONV_BF16) != 0, p, static_cast<hipStream_t>(stream));
}

