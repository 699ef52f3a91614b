// Host side of cdx_conv_f32: validation, tile-shape selection, weight packing.
// (U3/U4/U5/U8/U9 of SURVEY.md section 8a; no reference file exists to cite.)
#include <string.h>

#include "conv_kernel.h"

using namespace cdx;

namespace {

inline int chunks_of(int c) { return (c + CDX_CONV_KC - 1) / CDX_CONV_KC; }

}  // namespace

extern "C" size_t cdx_conv_packed_floats(int32_t c0, int32_t c1, int32_t cout, int32_t ksize) {
    if (c0 <= 0 || c1 < 0 || cout <= 0 || (ksize != 1 && ksize != 3)) return 0;
    const size_t ntiles = (cout + 31) / 32, nch = chunks_of(c0) + chunks_of(c1);
    return ntiles * nch * ksize * ksize * 1024 + 1024;   // + one fragment group of tail pad (prefetch overrun)
}

extern "C" int cdx_conv_pack_weights_f32(const float* w, int32_t c0, int32_t c1, int32_t cout, int32_t ksize,
                                         float* packed) {
    CDX_REQUIRE(w && packed && c0 > 0 && c1 >= 0 && cout > 0 && (ksize == 1 || ksize == 3));
    const int taps = ksize * ksize, ctot = c0 + c1;
    const int nch0 = chunks_of(c0), nch = nch0 + chunks_of(c1), ntiles = (cout + 31) / 32;
    float* o = packed;
    for (int nt = 0; nt < ntiles; ++nt)
        for (int ch = 0; ch < nch; ++ch)
            for (int tap = 0; tap < taps; ++tap)
                for (int s = 0; s < 4; ++s)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 4; ++e) {
                            const int n = nt * 32 + (lane & 31);
                            const int cl = (ch < nch0 ? ch : ch - nch0) * CDX_CONV_KC + 8 * s + 4 * (lane >> 5) + e;
                            const int csrc = ch < nch0 ? c0 : c1;
                            float v = 0.f;
                            if (n < cout && cl < csrc) {
                                const int c = (ch < nch0 ? 0 : c0) + cl;
                                v = w[((size_t)n * ctot + c) * taps + tap];
                            }
                            *o++ = v;
                        }
    memset(o, 0, 1024 * sizeof(float));
    return CDX_OK;
}

extern "C" size_t cdx_conv_f32_workspace(const cdx_conv_args*) { return 0; }

extern "C" int cdx_conv_f32(const cdx_conv_args* a, void*, size_t, cdx_stream_t stream) {
    CDX_REQUIRE(a && a->src0 && a->wpacked && a->out);
    CDX_REQUIRE(a->c0 > 0 && a->c1 >= 0 && (a->c0 % 4) == 0 && (a->c1 % 4) == 0);
    CDX_REQUIRE((a->c1 == 0) == (a->src1 == nullptr));
    if (a->c1) CDX_REQUIRE((a->c0 % CDX_CONV_KC) == 0 && (a->c1 % CDX_CONV_KC) == 0);
    CDX_REQUIRE(a->batch > 0 && a->hin > 0 && a->win > 0 && a->cout > 0);
    CDX_REQUIRE(a->ksize == 1 || a->ksize == 3);
    CDX_REQUIRE(a->stride == 1 || (a->stride == 2 && a->ksize == 3));
    const int ups = (a->flags & CDX_CONV_UPSAMPLE2X) ? 1 : 0;
    CDX_REQUIRE(!(ups && a->stride != 1));
    const int hv = a->hin << ups, wv = a->win << ups;
    CDX_REQUIRE(a->hout == (a->stride == 1 ? hv : (hv + 1) / 2) && a->wout == (a->stride == 1 ? wv : (wv + 1) / 2));
    CDX_REQUIRE(a->out_ld >= a->cout);
    CDX_REQUIRE(aligned16(a->src0) && aligned16(a->src1) && aligned16(a->wpacked));
    const bool gn = a->flags & CDX_CONV_GN;
    if (gn) CDX_REQUIRE(a->gn_scale && a->gn_shift && aligned16(a->gn_scale) && aligned16(a->gn_shift));
    if (a->temb) CDX_REQUIRE(a->temb_ld >= a->cout);
    // 32-bit pixel indexing inside the kernel
    CDX_REQUIRE((int64_t)a->batch * a->hin * a->win < (1ll << 31) && (int64_t)a->batch * a->hout * a->wout < (1ll << 31));

    ConvParams p;
    p.src[0] = a->src0;
    p.src[1] = a->src1 ? a->src1 : a->src0;
    p.csrc[0] = a->c0;
    p.csrc[1] = a->c1 ? a->c1 : a->c0;
    p.nchunk0 = chunks_of(a->c0);
    p.nchunks = p.nchunk0 + chunks_of(a->c1);
    p.ctot = a->c0 + a->c1;
    p.B = a->batch; p.Hin = a->hin; p.Win = a->win; p.Hout = a->hout; p.Wout = a->wout; p.Cout = a->cout;
    p.ups = ups; p.gn = gn ? 1 : 0; p.silu = (a->flags & CDX_CONV_SILU) ? 1 : 0;
    p.w = a->wpacked; p.bias = a->bias; p.gscale = a->gn_scale; p.gshift = a->gn_shift;
    p.temb = a->temb; p.temb_ld = a->temb_ld; p.residual = a->residual; p.out = a->out; p.out_ld = a->out_ld;

    const int logtw = a->wout >= 32 ? 5 : a->wout >= 16 ? 4 : a->wout >= 8 ? 3 : 2;
    int wcfg, bm;
    if (a->stride == 2) {
        wcfg = a->cout <= 64 ? WCFG_2x2x1 : WCFG_1x4x2;
        bm = 64;
    } else {
        wcfg = a->cout <= 32 ? WCFG_4x1x1 : a->cout <= 64 ? WCFG_2x2x2 : WCFG_1x4x4;
        bm = 128;
    }
    const int tw = 1 << logtw, th = bm / tw;
    p.tiles_x = ceil_div(a->wout, tw);
    p.tiles_y = ceil_div(a->hout, th);
    CDX_REQUIRE((int64_t)p.tiles_x * p.tiles_y * p.B < (1ll << 31));

    hipStream_t st = static_cast<hipStream_t>(stream);
    if (a->ksize == 1) return conv_dispatch_k1s1(logtw, wcfg, p, st);
    if (a->stride == 1) return conv_dispatch_k3s1(logtw, wcfg, p, st);
    return conv_dispatch_k3s2(logtw, wcfg, p, st);
}
