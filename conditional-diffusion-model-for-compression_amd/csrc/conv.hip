// Host side of cdx_conv_f32: validation, tile-shape selection, weight packing.
// (U3/U4/U5/U8/U9 of SURVEY.md section 8a; no reference file exists to cite.)
#include <string.h>

#include "conv_wino.h"

extern "C" int cdx_conv_f32_tile(const cdx_conv_args* a, int32_t tile, void*, size_t, cdx_stream_t stream);

using namespace cdx;

namespace cdx {
bool conv_split_ok(const cdx_conv_args* a);                       // conv_split.hip
bool conv_split_up_ok(const cdx_conv_args* a);
void conv_split_tile_shape(const cdx_conv_args* a, int& tw, int& th);
int conv_split_launch(const cdx_conv_args* a, hipStream_t stream, int variant = 0);
int conv_split_slots_per_tile(const cdx_conv_args* a);
int amax_launch(const float* x, int x_ld, int batch, int n, int channels, unsigned* out, hipStream_t stream);   // range.hip
}  // namespace cdx

namespace {

inline int chunks_of(int c) { return (c + CDX_CONV_KC - 1) / CDX_CONV_KC; }

}  // namespace

extern "C" size_t cdx_conv_packed_floats(int32_t c0, int32_t c1, int32_t cout, int32_t ksize) {
    if (c0 <= 0 || c1 < 0 || cout <= 0 || (ksize != 1 && ksize != 3)) return 0;
    const size_t ntiles = (cout + 31) / 32, nch = chunks_of(c0) + chunks_of(c1);
    return ntiles * nch * ksize * ksize * 1024 + 4096;   // + 16 KiB tail pad (prefetch-ring overrun)
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
    memset(o, 0, 4096 * sizeof(float));
    return CDX_OK;
}

extern "C" size_t cdx_conv_wino_packed_floats(int32_t c0, int32_t c1, int32_t cout) {
    if (c0 <= 0 || c1 < 0 || cout <= 0) return 0;
    const size_t ntiles = (cout + 31) / 32, nch = chunks_of(c0) + chunks_of(c1);
    return ntiles * nch * 16384 + 4096;
}

extern "C" int cdx_conv_pack_weights_wino_f32(const float* w, int32_t c0, int32_t c1, int32_t cout, float* packed) {
    CDX_REQUIRE(w && packed && c0 > 0 && c1 >= 0 && cout > 0);
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int ctot = c0 + c1;
    const int nch0 = chunks_of(c0), nch = nch0 + chunks_of(c1), ntiles = (cout + 31) / 32;
    float* o = packed;
    for (int nt = 0; nt < ntiles; ++nt)
        for (int ch = 0; ch < nch; ++ch)
            for (int s = 0; s < 4; ++s)
                for (int e = 0; e < 4; ++e)
                    for (int xq = 0; xq < 4; ++xq)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 4; ++j) {
                                const int n = nt * 32 + (lane & 31);
                                const int cl = (ch < nch0 ? ch : ch - nch0) * CDX_CONV_KC + 8 * s + 4 * (lane >> 5) + e;
                                const int csrc = ch < nch0 ? c0 : c1;
                                double u = 0.0;
                                if (n < cout && cl < csrc) {
                                    const int c = (ch < nch0 ? 0 : c0) + cl;
                                    const float* g = w + ((size_t)n * ctot + c) * 9;
                                    const int xi = 4 * xq + j, i = xi >> 2, jj = xi & 3;
                                    for (int a = 0; a < 3; ++a)
                                        for (int b = 0; b < 3; ++b) u += G[i][a] * (double)g[a * 3 + b] * G[jj][b];
                                }
                                *o++ = (float)u;
                            }
    memset(o, 0, 4096 * sizeof(float));
    return CDX_OK;
}

namespace {

struct Tile { int wcfg, bm, bn, wm; };

constexpr int kSplitKMaxPixels = 256;   // output pixels per image at or below which the split-K tiles are used

Tile tile_of(int wcfg) {
    switch (wcfg) {
        case WCFG_1x4x4: return {wcfg, 128, 128, 1};
        case WCFG_2x2x2: return {wcfg, 128, 64, 2};
        case WCFG_4x1x1: return {wcfg, 128, 32, 4};
        case WCFG_1x4x2: return {wcfg, 64, 128, 1};
        case WCFG_2x2x1: return {wcfg, 64, 64, 2};
        case WCFG_S32: return {wcfg, 32, 32, 1};
        case WCFG_S64: return {wcfg, 64, 32, 1};
        case WCFG_WINO: return {wcfg, 128, 128, 2};   // (wm = 2: the 8-wave kernel writes two GroupNorm slots per tile)
        case WCFG_CIN8: return {wcfg, 128, 128, 1};    // 128x128 tile that multiplies only the first 8 channels of the chunk
        case WCFG_SMALL: return {wcfg, 256, 4, 1};     // 8 x 32 pixels, cout <= 4 (4x4x1 MFMA form)
        case WCFG_SMALL_VALU: return {wcfg, 256, 4, 1};   // same tile, vector-ALU form (scalar-cache weights)
        case WCFG_SMALL_GEMM: return {wcfg, 512, 4, 1};   // 16 x 32 pixels, cout <= 3, taps as GEMM columns (v_mfma_f32_32x32x2_f32)
        case WCFG_SPLIT: return {wcfg, 128, 128, 1};      // fp16 matrix pipe, operands split hi/lo (conv16_kernel.h SPLIT)
        default: return {-1, 0, 0, 0};
    }
}

// Winograd F(2x2,3x3) applies to 3x3 stride-1 layers whose rows hold at least one 32-pixel tile row.
bool wino_ok(const cdx_conv_args* a) {
    return a->ksize == 3 && a->stride == 1 && a->wout >= 32 && a->wpacked_wino != nullptr && aligned16(a->wpacked_wino);
}

// cout <= 4 (conv_out): the vector-ALU kernel of conv_small.hip (no GroupNorm sums of the output: nothing normalises it)
bool small_ok(const cdx_conv_args* a) {
    return a->ksize == 3 && a->stride == 1 && a->cout <= 4 && a->wout >= 32 && a->stats_out == nullptr;
}

// ... and, for one source of 64 / 128 / 192 / 256 channels and cout <= 3, the GEMM form over (tap, cout) columns (HBM-bound)
bool small_gemm_ok(const cdx_conv_args* a) {
    return small_ok(a) && a->cout <= 3 && a->c1 == 0 && !(a->flags & CDX_CONV_UPSAMPLE2X) && a->c0 >= 64 && a->c0 <= 256 && a->c0 % 64 == 0;
}

// at most 8 input channels in ONE source (conv_in: x_t | cond | pad): the direct kernel without the 24 zero channels of its
// only chunk -- 36 MFMA k-steps per output block where the Winograd kernel spends a whole 128-MFMA chunk
bool cin8_ok(const cdx_conv_args* a) {
    return a->ksize == 3 && a->stride == 1 && a->c0 <= 8 && a->c1 == 0 && a->wout >= 32 && a->cout > 4;
}

// the split tile's geometry depends on the layer (128 / 64 pixels, slots per tile), not on the id alone
Tile split_tile(const cdx_conv_args* a) { return Tile{WCFG_SPLIT, (a->stride == 2 || a->wout < 32) ? 64 : 128, 128, conv_split_slots_per_tile(a)}; }

// Tile-shape heuristic.  Depends on the LAYER shape only, never on the batch: a different tile changes the
// summation order, and an image must decode to the same bits whatever batch / GPU shard it rides in.
Tile select_tile(const cdx_conv_args* a) {
    Tile t;
    if (a->stride == 2) t = tile_of(a->cout <= 64 ? WCFG_2x2x1 : WCFG_1x4x2);
    else t = tile_of(a->cout <= 32 ? WCFG_4x1x1 : a->cout <= 64 ? WCFG_2x2x2 : WCFG_1x4x4);
    // Low-resolution levels: 128-pixel tiles would give far fewer workgroups than the chip has CUs (8^2 x
    // 512 ch at batch 16 = 32 tiles of 128 x 128).  Use 32/64-pixel x 32-channel tiles whose 4 waves split K.
    const int hw = a->hout * a->wout;
    if (a->ksize == 1 && a->wout < 32) {
        // 1x1 at the low-resolution levels (attention qkv / proj, skip projections): measured at batch 16, 512 channels:
        // 8^2: 128x128 14 TF, 128x32 25, S32x32 50;  16^2: 128x128 48 TF, 128x64 65, S32x32 62
        t = tile_of(hw <= 64 ? WCFG_S32 : hw <= 256 ? WCFG_2x2x2 : t.wcfg);
    }
    // Layers at >= 32 pixels wide: the float32 product on the FP16 matrix pipe with split operands (3 MFMAs of 32 cycles
    // per 16 channels against 8 x 64 for the f32-input MFMA): same float32-level error, 2.4x less matrix-pipe time than
    // even the Winograd kernel.  Needs the wpacked_split image; otherwise the float32-MFMA kernels below are used.
    if (conv_split_ok(a)) return split_tile(a);
    if (a->ksize == 3) {
        if (hw <= kSplitKMaxPixels) t = tile_of(a->stride == 1 && hw >= 256 ? WCFG_S64 : WCFG_S32);
        else if (cin8_ok(a)) t = tile_of(WCFG_CIN8);
        else if (wino_ok(a) && a->cout >= 96) t = tile_of(WCFG_WINO);
        else if (small_gemm_ok(a)) t = tile_of(WCFG_SMALL_GEMM);
        else if (small_ok(a)) t = tile_of(WCFG_SMALL);
    }
    return t;
}

bool tile_allowed(const cdx_conv_args* a, int wcfg) {
    if (wcfg == WCFG_SPLIT) return conv_split_ok(a);
    if (a->stride == 2) return wcfg == WCFG_1x4x2 || wcfg == WCFG_2x2x1 || wcfg == WCFG_S32;
    if (a->ksize == 1) return wcfg == WCFG_1x4x4 || wcfg == WCFG_2x2x2 || wcfg == WCFG_4x1x1 || ((wcfg == WCFG_S32 || wcfg == WCFG_S64) && a->wout < 32);
    if (wcfg == WCFG_WINO) return wino_ok(a);
    if (wcfg == WCFG_CIN8) return cin8_ok(a);
    if (wcfg == WCFG_SMALL || wcfg == WCFG_SMALL_VALU) return small_ok(a);
    if (wcfg == WCFG_SMALL_GEMM) return small_gemm_ok(a);
    return wcfg == WCFG_1x4x4 || wcfg == WCFG_2x2x2 || wcfg == WCFG_4x1x1 || wcfg == WCFG_S32 || wcfg == WCFG_S64;
}

int validate(const cdx_conv_args* a) {
    CDX_REQUIRE(a && a->src0 && a->wpacked && a->out);
    CDX_REQUIRE(a->c0 > 0 && a->c1 >= 0 && (a->c0 % 4) == 0 && (a->c1 % 4) == 0);
    CDX_REQUIRE((a->c1 == 0) == (a->src1 == nullptr));
    if (a->c1) CDX_REQUIRE((a->c0 % CDX_CONV_KC) == 0 && (a->c1 % CDX_CONV_KC) == 0);
    CDX_REQUIRE(a->batch > 0 && a->hin > 0 && a->win > 0 && a->cout > 0);
    CDX_REQUIRE((a->flags & ~(CDX_CONV_UPSAMPLE2X | CDX_CONV_GN | CDX_CONV_SILU | CDX_CONV_GN_EXP)) == 0);   // unknown flag bits are an error (CDX_CONV_BF16: cdx_conv_f16 only)
    if (a->flags & CDX_CONV_GN_EXP) CDX_REQUIRE(a->flags & CDX_CONV_GN);
    CDX_REQUIRE(a->ksize == 1 || a->ksize == 3);
    CDX_REQUIRE(a->stride == 1 || (a->stride == 2 && a->ksize == 3));
    const int ups = (a->flags & CDX_CONV_UPSAMPLE2X) ? 1 : 0;
    CDX_REQUIRE(!(ups && a->stride != 1));
    const int hv = a->hin << ups, wv = a->win << ups;
    CDX_REQUIRE(a->hout == (a->stride == 1 ? hv : (hv + 1) / 2) && a->wout == (a->stride == 1 ? wv : (wv + 1) / 2));
    CDX_REQUIRE(a->out_ld >= a->cout);
    CDX_REQUIRE(aligned16(a->src0) && aligned16(a->src1) && aligned16(a->wpacked));
    if (a->flags & CDX_CONV_GN) CDX_REQUIRE(a->gn_scale && a->gn_shift && aligned16(a->gn_scale) && aligned16(a->gn_shift));
    if (a->wpacked_split) CDX_REQUIRE(aligned16(a->wpacked_split) && a->wsplit_unscale > 0.f);
    if (a->wpacked_split_up) CDX_REQUIRE(aligned16(a->wpacked_split_up) && a->wpacked_split != nullptr);
    CDX_REQUIRE(a->gn_exp >= -60 && a->gn_exp <= 60);
    if (!(a->flags & CDX_CONV_GN_EXP)) CDX_REQUIRE(a->gn_exp == 0);
    if (a->amax_out) CDX_REQUIRE((a->cout % 4) == 0 && (a->out_ld % 4) == 0 && aligned16(a->out));
    CDX_REQUIRE(((reinterpret_cast<uintptr_t>(a->src_amax0) | reinterpret_cast<uintptr_t>(a->src_amax1) | reinterpret_cast<uintptr_t>(a->amax_out)) & 63u) == 0);
    if (a->temb) CDX_REQUIRE(a->temb_ld >= a->cout);
    if (a->stats_out) CDX_REQUIRE((a->cout % 4) == 0 && (a->out_ld % 4) == 0);   // sums are produced by the packed epilogue
    if (a->residual && (a->cout % 4) == 0) CDX_REQUIRE(aligned16(a->residual));
    // 32-bit pixel indexing inside the kernel
    CDX_REQUIRE((int64_t)a->batch * a->hin * a->win < (1ll << 31) && (int64_t)a->batch * a->hout * a->wout < (1ll << 31));
    return CDX_OK;
}

}  // namespace

extern "C" size_t cdx_conv_f32_workspace(const cdx_conv_args*) { return 0; }

extern "C" int cdx_conv_select_tile(const cdx_conv_args* a) {
    const int rc = validate(a);
    return rc ? rc : select_tile(a).wcfg;
}

namespace {
int slots_of(const cdx_conv_args* a, const Tile& t) {
    // the four-phase form of an upsampled 3x3 layer tiles the LOW-resolution grid, once per phase
    if (t.wcfg == WCFG_SPLIT && conv_split_up_ok(a)) return 4 * ceil_div(a->win, 32) * ceil_div(a->hin, 4) * t.wm;
    if (t.wcfg == WCFG_SPLIT) {
        int tw, th;
        conv_split_tile_shape(a, tw, th);
        return ceil_div(a->wout, tw) * ceil_div(a->hout, th) * t.wm;
    }
    const int logtw = a->wout >= 32 ? 5 : a->wout >= 16 ? 4 : a->wout >= 8 ? 3 : 2;
    const int tw = 1 << logtw, th = t.bm / tw;
    return ceil_div(a->wout, tw) * ceil_div(a->hout, th) * t.wm;
}
}  // namespace

extern "C" int32_t cdx_conv_stats_slots(const cdx_conv_args* a) {
    if (validate(a)) return 0;
    // the question is asked BEFORE the caller has a buffer to put into stats_out: answer for the launch WITH sums
    // (a layer with cout <= 4 runs the matrix-pipe tile then, not the small kernels, which produce none)
    cdx_conv_args with_stats = *a;
    if (!with_stats.stats_out) with_stats.stats_out = reinterpret_cast<double*>(uintptr_t(16));
    if (validate(&with_stats)) return 0;
    return slots_of(a, select_tile(&with_stats));
}

extern "C" int cdx_conv_f32(const cdx_conv_args* a, void* ws, size_t ws_bytes, cdx_stream_t stream) {
    return cdx_conv_f32_tile(a, -1, ws, ws_bytes, stream);
}

extern "C" int cdx_conv_f32_tile(const cdx_conv_args* a, int32_t tile, void*, size_t, cdx_stream_t stream) {
    int rc = validate(a);
    if (rc) return rc;
#ifdef CDX_TUNING
    if (tile >= 60) {                       // split-tile ablations / variants (conv16.hip); same slot geometry as tile 11
        int rc60 = validate(a);
        if (rc60) return rc60;
        if (a->stats_out && select_tile(a).wcfg != WCFG_SPLIT) return CDX_EINVAL;
        return conv_split_ok(a) ? conv_split_launch(a, static_cast<hipStream_t>(stream), tile - 60) : CDX_ENOTSUP;
    }
    const bool experimental = tile >= 16;   // conv_exp.hip (16..30) / conv_wino.hip (31..): timing ablations (libcdx_tune.so)
#else
    if (tile >= 16) return CDX_ENOTSUP;     // the shipped library carries no tuning / ablation instantiations
    const bool experimental = false;
#endif
    Tile t = tile < 0 ? select_tile(a) : experimental ? Tile{tile, 128, 128, 1} : tile_of(tile);
    if (t.wcfg < 0 || (!experimental && !tile_allowed(a, t.wcfg))) return CDX_ENOTSUP;
    if (t.wcfg == WCFG_SPLIT) t = split_tile(a);
    if (experimental && !(a->ksize == 3 && a->stride == 1)) return CDX_ENOTSUP;
    const int ups = (a->flags & CDX_CONV_UPSAMPLE2X) ? 1 : 0;

    ConvParams p;
    p.src[0] = a->src0;
    p.src[1] = a->src1 ? a->src1 : a->src0;
    p.csrc[0] = a->c0;
    p.csrc[1] = a->c1 ? a->c1 : a->c0;
    p.nchunk0 = chunks_of(a->c0);
    p.nchunks = p.nchunk0 + chunks_of(a->c1);
    p.ctot = a->c0 + a->c1;
    p.B = a->batch; p.Hin = a->hin; p.Win = a->win; p.Hout = a->hout; p.Wout = a->wout; p.Cout = a->cout;
    p.ups = ups; p.gn = (a->flags & CDX_CONV_GN) ? 1 : 0; p.silu = (a->flags & CDX_CONV_SILU) ? 1 : 0;
    p.w = a->wpacked; p.bias = a->bias; p.gscale = a->gn_scale; p.gshift = a->gn_shift;
    p.temb = a->temb; p.temb_ld = a->temb_ld; p.residual = a->residual; p.out = a->out; p.out_ld = a->out_ld;
    p.stats = a->stats_out;
    if (a->stats_out && tile >= 0 && tile < 16 && tile != select_tile(a).wcfg) return CDX_EINVAL;   // slot count is defined for the library's own tile choice
    // the buffer was sized by an earlier cdx_conv_stats_slots call: refuse to write a different number of slots into it
    if (a->stats_out && !experimental) CDX_REQUIRE(a->stats_slots == slots_of(a, t));

    const int logtw = a->wout >= 32 ? 5 : a->wout >= 16 ? 4 : a->wout >= 8 ? 3 : 2;
    const int tw = 1 << logtw, th = t.bm / tw;
    p.tiles_x = ceil_div(a->wout, tw);
    p.tiles_y = ceil_div(a->hout, th);
    CDX_REQUIRE((int64_t)p.tiles_x * p.tiles_y * p.B < (1ll << 31));

    hipStream_t st = static_cast<hipStream_t>(stream);
    if (t.wcfg == WCFG_SPLIT) return conv_split_launch(a, st);      // (scales its activations by 2^gn_exp / amax; writes amax_out)
    // the f32-input MFMA kernels take GroupNorm scale / shift at unit scale: a pre-multiplied pair belongs to the split tile
    // (the host asks cdx_conv_select_tile first and passes out_exp = gn_exp = 0 otherwise)
    CDX_REQUIRE(a->gn_exp == 0 && !(a->flags & CDX_CONV_GN_EXP));
    if (t.wcfg == WCFG_WINO || (experimental && tile >= 31)) {
        if (!wino_ok(a)) return CDX_ENOTSUP;
        p.w = a->wpacked_wino;
        rc = conv_dispatch_wino(experimental ? tile : 0, p, st);
    } else if (t.wcfg == WCFG_SMALL || t.wcfg == WCFG_SMALL_VALU) rc = conv_dispatch_small(p, st, t.wcfg == WCFG_SMALL_VALU);
    else if (t.wcfg == WCFG_SMALL_GEMM) rc = conv_dispatch_small_gemm(p, st);
#ifdef CDX_TUNING
    else if (experimental) rc = conv_dispatch_exp(logtw, t.wcfg, p, st);
#endif
    else if (a->ksize == 1) rc = conv_dispatch_k1s1(logtw, t.wcfg, p, st);
    else if (a->stride == 1) rc = conv_dispatch_k3s1(logtw, t.wcfg, p, st);
    else rc = conv_dispatch_k3s2(logtw, t.wcfg, p, st);
    // amax_out of the tile shapes without a fused maximum: one more pass over the (small: these are the sub-8-pixel levels
    // and the layers a caller runs without the split image) output on the same stream
    if (rc == CDX_OK && a->amax_out) rc = amax_launch(a->out, a->out_ld, a->batch, a->hout * a->wout, a->cout, a->amax_out, st);
    return rc;
}
