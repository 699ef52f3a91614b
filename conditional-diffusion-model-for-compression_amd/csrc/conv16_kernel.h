// fp16-storage convolution: implicit GEMM on v_mfma_f32_32x32x16_f16 (fp16 operands, fp32 accumulate).
// BASELINE.json configs[4] ("fp16 UNet on CDNA4 MFMA"); SURVEY.md section 8(a) U3/U4 "_f16 variants".
//
// Same structure as conv_kernel.h -- a TH x TW pixel rectangle x 128 output channels per workgroup; per
// 32-channel chunk the input halo is gathered to LDS once with GroupNorm scale/shift (fp32), SiLU, nearest-2x
// upsampling and channel concat applied on the way and rounded to fp16; every tap reads its A fragments from that one
// image; weights come fragment-packed from L2 through a prefetch ring; bias/temb/residual/GroupNorm partial sums in the
// epilogue -- with the differences the 16x faster matrix instruction forces:
//   * one MFMA consumes 16 channels: lane (row i = lane&31, half h = lane>>5) supplies channels 8h..8h+7 of the
//     16-channel step as ONE 16-byte LDS read (A) / ONE 16-byte global load (B); two MFMAs per (tap, chunk, M-tile);
//   * LDS image in fp16: pixel stride 40 halves (32 + 8 pad = 80 B: ds_read_b128 lane groups hit 16 distinct 4-bank
//     slots), row stride a multiple of 256 B;
//   * sources may be fp32 (the sampler's x_t buffer feeding conv_in) or fp16; the output fp16 or fp32 (conv_out);
//   * GroupNorm statistics are taken from the fp32 accumulators BEFORE rounding (GN stays fp32).
// The kernel is no longer MFMA-bound (36.9k -> 2.3k MFMA cycles per chunk): the staging VALU, LDS reads and the
// L2 weight stream set its speed.
//
// SPLIT = 1: the SAME kernel computes a FLOAT32 convolution on the fp16 matrix pipe (cdx_conv_f32's "split" tile).
// Every float32 operand is split while staging into hi = fp16(v) and lo = fp16(v - hi) (v = hi + lo to ~2^-24 |v|; the
// weights are pre-scaled by a per-layer power of two on the host so that their lo parts are normal fp16 numbers), and
// each product block is three MFMAs: hi*hi + lo*hi + hi*lo, accumulated in float32 (the dropped lo*lo term is
// <= 2^-22 of the product).  Measured error vs float64: the same as the float32-MFMA kernels' (tests: unchanged
// tolerances).  v_mfma_f32_32x32x16_f16 retires 16 channels in 32 cycles where v_mfma_f32_32x32x2_f32 needs 8 x 64:
// 3 x 32 = 96 cycles against 512 (direct) or 228 (Winograd F(2x2,3x3)) -- and unlike the f32-input MFMA it leaves
// the vector ALU free for the staging work.  LDS image: per pixel 32 hi halves | 32 lo halves | 16 B pad = 144 B (the
// float32 kernels' conflict-free stride); sources, residual and output float32; GroupNorm sums float64 as everywhere.
#pragma once
#include <hip/hip_fp16.h>

#include <type_traits>

#include "common.h"

namespace cdx {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

struct Conv16Params {
    const void* src[2];
    int csrc[2];
    int src_f32;     // sources are float32 (else float16)
    int nchunk0, nchunks, ctot;
    int B, Hin, Win, Hout, Wout, Cout;
    int ups, gn, silu;
    int abl;         // timing ablation selector (diagnostics)
    float wunscale;  // SPLIT: 2^-s, undoes the host's power-of-two weight scaling (exact); 1 otherwise
    const _Float16* w;
    const float* bias;
    const float* gscale;
    const float* gshift;
    const float* temb;
    int temb_ld;
    const void* residual;   // fp16 (fp32 in SPLIT mode)
    void* out;
    int out_f32;
    int out_ld;
    double* stats;
    int tiles_x, tiles_y;
};

// ABL: timing-only ablations (wrong results; libcdx_tune.so only): 1 = no epilogue, 2 = stage only the first chunk,
// 4 = no weight refills, 8 = no LDS operand reads (registers reused)
// DB = 0: ONE halo image and two barriers per chunk (stride-2 SPLIT tiles, whose 5 x 65-pixel hi|lo image would
// otherwise leave room for a single workgroup per CU).
template <int KS_, int STRIDE_, int LOGTW_, int MT_, int PF_ = 3, int ABL_ = 0, int SPLIT_ = 0, int DB_ = 1>
struct Conv16Cfg {
    static constexpr int KS = KS_, STRIDE = STRIDE_, LOGTW = LOGTW_, MT = MT_, PF = PF_, ABL = ABL_, SPLIT = SPLIT_, DB = DB_;
    static constexpr int PLANES = SPLIT ? 2 : 1;                   // hi | lo
    static constexpr int KC = 32, PSH = KC * PLANES + 8;           // pixel stride in halves (80 B / 144 B)
    static constexpr int TAPS = KS * KS, PAD = KS / 2;
    static constexpr int TW = 1 << LOGTW;
    static constexpr int BM = MT * 32, BN = 128;
    static constexpr int TH = BM / TW;
    static constexpr int RPM = 32 / TW;
    static constexpr int HH = (TH - 1) * STRIDE + KS, HW = (TW - 1) * STRIDE + KS;
    static constexpr int RSH = ((HW * PSH + 127) / 128) * 128;     // row stride in halves (multiple of 256 B)
    static constexpr int LDS_HALVES = HH * RSH;
    static constexpr int NPIX = HH * HW;
    static constexpr int NPASS = (NPIX + 63) / 64;                 // 64 pixel slots x 4 channel octets per pass
    static constexpr int GPC = TAPS * 2;                           // (tap, 16-channel step) groups per chunk
    static_assert(TW <= 32 && BM % TW == 0, "tile shape");
    static_assert(LDS_HALVES * 2 * (DB ? 2 : 1) <= 160 * 1024, "LDS budget");
    static_assert(GPC % PF == 0 || PF > GPC, "ring depth");
};

__device__ __forceinline__ float silu16_f(float v) {
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f));
}

template <class C>
__global__ __launch_bounds__(256, 2) void conv16_kernel(const Conv16Params p) {
    constexpr int KC = C::KC, PSH = C::PSH, RSH = C::RSH, TAPS = C::TAPS, MT = C::MT, NPASS = C::NPASS, GPC = C::GPC;
    constexpr int PF = C::PF < GPC ? C::PF : GPC;
    // two halo images: chunk c+1 is staged into the other one WHILE chunk c's MFMAs run (one pass per MFMA group), one
    // barrier per chunk -- the f16 MFMA leaves the vector ALU free (unlike the f32 one), so the GroupNorm / SiLU /
    // fp16-rounding work of the staging hides under it instead of standing between two barriers.
    __shared__ __attribute__((aligned(16))) _Float16 lds_all[(C::DB ? 2 : 1) * C::LDS_HALVES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);

    int bx = blockIdx.x;
    const int tx = bx % p.tiles_x;
    bx /= p.tiles_x;
    const int ty = bx % p.tiles_y;
    const int b = bx / p.tiles_y;
    const int oy0 = ty * C::TH, ox0 = tx * C::TW;
    const int iy0 = oy0 * C::STRIDE - C::PAD, ix0 = ox0 * C::STRIDE - C::PAD;
    const int Hv = p.Hin << p.ups, Wv = p.Win << p.ups;

    // ---- loader: thread -> (pixel slot pl of 64, channel octet q of 4) ----
    const int q = tid & 3, pl = tid >> 2;
    int soff[NPASS];
    unsigned vmask = 0;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int hp = i * 64 + pl;
        const int hy = hp / C::HW, hx = hp - hy * C::HW;
        const int iy = iy0 + hy, ix = ix0 + hx;
        const bool ok = hp < C::NPIX && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv;
        soff[i] = ok ? ((b * p.Hin + (iy >> p.ups)) * p.Win + (ix >> p.ups)) : 0;
        vmask |= ok ? (1u << i) : 0u;
    }
    float pre[NPASS][8];
    f32x4 gsc[2], gsh[2];
    bool cvalid;

    auto issue_loads = [&](int chunk) {
        const int s = chunk >= p.nchunk0;
        const int cl = (s ? chunk - p.nchunk0 : chunk) * KC + q * 8;
        const int cs = p.csrc[s];
        cvalid = cl < cs;                       // channel counts are multiples of 8 (fp16) / 4 (fp32 conv_in: see host)
        const int c0 = cvalid ? cl : 0;
        if (p.src_f32) {
            const float* __restrict__ base = static_cast<const float*>(p.src[s]) + c0;
            const bool hi = c0 + 8 <= cs;       // fp32 sources may end on a 4-channel boundary
#pragma unroll
            for (int i = 0; i < NPASS; ++i) {
                const float* a = base + (size_t)soff[i] * cs;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(a);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(hi ? a + 4 : a);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pre[i][e] = v0[e];
                    pre[i][4 + e] = hi ? v1[e] : 0.f;
                }
            }
        } else {
            const _Float16* __restrict__ base = static_cast<const _Float16*>(p.src[s]) + c0;
#pragma unroll
            for (int i = 0; i < NPASS; ++i) {
                const f16x8 v = *reinterpret_cast<const f16x8*>(base + (size_t)soff[i] * cs);
#pragma unroll
                for (int e = 0; e < 8; ++e) pre[i][e] = (float)v[e];
            }
        }
        if (p.gn) {
            const int cg = cvalid ? (s ? p.csrc[0] : 0) + cl : 0;
            const float* gs = p.gscale + (size_t)b * p.ctot + cg;
            const float* gh = p.gshift + (size_t)b * p.ctot + cg;
            gsc[0] = *reinterpret_cast<const f32x4*>(gs);
            gsc[1] = *reinterpret_cast<const f32x4*>(gs + 4);
            gsh[0] = *reinterpret_cast<const f32x4*>(gh);
            gsh[1] = *reinterpret_cast<const f32x4*>(gh + 4);
        }
    };

    auto write_pass = [&](_Float16* lds, int i) {
        {
            const int hp = i * 64 + pl;
            const int hy = hp / C::HW, hx = hp - hy * C::HW;
            const bool ok = cvalid && ((vmask >> i) & 1u);
            f16x8 o, ol;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = pre[i][e];
                if (p.gn) v = fmaf(v, gsc[e >> 2][e & 3], gsh[e >> 2][e & 3]);
                if (p.silu) v = silu16_f(v);
                if constexpr (C::SPLIT) {
                    v = ok ? __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f) : 0.f;      // saturate instead of inf
                    o[e] = (_Float16)v;
                    ol[e] = (_Float16)(v - (float)o[e]);                               // exact difference, rounded once
                } else o[e] = (_Float16)(ok ? v : 0.f);
            }
            if (hp < C::NPIX) {
                *reinterpret_cast<f16x8*>(&lds[hy * RSH + hx * PSH + q * 8]) = o;
                if constexpr (C::SPLIT) *reinterpret_cast<f16x8*>(&lds[hy * RSH + hx * PSH + KC + q * 8]) = ol;
            }
        }
    };

    // ---- MFMA operand addressing ----
    const int li = lane & 31, lh = lane >> 5;
    const int a_base = ((li >> C::LOGTW) * C::STRIDE) * RSH + ((li & (C::TW - 1)) * C::STRIDE) * PSH + lh * 8;
    const int ntile = blockIdx.y * 4 + wn;
    const bool nvalid = ntile * 32 < p.Cout;
    // packed weights: [ntile][chunk][tap][j = 0..1][plane][lane][8 halves] -> one group = GH halves (1 KiB per plane)
    constexpr int GH = 512 * C::PLANES;
    const _Float16* __restrict__ wp = p.w + ((size_t)(nvalid ? ntile : 0) * p.nchunks * TAPS) * (2 * GH) + lane * 8;

    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    f16x8 ring[PF][C::PLANES];
#pragma unroll
    for (int j = 0; j < PF; ++j)
#pragma unroll
        for (int pl_ = 0; pl_ < C::PLANES; ++pl_) ring[j][pl_] = *reinterpret_cast<const f16x8*>(wp + j * GH + pl_ * 512);

    // packed-epilogue layout (see below)
    using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
    const int q4 = li & 3;
    const int cq = ntile * 32 + (li & ~3);                    // first of this quad's 4 channels
    const bool quad_ok = nvalid && cq < p.Cout;               // (cout is a multiple of 4 or the tail is zero-weighted)
    // staging schedule inside a chunk: passes at groups G0 .. G0+NPASS-1, the loads of the chunk after next right behind
    constexpr int G0 = GPC > NPASS + 1 ? GPC - NPASS - 1 : 0;
    static_assert(NPASS + 1 <= GPC || GPC <= 2, "staging passes must fit in the chunk's groups (1x1: done after the groups)");
    issue_loads(0);
#pragma unroll
    for (int i = 0; i < NPASS; ++i) write_pass(lds_all, i);
    if (p.nchunks > 1 && !(C::ABL & 2)) issue_loads(1);
    __syncthreads();
    f16x8 a[MT], al[MT];
    if constexpr (C::ABL & 8) {                 // ablation: operands read once
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            a[t] = *reinterpret_cast<const f16x8*>(&lds_all[a_base + t * C::RPM * C::STRIDE * RSH]);
            al[t] = *reinterpret_cast<const f16x8*>(&lds_all[a_base + t * C::RPM * C::STRIDE * RSH + 16]);
        }
    }
    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
        const _Float16* lds = lds_all + (C::DB ? (chunk & 1) * C::LDS_HALVES : 0);
        _Float16* nxt = lds_all + (C::DB ? ((chunk + 1) & 1) * C::LDS_HALVES : 0);
        const bool more = chunk + 1 < p.nchunks && !(C::ABL & 2);
        if (C::DB && (!nvalid || GPC <= 2)) {
            // a wave without output channels (or a 1x1 layer: two groups per chunk) stages in one go
            if (more) {
#pragma unroll
                for (int i = 0; i < NPASS; ++i) write_pass(nxt, i);
                if (chunk + 2 < p.nchunks) issue_loads(chunk + 2);
            }
        }
        if (nvalid) {
            const _Float16* __restrict__ wc = wp + (size_t)chunk * (TAPS * 2 * GH);
#pragma unroll
            for (int g = 0; g < GPC; ++g) {
                if (C::DB && GPC > 2 && more) {
                    if (g >= G0 && g < G0 + NPASS) write_pass(nxt, g - G0);
                    if (g == G0 + NPASS && chunk + 2 < p.nchunks) issue_loads(chunk + 2);
                }
                const int tap = g >> 1, j = g & 1, ky = tap / C::KS, kx = tap % C::KS;
                int ab = a_base;
                asm volatile("" : "+v"(ab));                 // no cross-tap CSE of LDS reads (see conv_kernel.h)
                __builtin_assume((ab & 7) == 0);
                if constexpr (!(C::ABL & 8)) {
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        a[t] = *reinterpret_cast<const f16x8*>(&lds[ab + (t * C::RPM * C::STRIDE + ky) * RSH + kx * PSH + j * 16]);
                        if constexpr (C::SPLIT)
                            al[t] = *reinterpret_cast<const f16x8*>(&lds[ab + (t * C::RPM * C::STRIDE + ky) * RSH + kx * PSH + KC + j * 16]);
                    }
                }
                const f16x8 bq = ring[g % PF][0];
                f16x8 bl;
                if constexpr (C::SPLIT) bl = ring[g % PF][1];
                if constexpr (!(C::ABL & 4)) {
#pragma unroll
                    for (int pl_ = 0; pl_ < C::PLANES; ++pl_)      // wraps into the next chunk / tail pad
                        ring[g % PF][pl_] = *reinterpret_cast<const f16x8*>(wc + (size_t)(g + PF) * GH + pl_ * 512);
                }
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t], bq, acc[t], 0, 0, 0);
                if constexpr (C::SPLIT) {      // the two cross terms (lo*lo, <= 2^-22 of the product, is dropped)
#pragma unroll
                    for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bq, acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t], bl, acc[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        if constexpr (!C::DB) {      // single image: every wave is done reading it; write the next chunk in place
            if (more) {
#pragma unroll
                for (int i = 0; i < NPASS; ++i) write_pass(nxt, i);
                if (chunk + 2 < p.nchunks) issue_loads(chunk + 2);
            }
            __syncthreads();
        }
    }

    // ---- epilogue ----
    if (!nvalid) return;
    if constexpr (C::ABL & 1) {
        float keep = 0.f;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) keep += acc[t][r];
        if (keep == 123.456f) static_cast<float*>(p.out)[0] = keep;
        return;
    }
    const int n = ntile * 32 + li;
    const bool nok = n < p.Cout;
    float add = 0.f;
    if (nok) {
        add = p.bias ? p.bias[n] : 0.f;
        if (p.temb) add += p.temb[(size_t)b * p.temb_ld + n];
    }
    // Packed epilogue: 4x4 blocks (4 consecutive pixels x the quad's 4 channels) are transposed across lane quads in
    // registers, so every lane stores / loads 4 consecutive channels of ONE pixel (8 B fp16, 16 B fp32) -- 4x fewer,
    // 4x wider memory instructions than the accumulator layout allows.  GroupNorm sums are reduced in that layout.
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    const float un = p.wunscale;            // SPLIT: undo the weights' power-of-two scaling (exact); 1 otherwise
    auto epilogue = [&](auto has_res, auto has_stats, auto out32) __attribute__((always_inline)) {
        using res_t = std::conditional_t<C::SPLIT != 0, f32x4, f16x4>;      // residual: float32 in SPLIT mode
        using rel_t = std::conditional_t<C::SPLIT != 0, float, _Float16>;
        res_t rv[MT][4];
        if constexpr (decltype(has_res)::value) {      // one batch of 8/16-byte loads (a load in the last chunk instead
#pragma unroll                                          // would queue the weight ring behind HBM misses: vmcnt is in-order)
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int m = t * 32 + 8 * k + q4 + 4 * lh;
                    const int oy = min(oy0 + (m >> C::LOGTW), p.Hout - 1), ox = min(ox0 + (m & (C::TW - 1)), p.Wout - 1);
                    rv[t][k] = *reinterpret_cast<const res_t*>(static_cast<const rel_t*>(p.residual) +
                                                               (((size_t)b * p.Hout + oy) * p.Wout + ox) * p.Cout + (quad_ok ? cq : 0));
                }
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float x[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) x[c] = C::SPLIT ? fmaf(acc[t][4 * k + c], un, add) : acc[t][4 * k + c] + add;
                quad_transpose(x, q4);                        // now: pixel 8k + q4 (+4 lh) of tile t, channels cq..cq+3
                const int m = t * 32 + 8 * k + q4 + 4 * lh;
                const int oy = oy0 + (m >> C::LOGTW), ox = ox0 + (m & (C::TW - 1));
                if (quad_ok && oy < p.Hout && ox < p.Wout) {
                    const size_t pix = ((size_t)b * p.Hout + oy) * p.Wout + ox;
                    if constexpr (decltype(has_res)::value) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) x[c] += (float)rv[t][k][c];
                    }
                    if constexpr (decltype(out32)::value) {
                        *reinterpret_cast<f32x4*>(static_cast<float*>(p.out) + pix * p.out_ld + cq) = f32x4{x[0], x[1], x[2], x[3]};
                    } else {
                        *reinterpret_cast<f16x4*>(static_cast<_Float16*>(p.out) + pix * p.out_ld + cq) =
                            f16x4{(_Float16)x[0], (_Float16)x[1], (_Float16)x[2], (_Float16)x[3]};
                    }
                    if constexpr (decltype(has_stats)::value) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const double d = (double)x[c];
                            s1[c] += d;
                            s2[c] = fma(d, d, s2[c]);
                        }
                    }
                }
            }
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    if constexpr (C::SPLIT) {          // float32 in, float32 out
        if (p.residual) {
            if (p.stats) epilogue(T_{}, T_{}, T_{}); else epilogue(T_{}, F_{}, T_{});
        } else {
            if (p.stats) epilogue(F_{}, T_{}, T_{}); else epilogue(F_{}, F_{}, T_{});
        }
    } else if (p.out_f32) {
        if (p.residual) epilogue(T_{}, F_{}, T_{}); else epilogue(F_{}, F_{}, T_{});
    } else if (p.residual) {
        if (p.stats) epilogue(T_{}, T_{}, F_{}); else epilogue(T_{}, F_{}, F_{});
    } else {
        if (p.stats) epilogue(F_{}, T_{}, F_{}); else epilogue(F_{}, F_{}, F_{});
    }
    if (p.stats) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {      // the quad's 4 lanes and the two lane halves hold different pixels
            s1[c] += __shfl_xor(s1[c], 1);
            s2[c] += __shfl_xor(s2[c], 1);
            s1[c] += __shfl_xor(s1[c], 2);
            s2[c] += __shfl_xor(s2[c], 2);
            s1[c] += __shfl_xor(s1[c], 32);
            s2[c] += __shfl_xor(s2[c], 32);
        }
        if (lh == 0 && q4 == 0 && quad_ok) {
            const int slot = ty * p.tiles_x + tx;
            const int nslots = p.tiles_y * p.tiles_x;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (cq + c < p.Cout) {
                    double* o = p.stats + (((size_t)b * nslots + slot) * p.Cout + cq + c) * 2;
                    o[0] = s1[c];
                    o[1] = s2[c];
                }
            }
        }
    }
}

template <class C>
inline int conv16_launch(const Conv16Params& p, hipStream_t stream) {
    dim3 grid(p.tiles_x * p.tiles_y * p.B, ceil_div(p.Cout, C::BN));
    hipLaunchKernelGGL(conv16_kernel<C>, grid, dim3(256), 0, stream, p);
    return check_launch();
}

int conv16_dispatch(int ks, int stride, int logtw, const Conv16Params& p, hipStream_t stream);

}  // namespace cdx
