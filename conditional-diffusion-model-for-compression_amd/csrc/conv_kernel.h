// Implicit-GEMM convolution on the gfx950 fp32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
// GEMM view: M = output pixels, N = cout, K = taps * cin.  One 256-thread workgroup (4 waves, one
// per SIMD) owns a TH x TW rectangle of output pixels of one image (BM = TH*TW pixels) and BN output
// channels.  Per 32-channel chunk of the input:
//   1. the (TH-1)*S+KS by (TW-1)*S+KS input halo tile is gathered NHWC -> LDS ONCE, with
//      GroupNorm scale/shift + SiLU, nearest-2x upsampling and channel concat applied on the way
//      (coalesced 128-B reads per pixel; zero padding applied after the activation);
//   2. all KS*KS taps read their A operand from that one LDS image at shifted addresses
//      (ds_read_b128 with immediate offsets: 9x reuse of every staged byte);
//   3. the B operand (weights) never touches LDS: the host pre-packs it so each lane's
//      16-byte fragment for 4 consecutive MFMAs is one coalesced global_load_dwordx4 (L2-resident),
//      prefetched one tap ahead.
// K-order inside a chunk is permuted so that one 16-B fragment feeds 4 MFMAs: MFMA e of group s
// consumes channel 8s + 4*(lane>>5) + e from both operands.
//
// LDS image: pixel stride PS = 36 floats (32 + 4 pad), row stride RS = multiple of 64 floats, which
// makes every ds_read_b128 lane group hit 16 distinct 4-bank slots (conflict-free) for TW >= 16.
#pragma once
#include <type_traits>

#include "common.h"

namespace cdx {

struct ConvParams {
    const float* src[2];
    int csrc[2];     // channels of each source
    int nchunk0;     // chunks belonging to src0
    int nchunks;     // total chunks
    int ctot;        // c0 + c1 (row length of gn_scale / gn_shift)
    int B, Hin, Win, Hout, Wout, Cout;
    int ups, gn, silu;
    const float* w;
    const float* bias;
    const float* gscale;
    const float* gshift;
    const float* temb;
    int temb_ld;
    const float* residual;
    float* out;
    int out_ld;
    double* stats;   // [B][tiles_y*tiles_x*WM][Cout][2] or nullptr
    int tiles_x, tiles_y;
};

// WM x WN x WK waves: WM/WN tile the output, WK splits K inside the workgroup (each wave takes every
// WK-th 8-channel group of every chunk; partial tiles are summed through LDS in fixed order at the end).
// PF = depth of the weight-fragment prefetch ring, in groups.
// OPT = tuning switches (A/B-tested in one binary through the experimental tile ids of conv_exp.hip):
enum { OPT_SPLIT_LDS_READS = 1,  // hide the 16-B alignment: 2 x ds_read2_b32 per fragment instead of ds_read_b128
       OPT_EXACT_SILU = 2,       // expf + IEEE divide instead of v_exp_f32 + v_rcp_f32
       OPT_OCC2 = 4,             // __launch_bounds__(256, 2): up to 256 VGPRs, 2 workgroups per CU
       OPT_STAGGER = 8,          // delay every second resident workgroup of the first round by ~half a tile
       // timing-only ablations (WRONG RESULTS; conv_exp.hip only):
       OPT_ABL_NO_STAGE = 16,    // chunks after the first skip the global loads + LDS writes (barriers stay)
       OPT_ABL_NO_EPILOGUE = 32, // no output stores / residual reads (one store keeps the accumulators live)
       OPT_ABL_ONE_WG = 64,      // declare 96 KiB of LDS: one workgroup per CU
       OPT_CIN8 = 256 };         // the layer has at most 8 input channels (conv_in): only the first 8-channel group of a
                                 // (chunk, tap) carries data, the other three (zeros) are not multiplied
template <int KS_, int STRIDE_, int LOGTW_, int WM_, int WN_, int MT_, int WK_ = 1, int PF_ = 1, int OPT_ = 0>
struct ConvCfg {
    static constexpr int KS = KS_, STRIDE = STRIDE_, LOGTW = LOGTW_, WM = WM_, WN = WN_, MT = MT_, WK = WK_, PF = PF_, OPT = OPT_;
    static constexpr int KC = CDX_CONV_KC, PS = KC + 4;
    static constexpr int TAPS = KS * KS, PAD = KS / 2;
    static constexpr int TW = 1 << LOGTW;
    static constexpr int BM = WM * MT * 32, BN = WN * 32;
    static constexpr int TH = BM / TW;
    static constexpr int RPM = 32 / TW;  // output rows per 32-pixel MFMA tile
    static constexpr int HH = (TH - 1) * STRIDE + KS, HW = (TW - 1) * STRIDE + KS;
    static constexpr int RS = ((HW * PS + 63) / 64) * 64;
    static constexpr int G = (OPT_ & 256) ? 1 : 4 / WK;   // 8-channel groups per (chunk, tap) per wave (OPT_CIN8: the first only)
    static constexpr int GPC = KS * KS * G;               // groups per chunk per wave
    static constexpr int RED_FLOATS = WK > 1 ? 4 * MT * 16 * 64 : 0;   // split-K reduction image
    static constexpr int LDS_FLOATS = HH * RS > RED_FLOATS ? HH * RS : RED_FLOATS;
    static constexpr int NPIX = HH * HW;
    static constexpr int NPASS = (NPIX + 31) / 32;
    static_assert(WM * WN * WK == 4, "4 waves per workgroup");
    static_assert(GPC % PF == 0 && PF <= GPC, "prefetch ring depth must divide the groups per chunk");
    static_assert(TW <= 32 && BM % TW == 0, "tile shape");
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
};

// x * sigmoid(x) on the transcendental unit: v_exp_f32 + v_rcp_f32 (1 ulp each) instead of the ~25-instruction
// expf + IEEE divide.  |error| <= ~4e-7 * |silu(x)|, the size of one float32 rounding of the result.
template <bool EXACT>
__device__ __forceinline__ float silu_f(float v) {
    if constexpr (EXACT) return v / (1.0f + expf(-v));
    else return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f));
}

template <class C>
__global__ __launch_bounds__(256, (C::OPT & OPT_OCC2) ? 2 : 3) void conv_kernel(const ConvParams p) {
    constexpr int KC = C::KC, PS = C::PS, RS = C::RS, TAPS = C::TAPS, MT = C::MT, NPASS = C::NPASS;
    __shared__ __attribute__((aligned(16))) float lds[(C::OPT & OPT_ABL_ONE_WG) ? 24576 : C::LDS_FLOATS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave % C::WK, wn = (wave / C::WK) % C::WN, wm = wave / (C::WK * C::WN);

    int bx = blockIdx.x;
    const int tx = bx % p.tiles_x;
    bx /= p.tiles_x;
    const int ty = bx % p.tiles_y;
    const int b = bx / p.tiles_y;
    const int oy0 = ty * C::TH, ox0 = tx * C::TW;
    const int iy0 = oy0 * C::STRIDE - C::PAD, ix0 = ox0 * C::STRIDE - C::PAD;
    const int Hv = p.Hin << p.ups, Wv = p.Win << p.ups;

    if constexpr (C::OPT & OPT_STAGGER) {
        // Workgroups that share a CU start together and would run their staging / epilogue phases (no MFMA)
        // in lockstep.  Offset the second resident set of the first dispatch round so one set's non-MFMA
        // phases fall under the other's MFMA phases.  Speed only: nothing depends on placement.
        const unsigned lin = blockIdx.x + blockIdx.y * gridDim.x;
        if (lin >= 256u && lin < 512u) {
            const int n = p.nchunks * 2;          // ~ half a tile: nchunks x 36.9k cycles / 2, in 8k-cycle sleeps
            for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
        }
    }

    // ---- loader geometry (chunk independent): thread -> (pixel slot pl, channel quad q) ----
    const int q = tid & 7, pl = tid >> 3;
    int soff[NPASS];        // source pixel index (b, sy, sx) flattened, or -1
    unsigned vmask = 0;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int hp = i * 32 + pl;
        const int hy = hp / C::HW, hx = hp - hy * C::HW;
        const int iy = iy0 + hy, ix = ix0 + hx;
        const bool ok = hp < C::NPIX && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv;
        soff[i] = ok ? ((b * p.Hin + (iy >> p.ups)) * p.Win + (ix >> p.ups)) : 0;
        vmask |= ok ? (1u << i) : 0u;
    }

    f32x4 pre[NPASS];
    f32x4 gsc, gsh;
    bool cvalid;   // this thread's channel quad exists in the current source

    auto issue_loads = [&](int chunk) {
        const int s = chunk >= p.nchunk0;
        const int cl = (s ? chunk - p.nchunk0 : chunk) * KC + q * 4;   // channel within source
        const int cs = p.csrc[s];
        cvalid = cl < cs;
        // Loads are unconditional (padding / tail lanes read a safe in-bounds address and are zeroed
        // at write time): a per-lane "load or zero" select would serialise the loads behind waits.
        const float* __restrict__ base = p.src[s] + (cvalid ? cl : 0);
#pragma unroll
        for (int i = 0; i < NPASS; ++i)
            pre[i] = *reinterpret_cast<const f32x4*>(base + (size_t)soff[i] * cs);
        if (p.gn) {
            const int cg = cvalid ? (s ? p.csrc[0] : 0) + cl : 0;
            gsc = *reinterpret_cast<const f32x4*>(p.gscale + (size_t)b * p.ctot + cg);
            gsh = *reinterpret_cast<const f32x4*>(p.gshift + (size_t)b * p.ctot + cg);
        }
    };

    auto write_lds = [&]() {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int hp = i * 32 + pl;
            const int hy = hp / C::HW, hx = hp - hy * C::HW;
            f32x4 v = pre[i];
            const bool ok = cvalid && ((vmask >> i) & 1u);
            if (p.gn) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], gsc[e], gsh[e]);
            }
            if (p.silu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = silu_f<(C::OPT & OPT_EXACT_SILU) != 0>(v[e]);
            }
            if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (hp < C::NPIX) *reinterpret_cast<f32x4*>(&lds[hy * RS + hx * PS + q * 4]) = v;
        }
    };

    // ---- MFMA operand addressing ----
    const int li = lane & 31, lh = lane >> 5;
    const int p0 = wm * MT * 32 + li;                       // pixel of M-tile 0, row li
    const int a_base = ((p0 >> C::LOGTW) * C::STRIDE) * RS + ((p0 & (C::TW - 1)) * C::STRIDE) * PS + lh * 4 + wk * 8;
    const int ntile = blockIdx.y * C::WN + wn;
    const bool nvalid = ntile * 32 < p.Cout;                // wave-uniform
    // Packed weights: group (chunk, tap, s) is 1 KiB = 64 lanes x 16 B; this wave uses s = wk + ss*WK.
    const float* __restrict__ wp =
        p.w + ((size_t)(nvalid ? ntile : 0) * p.nchunks * TAPS) * 1024 + wk * 256 + lane * 4;

    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // Weight fragments: one 16-B load per lane feeds 4 MFMAs x MT tiles (one "group").  A ring of PF
    // groups is kept in flight; group j of a chunk sits at float offset goff(j) from the chunk base, and
    // the ring wraps into the next chunk (the packed image carries a 16 KiB tail pad for the overrun).
    constexpr int G = C::G, GPC = C::GPC, PF = C::PF;
    auto goff = [](int j) { return ((j / G) * 4 + (j % G) * C::WK) * 256; };
    f32x4 ring[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) ring[j] = *reinterpret_cast<const f32x4*>(wp + goff(j));

    auto read_a = [&](f32x4 (&a)[MT], int ab, int j) {
        const int tap = j / G, ss = j % G, ky = tap / C::KS, kx = tap % C::KS;
#pragma unroll
        for (int t = 0; t < MT; ++t)
            a[t] = *reinterpret_cast<const f32x4*>(
                &lds[ab + (t * C::RPM * C::STRIDE + ky) * RS + kx * PS + ss * C::WK * 8]);
    };
    issue_loads(0);
    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
        if (!(C::OPT & OPT_ABL_NO_STAGE) || chunk == 0) write_lds();
        __syncthreads();
        if (!(C::OPT & OPT_ABL_NO_STAGE))
            if (chunk + 1 < p.nchunks) issue_loads(chunk + 1);   // lands under this chunk's MFMAs
        if (nvalid) {
            const float* __restrict__ wc = wp + (size_t)chunk * (TAPS * 1024);
            f32x4 an[MT];
            {
                int ab = a_base;
                asm volatile("" : "+v"(ab));
                if constexpr (!(C::OPT & OPT_SPLIT_LDS_READS)) __builtin_assume((ab & 3) == 0);
                read_a(an, ab, 0);
            }
#pragma unroll
            for (int j = 0; j < GPC; ++j) {
                f32x4 a[MT];
#pragma unroll
                for (int t = 0; t < MT; ++t) a[t] = an[t];
                if (j + 1 < GPC) {
                    // Opaque copy of the LDS base per group: M-tile t at tap row ky and M-tile t+1 at row
                    // ky-1 alias when a tile is one image row; without this hipcc keeps earlier fragments
                    // alive for reuse and spills them.  The assume keeps the 16-B alignment visible.
                    int ab = a_base;
                    asm volatile("" : "+v"(ab));
                    if constexpr (!(C::OPT & OPT_SPLIT_LDS_READS)) __builtin_assume((ab & 3) == 0);
                    read_a(an, ab, j + 1);      // next group's fragments land under this group's MFMAs
                }
                const f32x4 bcur = ring[j % PF];
                const int jn = j + PF;                      // refill this ring slot
                ring[j % PF] = *reinterpret_cast<const f32x4*>(
                    wc + (jn < GPC ? goff(jn) : TAPS * 1024 + goff(jn - GPC)));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int t = 0; t < MT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][e], bcur[e], acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- split-K: sum the WK partial tiles through LDS (fixed order wk = 0,1,2,3) ----
    if constexpr (C::WK > 1) {
        // (the chunk loop ended on a barrier: the halo image is dead)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) lds[((wave * MT + t) * 16 + r) * 64 + lane] = acc[t][r];
        __syncthreads();
        if (wk != 0) return;
#pragma unroll
        for (int k = 1; k < C::WK; ++k)
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] += lds[(((wave + k) * MT + t) * 16 + r) * 64 + lane];
    }

    // ---- epilogue: acc[t][r] is out[pixel (r&3) + 8*(r>>2) + 4*lh of tile t][channel li] ----
    if (!nvalid) return;
    if constexpr (C::OPT & OPT_ABL_NO_EPILOGUE) {
        float keep = 0.f;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) keep += acc[t][r];
        if (keep == 123.456f) p.out[0] = keep;
        return;
    }
    const int n = ntile * 32 + li;
    const bool nok = n < p.Cout;
    float add = 0.f;
    if (nok) {
        add = p.bias ? p.bias[n] : 0.f;
        if (p.temb) add += p.temb[(size_t)b * p.temb_ld + n];
    }
    // Packed epilogue: 4x4 blocks (4 consecutive pixels x the lane quad's 4 channels) are transposed across lane
    // quads in registers (quad_transpose), so every lane stores / loads 4 consecutive channels of ONE pixel as 16 bytes:
    // 4x fewer, 4x wider memory instructions than the accumulator layout allows.  GroupNorm sums (float64, of the
    // stored values) are reduced in that layout.  The residual / stats tests are hoisted out of the unrolled loops on
    // purpose: a per-element "load or not" makes hipcc branch around every load and wait vmcnt(0) each time.
    const int q4 = li & 3;
    const int cq = ntile * 32 + (li & ~3);                    // first of this quad's 4 channels
    const bool quad_ok = cq < p.Cout;
    const bool vec_ok = (p.out_ld & 3) == 0 && cq + 4 <= p.out_ld && (p.Cout & 3) == 0;
    if (!vec_ok) {
        // scalar fallback (cout not a multiple of 4, e.g. the 3-channel eps output with out_ld 3)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (wm * MT + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int oy = oy0 + (m >> C::LOGTW), ox = ox0 + (m & (C::TW - 1));
                if (nok && oy < p.Hout && ox < p.Wout) {
                    const size_t pix = ((size_t)b * p.Hout + oy) * p.Wout + ox;
                    float v = acc[t][r] + add;
                    if (p.residual) v += p.residual[pix * p.Cout + n];
                    p.out[pix * p.out_ld + n] = v;
                }
            }
        return;      // (GroupNorm sums are only requested for multiple-of-4 channel counts)
    }
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    auto epilogue = [&](auto has_res, auto has_stats) __attribute__((always_inline)) {
        f32x4 rv[MT][4];
        if constexpr (decltype(has_res)::value) {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int m = (wm * MT + t) * 32 + 8 * k + q4 + 4 * lh;
                    const int oy = min(oy0 + (m >> C::LOGTW), p.Hout - 1), ox = min(ox0 + (m & (C::TW - 1)), p.Wout - 1);
                    rv[t][k] = *reinterpret_cast<const f32x4*>(p.residual + (((size_t)b * p.Hout + oy) * p.Wout + ox) * p.Cout + (quad_ok ? cq : 0));
                }
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float x[4] = {acc[t][4 * k] + add, acc[t][4 * k + 1] + add, acc[t][4 * k + 2] + add, acc[t][4 * k + 3] + add};
                quad_transpose(x, q4);                        // now: pixel 8k + q4 (+4 lh) of tile t, channels cq..cq+3
                const int m = (wm * MT + t) * 32 + 8 * k + q4 + 4 * lh;
                const int oy = oy0 + (m >> C::LOGTW), ox = ox0 + (m & (C::TW - 1));
                if (quad_ok && oy < p.Hout && ox < p.Wout) {
                    const size_t pix = ((size_t)b * p.Hout + oy) * p.Wout + ox;
                    if constexpr (decltype(has_res)::value) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) x[c] += rv[t][k][c];
                    }
                    // (a global store: hipcc's hazard recogniser keeps a following write of these registers one wait state
                    // away -- the exemption that bit the Winograd kernel is for buffer stores with an SGPR soffset only)
                    *reinterpret_cast<f32x4*>(p.out + pix * p.out_ld + cq) = f32x4{x[0], x[1], x[2], x[3]};
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
    if (p.residual) { if (p.stats) epilogue(T_{}, T_{}); else epilogue(T_{}, F_{}); }
    else { if (p.stats) epilogue(F_{}, T_{}); else epilogue(F_{}, F_{}); }
    if (p.stats) {      // wave-uniform
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
            const int slot = (ty * p.tiles_x + tx) * C::WM + wm;
            const int nslots = p.tiles_y * p.tiles_x * C::WM;
            double* o = p.stats + (((size_t)b * nslots + slot) * p.Cout + cq) * 2;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                o[2 * c] = s1[c];
                o[2 * c + 1] = s2[c];
            }
        }
    }
}

template <class C>
inline int conv_launch(const ConvParams& p, hipStream_t stream) {
    dim3 grid(p.tiles_x * p.tiles_y * p.B, ceil_div(p.Cout, C::BN));
    hipLaunchKernelGGL(conv_kernel<C>, grid, dim3(256), 0, stream, p);
    return check_launch();
}

// tile-shape ids used by the dispatcher (WM x WN x MT [x WK]); the S* shapes split K over the 4 waves
// and exist for the low-resolution levels, where the big tiles would leave most CUs idle.
enum { WCFG_1x4x4 = 0, WCFG_2x2x2 = 1, WCFG_4x1x1 = 2, WCFG_1x4x2 = 3, WCFG_2x2x1 = 4, WCFG_S32 = 5, WCFG_S64 = 6,
       WCFG_WINO = 7, WCFG_SMALL = 8, WCFG_SMALL_VALU = 9, WCFG_CIN8 = 10, WCFG_SPLIT = 11, WCFG_SMALL_GEMM = 12 };
int conv_dispatch_small(const ConvParams& p, hipStream_t stream, bool valu_form);   // conv_small.hip: 3x3 stride 1, cout <= 4
int conv_dispatch_small_gemm(const ConvParams& p, hipStream_t stream);                // conv_small.hip: ... cout <= 3 as one GEMM over (tap, cout) columns

int conv_dispatch_k3s1(int logtw, int wcfg, const ConvParams& p, hipStream_t stream);
int conv_dispatch_k1s1(int logtw, int wcfg, const ConvParams& p, hipStream_t stream);
int conv_dispatch_k3s2(int logtw, int wcfg, const ConvParams& p, hipStream_t stream);
int conv_dispatch_exp(int logtw, int wcfg, const ConvParams& p, hipStream_t stream);   // experimental ids >= 16

#define CDX_CONV_DISPATCH_BODY(KS, ST)                                                         \
    switch (wcfg * 8 + logtw) {                                                                \
        CDX_CONV_CASES(KS, ST)                                                                 \
        default: return CDX_ENOTSUP;                                                           \
    }
#define CDX_CONV_CASE(KS, ST, LT, W, ...) \
    case (W) * 8 + (LT): return conv_launch<ConvCfg<KS, ST, LT, __VA_ARGS__>>(p, stream);

}  // namespace cdx
