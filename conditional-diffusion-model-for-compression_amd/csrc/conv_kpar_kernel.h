// Low-resolution form of the SPLIT convolution (float32 products as 3 fp16 MFMAs on hi | lo operands, conv16_kernel.h):
// layers 8 or 16 pixels wide, stride 1.  At batch 16 an 8x8 x 512-channel layer is 1024 output pixels: 128-channel tiles
// give 64 workgroups of 16 SERIAL input chunks each (measured no faster than the f32 split-K tiles).  Here the parallelism
// comes from the reduction dimension instead:
//   * a workgroup = 64 output pixels (8x8 or 4x16) x ONE 32-channel N-tile -> 256 workgroups for that layer;
//   * its 4 waves split the INPUT CHUNKS (wave w takes chunks w, w+4, ...): each wave stages its own chunk into its OWN
//     halo image in LDS (no workgroup barrier in the main loop: LDS operations of one wave execute in order, so the wave's
//     reads of a chunk follow its writes and precede the next chunk's writes), runs that chunk's 18 x 6 MFMAs, and keeps a
//     partial accumulator for both M-tiles;
//   * the four partial tiles are summed once through LDS in fixed order (wave 0 + 1 + 2 + 3: bits do not depend on timing);
//     wave w finishes registers 4w..4w+3 of both M-tiles (bias + temb + residual enter through wave 0's accumulator init),
//     stores them with the usual quad transposes and writes its own GroupNorm-sum slot (4 slots per tile).
// Same staging (GroupNorm scale/shift, SiLU, saturate, split), operand layouts, weight image and epilogue numerics as
// conv16_kernel.h SPLIT; summation order differs (chunks interleaved over waves), the tolerance does not.
#pragma once
#include "conv16_kernel.h"

namespace cdx {

// SPLIT = 0: the same tile for the 16-bit storage modes (fp16, BF = 1: bf16) -- one operand plane, one MFMA per (tap, 16
// channels, M-tile), 16-bit sources / residual / output, GroupNorm sums from the float32 accumulators (cfg5's 8 x 8 level:
// the 64-pixel x 128-channel tiles of conv16_kernel.h leave 3/4 of the CUs without a workgroup there).
template <int KS_, int LOGTW_, int SPLIT_ = 1, int BF_ = 0>
struct KparCfg {
    static constexpr int KS = KS_, LOGTW = LOGTW_, TAPS = KS * KS, PAD = KS / 2, SPLIT = SPLIT_, BF = BF_;
    static_assert(!(SPLIT && BF), "the split operands are fp16");
    using H = std::conditional_t<BF != 0, __bf16, _Float16>;
    static constexpr int PLANES = SPLIT ? 2 : 1;
    static constexpr int TW = 1 << LOGTW, MT = 2, BM = 64, TH = BM / TW, RPM = 32 / TW;
    static constexpr int KC = 32, PSH = PLANES * KC + 8;
    static constexpr int HH = TH + KS - 1, HW = TW + KS - 1;
    static constexpr int RSH = ((HW * PSH + 127) / 128) * 128;
    static constexpr int LDS_HALVES = HH * RSH;                    // per WAVE
    static constexpr int NPIX = HH * HW;
    static constexpr int NPASS = (NPIX + 15) / 16;                 // 16 pixel slots x 4 channel octets per pass and wave
    static constexpr int GPC = TAPS * 2;
    static constexpr int PF = GPC < 3 ? GPC : 3;
    static constexpr int SCRATCH_HALVES = 4 * 32 * 64 * 2;         // the halo images double as the reduction scratch (32 KiB of floats)
    static constexpr int LDS_TOTAL = 4 * LDS_HALVES > SCRATCH_HALVES ? 4 * LDS_HALVES : SCRATCH_HALVES;
    static_assert(TW <= 16 && LDS_TOTAL * 2 <= 80 * 1024, "two workgroups per CU");
};

template <class C, int STG>
__global__ __launch_bounds__(256, 2) void conv_kpar_kernel(const Conv16Params p) {
    constexpr bool kGN = STG == 1 || STG == 2, kSILU = STG == 2 || STG == 3;
    constexpr int KC = C::KC, PSH = C::PSH, RSH = C::RSH, TAPS = C::TAPS, MT = C::MT, NPASS = C::NPASS, GPC = C::GPC, PF = C::PF;
    using H = typename C::H;
    using h8 = __attribute__((ext_vector_type(8))) H;
    using h4 = __attribute__((ext_vector_type(4))) H;
    constexpr unsigned ES = C::SPLIT ? 4u : 2u;                    // bytes per source / residual / output element
    __shared__ __attribute__((aligned(16))) H lds_all[C::LDS_TOTAL];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    H* lds = lds_all + wv * C::LDS_HALVES;                         // this wave's halo image

    int bx = blockIdx.x;
    const int tx = bx % p.tiles_x;
    bx /= p.tiles_x;
    const int ty = bx % p.tiles_y;
    const int b = bx / p.tiles_y;
    const int oy0 = ty * C::TH, ox0 = tx * C::TW;
    const int iy0 = oy0 - C::PAD, ix0 = ox0 - C::PAD;
    const int Hv = p.Hin << p.ups, Wv = p.Win << p.ups;

    // ---- loader (per wave): lane -> (pixel slot pl of 16, channel octet q of 4) ----
    const int q = lane & 3, pl = lane >> 2;
    int soff[NPASS];                                               // pixel index inside image b
    unsigned vmask = 0;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int hp = i * 16 + pl;
        const int hy = hp / C::HW, hx = hp - hy * C::HW;
        const int iy = iy0 + hy, ix = ix0 + hx;
        const bool ok = hp < C::NPIX && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv;
        soff[i] = ok ? (iy >> p.ups) * p.Win + (ix >> p.ups) : 0;
        vmask |= ok ? (1u << i) : 0u;
    }
    const ActScale asc = act_scale_of<C::SPLIT != 0, kGN>(p, b);
    float pre[NPASS][8];
    f32x4 gsc[2], gsh[2];
    bool cvalid = false;
    // per-image buffer resource; padding pixels / slots past the halo / channel octets past the source's end read pixel 0 (in range)
    // and are zeroed by a select after the activation: conv16_kernel.h
    unsigned voff[NPASS];
    __amdgpu_buffer_rsrc_t srs;
    int cur_src = -1;
    auto bind_source = [&](int sidx) {
        const unsigned cs = (unsigned)p.csrc[sidx];
        const size_t img = (size_t)p.Hin * p.Win * cs * ES;
        srs = buf_rsrc(static_cast<const char*>(p.src[sidx]) + (size_t)b * img, (unsigned)img);
#pragma unroll
        for (int i = 0; i < NPASS; ++i) voff[i] = ((unsigned)soff[i] * cs + 8u * (unsigned)q) * ES;      // (soff = 0 for invalid slots)
        cur_src = sidx;
    };
    auto issue_loads = [&](int chunk) {
        const int s = chunk >= p.nchunk0;
        if (s != cur_src) bind_source(s);
        const int cc = s ? chunk - p.nchunk0 : chunk;
        const int cl = cc * KC + q * 8;
        cvalid = cl < p.csrc[s];
        const unsigned so = (unsigned)cc * (unsigned)KC * ES;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const unsigned vo = voff[i];
            if constexpr (C::SPLIT) {
                const f32x4 v0 = buf_load4(srs, vo, so), v1 = buf_load4(srs, vo + 16u, so);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pre[i][e] = v0[e];
                    pre[i][4 + e] = v1[e];
                }
            } else {
                const h8 v = __builtin_bit_cast(h8, buf_load4(srs, vo, so));
#pragma unroll
                for (int e = 0; e < 8; ++e) pre[i][e] = (float)v[e];
            }
        }
        if constexpr (kGN) {
            const int cg = cvalid ? (s ? p.csrc[0] : 0) + cl : 0;
            const float* gs = p.gscale + (size_t)b * p.ctot + cg;
            const float* gh = p.gshift + (size_t)b * p.ctot + cg;
            gsc[0] = *reinterpret_cast<const f32x4*>(gs);
            gsc[1] = *reinterpret_cast<const f32x4*>(gs + 4);
            gsh[0] = *reinterpret_cast<const f32x4*>(gh);
            gsh[1] = *reinterpret_cast<const f32x4*>(gh + 4);
        }
    };
    auto stage_pass = [&](int i) {                                 // GroupNorm / SiLU / saturate / split, one pixel slot x 8 channels
        const int hp = i * 16 + pl;
        const int hy = hp / C::HW, hx = hp - hy * C::HW;
        const bool ok = cvalid && ((vmask >> i) & 1u);            // (padding holds a dummy read: zeroed after the activation)
        using u4 = __attribute__((ext_vector_type(4))) unsigned;
        u4 o, ol;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            float v[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                v[k] = pre[i][e + k];
                if constexpr (kGN) v[k] = fmaf(v[k], gsc[(e + k) >> 2][(e + k) & 3], gsh[(e + k) >> 2][(e + k) & 3]);
                else if constexpr (C::SPLIT) v[k] *= asc.a;
                if constexpr (kSILU) v[k] = silu16_f(v[k], asc.ksilu);
                v[k] = ok ? v[k] : 0.f;
            }
            if constexpr (C::BF) {
                using b2 = __attribute__((ext_vector_type(2))) __bf16;
                o[e / 2] = __builtin_bit_cast(unsigned, b2{(__bf16)v[0], (__bf16)v[1]});
            } else {
                o[e / 2] = pack_hi_f16(v[0], v[1]);                // hi pair, lo pair = RN(v - hi) by mixed-precision FMA
                if constexpr (C::SPLIT) ol[e / 2] = pack_lo_f16(o[e / 2], v[0], v[1]);
            }
        }
        // slots past the halo's last pixel dump into the pad bytes of the lane's pass-0 pixel (no exec-mask branch)
        const bool real = (i + 1) * 16 <= C::NPIX || hp < C::NPIX;
        const int off = real ? hy * RSH + hx * PSH + q * 8 : (pl / C::HW) * RSH + (pl % C::HW) * PSH + C::PLANES * KC;
        if constexpr (C::SPLIT) {
            *reinterpret_cast<u4*>(&lds[off]) = o;
            if (real) *reinterpret_cast<u4*>(&lds[off + KC]) = ol;
        } else {
            if (real) *reinterpret_cast<u4*>(&lds[off]) = o;       // (a 16-byte dump would not fit the 16 pad bytes' alignment: skip it)
        }
    };

    // ---- MFMA operand addressing ----
    const int li = lane & 31, lh = lane >> 5;
    const int a_base = (li >> C::LOGTW) * RSH + (li & (C::TW - 1)) * PSH + lh * 8;
    const int ntile = blockIdx.y;
    constexpr int GH = 512 * C::PLANES;                            // halves per group: hi | lo fragments (SPLIT)
    const H* __restrict__ wp = static_cast<const H*>(p.w) + ((size_t)ntile * p.nchunks * TAPS) * (2 * GH) + lane * 8;
    auto wfrag = [&](int chunk, int g, int plane) {                // fragment of (chunk, group) -- reads past the image end hit the zero pad
        return *reinterpret_cast<const h8*>(wp + ((size_t)chunk * GPC + g) * GH + plane * 512);
    };

    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int n = ntile * 32 + li;
    // SPLIT (float32 results): bias + temb are added by the epilogue's FMA, not carried in the accumulators (conv16_kernel.h: a bias
    // that dwarfs the products would be re-rounded at its own ulp by every MFMA); the residual keeps entering through the init.
#ifdef CDX_TUNING
    const bool bias_late = C::SPLIT && !(p.abl & 2048);
#else
    constexpr bool bias_late = C::SPLIT != 0;
#endif
    float addv = 0.f;
    if (bias_late && n < p.Cout) {
        addv = p.bias ? p.bias[n] : 0.f;
        if (p.temb) addv += p.temb[(size_t)b * p.temb_ld + n];
    }
    if (wv == 0 && n < p.Cout && !asc.late && (p.residual || !bias_late)) {      // (bias + temb +) residual through wave 0's accumulator init
        const float inv = asc.inv;
        float add = 0.f;
        if (!bias_late) {
            add = p.bias ? p.bias[n] : 0.f;
            if (p.temb) add += p.temb[(size_t)b * p.temb_ld + n];
            add *= inv;
        }
        if (p.residual) {
            // oy0 / ox0 are a workgroup origin INSIDE the image (the grid covers ceil(H / TH) x ceil(W / TW) tiles), so first < total
            // always; the guard keeps the unmasked loads below safe against a caller-side change of that (conv16_kernel.h met the
            // underflowed form as a GPU memory fault in round 3)
            const size_t first = (((size_t)b * p.Hout + oy0) * p.Wout + ox0) * p.Cout;
            const size_t total = (size_t)p.B * p.Hout * p.Wout * p.Cout;
            const size_t left = first < total ? (total - first) * ES : 1;
            const __amdgpu_buffer_rsrc_t rr = buf_rsrc(static_cast<const char*>(p.residual) + (first < total ? first : 0) * ES, left > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)left);
            const unsigned voffr = ((unsigned)(4 * lh) * (unsigned)p.Cout + (unsigned)n) * ES;
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int mb = t * 32 + 8 * (r >> 2) + (r & 3);
                    const unsigned pix = (unsigned)(mb >> C::LOGTW) * (unsigned)p.Wout + (unsigned)(mb & (C::TW - 1));
                    float rv;
                    if constexpr (C::SPLIT) rv = buf_load1(rr, voffr, pix * (unsigned)p.Cout * ES);
                    else rv = (float)__builtin_bit_cast(H, __builtin_amdgcn_raw_buffer_load_b16(rr, voffr, pix * (unsigned)p.Cout * ES, 0));
                    acc[t][r] = fmaf(rv, inv, add);
                }
        } else {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = add;
        }
    }

    // ---- main loop: this wave's chunks ----
    if (wv < p.nchunks) {
        h8 ring[PF][C::PLANES];
#pragma unroll
        for (int j = 0; j < PF; ++j)
#pragma unroll
            for (int pl_ = 0; pl_ < C::PLANES; ++pl_) ring[j][pl_] = wfrag(wv, j, pl_);
        issue_loads(wv);
        for (int chunk = wv; chunk < p.nchunks; chunk += 4) {
#pragma unroll
            for (int i = 0; i < NPASS; ++i) stage_pass(i);
            if (chunk + 4 < p.nchunks) issue_loads(chunk + 4);
#pragma unroll
            for (int g = 0; g < GPC; ++g) {
                const int tap = g >> 1, j = g & 1, ky = tap / C::KS, kx = tap % C::KS;
                int ab = a_base;
                asm volatile("" : "+v"(ab));
                __builtin_assume((ab & 7) == 0);
                h8 a[MT], al[MT];
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    a[t] = *reinterpret_cast<const h8*>(&lds[ab + (t * C::RPM + ky) * RSH + kx * PSH + j * 16]);
                    if constexpr (C::SPLIT) al[t] = *reinterpret_cast<const h8*>(&lds[ab + (t * C::RPM + ky) * RSH + kx * PSH + KC + j * 16]);
                }
                const h8 bq = ring[g % PF][0];
                h8 bl;
                if constexpr (C::SPLIT) bl = ring[g % PF][1];
                // refill: PF groups ahead in THIS WAVE's sequence (the next chunk of the wave is chunk + 4)
                const int gn = g + PF < GPC ? g + PF : g + PF - GPC;
                const int cn = g + PF < GPC ? chunk : (chunk + 4 < p.nchunks ? chunk + 4 : chunk);      // (last chunk: a harmless re-read)
#pragma unroll
                for (int pl_ = 0; pl_ < C::PLANES; ++pl_) ring[g % PF][pl_] = wfrag(cn, gn, pl_);
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[t] = mfma_32x32x16(a[t], bq, acc[t]);
                if constexpr (C::SPLIT) {
#pragma unroll
                    for (int t = 0; t < MT; ++t) acc[t] = mfma_32x32x16(al[t], bq, acc[t]);
#pragma unroll
                    for (int t = 0; t < MT; ++t) acc[t] = mfma_32x32x16(a[t], bl, acc[t]);
                }
            }
        }
    }

    // ---- cross-wave reduction: scratch[wave][t][r][lane] (float), fixed order ----
    __syncthreads();                                               // every wave is done with its halo image
    float* scratch = reinterpret_cast<float*>(lds_all);
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) scratch[((wv * MT + t) * 16 + r) * 64 + lane] = acc[t][r];
    __syncthreads();
    const int q4 = li & 3;
    const int cq = ntile * 32 + (li & ~3);
    const bool quad_ok = cq < p.Cout;
    const float un = asc.un;
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    float am = 0.f;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        float x[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int r = 4 * wv + c;                               // this wave finishes registers 4 wv .. 4 wv + 3
            float v = scratch[((0 * MT + t) * 16 + r) * 64 + lane];
#pragma unroll
            for (int s = 1; s < 4; ++s) v += scratch[((s * MT + t) * 16 + r) * 64 + lane];
            x[c] = (asc.late && bias_late) ? v * un : fmaf(v, un, addv);      // (lane = channel n here: before the transpose)
        }
        quad_transpose(x, q4);                                     // pixel 8 wv + q4 (+ 4 lh) of tile t, channels cq .. cq + 3
        const int m = t * 32 + 8 * wv + q4 + 4 * lh;
        const int oy = oy0 + (m >> C::LOGTW), ox = ox0 + (m & (C::TW - 1));
        if (quad_ok && oy < p.Hout && ox < p.Wout) {
            const size_t pix = ((size_t)b * p.Hout + oy) * p.Wout + ox;
            if (asc.late) {                                        // (SPLIT, rare: ActScale) additive terms at their own scale
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (cq + c < p.Cout) {
                        float add = p.bias ? p.bias[cq + c] : 0.f;
                        if (p.temb) add += p.temb[(size_t)b * p.temb_ld + cq + c];
                        if (p.residual) add += static_cast<const float*>(p.residual)[pix * p.Cout + cq + c];
                        x[c] += add;
                    }
                }
            }
            if constexpr (C::SPLIT) *reinterpret_cast<f32x4*>(static_cast<float*>(p.out) + pix * p.out_ld + cq) = f32x4{x[0], x[1], x[2], x[3]};
            else *reinterpret_cast<h4*>(static_cast<H*>(p.out) + pix * p.out_ld + cq) = h4{(H)x[0], (H)x[1], (H)x[2], (H)x[3]};
            if (p.stats || p.amax_out) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float xs = cq + c < p.Cout ? x[c] : 0.f;
                    const double d = (double)xs;
                    s1[c] += d;
                    s2[c] = fma(d, d, s2[c]);
                    am = fmaxf(am, fabsf(xs));
                }
            }
        }
    }
    if (p.amax_out) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) am = fmaxf(am, __shfl_xor(am, off));
        if (lane == 0) amax_publish(p.amax_out, b, (blockIdx.x * 4 + wv) * 5 + blockIdx.y, am);
    }
    if (p.stats) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            s1[c] += __shfl_xor(s1[c], 1);
            s2[c] += __shfl_xor(s2[c], 1);
            s1[c] += __shfl_xor(s1[c], 2);
            s2[c] += __shfl_xor(s2[c], 2);
            s1[c] += __shfl_xor(s1[c], 32);
            s2[c] += __shfl_xor(s2[c], 32);
        }
        if (lh == 0 && q4 == 0 && quad_ok) {
            const int slot = (ty * p.tiles_x + tx) * 4 + wv;        // one slot per (tile, finishing wave)
            const int nslots = p.tiles_y * p.tiles_x * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (cq + c < p.Cout) {
                    double* o = p.stats + (((size_t)b * nslots + slot) * p.Cout + cq + c) * 2;
                    o[0] = s1[c];
                    o[1] = s2[c];
                }
        }
    }
}

template <class C>
inline int conv_kpar_launch(const Conv16Params& p, hipStream_t stream) {
    dim3 grid(p.tiles_x * p.tiles_y * p.B, ceil_div(p.Cout, 32));
    switch (p.gn ? (p.silu ? 2 : 1) : (p.silu ? 3 : 0)) {
        case 0: hipLaunchKernelGGL((conv_kpar_kernel<C, 0>), grid, dim3(256), 0, stream, p); break;
        case 1: hipLaunchKernelGGL((conv_kpar_kernel<C, 1>), grid, dim3(256), 0, stream, p); break;
        case 2: hipLaunchKernelGGL((conv_kpar_kernel<C, 2>), grid, dim3(256), 0, stream, p); break;
        default: hipLaunchKernelGGL((conv_kpar_kernel<C, 3>), grid, dim3(256), 0, stream, p); break;
    }
    return check_launch();
}

}  // namespace cdx
