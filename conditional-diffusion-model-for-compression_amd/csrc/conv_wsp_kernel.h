// PERSISTENT, WAVE-SPECIALISED form of the SPLIT convolution (float32 products as 3 fp16 MFMAs on hi | lo operands,
// conv16_kernel.h) for the layers that carry the FLOPs: stride 1, >= 32 pixels wide, output channels in blocks of 128,
// at least two 32-channel input chunks.  VERDICT r02 item 2: with four homogeneous waves per workgroup three quarters of
// the vector work executed while the SIMD's matrix pipe sat idle (SQ_VALU_MFMA_COEXEC 0.235 of the busy cycles), and a
// wave's prologue (first halo load + staging) and epilogue (64 stores, GroupNorm sums) had nothing of its own to hide under.
//
// One workgroup of 8 waves per CU (256 VGPRs per wave, 128 KiB of LDS), alive for the whole launch, walks a strided list of
// 128-pixel x 128-channel tiles:
//   * waves 0-3, the MFMA waves (one per SIMD): NOTHING but LDS operand reads (software-pipelined one group ahead), the weight
//     ring from L2 and MFMAs -- 216 per chunk -- into accumulators that start at ZERO; after a tile's last chunk they drop
//     the raw accumulators into an LDS tile (64 ds_write_b32 per lane) and carry on with the next tile's first chunk, which
//     is already staged;
//   * waves 4-7, the producers: everything else.  Per chunk slot: GroupNorm / SiLU / scaling / hi | lo split of the NEXT
//     chunk (which may belong to the next tile) into the other halo image, the halo loads of the chunk after that, and --
//     in the first slot of a tile -- the previous tile's epilogue from the LDS tile: x 2^-S, + bias + temb + residual (at
//     their own scale: no accumulator-init overflow question), 16-byte stores, GroupNorm sums (float64) and amax.
// Hand-off: the two halo images (as before) and the LDS output tile, all ordered by ONE workgroup barrier per chunk slot:
// the MFMA waves write the output tile before the barrier that ends a tile's last slot; the producers read it in the next
// slot, which ends before the MFMA waves can write it again (>= 2 chunks per tile).
// Summation order inside a tile = conv16_kernel.h's; bias / temb / residual are added after the products instead of seeding
// the accumulators (a different rounding of the same sum: tolerances unchanged, bits differ from CDX_TILE_SPLIT's 4-wave tile).
// GroupNorm-sum slots: 4 per tile (one per producer wave = tile row).
#pragma once
#include "conv16_kernel.h"

namespace cdx {

// ABL (timing ablations, tuning build only; wrong results): 1 = producers stage only the first two slots and write no outputs
// (the MFMA waves' own bound), 2 = MFMA waves issue no MFMAs / LDS reads (the producers' own bound), 4 = no weight refills
template <int KS_, int ABL_ = 0>
struct WspCfg {
    static constexpr int KS = KS_, TAPS = KS * KS, PAD = KS / 2, ABL = ABL_;
    using H = _Float16;
    static constexpr int KC = 32, PSH = 2 * KC + 8;                // hi | lo | pad: 144 B per pixel
    static constexpr int TW = 32, TH = 4, LOGTW = 5, MT = 4;
    static constexpr int HH = TH + KS - 1, HW = TW + KS - 1;
    static constexpr int RSH = ((HW * PSH + 127) / 128) * 128;
    static constexpr int LDS_HALVES = HH * RSH;
    static constexpr int NPIX = HH * HW;
    static constexpr int NPASS = (NPIX + 63) / 64;
    static constexpr int GPC = TAPS * 2;
    static constexpr int PF = GPC < 3 ? GPC : 3;
    static constexpr int OUT_LD = 136;                             // floats per pixel of the LDS output tile (128 + 8: the two lane
                                                                   // halves of a ds_write_b32 land 32 banks apart)
    static constexpr int LDS_BYTES = 2 * LDS_HALVES * 2 + 128 * OUT_LD * 4;
    static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
};

template <class C, int STG>
__global__ __launch_bounds__(512, 2) void conv_wsp_kernel(const Conv16Params p, const int ntiles) {
    constexpr bool kGN = STG == 1 || STG == 2, kSILU = STG == 2 || STG == 3;
    using H = typename C::H;
    using h8 = __attribute__((ext_vector_type(8))) H;
    constexpr int KC = C::KC, PSH = C::PSH, RSH = C::RSH, TAPS = C::TAPS, NPASS = C::NPASS, GPC = C::GPC, PF = C::PF, MT = C::MT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    H* halo = reinterpret_cast<H*>(smem);                                              // two images
    float* outbuf = reinterpret_cast<float*>(smem + 2 * C::LDS_HALVES * sizeof(H));    // [128 pixels][OUT_LD]

    const int tid = threadIdx.x & 255;
    const bool producer = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) != 0;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = blockIdx.y;                                     // 128-channel block
    // this workgroup's tiles: blockIdx.x, + gridDim.x, ...  (neighbouring workgroups work on neighbouring tiles at the same time)
    const int count = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int S = count * p.nchunks;                               // chunk slots
    const int tpi = p.tiles_x * p.tiles_y;

    if (producer) {
        // =================================================== producer waves ===================================================
        if constexpr ((C::ABL & 8) != 0) __builtin_amdgcn_s_setprio(3);      // (experiment: producers win issue arbitration)
        const int q = tid & 3, pl = tid >> 2;
        const int Hv = p.Hin << p.ups, Wv = p.Win << p.ups;
        // ---- loader state of the tile whose chunks are being LOADED ----
        int soff[NPASS];
        unsigned vmask = 0;
        int lb = 0;                                                // image of the loader's tile
        auto setup_tile = [&](int tile) {
            const int tx = tile % p.tiles_x;
            const int r = tile / p.tiles_x;
            const int ty = r % p.tiles_y;
            lb = r / p.tiles_y;
            const int iy0 = ty * C::TH - C::PAD, ix0 = tx * C::TW - C::PAD;
            vmask = 0;
#pragma unroll
            for (int i = 0; i < NPASS; ++i) {
                const int hp = i * 64 + pl;
                const int hy = hp / C::HW, hx = hp - hy * C::HW;
                const int iy = iy0 + hy, ix = ix0 + hx;
                const bool ok = hp < C::NPIX && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv;
                soff[i] = ok ? (iy >> p.ups) * p.Win + (ix >> p.ups) : 0;
                vmask |= ok ? (1u << i) : 0u;
            }
        };
        // ---- the chunk that is loaded but not yet staged: data + what its staging needs ----
        float pre[NPASS][8];
        f32x4 gsc[2], gsh[2];
        unsigned pmask = 0;                                        // vmask of ITS tile
        bool cvalid = false;
        float pa = 1.f, pk = -1.44269504088896341f;                // activation scale / SiLU constant of ITS image
        auto issue_loads = [&](int s) {                            // stream item s = (tile s / nchunks of mine, chunk s % nchunks)
            const int chunk = s % p.nchunks;
            if (chunk == 0) setup_tile((int)blockIdx.x + (s / p.nchunks) * (int)gridDim.x);
            const int src = chunk >= p.nchunk0;
            const unsigned cs = (unsigned)p.csrc[src];
            const size_t img = (size_t)p.Hin * p.Win * cs * 4u;
            const __amdgpu_buffer_rsrc_t srs = buf_rsrc(static_cast<const char*>(p.src[src]) + (size_t)lb * img, (unsigned)img);
            const int cc = src ? chunk - p.nchunk0 : chunk;
            const int cl = cc * KC + q * 8;
            cvalid = cl < (int)cs;
            pmask = vmask;
            const ActScale asc = act_scale_of<true, kGN>(p, lb);
            pa = asc.a;
            pk = asc.ksilu;
            const unsigned so = (unsigned)cc * (unsigned)KC * 4u;
#pragma unroll
            for (int i = 0; i < NPASS; ++i) {
                const unsigned vo = ((unsigned)soff[i] * cs + 8u * (unsigned)q) * 4u;      // (invalid slots: pixel 0, zeroed by the select)
                const f32x4 v0 = buf_load4(srs, vo, so), v1 = buf_load4(srs, vo + 16u, so);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pre[i][e] = v0[e];
                    pre[i][4 + e] = v1[e];
                }
            }
            if constexpr (kGN) {
                const int cg = cvalid ? (src ? p.csrc[0] : 0) + cl : 0;
                const float* gs = p.gscale + (size_t)lb * p.ctot + cg;
                const float* gh = p.gshift + (size_t)lb * p.ctot + cg;
                gsc[0] = *reinterpret_cast<const f32x4*>(gs);
                gsc[1] = *reinterpret_cast<const f32x4*>(gs + 4);
                gsh[0] = *reinterpret_cast<const f32x4*>(gh);
                gsh[1] = *reinterpret_cast<const f32x4*>(gh + 4);
            }
        };
        auto stage = [&](H* img) {                                 // the pending chunk -> halo image (hi | lo)
#pragma unroll
            for (int i = 0; i < NPASS; ++i) {
                const int hp = i * 64 + pl;
                const int hy = hp / C::HW, hx = hp - hy * C::HW;
                const bool ok = cvalid && ((pmask >> i) & 1u);
                using u4 = __attribute__((ext_vector_type(4))) unsigned;
                u4 o, ol;
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    float v[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        v[k] = pre[i][e + k];
                        if constexpr (kGN) v[k] = fmaf(v[k], gsc[(e + k) >> 2][(e + k) & 3], gsh[(e + k) >> 2][(e + k) & 3]);
                        else v[k] *= pa;
                        if constexpr (kSILU) v[k] = silu16_f(v[k], pk);
                        v[k] = ok ? v[k] : 0.f;
                    }
                    o[e / 2] = pack_hi_f16(v[0], v[1]);
                    ol[e / 2] = pack_lo_f16(o[e / 2], v[0], v[1]);
                }
                if ((i + 1) * 64 <= C::NPIX || hp < C::NPIX) {
                    const int off = hy * RSH + hx * PSH + q * 8;
                    *reinterpret_cast<u4*>(&img[off]) = o;
                    *reinterpret_cast<u4*>(&img[off + KC]) = ol;
                }
            }
        };
        // ---- epilogue of a finished tile from the LDS output tile: lane & 31 = channel quad, tid >> 5 = 16-pixel group.
        // Cut into 4 PORTIONS of 4 pixels, one per chunk slot of the NEXT tile, each with its residual loads issued one portion
        // ahead (a portion that waited for its own HBM loads cost ~1.5 us -- half a chunk slot -- four times per tile).
        const int cq4 = lane & 31, pg = tid >> 5;
        struct Epi {
            int b, ty, tx;
            float un;
            f32x4 add;
            bool rowok, nok;
            size_t rowpix;
            int ox0;
        } ep{};
        f32x4 rres[4];                                             // the next portion's residual values
        double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
        float am = 0.f;
        const int n0 = nb * 128 + cq4 * 4;
        auto epi_begin = [&](int tile) {                           // tile geometry + additive terms (before its portion 0)
            ep.tx = tile % p.tiles_x;
            const int r = tile / p.tiles_x;
            ep.ty = r % p.tiles_y;
            ep.b = r / p.tiles_y;
            ep.un = act_scale_of<true, kGN>(p, ep.b).un;
            ep.nok = n0 < p.Cout;                                  // (cout is a multiple of 4)
            ep.add = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ep.nok) {
                if (p.bias) ep.add = *reinterpret_cast<const f32x4*>(p.bias + n0);
                if (p.temb) {
                    const float* tp = p.temb + (size_t)ep.b * p.temb_ld + n0;      // (rows are only 4-byte aligned: offset by the layer's column)
                    ep.add += f32x4{tp[0], tp[1], tp[2], tp[3]};
                }
            }
            const int oy = ep.ty * C::TH + (pg >> 1);
            ep.ox0 = ep.tx * C::TW + (pg & 1) * 16;
            ep.rowok = oy < p.Hout && ep.nok;
            ep.rowpix = ((size_t)ep.b * p.Hout + (ep.rowok ? oy : 0)) * p.Wout;
#pragma unroll
            for (int c = 0; c < 4; ++c) s1[c] = s2[c] = 0.0;
            am = 0.f;
        };
        auto epi_prefetch = [&](int j) {                           // residual values of portion j
            if (!p.residual) return;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ox = ep.ox0 + 4 * j + k;
                const bool ok = ep.rowok && ox < p.Wout;
                rres[k] = ok ? *reinterpret_cast<const f32x4*>(static_cast<const float*>(p.residual) + (ep.rowpix + ox) * p.Cout + n0)
                             : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        auto epi_portion = [&](int j) {
            const bool want = p.stats || p.amax_out;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ox = ep.ox0 + 4 * j + k;
                if (ep.rowok && ox < p.Wout) {
                    const int m = (pg >> 1) * 32 + (pg & 1) * 16 + 4 * j + k;
                    f32x4 v = *reinterpret_cast<const f32x4*>(&outbuf[m * C::OUT_LD + cq4 * 4]);
                    v = v * ep.un + ep.add;
                    if (p.residual) v += rres[k];
                    *reinterpret_cast<f32x4*>(static_cast<float*>(p.out) + (ep.rowpix + ox) * p.out_ld + n0) = v;
                    if (want) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const double d = (double)v[c];
                            s1[c] += d;
                            s2[c] = fma(d, d, s2[c]);
                            am = fmaxf(am, fabsf(v[c]));
                        }
                    }
                }
            }
        };
        auto epi_end = [&](int tile) {
            if (p.stats) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {                      // the wave's two pixel groups = one tile row
                    s1[c] += __shfl_xor(s1[c], 32);
                    s2[c] += __shfl_xor(s2[c], 32);
                }
                if (lane < 32 && ep.nok) {
                    const int slot = (ep.ty * p.tiles_x + ep.tx) * 4 + wv;
                    const int nslots = tpi * 4;
                    double* o = p.stats + (((size_t)ep.b * nslots + slot) * p.Cout + n0) * 2;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        o[2 * c] = s1[c];
                        o[2 * c + 1] = s2[c];
                    }
                }
            }
            if (p.amax_out) {
                float m = am;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
                if (lane == 0) amax_publish(p.amax_out, ep.b, (unsigned)(tile * 4 + wv) * 5u + (unsigned)nb, m);
            }
        };
        auto tile_of = [&](int i) { return (int)blockIdx.x + i * (int)gridDim.x; };

        issue_loads(0);
        stage(halo);
        if (S > 1) issue_loads(1);
        __syncthreads();
        for (int s = 0; s < S; ++s) {
            const int chunk = s % p.nchunks, ti = s / p.nchunks;
            if (s + 1 < S && !((C::ABL & 1) && s >= 1)) {
                stage(halo + ((s + 1) & 1) * C::LDS_HALVES);
                if (s + 2 < S) issue_loads(s + 2);
            }
            if (ti > 0 && chunk < p.nchunks - 1 && !(C::ABL & 1)) {
                // the previous tile's epilogue: portion j in slot j (nchunks - 1) / 4 of this tile -- NEVER in the tile's last slot,
                // at whose end the MFMA waves overwrite the LDS output tile
                for (int j = 0; j < 4; ++j) {
                    if (j * (p.nchunks - 1) / 4 == chunk) {
                        epi_portion(j);
                        if (j < 3) epi_prefetch(j + 1);
                        else epi_end(tile_of(ti - 1));
                    }
                }
            }
            if (chunk == p.nchunks - 1 && !(C::ABL & 1)) {         // this tile finishes with this slot: get ready for its epilogue
                epi_begin(tile_of(ti));
                epi_prefetch(0);
            }
            __syncthreads();
        }
        if constexpr (C::ABL & 1) return;
        for (int j = 0; j < 4; ++j) {                              // the last tile
            epi_portion(j);
            if (j < 3) epi_prefetch(j + 1);
        }
        epi_end(tile_of(count - 1));
        return;
    }

    // ======================================================= MFMA waves =======================================================
    const int li = lane & 31, lh = lane >> 5;
    const int a_base = (li >> C::LOGTW) * RSH + (li & (C::TW - 1)) * PSH + lh * 8;
    const int ntile = nb * 4 + wv;
    const bool nvalid = ntile * 32 < p.Cout;
    constexpr int GH = 1024;                                       // halves per (tap, 16-channel step) group: hi | lo fragments
    const __amdgpu_buffer_rsrc_t wrs = buf_rsrc(static_cast<const H*>(p.w) + ((size_t)(nvalid ? ntile : 0) * p.nchunks * TAPS) * (2 * GH));
    const unsigned wlane = lane * 16u;
    auto wload = [&](unsigned half_off) { return __builtin_bit_cast(h8, buf_load4(wrs, wlane, half_off * 2u)); };
    const int gtot = p.nchunks * GPC;                              // groups per tile: the ring wraps to the tile's first group

    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    h8 ring[PF][2];
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        ring[j][0] = wload(j * GH);
        ring[j][1] = wload(j * GH + 512);
    }
    // operand registers of the group being multiplied and of the next one (read while the current group's MFMAs run)
    h8 a[2][MT], al[2][MT];
    auto read_ops = [&](const H* lds, int g, int slot) __attribute__((always_inline)) {
        const int tap = g >> 1, j = g & 1, ky = tap / C::KS, kx = tap % C::KS;
        int ab = a_base;
        asm volatile("" : "+v"(ab));                               // no cross-tap CSE of LDS reads (conv_kernel.h)
        __builtin_assume((ab & 7) == 0);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            a[slot][t] = *reinterpret_cast<const h8*>(&lds[ab + (t + ky) * RSH + kx * PSH + j * 16]);
            al[slot][t] = *reinterpret_cast<const h8*>(&lds[ab + (t + ky) * RSH + kx * PSH + KC + j * 16]);
        }
    };
    __syncthreads();
    for (int s = 0; s < S; ++s) {
        const int chunk = s % p.nchunks;
        const H* lds = halo + (s & 1) * C::LDS_HALVES;
        if (nvalid && !(C::ABL & 2)) {
            const int gbase = chunk * GPC;
            read_ops(lds, 0, 0);
#pragma unroll
            for (int g = 0; g < GPC; ++g) {
                // the NEXT group's operand reads are issued before this group's MFMAs and fenced there: left alone the scheduler
                // sinks them to just in front of their own MFMAs (same registers, no overlap -- the wave then waits ~150 cycles
                // of LDS latency per 384-cycle group: measured 0.69 ms for the MFMA waves alone against 0.54 of pure MFMA time)
                if (g + 1 < GPC) read_ops(lds, g + 1, (g + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                const h8 bq = ring[g % PF][0], bl = ring[g % PF][1];
                int gi = gbase + g + PF;                           // PF groups ahead, wrapping into the next tile's first groups
                gi = gi >= gtot ? gi - gtot : gi;
                if constexpr (!(C::ABL & 4)) {
                    ring[g % PF][0] = wload((unsigned)gi * GH);
                    ring[g % PF][1] = wload((unsigned)gi * GH + 512);
                }
                const int cur = g & 1;
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[t] = mfma_32x32x16(a[cur][t], bq, acc[t]);
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[t] = mfma_32x32x16(al[cur][t], bq, acc[t]);
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[t] = mfma_32x32x16(a[cur][t], bl, acc[t]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (chunk == p.nchunks - 1) {
                // raw accumulators -> LDS output tile: lane = channel, register r of tile t = pixel 32 t + 8 (r >> 2) + 4 lh + (r & 3)
                float* ob = outbuf + (4 * lh) * C::OUT_LD + wv * 32 + li;
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        ob[(t * 32 + 8 * (r >> 2) + (r & 3)) * C::OUT_LD] = acc[t][r];
                        acc[t][r] = 0.f;
                    }
            }
        }
        __syncthreads();
    }
}

int wsp_cu_count();      // conv_split.hip: CUs of the current device (cached)

template <class C>
inline int conv_wsp_launch(const Conv16Params& p, hipStream_t stream) {
    const int ntiles = p.tiles_x * p.tiles_y * p.B;
    const int nblocks = ceil_div(p.Cout, 128);
    int g = wsp_cu_count() / nblocks;
    g = g < 1 ? 1 : g > ntiles ? ntiles : g;
    dim3 grid(g, nblocks);
    auto go = [&](auto kern) -> int {
        // more dynamic LDS than the default allowance: declare it (idempotent, no sync)
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess) return CDX_ELAUNCH;
        hipLaunchKernelGGL(kern, grid, dim3(512), C::LDS_BYTES, stream, p, ntiles);
        return check_launch();
    };
    switch (p.gn ? (p.silu ? 2 : 1) : (p.silu ? 3 : 0)) {
        case 0: return go(conv_wsp_kernel<C, 0>);
        case 1: return go(conv_wsp_kernel<C, 1>);
        case 2: return go(conv_wsp_kernel<C, 2>);
        default: return go(conv_wsp_kernel<C, 3>);
    }
}

}  // namespace cdx
