// Winograd F(2x2, 3x3) convolution on the gfx950 fp32 matrix pipe: 16 transform-domain products per 2x2 output
// tile instead of 36 direct ones -> 2.25x fewer v_mfma_f32_32x32x2_f32 for the same float32 result
// (Lavin & Gray 2016;  Y = A^T [ (G g G^T) (.) (B^T d B) ] A,  B^T, G, A^T below).
//
// Mapping (why this fits CDNA4): the usual obstacles are 16 accumulator sets per output block and a 4x larger
// transformed-input image.  Here the 512-entry unified register file of a SIMD holds them:
//   * a workgroup (8 waves, two per SIMD) owns 4 x 32 output pixels = 32 Winograd tiles of one image x 128 output
//     channels; wave (wn, wh) owns 32 channels and V rows {2wh, 2wh+1}: acc[8 xi][32 tiles x 32 ch] = 128 registers;
//   * the input halo (6 x 34 pixels x 32 channels, GroupNorm/SiLU/upsample/concat applied while staging, exactly
//     as in conv_kernel.h) sits in LDS ONCE; each lane (tile = lane & 31, channel half = lane >> 5) reads its
//     patch rows and computes B^T d B IN REGISTERS -- no transformed image in LDS;
//   * weights U = G g G^T are transformed and packed on the host so that a lane's 16-B fragment carries four
//     xi of one channel: [ntile][chunk][s][e][xiq][lane][4] (fragment-ordered, L2-resident, ring-prefetched);
//   * the output transform A^T M A is in-register too (the accumulators of a lane share (tile, channel));
//   * LDS is double buffered: the next chunk is written while this chunk's MFMAs run, one barrier per chunk.
// (Earlier forms -- one 512-register wave per SIMD, and a persistent variant -- are in the git history of round 1;
// both measured slower)
// Numerics: float32 throughout; transform constants are exact (0, +-1, +-1/2) so the only extra error over the
// direct kernel is the reassociation of the sum (measured in tests against float64).
#pragma once
#include "conv_kernel.h"

namespace cdx {

template <int PF_, int OPT_>
struct WinoCfg {
    static constexpr int PF = PF_, OPT = OPT_;
    static constexpr int KC = CDX_CONV_KC, PS = KC + 4;
    static constexpr int TW = 32, TH = 4, LOGTW = 5;             // output pixels per workgroup: 4 rows x 32 cols
    static constexpr int HH = TH + 2, HW = TW + 2;
    static constexpr int RS = ((HW * PS + 63) / 64) * 64;
    static constexpr int BUF_FLOATS = HH * RS;
    static constexpr int NPIX = HH * HW;
    static constexpr int NPASS = (NPIX + 31) / 32;
    static constexpr int GPC = 16;                                // (s, e) groups per chunk, 16 MFMAs each
    static constexpr int BM = 128, BN = 128;
    static_assert(NPASS + 1 <= GPC, "staging passes must fit in the chunk's groups");
    static_assert(64 % (4 * PF) == 0, "ring depth (4*PF fragments) must divide the 64 fragments per chunk");
};

// ------------------------------------------------------------------------------------------------------------
// 8-wave form: two waves per SIMD.  Wave (wn = N quarter, wh = half) owns V rows {2*wh, 2*wh + 1} of every tile, i.e. 8
// of the 16 transform-domain products: 128 accumulator registers, half of the input-transform adds, 12 of the 16 patch
// pixels, its own half of the weight fragments.  Both waves of a SIMD run the same program out of phase, so one's
// VALU / LDS / staging work issues under the other's MFMAs (the overlap a lone 512-register wave cannot have).
// A^T M A is linear in M: each half produces a partial 2x2 output per tile and the halves are summed once per tile
// through LDS (registers 0..7 of a lane are finished by half 0, 8..15 by half 1).
// bias + temb + residual enter through the accumulator init: with Y = A^T M A, setting M[0][0] = R00, M[0][3] = -R01,
// M[3][0] = -R10, M[3][3] = R11 adds exactly R to the 2x2 output (row 0 -> half 0, row 3 -> half 1), so the residual loads
// overlap the first halo fetch and cost no registers of their own.
// (The body is a device function templated on the half WH: everything that depends on which V rows a wave owns is
// resolved at compile time.  With a runtime `wh` hipcc if-converts the branches -- BOTH halves' transforms plus a
// v_cndmask per value -- and a two-armed lambda inside one function re-creates the accumulator-phi spills.)
template <class C, int WH>
__device__ __forceinline__ void conv_wino8_body(const ConvParams& p, float* lds) {
    constexpr int KC = C::KC, PS = C::PS, RS = C::RS, GPC = C::GPC, PF = C::PF;
    constexpr int NP8 = 4;                                   // staging passes of 64 pixel slots
    constexpr int wh = WH;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1;

    int bx = blockIdx.x;
    const int tx = bx % p.tiles_x;
    bx /= p.tiles_x;
    const int ty = bx % p.tiles_y;
    const int b = bx / p.tiles_y;
    const int oy0 = ty * C::TH, ox0 = tx * C::TW;
    const int iy0 = oy0 - 1, ix0 = ox0 - 1;
    const int Hv = p.Hin << p.ups, Wv = p.Win << p.ups;

    // ---- halo loader: 64 pixel slots x 8 channel quads.  Passes 0..2 = halo rows 2i, 2i+1 x columns 0..31, pass 3 =
    // the two leftover columns of all six rows (12 slots) ----
    static_assert(C::HH == 6 && C::HW == 34, "loader geometry is written for the 4 x 32 tile");
    const int q = tid & 7, pl = tid >> 3;                    // pl in 0..63
    const int prow = pl >> 5, pcol = pl & 31;
    const int ixa = ix0 + pcol;
    const bool colok = ixa >= 0 && ixa < Wv;
    const int colx = colok ? (ixa >> p.ups) : 0;
    const int iy3 = iy0 + (pl >> 1), ix3 = ix0 + 32 + (pl & 1);
    const bool ok3 = pl < 12 && iy3 >= 0 && iy3 < Hv && ix3 >= 0 && ix3 < Wv;
    const int soff3 = ok3 ? ((b * p.Hin + (iy3 >> p.ups)) * p.Win + (ix3 >> p.ups)) : 0;
    const int wbase = prow * RS + pcol * PS + q * 4;          // + 2*i*RS
    const int wbase3 = (pl >> 1) * RS + (32 + (pl & 1)) * PS + q * 4;
    auto row_src = [&](int i) -> int {                        // per-thread (2 rows per pass): source row base or -1
        const int iy = iy0 + 2 * i + prow;
        return (iy >= 0 && iy < Hv) ? (b * p.Hin + (iy >> p.ups)) * p.Win : -1;
    };
    f32x4 pre[NP8];
    f32x4 gsc, gsh;
    bool cvalid;
    auto issue_loads = [&](int chunk) {
        const int s = chunk >= p.nchunk0;
        const int cl = (s ? chunk - p.nchunk0 : chunk) * KC + q * 4;
        const int cs = p.csrc[s];
        cvalid = cl < cs;
        const float* __restrict__ base = p.src[s] + (cvalid ? cl : 0);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int rs = row_src(i);
            pre[i] = *reinterpret_cast<const f32x4*>(base + (size_t)((rs < 0 ? 0 : rs) + colx) * cs);
        }
        pre[3] = *reinterpret_cast<const f32x4*>(base + (size_t)soff3 * cs);
        if (p.gn) {
            const int cg = cvalid ? (s ? p.csrc[0] : 0) + cl : 0;
            gsc = *reinterpret_cast<const f32x4*>(p.gscale + (size_t)b * p.ctot + cg);
            gsh = *reinterpret_cast<const f32x4*>(p.gshift + (size_t)b * p.ctot + cg);
        }
    };
    auto write_pass = [&](float* buf, int i) {
        f32x4 v = pre[i];
        const bool ok = cvalid && (i < 3 ? (colok && row_src(i) >= 0) : ok3);
        if (p.gn) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], gsc[e], gsh[e]);
        }
        if (p.silu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = silu_f<false>(v[e]);
        }
        if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (i < 3) *reinterpret_cast<f32x4*>(&buf[wbase + 2 * i * RS]) = v;
        else if (pl < 12) *reinterpret_cast<f32x4*>(&buf[wbase3]) = v;
    };

    // ---- operand addressing: lane = (Winograd tile li, channel half lh); this wave reads patch rows wh..wh+2 ----
    const int li = lane & 31, lh = lane >> 5;
    const int wty = li >> 4, wtx = li & 15;
    const int a_base = (2 * wty + wh) * RS + (2 * wtx) * PS + lh * 4;
    const int ntile = blockIdx.y * 4 + wn;
    const bool nvalid = ntile * 32 < p.Cout;
    // packed weights [ntile][chunk][s][e][xiq][lane][4]: this wave uses xiq = 2*wh, 2*wh + 1
    const float* __restrict__ wp = p.w + ((size_t)(nvalid ? ntile : 0) * p.nchunks) * 16384 + wh * 512;
    const int n = ntile * 32 + li;
    const bool nok = nvalid && n < p.Cout;
    const bool full_tile = oy0 + C::TH <= p.Hout && ox0 + C::TW <= p.Wout;      // wave-uniform

    // acc[2*x + j]: x = local V row (global row 2*wh + x), j = column
    f32x16 acc[8];
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
    if (nok) {
        float add = p.bias ? p.bias[n] : 0.f;
        if (p.temb) add += p.temb[(size_t)b * p.temb_ld + n];
        // half 0 owns M[0][0] = R00 and M[0][3] = -R01; half 1 owns M[3][0] = -R10 and M[3][3] = R11
        float rv[32];
        if (p.residual && full_tile && !(C::OPT & 256)) {      // OPT 256 (ablation): no residual loads
            // tile wholly inside the image (the usual case): no clamps, and everything but one per-lane offset is
            // wave-uniform -- 32 scalar bases (SALU) + one VGPR offset instead of 64-bit VALU address chains per load
            const __amdgpu_buffer_rsrc_t rr = buf_rsrc(p.residual + (((size_t)b * p.Hout + oy0 + wh) * p.Wout + ox0) * p.Cout);
            const unsigned voff = ((unsigned)(8 * lh) * (unsigned)p.Cout + (unsigned)n) * 4u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned pix = (unsigned)(2 * (r >> 3)) * (unsigned)p.Wout + (unsigned)(2 * ((r & 3) + 8 * ((r >> 2) & 1)));
                rv[2 * r] = buf_load1(rr, voff, pix * (unsigned)p.Cout * 4u);
                rv[2 * r + 1] = buf_load1(rr, voff, (pix + 1u) * (unsigned)p.Cout * 4u);
            }
        } else if (p.residual && !(C::OPT & 256)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tile = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int y = min(oy0 + 2 * (tile >> 4) + wh, p.Hout - 1);
                const int x0 = min(ox0 + 2 * (tile & 15), p.Wout - 1), x1 = min(ox0 + 2 * (tile & 15) + 1, p.Wout - 1);
                rv[2 * r] = p.residual[(((size_t)b * p.Hout + y) * p.Wout + x0) * p.Cout + n];
                rv[2 * r + 1] = p.residual[(((size_t)b * p.Hout + y) * p.Wout + x1) * p.Cout + n];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 32; ++k) rv[k] = 0.f;
        }
        // (accumulator indices must be compile-time: a runtime-indexed register array is placed in scratch memory)
        if constexpr (wh == 0) {   // global row 0 = local row 0: (+R00, -R01)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[0][r] = rv[2 * r] + add;
                acc[3][r] = -(rv[2 * r + 1] + add);
            }
        } else {             // global row 3 = local row 1: (-R10, +R11)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[4][r] = -(rv[2 * r] + add);
                acc[7][r] = rv[2 * r + 1] + add;
            }
        }
    }

    // weight ring: this wave consumes 2 fragments per group g = (s, e): f = 2*g + k, k = local xiq
    constexpr int RF = 2 * PF;
    f32x4 ring[RF];
    auto foff = [](int f) { return (f >> 1) * 1024 + (f & 1) * 256; };     // float offset of fragment f inside a chunk
    const __amdgpu_buffer_rsrc_t wrs = buf_rsrc(wp);                       // fragment address = scalar offset + lane * 16 B
    const unsigned lane16 = lane * 16;
#pragma unroll
    for (int f = 0; f < RF; ++f) ring[f] = buf_load4(wrs, lane16, foff(f) * 4u);

    using f32x2 = __attribute__((ext_vector_type(2))) float;
    auto load_half = [&](const float* buf, int hh, f32x2 (&dst)[12]) {      // patch rows wh..wh+2, channels of half hh
        int ab = a_base;
        asm volatile("" : "+v"(ab));
        __builtin_assume((ab & 1) == 0);
        const int coff = (hh >> 1) * 8 + (hh & 1) * 2;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb)
                dst[a * 4 + bb] = *reinterpret_cast<const f32x2*>(&buf[ab + a * RS + bb * PS + coff]);
    };
    // the two V rows of this wave, for BOTH channels of a half at once (the channel pair a ds_read_b64 delivers is a
    // 64-bit register pair, so every add of B^T d B is one v_pk_add_f32 = two results per VALU issue):
    // rows (0,1) from patch rows (0,1,2); rows (2,3) from patch rows (1,2,3).  Same operation tree as the scalar form.
    // (OPT 32: the adds as inline-asm v_pk_add_f32, which hipcc's post-RA peephole cannot split back into two
    // scalar adds when they sit in the shadow of an MFMA.)
    auto padd = [](f32x2 a, f32x2 b) -> f32x2 {
        if constexpr (C::OPT & 32) {
            f32x2 r;
            asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
            return r;
        } else return a + b;
    };
    auto psub = [](f32x2 a, f32x2 b) -> f32x2 {
        if constexpr (C::OPT & 32) {
            f32x2 r;
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
            return r;
        } else return a - b;
    };
    auto transform2 = [&](const f32x2 (&d)[12], f32x2 (&v)[8]) {
        f32x2 r0[4], r1[4];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const f32x2 u0 = d[0 + bb], u1 = d[4 + bb], u2 = d[8 + bb];
            if constexpr (wh == 0) {   // t0 = u0, t1 = u1, t2 = u2:  r[0] = t0 - t2, r[1] = t1 + t2
                r0[bb] = psub(u0, u2);
                r1[bb] = padd(u1, u2);
            } else {                 // t1 = u0, t2 = u1, t3 = u2:  r[2] = t2 - t1, r[3] = t1 - t3
                r0[bb] = psub(u1, u0);
                r1[bb] = psub(u0, u2);
            }
        }
        v[0] = psub(r0[0], r0[2]); v[1] = padd(r0[1], r0[2]); v[2] = psub(r0[2], r0[1]); v[3] = psub(r0[1], r0[3]);
        v[4] = psub(r1[0], r1[2]); v[5] = padd(r1[1], r1[2]); v[6] = psub(r1[2], r1[1]); v[7] = psub(r1[1], r1[3]);
    };

    // One chunk = 8 halves (channel pairs per lane) x 2 groups of 8 MFMAs.  Even group of half h: issue the LDS patch
    // reads of half h+1 (single patch buffer: it was consumed one group earlier); odd group: transform them into the
    // other operand set.  The second wave of the SIMD covers the LDS latency; a group is 512 MFMA cycles.
    auto chunk_body = [&](const int chunk, const bool more) __attribute__((always_inline)) {
        const float* cur = lds + (chunk & 1) * C::BUF_FLOATS;
        float* nxt = lds + ((chunk + 1) & 1) * C::BUF_FLOATS;
        const unsigned wcb = (unsigned)chunk * 65536u;      // byte offset of this chunk's fragments
        f32x2 dh[12];
        f32x2 vv[2][8];
        auto make_operands = [&](f32x2 (&v)[8]) {
            if constexpr (C::OPT & 8) {          // OPT 8 (ablation): no transform adds
#pragma unroll
                for (int x = 0; x < 8; ++x) v[x] = dh[x];
            } else transform2(dh, v);
        };
        load_half(cur, 0, dh);
        make_operands(vv[0]);
#pragma unroll
        for (int g = 0; g < GPC; ++g) {
            const int hh = g >> 1;
            if (hh + 1 < 8) {
                if ((g & 1) == 0) {
                    if (!(C::OPT & 2) || hh == 0) load_half(cur, hh + 1, dh);      // OPT 2 (ablation): one more half-load per chunk only
                } else make_operands(vv[(hh + 1) & 1]);
            }
            if (more && !(C::OPT & 4)) {      // OPT 4: no staging
                constexpr int G0 = GPC - NP8 - 1;
                if (g >= G0 && g < G0 + NP8) write_pass(nxt, g - G0);
                if (g == G0 + NP8 && chunk + 2 < p.nchunks) issue_loads(chunk + 2);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int f = g * 2 + k;
                const f32x4 bq = ring[f % RF];
                if constexpr (!(C::OPT & 1))     // OPT 1: no weight refills
                    ring[f % RF] = buf_load4(wrs, lane16, wcb + (f + RF < 32 ? foff(f + RF) : 16384 + foff(f + RF - 32)) * 4u);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[k * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[hh & 1][k * 4 + j][g & 1], bq[j], acc[k * 4 + j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- pipeline ----
    issue_loads(0);
#pragma unroll
    for (int i = 0; i < NP8; ++i) write_pass(lds, i);
    if (p.nchunks > 1) issue_loads(1);
    __syncthreads();

    if (!nvalid) {
        for (int chunk = 0; chunk < p.nchunks; ++chunk) {
            if (chunk + 1 < p.nchunks) {
#pragma unroll
                for (int i = 0; i < NP8; ++i) write_pass(lds + ((chunk + 1) & 1) * C::BUF_FLOATS, i);
                if (chunk + 2 < p.nchunks) issue_loads(chunk + 2);
            }
            __syncthreads();
        }
        if constexpr (!(C::OPT & 16)) __syncthreads();     // matches the exchange barrier below
        return;
    }
    for (int chunk = 0; chunk + 1 < p.nchunks; ++chunk) {
        chunk_body(chunk, true);
        __syncthreads();
    }
    chunk_body(p.nchunks - 1, false);
    if constexpr (C::OPT & 16) {   // OPT 16 (timing ablation): no output transform / exchange / stores
        float keep = 0.f;
#pragma unroll
        for (int x = 0; x < 8; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) keep += acc[x][r];
        if (keep == 123.456f) p.out[0] = keep;
        return;
    }
    __syncthreads();         // every wave is done with the halo buffers: they become the exchange image

    // ---- partial output transform: this half's rows of M -> partial Y (linear), then swap halves through LDS ----
    // tmp[0][j] = M0j + M1j + M2j, tmp[1][j] = M1j - M2j - M3j.  half 0 (rows 0,1): (M0j + M1j, M1j); half 1 (rows 2,3): (M2j, -M2j - M3j)
    // exchange image: xch[wave][lane][36]; registers 0..7 are finished by half 0, 8..15 by half 1.
    // (36 floats per lane: 16-byte accesses at a 144-byte lane stride are bank-conflict free, as in the halo image)
    constexpr int XS = 36;
    float* xch = lds;
    float mine[32];      // the 8 registers this wave finishes: partial y[4] each
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float t[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float m0 = acc[0 + j][r], m1 = acc[4 + j][r];          // local rows 0, 1
            if constexpr (wh == 0) {
                t[0][j] = m0 + m1;
                t[1][j] = m1;
            } else {
                t[0][j] = m0;
                t[1][j] = -m0 - m1;
            }
        }
        float y[4];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            y[a * 2 + 0] = t[a][0] + t[a][1] + t[a][2];
            y[a * 2 + 1] = t[a][1] - t[a][2] - t[a][3];
        }
        const bool keep = (r >> 3) == wh;       // compile-time: registers 0..7 are finished by half 0, 8..15 by half 1
        const int rr = r & 7;
        if (keep) {
#pragma unroll
            for (int k = 0; k < 4; ++k) mine[rr * 4 + k] = y[k];
        } else {                                 // the partner finishes this register: one 16-byte write per register
            *reinterpret_cast<f32x4*>(&xch[(wave * 64 + lane) * XS + rr * 4]) = f32x4{y[0], y[1], y[2], y[3]};
        }
    }
    __syncthreads();
    {
        const int partner = wave ^ 1;
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(&xch[(partner * 64 + lane) * XS + rr * 4]);
#pragma unroll
            for (int k = 0; k < 4; ++k) mine[rr * 4 + k] += o[k];
        }
    }

    // ---- GroupNorm sums of the output: taken HERE, where a lane still owns one channel (n) at 32 pixels -- one double
    // pair per lane and one lane-half exchange instead of four pairs and three exchanges after the transposes ----
    if (p.stats) {
        double s1 = 0.0, s2 = 0.0;
        if (full_tile) {
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const double dv = (double)mine[k];
                s1 += dv;
                s2 = fma(dv, dv, s2);
            }
        } else {
            // register r = 8*wh + rr -> tile (rr & 3) + 8*(rr >> 2) + 4*lh of tile row wh; value k -> pixel (2*wh + (k >> 1), 2*tilecol + (k & 1))
#pragma unroll
            for (int rr = 0; rr < 8; ++rr)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool inside = oy0 + 2 * wh + (k >> 1) < p.Hout && ox0 + 2 * ((rr & 3) + 8 * (rr >> 2) + 4 * lh) + (k & 1) < p.Wout;
                    const double dv = inside ? (double)mine[rr * 4 + k] : 0.0;      // branch-free: a zero adds nothing to either sum
                    s1 += dv;
                    s2 = fma(dv, dv, s2);
                }
        }
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (lh == 0 && nok) {
            // slot = (tile, half): two slots per spatial tile (cdx_conv_stats_slots accounts for it)
            const int slot = (ty * p.tiles_x + tx) * 2 + wh;
            const int nslots = p.tiles_y * p.tiles_x * 2;
            double* o = p.stats + (((size_t)b * nslots + slot) * p.Cout + n) * 2;
            o[0] = s1;
            o[1] = s2;
        }
    }

    // ---- packed stores (quad transposes) of this wave's 8 tile-registers x 4 positions ----
    // (buf_store4 carries its own wait states: variants of this kernel were flaky until the store-data hazard of
    // buffer stores with an SGPR soffset was found -- DESIGN.md section 8, tools/flake_probe.py, tools/microbench/store_hazard.hip)
    int eoy0 = oy0, eox0 = ox0, elh = lh;
    asm volatile("" : "+s"(eoy0), "+s"(eox0), "+v"(elh));
    const int q4 = li & 3;
    const int cq = ntile * 32 + (li & ~3);
    const bool quad_ok = cq < p.Cout;
    const bool vec_ok = (p.out_ld & 3) == 0 && cq + 4 <= p.out_ld;
    const bool fast_store = full_tile && vec_ok && ntile * 32 + 32 <= p.Cout;      // wave-uniform
    if (fast_store) {
        // scalar base per (kk, pos) + one per-lane offset: tile = 8*(2*wh + kk) + q4 + 4*lh -> row wh, column 8*kk + q4 + 4*lh
        const __amdgpu_buffer_rsrc_t ro = buf_rsrc(p.out + (((size_t)b * p.Hout + eoy0 + 2 * wh) * p.Wout + eox0) * p.out_ld);
        const unsigned voff = ((unsigned)(2 * (q4 + 4 * elh)) * (unsigned)p.out_ld + (unsigned)cq) * 4u;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)            // blocks of 4 registers: global registers 8*wh + 4*kk + i
#pragma unroll
            for (int pos = 0; pos < 4; ++pos) {
                float x[4] = {mine[(4 * kk + 0) * 4 + pos], mine[(4 * kk + 1) * 4 + pos], mine[(4 * kk + 2) * 4 + pos], mine[(4 * kk + 3) * 4 + pos]};
                quad_transpose(x, q4);
                const unsigned pix = (unsigned)(pos >> 1) * (unsigned)p.Wout + (unsigned)(16 * kk + (pos & 1));
                buf_store4(ro, voff, pix * (unsigned)p.out_ld * 4u, f32x4{x[0], x[1], x[2], x[3]});
            }
    } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int tile = 8 * (2 * wh + kk) + q4 + 4 * elh;
            const int oy = eoy0 + 2 * (tile >> 4), ox = eox0 + 2 * (tile & 15);
#pragma unroll
            for (int pos = 0; pos < 4; ++pos) {
                float x[4] = {mine[(4 * kk + 0) * 4 + pos], mine[(4 * kk + 1) * 4 + pos], mine[(4 * kk + 2) * 4 + pos], mine[(4 * kk + 3) * 4 + pos]};
                quad_transpose(x, q4);
                const int py = oy + (pos >> 1), px = ox + (pos & 1);
                if (quad_ok && py < p.Hout && px < p.Wout) {
                    const size_t pix = ((size_t)b * p.Hout + py) * p.Wout + px;
                    if (vec_ok) *reinterpret_cast<f32x4*>(p.out + pix * p.out_ld + cq) = f32x4{x[0], x[1], x[2], x[3]};
                    else
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (cq + c < p.Cout) p.out[pix * p.out_ld + cq + c] = x[c];
                }
            }
        }
    }
}

template <class C>
__global__ __launch_bounds__(512, 2) void conv_wino8_kernel(const ConvParams p) {
    constexpr int LDS_FLOATS = 2 * C::BUF_FLOATS > 8 * 64 * 36 ? 2 * C::BUF_FLOATS : 8 * 64 * 36;   // halo double buffer / exchange image
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    // wave-uniform: both arms execute the same number of barriers
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1) conv_wino8_body<C, 1>(p, lds);
    else conv_wino8_body<C, 0>(p, lds);
}

template <class C>
inline int conv_wino8_launch(const ConvParams& p, hipStream_t stream) {
    dim3 grid(p.tiles_x * p.tiles_y * p.B, ceil_div(p.Cout, C::BN));
    hipLaunchKernelGGL(conv_wino8_kernel<C>, grid, dim3(512), 0, stream, p);
    return check_launch();
}

int conv_dispatch_wino(int variant, const ConvParams& p, hipStream_t stream);

}  // namespace cdx
