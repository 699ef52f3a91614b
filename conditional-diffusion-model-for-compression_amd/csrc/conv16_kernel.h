// 16-bit-operand convolution: implicit GEMM on v_mfma_f32_32x32x16_f16 / _bf16 (fp32 accumulate).  Three uses of ONE kernel:
//   * fp16 storage  (BASELINE.json configs[4] "fp16 UNet on CDNA4 MFMA"; SURVEY.md section 8(a) U3/U4 "_f16 variants");
//   * bf16 storage  (Conv16Cfg<..., BF = 1>);
//   * FLOAT32 convolution with hi | lo split operands (SPLIT = 1, below): the dominant kernel of the float32 path.
// No reference file exists to cite (the reference snapshot is empty); semantics = F.conv2d with the fusions of cdx.h.
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
// In the 16-bit storage modes the kernel is not MFMA-bound (36.9k -> 2.3k MFMA cycles per chunk): the staging VALU, LDS
// reads and the L2 weight stream set its speed.  Staging is cut into units interleaved with the MFMAs (below), halo loads
// use buffer addressing, GroupNorm / SiLU modes are compile-time (STG), and the last channel block of a 128 + 64-channel
// layer runs a 2 x 2 wave layout (WM).
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

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
__device__ __forceinline__ f32x16 mfma_32x32x16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

// Two float32 values -> packed 16-bit pair(s).  SPLIT: hi = RN(v) as fp16 pair (v_cvt_pk_f16_f32), lo = RN(v - hi) computed by the
// mixed-precision FMA (-1.0 * hi[f16] + v[f32], rounded once to fp16: v_fma_mixlo/mixhi_f16) -- 3 instructions per pair
// where "convert, convert back, subtract, convert" took 7: beside the fp16 MFMA the staging VALU work is the co-limiter
// (SQ counters: ~3.7 vector instructions per MFMA and wave before this change).
__device__ __forceinline__ unsigned pack_hi_f16(float v0, float v1) {
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(v0), "v"(v1));
    return r;
}
__device__ __forceinline__ unsigned pack_lo_f16(unsigned hi2, float v0, float v1) {
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hi2), "v"(v0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(hi2), "v"(v1));
    return r;
}

struct Conv16Params {
    const void* src[2];
    int csrc[2];
    int src_f32;     // sources are float32 (else float16)
    int nchunk0, nchunks, ctot;
    int B, Hin, Win, Hout, Wout, Cout;
    int ups, gn, silu;
    int abl;         // timing ablation selector (diagnostics)
    float wunscale;  // SPLIT: 2^-s, undoes the host's power-of-two weight scaling (exact); 1 otherwise
    const void* w;           // fragment-packed weights: fp16 (bf16 in BF mode; fp16 hi | lo planes in SPLIT mode)
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
    int stats_wm;            // GroupNorm-sum slots per spatial tile (2 when the layer's last channel block runs the 2 x 2 layout)
    int tiles_x, tiles_y;
    // SPLIT range contract (cdx.h): the activation exponent comes from act_exp (GroupNorm-ed input: gscale / gshift arrive
    // pre-multiplied by 2^act_exp) or, per image, from the producers' amax words of the un-normalised sources
    // PHASE launches of a 3x3 convolution after nearest-2x upsampling (conv_split.hip: four 2x2 convolutions on the LOW-resolution
    // source, one per output phase (dy, dx); KS = 2): the tile walks low-resolution pixels (Hout x Wout), output pixel (y, x) of the
    // launch is stored at (ostep y + ody, ostep x + odx) of the full-resolution tensor, the halo starts pady / padx pixels up / left
    // of the tile, and the launch's GroupNorm-sum slots start at slot_base of nslots_total.  Plain launches: 1, 0, 0, PAD, PAD, 0, 0.
    int ostep, ody, odx, pady, padx, slot_base, nslots_total;
    int act_exp;
    const unsigned* amax[2]; // [B][CDX_AMAX_WORDS] float32 bit patterns (max over the words = max |x| of the image) per source, or null
    unsigned* amax_out;      // [B][CDX_AMAX_WORDS] or null: atomic max of one word with the bit pattern of the wave's max |out|
};

// Activation scaling of one SPLIT workgroup (all wave-uniform: scalar registers).  Staged activations are x 2^ea with
// max |x| 2^ea in [2^14, 2^15) (the weights sit in [2^13, 2^14) by the packer's 2^sw): hi = fp16(v), lo = fp16(v - hi) then
// keep 22 significant bits for every element down to 2^-17 of the tensor's maximum and an ABSOLUTE error of 2^-39 of the
// maximum below that -- float32-level error relative to the output scale at any input scale.  The accumulators hold
// out 2^(sw + ea) =: out 2^S.
struct ActScale {
    float a;        // 2^ea
    float ksilu;    // -log2(e) 2^-ea: SiLU of a pre-scaled value v2 = v 2^ea is v2 / (1 + exp2(v2 ksilu))
    float un;       // 2^-S: accumulators -> outputs
    float inv;      // 2^S: bias / temb / residual -> accumulator init
    bool late;      // S > 64: inv times an O(1) additive term could overflow float32 -- bias / temb / residual are then added in
                    // the epilogue (the accumulators start at 0); |S| is capped at 120 by giving up activation headroom
};
// max over the words of image b (b wave-uniform: one s_load_dwordx16 and scalar compares)
using u32x16 = __attribute__((ext_vector_type(16))) unsigned;
static_assert(CDX_AMAX_WORDS == 16, "amax_words reads one 64-byte row");
__device__ __forceinline__ unsigned amax_words(const unsigned* base, int b) {
    const u32x16 v = *reinterpret_cast<const u32x16*>(base + (size_t)b * CDX_AMAX_WORDS);
    unsigned m = v[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) m = v[k] > m ? v[k] : m;
    return m;
}
// one wave's contribution to amax_out: word chosen by `salt` (wave-global index), fire-and-forget.  (A read of the word first,
// to skip atomics that cannot raise it, made every wave END with a dependent load: +25 % on the short stride-2 launches in the
// forward, where the words start at zero; the atomics themselves overlap with the rest of the launch.)
__device__ __forceinline__ void amax_publish(unsigned* amax_out, int b, unsigned salt, float am) {
    if (am != 0.f) atomicMax(amax_out + (size_t)b * CDX_AMAX_WORDS + (salt & (CDX_AMAX_WORDS - 1)), __float_as_uint(am));
}
template <bool SPLIT, bool GN>
__device__ __forceinline__ ActScale act_scale_of(const Conv16Params& p, int b) {
    ActScale s{1.f, -1.44269504088896341f, 1.f, 1.f, false};
    if constexpr (SPLIT) {
        int ea = 0;
        if constexpr (GN) ea = p.act_exp;
        else if (p.amax[0]) {
            unsigned m = amax_words(p.amax[0], b);
            if (p.amax[1]) { const unsigned m1 = amax_words(p.amax[1], b); m = m1 > m ? m1 : m; }
            const int E = (int)(m >> 23);
            ea = (m == 0u || E >= 255) ? 0 : 141 - E;      // +-Inf / NaN bound, or an all-zero image: no scaling
        }
        ea = ea > 100 ? 100 : ea < -100 ? -100 : ea;
        const int sw = 127 - (int)(__float_as_uint(p.wunscale) >> 23);
        int S = sw + ea;
        if (S > 120) { ea -= S - 120; S = 120; }
        if (S < -120) { ea += -120 - S; S = -120; }
        s.a = __uint_as_float((unsigned)(ea + 127) << 23);
        s.ksilu = -1.44269504088896341f * __uint_as_float((unsigned)(127 - ea) << 23);
        s.un = __uint_as_float((unsigned)(127 - S) << 23);
        s.inv = __uint_as_float((unsigned)(127 + S) << 23);
        s.late = S > 64;
    }
    return s;
}

// channels left for the last 128-channel block; the 2 x 2 wave layout serves it when they fit two N-tiles
__host__ __device__ inline bool conv16_tail_2x2(int cout, int mt) { const int rem = cout % 128; return mt == 4 && rem > 0 && rem <= 64; }

// ABL: timing-only ablations (wrong results; libcdx_tune.so only): 1 = no epilogue, 2 = stage only the first chunk,
// 4 = no weight refills, 8 = no LDS operand reads (registers reused), 16 = no residual loads, 32 = no GroupNorm sums,
// 64 = halo global loads only for chunks 0 and 1 (later chunks restage stale registers: VALU + LDS-write cost stays),
// 128 = staging units not interleaved with the MFMA quarters, 256 = units computed but not stored to LDS,
// 512 = per-wave phase stamps instead of GroupNorm sums (use with 32)
// DB = 0: ONE halo image and two barriers per chunk (stride-2 SPLIT tiles, whose 5 x 65-pixel hi|lo image would
// otherwise leave room for a single workgroup per CU).
// BF = 1: bfloat16 storage and v_mfma_f32_32x32x16_bf16 (dtype "bf16"); never together with SPLIT (fp16 hi | lo).
// WS = 1: WAVE-SPECIALISED workgroup of 8 waves (512 threads): waves 0-3 are the MFMA waves of the 4-wave layout (LDS operand
// reads, weight ring, MFMAs, accumulator init, epilogue -- no staging arithmetic at all), waves 4-7 are PRODUCERS (halo loads,
// GroupNorm / SiLU / split, LDS stores of the next chunk -- no MFMA); hand-off through the same two halo images and the same
// one barrier per chunk.  The matrix pipe of a SIMD then never waits behind its own wave's vector work (round 2 measured
// SQ_VALU_MFMA_COEXEC = 0.235 of the busy cycles with homogeneous waves).  Two such workgroups per CU = 4 waves per SIMD:
// 128 VGPRs per wave, which the MFMA waves meet because the staging registers are gone.
template <int KS_, int STRIDE_, int LOGTW_, int MT_, int PF_ = 3, int ABL_ = 0, int SPLIT_ = 0, int DB_ = 1, int BF_ = 0, int WS_ = 0>
struct Conv16Cfg {
    static constexpr int KS = KS_, STRIDE = STRIDE_, LOGTW = LOGTW_, MT = MT_, PF = PF_, ABL = ABL_, SPLIT = SPLIT_, DB = DB_, BF = BF_, WS = WS_;
    static_assert(!(SPLIT && BF), "the split operands are fp16");
    static_assert(!WS || DB, "the producer waves fill the OTHER halo image");
    using H = std::conditional_t<BF != 0, __bf16, _Float16>;
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

// SiLU of a value that carries a power-of-two scale: v2 = v 2^e, ksilu = -log2(e) 2^-e  ->  silu(v) 2^e
__device__ __forceinline__ float silu16_f(float v2, float ksilu) {
    return v2 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v2 * ksilu));
}

// STG (compile-time staging mode, chosen at launch from the GN / SiLU flags -- a runtime flag costs a v_cndmask per element
// and flag in the staging code): 0 = plain, 1 = GroupNorm scale/shift, 2 = GroupNorm + SiLU, 3 = SiLU only.
// WM (wave layout of the 128-pixel x N tile): 1 = 1 x 4 (every wave owns all MT M-tiles of one 32-channel N-tile: 128
// channels per workgroup); 2 = 2 x 2 (wave = half of the M-tiles x one of TWO N-tiles: 64 channels per workgroup) -- used for
// the LAST channel block of a layer whose channel count leaves at most 64 channels there (cout = 192: 128 + 64), where the
// 1 x 4 layout would leave two of the four waves without output channels.
// XCD-contiguous workgroup order -- an EXPERIMENT (ABL 8192, tuning build only), measured and not shipped.  The dispatcher deals
// consecutive workgroup ids (x fastest, then y) round-robin over the chip's 8 XCDs, each with its own L2; remapped, XCD x works
// through ONE contiguous range of (tile, channel block) pairs, so that neighbouring tiles (41 % of an 8 x 16 tile's halo is shared)
// and the channel blocks of a tile (the same input) sit behind one L2.  Round 4, in-process A/B (tools/session/gpu_r4c.sh):
// 256^2 128->128 +0.6 %, 256->128 0, 128^2 256->256 +0.6 %, 256^2 384->384 (3 blocks) +2.3 %, 512^2 192->192 (128 + 64-channel tail
// block) -5.5 %; FETCH_SIZE per launch 631 vs 622 MiB, 2.81 vs 2.82 GiB, 1.93 vs 2.07 GiB -- the L2s do not turn the shared bytes into
// hits in either order (the extra fetch equals the whole halo overlap: 64 resident workgroups stream 14 MB through a 4 MB L2).
__device__ __forceinline__ int xcd_contiguous(int id, int n) {
    const int q = n >> 3, r = n & 7, x = id & 7, k = id >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}
template <class C>
__device__ __forceinline__ void conv16_block(int& bx, int& by) {
    bx = blockIdx.x;
    by = blockIdx.y;
    if constexpr ((C::ABL & 8192) != 0) {
        const int gx = gridDim.x, gy = gridDim.y;
        const int s = xcd_contiguous(by * gx + bx, gx * gy);
        if (gy == 1) { bx = s; }
        else { bx = s / gy; by = s - bx * gy; }
    }
}

template <class C, int STG, int WM>
__device__ __forceinline__ void conv16_body(const Conv16Params& p, typename C::H* lds_all, int bx, const int by) {
    constexpr bool kGN = STG == 1 || STG == 2, kSILU = STG == 2 || STG == 3;
    using H = typename C::H;                                       // 16-bit storage / operand type: _Float16 or __bf16
    using h8 = __attribute__((ext_vector_type(8))) H;
    using h4 = __attribute__((ext_vector_type(4))) H;
    constexpr int KC = C::KC, PSH = C::PSH, RSH = C::RSH, TAPS = C::TAPS, NPASS = C::NPASS, GPC = C::GPC;
    constexpr int WN = 4 / WM;
    constexpr int MT = C::MT / WM;                                 // M-tiles PER WAVE
    static_assert(C::MT % WM == 0 && (MT * 32) % C::TW == 0, "a wave's M-tiles must be whole tile rows");
    constexpr int PF = C::PF < GPC ? C::PF : GPC;
    // software-pipelined operand reads in the MFMA waves (below); ABL 4096 (tuning build): off, for the in-process A/B
    constexpr bool kPipe = C::WS && GPC > 2 && !(C::ABL & 4096);

    const int tid = C::WS ? (int)(threadIdx.x & 255u) : (int)threadIdx.x;      // WS: index inside the role's 4 waves
    const bool producer = C::WS && __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) != 0;      // wave-uniform
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wv % WN, wm = wv / WN;
    // A new workgroup's producer waves run their ~200-instruction prologue (tile arithmetic, buffer resources) at RAISED priority until the
    // first halo loads are out: stamped, that prologue took 3.6 us (16-bit tile) / 5.8 us (float32) of the 6 / 10 us the MFMA waves wait for
    // their first chunk -- its instructions queue behind the resident workgroup's MFMA waves on every SIMD (tools/ws_stamps.py,
    // profiles/r04_n_*).  ABL 65536 (tuning build): off.
    constexpr bool kPrio = C::WS && !(C::ABL & 65536);
    if constexpr (kPrio) {
        if (producer) __builtin_amdgcn_s_setprio(3);
    }

    const int bx0 = bx;
    const int tx = bx % p.tiles_x;
    bx /= p.tiles_x;
    const int ty = bx % p.tiles_y;
    const int b = bx / p.tiles_y;
    const int oy0 = ty * C::TH, ox0 = tx * C::TW;
    const int oy0w = oy0 + ((wm * MT * 32) >> C::LOGTW);           // first output row of THIS wave's M-tiles
    const int iy0 = oy0 * C::STRIDE - (C::KS == 2 ? p.pady : C::PAD), ix0 = ox0 * C::STRIDE - (C::KS == 2 ? p.padx : C::PAD);
    // ABL & 512 (diagnostic build): s_memtime stamps per wave at the phase boundaries, 16 per wave, into the buffer behind
    // p.stats (which then holds no sums): 0 entry, 1 first loads issued, 2 first chunk staged (barrier passed), 3 + c chunk c
    // done (c < 8), 12 stores issued, 13 HW_ID, 14 XCC_ID  (digest: tools/conv_bench.py --stamps)
    auto stamp = [&](int k) __attribute__((always_inline)) {
        if constexpr ((C::ABL & 512) != 0) {
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (lane == 0)
                reinterpret_cast<unsigned long long*>(p.stats)[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wv) * 16 + k] = t;
            // the constant 100 MHz counter beside the shader-clock one, at entry (slot 11) and at the last stamp (slot 10): the clock
            // the chip held over this wave's life = delta s_memtime / delta s_memrealtime x 100 MHz (MI355X_MICROARCH.md, DVFS give-back 6)
            if (k == 0 || k == 12) {
                unsigned long long rt;
                asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt) : : "memory");
                if (lane == 0)
                    reinterpret_cast<unsigned long long*>(p.stats)[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wv) * 16 + (k == 0 ? 11 : 10)] = rt;
            }
        }
    };
    // (WS, ABL & 512) barrier accounting: ticks a wave spends inside the per-chunk __syncthreads (slot 15 of an MFMA wave; the
    // producers write their own rows behind the MFMA waves': 0 entry, 1 ticks at barriers, 2 ticks staging, 3 exit) -- tools/ws_stamps.py
    auto memtime = [&]() __attribute__((always_inline)) {
        unsigned long long t = 0;
        if constexpr ((C::ABL & 512) != 0) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        return t;
    };
    unsigned long long bar_ticks = 0, stage_ticks = 0;
    auto timed_barrier = [&]() __attribute__((always_inline)) {
        if constexpr (C::WS && (C::ABL & 512) != 0) {
            const unsigned long long t0 = memtime();
            __syncthreads();
            bar_ticks += memtime() - t0;
        } else {
            __syncthreads();
        }
    };
    if (!producer) stamp(0);
    if constexpr ((C::ABL & 512) != 0) {
        if (producer) {
            if (lane == 0) reinterpret_cast<unsigned long long*>(p.stats)[((size_t)(gridDim.y * gridDim.x + blockIdx.y * gridDim.x + blockIdx.x) * 4 + wv) * 16 + 0] = memtime();
        }
    }
    if constexpr ((C::ABL & 512) != 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (lane == 0 && !producer) {
            reinterpret_cast<unsigned long long*>(p.stats)[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wv) * 16 + 13] = hw;
            reinterpret_cast<unsigned long long*>(p.stats)[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wv) * 16 + 14] = xcc;
        }
    }
    const int Hv = p.Hin << p.ups, Wv = p.Win << p.ups;
    const ActScale asc = act_scale_of<C::SPLIT != 0, kGN>(p, b);

    // ---- loader: thread -> (pixel slot pl of 64, channel octet q of 4) ----
    const int q = tid & 3, pl = tid >> 2;
    int soff[NPASS];                                  // pixel index inside image b
    unsigned vmask = 0;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int hp = i * 64 + pl;
        const int hy = hp / C::HW, hx = hp - hy * C::HW;
        const int iy = iy0 + hy, ix = ix0 + hx;
        const bool ok = hp < C::NPIX && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv;
        soff[i] = ok ? (iy >> p.ups) * p.Win + (ix >> p.ups) : 0;
        vmask |= ok ? (1u << i) : 0u;
    }
    float pre[NPASS][8];
    f32x4 gsc[2], gsh[2];
    bool cvalid;

    // Halo loads use buffer addressing: per-lane byte offset voff[i] = (pixel * channels + 8 q) * elem (one 32-bit VGPR per
    // pass, recomputed only when the chunk sequence moves from src0 to src1), scalar offset = chunk * 32 channels * elem.
    // hipcc's flat form spent ~25 VALU instructions of 64-bit address arithmetic per load (a 200-instruction clump per chunk).
    // The resource covers IMAGE b of the source only (the host checks that one image is below 2 GiB, so 32-bit offsets cannot
    // wrap whatever the batch).  Padding pixels, slots past the halo's last pixel and channel octets past the source's end
    // read pixel 0 of the image -- an IN-RANGE dummy: out-of-range lanes are not free on this chip (with offsets beyond the
    // resource for them the fp16 3x3 kernel measured 8 % slower, in-process A/B) -- and are zeroed by one select per value
    // after the activation, so neither GroupNorm's shift nor NaN / Inf in the dummy can leak into the padding.
    unsigned voff[NPASS];
    __amdgpu_buffer_rsrc_t srs;
    int cur_src = -1;
    const unsigned esz = (C::SPLIT || p.src_f32) ? 4u : 2u;
    auto bind_source = [&](int sidx) {
        const unsigned cs = (unsigned)p.csrc[sidx];
        const size_t img = (size_t)p.Hin * p.Win * cs * esz;
        srs = buf_rsrc(static_cast<const char*>(p.src[sidx]) + (size_t)b * img, (unsigned)img);
#pragma unroll
        for (int i = 0; i < NPASS; ++i) voff[i] = ((unsigned)soff[i] * cs + 8u * (unsigned)q) * esz;      // (soff = 0 for invalid slots)
        cur_src = sidx;
    };
    auto issue_loads = [&](int chunk) {
        const int s = chunk >= p.nchunk0;
        if (s != cur_src) bind_source(s);            // wave-uniform, at most twice per tile
        const int cc = s ? chunk - p.nchunk0 : chunk;
        const int cl = cc * KC + q * 8;
        const int cs = p.csrc[s];
        cvalid = cl < cs;                       // channel counts are multiples of 8 (fp16) / 4 (fp32 conv_in: see host)
        const unsigned so = (unsigned)cc * (unsigned)KC * esz;
        if (C::SPLIT || p.src_f32) {
            // fp32 sources of the fp16-storage path may end on a 4-channel boundary (SPLIT: multiples of 8, see conv_split_ok)
            const bool hi = C::SPLIT || cl + 8 <= cs;
#pragma unroll
            for (int i = 0; i < NPASS; ++i) {
                const f32x4 v0 = buf_load4(srs, voff[i], so);
                const f32x4 v1 = buf_load4(srs, voff[i] + 16u, so);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pre[i][e] = v0[e];
                    pre[i][4 + e] = hi ? v1[e] : 0.f;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NPASS; ++i) {
                const h8 v = __builtin_bit_cast(h8, buf_load4(srs, voff[i], so));
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

    // Staging is cut into UNITS of half a pass (one pixel slot x 4 channels per thread: ~45 VALU instructions), so that a
    // unit rides on ONE 12-MFMA group without exceeding the issue slots the fp16 MFMA leaves free (a whole pass between two
    // MFMAs was a 128-instruction clump behind an exec-mask branch).  unit u = (pass u / UPP, channel group u % UPP).
    // UPP units per pass: 2 (4 channels each) in SPLIT mode, whose groups hold 12 MFMAs; 4 (2 channels each) in the 16-bit
    // storage modes, whose groups hold only MT = 4 MFMAs (a 4-channel unit there is ~10 VALU per MFMA: over the budget).
    constexpr int UPP = (!C::SPLIT && 4 * NPASS + 1 <= GPC) ? 4 : 2, EPU = 8 / UPP;
    static_assert(EPU % 2 == 0, "units are built from channel pairs");
    unsigned uo[EPU / 2], uol[EPU / 2];               // the unit being computed: packed 16-bit pairs (hi | lo)
    auto unit_pair = [&](int u, int kp) {             // channels 2 kp, 2 kp + 1 of unit u
        const int i = u / UPP, e = EPU * (u % UPP) + 2 * kp;
        // padding pixels / channels past the source hold a dummy read: they must be zero AFTER the activation -- one select per value.
        // No clamp: NaN / Inf and values beyond the fp16 range (impossible for finite inputs under the range contract) propagate
        // as non-finite outputs, as F.conv2d's do.
        const bool ok = cvalid && ((vmask >> i) & 1u);
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
            uo[kp] = __builtin_bit_cast(unsigned, b2{(__bf16)v[0], (__bf16)v[1]});
        } else {
            uo[kp] = pack_hi_f16(v[0], v[1]);
            if constexpr (C::SPLIT) uol[kp] = pack_lo_f16(uo[kp], v[0], v[1]);
        }
    };
    auto unit_store = [&](H* lds, int u) {
        const int i = u / UPP, h = u % UPP;
        const int hp = i * 64 + pl;
        const int hy = hp / C::HW, hx = hp - hy * C::HW;
        // slots past the halo's last pixel (last pass only) dump into the 16 unused pad bytes of the thread's pass-0 pixel:
        // an unconditional store keeps the unit free of exec-mask branches (which pin its VALU work in one clump)
        const bool real = (i + 1) * 64 <= C::NPIX || hp < C::NPIX;
        const int off = real ? hy * RSH + hx * PSH + q * 8 + EPU * h : (pl / C::HW) * RSH + (pl % C::HW) * PSH + KC * C::PLANES;
        const int off_lo = real ? off + KC : off + 4;
        using uU = __attribute__((ext_vector_type(EPU / 2))) unsigned;
        uU w, wl;
#pragma unroll
        for (int k = 0; k < EPU / 2; ++k) {
            if constexpr (EPU == 2) { w = uo[0]; wl = uol[0]; }
            else { w[k] = uo[k]; wl[k] = uol[k]; }
        }
        if constexpr (C::ABL & 256) {      // ablation: units computed but not stored (keeps the values alive)
            if (w[0] == 0x12345678u) *reinterpret_cast<uU*>(&lds[off]) = w;
            if constexpr (C::SPLIT) if (wl[0] == 0x12345678u) *reinterpret_cast<uU*>(&lds[off_lo]) = wl;
            return;
        }
        *reinterpret_cast<uU*>(&lds[off]) = w;
        if constexpr (C::SPLIT) *reinterpret_cast<uU*>(&lds[off_lo]) = wl;
    };
    auto write_unit = [&](H* lds, int u) {
#pragma unroll
        for (int k = 0; k < EPU / 2; ++k) unit_pair(u, k);
        unit_store(lds, u);
    };
    auto write_pass = [&](H* lds, int i) {
#pragma unroll
        for (int h = 0; h < UPP; ++h) write_unit(lds, UPP * i + h);
    };

    if constexpr (C::WS) {
        if (producer) {      // ---- producer waves: stage every chunk, one barrier per chunk in step with the MFMA waves ----
            issue_loads(0);
            if constexpr (kPrio) __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int i = 0; i < NPASS; ++i) write_pass(lds_all, i);
            if (p.nchunks > 1 && !(C::ABL & 2)) issue_loads(1);
            __syncthreads();
            for (int chunk = 0; chunk < p.nchunks; ++chunk) {
                H* nxt = lds_all + ((chunk + 1) & 1) * C::LDS_HALVES;
                if (chunk + 1 < p.nchunks && !(C::ABL & 2)) {
                    const unsigned long long t0 = memtime();
#pragma unroll
                    for (int i = 0; i < NPASS; ++i) write_pass(nxt, i);
                    if (chunk + 2 < p.nchunks) issue_loads(chunk + 2);
                    stage_ticks += memtime() - t0;
                }
                timed_barrier();
            }
            if constexpr ((C::ABL & 512) != 0) {
                unsigned long long* row = reinterpret_cast<unsigned long long*>(p.stats) + ((size_t)(gridDim.y * gridDim.x + blockIdx.y * gridDim.x + blockIdx.x) * 4 + wv) * 16;
                const unsigned long long te = memtime();
                if (lane == 0) { row[1] = bar_ticks; row[2] = stage_ticks; row[3] = te; }
            }
            return;
        }
    }

    // ---- MFMA operand addressing ----
    const int li = lane & 31, lh = lane >> 5;
    const int a_base = ((li >> C::LOGTW) * C::STRIDE + wm * MT * C::RPM * C::STRIDE) * RSH + ((li & (C::TW - 1)) * C::STRIDE) * PSH + lh * 8;
    const int ntile = by * 4 + wn;
    const bool nvalid = ntile * 32 < p.Cout;
    // packed weights: [ntile][chunk][tap][j = 0..1][plane][lane][8 halves] -> one group = GH halves (1 KiB per plane)
    constexpr int GH = 512 * C::PLANES;
    // (buffer addressing: scalar resource per N-tile + lane offset + SCALAR fragment offset -- the flat form cost two 64-bit
    // VALU adds per group)
    const __amdgpu_buffer_rsrc_t wrs = buf_rsrc(static_cast<const H*>(p.w) + ((size_t)(nvalid ? ntile : 0) * p.nchunks * TAPS) * (2 * GH));
    const unsigned wlane = lane * 16u;
    auto wload = [&](unsigned half_off) { return __builtin_bit_cast(h8, buf_load4(wrs, wlane, half_off * 2u)); };

    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // bias + temb + residual enter through the accumulator init (SPLIT: scaled by 1/unscale, a power of two: exact), so the
    // residual tile is fetched under the first chunk's staging instead of standing between the last MFMA and the stores
    // (measured: the epilogue's residual loads were 6 % of the SPLIT kernel).  Accumulator layout of the 32x32 MFMA: lane =
    // channel n, register r of tile t = pixel m = 32 t + 8 (r >> 2) + 4 (lane >> 5) + (r & 3): one element per lane (float32
    // in SPLIT mode, the 16-bit storage type otherwise), 128 / 64 B per half wave.  (Tiles 4 pixels wide keep the epilogue
    // form: there 4 (lane >> 5) crosses tile rows.)
    constexpr bool kAccInit = C::SPLIT || C::LOGTW >= 3;
    // acc = acc * macc + (bias + temb + residual) * inv.  At the start (acc = 0, FIRST): the init; SPLIT tiles whose scale 2^S
    // could overflow the product (asc.late) run it at the end instead with macc = 2^-S, inv = 1.
    // SPLIT (float32 results): bias + temb do NOT ride in the accumulators -- they are added by the epilogue's own FMA
    // (out = acc 2^-S + bias: one rounding, no extra instruction).  In the accumulators a bias that dwarfs the products (conv_in
    // on a 1e-4-scale input: bias 0.1) is re-rounded at ITS ulp by every one of the 3 x 18 MFMAs of a chunk where F.conv2d adds it
    // once -- harmless relative to the output scale, but a following GroupNorm divides by the SIGNAL's sigma (VERDICT r03 weak #2:
    // the whole-forward error at input scale 1e-4).  The residual keeps entering through the init: it is a tensor at the output's
    // own scale, and its loads hide under the first chunk's staging.
#ifdef CDX_TUNING
    const bool bias_late = C::SPLIT && !(p.abl & 2048);      // (tuning build, abl 2048: round 3's form, for the before / after record)
#else
    constexpr bool bias_late = C::SPLIT != 0;
#endif
    auto add_terms = [&](auto first_, float macc, float inv) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_)::value;
        const int n = ntile * 32 + li;
        if (!(nvalid && n < p.Cout)) return;
        float add = 0.f;
        if (!bias_late) {
            add = p.bias ? p.bias[n] : 0.f;
            if (p.temb) add += p.temb[(size_t)b * p.temb_ld + n];
            add *= inv;
        } else if (!p.residual) {
            return;                                          // nothing to add here: the accumulators stay as they are
        }
        if (p.residual && !(C::ABL & 16)) {
            // rows past the image end / columns past the row end read other (or no: bounded resource) pixels; those
            // accumulators are never stored
            constexpr unsigned es = C::SPLIT ? 4u : 2u;
            // (a wave of the 2 x 2 layout may own tile rows that lie wholly past the image -- H = 2 with 4-row tiles: `first` is then at
            // or beyond the tensor's end and the resource must be EMPTY, not wrap to 4 GiB of whatever follows the tensor: found by
            // the 800-case fuzz of round 3 as a GPU memory fault; the unmasked loads below rely on the bound)
            const size_t first = (((size_t)b * p.Hout + oy0w) * p.Wout + ox0) * p.Cout;
            const size_t total = (size_t)p.B * p.Hout * p.Wout * p.Cout;
            const size_t left = first < total ? (total - first) * es : 1;          // (1 byte: every 2- / 4-byte access is out of range, whatever a zero size means)
            const __amdgpu_buffer_rsrc_t rr = buf_rsrc(static_cast<const char*>(p.residual) + (first < total ? first : 0) * es,
                                                       left > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)left);
            const unsigned voff = ((unsigned)(4 * lh) * (unsigned)p.Cout + (unsigned)n) * es;
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int mb = t * 32 + 8 * (r >> 2) + (r & 3);
                    const unsigned pix = (unsigned)(mb >> C::LOGTW) * (unsigned)p.Wout + (unsigned)(mb & (C::TW - 1));
                    float rv;
                    if constexpr (C::SPLIT) rv = buf_load1(rr, voff, pix * (unsigned)p.Cout * es);
                    else rv = (float)__builtin_bit_cast(H, __builtin_amdgcn_raw_buffer_load_b16(rr, voff, pix * (unsigned)p.Cout * es, 0));
                    const float t_ = fmaf(rv, inv, add);
                    acc[t][r] = FIRST ? t_ : fmaf(acc[t][r], macc, t_);
                }
        } else {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = FIRST ? add : fmaf(acc[t][r], macc, add);
        }
    };
    if constexpr (kAccInit) {
        if (!asc.late) add_terms(std::true_type{}, 1.f, asc.inv);
    }

    h8 ring[PF][C::PLANES];
#pragma unroll
    for (int j = 0; j < PF; ++j)
#pragma unroll
        for (int pl_ = 0; pl_ < C::PLANES; ++pl_) ring[j][pl_] = wload(j * GH + pl_ * 512);

    // packed-epilogue layout (see below)
    const int q4 = li & 3;
    const int cq = ntile * 32 + (li & ~3);                    // first of this quad's 4 channels
    const bool quad_ok = nvalid && cq < p.Cout;               // (cout is a multiple of 4 or the tail is zero-weighted)
    // staging schedule inside a chunk: passes at groups G0 .. G0+NPASS-1, the loads of the chunk after next right behind
    constexpr int NU = UPP * NPASS;
    constexpr int G0 = GPC > NU + 1 ? GPC - NU - 1 : 0;
    static_assert(NU + 1 <= GPC || GPC <= 2 || !C::DB, "staging units must fit in the chunk's groups (1x1: done after the groups)");
    if constexpr (!C::WS) {
        issue_loads(0);
        stamp(1);
#pragma unroll
        for (int i = 0; i < NPASS; ++i) write_pass(lds_all, i);
        if (p.nchunks > 1 && !(C::ABL & 2)) issue_loads(1);
    }
    __syncthreads();
    stamp(2);
    h8 a[MT], al[MT];
    if constexpr (C::ABL & 8) {                 // ablation: operands read once
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            a[t] = *reinterpret_cast<const h8*>(&lds_all[a_base + t * C::RPM * C::STRIDE * RSH]);
            al[t] = *reinterpret_cast<const h8*>(&lds_all[a_base + t * C::RPM * C::STRIDE * RSH + 16]);
        }
    }
    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
        const H* lds = lds_all + (C::DB ? (chunk & 1) * C::LDS_HALVES : 0);
        H* nxt = lds_all + (C::DB ? ((chunk + 1) & 1) * C::LDS_HALVES : 0);
        const bool more = chunk + 1 < p.nchunks && !(C::ABL & 2);
        if (!C::WS && C::DB && (!nvalid || GPC <= 2)) {
            // a wave without output channels (or a 1x1 layer: two groups per chunk) stages in one go
            if (more) {
#pragma unroll
                for (int i = 0; i < NPASS; ++i) write_pass(nxt, i);
                if (chunk + 2 < p.nchunks) issue_loads(chunk + 2);
            }
        }
        if (nvalid) {
            const unsigned wc = (unsigned)chunk * (unsigned)(TAPS * 2 * GH);      // halves, scalar
#pragma unroll
            for (int g = 0; g < GPC; ++g) {
                // Staging groups (g in [G0, G0 + NU), next chunk exists): the unit's four channel computations are placed
                // BETWEEN quarters of the group's MFMAs, fenced by sched_barriers -- left alone the scheduler emits the unit's
                // ~60 VALU instructions as one clump in front of the group's first MFMA (and sched_group_barrier pipelines
                // did not move them).  A quarter = 3 MFMAs (96 matrix-pipe cycles) + ~14 VALU instructions.
                constexpr int NM = MT * (C::SPLIT ? 3 : 1);                  // MFMAs per group
                constexpr bool kInterleave = C::DB && GPC > 2 && NM >= 4 && !(C::ABL & 128);
                // (compile-time: in the LAST chunk the unit restages stale registers into the idle image -- harmless, and it
                // keeps the unrolled chunk body free of runtime branches, which would cut it into small scheduling regions)
                const bool stage_here = !C::WS && C::DB && GPC > 2 && !(C::ABL & 2) && g >= G0 && g < G0 + NU;
                if (stage_here && !kInterleave) write_unit(nxt, g - G0);
                if (!C::WS && C::DB && GPC > 2 && more && g == G0 + NU && chunk + 2 < p.nchunks && !(C::ABL & 64)) issue_loads(chunk + 2);
                const int tap = g >> 1, j = g & 1, ky = tap / C::KS, kx = tap % C::KS;
                int ab = a_base;
                asm volatile("" : "+v"(ab));                 // no cross-tap CSE of LDS reads (see conv_kernel.h)
                __builtin_assume((ab & 7) == 0);
                // operand reads of group gg (same chunk): A fragment of M-tile t, hi plane / lo plane
                auto rd_a = [&](int gg, int t) __attribute__((always_inline)) {
                    const int tp = gg >> 1, jj = gg & 1, kyy = tp / C::KS, kxx = tp % C::KS;
                    a[t] = *reinterpret_cast<const h8*>(&lds[ab + (t * C::RPM * C::STRIDE + kyy) * RSH + kxx * PSH + jj * 16]);
                };
                auto rd_al = [&](int gg, int t) __attribute__((always_inline)) {
                    const int tp = gg >> 1, jj = gg & 1, kyy = tp / C::KS, kxx = tp % C::KS;
                    al[t] = *reinterpret_cast<const h8*>(&lds[ab + (t * C::RPM * C::STRIDE + kyy) * RSH + kxx * PSH + KC + jj * 16]);
                };
                if constexpr (!(C::ABL & 8)) {
                    if (!kPipe || g == 0) {                  // (kPipe: groups 1.. were read under the previous group's MFMAs)
#pragma unroll
                        for (int t = 0; t < MT; ++t) {
                            rd_a(g, t);
                            if constexpr (C::SPLIT) rd_al(g, t);
                        }
                    }
                }
                const h8 bq = ring[g % PF][0];
                h8 bl;
                if constexpr (C::SPLIT) bl = ring[g % PF][1];
                // (kPipe refills each plane of the ring slot right behind the last MFMA that reads it -- below -- so that the old
                // fragment and its in-flight successor never hold registers at the same time: the pipelined reads need them)
                if constexpr (!(C::ABL & 4) && !kPipe) {
#pragma unroll
                    for (int pl_ = 0; pl_ < C::PLANES; ++pl_)      // wraps into the next chunk / tail pad
                        ring[g % PF][pl_] = wload(wc + (unsigned)((g + PF) * GH + pl_ * 512));
                }
                if constexpr (kPipe) {
                    // Software-pipelined operand reads (MFMA waves of the wave-specialised tile): the NEXT group's A fragments are
                    // read under THIS group's MFMAs, into the registers the group has just finished with -- no extra registers.
                    // SPLIT order hi*hi, hi*lo, lo*hi: after the 8 MFMAs that use the hi fragments a[0..3] are dead and the next
                    // group's hi reads fly under the four lo*hi MFMAs (128 matrix-pipe cycles ~ the LDS latency); each lo fragment
                    // is re-read right behind its own lo*hi MFMA and is not needed for 8 MFMAs.  Without this every group began with
                    // 8 ds_read_b128 and an lgkmcnt wait in front of its first MFMA, covered only by the SIMD's other MFMA wave
                    // (tools/ws_stamps.py: a wave keeps the pipe ~58 % busy alone).  sched_barriers pin the order: hipcc sinks the
                    // reads back in front of their own MFMAs otherwise (round 3's persistent kernel: 0.76 -> 0.70 ms).
                    const bool nxt_in_chunk = g + 1 < GPC;
                    if constexpr (C::SPLIT) {
#pragma unroll
                        for (int t = 0; t < MT; ++t) acc[t] = mfma_32x32x16(a[t], bq, acc[t]);
#pragma unroll
                        for (int t = 0; t < MT; ++t) acc[t] = mfma_32x32x16(a[t], bl, acc[t]);
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (!(C::ABL & 4)) ring[g % PF][1] = wload(wc + (unsigned)((g + PF) * GH + 512));      // (lo plane: dead now)
                        if constexpr (!(C::ABL & 8)) {
                            if (nxt_in_chunk) {
#pragma unroll
                                for (int t = 0; t < MT; ++t) rd_a(g + 1, t);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int t = 0; t < MT; ++t) {
                            acc[t] = mfma_32x32x16(al[t], bq, acc[t]);
                            if constexpr (!(C::ABL & 8)) {
                                if (nxt_in_chunk) rd_al(g + 1, t);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (!(C::ABL & 4)) ring[g % PF][0] = wload(wc + (unsigned)((g + PF) * GH));
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
#pragma unroll
                        for (int t = 0; t < MT; ++t) {
                            acc[t] = mfma_32x32x16(a[t], bq, acc[t]);
                            if constexpr (!(C::ABL & 8)) {
                                if (nxt_in_chunk) rd_a(g + 1, t);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (!(C::ABL & 4)) ring[g % PF][0] = wload(wc + (unsigned)((g + PF) * GH));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    continue;
                }
                // MFMA m of the group: term m / MT (hi*hi, lo*hi, hi*lo; lo*lo <= 2^-22 of the product is dropped), tile m % MT
                auto mfma_at = [&](int m) {
                    const int t = m % MT, term = m / MT;
                    acc[t] = mfma_32x32x16(term == 1 ? al[t] : a[t], term == 2 ? bl : bq, acc[t]);
                };
                if (kInterleave && stage_here) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
#pragma unroll
                        for (int m = k * NM / 4; m < (k + 1) * NM / 4; ++m) mfma_at(m);      // (NM = 6: quarters of 1, 2, 1, 2)
                        if ((k & 1) == 0 && k / 2 < EPU / 2) unit_pair(g - G0, k / 2);      // pairs ride on quarters 0 and 2
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    unit_store(nxt, g - G0);
                } else {
#pragma unroll
                    for (int m = 0; m < NM; ++m) mfma_at(m);
                }
            }
        }
        timed_barrier();
        if (chunk < 8) stamp(3 + chunk);
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
        stamp(12);
        if constexpr (C::WS && (C::ABL & 512) != 0) {
            if (lane == 0) reinterpret_cast<unsigned long long*>(p.stats)[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wv) * 16 + 15] = bar_ticks;
        }
        return;
    }
    const int n = ntile * 32 + li;
    const bool nok = n < p.Cout;
    if constexpr (C::SPLIT) {
        // float32 out, straight from the accumulator layout (lane = channel): one dword per lane and register, 128 B per half
        // wave and pixel -- full cache lines, no cross-lane transposes, no per-store address arithmetic (scalar offset per
        // register, lane offset fixed), GroupNorm sums per lane = per channel.  (The 4 x 4 quad-transposed form below cost
        // ~43 vector instructions per 16-byte store, 8 of them quarter-rate integer multiplies for the flat address.)
        constexpr int ROWS = (MT * 32) >> C::LOGTW;
        const bool full = oy0w + ROWS <= p.Hout && ox0 + C::TW <= p.Wout;          // wave-uniform
        // (phase launches: output pixel (y, x) lands at (os y + ody, os x + odx) of the os-times larger tensor; os = 1 otherwise)
        const unsigned os = (unsigned)p.ostep, WoutF = (unsigned)p.Wout * os;
        const size_t first = (((size_t)b * p.Hout * os + (size_t)oy0w * os + p.ody) * WoutF + (size_t)ox0 * os + p.odx) * p.out_ld;
        const size_t total = (size_t)p.B * p.Hout * os * WoutF * p.out_ld;
        const size_t left = first < total ? (total - first) * 4 : 1;              // (tile rows wholly past the image: a resource no dword fits into)
        constexpr unsigned kDrop = 0x80000000u;                                    // beyond any resource: the store is dropped
        const __amdgpu_buffer_rsrc_t ors = buf_rsrc(static_cast<float*>(p.out) + (first < total ? first : 0), left > 0x7FFFFFFFull ? 0x7FFFFFFFu : (unsigned)left);
        const unsigned vbase = nok ? ((unsigned)(4 * lh) * os * (unsigned)p.out_ld + (unsigned)n) * 4u : kDrop;
        float un = asc.un;
        if (asc.late && (p.residual || !bias_late)) {   // (rare: see ActScale) outputs first, then the additive terms at their own scale
            add_terms(std::false_type{}, un, 1.f);   // (lanes without an output channel keep raw accumulators: never stored)
            un = 1.f;
        }
        float addv = 0.f;                            // bias + temb of this lane's channel, added by the store loop's FMA
        if (bias_late && nok) {
            addv = p.bias ? p.bias[n] : 0.f;
            if (p.temb) addv += p.temb[(size_t)b * p.temb_ld + n];
        }
        // (the sums variant also serves amax_out: a caller that wants only one of them pays for both)
        const bool want_stats = (p.stats || p.amax_out) && !(C::ABL & 32);
        double s1 = 0, s2 = 0;
        float am = 0.f;
        auto direct = [&](auto full_, auto has_stats) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int mb = t * 32 + 8 * (r >> 2) + (r & 3);
                    const int row = mb >> C::LOGTW, col = mb & (C::TW - 1);
                    const unsigned soff = ((unsigned)row * os * WoutF + (unsigned)col * os) * (unsigned)p.out_ld * 4u;
                    const float x = fmaf(acc[t][r], un, addv);
                    bool ok = true;
                    if constexpr (!decltype(full_)::value) ok = oy0w + row < p.Hout && ox0 + col + 4 * lh < p.Wout;
                    if constexpr (decltype(has_stats)::value) {
                        const float xs = ok ? x : 0.f;
                        const double d = (double)xs;
                        s1 += d;
                        s2 = fma(d, d, s2);
                        am = fmaxf(am, fabsf(xs));                      // (NaN skipped, Inf kept)
                    }
                    buf_store1(ors, ok ? vbase : kDrop, soff, x);       // (last use of x: the tie in buf_store1 then costs no copy)
                }
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        if (full) { if (want_stats) direct(T_{}, T_{}); else direct(T_{}, F_{}); }
        else      { if (want_stats) direct(F_{}, T_{}); else direct(F_{}, F_{}); }
        stamp(12);
        if constexpr (C::WS && (C::ABL & 512) != 0) {
            if (lane == 0) reinterpret_cast<unsigned long long*>(p.stats)[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wv) * 16 + 15] = bar_ticks;
        }
        if (p.stats && !(C::ABL & 32)) {
            s1 += __shfl_xor(s1, 32);                  // the two lane halves hold different pixels of the same channel
            s2 += __shfl_xor(s2, 32);
            if (lh == 0 && nok) {
                const int slot = p.slot_base + (ty * p.tiles_x + tx) * p.stats_wm + (WM == 2 ? wm : 0);
                const int nslots = p.nslots_total ? p.nslots_total : p.tiles_y * p.tiles_x * p.stats_wm;
                double* o = p.stats + (((size_t)b * nslots + slot) * p.Cout + n) * 2;
                o[0] = s1;
                o[1] = s2;
                if (WM == 1 && p.stats_wm == 2) {      // a 1 x 4 block of a layer whose LAST block runs 2 x 2: second slot = 0
                    o[(size_t)p.Cout * 2] = 0.0;
                    o[(size_t)p.Cout * 2 + 1] = 0.0;
                }
            }
        }
        if (p.amax_out && !(C::ABL & 32)) {
            if (!nok) am = 0.f;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) am = fmaxf(am, __shfl_xor(am, off));
            if (lane == 0) amax_publish(p.amax_out, b, (bx0 * 4 + wv) * 5 + by, am);
        }
        return;
    }
    if constexpr (kAccInit) {
        // 16-bit out from the accumulator layout: lanes (2c, 2c+1) exchange one value per register pair (r, r+1 = the next
        // pixel of the row), so the even lane stores channels (2c, 2c+1) of pixel r and the odd lane those of pixel r+1 --
        // one packed dword per lane, 64 B per pixel and half wave, scalar offset per pair, GroupNorm sums per lane = channel
        // (from the float32 accumulators, before rounding).  5 vector instructions per pair + the sums, where the
        // quad-transposed form below cost ~30 per 8-byte store.
        if (!p.out_f32 && !(p.Cout & 1)) {
            constexpr int ROWS = (MT * 32) >> C::LOGTW;
            const bool full = oy0w + ROWS <= p.Hout && ox0 + C::TW <= p.Wout;      // wave-uniform
            const size_t first = (((size_t)b * p.Hout + oy0w) * p.Wout + ox0) * p.out_ld;
            const size_t total = (size_t)p.B * p.Hout * p.Wout * p.out_ld;
            const size_t left = first < total ? (total - first) * 2 : 1;          // (tile rows wholly past the image: a resource no dword fits into)
            constexpr unsigned kDrop = 0x80000000u;
            const __amdgpu_buffer_rsrc_t ors = buf_rsrc(static_cast<H*>(p.out) + (first < total ? first : 0), left > 0x7FFFFFFFull ? 0x7FFFFFFFu : (unsigned)left);
            const int odd = li & 1;
            const unsigned vbase = nok ? ((unsigned)(4 * lh + odd) * (unsigned)p.out_ld + (unsigned)(n - odd)) * 2u : kDrop;
            const unsigned rot = odd ? 16u : 0u;
            const bool want_stats = p.stats && !(C::ABL & 32);
            double s1 = 0, s2 = 0;
            auto direct = [&](auto full_, auto has_stats) __attribute__((always_inline)) {
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const int mb = t * 32 + 8 * (r >> 2) + (r & 3);
                        const int row = mb >> C::LOGTW, col = mb & (C::TW - 1);
                        const unsigned soff = ((unsigned)row * (unsigned)p.Wout + (unsigned)col) * (unsigned)p.out_ld * 2u;
                        const float x0 = acc[t][r], x1 = acc[t][r + 1];
                        bool ok0 = true, ok1 = true;
                        if constexpr (!decltype(full_)::value) {
                            const bool rok = oy0w + row < p.Hout;
                            ok0 = rok && ox0 + col + 4 * lh < p.Wout;
                            ok1 = rok && ox0 + col + 4 * lh + 1 < p.Wout;
                        }
                        if constexpr (decltype(has_stats)::value) {
                            const double d0 = ok0 ? (double)x0 : 0.0, d1 = ok1 ? (double)x1 : 0.0;
                            s1 += d0;
                            s2 = fma(d0, d0, s2);
                            s1 += d1;
                            s2 = fma(d1, d1, s2);
                        }
                        const float got = quad_xor1(odd ? x0 : x1);          // the neighbour lane's value for MY pixel
                        const float keep = odd ? x1 : x0;
                        using h2 = __attribute__((ext_vector_type(2))) H;
                        const unsigned pk = __builtin_bit_cast(unsigned, h2{(H)keep, (H)got});      // even lane: (n, n+1)
                        buf_store1(ors, (odd ? ok1 : ok0) ? vbase : kDrop, soff, __builtin_amdgcn_alignbit(pk, pk, rot));   // odd: (n-1, n)
                    }
            };
            using T_ = std::true_type;
            using F_ = std::false_type;
            if (full) { if (want_stats) direct(T_{}, T_{}); else direct(T_{}, F_{}); }
            else      { if (want_stats) direct(F_{}, T_{}); else direct(F_{}, F_{}); }
            stamp(12);
            if constexpr (C::WS && (C::ABL & 512) != 0) {
                if (lane == 0) reinterpret_cast<unsigned long long*>(p.stats)[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wv) * 16 + 15] = bar_ticks;
            }
            if (want_stats) {
                s1 += __shfl_xor(s1, 32);
                s2 += __shfl_xor(s2, 32);
                if (lh == 0 && nok) {
                    const int slot = (ty * p.tiles_x + tx) * p.stats_wm + (WM == 2 ? wm : 0);
                    const int nslots = p.tiles_y * p.tiles_x * p.stats_wm;
                    double* o = p.stats + (((size_t)b * nslots + slot) * p.Cout + n) * 2;
                    o[0] = s1;
                    o[1] = s2;
                    if (WM == 1 && p.stats_wm == 2) {
                        o[(size_t)p.Cout * 2] = 0.0;
                        o[(size_t)p.Cout * 2 + 1] = 0.0;
                    }
                }
            }
            return;
        }
    }
    float add = 0.f;
    if (nok) {
        add = p.bias ? p.bias[n] : 0.f;
        if (p.temb) add += p.temb[(size_t)b * p.temb_ld + n];
    }
    // 16-bit storage -- packed epilogue: 4x4 blocks (4 consecutive pixels x the quad's 4 channels) are transposed across lane quads in
    // registers, so every lane stores / loads 4 consecutive channels of ONE pixel (8 B fp16, 16 B fp32) -- 4x fewer,
    // 4x wider memory instructions than the accumulator layout allows.  GroupNorm sums are reduced in that layout.
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    auto epilogue = [&](auto has_res, auto has_stats, auto out32) __attribute__((always_inline)) {
        using res_t = h4;
        using rel_t = H;
        res_t rv[MT][4];
        if constexpr (decltype(has_res)::value) {      // one batch of 8/16-byte loads (a load in the last chunk instead
#pragma unroll                                          // would queue the weight ring behind HBM misses: vmcnt is in-order)
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int m = t * 32 + 8 * k + q4 + 4 * lh;
                    const int oy = min(oy0w + (m >> C::LOGTW), p.Hout - 1), ox = min(ox0 + (m & (C::TW - 1)), p.Wout - 1);
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
                for (int c = 0; c < 4; ++c) x[c] = kAccInit ? acc[t][4 * k + c] : acc[t][4 * k + c] + add;
                quad_transpose(x, q4);                        // now: pixel 8k + q4 (+4 lh) of tile t, channels cq..cq+3
                const int m = t * 32 + 8 * k + q4 + 4 * lh;
                const int oy = oy0w + (m >> C::LOGTW), ox = ox0 + (m & (C::TW - 1));
                if (quad_ok && oy < p.Hout && ox < p.Wout) {
                    const size_t pix = ((size_t)b * p.Hout + oy) * p.Wout + ox;
                    if constexpr (decltype(has_res)::value) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) x[c] += (float)rv[t][k][c];
                    }
                    if constexpr (decltype(out32)::value) {
                        *reinterpret_cast<f32x4*>(static_cast<float*>(p.out) + pix * p.out_ld + cq) = f32x4{x[0], x[1], x[2], x[3]};
                    } else {
                        *reinterpret_cast<h4*>(static_cast<H*>(p.out) + pix * p.out_ld + cq) =
                            h4{(H)x[0], (H)x[1], (H)x[2], (H)x[3]};
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
    if constexpr (kAccInit) {   // 16-bit storage, bias / temb / residual already in the accumulators
        if (p.out_f32) epilogue(F_{}, F_{}, T_{});
        else if (p.stats) epilogue(F_{}, T_{}, F_{});
        else epilogue(F_{}, F_{}, F_{});
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
            // p.stats_wm slots per tile: a layer with a 2 x 2 last block has two (its 1 x 4 blocks fill the first and zero the second)
            const int slot = (ty * p.tiles_x + tx) * p.stats_wm + (WM == 2 ? wm : 0);
            const int nslots = p.tiles_y * p.tiles_x * p.stats_wm;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (cq + c < p.Cout) {
                    double* o = p.stats + (((size_t)b * nslots + slot) * p.Cout + cq + c) * 2;
                    o[0] = s1[c];
                    o[1] = s2[c];
                    if (WM == 1 && p.stats_wm == 2) {
                        o[(size_t)p.Cout * 2] = 0.0;
                        o[(size_t)p.Cout * 2 + 1] = 0.0;
                    }
                }
            }
        }
    }
}

template <class C, int STG>
__global__ __launch_bounds__(256, 2) void conv16_kernel(const Conv16Params p) {
    // two halo images: chunk c+1 is staged into the other one WHILE chunk c's MFMAs run (one unit per MFMA group), one
    // barrier per chunk (DB = 0: one image, two barriers)
    __shared__ __attribute__((aligned(16))) typename C::H lds_all[(C::DB ? 2 : 1) * C::LDS_HALVES];
    int bx, by;
    conv16_block<C>(bx, by);
    if constexpr (C::MT == 4) {
        // block-uniform: the last channel block of a layer with <= 64 channels left runs the 2 x 2 wave layout
        if (by == (int)gridDim.y - 1 && conv16_tail_2x2(p.Cout, C::MT)) {
            conv16_body<C, STG, 2>(p, lds_all, bx, by);
            return;
        }
    }
    conv16_body<C, STG, 1>(p, lds_all, bx, by);
}

// wave-specialised form: 8 waves, two workgroups per CU = 4 waves per SIMD (<= 128 VGPRs)
template <class C, int STG>
__global__ __launch_bounds__(512, 4) void conv16_ws_kernel(const Conv16Params p) {
    static_assert(C::WS && C::DB, "conv16_ws_kernel runs the WS configurations");
    __shared__ __attribute__((aligned(16))) typename C::H lds_all[2 * C::LDS_HALVES];
    int bx, by;
    conv16_block<C>(bx, by);
    if constexpr (C::MT == 4) {
        if (by == (int)gridDim.y - 1 && conv16_tail_2x2(p.Cout, C::MT)) {
            conv16_body<C, STG, 2>(p, lds_all, bx, by);
            return;
        }
    }
    conv16_body<C, STG, 1>(p, lds_all, bx, by);
}

#ifdef CDX_TUNING
#include <stdlib.h>
// tuning build only: CDX_NO_WS=1 routes the wave-specialised launches to the 4-wave kernel of the same configuration (same-box
// A/B of whole bench runs: tools/session/gpu_r3h.sh)
inline bool ws_disabled() {
    static const bool off = [] { const char* e = getenv("CDX_NO_WS"); return e && e[0] == '1'; }();
    return off;
}
#endif

template <class C>
inline int conv16_launch(const Conv16Params& p, hipStream_t stream);

template <class C>
inline int conv16_ws_launch(const Conv16Params& p, hipStream_t stream) {
#ifdef CDX_TUNING
    if (ws_disabled()) return conv16_launch<Conv16Cfg<C::KS, C::STRIDE, C::LOGTW, C::MT, C::PF, C::ABL, C::SPLIT, C::DB, C::BF, 0>>(p, stream);
#endif
    dim3 grid(p.tiles_x * p.tiles_y * p.B, ceil_div(p.Cout, C::BN));
    switch (p.gn ? (p.silu ? 2 : 1) : (p.silu ? 3 : 0)) {
        case 0: hipLaunchKernelGGL((conv16_ws_kernel<C, 0>), grid, dim3(512), 0, stream, p); break;
        case 1: hipLaunchKernelGGL((conv16_ws_kernel<C, 1>), grid, dim3(512), 0, stream, p); break;
        case 2: hipLaunchKernelGGL((conv16_ws_kernel<C, 2>), grid, dim3(512), 0, stream, p); break;
        default: hipLaunchKernelGGL((conv16_ws_kernel<C, 3>), grid, dim3(512), 0, stream, p); break;
    }
    return check_launch();
}

template <class C>
inline int conv16_launch(const Conv16Params& p, hipStream_t stream) {
    dim3 grid(p.tiles_x * p.tiles_y * p.B, ceil_div(p.Cout, C::BN));
    // (tuning build, abl 1024: unused dynamic LDS leaves room for ONE workgroup per CU -- the rate of a wave alone on its SIMD)
    const unsigned dyn = (p.abl & 1024) ? 88u * 1024u : 0u;
    auto go = [&](auto kern) {
        if (dyn) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        hipLaunchKernelGGL(kern, grid, dim3(256), dyn, stream, p);
    };
    switch (p.gn ? (p.silu ? 2 : 1) : (p.silu ? 3 : 0)) {
        case 0: go(conv16_kernel<C, 0>); break;
        case 1: go(conv16_kernel<C, 1>); break;
        case 2: go(conv16_kernel<C, 2>); break;
        default: go(conv16_kernel<C, 3>); break;
    }
    return check_launch();
}

int conv16_dispatch(int ks, int stride, int logtw, int mt, bool bf, const Conv16Params& p, hipStream_t stream);

}  // namespace cdx
