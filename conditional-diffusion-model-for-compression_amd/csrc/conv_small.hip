// 3x3 stride-1 convolution with at most 4 output channels (the UNet's conv_out: 128 -> 3 at full resolution).
// With N = 3 a 32-wide MFMA tile wastes 29/32 of the matrix pipe (measured: 10.9 TF, 0.67 ms per forward at configs[1]);
// here the products run on the vector ALU instead: thread = output pixel, the channel loop reads the same LDS halo
// image the other kernels stage (GroupNorm / SiLU / upsample / concat fused while staging), and the weights -- uniform
// across the wave -- come through the scalar cache straight out of the MFMA-packed image: for a (chunk, tap, s) group the
// 64 bytes at lane 0 hold W[n = 0..3][channels 8s .. 8s+3] and the 64 bytes at lane 32 the channels 8s+4 .. 8s+7.
// Every FMA takes its weight as an SGPR operand: 9 x Cin x NCO FMAs and 9 x Cin / 4 ds_read_b128 per pixel.
// Bound: VALU (3456 FMA issues per wave at 128 -> 3), ~0.1 ms at configs[1].  No reference file exists to cite.
#include <cstdlib>
#include "conv_kernel.h"

namespace cdx {

namespace {

constexpr int S_TW = 32, S_TH = 8, S_HH = S_TH + 2, S_HW = S_TW + 2;
constexpr int S_KC = CDX_CONV_KC, S_PS = S_KC + 4;
constexpr int S_RS = ((S_HW * S_PS + 63) / 64) * 64;
constexpr int S_NPIX = S_HH * S_HW, S_NPASS = (S_NPIX + 31) / 32;

template <int NCO>
__global__ __launch_bounds__(256, 2) void conv_small_kernel(const ConvParams p) {
    __shared__ __attribute__((aligned(16))) float lds[S_HH * S_RS];
    const int tid = threadIdx.x;
    int bx = blockIdx.x;
    const int tx = bx % p.tiles_x;
    bx /= p.tiles_x;
    const int ty = bx % p.tiles_y;
    const int b = bx / p.tiles_y;
    const int oy0 = ty * S_TH, ox0 = tx * S_TW;
    const int iy0 = oy0 - 1, ix0 = ox0 - 1;
    const int Hv = p.Hin << p.ups, Wv = p.Win << p.ups;

    // ---- halo loader: thread -> (pixel slot pl, channel quad q), S_NPASS passes of 32 slots ----
    const int q = tid & 7, pl = tid >> 3;
    int soff[S_NPASS];
    unsigned vmask = 0;
#pragma unroll
    for (int i = 0; i < S_NPASS; ++i) {
        const int hp = i * 32 + pl;
        const int hy = hp / S_HW, hx = hp - hy * S_HW;
        const int iy = iy0 + hy, ix = ix0 + hx;
        const bool ok = hp < S_NPIX && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv;
        soff[i] = ok ? ((b * p.Hin + (iy >> p.ups)) * p.Win + (ix >> p.ups)) : 0;
        vmask |= ok ? (1u << i) : 0u;
    }
    f32x4 pre[S_NPASS];
    f32x4 gsc, gsh;
    bool cvalid;
    auto issue_loads = [&](int chunk) {
        const int s = chunk >= p.nchunk0;
        const int cl = (s ? chunk - p.nchunk0 : chunk) * S_KC + q * 4;
        const int cs = p.csrc[s];
        cvalid = cl < cs;
        const float* __restrict__ base = p.src[s] + (cvalid ? cl : 0);      // padding lanes: safe address, zeroed at write time
#pragma unroll
        for (int i = 0; i < S_NPASS; ++i) pre[i] = *reinterpret_cast<const f32x4*>(base + (size_t)soff[i] * cs);
        if (p.gn) {
            const int cg = cvalid ? (s ? p.csrc[0] : 0) + cl : 0;
            gsc = *reinterpret_cast<const f32x4*>(p.gscale + (size_t)b * p.ctot + cg);
            gsh = *reinterpret_cast<const f32x4*>(p.gshift + (size_t)b * p.ctot + cg);
        }
    };
    auto write_lds = [&]() {
#pragma unroll
        for (int i = 0; i < S_NPASS; ++i) {
            const int hp = i * 32 + pl;
            const int hy = hp / S_HW, hx = hp - hy * S_HW;
            f32x4 v = pre[i];
            const bool ok = cvalid && ((vmask >> i) & 1u);
            if (p.gn) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], gsc[e], gsh[e]);
            }
            if (p.silu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = silu_f<false>(v[e]);
            }
            if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (hp < S_NPIX) *reinterpret_cast<f32x4*>(&lds[hy * S_RS + hx * S_PS + q * 4]) = v;
        }
    };

    const int py = tid >> 5, px = tid & 31;
    const int a_base = py * S_RS + px * S_PS;
    float acc[NCO];
#pragma unroll
    for (int n = 0; n < NCO; ++n) acc[n] = 0.f;

    issue_loads(0);
    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
        write_lds();
        __syncthreads();
        if (chunk + 1 < p.nchunks) issue_loads(chunk + 1);
        // packed weights of output tile 0: [chunk][tap][s][lane][4]; wave-uniform addresses -> scalar loads
        const float* __restrict__ wc = p.w + (size_t)chunk * (9 * 1024);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(&lds[a_base + ky * S_RS + kx * S_PS + s * 8 + h * 4]);
                    const float* __restrict__ wq = wc + (tap * 4 + s) * 256 + h * 128;      // [n][e]
#pragma unroll
                    for (int n = 0; n < NCO; ++n)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[n] = fmaf(a[e], wq[n * 4 + e], acc[n]);
                }
            }
        }
        __syncthreads();
    }

    const int oy = oy0 + py, ox = ox0 + px;
    if (oy < p.Hout && ox < p.Wout) {
        const size_t pix = ((size_t)b * p.Hout + oy) * p.Wout + ox;
#pragma unroll
        for (int n = 0; n < NCO; ++n) {
            if (n < p.Cout) {
                float v = acc[n];
                if (p.bias) v += p.bias[n];
                if (p.temb) v += p.temb[(size_t)b * p.temb_ld + n];
                if (p.residual) v += p.residual[pix * p.Cout + n];
                p.out[pix * p.out_ld + n] = v;
            }
        }
    }
}

// MFMA form: v_mfma_f32_4x4x1_16B_f32 = 16 independent 4x4 outer products per instruction.  Block b = lanes 4b..4b+3 =
// four consecutive pixels; A[i] = the activation of pixel 4b+i (each lane's own LDS read), B[j] = W[channel][cout j]
// (lane 4b+j reads the 16 bytes W[j][4 channels] of the chunk's weight image in LDS), D[i][j] accumulates in lane 4b+j,
// register i.  One MFMA per input channel: 2 x ds_read_b128 + 4 MFMAs per (tap, 4 channels), no VALU in the loop, no
// scalar-load latency (the VALU form above waits lgkmcnt(0) on every weight fetch).
template <int NCO>
__global__ __launch_bounds__(256, 2) void conv_small_mfma_kernel(const ConvParams p) {
    __shared__ __attribute__((aligned(16))) float lds[S_HH * S_RS + 9 * 8 * 16];
    float* const wl = lds + S_HH * S_RS;          // [tap][octet s*2+h][cout 4][e 4]
    const int tid = threadIdx.x;
    int bx = blockIdx.x;
    const int tx = bx % p.tiles_x;
    bx /= p.tiles_x;
    const int ty = bx % p.tiles_y;
    const int b = bx / p.tiles_y;
    const int oy0 = ty * S_TH, ox0 = tx * S_TW;
    const int iy0 = oy0 - 1, ix0 = ox0 - 1;
    const int Hv = p.Hin << p.ups, Wv = p.Win << p.ups;

    const int q = tid & 7, pl = tid >> 3;
    int soff[S_NPASS];
    unsigned vmask = 0;
#pragma unroll
    for (int i = 0; i < S_NPASS; ++i) {
        const int hp = i * 32 + pl;
        const int hy = hp / S_HW, hx = hp - hy * S_HW;
        const int iy = iy0 + hy, ix = ix0 + hx;
        const bool ok = hp < S_NPIX && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv;
        soff[i] = ok ? ((b * p.Hin + (iy >> p.ups)) * p.Win + (ix >> p.ups)) : 0;
        vmask |= ok ? (1u << i) : 0u;
    }
    f32x4 pre[S_NPASS];
    f32x4 gsc, gsh, wpre[2];
    bool cvalid;
    auto issue_loads = [&](int chunk) {
        const int s = chunk >= p.nchunk0;
        const int cl = (s ? chunk - p.nchunk0 : chunk) * S_KC + q * 4;
        const int cs = p.csrc[s];
        cvalid = cl < cs;
        const float* __restrict__ base = p.src[s] + (cvalid ? cl : 0);
#pragma unroll
        for (int i = 0; i < S_NPASS; ++i) pre[i] = *reinterpret_cast<const f32x4*>(base + (size_t)soff[i] * cs);
        if (p.gn) {
            const int cg = cvalid ? (s ? p.csrc[0] : 0) + cl : 0;
            gsc = *reinterpret_cast<const f32x4*>(p.gscale + (size_t)b * p.ctot + cg);
            gsh = *reinterpret_cast<const f32x4*>(p.gshift + (size_t)b * p.ctot + cg);
        }
        // this chunk's weights out of the MFMA-packed image of output tile 0: group (tap, s) = 1 KiB; the 64 bytes at lane 0
        // (h = 0) and at lane 32 (h = 1) hold W[n = 0..3][4 channels]: 288 float4 per chunk, threads 0..255 (+ 0..31 again)
        const float* __restrict__ wc = p.w + (size_t)chunk * (9 * 1024);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int v = tid + r * 256;                    // float4 index: ((tap*4 + s)*2 + h)*4 + n
            const int n = v & 3, h = (v >> 2) & 1, ts = v >> 3;
            wpre[r] = v < 288 ? *reinterpret_cast<const f32x4*>(wc + ts * 256 + h * 128 + n * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto write_lds = [&]() {
#pragma unroll
        for (int i = 0; i < S_NPASS; ++i) {
            const int hp = i * 32 + pl;
            const int hy = hp / S_HW, hx = hp - hy * S_HW;
            f32x4 v = pre[i];
            const bool ok = cvalid && ((vmask >> i) & 1u);
            if (p.gn) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], gsc[e], gsh[e]);
            }
            if (p.silu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = silu_f<false>(v[e]);
            }
            if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (hp < S_NPIX) *reinterpret_cast<f32x4*>(&lds[hy * S_RS + hx * S_PS + q * 4]) = v;
        }
        *reinterpret_cast<f32x4*>(&wl[tid * 4]) = wpre[0];
        if (tid < 32) *reinterpret_cast<f32x4*>(&wl[(tid + 256) * 4]) = wpre[1];
    };

    const int py = tid >> 5, px = tid & 31;
    const int a_base = py * S_RS + px * S_PS;
    const int w_base = (tid & 3) * 4;               // this lane's cout column of every [4][4] weight block
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};               // acc[i] = out[pixel 4*(lane/4) + i][cout lane & 3]

    issue_loads(0);
    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
        write_lds();
        __syncthreads();
        if (chunk + 1 < p.nchunks) issue_loads(chunk + 1);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int o = 0; o < 8; ++o) {           // o = s*2 + h: channels 4*o .. 4*o+3 of the chunk
                const f32x4 a = *reinterpret_cast<const f32x4*>(&lds[a_base + ky * S_RS + kx * S_PS + o * 4]);
                const f32x4 w = *reinterpret_cast<const f32x4*>(&wl[(tap * 8 + o) * 16 + w_base]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[e], w[e], acc, 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // lane (block blk = lane / 4, column j = lane & 3), register i -> pixel 4*blk + i of this wave's 64 pixels, cout j
    const int j = tid & 3;
    if (j < NCO && j < p.Cout) {
        float add = p.bias ? p.bias[j] : 0.f;
        if (p.temb) add += p.temb[(size_t)b * p.temb_ld + j];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pix_in_wg = (tid & ~63) + ((tid & 63) >> 2) * 4 + i;      // = wave * 64 + 4*blk + i
            const int oy = oy0 + (pix_in_wg >> 5), ox = ox0 + (pix_in_wg & 31);
            if (oy < p.Hout && ox < p.Wout) {
                const size_t pix = ((size_t)b * p.Hout + oy) * p.Wout + ox;
                float v = acc[i] + add;
                if (p.residual) v += p.residual[pix * p.Cout + j];
                p.out[pix * p.out_ld + j] = v;
            }
        }
    }
}

// GEMM form (CDX_TILE_SMALL_GEMM): the two kernels above give every output pixel its own pass over the 9 x Cin LDS
// neighbourhood -- 216 ds_read_b128 per wave and 32-channel chunk, which is what bounds them (the LDS pipe: 0.24 ms at
// configs[1] where the tensor streams in 0.11).  Here the taps become OUTPUT columns: Y[p][tap*3 + co] = sum_c X[p][c] W[co][c][tap]
// is ONE plain GEMM over the pixels of the 18 x 34 halo (N = 27 of the 32 columns of v_mfma_f32_32x32x2_f32, K = Cin), whose A
// operand goes from global memory through GroupNorm + SiLU straight into the MFMA -- no activation ever touches LDS -- and
// out[p][co] = sum_tap Y[p + tap offset][tap*3 + co] gathers 27 words per pixel from the 66 KB LDS image of Y.  Lane (m, half) of a
// 32-pixel block owns channels 32 J + 16 half + 4 i + e (4 back-to-back 16-byte loads = its 64-byte half of a 128-byte line);
// the matching weight fragments sit in registers (16 per 32 channels: one float4 each, straight out of the MFMA-packed image
// of output tile 0; above 128 channels in two passes over the tile, each with half of them, the second adding into Y).  20 blocks per 16 x 32-pixel tile (1.25 x the pixels), 5 per wave.  Bound: HBM (the input, once).
constexpr int G_TW = 32, G_TH = 16, G_HW = G_TW + 2, G_HH = G_TH + 2, G_NPIX = G_HH * G_HW, G_NBLK = (G_NPIX + 31) / 32, G_NV = 27;
static_assert(G_NBLK % 4 == 0, "blocks are dealt to 4 waves");

template <int NJ4, int KSPL, bool SILU, int ABL = 0>      // NJ4 = Cin / 32; KSPL passes over the tile, each with 1 / KSPL of the channels' weights in registers
__global__ __launch_bounds__(256, 2) void conv_small_gemm_kernel(const ConvParams p) {
    __shared__ float yl[G_NPIX * G_NV];
    __shared__ __attribute__((aligned(16))) float ssl[2][NJ4 * 32];
    constexpr int NJH = NJ4 / KSPL;
    static_assert(NJH * KSPL == NJ4 && NJH <= 4, "NJH x 16 weight registers per lane");
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, m = lane & 31, half = lane >> 5;
    int bx = blockIdx.x;
    const int tx = bx % p.tiles_x;
    bx /= p.tiles_x;
    const int ty = bx % p.tiles_y;
    const int b = bx / p.tiles_y;
    const int oy0 = ty * G_TH, ox0 = tx * G_TW;
    const int cin = NJ4 * 32;

    for (int c = tid; c < cin; c += 256) {
        ssl[0][c] = p.gn ? p.gscale[(size_t)b * p.ctot + c] : 1.f;
        ssl[1][c] = p.gn ? p.gshift[(size_t)b * p.ctot + c] : 0.f;
    }
    __syncthreads();

    const int tap = m / 3, co = m - 3 * tap;
    const bool wok = m < G_NV && co < p.Cout;
    const float* __restrict__ img = p.src[0] + (size_t)b * p.Hin * p.Win * cin + 16 * half;
    // this lane's pixel of block (pass ks, round k): its address (padding lanes: a safe one, zeroed after the arithmetic) and validity
    auto pixel_of = [&](int ks, int k, bool& inimg) -> const float* {
        const int blk = wv + 4 * k;
        const int hp = (ABL & 2) ? wv * 32 + m : blk * 32 + m;      // (ABL 2: every block re-reads block wv's pixels -- cache hits)
        const int hy = hp / G_HW, hx = hp - hy * G_HW;
        const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
        inimg = hp < G_NPIX && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
        return img + (inimg ? ((size_t)iy * p.Win + ix) * cin : 0) + ks * NJH * 32;
    };
    // the loads run ONE WHOLE BLOCK ahead of the arithmetic (NJH x 64 bytes per lane in flight): x[J] is refilled for the next block
    // as soon as step J of this one has read it -- with the loads of a block issued at its start the kernel took memory time
    // PLUS matrix time (0.118 + 0.132 of 0.227 ms at configs[1], tools/session/gpu_r3ac.sh)
    f32x4 x[NJH][4];
    bool in_cur;
    const float* __restrict__ px = pixel_of(0, 0, in_cur);
#pragma unroll
    for (int J = 0; J < NJH; ++J)
#pragma unroll
        for (int i = 0; i < 4; ++i) x[J][i] = *reinterpret_cast<const f32x4*>(px + 32 * J + 4 * i);
#pragma unroll 1
    for (int ks = 0; ks < KSPL; ++ks) {
        // B fragments: lane (n = tap*3 + co, half) holds W[co][32 J + 16 half + 4 i + e][tap] = the float4 at (chunk J, tap,
        // s = 2 half + i/2, lane (i & 1) * 32 + co) of the packed image; columns 27..31 and absent output channels multiply zeros
        f32x4 wreg[NJH][4];
#pragma unroll
        for (int J = 0; J < NJH; ++J)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                wreg[J][i] = wok ? *reinterpret_cast<const f32x4*>(p.w + (size_t)(ks * NJH + J) * (9 * 1024) + (tap * 4 + 2 * half + (i >> 1)) * 256 + (i & 1) * 128 + co * 4)
                                 : f32x4{0.f, 0.f, 0.f, 0.f};
        const float* __restrict__ ssc = &ssl[0][ks * NJH * 32 + 16 * half];
        const float* __restrict__ ssh = &ssl[1][ks * NJH * 32 + 16 * half];
#pragma unroll 1
        for (int k = 0; k < G_NBLK / 4; ++k) {
            const int blk = wv + 4 * k;
            const bool inimg = in_cur;
            const bool last_k = k + 1 == G_NBLK / 4;
            const bool more = !(last_k && ks + 1 == KSPL);                      // wave-uniform
            bool in_next = false;
            const float* __restrict__ pn = more ? pixel_of(last_k ? ks + 1 : ks, last_k ? 0 : k + 1, in_next) : px;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            int fresh = 0;                                   // (opaque: scale / shift are re-read from LDS per block -- hoisted out of
            asm volatile("" : "+v"(fresh));                  //  the loop they would pin 32 registers per 32 channels)
            // step (J, i) = 4 channels per lane = 4 MFMAs.  The vector work of step t + 1 stands BETWEEN the MFMA groups of steps
            // t - 1 and t in program order (a wave issues in order: with all of a J's arithmetic ahead of its 16 MFMAs the matrix
            // pipe idled through it -- same-box A/B of whole steps: 22.24 -> 22.20 ms at configs[1])
            auto act = [&](int J, int i) {
                const f32x4 sc = *reinterpret_cast<const f32x4*>(ssc + fresh + 32 * J + 4 * i);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(ssh + fresh + 32 * J + 4 * i);
                f32x4 v = x[J][i];
                if constexpr (!(ABL & 4)) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = fmaf(v[e], sc[e], sh[e]);
                        if constexpr (SILU) v[e] = silu_f<false>(v[e]);
                        v[e] = inimg ? v[e] : 0.f;                  // zero padding of the ACTIVATED tensor (and NaN-free dummy reads)
                    }
                }
                return v;
            };
            f32x4 vc = act(0, 0);
#pragma unroll
            for (int J = 0; J < NJH; ++J) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4 vn = vc;
                    const int t = J * 4 + i + 1;
                    if (t < NJH * 4) vn = act(t >> 2, t & 3);
                    if (i == 3 && more) {                           // x[J] is free: refill it for the next block
#pragma unroll
                        for (int ii = 0; ii < 4; ++ii) x[J][ii] = *reinterpret_cast<const f32x4*>(pn + 32 * J + 4 * ii);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if constexpr (ABL & 1) acc[e] = fmaf(vc[e], wreg[J][i][e], acc[e]);      // (ABL 1: no MFMA)
                        else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vc[e], wreg[J][i][e], acc, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    vc = vn;
                }
            }
            in_cur = in_next;
            // lane (n, half), register r -> Y[pixel 32 blk + 8 (r / 4) + 4 half + (r & 3)][n]; later passes add (same lane, same word)
            if (m < G_NV) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int hpr = blk * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
                    if (hpr < G_NPIX) yl[hpr * G_NV + m] = ks ? yl[hpr * G_NV + m] + acc[r] : acc[r];
                }
            }
        }
    }
    __syncthreads();

#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = tid + 256 * i;
        const int oyl = q >> 5, oxl = q & 31;
        float sum[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float* __restrict__ y = &yl[((oyl + tap / 3) * G_HW + oxl + tap % 3) * G_NV + tap * 3];
#pragma unroll
            for (int co = 0; co < 3; ++co) sum[co] += y[co];
        }
        const int oy = oy0 + oyl, ox = ox0 + oxl;
        if (oy < p.Hout && ox < p.Wout) {
            const size_t pix = ((size_t)b * p.Hout + oy) * p.Wout + ox;
#pragma unroll
            for (int co = 0; co < 3; ++co) {
                if (co < p.Cout) {
                    float v = sum[co];
                    if (p.bias) v += p.bias[co];
                    if (p.temb) v += p.temb[(size_t)b * p.temb_ld + co];
                    if (p.residual) v += p.residual[pix * p.Cout + co];
                    p.out[pix * p.out_ld + co] = v;
                }
            }
        }
    }
}

template <int NJ4, int KSPL>
int gemm_launch(const ConvParams& p, hipStream_t stream) {
    dim3 grid(p.tiles_x * p.tiles_y * p.B, 1);
#ifdef CDX_TUNING
    if (const char* e = getenv("CDX_GEMM_ABL")) {      // timing ablations (WRONG RESULTS): 1 no MFMA, 2 cached input, 4 no GroupNorm / SiLU arithmetic
        switch (atoi(e)) {
            case 1: hipLaunchKernelGGL((conv_small_gemm_kernel<NJ4, KSPL, true, 1>), grid, dim3(256), 0, stream, p); return check_launch();
            case 2: hipLaunchKernelGGL((conv_small_gemm_kernel<NJ4, KSPL, true, 2>), grid, dim3(256), 0, stream, p); return check_launch();
            case 4: hipLaunchKernelGGL((conv_small_gemm_kernel<NJ4, KSPL, true, 4>), grid, dim3(256), 0, stream, p); return check_launch();
            case 6: hipLaunchKernelGGL((conv_small_gemm_kernel<NJ4, KSPL, true, 6>), grid, dim3(256), 0, stream, p); return check_launch();
            case 7: hipLaunchKernelGGL((conv_small_gemm_kernel<NJ4, KSPL, true, 7>), grid, dim3(256), 0, stream, p); return check_launch();
            case 5: hipLaunchKernelGGL((conv_small_gemm_kernel<NJ4, KSPL, true, 5>), grid, dim3(256), 0, stream, p); return check_launch();
            default: break;
        }
    }
#endif
    if (p.silu) hipLaunchKernelGGL((conv_small_gemm_kernel<NJ4, KSPL, true>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((conv_small_gemm_kernel<NJ4, KSPL, false>), grid, dim3(256), 0, stream, p);
    return check_launch();
}

}  // namespace

// (conv.hip small_gemm_ok: one source of 64 / 128 / 192 / 256 channels, no upsampling, cout <= 3; 16 x 32-pixel tiles)
int conv_dispatch_small_gemm(const ConvParams& p, hipStream_t stream) {
    if (p.ups || p.nchunks != p.nchunk0 || p.Cout > 3 || p.Hin != p.Hout || p.Win != p.Wout) return CDX_ENOTSUP;
    switch (p.csrc[0]) {
        case 64: return gemm_launch<2, 1>(p, stream);
        case 128: return gemm_launch<4, 1>(p, stream);
        case 192: return gemm_launch<6, 2>(p, stream);
        case 256: return gemm_launch<8, 2>(p, stream);
        default: return CDX_ENOTSUP;
    }
}

int conv_dispatch_small(const ConvParams& p, hipStream_t stream, bool valu_form) {
    dim3 grid(p.tiles_x * p.tiles_y * p.B, 1);
    if (valu_form) {
        if (p.Cout <= 3) hipLaunchKernelGGL(conv_small_kernel<3>, grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL(conv_small_kernel<4>, grid, dim3(256), 0, stream, p);
    } else {
        if (p.Cout <= 3) hipLaunchKernelGGL(conv_small_mfma_kernel<3>, grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL(conv_small_mfma_kernel<4>, grid, dim3(256), 0, stream, p);
    }
    return check_launch();
}

}  // namespace cdx
