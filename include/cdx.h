/*
 * cdx.h -- C ABI of libcdx.so: the MI355X (gfx950) kernels behind the reverse-diffusion
 * decode hot path (UNet forward + DDIM/DDPM update).
 *
 * Reference interface replaced: NONE EXISTS.  The reference snapshot (/root/reference) holds
 * only an empty README.md (0 bytes) and a 27-line .gitignore (.gitignore:1-27); it has no
 * operator / plugin / FFI interface for this path.  The boundary is the one BASELINE.json
 * north_star dictates ("Python model/sampler API ... calling hand-written HIP kernels through a
 * thin C-ABI extension") and SURVEY.md section 8(b) specifies.  Each entry point below names
 * the SURVEY.md section 8(a) row (S-rows / U-rows) it implements; a reference-side torch.nn.functional
 * call it stands in for is given where one is obvious.
 *
 * Conventions
 *   - Every op:  int cdx_<op>(const cdx_<op>_args*, void* workspace, size_t workspace_bytes,
 *                              cdx_stream_t stream);   size_t cdx_<op>_workspace(const args*);
 *   - Returns CDX_OK or a negative status; never throws, aborts, allocates, frees or syncs
 *     (safe under hipGraph capture).  All work is enqueued on `stream` (a hipStream_t).
 *   - All pointers are DEVICE pointers owned by the caller, 16-byte aligned.
 *   - Activations are dense NHWC float32: x[b][y][x][c]; channel counts are multiples of 4.
 *   - Stateless and re-entrant: no global mutable state.
 */
#ifndef CDX_H
#define CDX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CDX_ABI_VERSION 5

typedef void* cdx_stream_t; /* hipStream_t */

enum {
    CDX_OK = 0,
    CDX_EINVAL = -1,  /* bad shape / null / misaligned pointer */
    CDX_ENOSPC = -2,  /* workspace too small */
    CDX_ELAUNCH = -3, /* hipGetLastError() != hipSuccess after the launch */
    CDX_ENOTSUP = -4  /* valid request this build has no kernel for */
};

int cdx_abi_version(void);
const char* cdx_strerror(int status);

/* ------------------------------------------------------------------------------------------
 * U3/U4/U5/U8/U9: float32 convolution as implicit GEMM on the matrix pipe: v_mfma_f32_32x32x2_f32 (direct / Winograd
 * tiles), or -- when wpacked_split is given -- three v_mfma_f32_32x32x16_f16 per product on hi | lo split operands.
 * Stands in for F.conv2d(silu(group_norm(cat[src0, src1]))) + bias + temb[:, :, None, None]
 * + residual, with nearest-2x upsampling and channel concat fused into the tile gather.
 * ------------------------------------------------------------------------------------------ */
enum {
    CDX_CONV_UPSAMPLE2X = 1, /* sources are nearest-upsampled x2 before the conv (U5) */
    CDX_CONV_GN = 2,         /* apply x*gn_scale[b][c] + gn_shift[b][c] while staging (U2) */
    CDX_CONV_SILU = 4,       /* then x * sigmoid(x) (U2) */
    CDX_CONV_BF16 = 8,       /* cdx_conv_f16 only: the 16-bit tensors and weights are bfloat16 (v_mfma_f32_32x32x16_bf16) */
    CDX_CONV_GN_EXP = 16     /* cdx_conv_f32 with CDX_CONV_GN (ABI v5): gn_exp is STATED -- gn_scale / gn_shift were written with
                                out_exp = gn_exp = cdx_gn_act_exp(...) of this GroupNorm.  Only such a launch may take the split
                                tile: a GroupNorm-ed launch that does not state its exponent (the plain scale / shift pair, whose
                                values could leave the fp16 range for large |gamma|) runs on the f32-input MFMA kernels */
};

#define CDX_CONV_KC 32 /* input-channel chunk of the packed weight layout */
#define CDX_AMAX_WORDS 16 /* words per image of the amax arrays (src_amax0/1, amax_out, cdx_amax_f32) */

typedef struct cdx_conv_args {
    const float* src0;     /* [batch, hin, win, c0] */
    const float* src1;     /* [batch, hin, win, c1] or NULL: channel-concatenated after src0 */
    int32_t c0, c1;        /* multiples of 4; with two sources both multiples of CDX_CONV_KC */
    int32_t batch, hin, win;
    int32_t hout, wout;    /* stride 1: = (2x) hin,win; stride 2: = ceil(hin/2), ceil(win/2) */
    int32_t cout;
    int32_t ksize;         /* 1 or 3 (pad = ksize/2) */
    int32_t stride;        /* 1 or 2 */
    int32_t flags;         /* CDX_CONV_* */
    const float* wpacked;  /* cdx_conv_pack_weights_f32 output, uploaded */
    const float* bias;     /* [cout] or NULL */
    const float* gn_scale; /* [batch, c0+c1] (CDX_CONV_GN) from cdx_gn_stats_f32 */
    const float* gn_shift; /* [batch, c0+c1] */
    const float* temb;     /* [batch, temb_ld] or NULL: out += temb[b*temb_ld + co] */
    int32_t temb_ld;
    const float* residual; /* [batch, hout, wout, cout] or NULL: out += residual */
    float* out;            /* [batch, hout, wout, out_ld], channels [0, cout) written */
    int32_t out_ld;
    const float* wpacked_wino; /* NULL, or the cdx_conv_pack_weights_wino_f32 image of the same weights: lets the library
                              run 3x3 stride-1 layers (wout >= 32, cout >= 96) as Winograd F(2x2,3x3) -- same float32
                              result up to summation order, 2.25x fewer MFMAs */
    double* stats_out;     /* NULL, or [batch, cdx_conv_stats_slots(a), cout, 2]: per-slot (sum, sum of squares) of the
                              values stored to `out`, float64, for cdx_gn_finalize_f32 (GroupNorm of `out` without
                              re-reading it).  Every (slot, channel) entry is written by the launch (unused ones as 0):
                              the buffer needs no initialisation. */
    const uint16_t* wpacked_split; /* NULL, or the cdx_conv_pack_weights_split_f16 image of the same weights (fp16 hi | lo
                              planes, pre-scaled by a power of two): lets the library run layers >= 8 pixels wide
                              (stride 2: >= 16) on the FP16 matrix pipe with split operands (hi*hi + lo*hi + hi*lo,
                              float32 accumulation), 3 MFMAs of 32 cycles per 16 channels where the f32-input MFMA needs
                              8 of 64.  RANGE CONTRACT (ABI v4): the activations are staged as x 2^e with a power of two
                              e that places the tensor's maximum in [2^14, 2^15), so the result keeps float32-level
                              error relative to the OUTPUT scale at ANY input scale.  e comes from
                                * gn_exp       for CDX_CONV_GN | CDX_CONV_GN_EXP launches (static: the normalised tensor's bound), or
                                * src_amax0/1  for un-normalised launches (per image, from the producing launches);
                              an un-normalised launch WITHOUT src_amax0 (or without src_amax1 when c1 > 0), and a GroupNorm-ed
                              launch WITHOUT CDX_CONV_GN_EXP, never take the split tile: they run on the f32-input MFMA
                              kernels, which need no scaling.
                              NaN / +-Inf inputs propagate as in F.conv2d (non-finite outputs over their footprint). */
    float wsplit_unscale;  /* the packer's `unscale` output (2^-s, exact); > 0 whenever wpacked_split is set */
    int32_t gn_exp;        /* CDX_CONV_GN: gn_scale / gn_shift hold scale 2^gn_exp, shift 2^gn_exp (cdx_gn_*_args.out_exp of
                              the launch that wrote them; |gn_exp| <= 60); the kernels undo it exactly.  Read only with
                              CDX_CONV_GN_EXP (must be 0 without it). */
    const uint32_t* src_amax0; /* NULL, or device [batch][CDX_AMAX_WORDS]: float32 BIT PATTERNS whose maximum is an upper bound
                              of max |x| over image b of src0 -- the `amax_out` words of the launch that produced src0, or
                              cdx_amax_f32's output (64-byte aligned) */
    const uint32_t* src_amax1; /* the same for src1 (c1 > 0) */
    const uint16_t* wpacked_split_up; /* NULL, or the cdx_conv_pack_weights_split_up_f16 image of the same 3x3 weights: lets the
                              library run a CDX_CONV_UPSAMPLE2X layer (low-resolution rows >= 32 pixels, no residual) as FOUR
                              2x2 convolutions on the low-resolution source, one per output phase -- 2.25x fewer multiply-adds
                              than the 3x3 convolution on the upsampled tensor, the same sum up to the rounding of the merged taps */
    float wsplit_up_unscale[4]; /* the packer's four `unscale` outputs */
    int32_t stats_slots;   /* with stats_out: the slot count the buffer was sized for = cdx_conv_stats_slots(a) asked with EVERY
                              other field final (the range fields above decide the tile, the tile the slots); a launch
                              whose tile writes a different count returns CDX_EINVAL instead of overrunning the buffer */
    uint32_t* amax_out;    /* NULL, or device [batch][CDX_AMAX_WORDS]: every wave max-combines (atomic, unsigned compare = float
                              compare for non-negative values) the bit pattern of the largest |out[b]| it stores into ONE of
                              image b's words (NaNs skipped; +-Inf counts): max over the words = max |out[b]|.  Several words
                              per image because same-address atomics serialise in L2 (measured: +14 us on a 150 us launch
                              with one word).  The CALLER zeroes them before the first launch that writes them
                              (cdx_fill_u32).  Honoured by every tile shape. */
} cdx_conv_args;

int cdx_conv_f32(const cdx_conv_args* a, void* workspace, size_t workspace_bytes, cdx_stream_t stream);
size_t cdx_conv_f32_workspace(const cdx_conv_args* a);
/* HOST: number of partial-sum slots per image cdx_conv_f32 writes to stats_out for this launch (0 on bad args). */
int32_t cdx_conv_stats_slots(const cdx_conv_args* a);

/* Diagnostics / tuning (not needed by a drop-in caller): the tile shape cdx_conv_f32 would pick for `a`
 * (>= 0, one of CDX_TILE_*; negative = status), and a launch that forces a given shape (CDX_ENOTSUP if the
 * shape is not built for that ksize/stride).  Tests use these to cover every instantiation; bench.py uses
 * the first to name the dominant kernel. */
enum {
    CDX_TILE_128x128 = 0, /* 4 waves 1x4, 4 M-tiles each            */
    CDX_TILE_128x64 = 1,  /* 2x2, 2 M-tiles                         */
    CDX_TILE_128x32 = 2,  /* 4x1, 1 M-tile (cout <= 32)             */
    CDX_TILE_64x128 = 3,  /* stride 2: 1x4, 2 M-tiles               */
    CDX_TILE_64x64 = 4,   /* stride 2: 2x2, 1 M-tile                */
    CDX_TILE_S32x32 = 5,  /* low-resolution levels: 4 waves split K */
    CDX_TILE_S64x32 = 6,
    CDX_TILE_WINO = 7,    /* Winograd F(2x2,3x3), 128 pixels x 128 channels, needs wpacked_wino */
    CDX_TILE_SMALL = 8,   /* 3x3 stride 1, cout <= 4, width >= 32: 4x4x1-MFMA kernel, 256 pixels  */
    CDX_TILE_SMALL_VALU = 9, /* the same tile on the vector ALU (weights through the scalar cache)  */
    CDX_TILE_CIN8 = 10,   /* 3x3 stride 1, at most 8 input channels (conv_in): 128x128, first channel group only */
    CDX_TILE_SMALL_GEMM = 12, /* 3x3 stride 1, cout <= 3, ONE source of 64 / 128 / 192 / 256 channels, no upsampling, width >= 32: the taps as
                               27 GEMM columns of v_mfma_f32_32x32x2_f32 over the pixels of a 16 x 32 tile's halo, activations straight
                               from memory into the MFMA, the 9 shifted partial sums gathered from LDS: HBM-bound (conv_out)           */
    CDX_TILE_SPLIT = 11   /* v_mfma_f32_32x32x16_f16 with hi/lo split operands (needs wpacked_split and an activation exponent, see
                             cdx_conv_args): 128 px x 128 ch at wout >= 32, 64 x 128 at wout 16..31 and for stride 2 (wout >= 16),
                             64 px x 32 ch with the input chunks split over the waves at wout 8..15 */
};
int cdx_conv_select_tile(const cdx_conv_args* a);
int cdx_conv_f32_tile(const cdx_conv_args* a, int32_t tile, void* workspace, size_t workspace_bytes, cdx_stream_t stream);

/* HOST helper: number of floats of the packed weight image, and the packer.
 * w_oihw: host [cout][c0+c1][ksize][ksize] (torch layout).  Packed layout:
 *   [ntile = ceil(cout/32)][chunk][tap = ky*ksize+kx][s = 0..3][lane = 0..63][e = 0..3]
 *   = W[n = 32*ntile + (lane&31)][c = chunk_base + 8*s + 4*(lane>>5) + e][ky][kx]
 * chunks: ceil(c0/32) of src0 then ceil(c1/32) of src1, zero-filled past each source's end; the image ends
 * with a 16 KiB zero pad (the kernels' weight prefetch ring reads past the last fragment). */
size_t cdx_conv_packed_floats(int32_t c0, int32_t c1, int32_t cout, int32_t ksize);
int cdx_conv_pack_weights_f32(const float* w_oihw, int32_t c0, int32_t c1, int32_t cout,
                              int32_t ksize, float* packed);
/* HOST: Winograd image of a 3x3 weight: U = G g G^T (float64, rounded once), fragment-ordered
 *   [ntile][chunk][s = 0..3][e = 0..3][xiq = 0..3][lane = 0..63][j = 0..3]
 *   = U[xi = 4*xiq + j][n = 32*ntile + (lane&31)][c = chunk_base + 8*s + 4*(lane>>5) + e],  + 16 KiB zero pad. */
size_t cdx_conv_wino_packed_floats(int32_t c0, int32_t c1, int32_t cout);
int cdx_conv_pack_weights_wino_f32(const float* w_oihw, int32_t c0, int32_t c1, int32_t cout, float* packed);
/* HOST: split image of a float32 weight for CDX_TILE_SPLIT: w' = w 2^s with s chosen so that max |w'| lies in [2^13, 2^14);
 * hi = fp16(w'), lo = fp16(w' - hi); *unscale = 2^-s (pass it as wsplit_unscale).  Layout (binary16 bit patterns):
 *   [ntile][chunk][tap][j = 0..1][plane = hi, lo][lane = 0..63][k = 0..7]
 *   = plane(W'[n = 32*ntile + (lane&31)][c = chunk_base + 16*j + 8*(lane>>5) + k][ky][kx]),  + 16 KiB zero pad.
 * Non-finite weights are rejected (CDX_EINVAL). */
size_t cdx_conv_split_packed_halves(int32_t c0, int32_t c1, int32_t cout, int32_t ksize);
/* HOST: phase images of a 3x3 weight for CDX_CONV_UPSAMPLE2X layers: phase (dy, dx), taps (ty, tx) of a 2x2 kernel,
 *   W2[2 dy + dx][n][c][ty][tx] = sum_{ky in K(dy,ty)} sum_{kx in K(dx,tx)} W[n][c][ky][kx],  K(0,0) = {0}, K(0,1) = {1,2}, K(1,0) = {0,1}, K(1,1) = {2}
 * (float64 sums, rounded once), each packed like a ksize-2 split image ([ntile][chunk][tap = 2 ty + tx][j][plane][lane][k]) with its
 * own power-of-two scale; unscale4[phase] = 2^-s.  The four images follow each other (cdx_conv_split_up_packed_halves / 4 halves each). */
size_t cdx_conv_split_up_packed_halves(int32_t c0, int32_t c1, int32_t cout);
int cdx_conv_pack_weights_split_up_f16(const float* w_oihw, int32_t c0, int32_t c1, int32_t cout, uint16_t* packed, float* unscale4);
int cdx_conv_pack_weights_split_f16(const float* w_oihw, int32_t c0, int32_t c1, int32_t cout, int32_t ksize,
                                    uint16_t* packed, float* unscale);

/* ------------------------------------------------------------------------------------------
 * fp16-storage variants (BASELINE.json configs[4]; SURVEY.md 8a "_f16 variants"): activations and weights in IEEE
 * binary16, products on v_mfma_f32_32x32x16_f16, float32 accumulation; GroupNorm scale/shift, bias, temb and the
 * GroupNorm partial sums stay float32 / float64.  Sources may be float32 (the sampler's x_t buffer) and the output
 * float32 (the eps buffer).  Channel counts: multiples of 8 for fp16 tensors, of 4 for fp32 sources.
 * ------------------------------------------------------------------------------------------ */
typedef uint16_t cdx_half; /* IEEE binary16 bit pattern */

typedef struct cdx_conv_f16_args {
    const void* src0;      /* [batch, hin, win, c0] fp16 (or fp32 if src_is_f32) */
    const void* src1;      /* second concat source or NULL */
    int32_t c0, c1;
    int32_t src_is_f32;
    int32_t batch, hin, win, hout, wout, cout, ksize, stride, flags; /* as cdx_conv_args */
    const cdx_half* wpacked; /* cdx_conv_pack_weights_f16 */
    const float* bias;
    const float* gn_scale;
    const float* gn_shift;
    const float* temb;
    int32_t temb_ld;
    const cdx_half* residual; /* [batch, hout, wout, cout] or NULL */
    void* out;                /* fp16 (or fp32 if out_is_f32) [batch, hout, wout, out_ld] */
    int32_t out_is_f32;
    int32_t out_ld;
    double* stats_out;        /* NULL or [batch, cdx_conv_f16_stats_slots(a), cout, 2] */
} cdx_conv_f16_args;

int cdx_conv_f16(const cdx_conv_f16_args* a, void* workspace, size_t workspace_bytes, cdx_stream_t stream);
size_t cdx_conv_f16_workspace(const cdx_conv_f16_args* a);
int32_t cdx_conv_f16_stats_slots(const cdx_conv_f16_args* a);
/* HOST: fp16 fragment image of float32 OIHW weights (round-to-nearest-even):
 *   [ntile][chunk][tap][j = 0..1][lane = 0..63][e = 0..7] = W[n = 32*ntile + (lane&31)][c = chunk_base + 16*j + 8*(lane>>5) + e][tap]
 * + 16 KiB zero pad.  Sizes are in halves. */
size_t cdx_conv_f16_packed_halves(int32_t c0, int32_t c1, int32_t cout, int32_t ksize);
int cdx_conv_pack_weights_f16(const float* w_oihw, int32_t c0, int32_t c1, int32_t cout, int32_t ksize, cdx_half* packed);
/* the same layout in bfloat16 (round to nearest even) for launches with CDX_CONV_BF16 */
int cdx_conv_pack_weights_bf16(const float* w_oihw, int32_t c0, int32_t c1, int32_t cout, int32_t ksize, uint16_t* packed);

/* ------------------------------------------------------------------------------------------
 * U2: GroupNorm statistics of cat[src0, src1] -> per-(batch, channel) scale / shift
 *   scale[b][c] = rstd[b][g(c)] * gamma[c];  shift[b][c] = beta[c] - mean[b][g(c)] * scale[b][c]
 * (F.group_norm's own two-step form).  Sums are carried in float64, fixed order (deterministic).
 * ------------------------------------------------------------------------------------------ */
typedef struct cdx_gn_stats_args {
    const float* src0;  /* [batch, hw, c0] */
    const float* src1;  /* [batch, hw, c1] or NULL */
    int32_t c0, c1;
    int32_t batch, hw;
    int32_t groups;
    float eps;
    const float* gamma; /* [c0+c1] */
    const float* beta;  /* [c0+c1] */
    float* scale;       /* [batch, c0+c1] */
    float* shift;       /* [batch, c0+c1] */
    float* mean;        /* [batch, groups] or NULL (diagnostic) */
    float* rstd;        /* [batch, groups] or NULL */
    int32_t out_exp;    /* scale and shift are written multiplied by 2^out_exp (exact; |out_exp| <= 60): pass the same
                           value as cdx_conv_args.gn_exp of the convolution that consumes them.  cdx_gn_act_exp() gives
                           the value that keeps the normalised tensor inside the fp16 range of the split tile. */
} cdx_gn_stats_args;

int cdx_gn_stats_f32(const cdx_gn_stats_args* a, void* workspace, size_t workspace_bytes, cdx_stream_t stream);
size_t cdx_gn_stats_f32_workspace(const cdx_gn_stats_args* a);

/* U2, fused form: the same scale / shift from partial sums that the producing convolutions left in their
 * stats_out buffers (one or two sources = channel concat).  Sums the slots in fixed order in float64. */
typedef struct cdx_gn_finalize_args {
    const double* part0; int32_t slots0; int32_t c0; /* [batch, slots0, c0, 2] */
    const double* part1; int32_t slots1; int32_t c1; /* [batch, slots1, c1, 2] or NULL / 0 / 0 */
    int32_t batch, hw;                               /* hw = pixels per image of the normalised tensor */
    int32_t groups;
    float eps;
    const float* gamma; /* [c0+c1] */
    const float* beta;
    float* scale;       /* [batch, c0+c1] */
    float* shift;
    float* mean;        /* [batch, groups] or NULL */
    float* rstd;
    int32_t out_exp;    /* as cdx_gn_stats_args.out_exp */
} cdx_gn_finalize_args;

int cdx_gn_finalize_f32(const cdx_gn_finalize_args* a, void* workspace, size_t workspace_bytes, cdx_stream_t stream);
size_t cdx_gn_finalize_f32_workspace(const cdx_gn_finalize_args* a);

/* HOST: the activation exponent of a GroupNorm-ed convolution input (gn_exp / out_exp above).  After GroupNorm every value
 * obeys |x^| <= sqrt(n - 1), n = elements per (image, group) = (channels / groups) * hw, so
 * |gamma_c x^ + beta_c| <= bound = max_c |gamma_c| sqrt(n) + max_c |beta_c| (SiLU only shrinks magnitudes).  Returns the
 * largest e with bound * 2^e < 2^15, clamped to [-60, 60] (0 for a zero or non-finite bound): staged values can then never
 * leave the fp16 range, and a unit-variance tensor sits 2^e above the range where the split's low part loses bits. */
int32_t cdx_gn_act_exp(const float* gamma_host, const float* beta_host, int32_t channels, int32_t groups, int32_t hw);

/* ------------------------------------------------------------------------------------------
 * U6/U7: multi-head attention core  out = softmax(q k^T * scale) v  on fp32 MFMA.
 * Token-major operands: q[b][i][h*head_dim + d] with leading dimension q_ld, etc.
 * ------------------------------------------------------------------------------------------ */
typedef struct cdx_attn_args {
    const float* q; int32_t q_ld;
    const float* k; int32_t k_ld;
    const float* v; int32_t v_ld;
    int32_t batch, nq, nk, heads, head_dim; /* head_dim == 64 */
    float scale;
    float* out; int32_t out_ld;             /* [batch, nq, out_ld] */
} cdx_attn_args;

int cdx_attn_f32(const cdx_attn_args* a, void* workspace, size_t workspace_bytes, cdx_stream_t stream);
size_t cdx_attn_f32_workspace(const cdx_attn_args* a);
/* attention core with fp16 q/k/v/out (float32 softmax and accumulation); same arguments as cdx_attn_f32 */
int cdx_attn_f16(const cdx_attn_args* a, void* workspace, size_t workspace_bytes, cdx_stream_t stream);
size_t cdx_attn_f16_workspace(const cdx_attn_args* a);
/* q / k / v / out in bfloat16 (v_mfma_f32_32x32x16_bf16) */
int cdx_attn_bf16(const cdx_attn_args* a, void* workspace, size_t workspace_bytes, cdx_stream_t stream);
size_t cdx_attn_bf16_workspace(const cdx_attn_args* a);

/* ------------------------------------------------------------------------------------------
 * U1: small-M linear  out[m][n] = sum_k act(x[m][k]) * w[n][k] + bias[n]   (F.linear)
 * ------------------------------------------------------------------------------------------ */
enum { CDX_LINEAR_SILU_IN = 1 };

typedef struct cdx_linear_args {
    const float* x; int32_t x_ld; /* [m, x_ld] */
    const float* w;               /* [n, k] row-major (torch layout) */
    const float* bias;            /* [n] or NULL */
    int32_t m, n, k;              /* k multiple of 4; any m, any k */
    int32_t flags;
    float* out; int32_t out_ld;
} cdx_linear_args;

int cdx_linear_f32(const cdx_linear_args* a, void* workspace, size_t workspace_bytes, cdx_stream_t stream);
size_t cdx_linear_f32_workspace(const cdx_linear_args* a);

/* U1: sinusoidal timestep embedding  out[b] = [sin(t_b f_k), cos(t_b f_k)], f_k = exp(-ln(1e4) k/(half-1)),
 * evaluated in float64 and rounded to float32. */
typedef struct cdx_timestep_embedding_args {
    const int32_t* t; /* [batch] */
    int32_t batch, dim;
    float* out;       /* [batch, dim] */
} cdx_timestep_embedding_args;

int cdx_timestep_embedding_f32(const cdx_timestep_embedding_args* a, void* workspace, size_t workspace_bytes, cdx_stream_t stream);
size_t cdx_timestep_embedding_f32_workspace(const cdx_timestep_embedding_args* a);

/* ------------------------------------------------------------------------------------------
 * S3/S4: one reverse-diffusion update on channels [0, channels) of an NHWC buffer, in place:
 *   x0h = clamp(ca*x + cb*eps, -1, 1) (clamp iff clip_x0);  x = cx*x + c0*x0h + ce*eps + sigma*z
 * z = N(0,1) from the counter generator, stream (seed, first_image + b, noise_stream), element
 * index (c*hw + pixel) (i.e. the NCHW flat index), only evaluated when sigma != 0.
 * ------------------------------------------------------------------------------------------ */
typedef struct cdx_diffusion_update_args {
    float* x; int32_t x_ld;           /* [batch, hw, x_ld] */
    const float* eps; int32_t eps_ld; /* [batch, hw, eps_ld] */
    int32_t batch, hw, channels;
    float ca, cb, cx, c0, ce, sigma;
    int32_t clip_x0;
    uint64_t seed; int64_t first_image; int32_t noise_stream;
} cdx_diffusion_update_args;

int cdx_diffusion_update_f32(const cdx_diffusion_update_args* a, void* workspace, size_t workspace_bytes, cdx_stream_t stream);
size_t cdx_diffusion_update_f32_workspace(const cdx_diffusion_update_args* a);

/* S4: x[b][p][c] = N(0,1), c < channels, same generator / indexing as above (x_T uses stream 1). */
typedef struct cdx_gauss_fill_args {
    float* x; int32_t x_ld;
    int32_t batch, hw, channels;
    uint64_t seed; int64_t first_image; int32_t noise_stream;
} cdx_gauss_fill_args;

int cdx_gauss_fill_f32(const cdx_gauss_fill_args* a, void* workspace, size_t workspace_bytes, cdx_stream_t stream);
size_t cdx_gauss_fill_f32_workspace(const cdx_gauss_fill_args* a);

/* U8: write nearest-resized cond (NCHW [batch, cc, hc, wc]) into channels [c_off, c_off+cc) of an NHWC
 * buffer [batch, h, w, x_ld] and zero channels [c_off+cc, x_ld)  (F.interpolate(mode="nearest") + cat). */
typedef struct cdx_cond_embed_args {
    const float* cond; int32_t cc, hc, wc;
    float* x; int32_t x_ld; int32_t c_off;
    int32_t batch, h, w;
} cdx_cond_embed_args;

int cdx_cond_embed_f32(const cdx_cond_embed_args* a, void* workspace, size_t workspace_bytes, cdx_stream_t stream);
size_t cdx_cond_embed_f32_workspace(const cdx_cond_embed_args* a);

/* S2 (exit): out NCHW [batch, channels, hw] = clamp(x[b][p][c], lo, hi). */
typedef struct cdx_export_image_args {
    const float* x; int32_t x_ld;
    int32_t batch, hw, channels;
    float lo, hi;
    float* out;
} cdx_export_image_args;

int cdx_export_image_f32(const cdx_export_image_args* a, void* workspace, size_t workspace_bytes, cdx_stream_t stream);
size_t cdx_export_image_f32_workspace(const cdx_export_image_args* a);

/* S5: blend decoded tiles into the full image.  tiles NCHW [batch*ny*nx, channels, tile, tile] (image-major, then
 * tile row, then tile column); tile (iy, ix) covers rows [y0[iy], y0[iy]+tile), columns [x0[ix], x0[ix]+tile).
 * out[b][c][y][x] = sum_t w_t(y,x) tile_t / sum_t w_t(y,x) with separable weights: along each axis
 * w(u) = min(1, (u+0.5)/ov_lo) * min(1, (tile-u-0.5)/ov_hi), u = local coordinate, ov_lo / ov_hi = overlap with the
 * previous / next tile (1 = no ramp at an image border). */
typedef struct cdx_tile_blend_args {
    const float* tiles;
    int32_t batch, channels, tile;
    int32_t ny, nx;
    const int32_t* y0; /* [ny] device */
    const int32_t* x0; /* [nx] device */
    int32_t h, w;      /* full image size */
    float* out;        /* NCHW [batch, channels, h, w] */
} cdx_tile_blend_args;

int cdx_tile_blend_f32(const cdx_tile_blend_args* a, void* workspace, size_t workspace_bytes, cdx_stream_t stream);
size_t cdx_tile_blend_f32_workspace(const cdx_tile_blend_args* a);

/* ------------------------------------------------------------------------------------------
 * Range bookkeeping of the split tile (cdx_conv_args.src_amax0/1, amax_out) and debug checks.
 * ------------------------------------------------------------------------------------------ */
/* max_k out[b][k] = max(itself, bit pattern of max_{p < n, c < channels} |x[b][p][c]|)  (NaNs skipped; atomic max-combine
 * spread over the CDX_AMAX_WORDS words of image b: zero `out` first with cdx_fill_u32 unless the combination with earlier
 * producers is wanted).  For tensors no convolution
 * of this library produced: caller data (latent, context tokens), the sampler's x_t | cond buffer, attention outputs. */
typedef struct cdx_amax_args {
    const float* x; int32_t x_ld; /* [batch, n, x_ld] */
    int32_t batch, n, channels;   /* channels <= x_ld, multiples of 4 */
    uint32_t* out;                /* device [batch][CDX_AMAX_WORDS], 64-byte aligned */
} cdx_amax_args;

int cdx_amax_f32(const cdx_amax_args* a, void* workspace, size_t workspace_bytes, cdx_stream_t stream);
size_t cdx_amax_f32_workspace(const cdx_amax_args* a);

/* x[0 .. n) = value (32-bit words): zeroes the amax words / status words at the start of a forward pass. */
typedef struct cdx_fill_u32_args {
    uint32_t* x; int64_t n; uint32_t value;
} cdx_fill_u32_args;

int cdx_fill_u32(const cdx_fill_u32_args* a, void* workspace, size_t workspace_bytes, cdx_stream_t stream);
size_t cdx_fill_u32_workspace(const cdx_fill_u32_args* a);

/* Debug: *status |= 1 if any of x[b][p][c], c < channels, is NaN or +-Inf; |= 2 if any |x| exceeds `limit` (limit > 0).
 * The Python host runs it after every launch in debug mode (ops.conv(..., debug=True), Plan.run(debug=True)) and raises
 * on a non-zero word; the product path never launches it. */
typedef struct cdx_check_finite_args {
    const float* x; int32_t x_ld;
    int64_t rows;                 /* batch * pixels */
    int32_t channels;
    float limit;                  /* 0 = no magnitude check */
    int32_t* status;              /* device int32, caller-zeroed */
} cdx_check_finite_args;

int cdx_check_finite_f32(const cdx_check_finite_args* a, void* workspace, size_t workspace_bytes, cdx_stream_t stream);
size_t cdx_check_finite_f32_workspace(const cdx_check_finite_args* a);

/* ------------------------------------------------------------------------------------------
 * (f4) Bitstream side: rANS decode of the quantised latent (format build-defined, "CDXL" v1).
 * One stream per (image, latent channel); symbol s in [0, alphabet) stands for the latent value (s - qmax) * step;
 * ONE static frequency table (sum = 2^prob_bits, every entry >= 1).  Stream layout (16-bit words): final encoder state
 * (high word, low word), then the renormalisation words in DECODE order.  32-bit state, lower bound 2^16:
 *   slot = x & (2^pb - 1); s: cum[s] <= slot < cum[s] + freq[s]; x = freq[s] (x >> pb) + slot - cum[s];
 *   if (x < 2^16) x = (x << 16) | next word.
 * out[stream][i] = dequantised value (float32), symbols (optional) = s - qmax as int16.  *status (optional, device,
 * caller-zeroed) gets bit 0 set if any stream is malformed (runs past stream_len, or does not end in state 2^16 with
 * every word consumed); reads never leave [stream_off, stream_off + stream_len).
 * ------------------------------------------------------------------------------------------ */
typedef struct cdx_rans_decode_args {
    const uint16_t* words;      /* device: payload words of all streams */
    const uint32_t* stream_off; /* device [nstreams]: first word of each stream */
    const uint32_t* stream_len; /* device [nstreams]: words in each stream */
    const uint16_t* freq;       /* device [alphabet] */
    int32_t nstreams, nsym;     /* nsym symbols per stream (= latent h * w) */
    int32_t alphabet, prob_bits, qmax; /* alphabet == 2 qmax + 1 <= 2^prob_bits, prob_bits <= 12 */
    float step;                 /* dequantisation step */
    float* out;                 /* device [nstreams, nsym] */
    int16_t* symbols;           /* device [nstreams, nsym] or NULL */
    int32_t* status;            /* device int32 or NULL */
} cdx_rans_decode_args;

int cdx_rans_decode_i16(const cdx_rans_decode_args* a, void* workspace, size_t workspace_bytes, cdx_stream_t stream);
size_t cdx_rans_decode_i16_workspace(const cdx_rans_decode_args* a);

/* ------------------------------------------------------------------------------------------
 * (f4) Image side: 8-bit export of the decoded tensor and the two quality metrics of the codec literature.
 * ------------------------------------------------------------------------------------------ */
/* out[b][p][c] = floor((clamp(x[b][p][c], lo, hi) - lo) * 255 / (hi - lo) + 0.5) as uint8, c < channels: interleaved rows, the
 * scanline order of PPM (P6) and PNG; x is the NHWC state buffer of the sampler (x_ld floats per pixel). */
typedef struct cdx_export_u8_args {
    const float* x; int32_t x_ld;
    int32_t batch, hw, channels;
    float lo, hi;
    uint8_t* out;                 /* device [batch, hw, channels] */
} cdx_export_u8_args;

int cdx_export_u8(const cdx_export_u8_args* a, void* workspace, size_t workspace_bytes, cdx_stream_t stream);
size_t cdx_export_u8_workspace(const cdx_export_u8_args* a);

/* out[i] = 10 log10(range^2 / mean((a[i] - b[i])^2)) over the n elements of image i (float64 sums, fixed order; +inf if equal). */
typedef struct cdx_psnr_args {
    const float* a; const float* b; /* [batch, n] */
    int32_t batch; int64_t n;
    float range;                  /* peak-to-peak of the data (2 for [-1, 1]) */
    float* out;                   /* device [batch] */
} cdx_psnr_args;

int cdx_psnr_f32(const cdx_psnr_args* a, void* workspace, size_t workspace_bytes, cdx_stream_t stream);
size_t cdx_psnr_f32_workspace(const cdx_psnr_args* a);

/* Multi-scale SSIM (Wang, Simoncelli, Bovik 2003) per image of NCHW tensors: 5 scales, 11 x 11 Gaussian window (sigma 1.5, valid
 * support), 2 x 2 average pooling between scales, C1 = (0.01 range)^2, C2 = (0.03 range)^2,
 *   out[b] = prod_{j<4} mean(cs_j)^w_j * mean(ssim_4)^w_4,  w = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333),
 * means over the channels and valid pixels of image b (negative means clamp to 0).  h, w >= 176.  per_scale (optional)
 * receives the five means. */
typedef struct cdx_msssim_args {
    const float* x; const float* y; /* [batch, channels, h, w] */
    int32_t batch, channels, h, w;
    float range;
    float* out;                   /* device [batch] */
    float* per_scale;             /* device [batch, 5] or NULL */
} cdx_msssim_args;

int cdx_msssim_f32(const cdx_msssim_args* a, void* workspace, size_t workspace_bytes, cdx_stream_t stream);
size_t cdx_msssim_f32_workspace(const cdx_msssim_args* a);

/* Diagnostics: monotonically counts kernel launches made through this library (relaxed atomic;
 * the only process-global the library keeps, used by tests to prove the HIP path ran). */
uint64_t cdx_launch_count(void);

#ifdef __cplusplus
}
#endif
#endif /* CDX_H */
