// U6/U7: multi-head attention core on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32), head_dim 64.
//
// One workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 query rows.
// Keys / values stream through LDS in blocks of 32 (shared by the 4 waves).  Two passes over the
// keys -- pass 1: row maxima, pass 2: exp / row sums / P V -- instead of online rescaling: attention
// is < 1.5 % of the UNet's FLOPs, sequences are <= 1024 tokens, and the two-pass form has exactly the
// numerics of softmax-then-matmul (max subtracted once, one normalisation at the end).
//
// The score tile is computed TRANSPOSED (S^T = K Q^T: keys on the accumulator rows, queries on the
// lanes), so (a) row maxima / sums are in-register reductions plus one lane-half exchange, and (b) the
// probability tile already has the MFMA A-operand layout of the P V product -- it never leaves
// registers (cdna guide, "An accumulator tile as the next MFMA's operand").
// No reference file exists to cite (reference snapshot is empty); semantics = softmax(q k^T * scale) v.
#include <hip/hip_fp16.h>

#include "common.h"

using namespace cdx;

namespace {

constexpr int HD = 64;
constexpr int KSTR = HD + 4;   // padded K row: conflict-free ds_read_b128 across 16 keys

// attn_kernel<float>: float32 storage and arithmetic (the 16-bit storage types use attn16_kernel below)
template <typename T>
__device__ __forceinline__ f32x4 ld4(const T* p) {
    if constexpr (sizeof(T) == 4) return *reinterpret_cast<const f32x4*>(p);
    else {
        using h4 = __attribute__((ext_vector_type(4))) _Float16;
        const h4 v = *reinterpret_cast<const h4*>(p);
        return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
}

template <typename T>
struct AttnParams {
    const T *q, *k, *v;
    int q_ld, k_ld, v_ld;
    int nq, nk;
    float scale;
    T* out;
    int out_ld;
};

template <typename T>
__global__ __launch_bounds__(256) void attn_kernel(const AttnParams<T> p) {
    __shared__ __attribute__((aligned(16))) float smem[32 * KSTR + 32 * HD];
    float* Ks = smem;
    float* Vs = smem + 32 * KSTR;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int qb = blockIdx.x * 128 + wave * 32;

    // Q^T as the B operand: lane (q = li, half lh) holds Q[q][32*lh + s], s = 0..31, pre-scaled.
    f32x4 qf[8];
    {
        const int row = min(qb + li, p.nq - 1);
        const T* src = p.q + ((size_t)b * p.nq + row) * p.q_ld + head * HD + lh * 32;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            qf[i] = ld4(src + i * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) qf[i][e] *= p.scale;
        }
    }

    const int skey = tid >> 3, sd = (tid & 7) * 8;   // staging: key row, first of 8 floats
    auto stage = [&](int kb, bool with_v) {
        const int krow = min(kb + skey, p.nk - 1);
        const T* ks = p.k + ((size_t)b * p.nk + krow) * p.k_ld + head * HD + sd;
        const f32x4 k0 = ld4(ks), k1 = ld4(ks + 4);
        *reinterpret_cast<f32x4*>(Ks + skey * KSTR + sd) = k0;
        *reinterpret_cast<f32x4*>(Ks + skey * KSTR + sd + 4) = k1;
        if (with_v) {
            const T* vs = p.v + ((size_t)b * p.nk + krow) * p.v_ld + head * HD + sd;
            const f32x4 v0 = ld4(vs), v1 = ld4(vs + 4);
            *reinterpret_cast<f32x4*>(Vs + skey * HD + sd) = v0;
            *reinterpret_cast<f32x4*>(Vs + skey * HD + sd + 4) = v1;
        }
    };

    // S^T tile: rows = keys (A = K), cols = queries (B = Q^T).
    auto scores = [&]() {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + li * KSTR + lh * 32 + i * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[i][e], acc, 0, 0, 0);
        }
        return acc;
    };

    // pass 1: row maxima
    float m = -INFINITY;
    for (int kb = 0; kb < p.nk; kb += 32) {
        stage(kb, false);
        __syncthreads();
        const f32x16 s = scores();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (key < p.nk) m = fmaxf(m, s[r]);
        }
        __syncthreads();
    }
    m = fmaxf(m, __shfl_xor(m, 32));

    // pass 2: P = exp(S - m), row sums, O += P V
    float l = 0.f;
    f32x16 oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    for (int kb = 0; kb < p.nk; kb += 32) {
        stage(kb, true);
        __syncthreads();
        f32x16 s = scores();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float e = key < p.nk ? expf(s[r] - m) : 0.f;
            s[r] = e;
            l += e;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int krow = (r & 3) + 8 * (r >> 2) + 4 * lh;   // the key this lane half contributes at step r
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[r], Vs[krow * HD + dt * 32 + li], oacc[dt], 0, 0, 0);
        }
        __syncthreads();
    }
    l += __shfl_xor(l, 32);
    const float linv = 1.0f / l;

    // O tile: rows = queries (register index), cols = head-dim (lane).
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int qrow = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float inv = __shfl(linv, qrow);
        if (qb + qrow < p.nq) {
            T* o = p.out + ((size_t)b * p.nq + qb + qrow) * p.out_ld + head * HD + li;
            o[0] = (T)(oacc[0][r] * inv);
            o[32] = (T)(oacc[1][r] * inv);
        }
    }
}


// ------------------------------------------------------------------------------------------------------------
// 16-bit storage (fp16 / bf16): BOTH contractions on the 16-bit matrix pipe (v_mfma_f32_32x32x16_f16 / _bf16, float32
// accumulation), softmax in float32 (BASELINE.json configs[4]: "fp16 UNet on CDNA4 MFMA attention").  Same two-pass
// structure and the same transposed score tile as above; what changes is the operand plumbing of a K = 16 MFMA:
//   * S^T = K Q^T: lane (key / query = lane & 31, half = lane >> 5) supplies 8 consecutive head-dim values per step
//     (4 steps cover head_dim 64) -- ONE 16-byte LDS read (K, row stride 144 B: conflict-free) / register fragment (Q);
//   * P V: the score accumulator holds, per lane (query = lane & 31), the 16 keys 8 (r >> 2) + 4 (lane >> 5) + (r & 3).
//     Registers 0..7 / 8..15 of the two lane halves together are keys 0..15 / 16..31, so the probabilities, rounded to
//     the storage type, are directly the A operands of two K = 16 MFMAs -- provided V is fed in the same key order:
//     V is staged TRANSPOSED and key-permuted, Vt[d][m][half][e] = V[key = 16 m + 8 (e >> 2) + 4 half + (e & 3)][d],
//     so that a lane's B operand (column d, half) is one 16-byte read.
template <typename H>
struct mfma16;
template <>
struct mfma16<_Float16> {
    using v8 = __attribute__((ext_vector_type(8))) _Float16;
    static __device__ __forceinline__ f32x16 run(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <>
struct mfma16<__bf16> {
    using v8 = __attribute__((ext_vector_type(8))) __bf16;
    static __device__ __forceinline__ f32x16 run(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};

template <typename H>
__global__ __launch_bounds__(256) void attn16_kernel(const AttnParams<H> p) {
    using v8 = typename mfma16<H>::v8;
    constexpr int KSH = HD + 8;                                     // K row stride in halves: 144 B
    constexpr int VSH = 32 + 8;                                     // Vt row (one d): 32 permuted keys + pad = 80 B
    __shared__ __attribute__((aligned(16))) H Ks[32 * KSH];
    __shared__ __attribute__((aligned(16))) H Vt[HD * VSH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int qb = blockIdx.x * 128 + wave * 32;

    // Q^T as the B operand: lane (q = li, half lh), step i: Q[q][16 i + 8 lh .. + 7]
    v8 qf[4];
    {
        const int row = min(qb + li, p.nq - 1);
        const H* src = p.q + ((size_t)b * p.nq + row) * p.q_ld + head * HD + lh * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) qf[i] = *reinterpret_cast<const v8*>(src + i * 16);
    }

    const int skey = tid >> 3, sd = (tid & 7) * 8;                  // staging: key row, first of 8 head-dim values
    const int vpos = (skey >> 4) * 16 + ((skey >> 2) & 1) * 8 + ((skey >> 3) & 1) * 4 + (skey & 3);   // [m][half][e] of this key
    auto stage = [&](int kb, bool with_v) {
        const int krow = min(kb + skey, p.nk - 1);
        *reinterpret_cast<v8*>(&Ks[skey * KSH + sd]) =
            *reinterpret_cast<const v8*>(p.k + ((size_t)b * p.nk + krow) * p.k_ld + head * HD + sd);
        if (with_v) {
            const v8 v = *reinterpret_cast<const v8*>(p.v + ((size_t)b * p.nk + krow) * p.v_ld + head * HD + sd);
#pragma unroll
            for (int e = 0; e < 8; ++e) Vt[(sd + e) * VSH + vpos] = v[e];
        }
    };
    auto scores = [&]() {                                           // S^T tile (keys x queries), scaled, float32
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            acc = mfma16<H>::run(*reinterpret_cast<const v8*>(&Ks[li * KSH + i * 16 + lh * 8]), qf[i], acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] *= p.scale;
        return acc;
    };

    float m = -INFINITY;
    for (int kb = 0; kb < p.nk; kb += 32) {
        stage(kb, false);
        __syncthreads();
        const f32x16 s = scores();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (key < p.nk) m = fmaxf(m, s[r]);
        }
        __syncthreads();
    }
    m = fmaxf(m, __shfl_xor(m, 32));

    float l = 0.f;
    f32x16 oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    for (int kb = 0; kb < p.nk; kb += 32) {
        stage(kb, true);
        __syncthreads();
        const f32x16 s = scores();
        v8 pf[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float e = key < p.nk ? expf(s[r] - m) : 0.f;
            const H eh = (H)e;
            pf[r >> 3][r & 7] = eh;
            l += (float)eh;                                        // the row sum of what is actually multiplied
        }
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
                oacc[dt] = mfma16<H>::run(pf[mm], *reinterpret_cast<const v8*>(&Vt[(dt * 32 + li) * VSH + mm * 16 + lh * 8]), oacc[dt]);
        __syncthreads();
    }
    l += __shfl_xor(l, 32);
    const float linv = 1.0f / l;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int qrow = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float inv = __shfl(linv, qrow);
        if (qb + qrow < p.nq) {
            H* o = p.out + ((size_t)b * p.nq + qb + qrow) * p.out_ld + head * HD + li;
            o[0] = (H)(oacc[0][r] * inv);
            o[32] = (H)(oacc[1][r] * inv);
        }
    }
}

template <typename T>
int attn_launch(const cdx_attn_args* a, cdx_stream_t stream) {
    CDX_REQUIRE(a && a->q && a->k && a->v && a->out);
    CDX_REQUIRE(a->batch > 0 && a->batch <= 65535 && a->heads > 0 && a->heads <= 65535 && a->nq > 0 && a->nk > 0);
    if (a->head_dim != HD) return CDX_ENOTSUP;
    const int c = a->heads * HD;
    CDX_REQUIRE(a->q_ld >= c && a->k_ld >= c && a->v_ld >= c && a->out_ld >= c);
    CDX_REQUIRE((a->q_ld % 4) == 0 && (a->k_ld % 4) == 0 && (a->v_ld % 4) == 0);
    CDX_REQUIRE(aligned16(a->q) && ((uintptr_t)a->k % 8) == 0 && ((uintptr_t)a->v % 8) == 0);
    AttnParams<T> p{reinterpret_cast<const T*>(a->q), reinterpret_cast<const T*>(a->k), reinterpret_cast<const T*>(a->v),
                    a->q_ld, a->k_ld, a->v_ld, a->nq, a->nk, a->scale, reinterpret_cast<T*>(a->out), a->out_ld};
    const dim3 grid((a->nq + 127) / 128, a->heads, a->batch);
    if constexpr (sizeof(T) == 4) {
        hipLaunchKernelGGL(attn_kernel<T>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p);
    } else {
        CDX_REQUIRE((a->q_ld % 8) == 0 && (a->k_ld % 8) == 0 && (a->v_ld % 8) == 0);      // 16-byte operand reads
        CDX_REQUIRE(aligned16(a->k) && aligned16(a->v));
        hipLaunchKernelGGL(attn16_kernel<T>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p);
    }
    return check_launch();
}

}  // namespace

extern "C" size_t cdx_attn_f32_workspace(const cdx_attn_args*) { return 0; }
extern "C" int cdx_attn_f32(const cdx_attn_args* a, void*, size_t, cdx_stream_t stream) { return attn_launch<float>(a, stream); }
extern "C" size_t cdx_attn_f16_workspace(const cdx_attn_args*) { return 0; }
extern "C" int cdx_attn_f16(const cdx_attn_args* a, void*, size_t, cdx_stream_t stream) { return attn_launch<_Float16>(a, stream); }
extern "C" size_t cdx_attn_bf16_workspace(const cdx_attn_args*) { return 0; }
extern "C" int cdx_attn_bf16(const cdx_attn_args* a, void*, size_t, cdx_stream_t stream) { return attn_launch<__bf16>(a, stream); }
