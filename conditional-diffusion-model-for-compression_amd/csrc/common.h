// Shared helpers for libcdx.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cdx.h"

namespace cdx {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

void count_launch();

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int check_launch() {
    count_launch();
    return hipGetLastError() == hipSuccess ? CDX_OK : CDX_ELAUNCH;
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// splitmix64 finaliser and the stream key of the counter generator (rng.py / SURVEY.md S4).
__host__ __device__ inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
constexpr uint64_t kGold = 0x9E3779B97F4A7C15ull;
__host__ __device__ inline uint64_t stream_key(uint64_t seed, uint64_t a, uint64_t b) {
    uint64_t k = mix64(seed + kGold);
    k = mix64(k ^ (a + kGold));
    return mix64(k ^ (b + kGold));
}

// 4x4 transpose across a lane quad (lanes 4k..4k+3), in registers: on entry lane q holds x[i] = value(row i, col q);
// on exit lane q holds x[j] = value(row q, col j).  Two butterfly stages (xor 1, xor 2 -> DPP quad permutes).
// Used by the convolution epilogues: accumulator layout (lane = channel, register = pixel) -> (lane = pixel, 4
// consecutive channels in registers), so outputs / residuals move as 8- or 16-byte accesses instead of 2- or 4-byte.
// (the lane exchanges are DPP quad permutes -- one VALU move each, no LDS round trip: __shfl_xor compiles to
// ds_bpermute_b32 + an lgkmcnt wait, and the epilogues are VALU/issue-bound)
__device__ __forceinline__ float quad_xor1(float v) {      // value of lane ^ 1: quad_perm [1,0,3,2]
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_xor2(float v) {      // value of lane ^ 2: quad_perm [2,3,0,1]
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ void quad_transpose(float (&x)[4], int q) {
    const bool odd = q & 1;
    const float r0 = quad_xor1(odd ? x[0] : x[1]), r1 = quad_xor1(odd ? x[2] : x[3]);
    const float a0 = odd ? r0 : x[0], a1 = odd ? x[1] : r0, a2 = odd ? r1 : x[2], a3 = odd ? x[3] : r1;
    const bool hi = q & 2;
    const float u0 = quad_xor2(hi ? a0 : a2), u1 = quad_xor2(hi ? a1 : a3);
    x[0] = hi ? u0 : a0;
    x[1] = hi ? u1 : a1;
    x[2] = hi ? a2 : u0;
    x[3] = hi ? a3 : u1;
}

// Buffer addressing (scalar resource + 32-bit per-lane byte offset + SCALAR byte offset): the only addressing form
// on gfx950 that costs no VALU instruction per access -- flat/global accesses from hipcc carry a 64-bit VALU add each,
// and next to the f32 MFMA every VALU instruction is paid in full.  The resource spans 4 GiB
// from `base`: callers put the 64-bit part of an address (image / tile / weight-tile base) into `base`.
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);
}
// bounded: accesses at byte offsets >= `bytes` read 0 / are dropped (used to make padding lanes safe without clamps)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// NOTE (tools/microbench/store_hazard.hip): for a buffer store whose soffset is a register hipcc
// (ROCm 7.2) inserts NO wait state before a following VALU write to one of the data registers, and gfx950 then stores the
// NEW value in some lanes (lanes 12-15 of every 16, last dword).  The two wait states are therefore issued here, TIED to the
// data registers ("+v": the asm formally rewrites them), so that no later write can be scheduled in front of them.
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
    asm volatile("s_nop 1" : "+v"(v));
}
// (one dword of data: no such hazard is documented, but the same tie costs one scalar slot and keeps the store's data
// register out of reach of the next instruction all the same)
__device__ __forceinline__ void buf_store1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
    asm volatile("s_nop 0" : "+v"(v));
}
__device__ __forceinline__ void buf_store1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, unsigned v) {
    __builtin_amdgcn_raw_buffer_store_b32(v, r, voff, soff, 0);
    asm volatile("s_nop 0" : "+v"(v));
}

}  // namespace cdx

#define CDX_REQUIRE(cond) \
    do {                  \
        if (!(cond)) return CDX_EINVAL; \
    } while (0)
