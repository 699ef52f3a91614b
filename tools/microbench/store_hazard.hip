// Does a 16-byte buffer store pick up values written to its data registers AFTER it was issued?
// Mimics the epilogue that was flaky in the 8-wave Winograd kernel: 8 waves of a 512-thread
// workgroup (two per SIMD, 200+ VGPRs each so that exactly one workgroup fits a CU) leave a barrier together and each
// issues 8 x (4x4 quad transpose through ds_bpermute -> 16-byte store), the compiler free to recycle the four data
// registers for the next transpose: in the ISA a VALU write to a data register follows the store directly.  Every
// stored dword is checked on the host.  Result on MI355X / ROCm 7.2 (hipcc clang 22):
//   mode 0  buffer_store_dwordx4 v[a:a+3], voff, rsrc, sN offen   (SGPR soffset)  -> corrupted (lanes 12-15 of every 16,
//           dword 3, mostly the second wave of a SIMD): hipcc inserts NO wait state between the store and the VALU write
//           (its ">64-bit store data" hazard rule exempts MUBUF stores whose soffset is a register)
//   mode 1  the same store with soffset = 0 (offset folded into the VGPR)        -> clean: hipcc inserts s_nop 0
//   mode 2  global_store_dwordx4                                                 -> clean (rule applies to FLAT)
//   hipcc -O3 --offload-arch=gfx950 store_hazard.hip -o store_hazard && ./store_hazard [launches]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

__device__ __forceinline__ void quad_transpose(float (&x)[4], int q) {
    const bool odd = q & 1;
    const float r0 = __shfl_xor(odd ? x[0] : x[1], 1), r1 = __shfl_xor(odd ? x[2] : x[3], 1);
    const float a0 = odd ? r0 : x[0], a1 = odd ? x[1] : r0, a2 = odd ? r1 : x[2], a3 = odd ? x[3] : r1;
    const bool hi = q & 2;
    const float u0 = __shfl_xor(hi ? a0 : a2, 2), u1 = __shfl_xor(hi ? a1 : a3, 2);
    x[0] = hi ? u0 : a0;
    x[1] = hi ? u1 : a1;
    x[2] = hi ? a2 : u0;
    x[3] = hi ? a3 : u1;
}

// value stored at (workgroup g, wave w, store s, lane l, dword c) -- exactly representable
__host__ __device__ inline float expect(int g, int w, int s, int l, int c) { return (float)(((g * 8 + w) * 8 + s) * 256 + l * 4 + c); }

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, const float* pad_src, int pad_n, int spread) {
    __shared__ float lds[4096];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), g = blockIdx.x;
    // ballast: ~200 live registers so that two waves per SIMD fill the register file (one workgroup per CU)
    float ballast[192];
#pragma unroll
    for (int i = 0; i < 192; ++i) ballast[i] = pad_src[(tid + i * 7) % pad_n];
    lds[tid] = ballast[tid & 127];
    __syncthreads();
    const int q = lane & 3;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)(g * 8 + w) * 8 * 256, 0, -1, 0x00020000);
    // pre-transposed source values: lane l holds, for store s, the 4 values that lanes of its quad need
    float src[8][4];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // after the transpose lane (quad base + i') dword c must be expect(.., lane = quad base + i', c):
            // before it, lane q holds x[i] = value(row i, col q) = final lane (base + i), dword q
            src[s][i] = expect(g, w, s, (lane & ~3) + i, q);
        }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        float x[4] = {src[s][0], src[s][1], src[s][2], src[s][3]};
        quad_transpose(x, q);
        // MODE 0: scalar offset in an SGPR (what the kernel did); 1: the same offset folded into the VGPR offset, soffset = 0;
        // 2: global_store_dwordx4 through a flat pointer
        const f32x4 v = f32x4{x[0], x[1], x[2], x[3]};
        if (MODE == 0) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, lane * 16, s * 1024 * spread, 0);
        else if (MODE == 1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, lane * 16 + s * 1024 * spread, 0, 0);
        else if (MODE == 2) *reinterpret_cast<f32x4*>(out + (size_t)(g * 8 + w) * 8 * 256 + s * 256 + lane * 4) = v;
        else {      // 3: SGPR soffset, followed by wait states that are TIED to the data registers (the asm "rewrites" them, so
                    // no later write can be scheduled in front of it)
            f32x4 t = v;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t), rs, lane * 16, s * 1024 * spread, 0);
            asm volatile("s_nop 1" : "+v"(t));
            x[0] = t[0];      // keep the tie alive into the next iteration's registers
        }
    }
    float keep = 0.f;
#pragma unroll
    for (int i = 0; i < 192; ++i) keep += ballast[i];
    if (keep == 123.456f) out[0] = keep + lds[lane];
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 200;
    const int G = 2048;
    const size_t n = (size_t)G * 8 * 8 * 256;
    float *out, *pad;
    hipMalloc(&out, n * 4);
    hipMalloc(&pad, 4096 * 4);
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)(i % 97) * 0.5f;
    hipMemcpy(pad, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    std::vector<float> res(n);
  for (int mode = 0; mode < 4; ++mode) {
    long bad_launches = 0, bad_dwords = 0;
    int shown = 0;
    for (int it = 0; it < launches; ++it) {
        hipMemset(out, 0xff, n * 4);
        if (mode == 0) k<0><<<G, 512>>>(out, pad, 4096, 1);
        else if (mode == 1) k<1><<<G, 512>>>(out, pad, 4096, 1);
        else if (mode == 2) k<2><<<G, 512>>>(out, pad, 4096, 1);
        else k<3><<<G, 512>>>(out, pad, 4096, 1);
        hipMemcpy(res.data(), out, n * 4, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int g = 0; g < G; ++g)
            for (int w = 0; w < 8; ++w)
                for (int s = 0; s < 8; ++s)
                    for (int l = 0; l < 64; ++l)
                        for (int c = 0; c < 4; ++c) {
                            const float v = res[(((size_t)(g * 8 + w) * 8 + s) * 64 + l) * 4 + c];
                            if (v != expect(g, w, s, l, c)) {
                                ++bad;
                                if (shown < 3) {
                                    printf("launch %d: wg %d wave %d store %d lane %d dword %d: got %.1f want %.1f\n", it, g, w, s, l, c, v, expect(g, w, s, l, c));
                                    ++shown;
                                }
                            }
                        }
        bad_launches += bad != 0;
        bad_dwords += bad;
    }
    printf("store_hazard mode %d (0 = buffer store with an SGPR soffset, 1 = buffer store with soffset 0, 2 = global store, 3 = mode 0 + s_nop 1 tied to the data registers): %ld bad launches of %d, %ld bad dwords (%s)\n", mode, bad_launches, launches, bad_dwords, hipGetErrorString(hipGetLastError()));
  }
    return 0;
}
