// Sustained rate of v_mfma_f32_32x32x2_f32 on this device: the practical ceiling the convolution kernels are
// measured against (the 157.3 TF spec figure assumes the 2.4 GHz peak clock for the whole run).
//   hipcc -O3 --offload-arch=gfx950 mfma_peak.hip -o mfma_peak && ./mfma_peak [waves_per_simd] [valu_per_mfma]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int VALU, bool PK = false>
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters, float a, float b) {
    f32x16 acc[8];
    for (int x = 0; x < 8; ++x)
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + threadIdx.x * 1e-6f + i;
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    f32x2 pk[4] = {{a, b}, {a, b}, {a, b}, {a, b}}, pb = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[x], b, acc[x], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < VALU; ++k) {
                    if constexpr (PK) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pk[(x + k) & 3]) : "v"(pb));
                    else asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(x + k + 1) & 7]) : "v"(b));
                }
            }
        }
    }
    float s = 0.f;
    for (int x = 0; x < 8; ++x)
        for (int r = 0; r < 16; ++r) s += acc[x][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += pk[i][0] + pk[i][1];
    if (s == 123.456f) out[0] = s;
}

template <int VALU, bool PK = false>
double run(int threads, int blocks, int iters) {
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    mfma_loop<VALU, PK><<<blocks, threads>>>(out, iters, 1.f, 0.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) mfma_loop<VALU, PK><<<blocks, threads>>>(out, iters, 1.f, 0.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 5.0 * blocks * (threads / 64) * (double)iters * 32 * 4096;
    hipFree(out);
    return flops / (ms * 1e-3) / 1e12;
}

int main(int argc, char** argv) {
    const int iters = 4000;
    for (int wps = 1; wps <= 2; ++wps) {
        const int threads = 256 * wps, blocks = 256 * 4;
        printf("waves/SIMD %d: MFMA only %.1f TF | +1 VALU/MFMA %.1f | +2 %.1f | +3 %.1f | +4 %.1f | +6 %.1f\n", wps,
               run<0>(threads, blocks, iters), run<1>(threads, blocks, iters), run<2>(threads, blocks, iters),
               run<3>(threads, blocks, iters), run<4>(threads, blocks, iters), run<6>(threads, blocks, iters));
        printf("              v_pk_add_f32: +1 %.1f | +2 %.1f | +3 %.1f\n", run<1, true>(threads, blocks, iters),
               run<2, true>(threads, blocks, iters), run<3, true>(threads, blocks, iters));
    }
    return 0;
}
