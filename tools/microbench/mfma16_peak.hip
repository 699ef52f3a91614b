// Sustained rate of the fp16 MFMAs on THIS device with random (non-trivial) operands, for the two instruction shapes:
// the practical ceiling the split / fp16 convolution kernels are measured against.  The 2.5 PF spec figure assumes the
// 2.4 GHz peak clock; under dense fp16 MFMA load the chip clocks lower (MI355X_MICROARCH.md, DVFS give-back).
//   hipcc -O3 --offload-arch=gfx950 mfma16_peak.hip -o mfma16_peak && ./mfma16_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

// SHAPE 0: v_mfma_f32_32x32x16_f16 (4 accumulators of 16 registers); SHAPE 1: v_mfma_f32_16x16x32_f16 (16 x 4 registers)
template <int SHAPE>
__global__ __launch_bounds__(256) void mfma16_loop(float* out, const _Float16* src, int iters) {
    f16x8 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const f16x8*>(src + (threadIdx.x * 6 + i) * 8);
    for (int i = 0; i < 2; ++i) b[i] = *reinterpret_cast<const f16x8*>(src + (threadIdx.x * 6 + 4 + i) * 8);
    float s = 0.f;
    if constexpr (SHAPE == 0) {
        f32x16 acc[4];
        for (int x = 0; x < 4; ++x)
            for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 6; ++rep)
#pragma unroll
                for (int x = 0; x < 4; ++x) acc[x] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[x], b[rep & 1], acc[x], 0, 0, 0);
        }
        for (int x = 0; x < 4; ++x)
            for (int r = 0; r < 16; ++r) s += acc[x][r];
    } else {
        f32x4 acc[16];
        for (int x = 0; x < 16; ++x)
            for (int r = 0; r < 4; ++r) acc[x][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 3; ++rep)
#pragma unroll
                for (int x = 0; x < 16; ++x) acc[x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[x & 3], b[(x >> 2) & 1], acc[x], 0, 0, 0);
        }
        for (int x = 0; x < 16; ++x)
            for (int r = 0; r < 4; ++r) s += acc[x][r];
    }
    if (s == 123.456f) out[0] = s;
}

template <int SHAPE>
double run(const _Float16* src, float* out, int wgs_per_cu, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256 * wgs_per_cu;
    mfma16_loop<SHAPE><<<blocks, 256>>>(out, src, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) mfma16_loop<SHAPE><<<blocks, 256>>>(out, src, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // per wave per iteration: 24 MFMAs of 32768 FLOP (32x32x16) or 48 of 16384 (16x16x32)
    const double flops = 10.0 * blocks * 4 * (double)iters * 24 * 32768.0;
    return flops / (ms * 1e-3) / 1e15;
}

int main() {
    const int n = 256 * 6 * 8;
    _Float16* h = (_Float16*)malloc(n * 2);
    float* out;
    _Float16* src;
    hipMalloc(&out, 4);
    hipMalloc(&src, n * 2);
    for (int zero = 0; zero < 2; ++zero) {
        srand(1);
        for (int i = 0; i < n; ++i) h[i] = zero ? (_Float16)0.f : (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
        hipMemcpy(src, h, n * 2, hipMemcpyHostToDevice);
        for (int w = 1; w <= 2; ++w)
            printf("%s operands, %d wave(s)/SIMD: 32x32x16 %.3f PF | 16x16x32 %.3f PF\n", zero ? "zero  " : "random", w,
                   run<0>(src, out, w, 20000), run<1>(src, out, w, 20000));
    }
    return 0;
}
