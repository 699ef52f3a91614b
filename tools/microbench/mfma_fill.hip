// What rides for free in the shadow of v_mfma_f32_32x32x16_f16?  One wave (or two) per SIMD issues MFMAs with NF independent
// filler instructions behind each: v_fma_f32 (kind 0), v_exp_f32 (1), ds_read_b128 (2), buffer-style global loads (3).
// Prints the time per MFMA in units of the filler-free loop.   hipcc -O3 --offload-arch=gfx950 mfma_fill.hip -o mfma_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

template <int KIND, int NF>
__global__ __launch_bounds__(256) void loop(float* out, const _Float16* src, const f32x4* g, int iters) {
    __shared__ f32x4 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    f16x8 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const f16x8*>(src + (threadIdx.x * 6 + i) * 8);
    for (int i = 0; i < 2; ++i) b[i] = *reinterpret_cast<const f16x8*>(src + (threadIdx.x * 6 + 4 + i) * 8);
    f32x16 acc[4];
    for (int x = 0; x < 4; ++x)
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
    float f[8];
    for (int k = 0; k < 8; ++k) f[k] = 0.5f + k + threadIdx.x * 1e-3f;
    f32x4 v[8];
    for (int k = 0; k < 8; ++k) v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 6; ++rep)
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                __builtin_amdgcn_sched_barrier(0);
                acc[x] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[x], b[rep & 1], acc[x], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < NF; ++k) {
                    if constexpr (KIND == 0) f[k & 7] = __builtin_fmaf(f[k & 7], 1.0001f, 0.25f);
                    if constexpr (KIND == 1) f[k & 7] = __builtin_amdgcn_exp2f(f[k & 7]) * 0.f + 0.5f * (k + 1);
                    if constexpr (KIND == 2) v[k & 7] = lds[(threadIdx.x + 64 * k + it) & 1023];
                    if constexpr (KIND == 3) v[k & 7] = g[(threadIdx.x + 256 * ((k + it) & 63))];
                }
            }
    }
    float s = 0.f;
    for (int x = 0; x < 4; ++x)
        for (int r = 0; r < 16; ++r) s += acc[x][r];
    for (int k = 0; k < 8; ++k) s += f[k] + v[k][0] + v[k][3];
    if (s == 123.456f) out[0] = s;
}

template <int KIND, int NF>
double run(const _Float16* src, const f32x4* g, float* out, int wgs_per_cu, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256 * wgs_per_cu;
    loop<KIND, NF><<<blocks, 256>>>(out, src, g, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) loop<KIND, NF><<<blocks, 256>>>(out, src, g, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

template <int KIND>
void sweep(const char* name, const _Float16* src, const f32x4* g, float* out) {
    for (int w = 1; w <= 2; ++w) {
        const double base = run<0, 0>(src, g, out, w, 4000);
        printf("%-14s %d wave(s)/SIMD: time per MFMA relative to the bare loop, fillers per MFMA 1 2 3 4 6 8: %.3f %.3f %.3f %.3f %.3f %.3f\n", name, w,
               run<KIND, 1>(src, g, out, w, 4000) / base, run<KIND, 2>(src, g, out, w, 4000) / base, run<KIND, 3>(src, g, out, w, 4000) / base,
               run<KIND, 4>(src, g, out, w, 4000) / base, run<KIND, 6>(src, g, out, w, 4000) / base, run<KIND, 8>(src, g, out, w, 4000) / base);
    }
}

int main() {
    const int n = 256 * 6 * 8;
    _Float16* h = (_Float16*)malloc(n * 2);
    float* out;
    _Float16* src;
    f32x4* g;
    hipMalloc(&out, 4);
    hipMalloc(&src, n * 2);
    hipMalloc(&g, 256 * 64 * 16);
    hipMemset(g, 0, 256 * 64 * 16);
    srand(1);
    for (int i = 0; i < n; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
    hipMemcpy(src, h, n * 2, hipMemcpyHostToDevice);
    sweep<0>("v_fma_f32", src, g, out);
    sweep<1>("v_exp_f32+fma", src, g, out);
    sweep<2>("ds_read_b128", src, g, out);
    sweep<3>("global_load x4", src, g, out);
    return 0;
}
