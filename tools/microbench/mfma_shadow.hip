// What does each instruction class cost in the shadow of v_mfma_f32_32x32x2_f32 on gfx950?
// A group = 8 MFMAs (512 matrix-pipe cycles) preceded by V scalar VALU adds, L ds_read_b64, G global_load_dwordx4 (L2 hits);
// optional workgroup barrier every 16 groups.  Reported: TF and the extra cycles per group relative to MFMA only.
//   hipcc -O3 --offload-arch=gfx950 mfma_shadow.hip -o mfma_shadow && ./mfma_shadow
#include <hip/hip_runtime.h>
#include <cstdio>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

template <int V, int L, int G, int BAR>
__global__ __launch_bounds__(512) void k(float* out, const float* __restrict__ src, int iters, float b) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    f32x16 acc[8];
    for (int x = 0; x < 8; ++x)
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = b + threadIdx.x * 1e-6f + i;
    f32x2 d[L > 0 ? L : 1];
    f32x4 g[G > 0 ? G : 1];
    for (int i = 0; i < (L > 0 ? L : 1); ++i) d[i] = f32x2{0.f, 0.f};
    for (int i = 0; i < (G > 0 ? G : 1); ++i) g[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int lo = (threadIdx.x & 63) * 2;
    const float* gp = src + (threadIdx.x & 63) * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int grp = 0; grp < 16; ++grp) {
            // consume what the previous group fetched (keeps the loads alive, as operands would)
#pragma unroll
            for (int i = 0; i < L; ++i) v[i & 7] += d[i][0];
#pragma unroll
            for (int i = 0; i < G; ++i) v[i & 7] += g[i][0];
#pragma unroll
            for (int i = 0; i < L; ++i) d[i] = *reinterpret_cast<const f32x2*>(&lds[lo + ((grp * L + i) & 31) * 128]);
#pragma unroll
            for (int i = 0; i < G; ++i) g[i] = *reinterpret_cast<const f32x4*>(gp + ((it * 16 + grp) * G + i) % 64 * 256);
#pragma unroll
            for (int i = 0; i < V; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i & 7]) : "v"(b));
#pragma unroll
            for (int x = 0; x < 8; ++x) acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[x], b, acc[x], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (BAR) __syncthreads();
    }
    float s = 0.f;
    for (int x = 0; x < 8; ++x)
        for (int r = 0; r < 16; ++r) s += acc[x][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 123.456f) out[0] = s;
}

static float* g_out;
static float* g_src;

template <int V, int L, int G, int BAR>
double run(int threads) {
    const int blocks = 1024, iters = 300;
    k<V, L, G, BAR><<<blocks, threads>>>(g_out, g_src, iters, 0.f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) k<V, L, G, BAR><<<blocks, threads>>>(g_out, g_src, iters, 0.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return 3.0 * blocks * (threads / 64) * (double)iters * 128 * 4096 / (ms * 1e-3) / 1e12;
}

template <int V, int L, int G, int BAR>
void report(const char* what, double base) {
    const double tf = run<V, L, G, BAR>(512);
    printf("%-46s %6.1f TF   +%5.1f cycles per 8-MFMA group (512)\n", what, tf, 512.0 * (base / tf - 1.0));
}

int main() {
    hipMalloc(&g_out, 4);
    hipMalloc(&g_src, 64 * 256 * 4 + 4096);
    hipMemset(g_src, 0, 64 * 256 * 4 + 4096);
    const double base = run<0, 0, 0, 0>(512);
    printf("8 waves per workgroup (2 per SIMD), 1024 workgroups; MFMA only: %.1f TF\n", base);
    report<8, 0, 0, 0>("8 VALU / group", base);
    report<16, 0, 0, 0>("16 VALU / group", base);
    report<24, 0, 0, 0>("24 VALU / group", base);
    report<32, 0, 0, 0>("32 VALU / group", base);
    report<0, 3, 0, 0>("3 ds_read_b64 / group", base);
    report<0, 6, 0, 0>("6 ds_read_b64 / group", base);
    report<0, 12, 0, 0>("12 ds_read_b64 / group", base);
    report<0, 0, 2, 0>("2 global_load_dwordx4 / group", base);
    report<0, 0, 4, 0>("4 global_load_dwordx4 / group", base);
    report<0, 0, 0, 1>("barrier / 16 groups", base);
    report<24, 6, 2, 1>("24 VALU + 6 LDS + 2 global + barrier", base);
    return 0;
}
