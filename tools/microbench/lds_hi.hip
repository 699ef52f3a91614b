// Does a 156 KiB static LDS workgroup (512 threads, 2 waves/SIMD, raw s_barrier) read back what other waves wrote above 128 KiB?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512, 2) void k(unsigned* out, int rounds) {
    __shared__ float lds[39936];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* xch = lds + 31744;
    unsigned bad = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int k = 0; k < 16; ++k) xch[(wave * 16 + k) * 64 + lane] = (float)(r * 1000 + wave * 16 + k) + lane * 0.001f;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int partner = wave ^ 1;
        for (int k = 0; k < 16; ++k) {
            const float v = xch[(partner * 16 + k) * 64 + lane];
            if (v != (float)(r * 1000 + partner * 16 + k) + lane * 0.001f) bad++;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    atomicAdd(out, bad);
}
int main() {
    unsigned* d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
    k<<<1024, 512>>>(d, 200);
    unsigned h = 0; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("mismatches: %u (%s)\n", h, hipGetErrorString(hipGetLastError()));
    return 0;
}
