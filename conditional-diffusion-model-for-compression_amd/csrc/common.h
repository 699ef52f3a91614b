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

}  // namespace cdx

#define CDX_REQUIRE(cond) \
    do {                  \
        if (!(cond)) return CDX_EINVAL; \
    } while (0)
