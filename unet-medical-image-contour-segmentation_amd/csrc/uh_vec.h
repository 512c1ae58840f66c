// V-wide channel vector load/store helpers (V = 1 scalar fallback, V = 16 bytes worth otherwise).
#pragma once
#include "uh_common.h"

template <typename T, int V>
__device__ __forceinline__ void uh_load(const T* p, float (&o)[V]) {
    if constexpr (V == 1) {
        o[0] = uh_to_f32(p[0]);
    } else if constexpr (sizeof(T) == 2) {
        static_assert(V == 8, "bf16 vector is 8 wide");
        bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
    } else {
        static_assert(V == 4, "f32 vector is 4 wide");
        f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = v[i];
    }
}

template <typename T, int V>
__device__ __forceinline__ void uh_store(T* p, const float (&o)[V]) {
    if constexpr (V == 1) {
        p[0] = uh_from_f32<T>(o[0]);
    } else if constexpr (sizeof(T) == 2) {
        bf16x8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (bf16_t)o[i];
        *reinterpret_cast<bf16x8*>(p) = v;
    } else {
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = o[i];
        *reinterpret_cast<f32x4*>(p) = v;
    }
}

// can a tensor (pointer, pixel stride, channel count) be walked in 16-byte channel vectors?
template <typename T>
static inline bool uh_vec_ok(const void* p, int ld, int C) {
    constexpr int VEC = 16 / (int)sizeof(T);
    return uh_aligned16(p) && (ld % VEC == 0) && (C % VEC == 0);
}

// Dispatch helper: calls f(std::integral_constant<int,V>) with V = VEC or 1.
#define UH_DISPATCH_DT(dt, T, ...)                         \
    do {                                                   \
        if ((dt) == UH_BF16) { using T = bf16_t; __VA_ARGS__ } \
        else { using T = float; __VA_ARGS__ }              \
    } while (0)
