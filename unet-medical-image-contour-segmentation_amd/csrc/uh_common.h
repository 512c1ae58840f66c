// Shared device/host helpers for libunet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "unet_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define UH_WAVE 64

// ------------------------------------------------------------------ error plumbing (host)
void uh_set_error(const char* fmt, ...);
#define UH_REQUIRE(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) {                                          \
            uh_set_error(__VA_ARGS__);                          \
            return UH_EINVAL;                                   \
        }                                                       \
    } while (0)
#define UH_CHECK_LAUNCH(name)                                                         \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) {                                                      \
            uh_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
            return UH_ELAUNCH;                                                        \
        }                                                                             \
    } while (0)

static inline bool uh_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// stem_mfma.hip (internal): the recomputed stem on the matrix pipe; mode 0 statistics, 1 activation, 2 BatchNorm-backward sums,
// 3 filter-gradient slabs.  Called by the uh_stem_* entry points of conv3x3.hip.
int uh_stem_mfma_launch(int mode, const void* x, int ldx, const void* w, const float* scale, const float* shift, const float* mean,
                        const float* rstd, const float* dgamma, const float* dbeta, float inv_n, const void* dz, int lddz, void* z,
                        int ldz, float* out, int B, int H, int W, int grid, uh_stream stream);

// ------------------------------------------------------------------ dtype helpers (device)
template <typename T> struct uh_traits;
template <> struct uh_traits<float> {
    static constexpr int VEC = 4;   // elements per 16 bytes
};
template <> struct uh_traits<bf16_t> {
    static constexpr int VEC = 8;
};

__device__ __forceinline__ float uh_bf16_bits_to_f32(unsigned short b) {
    return __uint_as_float(((unsigned int)b) << 16);
}
__device__ __forceinline__ float uh_to_f32(float v) { return v; }
__device__ __forceinline__ float uh_to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T uh_from_f32(float v);
template <> __device__ __forceinline__ float uh_from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t uh_from_f32<bf16_t>(float v) { return (bf16_t)v; }
// value as it will be read back from a T store
template <typename T> __device__ __forceinline__ float uh_round_as(float v) { return uh_to_f32(uh_from_f32<T>(v)); }

// 16-byte vector of T <-> floats
template <typename T> struct uh_vec16;
template <> struct uh_vec16<float> {
    f32x4 v;
    static constexpr int N = 4;
    __device__ __forceinline__ float get(int i) const { return v[i]; }
    __device__ __forceinline__ void set(int i, float f) { v[i] = f; }
};
template <> struct uh_vec16<bf16_t> {
    bf16x8 v;
    static constexpr int N = 8;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float f) { v[i] = (bf16_t)f; }
};

// NaN-propagating max (torch.relu / max_pool2d keep a NaN; fmaxf would launder it into a number and the NaN-loss
// check of train.py:149-151 could never fire).  One v_maximum3_f32 on gfx950.
__device__ __forceinline__ float uh_max_nan(float a, float b) { return __builtin_elementwise_maximum(a, b); }
__device__ __forceinline__ float uh_relu(float v) { return __builtin_elementwise_maximum(v, 0.f); }

__device__ __forceinline__ float uh_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// sum over the 16 lanes of a DPP row (lanes 16r..16r+15) with 4 DPP adds; every lane gets the total
template <int CTRL> __device__ __forceinline__ float uh_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float uh_row16_sum(float v) {
    v += uh_dpp<0xB1>(v);     // quad_perm [1,0,3,2]
    v += uh_dpp<0x4E>(v);     // quad_perm [2,3,0,1]
    v += uh_dpp<0x141>(v);    // row_half_mirror
    v += uh_dpp<0x140>(v);    // row_mirror
    return v;
}

__device__ __forceinline__ float uh_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
// softplus-form BCE-with-logits term: max(x,0) - x*t + log1p(exp(-|x|))
__device__ __forceinline__ float uh_bce_logits(float x, float t) {
    return fmaxf(x, 0.0f) - x * t + log1pf(expf(-fabsf(x)));
}
