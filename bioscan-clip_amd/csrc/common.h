// Shared device/host helpers for libbsclip_hip.so (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bsclip.h"

typedef unsigned short bf16_t;  // raw bfloat16 bits in HBM
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;

#define BSCLIP_WAVE 64

// ---- status / error string -------------------------------------------------------------------------
void bsclip_set_error(const char* fmt, ...);
#define BSCLIP_REQUIRE(cond, ...)                    \
    do {                                             \
        if (!(cond)) {                               \
            bsclip_set_error(__VA_ARGS__);           \
            return BSCLIP_ERR_INVALID;               \
        }                                            \
    } while (0)
#define BSCLIP_LAUNCH_CHECK()                                                        \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) {                                                     \
            bsclip_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,          \
                             hipGetErrorString(e__));                                \
            return BSCLIP_ERR_LAUNCH;                                                \
        }                                                                            \
    } while (0)

// ---- bf16 <-> f32 ----------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return __builtin_bit_cast(unsigned short, h);
}
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
    const f32x2_t v = {lo, hi};  // one v_cvt_pk_bf16_f32 (the scalar-cast-and-or form costs 3 VALU per pair)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

// ---- wave (64-lane) reductions ---------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- exact-GELU pieces (erf via Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7) ---------------------------
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    float p = 1.061405429f;
    p = p * t - 1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t - 0.284496736f;
    p = p * t + 0.254829592f;
    const float y = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(y, x);
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752f)); }
// d/dx gelu(x) = Phi(x) + x * phi(x)
__device__ __forceinline__ float dgelu_f(float x) {
    const float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118654752f));
    const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// ---- async global -> LDS, 16 B per lane (LDS dest = wave-uniform base + lane*16) ---------------------
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
