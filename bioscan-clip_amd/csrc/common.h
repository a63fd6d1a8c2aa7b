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

// ---- dropout: counter-based, layout-independent ----------------------------------------------------
// The decision for logical element `idx` of a site depends only on (seed, idx): every kernel that touches the
// element (forward epilogue, the backward pass that regenerates the mask, either attention orientation) gets the
// same answer without storing masks.  One 32-bit hash serves the element pair (idx & ~1, idx | 1): 16 bits each,
// keep <=> bits >= thr16 with thr16 = round(p * 65536).  HF BERT: hidden_dropout_prob /
// attention_probs_dropout_prob = 0.1, active under model.train().
struct DropCfg {
    unsigned thr16;  // 0 = dropout off
    unsigned seed;
    float scale;     // 1 / (1 - p)
    const unsigned* step;  // device word mixed into the seed when the kernel RUNS (nullable): lets a captured hipGraph draw new
                           // masks on every replay -- kernel arguments are frozen at capture, device memory is not
};
// first statement of every kernel that takes a DropCfg by value.  The step goes through the mixer: added linearly with the
// element multiplier (round 2) it made step s+1's mask the mask of step s slid by one element pair.
#define BSCLIP_DROP_RESOLVE(d)                                                                    \
    do {                                                                                          \
        if ((d).thr16 && (d).step) (d).seed = hash32((d).seed ^ hash32(*(d).step + 0x85EBCA6BU)); \
    } while (0)
__device__ __forceinline__ unsigned hash32(unsigned x) {  // "lowbias32" integer mixer
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned drop_pair_bits(const DropCfg& d, unsigned idx) {
    return hash32((idx >> 1) * 0x9E3779B9U + d.seed);
}
__device__ __forceinline__ float drop_factor(const DropCfg& d, unsigned idx) {  // 1/(1-p) if kept, 0 if dropped
    const unsigned bits = (drop_pair_bits(d, idx) >> ((idx & 1) * 16)) & 0xffffU;
    return bits >= d.thr16 ? d.scale : 0.f;
}
// four consecutive elements starting at idx (idx % 4 == 0)
__device__ __forceinline__ f32x4 drop4(const DropCfg& d, unsigned idx, f32x4 v) {
    const unsigned b0 = drop_pair_bits(d, idx), b1 = drop_pair_bits(d, idx + 2);
    v[0] = (b0 & 0xffffU) >= d.thr16 ? v[0] * d.scale : 0.f;
    v[1] = (b0 >> 16) >= d.thr16 ? v[1] * d.scale : 0.f;
    v[2] = (b1 & 0xffffU) >= d.thr16 ? v[2] * d.scale : 0.f;
    v[3] = (b1 >> 16) >= d.thr16 ? v[3] * d.scale : 0.f;
    return v;
}
const unsigned* bsclip_current_dropout_step();  // api.hip: the calling thread's bsclip_set_dropout_step pointer
static inline DropCfg make_drop(float p, unsigned seed) {
    DropCfg d;
    d.thr16 = p > 0.f ? (unsigned)(p * 65536.0f + 0.5f) : 0u;
    d.seed = seed;
    d.step = p > 0.f ? bsclip_current_dropout_step() : nullptr;
    d.scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    return d;
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
// One exp and one rcp per element serve both gelu and gelu': with u = |x|/sqrt2, E = exp(-u^2) = exp(-x^2/2),
//   erf(u) = 1 - poly(t) E,  t = 1/(1 + p u);   Phi(x) = 0.5 (1 + sign(x) erf(u));   phi(x) = E / sqrt(2 pi).
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& e) {
    const float u = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, u, 1.0f));
    float p = 1.061405429f;
    p = fmaf(p, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170f);  // exp(-x^2/2) = 2^(-x^2 / (2 ln 2))
    const float half_tail = 0.5f * p * t * e;                   // 0.5 * erfc(u)
    cdf = x >= 0.f ? 1.0f - half_tail : half_tail;
}
__device__ __forceinline__ float gelu_f(float x) {
    float cdf, e;
    gelu_parts(x, cdf, e);
    return x * cdf;
}
// gelu(x) and d/dx gelu(x) = Phi(x) + x * phi(x) from one exp + one rcp
__device__ __forceinline__ void gelu_both(float x, float& gl, float& dg) {
    float cdf, e;
    gelu_parts(x, cdf, e);
    gl = x * cdf;
    dg = fmaf(x * 0.39894228040143268f, e, cdf);
}
__device__ __forceinline__ float dgelu_f(float x) {
    float cdf, e;
    gelu_parts(x, cdf, e);
    return fmaf(x * 0.39894228040143268f, e, cdf);
}

// ---- OCP fp8 e4m3 (gfx950: e4m3fn, max 448, no infinities) ---------------------------------------------
// v_cvt_pk_fp8_f32 rounds to nearest even; inputs are clamped to +-448 first so an overflow saturates instead of becoming
// NaN (0x7f).  Four values -> one dword, element i in byte i.
__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d) {
    a = __builtin_amdgcn_fmed3f(a, -448.0f, 448.0f);
    b = __builtin_amdgcn_fmed3f(b, -448.0f, 448.0f);
    c = __builtin_amdgcn_fmed3f(c, -448.0f, 448.0f);
    d = __builtin_amdgcn_fmed3f(d, -448.0f, 448.0f);
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}
__device__ __forceinline__ float fp8_to_f32(unsigned w, int byte) {  // byte is a compile-time constant at every call site
    switch (byte) {
        case 0: return __builtin_amdgcn_cvt_f32_fp8((int)w, 0);
        case 1: return __builtin_amdgcn_cvt_f32_fp8((int)w, 1);
        case 2: return __builtin_amdgcn_cvt_f32_fp8((int)w, 2);
        default: return __builtin_amdgcn_cvt_f32_fp8((int)w, 3);
    }
}

// ---- 8-bit side band for gelu'(pre-activation) -------------------------------------------------------
// The forward fc1 epilogue saves gelu'(z) for the backward pass (so dfc2's epilogue is one multiply).  gelu' lives in
// [-0.1290, 1.1290]; a uniform 8-bit code over [-0.13, 1.13] has step 4.9e-3, i.e. rounding noise 1.4e-3 rms -- the
// size of the bf16 rounding of a value near 1 (2^-9 .. 2^-8) -- at half the bytes of the bf16 band it replaces
// (fc1 wrote 12 KB per row, now 9 KB; dfc2 read 6 KB of side band per row, now 3 KB).
constexpr float DG8_OFF = 0.13f, DG8_SCALE = 255.0f / 1.26f, DG8_STEP = 1.26f / 255.0f;
// v_cvt_pk_u8_f32 converts (round to nearest even, saturating to [0, 255]: tools/probe/cvt_pk_u8_probe.hip) AND inserts the byte
// into a dword: one instruction per value where clamp + convert + shift/or took four (the GELU epilogue is VALU-bound).
__device__ __forceinline__ unsigned dg8_pack4(float a, float b, float c, float d) {
    unsigned w = 0;
    w = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(a, DG8_SCALE, DG8_OFF * DG8_SCALE), 0, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(b, DG8_SCALE, DG8_OFF * DG8_SCALE), 1, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(c, DG8_SCALE, DG8_OFF * DG8_SCALE), 2, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(d, DG8_SCALE, DG8_OFF * DG8_SCALE), 3, w);
    return w;
}
// the same for two pairs, the scaling on the packed-f32 VALU (one v_pk_fma_f32 per pair instead of two v_fma_f32)
typedef float dg8_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned dg8_pack4(dg8_f32x2 ab, dg8_f32x2 cd) {
    ab = ab * DG8_SCALE + DG8_OFF * DG8_SCALE;
    cd = cd * DG8_SCALE + DG8_OFF * DG8_SCALE;
    unsigned w = 0;
    w = __builtin_amdgcn_cvt_pk_u8_f32(ab[0], 0, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(ab[1], 1, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(cd[0], 2, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(cd[1], 3, w);
    return w;
}
__device__ __forceinline__ f32x4 dg8_unpack4(unsigned w) {
    return f32x4{fmaf((float)(w & 0xff), DG8_STEP, -DG8_OFF), fmaf((float)((w >> 8) & 0xff), DG8_STEP, -DG8_OFF),
                 fmaf((float)((w >> 16) & 0xff), DG8_STEP, -DG8_OFF), fmaf((float)(w >> 24), DG8_STEP, -DG8_OFF)};
}

// ---- workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope release/acquire fence: the compiler
// puts s_waitcnt vmcnt(0) in front of the s_barrier, i.e. every global load issued "ahead" is waited for at the next barrier and
// every global store has to be acknowledged before the workgroup moves on.  An epilogue that stages through LDS only needs its
// ds_writes / ds_reads ordered; global loads are waited for where their values are used, stores drain on their own. ----
// (Fences scoped to the "local" address space keep the compiler's own counter bookkeeping intact; an inline-asm
// "s_waitcnt lgkmcnt(0); s_barrier" makes it fall back to vmcnt(0) before every later global access.)
#define BSCLIP_LDS_BARRIER()                                               \
    do {                                                                   \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");    \
        __builtin_amdgcn_s_barrier();                                      \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");    \
    } while (0)

// ---- non-temporal (streaming) global stores for kernel outputs that nothing in the same kernel reads back: the lines do not
// displace the operand tiles other workgroups are re-reading from L2 (GEMM epilogues: csrc/gemm_pers.h has the measurement) ----
typedef unsigned nt_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned nt_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void nt_store(void* p, uint4 v) {
    __builtin_nontemporal_store(nt_u32x4{v.x, v.y, v.z, v.w}, static_cast<nt_u32x4*>(p));
}
__device__ __forceinline__ void nt_store(void* p, uint2 v) {
    __builtin_nontemporal_store(nt_u32x2{v.x, v.y}, static_cast<nt_u32x2*>(p));
}
__device__ __forceinline__ void nt_store(void* p, f32x4 v) { __builtin_nontemporal_store(v, static_cast<f32x4*>(p)); }

// ---- streaming accesses of the memory-bound kernels (rows read once, written once).  Compile-time switch for A/B runs:
// -DBSCLIP_STREAM_NT marks them non-temporal so that they do not displace the operand tiles of a GEMM running on the other
// tower's stream (`make exp` builds ../lib/libbsclip_hip_exp.so with it; BSCLIP_LIB selects the library) ----
#ifdef BSCLIP_STREAM_NT
__device__ __forceinline__ uint2 ld_stream(const uint2* p) {
    const nt_u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u32x2*>(p));
    return uint2{v[0], v[1]};
}
__device__ __forceinline__ f32x4 ld_stream(const f32x4* p) { return __builtin_nontemporal_load(p); }
template <class T>
__device__ __forceinline__ void st_stream(void* p, T v) { nt_store(p, v); }
__device__ __forceinline__ void st_stream(void* p, unsigned v) { __builtin_nontemporal_store(v, static_cast<unsigned*>(p)); }
#else
__device__ __forceinline__ uint2 ld_stream(const uint2* p) { return *p; }
__device__ __forceinline__ f32x4 ld_stream(const f32x4* p) { return *p; }
template <class T>
__device__ __forceinline__ void st_stream(void* p, T v) { *static_cast<T*>(p) = v; }
#endif

// ---- async global -> LDS, 16 B per lane (LDS dest = wave-uniform base + lane*16) ---------------------
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// fullft.hip, internal: out0 / out1 [c] += sum_b partial[b][c] in a fixed order (columns < split go to out0, the rest to out1)
void bsclip_launch_slab_reduce_add(const float* partial, int nblocks, int n, float* out0, float* out1, int split, hipStream_t s);

// attn_x3.hip, internal: exact-mode attention on split-bf16 operands (argument checks: exact.hip's bsclip_attn_fwd_f32 / bwd_f32)
void bsclip_launch_attn_fwd_x3(const float* qkv, int ld_qkv, int B, int S, int heads, const float* key_bias, float scale, float* ctx,
                               int ld_ctx, float* lse, const DropCfg& drop, bf16_t* ctx3, int ld_c3, hipStream_t s);
void bsclip_launch_attn_bwd_x3(const float* qkv, int ld_qkv, const float* dctx, int ld_dctx, const float* ctx, int ld_ctx, const float* lse,
                               int B, int S, int heads, const float* key_bias, float scale, float* dqkv, int ld_dqkv, const DropCfg& drop,
                               bf16_t* dqkv3, int ld_d3, hipStream_t s);

// gemm.hip, internal: the fused InfoNCE products on the 256x256 ping-pong GEMM (see the EPI_LSE_PART / EPI_LOSS_W epilogues)
int bsclip_gemm_infonce(int mode, const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                        const int64_t* labels, const float* cnt, const float* lse_row, const float* lse_col, float* part,
                        float logit_scale, float coef, int n_valid, int row_base, void* stream);
