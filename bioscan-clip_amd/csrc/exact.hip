// "Exact" forward mode (BSCLIP_PARITY=2, round 4): north_star's 1e-3 against the f32 reference on bf16 matrix cores.
//
// A bf16 MFMA GEMM rounds both operands to 8 mantissa bits: 2.4e-3 .. 5.3e-3 per GEMM against f32, 1e-2 .. 2e-2 after twelve
// peaked-softmax layers.  The same matrix cores are exact to ~2^-16 when every operand is carried as hi + lo
// (hi = bf16(x), lo = bf16(x - hi)) and the product is formed as hi.hi + lo.hi + hi.lo -- one GEMM with K tripled:
//     A rows [hi | lo | hi]  x  W rows [hi | hi | lo]
// (the form the patch embedding and the InfoNCE logits already use).  This file holds what that needs beside the GEMM itself:
//   bsclip_split3_rows    f32 activation [M, K]            -> bf16 [M, 3K] = [hi | lo | hi]
//   bsclip_split3_weight  f32 weight [N, K] (+ LoRA B.A)   -> bf16 [N, 3K] = [hi | hi | lo]   (LoRA folded in f32: W + B A)
//   bsclip_gelu_split3    f32 pre-activation [M, N]        -> exact-erf GELU as [hi | lo | hi] + the 8-bit gelu' side band
//   bsclip_attn_fwd_f32   softmax(q k^T scale + bias) v in f32 on the vector ALU (attention is 4 % of the step's FLOPs; the mode
//                         is a parity mode, its speed is reported, not defended), same dropout masks as the bf16 kernels
// The backward pass is the default one: it runs on the bf16 copies of what this forward produced.
// Reference semantics: timm Attention / Mlp (image_encoder.py:108-109), HF BertSelfAttention / BertIntermediate (dna_encoder.py:105).
#include <math.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned ex_u32x4;

__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& lo) {
    hi = pack_bf2(x0, x1);
    lo = pack_bf2(x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u));
}

// one thread: 4 consecutive columns of one row
__global__ void split3_rows_kernel(const float* __restrict__ src, int ld, int M, int K, bf16_t* __restrict__ dst, int ldd) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int kq = K / 4;
    if (t >= (size_t)M * kq) return;
    const int m = (int)(t / kq), k = (int)(t % kq) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)m * ld + k);
    uint2 hi, lo;
    split2(v[0], v[1], hi.x, lo.x);
    split2(v[2], v[3], hi.y, lo.y);
    bf16_t* row = dst + (size_t)m * ldd;
    *reinterpret_cast<uint2*>(row + k) = hi;
    *reinterpret_cast<uint2*>(row + K + k) = lo;
    *reinterpret_cast<uint2*>(row + 2 * K + k) = hi;
}

// W_eff = W + B A on the q rows [0, H) and the v rows [2H, 3H) when lora_a / lora_b are given (lora_a [8, K]: A_q rows 0..3, A_v rows
// 4..7; lora_b [2, H, 4]); rows [hi | hi | lo]
__global__ void split3_weight_kernel(const float* __restrict__ w, int ldw, int N, int K, const float* __restrict__ lora_a,
                                     const float* __restrict__ lora_b, int H, bf16_t* __restrict__ dst, int ldd) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int kq = K / 4;
    if (t >= (size_t)N * kq) return;
    const int n = (int)(t / kq), k = (int)(t % kq) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(w + (size_t)n * ldw + k);
    if (lora_a != nullptr && (n < H || n >= 2 * H)) {
        const int part = n < H ? 0 : 1, r = n < H ? n : n - 2 * H;
        const f32x4 b = *reinterpret_cast<const f32x4*>(lora_b + ((size_t)part * H + r) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(lora_a + (size_t)(4 * part + j) * K + k);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaf(b[j], a[i], v[i]);
        }
    }
    uint2 hi, lo;
    split2(v[0], v[1], hi.x, lo.x);
    split2(v[2], v[3], hi.y, lo.y);
    bf16_t* row = dst + (size_t)n * ldd;
    *reinterpret_cast<uint2*>(row + k) = hi;
    *reinterpret_cast<uint2*>(row + K + k) = hi;
    *reinterpret_cast<uint2*>(row + 2 * K + k) = lo;
}

__global__ void gelu_split3_kernel(const float* __restrict__ z, int ldz, int M, int N, bf16_t* __restrict__ dst, int ldd,
                                   unsigned char* __restrict__ codes, int ldc, float* __restrict__ g32, int ldg) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nq = N / 4;
    if (t >= (size_t)M * nq) return;
    const int m = (int)(t / nq), n = (int)(t % nq) * 4;
    const f32x4 x = *reinterpret_cast<const f32x4*>(z + (size_t)m * ldz + n);
    float g[4], d[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float cdf = 0.5f * (1.0f + erff(x[i] * 0.70710678118654752f));          // exact (erf) GELU, f32
        g[i] = x[i] * cdf;
        d[i] = fmaf(x[i] * 0.39894228040143268f, __expf(-0.5f * x[i] * x[i]), cdf);   // gelu' = Phi + x phi
    }
    if (dst != nullptr) {
        uint2 hi, lo;
        split2(g[0], g[1], hi.x, lo.x);
        split2(g[2], g[3], hi.y, lo.y);
        bf16_t* row = dst + (size_t)m * ldd;
        *reinterpret_cast<uint2*>(row + n) = hi;
        *reinterpret_cast<uint2*>(row + N + n) = lo;
        *reinterpret_cast<uint2*>(row + 2 * N + n) = hi;
    }
    if (g32 != nullptr) *reinterpret_cast<f32x4*>(g32 + (size_t)m * ldg + n) = f32x4{g[0], g[1], g[2], g[3]};
    if (codes != nullptr) *reinterpret_cast<unsigned*>(codes + (size_t)m * ldc + n) = dg8_pack4(d[0], d[1], d[2], d[3]);
}

// mean over the S tokens of every sequence, f32 in and out (HF last_hidden_state.mean(dim=1), language_encoder.py:89)
__global__ void meanpool_f32_kernel(const float* __restrict__ x, int B, int S, int H, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * H) return;
    const int b = t / H, h = t % H;
    float acc = 0.f;
    for (int s = 0; s < S; ++s) acc += x[((size_t)b * S + s) * H + h];
    out[(size_t)b * H + h] = acc / (float)S;
}

// f32 attention forward on the vector ALU: one workgroup per (batch, head), K and V of the head in LDS as f32, one query row per
// thread, online softmax (the output is rescaled only when the running maximum moves).
constexpr int AF_THREADS = 256, AF_SMAX = 224;

template <bool DROP>
__global__ __launch_bounds__(AF_THREADS) void attn_fwd_f32_kernel(const float* __restrict__ qkv, int ld, int S, int heads,
                                                                  const float* __restrict__ key_bias, float scale,
                                                                  float* __restrict__ ctx, int ld_ctx, float* __restrict__ lse,
                                                                  DropCfg drop) {
    BSCLIP_DROP_RESOLVE(drop);
    __shared__ __attribute__((aligned(16))) float sK[AF_SMAX * 64];
    __shared__ __attribute__((aligned(16))) float sV[AF_SMAX * 64];
    __shared__ float sBias[AF_SMAX];
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads, tid = threadIdx.x;
    const int HW = heads * 64;
    const float* qb = qkv + (size_t)b * S * ld + hd * 64;
    for (int i = tid; i < S * 16; i += AF_THREADS) {
        const int r = i >> 4, c = (i & 15) * 4;
        *reinterpret_cast<f32x4*>(sK + r * 64 + c) = *reinterpret_cast<const f32x4*>(qb + (size_t)r * ld + HW + c);
        *reinterpret_cast<f32x4*>(sV + r * 64 + c) = *reinterpret_cast<const f32x4*>(qb + (size_t)r * ld + 2 * HW + c);
    }
    for (int k = tid; k < S; k += AF_THREADS) sBias[k] = key_bias ? key_bias[(size_t)b * S + k] : 0.f;
    __syncthreads();
    const int SP = (S + 31) / 32 * 32;   // the bf16 kernels' padded length: part of the dropout element index
    for (int q = tid; q < S; q += AF_THREADS) {
        float qv[64], o[64];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(qb + (size_t)q * ld + 4 * c);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                qv[4 * c + i] = v[i];
                o[4 * c + i] = 0.f;
            }
        }
        float m = -INFINITY, l = 0.f;
        const unsigned dbase = ((unsigned)(b * heads + hd) * S + (unsigned)q) * SP;
        for (int k = 0; k < S; ++k) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;   // four partial sums: shorter dependency chains, fixed order
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const f32x4 kv = *reinterpret_cast<const f32x4*>(sK + k * 64 + 4 * c);
                s0 = fmaf(qv[4 * c + 0], kv[0], s0);
                s1 = fmaf(qv[4 * c + 1], kv[1], s1);
                s2 = fmaf(qv[4 * c + 2], kv[2], s2);
                s3 = fmaf(qv[4 * c + 3], kv[3], s3);
            }
            const float s = fmaf((s0 + s1) + (s2 + s3), scale, sBias[k]);
            if (s > m) {   // the maximum moves: rescale what has been accumulated (exp(-inf) = 0 on the first key)
                const float corr = __expf(m - s);
                l *= corr;
#pragma unroll
                for (int d = 0; d < 64; ++d) o[d] *= corr;
                m = s;
            }
            const float p = __expf(s - m);
            l += p;                                   // the softmax sum is taken before dropout (HF: dropout(softmax(.)))
            float pk = p;
            if constexpr (DROP) pk = p * drop_factor(drop, dbase + (unsigned)k);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const f32x4 vv = *reinterpret_cast<const f32x4*>(sV + k * 64 + 4 * c);
#pragma unroll
                for (int i = 0; i < 4; ++i) o[4 * c + i] = fmaf(pk, vv[i], o[4 * c + i]);
            }
        }
        const float inv = 1.0f / l;
        float* out = ctx + (size_t)(b * S + q) * ld_ctx + hd * 64;
#pragma unroll
        for (int c = 0; c < 16; ++c)
            *reinterpret_cast<f32x4*>(out + 4 * c) = f32x4{o[4 * c] * inv, o[4 * c + 1] * inv, o[4 * c + 2] * inv, o[4 * c + 3] * inv};
        lse[((size_t)b * heads + hd) * S + q] = m + __logf(l);
    }
}

}  // namespace

extern "C" int bsclip_split3_rows(const float* src, int ld_src, int M, int K, void* dst, int ld_dst, void* stream) {
    BSCLIP_REQUIRE(src && dst, "bsclip_split3_rows: null pointer");
    BSCLIP_REQUIRE(M > 0 && K > 0 && K % 4 == 0 && ld_src >= K && ld_src % 4 == 0 && ld_dst >= 3 * K && ld_dst % 4 == 0,
                   "bsclip_split3_rows: M=%d K=%d ld_src=%d ld_dst=%d", M, K, ld_src, ld_dst);
    const size_t n = (size_t)M * (K / 4);
    hipLaunchKernelGGL(split3_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), src, ld_src,
                       M, K, static_cast<bf16_t*>(dst), ld_dst);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int bsclip_split3_weight(const float* w, int ld_w, int N, int K, const float* lora_a, const float* lora_b, int H, void* dst,
                                    int ld_dst, void* stream) {
    BSCLIP_REQUIRE(w && dst, "bsclip_split3_weight: null pointer");
    BSCLIP_REQUIRE((lora_a == nullptr) == (lora_b == nullptr), "bsclip_split3_weight: lora_a and lora_b go together");
    BSCLIP_REQUIRE(N > 0 && K > 0 && K % 4 == 0 && ld_w >= K && ld_w % 4 == 0 && ld_dst >= 3 * K && ld_dst % 4 == 0 &&
                       (lora_a == nullptr || (N == 3 * H && K == H)),
                   "bsclip_split3_weight: N=%d K=%d H=%d ld_w=%d ld_dst=%d", N, K, H, ld_w, ld_dst);
    const size_t n = (size_t)N * (K / 4);
    hipLaunchKernelGGL(split3_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), w, ld_w, N,
                       K, lora_a, lora_b, H, static_cast<bf16_t*>(dst), ld_dst);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int bsclip_gelu_split3(const float* z, int ld_z, int M, int N, void* dst, int ld_dst, void* codes, int ld_codes, float* g32,
                                  int ld_g32, void* stream) {
    BSCLIP_REQUIRE(z && (dst || g32), "bsclip_gelu_split3: null pointer");
    BSCLIP_REQUIRE(M > 0 && N > 0 && N % 4 == 0 && ld_z >= N && ld_z % 4 == 0 && (dst == nullptr || (ld_dst >= 3 * N && ld_dst % 4 == 0)) &&
                       (codes == nullptr || (ld_codes >= N && ld_codes % 4 == 0)) && (g32 == nullptr || (ld_g32 >= N && ld_g32 % 4 == 0)),
                   "bsclip_gelu_split3: M=%d N=%d ld_z=%d ld_dst=%d ld_codes=%d ld_g32=%d", M, N, ld_z, ld_dst, ld_codes, ld_g32);
    const size_t n = (size_t)M * (N / 4);
    hipLaunchKernelGGL(gelu_split3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), z, ld_z, M, N,
                       static_cast<bf16_t*>(dst), ld_dst, static_cast<unsigned char*>(codes), ld_codes, g32, ld_g32);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int bsclip_meanpool_tokens_f32(const float* x, int B, int S, int H, float* out, void* stream) {
    BSCLIP_REQUIRE(x && out && B > 0 && S > 0 && H > 0, "bsclip_meanpool_tokens_f32: bad arguments");
    hipLaunchKernelGGL(meanpool_f32_kernel, dim3((B * H + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), x, B, S, H, out);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int bsclip_attn_fwd_f32(const float* qkv, int ld_qkv, int B, int S, int heads, const float* key_bias, float scale, float* ctx,
                                   int ld_ctx, float* lse, float dropout_p, uint32_t dropout_seed, void* stream) {
    BSCLIP_REQUIRE(qkv && ctx && lse, "bsclip_attn_fwd_f32: null pointer");
    BSCLIP_REQUIRE(B > 0 && heads > 0 && S > 0 && S <= AF_SMAX, "bsclip_attn_fwd_f32: B=%d heads=%d S=%d (S <= 224)", B, heads, S);
    BSCLIP_REQUIRE(ld_qkv >= 3 * heads * 64 && ld_qkv % 4 == 0 && ld_ctx >= heads * 64 && ld_ctx % 4 == 0,
                   "bsclip_attn_fwd_f32: ld_qkv=%d ld_ctx=%d", ld_qkv, ld_ctx);
    BSCLIP_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "bsclip_attn_fwd_f32: dropout_p=%f", dropout_p);
    const DropCfg drop = make_drop(dropout_p, dropout_seed);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (drop.thr16)
        hipLaunchKernelGGL((attn_fwd_f32_kernel<true>), dim3(B * heads), dim3(AF_THREADS), 0, s, qkv, ld_qkv, S, heads, key_bias, scale, ctx,
                           ld_ctx, lse, drop);
    else
        hipLaunchKernelGGL((attn_fwd_f32_kernel<false>), dim3(B * heads), dim3(AF_THREADS), 0, s, qkv, ld_qkv, S, heads, key_bias, scale, ctx,
                           ld_ctx, lse, drop);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}
