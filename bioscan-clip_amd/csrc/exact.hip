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
//   bsclip_attn_fwd_f32 / bsclip_attn_bwd_f32   softmax(q k^T scale + bias) v and its backward in f32, on the matrix pipe's f32
//                         form (v_mfma_f32_32x32x2_f32: exact f32 FMA chains at the vector-ALU peak rate without the LDS-broadcast
//                         bottleneck of a one-row-per-thread kernel, which is kept as a second implementation:
//                         bsclip_exact_attn_set_impl(1)); same dropout masks as the bf16 kernels
//   ... and the exact backward: bsclip_dgelu_split3, bsclip_split3_transpose, bsclip_softmax_meanpool_bwd_f32 (bsclip_lora_grad_f32: optim.hip).
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

// ---- f32 attention on the matrix pipe: v_mfma_f32_32x32x2_f32 (f32 operands, exact f32 FMA chains; 64 cycles per SIMD) -------------
// One workgroup per (batch, head), four waves; K and V of the head in LDS as f32 (row pitch 65 floats: a column read by 32 lanes, one
// key each, hits 32 banks).  A wave owns query blocks of 32 and works on TRANSPOSED tiles so that the query sits on the lane:
//   S^T[key, q] = K_j (A operand: lane = key row) x Q_i^T (B operand: lane = query column, the lane's half of the head dims in 32
//   registers).  In the 32x32 accumulator lane (q, h) then holds the 16 keys 8g + 4h + c (register 4g + c): the softmax statistics
//   of a query are two lanes (h = 0 / 1) wide -- one cross-lane exchange per tile -- and the same 16 registers ARE the B operand
//   of the second product, O^T[d, q] += V_j^T (A: lane = head dim, row = the key the B lane holds at this step) x P^T: no transposes.
typedef float mf32x16 __attribute__((ext_vector_type(16)));
constexpr int MF_KP = 65;   // LDS row pitch in floats
constexpr int MF_THREADS = 512, MF_WAVES = MF_THREADS / 64;   // two waves per SIMD: one's softmax under the other's MFMAs; 7 (5) blocks over 8 waves

__device__ __forceinline__ mf32x16 mfma_f32(float a, float b, mf32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

template <bool DROP>
__global__ __launch_bounds__(MF_THREADS) void attn_fwd_mf32_kernel(const float* __restrict__ qkv, int ld, int S, int heads,
                                                                   const float* __restrict__ key_bias, float scale,
                                                                   float* __restrict__ ctx, int ld_ctx, float* __restrict__ lse,
                                                                   DropCfg drop) {
    BSCLIP_DROP_RESOLVE(drop);
    __shared__ float sK[AF_SMAX * MF_KP];
    __shared__ float sV[AF_SMAX * MF_KP];
    __shared__ float sBias[AF_SMAX];
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, ln = lane & 31, hf = lane >> 5;
    const int HW = heads * 64;
    const float* qb = qkv + (size_t)b * S * ld + hd * 64;
    const int NB = (S + 31) / 32, SP = NB * 32;
    for (int i = tid; i < SP * 64; i += MF_THREADS) {
        const int r = i >> 6, c = i & 63;
        sK[r * MF_KP + c] = r < S ? qb[(size_t)r * ld + HW + c] : 0.f;
        sV[r * MF_KP + c] = r < S ? qb[(size_t)r * ld + 2 * HW + c] : 0.f;
    }
    for (int k = tid; k < SP; k += MF_THREADS) sBias[k] = k < S ? (key_bias ? key_bias[(size_t)b * S + k] : 0.f) : -INFINITY;
    __syncthreads();
    const unsigned dhead = (unsigned)(b * heads + hd) * S;
    for (int qi = wave; qi < NB; qi += MF_WAVES) {
        const int q = qi * 32 + ln, qc = min(q, S - 1);
        float qreg[32];   // Q[q][2 s + hf]: the B operand of step s
#pragma unroll
        for (int s2 = 0; s2 < 32; ++s2) qreg[s2] = qb[(size_t)qc * ld + 2 * s2 + hf];
        mf32x16 o0 = {0}, o1 = {0};
        float m = -INFINITY, lsum = 0.f;
        const unsigned dbase = (dhead + (unsigned)qc) * SP;
        for (int kj = 0; kj < NB; ++kj) {
            mf32x16 acc = {0};
            const float* kp = sK + (kj * 32 + ln) * MF_KP + hf;
#pragma unroll
            for (int s2 = 0; s2 < 32; ++s2) acc = mfma_f32(kp[2 * s2], qreg[s2], acc);
            float sv[16], mb = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kj * 32 + 8 * (r >> 2) + 4 * hf + (r & 3);
                sv[r] = fmaf(acc[r], scale, sBias[key]);
                mb = fmaxf(mb, sv[r]);
            }
            mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
            const float mn = fmaxf(m, mb);
            const float corr = __expf(m - mn);      // exp(-inf) = 0 on the first tile
            m = mn;
            lsum *= corr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o0[r] *= corr;
                o1[r] *= corr;
            }
            float pk[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __expf(sv[r] - m);
                lsum += p;                            // the softmax sum is taken before dropout (HF: dropout(softmax(.)))
                pk[r] = p;
                if constexpr (DROP) pk[r] = p * drop_factor(drop, dbase + (unsigned)(kj * 32 + 8 * (r >> 2) + 4 * hf + (r & 3)));
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) {            // contraction index (t, hf) = key 8 (t / 4) + 4 hf + t % 4: the key pk[t] belongs to
                const float* vp = sV + (kj * 32 + 8 * (t >> 2) + 4 * hf + (t & 3)) * MF_KP + ln;
                o0 = mfma_f32(vp[0], pk[t], o0);
                o1 = mfma_f32(vp[32], pk[t], o1);
            }
        }
        lsum += __shfl_xor(lsum, 32, 64);
        if (q < S) {
            const float inv = 1.0f / lsum;
            float* out = ctx + (size_t)(b * S + q) * ld_ctx + hd * 64;
#pragma unroll
            for (int g = 0; g < 4; ++g) {             // O^T[d = 8 g + 4 hf + c (+ 32)][q]
                *reinterpret_cast<f32x4*>(out + 8 * g + 4 * hf) = f32x4{o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv};
                *reinterpret_cast<f32x4*>(out + 32 + 8 * g + 4 * hf) =
                    f32x4{o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv};
            }
            if (hf == 0) lse[((size_t)b * heads + hd) * S + q] = m + __logf(lsum);
        }
    }
}

// f32 attention backward on the matrix pipe (see attn_fwd_mf32_kernel for the transposed-tile scheme), two phases over one LDS image:
//   Q phase  K, V resident; a wave owns query blocks, the query on the lane:  S^T = K Q^T, dP^T = V dO^T, dS^T = P^T (f dP^T - delta),
//            dQ^T += K^T dS^T  (the dS^T registers are the B operand).  delta = dO . O per query, kept for the second phase.
//   K phase  Q, dO resident; a wave owns key blocks, the key on the lane:     S = Q K^T, dP = dO V^T, dV^T += dO^T (P f), dK^T += Q^T dS.
// 96 + 128 MFMAs of 32x32x2 per pair of 32-blocks (the scores are formed once per phase: the two phases need them in transposed
// register layouts, and a 32x32 f32 transpose through LDS costs more than 32 MFMAs here).
template <bool DROP>
__global__ __launch_bounds__(MF_THREADS) void attn_bwd_mf32_kernel(const float* __restrict__ qkv, int ld, const float* __restrict__ dctx,
                                                                   int ld_d, const float* __restrict__ ctx, int ld_c,
                                                                   const float* __restrict__ lse, int S, int heads,
                                                                   const float* __restrict__ key_bias, float scale,
                                                                   float* __restrict__ dqkv, int ld_g, DropCfg drop) {
    BSCLIP_DROP_RESOLVE(drop);
    __shared__ float sA[AF_SMAX * MF_KP];   // K, then Q
    __shared__ float sB[AF_SMAX * MF_KP];   // V, then dO
    __shared__ float sBias[AF_SMAX], sLse[AF_SMAX], sDelta[AF_SMAX];
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, ln = lane & 31, hf = lane >> 5;
    const int HW = heads * 64;
    const float* qb = qkv + (size_t)b * S * ld + hd * 64;
    const float* db = dctx + (size_t)b * S * ld_d + hd * 64;
    const float* cb = ctx + (size_t)b * S * ld_c + hd * 64;
    float* gb = dqkv + (size_t)b * S * ld_g + hd * 64;
    const int NB = (S + 31) / 32, SP = NB * 32;
    const unsigned dhead = (unsigned)(b * heads + hd) * S;
    for (int i = tid; i < SP * 64; i += MF_THREADS) {
        const int r = i >> 6, c = i & 63;
        sA[r * MF_KP + c] = r < S ? qb[(size_t)r * ld + HW + c] : 0.f;
        sB[r * MF_KP + c] = r < S ? qb[(size_t)r * ld + 2 * HW + c] : 0.f;
    }
    for (int k = tid; k < SP; k += MF_THREADS) {
        sBias[k] = k < S ? (key_bias ? key_bias[(size_t)b * S + k] : 0.f) : -INFINITY;
        sLse[k] = k < S ? lse[((size_t)b * heads + hd) * S + k] : INFINITY;    // a padded query row: p = exp(. - inf) = 0
    }
    __syncthreads();
    // ---------------------------------------------------------------- Q phase
    for (int qi = wave; qi < NB; qi += MF_WAVES) {
        const int q = qi * 32 + ln, qc = min(q, S - 1);
        float qreg[32], doreg[32], dpart = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 32; ++s2) {
            qreg[s2] = qb[(size_t)qc * ld + 2 * s2 + hf];
            doreg[s2] = db[(size_t)qc * ld_d + 2 * s2 + hf];
            dpart = fmaf(doreg[s2], cb[(size_t)qc * ld_c + 2 * s2 + hf], dpart);
        }
        const float delta = dpart + __shfl_xor(dpart, 32, 64);
        const float lq = sLse[q];
        if (hf == 0) sDelta[q] = q < S ? delta : 0.f;
        mf32x16 g0 = {0}, g1 = {0};
        const unsigned dbase = (dhead + (unsigned)qc) * SP;
        for (int kj = 0; kj < NB; ++kj) {
            mf32x16 acc = {0}, acp = {0};
            const float* kp = sA + (kj * 32 + ln) * MF_KP + hf;
            const float* vp = sB + (kj * 32 + ln) * MF_KP + hf;
#pragma unroll
            for (int s2 = 0; s2 < 32; ++s2) {
                acc = mfma_f32(kp[2 * s2], qreg[s2], acc);
                acp = mfma_f32(vp[2 * s2], doreg[s2], acp);
            }
            float ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kj * 32 + 8 * (r >> 2) + 4 * hf + (r & 3);
                const float p = __expf(fmaf(acc[r], scale, sBias[key]) - lq);
                float dp = acp[r];
                if constexpr (DROP) dp *= drop_factor(drop, dbase + (unsigned)key);
                ds[r] = p * (dp - delta);
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float* kt = sA + (kj * 32 + 8 * (t >> 2) + 4 * hf + (t & 3)) * MF_KP + ln;
                g0 = mfma_f32(kt[0], ds[t], g0);
                g1 = mfma_f32(kt[32], ds[t], g1);
            }
        }
        if (q < S) {
            float* out = gb + (size_t)q * ld_g;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                *reinterpret_cast<f32x4*>(out + 8 * g + 4 * hf) =
                    f32x4{g0[4 * g] * scale, g0[4 * g + 1] * scale, g0[4 * g + 2] * scale, g0[4 * g + 3] * scale};
                *reinterpret_cast<f32x4*>(out + 32 + 8 * g + 4 * hf) =
                    f32x4{g1[4 * g] * scale, g1[4 * g + 1] * scale, g1[4 * g + 2] * scale, g1[4 * g + 3] * scale};
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < SP * 64; i += MF_THREADS) {
        const int r = i >> 6, c = i & 63;
        sA[r * MF_KP + c] = r < S ? qb[(size_t)r * ld + c] : 0.f;
        sB[r * MF_KP + c] = r < S ? db[(size_t)r * ld_d + c] : 0.f;
    }
    __syncthreads();
    // ---------------------------------------------------------------- K phase
    for (int kj = wave; kj < NB; kj += MF_WAVES) {
        const int k = kj * 32 + ln, kc = min(k, S - 1);
        float kreg[32], vreg[32];
#pragma unroll
        for (int s2 = 0; s2 < 32; ++s2) {
            kreg[s2] = qb[(size_t)kc * ld + HW + 2 * s2 + hf];
            vreg[s2] = qb[(size_t)kc * ld + 2 * HW + 2 * s2 + hf];
        }
        const float bias = sBias[k];
        mf32x16 k0 = {0}, k1 = {0}, v0 = {0}, v1 = {0};
        for (int qi = 0; qi < NB; ++qi) {
            mf32x16 acc = {0}, acp = {0};
            const float* qp = sA + (qi * 32 + ln) * MF_KP + hf;
            const float* dp_ = sB + (qi * 32 + ln) * MF_KP + hf;
#pragma unroll
            for (int s2 = 0; s2 < 32; ++s2) {
                acc = mfma_f32(qp[2 * s2], kreg[s2], acc);
                acp = mfma_f32(dp_[2 * s2], vreg[s2], acp);
            }
            float ds[16], pf[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qq = qi * 32 + 8 * (r >> 2) + 4 * hf + (r & 3);
                const float p = __expf(fmaf(acc[r], scale, bias) - sLse[qq]);
                float f = 1.0f;
                if constexpr (DROP) f = drop_factor(drop, (dhead + (unsigned)min(qq, S - 1)) * SP + (unsigned)kc);
                pf[r] = p * f;
                ds[r] = p * (f * acp[r] - sDelta[qq]);
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int row = (qi * 32 + 8 * (t >> 2) + 4 * hf + (t & 3)) * MF_KP + ln;
                v0 = mfma_f32(sB[row], pf[t], v0);
                v1 = mfma_f32(sB[row + 32], pf[t], v1);
                k0 = mfma_f32(sA[row], ds[t], k0);
                k1 = mfma_f32(sA[row + 32], ds[t], k1);
            }
        }
        if (k < S) {
            float* ok = gb + (size_t)k * ld_g + HW;
            float* ov = gb + (size_t)k * ld_g + 2 * HW;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                *reinterpret_cast<f32x4*>(ok + 8 * g + 4 * hf) =
                    f32x4{k0[4 * g] * scale, k0[4 * g + 1] * scale, k0[4 * g + 2] * scale, k0[4 * g + 3] * scale};
                *reinterpret_cast<f32x4*>(ok + 32 + 8 * g + 4 * hf) =
                    f32x4{k1[4 * g] * scale, k1[4 * g + 1] * scale, k1[4 * g + 2] * scale, k1[4 * g + 3] * scale};
                *reinterpret_cast<f32x4*>(ov + 8 * g + 4 * hf) = f32x4{v0[4 * g], v0[4 * g + 1], v0[4 * g + 2], v0[4 * g + 3]};
                *reinterpret_cast<f32x4*>(ov + 32 + 8 * g + 4 * hf) = f32x4{v1[4 * g], v1[4 * g + 1], v1[4 * g + 2], v1[4 * g + 3]};
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ exact backward
// dst = split of (dact * gelu'(z)) as [hi | lo | hi] (the A operand of fc1's dX GEMM) and / or the same product in f32
__global__ void dgelu_split3_kernel(const float* __restrict__ dact, int ldd_, const float* __restrict__ z, int ldz, int M, int N,
                                    bf16_t* __restrict__ dst, int ldd, float* __restrict__ out32, int ldo) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nq = N / 4;
    if (t >= (size_t)M * nq) return;
    const int m = (int)(t / nq), n = (int)(t % nq) * 4;
    const f32x4 x = *reinterpret_cast<const f32x4*>(z + (size_t)m * ldz + n);
    const f32x4 g = *reinterpret_cast<const f32x4*>(dact + (size_t)m * ldd_ + n);
    float d[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float cdf = 0.5f * (1.0f + erff(x[i] * 0.70710678118654752f));
        d[i] = g[i] * fmaf(x[i] * 0.39894228040143268f, __expf(-0.5f * x[i] * x[i]), cdf);
    }
    if (dst != nullptr) {
        uint2 hi, lo;
        split2(d[0], d[1], hi.x, lo.x);
        split2(d[2], d[3], hi.y, lo.y);
        bf16_t* row = dst + (size_t)m * ldd;
        *reinterpret_cast<uint2*>(row + n) = hi;
        *reinterpret_cast<uint2*>(row + N + n) = lo;
        *reinterpret_cast<uint2*>(row + 2 * N + n) = hi;
    }
    if (out32 != nullptr) *reinterpret_cast<f32x4*>(out32 + (size_t)m * ldo + n) = f32x4{d[0], d[1], d[2], d[3]};
}

// dst[c, :] = split of column c of src [R, C] over Rp >= R positions (zeros beyond R): the reduction dimension of a dW = dY^T X
// GEMM (both operands transposed, order 0 = [hi | lo | hi], 1 = [hi | hi | lo]) or a frozen weight's transpose for a dX GEMM
// (order 1; LoRA folded as in split3_weight_kernel when lora_a is given: src is then the [3H, H] QKV weight).
// One workgroup: a 64 x 64 tile through LDS.
__global__ __launch_bounds__(256) void split3_transpose_kernel(const float* __restrict__ src, int ld, int R, int C, int Rp, int order,
                                                               const float* __restrict__ lora_a, const float* __restrict__ lora_b,
                                                               int H, bf16_t* __restrict__ dst, int ldd) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tid = threadIdx.x;
    for (int i = tid; i < 64 * 64; i += 256) {
        const int r = r0 + (i >> 6), c = c0 + (i & 63);
        float v = 0.f;
        if (r < R && c < C) {
            v = src[(size_t)r * ld + c];
            if (lora_a != nullptr && (r < H || r >= 2 * H)) {
                const int part = r < H ? 0 : 1, rr = r < H ? r : r - 2 * H;
#pragma unroll
                for (int j = 0; j < 4; ++j) v = fmaf(lora_b[((size_t)part * H + rr) * 4 + j], lora_a[(size_t)(4 * part + j) * C + c], v);
            }
        }
        tile[i >> 6][i & 63] = v;
    }
    __syncthreads();
    const int c = c0 + (tid >> 2), rs = (tid & 3) * 16;
    if (c >= C) return;
    ex_u32x4 hi[2], lo[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        unsigned h, l;
        split2(tile[rs + 2 * j][tid >> 2], tile[rs + 2 * j + 1][tid >> 2], h, l);
        hi[j >> 2][j & 3] = h;
        lo[j >> 2][j & 3] = l;
    }
    bf16_t* row = dst + (size_t)c * ldd + r0 + rs;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        *reinterpret_cast<ex_u32x4*>(row + 8 * j) = hi[j];
        *reinterpret_cast<ex_u32x4*>(row + Rp + 8 * j) = order ? hi[j] : lo[j];
        *reinterpret_cast<ex_u32x4*>(row + 2 * Rp + 8 * j) = order ? lo[j] : hi[j];
    }
}

// dlogits[b,t,c] = p_c (g_c - sum_j p_j g_j),  g = d_pooled[b] / S, in f32 (heads.hip's kernel rounds to bf16)
template <int C>
__global__ __launch_bounds__(256) void softmax_meanpool_bwd_f32_kernel(const float* __restrict__ logits, const float* __restrict__ stats,
                                                                        const float* __restrict__ d_pooled, int M, int S,
                                                                        float* __restrict__ dlogits, int ld_d) {
    constexpr int NV = C / 256;
    const int lane = threadIdx.x & 63;
    const int row = (blockIdx.x * 256 + threadIdx.x) >> 6;
    if (row >= M) return;
    const int b = row / S;
    const float m = stats[2 * (size_t)row], inv = 1.0f / stats[2 * (size_t)row + 1];
    const float invS = 1.0f / (float)S;
    f32x4 p[NV], g[NV];
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(logits + (size_t)row * C + j * 256 + lane * 4);
        g[j] = *reinterpret_cast<const f32x4*>(d_pooled + (size_t)b * C + j * 256 + lane * 4) * invS;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            p[j][i] = __expf(v[i] - m) * inv;
            dot += p[j][i] * g[j][i];
        }
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int j = 0; j < NV; ++j)
        *reinterpret_cast<f32x4*>(dlogits + (size_t)row * ld_d + j * 256 + lane * 4) = p[j] * (g[j] - dot);
}

// f32 attention backward on the vector ALU, one workgroup per (batch, head), two phases over the same LDS:
//   A  K, V resident; one query row per thread: delta = dO . O, p = exp(s - lse), dS = p (f dO.v - delta), dQ = scale dS K
//   B  Q, dO resident; one key row per thread:  dV = (p f)^T dO, dK = scale dS^T Q
// (f = the dropout factor of the probability, the same (seed, element) hash as every other attention kernel).
template <bool DROP>
__global__ __launch_bounds__(AF_THREADS) void attn_bwd_f32_kernel(const float* __restrict__ qkv, int ld, const float* __restrict__ dctx,
                                                                  int ld_d, const float* __restrict__ ctx, int ld_c,
                                                                  const float* __restrict__ lse, int S, int heads,
                                                                  const float* __restrict__ key_bias, float scale,
                                                                  float* __restrict__ dqkv, int ld_g, DropCfg drop) {
    BSCLIP_DROP_RESOLVE(drop);
    __shared__ __attribute__((aligned(16))) float sA[AF_SMAX * 64];
    __shared__ __attribute__((aligned(16))) float sB[AF_SMAX * 64];
    __shared__ float sBias[AF_SMAX], sLse[AF_SMAX], sDelta[AF_SMAX];
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads, tid = threadIdx.x;
    const int HW = heads * 64;
    const float* qb = qkv + (size_t)b * S * ld + hd * 64;
    const float* db = dctx + (size_t)b * S * ld_d + hd * 64;
    const float* cb = ctx + (size_t)b * S * ld_c + hd * 64;
    float* gb = dqkv + (size_t)b * S * ld_g + hd * 64;
    const int SP = (S + 31) / 32 * 32;
    const unsigned dhead = (unsigned)(b * heads + hd) * S;
    for (int i = tid; i < S * 16; i += AF_THREADS) {
        const int r = i >> 4, c = (i & 15) * 4;
        *reinterpret_cast<f32x4*>(sA + r * 64 + c) = *reinterpret_cast<const f32x4*>(qb + (size_t)r * ld + HW + c);
        *reinterpret_cast<f32x4*>(sB + r * 64 + c) = *reinterpret_cast<const f32x4*>(qb + (size_t)r * ld + 2 * HW + c);
    }
    for (int k = tid; k < S; k += AF_THREADS) {
        sBias[k] = key_bias ? key_bias[(size_t)b * S + k] : 0.f;
        sLse[k] = lse[((size_t)b * heads + hd) * S + k];
    }
    __syncthreads();
    for (int q = tid; q < S; q += AF_THREADS) {
        float qv[64], dov[64], dq[64];
        float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(qb + (size_t)q * ld + 4 * c);
            const f32x4 g = *reinterpret_cast<const f32x4*>(db + (size_t)q * ld_d + 4 * c);
            const f32x4 o = *reinterpret_cast<const f32x4*>(cb + (size_t)q * ld_c + 4 * c);
            d0 = fmaf(g[0], o[0], d0);
            d1 = fmaf(g[1], o[1], d1);
            d2 = fmaf(g[2], o[2], d2);
            d3 = fmaf(g[3], o[3], d3);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                qv[4 * c + i] = v[i];
                dov[4 * c + i] = g[i];
                dq[4 * c + i] = 0.f;
            }
        }
        const float delta = (d0 + d1) + (d2 + d3), lq = sLse[q];
        sDelta[q] = delta;
        const unsigned dbase = (dhead + (unsigned)q) * SP;
        for (int k = 0; k < S; ++k) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const f32x4 kv = *reinterpret_cast<const f32x4*>(sA + k * 64 + 4 * c);
                const f32x4 vv = *reinterpret_cast<const f32x4*>(sB + k * 64 + 4 * c);
                s0 = fmaf(qv[4 * c + 0], kv[0], s0);
                s1 = fmaf(qv[4 * c + 1], kv[1], s1);
                s2 = fmaf(qv[4 * c + 2], kv[2], s2);
                s3 = fmaf(qv[4 * c + 3], kv[3], s3);
                p0 = fmaf(dov[4 * c + 0], vv[0], p0);
                p1 = fmaf(dov[4 * c + 1], vv[1], p1);
                p2 = fmaf(dov[4 * c + 2], vv[2], p2);
                p3 = fmaf(dov[4 * c + 3], vv[3], p3);
            }
            const float s = fmaf((s0 + s1) + (s2 + s3), scale, sBias[k]);
            const float p = __expf(s - lq);
            float dp = (p0 + p1) + (p2 + p3);
            if constexpr (DROP) dp *= drop_factor(drop, dbase + (unsigned)k);
            const float ds = p * (dp - delta);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const f32x4 kv = *reinterpret_cast<const f32x4*>(sA + k * 64 + 4 * c);
#pragma unroll
                for (int i = 0; i < 4; ++i) dq[4 * c + i] = fmaf(ds, kv[i], dq[4 * c + i]);
            }
        }
#pragma unroll
        for (int c = 0; c < 16; ++c)
            *reinterpret_cast<f32x4*>(gb + (size_t)q * ld_g + 4 * c) =
                f32x4{dq[4 * c] * scale, dq[4 * c + 1] * scale, dq[4 * c + 2] * scale, dq[4 * c + 3] * scale};
    }
    __syncthreads();
    for (int i = tid; i < S * 16; i += AF_THREADS) {
        const int r = i >> 4, c = (i & 15) * 4;
        *reinterpret_cast<f32x4*>(sA + r * 64 + c) = *reinterpret_cast<const f32x4*>(qb + (size_t)r * ld + c);
        *reinterpret_cast<f32x4*>(sB + r * 64 + c) = *reinterpret_cast<const f32x4*>(db + (size_t)r * ld_d + c);
    }
    __syncthreads();
    for (int k = tid; k < S; k += AF_THREADS) {
        float kv[64], vv[64], dk[64], dv[64];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(qb + (size_t)k * ld + HW + 4 * c);
            const f32x4 v = *reinterpret_cast<const f32x4*>(qb + (size_t)k * ld + 2 * HW + 4 * c);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                kv[4 * c + i] = a[i];
                vv[4 * c + i] = v[i];
                dk[4 * c + i] = 0.f;
                dv[4 * c + i] = 0.f;
            }
        }
        const float bias = sBias[k];
        for (int q = 0; q < S; ++q) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const f32x4 qv = *reinterpret_cast<const f32x4*>(sA + q * 64 + 4 * c);
                const f32x4 g = *reinterpret_cast<const f32x4*>(sB + q * 64 + 4 * c);
                s0 = fmaf(qv[0], kv[4 * c + 0], s0);
                s1 = fmaf(qv[1], kv[4 * c + 1], s1);
                s2 = fmaf(qv[2], kv[4 * c + 2], s2);
                s3 = fmaf(qv[3], kv[4 * c + 3], s3);
                p0 = fmaf(g[0], vv[4 * c + 0], p0);
                p1 = fmaf(g[1], vv[4 * c + 1], p1);
                p2 = fmaf(g[2], vv[4 * c + 2], p2);
                p3 = fmaf(g[3], vv[4 * c + 3], p3);
            }
            const float s = fmaf((s0 + s1) + (s2 + s3), scale, bias);
            const float p = __expf(s - sLse[q]);
            float f = 1.0f;
            if constexpr (DROP) f = drop_factor(drop, (dhead + (unsigned)q) * SP + (unsigned)k);
            const float pf = p * f;
            const float ds = p * (f * ((p0 + p1) + (p2 + p3)) - sDelta[q]);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const f32x4 qv = *reinterpret_cast<const f32x4*>(sA + q * 64 + 4 * c);
                const f32x4 g = *reinterpret_cast<const f32x4*>(sB + q * 64 + 4 * c);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    dk[4 * c + i] = fmaf(ds, qv[i], dk[4 * c + i]);
                    dv[4 * c + i] = fmaf(pf, g[i], dv[4 * c + i]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            *reinterpret_cast<f32x4*>(gb + (size_t)k * ld_g + HW + 4 * c) =
                f32x4{dk[4 * c] * scale, dk[4 * c + 1] * scale, dk[4 * c + 2] * scale, dk[4 * c + 3] * scale};
            *reinterpret_cast<f32x4*>(gb + (size_t)k * ld_g + 2 * HW + 4 * c) = f32x4{dv[4 * c], dv[4 * c + 1], dv[4 * c + 2], dv[4 * c + 3]};
        }
    }
}

}  // namespace

static int g_exact_attn_impl = 0;
extern "C" int bsclip_exact_attn_set_impl(int impl) {
    BSCLIP_REQUIRE(impl >= 0 && impl <= 2, "bsclip_exact_attn_set_impl: %d (0 = split-bf16 MFMA, 1 = vector ALU, 2 = f32 MFMA)", impl);
    g_exact_attn_impl = impl;
    return BSCLIP_OK;
}

extern "C" int bsclip_split3_rows(const float* src, int ld_src, int M, int K, void* dst, int ld_dst, void* stream) {
    BSCLIP_REQUIRE(src && dst, "bsclip_split3_rows: null pointer");
    BSCLIP_REQUIRE((reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 7) == 0,
                   "bsclip_split3_rows: src must be 16-byte, dst 8-byte aligned");
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
    // the kernel reads w, lora_a and lora_b as f32x4 and writes dst as 8-byte words: views into a flat parameter buffer are aligned
    // only while every parameter's numel is a multiple of 4 -- a misaligned view must be an error code, not a GPU fault
    BSCLIP_REQUIRE(((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(lora_a) | reinterpret_cast<uintptr_t>(lora_b)) & 15) == 0 &&
                       (reinterpret_cast<uintptr_t>(dst) & 7) == 0,
                   "bsclip_split3_weight: w / lora_a / lora_b must be 16-byte, dst 8-byte aligned");
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
    BSCLIP_REQUIRE(((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(g32)) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 7) == 0 &&
                       (reinterpret_cast<uintptr_t>(codes) & 3) == 0,
                   "bsclip_gelu_split3: z / g32 must be 16-byte, dst 8-byte, codes 4-byte aligned");
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

// ctx_split3 / dqkv_split3 (nullable, round 5): the output once more as the A operand of the next split-bf16 GEMM ([hi | lo | hi], bf16
// [B S, ld >= 3 x width]) -- written by the split-bf16 kernels themselves (impl 0), by a split pass behind the f32 kernels (impl 1 / 2)
static void split3_after(const float* src, int ld_src, int M, int K, void* dst, int ld_dst, hipStream_t s) {
    const size_t n = (size_t)M * (K / 4);
    hipLaunchKernelGGL(split3_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, ld_src, M, K, static_cast<bf16_t*>(dst), ld_dst);
}

extern "C" int bsclip_attn_fwd_f32(const float* qkv, int ld_qkv, int B, int S, int heads, const float* key_bias, float scale, float* ctx,
                                   int ld_ctx, float* lse, void* ctx_split3, int ld_c3, float dropout_p, uint32_t dropout_seed, void* stream) {
    BSCLIP_REQUIRE(qkv && ctx && lse, "bsclip_attn_fwd_f32: null pointer");
    BSCLIP_REQUIRE(!ctx_split3 || (ld_c3 >= 3 * heads * 64 && ld_c3 % 4 == 0 && (reinterpret_cast<uintptr_t>(ctx_split3) & 7) == 0),
                   "bsclip_attn_fwd_f32: ctx_split3 bf16 [B S, ld_c3 >= 3 heads 64], 8-byte aligned (ld_c3=%d)", ld_c3);
    BSCLIP_REQUIRE(B > 0 && heads > 0 && S > 0 && S <= AF_SMAX, "bsclip_attn_fwd_f32: B=%d heads=%d S=%d (S <= 224)", B, heads, S);
    BSCLIP_REQUIRE(ld_qkv >= 3 * heads * 64 && ld_qkv % 4 == 0 && ld_ctx >= heads * 64 && ld_ctx % 4 == 0,
                   "bsclip_attn_fwd_f32: ld_qkv=%d ld_ctx=%d", ld_qkv, ld_ctx);
    BSCLIP_REQUIRE(((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(ctx)) & 15) == 0,
                   "bsclip_attn_fwd_f32: qkv and ctx must be 16-byte aligned (vector loads / stores)");
    BSCLIP_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "bsclip_attn_fwd_f32: dropout_p=%f", dropout_p);
    const DropCfg drop = make_drop(dropout_p, dropout_seed);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (g_exact_attn_impl == 0) {   // round 5: split-bf16 operands on the bf16 matrix cores (attn_x3.hip), 16 x the f32 MFMA rate
        bsclip_launch_attn_fwd_x3(qkv, ld_qkv, B, S, heads, key_bias, scale, ctx, ld_ctx, lse, drop, static_cast<bf16_t*>(ctx_split3), ld_c3, s);
        BSCLIP_LAUNCH_CHECK();
        return BSCLIP_OK;
    }
    if (g_exact_attn_impl == 2) {   // f32 operands on the matrix pipe (round 4); impl 1 = the one-row-per-thread vector-ALU kernel
        if (drop.thr16)
            hipLaunchKernelGGL((attn_fwd_mf32_kernel<true>), dim3(B * heads), dim3(MF_THREADS), 0, s, qkv, ld_qkv, S, heads, key_bias, scale,
                               ctx, ld_ctx, lse, drop);
        else
            hipLaunchKernelGGL((attn_fwd_mf32_kernel<false>), dim3(B * heads), dim3(MF_THREADS), 0, s, qkv, ld_qkv, S, heads, key_bias, scale,
                               ctx, ld_ctx, lse, drop);
        if (ctx_split3) split3_after(ctx, ld_ctx, B * S, heads * 64, ctx_split3, ld_c3, s);
        BSCLIP_LAUNCH_CHECK();
        return BSCLIP_OK;
    }
    if (drop.thr16)
        hipLaunchKernelGGL((attn_fwd_f32_kernel<true>), dim3(B * heads), dim3(AF_THREADS), 0, s, qkv, ld_qkv, S, heads, key_bias, scale, ctx,
                           ld_ctx, lse, drop);
    else
        hipLaunchKernelGGL((attn_fwd_f32_kernel<false>), dim3(B * heads), dim3(AF_THREADS), 0, s, qkv, ld_qkv, S, heads, key_bias, scale, ctx,
                           ld_ctx, lse, drop);
    if (ctx_split3) split3_after(ctx, ld_ctx, B * S, heads * 64, ctx_split3, ld_c3, s);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int bsclip_dgelu_split3(const float* dact, int ld_dact, const float* z, int ld_z, int M, int N, void* dst, int ld_dst,
                                   float* out32, int ld_out32, void* stream) {
    BSCLIP_REQUIRE(dact && z && (dst || out32), "bsclip_dgelu_split3: null pointer");
    BSCLIP_REQUIRE(((reinterpret_cast<uintptr_t>(dact) | reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(out32)) & 15) == 0 &&
                       (reinterpret_cast<uintptr_t>(dst) & 7) == 0,
                   "bsclip_dgelu_split3: dact / z / out32 must be 16-byte, dst 8-byte aligned");
    BSCLIP_REQUIRE(M > 0 && N > 0 && N % 4 == 0 && ld_dact >= N && ld_dact % 4 == 0 && ld_z >= N && ld_z % 4 == 0 &&
                       (dst == nullptr || (ld_dst >= 3 * N && ld_dst % 4 == 0)) && (out32 == nullptr || (ld_out32 >= N && ld_out32 % 4 == 0)),
                   "bsclip_dgelu_split3: M=%d N=%d ld_dact=%d ld_z=%d ld_dst=%d ld_out32=%d", M, N, ld_dact, ld_z, ld_dst, ld_out32);
    const size_t n = (size_t)M * (N / 4);
    hipLaunchKernelGGL(dgelu_split3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), dact, ld_dact,
                       z, ld_z, M, N, static_cast<bf16_t*>(dst), ld_dst, out32, ld_out32);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int bsclip_split3_transpose(const float* src, int ld_src, int R, int C, int Rp, int order, const float* lora_a,
                                       const float* lora_b, int H, void* dst, int ld_dst, void* stream) {
    BSCLIP_REQUIRE(src && dst, "bsclip_split3_transpose: null pointer");
    BSCLIP_REQUIRE((lora_a == nullptr) == (lora_b == nullptr), "bsclip_split3_transpose: lora_a and lora_b go together");
    BSCLIP_REQUIRE(R > 0 && C > 0 && Rp >= R && Rp % 64 == 0 && ld_src >= C && ld_dst >= 3 * Rp && ld_dst % 8 == 0 && (order == 0 || order == 1) &&
                       (lora_a == nullptr || (R == 3 * H && C == H)) && (reinterpret_cast<uintptr_t>(dst) & 15) == 0,
                   "bsclip_split3_transpose: R=%d C=%d Rp=%d order=%d H=%d ld_src=%d ld_dst=%d", R, C, Rp, order, H, ld_src, ld_dst);
    hipLaunchKernelGGL(split3_transpose_kernel, dim3(Rp / 64, (C + 63) / 64), dim3(256), 0, static_cast<hipStream_t>(stream), src, ld_src, R,
                       C, Rp, order, lora_a, lora_b, H, static_cast<bf16_t*>(dst), ld_dst);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int bsclip_softmax_meanpool_bwd_f32(const float* logits, const float* stats, const float* d_pooled, int B, int S, int C,
                                               float* dlogits, int ld_d, void* stream) {
    BSCLIP_REQUIRE(logits && stats && d_pooled && dlogits && B > 0 && S > 0, "bsclip_softmax_meanpool_bwd_f32: bad args");
    BSCLIP_REQUIRE(C == 768 && ld_d >= C && ld_d % 4 == 0, "bsclip_softmax_meanpool_bwd_f32: C=%d ld_d=%d", C, ld_d);
    const int M = B * S;
    hipLaunchKernelGGL((softmax_meanpool_bwd_f32_kernel<768>), dim3(ceil_div(M, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), logits,
                       stats, d_pooled, M, S, dlogits, ld_d);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int bsclip_attn_bwd_f32(const float* qkv, int ld_qkv, const float* dctx, int ld_dctx, const float* ctx, int ld_ctx,
                                   const float* lse, int B, int S, int heads, const float* key_bias, float scale, float* dqkv, int ld_dqkv,
                                   void* dqkv_split3, int ld_d3, float dropout_p, uint32_t dropout_seed, void* stream) {
    BSCLIP_REQUIRE(qkv && dctx && ctx && lse && dqkv, "bsclip_attn_bwd_f32: null pointer");
    BSCLIP_REQUIRE(!dqkv_split3 || (ld_d3 >= 9 * heads * 64 && ld_d3 % 4 == 0 && (reinterpret_cast<uintptr_t>(dqkv_split3) & 7) == 0),
                   "bsclip_attn_bwd_f32: dqkv_split3 bf16 [B S, ld_d3 >= 9 heads 64], 8-byte aligned (ld_d3=%d)", ld_d3);
    BSCLIP_REQUIRE(B > 0 && heads > 0 && S > 0 && S <= AF_SMAX, "bsclip_attn_bwd_f32: B=%d heads=%d S=%d (S <= 224)", B, heads, S);
    BSCLIP_REQUIRE(ld_qkv >= 3 * heads * 64 && ld_qkv % 4 == 0 && ld_dqkv >= 3 * heads * 64 && ld_dqkv % 4 == 0 && ld_ctx >= heads * 64 &&
                       ld_ctx % 4 == 0 && ld_dctx >= heads * 64 && ld_dctx % 4 == 0,
                   "bsclip_attn_bwd_f32: ld_qkv=%d ld_dqkv=%d ld_ctx=%d ld_dctx=%d", ld_qkv, ld_dqkv, ld_ctx, ld_dctx);
    BSCLIP_REQUIRE(((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(ctx) | reinterpret_cast<uintptr_t>(dctx) |
                     reinterpret_cast<uintptr_t>(dqkv)) & 15) == 0,
                   "bsclip_attn_bwd_f32: qkv, ctx, dctx and dqkv must be 16-byte aligned (vector loads / stores)");
    BSCLIP_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "bsclip_attn_bwd_f32: dropout_p=%f", dropout_p);
    const DropCfg drop = make_drop(dropout_p, dropout_seed);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (g_exact_attn_impl == 0) {
        bsclip_launch_attn_bwd_x3(qkv, ld_qkv, dctx, ld_dctx, ctx, ld_ctx, lse, B, S, heads, key_bias, scale, dqkv, ld_dqkv, drop,
                                  static_cast<bf16_t*>(dqkv_split3), ld_d3, s);
        BSCLIP_LAUNCH_CHECK();
        return BSCLIP_OK;
    }
    if (g_exact_attn_impl == 2) {
        if (drop.thr16)
            hipLaunchKernelGGL((attn_bwd_mf32_kernel<true>), dim3(B * heads), dim3(MF_THREADS), 0, s, qkv, ld_qkv, dctx, ld_dctx, ctx, ld_ctx,
                               lse, S, heads, key_bias, scale, dqkv, ld_dqkv, drop);
        else
            hipLaunchKernelGGL((attn_bwd_mf32_kernel<false>), dim3(B * heads), dim3(MF_THREADS), 0, s, qkv, ld_qkv, dctx, ld_dctx, ctx, ld_ctx,
                               lse, S, heads, key_bias, scale, dqkv, ld_dqkv, drop);
        if (dqkv_split3) split3_after(dqkv, ld_dqkv, B * S, 3 * heads * 64, dqkv_split3, ld_d3, s);
        BSCLIP_LAUNCH_CHECK();
        return BSCLIP_OK;
    }
    if (drop.thr16)
        hipLaunchKernelGGL((attn_bwd_f32_kernel<true>), dim3(B * heads), dim3(AF_THREADS), 0, s, qkv, ld_qkv, dctx, ld_dctx, ctx, ld_ctx, lse,
                           S, heads, key_bias, scale, dqkv, ld_dqkv, drop);
    else
        hipLaunchKernelGGL((attn_bwd_f32_kernel<false>), dim3(B * heads), dim3(AF_THREADS), 0, s, qkv, ld_qkv, dctx, ld_dctx, ctx, ld_ctx, lse,
                           S, heads, key_bias, scale, dqkv, ld_dqkv, drop);
    if (dqkv_split3) split3_after(dqkv, ld_dqkv, B * S, 3 * heads * 64, dqkv_split3, ld_d3, s);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}
