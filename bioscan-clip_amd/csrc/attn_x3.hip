// Exact-mode attention on the bf16 matrix cores (round 5): f32 in, f32 out, every product formed on split operands.
//
// Replaces, for BSCLIP_PARITY=2, the f32-operand MFMA kernels of exact.hip (v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate: the
// f32 attention was 36 % of the exact mode's 158 ms) for the reference's timm Attention.forward / HF
// BertSelfAttention (image_encoder.py:108-109, dna_encoder.py:105, language_encoder.py:89) and their autograd.
//
// A bf16 MFMA product is exact to ~2^-16 when both operands are carried as hi + lo (hi = bf16(x), lo = bf16(x - hi)) and the product
// is formed as hi.hi + lo.hi + hi.lo -- what the exact mode's GEMMs do along K (exact.hip), done here inside the attention kernels:
//   * q, k, v, dO arrive as f32 and are split when they are staged (LDS tiles hi | lo, XOR-swizzled row-major as in attn.hip) or
//     fetched as fragments;
//   * the score / dP tiles take three MFMAs per k-step instead of one, f32 accumulation;
//   * the probabilities P and the score gradients dS leave the softmax arithmetic in f32 and are split in registers (the accumulator
//     tile IS the next product's operand, as in attn.hip): P_hi.V_hi + P_hi.V_lo + P_lo.V_hi and likewise for dV, dK, dQ.
// Structure = attn.hip's (scores transposed: key on the accumulator rows, query on the lane; a wave owns 32-row blocks; the backward
// in a query-owner and a key-owner phase, nothing accumulated across waves), with four LDS tiles (112 KB at S = 197): one workgroup
// of eight waves per CU.  delta = dO . O comes from the f32 forward output (exact in f32: no cancellation to protect), so the
// query-owner phase is a single pass.  36 + 48 MFMAs per pair of 32-blocks (the bf16 kernel: 32), at 16 x the f32 MFMA rate.
// Dropout: the same (seed, element) hash and element indices as every other attention kernel.
#include "attn_common.h"

namespace {

constexpr int X3_WAVES = 8;

// 8 consecutive f32 -> the bf16x8 fragments hi = bf16(x), lo = bf16(x - hi)
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
    u32x4 h, l;
    h[0] = pack_bf2(a[0], a[1]);
    h[1] = pack_bf2(a[2], a[3]);
    h[2] = pack_bf2(b[0], b[1]);
    h[3] = pack_bf2(b[2], b[3]);
    l[0] = pack_bf2(a[0] - bf_lo(h[0]), a[1] - bf_hi(h[0]));
    l[1] = pack_bf2(a[2] - bf_lo(h[1]), a[3] - bf_hi(h[1]));
    l[2] = pack_bf2(b[0] - bf_lo(h[2]), b[1] - bf_hi(h[2]));
    l[3] = pack_bf2(b[2] - bf_lo(h[3]), b[3] - bf_hi(h[3]));
    hi = __builtin_bit_cast(bf16x8, h);
    lo = __builtin_bit_cast(bf16x8, l);
}

// registers [8 s2, 8 s2 + 8) of an f32 accumulator tile -> operand fragments hi / lo (element j = register 8 s2 + j, as pack8)
__device__ __forceinline__ void split_acc8(const f32x16& x, int s2, bf16x8& hi, bf16x8& lo) {
    split8(f32x4{x[8 * s2 + 0], x[8 * s2 + 1], x[8 * s2 + 2], x[8 * s2 + 3]},
           f32x4{x[8 * s2 + 4], x[8 * s2 + 5], x[8 * s2 + 6], x[8 * s2 + 7]}, hi, lo);
}

// acc += a.b with both operands split: hi.hi + lo.hi + hi.lo (the dropped lo.lo term is 2^-16 relative)
__device__ __forceinline__ f32x16 mfma3(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x16 c) {
    c = mfma32(ah, bh, c);
    c = mfma32(al, bh, c);
    return mfma32(ah, bl, c);
}

// Stage a [S][64] f32 matrix (row stride ld floats) as TWO row-major XOR-swizzled bf16 tiles of SP rows (hi, lo): through registers --
// the split is arithmetic, LDS-DMA cannot do it.  Rows >= S repeat row S - 1 (padded keys / queries carry probability 0).
template <int SP>
__device__ __forceinline__ void stage_split(const float* __restrict__ src, int ld, int S, char* hi_t, char* lo_t, int tid) {
    for (int idx = tid; idx < SP * 8; idx += X3_WAVES * 64) {
        const int row = idx >> 3, ch = idx & 7;
        const float* p = src + (size_t)min(row, S - 1) * ld + ch * 8;
        bf16x8 hi, lo;
        split8(*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4), hi, lo);
        *reinterpret_cast<bf16x8*>(hi_t + rm_off(row, ch)) = hi;
        *reinterpret_cast<bf16x8*>(lo_t + rm_off(row, ch)) = lo;
    }
}

// B-operand fragments (hi, lo) straight from global: row of a f32 [S][64] matrix, k = 16 ks + 8 (lane >> 5) + j
__device__ __forceinline__ void frag_global_split(const float* base, int ld, int row, int ks, int lane, bf16x8& hi, bf16x8& lo) {
    const float* p = base + (size_t)row * ld + ks * 16 + 8 * (lane >> 5);
    split8(*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4), hi, lo);
}

// store a [64 (d) x 32 (token on lane)] f32 result held as 2 accumulator tiles into out_row[d] (f32, 16-byte pieces)
__device__ __forceinline__ void store_dt_f32(const f32x16 (&acc)[2], float mul, float* out_row, int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(out_row + 32 * dt + 8 * g + 4 * h) =
                f32x4{acc[dt][4 * g + 0] * mul, acc[dt][4 * g + 1] * mul, acc[dt][4 * g + 2] * mul, acc[dt][4 * g + 3] * mul};
}

// the same result as the A operand of a split-bf16 GEMM: row3[c] = hi, row3[K + c] = lo, row3[2 K + c] = hi (c = column of the [M, K] matrix:
// the layout bsclip_split3_rows produces; hi = bf16(x), lo = bf16(x - hi)) -- the consumer GEMM needs no stand-alone split pass
__device__ __forceinline__ void store_dt_split3(const f32x16 (&acc)[2], float mul, bf16_t* row3, int K, int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float x0 = acc[dt][4 * g + 0] * mul, x1 = acc[dt][4 * g + 1] * mul, x2 = acc[dt][4 * g + 2] * mul, x3 = acc[dt][4 * g + 3] * mul;
            uint2 hi, lo;
            hi.x = pack_bf2(x0, x1);
            hi.y = pack_bf2(x2, x3);
            lo.x = pack_bf2(x0 - bf_lo(hi.x), x1 - bf_hi(hi.x));
            lo.y = pack_bf2(x2 - bf_lo(hi.y), x3 - bf_hi(hi.y));
            bf16_t* p = row3 + 32 * dt + 8 * g + 4 * h;
            *reinterpret_cast<uint2*>(p) = hi;
            *reinterpret_cast<uint2*>(p + K) = lo;
            *reinterpret_cast<uint2*>(p + 2 * K) = hi;
        }
}

template <int NB, bool DROP>
__global__ __launch_bounds__(X3_WAVES * 64, 2) void attn_fwd_x3_kernel(const float* __restrict__ qkv, int ld, int S, int heads,
                                                                     const float* __restrict__ key_bias, float scale,
                                                                     float* __restrict__ ctx, int ld_ctx, float* __restrict__ lse,
                                                                     DropCfg drop, bf16_t* __restrict__ ctx3, int ld_c3) {
    constexpr int SP = NB * 32, RM = SP * ROWB;
    BSCLIP_DROP_RESOLVE(drop);
    __shared__ __attribute__((aligned(16))) char smem[4 * RM + SP * 4];
    char *sKh = smem, *sKl = smem + RM, *sVh = smem + 2 * RM, *sVl = smem + 3 * RM;
    float* sBias = reinterpret_cast<float*>(smem + 4 * RM);

    const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = heads * 64;
    const float* qb = qkv + (size_t)b * S * ld + hd * 64;
    const float* kb = qb + HW;
    const float* vb = kb + HW;

    stage_split<SP>(kb, ld, S, sKh, sKl, tid);
    stage_split<SP>(vb, ld, S, sVh, sVl, tid);
    const float inv_scale = 1.0f / scale;   // the additive key bias rides in the accumulator's start value (attn.hip)
    for (int k = tid; k < SP; k += X3_WAVES * 64)
        sBias[k] = (k < S) ? (key_bias ? key_bias[(size_t)b * S + k] * inv_scale : 0.f) : -INFINITY;
    __syncthreads();
    const float scale2 = scale * LOG2E;

#pragma unroll 1
    for (int blk = wave; blk < NB; blk += X3_WAVES) {
        asm volatile("" ::: "memory");  // LDS tiles are loop-invariant: stop LICM from hoisting the fragment reads
        const int q0 = blk * 32;
        const int qrow = min(q0 + (lane & 31), S - 1);
        bf16x8 qh[4], ql[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) frag_global_split(qb, ld, qrow, ks, lane, qh[ks], ql[ks]);

        f32x16 p[NB];
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NB; ++kt) {
            f32x16 acc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bias = *reinterpret_cast<const f32x4*>(sBias + 32 * kt + 8 * g + 4 * h);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[4 * g + i] = bias[i];
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                acc = mfma3(frag_rm(sKh, 32 * kt, ks, lane), frag_rm(sKl, 32 * kt, ks, lane), qh[ks], ql[ks], acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[r]);
            p[kt] = acc;
        }
        m = fmaxf(m, __shfl_xor(m, 32, 64));  // raw-score units
        const float nm2 = -m * scale2;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NB; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(p[kt][r], scale2, nm2));
                p[kt][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 32, 64);
        if constexpr (DROP) {  // HF: dropout on the normalised probabilities (the row sum above is taken before it)
            const unsigned base = ((unsigned)(b * heads + hd) * S + (unsigned)qrow) * SP + 4 * h;
#pragma unroll
            for (int kt = 0; kt < NB; ++kt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 k4 = keep4(drop, base + 32 * kt + 8 * g);
#pragma unroll
                    for (int i = 0; i < 4; ++i) p[kt][4 * g + i] *= k4[i];
                }
        }

        // O^T[d, query] = sum_key V^T[d, key] P^T[key, query], both operands split
        f32x16 o[2] = {zero16(), zero16()};
#pragma unroll
        for (int kt = 0; kt < NB; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 ph, pl;
                split_acc8(p[kt], s2, ph, pl);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    o[dt] = mfma3(frag_tr(sVh, 32 * dt, 32 * kt + 16 * s2, lane), frag_tr(sVl, 32 * dt, 32 * kt + 16 * s2, lane), ph, pl, o[dt]);
            }

        const int q = q0 + (lane & 31);
        if (q < S) {
            store_dt_f32(o, 1.0f / sum, ctx + (size_t)(b * S + q) * ld_ctx + hd * 64, lane);
            if (ctx3 != nullptr) store_dt_split3(o, 1.0f / sum, ctx3 + (size_t)(b * S + q) * ld_c3 + hd * 64, HW, lane);   // the out-projection's operand
            if (h == 0) lse[((size_t)b * heads + hd) * S + q] = (__log2f(sum) - nm2) * LN2;  // natural-log LSE
        }
    }
}

template <int NB, bool DROP>
__global__ __launch_bounds__(X3_WAVES * 64, 2) void attn_bwd_x3_kernel(const float* __restrict__ qkv, int ld, const float* __restrict__ dctx,
                                                                     int ld_dc, const float* __restrict__ ctx, int ld_c,
                                                                     const float* __restrict__ lse, int S, int heads,
                                                                     const float* __restrict__ key_bias, float scale,
                                                                     float* __restrict__ dqkv, int ld_d, DropCfg drop,
                                                                     bf16_t* __restrict__ dqkv3, int ld_d3) {
    constexpr int SP = NB * 32, RM = SP * ROWB;
    BSCLIP_DROP_RESOLVE(drop);
    __shared__ __attribute__((aligned(16))) char smem[4 * RM + 3 * SP * 4];
    char *sAh = smem, *sAl = smem + RM;             // phase 1: K hi / lo | phase 2: Q hi / lo
    char *sBh = smem + 2 * RM, *sBl = smem + 3 * RM;   // phase 1: V hi / lo | phase 2: dO hi / lo
    float* sLse = reinterpret_cast<float*>(smem + 4 * RM);
    float* sDelta = sLse + SP;
    float* sBias = sDelta + SP;

    const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = heads * 64;
    const float* qb = qkv + (size_t)b * S * ld + hd * 64;
    const float* kb = qb + HW;
    const float* vb = kb + HW;
    const float* dob = dctx + (size_t)b * S * ld_dc + hd * 64;
    const float* ob = ctx + (size_t)b * S * ld_c + hd * 64;
    float* dqb = dqkv + (size_t)b * S * ld_d + hd * 64;
    const unsigned bh = (unsigned)(b * heads + hd);

    // ---------------- phase 1 staging: K, V (hi | lo), lse, bias ----------------
    stage_split<SP>(kb, ld, S, sAh, sAl, tid);
    stage_split<SP>(vb, ld, S, sBh, sBl, tid);
    for (int k = tid; k < SP; k += X3_WAVES * 64) {
        sBias[k] = (k < S) ? (key_bias ? key_bias[(size_t)b * S + k] * (1.0f / scale) : 0.f) : -INFINITY;
        sLse[k] = (k < S) ? lse[((size_t)b * heads + hd) * S + k] * LOG2E : INFINITY;  // padded queries -> p = 0
    }
    __syncthreads();
    const float scale2 = scale * LOG2E;
    // ---------------- phase 1: a wave owns queries [q0, q0 + 32): delta = dO . O, then dQ in one pass ----------------
#pragma unroll 1
    for (int blk = wave; blk < NB; blk += X3_WAVES) {
        asm volatile("" ::: "memory");
        const int q0 = blk * 32;
        const int qrow = min(q0 + (lane & 31), S - 1);
        bf16x8 qh[4], ql[4], doh[4], dol[4];
        float dpart = 0.f;   // this lane half's 32 head dims of dO . O (f32 in, f32 sum: exact to rounding)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            frag_global_split(qb, ld, qrow, ks, lane, qh[ks], ql[ks]);
            const float* dp_ = dob + (size_t)qrow * ld_dc + ks * 16 + 8 * h;
            const float* op_ = ob + (size_t)qrow * ld_c + ks * 16 + 8 * h;
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(dp_), d1 = *reinterpret_cast<const f32x4*>(dp_ + 4);
            const f32x4 o0 = *reinterpret_cast<const f32x4*>(op_), o1 = *reinterpret_cast<const f32x4*>(op_ + 4);
            split8(d0, d1, doh[ks], dol[ks]);
#pragma unroll
            for (int i = 0; i < 4; ++i) dpart = fmaf(d0[i], o0[i], fmaf(d1[i], o1[i], dpart));
        }
        const float delta_q = dpart + __shfl_xor(dpart, 32, 64);
        if (h == 0) sDelta[q0 + (lane & 31)] = delta_q;
        const float nlse_q = -sLse[q0 + (lane & 31)];
        const unsigned dbase = (bh * S + (unsigned)qrow) * SP + 4 * h;  // dropout index of (q, key 4h)
        f32x16 dq[2] = {zero16(), zero16()};
#pragma unroll
        for (int kt = 0; kt < NB; ++kt) {
            asm volatile("" ::: "memory");   // keep each tile's LDS reads inside its own iteration (register pressure)
            f32x16 s, dp = zero16();
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(sBias + 32 * kt + 8 * g + 4 * h);
#pragma unroll
                for (int i = 0; i < 4; ++i) s[4 * g + i] = b4[i];
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = mfma3(frag_rm(sAh, 32 * kt, ks, lane), frag_rm(sAl, 32 * kt, ks, lane), qh[ks], ql[ks], s);       // S^T[key, q] + bias / scale
                dp = mfma3(frag_rm(sBh, 32 * kt, ks, lane), frag_rm(sBl, 32 * kt, ks, lane), doh[ks], dol[ks], dp);  // dP^T[key, q]
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 k4 = {1.f, 1.f, 1.f, 1.f};
                if constexpr (DROP) k4 = keep4(drop, dbase + 32 * kt + 8 * g);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float pr = __builtin_amdgcn_exp2f(fmaf(s[4 * g + i], scale2, nlse_q));
                    dp[4 * g + i] = pr * (dp[4 * g + i] * k4[i] - delta_q);   // dS^T (scale applied to dQ at the end)
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 dsh, dsl;
                split_acc8(dp, s2, dsh, dsl);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)   // dQ^T += K^T dS^T
                    dq[dt] = mfma3(frag_tr(sAh, 32 * dt, 32 * kt + 16 * s2, lane), frag_tr(sAl, 32 * dt, 32 * kt + 16 * s2, lane), dsh, dsl, dq[dt]);
            }
        }
        const int q = q0 + (lane & 31);
        if (q < S) {
            store_dt_f32(dq, scale, dqb + (size_t)q * ld_d, lane);
            if (dqkv3 != nullptr) store_dt_split3(dq, scale, dqkv3 + (size_t)(b * S + q) * ld_d3 + hd * 64, 3 * HW, lane);   // [dq | dk | dv] as the dX GEMM's operand
        }
    }
    __syncthreads();
    // ---------------- phase 2 staging: Q, dO (hi | lo) ----------------
    stage_split<SP>(qb, ld, S, sAh, sAl, tid);
    stage_split<SP>(dob, ld_dc, S, sBh, sBl, tid);
    __syncthreads();
    // ---------------- phase 2: a wave owns keys [k0, k0 + 32): dV, dK ----------------
#pragma unroll 1
    for (int blk = wave; blk < NB; blk += X3_WAVES) {
        asm volatile("" ::: "memory");
        const int k0 = blk * 32;
        const int krow = min(k0 + (lane & 31), S - 1);
        bf16x8 kh[4], kl[4], vh[4], vl[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            frag_global_split(kb, ld, krow, ks, lane, kh[ks], kl[ks]);
            frag_global_split(vb, ld, krow, ks, lane, vh[ks], vl[ks]);
        }
        f32x16 bk16;  // the key's bias / scale as the score accumulator's start value (same for every query row)
        {
            const float bias_k = sBias[k0 + (lane & 31)];
#pragma unroll
            for (int r = 0; r < 16; ++r) bk16[r] = bias_k;
        }
        f32x16 dv[2] = {zero16(), zero16()}, dk[2] = {zero16(), zero16()};
#pragma unroll 1
        for (int qt = 0; qt < NB; ++qt) {
            f32x16 s = bk16, dp = zero16();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = mfma3(frag_rm(sAh, 32 * qt, ks, lane), frag_rm(sAl, 32 * qt, ks, lane), kh[ks], kl[ks], s);      // S[q, key] + bias / scale
                dp = mfma3(frag_rm(sBh, 32 * qt, ks, lane), frag_rm(sBl, 32 * qt, ks, lane), vh[ks], vl[ks], dp);    // dP[q, key]
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(sLse + 32 * qt + 8 * g + 4 * h);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(sDelta + 32 * qt + 8 * g + 4 * h);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float pr = __builtin_amdgcn_exp2f(fmaf(s[4 * g + i], scale2, -l4[i]));
                    if constexpr (DROP) {
                        const int q = min(32 * qt + 8 * g + 4 * h + i, S - 1);
                        const float keep = drop_factor(drop, (bh * S + (unsigned)q) * SP + (unsigned)krow);
                        s[4 * g + i] = pr * keep;                               // dropped P (feeds dV)
                        dp[4 * g + i] = pr * (dp[4 * g + i] * keep - d4[i]);    // dS (scale applied to dK at the end)
                    } else {
                        s[4 * g + i] = pr;
                        dp[4 * g + i] = pr * (dp[4 * g + i] - d4[i]);
                    }
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 ph, pl, dsh, dsl;
                split_acc8(s, s2, ph, pl);
                split_acc8(dp, s2, dsh, dsl);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[dt] = mfma3(frag_tr(sBh, 32 * dt, 32 * qt + 16 * s2, lane), frag_tr(sBl, 32 * dt, 32 * qt + 16 * s2, lane), ph, pl, dv[dt]);     // dO^T P
                    dk[dt] = mfma3(frag_tr(sAh, 32 * dt, 32 * qt + 16 * s2, lane), frag_tr(sAl, 32 * dt, 32 * qt + 16 * s2, lane), dsh, dsl, dk[dt]);   // Q^T dS
                }
            }
        }
        const int key = k0 + (lane & 31);
        if (key < S) {
            store_dt_f32(dk, scale, dqb + (size_t)key * ld_d + HW, lane);
            store_dt_f32(dv, 1.0f, dqb + (size_t)key * ld_d + 2 * HW, lane);
            if (dqkv3 != nullptr) {
                bf16_t* r3 = dqkv3 + (size_t)(b * S + key) * ld_d3 + hd * 64;
                store_dt_split3(dk, scale, r3 + HW, 3 * HW, lane);
                store_dt_split3(dv, 1.0f, r3 + 2 * HW, 3 * HW, lane);
            }
        }
    }
}

}  // namespace

#define X3_FWD_CASE(NBV)                                                                                                        \
    case NBV:                                                                                                                   \
        if (drop.thr16)                                                                                                         \
            hipLaunchKernelGGL((attn_fwd_x3_kernel<NBV, true>), dim3(B * heads), dim3(X3_WAVES * 64), 0, s, qkv, ld_qkv, S, heads, key_bias, \
                               scale, ctx, ld_ctx, lse, drop, ctx3, ld_c3);                                                     \
        else                                                                                                                    \
            hipLaunchKernelGGL((attn_fwd_x3_kernel<NBV, false>), dim3(B * heads), dim3(X3_WAVES * 64), 0, s, qkv, ld_qkv, S, heads, key_bias, \
                               scale, ctx, ld_ctx, lse, drop, ctx3, ld_c3);                                                     \
        break;

// internal (called by exact.hip's bsclip_attn_fwd_f32 / bsclip_attn_bwd_f32 after their argument checks)
void bsclip_launch_attn_fwd_x3(const float* qkv, int ld_qkv, int B, int S, int heads, const float* key_bias, float scale, float* ctx,
                               int ld_ctx, float* lse, const DropCfg& drop, bf16_t* ctx3, int ld_c3, hipStream_t s) {
    switch ((S + 31) / 32) {
        X3_FWD_CASE(1) X3_FWD_CASE(2) X3_FWD_CASE(3) X3_FWD_CASE(4) X3_FWD_CASE(5) X3_FWD_CASE(6) X3_FWD_CASE(7)
    }
}

#define X3_BWD_CASE(NBV)                                                                                                        \
    case NBV:                                                                                                                   \
        if (drop.thr16)                                                                                                         \
            hipLaunchKernelGGL((attn_bwd_x3_kernel<NBV, true>), dim3(B * heads), dim3(X3_WAVES * 64), 0, s, qkv, ld_qkv, dctx, ld_dctx, ctx, \
                               ld_ctx, lse, S, heads, key_bias, scale, dqkv, ld_dqkv, drop, dqkv3, ld_d3);                      \
        else                                                                                                                    \
            hipLaunchKernelGGL((attn_bwd_x3_kernel<NBV, false>), dim3(B * heads), dim3(X3_WAVES * 64), 0, s, qkv, ld_qkv, dctx, ld_dctx, ctx, \
                               ld_ctx, lse, S, heads, key_bias, scale, dqkv, ld_dqkv, drop, dqkv3, ld_d3);                      \
        break;

void bsclip_launch_attn_bwd_x3(const float* qkv, int ld_qkv, const float* dctx, int ld_dctx, const float* ctx, int ld_ctx, const float* lse,
                               int B, int S, int heads, const float* key_bias, float scale, float* dqkv, int ld_dqkv, const DropCfg& drop,
                               bf16_t* dqkv3, int ld_d3, hipStream_t s) {
    switch ((S + 31) / 32) {
        X3_BWD_CASE(1) X3_BWD_CASE(2) X3_BWD_CASE(3) X3_BWD_CASE(4) X3_BWD_CASE(5) X3_BWD_CASE(6) X3_BWD_CASE(7)
    }
}
