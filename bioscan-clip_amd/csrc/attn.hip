// Self-attention forward / backward for short sequences (S <= 224, head_dim 64) on gfx950.
//
// Replaces timm Attention.forward (softmax(q k^T * 64^-0.5) v; reached from image_encoder.py:108-109) and HF
// BertSelfAttention (same, plus the additive key mask used by the text tower, language_encoder.py:89).
//
// One workgroup (4 waves) per (batch, head); the whole K / V (or Q / dO) of that head lives in LDS as row-major
// XOR-swizzled tiles; a wave owns 32-row blocks w, w+4, ...  LDS <= 58 KiB, so two workgroups share a CU and one's
// staging/barriers hide behind the other's MFMA work.  All five/seven products run on v_mfma_f32_32x32x16_bf16.  Score tiles are computed TRANSPOSED
// (key on the accumulator rows, query on the lane) so that
//   * the softmax reduction over keys is in-lane (16 registers per tile) plus one xor-32 shuffle, and
//   * the probability tile feeds the next MFMA straight from the accumulator registers ("accumulator tile as the
//     next MFMA's operand", cdna_hip_programming.md 3): P^T never goes through LDS.
// The operand that must be k-strided for that second product (V^T, Q^T, dO^T, K^T) is read TRANSPOSED out of the same
// row-major image with ds_read_b64_tr_b16 -- no second LDS copy, no 2-byte scatter writes.
// Backward runs two phases in one launch: query-owner waves produce delta = rowsum(P.dP) and dQ, then key-owner waves
// produce dK/dV, so nothing is accumulated across waves (no atomics, bitwise reproducible).
#include "attn_common.h"

namespace {

// V2 (the forward half of attn_sweep.hip's backward): instead of the log-sum-exp the kernel leaves, per query row, what lets
// the backward take delta = rowsum(P dP) from the forward's OUTPUT without losing the softmax-backward cancellation:
//   stats[row] = (nm2, inv, rZ, 0):  e_k = exp2(s_k * scale2 + nm2), inv = 1 / sum_k e_k, rZ = sum_k e_k / Z' with
//   Z' = sum over kept keys of pd_k / keep_scale + sum over dropped keys of e_k,  pd_k = bf16(e_k * keep_k) -- the operand the
//   P.V product really used -- so that sum_k dS_k = 0 holds to f32 rounding for dS_k = inv pd_k (dP_k - delta / keep_scale),
//   delta = (dO . O) rZ;  and O itself to 16 mantissa bits: ctx = bf16(O), ctx_lo = bf16(O - ctx).
// KB (DROP only): the forward leaves its keep decisions as bit words (attn_common.h KEEP_WORDS) for the backward.  A template parameter,
// not a test of the pointer: behind a run-time branch the compiler sinks the whole bit computation into the branch and keeps all 80
// factors (or compare masks) of a query block alive until then (+46 VGPRs at S = 133, spills at S = 197).
template <int NB, bool DROP, int TAIL = 32, bool V2 = false, bool KB = false>
__global__ __launch_bounds__(ATT_WAVES * 64, 2) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, int ld, int S,
                                                                   int heads, const float* __restrict__ key_bias,
                                                                   float scale, bf16_t* __restrict__ ctx, int ld_ctx,
                                                                   float* __restrict__ lse, DropCfg drop, int nqb,
                                                                   bf16_t* __restrict__ ctx_lo = nullptr,
                                                                   unsigned* __restrict__ kbits = nullptr) {
    constexpr int SP = NB * 32;
    BSCLIP_DROP_RESOLVE(drop);
    __shared__ __attribute__((aligned(16))) char smem[2 * SP * ROWB + SP * 4];
    char* sK = smem;
    char* sV = smem + SP * ROWB;
    float* sBias = reinterpret_cast<float*>(smem + 2 * SP * ROWB);

    const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = heads * 64;
    const bf16_t* qb = qkv + (size_t)b * S * ld + hd * 64;
    const bf16_t* kb = qb + HW;
    const bf16_t* vb = kb + HW;

    stage_tile<SP>(kb, ld, S, sK, wave, lane);
    stage_tile<SP>(vb, ld, S, sV, wave, lane);
    // The additive key bias rides in the MFMA accumulator: the score tile starts from bias / scale instead of zero, so
    // acc = q.k + bias / scale and p = exp2(acc * scale2 - m * scale2) is one fma + one v_exp_f32 per score (scale2 = scale *
    // log2 e).  Padded keys start from -inf (HF's finfo.min mask overflows to -inf as well: probability exactly 0).
    const float inv_scale = 1.0f / scale;
    for (int k = tid; k < SP; k += ATT_WAVES * 64)
        sBias[k] = (k < S) ? (key_bias ? key_bias[(size_t)b * S + k] * inv_scale : 0.f) : -INFINITY;
    stage_wait();
    __syncthreads();
    const float scale2 = scale * LOG2E;

#pragma unroll 1
    for (int blk = wave; blk < nqb; blk += ATT_WAVES) {  // nqb <= NB: only the leading query blocks are wanted
        asm volatile("" ::: "memory");  // LDS tiles are loop-invariant: stop LICM from hoisting ~100 fragment registers
        const int q0 = blk * 32;
        const int qrow = min(q0 + (lane & 31), S - 1);
        bf16x8 qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = frag_global(qb, ld, qrow, ks, lane);

        // S^T tiles: rows = keys, lane = query
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
            for (int ks = 0; ks < 4; ++ks) acc = mfma32(frag_rm(sK, 32 * kt, ks, lane), qf[ks], acc);
            const int nr = 4 * (kt == NB - 1 ? tail_groups<TAIL>() : 4);   // registers that can hold a valid key
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (r < nr) m = fmaxf(m, acc[r]);
            p[kt] = acc;
        }
        m = fmaxf(m, __shfl_xor(m, 32, 64));  // raw-score units
        const float nm2 = -m * scale2;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NB; ++kt) {
            const int nr = 4 * (kt == NB - 1 ? tail_groups<TAIL>() : 4);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (r < nr) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(p[kt][r], scale2, nm2));
                    p[kt][r] = e;
                    sum += e;
                } else {
                    p[kt][r] = 0.f;   // rows past the sequence: probability 0 by construction
                }
            }
        }
        sum += __shfl_xor(sum, 32, 64);
        float sumx = 0.f, zacc = 0.f;   // V2: sum of e_k * keep_k before rounding, sum of the rounded operand pd_k
        if constexpr (DROP) {  // HF: dropout on the normalised probabilities (the row sum above is taken before it)
            const unsigned base = ((unsigned)(b * heads + hd) * S + (unsigned)qrow) * SP + 4 * h;
            if constexpr (KB) {
                // Bits first, factors from the bits: the compiler sinks the p *= keep multiplies into the P.V loop below, and with the
                // factors taken from the hashes it kept all 80 of a query block alive until then (+50 VGPRs at S = 133, spills at
                // S = 197); a factor re-derived from its bit (v_bfe_i32 + v_and) needs only the row's NB words.
                unsigned kw[KEEP_WORDS];   // this lane half's bits of the row's keep words: bit 8 g + i = key 32 kt + 8 g + 4 h + i
#pragma unroll
                for (int kt = 0; kt < KEEP_WORDS; ++kt) kw[kt] = 0u;
#pragma unroll
                for (int kt = 0; kt < NB; ++kt)
#pragma unroll
                    for (int g = 0; g < (kt == NB - 1 ? tail_groups<TAIL>() : 4); ++g)
                        kw[kt] |= keep4_nibble(drop, base + 32 * kt + 8 * g) << (8 * g);
                if (q0 + (lane & 31) < S) {   // each lane half stores ITS bits (no exchange): [row][half][KEEP_WORDS]
                    unsigned* kr = kbits + (((size_t)(b * heads + hd) * S + q0 + (lane & 31)) * 2 + h) * KEEP_WORDS;
                    *reinterpret_cast<u32x4*>(kr) = u32x4{kw[0], kw[1], kw[2], kw[3]};
                    if constexpr (NB > 4) *reinterpret_cast<u32x4*>(kr + 4) = u32x4{kw[4], kw[5], kw[6], kw[7]};
                }
#pragma unroll
                for (int kt = 0; kt < NB; ++kt)
#pragma unroll
                    for (int r = 0; r < 4 * (kt == NB - 1 ? tail_groups<TAIL>() : 4); ++r) {
                        p[kt][r] *= keep_of_bit(kw[kt], 8 * (r >> 2) + (r & 3), drop.scale);
                        if constexpr (V2) sumx += p[kt][r];
                    }
            } else {
#pragma unroll
                for (int kt = 0; kt < NB; ++kt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (g >= (kt == NB - 1 ? tail_groups<TAIL>() : 4)) continue;
                        const f32x4 k4 = keep4(drop, base + 32 * kt + 8 * g);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            p[kt][4 * g + i] *= k4[i];
                            if constexpr (V2) sumx += p[kt][4 * g + i];
                        }
                    }
            }
        }

        // O^T[d, query] = sum_key V^T[d, key] P^T[key, query]
        f32x16 o[2] = {zero16(), zero16()};
#pragma unroll
        for (int kt = 0; kt < NB; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                if (s2 >= (kt == NB - 1 ? tail_ksteps<TAIL>() : 2)) continue;   // keys past the sequence carry probability 0
                const bf16x8 pb = pack8(p[kt], s2);
                if constexpr (V2) {
                    const u32x4 pw = __builtin_bit_cast(u32x4, pb);
#pragma unroll
                    for (int i = 0; i < 4; ++i) zacc = bf_pair_sum(pw[i], zacc);
                }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) o[dt] = mfma32(frag_tr(sV, 32 * dt, 32 * kt + 16 * s2, lane), pb, o[dt]);
            }

        const int q = q0 + (lane & 31);
        if constexpr (V2) {
            zacc += __shfl_xor(zacc, 32, 64);
            if constexpr (DROP) sumx += __shfl_xor(sumx, 32, 64);
            const float inv = 1.0f / sum;
            const float zp = DROP ? sum + (zacc - sumx) / drop.scale : zacc;
            if (q < S) {
                store_dt_hilo(o, inv, ctx + (size_t)(b * S + q) * ld_ctx + hd * 64, ctx_lo + (size_t)(b * S + q) * ld_ctx + hd * 64, lane);
                if (h == 0)
                    *reinterpret_cast<f32x4*>(lse + (((size_t)b * heads + hd) * S + q) * 4) = f32x4{nm2, inv, sum / zp, 0.f};
            }
        } else if (q < S) {
            store_dt(o, 1.0f / sum, ctx + (size_t)(b * S + q) * ld_ctx + hd * 64, lane);
            if (h == 0) lse[((size_t)b * heads + hd) * S + q] = (__log2f(sum) - nm2) * LN2;  // natural-log LSE
        }
    }
}

// BITS (DROP only): the dropout decisions come from the keep-bit words the forward left (attn_common.h KEEP_WORDS) instead of being
// re-hashed per element pair (query-owner phase) / per element (key-owner phase).  PRE (no dropout): the key-owner phase starts the
// dP accumulator from -delta ("row constants as the initial accumulator", cdna_hip_programming.md, attention backward): dS = P dP'.
//
// LORA (round 5): the kernel also leaves what the rank-4 LoRA branches on q and v need from dq / dv while those still sit in the
// accumulators -- lora_grad's first pass (optim.hip lora_grad_dt_db_kernel) read dq and dv back from HBM for it (155 MB per ViT layer):
//   dt_partial[head][q | v][token][0:4] = dq_head . Bq_head | dv_head . Bv_head   (this head's 64 of the 768 terms; summed over heads later;
//                                          a wave's 32 tokens are 512 contiguous bytes: whole lines, written once)
//   db_partial[item][q | v][j][d]      = sum_token t[token][j] * dq | dv[token][d]  (this item's tokens; summed over the batch later)
// Both are MFMA products.  dt: the accumulator tile (d on rows, token on the lane) is the B operand as it stands (k = d), the A operand
// is B^T of the head -- hi + lo bf16 parts of the f32 master, so the product is the f32 one up to accumulation order.  db contracts
// over tokens, which sit on the LANES of the accumulator: the bf16 tile goes through a small per-wave LDS tile ([32 tokens][32 d], 72-byte
// rows) and comes back token-strided with ds_read_b64_tr_b16.  Only 4 accumulator rows (j) of either product are used; the rows of
// the small A operands past j = 3 are zero.
struct LoraPart {
    const bf16_t* t;     // [tokens, ld_t]: t_q at columns 0..3, t_v at 4..7 (the LayerNorm's t block of the QKV GEMM's operand)
    int ld_t;
    const float* b;      // LoRA-B master [2][heads * 64][4]
    float* dtp;          // [heads][2][tokens][4]
    float* dbp;          // [B * heads][2][4][64]
};
constexpr int LORA_TILE = 32 * 72;   // bytes of one wave's transposition tile

// the 8 k-elements of a [4][len] bf16 row-major table as an MFMA A fragment: row j = lane & 31 (zero for j >= 4), elements in the k order
// of pack8() / frag_tr: c0 + 8 (jj >> 2) + 4 (lane >> 5) + (jj & 3)
__device__ __forceinline__ bf16x8 small_frag(const bf16_t* tab, int len, int c0, int lane) {
    const int j = lane & 31, h = lane >> 5;
    uint2 a = {0u, 0u}, b = {0u, 0u};
    if (j < 4) {
        a = *reinterpret_cast<const uint2*>(tab + j * len + c0 + 4 * h);
        b = *reinterpret_cast<const uint2*>(tab + j * len + c0 + 8 + 4 * h);
    }
    return __builtin_bit_cast(bf16x8, u32x4{a.x, a.y, b.x, b.y});
}

template <int NB, bool DROP, bool DIAG = false, int TAIL = 32, bool BITS = false, bool PRE = false, bool LORA = false>
__global__ __launch_bounds__(ATT_WAVES * 64, 2) void attn_bwd_kernel(const bf16_t* __restrict__ qkv, int ld,
                                                                   const bf16_t* __restrict__ dctx, int ld_ctx,
                                                                   const float* __restrict__ lse, int S, int heads,
                                                                   const float* __restrict__ key_bias, float scale,
                                                                   bf16_t* __restrict__ dqkv, int ld_d, DropCfg drop,
                                                                   int nqb, unsigned long long* diag = nullptr,
                                                                   const unsigned* __restrict__ kbits = nullptr,
                                                                   LoraPart lp = LoraPart{}) {
    static_assert(!(BITS && !DROP) && !(PRE && DROP), "BITS needs dropout, PRE excludes it");
    constexpr int SP = NB * 32;
    constexpr int RM = SP * ROWB;
    BSCLIP_DROP_RESOLVE(drop);
    auto stamp = [&](int i) {  // diagnostic build: per-wave section times (100 MHz wall clock); tools/attn_phases.py
        if constexpr (DIAG) {
            if ((threadIdx.x & 63) == 0) diag[((size_t)blockIdx.x * ATT_WAVES + (threadIdx.x >> 6)) * 8 + i] = wall_clock64();
        }
    };
    stamp(0);
    constexpr int KB_BYTES = BITS ? NB * SP * 4 : 0;
    constexpr int LORA_BYTES = LORA ? ATT_WAVES * LORA_TILE + 2 * 2 * 4 * 64 * 2 + 8 * SP * 2 : 0;
    __shared__ __attribute__((aligned(16))) char smem[2 * RM + 3 * SP * 4 + KB_BYTES + LORA_BYTES];
    char* sR0 = smem;       // phase 1: K | phase 2: Q
    char* sR1 = smem + RM;  // phase 1: V | phase 2: dO
    float* sLse = reinterpret_cast<float*>(smem + 2 * RM);
    float* sDelta = sLse + SP;
    float* sBias = sDelta + SP;
    [[maybe_unused]] unsigned* sKb = reinterpret_cast<unsigned*>(sBias + SP);   // BITS, phase 2: keep words [key tile][query]
    [[maybe_unused]] char* sTr = smem + 2 * RM + 3 * SP * 4 + KB_BYTES;           // LORA: per-wave transposition tiles, then the dB sums
    [[maybe_unused]] bf16_t* sBt = reinterpret_cast<bf16_t*>(sTr + ATT_WAVES * LORA_TILE);   // B^T of the head: [q | v][hi | lo][j][d]
    [[maybe_unused]] bf16_t* sTt = sBt + 2 * 2 * 4 * 64;                          // t^T of the item: [8 j][SP tokens], zero past S

    const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = heads * 64;
    const bf16_t* qb = qkv + (size_t)b * S * ld + hd * 64;
    const bf16_t* kb = qb + HW;
    const bf16_t* vb = kb + HW;
    const bf16_t* dob = dctx + (size_t)b * S * ld_ctx + hd * 64;
    bf16_t* dqb = dqkv + (size_t)b * S * ld_d + hd * 64;
    const unsigned bh = (unsigned)(b * heads + hd);

    // LORA: the head's rows of B (16 bytes each: one load per thread of the first two waves) and the item's t rows are requested FIRST, so
    // that they are in flight under the tile copies issued next and are waited for without waiting for those (loads return in order)
    [[maybe_unused]] f32x4 b_row = {0.f, 0.f, 0.f, 0.f};
    [[maybe_unused]] u32x4 t_row = {0u, 0u, 0u, 0u};
    if constexpr (LORA) {
        static_assert(SP <= ATT_WAVES * 64, "one t row per thread");
        if (tid < 128) b_row = *reinterpret_cast<const f32x4*>(lp.b + ((size_t)(tid >> 6) * HW + hd * 64 + (tid & 63)) * 4);
        if (tid < S) t_row = *reinterpret_cast<const u32x4*>(lp.t + (size_t)(b * S + tid) * lp.ld_t);
    }
    // ---------------- phase 1 staging: K, V row-major, lse, bias ----------------
    stage_tile<SP>(kb, ld, S, sR0, wave, lane);
    stage_tile<SP>(vb, ld, S, sR1, wave, lane);
    for (int k = tid; k < SP; k += ATT_WAVES * 64) {
        // log2 domain: p = exp2(acc * scale2 - lse2) with acc = q.k + bias / scale (the bias is the accumulator's start value)
        sBias[k] = (k < S) ? (key_bias ? key_bias[(size_t)b * S + k] * (1.0f / scale) : 0.f) : -INFINITY;
        sLse[k] = (k < S) ? lse[((size_t)b * heads + hd) * S + k] * LOG2E : INFINITY;  // padded queries -> p = 0
    }
    if constexpr (LORA) {
        if (tid < 128) {   // B^T of the head as hi + lo bf16 parts: [q | v][hi | lo][j][d]
            const int is_v = tid >> 6, d = tid & 63;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16_t hi = f2bf(b_row[j]);
                sBt[((is_v * 2 + 0) * 4 + j) * 64 + d] = hi;
                sBt[((is_v * 2 + 1) * 4 + j) * 64 + d] = f2bf(b_row[j] - bf2f(hi));
            }
        }
        if (tid < SP) {    // t^T of the item: [8 j][SP tokens], zero past S
#pragma unroll
            for (int j = 0; j < 8; ++j) sTt[j * SP + tid] = (bf16_t)((t_row[j >> 1] >> (16 * (j & 1))) & 0xffffu);
        }
    }
    // LoRA partial products of one 32-token block whose dq (is_v = 0) or dv (1) sits in acc (d on rows, token on the lane), already scaled
    [[maybe_unused]] auto lora_part = [&](const f32x16 (&acc)[2], int is_v, int tok0, f32x4 (&kacc)[2]) {
        char* tile = sTr + wave * LORA_TILE;
        const bf16_t* bt = sBt + is_v * (2 * 4 * 64);
        const bf16_t* tt = sTt + is_v * 4 * SP;
        // order: the first half's tile goes to LDS, the dt products (independent of it) run while it lands, then the dB products
        auto put_tile = [&](int dt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {   // [token][d - 32 dt] bf16, 72-byte rows
                uint2 o;
                o.x = pack_bf2(acc[dt][4 * g + 0], acc[dt][4 * g + 1]);
                o.y = pack_bf2(acc[dt][4 * g + 2], acc[dt][4 * g + 3]);
                *reinterpret_cast<uint2*>(tile + (lane & 31) * 72 + (8 * g + 4 * h) * 2) = o;
            }
        };
        auto db_half = [&](int dt) {   // dB^T[j, d] += t^T[j, token] x[token, d]: the tile read token-strided
            f32x16 dbacc = zero16();
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
                dbacc = mfma32(small_frag(tt, SP, tok0 + 16 * s2, lane), frag_tr_lin(tile, 72, 0, 16 * s2, lane), dbacc);
            kacc[dt] += f32x4{dbacc[0], dbacc[1], dbacc[2], dbacc[3]};
        };
        put_tile(0);
        f32x16 dtacc = zero16(), dtacc2 = zero16();   // two chains: the hi and the lo part of B
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 xb = pack8(acc[dt], s2);   // k = d = 32 dt + 16 s2 + (pack8 order)
                dtacc = mfma32(small_frag(bt, 64, 32 * dt + 16 * s2, lane), xb, dtacc);
                dtacc2 = mfma32(small_frag(bt + 4 * 64, 64, 32 * dt + 16 * s2, lane), xb, dtacc2);
            }
        db_half(0);
        put_tile(1);
        const int tok = tok0 + (lane & 31);
        if (h == 0 && tok < S)   // accumulator rows 0..3 = j, on lane half 0
            *reinterpret_cast<f32x4*>(lp.dtp + ((((size_t)hd * 2 + is_v) * (gridDim.x / heads) + b) * S + tok) * 4) =
                f32x4{dtacc[0] + dtacc2[0], dtacc[1] + dtacc2[1], dtacc[2] + dtacc2[2], dtacc[3] + dtacc2[3]};
        db_half(1);
    };
    // a phase's dB^T sums of the 4 waves -> db_partial[item][is_v][j][d]; call with all the workgroup's threads
    [[maybe_unused]] auto lora_flush = [&](const f32x4 (&kacc)[2], int is_v) {
        float* mine = reinterpret_cast<float*>(sTr + wave * LORA_TILE);
        if (h == 0) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int j = 0; j < 4; ++j) mine[j * 64 + 32 * dt + lane] = kacc[dt][j];
        }
        __syncthreads();
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < ATT_WAVES; ++w) v += reinterpret_cast<const float*>(sTr + w * LORA_TILE)[tid];
        lp.dbp[((size_t)bh * 2 + is_v) * 256 + tid] = v;
    };
    stage_wait();
    __syncthreads();
    stamp(1);
    const float scale2 = scale * LOG2E;
    // ---------------- phase 1: a wave owns queries [q0, q0+32): delta, then dQ ----------------
    // Query blocks >= nqb carry a zero upstream gradient by contract (last ViT block: only token 0 feeds the head): their
    // dQ rows are written as zeros and they are skipped in both phases.
    for (int blk = nqb + wave; blk < NB; blk += ATT_WAVES) {
        const int q = blk * 32 + (lane & 31);
        if (q < S) {
            bf16_t* o = dqb + (size_t)q * ld_d + 32 * (lane >> 5);
#pragma unroll
            for (int c = 0; c < 4; ++c) *reinterpret_cast<u32x4*>(o + 8 * c) = u32x4{0u, 0u, 0u, 0u};
            if constexpr (LORA) {
                if (lane < 32)
                    *reinterpret_cast<f32x4*>(lp.dtp + (((size_t)hd * 2 * (gridDim.x / heads) + b) * S + q) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
    [[maybe_unused]] f32x4 kacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll 1
    for (int blk = wave; blk < nqb; blk += ATT_WAVES) {
        asm volatile("" ::: "memory");  // LDS tiles are loop-invariant: stop LICM from hoisting ~100 fragment registers
        const int q0 = blk * 32;
        const int qrow = min(q0 + (lane & 31), S - 1);
        bf16x8 qf[4], dof[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = frag_global(qb, ld, qrow, ks, lane);
            dof[ks] = frag_global(dob, ld_ctx, qrow, ks, lane);
        }
        const float nlse_q = -sLse[q0 + (lane & 31)];
        const unsigned dbase = (bh * S + (unsigned)qrow) * SP + 4 * h;  // dropout index of (q, key 4h)
        [[maybe_unused]] unsigned kwq[NB];   // BITS: the query row's keep words of this lane half: bit 8 g + i is key 32 kt + 8 g + 4 h + i
        if constexpr (BITS) {   // this lane half's own words, as the forward's lane half (same query, same h) stored them
            const unsigned* kr = kbits + (((size_t)bh * S + qrow) * 2 + h) * KEEP_WORDS;
            const u32x4 wa = *reinterpret_cast<const u32x4*>(kr);
            u32x4 wb = {0u, 0u, 0u, 0u};
            if constexpr (NB > 4) wb = *reinterpret_cast<const u32x4*>(kr + 4);
#pragma unroll
            for (int kt = 0; kt < NB; ++kt) kwq[kt] = kt < 4 ? wa[kt & 3] : wb[kt & 3];
        }
        auto keep_q = [&](int kt, int g) -> f32x4 {   // keep factors of keys 32 kt + 8 g + 4 h + (0..3) of this lane's query row
            if constexpr (BITS) {
                return f32x4{keep_of_bit(kwq[kt], 8 * g + 0, drop.scale), keep_of_bit(kwq[kt], 8 * g + 1, drop.scale),
                             keep_of_bit(kwq[kt], 8 * g + 2, drop.scale), keep_of_bit(kwq[kt], 8 * g + 3, drop.scale)};
            } else {
                return keep4(drop, dbase + 32 * kt + 8 * g);
            }
        };
        // pass 1: delta_q = sum_key P[q,key] dP[q,key], from the SAME P and dP the gradient uses, so that
        // sum_key dS[q,key] = 0 holds to f32 rounding (delta from the bf16-rounded O does not: it loses the cancellation
        // whenever the values of a head are nearly equal across keys).  Round 3: P is formed ONCE -- pass 1 keeps it as packed
        // bf16 (8 registers per key tile, 56 for S = 197), pass 2 recomputes only dP (f32: dP - delta is where the cancellation
        // lives) and multiplies: one S product, 16 exp and 16 fma per tile less.  A bf16 P puts a 2^-9 RELATIVE error on each
        // dS element (it is rounded to bf16 for the dQ product anyway); delta itself still comes from the f32 P.
        float dpart = 0.f;
        bf16x8 p16[NB][2];
#pragma unroll
        for (int kt = 0; kt < NB; ++kt) {   // unrolled: p16 must be indexed statically (a runtime index would put it in scratch)
            asm volatile("" ::: "memory");    // ... but keep each tile's LDS reads inside its own iteration (register pressure)
            f32x16 s, dp = zero16();
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(sBias + 32 * kt + 8 * g + 4 * h);
#pragma unroll
                for (int i = 0; i < 4; ++i) s[4 * g + i] = b4[i];
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = mfma32(frag_rm(sR0, 32 * kt, ks, lane), qf[ks], s);     // S^T[key, q] + bias / scale
                dp = mfma32(frag_rm(sR1, 32 * kt, ks, lane), dof[ks], dp);  // dP^T[key, q]
            }
            const int ng = kt == NB - 1 ? tail_groups<TAIL>() : 4;   // register groups that can hold a valid key (compile time)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g >= ng) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) s[4 * g + i] = 0.f;
                    continue;
                }
                f32x4 k4 = {1.f, 1.f, 1.f, 1.f};
                if constexpr (DROP) k4 = keep_q(kt, g);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float pr = __builtin_amdgcn_exp2f(fmaf(s[4 * g + i], scale2, nlse_q));
                    if constexpr (DROP) dpart = fmaf(pr, dp[4 * g + i] * k4[i], dpart);
                    else dpart = fmaf(pr, dp[4 * g + i], dpart);
                    s[4 * g + i] = pr;
                }
            }
            p16[kt][0] = pack8(s, 0);
            p16[kt][1] = pack8(s, 1);
        }
        const float delta_q = dpart + __shfl_xor(dpart, 32, 64);
        if (h == 0) sDelta[q0 + (lane & 31)] = PRE ? -delta_q : delta_q;
        if (blk == wave) stamp(6);
        // pass 2: dS^T = P^T (dP^T - delta);  dQ^T += K^T dS^T, scaled once at the end
        f32x16 dq[2] = {zero16(), zero16()};
#pragma unroll
        for (int kt = 0; kt < NB; ++kt) {
            asm volatile("" ::: "memory");
            f32x16 dp = zero16();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) dp = mfma32(frag_rm(sR1, 32 * kt, ks, lane), dof[ks], dp);
            const int ng = kt == NB - 1 ? tail_groups<TAIL>() : 4, ns2 = kt == NB - 1 ? tail_ksteps<TAIL>() : 2;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                if (s2 >= ns2) continue;
                const u32x4 pu = __builtin_bit_cast(u32x4, p16[kt][s2]);
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    const int g = 2 * s2 + g2;
                    if (g >= ng) {   // keys past the sequence inside a used k-step: dS = 0
#pragma unroll
                        for (int i = 0; i < 4; ++i) dp[4 * g + i] = 0.f;
                        continue;
                    }
                    f32x4 k4 = {1.f, 1.f, 1.f, 1.f};
                    if constexpr (DROP) k4 = keep_q(kt, g);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const unsigned w = pu[2 * g2 + (i >> 1)];
                        const float pr = __uint_as_float((i & 1) ? (w & 0xffff0000u) : (w << 16));
                        if constexpr (DROP) dp[4 * g + i] = pr * (dp[4 * g + i] * k4[i] - delta_q);
                        else dp[4 * g + i] = pr * (dp[4 * g + i] - delta_q);
                    }
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                if (s2 >= ns2) continue;
                const bf16x8 dsb = pack8(dp, s2);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    dq[dt] = mfma32(frag_tr(sR0, 32 * dt, 32 * kt + 16 * s2, lane), dsb, dq[dt]);  // K^T dS^T
            }
        }
        const int q = q0 + (lane & 31);
        if constexpr (LORA) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) dq[dt][r] *= scale;
            lora_part(dq, 0, q0, kacc);
            if (q < S) store_dt(dq, 1.0f, dqb + (size_t)q * ld_d, lane);
        } else {
            if (q < S) store_dt(dq, scale, dqb + (size_t)q * ld_d, lane);
        }
        if (blk == wave) stamp(7);
    }
    stamp(2);
    if constexpr (LORA) {
        lora_flush(kacc, 0);   // (its barrier is the phase boundary)
        kacc[0] = kacc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
        __syncthreads();
    }
    stamp(3);
    // ---------------- phase 2 staging: Q, dO row-major ----------------
    stage_tile<SP>(qb, ld, S, sR0, wave, lane);
    stage_tile<SP>(dob, ld_ctx, S, sR1, wave, lane);
    if constexpr (BITS) {   // the item's keep words, transposed to [key tile][query]: a lane's 4 consecutive query rows are one 16-byte read
        for (int i = tid; i < S * NB; i += ATT_WAVES * 64) {   // bit j of the staged word = key 32 kt + j: half 1's nibbles sit 4 keys up
            const int q = i / NB, kt = i - q * NB;
            const unsigned* kr = kbits + ((size_t)bh * S + q) * 2 * KEEP_WORDS + kt;
            sKb[kt * SP + q] = kr[0] | (kr[KEEP_WORDS] << 4);
        }
    }
    stage_wait();
    __syncthreads();
    stamp(4);
    // ---------------- phase 2: a wave owns keys [k0, k0+32): dV, dK ----------------
#pragma unroll 1
    for (int blk = wave; blk < NB; blk += ATT_WAVES) {
        asm volatile("" ::: "memory");  // LDS tiles are loop-invariant: stop LICM from hoisting ~100 fragment registers
        const int k0 = blk * 32;
        const int krow = min(k0 + (lane & 31), S - 1);
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf[ks] = frag_global(kb, ld, krow, ks, lane);
            vf[ks] = frag_global(vb, ld, krow, ks, lane);
        }
        f32x16 bk16;  // the key's bias / scale as the score accumulator's start value (same for every query row)
        {
            const float bias_k = sBias[k0 + (lane & 31)];
#pragma unroll
            for (int r = 0; r < 16; ++r) bk16[r] = bias_k;
        }
        f32x16 dv[2] = {zero16(), zero16()}, dk[2] = {zero16(), zero16()};
        // one query tile; TAILED = the last tile of the sequence (rows past S carry probability 0): only its first groups / k-steps
        auto qtile = [&](int qt, auto tailed) {
            constexpr bool TAILED = decltype(tailed)::value;
            constexpr int ng = TAILED ? tail_groups<TAIL>() : 4, ns2 = TAILED ? tail_ksteps<TAIL>() : 2;
            f32x16 s = bk16, dp = zero16();
            if constexpr (PRE) {   // dP' = dP - delta: the row constant is the accumulator's start value (sDelta holds -delta)
#pragma unroll
                for (int g = 0; g < ng; ++g) {
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(sDelta + 32 * qt + 8 * g + 4 * h);
#pragma unroll
                    for (int i = 0; i < 4; ++i) dp[4 * g + i] = d4[i];
                }
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = mfma32(frag_rm(sR0, 32 * qt, ks, lane), kf[ks], s);    // S[q, key] + bias / scale
                dp = mfma32(frag_rm(sR1, 32 * qt, ks, lane), vf[ks], dp);  // dP[q, key]
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g >= ng) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) s[4 * g + i] = dp[4 * g + i] = 0.f;
                    continue;
                }
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(sLse + 32 * qt + 8 * g + 4 * h);
                [[maybe_unused]] f32x4 d4;
                if constexpr (!PRE) d4 = *reinterpret_cast<const f32x4*>(sDelta + 32 * qt + 8 * g + 4 * h);
                [[maybe_unused]] u32x4 w4;
                if constexpr (BITS) w4 = *reinterpret_cast<const u32x4*>(sKb + blk * SP + 32 * qt + 8 * g + 4 * h);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float pr = __builtin_amdgcn_exp2f(fmaf(s[4 * g + i], scale2, -l4[i]));
                    if constexpr (DROP) {
                        float keep;
                        if constexpr (BITS) {
                            keep = keep_of_bit(w4[i], (unsigned)(lane & 31), drop.scale);   // bit (key in tile) of (query, key tile blk)
                        } else {
                            const int q = min(32 * qt + 8 * g + 4 * h + i, S - 1);
                            keep = drop_factor(drop, (bh * S + (unsigned)q) * SP + (unsigned)krow);
                        }
                        s[4 * g + i] = pr * keep;                               // dropped P (feeds dV)
                        dp[4 * g + i] = pr * (dp[4 * g + i] * keep - d4[i]);    // dS (scale applied to dK at the end)
                    } else if constexpr (PRE) {
                        s[4 * g + i] = pr;
                        dp[4 * g + i] = pr * dp[4 * g + i];
                    } else {
                        s[4 * g + i] = pr;
                        dp[4 * g + i] = pr * (dp[4 * g + i] - d4[i]);
                    }
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < ns2; ++s2) {
                const bf16x8 pb = pack8(s, s2), dsb = pack8(dp, s2);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[dt] = mfma32(frag_tr(sR1, 32 * dt, 32 * qt + 16 * s2, lane), pb, dv[dt]);   // dO^T P
                    dk[dt] = mfma32(frag_tr(sR0, 32 * dt, 32 * qt + 16 * s2, lane), dsb, dk[dt]);  // Q^T dS
                }
            }
        };
        const int nfull = (TAIL < 32 && nqb == NB) ? nqb - 1 : nqb;   // wave-uniform
#pragma unroll 1
        for (int qt = 0; qt < nfull; ++qt) qtile(qt, std::false_type{});
        if (nfull < nqb) qtile(nqb - 1, std::true_type{});
        const int key = k0 + (lane & 31);
        if constexpr (LORA) lora_part(dv, 1, k0, kacc);
        if (key < S) {
            store_dt(dk, scale, dqb + (size_t)key * ld_d + HW, lane);
            store_dt(dv, 1.0f, dqb + (size_t)key * ld_d + 2 * HW, lane);
        }
    }
    if constexpr (LORA) lora_flush(kacc, 1);
    stamp(5);
}

}  // namespace

#define ATTN_FWD_LAUNCH(NBV, DR, TL, KBV)                                                                       \
    hipLaunchKernelGGL((attn_fwd_kernel<NBV, DR, TL, false, KBV>), dim3(B * heads), dim3(ATT_WAVES * 64), 0, s, \
                       static_cast<const bf16_t*>(qkv), ld_qkv, S, heads, key_bias, scale, static_cast<bf16_t*>(ctx), ld_ctx, \
                       lse, drop, nqb, static_cast<bf16_t*>(nullptr), static_cast<unsigned*>(keep_bits))
#define ATTN_FWD_PICK(NBV, TL)                                                                                  \
    do {                                                                                                        \
        if (drop.thr16 && keep_bits) ATTN_FWD_LAUNCH(NBV, true, TL, true);                                      \
        else if (drop.thr16) ATTN_FWD_LAUNCH(NBV, true, TL, false);                                             \
        else ATTN_FWD_LAUNCH(NBV, false, TL, false);                                                            \
    } while (0)
#define ATTN_FWD_CASE(NBV)                                                                                      \
    case NBV:                                                                                                   \
        ATTN_FWD_PICK(NBV, 32);                                                                                 \
        break;

extern "C" int bsclip_attn_fwd(const void* qkv, int ld_qkv, int B, int S, int heads, const float* key_bias,
                               float scale, void* ctx, int ld_ctx, float* lse, int q_rows, void* keep_bits, float dropout_p,
                               uint32_t dropout_seed, void* stream) {
    BSCLIP_REQUIRE(qkv && ctx && lse, "bsclip_attn_fwd: null pointer");
    BSCLIP_REQUIRE((reinterpret_cast<uintptr_t>(keep_bits) & 15) == 0, "bsclip_attn_fwd: keep_bits must be 16-byte aligned");
    BSCLIP_REQUIRE(B > 0 && heads > 0 && S > 0 && S <= 224, "bsclip_attn_fwd: B=%d heads=%d S=%d (S <= 224)", B, heads, S);
    BSCLIP_REQUIRE(ld_qkv >= 3 * heads * 64 && ld_qkv % 8 == 0 && ld_ctx >= heads * 64 && ld_ctx % 4 == 0,
                   "bsclip_attn_fwd: ld_qkv=%d ld_ctx=%d", ld_qkv, ld_ctx);
    BSCLIP_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "bsclip_attn_fwd: dropout_p=%f", dropout_p);
    BSCLIP_REQUIRE(q_rows >= 0 && q_rows <= S, "bsclip_attn_fwd: q_rows=%d (0 = all, <= S)", q_rows);
    const int nqb = q_rows > 0 ? (q_rows + 31) / 32 : (S + 31) / 32;
    const DropCfg drop = make_drop(dropout_p, dropout_seed);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // the production sequence lengths get the instantiation that knows their last tile holds 5 rows (see tail_groups)
    if (S == 197) {
        ATTN_FWD_PICK(7, 5);
    } else if (S == 133) {
        ATTN_FWD_PICK(5, 5);
    } else {
        switch ((S + 31) / 32) {
            ATTN_FWD_CASE(1) ATTN_FWD_CASE(2) ATTN_FWD_CASE(3) ATTN_FWD_CASE(4) ATTN_FWD_CASE(5) ATTN_FWD_CASE(6)
            ATTN_FWD_CASE(7)
        }
    }
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

#ifdef BSCLIP_DIAG   // experiment (round 4): only in libbsclip_hip_diag.so, like the backward it serves (attn_sweep.hip / attn_pers.hip)
#define ATTN_FWD2_LAUNCH(NBV, DR, TL)                                                                           \
    hipLaunchKernelGGL((attn_fwd_kernel<NBV, DR, TL, true>), dim3(B * heads), dim3(ATT_WAVES * 64), 0, s,       \
                       static_cast<const bf16_t*>(qkv), ld_qkv, S, heads, key_bias, scale, static_cast<bf16_t*>(ctx), ld_ctx, \
                       stats, drop, nqb, static_cast<bf16_t*>(ctx_lo))
#define ATTN_FWD2_CASE(NBV)                                                                                     \
    case NBV:                                                                                                   \
        if (drop.thr16) ATTN_FWD2_LAUNCH(NBV, true, 32);                                                        \
        else ATTN_FWD2_LAUNCH(NBV, false, 32);                                                                  \
        break;

// Forward for the key-owner-sweep backward (bsclip_attn_bwd2): same attention, but instead of the log-sum-exp it leaves
// stats[B, heads, S, 4] = (nm2, inv, rZ, 0) and the output to 16 mantissa bits (ctx = bf16(O), ctx_lo = bf16(O - ctx), both
// with row stride ld_ctx) -- see attn_fwd_kernel's V2 note.
extern "C" int bsclip_attn_fwd2(const void* qkv, int ld_qkv, int B, int S, int heads, const float* key_bias, float scale,
                                void* ctx, void* ctx_lo, int ld_ctx, float* stats, float dropout_p, uint32_t dropout_seed,
                                void* stream) {
    BSCLIP_REQUIRE(qkv && ctx && ctx_lo && stats, "bsclip_attn_fwd2: null pointer");
    BSCLIP_REQUIRE(B > 0 && heads > 0 && S > 0 && S <= 224, "bsclip_attn_fwd2: B=%d heads=%d S=%d (S <= 224)", B, heads, S);
    BSCLIP_REQUIRE(ld_qkv >= 3 * heads * 64 && ld_qkv % 8 == 0 && ld_ctx >= heads * 64 && ld_ctx % 4 == 0,
                   "bsclip_attn_fwd2: ld_qkv=%d ld_ctx=%d", ld_qkv, ld_ctx);
    BSCLIP_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "bsclip_attn_fwd2: dropout_p=%f", dropout_p);
    BSCLIP_REQUIRE((reinterpret_cast<uintptr_t>(stats) & 15) == 0, "bsclip_attn_fwd2: stats must be 16-byte aligned");
    const int nqb = (S + 31) / 32;
    const DropCfg drop = make_drop(dropout_p, dropout_seed);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (S == 197) {
        if (drop.thr16) ATTN_FWD2_LAUNCH(7, true, 5);
        else ATTN_FWD2_LAUNCH(7, false, 5);
    } else if (S == 133) {
        if (drop.thr16) ATTN_FWD2_LAUNCH(5, true, 5);
        else ATTN_FWD2_LAUNCH(5, false, 5);
    } else {
        switch ((S + 31) / 32) {
            ATTN_FWD2_CASE(1) ATTN_FWD2_CASE(2) ATTN_FWD2_CASE(3) ATTN_FWD2_CASE(4) ATTN_FWD2_CASE(5) ATTN_FWD2_CASE(6)
            ATTN_FWD2_CASE(7)
        }
    }
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}
#endif  // BSCLIP_DIAG

#define ATTN_BWD_LAUNCH(NBV, DR, TL, BT, PR, LR)                                                                 \
    hipLaunchKernelGGL((attn_bwd_kernel<NBV, DR, false, TL, BT, PR, LR>), dim3(B * heads), dim3(ATT_WAVES * 64), 0, s, \
                       static_cast<const bf16_t*>(qkv), ld_qkv, static_cast<const bf16_t*>(dctx), ld_ctx, lse, S, heads, \
                       key_bias, scale, static_cast<bf16_t*>(dqkv), ld_dqkv, drop, nqb, static_cast<unsigned long long*>(nullptr), \
                       static_cast<const unsigned*>(keep_bits), lp)
// dropout: from the forward's keep words when the caller has them, re-hashed otherwise; no dropout: delta preloaded (PRE).
// The LoRA partial products ride on the two forms the engines run (keep words / preloaded delta).
#define ATTN_BWD_PICK(NBV, TL)                                                                                   \
    do {                                                                                                         \
        if (lp.dtp && drop.thr16) ATTN_BWD_LAUNCH(NBV, true, TL, true, false, true);                             \
        else if (lp.dtp) ATTN_BWD_LAUNCH(NBV, false, TL, false, true, true);                                     \
        else if (drop.thr16 && keep_bits) ATTN_BWD_LAUNCH(NBV, true, TL, true, false, false);                    \
        else if (drop.thr16) ATTN_BWD_LAUNCH(NBV, true, TL, false, false, false);                                \
        else if (g_attn_preload) ATTN_BWD_LAUNCH(NBV, false, TL, false, true, false);                            \
        else ATTN_BWD_LAUNCH(NBV, false, TL, false, false, false);                                               \
    } while (0)
#define ATTN_BWD_CASE(NBV)                                                                                       \
    case NBV:                                                                                                    \
        ATTN_BWD_PICK(NBV, 32);                                                                                  \
        break;
// A/B switch of the round-5 "row constant as the initial accumulator" form of the no-dropout key-owner phase (default on)
static const bool g_attn_preload = !(getenv("BSCLIP_ATTN_PRELOAD") && atoi(getenv("BSCLIP_ATTN_PRELOAD")) == 0);

static int attn_bwd_launch(const void* qkv, int ld_qkv, const void* dctx, int ld_ctx, const float* lse, int B, int S, int heads,
                           const float* key_bias, float scale, void* dqkv, int ld_dqkv, int q_rows, const void* keep_bits,
                           float dropout_p, uint32_t dropout_seed, LoraPart lp, void* stream) {
    BSCLIP_REQUIRE(qkv && dctx && lse && dqkv, "bsclip_attn_bwd: null pointer");
    BSCLIP_REQUIRE((reinterpret_cast<uintptr_t>(keep_bits) & 15) == 0, "bsclip_attn_bwd: keep_bits must be 16-byte aligned");
    BSCLIP_REQUIRE(B > 0 && heads > 0 && S > 0 && S <= 224, "bsclip_attn_bwd: B=%d heads=%d S=%d (S <= 224)", B, heads, S);
    BSCLIP_REQUIRE(ld_qkv >= 3 * heads * 64 && ld_qkv % 8 == 0 && ld_dqkv >= 3 * heads * 64 && ld_dqkv % 4 == 0 &&
                       ld_ctx >= heads * 64 && ld_ctx % 8 == 0,
                   "bsclip_attn_bwd: ld_qkv=%d ld_dqkv=%d ld_ctx=%d", ld_qkv, ld_dqkv, ld_ctx);
    BSCLIP_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "bsclip_attn_bwd: dropout_p=%f", dropout_p);
    BSCLIP_REQUIRE(q_rows >= 0 && q_rows <= S, "bsclip_attn_bwd: q_rows=%d (0 = all, <= S)", q_rows);
    const int nqb = q_rows > 0 ? (q_rows + 31) / 32 : (S + 31) / 32;
    const DropCfg drop = make_drop(dropout_p, dropout_seed);
    BSCLIP_REQUIRE(!lp.dtp || !drop.thr16 || keep_bits, "bsclip_attn_bwd_lora: with dropout the forward's keep_bits are required");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // S = 133 gains 5 % from the trimmed last tile; at S = 197 the trimmed instantiation schedules worse (268.6 vs 265.0 us)
    if (S == 133) {
        ATTN_BWD_PICK(5, 5);
    } else {
        switch ((S + 31) / 32) {
            ATTN_BWD_CASE(1) ATTN_BWD_CASE(2) ATTN_BWD_CASE(3) ATTN_BWD_CASE(4) ATTN_BWD_CASE(5) ATTN_BWD_CASE(6)
            ATTN_BWD_CASE(7)
        }
    }
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int bsclip_attn_bwd(const void* qkv, int ld_qkv, const void* dctx, int ld_ctx, const float* lse, int B,
                               int S, int heads, const float* key_bias, float scale, void* dqkv, int ld_dqkv, int q_rows,
                               const void* keep_bits, float dropout_p, uint32_t dropout_seed, void* stream) {
    return attn_bwd_launch(qkv, ld_qkv, dctx, ld_ctx, lse, B, S, heads, key_bias, scale, dqkv, ld_dqkv, q_rows, keep_bits, dropout_p,
                           dropout_seed, LoraPart{}, stream);
}

// bsclip_attn_bwd that also leaves the per-head / per-item partial sums of the LoRA gradients (see LoraPart): bsclip_lora_grad_heads
// reduces them.  t_aug: the bf16 [B S, ld_t] block whose columns 0..7 hold t = y A^T (q ranks, then v ranks).
extern "C" int bsclip_attn_bwd_lora(const void* qkv, int ld_qkv, const void* dctx, int ld_ctx, const float* lse, int B, int S,
                                    int heads, const float* key_bias, float scale, void* dqkv, int ld_dqkv, int q_rows,
                                    const void* keep_bits, const void* t_aug, int ld_t, const float* lora_b, float* dt_partial,
                                    float* db_partial, float dropout_p, uint32_t dropout_seed, void* stream) {
    BSCLIP_REQUIRE(t_aug && lora_b && dt_partial && db_partial, "bsclip_attn_bwd_lora: null pointer");
    BSCLIP_REQUIRE(ld_t >= 8 && ld_t % 8 == 0 && (reinterpret_cast<uintptr_t>(t_aug) & 15) == 0 &&
                       (reinterpret_cast<uintptr_t>(dt_partial) & 15) == 0 && (reinterpret_cast<uintptr_t>(lora_b) & 3) == 0,
                   "bsclip_attn_bwd_lora: t_aug / dt_partial must be 16-byte aligned, ld_t=%d a multiple of 8", ld_t);
    return attn_bwd_launch(qkv, ld_qkv, dctx, ld_ctx, lse, B, S, heads, key_bias, scale, dqkv, ld_dqkv, q_rows, keep_bits, dropout_p,
                           dropout_seed, LoraPart{static_cast<const bf16_t*>(t_aug), ld_t, lora_b, dt_partial, db_partial}, stream);
}

#ifdef BSCLIP_DIAG
// Diagnostic build of the backward kernel (S = 197 / 133 instances): per-wave wall-clock stamps
// [start, K/V staged, phase 1 done, barrier passed, Q/dO staged, end] in diag[(B*heads) * 4 waves * 8].  Never used by the product.
extern "C" int bsclip_attn_bwd_diag(const void* qkv, int ld_qkv, const void* dctx, int ld_ctx, const float* lse, int B,
                                    int S, int heads, float scale, void* dqkv, int ld_dqkv, unsigned long long* diag,
                                    void* stream) {
    BSCLIP_REQUIRE(qkv && dctx && lse && dqkv && diag, "bsclip_attn_bwd_diag: null pointer");
    BSCLIP_REQUIRE(S == 197 || S == 133, "bsclip_attn_bwd_diag: S=%d (197 or 133)", S);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const DropCfg drop = make_drop(0.f, 0);
    if (S == 197)
        hipLaunchKernelGGL((attn_bwd_kernel<7, false, true>), dim3(B * heads), dim3(ATT_WAVES * 64), 0, s,
                           static_cast<const bf16_t*>(qkv), ld_qkv, static_cast<const bf16_t*>(dctx), ld_ctx, lse, S, heads,
                           (const float*)nullptr, scale, static_cast<bf16_t*>(dqkv), ld_dqkv, drop, 7, diag);
    else
        hipLaunchKernelGGL((attn_bwd_kernel<5, false, true>), dim3(B * heads), dim3(ATT_WAVES * 64), 0, s,
                           static_cast<const bf16_t*>(qkv), ld_qkv, static_cast<const bf16_t*>(dctx), ld_ctx, lse, S, heads,
                           (const float*)nullptr, scale, static_cast<bf16_t*>(dqkv), ld_dqkv, drop, 5, diag);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}
#endif  // BSCLIP_DIAG
