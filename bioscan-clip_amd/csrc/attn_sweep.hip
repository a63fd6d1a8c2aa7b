// Attention backward as ONE sweep over the query blocks (round 4; replaces the two-phase kernel of attn.hip for full-sequence
// gradients -- autograd of timm Attention.forward / HF BertSelfAttention, reached from image_encoder.py:108-109, dna_encoder.py:105,
// language_encoder.py:89).
//
// One workgroup per (batch, head), ONE barrier (tiles staged): NB key-owner waves + one dQ wave (NB = number of 32-key tiles, <= 7: S <= 224).
//   * Key-owner wave w keeps K_w, V_w (its 32 keys) as MFMA B operands and dK_w^T, dV_w^T in 64 accumulator registers for the
//     whole sweep.  Per 32-query block: S = Q K_w^T and dP = dO V_w^T with the KEY ON THE LANE (the accumulators are then
//     already the B operands of dV^T += dO^T P and dK^T += Q^T dS: cdna_hip_programming.md, "Attention backward"), the
//     probabilities and dS in registers, and the dS^T tile -- 2 KB of bf16 -- written to LDS for the dQ wave.
//   * The dQ wave runs one query block behind: dQ^T[d, q] = sum over ALL keys of K^T dS^T, both operands by transposing LDS reads
//     (ds_read_b64_tr_b16), 4 NB MFMAs per block; nothing is summed across waves or workgroups (no atomics).
//   * delta = rowsum(P dP) is NOT recomputed from the scores: delta_q = (dO_q . O_q) rZ_q with O to 16 mantissa bits and the
//     row statistics of the forward (bsclip_attn_fwd2).  The two-phase kernel formed S and dP twice (once for delta, once for
//     dK / dV) and dP a third time for dQ: 32 MFMAs + 32 exp per 32x32 tile pair; this one executes 20 + 16.
// Exactness of the softmax-backward cancellation (sum_k dS_k = 0): the forward's P.V product used pd_k = bf16(e_k keep_k); the
// same pd_k is re-formed here bit for bit (same score accumulation, same exp2 argument) and
//     dS_k = inv pd_k (dP_k - delta / keep_scale)   (kept keys),     dS_k = - inv e_k delta   (dropped keys)
// sums to inv [ (dO . O) / inv - delta Z' ] = 0 by the definition of Z' (attn_fwd_kernel, V2).
#include <stdlib.h>

#include "attn_common.h"

namespace {

constexpr int DS_ROWB = 72;            // a dS^T tile: [32 keys][32 queries] bf16, rows padded 64 -> 72 B (conflict-free ds_write_b64)
constexpr int DS_TILE = 32 * DS_ROWB;

template <int NB, bool DROP, bool DIAG = false>
__global__ __launch_bounds__((NB + 1) * 64) void attn_bwd_sweep_kernel(
    const bf16_t* __restrict__ qkv, int ld, const bf16_t* __restrict__ dctx, int ld_ctx, const bf16_t* __restrict__ ohi,
    const bf16_t* __restrict__ olo, int ld_o, const float* __restrict__ stats, int S, int heads,
    const float* __restrict__ key_bias, float scale, bf16_t* __restrict__ dqkv, int ld_d, DropCfg drop,
    unsigned long long* diag = nullptr) {
    constexpr int SP = NB * 32, NW = NB + 1, NT = NW * 64, RM = SP * ROWB;
    constexpr int RING = 4;   // dS^T slots: a key owner may run up to RING - 1 query blocks ahead of the dQ wave
    BSCLIP_DROP_RESOLVE(drop);
    auto stamp = [&](int i) {  // diagnostic build: per-wave section times (100 MHz wall clock); tools/attn_sweep_phases.py
        if constexpr (DIAG) {
            if ((threadIdx.x & 63) == 0) diag[((size_t)blockIdx.x * NW + (threadIdx.x >> 6)) * 8 + i] = wall_clock64();
        }
    };
    stamp(0);
    __shared__ __attribute__((aligned(16))) char smem[3 * RM + RING * NB * DS_TILE + 4 * SP * 4];
    // Hand-off words (LDS): sDone[r] = key-owner tiles written into ring slot r so far (block c is complete at NB (c / RING + 1)),
    // sCons = query blocks the dQ wave has finished.  The waves of a workgroup meet at ONE barrier (tiles staged); after it the key
    // owners run free of each other -- with a barrier per query block all eight waves did their matrix work, their vector work and
    // their LDS bursts at the same time (2 750 cycles per block, the SUM of the three) instead of beside each other.
    __shared__ unsigned sDone[RING], sCons, sReady[8];
    // Streaming (experiment, off): the prologue brings K and only the first LEAD query blocks of Q / dO / O; key-owner waves 0-3
    // ("committers") bring block c + LEAD while block c is being swept -- two LDS-DMA pieces and one delta task per lane each -- and
    // count it into sReady[block] (4 = complete).  Measured (profiles/r04_g_attn_sweep_phases.log): the workgroup starts after 5.2
    // instead of 9.0 us, but a block's sweep (1.2 us) is shorter than a load's round trip under load (2-4 us), so every commit
    // waits: 11.4 instead of 7.3 us for blocks 1..6 -- no gain.  Hiding the loads needs the NEXT head's tiles in flight (a
    // persistent workgroup with double-buffered tiles: 172 KB of LDS at S = 197), not this head's later blocks.
    constexpr bool STREAM = false;   // measured: no gain (see the note below); NB >= 5 enables it
    constexpr int LEAD = STREAM ? 2 : NB, NCOMMIT = 4;
    char* sQ = smem;
    char* sDO = smem + RM;
    char* sK = smem + 2 * RM;
    char* sDS = smem + 3 * RM;                                  // [RING][NB][DS_TILE]
    float* sNm2 = reinterpret_cast<float*>(sDS + RING * NB * DS_TILE);
    float* sInv = sNm2 + SP;
    float* sDel = sInv + SP;                                    // delta (DROP: delta / keep_scale)
    float* sNid = sDel + SP;                                    // DROP: -inv * delta (dropped keys)

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
    const float* st = stats + (size_t)bh * S * 4;

    if (tid < RING) sDone[tid] = 0u;
    if (tid == RING) sCons = 0u;
    if (tid >= 8 && tid < 16) sReady[tid - 8] = 0u;
    stage_tile<32 * LEAD, NW>(qb, ld, S, sQ, wave, lane);
    stage_tile<32 * LEAD, NW>(dob, ld_ctx, S, sDO, wave, lane);
    stage_tile<SP, NW>(kb, ld, S, sK, wave, lane);

    // a key owner's B operands (its 32 keys of K and V) and bias, requested before anything waits: they are first used a whole
    // staging phase later (the dQ wave, wave NB, loads a clamped duplicate it never uses: the branch stays out of the prologue)
    const int k0 = min(wave, NB - 1) * 32;
    const int key = k0 + (lane & 31);
    const int krow = min(key, S - 1);
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        kf[ks] = frag_global(kb, ld, krow, ks, lane);
        vf[ks] = frag_global(vb, ld, krow, ks, lane);
    }
    // the key's bias / scale as the score accumulator's start value, as in the forward; padded keys start from -inf
    const float bias_k = key < S ? (key_bias ? key_bias[(size_t)b * S + key] * (1.0f / scale) : 0.f) : -INFINITY;

    // ---- row statistics and delta_q = (dO_q . (O_hi + O_lo)_q) * rZ_q: 8 lanes per row, 8 columns each ----
    const bf16_t* oh = ohi + (size_t)b * S * ld_o + hd * 64;
    const bf16_t* ol = olo + (size_t)b * S * ld_o + hd * 64;
    {
        // every load of every task goes out before the first use: the loop is a handful of dependent HBM round trips otherwise
        constexpr int NTASK = (32 * LEAD * 8 + NT - 1) / NT;
        u32x4 a[NTASK], x[NTASK], y[NTASK];
        f32x4 s4[NTASK];
#pragma unroll
        for (int t = 0; t < NTASK; ++t) {
            const int task = tid + t * NT, row = task >> 3, c = task & 7;
            a[t] = x[t] = y[t] = u32x4{0u, 0u, 0u, 0u};
            s4[t] = f32x4{-INFINITY, 0.f, 0.f, 0.f};   // padded query rows: e = exp2(-inf) = 0
            if (task < 32 * LEAD * 8 && row < S) {
                a[t] = *reinterpret_cast<const u32x4*>(dob + (size_t)row * ld_ctx + c * 8);
                x[t] = *reinterpret_cast<const u32x4*>(oh + (size_t)row * ld_o + c * 8);
                y[t] = *reinterpret_cast<const u32x4*>(ol + (size_t)row * ld_o + c * 8);
                if (c == 0) s4[t] = *reinterpret_cast<const f32x4*>(st + (size_t)row * 4);
            }
        }
#pragma unroll
        for (int t = 0; t < NTASK; ++t) {
            const int task = tid + t * NT, row = task >> 3, c = task & 7;
            float part = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                part = fmaf(bf_lo(a[t][i]), bf_lo(x[t][i]) + bf_lo(y[t][i]), part);
                part = fmaf(bf_hi(a[t][i]), bf_hi(x[t][i]) + bf_hi(y[t][i]), part);
            }
            part += __shfl_xor(part, 1, 64);
            part += __shfl_xor(part, 2, 64);
            part += __shfl_xor(part, 4, 64);
            if (c == 0 && task < 32 * LEAD * 8) {
                const float delta = part * s4[t][2];
                sNm2[row] = s4[t][0];
                sInv[row] = s4[t][1];
                sDel[row] = DROP ? delta / drop.scale : delta;
                sNid[row] = -s4[t][1] * delta;
            }
        }
    }
    stamp(1);
    stage_wait();
    __syncthreads();
    stamp(2);
    const float scale2 = scale * LOG2E;

    if (wave < NB) {
        // ------------------------------------------------ key-owner wave: keys [k0, k0 + 32) ------------------------------------------------
        f32x16 bk16;
#pragma unroll
        for (int r = 0; r < 16; ++r) bk16[r] = bias_k;
        f32x16 dv[2] = {zero16(), zero16()}, dk[2] = {zero16(), zero16()};
        // S and dP of a query block: 8 MFMAs that depend on nothing the VALU section produces -- issued one block AHEAD, so that
        // the matrix pipe works through them while this wave's VALU section (exp, rounding, dS) of the current block issues.
        // (All waves of the workgroup meet at one barrier per block and would otherwise run matrix and vector work in lockstep.)
        auto scores = [&](int q0, f32x16& s_out, f32x16& dp_out) {
            s_out = bk16;
            dp_out = zero16();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s_out = mfma32(frag_rm(sQ, q0, ks, lane), kf[ks], s_out);       // S[q, key] + bias / scale
                dp_out = mfma32(frag_rm(sDO, q0, ks, lane), vf[ks], dp_out);   // dP[q, key] = dO_q . V_key
            }
        };
        f32x16 s, dp, s_n, dp_n;
        scores(0, s, dp);
#pragma unroll 1
        for (int qb_i = 0; qb_i < NB; ++qb_i) {
            asm volatile("" ::: "memory");   // the LDS tiles are loop-invariant: keep LICM from hoisting the fragment reads
            const int q0 = qb_i * 32;
            // committer: request block qb_i + LEAD (two 8-row pieces of its Q / dO slab by LDS-DMA, this lane's 8 columns of dO, O)
            const int xb = qb_i + LEAD;
            const bool bring = STREAM && wave < NCOMMIT && xb < NB;   // wave-uniform
            u32x4 la = {0u, 0u, 0u, 0u}, lx = la, ly = la;
            f32x4 ls4 = {-INFINITY, 0.f, 0.f, 0.f};
            const int lrow = 32 * xb + ((64 * wave + lane) >> 3), lc = lane & 7;
            if (bring) {
#pragma unroll
                for (int pi = 0; pi < 2; ++pi) {
                    const int p = 2 * wave + pi;                       // 0-3: Q rows 8 (p & 3) .., 4-7: dO rows
                    const int chunk = 4 * xb + (p & 3), row = 8 * chunk + (lane >> 3);
                    const int sc = (lane & 7) ^ tile_sw(row);
                    if (p < 4) glds16(qb + (size_t)min(row, S - 1) * ld + sc * 8, sQ + chunk * 1024);
                    else glds16(dob + (size_t)min(row, S - 1) * ld_ctx + sc * 8, sDO + chunk * 1024);
                }
                if (lrow < S) {
                    la = *reinterpret_cast<const u32x4*>(dob + (size_t)lrow * ld_ctx + lc * 8);
                    lx = *reinterpret_cast<const u32x4*>(oh + (size_t)lrow * ld_o + lc * 8);
                    ly = *reinterpret_cast<const u32x4*>(ol + (size_t)lrow * ld_o + lc * 8);
                    if (lc == 0) ls4 = *reinterpret_cast<const f32x4*>(st + (size_t)lrow * 4);
                }
            }
            if (STREAM && qb_i + 1 >= LEAD && qb_i + 1 < NB) {   // the next block's slab (for the MFMAs issued now) must have landed
                while (__hip_atomic_load(&sReady[qb_i + 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)NCOMMIT)
                    __builtin_amdgcn_s_sleep(1);
            }
            scores(min(qb_i + 1, NB - 1) * 32, s_n, dp_n);   // the last block recomputes itself (unused): no branch in the loop
            __builtin_amdgcn_sched_barrier(0);               // ... and the compiler may not sink them below the VALU section
            if (qb_i >= RING) {                              // the slot's previous tenant (block qb_i - RING) must have been consumed
                while (__hip_atomic_load(&sCons, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)(qb_i - RING + 1))
                    __builtin_amdgcn_s_sleep(1);
            }
            char* slot = sDS + ((qb_i % RING) * NB + wave) * DS_TILE;
            unsigned pw[8], dw[8];   // packed bf16 pairs (registers 2i, 2i+1): inv pd (-> dV) and dS (-> dK, dQ)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r0 = q0 + 8 * g + 4 * h;
                const f32x4 n4 = *reinterpret_cast<const f32x4*>(sNm2 + r0);
                const f32x4 i4 = *reinterpret_cast<const f32x4*>(sInv + r0);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(sDel + r0);
                f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                if constexpr (DROP) z4 = *reinterpret_cast<const f32x4*>(sNid + r0);
                float e[4], kp[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    e[i] = __builtin_amdgcn_exp2f(fmaf(s[4 * g + i], scale2, n4[i]));
                    kp[i] = 1.f;
                    if constexpr (DROP) {
                        const int q = min(r0 + i, S - 1);
                        kp[i] = drop_factor(drop, (bh * S + (unsigned)q) * SP + (unsigned)key);
                    }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const unsigned w = DROP ? pack_bf2(e[2 * j] * kp[2 * j], e[2 * j + 1] * kp[2 * j + 1]) : pack_bf2(e[2 * j], e[2 * j + 1]);
                    const float t0 = bf_lo(w) * i4[2 * j], t1 = bf_hi(w) * i4[2 * j + 1];
                    float ds0 = t0 * (dp[4 * g + 2 * j] - d4[2 * j]), ds1 = t1 * (dp[4 * g + 2 * j + 1] - d4[2 * j + 1]);
                    if constexpr (DROP) {
                        ds0 = kp[2 * j] != 0.f ? ds0 : e[2 * j] * z4[2 * j];
                        ds1 = kp[2 * j + 1] != 0.f ? ds1 : e[2 * j + 1] * z4[2 * j + 1];
                    }
                    pw[2 * g + j] = pack_bf2(t0, t1);
                    dw[2 * g + j] = pack_bf2(ds0, ds1);
                }
                // dS^T[key][q0 + 8g + 4h .. + 3] -> the dQ wave's tile (row = key, 8 B per group)
                *reinterpret_cast<uint2*>(slot + (lane & 31) * DS_ROWB + (8 * g + 4 * h) * 2) = uint2{dw[2 * g], dw[2 * g + 1]};
            }
            // tile written (the release orders this wave's ds_writes before the count): one more of the NB tiles of block qb_i
            if (lane == 0) __hip_atomic_fetch_add(&sDone[qb_i % RING], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pb = __builtin_bit_cast(bf16x8, u32x4{pw[4 * s2], pw[4 * s2 + 1], pw[4 * s2 + 2], pw[4 * s2 + 3]});
                const bf16x8 dsb = __builtin_bit_cast(bf16x8, u32x4{dw[4 * s2], dw[4 * s2 + 1], dw[4 * s2 + 2], dw[4 * s2 + 3]});
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[dt] = mfma32(frag_tr(sDO, 32 * dt, q0 + 16 * s2, lane), pb, dv[dt]);   // dV^T += dO^T P
                    dk[dt] = mfma32(frag_tr(sQ, 32 * dt, q0 + 16 * s2, lane), dsb, dk[dt]);   // dK^T += Q^T dS
                }
            }
            if (bring) {   // commit block xb: everything this wave requested for it has arrived (it has nothing else in flight)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                float part = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    part = fmaf(bf_lo(la[i]), bf_lo(lx[i]) + bf_lo(ly[i]), part);
                    part = fmaf(bf_hi(la[i]), bf_hi(lx[i]) + bf_hi(ly[i]), part);
                }
                part += __shfl_xor(part, 1, 64);
                part += __shfl_xor(part, 2, 64);
                part += __shfl_xor(part, 4, 64);
                if (lc == 0) {
                    const float delta = part * ls4[2];
                    sNm2[lrow] = ls4[0];
                    sInv[lrow] = ls4[1];
                    sDel[lrow] = DROP ? delta / drop.scale : delta;
                    sNid[lrow] = -ls4[1] * delta;
                }
                if (lane == 0) __hip_atomic_fetch_add(&sReady[xb], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            s = s_n;
            dp = dp_n;
            if (qb_i == 0) stamp(3);
        }
        stamp(4);
        if (key < S) {
            store_dt(dk, scale, dqb + (size_t)key * ld_d + HW, lane);
            store_dt(dv, 1.0f, dqb + (size_t)key * ld_d + 2 * HW, lane);
        }
        stamp(5);
    } else {
        // ------------------------------------------------ dQ wave: follows the key owners block by block ------------------------------------------------
        // K^T fragments are the same for every query block: read once (transposing LDS reads), 16 NB registers
        bf16x8 ktf[NB][2][2];
#pragma unroll
        for (int kt = 0; kt < NB; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) ktf[kt][s2][dt] = frag_tr(sK, 32 * dt, 32 * kt + 16 * s2, lane);
#pragma unroll 1
        for (int qb_i = 0; qb_i < NB; ++qb_i) {
            const unsigned want = (unsigned)(NB * (qb_i / RING + 1));
            while (__hip_atomic_load(&sDone[qb_i % RING], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < want)
                __builtin_amdgcn_s_sleep(1);
            if (qb_i == 0) stamp(3);
            if (qb_i == NB - 1) stamp(4);
            const char* tiles = sDS + (qb_i % RING) * NB * DS_TILE;
            f32x16 dq[2] = {zero16(), zero16()};
#pragma unroll
            for (int kt = 0; kt < NB; ++kt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 dsb = frag_tr_lin(tiles + kt * DS_TILE, DS_ROWB, 0, 16 * s2, lane);   // dS^T[key, q], k = key
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) dq[dt] = mfma32(ktf[kt][s2][dt], dsb, dq[dt]);      // dQ^T += K^T dS^T
                }
            // the slot is free once its tiles are in registers (the MFMAs above have consumed them: their operands were waited for)
            if (lane == 0) __hip_atomic_fetch_add(&sCons, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int q = qb_i * 32 + (lane & 31);
            if (q < S) store_dt(dq, scale, dqb + (size_t)q * ld_d, lane);
        }
        stamp(5);
    }
}

}  // namespace

// attn_pers.hip: the persistent form (next item's tiles prefetched under the current sweep) for the production sequence lengths
bool bsclip_attn_bwd_pers_launch(const void* qkv, int ld_qkv, const void* dctx, int ld_dctx, const void* ctx, const void* ctx_lo,
                                 int ld_ctx, const float* stats, int B, int S, int heads, float scale, void* dqkv, int ld_dqkv,
                                 const DropCfg& drop, hipStream_t s);
static const bool g_attn_pers = !(getenv("BSCLIP_ATTN_PERS") && atoi(getenv("BSCLIP_ATTN_PERS")) == 0);   // A/B switch

#define ATTN_SWEEP_LAUNCH(NBV, DR)                                                                                      \
    hipLaunchKernelGGL((attn_bwd_sweep_kernel<NBV, DR>), dim3(B * heads), dim3((NBV + 1) * 64), 0, s,                    \
                       static_cast<const bf16_t*>(qkv), ld_qkv, static_cast<const bf16_t*>(dctx), ld_dctx,               \
                       static_cast<const bf16_t*>(ctx), static_cast<const bf16_t*>(ctx_lo), ld_ctx, stats, S, heads,     \
                       key_bias, scale, static_cast<bf16_t*>(dqkv), ld_dqkv, drop)
#define ATTN_SWEEP_CASE(NBV)                              \
    case NBV:                                             \
        if (drop.thr16) ATTN_SWEEP_LAUNCH(NBV, true);     \
        else ATTN_SWEEP_LAUNCH(NBV, false);               \
        break;

// Backward of bsclip_attn_fwd2 (same qkv, key_bias, scale, dropout arguments; ctx / ctx_lo / stats as that call left them):
// dqkv[B*S, 3*heads*64] = (dQ | dK | dV).  Every query row receives a gradient (no q_rows form: the last ViT block, whose
// gradient enters through token 0 only, stays on bsclip_attn_fwd / bsclip_attn_bwd).
extern "C" int bsclip_attn_bwd2(const void* qkv, int ld_qkv, const void* dctx, int ld_dctx, const void* ctx, const void* ctx_lo,
                                int ld_ctx, const float* stats, int B, int S, int heads, const float* key_bias, float scale,
                                void* dqkv, int ld_dqkv, float dropout_p, uint32_t dropout_seed, void* stream) {
    BSCLIP_REQUIRE(qkv && dctx && ctx && ctx_lo && stats && dqkv, "bsclip_attn_bwd2: null pointer");
    BSCLIP_REQUIRE(B > 0 && heads > 0 && S > 0 && S <= 224, "bsclip_attn_bwd2: B=%d heads=%d S=%d (S <= 224)", B, heads, S);
    BSCLIP_REQUIRE(ld_qkv >= 3 * heads * 64 && ld_qkv % 8 == 0 && ld_dqkv >= 3 * heads * 64 && ld_dqkv % 4 == 0 &&
                       ld_dctx >= heads * 64 && ld_dctx % 8 == 0 && ld_ctx >= heads * 64 && ld_ctx % 8 == 0,
                   "bsclip_attn_bwd2: ld_qkv=%d ld_dqkv=%d ld_dctx=%d ld_ctx=%d", ld_qkv, ld_dqkv, ld_dctx, ld_ctx);
    BSCLIP_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "bsclip_attn_bwd2: dropout_p=%f", dropout_p);
    BSCLIP_REQUIRE((reinterpret_cast<uintptr_t>(stats) & 15) == 0, "bsclip_attn_bwd2: stats must be 16-byte aligned");
    const DropCfg drop = make_drop(dropout_p, dropout_seed);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (g_attn_pers && !key_bias &&
        bsclip_attn_bwd_pers_launch(qkv, ld_qkv, dctx, ld_dctx, ctx, ctx_lo, ld_ctx, stats, B, S, heads, scale, dqkv, ld_dqkv, drop, s)) {
        BSCLIP_LAUNCH_CHECK();
        return BSCLIP_OK;
    }
    switch ((S + 31) / 32) {
        ATTN_SWEEP_CASE(1) ATTN_SWEEP_CASE(2) ATTN_SWEEP_CASE(3) ATTN_SWEEP_CASE(4) ATTN_SWEEP_CASE(5) ATTN_SWEEP_CASE(6)
        ATTN_SWEEP_CASE(7)
    }
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

#ifdef BSCLIP_DIAG
// Diagnostic build of the sweep kernel (S = 197 / 133, no dropout): per-wave wall-clock stamps [start, delta loads issued + computed,
// tiles landed, first query block done, last query block done (dQ wave: last barrier passed), end] in diag[(B*heads) * (NB+1) * 8].
extern "C" int bsclip_attn_bwd2_diag(const void* qkv, int ld_qkv, const void* dctx, int ld_dctx, const void* ctx, const void* ctx_lo,
                                     int ld_ctx, const float* stats, int B, int S, int heads, float scale, void* dqkv, int ld_dqkv,
                                     unsigned long long* diag, void* stream) {
    BSCLIP_REQUIRE(qkv && dctx && ctx && ctx_lo && stats && dqkv && diag, "bsclip_attn_bwd2_diag: null pointer");
    BSCLIP_REQUIRE(S == 197 || S == 133, "bsclip_attn_bwd2_diag: S=%d (197 or 133)", S);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const DropCfg drop = make_drop(0.f, 0);
    const float* key_bias = nullptr;
    if (S == 197)
        hipLaunchKernelGGL((attn_bwd_sweep_kernel<7, false, true>), dim3(B * heads), dim3(8 * 64), 0, s, static_cast<const bf16_t*>(qkv),
                           ld_qkv, static_cast<const bf16_t*>(dctx), ld_dctx, static_cast<const bf16_t*>(ctx),
                           static_cast<const bf16_t*>(ctx_lo), ld_ctx, stats, S, heads, key_bias, scale, static_cast<bf16_t*>(dqkv),
                           ld_dqkv, drop, diag);
    else
        hipLaunchKernelGGL((attn_bwd_sweep_kernel<5, false, true>), dim3(B * heads), dim3(6 * 64), 0, s, static_cast<const bf16_t*>(qkv),
                           ld_qkv, static_cast<const bf16_t*>(dctx), ld_dctx, static_cast<const bf16_t*>(ctx),
                           static_cast<const bf16_t*>(ctx_lo), ld_ctx, stats, S, heads, key_bias, scale, static_cast<bf16_t*>(dqkv),
                           ld_dqkv, drop, diag);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}
#endif  // BSCLIP_DIAG
