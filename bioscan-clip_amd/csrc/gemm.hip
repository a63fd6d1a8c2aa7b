// bf16 MFMA GEMM family for gfx950:  C = epilogue(A[M,K] * B[N,K]^T)
//
// Takes over every torch.nn.Linear (and its dX autograd pass) on the contrastive-training path -- see
// include/bsclip.h for the reference call sites.  Design (MI355X_MICROARCH / cdna_hip_programming guides):
//   * BK = 64 K-tiles staged HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip), double buffered;
//   * LDS image is lane-linear per wave instruction (8 rows x 128 B); the 16-B chunk index is XOR-swizzled with
//     (row>>1)&7 on the SOURCE address and again on the ds_read_b128 address, which makes every 16-lane
//     ds_read_b128 group hit 16 distinct 16-B slots of the 256-B bank row (conflict-free);
//   * v_mfma_f32_16x16x32_bf16 with the operands swapped (weights as the MFMA "A" side) so each lane ends up
//     holding 4 consecutive output columns of one row -> 8-B (bf16) / 16-B (f32) epilogue stores;
//   * XCD-aware bijective blockIdx remap: the 8 XCDs each walk a contiguous range of tiles, N fastest, so the
//     A row-panel and the (small) weight matrix are re-used out of that XCD's private L2;
//   * epilogues fused in registers: bias, exact GELU (+ saved pre-activation), residual add in f32, GELU'
//     scaling for the backward pass (the forward saves gelu'(pre-activation), so the backward epilogue is one multiply),
//     and the ViT patch-embed row remap + position add.  Bias is loaded once per
//     thread before the stores and every epilogue is branch-free per element (a per-element "if (bias)" makes
//     hipcc wait vmcnt(0) around each load: 32 serial L2 round trips per tile).
//
// Two main loops:
//   gemm_nt_kernel     generic BMxBN tile, one barrier per K-tile (prefetch of tile t+1 behind the MFMAs of tile t);
//                      used for small / odd shapes (128x128, 256x128).
//   gemm_nt_pp_kernel  256x256 tile, 8 waves = two groups of four that run half a phase apart ("ping-pong"): each
//                      K-tile is four phases {ds_read fragments + issue one half-tile of LDS-DMA | barrier | 16 MFMA |
//                      barrier}; while one group's waves are in their MFMA segment their SIMD partners (other group)
//                      are in the LDS/DMA segment.  LDS-DMA stays in flight across barriers: one counted
//                      s_waitcnt vmcnt(2) per K-tile (never 0 in the steady state), A halves issued 3-4 phases ahead.
#include <type_traits>

#include "common.h"

namespace {

constexpr int BK = 64;            // K tile (bf16 elements) = 128 B per row
constexpr int ROW_BYTES = BK * 2;  // 128

__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

struct EpiArgs {
    const float* bias;
    const float* resid;   // RESID_BF16: bf16 data behind the same pointer
    int ld_resid;
    unsigned char* aux;  // gelu' side band, 8-bit codes (common.h dg8_*)
    int ld_aux;
    DropCfg drop;  // RESID only: C = dropout(acc + bias) + resid  (HF BertSelfOutput / BertOutput)
    int n_total;   // logical row width for the dropout element index
    unsigned long long* diag;  // diagnostic build only: per-workgroup phase stamps (100 MHz wall clock)
    int pers_gw;               // persistent kernel: column tiles per super-column of its tile order (set by launch_pers)
    // fp8 main operands (OP != 0): C = epilogue(alpha[n] * (A8 . B8^T + A_aug . B_aug^T) + bias)
    const float* alpha;                   // [N] dequantisation scale of output column n (activation scale x weight-row scale)
    const bf16_t* a_aug;                  // bf16 [M, 64] K-augmentation block (LoRA t columns), nullable
    const bf16_t* b_aug;                  // bf16 [N, 64] matching weight columns (LoRA B / alpha[n])
    int ld_a_aug, ld_b_aug;
    // fused InfoNCE epilogues (EPI_LSE_PART / EPI_LOSS_W, internal: bsclip_gemm_infonce): the accumulator tile holds cosines
    const int64_t* labels;                // [>= n_valid]
    const float* cnt;                     // cnt[i] = #{j : label_j == label_i}
    const float* lse_row;                 // LSE of the rows of THIS product (a -> b), indexed by global row
    const float* lse_col;                 // LSE of the transposed product (b -> a), indexed by column
    float* part;                          // EPI_LSE_PART out: [M, tiles_n, 4] = (row max, sum exp, sum_j T_ij x_ij, -)
    float logit_scale, coef;              // x = logit_scale * cosine;  w = coef * (...)
    int n_valid, row_base;                // columns >= n_valid are padding; global row of local row 0
    // split-K (bsclip_gemm_splitk_f32; EPI_F32 without bias only): workgroups with blockIdx.y = z reduce K columns
    // [z K, (z + 1) K) of the operands into their own f32 slab C + z * split_stride
    long split_stride;
};
constexpr int EPI_LSE_PART = 100;  // internal epilogues, not part of the public enum
constexpr int EPI_LOSS_W = 101;

// ---------------------------------------------------------------------------------------------------------------
// GELU table for the 256x256 kernel's epilogue
// ---------------------------------------------------------------------------------------------------------------
// The two-output GELU epilogue (gelu and gelu' of 128 values per lane) computed with exp + rcp was VALU-bound (~24
// instructions per value, 12 us per tile with no MFMA to hide behind), and a 16-B-per-entry interpolation table was bound by
// the LDS array instead: 64 lanes gathering random 16-B entries conflict ~3x per 16-lane group (phase stamps: 4.4 us per
// 64-row slab).  Entries are therefore 8 B -- {Phi(x_i), phi(x_i)}, nearest grid point, one ds_read_b64 per value -- and the
// neighbourhood comes from the derivatives, which are free: Phi' = phi, phi' = -x phi.
// Round 3: the epilogue is still VALU-bound (8.8 us of a 30-us fc1 tile), so the index arithmetic went on a diet.  256 entries on
// a 1/16 grid over [-8, 8): the index is ONE v_cvt_pk_u8_f32 (round to nearest even, saturating at 0 / 255) of t = 16 x + 128
// -- no clamp, no float->int->float round trip beyond v_cvt_f32_ubyte0 -- and saturated inputs are harmless because
// phi(+-8) ~ 5e-15 multiplies whatever distance they have from the last grid point.  With d = x - x_i, |d| <= 1/32:
//   Phi(x_i + d) = Phi_i + d phi_i (1 - x_i d / 2) + O(d^3 |phi''| / 6) <= 2e-6   (first order alone leaves 1.2e-4: a third of
//                                                                       the e4m3 / a tenth of the bf16 rounding of the output),
//   phi(x_i + d) = phi_i (1 - x_i d)               + O(d^2 phi / 2)     <= 2e-4   (8-bit code step of gelu': 4.9e-3).
// Computed once on the device in f64 with erf().
constexpr int GELU_LUT_N = 255;                       // last index
constexpr int GELU_LUT_BYTES = (GELU_LUT_N + 1) * 8;  // 2048
__device__ float2 g_gelu_lut[GELU_LUT_N + 1];

__global__ void gelu_lut_init_kernel() {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > GELU_LUT_N) return;
    const double x = -8.0 + i / 16.0;
    g_gelu_lut[i] = float2{(float)(0.5 * (1.0 + erf(x * 0.70710678118654752440))),
                           (float)(0.39894228040143267794 * exp(-0.5 * x * x))};
}

// Two values per call so that the arithmetic runs on the packed-f32 VALU (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32); the
// index conversions and the table address stay per value.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_lut2(const char* lut, f32x2 x, f32x2& gl, f32x2& dg) {
    const f32x2 t = x * 16.0f + 128.0f;
    f32x2 Phi, phi;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const unsigned i = __builtin_amdgcn_cvt_pk_u8_f32(t[k], 0, 0u);   // nearest grid point (RNE), saturated to [0, 255]
        const float2 e = *reinterpret_cast<const float2*>(lut + i * 8);
        Phi[k] = e.x;
        phi[k] = e.y;
    }
    // the grid point as a float without a per-value conversion: (t + 1.5 * 2^23) - 1.5 * 2^23 = RNE(t) for |t| < 2^22, packed.
    // A saturated index keeps its small d here and takes its table entry from the end of the grid, where phi ~ 5e-15.
    f32x2 fi = t + 12582912.0f;
    asm volatile("" : "+v"(fi));   // keep the two additions apart
    fi -= 12582912.0f;
    const f32x2 d16 = t - fi;                       // 16 (x - x_i)
    const f32x2 e = d16 * phi;                      // 16 d phi_i
    const f32x2 xd = (fi * 0.00390625f - 0.5f) * d16;   // x_i d  (x_i / 16 = i / 256 - 1 / 2)
    const f32x2 cdf = e * (0.0625f - xd * 0.03125f) + Phi;   // Phi_i + d phi_i (1 - x_i d / 2)
    const f32x2 pdf = phi - xd * phi;                        // phi_i (1 - x_i d)
    gl = x * cdf;
    dg = x * pdf + cdf;
}

// 4 NV values per call: ALL the table gathers are issued before the first result is used.  Written pair by pair (above)
// the compiler waits for each pair's two ds_read_b64 before the next pair's go out -- 64 serial LDS round trips per 128 values and
// lane with two waves per SIMD to hide them: the GELU epilogue was bound by LDS LATENCY (4.8 us per 64-row slab; cutting its VALU
// count by a fifth changed nothing), not by the VALU or the LDS array.
template <int NV>   // NV vectors of four values
__device__ __forceinline__ void gelu_lut_batch(const char* lut, const f32x4 (&x)[NV], f32x4 (&gl)[NV], f32x4 (&dg)[NV]) {
    f32x4 t[NV], Phi[NV], phi[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) t[q] = x[q] * 16.0f + 128.0f;
#pragma unroll
    for (int q = 0; q < NV; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned i = __builtin_amdgcn_cvt_pk_u8_f32(t[q][c], 0, 0u);   // nearest grid point (RNE), saturated to [0, 255]
            const float2 e = *reinterpret_cast<const float2*>(lut + i * 8);
            Phi[q][c] = e.x;
            phi[q][c] = e.y;
        }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        // the grid point as a float without a per-value conversion: (t + 1.5 * 2^23) - 1.5 * 2^23 = RNE(t) for |t| < 2^22, packed.
        // A saturated index keeps its small d here and takes its table entry from the end of the grid, where phi ~ 5e-15.
        f32x4 fi = t[q] + 12582912.0f;
        asm volatile("" : "+v"(fi));   // keep the two additions apart
        fi -= 12582912.0f;
        const f32x4 d16 = t[q] - fi;                       // 16 (x - x_i)
        const f32x4 e = d16 * phi[q];                      // 16 d phi_i
        const f32x4 xd = (fi * 0.00390625f - 0.5f) * d16;  // x_i d  (x_i / 16 = i / 256 - 1 / 2)
        const f32x4 cdf = e * (0.0625f - xd * 0.03125f) + Phi[q];   // Phi_i + d phi_i (1 - x_i d / 2)
        const f32x4 pdf = phi[q] - xd * phi[q];                     // phi_i (1 - x_i d)
        gl[q] = x[q] * cdf;
        dg[q] = x[q] * pdf + cdf;
    }
}

__device__ __forceinline__ f32x4 bf4_to_f32(uint2 u) {
    return f32x4{bf2f(u.x & 0xffff), bf2f(u.x >> 16), bf2f(u.y & 0xffff), bf2f(u.y >> 16)};
}
__device__ __forceinline__ uint2 f32_to_bf4(f32x4 v) { return uint2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])}; }
constexpr bool epi_is_resid(int epi) { return epi == BSCLIP_EPI_RESID_F32 || epi == BSCLIP_EPI_RESID_BF16; }
constexpr bool epi_is_patch(int epi) { return epi == BSCLIP_EPI_PATCH_F32 || epi == BSCLIP_EPI_PATCH_BF16; }
constexpr bool epi_out_bf16_from_f32_slab(int epi) {
    return epi == BSCLIP_EPI_DGELU_BF16 || epi == BSCLIP_EPI_RESID_BF16 || epi == BSCLIP_EPI_PATCH_BF16;
}

// v already holds acc (+ bias).  No data-dependent branch guards a load.
template <int EPI>
__device__ __forceinline__ void epilogue_store(f32x4 v, int m, int n, void* C, int ldc, const EpiArgs& e) {
    if constexpr (EPI == BSCLIP_EPI_BF16) {
        uint2 o;
        o.x = pack_bf2(v[0], v[1]);
        o.y = pack_bf2(v[2], v[3]);
        *reinterpret_cast<uint2*>(static_cast<bf16_t*>(C) + (size_t)m * ldc + n) = o;
    } else if constexpr (EPI == BSCLIP_EPI_F32) {
        *reinterpret_cast<f32x4*>(static_cast<float*>(C) + (size_t)m * ldc + n) = v;
    } else if constexpr (EPI == BSCLIP_EPI_GELU_BF16) {
        // the same table-driven GELU as the 256x256 kernel (table read from global memory here: 8 KB, cache-resident), so a
        // row's result does not depend on which tile shape its batch size selects (tests/test_70_configs_gpu.py)
        f32x2 g0, d0, g1, d1;
        gelu_lut2(reinterpret_cast<const char*>(g_gelu_lut), f32x2{v[0], v[1]}, g0, d0);
        gelu_lut2(reinterpret_cast<const char*>(g_gelu_lut), f32x2{v[2], v[3]}, g1, d1);
        const float gl[4] = {g0[0], g0[1], g1[0], g1[1]}, dg[4] = {d0[0], d0[1], d1[0], d1[1]};
        if (e.aux)  // store only: gelu'(pre-activation) as 8-bit codes, all the backward pass needs
            *reinterpret_cast<unsigned*>(e.aux + (size_t)m * e.ld_aux + n) = dg8_pack4(dg[0], dg[1], dg[2], dg[3]);
        uint2 o;
        o.x = pack_bf2(gl[0], gl[1]);
        o.y = pack_bf2(gl[2], gl[3]);
        *reinterpret_cast<uint2*>(static_cast<bf16_t*>(C) + (size_t)m * ldc + n) = o;
    } else if constexpr (EPI == BSCLIP_EPI_RESID_F32) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(e.resid + (size_t)m * e.ld_resid + n);
        if (e.drop.thr16) v = drop4(e.drop, (unsigned)m * (unsigned)e.n_total + (unsigned)n, v);
        v += r;
        *reinterpret_cast<f32x4*>(static_cast<float*>(C) + (size_t)m * ldc + n) = v;
    } else if constexpr (EPI == BSCLIP_EPI_DGELU_BF16) {
        const f32x4 z = dg8_unpack4(*reinterpret_cast<const unsigned*>(e.aux + (size_t)m * e.ld_aux + n));
        uint2 o;
        o.x = pack_bf2(v[0] * z[0], v[1] * z[1]);
        o.y = pack_bf2(v[2] * z[2], v[3] * z[3]);
        *reinterpret_cast<uint2*>(static_cast<bf16_t*>(C) + (size_t)m * ldc + n) = o;
    } else if constexpr (EPI == BSCLIP_EPI_PATCH_F32) {
        const int b = m / 196, p = m - b * 196;
        const f32x4 pos = *reinterpret_cast<const f32x4*>(e.resid + (size_t)(1 + p) * e.ld_resid + n);
        v += pos;
        *reinterpret_cast<f32x4*>(static_cast<float*>(C) + (size_t)(b * 197 + 1 + p) * ldc + n) = v;
    } else if constexpr (EPI == BSCLIP_EPI_RESID_BF16) {
        const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(e.resid) + (size_t)m * e.ld_resid + n);
        if (e.drop.thr16) v = drop4(e.drop, (unsigned)m * (unsigned)e.n_total + (unsigned)n, v);
        v += bf4_to_f32(r);
        *reinterpret_cast<uint2*>(static_cast<bf16_t*>(C) + (size_t)m * ldc + n) = f32_to_bf4(v);
    } else if constexpr (EPI == BSCLIP_EPI_PATCH_BF16) {
        const int b = m / 196, p = m - b * 196;
        const f32x4 pos = *reinterpret_cast<const f32x4*>(e.resid + (size_t)(1 + p) * e.ld_resid + n);
        v += pos;
        *reinterpret_cast<uint2*>(static_cast<bf16_t*>(C) + (size_t)(b * 197 + 1 + p) * ldc + n) = f32_to_bf4(v);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// generic tile, one barrier per K-tile
// ---------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, bool HAS_BIAS>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_nt_kernel(const bf16_t* __restrict__ A, int lda,
                                                                         const bf16_t* __restrict__ B, int ldb,
                                                                         void* __restrict__ C, int ldc, int M, int N,
                                                                         int K, int tiles_n, EpiArgs e) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;  // per-wave output tile
    if constexpr (epi_is_resid(EPI)) BSCLIP_DROP_RESOLVE(e.drop);
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int A_PER_WAVE = BM / 8 / NW, B_PER_WAVE = BN / 8 / NW;
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "staging split");
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = wg % tiles_n, tile_m = wg / tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // ---- staging sources: wave-instruction q covers tile rows [8q, 8q+8); lane -> (row, swizzled chunk) ----
    const bf16_t* a_src[A_PER_WAVE];
    const bf16_t* b_src[B_PER_WAVE];
#pragma unroll
    for (int i = 0; i < A_PER_WAVE; ++i) {
        const int row = (wave + i * NW) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const int grow = min(m0 + row, M - 1);
        a_src[i] = A + (size_t)grow * lda + c * 8;
    }
#pragma unroll
    for (int i = 0; i < B_PER_WAVE; ++i) {
        const int row = (wave + i * NW) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const int grow = min(n0 + row, N - 1);
        b_src[i] = B + (size_t)grow * ldb + c * 8;
    }
    auto stage = [&](int buf, int k0) {
        char* sA = smem + buf * STAGE_BYTES;
        char* sB = sA + A_BYTES;
#pragma unroll
        for (int i = 0; i < A_PER_WAVE; ++i) glds16(a_src[i] + k0, sA + (wave + i * NW) * 1024);
#pragma unroll
        for (int i = 0; i < B_PER_WAVE; ++i) glds16(b_src[i] + k0, sB + (wave + i * NW) * 1024);
    };

    // ---- fragment read offsets (bytes) within a tile ----
    const int fr = lane & 15, fq = lane >> 4;
    const int sw = fr >> 1;  // (row>>1)&7 for rows that are 16-aligned + fr
    const int a_off0 = (wm * WTM + fr) * ROW_BYTES + ((fq ^ sw) << 4);        // ks = 0
    const int b_off0 = (wn * WTN + fr) * ROW_BYTES + ((fq ^ sw) << 4);
    // ks = 1 adds chunk 4: (4 + fq) ^ sw == (fq ^ sw) ^ 4  -> byte offset ^ 64

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    stage(0, 0);
    __syncthreads();  // drains the LDS-DMA (vmcnt(0)) and publishes the tile

    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        if (t + 1 < nk) stage(cur ^ 1, (t + 1) * BK);
        const char* sA = smem + cur * STAGE_BYTES;
        const char* sB = sA + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[TM], bfr[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bfr[j] = *reinterpret_cast<const bf16x8*>(sB + ((b_off0 ^ (ks << 6)) + j * 16 * ROW_BYTES));
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const bf16x8*>(sA + ((a_off0 ^ (ks << 6)) + i * 16 * ROW_BYTES));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: lane holds C[m][n..n+3] with m = ... + (lane&15), n = ... + (lane>>4)*4 ----
    f32x4 bias[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        bias[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (HAS_BIAS) bias[j] = *reinterpret_cast<const f32x4*>(e.bias + n0 + wn * WTN + j * 16 + fq * 4);
    }
    // bias is consumed before the row guards: a loaded register live across a divergent branch makes hipcc wait
    // vmcnt(0) (which also drains the previous row's stores) at the top of every guarded block
    if constexpr (HAS_BIAS) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] += bias[j];
    }
    // Row guards only on the last (partial) M tile: per-row divergent branches make hipcc drain vmcnt(0) -- loads AND
    // the previous row's stores -- at the top of every guarded block.
    auto store_rows = [&](auto guard) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * WTM + i * 16 + fr;
            if (!decltype(guard)::value || m < M) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + wn * WTN + j * 16 + fq * 4;
                    epilogue_store<EPI>(acc[i][j], m, n, C, ldc, e);
                }
            }
        }
    };
    if (m0 + BM <= M) store_rows(std::false_type{});
    else store_rows(std::true_type{});
}

// ---------------------------------------------------------------------------------------------------------------
// 256x256 ping-pong kernel
// ---------------------------------------------------------------------------------------------------------------
// LDS: 2 buffer sets x {A[256][64], B[256][64]} bf16 = 128 KiB; a "half" is 128 rows (16 KiB) = 16 wave-instructions of
// LDS-DMA, two per wave.  Wave w: group g = w>>2 owns output rows [128g, 128g+128) (so it reads only A-half g),
// wc = w&3 owns output columns [64wc, 64wc+64) (B-half wc>>1).  Per K-tile the wave holds ALL its fragments in
// registers (A: 2 blocks of 64 rows, B: 2 blocks of 32 columns = 96 VGPRs), loaded in phases 0-2:
//   phase 0: read A(m0), B(n0) -> MFMA (m0,n0)      DMA: A-half1 of tile t+1
//   phase 1: read A(m1)        -> MFMA (m1,n0)      DMA: B-half0 of tile t+1
//   phase 2: read B(n1)        -> MFMA (m1,n1)      DMA: B-half1 of tile t+1
//   phase 3: (no reads)        -> MFMA (m0,n1)      DMA: A-half0 of tile t+2 ; s_waitcnt vmcnt(2)
// Hazards (instants = barrier releases; group 1 runs one instant behind group 0):
//   WAR  A[set] is last read in phase 1, B[set] in phase 2; their re-staging starts two phases later (phase 3 / phase 1
//        of the next tile), i.e. >= 2 barrier instants after the slower group's reads have been waited for.
//   RAW  every wave waits (vmcnt) for its own DMA pieces of tile t+1 BEFORE the first barrier of phase 3; the first
//        reads of tile t+1 (phase 0) come after at least one more barrier for both groups.
// ABL (diagnostic builds only): ablation mask for tools/gemm_ablate.py -- 1: no MFMA, 2: no LDS fragment reads, 4: no DMA,
// 8: no barriers.  Results are garbage then; only the K-loop time is of interest.
// OP: operand type of the main K loop.  0 = bf16 (v_mfma_f32_16x16x32_bf16).  1 / 2 = OCP fp8 e4m3 (BASELINE configs[4]): the
// SAME LDS image, DMA schedule and ds_read_b128 fragment reads -- a 128-byte LDS row now holds 128 k-elements instead of 64, so
// LDS-DMA and LDS-read bytes per FLOP halve (tools/gemm_ablate.py: the bf16 loop is limited as much by the DMA stream as by the
// matrix pipe).  A lane's 16-byte fragment is 16 consecutive k of one row; any assignment of those bytes to MFMA k-slots is
// valid as long as A and B use the same one, so
//   OP 1: the two 8-byte halves feed two v_mfma_f32_16x16x32_fp8_fp8 (bf16 rate: gains the bytes only);
//   OP 2: the fragments of both half-tiles (32 bytes) feed ONE v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales
//         (E8M0 127): 2x the bf16 MFMA rate.
// An optional LAST K-tile is bf16 (e.a_aug / e.b_aug, 64 columns): the LoRA branch t . B^T stays in bf16 and is added in the
// accumulator's units (B_aug rows pre-divided by alpha[n]), then alpha[n] dequantises the sum in the epilogue.
template <int EPI, bool HAS_BIAS, bool DIAG = false, int ABL = 0, int SCHED = 0, int OP = 0>
__global__ __launch_bounds__(512) void gemm_nt_pp_kernel(const bf16_t* __restrict__ A, int lda,
                                                          const bf16_t* __restrict__ B, int ldb, void* __restrict__ C,
                                                          int ldc, int M, int N, int K, int tiles_n, EpiArgs e) {
    static_assert(OP == 0 || (SCHED == 0 && ABL == 0), "fp8 operands: production schedule only");
    if constexpr (EPI == BSCLIP_EPI_F32 && !HAS_BIAS && OP == 0) {   // split-K slabs (gridDim.y == 1: no-op)
        A += (size_t)blockIdx.y * K;
        B += (size_t)blockIdx.y * K;
        C = static_cast<float*>(C) + (size_t)blockIdx.y * e.split_stride;
    }
    if constexpr (epi_is_resid(EPI)) BSCLIP_DROP_RESOLVE(e.drop);
    constexpr int ESZ = OP == 0 ? 2 : 1;   // bytes per element of the main operands
    constexpr int SET = 65536, HALF = 16384, B_OFF = 32768;
    // main loop 128 KiB; epilogue slabs 2x64x1040 (f32) or 4x64x528 (bf16 x2) = 132 KiB, + 16 KiB GELU table
    __shared__ __attribute__((aligned(16))) char smem[4 * 64 * 528 + (EPI == BSCLIP_EPI_GELU_BF16 || EPI == BSCLIP_EPI_GELU_FP8 ? GELU_LUT_BYTES : 0)];

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = wg % tiles_n, tile_m = wg / tiles_n;
    const int m0 = tile_m * 256, n0 = tile_n * 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, wc = wave & 3;

    // ---- LDS-DMA sources: for half h, this wave moves chunks (wave) and (wave+8): rows 128h + 8*chunk + (lane>>3) ----
    // sources are byte pointers; K-tile t of the main operands starts t * 128 bytes into the row (64 bf16 / 128 fp8 elements)
    const char* srcA[2][2];
    const char* srcB[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 128 * h + 8 * (wave + 8 * i) + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            srcA[h][i] = reinterpret_cast<const char*>(A) + (size_t)min(m0 + row, M - 1) * lda * ESZ + c * 16;
            srcB[h][i] = reinterpret_cast<const char*>(B) + (size_t)min(n0 + row, N - 1) * ldb * ESZ + c * 16;
        }
    const int nk_main = K / (BK * 2 / ESZ);
    const bool aug = OP != 0 && e.a_aug != nullptr;   // wave-uniform
    const int dma_off = wave * 1024;  // + i*8192 + half*HALF (+ B_OFF) + set*SET
    // the bf16 K-augmentation tile: its per-lane sources are recomputed when it is staged (once per workgroup) instead of
    // living in 16 more VGPRs through the whole loop
    auto aug_src = [&](const bf16_t* base, int ld, int r0, int rmax, int h, int i) {
        const int row = 128 * h + 8 * (wave + 8 * i) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        return reinterpret_cast<const char*>(base) + (size_t)min(r0 + row, rmax) * ld * 2 + c * 16;
    };
    auto dmaA = [&](int set, int h, int k0) {   // k0 = K-tile index * 128 (byte offset into the row)
        if constexpr (ABL & 4) return;
        char* d = smem + set * SET + h * HALF + dma_off;
        if (OP != 0 && aug && k0 == nk_main * 128) {
            glds16(aug_src(e.a_aug, e.ld_a_aug, m0, M - 1, h, 0), d);
            glds16(aug_src(e.a_aug, e.ld_a_aug, m0, M - 1, h, 1), d + 8192);
        } else {
            glds16(srcA[h][0] + k0, d);
            glds16(srcA[h][1] + k0, d + 8192);
        }
    };
    auto dmaB = [&](int set, int h, int k0) {
        if constexpr (ABL & 4) return;
        char* d = smem + set * SET + B_OFF + h * HALF + dma_off;
        if (OP != 0 && aug && k0 == nk_main * 128) {
            glds16(aug_src(e.b_aug, e.ld_b_aug, n0, N - 1, h, 0), d);
            glds16(aug_src(e.b_aug, e.ld_b_aug, n0, N - 1, h, 1), d + 8192);
        } else {
            glds16(srcB[h][0] + k0, d);
            glds16(srcB[h][1] + k0, d + 8192);
        }
    };

    // ---- fragment read offsets ----
    const int fr = lane & 15, fq = lane >> 4;
    const int sw = fr >> 1;
    const int a_off = (128 * g + fr) * ROW_BYTES + ((fq ^ sw) << 4);
    const int b_off = B_OFF + (64 * wc + fr) * ROW_BYTES + ((fq ^ sw) << 4);

    f32x4 acc[2][2][4][2];  // [m block][n block][row tile][col tile]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 fa[2][2][4];  // [m block][ks][row tile]
    bf16x8 fb[2][2][2];  // [n block][ks][col tile]
    if constexpr (ABL & 2) {  // fragments must still hold something defined
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[a][k][i] = bf16x8{1, 2, 3, 4, 5, 6, 7, 8};
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[a][k][j] = bf16x8{8, 7, 6, 5, 4, 3, 2, 1};
            }
    }
    // OP 2 keeps a tile's two 16-byte fragments in ONE 8-register vector (the operand of the block-scaled MFMA), so no
    // register copies are needed to form it; the peeled bf16 tile takes its halves
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef int i32x8 __attribute__((ext_vector_type(8)));
    i32x8 fa8[OP == 2 ? 2 : 1][OP == 2 ? 4 : 1];
    i32x8 fb8[OP == 2 ? 2 : 1][OP == 2 ? 2 : 1];
    auto readA = [&](const char* base, int mi) {
        if constexpr (ABL & 2) return;
        if constexpr (OP == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const i32x4 lo = *reinterpret_cast<const i32x4*>(base + (a_off + (64 * mi + 16 * i) * ROW_BYTES));
                const i32x4 hi = *reinterpret_cast<const i32x4*>(base + ((a_off ^ 64) + (64 * mi + 16 * i) * ROW_BYTES));
                fa8[mi][i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    fa[mi][ks][i] = *reinterpret_cast<const bf16x8*>(base + ((a_off ^ (ks << 6)) + (64 * mi + 16 * i) * ROW_BYTES));
        }
    };
    auto readB = [&](const char* base, int ni) {
        if constexpr (ABL & 2) return;
        if constexpr (OP == 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const i32x4 lo = *reinterpret_cast<const i32x4*>(base + (b_off + (32 * ni + 16 * j) * ROW_BYTES));
                const i32x4 hi = *reinterpret_cast<const i32x4*>(base + ((b_off ^ 64) + (32 * ni + 16 * j) * ROW_BYTES));
                fb8[ni][j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    fb[ni][ks][j] = *reinterpret_cast<const bf16x8*>(base + ((b_off ^ (ks << 6)) + (32 * ni + 16 * j) * ROW_BYTES));
        }
    };
    // BF16TILE (compile-time tag): bf16 MFMAs -- always for OP 0, for the peeled K-augmentation tile of the fp8 builds
    auto mma = [&](int mi, int ni, auto bf16_tag = std::true_type{}) {
        if constexpr (ABL & 1) return;
        constexpr bool BF16TILE = OP == 0 || decltype(bf16_tag)::value;
        __builtin_amdgcn_s_setprio(1);
        if constexpr (BF16TILE && OP == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const i32x8 a = fa8[mi][i], b = fb8[ni][j];
                    acc[mi][ni][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, __builtin_shufflevector(b, b, 0, 1, 2, 3)),
                        __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, a, 0, 1, 2, 3)), acc[mi][ni][i][j], 0, 0, 0);
                    acc[mi][ni][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, __builtin_shufflevector(b, b, 4, 5, 6, 7)),
                        __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, a, 4, 5, 6, 7)), acc[mi][ni][i][j], 0, 0, 0);
                }
        } else if constexpr (BF16TILE) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[mi][ni][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ni][ks][j], fa[mi][ks][i], acc[mi][ni][i][j], 0, 0, 0);
        } else if constexpr (OP == 1) {
            typedef long i64x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const i64x2 a2 = __builtin_bit_cast(i64x2, fa[mi][ks][i]), b2 = __builtin_bit_cast(i64x2, fb[ni][ks][j]);
                        acc[mi][ni][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b2[0], a2[0], acc[mi][ni][i][j], 0, 0, 0);
                        acc[mi][ni][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b2[1], a2[1], acc[mi][ni][i][j], 0, 0, 0);
                    }
        } else if constexpr (OP == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    // cbsz = blgp = 0: both operands OCP e4m3; block scales 2^(127-127) = 1 for every 32-element block
                    acc[mi][ni][i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb8[ni][j], fa8[mi][i], acc[mi][ni][i][j], 0, 0,
                                                                                         0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
        __builtin_amdgcn_s_setprio(0);
    };
#define PP_BARRIER()                                           \
    do {                                                       \
        __builtin_amdgcn_sched_barrier(0);                     \
        if constexpr (!(ABL & 8)) __builtin_amdgcn_s_barrier(); \
        __builtin_amdgcn_sched_barrier(0);                     \
    } while (0)

    auto stamp = [&](int i) {
        if constexpr (DIAG) {
            if (lane == 0 && wc == 0) e.diag[(size_t)blockIdx.x * 16 + g * 8 + i] = wall_clock64();
        }
    };
    stamp(0);
    const int nk = nk_main + (aug ? 1 : 0);
#ifdef BSCLIP_DIAG
    if constexpr (SCHED == 1) {
        // Deep-prefetch variant, kept for comparison (bsclip_gemm_set_tile(6)); NOT the default.  tools/gemm_ablate.py shows the
        // K loop bound as much by the LDS-DMA stream as by the matrix pipe (DMA + barriers alone 1.1 us per K-tile, MFMA +
        // barriers alone 1.0, both 1.45).  If the DMA side were latency x bytes in flight, refilling a set with tile t+2 as
        // soon as tile t has been read out of it -- A after phase 1, B after phase 2, ~100 KB per CU in flight instead of ~40 --
        // would fix it.  It does not: DMA + barriers alone stay at 1.05 us per K-tile (a throughput limit of the L2 -> LDS path,
        // ~61 GB/s per CU) and the full loop gets slower (1.56 us), so the one-tile-ahead schedule below remains in use.
        // WAR: group 1 runs one barrier instant behind group 0.  A[set] is last read in phase 1; group 1's reads are waited for
        //      before its mma(1,0), i.e. before global instant b4 (the barrier group 0 passes after its phase-2 reads), and A
        //      DMAs are issued after that barrier (group 1: after its own, one instant later).  B[set] is last read in phase 2,
        //      complete by b6; B DMAs are issued after the phase-3 barrier.
        // RAW: before the phase-3 barrier every wave waits until only its 4 newest DMA instructions (A of tile t+2) are
        //      outstanding: VMEM returns in order, so all its pieces of tile t+1 have landed; tile t+1 is first read after at
        //      least one more barrier by either group.
        dmaA(0, 0, 0);
        dmaA(0, 1, 0);
        dmaB(0, 0, 0);
        dmaB(0, 1, 0);
        if (nk > 1) {
            dmaA(1, 0, 2 * BK);
            dmaA(1, 1, 2 * BK);
            dmaB(1, 0, 2 * BK);
            dmaB(1, 1, 2 * BK);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PP_BARRIER();
        if (g == 1) PP_BARRIER();  // group 1 runs one barrier behind group 0 from here on
        stamp(1);
        for (int t = 0; t < nk; ++t) {
            const int set = t & 1;
            const char* base = smem + set * SET;
            const bool has2 = t + 2 < nk;
            const int k2 = (t + 2) * 2 * BK;
            // ---- phase 0 ----
            readA(base, 0);
            readB(base, 0);
            PP_BARRIER();
            mma(0, 0, std::true_type{});
            PP_BARRIER();
            // ---- phase 1 ----
            readA(base, 1);
            PP_BARRIER();
            mma(1, 0, std::true_type{});
            PP_BARRIER();
            // ---- phase 2 ----
            readB(base, 1);
            PP_BARRIER();
            if (has2) {
                dmaA(set, 0, k2);
                dmaA(set, 1, k2);
            }
            mma(1, 1, std::true_type{});
            PP_BARRIER();
            // ---- phase 3 ----
            if (has2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // tile t+1 complete; A of tile t+2 may fly
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PP_BARRIER();
            if (has2) {
                dmaB(set, 0, k2);
                dmaB(set, 1, k2);
            }
            mma(0, 1, std::true_type{});
            PP_BARRIER();
        }
    } else if constexpr (SCHED == 2) {
        // Four barriers per K-tile ("half-barrier ping-pong", experimental: bsclip_gemm_set_tile(7)).  One s_barrier per phase,
        // call it b(t,p); group 0 runs   reads(p) | b(t,p) | mma(p)   and group 1 runs   b(t,p) | reads(p), mma(p),   so between
        // two barriers group 0 does [mma(p), reads(p+1)] while group 1 does [reads(p), mma(p)]: matrix beside LDS in each half.
        // Reads of tile t: A blocks in phases 0 and 1, B blocks in phases 0 and 2.  Group 0 issues reads(p) before b(t,p) and
        // waits for them after it; group 1 issues and waits between b(t,p) and b(t,p+1).  Hence every read of A[set(t)] is
        // complete before b(t,2) and every read of B[set(t)] before b(t,3).
        // WAR: A[set(t)] may be refilled (tile t+2) after b(t,2), B[set(t)] after b(t,3).
        // RAW: group 0 reads tile t+1 between b(t,3) and b(t+1,0), so every wave waits for its own pieces of tile t+1 BEFORE
        //      it calls b(t,3).  Group 0's code before b(t,3) is its phase-3 slot (as in the classic schedule); group 1's is the
        //      end of its phase 2, so group 1 issues its pieces one phase earlier than group 0:
        //        group 0, tile t:  ph0 A-half1(t+1)  ph1 B-half0(t+1)  ph2 B-half1(t+1)  ph3 A-half0(t+2), wait vmcnt(2)
        //        group 1, tile t:  ph0 B-half0(t+1)  ph1 B-half1(t+1)  ph2 wait vmcnt(0)  ph3 A-half0(t+2), A-half1(t+2)
        //      (group 1's phase-3 code runs after b(t,3) > b(t,2): A[set(t)] is free; its B pieces go out after b(t,0), b(t,1),
        //      both later than b(t-1,3)).
        dmaA(0, 0, 0);
        dmaA(0, 1, 0);
        dmaB(0, 0, 0);
        dmaB(0, 1, 0);
        if (nk > 1) {
            dmaA(1, 0, 2 * BK);
            if (g == 1) {
                dmaA(1, 1, 2 * BK);
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PP_BARRIER();  // tile 0 published
        stamp(1);
        // one code body for both groups (two copies of the loop spilled: 308 B of scratch per lane); only the barrier position
        // and the DMA piece of each phase depend on the group
#define BAR_G0() do { if (g == 0) PP_BARRIER(); } while (0)
#define BAR_G1() do { if (g == 1) PP_BARRIER(); } while (0)
        for (int t = 0; t < nk; ++t) {
            const int set = t & 1;
            const char* base = smem + set * SET;
            const bool has1 = t + 1 < nk, has2 = t + 2 < nk;
            const int k1 = (t + 1) * 2 * BK, k2 = (t + 2) * 2 * BK;
            // ---- phase 0 ----
            BAR_G1();
            if (has1) {
                if (g == 0) dmaA(set ^ 1, 1, k1);
                else dmaB(set ^ 1, 0, k1);
            }
            readA(base, 0);
            readB(base, 0);
            BAR_G0();
            mma(0, 0, std::true_type{});
            // ---- phase 1 ----
            BAR_G1();
            if (has1) {
                if (g == 0) dmaB(set ^ 1, 0, k1);
                else dmaB(set ^ 1, 1, k1);
            }
            readA(base, 1);
            BAR_G0();
            mma(1, 0, std::true_type{});
            // ---- phase 2 ----
            BAR_G1();
            if (has1 && g == 0) dmaB(set ^ 1, 1, k1);
            readB(base, 1);
            BAR_G0();
            mma(1, 1, std::true_type{});
            // ---- phase 3 ----
            if (g == 0) {
                if (has2) {
                    dmaA(set, 0, k2);
                    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // tile t+1 landed; A-half0(t+2) may fly
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of tile t+1 have landed
            }
            PP_BARRIER();  // b(t,3): the one barrier both groups call at the same point of their code
            if (g == 1 && has2) {
                dmaA(set, 0, k2);
                dmaA(set, 1, k2);
            }
            mma(0, 1, std::true_type{});
        }
#undef BAR_G0
#undef BAR_G1
    } else
#endif
    {
        // ---- prologue: tile 0 complete, plus the first piece of tile 1 (the "phase 3 of tile -1" slot) ----
        dmaA(0, 0, 0);
        dmaA(0, 1, 0);
        dmaB(0, 0, 0);
        dmaB(0, 1, 0);
        if (nk > 1) {
            dmaA(1, 0, 2 * BK);
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PP_BARRIER();
        if (g == 1) PP_BARRIER();  // group 1 runs one barrier behind group 0 from here on
        stamp(1);

        auto k_tile = [&](int t, auto tag) {
            const int set = t & 1;
            const char* base = smem + set * SET;
            const bool has1 = t + 1 < nk, has2 = t + 2 < nk;
            const int k1 = (t + 1) * 2 * BK, k2 = (t + 2) * 2 * BK;   // byte offsets of K-tiles t+1, t+2
            // ---- phase 0 ----
            if (has1) dmaA(set ^ 1, 1, k1);
            readA(base, 0);
            readB(base, 0);
            PP_BARRIER();
            mma(0, 0, tag);
            PP_BARRIER();
            // ---- phase 1 ----
            if (has1) dmaB(set ^ 1, 0, k1);
            readA(base, 1);
            PP_BARRIER();
            mma(1, 0, tag);
            PP_BARRIER();
            // ---- phase 2 ----
            if (has1) dmaB(set ^ 1, 1, k1);
            readB(base, 1);
            PP_BARRIER();
            mma(1, 1, tag);
            PP_BARRIER();
            // ---- phase 3 ----
            if (has2) {
                dmaA(set, 0, k2);
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // everything of tile t+1 has landed; A-half0(t+2) may fly
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            PP_BARRIER();
            mma(0, 1, tag);
            PP_BARRIER();
        };
        if constexpr (OP == 0) {
            for (int t = 0; t < nk; ++t) k_tile(t, std::true_type{});
        } else {  // fp8 tiles, then the peeled bf16 K-augmentation tile (same schedule, bf16 MFMAs)
            for (int t = 0; t < nk_main; ++t) k_tile(t, std::false_type{});
            if (aug) k_tile(nk_main, std::true_type{});
        }
    }
    if constexpr (SCHED != 2) {
        if (g == 0) PP_BARRIER();  // balance group 1's extra barrier
    }
#undef PP_BARRIER
    stamp(2);

    // ---- epilogue ----
    f32x4 bias[2][2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bias[ni][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (HAS_BIAS)
                bias[ni][j] = *reinterpret_cast<const f32x4*>(e.bias + n0 + 64 * wc + 32 * ni + 16 * j + fq * 4);
        }
    if constexpr (OP != 0) {  // dequantise: alpha[n] = activation scale x weight-row scale
        f32x4 al[2][2];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                al[ni][j] = *reinterpret_cast<const f32x4*>(e.alpha + n0 + 64 * wc + 32 * ni + 16 * j + fq * 4);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[mi][ni][i][j] *= al[ni][j];
    }
    if constexpr (HAS_BIAS) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[mi][ni][i][j] += bias[ni][j];
    }
    // ---- fused InfoNCE, pass 1: the logits tile never leaves the accumulators --------------------------------------
    // Per output row the tile's 256 columns live in 4 waves (wc) x 4 lanes (fq) x 16 registers: in-lane reduction, two xor
    // shuffles (16, 32) across the lanes that share a row, LDS across the four waves.  Written per (row, column tile):
    // (max, sum exp(x - max), sum over same-label columns of x); infonce_combine_kernel merges the column tiles.
    if constexpr (EPI == EPI_LSE_PART) {
        __syncthreads();  // every wave is past its last fragment read: the staging LDS is free
        float* red = reinterpret_cast<float*>(smem);  // [256 rows][4 waves][4]
        long lcol[2][2][4];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int n = n0 + 64 * wc + 32 * ni + 16 * j + 4 * fq + c;
                    lcol[ni][j][c] = n < e.n_valid ? (long)e.labels[n] : (long)0x7fffffffffffffffLL;
                }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 128 * g + 64 * mi + 16 * i + fr;
                const int grow = e.row_base + min(m0 + r, M - 1);
                const long li = (long)e.labels[min(grow, e.n_valid - 1)];
                float x[16];
                float mx = -INFINITY;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int n = n0 + 64 * wc + 32 * ni + 16 * j + 4 * fq + c;
                            const float v = n < e.n_valid ? acc[mi][ni][i][j][c] * e.logit_scale : -INFINITY;
                            x[(ni * 2 + j) * 4 + c] = v;
                            mx = fmaxf(mx, v);
                        }
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                float se = 0.f, dot = 0.f;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float v = x[(ni * 2 + j) * 4 + c];
                            se += mx == -INFINITY ? 0.f : __expf(v - mx);
                            dot += lcol[ni][j][c] == li ? v : 0.f;
                        }
                se += __shfl_xor(se, 16, 64);
                se += __shfl_xor(se, 32, 64);
                dot += __shfl_xor(dot, 16, 64);
                dot += __shfl_xor(dot, 32, 64);
                if (fq == 0) *reinterpret_cast<f32x4*>(red + (r * 4 + wc) * 4) = f32x4{mx, se, dot, 0.f};
            }
        __syncthreads();
        if (tid < 256 && m0 + tid < M) {
            float mm = -INFINITY, ss = 0.f, dd = 0.f;
            f32x4 p[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                p[w] = *reinterpret_cast<const f32x4*>(red + (tid * 4 + w) * 4);
                mm = fmaxf(mm, p[w][0]);
            }
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                ss += p[w][0] == -INFINITY ? 0.f : p[w][1] * __expf(p[w][0] - mm);
                dd += p[w][2];
            }
            *reinterpret_cast<f32x4*>(e.part + ((size_t)(m0 + tid) * tiles_n + tile_n) * 4) = f32x4{mm, ss, dd, 0.f};
        }
        return;
    }
    // ---- fused InfoNCE, pass 2: dLoss/dG of the tile, straight from the accumulators, in split-bf16 form --------------
    // w_ij = coef (cnt_i exp(x - lse_ab[i]) + cnt_j exp(x - lse_ba[j]) - 2 T_ij) for j < n_valid, else 0; written as the
    // operand [hi | hi | lo] of the gradient GEMM (C bf16 [M, ldc = 3 * Np]) -- the logits themselves are never stored.
    if constexpr (EPI == EPI_LOSS_W) {
        constexpr int SBW = 528;
        __syncthreads();
        char* slab_hi = smem + g * (64 * SBW);
        char* slab_lo = smem + 2 * 64 * SBW + g * (64 * SBW);
        long lcol[2][2][4];
        float ccol[2][2][4], lcolse[2][2][4];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int n = n0 + 64 * wc + 32 * ni + 16 * j + 4 * fq + c;
                    const bool ok = n < e.n_valid;
                    lcol[ni][j][c] = ok ? (long)e.labels[n] : (long)0x7fffffffffffffffLL;
                    ccol[ni][j][c] = ok ? e.cnt[n] : 0.f;
                    lcolse[ni][j][c] = ok ? e.lse_col[n] : 0.f;
                }
        const int np = ldc / 3;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 128 * g + 64 * mi + 16 * i + fr;
                const int grow = min(e.row_base + min(m0 + r, M - 1), e.n_valid - 1);
                const long li = (long)e.labels[grow];
                const float ci = e.cnt[grow], lr = e.lse_row[grow];
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float wv[4], hi[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int n = n0 + 64 * wc + 32 * ni + 16 * j + 4 * fq + c;
                            const float xv = acc[mi][ni][i][j][c] * e.logit_scale;
                            const float t = lcol[ni][j][c] == li ? 2.0f : 0.0f;
                            wv[c] = n < e.n_valid ? e.coef * (ci * __expf(xv - lr) + ccol[ni][j][c] * __expf(xv - lcolse[ni][j][c]) - t) : 0.f;
                            hi[c] = bf2f(f2bf(wv[c]));
                        }
                        const int off = (16 * i + fr) * SBW + (64 * wc + 32 * ni + 16 * j + 4 * fq) * 2;
                        *reinterpret_cast<uint2*>(slab_hi + off) = uint2{pack_bf2(wv[0], wv[1]), pack_bf2(wv[2], wv[3])};
                        *reinterpret_cast<uint2*>(slab_lo + off) =
                            uint2{pack_bf2(wv[0] - hi[0], wv[1] - hi[1]), pack_bf2(wv[2] - hi[2], wv[3] - hi[3])};
                    }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int r = it * 8 + (wave & 3) * 2 + (lane >> 5);
                const int m = m0 + 128 * g + 64 * mi + r;
                const uint4 vh = *reinterpret_cast<const uint4*>(slab_hi + r * SBW + (lane & 31) * 16);
                const uint4 vl = *reinterpret_cast<const uint4*>(slab_lo + r * SBW + (lane & 31) * 16);
                if (m < M) {
                    bf16_t* dst = static_cast<bf16_t*>(C) + (size_t)m * ldc + n0 + (lane & 31) * 8;
                    *reinterpret_cast<uint4*>(dst) = vh;
                    *reinterpret_cast<uint4*>(dst + np) = vh;
                    *reinterpret_cast<uint4*>(dst + 2 * np) = vl;
                }
            }
            __syncthreads();
        }
        return;
    }
    // ---- LDS-staged, row-coalesced stores --------------------------------------------------------------------
    // In the accumulator layout a lane owns 4 consecutive columns of 16 different rows, so direct stores are 8-B
    // (bf16) pieces of 32-B row segments: 64-128 store instructions per lane, store-ISSUE bound (the epilogue cost
    // more than the 12-tile K loop of the K=768 GEMMs).  The main-loop LDS is free now: each group stages a 64-row
    // slab of its output, then its 256 threads walk the slab row-wise with 16 B per lane, so every global access
    // (C, residual, saved pre-activation) is a full 512-B / 1-KiB row segment.
    constexpr int SB = 528;   // bf16 slab row stride (256*2 + 16)
    constexpr int SF = 1040;  // f32 slab row stride (256*4 + 16)
    char* slab = smem + g * (64 * SF);
    const int wq = wave & 3;
    __syncthreads();  // every wave is past its last fragment read
    const char* lut = smem + 4 * 64 * SB;
    constexpr bool GELU = EPI == BSCLIP_EPI_GELU_BF16 || EPI == BSCLIP_EPI_GELU_FP8;
    if constexpr (GELU) {  // 16 KiB table, L2-resident, behind the slabs
        for (int i = tid; i <= GELU_LUT_N; i += 512)
            *reinterpret_cast<float2*>(smem + 4 * 64 * SB + i * 8) = g_gelu_lut[i];
        BSCLIP_LDS_BARRIER();
    }
    constexpr int S8 = 272;   // 8-bit slab row stride (256 + 16)
    char* slab2 = smem + 2 * 64 * SB + g * (64 * S8);  // second slab (GELU: gelu' side band, 8-bit codes)
    auto stage_bf16 = [&](int mi) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f32x4 v = acc[mi][ni][i][j];
                    const int off = (16 * i + fr) * SB + (64 * wc + 32 * ni + 16 * j + 4 * fq) * 2;
                    uint2 o;
                    if constexpr (GELU) {
                        f32x2 gl0, dg0, gl1, dg1;
                        gelu_lut2(lut, f32x2{v[0], v[1]}, gl0, dg0);
                        gelu_lut2(lut, f32x2{v[2], v[3]}, gl1, dg1);
                        *reinterpret_cast<unsigned*>(slab2 + (16 * i + fr) * S8 + (64 * wc + 32 * ni + 16 * j + 4 * fq)) =
                            dg8_pack4(dg0, dg1);
                        if constexpr (EPI == BSCLIP_EPI_GELU_FP8) {  // the next GEMM's fp8 operand: e4m3, scale 1, saturated
                            *reinterpret_cast<unsigned*>((g ? smem + 64 * SB : smem) + (16 * i + fr) * S8 +
                                                         (64 * wc + 32 * ni + 16 * j + 4 * fq)) =
                                pack_fp8x4(gl0[0], gl0[1], gl1[0], gl1[1]);
                            continue;
                        }
                        o.x = pack_bf2(gl0[0], gl0[1]);
                        o.y = pack_bf2(gl1[0], gl1[1]);
                    } else {
                        o.x = pack_bf2(v[0], v[1]);
                        o.y = pack_bf2(v[2], v[3]);
                    }
                    *reinterpret_cast<uint2*>((g ? smem + 64 * SB : smem) + off) = o;
                }
    };
    auto rows_bf16 = [&](int mi, const char* src, bf16_t* dst, int ld) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int r = it * 8 + wq * 2 + (lane >> 5);
            const int m = m0 + 128 * g + 64 * mi + r;
            const uint4 v = *reinterpret_cast<const uint4*>(src + r * SB + (lane & 31) * 16);
            if (m < M) nt_store(dst + (size_t)m * ld + n0 + (lane & 31) * 8, v);
        }
    };
    // 64 x 256 B slab of 8-bit codes: 16 lanes x 16 B per row, 16 rows per pass of the group's 256 threads
    auto rows_u8 = [&](int mi, const char* src, unsigned char* dst, int ld) {
        const int t = tid & 255;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = it * 16 + (t >> 4);
            const int m = m0 + 128 * g + 64 * mi + r;
            const uint4 v = *reinterpret_cast<const uint4*>(src + r * S8 + (t & 15) * 16);
            if (m < M) nt_store(dst + (size_t)m * ld + n0 + (t & 15) * 16, v);
        }
    };
    auto stage_f32 = [&](int mi) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    *reinterpret_cast<f32x4*>(slab + (16 * i + fr) * SF + (64 * wc + 32 * ni + 16 * j + 4 * fq) * 4) =
                        acc[mi][ni][i][j];
    };
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        if constexpr (EPI == BSCLIP_EPI_BF16 || GELU) {
            stage_bf16(mi);
            BSCLIP_LDS_BARRIER();
            stamp(4 + 2 * mi);
            if constexpr (EPI == BSCLIP_EPI_GELU_FP8) rows_u8(mi, smem + g * (64 * SB), static_cast<unsigned char*>(C), ldc);
            else rows_bf16(mi, smem + g * (64 * SB), static_cast<bf16_t*>(C), ldc);
            if constexpr (GELU) {
                if (e.aux) rows_u8(mi, slab2, e.aux, e.ld_aux);  // gelu'(pre-activation) for the backward pass
            }
            BSCLIP_LDS_BARRIER();
            stamp(5 + 2 * mi);
        }
    }
    if constexpr (!(EPI == BSCLIP_EPI_BF16 || GELU)) {
        // f32-staged epilogues read a second operand (residual stream / saved gelu') row-wise.  Those loads do not
        // depend on the accumulators, so they are issued one slab ahead -- before the staging barrier -- and have the
        // whole LDS round trip to land (issued just-in-time they were 16 serial HBM round trips per wave: 26 us/tile).
        f32x4 pre[2][16];
        auto prefetch = [&](int mi, f32x4 (&R)[16]) {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int m = min(m0 + 128 * g + 64 * mi + it * 4 + wq, M - 1);
                const int n = n0 + lane * 4;
                if constexpr (EPI == BSCLIP_EPI_RESID_F32) {
                    R[it] = *reinterpret_cast<const f32x4*>(e.resid + (size_t)m * e.ld_resid + n);
                } else if constexpr (EPI == BSCLIP_EPI_RESID_BF16) {
                    R[it] = bf4_to_f32(*reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(e.resid) + (size_t)m * e.ld_resid + n));
                } else if constexpr (EPI == BSCLIP_EPI_DGELU_BF16) {
                    R[it] = dg8_unpack4(*reinterpret_cast<const unsigned*>(e.aux + (size_t)m * e.ld_aux + n));
                } else if constexpr (epi_is_patch(EPI)) {
                    R[it] = *reinterpret_cast<const f32x4*>(e.resid + (size_t)(1 + m % 196) * e.ld_resid + n);
                } else {
                    R[it] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        };
        auto consume = [&](int mi, const f32x4 (&R)[16]) {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int r = it * 4 + wq;  // one 1-KiB row per wave instruction
                const int m = m0 + 128 * g + 64 * mi + r;
                const int n = n0 + lane * 4;
                f32x4 v = *reinterpret_cast<const f32x4*>(slab + r * SF + lane * 16);
                if (m < M) {
                    if constexpr (EPI == BSCLIP_EPI_F32) {
                        nt_store(static_cast<float*>(C) + (size_t)m * ldc + n, v);
                    } else if constexpr (EPI == BSCLIP_EPI_RESID_F32) {
                        if (e.drop.thr16) v = drop4(e.drop, (unsigned)m * (unsigned)e.n_total + (unsigned)n, v);
                        v += R[it];
                        nt_store(static_cast<float*>(C) + (size_t)m * ldc + n, v);
                    } else if constexpr (EPI == BSCLIP_EPI_DGELU_BF16) {
                        v *= R[it];
                        uint2 o;
                        o.x = pack_bf2(v[0], v[1]);
                        o.y = pack_bf2(v[2], v[3]);
                        nt_store(static_cast<bf16_t*>(C) + (size_t)m * ldc + n, o);
                    } else if constexpr (EPI == BSCLIP_EPI_PATCH_F32) {
                        const int b = m / 196, p = m - b * 196;
                        v += R[it];
                        nt_store(static_cast<float*>(C) + (size_t)(b * 197 + 1 + p) * ldc + n, v);
                    } else if constexpr (EPI == BSCLIP_EPI_RESID_BF16) {
                        if (e.drop.thr16) v = drop4(e.drop, (unsigned)m * (unsigned)e.n_total + (unsigned)n, v);
                        v += R[it];
                        nt_store(static_cast<bf16_t*>(C) + (size_t)m * ldc + n, f32_to_bf4(v));
                    } else if constexpr (EPI == BSCLIP_EPI_PATCH_BF16) {
                        const int b = m / 196, p = m - b * 196;
                        v += R[it];
                        nt_store(static_cast<bf16_t*>(C) + (size_t)(b * 197 + 1 + p) * ldc + n, f32_to_bf4(v));
                    }
                }
            }
        };
        prefetch(0, pre[0]);
        stage_f32(0);
        BSCLIP_LDS_BARRIER();
        stamp(4);
        prefetch(1, pre[1]);
        consume(0, pre[0]);
        BSCLIP_LDS_BARRIER();
        stamp(5);
        stage_f32(1);
        BSCLIP_LDS_BARRIER();
        stamp(6);
        consume(1, pre[1]);
        stamp(7);
    }
    stamp(3);
}

#include "gemm_duo.h"
#include "gemm_pers.h"

int g_tile_override = 0;
// selection of the persistent kernel, tuned IN THE STEP (profiles/r03_m_pers_selection.log): per-shape microbenchmarks favour the
// duo kernel on the ViT's 768x768 GEMMs and tie on K >= 2304, but with two tower streams sharing the chip the step is fastest
// when every 256x256-tile GEMM with more tiles than CUs is persistent (39.24 -> 38.59 ms on one box)
const int g_pers_max_k = getenv("BSCLIP_PERS_MAX_K") ? atoi(getenv("BSCLIP_PERS_MAX_K")) : 4096;
const int g_pers_min_tiles = getenv("BSCLIP_PERS_MIN_TILES") ? atoi(getenv("BSCLIP_PERS_MIN_TILES")) : 256;
const bool g_duo_off = !(getenv("BSCLIP_GEMM_DUO") && atoi(getenv("BSCLIP_GEMM_DUO")) == 1);   // bsclip_gemm_set_tile(5) still selects it
const bool g_pers_off = getenv("BSCLIP_GEMM_PERSISTENT") && atoi(getenv("BSCLIP_GEMM_PERSISTENT")) == 0;   // A/B switch
[[maybe_unused]] int g_diag_ablate = 0;  // tools/gemm_ablate.py: which parts of the K loop the diagnostic EPI_BF16 build leaves out

template <int BM, int BN, int WM, int WN, int EPI, bool HB>
void launch_cfg(const bf16_t* A, int lda, const bf16_t* B, int ldb, void* C, int ldc, int M, int N, int K,
                const EpiArgs& e, hipStream_t s) {
    const int tiles_m = ceil_div(M, BM), tiles_n = N / BN;
    hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WM, WN, EPI, HB>), dim3(tiles_m * tiles_n), dim3(WM * WN * 64), 0, s, A,
                       lda, B, ldb, C, ldc, M, N, K, tiles_n, e);
}

bool g_lut_ready = false;

template <int EPI, bool HB, int SCHED = 0>
void launch_pp(const bf16_t* A, int lda, const bf16_t* B, int ldb, void* C, int ldc, int M, int N, int K,
               const EpiArgs& e, hipStream_t s) {
    const int tiles_m = ceil_div(M, 256), tiles_n = N / 256;
    hipLaunchKernelGGL((gemm_nt_pp_kernel<EPI, HB, false, 0, SCHED>), dim3(tiles_m * tiles_n), dim3(512), 0, s, A, lda, B,
                       ldb, C, ldc, M, N, K, tiles_n, e);
}

template <int EPI, bool HB>
void launch_epi(const bf16_t* A, int lda, const bf16_t* B, int ldb, void* C, int ldc, int M, int N, int K,
                const EpiArgs& e, hipStream_t s) {
    if (EPI == BSCLIP_EPI_GELU_BF16 && !g_lut_ready) {  // once per process, stream-ordered ahead of the first consumer
        hipLaunchKernelGGL(gelu_lut_init_kernel, dim3(ceil_div(GELU_LUT_N + 1, 256)), dim3(256), 0, s);
        g_lut_ready = true;
    }
    int tile = g_tile_override;
    if (tile == 0) {
        // 256x256 (8 waves, 1 block/CU) halves L2->LDS traffic per FLOP; fall back when N is not a multiple of
        // 256 or the grid would not fill the 256 CUs.
        // measured on MI355X (profiles/r01_b_gemm_tiles.log): the ping-pong 256x256 kernel wins on every encoder shape
        // once the grid covers the chip; 128x128 (2 workgroups/CU) is the better small-grid choice.
        const long t256 = (long)ceil_div(M, 256) * (N / 256);
        if (N % 256 == 0 && t256 >= 192) tile = 4;
        else tile = 1;
        // the two-workgroups-per-CU kernel (gemm_duo.h) ties the ping-pong kernel on most shapes and wins, per shape, where a tile
        // is short (12 K-tiles) and the grid is a few rounds deep: the ViT's N = K = 768 GEMMs (60 vs 67 us, 76 vs 79 us at
        // M = 50 432; profiles/r03_f_gemm_tiles.log).  Off by default since the end of round 3 (BSCLIP_GEMM_DUO=1): in the step,
        // beside the other tower's persistent kernels, it loses (profiles/r03_m_pers_selection.log).
        if (tile == 4 && N == 768 && K == 768 && t256 >= 500 && EPI != BSCLIP_EPI_GELU_BF16 && !g_duo_off) tile = 5;
        // the persistent form (gemm_pers.h) hides a tile's 2.5-3 us prologue under the previous tile.  Per shape it pays where
        // tiles are short (K <= 832) and ties at 36-48 K-tiles (profiles/r03_i_gemm_pers.log); in the step it pays everywhere a
        // workgroup has more than one tile to walk (g_pers_max_k / g_pers_min_tiles above)
        if (tile == 4 && pers_supported(EPI) && K <= g_pers_max_k && K >= 2 * BK && t256 >= g_pers_min_tiles && !g_pers_off) tile = 8;
    }
    if ((tile == 3 || tile == 4 || tile == 6 || tile == 7 || tile == 8) && N % 256 != 0) tile = 2;
    switch (tile) {
        case 4: launch_pp<EPI, HB>(A, lda, B, ldb, C, ldc, M, N, K, e, s); break;
        case 5: launch_duo<EPI, HB>(A, lda, B, ldb, C, ldc, M, N, K, e, s); break;
        case 8:
            if constexpr (pers_supported(EPI)) {
                if (K >= 2 * BK && (size_t)M * lda * 2 < (1ull << 32) && (size_t)N * ldb * 2 < (1ull << 32)) {
                    launch_pers<EPI, HB>(A, lda, B, ldb, C, ldc, M, N, K, e, s);
                    break;
                }
            }
            launch_pp<EPI, HB>(A, lda, B, ldb, C, ldc, M, N, K, e, s);
            break;
#ifdef BSCLIP_DIAG
        case 6: launch_pp<EPI, HB, 1>(A, lda, B, ldb, C, ldc, M, N, K, e, s); break;  // two-tiles-ahead DMA (comparison)
        case 7: launch_pp<EPI, HB, 2>(A, lda, B, ldb, C, ldc, M, N, K, e, s); break;  // four barriers per K-tile (experimental)
#endif
        case 3: launch_cfg<256, 256, 2, 4, EPI, HB>(A, lda, B, ldb, C, ldc, M, N, K, e, s); break;
        case 2: launch_cfg<256, 128, 4, 2, EPI, HB>(A, lda, B, ldb, C, ldc, M, N, K, e, s); break;
        default: launch_cfg<128, 128, 2, 2, EPI, HB>(A, lda, B, ldb, C, ldc, M, N, K, e, s); break;
    }
}

template <int EPI, int OP>
void launch_f8(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const EpiArgs& e,
               hipStream_t s) {
    if (EPI == BSCLIP_EPI_GELU_FP8 && !g_lut_ready) {
        hipLaunchKernelGGL(gelu_lut_init_kernel, dim3(ceil_div(GELU_LUT_N + 1, 256)), dim3(256), 0, s);
        g_lut_ready = true;
    }
    const int tiles_m = ceil_div(M, 256), tiles_n = N / 256;
    hipLaunchKernelGGL((gemm_nt_pp_kernel<EPI, true, false, 0, 0, OP>), dim3(tiles_m * tiles_n), dim3(512), 0, s,
                       static_cast<const bf16_t*>(A), lda, static_cast<const bf16_t*>(B), ldb, C, ldc, M, N, K, tiles_n, e);
}

template <int EPI>
void launch_bias(const bf16_t* A, int lda, const bf16_t* B, int ldb, void* C, int ldc, int M, int N, int K,
                 const EpiArgs& e, hipStream_t s) {
    if (e.bias) launch_epi<EPI, true>(A, lda, B, ldb, C, ldc, M, N, K, e, s);
    else launch_epi<EPI, false>(A, lda, B, ldb, C, ldc, M, N, K, e, s);
}

}  // namespace

#ifdef BSCLIP_DIAG
// Diagnostic: the ping-pong kernel with four phase stamps per workgroup and wave group (start, prologue done, K loop
// done, end) written to diag[grid*16] (8 per wave group: start, prologue, K loop, end, 4 epilogue sections) (100 MHz ticks).  Used by tools/gemm_phases.py; not on any product path.
extern "C" int bsclip_gemm_diag(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                                int epilogue, const bsclip_epi_args* args, unsigned long long* diag, void* stream) {
    BSCLIP_REQUIRE(A && B && C && diag && args, "bsclip_gemm_diag: null pointer");
    BSCLIP_REQUIRE(K % 64 == 0 && N % 256 == 0, "bsclip_gemm_diag: K %% 64, N %% 256");
    EpiArgs e{};
    e.bias = args->bias;
    e.resid = static_cast<const float*>(args->resid);
    e.ld_resid = args->ld_resid;
    e.aux = static_cast<unsigned char*>(args->aux);
    e.ld_aux = args->ld_aux;
    e.drop = make_drop(0.f, 0);
    e.n_total = N;
    e.diag = diag;
    const int tiles_m = ceil_div(M, 256), tiles_n = N / 256;
    const dim3 grid(tiles_m * tiles_n), block(512);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bf16_t* a = static_cast<const bf16_t*>(A);
    const bf16_t* b = static_cast<const bf16_t*>(B);
    switch (epilogue) {
        case BSCLIP_EPI_BF16:
#define DIAG_ABL(mask)                                                                                                      \
    case mask:                                                                                                             \
        hipLaunchKernelGGL((gemm_nt_pp_kernel<BSCLIP_EPI_BF16, false, true, mask>), grid, block, 0, s, a, lda, b, ldb, C, ldc, \
                           M, N, K, tiles_n, e);                                                                           \
        break;
            switch (g_diag_ablate) {
                DIAG_ABL(0) DIAG_ABL(1) DIAG_ABL(2) DIAG_ABL(4) DIAG_ABL(8) DIAG_ABL(6) DIAG_ABL(3) DIAG_ABL(5) DIAG_ABL(9)
                default: BSCLIP_REQUIRE(false, "bsclip_gemm_diag: ablation mask %d not instantiated", g_diag_ablate);
            }
#undef DIAG_ABL
            break;
        case BSCLIP_EPI_GELU_BF16:
            hipLaunchKernelGGL((gemm_nt_pp_kernel<BSCLIP_EPI_GELU_BF16, true, true>), grid, block, 0, s, a, lda, b, ldb, C, ldc, M, N, K, tiles_n, e);
            break;
        case BSCLIP_EPI_RESID_F32:
            hipLaunchKernelGGL((gemm_nt_pp_kernel<BSCLIP_EPI_RESID_F32, true, true>), grid, block, 0, s, a, lda, b, ldb, C, ldc, M, N, K, tiles_n, e);
            break;
        case BSCLIP_EPI_DGELU_BF16:
            hipLaunchKernelGGL((gemm_nt_pp_kernel<BSCLIP_EPI_DGELU_BF16, false, true>), grid, block, 0, s, a, lda, b, ldb, C, ldc, M, N, K, tiles_n, e);
            break;
        default: BSCLIP_REQUIRE(false, "bsclip_gemm_diag: epilogue %d has no diagnostic build", epilogue);
    }
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

// The duo kernel with per-workgroup stamps: diag[grid * 8] = {start, tile 0 landed, K loop done, end, HW_ID, XCC_ID, -, -}.
// tools/gemm_duo_phases.py reads co-residency (two workgroups with overlapping lifetimes on one CU) and section times from it.
extern "C" int bsclip_gemm_duo_diag(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                                    int epilogue, const bsclip_epi_args* args, unsigned long long* diag, void* stream) {
    BSCLIP_REQUIRE(A && B && C && diag && args, "bsclip_gemm_duo_diag: null pointer");
    BSCLIP_REQUIRE(K % 64 == 0 && N % 128 == 0, "bsclip_gemm_duo_diag: K %% 64, N %% 128");
    EpiArgs e{};
    e.bias = args->bias;
    e.resid = static_cast<const float*>(args->resid);
    e.ld_resid = args->ld_resid;
    e.aux = static_cast<unsigned char*>(args->aux);
    e.ld_aux = args->ld_aux;
    e.drop = make_drop(0.f, 0);
    e.n_total = N;
    e.diag = diag;
    const int tiles_m = ceil_div(M, 256), tiles_n = N / 128;
    const dim3 grid(tiles_m * tiles_n), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bf16_t* a = static_cast<const bf16_t*>(A);
    const bf16_t* b = static_cast<const bf16_t*>(B);
    switch (epilogue) {
        case BSCLIP_EPI_BF16:
            hipLaunchKernelGGL((gemm_nt_duo_kernel<BSCLIP_EPI_BF16, false, true>), grid, block, 0, s, a, lda, b, ldb, C, ldc, M, N, K, tiles_n, e);
            break;
        case BSCLIP_EPI_GELU_BF16:
            hipLaunchKernelGGL((gemm_nt_duo_kernel<BSCLIP_EPI_GELU_BF16, true, true>), grid, block, 0, s, a, lda, b, ldb, C, ldc, M, N, K, tiles_n, e);
            break;
        case BSCLIP_EPI_RESID_F32:
            hipLaunchKernelGGL((gemm_nt_duo_kernel<BSCLIP_EPI_RESID_F32, true, true>), grid, block, 0, s, a, lda, b, ldb, C, ldc, M, N, K, tiles_n, e);
            break;
        case BSCLIP_EPI_DGELU_BF16:
            hipLaunchKernelGGL((gemm_nt_duo_kernel<BSCLIP_EPI_DGELU_BF16, false, true>), grid, block, 0, s, a, lda, b, ldb, C, ldc, M, N, K, tiles_n, e);
            break;
        default: BSCLIP_REQUIRE(false, "bsclip_gemm_duo_diag: epilogue %d has no diagnostic build", epilogue);
    }
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

// The persistent kernel with per-workgroup stamps, diag[grid * 32] (layout: gemm_pers.h); tools/gemm_pers_phases.py.
extern "C" int bsclip_gemm_pers_diag(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                                     int epilogue, const bsclip_epi_args* args, unsigned long long* diag, int workgroups,
                                     void* stream) {
    BSCLIP_REQUIRE(A && B && C && diag && args, "bsclip_gemm_pers_diag: null pointer");
    BSCLIP_REQUIRE(K % 64 == 0 && K >= 128 && N % 256 == 0 && workgroups > 0, "bsclip_gemm_pers_diag: K %% 64, K >= 128, N %% 256");
    EpiArgs e{};
    e.bias = args->bias;
    e.resid = static_cast<const float*>(args->resid);
    e.ld_resid = args->ld_resid;
    e.aux = static_cast<unsigned char*>(args->aux);
    e.ld_aux = args->ld_aux;
    e.drop = make_drop(0.f, 0);
    e.n_total = N;
    e.diag = diag;
    const int tiles_m = ceil_div(M, 256), tiles_n = N / 256, nt = tiles_m * tiles_n;
    e.pers_gw = tiles_n | (getenv("BSCLIP_PERS_ABL") && atoi(getenv("BSCLIP_PERS_ABL")) == 1 ? 0x40000000 : 0);
    const dim3 grid(nt < workgroups ? nt : workgroups), block(512);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bf16_t* a = static_cast<const bf16_t*>(A);
    const bf16_t* b = static_cast<const bf16_t*>(B);
#define PERS_DIAG(EPI, HB)                                                                                                  \
    case EPI:                                                                                                               \
        hipLaunchKernelGGL((gemm_nt_pers_kernel<EPI, HB, true>), grid, block, 0, s, a, lda, b, ldb, C, ldc, M, N, K, tiles_n, nt, e); \
        break;
    switch (epilogue) {
        PERS_DIAG(BSCLIP_EPI_BF16, false)
        PERS_DIAG(BSCLIP_EPI_GELU_BF16, true)
        PERS_DIAG(BSCLIP_EPI_RESID_F32, true)
        PERS_DIAG(BSCLIP_EPI_RESID_BF16, true)
        PERS_DIAG(BSCLIP_EPI_DGELU_BF16, false)
        default: BSCLIP_REQUIRE(false, "bsclip_gemm_pers_diag: epilogue %d has no diagnostic build", epilogue);
    }
#undef PERS_DIAG
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}
#endif  // BSCLIP_DIAG

// Fills the device-side GELU table.  Stream-ordered; the GEMM entry point also does this lazily on its own stream, so a
// single-stream caller never needs it -- callers that launch GEMMs on several streams call it once up front.
extern "C" int bsclip_init_tables(void* stream) {
    hipLaunchKernelGGL(gelu_lut_init_kernel, dim3(ceil_div(GELU_LUT_N + 1, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream));
    BSCLIP_LAUNCH_CHECK();
    g_lut_ready = true;
    return BSCLIP_OK;
}

extern "C" int bsclip_gemm_set_tile(int tile) {
#ifdef BSCLIP_DIAG
    BSCLIP_REQUIRE(tile >= 0 && tile <= 8, "bsclip_gemm_set_tile: tile %d not in 0..8", tile);
#else
    BSCLIP_REQUIRE((tile >= 0 && tile <= 5) || tile == 8, "bsclip_gemm_set_tile: tile %d not in 0..5, 8", tile);
#endif
    g_tile_override = tile;
    return BSCLIP_OK;
}

extern "C" int bsclip_gemm_set_persistent_grid(int workgroups) {
    BSCLIP_REQUIRE(workgroups >= 0 && workgroups <= 4096, "bsclip_gemm_set_persistent_grid: %d not in 0..4096", workgroups);
    g_pers_grid = workgroups;
    return BSCLIP_OK;
}

#ifdef BSCLIP_DIAG
extern "C" int bsclip_gemm_diag_ablate(int mask) {
    BSCLIP_REQUIRE(mask >= 0 && mask < 16, "bsclip_gemm_diag_ablate: mask %d", mask);
    g_diag_ablate = mask;
    return BSCLIP_OK;
}
#endif

extern "C" int bsclip_epi_args_size(void) { return (int)sizeof(bsclip_epi_args); }

extern "C" int bsclip_gemm_bf16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                                int epilogue, const bsclip_epi_args* args, void* stream) {
    BSCLIP_REQUIRE(A && B && C, "bsclip_gemm_bf16: null operand");
    BSCLIP_REQUIRE(M > 0 && N > 0 && K > 0, "bsclip_gemm_bf16: bad shape M=%d N=%d K=%d", M, N, K);
    BSCLIP_REQUIRE(K % 64 == 0, "bsclip_gemm_bf16: K=%d must be a multiple of 64", K);
    BSCLIP_REQUIRE(N % 128 == 0, "bsclip_gemm_bf16: N=%d must be a multiple of 128", N);
    BSCLIP_REQUIRE(lda >= K && ldb >= K && lda % 8 == 0 && ldb % 8 == 0, "bsclip_gemm_bf16: lda=%d ldb=%d (K=%d)", lda,
                   ldb, K);
    BSCLIP_REQUIRE(ldc >= N && ldc % 4 == 0, "bsclip_gemm_bf16: ldc=%d (N=%d)", ldc, N);
    BSCLIP_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) == 0, "bsclip_gemm_bf16: 16-B alignment");
    EpiArgs e{};
    if (args) {
        BSCLIP_REQUIRE(args->struct_size == sizeof(bsclip_epi_args),
                       "bsclip_gemm_bf16: args->struct_size=%u, this library's bsclip_epi_args is %zu bytes (binding out of date?)",
                       args->struct_size, sizeof(bsclip_epi_args));
        e.bias = args->bias;
        e.resid = static_cast<const float*>(args->resid);
        e.ld_resid = args->ld_resid;
        e.aux = static_cast<unsigned char*>(args->aux);
        e.ld_aux = args->ld_aux;
        BSCLIP_REQUIRE(!e.aux || (e.ld_aux >= N && e.ld_aux % 16 == 0 && (((uintptr_t)e.aux) & 15) == 0),
                       "bsclip_gemm_bf16: aux (8-bit gelu' band) needs ld_aux >= N, ld_aux %% 16 == 0, 16-B alignment (ld_aux=%d)",
                       args->ld_aux);
        BSCLIP_REQUIRE(args->dropout_p >= 0.f && args->dropout_p < 1.f, "bsclip_gemm_bf16: dropout_p=%f", args->dropout_p);
        BSCLIP_REQUIRE(args->dropout_p == 0.f || epilogue == BSCLIP_EPI_RESID_F32 || epilogue == BSCLIP_EPI_RESID_BF16,
                       "bsclip_gemm_bf16: dropout is only defined for BSCLIP_EPI_RESID_F32 / _BF16");
        e.drop = make_drop(args->dropout_p, args->dropout_seed);
    }
    e.n_total = N;
    BSCLIP_REQUIRE(!e.bias || (((uintptr_t)e.bias) & 15) == 0, "bsclip_gemm_bf16: bias must be 16-B aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bf16_t* a = static_cast<const bf16_t*>(A);
    const bf16_t* b = static_cast<const bf16_t*>(B);
    switch (epilogue) {
        case BSCLIP_EPI_BF16: launch_bias<BSCLIP_EPI_BF16>(a, lda, b, ldb, C, ldc, M, N, K, e, s); break;
        case BSCLIP_EPI_F32: launch_bias<BSCLIP_EPI_F32>(a, lda, b, ldb, C, ldc, M, N, K, e, s); break;
        case BSCLIP_EPI_GELU_BF16: launch_bias<BSCLIP_EPI_GELU_BF16>(a, lda, b, ldb, C, ldc, M, N, K, e, s); break;
        case BSCLIP_EPI_RESID_F32:
            BSCLIP_REQUIRE(e.resid && e.ld_resid >= N, "bsclip_gemm_bf16: RESID needs resid/ld_resid");
            launch_bias<BSCLIP_EPI_RESID_F32>(a, lda, b, ldb, C, ldc, M, N, K, e, s);
            break;
        case BSCLIP_EPI_DGELU_BF16:
            BSCLIP_REQUIRE(e.aux && e.ld_aux >= N, "bsclip_gemm_bf16: DGELU needs aux/ld_aux");
            launch_bias<BSCLIP_EPI_DGELU_BF16>(a, lda, b, ldb, C, ldc, M, N, K, e, s);
            break;
        case BSCLIP_EPI_PATCH_F32:
            BSCLIP_REQUIRE(e.resid && e.ld_resid >= N && M % 196 == 0, "bsclip_gemm_bf16: PATCH needs pos, M%%196==0");
            launch_bias<BSCLIP_EPI_PATCH_F32>(a, lda, b, ldb, C, ldc, M, N, K, e, s);
            break;
        case BSCLIP_EPI_RESID_BF16:
            BSCLIP_REQUIRE(e.resid && e.ld_resid >= N && e.ld_resid % 4 == 0 && (((uintptr_t)e.resid) & 7) == 0,
                           "bsclip_gemm_bf16: RESID_BF16 needs resid (bf16, 8-byte aligned rows) / ld_resid");
            launch_bias<BSCLIP_EPI_RESID_BF16>(a, lda, b, ldb, C, ldc, M, N, K, e, s);
            break;
        case BSCLIP_EPI_PATCH_BF16:
            BSCLIP_REQUIRE(e.resid && e.ld_resid >= N && M % 196 == 0, "bsclip_gemm_bf16: PATCH needs pos, M%%196==0");
            launch_bias<BSCLIP_EPI_PATCH_BF16>(a, lda, b, ldb, C, ldc, M, N, K, e, s);
            break;
        default: BSCLIP_REQUIRE(false, "bsclip_gemm_bf16: unknown epilogue %d", epilogue);
    }
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// split-K product with f32 accumulation into C: the weight-gradient GEMMs of full fine-tuning
// ---------------------------------------------------------------------------------------------------------------
// dW[N_out, K_in] = dY^T X has a small output (9 .. 36 tiles of 256 x 256) and a reduction over all tokens (50 432 for the ViT
// at batch 256): one workgroup per output tile would use 4 .. 14 % of the chip.  The reduction is cut into `splits` equal
// column ranges, every (tile, range) pair is one workgroup of the ping-pong kernel writing an f32 slab, and the slabs are
// summed in the fixed order z = 0 .. splits - 1 and added to C (no atomics: the result does not depend on the schedule).
__global__ __launch_bounds__(256) void splitk_reduce_add_kernel(const float* __restrict__ partial, int splits, int M, int N,
                                                                float* __restrict__ C, int ldc) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const size_t total = (size_t)M * N;
    if (i >= total) return;
    f32x4 acc = *reinterpret_cast<const f32x4*>(partial + i);
    for (int z = 1; z < splits; ++z) acc += *reinterpret_cast<const f32x4*>(partial + (size_t)z * total + i);
    const size_t m = i / N, n = i - m * N;
    f32x4* dst = reinterpret_cast<f32x4*>(C + m * ldc + n);
    *dst = *dst + acc;
}

extern "C" int bsclip_gemm_splitk_f32(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K,
                                      int splits, float* partial, void* stream) {
    BSCLIP_REQUIRE(A && B && C && partial, "bsclip_gemm_splitk_f32: null operand");
    BSCLIP_REQUIRE(M > 0 && N > 0 && N % 256 == 0 && splits >= 1 && K > 0 && K % (64 * splits) == 0,
                   "bsclip_gemm_splitk_f32: M=%d N=%d (multiple of 256) K=%d (multiple of 64 * splits=%d)", M, N, K, splits);
    BSCLIP_REQUIRE(lda >= K && ldb >= K && lda % 8 == 0 && ldb % 8 == 0 && ldc >= N && ldc % 4 == 0,
                   "bsclip_gemm_splitk_f32: lda=%d ldb=%d ldc=%d", lda, ldb, ldc);
    BSCLIP_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)partial) & 15) == 0, "bsclip_gemm_splitk_f32: 16-B alignment");
    EpiArgs e{};
    e.n_total = N;
    e.drop = make_drop(0.f, 0);
    e.split_stride = (long)M * N;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int tiles_m = ceil_div(M, 256), tiles_n = N / 256;
    hipLaunchKernelGGL((gemm_nt_pp_kernel<BSCLIP_EPI_F32, false>), dim3(tiles_m * tiles_n, splits), dim3(512), 0, s,
                       static_cast<const bf16_t*>(A), lda, static_cast<const bf16_t*>(B), ldb, partial, N, M, N, K / splits, tiles_n, e);
    hipLaunchKernelGGL(splitk_reduce_add_kernel, dim3(ceil_div((int)((long)M * N / 4), 256)), dim3(256), 0, s, partial, splits, M, N, C, ldc);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// fused InfoNCE products (called by loss.hip; not part of the public ABI)
// ---------------------------------------------------------------------------------------------------------------
// mode 0: pass 1 -- part[M, N/256, 4] from cosines A[M,K] . B[N,K]^T;  mode 1: pass 2 -- C = dL/dG in split form [M, 3*Np].
int bsclip_gemm_infonce(int mode, const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                        const int64_t* labels, const float* cnt, const float* lse_row, const float* lse_col, float* part,
                        float logit_scale, float coef, int n_valid, int row_base, void* stream) {
    BSCLIP_REQUIRE(A && B && labels && M > 0 && N % 256 == 0 && K % 64 == 0, "bsclip_gemm_infonce: bad shape M=%d N=%d K=%d", M, N, K);
    EpiArgs e{};
    e.labels = labels;
    e.cnt = cnt;
    e.lse_row = lse_row;
    e.lse_col = lse_col;
    e.part = part;
    e.logit_scale = logit_scale;
    e.coef = coef;
    e.n_valid = n_valid;
    e.row_base = row_base;
    e.n_total = N;
    e.drop = make_drop(0.f, 0);
    const int tiles_m = ceil_div(M, 256), tiles_n = N / 256;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bf16_t* a = static_cast<const bf16_t*>(A);
    const bf16_t* b = static_cast<const bf16_t*>(B);
    if (mode == 0) {
        BSCLIP_REQUIRE(part, "bsclip_gemm_infonce: part is null");
        hipLaunchKernelGGL((gemm_nt_pp_kernel<EPI_LSE_PART, false>), dim3(tiles_m * tiles_n), dim3(512), 0, s, a, lda, b, ldb, C,
                           ldc, M, N, K, tiles_n, e);
    } else {
        BSCLIP_REQUIRE(C && cnt && lse_row && lse_col && ldc % 3 == 0, "bsclip_gemm_infonce: pass 2 operands");
        hipLaunchKernelGGL((gemm_nt_pp_kernel<EPI_LOSS_W, false>), dim3(tiles_m * tiles_n), dim3(512), 0, s, a, lda, b, ldb, C,
                           ldc, M, N, K, tiles_n, e);
    }
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// fp8 entry points (BASELINE configs[4])
// ---------------------------------------------------------------------------------------------------------------
extern "C" int bsclip_gemm_fp8(const void* A8, int lda, const void* B8, int ldb, void* C, int ldc, int M, int N, int K,
                               int epilogue, const bsclip_epi_args* args, const bsclip_fp8_args* f8, void* stream) {
    BSCLIP_REQUIRE(A8 && B8 && C && args && f8, "bsclip_gemm_fp8: null pointer");
    BSCLIP_REQUIRE(args->struct_size == sizeof(bsclip_epi_args) && f8->struct_size == sizeof(bsclip_fp8_args),
                   "bsclip_gemm_fp8: struct_size mismatch (epi %u/%zu, fp8 %u/%zu)", args->struct_size,
                   sizeof(bsclip_epi_args), f8->struct_size, sizeof(bsclip_fp8_args));
    BSCLIP_REQUIRE(M > 0 && N > 0 && K > 0 && K % 128 == 0 && N % 256 == 0, "bsclip_gemm_fp8: M=%d N=%d K=%d (K %% 128, N %% 256)",
                   M, N, K);
    BSCLIP_REQUIRE(lda >= K && ldb >= K && lda % 16 == 0 && ldb % 16 == 0, "bsclip_gemm_fp8: lda=%d ldb=%d (K=%d)", lda, ldb, K);
    BSCLIP_REQUIRE(ldc >= N && ldc % 4 == 0 && (epilogue != BSCLIP_EPI_GELU_FP8 || ldc % 16 == 0), "bsclip_gemm_fp8: ldc=%d", ldc);
    BSCLIP_REQUIRE((((uintptr_t)A8 | (uintptr_t)B8 | (uintptr_t)C) & 15) == 0, "bsclip_gemm_fp8: 16-B alignment");
    BSCLIP_REQUIRE(args->bias && f8->alpha && ((((uintptr_t)args->bias | (uintptr_t)f8->alpha)) & 15) == 0,
                   "bsclip_gemm_fp8: bias and alpha are required, 16-B aligned");
    BSCLIP_REQUIRE((f8->a_aug == nullptr) == (f8->b_aug == nullptr), "bsclip_gemm_fp8: a_aug and b_aug go together");
    BSCLIP_REQUIRE(!f8->a_aug || (f8->ld_a_aug >= 64 && f8->ld_b_aug >= 64 && f8->ld_a_aug % 8 == 0 && f8->ld_b_aug % 8 == 0 &&
                                  (((uintptr_t)f8->a_aug | (uintptr_t)f8->b_aug) & 15) == 0),
                   "bsclip_gemm_fp8: K-augmentation block needs ld >= 64, ld %% 8 == 0, 16-B alignment");
    BSCLIP_REQUIRE(f8->form == 1 || f8->form == 2, "bsclip_gemm_fp8: form %d (1 or 2)", f8->form);
    EpiArgs e{};
    e.bias = args->bias;
    e.resid = static_cast<const float*>(args->resid);
    e.ld_resid = args->ld_resid;
    e.aux = static_cast<unsigned char*>(args->aux);
    e.ld_aux = args->ld_aux;
    BSCLIP_REQUIRE(!e.aux || (e.ld_aux >= N && e.ld_aux % 16 == 0 && (((uintptr_t)e.aux) & 15) == 0), "bsclip_gemm_fp8: aux band");
    BSCLIP_REQUIRE(args->dropout_p >= 0.f && args->dropout_p < 1.f && (args->dropout_p == 0.f || epilogue == BSCLIP_EPI_RESID_F32),
                   "bsclip_gemm_fp8: dropout_p=%f", args->dropout_p);
    e.drop = make_drop(args->dropout_p, args->dropout_seed);
    e.n_total = N;
    e.alpha = f8->alpha;
    e.a_aug = static_cast<const bf16_t*>(f8->a_aug);
    e.b_aug = static_cast<const bf16_t*>(f8->b_aug);
    e.ld_a_aug = f8->ld_a_aug;
    e.ld_b_aug = f8->ld_b_aug;
    hipStream_t s = static_cast<hipStream_t>(stream);
#define F8_CASE(EPI)                                                                   \
    case EPI:                                                                          \
        if (f8->form == 1) launch_f8<EPI, 1>(A8, lda, B8, ldb, C, ldc, M, N, K, e, s); \
        else launch_f8<EPI, 2>(A8, lda, B8, ldb, C, ldc, M, N, K, e, s);               \
        break;
    switch (epilogue) {
        F8_CASE(BSCLIP_EPI_BF16)
        F8_CASE(BSCLIP_EPI_F32)
        F8_CASE(BSCLIP_EPI_GELU_FP8)
        case BSCLIP_EPI_RESID_F32:
            BSCLIP_REQUIRE(e.resid && e.ld_resid >= N, "bsclip_gemm_fp8: RESID needs resid/ld_resid");
            if (f8->form == 1) launch_f8<BSCLIP_EPI_RESID_F32, 1>(A8, lda, B8, ldb, C, ldc, M, N, K, e, s);
            else launch_f8<BSCLIP_EPI_RESID_F32, 2>(A8, lda, B8, ldb, C, ldc, M, N, K, e, s);
            break;
        default: BSCLIP_REQUIRE(false, "bsclip_gemm_fp8: epilogue %d not available with fp8 operands", epilogue);
    }
#undef F8_CASE
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

namespace {
// one wave per row: amax -> scale = amax / 448, then e4m3 codes of src / scale
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(const float* __restrict__ src, int R, int C,
                                                                unsigned char* __restrict__ dst, int ld, float* __restrict__ scale) {
    const int lane = threadIdx.x & 63;
    const int row = (blockIdx.x * 256 + threadIdx.x) >> 6;
    if (row >= R) return;
    const float* p = src + (size_t)row * C;
    float m = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + c);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    m = wave_max(m);
    const float sc = m > 0.f ? m * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / sc;
    if (lane == 0) scale[row] = sc;
    for (int c = lane * 4; c < C; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + c);
        *reinterpret_cast<unsigned*>(dst + (size_t)row * ld + c) = pack_fp8x4(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv);
    }
}

__global__ void lora_baug_set_kernel(bf16_t* __restrict__ b_aug, int ld, int H, const float* __restrict__ bq,
                                     const float* __restrict__ bv, const float* __restrict__ alpha) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // over H * 4
    if (i >= H * 4) return;
    const int n = i >> 2, r = i & 3;
    b_aug[(size_t)n * ld + r] = f2bf(bq[i] / alpha[n]);
    b_aug[(size_t)(2 * H + n) * ld + 4 + r] = f2bf(bv[i] / alpha[2 * H + n]);
}
}  // namespace

extern "C" int bsclip_quantize_rows_fp8(const float* src, int R, int C, void* dst, int ld_dst, float* scale, void* stream) {
    BSCLIP_REQUIRE(src && dst && scale && R > 0 && C > 0 && C % 4 == 0 && ld_dst >= C && ld_dst % 4 == 0,
                   "bsclip_quantize_rows_fp8: R=%d C=%d ld_dst=%d", R, C, ld_dst);
    BSCLIP_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "bsclip_quantize_rows_fp8: 16-B alignment");
    hipLaunchKernelGGL(quantize_rows_fp8_kernel, dim3(ceil_div(R, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), src, R, C,
                       static_cast<unsigned char*>(dst), ld_dst, scale);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int bsclip_lora_baug_set(void* b_aug, int ld_b, int H, const float* lora_bq, const float* lora_bv,
                                    const float* alpha, void* stream) {
    BSCLIP_REQUIRE(b_aug && lora_bq && lora_bv && alpha && H > 0 && ld_b >= 8, "bsclip_lora_baug_set: bad args");
    hipLaunchKernelGGL(lora_baug_set_kernel, dim3(ceil_div(H * 4, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<bf16_t*>(b_aug), ld_b, H, lora_bq, lora_bv, alpha);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}
