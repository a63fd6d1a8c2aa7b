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
    const float* resid;
    int ld_resid;
    bf16_t* aux;
    int ld_aux;
    DropCfg drop;  // RESID only: C = dropout(acc + bias) + resid  (HF BertSelfOutput / BertOutput)
    int n_total;   // logical row width for the dropout element index
    unsigned long long* diag;  // diagnostic build only: per-workgroup phase stamps (100 MHz wall clock)
};

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
        float gl[4], dg[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) gelu_both(v[i], gl[i], dg[i]);
        if (e.aux) {  // store only: gelu'(pre-activation), all the backward pass needs
            uint2 z;
            z.x = pack_bf2(dg[0], dg[1]);
            z.y = pack_bf2(dg[2], dg[3]);
            *reinterpret_cast<uint2*>(e.aux + (size_t)m * e.ld_aux + n) = z;
        }
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
        const uint2 z = *reinterpret_cast<const uint2*>(e.aux + (size_t)m * e.ld_aux + n);
        uint2 o;
        o.x = pack_bf2(v[0] * bf2f(z.x & 0xffff), v[1] * bf2f(z.x >> 16));
        o.y = pack_bf2(v[2] * bf2f(z.y & 0xffff), v[3] * bf2f(z.y >> 16));
        *reinterpret_cast<uint2*>(static_cast<bf16_t*>(C) + (size_t)m * ldc + n) = o;
    } else if constexpr (EPI == BSCLIP_EPI_PATCH_F32) {
        const int b = m / 196, p = m - b * 196;
        const f32x4 pos = *reinterpret_cast<const f32x4*>(e.resid + (size_t)(1 + p) * e.ld_resid + n);
        v += pos;
        *reinterpret_cast<f32x4*>(static_cast<float*>(C) + (size_t)(b * 197 + 1 + p) * ldc + n) = v;
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
// GELU table for the 256x256 kernel's epilogue
// ---------------------------------------------------------------------------------------------------------------
// The two-output GELU epilogue (gelu and gelu' of 128 values per lane) computed with exp + rcp was VALU-bound (~24
// instructions per value, 12 us per tile with no MFMA to hide behind), and a 16-B-per-entry interpolation table was bound by
// the LDS array instead: 64 lanes gathering random 16-B entries conflict ~3x per 16-lane group (phase stamps: 4.4 us per
// 64-row slab).  Entries are therefore 8 B -- {Phi(x_i), phi(x_i)} on a 1/64 grid over [-8, 8], nearest grid point, one
// ds_read_b64 per value -- and the neighbourhood comes from the derivatives, which are free: Phi' = phi, phi' = -x phi, so
//   Phi(x_i + d) = Phi_i + d phi_i (1 - x_i d / 2) + O(d^3 |phi''| / 6)   <= 2e-8   for |d| <= 1/128,
//   phi(x_i + d) = phi_i (1 - x_i d)               + O(d^2 |phi''| / 2)   <= 1.3e-5,
// both far inside the bf16 rounding of the outputs.  Computed once on the device in f64 with erf().
constexpr int GELU_LUT_N = 1024;                      // intervals
constexpr int GELU_LUT_BYTES = (GELU_LUT_N + 1) * 8;  // 8200
__device__ float2 g_gelu_lut[GELU_LUT_N + 1];

__global__ void gelu_lut_init_kernel() {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > GELU_LUT_N) return;
    const double x = -8.0 + i / 64.0;
    g_gelu_lut[i] = float2{(float)(0.5 * (1.0 + erf(x * 0.70710678118654752440))),
                           (float)(0.39894228040143267794 * exp(-0.5 * x * x))};
}

// Two values per call so that the arithmetic runs on the packed-f32 VALU (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32);
// the clamp, the float<->int conversions and the table address stay per value.  Phi is taken to first order as well
// (error d^2 |phi'| / 2 <= 7.4e-6 for |d| <= 1/128).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_lut2(const char* lut, f32x2 x, f32x2& gl, f32x2& dg) {
    f32x2 xc, fi, Phi, phi;
#pragma unroll
    for (int k = 0; k < 2; ++k) xc[k] = __builtin_amdgcn_fmed3f(x[k], -8.0f, 8.0f);
    const f32x2 t = xc * 64.0f + 512.5f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = (int)t[k];  // t in [0.5, 1024.5]: truncation = nearest grid point
        const float2 e = *reinterpret_cast<const float2*>(lut + i * 8);
        fi[k] = (float)i;
        Phi[k] = e.x;
        phi[k] = e.y;
    }
    const f32x2 xi = fi * 0.015625f - 8.0f;
    const f32x2 d = xc - xi;
    const f32x2 cdf = d * phi + Phi;
    const f32x2 pdf = phi - (xi * d) * phi;
    gl = x * cdf;
    dg = x * pdf + cdf;
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
template <int EPI, bool HAS_BIAS, bool DIAG = false, int ABL = 0, int SCHED = 0>
__global__ __launch_bounds__(512) void gemm_nt_pp_kernel(const bf16_t* __restrict__ A, int lda,
                                                          const bf16_t* __restrict__ B, int ldb, void* __restrict__ C,
                                                          int ldc, int M, int N, int K, int tiles_n, EpiArgs e) {
    constexpr int SET = 65536, HALF = 16384, B_OFF = 32768;
    // main loop 128 KiB; epilogue slabs 2x64x1040 (f32) or 4x64x528 (bf16 x2) = 132 KiB, + 16 KiB GELU table
    __shared__ __attribute__((aligned(16))) char smem[4 * 64 * 528 + (EPI == BSCLIP_EPI_GELU_BF16 ? GELU_LUT_BYTES : 0)];

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = wg % tiles_n, tile_m = wg / tiles_n;
    const int m0 = tile_m * 256, n0 = tile_n * 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, wc = wave & 3;

    // ---- LDS-DMA sources: for half h, this wave moves chunks (wave) and (wave+8): rows 128h + 8*chunk + (lane>>3) ----
    const bf16_t* srcA[2][2];
    const bf16_t* srcB[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 128 * h + 8 * (wave + 8 * i) + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            srcA[h][i] = A + (size_t)min(m0 + row, M - 1) * lda + c * 8;
            srcB[h][i] = B + (size_t)min(n0 + row, N - 1) * ldb + c * 8;
        }
    const int dma_off = wave * 1024;  // + i*8192 + half*HALF (+ B_OFF) + set*SET
    auto dmaA = [&](int set, int h, int k0) {
        if constexpr (ABL & 4) return;
        char* d = smem + set * SET + h * HALF + dma_off;
        glds16(srcA[h][0] + k0, d);
        glds16(srcA[h][1] + k0, d + 8192);
    };
    auto dmaB = [&](int set, int h, int k0) {
        if constexpr (ABL & 4) return;
        char* d = smem + set * SET + B_OFF + h * HALF + dma_off;
        glds16(srcB[h][0] + k0, d);
        glds16(srcB[h][1] + k0, d + 8192);
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
    auto readA = [&](const char* base, int mi) {
        if constexpr (ABL & 2) return;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                fa[mi][ks][i] = *reinterpret_cast<const bf16x8*>(base + ((a_off ^ (ks << 6)) + (64 * mi + 16 * i) * ROW_BYTES));
    };
    auto readB = [&](const char* base, int ni) {
        if constexpr (ABL & 2) return;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                fb[ni][ks][j] = *reinterpret_cast<const bf16x8*>(base + ((b_off ^ (ks << 6)) + (32 * ni + 16 * j) * ROW_BYTES));
    };
    auto mma = [&](int mi, int ni) {
        if constexpr (ABL & 1) return;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[mi][ni][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ni][ks][j], fa[mi][ks][i], acc[mi][ni][i][j], 0, 0, 0);
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
    const int nk = K / BK;
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
            dmaA(1, 0, BK);
            dmaA(1, 1, BK);
            dmaB(1, 0, BK);
            dmaB(1, 1, BK);
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
            const int k2 = (t + 2) * BK;
            // ---- phase 0 ----
            readA(base, 0);
            readB(base, 0);
            PP_BARRIER();
            mma(0, 0);
            PP_BARRIER();
            // ---- phase 1 ----
            readA(base, 1);
            PP_BARRIER();
            mma(1, 0);
            PP_BARRIER();
            // ---- phase 2 ----
            readB(base, 1);
            PP_BARRIER();
            if (has2) {
                dmaA(set, 0, k2);
                dmaA(set, 1, k2);
            }
            mma(1, 1);
            PP_BARRIER();
            // ---- phase 3 ----
            if (has2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // tile t+1 complete; A of tile t+2 may fly
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PP_BARRIER();
            if (has2) {
                dmaB(set, 0, k2);
                dmaB(set, 1, k2);
            }
            mma(0, 1);
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
            dmaA(1, 0, BK);
            if (g == 1) {
                dmaA(1, 1, BK);
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
            const int k1 = (t + 1) * BK, k2 = (t + 2) * BK;
            // ---- phase 0 ----
            BAR_G1();
            if (has1) {
                if (g == 0) dmaA(set ^ 1, 1, k1);
                else dmaB(set ^ 1, 0, k1);
            }
            readA(base, 0);
            readB(base, 0);
            BAR_G0();
            mma(0, 0);
            // ---- phase 1 ----
            BAR_G1();
            if (has1) {
                if (g == 0) dmaB(set ^ 1, 0, k1);
                else dmaB(set ^ 1, 1, k1);
            }
            readA(base, 1);
            BAR_G0();
            mma(1, 0);
            // ---- phase 2 ----
            BAR_G1();
            if (has1 && g == 0) dmaB(set ^ 1, 1, k1);
            readB(base, 1);
            BAR_G0();
            mma(1, 1);
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
            mma(0, 1);
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
            dmaA(1, 0, BK);
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PP_BARRIER();
        if (g == 1) PP_BARRIER();  // group 1 runs one barrier behind group 0 from here on
        stamp(1);

        for (int t = 0; t < nk; ++t) {
            const int set = t & 1;
            const char* base = smem + set * SET;
            const bool has1 = t + 1 < nk, has2 = t + 2 < nk;
            const int k1 = (t + 1) * BK, k2 = (t + 2) * BK;
            // ---- phase 0 ----
            if (has1) dmaA(set ^ 1, 1, k1);
            readA(base, 0);
            readB(base, 0);
            PP_BARRIER();
            mma(0, 0);
            PP_BARRIER();
            // ---- phase 1 ----
            if (has1) dmaB(set ^ 1, 0, k1);
            readA(base, 1);
            PP_BARRIER();
            mma(1, 0);
            PP_BARRIER();
            // ---- phase 2 ----
            if (has1) dmaB(set ^ 1, 1, k1);
            readB(base, 1);
            PP_BARRIER();
            mma(1, 1);
            PP_BARRIER();
            // ---- phase 3 ----
            if (has2) {
                dmaA(set, 0, k2);
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // everything of tile t+1 has landed; A-half0(t+2) may fly
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            PP_BARRIER();
            mma(0, 1);
            PP_BARRIER();
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
    if constexpr (EPI == BSCLIP_EPI_GELU_BF16) {  // 16 KiB table, L2-resident, behind the slabs
        for (int i = tid; i <= GELU_LUT_N; i += 512)
            *reinterpret_cast<float2*>(smem + 4 * 64 * SB + i * 8) = g_gelu_lut[i];
        __syncthreads();
    }
    char* slab2 = smem + 2 * 64 * SB + g * (64 * SB);  // second bf16 slab (GELU: gelu' side band)
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
                    if constexpr (EPI == BSCLIP_EPI_GELU_BF16) {
                        f32x2 gl0, dg0, gl1, dg1;
                        gelu_lut2(lut, f32x2{v[0], v[1]}, gl0, dg0);
                        gelu_lut2(lut, f32x2{v[2], v[3]}, gl1, dg1);
                        o.x = pack_bf2(gl0[0], gl0[1]);
                        o.y = pack_bf2(gl1[0], gl1[1]);
                        uint2 d;
                        d.x = pack_bf2(dg0[0], dg0[1]);
                        d.y = pack_bf2(dg1[0], dg1[1]);
                        *reinterpret_cast<uint2*>(slab2 + off) = d;
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
            if (m < M) *reinterpret_cast<uint4*>(dst + (size_t)m * ld + n0 + (lane & 31) * 8) = v;
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
        if constexpr (EPI == BSCLIP_EPI_BF16 || EPI == BSCLIP_EPI_GELU_BF16) {
            stage_bf16(mi);
            __syncthreads();
            stamp(4 + 2 * mi);
            rows_bf16(mi, smem + g * (64 * SB), static_cast<bf16_t*>(C), ldc);
            if constexpr (EPI == BSCLIP_EPI_GELU_BF16) {
                if (e.aux) rows_bf16(mi, slab2, e.aux, e.ld_aux);  // gelu'(pre-activation) for the backward pass
            }
            __syncthreads();
            stamp(5 + 2 * mi);
        }
    }
    if constexpr (!(EPI == BSCLIP_EPI_BF16 || EPI == BSCLIP_EPI_GELU_BF16)) {
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
                } else if constexpr (EPI == BSCLIP_EPI_DGELU_BF16) {
                    const uint2 z = *reinterpret_cast<const uint2*>(e.aux + (size_t)m * e.ld_aux + n);
                    R[it] = f32x4{bf2f(z.x & 0xffff), bf2f(z.x >> 16), bf2f(z.y & 0xffff), bf2f(z.y >> 16)};
                } else if constexpr (EPI == BSCLIP_EPI_PATCH_F32) {
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
                        *reinterpret_cast<f32x4*>(static_cast<float*>(C) + (size_t)m * ldc + n) = v;
                    } else if constexpr (EPI == BSCLIP_EPI_RESID_F32) {
                        if (e.drop.thr16) v = drop4(e.drop, (unsigned)m * (unsigned)e.n_total + (unsigned)n, v);
                        v += R[it];
                        *reinterpret_cast<f32x4*>(static_cast<float*>(C) + (size_t)m * ldc + n) = v;
                    } else if constexpr (EPI == BSCLIP_EPI_DGELU_BF16) {
                        v *= R[it];
                        uint2 o;
                        o.x = pack_bf2(v[0], v[1]);
                        o.y = pack_bf2(v[2], v[3]);
                        *reinterpret_cast<uint2*>(static_cast<bf16_t*>(C) + (size_t)m * ldc + n) = o;
                    } else if constexpr (EPI == BSCLIP_EPI_PATCH_F32) {
                        const int b = m / 196, p = m - b * 196;
                        v += R[it];
                        *reinterpret_cast<f32x4*>(static_cast<float*>(C) + (size_t)(b * 197 + 1 + p) * ldc + n) = v;
                    }
                }
            }
        };
        prefetch(0, pre[0]);
        stage_f32(0);
        __syncthreads();
        stamp(4);
        prefetch(1, pre[1]);
        consume(0, pre[0]);
        __syncthreads();
        stamp(5);
        stage_f32(1);
        __syncthreads();
        stamp(6);
        consume(1, pre[1]);
        stamp(7);
    }
    stamp(3);
}

int g_tile_override = 0;
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
    if (EPI == BSCLIP_EPI_GELU_BF16 && !g_lut_ready) {  // once per process, stream-ordered ahead of the first consumer
        hipLaunchKernelGGL(gelu_lut_init_kernel, dim3(ceil_div(GELU_LUT_N + 1, 256)), dim3(256), 0, s);
        g_lut_ready = true;
    }
    const int tiles_m = ceil_div(M, 256), tiles_n = N / 256;
    hipLaunchKernelGGL((gemm_nt_pp_kernel<EPI, HB, false, 0, SCHED>), dim3(tiles_m * tiles_n), dim3(512), 0, s, A, lda, B,
                       ldb, C, ldc, M, N, K, tiles_n, e);
}

template <int EPI, bool HB>
void launch_epi(const bf16_t* A, int lda, const bf16_t* B, int ldb, void* C, int ldc, int M, int N, int K,
                const EpiArgs& e, hipStream_t s) {
    int tile = g_tile_override;
    if (tile == 0) {
        // 256x256 (8 waves, 1 block/CU) halves L2->LDS traffic per FLOP; fall back when N is not a multiple of
        // 256 or the grid would not fill the 256 CUs.
        // measured on MI355X (profiles/r01_b_gemm_tiles.log): the ping-pong 256x256 kernel wins on every encoder shape
        // once the grid covers the chip; 128x128 (2 workgroups/CU) is the better small-grid choice.
        const long t256 = (long)ceil_div(M, 256) * (N / 256);
        if (N % 256 == 0 && t256 >= 192) tile = 4;
        else tile = 1;
    }
    if ((tile == 3 || tile == 4 || tile == 6 || tile == 7) && N % 256 != 0) tile = 2;
    switch (tile) {
        case 4: launch_pp<EPI, HB>(A, lda, B, ldb, C, ldc, M, N, K, e, s); break;
#ifdef BSCLIP_DIAG
        case 6: launch_pp<EPI, HB, 1>(A, lda, B, ldb, C, ldc, M, N, K, e, s); break;  // two-tiles-ahead DMA (comparison)
        case 7: launch_pp<EPI, HB, 2>(A, lda, B, ldb, C, ldc, M, N, K, e, s); break;  // four barriers per K-tile (experimental)
#endif
        case 3: launch_cfg<256, 256, 2, 4, EPI, HB>(A, lda, B, ldb, C, ldc, M, N, K, e, s); break;
        case 2: launch_cfg<256, 128, 4, 2, EPI, HB>(A, lda, B, ldb, C, ldc, M, N, K, e, s); break;
        default: launch_cfg<128, 128, 2, 2, EPI, HB>(A, lda, B, ldb, C, ldc, M, N, K, e, s); break;
    }
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
    e.resid = args->resid;
    e.ld_resid = args->ld_resid;
    e.aux = static_cast<bf16_t*>(args->aux);
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
    BSCLIP_REQUIRE(tile >= 0 && tile <= 7 && tile != 5, "bsclip_gemm_set_tile: tile %d not in {0,1,2,3,4,6,7}", tile);
#else
    BSCLIP_REQUIRE(tile >= 0 && tile <= 4, "bsclip_gemm_set_tile: tile %d not in {0,1,2,3,4}", tile);
#endif
    g_tile_override = tile;
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
        e.resid = args->resid;
        e.ld_resid = args->ld_resid;
        e.aux = static_cast<bf16_t*>(args->aux);
        e.ld_aux = args->ld_aux;
        BSCLIP_REQUIRE(args->dropout_p >= 0.f && args->dropout_p < 1.f, "bsclip_gemm_bf16: dropout_p=%f", args->dropout_p);
        BSCLIP_REQUIRE(args->dropout_p == 0.f || epilogue == BSCLIP_EPI_RESID_F32,
                       "bsclip_gemm_bf16: dropout is only defined for BSCLIP_EPI_RESID_F32");
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
        default: BSCLIP_REQUIRE(false, "bsclip_gemm_bf16: unknown epilogue %d", epilogue);
    }
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}
