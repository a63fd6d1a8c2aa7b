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
//     scaling for the backward pass, and the ViT patch-embed row remap + position add.
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
};

template <int EPI>
__device__ __forceinline__ void epilogue_store(f32x4 v, int m, int n, void* C, int ldc, const EpiArgs& e) {
    if (e.bias) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(e.bias + n);
        v += b;
    }
    if constexpr (EPI == BSCLIP_EPI_BF16) {
        uint2 o;
        o.x = pack_bf2(v[0], v[1]);
        o.y = pack_bf2(v[2], v[3]);
        *reinterpret_cast<uint2*>(static_cast<bf16_t*>(C) + (size_t)m * ldc + n) = o;
    } else if constexpr (EPI == BSCLIP_EPI_F32) {
        *reinterpret_cast<f32x4*>(static_cast<float*>(C) + (size_t)m * ldc + n) = v;
    } else if constexpr (EPI == BSCLIP_EPI_GELU_BF16) {
        if (e.aux) {
            uint2 z;
            z.x = pack_bf2(v[0], v[1]);
            z.y = pack_bf2(v[2], v[3]);
            *reinterpret_cast<uint2*>(e.aux + (size_t)m * e.ld_aux + n) = z;
        }
        uint2 o;
        o.x = pack_bf2(gelu_f(v[0]), gelu_f(v[1]));
        o.y = pack_bf2(gelu_f(v[2]), gelu_f(v[3]));
        *reinterpret_cast<uint2*>(static_cast<bf16_t*>(C) + (size_t)m * ldc + n) = o;
    } else if constexpr (EPI == BSCLIP_EPI_RESID_F32) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(e.resid + (size_t)m * e.ld_resid + n);
        v += r;
        *reinterpret_cast<f32x4*>(static_cast<float*>(C) + (size_t)m * ldc + n) = v;
    } else if constexpr (EPI == BSCLIP_EPI_DGELU_BF16) {
        const uint2 z = *reinterpret_cast<const uint2*>(e.aux + (size_t)m * e.ld_aux + n);
        uint2 o;
        o.x = pack_bf2(v[0] * dgelu_f(bf2f(z.x & 0xffff)), v[1] * dgelu_f(bf2f(z.x >> 16)));
        o.y = pack_bf2(v[2] * dgelu_f(bf2f(z.y & 0xffff)), v[3] * dgelu_f(bf2f(z.y >> 16)));
        *reinterpret_cast<uint2*>(static_cast<bf16_t*>(C) + (size_t)m * ldc + n) = o;
    } else if constexpr (EPI == BSCLIP_EPI_PATCH_F32) {
        const int b = m / 196, p = m - b * 196;
        const f32x4 pos = *reinterpret_cast<const f32x4*>(e.resid + (size_t)(1 + p) * e.ld_resid + n);
        v += pos;
        *reinterpret_cast<f32x4*>(static_cast<float*>(C) + (size_t)(b * 197 + 1 + p) * ldc + n) = v;
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI>
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
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * WTM + i * 16 + fr;
        if (m < M) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 16 + fq * 4;
                epilogue_store<EPI>(acc[i][j], m, n, C, ldc, e);
            }
        }
    }
}

int g_tile_override = 0;

template <int BM, int BN, int WM, int WN, int EPI>
int launch_cfg(const bf16_t* A, int lda, const bf16_t* B, int ldb, void* C, int ldc, int M, int N, int K,
               const EpiArgs& e, hipStream_t s) {
    const int tiles_m = ceil_div(M, BM), tiles_n = N / BN;
    const dim3 grid(tiles_m * tiles_n), block(WM * WN * 64);
    hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WM, WN, EPI>), grid, block, 0, s, A, lda, B, ldb, C, ldc, M, N, K,
                       tiles_n, e);
    return 0;
}

template <int EPI>
int launch_epi(const bf16_t* A, int lda, const bf16_t* B, int ldb, void* C, int ldc, int M, int N, int K,
               const EpiArgs& e, hipStream_t s) {
    int tile = g_tile_override;
    if (tile == 0) {
        // 256x256 (8 waves, 1 block/CU) halves L2->LDS traffic per FLOP; fall back when N is not a multiple of
        // 256 or the grid would not fill the 256 CUs.
        const long t256 = (long)ceil_div(M, 256) * (N / 256);
        if (N % 256 == 0 && t256 >= 512) tile = 3;
        else if (M >= 2048) tile = 2;
        else tile = 1;
    }
    if (tile == 3 && N % 256 != 0) tile = 2;
    switch (tile) {
        case 3: return launch_cfg<256, 256, 2, 4, EPI>(A, lda, B, ldb, C, ldc, M, N, K, e, s);
        case 2: return launch_cfg<256, 128, 4, 2, EPI>(A, lda, B, ldb, C, ldc, M, N, K, e, s);
        default: return launch_cfg<128, 128, 2, 2, EPI>(A, lda, B, ldb, C, ldc, M, N, K, e, s);
    }
}

}  // namespace

extern "C" int bsclip_gemm_set_tile(int tile) {
    BSCLIP_REQUIRE(tile >= 0 && tile <= 3, "bsclip_gemm_set_tile: tile %d not in [0,3]", tile);
    g_tile_override = tile;
    return BSCLIP_OK;
}

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
        e.bias = args->bias;
        e.resid = args->resid;
        e.ld_resid = args->ld_resid;
        e.aux = static_cast<bf16_t*>(args->aux);
        e.ld_aux = args->ld_aux;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bf16_t* a = static_cast<const bf16_t*>(A);
    const bf16_t* b = static_cast<const bf16_t*>(B);
    switch (epilogue) {
        case BSCLIP_EPI_BF16: launch_epi<BSCLIP_EPI_BF16>(a, lda, b, ldb, C, ldc, M, N, K, e, s); break;
        case BSCLIP_EPI_F32: launch_epi<BSCLIP_EPI_F32>(a, lda, b, ldb, C, ldc, M, N, K, e, s); break;
        case BSCLIP_EPI_GELU_BF16: launch_epi<BSCLIP_EPI_GELU_BF16>(a, lda, b, ldb, C, ldc, M, N, K, e, s); break;
        case BSCLIP_EPI_RESID_F32:
            BSCLIP_REQUIRE(e.resid && e.ld_resid >= N, "bsclip_gemm_bf16: RESID needs resid/ld_resid");
            launch_epi<BSCLIP_EPI_RESID_F32>(a, lda, b, ldb, C, ldc, M, N, K, e, s);
            break;
        case BSCLIP_EPI_DGELU_BF16:
            BSCLIP_REQUIRE(e.aux && e.ld_aux >= N, "bsclip_gemm_bf16: DGELU needs aux/ld_aux");
            launch_epi<BSCLIP_EPI_DGELU_BF16>(a, lda, b, ldb, C, ldc, M, N, K, e, s);
            break;
        case BSCLIP_EPI_PATCH_F32:
            BSCLIP_REQUIRE(e.resid && e.ld_resid >= N && M % 196 == 0, "bsclip_gemm_bf16: PATCH needs pos, M%%196==0");
            launch_epi<BSCLIP_EPI_PATCH_F32>(a, lda, b, ldb, C, ldc, M, N, K, e, s);
            break;
        default: BSCLIP_REQUIRE(false, "bsclip_gemm_bf16: unknown epilogue %d", epilogue);
    }
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}
