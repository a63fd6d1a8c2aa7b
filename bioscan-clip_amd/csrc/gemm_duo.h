// 256x128 "duo" GEMM kernel: TWO workgroups per CU, so one workgroup's epilogue runs under the other's K loop.
//
// Included by gemm.hip inside its anonymous namespace (shares EpiArgs, the GELU table and the epilogue helpers).
//
// Why: the 256x256 ping-pong kernel owns a CU alone (8 waves x 256 VGPRs, 128 KiB of LDS).  Its K loop runs at the level of
// the best plain-HIP template, but nothing overlaps its epilogue: with K = 768 a tile is 12 K-tiles (17 us) + 10 us (GELU, two
// outputs) or + 16-35 us (f32 residual read + write at the HBM share of a CU) during which the matrix pipe idles
// (profiles/r02_c_fc1_pmc.txt: busy 32 %).  Registers are what forbids a persistent epilogue/K-loop overlap inside one
// workgroup (128 accumulator + 96 fragment VGPRs per wave), so the overlap is bought with occupancy instead:
//   * 4 waves per workgroup, 2 (M) x 2 (N), each wave a 128 x 64 output block = the ping-pong kernel's per-wave shape
//     (128 accumulator VGPRs, fragments of one 64-row half at a time: 64 VGPRs);  __launch_bounds__(256, 2);
//   * 80 KiB of LDS per workgroup (two of them fill the CU's 160 KiB exactly): a ring of FIVE 16-KiB slots over the stream of
//     16-KiB pieces  P(3t) = B rows of K-tile t,  P(3t+1) = A rows {128 wm + [0,64)},  P(3t+2) = A rows {128 wm + [64,128)}.
//     BK = 64 keeps every LDS-DMA instruction on full 128-byte lines (8 rows x 128 B, the ping-pong kernel's image and swizzle);
//   * two phases per K-tile, ONE barrier each:
//        phase (t,0):  vmcnt -> barrier b(t,0) -> issue P(3t+4) -> read B(t), A(t, mi=0) -> 32 MFMA
//        phase (t,1):  vmcnt -> barrier b(t,1) -> issue P(3t+5), P(3t+6) -> read A(t, mi=1) -> 32 MFMA
//     While one workgroup's waves wait at a barrier, read fragments or store an epilogue, the co-resident workgroup's waves
//     (one per SIMD each) own the matrix pipe.
// Hazards.  RAW: a wave waits for its own LDS-DMA instructions of the pieces the next phase reads (counted vmcnt: VMEM
//   returns in order; pieces issued later stay in flight), THEN joins the barrier, THEN reads: every wave's parts have landed.
//   WAR: P(3t+4) goes to the slot of P(3t-1), read in phase (t-1,1); P(3t+5), P(3t+6) to the slots of P(3t), P(3t+1), read in
//   phase (t,0).  Every wave finishes a phase's ds_reads (lgkmcnt(0) before its MFMAs) before it reaches the next barrier, and
//   the refill is issued after that barrier.
// DMA bytes per FLOP are 1.5x the 256x256 tile's (48 KiB per 256x128x64 vs 64 KiB per 256x256x64), which is why the ping-pong
// kernel stays in use where the epilogue is light (tools/gemm_bench.py decides per shape; bsclip_gemm_set_tile(5) forces this one).
template <int EPI, bool HAS_BIAS, bool DIAG = false>
__global__ __launch_bounds__(256, 2) void gemm_nt_duo_kernel(const bf16_t* __restrict__ A, int lda,
                                                             const bf16_t* __restrict__ B, int ldb, void* __restrict__ C,
                                                             int ldc, int M, int N, int K, int tiles_n, EpiArgs e) {
    if constexpr (epi_is_resid(EPI)) BSCLIP_DROP_RESOLVE(e.drop);
    constexpr int PIECE = 16384, NSLOT = 5;
    __shared__ __attribute__((aligned(16))) char smem[NSLOT * PIECE];

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = wg % tiles_n, tile_m = wg / tiles_n;
    const int m0 = tile_m * 256, n0 = tile_n * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    auto stamp = [&](int i) {  // diagnostic build only (tools/gemm_duo_phases.py): 100 MHz wall clock per workgroup
        if constexpr (DIAG) {
            if (tid == 0) e.diag[(size_t)blockIdx.x * 8 + i] = wall_clock64();
        }
    };
    if constexpr (DIAG) {
        if (tid == 0) {
            e.diag[(size_t)blockIdx.x * 8 + 4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
            e.diag[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
        }
    }
    stamp(0);

    // ---- LDS-DMA sources.  Instruction q = wave + 4 i of a piece covers its LDS rows [8q, 8q + 8); a lane moves 16 bytes of
    // row 8q + (lane >> 3).  The 16-byte chunk is XOR-swizzled with (row >> 1) & 7 = (4 (wave & 1) + (lane >> 4)) & 7. ----
    const int lrow = lane >> 3;
    const int chunk = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
    unsigned offA[2][4], offB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int trow = 128 * (i >> 1) + 64 * mi + 32 * (i & 1) + 8 * wave + lrow;  // LDS row 8 q + lrow <-> wm = i >> 1
            offA[mi][i] = (unsigned)min(m0 + trow, M - 1) * (unsigned)(lda * 2) + chunk * 16;
        }
        offB[i] = (unsigned)min(n0 + 8 * (wave + 4 * i) + lrow, N - 1) * (unsigned)(ldb * 2) + chunk * 16;
    }
    const char* Ab = reinterpret_cast<const char*>(A);
    const char* Bb = reinterpret_cast<const char*>(B);
    auto dmaA = [&](int mi, int slot, int kb) {
        char* d = smem + slot * PIECE + wave * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(Ab + kb + offA[mi][i], d + i * 4096);
    };
    auto dmaB = [&](int slot, int kb) {
        char* d = smem + slot * PIECE + wave * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(Bb + kb + offB[i], d + i * 4096);
    };

    // ---- fragment read offsets within a piece ----
    const int fr = lane & 15, fq = lane >> 4;
    const int a_off = (64 * wm + fr) * ROW_BYTES + ((fq ^ (fr >> 1)) << 4);
    const int b_off = (64 * wn + fr) * ROW_BYTES + ((fq ^ (fr >> 1)) << 4);

    f32x4 acc[2][4][4];  // [m half][row tile][col tile]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[a][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[2][4], fb[2][4];  // [ks][tile]

#define DUO_BARRIER()                       \
    do {                                    \
        __builtin_amdgcn_sched_barrier(0);  \
        __builtin_amdgcn_s_barrier();       \
        __builtin_amdgcn_sched_barrier(0);  \
    } while (0)
    // fragments of one 32-deep k-step (ks) at a time, so the MFMAs of ks = 0 start while the reads of ks = 1 are in flight
    auto readAB = [&](int slotA, int slotB, int ks, bool withB) {
        const char* pa = smem + slotA * PIECE;
        const char* pb = smem + slotB * PIECE;
        if (withB) {
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[ks][j] = *reinterpret_cast<const bf16x8*>(pb + ((b_off ^ (ks << 6)) + j * 16 * ROW_BYTES));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[ks][i] = *reinterpret_cast<const bf16x8*>(pa + ((a_off ^ (ks << 6)) + i * 16 * ROW_BYTES));
    };
    auto mma = [&](int mi, int ks) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[mi][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][j], fa[ks][i], acc[mi][i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    const int nk = K / BK;
    int sB = 0, sA0 = 1, sA1 = 2;  // slots of P(3t), P(3t+1), P(3t+2)
    dmaB(0, 0);
    dmaA(0, 1, 0);
    dmaA(1, 2, 0);
    if (nk > 1) dmaB(3, ROW_BYTES);
    for (int t = 0; t < nk; ++t) {
        const bool has1 = t + 1 < nk, has2 = t + 2 < nk;
        const int kb1 = (t + 1) * ROW_BYTES, kb2 = (t + 2) * ROW_BYTES;
        // ---- phase (t,0): pieces 3t, 3t+1 landed; 3t+2 (and 3t+3) may fly ----
        if (has1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        DUO_BARRIER();
        if (t == 0) stamp(1);
        readAB(sA0, sB, 0, true);
        readAB(sA0, sB, 1, true);
        __builtin_amdgcn_sched_barrier(0);
        mma(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (has1) dmaA(0, sB + 4 >= NSLOT ? sB + 4 - NSLOT : sB + 4, kb1);  // P(3t+4) -> slot of P(3t-1); issued under the MFMAs
        __builtin_amdgcn_sched_barrier(0);
        mma(0, 1);
        // ---- phase (t,1): piece 3t+2 landed; 3t+3, 3t+4 may fly ----
        if (has1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        DUO_BARRIER();
        readAB(sA1, sB, 0, false);
        readAB(sA1, sB, 1, false);
        __builtin_amdgcn_sched_barrier(0);
        mma(1, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (has1) dmaA(1, sB, kb1);   // P(3t+5) -> slot of P(3t)
        if (has2) dmaB(sA0, kb2);     // P(3t+6) -> slot of P(3t+1)
        __builtin_amdgcn_sched_barrier(0);
        mma(1, 1);
        sB = sB + 3 >= NSLOT ? sB + 3 - NSLOT : sB + 3;
        sA0 = sA0 + 3 >= NSLOT ? sA0 + 3 - NSLOT : sA0 + 3;
        sA1 = sA1 + 3 >= NSLOT ? sA1 + 3 - NSLOT : sA1 + 3;
    }
#undef DUO_BARRIER
    stamp(2);

    // ---- epilogue -------------------------------------------------------------------------------------------------
    // bias in the accumulator layout: lane owns columns n0 + 64 wn + 16 j + 4 fq + [0, 4)
    if constexpr (HAS_BIAS) {
        f32x4 bias[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bias[j] = *reinterpret_cast<const f32x4*>(e.bias + n0 + 64 * wn + 16 * j + fq * 4);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[a][i][j] += bias[j];
    }
    // LDS-staged, row-coalesced stores (see the ping-pong kernel): per m half a slab of 128 rows (64 of each wm) x 128 columns.
    // Slab row r <-> output row m0 + 128 (r >> 6) + 64 mi + (r & 63).
    constexpr int SB = 272;   // bf16 slab row stride (128 * 2 + 16)
    constexpr int SF = 528;   // f32 slab row stride (128 * 4 + 16)
    constexpr int S8 = 144;   // 8-bit slab row stride (128 + 16)
    constexpr bool GELU = EPI == BSCLIP_EPI_GELU_BF16;
    __syncthreads();  // every wave is past its last fragment read; no LDS-DMA is outstanding (vmcnt(0) above)
    if constexpr (EPI == BSCLIP_EPI_BF16 || GELU) {
        char* slab = smem;                       // 128 x 272 = 34 816
        char* slab2 = smem + 128 * SB;           // 128 x 144 = 18 432 (gelu' codes)
        const char* lut = smem + 128 * SB + 128 * S8;
        if constexpr (GELU) {
            for (int i = tid; i <= GELU_LUT_N; i += 256)
                *reinterpret_cast<float2*>(smem + 128 * SB + 128 * S8 + i * 8) = g_gelu_lut[i];
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            if (mi == 1 || GELU) BSCLIP_LDS_BARRIER();  // slab free again (mi = 1) / table visible (mi = 0)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v = acc[mi][i][j];
                    const int r = 64 * wm + 16 * i + fr, cn = 64 * wn + 16 * j + 4 * fq;
                    uint2 o;
                    if constexpr (GELU) {
                        f32x2 gl0, dg0, gl1, dg1;
                        gelu_lut2(lut, f32x2{v[0], v[1]}, gl0, dg0);
                        gelu_lut2(lut, f32x2{v[2], v[3]}, gl1, dg1);
                        *reinterpret_cast<unsigned*>(slab2 + r * S8 + cn) = dg8_pack4(dg0, dg1);
                        o.x = pack_bf2(gl0[0], gl0[1]);
                        o.y = pack_bf2(gl1[0], gl1[1]);
                    } else {
                        o.x = pack_bf2(v[0], v[1]);
                        o.y = pack_bf2(v[2], v[3]);
                    }
                    *reinterpret_cast<uint2*>(slab + r * SB + cn * 2) = o;
                }
            BSCLIP_LDS_BARRIER();
            // bf16 rows: 256 B = 16 lanes x 16 B, 16 rows per pass of the 256 threads
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int r = it * 16 + (tid >> 4);
                const int m = m0 + 128 * (r >> 6) + 64 * mi + (r & 63);
                const uint4 v = *reinterpret_cast<const uint4*>(slab + r * SB + (tid & 15) * 16);
                if (m < M) nt_store(static_cast<bf16_t*>(C) + (size_t)m * ldc + n0 + (tid & 15) * 8, v);
            }
            if constexpr (GELU) {
                if (e.aux) {  // 8-bit rows: 128 B = 8 lanes x 16 B, 32 rows per pass
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int r = it * 32 + (tid >> 3);
                        const int m = m0 + 128 * (r >> 6) + 64 * mi + (r & 63);
                        const uint4 v = *reinterpret_cast<const uint4*>(slab2 + r * S8 + (tid & 7) * 16);
                        if (m < M) nt_store(e.aux + (size_t)m * e.ld_aux + n0 + (tid & 7) * 16, v);
                    }
                }
            }
        }
    } else {
        // f32-staged epilogues.  The second operand (residual rows / gelu' codes) does not depend on the accumulators: it is
        // loaded before the slab is staged and has the LDS round trip to land.  Row walk: 512 B = 32 lanes x 16 B, 8 rows per pass.
        char* slab = smem;  // 128 x 528 = 67 584
        f32x4 pre[2][16];
        auto prefetch = [&](int mi, f32x4 (&R)[16]) {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int r = it * 8 + (tid >> 5);
                const int m = min(m0 + 128 * (r >> 6) + 64 * mi + (r & 63), M - 1);
                const int n = n0 + (tid & 31) * 4;
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
        prefetch(0, pre[0]);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            if (mi == 1) BSCLIP_LDS_BARRIER();
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<f32x4*>(slab + (64 * wm + 16 * i + fr) * SF + (64 * wn + 16 * j + 4 * fq) * 4) = acc[mi][i][j];
            BSCLIP_LDS_BARRIER();
            if (mi == 0) prefetch(1, pre[1]);  // the second half's rows fly while the first half is stored
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int r = it * 8 + (tid >> 5);
                const int m = m0 + 128 * (r >> 6) + 64 * mi + (r & 63);
                const int n = n0 + (tid & 31) * 4;
                f32x4 v = *reinterpret_cast<const f32x4*>(slab + r * SF + (tid & 31) * 16);
                if (m < M) {
                    if constexpr (EPI == BSCLIP_EPI_F32) {
                        nt_store(static_cast<float*>(C) + (size_t)m * ldc + n, v);
                    } else if constexpr (EPI == BSCLIP_EPI_RESID_F32) {
                        if (e.drop.thr16) v = drop4(e.drop, (unsigned)m * (unsigned)e.n_total + (unsigned)n, v);
                        v += pre[mi][it];
                        nt_store(static_cast<float*>(C) + (size_t)m * ldc + n, v);
                    } else if constexpr (EPI == BSCLIP_EPI_DGELU_BF16) {
                        v *= pre[mi][it];
                        uint2 o;
                        o.x = pack_bf2(v[0], v[1]);
                        o.y = pack_bf2(v[2], v[3]);
                        nt_store(static_cast<bf16_t*>(C) + (size_t)m * ldc + n, o);
                    } else if constexpr (EPI == BSCLIP_EPI_PATCH_F32) {
                        const int b = m / 196, p = m - b * 196;
                        v += pre[mi][it];
                        nt_store(static_cast<float*>(C) + (size_t)(b * 197 + 1 + p) * ldc + n, v);
                    } else if constexpr (EPI == BSCLIP_EPI_RESID_BF16) {
                        if (e.drop.thr16) v = drop4(e.drop, (unsigned)m * (unsigned)e.n_total + (unsigned)n, v);
                        v += pre[mi][it];
                        nt_store(static_cast<bf16_t*>(C) + (size_t)m * ldc + n, f32_to_bf4(v));
                    } else if constexpr (EPI == BSCLIP_EPI_PATCH_BF16) {
                        const int b = m / 196, p = m - b * 196;
                        v += pre[mi][it];
                        nt_store(static_cast<bf16_t*>(C) + (size_t)(b * 197 + 1 + p) * ldc + n, f32_to_bf4(v));
                    }
                }
            }
        }
    }
    stamp(3);
}

template <int EPI, bool HB>
void launch_duo(const bf16_t* A, int lda, const bf16_t* B, int ldb, void* C, int ldc, int M, int N, int K, const EpiArgs& e,
                hipStream_t s) {
    const int tiles_m = ceil_div(M, 256), tiles_n = N / 128;
    hipLaunchKernelGGL((gemm_nt_duo_kernel<EPI, HB>), dim3(tiles_m * tiles_n), dim3(256), 0, s, A, lda, B, ldb, C, ldc, M, N, K,
                       tiles_n, e);
}
