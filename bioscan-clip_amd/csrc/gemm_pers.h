// Persistent form of the 256x256 ping-pong GEMM kernel: one workgroup per CU walks tiles v = blockIdx.x, + gridDim.x, ...
//
// Included by gemm.hip inside its anonymous namespace (shares EpiArgs, the GELU table, BK / ROW_BYTES and the epilogue helpers).
//
// Why: with K = 768 a tile is 12 K-tiles (17.6 us); tools/gemm_phases.py shows 2.5-3.2 us before its first MFMA (launch, address
// arithmetic, the first LDS-DMA round trip with nothing to hide behind) and the DMA stream idle through the epilogue.  A workgroup
// that keeps its CU simply continues the K-tile stream across the tile boundary: the pieces of the NEXT tile's K-tile 0 take the
// DMA slots of the current tile's last two K-tiles, so a tile after the first starts with its operands in LDS.
//
// K loop, phases, barriers and hazards: exactly gemm_nt_pp_kernel's production schedule (see its header).  Differences:
//   * LDS 160 KiB = set 0 [0, 64K) | spare [64K, 96K) | set 1 [96K, 160K).  K-tile t of a tile uses set (p + t) & 1, p = parity
//     of the K-tiles this workgroup has consumed before the tile.  The epilogue stages only in set E = (p + nk - 1) & 1 (the set
//     of the tile's last K-tile, dead after the loop) plus the spare; the other set already holds the next tile's K-tile 0:
//     bf16 / GELU outputs [E base, + 51 200) with the GELU table resident in the spare; f32-staged outputs in 66 560 B from
//     (E ? 64K : 0), i.e. set E and the spare next to it.
//   * hence 32-row slabs per wave group (four stage / store rounds per tile instead of two 64-row rounds).
//   * DMA sources are 32-bit byte offsets from the operand base (8 VGPRs instead of 16 pointers); they are re-aimed at the next
//     tile in phase 3 of K-tile nk - 2, after the current tile's last piece has been issued.
//   * tile boundary: phase 3 of K-tile nk - 2 issues A-half0 of the next tile's K-tile 0, phases 0-2 of K-tile nk - 1 the rest,
//     phase 3 of K-tile nk - 1 waits for all of it (vmcnt(0)); nothing is in flight during the epilogue; after the epilogue's
//     last barrier A-half0 of K-tile 1 goes into set E -- the loop invariant of the prologue ("tile 0 complete, first piece of
//     tile 1 issued").  Epilogue stores still in flight are older than every later LDS-DMA: a counted vmcnt that leaves the two
//     newest instructions outstanding has the loads before them landed whatever the stores do (loads return in order).
// Needs nk >= 2, N % 256 == 0, operands below 4 GiB.  Tile order = the ping-pong kernel's (xcd_remap of the virtual block id v).
constexpr bool pers_supported(int epi) {
    return epi == BSCLIP_EPI_BF16 || epi == BSCLIP_EPI_F32 || epi == BSCLIP_EPI_GELU_BF16 || epi == BSCLIP_EPI_RESID_F32 ||
           epi == BSCLIP_EPI_DGELU_BF16 || epi == BSCLIP_EPI_RESID_BF16;
}

typedef unsigned pers_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned pers_u32x2 __attribute__((ext_vector_type(2)));
template <bool NT, class T>
__device__ __forceinline__ void pers_store(void* p, T v, bool skip = false) {
    if (skip) return;   // diagnostic build only (tools/gemm_pers_phases.py ABL=1): is the slow start of a tile's K loop the stores?
    if constexpr (NT) __builtin_nontemporal_store(v, static_cast<T*>(p));
    else *static_cast<T*>(p) = v;
}

template <int EPI, bool HAS_BIAS, bool DIAG = false, bool NT = false>
__global__ __launch_bounds__(512) void gemm_nt_pers_kernel(const bf16_t* __restrict__ A, int lda,
                                                            const bf16_t* __restrict__ B, int ldb, void* __restrict__ C,
                                                            int ldc, int M, int N, int K, int tiles_n, int ntiles, EpiArgs e) {
    static_assert(pers_supported(EPI), "persistent kernel: epilogue not instantiated");
    const int gw = e.pers_gw & 0x3fffffff;       // column tiles per super-column, divides tiles_n (any tiles_n: the flag sits in bit 30)
    const bool abl_st = DIAG && (e.pers_gw & 0x40000000);   // diagnostic build: leave the output stores out
    const int tiles_m = ntiles / tiles_n;
    if constexpr (epi_is_resid(EPI)) BSCLIP_DROP_RESOLVE(e.drop);
    constexpr int SET1 = 98304, SPARE = 65536, HALF = 16384, B_OFF = 32768;
    constexpr bool GELU = EPI == BSCLIP_EPI_GELU_BF16;
    constexpr bool BF16_STAGED = EPI == BSCLIP_EPI_BF16 || GELU;
    __shared__ __attribute__((aligned(16))) char smem[163840];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, wc = wave & 3;
    const int nk = K / BK;
    const char* Ab = reinterpret_cast<const char*>(A);
    const char* Bb = reinterpret_cast<const char*>(B);

    // ---- LDS-DMA sources of the tile the DMA stream is aimed at: half h, chunk (wave) and (wave + 8) ----
    unsigned offA[2][2], offB[2][2];
    int dm0 = 0, dn0 = 0;
    // The offsets of the NEXT tile are formed in three pieces (tile coordinates, A rows, B rows) so that each fits the "load" half
    // of a phase -- the time the other wave group spends in its MFMAs; formed in one go in phase 3 of K-tile nk - 2 they stretched
    // that K-tile from 1.48 to 1.84 us (tools/gemm_pers_phases.py, KTILES=1).
    unsigned nxtA[2][2], nxtB[2][2];
    int nm0 = 0, nn0 = 0;
    auto plan_tile = [&](int v) {
        // logical order: "super-columns" of gw column tiles, all row panels of one before the next -- an XCD's 32 concurrent tiles
        // then share gw column tiles of B (which stays in its 4-MiB L2: fc1's B is 4.7 MB, 12 column tiles) instead of all of them
        const int wgid = xcd_remap(v, ntiles);
        const int per_sc = tiles_m * gw;
        const int sc = wgid / per_sc, rem = wgid - sc * per_sc;
        nn0 = (sc * gw + rem % gw) * 256;
        nm0 = (rem / gw) * 256;
    };
    auto plan_rows = [&](unsigned (&o)[2][2], int r0, int rmax, int ld) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = 128 * h + 8 * (wave + 8 * i) + (lane >> 3);
                const int c = (lane & 7) ^ ((row >> 1) & 7);
                o[h][i] = (unsigned)min(r0 + row, rmax) * (unsigned)(ld * 2) + c * 16;
            }
    };
    auto take_plan = [&]() {
        dm0 = nm0;
        dn0 = nn0;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                offA[h][i] = nxtA[h][i];
                offB[h][i] = nxtB[h][i];
            }
    };
    auto aim = [&](int v) {
        plan_tile(v);
        plan_rows(nxtA, nm0, M - 1, lda);
        plan_rows(nxtB, nn0, N - 1, ldb);
        take_plan();
    };
    const int dma_off = wave * 1024;
    auto dmaA = [&](int set, int h, int k0) {   // k0 = byte offset of the K-tile in the row
        char* d = smem + set * SET1 + h * HALF + dma_off;
        glds16(Ab + k0 + offA[h][0], d);
        glds16(Ab + k0 + offA[h][1], d + 8192);
    };
    auto dmaB = [&](int set, int h, int k0) {
        char* d = smem + set * SET1 + B_OFF + h * HALF + dma_off;
        glds16(Bb + k0 + offB[h][0], d);
        glds16(Bb + k0 + offB[h][1], d + 8192);
    };

    // ---- fragment read offsets ----
    const int fr = lane & 15, fq = lane >> 4;
    const int sw = fr >> 1;
    const int a_off = (128 * g + fr) * ROW_BYTES + ((fq ^ sw) << 4);
    const int b_off = B_OFF + (64 * wc + fr) * ROW_BYTES + ((fq ^ sw) << 4);

    f32x4 acc[2][2][4][2];  // [m block][n block][row tile][col tile]
    bf16x8 fa[2][2][4];     // [m block][ks][row tile]
    bf16x8 fb[2][2][2];     // [n block][ks][col tile]
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto readA = [&](const char* base, int mi) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                fa[mi][ks][i] = *reinterpret_cast<const bf16x8*>(base + ((a_off ^ (ks << 6)) + (64 * mi + 16 * i) * ROW_BYTES));
    };
    auto readB = [&](const char* base, int ni) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                fb[ni][ks][j] = *reinterpret_cast<const bf16x8*>(base + ((b_off ^ (ks << 6)) + (32 * ni + 16 * j) * ROW_BYTES));
    };
    auto mma = [&](int mi, int ni) {
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
#define PERS_BARRIER()                         \
    do {                                       \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)
    // diagnostic build, per workgroup (100 MHz ticks): {start, end, tiles done, s_memtime cycles start -> end, tile 1: K loop start, K loop end, then after each
    // of the 8 epilogue barriers (stage q / store q, q = 0..3), tile 0: K loop start, K loop end, tile 1: end of K-tile 0..15};
    // 32 slots per workgroup
    auto stamp = [&](int i) {
        if constexpr (DIAG) {
            if (tid == 0 && i < 32) e.diag[(size_t)blockIdx.x * 32 + i] = wall_clock64();
        }
    };
    stamp(0);
    [[maybe_unused]] const unsigned long long cyc0 = DIAG ? __builtin_readcyclecounter() : 0ull;

    int v = blockIdx.x;
    aim(v);
    int m0 = dm0, n0 = dn0;
    int p = 0;
    if constexpr (GELU) {   // resident table in the spare; visible after the prologue's barrier
        for (int i = tid; i <= GELU_LUT_N; i += 512) *reinterpret_cast<float2*>(smem + SPARE + i * 8) = g_gelu_lut[i];
    }
    // ---- prologue of the first tile: K-tile 0 complete, plus the first piece of K-tile 1 ----
    dmaA(0, 0, 0);
    dmaA(0, 1, 0);
    dmaB(0, 0, 0);
    dmaB(0, 1, 0);
    dmaA(1, 0, 2 * BK);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    if constexpr (GELU) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PERS_BARRIER();
    zero_acc();

    const int wq = wave & 3;
    int done = 0;
    for (;;) {
        const int vn = v + (int)gridDim.x;
        const bool hasN = vn < ntiles;   // wave-uniform
        if (g == 1) PERS_BARRIER();      // group 1 runs one barrier behind group 0 through the K loop
        stamp(done == 0 ? 14 : done == 1 ? 4 : 99);
        for (int t = 0; t < nk; ++t) {
            const int set = (p + t) & 1;
            const char* base = smem + set * SET1;
            const bool in1 = t + 1 < nk, in2 = t + 2 < nk;
            const bool do1 = in1 || hasN;
            const int k1 = in1 ? (t + 1) * 2 * BK : 0;   // past the tile's end: K-tile 0 of the next tile (offsets re-aimed)
            const bool plan = in1 && !in2 && hasN;       // K-tile nk - 2: plan the next tile's offsets, a piece per phase
            // ---- phase 0 ----
            if (do1) dmaA(set ^ 1, 1, k1);
            readA(base, 0);
            readB(base, 0);
            if (plan) {
                int vv = vn;   // the opaque copy keeps this arithmetic HERE (hoisted to the top of the tile it lives through the K loop)
                asm volatile("" : "+s"(vv));
                plan_tile(vv);
            }
            PERS_BARRIER();
            mma(0, 0);
            PERS_BARRIER();
            // ---- phase 1 ----
            if (do1) dmaB(set ^ 1, 0, k1);
            readA(base, 1);
            if (plan) plan_rows(nxtA, nm0, M - 1, lda);
            PERS_BARRIER();
            mma(1, 0);
            PERS_BARRIER();
            // ---- phase 2 ----
            if (do1) dmaB(set ^ 1, 1, k1);
            readB(base, 1);
            if (plan) plan_rows(nxtB, nn0, N - 1, ldb);
            PERS_BARRIER();
            mma(1, 1);
            PERS_BARRIER();
            // ---- phase 3 ----
            if (in2) {
                dmaA(set, 0, (t + 2) * 2 * BK);
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   // K-tile t+1 has landed; A-half0 of t+2 may fly
            } else if (plan) {
                take_plan();   // every piece of this tile has been issued: the DMA stream moves on to the next tile
                dmaA(set, 0, 0);
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            PERS_BARRIER();
            mma(0, 1);
            PERS_BARRIER();
            stamp(done == 1 ? 16 + t : 99);   // diagnostic build: end of K-tile t of the second tile
        }
        if (g == 0) PERS_BARRIER();  // balance group 1's extra barrier
        stamp(done == 0 ? 15 : done == 1 ? 5 : 99);

        // ---- epilogue of tile (m0, n0); staging in set E and the spare only ----
        const int E = (p + nk - 1) & 1;
        // lane-derived epilogue addresses must not be hoisted over the K loop (they would live beside 128 accumulator and 80
        // fragment VGPRs and spill): the epilogue derives them from an opaque copy of the lane id
        int le = lane;
        asm volatile("" : "+v"(le));
        const int fre = le & 15, fqe = le >> 4;
        {
            f32x4 bias[2][2];
            if constexpr (HAS_BIAS) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        bias[ni][j] = *reinterpret_cast<const f32x4*>(e.bias + n0 + 64 * wc + 32 * ni + 16 * j + fqe * 4);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) acc[mi][ni][i][j] += bias[ni][j];
            }
        }
        constexpr int SB = 528, SF = 1040, S8 = 272;   // slab row strides: bf16, f32, 8-bit
        if constexpr (BF16_STAGED) {
            char* stg = smem + (E ? SET1 : 0);
            char* slab = stg + g * (32 * SB);
            char* slab2 = stg + 2 * 32 * SB + g * (32 * S8);   // GELU: gelu' side band, 8-bit codes
            const char* lut = smem + SPARE;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int mi = q >> 1, ih = q & 1;
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2) {
                    f32x4 x[4], gl[4], dg[4];   // [2 ni + j]
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            x[2 * ni + j] = acc[mi][ni][2 * ih + i2][j];
                            acc[mi][ni][2 * ih + i2][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                    // two batches of eight values: sixteen at once spill (2 VGPRs, reloaded behind a vmcnt(0)).  Tried: the gathers of
                    // batch b + 1 issued before the arithmetic of batch b (software pipeline over the slab's four batches) -- stage
                    // time unchanged (2.1 us per slab): at the measured 2.0 GHz the slab's ~3 000 VALU cycles per SIMD are 1.5 us of it
                    if constexpr (GELU) {
                        f32x4 (&x2)[2][2] = reinterpret_cast<f32x4 (&)[2][2]>(x);
                        f32x4 (&gl2)[2][2] = reinterpret_cast<f32x4 (&)[2][2]>(gl);
                        f32x4 (&dg2)[2][2] = reinterpret_cast<f32x4 (&)[2][2]>(dg);
                        gelu_lut_batch<2>(lut, x2[0], gl2[0], dg2[0]);
                        gelu_lut_batch<2>(lut, x2[1], gl2[1], dg2[1]);
                    }
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int col = 64 * wc + 32 * ni + 16 * j + 4 * fqe;
                            const int k = 2 * ni + j;
                            uint2 o;
                            if constexpr (GELU) {
                                *reinterpret_cast<unsigned*>(slab2 + (16 * i2 + fre) * S8 + col) =
                                    dg8_pack4(dg8_f32x2{dg[k][0], dg[k][1]}, dg8_f32x2{dg[k][2], dg[k][3]});
                                o.x = pack_bf2(gl[k][0], gl[k][1]);
                                o.y = pack_bf2(gl[k][2], gl[k][3]);
                            } else {
                                o.x = pack_bf2(x[k][0], x[k][1]);
                                o.y = pack_bf2(x[k][2], x[k][3]);
                            }
                            *reinterpret_cast<uint2*>(slab + (16 * i2 + fre) * SB + col * 2) = o;
                        }
                }
                BSCLIP_LDS_BARRIER();
                stamp(done == 1 ? 6 + 2 * q : 99);
#pragma unroll
                for (int it = 0; it < 4; ++it) {   // 32 rows x 512 B: 32 lanes per row, 8 rows per pass of the group's 256 threads
                    const int r = it * 8 + wq * 2 + (le >> 5);
                    const int m = m0 + 128 * g + 32 * q + r;
                    const pers_u32x4 w = *reinterpret_cast<const pers_u32x4*>(slab + r * SB + (le & 31) * 16);
                    if (m < M) pers_store<NT>(static_cast<bf16_t*>(C) + (size_t)m * ldc + n0 + (le & 31) * 8, w, abl_st);
                }
                if constexpr (GELU) {
                    if (e.aux) {
                        const int t8 = wq * 64 + le;
#pragma unroll
                        for (int it = 0; it < 2; ++it) {   // 32 rows x 256 B: 16 lanes per row
                            const int r = it * 16 + (t8 >> 4);
                            const int m = m0 + 128 * g + 32 * q + r;
                            const pers_u32x4 w = *reinterpret_cast<const pers_u32x4*>(slab2 + r * S8 + (t8 & 15) * 16);
                            if (m < M) pers_store<NT>(e.aux + (size_t)m * e.ld_aux + n0 + (t8 & 15) * 16, w, abl_st);
                        }
                    }
                }
                if (q < 3 || hasN) BSCLIP_LDS_BARRIER();
                stamp(done == 1 ? 7 + 2 * q : 99);
            }
        } else {
            // f32-staged epilogues read a second operand (residual stream / saved gelu') row-wise: issued one slab ahead
            char* slab = smem + (E ? SPARE : 0) + g * (32 * SF);
            // Raw (unconverted) words, so that nothing waits for a load where it is issued.  bf16 residual rows (2 VGPRs per piece)
            // and 8-bit gelu' codes (1 VGPR) are fetched for ALL FOUR slabs of the tile up front -- the fragment registers are dead
            // here -- so only the first slab's consume can see HBM latency; f32 residual rows (4 VGPRs) stay one slab ahead.
            constexpr bool RAW2 = EPI == BSCLIP_EPI_RESID_BF16, RAW1 = EPI == BSCLIP_EPI_DGELU_BF16;
            constexpr int DEPTH = (RAW2 || RAW1) ? 4 : 2;
            using raw_t = std::conditional_t<RAW2, uint2, std::conditional_t<RAW1, unsigned, f32x4>>;
            raw_t pre[DEPTH][8];
            auto prefetch = [&](int q, raw_t (&R)[8]) {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int m = min(m0 + 128 * g + 32 * q + it * 4 + wq, M - 1);
                    const int n = n0 + le * 4;
                    if constexpr (EPI == BSCLIP_EPI_RESID_F32) {
                        R[it] = *reinterpret_cast<const f32x4*>(e.resid + (size_t)m * e.ld_resid + n);
                    } else if constexpr (RAW2) {
                        R[it] = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(e.resid) + (size_t)m * e.ld_resid + n);
                    } else if constexpr (RAW1) {
                        R[it] = *reinterpret_cast<const unsigned*>(e.aux + (size_t)m * e.ld_aux + n);
                    }
                }
            };
            auto cooked = [&](const raw_t& r) -> f32x4 {
                if constexpr (RAW2) return bf4_to_f32(r);
                else if constexpr (RAW1) return dg8_unpack4(r);
                else return r;
            };
            auto stage = [&](int q) {
                const int mi = q >> 1, ih = q & 1;
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            *reinterpret_cast<f32x4*>(slab + (16 * i2 + fre) * SF + (64 * wc + 32 * ni + 16 * j + 4 * fqe) * 4) =
                                acc[mi][ni][2 * ih + i2][j];
                            acc[mi][ni][2 * ih + i2][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
            };
            auto consume = [&](int q, const raw_t (&R)[8]) {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int r = it * 4 + wq;  // one 1-KiB row per wave instruction
                    const int m = m0 + 128 * g + 32 * q + r;
                    const int n = n0 + le * 4;
                    f32x4 x = *reinterpret_cast<const f32x4*>(slab + r * SF + le * 16);
                    if (m < M) {
                        if constexpr (EPI == BSCLIP_EPI_F32) {
                            pers_store<NT>(static_cast<float*>(C) + (size_t)m * ldc + n, x, abl_st);
                        } else if constexpr (EPI == BSCLIP_EPI_RESID_F32) {
                            if (e.drop.thr16) x = drop4(e.drop, (unsigned)m * (unsigned)e.n_total + (unsigned)n, x);
                            x += cooked(R[it]);
                            pers_store<NT>(static_cast<float*>(C) + (size_t)m * ldc + n, x, abl_st);
                        } else if constexpr (EPI == BSCLIP_EPI_DGELU_BF16) {
                            x *= cooked(R[it]);
                            {
                                const uint2 o = f32_to_bf4(x);
                                pers_store<NT>(static_cast<bf16_t*>(C) + (size_t)m * ldc + n, pers_u32x2{o.x, o.y}, abl_st);
                            }
                        } else if constexpr (EPI == BSCLIP_EPI_RESID_BF16) {
                            if (e.drop.thr16) x = drop4(e.drop, (unsigned)m * (unsigned)e.n_total + (unsigned)n, x);
                            x += cooked(R[it]);
                            {
                                const uint2 o = f32_to_bf4(x);
                                pers_store<NT>(static_cast<bf16_t*>(C) + (size_t)m * ldc + n, pers_u32x2{o.x, o.y}, abl_st);
                            }
                        }
                    }
                }
            };
            if constexpr (EPI != BSCLIP_EPI_F32) {
#pragma unroll
                for (int q = 0; q < DEPTH - 1; ++q) prefetch(q, pre[q]);
            }
            stage(0);
            BSCLIP_LDS_BARRIER();
            stamp(done == 1 ? 6 : 99);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (EPI != BSCLIP_EPI_F32) {
                    if (q + DEPTH - 1 < 4) prefetch(q + DEPTH - 1, pre[(q + DEPTH - 1) % DEPTH]);
                }
                consume(q, pre[q % DEPTH]);
                if (q < 3 || hasN) BSCLIP_LDS_BARRIER();
                stamp(done == 1 ? 7 + 2 * q : 99);
                if (q < 3) {
                    stage(q + 1);
                    BSCLIP_LDS_BARRIER();
                    stamp(done == 1 ? 8 + 2 * q : 99);
                }
            }
        }
        ++done;
        if (!hasN) break;
        // ---- the next tile: its K-tile 0 is in set p' = E ^ 1 (waited for in the last phase 3, barriers since) ----
        p = (p + nk) & 1;
        v = vn;
        m0 = dm0;
        n0 = dn0;
        dmaA(p ^ 1, 0, 2 * BK);
    }
#undef PERS_BARRIER
    stamp(1);
    if constexpr (DIAG) {
        if (tid == 0) {
            e.diag[(size_t)blockIdx.x * 32 + 2] = done;
            e.diag[(size_t)blockIdx.x * 32 + 3] = __builtin_readcyclecounter() - cyc0;   // shader-clock cycles of this workgroup's life
        }
    }
}

// non-temporal output stores (default): a tile's 128-192 KB of output no longer displaces operand lines in L2 -- dfc2 269 -> 238 us,
// fc1 ~ -3 %, the step 40.0 -> 39.4 ms (profiles/r03_i_gemm_pers.log); BSCLIP_GEMM_NT=0 is the A/B switch
const bool g_pers_nt = !(getenv("BSCLIP_GEMM_NT") && atoi(getenv("BSCLIP_GEMM_NT")) == 0);
const int g_pers_gw = getenv("BSCLIP_GEMM_GW") ? atoi(getenv("BSCLIP_GEMM_GW")) : 0;   // experiment: super-column width
int g_pers_grid = getenv("BSCLIP_GEMM_WGS") ? atoi(getenv("BSCLIP_GEMM_WGS")) : 0;   // bsclip_gemm_set_persistent_grid: workgroups of the persistent launch, 0 = one per CU

template <int EPI, bool HB>
void launch_pers(const bf16_t* A, int lda, const bf16_t* B, int ldb, void* C, int ldc, int M, int N, int K, const EpiArgs& e_in,
                 hipStream_t s) {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (cus <= 0) cus = 256;
    }
    const int tiles_m = ceil_div(M, 256), tiles_n = N / 256, nt = tiles_m * tiles_n;
    const int wgs = g_pers_grid > 0 ? g_pers_grid : cus;
    EpiArgs e = e_in;
    e.pers_gw = tiles_n;
    {
        const int want = g_pers_gw > 0 ? g_pers_gw : 6;
        if (tiles_n > want)
            for (int d = want; d >= 2; --d)
                if (tiles_n % d == 0) {
                    e.pers_gw = d;
                    break;
                }
    }
    if (g_pers_nt)
        hipLaunchKernelGGL((gemm_nt_pers_kernel<EPI, HB, false, true>), dim3(nt < wgs ? nt : wgs), dim3(512), 0, s, A, lda, B, ldb, C,
                           ldc, M, N, K, tiles_n, nt, e);
    else
        hipLaunchKernelGGL((gemm_nt_pers_kernel<EPI, HB>), dim3(nt < wgs ? nt : wgs), dim3(512), 0, s, A, lda, B, ldb, C, ldc, M, N,
                           K, tiles_n, nt, e);
}
