// Exact inner-product top-k retrieval (gfx950) -- SURVEY.md 8f rank 1.
//
// Replaces faiss `IndexFlatIP(768).add(keys); .search(queries, max_k)` in `make_prediction`
// (reference scripts/inference_and_eval.py:414-445), including the sklearn `normalize(..., norm="l2")` of both sides
// (:416-417).  IndexFlatIP is brute force, so the restatement is exact: scores = Qn Kn^T, top max_k per query, ordered
// by descending score (ties: lower key index first).
//   * scores on the bf16 MFMA GEMM with f32-accurate split operands (x = hi + lo in bf16; hi.hi + hi.lo + lo.hi + lo.lo
//     along K = 4 D -- all four terms, so that a vector scores 1 against itself to f32 rounding), in slabs of <= 1024
//     queries so the [Q x K] matrix never exists in HBM;
//   * selection: one 64-lane wave per query row; every lane keeps a sorted top-k of its strided share in registers,
//     then k rounds of a wavefront arg-max (shuffles) merge the 64 lists.  HBM-bound: one pass over the score slab.
#include <math.h>

#include "common.h"

namespace {

constexpr int TOPK_SLAB = 1024;

// One wave per row i in [0, Np): x = z_i / ||z_i|| (zero rows and rows >= N give zeros), split x = hi + lo in bf16,
// P[i] = [lo|lo|hi|hi] (query side) or [lo|hi|lo|hi] (key side), bf16 [Np, 4 D].
template <bool KEY_SIDE>
__global__ __launch_bounds__(256) void normalize_split4_kernel(const float* __restrict__ z, int N, int Np, int D,
                                                                bf16_t* __restrict__ P) {
    const int lane = threadIdx.x & 63;
    const int row = (blockIdx.x * 256 + threadIdx.x) >> 6;
    if (row >= Np) return;
    float scale = 0.f;
    if (row < N) {
        float ss = 0.f;
        for (int c = lane * 4; c < D; c += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(z + (size_t)row * D + c);
            ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
        ss = wave_sum(ss);
        scale = ss > 0.f ? 1.0f / sqrtf(ss) : 0.f;  // sklearn normalize leaves zero rows at zero
    }
    for (int c = lane * 4; c < D; c += 256) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < N) v = *reinterpret_cast<const f32x4*>(z + (size_t)row * D + c) * scale;
        unsigned h[4], l[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16_t hi = f2bf(v[i]);
            h[i] = hi;
            l[i] = f2bf(v[i] - bf2f(hi));
        }
        const uint2 hh = {h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
        const uint2 ll = {l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
        bf16_t* p = P + (size_t)row * 4 * D + c;
        // small terms first along K, so the f32 accumulator only becomes large for the last D columns (hi.hi)
        *reinterpret_cast<uint2*>(p) = ll;
        *reinterpret_cast<uint2*>(p + D) = KEY_SIDE ? hh : ll;
        *reinterpret_cast<uint2*>(p + 2 * D) = KEY_SIDE ? ll : hh;
        *reinterpret_cast<uint2*>(p + 3 * D) = hh;
    }
}

constexpr int TOPK_CAND = 256;  // candidate slots per row in LDS before the exact fallback scan takes over

// descending (value, then lower index first) insertion into a per-lane sorted list held in registers
template <int MAXK>
__device__ __forceinline__ void topk_insert(float (&v)[MAXK], int (&ix)[MAXK], float x, int xi) {
#pragma unroll
    for (int j = 0; j < MAXK; ++j) {
        const bool better = x > v[j] || (x == v[j] && xi < ix[j]);
        const float tv = better ? v[j] : x;
        const int ti = better ? ix[j] : xi;
        v[j] = better ? x : v[j];
        ix[j] = better ? xi : ix[j];
        x = tv;
        xi = ti;
    }
}

// One wave per query row.  Pass 1: every lane takes the maximum of its strided share (16-B loads, 4 in flight); the
// k-th largest of the 64 lane maxima is a threshold T with at least k elements >= T, so the row's top k all pass it.
// Pass 2: re-scan (the slab is L2 / Infinity-Cache resident) and compact the few elements >= T into LDS.  Pass 3: exact
// top-k of the candidates -- per-lane sorted lists in registers, then k rounds of a wavefront arg-max.  A row with more
// than TOPK_CAND candidates (many equal scores) falls back to inserting every element, which is exact for any input.
template <int MAXK>
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ scores, int ld, int nrows, int K, int k,
                                                         float* __restrict__ out_s, int64_t* __restrict__ out_i,
                                                         int out_ld) {
    __shared__ float cand_v[4][TOPK_CAND];
    __shared__ int cand_i[4][TOPK_CAND];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + w;
    if (r >= nrows) return;  // whole waves leave; no block-wide barrier below
    const float* row = scores + (size_t)r * ld;
    const int K4 = (K + 3) & ~3;  // ld is a multiple of 128, so the last 16-B chunk is readable; columns >= K are masked

    float m = -INFINITY;
    for (int c0 = lane * 4; c0 < K4; c0 += 1024) {
        f32x4 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * 256;
            x[u] = c < K4 ? *reinterpret_cast<const f32x4*>(row + c) : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c0 + u * 256 + e < K) m = fmaxf(m, x[u][e]);
    }
    // T = k-th largest lane maximum (rank by value, ties by lane)
    int rank = 0;
    for (int j = 0; j < 64; ++j) {
        const float o = __shfl(m, j, 64);
        rank += (o > m || (o == m && j < lane)) ? 1 : 0;
    }
    const int kk = k < 64 ? k : 64;
    const unsigned long long sel = __ballot(rank == kk - 1);
    const float T = __shfl(m, __ffsll((long long)sel) - 1, 64);

    int count = 0;  // wave-uniform: every lane runs every trip and takes part in every ballot
    for (int base = 0; base < K4 && count <= TOPK_CAND; base += 256) {
        const int c0 = base + lane * 4;
        f32x4 x = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (c0 < K4) x = *reinterpret_cast<const f32x4*>(row + c0);
        bool hit = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) hit |= (c0 + e < K) && x[e] >= T;
        if (__ballot(hit) == 0ull) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool h = (c0 + e < K) && x[e] >= T;
            const unsigned long long b = __ballot(h);
            const int pos = count + __popcll(b & ((1ull << lane) - 1ull));
            if (h && pos < TOPK_CAND) {
                cand_v[w][pos] = x[e];
                cand_i[w][pos] = c0 + e;
            }
            count += __popcll(b);
        }
    }
    float v[MAXK];
    int ix[MAXK];
#pragma unroll
    for (int j = 0; j < MAXK; ++j) {
        v[j] = -INFINITY;
        ix[j] = 0x7fffffff;
    }
    if (count <= TOPK_CAND) {
        __builtin_amdgcn_wave_barrier();
        for (int c = lane; c < count; c += 64) topk_insert<MAXK>(v, ix, cand_v[w][c], cand_i[w][c]);
    } else {
        for (int c = lane; c < K; c += 64) topk_insert<MAXK>(v, ix, row[c], c);
    }
    for (int o = 0; o < k; ++o) {
        float bv = v[0];
        int bi = ix[0];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            const bool take = ov > bv || (ov == bv && oi < bi);
            bv = take ? ov : bv;
            bi = take ? oi : bi;
        }
        if (lane == 0) {
            out_s[(size_t)r * out_ld + o] = bv;
            out_i[(size_t)r * out_ld + o] = bi;
        }
        if (ix[0] == bi) {  // the winning lane pops its head
#pragma unroll
            for (int j = 0; j + 1 < MAXK; ++j) {
                v[j] = v[j + 1];
                ix[j] = ix[j + 1];
            }
            v[MAXK - 1] = -INFINITY;
            ix[MAXK - 1] = 0x7fffffff;
        }
    }
}

inline int64_t al4(int64_t x) { return (x + 3) & ~(int64_t)3; }
inline int pad_keys(int n) { return (n + 255) / 256 * 256; }  // multiple of 256: the 256x256 ping-pong GEMM tile

}  // namespace

extern "C" int64_t bsclip_topk_ip_workspace_floats(int Q, int K, int D) {
    if (Q <= 0 || K <= 0 || D <= 0) return -1;
    const int64_t Kp = pad_keys(K), Qs = Q < TOPK_SLAB ? Q : TOPK_SLAB;
    // key operand bf16 [Kp, 4D] + query operand bf16 [Qs, 4D] + one f32 score slab [Qs, Kp]
    return al4(Kp * 2 * D) + al4(Qs * 2 * D) + al4(Qs * Kp);
}

extern "C" int bsclip_topk_ip(const float* queries, int Q, const float* keys, int K, int D, int k, float* scores_out,
                              int64_t* idx_out, float* workspace, void* stream) {
    BSCLIP_REQUIRE(queries && keys && scores_out && idx_out && workspace, "bsclip_topk_ip: null pointer");
    BSCLIP_REQUIRE(Q > 0 && K > 0 && D > 0 && D % 64 == 0, "bsclip_topk_ip: Q=%d K=%d D=%d (D %% 64 == 0)", Q, K, D);
    BSCLIP_REQUIRE(k >= 1 && k <= 16 && k <= K, "bsclip_topk_ip: k=%d (1..16, <= K)", k);
    BSCLIP_REQUIRE((((uintptr_t)workspace) & 15) == 0, "bsclip_topk_ip: workspace must be 16-B aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int Kp = pad_keys(K);
    const int Qs = Q < TOPK_SLAB ? Q : TOPK_SLAB;
    float* ws = workspace;
    bf16_t* kP = reinterpret_cast<bf16_t*>(ws); ws += al4((int64_t)Kp * 2 * D);
    bf16_t* qP = reinterpret_cast<bf16_t*>(ws); ws += al4((int64_t)Qs * 2 * D);
    float* sc = ws;
    hipLaunchKernelGGL((normalize_split4_kernel<true>), dim3(ceil_div(Kp, 4)), dim3(256), 0, s, keys, K, Kp, D, kP);
    for (int q0 = 0; q0 < Q; q0 += TOPK_SLAB) {
        const int nq = Q - q0 < TOPK_SLAB ? Q - q0 : TOPK_SLAB;
        hipLaunchKernelGGL((normalize_split4_kernel<false>), dim3(ceil_div(nq, 4)), dim3(256), 0, s,
                           queries + (size_t)q0 * D, nq, nq, D, qP);
        const int rc = bsclip_gemm_bf16(qP, 4 * D, kP, 4 * D, sc, Kp, nq, Kp, 4 * D, BSCLIP_EPI_F32, nullptr, stream);
        if (rc) return rc;
        float* so = scores_out + (size_t)q0 * k;
        int64_t* io = idx_out + (size_t)q0 * k;
        if (k <= 8)
            hipLaunchKernelGGL((topk_rows_kernel<8>), dim3(ceil_div(nq, 4)), dim3(256), 0, s, sc, Kp, nq, K, k, so, io, k);
        else
            hipLaunchKernelGGL((topk_rows_kernel<16>), dim3(ceil_div(nq, 4)), dim3(256), 0, s, sc, Kp, nq, K, k, so, io, k);
    }
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}
