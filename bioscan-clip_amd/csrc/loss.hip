// Multi-pair soft-target InfoNCE (gfx950): ContrastiveLoss.forward, bioscanclip/model/loss_func.py:29-54, with
// construct_label_metrix (:18-21), forward + backward in one call.
//
// loss = 1/(P N) * sum over directed pairs (a,b), a != b, of  sum_i [ cnt_i * LSE_j(s z^a_i . z^b_j) - sum_j T_ij s z^a_i . z^b_j ]
// (every directed matrix appears twice in the reference's list, which leaves the mean unchanged).
//
// The temperature (s = 1/0.07) amplifies logit error 14x, so the N x N products must be f32-accurate, but they
// should still run on the bf16 MFMA GEMM.  Each f32 operand x is split x = hi + lo (two bf16) and the product is
// formed as hi.hi + hi.lo + lo.hi by concatenating along K ([hi|hi|lo] x [hi|lo|hi]^T): one MFMA product with K = 3*768,
// relative error ~2^-16.  The second F.normalize (loss_func.py:43-44) and its Jacobian are applied here.
//
// Fused form (default; north_star's "fused logits kernel"): the logits are never written to memory.
//   pass 1, one launch per directed pair (a,b): the 256x256 ping-pong GEMM streams z^b's tiles through LDS, keeps the logits
//     tile in its MFMA accumulators and reduces row max / sum exp / sum_j T_ij x_ij in place -- in-lane, two xor shuffles
//     across the lanes that share a row, LDS across the four waves (gemm.hip, EPI_LSE_PART) -- to one (max, sum, dot) triple
//     per row and column tile; infonce_combine_kernel merges the N/256 triples of a row into LSE_i and the loss term.
//     The column statistics of (a,b) are the row statistics of the transposed pass (b,a).
//   pass 2 (only the rank's own rows): the same product again, its epilogue turns the accumulator tile into dL/dG
//     (EPI_LOSS_W: needs LSE of both directions) and writes it as the split-bf16 operand of the gradient GEMM
//     dz^a += W z^b, accumulated in f32.
// Slab form (bsclip_infonce_set_impl(1), kept for A/B timing and as a second implementation to test against): logits in row
// slabs of <= 1024 rows as f32 in the workspace, a separate row-reduce / dL/dG kernel reads them back.
#include <math.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ void split_bf(float x, bf16_t& hi, bf16_t& lo) {
    hi = f2bf(x);
    lo = f2bf(x - bf2f(hi));
}

// One wave per row i in [0, Np): zn = z/max(||z||,eps) (rows >= N are zero), inv, PA = [hi|hi|lo], PB = [hi|lo|hi]
__global__ __launch_bounds__(256) void loss_prep_kernel(const float* __restrict__ z, int N, int Np, int D,
                                                         float* __restrict__ zn, float* __restrict__ inv,
                                                         bf16_t* __restrict__ PA, bf16_t* __restrict__ PB) {
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
        scale = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
    }
    if (lane == 0) inv[row] = scale;
    for (int c = lane * 4; c < D; c += 256) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < N) v = *reinterpret_cast<const f32x4*>(z + (size_t)row * D + c) * scale;
        *reinterpret_cast<f32x4*>(zn + (size_t)row * D + c) = v;
        bf16_t h[4], l[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) split_bf(v[i], h[i], l[i]);
        const uint2 hh = {(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
        const uint2 ll = {(unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16)};
        bf16_t* pa = PA + (size_t)row * 3 * D + c;
        bf16_t* pb = PB + (size_t)row * 3 * D + c;
        *reinterpret_cast<uint2*>(pa) = hh;
        *reinterpret_cast<uint2*>(pa + D) = hh;
        *reinterpret_cast<uint2*>(pa + 2 * D) = ll;
        *reinterpret_cast<uint2*>(pb) = hh;
        *reinterpret_cast<uint2*>(pb + D) = ll;
        *reinterpret_cast<uint2*>(pb + 2 * D) = hh;
    }
}

// PBt[d, :] = [zn^T hi | zn^T lo | zn^T hi]  (bf16 [D, 3*Np]) through a 64x64 f32 LDS tile
__global__ __launch_bounds__(256) void loss_transpose_split_kernel(const float* __restrict__ zn, int Np, int D,
                                                                    bf16_t* __restrict__ PBt) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;  // r over Np rows, c over D cols
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) tile[i][tx] = zn[(size_t)(r0 + i) * D + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const float v = tile[tx][i];  // zn[r0+tx][c0+i]
        bf16_t h, l;
        split_bf(v, h, l);
        bf16_t* dst = PBt + (size_t)(c0 + i) * 3 * Np + r0 + tx;
        dst[0] = h;
        dst[Np] = l;
        dst[2 * Np] = h;
    }
}

// cnt[i] = #{j : label_j == label_i}; one thread per i (N <= 8192: O(N^2) int compares, negligible)
__global__ void loss_count_kernel(const int64_t* __restrict__ labels, int N, float* __restrict__ cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int64_t li = labels[i];
    int c = 0;
    for (int j = 0; j < N; ++j) c += (labels[j] == li);
    cnt[i] = (float)c;
}

// One wave per row i < N of G (cosines, ld = Np): LSE_i of s*G over j < N, and contrib_i = cnt_i*LSE_i - sum_j T_ij s G_ij
__global__ __launch_bounds__(256) void loss_row_reduce_kernel(const float* __restrict__ G, int row0, int nrows, int N,
                                                               int Np, float s, const int64_t* __restrict__ labels,
                                                               const float* __restrict__ cnt, float* __restrict__ lse,
                                                               float* __restrict__ contrib) {
    const int lane = threadIdx.x & 63;
    const int r = (blockIdx.x * 256 + threadIdx.x) >> 6;  // row inside the slab
    if (r >= nrows) return;
    const int row = row0 + r;
    const float* g = G + (size_t)r * Np;
    const int64_t li = labels[row];
    float m = -INFINITY;
    for (int j = lane; j < N; j += 64) m = fmaxf(m, g[j]);
    m = wave_max(m) * s;
    float se = 0.f, dot = 0.f;
    for (int j = lane; j < N; j += 64) {
        const float x = g[j] * s;
        se += __expf(x - m);
        if (labels[j] == li) dot += x;
    }
    se = wave_sum(se);
    dot = wave_sum(dot);
    const float l = m + __logf(se);
    if (lane == 0) {
        lse[row] = l;
        contrib[row] = cnt[row] * l - dot;
    }
}

// W[i_local, j] = coef * ( cnt_i exp(sG - lse_ab[i]) + cnt_j exp(sG - lse_ba[j]) - 2 T_ij ), j < N, else 0,
// written as the split GEMM operand [hi | hi | lo] (bf16 [n_local, 3*Np]).
__global__ __launch_bounds__(256) void loss_w_kernel(const float* __restrict__ G, int N, int Np, int row0, int n_local,
                                                      float s, float coef, const int64_t* __restrict__ labels,
                                                      const float* __restrict__ cnt, const float* __restrict__ lse_ab,
                                                      const float* __restrict__ lse_ba, bf16_t* __restrict__ W) {
    const long total = (long)n_local * Np;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
        const int il = (int)(it / Np), j = (int)(it % Np);
        float w = 0.f;
        if (j < N) {
            const int i = row0 + il;
            const float x = G[(size_t)il * Np + j] * s;
            const float t = (labels[i] == labels[j]) ? 2.0f : 0.0f;
            w = coef * (cnt[i] * __expf(x - lse_ab[i]) + cnt[j] * __expf(x - lse_ba[j]) - t);
        }
        bf16_t h, l;
        split_bf(w, h, l);
        bf16_t* dst = W + (size_t)il * 3 * Np + j;
        dst[0] = h;
        dst[Np] = h;
        dst[2 * Np] = l;
    }
}

// loss = scale * sum over the directed pairs a != b of contrib[(a*nmod+b)*Np + i], i < N -- single workgroup, fixed order
// (bitwise reproducible).  Only entries pass 1 wrote are read: the (a,a) slots and the padding rows [N, Np) are skipped, so no
// hipMemsetAsync is needed (memset nodes inside a captured stream came back mis-ordered on replay: garbage loss, exact gradients)
__global__ __launch_bounds__(256) void loss_final_kernel(const float* __restrict__ contrib, int nmod, int Np, int N, float scale,
                                                          float* __restrict__ loss_out) {
    __shared__ float red[256];
    float s = 0.f;
    for (int slot = 0; slot < nmod * nmod; ++slot) {
        if (slot / nmod == slot % nmod) continue;
        for (int i = threadIdx.x; i < N; i += 256) s += contrib[(size_t)slot * Np + i];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_out[0] = red[0] * scale;
}

// merge the per-column-tile triples of pass 1: LSE_i = M + log sum_t s_t exp(m_t - M), contrib_i = cnt_i LSE_i - sum_t dot_t
__global__ __launch_bounds__(256) void infonce_combine_kernel(const float* __restrict__ part, int N, int tiles,
                                                              const float* __restrict__ cnt, float* __restrict__ lse,
                                                              float* __restrict__ contrib) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const f32x4* p = reinterpret_cast<const f32x4*>(part) + (size_t)i * tiles;
    float m = -INFINITY;
    for (int t = 0; t < tiles; ++t) m = fmaxf(m, p[t][0]);
    float s = 0.f, dot = 0.f;
    for (int t = 0; t < tiles; ++t) {   // fixed order: bitwise reproducible
        const f32x4 v = p[t];
        s += v[0] == -INFINITY ? 0.f : v[1] * __expf(v[0] - m);
        dot += v[2];
    }
    const float l = m + __logf(s);
    lse[i] = l;
    contrib[i] = cnt[i] * l - dot;
}

// 0 = by size (default): below 2 048 padded rows a product is <= 8 x 8 tiles of the 256 x 256 GEMM -- at the bench's N = 256 ONE
// tile per directed pair, 255 CUs idle -- and the slab form on 128 x 128 tiles is faster (0.18 vs 0.28 ms at N = 256, 1.35 vs
// 1.27 at N = 2 048, 8.3 vs 4.5 at N = 8 192: profiles/r02_e_infonce_fused_vs_slab.log); 1 = f32 logits slabs + separate reduce
// kernels; 2 = fused epilogues
int g_infonce_impl = 0;

inline int64_t align4(int64_t x) { return (x + 3) & ~(int64_t)3; }

constexpr int SLAB_ROWS = 1024;  // logits are formed [SLAB_ROWS x N] at a time: the N x N matrix never exists in HBM

struct Layout {
    int Np, slab;
    int64_t zn, inv, PA, PB, PBt, G, W, lse, contrib, cnt, dacc, part, total;
};

Layout make_layout(int N, int nmod, int D) {
    Layout L;
    L.Np = (N + 255) / 256 * 256;   // a whole number of 256-wide column tiles for the fused epilogues
    L.slab = L.Np < SLAB_ROWS ? L.Np : SLAB_ROWS;
    const int64_t Np = L.Np, SL = L.slab;
    int64_t o = 0;
    L.zn = o;      o += align4((int64_t)nmod * Np * D);
    L.inv = o;     o += align4((int64_t)nmod * Np);
    L.PA = o;      o += align4((int64_t)nmod * Np * 3 * D / 2);
    L.PB = o;      o += align4((int64_t)nmod * Np * 3 * D / 2);
    L.PBt = o;     o += align4((int64_t)nmod * D * 3 * Np / 2);
    L.G = o;       o += align4(SL * Np);
    L.W = o;       o += align4(SL * 3 * Np / 2);
    L.lse = o;     o += align4((int64_t)nmod * nmod * Np);
    L.contrib = o; o += align4((int64_t)nmod * nmod * Np);
    L.cnt = o;     o += align4(Np);
    L.dacc = o;    o += align4((int64_t)nmod * Np * D);
    L.part = o;    o += align4(Np * (Np / 256) * 4);
    L.total = o;
    return L;
}

}  // namespace

extern "C" int64_t bsclip_infonce_workspace_floats(int N, int nmod) {
    if (N <= 0 || nmod < 2 || nmod > 3) return -1;
    return make_layout(N, nmod, 768).total;
}

extern "C" int bsclip_infonce_fwd_bwd(const float* const* z, int nmod, const int64_t* labels, int N, int D, float scale,
                                      int row0, int n_local, float* loss_out, float* const* dz, float* workspace,
                                      void* stream) {
    // reference error behaviour: loss_func.py:35-36 raises ValueError for < 2 modalities (mapped by the host layer)
    BSCLIP_REQUIRE(nmod >= 2 && nmod <= 3, "Too less element for calculating the contrastive loss.");
    BSCLIP_REQUIRE(z && labels && loss_out && workspace, "bsclip_infonce_fwd_bwd: null pointer");
    BSCLIP_REQUIRE(D == 768, "bsclip_infonce_fwd_bwd: D=%d (supported: 768)", D);
    BSCLIP_REQUIRE(N > 0 && N <= 16384 && row0 >= 0 && n_local >= 0 && row0 + n_local <= N,
                   "bsclip_infonce_fwd_bwd: N=%d row0=%d n_local=%d", N, row0, n_local);
    BSCLIP_REQUIRE((((uintptr_t)workspace) & 15) == 0, "bsclip_infonce_fwd_bwd: workspace must be 16-B aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Layout L = make_layout(N, nmod, D);
    const int Np = L.Np;
    float* ws = workspace;
    float* zn = ws + L.zn;
    float* inv = ws + L.inv;
    bf16_t* PA = reinterpret_cast<bf16_t*>(ws + L.PA);
    bf16_t* PB = reinterpret_cast<bf16_t*>(ws + L.PB);
    bf16_t* PBt = reinterpret_cast<bf16_t*>(ws + L.PBt);
    float* G = ws + L.G;
    bf16_t* W = reinterpret_cast<bf16_t*>(ws + L.W);
    float* lse = ws + L.lse;
    float* contrib = ws + L.contrib;
    float* cnt = ws + L.cnt;
    float* dacc = ws + L.dacc;
    float* part = ws + L.part;
    const bool fused = g_infonce_impl == 2 || (g_infonce_impl == 0 && Np >= 2048);
    const size_t opA = (size_t)Np * 3 * D;  // elements per modality in PA / PB
    const size_t opT = (size_t)D * 3 * Np;  // elements per modality in PBt

    for (int m = 0; m < nmod; ++m) {
        BSCLIP_REQUIRE(z[m], "bsclip_infonce_fwd_bwd: z[%d] is null", m);
        hipLaunchKernelGGL(loss_prep_kernel, dim3(ceil_div(Np, 4)), dim3(256), 0, s, z[m], N, Np, D,
                           zn + (size_t)m * Np * D, inv + (size_t)m * Np, PA + m * opA, PB + m * opA);
        if (dz && n_local > 0)
            hipLaunchKernelGGL(loss_transpose_split_kernel, dim3(D / 64, Np / 64), dim3(256), 0, s,
                               zn + (size_t)m * Np * D, Np, D, PBt + m * opT);
    }
    hipLaunchKernelGGL(loss_count_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, s, labels, N, cnt);
    BSCLIP_LAUNCH_CHECK();

    const int ndir = nmod * (nmod - 1);
    // ---- pass 1: row LSE + loss contribution of every directed pair ----
    for (int a = 0; a < nmod; ++a)
        for (int b = 0; b < nmod; ++b) {
            if (a == b) continue;
            const int slot = a * nmod + b;
            if (fused) {
                int rc = bsclip_gemm_infonce(0, PA + a * opA, 3 * D, PB + b * opA, 3 * D, nullptr, 0, N, Np, 3 * D, labels, cnt,
                                             nullptr, nullptr, part, scale, 0.f, N, 0, stream);
                if (rc) return rc;
                hipLaunchKernelGGL(infonce_combine_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, s, part, N, Np / 256, cnt,
                                   lse + (size_t)slot * Np, contrib + (size_t)slot * Np);
                continue;
            }
            for (int r0 = 0; r0 < N; r0 += L.slab) {
                const int nr = N - r0 < L.slab ? N - r0 : L.slab;
                int rc = bsclip_gemm_bf16(PA + a * opA + (size_t)r0 * 3 * D, 3 * D, PB + b * opA, 3 * D, G, Np, nr, Np,
                                          3 * D, BSCLIP_EPI_F32, nullptr, stream);
                if (rc) return rc;
                hipLaunchKernelGGL(loss_row_reduce_kernel, dim3(ceil_div(nr, 4)), dim3(256), 0, s, G, r0, nr, N, Np, scale,
                                   labels, cnt, lse + (size_t)slot * Np, contrib + (size_t)slot * Np);
            }
        }
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, s, contrib, nmod, Np, N, 1.0f / ((float)ndir * (float)N),
                       loss_out);
    BSCLIP_LAUNCH_CHECK();
    if (!dz || n_local == 0) return BSCLIP_OK;

    // ---- pass 2: d loss / d zn_a[rows row0 .. row0+n_local) = sum_b W_ab zn_b ----
    const float coef = scale / ((float)ndir * (float)N);
    for (int a = 0; a < nmod; ++a) {
        BSCLIP_REQUIRE(dz[a], "bsclip_infonce_fwd_bwd: dz[%d] is null", a);
        float* da = dacc + (size_t)a * Np * D;
        bool first = true;
        for (int b = 0; b < nmod; ++b) {
            if (a == b) continue;
            for (int l0 = 0; l0 < n_local; l0 += L.slab) {  // row slabs of the rank's own rows
                const int nr = n_local - l0 < L.slab ? n_local - l0 : L.slab;
                int rc;
                if (fused) {
                    rc = bsclip_gemm_infonce(1, PA + a * opA + (size_t)(row0 + l0) * 3 * D, 3 * D, PB + b * opA, 3 * D, W, 3 * Np,
                                             nr, Np, 3 * D, labels, cnt, lse + (size_t)(a * nmod + b) * Np,
                                             lse + (size_t)(b * nmod + a) * Np, nullptr, scale, coef, N, row0 + l0, stream);
                    if (rc) return rc;
                } else {
                    rc = bsclip_gemm_bf16(PA + a * opA + (size_t)(row0 + l0) * 3 * D, 3 * D, PB + b * opA, 3 * D, G, Np, nr, Np,
                                          3 * D, BSCLIP_EPI_F32, nullptr, stream);
                    if (rc) return rc;
                    long tot = (long)nr * Np;
                    long blocks = (tot + 255) / 256;
                    if (blocks > 4096) blocks = 4096;
                    hipLaunchKernelGGL(loss_w_kernel, dim3((unsigned)blocks), dim3(256), 0, s, G, N, Np, row0 + l0, nr, scale,
                                       coef, labels, cnt, lse + (size_t)(a * nmod + b) * Np, lse + (size_t)(b * nmod + a) * Np,
                                       W);
                }
                float* dar = da + (size_t)l0 * D;
                bsclip_epi_args ea{};
                ea.struct_size = sizeof(ea);
                ea.resid = dar;
                ea.ld_resid = D;
                rc = bsclip_gemm_bf16(W, 3 * Np, PBt + b * opT, 3 * Np, dar, D, nr, D, 3 * Np,
                                      first ? BSCLIP_EPI_F32 : BSCLIP_EPI_RESID_F32, first ? nullptr : &ea, stream);
                if (rc) return rc;
            }
            first = false;
        }
        // Jacobian of the in-loss F.normalize: dz = (g - zn (zn . g)) / ||z||
        int rc = bsclip_l2norm_bwd(zn + ((size_t)a * Np + row0) * D, inv + (size_t)a * Np + row0, da, n_local, D, dz[a],
                                   stream);
        if (rc) return rc;
    }
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int bsclip_infonce_set_impl(int impl) {
    BSCLIP_REQUIRE(impl >= 0 && impl <= 2, "bsclip_infonce_set_impl: %d (0 = by size, 1 = logits slabs, 2 = fused epilogues)", impl);
    g_infonce_impl = impl;
    return BSCLIP_OK;
}
