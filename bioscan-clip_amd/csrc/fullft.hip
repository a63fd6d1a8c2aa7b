// Gradient kernels that only full fine-tuning needs (`disable_lora: true`, reference
// bioscanclip/model/simple_clip.py:199-201 unfreezes every parameter): the LoRA regime never differentiates LayerNorm gains,
// embeddings or the patch filters, so the step kernels do not produce these.  Weight gradients of the Linear layers reuse the
// MFMA GEMM (dW = dY^T X on transposed operands, as the trainable heads always did); what is new here is
//   * LayerNorm gamma / beta gradients (autograd of timm / HF LayerNorm): d_gamma[c] = sum_r dy[r,c] xhat[r,c],
//     d_beta[c] = sum_r dy[r,c], with dy assembled exactly as bsclip_layernorm_bwd assembles it;
//   * BertEmbeddings gradients (word rows scattered by id, position rows summed over the batch, token-type rows);
//   * a row gather + cast (f32 rows with a periodic row map -> bf16), which feeds the patch-embedding dW GEMM.
// All HBM-bound; every reduction is ordered (per-workgroup slabs summed in a fixed order; embedding rows owned by one workgroup
// each): no float atomics, a step is bitwise reproducible in this regime too.
#include "common.h"

namespace {

constexpr int PG_BLOCK = 256;   // 4 waves
constexpr int PG_MAX_BLOCKS = 1024;

template <int H, bool X_BF16>
__global__ __launch_bounds__(PG_BLOCK) void ln_pgrad_kernel(const void* __restrict__ x, int ld_x, const float* __restrict__ stats,
                                                            int M, const float* __restrict__ g_resid, int ld_gr,
                                                            const bf16_t* __restrict__ g_gemm, int ld_g,
                                                            const float* __restrict__ dt, const float* __restrict__ lora_a,
                                                            int mode, DropCfg in_drop, float* __restrict__ partial) {
    BSCLIP_DROP_RESOLVE(in_drop);
    constexpr int NV = H / 256;
    __shared__ float red[2 * H];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int wave = (blockIdx.x * PG_BLOCK + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * PG_BLOCK) >> 6;
    for (int i = threadIdx.x; i < 2 * H; i += PG_BLOCK) red[i] = 0.f;
    f32x4 dg[NV], db[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        dg[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        db[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int row = wave; row < M; row += nwaves) {
        const float mean = stats[2 * (size_t)row], rstd = stats[2 * (size_t)row + 1];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = j * 256 + lane * 4;
            f32x4 v;
            if constexpr (X_BF16) {
                const uint2 u = *reinterpret_cast<const uint2*>(static_cast<const bf16_t*>(x) + (size_t)row * ld_x + c);
                v = f32x4{bf2f(u.x & 0xffff), bf2f(u.x >> 16), bf2f(u.y & 0xffff), bf2f(u.y >> 16)};
            } else {
                v = *reinterpret_cast<const f32x4*>(static_cast<const float*>(x) + (size_t)row * ld_x + c);
            }
            f32x4 dy = {0.f, 0.f, 0.f, 0.f};
            if (g_gemm) {
                const uint2 u = *reinterpret_cast<const uint2*>(g_gemm + (size_t)row * ld_g + c);
                dy = f32x4{bf2f(u.x & 0xffff), bf2f(u.x >> 16), bf2f(u.y & 0xffff), bf2f(u.y >> 16)};
            }
            if (dt) {
#pragma unroll
                for (int r = 0; r < 8; ++r) dy += dt[(size_t)row * 8 + r] * *reinterpret_cast<const f32x4*>(lora_a + r * H + c);
            }
            if (mode == 1 && g_resid) dy += *reinterpret_cast<const f32x4*>(g_resid + (size_t)row * ld_gr + c);
            if (in_drop.thr16) dy = drop4(in_drop, (unsigned)row * H + c, dy);   // the LN output was dropped in forward
            dg[j] += dy * ((v - mean) * rstd);
            db[j] += dy;
        }
    }
    for (int w = 0; w < PG_BLOCK / 64; ++w) {   // cross-wave sum in wave order (fixed order of float adds)
        __syncthreads();
        if (wib == w) {
#pragma unroll
            for (int j = 0; j < NV; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    red[j * 256 + lane * 4 + i] += dg[j][i];
                    red[H + j * 256 + lane * 4 + i] += db[j][i];
                }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * H; i += PG_BLOCK) partial[(size_t)blockIdx.x * 2 * H + i] = red[i];
}

// out[c] += sum_b partial[b][c], c < n, in a fixed order: thread (g, c) sums the slabs b = g, g + 8, ... (four loads in flight),
// then the eight group sums are added g = 0 .. 7.  32 columns per workgroup: a serial walk over 1 024 slabs by one thread per
// column (the first version) took 226 us for n = 1 536.
__global__ __launch_bounds__(256) void slab_reduce_add_kernel(const float* __restrict__ partial, int nblocks, int n,
                                                              float* __restrict__ out0, float* __restrict__ out1, int split) {
    __shared__ float red[8][32];
    const int cc = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cc;
    float s = 0.f;
    if (c < n) {
        int b = g;
        for (; b + 24 < nblocks; b += 32) {
            const float v0 = partial[(size_t)b * n + c], v1 = partial[(size_t)(b + 8) * n + c];
            const float v2 = partial[(size_t)(b + 16) * n + c], v3 = partial[(size_t)(b + 24) * n + c];
            s += v0;
            s += v1;
            s += v2;
            s += v3;
        }
        for (; b < nblocks; b += 8) s += partial[(size_t)b * n + c];
    }
    red[g][cc] = s;
    __syncthreads();
    if (g == 0 && c < n) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[i][cc];
        if (c < split) out0[c] += t;
        else out1[c - split] += t;
    }
}

// d_pos[s, :] += sum_b d[b, s, :]   (ordered)
__global__ __launch_bounds__(256) void embed_grad_pos_kernel(const float* __restrict__ d, int B, int S, int H,
                                                             float* __restrict__ d_pos) {
    const int i = blockIdx.x * 256 + threadIdx.x;   // over S * H / 4
    if (i >= S * H / 4) return;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < B; ++b) s += *reinterpret_cast<const f32x4*>(d + ((size_t)b * S * H) + (size_t)i * 4);
    f32x4* o = reinterpret_cast<f32x4*>(d_pos + (size_t)i * 4);
    *o = *o + s;
}
// Word rows: workgroup v owns vocabulary row v.  Every wave scans the ids 64 at a time (one per lane), `ballot` marks the tokens
// that hold v, and their gradient rows are added in token order: a fixed order, no atomics.  The ids (a few hundred KB) come
// out of L2 for all workgroups; the gradient rows are read once overall.  nn.Embedding(padding_idx): that row gets none.
__global__ __launch_bounds__(256) void embed_grad_word_kernel(const int64_t* __restrict__ ids, int M, int H, int vocab, int pad_id,
                                                              const float* __restrict__ d, float* __restrict__ d_word) {
    const int v = blockIdx.x;
    if (v == pad_id) return;
    const int lane = threadIdx.x & 63, c = threadIdx.x * 4;
    const bool active = c < H;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    bool any = false;
    for (int base = 0; base < M; base += 64) {
        long id = -1;
        if (base + lane < M) {
            id = ids[base + lane];
            id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);   // the forward lookup clamps the same way
        }
        unsigned long long mask = __ballot(id == v);
        while (mask) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            any = true;
            if (active) acc += *reinterpret_cast<const f32x4*>(d + (size_t)(base + j) * H + c);
        }
    }
    if (any && active) {
        f32x4* o = reinterpret_cast<f32x4*>(d_word + (size_t)v * H + c);
        *o = *o + acc;
    }
}

// Token-type rows (two): per-workgroup slabs over contiguous row ranges, summed afterwards in slab order.
__global__ __launch_bounds__(256) void embed_grad_type_kernel(const int64_t* __restrict__ type_ids, int M, int H, int rows_per_block,
                                                              const float* __restrict__ d, float* __restrict__ partial) {
    const int c = threadIdx.x * 4;
    if (c >= H) return;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    for (int r = r0; r < r1; ++r) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(d + (size_t)r * H + c);
        if (type_ids && type_ids[r] != 0) a1 += v;
        else a0 += v;
    }
    float* p = partial + (size_t)blockIdx.x * 2 * H;
    *reinterpret_cast<f32x4*>(p + c) = a0;
    *reinterpret_cast<f32x4*>(p + H + c) = a1;
}

// dst[r, :] = bf16(src[(r / p_out) * p_in + off + r % p_out, :])   (rows of H f32 -> bf16), 4 values per thread
__global__ __launch_bounds__(256) void gather_cast_rows_kernel(const float* __restrict__ src, int ld_src, int rows_out, int p_in,
                                                               int p_out, int off, int H, bf16_t* __restrict__ dst, int ld_dst) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;   // over rows_out * H / 4
    const int per = H / 4;
    if (i >= (long)rows_out * per) return;
    const int r = (int)(i / per), c = (int)(i % per) * 4;
    const int rs = (r / p_out) * p_in + off + r % p_out;
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)rs * ld_src + c);
    uint2 o;
    o.x = pack_bf2(v[0], v[1]);
    o.y = pack_bf2(v[2], v[3]);
    *reinterpret_cast<uint2*>(dst + (size_t)r * ld_dst + c) = o;
}

}  // namespace

void bsclip_launch_slab_reduce_add(const float* partial, int nblocks, int n, float* out0, float* out1, int split, hipStream_t s) {
    hipLaunchKernelGGL(slab_reduce_add_kernel, dim3(ceil_div(n, 32)), dim3(256), 0, s, partial, nblocks, n, out0, out1, split);
}

extern "C" int64_t bsclip_ln_param_grad_workspace_floats(int H) { return (int64_t)PG_MAX_BLOCKS * 2 * H; }

extern "C" int bsclip_ln_param_grad(const void* x, int ld_x, int x_bf16, const float* stats, int M, int H,
                                    const float* g_resid, int ld_gr, const void* g_gemm, int ld_g, const float* dt,
                                    const float* lora_a, int mode, float in_dropout_p, uint32_t in_dropout_seed,
                                    float* d_gamma, float* d_beta, float* workspace, void* stream) {
    BSCLIP_REQUIRE(x && stats && d_gamma && d_beta && workspace && M > 0, "bsclip_ln_param_grad: null/empty input");
    BSCLIP_REQUIRE(H == 768 || H == 512, "bsclip_ln_param_grad: H=%d (supported: 768, 512)", H);
    BSCLIP_REQUIRE(ld_x >= H && ld_x % 4 == 0 && (!g_gemm || (ld_g >= H && ld_g % 4 == 0)) && (!g_resid || (ld_gr >= H && ld_gr % 4 == 0)),
                   "bsclip_ln_param_grad: leading dimensions");
    BSCLIP_REQUIRE((dt == nullptr) == (lora_a == nullptr), "bsclip_ln_param_grad: dt and lora_a go together");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int blocks = ceil_div(M, 4 * 8);
    if (blocks > PG_MAX_BLOCKS) blocks = PG_MAX_BLOCKS;
    const DropCfg drop = make_drop(in_dropout_p, in_dropout_seed);
    const bf16_t* gg = static_cast<const bf16_t*>(g_gemm);
#define PG_LAUNCH(HH, XB)                                                                                                    \
    hipLaunchKernelGGL((ln_pgrad_kernel<HH, XB>), dim3(blocks), dim3(PG_BLOCK), 0, s, x, ld_x, stats, M, g_resid, ld_gr, gg, ld_g, \
                       dt, lora_a, mode, drop, workspace)
    if (H == 768) { if (x_bf16) PG_LAUNCH(768, true); else PG_LAUNCH(768, false); }
    else          { if (x_bf16) PG_LAUNCH(512, true); else PG_LAUNCH(512, false); }
#undef PG_LAUNCH
    hipLaunchKernelGGL(slab_reduce_add_kernel, dim3(ceil_div(2 * H, 32)), dim3(256), 0, s, workspace, blocks, 2 * H, d_gamma,
                       d_beta, H);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

constexpr int EG_MAX_BLOCKS = 512;

extern "C" int64_t bsclip_embed_grad_workspace_floats(int H) { return (int64_t)EG_MAX_BLOCKS * 2 * H; }

extern "C" int bsclip_embed_grad(const int64_t* ids, const int64_t* type_ids, int B, int S, int H, int vocab, int pad_id,
                                 const float* d_emb, float* d_word, float* d_pos, float* d_type, float* workspace, void* stream) {
    BSCLIP_REQUIRE(ids && d_emb && d_word && d_pos && d_type && workspace && B > 0 && S > 0 && H % 4 == 0 && H <= 1024 && vocab > 0,
                   "bsclip_embed_grad: null/empty input or H=%d not in (0, 1024], H %% 4 == 0", H);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int M = B * S;
    hipLaunchKernelGGL(embed_grad_pos_kernel, dim3(ceil_div(S * H / 4, 256)), dim3(256), 0, s, d_emb, B, S, H, d_pos);
    hipLaunchKernelGGL(embed_grad_word_kernel, dim3(vocab), dim3(256), 0, s, ids, M, H, vocab, pad_id, d_emb, d_word);
    const int rows_per_block = max(16, ceil_div(M, EG_MAX_BLOCKS));
    const int blocks = ceil_div(M, rows_per_block);
    hipLaunchKernelGGL(embed_grad_type_kernel, dim3(blocks), dim3(256), 0, s, type_ids, M, H, rows_per_block, d_emb, workspace);
    bsclip_launch_slab_reduce_add(workspace, blocks, 2 * H, d_type, d_type, 2 * H, s);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int bsclip_gather_cast_rows(const float* src, int ld_src, int rows_out, int period_in, int period_out, int offset,
                                       int H, void* dst_bf16, int ld_dst, void* stream) {
    BSCLIP_REQUIRE(src && dst_bf16 && rows_out > 0 && period_in >= period_out && period_out > 0 && offset >= 0 &&
                       offset + period_out <= period_in && H % 4 == 0 && ld_src >= H && ld_dst >= H && ld_src % 4 == 0 && ld_dst % 4 == 0,
                   "bsclip_gather_cast_rows: bad arguments");
    const long n = (long)rows_out * (H / 4);
    hipLaunchKernelGGL(gather_cast_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       src, ld_src, rows_out, period_in, period_out, offset, H, static_cast<bf16_t*>(dst_bf16), ld_dst);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}
