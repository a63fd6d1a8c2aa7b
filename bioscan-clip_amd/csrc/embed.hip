// HBM-bound data-movement kernels around the encoders (gfx950): patch im2col, cls rows, BERT embedding sums,
// bf16 transpose, f32->bf16 cast, LoRA-B refresh of the augmented QKV weight.  16-B accesses, coalesced on the
// write side.  Reference sites: timm PatchEmbed / cls_token / pos_embed (via image_encoder.py:108-109), HF
// BertEmbeddings (via dna_encoder.py:105, language_encoder.py:89).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// out[(b*196 + py*14 + px), c*256 + ky*16 + kx] = image[b, c, py*16+ky, px*16+kx]; one thread = 8 output columns
// SPLIT: the row is written as [hi | lo | hi] (3 x 768 columns), hi = bf16(x), lo = bf16(x - hi): against a weight stored as
// [hi | hi | lo] one bf16 MFMA GEMM with K = 2304 forms hi.hi + lo.hi + hi.lo, i.e. the product to ~2^-16 instead of 2^-8
// relative (the trick the loss GEMM uses).  The patch embedding is 0.4 % of the step's FLOPs and the one GEMM whose operand
// rounding (5.3e-3 at its output) every later block amplifies: DESIGN.md 4.
template <bool SPLIT>
__global__ __launch_bounds__(256) void im2col_patch16_kernel(const float* __restrict__ img, int B,
                                                              bf16_t* __restrict__ out, int ld) {
    const long total = (long)B * 196 * 96;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
        const int chunk = (int)(it % 96);
        const long row = it / 96;
        const int b = (int)(row / 196), p = (int)(row % 196);
        const int py = p / 14, px = p % 14;
        const int col = chunk * 8;
        const int c = col >> 8, ky = (col >> 4) & 15, kx0 = col & 15;
        const float* src = img + (((size_t)b * 3 + c) * 224 + (py * 16 + ky)) * 224 + px * 16 + kx0;
        const f32x4 a = *reinterpret_cast<const f32x4*>(src);
        const f32x4 d = *reinterpret_cast<const f32x4*>(src + 4);
        u32x4 o;
        o[0] = pack_bf2(a[0], a[1]);
        o[1] = pack_bf2(a[2], a[3]);
        o[2] = pack_bf2(d[0], d[1]);
        o[3] = pack_bf2(d[2], d[3]);
        *reinterpret_cast<u32x4*>(out + (size_t)row * ld + col) = o;
        if constexpr (SPLIT) {
            auto hi = [](unsigned w, int k) { return bf2f((bf16_t)(k ? w >> 16 : w & 0xffff)); };
            u32x4 l;
            l[0] = pack_bf2(a[0] - hi(o[0], 0), a[1] - hi(o[0], 1));
            l[1] = pack_bf2(a[2] - hi(o[1], 0), a[3] - hi(o[1], 1));
            l[2] = pack_bf2(d[0] - hi(o[2], 0), d[1] - hi(o[2], 1));
            l[3] = pack_bf2(d[2] - hi(o[3], 0), d[3] - hi(o[3], 1));
            *reinterpret_cast<u32x4*>(out + (size_t)row * ld + 768 + col) = l;
            *reinterpret_cast<u32x4*>(out + (size_t)row * ld + 1536 + col) = o;
        }
    }
}

template <bool X_BF16>
__global__ void vit_cls_rows_kernel(void* __restrict__ x, const float* __restrict__ cls,
                                    const float* __restrict__ pos, int B, int S, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, c = i % H;
    if constexpr (X_BF16) static_cast<bf16_t*>(x)[(size_t)b * S * H + c] = f2bf(cls[c] + pos[c]);
    else static_cast<float*>(x)[(size_t)b * S * H + c] = cls[c] + pos[c];
}

// one wave per token row
__global__ __launch_bounds__(256) void bert_embed_kernel(const int64_t* __restrict__ ids,
                                                          const int64_t* __restrict__ type_ids, int M, int S, int H,
                                                          const float* __restrict__ word, int vocab,
                                                          const float* __restrict__ pos, const float* __restrict__ type,
                                                          float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = (blockIdx.x * 256 + threadIdx.x) >> 6;
    if (row >= M) return;
    long id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // never read outside the table
    const int t = row % S;
    const long tt = type_ids ? type_ids[row] : 0;
    const float* w = word + (size_t)id * H;
    const float* p = pos + (size_t)t * H;
    const float* ty = type + (size_t)(tt ? 1 : 0) * H;
    for (int c = lane * 4; c < H; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(w + c) + *reinterpret_cast<const f32x4*>(p + c) +
                        *reinterpret_cast<const f32x4*>(ty + c);
        *reinterpret_cast<f32x4*>(out + (size_t)row * H + c) = v;
    }
}

// out[C,R] = in[R,C]^T through a 64x64 LDS tile (+1 pad column against bank conflicts)
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, int ld_in, int R, int C,
                                                              bf16_t* __restrict__ out, int ld_out) {
    __shared__ bf16_t tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? in[(size_t)r * ld_in + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < R) out[(size_t)c * ld_out + r] = tile[tx][i];
    }
}

// The same transpose for 16-byte-aligned operands (ld % 8 == 0): every thread moves two 16-byte chunks in and two out, the
// 64 x 64 tile sits in LDS with a 33-dword row stride (the 8 source rows a thread gathers for one output chunk land in 8
// different banks).  COLSUM: the workgroup also emits the column sums of its 64 source rows, partial[blockIdx.y][c] (f32,
// summed in row order) -- with dY as the source that is the bias gradient's slab, for free while the tile is in registers.
template <bool COLSUM>
__global__ __launch_bounds__(256) void transpose64_kernel(const bf16_t* __restrict__ in, int ld_in, int R, int C,
                                                          bf16_t* __restrict__ out, int ld_out, float* __restrict__ partial) {
    __shared__ unsigned tile[64][33];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int chunk = threadIdx.x + 256 * k, row = chunk >> 3, cc = chunk & 7;
        const int r = r0 + row, c = c0 + cc * 8;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r < R) {
            if (c + 8 <= C) {
                v = *reinterpret_cast<const u32x4*>(in + (size_t)r * ld_in + c);
            } else {
                for (int j = 0; j < 8; ++j)
                    if (c + j < C) v[j >> 1] |= (unsigned)(*reinterpret_cast<const unsigned short*>(in + (size_t)r * ld_in + c + j)) << (16 * (j & 1));
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[row][cc * 4 + j] = v[j];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int chunk = threadIdx.x + 256 * k, orow = chunk >> 3, oc = chunk & 7;   // output row = source column c0 + orow
        unsigned short e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned w = tile[oc * 8 + j][orow >> 1];
            e[j] = (unsigned short)((orow & 1) ? (w >> 16) : (w & 0xffff));
        }
        const int c = c0 + orow, r = r0 + oc * 8;
        if (c < C) {
            if (r + 8 <= R) {
                u32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (unsigned)e[2 * j] | ((unsigned)e[2 * j + 1] << 16);
                *reinterpret_cast<u32x4*>(out + (size_t)c * ld_out + r) = o;
            } else {
                for (int j = 0; j < 8; ++j)
                    if (r + j < R) *reinterpret_cast<unsigned short*>(out + (size_t)c * ld_out + r + j) = e[j];
            }
        }
        if constexpr (COLSUM) {
            float sum = 0.f;   // rows >= R were loaded as zeros
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += bf2f(e[j]);
            // the 8 lanes oc = 0 .. 7 of one output row are consecutive: ordered sum of their 8-row pieces
            float tot = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) tot += __shfl(sum, (threadIdx.x & 56) + q, 64);
            if (oc == 0 && c < C) partial[(size_t)blockIdx.y * C + c] = tot;
        }
    }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ in, long n,
                                                             bf16_t* __restrict__ out) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(in + 4 * i);
        uint2 o;
        o.x = pack_bf2(v[0], v[1]);
        o.y = pack_bf2(v[2], v[3]);
        *reinterpret_cast<uint2*>(out + 4 * i) = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = (n4 << 2) + threadIdx.x;
        out[i] = f2bf(in[i]);
    }
}

// W_aug rows [0,H): cols [H,H+4) = B_q[n,:]; rows [2H,3H): cols [H+4,H+8) = B_v[n,:]
// for every LoRA layer of an encoder in one launch (blockIdx.y = layer): table[l] = {W_aug, B_q, B_v} device addresses
__global__ void waug_set_lora_layers_kernel(const int64_t* __restrict__ table, int ld_w, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * H) return;
    const int64_t* t = table + 3 * (size_t)blockIdx.y;
    bf16_t* w = reinterpret_cast<bf16_t*>(t[0]);
    const bool is_v = i >= H;
    const int n = is_v ? i - H : i;
    const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(is_v ? t[2] : t[1]) + (size_t)n * 4);
    uint2 o;
    o.x = pack_bf2(v[0], v[1]);
    o.y = pack_bf2(v[2], v[3]);
    *reinterpret_cast<uint2*>(w + (size_t)(is_v ? 2 * H + n : n) * ld_w + H + (is_v ? 4 : 0)) = o;
}

// HF "extended attention mask": (1 - m) * finfo(float32).min, added to the scores of padded keys
__global__ void mask_to_bias_kernel(const int64_t* __restrict__ mask, int n, float* __restrict__ bias) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) bias[i] = mask[i] ? 0.0f : -3.4028234663852886e38f;
}

}  // namespace

extern "C" int bsclip_mask_to_bias(const int64_t* mask, int n, float* bias, void* stream) {
    BSCLIP_REQUIRE(mask && bias && n > 0, "bsclip_mask_to_bias: bad args");
    hipLaunchKernelGGL(mask_to_bias_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), mask,
                       n, bias);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int bsclip_im2col_patch16(const float* image, int B, void* cols_bf16, int ld_cols, int split, void* stream) {
    BSCLIP_REQUIRE(image && cols_bf16 && B > 0, "bsclip_im2col_patch16: bad args");
    BSCLIP_REQUIRE(ld_cols % 8 == 0 && ld_cols >= (split ? 2304 : 768) && (((uintptr_t)cols_bf16) & 15) == 0,
                   "bsclip_im2col_patch16: ld_cols=%d (>= %d, multiple of 8, 16-byte aligned rows)", ld_cols, split ? 2304 : 768);
    const long total = (long)B * 196 * 96;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (split)
        hipLaunchKernelGGL((im2col_patch16_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                           image, B, static_cast<bf16_t*>(cols_bf16), ld_cols);
    else
        hipLaunchKernelGGL((im2col_patch16_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                           image, B, static_cast<bf16_t*>(cols_bf16), ld_cols);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int bsclip_vit_cls_rows(void* x, int x_bf16, const float* cls_token, const float* pos_embed, int B, int S, int H,
                                   void* stream) {
    BSCLIP_REQUIRE(x && cls_token && pos_embed && B > 0, "bsclip_vit_cls_rows: bad args");
    if (x_bf16)
        hipLaunchKernelGGL((vit_cls_rows_kernel<true>), dim3(ceil_div(B * H, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           x, cls_token, pos_embed, B, S, H);
    else
        hipLaunchKernelGGL((vit_cls_rows_kernel<false>), dim3(ceil_div(B * H, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           x, cls_token, pos_embed, B, S, H);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int bsclip_bert_embed(const int64_t* ids, const int64_t* type_ids, int B, int S, int H, const float* word,
                                 int vocab, const float* pos, const float* type, float* out, void* stream) {
    BSCLIP_REQUIRE(ids && word && pos && type && out && B > 0 && S > 0 && H % 4 == 0 && vocab > 0,
                   "bsclip_bert_embed: bad args");
    const int M = B * S;
    hipLaunchKernelGGL(bert_embed_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), ids,
                       type_ids, M, S, H, word, vocab, pos, type, out);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

static bool transpose_aligned(const void* in, int ld_in, const void* out, int ld_out) {
    return ld_in % 8 == 0 && ld_out % 8 == 0 && ((((uintptr_t)in) | ((uintptr_t)out)) & 15) == 0;
}

extern "C" int bsclip_transpose_bf16(const void* in, int ld_in, int R, int C, void* out, int ld_out, void* stream) {
    BSCLIP_REQUIRE(in && out && R > 0 && C > 0 && ld_in >= C && ld_out >= R, "bsclip_transpose_bf16: bad args");
    const dim3 grid(ceil_div(C, 64), ceil_div(R, 64));
    if (transpose_aligned(in, ld_in, out, ld_out))
        hipLaunchKernelGGL((transpose64_kernel<false>), grid, dim3(256), 0, static_cast<hipStream_t>(stream),
                           static_cast<const bf16_t*>(in), ld_in, R, C, static_cast<bf16_t*>(out), ld_out, nullptr);
    else
        hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream),
                           static_cast<const bf16_t*>(in), ld_in, R, C, static_cast<bf16_t*>(out), ld_out);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int64_t bsclip_transpose_colsum_workspace_floats(int R, int C) { return (int64_t)ceil_div(R, 64) * C; }

extern "C" int bsclip_transpose_colsum_bf16(const void* in, int ld_in, int R, int C, void* out, int ld_out, float* colsum,
                                            float* workspace, void* stream) {
    BSCLIP_REQUIRE(in && out && colsum && workspace && R > 0 && C > 0 && ld_in >= C && ld_out >= R,
                   "bsclip_transpose_colsum_bf16: bad args");
    BSCLIP_REQUIRE(transpose_aligned(in, ld_in, out, ld_out), "bsclip_transpose_colsum_bf16: operands must be 16-byte aligned, ld %% 8 == 0");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int slabs = ceil_div(R, 64);
    hipLaunchKernelGGL((transpose64_kernel<true>), dim3(ceil_div(C, 64), slabs), dim3(256), 0, s, static_cast<const bf16_t*>(in),
                       ld_in, R, C, static_cast<bf16_t*>(out), ld_out, workspace);
    bsclip_launch_slab_reduce_add(workspace, slabs, C, colsum, colsum, C, s);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int bsclip_cast_f32_bf16(const float* in, int64_t n, void* out, void* stream) {
    BSCLIP_REQUIRE(in && out && n > 0, "bsclip_cast_f32_bf16: bad args");
    long blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), in,
                       (long)n, static_cast<bf16_t*>(out));
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int bsclip_waug_set_lora_layers(const int64_t* table_dev, int layers, int ld_w, int H, void* stream) {
    BSCLIP_REQUIRE(table_dev && layers > 0 && layers <= 65535 && ld_w >= H + BSCLIP_KPAD && ld_w % 4 == 0 && H % 4 == 0,
                   "bsclip_waug_set_lora_layers: bad args");
    hipLaunchKernelGGL(waug_set_lora_layers_kernel, dim3(ceil_div(2 * H, 256), layers), dim3(256), 0, static_cast<hipStream_t>(stream),
                       table_dev, ld_w, H);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}
