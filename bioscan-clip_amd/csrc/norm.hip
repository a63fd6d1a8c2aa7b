// LayerNorm forward/backward and L2-normalise for gfx950.  HBM-bound row kernels: one 64-lane wave per row,
// 16-B coalesced loads, statistics in f32 via wavefront shuffles, everything a consumer needs fused in:
//   fwd: bf16 GEMM operand, optional f32 residual copy, (mean,rstd), and the rank-8 LoRA projection t = y A^T
//        written into the K-augmentation columns of the operand (so the LoRA branch costs no extra pass);
//   bwd: gradient assembly (f32 residual grad + bf16 GEMM grad + dt . A), dx in f32 and as the next bf16 operand.
// Reference arithmetic replaced: timm norm1/norm2/norm (eps 1e-6) reached from image_encoder.py:108-109; HF
// BertLayerNorm (eps 1e-12) reached from dna_encoder.py:105 / language_encoder.py:89; F.normalize at
// simple_clip.py:34,47,49.
#include "common.h"

namespace {

constexpr int LN_BLOCK = 256;  // 4 waves = 4 rows in flight per workgroup

// Reduce 8 per-lane partial sums over the 64 lanes with 10 shuffles (halving the live set each step).
// On return every lane holds, in its return value, the total of index r = ((lane>>5)&1)*4 + ((lane>>4)&1)*2 + ((lane>>3)&1).
__device__ __forceinline__ float reduce8(float (&p)[8], int lane) {
    float q[4];
    const bool hi5 = lane & 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float send = hi5 ? p[i] : p[i + 4];
        const float keep = hi5 ? p[i + 4] : p[i];
        q[i] = keep + __shfl_xor(send, 32, 64);
    }
    float r2[2];
    const bool hi4 = lane & 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float send = hi4 ? q[i] : q[i + 2];
        const float keep = hi4 ? q[i + 2] : q[i];
        r2[i] = keep + __shfl_xor(send, 16, 64);
    }
    const bool hi3 = lane & 8;
    float v = (hi3 ? r2[1] : r2[0]) + __shfl_xor(hi3 ? r2[0] : r2[1], 8, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 1, 64);
    return v;
}

template <int H, bool X_BF16>
__device__ __forceinline__ void load_row(const void* x, int ld_x, int row, int lane, f32x4 (&v)[H / 256]) {
    constexpr int NV = H / 256;
    if constexpr (X_BF16) {
        const bf16_t* p = static_cast<const bf16_t*>(x) + (size_t)row * ld_x;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const uint2 u = ld_stream(reinterpret_cast<const uint2*>(p + j * 256 + lane * 4));
            v[j] = f32x4{bf2f(u.x & 0xffff), bf2f(u.x >> 16), bf2f(u.y & 0xffff), bf2f(u.y >> 16)};
        }
    } else {
        const float* p = static_cast<const float*>(x) + (size_t)row * ld_x;
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = ld_stream(reinterpret_cast<const f32x4*>(p + j * 256 + lane * 4));
    }
}

// Y_FP8: the GEMM operand is written as OCP fp8 e4m3 (scale 1: LayerNorm outputs are O(1); saturated at +-448) into y_bf16
// reinterpreted as bytes (row stride ld_y BYTES), and the LoRA t block as bf16 into its own buffer t_aug (row stride ld_t
// elements, 64 columns: t in [0,8), zeros after) -- the bf16 K-augmentation tile of bsclip_gemm_fp8 (BASELINE configs[4]).
template <int H, bool X_BF16, bool LORA, bool Y_FP8 = false>
__global__ __launch_bounds__(LN_BLOCK) void layernorm_fwd_kernel(const void* __restrict__ x, int ld_x, int M,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float eps,
                                                                  bf16_t* __restrict__ y_bf16, int ld_y,
                                                                  float* __restrict__ y_f32,
                                                                  const float* __restrict__ lora_a,
                                                                  float* __restrict__ stats, DropCfg drop,
                                                                  bf16_t* __restrict__ t_aug = nullptr, int ld_t = 0,
                                                                  bf16_t* __restrict__ y3 = nullptr, int ld_y3 = 0) {
    // y3 (exact mode, round 5): the output as the split-bf16 GEMM operand [hi | lo | hi] (bf16 [M, 3H]) -- what a stand-alone
    // split3_rows pass over y_f32 used to produce
    constexpr int NV = H / 256;
    BSCLIP_DROP_RESOLVE(drop);
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * LN_BLOCK + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * LN_BLOCK) >> 6;

    f32x4 g[NV], b[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        g[j] = *reinterpret_cast<const f32x4*>(gamma + j * 256 + lane * 4);
        b[j] = *reinterpret_cast<const f32x4*>(beta + j * 256 + lane * 4);
    }
    // LoRA A [8, H] f32 lives in LDS (24 KiB per workgroup), read back as conflict-free 16-byte pieces for every row: in
    // registers it cost 96 VGPRs and held the kernel at two waves per SIMD (177 VGPRs), HBM latency exposed.
    __shared__ __attribute__((aligned(16))) float sA[LORA ? 8 * H : 4];
    if constexpr (LORA) {
        for (int i = threadIdx.x * 4; i < 8 * H; i += LN_BLOCK * 4)
            *reinterpret_cast<f32x4*>(sA + i) = *reinterpret_cast<const f32x4*>(lora_a + i);
        __syncthreads();
    }

    // The wave's next row is loaded before the current one is reduced (one row in flight per wave was latency-bound).
    f32x4 nxt[NV];
    if (wave < M) load_row<H, X_BF16>(x, ld_x, wave, lane, nxt);
    for (int row = wave; row < M; row += nwaves) {
        f32x4 v[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = nxt[j];
        if (row + nwaves < M) load_row<H, X_BF16>(x, ld_x, row + nwaves, lane, nxt);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        const float mean = wave_sum(s) * (1.0f / H);
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            v[j] -= mean;
            ss += (v[j][0] * v[j][0] + v[j][1] * v[j][1]) + (v[j][2] * v[j][2] + v[j][3] * v[j][3]);
        }
        const float var = wave_sum(ss) * (1.0f / H);
        const float rstd = rsqrtf(var + eps);
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = v[j] * rstd * g[j] + b[j];
        if (drop.thr16) {  // HF BertEmbeddings: dropout(LayerNorm(.)); everything downstream sees the dropped values
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j] = drop4(drop, (unsigned)row * H + j * 256 + lane * 4, v[j]);
        }

        if (stats && lane == 0) {
            stats[2 * (size_t)row] = mean;
            stats[2 * (size_t)row + 1] = rstd;
        }
        if (y_f32) {
#pragma unroll
            for (int j = 0; j < NV; ++j)
                st_stream(y_f32 + (size_t)row * H + j * 256 + lane * 4, v[j]);
        }
        if (y3) {
            bf16_t* r3 = y3 + (size_t)row * ld_y3 + lane * 4;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                uint2 hi, lo;
                hi.x = pack_bf2(v[j][0], v[j][1]);
                hi.y = pack_bf2(v[j][2], v[j][3]);
                lo.x = pack_bf2(v[j][0] - bf2f(hi.x & 0xffff), v[j][1] - bf2f(hi.x >> 16));
                lo.y = pack_bf2(v[j][2] - bf2f(hi.y & 0xffff), v[j][3] - bf2f(hi.y >> 16));
                st_stream(r3 + j * 256, hi);
                st_stream(r3 + H + j * 256, lo);
                st_stream(r3 + 2 * H + j * 256, hi);
            }
        }
        if (y_bf16) {
            bf16_t* yr = y_bf16 + (size_t)row * ld_y;
            if constexpr (Y_FP8) {
                unsigned char* y8 = reinterpret_cast<unsigned char*>(y_bf16) + (size_t)row * ld_y;
#pragma unroll
                for (int j = 0; j < NV; ++j)
                    st_stream(y8 + j * 256 + lane * 4, pack_fp8x4(v[j][0], v[j][1], v[j][2], v[j][3]));
                yr = t_aug + (size_t)row * ld_t - H;   // so that yr[H + lane] below is t_aug[row][lane]
            } else {
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    uint2 o;
                    o.x = pack_bf2(v[j][0], v[j][1]);
                    o.y = pack_bf2(v[j][2], v[j][3]);
                    st_stream(yr + j * 256 + lane * 4, o);
                }
            }
            if constexpr (LORA) {
                float p[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    float d = 0.f;
#pragma unroll
                    for (int j = 0; j < NV; ++j) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(sA + r * H + j * 256 + lane * 4);
                        d += (v[j][0] * a[0] + v[j][1] * a[1]) + (v[j][2] * a[2] + v[j][3] * a[3]);
                    }
                    p[r] = d;
                }
                const float t = reduce8(p, lane);  // lane group (bits 5,4,3) owns r
                // lane l in [0,8): fetch t[l] from lane with bits (5,4,3) = (l>>2&1, l>>1&1, l&1)
                const int src = ((lane >> 2) & 1) * 32 + ((lane >> 1) & 1) * 16 + (lane & 1) * 8;
                const float tl = __shfl(t, src, 64);
                // K-augmentation block: cols [H, H+8) = t, [H+8, H+64) = 0
                yr[H + lane] = (lane < 8) ? f2bf(tl) : (bf16_t)0;
            }
        }
    }
}

template <int H, bool X_BF16, bool LORA>
__global__ __launch_bounds__(LN_BLOCK) void layernorm_bwd_kernel(const void* __restrict__ x, int ld_x,
                                                                  const float* __restrict__ stats,
                                                                  const float* __restrict__ gamma, int M,
                                                                  const void* __restrict__ g_resid, int ld_gr,
                                                                  const void* __restrict__ g_gemm, int ld_g,
                                                                  const float* __restrict__ dt,
                                                                  const float* __restrict__ lora_a, int mode,
                                                                  void* __restrict__ dx_res, int ld_dx,
                                                                  void* __restrict__ dx_bf16, int ld_dxb,
                                                                  DropCfg drop, DropCfg in_drop, int resid_flags) {
    // resid_flags: bit 0 = g_resid is bf16, bit 1 = dx_res is bf16 (the residual-gradient stream kept in bf16: 2 instead of
    // 4 bytes per element read and written by every LayerNorm backward -- the kernel is HBM-bound)
    // bit 2 = g_gemm is f32, bit 3 = the operand output ("dx_bf16") is f32: the exact backward (exact.hip) keeps both in f32
    // bit 4 (exact mode, round 5) = the operand output is the split-bf16 GEMM operand [hi | lo | hi] (bf16 [M, ld_dxb >= 3H]) instead of
    //         the f32 tensor a stand-alone split3_rows pass would read
    const bool gr_bf16 = resid_flags & 1, dr_bf16 = resid_flags & 2, gg_f32 = resid_flags & 4, op_f32 = resid_flags & 8, op_s3 = resid_flags & 16;
    BSCLIP_DROP_RESOLVE(drop);
    BSCLIP_DROP_RESOLVE(in_drop);
    constexpr int NV = H / 256;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * LN_BLOCK + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * LN_BLOCK) >> 6;

    f32x4 g[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) g[j] = *reinterpret_cast<const f32x4*>(gamma + j * 256 + lane * 4);
    __shared__ __attribute__((aligned(16))) float sA[LORA ? 8 * H : 4];   // LoRA A in LDS, as in the forward kernel
    if constexpr (LORA) {
        for (int i = threadIdx.x * 4; i < 8 * H; i += LN_BLOCK * 4)
            *reinterpret_cast<f32x4*>(sA + i) = *reinterpret_cast<const f32x4*>(lora_a + i);
        __syncthreads();
    }

    for (int row = wave; row < M; row += nwaves) {
        if constexpr (LORA) asm volatile("" ::: "memory");   // sA is loop-invariant: keep its 24 reads out of the registers
        f32x4 v[NV], dy[NV], res[NV];
        load_row<H, X_BF16>(x, ld_x, row, lane, v);
        const float mean = stats[2 * (size_t)row], rstd = stats[2 * (size_t)row + 1];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            dy[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            res[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (g_resid) {
            if (gr_bf16) {
                const bf16_t* p = static_cast<const bf16_t*>(g_resid) + (size_t)row * ld_gr;
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const uint2 u = ld_stream(reinterpret_cast<const uint2*>(p + j * 256 + lane * 4));
                    res[j] = f32x4{bf2f(u.x & 0xffff), bf2f(u.x >> 16), bf2f(u.y & 0xffff), bf2f(u.y >> 16)};
                }
            } else {
#pragma unroll
                for (int j = 0; j < NV; ++j)
                    res[j] = ld_stream(reinterpret_cast<const f32x4*>(static_cast<const float*>(g_resid) + (size_t)row * ld_gr + j * 256 + lane * 4));
            }
        }
        if (g_gemm && gg_f32) {
#pragma unroll
            for (int j = 0; j < NV; ++j)
                dy[j] = ld_stream(reinterpret_cast<const f32x4*>(static_cast<const float*>(g_gemm) + (size_t)row * ld_g + j * 256 + lane * 4));
        } else if (g_gemm) {
            const bf16_t* p = static_cast<const bf16_t*>(g_gemm) + (size_t)row * ld_g;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const uint2 u = ld_stream(reinterpret_cast<const uint2*>(p + j * 256 + lane * 4));
                dy[j] = f32x4{bf2f(u.x & 0xffff), bf2f(u.x >> 16), bf2f(u.y & 0xffff), bf2f(u.y >> 16)};
            }
        }
        if constexpr (LORA) {
            // dy += dt[row, 0:8] . A   (gradient of the LoRA-A projection folded into this LN's output)
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(dt + (size_t)row * 8);
            const f32x4 d1 = *reinterpret_cast<const f32x4*>(dt + (size_t)row * 8 + 4);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                auto A = [&](int r) { return *reinterpret_cast<const f32x4*>(sA + r * H + j * 256 + lane * 4); };
                dy[j] += d0[0] * A(0) + d0[1] * A(1) + d0[2] * A(2) + d0[3] * A(3);
                dy[j] += d1[0] * A(4) + d1[1] * A(5) + d1[2] * A(6) + d1[3] * A(7);
            }
        }
        if (mode == 1) {
#pragma unroll
            for (int j = 0; j < NV; ++j) dy[j] += res[j];
        }
        if (in_drop.thr16) {  // this LN's OUTPUT was dropped in forward (BertEmbeddings): the same mask on its gradient
#pragma unroll
            for (int j = 0; j < NV; ++j) dy[j] = drop4(in_drop, (unsigned)row * H + j * 256 + lane * 4, dy[j]);
        }
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            v[j] = (v[j] - mean) * rstd;  // xhat
            dy[j] *= g[j];                // dL/dxhat
            c1 += (dy[j][0] + dy[j][1]) + (dy[j][2] + dy[j][3]);
            c2 += (dy[j][0] * v[j][0] + dy[j][1] * v[j][1]) + (dy[j][2] * v[j][2] + dy[j][3] * v[j][3]);
        }
        c1 = wave_sum(c1) * (1.0f / H);
        c2 = wave_sum(c2) * (1.0f / H);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            f32x4 d = (dy[j] - c1 - v[j] * c2) * rstd;
            if (mode == 0) d += res[j];
            if (dx_res) {
                if (dr_bf16) {
                    uint2 o;
                    o.x = pack_bf2(d[0], d[1]);
                    o.y = pack_bf2(d[2], d[3]);
                    st_stream(static_cast<bf16_t*>(dx_res) + (size_t)row * ld_dx + j * 256 + lane * 4, o);
                } else {
                    st_stream(static_cast<float*>(dx_res) + (size_t)row * ld_dx + j * 256 + lane * 4, d);
                }
            }
            if (dx_bf16) {
                // gradient w.r.t. the output of the Linear whose forward result was dropped with this (p, seed)
                if (drop.thr16) d = drop4(drop, (unsigned)row * H + j * 256 + lane * 4, d);
                if (op_s3) {
                    bf16_t* r3 = static_cast<bf16_t*>(dx_bf16) + (size_t)row * ld_dxb + j * 256 + lane * 4;
                    uint2 hi, lo;
                    hi.x = pack_bf2(d[0], d[1]);
                    hi.y = pack_bf2(d[2], d[3]);
                    lo.x = pack_bf2(d[0] - bf2f(hi.x & 0xffff), d[1] - bf2f(hi.x >> 16));
                    lo.y = pack_bf2(d[2] - bf2f(hi.y & 0xffff), d[3] - bf2f(hi.y >> 16));
                    st_stream(r3, hi);
                    st_stream(r3 + H, lo);
                    st_stream(r3 + 2 * H, hi);
                    continue;
                }
                if (op_f32) {
                    st_stream(static_cast<float*>(dx_bf16) + (size_t)row * ld_dxb + j * 256 + lane * 4, d);
                    continue;
                }
                uint2 o;
                o.x = pack_bf2(d[0], d[1]);
                o.y = pack_bf2(d[2], d[3]);
                st_stream(static_cast<bf16_t*>(dx_bf16) + (size_t)row * ld_dxb + j * 256 + lane * 4, o);
            }
        }
    }
}

// F.normalize(p=2, dim=-1, eps=1e-12): y = x / max(||x||, eps).  D = 768, one wave per row.
__global__ __launch_bounds__(LN_BLOCK) void l2norm_fwd_kernel(const float* __restrict__ x, int M, int D,
                                                               float* __restrict__ y, float* __restrict__ inv_norm) {
    const int lane = threadIdx.x & 63;
    const int row = (blockIdx.x * LN_BLOCK + threadIdx.x) >> 6;
    if (row >= M) return;
    const float* xr = x + (size_t)row * D;
    float ss = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
        ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    ss = wave_sum(ss);
    const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
    if (inv_norm && lane == 0) inv_norm[row] = inv;
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
        *reinterpret_cast<f32x4*>(y + (size_t)row * D + c) = v * inv;
    }
}

// dx = inv * (dy - y (y . dy))   (exact for ||x|| > eps, which holds for every non-degenerate embedding)
__global__ __launch_bounds__(LN_BLOCK) void l2norm_bwd_kernel(const float* __restrict__ y,
                                                               const float* __restrict__ inv_norm,
                                                               const float* __restrict__ dy, int M, int D,
                                                               float* __restrict__ dx) {
    const int lane = threadIdx.x & 63;
    const int row = (blockIdx.x * LN_BLOCK + threadIdx.x) >> 6;
    if (row >= M) return;
    const float* yr = y + (size_t)row * D;
    const float* gr = dy + (size_t)row * D;
    float dot = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(yr + c);
        const f32x4 g = *reinterpret_cast<const f32x4*>(gr + c);
        dot += (a[0] * g[0] + a[1] * g[1]) + (a[2] * g[2] + a[3] * g[3]);
    }
    dot = wave_sum(dot);
    const float inv = inv_norm[row];
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(yr + c);
        const f32x4 g = *reinterpret_cast<const f32x4*>(gr + c);
        *reinterpret_cast<f32x4*>(dx + (size_t)row * D + c) = (g - a * dot) * inv;
    }
}

int ln_grid(int M, int resident_per_cu = 8) {
    const int blocks = ceil_div(M, LN_BLOCK / 64);
    const int cap = 256 * resident_per_cu;   // grid-stride beyond what is resident at once (LoRA variants: 24 KiB of LDS each)
    return blocks < cap ? blocks : cap;
}

}  // namespace

#define LN_FWD_LAUNCH(HH, XB, LO)                                                                              \
    hipLaunchKernelGGL((layernorm_fwd_kernel<HH, XB, LO>), dim3(ln_grid(M, LO ? 5 : 8)), dim3(LN_BLOCK), 0, s, x, ld_x, M, \
                       gamma, beta, eps, static_cast<bf16_t*>(y_bf16), ld_y, y_f32, lora_a, stats, drop,               \
                       static_cast<bf16_t*>(nullptr), 0, static_cast<bf16_t*>(y_split3), ld_y3)

extern "C" int bsclip_layernorm_fwd(const void* x, int ld_x, int x_bf16, int M, int H, const float* gamma,
                                    const float* beta, float eps, void* y_bf16, int ld_y, float* y_f32, void* y_split3, int ld_y3,
                                    const float* lora_a, float* stats, float dropout_p, uint32_t dropout_seed, void* stream) {
    BSCLIP_REQUIRE(x && gamma && beta && M > 0, "bsclip_layernorm_fwd: null/empty input");
    BSCLIP_REQUIRE(H == 768 || H == 512, "bsclip_layernorm_fwd: H=%d (supported: 768, 512)", H);
    BSCLIP_REQUIRE(ld_x >= H && ld_x % 4 == 0, "bsclip_layernorm_fwd: ld_x=%d", ld_x);
    BSCLIP_REQUIRE(!y_bf16 || (ld_y % 4 == 0 && ld_y >= H + (lora_a ? BSCLIP_KPAD : 0)),
                   "bsclip_layernorm_fwd: ld_y=%d too small for H=%d%s", ld_y, H, lora_a ? "+KPAD" : "");
    BSCLIP_REQUIRE(!lora_a || y_bf16, "bsclip_layernorm_fwd: lora_a needs y_bf16");
    BSCLIP_REQUIRE(!y_split3 || (ld_y3 >= 3 * H && ld_y3 % 4 == 0 && (reinterpret_cast<uintptr_t>(y_split3) & 7) == 0),
                   "bsclip_layernorm_fwd: y_split3 bf16 [M, ld_y3 >= 3H], 8-byte aligned (ld_y3=%d)", ld_y3);
    BSCLIP_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "bsclip_layernorm_fwd: dropout_p=%f", dropout_p);
    const DropCfg drop = make_drop(dropout_p, dropout_seed);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool lo = lora_a != nullptr;
    if (H == 768) {
        if (x_bf16) { if (lo) LN_FWD_LAUNCH(768, true, true); else LN_FWD_LAUNCH(768, true, false); }
        else        { if (lo) LN_FWD_LAUNCH(768, false, true); else LN_FWD_LAUNCH(768, false, false); }
    } else {
        if (x_bf16) { if (lo) LN_FWD_LAUNCH(512, true, true); else LN_FWD_LAUNCH(512, true, false); }
        else        { if (lo) LN_FWD_LAUNCH(512, false, true); else LN_FWD_LAUNCH(512, false, false); }
    }
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

#define LN_FWD8_LAUNCH(HH, XB, LO)                                                                                   \
    hipLaunchKernelGGL((layernorm_fwd_kernel<HH, XB, LO, true>), dim3(ln_grid(M, LO ? 5 : 8)), dim3(LN_BLOCK), 0, s, x, ld_x, M, gamma, \
                       beta, eps, static_cast<bf16_t*>(y_fp8), ld_y, y_f32, lora_a, stats, drop, static_cast<bf16_t*>(t_aug), ld_t)

extern "C" int bsclip_layernorm_fwd_fp8(const void* x, int ld_x, int x_bf16, int M, int H, const float* gamma,
                                        const float* beta, float eps, void* y_fp8, int ld_y, void* t_aug, int ld_t,
                                        float* y_f32, const float* lora_a, float* stats, float dropout_p,
                                        uint32_t dropout_seed, void* stream) {
    BSCLIP_REQUIRE(x && gamma && beta && y_fp8 && M > 0, "bsclip_layernorm_fwd_fp8: null/empty input");
    BSCLIP_REQUIRE(H == 768 || H == 512, "bsclip_layernorm_fwd_fp8: H=%d (supported: 768, 512)", H);
    BSCLIP_REQUIRE(ld_x >= H && ld_x % 4 == 0 && ld_y >= H && ld_y % 16 == 0, "bsclip_layernorm_fwd_fp8: ld_x=%d ld_y=%d", ld_x, ld_y);
    BSCLIP_REQUIRE((lora_a == nullptr) == (t_aug == nullptr) && (!t_aug || (ld_t >= BSCLIP_KPAD && ld_t % 8 == 0)),
                   "bsclip_layernorm_fwd_fp8: lora_a and t_aug go together, ld_t >= %d (ld_t=%d)", BSCLIP_KPAD, ld_t);
    BSCLIP_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "bsclip_layernorm_fwd_fp8: dropout_p=%f", dropout_p);
    const DropCfg drop = make_drop(dropout_p, dropout_seed);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool lo = lora_a != nullptr;
    if (H == 768) {
        if (x_bf16) { if (lo) LN_FWD8_LAUNCH(768, true, true); else LN_FWD8_LAUNCH(768, true, false); }
        else        { if (lo) LN_FWD8_LAUNCH(768, false, true); else LN_FWD8_LAUNCH(768, false, false); }
    } else {
        if (x_bf16) { if (lo) LN_FWD8_LAUNCH(512, true, true); else LN_FWD8_LAUNCH(512, true, false); }
        else        { if (lo) LN_FWD8_LAUNCH(512, false, true); else LN_FWD8_LAUNCH(512, false, false); }
    }
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

#define LN_BWD_LAUNCH(HH, XB, LO)                                                                                \
    hipLaunchKernelGGL((layernorm_bwd_kernel<HH, XB, LO>), dim3(ln_grid(M, LO ? 4 : 8)), dim3(LN_BLOCK), 0, s, x, ld_x, stats, \
                       gamma, M, g_resid, ld_gr, g_gemm, ld_g, dt, lora_a, mode, dx_f32, \
                       ld_dx, dx_bf16, ld_dxb, drop, in_drop, resid_flags)

extern "C" int bsclip_layernorm_bwd(const void* x, int ld_x, int x_bf16, const float* stats, const float* gamma, int M,
                                    int H, const void* g_resid, int ld_gr, const void* g_gemm, int ld_g,
                                    const float* dt, const float* lora_a, int mode, void* dx_f32, int ld_dx,
                                    void* dx_bf16, int ld_dxb, float dropout_p, uint32_t dropout_seed, float in_dropout_p,
                                    uint32_t in_dropout_seed, int resid_flags, void* stream) {
    BSCLIP_REQUIRE(x && stats && gamma && M > 0, "bsclip_layernorm_bwd: null/empty input");
    BSCLIP_REQUIRE(H == 768 || H == 512, "bsclip_layernorm_bwd: H=%d (supported: 768, 512)", H);
    BSCLIP_REQUIRE(g_resid || g_gemm, "bsclip_layernorm_bwd: no incoming gradient");
    BSCLIP_REQUIRE(mode == 0 || mode == 1, "bsclip_layernorm_bwd: mode=%d", mode);
    BSCLIP_REQUIRE(resid_flags >= 0 && resid_flags <= 31 && !((resid_flags & 16) && (resid_flags & 8)), "bsclip_layernorm_bwd: resid_flags=%d", resid_flags);
    BSCLIP_REQUIRE((dt == nullptr) == (lora_a == nullptr), "bsclip_layernorm_bwd: dt and lora_a go together");
    BSCLIP_REQUIRE(!g_gemm || (ld_g >= H && ld_g % 4 == 0), "bsclip_layernorm_bwd: ld_g=%d", ld_g);
    BSCLIP_REQUIRE(!g_resid || (ld_gr >= H && ld_gr % 4 == 0), "bsclip_layernorm_bwd: ld_gr=%d", ld_gr);
    BSCLIP_REQUIRE(!dx_f32 || (ld_dx >= H && ld_dx % 4 == 0), "bsclip_layernorm_bwd: ld_dx=%d", ld_dx);
    BSCLIP_REQUIRE(!dx_bf16 || (ld_dxb >= ((resid_flags & 16) ? 3 * H : H) && ld_dxb % 4 == 0), "bsclip_layernorm_bwd: ld_dxb=%d", ld_dxb);
    BSCLIP_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "bsclip_layernorm_bwd: dropout_p=%f", dropout_p);
    const DropCfg drop = make_drop(dropout_p, dropout_seed);
    BSCLIP_REQUIRE(in_dropout_p >= 0.f && in_dropout_p < 1.f, "bsclip_layernorm_bwd: in_dropout_p=%f", in_dropout_p);
    const DropCfg in_drop = make_drop(in_dropout_p, in_dropout_seed);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool lo = lora_a != nullptr;
    if (H == 768) {
        if (x_bf16) { if (lo) LN_BWD_LAUNCH(768, true, true); else LN_BWD_LAUNCH(768, true, false); }
        else        { if (lo) LN_BWD_LAUNCH(768, false, true); else LN_BWD_LAUNCH(768, false, false); }
    } else {
        if (x_bf16) { if (lo) LN_BWD_LAUNCH(512, true, true); else LN_BWD_LAUNCH(512, true, false); }
        else        { if (lo) LN_BWD_LAUNCH(512, false, true); else LN_BWD_LAUNCH(512, false, false); }
    }
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int bsclip_l2norm_fwd(const float* x, int M, int D, float* y, float* inv_norm, void* stream) {
    BSCLIP_REQUIRE(x && y && M > 0 && D % 4 == 0, "bsclip_l2norm_fwd: bad args (M=%d D=%d)", M, D);
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(ceil_div(M, LN_BLOCK / 64)), dim3(LN_BLOCK), 0,
                       static_cast<hipStream_t>(stream), x, M, D, y, inv_norm);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int bsclip_l2norm_bwd(const float* y, const float* inv_norm, const float* dy, int M, int D, float* dx,
                                 void* stream) {
    BSCLIP_REQUIRE(y && inv_norm && dy && dx && M > 0 && D % 4 == 0, "bsclip_l2norm_bwd: bad args (M=%d D=%d)", M, D);
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(ceil_div(M, LN_BLOCK / 64)), dim3(LN_BLOCK), 0,
                       static_cast<hipStream_t>(stream), y, inv_norm, dy, M, D, dx);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}
