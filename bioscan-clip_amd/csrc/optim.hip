// Trainable-parameter gradients that are not GEMM-shaped, and the optimiser (gfx950).
//   lora_grad : autograd of the rank-4 LoRA branches (image_encoder.py:44-47, dna_encoder.py:47-49) -- skinny
//               reductions over all M tokens, HBM-bound: one pass over dq/dv/t (dt + dB), one over y (dA).
//   colsum    : bias gradients of the trainable heads.
//   adamw     : torch.optim.AdamW defaults (scripts/train_cl.py:158), one launch over the flat trainable buffer.
#include <math.h>

#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ f32x4 ld_bf4(const bf16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return f32x4{bf2f(u.x & 0xffff), bf2f(u.x >> 16), bf2f(u.y & 0xffff), bf2f(u.y >> 16)};
}

// same 8-way transpose-reduce as norm.hip (lane group bits (5,4,3) ends up owning index r)
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

constexpr int LG_BLOCK = 256;

// One value in all four elements, in registers of its own (opaque to the compiler): the packed FMAs that follow take plain operands
// instead of broadcasting one half of a register pair through op_sel.
// Why (round 5, docs/HISTORY.md "the lost quarter-wave"): in the exact mode's three-tower step the BarcodeBERT tower's first
// bsclip_lora_grad_f32 of a backward -- and only that launch, the one that runs beside the other towers' first backward kernels --
// left, once in a few dozen steps, 16-80 words of its dA / dB slabs off by one row's contribution: always lanes 48-63, always the LOW
// half of a `v_pk_fma_f32 acc, y, d op_sel:[0,1,0]` whose multiplier is the HIGH register of a pair filled by a wave-uniform
// global_load_dwordx4 (dt[row, 0:8] / a row of dq), identical inputs, never reproducible with the kernel alone under synthetic load
// (tools/stress_lora_f32.py: 0 of 8 000).  tests/test_30's exact three-tower graph-vs-eager case failed in ~40 % of the suite runs;
// with the multipliers splatted first (this function) 120 of 120 iterations of tools/debug_graph_flake.py are bit-equal (before: a mismatch
// within 4 iterations in 9 of 10 runs).  The cause below the ISA is not established; the form that does not show it is kept.  (Also
// tried and NOT the cause: the cross-wave LDS sums as ds_add_f32 instead of ds_read2 / v_add / ds_write2 -- the flake stayed, and LDS
// float atomics cost the dA pass 111 instead of 23 us.)  The dt / dB pass pays for it (no packed FMAs left: 41 -> ~70 us); it only runs
// in the fp8 and exact configurations since the attention backward leaves those sums itself.
__device__ __forceinline__ f32x4 splat4(float v) {
    f32x4 s = {v, v, v, v};
    asm("" : "+v"(s));   // not volatile: as an ordering point between the row loads it cost the da kernel 111 instead of 23 us
    return s;
}

// 4-way transpose-reduce: lane bits (5,4) end up owning index r
__device__ __forceinline__ float reduce4(float (&p)[4], int lane) {
    float q[2];
    const bool hi5 = lane & 32;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float send = hi5 ? p[i] : p[i + 2];
        const float keep = hi5 ? p[i + 2] : p[i];
        q[i] = keep + __shfl_xor(send, 32, 64);
    }
    const bool hi4 = lane & 16;
    float v = (hi4 ? q[1] : q[0]) + __shfl_xor(hi4 ? q[0] : q[1], 16, 64);
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 1, 64);
    return v;
}

// pass 1: dt[M,8] = [dq.B_q | dv.B_v];  dBq[H,4] += dq^T t_q;  dBv[H,4] += dv^T t_v
// haug / ld_h: the bf16 block holding t -- the LN output row (t at column H, bf16 path) or the separate t_aug buffer
// (t at column 0, fp8 path; the caller passes haug = t_aug - H)
// The waves of a workgroup come in pairs: the even wave of a pair owns the q half of a row (dq, B_q, t_q), the odd wave the v
// half.  With both halves in one wave the kernel held 2 x (B + accumulators) = 192 VGPRs plus the rows in flight and ran at two
// waves per SIMD, HBM latency exposed (59 us for 155 MB); a half is 96 + rows, three waves per SIMD.
// F32IN (exact mode, round 5): dqkv is f32 [M, ld] and t comes from an f32 [M, 8] buffer (haug = that buffer, ld_h = 8) -- the same
// kernel on the exact backward's operands; the first f32 version (one workgroup per row group, 16 wave reductions per row, two
// barriers per four rows) took 640 us per layer at B = 256 against this kernel's 41 us on bf16 rows.
template <int H, bool F32IN = false>
__global__ __launch_bounds__(LG_BLOCK) void lora_grad_dt_db_kernel(const void* __restrict__ dqkv_, int ld,
                                                                    const void* __restrict__ haug_, int ld_h, int M,
                                                                    const float* __restrict__ lora_b,
                                                                    float* __restrict__ dt,
                                                                    float* __restrict__ partial) {
    const bf16_t* dqkv = static_cast<const bf16_t*>(dqkv_);
    const bf16_t* haug = static_cast<const bf16_t*>(haug_);
    constexpr int NV = H / 256;
    __shared__ float red[2 * NV * 16 * 64];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int is_v = wib & 1;
    const int pair = blockIdx.x * (LG_BLOCK / 128) + (wib >> 1);
    const int npairs = gridDim.x * (LG_BLOCK / 128);
    for (int i = threadIdx.x; i < 2 * NV * 16 * 64; i += LG_BLOCK) red[i] = 0.f;

    f32x4 bx[NV][4];  // [chunk][column-in-chunk] -> 4 ranks of this half's B
    f32x4 ax[NV][4];  // accumulators, same indexing
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = j * 256 + lane * 4 + i;
            bx[j][i] = *reinterpret_cast<const f32x4*>(lora_b + (size_t)(is_v * H + c) * 4);
            ax[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

    // Rows are software-pipelined three deep: a wave keeps the loads of its next PF half-rows in flight while it reduces one.
    constexpr int PF = 3;
    using row_t = std::conditional_t<F32IN, f32x4, uint2>;
    row_t rx[PF][NV];
    row_t rt[PF];
    auto fetch = [&](int slot, int row) {
        if constexpr (F32IN) {
            const float* g = static_cast<const float*>(dqkv_) + (size_t)row * ld + is_v * 2 * H;
#pragma unroll
            for (int j = 0; j < NV; ++j) rx[slot][j] = *reinterpret_cast<const f32x4*>(g + j * 256 + lane * 4);
            rt[slot] = *reinterpret_cast<const f32x4*>(static_cast<const float*>(haug_) + (size_t)row * ld_h + 4 * is_v);
        } else {
            const bf16_t* g = dqkv + (size_t)row * ld + is_v * 2 * H;
#pragma unroll
            for (int j = 0; j < NV; ++j) rx[slot][j] = *reinterpret_cast<const uint2*>(g + j * 256 + lane * 4);
            rt[slot] = *reinterpret_cast<const uint2*>(haug + (size_t)row * ld_h + H + 4 * is_v);  // this half's t (4), broadcast
        }
    };
#pragma unroll
    for (int p = 0; p < PF; ++p)
        if (pair + p * npairs < M) fetch(p, pair + p * npairs);
    auto body = [&](int slot, int row) {
        f32x4 dx[NV];
        f32x4 tx;
        if constexpr (F32IN) {
#pragma unroll
            for (int j = 0; j < NV; ++j) dx[j] = rx[slot][j];
            tx = rt[slot];
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j)
                dx[j] = f32x4{bf2f(rx[slot][j].x & 0xffff), bf2f(rx[slot][j].x >> 16), bf2f(rx[slot][j].y & 0xffff),
                              bf2f(rx[slot][j].y >> 16)};
            const uint2 tu = rt[slot];
            tx = f32x4{bf2f(tu.x & 0xffff), bf2f(tu.x >> 16), bf2f(tu.y & 0xffff), bf2f(tu.y >> 16)};
        }
        if (row + PF * npairs < M) fetch(slot, row + PF * npairs);
        f32x4 px = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 d4 = splat4(dx[j][i]);
                px += d4 * bx[j][i];
                ax[j][i] += d4 * tx;
            }
        float p[4] = {px[0], px[1], px[2], px[3]};
        const float tot = reduce4(p, lane);
        const float tl = __shfl(tot, ((lane >> 1) & 1) * 32 + (lane & 1) * 16, 64);
        if (lane < 4) dt[(size_t)row * 8 + 4 * is_v + lane] = tl;
    };
    for (int row = pair; row < M; row += PF * npairs) {  // slots are compile-time indices: registers, not scratch
#pragma unroll
        for (int p = 0; p < PF; ++p)
            if (row + p * npairs < M) body(p, row + p * npairs);
    }
    // cross-wave sum in wave order (no LDS atomics: the order of float adds is fixed): waves 0, 2 into the q region, 1, 3 into v
    float* mine = red + is_v * (NV * 16 * 64);
    for (int w = 0; w < LG_BLOCK / 64; ++w) {
        __syncthreads();
        if (wib == w) {
#pragma unroll
            for (int j = 0; j < NV; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mine[((j * 16 + i * 4 + r) * 64) + lane] += ax[j][i][r];
        }
    }
    __syncthreads();
    // per-workgroup partial slab (plain coalesced stores; summed in a fixed order by lora_grad_reduce_kernel)
    for (int idx = threadIdx.x; idx < 2 * NV * 16 * 64; idx += LG_BLOCK)
        partial[(size_t)blockIdx.x * (2 * NV * 16 * 64) + idx] = red[idx];
}

// pass 2: dA[8,H] += dt^T y.  Y_FP8: y is the fp8 e4m3 LN output (row stride ld_h bytes) instead of bf16.
template <int H, bool Y_FP8 = false, bool Y_F32 = false>
__global__ __launch_bounds__(LG_BLOCK) void lora_grad_da_kernel(const void* __restrict__ haug_, int ld_h, int M,
                                                                 const float* __restrict__ dt,
                                                                 float* __restrict__ partial) {
    const bf16_t* haug = static_cast<const bf16_t*>(haug_);
    constexpr int NV = H / 256;
    __shared__ float red[8 * NV * 4 * 64];
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * LG_BLOCK + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * LG_BLOCK) >> 6;
    for (int i = threadIdx.x; i < 8 * NV * 4 * 64; i += LG_BLOCK) red[i] = 0.f;
    f32x4 acc[8][NV];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int j = 0; j < NV; ++j) acc[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // three rows per iteration: all their loads are issued before the first is consumed
    constexpr int PF = 3;
    for (int row0 = wave; row0 < M; row0 += PF * nwaves) {
        f32x4 d0[PF], d1[PF];
        std::conditional_t<Y_F32, f32x4, uint2> yr[PF][NV];
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int row = min(row0 + p * nwaves, M - 1);
            d0[p] = *reinterpret_cast<const f32x4*>(dt + (size_t)row * 8);
            d1[p] = *reinterpret_cast<const f32x4*>(dt + (size_t)row * 8 + 4);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                if constexpr (Y_F32) {
                    yr[p][j] = *reinterpret_cast<const f32x4*>(static_cast<const float*>(haug_) + (size_t)row * ld_h + j * 256 + lane * 4);
                } else if constexpr (Y_FP8) {
                    yr[p][j].x = *reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned char*>(haug) +
                                                                    (size_t)row * ld_h + j * 256 + lane * 4);
                    yr[p][j].y = 0;
                } else {
                    yr[p][j] = *reinterpret_cast<const uint2*>(haug + (size_t)row * ld_h + j * 256 + lane * 4);
                }
            }
        }
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            if (row0 + p * nwaves >= M) break;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                f32x4 y;
                if constexpr (Y_F32)
                    y = yr[p][j];
                else if constexpr (Y_FP8)
                    y = f32x4{fp8_to_f32(yr[p][j].x, 0), fp8_to_f32(yr[p][j].x, 1), fp8_to_f32(yr[p][j].x, 2), fp8_to_f32(yr[p][j].x, 3)};
                else
                    y = f32x4{bf2f(yr[p][j].x & 0xffff), bf2f(yr[p][j].x >> 16), bf2f(yr[p][j].y & 0xffff), bf2f(yr[p][j].y >> 16)};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[r][j] += splat4(d0[p][r]) * y;
                    acc[4 + r][j] += splat4(d1[p][r]) * y;
                }
            }
        }
    }
    const int wib = threadIdx.x >> 6;
    for (int w = 0; w < LG_BLOCK / 64; ++w) {
        __syncthreads();
        if (wib == w) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int j = 0; j < NV; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) red[(((r * NV + j) * 4 + i) * 64) + lane] += acc[r][j][i];
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 8 * NV * 4 * 64; idx += LG_BLOCK)
        partial[(size_t)blockIdx.x * (8 * NV * 4 * 64) + idx] = red[idx];
}

// t[row, 0:8] = y[row, :] . A[0:8, :]^T in f32 (exact mode: the forward folds W + B A into the weight, so t is rebuilt for the backward):
// one wave per row, A in LDS as in the LayerNorm kernels
template <int H>
__global__ __launch_bounds__(LG_BLOCK) void lora_t_f32_kernel(const float* __restrict__ y, int ld_y, int M, const float* __restrict__ A,
                                                              float* __restrict__ t) {
    constexpr int NV = H / 256;
    __shared__ __attribute__((aligned(16))) float sA[8 * H];
    for (int i = threadIdx.x * 4; i < 8 * H; i += LG_BLOCK * 4) *reinterpret_cast<f32x4*>(sA + i) = *reinterpret_cast<const f32x4*>(A + i);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * LG_BLOCK + threadIdx.x) >> 6, nwaves = (gridDim.x * LG_BLOCK) >> 6;
    for (int row = wave; row < M; row += nwaves) {
        asm volatile("" ::: "memory");   // sA is loop-invariant: keep its reads out of the registers
        f32x4 v[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = *reinterpret_cast<const f32x4*>(y + (size_t)row * ld_y + j * 256 + lane * 4);
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
        const float tt = reduce8(p, lane);  // lane group (bits 5,4,3) owns r
        const float tl = __shfl(tt, ((lane >> 2) & 1) * 32 + ((lane >> 1) & 1) * 16 + (lane & 1) * 8, 64);
        if (lane < 8) t[(size_t)row * 8 + lane] = tl;
    }
}

// Sum the per-workgroup slabs in a fixed order (bitwise reproducible) and accumulate into dBq/dBv/dA.
// Slab A (8H floats): index k*64 + l, k = [is_v][j][i][r];  slab B (8H floats): index ((r*NV + j)*4 + i)*64 + l.
// Workgroup = 32 consecutive slab elements x 32 slab groups: thread (q, p) sums 4 consecutive elements (one 16-byte load per slab) of
// slabs p, p + 32, ... with four loads in flight; the 32 partial sums are combined in p order through LDS.  (Round 5: 4-byte loads
// over 8 groups of 96 slabs took 13 us for the step's 19 MB of dA slabs.)
// first_block = SLAB / 32 with a grid of SLAB / 32 workgroups sums the dA slabs only (bsclip_lora_grad_heads: dB comes from the
// attention kernel's partials).
template <int H>
__global__ __launch_bounds__(256) void lora_grad_reduce_kernel(const float* __restrict__ pa,
                                                                const float* __restrict__ pb, int nblocks,
                                                                float* __restrict__ dA, float* __restrict__ dBq,
                                                                float* __restrict__ dBv, int first_block = 0) {
    constexpr int NV = H / 256;
    constexpr int SLAB = 8 * H;
    __shared__ f32x4 red[32][8];
    const int q = threadIdx.x & 7, p = threadIdx.x >> 3;
    const int idx = (blockIdx.x + first_block) * 32 + q * 4;  // < 2*SLAB by construction (grid = 2*SLAB/32)
    const bool second = idx >= SLAB;
    const int e = second ? idx - SLAB : idx;
    const float* src = (second ? pb : pa) + e;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int b = p;
    for (; b + 96 < nblocks; b += 128) {   // four loads in flight; the adds keep the slab order
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (size_t)(b + 32 * u) * SLAB);
#pragma unroll
        for (int u = 0; u < 4; ++u) s += v[u];
    }
    for (; b < nblocks; b += 32) s += *reinterpret_cast<const f32x4*>(src + (size_t)b * SLAB);
    red[p][q] = s;
    __syncthreads();
    if (p != 0) return;
    s = red[0][q];
#pragma unroll
    for (int g = 1; g < 32; ++g) s += red[g][q];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int ec = e + c;
        const int l = ec & 63, k = ec >> 6;
        if (!second) {
            const bool is_v = k >= NV * 16;
            const int kk = is_v ? k - NV * 16 : k;
            const int j = kk >> 4, ii = (kk >> 2) & 3, r = kk & 3;
            const int col = j * 256 + l * 4 + ii;
            float* dst = (is_v ? dBv : dBq) + (size_t)col * 4 + r;
            *dst += s[c];
        } else {
            const int ii = k & 3, j = (k >> 2) % NV, r = (k >> 2) / NV;
            dA[(size_t)r * H + j * 256 + l * 4 + ii] += s[c];
        }
    }
}

// Sums what bsclip_attn_bwd_lora left (attn.hip LoraPart), in a fixed order:
//   workgroups [0, nA):  dt[row][4 half : 4 half + 4] = sum_head dt_partial[head][half][row][0:4]   (thread = (row, half); 12 loads in flight)
//   the rest:            dB[q | v][head * 64 + d][j] += sum_item db_partial[item][q | v][j][d]   (as lora_grad_reduce_kernel: 32
//                        consecutive elements x 8 item groups per workgroup, combined in group order through LDS)
__global__ __launch_bounds__(256) void lora_heads_reduce_kernel(const float* __restrict__ dtp, const float* __restrict__ dbp, int M,
                                                                 int heads, int items, int nA, float* __restrict__ dt,
                                                                 float* __restrict__ dBq, float* __restrict__ dBv) {
    if ((int)blockIdx.x < nA) {
        const int g = blockIdx.x * 256 + threadIdx.x, row = g >> 1, half = g & 1;
        if (row >= M) return;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int hd = 0; hd < heads; hd += 4) {   // heads is a multiple of 4 (8 or 12)
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(dtp + (((size_t)(hd + u) * 2 + half) * M + row) * 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) s += v[u];
        }
        *reinterpret_cast<f32x4*>(dt + (size_t)row * 8 + 4 * half) = s;
        return;
    }
    __shared__ float red[8][32];
    const int i = threadIdx.x & 31, p = threadIdx.x >> 5;
    const int e = ((int)blockIdx.x - nA) * 32 + i;   // < heads * 512 by construction
    const size_t stride = (size_t)heads * 512;
    const float* src = dbp + e;
    float s = 0.f;
    int b = p;
    for (; b + 56 < items; b += 64) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(b + 8 * u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < items; b += 8) s += src[(size_t)b * stride];
    red[p][i] = s;
    __syncthreads();
    if (p != 0) return;
    s = ((red[0][i] + red[1][i]) + (red[2][i] + red[3][i])) + ((red[4][i] + red[5][i]) + (red[6][i] + red[7][i]));
    const int hd = e >> 9, is_v = (e >> 8) & 1, j = (e >> 6) & 3, d = e & 63;
    float* dst = (is_v ? dBv : dBq) + (size_t)(hd * 64 + d) * 4 + j;
    *dst += s;
}

// out[n] += sum_r g[r, n] without atomics (bitwise reproducible) and without a workspace.
// Wide version (N % 8 == 0): one workgroup per 8 columns, thread t sums rows t, t+256, ... with one 16-B (bf16) / 2x16-B
// (f32) load per row, 4 rows in flight; the 256 partials are combined in a fixed order through LDS.  N/8 workgroups (96
// for N = 768) keep enough CUs busy for the one large caller, the MLM decoder bias gradient ([B*133, 768] bf16).
template <bool BF16>
__global__ __launch_bounds__(256) void colsum8_kernel(const void* __restrict__ g, int ld, int M, int N,
                                                       float* __restrict__ out) {
    __shared__ float part[256][9];
    __shared__ float part2[8][8];
    const int t = threadIdx.x;
    const int n0 = blockIdx.x * 8;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto add_row = [&](int r) {
        if constexpr (BF16) {
            const u32x4 u = *reinterpret_cast<const u32x4*>(static_cast<const bf16_t*>(g) + (size_t)r * ld + n0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s[2 * k] += bf2f(u[k] & 0xffff);
                s[2 * k + 1] += bf2f(u[k] >> 16);
            }
        } else {
            const float* p = static_cast<const float*>(g) + (size_t)r * ld + n0;
            const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s[k] += a[k];
                s[4 + k] += b[k];
            }
        }
    };
    int r = t;
    for (; r + 768 < M; r += 1024) {
        add_row(r);
        add_row(r + 256);
        add_row(r + 512);
        add_row(r + 768);
    }
    for (; r < M; r += 256) add_row(r);
#pragma unroll
    for (int k = 0; k < 8; ++k) part[t][k] = s[k];
    __syncthreads();
    if (t < 64) {  // 8 columns x 8 row-octets: thread (c, o) sums partials 32*o .. 32*o+31 in order
        const int c = t & 7, o = t >> 3;
        float a = 0.f;
        for (int i = 0; i < 32; ++i) a += part[32 * o + i][c];
        part2[o][c] = a;
    }
    __syncthreads();
    if (t < 8) {
        float a = 0.f;
#pragma unroll
        for (int o = 0; o < 8; ++o) a += part2[o][t];
        out[n0 + t] += a;
    }
}

// Narrow fallback (any N): one workgroup per 32 columns, 8 row groups x 32 columns.
template <bool BF16>
__global__ __launch_bounds__(256) void colsum_kernel(const void* __restrict__ g, int ld, int M, int N,
                                                      float* __restrict__ out) {
    __shared__ float part[8][32];
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + c;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
        auto at = [&](int r) -> float {
            if constexpr (BF16) return bf2f(static_cast<const bf16_t*>(g)[(size_t)r * ld + n]);
            else return static_cast<const float*>(g)[(size_t)r * ld + n];
        };
        int r = rg;
        for (; r + 24 < M; r += 32) {
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] += at(r + 8 * u);
        }
        for (; r < M; r += 8) s[0] += at(r);
    }
    part[rg][c] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    if (rg == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += part[k][c];
        out[n] += t;
    }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                     float* __restrict__ m, float* __restrict__ v, long n, float lr,
                                                     float beta1, float beta2, float eps, float wd, float inv_bc1,
                                                     float inv_sqrt_bc2, float grad_scale) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i] * grad_scale;
        float pi = p[i] * (1.0f - lr * wd);
        const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        pi -= (lr * inv_bc1) * (mi / denom);
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
    }
}

// lr and step come from device memory: hyper[0] = lr (f32), hyper[1] = step count (uint32 bits, advanced by a
// bsclip_counter_add node of the captured step, so no in-flight node ever reads host memory); the bias corrections are
// formed in f64 like the host form does
__global__ __launch_bounds__(256) void adamw_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, long n,
                                                         const float* __restrict__ hyper, float beta1, float beta2, float eps,
                                                         float wd, float grad_scale) {
    const float lr = hyper[0];
    const double step = (double)reinterpret_cast<const unsigned*>(hyper)[1];
    const float inv_bc1 = (float)(1.0 / (1.0 - pow((double)beta1, step)));
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)beta2, step)));
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i] * grad_scale;
        float pi = p[i] * (1.0f - lr * wd);
        const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        pi -= (lr * inv_bc1) * (mi / denom);
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
    }
}

}  // namespace

constexpr int LG_MAX_BLOCKS = 768;   // three 4-wave workgroups per CU

extern "C" int64_t bsclip_lora_grad_workspace_floats(int H) { return (int64_t)LG_MAX_BLOCKS * 16 * H; }

extern "C" int bsclip_lora_grad(const void* dqkv, int ld_dqkv, const void* h, int ld_h, int M, int H,
                                const float* lora_b, float* dt, float* dA, float* dBq, float* dBv, float* workspace,
                                void* stream) {
    BSCLIP_REQUIRE(dqkv && h && lora_b && dt && dA && dBq && dBv && workspace && M > 0,
                   "bsclip_lora_grad: null/empty input");
    BSCLIP_REQUIRE(H == 768 || H == 512, "bsclip_lora_grad: H=%d (supported: 768, 512)", H);
    BSCLIP_REQUIRE(ld_dqkv >= 3 * H && ld_dqkv % 4 == 0 && ld_h >= H + 8 && ld_h % 8 == 0,
                   "bsclip_lora_grad: ld_dqkv=%d ld_h=%d", ld_dqkv, ld_h);
    hipStream_t s = static_cast<hipStream_t>(stream);
    int blocks = ceil_div(M, 4 * 8);  // >= 8 rows per wave
    if (blocks > LG_MAX_BLOCKS) blocks = LG_MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    const bf16_t* g = static_cast<const bf16_t*>(dqkv);
    const bf16_t* hh = static_cast<const bf16_t*>(h);
    float* pa = workspace;
    float* pb = workspace + (size_t)LG_MAX_BLOCKS * 8 * H;
    if (H == 768) {
        hipLaunchKernelGGL((lora_grad_dt_db_kernel<768>), dim3(blocks), dim3(LG_BLOCK), 0, s, g, ld_dqkv, hh, ld_h, M,
                           lora_b, dt, pa);
        hipLaunchKernelGGL((lora_grad_da_kernel<768>), dim3(blocks), dim3(LG_BLOCK), 0, s, hh, ld_h, M, dt, pb);
        hipLaunchKernelGGL((lora_grad_reduce_kernel<768>), dim3(16 * 768 / 32), dim3(256), 0, s, pa, pb,
                           blocks, dA, dBq, dBv);
    } else {
        hipLaunchKernelGGL((lora_grad_dt_db_kernel<512>), dim3(blocks), dim3(LG_BLOCK), 0, s, g, ld_dqkv, hh, ld_h, M,
                           lora_b, dt, pa);
        hipLaunchKernelGGL((lora_grad_da_kernel<512>), dim3(blocks), dim3(LG_BLOCK), 0, s, hh, ld_h, M, dt, pb);
        hipLaunchKernelGGL((lora_grad_reduce_kernel<512>), dim3(16 * 512 / 32), dim3(256), 0, s, pa, pb,
                           blocks, dA, dBq, dBv);
    }
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

// bsclip_lora_grad when the attention backward already left dt / dB as partial sums (bsclip_attn_bwd_lora): reduce them, then the dA pass.
// dt_partial f32 [heads][2][M][4], db_partial f32 [B * heads][2][4][64] (B sequences of M / B tokens; item = sequence * heads + head).
extern "C" int bsclip_lora_grad_heads(const void* h, int ld_h, int M, int H, int B, const float* dt_partial,
                                      const float* db_partial, float* dt, float* dA, float* dBq, float* dBv, float* workspace,
                                      void* stream) {
    BSCLIP_REQUIRE(h && dt_partial && db_partial && dt && dA && dBq && dBv && workspace && M > 0 && B > 0 && M % B == 0,
                   "bsclip_lora_grad_heads: null/empty input (M=%d B=%d)", M, B);
    BSCLIP_REQUIRE(H == 768 || H == 512, "bsclip_lora_grad_heads: H=%d (supported: 768, 512)", H);
    BSCLIP_REQUIRE(ld_h >= H && ld_h % 8 == 0, "bsclip_lora_grad_heads: ld_h=%d", ld_h);
    BSCLIP_REQUIRE(((reinterpret_cast<uintptr_t>(dt_partial) | reinterpret_cast<uintptr_t>(dt)) & 15) == 0,
                   "bsclip_lora_grad_heads: dt_partial and dt must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int heads = H / 64;
    int blocks = ceil_div(M, 4 * 8);
    if (blocks > LG_MAX_BLOCKS) blocks = LG_MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    const bf16_t* hh = static_cast<const bf16_t*>(h);
    float* pb = workspace + (size_t)LG_MAX_BLOCKS * 8 * H;
    const int nA = ceil_div(2 * M, 256);
    hipLaunchKernelGGL(lora_heads_reduce_kernel, dim3(nA + heads * 512 / 32), dim3(256), 0, s, dt_partial, db_partial, M, heads, B, nA,
                       dt, dBq, dBv);
    if (H == 768) {
        hipLaunchKernelGGL((lora_grad_da_kernel<768>), dim3(blocks), dim3(LG_BLOCK), 0, s, hh, ld_h, M, dt, pb);
        hipLaunchKernelGGL((lora_grad_reduce_kernel<768>), dim3(8 * 768 / 32), dim3(256), 0, s, (const float*)nullptr, pb, blocks, dA,
                           dBq, dBv, 8 * 768 / 32);
    } else {
        hipLaunchKernelGGL((lora_grad_da_kernel<512>), dim3(blocks), dim3(LG_BLOCK), 0, s, hh, ld_h, M, dt, pb);
        hipLaunchKernelGGL((lora_grad_reduce_kernel<512>), dim3(8 * 512 / 32), dim3(256), 0, s, (const float*)nullptr, pb, blocks, dA,
                           dBq, dBv, 8 * 512 / 32);
    }
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

// exact mode (BSCLIP_PARITY=2): f32 dq / dv and the f32 LayerNorm output y; t = A y is rebuilt (the forward folds W + B A), then the
// kernels of bsclip_lora_grad on f32 rows.  workspace: [t: 8 M][dt: 8 M][slabs: LG_MAX_BLOCKS * 16 H] floats.
// Reference lora_layer.py:16-39 / image_encoder.py:44-47 / dna_encoder.py:47-49:
//   u_q = B_q^T dq, u_v = B_v^T dv;  dA_q += u_q y^T, dA_v += u_v y^T, dB_q += dq t_q^T, dB_v += dv t_v^T
extern "C" int64_t bsclip_lora_grad_f32_workspace_floats(int M, int H) {
    return M > 0 && H > 0 ? 16 * (int64_t)((M + 3) / 4 * 4) + (int64_t)LG_MAX_BLOCKS * 16 * H : -1;
}

extern "C" int bsclip_lora_grad_f32(const float* dqkv, int ld_dqkv, const float* y, int ld_y, int M, int H, const float* lora_a,
                                    const float* lora_b, float* dA, float* dB, float* workspace, void* stream) {
    BSCLIP_REQUIRE(dqkv && y && lora_a && lora_b && dA && dB && workspace, "bsclip_lora_grad_f32: null pointer");
    BSCLIP_REQUIRE(M > 0 && (H == 768 || H == 512) && ld_dqkv >= 3 * H && ld_dqkv % 4 == 0 && ld_y >= H && ld_y % 4 == 0,
                   "bsclip_lora_grad_f32: M=%d H=%d ld_dqkv=%d ld_y=%d", M, H, ld_dqkv, ld_y);
    BSCLIP_REQUIRE(((reinterpret_cast<uintptr_t>(dqkv) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(lora_a) |
                     reinterpret_cast<uintptr_t>(lora_b) | reinterpret_cast<uintptr_t>(workspace)) & 15) == 0,
                   "bsclip_lora_grad_f32: dqkv, y, lora_a, lora_b and workspace must be 16-byte aligned (vector loads)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int blocks = ceil_div(M, 4 * 8);
    if (blocks > LG_MAX_BLOCKS) blocks = LG_MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    const size_t Mp = (size_t)(M + 3) / 4 * 4;
    float* t = workspace;
    float* dt = workspace + 8 * Mp;
    float* pa = workspace + 16 * Mp;
    float* pb = pa + (size_t)LG_MAX_BLOCKS * 8 * H;
    float* dBq = dB;
    float* dBv = dB + (size_t)H * 4;
    if (H == 768) {
        hipLaunchKernelGGL((lora_t_f32_kernel<768>), dim3(blocks), dim3(LG_BLOCK), 0, s, y, ld_y, M, lora_a, t);
        hipLaunchKernelGGL((lora_grad_dt_db_kernel<768, true>), dim3(blocks), dim3(LG_BLOCK), 0, s, dqkv, ld_dqkv, t, 8, M, lora_b, dt, pa);
        hipLaunchKernelGGL((lora_grad_da_kernel<768, false, true>), dim3(blocks), dim3(LG_BLOCK), 0, s, y, ld_y, M, dt, pb);
        hipLaunchKernelGGL((lora_grad_reduce_kernel<768>), dim3(16 * 768 / 32), dim3(256), 0, s, pa, pb, blocks, dA, dBq, dBv);
    } else {
        hipLaunchKernelGGL((lora_t_f32_kernel<512>), dim3(blocks), dim3(LG_BLOCK), 0, s, y, ld_y, M, lora_a, t);
        hipLaunchKernelGGL((lora_grad_dt_db_kernel<512, true>), dim3(blocks), dim3(LG_BLOCK), 0, s, dqkv, ld_dqkv, t, 8, M, lora_b, dt, pa);
        hipLaunchKernelGGL((lora_grad_da_kernel<512, false, true>), dim3(blocks), dim3(LG_BLOCK), 0, s, y, ld_y, M, dt, pb);
        hipLaunchKernelGGL((lora_grad_reduce_kernel<512>), dim3(16 * 512 / 32), dim3(256), 0, s, pa, pb, blocks, dA, dBq, dBv);
    }
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

// fp8 operand path (BASELINE configs[4]): y is the fp8 LN output [M, ld_y bytes], t the separate bf16 block [M, ld_t]
extern "C" int bsclip_lora_grad_fp8(const void* dqkv, int ld_dqkv, const void* y_fp8, int ld_y, const void* t_aug, int ld_t,
                                    int M, int H, const float* lora_b, float* dt, float* dA, float* dBq, float* dBv,
                                    float* workspace, void* stream) {
    BSCLIP_REQUIRE(dqkv && y_fp8 && t_aug && lora_b && dt && dA && dBq && dBv && workspace && M > 0,
                   "bsclip_lora_grad_fp8: null/empty input");
    BSCLIP_REQUIRE(H == 768 || H == 512, "bsclip_lora_grad_fp8: H=%d (supported: 768, 512)", H);
    BSCLIP_REQUIRE(ld_dqkv >= 3 * H && ld_dqkv % 4 == 0 && ld_y >= H && ld_y % 4 == 0 && ld_t >= 8 && ld_t % 8 == 0,
                   "bsclip_lora_grad_fp8: ld_dqkv=%d ld_y=%d ld_t=%d", ld_dqkv, ld_y, ld_t);
    hipStream_t s = static_cast<hipStream_t>(stream);
    int blocks = ceil_div(M, 4 * 8);
    if (blocks > LG_MAX_BLOCKS) blocks = LG_MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    const bf16_t* g = static_cast<const bf16_t*>(dqkv);
    const bf16_t* tt = static_cast<const bf16_t*>(t_aug) - H;   // the kernel reads t at column H of its "haug" row
    const bf16_t* yy = static_cast<const bf16_t*>(y_fp8);
    float* pa = workspace;
    float* pb = workspace + (size_t)LG_MAX_BLOCKS * 8 * H;
    if (H == 768) {
        hipLaunchKernelGGL((lora_grad_dt_db_kernel<768>), dim3(blocks), dim3(LG_BLOCK), 0, s, g, ld_dqkv, tt, ld_t, M, lora_b, dt, pa);
        hipLaunchKernelGGL((lora_grad_da_kernel<768, true>), dim3(blocks), dim3(LG_BLOCK), 0, s, yy, ld_y, M, dt, pb);
        hipLaunchKernelGGL((lora_grad_reduce_kernel<768>), dim3(16 * 768 / 32), dim3(256), 0, s, pa, pb, blocks, dA, dBq, dBv);
    } else {
        hipLaunchKernelGGL((lora_grad_dt_db_kernel<512>), dim3(blocks), dim3(LG_BLOCK), 0, s, g, ld_dqkv, tt, ld_t, M, lora_b, dt, pa);
        hipLaunchKernelGGL((lora_grad_da_kernel<512, true>), dim3(blocks), dim3(LG_BLOCK), 0, s, yy, ld_y, M, dt, pb);
        hipLaunchKernelGGL((lora_grad_reduce_kernel<512>), dim3(16 * 512 / 32), dim3(256), 0, s, pa, pb, blocks, dA, dBq, dBv);
    }
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int bsclip_colsum(const void* g, int ld_g, int g_is_bf16, int M, int N, float* out, void* stream) {
    BSCLIP_REQUIRE(g && out && M > 0 && N > 0 && ld_g >= N, "bsclip_colsum: bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool wide = N % 8 == 0 && ld_g % 8 == 0 && (((uintptr_t)g) & 15) == 0;
    if (wide) {
        if (g_is_bf16) hipLaunchKernelGGL((colsum8_kernel<true>), dim3(N / 8), dim3(256), 0, s, g, ld_g, M, N, out);
        else hipLaunchKernelGGL((colsum8_kernel<false>), dim3(N / 8), dim3(256), 0, s, g, ld_g, M, N, out);
    } else {
        const dim3 grid(ceil_div(N, 32));
        if (g_is_bf16) hipLaunchKernelGGL((colsum_kernel<true>), grid, dim3(256), 0, s, g, ld_g, M, N, out);
        else hipLaunchKernelGGL((colsum_kernel<false>), grid, dim3(256), 0, s, g, ld_g, M, N, out);
    }
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int bsclip_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                                 float beta2, float eps, float weight_decay, int step, float grad_scale,
                                 void* stream) {
    BSCLIP_REQUIRE(p && g && m && v && n > 0 && step >= 1, "bsclip_adamw_step: bad args (n=%ld step=%d)", (long)n, step);
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, m, v,
                       (long)n, lr, beta1, beta2, eps, weight_decay, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)),
                       grad_scale);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int bsclip_adamw_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper_dev,
                                     float beta1, float beta2, float eps, float weight_decay, float grad_scale, void* stream) {
    BSCLIP_REQUIRE(p && g && m && v && hyper_dev && n > 0, "bsclip_adamw_step_dev: null/empty input");
    long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adamw_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, m, v,
                       (long)n, hyper_dev, beta1, beta2, eps, weight_decay, grad_scale);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}
