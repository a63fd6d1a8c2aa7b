// Pooling heads (gfx950), HBM-bound, wave-per-row with shuffle reductions.
//   softmax_meanpool: LoRA_barcode_bert.forward `logits.softmax(dim=-1).mean(dim=1)` (dna_encoder.py:105)
//   meanpool_tokens : LoRA_bert.forward `last_hidden_state.mean(dim=1)` (language_encoder.py:89; padding included)
#include <math.h>

#include "common.h"

namespace {

// One workgroup (4 waves) per sequence; wave w walks tokens w, w+4, ...; each lane owns C/64 columns.
template <int C>
__global__ __launch_bounds__(256) void softmax_meanpool_fwd_kernel(const float* __restrict__ logits, int S,
                                                                    float* __restrict__ pooled,
                                                                    float* __restrict__ stats) {
    constexpr int NV = C / 256;
    __shared__ f32x4 red[4][NV][64];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int t = wave; t < S; t += 4) {
        const size_t row = (size_t)b * S + t;
        f32x4 v[NV];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            v[j] = *reinterpret_cast<const f32x4*>(logits + row * C + j * 256 + lane * 4);
            m = fmaxf(m, fmaxf(fmaxf(v[j][0], v[j][1]), fmaxf(v[j][2], v[j][3])));
        }
        m = wave_max(m);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[j][i] = __expf(v[j][i] - m);
                s += v[j][i];
            }
        }
        s = wave_sum(s);
        const float inv = 1.0f / s;
#pragma unroll
        for (int j = 0; j < NV; ++j) acc[j] += v[j] * inv;
        if (stats && lane == 0) {
            stats[2 * row] = m;
            stats[2 * row + 1] = s;
        }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) red[wave][j][lane] = acc[j];
    __syncthreads();
    if (wave == 0) {
        const float invS = 1.0f / (float)S;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const f32x4 t = (red[0][j][lane] + red[1][j][lane]) + (red[2][j][lane] + red[3][j][lane]);
            *reinterpret_cast<f32x4*>(pooled + (size_t)b * C + j * 256 + lane * 4) = t * invS;
        }
    }
}

// dlogits[b,t,c] = p_c (g_c - sum_j p_j g_j),  g = d_pooled[b] / S;  one wave per (b,t) row
template <int C>
__global__ __launch_bounds__(256) void softmax_meanpool_bwd_kernel(const float* __restrict__ logits,
                                                                    const float* __restrict__ stats,
                                                                    const float* __restrict__ d_pooled, int M, int S,
                                                                    bf16_t* __restrict__ dlogits, int ld_d) {
    constexpr int NV = C / 256;
    const int lane = threadIdx.x & 63;
    const int row = (blockIdx.x * 256 + threadIdx.x) >> 6;
    if (row >= M) return;
    const int b = row / S;
    const float m = stats[2 * (size_t)row], inv = 1.0f / stats[2 * (size_t)row + 1];
    const float invS = 1.0f / (float)S;
    f32x4 p[NV], g[NV];
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(logits + (size_t)row * C + j * 256 + lane * 4);
        g[j] = *reinterpret_cast<const f32x4*>(d_pooled + (size_t)b * C + j * 256 + lane * 4) * invS;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            p[j][i] = __expf(v[i] - m) * inv;
            dot += p[j][i] * g[j][i];
        }
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const f32x4 d = p[j] * (g[j] - dot);
        uint2 o;
        o.x = pack_bf2(d[0], d[1]);
        o.y = pack_bf2(d[2], d[3]);
        *reinterpret_cast<uint2*>(dlogits + (size_t)row * ld_d + j * 256 + lane * 4) = o;
    }
}

__global__ __launch_bounds__(256) void meanpool_tokens_fwd_kernel(const float* __restrict__ x, int B, int S, int H,
                                                                   bf16_t* __restrict__ out, int ld_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;  // one thread = 4 columns of one sequence
    const int per = H / 4;
    if (i >= B * per) return;
    const int b = i / per, c = (i % per) * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < S; ++t) acc += *reinterpret_cast<const f32x4*>(x + ((size_t)b * S + t) * H + c);
    acc *= 1.0f / (float)S;
    uint2 o;
    o.x = pack_bf2(acc[0], acc[1]);
    o.y = pack_bf2(acc[2], acc[3]);
    *reinterpret_cast<uint2*>(out + (size_t)b * ld_out + c) = o;
}

// dx[b,t,:] = d_pooled[b,:] / S
__global__ __launch_bounds__(256) void meanpool_tokens_bwd_kernel(const float* __restrict__ d_pooled, int ld_d, int B,
                                                                   int S, int H, float* __restrict__ dx) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int per = H / 4;
    if (i >= (long)B * S * per) return;
    const long row = i / per;
    const int c = (int)(i % per) * 4;
    const int b = (int)(row / S);
    const f32x4 g = *reinterpret_cast<const f32x4*>(d_pooled + (size_t)b * ld_d + c) * (1.0f / (float)S);
    *reinterpret_cast<f32x4*>(dx + (size_t)row * H + c) = g;
}

// out = g * d (d = gelu'(pre-activation) saved by the forward GELU epilogue as 8-bit codes), 4 values per thread
__global__ __launch_bounds__(256) void dgelu_mul_kernel(const bf16_t* __restrict__ g, int ld_g,
                                                         const unsigned char* __restrict__ z, int ld_z, int M, int N,
                                                         bf16_t* __restrict__ out, int ld_o) {
    const int per = N / 4;
    const long total = (long)M * per;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
        const long row = it / per;
        const int c = (int)(it % per) * 4;
        const uint2 gu = *reinterpret_cast<const uint2*>(g + row * ld_g + c);
        const f32x4 d = dg8_unpack4(*reinterpret_cast<const unsigned*>(z + row * ld_z + c));
        uint2 o;
        o.x = pack_bf2(bf2f(gu.x & 0xffff) * d[0], bf2f(gu.x >> 16) * d[1]);
        o.y = pack_bf2(bf2f(gu.y & 0xffff) * d[2], bf2f(gu.y >> 16) * d[3]);
        *reinterpret_cast<uint2*>(out + row * ld_o + c) = o;
    }
}

}  // namespace

extern "C" int bsclip_dgelu_mul(const void* g, int ld_g, const void* z, int ld_z, int M, int N, void* out, int ld_o,
                                void* stream) {
    BSCLIP_REQUIRE(g && z && out && M > 0 && N > 0 && N % 4 == 0, "bsclip_dgelu_mul: bad args");
    BSCLIP_REQUIRE(ld_g >= N && ld_z >= N && ld_o >= N && ld_g % 4 == 0 && ld_z % 4 == 0 && ld_o % 4 == 0,
                   "bsclip_dgelu_mul: ld_g=%d ld_z=%d ld_o=%d", ld_g, ld_z, ld_o);
    long blocks = ((long)M * (N / 4) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(dgelu_mul_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const bf16_t*>(g), ld_g, static_cast<const unsigned char*>(z), ld_z, M, N,
                       static_cast<bf16_t*>(out), ld_o);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int bsclip_softmax_meanpool_fwd(const float* logits, int B, int S, int C, float* pooled, float* stats,
                                           void* stream) {
    BSCLIP_REQUIRE(logits && pooled && B > 0 && S > 0, "bsclip_softmax_meanpool_fwd: bad args");
    BSCLIP_REQUIRE(C == 768, "bsclip_softmax_meanpool_fwd: C=%d (supported: 768)", C);
    hipLaunchKernelGGL((softmax_meanpool_fwd_kernel<768>), dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream),
                       logits, S, pooled, stats);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int bsclip_softmax_meanpool_bwd(const float* logits, const float* stats, const float* d_pooled, int B, int S,
                                           int C, void* dlogits_bf16, int ld_d, void* stream) {
    BSCLIP_REQUIRE(logits && stats && d_pooled && dlogits_bf16 && B > 0 && S > 0, "bsclip_softmax_meanpool_bwd: bad args");
    BSCLIP_REQUIRE(C == 768 && ld_d >= C && ld_d % 4 == 0, "bsclip_softmax_meanpool_bwd: C=%d ld_d=%d", C, ld_d);
    const int M = B * S;
    hipLaunchKernelGGL((softmax_meanpool_bwd_kernel<768>), dim3(ceil_div(M, 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), logits, stats, d_pooled, M, S,
                       static_cast<bf16_t*>(dlogits_bf16), ld_d);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int bsclip_meanpool_tokens_fwd(const float* x, int B, int S, int H, void* out_bf16, int ld_out,
                                          void* stream) {
    BSCLIP_REQUIRE(x && out_bf16 && B > 0 && S > 0 && H % 4 == 0 && ld_out >= H && ld_out % 4 == 0,
                   "bsclip_meanpool_tokens_fwd: bad args");
    hipLaunchKernelGGL(meanpool_tokens_fwd_kernel, dim3(ceil_div(B * (H / 4), 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, B, S, H, static_cast<bf16_t*>(out_bf16), ld_out);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int bsclip_meanpool_tokens_bwd(const float* d_pooled, int ld_d, int B, int S, int H, float* dx,
                                          void* stream) {
    BSCLIP_REQUIRE(d_pooled && dx && B > 0 && S > 0 && H % 4 == 0 && ld_d >= H && ld_d % 4 == 0,
                   "bsclip_meanpool_tokens_bwd: bad args");
    const long n = (long)B * S * (H / 4);
    hipLaunchKernelGGL(meanpool_tokens_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), d_pooled, ld_d, B, S, H, dx);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}
