// Input pipeline on the GPU: the two per-sample transforms that sit between the stored data and the encoders.
//
//   * 5-mer tokeniser: reference get_sequence_pipeline (bioscanclip/model/dna_encoder.py:25-35) = PadSequence(660) +
//     KmerTokenizer(k=5, stride=5) (bioscanclip/util/util.py:48-69) + torchtext vocab lookup + the literal 0 in front.
//   * training augmentation chain of Dataset_for_CL (bioscanclip/util/dataset.py:171-181): ToTensor -> Resize(256,
//     antialias) -> RandomResizedCrop(224, antialias) -> RandomHorizontalFlip -> RandomVerticalFlip -> RandomRotation(+-45 deg),
//     with the random draws made by the caller (seeded, per sample) and passed as a parameter record, so the arithmetic is a
//     pure function that a CPU oracle can check.  Evaluation (Resize(256) -> CenterCrop(224)) is the same kernels with a
//     centred box, no flips, angle 0.
// At >= 50 k pairs/s/node the CPU version of this chain is what starves the step; decoded uint8 images go in,
// the encoder's f32 [B,3,224,224] comes out.  Both kernels are byte/float streaming work: coalesced reads of small rows,
// no MFMA, bound by HBM / L2.
#include "common.h"

namespace {

// ---- tokeniser --------------------------------------------------------------------------------------------------------
// one thread per token: token 0 = 0 (<MASK>); token t >= 1 = k-mer over padded[(t-1)*5 .. +5), padded = seq truncated to
// 660 or right-padded with 'N'.  id = 3 + sum_i d(c_i) 4^(4-i), d(A,C,G,T) = 0..3 (torchtext orders the equal-frequency
// k-mers as itertools.product generates them); any other character (lower case included: the reference's vocab holds upper
// case only) makes the k-mer <UNK> = 2.
__global__ __launch_bounds__(256) void kmer_tokenize_kernel(const unsigned char* __restrict__ seqs,
                                                            const int64_t* __restrict__ offsets, int B, int max_len, int k,
                                                            int64_t* __restrict__ ids) {
    const int ntok = max_len / k + 1;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * ntok) return;
    const int b = i / ntok, t = i % ntok;
    if (t == 0) {
        ids[i] = 0;
        return;
    }
    const int64_t beg = offsets[b], len = offsets[b + 1] - beg;
    int64_t id = 0;
    bool unk = false;
    for (int j = 0; j < k; ++j) {
        const int pos = (t - 1) * k + j;
        const unsigned char c = pos < len ? seqs[beg + pos] : (unsigned char)'N';
        int d;
        switch (c) {
            case 'A': d = 0; break;
            case 'C': d = 1; break;
            case 'G': d = 2; break;
            case 'T': d = 3; break;
            default: d = 0; unk = true;
        }
        id = id * 4 + d;
    }
    ids[i] = unk ? 2 : 3 + id;
}

// ---- antialiased bilinear resampling (torch _upsample_bilinear2d_aa, which torchvision's Resize / RandomResizedCrop call) ----
// Output index o of a length-`out` axis resampled from `in` samples of the source interval [lo, lo + in): with scale = in / out,
//   support = max(scale, 1), centre = scale * (o + 0.5), xmin = max(0, int(centre - support + 0.5)),
//   xsize = min(in, int(centre + support + 0.5)) - xmin, weight_j = tri((j + xmin - centre + 0.5) / max(scale, 1)) / sum.
struct Taps {
    int xmin, xsize;
    float centre, inv;
};
__device__ __forceinline__ Taps aa_taps(int o, int in, int out) {
    const float scale = (float)in / (float)out;
    const float support = scale >= 1.0f ? scale : 1.0f;
    Taps t;
    t.centre = scale * ((float)o + 0.5f);
    t.inv = scale >= 1.0f ? 1.0f / scale : 1.0f;
    t.xmin = max(0, (int)(t.centre - support + 0.5f));
    t.xsize = min(in, (int)(t.centre + support + 0.5f)) - t.xmin;
    return t;
}
__device__ __forceinline__ float tri(float x) {
    x = fabsf(x);
    return x < 1.0f ? 1.0f - x : 0.0f;
}
__device__ __forceinline__ float tap_w(const Taps& t, int j) { return tri(((float)(j + t.xmin) - t.centre + 0.5f) * t.inv); }

// Per-image record (host-built, 16 ints): source offset (2 ints: low/high), H0, W0, H1, W1 (size after Resize(256)), crop top,
// left, height, width (in the resized image), hflip, vflip, rotate flag, cos(angle), sin(angle) (f32 bits; angle in degrees,
// counter-clockwise, cos/sin formed on the host in f64 like torchvision's _get_inverse_affine_matrix), 1 spare.
constexpr int REC = 16;

// pass 1: ToTensor + Resize(256): uint8 HWC source -> f32 [3, H1, W1] at mid[b * 3 * cap]  (cap = max H1*W1 over the batch).
// Only the pixels of the crop box are produced: pass 2's taps are clipped to the box (torchvision crops, then resizes), so the
// rest of the resized image is never read -- RandomResizedCrop keeps 8..100 % of the area (54 % on average), and the f32
// intermediate was the largest stream of the chain (round 2: 693 MB per 256-image batch moved).
__global__ __launch_bounds__(256) void augment_resize_kernel(const unsigned char* __restrict__ src, const int* __restrict__ rec,
                                                             int B, long cap, float* __restrict__ mid) {
    const int b = blockIdx.y;
    const int* r = rec + b * REC;
    const long off = (long)(unsigned)r[0] | ((long)r[1] << 32);
    const int H0 = r[2], W0 = r[3], H1 = r[4], W1 = r[5];
    const int top = r[6], left = r[7], ch = r[8], cw = r[9];
    const unsigned char* im = src + off;
    float* out = mid + (size_t)b * 3 * cap;
    for (int ib = blockIdx.x * 256 + threadIdx.x; ib < ch * cw; ib += gridDim.x * 256) {
        const int y = top + ib / cw, x = left + ib % cw;
        const int i = y * W1 + x;
        const Taps ty = aa_taps(y, H0, H1), tx = aa_taps(x, W0, W1);
        float wy_sum = 0.f, wx_sum = 0.f;
        for (int j = 0; j < ty.xsize; ++j) wy_sum += tap_w(ty, j);
        for (int j = 0; j < tx.xsize; ++j) wx_sum += tap_w(tx, j);
        // torch resamples separably, horizontal pass first then vertical, each in f32: keep that order of accumulation
        float acc[3] = {0.f, 0.f, 0.f};
        for (int jy = 0; jy < ty.xsize; ++jy) {
            const float wy = tap_w(ty, jy) / wy_sum;
            const unsigned char* row = im + ((size_t)(ty.xmin + jy) * W0 + tx.xmin) * 3;
            float h[3] = {0.f, 0.f, 0.f};
            for (int jx = 0; jx < tx.xsize; ++jx) {
                const float wx = tap_w(tx, jx) / wx_sum;
                h[0] += wx * ((float)row[jx * 3 + 0] / 255.0f);   // ToTensor: uint8 / 255 (a true division, as torch does)
                h[1] += wx * ((float)row[jx * 3 + 1] / 255.0f);
                h[2] += wx * ((float)row[jx * 3 + 2] / 255.0f);
            }
            acc[0] += wy * h[0];
            acc[1] += wy * h[1];
            acc[2] += wy * h[2];
        }
        out[i] = acc[0];
        out[cap + i] = acc[1];
        out[2 * cap + i] = acc[2];
    }
}

// pass 2: RandomResizedCrop(224) + flips + rotation, fused: output pixel -> inverse rotation (nearest, zeros outside) -> flips ->
// antialiased bilinear sample of the crop box of the resized image.
__global__ __launch_bounds__(256) void augment_crop_kernel(const float* __restrict__ mid, const int* __restrict__ rec, int B,
                                                           long cap, int S, float* __restrict__ out) {
    const int b = blockIdx.y;
    const int* r = rec + b * REC;
    const int W1 = r[5], top = r[6], left = r[7], ch = r[8], cw = r[9], hflip = r[10], vflip = r[11];
    const int rotate = r[12];
    const float ca = __int_as_float(r[13]), sa = __int_as_float(r[14]);
    const float* im = mid + (size_t)b * 3 * cap;
    float* o = out + (size_t)b * 3 * S * S;
    // torchvision F.rotate(angle) on a tensor (functional_tensor.rotate): theta = _get_inverse_affine_matrix(centre 0, -angle)
    // = [cos a, -sin a, 0; sin a, cos a, 0], affine grid over base coordinates (x + 0.5 - S/2, y + 0.5 - S/2) divided by S/2,
    // grid_sample(nearest, zeros, align_corners=False) -> source pixel = nearbyint(((g + 1) S - 1) / 2)
    const float t00 = ca / (0.5f * S), t10 = -sa / (0.5f * S), t01 = sa / (0.5f * S), t11 = ca / (0.5f * S);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < S * S; i += gridDim.x * 256) {
        const int y = i / S, x = i % S;
        int ys = y, xs = x;
        if (rotate) {
            const float xb = (float)x + 0.5f - 0.5f * S, yb = (float)y + 0.5f - 0.5f * S;
            const float gx = xb * t00 + yb * t10, gy = xb * t01 + yb * t11;
            xs = (int)nearbyintf(((gx + 1.0f) * S - 1.0f) * 0.5f);   // grid_sample 'nearest': std::nearbyint
            ys = (int)nearbyintf(((gy + 1.0f) * S - 1.0f) * 0.5f);
        }
        float v[3] = {0.f, 0.f, 0.f};
        if (xs >= 0 && xs < S && ys >= 0 && ys < S) {
            const int yy = vflip ? S - 1 - ys : ys, xx = hflip ? S - 1 - xs : xs;
            const Taps ty = aa_taps(yy, ch, S), tx = aa_taps(xx, cw, S);
            float wy_sum = 0.f, wx_sum = 0.f;
            for (int j = 0; j < ty.xsize; ++j) wy_sum += tap_w(ty, j);
            for (int j = 0; j < tx.xsize; ++j) wx_sum += tap_w(tx, j);
            for (int jy = 0; jy < ty.xsize; ++jy) {
                const float wy = tap_w(ty, jy) / wy_sum;
                const float* row = im + (size_t)(top + ty.xmin + jy) * W1 + left + tx.xmin;
                float h[3] = {0.f, 0.f, 0.f};
                for (int jx = 0; jx < tx.xsize; ++jx) {
                    const float wx = tap_w(tx, jx) / wx_sum;
                    h[0] += wx * row[jx];
                    h[1] += wx * row[cap + jx];
                    h[2] += wx * row[2 * cap + jx];
                }
                v[0] += wy * h[0];
                v[1] += wy * h[1];
                v[2] += wy * h[2];
            }
        }
        o[i] = v[0];
        o[S * S + i] = v[1];
        o[2 * S * S + i] = v[2];
    }
}

}  // namespace

extern "C" int bsclip_kmer_tokenize(const void* seqs, const int64_t* offsets, int B, int max_len, int k, int64_t* ids,
                                    void* stream) {
    BSCLIP_REQUIRE(seqs && offsets && ids && B > 0, "bsclip_kmer_tokenize: null/empty input");
    BSCLIP_REQUIRE(k >= 1 && k <= 12 && max_len >= k && max_len % k == 0, "bsclip_kmer_tokenize: k=%d max_len=%d", k, max_len);
    const int n = B * (max_len / k + 1);
    hipLaunchKernelGGL(kmer_tokenize_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const unsigned char*>(seqs), offsets, B, max_len, k, ids);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int bsclip_augment_images(const void* src_u8, const int32_t* records, int B, int64_t mid_capacity, float* mid,
                                     int out_size, float* out, void* stream) {
    BSCLIP_REQUIRE(src_u8 && records && mid && out && B > 0 && mid_capacity > 0 && out_size > 0,
                   "bsclip_augment_images: null/empty input");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 g1((unsigned)min((int64_t)1024, (mid_capacity + 255) / 256), B);
    hipLaunchKernelGGL(augment_resize_kernel, g1, dim3(256), 0, s, static_cast<const unsigned char*>(src_u8), records, B,
                       (long)mid_capacity, mid);
    const dim3 g2(ceil_div(out_size * out_size, 256), B);
    hipLaunchKernelGGL(augment_crop_kernel, g2, dim3(256), 0, s, mid, records, B, (long)mid_capacity, out_size, out);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}
