// Helpers shared by the attention kernels (attn.hip: one workgroup of 4 query-/key-owner waves per head, two phases;
// attn_sweep.hip: key-owner sweep with a dQ wave): MFMA wrappers, the XOR-swizzled row-major LDS tile, its plain and
// TRANSPOSED (ds_read_b64_tr_b16) operand fragments, LDS-DMA staging, dropout keep-factors, last-tile trimming.
#pragma once
#include <math.h>

#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
// registers [8*s2, 8*s2+8) of an accumulator tile -> bf16x8 operand fragment (element j = register 8*s2 + j)
__device__ __forceinline__ bf16x8 pack8(const f32x16& x, int s2) {
    u32x4 u;
    u[0] = pack_bf2(x[8 * s2 + 0], x[8 * s2 + 1]);
    u[1] = pack_bf2(x[8 * s2 + 2], x[8 * s2 + 3]);
    u[2] = pack_bf2(x[8 * s2 + 4], x[8 * s2 + 5]);
    u[3] = pack_bf2(x[8 * s2 + 6], x[8 * s2 + 7]);
    return __builtin_bit_cast(bf16x8, u);
}

// the two bf16 halves of a packed word as floats
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
constexpr int ROWB = 128;  // bytes per row of a row-major [rows][64] bf16 tile
constexpr int ATT_WAVES = 4;  // waves per workgroup; wave w owns 32-row blocks w, w+4, ...

// Row-major tile of 128-byte rows, the 16-B chunk index XOR-swizzled with tile_sw(row).  Two rows share one 256-byte bank row; with
// t = row >> 1 the swizzle is the bijection t -> (t >> 1) | ((t & 1) << 2) of 0..7 (round 3 used t itself):
//   * ds_read_b128 (frag_rm): a 16-lane service group reads one chunk column of rows {0-3, 12-15, 20-27} (+ the analogous sets): the 8
//     rows of one parity have 8 different t mod 8, hence 8 different chunk positions -- conflict-free, as before;
//   * ds_read_b64_tr_b16 (frag_tr): a 32-lane group reads 4 aligned chunks of rows c0 .. c0 + 3; rows c0 and c0 + 2 share the half bank
//     row and have t = 2u, 2u + 1: their swizzles differ in bit 2, so the two 4-chunk groups land in different halves of the row.
//     With the identity they differed in bit 0 only -- the same four positions, a 2-way conflict on every transposing read (21 % of the
//     LDS-active cycles of the attention kernels: profiles/r04_b_attn_pmc.txt).
__device__ __forceinline__ int tile_sw(int row) { return ((row >> 2) & 3) | (((row >> 1) & 1) << 2); }
__device__ __forceinline__ int rm_off(int row, int chunk) { return row * ROWB + ((chunk ^ tile_sw(row)) << 4); }

// Stage a [S][64] bf16 matrix (row stride ld elements) into a row-major, XOR-swizzled LDS tile of SP rows with LDS-DMA
// (global_load_lds_dwordx4: no staging registers, every piece in flight at once; the register-staged version issued its
// 7 loads per thread one HBM round trip after the other: 11 us per K+V staging, a fifth of the backward kernel).
// One wave-instruction fills 1 KiB = 8 rows; the swizzle is applied on the source side (LDS destination is lane-linear).
// Rows >= S repeat row S-1: padded keys / queries always carry probability 0, so their content only has to be finite.
// The caller waits (vmcnt(0)) and synchronises.
template <int SP, int NWAVES = 4>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ src, int ld, int S, char* dst_rm, int wave,
                                           int lane) {
    const int r8 = lane >> 3, pc = lane & 7;
#pragma unroll
    for (int c = 0; c < (SP / 8 + NWAVES - 1) / NWAVES; ++c) {
        const int chunk = wave + NWAVES * c;
        if (chunk < SP / 8) {
            const int row = 8 * chunk + r8;
            const int lc = pc ^ tile_sw(row);
            glds16(src + (size_t)min(row, S - 1) * ld + lc * 8, dst_rm + chunk * 1024);
        }
    }
}
__device__ __forceinline__ void stage_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// A-operand fragment of a row-major tile: rows r0 + (lane&31), k = 16*ks + 8*(lane>>5) + j
__device__ __forceinline__ bf16x8 frag_rm(const char* tile, int r0, int ks, int lane) {
    return *reinterpret_cast<const bf16x8*>(tile + rm_off(r0 + (lane & 31), 2 * ks + (lane >> 5)));
}
// TRANSPOSED A-operand fragment of the same row-major tile, for the accumulator-as-B product: MFMA row = tile COLUMN
// d0 + (lane&31); element j is tile ROW c0 + 8*(j>>2) + 4*(lane>>5) + (j&3) (the k order of pack8()).
// ds_read_b64_tr_b16 (verified on hardware by tools/probe/tr_probe.py): within each 16-lane group, lane 4q+p supplies
// the address of row q / columns 4p..4p+3 of a 4x16 block and lane i receives column i of the 4 rows.  Group
// gi = lane>>4 serves columns d0 + 16*(gi&1) + [0,16) and rows c0 + 4*(gi>>1) + [0,4)  (gi>>1 == lane>>5).
typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ bf16x8 frag_tr(const char* tile, int d0, int c0, int lane) {
    const int gi = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int row = c0 + 4 * (gi >> 1) + q;
    const int col = d0 + 16 * (gi & 1) + 4 * pp;  // 4 consecutive bf16 = 8 B inside one 16-B chunk
    const char* p0 = tile + rm_off(row, col >> 3) + (col & 7) * 2;
    const char* p1 = tile + rm_off(row + 8, col >> 3) + (col & 7) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
// B-operand fragment straight from global: row (clamped) of a [S][64] matrix, k = 16*ks + 8*(lane>>5) + j
__device__ __forceinline__ bf16x8 frag_global(const bf16_t* base, int ld, int row, int ks, int lane) {
    return *reinterpret_cast<const bf16x8*>(base + (size_t)row * ld + ks * 16 + 8 * (lane >> 5));
}
// accumulator row index of register r for lane half h (C/D map of the 32x32 MFMA)
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// store a [64 (d) x 32 (token on lane)] result held as 2 accumulator tiles into out[token][col0 + d]
__device__ __forceinline__ void store_dt(const f32x16 (&acc)[2], float mul, bf16_t* out_row, int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 o;
            o.x = pack_bf2(acc[dt][4 * g + 0] * mul, acc[dt][4 * g + 1] * mul);
            o.y = pack_bf2(acc[dt][4 * g + 2] * mul, acc[dt][4 * g + 3] * mul);
            *reinterpret_cast<uint2*>(out_row + 32 * dt + 8 * g + 4 * h) = o;
        }
}

// the same, to 16 mantissa bits: hi = bf16(x), lo = bf16(x - hi)
__device__ __forceinline__ void store_dt_hilo(const f32x16 (&acc)[2], float mul, bf16_t* hi_row, bf16_t* lo_row, int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float x0 = acc[dt][4 * g + 0] * mul, x1 = acc[dt][4 * g + 1] * mul, x2 = acc[dt][4 * g + 2] * mul,
                        x3 = acc[dt][4 * g + 3] * mul;
            uint2 o, l;
            o.x = pack_bf2(x0, x1);
            o.y = pack_bf2(x2, x3);
            l.x = pack_bf2(x0 - bf_lo(o.x), x1 - bf_hi(o.x));
            l.y = pack_bf2(x2 - bf_lo(o.y), x3 - bf_hi(o.y));
            *reinterpret_cast<uint2*>(hi_row + 32 * dt + 8 * g + 4 * h) = o;
            *reinterpret_cast<uint2*>(lo_row + 32 * dt + 8 * g + 4 * h) = l;
        }
}

// keep-factors (0 or 1/(1-p)) of the 4 consecutive keys idx .. idx+3 of one query row (idx % 4 == 0): two hashes
__device__ __forceinline__ f32x4 keep4(const DropCfg& d, unsigned idx) {
    const unsigned b0 = drop_pair_bits(d, idx), b1 = drop_pair_bits(d, idx + 2);
    f32x4 k;
    k[0] = (b0 & 0xffffU) >= d.thr16 ? d.scale : 0.f;
    k[1] = (b0 >> 16) >= d.thr16 ? d.scale : 0.f;
    k[2] = (b1 & 0xffffU) >= d.thr16 ? d.scale : 0.f;
    k[3] = (b1 >> 16) >= d.thr16 ? d.scale : 0.f;
    return k;
}
// the four keep decisions of keys idx .. idx+3 as a nibble (bit i = key idx + i kept): what the forward leaves in the keep-bit words
__device__ __forceinline__ unsigned keep4_nibble(const DropCfg& d, unsigned idx) {
    const unsigned b0 = drop_pair_bits(d, idx), b1 = drop_pair_bits(d, idx + 2);
    return ((b0 & 0xffffU) >= d.thr16 ? 1u : 0u) | ((b0 >> 16) >= d.thr16 ? 2u : 0u) | ((b1 & 0xffffU) >= d.thr16 ? 4u : 0u) |
           ((b1 >> 16) >= d.thr16 ? 8u : 0u);
}
// Keep-bit words (round 5): the forward leaves, per (batch, head) item and query row, 2 x KEEP_WORDS 32-bit words -- one set per lane
// half h of the wave that owned the row (no cross-lane exchange): bit 8 g + i of word kt of half h = "key 32 kt + 8 g + 4 h + i of this
// query row was kept", i.e. exactly the keys that lane half holds in the score accumulators -- so that the backward reads the decisions
// instead of re-hashing every element (its key-owner phase hashed once per ELEMENT: 67 of 218 us at S = 133,
// profiles/r04_b_attn_bench.log).  The query-owner phase reads its own half's words; the key-owner phase stages W0 | (W1 << 4)
// (bit j = key 32 kt + j).  Same decisions, same arithmetic: the gradients are bit for bit those of the hashing form.
constexpr int KEEP_WORDS = 8;   // words per lane half (S <= 224 needs 7; 8 keeps 16-byte stores aligned): 64 B per query row
// keep factor (0 or scale) of bit `pos` of a keep word: one v_bfe_i32 (0 / -1) + one v_and
__device__ __forceinline__ float keep_of_bit(unsigned w, unsigned pos, float scale) {
    return __int_as_float(__builtin_amdgcn_sbfe((int)w, pos, 1u) & __float_as_int(scale));
}
// Dropout element index of P[q, key] for head-instance bh: ((bh * S + q) * SP + key), SP = padded length (multiple of
// 32), so a lane's 4 consecutive keys share two hash pairs.

// TAIL = number of valid rows of the LAST 32-row key / query tile (S - 32 (NB - 1)), as a template parameter for the production
// sequence lengths (197 and 133 both leave 5) and 32 ("treat the tile as full") for every other S.  An accumulator register r of
// a 32x32 tile holds row (r & 3) + 8 (r >> 2) + 4 h: the 4-register group g = r >> 2 covers rows [8g, 8g + 8), so in the last
// tile only the first ceil(TAIL / 8) groups can hold a non-zero probability and only the first ceil(TAIL / 16) 16-deep k-steps
// of a product over that tile's rows contribute.  Skipping the rest is decided at compile time (the tile loops are unrolled) --
// round 2's attempt with wave-uniform RUNTIME branches cost the schedule more than it saved.  S = 197 pads to 224 rows (29 %
// more tile pairs than needed), S = 133 to 160 (45 %): this removes the VALU share and a quarter of the MFMAs of that padding.
template <int TAIL> constexpr int tail_groups() { return TAIL >= 32 ? 4 : (TAIL + 7) / 8; }
template <int TAIL> constexpr int tail_ksteps() { return TAIL > 16 ? 2 : 1; }

// TRANSPOSED fragment of a small UNSWIZZLED row-major bf16 tile with `rowb` bytes per row (the dS^T tiles of attn_sweep.hip:
// [32 keys][32 queries], rows padded to 72 B): MFMA row / column index = tile COLUMN d0 + (lane & 31), element j = tile ROW
// c0 + 8 (j >> 2) + 4 (lane >> 5) + (j & 3) -- the same mapping as frag_tr, so the two can meet in one MFMA.
__device__ __forceinline__ bf16x8 frag_tr_lin(const char* tile, int rowb, int d0, int c0, int lane) {
    const int gi = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int row = c0 + 4 * (gi >> 1) + q;
    const int col = d0 + 16 * (gi & 1) + 4 * pp;
    const char* p0 = tile + row * rowb + col * 2;
    const char* p1 = p0 + 8 * rowb;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
// acc += lo(w) + hi(w) for a packed bf16 pair: one v_dot2c_f32_bf16 against (1, 1)
__device__ __forceinline__ float bf_pair_sum(unsigned w, float acc) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, w), __builtin_bit_cast(bf2_t, 0x3f803f80u), acc, false);
}

}  // namespace
