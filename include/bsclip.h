/* libbsclip_hip.so -- C ABI of the MI355X (gfx950) BIOSCAN-CLIP contrastive-training-step kernels.
 *
 * The reference (bioscan-ml/bioscan-clip) has no FFI: its boundary for this path is the Python surface of
 * bioscanclip.model.* (SURVEY.md 8b).  Every arithmetic op that surface performs through torch/cuDNN/cuBLAS is
 * replaced by one entry point below; the comment on each names the reference site (file:line under
 * /root/reference) whose arithmetic it takes over.  bioscan-clip_amd/bioscanclip/hip/lib.py binds these with
 * ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is DEVICE memory unless named host_*.
 *   - bf16 tensors are raw uint16 bit patterns; "f32" = float; ids/labels are int64.
 *   - row-major; ld* = leading dimension in ELEMENTS.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  No entry point allocates,
 *     synchronises or touches the host; all are capturable into a hipGraph.
 *   - return 0 on success, <0 on error; bsclip_last_error() describes the last failure of the calling thread.
 */
#ifndef BSCLIP_H
#define BSCLIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define BSCLIP_OK 0
#define BSCLIP_ERR_INVALID (-1)
#define BSCLIP_ERR_LAUNCH (-2)

#define BSCLIP_LORA_COLS 8 /* r=4 for Q plus r=4 for V */
#define BSCLIP_KPAD 64     /* K-augmentation block appended to LN outputs: [t_q(4) t_v(4) 0...] */

const char* bsclip_last_error(void);
int bsclip_abi_version(void);

/* Clock counters sampled when `stream` reaches this node: for XCD x (HW_REG_XCC_ID, 0..15) out32[2x] = its shader-clock counter
 * (s_memtime), out32[2x + 1] = the 100 MHz real-time counter (s_memrealtime); entries of absent XCDs stay untouched (zero them
 * first).  Two probes around a region give the clock the chip actually ran at in it, per XCD: delta(s_memtime) /
 * delta(s_memrealtime) x 100 MHz.  bench.py reports the median, because the dense-MFMA peak a roofline is priced against
 * assumes the 2.4 GHz peak engine clock. */
int bsclip_clock_probe(unsigned long long* out32_dev, void* stream);
/* Dropout step word (hipGraph support).  Kernel arguments are frozen when a launch is captured into a graph, so a seed passed
 * by value would repeat its mask on every replay.  After bsclip_set_dropout_step(ptr) every dropout-carrying launch made BY
 * THE CALLING THREAD adds (*ptr * 0x9E3779B9) to its seed when the kernel runs (NULL: seeds are used as passed).  The word
 * is advanced once per training step with bsclip_counter_add (itself a capturable launch); forward and backward of a step read
 * the same value, so regenerated masks match. */
int bsclip_set_dropout_step(const uint32_t* step_dev);
int bsclip_counter_add(uint32_t* counter_dev, uint32_t inc, void* stream);
/* *counter_dev += the number of Inf / NaN elements among the n contiguous f32 (is_bf16 = 0) or bf16 (1) values at x.  The check behind
 * BSCLIP_DETECT_ANOMALY=1, this build's counterpart of torch.autograd.set_detect_anomaly(True) in the reference loop
 * (train_epoch.py:12): the engines probe their outputs, gradients and per-layer activations and name where non-finite values
 * first appear. */
int bsclip_count_nonfinite(const void* x, int64_t n, int is_bf16, uint32_t* counter_dev, void* stream);

/* ---- GEMM family: C = epilogue(A[M,K] * B[N,K]^T), bf16 operands, f32 accumulate (MFMA 16x16x32) -------------
 * Replaces every torch.nn.Linear on the path: timm Attention.qkv/proj, Mlp.fc1/fc2 (via
 * bioscanclip/model/image_encoder.py:108-109), HF BertSelfAttention q/k/v, BertSelfOutput.dense,
 * BertIntermediate.dense, BertOutput.dense, cls.predictions.{transform.dense,decoder}
 * (bioscanclip/model/dna_encoder.py:105), LoRA_bert.proj (language_encoder.py:89) and their autograd dX passes.
 * The LoRA side branch (image_encoder.py:42-48, dna_encoder.py:47-49) rides in the K dimension: A carries
 * BSCLIP_KPAD extra columns holding t = x A_lora^T and B carries the matching LoRA-B columns.
 * Requirements: K % 64 == 0, N % 128 == 0, lda/ldb % 8 == 0, 16-byte aligned bases. M is arbitrary. */
enum bsclip_epilogue {
    BSCLIP_EPI_BF16 = 0,       /* C bf16 = acc + bias                                               */
    BSCLIP_EPI_F32 = 1,        /* C f32  = acc + bias                                               */
    BSCLIP_EPI_GELU_BF16 = 2,  /* C bf16 = gelu(acc + bias); aux (uint8 codes, optional) = gelu'(acc + bias) */
    BSCLIP_EPI_RESID_F32 = 3,  /* C f32  = acc + bias + resid                                       */
    BSCLIP_EPI_DGELU_BF16 = 4, /* C bf16 = acc * aux          (aux = gelu' codes saved by the forward) */
    BSCLIP_EPI_PATCH_F32 = 5,  /* C f32 row (b*197+1+p) = acc + bias + pos[1+p], input row b*196+p  */
    BSCLIP_EPI_GELU_FP8 = 6,   /* bsclip_gemm_fp8 only: C fp8 e4m3 = gelu(..) (the next GEMM's operand); aux as GELU_BF16 */
    /* the residual stream stored as bf16 (the sum is formed in f32 and rounded once): 4 instead of 8 bytes per element moved */
    BSCLIP_EPI_RESID_BF16 = 7, /* C bf16 = acc + bias + resid, resid bf16 [M, ld_resid] (+ dropout as RESID_F32)  */
    BSCLIP_EPI_PATCH_BF16 = 8  /* as PATCH_F32 with a bf16 C (pos_embed stays f32)                         */
};
typedef struct bsclip_epi_args {
    uint32_t struct_size; /* = sizeof(bsclip_epi_args) = bsclip_epi_args_size(); a mismatch is rejected (ABI drift guard) */
    const float* bias;    /* [N] or NULL */
    const void* resid;  /* RESID_F32: f32 [M, ld_resid]; RESID_BF16: bf16 [M, ld_resid]; PATCH_*: pos_embed f32 [197, N] */
    int ld_resid;
    void* aux; /* gelu' side band, uint8 [M, ld_aux]: code = round((gelu' + 0.13) * 255 / 1.26); GELU: out (nullable), DGELU: in;
                * ld_aux % 16 == 0, 16-byte aligned */
    int ld_aux;
    float dropout_p;        /* RESID_* only: C = dropout(acc + bias) + resid (HF hidden_dropout_prob); 0 = off */
    uint32_t dropout_seed;  /* decision of element (m,n) = f(seed, m*N + n): see bsclip_layernorm_bwd */
} bsclip_epi_args;
int bsclip_gemm_bf16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                     int epilogue, const bsclip_epi_args* args, void* stream);
/* sizeof(bsclip_epi_args) as this library was compiled; bindings assert their own struct against it */
int bsclip_epi_args_size(void);

/* ---- fp8 GEMM (BASELINE.json configs[4]: fp8 MFMA encoders, bf16 LoRA / loss) ------------------------------------
 * C = epilogue(alpha[n] * (A8[M,K] . B8[N,K]^T + A_aug[M,64] . B_aug[N,64]^T) + bias[n]) on the 256x256 ping-pong kernel.
 * A8 / B8: OCP fp8 e4m3 bytes, row-major, lda/ldb in elements (= bytes), K % 128 == 0, N % 256 == 0, lda/ldb % 16 == 0.
 * alpha f32 [N]: dequantisation scale per output column (activation scale x weight-row scale; bsclip_quantize_rows_fp8).
 * a_aug / b_aug (both or neither): bf16 K-augmentation block multiplied on the bf16 MFMA as the last K-tile -- the LoRA
 *   branch: a_aug = t columns written by bsclip_layernorm_fwd (y_fp8 mode), b_aug = LoRA-B columns / alpha[n]
 *   (bsclip_lora_baug_set).  ld_* in bf16 elements, % 8 == 0.
 * form: 1 = v_mfma_f32_16x16x32_fp8_fp8 (bf16 MFMA rate, half the operand bytes); 2 = v_mfma_scale_f32_16x16x128_f8f6f4 with
 *   unit block scales (2x the bf16 MFMA rate).  Identical results (exact products, f32 accumulation; summation order differs).
 * epilogue: BSCLIP_EPI_BF16, _F32, _RESID_F32 (+ dropout), _GELU_FP8.  args->bias is required. */
typedef struct bsclip_fp8_args {
    uint32_t struct_size; /* = sizeof(bsclip_fp8_args) */
    int form;
    const float* alpha;
    const void* a_aug;
    int ld_a_aug;
    const void* b_aug;
    int ld_b_aug;
} bsclip_fp8_args;
int bsclip_gemm_fp8(const void* A8, int lda, const void* B8, int ldb, void* C, int ldc, int M, int N, int K, int epilogue,
                    const bsclip_epi_args* args, const bsclip_fp8_args* f8, void* stream);
/* Row-wise quantisation of frozen weights: src f32 [R, C] -> dst fp8 e4m3 [R, ld_dst], scale[r] = amax(src[r,:]) / 448
 * (1 for an all-zero row), dst = round(src / scale[r]).  C % 4 == 0. */
int bsclip_quantize_rows_fp8(const float* src, int R, int C, void* dst, int ld_dst, float* scale, void* stream);
/* b_aug[3H, 64] bf16 (zero outside the LoRA columns): cols [0,4) of rows [0,H) = B_q / alpha, cols [4,8) of rows [2H,3H) =
 * B_v / alpha, refreshed every step from the f32 masters (the fp8 counterpart of bsclip_waug_set_lora_layers). */
int bsclip_lora_baug_set(void* b_aug, int ld_b, int H, const float* lora_bq, const float* lora_bv, const float* alpha,
                         void* stream);
/* one-time device tables (GELU Phi/phi table of the 256x256 kernel's epilogue).  bsclip_gemm_bf16 fills them lazily on
 * its own stream; call this once (and synchronise) before launching GEMMs from several streams. */
int bsclip_init_tables(void* stream);
/* tile override for benchmarking: 0 = auto, 1 = 128x128, 2 = 256x128, 3 = 256x256, 4 = 256x256 ping-pong,
 * 5 = the 256x128 kernel that runs two workgroups per CU (csrc/gemm_duo.h), 8 = the persistent form of 4 (one workgroup per CU
 * walks the tiles, the next tile's first K-tile staged under the current one's last; csrc/gemm_pers.h); the diagnostic
 * library (-DBSCLIP_DIAG, `make diag`) adds 6 = ping-pong with the LDS-DMA two K-tiles ahead (slower) and 7 = four instead of
 * eight barriers per K-tile (equal), kept for comparison only */
int bsclip_gemm_set_tile(int tile);
/* workgroups of the persistent kernel's launch: 0 = one per CU (the default); a small number makes every workgroup walk many
 * tiles of a small problem (tests) */
int bsclip_gemm_set_persistent_grid(int workgroups);
#ifdef BSCLIP_DIAG
/* ---- diagnostic builds: only in libbsclip_hip_diag.so (`make -C bioscan-clip_amd/csrc diag`), never in the product library.
 * bsclip_gemm_diag: the 256x256 kernel with per-workgroup phase stamps (start, prologue, K loop, end, epilogue sections) in
 * 100 MHz ticks, diag[grid * 16]; tools/gemm_phases.py.
 * bsclip_gemm_diag_ablate: which parts of the K loop the diagnostic EPI_BF16 build leaves out (1 MFMA, 2 LDS reads, 4 DMA,
 * 8 barriers; sums of two for the instantiated pairs): timing experiments only, results are garbage.  tools/gemm_ablate.py */
int bsclip_gemm_diag_ablate(int mask);
int bsclip_gemm_diag(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                     int epilogue, const bsclip_epi_args* args, unsigned long long* diag, void* stream);
/* the 256x128 two-workgroups-per-CU kernel (bsclip_gemm_set_tile(5)) with per-workgroup stamps, diag[grid * 8] = {start, tile 0
 * landed, K loop done, end, HW_ID, XCC_ID, -, -}: section times and co-residency; tools/gemm_duo_phases.py */
int bsclip_gemm_duo_diag(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                         int epilogue, const bsclip_epi_args* args, unsigned long long* diag, void* stream);
/* the persistent 256x256 kernel (bsclip_gemm_set_tile(8)) on `workgroups` workgroups with per-workgroup stamps, diag[grid * 32] =
 * {start, end, tiles done, -, second tile: K loop start, K loop end, after each of the 8 epilogue barriers, first tile: K loop
 * start, K loop end, second tile: end of K-tile 0..15}; tools/gemm_pers_phases.py */
int bsclip_gemm_pers_diag(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                          int epilogue, const bsclip_epi_args* args, unsigned long long* diag, int workgroups, void* stream);
/* the attention backward kernel (S = 197 or 133, no mask, no dropout) with per-wave section stamps in 100 MHz ticks,
 * diag[B*heads*4*8]; tools/attn_phases.py */
int bsclip_attn_bwd_diag(const void* qkv, int ld_qkv, const void* dctx, int ld_ctx, const float* lse, int B, int S,
                         int heads, float scale, void* dqkv, int ld_dqkv, unsigned long long* diag, void* stream);
/* Round 4 experiment, NOT a product kernel since ABI 9 (the pair is a wash against bsclip_attn_fwd / bsclip_attn_bwd, docs/HISTORY.md 6.0):
 * the same attention with a backward that forms every product once (csrc/attn_sweep.hip: key-owner waves sweep the query
 * blocks, a dQ wave follows; 20 MFMAs + 16 exp per 32x32 tile pair instead of 32 + 32).  delta = rowsum(P . dP) is taken from
 * the forward's OUTPUT, which therefore leaves O to 16 mantissa bits -- ctx = bf16(O) (the out-projection's operand, as before)
 * and ctx_lo = bf16(O - ctx), same layout -- and per query row stats[B, heads, S, 4] f32 = (nm2, inv, rZ, 0): e_k = exp2(s_k
 * scale log2e + nm2), inv = 1 / sum_k e_k, rZ = sum_k e_k / Z' where Z' sums the bf16-ROUNDED operands of the P.V product
 * (dropped keys: their e_k), so that sum_k dS_k = 0 holds to f32 rounding with delta = (dctx . O) rZ.  stats 16-byte aligned.
 * No q_rows form (every query row carries a gradient); argument meaning otherwise as bsclip_attn_fwd / bsclip_attn_bwd. */
int bsclip_attn_fwd2(const void* qkv, int ld_qkv, int B, int S, int heads, const float* key_bias, float scale, void* ctx,
                     void* ctx_lo, int ld_ctx, float* stats, float dropout_p, uint32_t dropout_seed, void* stream);
int bsclip_attn_bwd2(const void* qkv, int ld_qkv, const void* dctx, int ld_dctx, const void* ctx, const void* ctx_lo, int ld_ctx,
                     const float* stats, int B, int S, int heads, const float* key_bias, float scale, void* dqkv, int ld_dqkv,
                     float dropout_p, uint32_t dropout_seed, void* stream);
/* the key-owner-sweep backward (bsclip_attn_bwd2; S = 197 or 133, no mask, no dropout) with per-wave section stamps,
 * diag[B*heads*(NB+1)*8], NB = ceil(S / 32) key-owner waves + the dQ wave; tools/attn_sweep_phases.py */
int bsclip_attn_bwd2_diag(const void* qkv, int ld_qkv, const void* dctx, int ld_dctx, const void* ctx, const void* ctx_lo,
                          int ld_ctx, const float* stats, int B, int S, int heads, float scale, void* dqkv, int ld_dqkv,
                          unsigned long long* diag, void* stream);
/* the persistent form of the same (csrc/attn_pers.hip): per-wave stamps of each workgroup's second item, diag[grid * (NB+1) * 16];
 * tools/attn_pers_phases.py */
int bsclip_attn_bwd_pers_diag(const void* qkv, int ld_qkv, const void* dctx, int ld_dctx, const void* ctx, const void* ctx_lo,
                              int ld_ctx, const float* stats, int B, int S, int heads, float scale, void* dqkv, int ld_dqkv,
                              unsigned long long* diag, void* stream);
#endif

/* ---- LayerNorm (timm norm1/norm2/norm eps 1e-6; HF BertLayerNorm eps 1e-12) ------------------------------------
 * fwd: y = LN(x) * gamma + beta for f32 rows x[M,H] (H = 768 or 512).
 *   y_bf16 [M, ld_y]  (nullable): bf16(y) in cols [0,H); if lora_a != NULL cols [H,H+8) = bf16(y . lora_a^T) and
 *                     cols [H+8, H+BSCLIP_KPAD) = 0 (the K-augmentation block consumed by the QKV GEMM).
 *   y_f32  [M, H]     (nullable): f32 copy (post-LN residual stream of BERT).
 *   y_split3 [M, ld_y3 >= 3H] bf16 (nullable; exact mode, ABI 9): the output as the A operand of a split-bf16 GEMM, [hi | lo | hi] with
 *                     hi = bf16(y), lo = bf16(y - hi) -- what bsclip_split3_rows makes of y_f32 in a second pass.
 *   stats  [M, 2]     (nullable): (mean, rstd) saved for backward.
 *   x_bf16 != 0: x is bf16 [M, ld_x] instead of f32 (MLM transform LN after GELU).
 * bwd (gamma/beta frozen -> only dx): dy = g_resid(f32, nullable) + g_gemm(bf16, nullable) + dt[M,8] . lora_a
 *   mode 0 (pre-LN, ViT):  dx = g_resid + LNbwd(g_gemm + dt.lora_a)
 *   mode 1 (post-LN, BERT): dx = LNbwd(g_resid + g_gemm + dt.lora_a)
 *   Dropout (HF BERT, K15): fwd applies it to y (both copies; the embeddings LayerNorm); bwd applies it to dx_bf16 only --
 *   dx_bf16 is the operand of the dX GEMM of a Linear whose FORWARD output was dropped with the same (p, seed), and the
 *   mask of element (row, col) is a pure function of (seed, row*H + col), so it is regenerated, never stored.
 *   writes dx_f32 [M, ld_dx] (nullable) and dx_bf16 [M, ld_dxb] (nullable).  Row strides let the final ViT norm run on
 *   the token-0 rows only (x, g and dx all strided by 197*H).  stats is indexed by the compact row number.
 *   in_dropout_p > 0 (full fine-tuning, BertEmbeddings): the LN's own OUTPUT was dropped in forward with (p, seed); the
 *   assembled dy is masked the same way before it is differentiated. */
int bsclip_layernorm_fwd(const void* x, int ld_x, int x_bf16, int M, int H, const float* gamma, const float* beta,
                         float eps, void* y_bf16, int ld_y, float* y_f32, void* y_split3, int ld_y3, const float* lora_a, float* stats,
                         float dropout_p, uint32_t dropout_seed, void* stream);
/* fp8 operand variant (configs[4]): y_fp8 [M, ld_y bytes] = e4m3(LN(x)) (scale 1, saturated), t_aug bf16 [M, ld_t] = the LoRA
 * block (t in cols [0,8), zeros to col 64) when lora_a != NULL -- the operands of bsclip_gemm_fp8.  Other arguments as above. */
int bsclip_layernorm_fwd_fp8(const void* x, int ld_x, int x_bf16, int M, int H, const float* gamma, const float* beta,
                             float eps, void* y_fp8, int ld_y, void* t_aug, int ld_t, float* y_f32, const float* lora_a,
                             float* stats, float dropout_p, uint32_t dropout_seed, void* stream);
/* resid_flags: the residual-gradient stream (g_resid in, dx_f32 out) may be kept in bf16 -- bit 0: g_resid is bf16 [M, ld_gr],
 *   bit 1: dx_f32 points to a bf16 [M, ld_dx] buffer (the undropped gradient, rounded once); 0 = both f32.
 *   Exact mode: bit 2: g_gemm is f32; bit 3: the operand output dx_bf16 is f32 [M, ld_dxb]; bit 4 (ABI 9, excludes bit 3): the operand
 *   output is the split-bf16 GEMM operand [hi | lo | hi], bf16 [M, ld_dxb >= 3H]. */
int bsclip_layernorm_bwd(const void* x, int ld_x, int x_bf16, const float* stats, const float* gamma, int M, int H,
                         const void* g_resid, int ld_gr, const void* g_gemm, int ld_g, const float* dt,
                         const float* lora_a, int mode, void* dx_f32, int ld_dx, void* dx_bf16, int ld_dxb,
                         float dropout_p, uint32_t dropout_seed, float in_dropout_p, uint32_t in_dropout_seed,
                         int resid_flags, void* stream);

/* ---- self-attention (timm Attention.forward; HF BertSelfAttention) ---------------------------------------------
 * qkv bf16 [B*S, ld_qkv] with columns [q | k | v], each heads*64 wide; ctx bf16 [B*S, ld_ctx];
 * key_bias f32 [B,S] additive per key (HF extended attention mask, language_encoder.py:89) or NULL;
 * lse f32 [B, heads, S] = log-sum-exp of the scaled scores (saved for backward).  head_dim is 64.
 * dropout_p > 0: HF attention_probs_dropout_prob on the normalised probabilities (mask of (b,head,q,k) regenerated in bwd).
 * q_rows: 0 = every query; n > 0 = only the first n query rows of each sequence matter (rounded up to a block of 32):
 * fwd writes ctx/lse for those rows only, bwd takes dctx as zero on the others and writes zeros into their dq rows (the last
 * ViT block, whose output is read at token 0 only: image_encoder.py:108-109 -> timm global_pool='token').
 * S <= 224.  bwd recomputes P from qkv + lse, forms delta = rowsum(P . dP) in f32 from the same tiles (not from the
 * bf16-rounded ctx: that loses the softmax-backward cancellation), and writes dqkv in the same layout as qkv.
 * keep_bits (ABI 9, nullable; read / written only when dropout_p > 0): uint32 [B * heads, S, 2, 8], 16-byte aligned -- bit 8 g + i of word
 * kt of half h of row (b, head, q) = "key 32 kt + 8 g + 4 h + i of that query row was kept" (the key set one lane half of the kernels holds;
 * words kt >= ceil(S / 32) are unused).  The forward writes the words from the decisions it hashes; a
 * backward that is handed them reads the decisions instead of re-hashing (the same masks, hence the same gradients bit for bit;
 * the key-owner phase hashed once per element: a quarter of the S = 133 launch).  Without them the backward re-hashes, as before. */
int bsclip_attn_fwd(const void* qkv, int ld_qkv, int B, int S, int heads, const float* key_bias, float scale,
                    void* ctx, int ld_ctx, float* lse, int q_rows, void* keep_bits, float dropout_p, uint32_t dropout_seed,
                    void* stream);
int bsclip_attn_bwd(const void* qkv, int ld_qkv, const void* dctx, int ld_ctx, const float* lse, int B, int S,
                    int heads, const float* key_bias, float scale, void* dqkv, int ld_dqkv, int q_rows, const void* keep_bits,
                    float dropout_p, uint32_t dropout_seed, void* stream);
/* bsclip_attn_bwd that also leaves, from dq and dv while they are still in the accumulators, the partial sums of the rank-4 LoRA
 * gradients on q and v (ABI 10; the first pass of bsclip_lora_grad re-read dq and dv from HBM for them):
 *   dt_partial f32 [heads, 2, B*S, 4]:    [h][0][m][0:4] = dq[m, head h] . B_q[head h],  [h][1][m][0:4] = dv[m, head h] . B_v[head h]
 *   db_partial f32 [B*heads, 2, 4, 64]:   [b*heads+h][q|v][j][d] = sum over the tokens of sequence b of t[m][j (+4 for v)] * dq|dv[m][64 h + d]
 * t_aug: bf16 [B*S, ld_t], t = y A^T in columns 0..7 (the LayerNorm's block of the QKV operand), 16-byte aligned, ld_t % 8 == 0;
 * lora_b f32 [2, heads*64, 4] as bsclip_lora_grad.  With dropout the forward's keep_bits are required.  The products run on the matrix
 * pipe with B as hi + lo bf16 parts: f32 results up to accumulation order.  bsclip_lora_grad_heads reduces the partials. */
int bsclip_attn_bwd_lora(const void* qkv, int ld_qkv, const void* dctx, int ld_ctx, const float* lse, int B, int S, int heads,
                         const float* key_bias, float scale, void* dqkv, int ld_dqkv, int q_rows, const void* keep_bits,
                         const void* t_aug, int ld_t, const float* lora_b, float* dt_partial, float* db_partial, float dropout_p,
                         uint32_t dropout_seed, void* stream);

/* ---- "exact" forward mode (BSCLIP_PARITY=2; csrc/exact.hip): every trunk GEMM on split-bf16 operands, f32 attention -----------
 * A bf16 MFMA GEMM is exact to ~2^-16 when both operands are carried as hi + lo (hi = bf16(x), lo = bf16(x - hi)) and the product is
 * formed as hi.hi + lo.hi + hi.lo: ONE bsclip_gemm_bf16 call with K tripled, A rows [hi | lo | hi] against W rows [hi | hi | lo].
 *   split3_rows:   src f32 [M, ld_src >= K] -> dst bf16 [M, ld_dst >= 3K] = [hi | lo | hi]   (K % 4 == 0)
 *   split3_weight: w f32 [N, K] -> dst bf16 [N, 3K] = [hi | hi | lo]; with lora_a f32 [8, K] (A_q rows 0..3, A_v rows 4..7) and lora_b
 *                  f32 [2, H, 4] the LoRA update is folded in f32 first (reference image_encoder.py:43-47, dna_encoder.py:47-49):
 *                  rows [0, H) += B_q A_q, rows [2H, 3H) += B_v A_v  (N = 3H, K = H)
 *   gelu_split3:   z f32 [M, N] -> exact (erf) GELU as bf16 [M, 3N] = [hi | lo | hi] (dst nullable) and / or as f32 [M, ld_g32] (g32
 *                  nullable; the MLM transform's GELU feeds a LayerNorm, not a GEMM); codes (nullable) u8 [M, ld_codes]: the 8-bit
 *                  gelu' side band the default backward reads
 *   meanpool_tokens_f32: x f32 [B*S, H] -> out f32 [B, H], the mean over each sequence's S tokens (language_encoder.py:89)
 *   attn_fwd_f32:  qkv f32 [B*S, ld_qkv] ([q | k | v], heads*64 each) -> ctx f32 [B*S, ld_ctx], lse f32 [B, heads, S]; f32 softmax
 *                  arithmetic, every product exact to ~2^-16 (bsclip_exact_attn_set_impl: split-bf16 operands by default, f32-operand
 *                  MFMA / vector ALU as second implementations; S <= 224), the bf16 kernels' dropout masks; argument meaning as
 *                  bsclip_attn_fwd.  ctx_split3 (nullable): ctx once more as bf16 [B*S, ld_c3 >= 3 heads 64] = [hi | lo | hi], the
 *                  out-projection GEMM's split operand (no stand-alone bsclip_split3_rows pass)
 * The exact backward (BSCLIP_PARITY=2) keeps every gradient in f32 and runs its dX / dW GEMMs on split operands as well:
 *   dgelu_split3:  dact f32 [M, N], z f32 [M, N] (fc1 pre-activation) -> dact * gelu'(z) as bf16 [M, 3N] = [hi | lo | hi] (dst nullable)
 *                  and / or f32 [M, ld_out32] (out32 nullable)
 *   split3_transpose: src f32 [R, C] -> dst bf16 [C, 3 Rp]: row c = the split of column c over Rp >= R positions (Rp % 64 == 0, zeros
 *                  beyond R), order 0 = [hi | lo | hi], 1 = [hi | hi | lo].  With order 0 / 1 on dY / X it builds the two operands of a
 *                  dW = dY^T X GEMM (the reduction runs over the R rows); with order 1 on a frozen weight [N, K] it builds W^T for the
 *                  dX GEMM; lora_a / lora_b (nullable, R = 3H, C = H) fold W + B A as in split3_weight
 *   softmax_meanpool_bwd_f32: bsclip_softmax_meanpool_bwd with f32 dlogits [B*S, ld_d]
 *   lora_grad_f32: dA [8, H] += (B^T dq | B^T dv) y^T, dB [2, H, 4] += (dq | dv)^T (A y) from dqkv f32 [M, ld_dqkv] ([dq | dk | dv]) and
 *                  the LayerNorm output y f32 [M, ld_y] (reference lora_layer.py:16-39); workspace:
 *                  bsclip_lora_grad_f32_workspace_floats(M, H) floats (t and dt [M, 8] + the per-workgroup slabs); sums in a fixed order
 *                  (round 5: t = A y by one pass, then bsclip_lora_grad's pipelined wave-per-row kernels on f32 rows)
 *   attn_bwd_f32:  dqkv f32 [B*S, ld_dqkv] from qkv f32, dctx f32, the forward's ctx f32 (delta = dctx . ctx) and lse; the forward's
 *                  implementation and dropout masks; argument meaning as bsclip_attn_bwd.  dqkv_split3 (nullable): [dq | dk | dv] once
 *                  more as bf16 [B*S, ld_d3 >= 9 heads 64] = [hi | lo | hi], the QKV dX GEMM's split operand
 * bsclip_layernorm_bwd takes the f32 GEMM gradient / writes the f32 or the split operand through resid_flags bits 2 / 3 / 4. */
int bsclip_split3_rows(const float* src, int ld_src, int M, int K, void* dst, int ld_dst, void* stream);
int bsclip_split3_weight(const float* w, int ld_w, int N, int K, const float* lora_a, const float* lora_b, int H, void* dst, int ld_dst,
                         void* stream);
int bsclip_gelu_split3(const float* z, int ld_z, int M, int N, void* dst, int ld_dst, void* codes, int ld_codes, float* g32, int ld_g32,
                       void* stream);
int bsclip_meanpool_tokens_f32(const float* x, int B, int S, int H, float* out, void* stream);
int bsclip_attn_fwd_f32(const float* qkv, int ld_qkv, int B, int S, int heads, const float* key_bias, float scale, float* ctx, int ld_ctx,
                        float* lse, void* ctx_split3, int ld_c3, float dropout_p, uint32_t dropout_seed, void* stream);
int bsclip_dgelu_split3(const float* dact, int ld_dact, const float* z, int ld_z, int M, int N, void* dst, int ld_dst, float* out32,
                        int ld_out32, void* stream);
int bsclip_split3_transpose(const float* src, int ld_src, int R, int C, int Rp, int order, const float* lora_a, const float* lora_b, int H,
                            void* dst, int ld_dst, void* stream);
int bsclip_softmax_meanpool_bwd_f32(const float* logits, const float* stats, const float* d_pooled, int B, int S, int C, float* dlogits,
                                    int ld_d, void* stream);
int64_t bsclip_lora_grad_f32_workspace_floats(int M, int H);
int bsclip_lora_grad_f32(const float* dqkv, int ld_dqkv, const float* y, int ld_y, int M, int H, const float* lora_a, const float* lora_b,
                         float* dA, float* dB, float* workspace, void* stream);
/* 0 (default, round 5) = split-bf16 operands on the bf16 matrix cores (csrc/attn_x3.hip: every product as hi.hi + lo.hi + hi.lo, f32 softmax
 * arithmetic; ~2^-16 per product, 16 x the f32 MFMA rate), 2 = f32 operands on the matrix pipe (v_mfma_f32_32x32x2_f32, round 4's default),
 * 1 = the one-row-per-thread vector-ALU kernels -- 1 and 2 are exact-f32 second implementations kept to test against */
int bsclip_exact_attn_set_impl(int impl);
int bsclip_attn_bwd_f32(const float* qkv, int ld_qkv, const float* dctx, int ld_dctx, const float* ctx, int ld_ctx, const float* lse, int B,
                        int S, int heads, const float* key_bias, float scale, float* dqkv, int ld_dqkv, void* dqkv_split3, int ld_d3,
                        float dropout_p, uint32_t dropout_seed, void* stream);

/* ---- embeddings ------------------------------------------------------------------------------------------------
 * im2col for timm PatchEmbed Conv2d(3,768,k=16,s=16): image f32 [B,3,224,224] -> bf16 [B*196, 768], column
 * order (c, ky, kx) = conv weight.flatten(1).  cls rows: x[b*197] = cls + pos[0].
 * bert_embed: word[id] + pos[t] + type[tt] -> f32 [B*S, H] (HF BertEmbeddings before LayerNorm). */
/* split != 0: each row is written as [hi | lo | hi] (3 x 768 columns, ld_cols >= 2304) for the split-bf16 patch-embed GEMM
 * against a weight stored as [hi | hi | lo] (K = 2304: hi.hi + lo.hi + hi.lo, ~2^-16 relative instead of 2^-8) */
int bsclip_im2col_patch16(const float* image, int B, void* cols_bf16, int ld_cols, int split, void* stream);
/* HF extended attention mask (BertModel, language_encoder.py:89): bias[i] = mask[i] ? 0 : finfo(f32).min */
int bsclip_mask_to_bias(const int64_t* mask, int n, float* bias, void* stream);
/* x[b*S, :] = cls_token + pos_embed[0] (x f32, or bf16 when x_bf16 != 0: the bf16 residual stream) */
int bsclip_vit_cls_rows(void* x, int x_bf16, const float* cls_token, const float* pos_embed, int B, int S, int H, void* stream);
int bsclip_bert_embed(const int64_t* ids, const int64_t* type_ids, int B, int S, int H, const float* word,
                      int vocab, const float* pos, const float* type, float* out, void* stream);

/* ---- heads -----------------------------------------------------------------------------------------------------
 * softmax_meanpool: LoRA_barcode_bert.forward `logits.softmax(-1).mean(1)` (dna_encoder.py:105):
 *   logits f32 [B*S, C] -> pooled f32 [B, C]; stats [B*S, 2] = (row max, sum exp) for backward.
 *   bwd: d_pooled f32 [B,C] -> dlogits bf16 [B*S, ld_d].
 * meanpool_tokens: LoRA_bert.forward `last_hidden_state.mean(1)` (language_encoder.py:89): f32 [B,S,H] -> bf16 [B,H].
 * l2norm: F.normalize(p=2, dim=-1, eps=1e-12) (simple_clip.py:34,47,49). */
int bsclip_softmax_meanpool_fwd(const float* logits, int B, int S, int C, float* pooled, float* stats, void* stream);
int bsclip_softmax_meanpool_bwd(const float* logits, const float* stats, const float* d_pooled, int B, int S, int C,
                                void* dlogits_bf16, int ld_d, void* stream);
int bsclip_meanpool_tokens_fwd(const float* x, int B, int S, int H, void* out_bf16, int ld_out, void* stream);
/* autograd of the mean: dx[b,t,:] = d_pooled[b,:] / S  (f32 [B*S, H]) */
int bsclip_meanpool_tokens_bwd(const float* d_pooled, int ld_d, int B, int S, int H, float* dx, void* stream);
/* out = g * z elementwise (g, out bf16 [M,N]; z uint8 codes), z = gelu'(pre-activation) as saved by BSCLIP_EPI_GELU_BF16: autograd of the
 * GELU inside cls.predictions.transform, where the producer of g is the LayerNorm backward rather than a GEMM. */
int bsclip_dgelu_mul(const void* g, int ld_g, const void* z, int ld_z, int M, int N, void* out, int ld_o, void* stream);
int bsclip_l2norm_fwd(const float* x, int M, int D, float* y, float* inv_norm, void* stream);
int bsclip_l2norm_bwd(const float* y, const float* inv_norm, const float* dy, int M, int D, float* dx, void* stream);

/* ---- contrastive loss ------------------------------------------------------------------------------------------
 * ContrastiveLoss.forward (bioscanclip/model/loss_func.py:29-54) with construct_label_metrix (:18-21) for
 * nmod = 2 or 3 modalities z[i] f32 [N, D] (D = 768).  The second F.normalize (:43-44) is applied inside
 * (forward and backward).  loss_out[0] = mean over all 2*nmod*(nmod-1) CE terms.
 * dz[i] f32 [n_local, D] receives dLoss/dz[i] for rows [row0, row0 + n_local) only (the rank's own slice of an
 * all-gathered batch); pass row0 = 0, n_local = N for the local-batch loss.
 * workspace: f32, at least bsclip_infonce_workspace_floats(N, nmod) elements. */
int64_t bsclip_infonce_workspace_floats(int N, int nmod);
/* implementation switch for benchmarking / cross-checking: 2 = fused -- the logits tile stays in the MFMA
 * accumulators, row max / log-sum-exp reduced with wave shuffles in the GEMM epilogue, pass 2 emits dL/dG from the
 * accumulators; 1 = logits formed in f32 row slabs in the workspace and reduced by separate kernels (round 1);
 * 0 (default) = by size: fused from 2 048 (padded) rows up, slabs below, where a product is too few 256 x 256 tiles to fill the chip */
int bsclip_infonce_set_impl(int impl);
int bsclip_infonce_fwd_bwd(const float* const* z, int nmod, const int64_t* labels, int N, int D, float scale,
                           int row0, int n_local, float* loss_out, float* const* dz, float* workspace,
                           void* stream);

/* ---- retrieval --------------------------------------------------------------------------------
 * make_prediction (scripts/inference_and_eval.py:414-445): faiss IndexFlatIP over L2-normalised keys f32 [K, D], searched
 * with L2-normalised queries f32 [Q, D] (sklearn normalize, :416-417).  Outputs the top k (<= 16) inner products per query in
 * descending order (ties: lower key index first): scores_out f32 [Q, k], idx_out int64 [Q, k].  D % 64 == 0. */
int64_t bsclip_topk_ip_workspace_floats(int Q, int K, int D);
int bsclip_topk_ip(const float* queries, int Q, const float* keys, int K, int D, int k, float* scores_out,
                   int64_t* idx_out, float* workspace, void* stream);

/* ---- RCCL collectives of the global-batch step ---------------------------------------------------
 * One process per GPU.  bsclip_comm_unique_id on rank 0 -> the caller ships the bsclip_comm_unique_id_bytes() bytes to the
 * other ranks (any channel) -> bsclip_comm_init on every rank (ncclCommInitRank).  The collectives run on `comm_stream`;
 * `wait_event` (hipEvent_t as void*, nullable) is waited for on that stream first -- record it on the stream that produced
 * the payload -- and `done_event` (nullable) is recorded behind the collective for the consumer to wait on.  No host sync.
 *   allgather_embeddings: gather_features (bioscanclip/model/loss_func.py:58-91, :84-89): local f32 [count] (= B*D) ->
 *     gathered f32 [world*count], rank-major = row-major [world*B, D]; remote rows carry no gradient.
 *   allgather_labels: the label all-gather of ClipLoss (loss_func.py:122), int64.
 *   allreduce_grads: SUM over ranks of a flat f32 trainable-gradient buffer, in place (the gradient synchronisation
 *     scripts/train_cl.py lacks).
 * RCCL is bound with dlopen at first use; every call fails with a message when it cannot be found. */
int bsclip_comm_unique_id_bytes(void);
int bsclip_comm_unique_id(void* id_out);
int bsclip_comm_init(void** comm_out, const void* unique_id, int rank, int world);
int bsclip_comm_destroy(void* comm);
int bsclip_allgather_embeddings(void* comm, const float* local, float* gathered, int64_t count, void* comm_stream,
                                void* wait_event, void* done_event);
int bsclip_allgather_labels(void* comm, const int64_t* local, int64_t* gathered, int64_t count, void* comm_stream,
                            void* wait_event, void* done_event);
int bsclip_allreduce_grads(void* comm, float* grads, int64_t count, void* comm_stream, void* wait_event, void* done_event);

/* ---- input pipeline --------------------------------------------------------------------------------
 * kmer_tokenize: get_sequence_pipeline(k) (bioscanclip/model/dna_encoder.py:25-35; PadSequence / KmerTokenizer,
 *   bioscanclip/util/util.py:48-69).  seqs: the batch's nucleotide bytes back to back, offsets int64 [B+1]; each sequence is
 *   truncated to max_len or right-padded with 'N', cut into max_len/k non-overlapping k-mers, id = 3 + base-4 value over
 *   A,C,G,T (any other byte in the k-mer -> 2 = <UNK>), and a literal 0 (<MASK>) is put in front: ids int64 [B, max_len/k + 1].
 * augment_images: the image transforms of Dataset_for_CL (bioscanclip/util/dataset.py:171-200) for B decoded uint8 HWC images
 *   stored back to back in src_u8.  records int32 [B, 16] per image: {src offset lo, hi, H0, W0, H1, W1 (size after
 *   Resize(256)), crop top, left, height, width (in the resized image), hflip, vflip, rotate flag, cos(angle), sin(angle) as
 *   f32 bits, 0}.  mid: f32 scratch [B, 3, mid_capacity] (mid_capacity >= max H1*W1); out: f32 [B, 3, out_size, out_size].
 *   Arithmetic: ToTensor, antialiased bilinear resize (torch _upsample_bilinear2d_aa) twice, flips, nearest-neighbour
 *   rotation with zeros outside (torchvision functional_tensor.rotate).  The random draws are the caller's. */
int bsclip_kmer_tokenize(const void* seqs, const int64_t* offsets, int B, int max_len, int k, int64_t* ids, void* stream);
int bsclip_augment_images(const void* src_u8, const int32_t* records, int B, int64_t mid_capacity, float* mid, int out_size,
                          float* out, void* stream);

/* ---- LoRA / head gradients -------------------------------------------------------------------------------------
 * lora_grad: for one layer, from dqkv (bf16 [M, ld_dqkv], q cols [0,H), v cols [2H,3H)) and the augmented LN
 *   output h (bf16 [M, ld_h]: cols [0,H) = y, [H,H+4) = t_q, [H+4,H+8) = t_v):
 *     dt[M,8]   = [dq . B_q | dv . B_v]                     (f32 out, feeds layernorm_bwd)
 *     dA[8,H]  += dt^T h[:, :H]        (rows 0-3 = dA_q, 4-7 = dA_v)      (f32 accumulate)
 *     dBq[H,4] += dq^T t_q ; dBv[H,4] += dv^T t_v                         (f32 accumulate)
 *   lora_b f32 [2, H, 4] = (B_q, B_v).  Autograd of image_encoder.py:44-47 / dna_encoder.py:47-49.
 *   workspace: f32, bsclip_lora_grad_workspace_floats(H) elements (per-workgroup partial slabs; the final sum runs in a
 *   fixed order, so results are bitwise reproducible -- no float atomics).
 * colsum: db[N] += sum_m g[m, n] (bias gradients of the trainable heads), g bf16 or f32.
 * transpose_bf16: out[C,R] = in[R,C]^T (operands of the dW = dY^T X head GEMMs). */
int64_t bsclip_lora_grad_workspace_floats(int H);
int bsclip_lora_grad(const void* dqkv, int ld_dqkv, const void* h, int ld_h, int M, int H, const float* lora_b,
                     float* dt, float* dA, float* dBq, float* dBv, float* workspace, void* stream);
/* lora_grad from the partial sums of bsclip_attn_bwd_lora (B sequences, M = B*S tokens, heads = H / 64): dt[M,8] = sum over heads of
 * dt_partial, dBq / dBv += sum over sequences of db_partial, then dA += dt^T h[:, :H] as bsclip_lora_grad.  Same workspace; fixed
 * summation order (bitwise reproducible). */
int bsclip_lora_grad_heads(const void* h, int ld_h, int M, int H, int B, const float* dt_partial, const float* db_partial, float* dt,
                           float* dA, float* dBq, float* dBv, float* workspace, void* stream);
/* lora_grad with the fp8 operand layout: y_fp8 [M, ld_y bytes] (LN output, e4m3) and t_aug bf16 [M, ld_t] (t in cols [0,8)) */
int bsclip_lora_grad_fp8(const void* dqkv, int ld_dqkv, const void* y_fp8, int ld_y, const void* t_aug, int ld_t, int M,
                         int H, const float* lora_b, float* dt, float* dA, float* dBq, float* dBv, float* workspace,
                         void* stream);
int bsclip_colsum(const void* g, int ld_g, int g_is_bf16, int M, int N, float* out, void* stream);
int bsclip_transpose_bf16(const void* in, int ld_in, int R, int C, void* out, int ld_out, void* stream);
/* the same transpose, and colsum[c] += sum_r in[r, c] from the tiles while they are in registers (ordered sums): with in = dY
 * this is the operand of the weight-gradient GEMM and the bias gradient of a Linear in one pass over dY (full fine-tuning,
 * simple_clip.py:199-201).  workspace: bsclip_transpose_colsum_workspace_floats(R, C) floats; 16-byte aligned operands, ld % 8 == 0. */
int64_t bsclip_transpose_colsum_workspace_floats(int R, int C);
int bsclip_transpose_colsum_bf16(const void* in, int ld_in, int R, int C, void* out, int ld_out, float* colsum,
                                 float* workspace, void* stream);
int bsclip_cast_f32_bf16(const float* in, int64_t n, void* out, void* stream);
/* W_aug[3H, H+KPAD] bf16: cols [H,H+4) of rows [0,H) = B_q, cols [H+4,H+8) of rows [2H,3H) = B_v (refreshed
 * every step from the f32 masters; the frozen [3H,H] block is written once at pack time).
 * One launch for every LoRA layer of an encoder: table_dev[l] = {W_aug, B_q, B_v} as three 64-bit device addresses (the buffers live
 * as long as the encoder: the table is built once); all layers share ld_w and H.  (ABI 9: the one-layer form bsclip_waug_set_lora is
 * gone -- no engine called it.) */
int bsclip_waug_set_lora_layers(const int64_t* table_dev, int layers, int ld_w, int H, void* stream);

/* ---- full fine-tuning only (reference simple_clip.py:199-201 unfreezes every parameter) ----------------
 * ln_param_grad: d_gamma[c] += sum_r dy[r,c] xhat[r,c], d_beta[c] += sum_r dy[r,c] with dy assembled like
 *   bsclip_layernorm_bwd does (g_gemm bf16 + dt.lora_a + (mode 1: g_resid)), optionally masked by the dropout the LN's
 *   OUTPUT was subjected to in forward (BertEmbeddings).  workspace: bsclip_ln_param_grad_workspace_floats(H) floats.
 * embed_grad: autograd of HF BertEmbeddings' three lookups from d_emb f32 [B*S, H] (gradient at the input of its LayerNorm):
 *   d_pos[s] += sum_b (ordered), d_word[id] += rows in token order (one workgroup per vocabulary row scans the ids), d_type[tt] +=
 *   rows through ordered slabs -- no atomics; rows with id == pad_id get none (HF: nn.Embedding(padding_idx = config.pad_token_id
 *   = 0); pass -1 for "no padding row").  workspace: bsclip_embed_grad_workspace_floats(H) floats.
 * gather_cast_rows: dst bf16 [rows_out, H], dst[r] = src[(r / period_out) * period_in + offset + r % period_out] (f32): e.g. the
 *   196 patch rows of each image out of the 197-row residual gradient -> operand of the patch-embedding dW GEMM. */
int64_t bsclip_ln_param_grad_workspace_floats(int H);
int bsclip_ln_param_grad(const void* x, int ld_x, int x_bf16, const float* stats, int M, int H, const float* g_resid,
                         int ld_gr, const void* g_gemm, int ld_g, const float* dt, const float* lora_a, int mode,
                         float in_dropout_p, uint32_t in_dropout_seed, float* d_gamma, float* d_beta, float* workspace,
                         void* stream);
int64_t bsclip_embed_grad_workspace_floats(int H);
int bsclip_embed_grad(const int64_t* ids, const int64_t* type_ids, int B, int S, int H, int vocab, int pad_id,
                      const float* d_emb, float* d_word, float* d_pos, float* d_type, float* workspace, void* stream);
int bsclip_gather_cast_rows(const float* src, int ld_src, int rows_out, int period_in, int period_out, int offset, int H,
                            void* dst_bf16, int ld_dst, void* stream);
/* C[M, N] (f32) += A[M, K] . B[N, K]^T with the reduction cut into `splits` equal K ranges that run as separate workgroups
 * (the weight gradient dW = dY^T X of a Linear, simple_clip.py:199-201: a 768 x 3072 output reduced over 50 432 tokens would
 * otherwise occupy 36 of 256 CUs).  partial: f32 scratch of splits * M * N elements; the ranges are summed in a fixed order
 * (no atomics).  N % 256 == 0, K % (64 * splits) == 0; operands bf16 row-major, 16-byte aligned. */
int bsclip_gemm_splitk_f32(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K, int splits,
                           float* partial, void* stream);

/* ---- optimiser: torch.optim.AdamW defaults (scripts/train_cl.py:158), one launch over a flat f32 buffer ---------- */
int bsclip_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                      float eps, float weight_decay, int step, float grad_scale, void* stream);
/* the same update with the two per-step quantities read from device memory when the kernel runs: hyper_dev[0] = lr (f32),
 * hyper_dev[1] = step count as a uint32 word (advance it with bsclip_counter_add, a capturable launch) -- the form a captured
 * hipGraph replays with a moving LR schedule; neither word is ever read from host memory by an in-flight node */
int bsclip_adamw_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper_dev, float beta1,
                          float beta2, float eps, float weight_decay, float grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif
