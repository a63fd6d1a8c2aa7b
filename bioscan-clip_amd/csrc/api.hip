// Host-side support of the C ABI: per-thread error string, ABI version.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

static thread_local char g_err[512] = "";

void bsclip_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* bsclip_last_error(void) { return g_err; }
extern "C" int bsclip_abi_version(void) { return 10; }  // 10: bsclip_attn_bwd_lora / bsclip_lora_grad_heads (LoRA dt / dB partial sums out of the attention backward); 9: bsclip_attn_fwd2 / bsclip_attn_bwd2 leave the product library (diagnostic library only); 8: exact mode (split-operand GEMM helpers, f32 attention forward / backward, f32 LoRA gradients; layernorm_bwd resid_flags bits 2 / 3); 7: bsclip_attn_fwd2 / bsclip_attn_bwd2 (key-owner-sweep attention backward); 6: persistent GEMM (set_tile 8, set_persistent_grid), bsclip_clock_probe; 5: adamw_step_dev step word is uint32 (device-advanced), dropout step passes through the mixer; 4: layernorm_bwd in_dropout, fp8 / pipeline / comm / full-FT entry points, 8-bit gelu side band

// ---- dropout step word ------------------------------------------------------------------------------------------
static thread_local const unsigned* g_drop_step = nullptr;
const unsigned* bsclip_current_dropout_step() { return g_drop_step; }

extern "C" int bsclip_set_dropout_step(const uint32_t* step_dev) {
    g_drop_step = step_dev;
    return BSCLIP_OK;
}

namespace {
__global__ void counter_add_kernel(unsigned* p, unsigned inc) { *p += inc; }
}  // namespace

namespace {
// per XCD x: out[2x] = shader-clock counter (s_memtime), out[2x + 1] = 100 MHz real-time counter (s_memrealtime), sampled when the
// stream reaches this node.  The shader-clock counter is per XCD, so every XCD records its own pair (64 one-wave workgroups cover the
// eight XCDs; workgroups of one XCD write near-identical values).
__global__ void clock_probe_kernel(unsigned long long* out) {
    if (threadIdx.x != 0) return;
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF;   // HW_REG_XCC_ID
    out[2 * xcc] = __builtin_readcyclecounter();
    out[2 * xcc + 1] = wall_clock64();
}
}  // namespace

extern "C" int bsclip_clock_probe(unsigned long long* out32_dev, void* stream) {
    BSCLIP_REQUIRE(out32_dev, "bsclip_clock_probe: null pointer");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(64), dim3(64), 0, static_cast<hipStream_t>(stream), out32_dev);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

// *counter += number of Inf / NaN elements of x (exponent field all ones): the check behind BSCLIP_DETECT_ANOMALY, the build's
// counterpart of the reference's torch.autograd.set_detect_anomaly(True) (train_epoch.py:12)
__global__ __launch_bounds__(256) void count_nonfinite_kernel(const void* __restrict__ x, long n, int is_bf16, uint32_t* __restrict__ counter) {
    unsigned bad = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        if (is_bf16) bad += (static_cast<const unsigned short*>(x)[i] & 0x7f80u) == 0x7f80u;
        else bad += (static_cast<const unsigned*>(x)[i] & 0x7f800000u) == 0x7f800000u;
    }
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(counter, bad);
}

extern "C" int bsclip_count_nonfinite(const void* x, int64_t n, int is_bf16, uint32_t* counter_dev, void* stream) {
    BSCLIP_REQUIRE(x && counter_dev && n > 0 && (is_bf16 == 0 || is_bf16 == 1), "bsclip_count_nonfinite: bad arguments");
    long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(count_nonfinite_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, (long)n, is_bf16,
                       counter_dev);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}

extern "C" int bsclip_counter_add(uint32_t* counter_dev, uint32_t inc, void* stream) {
    BSCLIP_REQUIRE(counter_dev, "bsclip_counter_add: null pointer");
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), counter_dev, inc);
    BSCLIP_LAUNCH_CHECK();
    return BSCLIP_OK;
}
