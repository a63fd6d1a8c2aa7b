// What does v_cvt_pk_u8_f32 do with fractions, negatives and overflow?  (rounding mode of the f32 -> u8 conversion)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* x, unsigned* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 1, 0xAABBCCDDu);
}
int main() {
    const float h[] = {-3.f, -0.6f, -0.4f, 0.f, 0.4f, 0.5f, 0.6f, 1.49f, 1.5f, 1.51f, 2.5f, 3.5f, 254.4f, 254.5f, 254.6f, 255.4f, 255.6f, 300.f, 1e9f};
    const int n = sizeof(h) / sizeof(h[0]);
    float* d; unsigned* o; unsigned ho[64];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, n * 4);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
    hipMemcpy(ho, o, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("%10.3f -> byte %3u  word %08x\n", h[i], (ho[i] >> 8) & 0xff, ho[i]);
    return 0;
}
