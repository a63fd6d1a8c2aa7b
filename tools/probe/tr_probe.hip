#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
extern "C" __global__ void k(const unsigned short* in, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[64*64];
  for (int i = threadIdx.x; i < 64*64; i += 64) lds[i] = in[i];
  __syncthreads();
  const int l = threadIdx.x & 63;
  // lane 4q+p of each 16-lane group supplies row q, cols 4p..4p+3 of a 4x16 block; group g -> block at rows 4g.., cols 0..15
  const int g = l >> 4, q = (l >> 2) & 3, p = l & 3;
  auto ptr = (__attribute__((address_space(3))) s16x4*)(lds + (4*g + q) * 64 + 4 * p);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
  for (int j = 0; j < 4; ++j) out[l*4 + j] = (unsigned short)v[j];
}
extern "C" void run_probe(const unsigned short* in, unsigned short* out, void* stream) {
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, (hipStream_t)stream, in, out);
}
