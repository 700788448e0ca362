// raw-buffer loads / stores with sc1 (aux = 16) through __builtin_amdgcn_make_buffer_rsrc: do they address what we think?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* p, int nfloats) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, nfloats * 4, 0x00020000);
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16, 0, 16);
  v.x = __builtin_bit_cast(unsigned, __builtin_bit_cast(float, v.x) + 1000.f);
  __builtin_amdgcn_raw_buffer_store_b128(v, r, threadIdx.x * 16 + 64 * 16, 0, 16);
}
int main() {
  float h[512], *d;
  for (int i = 0; i < 512; ++i) h[i] = i;
  (void)hipMalloc(&d, sizeof h); (void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 512);
  (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("out[256..263] = %g %g %g %g %g %g %g %g (expect 1000 1 2 3 1004 5 6 7)\n", h[256], h[257], h[258], h[259], h[260], h[261], h[262], h[263]);
  return 0;
}
