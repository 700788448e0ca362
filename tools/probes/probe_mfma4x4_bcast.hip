// Probe cbsz/abid (A-operand block broadcast) of v_mfma_f32_4x4x1_16b_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CBSZ, int ABID>
__global__ void k(float* out) {
  int l = threadIdx.x;
  float a = 1.0f + l, b = 1000.0f + l;
  f32x4 c = {0, 0, 0, 0};
  f32x4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, CBSZ, ABID, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
}
static void show(const char* tag, float* d) {
  float h[256]; (void)hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  printf("%s\n", tag);
  for (int l = 0; l < 64; l += 1) {
    int r = 1; float v = h[l * 4 + r]; int fa = -1, fb = -1;
    for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) if ((1.0f + la) * (1000.0f + lb) == v) { fa = la; fb = lb; }
    printf("  lane %2d reg1: A[%2d]*B[%2d]%s", l, fa, fb, (l % 4 == 3) ? "\n" : " |");
  }
}
int main() {
  float* d; (void)hipMalloc(&d, 1024);
  hipLaunchKernelGGL((k<4, 0>), dim3(1), dim3(64), 0, 0, d); show("cbsz=4 abid=0", d);
  hipLaunchKernelGGL((k<4, 5>), dim3(1), dim3(64), 0, 0, d); show("cbsz=4 abid=5", d);
  hipLaunchKernelGGL((k<2, 1>), dim3(1), dim3(64), 0, 0, d); show("cbsz=2 abid=1", d);
  return 0;
}
