// Can the fp32 matrix work (v_mfma_f32_32x32x2_f32) and the fp32 vector work (v_exp_f32, v_pk_fma_f32, v_pk_add_f32) of the
// attention key-tile loop overlap on one SIMD, and in which arrangement of waves?  One workgroup, W waves per SIMD.
// Work unit = what the attention kernel does per (key tile x 3 q-tiles): 9 MFMAs (3 chains of 3) + 48 exp + 24 pk_add +
// 144 pk_fma.  Arrangements:
//   same     : every wave runs  [MFMA block ; vector block on the MFMA results]  per unit (the kernel today)
//   stagger  : the same, waves of a SIMD start half a unit apart
//   split    : wave 0 of a SIMD runs only MFMA blocks (W-1 units' worth per iteration... see below), the others only vector blocks
//   mfma / valu : one kind alone (floors)
// Prints shader cycles per unit PER SIMD (lower is better): ideal no-overlap = 576 + 1056 = 1632, perfect overlap = 1056.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void mfma_block(f32x16 (&sc)[3], const float (&kf)[3], const float (&qf)[3][3]) {
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 3; ++s) z = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qf[g][s], z, 0, 0, 0);
    sc[g] = z;
  }
}
__device__ __forceinline__ void valu_block(f32x16 (&sc)[3], f32x2 (&acc)[3][3], f32x2 (&lsum)[3], const f32x2 (&vv)[3]) {
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      float p0, p1;
      asm volatile("v_exp_f32 %0, %1" : "=v"(p0) : "v"(sc[g][r]));
      asm volatile("v_exp_f32 %0, %1" : "=v"(p1) : "v"(sc[g][r + 1]));
      sc[g][r] = p0, sc[g][r + 1] = p1;
      asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(lsum[g]) : "v"(f32x2{p0, p1}));
    }
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const f32x2 pp = f32x2{sc[g][r & ~1], sc[g][(r & ~1) + 1]};
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        if (r & 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[g][e]) : "v"(pp), "v"(vv[e]));
        else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[g][e]) : "v"(pp), "v"(vv[e]));
      }
    }
}

// MODE 0 same, 1 stagger, 2 split (wave slot 0 of each SIMD = MFMA only), 3 mfma alone, 4 valu alone
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int units, int W) {
  const int wave = threadIdx.x >> 6;
  const int slot = wave >> 2;  // waves are dealt to the four SIMDs in turn: slot = index among the waves of one SIMD
  float kf[3], qf[3][3];
  for (int s = 0; s < 3; ++s) {
    kf[s] = 1e-3f * (threadIdx.x + s);
    for (int g = 0; g < 3; ++g) qf[g][s] = 1e-3f * (threadIdx.x + 3 * g + s);
  }
  f32x16 sc[3];
  f32x2 acc[3][3], lsum[3], vv[3];
  for (int g = 0; g < 3; ++g) {
    sc[g] = f32x16{0};
    lsum[g] = f32x2{0, 0};
    for (int e = 0; e < 3; ++e) acc[g][e] = f32x2{0, 0};
    vv[g] = f32x2{1.0f + g, 0.5f};
  }
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  if (MODE == 0 || MODE == 1) {
    if (MODE == 1 && slot > 0) {  // start half a unit (slot 1) / a quarter (slot 2 ...) late: vector work first
      for (int i = 0; i < slot; ++i) valu_block(sc, acc, lsum, vv);
    }
    for (int u = 0; u < units; ++u) {
      mfma_block(sc, kf, qf);
      valu_block(sc, acc, lsum, vv);
    }
  } else if (MODE == 2) {
    if (slot == 0) {
      for (int u = 0; u < units * W; ++u) {  // all the matrix work of this SIMD's W units
        mfma_block(sc, kf, qf);
        asm volatile("" ::"v"(sc[0]), "v"(sc[1]), "v"(sc[2]));
      }
    } else {
      const int mine = (units * W) / (W - 1);  // the vector work, shared by the other W - 1 waves
      for (int u = 0; u < mine; ++u) valu_block(sc, acc, lsum, vv);
    }
  } else if (MODE == 3) {
    for (int u = 0; u < units; ++u) {
      mfma_block(sc, kf, qf);
      asm volatile("" ::"v"(sc[0]), "v"(sc[1]), "v"(sc[2]));
    }
  } else {
    for (int u = 0; u < units; ++u) valu_block(sc, acc, lsum, vv);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int g = 0; g < 3; ++g) {
    s += sc[g][0] + lsum[g].x + lsum[g].y;
    for (int e = 0; e < 3; ++e) s += acc[g][e].x + acc[g][e].y;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}

int main() {
  float* o;
  long long* c;
  (void)hipMalloc(&o, 1 << 22);
  (void)hipMalloc(&c, 64 * 16);
  const int units = 400;
  long long h[16];
  const char* names[5] = {"same program, started together", "same program, staggered", "split roles (1 MFMA wave + W-1 vector waves)",
                          "MFMA blocks alone", "vector blocks alone"};
  for (int W = 1; W <= 4; ++W) {
    for (int mode = 0; mode < 5; ++mode) {
      if (mode == 2 && W < 2) continue;
      for (int rep = 0; rep < 2; ++rep) {
        switch (mode) {
          case 0: hipLaunchKernelGGL((k<0>), dim3(1), dim3(256 * W), 0, 0, o, c, units, W); break;
          case 1: hipLaunchKernelGGL((k<1>), dim3(1), dim3(256 * W), 0, 0, o, c, units, W); break;
          case 2: hipLaunchKernelGGL((k<2>), dim3(1), dim3(256 * W), 0, 0, o, c, units, W); break;
          case 3: hipLaunchKernelGGL((k<3>), dim3(1), dim3(256 * W), 0, 0, o, c, units, W); break;
          default: hipLaunchKernelGGL((k<4>), dim3(1), dim3(256 * W), 0, 0, o, c, units, W); break;
        }
        (void)hipMemcpy(h, c, 8 * 4 * W, hipMemcpyDeviceToHost);
      }
      long long mx = 0;
      for (int i = 0; i < 4 * W; ++i) mx = h[i] > mx ? h[i] : mx;
      // units of work finished per SIMD: W x units (modes 0-2), W x units of one kind (3, 4)
      printf("W=%d  %-46s  %8.1f cycles per unit per SIMD   (slowest wave %lld cycles)\n", W, names[mode],
             (double)mx / ((double)units * W), mx);
    }
  }
  return 0;
}
