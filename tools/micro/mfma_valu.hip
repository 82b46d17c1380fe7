// Do the matrix pipe and the packed-float32 VALU of a SIMD run side by side?  (The float32 peaks of the two are the same
// 157 TFLOP/s; a convolution that fed both would have twice the float32 rate the net kernel's roofline is priced on.)
// Workgroups of 768 threads: waves 0-7 (two per SIMD) issue dependent v_mfma_f32_32x32x2_f32 chains, waves 8-11 (one per
// SIMD) issue v_pk_fma_f32 on 16 independent register pairs with one operand in SGPRs.  Each stream alone, then both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int DO_M, int DO_V>
__global__ __launch_bounds__(768) void k(float* out, unsigned long long* cyc, int n_mfma, int n_vrounds, float a, float b) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned long long t0 = 0, t1 = 0;
  float r = 0.f;
  __syncthreads();
  if (wave < 8) {
    if (DO_M) {
      f32x16 acc;
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      t0 = __builtin_amdgcn_s_memtime();
      for (int i = 0; i < n_mfma; i += 4) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
      t1 = __builtin_amdgcn_s_memtime();
      for (int e = 0; e < 16; ++e) r += acc[e];
    }
  } else if (DO_V) {
    f2 c[16];
    for (int e = 0; e < 16; ++e) c[e] = f2{(float)lane, (float)e};
    const f2 w = f2{a, b};
    const float x = b * 0.5f;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n_vrounds; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(c[e]) : "v"(f2{x, x}), "s"(w));
    }
    t1 = __builtin_amdgcn_s_memtime();
    for (int e = 0; e < 16; ++e) r += c[e].x + c[e].y;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (lane == 0) cyc[blockIdx.x * 12 + wave] = t1 - t0;
}

template <int DO_M, int DO_V>
void run(const char* name) {
  const int blocks = 256, n_mfma = 4096, n_vr = 4096;   // 4096 MFMAs per wave; 4096 x 16 pk_fma per VALU wave
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, blocks * 768 * 4); (void)hipMalloc(&cyc, blocks * 12 * 8);
  (void)hipMemset(cyc, 0, blocks * 12 * 8);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<DO_M, DO_V>), dim3(blocks), dim3(768), 0, 0, out, cyc, n_mfma, n_vr, 1.0f, 0.5f);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 12);
  (void)hipMemcpy(h.data(), cyc, blocks * 12 * 8, hipMemcpyDeviceToHost);
  double m = 0, v = 0;
  for (int b = 0; b < blocks; ++b) {
    for (int w = 0; w < 8; ++w) m += h[b * 12 + w];
    for (int w = 8; w < 12; ++w) v += h[b * 12 + w];
  }
  m /= blocks * 8; v /= blocks * 4;
  printf("%-44s", name);
  if (DO_M) printf("  MFMA: %.1f cycles per MFMA on the SIMD (2 waves)", m / n_mfma / 2.0);
  if (DO_V) printf("  VALU: %.2f cycles per v_pk_fma_f32 (%.0f %% of one per 4 cycles)", v / (n_vr * 16.0), 100.0 * 4.0 / (v / (n_vr * 16.0)));
  printf("\n");
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run<1, 0>("matrix stream alone");
  run<0, 1>("packed-fma stream alone");
  run<1, 1>("both");
  return 0;
}
