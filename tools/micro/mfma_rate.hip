// Issue rate of v_mfma_f32_32x32x2_f32 under the conditions of the net kernel's trunk: how many cycles per MFMA
// does one SIMD sustain with W waves per SIMD, each running chains of dependent MFMAs on ACC accumulators?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o gpurun_out/mfma_rate && gpurun_out/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ACC, int GROUP>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, float a0, float b0) {
  f32x16 acc[ACC];
  for (int i = 0; i < ACC; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < ACC; ++i)
#pragma unroll
      for (int g = 0; g < GROUP; ++g) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < ACC; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int ACC, int GROUP>
void run(int threads, const char* name) {
  const int blocks = 256, iters = 2000;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, blocks * 8 * 8);
  hipMemset(cyc, 0, blocks * 64);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<ACC, GROUP>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0f, 0.5f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 8);
  hipMemcpy(h.data(), cyc, blocks * 64, hipMemcpyDeviceToHost);
  double sum = 0; int n = 0;
  for (int b = 0; b < blocks; ++b) for (int w = 0; w < threads / 64; ++w) { sum += h[b * 8 + w]; ++n; }
  const double per_wave = sum / n / (double)(iters * ACC * GROUP);
  const int waves_per_simd = threads / 256;
  printf("%-44s waves/SIMD %d: %.1f cycles per MFMA per wave -> %.1f cycles per MFMA on the SIMD\n", name,
         waves_per_simd ? waves_per_simd : 1, per_wave, per_wave / (waves_per_simd ? waves_per_simd : 1));
  hipFree(out); hipFree(cyc);
}
int main() {
  run<1, 4>(256, "1 accumulator, chains of 4 (1 wave/SIMD)");
  run<1, 4>(512, "1 accumulator, chains of 4 (2 waves/SIMD)");
  run<2, 1>(256, "2 accumulators alternating (1 wave/SIMD)");
  run<2, 1>(512, "2 accumulators alternating (2 waves/SIMD)");
  run<4, 1>(256, "4 accumulators alternating (1 wave/SIMD)");
  run<4, 1>(512, "4 accumulators alternating (2 waves/SIMD)");
  run<2, 4>(256, "2 accumulators, chains of 4 (1 wave/SIMD)");
  return 0;
}
