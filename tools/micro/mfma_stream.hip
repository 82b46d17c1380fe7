// What in the net kernel's trunk stream keeps v_mfma_f32_32x32x2_f32 from issuing every 64 cycles?  The stream of
// one wave, rebuilt piece by piece (512 threads = 2 waves per SIMD, as the kernel):
//   sets of 4 dependent MFMAs; + per set 3 ds_read_b128 of the NEXT set issued before them (software pipeline);
//   + the 4 v_fma input transform; + a workgroup barrier every 8 sets; order variants.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int DELAY>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* cyc, int sets, float sg) {
  __shared__ __attribute__((aligned(16))) float lds[40960];
  for (int i = threadIdx.x; i < 40960; i += blockDim.x) lds[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* pa = lds + ((wave * 64 + lane) * 4 & 16383);
  const float* pb = lds + 16384 + (lane * 4);
  float4 xa = *reinterpret_cast<const float4*>(pa), xs = *reinterpret_cast<const float4*>(pa + 256),
         xb = *reinterpret_cast<const float4*>(pb);
  float4 ya = xa, ys = xs, yb = xb;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < sets; s += 2) {
#define LOAD(A_, S_, B_, K)                                                      \
  if (MODE & 1) {                                                                \
    A_ = *reinterpret_cast<const float4*>(pa + (((s + K) * 64) & 8191));         \
    S_ = *reinterpret_cast<const float4*>(pa + 256 + (((s + K) * 64) & 8191));   \
    B_ = *reinterpret_cast<const float4*>(pb + (((s + K) * 256) & 16383));       \
  }
#define MF(A_, S_, B_)                                                                                     \
  {                                                                                                        \
    float v0 = A_.x, v1 = A_.y, v2 = A_.z, v3 = A_.w;                                                       \
    if (MODE & 2) { v0 = fmaf(sg, S_.x, A_.x); v1 = fmaf(sg, S_.y, A_.y); v2 = fmaf(sg, S_.z, A_.z); v3 = fmaf(sg, S_.w, A_.w); } \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(B_.x, v0, acc, 0, 0, 0);                                    \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(B_.y, v1, acc, 0, 0, 0);                                    \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(B_.z, v2, acc, 0, 0, 0);                                    \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(B_.w, v3, acc, 0, 0, 0);                                    \
  }
    LOAD(ya, ys, yb, 1)
    __builtin_amdgcn_sched_barrier(0);
    MF(xa, xs, xb)
    __builtin_amdgcn_sched_barrier(0);
    LOAD(xa, xs, xb, 2)
    __builtin_amdgcn_sched_barrier(0);
    MF(ya, ys, yb)
    __builtin_amdgcn_sched_barrier(0);
    if ((MODE & 4) && (s & 7) == 6) {
      __syncthreads();
      // de-phase the two waves of a SIMD (waves w and w + 4): the later half idles for DELAY x 64 cycles
      if (DELAY > 0 && wave >= 4) __builtin_amdgcn_s_sleep(DELAY);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
  for (int e = 0; e < 16; ++e) r += acc[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, int DELAY = 0>
void run(const char* name) {
  const int blocks = 256, sets = 2048;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, blocks * 64);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<MODE, DELAY>), dim3(blocks), dim3(512), 0, 0, out, cyc, sets, -1.0f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 8);
  hipMemcpy(h.data(), cyc, blocks * 64, hipMemcpyDeviceToHost);
  double sum = 0;
  for (auto x : h) sum += x;
  printf("%-58s %.1f cycles per MFMA on the SIMD\n", name, sum / h.size() / (sets * 4.0) / 2.0);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0>("MFMA sets only");
  run<1>("+ 3 ds_read_b128 per set (next set, ahead of the MFMAs)");
  run<2>("+ 4 v_fma per set");
  run<3>("+ reads + v_fma");
  run<7>("+ reads + v_fma + barrier every 8 sets");
  run<5>("+ reads + barrier every 8 sets");
  run<7, 1>("+ reads + v_fma + barrier every 8 sets, waves 4-7 sleep 64 after it");
  run<7, 2>("... sleep 128");
  run<7, 3>("... sleep 192");
  run<7, 4>("... sleep 256");
  run<7, 6>("... sleep 384");
  return 0;
}
