// Candidate structure for the net trunk: ONE wave per SIMD (256 threads), wave = one row tile x BOTH column tiles:
// per operand set 2 activation reads + 2 weight reads (ds_read_b128), 4 v_fma, 8 MFMAs on two accumulators;
// a workgroup barrier every 12 sets (as two per 24-set weight chunk).  Compared with the current 2-waves-per-SIMD
// stream (mfma_stream.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NTH>
__global__ __launch_bounds__(NTH) void k(float* out, unsigned long long* cyc, int sets, float sg) {
  __shared__ __attribute__((aligned(16))) float lds[40960];
  for (int i = threadIdx.x; i < 40960; i += blockDim.x) lds[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  f32x16 acc0, acc1;
  for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* pa = lds + ((wave * 64 + lane) * 4 & 16383);
  const float* pb = lds + 16384 + (lane * 4);
  float4 xa = *reinterpret_cast<const float4*>(pa), xs = *reinterpret_cast<const float4*>(pa + 256),
         xb0 = *reinterpret_cast<const float4*>(pb), xb1 = *reinterpret_cast<const float4*>(pb + 4096);
  float4 ya = xa, ys = xs, yb0 = xb0, yb1 = xb1;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < sets; s += 2) {
#define LOAD(A_, S_, B0_, B1_, K)                                                \
  {                                                                              \
    A_ = *reinterpret_cast<const float4*>(pa + (((s + K) * 64) & 8191));         \
    S_ = *reinterpret_cast<const float4*>(pa + 256 + (((s + K) * 64) & 8191));   \
    B0_ = *reinterpret_cast<const float4*>(pb + (((s + K) * 256) & 8191));       \
    B1_ = *reinterpret_cast<const float4*>(pb + 8192 + (((s + K) * 256) & 8191)); \
  }
#define MF(A_, S_, B0_, B1_)                                                                               \
  {                                                                                                        \
    const float v0 = fmaf(sg, S_.x, A_.x), v1 = fmaf(sg, S_.y, A_.y), v2 = fmaf(sg, S_.z, A_.z), v3 = fmaf(sg, S_.w, A_.w); \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(B0_.x, v0, acc0, 0, 0, 0);                                 \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(B1_.x, v0, acc1, 0, 0, 0);                                 \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(B0_.y, v1, acc0, 0, 0, 0);                                 \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(B1_.y, v1, acc1, 0, 0, 0);                                 \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(B0_.z, v2, acc0, 0, 0, 0);                                 \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(B1_.z, v2, acc1, 0, 0, 0);                                 \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(B0_.w, v3, acc0, 0, 0, 0);                                 \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(B1_.w, v3, acc1, 0, 0, 0);                                 \
  }
    LOAD(ya, ys, yb0, yb1, 1)
    __builtin_amdgcn_sched_barrier(0);
    MF(xa, xs, xb0, xb1)
    __builtin_amdgcn_sched_barrier(0);
    LOAD(xa, xs, xb0, xb1, 2)
    __builtin_amdgcn_sched_barrier(0);
    MF(ya, ys, yb0, yb1)
    __builtin_amdgcn_sched_barrier(0);
    if ((MODE & 4) && (s % 12) == 10) __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
  for (int e = 0; e < 16; ++e) r += acc0[e] + acc1[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, int NTH>
void run(const char* name) {
  const int blocks = 256, sets = 1536;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, blocks * 64);
  hipMemset(cyc, 0, blocks * 64);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<MODE, NTH>), dim3(blocks), dim3(NTH), 0, 0, out, cyc, sets, -1.0f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 8);
  hipMemcpy(h.data(), cyc, blocks * 64, hipMemcpyDeviceToHost);
  double sum = 0; int n = 0;
  for (int b = 0; b < blocks; ++b) for (int w = 0; w < NTH / 64; ++w) { sum += h[b * 8 + w]; ++n; }
  printf("%-64s %.1f cycles per MFMA on the SIMD\n", name, sum / n / (sets * 8.0) / (NTH / 256));
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0, 256>("1 wave/SIMD, sets of 8 MFMA (2 acc), 4 reads + 4 v_fma per set");
  run<4, 256>("  + workgroup barrier every 12 sets");
  run<0, 512>("2 waves/SIMD, same stream");
  run<4, 512>("  + workgroup barrier every 12 sets");
  return 0;
}
