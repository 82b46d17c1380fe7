// Does a workgroup barrier still cost the matrix pipe when the two waves of a SIMD belong to DIFFERENT workgroups?
// 256-thread workgroups (one wave per SIMD), two of them per compute unit (64 KiB LDS each), same per-wave stream as
// mfma_stream.hip: sets of 4 dependent MFMAs, 3 ds_read_b128 + 4 v_fma per set, a workgroup barrier every BAR sets.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int BAR, int NTH, int LDSF>
__global__ __launch_bounds__(NTH) void k(float* out, unsigned long long* cyc, int sets, float sg) {
  __shared__ __attribute__((aligned(16))) float lds[LDSF];
  for (int i = threadIdx.x; i < LDSF; i += blockDim.x) lds[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* pa = lds + ((wave * 64 + lane) * 4 & 4095);
  const float* pb = lds + 8192 + (lane * 4);
  float4 xa = *reinterpret_cast<const float4*>(pa), xs = *reinterpret_cast<const float4*>(pa + 256),
         xb = *reinterpret_cast<const float4*>(pb);
  float4 ya = xa, ys = xs, yb = xb;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < sets; s += 2) {
#define LOAD(A_, S_, B_, K)                                                      \
  {                                                                              \
    A_ = *reinterpret_cast<const float4*>(pa + (((s + K) * 64) & 4095));         \
    S_ = *reinterpret_cast<const float4*>(pa + 256 + (((s + K) * 64) & 4095));   \
    B_ = *reinterpret_cast<const float4*>(pb + (((s + K) * 256) & 4095));        \
  }
#define MF(A_, S_, B_)                                                                                     \
  {                                                                                                        \
    const float v0 = fmaf(sg, S_.x, A_.x), v1 = fmaf(sg, S_.y, A_.y), v2 = fmaf(sg, S_.z, A_.z), v3 = fmaf(sg, S_.w, A_.w); \
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
    if (BAR > 0 && (s % BAR) == BAR - 2) __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
  for (int e = 0; e < 16; ++e) r += acc[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int BAR, int NTH, int LDSF>
void run(int blocks, const char* name) {
  const int sets = 2048;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, blocks * 64);
  hipMemset(cyc, 0, blocks * 64);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<BAR, NTH, LDSF>), dim3(blocks), dim3(NTH), 0, 0, out, cyc, sets, -1.0f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 8);
  hipMemcpy(h.data(), cyc, blocks * 64, hipMemcpyDeviceToHost);
  double sum = 0; int n = 0;
  for (int b = 0; b < blocks; ++b) for (int w = 0; w < NTH / 64; ++w) { sum += h[b * 8 + w]; ++n; }
  // two waves per SIMD in every configuration below: SIMD cycles per MFMA = wave cycles per MFMA / 2
  printf("%-72s %.1f cycles per MFMA on the SIMD\n", name, sum / n / (sets * 4.0) / 2.0);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0, 512, 40960>(256, "one 512-thread workgroup per CU, no barrier");
  run<8, 512, 40960>(256, "one 512-thread workgroup per CU, barrier every 8 sets");
  run<0, 256, 16384>(512, "two 256-thread workgroups per CU, no barrier");
  run<8, 256, 16384>(512, "two 256-thread workgroups per CU, barrier every 8 sets");
  run<4, 256, 16384>(512, "two 256-thread workgroups per CU, barrier every 4 sets");
  run<8, 256, 16384>(256, "(one 256-thread workgroup per CU, barrier every 8 sets: 1 wave/SIMD; value x2)");
  return 0;
}
