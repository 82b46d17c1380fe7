// Candidate trunk stream: the four transformed taps p of the row-Winograd form processed TOGETHER per (dx, channel
// granule): the four input rows d0..d3 are read once (4 ds_read_b128) and give V0 = d0-d2, V1 = d1+d2, V2 = d2-d1,
// V3 = d1-d3; with the four weight granules (4 ds_read_b128) that is 16 MFMAs on four independent accumulators
// for 8 reads -- 0.5 reads per MFMA against 0.75 in the present stream (one p at a time: 2 rows + 1 weight granule
// per 4 MFMAs).  512 threads, 2 waves per SIMD, makespan of the SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int BAR>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* cyc, int sets, float sg) {
  __shared__ __attribute__((aligned(16))) float lds[36864];
  for (int i = threadIdx.x; i < 36864; i += blockDim.x) lds[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  f32x16 m0, m1, m2, m3;
  for (int e = 0; e < 16; ++e) { m0[e] = 0.f; m1[e] = 0.f; m2[e] = 0.f; m3[e] = 0.f; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)(lds + ((wave * 64 + lane) * 4 & 8191));
  const unsigned lb = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)(lds + 16384 + lane * 4);
  f4 D0, D1, D2, D3, W0, W1, W2, W3, E0, E1, E2, E3, X0, X1, X2, X3;
  int s = 0;
#define ALOAD(A0, A1, A2, A3, B0, B1, B2, B3, K)                                                       \
  {                                                                                                    \
    const unsigned oa = la + 4u * (((s + K) * 64) & 4095), ob = lb + 4u * (((s + K) * 256) & 4095);    \
    asm volatile("ds_read_b128 %0, %1" : "=v"(A0) : "v"(oa));                                          \
    asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(A1) : "v"(oa));                              \
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(A2) : "v"(oa));                              \
    asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(A3) : "v"(oa));                              \
    asm volatile("ds_read_b128 %0, %1" : "=v"(B0) : "v"(ob));                                          \
    asm volatile("ds_read_b128 %0, %1 offset:16384" : "=v"(B1) : "v"(ob));                             \
    asm volatile("ds_read_b128 %0, %1 offset:32768" : "=v"(B2) : "v"(ob));                             \
    asm volatile("ds_read_b128 %0, %1 offset:49152" : "=v"(B3) : "v"(ob));                             \
  }
#define AWAIT(A0, A1, A2, A3, B0, B1, B2, B3)                                                          \
  asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(B0), "+v"(B1), "+v"(B2), "+v"(B3));
#define SB __builtin_amdgcn_sched_barrier(0);
#define BURST(A0, A1, A2, A3, B0, B1, B2, B3)                                                          \
  {                                                                                                    \
    const f4 v0 = A0 - A2, v1 = A1 + A2, v2 = A2 - A1, v3 = A1 - A3; SB                                \
    m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(B0.x, v0.x, m0, 0, 0, 0);                                \
    m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(B1.x, v1.x, m1, 0, 0, 0);                                \
    m2 = __builtin_amdgcn_mfma_f32_32x32x2f32(B2.x, v2.x, m2, 0, 0, 0);                                \
    m3 = __builtin_amdgcn_mfma_f32_32x32x2f32(B3.x, v3.x, m3, 0, 0, 0);                                \
    m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(B0.y, v0.y, m0, 0, 0, 0);                                \
    m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(B1.y, v1.y, m1, 0, 0, 0);                                \
    m2 = __builtin_amdgcn_mfma_f32_32x32x2f32(B2.y, v2.y, m2, 0, 0, 0);                                \
    m3 = __builtin_amdgcn_mfma_f32_32x32x2f32(B3.y, v3.y, m3, 0, 0, 0);                                \
    m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(B0.z, v0.z, m0, 0, 0, 0);                                \
    m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(B1.z, v1.z, m1, 0, 0, 0);                                \
    m2 = __builtin_amdgcn_mfma_f32_32x32x2f32(B2.z, v2.z, m2, 0, 0, 0);                                \
    m3 = __builtin_amdgcn_mfma_f32_32x32x2f32(B3.z, v3.z, m3, 0, 0, 0);                                \
    m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(B0.w, v0.w, m0, 0, 0, 0);                                \
    m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(B1.w, v1.w, m1, 0, 0, 0);                                \
    m2 = __builtin_amdgcn_mfma_f32_32x32x2f32(B2.w, v2.w, m2, 0, 0, 0);                                \
    m3 = __builtin_amdgcn_mfma_f32_32x32x2f32(B3.w, v3.w, m3, 0, 0, 0); SB                             \
  }
  ALOAD(D0, D1, D2, D3, W0, W1, W2, W3, 0)
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int since = 0;
  for (s = 0; s < sets; s += 2) {
    ALOAD(E0, E1, E2, E3, X0, X1, X2, X3, 1) SB
    AWAIT(D0, D1, D2, D3, W0, W1, W2, W3) SB
    BURST(D0, D1, D2, D3, W0, W1, W2, W3)
    ALOAD(D0, D1, D2, D3, W0, W1, W2, W3, 2) SB
    AWAIT(E0, E1, E2, E3, X0, X1, X2, X3) SB
    BURST(E0, E1, E2, E3, X0, X1, X2, X3)
    since += 2;
    if (BAR && since >= BAR) {
      since = 0;
      __syncthreads();
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(D0), "+v"(W0));
  float r = D0.x + W0.x;
  for (int e = 0; e < 16; ++e) r += m0[e] + m1[e] + m2[e] + m3[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (lane == 0) { cyc[blockIdx.x * 16 + 2 * wave] = t0; cyc[blockIdx.x * 16 + 2 * wave + 1] = t1; }
}

template <int BAR>
void run(const char* name) {
  const int blocks = 256, sets = 432;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, blocks * 128);
  hipMemset(cyc, 0, blocks * 128);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<BAR>), dim3(blocks), dim3(512), 0, 0, out, cyc, sets, -1.0f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 16);
  hipMemcpy(h.data(), cyc, blocks * 128, hipMemcpyDeviceToHost);
  double sum = 0; int n = 0;
  for (int b = 0; b < blocks; ++b)
    for (int sd = 0; sd < 4; ++sd) {
      unsigned long long a = ~0ull, e = 0;
      for (int w = sd; w < 8; w += 4) { a = std::min(a, h[b * 16 + 2 * w]); e = std::max(e, h[b * 16 + 2 * w + 1]); }
      sum += (double)(e - a); ++n;
    }
  printf("%-70s %.1f cycles per MFMA (SIMD makespan)\n", name, sum / n / (sets * 16.0 * 2));
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run<0>("four taps together: 8 reads, 16 VALU, 16 MFMAs per set; no barrier");
  run<4>("  + workgroup barrier every 4 sets (one per 64 MFMAs)");
  run<2>("  + workgroup barrier every 2 sets");
  return 0;
}
