// The trunk's stream in its present form (wave = 32 tiles x 32 channels, all of K: per operand set 3 ds_read_b128,
// 4 v_fma, 4 MFMAs on one accumulator) against a K-split form with the same 8 waves (wave = 32 tiles x 64 channels,
// half of K: per set 4 ds_read_b128, 4 v_fma, 8 MFMAs on two accumulators), at EQUAL barrier spacing in MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ORD 3..5: as 0, the older wave of a SIMD (waves 0-3) sleeps 1..3 x 64 cycles after each burst.  ORD 6: as 0, the two waves of a
// SIMD take turns burst by burst (progress counters in LDS, bounded polling).
// ORD 7: reads of set s+2 ahead of the burst of set s (three operand sets in flight).
// ORD 8: as 7 with the reads and their waits written by hand (ds_read_b128 + s_waitcnt lgkmcnt(2 sets)): hipcc waits for
// lgkmcnt(0), i.e. also for the reads it has just issued.
// ORD 0: reads of set s+1, then the v_fma of set s, then its MFMAs (the trunk as it is); ORD 1: v_fma of set s first (its operands
// were requested a whole burst ago), then the reads of set s+1, then the MFMAs: the wait in front of the v_fma no longer covers
// the reads just issued.  ORD 2: as 0 with s_setprio 3 from the first MFMA of a burst to its last.
template <int WIDE, int BAR, int ORD>  // BAR: MFMAs per wave between workgroup barriers (0: none)
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* cyc, int mfmas, float sg) {
  __shared__ __attribute__((aligned(16))) float lds[40000];
  __shared__ int prog[8];
  if (threadIdx.x < 8) prog[threadIdx.x] = 0;
  for (int i = threadIdx.x; i < 40000; i += blockDim.x) lds[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  f32x16 acc0, acc1;
  for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* pa = lds + ((wave * 64 + lane) * 4 & 16383);
  const float* pb = lds + 16384 + (lane * 4);
  constexpr int PER = WIDE ? 8 : 4;
  const int sets = mfmas / PER;
  float4 xa, xs, xb0, xb1, ya, ys, yb0, yb1;
#define LOAD(A_, S_, B0_, B1_, K)                                                          \
  {                                                                                        \
    A_ = *reinterpret_cast<const float4*>(pa + (((s + K) * 64) & 8191));                   \
    S_ = *reinterpret_cast<const float4*>(pa + 256 + (((s + K) * 64) & 8191));             \
    B0_ = *reinterpret_cast<const float4*>(pb + (((s + K) * 256) & 8191));                 \
    if (WIDE) B1_ = *reinterpret_cast<const float4*>(pb + 8192 + (((s + K) * 256) & 8191)); \
  }
#define TURN(I)                                                                                  \
  if (ORD == 6) {                                                                                \
    const int need = (I) + (wave >= 4 ? 1 : 0);                                                  \
    for (int tries = 0; tries < 32; ++tries)                                                     \
      if (__hip_atomic_load(&prog[wave ^ 4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= need) break;                                        \
    __builtin_amdgcn_sched_barrier(0);                                                           \
  }
#define FM(A_, S_, V_)                                                                                       \
  V_.x = fmaf(sg, S_.x, A_.x); V_.y = fmaf(sg, S_.y, A_.y); V_.z = fmaf(sg, S_.z, A_.z); V_.w = fmaf(sg, S_.w, A_.w);
#define MM(V_, B0_, B1_)                                                                                     \
  {                                                                                                          \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(B0_.x, V_.x, acc0, 0, 0, 0);                                 \
    if (ORD == 2) __builtin_amdgcn_s_setprio(3);                                                             \
    if (WIDE) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(B1_.x, V_.x, acc1, 0, 0, 0);                       \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(B0_.y, V_.y, acc0, 0, 0, 0);                                 \
    if (WIDE) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(B1_.y, V_.y, acc1, 0, 0, 0);                       \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(B0_.z, V_.z, acc0, 0, 0, 0);                                 \
    if (WIDE) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(B1_.z, V_.z, acc1, 0, 0, 0);                       \
    if (ORD == 6) { ++mine; if (lane == 0) __hip_atomic_store(&prog[wave], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); __builtin_amdgcn_sched_barrier(0); } \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(B0_.w, V_.w, acc0, 0, 0, 0);                                 \
    if (WIDE) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(B1_.w, V_.w, acc1, 0, 0, 0);                       \
    if (ORD == 2) __builtin_amdgcn_s_setprio(0);                                                             \
    if (ORD >= 3 && ORD <= 5 && wave < 4) __builtin_amdgcn_s_sleep(ORD - 2);                                 \
  }
  int s = 0;
  LOAD(xa, xs, xb0, xb1, 0)
  yb1 = xb1 = xb0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int since = 0;
  unsigned long long waited = 0;
  int mine = 0;
  if (ORD == 8) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 Xa, Xs, Xb0, Xb1, Ya, Ys, Yb0, Yb1, Za, Zs, Zb0, Zb1;
    const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)pa;
    const unsigned lb = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)pb;
#define ALOAD(A_, S_, B0_, B1_, K)                                                                             \
  {                                                                                                            \
    const unsigned oa = la + 4u * (((s + K) * 64) & 8191), ob = lb + 4u * (((s + K) * 256) & 8191);            \
    asm volatile("ds_read_b128 %0, %1" : "=v"(A_) : "v"(oa));                                                  \
    asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(S_) : "v"(oa));                                      \
    asm volatile("ds_read_b128 %0, %1" : "=v"(B0_) : "v"(ob));                                                 \
    if (WIDE) asm volatile("ds_read_b128 %0, %1 offset:32768" : "=v"(B1_) : "v"(ob));                          \
  }
#define AWAIT(A_, S_, B0_, B1_)                                                                                \
  if (WIDE) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(A_), "+v"(S_), "+v"(B0_), "+v"(B1_));                   \
  else asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(A_), "+v"(S_), "+v"(B0_));
    s = 0;
    ALOAD(Xa, Xs, Xb0, Xb1, 0)
    ALOAD(Ya, Ys, Yb0, Yb1, 1)
    if (!WIDE) { Xb1 = Xb0; Yb1 = Yb0; Zb1 = Xb0; }
    for (s = 0; s < sets; s += 3) {
      f4 v;
      ALOAD(Za, Zs, Zb0, Zb1, 2)
      AWAIT(Xa, Xs, Xb0, Xb1)
      __builtin_amdgcn_sched_barrier(0);
      FM(Xa, Xs, v)
      __builtin_amdgcn_sched_barrier(0);
      MM(v, Xb0, Xb1)
      __builtin_amdgcn_sched_barrier(0);
      ALOAD(Xa, Xs, Xb0, Xb1, 3)
      AWAIT(Ya, Ys, Yb0, Yb1)
      __builtin_amdgcn_sched_barrier(0);
      FM(Ya, Ys, v)
      __builtin_amdgcn_sched_barrier(0);
      MM(v, Yb0, Yb1)
      __builtin_amdgcn_sched_barrier(0);
      ALOAD(Ya, Ys, Yb0, Yb1, 4)
      AWAIT(Za, Zs, Zb0, Zb1)
      __builtin_amdgcn_sched_barrier(0);
      FM(Za, Zs, v)
      __builtin_amdgcn_sched_barrier(0);
      MM(v, Zb0, Zb1)
      __builtin_amdgcn_sched_barrier(0);
      since += 3 * PER;
      if (BAR && since >= BAR) {
        since = 0;
        const unsigned long long b0 = __builtin_amdgcn_s_memtime();
        __syncthreads();
        waited += __builtin_amdgcn_s_memtime() - b0;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Xa), "+v"(Ya));
    acc0[0] += Xa.x + Ya.x;
  } else
  if (ORD == 7) {
    float4 za, zs, zb0, zb1;
    LOAD(ya, ys, yb0, yb1, 1)
    zb1 = yb1;
    for (s = 0; s < sets; s += 3) {
      float4 v;
      LOAD(za, zs, zb0, zb1, 2)
      __builtin_amdgcn_sched_barrier(0);
      FM(xa, xs, v)
      __builtin_amdgcn_sched_barrier(0);
      MM(v, xb0, xb1)
      __builtin_amdgcn_sched_barrier(0);
      LOAD(xa, xs, xb0, xb1, 3)
      __builtin_amdgcn_sched_barrier(0);
      FM(ya, ys, v)
      __builtin_amdgcn_sched_barrier(0);
      MM(v, yb0, yb1)
      __builtin_amdgcn_sched_barrier(0);
      LOAD(ya, ys, yb0, yb1, 4)
      __builtin_amdgcn_sched_barrier(0);
      FM(za, zs, v)
      __builtin_amdgcn_sched_barrier(0);
      MM(v, zb0, zb1)
      __builtin_amdgcn_sched_barrier(0);
      since += 3 * PER;
      if (BAR && since >= BAR) {
        since = 0;
        const unsigned long long b0 = __builtin_amdgcn_s_memtime();
        __syncthreads();
        waited += __builtin_amdgcn_s_memtime() - b0;
      }
    }
  } else
  for (s = 0; s < sets; s += 2) {
    float4 v;
    if (ORD == 1) {
      FM(xa, xs, v)
      __builtin_amdgcn_sched_barrier(0);
      LOAD(ya, ys, yb0, yb1, 1)
    } else {
      LOAD(ya, ys, yb0, yb1, 1)
      __builtin_amdgcn_sched_barrier(0);
      FM(xa, xs, v)
    }
    __builtin_amdgcn_sched_barrier(0);
    TURN(s)
    MM(v, xb0, xb1)
    __builtin_amdgcn_sched_barrier(0);
    if (ORD == 1) {
      FM(ya, ys, v)
      __builtin_amdgcn_sched_barrier(0);
      LOAD(xa, xs, xb0, xb1, 2)
    } else {
      LOAD(xa, xs, xb0, xb1, 2)
      __builtin_amdgcn_sched_barrier(0);
      FM(ya, ys, v)
    }
    __builtin_amdgcn_sched_barrier(0);
    TURN(s + 1)
    MM(v, yb0, yb1)
    __builtin_amdgcn_sched_barrier(0);
    since += 2 * PER;
    if (BAR && since >= BAR) {
      since = 0;
      const unsigned long long b0 = __builtin_amdgcn_s_memtime();
      __syncthreads();
      waited += __builtin_amdgcn_s_memtime() - b0;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
  for (int e = 0; e < 16; ++e) r += acc0[e] + acc1[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (lane == 0) { cyc[blockIdx.x * 8 + wave] = t1 - t0; cyc[2048 + blockIdx.x * 8 + wave] = waited; }
}

template <int WIDE, int BAR, int ORD>
void run(const char* name) {
  const int blocks = 256, mfmas = 6912;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, 2 * blocks * 64);
  hipMemset(cyc, 0, 2 * blocks * 64);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<WIDE, BAR, ORD>), dim3(blocks), dim3(512), 0, 0, out, cyc, mfmas, -1.0f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(2 * blocks * 8);
  hipMemcpy(h.data(), cyc, 2 * blocks * 64, hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < blocks * 8; ++i) sum += h[i];
  printf("%-72s %.1f cycles per MFMA on the SIMD\n", name, sum / (blocks * 8) / mfmas / 2.0);
  if (BAR) {
    printf("    share of the time spent waiting at barriers, by wave:");
    for (int w = 0; w < 8; ++w) {
      double a = 0, t = 0;
      for (int b = 0; b < blocks; ++b) { a += h[2048 + b * 8 + w]; t += h[b * 8 + w]; }
      printf(" %.3f", a / t);
    }
    printf("\n");
  }
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0, 0, 0>("narrow (3 reads, 4 fma, 4 MFMA per set), no barrier");
  run<0, 48, 0>("narrow, barrier every 48 MFMAs (the trunk's average: 2 per 96)");
  run<0, 48, 7>("narrow, reads two sets ahead, barrier every 48");
  run<0, 0, 8>("narrow, reads two sets ahead + counted waits, no barrier");
  run<0, 48, 8>("narrow, reads two sets ahead + counted waits, barrier every 48");
  run<1, 0, 0>("wide (4 reads, 4 fma, 8 MFMA per set), no barrier");
  run<1, 48, 0>("wide, barrier every 48 MFMAs");
  run<1, 48, 7>("wide, reads two sets ahead, barrier every 48");
  run<1, 0, 8>("wide, reads two sets ahead + counted waits, no barrier");
  run<1, 48, 8>("wide, reads two sets ahead + counted waits, barrier every 48");
  return 0;
}
