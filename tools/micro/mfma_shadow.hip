// The trunk's narrow stream (per operand set 3 ds_read_b128, 4 v_fma, 4 dependent MFMAs) with the work of the NEXT sets
// placed in the shadow of the current set's MFMAs: after MFMA 1 the reads of set s+2, after MFMA 2 and 3 the wait for and
// the v_fma of set s+1.  A wave then never stands between two bursts with nothing in the pipe.
// AGPR 1: the accumulators live in accumulation registers (a[..]) instead of v[..].
//   run<WAVES, BAR, SHADOW>: WAVES per SIMD 1 / 2; BAR: bursts between workgroup barriers (0 none); makespan of the SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NTH, int BAR, int SHADOW, int WIDE, int AGPR>
__global__ __launch_bounds__(NTH) void k(float* out, unsigned long long* cyc, int bursts, float sg) {
  __shared__ __attribute__((aligned(16))) float lds[36864];
  for (int i = threadIdx.x; i < 36864; i += blockDim.x) lds[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  f32x16 acc0, acc1;
  for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* pa = lds + ((wave * 64 + lane) * 4 & 16383);
  const float* pb = lds + 16384 + (lane * 4);
  const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)pa;
  const unsigned lb = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)pb;
  f4 Xa, Xs, Xb, Xc, Ya, Ys, Yb, Yc, Za, Zs, Zb, Zc, vx, vy, vz;
  int s = 0;
#define ALOAD(A_, S_, B_, C_, K)                                                                  \
  {                                                                                               \
    const unsigned oa = la + 4u * (((s + K) * 64) & 8191), ob = lb + 4u * (((s + K) * 256) & 8191); \
    asm volatile("ds_read_b128 %0, %1" : "=v"(A_) : "v"(oa));                                     \
    asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(S_) : "v"(oa));                         \
    asm volatile("ds_read_b128 %0, %1" : "=v"(B_) : "v"(ob));                                     \
    if (WIDE) asm volatile("ds_read_b128 %0, %1 offset:32768" : "=v"(C_) : "v"(ob));              \
  }
#define AWAIT(A_, S_, B_, C_)                                                                     \
  if (WIDE) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(A_), "+v"(S_), "+v"(B_), "+v"(C_));        \
  else asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(A_), "+v"(S_), "+v"(B_));
#define SB __builtin_amdgcn_sched_barrier(0);
#define MF(B_, C_, V_, E)                                                                         \
  if (AGPR) {                                                                                     \
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc0) : "v"(B_.E), "v"(V_.E));    \
    if (WIDE) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc1) : "v"(C_.E), "v"(V_.E)); \
  } else {                                                                                        \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(B_.E, V_.E, acc0, 0, 0, 0);                       \
    if (WIDE) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(C_.E, V_.E, acc1, 0, 0, 0);             \
  }
// burst of set (B_, V_); in its shadow: reads of the set after next into (NA_, NS_, NB_), v_fma of the next set (A1_, S1_) -> V1_
#define STEP(B_, C_, V_, NA_, NS_, NB_, NC_, K, A1_, S1_, B1_, C1_, V1_)                          \
  if (SHADOW) {                                                                                   \
    MF(B_, C_, V_, x) SB                                                                          \
    ALOAD(NA_, NS_, NB_, NC_, K) SB                                                               \
    MF(B_, C_, V_, y) SB                                                                          \
    AWAIT(A1_, S1_, B1_, C1_) SB                                                                  \
    V1_.x = fmaf(sg, S1_.x, A1_.x); V1_.y = fmaf(sg, S1_.y, A1_.y); SB                            \
    MF(B_, C_, V_, z) SB                                                                          \
    V1_.z = fmaf(sg, S1_.z, A1_.z); V1_.w = fmaf(sg, S1_.w, A1_.w); SB                            \
    MF(B_, C_, V_, w) SB                                                                          \
  } else {                                                                                        \
    ALOAD(NA_, NS_, NB_, NC_, K) SB                                                               \
    AWAIT(A1_, S1_, B1_, C1_) SB                                                                  \
    V1_.x = fmaf(sg, S1_.x, A1_.x); V1_.y = fmaf(sg, S1_.y, A1_.y);                               \
    V1_.z = fmaf(sg, S1_.z, A1_.z); V1_.w = fmaf(sg, S1_.w, A1_.w); SB                            \
    MF(B_, C_, V_, x) MF(B_, C_, V_, y) MF(B_, C_, V_, z) MF(B_, C_, V_, w) SB                    \
  }
  ALOAD(Xa, Xs, Xb, Xc, 0)
  ALOAD(Ya, Ys, Yb, Yc, 1)
  if (!WIDE) { Xc = Xb; Yc = Yb; Zc = Xb; }
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Xa), "+v"(Xs), "+v"(Xb), "+v"(Xc));
  vx.x = fmaf(sg, Xs.x, Xa.x); vx.y = fmaf(sg, Xs.y, Xa.y); vx.z = fmaf(sg, Xs.z, Xa.z); vx.w = fmaf(sg, Xs.w, Xa.w);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int since = 0;
  for (int b = 0; b < bursts; b += 3, s += 3) {
    // set s in X (vx ready), set s+1 in Y (loaded), set s+2 -> Z
    STEP(Xb, Xc, vx, Za, Zs, Zb, Zc, 2, Ya, Ys, Yb, Yc, vy)
    STEP(Yb, Yc, vy, Xa, Xs, Xb, Xc, 3, Za, Zs, Zb, Zc, vz)
    STEP(Zb, Zc, vz, Ya, Ys, Yb, Yc, 4, Xa, Xs, Xb, Xc, vx)
    since += 3;
    if (BAR && since >= BAR) {
      since = 0;
      __syncthreads();
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Xa), "+v"(Ya));
  float r = Xa.x + Ya.x + vx.x;
  for (int e = 0; e < 16; ++e) r += acc0[e] + acc1[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (lane == 0) { cyc[blockIdx.x * 16 + 2 * wave] = t0; cyc[blockIdx.x * 16 + 2 * wave + 1] = t1; }
}

template <int NTH, int BAR, int SHADOW, int WIDE, int AGPR>
void run(const char* name) {
  const int blocks = 256, bursts = 1728;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, blocks * 128);
  hipMemset(cyc, 0, blocks * 128);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<NTH, BAR, SHADOW, WIDE, AGPR>), dim3(blocks), dim3(NTH), 0, 0, out, cyc, bursts, -1.0f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 16);
  hipMemcpy(h.data(), cyc, blocks * 128, hipMemcpyDeviceToHost);
  // makespan per SIMD: from the first start to the last end of the waves that share it (waves w and w + 4)
  double sum = 0; int n = 0;
  const int nw = NTH / 64;
  for (int b = 0; b < blocks; ++b)
    for (int sd = 0; sd < 4; ++sd) {
      unsigned long long a = ~0ull, e = 0;
      for (int w = sd; w < nw; w += 4) { a = std::min(a, h[b * 16 + 2 * w]); e = std::max(e, h[b * 16 + 2 * w + 1]); }
      sum += (double)(e - a); ++n;
    }
  const double mf = (double)bursts * (WIDE ? 8 : 4) * (nw / 4);
  printf("%-78s %.1f cycles per MFMA (SIMD makespan)\n", name, sum / n / mf);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<256, 0, 0, 0, 0>("narrow, 1 wave/SIMD, acc in VGPRs");
  run<256, 0, 0, 0, 1>("narrow, 1 wave/SIMD, acc in AGPRs");
  run<256, 0, 1, 0, 1>("narrow, 1 wave/SIMD, acc in AGPRs, shadows");
  run<512, 0, 0, 0, 0>("narrow, 2 waves/SIMD, acc in VGPRs");
  run<512, 0, 0, 0, 1>("narrow, 2 waves/SIMD, acc in AGPRs");
  run<512, 12, 0, 0, 1>("narrow, 2 waves/SIMD, acc in AGPRs, barrier every 12 bursts");
  run<512, 12, 1, 0, 1>("narrow, 2 waves/SIMD, acc in AGPRs, shadows, barrier every 12 bursts");
  run<512, 0, 0, 1, 0>("wide, 2 waves/SIMD, acc in VGPRs");
  run<512, 0, 0, 1, 1>("wide, 2 waves/SIMD, acc in AGPRs");
  run<512, 6, 0, 1, 1>("wide, 2 waves/SIMD, acc in AGPRs, barrier every 6 bursts");
  return 0;
}
