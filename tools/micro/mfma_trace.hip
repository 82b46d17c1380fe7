// Timeline of the two waves of a SIMD between two workgroup barriers: s_memtime at the start of every burst of four
// MFMAs (narrow stream, hand-written reads two sets ahead, counted waits), waves 0 and 4 of one workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int NB = 48;  // bursts per interval

template <int BAR>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* trace, int intervals, float sg) {
  __shared__ __attribute__((aligned(16))) float lds[36864];
  __shared__ unsigned long long ts[8][NB + 2];
  for (int i = threadIdx.x; i < 36864; i += blockDim.x) lds[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  f32x16 acc0;
  for (int e = 0; e < 16; ++e) acc0[e] = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* pa = lds + ((wave * 64 + lane) * 4 & 16383);
  const float* pb = lds + 16384 + (lane * 4);
  const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)pa;
  const unsigned lb = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)pb;
  f4 Xa, Xs, Xb, Ya, Ys, Yb, Za, Zs, Zb;
  int s = 0;
#define ALOAD(A_, S_, B_, K)                                                                      \
  {                                                                                               \
    const unsigned oa = la + 4u * (((s + K) * 64) & 8191), ob = lb + 4u * (((s + K) * 256) & 8191); \
    asm volatile("ds_read_b128 %0, %1" : "=v"(A_) : "v"(oa));                                     \
    asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(S_) : "v"(oa));                         \
    asm volatile("ds_read_b128 %0, %1" : "=v"(B_) : "v"(ob));                                     \
  }
#define AWAIT(A_, S_, B_) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(A_), "+v"(S_), "+v"(B_));
#define BURST(A_, S_, B_, I)                                                                      \
  {                                                                                               \
    f4 v;                                                                                         \
    v.x = fmaf(sg, S_.x, A_.x); v.y = fmaf(sg, S_.y, A_.y); v.z = fmaf(sg, S_.z, A_.z); v.w = fmaf(sg, S_.w, A_.w); \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    if (last) stamp[(I)] = __builtin_amdgcn_s_memtime();                                          \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(B_.x, v.x, acc0, 0, 0, 0);                        \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(B_.y, v.y, acc0, 0, 0, 0);                        \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(B_.z, v.z, acc0, 0, 0, 0);                        \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(B_.w, v.w, acc0, 0, 0, 0);                        \
  }
  ALOAD(Xa, Xs, Xb, 0)
  ALOAD(Ya, Ys, Yb, 1)
  __syncthreads();
  unsigned long long* stamp = ts[wave];
  for (int it = 0; it < intervals; ++it) {
    const bool last = it == intervals - 1 && lane == 0;
    for (int b = 0; b < NB; b += 3, s += 3) {
      ALOAD(Za, Zs, Zb, 2)
      AWAIT(Xa, Xs, Xb)
      __builtin_amdgcn_sched_barrier(0);
      BURST(Xa, Xs, Xb, b)
      __builtin_amdgcn_sched_barrier(0);
      ALOAD(Xa, Xs, Xb, 3)
      AWAIT(Ya, Ys, Yb)
      __builtin_amdgcn_sched_barrier(0);
      BURST(Ya, Ys, Yb, b + 1)
      __builtin_amdgcn_sched_barrier(0);
      ALOAD(Ya, Ys, Yb, 4)
      AWAIT(Za, Zs, Zb)
      __builtin_amdgcn_sched_barrier(0);
      BURST(Za, Zs, Zb, b + 2)
      __builtin_amdgcn_sched_barrier(0);
    }
    if (last) stamp[NB] = __builtin_amdgcn_s_memtime();
    if (BAR) __syncthreads();
    if (last) stamp[NB + 1] = __builtin_amdgcn_s_memtime();
  }
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Xa), "+v"(Ya));
  float r = Xa.x + Ya.x;
  for (int e = 0; e < 16; ++e) r += acc0[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  __syncthreads();
  if (blockIdx.x == 7)
    for (int i = threadIdx.x; i < 8 * (NB + 2); i += blockDim.x) trace[i] = ts[i / (NB + 2)][i % (NB + 2)];
}

template <int BAR>
void run(const char* name) {
  const int blocks = 256;
  float* out; unsigned long long* tr;
  hipMalloc(&out, blocks * 512 * 4); hipMalloc(&tr, 8 * (NB + 2) * 8);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<BAR>), dim3(blocks), dim3(512), 0, 0, out, tr, 40, -1.0f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(8 * (NB + 2));
  hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost);
  printf("%s\n", name);
  const unsigned long long t0 = std::min(h[0], h[4 * (NB + 2)]);
  for (int w : {0, 4}) {
    printf(" wave %d burst starts (cycles after the interval's first):", w);
    for (int b = 0; b <= NB + 1; ++b) printf(" %llu", h[w * (NB + 2) + b] - t0);
    printf("\n");
  }
  hipFree(out); hipFree(tr);
}
int main() {
  run<1>("barrier every 48 bursts (last two numbers: arrival at the barrier, release)");
  run<0>("no barrier");
  return 0;
}
