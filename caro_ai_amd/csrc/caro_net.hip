// caro_net.hip -- fused float32 inference of the policy/value net (reference
// lib/model.py:10-94, eval-mode batch-norm folded) for the leaf batch of the
// self-play engine, as ONE kernel per minibatch.
//
// Why a kernel of our own: the leaf count L changes every minibatch and lives in
// device memory; a torch forward needs L on the host (a stream sync per
// minibatch) and ~45 launches; MIOpen has no gfx950 database in this image.
// Here the grid is sized for the maximum and every workgroup reads L itself.
//
// Structure (one workgroup = 512 threads = 8 waves = one CU, two waves per SIMD so that one wave's
// MFMAs cover the other's LDS operand reads; TB boards):
//   rows r = board*HW + cell, at most 255 real rows; row 255 is a permanent zero
//   row (3x3 padding).  Activations X[row][64] float32 stay in LDS for the whole
//   trunk in ONE 64 KiB buffer (XOR-swizzled 16-byte granules) that is updated in
//   place: a layer's outputs live in the MFMA accumulators until every wave has
//   finished reading the layer's input, then overwrite it.  The
//   3x3 convolutions are implicit GEMMs on v_mfma_f32_32x32x2_f32:
//       M = 256 rows (8 row tiles), N = 64 (2 col tiles), K = 9 taps x 64 channels,
//   wave w owns row tile w x both col tiles (2 accumulators of 16 regs).
//   K order inside a tap: MFMA k-half h = lane>>5 carries channel 32h + j, so a
//   lane's A operands for 4 consecutive k-steps are one ds_read_b128.
//   Weights stream from L2 in chunks of three taps (3 x 64x64 floats = 48 KiB)
//   through a double-buffered LDS stage: global loads for chunk c+1 are issued
//   before the MFMAs of chunk c and written to the other buffer after them
//   (issue-early / write-late), so there is one workgroup barrier per chunk
//   plus two per layer around the in-place epilogue.
//   conv_in (K = 18), the 1x1 heads, the two FC heads, tanh and the softmax
//   run on the VALU in the same kernel.
// float32 throughout: MFMA f32 is an exact fma chain in k order.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/caro_hip.h"

namespace cnet {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NF = 64;          // filters
constexpr int ZROW = 255;       // permanent zero row
constexpr int ACT = 256 * NF;   // floats per activation buffer
constexpr int WCHUNK = 64 * 64; // floats per tap chunk
constexpr int NRES = 5;
constexpr int TPC = 3;                          // taps per weight chunk (3 chunks per layer)
constexpr int NTAPS = NRES * 9;                 // 45
constexpr int NCHUNK = (NTAPS + TPC - 1) / TPC; // 15
constexpr int LDS_FLOATS = ACT + 2 * TPC * WCHUNK;

struct NetParams {
  int H, W, HW, A, TB;
  float slope;
  const float* w_in;    // [9][2][64]
  const float* b_in;    // [64]
  const float* w_res;   // [5][9][4096]  LDS image order (see pack_res_index)
  const float* b_res;   // [5][64]
  const float* w_head;  // [3][64]  (value, policy0, policy1)
  const float* b_head;  // [3]
  const float* w_v1;    // [20][HW]
  const float* b_v1;    // [20]
  const float* w_v2;    // [20]
  const float* b_v2;    // [1]
  const float* w_p;     // [A][2*HW]
  const float* b_p;     // [A]
  const uint4* w3;      // 3xbf16 mode: [45 taps][1536 granules] LDS image of the split residual weights, or null
};

__device__ __forceinline__ int aoff(int row, int c) {
  return row * NF + ((((c >> 2) ^ (row & 15)) << 2) | (c & 3));
}
__device__ __forceinline__ float leaky(float x, float slope) { return x > 0.f ? x : x * slope; }

constexpr int NT = 512;  // threads per workgroup

// `which` = 0 / 1: rows of that net only (p0 is used).  `which` = 2: both nets in ONE launch -- tiles
// [0, ceil(L0/TB)) run net 0 on rows [0, L0), the following tiles run net 1 (p1) on rows [L0, L0+L1).
__global__ __launch_bounds__(NT, 2) void k_net_forward(NetParams p0, NetParams p1, const float* __restrict__ planes,
                                                         const int32_t* __restrict__ counts, int which,
                                                         float* __restrict__ probs, float* __restrict__ values,
                                                         unsigned long long* __restrict__ stamps) {
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  float* act = lds;
  float* wbuf = lds + ACT;

  int L, row0, board0;
  bool second = false;
  if (which < 2) {
    L = counts[which];
    row0 = which ? counts[0] : 0;
    board0 = blockIdx.x * p0.TB;
  } else {
    const int L0 = counts[0];
    const int t0 = (L0 + p0.TB - 1) / p0.TB;
    second = (int)blockIdx.x >= t0;
    L = second ? counts[1] : L0;
    row0 = second ? L0 : 0;
    board0 = (second ? (int)blockIdx.x - t0 : (int)blockIdx.x) * p0.TB;
  }
  if (board0 >= L) return;
  const NetParams& p = second ? p1 : p0;
  const float slope = p.slope;
  // diagnostic only (stamps == nullptr in every product launch): shader clock vs 100 MHz wall clock
  unsigned long long t_c0 = 0, t_r0 = 0;
  if (stamps) {
    t_c0 = __builtin_amdgcn_s_memtime();
    t_r0 = __builtin_amdgcn_s_memrealtime();
  }
  const int nb = min(p.TB, L - board0);
  const int HW = p.HW;
  const int R = nb * HW;  // real rows
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int i = lane & 31, h = lane >> 5;

  // zero the activation buffer (dummy rows and the zero row stay zero for ever)
  for (int k = tid; k < ACT / 4; k += NT) reinterpret_cast<float4*>(lds)[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  // conv_in weights into wbuf: [9][2][64] = 1152 floats
  for (int k = tid; k < 9 * 2 * NF; k += NT) wbuf[k] = p.w_in[k];
  __syncthreads();

  // ---- conv_in on the VALU: two threads per row, 32 output channels each
  {
    const int r = tid & 255;
    const int chalf = tid >> 8;
    if (r < R) {
      const int bi = r / HW, cell = r - bi * HW;
      const int y = cell / p.W, x = cell - y * p.W;
      const float* pl = planes + (size_t)(row0 + board0 + bi) * 2 * HW;
      float in0[9], in1[9];  // the 18 inputs of this row (statically indexed: stays in registers)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int ny = y + t / 3 - 1, nx = x + t % 3 - 1;
        const bool ok = ny >= 0 && ny < p.H && nx >= 0 && nx < p.W;
        in0[t] = ok ? pl[ny * p.W + nx] : 0.f;
        in1[t] = ok ? pl[HW + ny * p.W + nx] : 0.f;
      }
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) {
        const int c4 = chalf * 8 + cc;
        float o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) o[u] = p.b_in[c4 * 4 + u];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const float i0 = in0[t], i1 = in1[t];
          const float4 w0 = *reinterpret_cast<const float4*>(wbuf + (2 * t) * NF + c4 * 4);
          const float4 w1 = *reinterpret_cast<const float4*>(wbuf + (2 * t + 1) * NF + c4 * 4);
          o[0] = fmaf(i0, w0.x, o[0]); o[1] = fmaf(i0, w0.y, o[1]); o[2] = fmaf(i0, w0.z, o[2]); o[3] = fmaf(i0, w0.w, o[3]);
          o[0] = fmaf(i1, w1.x, o[0]); o[1] = fmaf(i1, w1.y, o[1]); o[2] = fmaf(i1, w1.z, o[2]); o[3] = fmaf(i1, w1.w, o[3]);
        }
        float4 out = make_float4(leaky(o[0], slope), leaky(o[1], slope), leaky(o[2], slope),
                                 leaky(o[3], slope));
        *reinterpret_cast<float4*>(act + r * NF + ((c4 ^ (r & 15)) << 2)) = out;
      }
    }
  }
  __syncthreads();

  unsigned long long t_trunk0 = 0;
  if (stamps) t_trunk0 = __builtin_amdgcn_s_memtime();
  // ---- stage weight chunk 0 (taps 0 and 1 of layer 0)
  {
    const float4* src = reinterpret_cast<const float4*>(p.w_res);
#pragma unroll
    for (int m = 0; m < 2 * TPC; ++m) reinterpret_cast<float4*>(wbuf)[tid + NT * m] = src[tid + NT * m];
  }
  __syncthreads();

  // per-lane geometry of its row tile
  const int myrow = wave * 32 + i;
  const bool rvalid = myrow < R;
  const int rbi = myrow / HW;
  const int rcell = myrow - rbi * HW;
  const int ry = rcell / p.W, rx = rcell - ry * p.W;
  const int bswz = (i >> 1) & 7;

  f32x16 acc0, acc1;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    acc0[e] = 0.f;
    acc1[e] = 0.f;
  }
  static_assert(TPC == 3, "the staging registers below are written out for three taps per chunk");
  float4 wn0, wn1, wn2, wn3, wn4, wn5;  // next weight chunk in flight (lives across the taps of a chunk)
  wn0 = wn1 = wn2 = wn3 = wn4 = wn5 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int ft = 0; ft < NTAPS; ++ft) {  // flat tap index over the five residual layers
    const int c = ft / TPC, within = ft % TPC, cur = c & 1;
    const int layer = ft / 9, tap = ft % 9;
    const bool last_in_chunk = within == TPC - 1 || ft == NTAPS - 1;
    const bool has_next = c + 1 < NCHUNK;
    if (within == 0 && has_next) {  // issue early
      // the last chunk holds one tap only; its second half reads the zero padding behind the packed weights
      const float4* src = reinterpret_cast<const float4*>(p.w_res + (size_t)(c + 1) * TPC * WCHUNK);
      wn0 = src[tid];
      wn1 = src[tid + NT];
      wn2 = src[tid + 2 * NT];
      wn3 = src[tid + 3 * NT];
      wn4 = src[tid + 4 * NT];
      wn5 = src[tid + 5 * NT];
    }
    const float* wcur = wbuf + cur * TPC * WCHUNK + within * WCHUNK;
    const int ny = ry + tap / 3 - 1, nx = rx + tap % 3 - 1;
    const bool ok = rvalid && ny >= 0 && ny < p.H && nx >= 0 && nx < p.W;
    const int nrow = ok ? rbi * HW + ny * p.W + nx : ZROW;
    const float* abase = act + nrow * NF;
    const int aswz = nrow & 15;
    const float* bbase0 = wcur + (h * 64 + i) * 32;
    const float* bbase1 = wcur + (h * 64 + 32 + i) * 32;
    // software pipeline with two explicit operand register sets: the reads of group q+1 are ISSUED before the
    // eight MFMAs of group q (sched_barrier keeps hipcc from sinking them next to their consumers, which it
    // otherwise does to save registers and which exposes one LDS latency per group)
#define CARO_LOAD_SET(A_, B0_, B1_, Q_)                                                             \
  A_ = *reinterpret_cast<const float4*>(abase + (((h * 8 + (Q_)) ^ aswz) << 2));                    \
  B0_ = *reinterpret_cast<const float4*>(bbase0 + (((Q_) ^ bswz) << 2));                            \
  B1_ = *reinterpret_cast<const float4*>(bbase1 + (((Q_) ^ bswz) << 2));
#define CARO_MFMA_SET(A_, B0_, B1_)                                                \
  acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.x, B0_.x, acc0, 0, 0, 0);        \
  acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.x, B1_.x, acc1, 0, 0, 0);        \
  acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.y, B0_.y, acc0, 0, 0, 0);        \
  acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.y, B1_.y, acc1, 0, 0, 0);        \
  acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.z, B0_.z, acc0, 0, 0, 0);        \
  acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.z, B1_.z, acc1, 0, 0, 0);        \
  acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.w, B0_.w, acc0, 0, 0, 0);        \
  acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.w, B1_.w, acc1, 0, 0, 0);
    float4 xa, xb0, xb1, ya, yb0, yb1;
    CARO_LOAD_SET(xa, xb0, xb1, 0)
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
      CARO_LOAD_SET(ya, yb0, yb1, q + 1)
      __builtin_amdgcn_sched_barrier(0);
      CARO_MFMA_SET(xa, xb0, xb1)
      __builtin_amdgcn_sched_barrier(0);
      if (q + 2 < 8) {
        CARO_LOAD_SET(xa, xb0, xb1, q + 2)
      }
      __builtin_amdgcn_sched_barrier(0);
      CARO_MFMA_SET(ya, yb0, yb1)
      __builtin_amdgcn_sched_barrier(0);
    }
#undef CARO_LOAD_SET
#undef CARO_MFMA_SET
    if (last_in_chunk && has_next) {  // write late: the other buffer was last read one chunk ago
      float4* dst = reinterpret_cast<float4*>(wbuf + (cur ^ 1) * TPC * WCHUNK);
      dst[tid] = wn0;
      dst[tid + NT] = wn1;
      dst[tid + 2 * NT] = wn2;
      dst[tid + 3 * NT] = wn3;
      dst[tid + 4 * NT] = wn4;
      dst[tid + 5 * NT] = wn5;
    }
    if (tap == 8) {
      __syncthreads();  // every wave has read this layer's input activations: they may be overwritten
      // epilogue, in place: v = v + leaky(conv(v) + b)   (lib/model.py:85-89).  Branch-free: rows >= R (dummy
      // rows and the zero row) are rewritten with zeros; all 32 residual reads are issued before the first use.
      const float* bias = p.b_res + layer * NF;
      const float bc0 = bias[i], bc1 = bias[32 + i];
      float old0[16], old1[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        old0[e] = act[aoff(row, i)];
        old1[e] = act[aoff(row, 32 + i)];
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const bool real = row < R;
        const float n0 = old0[e] + leaky(acc0[e] + bc0, slope);
        const float n1 = old1[e] + leaky(acc1[e] + bc1, slope);
        act[aoff(row, i)] = real ? n0 : 0.f;
        act[aoff(row, 32 + i)] = real ? n1 : 0.f;
        acc0[e] = 0.f;
        acc1[e] = 0.f;
      }
    }
    if (last_in_chunk || tap == 8) __syncthreads();  // staged weights / new activations visible to every wave
  }
  unsigned long long t_trunk1 = 0;
  if (stamps) t_trunk1 = __builtin_amdgcn_s_memtime();
  // `act` now holds the trunk output; the weight stage is free scratch
  float* feat = wbuf;  // [3][256]: value plane, policy plane 0, policy plane 1 (row indexed)
  {
    const int r = tid;
    if (r < R) {
      float s0 = p.b_head[0], s1 = p.b_head[1], s2 = p.b_head[2];
      for (int g = 0; g < 16; ++g) {
        const float4 v = *reinterpret_cast<const float4*>(act + r * NF + ((g ^ (r & 15)) << 2));
        const int c = g * 4;
        s0 = fmaf(v.x, p.w_head[c], s0); s0 = fmaf(v.y, p.w_head[c + 1], s0);
        s0 = fmaf(v.z, p.w_head[c + 2], s0); s0 = fmaf(v.w, p.w_head[c + 3], s0);
        s1 = fmaf(v.x, p.w_head[NF + c], s1); s1 = fmaf(v.y, p.w_head[NF + c + 1], s1);
        s1 = fmaf(v.z, p.w_head[NF + c + 2], s1); s1 = fmaf(v.w, p.w_head[NF + c + 3], s1);
        s2 = fmaf(v.x, p.w_head[2 * NF + c], s2); s2 = fmaf(v.y, p.w_head[2 * NF + c + 1], s2);
        s2 = fmaf(v.z, p.w_head[2 * NF + c + 2], s2); s2 = fmaf(v.w, p.w_head[2 * NF + c + 3], s2);
      }
      feat[r] = leaky(s0, slope);
      feat[256 + r] = leaky(s1, slope);
      feat[512 + r] = leaky(s2, slope);
    }
  }
  __syncthreads();
  float* hid = feat + 768;      // [TB][20]
  float* logit = feat + 768 + 20 * 32;  // [TB * A] (TB*A <= 255*... see host check)
  // value head: Linear(HW,20) + LeakyReLU
  for (int k = tid; k < nb * 20; k += NT) {
    const int bi = k / 20, u = k - bi * 20;
    float s = p.b_v1[u];
    const float* w = p.w_v1 + u * HW;
    const float* f = feat + bi * HW;
    for (int c = 0; c < HW; ++c) s = fmaf(f[c], w[c], s);
    hid[k] = leaky(s, slope);
  }
  // policy head: Linear(2*HW, A) on the (c, y, x)-flattened planes
  for (int k = tid; k < nb * p.A; k += NT) {
    const int bi = k / p.A, a = k - bi * p.A;
    float s = p.b_p[a];
    const float* w = p.w_p + (size_t)a * 2 * HW;
    const float* f0 = feat + 256 + bi * HW;
    const float* f1 = feat + 512 + bi * HW;
    for (int c = 0; c < HW; ++c) s = fmaf(f0[c], w[c], s);
    for (int c = 0; c < HW; ++c) s = fmaf(f1[c], w[HW + c], s);
    logit[k] = s;
  }
  __syncthreads();
  // Linear(20,1) + tanh; softmax statistics per board
  float* stat = logit + 256 * 4;  // [TB][2] max, sum  (logit region sized 1024 floats)
  if (tid < nb) {
    float s = p.b_v2[0];
    for (int u = 0; u < 20; ++u) s = fmaf(hid[tid * 20 + u], p.w_v2[u], s);
    values[row0 + board0 + tid] = tanhf(s);
    float mx = -3.4e38f;
    for (int a = 0; a < p.A; ++a) mx = fmaxf(mx, logit[tid * p.A + a]);
    float sum = 0.f;
    for (int a = 0; a < p.A; ++a) sum += expf(logit[tid * p.A + a] - mx);
    stat[2 * tid] = mx;
    stat[2 * tid + 1] = sum;
  }
  __syncthreads();
  for (int k = tid; k < nb * p.A; k += NT) {
    const int bi = k / p.A;
    probs[(size_t)(row0 + board0) * p.A + k] = expf(logit[k] - stat[2 * bi]) / stat[2 * bi + 1];
  }
  if (stamps && tid == 0) {
    stamps[4 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t_c0;
    stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - t_r0;
    stamps[4 * blockIdx.x + 2] = t_trunk0 - t_c0;
    stamps[4 * blockIdx.x + 3] = t_trunk1 - t_c0;
  }
}


// ===================================================================================================
// 3 x bf16 mode (opt-in): the same network function with the 3x3 convolutions evaluated on the bf16 MFMA
// pipe.  Every float32 operand x is split exactly into three bf16 terms x = hi + mid + lo (+ <= 2^-27 |x|),
// and a product is accumulated as hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid in float32: the dropped
// terms are <= 2^-26 relative, i.e. below float32 rounding.  v_mfma_f32_32x32x16_bf16 covers 16 k per 32
// cycles against 2 k per 64 for the f32 form, so the six MFMAs cost 6/16 of the float32 issue time.
// Layout: activations as three bf16 planes per row (24 granules of 16 B, XOR-swizzled), weights per tap as
// [split][co][k] bf16 rows; the MFMA is issued with the WEIGHTS as first operand, so a lane's 16 results are
// 4 groups of 4 consecutive channels of ONE activation row and go back to LDS as 8-byte writes.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int ROW_G = 24;                 // granules per activation row: 3 splits x 8
constexpr int ACT3_G = 256 * ROW_G;       // 6144 granules = 96 KiB
constexpr int W3_G = 3 * 64 * 8;          // 1536 granules = 24 KiB per tap
constexpr int LDS3_G = ACT3_G + 2 * W3_G; // 144 KiB

__device__ __forceinline__ int agran(int row, int s, int g) { return row * ROW_G + s * 8 + (g ^ ((row >> 1) & 7)); }
__device__ __forceinline__ unsigned bf16_rne(float x) {
  const unsigned u = __float_as_uint(x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void split3(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = bf16_rne(x);
  float r = x - __uint_as_float(hi << 16);   // exact
  mid = bf16_rne(r);
  r = r - __uint_as_float(mid << 16);        // exact
  lo = bf16_rne(r);
}
// four float32 values -> three 8-byte groups of bf16
__device__ __forceinline__ void split3x4(const float* x, uint2& H, uint2& M, uint2& L) {
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) split3(x[j], h[j], m[j], l[j]);
  H = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
  M = make_uint2(m[0] | (m[1] << 16), m[2] | (m[3] << 16));
  L = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
}
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }
__device__ __forceinline__ void join3x4(uint2 H, uint2 M, uint2 L, float* x) {
  x[0] = (bf_lo(H.x) + bf_lo(M.x)) + bf_lo(L.x);
  x[1] = (bf_hi(H.x) + bf_hi(M.x)) + bf_hi(L.x);
  x[2] = (bf_lo(H.y) + bf_lo(M.y)) + bf_lo(L.y);
  x[3] = (bf_hi(H.y) + bf_hi(M.y)) + bf_hi(L.y);
}

__global__ __launch_bounds__(NT, 2) void k_net_forward_3x(NetParams p0, NetParams p1,
                                                            const float* __restrict__ planes,
                                                            const int32_t* __restrict__ counts, int which,
                                                            float* __restrict__ probs, float* __restrict__ values) {
  __shared__ uint4 lds[LDS3_G];
  uint4* wbuf = lds + ACT3_G;
  char* actb = reinterpret_cast<char*>(lds);

  int L, row0, board0;
  bool second = false;
  if (which < 2) {
    L = counts[which];
    row0 = which ? counts[0] : 0;
    board0 = blockIdx.x * p0.TB;
  } else {
    const int L0 = counts[0];
    const int t0 = (L0 + p0.TB - 1) / p0.TB;
    second = (int)blockIdx.x >= t0;
    L = second ? counts[1] : L0;
    row0 = second ? L0 : 0;
    board0 = (second ? (int)blockIdx.x - t0 : (int)blockIdx.x) * p0.TB;
  }
  if (board0 >= L) return;
  const NetParams& p = second ? p1 : p0;
  const float slope = p.slope;
  const int nb = min(p.TB, L - board0);
  const int HW = p.HW;
  const int R = nb * HW;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int i = lane & 31, h = lane >> 5;

  for (int k = tid; k < ACT3_G; k += NT) lds[k] = make_uint4(0u, 0u, 0u, 0u);
  float* wf = reinterpret_cast<float*>(wbuf);
  for (int k = tid; k < 9 * 2 * NF; k += NT) wf[k] = p.w_in[k];
  __syncthreads();

  // ---- conv_in (float32 on the VALU), result split into the three planes
  {
    const int r = tid & 255;
    const int chalf = tid >> 8;
    if (r < R) {
      const int bi = r / HW, cell = r - bi * HW;
      const int y = cell / p.W, x = cell - y * p.W;
      const float* pl = planes + (size_t)(row0 + board0 + bi) * 2 * HW;
      float in0[9], in1[9];  // the 18 inputs of this row (statically indexed: stays in registers)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int ny = y + t / 3 - 1, nx = x + t % 3 - 1;
        const bool ok = ny >= 0 && ny < p.H && nx >= 0 && nx < p.W;
        in0[t] = ok ? pl[ny * p.W + nx] : 0.f;
        in1[t] = ok ? pl[HW + ny * p.W + nx] : 0.f;
      }
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) {
        const int c4 = chalf * 8 + cc;
        float o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) o[u] = p.b_in[c4 * 4 + u];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const float i0 = in0[t], i1 = in1[t];
          const float4 w0 = *reinterpret_cast<const float4*>(wf + (2 * t) * NF + c4 * 4);
          const float4 w1 = *reinterpret_cast<const float4*>(wf + (2 * t + 1) * NF + c4 * 4);
          o[0] = fmaf(i0, w0.x, o[0]); o[1] = fmaf(i0, w0.y, o[1]); o[2] = fmaf(i0, w0.z, o[2]); o[3] = fmaf(i0, w0.w, o[3]);
          o[0] = fmaf(i1, w1.x, o[0]); o[1] = fmaf(i1, w1.y, o[1]); o[2] = fmaf(i1, w1.z, o[2]); o[3] = fmaf(i1, w1.w, o[3]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) o[u] = leaky(o[u], slope);
        uint2 H, M, Lo;
        split3x4(o, H, M, Lo);
        const int g = c4 >> 1, half = (c4 & 1) * 8;
        *reinterpret_cast<uint2*>(actb + agran(r, 0, g) * 16 + half) = H;
        *reinterpret_cast<uint2*>(actb + agran(r, 1, g) * 16 + half) = M;
        *reinterpret_cast<uint2*>(actb + agran(r, 2, g) * 16 + half) = Lo;
      }
    }
  }
  __syncthreads();
  // ---- stage the weights of tap 0
#pragma unroll
  for (int m = 0; m < 3; ++m) wbuf[tid + NT * m] = p.w3[tid + NT * m];
  __syncthreads();

  const int myrow = wave * 32 + i;
  const bool rvalid = myrow < R;
  const int rbi = myrow / HW;
  const int rcell = myrow - rbi * HW;
  const int ry = rcell / p.W, rx = rcell - ry * p.W;
  const int wswz = (i >> 1) & 7;

  f32x16 acc0, acc1;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    acc0[e] = 0.f;
    acc1[e] = 0.f;
  }
  uint4 wn0, wn1, wn2;
  wn0 = wn1 = wn2 = make_uint4(0u, 0u, 0u, 0u);
  for (int ft = 0; ft < NTAPS; ++ft) {
    const int cur = ft & 1;
    const int layer = ft / 9, tap = ft % 9;
    const bool has_next = ft + 1 < NTAPS;
    if (has_next) {  // issue early
      const uint4* src = p.w3 + (size_t)(ft + 1) * W3_G;
      wn0 = src[tid];
      wn1 = src[tid + NT];
      wn2 = src[tid + 2 * NT];
    }
    const int ny = ry + tap / 3 - 1, nx = rx + tap % 3 - 1;
    const bool ok = rvalid && ny >= 0 && ny < p.H && nx >= 0 && nx < p.W;
    const int nrow = ok ? rbi * HW + ny * p.W + nx : ZROW;
    const uint4* arow = lds + nrow * ROW_G;
    const int aswz = (nrow >> 1) & 7;
    const uint4* w0 = wbuf + cur * W3_G + i * 8;          // co = i      (col tile 0), split s adds 64*8 granules
    const uint4* w1 = wbuf + cur * W3_G + (32 + i) * 8;   // co = 32 + i (col tile 1)
#define CARO_LOAD3(AH, AM, AL, W0H, W0M, W0L, W1H, W1M, W1L, KB)                    \
  {                                                                                  \
    const int ga = ((2 * (KB) + h) ^ aswz), gw = ((2 * (KB) + h) ^ wswz);            \
    AH = arow[ga]; AM = arow[8 + ga]; AL = arow[16 + ga];                             \
    W0H = w0[gw]; W0M = w0[512 + gw]; W0L = w0[1024 + gw];                            \
    W1H = w1[gw]; W1M = w1[512 + gw]; W1L = w1[1024 + gw];                            \
  }
#define CARO_BF(X) __builtin_bit_cast(bf16x8, X)
#define CARO_MFMA3(ACC, WH, WM, WL, AH, AM, AL)                                                      \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(CARO_BF(WM), CARO_BF(AM), ACC, 0, 0, 0);             \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(CARO_BF(WH), CARO_BF(AL), ACC, 0, 0, 0);             \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(CARO_BF(WL), CARO_BF(AH), ACC, 0, 0, 0);             \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(CARO_BF(WH), CARO_BF(AM), ACC, 0, 0, 0);             \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(CARO_BF(WM), CARO_BF(AH), ACC, 0, 0, 0);             \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(CARO_BF(WH), CARO_BF(AH), ACC, 0, 0, 0);
    uint4 xah, xam, xal, xw0h, xw0m, xw0l, xw1h, xw1m, xw1l;
    uint4 yah, yam, yal, yw0h, yw0m, yw0l, yw1h, yw1m, yw1l;
    CARO_LOAD3(xah, xam, xal, xw0h, xw0m, xw0l, xw1h, xw1m, xw1l, 0)
#pragma unroll
    for (int kb = 0; kb < 4; kb += 2) {
      CARO_LOAD3(yah, yam, yal, yw0h, yw0m, yw0l, yw1h, yw1m, yw1l, kb + 1)
      __builtin_amdgcn_sched_barrier(0);
      CARO_MFMA3(acc0, xw0h, xw0m, xw0l, xah, xam, xal)
      CARO_MFMA3(acc1, xw1h, xw1m, xw1l, xah, xam, xal)
      __builtin_amdgcn_sched_barrier(0);
      if (kb + 2 < 4) CARO_LOAD3(xah, xam, xal, xw0h, xw0m, xw0l, xw1h, xw1m, xw1l, kb + 2)
      __builtin_amdgcn_sched_barrier(0);
      CARO_MFMA3(acc0, yw0h, yw0m, yw0l, yah, yam, yal)
      CARO_MFMA3(acc1, yw1h, yw1m, yw1l, yah, yam, yal)
      __builtin_amdgcn_sched_barrier(0);
    }
#undef CARO_LOAD3
#undef CARO_MFMA3
#undef CARO_BF
    if (has_next) {  // write late: the other buffer was last read one tap ago
      uint4* dst = wbuf + (cur ^ 1) * W3_G;
      dst[tid] = wn0;
      dst[tid + NT] = wn1;
      dst[tid + 2 * NT] = wn2;
    }
    if (tap == 8) {
      __syncthreads();  // every wave has read this layer's input activations
      // epilogue, in place: v = v + leaky(conv(v) + b); a lane holds 4 x 4 consecutive channels of its row
      const float* bias = p.b_res + layer * NF;
      const int row = wave * 32 + i;
      const bool real = row < R;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c0 = ct * 32 + 8 * q + 4 * h;
          const int g = c0 >> 3, half = (c0 & 7) * 2;
          char* ph = actb + agran(row, 0, g) * 16 + half;
          char* pm = actb + agran(row, 1, g) * 16 + half;
          char* pl3 = actb + agran(row, 2, g) * 16 + half;
          float oldv[4], nv[4];
          join3x4(*reinterpret_cast<uint2*>(ph), *reinterpret_cast<uint2*>(pm), *reinterpret_cast<uint2*>(pl3), oldv);
          const float4 bc = *reinterpret_cast<const float4*>(bias + c0);
          const float bb[4] = {bc.x, bc.y, bc.z, bc.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float a = ct == 0 ? acc0[4 * q + j] : acc1[4 * q + j];
            nv[j] = real ? oldv[j] + leaky(a + bb[j], slope) : 0.f;
          }
          uint2 H, M, Lo;
          split3x4(nv, H, M, Lo);
          *reinterpret_cast<uint2*>(ph) = H;
          *reinterpret_cast<uint2*>(pm) = M;
          *reinterpret_cast<uint2*>(pl3) = Lo;
        }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        acc0[e] = 0.f;
        acc1[e] = 0.f;
      }
    }
    __syncthreads();  // staged weights / new activations visible to every wave
  }

  // ---- heads: reconstruct float32 activations of the row, then as the float32 kernel
  float* feat = reinterpret_cast<float*>(wbuf);  // [3][256]
  {
    const int r = tid;
    if (r < R) {
      float s0 = p.b_head[0], s1 = p.b_head[1], s2 = p.b_head[2];
      for (int g = 0; g < 8; ++g) {
        const uint4 H = lds[agran(r, 0, g)], M = lds[agran(r, 1, g)], Lo = lds[agran(r, 2, g)];
        float v[8];
        join3x4(make_uint2(H.x, H.y), make_uint2(M.x, M.y), make_uint2(Lo.x, Lo.y), v);
        join3x4(make_uint2(H.z, H.w), make_uint2(M.z, M.w), make_uint2(Lo.z, Lo.w), v + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = g * 8 + j;
          s0 = fmaf(v[j], p.w_head[c], s0);
          s1 = fmaf(v[j], p.w_head[NF + c], s1);
          s2 = fmaf(v[j], p.w_head[2 * NF + c], s2);
        }
      }
      feat[r] = leaky(s0, slope);
      feat[256 + r] = leaky(s1, slope);
      feat[512 + r] = leaky(s2, slope);
    }
  }
  __syncthreads();
  float* hid = feat + 768;
  float* logit = feat + 768 + 20 * 32;
  for (int k = tid; k < nb * 20; k += NT) {
    const int bi = k / 20, u = k - bi * 20;
    float s = p.b_v1[u];
    const float* w = p.w_v1 + u * HW;
    const float* f = feat + bi * HW;
    for (int c = 0; c < HW; ++c) s = fmaf(f[c], w[c], s);
    hid[k] = leaky(s, slope);
  }
  for (int k = tid; k < nb * p.A; k += NT) {
    const int bi = k / p.A, a = k - bi * p.A;
    float s = p.b_p[a];
    const float* w = p.w_p + (size_t)a * 2 * HW;
    const float* f0 = feat + 256 + bi * HW;
    const float* f1 = feat + 512 + bi * HW;
    for (int c = 0; c < HW; ++c) s = fmaf(f0[c], w[c], s);
    for (int c = 0; c < HW; ++c) s = fmaf(f1[c], w[HW + c], s);
    logit[k] = s;
  }
  __syncthreads();
  float* stat = logit + 256 * 4;
  if (tid < nb) {
    float s = p.b_v2[0];
    for (int u = 0; u < 20; ++u) s = fmaf(hid[tid * 20 + u], p.w_v2[u], s);
    values[row0 + board0 + tid] = tanhf(s);
    float mx = -3.4e38f;
    for (int a = 0; a < p.A; ++a) mx = fmaxf(mx, logit[tid * p.A + a]);
    float sum = 0.f;
    for (int a = 0; a < p.A; ++a) sum += expf(logit[tid * p.A + a] - mx);
    stat[2 * tid] = mx;
    stat[2 * tid + 1] = sum;
  }
  __syncthreads();
  for (int k = tid; k < nb * p.A; k += NT) {
    const int bi = k / p.A;
    probs[(size_t)(row0 + board0) * p.A + k] = expf(logit[k] - stat[2 * bi]) / stat[2 * bi + 1];
  }
}

}  // namespace cnet

// =================================================================== host side
extern "C" void caro__set_error(const char* msg);  // caro_engine.hip

struct caro_net {
  cnet::NetParams p;
  float* dev;
  uint4* w3_dev;  // split residual weights (3xbf16 mode), or null
  int device;
};

static int nfail(int code, const std::string& m) {
  caro__set_error(m.c_str());
  return code;
}

extern "C" {

// number of floats caro_net_create expects for an H x W board with A actions
int64_t caro_net_packed_size(int H, int W, int A) {
  const int64_t HW = (int64_t)H * W;
  return 9 * 2 * 64 + 64 + (int64_t)cnet::NRES * 9 * cnet::WCHUNK + cnet::NRES * 64 + 3 * 64 + 3 +
         20 * HW + 20 + 20 + 1 + (int64_t)A * 2 * HW + A;
}

int caro_net_create(int H, int W, int A, float negative_slope, const float* packed_host, int64_t n_floats,
                    int device_id, caro_net** out) {
  if (!packed_host || !out) return nfail(CARO_E_INVAL, "null argument");
  if (H < 2 || W < 2 || H > 15 || W > 15 || A < 1 || A > 255) return nfail(CARO_E_INVAL, "unsupported board / action count");
  if (n_floats != caro_net_packed_size(H, W, A)) return nfail(CARO_E_INVAL, "packed weight buffer has the wrong size");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return nfail(CARO_E_NODEV, "no HIP device: libcaro_hip needs a GPU (there is no CPU fallback)");
  if (device_id < 0 || device_id >= ndev) return nfail(CARO_E_INVAL, "device_id out of range");
  if (hipSetDevice(device_id) != hipSuccess) return nfail(CARO_E_HIP, "hipSetDevice failed");
  caro_net* n = new caro_net();
  n->device = device_id;
  n->w3_dev = nullptr;
  const size_t pad = (size_t)cnet::TPC * cnet::WCHUNK;  // k_net_forward reads one chunk past the last tap
  if (hipMalloc((void**)&n->dev, (n_floats + pad) * sizeof(float)) != hipSuccess) {
    delete n;
    return nfail(CARO_E_NOMEM, "hipMalloc failed");
  }
  if (hipMemset(n->dev + n_floats, 0, pad * sizeof(float)) != hipSuccess ||
      hipMemcpy(n->dev, packed_host, n_floats * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(n->dev);
    delete n;
    return nfail(CARO_E_HIP, "hipMemcpy failed");
  }
  cnet::NetParams& p = n->p;
  const int HW = H * W;
  p.H = H; p.W = W; p.HW = HW; p.A = A; p.TB = 255 / HW; p.slope = negative_slope;
  if (p.TB * A > 1024 || p.TB > 32) p.TB = p.TB > 32 ? 32 : p.TB;
  const float* q = n->dev;
  p.w_in = q;   q += 9 * 2 * 64;
  p.b_in = q;   q += 64;
  p.w_res = q;  q += (size_t)cnet::NRES * 9 * cnet::WCHUNK;
  p.b_res = q;  q += cnet::NRES * 64;
  p.w_head = q; q += 3 * 64;
  p.b_head = q; q += 3;
  p.w_v1 = q;   q += 20 * HW;
  p.b_v1 = q;   q += 20;
  p.w_v2 = q;   q += 20;
  p.b_v2 = q;   q += 1;
  p.w_p = q;    q += (size_t)A * 2 * HW;
  p.b_p = q;    q += A;
  p.w3 = nullptr;
  *out = n;
  return 0;
}

/* opt-in 3xbf16 mode: upload the split residual weights (45 taps x 1536 granules x 8 bf16, LDS image order of
 * k_net_forward_3x, packed by caro_ai_amd/net_hip.py); from then on the forward calls of this net use the
 * bf16 MFMA pipe for the 3x3 convolutions (float32 accumulate, error below float32 rounding). */
int caro_net_enable_3xbf16(caro_net* n, const uint16_t* w3_host, int64_t n_u16) {
  if (!n || !w3_host) return nfail(CARO_E_INVAL, "null argument");
  const int64_t want = (int64_t)cnet::NTAPS * cnet::W3_G * 8;
  if (n_u16 != want) return nfail(CARO_E_INVAL, "split weight image has the wrong size");
  if (hipSetDevice(n->device) != hipSuccess) return nfail(CARO_E_HIP, "hipSetDevice failed");
  if (!n->w3_dev && hipMalloc((void**)&n->w3_dev, want * 2) != hipSuccess) return nfail(CARO_E_NOMEM, "hipMalloc failed");
  if (hipMemcpy(n->w3_dev, w3_host, want * 2, hipMemcpyHostToDevice) != hipSuccess) return nfail(CARO_E_HIP, "hipMemcpy failed");
  n->p.w3 = n->w3_dev;
  return 0;
}

void caro_net_destroy(caro_net* n) {
  if (!n) return;
  if (n->w3_dev) (void)hipFree(n->w3_dev);
  (void)hipFree(n->dev);
  delete n;
}

int caro_net_boards_per_workgroup(const caro_net* n) { return n ? n->p.TB : 0; }

int caro_net_forward(caro_net* n, const float* planes_dev, const int32_t* counts_dev, int which, int64_t max_rows,
                     float* probs_dev, float* values_dev, void* stream) {
  if (!n || !planes_dev || !counts_dev || !probs_dev || !values_dev) return nfail(CARO_E_INVAL, "null argument");
  if (which != 0 && which != 1) return nfail(CARO_E_INVAL, "which must be 0 or 1");
  if (max_rows <= 0) return 0;
  const unsigned grid = (unsigned)((max_rows + n->p.TB - 1) / n->p.TB);
  if (n->p.w3)
    hipLaunchKernelGGL(cnet::k_net_forward_3x, dim3(grid), dim3(cnet::NT), 0, (hipStream_t)stream, n->p, n->p,
                       planes_dev, counts_dev, which, probs_dev, values_dev);
  else
    hipLaunchKernelGGL(cnet::k_net_forward, dim3(grid), dim3(cnet::NT), 0, (hipStream_t)stream, n->p, n->p,
                       planes_dev, counts_dev, which, probs_dev, values_dev, (unsigned long long*)nullptr);
  if (hipGetLastError() != hipSuccess) return nfail(CARO_E_HIP, "k_net_forward launch failed");
  return 0;
}

/* both nets of an arena in one launch: rows [0, L0) through n0, rows [L0, L0+L1) through n1 */
int caro_net_forward_pair(caro_net* n0, caro_net* n1, const float* planes_dev, const int32_t* counts_dev,
                          int64_t max_rows, float* probs_dev, float* values_dev, void* stream) {
  if (!n0 || !n1 || !planes_dev || !counts_dev || !probs_dev || !values_dev) return nfail(CARO_E_INVAL, "null argument");
  if (n0->p.H != n1->p.H || n0->p.W != n1->p.W || n0->p.A != n1->p.A) return nfail(CARO_E_INVAL, "nets differ in shape");
  if (max_rows <= 0) return 0;
  if ((n0->p.w3 == nullptr) != (n1->p.w3 == nullptr)) return nfail(CARO_E_INVAL, "nets differ in arithmetic mode");
  const unsigned grid = (unsigned)((max_rows + n0->p.TB - 1) / n0->p.TB + 1);  // +1: each class rounds up
  if (n0->p.w3)
    hipLaunchKernelGGL(cnet::k_net_forward_3x, dim3(grid), dim3(cnet::NT), 0, (hipStream_t)stream, n0->p, n1->p,
                       planes_dev, counts_dev, 2, probs_dev, values_dev);
  else
    hipLaunchKernelGGL(cnet::k_net_forward, dim3(grid), dim3(cnet::NT), 0, (hipStream_t)stream, n0->p, n1->p,
                       planes_dev, counts_dev, 2, probs_dev, values_dev, (unsigned long long*)nullptr);
  if (hipGetLastError() != hipSuccess) return nfail(CARO_E_HIP, "k_net_forward launch failed");
  return 0;
}

/* diagnostic: same launch, and per workgroup (total cycles, 100 MHz ticks, cycles at trunk start, at trunk end) into stamps_dev u64[4*grid] */
int caro_net_forward_stamped(caro_net* n, const float* planes_dev, const int32_t* counts_dev, int which,
                             int64_t max_rows, float* probs_dev, float* values_dev, uint64_t* stamps_dev,
                             void* stream) {
  if (!n || !stamps_dev) return nfail(CARO_E_INVAL, "null argument");
  const unsigned grid = (unsigned)((max_rows + n->p.TB - 1) / n->p.TB);
  hipLaunchKernelGGL(cnet::k_net_forward, dim3(grid), dim3(cnet::NT), 0, (hipStream_t)stream, n->p, n->p, planes_dev,
                     counts_dev, which, probs_dev, values_dev, (unsigned long long*)stamps_dev);
  if (hipGetLastError() != hipSuccess) return nfail(CARO_E_HIP, "k_net_forward launch failed");
  return 0;
}

}  // extern "C"
