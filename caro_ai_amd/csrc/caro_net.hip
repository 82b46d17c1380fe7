// caro_net.hip -- fused float32 inference of the policy/value net (reference
// lib/model.py:10-94, eval-mode batch-norm folded) for the leaf batch of the
// self-play engine, as ONE kernel per minibatch.
//
// Why a kernel of our own: the leaf count L changes every minibatch and lives in
// device memory; a torch forward needs L on the host (a stream sync per
// minibatch) and ~45 launches; MIOpen has no gfx950 database in this image.
// Here the grid is sized for the maximum and every workgroup reads L itself.
//
// Three kernels share conv_in, the heads and the launch interface (all float32 on v_mfma_f32_32x32x2_f32):
//   k_net_forward_w  (default)  3x3 convolutions in row-Winograd F(2,3) form -- see trunk_w
//   k_net_forward_w2 (default on 13x13 .. 15x15 boards)  2-D Winograd F(2x2,3x3) form + k_net_heads -- see trunk_w2d
//   k_net_forward               direct 3x3 form (the round-1 kernel; `--net hip`, the A/B baseline of the Winograd forms)
// (A fourth, direct convolutions on the bf16 pipe with three-way split operands, was removed in round 5: untuned since
// round 1, no large-board or K-split form, and the headline stays on the float32 pipe.  git history has it.)
// Common structure (one workgroup = 512 threads = 8 waves = one CU, two waves per SIMD; TB boards):
//   rows r = board*HW + cell, at most 255 real rows; row 255 is a permanent zero
//   row (3x3 padding).  Activations X[row][64] float32 stay in LDS for the whole
//   trunk in ONE 64 KiB buffer (XOR-swizzled 16-byte granules) that is updated in
//   place: a layer's outputs live in the MFMA accumulators until every wave has
//   finished reading the layer's input, then overwrite it.  The 3x3 convolutions
//   are implicit GEMMs on v_mfma_f32_32x32x2_f32 with the WEIGHTS as first operand.
//   K order inside a tap: MFMA k-half h = lane>>5 carries channel 32h + j, so a
//   lane's operands for 4 consecutive k-steps are one ds_read_b128.
//   Weights stream from L2 through the other 96 KiB of LDS (k_net_forward: two buffers of three taps, staged through
//   registers; k_net_forward_w: a ring of three 32 KiB chunks filled by global_load_lds, one workgroup barrier per
//   chunk plus two per layer around the in-place epilogue).
//   conv_in (K = 18: k_net_forward_w runs it on the matrix pipe too, one MFMA per tap; the other two kernels on the
//   VALU), the 1x1 heads, the two FC heads, tanh and the softmax run in the same kernel.
// float32 throughout: MFMA f32 is an exact fma chain in k order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/caro_hip.h"
#include "../../include/caro_noise.h"

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
  const float* w_pT;    // w_p quad-transposed (made at upload: per plane [HW/4][A][4] + [HW%4][A]), for boards whose head block is not staged in LDS
  const float* ww;      // f32w mode: [5][3 dx][2 granule halves] chunks of [4 p][2 h][64 co][16] transformed residual weights (LDS image order of trunk_w), or null
  const uint32_t* wtab; // f32w mode: [128] tile of MFMA row (row tile, lane): board | ty << 8 | x << 16 | valid << 24
  const float* ww2;     // f32w2 mode: [5][8 chunks][2 b][4 a][2 h][64 co][8] 2-D Winograd F(2x2,3x3) transformed residual weights (LDS image order of trunk_w2d), or null
  const uint16_t* wx3;  // bf16x3 mode: [45 taps][2 c][3 parts][4 kg][64 co][8 ci] bfloat16 split residual weights (k_net_forward_x3), or null
  int ncu, TB2, TB4;    // f32w mode: compute units; boards per workgroup of the 2- / 4-way K-split overflow tiles (0: off)
};

__device__ __forceinline__ int aoff(int row, int c) {
  return row * NF + ((((c >> 2) ^ (row & 15)) << 2) | (c & 3));
}
__device__ __forceinline__ float leaky(float x, float slope) { return x > 0.f ? x : x * slope; }

// Swizzle key of activation row r: granule g of the row lives in 16-byte slot g ^ key.  The default (r & 15) serves
// the row-oriented kernels.  K2D (one board per workgroup, the 2-D Winograd trunk): the 16 lanes of a ds_read_b128
// group there read the same cell offset of 16 tiles (8 tile columns x 2 tile rows), so the key is built from the
// cell's tile coordinates -- ((x + 1) >> 1) & 7 | ((y + 1) >> 1 & 1) << 3 -- and is distinct over such a group.
template <bool K2D>
__device__ __forceinline__ int akey(int r, int W) {
  if constexpr (!K2D) return r & 15;
  const int y = r / W, x = r - y * W;
  return (((x + 1) >> 1) & 7) | ((((y + 1) >> 1) & 1) << 3);
}

constexpr int NT = 512;  // threads per workgroup

// conv_in on the VALU (lib/model.py:24-28 folded): two threads per row, 32 output channels each.
// `planes` = the launch's plane rows, `smap[bi]` = row of this workgroup's board bi (LDS), `win` = the [9][2][64]
// weights staged in LDS.
__device__ __forceinline__ void conv_in_f32(const NetParams& p, const float* __restrict__ planes, const int* smap,
                                            float* act, const float* win, int R, int tid) {
  const int HW = p.HW;
  const float slope = p.slope;
  const int r = tid & 255;
  const int chalf = tid >> 8;
  if (r < R) {
    const int bi = r / HW, cell = r - bi * HW;
    const int y = cell / p.W, x = cell - y * p.W;
    const float* pl = planes + (size_t)smap[bi] * 2 * HW;
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
        const float4 w0 = *reinterpret_cast<const float4*>(win + (2 * t) * NF + c4 * 4);
        const float4 w1 = *reinterpret_cast<const float4*>(win + (2 * t + 1) * NF + c4 * 4);
        o[0] = fmaf(i0, w0.x, o[0]); o[1] = fmaf(i0, w0.y, o[1]); o[2] = fmaf(i0, w0.z, o[2]); o[3] = fmaf(i0, w0.w, o[3]);
        o[0] = fmaf(i1, w1.x, o[0]); o[1] = fmaf(i1, w1.y, o[1]); o[2] = fmaf(i1, w1.z, o[2]); o[3] = fmaf(i1, w1.w, o[3]);
      }
      float4 out = make_float4(leaky(o[0], slope), leaky(o[1], slope), leaky(o[2], slope), leaky(o[3], slope));
      *reinterpret_cast<float4*>(act + r * NF + ((c4 ^ (r & 15)) << 2)) = out;
    }
  }
}

// conv_in on the matrix pipe (k_net_forward_w): the same K = 18 dot products as conv_in_f32 -- bias, then for every tap
// the product of plane 0, then of plane 1, which is the k order of one v_mfma_f32_32x32x2_f32 per tap (k = plane) with
// the weights as first operand, i.e. the same fma chain -- as 9 MFMAs per (32 rows x 32 channels) instead of 576 fma
// per thread whose 144 weight reads per thread kept the LDS return path busy for 9 k cycles.  Wave = column tile
// (wave & 1) x row tiles (wave >> 1) and (wave >> 1) + 4; lane (i, h) feeds row 32 rt + i with plane h: 9 loads per row
// tile instead of 18.  Output layout = the trunk's (a lane holds 4 groups of 4 consecutive channels of its row).
template <bool K2D = false, int NTH = NT>
__device__ __forceinline__ void conv_in_mfma(const NetParams& p, const float* __restrict__ planes, const int* smap,
                                             float* act, const float* win, int R, int tid) {
  const int HW = p.HW;
  const float slope = p.slope;
  const int wave = tid >> 6, lane = tid & 63;
  const int i = lane & 31, h = lane >> 5;
  const int ct = wave & 1;
  float wa[9];  // A operands: w[co = ct*32 + i][k = (tap, plane h)], win = [tap][plane][64]
#pragma unroll
  for (int t = 0; t < 9; ++t) wa[t] = win[(2 * t + h) * NF + ct * 32 + i];
  float4 bias[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) bias[q] = *reinterpret_cast<const float4*>(p.b_in + ct * 32 + 8 * q + 4 * h);
  constexpr int RTS = NTH / 128;  // row tiles taken per pass: the workgroup's waves / 2 column tiles
#pragma unroll
  for (int half = 0; half < 8 / RTS; ++half) {
    const int rt = (wave >> 1) + RTS * half;
    if (rt * 32 >= R) continue;  // uniform per wave: no real row in this tile
    const int r = rt * 32 + i;
    const bool rv = r < R;
    const int bi = rv ? r / HW : 0, cell = rv ? r - bi * HW : 0;
    const int y = cell / p.W, x = cell - y * p.W;
    const float* pl = planes + (size_t)smap[bi] * 2 * HW + h * HW;
    float in[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int ny = y + t / 3 - 1, nx = x + t % 3 - 1;
      const bool ok = rv && ny >= 0 && ny < p.H && nx >= 0 && nx < p.W;
      in[t] = ok ? pl[ny * p.W + nx] : 0.f;
    }
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc[4 * q] = bias[q].x; acc[4 * q + 1] = bias[q].y; acc[4 * q + 2] = bias[q].z; acc[4 * q + 3] = bias[q].w;
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[t], in[t], acc, 0, 0, 0);
    if (rv) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c4 = ct * 8 + 2 * q + h;
        const float4 out = make_float4(leaky(acc[4 * q], slope), leaky(acc[4 * q + 1], slope),
                                       leaky(acc[4 * q + 2], slope), leaky(acc[4 * q + 3], slope));
        *reinterpret_cast<float4*>(act + r * NF + ((c4 ^ akey<K2D>(r, p.W)) << 2)) = out;
      }
    }
  }
}

// 1x1 heads, the two FC heads, tanh and the softmax (lib/model.py:44-67, lib/mcts.py:216) from the trunk
// output in `act`; `scratch` = the (now free) weight stage.  probs / values are the launch's output rows;
// `slot_v` = output row of board `tid` (threads tid < nb), re-published through the scratch for the prob rows.
// The head parameters [w_head 3x64][b_head 3][w_v1 20 HW][b_v1 20][w_v2 20][b_v2 1][w_p A x 2HW][b_p A] are one
// contiguous block of the packed buffer.  STAGED: the block already sits in LDS at scratch + HEAD_STAGE_AT (k_net_forward_w
// fetches it during the trunk's last chunks when it fits) -- the dot products then never wait for L2.
// Every output is the same fma chain in the same order in both forms; only who computes what changed: the value and
// the policy rows of the FC stage, and the tanh and the softmax terms, run on different waves side by side.
constexpr int HEAD_STAGE_AT = 4096;   // floats into the scratch
constexpr int HEAD_STAGE_MAX = 4096;  // floats
__device__ __forceinline__ int head_span(int HW, int A) { return 3 * NF + 3 + 20 * HW + 20 + 20 + 1 + A * 2 * HW + A; }
static inline int head_span_host(int HW, int A) { return 3 * NF + 3 + 20 * HW + 20 + 20 + 1 + A * 2 * HW + A; }

template <bool STAGED, bool K2D = false, int NTH = NT>
__device__ __forceinline__ void heads_f32(const NetParams& p, const float* act, float* scratch,
                                          float* __restrict__ probs, float* __restrict__ values, int slot_v, int nb,
                                          int R, int tid, float* __restrict__ featbuf = nullptr,
                                          int32_t* __restrict__ rowlist = nullptr, int dense0 = 0) {
  const int HW = p.HW, A = p.A;
  const float slope = p.slope;
  // !STAGED (boards from 7x7 up): the parameters up to b_v2 (at most 4 740 floats) are copied to the place of the staged
  // block now, all loads in flight together: the 20 threads of a board's value head each walked a row of w_v1 in global
  // memory (225 dependent batches of loads at 15x15)
  constexpr int NSM = (3 * NF + 3 + 20 * 225 + 20 + 20 + 1 + NTH - 1) / NTH;  // floats per thread at the largest board
  const int nsmall = 3 * NF + 3 + 20 * HW + 20 + 20 + 1;
  if (!STAGED && !featbuf) {
    float tmp[NSM];
#pragma unroll
    for (int u = 0; u < NSM; ++u) tmp[u] = tid + u * NTH < nsmall ? p.w_head[tid + u * NTH] : 0.f;
#pragma unroll
    for (int u = 0; u < NSM; ++u)
      if (tid + u * NTH < nsmall) scratch[HEAD_STAGE_AT + tid + u * NTH] = tmp[u];
    __syncthreads();
  }
  // (featbuf: only the 1x1 convolutions are computed here -- their 195 parameters come straight from the packed buffer)
  const float* hp = featbuf ? p.w_head : scratch + HEAD_STAGE_AT;
  const float* w_head = hp;
  const float* b_head = w_head + 3 * NF;
  const float* w_v1 = b_head + 3;
  const float* b_v1 = w_v1 + 20 * HW;
  const float* w_v2 = b_v1 + 20;
  const float* b_v2 = w_v2 + 20;
  const float* w_p = STAGED ? b_v2 + 1 : p.w_head + nsmall;
  const float* b_p = w_p + (size_t)A * 2 * HW;
  float* feat = scratch;  // [3][256]: value plane, policy plane 0, policy plane 1 (row indexed)
  /*@HST_BEGIN(scratch, tid)*/
  /*@HST(0)*/
  float* hid = feat + 768;              // [TB][20]
  float* logit = feat + 768 + 20 * 32;  // [TB * A] (TB * A <= 1024, see caro_net_create)
  float* stat = logit + 256 * 4;        // [TB][2] max, sum  (logit region sized 1024 floats)
  int* omap = reinterpret_cast<int*>(stat + 64);  // [TB] output rows
  float* ebuf = stat + 128;             // [TB * A] exp(logit - max)
  if (tid < nb) omap[tid] = slot_v;
  // the 1x1 convolutions: 3 R dot products of 64 (row r, plane o), each a chain in channel order, dealt to all threads
  for (int q = tid; q < 3 * R; q += NTH) {
    const int o = q / R, r = q - o * R;
    const float* wo = w_head + o * NF;
    float s0 = b_head[o];
    const int rkey = akey<K2D>(r, p.W);
#pragma unroll 8
    for (int g = 0; g < 16; ++g) {  // eight pairs of reads in flight, the fma chain in channel order
      const float4 v = *reinterpret_cast<const float4*>(act + r * NF + ((g ^ rkey) << 2));
      const float4 w = *reinterpret_cast<const float4*>(wo + g * 4);
      s0 = fmaf(v.x, w.x, s0); s0 = fmaf(v.y, w.y, s0);
      s0 = fmaf(v.z, w.z, s0); s0 = fmaf(v.w, w.w, s0);
    }
    feat[o * 256 + r] = leaky(s0, slope);
  }
  __syncthreads();
  if (featbuf) {
    // Batched heads (k_net_heads: large boards, one board per workgroup): the board's three feature planes go to row
    // `dense` of the launch's feature buffer, its output row to the row list; the two FC heads, tanh and the softmax
    // of 32 boards at a time follow in their own launch, which reads the policy matrix once per 32 boards instead of
    // once per board (405 KB at 15x15: this stage was 10 k cycles of every workgroup's 275 k, streaming at the compute
    // unit's L2 bandwidth).
    for (int q = tid; q < 3 * R; q += NTH) {
      const int o = q / R, r = q - o * R;
      const int bi = r / HW, c = r - bi * HW;
      featbuf[((size_t)(dense0 + bi) * 3 + o) * HW + c] = feat[o * 256 + r];
    }
    if (tid < nb) rowlist[dense0 + tid] = slot_v;
    return;
  }
  /*@HST(1)*/
  /*@PST(9)*/
  // value head: Linear(HW,20) + LeakyReLU -- threads from 0 up
  for (int k = tid; k < nb * 20; k += NTH) {
    const int bi = k / 20, u = k - bi * 20;
    float s = b_v1[u];
    const float* w = w_v1 + u * HW;
    const float* f = feat + bi * HW;
#pragma unroll 8
    for (int c = 0; c < HW; ++c) s = fmaf(f[c], w[c], s);  // reads of eight steps in flight, the chain in cell order
    hid[k] = leaky(s, slope);
  }
  // policy head: Linear(2*HW, A) on the (c, y, x)-flattened planes -- threads from the middle up, beside the value rows
  for (int k = (tid + NTH / 2) % NTH; k < nb * A; k += NTH) {
    const int bi = k / A, a = k - bi * A;
    float s = b_p[a];
    const float* f0 = feat + 256 + bi * HW;
    const float* f1 = feat + 512 + bi * HW;
    if (STAGED || !p.w_pT) {
      const float* w = w_p + (size_t)a * 2 * HW;
#pragma unroll 8
      for (int c = 0; c < HW; ++c) s = fmaf(f0[c], w[c], s);
#pragma unroll 8
      for (int c = 0; c < HW; ++c) s = fmaf(f1[c], w[HW + c], s);
    } else {
      // large boards (the matrix is 405 KB at 15x15 and stays in L2): the same chain from a QUAD-TRANSPOSED image of the
      // matrix (made at upload: per plane [cell / 4][A][4], then the HW % 4 last cells as [cell][A]), so that the lanes
      // of a wave -- consecutive outputs a -- read consecutive 16-byte granules and a lane gets four cells per load.
      // Row-major, every lane walked its own 1.8 KB row (64 cache lines per wave instruction); cell-major with one
      // float per load (round 2) the chain waited for 450 loads, 16 in flight: 60 cycles per cell.
      const int nq = HW >> 2, rem = HW & 3;
      const size_t half = ((size_t)HW * A + 3) & ~(size_t)3;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        const float* f = pl ? f1 : f0;
        const float* wh = p.w_pT + pl * half;
        const float4* wq = reinterpret_cast<const float4*>(wh) + a;
        // (eight granules in flight per lane; sixteen, or the next batch requested under the current one's 32 dependent
        // fma, ran slower: 13.4-13.8 k cycles for the FC stage against 10.2 k -- the stage streams the 405 KB matrix at
        // 40 bytes per cycle and compute unit, more requests in flight only evict each other)
#pragma unroll 8
        for (int q = 0; q < nq; ++q) {
          const float4 w = wq[(size_t)q * A];
          s = fmaf(f[4 * q], w.x, s); s = fmaf(f[4 * q + 1], w.y, s);
          s = fmaf(f[4 * q + 2], w.z, s); s = fmaf(f[4 * q + 3], w.w, s);
        }
        const float* wr = wh + (size_t)nq * A * 4 + a;
        for (int r = 0; r < rem; ++r) s = fmaf(f[4 * nq + r], wr[(size_t)r * A], s);
      }
    }
    logit[k] = s;
  }
  __syncthreads();
  /*@HST(2)*/
  /*@PST(10)*/
  // Linear(20,1) + tanh on the last wave; beside it the softmax terms exp(logit - max), one thread per action
  {
    const int vt = tid - (NTH - 64);
    if (vt >= 0 && vt < nb) {
      float s = b_v2[0];
      for (int u = 0; u < 20; ++u) s = fmaf(hid[vt * 20 + u], w_v2[u], s);
      values[omap[vt]] = tanhf(s);
    }
  }
  // the row maximum (exact in any order).  Small action counts: every thread scans its board's row.  Large ones (A > 32:
  // 225 reads per thread at 15x15): sixteen threads per board take a strided sixteenth each, then every thread reads
  // the sixteen partial maxima -- 30 reads instead of 225 for one more barrier.
  float* pmax = ebuf + 1024;  // [TB][16]   (TB * A <= 1024 and A > 32: at most 512 floats, up to HEAD_STAGE_AT)
  static_assert(768 + 20 * 32 + 256 * 4 + 128 + 1024 + 512 <= HEAD_STAGE_AT, "heads scratch layout");
  const bool wide = A > 32;
  if (wide) {
    for (int idx = tid; idx < nb * 16; idx += NTH) {
      const int bi = idx >> 4, j = idx & 15;
      const float* lg = logit + bi * A;
      float m = -3.4e38f;
      for (int a = j; a < A; a += 16) m = fmaxf(m, lg[a]);
      pmax[idx] = m;
    }
    __syncthreads();
  }
  for (int k = tid; k < nb * A; k += NTH) {
    const int bi = k / A;
    float mx = -3.4e38f;
    if (wide) {
      const float* pm = pmax + bi * 16;
#pragma unroll
      for (int j = 0; j < 16; ++j) mx = fmaxf(mx, pm[j]);
    } else {
      const float* lg = logit + bi * A;
      int a = 0;
      for (; a + 8 <= A; a += 8) {  // eight reads in flight
        const float t0 = lg[a], t1 = lg[a + 1], t2 = lg[a + 2], t3 = lg[a + 3], t4 = lg[a + 4], t5 = lg[a + 5], t6 = lg[a + 6], t7 = lg[a + 7];
        mx = fmaxf(fmaxf(fmaxf(fmaxf(mx, t0), fmaxf(t1, t2)), fmaxf(fmaxf(t3, t4), fmaxf(t5, t6))), t7);
      }
      for (; a < A; ++a) mx = fmaxf(mx, lg[a]);
    }
    ebuf[k] = expf(logit[k] - mx);
  }
  __syncthreads();
  if (wide) {
    // large action counts (225 adds on one thread: ~3 k cycles with nothing beside them): sixteen threads per board sum
    // a strided sixteenth each in action order, one thread adds the sixteen partial sums in order (pmax is free again)
    for (int idx = tid; idx < nb * 16; idx += NTH) {
      const int bi = idx >> 4, j = idx & 15;
      const float* eb = ebuf + bi * A;
      float part = 0.f;
      for (int a = j; a < A; a += 16) part += eb[a];
      pmax[idx] = part;
    }
    __syncthreads();
    if (tid < nb) {
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) sum += pmax[tid * 16 + j];
      stat[2 * tid + 1] = sum;
    }
  } else if (tid < nb) {  // the sum in action order, as a sequential softmax does
    float sum = 0.f;
    const float* eb = ebuf + tid * A;
    int a = 0;
    for (; a + 8 <= A; a += 8) {  // eight reads in flight, the adds in action order
      const float t0 = eb[a], t1 = eb[a + 1], t2 = eb[a + 2], t3 = eb[a + 3], t4 = eb[a + 4], t5 = eb[a + 5], t6 = eb[a + 6], t7 = eb[a + 7];
      sum += t0; sum += t1; sum += t2; sum += t3; sum += t4; sum += t5; sum += t6; sum += t7;
    }
    for (; a < A; ++a) sum += eb[a];
    stat[2 * tid + 1] = sum;
  }
  __syncthreads();
  /*@HST(3)*/
  /*@PST(11)*/
  for (int k = tid; k < nb * A; k += NTH) {
    const int bi = k / A;
    probs[(size_t)omap[bi] * A + (k - bi * A)] = ebuf[k] / stat[2 * bi + 1];
  }
}

// Where a workgroup's boards come from and go to.  Dense form (gpack == nullptr): board bi of the tile is row
// base + bi of planes / probs / values.  Slot form (the fused tree kernel, caro_engine.hip k_tree): game g keeps
// its j-th unique leaf at row g * B + j, gpack[g] = count | net class << 8, and the tile's boards are the
// leaves of class `cls` number [board0, board0 + nb) in GAME order -- every workgroup finds them itself with a
// prefix sum over gpack (G ints from L2), so which leaf meets which tile never depends on block arrival order.
// `sc` = 32 ints of LDS scratch, `smap` = nb ints of LDS; ends with a barrier.
template <int NTH = NT>
__device__ __forceinline__ void tile_rows(const int32_t* __restrict__ gpack, int G, int B, int cls, int base,
                                          int board0, int nb, int* sc, int* smap, int tid) {
  if (!gpack) {
    if (tid < nb) smap[tid] = base + tid;
    __syncthreads();
    return;
  }
  const int lane = tid & 63, wave = tid >> 6;
  const int cpt = (G + NTH - 1) / NTH;
  const int g_lo = min(G, tid * cpt), g_hi = min(G, g_lo + cpt);
  int mine = 0;
  for (int g = g_lo; g < g_hi; ++g) {
    const int v = gpack[g];
    mine += (v >> 8) == cls ? (v & 0xFF) : 0;
  }
  int inc = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  if (lane == 63) sc[wave] = inc;
  __syncthreads();
  int ex = inc - mine;
  for (int w = 0; w < wave; ++w) ex += sc[w];
  if (ex < board0 + nb && ex + mine > board0) {
    for (int g = g_lo; g < g_hi; ++g) {
      const int v = gpack[g];
      if ((v >> 8) != cls) continue;
      const int n = v & 0xFF;
      for (int j = 0; j < n; ++j) {
        const int r = ex + j - board0;
        if (r >= 0 && r < nb) smap[r] = g * B + j;
      }
      ex += n;
    }
  }
  __syncthreads();
}


// tile_rows with this thread's share of gpack already in registers (k_net_forward_w requests it first thing, beside
// the leaf count: as a load inside tile_rows -- behind the weight transfers this kernel has issued by then -- its wait
// was a wait for the two 32 KiB weight chunks as well, 4 k cycles that the dense form spends in conv_in).
// gv[u] = gpack[tid * cpt + u] (0 beyond G), mine = this thread's leaf count of class cls.
constexpr int GP_PRE = 4;  // games per thread held in registers (G <= GP_PRE * NT; more: tile_rows)
__device__ __forceinline__ void tile_rows_pre(const int (&gv)[GP_PRE], int cpt, int mine, int B, int cls, int board0,
                                              int nb, int* sc, int* smap, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  int inc = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  if (lane == 63) sc[wave] = inc;
  __syncthreads();
  int ex = inc - mine;
  for (int w = 0; w < wave; ++w) ex += sc[w];
  if (ex < board0 + nb && ex + mine > board0) {
#pragma unroll
    for (int u = 0; u < GP_PRE; ++u) {
      const int v = gv[u];
      if (u >= cpt || (v >> 8) != cls) continue;
      const int n = v & 0xFF, g = tid * cpt + u;
      for (int j = 0; j < n; ++j) {
        const int r = ex + j - board0;
        if (r >= 0 && r < nb) smap[r] = g * B + j;
      }
      ex += n;
    }
  }
  __syncthreads();
}

// `which` = 0 / 1: rows of that net only (p0 is used).  `which` = 2: both nets in ONE launch -- tiles
// [0, ceil(L0/TB)) run net 0 on rows [0, L0), the following tiles run net 1 (p1) on rows [L0, L0+L1).
__global__ __launch_bounds__(NT, 2) void k_net_forward(NetParams p0, NetParams p1, const float* __restrict__ planes,
                                                         const int32_t* __restrict__ counts, int which, int row1,
                                                         float* __restrict__ probs, float* __restrict__ values,
                                                         unsigned long long* __restrict__ stamps,
                                                         const int32_t* __restrict__ gpack, int gG, int gB) {
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
    row0 = second ? (row1 >= 0 ? row1 : L0) : 0;
    board0 = (second ? (int)blockIdx.x - t0 : (int)blockIdx.x) * p0.TB;
  }
  if (board0 >= L) return;
  const NetParams p = second ? p1 : p0;
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
  int* smap = reinterpret_cast<int*>(wbuf + 1536);  // [TB] plane / output row of every board of this tile
  tile_rows(gpack, gG, gB, second ? 1 : 0, row0 + board0, board0, nb, smap + 64, smap, tid);  // ends with a barrier

  conv_in_f32(p, planes, smap, act, wbuf, R, tid);
  const int slot_v = tid < nb ? smap[tid] : 0;
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
  heads_f32<false>(p, act, wbuf, probs, values, slot_v, nb, R, tid);
  if (stamps && tid == 0) {
    stamps[4 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t_c0;
    stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - t_r0;
    stamps[4 * blockIdx.x + 2] = t_trunk0 - t_c0;
    stamps[4 * blockIdx.x + 3] = t_trunk1 - t_c0;
  }
}

// ---- global -> LDS transfers that bypass the registers (used by the bf16x3 kernel below and by the Winograd kernels)
__device__ __forceinline__ unsigned lds_addr(const void* q) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)q;
}
// 16 bytes per lane global -> LDS, not tracked by the compiler (it would wait for vmcnt(0) in front of every later
// ds_read): lane l of the wave writes lds_wave_base + 16 l.  The issuer waits (vmcnt(0)) before the barrier that
// publishes the data.  M0 is a reserved register: the compiler loads it right in front of each of its own uses and keeps
// nothing alive in it, so it is not on the clobber list (hipcc warns if it is).
__device__ __forceinline__ void dma_b128(const void* gsrc, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_wave_base), "v"(gsrc) : "memory");
}
// The same transfer addressed as UNIFORM base (a scalar register pair) + 32-bit lane offset: the lane holds one
// register (16 tid) for the whole kernel instead of a 64-bit pointer per thread and the per-chunk address arithmetic
// becomes scalar -- what k_net_forward_w2, which runs at its 256-register limit, uses (its pointer pairs were spilled).
__device__ __forceinline__ void dma_b128_s(const void* sbase, unsigned voff, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_wave_base), "v"(voff), "s"(sbase) : "memory");
}

// ===================================================================================================
// Split-operand form ("bf16x3"): an EXTRA arithmetic mode, never the default and never the bench's headline.  Every
// float32 operand of the residual trunk is written as the sum of three bfloat16 parts (hi + mid + lo = the float32 value,
// 8 + 8 + 8 significant bits) and a product a*b is taken as the six part products of weight 2^-16 and above
// (ah bh, ah bm, am bh, ah bl, al bh, am bm) on v_mfma_f32_16x16x32_bf16 with float32 accumulation: per wave and tap
// 96 instructions of 16 cycles where the float32 direct form issues 64 of 64 cycles.  The dropped products are below
// 2^-24 of |a b|: the result is not bit-identical to the float32 kernels but in their own error class
// (tests/test_gpu_net.py states the gates).  conv_in, the biases, the residual adds, LeakyReLU and the heads stay float32
// code exactly as in the float32 kernels.
//   Why the 16x16x32 shape: a full launch of this kernel is POWER-limited -- with v_mfma_f32_32x32x16_bf16 the chip held
//   1.8-1.95 GHz and every cycle saved came back as a lower clock; under 16x16x32 it holds 2.2-2.3 GHz at the same flops
//   and LDS traffic (same box, same launch: 102 -> 92 us; NOTES.md, profiles/r06_x3_*).
//   LDS: [0, 64 KB) float32 activations at both ends of the trunk; in between the ring of staged weight halves (60 KB),
//        the five layers' biases and every thread's output row;
//        [64 KB, 160 KB) the split activations [3 parts][8 granules of 8 channels][256 rows][8 bf16].
//   The weights arrive pre-split from the host: per (layer, tap) [2 c][3 parts][4 kg][64 co][8 ci] bf16 (24 576 B),
//   ci = 32 c + 8 kg + 0..7.  The MFMA takes the WEIGHTS as its first operand: a lane then owns 4 x 4 consecutive output
//   channels of TWO rows, keeps their residual stream in float32 registers across the layers, and the epilogue writes
//   the split image of the new activations in 8-byte pieces of those rows.
//   -DCARO_X3_TIMERS=1: a diagnostic build whose stamped launches (caro_net_forward_stamped) also report the cycles a
//   wave spends in each phase of a tap (tools/probe_clock.py); no stamp executes in the product build's launches.
#ifndef CARO_X3_TIMERS
#define CARO_X3_TIMERS 0
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int X3_H = 3 * 4 * 64;        // uint4 (8 bf16) per half tap = 32 input channels: [3 parts][4 kg][64 co] = 768
constexpr int X3_TAP_U4 = 2 * X3_H;     // per tap image: 1536 (24 576 B)
constexpr int X3_PAD_TAPS = 2;          // zero taps behind the image: the staging loads run two taps ahead, unconditionally
constexpr int X3_RING_H1 = 3 * X3_H;    // ring: three slots of half 0, then two slots of half 1 (61 440 B)

// two float32 <-> two bfloat16 in one register (v_cvt_pk_bf16_f32, round to nearest even)
__device__ __forceinline__ uint32_t pk_bf16(f32x2 v) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2)); }
__device__ __forceinline__ f32x2 unpk_bf16(uint32_t w) {
  f32x2 r;
  r.x = __builtin_bit_cast(float, w << 16);
  r.y = __builtin_bit_cast(float, w & 0xFFFF0000u);
  return r;
}
__device__ __forceinline__ void split3(f32x2 v, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  hi = pk_bf16(v);
  const f32x2 r1 = v - unpk_bf16(hi);  // exact
  mid = pk_bf16(r1);
  const f32x2 r2 = r1 - unpk_bf16(mid);  // exact
  lo = pk_bf16(r2);
}
__device__ __forceinline__ f32x2 join3(uint32_t hi, uint32_t mid, uint32_t lo) {
  return (unpk_bf16(hi) + unpk_bf16(mid)) + unpk_bf16(lo);
}

__global__ __launch_bounds__(NT, 2) void k_net_forward_x3(NetParams p0, NetParams p1, const float* __restrict__ planes,
                                                            const int32_t* __restrict__ counts, int which, int row1,
                                                            float* __restrict__ probs, float* __restrict__ values,
                                                            unsigned long long* __restrict__ stamps,
                                                            const int32_t* __restrict__ gpack, int gG, int gB,
                                                            float* __restrict__ featbuf, int32_t* __restrict__ rowlist) {
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
    row0 = second ? (row1 >= 0 ? row1 : L0) : 0;
    board0 = (second ? (int)blockIdx.x - t0 : (int)blockIdx.x) * p0.TB;
  }
  if (board0 >= L) return;
  const NetParams p = second ? p1 : p0;
  const float slope = p.slope;
  unsigned long long t_c0 = 0, t_r0 = 0, t_epi = 0;  // diagnostic only (stamps == nullptr in every product launch)
  if (stamps) {
    t_c0 = __builtin_amdgcn_s_memtime();
    t_r0 = __builtin_amdgcn_s_memrealtime();
  }
  const int nb = min(p.TB, L - board0);
  const int HW = p.HW;
  const int R = nb * HW;  // real rows
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;

  for (int k = tid; k < ACT / 4; k += NT) reinterpret_cast<float4*>(lds)[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = tid; k < 9 * 2 * NF; k += NT) wbuf[k] = p.w_in[k];
  int* smap = reinterpret_cast<int*>(wbuf + 1536);
  tile_rows(gpack, gG, gB, second ? 1 : 0, row0 + board0, board0, nb, smap + 64, smap, tid);  // ends with a barrier
  conv_in_mfma(p, planes, smap, act, wbuf, R, tid);
  const int slot_v0 = tid < nb ? smap[tid] : 0;
  __syncthreads();

  unsigned long long t_trunk0 = 0;
  if (stamps) t_trunk0 = __builtin_amdgcn_s_memtime();
  // MFMA geometry (v_mfma_f32_16x16x32_bf16, the weights as first operand): lane (r16, kg) feeds input channels
  // 32 c + 8 kg + 0..7 of output channel 16 cb + r16 (weights) and of the neighbours of rows wave * 32 + 16 rb + r16
  // (activations), and receives output channels 16 cb + 4 kg + 0..3 of those two rows.
  const int r16 = lane & 15, kg = lane >> 4;
  const int myrow0 = wave * 32 + r16, myrow1 = myrow0 + 16;
  const bool rvalid0 = myrow0 < R, rvalid1 = myrow1 < R;
  // The residual stream of this lane's 32 outputs stays in float32 registers across the layers: the epilogue adds to it
  // and writes its split image for the next layer's MFMAs.
  f32x2 res[2][4][2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const int row = rb ? myrow1 : myrow0;
      const float4 v = *reinterpret_cast<const float4*>(act + row * NF + (((4 * cb + kg) ^ (row & 15)) << 2));
      res[rb][cb][0] = f32x2{v.x, v.y};
      res[rb][cb][1] = f32x2{v.z, v.w};
    }
  // ---- float32 activations -> three bf16 part planes (rows >= R are zero in `act`, so they are zero here)
  uint4* parts = reinterpret_cast<uint4*>(wbuf);  // [part * 8 + g][256 rows]
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int k = tid + NT * m, row = k & 255, g = k >> 8;
    const float4 lo4 = *reinterpret_cast<const float4*>(act + row * NF + (((2 * g) ^ (row & 15)) << 2));
    const float4 hi4 = *reinterpret_cast<const float4*>(act + row * NF + (((2 * g + 1) ^ (row & 15)) << 2));
    const f32x2 v[4] = {{lo4.x, lo4.y}, {lo4.z, lo4.w}, {hi4.x, hi4.y}, {hi4.z, hi4.w}};
    uint32_t ph[4], pm[4], pl[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split3(v[e], ph[e], pm[e], pl[e]);
    parts[(0 * 8 + g) * 256 + row] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
    parts[(1 * 8 + g) * 256 + row] = make_uint4(pm[0], pm[1], pm[2], pm[3]);
    parts[(2 * 8 + g) * 256 + row] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
  }
  __syncthreads();  // every thread is done with `act`: it becomes the weight ring
  // The ring.  Half 0 of a tap (its first 32 input channels) has three slots (tap % 3), half 1 two (tap & 1): while tap t
  // runs, half 1 of tap t+1 and half 0 of tap t+2 are staged, so that the first operand sets of tap t+1 can be requested
  // BEFORE the barrier that ends tap t (its half 0 was complete one barrier earlier).
  uint4* ring = reinterpret_cast<uint4*>(act);
  const uint4* wsrc = reinterpret_cast<const uint4*>(p.wx3);
  // what this thread stages per tap (global -> LDS directly, 16 bytes per lane and transfer): item tid and (waves 0-3)
  // item 512 + tid of the 768 of half 1 of the next tap; (waves 4-7) item tid - 256 and (all) item 256 + tid of the 768 of
  // half 0 of the tap after it
  const bool stage_h0 = tid >= 256;
  const unsigned ring_lds = lds_addr(ring);
  const unsigned v0off = (unsigned)tid * 16u;  // byte offset of item `tid` in an image
  const unsigned v1off = (unsigned)(stage_h0 ? X3_TAP_U4 + (tid - 256) : X3_H + 512 + tid) * 16u;  // relative to the next tap's image
  {
    ring[0 * X3_H + tid] = wsrc[tid];                                       // tap 0, half 0 (slot 0)
    if (tid < 256) ring[0 * X3_H + 512 + tid] = wsrc[512 + tid];
    ring[1 * X3_H + tid] = wsrc[X3_TAP_U4 + tid];                           // tap 1, half 0 (slot 1)
    if (tid < 256) ring[1 * X3_H + 512 + tid] = wsrc[X3_TAP_U4 + 512 + tid];
    ring[X3_RING_H1 + tid] = wsrc[X3_H + tid];                              // tap 0, half 1 (slot 0)
    if (tid < 256) ring[X3_RING_H1 + 512 + tid] = wsrc[X3_H + 512 + tid];
  }
  // the 4 KB behind the ring: the five layers' biases (read by the epilogues) and every thread's output row (read by the
  // heads) -- values that would otherwise occupy registers across the whole trunk
  static_assert((X3_RING_H1 + 2 * X3_H) * 16 + NRES * NF * 4 + NT * 4 <= ACT * 4, "ring + biases + output rows must fit the 64 KB of the float32 activations");
  float* bias_l = reinterpret_cast<float*>(ring + X3_RING_H1 + 2 * X3_H);  // [5][64]
  int* slot_l = reinterpret_cast<int*>(bias_l + NRES * NF);                 // [NT]
  if (tid < NRES * NF) bias_l[tid] = p.b_res[tid];
  slot_l[tid] = slot_v0;
  __syncthreads();

  int nrow9[9];  // the neighbour rows of the lane's two rows per tap (ZROW outside the board): row 0 | row 1 << 16
#pragma unroll
  for (int t = 0; t < 9; ++t) nrow9[t] = 0;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int row = rb ? myrow1 : myrow0;
    const bool rv = rb ? rvalid1 : rvalid0;
    const int rbi = row / HW;
    const int rcell = row - rbi * HW;
    const int ry = rcell / p.W, rx = rcell - ry * p.W;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int ny = ry + t / 3 - 1, nx = rx + t % 3 - 1;
      const bool ok = rv && ny >= 0 && ny < p.H && nx >= 0 && nx < p.W;
      nrow9[t] |= (ok ? rbi * HW + ny * p.W + nx : (int)ZROW) << (16 * rb);
    }
  }

  f32x4v acc[2][4];  // [rb][cb]
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[rb][cb] = f32x4v{0.f, 0.f, 0.f, 0.f};
  // operand sets of six 16-byte fragments, index part * 2 + (rb | cb & 1):
  //   activations of 32 input channels c: the three parts of the neighbour rows of the lane's two rows
  //   weights of 32 input channels c and a pair of channel blocks cp: the three parts of blocks 2 cp, 2 cp + 1
  const uint4* arow = parts + kg * 256;  // + (part * 8 + 4 c) * 256 + neighbour row
#define CARO_X3_LOAD_ACT(S_, C_, N0_, N1_)                                  \
  _Pragma("unroll") for (int pt_ = 0; pt_ < 3; ++pt_) {                     \
    S_[pt_ * 2] = arow[(pt_ * 8 + 4 * (C_)) * 256 + (N0_)];                 \
    S_[pt_ * 2 + 1] = arow[(pt_ * 8 + 4 * (C_)) * 256 + (N1_)];             \
  }
#define CARO_X3_LOAD_W(S_, HB_, CP_)                                        \
  _Pragma("unroll") for (int pt_ = 0; pt_ < 3; ++pt_) {                     \
    S_[pt_ * 2] = (HB_)[pt_ * 256 + (CP_) * 32];                            \
    S_[pt_ * 2 + 1] = (HB_)[pt_ * 256 + (CP_) * 32 + 16];                   \
  }
  // the six part products of weight 2^-16 and above, smallest first (the float32 accumulator takes the low-order
  // corrections before the leading term); the four accumulators of a segment take turns
#define CARO_X3_PROD(W_, A_, CP_, WP_, AP_)                                                                     \
  _Pragma("unroll") for (int cbi_ = 0; cbi_ < 2; ++cbi_) _Pragma("unroll") for (int rb_ = 0; rb_ < 2; ++rb_)   \
    acc[rb_][2 * (CP_) + cbi_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                       \
        __builtin_bit_cast(bf16x8, W_[(WP_) * 2 + cbi_]), __builtin_bit_cast(bf16x8, A_[(AP_) * 2 + rb_]),      \
        acc[rb_][2 * (CP_) + cbi_], 0, 0, 0);
#define CARO_X3_SEG(W_, A_, CP_)  \
  CARO_X3_PROD(W_, A_, CP_, 1, 1) \
  CARO_X3_PROD(W_, A_, CP_, 2, 0) \
  CARO_X3_PROD(W_, A_, CP_, 0, 2) \
  CARO_X3_PROD(W_, A_, CP_, 1, 0) \
  CARO_X3_PROD(W_, A_, CP_, 0, 1) \
  CARO_X3_PROD(W_, A_, CP_, 0, 0)
  // A segment = 24 MFMAs of 16 cycles with the operand reads of a LATER segment woven into their gaps, one LDS instruction
  // behind every second MFMA (every MFMA in the tap's last segment): issued in the shadow of an MFMA an LDS instruction
  // costs the wave nothing; issued as a burst between the segments it holds the wave -- and with it the matrix pipe,
  // which its partner on the SIMD is not using either, the two being in the same phase after every barrier.
#define CARO_X3_WEAVE(NR_)                                  \
  _Pragma("unroll") for (int q_ = 0; q_ < (NR_); ++q_) {    \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      \
  }                                                         \
  __builtin_amdgcn_sched_group_barrier(0x008, 24 - 2 * (NR_), 0);
#if CARO_X3_TIMERS  /* diagnostic build (tools/probe_clock.py): per-phase cycles of a wave, summed over the 45 taps */
  unsigned long long tq[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl = 0;
#define CARO_T0 tl = __builtin_amdgcn_s_memtime();
#define CARO_T(K_) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tq[K_] += t_ - tl; tl = t_; }
#else
#define CARO_T0
#define CARO_T(K_)
#endif
  uint4 X[6], Y[6], P[6], Q[6];
  const uint4* wcol = ring + kg * 64 + r16;  // + slot + part * 256 + 16 cb
  CARO_X3_LOAD_ACT(X, 0, nrow9[0] & 0xFFFF, nrow9[0] >> 16)
  CARO_X3_LOAD_W(P, wcol, 0)
  const char* wnext = reinterpret_cast<const char*>(wsrc + X3_TAP_U4);  // image of the tap after the current one
  for (int layer = 0; layer < NRES; ++layer) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      // 9 taps per layer: the half-0 slot of a tap is tap % 3 in every layer, the slot of its half 1 alternates
      const int cur = (layer + tap) & 1;
      const int slot0 = tap % 3, slot1 = (tap + 1) % 3, slot2 = (tap + 2) % 3;
      CARO_T0
      {  // The staged halves, on their way while the tap computes (the image is followed by X3_PAD_TAPS zero taps: no
         // bounds to check).  Their slots were last read one tap ago, before the barrier that ended it; the transfers
         // are waited for in front of this tap's closing barrier.
        const unsigned d1 = ring_lds + (unsigned)(X3_RING_H1 + (cur ^ 1) * X3_H) * 16u;
        const unsigned d0 = ring_lds + (unsigned)(slot2 * X3_H) * 16u;
        const unsigned w1k = (unsigned)wave * 1024u;
        dma_b128_s(wnext + X3_H * 16, v0off, __builtin_amdgcn_readfirstlane(d1 + w1k));
        dma_b128_s(wnext, v1off, __builtin_amdgcn_readfirstlane(stage_h0 ? d0 + w1k - 4096u : d1 + 8192u + w1k));
        dma_b128_s(wnext + (X3_TAP_U4 + 256) * 16, v0off, __builtin_amdgcn_readfirstlane(d0 + 4096u + w1k));
        wnext += X3_TAP_U4 * 16;
      }
      const uint4* hb0 = wcol + slot0 * X3_H;
      const uint4* hb1 = wcol + X3_RING_H1 + cur * X3_H;
      const int n0 = nrow9[tap] & 0xFFFF, n1 = nrow9[tap] >> 16;
      __builtin_amdgcn_sched_barrier(0);
      CARO_T(0)
      CARO_X3_LOAD_W(Q, hb0, 1)
      CARO_X3_SEG(P, X, 0)
      CARO_X3_WEAVE(6)
      __builtin_amdgcn_sched_barrier(0);
      CARO_T(1)
      CARO_X3_LOAD_ACT(Y, 1, n0, n1)
      CARO_X3_LOAD_W(P, hb1, 0)
      CARO_X3_SEG(Q, X, 1)
      CARO_X3_WEAVE(12)
      __builtin_amdgcn_sched_barrier(0);
      CARO_T(2)
      CARO_X3_LOAD_W(Q, hb1, 1)
      CARO_X3_SEG(P, Y, 0)
      CARO_X3_WEAVE(6)
      __builtin_amdgcn_sched_barrier(0);
      CARO_T(3)
      // the next tap's first operand sets -- at a layer's last tap the activations are read again behind the epilogue
      CARO_X3_LOAD_ACT(X, 0, nrow9[tap == 8 ? 0 : tap + 1] & 0xFFFF, nrow9[tap == 8 ? 0 : tap + 1] >> 16)
      CARO_X3_LOAD_W(P, wcol + slot1 * X3_H, 0)
      CARO_X3_SEG(Q, Y, 1)
      CARO_X3_WEAVE(12)
      __builtin_amdgcn_sched_barrier(0);
      CARO_T(4)
      if (tap == 8) {
        unsigned long long t_e0 = 0;
        if (stamps) t_e0 = __builtin_amdgcn_s_memtime();
        __syncthreads();  // every wave has read this layer's input activations: they may be overwritten
        // epilogue, in place: v = v + leaky(conv(v) + b)  (lib/model.py:85-89).  acc[rb][cb][e] = output channel
        // 16 cb + 4 kg + e of row wave * 32 + 16 rb + r16: four consecutive channels per 8-byte piece of a part plane.
        const f32x2 slope2 = {slope, slope};
        float4 bq[4];  // the layer's biases of this lane's channels (16 cb + 4 kg + 0..3)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) bq[cb] = *reinterpret_cast<const float4*>(bias_l + layer * NF + 16 * cb + 4 * kg);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const int row = rb ? myrow1 : myrow0;
          const bool rv = rb ? rvalid1 : rvalid0;
          const f32x2 keep = rv ? f32x2{1.f, 1.f} : f32x2{0.f, 0.f};  // rows this tile does not have stay zero
          uint2* mine = reinterpret_cast<uint2*>(parts + (kg >> 1) * 256 + row) + (kg & 1);  // + (part * 8 + 2 cb) * 512
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) {
            const f32x4v a = acc[rb][cb];
            const f32x2 t01 = f32x2{a[0], a[1]} + f32x2{bq[cb].x, bq[cb].y}, t23 = f32x2{a[2], a[3]} + f32x2{bq[cb].z, bq[cb].w};
            const f32x2 s01 = t01 * slope2, s23 = t23 * slope2;
            const f32x2 l01 = {t01.x > 0.f ? t01.x : s01.x, t01.y > 0.f ? t01.y : s01.y};
            const f32x2 l23 = {t23.x > 0.f ? t23.x : s23.x, t23.y > 0.f ? t23.y : s23.y};
            res[rb][cb][0] = (res[rb][cb][0] + l01) * keep;
            res[rb][cb][1] = (res[rb][cb][1] + l23) * keep;
            uint2 wh, wm, wl;
            split3(res[rb][cb][0], wh.x, wm.x, wl.x);
            split3(res[rb][cb][1], wh.y, wm.y, wl.y);
            mine[(0 * 8 + 2 * cb) * 512] = wh;
            mine[(1 * 8 + 2 * cb) * 512] = wm;
            mine[(2 * 8 + 2 * cb) * 512] = wl;
            acc[rb][cb] = f32x4v{0.f, 0.f, 0.f, 0.f};
          }
        }
        if (stamps) t_epi += __builtin_amdgcn_s_memtime() - t_e0;
      }
      CARO_T0
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's transfers have landed
      __syncthreads();  // the staged halves / the new activations are visible to every wave
      CARO_T(5)
      if (tap == 8 && layer + 1 < NRES) {  // the next layer's first activation set reads the rows just written
        CARO_X3_LOAD_ACT(X, 0, nrow9[0] & 0xFFFF, nrow9[0] >> 16)
      }
    }
  }
#undef CARO_X3_WEAVE
#undef CARO_X3_LOAD_ACT
#undef CARO_X3_LOAD_W
#undef CARO_X3_PROD
#undef CARO_X3_SEG
  // ---- the trunk output back to float32 in `act` (the ring is dead: the loop ended with a barrier), then the float32
  // heads with the parts' region as scratch
  const int slot_v = slot_l[tid];
  __syncthreads();  // (slot_l sits where the last rows of `act` go)
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const int row = rb ? myrow1 : myrow0;
      *reinterpret_cast<float4*>(act + row * NF + (((4 * cb + kg) ^ (row & 15)) << 2)) =
          make_float4(res[rb][cb][0].x, res[rb][cb][0].y, res[rb][cb][1].x, res[rb][cb][1].y);
    }
  __syncthreads();
  unsigned long long t_trunk1 = 0;
  if (stamps) t_trunk1 = __builtin_amdgcn_s_memtime();
  // featbuf != null (one board per workgroup): only the 1x1 convolutions here, the FC heads of the whole launch follow in
  // k_net_heads (row0 + board0 = this board's dense index)
  heads_f32<false>(p, act, wbuf, probs, values, slot_v, nb, R, tid, featbuf, rowlist, row0 + board0);
  if (stamps && tid == 0) {
    stamps[4 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t_c0;
    stamps[4 * blockIdx.x + 1] = ((__builtin_amdgcn_s_memrealtime() - t_r0) & 0xFFFFFull) | (t_epi << 20);  // + the five epilogues
    stamps[4 * blockIdx.x + 2] = t_trunk0 - t_c0;
    stamps[4 * blockIdx.x + 3] = t_trunk1 - t_c0;
#if CARO_X3_TIMERS
    for (int q = 0; q < 8; ++q) stamps[4 * 512 + 8 * blockIdx.x + q] = tq[q];
#endif
  }
}

// ===================================================================================================
// Winograd form F(2,3) along the board rows ("f32w"): the same float32 network function with one third fewer
// MFMAs.  A 3x3 convolution is 3 column taps dx of a 3-tap convolution along y; for a pair of output rows
// (2ty, 2ty+1) the latter needs 4 products per (dx, ci) instead of 6:
//     d0..d3 = input rows 2ty-1 .. 2ty+2 (zero outside the board), g0..g2 = the kernel column
//     m0 = (d0-d2) g0, m1 = (d1+d2)(g0+g1+g2)/2, m2 = (d2-d1)(g0-g1+g2)/2, m3 = (d1-d3) g2
//     y(2ty) = m0+m1+m2,  y(2ty+1) = m1-m2-m3
// GEMM rows are TILES t = board*TPB + ty*W + x (TPB = ceil(H/2)*W; 21 for connect four, so 6 boards = 126
// of 128 rows), K = 4 transformed taps p x 3 dx x 64 channels, U[p][dx] = sum_ky G[p][ky] w[ky][dx] is
// packed by the host (float64 sum, one rounding).  The transformed input d_a +- d_b is formed in registers
// from two LDS reads when the A operand is loaded, so nothing but the activations themselves lives in LDS.
// Wave w owns row tile w>>1 x col tile w&1 (32 tiles x 32 channels) and keeps FOUR accumulators, one per
// transformed tap p.  The stream is ordered so that all four p of one (dx, channel granule) are processed
// together: the four input rows d0..d3 of a tile are read once (4 ds_read_b128) and give V0 = d0-d2, V1 = d1+d2,
// V2 = d2-d1, V3 = d1-d3; with the four weight granules (4 ds_read_b128) that is 16 MFMAs for 8 LDS reads.
// Measured (tools/micro/mfma_shadow.hip, mfma_pfused.hip): on this stream a wave-wide ds_read_b128 costs the SIMD
// ~13 cycles of matrix-pipe time on top of the 64 of an MFMA, whatever the prefetch distance, barriers or
// priorities; one p at a time (2 rows + 1 weight granule per 4 MFMAs, 0.75 reads per MFMA) ran at 76 cycles
// per MFMA, this order (0.5 reads per MFMA) at 68-70.  The sequence of MFMAs into each accumulator -- dx, then
// channel granule, then k -- and the output transform are those of the one-p-at-a-time form, so the results
// are bit-identical to it.
// Weight chunk = one dx, four channel granules of each lane half, all four p: [p][h][co][16 floats] = 32 KiB,
// 6 chunks per layer, a ring of three LDS buffers.
constexpr int WTAPS = NRES * 12;          // 60 transformed taps of 4096 floats
constexpr int WCH = 4 * 2 * 64 * 16;      // floats per chunk (8192)
constexpr int WNCHUNK = NRES * 6;         // 30
constexpr int WNBUF = 3;
constexpr int XROW4 = 63;                 // ... of the 4-way tiles' (48 KiB: rows 63..254)
constexpr int XROW2 = 127;                // first row of the 2-way tiles' exchange area (32 KiB: rows 127..254)
static_assert(WNBUF * WCH == 2 * TPC * WCHUNK, "the ring takes the place of the two three-tap buffers");

typedef float f4v __attribute__((ext_vector_type(4)));

// chunk c of the transformed weights -> ring buffer c % WNBUF (every thread moves 4 x 16 bytes)
__device__ __forceinline__ void fetch_chunk(const float* ww, int c, unsigned wring, int tid) {
  const float4* src = reinterpret_cast<const float4*>(ww + (size_t)c * WCH) + tid;
  const unsigned dst = __builtin_amdgcn_readfirstlane(wring + (unsigned)(c % WNBUF) * (WCH * 4) + (unsigned)(tid >> 6) * 1024u);
#pragma unroll
  for (int m = 0; m < WCH / 4 / NT; ++m) dma_b128(src + m * NT, dst + m * NT * 16);
}

__device__ __forceinline__ void fetch_chunk_s(const float* ww, int c, unsigned wring, int tid) {
  const char* base = reinterpret_cast<const char*>(ww) + (size_t)c * (WCH * 4);
  const unsigned voff = (unsigned)tid * 16u;
  const unsigned dst = __builtin_amdgcn_readfirstlane(wring + (unsigned)(c % WNBUF) * (WCH * 4) + (unsigned)(tid >> 6) * 1024u);
#pragma unroll
  for (int m = 0; m < WCH / 4 / NT; ++m) dma_b128_s(base + m * NT * 16, voff, dst + m * NT * 16);
}

// the head parameters (heads_f32<true>) -> ring buffer 0 + HEAD_STAGE_AT, issued when that buffer has seen its last
// chunk; over-reads up to 8 KiB past the block (the packed buffer is padded by a whole tap chunk)
__device__ __forceinline__ void fetch_heads(const float* hp, int hspan, unsigned wring, int tid) {
  const float4* src = reinterpret_cast<const float4*>(hp) + tid;
  const unsigned dst = __builtin_amdgcn_readfirstlane(wring + HEAD_STAGE_AT * 4u + (unsigned)(tid >> 6) * 1024u);
  dma_b128(src, dst);
  if (hspan > NT * 4) dma_b128(src + NT, dst + NT * 16);
}

// Comments of the form /*@NAME(...)*/ are the probe points of the timing experiments: tools/exp/build_exp.py turns them
// into macro calls in a COPY of this file (tools/exp/caro_net_exp.h); here they are comments and nothing else.
// (Measured that way at 1434 leaves: trunk 296 k cycles; 293 k / 289 k / 281 k without chunk barriers / weight fetches /
// both: the fetches cost what their 960 KiB per workgroup take of the LDS write port.)
template <int KS>
__device__ __forceinline__ void trunk_w(const NetParams& p, float* act, float* wbuf, int nb, int tid, int hspan) {
  const float slope = p.slope;
  const int HW = p.HW;
  const int wave = tid >> 6, lane = tid & 63;
  const int i = lane & 31, h = lane >> 5;
  // KS = 1: wave = row tile (wave >> 1) x col tile (wave & 1), all of K.  KS = 2 / 4 (small launches and the
  // second round of a launch that overflows one round, see k_net_forward_w): 4 / KS row tiles, and KS waves share
  // one (row tile, col tile), wave kq taking the channel granules G with G % KS == kq; their partial sums meet in
  // LDS before the in-place epilogue.
  constexpr int NRT = 4 / KS;        // row tiles
  constexpr int NQ = 4 / KS;         // operand sets per chunk and wave
  constexpr int NSET = 6 * NQ;       // operand sets per layer and wave
  const int ct = wave & 1, rt = (wave >> 1) % NRT, kq = (wave >> 1) / NRT;
  // this lane's tile.  KS = 1: which tile sits on which MFMA row is a host-built table: a ds_read_b128 is served
  // in 16-lane groups, a group is conflict-free when its 16 activation rows differ mod 16 (the swizzle key), and
  // the table picks the tiles of each group accordingly (caro_net_enable_winograd).  KS > 1: tiles in order.
  int tbi, tty, tx;
  bool tvalid;
  if (KS == 1) {
    const uint32_t tent = p.wtab[rt * 32 + i];
    tbi = tent & 0xFF; tty = (tent >> 8) & 0xFF; tx = (tent >> 16) & 0xFF;
    tvalid = (tent >> 24) != 0 && tbi < nb;
  } else {
    const int tpb = ((p.H + 1) >> 1) * p.W;
    const int mt = rt * 32 + i;
    tbi = mt / tpb;
    const int rem = mt - tbi * tpb;
    tty = rem / p.W; tx = rem - tty * p.W;
    tvalid = mt < nb * tpb;
  }
  // output side: with the WEIGHTS as first MFMA operand a lane's 16 accumulator registers are 4 groups of 4
  // consecutive channels (ct*32 + 8q + 4h + 0..3) of ITS tile, so the epilogue moves float4s
  const int orow0 = tbi * HW + 2 * tty * p.W + tx;  // activation row of output row 2ty
  const bool ovalid0 = tvalid, ovalid1 = tvalid && 2 * tty + 1 < p.H;
  // LDS byte address of (input row r of the tile = board row 2ty-1+r, column tx+dx-1, granule 8h + kq) with the
  // row's swizzle key folded in; the granule G of a set is XORed in afterwards ((8h + G) ^ key == ((8h) ^ key) ^ G).
  // A cell outside the board reads the zero row.  All 16 granules of that row are zero, so the lane may take ANY
  // of them: it takes the one its own (virtual) row would have used, which keeps the 16 lanes of a ds_read_b128
  // group on 16 different banks (they all have different row residues by construction of the tile table).
  const unsigned abase = lds_addr(act);
  unsigned ra[4][3];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int y = 2 * tty - 1 + r;
    const bool oky = tvalid && y >= 0 && y < p.H;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int nx = tx + d - 1;
      const int v = tbi * HW + y * p.W + nx;
      const int row = oky && nx >= 0 && nx < p.W ? v : ZROW;
      ra[r][d] = abase + (unsigned)(row * NF + (((h * 8 + kq) ^ (v & 15)) << 2)) * 4u;
    }
  }
  // this lane's weight row in ring buffer 0: [p = 0][h][co = ct*32 + i], granule swizzle (co >> 2) & 3 = (i >> 2) & 3
  // (the 16 lanes of a ds_read_b128 group hold every residue of i mod 4 four times, with four different (i >> 2) & 3)
  const unsigned wlane = lds_addr(wbuf) + (unsigned)((h * 64 + ct * 32 + i) * 16 + ((kq ^ ((i >> 2) & 3)) << 2)) * 4u;
  const unsigned wring = lds_addr(wbuf);

  f4v D0, D1, D2, D3;                  // the four input rows of the operand set in flight
  f4v W0, W1, W2, W3, X0, X1, X2, X3;  // its weight granules: two sets, the MFMAs read one while the other loads
// operand set T of the layer: chunk T / NQ (dx = chunk >> 1, granule half = chunk & 1), granule (T % NQ) * KS of the half
#define CARO_ALOAD(B0, B1, B2, B3, T)                                                                        \
  {                                                                                                          \
    constexpr int cc_ = (T) / NQ, dx_ = cc_ >> 1, g8_ = (cc_ & 1) * 4 + ((T) % NQ) * KS;                     \
    const unsigned wa_ = (wlane + (unsigned)(cc_ % WNBUF) * (WCH * 4)) ^ ((unsigned)(((T) % NQ) * KS) << 4); \
    asm volatile("ds_read_b128 %0, %1" : "=v"(D0) : "v"(ra[0][dx_] ^ (g8_ << 4)));                           \
    asm volatile("ds_read_b128 %0, %1" : "=v"(D1) : "v"(ra[1][dx_] ^ (g8_ << 4)));                           \
    asm volatile("ds_read_b128 %0, %1" : "=v"(D2) : "v"(ra[2][dx_] ^ (g8_ << 4)));                           \
    asm volatile("ds_read_b128 %0, %1" : "=v"(D3) : "v"(ra[3][dx_] ^ (g8_ << 4)));                           \
    asm volatile("ds_read_b128 %0, %1" : "=v"(B0) : "v"(wa_));                                               \
    asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(B1) : "v"(wa_));                                   \
    asm volatile("ds_read_b128 %0, %1 offset:16384" : "=v"(B2) : "v"(wa_));                                  \
    asm volatile("ds_read_b128 %0, %1 offset:24576" : "=v"(B3) : "v"(wa_));                                  \
  }
// every LDS read this wave has issued is complete (the compiler does not know of the reads above)
#define CARO_AWAIT(B0, B1, B2, B3)                                                                           \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(D0), "+v"(D1), "+v"(D2), "+v"(D3), "+v"(B0), "+v"(B1), "+v"(B2), "+v"(B3));
// One step of the layer's software pipeline.  Set T has been requested a whole burst ago: wait for it, form the four
// transformed operands (the row registers are free again), request set T+1 -- rows into the same registers, weights
// into the other set -- and issue the 16 MFMAs of set T on four independent accumulators.
// At the first set of a chunk (T % NQ == 0) sits the ONE workgroup barrier of the chunk:
//   every read this wave has issued is complete, its share of chunk c+1 has arrived (vmcnt(0), issued a whole chunk
//   ago) -> barrier -> chunk c+1 is visible to every wave and no wave reads chunk c-1 any more -> chunk c+2 is
//   fetched into that buffer.  (For a layer's first chunk the barrier is the one that ends the previous layer's
//   epilogue, or the kernel's barrier after conv_in.)
#define CARO_STEP(T, B0, B1, B2, B3, NB0, NB1, NB2, NB3)                                                     \
  CARO_AWAIT(B0, B1, B2, B3)                                                                                 \
  if ((T) % NQ == 0) {                                                                                       \
    if ((T) != 0) {                                                                                          \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                       \
      /*@CHUNK_BARRIER*/ __syncthreads();                                                                    \
    }                                                                                                        \
    if (/*@FETCH_ON*/ c0 + (T) / NQ + 2 < WNCHUNK) fetch_chunk(p.ww, c0 + (T) / NQ + 2, wring, tid);         \
    else if (c0 + (T) / NQ + 2 == WNCHUNK && hspan) fetch_heads(p.w_head, hspan, wring, tid);                \
  }                                                                                                          \
  {                                                                                                          \
    /* Issue order (round 3): each transformed operand is formed right in front of its first MFMA, and the requests   \
       of the next set go BEHIND the first round of MFMAs, where the matrix pipe is busy for 256 cycles anyway: the    \
       pipe starts 4 instructions after the wait instead of 28 (full tile 312 k -> 308 k cycles, 2-way 185 k -> 177 k,\
       4-way 122 k -> 113 k).  The chunk fetch stays in FRONT: moved behind the first MFMAs as well it cost 13 k. */    \
    const f4v v0 = D0 - D2;                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    accM0 = __builtin_amdgcn_mfma_f32_32x32x2f32(B0.x, v0.x, accM0, 0, 0, 0);                                \
    const f4v v1 = D1 + D2;                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    accM1 = __builtin_amdgcn_mfma_f32_32x32x2f32(B1.x, v1.x, accM1, 0, 0, 0);                                \
    const f4v v2 = D2 - D1;                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    accM2 = __builtin_amdgcn_mfma_f32_32x32x2f32(B2.x, v2.x, accM2, 0, 0, 0);                                \
    const f4v v3 = D1 - D3;                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    accM3 = __builtin_amdgcn_mfma_f32_32x32x2f32(B3.x, v3.x, accM3, 0, 0, 0);                                \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    if ((T) + 1 < NSET) CARO_ALOAD(NB0, NB1, NB2, NB3, ((T) + 1) % NSET)                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    accM0 = __builtin_amdgcn_mfma_f32_32x32x2f32(B0.y, v0.y, accM0, 0, 0, 0);                                \
    accM1 = __builtin_amdgcn_mfma_f32_32x32x2f32(B1.y, v1.y, accM1, 0, 0, 0);                                \
    accM2 = __builtin_amdgcn_mfma_f32_32x32x2f32(B2.y, v2.y, accM2, 0, 0, 0);                                \
    accM3 = __builtin_amdgcn_mfma_f32_32x32x2f32(B3.y, v3.y, accM3, 0, 0, 0);                                \
    accM0 = __builtin_amdgcn_mfma_f32_32x32x2f32(B0.z, v0.z, accM0, 0, 0, 0);                                \
    accM1 = __builtin_amdgcn_mfma_f32_32x32x2f32(B1.z, v1.z, accM1, 0, 0, 0);                                \
    accM2 = __builtin_amdgcn_mfma_f32_32x32x2f32(B2.z, v2.z, accM2, 0, 0, 0);                                \
    accM3 = __builtin_amdgcn_mfma_f32_32x32x2f32(B3.z, v3.z, accM3, 0, 0, 0);                                \
    accM0 = __builtin_amdgcn_mfma_f32_32x32x2f32(B0.w, v0.w, accM0, 0, 0, 0);                                \
    accM1 = __builtin_amdgcn_mfma_f32_32x32x2f32(B1.w, v1.w, accM1, 0, 0, 0);                                \
    accM2 = __builtin_amdgcn_mfma_f32_32x32x2f32(B2.w, v2.w, accM2, 0, 0, 0);                                \
    accM3 = __builtin_amdgcn_mfma_f32_32x32x2f32(B3.w, v3.w, accM3, 0, 0, 0);                                \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
  }

  float4 res0[4], res1[4];  // full tile: the lane's own output cells of the last layer (the next layer's residual input)
#pragma unroll
  for (int q = 0; q < 4; ++q) res0[q] = res1[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  /*@SHIFT_BEGIN(KS, wave)*/
  for (int layer = 0; layer < NRES; ++layer) {
    const int c0 = layer * 6;  // first chunk of the layer; 6 % WNBUF == 0, so chunk c0 + k sits in buffer k % WNBUF
    /*@LST(layer, 0)*/
    f32x16 accM0, accM1, accM2, accM3;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      accM0[e] = 0.f;
      accM1[e] = 0.f;
      accM2[e] = 0.f;
      accM3[e] = 0.f;
    }
    // the per-set addresses (ra ^ granule) are re-formed in every layer: hoisted out of the layer loop -- as the compiler
    // would -- they take a hundred registers and the kernel spills (4 MB of scratch writes per launch); neither keeping
    // part of them, nor packed adds for the transforms, nor a table of weight addresses changed the trunk's time
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int d = 0; d < 3; ++d) asm volatile("" : "+v"(ra[r][d]));
    CARO_ALOAD(W0, W1, W2, W3, 0)
#define CARO_PAIR(T)                                                                                         \
  CARO_STEP(T, W0, W1, W2, W3, X0, X1, X2, X3)                                                               \
  CARO_STEP((T) + 1, X0, X1, X2, X3, W0, W1, W2, W3)
    CARO_PAIR(0) CARO_PAIR(2) CARO_PAIR(4)
    if constexpr (NSET > 6) { CARO_PAIR(6) CARO_PAIR(8) CARO_PAIR(10) }
    if constexpr (NSET > 12) { CARO_PAIR(12) CARO_PAIR(14) CARO_PAIR(16) CARO_PAIR(18) CARO_PAIR(20) CARO_PAIR(22) }
#undef CARO_PAIR
    /*@LST(layer, 1)*/
    if constexpr (KS == 2) {
      // 2-way K-split, balanced epilogue (round 3): the two waves of a tile (kq = 0 / 1: the halves of K) each FINISH one
      // of the tile's two output rows -- wave kq row 2ty + kq -- instead of wave 0 finishing both while wave 1 idles.
      // Each forms both rows' partial sums (the same fma chains as before, the coefficients picked by kq), keeps its own
      // row's and hands the other to its partner through LDS: 4 ds_write_b128 + 4 ds_read_b128 per wave instead of 32 +
      // 32 dword accesses on one wave each, and half the bias / LeakyReLU / write-back per wave.  The exchange area is
      // the part of the activation buffer a 2-way tile never touches (rows 127..254: TB2 * HW <= 127, checked at upload),
      // so it is written BEFORE the "inputs read" barrier, which then publishes it as well: two barriers per layer
      // instead of three.  Sums: partial(kq 0) + partial(kq 1) as before (the addition commutes): bit-identical.
      const bool up = kq != 0;  // wave-uniform
      const float m0 = up ? 0.f : 1.f, m2 = up ? -1.f : 1.f, m3 = up ? -1.f : 0.f;   // my row:      {1,1,1,0} / {0,1,-1,-1}
      const float s0 = up ? 1.f : 0.f, s2 = up ? 1.f : -1.f, s3 = up ? 0.f : -1.f;   // partner's row
      const float* bias = p.b_res + layer * NF + ct * 32 + 4 * h;
      const int g0 = ct * 8 + h;  // granule of group q is g0 + 2q
      float* rowp = act + (orow0 + (up ? p.W : 0)) * NF;
      const int kk = (orow0 + (up ? p.W : 0)) & 15;
      const bool ov = up ? ovalid1 : ovalid0;
      float4 bq[4], old[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const float4*>(bias + 8 * q);
      if (layer == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          old[q] = ov ? *reinterpret_cast<const float4*>(rowp + (((g0 + 2 * q) ^ kk) << 2)) : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {  // the cells this wave finished in the last layer (res0, as in the full tile)
#pragma unroll
        for (int q = 0; q < 4; ++q) old[q] = res0[q];
      }
      f32x16 mine, send;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float a = fmaf(m0, accM0[e], 0.f), b = fmaf(s0, accM0[e], 0.f);
        a = fmaf(1.f, accM1[e], a); b = fmaf(1.f, accM1[e], b);
        a = fmaf(m2, accM2[e], a); b = fmaf(s2, accM2[e], b);
        a = fmaf(m3, accM3[e], a); b = fmaf(s3, accM3[e], b);
        mine[e] = a;
        send[e] = b;
      }
      float* xw = act + XROW2 * NF + ((rt * 2 + ct) * 2 + kq) * 1024 + lane * 4;         // [q][lane] float4
      const float* xr = act + XROW2 * NF + ((rt * 2 + ct) * 2 + (kq ^ 1)) * 1024 + lane * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(xw + q * 256) = make_float4(send[4 * q], send[4 * q + 1], send[4 * q + 2], send[4 * q + 3]);
      /*@LST(layer, 2)*/
      // every wave has read this layer's input activations, and the partner's partial sums have been written
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      /*@LST(layer, 3)*/
      float4 part[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) part[q] = *reinterpret_cast<const float4*>(xr + q * 256);
      /*@LST(layer, 4)*/
      // in place: v = v + leaky(conv(v) + b)   (lib/model.py:85-89); only real cells are written, the rest stay 0
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 n;
        n.x = old[q].x + leaky((mine[4 * q] + part[q].x) + bq[q].x, slope);
        n.y = old[q].y + leaky((mine[4 * q + 1] + part[q].y) + bq[q].y, slope);
        n.z = old[q].z + leaky((mine[4 * q + 2] + part[q].z) + bq[q].z, slope);
        n.w = old[q].w + leaky((mine[4 * q + 3] + part[q].w) + bq[q].w, slope);
        if (ov) *reinterpret_cast<float4*>(rowp + (((g0 + 2 * q) ^ kk) << 2)) = n;
        res0[q] = ov ? n : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else if constexpr (KS == 4) {
      // 4-way K-split, balanced epilogue: the four waves of a tile (kq = 0..3: the quarters of K) each FINISH a quarter of
      // the tile's outputs -- output row 2ty + (kq >> 1), channel groups 2 (kq & 1) and 2 (kq & 1) + 1 -- instead of wave 0
      // collecting 96 floats per lane in two rounds.  Every wave forms all 32 partial sums, keeps its quarter and hands
      // the other three to their owners through activation rows the tile never touches (rows 63..254: TB4 * HW <= 63,
      // checked at upload): 6 ds_write_b128 in front of the "inputs read" barrier, 6 ds_read_b128 behind it.  The sums
      // are taken in the order kq = 0, 1, 2, 3 as before: bit-identical.  Two barriers per layer instead of four.
      f32x16 accY0, accY1;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float y0 = fmaf(1.f, accM0[e], 0.f), y1 = fmaf(0.f, accM0[e], 0.f);
        y0 = fmaf(1.f, accM1[e], y0); y1 = fmaf(1.f, accM1[e], y1);
        y0 = fmaf(1.f, accM2[e], y0); y1 = fmaf(-1.f, accM2[e], y1);
        y0 = fmaf(0.f, accM3[e], y0); y1 = fmaf(-1.f, accM3[e], y1);
        accY0[e] = y0;
        accY1[e] = y1;
      }
      const int myrow = kq >> 1, qb = (kq & 1) * 2;  // wave-uniform
      const float* bias = p.b_res + layer * NF + ct * 32 + 4 * h;
      const int g0 = ct * 8 + h;  // granule of group q is g0 + 2q
      float* rowp = act + (orow0 + (myrow ? p.W : 0)) * NF;
      const int kk = (orow0 + (myrow ? p.W : 0)) & 15;
      const bool ov = myrow ? ovalid1 : ovalid0;
      float4 bq[2], old[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) bq[j] = *reinterpret_cast<const float4*>(bias + 8 * (qb + j));
      if (layer == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          old[j] = ov ? *reinterpret_cast<const float4*>(rowp + (((g0 + 2 * (qb + j)) ^ kk) << 2)) : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) old[j] = res0[j];
      }
// quarter D of the partial sums, group J of it: row D >> 1, channel group 2 (D & 1) + J   (D, J compile-time)
#define CARO_QUARTER(D, J)                                                                                              \
  (((D) >> 1) ? make_float4(accY1[4 * (2 * ((D) & 1) + (J))], accY1[4 * (2 * ((D) & 1) + (J)) + 1],                      \
                            accY1[4 * (2 * ((D) & 1) + (J)) + 2], accY1[4 * (2 * ((D) & 1) + (J)) + 3])                  \
              : make_float4(accY0[4 * (2 * ((D) & 1) + (J))], accY0[4 * (2 * ((D) & 1) + (J)) + 1],                      \
                            accY0[4 * (2 * ((D) & 1) + (J)) + 2], accY0[4 * (2 * ((D) & 1) + (J)) + 3]))
      float* xa = act + XROW4 * NF + lane * 4;  // [ct][owner d][source s' (the other three in order)][j][lane] float4
      float4 mine[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        if (d == kq) {
          mine[0] = CARO_QUARTER(d, 0);
          mine[1] = CARO_QUARTER(d, 1);
        } else {
          const int sp = kq < d ? kq : kq - 1;
          float* dst = xa + (((ct * 4 + d) * 3 + sp) * 2) * 256;
          *reinterpret_cast<float4*>(dst) = CARO_QUARTER(d, 0);
          *reinterpret_cast<float4*>(dst + 256) = CARO_QUARTER(d, 1);
        }
      }
#undef CARO_QUARTER
      /*@LST(layer, 2)*/
      // every wave has read this layer's input activations, and the partial sums for the other waves have been written
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      /*@LST(layer, 3)*/
      float4 part[4][2];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        part[k][0] = part[k][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k != kq) {
          const int sp = k < kq ? k : k - 1;
          const float* src = xa + (((ct * 4 + kq) * 3 + sp) * 2) * 256;
          part[k][0] = *reinterpret_cast<const float4*>(src);
          part[k][1] = *reinterpret_cast<const float4*>(src + 256);
        }
      }
      /*@LST(layer, 4)*/
      // in place: v = v + leaky(conv(v) + b)   (lib/model.py:85-89); only real cells are written, the rest stay 0
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float4 v = kq == 0 ? mine[j] : part[0][j];
#pragma unroll
        for (int k = 1; k < 4; ++k) {
          const float4 t = k == kq ? mine[j] : part[k][j];
          v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        float4 n;
        n.x = old[j].x + leaky(v.x + bq[j].x, slope);
        n.y = old[j].y + leaky(v.y + bq[j].y, slope);
        n.z = old[j].z + leaky(v.z + bq[j].z, slope);
        n.w = old[j].w + leaky(v.w + bq[j].w, slope);
        if (ov) *reinterpret_cast<float4*>(rowp + (((g0 + 2 * (qb + j)) ^ kk) << 2)) = n;
        res0[j] = ov ? n : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
    // output transform: Y0 += {1,1,1,0}[p] * M_p,  Y1 += {0,1,-1,-1}[p] * M_p, p = 0..3 in this order
    f32x16 accY0, accY1;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float y0 = fmaf(1.f, accM0[e], 0.f), y1 = fmaf(0.f, accM0[e], 0.f);
      y0 = fmaf(1.f, accM1[e], y0); y1 = fmaf(1.f, accM1[e], y1);
      y0 = fmaf(1.f, accM2[e], y0); y1 = fmaf(-1.f, accM2[e], y1);
      y0 = fmaf(0.f, accM3[e], y0); y1 = fmaf(-1.f, accM3[e], y1);
      accY0[e] = y0;
      accY1[e] = y1;
    }
    // what the in-place update needs besides the sums -- the bias (global memory: an L2 latency) and this lane's OLD
    // activations (its own cells: nobody else writes them) -- is requested in front of the barrier and arrives while
    // the waves wait for one another (requested behind it, as before, it cost every layer 2 k idle cycles)
    const float* bias = p.b_res + layer * NF + ct * 32 + 4 * h;
    const int g0 = ct * 8 + h;  // granule of group q is g0 + 2q
    float* row0p = act + orow0 * NF;
    float* row1p = row0p + p.W * NF;
    const int k0 = orow0 & 15, k1 = (orow0 + p.W) & 15;
    float4 bq[4], old0[4], old1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const float4*>(bias + 8 * q);
    if (layer == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        old0[q] = ovalid0 ? *reinterpret_cast<const float4*>(row0p + (((g0 + 2 * q) ^ k0) << 2)) : make_float4(0.f, 0.f, 0.f, 0.f);
        old1[q] = ovalid1 ? *reinterpret_cast<const float4*>(row1p + (((g0 + 2 * q) ^ k1) << 2)) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {  // this lane wrote these very cells in the last layer's epilogue: they stayed in registers
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        old0[q] = res0[q];
        old1[q] = res1[q];
      }
    }
    /*@LST(layer, 2)*/
    // every wave has read this layer's input activations (its LDS reads were waited for in the last step): they may be
    // overwritten.  A bare s_barrier: __syncthreads() would first wait for the loads just requested
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    /*@LST(layer, 3)*/
    /*@LST(layer, 4)*/
    const bool writer = true;  // (the K-split tiles have their own epilogues above)
    // in place: v = v + leaky(conv(v) + b)   (lib/model.py:85-89); only real cells are written, the rest stay 0
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 n0, n1;
      n0.x = old0[q].x + leaky(accY0[4 * q] + bq[q].x, slope);
      n0.y = old0[q].y + leaky(accY0[4 * q + 1] + bq[q].y, slope);
      n0.z = old0[q].z + leaky(accY0[4 * q + 2] + bq[q].z, slope);
      n0.w = old0[q].w + leaky(accY0[4 * q + 3] + bq[q].w, slope);
      n1.x = old1[q].x + leaky(accY1[4 * q] + bq[q].x, slope);
      n1.y = old1[q].y + leaky(accY1[4 * q + 1] + bq[q].y, slope);
      n1.z = old1[q].z + leaky(accY1[4 * q + 2] + bq[q].z, slope);
      n1.w = old1[q].w + leaky(accY1[4 * q + 3] + bq[q].w, slope);
      if (ovalid0 && writer) *reinterpret_cast<float4*>(row0p + (((g0 + 2 * q) ^ k0) << 2)) = n0;
      if (ovalid1 && writer) *reinterpret_cast<float4*>(row1p + (((g0 + 2 * q) ^ k1) << 2)) = n1;
      res0[q] = ovalid0 ? n0 : make_float4(0.f, 0.f, 0.f, 0.f);
      res1[q] = ovalid1 ? n1 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    }
    /*@LST(layer, 5)*/
    // new activations visible to every wave; it is also the chunk barrier of the next layer's first chunk (this
    // wave's share of its second chunk has arrived, nobody reads this layer's last chunks any more)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    /*@LST(layer, 6)*/
  }
  /*@SHIFT_END(KS, wave)*/
#undef CARO_STEP
#undef CARO_AWAIT
#undef CARO_ALOAD
}

__global__ __launch_bounds__(NT, 2) void k_net_forward_w(NetParams p0, NetParams p1,
                                                           const float* __restrict__ planes,
                                                           const int32_t* __restrict__ counts, int which, int row1,
                                                           float* __restrict__ probs, float* __restrict__ values,
                                                           unsigned long long* __restrict__ stamps,
                                                           const int32_t* __restrict__ gpack, int gG, int gB) {
  __shared__ __attribute__((aligned(256))) float lds[LDS_FLOATS];  // trunk_w XORs granule bits into LDS addresses
  float* act = lds;
  float* wbuf = lds + ACT;

  /*@PST(0)*/
  const unsigned long long t_abs0 = stamps ? __builtin_amdgcn_s_memrealtime() : 0;  // diagnostic only
  // slot form: this thread's share of the games' leaf counts is requested beside the launch's totals (tile_rows_pre)
  const int gcpt = gpack ? (gG + NT - 1) / NT : 0;
  const bool gpre = gpack && gcpt <= GP_PRE;
  int gv[GP_PRE];
#pragma unroll
  for (int u = 0; u < GP_PRE; ++u) {
    const int g = (int)threadIdx.x * gcpt + u;
    gv[u] = gpre && u < gcpt && g < gG ? gpack[g] : 0;
  }
  int L, row0, board0, nb, nb_cap = p0.TB;
  int ks = 1;  // K-split of this workgroup's tiles (1: a full tile of TB boards)
  bool second = false;
  if (which < 2) {
    L = counts[which];
    row0 = which ? counts[0] : 0;
    board0 = blockIdx.x * p0.TB;
    // Tile size by launch size.  A full tile is TB boards (128 GEMM rows); smaller tiles split the K loop over the
    // waves instead (2-way: TB2 boards, 0.6 of a full tile's time; 4-way: TB4 boards, 0.4).  One round of the chip
    // holds ncu workgroups, so
    //   L <= ncu * TB4 / ncu * TB2 : every tile is a 4-way / 2-way tile (small launches finish sooner);
    //   L a little above one round of full tiles: workgroups [0, ncu) stay full, the overflow goes into small
    //   tiles and the second round is short;
    //   otherwise full tiles.
    const int full = p0.ncu * p0.TB;
    if (p0.ncu > 0) {
      if (p0.TB4 > 0 && L <= p0.ncu * p0.TB4) ks = 4;
      else if (p0.TB2 > 0 && L <= p0.ncu * p0.TB2) ks = 2;
      if (ks > 1) {
        nb_cap = ks == 4 ? p0.TB4 : p0.TB2;
        board0 = (int)blockIdx.x * nb_cap;
      } else if (L > full) {
        const int over = L - full;
        if (p0.TB4 > 0 && over <= p0.ncu * p0.TB4) ks = 4;
        else if (p0.TB2 > 0 && over <= p0.ncu * p0.TB2) ks = 2;
        if (ks > 1) {
          if ((int)blockIdx.x < p0.ncu) {
            ks = 1;
          } else {
            nb_cap = ks == 4 ? p0.TB4 : p0.TB2;
            board0 = full + ((int)blockIdx.x - p0.ncu) * nb_cap;
          }
        }
      }
    }
  } else {
    // two nets in one launch: the tile size follows the sum (one workgroup of slack: each class rounds up)
    const int L0 = counts[0], L1 = counts[1];
    if (p0.ncu > 1) {
      if (p0.TB4 > 0 && L0 + L1 <= (p0.ncu - 1) * p0.TB4) ks = 4;
      else if (p0.TB2 > 0 && L0 + L1 <= (p0.ncu - 1) * p0.TB2) ks = 2;
      if (ks > 1) nb_cap = ks == 4 ? p0.TB4 : p0.TB2;
    }
    const int t0 = (L0 + nb_cap - 1) / nb_cap;
    second = (int)blockIdx.x >= t0;
    L = second ? L1 : L0;
    row0 = second ? (row1 >= 0 ? row1 : L0) : 0;
    board0 = (second ? (int)blockIdx.x - t0 : (int)blockIdx.x) * nb_cap;
  }
  if (board0 >= L) return;
  /*@PST(1)*/
  const NetParams p = second ? p1 : p0;
  unsigned long long t_c0 = 0, t_r0 = 0;  // diagnostic only, as in k_net_forward
  if (stamps) {
    t_c0 = __builtin_amdgcn_s_memtime();
    t_r0 = __builtin_amdgcn_s_memrealtime();
  }
  nb = min(nb_cap, L - board0);
  const int HW = p.HW;
  const int R = nb * HW;  // real rows
  const int tid = threadIdx.x;

  // this thread's leaf count of the tile's class: consumed HERE, in front of the weight transfers (the wait for gv is
  // then a wait for gv alone)
  int gmine = 0;
#pragma unroll
  for (int u = 0; u < GP_PRE; ++u) gmine += (gv[u] >> 8) == (second ? 1 : 0) ? (gv[u] & 0xFF) : 0;
  asm volatile("" : "+v"(gmine));
  // the first two weight chunks are on their way into ring buffers 0 and 1 while conv_in runs; its scratch (the
  // conv_in weights, the row map) sits in buffer 2, which is fetched into only after the trunk's first barrier
  float* win = wbuf + 2 * WCH;
  // conv_in's weights first (1152 floats: the first five waves move 16 bytes per lane, b_in and a few floats more
  // come along), then the two chunks: the wait below is for the oldest transfer only
  if (tid < 320) dma_b128(reinterpret_cast<const float4*>(p.w_in) + tid,
                          __builtin_amdgcn_readfirstlane(lds_addr(win) + (unsigned)(tid >> 6) * 1024u));
  fetch_chunk(p.ww, 0, lds_addr(wbuf), tid);
  fetch_chunk(p.ww, 1, lds_addr(wbuf), tid);
  // rows of boards this tile does not have, the spare rows and the zero row stay zero for ever; the others are written
  // by conv_in
  for (int k = tid + (R * NF) / 4; k < ACT / 4; k += NT) reinterpret_cast<float4*>(lds)[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  // the wait below leaves exactly this thread's 2 chunks x (WCH / 4 / NT) transfers in flight (conv_in's weights are
  // the oldest transfer): the immediate is tied to the constants here.  conv_in reads w_in from LDS and b_in from
  // global memory, so only w_in has to be covered by the 320-lane transfer (what comes along behind it is not used).
  static_assert(2 * (WCH / 4 / NT) == 8, "s_waitcnt vmcnt(8) below counts 2 chunks x WCH / 4 / NT transfers per thread");
  static_assert(320 * 4 >= 9 * 2 * NF, "the 320-lane transfer must cover w_in [9][2][64]");
  /*@PST(2)*/
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // the 2 x 4 chunk transfers of this thread may still be on their way
  /*@PST(3)*/
  int* smap = reinterpret_cast<int*>(win + 1536);  // [TB] plane / output row of every board of this tile
  if (gpre) tile_rows_pre(gv, gcpt, gmine, gB, second ? 1 : 0, board0, nb, smap + 64, smap, tid);  // ends with a barrier
  else tile_rows(gpack, gG, gB, second ? 1 : 0, row0 + board0, board0, nb, smap + 64, smap, tid);
  /*@PST(4)*/
  conv_in_mfma(p, planes, smap, act, win, R, tid);
  /*@PST(5)*/
  const int slot_v = tid < nb ? smap[tid] : 0;
  unsigned long long t_trunk0 = 0;
  if (stamps) t_trunk0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // conv_in's output and the two chunks are visible to every wave
  /*@PST(6)*/
  // head parameters staged in LDS during the trunk's last chunks when they fit (heads_f32)
  const int hspan = head_span(HW, p.A) <= HEAD_STAGE_MAX ? head_span(HW, p.A) : 0;
  if (ks == 1) trunk_w<1>(p, act, wbuf, nb, tid, hspan);
  else if (ks == 2) trunk_w<2>(p, act, wbuf, nb, tid, hspan);
  else trunk_w<4>(p, act, wbuf, nb, tid, hspan);
  unsigned long long t_trunk1 = 0;
  if (stamps) t_trunk1 = __builtin_amdgcn_s_memtime();
  /*@PST(8)*/
  if (hspan) heads_f32<true>(p, act, wbuf, probs, values, slot_v, nb, R, tid);
  else heads_f32<false>(p, act, wbuf, probs, values, slot_v, nb, R, tid);
  /*@PST(12)*/
  if (stamps && tid == 0) {
    stamps[4 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t_c0;
    // slot launches (the engine's own: caro_net_debug_stamps) carry the wall clock of the workgroup's START above bit 20
    stamps[4 * blockIdx.x + 1] = gpack ? ((__builtin_amdgcn_s_memrealtime() - t_abs0) & 0xFFFFFull) | (t_abs0 << 20)
                                       : __builtin_amdgcn_s_memrealtime() - t_r0;
    stamps[4 * blockIdx.x + 2] = t_trunk0 - t_c0;
    stamps[4 * blockIdx.x + 3] = t_trunk1 - t_c0;
    /*@HST_PUBLISH(stamps, wbuf)*/
  }
}


// ===================================================================================================
// 2-D Winograd form F(2x2,3x3) ("f32w2") for LARGE boards (one board per workgroup: 12x12 .. 15x15): the same float32
// network function with 16 / 36 of the direct form's multiplies (the row form above: 24 / 36).  Reference:
// lib/model.py:36-47,85-89 (five 64 -> 64 3x3 convolutions with a residual each).
//     output tile (ty, tx) = cells (2ty + u, 2tx + v), u, v in {0, 1};  d[r][j] = input cell (2ty-1+r, 2tx-1+j), zero outside
//     V[a][b] = sum_rj Bt[a][r] Bt[b][j] d[r][j]      Bt = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]      (adds only)
//     U[a][b] = sum    G[a][ky] G[b][kx] w[ky][kx]    G  = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]          (host, float64, one rounding)
//     M[a][b] = sum_ci U[a][b][co][ci] V[a][b][ci]                                                       (MFMA)
//     Y[u][v] = sum_ab At[u][a] At[v][b] M[a][b]      At = [1 1 1 0; 0 1 -1 -1]
// GEMM rows are the 64 tiles of a board (8 x 8 for 15 x 15: exactly two 32-row MFMA blocks, no padding rows; the row
// form needs 120 tiles in 128 rows), 16 transformed taps of K = 64.  Work split: wave = (row tile rt, column tile ct,
// bh) -- 32 tiles x 32 channels, and the wave's HALF of the taps: b in {1, 0} for bh = 0, b in {2, 3} for bh = 1, all
// four a.  A wave keeps the four accumulators M[0..3][b] of ONE b at a time:
//     phase 0  b = 1 | 2:  K loop, then Z[u] = sum_a At[u][a] M[a][b] (32 registers) stays behind
//     phase 1  b = 0 | 3:  K loop, then the same fold
// and FINISHES output column v = bh of its tiles:  Y[u][0] = Z[u][b=0] + Z[u][1] + Z[u][2]  (bh = 0; Z[u][2] comes from
// the partner wave), Y[u][1] = Z[u][1] - Z[u][2] - Z[u][3]  (bh = 1; Z[u][1] from the partner).  What a wave hands to its
// partner is exactly its phase-0 fold.  The exchange needs no extra LDS: after the "inputs read" barrier the activation
// buffer is dead, so the partner's partial sums are written INTO THE PARTNER'S OUTPUT CELLS, read back from there behind
// a second barrier and overwritten with the finished activations (three barriers per layer).
// Operand stream per (b, channel granule): 8 cell reads (4 tile rows x the 2 columns b combines) + 4 weight granules
// -> column combine, row transform (as the row form: V0 = c0-c2, V1 = c1+c2, V2 = c2-c1, V3 = c1-c3) -> 16 MFMAs.
// Weight chunk = 32 KiB = [2 b of the phase][4 a][2 h][64 co][2 channel granules]: 8 chunks per layer (4 per phase)
// through the same ring of three buffers, one workgroup barrier per chunk (32 MFMAs of a wave).
// Activation rows keep the index cell = y * W + x; their granule swizzle key is akey<true> (tile coordinates), under
// which the 16 lanes of every ds_read_b128 group -- 8 tile columns x 2 tile rows, see the lane -> tile map below --
// read 16 different slots for any cell offset.
constexpr int W2NCHUNK = NRES * 8;        // 40 chunks of WCH floats
constexpr int W2SETS = 16;                // operand sets per layer and wave: 2 phases x 8 channel granules

__device__ __forceinline__ void trunk_w2d(const NetParams& p, float* act, float* wbuf, int tid) {
  const float slope = p.slope;
  const int H = p.H, W = p.W;
  const int wave = tid >> 6, lane = tid & 63;
  const int i = lane & 31, h = lane >> 5;
  const int ct = wave & 1, rt = (wave >> 1) & 1, bh = wave >> 2;
  // lane -> tile.  ds_read_b128 is served in the lane groups {0-3,12-15,20-27} and {4-11,16-19,28-31} of each half
  // (MI355X_MICROARCH.md, LDS): quads 0,3,5,6 form group 0 and quads 1,2,4,7 group 1, and quad >> 1 numbers the quads
  // of either group 0..3.  Position q = 0..15 inside the group -> tile column q & 7, tile row (group * 2 + (q >> 3)).
  const int quad = i >> 2, grp = (0x96 >> quad) & 1, qq = ((quad >> 1) << 2) | (i & 3);
  const int tx = qq & 7, ty = rt * 4 + grp * 2 + (qq >> 3);
  const bool tvalid = ty < ((H + 1) >> 1) && tx < ((W + 1) >> 1);
  const unsigned abase = lds_addr(act);
  const unsigned wring = lds_addr(wbuf);
  // LDS byte address of input cell (r, j) of the tile, channel granule 8h (+ G, XORed in per set), swizzle key folded
  // in.  A cell outside the board reads the zero row -- any of its slots: the one the virtual cell's key selects, which
  // keeps the lanes of a group on different slots.
  auto caddr = [&](int r, int j) -> unsigned {
    const int y = 2 * ty - 1 + r, x = 2 * tx - 1 + j;
    const bool ok = tvalid && y >= 0 && y < H && x >= 0 && x < W;
    const int row = ok ? y * W + x : ZROW;
    const int key = ((tx + (j >> 1)) & 7) | (((ty + (r >> 1)) & 1) << 3);
    return abase + (unsigned)(row * NF + (((h * 8) ^ key) << 2)) * 4u;
  };
  // the two input columns a phase combines, c_r = d[r][jA] + sg * d[r][jB]:
  //   bh 0: phase 0 (b = 1)  x1 + x2      phase 1 (b = 0)  x0 - x2
  //   bh 1: phase 0 (b = 2)  x2 - x1      phase 1 (b = 3)  x1 - x3
  const int jA0 = bh ? 2 : 1, jB0 = bh ? 1 : 2, jA1 = bh ? 1 : 0, jB1 = bh ? 3 : 2;
  const float sg0 = bh ? -1.f : 1.f, sg1 = -1.f;
  unsigned ra[2][4], rb[2][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    ra[0][r] = caddr(r, jA0); rb[0][r] = caddr(r, jB0);
    ra[1][r] = caddr(r, jA1); rb[1][r] = caddr(r, jB1);
  }
  // this lane's weight row in ring buffer 0: [b = bh][a = 0][h][co = ct*32 + i], 2 granules of 16 bytes, the granule
  // index XORed with (co >> 3) & 1 (lanes i and i ^ 8.. of a read group then sit on different slots)
  const unsigned wlane = wring + (unsigned)(bh * 16384 + (h * 64 + ct * 32 + i) * 32 + (((i >> 3) & 1) << 4));
  // output side (weights are the first MFMA operand: a lane's 16 accumulator registers are 4 groups q of 4 consecutive
  // channels ct*32 + 8q + 4h + 0..3 of ITS tile): this wave finishes cells (2ty + u, 2tx + bh), its partner the
  // cells (2ty + u, 2tx + 1 - bh)
  const int xo = 2 * tx + bh, xp = 2 * tx + 1 - bh;
  const int g0 = ct * 8 + h;  // granule of channel group q is g0 + 2q
  // float index (into act) of channel group 0 of an output cell; group q sits at index ^ (8 q): its granule is
  // (g0 + 2q) ^ key, g0 + 2q = g0 | 2q (g0 = 8 ct + h never carries into bits 1-2), so the XOR with 2q commutes with
  // the key.  ONE index per cell instead of four addresses: the epilogue's eight + eight addresses held across the
  // main loops were what the register allocator spilled (14 registers, reloaded one by one in front of their uses).
  int oidx[2], pidx[2];
  bool ovalid[2], pvalid[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int y = 2 * ty + u;
    ovalid[u] = tvalid && y < H && xo < W;
    pvalid[u] = tvalid && y < H && xp < W;
    const int okey = (((xo + 1) >> 1) & 7) | ((((y + 1) >> 1) & 1) << 3);
    const int pkey = (((xp + 1) >> 1) & 7) | ((((y + 1) >> 1) & 1) << 3);
    oidx[u] = (y * W + xo) * NF + ((g0 ^ okey) << 2);
    pidx[u] = (y * W + xp) * NF + ((g0 ^ pkey) << 2);
  }

  f4v CA0, CA1, CA2, CA3, CB0, CB1, CB2, CB3;  // the 4 x 2 input cells of the operand set in flight
  f4v W0, W1, W2, W3, X0, X1, X2, X3;          // its weight granules: two sets, the MFMAs read one while the other loads
// operand set T of the layer: phase T >> 3, channel granule G = T & 7 of the lane half, chunk T >> 1 of the layer
#define W2_ALOAD(B0, B1, B2, B3, T)                                                                          \
  {                                                                                                          \
    constexpr int ph_ = (T) >> 3, G_ = (T) & 7, cq_ = (T) >> 1;                                              \
    const unsigned wa_ = (wlane + rbuf[cq_ % WNBUF]) ^ ((unsigned)(G_ & 1) << 4);                            \
    asm volatile("ds_read_b128 %0, %1" : "=v"(CA0) : "v"(ra[ph_][0] ^ (G_ << 4)));                           \
    asm volatile("ds_read_b128 %0, %1" : "=v"(CB0) : "v"(rb[ph_][0] ^ (G_ << 4)));                           \
    asm volatile("ds_read_b128 %0, %1" : "=v"(CA1) : "v"(ra[ph_][1] ^ (G_ << 4)));                           \
    asm volatile("ds_read_b128 %0, %1" : "=v"(CB1) : "v"(rb[ph_][1] ^ (G_ << 4)));                           \
    asm volatile("ds_read_b128 %0, %1" : "=v"(CA2) : "v"(ra[ph_][2] ^ (G_ << 4)));                           \
    asm volatile("ds_read_b128 %0, %1" : "=v"(CB2) : "v"(rb[ph_][2] ^ (G_ << 4)));                           \
    asm volatile("ds_read_b128 %0, %1" : "=v"(CA3) : "v"(ra[ph_][3] ^ (G_ << 4)));                           \
    asm volatile("ds_read_b128 %0, %1" : "=v"(CB3) : "v"(rb[ph_][3] ^ (G_ << 4)));                           \
    asm volatile("ds_read_b128 %0, %1" : "=v"(B0) : "v"(wa_));                                               \
    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(B1) : "v"(wa_));                                   \
    asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(B2) : "v"(wa_));                                   \
    asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(B3) : "v"(wa_));                                  \
  }
// every LDS read this wave has issued is complete (the compiler does not know of the reads above)
#define W2_AWAIT(B0, B1, B2, B3)                                                                             \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                        \
               : "+v"(CA0), "+v"(CA1), "+v"(CA2), "+v"(CA3), "+v"(CB0), "+v"(CB1), "+v"(CB2), "+v"(CB3),     \
                 "+v"(B0), "+v"(B1), "+v"(B2), "+v"(B3));
// One step of the layer's software pipeline (the structure of trunk_w's): set T was requested a whole burst ago; wait,
// at the first set of a chunk pass the chunk's ONE workgroup barrier (this wave's share of chunk c+1 has arrived ->
// barrier -> chunk c+1 visible to every wave, nobody reads chunk c-1 any more -> chunk c+2 goes into that buffer),
// combine the columns, form each transformed operand in front of its first MFMA, request set T+1 behind the first
// round of MFMAs, issue the other twelve.
#define W2_STEP(T, B0, B1, B2, B3, NB0, NB1, NB2, NB3)                                                       \
  W2_AWAIT(B0, B1, B2, B3)                                                                                   \
  if ((T) % 2 == 0) {                                                                                        \
    if ((T) != 0) {                                                                                          \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                       \
      /*@CHUNK_BARRIER*/ __syncthreads();                                                                    \
    }                                                                                                        \
    if (/*@FETCH_ON*/ c0 + (T) / 2 + 2 < W2NCHUNK) fetch_chunk_s(p.ww2, c0 + (T) / 2 + 2, wring, tid);       \
  }                                                                                                          \
  {                                                                                                          \
    const float sg_ = ((T) >> 3) ? sg1 : sg0;                                                                \
    const f4v c0_ = CA0 + sg_ * CB0, c1_ = CA1 + sg_ * CB1, c2_ = CA2 + sg_ * CB2, c3_ = CA3 + sg_ * CB3;    \
    const f4v v0 = c0_ - c2_;                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    /* a phase's first MFMAs start from C = 0 (an inline constant, no registers to clear) */                    \
    accM0 = __builtin_amdgcn_mfma_f32_32x32x2f32(B0.x, v0.x, (T) % 8 == 0 ? zero16 : accM0, 0, 0, 0);        \
    const f4v v1 = c1_ + c2_;                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    accM1 = __builtin_amdgcn_mfma_f32_32x32x2f32(B1.x, v1.x, (T) % 8 == 0 ? zero16 : accM1, 0, 0, 0);        \
    const f4v v2 = c2_ - c1_;                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    accM2 = __builtin_amdgcn_mfma_f32_32x32x2f32(B2.x, v2.x, (T) % 8 == 0 ? zero16 : accM2, 0, 0, 0);        \
    const f4v v3 = c1_ - c3_;                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    accM3 = __builtin_amdgcn_mfma_f32_32x32x2f32(B3.x, v3.x, (T) % 8 == 0 ? zero16 : accM3, 0, 0, 0);        \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    if ((T) + 1 < W2SETS) W2_ALOAD(NB0, NB1, NB2, NB3, ((T) + 1) % W2SETS)                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    accM0 = __builtin_amdgcn_mfma_f32_32x32x2f32(B0.y, v0.y, accM0, 0, 0, 0);                                \
    accM1 = __builtin_amdgcn_mfma_f32_32x32x2f32(B1.y, v1.y, accM1, 0, 0, 0);                                \
    accM2 = __builtin_amdgcn_mfma_f32_32x32x2f32(B2.y, v2.y, accM2, 0, 0, 0);                                \
    accM3 = __builtin_amdgcn_mfma_f32_32x32x2f32(B3.y, v3.y, accM3, 0, 0, 0);                                \
    accM0 = __builtin_amdgcn_mfma_f32_32x32x2f32(B0.z, v0.z, accM0, 0, 0, 0);                                \
    accM1 = __builtin_amdgcn_mfma_f32_32x32x2f32(B1.z, v1.z, accM1, 0, 0, 0);                                \
    accM2 = __builtin_amdgcn_mfma_f32_32x32x2f32(B2.z, v2.z, accM2, 0, 0, 0);                                \
    accM3 = __builtin_amdgcn_mfma_f32_32x32x2f32(B3.z, v3.z, accM3, 0, 0, 0);                                \
    accM0 = __builtin_amdgcn_mfma_f32_32x32x2f32(B0.w, v0.w, accM0, 0, 0, 0);                                \
    accM1 = __builtin_amdgcn_mfma_f32_32x32x2f32(B1.w, v1.w, accM1, 0, 0, 0);                                \
    accM2 = __builtin_amdgcn_mfma_f32_32x32x2f32(B2.w, v2.w, accM2, 0, 0, 0);                                \
    accM3 = __builtin_amdgcn_mfma_f32_32x32x2f32(B3.w, v3.w, accM3, 0, 0, 0);                                \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
  }
#define W2_PAIR(T)                                                                                           \
  W2_STEP(T, W0, W1, W2, W3, X0, X1, X2, X3)                                                                 \
  W2_STEP((T) + 1, X0, X1, X2, X3, W0, W1, W2, W3)
// Z[u] = sum_a At[u][a] M[a]: u = 0: M0 + M1 + M2, u = 1: M1 - M2 - M3 (left to right)
#define W2_FOLD(Z0_, Z1_)                                                                                    \
  _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                                           \
    Z0_[e] = (accM0[e] + accM1[e]) + accM2[e];                                                               \
    Z1_[e] = (accM1[e] - accM2[e]) - accM3[e];                                                               \
  }

  for (int layer = 0; layer < NRES; ++layer) {
    const int c0 = layer * 8;  // first chunk of the layer; chunk c0 + k sits in ring buffer (c0 + k) % WNBUF
    unsigned rbuf[WNBUF];
#pragma unroll
    for (int k = 0; k < WNBUF; ++k) rbuf[k] = (unsigned)((c0 + k) % WNBUF) * (WCH * 4);
    const f32x16 zero16 = {};
    f32x16 accM0, accM1, accM2, accM3, Zs0, Zs1, Zt0, Zt1;
    // (the per-set addresses are re-formed in every layer: hoisted out of the layer loop they would take 128 registers)
#pragma unroll
    for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(ra[0][r]), "+v"(rb[0][r]), "+v"(ra[1][r]), "+v"(rb[1][r]));
    /*@LST(layer, 0)*/
    W2_ALOAD(W0, W1, W2, W3, 0)
    W2_PAIR(0) W2_PAIR(2) W2_PAIR(4)
    // the bias of this lane's 16 channels, requested two steps ahead of the fold that adds it
    const float* bias = p.b_res + layer * NF + ct * 32 + 4 * h;
    float4 bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const float4*>(bias + 8 * q);
    W2_PAIR(6)
    /*@LST(layer, 1)*/
    W2_FOLD(Zs0, Zs1)   // phase 0 (b = 1 | 2): what this wave keeps AND what it hands to its partner
    // the bias joins Z[b = 1] (the bh = 0 waves' phase 0), which enters all four outputs of a tile with +1: the bh = 1
    // waves get it with the partial sums their partners hand over
    if (bh == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        Zs0[4 * q] += bq[q].x; Zs0[4 * q + 1] += bq[q].y; Zs0[4 * q + 2] += bq[q].z; Zs0[4 * q + 3] += bq[q].w;
        Zs1[4 * q] += bq[q].x; Zs1[4 * q + 1] += bq[q].y; Zs1[4 * q + 2] += bq[q].z; Zs1[4 * q + 3] += bq[q].w;
      }
    }
    /*@LST(layer, 2)*/
    W2_PAIR(8) W2_PAIR(10) W2_PAIR(12) W2_PAIR(14)
    /*@LST(layer, 3)*/
    W2_FOLD(Zt0, Zt1)   // phase 1 (b = 0 | 3)
    // ---- epilogue.  Requested in front of the barrier: the OLD values of this wave's output cells (the residual input;
    // only this wave ever writes these cells)
    float4 old0[4], old1[4];
    // (the four indices are loop-invariant; "rewritten" here so that the per-group addresses are formed where they are
    // used instead of living in sixteen registers through the main loops)
    asm volatile("" : "+v"(oidx[0]), "+v"(oidx[1]), "+v"(pidx[0]), "+v"(pidx[1]));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      old0[q] = ovalid[0] ? *reinterpret_cast<const float4*>(act + (oidx[0] ^ (8 * q))) : make_float4(0.f, 0.f, 0.f, 0.f);
      old1[q] = ovalid[1] ? *reinterpret_cast<const float4*>(act + (oidx[1] ^ (8 * q))) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // this wave's own share of its outputs, formed while the loads travel: bh 0: Z[b=0] + Z[b=1], bh 1: Z[b=2] + Z[b=3]
    f32x16 S0, S1;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      S0[e] = Zt0[e] + Zs0[e];
      S1[e] = Zt1[e] + Zs1[e];
    }
    /*@LST(layer, 4)*/
    // every wave has read this layer's input activations (and this wave its old values): the buffer may be overwritten
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    /*@LST(layer, 5)*/
    {  // the partner's partial sums go into the partner's output cells
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (pvalid[0])
          *reinterpret_cast<float4*>(act + (pidx[0] ^ (8 * q))) = make_float4(Zs0[4 * q], Zs0[4 * q + 1], Zs0[4 * q + 2], Zs0[4 * q + 3]);
        if (pvalid[1])
          *reinterpret_cast<float4*>(act + (pidx[1] ^ (8 * q))) = make_float4(Zs1[4 * q], Zs1[4 * q + 1], Zs1[4 * q + 2], Zs1[4 * q + 3]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    /*@LST(layer, 6)*/
    // finish: bh 0: Y = (Z[b=0] + Z[b=1]) + Z[b=2](received);  bh 1: Y = Z[b=1](received) - (Z[b=2] + Z[b=3])  (the bias
    // came in with Z[b=1]); then in place v = v + leaky(conv(v) + bias)  (lib/model.py:85-89); only real cells are written
    const float sgn = bh ? -1.f : 1.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 r0 = ovalid[0] ? *reinterpret_cast<const float4*>(act + (oidx[0] ^ (8 * q))) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 r1 = ovalid[1] ? *reinterpret_cast<const float4*>(act + (oidx[1] ^ (8 * q))) : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 n0, n1;
      n0.x = old0[q].x + leaky(fmaf(sgn, S0[4 * q], r0.x), slope);
      n0.y = old0[q].y + leaky(fmaf(sgn, S0[4 * q + 1], r0.y), slope);
      n0.z = old0[q].z + leaky(fmaf(sgn, S0[4 * q + 2], r0.z), slope);
      n0.w = old0[q].w + leaky(fmaf(sgn, S0[4 * q + 3], r0.w), slope);
      n1.x = old1[q].x + leaky(fmaf(sgn, S1[4 * q], r1.x), slope);
      n1.y = old1[q].y + leaky(fmaf(sgn, S1[4 * q + 1], r1.y), slope);
      n1.z = old1[q].z + leaky(fmaf(sgn, S1[4 * q + 2], r1.z), slope);
      n1.w = old1[q].w + leaky(fmaf(sgn, S1[4 * q + 3], r1.w), slope);
      if (ovalid[0]) *reinterpret_cast<float4*>(act + (oidx[0] ^ (8 * q))) = n0;
      if (ovalid[1]) *reinterpret_cast<float4*>(act + (oidx[1] ^ (8 * q))) = n1;
    }
    // new activations visible to every wave; also the chunk barrier of the next layer's first chunk (this wave's share
    // of its second chunk has arrived, nobody reads this layer's last chunks any more)
    /*@LST(layer, 7)*/
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#undef W2_FOLD
#undef W2_PAIR
#undef W2_STEP
#undef W2_AWAIT
#undef W2_ALOAD
}

// One board per workgroup (TB = 1); launch interface, prologue (slot-row map, conv_in on the matrix pipe) and heads of
// k_net_forward_w, activation rows keyed by akey<true>.
__global__ __launch_bounds__(NT, 2) void k_net_forward_w2(NetParams p0, NetParams p1,
                                                            const float* __restrict__ planes,
                                                            const int32_t* __restrict__ counts, int which, int row1,
                                                            float* __restrict__ probs, float* __restrict__ values,
                                                            unsigned long long* __restrict__ stamps,
                                                            const int32_t* __restrict__ gpack, int gG, int gB,
                                                            float* __restrict__ featbuf, int32_t* __restrict__ rowlist,
                                                            const int32_t* __restrict__ slist) {
  __shared__ __attribute__((aligned(256))) float lds[LDS_FLOATS];  // trunk_w2d XORs granule bits into LDS addresses
  float* act = lds;
  float* wbuf = lds + ACT;
  // slot form: this thread's share of the games' leaf counts is requested beside the launch's totals (tile_rows_pre).
  // slist (caro_net_forward_slot_list): the slot row of every dense board comes from the producer's list instead -- one
  // board per workgroup leaves no tile to fill, so no workgroup needs the prefix sum over the G leaf counts (1.2 us of
  // each workgroup: 37 us of a 7 600-board launch), and which dense index a board got does not touch its arithmetic
  const int gcpt = gpack && !slist ? (gG + NT - 1) / NT : 0;
  const bool gpre = gpack && !slist && gcpt <= GP_PRE;
  int gv[GP_PRE];
#pragma unroll
  for (int u = 0; u < GP_PRE; ++u) {
    const int g = (int)threadIdx.x * gcpt + u;
    gv[u] = gpre && u < gcpt && g < gG ? gpack[g] : 0;
  }
  int L, row0, board0;
  bool second = false;
  if (which < 2) {
    L = counts[which];
    row0 = which ? counts[0] : 0;
    board0 = blockIdx.x;
  } else {
    const int L0 = counts[0];
    second = (int)blockIdx.x >= L0;
    L = second ? counts[1] : L0;
    row0 = second ? (row1 >= 0 ? row1 : L0) : 0;
    board0 = second ? (int)blockIdx.x - L0 : (int)blockIdx.x;
  }
  if (board0 >= L) return;
  const int listed = slist ? slist[(second ? gG * gB : 0) + board0] : 0;  // (requested here, used behind conv_in's weights)
  const NetParams p = second ? p1 : p0;
  unsigned long long t_c0 = 0, t_r0 = 0;  // diagnostic only (stamps == nullptr in every product launch)
  if (stamps) {
    t_c0 = __builtin_amdgcn_s_memtime();
    t_r0 = __builtin_amdgcn_s_memrealtime();
  }
  const int nb = 1;
  const int HW = p.HW;
  const int R = HW;
  const int tid = threadIdx.x;
  int gmine = 0;
#pragma unroll
  for (int u = 0; u < GP_PRE; ++u) gmine += (gv[u] >> 8) == (second ? 1 : 0) ? (gv[u] & 0xFF) : 0;
  asm volatile("" : "+v"(gmine));
  // the first two weight chunks are on their way into ring buffers 0 and 1 while conv_in runs; its scratch (the conv_in
  // weights, the row map) sits in buffer 2, which is fetched into only after the trunk's first barrier
  float* win = wbuf + 2 * WCH;
  if (tid < 320) dma_b128(reinterpret_cast<const float4*>(p.w_in) + tid,
                          __builtin_amdgcn_readfirstlane(lds_addr(win) + (unsigned)(tid >> 6) * 1024u));
  fetch_chunk_s(p.ww2, 0, lds_addr(wbuf), tid);
  fetch_chunk_s(p.ww2, 1, lds_addr(wbuf), tid);
  for (int k = tid + (R * NF) / 4; k < ACT / 4; k += NT) reinterpret_cast<float4*>(lds)[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  static_assert(2 * (WCH / 4 / NT) == 8, "s_waitcnt vmcnt(8) below counts 2 chunks x WCH / 4 / NT transfers per thread");
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // conv_in's weights have arrived; the 2 x 4 chunk transfers may be on their way
  int* smap = reinterpret_cast<int*>(win + 1536);
  if (slist) {
    if (tid == 0) smap[0] = listed;
    __syncthreads();
  } else if (gpre) tile_rows_pre(gv, gcpt, gmine, gB, second ? 1 : 0, board0, nb, smap + 64, smap, tid);  // ends with a barrier
  else tile_rows(gpack, gG, gB, second ? 1 : 0, row0 + board0, board0, nb, smap + 64, smap, tid);
  conv_in_mfma<true>(p, planes, smap, act, win, R, tid);
  const int slot_v = tid < nb ? smap[tid] : 0;
  unsigned long long t_trunk0 = 0;
  if (stamps) t_trunk0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // conv_in's output and the two chunks are visible to every wave
  trunk_w2d(p, act, wbuf, tid);
  unsigned long long t_trunk1 = 0;
  if (stamps) t_trunk1 = __builtin_amdgcn_s_memtime();
  // featbuf != null: the FC heads of the whole launch follow in k_net_heads (row0 + board0 = this board's dense index)
  heads_f32<false, true>(p, act, wbuf, probs, values, slot_v, nb, R, tid, featbuf, rowlist, row0 + board0);
  if (stamps && tid == 0) {
    stamps[4 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t_c0;
    stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - t_r0;
    stamps[4 * blockIdx.x + 2] = t_trunk0 - t_c0;
    stamps[4 * blockIdx.x + 3] = t_trunk1 - t_c0;
  }
}

// The two FC heads, tanh and the softmax (lib/model.py:50-67, lib/mcts.py:216) of a k_net_forward_w2 launch, 32 boards
// per workgroup: logits[32 boards][A] = features[32][2 HW] x w_p^T on the matrix pipe (wave w takes the 32 actions from
// 32 w; K runs over the (plane, cell) order of the reference's flatten, four cells per lane and load: the A operand from
// the staged features, the B operand from the quad-transposed policy matrix w_pT, whose 16-byte granules are
// consecutive in the action), the value head Linear(HW,20) + LeakyReLU + Linear(20,1) + tanh as fma chains in the
// reference's order, the softmax with sixteen threads per board.  One read of the policy matrix serves 32 boards.
constexpr int HB = 32;               // boards per workgroup
constexpr int HEADS_MAX_CELLS = 225;  // largest board k_net_heads is sized for (caro_net_winograd2d_supported)
constexpr int HPL = 228;             // floats per staged feature plane (225 cells, rows 16-byte aligned)
static_assert(HPL >= HEADS_MAX_CELLS && HPL % 4 == 0, "a staged feature plane holds every cell of the largest board");
constexpr int HFS = 3 * HPL + 8;     // floats per staged board (692 = 52 mod 64: the 32 rows of an operand read spread over the banks)
constexpr int HLG = 256;             // floats per logit row (A <= 255)
__global__ __launch_bounds__(NT, 1) void k_net_heads(NetParams p0, NetParams p1, const int32_t* __restrict__ counts,
                                                       int which, int row1, const float* __restrict__ featbuf,
                                                       const int32_t* __restrict__ rowlist, float* __restrict__ probs,
                                                       float* __restrict__ values) {
  __shared__ __attribute__((aligned(16))) float F[HB * HFS];
  __shared__ __attribute__((aligned(16))) float logit[HB * HLG];
  __shared__ float wv1[20 * HEADS_MAX_CELLS];
  __shared__ float hid[HB * 20];
  __shared__ float part[HB * 16];
  __shared__ float rsum[HB];
  __shared__ int orow[HB];
  int L, base, blk;
  bool second = false;
  if (which < 2) {
    L = counts[which];
    base = which ? counts[0] : 0;
    blk = blockIdx.x;
  } else {
    const int L0 = counts[0], nb0 = (L0 + HB - 1) / HB;
    second = (int)blockIdx.x >= nb0;
    L = second ? counts[1] : L0;
    base = second ? (row1 >= 0 ? row1 : L0) : 0;
    blk = second ? (int)blockIdx.x - nb0 : (int)blockIdx.x;
  }
  const int i0 = blk * HB;
  if (i0 >= L) return;
  const int nbrd = L - i0 < HB ? L - i0 : HB;
  const NetParams p = second ? p1 : p0;
  const int HW = p.HW, A = p.A, tid = threadIdx.x;
  const float slope = p.slope;
  // the boards' features and the value head's first matrix (rows behind the last board stay as they are: a row of the
  // product depends on its own board only, and those rows are never written out)
  {
    // (the dense boards of a workgroup are consecutive rows of the feature buffer: one flat copy, eight requests per
    // thread in flight)
    const float* src = featbuf + (size_t)(base + i0) * 3 * HW;
    const int total = nbrd * 3 * HW;
    for (int j0 = tid; j0 < total; j0 += 8 * NT) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = j0 + u * NT < total ? src[j0 + u * NT] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = j0 + u * NT;
        if (j < total) {
          const int i = j / (3 * HW), r = j - i * 3 * HW, o = (r >= HW) + (r >= 2 * HW);
          F[i * HFS + o * HPL + (r - o * HW)] = t[u];
        }
      }
    }
    static_assert(9 * NT >= 20 * HEADS_MAX_CELLS, "nine rounds of the block cover the value head's first matrix");
    float tw[9];
#pragma unroll
    for (int u = 0; u < 9; ++u) tw[u] = tid + u * NT < 20 * HW ? p.w_v1[tid + u * NT] : 0.f;
#pragma unroll
    for (int u = 0; u < 9; ++u)
      if (tid + u * NT < 20 * HW) wv1[tid + u * NT] = tw[u];
  }
  if (tid < HB) orow[tid] = tid < nbrd ? rowlist[base + i0 + tid] : 0;
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  // ---- policy logits: wave -> 32 actions, lane -> (board lane & 31 of the A operand | action lane & 31 of the B operand, k half)
  if (wave * 32 < A) {
    const int i = lane & 31, h = lane >> 5, a = wave * 32 + i;
    const bool av = a < A;
    const int nq = HW >> 2, rem = HW & 3;
    const int npair = (nq + 1) / 2;  // steps of two quads (one per k half)
    const size_t half = ((size_t)HW * A + 3) & ~(size_t)3;
    constexpr int U = 4;  // pairs per group: the next group's operands are requested under the current group's 16 MFMAs
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int pl = 0; pl < 2; ++pl) {
      const float* frow = F + i * HFS + (1 + pl) * HPL;
      const float* wh = p.w_pT + pl * half;
      const float4* wq = reinterpret_cast<const float4*>(wh) + (av ? a : 0);
      float4 bw[U], fa[U], bn[U], fn[U];
      auto fetch = [&](int g, float4* b_, float4* f_) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int q = 2 * (g * U + u) + h;  // this k half's quad of cells
          const bool qv = q < nq && g * U + u < npair;
          b_[u] = (qv && av) ? wq[(size_t)q * A] : make_float4(0.f, 0.f, 0.f, 0.f);
          f_[u] = qv ? *reinterpret_cast<const float4*>(frow + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      };
      const int ngrp = (npair + U - 1) / U;
      fetch(0, bn, fn);
      for (int g = 0; g < ngrp; ++g) {
#pragma unroll
        for (int u = 0; u < U; ++u) { bw[u] = bn[u]; fa[u] = fn[u]; }
        if (g + 1 < ngrp) fetch(g + 1, bn, fn);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u].x, bw[u].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u].y, bw[u].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u].z, bw[u].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u].w, bw[u].w, acc, 0, 0, 0);
        }
      }
      const float* wr = wh + (size_t)nq * A * 4;
      for (int r = 0; r < rem; ++r) {  // the last HW % 4 cells: the lower k half carries them, the upper one zeros
        const float fr = h == 0 ? frow[4 * nq + r] : 0.f;
        const float br = (h == 0 && av) ? wr[(size_t)r * A + a] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fr, br, acc, 0, 0, 0);
      }
    }
    // C[board 8 (r >> 2) + 4 h + (r & 3)][action lane & 31]
    if (av) {
      const float bias = p.b_p[a];
#pragma unroll
      for (int r = 0; r < 16; ++r) logit[(8 * (r >> 2) + 4 * h + (r & 3)) * HLG + a] = acc[r] + bias;
    }
  }
  // ---- value head, first layer: Linear(HW,20) + LeakyReLU, one (board, unit) per thread, the chain in cell order
  for (int k = tid; k < nbrd * 20; k += NT) {
    const int bi = k / 20, u = k - bi * 20;
    float s = p.b_v1[u];
    const float* w = wv1 + u * HW;
    const float* f = F + bi * HFS;
#pragma unroll 8
    for (int c = 0; c < HW; ++c) s = fmaf(f[c], w[c], s);
    hid[k] = leaky(s, slope);
  }
  __syncthreads();
  // ---- Linear(20,1) + tanh; the softmax: sixteen threads per board
  if (tid < nbrd) {
    float s = p.b_v2[0];
    for (int u = 0; u < 20; ++u) s = fmaf(hid[tid * 20 + u], p.w_v2[u], s);
    values[orow[tid]] = tanhf(s);
  }
  const int bi = tid >> 4, j = tid & 15;
  float* lg = logit + bi * HLG;
  float mx = -3.4e38f;
  for (int a = j; a < A; a += 16) mx = fmaxf(mx, lg[a]);
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 16));
  float ps = 0.f;
  for (int a = j; a < A; a += 16) {  // exp(logit - max) in place, the thread's strided share summed in action order
    const float e = expf(lg[a] - mx);
    lg[a] = e;
    ps += e;
  }
  part[tid] = ps;
  __syncthreads();
  if (j == 0) {
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) sum += part[bi * 16 + q];
    rsum[bi] = sum;
  }
  __syncthreads();
  if (bi < nbrd) {
    const float sum = rsum[bi];
    float* out = probs + (size_t)orow[bi] * A;
    for (int a = j; a < A; a += 16) out[a] = lg[a] / sum;
  }
}


// ===================================================================================================
// Table evaluator (caro_net_create_hash): priors and value are exact dyadic float32 functions of a 64-bit
// hash of the leaf's planes (the arithmetic is integer only, so every implementation of the definition in
// include/caro_hip.h gives the same bits).  It takes the place of the conv net wherever the search itself is to
// be checked bit for bit: same launch interface, leaf counts read on device, dense or slot rows.
// One wavefront per row; 4 rows per workgroup.
__global__ __launch_bounds__(256) void k_net_hash(int HW2, int A, unsigned long long salt0, unsigned long long salt1,
                                                  const float* __restrict__ planes,
                                                  const int32_t* __restrict__ counts, int which, int row1,
                                                  float* __restrict__ probs, float* __restrict__ values,
                                                  const int32_t* __restrict__ gpack, int gG, int gB) {
  const int lane = threadIdx.x & 63;
  const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  long long row;
  int cls = 0;
  if (gpack) {  // slot rows: row g * B + j is live when j < count of game g
    const int g = (int)(w / gB), j = (int)(w - (long long)g * gB);
    if (g >= gG) return;
    const int v = gpack[g];
    if (j >= (v & 0xFF)) return;
    cls = v >> 8;
    row = w;
  } else if (which < 2) {
    if (w >= counts[which]) return;
    row = w + (which ? counts[0] : 0);
    cls = 0;  // a single net: its own salt is salt0
  } else {
    const int L0 = counts[0], L1 = counts[1];
    if (w < L0) { row = w; cls = 0; }
    else if (w < L0 + L1) { row = (row1 >= 0 ? row1 : L0) + (w - L0); cls = 1; }
    else return;
  }
  const float* pl = planes + (size_t)row * HW2;
  unsigned long long h = 0;
  for (int j = lane; j < HW2; j += 64)
    if (pl[j] != 0.0f) h += caro_mix64(0x5851f42d4c957f2dULL + (unsigned long long)j) | 1ULL;
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) h += __shfl_xor(h, m, 64);
  h += cls ? salt1 : salt0;
  for (int a = lane; a < A; a += 64) {
    const unsigned long long ha = caro_mix64(h + 0x9E3779B97F4A7C15ULL * (unsigned long long)(a + 1));
    probs[(size_t)row * A + a] = (float)(((ha >> 20) & 1023ULL) + 1ULL) / 8192.0f;
  }
  if (lane == 0) {
    const unsigned long long hv = caro_mix64(h ^ 0xA5A5A5A5A5A5A5A5ULL);
    values[row] = (float)((long long)((hv >> 20) % 2001ULL) - 1000LL) / 1024.0f;
  }
}

}  // namespace cnet

// =================================================================== host side
extern "C" void caro__set_error(const char* msg);  // caro_engine.hip

struct caro_net {
  int kind;            // 0: conv net (lib/model.py), 1: table evaluator (k_net_hash)
  uint64_t salt;       // table evaluator
  cnet::NetParams p;
  float* dev;
  float* ww_dev;  // transformed residual weights (f32w mode), or null
  uint32_t* wtab_dev;  // tile table (f32w mode), or null
  float* ww2_dev;      // 2-D Winograd transformed residual weights (f32w2 mode), or null
  float* wpT_dev;      // policy matrix transposed, or null
  uint16_t* wx3_dev;   // split bfloat16 residual weights (bf16x3 mode), or null
  // f32w2 mode: the feature rows [rows][3][HW] that travel from k_net_forward_w2 to k_net_heads and the output row of
  // every dense board -- one set per stream the handle is launched on (launches on different streams may overlap)
  struct HeadRows { void* stream; float* feat; int32_t* rowl; int64_t rows; uint64_t used; } hrows[8];
  int n_hrows;
  uint64_t hrows_clock;  // launches through the table so far: `used` of a slot = the clock of its last launch (LRU)
  int64_t hrows_evictions;  // times a slot of the table changed its stream (caro_net_stream_evictions)
  int device;
  unsigned long long* dbg_stamps;  // diagnostic (caro_net_debug_stamps): per-workgroup stamps of the slot launches too
};

static int nfail(int code, const std::string& m) {
  caro__set_error(m.c_str());
  return code;
}

// workgroups of a launch over at most max_rows rows: full tiles, or (f32w, overflow by a little) ncu full tiles +
// the overflow in split tiles -- never more than 2 * ncu of those, see k_net_forward_w
static unsigned net_grid(const caro_net* n, int64_t max_rows) {
  int64_t grid = (max_rows + n->p.TB - 1) / n->p.TB;
  if (n->p.ww && n->p.ncu > 0) {  // the kernel picks the tile size from L <= max_rows: cover every choice it can make
    const int64_t ncu = n->p.ncu, full = ncu * n->p.TB;
    const int tbs[2] = {n->p.TB4, n->p.TB2};
    for (int tb : tbs) {
      if (tb <= 0) continue;
      int64_t small = (max_rows + tb - 1) / tb;  // every tile small (L <= ncu * tb)
      if (small > ncu) small = ncu;
      if (small > grid) grid = small;
      if (max_rows > full) {                     // ncu full tiles + the overflow in small tiles
        int64_t split = ncu + (max_rows - full + tb - 1) / tb;
        if (split > 2 * ncu) split = 2 * ncu;
        if (split > grid) grid = split;
      }
    }
  }
  return (unsigned)grid;
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
  n->kind = 0;
  n->salt = 0;
  n->device = device_id;
  n->ww_dev = nullptr;
  n->wtab_dev = nullptr;
  n->ww2_dev = nullptr;
  n->wpT_dev = nullptr;
  n->wx3_dev = nullptr;
  n->n_hrows = 0;
  n->hrows_clock = 0;
  n->hrows_evictions = 0;
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
  p.ww = nullptr;
  p.wtab = nullptr;
  p.ww2 = nullptr;
  p.w_pT = nullptr;
  p.wx3 = nullptr;
  p.ncu = 0; p.TB2 = 0; p.TB4 = 0;
  if (cnet::head_span_host(HW, A) > cnet::HEAD_STAGE_MAX) {  // the policy matrix column-major for the large-board heads
    // per plane: [cell / 4][A][4], then the last HW % 4 cells as [cell][A]; a plane's image starts on a 16-byte boundary
    const size_t half = ((size_t)HW * A + 3) & ~(size_t)3, np = 2 * half;
    const float* wp_host = packed_host + (p.w_p - n->dev);
    std::vector<float> t(np, 0.f);
    const int nq = HW / 4;
    for (int a = 0; a < A; ++a)
      for (int pl = 0; pl < 2; ++pl)
        for (int c = 0; c < HW; ++c) {
          const float w = wp_host[(size_t)a * 2 * HW + pl * HW + c];
          const size_t at = c < 4 * nq ? ((size_t)(c >> 2) * A + a) * 4 + (c & 3) : (size_t)nq * A * 4 + (size_t)(c - 4 * nq) * A + a;
          t[pl * half + at] = w;
        }
    if (hipMalloc((void**)&n->wpT_dev, np * sizeof(float)) != hipSuccess ||
        hipMemcpy(n->wpT_dev, t.data(), np * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
      if (n->wpT_dev) (void)hipFree(n->wpT_dev);
      (void)hipFree(n->dev);
      delete n;
      return nfail(CARO_E_NOMEM, "hipMalloc / hipMemcpy failed");
    }
    p.w_pT = n->wpT_dev;
  }
  *out = n;
  return 0;
}

/* f32w mode: upload the row-Winograd F(2,3) transformed residual weights ([5][4 p][3 dx] chunks of 4096 floats in
 * the LDS image order of the plain residual weights, packed by caro_ai_amd/net_hip.py:pack_net_w); from then on the
 * forward calls of this net run k_net_forward_w: float32 MFMA like the default, two thirds of the multiplies.
 * The boards-per-workgroup figure may shrink (128 tiles of ceil(H/2) x W per workgroup). */
int caro_net_enable_winograd(caro_net* n, const float* ww_host, int64_t n_floats) {
  if (!n || !ww_host) return nfail(CARO_E_INVAL, "null argument");
  if (n->kind != 0) return nfail(CARO_E_STATE, "not a conv net");
  if (n->p.ww2 || n->p.wx3) return nfail(CARO_E_STATE, "net is already in another arithmetic mode");
  const int64_t want = (int64_t)cnet::WTAPS * cnet::WCHUNK;
  if (n_floats != want) return nfail(CARO_E_INVAL, "transformed weight image has the wrong size");
  if (hipSetDevice(n->device) != hipSuccess) return nfail(CARO_E_HIP, "hipSetDevice failed");
  if (!n->ww_dev && hipMalloc((void**)&n->ww_dev, want * sizeof(float)) != hipSuccess)
    return nfail(CARO_E_NOMEM, "hipMalloc failed");
  {
    // The kernel streams the weights in chunks of (layer, dx, granule half): [p][h][co][4 granules of 4 k] with the
    // granule index XORed by (co >> 2) & 3 (trunk_w); the caller's image is [layer][p][dx][h][co][8 granules of 4 k]
    // with the granule index XORed by (co >> 1) & 7 (the order of the plain residual weights).
    std::vector<float> img((size_t)want);
    for (int layer = 0; layer < cnet::NRES; ++layer)
      for (int dx = 0; dx < 3; ++dx)
        for (int hq = 0; hq < 2; ++hq)
          for (int pp = 0; pp < 4; ++pp)
            for (int h = 0; h < 2; ++h)
              for (int co = 0; co < 64; ++co)
                for (int g = 0; g < 4; ++g) {
                  const float* src = ww_host + ((size_t)(layer * 4 + pp) * 3 + dx) * cnet::WCHUNK + (h * 64 + co) * 32 +
                                     (((hq * 4 + g) ^ ((co >> 1) & 7)) << 2);
                  float* dst = img.data() + ((size_t)(layer * 3 + dx) * 2 + hq) * cnet::WCH +
                               ((pp * 2 + h) * 64 + co) * 16 + ((g ^ ((co >> 2) & 3)) << 2);
                  for (int j = 0; j < 4; ++j) dst[j] = src[j];
                }
    if (hipMemcpy(n->ww_dev, img.data(), want * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
      return nfail(CARO_E_HIP, "hipMemcpy failed");
  }
  const int H = n->p.H, W = n->p.W, HW = n->p.HW;
  const int tpb = ((H + 1) / 2) * W;
  if (tpb > 128) return nfail(CARO_E_INVAL, "board too large for the 128-tile workgroup");
  const int tbw = 128 / tpb;
  if (tbw < n->p.TB) n->p.TB = tbw;
  // tile -> (row tile, lane) table.  ds_read_b128 lane groups (MI355X_MICROARCH.md, LDS): within a 32-lane half
  // {0-3,12-15,20-27} and {4-11,16-19,28-31}; 4 row tiles x 2 groups = 8 groups of 16 MFMA rows.  A tile's
  // activation rows are base + const with base = board*HW + 2*ty*W + x, so a group is conflict-free when its
  // bases differ mod 16: residues are dealt to the groups most frequent first, what does not fit goes anywhere.
  static const int kLanes[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                    {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
  std::vector<uint32_t> by_res[16], group[8], left;
  bool has[8][16] = {};
  for (int bi = 0; bi < n->p.TB; ++bi)
    for (int ty = 0; ty < (H + 1) / 2; ++ty)
      for (int x = 0; x < W; ++x)
        by_res[(bi * HW + 2 * ty * W + x) & 15].push_back((uint32_t)bi | (uint32_t)ty << 8 | (uint32_t)x << 16 | 1u << 24);
  int order[16];
  for (int r = 0; r < 16; ++r) order[r] = r;
  std::sort(order, order + 16, [&](int a, int b) { return by_res[a].size() > by_res[b].size(); });
  for (int oi = 0; oi < 16; ++oi) {
    const int r = order[oi];
    for (uint32_t t : by_res[r]) {
      int best = -1;
      for (int g = 0; g < 8; ++g)
        if (!has[g][r] && group[g].size() < 16 && (best < 0 || group[g].size() < group[best].size())) best = g;
      if (best < 0) {
        left.push_back(t);
      } else {
        has[best][r] = true;
        group[best].push_back(t);
      }
    }
  }
  for (uint32_t t : left) {
    int best = 0;
    for (int g = 1; g < 8; ++g)
      if (group[g].size() < group[best].size()) best = g;
    group[best].push_back(t);
  }
  uint32_t tab[128] = {};
  for (int g = 0; g < 8; ++g)
    for (size_t k = 0; k < group[g].size(); ++k) tab[(g >> 1) * 32 + kLanes[g & 1][k]] = group[g][k];
  if (!n->wtab_dev && hipMalloc((void**)&n->wtab_dev, sizeof(tab)) != hipSuccess) return nfail(CARO_E_NOMEM, "hipMalloc failed");
  if (hipMemcpy(n->wtab_dev, tab, sizeof(tab), hipMemcpyHostToDevice) != hipSuccess) return nfail(CARO_E_HIP, "hipMemcpy failed");
  n->p.wtab = n->wtab_dev;
  n->p.ww = n->ww_dev;
  // overflow tiles (k_net_forward_w): 64 / 32 GEMM rows with the K loop split over 2 / 4 waves
  int ncu = 0;
  if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, n->device) != hipSuccess) ncu = 0;
  const char* off = getenv("CARO_NO_SPLIT_TILES");  // A/B measurements
  if (off && off[0] == '1') ncu = 0;
  n->p.ncu = ncu;
  n->p.TB2 = 64 / tpb < n->p.TB ? 64 / tpb : 0;
  // the 2-way tiles exchange their partial sums through activation rows XROW2..254: the tile's own rows must end below
  // (boards with an even height whose tile count divides 64 -- 8x8 -- fill 128 rows: no 2-way tiles for them)
  if (n->p.TB2 * H * W > cnet::XROW2) n->p.TB2 = 0;
  n->p.TB4 = 32 / tpb < n->p.TB ? 32 / tpb : 0;
  if (n->p.TB4 * H * W > cnet::XROW4) n->p.TB4 = 0;  // likewise the 4-way tiles: rows XROW4..254
  return 0;
}

/* f32w2 mode (large boards): upload the 2-D Winograd F(2x2,3x3) transformed residual weights, already in the LDS image
 * order of trunk_w2d -- [5 layers][8 chunks = phase * 4 + granule pair][2 b of the phase][4 a][2 h][64 co][8 floats],
 * packed by caro_ai_amd/net_hip.py:pack_net_w2 -- from then on the forward calls of this net run k_net_forward_w2.
 * One board per workgroup, its 2x2-output tiles in two 32-row MFMA blocks: boards of 12x12 .. 15x15 cells. */
int caro_net_winograd2d_size(void) { return cnet::W2NCHUNK * cnet::WCH; }
int caro_net_winograd2d_supported(int H, int W) {
  // H*W <= HEADS_MAX_CELLS: k_net_heads stages HPL floats per feature plane and keeps 20 * 225 value-head weights
  // in LDS -- a 15x16 board (240 cells, 8 x 8 tiles) would pass the tile test and overrun both
  return (H * W >= 128 && H * W <= cnet::HEADS_MAX_CELLS && ((H + 1) / 2) <= 8 && ((W + 1) / 2) <= 8) ? 1 : 0;
}
int caro_net_enable_winograd2d(caro_net* n, const float* ww2_host, int64_t n_floats) {
  if (!n || !ww2_host) return nfail(CARO_E_INVAL, "null argument");
  if (n->kind != 0) return nfail(CARO_E_STATE, "not a conv net");
  if (n->p.ww || n->p.wx3) return nfail(CARO_E_STATE, "net is already in another arithmetic mode");
  if (!caro_net_winograd2d_supported(n->p.H, n->p.W))
    return nfail(CARO_E_INVAL, "2-D Winograd form: boards of one per workgroup with at most 8 x 8 tiles (12x12 .. 15x15)");
  const int64_t want = (int64_t)cnet::W2NCHUNK * cnet::WCH;
  if (n_floats != want) return nfail(CARO_E_INVAL, "transformed weight image has the wrong size");
  if (!n->p.w_pT) return nfail(CARO_E_STATE, "2-D Winograd form: the batched heads need the transposed policy matrix");
  if (hipSetDevice(n->device) != hipSuccess) return nfail(CARO_E_HIP, "hipSetDevice failed");
  if (!n->ww2_dev && hipMalloc((void**)&n->ww2_dev, want * sizeof(float)) != hipSuccess)
    return nfail(CARO_E_NOMEM, "hipMalloc failed");
  if (hipMemcpy(n->ww2_dev, ww2_host, want * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
    return nfail(CARO_E_HIP, "hipMemcpy failed");
  n->p.TB = 1;
  n->p.ww2 = n->ww2_dev;
  return 0;
}

/* bf16x3 mode (an extra arithmetic mode, see k_net_forward_x3): upload the residual weights split into three bfloat16
 * parts, [45 (layer, tap)][2 c][3 parts][4 kg][64 co][8 ci] uint16 with ci = 32 c + 8 kg + 0..7, packed by
 * caro_ai_amd/net_hip.py:pack_net_x3; from then on the forward calls of this net run k_net_forward_x3. */
int64_t caro_net_split_bf16_size(void) { return (int64_t)cnet::NTAPS * cnet::X3_TAP_U4 * 8; }
int caro_net_enable_split_bf16(caro_net* n, const uint16_t* parts_host, int64_t n_u16) {
  if (!n || !parts_host) return nfail(CARO_E_INVAL, "null argument");
  if (n->kind != 0) return nfail(CARO_E_STATE, "not a conv net");
  if (n->p.ww || n->p.ww2) return nfail(CARO_E_STATE, "net is already in another arithmetic mode");
  const int64_t want = caro_net_split_bf16_size();
  if (n_u16 != want) return nfail(CARO_E_INVAL, "split weight image has the wrong size");
  if (hipSetDevice(n->device) != hipSuccess) return nfail(CARO_E_HIP, "hipSetDevice failed");
  const size_t pad = (size_t)cnet::X3_PAD_TAPS * cnet::X3_TAP_U4 * 8;  // the kernel's staging loads run two taps ahead
  if (!n->wx3_dev) {
    if (hipMalloc((void**)&n->wx3_dev, (want + pad) * sizeof(uint16_t)) != hipSuccess) return nfail(CARO_E_NOMEM, "hipMalloc failed");
    if (hipMemset(n->wx3_dev + want, 0, pad * sizeof(uint16_t)) != hipSuccess) return nfail(CARO_E_HIP, "hipMemset failed");
  }
  if (hipMemcpy(n->wx3_dev, parts_host, want * sizeof(uint16_t), hipMemcpyHostToDevice) != hipSuccess)
    return nfail(CARO_E_HIP, "hipMemcpy failed");
  n->p.wx3 = n->wx3_dev;
  return 0;
}

int64_t caro_net_stream_evictions(const caro_net* n) { return n ? n->hrows_evictions : 0; }

void caro_net_destroy(caro_net* n) {
  if (!n) return;
  if (n->ww_dev) (void)hipFree(n->ww_dev);
  if (n->wtab_dev) (void)hipFree(n->wtab_dev);
  if (n->ww2_dev) (void)hipFree(n->ww2_dev);
  if (n->wpT_dev) (void)hipFree(n->wpT_dev);
  if (n->wx3_dev) (void)hipFree(n->wx3_dev);
  for (int k = 0; k < n->n_hrows; ++k) {
    if (n->hrows[k].feat) (void)hipFree(n->hrows[k].feat);
    if (n->hrows[k].rowl) (void)hipFree(n->hrows[k].rowl);
  }
  if (n->dev) (void)hipFree(n->dev);
  delete n;
}

/* table evaluator: see include/caro_hip.h for the definition of P and v */
int caro_net_create_hash(int H, int W, int A, uint64_t salt, int device_id, caro_net** out) {
  if (!out) return nfail(CARO_E_INVAL, "null argument");
  if (H < 2 || W < 2 || H > 15 || W > 15 || A < 1 || A > 255) return nfail(CARO_E_INVAL, "unsupported board / action count");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return nfail(CARO_E_NODEV, "no HIP device: libcaro_hip needs a GPU (there is no CPU fallback)");
  if (device_id < 0 || device_id >= ndev) return nfail(CARO_E_INVAL, "device_id out of range");
  caro_net* n = new caro_net();
  memset(&n->p, 0, sizeof n->p);
  n->kind = 1;
  n->salt = salt;
  n->device = device_id;
  n->dev = nullptr;
  n->ww_dev = nullptr;
  n->wtab_dev = nullptr;
  n->ww2_dev = nullptr;
  n->wpT_dev = nullptr;
  n->wx3_dev = nullptr;
  n->n_hrows = 0;
  n->hrows_clock = 0;
  n->hrows_evictions = 0;
  n->p.H = H; n->p.W = W; n->p.HW = H * W; n->p.A = A; n->p.TB = 4;
  *out = n;
  return 0;
}

// The feature rows of a launch whose FC heads follow in k_net_heads, per (net handle, stream): the table slot of `stream`
// (least recently used one re-keyed when the table is full), grown to `need` rows.
static int head_rows_for(caro_net* n0, void* stream, int64_t need, caro_net::HeadRows** out) {
  caro_net::HeadRows* hr = nullptr;
  for (int k = 0; k < n0->n_hrows; ++k)
    if (n0->hrows[k].stream == stream) hr = &n0->hrows[k];
  if (!hr) {
    if (n0->n_hrows == 8) {
      // table full (a long-lived net launched on transient streams: engines recreated per iteration): the least
      // recently used slot goes.
      hr = &n0->hrows[0];
      for (int k = 1; k < 8; ++k)
        if (n0->hrows[k].used < hr->used) hr = &n0->hrows[k];
      // The slot's buffers are KEPT and only re-keyed (they are re-allocated below if the new stream needs more
      // rows): one device synchronisation -- launches of the old stream may still read them -- instead of a
      // synchronising hipFree pair + two hipMallocs.  A net used round-robin on more than 8 streams pays this on
      // every launch: caro_net_stream_evictions() counts them.
      if (hipDeviceSynchronize() != hipSuccess) return nfail(CARO_E_HIP, "hipDeviceSynchronize failed");
      n0->hrows_evictions += 1;
      hr->stream = stream;
    } else {
      hr = &n0->hrows[n0->n_hrows++];
      hr->stream = stream; hr->feat = nullptr; hr->rowl = nullptr; hr->rows = 0;
    }
  }
  hr->used = ++n0->hrows_clock;
  if (hr->rows < need) {
    if (hr->feat) (void)hipFree(hr->feat);
    if (hr->rowl) (void)hipFree(hr->rowl);
    hr->feat = nullptr; hr->rowl = nullptr; hr->rows = 0;
    if (hipMalloc((void**)&hr->feat, (size_t)need * 3 * n0->p.HW * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&hr->rowl, (size_t)need * sizeof(int32_t)) != hipSuccess)
      return nfail(CARO_E_NOMEM, "hipMalloc of the head feature rows failed");
    hr->rows = need;
  }
  *out = hr;
  return 0;
}

// one launch of whichever kernel serves this pair of nets.  which 0 / 1: net n0 on its class' rows; 2: both.
// gpack != null: slot rows (caro_net_forward_slots), otherwise dense rows.
static int net_launch(caro_net* n0, caro_net* n1, const float* planes_dev, const int32_t* counts_dev, int which,
                      int row1, int64_t max_rows, float* probs_dev, float* values_dev, const int32_t* gpack, int G,
                      int B, unsigned long long* stamps, void* stream, const int32_t* slist = nullptr) {
  hipStream_t st = (hipStream_t)stream;
  if (n0->kind == 1) {
    const int64_t waves = gpack ? (int64_t)G * B : max_rows;
    const unsigned grid = (unsigned)((waves + 3) / 4);
    hipLaunchKernelGGL(cnet::k_net_hash, dim3(grid), dim3(256), 0, st, 2 * n0->p.HW, n0->p.A,
                       (unsigned long long)n0->salt, (unsigned long long)n1->salt, planes_dev, counts_dev, which, row1,
                       probs_dev, values_dev, gpack, G, B);
  } else {
    const unsigned grid = net_grid(n0, max_rows) + (which == 2 ? 1u : 0u);  // +1: each class rounds up
    if (n0->p.ww2) {
      // the trunk launch leaves every board's three feature planes in n0's feature buffer; the FC heads of the whole
      // launch follow, 32 boards per workgroup (k_net_heads).  The buffer grows to the largest launch seen (first call).
      caro_net::HeadRows* hr = nullptr;
      if (int rc = head_rows_for(n0, stream, max_rows + (row1 > 0 ? row1 : 0) + 64, &hr)) return rc;
      hipLaunchKernelGGL(cnet::k_net_forward_w2, dim3(grid), dim3(cnet::NT), 0, st, n0->p, n1->p, planes_dev,
                         counts_dev, which, row1, probs_dev, values_dev, stamps ? stamps : n0->dbg_stamps, gpack, G, B,
                         hr->feat, hr->rowl, slist);
      if (hipGetLastError() != hipSuccess) return nfail(CARO_E_HIP, "net kernel launch failed");
      const unsigned hgrid = (unsigned)((max_rows + cnet::HB - 1) / cnet::HB) + (which == 2 ? 1u : 0u);
      hipLaunchKernelGGL(cnet::k_net_heads, dim3(hgrid), dim3(cnet::NT), 0, st, n0->p, n1->p, counts_dev, which, row1,
                         hr->feat, hr->rowl, probs_dev, values_dev);
    }
    else if (n0->p.ww)
      hipLaunchKernelGGL(cnet::k_net_forward_w, dim3(grid), dim3(cnet::NT), 0, st, n0->p, n1->p, planes_dev,
                         counts_dev, which, row1, probs_dev, values_dev, stamps ? stamps : n0->dbg_stamps, gpack, G, B);
    else if (n0->p.wx3) {
      // one board per workgroup (12x12 .. 15x15) and the transposed policy matrix at hand: the FC heads of the whole
      // launch follow in k_net_heads, as for the 2-D Winograd form
      const bool batched = n0->p.TB == 1 && n0->p.w_pT && n0->p.HW <= cnet::HEADS_MAX_CELLS;
      caro_net::HeadRows* hr = nullptr;
      if (batched)
        if (int rc = head_rows_for(n0, stream, max_rows + (row1 > 0 ? row1 : 0) + 64, &hr)) return rc;
      hipLaunchKernelGGL(cnet::k_net_forward_x3, dim3(grid), dim3(cnet::NT), 0, st, n0->p, n1->p, planes_dev,
                         counts_dev, which, row1, probs_dev, values_dev, stamps, gpack, G, B, hr ? hr->feat : nullptr,
                         hr ? hr->rowl : nullptr);
      if (hr) {
        if (hipGetLastError() != hipSuccess) return nfail(CARO_E_HIP, "net kernel launch failed");
        const unsigned hgrid = (unsigned)((max_rows + cnet::HB - 1) / cnet::HB) + (which == 2 ? 1u : 0u);
        hipLaunchKernelGGL(cnet::k_net_heads, dim3(hgrid), dim3(cnet::NT), 0, st, n0->p, n1->p, counts_dev, which, row1,
                           hr->feat, hr->rowl, probs_dev, values_dev);
      }
    }
    else
      hipLaunchKernelGGL(cnet::k_net_forward, dim3(grid), dim3(cnet::NT), 0, st, n0->p, n1->p, planes_dev,
                         counts_dev, which, row1, probs_dev, values_dev, stamps, gpack, G, B);
  }
  if (hipGetLastError() != hipSuccess) return nfail(CARO_E_HIP, "net kernel launch failed");
  return 0;
}
static int pair_ok(const caro_net* n0, const caro_net* n1) {
  if (n0->p.H != n1->p.H || n0->p.W != n1->p.W || n0->p.A != n1->p.A) return nfail(CARO_E_INVAL, "nets differ in shape");
  if (n0->kind != n1->kind || 
      (n0->p.ww == nullptr) != (n1->p.ww == nullptr) || (n0->p.ww2 == nullptr) != (n1->p.ww2 == nullptr) ||
      (n0->p.wx3 == nullptr) != (n1->p.wx3 == nullptr))
    return nfail(CARO_E_INVAL, "nets differ in kind / arithmetic mode");
  return 0;
}

int caro_net_boards_per_workgroup(const caro_net* n) { return n ? n->p.TB : 0; }

int caro_net_forward(caro_net* n, const float* planes_dev, const int32_t* counts_dev, int which, int64_t max_rows,
                     float* probs_dev, float* values_dev, void* stream) {
  if (!n || !planes_dev || !counts_dev || !probs_dev || !values_dev) return nfail(CARO_E_INVAL, "null argument");
  if (which != 0 && which != 1) return nfail(CARO_E_INVAL, "which must be 0 or 1");
  if (max_rows <= 0) return 0;
  return net_launch(n, n, planes_dev, counts_dev, which, -1, max_rows, probs_dev, values_dev, nullptr, 0, 0, nullptr,
                    stream);
}

/* both nets of an arena in one launch: rows [0, L0) through n0; n1's rows start at row1_base, or at L0 when
 * row1_base < 0 */
int caro_net_forward_pair_at(caro_net* n0, caro_net* n1, const float* planes_dev, const int32_t* counts_dev,
                             int64_t row1_base, int64_t max_rows, float* probs_dev, float* values_dev, void* stream) {
  if (!n0 || !n1 || !planes_dev || !counts_dev || !probs_dev || !values_dev) return nfail(CARO_E_INVAL, "null argument");
  if (int rc = pair_ok(n0, n1)) return rc;
  if (max_rows <= 0) return 0;
  return net_launch(n0, n1, planes_dev, counts_dev, 2, (int)row1_base, max_rows, probs_dev, values_dev, nullptr, 0, 0,
                    nullptr, stream);
}

int caro_net_forward_pair(caro_net* n0, caro_net* n1, const float* planes_dev, const int32_t* counts_dev,
                          int64_t max_rows, float* probs_dev, float* values_dev, void* stream) {
  return caro_net_forward_pair_at(n0, n1, planes_dev, counts_dev, -1, max_rows, probs_dev, values_dev, stream);
}

/* slot rows (the fused tree kernel): game g's j-th unique leaf sits at row g * batch + j of planes / probs / values,
 * gpack_dev[g] = count | net class << 8, counts_dev = {L0, L1} totals.  n1 == NULL: one net (every game class 0). */
int caro_net_forward_slots(caro_net* n0, caro_net* n1, const float* planes_dev, const int32_t* counts_dev,
                           const int32_t* gpack_dev, int n_games, int batch, float* probs_dev, float* values_dev,
                           void* stream) {
  if (!n0 || !planes_dev || !counts_dev || !gpack_dev || !probs_dev || !values_dev)
    return nfail(CARO_E_INVAL, "null argument");
  if (n_games < 1 || batch < 1 || batch > 255) return nfail(CARO_E_INVAL, "bad slot geometry");
  if (n1)
    if (int rc = pair_ok(n0, n1)) return rc;
  return net_launch(n0, n1 ? n1 : n0, planes_dev, counts_dev, n1 ? 2 : 0, -1, (int64_t)n_games * batch, probs_dev,
                    values_dev, gpack_dev, n_games, batch, nullptr, stream);
}

/* caro_net_forward_slots with the dense order of the leaves GIVEN: slot_list_dev i32[2][n_games * batch], entry
 * [c][i] = slot row of the i-th leaf of net class c, i < counts_dev[c], in ANY order (the fused multi-wave tree kernel
 * appends each game's rows when the game's block gets there).  Used by the kernels that serve one board per workgroup
 * (the 2-D Winograd form: a board's arithmetic does not depend on its dense index); the other forms ignore the list and
 * keep the game-order map they compute from gpack_dev -- their tiles hold several boards, and which boards share a tile
 * must stay a function of the games' states.  slot_list_dev == NULL: caro_net_forward_slots. */
int caro_net_forward_slot_list(caro_net* n0, caro_net* n1, const float* planes_dev, const int32_t* counts_dev,
                               const int32_t* gpack_dev, const int32_t* slot_list_dev, int n_games, int batch,
                               float* probs_dev, float* values_dev, void* stream) {
  if (!n0 || !planes_dev || !counts_dev || !gpack_dev || !probs_dev || !values_dev)
    return nfail(CARO_E_INVAL, "null argument");
  if (n_games < 1 || batch < 1 || batch > 255) return nfail(CARO_E_INVAL, "bad slot geometry");
  if (n1)
    if (int rc = pair_ok(n0, n1)) return rc;
  return net_launch(n0, n1 ? n1 : n0, planes_dev, counts_dev, n1 ? 2 : 0, -1, (int64_t)n_games * batch, probs_dev,
                    values_dev, gpack_dev, n_games, batch, nullptr, stream, n0->p.ww2 ? slot_list_dev : nullptr);
}

/* diagnostic: same launch, and per workgroup (total cycles, 100 MHz ticks, cycles at trunk start, at trunk end) into stamps_dev u64[4*grid] */
int caro_net_forward_stamped(caro_net* n, const float* planes_dev, const int32_t* counts_dev, int which,
                             int64_t max_rows, float* probs_dev, float* values_dev, uint64_t* stamps_dev,
                             void* stream) {
  if (!n || !stamps_dev) return nfail(CARO_E_INVAL, "null argument");
  if (n->kind != 0) return nfail(CARO_E_STATE, "stamps: conv kernels only");
  return net_launch(n, n, planes_dev, counts_dev, which, -1, max_rows, probs_dev, values_dev, nullptr, 0, 0,
                    (unsigned long long*)stamps_dev, stream);
}

/* diagnostic: every later launch of this net's float32 kernels -- the engine's slot launches included -- writes its
 * per-workgroup stamps (as caro_net_forward_stamped) to stamps_dev u64[4 * grid]; NULL switches it off */
int caro_net_debug_stamps(caro_net* n, uint64_t* stamps_dev) {
  if (!n) return nfail(CARO_E_INVAL, "null argument");
  n->dbg_stamps = (unsigned long long*)stamps_dev;
  return 0;
}

}  // extern "C"
