// caro_net_exp.h -- the switches of the net kernel's TIMING EXPERIMENTS.  NOT part of the product: caro_net.hip does not
// include it.  tools/exp/build_exp.py N turns the product source's /*@NAME(...)*/ comments into these macros in a copy
// of the file and compiles that copy with -DCARO_EXP=N into tools/exp/_build/libcaro_exp<N>.so (the results of the
// removal builds are WRONG, only the clock is read):
//   20  phase stamps of the heads (1x1 convolutions | FC stage | softmax), packed into stamp word 1
//   21  no chunk barriers in the trunk      22  no weight fetches      23  neither
#ifndef CARO_NET_EXP_H
#define CARO_NET_EXP_H

#if defined(CARO_EXP) && CARO_EXP == 20
#define CARO_HST_BEGIN(scratch, tid)                                                                              \
  unsigned long long* hst = reinterpret_cast<unsigned long long*>((scratch) + HEAD_STAGE_AT + HEAD_STAGE_MAX + 8); \
  const int hst_tid = (tid);
#define CARO_HST(n) if (hst_tid == 0) hst[n] = __builtin_amdgcn_s_memtime();
#define CARO_HST_PUBLISH(stamps, wbuf)                                                                                       \
  {                                                                                                                          \
    const unsigned long long* hst = reinterpret_cast<const unsigned long long*>((wbuf) + HEAD_STAGE_AT + HEAD_STAGE_MAX + 8); \
    (stamps)[4 * blockIdx.x + 1] = (hst[1] - hst[0]) | (hst[2] - hst[1]) << 20 | (hst[3] - hst[2]) << 40;                    \
  }
#else
#define CARO_HST_BEGIN(scratch, tid)
#define CARO_HST(n)
#define CARO_HST_PUBLISH(stamps, wbuf)
#endif

#if defined(CARO_EXP) && (CARO_EXP == 21 || CARO_EXP == 23)
#define CARO_CHUNK_BARRIER
#else
#define CARO_CHUNK_BARRIER __syncthreads();
#endif

#if defined(CARO_EXP) && (CARO_EXP == 22 || CARO_EXP == 23)
#define CARO_FETCH_ON 0
#else
#define CARO_FETCH_ON 1
#endif

// 24: per-layer stamps of the row-Winograd trunk (workgroups 0..63, thread 0): layer start | main loop done | output
//     transform done | partial sums exchanged | activations written | the layer's closing barrier passed
#if defined(CARO_EXP) && CARO_EXP == 24
__device__ unsigned long long g_caro_lst[64 * 5 * 8];
#define CARO_LST(layer, n)                                                                                     \
  if (tid == 0 && blockIdx.x < 64) g_caro_lst[(blockIdx.x * 5 + (layer)) * 8 + (n)] = __builtin_amdgcn_s_memtime();
extern "C" int caro_exp_read_lst(unsigned long long* out_host) {
  return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_caro_lst), sizeof(unsigned long long) * 64 * 5 * 8);
}
// ... and of the prologue / the heads: kernel start | leaf count known | LDS zeroed, fetches issued | conv_in weights
//     arrived | slot rows mapped | conv_in done | trunk may start || heads start | 1x1 convolutions | FC stage | end
__device__ unsigned long long g_caro_pst[64 * 16];
#define CARO_PST(n) \
  if (threadIdx.x == 0 && blockIdx.x < 64) g_caro_pst[blockIdx.x * 16 + (n)] = __builtin_amdgcn_s_memtime();
extern "C" int caro_exp_read_pst(unsigned long long* out_host) {
  return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_caro_pst), sizeof(unsigned long long) * 64 * 16);
}
#else
#define CARO_LST(layer, n)
#define CARO_PST(n)
#endif

// 25: TIMING ONLY (results wrong): the waves of row tiles 2, 3 run one barrier behind those of row tiles 0, 1 through the
//     whole trunk of a full tile -- the question is what the matrix pipe gains when one half's layer epilogue falls into
//     the other half's main loop
#if defined(CARO_EXP) && CARO_EXP == 25
#define CARO_SHIFT_BEGIN(KS_, wave_) if ((KS_) == 1 && (wave_) >= 4) __builtin_amdgcn_s_barrier();
#define CARO_SHIFT_END(KS_, wave_) if ((KS_) == 1 && (wave_) < 4) __builtin_amdgcn_s_barrier();
#else
#define CARO_SHIFT_BEGIN(KS_, wave_)
#define CARO_SHIFT_END(KS_, wave_)
#endif

#endif
