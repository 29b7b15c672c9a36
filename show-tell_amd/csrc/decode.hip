// Greedy caption decoding (reference rnn.py:37-58, LSTM/rnn_lstm.py:35-57): exactly `steps`
// iterations of  unit(x, h) -> linear -> max(1)[1] -> embeddings,  no early stop.
//
// Per step: one fused launch per layer (x W_ih^T and h W_hh^T accumulate in the same MFMA
// kernel, gates in its epilogue), one MFMA GEMM for the vocabulary projection and one
// arg-max + embedding-gather kernel.  Everything is issued from this one C call.
#include "common.h"
#include <stdlib.h>
#include "rnn_kernels.h"
#include <string.h>

namespace {

inline size_t al(size_t v) { return (v + 255) & ~(size_t)255; }
inline int up8(int v) { return (v + 7) & ~7; }

// first-maximum rule of torch.max(1): among equal maxima the lowest index wins
template <typename T>
__global__ __launch_bounds__(256) void argmax_embed_kernel(const float* __restrict__ logits, int ldl, int V,
                                                           long* __restrict__ ids, int ids_stride, int t,
                                                           const T* __restrict__ emb, T* __restrict__ x, int E) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const int row = blockIdx.x;
  const float* l = logits + (long)row * ldl;
  float best = -INFINITY; int bi = 0x7fffffff;
  for (int i = threadIdx.x; i < V; i += blockDim.x) {
    const float v = l[i];
    if (v > best || (v == best && i < bi)) { best = v; bi = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { sv[wid] = best; si[wid] = bi; }
  __syncthreads();
  best = sv[0]; bi = si[0];
  for (int w = 1; w < 4; ++w) if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
  if (bi >= V) bi = 0;   // all-NaN row: keep the index in range
  if (threadIdx.x == 0) ids[(long)row * ids_stride + t] = bi;
  if (x) {
    constexpr int N = 16 / (int)sizeof(T);
    for (int c = threadIdx.x * N; c < E; c += blockDim.x * N)
      *reinterpret_cast<u32x4*>(x + (long)row * E + c) = *reinterpret_cast<const u32x4*>(emb + (long)bi * E + c);
  }
}


// x[r] = emb[ids[r]]  (rows of E elements, 16 bytes per lane)
template <typename T>
__global__ __launch_bounds__(256) void embedding_rows_kernel(const T* __restrict__ emb, const long* __restrict__ ids, T* __restrict__ out,
                                                             int n, int E, int V, int ldo) {
  constexpr int N = 16 / (int)sizeof(T);
  const int cpr = E / N;
  const long total = (long)n * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cpr), c = (int)(i - (long)r * cpr) * N;
    long t = ids[r]; t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    *reinterpret_cast<u32x4*>(out + (long)r * ldo + c) = *reinterpret_cast<const u32x4*>(emb + t * E + c);
  }
}

// dst[l][r] = src[l][idx[r]]  (beam bookkeeping: children inherit their parent's state)
template <typename T>
__global__ __launch_bounds__(256) void gather_state_kernel(const T* __restrict__ src, const int* __restrict__ idx, T* __restrict__ dst,
                                                           int L, int n_src, int n_dst, int H) {
  constexpr int N = 16 / (int)sizeof(T);
  const int cpr = H / N;
  const long total = (long)L * n_dst * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpr) * N;
    long t = i / cpr;
    const int r = (int)(t % n_dst), l = (int)(t / n_dst);
    int s = idx[r]; s = s < 0 ? 0 : (s >= n_src ? n_src - 1 : s);
    *reinterpret_cast<u32x4*>(dst + ((long)l * n_dst + r) * H + c) = *reinterpret_cast<const u32x4*>(src + ((long)l * n_src + s) * H + c);
  }
}

// Per row: softmax probabilities (fp32) of the k largest logits, in DESCENDING order, with their indices.
// raw != 0: rank and return the raw logits instead (rnn.py:90-91 uses topk on result_state directly).
__global__ __launch_bounds__(256) void softmax_topk_kernel(const float* __restrict__ logits, int ldl, int V, int k,
                                                           float* __restrict__ top_p, long* __restrict__ top_id, int raw) {
  __shared__ float sv[4]; __shared__ int si[4]; __shared__ float ssum[4];
  __shared__ int chosen[32];
  const int row = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const float* l = logits + (long)row * ldl;
  float M = -INFINITY;
  for (int i = threadIdx.x; i < V; i += blockDim.x) M = fmaxf(M, l[i]);
  M = wave_max(M);
  if (lane == 0) sv[wid] = M;
  __syncthreads();
  M = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
  float S = 0.f;
  for (int i = threadIdx.x; i < V; i += blockDim.x) S += expf(l[i] - M);
  S = wave_sum(S);
  if (lane == 0) ssum[wid] = S;
  __syncthreads();
  S = ssum[0] + ssum[1] + ssum[2] + ssum[3];
  for (int j = 0; j < k; ++j) {
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < V; i += blockDim.x) {
      bool used = false;
      for (int q = 0; q < j; ++q) used |= (chosen[q] == i);
      const float v = l[i];
      if (!used && (v > best || (v == best && i < bi))) { best = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    __syncthreads();
    if (lane == 0) { sv[wid] = best; si[wid] = bi; }
    __syncthreads();
    best = sv[0]; bi = si[0];
    for (int w = 1; w < 4; ++w) if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
    if (bi >= V) bi = 0;
    if (threadIdx.x == 0) {
      chosen[j] = bi;
      top_id[(long)row * k + j] = bi;
      top_p[(long)row * k + j] = raw ? best : expf(best - M) / S;
    }
    __syncthreads();
  }
}


// The same result for k <= 8 in TWO passes over the row instead of 2 + k: pass 1 keeps, per thread, the maximum and a sorted
// list of its k best (value descending, index ascending: a thread meets its indices in increasing order, so strict '>' on
// insertion keeps first-index-wins); pass 2 is the sum of exponentials in exactly the order of softmax_topk_kernel (the
// probabilities are bit-identical); then k rounds of a block-wide arg-max over the threads' list heads pick the winners.
template <int KMAX>
__global__ __launch_bounds__(256) void softmax_topk_fast_kernel(const float* __restrict__ logits, int ldl, int V, int k,
                                                                float* __restrict__ top_p, long* __restrict__ top_id, int raw) {
  __shared__ float sv[4]; __shared__ int si[4]; __shared__ float ssum[4];
  const int row = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const float* l = logits + (long)row * ldl;
  float lv[KMAX]; int li[KMAX];
#pragma unroll
  for (int q = 0; q < KMAX; ++q) { lv[q] = -INFINITY; li[q] = 0x7fffffff; }
  float M = -INFINITY;
  constexpr int UN = 8;                                      // loads in flight per thread: the passes are L2-latency bound
  for (int i0 = threadIdx.x; i0 < V; i0 += UN * 256) {
    float vv[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) vv[u] = i0 + u * 256 < V ? l[i0 + u * 256] : -INFINITY;
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int i = i0 + u * 256;
      const float v = vv[u];
      if (i >= V) continue;
      M = fmaxf(M, v);
      if (v > lv[KMAX - 1] || li[KMAX - 1] == 0x7fffffff) {  // enters the list (an empty slot takes anything)
        float cv = v; int cidx = i;
#pragma unroll
        for (int q = 0; q < KMAX; ++q) {
          const bool before = cv > lv[q] || (li[q] == 0x7fffffff && cidx != 0x7fffffff);
          if (before) { const float tv = lv[q]; const int ti = li[q]; lv[q] = cv; li[q] = cidx; cv = tv; cidx = ti; }
        }
      }
    }
  }
  M = wave_max(M);
  if (lane == 0) sv[wid] = M;
  __syncthreads();
  M = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
  float S = 0.f;
  if (!raw) {
    for (int i0 = threadIdx.x; i0 < V; i0 += UN * 256) {     // same order of additions as softmax_topk_kernel
      float vv[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) vv[u] = i0 + u * 256 < V ? l[i0 + u * 256] : 0.f;
#pragma unroll
      for (int u = 0; u < UN; ++u) if (i0 + u * 256 < V) S += expf(vv[u] - M);
    }
    S = wave_sum(S);
    if (lane == 0) ssum[wid] = S;
  }
  __syncthreads();
  if (!raw) S = ssum[0] + ssum[1] + ssum[2] + ssum[3];
  for (int j = 0; j < k; ++j) {
    float best = lv[0]; int bi = li[0];                      // this thread's best remaining candidate
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    __syncthreads();
    if (lane == 0) { sv[wid] = best; si[wid] = bi; }
    __syncthreads();
    best = sv[0]; bi = si[0];
    for (int w = 1; w < 4; ++w) if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
    if (li[0] == bi && bi != 0x7fffffff) {                   // the winner's owner moves on to its next candidate
#pragma unroll
      for (int q = 0; q + 1 < KMAX; ++q) { lv[q] = lv[q + 1]; li[q] = li[q + 1]; }
      lv[KMAX - 1] = -INFINITY; li[KMAX - 1] = 0x7fffffff;
    }
    if (bi >= V) bi = 0;
    if (threadIdx.x == 0) {
      top_id[(long)row * k + j] = bi;
      top_p[(long)row * k + j] = raw ? best : expf(best - M) / S;
    }
  }
}


// The same result again, without the per-thread sorted lists (their insertion code ran for nearly every element: with 64 lanes some
// lane always had a new list entry, 47 us for 1280 x 10000 logits against 8.5 us of HBM time):
//   pass 1: every thread keeps only its MAXIMUM; the block's k-th largest thread maximum T is a lower bound of the row's k-th largest
//           element (k threads hold an element >= T), so the row's top k are among the elements >= T;
//   pass 2: the sum of exponentials in the order of softmax_topk_kernel (probabilities bit-identical) -- and the elements >= T are
//           appended to a candidate list in LDS (typically k .. 2k entries);
//   then k rounds of a block-wide arg-max over the candidates, value descending, first index wins: the reference's order.
// More than 256 candidates (rows of equal logits): the block falls back to a serial selection by one wave over the row.
__global__ __launch_bounds__(256) void softmax_topk_thr_kernel(const float* __restrict__ logits, int ldl, int V, int k,
                                                               float* __restrict__ top_p, long* __restrict__ top_id, int raw) {
  constexpr int CAP = 256, UN = 8;
  __shared__ float sv[4]; __shared__ int si[4]; __shared__ float ssum[4];
  __shared__ float cv[CAP]; __shared__ int ci[CAP]; __shared__ int cnt;
  const int row = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const float* l = logits + (long)row * ldl;
  if (threadIdx.x == 0) cnt = 0;
  // rows of up to 256 x 40 entries (V = 10000) stay in registers between the passes: the row is read ONCE
  constexpr int NK = 40;
  const bool keep = V <= 256 * NK;
  float rv[NK];
  float tm = -INFINITY;
  if (keep) {
#pragma unroll
    for (int j = 0; j < NK; ++j) { const int i = (int)threadIdx.x + 256 * j; rv[j] = i < V ? l[i] : -INFINITY; }
#pragma unroll
    for (int j = 0; j < NK; ++j) tm = fmaxf(tm, rv[j]);
  } else {
    for (int i0 = threadIdx.x; i0 < V; i0 += UN * 256) {
      float vv[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) vv[u] = i0 + u * 256 < V ? l[i0 + u * 256] : -INFINITY;
#pragma unroll
      for (int u = 0; u < UN; ++u) tm = fmaxf(tm, vv[u]);
    }
  }
  float M = wave_max(tm);
  if (lane == 0) sv[wid] = M;
  __syncthreads();
  M = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
  // T = the k-th largest thread maximum (k rounds of a block arg-max over (tm, thread); the winner retires)
  float T = -INFINITY, mine = tm;
  for (int j = 0; j < k; ++j) {
    float best = mine; int bi = (int)threadIdx.x;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    __syncthreads();
    if (lane == 0) { sv[wid] = best; si[wid] = bi; }
    __syncthreads();
    best = sv[0]; bi = si[0];
    for (int w = 1; w < 4; ++w) if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
    if ((int)threadIdx.x == bi) mine = -INFINITY;
    T = best;
  }
  float S = 0.f;
  auto visit = [&](float v, int i) {                         // (elements in the order of softmax_topk_kernel: i = thread, thread + 256, ..)
    if (!raw) S += expf(v - M);
    if (v >= T) { const int p = atomicAdd(&cnt, 1); if (p < CAP) { cv[p] = v; ci[p] = i; } }
  };
  if (keep) {
#pragma unroll
    for (int j = 0; j < NK; ++j) { const int i = (int)threadIdx.x + 256 * j; if (i < V) visit(rv[j], i); }
  } else {
    for (int i0 = threadIdx.x; i0 < V; i0 += UN * 256) {
      float vv[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) vv[u] = i0 + u * 256 < V ? l[i0 + u * 256] : -INFINITY;
#pragma unroll
      for (int u = 0; u < UN; ++u) { const int i = i0 + u * 256; if (i < V) visit(vv[u], i); }
    }
  }
  if (!raw) {
    S = wave_sum(S);
    if (lane == 0) ssum[wid] = S;
  }
  __syncthreads();
  if (!raw) S = ssum[0] + ssum[1] + ssum[2] + ssum[3];
  const int n = cnt;
  if (n > CAP) {                                             // (block-uniform) a row with > 256 elements >= T: serial selection by wave 0
    if (wid == 0) {
      int chosen[32];
      for (int j = 0; j < k; ++j) {
        float best = -INFINITY; int bi = 0x7fffffff;
        for (int i = lane; i < V; i += 64) {
          bool used = false;
          for (int q = 0; q < j; ++q) used |= (chosen[q] == i);
          const float v = l[i];
          if (!used && (v > best || (v == best && i < bi))) { best = v; bi = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
          if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (bi >= V) bi = 0;
        chosen[j] = bi;
        if (lane == 0) { top_id[(long)row * k + j] = bi; top_p[(long)row * k + j] = raw ? best : expf(best - M) / S; }
      }
    }
    return;
  }
  float myv = (int)threadIdx.x < n ? cv[threadIdx.x] : -INFINITY;
  int myi = (int)threadIdx.x < n ? ci[threadIdx.x] : 0x7fffffff;
  for (int j = 0; j < k; ++j) {
    float best = myv; int bi = myi;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    __syncthreads();
    if (lane == 0) { sv[wid] = best; si[wid] = bi; }
    __syncthreads();
    best = sv[0]; bi = si[0];
    for (int w = 1; w < 4; ++w) if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
    if (myi == bi && bi != 0x7fffffff) { myv = -INFINITY; myi = 0x7fffffff; }
    if (bi >= V) bi = 0;
    if (threadIdx.x == 0) {
      top_id[(long)row * k + j] = bi;
      top_p[(long)row * k + j] = raw ? best : expf(best - M) / S;
    }
  }
}

// Vocabulary projection with a fused running arg-max: logits[m][n] = h[m] . W[n] + b[n] are never written; per row the
// (value, first index) maximum is merged into a packed 64-bit key by atomic max.  One block = 16 rows x 128 vocabulary
// entries; a wave owns two 16-entry tiles and requests all of a tile's K fragments before its first MFMA.
template <typename T> struct MfmaD;
template <> struct MfmaD<bf16_t> {
  static constexpr int EPC = 8;
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
  }
};
template <> struct MfmaD<float> {
  static constexpr int EPC = 4;
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
#pragma unroll
    for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[e]), __uint_as_float(b[e]), c, 0, 0, 0);
  }
};

template <typename T>
__global__ __launch_bounds__(256) void vocab_argmax_kernel(const T* __restrict__ A, int lda, const T* __restrict__ W, int ldw,
                                                           const float* __restrict__ bias, int M, int N, int K,
                                                           unsigned long long* __restrict__ keys) {
  constexpr int EPC = MfmaD<T>::EPC, UNR = 16;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int r16 = lane & 15, q4 = lane >> 4;
  // 1-D grid: blocks that share a 128-entry slice of W are 8 apart (same XCD under round-robin placement), so the
  // slice is fetched into ONE L2 and re-used by the other row tiles (speed only, any placement is correct)
  const int nmt = (M + 15) / 16;
  const int grp = blockIdx.x / (8 * nmt), rem = blockIdx.x - grp * 8 * nmt;
  const int nslice = grp * 8 + (rem & 7), mt = rem >> 3;
  const int m0 = mt * 16, m = m0 + r16;
  const bool mok = m < M;
  const T* Ar = A + (long)m * lda;
  const int nsteps = (K + 4 * EPC - 1) / (4 * EPC);
  float best = -INFINITY; int bi = 0x7fffffff;
  for (int tt = 0; tt < 2; ++tt) {
    const int n0 = ((nslice * 4 + wid) * 2 + tt) * 16;
    if (n0 >= N) break;
    const int nr = n0 + r16;
    const bool nok = nr < N;
    const T* Wr = W + (long)nr * ldw;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < nsteps; s0 += UNR) {
      u32x4 fa[UNR], fw[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int k = ((s0 + u) * 4 + q4) * EPC;
        const bool kok = k < K;
        fa[u] = u32x4{0u, 0u, 0u, 0u}; fw[u] = u32x4{0u, 0u, 0u, 0u};
        if (mok && kok) fa[u] = *reinterpret_cast<const u32x4*>(Ar + k);
        if (nok && kok) fw[u] = *reinterpret_cast<const u32x4*>(Wr + k);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) MfmaD<T>::run(fw[u], fa[u], acc);
    }
    const int n = n0 + 4 * q4;      // lane: row m, entries n .. n+3
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (n + e < N) {
        const float v = acc[e] + bias[n + e];
        if (v > best || (v == best && n + e < bi)) { best = v; bi = n + e; }
      }
    }
  }
#pragma unroll
  for (int o = 16; o < 64; o <<= 1) {
    const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (q4 == 0 && mok && bi != 0x7fffffff) {
    unsigned u = __float_as_uint(best);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    atomicMax(keys + m, ((unsigned long long)u << 32) | (unsigned long long)(0xffffffffu - (unsigned)bi));
  }
}

// Same product, LDS-staged form for the shapes that fit (rows*K*sizeof(T) <= 128 KiB per pass, K <= 16 MFMA steps):
// the block stages ALL activation rows in LDS once (XOR-swizzled 16-byte chunks, conflict-free fragment reads), wave w
// keeps the weight fragments of ITS 16 vocabulary entries in registers and sweeps the row tiles from LDS.  The
// projection matrix leaves L2 once chip-wide and the activations once per block -- the fragment-direct kernel above
// re-reads both 8x through L2 (160 MB per step at B = 128, V = 10000), which is all of its 25 us.
template <typename T>
__global__ __launch_bounds__(256) void vocab_argmax_lds_kernel(const T* __restrict__ A, int lda, const T* __restrict__ W, int ldw,
                                                               const float* __restrict__ bias, int M, int N, int K,
                                                               unsigned long long* __restrict__ keys, int rows_per_pass, int ysplit) {
  constexpr int EPC = MfmaD<T>::EPC, KS = 16;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  unsigned long long* wbest = reinterpret_cast<unsigned long long*>(lds + (size_t)rows_per_pass * K * sizeof(T));   // [4][rows_per_pass]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int r16 = lane & 15, q4 = lane >> 4;
  // 1-D grid in groups of 8 * ysplit blocks: block j of a group takes vocabulary slice (group * 8 + j % 8) and row part
  // j / 8, so the ysplit blocks that read the SAME 64-entry slice of W are 8 apart = on the same XCD under round-robin
  // placement and the slice leaves HBM once, not once per row part (PMC: 18.3 MB per launch for a 10.2 MB matrix with a
  // (slices, ysplit) 2-D grid, whose row parts land on different XCDs).  Speed only: any placement is correct.
  const int grp = blockIdx.x / (8 * ysplit), j = blockIdx.x - grp * 8 * ysplit;
  const int slice = grp * 8 + (j & 7), ypart = j >> 3;
  const int n0 = (slice * 4 + wid) * 16;
  if (slice * 64 >= N) return;                          // padding blocks of the last group (whole block, before any barrier)
  const int nr = n0 + r16;
  const bool nok = nr < N;
  const T* Wr = W + (long)nr * ldw;
  u32x4 fw[KS];
#pragma unroll
  for (int u = 0; u < KS; ++u) {
    const int k = (u * 4 + q4) * EPC;
    fw[u] = u32x4{0u, 0u, 0u, 0u};
    if (nok && k < K) fw[u] = *reinterpret_cast<const u32x4*>(Wr + k);
  }
  const int n = n0 + 4 * q4;                            // lane: entries n .. n+3 of its row
  float bz[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) bz[e] = n + e < N ? bias[n + e] : 0.f;
  const int cpr = K / EPC;                              // 16-byte chunks per row (multiple of 8: host check)
  const int rowb = K * (int)sizeof(T);
  for (int mb = ypart * rows_per_pass; mb < M; mb += ysplit * rows_per_pass) {   // grid.y splits the rows when entries alone leave CUs idle
    const int rows = min(rows_per_pass, M - mb);
    __syncthreads();                                    // previous pass has left the buffer
    // 8 chunks per thread in flight: the staging pass is one or two L2 round trips, not one per chunk
    for (int i0 = threadIdx.x; i0 < rows_per_pass * cpr; i0 += 256 * 8) {
      u32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = i0 + j * 256, row = i / cpr, c = i - row * cpr;
        v[j] = u32x4{0u, 0u, 0u, 0u};
        if (i < rows_per_pass * cpr && row < rows) v[j] = *reinterpret_cast<const u32x4*>(A + (long)(mb + row) * lda + c * EPC);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = i0 + j * 256, row = i / cpr, c = i - row * cpr;
        if (i < rows_per_pass * cpr) *reinterpret_cast<u32x4*>(lds + row * rowb + ((c ^ (row & 7)) << 4)) = v[j];
      }
    }
    __syncthreads();
    const int ntile = (rows + 15) / 16;
    // per row tile: (value, first index) over this wave's 16 entries -> wbest
    auto finish = [&](const f32x4& acc, int mt) {
      float best = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (n + e < N) {
          const float v = acc[e] + bz[e];
          if (v > best || (v == best && n + e < bi)) { best = v; bi = n + e; }
        }
      }
#pragma unroll
      for (int o = 16; o < 64; o <<= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
      }
      if (q4 == 0) {
        unsigned long long key = 0ull;                  // 0 = nothing (real keys have a non-zero index half)
        if (bi != 0x7fffffff) {
          unsigned u = __float_as_uint(best);
          u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
          key = ((unsigned long long)u << 32) | (unsigned long long)(0xffffffffu - (unsigned)bi);
        }
        wbest[wid * rows_per_pass + mt * 16 + r16] = key;
      }
    };
    if (K == KS * 4 * EPC) {
      // full-depth rows (H = 512 in bf16): no per-step guard, so the 16 fragment reads of a tile are issued ahead of its
      // MFMAs, and two row tiles run as independent accumulator chains (the guarded loop below serialises
      // read -> wait -> MFMA sixteen times per tile: ~10 of the kernel's 13 us at B = 128, V = 10000)
      for (int mt = 0; mt < ntile; mt += 2) {
        const int mt1 = mt + 1 < ntile ? mt + 1 : mt;   // odd tail: the second chain repeats the first and is dropped
        const int row0 = mt * 16 + r16, row1 = mt1 * 16 + r16;
        const char* p0 = lds + row0 * rowb; const char* p1 = lds + row1 * rowb;
        const int s0 = row0 & 7, s1 = row1 & 7;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < KS; ++u) {
          const u32x4 fa0 = *reinterpret_cast<const u32x4*>(p0 + (((u * 4 + q4) ^ s0) << 4));
          const u32x4 fa1 = *reinterpret_cast<const u32x4*>(p1 + (((u * 4 + q4) ^ s1) << 4));
          MfmaD<T>::run(fw[u], fa0, acc0);
          MfmaD<T>::run(fw[u], fa1, acc1);
        }
        finish(acc0, mt);
        if (mt1 != mt) finish(acc1, mt1);
      }
    } else {
      for (int mt = 0; mt < ntile; ++mt) {
        const int row = mt * 16 + r16;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < KS; ++u) {
          if (u * 4 * EPC < K) {
            const u32x4 fa = *reinterpret_cast<const u32x4*>(lds + row * rowb + (((u * 4 + q4) ^ (row & 7)) << 4));
            MfmaD<T>::run(fw[u], fa, acc);
          }
        }
        finish(acc, mt);
      }
    }
    __syncthreads();
    for (int r = threadIdx.x; r < rows; r += 256) {
      unsigned long long k = wbest[r];
#pragma unroll
      for (int w = 1; w < 4; ++w) { const unsigned long long o = wbest[w * rows_per_pass + r]; k = o > k ? o : k; }
      if (k) atomicMax(keys + mb + r, k);
    }
  }
}

// packed (value, index) maxima -> token ids of step t, next-step embedding rows; keys are re-armed (0) for the next step
template <typename T>
__global__ __launch_bounds__(256) void keys_to_ids_embed_kernel(unsigned long long* __restrict__ keys, long* __restrict__ ids,
                                                                int ids_stride, int t, const T* __restrict__ emb, T* __restrict__ x,
                                                                int E, int V) {
  const int row = blockIdx.x;
  const unsigned long long key = keys[row];
  int bi = (int)(0xffffffffu - (unsigned)(key & 0xffffffffull));
  if (bi < 0 || bi >= V) bi = 0;
  __syncthreads();
  if (threadIdx.x == 0) { ids[(long)row * ids_stride + t] = bi; keys[row] = 0ull; }
  constexpr int N = 16 / (int)sizeof(T);
  for (int c = threadIdx.x * N; c < E; c += blockDim.x * N)
    *reinterpret_cast<u32x4*>(x + (long)row * E + c) = *reinterpret_cast<const u32x4*>(emb + (long)bi * E + c);
}

int gemm_nt(const void* a, int lda, const void* w, int ldw, void* y, int ldy, int M, int N, int K, int dtype, int out_dtype,
            const float* bias, void* stream) {
  st_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.x = a; d.w = w; d.y = y; d.bias = bias; d.dtype = dtype; d.out_dtype = out_dtype;
  d.B = M; d.Hin = 1; d.Win = 1; d.Cin = K; d.Ho = 1; d.Wo = 1; d.N = N; d.KH = 1; d.KW = 1; d.stride = 1; d.pad = 0;
  d.ldx = lda; d.ldw = ldw; d.ldy = ldy;
  return st_conv(&d, stream);
}

}  // namespace

// Greedy decoding keeps one arg-max key row per step (no re-arming between steps) for up to kFusedSteps steps, so that
// layer 0 of step t+1 can gather its embedding rows from step t's keys itself (one launch less per step).
constexpr int kFusedSteps = 64;
constexpr int kPipeSteps = 32;     // the pipelined decoder's workspace is sized for up to this many steps (rnn.py:39: 25)

extern "C" size_t st_rnn_greedy_workspace_bytes(const st_rnn_params* p, int B) {
  if (!p || B <= 0) return 0;
  const size_t es = st_dtype_size(p->dtype);
  const size_t hb = al((size_t)p->L * B * p->H * es);
  const int ng = p->cell == ST_CELL_GRU ? 3 : 4;
  return 4 * hb + al((size_t)B * p->E * es) + al((size_t)B * up8(p->V) * sizeof(float)) + al((size_t)B * kFusedSteps * sizeof(unsigned long long)) +
         al((size_t)p->L * B * ng * p->H * sizeof(float)) +      // recurrent halves of the split step (gh)
         rnn_greedy_pipe_bytes(p, B, kPipeSteps);                // the pipelined decoder's buffers (0 when the configuration is not eligible)
}

extern "C" int st_rnn_greedy(const st_rnn_params* p, const void* feat, int B, int steps, void* workspace, size_t workspace_bytes,
                             long* ids_out, float* logits_out, void* stream) {
  ST_CHECK(p && feat && workspace && ids_out, "st_rnn_greedy: null pointer");
  ST_CHECK(p->L >= 1 && p->L <= ST_MAX_LAYERS && p->in0 == p->E, "st_rnn_greedy: bad decoder configuration");
  ST_CHECK(p->H % 8 == 0 && p->E % 8 == 0, "st_rnn_greedy: E=%d and H=%d must be multiples of 8", p->E, p->H);
  ST_CHECK(p->emb && p->w_lin && p->b_lin, "st_rnn_greedy: null weights");
  ST_CHECK(workspace_bytes >= st_rnn_greedy_workspace_bytes(p, B), "st_rnn_greedy: workspace too small");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int dt = p->dtype, H = p->H, E = p->E, V = p->V, Vp = up8(V), L = p->L;
  const size_t es = st_dtype_size(dt);
  const size_t hb = al((size_t)L * B * H * es);
  char* ws = reinterpret_cast<char*>(workspace);
  char* hbuf[2] = {ws, ws + hb};
  char* cbuf[2] = {ws + 2 * hb, ws + 3 * hb};
  char* xbuf = ws + 4 * hb;
  float* logits = reinterpret_cast<float*>(xbuf + al((size_t)B * E * es));
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(logits) + al((size_t)B * Vp * sizeof(float)));
  // bf16 GRU at the BASELINE decoder shape: the whole loop as ONE persistent kernel, a layer per XCD (csrc/decode_pipe.hip); it gives
  // up (rc 2) when the configuration is not eligible or the grid could not be made co-resident -- then the launch chain below runs
  if (!logits_out && steps <= kPipeSteps) {
    const size_t pipe_bytes = rnn_greedy_pipe_bytes(p, B, kPipeSteps);
    if (pipe_bytes) {
      const size_t base = st_rnn_greedy_workspace_bytes(p, B) - pipe_bytes;
      const int rc = rnn_greedy_pipe(p, feat, B, steps, ws + base, pipe_bytes, ids_out, st);
      if (rc == 0) return 0;
      if (rc == 1) return 1;
    }
  }
  const bool fused = !logits_out && steps <= kFusedSteps;   // keys[t][B]; otherwise one row, re-armed by keys_to_ids_embed_kernel
  if (!logits_out && hipMemsetAsync(keys, 0, (size_t)B * (fused ? steps : 1) * sizeof(unsigned long long), st) != hipSuccess) { st_set_error("memset failed"); return 1; }
  const void* x = feat;
  int cur = 0;
  // Split step (greedy fast path; ST_DECODE_SPLIT=0 restores the fused cells): the recurrent half W_hh h_l(t) + b_hh of layer l
  // is NOT on the token -> token chain once h_l(t) exists, so it is computed as extra blocks of the NEXT launch on the stream
  // (cell l+1 of the same step; layer L-1's rides with cell 0 of the next step) and handed over as fp32 `gh`; the cells on the
  // chain multiply only the input half (x, W_ih): half the operand bytes and MFMAs per dependent launch.
  static const bool split_env = [] { const char* e = getenv("ST_DECODE_SPLIT"); return !e || atoi(e) != 0; }();
  const int NGc = p->cell == ST_CELL_GRU ? 3 : 4;
  float* gh = reinterpret_cast<float*>(reinterpret_cast<char*>(keys) + al((size_t)B * kFusedSteps * sizeof(unsigned long long)));
  const size_t gh_l = (size_t)B * NGc * H;
  const bool split = fused && split_env && L >= 2;
  auto gh_cell = [&](int l, const void* hsrc) {          // gh[l] = W_hh[l] hsrc + b_hh[l]  (hsrc == NULL: the bias alone, h = 0)
    RnnGemmArgs g;
    memset(&g, 0, sizeof(g));
    g.M = B; g.N = H; g.gstride = H;
    g.A = hsrc; g.W = p->w_hh[l]; g.K = H; g.lda = H; g.ldw = H;
    g.bias_h = p->b_hh[l]; g.accumulate = kCellRawOut; g.out_f32 = gh + (size_t)l * gh_l; g.ldo = NGc * H;
    return g;
  };
  if (split) {                                           // step 0: h(-1) = 0 -> gh = b_hh for every layer, one launch
    RnnGemmArgs cells[ST_MAX_LAYERS];
    for (int l = 0; l < L; ++l) cells[l] = gh_cell(l, nullptr);
    if (rnn_gemm_launch_batch(cells, L, dt, p->cell == ST_CELL_GRU ? 1 : 2, 0, st)) return 1;
  }
  for (int t = 0; t < steps; ++t) {
    const int nxt = cur ^ 1;
    if (split) {
      for (int l = 0; l < L; ++l) {
        RnnGemmArgs c2[2];
        RnnGemmArgs& a = c2[0];
        memset(&a, 0, sizeof(a));
        a.M = B; a.N = H; a.gstride = H;
        a.A = l == 0 ? x : hbuf[nxt] + (size_t)(l - 1) * B * H * es;         // input half on the MFMA operand pair
        a.W = p->w_ih[l]; a.K = l == 0 ? E : H; a.lda = a.K; a.ldw = a.K;
        if (l == 0 && t > 0) {                            // embedding rows of step t-1's tokens, gathered by the cell
          a.A = p->emb; a.x_keys = keys + (size_t)(t - 1) * B; a.x_V = V;
          a.ids_out = ids_out; a.ids_stride = steps; a.ids_t = t - 1;
        }
        a.bias_h = p->b_ih[l];
        a.accumulate = kCellSplit; a.gx = gh + (size_t)l * gh_l; a.ldgx = NGc * H;
        a.hprev = t > 0 ? hbuf[cur] + (size_t)l * B * H * es : nullptr; a.ldhp = H;
        a.hout = hbuf[nxt] + (size_t)l * B * H * es; a.ldho = H;
        if (p->cell == ST_CELL_LSTM) {
          a.cprev = t > 0 ? cbuf[cur] + (size_t)l * B * H * es : nullptr;
          a.cout = cbuf[nxt] + (size_t)l * B * H * es;
        }
        int n = 1;
        // riding along: the recurrent half some LATER cell needs, from a state that is already complete
        if (l == 0) { if (t > 0) c2[n++] = gh_cell(L - 1, hbuf[cur] + (size_t)(L - 1) * B * H * es); }   // h_{L-1}(t-1) -> cell L-1 of THIS step
        else if (t + 1 < steps) c2[n++] = gh_cell(l - 1, hbuf[nxt] + (size_t)(l - 2 + 1) * B * H * es);   // h_{l-1}(t) -> cell l-1 of step t+1
        if (rnn_gemm_launch_batch(c2, n, dt, p->cell == ST_CELL_GRU ? 1 : 2, 0, st)) return 1;
      }
    } else
    for (int l = 0; l < L; ++l) {
      RnnGemmArgs a;
      memset(&a, 0, sizeof(a));
      a.M = B; a.N = H; a.gstride = H;
      a.A2 = l == 0 ? x : hbuf[nxt] + (size_t)(l - 1) * B * H * es;
      a.W2 = p->w_ih[l]; a.K2 = l == 0 ? E : H; a.lda2 = a.K2; a.ldw2 = a.K2;
      if (l == 0 && fused && t > 0) {                   // x = embedding rows of step t-1's tokens, gathered by the cell
        a.A2 = p->emb; a.x_keys = keys + (size_t)(t - 1) * B; a.x_V = V;
        a.ids_out = ids_out; a.ids_stride = steps; a.ids_t = t - 1;
      }
      a.A = t > 0 ? hbuf[cur] + (size_t)l * B * H * es : nullptr;
      a.W = p->w_hh[l]; a.K = H; a.lda = H; a.ldw = H;
      a.hprev = a.A; a.ldhp = H;
      a.bias_h = p->b_hh[l]; a.bias_x = p->b_ih[l];
      a.hout = hbuf[nxt] + (size_t)l * B * H * es; a.ldho = H;
      if (p->cell == ST_CELL_LSTM) {
        a.cprev = t > 0 ? cbuf[cur] + (size_t)l * B * H * es : nullptr;
        a.cout = cbuf[nxt] + (size_t)l * B * H * es;
      }
      if (rnn_gemm_launch(a, dt, p->cell == ST_CELL_GRU ? 1 : 2, 1, st)) return 1;
    }
    if (!logits_out) {
      // fast path: vocabulary projection with a fused running arg-max (the logits are never written)
      const char* htop = hbuf[nxt] + (size_t)(L - 1) * B * H * es;
      unsigned long long* kt = fused ? keys + (size_t)t * B : keys;
      const int epc = dt == ST_BF16 ? 8 : 4;
      int rpp = (int)((128 * 1024) / ((size_t)H * es)) & ~15;
      if (rpp > ((B + 15) & ~15)) rpp = (B + 15) & ~15;
      int ysplit = 1;                                   // V/64 blocks alone fill 157 of 256 CUs at V = 10000: split the rows too
      while ((V + 63) / 64 * ysplit < 256 && rpp >= 32 && rpp % 32 == 0 && ysplit < 4) { ysplit *= 2; rpp /= 2; }
      if (H <= 64 * epc && H % (8 * epc) == 0 && rpp >= 16) {
        // LDS-staged form: activations once per block, projection matrix once chip-wide
        const size_t lds = (size_t)rpp * H * es + (size_t)4 * rpp * sizeof(unsigned long long);
        static bool attr_set[64] = {};                  // the attribute belongs to the (function, device) pair
        int dev_ = 0;
        (void)hipGetDevice(&dev_);
        if (dev_ >= 0 && dev_ < 64 && !attr_set[dev_]) {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vocab_argmax_lds_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vocab_argmax_lds_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          attr_set[dev_] = true;
        }
        const dim3 vgrid(((V + 63) / 64 + 7) / 8 * 8 * ysplit);
        if (dt == ST_BF16)
          hipLaunchKernelGGL(vocab_argmax_lds_kernel<bf16_t>, vgrid, dim3(256), lds, st, (const bf16_t*)htop, H, (const bf16_t*)p->w_lin, H, p->b_lin, B, V, H, kt, rpp, ysplit);
        else
          hipLaunchKernelGGL(vocab_argmax_lds_kernel<float>, vgrid, dim3(256), lds, st, (const float*)htop, H, (const float*)p->w_lin, H, p->b_lin, B, V, H, kt, rpp, ysplit);
      } else {
        const int nsl = (V + 127) / 128, nmt = (B + 15) / 16;
        const dim3 vgrid(((nsl + 7) / 8) * 8 * nmt);
        if (dt == ST_BF16)
          hipLaunchKernelGGL(vocab_argmax_kernel<bf16_t>, vgrid, dim3(256), 0, st, (const bf16_t*)htop, H, (const bf16_t*)p->w_lin, H, p->b_lin, B, V, H, kt);
        else
          hipLaunchKernelGGL(vocab_argmax_kernel<float>, vgrid, dim3(256), 0, st, (const float*)htop, H, (const float*)p->w_lin, H, p->b_lin, B, V, H, kt);
      }
      ST_LAUNCH_CHECK();
      if (!fused || t == steps - 1) {                   // fused: only the last step's ids are left to write
        if (dt == ST_BF16)
          hipLaunchKernelGGL(keys_to_ids_embed_kernel<bf16_t>, dim3(B), dim3(64), 0, st, kt, ids_out, steps, t, (const bf16_t*)p->emb, (bf16_t*)xbuf, E, V);
        else
          hipLaunchKernelGGL(keys_to_ids_embed_kernel<float>, dim3(B), dim3(64), 0, st, kt, ids_out, steps, t, (const float*)p->emb, (float*)xbuf, E, V);
        ST_LAUNCH_CHECK();
      }
      x = xbuf;
      cur = nxt;
      continue;
    }
    float* lg = logits_out + (size_t)t * B * Vp;
    if (gemm_nt(hbuf[nxt] + (size_t)(L - 1) * B * H * es, H, p->w_lin, H, lg, Vp, B, V, H, dt, ST_F32, p->b_lin, stream)) return 1;
    if (dt == ST_BF16)
      hipLaunchKernelGGL(argmax_embed_kernel<bf16_t>, dim3(B), dim3(256), 0, st, lg, Vp, V, ids_out, steps, t, (const bf16_t*)p->emb, (bf16_t*)xbuf, E);
    else
      hipLaunchKernelGGL(argmax_embed_kernel<float>, dim3(B), dim3(256), 0, st, lg, Vp, V, ids_out, steps, t, (const float*)p->emb, (float*)xbuf, E);
    ST_LAUNCH_CHECK();
    x = xbuf;
    cur = nxt;
  }
  return 0;
}

extern "C" int st_embedding_rows(const void* emb, const long* ids, void* out, int n, int E, int V, int ldo, int dtype, void* stream) {
  ST_CHECK(emb && ids && out, "st_embedding_rows: null pointer");
  const int nn = dtype == ST_BF16 ? 8 : 4;
  ST_CHECK(E % nn == 0 && ldo % nn == 0 && ldo >= E, "st_embedding_rows: E=%d / ldo=%d must be multiples of %d", E, ldo, nn);
  if (n <= 0) return 0;
  long b = ((long)n * (E / nn) + 255) / 256; if (b > 2048) b = 2048;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == ST_BF16) hipLaunchKernelGGL(embedding_rows_kernel<bf16_t>, dim3((int)b), dim3(256), 0, st, (const bf16_t*)emb, ids, (bf16_t*)out, n, E, V, ldo);
  else hipLaunchKernelGGL(embedding_rows_kernel<float>, dim3((int)b), dim3(256), 0, st, (const float*)emb, ids, (float*)out, n, E, V, ldo);
  ST_LAUNCH_CHECK();
  return 0;
}

extern "C" int st_gather_state(const void* src, const int* idx, void* dst, int L, int n_src, int n_dst, int H, int dtype, void* stream) {
  ST_CHECK(src && idx && dst, "st_gather_state: null pointer");
  const int nn = dtype == ST_BF16 ? 8 : 4;
  ST_CHECK(H % nn == 0, "st_gather_state: H=%d must be a multiple of %d", H, nn);
  if (n_dst <= 0) return 0;
  long b = ((long)L * n_dst * (H / nn) + 255) / 256; if (b > 2048) b = 2048;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == ST_BF16) hipLaunchKernelGGL(gather_state_kernel<bf16_t>, dim3((int)b), dim3(256), 0, st, (const bf16_t*)src, idx, (bf16_t*)dst, L, n_src, n_dst, H);
  else hipLaunchKernelGGL(gather_state_kernel<float>, dim3((int)b), dim3(256), 0, st, (const float*)src, idx, (float*)dst, L, n_src, n_dst, H);
  ST_LAUNCH_CHECK();
  return 0;
}

extern "C" int st_softmax_topk(const float* logits, int ldl, int n, int V, int k, float* top_p, long* top_id, int raw, void* stream) {
  ST_CHECK(logits && top_p && top_id, "st_softmax_topk: null pointer");
  ST_CHECK(k >= 1 && k <= 32 && k <= V && ldl >= V, "st_softmax_topk: need 1 <= k <= min(32, V)");
  if (n <= 0) return 0;
  // k <= 8 (every beam width in use): two passes over the row; the 2 + k pass form stays as the general case and as the
  // cross-check (ST_TOPK_SLOW=1)
  static int slow = -1;
  if (slow < 0) { const char* e = getenv("ST_TOPK_SLOW"); slow = e ? atoi(e) : 0; }
  static const bool thr = [] { const char* e = getenv("ST_TOPK_THR"); return !e || atoi(e) != 0; }();   // A/B switch: 0 = the sorted-list form
  if (k <= 8 && !slow && thr && (long)k * 256 <= V)
    hipLaunchKernelGGL(softmax_topk_thr_kernel, dim3(n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), logits, ldl, V, k, top_p, top_id, raw);
  else if (k <= 5 && !slow && (long)k * 256 <= V)
    hipLaunchKernelGGL(softmax_topk_fast_kernel<5>, dim3(n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), logits, ldl, V, k, top_p, top_id, raw);
  else if (k <= 8 && !slow && (long)k * 256 <= V)
    hipLaunchKernelGGL(softmax_topk_fast_kernel<8>, dim3(n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), logits, ldl, V, k, top_p, top_id, raw);
  else
    hipLaunchKernelGGL(softmax_topk_kernel, dim3(n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), logits, ldl, V, k, top_p, top_id, raw);
  ST_LAUNCH_CHECK();
  return 0;
}

// One iteration of beam_search.py:69-94 for every image at once, on the device (no host round trip per step).
// A block = one image; slot (b, w) is fringe node w.  From the OLD fringe (tok, cost; cost = +inf marks an empty slot):
//   ended = slot holds <end> and the image is not finished            (beam_search.py:72-76: harvested by the host afterwards)
//   live  = the other occupied slots; an image without live slots is finished from now on (:78-79)
// candidates in the reference's order -- node-major, each node's k successors in ASCENDING probability (argsort(p)[-k:]) --
// cost = float32(cost[w] + (-log p)); the W best by a STABLE ranking (sorted(...)[:beam_width], :94) become the new fringe.
// Outputs: new tok / cost, the parent slot of every new node, gather[b*W + w] = row of the parent's recurrent state.
__global__ __launch_bounds__(64) void beam_select_kernel(const long* __restrict__ tok, const float* __restrict__ cost, uint8_t* __restrict__ done,
                                                         const float* __restrict__ top_p, const long* __restrict__ top_id, int W, int k, long end_id,
                                                         long* __restrict__ new_tok, float* __restrict__ new_cost, int* __restrict__ parent,
                                                         uint8_t* __restrict__ ended_rec, int* __restrict__ gather) {
  __shared__ float cc[64];
  __shared__ int any_live;
  const int b = blockIdx.x, c = threadIdx.x;
  const bool was_done = done[b] != 0;
  if (c == 0) any_live = 0;
  __syncthreads();
  if (c < W) {
    const bool occupied = cost[b * W + c] < INFINITY;
    const bool ended = occupied && tok[b * W + c] == end_id && !was_done;
    ended_rec[b * W + c] = ended ? 1 : 0;
    if (occupied && !ended && !was_done) any_live = 1;
  }
  __syncthreads();
  const bool finished = was_done || !any_live;
  if (c == 0) done[b] = finished ? 1 : 0;
  float my = INFINITY;
  const int n = W * k;
  if (c < n && !finished) {
    const int w = c / k, j = c - w * k;                      // successor j in ascending probability = top-k entry k-1-j
    const float cw = cost[b * W + w];
    if (cw < INFINITY && tok[b * W + w] != end_id)
      my = cw + (float)(-log((double)top_p[(long)(b * W + w) * k + (k - 1 - j)]));
  }
  cc[c] = c < n ? my : INFINITY;
  __syncthreads();
  if (c < n) {
    int rank = 0;                                            // stable: ties keep candidate order
    for (int o = 0; o < n; ++o) rank += (cc[o] < my || (cc[o] == my && o < c)) ? 1 : 0;
    if (rank < W) {
      const int w = c / k, j = c - w * k;
      const bool ok = my < INFINITY;
      new_cost[b * W + rank] = ok ? my : INFINITY;
      new_tok[b * W + rank] = ok ? top_id[(long)(b * W + w) * k + (k - 1 - j)] : 0;
      parent[b * W + rank] = ok ? w : -1;
      gather[b * W + rank] = b * W + (ok ? w : 0);
    }
  }
}

extern "C" int st_beam_select(const long* tok, const float* cost, uint8_t* done, const float* top_p, const long* top_id, int B, int W, int k,
                              long end_id, long* new_tok, float* new_cost, int* parent, uint8_t* ended, int* gather, void* stream) {
  ST_CHECK(tok && cost && done && top_p && top_id && new_tok && new_cost && parent && ended && gather, "st_beam_select: null pointer");
  ST_CHECK(B > 0 && W >= 1 && k >= 1 && k <= W && W * k <= 64, "st_beam_select: need 1 <= k <= W and W * k <= 64 (got W=%d k=%d)", W, k);
  hipLaunchKernelGGL(beam_select_kernel, dim3(B), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), tok, cost, done, top_p, top_id, W, k, end_id,
                     new_tok, new_cost, parent, ended, gather);
  ST_LAUNCH_CHECK();
  return 0;
}

// One timestep of the multi-layer cell for n independent rows (+ optional vocabulary projection).
extern "C" int st_rnn_step(const st_rnn_params* p, const void* x, int n, const void* h_in, const void* c_in,
                           void* h_out, void* c_out, float* logits, int ldl, void* stream) {
  ST_CHECK(p && x && h_out, "st_rnn_step: null pointer");
  ST_CHECK(p->L >= 1 && p->L <= ST_MAX_LAYERS, "st_rnn_step: bad layer count");
  ST_CHECK(p->H % 8 == 0 && p->in0 % 8 == 0, "st_rnn_step: in0=%d and H=%d must be multiples of 8", p->in0, p->H);
  ST_CHECK(p->cell == ST_CELL_GRU || c_out, "st_rnn_step: LSTM needs c_out");
  ST_CHECK(h_in != h_out, "st_rnn_step: h_in and h_out must not alias");
  if (n <= 0) return 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int dt = p->dtype, H = p->H;
  const size_t es = st_dtype_size(dt);
  for (int l = 0; l < p->L; ++l) {
    RnnGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.M = n; a.N = H; a.gstride = H;
    a.A2 = l == 0 ? x : reinterpret_cast<const char*>(h_out) + (size_t)(l - 1) * n * H * es;
    a.W2 = p->w_ih[l]; a.K2 = l == 0 ? p->in0 : H; a.lda2 = a.K2; a.ldw2 = a.K2;
    a.A = h_in ? reinterpret_cast<const char*>(h_in) + (size_t)l * n * H * es : nullptr;
    a.W = p->w_hh[l]; a.K = H; a.lda = H; a.ldw = H;
    a.hprev = a.A; a.ldhp = H;
    a.bias_h = p->b_hh[l]; a.bias_x = p->b_ih[l];
    a.hout = reinterpret_cast<char*>(h_out) + (size_t)l * n * H * es; a.ldho = H;
    if (p->cell == ST_CELL_LSTM) {
      a.cprev = c_in ? reinterpret_cast<const char*>(c_in) + (size_t)l * n * H * es : nullptr;
      a.cout = reinterpret_cast<char*>(c_out) + (size_t)l * n * H * es;
    }
    if (rnn_gemm_launch(a, dt, p->cell == ST_CELL_GRU ? 1 : 2, 1, st)) return 1;
  }
  if (logits) {
    ST_CHECK(p->w_lin && p->b_lin && ldl % 4 == 0 && ldl >= p->V, "st_rnn_step: bad logits buffer");
    if (gemm_nt(reinterpret_cast<const char*>(h_out) + (size_t)(p->L - 1) * n * H * es, H, p->w_lin, H, logits, ldl, n, p->V, H, dt, ST_F32,
                p->b_lin, stream)) return 1;
  }
  return 0;
}
