// Recurrent-decoder kernels for gfx950: packed-sequence input gather, the per-timestep
// "skinny" MFMA GEMM with fused GRU / LSTM gate epilogues, the BPTT gate-gradient kernels,
// embedding scatter-add, cross-entropy, column sums.
//
// Replaces what the reference delegates to cuDNN's fused RNN and to torch ops:
//   rnn.py:29-31   embedding + cat + pack_padded_sequence      -> pack_inputs_kernel
//   rnn.py:32      nn.GRU over a PackedSequence (fwd + autograd) -> rnn_gemm_kernel<GRU_*>, gru_bwd_gates_kernel
//   rnn_lstm.py:30 nn.LSTM                                       -> rnn_gemm_kernel<LSTM_*>, lstm_bwd_gates_kernel
//   main.py:149    nn.CrossEntropyLoss (fwd + bwd)               -> ce_kernel
//
// The recurrent product h_{t-1} W_hh^T has only B_t <= batch rows, so it is launch/latency
// bound, not FLOP bound: one block owns 16 hidden units (all gates of those units) x 16 batch
// rows, operands go straight from L2 to MFMA fragments (no LDS: each fragment is used once),
// and the whole gate nonlinearity runs in the epilogue so a timestep is ONE launch.
#include "common.h"
#include "rnn_kernels.h"
#include <string.h>

namespace {

template <typename T> struct Mfma;
template <> struct Mfma<bf16_t> {
  static constexpr int EPC = 8;
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a),
                                                *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
  }
};
template <> struct Mfma<float> {
  static constexpr int EPC = 4;
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[e]), __uint_as_float(b[e]), c, 0, 0, 0);
  }
};

// 4 consecutive elements of T <-> 4 floats
template <typename T> __device__ __forceinline__ void load4(const T* p, float* v);
template <> __device__ __forceinline__ void load4<float>(const float* p, float* v) {
  const f32x4 t = *reinterpret_cast<const f32x4*>(p);
  v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float* v) {
  const u32x2 t = *reinterpret_cast<const u32x2*>(p);
  v[0] = __uint_as_float(t[0] << 16); v[1] = __uint_as_float(t[0] & 0xffff0000u);
  v[2] = __uint_as_float(t[1] << 16); v[3] = __uint_as_float(t[1] & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float* v);
template <> __device__ __forceinline__ void store4<float>(float* p, const float* v) {
  *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float* v) {
  *reinterpret_cast<u32x2*>(p) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
}

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---------------------------------------------------------------------------------------
// skinny GEMM  acc[g][m][n] = sum_k A[m][k] * W[g*gstride + n][k]   (+ optional second pair)
// ---------------------------------------------------------------------------------------
template <typename T, int NG, bool HAS_X, int MT>
__device__ __forceinline__ void skinny_mma(const RnnGemmArgs& a, int mbase, int n0, int r16, int q4, int kslice,
                                           f32x4 (&accH)[MT][NG], f32x4 (&accX)[MT][NG]) {
  constexpr int EPC = Mfma<T>::EPC;
  // K steps requested per round trip; the fused two-operand, two-row-tile form halves it to stay within 2 waves per SIMD
  // (260 -> ~170 VGPRs: two blocks per CU overlap instead of one)
  constexpr int UNR = (HAS_X && MT == 2) ? 2 : 4;
  const int n = n0 + r16;
  const bool nok = n < a.N;
  // The kernel is latency bound (every fragment comes from L2), so bytes in flight per wave are the lever: K is split
  // over the block's 4 waves and a group of UNR K-steps of BOTH operand pairs (h W_hh and, in the fused form, x W_ih)
  // is requested before the first MFMA of the group -- one round trip per group.  With MT > 1 the block covers MT
  // 16-row tiles with the SAME weight fragments held in registers: the weights, the bulk of the L2 traffic when many
  // cells run in one launch, are fetched once per MT tiles.
  const bool hasH = a.A != nullptr, hasX = HAS_X && a.A2 != nullptr;
  const T* WH = reinterpret_cast<const T*>(a.W) + (long)n * a.ldw;
  const long gsH = (long)a.gstride * a.ldw;
  const T* WX = reinterpret_cast<const T*>(a.W2) + (long)n * a.ldw2;
  const long gsX = (long)a.gstride * a.ldw2;
  const T* AH[MT]; const T* AX[MT]; bool mok[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int m = mbase + t * 16 + r16;
    mok[t] = m < a.M;
    AH[t] = reinterpret_cast<const T*>(a.A) + (long)m * a.lda;
    if (!HAS_X && a.x_keys && a.A) {                   // split decode step: the A operand is the embedding row of the previous token
      int tok = 0;
      if (mok[t]) {
        tok = (int)(0xffffffffu - (unsigned)(a.x_keys[m] & 0xffffffffull));
        if (tok < 0 || tok >= a.x_V) tok = 0;
        if (a.ids_out && n0 == 0 && kslice == 0 && q4 == 0) a.ids_out[(long)m * a.ids_stride + a.ids_t] = tok;
      }
      AH[t] = reinterpret_cast<const T*>(a.A) + (long)tok * a.lda;
    }
    long xrow = m;
    if (HAS_X && a.x_keys) {                           // token of the previous step -> embedding row
      int tok = 0;
      if (mok[t]) {
        tok = (int)(0xffffffffu - (unsigned)(a.x_keys[m] & 0xffffffffull));
        if (tok < 0 || tok >= a.x_V) tok = 0;
        if (a.ids_out && n0 == 0 && kslice == 0 && q4 == 0) a.ids_out[(long)m * a.ids_stride + a.ids_t] = tok;
      }
      xrow = tok;
    }
    AX[t] = reinterpret_cast<const T*>(a.A2) + xrow * a.lda2;
  }
  const int nsH = hasH ? (a.K + 4 * EPC - 1) / (4 * EPC) : 0, nsX = hasX ? (a.K2 + 4 * EPC - 1) / (4 * EPC) : 0;
  const int spwH = (nsH + 3) / 4, spwX = (nsX + 3) / 4;
  const int begH = kslice * spwH, endH = min(nsH, begH + spwH), begX = kslice * spwX, endX = min(nsX, begX + spwX);
  const int ngroups = max((endH - begH + UNR - 1) / UNR, (endX - begX + UNR - 1) / UNR);
  for (int gi = 0; gi < ngroups; ++gi) {
    u32x4 fa[MT][UNR], fw[UNR][NG], xa[MT][UNR], xw[UNR][NG];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int sH = begH + gi * UNR + u, kH = (sH * 4 + q4) * EPC;
      const bool okH = sH < endH && kH < a.K;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        fa[t][u] = u32x4{0u, 0u, 0u, 0u};
        if (mok[t] && okH) fa[t][u] = *reinterpret_cast<const u32x4*>(AH[t] + kH);
      }
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        fw[u][g] = u32x4{0u, 0u, 0u, 0u};
        if (nok && okH) fw[u][g] = *reinterpret_cast<const u32x4*>(WH + g * gsH + kH);
      }
      if (HAS_X) {
        const int sX = begX + gi * UNR + u, kX = (sX * 4 + q4) * EPC;
        const bool okX = sX < endX && kX < a.K2;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          xa[t][u] = u32x4{0u, 0u, 0u, 0u};
          if (mok[t] && okX) xa[t][u] = *reinterpret_cast<const u32x4*>(AX[t] + kX);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          xw[u][g] = u32x4{0u, 0u, 0u, 0u};
          if (nok && okX) xw[u][g] = *reinterpret_cast<const u32x4*>(WX + g * gsX + kX);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          Mfma<T>::run(fw[u][g], fa[t][u], accH[t][g]);
          if (HAS_X) Mfma<T>::run(xw[u][g], xa[t][u], accX[t][g]);
        }
  }
}

// EPI 0: out_f32[m][n] (+)= acc (+bias)         (BPTT dh += dgh W_hh, generic small products)
// EPI 1: GRU gates (training fwd with precomputed gx, or decode with fused x-projection)
// EPI 2: LSTM gates
template <typename T, int NG, int EPI, bool HAS_X, int MT>
__global__ __launch_bounds__(256) void rnn_gemm_kernel(RnnGemmBatch batch) {
  // block = MT x 16 rows x 16 units (all NG gates); its 4 waves each take a quarter of K and the partial
  // accumulators meet in LDS: one round of L2 latency per launch instead of four.  Wave t < MT then runs the
  // epilogue of row tile t.  blockIdx.z selects one of the launch's independent cells (kernel-argument segment).
  const RnnGemmArgs& a = batch.c[blockIdx.z];
  __shared__ f32x4 red[4][MT][2 * NG][64];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r16 = lane & 15, q4 = lane >> 4;
  const int n0 = blockIdx.x * 16, mbase = blockIdx.y * 16 * MT;
  if (mbase >= a.M || n0 >= a.N) return;            // the grid is sized for the launch's largest cell
  const int m0 = mbase + (wid < MT ? wid : 0) * 16;  // the row tile whose epilogue this wave owns
  f32x4 accH[MT][NG], accX[MT][NG];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int g = 0; g < NG; ++g) { accH[t][g] = f32x4{0.f, 0.f, 0.f, 0.f}; accX[t][g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  // the epilogue operands (biases, precomputed x-projection, previous state) are requested BEFORE the fragment loads
  // so that they arrive under the same round trip
  const int pm = m0 + r16, pn = n0 + 4 * q4;
  const bool epi_ok = wid < MT && pm < a.M && pn < a.N && EPI != 3;
  float ex[NG][4], eb[NG][4], es[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) { ex[g][e] = 0.f; eb[g][e] = 0.f; }
  if (epi_ok) {
    if (EPI == 0) {
      if (a.bias_h) load4<float>(a.bias_h + pn, eb[0]);
      if (a.accumulate) load4<float>(a.out_f32 + (long)pm * a.ldo + pn, es);
    } else {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        load4<float>(a.bias_h + g * a.N + pn, eb[g]);
        if (HAS_X) load4<float>(a.bias_x + g * a.N + pn, ex[g]);
        else if (a.accumulate == kCellSplit) load4<float>(reinterpret_cast<const float*>(a.gx) + (long)pm * a.ldgx + g * a.N + pn, ex[g]);   // recurrent half, already holds b_hh
        else if (a.accumulate != kCellRawOut) load4<T>(reinterpret_cast<const T*>(a.gx) + (long)pm * a.ldgx + g * a.N + pn, ex[g]);   // already holds b_ih
      }
      const void* prev = EPI == 1 ? a.hprev : a.cprev;
      if (prev && a.accumulate != kCellRawOut) load4<T>(reinterpret_cast<const T*>(prev) + (long)pm * a.ldhp + pn, es);
    }
  }
  skinny_mma<T, NG, HAS_X, MT>(a, mbase, n0, r16, q4, wid, accH, accX);
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int g = 0; g < NG; ++g) { red[wid][t][g][lane] = accH[t][g]; if (HAS_X) red[wid][t][NG + g][lane] = accX[t][g]; }
  __syncthreads();
  if (wid >= MT) return;
  f32x4 sumH[NG], sumX[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    sumH[g] = red[0][wid][g][lane];
    if (HAS_X) sumX[g] = red[0][wid][NG + g][lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      sumH[g] += red[w][wid][g][lane];
      if (HAS_X) sumX[g] += red[w][wid][NG + g][lane];
    }
  }

  // lane owns row m = m0 + r16 and units n = n0 + 4*q4 + {0..3}
  const int m = m0 + r16, n = n0 + 4 * q4;
  if (EPI != 3 && (m >= a.M || n >= a.N)) return;
  const int H = a.N;
  if (EPI == 3) {
    // greedy decoding: the logits are never written; (value, first index) per row by 64-bit atomic max
    float best = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (n + e < a.N) {
        const float v = sumH[0][e] + (a.bias_h ? a.bias_h[n + e] : 0.f);
        if (v > best) { best = v; bi = n + e; }
      }
    }
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
      const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (q4 == 0 && bi != 0x7fffffff && m < a.M) {   // rows past M took part in the shuffles only
      unsigned u = __float_as_uint(best);
      u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
      const unsigned long long key = ((unsigned long long)u << 32) | (unsigned long long)(0xffffffffu - (unsigned)bi);
      atomicMax(a.argmax_keys + m, key);
    }
    return;
  }
  if (EPI == 0) {
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = sumH[0][e] + eb[0][e] + es[e];
    if (a.hout) store4<T>(reinterpret_cast<T*>(a.hout) + (long)m * a.ldho + n, v);      // output in the storage type (ldho)
    else *reinterpret_cast<f32x4*>(a.out_f32 + (long)m * a.ldo + n) = f32x4{v[0], v[1], v[2], v[3]};
  } else if (!HAS_X && a.accumulate == kCellRawOut) {
    // recurrent half for the NEXT use of this layer's cell: gate sums + b_hh, fp32, [m][g * N + n]
#pragma unroll
    for (int g = 0; g < NG; ++g)
      *reinterpret_cast<f32x4*>(a.out_f32 + (long)m * a.ldo + g * a.N + n) =
          f32x4{sumH[g][0] + eb[g][0], sumH[g][1] + eb[g][1], sumH[g][2] + eb[g][2], sumH[g][3] + eb[g][3]};
  } else if (EPI == 1) {
    // r,z,n order (torch.nn.GRU): r = s(xr+hr), z = s(xz+hz), n = tanh(xn + r*(hn)), h' = (1-z) n + z h
    float xg[3][4];
    const float* hp = es;
    const bool split = !HAS_X && a.accumulate == kCellSplit;   // MFMA sums = input half (+ b_ih in eb), ex = recurrent half (+ b_hh)
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) xg[g][e] = split ? sumH[g][e] + eb[g][e] : (HAS_X ? sumX[g][e] : 0.f) + ex[g][e];
    float hn[4], r[4], z[4], nn[4], hnew[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float hr = split ? ex[0][e] : sumH[0][e] + eb[0][e], hz = split ? ex[1][e] : sumH[1][e] + eb[1][e];
      hn[e] = split ? ex[2][e] : sumH[2][e] + eb[2][e];
      hnew[e] = st_gru_unit(xg[0][e], xg[1][e], xg[2][e], hr, hz, hn[e], hp[e], r[e], z[e], nn[e]);
    }
    store4<T>(reinterpret_cast<T*>(a.hout) + (long)m * a.ldho + n, hnew);
    if (a.hout2) store4<T>(reinterpret_cast<T*>(a.hout2) + (long)m * a.ldho2 + n, hnew);
    if (a.cache) {
      T* c = reinterpret_cast<T*>(a.cache) + (long)m * a.ldcache + n;
      store4<T>(c, r); store4<T>(c + H, z); store4<T>(c + 2 * H, nn); store4<T>(c + 3 * H, hn);
    }
  } else {
    // i,f,g,o order (torch.nn.LSTM): c' = f c + i g ; h' = o tanh(c')
    float pre[4][4];
    const float* cp = es;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) pre[g][e] = (HAS_X ? sumX[g][e] : 0.f) + ex[g < NG ? g : 0][e] + sumH[g < NG ? g : 0][e] + eb[g < NG ? g : 0][e];
    float ig[4], fg[4], gg[4], og[4], cn[4], hnew[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) hnew[e] = st_lstm_unit(pre[0][e], pre[1][e], pre[2][e], pre[3][e], cp[e], ig[e], fg[e], gg[e], og[e], cn[e]);
    store4<T>(reinterpret_cast<T*>(a.hout) + (long)m * a.ldho + n, hnew);
    if (a.hout2) store4<T>(reinterpret_cast<T*>(a.hout2) + (long)m * a.ldho2 + n, hnew);
    store4<T>(reinterpret_cast<T*>(a.cout) + (long)m * a.ldho + n, cn);
    if (a.cache) {
      T* c = reinterpret_cast<T*>(a.cache) + (long)m * a.ldcache + n;
      store4<T>(c, ig); store4<T>(c + H, fg); store4<T>(c + 2 * H, gg); store4<T>(c + 3 * H, og);
    }
  }
}

// ---------------------------------------------------------------------------------------
// packed input rows: x0[row(t,b)] = t == 0 ? feat[b] : emb[caption[b][t-1]]   (rnn.py:29-31)
//                    target[row(t,b)] = caption[b][t]                          (main.py:145)
// mode 1 (attention decoder, rnn_attn.py:70): x0[row] = emb[caption[b][t]]
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pack_inputs_kernel(const T* __restrict__ feat, const T* __restrict__ emb,
                                                          const long* __restrict__ cap, int Tcap,
                                                          const int* __restrict__ rows_b, const int* __restrict__ rows_t,
                                                          T* __restrict__ x0, long* __restrict__ target,
                                                          int ntok, int E, int V, int mode) {
  constexpr int N = 16 / (int)sizeof(T);
  const int cpr = E / N;
  const long total = (long)ntok * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i / cpr), c = (int)(i - (long)row * cpr) * N;
    const int b = rows_b[row], t = rows_t[row];
    const T* src;
    if (mode == 0 && t == 0) src = feat + (long)b * E + c;
    else {
      long tok = cap[(long)b * Tcap + (mode == 0 ? t - 1 : t)];
      tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);   // keep a corrupt id from faulting the GPU
      src = emb + tok * E + c;
    }
    *reinterpret_cast<u32x4*>(x0 + (long)row * E + c) = *reinterpret_cast<const u32x4*>(src);
    if (c == 0 && target) target[row] = cap[(long)b * Tcap + t];
  }
}

// dX0 rows -> dfeat (t == 0, plain store) and dEmb (t >= 1, fp32 atomics; 256 contiguous bytes per wave-instruction)
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const float* __restrict__ dx0, const long* __restrict__ cap, int Tcap,
                                                            const int* __restrict__ rows_b, const int* __restrict__ rows_t,
                                                            float* __restrict__ dfeat, float* __restrict__ demb,
                                                            int ntok, int E, int V, int mode) {
  const long total = (long)ntok * E;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i / E), c = (int)(i - (long)row * E);
    const int b = rows_b[row], t = rows_t[row];
    const float v = dx0[i];
    if (mode == 0 && t == 0) { if (dfeat) dfeat[(long)b * E + c] = v; }
    else {
      long tok = cap[(long)b * Tcap + (mode == 0 ? t - 1 : t)];
      tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
      atomicAdd(demb + tok * E + c, v);
    }
  }
}

// hprev[row(t,b)] = t == 0 ? h0 : y[row(t-1,b)]   (operand of dW_hh = sum_t dgh_t^T h_{t-1})
template <typename T>
__global__ __launch_bounds__(256) void gather_hprev_kernel(const T* __restrict__ y, const int* __restrict__ rows_t,
                                                           const int* __restrict__ prev_row, T* __restrict__ hp, int ntok, int H,
                                                           const T* __restrict__ h0, const int* __restrict__ rows_b) {
  constexpr int N = 16 / (int)sizeof(T);
  const int cpr = H / N;
  const long total = (long)ntok * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i / cpr), c = (int)(i - (long)row * cpr) * N;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (rows_t[row] > 0) v = *reinterpret_cast<const u32x4*>(y + (long)prev_row[row] * H + c);
    else if (h0) v = *reinterpret_cast<const u32x4*>(h0 + (long)rows_b[row] * H + c);
    *reinterpret_cast<u32x4*>(hp + (long)row * H + c) = v;
  }
}

// GRU BPTT gate gradients for the B_t rows of one timestep (see DESIGN.md for the algebra)
template <typename T>
__global__ __launch_bounds__(256) void gru_bwd_gates_kernel(RnnBwdBatch batch, int H) {
  const RnnBwdCell& cl = batch.c[blockIdx.y];
  const float* __restrict__ dy = cl.dy; float* __restrict__ dhc = cl.dhc;
  const T* __restrict__ cache = reinterpret_cast<const T*>(cl.cache);
  const T* __restrict__ hprev = reinterpret_cast<const T*>(cl.hprev);
  T* __restrict__ dgx = reinterpret_cast<T*>(cl.dgx); T* __restrict__ dgh = reinterpret_cast<T*>(cl.dgh);
  const int Bt = cl.Bt;
  const int total = Bt * (H / 4);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int b = i / (H / 4), j = (i - b * (H / 4)) * 4;
  float g[4], c[4], r[4], z[4], n[4], hn[4], hp[4] = {0.f, 0.f, 0.f, 0.f};
  load4<float>(dy + (long)b * H + j, g);
  load4<float>(dhc + (long)b * H + j, c);
  const T* cr = cache + (long)b * 4 * H + j;
  load4<T>(cr, r); load4<T>(cr + H, z); load4<T>(cr + 2 * H, n); load4<T>(cr + 3 * H, hn);
  if (hprev) load4<T>(hprev + (long)b * H + j, hp);
  float drp[4], dzp[4], dnp[4], dnr[4], dhz[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float dh = g[e] + c[e];
    const float dn = dh * (1.f - z[e]);
    const float dz = dh * (hp[e] - n[e]);
    dnp[e] = dn * (1.f - n[e] * n[e]);
    drp[e] = dnp[e] * hn[e] * r[e] * (1.f - r[e]);
    dzp[e] = dz * z[e] * (1.f - z[e]);
    dnr[e] = dnp[e] * r[e];
    dhz[e] = dh * z[e];
  }
  T* ox = dgx + (long)b * 3 * H + j;
  T* oh = dgh + (long)b * 3 * H + j;
  store4<T>(ox, drp); store4<T>(ox + H, dzp); store4<T>(ox + 2 * H, dnp);
  store4<T>(oh, drp); store4<T>(oh + H, dzp); store4<T>(oh + 2 * H, dnr);
  store4<float>(dhc + (long)b * H + j, dhz);
}

// LSTM BPTT gate gradients: dg is shared by the x- and h-projections (all four gates are sums)
template <typename T>
__global__ __launch_bounds__(256) void lstm_bwd_gates_kernel(RnnBwdBatch batch, int H) {
  const RnnBwdCell& cl = batch.c[blockIdx.y];
  const float* __restrict__ dy = cl.dy; float* __restrict__ dhc = cl.dhc; float* __restrict__ dcc = cl.dcc;
  const T* __restrict__ cache = reinterpret_cast<const T*>(cl.cache);
  const T* __restrict__ cnew = reinterpret_cast<const T*>(cl.cnew);
  const T* __restrict__ cprev = reinterpret_cast<const T*>(cl.cprev);
  T* __restrict__ dg = reinterpret_cast<T*>(cl.dgx);
  const int Bt = cl.Bt;
  const int total = Bt * (H / 4);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int b = i / (H / 4), j = (i - b * (H / 4)) * 4;
  float g[4], ch[4], cc[4], ig[4], fg[4], gg[4], og[4], cn[4], cp[4] = {0.f, 0.f, 0.f, 0.f};
  load4<float>(dy + (long)b * H + j, g);
  load4<float>(dhc + (long)b * H + j, ch);
  load4<float>(dcc + (long)b * H + j, cc);
  const T* cr = cache + (long)b * 4 * H + j;
  load4<T>(cr, ig); load4<T>(cr + H, fg); load4<T>(cr + 2 * H, gg); load4<T>(cr + 3 * H, og);
  load4<T>(cnew + (long)b * H + j, cn);
  if (cprev) load4<T>(cprev + (long)b * H + j, cp);
  float di[4], df[4], dgg[4], dob[4], dcp[4], zero[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float dh = g[e] + ch[e];
    const float tc = tanhf(cn[e]);
    const float dc = cc[e] + dh * og[e] * (1.f - tc * tc);
    dob[e] = dh * tc * og[e] * (1.f - og[e]);
    di[e] = dc * gg[e] * ig[e] * (1.f - ig[e]);
    df[e] = dc * cp[e] * fg[e] * (1.f - fg[e]);
    dgg[e] = dc * ig[e] * (1.f - gg[e] * gg[e]);
    dcp[e] = dc * fg[e];
  }
  T* o = dg + (long)b * 4 * H + j;
  store4<T>(o, di); store4<T>(o + H, df); store4<T>(o + 2 * H, dgg); store4<T>(o + 3 * H, dob);
  store4<float>(dcc + (long)b * H + j, dcp);
  store4<float>(dhc + (long)b * H + j, zero);   // dh_{t-1} is entirely the GEMM term, accumulated next
}

// ---------------------------------------------------------------------------------------
// cross entropy over rows (mean reduction):  loss += -log softmax(x)[t] / nrows
// dlogits = (softmax - onehot) * gscale   written in TD (may alias the logits when TD == TL)
// one block per row; the row is read twice from L2 (max+sum online, then gradient)
// ---------------------------------------------------------------------------------------
template <typename TL, typename TD>
__global__ __launch_bounds__(256) void ce_kernel(const TL* __restrict__ logits, const long* __restrict__ target,
                                                 float* __restrict__ loss, TD* __restrict__ dlogits,
                                                 int V, int ldl, int ldd, float inv_rows, float gscale, const float* __restrict__ gscale_dev) {
  __shared__ float sm[8], ss[8];
  const int row = blockIdx.x;
  const TL* x = logits + (long)row * ldl;
  float m = -INFINITY, s = 0.f;
  for (int i = threadIdx.x; i < V; i += blockDim.x) {
    const float v = to_f32<TL>(x[i]);
    const float nm = fmaxf(m, v);
    s = s * __expf(m - nm) + __expf(v - nm);
    m = nm;
  }
  // combine (m, s) pairs across the wave, then across waves
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(m, o, 64), os = __shfl_xor(s, o, 64);
    const float nm = fmaxf(m, om);
    s = (m == -INFINITY ? 0.f : s * __expf(m - nm)) + (om == -INFINITY ? 0.f : os * __expf(om - nm));
    m = nm;
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { sm[wid] = m; ss[wid] = s; }
  __syncthreads();
  float M = sm[0], S = ss[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
    const float nm = fmaxf(M, sm[w]);
    S = S * __expf(M - nm) + ss[w] * __expf(sm[w] - nm);
    M = nm;
  }
  long t = target[row];
  t = t < 0 ? 0 : (t >= V ? V - 1 : t);
  const float lse = M + __logf(S);
  const float xt = to_f32<TL>(x[t]);
  __syncthreads();   // dlogits may alias the logits: every read of x[t] happens before any write
  if (threadIdx.x == 0 && loss) atomicAdd(loss, (lse - xt) * inv_rows);
  if (dlogits) {
    if (gscale_dev) gscale *= *gscale_dev;
    TD* d = dlogits + (long)row * ldd;
    for (int i = threadIdx.x; i < V; i += blockDim.x) {
      const float p = __expf(to_f32<TL>(x[i]) - lse);
      d[i] = from_f32<TD>((p - (i == t ? 1.f : 0.f)) * gscale);
    }
    for (int i = V + threadIdx.x; i < ldd; i += blockDim.x) d[i] = from_f32<TD>(0.f);   // pad columns feed GEMMs as K
  }
}

// out[n] += sum_rows x[row][n]      (bias gradients); one thread per column, coalesced over n
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, int rows, int cols, int ldx, int rows_per_block) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= cols) return;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += to_f32<T>(x[(long)r * ldx + n]);
  atomicAdd(out + n, s);
}

}  // namespace

// ---------------------------------------------------------------------------------------
// internal launchers (declared in rnn_kernels.h)
// ---------------------------------------------------------------------------------------
int rnn_gemm_launch_batch(const RnnGemmArgs* cells, int ncells, int dtype, int epi, int has_x, hipStream_t st) {
  ST_CHECK(ncells >= 0 && ncells <= kRnnBatch, "rnn_gemm: %d cells in one launch (max %d)", ncells, kRnnBatch);
  RnnGemmBatch b;
  memset(&b, 0, sizeof(b));
  const int epc = dtype == ST_BF16 ? 8 : 4;
  int nc = 0, maxM = 0, maxN = 0;
  for (int i = 0; i < ncells; ++i) {
    const RnnGemmArgs& a = cells[i];
    if (a.M <= 0) continue;
    ST_CHECK(epi == 3 || a.N % 4 == 0, "rnn_gemm: N=%d must be a multiple of 4", a.N);
    ST_CHECK(a.K % epc == 0 && a.lda % epc == 0 && a.ldw % epc == 0, "rnn_gemm: K/lda/ldw must be multiples of %d", epc);
    if (has_x) ST_CHECK(a.K2 % epc == 0 && a.lda2 % epc == 0 && a.ldw2 % epc == 0, "rnn_gemm: K2/lda2/ldw2 must be multiples of %d", epc);
    b.c[nc++] = a;
    if (a.M > maxM) maxM = a.M;
    if (a.N > maxN) maxN = a.N;
  }
  if (nc == 0) return 0;
  const dim3 grid((maxN + 15) / 16, (maxM + 15) / 16, nc), block(256);
  // many cells in one launch: the weights are most of the L2 traffic -> two row tiles per block share them
  // ... and so do the 16-row tiles of one tall cell (beam search steps 1280 rows at a time: 80 row tiles re-read the weights)
  const bool mt2 = (nc >= 2 || maxM >= 512) && maxM > 16 && epi != 3;
  const dim3 grid2((maxN + 15) / 16, (maxM + 31) / 32, nc);
#define RG(T, NG, EPI, HX) do { if (mt2) hipLaunchKernelGGL((rnn_gemm_kernel<T, NG, EPI, HX, 2>), grid2, block, 0, st, b); \
                               else hipLaunchKernelGGL((rnn_gemm_kernel<T, NG, EPI, HX, 1>), grid, block, 0, st, b); } while (0)
  if (dtype == ST_BF16) {
    if (epi == 0) RG(bf16_t, 1, 0, false);
    else if (epi == 3) RG(bf16_t, 1, 3, false);
    else if (epi == 1) { if (has_x) RG(bf16_t, 3, 1, true); else RG(bf16_t, 3, 1, false); }
    else { if (has_x) RG(bf16_t, 4, 2, true); else RG(bf16_t, 4, 2, false); }
  } else {
    if (epi == 0) RG(float, 1, 0, false);
    else if (epi == 3) RG(float, 1, 3, false);
    else if (epi == 1) { if (has_x) RG(float, 3, 1, true); else RG(float, 3, 1, false); }
    else { if (has_x) RG(float, 4, 2, true); else RG(float, 4, 2, false); }
  }
#undef RG
  ST_LAUNCH_CHECK();
  return 0;
}

int rnn_gemm_launch(const RnnGemmArgs& a, int dtype, int epi, int has_x, hipStream_t st) {
  return rnn_gemm_launch_batch(&a, 1, dtype, epi, has_x, st);
}

int rnn_bwd_gates_launch_batch(const RnnBwdCell* cells, int ncells, int H, int cell_kind, int dtype, hipStream_t st) {
  ST_CHECK(ncells >= 0 && ncells <= kRnnBatch, "rnn_bwd_gates: %d cells in one launch (max %d)", ncells, kRnnBatch);
  RnnBwdBatch b;
  memset(&b, 0, sizeof(b));
  int nc = 0, maxB = 0;
  for (int i = 0; i < ncells; ++i) {
    if (cells[i].Bt <= 0) continue;
    b.c[nc++] = cells[i];
    if (cells[i].Bt > maxB) maxB = cells[i].Bt;
  }
  if (nc == 0) return 0;
  const dim3 grid((maxB * (H / 4) + 255) / 256, nc);
  if (cell_kind == ST_CELL_GRU) {
    if (dtype == ST_BF16) hipLaunchKernelGGL(gru_bwd_gates_kernel<bf16_t>, grid, dim3(256), 0, st, b, H);
    else hipLaunchKernelGGL(gru_bwd_gates_kernel<float>, grid, dim3(256), 0, st, b, H);
  } else {
    if (dtype == ST_BF16) hipLaunchKernelGGL(lstm_bwd_gates_kernel<bf16_t>, grid, dim3(256), 0, st, b, H);
    else hipLaunchKernelGGL(lstm_bwd_gates_kernel<float>, grid, dim3(256), 0, st, b, H);
  }
  ST_LAUNCH_CHECK();
  return 0;
}

static inline int grid1d(long work, int threads = 256) {
  long b = (work + threads - 1) / threads;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

int pack_inputs_launch(const void* feat, const void* emb, const long* cap, int Tcap, const int* rows_b, const int* rows_t,
                       void* x0, long* target, int ntok, int E, int V, int mode, int dtype, hipStream_t st) {
  if (ntok <= 0) return 0;
  const int n = dtype == ST_BF16 ? 8 : 4;
  ST_CHECK(E % n == 0, "pack_inputs: E=%d must be a multiple of %d", E, n);
  const int grid = grid1d((long)ntok * (E / n));
  if (dtype == ST_BF16) hipLaunchKernelGGL(pack_inputs_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)feat, (const bf16_t*)emb, cap, Tcap, rows_b, rows_t, (bf16_t*)x0, target, ntok, E, V, mode);
  else hipLaunchKernelGGL(pack_inputs_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)feat, (const float*)emb, cap, Tcap, rows_b, rows_t, (float*)x0, target, ntok, E, V, mode);
  ST_LAUNCH_CHECK();
  return 0;
}

int embedding_bwd_launch(const float* dx0, const long* cap, int Tcap, const int* rows_b, const int* rows_t,
                         float* dfeat, float* demb, int ntok, int E, int V, int mode, hipStream_t st) {
  if (ntok <= 0) return 0;
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3(grid1d((long)ntok * E)), dim3(256), 0, st, dx0, cap, Tcap, rows_b, rows_t, dfeat, demb, ntok, E, V, mode);
  ST_LAUNCH_CHECK();
  return 0;
}

int gather_hprev_launch(const void* y, const int* rows_t, const int* prev_row, void* hp, int ntok, int H, int dtype, hipStream_t st,
                        const void* h0, const int* rows_b) {
  if (ntok <= 0) return 0;
  const int n = dtype == ST_BF16 ? 8 : 4;
  ST_CHECK(H % n == 0, "gather_hprev: H=%d must be a multiple of %d", H, n);
  const int grid = grid1d((long)ntok * (H / n));
  if (dtype == ST_BF16) hipLaunchKernelGGL(gather_hprev_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)y, rows_t, prev_row, (bf16_t*)hp, ntok, H, (const bf16_t*)h0, rows_b);
  else hipLaunchKernelGGL(gather_hprev_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)y, rows_t, prev_row, (float*)hp, ntok, H, (const float*)h0, rows_b);
  ST_LAUNCH_CHECK();
  return 0;
}

int gru_bwd_gates_launch(const float* dy, float* dhc, const void* cache, const void* hprev, void* dgx, void* dgh,
                         int Bt, int H, int dtype, hipStream_t st) {
  RnnBwdCell c;
  memset(&c, 0, sizeof(c));
  c.dy = dy; c.dhc = dhc; c.cache = cache; c.hprev = hprev; c.dgx = dgx; c.dgh = dgh; c.Bt = Bt;
  return rnn_bwd_gates_launch_batch(&c, 1, H, ST_CELL_GRU, dtype, st);
}

int lstm_bwd_gates_launch(const float* dy, float* dhc, float* dcc, const void* cache, const void* cnew, const void* cprev,
                          void* dg, int Bt, int H, int dtype, hipStream_t st) {
  RnnBwdCell c;
  memset(&c, 0, sizeof(c));
  c.dy = dy; c.dhc = dhc; c.dcc = dcc; c.cache = cache; c.cnew = cnew; c.cprev = cprev; c.dgx = dg; c.Bt = Bt;
  return rnn_bwd_gates_launch_batch(&c, 1, H, ST_CELL_LSTM, dtype, st);
}

int colsum_launch(const void* x, float* out, int rows, int cols, int ldx, int dtype, hipStream_t st) {
  if (rows <= 0 || cols <= 0) return 0;
  const int rpb = 64;
  const dim3 grid((cols + 255) / 256, (rows + rpb - 1) / rpb);
  if (dtype == ST_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, out, rows, cols, ldx, rpb);
  else hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, st, (const float*)x, out, rows, cols, ldx, rpb);
  ST_LAUNCH_CHECK();
  return 0;
}

extern "C" int st_cross_entropy(const void* logits, int logits_dtype, const long* target, int rows, int V, int ldl,
                                float* loss_accum, void* dlogits, int dlogits_dtype, int ldd, float grad_scale,
                                const float* grad_scale_dev, void* stream) {
  ST_CHECK(logits && target, "st_cross_entropy: null pointer");
  ST_CHECK(rows >= 0 && V > 0 && ldl >= V, "st_cross_entropy: bad shape");
  if (rows == 0) return 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const float inv = 1.0f / rows, gs = grad_scale / rows;
  const dim3 grid(rows), block(256);
  if (logits_dtype == ST_F32 && (dlogits_dtype == ST_F32 || !dlogits))
    hipLaunchKernelGGL((ce_kernel<float, float>), grid, block, 0, st, (const float*)logits, target, loss_accum, (float*)dlogits, V, ldl, ldd, inv, gs, grad_scale_dev);
  else if (logits_dtype == ST_F32)
    hipLaunchKernelGGL((ce_kernel<float, bf16_t>), grid, block, 0, st, (const float*)logits, target, loss_accum, (bf16_t*)dlogits, V, ldl, ldd, inv, gs, grad_scale_dev);
  else if (dlogits_dtype == ST_BF16 || !dlogits)
    hipLaunchKernelGGL((ce_kernel<bf16_t, bf16_t>), grid, block, 0, st, (const bf16_t*)logits, target, loss_accum, (bf16_t*)dlogits, V, ldl, ldd, inv, gs, grad_scale_dev);
  else
    hipLaunchKernelGGL((ce_kernel<bf16_t, float>), grid, block, 0, st, (const bf16_t*)logits, target, loss_accum, (float*)dlogits, V, ldl, ldd, inv, gs, grad_scale_dev);
  ST_LAUNCH_CHECK();
  return 0;
}
