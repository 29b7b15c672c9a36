// Greedy caption decoding (reference rnn.py:37-58, LSTM/rnn_lstm.py:35-57): exactly `steps`
// iterations of  unit(x, h) -> linear -> max(1)[1] -> embeddings,  no early stop.
//
// Per step: one fused launch per layer (x W_ih^T and h W_hh^T accumulate in the same MFMA
// kernel, gates in its epilogue), one MFMA GEMM for the vocabulary projection and one
// arg-max + embedding-gather kernel.  Everything is issued from this one C call.
#include "common.h"
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

extern "C" size_t st_rnn_greedy_workspace_bytes(const st_rnn_params* p, int B) {
  if (!p || B <= 0) return 0;
  const size_t es = st_dtype_size(p->dtype);
  const size_t hb = al((size_t)p->L * B * p->H * es);
  return 4 * hb + al((size_t)B * p->E * es) + al((size_t)B * up8(p->V) * sizeof(float));
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
  const void* x = feat;
  int cur = 0;
  for (int t = 0; t < steps; ++t) {
    const int nxt = cur ^ 1;
    for (int l = 0; l < L; ++l) {
      RnnGemmArgs a;
      memset(&a, 0, sizeof(a));
      a.M = B; a.N = H; a.gstride = H;
      a.A2 = l == 0 ? x : hbuf[nxt] + (size_t)(l - 1) * B * H * es;
      a.W2 = p->w_ih[l]; a.K2 = l == 0 ? E : H; a.lda2 = a.K2; a.ldw2 = a.K2;
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
    float* lg = logits_out ? logits_out + (size_t)t * B * Vp : logits;
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
