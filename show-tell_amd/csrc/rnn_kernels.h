// Internal (non-ABI) launch interface of the recurrent-decoder kernels.
#pragma once
#include "common.h"

struct RnnGemmArgs {
  // acc[g][m][n] = sum_k A[m][k] W[g*gstride + n][k]  (+ second operand pair A2/W2 when HAS_X)
  const void* A; const void* W; int M, N, K, lda, ldw, gstride;
  const void* A2; const void* W2; int K2, lda2, ldw2;
  // EPI 0
  float* out_f32; int ldo; int accumulate;
  // gate epilogues
  const float* bias_h; const float* bias_x;
  const void* gx; int ldgx;            // precomputed x-projection (+b_ih) rows of this step
  const void* hprev; const void* cprev; int ldhp;
  void* hout; void* cout; int ldho;
  void* hout2; int ldho2;              // optional second copy of h' (decode: running state + layer output)
  void* cache; int ldcache;            // saved gates for BPTT (NULL at inference)
  unsigned long long* argmax_keys;     // EPI 3: per-row packed (value, index) maxima
  // greedy decode, layer 0: row m of the x operand is A2 + token(x_keys[m]) * lda2 (A2 = the embedding table), i.e. the
  // embedding gather of the previous step's arg-max happens inside the cell; block column 0 also writes the token ids
  const unsigned long long* x_keys; int x_V;
  long* ids_out; int ids_stride, ids_t;
  // split decode step (HAS_X = false forms of EPI 1 / 2): the recurrent half W_hh h + b_hh of a cell is produced OFF the
  // critical chain one launch earlier (raw_out cell: the NG gate sums + bias_h go to out_f32[m][g * N + n], fp32) and handed to
  // the cell as `gh`; the cell's MFMA operand pair (A, W) is then the INPUT half (x, W_ih), bias_h = b_ih, and x_keys gathers
  // the rows of A.  Half the operand bytes and MFMAs per launch on the chain.
  // No new fields (16 cells must fit the 4 KiB kernel-argument segment): in these forms `accumulate` carries the mode
  // (kCellRawOut / kCellSplit) and (gx, ldgx) point at the fp32 gh rows.
};
constexpr int kCellRawOut = 1, kCellSplit = 2;

// Up to kRnnBatch independent cells in ONE launch (blockIdx.z picks the cell): the (layer, time) wavefront of the
// teacher-forced decoder, where the cells of a diagonal layer + time = d do not depend on each other.
constexpr int kRnnBatch = 16;
struct RnnGemmBatch { RnnGemmArgs c[kRnnBatch]; };
static_assert(sizeof(RnnGemmBatch) <= 4000, "kernel-argument segment is 4 KiB");
struct RnnBwdCell {                    // BPTT gate gradients of one (layer, time) cell
  const float* dy; float* dhc; float* dcc; const void* cache; const void* hprev; const void* cnew; const void* cprev;
  void* dgx; void* dgh; int Bt;
};
struct RnnBwdBatch { RnnBwdCell c[kRnnBatch]; };

// One GRU unit (torch.nn.GRU, gate order r, z, n; reference rnn.py:32 / rnn.py:49):  x* = W_i* x + b_i*, h* = W_h* h + b_h*,
//   r = s(xr + hr), z = s(xz + hz), n = tanh(xn + r hn), h' = (1 - z) n + z h.
// ONE definition with explicit fused multiply-adds for every kernel that evaluates a cell (rnn_gemm_kernel, decode_pipe_kernel): the
// launch-chain decoder and the pipelined decoder must produce the same bits (-ffp-contract=fast would otherwise fuse per call site).
#ifdef __HIPCC__
__device__ __forceinline__ float st_sigm(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float st_gru_unit(float xr, float xz, float xn, float hr, float hz, float hn, float hprev, float& r, float& z, float& n) {
  r = st_sigm(xr + hr);
  z = st_sigm(xz + hz);
  n = tanhf(__builtin_fmaf(r, hn, xn));
  return __builtin_fmaf(z, hprev, (1.f - z) * n);
}
// One LSTM unit (torch.nn.LSTM, gate order i, f, g, o; reference LSTM/rnn_lstm.py:30): pre* = W_i* x + b_i* + W_h* h + b_h*,
//   c' = s(f) c + s(i) tanh(g), h' = s(o) tanh(c').  Same purpose as st_gru_unit: one definition, explicit FMA.
__device__ __forceinline__ float st_lstm_unit(float pi, float pf, float pg, float po, float cprev, float& ig, float& fg, float& gg, float& og, float& cn) {
  ig = st_sigm(pi); fg = st_sigm(pf); gg = tanhf(pg); og = st_sigm(po);
  cn = __builtin_fmaf(fg, cprev, ig * gg);
  return og * tanhf(cn);
}
#endif

// the pipelined (layer-per-XCD) greedy decoder, csrc/decode_pipe.hip: bytes it needs behind the launch chain's workspace (0: the
// configuration stays on the launch chain) and the run itself (0: done, 1: error, 2: not run / gave up -- use the launch chain)
size_t rnn_greedy_pipe_bytes(const st_rnn_params* p, int B, int steps);
int rnn_greedy_pipe(const st_rnn_params* p, const void* feat, int B, int steps, void* ws, size_t ws_bytes, long* ids_out, hipStream_t st);

// vocabulary projection + cross entropy without a logits tensor, csrc/vocab_ce.hip
int vocab_ce_supported(int dtype, int H);
int vocab_ce_tiles(int V);
int vocab_ce_forward(const void* y, const void* w, const float* bias, const long* target, int n, int V, float* partial, float* tgt,
                     float* lse, float* loss, hipStream_t st);
int vocab_ce_dlogits(const void* y, const void* w, const float* bias, const long* target, const float* lse, int n, int V,
                     void* dlogits, int ldd, float gscale, const float* gscale_dev, hipStream_t st);

int rnn_gemm_launch(const RnnGemmArgs& a, int dtype, int epi, int has_x, hipStream_t st);
int rnn_gemm_launch_batch(const RnnGemmArgs* cells, int ncells, int dtype, int epi, int has_x, hipStream_t st);
int rnn_bwd_gates_launch_batch(const RnnBwdCell* cells, int ncells, int H, int cell_kind, int dtype, hipStream_t st);
int pack_inputs_launch(const void* feat, const void* emb, const long* cap, int Tcap, const int* rows_b, const int* rows_t,
                       void* x0, long* target, int ntok, int E, int V, int mode, int dtype, hipStream_t st);
int embedding_bwd_launch(const float* dx0, const long* cap, int Tcap, const int* rows_b, const int* rows_t,
                         float* dfeat, float* demb, int ntok, int E, int V, int mode, hipStream_t st);
int gather_hprev_launch(const void* y, const int* rows_t, const int* prev_row, void* hp, int ntok, int H, int dtype, hipStream_t st,
                        const void* h0 = nullptr, const int* rows_b = nullptr);
int gru_bwd_gates_launch(const float* dy, float* dhc, const void* cache, const void* hprev, void* dgx, void* dgh,
                         int Bt, int H, int dtype, hipStream_t st);
int lstm_bwd_gates_launch(const float* dy, float* dhc, float* dcc, const void* cache, const void* cnew, const void* cprev,
                          void* dg, int Bt, int H, int dtype, hipStream_t st);
int colsum_launch(const void* x, float* out, int rows, int cols, int ldx, int dtype, hipStream_t st);
