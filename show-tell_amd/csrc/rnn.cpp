// Teacher-forced decoder (GRU / LSTM over a packed sequence): forward and BPTT backward,
// each as ONE C-ABI call that issues every kernel of the pass on one stream.
//
// Replaces RNN.forward (reference rnn.py:27-35, LSTM/rnn_lstm.py:25-33) and the autograd
// backward torch builds for it (main.py:151).  Rows are time-major packed exactly as
// pack_padded_sequence orders them: row(t, b) = step_off[t] + b for b < batch_sizes[t].
//
// Both passes walk the (layer, time) grid by DIAGONALS: cell (l, t) needs (l-1, t) and (l, t-1) [forward] or
// (l+1, t) and (l, t+1) [backward], so the cells with l + t = d are independent and go out as ONE launch
// (blockIdx.z = cell).  A 5-layer, 18-step caption batch is 22 dependent launches instead of 90: the recurrence
// is launch-latency bound (a cell is ~0.2 GFLOP), so the count of dependent launches is what it costs.
//
// forward:  x0 = [feat ; emb(caption)] gathered straight into packed rows (no cat/pack copies)
//           per diagonal one fused launch: x W_ih^T + h W_hh^T + gates for every cell -> st_rnn_forward
//           logits = y_top W_lin^T + b
// backward: dlogits -> dW_lin, db_lin, dy_top; per diagonal (reversed) one gate-gradient launch and one skinny-GEMM
//           launch (dh_{t-1} += dgh_t W_hh and dx_t = dgx_t W_ih of every cell); then per layer two MFMA GEMMs
//           (dW_ih, dW_hh) whose K-major operands come from transposes that also sum the bias gradients
//           -> st_rnn_backward
#include "common.h"
#include "rnn_kernels.h"
#include <string.h>
#include <vector>

namespace {

inline size_t al(size_t v) { return (v + 255) & ~(size_t)255; }
inline int up8(int v) { return (v + 7) & ~7; }

struct Plan {
  int G, GH, Np, Vp, maxw;
  size_t es;
  size_t x0, y, gates, cst, dyl, dx0, dgx, dgh, dhc, dcc, tA, tB, wT, wThh, wTih, gA, gB, total;
};

Plan make_plan(const st_rnn_params* p, const st_packed_seq* s) {
  Plan q;
  q.G = p->cell == ST_CELL_GRU ? 3 : 4;
  q.GH = q.G * p->H;
  q.Np = up8(s->ntok);
  q.Vp = st_rnn_vocab_ld(p->V);
  q.es = st_dtype_size(p->dtype);
  q.maxw = p->in0 > p->H ? p->in0 : p->H;
  const size_t n = s->ntok, L = p->L, H = p->H;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += al(bytes); return r; };
  q.x0 = take(n * p->in0 * q.es);
  q.y = take(L * n * H * q.es);
  q.gates = take(L * n * 4 * H * q.es);
  q.cst = take(p->cell == ST_CELL_LSTM ? L * n * H * q.es : 0);
  q.dyl = take(L * n * H * sizeof(float));          // d loss / d y_l, one [ntok][H] fp32 matrix per layer
  q.dx0 = take(n * p->in0 * sizeof(float));         // d loss / d x0
  q.dgx = take(L * n * q.GH * q.es);                // gate gradients of every layer (operands of the dW GEMMs)
  q.dgh = take(p->cell == ST_CELL_GRU ? L * n * q.GH * q.es : 0);
  q.dhc = take(L * (size_t)s->B * H * sizeof(float));
  q.dcc = take(L * (size_t)s->B * H * sizeof(float));
  // transposed operands: tA <= max(V, GH) x Np ; tB <= max(H, in0) x Np ; wT <= max(H x Vp, maxw x GH)
  const size_t ra = (size_t)(p->V > q.GH ? p->V : q.GH);
  q.tA = take(ra * q.Np * q.es);
  q.tB = take((size_t)q.maxw * q.Np * q.es);
  const size_t w1 = (size_t)H * q.Vp, w2 = (size_t)q.maxw * q.GH;
  q.wT = take((w1 > w2 ? w1 : w2) * q.es);
  q.wThh = take(L * H * q.GH * q.es);               // W_hh^T and W_ih^T of every layer: the backward wavefront needs them all
  q.wTih = take(L * (size_t)q.maxw * q.GH * q.es);
  // K-major operands of the 2L weight-gradient GEMMs (dgx^T, x^T, dgh^T, hprev^T of every layer), alive together so that
  // the GEMMs can go out as one grouped launch
  q.gA = take(2 * L * (size_t)q.GH * q.Np * q.es);
  q.gB = take(2 * L * (size_t)q.maxw * q.Np * q.es);
  q.total = o;
  return q;
}

int gemm_nt(const void* a, int lda, const void* w, int ldw, void* y, int ldy, int M, int N, int K, int dtype, int out_dtype,
            const float* bias, int accumulate, void* stream, int split_k = 0) {
  if (M <= 0 || N <= 0) return 0;
  st_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.split_k = split_k;
  d.x = a; d.w = w; d.y = y; d.bias = bias; d.dtype = dtype; d.out_dtype = out_dtype;
  d.B = M; d.Hin = 1; d.Win = 1; d.Cin = K; d.Ho = 1; d.Wo = 1; d.N = N; d.KH = 1; d.KW = 1; d.stride = 1; d.pad = 0;
  d.ldx = lda; d.ldw = ldw; d.ldy = ldy; d.accumulate = accumulate;
  return st_conv(&d, stream);
}

int check_common(const st_rnn_params* p, const st_packed_seq* s, const char* who) {
  ST_CHECK(p && s, "%s: null descriptor", who);
  ST_CHECK(p->cell == ST_CELL_GRU || p->cell == ST_CELL_LSTM, "%s: bad cell %d", who, p->cell);
  ST_CHECK(p->dtype == ST_F32 || p->dtype == ST_BF16, "%s: bad dtype %d", who, p->dtype);
  ST_CHECK(p->L >= 1 && p->L <= ST_MAX_LAYERS, "%s: num_layers=%d out of range [1,%d]", who, p->L, ST_MAX_LAYERS);
  ST_CHECK(p->H % 8 == 0 && p->in0 % 8 == 0 && p->E % 8 == 0, "%s: E=%d, in0=%d and H=%d must be multiples of 8", who, p->E, p->in0, p->H);
  ST_CHECK(p->V > 0, "%s: bad vocabulary size", who);
  ST_CHECK(s->B > 0 && s->T > 0 && s->ntok > 0 && s->batch_sizes_host && s->rows_b && s->rows_t && s->prev_row && s->caption,
           "%s: bad packed-sequence descriptor", who);
  int sum = 0, prev = s->B;
  for (int t = 0; t < s->T; ++t) {
    const int b = s->batch_sizes_host[t];
    ST_CHECK(b > 0 && b <= prev, "%s: batch_sizes must be positive and non-increasing (captions sorted by length, utils.py:66)", who);
    prev = b; sum += b;
  }
  ST_CHECK(sum == s->ntok && s->batch_sizes_host[0] == s->B, "%s: batch_sizes do not add up to ntok", who);
  for (int l = 0; l < p->L; ++l)
    ST_CHECK(p->w_ih[l] && p->w_hh[l] && p->b_ih[l] && p->b_hh[l], "%s: null weights for layer %d", who, l);
  return 0;
}

}  // namespace

// Leading dimension to give logits / dlogits rows: V rounded up to 8 elements for small vocabularies, to 512 (8 K tiles of
// 64) from 2048 entries on, so that the K = V product of the backward pass can be split evenly.
extern "C" int st_rnn_vocab_ld(int V) { return V < 2048 ? up8(V) : (V + 511) / 512 * 512; }

extern "C" size_t st_rnn_workspace_bytes(const st_rnn_params* p, const st_packed_seq* s) {
  if (!p || !s) return 0;
  return make_plan(p, s).total;
}

extern "C" int st_rnn_forward(const st_rnn_params* p, const st_packed_seq* s, const void* x0_override, const void* feat,
                              void* workspace, size_t workspace_bytes, void* logits, int logits_dtype, int ldl,
                              long* targets, int save_for_backward, void* stream) {
  if (check_common(p, s, "st_rnn_forward")) return 1;
  ST_CHECK(workspace && (x0_override || (feat && p->emb)), "st_rnn_forward: null pointer");
  const Plan q = make_plan(p, s);
  ST_CHECK(workspace_bytes >= q.total, "st_rnn_forward: workspace too small (%zu < %zu)", workspace_bytes, q.total);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  char* ws = reinterpret_cast<char*>(workspace);
  const int dt = p->dtype, H = p->H, n = s->ntok;
  const size_t es = q.es;

  const void* x0 = x0_override;
  if (!x0) {
    ST_CHECK(p->in0 == p->E, "st_rnn_forward: in0 != E needs x0_override");
    if (pack_inputs_launch(feat, p->emb, s->caption, s->Tcap, s->rows_b, s->rows_t, ws + q.x0, targets, n, p->E, p->V, 0, dt, st)) return 1;
    x0 = ws + q.x0;
  }
  std::vector<int> off(s->T + 1, 0);
  for (int t = 0; t < s->T; ++t) off[t + 1] = off[t] + s->batch_sizes_host[t];

  const int L = p->L, T = s->T;
  for (int d = 0; d < T + L - 1; ++d) {
    RnnGemmArgs cells[ST_MAX_LAYERS];
    int nc = 0;
    for (int l = d < T ? 0 : d - (T - 1); l <= d && l < L; ++l) {
      const int t = d - l;
      const int bt = s->batch_sizes_host[t];
      const char* xl = l == 0 ? reinterpret_cast<const char*>(x0) : ws + q.y + (size_t)(l - 1) * n * H * es;
      const int in = l == 0 ? p->in0 : H;
      char* yl = ws + q.y + (size_t)l * n * H * es;
      char* gl = ws + q.gates + (size_t)l * n * 4 * H * es;
      char* cl = ws + q.cst + (size_t)l * n * H * es;
      RnnGemmArgs& a = cells[nc++];
      memset(&a, 0, sizeof(a));
      a.M = bt; a.N = H; a.K = H; a.lda = H; a.ldw = H; a.gstride = H;
      a.W = p->w_hh[l];
      a.A = t > 0 ? yl + (size_t)off[t - 1] * H * es : nullptr;
      a.hprev = a.A; a.ldhp = H;
      a.bias_h = p->b_hh[l];
      a.A2 = xl + (size_t)off[t] * in * es; a.K2 = in; a.lda2 = in; a.W2 = p->w_ih[l]; a.ldw2 = in; a.bias_x = p->b_ih[l];
      a.hout = yl + (size_t)off[t] * H * es; a.ldho = H;
      if (save_for_backward) { a.cache = gl + (size_t)off[t] * 4 * H * es; a.ldcache = 4 * H; }
      if (p->cell == ST_CELL_LSTM) {
        a.cprev = t > 0 ? cl + (size_t)off[t - 1] * H * es : nullptr;
        a.cout = cl + (size_t)off[t] * H * es;
      }
    }
    if (rnn_gemm_launch_batch(cells, nc, dt, p->cell == ST_CELL_GRU ? 1 : 2, 1, st)) return 1;
  }
  if (logits) {
    ST_CHECK(p->w_lin && p->b_lin, "st_rnn_forward: logits requested without the vocabulary projection");
    const char* ytop = ws + q.y + (size_t)(p->L - 1) * n * H * es;
    if (gemm_nt(ytop, H, p->w_lin, H, logits, ldl, n, p->V, H, dt, logits_dtype, p->b_lin, 0, stream)) return 1;
  }
  return 0;
}

// ---- vocabulary projection + nn.CrossEntropyLoss() fused (csrc/vocab_ce.hip): the logits are never written ---------------------------------
// scratch (floats): lse[ntok] | target logit[ntok] | partial[ntok][tiles][2]; lse stays valid for st_rnn_fused_dlogits
extern "C" int st_rnn_fused_loss_supported(const st_rnn_params* p) {
  return p && p->w_lin && p->b_lin && vocab_ce_supported(p->dtype, p->H) ? 1 : 0;
}
extern "C" size_t st_rnn_fused_loss_bytes(const st_rnn_params* p, const st_packed_seq* s) {
  if (!p || !s) return 0;
  return ((size_t)s->ntok * (2 + 2 * (size_t)vocab_ce_tiles(p->V))) * sizeof(float);
}
extern "C" int st_rnn_fused_loss(const st_rnn_params* p, const st_packed_seq* s, const void* workspace, size_t workspace_bytes,
                                 const long* targets, float* scratch, size_t scratch_bytes, float* loss_accum, void* stream) {
  if (check_common(p, s, "st_rnn_fused_loss")) return 1;
  ST_CHECK(workspace && targets && scratch && loss_accum, "st_rnn_fused_loss: null pointer");
  ST_CHECK(st_rnn_fused_loss_supported(p), "st_rnn_fused_loss: needs bf16, H = 512 and the vocabulary projection (use st_rnn_forward's logits + st_cross_entropy)");
  const Plan q = make_plan(p, s);
  ST_CHECK(workspace_bytes >= q.total && scratch_bytes >= st_rnn_fused_loss_bytes(p, s), "st_rnn_fused_loss: workspace / scratch too small");
  const char* ytop = reinterpret_cast<const char*>(workspace) + q.y + (size_t)(p->L - 1) * s->ntok * p->H * q.es;
  const int n = s->ntok;
  return vocab_ce_forward(ytop, p->w_lin, p->b_lin, targets, n, p->V, scratch + 2 * (size_t)n, scratch + n, scratch, loss_accum,
                          reinterpret_cast<hipStream_t>(stream));
}
// dlogits[ntok][ldd] (bf16) = (softmax - onehot) / ntok * *grad_scale_dev, from the same tile products; pad columns [V, ldd) zero
extern "C" int st_rnn_fused_dlogits(const st_rnn_params* p, const st_packed_seq* s, const void* workspace, size_t workspace_bytes,
                                    const long* targets, const float* scratch, const float* grad_scale_dev, void* dlogits, int ldd, void* stream) {
  if (check_common(p, s, "st_rnn_fused_dlogits")) return 1;
  ST_CHECK(workspace && targets && scratch && dlogits, "st_rnn_fused_dlogits: null pointer");
  ST_CHECK(st_rnn_fused_loss_supported(p), "st_rnn_fused_dlogits: unsupported configuration");
  const Plan q = make_plan(p, s);
  ST_CHECK(workspace_bytes >= q.total, "st_rnn_fused_dlogits: workspace too small");
  const char* ytop = reinterpret_cast<const char*>(workspace) + q.y + (size_t)(p->L - 1) * s->ntok * p->H * q.es;
  return vocab_ce_dlogits(ytop, p->w_lin, p->b_lin, targets, scratch, s->ntok, p->V, dlogits, ldd, 1.0f, grad_scale_dev,
                          reinterpret_cast<hipStream_t>(stream));
}

extern "C" int st_rnn_backward(const st_rnn_params* p, const st_rnn_grads* g, const st_packed_seq* s,
                               const void* x0_override, const void* dlogits, int ldd, const float* dy_top_extra,
                               void* workspace, size_t workspace_bytes, float* dfeat, float* dx0_out, void* stream) {
  if (check_common(p, s, "st_rnn_backward")) return 1;
  ST_CHECK(g && workspace, "st_rnn_backward: null pointer");
  const Plan q = make_plan(p, s);
  ST_CHECK(workspace_bytes >= q.total, "st_rnn_backward: workspace too small (%zu < %zu)", workspace_bytes, q.total);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  char* ws = reinterpret_cast<char*>(workspace);
  const int dt = p->dtype, H = p->H, n = s->ntok, Np = q.Np, GH = q.GH;
  const size_t es = q.es;
  std::vector<int> off(s->T + 1, 0);
  for (int t = 0; t < s->T; ++t) off[t + 1] = off[t] + s->batch_sizes_host[t];
  const int L = p->L, T = s->T;
  auto dyl = [&](int l) { return reinterpret_cast<float*>(ws + q.dyl) + (size_t)l * n * H; };   // d loss / d y_l
  float* dy = dyl(L - 1);
  const char* ytop = ws + q.y + (size_t)(p->L - 1) * n * H * es;

  if (dlogits) {
    ST_CHECK(p->w_lin && g->w_lin && g->b_lin, "st_rnn_backward: vocabulary projection gradients requested without buffers");
    ST_CHECK(ldd >= up8(p->V) && ldd % 8 == 0 && ldd <= q.Vp, "st_rnn_backward: dlogits leading dimension %d must be a multiple of 8 in [%d, %d]",
             ldd, up8(p->V), q.Vp);
    // db = colsum(dlogits);  dW_lin += dlogits^T y_top;  dy_top = dlogits W_lin
    if (st_transpose_colsum(dlogits, ws + q.tA, g->b_lin, dt, n, p->V, ldd, Np, stream)) return 1;
    if (st_transpose(ytop, ws + q.tB, dt, n, H, H, Np, stream)) return 1;
    if (gemm_nt(ws + q.tA, Np, ws + q.tB, Np, g->w_lin, H, p->V, H, Np, dt, ST_F32, nullptr, 1, stream)) return 1;
    // dy = dlogits W_lin is 15 x 4 output tiles over K = V: with the padded leading dimension (st_rnn_vocab_ld: a multiple
    // of 8 K tiles, pad columns zero on both sides) it runs as 8 K slices in one grouped launch
    if (st_transpose(p->w_lin, ws + q.wT, dt, p->V, H, H, ldd, stream)) return 1;
    const int bk = dt == ST_BF16 ? 64 : 32;
    const int split = (ldd % (8 * bk) == 0 && ldd >= 16 * bk) ? 8 : 0;
    if (gemm_nt(dlogits, ldd, ws + q.wT, ldd, dy, H, n, H, ldd, dt, ST_F32, nullptr, 0, stream, split)) return 1;
    if (dy_top_extra) { st_set_error("st_rnn_backward: dlogits and dy_top_extra are exclusive"); return 1; }
  } else {
    ST_CHECK(dy_top_extra, "st_rnn_backward: need dlogits or dy_top");
    if (hipMemcpyAsync(dy, dy_top_extra, (size_t)n * H * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
      st_set_error("st_rnn_backward: copy failed"); return 1;
    }
  }

  const void* x0 = x0_override ? x0_override : ws + q.x0;
  const bool need_dx0 = dfeat || dx0_out || g->emb;
  float* dx0 = dx0_out ? dx0_out : reinterpret_cast<float*>(ws + q.dx0);
  const bool gru = p->cell == ST_CELL_GRU;
  auto dgxl = [&](int l) { return ws + q.dgx + (size_t)l * n * GH * es; };
  auto dghl = [&](int l) { return gru ? ws + q.dgh + (size_t)l * n * GH * es : dgxl(l); };   // LSTM: one gradient feeds both projections
  auto dhcl = [&](int l) { return reinterpret_cast<float*>(ws + q.dhc) + (size_t)l * s->B * H; };
  auto dccl = [&](int l) { return reinterpret_cast<float*>(ws + q.dcc) + (size_t)l * s->B * H; };
  auto wThh = [&](int l) { return ws + q.wThh + (size_t)l * H * GH * es; };
  auto wTih = [&](int l) { return ws + q.wTih + (size_t)l * q.maxw * GH * es; };
  if (hipMemsetAsync(ws + q.dhc, 0, (size_t)L * s->B * H * sizeof(float), st) != hipSuccess ||
      hipMemsetAsync(ws + q.dcc, 0, (size_t)L * s->B * H * sizeof(float), st) != hipSuccess) { st_set_error("memset failed"); return 1; }
  {   // K-major operands of dh_{t-1} += dgh_t W_hh  and  dx_t = dgx_t W_ih: one batched transpose per kind
    const void* wx[2][ST_MAX_LAYERS]; void* wy[2][ST_MAX_LAYERS];
    int ni = 0;
    for (int l = 0; l < L; ++l) {
      const int in = l == 0 ? p->in0 : H;
      wx[0][l] = p->w_hh[l]; wy[0][l] = wThh(l);
      if (!(l > 0 || need_dx0)) continue;
      if (in == H) { wx[1][ni] = p->w_ih[l]; wy[1][ni] = wTih(l); ++ni; }
      else if (st_transpose(p->w_ih[l], wTih(l), dt, GH, in, in, GH, stream)) return 1;
    }
    if (st_transpose_batch(wx[0], wy[0], nullptr, L, dt, GH, H, H, GH, nullptr, nullptr, stream)) return 1;
    if (ni && st_transpose_batch(wx[1], wy[1], nullptr, ni, dt, GH, H, H, GH, nullptr, nullptr, stream)) return 1;
  }
  // reversed wavefront: the cells of a diagonal need only cells of the diagonal above
  for (int d = T + L - 2; d >= 0; --d) {
    RnnBwdCell gc[ST_MAX_LAYERS];
    RnnGemmArgs mc[2 * ST_MAX_LAYERS];
    int ng = 0, nm = 0;
    for (int l = d < T ? 0 : d - (T - 1); l <= d && l < L; ++l) {
      const int t = d - l;
      const int bt = s->batch_sizes_host[t];
      const int in = l == 0 ? p->in0 : H;
      char* yl = ws + q.y + (size_t)l * n * H * es;
      char* gl = ws + q.gates + (size_t)l * n * 4 * H * es;
      char* cl = ws + q.cst + (size_t)l * n * H * es;
      RnnBwdCell& c = gc[ng++];
      memset(&c, 0, sizeof(c));
      c.dy = dyl(l) + (size_t)off[t] * H; c.dhc = dhcl(l); c.dcc = dccl(l);
      c.cache = gl + (size_t)off[t] * 4 * H * es;
      c.hprev = t > 0 ? yl + (size_t)off[t - 1] * H * es : nullptr;
      c.cnew = cl + (size_t)off[t] * H * es;
      c.cprev = t > 0 ? cl + (size_t)off[t - 1] * H * es : nullptr;
      c.dgx = dgxl(l) + (size_t)off[t] * GH * es; c.dgh = dghl(l) + (size_t)off[t] * GH * es; c.Bt = bt;
      if (t > 0) {                                 // dh_{t-1} += dgh_t W_hh
        RnnGemmArgs& a = mc[nm++];
        memset(&a, 0, sizeof(a));
        a.A = dghl(l) + (size_t)off[t] * GH * es; a.W = wThh(l); a.M = bt; a.N = H; a.K = GH; a.lda = GH; a.ldw = GH;
        a.out_f32 = dhcl(l); a.ldo = H; a.accumulate = 1;
      }
      if (l > 0 || need_dx0) {                     // dx_t = dgx_t W_ih: the gradient the layer below reads at this step
        RnnGemmArgs& a = mc[nm++];
        memset(&a, 0, sizeof(a));
        a.A = dgxl(l) + (size_t)off[t] * GH * es; a.W = wTih(l); a.M = bt; a.N = in; a.K = GH; a.lda = GH; a.ldw = GH;
        a.out_f32 = (l > 0 ? dyl(l - 1) + (size_t)off[t] * H : dx0 + (size_t)off[t] * in); a.ldo = in;
      }
    }
    if (rnn_bwd_gates_launch_batch(gc, ng, H, p->cell, dt, st)) return 1;
    if (rnn_gemm_launch_batch(mc, nm, dt, 0, 0, st)) return 1;
  }
  // parameter gradients: K-major copies of every layer's operands (the transposes also sum the bias gradients), then the
  // 2L weight-gradient GEMMs -- 48 tiles each -- as grouped launches (st_conv_batch) that fill the chip
  st_conv_desc gd[2 * ST_MAX_LAYERS];
  int ng = 0;
  auto wgrad = [&](const void* aT, const void* bT, float* dw, int rows, int cols) {   // dw[rows][cols] += aT[rows][Np] . bT[cols][Np]^T
    st_conv_desc& d = gd[ng++];
    memset(&d, 0, sizeof(d));
    d.x = aT; d.w = bT; d.y = dw; d.dtype = dt; d.out_dtype = ST_F32;
    d.B = rows; d.Hin = 1; d.Win = 1; d.Cin = Np; d.Ho = 1; d.Wo = 1; d.N = cols; d.KH = 1; d.KW = 1; d.stride = 1; d.pad = 0;
    d.ldx = Np; d.ldw = Np; d.ldy = cols; d.accumulate = 1;
  };
  const void* tx[4][ST_MAX_LAYERS]; void* ty[4][ST_MAX_LAYERS]; float* tc[4][ST_MAX_LAYERS];   // dgx, x, dgh, y(->hprev) of every layer
  int nx = 0;                                                                                   // layers whose input width is H
  for (int l = 0; l < L; ++l) {
    const int in = l == 0 ? p->in0 : H;
    const char* xl = l == 0 ? reinterpret_cast<const char*>(x0) : ws + q.y + (size_t)(l - 1) * n * H * es;
    char* aIh = ws + q.gA + (size_t)(2 * l) * GH * Np * es;
    char* aHh = ws + q.gA + (size_t)(2 * l + 1) * GH * Np * es;
    char* bIh = ws + q.gB + (size_t)(2 * l) * q.maxw * Np * es;
    char* bHh = ws + q.gB + (size_t)(2 * l + 1) * q.maxw * Np * es;
    tx[0][l] = dgxl(l); ty[0][l] = aIh; tc[0][l] = g->b_ih[l];
    tx[2][l] = dghl(l); ty[2][l] = aHh; tc[2][l] = g->b_hh[l];
    tx[3][l] = ws + q.y + (size_t)l * n * H * es; ty[3][l] = bHh; tc[3][l] = nullptr;
    if (in == H) {                       // same shape as the recurrent one: joins the groups
      tx[1][nx] = xl; ty[1][nx] = bIh; tc[1][nx] = nullptr; ++nx;
      wgrad(aIh, bIh, g->w_ih[l], GH, in);
    } else {
      if (st_transpose(xl, bIh, dt, n, in, in, Np, stream)) return 1;
    }
    wgrad(gru ? aHh : aIh, bHh, g->w_hh[l], GH, H);
    if (!gru && colsum_launch(dghl(l), g->b_hh[l], n, GH, GH, dt, st)) return 1;   // LSTM: dgh == dgx, transposed once
  }
  // one launch per operand kind over all layers; the bias gradients ride on the gate-gradient transposes and h_{t-1}
  // is gathered by the transpose itself (packed-sequence prev_row table)
  if (st_transpose_batch(tx[0], ty[0], tc[0], L, dt, n, GH, GH, Np, nullptr, nullptr, stream)) return 1;
  if (nx && st_transpose_batch(tx[1], ty[1], nullptr, nx, dt, n, H, H, Np, nullptr, nullptr, stream)) return 1;
  if (gru && st_transpose_batch(tx[2], ty[2], tc[2], L, dt, n, GH, GH, Np, nullptr, nullptr, stream)) return 1;
  if (st_transpose_batch(tx[3], ty[3], nullptr, L, dt, n, H, H, Np, s->rows_t, s->prev_row, stream)) return 1;
  for (int l = 0; l < L; ++l) {          // layers with a different input width: their dW_ih on its own
    const int in = l == 0 ? p->in0 : H;
    if (in != H && gemm_nt(ws + q.gA + (size_t)(2 * l) * GH * Np * es, Np, ws + q.gB + (size_t)(2 * l) * q.maxw * Np * es, Np,
                           g->w_ih[l], in, GH, in, Np, dt, ST_F32, nullptr, 1, stream)) return 1;
  }
  for (int i = 0; i < ng; i += 12) {
    if (st_conv_batch(gd + i, ng - i < 12 ? ng - i : 12, stream)) return 1;
  }
  if (need_dx0 && !x0_override) {
    ST_CHECK(g->emb, "st_rnn_backward: embedding gradient buffer missing");
    if (embedding_bwd_launch(dx0, s->caption, s->Tcap, s->rows_b, s->rows_t, dfeat, g->emb, n, p->E, p->V, 0, st)) return 1;
  }
  return 0;
}
