// Soft-attention decoder (reference Attention/rnn_attn.py:33-145, rnn_attn_LSTM.py): teacher-forced forward,
// BPTT backward (incl. the doubly-stochastic regulariser of main_attn.py:131) and greedy decoding, each as one
// C-ABI call that issues every kernel of the pass.
//
// Differences from the reference's evaluation ORDER (never from its result):
//   * encoder_att(feat) is time-invariant and computed once per batch (reference: every step, rnn_attn.py:23);
//     its weight gradient is one GEMM over the time-summed d(att1);
//   * hidden state is kept [L][rows][H] (no per-step transposes, rnn_attn.py:70,74); predictions are written
//     straight into time-major packed rows (no (B,T,V) zero tensor + pack, rnn_attn.py:64,72,115).
//
// Backward algebra per step t (rows b < B_t), top layer index L-1, u = att1 + att2:
//   dx0 = [d emb(cap[:,t]) ; d ez]   ez = W_m z + b_m ;  dz = d ez W_m
//   d alpha_p = dz . feat_p + reg_p ; de = alpha (d alpha - <alpha, d alpha>) ; du_p = de_p w_f lrelu'(u_p)
//   d att2 = sum_p du_p ; d att1[b,p] += du_p ; dw_f += sum_p de_p lrelu(u_p) ; db_f += sum_p de_p
//   dh_{t-1}[L-1] += d att2 W_d      (attention at step t is keyed on the PREVIOUS top-layer state)
#include "common.h"
#include "rnn_kernels.h"
#include "attn_kernels.h"
#include <string.h>
#include <vector>

namespace {

inline size_t al(size_t v) { return (v + 255) & ~(size_t)255; }
inline int up8(int v) { return (v + 7) & ~7; }
inline size_t mx(size_t a, size_t b) { return a > b ? a : b; }

struct Plan {
  int G, GH, Np, Vp, BP;
  size_t es;
  size_t feat, mean, h0, c0, att1, xp, y, gates, cst, z, att2, tokT;
  size_t dytop, dxa, dxb, dhc, dcc, dgx, dgh, dez, datt2p, datt2, dz, datt1, dalpha, hprev, tA, tB, wThh, wTih, wTmisc, cast, total;
};

Plan make_plan(const st_attn_params* p, const st_packed_seq* s) {
  const st_rnn_params& r = p->rnn;
  Plan q;
  q.G = r.cell == ST_CELL_GRU ? 3 : 4;
  q.GH = q.G * r.H;
  q.BP = s->B * p->P;
  q.Np = up8(s->ntok > q.BP ? s->ntok : q.BP);
  q.Vp = up8(r.V);
  q.es = st_dtype_size(r.dtype);
  const size_t n = s->ntok, L = r.L, H = r.H, E = r.E, B = s->B, es = q.es;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t x = o; o += al(bytes); return x; };
  q.feat = take((size_t)q.BP * p->F * es);
  q.mean = take(B * p->F * es);
  q.h0 = take(B * H * es);
  q.c0 = take(B * H * es);
  q.att1 = take((size_t)q.BP * p->A * es);
  q.xp = take(n * 2 * E * es);
  q.y = take(L * n * H * es);
  q.gates = take(L * n * 4 * H * es);
  q.cst = take(r.cell == ST_CELL_LSTM ? L * n * H * es : 0);
  q.z = take(n * p->F * es);
  q.att2 = take(n * p->A * sizeof(float));
  q.tokT = 0;
  // backward
  q.dytop = take(n * H * sizeof(float));
  const size_t wmax = mx(H, 2 * E);
  q.dxa = take(B * wmax * sizeof(float));
  q.dxb = take(B * wmax * sizeof(float));
  q.dhc = take(L * B * H * sizeof(float));
  q.dcc = take(L * B * H * sizeof(float));
  q.dgx = take(L * n * q.GH * es);
  q.dgh = take(r.cell == ST_CELL_GRU ? L * n * q.GH * es : 0);
  q.dez = take(n * E * es);
  q.datt2p = take(n * p->A * es);
  q.datt2 = take(B * p->A * sizeof(float));
  q.dz = take(B * p->F * sizeof(float));
  q.datt1 = take((size_t)q.BP * p->A * sizeof(float));
  q.dalpha = take((size_t)q.BP * sizeof(float));
  q.hprev = take(n * H * es);
  const size_t ra = mx(mx((size_t)r.V, (size_t)q.GH), mx((size_t)p->A, mx(E, H)));
  const size_t rb = mx(mx((size_t)p->F, 2 * E), H);
  q.tA = take(ra * q.Np * es);
  q.tB = take(rb * q.Np * es);
  q.wThh = take(L * H * q.GH * es);
  q.wTih = take(L * wmax * q.GH * es);
  q.wTmisc = take(mx(mx((size_t)H * q.Vp, (size_t)p->F * E), (size_t)H * p->A) * es);
  q.cast = take(mx(mx((size_t)q.BP * p->A, B * H), (size_t)H * p->A) * es);
  q.total = o;
  return q;
}

int gemm_nt(const void* a, int lda, const void* w, int ldw, void* y, int ldy, int M, int N, int K, int dtype, int out_dtype,
            const float* bias, int accumulate, void* stream) {
  if (M <= 0 || N <= 0) return 0;
  st_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.x = a; d.w = w; d.y = y; d.bias = bias; d.dtype = dtype; d.out_dtype = out_dtype;
  d.B = M; d.Hin = 1; d.Win = 1; d.Cin = K; d.Ho = 1; d.Wo = 1; d.N = N; d.KH = 1; d.KW = 1; d.stride = 1; d.pad = 0;
  d.ldx = lda; d.ldw = ldw; d.ldy = ldy; d.accumulate = accumulate;
  return st_conv(&d, stream);
}

// out_f32[M][ldo] (+)= A[M][K] W[N][K]^T (+bias): skinny MFMA product for the <= batch rows of one timestep
int skinny(const void* A, int lda, const void* W, int ldw, float* out, int ldo, int M, int N, int K, const float* bias, int accumulate,
           int dtype, hipStream_t st) {
  RnnGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.W = W; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.gstride = 0;
  a.out_f32 = out; a.ldo = ldo; a.accumulate = accumulate; a.bias_h = bias;
  return rnn_gemm_launch(a, dtype, 0, 0, st);
}

// y[M][ldy] (storage dtype) = A W^T + bias through the skinny kernel: for the per-step context embedding ([B_t x F] x [F x E],
// K = 2048): 4 output tiles of a 128x128 MFMA tile walk K for ~37 us, 128 blocks of 16x16 with K split over 4 waves for ~10
int skinny_t(const void* A, int lda, const void* W, int ldw, void* y, int ldy, int M, int N, int K, const float* bias, int dtype, hipStream_t st) {
  RnnGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.W = W; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.gstride = 0;
  a.hout = y; a.ldho = ldy; a.bias_h = bias;
  return rnn_gemm_launch(a, dtype, 0, 0, st);
}

int check_common(const st_attn_params* p, const st_packed_seq* s, const char* who) {
  ST_CHECK(p, "%s: null descriptor", who);
  const st_rnn_params& r = p->rnn;
  ST_CHECK(r.cell == ST_CELL_GRU || r.cell == ST_CELL_LSTM, "%s: bad cell", who);
  ST_CHECK(r.L >= 1 && r.L <= ST_MAX_LAYERS, "%s: bad layer count %d", who, r.L);
  ST_CHECK(r.in0 == 2 * r.E, "%s: the attention decoder's layer-0 input is 2*embed_dim (rnn_attn.py:50)", who);
  ST_CHECK(r.H % 8 == 0 && r.E % 8 == 0 && p->A % 8 == 0 && p->F % 8 == 0, "%s: need H, E, A and F multiples of 8 (H=%d E=%d A=%d F=%d)",
           who, r.H, r.E, p->A, p->F);
  ST_CHECK(p->P >= 1 && p->P <= 64, "%s: at most 64 pixels (P=%d)", who, p->P);
  ST_CHECK(p->w_enc && p->b_enc && p->w_dec && p->b_dec && p->w_full && p->b_full && p->w_init_h && p->b_init_h && p->w_embed && p->b_embed && r.emb,
           "%s: null attention weights", who);
  ST_CHECK(r.cell == ST_CELL_GRU || (p->w_init_c && p->b_init_c), "%s: LSTM needs init_c", who);
  if (s) {
    ST_CHECK(s->B > 0 && s->T > 0 && s->ntok > 0 && s->batch_sizes_host && s->rows_b && s->rows_t && s->prev_row, "%s: bad packed-sequence descriptor", who);
    int sum = 0, prev = s->B;
    for (int t = 0; t < s->T; ++t) {
      const int b = s->batch_sizes_host[t];
      ST_CHECK(b > 0 && b <= prev, "%s: batch_sizes must be positive and non-increasing", who);
      prev = b; sum += b;
    }
    ST_CHECK(sum == s->ntok && s->batch_sizes_host[0] == s->B, "%s: batch_sizes do not add up", who);
    ST_CHECK(s->Tcap >= s->T, "%s: caption width %d smaller than the longest length %d", who, s->Tcap, s->T);
  }
  return 0;
}

// feat_pf, mean, h0 (c0), att1 -- shared by training and greedy
int prepare(const st_attn_params* p, const float* cnn_feature, int B, char* feat, char* mean, char* h0, char* c0, char* att1, void* stream) {
  const st_rnn_params& r = p->rnn;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int dt = r.dtype, BP = B * p->P;
  if (ncp_to_pf_launch(cnn_feature, feat, B, p->F, p->P, dt, st)) return 1;
  if (st_global_avgpool(feat, mean, dt, dt, B, p->P, p->F, stream)) return 1;                        // cnn_feature.mean(dim=2), rnn_attn.py:62
  if (gemm_nt(mean, p->F, p->w_init_h, p->F, h0, r.H, B, r.H, p->F, dt, dt, p->b_init_h, 0, stream)) return 1;
  if (r.cell == ST_CELL_LSTM && gemm_nt(mean, p->F, p->w_init_c, p->F, c0, r.H, B, r.H, p->F, dt, dt, p->b_init_c, 0, stream)) return 1;
  return gemm_nt(feat, p->F, p->w_enc, p->F, att1, p->A, BP, p->A, p->F, dt, dt, p->b_enc, 0, stream);   // hoisted encoder_att
}

}  // namespace

extern "C" size_t st_attn_workspace_bytes(const st_attn_params* p, const st_packed_seq* s) {
  if (!p || !s) return 0;
  return make_plan(p, s).total;
}

extern "C" int st_attn_forward(const st_attn_params* p, const st_packed_seq* s, const float* cnn_feature, const long* caption_T,
                               void* workspace, size_t workspace_bytes, void* logits, int logits_dtype, int ldl,
                               float* alphas, int save_for_backward, void* stream) {
  if (check_common(p, s, "st_attn_forward")) return 1;
  ST_CHECK(cnn_feature && caption_T && workspace && alphas, "st_attn_forward: null pointer");
  const Plan q = make_plan(p, s);
  ST_CHECK(workspace_bytes >= q.total, "st_attn_forward: workspace too small (%zu < %zu)", workspace_bytes, q.total);
  const st_rnn_params& r = p->rnn;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  char* ws = reinterpret_cast<char*>(workspace);
  const int dt = r.dtype, H = r.H, E = r.E, n = s->ntok, B = s->B, L = r.L, P = p->P, A = p->A, F = p->F;
  const size_t es = q.es;
  (void)save_for_backward;   // every saved tensor is also the forward's own scratch
  std::vector<int> off(s->T + 1, 0);
  for (int t = 0; t < s->T; ++t) off[t + 1] = off[t] + s->batch_sizes_host[t];

  if (prepare(p, cnn_feature, B, ws + q.feat, ws + q.mean, ws + q.h0, ws + q.c0, ws + q.att1, stream)) return 1;

  for (int t = 0; t < s->T; ++t) {
    const int bt = s->batch_sizes_host[t];
    const char* htop_prev = t > 0 ? ws + q.y + ((size_t)(L - 1) * n + off[t - 1]) * H * es : ws + q.h0;
    float* att2 = reinterpret_cast<float*>(ws + q.att2) + (size_t)off[t] * A;
    char* zt = ws + q.z + (size_t)off[t] * F * es;
    char* xt = ws + q.xp + (size_t)off[t] * 2 * E * es;
    if (skinny(htop_prev, H, p->w_dec, H, att2, A, bt, A, H, p->b_dec, 0, dt, st)) return 1;
    if (attn_fwd_launch(ws + q.att1, att2, p->w_full, p->b_full, ws + q.feat, alphas + (size_t)t * P, (long)s->Tcap * P, zt, bt, P, A, F, dt, st)) return 1;
    if (st_embedding_rows(r.emb, caption_T + (size_t)t * B, xt, bt, E, r.V, 2 * E, dt, stream)) return 1;
    if (skinny_t(zt, F, p->w_embed, F, xt + (size_t)E * es, 2 * E, bt, E, F, p->b_embed, dt, st)) return 1;
    for (int l = 0; l < L; ++l) {
      char* yl = ws + q.y + (size_t)l * n * H * es;
      RnnGemmArgs a;
      memset(&a, 0, sizeof(a));
      a.M = bt; a.N = H; a.gstride = H;
      a.A2 = l == 0 ? xt : ws + q.y + ((size_t)(l - 1) * n + off[t]) * H * es;
      a.W2 = r.w_ih[l]; a.K2 = l == 0 ? 2 * E : H; a.lda2 = a.K2; a.ldw2 = a.K2;
      a.A = t > 0 ? yl + (size_t)off[t - 1] * H * es : ws + q.h0;      // every layer starts from the same h0 (rnn_attn.py:62)
      a.W = r.w_hh[l]; a.K = H; a.lda = H; a.ldw = H;
      a.hprev = a.A; a.ldhp = H;
      a.bias_h = r.b_hh[l]; a.bias_x = r.b_ih[l];
      a.hout = yl + (size_t)off[t] * H * es; a.ldho = H;
      a.cache = ws + q.gates + ((size_t)l * n + off[t]) * 4 * H * es; a.ldcache = 4 * H;
      if (r.cell == ST_CELL_LSTM) {
        char* cl = ws + q.cst + (size_t)l * n * H * es;
        a.cprev = t > 0 ? cl + (size_t)off[t - 1] * H * es : ws + q.c0;
        a.cout = cl + (size_t)off[t] * H * es;
      }
      if (rnn_gemm_launch(a, dt, r.cell == ST_CELL_GRU ? 1 : 2, 1, st)) return 1;
    }
  }
  if (logits) {
    ST_CHECK(r.w_lin && r.b_lin, "st_attn_forward: logits requested without the vocabulary projection");
    if (gemm_nt(ws + q.y + (size_t)(L - 1) * n * H * es, H, r.w_lin, H, logits, ldl, n, r.V, H, dt, logits_dtype, r.b_lin, 0, stream)) return 1;
  }
  return 0;
}

extern "C" int st_attn_reg_loss(const float* alphas, int B, int T, int P, float alpha_c, float* loss_accum, void* stream) {
  ST_CHECK(alphas && loss_accum, "st_attn_reg_loss: null pointer");
  return attn_reg_launch(alphas, B, T, P, alpha_c, loss_accum, nullptr, nullptr, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int st_attn_backward(const st_attn_params* p, const st_attn_grads* g, const st_packed_seq* s, const long* caption_T,
                                const void* dlogits, int ldd, const float* alphas, const float* dalphas, float alpha_c,
                                const float* grad_scale_dev, void* workspace, size_t workspace_bytes, void* stream) {
  if (check_common(p, s, "st_attn_backward")) return 1;
  ST_CHECK(g && caption_T && dlogits && alphas && workspace, "st_attn_backward: null pointer");
  const Plan q = make_plan(p, s);
  ST_CHECK(workspace_bytes >= q.total, "st_attn_backward: workspace too small (%zu < %zu)", workspace_bytes, q.total);
  const st_rnn_params& r = p->rnn;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  char* ws = reinterpret_cast<char*>(workspace);
  const int dt = r.dtype, H = r.H, E = r.E, n = s->ntok, B = s->B, L = r.L, P = p->P, A = p->A, F = p->F, GH = q.GH, Np = q.Np, BP = q.BP;
  const size_t es = q.es;
  ST_CHECK(ldd >= q.Vp && ldd % 8 == 0, "st_attn_backward: dlogits leading dimension must be a multiple of 8 and >= %d", q.Vp);
  ST_CHECK(Np >= n && Np >= BP, "st_attn_backward: plan Np=%d n=%d BP=%d P=%d B=%d T=%d", Np, n, BP, P, B, s->T);
  std::vector<int> off(s->T + 1, 0);
  for (int t = 0; t < s->T; ++t) off[t + 1] = off[t] + s->batch_sizes_host[t];
  const char* ytop = ws + q.y + (size_t)(L - 1) * n * H * es;
  float* dytop = reinterpret_cast<float*>(ws + q.dytop);
  float* dhc = reinterpret_cast<float*>(ws + q.dhc);
  float* dcc = reinterpret_cast<float*>(ws + q.dcc);
  float* datt1 = reinterpret_cast<float*>(ws + q.datt1);
  float* dalpha = reinterpret_cast<float*>(ws + q.dalpha);
  float* dz = reinterpret_cast<float*>(ws + q.dz);
  float* datt2 = reinterpret_cast<float*>(ws + q.datt2);

  // vocabulary projection
  if (st_transpose_colsum(dlogits, ws + q.tA, g->rnn.b_lin, dt, n, r.V, ldd, Np, stream)) return 1;
  if (st_transpose(ytop, ws + q.tB, dt, n, H, H, Np, stream)) return 1;
  if (gemm_nt(ws + q.tA, Np, ws + q.tB, Np, g->rnn.w_lin, H, r.V, H, Np, dt, ST_F32, nullptr, 1, stream)) return 1;
  if (st_transpose(r.w_lin, ws + q.wTmisc, dt, r.V, H, H, q.Vp, stream)) return 1;
  if (gemm_nt(dlogits, ldd, ws + q.wTmisc, q.Vp, dytop, H, n, H, q.Vp, dt, ST_F32, nullptr, 0, stream)) return 1;

  // transposed operands used inside the time loop
  for (int l = 0; l < L; ++l) {
    const int in = l == 0 ? 2 * E : H;
    if (st_transpose(r.w_hh[l], ws + q.wThh + (size_t)l * H * GH * es, dt, GH, H, H, GH, stream)) return 1;
    if (st_transpose(r.w_ih[l], ws + q.wTih + (size_t)l * mx(H, 2 * E) * GH * es, dt, GH, in, in, GH, stream)) return 1;
  }
  char* wTembed = ws + q.wTmisc;                                   // [F][E]   (w_lin^T is no longer needed)
  if (st_transpose(p->w_embed, wTembed, dt, E, F, F, E, stream)) return 1;
  char* wTdec = ws + q.cast;                                       // [H][A]
  if (st_transpose(p->w_dec, wTdec, dt, A, H, H, A, stream)) return 1;

  if (hipMemsetAsync(dhc, 0, (size_t)L * B * H * sizeof(float), st) != hipSuccess ||
      hipMemsetAsync(dcc, 0, (size_t)L * B * H * sizeof(float), st) != hipSuccess ||
      hipMemsetAsync(datt1, 0, (size_t)BP * A * sizeof(float), st) != hipSuccess) { st_set_error("memset failed"); return 1; }
  if (!dalphas && attn_reg_launch(alphas, B, s->Tcap, P, alpha_c, nullptr, dalpha, grad_scale_dev, st)) return 1;

  for (int t = s->T - 1; t >= 0; --t) {
    const int bt = s->batch_sizes_host[t];
    const float* dy = dytop + (size_t)off[t] * H;
    float* dxa = reinterpret_cast<float*>(ws + q.dxa);
    float* dxb = reinterpret_cast<float*>(ws + q.dxb);
    for (int l = L - 1; l >= 0; --l) {
      const int in = l == 0 ? 2 * E : H;
      char* yl = ws + q.y + (size_t)l * n * H * es;
      const void* hprev = t > 0 ? yl + (size_t)off[t - 1] * H * es : ws + q.h0;
      char* dgx = ws + q.dgx + ((size_t)l * n + off[t]) * GH * es;
      char* dgh = r.cell == ST_CELL_GRU ? ws + q.dgh + ((size_t)l * n + off[t]) * GH * es : dgx;
      const char* cache = ws + q.gates + ((size_t)l * n + off[t]) * 4 * H * es;
      float* dhl = dhc + (size_t)l * B * H;
      if (r.cell == ST_CELL_GRU) {
        if (gru_bwd_gates_launch(dy, dhl, cache, hprev, dgx, dgh, bt, H, dt, st)) return 1;
      } else {
        char* cl = ws + q.cst + (size_t)l * n * H * es;
        const void* cprev = t > 0 ? cl + (size_t)off[t - 1] * H * es : ws + q.c0;
        if (lstm_bwd_gates_launch(dy, dhl, dcc + (size_t)l * B * H, cache, cl + (size_t)off[t] * H * es, cprev, dgx, bt, H, dt, st)) return 1;
      }
      // dh_{t-1}[l] += dgh W_hh   (also at t == 0: the gradient reaches h0 -> init_h)
      if (skinny(dgh, GH, ws + q.wThh + (size_t)l * H * GH * es, GH, dhl, H, bt, H, GH, nullptr, 1, dt, st)) return 1;
      // dx_l = dgx W_ih
      if (skinny(dgx, GH, ws + q.wTih + (size_t)l * mx(H, 2 * E) * GH * es, GH, dxa, in, bt, in, GH, nullptr, 0, dt, st)) return 1;
      dy = dxa;
      float* tmp = dxa; dxa = dxb; dxb = tmp;
    }
    // dy now holds dx0 [bt][2E]
    char* dez = ws + q.dez + (size_t)off[t] * E * es;
    if (split_dx0_launch(dy, caption_T + (size_t)t * B, g->rnn.emb, dez, bt, E, r.V, dt, st)) return 1;
    if (skinny(dez, E, wTembed, E, dz, F, bt, F, E, nullptr, 0, dt, st)) return 1;                  // dz = d ez W_embed
    const float* att2 = reinterpret_cast<const float*>(ws + q.att2) + (size_t)off[t] * A;
    if (attn_bwd_launch(dz, dalphas ? dalphas + (size_t)t * P : dalpha, dalphas ? (long)s->Tcap * P : (long)P, alphas + (size_t)t * P, (long)s->Tcap * P,
                        ws + q.att1, att2, p->w_full, ws + q.feat,
                        datt2, datt1, g->w_full, g->b_full, bt, P, A, F, dt, st)) return 1;
    char* datt2p = ws + q.datt2p + (size_t)off[t] * A * es;
    if (st_cast(datt2, datt2p, ST_F32, dt, (long)bt * A, stream)) return 1;
    // the attention of step t was keyed on the top layer's PREVIOUS state
    if (skinny(datt2p, A, wTdec, A, dhc + (size_t)(L - 1) * B * H, H, bt, H, A, nullptr, 1, dt, st)) return 1;
  }

  // ---- parameter gradients as large GEMMs over the packed rows -------------------------------------------
  for (int l = 0; l < L; ++l) {
    const int in = l == 0 ? 2 * E : H;
    const char* xl = l == 0 ? ws + q.xp : ws + q.y + (size_t)(l - 1) * n * H * es;
    char* dgx = ws + q.dgx + (size_t)l * n * GH * es;
    char* dgh = r.cell == ST_CELL_GRU ? ws + q.dgh + (size_t)l * n * GH * es : dgx;
    if (r.cell != ST_CELL_GRU && colsum_launch(dgh, g->rnn.b_hh[l], n, GH, GH, dt, st)) return 1;   // LSTM: dgh == dgx
    if (st_transpose_colsum(dgx, ws + q.tA, g->rnn.b_ih[l], dt, n, GH, GH, Np, stream)) return 1;
    if (st_transpose(xl, ws + q.tB, dt, n, in, in, Np, stream)) return 1;
    if (gemm_nt(ws + q.tA, Np, ws + q.tB, Np, g->rnn.w_ih[l], in, GH, in, Np, dt, ST_F32, nullptr, 1, stream)) return 1;
    if (gather_hprev_launch(ws + q.y + (size_t)l * n * H * es, s->rows_t, s->prev_row, ws + q.hprev, n, H, dt, st, ws + q.h0, s->rows_b)) return 1;
    if (r.cell == ST_CELL_GRU && st_transpose_colsum(dgh, ws + q.tA, g->rnn.b_hh[l], dt, n, GH, GH, Np, stream)) return 1;
    if (st_transpose(ws + q.hprev, ws + q.tB, dt, n, H, H, Np, stream)) return 1;
    if (gemm_nt(ws + q.tA, Np, ws + q.tB, Np, g->rnn.w_hh[l], H, GH, H, Np, dt, ST_F32, nullptr, 1, stream)) return 1;
  }
  // decoder_att: d att2 rows x previous top state (hprev of the last layer is still in q.hprev / tB)
  if (st_transpose_colsum(ws + q.datt2p, ws + q.tA, g->b_dec, dt, n, A, A, Np, stream)) return 1;
  if (gemm_nt(ws + q.tA, Np, ws + q.tB, Np, g->w_dec, H, A, H, Np, dt, ST_F32, nullptr, 1, stream)) return 1;
  // embed: d ez rows x z rows
  if (st_transpose_colsum(ws + q.dez, ws + q.tA, g->b_embed, dt, n, E, E, Np, stream)) return 1;
  if (st_transpose(ws + q.z, ws + q.tB, dt, n, F, F, Np, stream)) return 1;
  if (gemm_nt(ws + q.tA, Np, ws + q.tB, Np, g->w_embed, F, E, F, Np, dt, ST_F32, nullptr, 1, stream)) return 1;
  // encoder_att: time-summed d att1 (B*P rows) x feat
  if (st_cast(datt1, ws + q.cast, ST_F32, dt, (long)BP * A, stream)) return 1;
  if (st_transpose_colsum(ws + q.cast, ws + q.tA, g->b_enc, dt, BP, A, A, Np, stream)) return 1;
  if (st_transpose(ws + q.feat, ws + q.tB, dt, BP, F, F, Np, stream)) return 1;
  if (gemm_nt(ws + q.tA, Np, ws + q.tB, Np, g->w_enc, F, A, F, Np, dt, ST_F32, nullptr, 1, stream)) return 1;
  // init_h (init_c): every layer started from the same h0, so its gradient is the sum over layers
  const int Bp = up8(B);
  for (int pass = 0; pass < (r.cell == ST_CELL_LSTM ? 2 : 1); ++pass) {
    float* acc = pass == 0 ? dhc : dcc;
    for (int l = 1; l < L; ++l) if (add_rows_launch(acc, acc + (size_t)l * B * H, (long)B * H, st)) return 1;
    if (st_cast(acc, ws + q.cast, ST_F32, dt, (long)B * H, stream)) return 1;
    float* gw = pass == 0 ? g->w_init_h : g->w_init_c;
    float* gb = pass == 0 ? g->b_init_h : g->b_init_c;
    if (st_transpose_colsum(ws + q.cast, ws + q.tA, gb, dt, B, H, H, Bp, stream)) return 1;
    if (st_transpose(ws + q.mean, ws + q.tB, dt, B, F, F, Bp, stream)) return 1;
    if (gemm_nt(ws + q.tA, Bp, ws + q.tB, Bp, gw, F, H, F, Bp, dt, ST_F32, nullptr, 1, stream)) return 1;
  }
  return 0;
}

// ---- greedy decoding (rnn_attn.py:77-94,120-145) -----------------------------------------------------------
namespace {
struct GPlan { size_t feat, mean, h0, c0, att1, h[2], c[2], x, z, att2, logits, ids, total; };
GPlan make_gplan(const st_attn_params* p, int B) {
  const st_rnn_params& r = p->rnn;
  const size_t es = st_dtype_size(r.dtype);
  GPlan q; size_t o = 0;
  auto take = [&](size_t bytes) { size_t x = o; o += al(bytes); return x; };
  q.feat = take((size_t)B * p->P * p->F * es); q.mean = take((size_t)B * p->F * es);
  q.h0 = take((size_t)B * r.H * es); q.c0 = take((size_t)B * r.H * es);
  q.att1 = take((size_t)B * p->P * p->A * es);
  for (int i = 0; i < 2; ++i) { q.h[i] = take((size_t)r.L * B * r.H * es); q.c[i] = take((size_t)r.L * B * r.H * es); }
  q.x = take((size_t)B * 2 * r.E * es); q.z = take((size_t)B * p->F * es);
  q.att2 = take((size_t)B * p->A * sizeof(float));
  q.logits = take((size_t)B * up8(r.V) * sizeof(float));
  q.ids = take((size_t)B * sizeof(long) + (size_t)B * p->P * sizeof(float));
  q.total = o;
  return q;
}

// arg-max over the vocabulary (first maximum, like torch.max) -> ids column t and the running token buffer
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ logits, int ldl, int V, long* __restrict__ ids,
                                                     int ids_stride, int t, long* __restrict__ cur) {
  __shared__ float sv[4]; __shared__ int si[4];
  const int row = blockIdx.x;
  const float* l = logits + (long)row * ldl;
  float best = -INFINITY; int bi = 0x7fffffff;
  for (int i = threadIdx.x; i < V; i += blockDim.x) { const float v = l[i]; if (v > best || (v == best && i < bi)) { best = v; bi = i; } }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    best = sv[0]; bi = si[0];
    for (int w = 1; w < 4; ++w) if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
    if (bi >= V) bi = 0;
    ids[(long)row * ids_stride + t] = bi;
    cur[row] = bi;
  }
}
__global__ void fill_ids_kernel(long* cur, int n, long v) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) cur[i] = v; }
}  // namespace

extern "C" size_t st_attn_greedy_workspace_bytes(const st_attn_params* p, int B) {
  if (!p || B <= 0) return 0;
  return make_gplan(p, B).total;
}

extern "C" int st_attn_greedy(const st_attn_params* p, const float* cnn_feature, int B, int steps, long start_id,
                              void* workspace, size_t workspace_bytes, long* ids_out, void* stream) {
  if (check_common(p, nullptr, "st_attn_greedy")) return 1;
  ST_CHECK(cnn_feature && workspace && ids_out && B > 0 && steps > 0, "st_attn_greedy: bad arguments");
  const st_rnn_params& r = p->rnn;
  ST_CHECK(r.w_lin && r.b_lin, "st_attn_greedy: null vocabulary projection");
  const GPlan q = make_gplan(p, B);
  ST_CHECK(workspace_bytes >= q.total, "st_attn_greedy: workspace too small");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  char* ws = reinterpret_cast<char*>(workspace);
  const int dt = r.dtype, H = r.H, E = r.E, L = r.L, P = p->P, A = p->A, F = p->F, Vp = up8(r.V);
  const size_t es = st_dtype_size(dt);
  if (prepare(p, cnn_feature, B, ws + q.feat, ws + q.mean, ws + q.h0, ws + q.c0, ws + q.att1, stream)) return 1;
  if (replicate_rows_launch(ws + q.h0, ws + q.h[0], (long)B * H, L, dt, st)) return 1;            // h0 repeated over layers (rnn_attn.py:62)
  if (r.cell == ST_CELL_LSTM && replicate_rows_launch(ws + q.c0, ws + q.c[0], (long)B * H, L, dt, st)) return 1;
  long* cur = reinterpret_cast<long*>(ws + q.ids);
  float* alpha_scratch = reinterpret_cast<float*>(ws + q.ids + (size_t)B * sizeof(long));
  hipLaunchKernelGGL(fill_ids_kernel, dim3((B + 255) / 256), dim3(256), 0, st, cur, B, start_id);
  ST_LAUNCH_CHECK();
  float* att2 = reinterpret_cast<float*>(ws + q.att2);
  float* logits = reinterpret_cast<float*>(ws + q.logits);
  int c = 0;
  for (int t = 0; t < steps; ++t) {
    const int nx = c ^ 1;
    const char* htop = ws + q.h[c] + (size_t)(L - 1) * B * H * es;
    if (skinny(htop, H, p->w_dec, H, att2, A, B, A, H, p->b_dec, 0, dt, st)) return 1;
    if (attn_fwd_launch(ws + q.att1, att2, p->w_full, p->b_full, ws + q.feat, alpha_scratch, P, ws + q.z, B, P, A, F, dt, st)) return 1;
    if (st_embedding_rows(r.emb, cur, ws + q.x, B, E, r.V, 2 * E, dt, stream)) return 1;
    if (skinny_t(ws + q.z, F, p->w_embed, F, ws + q.x + (size_t)E * es, 2 * E, B, E, F, p->b_embed, dt, st)) return 1;
    if (st_rnn_step(&r, ws + q.x, B, ws + q.h[c], ws + q.c[c], ws + q.h[nx], ws + q.c[nx], logits, Vp, stream)) return 1;
    hipLaunchKernelGGL(argmax_kernel, dim3(B), dim3(256), 0, st, logits, Vp, r.V, ids_out, steps, t, cur);
    ST_LAUNCH_CHECK();
    c = nx;
  }
  return 0;
}
