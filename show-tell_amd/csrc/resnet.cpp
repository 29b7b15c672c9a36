// ResNet-{18,34,50,101,152} encoder forward as one C-ABI call (gfx950).
//
// Replaces `self.model(x)` of the reference encoder (cnn.py:46 / cnn_attn.py:46), i.e.
// torchvision's resnet children()[:-1] / [:-2], and the reshape that follows it
// (cnn.py:48 flatten of the pooled map, cnn_attn.py:49 view(B,2048,49)).
//
// The whole layer sequence is issued from C++ on one stream (no Python per layer, nothing
// allocated or synchronised -> capturable in a hipGraph).  Activations are NHWC in `dtype`.
//   train != 0 : every BatchNorm2d uses batch statistics (main.py:125 calls cnn.train()):
//                conv epilogue accumulates per-channel sum / sum-of-squares, a fused
//                elementwise pass normalises (+residual)(+ReLU) in place, one kernel updates
//                all running buffers at the end.
//   train == 0 : BN is folded to per-channel scale/shift once per call and applied in the
//                conv epilogue together with the residual add and the ReLU.
#include "common.h"
#include <vector>
#include <map>
#include <mutex>
#include <new>
#include <string.h>
#include <stdlib.h>

struct ConvL { int cin, cout, k, stride, pad; size_t woff; size_t bnoff; int korder; size_t woff_frag; int ntw; };   // woff_frag: fragment-major copy for st_conv3x3_img (ntw > 0)
struct BlockL { int c1, c2, c3, ds; int stride; };
// The activation-stationary pointwise kernel (one workgroup per 112 rows) takes the 256 -> 1024 conv3 of layer3, the stride-2
// downsample convs of layer2 / layer3 (256 -> 512, 512 -> 1024: many output-channel slices per input row) and layer4's 512 -> 2048
// (49 rows per image: 56 row blocks at B = 128, so the channels are cut into four parts per row block there -- as one workgroup
// per row block it measured 63 us against 32 on st_conv1x1_wreg).  Static per layer: the fragment-major copy is packed for ONE
// kernel's channel permutation (ntw).
static inline bool use_astat(const ConvL& c) {
  if (c.k != 1) return false;
  if (c.stride == 1) return (c.cin == 256 && c.cout == 1024) || (c.cin == 512 && c.cout == 2048);   // conv3 of layer3 / layer4
  return c.stride == 2 && ((c.cin == 256 && c.cout == 512) || (c.cin == 512 && c.cout == 1024));   // downsample convs of layer2 / layer3
}

struct BnTable { int n; int end[160]; float count[160]; int soff[160]; int rep[160]; };   // soff: float offset of the layer's [rep][2C] statistics
struct PendingUpdate { BnTable tab; const float* stats; };

struct st_resnet {
  int version, dtype, bottleneck, cpad0;
  std::vector<ConvL> convs;
  std::vector<BlockL> blocks;
  size_t wtotal = 0, bntotal = 0;
  int feat_dim = 0;
  // train == 2 (statistics now, running buffers later): the layer table of a forward waits here, keyed by its workspace,
  // for st_resnet_update_running -- lets concurrent forwards on different streams apply their momentum updates in order
  mutable std::map<const void*, PendingUpdate> pending;
  mutable std::mutex mu;
  // test / diagnosis aid (st_resnet_set_taps): when set, every forward copies each residual block's OUTPUT ([B][h][w][C], compute
  // dtype) into this buffer, block after block -- the block-by-block parity test feeds them to the oracle one block at a time
  void* taps = nullptr; size_t taps_bytes = 0;
};

namespace {

constexpr int kStatsRepFloats = 8192;   // cap on rep * 2C per layer (what one bn_act block sums in its preamble)

__global__ void bn_update_all_kernel(const float* __restrict__ stats, float* __restrict__ rm, float* __restrict__ rv,
                                     int total, float mom, BnTable t) {
  // stats holds, per layer l with channels [start_l, end_l): rep_l replicas of [sum(C_l) | sumsq(C_l)] at soff_l
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= total) return;
  int l = 0, hi = t.n - 1;                 // first layer whose channel range ends past c (binary search: each probe is a dependent load)
  while (l < hi) {
    const int mid = (l + hi) >> 1;
    if (c >= t.end[mid]) l = mid + 1; else hi = mid;
  }
  const int start = l == 0 ? 0 : t.end[l - 1];
  const int C = t.end[l] - start;
  const float cnt = t.count[l];
  const float* sl = stats + t.soff[l];
  float s = 0.f, ss = 0.f;
  for (int r = 0; r < t.rep[l]; ++r) { s += sl[r * 2 * C + (c - start)]; ss += sl[r * 2 * C + C + (c - start)]; }
  const float mean = s / cnt;
  const float var = fmaxf(ss / cnt - mean * mean, 0.f);
  const float unb = cnt > 1.f ? var * cnt / (cnt - 1.f) : var;
  rm[c] = (1.f - mom) * rm[c] + mom * mean;
  rv[c] = (1.f - mom) * rv[c] + mom * unb;
}

// replica 0 += replicas 1..R-1 of a layer's [R][2C] statistics: one tiny launch after the conv, so that every block of
// the normalise pass reads 2C floats instead of summing R x 2C itself
__global__ void bn_reduce_replicas_kernel(float* __restrict__ stats, int R, int C2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C2) return;
  // eight replicas per round trip (one dependent load per replica cost the stem's 64-replica reduction 16 us)
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int r = 0;
  for (; r + 8 <= R; r += 8) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = stats[(size_t)(r + q) * C2 + i];
#pragma unroll
    for (int q = 0; q < 8; ++q) s[q] += v[q];
  }
  for (; r < R; ++r) s[0] += stats[(size_t)r * C2 + i];
  stats[i] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

__global__ void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ rm, const float* __restrict__ rv,
                               float* __restrict__ scale, float* __restrict__ shift, int total, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= total) return;
  const float sc = gamma[c] * rsqrtf(rv[c] + eps);
  scale[c] = sc;
  shift[c] = beta[c] - rm[c] * sc;
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
inline int conv_out(int h, int k, int s, int p) { return (h + 2 * p - k) / s + 1; }

}  // namespace

extern "C" int st_resnet_create(int version, int dtype, st_resnet** out) {
  ST_CHECK(out, "st_resnet_create: null out");
  ST_CHECK(dtype == ST_F32 || dtype == ST_BF16, "st_resnet_create: bad dtype %d", dtype);
  int nb[4];
  int bott;
  switch (version) {
    case 18: nb[0] = 2; nb[1] = 2; nb[2] = 2; nb[3] = 2; bott = 0; break;
    case 34: nb[0] = 3; nb[1] = 4; nb[2] = 6; nb[3] = 3; bott = 0; break;
    case 50: nb[0] = 3; nb[1] = 4; nb[2] = 6; nb[3] = 3; bott = 1; break;
    case 101: nb[0] = 3; nb[1] = 4; nb[2] = 23; nb[3] = 3; bott = 1; break;
    case 152: nb[0] = 3; nb[1] = 8; nb[2] = 36; nb[3] = 3; bott = 1; break;
    default:
      // same message as the reference (cnn.py:33)
      st_set_error("Please specify a valid ResNet version. %d doesn't exist.", version);
      return 2;
  }
  st_resnet* r = new (std::nothrow) st_resnet();
  ST_CHECK(r, "st_resnet_create: out of memory");
  r->version = version; r->dtype = dtype; r->bottleneck = bott;
  const int epc = dtype == ST_BF16 ? 8 : 4;
  r->cpad0 = epc;  // 3 input channels zero-padded to one 16-byte chunk
  auto add = [&](int cin, int cout, int k, int s, int p) {
    const int ch = dtype == ST_BF16 ? 64 : 32;
    ConvL c{cin, cout, k, s, p, r->wtotal, r->bntotal, (k > 1 && cin % ch == 0) ? 1 : 0, 0, 0};
    const int cin_p = (cin == 3) ? r->cpad0 : cin;
    r->wtotal += (size_t)cout * k * k * cin_p;
    // 3x3 stride-1 layers get a second, fragment-major copy of their filters for the image-resident kernel (bf16 only);
    // which kernel runs is decided per call from the map size (224 x 224 is nominal: 8 x 8 asks "is there a kernel at all")
    if (dtype == ST_BF16 && k == 3 && s == 1 && p == 1) {
      c.ntw = st_conv3x3_img_supported(8, 8, cin, cout);
      if (c.ntw > 0) { c.woff_frag = r->wtotal; r->wtotal += (size_t)cout * 9 * cin; }
    }
    // the stride-2 3x3 convs of the bottleneck nets (conv2 of the first block of layer2 / 3 / 4): K-streaming kernel, fragment-major copy
    if (dtype == ST_BF16 && k == 3 && s == 2 && p == 1) {
      c.ntw = st_conv3x3_s2_supported(cin, cout);
      if (c.ntw > 0) { c.woff_frag = r->wtotal; r->wtotal += (size_t)cout * 9 * cin; }
    }
    // pointwise layers with <= 512 input channels: fragment-major copy for the register-resident-filter kernel (st_conv1x1_wreg);
    // a stride-2 512-channel layer the activation-stationary kernel does not take stays with st_conv (measured slower on wreg)
    if (dtype == ST_BF16 && k == 1 && p == 0 && (use_astat(c) || !(s == 2 && cin == 512))) {
      c.ntw = use_astat(c) ? st_conv1x1_astat_supported(cin, cout) : 0;    // conv3 of layer3, downsample of layer2 / layer3
      if (c.ntw == 0) c.ntw = st_conv1x1_wreg_supported(cin, cout);
      if (c.ntw == 0) c.ntw = st_conv1x1_kstream_supported(cin, cout);     // 1024 / 2048 input channels: the K-streaming kernel
      if (c.ntw > 0) { c.woff_frag = r->wtotal; r->wtotal += (size_t)cout * cin; }
    }
    r->bntotal += cout;
    r->convs.push_back(c);
    return (int)r->convs.size() - 1;
  };
  add(3, 64, 7, 2, 3);
  int inpl = 64;
  const int planes[4] = {64, 128, 256, 512};
  const int exp = bott ? 4 : 1;
  for (int li = 0; li < 4; ++li)
    for (int bi = 0; bi < nb[li]; ++bi) {
      const int s = bi == 0 ? (li == 0 ? 1 : 2) : 1;
      BlockL b; b.stride = s; b.c3 = -1; b.ds = -1;
      if (bott) {
        b.c1 = add(inpl, planes[li], 1, 1, 0);
        b.c2 = add(planes[li], planes[li], 3, s, 1);
        b.c3 = add(planes[li], planes[li] * 4, 1, 1, 0);
      } else {
        b.c1 = add(inpl, planes[li], 3, s, 1);
        b.c2 = add(planes[li], planes[li], 3, 1, 1);
      }
      if (bi == 0 && (s != 1 || inpl != planes[li] * exp)) b.ds = add(inpl, planes[li] * exp, 1, s, 0);
      inpl = planes[li] * exp;
      r->blocks.push_back(b);
    }
  r->feat_dim = inpl;
  ST_CHECK(r->convs.size() <= 160, "st_resnet_create: too many layers");
  *out = r;
  return 0;
}

extern "C" void st_resnet_destroy(st_resnet* r) { delete r; }
extern "C" int st_resnet_num_convs(const st_resnet* r) { return r ? (int)r->convs.size() : -1; }
extern "C" int st_resnet_feat_dim(const st_resnet* r) { return r ? r->feat_dim : -1; }
extern "C" size_t st_resnet_weight_elems(const st_resnet* r) { return r ? r->wtotal : 0; }
extern "C" size_t st_resnet_bn_channels(const st_resnet* r) { return r ? r->bntotal : 0; }

extern "C" int st_resnet_conv_info(const st_resnet* r, int i, int* cin, int* cout, int* k, int* stride, int* pad,
                                   int* cin_padded, size_t* weight_offset, size_t* bn_offset, int* k_order,
                                   size_t* frag_weight_offset, int* frag_ntw) {
  ST_CHECK(r && i >= 0 && i < (int)r->convs.size(), "st_resnet_conv_info: bad index %d", i);
  const ConvL& c = r->convs[i];
  if (cin) *cin = c.cin;
  if (cout) *cout = c.cout;
  if (k) *k = c.k;
  if (stride) *stride = c.stride;
  if (pad) *pad = c.pad;
  if (cin_padded) *cin_padded = c.cin == 3 ? r->cpad0 : c.cin;
  if (weight_offset) *weight_offset = c.woff;
  if (bn_offset) *bn_offset = c.bnoff;
  if (k_order) *k_order = c.korder;
  if (frag_weight_offset) *frag_weight_offset = c.woff_frag;
  if (frag_ntw) *frag_ntw = c.ntw;
  return 0;
}

extern "C" int st_resnet_set_taps(st_resnet* r, void* buf, size_t bytes) {
  ST_CHECK(r && (buf || bytes == 0), "st_resnet_set_taps: null pointer");
  std::lock_guard<std::mutex> lk(r->mu);
  r->taps = buf; r->taps_bytes = buf ? bytes : 0;
  return 0;
}

namespace {
// ST_LAYER_LOG=<file>: one line per kernel launch of a forward, in launch order -- which kernel family took which layer, its
// geometry, algorithmic FLOPs and HBM bytes (tools/layer_table.py joins it with the rocprofv3 kernel trace of the same process)
FILE* layer_log() {
  static FILE* f = [] { const char* e = getenv("ST_LAYER_LOG"); return e && *e ? fopen(e, "w") : nullptr; }();
  return f;
}
void log_launch(const char* kernel, const char* what, int cin, int cout, int k, int stride, int hin, int win, double flops, double bytes) {
  if (FILE* f = layer_log()) {
    fprintf(f, "%s,%s,%d,%d,%d,%d,%d,%d,%.0f,%.0f\n", kernel, what, cin, cout, k, stride, hin, win, flops, bytes);
    fflush(f);
  }
}
struct Plan {
  size_t in_bytes, s2dw_bytes, stem_bytes, wide_bytes, narrow_bytes, stats_bytes, fold_bytes, total;
};
// Space-to-depth stem (even H, W): the 7x7 stride-2 pad-3 conv over 3 channels is the same sum as a 4x4 stride-1
// conv over the 2x2-blocked image (12 channels, padded to 16; 2 zero rows/cols before, 1 after).  Four neighbouring
// blocked pixels are 128 contiguous bytes, so each filter row is one whole-line tap of the fast loader
// (st_conv sliding-window form) instead of 49 scattered 16-byte gathers per output pixel.
inline bool stem_s2d(int H, int W) { return H % 2 == 0 && W % 2 == 0; }
Plan make_plan(const st_resnet* r, int B, int H, int W) {
  const size_t es = st_dtype_size(r->dtype);
  Plan p;
  p.in_bytes = align256((size_t)B * H * W * r->cpad0 * es);
  if (stem_s2d(H, W)) { const size_t b2 = align256((size_t)B * (H / 2 + 3) * (W / 2 + 3) * 16 * es); if (b2 > p.in_bytes) p.in_bytes = b2; }
  p.s2dw_bytes = align256((size_t)64 * 256 * es);
  const int h1 = conv_out(H, 7, 2, 3), w1 = conv_out(W, 7, 2, 3);
  p.stem_bytes = align256((size_t)B * h1 * w1 * 64 * es);
  const int h2 = conv_out(h1, 3, 2, 1), w2 = conv_out(w1, 3, 2, 1);
  // widest block tensors live at layer1 resolution
  const size_t wide_c = r->bottleneck ? 256 : 64;
  p.wide_bytes = align256((size_t)B * h2 * w2 * wide_c * es);
  p.narrow_bytes = align256((size_t)B * h2 * w2 * 128 * es);  // >= any conv1/conv2 output (layer2 conv1 at layer1 res: 128 ch)
  p.stats_bytes = align256((2 * r->bntotal + (size_t)kStatsRepFloats * r->convs.size()) * sizeof(float));
  p.fold_bytes = align256(2 * r->bntotal * sizeof(float));
  p.total = p.in_bytes + p.s2dw_bytes + p.stem_bytes + 3 * p.wide_bytes + 2 * p.narrow_bytes + p.stats_bytes + p.fold_bytes;
  return p;
}
}  // namespace

extern "C" size_t st_resnet_workspace_bytes(const st_resnet* r, int B, int H, int W) {
  if (!r || B <= 0 || H < 32 || W < 32) return 0;
  return make_plan(r, B, H, W).total;
}

extern "C" int st_resnet_forward(const st_resnet* r, const float* images_nchw, int B, int H, int W,
                                 const void* weights, const float* bn_gamma, const float* bn_beta,
                                 float* bn_running_mean, float* bn_running_var,
                                 int train, float momentum, float eps,
                                 void* workspace, size_t workspace_bytes,
                                 void* feat_nhwc_out, void* pooled_out, int pooled_dtype, float* ncp_out,
                                 void* stream) {
  ST_CHECK(r && images_nchw && weights && bn_gamma && bn_beta && bn_running_mean && bn_running_var && workspace,
           "st_resnet_forward: null pointer");
  ST_CHECK(B > 0 && H >= 32 && W >= 32, "st_resnet_forward: bad input size %dx%dx%d", B, H, W);
  const Plan p = make_plan(r, B, H, W);
  ST_CHECK(workspace_bytes >= p.total, "st_resnet_forward: workspace too small (%zu < %zu)", workspace_bytes, p.total);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int dt = r->dtype;
  const size_t es = st_dtype_size(dt);
  char* ws = reinterpret_cast<char*>(workspace);
  char* in8 = ws; ws += p.in_bytes;
  char* s2dw = ws; ws += p.s2dw_bytes;
  char* stem = ws; ws += p.stem_bytes;
  char* wide[3] = {ws, ws + p.wide_bytes, ws + 2 * p.wide_bytes}; ws += 3 * p.wide_bytes;
  char* narrow[2] = {ws, ws + p.narrow_bytes}; ws += 2 * p.narrow_bytes;
  float* stats = reinterpret_cast<float*>(ws); ws += p.stats_bytes;
  float* fold = reinterpret_cast<float*>(ws);
  const int total = (int)r->bntotal;
  float* fscale = fold; float* fshift = fold + total;

  if (train) {
    if (hipMemsetAsync(stats, 0, p.stats_bytes, st) != hipSuccess) { st_set_error("memset failed"); return 1; }
  } else {
    hipLaunchKernelGGL(bn_fold_kernel, dim3((total + 255) / 256), dim3(256), 0, st, bn_gamma, bn_beta, bn_running_mean,
                       bn_running_var, fscale, fshift, total, eps);
    ST_LAUNCH_CHECK();
  }
  const bool s2d = stem_s2d(H, W);
  if (s2d) {
    if (st_nchw_to_s2d16(images_nchw, in8, dt, B, H, W, stream)) return 1;
  } else if (st_nchw_to_nhwc(images_nchw, in8, dt, B, 3, H, W, r->cpad0, stream)) return 1;

  BnTable tab; tab.n = (int)r->convs.size();
  ST_CHECK(tab.n <= 160, "st_resnet_forward: too many layers");
  for (int i = 0; i < tab.n; ++i) {
    tab.end[i] = (int)(r->convs[i].bnoff + r->convs[i].cout); tab.count[i] = 1.f; tab.soff[i] = 0; tab.rep[i] = 1;
  }
  int stats_used = 0;   // floats handed out so far
  static const bool img_env = [] { const char* e = getenv("ST_CONV_IMG"); return !e || atoi(e) != 0; }();
  const bool use_img = img_env && dt == ST_BF16;   // ST_CONV_IMG=0: every 3x3 through st_conv (A/B switch)
  static const bool s2_env = [] { const char* e = getenv("ST_CONV_S2"); return !e || atoi(e) != 0; }();
  const bool use_s2 = use_img && s2_env;            // ST_CONV_S2=0: the stride-2 3x3 convs stay on st_conv

  // conv: train -> raw output + statistics; eval -> folded BN (+residual)(+ReLU) in the epilogue
  // in_ci >= 0 (train): x is the RAW output of conv in_ci; this conv reads relu(bn_{in_ci}(x)) in its loader
  // keep_rep: the consumer of this layer's statistics sums the replicas itself (everything but st_conv's input transform and
  // the stem's pool): no reduction launch
  // fz (train): x is the RAW conv3 output of the previous block (conv fz->ci); the loader forms relu(bn(x) + identity) itself, writes
  // it to fz->xout (this block's input / identity) and the previous block's separate normalise pass does not run (st_conv1x1_kfuse)
  // b2b: conv fz->ci itself (64 -> 256) is recomputed too, from its RAW input fz->raw2 (output of conv fz->c2ci): st_conv_b2b
  struct FuseIn { int ci; const void* res; int res_ci; void* xout; bool b2b; const void* raw2; int c2ci; };
  auto conv = [&](int ci, const void* x, int hin, int win, void* y, const void* eval_res, int eval_relu,
                  int* ho, int* wo, int in_ci = -1, bool keep_rep = true, const FuseIn* fz = nullptr) -> int {
    const ConvL& c = r->convs[ci];
    st_conv_desc d;
    memset(&d, 0, sizeof(d));
    const int cin_p = c.cin == 3 ? r->cpad0 : c.cin;
    d.x = x; d.w = reinterpret_cast<const char*>(weights) + c.woff * es; d.y = y;
    d.dtype = dt; d.out_dtype = dt;
    d.B = B; d.Hin = hin; d.Win = win; d.Cin = cin_p;
    d.Ho = conv_out(hin, c.k, c.stride, c.pad); d.Wo = conv_out(win, c.k, c.stride, c.pad);
    d.N = c.cout; d.KH = c.k; d.KW = c.k; d.stride = c.stride; d.pad = c.pad;
    d.ldx = cin_p; d.ldw = c.k * c.k * cin_p; d.ldy = c.cout; d.Cin_logical = c.cin; d.k_order = c.korder;
    if (ci == 0 && s2d) {
      d.w = s2dw; d.Hin = hin / 2 + 3; d.Win = win / 2 + 3; d.Cin = 64; d.ldx = 16; d.KH = 4; d.KW = 1; d.stride = 1; d.pad = 0;
      d.ldw = 256; d.Cin_logical = 36;   // 4 rows x 36 = 144 of the 147 real taps: the profiler's FLOP count stays below the algorithmic one
    }
    if (train && in_ci >= 0) {
      const ConvL& pc = r->convs[in_ci];
      d.in_stats = stats + tab.soff[in_ci]; d.in_gamma = bn_gamma + pc.bnoff; d.in_beta = bn_beta + pc.bnoff;
      d.in_count = tab.count[in_ci]; d.in_eps = eps;
    }
    if (train) {
      // replicas: keep ~128-256 pixel tiles per replica (same-address atomics serialise), at most kStatsRepFloats per layer
      const long tiles = ((long)B * d.Ho * d.Wo + 127) / 128;
      int rep = 1;
      while (rep < 64 && tiles / (rep * 2) >= 128 && (rep * 2) * 2 * c.cout <= kStatsRepFloats) rep *= 2;
      tab.soff[ci] = stats_used; tab.rep[ci] = rep;
      stats_used += rep * 2 * c.cout;
      d.stats = stats + tab.soff[ci]; d.stats_replicas = rep;
    } else {
      d.scale = fscale + c.bnoff; d.shift = fshift + c.bnoff; d.residual = eval_res; d.relu = eval_relu;
    }
    *ho = d.Ho; *wo = d.Wo;
    tab.count[ci] = (float)((long)B * d.Ho * d.Wo);
    // (eval mode: conv3 takes the block's identity in its epilogue -- the register-filter and activation-stationary kernels have that
    // form for stride 1 and <= 512 input channels, i.e. every conv3 of a Bottleneck)
    if (c.ntw > 0 && c.k == 1 && use_img && (!d.residual || (c.stride == 1 && c.cin <= 512))) {
      // pointwise, filter slice in registers (conv_img.hip): producer's BatchNorm + ReLU in the row loader, replicated statistics
      st_conv1x1_wreg_desc g;
      memset(&g, 0, sizeof(g));
      g.x = x; g.w_frag = reinterpret_cast<const char*>(weights) + c.woff_frag * es; g.y = y;
      g.B = B; g.Hin = hin; g.Win = win; g.C = c.cin; g.N = c.cout; g.stride = c.stride;
      g.in_stats = d.in_stats; g.in_gamma = d.in_gamma; g.in_beta = d.in_beta; g.in_count = d.in_count; g.in_eps = d.in_eps;
      g.in_stats_replicas = in_ci >= 0 ? tab.rep[in_ci] : 0;
      g.scale = d.scale; g.shift = d.shift; g.relu = d.relu; g.residual = d.residual;
      if (train) {
        // four replicas: the statistics leave a workgroup as full-wave atomics over consecutive channels (block_stats_flush), so the
        // same-address queue is what is left to spread -- and every consumer adds the replicas up in its prologue (cheap at 4)
        int rep = 1;
        while (rep < 4 && (rep * 2) * 2 * c.cout <= kStatsRepFloats) rep *= 2;
        stats_used -= tab.rep[ci] * 2 * c.cout;
        tab.rep[ci] = rep; stats_used += rep * 2 * c.cout;
        g.stats = stats + tab.soff[ci]; g.stats_replicas = rep;
      }
      if (fz && fz->b2b && st_conv_c3c1_supported(r->convs[fz->ci].cin, r->convs[fz->ci].cout, c.cout)) {
        // 14 x 14 and 28 x 28 blocks: conv3 of the previous block (statistics-only pass already run in train mode), its block end and
        // this conv1 in ONE kernel (conv_c3c1.hip) -- conv3's output chunks are this conv1's K-slabs
        const ConvL& pc = r->convs[fz->ci];
        const ConvL& c2 = r->convs[fz->c2ci];
        st_conv_c3c1_desc k;
        memset(&k, 0, sizeof(k));
        k.x2 = fz->raw2; k.w3_frag = reinterpret_cast<const char*>(weights) + pc.woff_frag * es; k.identity = fz->res; k.x_out = fz->xout;
        k.w1_frag = g.w_frag; k.y = y;
        if (train) {
          k.stats = g.stats; k.stats_replicas = g.stats_replicas;
          k.bn2_stats = stats + tab.soff[fz->c2ci]; k.bn2_gamma = bn_gamma + c2.bnoff; k.bn2_beta = bn_beta + c2.bnoff; k.bn2_replicas = tab.rep[fz->c2ci];
          k.bn3_stats = stats + tab.soff[fz->ci]; k.bn3_gamma = bn_gamma + pc.bnoff; k.bn3_beta = bn_beta + pc.bnoff; k.bn3_replicas = tab.rep[fz->ci];
          if (fz->res_ci >= 0) {     // the identity is a raw downsample-conv output: its BatchNorm rides in the kernel's epilogue
            const ConvL& rc = r->convs[fz->res_ci];
            k.id_stats = stats + tab.soff[fz->res_ci]; k.id_gamma = bn_gamma + rc.bnoff; k.id_beta = bn_beta + rc.bnoff; k.id_replicas = tab.rep[fz->res_ci];
          }
          k.count = tab.count[fz->ci]; k.eps = eps;
        } else {
          k.scale3 = fscale + pc.bnoff; k.shift3 = fshift + pc.bnoff; k.scale1 = d.scale; k.shift1 = d.shift; k.relu1 = d.relu;
        }
        k.rows = (long)B * hin * win; k.C1 = pc.cin; k.C2 = pc.cout; k.N = c.cout;
        if (st_conv_c3c1(&k, stream)) return 1;
        log_launch("conv_c3c1", "conv3 + block end (bn3 + identity + relu) + next conv1", pc.cin, c.cout, 1, 1, hin, win,
                   2.0 * k.rows * ((double)pc.cin * pc.cout + (double)c.cin * c.cout),
                   ((double)k.rows * (pc.cin + 2.0 * pc.cout + c.cout) + (double)pc.cin * pc.cout + (double)c.cin * c.cout) * es);
      } else if (fz && fz->b2b) {
        const ConvL& pc = r->convs[fz->ci];
        const ConvL& c2 = r->convs[fz->c2ci];
        st_conv_b2b_desc k;
        memset(&k, 0, sizeof(k));
        k.raw2 = fz->raw2; k.w3_frag = reinterpret_cast<const char*>(weights) + pc.woff_frag * es; k.identity = fz->res; k.x_out = fz->xout;
        k.w1_frag = g.w_frag; k.y = y; k.stats = g.stats; k.stats_replicas = g.stats_replicas;
        k.bn2_stats = stats + tab.soff[fz->c2ci]; k.bn2_gamma = bn_gamma + c2.bnoff; k.bn2_beta = bn_beta + c2.bnoff; k.bn2_replicas = tab.rep[fz->c2ci];
        k.bn3_stats = stats + tab.soff[fz->ci]; k.bn3_gamma = bn_gamma + pc.bnoff; k.bn3_beta = bn_beta + pc.bnoff; k.bn3_replicas = tab.rep[fz->ci];
        if (fz->res_ci >= 0) {
          const ConvL& rc = r->convs[fz->res_ci];
          k.id_stats = stats + tab.soff[fz->res_ci]; k.id_gamma = bn_gamma + rc.bnoff; k.id_beta = bn_beta + rc.bnoff; k.id_replicas = tab.rep[fz->res_ci];
        }
        k.count = tab.count[fz->ci]; k.eps = eps; k.rows = (long)B * hin * win; k.C1 = pc.cin; k.C2 = c.cin; k.N = c.cout;
        if (st_conv_b2b(&k, stream)) return 1;
        log_launch("conv_b2b", "conv3 recomputed + bn3 + identity + relu + next conv1", pc.cin, c.cout, 1, 1, hin, win,
                   2.0 * k.rows * ((double)pc.cin * pc.cout + (double)c.cin * c.cout), (double)k.rows * (pc.cin + 2.0 * pc.cout + c.cout) * es);
      } else if (fz) {
        const ConvL& pc = r->convs[fz->ci];
        st_conv1x1_kfuse_desc k;
        memset(&k, 0, sizeof(k));
        k.raw = x; k.identity = fz->res; k.x_out = fz->xout; k.w_frag = g.w_frag; k.y = y;
        k.stats = g.stats; k.stats_replicas = g.stats_replicas;
        k.f_stats = stats + tab.soff[fz->ci]; k.f_gamma = bn_gamma + pc.bnoff; k.f_beta = bn_beta + pc.bnoff;
        k.f_count = tab.count[fz->ci]; k.f_eps = eps; k.f_stats_replicas = tab.rep[fz->ci];
        k.rows = (long)B * hin * win; k.C = c.cin; k.N = c.cout;
        if (fz->res_ci >= 0) {
          const ConvL& rc = r->convs[fz->res_ci];
          k.id_stats = stats + tab.soff[fz->res_ci]; k.id_gamma = bn_gamma + rc.bnoff; k.id_beta = bn_beta + rc.bnoff;
          k.id_stats_replicas = tab.rep[fz->res_ci];
        }
        if (st_conv1x1_kfuse(&k, stream)) return 1;
        log_launch("conv1x1_wreg", "block end (bn3 + identity + relu) fused into conv1", c.cin, c.cout, 1, 1, hin, win,
                   2.0 * k.rows * (double)c.cin * c.cout, (double)k.rows * (3.0 * c.cin + c.cout) * es);
      } else {
        const double rows_o = (double)B * d.Ho * d.Wo;
        const double fl = 2.0 * rows_o * c.cin * c.cout;
        const double by = (rows_o * c.cin + (double)c.cin * c.cout + (y ? rows_o * c.cout * (g.residual ? 2.0 : 1.0) : 0.0)) * es;
        if (use_astat(c)) {
          if (st_conv1x1_astat(&g, stream)) return 1;
          log_launch("conv1x1_astat", "1x1", c.cin, c.cout, 1, c.stride, hin, win, fl, by);
        } else if (c.cin > 512) {
          ST_CHECK(!g.in_stats, "st_resnet_forward: the long-K pointwise kernel has no input transform");
          if (st_conv1x1_kstream(&g, stream)) return 1;
          log_launch("conv1x1_kstream", "1x1", c.cin, c.cout, 1, c.stride, hin, win, fl, by);
        } else {
          if (st_conv1x1_wreg(&g, stream)) return 1;
          log_launch("conv1x1_wreg", y ? "1x1" : "1x1 statistics only", c.cin, c.cout, 1, c.stride, hin, win, fl, by);
        }
      }
    } else if (c.ntw > 0 && c.k == 3 && c.stride == 1 && use_img && !d.residual && st_conv3x3_img_supported(hin, win, c.cin, c.cout) == c.ntw) {
      // image-resident 3x3 (conv_img.hip): the producer's BatchNorm + ReLU ride in its fill, replicated statistics in and out
      st_conv3x3_img_desc g;
      memset(&g, 0, sizeof(g));
      g.x = x; g.w_frag = reinterpret_cast<const char*>(weights) + c.woff_frag * es; g.y = y;
      g.B = B; g.H = hin; g.W = win; g.C = c.cin; g.N = c.cout;
      g.in_stats = d.in_stats; g.in_gamma = d.in_gamma; g.in_beta = d.in_beta; g.in_count = d.in_count; g.in_eps = d.in_eps;
      g.in_stats_replicas = in_ci >= 0 ? tab.rep[in_ci] : 0;
      g.scale = d.scale; g.shift = d.shift; g.relu = d.relu;
      if (train) {
        int rep = 1;
        const long items = (long)B * ((hin * (win + 2) + 223) / 224);         // ~ workgroups along M
        while (rep < 4 && items / (rep * 2) >= 4 && (rep * 2) * 2 * c.cout <= kStatsRepFloats) rep *= 2;
        stats_used -= tab.rep[ci] * 2 * c.cout;                                // re-plan this layer's replicas
        tab.rep[ci] = rep; stats_used += rep * 2 * c.cout;
        g.stats = stats + tab.soff[ci]; g.stats_replicas = rep;
      }
      if (st_conv3x3_img(&g, stream)) return 1;
      log_launch("conv3x3_img", "3x3", c.cin, c.cout, 3, 1, hin, win, 2.0 * B * hin * win * 9.0 * c.cin * c.cout,
                 ((double)B * hin * win * (c.cin + c.cout) + 9.0 * c.cin * c.cout) * es);
    } else if (c.ntw > 0 && c.k == 3 && c.stride == 2 && use_s2 && !d.residual && st_conv3x3_s2_supported(c.cin, c.cout) == c.ntw &&
               (long)B * hin * win * c.cin * 2 < (1L << 31)) {
      // stride-2 3x3 (conv_s2.hip): K-streaming implicit GEMM with gathered rows; bn1 + ReLU of the producer in its loader
      st_conv3x3_img_desc g;
      memset(&g, 0, sizeof(g));
      g.x = x; g.w_frag = reinterpret_cast<const char*>(weights) + c.woff_frag * es; g.y = y;
      g.B = B; g.H = hin; g.W = win; g.C = c.cin; g.N = c.cout;
      g.in_stats = d.in_stats; g.in_gamma = d.in_gamma; g.in_beta = d.in_beta; g.in_count = d.in_count; g.in_eps = d.in_eps;
      g.in_stats_replicas = in_ci >= 0 ? tab.rep[in_ci] : 0;
      g.scale = d.scale; g.shift = d.shift; g.relu = d.relu;
      if (train) {
        int rep = 1;
        while (rep < 4 && (rep * 2) * 2 * c.cout <= kStatsRepFloats) rep *= 2;
        stats_used -= tab.rep[ci] * 2 * c.cout;
        tab.rep[ci] = rep; stats_used += rep * 2 * c.cout;
        g.stats = stats + tab.soff[ci]; g.stats_replicas = rep;
      }
      if (st_conv3x3_s2(&g, stream)) return 1;
      log_launch("conv3x3_s2", "3x3 stride 2", c.cin, c.cout, 3, 2, hin, win, 2.0 * B * d.Ho * d.Wo * 9.0 * c.cin * c.cout,
                 ((double)B * hin * win * c.cin + (double)B * d.Ho * d.Wo * c.cout + 9.0 * c.cin * c.cout) * es);
    } else {
      if (st_conv(&d, stream)) return 1;
      const double rows_o = (double)B * d.Ho * d.Wo;
      log_launch("igemm", d.residual ? "conv + residual (eval)" : "conv", c.cin, c.cout, c.k, c.stride, hin, win, 2.0 * rows_o * c.k * c.k * c.cin * c.cout,
                 ((double)B * hin * win * cin_p + (double)c.k * c.k * cin_p * c.cout + rows_o * c.cout * (d.residual ? 2.0 : 1.0)) * es);
    }
    // every consumer of a layer's statistics sums up to 16 replicas itself (bn_act's register-coefficient kernel, the
    // conv_img.hip loaders, the running-buffer update); only st_conv's input transform and the stem's pool read replica 0
    if (train && tab.rep[ci] > 1 && !(keep_rep && tab.rep[ci] <= 16)) {
      const int c2 = 2 * c.cout;
      hipLaunchKernelGGL(bn_reduce_replicas_kernel, dim3((c2 + 255) / 256), dim3(256), 0, st, stats + tab.soff[ci], tab.rep[ci], c2);
      ST_LAUNCH_CHECK();
      log_launch("bn_reduce_replicas", "statistics replicas -> replica 0", c.cout, c.cout, 0, 0, 0, 0, 0.0, 0.0);
      tab.rep[ci] = 1;   // consumers (bn_act, running-buffer update) read replica 0
    }
    return 0;
  };
  // train-mode normalise (+residual [+its BN]) (+ReLU), in place
  auto bnact = [&](int ci, void* x, long rows, int relu, const void* res, int res_ci) -> int {
    const ConvL& c = r->convs[ci];
    st_bn_act_desc d;
    memset(&d, 0, sizeof(d));
    d.x = x; d.y = x; d.res = res;
    d.stats = stats + tab.soff[ci]; d.stats_replicas = tab.rep[ci]; d.gamma = bn_gamma + c.bnoff; d.beta = bn_beta + c.bnoff;
    if (res_ci >= 0) {
      const ConvL& rc = r->convs[res_ci];
      d.res_bn = 1; d.res_stats = stats + tab.soff[res_ci]; d.res_stats_replicas = tab.rep[res_ci]; d.res_gamma = bn_gamma + rc.bnoff; d.res_beta = bn_beta + rc.bnoff;
    }
    d.dtype = dt; d.rows = rows; d.C = c.cout; d.count = (float)rows; d.eps = eps; d.relu = relu;
    if (st_bn_act(&d, stream)) return 1;
    log_launch("bn_act", res ? "bn + identity + relu" : "bn + relu", c.cout, c.cout, 0, 0, 0, 0, 0.0, (double)rows * c.cout * (res ? 3.0 : 2.0) * es);
    return 0;
  };

  int h, w;
  // bf16 bottleneck nets on even image sizes: the whole stem (conv1 + statistics + maxpool) is ONE kernel (conv_stem.hip).  In train
  // mode it leaves the pooled RAW conv output in wide[0]; bn1 + relu are applied by the loaders of its two consumers (conv1 and the
  // downsample conv of the first block) -- pooling commutes with the monotone per-channel map, see conv_stem.hip.
  static const bool stem_env = [] { const char* e = getenv("ST_STEM_FUSE"); return !e || atoi(e) != 0; }();
  bool stem_fused = stem_env && s2d && use_img && r->bottleneck && !r->blocks.empty();
  if (stem_fused) {
    const BlockL& b0 = r->blocks[0];
    stem_fused = b0.ds >= 0 && r->convs[b0.c1].k == 1 && r->convs[b0.c1].stride == 1 && r->convs[b0.c1].ntw > 0 && r->convs[b0.ds].ntw > 0 &&
                 r->convs[b0.c1].cin == 64 && !use_astat(r->convs[b0.ds]);
  }
  if (stem_fused) {
    const ConvL& c0 = r->convs[0];
    char* wfrag = stem;                                 // the 205-MB raw-output buffer is free in this form: 32 KB of it hold the filters
    if (st_stem_weight_frag_packed(reinterpret_cast<const char*>(weights) + r->convs[0].woff * es, r->cpad0, wfrag, stream)) return 1;
    st_stem_conv_pool_desc sd;
    memset(&sd, 0, sizeof(sd));
    sd.x_s2d = in8; sd.w_frag = wfrag; sd.y = wide[0]; sd.B = B; sd.H = H; sd.W = W;
    h = conv_out(H, 7, 2, 3); w = conv_out(W, 7, 2, 3);
    tab.count[0] = (float)((long)B * h * w);
    if (train) {
      tab.soff[0] = stats_used; tab.rep[0] = 8; stats_used += 8 * 2 * c0.cout;
      sd.stats = stats + tab.soff[0]; sd.stats_replicas = 8; sd.gamma = bn_gamma + c0.bnoff;
    } else {
      sd.scale = fscale + c0.bnoff; sd.shift = fshift + c0.bnoff;
    }
    if (st_stem_conv_pool(&sd, stream)) return 1;
    log_launch("stem_pool", "7x7/2 conv + statistics + 3x3/2 pool", 3, 64, 7, 2, H, W, 2.0 * B * h * w * 147.0 * 64,
               (double)B * (H / 2 + 3) * (W / 2 + 3) * 16 * es + (double)B * conv_out(h, 3, 2, 1) * conv_out(w, 3, 2, 1) * 64 * es);
  } else {
    if (s2d && st_stem_weight_s2d(reinterpret_cast<const char*>(weights) + r->convs[0].woff * es, s2dw, dt, r->cpad0, stream)) return 1;
    if (conv(0, in8, H, W, stem, nullptr, 1, &h, &w, -1, false)) return 1;
    if (train) {   // bn1 + relu folded into the pool: the 64-channel 112x112 map is read once instead of three times
      const ConvL& c0 = r->convs[0];
      if (st_maxpool3x3s2_bn(stem, wide[0], dt, B, h, w, 64, stats + tab.soff[0], bn_gamma + c0.bnoff, bn_beta + c0.bnoff,
                             nullptr, nullptr, (float)((long)B * h * w), eps, stream)) return 1;
    } else if (st_maxpool3x3s2(stem, wide[0], dt, B, h, w, 64, stream)) return 1;
  }
  h = conv_out(h, 3, 2, 1); w = conv_out(w, 3, 2, 1);
  const int stem_in = (stem_fused && train) ? 0 : -1;   // the first block's conv1 / downsample read relu(bn1(.)) of wide[0] in their loaders
  int cur = 0;  // wide[cur] holds the block input
  size_t taps_used = 0;
  auto tap = [&](const void* src, int hh, int ww, int C) -> int {     // st_resnet_set_taps: block outputs, one after the other
    if (!r->taps) return 0;
    const size_t nb = (size_t)B * hh * ww * C * es;
    ST_CHECK(taps_used + nb <= r->taps_bytes, "st_resnet_forward: taps buffer too small (%zu needed so far, %zu given)", taps_used + nb, r->taps_bytes);
    if (hipMemcpyAsync(reinterpret_cast<char*>(r->taps) + taps_used, src, nb, hipMemcpyDeviceToDevice, st) != hipSuccess) { st_set_error("st_resnet_forward: tap copy failed"); return 1; }
    taps_used += nb;
    return 0;
  };

  // Train, 256-channel block inputs (layer1 and the first block of layer2): the block-end pass relu(bn3(raw) + identity) is formed by
  // the NEXT block's conv1 loader (st_conv1x1_kfuse) -- that pass and conv1 are both HBM time there and the fusion drops one full read
  // of the widest tensor (measured 168 -> 125 us per transition at 56 x 56, B = 128).  Wider inputs stay separate: their conv1 needs
  // several channel slices per row (each would re-read both inputs) or, at 1024 channels, is MFMA-bound (measured slower fused).
  static const bool kfuse_env = [] { const char* e = getenv("ST_BLOCK_FUSE"); return !e || atoi(e) != 0; }();
  static const bool b2b_env = [] { const char* e = getenv("ST_BLOCK_B2B"); return !e || atoi(e) != 0; }();
  struct Pending { bool on; int ci; const void* res; int res_ci; int raw_buf, res_buf; bool b2b; const void* raw2; int c2ci; } pend{false, -1, nullptr, -1, 0, 0, false, nullptr, -1};
  for (size_t bi = 0; bi < r->blocks.size(); ++bi) {
    const BlockL& b = r->blocks[bi];
    FuseIn fzv; const FuseIn* fz = nullptr;
    if (pend.on) {   // wide[cur] holds the previous block's RAW conv3 output; the buffer that is neither it nor the identity takes x
                     // (b2b: conv3's output was never written -- its buffer takes x)
      const int freeb = pend.b2b ? pend.raw_buf : 3 - pend.raw_buf - pend.res_buf;
      fzv = FuseIn{pend.ci, pend.res, pend.res_ci, wide[freeb], pend.b2b, pend.raw2, pend.c2ci};
      fz = &fzv;
      cur = freeb;
    }
    const int oth = (cur + 1) % 3, dsb = (cur + 2) % 3;
    int h1, w1, h2, w2, h3, w3, hd, wd;
    const void* xin = wide[cur];
    const void* c1_in = fz ? static_cast<const void*>(wide[pend.raw_buf]) : xin;
    pend.on = false;
    if (r->bottleneck) {
      // train: bn1 + relu ride in conv2's fill when the image-resident kernel takes conv2 (one pass over the tensor less)
      const ConvL& c2 = r->convs[b.c2];
      const ConvL& c3 = r->convs[b.c3];
      const bool fuse1 = train && use_img && c2.ntw > 0 && c2.stride == 1 && st_conv3x3_img_supported(h, w, c2.cin, c2.cout) == c2.ntw;   // conv1 keeps the map size
      const bool c3_sums = train && use_img && c3.ntw > 0;       // conv3 on st_conv1x1_wreg: sums conv2's replicated statistics itself
      // conv2 on st_conv (the three stride-2 3x3s): its loader takes bn1 + relu too (igemm MODE 2, padding taps stay zero); it reads
      // replica 0 of conv1's statistics, so conv1's replicas are reduced by a launch (5 us against a 14 - 44 us pass over the tensor)
      static const bool xf_env = [] { const char* e = getenv("ST_CONV2_XF"); return !e || atoi(e) != 0; }();
      const bool fuse1x = xf_env && train && use_img && !fuse1 && c2.k == 3 && c2.cin % 64 == 0;
      // (the K-streaming stride-2 kernel sums conv1's statistics replicas itself: no reduction launch in front of it)
      const bool c2_s2k = use_s2 && c2.k == 3 && c2.stride == 2 && c2.ntw > 0 && st_conv3x3_s2_supported(c2.cin, c2.cout) == c2.ntw;
      if (conv(b.c1, c1_in, h, w, narrow[0], nullptr, 1, &h1, &w1, bi == 0 ? stem_in : -1, !fuse1x || c2_s2k, fz)) return 1;
      if (fz && tap(xin, h, w, r->convs[b.c1].cin)) return 1;      // the PREVIOUS block's output was formed by this conv1's loader
      if (train && !fuse1 && !fuse1x && bnact(b.c1, narrow[0], (long)B * h1 * w1, 1, nullptr, -1)) return 1;
      const bool fuse2_ = train && c3.cin % 64 == 0;              // conv3 applies bn2 + relu in its loader
      if (conv(b.c2, narrow[0], h1, w1, narrow[1], nullptr, 1, &h2, &w2, (fuse1 || fuse1x) ? b.c1 : -1, c3_sums || !fuse2_)) return 1;
      // train: bn2 + relu are applied by conv3's loader (no separate pass over the 3x3 output); needs whole 64-channel
      // (f32: 32) K tiles, which every bottleneck width satisfies
      const bool fuse2 = train && r->convs[b.c3].cin % 64 == 0;
      if (train && !fuse2 && bnact(b.c2, narrow[1], (long)B * h2 * w2, 1, nullptr, -1)) return 1;
      const void* res = xin;
      if (b.ds >= 0) {
        if (conv(b.ds, xin, h, w, wide[dsb], nullptr, 0, &hd, &wd, bi == 0 ? stem_in : -1)) return 1;
        res = wide[dsb];
      }
      bool defer = false, b2b = false;
      // 14 x 14 blocks (256 -> 1024 -> next conv1 1024 -> 256) and 28 x 28 blocks (128 -> 512 -> 128), train AND eval: conv3 + block
      // end + next conv1 as one kernel (st_conv_c3c1).  Train: conv3 runs here as a statistics-only pass (y == NULL); an identity that
      // still needs its own BatchNorm (the block after a downsample conv) keeps the separate path.  Eval: conv3 is not launched at all.
      // ST_C3C1: bit 0 = the 14 x 14 blocks, bit 1 = the 28 x 28 blocks (A/B switch; default both)
      static const int c3c1_env = [] { const char* e = getenv("ST_C3C1"); return e ? atoi(e) : 3; }();
      // the block behind a downsample conv (its identity still needs that conv's BatchNorm): the kernel's IDBN form.  ST_C3C1_IDBN=0: A/B switch
      static const bool c3c1_idbn = [] { const char* e = getenv("ST_C3C1_IDBN"); return !e || atoi(e) != 0; }();
      bool c3c1 = false;
      if (use_img && (c3c1_env & (c3.cout == 1024 ? 1 : 2)) && bi + 1 < r->blocks.size()) {
        const BlockL& nb = r->blocks[bi + 1];
        const ConvL& n1 = r->convs[nb.c1];
        // the kernel indexes both fragment-major filter copies with fixed tile permutations: conv3 packed with ntw = 2, conv1 with N / 64
        c3c1 = n1.k == 1 && n1.stride == 1 && c3.k == 1 && c3.stride == 1 && c3.cout == n1.cin && st_conv_c3c1_supported(c3.cin, c3.cout, n1.cout) &&
               c3.ntw == 2 && n1.ntw == n1.cout / 64 && (!train || (fuse2_ && (b.ds < 0 || c3c1_idbn))) && (long)B * h2 * w2 * c3.cout * 2 < (1L << 31);
      }
      if (c3c1) { defer = true; b2b = true; }
      else if (train && use_img && kfuse_env && bi + 1 < r->blocks.size()) {
        const BlockL& nb = r->blocks[bi + 1];
        const ConvL& n1 = r->convs[nb.c1];
        defer = n1.k == 1 && n1.stride == 1 && n1.cin == 256 && n1.ntw > 0 && !use_astat(n1) && c3.cout == n1.cin &&
                st_conv1x1_kfuse_supported(n1.cin, n1.cout) == n1.ntw;
        // the 56 x 56 boundaries: conv3 (64 -> 256) is recomputed inside the fused kernel (st_conv_b2b) and runs here for its statistics only
        // conv_b2b_kernel indexes the fragment-major filters with FIXED tile permutations: conv3 packed with ntw = 2, the next conv1
        // with ntw = N / 64 (conv_b2b.hip:63-65); a retuned pw_cfg table / ST_PW_CFG that packs another ntw must not reach it
        b2b = defer && b2b_env && fuse2 && c3.k == 1 && c3.stride == 1 && c3.ntw == 2 && !use_astat(c3) &&
              c3.ntw == st_conv1x1_wreg_supported(c3.cin, c3.cout) && n1.ntw == st_conv1x1_wreg_supported(n1.cin, n1.cout) &&
              n1.ntw == n1.cout / 64 && st_conv_b2b_supported(c3.cin, c3.cout, n1.cout);
      }
      // the 28 x 28 blocks (conv3 128 -> 512): the same recomputation, stopping at the block output -- conv3 runs for its statistics,
      // st_conv_b2b (N = 0) forms x = relu(bn3(conv3(..)) + identity) from the narrow tensor: the raw 512-channel tensor (103 MB at
      // B = 128) is neither written nor read
      const bool lite = !defer && train && use_img && b2b_env && fuse2 && c3.k == 1 && c3.stride == 1 && c3.ntw == 2 && !use_astat(c3) &&
                        c3.ntw == st_conv1x1_wreg_supported(c3.cin, c3.cout) && st_conv_b2b_supported(c3.cin, c3.cout, 0);
      if (c3c1 && !train) { h3 = h2; w3 = w2; }                  // eval: conv3 is computed where it is consumed (the next block's conv1 launch)
      else if (conv(b.c3, narrow[1], h2, w2, (b2b || lite) ? nullptr : wide[oth], res, 1, &h3, &w3, fuse2 ? b.c2 : -1)) return 1;
      if (defer) pend = Pending{true, b.c3, res, b.ds, oth, b.ds >= 0 ? dsb : cur, b2b, narrow[1], b.c2};
      else if (lite) {
        const ConvL& c2l = r->convs[b.c2];
        st_conv_b2b_desc k;
        memset(&k, 0, sizeof(k));
        k.raw2 = narrow[1]; k.w3_frag = reinterpret_cast<const char*>(weights) + c3.woff_frag * es; k.identity = res; k.x_out = wide[oth];
        k.bn2_stats = stats + tab.soff[b.c2]; k.bn2_gamma = bn_gamma + c2l.bnoff; k.bn2_beta = bn_beta + c2l.bnoff; k.bn2_replicas = tab.rep[b.c2];
        k.bn3_stats = stats + tab.soff[b.c3]; k.bn3_gamma = bn_gamma + c3.bnoff; k.bn3_beta = bn_beta + c3.bnoff; k.bn3_replicas = tab.rep[b.c3];
        if (b.ds >= 0) {
          const ConvL& rc = r->convs[b.ds];
          k.id_stats = stats + tab.soff[b.ds]; k.id_gamma = bn_gamma + rc.bnoff; k.id_beta = bn_beta + rc.bnoff; k.id_replicas = tab.rep[b.ds];
        }
        k.count = tab.count[b.c3]; k.eps = eps; k.rows = (long)B * h3 * w3; k.C1 = c3.cin; k.C2 = c3.cout; k.N = 0;
        if (st_conv_b2b(&k, stream)) return 1;
      } else if (train && bnact(b.c3, wide[oth], (long)B * h3 * w3, 1, res, b.ds)) return 1;
    } else {
      const ConvL& c1 = r->convs[b.c1];
      const ConvL& c2 = r->convs[b.c2];
      const int h1p = conv_out(h, c1.k, c1.stride, c1.pad), w1p = conv_out(w, c1.k, c1.stride, c1.pad);
      const bool fuse1 = train && use_img && c2.ntw > 0 && st_conv3x3_img_supported(h1p, w1p, c2.cin, c2.cout) == c2.ntw;
      if (conv(b.c1, xin, h, w, narrow[0], nullptr, 1, &h1, &w1)) return 1;
      if (train && !fuse1 && bnact(b.c1, narrow[0], (long)B * h1 * w1, 1, nullptr, -1)) return 1;
      const void* res = xin;
      if (b.ds >= 0) {
        if (conv(b.ds, xin, h, w, wide[dsb], nullptr, 0, &hd, &wd)) return 1;
        res = wide[dsb];
      }
      // (eval mode hands the residual to the conv epilogue, which the image-resident kernel does not have: st_conv there)
      if (conv(b.c2, narrow[0], h1, w1, wide[oth], res, 1, &h3, &w3, fuse1 ? b.c1 : -1)) return 1;
      if (train && bnact(b.c2, wide[oth], (long)B * h3 * w3, 1, res, b.ds)) return 1;
    }
    h = h3; w = w3; cur = oth;
    if (!pend.on && tap(wide[cur], h, w, r->bottleneck ? r->convs[b.c3].cout : r->convs[b.c2].cout)) return 1;
  }

  const size_t feat_bytes = (size_t)B * h * w * r->feat_dim * es;
  if (feat_nhwc_out) {
    if (hipMemcpyAsync(feat_nhwc_out, wide[cur], feat_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) {
      st_set_error("st_resnet_forward: copy failed"); return 1;
    }
  }
  if (pooled_out && st_global_avgpool(wide[cur], pooled_out, dt, pooled_dtype, B, h * w, r->feat_dim, stream)) return 1;
  if (ncp_out && st_nhwc_to_ncp_f32(wide[cur], ncp_out, dt, B, h * w, r->feat_dim, stream)) return 1;

  if (train == 2) {
    std::lock_guard<std::mutex> lk(r->mu);
    r->pending[workspace] = PendingUpdate{tab, stats};
  } else if (train) {
    hipLaunchKernelGGL(bn_update_all_kernel, dim3((total + 255) / 256), dim3(256), 0, st, stats, bn_running_mean,
                       bn_running_var, total, momentum, tab);
    ST_LAUNCH_CHECK();
  }
  return 0;
}

// Second half of a train == 2 forward: the momentum update of every running_mean / running_var from the statistics that
// forward left in `workspace`.  Stream-ordered after that forward by the caller; calls for successive minibatches must be
// ordered among themselves (an event chain when the forwards run on different streams).
extern "C" int st_resnet_update_running(const st_resnet* r, const void* workspace, float* bn_running_mean, float* bn_running_var,
                                        float momentum, void* stream) {
  ST_CHECK(r && workspace && bn_running_mean && bn_running_var, "st_resnet_update_running: null pointer");
  PendingUpdate p;
  {
    std::lock_guard<std::mutex> lk(r->mu);
    auto it = r->pending.find(workspace);
    ST_CHECK(it != r->pending.end(), "st_resnet_update_running: no train == 2 forward is pending for this workspace");
    p = it->second;
    r->pending.erase(it);
  }
  const int total = (int)r->bntotal;
  hipLaunchKernelGGL(bn_update_all_kernel, dim3((total + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p.stats,
                     bn_running_mean, bn_running_var, total, momentum, p.tab);
  ST_LAUNCH_CHECK();
  return 0;
}
