// Vocabulary projection + log-softmax + NLL without a logits tensor (gfx950 / MI355X, bf16; reference rnn.py:33 `self.linear(..)` followed by
// main.py:149 `nn.CrossEntropyLoss()`).
//
// The launch chain writes the N_tok x V logits (32 MB at 1600 tokens x 10000 entries), reads them for the loss, reads them again for the
// gradient and overwrites them.  Here the logits only ever exist as MFMA accumulators of a 64-token x 256-entry tile:
//   forward   vocab_ce_kernel<0>: tile logits -> per token the (max, sum of exponentials) of the tile's 256 entries -> partial[tile][token],
//             and the target's logit where the tile holds it;  vocab_ce_reduce_kernel merges the partials: lse[token], loss += (lse - x_t) / N
//   backward  vocab_ce_kernel<1>: the SAME tile product again (same MFMA order: the same bits), dlogits = (exp(x - lse) - onehot) * scale in bf16,
//             staged through LDS and stored as full 512-byte rows -- written once, never read by this file (st_rnn_backward's GEMMs consume it).
// Tile: the 64 tokens' hidden states (64 x 512 bf16) sit in LDS with padded rows; a wave owns 64 of the 256 entries and streams their weight
// rows straight from memory as MFMA operands through a 4-K-step register ring.  Two workgroups per CU (70 KB of LDS, <= 256 registers).
#include "common.h"
#include "rnn_kernels.h"

namespace {

constexpr int VH = 512;                      // hidden size (K)
constexpr int VPIX = 2 * VH + 32;            // padded LDS row of the token tile (bytes)
constexpr int VTM = 4, VBM = 16 * VTM;       // 64 tokens per workgroup
constexpr int VNT = 4;                       // 16-entry tiles per wave
constexpr int VBN = 64 * VNT;                // 256 vocabulary entries per workgroup (64 per wave)
constexpr int VKS = VH / 32;                 // 16 K-steps
constexpr int VWR = 4;                       // weight-fragment ring (K-steps in flight)
constexpr int VLDS = VBM * VPIX + 4 * VBM * 2 * 4;   // token tile + cross-wave (max, sum) exchange
// (tile size at 1600 tokens x 10000 entries: 64 x 128 at two per CU = 2000 workgroups, 3.9 rounds: 47 - 50 us per pass; 128 x 256 at one per
// CU = 520 workgroups, 2.03 rounds: the same; 64 x 256 at two per CU = 1000 workgroups, 1.95 rounds)

struct VceArgs {
  const bf16_t* y;        // [n][512] top-layer outputs (packed-sequence rows)
  const bf16_t* w;        // [V][512]
  const float* bias;      // [V]
  const long* target;     // [n]
  float* partial;         // MODE 0: [ntile][n][2]
  float* tgt_logit;       // MODE 0: [n]
  const float* lse;       // MODE 1: [n]
  bf16_t* dlogits;        // MODE 1: [n][ldd]
  const float* gscale_dev; float gscale;   // MODE 1: dlogits = (p - onehot) * gscale * (*gscale_dev)
  int n, V, ntile, ldd;
};

__device__ __forceinline__ f32x4 vmfma(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void vocab_ce_kernel(VceArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xch = reinterpret_cast<float*>(smem + VBM * VPIX);          // [4 waves][128 tokens][2]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, q4 = lane >> 4;
  const int mtiles = (a.n + VBM - 1) / VBM;
  const int nt = blockIdx.x / mtiles, mt = blockIdx.x - nt * mtiles;   // the token tiles of one entry tile are neighbours: its weight rows stay in L2
  const int m0 = mt * VBM, n0 = nt * VBN;

  // weight fragments of this wave's four 16-entry tiles stream straight from memory (row-major W_lin: a lane's fragment is 16 contiguous
  // bytes of one row); the first ring entries are in flight while the token tile is staged
  const bf16_t* wp[VNT];
#pragma unroll
  for (int j = 0; j < VNT; ++j) { int r = n0 + wid * 64 + 16 * j + r16; r = r < a.V ? r : a.V - 1; wp[j] = a.w + (size_t)r * VH + q4 * 16; }
  // K order: K-steps 2p and 2p + 1 take, in lane group q4, the elements [64 p + 16 q4, + 8) and [64 p + 16 q4 + 8, + 8): the two 16-byte loads
  // of a row are neighbours, so a wave's pair of loads covers whole 128-byte lines (row-major rows read 64 bytes at a time fetched every line
  // twice).  The token operands in LDS are read in the same order; forward and backward use the same order (the same logits).
  auto koff = [](int ks) { return (ks >> 1) * 64 + (ks & 1) * 8; };   // element offset inside a row, without the lane group's 16 q4
  u32x4 wq[VWR][VNT];
#pragma unroll
  for (int g = 0; g < VWR; ++g)
#pragma unroll
    for (int j = 0; j < VNT; ++j) wq[g][j] = *reinterpret_cast<const u32x4*>(wp[j] + koff(g));
  {   // token tile -> LDS: 64 rows x 64 16-byte chunks, 16 per thread (rows past n re-read row n - 1; never used)
#pragma unroll
    for (int h = 0; h < VBM / 64; ++h) {
      u32x4 v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int c = tid + 256 * (16 * h + i), row = c >> 6, ch = c & 63;
        int m = m0 + row; m = m < a.n ? m : a.n - 1;
        v[i] = *reinterpret_cast<const u32x4*>(a.y + (size_t)m * VH + ch * 8);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) { const int c = tid + 256 * (16 * h + i); *reinterpret_cast<u32x4*>(smem + (c >> 6) * VPIX + (c & 63) * 16) = v[i]; }
    }
  }
  __syncthreads();
  f32x4 acc[VTM][VNT];
#pragma unroll
  for (int i = 0; i < VTM; ++i)
#pragma unroll
    for (int j = 0; j < VNT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const char* abase = smem + r16 * VPIX + q4 * 32;
  auto read_a = [&](u32x4 (&f)[VTM], int ks) {
#pragma unroll
    for (int i = 0; i < VTM; ++i) f[i] = *reinterpret_cast<const u32x4*>(abase + i * 16 * VPIX + koff(ks) * 2);
  };
  u32x4 fa0[VTM], fa1[VTM];
  read_a(fa0, 0);
#pragma clang loop unroll(full)
  for (int ks = 0; ks < VKS; ++ks) {
    u32x4 (&fa)[VTM] = (ks & 1) ? fa1 : fa0;
    u32x4 (&fn)[VTM] = (ks & 1) ? fa0 : fa1;
    if (ks + 1 < VKS) read_a(fn, ks + 1);
#pragma unroll
    for (int i = 0; i < VTM; ++i)
#pragma unroll
      for (int j = 0; j < VNT; ++j) acc[i][j] = vmfma(wq[ks % VWR][j], fa[i], acc[i][j]);
    if (ks + VWR < VKS) {
#pragma unroll
      for (int j = 0; j < VNT; ++j) wq[ks % VWR][j] = *reinterpret_cast<const u32x4*>(wp[j] + koff(ks + VWR));
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  // lane: token m0 + 16 i + r16, entries n0 + 64 wid + 16 j + 4 q4 + e
  float bz[VNT][4]; bool vok[VNT][4];
#pragma unroll
  for (int j = 0; j < VNT; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int v = n0 + wid * 64 + 16 * j + 4 * q4 + e;
      vok[j][e] = v < a.V;
      bz[j][e] = vok[j][e] ? a.bias[v] : 0.f;
    }
  if constexpr (MODE == 0) {
#pragma unroll
    for (int i = 0; i < VTM; ++i) {
      const int m = m0 + 16 * i + r16;
      long t = a.target[m < a.n ? m : a.n - 1];
      t = t < 0 ? 0 : (t >= a.V ? a.V - 1 : t);
      float x[4 * VNT], mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < VNT; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          x[4 * j + e] = vok[j][e] ? acc[i][j][e] + bz[j][e] : -INFINITY;
          mx = fmaxf(mx, x[4 * j + e]);
          if (vok[j][e] && m < a.n && (long)(n0 + wid * 64 + 16 * j + 4 * q4 + e) == t) a.tgt_logit[m] = x[4 * j + e];
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));      // over the wave's 64 entries
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < 4 * VNT; ++c) s += x[c] == -INFINITY ? 0.f : __expf(x[c] - mx);
      s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
      if (q4 == 0) { xch[(wid * VBM + 16 * i + r16) * 2] = mx; xch[(wid * VBM + 16 * i + r16) * 2 + 1] = s; }
    }
    __syncthreads();
    if (tid < VBM && m0 + tid < a.n) {                               // the four waves' (max, sum) of a token -> the tile's partial
      float M = -INFINITY, S = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float mw = xch[(w * VBM + tid) * 2], sw = xch[(w * VBM + tid) * 2 + 1];
        const float nm = fmaxf(M, mw);
        S = (M == -INFINITY ? 0.f : S * __expf(M - nm)) + (mw == -INFINITY ? 0.f : sw * __expf(mw - nm));
        M = nm;
      }
      float* p = a.partial + ((size_t)nt * a.n + m0 + tid) * 2;      // [tile][token]: the reduction reads coalesced
      p[0] = M; p[1] = S;
    }
  } else {
    float gs = a.gscale;
    if (a.gscale_dev) gs *= *a.gscale_dev;
    __syncthreads();                                                  // every wave has read its last token operands: the tile area is free
    char* stage = smem;                                               // [128 tokens][256 entries] bf16, rows of 512 + 16 bytes
    constexpr int SROW = VBN * 2 + 16;
#pragma unroll
    for (int i = 0; i < VTM; ++i) {
      const int m = m0 + 16 * i + r16;
      const int mc = m < a.n ? m : a.n - 1;
      long t = a.target[mc];
      t = t < 0 ? 0 : (t >= a.V ? a.V - 1 : t);
      const float l = a.lse[mc];
#pragma unroll
      for (int j = 0; j < VNT; ++j) {
        float d[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int v = n0 + wid * 64 + 16 * j + 4 * q4 + e;
          const float p = vok[j][e] ? __expf(acc[i][j][e] + bz[j][e] - l) : 0.f;
          d[e] = vok[j][e] ? (p - ((long)v == t ? 1.f : 0.f)) * gs : 0.f;   // pad entries (V .. ldd) are zero: they feed GEMMs as K
        }
        *reinterpret_cast<u32x2*>(stage + (16 * i + r16) * SROW + (wid * 64 + 16 * j + 4 * q4) * 2) = u32x2{pack_bf16x2(d[0], d[1]), pack_bf16x2(d[2], d[3])};
      }
    }
    __syncthreads();
    // 64 rows x 32 16-byte chunks: eight per thread, a row's 512 bytes leave as one contiguous piece (columns past ldd are not written)
#pragma unroll
    for (int i = 0; i < VBM / 8; ++i) {
      const int c = tid + 256 * i, row = c >> 5, ch = c & 31;
      const int m = m0 + row, col = n0 + ch * 8;
      if (m < a.n && col < a.ldd)
        *reinterpret_cast<u32x4*>(a.dlogits + (size_t)m * a.ldd + col) = *reinterpret_cast<const u32x4*>(stage + row * SROW + ch * 16);
    }
  }
}

// lse[m] = log sum_v exp(x[m][v]) from the tiles' partials (merged in tile order); loss += (lse - x[m][target]) / n
__global__ __launch_bounds__(256) void vocab_ce_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ tgt_logit,
                                                              float* __restrict__ lse, float* __restrict__ loss, int n, int ntile, float inv_rows) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  float term = 0.f;
  if (m < n) {
    float M = -INFINITY, S = 0.f;
    for (int j0 = 0; j0 < ntile; j0 += 8) {                           // eight partials in flight, merged in tile order
      float2 pj[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) pj[u] = *reinterpret_cast<const float2*>(partial + ((size_t)(j0 + u < ntile ? j0 + u : ntile - 1) * n + m) * 2);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (j0 + u < ntile) {
          const float mj = pj[u].x, sj = pj[u].y;
          const float nm = fmaxf(M, mj);
          S = (M == -INFINITY ? 0.f : S * __expf(M - nm)) + (mj == -INFINITY ? 0.f : sj * __expf(mj - nm));
          M = nm;
        }
      }
    }
    const float l = M + __logf(S);
    lse[m] = l;
    term = (l - tgt_logit[m]) * inv_rows;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) term += __shfl_xor(term, o, 64);
  if ((threadIdx.x & 63) == 0 && loss) atomicAdd(loss, term);
}

int vce_launch_attr() {
  static int attr_set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vocab_ce_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, VLDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vocab_ce_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, VLDS);
    attr_set[dev] = 1;
  }
  return 0;
}

}  // namespace

// 1: the fused loss exists for this configuration (bf16, H = 512); 0: use st_rnn_forward's logits + st_cross_entropy
int vocab_ce_supported(int dtype, int H) { return dtype == ST_BF16 && H == VH ? 1 : 0; }
int vocab_ce_tiles(int V) { return (V + VBN - 1) / VBN; }

// forward: y [n][512] bf16, w [V][512] bf16 -> lse[n], *loss += mean NLL.  partial: n * vocab_ce_tiles(V) * 2 floats; tgt: n floats
int vocab_ce_forward(const void* y, const void* w, const float* bias, const long* target, int n, int V, float* partial, float* tgt,
                     float* lse, float* loss, hipStream_t st) {
  if (n <= 0) return 0;
  vce_launch_attr();
  VceArgs a{};
  a.y = reinterpret_cast<const bf16_t*>(y); a.w = reinterpret_cast<const bf16_t*>(w); a.bias = bias; a.target = target;
  a.partial = partial; a.tgt_logit = tgt; a.n = n; a.V = V; a.ntile = vocab_ce_tiles(V);
  const int mtiles = (n + VBM - 1) / VBM;
  hipLaunchKernelGGL(vocab_ce_kernel<0>, dim3(a.ntile * mtiles), dim3(256), VLDS, st, a);
  hipLaunchKernelGGL(vocab_ce_reduce_kernel, dim3((n + 63) / 64), dim3(64), 0, st, partial, tgt, lse, loss, n, a.ntile, 1.0f / n);
  ST_LAUNCH_CHECK();
  return 0;
}

// backward: dlogits[n][ldd] bf16 = (softmax - onehot) * gscale / n (* *gscale_dev); columns V .. ldd are written as zeros
int vocab_ce_dlogits(const void* y, const void* w, const float* bias, const long* target, const float* lse, int n, int V,
                     void* dlogits, int ldd, float gscale, const float* gscale_dev, hipStream_t st) {
  if (n <= 0) return 0;
  ST_CHECK(ldd % 8 == 0 && ldd >= V, "vocab_ce_dlogits: ldd=%d must be a multiple of 8 and >= V", ldd);
  vce_launch_attr();
  VceArgs a{};
  a.y = reinterpret_cast<const bf16_t*>(y); a.w = reinterpret_cast<const bf16_t*>(w); a.bias = bias; a.target = target;
  a.lse = lse; a.dlogits = reinterpret_cast<bf16_t*>(dlogits); a.gscale = gscale / n; a.gscale_dev = gscale_dev;
  a.n = n; a.V = V; a.ldd = ldd; a.ntile = (ldd + VBN - 1) / VBN;     // the tiles also cover the pad columns
  const int mtiles = (n + VBM - 1) / VBM;
  hipLaunchKernelGGL(vocab_ce_kernel<1>, dim3(a.ntile * mtiles), dim3(256), VLDS, st, a);
  ST_LAUNCH_CHECK();
  return 0;
}
