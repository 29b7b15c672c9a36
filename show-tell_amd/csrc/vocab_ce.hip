// Vocabulary projection + log-softmax + NLL without a logits tensor (gfx950 / MI355X, bf16; reference rnn.py:33 `self.linear(..)` followed by
// main.py:149 `nn.CrossEntropyLoss()`).
//
// The launch chain writes the N_tok x V logits (32 MB at 1600 tokens x 10000 entries), reads them for the loss, reads them again for the
// gradient and overwrites them.  Here the logits only ever exist as MFMA accumulators of a 32-token x 128-entry tile:
//   forward   vocab_ce_kernel<0>: tile logits -> per token the (max, sum of exponentials) of the tile's 128 entries -> partial[tile][token],
//             and the target's logit where the tile holds it;  vocab_ce_reduce_kernel merges the partials: lse[token], loss += (lse - x_t) / N
//   backward  vocab_ce_kernel<1>: the SAME tile product again (same MFMA order: the same bits), dlogits = (exp(x - lse) - onehot) * scale in bf16,
//             staged through LDS and stored as 256-byte row pieces -- written once, never read by this file (st_rnn_backward's GEMMs consume it).
#include "common.h"
#include "rnn_kernels.h"

namespace {

constexpr int VH = 512;                      // hidden size (K)
constexpr int VPIX = 2 * VH + 32;            // padded LDS row of a token tile (bytes)
constexpr int VTM = 2, VBM = 16 * VTM;       // 32 tokens per token tile (its staging registers, the weights and the accumulators: 256 registers)
constexpr int VNL = VBM * 64 / 256;          // 16-byte chunks of a token tile per thread
constexpr int VNT = 2;                       // 16-entry tiles per wave
constexpr int VBN = 64 * VNT;                // 128 vocabulary entries per workgroup (32 per wave)
constexpr int VKS = VH / 32;                 // 16 K-steps
constexpr int VSROW = VBN * 2 + 16;          // MODE 1: padded row of the dlogits staging tile (bytes)
constexpr int VTILE = VBM * VPIX;            // one token tile in LDS
constexpr int VLDS = VTILE + 4 * VBM * 2 * 4;   // one token tile (the dlogits staging tile overlays it) + (max, sum) exchange: 70 KB, two per CU

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
  int n, V, ntile, ldd, nsplit;
};

__device__ __forceinline__ f32x4 vmfma(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
}

// WEIGHTS STATIONARY: a workgroup owns 128 vocabulary entries -- each wave keeps its 32 entries' weight rows (32 x 512 bf16 = 128 registers of
// MFMA operands, read once) -- and walks the token tiles mt = split, split + nsplit, ..: tile mt + nsplit is requested from memory (into
// registers) while tile mt is multiplied out of LDS.  TWO workgroups per CU: a tile's request is one round trip that the workgroup's own
// 128 MFMAs per wave cannot cover; the other workgroup's can.  Measured before: weight rows streamed per (64 x 256) tile: 43 us per pass (17 of
// them waiting for the rows, 19 staging and launching 1000 workgroups; 64 x 128 / 128 x 256 tiles, deeper rings, line-paired rows: the same);
// this form with a double-buffered tile at one workgroup per CU: 49 us.
template <int MODE>
__global__ __launch_bounds__(256, 2) void vocab_ce_kernel(VceArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xch = reinterpret_cast<float*>(smem + VTILE);               // [4 waves][64 tokens][2]
  char* stage = smem;                                                // MODE 1: [64 tokens][128 entries] bf16 over the (consumed) token tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, q4 = lane >> 4;
  const int mtiles = (a.n + VBM - 1) / VBM;
  const int nt = blockIdx.x / a.nsplit, sp = blockIdx.x - nt * a.nsplit;   // the splits of one entry tile are neighbours: its weight rows stay in L2
  const int n0 = nt * VBN;
  if (sp >= mtiles) return;

  // token tile -> registers -> LDS: rows x 64 16-byte chunks (rows past n re-read row n - 1; never used)
  u32x4 tv[VNL];
  auto tile_request = [&](int mt) {
#pragma unroll
    for (int i = 0; i < VNL; ++i) {
      const int c = tid + 256 * i, row = c >> 6, ch = c & 63;
      int m = mt * VBM + row; m = m < a.n ? m : a.n - 1;
      tv[i] = *reinterpret_cast<const u32x4*>(a.y + (size_t)m * VH + ch * 8);
    }
  };
  auto tile_store = [&]() {
#pragma unroll
    for (int i = 0; i < VNL; ++i) { const int c = tid + 256 * i; *reinterpret_cast<u32x4*>(smem + (c >> 6) * VPIX + (c & 63) * 16) = tv[i]; }
  };
  tile_request(sp);
  // the wave's weight operands: entries n0 + 32 wid + 16 j + r16, all of K (rows past V re-read row V - 1; masked in the epilogue)
  u32x4 wf[VNT][VKS];
#pragma unroll
  for (int j = 0; j < VNT; ++j) {
    int r = n0 + wid * 32 + 16 * j + r16; r = r < a.V ? r : a.V - 1;
    const bf16_t* wp = a.w + (size_t)r * VH + q4 * 8;
#pragma unroll
    for (int ks = 0; ks < VKS; ++ks) wf[j][ks] = *reinterpret_cast<const u32x4*>(wp + ks * 32);
  }
  // lane: token (tile) 16 i + r16, entries n0 + 32 wid + 16 j + 4 q4 + e
  float bz[VNT][4]; bool vok[VNT][4];
#pragma unroll
  for (int j = 0; j < VNT; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int v = n0 + wid * 32 + 16 * j + 4 * q4 + e;
      vok[j][e] = v < a.V;
      bz[j][e] = vok[j][e] ? a.bias[v] : 0.f;
    }
  float gs = a.gscale;
  if (MODE == 1 && a.gscale_dev) gs *= *a.gscale_dev;
  tile_store();
  __syncthreads();

  for (int mt = sp; mt < mtiles; mt += a.nsplit) {
    const int m0 = mt * VBM;
    const bool more = mt + a.nsplit < mtiles;                        // (workgroup-uniform)
    if (more) tile_request(mt + a.nsplit);
    f32x4 acc[VTM][VNT];
#pragma unroll
    for (int i = 0; i < VTM; ++i)
#pragma unroll
      for (int j = 0; j < VNT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* abase = smem + r16 * VPIX + q4 * 16;
#pragma unroll
    for (int ks = 0; ks < VKS; ++ks) {
      u32x4 fa[VTM];
#pragma unroll
      for (int i = 0; i < VTM; ++i) fa[i] = *reinterpret_cast<const u32x4*>(abase + i * 16 * VPIX + ks * 64);
#pragma unroll
      for (int i = 0; i < VTM; ++i)
#pragma unroll
        for (int j = 0; j < VNT; ++j) acc[i][j] = vmfma(wf[j][ks], fa[i], acc[i][j]);
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
            if (vok[j][e] && m < a.n && (long)(n0 + wid * 32 + 16 * j + 4 * q4 + e) == t) a.tgt_logit[m] = x[4 * j + e];
          }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));      // over the wave's 32 entries
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 4 * VNT; ++c) s += x[c] == -INFINITY ? 0.f : __expf(x[c] - mx);
        s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
        if (q4 == 0) { xch[(wid * VBM + 16 * i + r16) * 2] = mx; xch[(wid * VBM + 16 * i + r16) * 2 + 1] = s; }
      }
      __syncthreads();
      if (tid < VBM && m0 + tid < a.n) {                             // the four waves' (max, sum) of a token -> the tile's partial
        float M = -INFINITY, S = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const float mw = xch[(w * VBM + tid) * 2], sw = xch[(w * VBM + tid) * 2 + 1];
          const float nm = fmaxf(M, mw);
          S = (M == -INFINITY ? 0.f : S * __expf(M - nm)) + (mw == -INFINITY ? 0.f : sw * __expf(mw - nm));
          M = nm;
        }
        float* p = a.partial + ((size_t)nt * a.n + m0 + tid) * 2;    // [tile][token]: the reduction reads coalesced
        p[0] = M; p[1] = S;
      }
    } else {
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
            const int v = n0 + wid * 32 + 16 * j + 4 * q4 + e;
            const float p = vok[j][e] ? __expf(acc[i][j][e] + bz[j][e] - l) : 0.f;
            d[e] = vok[j][e] ? (p - ((long)v == t ? 1.f : 0.f)) * gs : 0.f;   // pad entries (V .. ldd) are zero: they feed GEMMs as K
          }
          if (i == 0 && j == 0) __syncthreads();                      // (uniform) every wave has read its last operands: the tile area is free
          *reinterpret_cast<u32x2*>(stage + (16 * i + r16) * VSROW + (wid * 32 + 16 * j + 4 * q4) * 2) = u32x2{pack_bf16x2(d[0], d[1]), pack_bf16x2(d[2], d[3])};
        }
      }
      __syncthreads();
      // rows x 16 16-byte chunks: a row's 256 bytes leave as one contiguous piece (columns past ldd are not written)
#pragma unroll
      for (int i = 0; i < VBM * 16 / 256; ++i) {
        const int c = tid + 256 * i, row = c >> 4, ch = c & 15;
        const int m = m0 + row, col = n0 + ch * 8;
        if (m < a.n && col < a.ldd)
          *reinterpret_cast<u32x4*>(a.dlogits + (size_t)m * a.ldd + col) = *reinterpret_cast<const u32x4*>(stage + row * VSROW + ch * 16);
      }
    }
    __syncthreads();                                                 // tile, staging tile and exchange area are consumed
    if (more) { tile_store(); __syncthreads(); }                     // tile mt + nsplit is in LDS
  }
}

// lse[m] = log sum_v exp(x[m][v]) from the tiles' partials; loss += (lse - x[m][target]) / n.  A block = 64 tokens x 4 tile groups (group g
// merges tiles g, g + 4, .. in order, the groups meet in LDS in group order): the merge is a chain of dependent exponentials per token.
__global__ __launch_bounds__(256) void vocab_ce_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ tgt_logit,
                                                              float* __restrict__ lse, float* __restrict__ loss, int n, int ntile, float inv_rows) {
  __shared__ float gm[4][64], gsum[4][64];
  const int tk = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int m = blockIdx.x * 64 + tk, mc = m < n ? m : n - 1;
  float M = -INFINITY, S = 0.f;
  for (int j0 = g; j0 < ntile; j0 += 32) {                            // eight partials in flight
    float2 pj[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int j = j0 + 4 * u; pj[u] = *reinterpret_cast<const float2*>(partial + ((size_t)(j < ntile ? j : ntile - 1) * n + mc) * 2); }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (j0 + 4 * u < ntile) {
        const float mj = pj[u].x, sj = pj[u].y;
        const float nm = fmaxf(M, mj);
        S = (M == -INFINITY ? 0.f : S * __expf(M - nm)) + (mj == -INFINITY ? 0.f : sj * __expf(mj - nm));
        M = nm;
      }
    }
  }
  gm[g][tk] = M; gsum[g][tk] = S;
  __syncthreads();
  float term = 0.f;
  if (g == 0) {
    M = -INFINITY; S = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float mj = gm[q][tk], sj = gsum[q][tk];
      const float nm = fmaxf(M, mj);
      S = (M == -INFINITY ? 0.f : S * __expf(M - nm)) + (mj == -INFINITY ? 0.f : sj * __expf(mj - nm));
      M = nm;
    }
    if (m < n) {
      const float l = M + __logf(S);
      lse[m] = l;
      term = (l - tgt_logit[m]) * inv_rows;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) term += __shfl_xor(term, o, 64);
    if (tk == 0 && loss) atomicAdd(loss, term);
  }
}

// token-tile splits per entry tile: as many workgroups as fit one round of the chip (two per CU), never more than the token tiles
int vce_splits(int ntile, int mtiles) {
  int s = 512 / (ntile > 0 ? ntile : 1);                             // two workgroups per CU
  if (s < 1) s = 1;
  if (s > mtiles) s = mtiles;
  return s;
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
  a.nsplit = vce_splits(a.ntile, (n + VBM - 1) / VBM);
  hipLaunchKernelGGL(vocab_ce_kernel<0>, dim3(a.ntile * a.nsplit), dim3(256), VLDS, st, a);
  hipLaunchKernelGGL(vocab_ce_reduce_kernel, dim3((n + 63) / 64), dim3(256), 0, st, partial, tgt, lse, loss, n, a.ntile, 1.0f / n);
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
  a.nsplit = vce_splits(a.ntile, (n + VBM - 1) / VBM);
  hipLaunchKernelGGL(vocab_ce_kernel<1>, dim3(a.ntile * a.nsplit), dim3(256), VLDS, st, a);
  ST_LAUNCH_CHECK();
  return 0;
}
