// The ResNet stem in ONE kernel for gfx950 (MI355X): 7x7 stride-2 convolution + BatchNorm statistics + 3x3 stride-2 max-pool
// (torchvision resnet: conv1 / bn1 / relu / maxpool, the first four modules of the reference's `self.model`, cnn.py:46), bf16.
//
// The two-kernel form (st_conv on the 2x2-blocked image, then st_maxpool3x3s2_bn) writes the 112 x 112 x 64 convolution output
// (205 MB at B = 128) and reads it back to pool it; both halves of that round trip are what the stem costs (153 + 74 us).  Here a
// workgroup computes a 17 x 17 patch of convolution outputs (the 16 x 16 it owns + the row / column the pool windows reach back
// into), keeps it in LDS and writes only the 8 x 8 pooled outputs: 51 MB leave the chip instead of 205 + 51, nothing is re-read.
//
// Train mode needs the batch statistics of the RAW convolution output before bn1 can be applied, so the pool cannot run on
// relu(bn1(x)).  It does not have to: v -> round_bf16(relu(v * sc + sh)) is monotone, non-decreasing for sc >= 0 and non-
// increasing for sc < 0, and sign(sc) = sign(gamma) is known before any statistic.  The kernel pools the raw bf16 values with
// MAX on channels with gamma >= 0 and MIN on the others; bn1 + relu are then applied to the pooled tensor by its consumers'
// loaders (st_conv1x1_wreg's input transform) -- bit for bit what pooling the normalised map gives.  Eval mode applies the
// folded scale / shift + relu before the pool, as the two-kernel form does.
//
// Convolution: the blocked image [B][H/2+3][W/2+3][16] of st_nchw_to_s2d16 turns the 7x7/2 filter into a 4x4/1 one over
// 16-channel pixels, K = 4 rows x 64 = 256 (st_stem_weight_s2d); a filter row's operand is 128 contiguous bytes of the LDS
// tile.  The 289 patch positions are flattened into 19 MFMA tiles of 16 (a lane's LDS base per tile is computed once); a wave
// takes every fourth tile x all 64 output channels.  Filters (32 KB, fragment-major) sit in LDS for the life of the
// workgroup, which walks patches persistently; the next patch's input tile is in flight in registers while this one is
// multiplied.  Two workgroups per CU: one's epilogue / pool runs under the other's MFMAs.
#include "common.h"
#include "prof.h"
#include <stdlib.h>

namespace {

struct StemArgs {
  const bf16_t* x; const bf16_t* w; bf16_t* y;
  float* stats; int srep;                     // train: [srep][sum(64) | sumsq(64)] of the raw convolution output
  const float* gamma;                         // train: bn1.weight, its sign picks max / min
  const float* scale; const float* shift;     // eval: folded bn1
  int B, Hp, Wp, OH, OW, PH, PW, tiles_y, tiles_x, ntiles;
};

constexpr int PT = 17, IT = 20, NPIX = PT * PT;      // patch edge, input tile edge (blocked pixels), patch positions
constexpr int PPIX = 144;                            // LDS bytes per patch position: 64 bf16 + 16 (bank spread)
constexpr int W_BYTES = 64 * 256 * 2;
constexpr int TILE_BYTES = IT * IT * 32;
constexpr int UNION_BYTES = NPIX * PPIX;             // the patch image overlays the (dead) input tile
static_assert(UNION_BYTES >= TILE_BYTES && UNION_BYTES >= 4 * 128 * 4, "union region");
constexpr int STEM_LDS = W_BYTES + ((UNION_BYTES + 255) / 256) * 256;

template <int CTRL> __device__ __forceinline__ float dpp_rot_(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum_(float v) {
  v += dpp_rot_<0x128>(v); v += dpp_rot_<0x124>(v); v += dpp_rot_<0x122>(v); v += dpp_rot_<0x121>(v);
  return v;
}
__device__ __forceinline__ f32x4 mfma_bf16(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
}

typedef short s16x2 __attribute__((ext_vector_type(2)));
// two bf16 -> two int16 whose signed order is the floats' order (sign-magnitude -> two's complement); its own inverse
__device__ __forceinline__ uint32_t bf16x2_key(uint32_t w) {
  s16x2 k = __builtin_bit_cast(s16x2, w);
  k ^= (k >> (short)15) & (short)0x7fff;
  return __builtin_bit_cast(uint32_t, k);
}
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t x, uint32_t y) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, x), __builtin_bit_cast(s16x2, y)));
}

template <bool EVAL>
__global__ __launch_bounds__(256, 2) void stem_pool_kernel(StemArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wl = smem;
  char* un = smem + W_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, q4 = lane >> 4;

  // filters -> LDS, once
#pragma unroll
  for (int i = 0; i < W_BYTES / 16 / 256; ++i)
    reinterpret_cast<u32x4*>(wl)[tid + 256 * i] = reinterpret_cast<const u32x4*>(a.w)[tid + 256 * i];

  // this lane's position in each of its wave's tiles (t = wid + 4 i): flattened patch index p -> (row, column) of the 17 x 17 patch
  // (kept as ONE packed word per tile: row | column << 8 | valid << 16 -- the kernel runs two workgroups per CU on 256 registers)
  int abase[5], ppos[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int p = 16 * (wid + 4 * i) + r16;
    const int pc = p < NPIX ? p : NPIX - 1;
    const int lr = pc / PT, lc = pc - PT * lr;
    abase[i] = (lr * IT + lc) * 32 + q4 * 16;
    ppos[i] = lr | (lc << 8) | ((p < NPIX ? 1 : 0) << 16);
  }
  const int c0 = 16 * q4;                       // this lane's 16 consecutive output channels
  float es[EVAL ? 1 : 16], ess[EVAL ? 1 : 16], scv[EVAL ? 16 : 1], shv[EVAL ? 16 : 1];
  if constexpr (EVAL) {
#pragma unroll
    for (int c = 0; c < 16; ++c) { scv[c] = a.scale[c0 + c]; shv[c] = a.shift[c0 + c]; }
  } else {
#pragma unroll
    for (int c = 0; c < 16; ++c) { es[c] = 0.f; ess[c] = 0.f; }
  }
  // MIN-pooled channels (gamma < 0) carry negated keys: nm = this lane's 16 accumulator channels (two per word), pm = the 8 channels
  // (group cg) of this thread's pool items
  const int cg = tid & 7;
  uint32_t nm[8], pm[4];
#pragma unroll
  for (int w2 = 0; w2 < 8; ++w2) nm[w2] = 0u;
#pragma unroll
  for (int d = 0; d < 4; ++d) pm[d] = 0u;
  if constexpr (!EVAL) {
#pragma unroll
    for (int w2 = 0; w2 < 8; ++w2)
      nm[w2] = (a.gamma[c0 + 2 * w2] < 0.f ? 0xffffu : 0u) | (a.gamma[c0 + 2 * w2 + 1] < 0.f ? 0xffff0000u : 0u);
#pragma unroll
    for (int d = 0; d < 4; ++d)
      pm[d] = (a.gamma[cg * 8 + 2 * d] < 0.f ? 0xffffu : 0u) | (a.gamma[cg * 8 + 2 * d + 1] < 0.f ? 0xffff0000u : 0u);
  }

  const int per_img = a.tiles_y * a.tiles_x;
  u32x4 pre[4];
  auto prefetch = [&](int patch) {              // the 20 x 20 blocked pixels under patch `patch` -> registers (zero outside the image)
    const int b = patch / per_img, rem = patch - b * per_img, ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    const int oy0 = 16 * ty - 1, ox0 = 16 * tx - 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int q = tid + 256 * j;
      q = q < IT * IT * 2 ? q : IT * IT * 2 - 1;
      const int row = q / (2 * IT), c16 = q - row * 2 * IT;
      const int gy = oy0 + row, gx = ox0 + (c16 >> 1);
      const bool ok = gy >= 0 && gy < a.Hp && gx >= 0 && gx < a.Wp;
      const int gyc = gy < 0 ? 0 : (gy < a.Hp ? gy : a.Hp - 1), gxc = gx < 0 ? 0 : (gx < a.Wp ? gx : a.Wp - 1);
      u32x4 v = *reinterpret_cast<const u32x4*>(a.x + (((size_t)b * a.Hp + gyc) * a.Wp + gxc) * 16 + (c16 & 1) * 8);
      pre[j] = ok ? v : u32x4{0u, 0u, 0u, 0u};
    }
  };

  int patch = blockIdx.x;
  if (patch < a.ntiles) prefetch(patch);
  for (; patch < a.ntiles; patch += gridDim.x) {
    const int b = patch / per_img, rem = patch - b * per_img, ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    const int oy0 = 16 * ty - 1, ox0 = 16 * tx - 1;    // convolution output under patch position (0, 0)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = tid + 256 * j;
      if (q < IT * IT * 2) *reinterpret_cast<u32x4*>(un + q * 16) = pre[j];
    }
    __syncthreads();
    {
      const int nxt = patch + gridDim.x;
      prefetch(nxt < a.ntiles ? nxt : a.ntiles - 1);    // always issued (the last one re-reads a tile it never uses)
    }

    f32x4 acc[5][4];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int off = (ks >> 1) * (IT * 32) + (ks & 1) * 64;
      u32x4 fa[5], wf[4];
#pragma unroll
      for (int n = 0; n < 4; ++n) wf[n] = *reinterpret_cast<const u32x4*>(wl + ((n * 8 + ks) * 64 + lane) * 16);
#pragma unroll
      for (int i = 0; i < 5; ++i) fa[i] = *reinterpret_cast<const u32x4*>(un + abase[i] + off);
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[i][n] = mfma_bf16(wf[n], fa[i], acc[i][n]);
      __builtin_amdgcn_sched_barrier(0);                // operands of one K-step at a time (register budget)
    }
    __syncthreads();                                    // every wave is done with the input tile: the patch image overlays it

#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int lrw = ppos[i] & 0xff, lcl = (ppos[i] >> 8) & 0xff;
      if (ppos[i] >> 16) {
        float v[16];
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[4 * n + e] = acc[i][n][e];
        const int oy = oy0 + lrw, ox = ox0 + lcl;
        const bool inimg = oy >= 0 && oy < a.OH && ox >= 0 && ox < a.OW;
        if constexpr (EVAL) {
#pragma unroll
          for (int c = 0; c < 16; ++c) v[c] = fmaxf(v[c] * scv[c] + shv[c], 0.f);
        } else {
          // statistics: every convolution output belongs to exactly one patch (rows / columns 1..16 of it), from the fp32 accumulators
          if (lrw >= 1 && lcl >= 1 && inimg) {
#pragma unroll
            for (int c = 0; c < 16; ++c) { es[c] += v[c]; ess[c] += v[c] * v[c]; }
          }
        }
        // the patch image holds POOL KEYS: the bf16 bits mapped to order-preserving int16 (bf16x2_key), negated on the channels that
        // pool with MIN, and the smallest key at positions outside the map -- the pool is then one packed integer max per two channels
        uint32_t k[8];
#pragma unroll
        for (int w2 = 0; w2 < 8; ++w2) {
          const uint32_t key = bf16x2_key(pack_bf16x2(v[2 * w2], v[2 * w2 + 1])) ^ nm[w2];
          k[w2] = inimg ? key : 0x80008000u;
        }
        char* dst = un + (lrw * PT + lcl) * PPIX + q4 * 32;
        *reinterpret_cast<u32x4*>(dst) = u32x4{k[0], k[1], k[2], k[3]};
        *reinterpret_cast<u32x4*>(dst + 16) = u32x4{k[4], k[5], k[6], k[7]};
      }
    }
    __syncthreads();

    // 3x3 stride-2 pool (pad 1: positions outside the map hold the smallest key) over the patch image -> 8 x 8 outputs x 64 channels
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pp = (tid >> 3) + 32 * j, py = pp >> 3, px = pp & 7;
      const int gpy = 8 * ty + py, gpx = 8 * tx + px;
      if (gpy < a.PH && gpx < a.PW) {
        u32x4 m = u32x4{0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u};
        const char* src = un + (2 * py * PT + 2 * px) * PPIX + cg * 16;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const u32x4 t = *reinterpret_cast<const u32x4*>(src + (dy * PT + dx) * PPIX);
#pragma unroll
            for (int d = 0; d < 4; ++d) m[d] = pk_max_i16(m[d], t[d]);
          }
#pragma unroll
        for (int d = 0; d < 4; ++d) m[d] = bf16x2_key(m[d] ^ pm[d]);
        *reinterpret_cast<u32x4*>(a.y + (((size_t)b * a.PH + gpy) * a.PW + gpx) * 64 + cg * 8) = m;
      }
    }
    __syncthreads();                                    // the next tile overwrites the patch image
  }

  if constexpr (!EVAL) {
    // per-channel sums: 16 position lanes (DPP) -> 4 waves (LDS) -> one replica (atomics over consecutive channels)
    float* sred = reinterpret_cast<float*>(un);        // [4 waves][sum(64) | sumsq(64)]
#pragma unroll
    for (int c = 0; c < 16; ++c) { es[c] = row16_sum_(es[c]); ess[c] = row16_sum_(ess[c]); }
    if (r16 == 0) {
#pragma unroll
      for (int c = 0; c < 16; ++c) { sred[wid * 128 + c0 + c] = es[c]; sred[wid * 128 + 64 + c0 + c] = ess[c]; }
    }
    __syncthreads();
    if (tid < 128) {
      const float s = (sred[tid] + sred[128 + tid]) + (sred[256 + tid] + sred[384 + tid]);
      atomicAdd(a.stats + (a.srep > 1 ? (size_t)(blockIdx.x % a.srep) * 128 : 0) + tid, s);
    }
  }
}

// [64][256] blocked stem filters (st_stem_weight_s2d) -> fragment-major MFMA operands: element ((T * 8 + ks) * 64 + lane) * 8 + j is
// w[16 (r / 4) + 4 T + r % 4][32 ks + 8 (lane >> 4) + j], r = lane & 15: a lane's accumulators of the four tiles T are 16
// consecutive output channels (the ntw = 4 permutation of st_pack_conv_weight_frag)
__global__ void stem_weight_frag_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ out) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;          // 0 .. 16383
  const int j = o & 7, ln = (o >> 3) & 63, ks = (o >> 9) & 7, T = o >> 12;
  const int r = ln & 15, ch = 16 * (r >> 2) + 4 * T + (r & 3), k = 32 * ks + 8 * (ln >> 4) + j;
  out[o] = w[ch * 256 + k];
}

// the same operands straight from the packed [64][7][7][Cpad] filters: st_stem_weight_s2d's index map ([64][4][4][16]: tap (i, j), channel
// (dy * 2 + dx) * 3 + c <- the 7 x 7 weight at (2 i + dy - 1, 2 j + dx - 1), zero outside) composed with the permutation above --
// one launch per forward instead of two
__global__ void stem_weight_frag_packed_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ out, int Cpad) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;          // 0 .. 16383
  const int j8 = o & 7, ln = (o >> 3) & 63, ks = (o >> 9) & 7, T = o >> 12;
  const int r = ln & 15, n = 16 * (r >> 2) + 4 * T + (r & 3), t = 32 * ks + 8 * (ln >> 4) + j8;
  const int i = t >> 6, j = (t >> 4) & 3, ch = t & 15;
  bf16_t v = from_f32<bf16_t>(0.f);
  if (ch < 12) {
    const int blk = ch / 3, c = ch - blk * 3, dy = blk >> 1, dx = blk & 1;
    const int ky = 2 * i + dy - 1, kx = 2 * j + dx - 1;
    if (ky >= 0 && ky < 7 && kx >= 0 && kx < 7) v = w[((n * 7 + ky) * 7 + kx) * Cpad + c];
  }
  out[o] = v;
}

}  // namespace

extern "C" int st_stem_weight_frag_packed(const void* w_packed, int Cpad, void* out, void* stream) {
  ST_CHECK(w_packed && out && Cpad >= 3, "st_stem_weight_frag_packed: null pointer or Cpad=%d < 3", Cpad);
  hipLaunchKernelGGL(stem_weight_frag_packed_kernel, dim3(64), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const bf16_t*>(w_packed), reinterpret_cast<bf16_t*>(out), Cpad);
  ST_LAUNCH_CHECK();
  return 0;
}

extern "C" int st_stem_weight_frag(const void* w_s2d, void* out, void* stream) {
  ST_CHECK(w_s2d && out, "st_stem_weight_frag: null pointer");
  hipLaunchKernelGGL(stem_weight_frag_kernel, dim3(64), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const bf16_t*>(w_s2d), reinterpret_cast<bf16_t*>(out));
  ST_LAUNCH_CHECK();
  return 0;
}

extern "C" int st_stem_conv_pool(const st_stem_conv_pool_desc* d, void* stream) {
  ST_CHECK(d && d->x_s2d && d->w_frag && d->y, "st_stem_conv_pool: null pointer");
  ST_CHECK(d->B > 0 && d->H >= 8 && d->W >= 8 && d->H % 2 == 0 && d->W % 2 == 0, "st_stem_conv_pool: H=%d, W=%d must be even and >= 8", d->H, d->W);
  const bool eval = d->scale != nullptr;
  ST_CHECK((d->scale == nullptr) == (d->shift == nullptr), "st_stem_conv_pool: scale and shift go together");
  ST_CHECK(eval || (d->stats && d->gamma), "st_stem_conv_pool: train mode needs stats and gamma (eval: scale and shift)");
  ST_CHECK(d->stats_replicas >= 0 && d->stats_replicas <= 1024, "st_stem_conv_pool: bad stats_replicas");
  StemArgs a{};
  a.x = reinterpret_cast<const bf16_t*>(d->x_s2d); a.w = reinterpret_cast<const bf16_t*>(d->w_frag); a.y = reinterpret_cast<bf16_t*>(d->y);
  a.stats = d->stats; a.srep = d->stats_replicas; a.gamma = d->gamma; a.scale = d->scale; a.shift = d->shift;
  a.B = d->B; a.Hp = d->H / 2 + 3; a.Wp = d->W / 2 + 3; a.OH = d->H / 2; a.OW = d->W / 2;
  a.PH = (a.OH - 1) / 2 + 1; a.PW = (a.OW - 1) / 2 + 1;
  a.tiles_y = (a.PH + 7) / 8; a.tiles_x = (a.PW + 7) / 8;
  const long nt = (long)d->B * a.tiles_y * a.tiles_x;
  ST_CHECK(nt < (1L << 30), "st_stem_conv_pool: too many patches");
  a.ntiles = (int)nt;
  static int attr_set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_pool_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, STEM_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_pool_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, STEM_LDS);
    attr_set[dev] = 1;
  }
  static const int grid_env = [] { const char* e = getenv("ST_STEM_GRID"); return e ? atoi(e) : 0; }();   // tuning knob
  int grid = grid_env > 0 ? grid_env : 512; if (grid > a.ntiles) grid = a.ntiles;      // two workgroups per CU, persistent over the patches
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  StProfScope prof(20, 2.0 * (double)d->B * a.OH * a.OW * 64.0 * 147.0, st);
  if (eval) hipLaunchKernelGGL(stem_pool_kernel<true>, dim3(grid), dim3(256), STEM_LDS, st, a);
  else hipLaunchKernelGGL(stem_pool_kernel<false>, dim3(grid), dim3(256), STEM_LDS, st, a);
  prof.end(st);
  ST_LAUNCH_CHECK();
  return 0;
}
