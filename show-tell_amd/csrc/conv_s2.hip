// The three stride-2 3x3 convolutions of torchvision's ResNet-50/101/152 (conv2 of the first block of layer2 / layer3 / layer4,
// reference cnn.py:46: 128 -> 128 at 56 -> 28, 256 -> 256 at 28 -> 14, 512 -> 512 at 14 -> 7), bf16, as a K-STREAMING implicit GEMM
// (gfx950 / MI355X).
//
// They were the last layers on the round-1 implicit-GEMM template: 64 - 92 us each, 3.4 - 7.7 x their floors
// (profiles/r03a_layer_table_train.csv).  An image-resident form does not exist for them: a stride-2 output band needs its whole
// input band at four times the area (207 KB for one 14 x 14 x 512 image).  What does carry over from conv1x1_kstream_kernel:
//   * a workgroup owns 112 output positions x 64 NTW output channels; K = 9 C runs (tap, channel) in slabs of 128 channels;
//   * per slab the 112 x 128 activation block goes global -> registers -> LDS (padded rows, two-slab ring, two register sets in
//     flight); here the rows are GATHERED: row (b, ho, wo), tap (kh, kw) reads pixel (2 ho + kh - 1, 2 wo + kw - 1) or zero --
//     unconditional loads from a clamped address, zeroed by a select (a load under a branch would collapse the counted waits);
//   * train mode: the producer's BatchNorm + ReLU (bn1 of the block) ride in the loader (padding stays zero AFTER the transform),
//     replicated statistics summed in the prologue;
//   * the filters never touch LDS: fragment-major (st_pack_conv_weight_frag, KH = KW = 3), through a register ring;
//   * epilogue as conv1x1_kstream_kernel's: statistics through LDS as full-wave atomics | scale / shift / ReLU, 16-byte stores.
#include "common.h"
#include "prof.h"

namespace {

struct S2Args {
  const bf16_t* x; const bf16_t* w; bf16_t* y;
  float* stats; int srep;
  const float* scale; const float* shift; int relu;
  const float* in_stats; const float* in_gamma; const float* in_beta; float in_count, in_eps; int in_srep;
  int M, N, nbn, Hin, Win, Ho, Wo;
};

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

// C = 128 (56 x 56 -> 28 x 28: 896 workgroups) and C = 256 (28 x 28 -> 14 x 14: 224 row blocks x two 128-channel halves): two workgroups per CU
// (65 KB of LDS; the register allocation is cut to 256 per wave, ring of 4 K-steps) -- 3.5 rounds of 256 become 1.75 of 512 at C = 128, and two
// waves per SIMD fill each other's gather and filter waits
template <int C, int NTW, bool XF, bool AFFINE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(C <= 256 ? 2 : 1, C <= 256 ? 2 : 1))) void conv3x3s2_kstream_kernel(S2Args a) {
  constexpr int TM = 7, BM = 16 * TM;
  constexpr int SLAB = 128, PIX = 2 * SLAB + 32, KSS = SLAB / 32;      // 4 K-steps per slab
  constexpr int SPT = C / SLAB;                                        // slabs per tap
  constexpr int NSLAB = 9 * SPT, KS = 9 * C / 32;
  constexpr int SLAB_BYTES = BM * PIX;
  constexpr int NL = BM * (SLAB / 8) / 256;                            // 7 16-byte chunks per thread per slab
  constexpr int WR = NTW == 4 ? 6 : (C <= 256 ? 4 : 8);                // filter ring (K-steps in flight)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* coef = reinterpret_cast<float*>(smem + 2 * SLAB_BYTES);       // XF: [scale(C) | shift(C)]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lid;
  {
    const int nblk = gridDim.x, id = blockIdx.x, xcd = id & 7, q = nblk >> 3, r = nblk & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int nb = lid % a.nbn, bm = lid / a.nbn;
  const int r16 = lane & 15, q4 = lane >> 4;

  // the producer's statistics head the load queue (XF: C / 256 channels per thread, at least one)
  constexpr int NSH = C > 256 ? C / 256 : 1;
  StatHead shd[NSH];
  if constexpr (XF) {
#pragma unroll
    for (int k = 0; k < NSH; ++k) { const int c = (tid + 256 * k) % C; stat_head_issue(shd[k], a.in_stats, a.in_srep, C, c, a.in_gamma, a.in_beta); }
  }

  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.w), 0, 0x7fffffff, 0x00020000);
  // fragment (T, ks) at ((T * KS + ks) * 64 + lane) * 16 bytes; this wave's tiles (nb * 4 + wid) * NTW + j
  const int voffw = (((nb * 4 + wid) * NTW) * KS * 64 + lane) * 16;
  auto wfrag = [&](int ks, int j) { return __builtin_amdgcn_raw_buffer_load_b128(rs_w, voffw, (j * KS + ks) * 1024, 0); };

  // ---- loader: thread -> (16-byte chunk cch of the slab's 128 channels, rows lrow + 16 i); per row: the pixel of tap (1, 1) (always inside
  // the image) and a 9-bit mask of the taps that fall inside
  const int cch = tid & 15, lrow = tid >> 4;
  int pix0[NL]; unsigned tmask[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    int m = bm * BM + lrow + 16 * i;
    m = m < a.M ? m : a.M - 1;
    const int hw = a.Ho * a.Wo, b = m / hw, rem = m - b * hw, ho = rem / a.Wo, wo = rem - ho * a.Wo;
    pix0[i] = (b * a.Hin + 2 * ho) * a.Win + 2 * wo;
    unsigned mk = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int hi = 2 * ho + t / 3 - 1, wi = 2 * wo + t % 3 - 1;
      if ((unsigned)hi < (unsigned)a.Hin && (unsigned)wi < (unsigned)a.Win) mk |= 1u << t;
    }
    tmask[i] = mk;
  }
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.x), 0, 0x7fffffff, 0x00020000);
  u32x4 ra[2][NL];
  auto gload = [&](u32x4 (&r)[NL], int slab) {
    const int sl = slab < NSLAB ? slab : NSLAB - 1;                   // past the end: re-read the last slab (counted waits stay valid)
    const int tap = sl / SPT, cs = sl - tap * SPT;
    const int doff = (tap / 3 - 1) * a.Win + (tap % 3 - 1);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const bool ok = (tmask[i] >> tap) & 1u;
      const int p = pix0[i] + (ok ? doff : 0);                        // an outside tap reads the centre pixel, zeroed at the LDS store
      r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (p * C + cch * 8) * 2, cs * SLAB * 2, 0);
    }
  };
  auto lstore = [&](u32x4 (&r)[NL], int buf, int slab) {
    const int tap = slab / SPT, cs = slab - tap * SPT;
    float sc[8], sh[8];
    if constexpr (XF) {
      const float* cp = coef + cs * SLAB + cch * 8;
#pragma unroll
      for (int e = 0; e < 8; e += 4) {
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(cp + e), h4 = *reinterpret_cast<const f32x4*>(cp + C + e);
#pragma unroll
        for (int q = 0; q < 4; ++q) { sc[e + q] = s4[q]; sh[e + q] = h4[q]; }
      }
    }
    char* base = smem + buf * SLAB_BYTES + lrow * PIX + cch * 16;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      u32x4 v = r[i];
      if constexpr (XF) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const float lo = fmaxf(__uint_as_float(v[d] << 16) * sc[2 * d] + sh[2 * d], 0.f);
          const float hi = fmaxf(__uint_as_float(v[d] & 0xffff0000u) * sc[2 * d + 1] + sh[2 * d + 1], 0.f);
          v[d] = pack_bf16x2(lo, hi);
        }
      }
      if (!((tmask[i] >> tap) & 1u)) v = u32x4{0u, 0u, 0u, 0u};       // padding is zero AFTER the producer's BatchNorm + ReLU
      *reinterpret_cast<u32x4*>(base + 16 * i * PIX) = v;
    }
  };

  u32x4 wq[WR][NTW];
  gload(ra[0], 0);
#pragma unroll
  for (int s = 0; s < WR; ++s)
#pragma unroll
    for (int j = 0; j < NTW; ++j) wq[s][j] = wfrag(s, j);
  gload(ra[1], 1);
  if constexpr (XF) {
    const float inv = 1.0f / a.in_count;
#pragma unroll
    for (int k = 0; k < NSH; ++k) {
      const int c = tid + 256 * k;
      if (c < C) stat_head_finish(shd[k], a.in_stats, a.in_srep, C, c, inv, a.in_eps, coef[c], coef[C + c]);
    }
    __syncthreads();
  }
  lstore(ra[0], 0, 0);
  gload(ra[0], 2);
  __syncthreads();

  f32x4 acc[TM][NTW];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const char* abase = smem + r16 * PIX + q4 * 16;

  u32x4 fa0[TM], fa1[TM];
#pragma clang loop unroll(full)
  for (int s = 0; s < NSLAB; ++s) {
    const char* ab = abase + (s & 1) * SLAB_BYTES;
    auto read_a = [&](u32x4 (&f)[TM], int kk) {
#pragma unroll
      for (int i = 0; i < TM; ++i) f[i] = *reinterpret_cast<const u32x4*>(ab + i * 16 * PIX + kk * 64);
    };
    read_a(fa0, 0);
#pragma clang loop unroll(full)
    for (int kk = 0; kk < KSS; ++kk) {
      const int ks = s * KSS + kk;
      u32x4 (&fa)[TM] = (kk & 1) ? fa1 : fa0;
      u32x4 (&fn)[TM] = (kk & 1) ? fa0 : fa1;
      if (kk + 1 < KSS) read_a(fn, kk + 1);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = mfma_bf16(wq[ks % WR][j], fa[i], acc[i][j]);
      if (ks + WR < KS) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) wq[ks % WR][j] = wfrag(ks + WR, j);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (s + 1 < NSLAB) {
      lstore(ra[(s + 1) & 1], (s + 1) & 1, s + 1);       // slab s+1 -> the ring half slab s-1 was read from
      gload(ra[(s + 1) & 1], s + 3);
    }
    __syncthreads();
  }

  // ---- epilogue ------------------------------------------------------------------------------------------------------------------
  constexpr int NC = 4 * NTW;
  const int cb = ((nb * 4 + wid) * NTW) * 16 + NC * q4;
  float es[NC], ess[NC], scv[NC], shv[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) { es[c] = 0.f; ess[c] = 0.f; scv[c] = 1.f; shv[c] = 0.f; }
  if constexpr (AFFINE) {
#pragma unroll
    for (int c = 0; c < NC; ++c) { scv[c] = a.scale[cb + c]; shv[c] = a.shift[cb + c]; }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = bm * BM + i * 16 + r16;
    if (m < a.M) {
      float v[NC];
#pragma unroll
      for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * j + e] = acc[i][j][e];
#pragma unroll
      for (int c = 0; c < NC; ++c) { es[c] += v[c]; ess[c] += v[c] * v[c]; }
      if constexpr (AFFINE) {
#pragma unroll
        for (int c = 0; c < NC; ++c) v[c] = v[c] * scv[c] + shv[c];
        if (a.relu) {
#pragma unroll
          for (int c = 0; c < NC; ++c) v[c] = fmaxf(v[c], 0.f);
        }
      }
      const long dst = ((long)m * a.N + cb) * 2;
      if constexpr (NTW == 1) st_out_store8(a.y, dst, u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])});
      else {
#pragma unroll
        for (int h = 0; h < NTW / 2; ++h)
          st_out_store16(a.y, dst + 16 * h, u32x4{pack_bf16x2(v[8 * h], v[8 * h + 1]), pack_bf16x2(v[8 * h + 2], v[8 * h + 3]),
                                                  pack_bf16x2(v[8 * h + 4], v[8 * h + 5]), pack_bf16x2(v[8 * h + 6], v[8 * h + 7])});
      }
    }
  }
  if (a.stats) {
    float* sdst = a.stats + (a.srep > 1 ? (size_t)(bm % a.srep) * 2 * a.N : 0);
#pragma unroll
    for (int c = 0; c < NC; ++c) { es[c] = row16_sum_(es[c]); ess[c] = row16_sum_(ess[c]); }
    constexpr int BNLOC = 64 * NTW;
    float* sred = reinterpret_cast<float*>(smem);
    const int local = wid * 16 * NTW + NC * q4;
    __syncthreads();
    if (r16 == 0) {
#pragma unroll
      for (int c = 0; c < NC; ++c) { sred[local + c] = es[c]; sred[BNLOC + local + c] = ess[c]; }
    }
    __syncthreads();
    const int chan0 = nb * BNLOC;
    for (int t = tid; t < 2 * BNLOC; t += 256) atomicAdd(sdst + (t < BNLOC ? chan0 + t : a.N + chan0 + t - BNLOC), sred[t]);
  }
}

template <int C, int NTW, bool XF, bool AFFINE>
int launch_s2__(S2Args& a, hipStream_t st, double flops) {
  constexpr int lds = 2 * 112 * (2 * 128 + 32) + 2 * C * 4;
  static int attr_set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3s2_kstream_kernel<C, NTW, XF, AFFINE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set[dev] = 1;
  }
  a.nbn = a.N / (64 * NTW);
  StProfScope prof(23, flops, st);
  hipLaunchKernelGGL((conv3x3s2_kstream_kernel<C, NTW, XF, AFFINE>), dim3(((a.M + 111) / 112) * a.nbn), dim3(256), lds, st, a);
  prof.end(st);
  ST_LAUNCH_CHECK();
  return 0;
}
template <int C, int NTW>
int launch_s2(S2Args& a, hipStream_t st, double flops) {
  if (a.in_stats) return a.scale ? launch_s2__<C, NTW, true, true>(a, st, flops) : launch_s2__<C, NTW, true, false>(a, st, flops);
  return a.scale ? launch_s2__<C, NTW, false, true>(a, st, flops) : launch_s2__<C, NTW, false, false>(a, st, flops);
}

}  // namespace

// > 0: supported, the value is the `ntw` of the fragment-major weights (st_pack_conv_weight_frag with KH = KW = 3); 0: use st_conv
extern "C" int st_conv3x3_s2_supported(int C, int N) {
  if (C == 128 && N == 128) return 2;
  if (C == 256 && N == 256) return 2;
  if (C == 512 && N == 512) return 2;
  return 0;
}

extern "C" int st_conv3x3_s2(const st_conv3x3_img_desc* d, void* stream) {
  ST_CHECK(d && d->x && d->w_frag && d->y, "st_conv3x3_s2: null pointer");
  ST_CHECK(st_conv3x3_s2_supported(d->C, d->N), "st_conv3x3_s2: unsupported geometry C=%d N=%d", d->C, d->N);
  ST_CHECK(d->B > 0 && d->H >= 1 && d->W >= 1 && (long)d->B * d->H * d->W * d->C * 2 < (1L << 31), "st_conv3x3_s2: bad geometry");
  ST_CHECK((d->scale == nullptr) == (d->shift == nullptr) && (d->scale || !d->relu), "st_conv3x3_s2: scale, shift (and relu) go together");
  ST_CHECK(!d->in_stats || (d->in_gamma && d->in_beta && d->in_count > 0.f), "st_conv3x3_s2: input transform needs gamma, beta, count");
  ST_CHECK(d->stats_replicas >= 0 && d->stats_replicas <= 1024 && d->in_stats_replicas >= 0 && d->in_stats_replicas <= 1024, "st_conv3x3_s2: bad stats_replicas");
  S2Args a{};
  a.x = reinterpret_cast<const bf16_t*>(d->x); a.w = reinterpret_cast<const bf16_t*>(d->w_frag); a.y = reinterpret_cast<bf16_t*>(d->y);
  a.stats = d->stats; a.srep = d->stats_replicas; a.scale = d->scale; a.shift = d->shift; a.relu = d->relu;
  a.in_stats = d->in_stats; a.in_gamma = d->in_gamma; a.in_beta = d->in_beta; a.in_count = d->in_count; a.in_eps = d->in_eps;
  a.in_srep = d->in_stats_replicas > 1 ? d->in_stats_replicas : 1;
  a.Hin = d->H; a.Win = d->W; a.Ho = (d->H - 1) / 2 + 1; a.Wo = (d->W - 1) / 2 + 1;
  a.M = d->B * a.Ho * a.Wo; a.N = d->N;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const double flops = 2.0 * (double)a.M * d->N * 9.0 * d->C;
  if (d->C == 128) return launch_s2<128, 2>(a, st, flops);
  if (d->C == 256) return launch_s2<256, 2>(a, st, flops);
  return launch_s2<512, 2>(a, st, flops);
}
