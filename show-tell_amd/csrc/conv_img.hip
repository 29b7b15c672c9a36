// Image-resident convolutions for gfx950 (MI355X): the 3x3 stride-1 layers of torchvision's ResNet bottlenecks
// (reference cnn.py:46 / cnn_attn.py:46, `self.model(x)`), bf16 storage, fp32 accumulation on v_mfma_f32_16x16x32_bf16.
//
// Why a second convolution kernel beside igemm.hip.  A 128x128 implicit-GEMM tile moves 32 KB through the CU's L2->LDS
// path per 512 MFMA cycles and re-stages every input pixel once per filter tap; on the 14x14 / 28x28 layers that path,
// not the matrix pipe, sets the time.  Here a workgroup owns a BAND of one image:
//   * the band's input pixels (+ a zero halo) are read from HBM/L2 ONCE, normalised on the way in (the producer's
//     BatchNorm + ReLU: `relu(x * scale + shift)` from the producer's batch statistics -- once per element, not once
//     per tap) and stay in LDS for all nine taps: a tap is a constant shift of the LDS address;
//   * pixel rows are padded to 2C + 32 bytes, which makes the ds_read_b128 of a 16-pixel MFMA operand conflict-free
//     for every tap shift without any per-lane swizzle arithmetic (every read is base register + immediate);
//   * the filter bank never touches LDS: it is stored fragment-major in HBM (st_pack_conv_weight_frag), so a wave's
//     operand for 16 output channels x 32 K is ONE coalesced 1-KiB global_load_dwordx4, prefetched one tap ahead;
//   * the K loop has NO barrier: four waves (one per SIMD, 32 output channels each) run independently from the
//     single barrier after the fill to the end;
//   * output channels are permuted inside a wave's fragments so that a lane ends up with 8 CONSECUTIVE channels of one
//     pixel: the epilogue is straight 16-byte stores from the accumulators (no LDS staging), BatchNorm statistics are
//     a DPP row reduction + one atomic per channel per workgroup.
// M enumeration: output positions are numbered over the PADDED row pitch (m' = row * (W + 2) + col); columns W, W+1 of
// each row are computed and discarded (12.5 % at 14x14, 6.7 % at 28x28, 3.4 % at 56x56) -- that is what buys the
// constant tap shift.
#include "common.h"
#include "prof.h"

namespace {

struct ImgArgs {
  const bf16_t* x; const bf16_t* w; bf16_t* y;
  float* stats; int srep;
  const float* scale; const float* shift; int relu;                                             // eval-mode epilogue
  const float* in_stats; const float* in_gamma; const float* in_beta; float in_count, in_eps; int in_srep;   // producer's BatchNorm
  int B, H, W, N, Wp, R, bands, nbn;
  unsigned long long* stamps;   // debug (tools/conv_stamps.py): per-wave s_memtime at the phase boundaries, normally NULL
};

template <int CTRL> __device__ __forceinline__ float dpp_rot_(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum_(float v) {   // sum over the 16 lanes with equal lane >> 4; every lane gets it
  v += dpp_rot_<0x128>(v); v += dpp_rot_<0x124>(v); v += dpp_rot_<0x122>(v); v += dpp_rot_<0x121>(v);
  return v;
}
__device__ __forceinline__ f32x4 mfma_bf16(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
}

// C input = output-side channel count of the fill (input channels); TM 16-position tiles per band; NTW 16-channel tiles per wave
template <int C, int TM, int NTW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_img_kernel(ImgArgs a) {
  constexpr int PIX = 2 * C + 32;            // bytes per LDS pixel row
  constexpr int CS = C / 32;                 // k-steps per filter tap
  constexpr int KS = 9 * CS;                 // k-steps in all
  constexpr int CH8 = C / 8;                 // 16-byte chunks per pixel
  constexpr int PSTEP = 256 / CH8;           // pixels per fill pass of the workgroup
  constexpr bool ALLW = KS * NTW * 4 <= 96;  // the whole filter slice of a wave fits in registers (C = 64)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lid;
  {   // blocks that share an XCD (equal blockIdx % 8) take neighbouring work items: the N blocks of one band meet in one L2
    const int nblk = gridDim.x, id = blockIdx.x, xcd = id & 7, q = nblk >> 3, r = nblk & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int nb = lid % a.nbn;
  const int bi = lid / a.nbn;
  const int band = bi % a.bands, img = bi / a.bands;
  const int r0 = band * a.R;
  const int rows = a.H - r0 < a.R ? a.H - r0 : a.R;
  const int Wp = a.Wp;

#define IMG_STAMP(i) do { if (a.stamps && lane == 0) a.stamps[((size_t)blockIdx.x * 4 + wid) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
  IMG_STAMP(0);
  // the first filter fragments are requested before the fill: they land while the band is staged
  const int r16 = lane & 15, q4 = lane >> 4;
  const int T0 = (nb * 4 + wid) * NTW;                              // first 16-channel tile of this wave
  const u32x4* wl = reinterpret_cast<const u32x4*>(a.w) + lane;     // fragment (T, ks) = wl[(T * KS + ks) * 64]
  constexpr int WQ = ALLW ? KS : CS;
  u32x4 wq[WQ][NTW];
#pragma unroll
  for (int s = 0; s < WQ; ++s)
#pragma unroll
    for (int j = 0; j < NTW; ++j) wq[s][j] = wl[((size_t)(T0 + j) * KS + s) * 64];

  // ---- fill: padded band [rows + 2][Wp] (+ everything a discarded position may touch, zeroed) ---------------------
  // Every global load of the band is in flight before the first LDS write; the producer's BatchNorm coefficients are
  // derived meanwhile (replicated statistics are summed here: no separate reduction launch).
  {
    const int cch = tid % CH8;
    float sc[8], sh[8];
    const bool xf = a.in_stats != nullptr;
    const int total = 16 * TM + 2 * Wp + 2;                        // LDS pixels
    float* coef = reinterpret_cast<float*>(smem + (size_t)total * PIX);   // [scale(C) | shift(C)] behind the band
    const bf16_t* ximg = a.x + (size_t)img * a.H * a.W * C + cch * 8;
    int pp = tid / CH8;
    int pr = pp / Wp, pc = pp - pr * Wp;
    constexpr int U = C == 256 ? 33 : C == 128 ? 18 : C == 64 ? 12 : 21;   // one round for the bands of a 224 x 224 image
    bool first = true;
    for (; pp < total; pp += U * PSTEP) {
      u32x4 v[U]; bool ok[U]; int dst[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int hi = r0 - 1 + pr, wi = pc - 1;
        ok[u] = pr < rows + 2 && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
        dst[u] = pp + u * PSTEP < total ? (pp + u * PSTEP) * PIX + cch * 16 : -1;
        v[u] = u32x4{0u, 0u, 0u, 0u};
        if (ok[u]) v[u] = *reinterpret_cast<const u32x4*>(ximg + (size_t)(hi * a.W + wi) * C);
        pc += PSTEP;
        while (pc >= Wp) { pc -= Wp; ++pr; }
      }
      if (xf && first) {
        const float inv = 1.0f / a.in_count;
        for (int c = tid; c < C; c += 256) {
          float sm = 0.f, sq = 0.f;
          for (int r = 0; r < a.in_srep; ++r) { sm += a.in_stats[(size_t)r * 2 * C + c]; sq += a.in_stats[(size_t)r * 2 * C + C + c]; }
          const float mean = sm * inv;
          const float var = fmaxf(sq * inv - mean * mean, 0.f);
          const float scv = a.in_gamma[c] * rsqrtf(var + a.in_eps);
          coef[c] = scv; coef[C + c] = a.in_beta[c] - mean * scv;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = coef[cch * 8 + e]; sh[e] = coef[C + cch * 8 + e]; }
        first = false;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (xf && ok[u]) {
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const float lo = fmaxf(__uint_as_float(v[u][d] << 16) * sc[2 * d] + sh[2 * d], 0.f);
            const float hi = fmaxf(__uint_as_float(v[u][d] & 0xffff0000u) * sc[2 * d + 1] + sh[2 * d + 1], 0.f);
            v[u][d] = pack_bf16x2(lo, hi);
          }
        }
        if (dst[u] >= 0) *reinterpret_cast<u32x4*>(smem + dst[u]) = v[u];
      }
    }
  }
  IMG_STAMP(1);
  __syncthreads();
  IMG_STAMP(2);

  // ---- K loop: no barrier from here on ----------------------------------------------------------------------------
  f32x4 acc[TM][NTW];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Straight-line K loop (fully unrolled: no loop-carried register shuffling), software-pipelined in source order and
  // pinned with scheduling barriers: step s issues the LDS reads of step s + 1, runs its TM x NTW MFMAs, then re-requests
  // its filter registers for step s + WQ.  The waits the compiler inserts are then counted (lgkmcnt(TM), vmcnt((WQ-1) NTW)).
  constexpr int TH = TM > 7 ? 7 : TM;                               // tiles addressed from the first base register (16-bit ds offsets)
  const char* abase = smem + r16 * PIX + q4 * 16;
  auto read_a = [&](u32x4 (&fa)[TM], int tap, int cs) {
    const char* ab = abase + ((tap / 3) * Wp + tap % 3) * PIX;
    const char* ab2 = ab + TH * 16 * PIX;
#pragma unroll
    for (int i = 0; i < TM; ++i)
      fa[i] = i < TH ? *reinterpret_cast<const u32x4*>(ab + i * 16 * PIX + cs * 64) : *reinterpret_cast<const u32x4*>(ab2 + (i - TH) * 16 * PIX + cs * 64);
  };
  u32x4 fa0[TM], fa1[TM];
  read_a(fa0, 0, 0);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    u32x4 (&cur)[TM] = (s & 1) ? fa1 : fa0;
    u32x4 (&nxt)[TM] = (s & 1) ? fa0 : fa1;
    if (s + 1 < KS) read_a(nxt, (s + 1) / CS, (s + 1) % CS);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[i][j] = mfma_bf16(wq[s % WQ][j], cur[i], acc[i][j]);
    if (s + WQ < KS) {
#pragma unroll
      for (int j = 0; j < NTW; ++j) wq[s % WQ][j] = wl[((size_t)(T0 + j) * KS + s + WQ) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  IMG_STAMP(3);
  // ---- epilogue: accumulators -> (statistics) -> (scale/shift, ReLU) -> bf16 -> 8 * NTW-byte stores ----------------
  constexpr int NC = 4 * NTW;                                        // consecutive channels per lane
  const int cb = T0 * 16 + NC * q4;                                  // tile j, register e <-> channel cb + 4 j + e
  float es[NC], ess[NC], scv[NC], shv[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) { es[c] = 0.f; ess[c] = 0.f; scv[c] = 1.f; shv[c] = 0.f; }
  const bool affine = a.scale != nullptr;
  if (affine) {
#pragma unroll
    for (int c = 0; c < NC; ++c) { scv[c] = a.scale[cb + c]; shv[c] = a.shift[cb + c]; }
  }
  int ho = r0, wo = r16;
  while (wo >= Wp) { wo -= Wp; ++ho; }
  bf16_t* yimg = a.y + (size_t)img * a.H * a.W * a.N + cb;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const bool valid = wo < a.W && ho < r0 + rows;
    if (valid) {
      float v[NC];
#pragma unroll
      for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * j + e] = acc[i][j][e];
#pragma unroll
      for (int c = 0; c < NC; ++c) { es[c] += v[c]; ess[c] += v[c] * v[c]; }
      if (affine) {
#pragma unroll
        for (int c = 0; c < NC; ++c) v[c] = v[c] * scv[c] + shv[c];
      }
      if (a.relu) {
#pragma unroll
        for (int c = 0; c < NC; ++c) v[c] = fmaxf(v[c], 0.f);
      }
      bf16_t* dst = yimg + (size_t)(ho * a.W + wo) * a.N;
      if constexpr (NTW == 1) {
        *reinterpret_cast<u32x2*>(dst) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      } else {
#pragma unroll
        for (int h = 0; h < NTW / 2; ++h)
          *reinterpret_cast<u32x4*>(dst + 8 * h) = u32x4{pack_bf16x2(v[8 * h], v[8 * h + 1]), pack_bf16x2(v[8 * h + 2], v[8 * h + 3]),
                                                         pack_bf16x2(v[8 * h + 4], v[8 * h + 5]), pack_bf16x2(v[8 * h + 6], v[8 * h + 7])};
      }
    }
    wo += 16;
    while (wo >= Wp) { wo -= Wp; ++ho; }
  }
  if (a.stats) {
    float* sdst = a.stats + (a.srep > 1 ? (size_t)(bi % a.srep) * 2 * a.N : 0);
#pragma unroll
    for (int c = 0; c < NC; ++c) { es[c] = row16_sum_(es[c]); ess[c] = row16_sum_(ess[c]); }
    if (r16 == 0) {
#pragma unroll
      for (int c = 0; c < NC; ++c) { atomicAdd(sdst + cb + c, es[c]); atomicAdd(sdst + a.N + cb + c, ess[c]); }
    }
  }
  IMG_STAMP(4);
#undef IMG_STAMP
}

// ---- fragment-major filter bank ---------------------------------------------------------------------------------
// out[((T * KS + ks) * 64 + lane) * 8 + j], T = 16-channel tile, ks = 32-deep K step (k = tap * Cin + c), lane = (q4, r16):
// MFMA row r16 of tile T is output channel ch(T, r16) = (T / NTW) * 16 NTW + 4 NTW (r16 / 4) + 4 (T % NTW) + r16 % 4
// (so that a lane's accumulators of a wave's NTW tiles are 4 NTW consecutive channels), k = 32 ks + 8 q4 + j.
__global__ void pack_frag_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout, int Cin, int KH, int KW, int ntw) {
  const long total = (long)Cout * KH * KW * Cin;
  const int KS = KH * KW * Cin / 32;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int j = idx & 7, lane = (idx >> 3) & 63;
    const long f = idx >> 9;
    const int ks = (int)(f % KS), T = (int)(f / KS);
    const int r16 = lane & 15, q4 = lane >> 4;
    const int ch = (T / ntw) * 16 * ntw + 4 * ntw * (r16 >> 2) + 4 * (T % ntw) + (r16 & 3);
    const int k = 32 * ks + 8 * q4 + j;
    const int tap = k / Cin, c = k - tap * Cin;
    const int kh = tap / KW, kw = tap - kh * KW;
    out[idx] = (bf16_t)w[(((long)ch * Cin + c) * KH + kh) * KW + kw];
  }
}

struct ImgCfg { int tm, ntw; };
// the instantiations: (C, TM, NTW)
inline bool img_cfg(int C, int N, ImgCfg* c) {
  switch (C) {
    case 64:  *c = ImgCfg{15, 1}; return N % 64 == 0;
    case 128: *c = ImgCfg{14, 2}; return N % 128 == 0;
    case 256: *c = ImgCfg{14, 2}; return N % 128 == 0;
    case 512: *c = ImgCfg{4, 2};  return N % 128 == 0;
  }
  return false;
}

template <int C, int TM, int NTW>
int launch_img(const ImgArgs& a, int lds, hipStream_t st, double flops) {
  static int attr_set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && attr_set[dev] < lds) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_img_kernel<C, TM, NTW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set[dev] = 160 * 1024;
  }
  StProfScope prof(8, flops, st);
  hipLaunchKernelGGL((conv3x3_img_kernel<C, TM, NTW>), dim3(a.B * a.bands * a.nbn), dim3(256), lds, st, a);
  prof.end(st);
  ST_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// > 0: the NTW (16-channel tiles per wave) of the fragment-major weight layout this geometry runs with; 0: not supported
extern "C" int st_conv3x3_img_supported(int H, int W, int C, int N) {
  ImgCfg c;
  if (!img_cfg(C, N, &c)) return 0;
  const int Wp = W + 2;
  if (H < 1 || W < 1 || 16 * c.tm < Wp) return 0;                  // at least one output row per band
  const long lds = (long)(16 * c.tm + 2 * Wp + 2) * (2 * C + 32) + 2 * C * (long)sizeof(float);
  return lds <= 160 * 1024 ? c.ntw : 0;
}

extern "C" int st_conv3x3_img(const st_conv3x3_img_desc* d, void* stream) {
  ST_CHECK(d && d->x && d->w_frag && d->y, "st_conv3x3_img: null pointer");
  ImgCfg c;
  ST_CHECK(st_conv3x3_img_supported(d->H, d->W, d->C, d->N) > 0 && img_cfg(d->C, d->N, &c),
           "st_conv3x3_img: unsupported geometry H=%d W=%d C=%d N=%d", d->H, d->W, d->C, d->N);
  ST_CHECK(d->B > 0 && (long)d->B * d->H * d->W * (d->C > d->N ? d->C : d->N) < (1L << 31), "st_conv3x3_img: bad batch");
  ST_CHECK((d->scale == nullptr) == (d->shift == nullptr), "st_conv3x3_img: scale and shift go together");
  ST_CHECK(!d->in_stats || (d->in_gamma && d->in_beta && d->in_count > 0.f), "st_conv3x3_img: input transform needs gamma, beta, count");
  ST_CHECK(d->stats_replicas >= 0 && d->stats_replicas <= 1024 && d->in_stats_replicas >= 0 && d->in_stats_replicas <= 1024, "st_conv3x3_img: bad stats_replicas");
  ImgArgs a;
  a.x = reinterpret_cast<const bf16_t*>(d->x); a.w = reinterpret_cast<const bf16_t*>(d->w_frag); a.y = reinterpret_cast<bf16_t*>(d->y);
  a.stats = d->stats; a.srep = d->stats_replicas; a.scale = d->scale; a.shift = d->shift; a.relu = d->relu;
  a.in_stats = d->in_stats; a.in_gamma = d->in_gamma; a.in_beta = d->in_beta; a.in_count = d->in_count; a.in_eps = d->in_eps; a.in_srep = d->in_stats_replicas > 1 ? d->in_stats_replicas : 1;
  a.B = d->B; a.H = d->H; a.W = d->W; a.N = d->N; a.Wp = d->W + 2;
  a.R = 16 * c.tm / a.Wp; if (a.R > d->H) a.R = d->H;
  a.bands = (d->H + a.R - 1) / a.R;
  a.nbn = d->N / (64 * c.ntw);
  a.stamps = st_debug_stamps_ptr();
  const int lds = (16 * c.tm + 2 * a.Wp + 2) * (2 * d->C + 32) + 2 * d->C * (int)sizeof(float);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const double flops = 2.0 * d->B * d->H * d->W * (double)d->N * 9.0 * d->C;
  switch (d->C) {
    case 64:  return launch_img<64, 15, 1>(a, lds, st, flops);
    case 128: return launch_img<128, 14, 2>(a, lds, st, flops);
    case 256: return launch_img<256, 14, 2>(a, lds, st, flops);
    case 512: return launch_img<512, 4, 2>(a, lds, st, flops);
  }
  st_set_error("st_conv3x3_img: no kernel for C=%d", d->C);
  return 1;
}

extern "C" int st_pack_conv_weight_frag(const float* w, void* out, int Cout, int Cin, int KH, int KW, int ntw, void* stream) {
  ST_CHECK(w && out, "st_pack_conv_weight_frag: null pointer");
  ST_CHECK(ntw >= 1 && ntw <= 16 && Cout % (16 * ntw) == 0 && (KH * KW * Cin) % 32 == 0 && Cin % 8 == 0,
           "st_pack_conv_weight_frag: Cout=%d must be a multiple of %d and Cin=%d of 8 (K %% 32 == 0)", Cout, 16 * ntw, Cin);
  const long total = (long)Cout * KH * KW * Cin;
  long grid = (total + 255) / 256; if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(pack_frag_kernel, dim3((unsigned)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w,
                     reinterpret_cast<bf16_t*>(out), Cout, Cin, KH, KW, ntw);
  ST_LAUNCH_CHECK();
  return 0;
}
