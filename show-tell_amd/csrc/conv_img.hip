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
#include <stdlib.h>
#include <stdio.h>

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

// Per-channel statistics of a workgroup -> global replicas.  A float atomic costs the CU ~50 ns per wave-INSTRUCTION whatever its
// active lanes (MI355X_MICROARCH, global float atomics): four lanes x 16-32 instructions per wave was 3-6 us of epilogue.
// The per-wave partials meet in LDS instead and leave as full 64-lane instructions over consecutive channels.
// es / ess: this lane's NC channel sums (already reduced over the 16 pixel lanes); local = channel index inside the workgroup.
template <int NC, int BNLOC>
__device__ __forceinline__ void block_stats_flush(const float (&es)[NC], const float (&ess)[NC], bool writer, int local, char* smem,
                                                  float* sdst, int chan0, int N, int tid) {
  float* sred = reinterpret_cast<float*>(smem);                     // [2][BNLOC]; the K loop's LDS image is dead
  __syncthreads();
  if (writer) {
#pragma unroll
    for (int c = 0; c < NC; ++c) { sred[local + c] = es[c]; sred[BNLOC + local + c] = ess[c]; }
  }
  __syncthreads();
  for (int t = tid; t < 2 * BNLOC; t += 256) atomicAdd(sdst + (t < BNLOC ? chan0 + t : N + chan0 + t - BNLOC), sred[t]);
}

__device__ __forceinline__ void bn_relu_chunk(u32x4& v, const float* sc, const float* sh);

// C input = output-side channel count of the fill (input channels); TM 16-position tiles per band; NTW 16-channel tiles per wave
// HALVES = 2 (C = 256): the band is staged in two channel halves and K runs (half, tap, channel) -- half 0 is filled before the
// K loop, half 1's transform + LDS writes ride one 16-byte chunk per K-step under the first half's MFMAs (its loads were issued
// with half 0's), one barrier between the halves: 4 us of the 7.8 us fill leave the critical path.  The fragment-major filters
// carry the same K order (layout code of st_conv3x3_img_supported).
template <int C, int TM, int NTW, int HALVES, bool AFFINE, int MS = 1>
__device__ __forceinline__ void conv3x3_img_body(const ImgArgs& a) {
  // MS = 2 (C = 64): the four waves are 2 (position halves) x 2 (channel halves).  With one 16-channel tile per wave (NTW = 1) every
  // ds_read_b128 of a 16-position operand fed ONE MFMA and the kernel ran at the LDS read rate (1080 KB per workgroup at 128 B/clk =
  // twice its MFMA time); two tiles per wave over half the positions halve that traffic.
  constexpr int TMW = TM / MS;                // position tiles per wave
  static_assert(TM % MS == 0 && (MS == 1 || MS == 2) && (MS == 1 || HALVES == 1), "position split");
  constexpr int PIX = 2 * C + 32;            // bytes per LDS pixel row
  constexpr int CS = C / 32;                 // k-steps per filter tap
  constexpr int CSH = CS / HALVES;           // ... per channel half
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
  // the producer's BatchNorm statistics of this thread's channel(s) head the load queue (stat_head_issue, common.h): C / 256 channels per thread
  constexpr int NSH = C > 256 ? C / 256 : 1;
  StatHead shd[NSH];
  if (a.in_stats) {
#pragma unroll
    for (int k = 0; k < NSH; ++k) { const int c = (tid + 256 * k) % C; stat_head_issue(shd[k], a.in_stats, a.in_srep, C, c, a.in_gamma, a.in_beta); }
  }
  // the first filter fragments are requested before the fill: they land while the band is staged
  const int r16 = lane & 15, q4 = lane >> 4;
  const int wn = MS == 1 ? wid : (wid & 1), wm = MS == 1 ? 0 : (wid >> 1);
  const int T0 = (nb * (4 / MS) + wn) * NTW;                        // first 16-channel tile of this wave
  const u32x4* wl = reinterpret_cast<const u32x4*>(a.w) + lane;     // fragment (T, ks) = wl[(T * KS + ks) * 64]
  constexpr int WQ = ALLW ? KS : CS;                               // (HALVES = 2: two taps of lead)
  u32x4 wq[WQ][NTW];
#pragma unroll
  for (int s = 0; s < WQ; ++s)
#pragma unroll
    for (int j = 0; j < NTW; ++j) wq[s][j] = wl[((size_t)(T0 + j) * KS + s) * 64];

  // ---- fill: padded band [rows + 2][Wp] (+ everything a discarded position may touch, zeroed) ---------------------
  // Every global load of the band is in flight before the first LDS write; the producer's BatchNorm coefficients are
  // derived meanwhile (replicated statistics are summed here: no separate reduction launch).
  constexpr int CHH = CH8 / 2, U2 = TM >= 14 ? 19 : TM + 3;        // HALVES == 2: chunks per pixel per half; passes of 16 pixels (<= 297 | 16 TM + 48 LDS pixels)
  u32x4 v1[HALVES == 2 ? U2 : 1]; bool ok1[HALVES == 2 ? U2 : 1];  // second channel half: loaded now, written under the first half's MFMAs
  float sc1[8], sh1[8];
  const int cch2 = tid % CHH, pp2 = tid / CHH;
  const bool xf_ = a.in_stats != nullptr;
  if constexpr (HALVES == 2) {
    const int total = 16 * TM + 2 * Wp + 2;
    float* coef = reinterpret_cast<float*>(smem + (size_t)total * PIX);
    const bf16_t* ximg = a.x + (size_t)img * a.H * a.W * C + cch2 * 8;
    u32x4 v0[U2];
    int pr = pp2 / Wp, pc = pp2 - pr * Wp;
#pragma unroll
    for (int u = 0; u < U2; ++u) {
      const int hi = r0 - 1 + pr, wi = pc - 1;
      ok1[u] = pr < rows + 2 && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
      // unconditional loads from a clamped in-image address, zeroed by a select: a load under a branch drains vmcnt at the join and
      // the 38 loads of the fill would go out one round trip at a time
      const int hc = hi < 0 ? 0 : (hi >= a.H ? a.H - 1 : hi), wc = wi < 0 ? 0 : (wi >= a.W ? a.W - 1 : wi);
      v0[u] = *reinterpret_cast<const u32x4*>(ximg + (size_t)(hc * a.W + wc) * C);
      v1[u] = *reinterpret_cast<const u32x4*>(ximg + (size_t)(hc * a.W + wc) * C + C / 2);
      if (!ok1[u]) { v0[u] = u32x4{0u, 0u, 0u, 0u}; v1[u] = u32x4{0u, 0u, 0u, 0u}; }
      pc += 16;
      while (pc >= Wp) { pc -= Wp; ++pr; }
    }
    float sc0[8], sh0[8];
    if (xf_) {
      const float inv = 1.0f / a.in_count;
#pragma unroll
      for (int k = 0; k < NSH; ++k) {
        const int c = tid + 256 * k;
        if (c < C) stat_head_finish(shd[k], a.in_stats, a.in_srep, C, c, inv, a.in_eps, coef[c], coef[C + c]);
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sc0[e] = coef[cch2 * 8 + e]; sh0[e] = coef[C + cch2 * 8 + e];
        sc1[e] = coef[C / 2 + cch2 * 8 + e]; sh1[e] = coef[C + C / 2 + cch2 * 8 + e];
      }
    }
#pragma unroll
    for (int u = 0; u < U2; ++u) {
      if (xf_ && ok1[u]) bn_relu_chunk(v0[u], sc0, sh0);
      if (pp2 + 16 * u < total) *reinterpret_cast<u32x4*>(smem + (size_t)(pp2 + 16 * u) * PIX + cch2 * 16) = v0[u];
    }
  } else
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
        const int hc = hi < 0 ? 0 : (hi >= a.H ? a.H - 1 : hi), wc = wi < 0 ? 0 : (wi >= a.W ? a.W - 1 : wi);
        v[u] = *reinterpret_cast<const u32x4*>(ximg + (size_t)(hc * a.W + wc) * C);     // unconditional (clamped), zeroed by a select
        if (!ok[u]) v[u] = u32x4{0u, 0u, 0u, 0u};
        pc += PSTEP;
        while (pc >= Wp) { pc -= Wp; ++pr; }
      }
      if (xf && first) {
        const float inv = 1.0f / a.in_count;
#pragma unroll
        for (int k = 0; k < NSH; ++k) {
          const int c = tid + 256 * k;
          if (c < C) stat_head_finish(shd[k], a.in_stats, a.in_srep, C, c, inv, a.in_eps, coef[c], coef[C + c]);
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
  f32x4 acc[TMW][NTW];
#pragma unroll
  for (int i = 0; i < TMW; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Straight-line K loop (fully unrolled: no loop-carried register shuffling), software-pipelined in source order and
  // pinned with scheduling barriers: step s issues the LDS reads of step s + 1, runs its TM x NTW MFMAs, then re-requests
  // its filter registers for step s + WQ.  The waits the compiler inserts are then counted (lgkmcnt(TM), vmcnt((WQ-1) NTW)).
  constexpr int TH = TMW > 7 ? 7 : TMW;                             // tiles addressed from the first base register (16-bit ds offsets)
  const char* abase = smem + (wm * TMW * 16 + r16) * PIX + q4 * 16;
  // K-step s -> (channel half, tap, 32-channel step inside the half): s = (half * 9 + tap) * CSH + cs
  auto read_a = [&](u32x4 (&fa)[TMW], int s) {
    const int half = s / (9 * CSH), tap = (s / CSH) % 9, cs = half * CSH + s % CSH;
    const char* ab = abase + ((tap / 3) * Wp + tap % 3) * PIX;
    const char* ab2 = ab + TH * 16 * PIX;
#pragma unroll
    for (int i = 0; i < TMW; ++i)
      fa[i] = i < TH ? *reinterpret_cast<const u32x4*>(ab + i * 16 * PIX + cs * 64) : *reinterpret_cast<const u32x4*>(ab2 + (i - TH) * 16 * PIX + cs * 64);
  };
  u32x4 fa0[TMW], fa1[TMW];
  read_a(fa0, 0);
#pragma clang loop unroll(full)
  for (int s = 0; s < KS; ++s) {
    u32x4 (&cur)[TMW] = (s & 1) ? fa1 : fa0;
    u32x4 (&nxt)[TMW] = (s & 1) ? fa0 : fa1;
    constexpr int SPLIT = HALVES == 2 ? KS / 2 : -1;               // first K-step that reads the second channel half
    if (s + 1 < KS && s + 1 != SPLIT) read_a(nxt, s + 1);
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[i][j] = mfma_bf16(wq[s % WQ][j], cur[i], acc[i][j]);
    if (s + WQ < KS) {
#pragma unroll
      for (int j = 0; j < NTW; ++j) wq[s % WQ][j] = wl[((size_t)(T0 + j) * KS + s + WQ) * 64];
    }
    if constexpr (HALVES == 2) {
      if (s < U2) {                                                 // this K-step's share of the second half: one chunk per thread
        if (xf_ && ok1[s]) bn_relu_chunk(v1[s], sc1, sh1);
        if (pp2 + 16 * s < 16 * TM + 2 * Wp + 2) *reinterpret_cast<u32x4*>(smem + (size_t)(pp2 + 16 * s) * PIX + C + cch2 * 16) = v1[s];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (HALVES == 2) {
      if (s + 1 == SPLIT) { __syncthreads(); read_a(nxt, s + 1); }   // the second half is complete in LDS for every wave
    }
  }

  IMG_STAMP(3);
  // ---- epilogue: accumulators -> (statistics) -> (scale/shift, ReLU) -> bf16 -> 8 * NTW-byte stores ----------------
  constexpr int NC = 4 * NTW;                                        // consecutive channels per lane
  const int cb = T0 * 16 + NC * q4;                                  // tile j, register e <-> channel cb + 4 j + e
  float es[NC], ess[NC], scv[NC], shv[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) { es[c] = 0.f; ess[c] = 0.f; scv[c] = 1.f; shv[c] = 0.f; }
  if constexpr (AFFINE) {   // eval-mode scale / shift: loaded and waited for HERE (a first use under the per-tile branch put vmcnt(0) into
                            // every tile of the epilogue, train mode included: each tile's stores were drained before the next tile)
#pragma unroll
    for (int c = 0; c < NC; ++c) { scv[c] = a.scale[cb + c]; shv[c] = a.shift[cb + c]; }
#pragma unroll
    for (int c = 0; c < NC; ++c) asm volatile("" : "+v"(scv[c]), "+v"(shv[c]));
  }
  int ho = r0, wo = r16 + wm * TMW * 16;
  while (wo >= Wp) { wo -= Wp; ++ho; }
  const long yimg = ((long)img * a.H * a.W * a.N + cb) * 2;           // byte offset into a.y (st_out_store16: uniform base + lane offset)
#pragma unroll
  for (int i = 0; i < TMW; ++i) {
    const bool valid = wo < a.W && ho < r0 + rows;
    if (valid) {
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
      const long dst = yimg + (long)(ho * a.W + wo) * a.N * 2;
      if constexpr (NTW == 1) {
        st_out_store8(a.y, dst, u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])});
      } else {
#pragma unroll
        for (int h = 0; h < NTW / 2; ++h)
          st_out_store16(a.y, dst + 16 * h, u32x4{pack_bf16x2(v[8 * h], v[8 * h + 1]), pack_bf16x2(v[8 * h + 2], v[8 * h + 3]),
                                                  pack_bf16x2(v[8 * h + 4], v[8 * h + 5]), pack_bf16x2(v[8 * h + 6], v[8 * h + 7])});
      }
    }
    wo += 16;
    while (wo >= Wp) { wo -= Wp; ++ho; }
  }
  if (a.stats) {
    float* sdst = a.stats + (a.srep > 1 ? (size_t)(bi % a.srep) * 2 * a.N : 0);
#pragma unroll
    for (int c = 0; c < NC; ++c) { es[c] = row16_sum_(es[c]); ess[c] = row16_sum_(ess[c]); }
    if constexpr (MS == 1) {
      block_stats_flush<NC, 64 * NTW>(es, ess, r16 == 0, wid * 16 * NTW + NC * q4, smem, sdst, nb * 64 * NTW, a.N, tid);
    } else {   // two waves (position halves) hold partials of the same channels: the second adds to what the first parked
      constexpr int BNLOC = (4 / MS) * 16 * NTW;
      float* sred = reinterpret_cast<float*>(smem);
      const int local = wn * 16 * NTW + NC * q4;
      __syncthreads();
      if (r16 == 0 && wm == 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) { sred[local + c] = es[c]; sred[BNLOC + local + c] = ess[c]; }
      }
      __syncthreads();
      if (r16 == 0 && wm == 1) {
#pragma unroll
        for (int c = 0; c < NC; ++c) { sred[local + c] += es[c]; sred[BNLOC + local + c] += ess[c]; }
      }
      __syncthreads();
      const int chan0 = nb * BNLOC;
      for (int t = tid; t < 2 * BNLOC; t += 256) atomicAdd(sdst + (t < BNLOC ? chan0 + t : a.N + chan0 + t - BNLOC), sred[t]);
    }
  }
  IMG_STAMP(4);
#undef IMG_STAMP
}

// Two entry points over one body: the 128..512-channel forms want the whole register file of a SIMD for ONE wave (waves_per_eu(1, 1)
// lets the allocator use it); the 64-channel form (202 registers, 58 KB of LDS) runs two workgroups per CU -- under (1, 1) it was
// held to one and took 7 rounds instead of 3.5 on the 56 x 56 layers.
template <int C, int TM, int NTW, int HALVES, bool AFFINE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_img_kernel(ImgArgs a) {
  conv3x3_img_body<C, TM, NTW, HALVES, AFFINE>(a);
}
template <int C, int TM, int NTW, int HALVES, bool AFFINE>
__global__ __launch_bounds__(256, 2) void conv3x3_img_kernel_occ2(ImgArgs a) {
  conv3x3_img_body<C, TM, NTW, HALVES, AFFINE, (C == 64 ? 2 : 1)>(a);
}

// =====================================================================================================================
// Pointwise (1x1) convolution with the filter slice in REGISTERS: K <= 512 input channels (conv3 and the downsample
// convs of torchvision's Bottleneck, the 64/256-channel conv1s of layer1; reference cnn.py:46).
//   * a wave owns 16 NTW output channels for the whole launch and keeps their K x 16 NTW filter slice as MFMA operand
//     fragments in K/32 x NTW x 4 VGPRs (read once, fragment-major: one coalesced 1-KiB load per operand);
//   * the workgroup (4 waves = 64 NTW channels) walks a range of 16 TMS-row stages: the rows of stage s+1 are requested
//     from HBM/L2 while stage s is multiplied, normalised on the way in (producer's BatchNorm + ReLU, once per element
//     per channel slice), written to the other half of a two-stage LDS ring (rows padded to 2K + 32 bytes: conflict-
//     free ds_read_b128) -- ONE barrier per stage, none inside it;
//   * a lane ends with 4 NTW consecutive channels of a row: direct 8/16-byte stores, per-lane statistics partials are
//     carried in registers across the whole range and reduced once (DPP + one atomic per channel per workgroup).
// Small per-wave footprint (<= 128 VGPRs at K = 256, NTW = 2): two or more workgroups per CU overlap each other's stage edges.
struct PwArgs {
  const bf16_t* x; const bf16_t* w; bf16_t* y;
  float* stats; int srep;
  const float* scale; const float* shift; int relu;
  const float* in_stats; const float* in_gamma; const float* in_beta; float in_count, in_eps; int in_srep;
  int M, N, nbn, nstage, spb, mbs;       // rows, channels, channel slices, 16 TMS-row stages in all / per workgroup, row-range workgroups (0: contiguous ranges)
  int Hin, Win, Ho, Wo, stride;          // stride > 1: output row (b, ho, wo) reads input pixel (b, ho * stride, wo * stride)
  // FUSE (conv1 fused with the previous block's end): x is the RAW conv3 output, in_stats.. its BatchNorm; the loader forms
  // relu(bn(x) + bn_r(res)) (bn_r = identity when res_stats is NULL), feeds it to the MFMAs and writes it ONCE to xout
  const bf16_t* res; bf16_t* xout;
  const float* res_stats; const float* res_gamma; const float* res_beta; int res_srep;
  // RES (eval mode, conv3 of a Bottleneck): y = relu(conv(x) * scale + shift + identity); identity: [M][N]
  const bf16_t* idn;
};

// relu(x * sc + sh) on 8 bf16 (one 16-byte chunk), rounded once: the same arithmetic as bn_act's pass
__device__ __forceinline__ void bn_relu_chunk(u32x4& v, const float* sc, const float* sh) {
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const float lo = fmaxf(__uint_as_float(v[d] << 16) * sc[2 * d] + sh[2 * d], 0.f);
    const float hi = fmaxf(__uint_as_float(v[d] & 0xffff0000u) * sc[2 * d + 1] + sh[2 * d + 1], 0.f);
    v[d] = pack_bf16x2(lo, hi);
  }
}

template <int K, int NTW, int TMS, int D, bool STRIDED, bool AFFINE, bool FUSE = false, bool RES = false>
__global__ __launch_bounds__(256) void conv1x1_wreg_kernel(PwArgs a) {
  static_assert(!RES || (AFFINE && !STRIDED && !FUSE), "the residual epilogue is the eval-mode conv3 form");
  constexpr int PIX = 2 * K + 32;
  constexpr int KS = K / 32;
  constexpr int CH8 = K / 8;                  // 16-byte chunks per row
  constexpr int SM = 16 * TMS;                // rows per stage
  constexpr int RPP = 256 / CH8;              // rows per loader pass
  constexpr int NL = SM / RPP;                // loads per thread per stage
  static_assert(SM % RPP == 0 && NL >= 1, "stage/loader mismatch");
  constexpr int STAGE_BYTES = SM * PIX;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lid;
  {
    const int nblk = gridDim.x, id = blockIdx.x, xcd = id & 7, q = nblk >> 3, r = nblk & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int slice = lid % a.nbn, mb = lid / a.nbn;
  // A workgroup walks spb stages.  a.mbs > 0 (default): INTERLEAVED -- local stage k is global stage mb + k mbs, so at any time the
  // whole grid reads and writes inside one moving window of mbs stages (a few MB), like a grid-stride pass; with a contiguous range per
  // workgroup (a.mbs == 0) the ~500 workgroups stream ~2000 separate regions of the tensors at once and the HBM-bound layers ran at
  // 4.2 - 4.6 TB/s against bn_act's 5.8 on the same bytes.  Stages past the end hold no rows (their loads re-read row M - 1, masked).
  const int s_begin = 0;
  auto gs = [&](int k) { return a.mbs > 0 ? mb + k * a.mbs : mb * a.spb + k; };
  if (gs(0) >= a.nstage) return;

  // ---- filter slice -> registers --------------------------------------------------------------------------------
  const int r16 = lane & 15, q4 = lane >> 4;
  const int T0 = (slice * 4 + wid) * NTW;
  const u32x4* wl = reinterpret_cast<const u32x4*>(a.w) + lane;
  u32x4 wq[KS][NTW];
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int j = 0; j < NTW; ++j) wq[s][j] = wl[((size_t)(T0 + j) * KS + s) * 64];

  // ---- loader state ---------------------------------------------------------------------------------------------
  const int cch = tid % CH8, lrow = tid / CH8;
  const bool xf = a.in_stats != nullptr;
  const bool rbn = FUSE && a.res_stats != nullptr;
  float sc[8], sh[8], sc2[FUSE ? 8 : 1], sh2[FUSE ? 8 : 1];
  // D register sets: the rows of the next D stages are in flight while a stage is multiplied (a stage is shorter than one
  // HBM / L2 round trip: with one set every stage paid that round trip in full)
  u32x4 ra[D][NL]; bool rok[D][NL];
  u32x4 rb[FUSE ? D : 1][FUSE ? NL : 1];           // FUSE: the residual rows travel with the raw ones
  auto gload = [&](u32x4 (&r)[NL], bool (&ok)[NL], int stage, u32x4* q = nullptr) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      // UNCONDITIONAL loads (rows past the end re-read the last row; their results are never stored): a load under a branch
      // makes the compiler drain vmcnt at the join, which would serialise the prefetch
      int m = gs(stage) * SM + lrow + i * RPP;
      ok[i] = m < a.M && stage < a.spb;            // local stages past the range (the prefetch runs D ahead): every lane re-reads row M - 1
      m = ok[i] ? m : a.M - 1;
      long src = m;
      if constexpr (STRIDED) {
        const int hw = a.Ho * a.Wo, b = m / hw, rem = m - b * hw, ho = rem / a.Wo, wo = rem - ho * a.Wo;
        src = ((long)b * a.Hin + ho * a.stride) * a.Win + wo * a.stride;
      }
      r[i] = *reinterpret_cast<const u32x4*>(a.x + src * K + cch * 8);
      if constexpr (FUSE) q[i] = *reinterpret_cast<const u32x4*>(a.res + src * K + cch * 8);
    }
  };
  auto lstore = [&](u32x4 (&r)[NL], bool (&ok)[NL], int buf, int stage = 0, const u32x4* q = nullptr) {
    char* base = smem + buf * STAGE_BYTES + lrow * PIX + cch * 16;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      if constexpr (FUSE) {
        // relu(bn(raw) + identity): st_bn_act's arithmetic (one fma per term, one rounding to bf16)
        u32x4 v;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          float rl = __uint_as_float(q[i][d] << 16), rh = __uint_as_float(q[i][d] & 0xffff0000u);
          if (rbn) { rl = __builtin_fmaf(rl, sc2[2 * d], sh2[2 * d]); rh = __builtin_fmaf(rh, sc2[2 * d + 1], sh2[2 * d + 1]); }
          const float lo = fmaxf(__builtin_fmaf(__uint_as_float(r[i][d] << 16), sc[2 * d], sh[2 * d]) + rl, 0.f);
          const float hi = fmaxf(__builtin_fmaf(__uint_as_float(r[i][d] & 0xffff0000u), sc[2 * d + 1], sh[2 * d + 1]) + rh, 0.f);
          v[d] = pack_bf16x2(lo, hi);
        }
        *reinterpret_cast<u32x4*>(base + i * RPP * PIX) = v;
        // every channel slice forms the same rows; slice 0 writes them back (the next identity)
        if (slice == 0 && ok[i]) *reinterpret_cast<u32x4*>(a.xout + (size_t)(gs(stage) * SM + lrow + i * RPP) * K + cch * 8) = v;
      } else {
        if (xf) bn_relu_chunk(r[i], sc, sh);
        *reinterpret_cast<u32x4*>(base + i * RPP * PIX) = r[i];
      }
    }
  };

  // RES: the identity pieces of a stage in the ACCUMULATOR layout (this lane's 4 NTW channels of row i * 16 + r16), requested D stages
  // ahead into register set (stage % D) -- right after that set's previous use, unconditional, rows past the end clamped
  constexpr int NCQ = RES ? NTW : 1;                 // dwordx2 pieces per tile (4 channels each)
  u32x2 rq[RES ? D : 1][RES ? TMS : 1][NCQ];
  auto qload = [&](u32x2 (&q)[RES ? TMS : 1][NCQ], int stage) {
    if constexpr (RES) {
      const int cbq = ((slice * 4 + wid) * NTW) * 16 + 4 * NTW * q4;
#pragma unroll
      for (int i = 0; i < TMS; ++i) {
        int m = gs(stage) * SM + i * 16 + r16;
        m = (m < a.M && stage < a.spb) ? m : a.M - 1;
        const bf16_t* src = a.idn + (size_t)m * a.N + cbq;
        if constexpr (NTW == 1) q[i][0] = *reinterpret_cast<const u32x2*>(src);
        else {
#pragma unroll
          for (int h = 0; h < NTW / 2; ++h) {
            const u32x4 t = *reinterpret_cast<const u32x4*>(src + 8 * h);
            q[i][2 * h] = u32x2{t[0], t[1]}; q[i][2 * h + 1] = u32x2{t[2], t[3]};
          }
        }
      }
    }
  };
  gload(ra[0], rok[0], s_begin, rb[0]);
  if (xf) {   // producer's BatchNorm coefficients (replicated statistics summed here), while the first rows are in flight
    float* coef = reinterpret_cast<float*>(smem + 2 * STAGE_BYTES);
    const float inv = 1.0f / a.in_count;
    for (int c = tid; c < K; c += 256) {
      float sm, sq;
      stat_sum(a.in_stats, a.in_srep, K, c, sm, sq);
      bn_scale_shift(sm, sq, inv, a.in_gamma[c], a.in_beta[c], a.in_eps, coef[c], coef[K + c]);
      if constexpr (FUSE) {
        float s2 = 1.f, h2 = 0.f;
        if (rbn) {
          float rm, rq;
          stat_sum(a.res_stats, a.res_srep, K, c, rm, rq);
          bn_scale_shift(rm, rq, inv, a.res_gamma[c], a.res_beta[c], a.in_eps, s2, h2);
        }
        coef[2 * K + c] = s2; coef[3 * K + c] = h2;
      }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = coef[cch * 8 + e]; sh[e] = coef[K + cch * 8 + e]; }
    if constexpr (FUSE) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { sc2[e] = coef[2 * K + cch * 8 + e]; sh2[e] = coef[3 * K + cch * 8 + e]; }
    }
  }
  lstore(ra[0], rok[0], 0, s_begin, rb[0]);
#pragma unroll
  for (int j = 1; j <= D; ++j)                       // stage s_begin + j waits in register set j % D (always issued: counted waits)
    gload(ra[j % D], rok[j % D], s_begin + j, rb[FUSE ? j % D : 0]);
  if constexpr (RES) {
#pragma unroll
    for (int j = 0; j < D; ++j) qload(rq[j], s_begin + j);
  }
  __syncthreads();

  constexpr int NC = 4 * NTW;
  const int cb = T0 * 16 + NC * q4;
  float es[NC], ess[NC], scv[NC], shv[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) { es[c] = 0.f; ess[c] = 0.f; scv[c] = 1.f; shv[c] = 0.f; }
  if constexpr (AFFINE) {   // eval-mode scale / shift: loaded and WAITED FOR here -- a first use inside the loop would put vmcnt(0) there
#pragma unroll
    for (int c = 0; c < NC; ++c) { scv[c] = a.scale[cb + c]; shv[c] = a.shift[cb + c]; }
#pragma unroll
    for (int c = 0; c < NC; ++c) asm volatile("" : "+v"(scv[c]), "+v"(shv[c]));
  }
  const char* abase = smem + r16 * PIX + q4 * 16;

  // The host makes the range a multiple of D stages (rows past M are masked), so the D unrolled copies below form one
  // straight-line body: every prefetch is issued on every path and the compiler's waits are COUNTED (vmcnt((D-1) NL + ..)).
  // A load or a wait under a branch here collapses them to vmcnt(0) and with it the prefetch depth to one stage.
  for (int s0 = s_begin; s0 < s_begin + a.spb; s0 += D) {
#pragma unroll
   for (int u = 0; u < D; ++u) {
    const int s = s0 + u;
    {
    const int buf = (s - s_begin) & 1;
    const char* ab = abase + buf * STAGE_BYTES;
    f32x4 acc[TMS][NTW];
#pragma unroll
    for (int i = 0; i < TMS; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u32x4 fa[TMS];
#pragma unroll
      for (int i = 0; i < TMS; ++i) fa[i] = *reinterpret_cast<const u32x4*>(ab + i * 16 * PIX + ks * 64);
#pragma unroll
      for (int i = 0; i < TMS; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = mfma_bf16(wq[ks][j], fa[i], acc[i][j]);
    }
    // epilogue of the stage: straight from the accumulators
#pragma unroll
    for (int i = 0; i < TMS; ++i) {
      const int m = gs(s) * SM + i * 16 + r16;
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
          if constexpr (RES) {
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
              v[4 * j] += __uint_as_float(rq[u][i][j][0] << 16); v[4 * j + 1] += __uint_as_float(rq[u][i][j][0] & 0xffff0000u);
              v[4 * j + 2] += __uint_as_float(rq[u][i][j][1] << 16); v[4 * j + 3] += __uint_as_float(rq[u][i][j][1] & 0xffff0000u);
            }
          }
          if (a.relu) {
#pragma unroll
            for (int c = 0; c < NC; ++c) v[c] = fmaxf(v[c], 0.f);
          }
        }
        const long dst = ((long)m * a.N + cb) * 2;
        if (a.y) {                                     // y == NULL: statistics only (st_conv_b2b recomputes the output where it is consumed)
        if constexpr (NTW == 1) {
          st_out_store8(a.y, dst, u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])});
        } else {
#pragma unroll
          for (int h = 0; h < NTW / 2; ++h)
            st_out_store16(a.y, dst + 16 * h, u32x4{pack_bf16x2(v[8 * h], v[8 * h + 1]), pack_bf16x2(v[8 * h + 2], v[8 * h + 3]),
                                                    pack_bf16x2(v[8 * h + 4], v[8 * h + 5]), pack_bf16x2(v[8 * h + 6], v[8 * h + 7])});
        }
        }
      }
    }
    // next stage: its register set -> the other ring half, then that set requests stage s + 1 + D
    lstore(ra[(u + 1) % D], rok[(u + 1) % D], buf ^ 1, s + 1, rb[FUSE ? (u + 1) % D : 0]);
    gload(ra[(u + 1) % D], rok[(u + 1) % D], s + 1 + D, rb[FUSE ? (u + 1) % D : 0]);
    if constexpr (RES) qload(rq[u], s + D);
    __syncthreads();
    }
   }
  }

  if (a.stats) {
    float* sdst = a.stats + (a.srep > 1 ? (size_t)(mb % a.srep) * 2 * a.N : 0);
#pragma unroll
    for (int c = 0; c < NC; ++c) { es[c] = row16_sum_(es[c]); ess[c] = row16_sum_(ess[c]); }
    block_stats_flush<NC, 64 * NTW>(es, ess, r16 == 0, wid * 16 * NTW + NC * q4, smem, sdst, slice * 64 * NTW, a.N, tid);
  }
}

// =====================================================================================================================
// Pointwise convolution with LONG K (1024 / 2048 input channels: the conv1 of the layer3 / layer4 Bottlenecks, reference
// cnn.py:46): a workgroup owns 112 rows x 256 output channels and streams K.
//   * per 128-channel slab the 112 x 128 activation block goes global -> registers -> LDS (padded 288-byte rows, two-slab
//     ring, two register sets in flight: counted waits), ONE barrier per slab (4 K-steps, 112 MFMAs per wave);
//   * the filters never touch LDS: each wave (64 channels) pulls its fragment-major operands (1 KiB per 16 channels x 32 K)
//     through a 6-K-step register ring, re-requested right after use (lead time: 1.5 slabs);
//   * straight-line code (K is a template parameter), scheduling barriers between K-steps as in conv3x3_img_kernel;
//   * 112 x 256 tiles read every activation ONCE (N <= 256) and the filter bank once per 112 rows: 165 MB of L2->CU
//     traffic for the 1024 -> 256 layer at 14 x 14 against 200 MB with 128 x 128 tiles, and no tile-count quantisation
//     (224 workgroups on 256 CUs).
struct KsArgs {
  const bf16_t* x; const bf16_t* w; bf16_t* y;
  float* stats; int srep;
  const float* scale; const float* shift; int relu;
  int M, N, nbn;
  int Hin, Win, Ho, Wo, stride;
  unsigned long long* stamps;   // debug: per-wave s_memtime at phase boundaries (tools/ks_stamps.py), normally NULL
};

// NTW = 2 (32 channels per wave, 128 per workgroup): <= 256 registers, 65 KB of LDS -- two workgroups per CU (every K = 1024 / 2048 layer except
// the 1024 -> 256 conv1s, whose ntw = 4 packing st_conv_c3c1 indexes)
template <int K, bool STRIDED, bool AFFINE, int NTW = 4>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NTW == 2 ? 2 : 1, NTW == 2 ? 2 : 1))) void conv1x1_kstream_kernel(KsArgs a) {
  constexpr int TM = 7, BM = 16 * TM;
  constexpr int SLAB = 128, PIX = 2 * SLAB + 32, KSS = SLAB / 32;      // 4 K-steps per slab
  constexpr int NSLAB = K / SLAB, KS = K / 32;
  constexpr int SLAB_BYTES = BM * PIX;
  constexpr int NL = BM * (SLAB / 8) / 256;                            // 7 16-byte loads per thread per slab
  constexpr int WR = 6;                                                // filter ring: K-steps in flight (1.5 slabs of lead; 8 ran out of VGPRs)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lid;
  {
    const int nblk = gridDim.x, id = blockIdx.x, xcd = id & 7, q = nblk >> 3, r = nblk & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int nb = lid % a.nbn, bm = lid / a.nbn;
  const int r16 = lane & 15, q4 = lane >> 4;
  const int T0 = (nb * 4 + wid) * NTW;
  const u32x4* wl = reinterpret_cast<const u32x4*>(a.w) + lane;
#define KS_STAMP(i) do { if (a.stamps && lane == 0) a.stamps[((size_t)blockIdx.x * 4 + wid) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
  KS_STAMP(0);
  if (a.stamps && lane == 0) a.stamps[((size_t)blockIdx.x * 4 + wid) * 8 + 6] = __builtin_amdgcn_s_memrealtime();

  // ---- loader: thread -> (chunk of the slab row, 7 rows 16 apart); rows past M re-read row M - 1 (never stored) ---------
  const int cch = tid & 15, lrow = tid >> 4;
  const bf16_t* src[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    int m = bm * BM + lrow + 16 * i;
    m = m < a.M ? m : a.M - 1;
    long row = m;
    if constexpr (STRIDED) {
      const int hw = a.Ho * a.Wo, b = m / hw, rem = m - b * hw, ho = rem / a.Wo, wo = rem - ho * a.Wo;
      row = ((long)b * a.Hin + ho * a.stride) * a.Win + wo * a.stride;
    }
    src[i] = a.x + row * K + cch * 8;
  }
  u32x4 ra[2][NL];
  auto gload = [&](u32x4 (&r)[NL], int slab) {
#pragma unroll
    for (int i = 0; i < NL; ++i) r[i] = *reinterpret_cast<const u32x4*>(src[i] + (slab < NSLAB ? slab : NSLAB - 1) * SLAB);
  };
  auto lstore = [&](u32x4 (&r)[NL], int buf) {
    char* base = smem + buf * SLAB_BYTES + lrow * PIX + cch * 16;
#pragma unroll
    for (int i = 0; i < NL; ++i) *reinterpret_cast<u32x4*>(base + 16 * i * PIX) = r[i];
  };

  u32x4 wq[WR][NTW];
  gload(ra[0], 0);
#pragma unroll
  for (int s = 0; s < WR; ++s)
#pragma unroll
    for (int j = 0; j < NTW; ++j) wq[s][j] = wl[((size_t)(T0 + j) * KS + s) * 64];
  gload(ra[1], 1);
  lstore(ra[0], 0);
  gload(ra[0], 2);
  __syncthreads();
  KS_STAMP(1);

  f32x4 acc[TM][NTW];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const char* abase = smem + r16 * PIX + q4 * 16;

  u32x4 fa0[TM], fa1[TM];
#pragma unroll
  for (int s = 0; s < NSLAB; ++s) {
    const char* ab = abase + (s & 1) * SLAB_BYTES;
    auto read_a = [&](u32x4 (&f)[TM], int kk) {
#pragma unroll
      for (int i = 0; i < TM; ++i) f[i] = *reinterpret_cast<const u32x4*>(ab + i * 16 * PIX + kk * 64);
    };
    read_a(fa0, 0);                                   // the first K-step of a slab cannot be read before the slab's barrier
#pragma unroll
    for (int kk = 0; kk < KSS; ++kk) {
      const int ks = s * KSS + kk;
      u32x4 (&fa)[TM] = (kk & 1) ? fa1 : fa0;
      u32x4 (&fn)[TM] = (kk & 1) ? fa0 : fa1;
      if (kk + 1 < KSS) read_a(fn, kk + 1);           // the next K-step's operands ride under this K-step's MFMAs
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = mfma_bf16(wq[ks % WR][j], fa[i], acc[i][j]);
      if (ks + WR < KS) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) wq[ks % WR][j] = wl[((size_t)(T0 + j) * KS + ks + WR) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (s + 1 < NSLAB) {
      lstore(ra[(s + 1) & 1], (s + 1) & 1);            // slab s+1 -> the ring half slab s-1 was read from (all waves passed the last barrier)
      gload(ra[(s + 1) & 1], s + 3);                   // that register set requests slab s+3 (clamped past the end: counted waits stay valid)
    }
    __syncthreads();
    if (s == 0) KS_STAMP(2);
    if (s == 3) KS_STAMP(3);
    if (s == NSLAB - 1) KS_STAMP(4);
  }

  // ---- epilogue ----------------------------------------------------------------------------------------------------------
  constexpr int NC = 4 * NTW;
  const int cb = T0 * 16 + NC * q4;
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
#pragma unroll
      for (int h = 0; h < NTW / 2; ++h)
        st_out_store16(a.y, dst + 16 * h, u32x4{pack_bf16x2(v[8 * h], v[8 * h + 1]), pack_bf16x2(v[8 * h + 2], v[8 * h + 3]),
                                                pack_bf16x2(v[8 * h + 4], v[8 * h + 5]), pack_bf16x2(v[8 * h + 6], v[8 * h + 7])});
    }
  }
  if (a.stats) {
    float* sdst = a.stats + (a.srep > 1 ? (size_t)(bm % a.srep) * 2 * a.N : 0);
#pragma unroll
    for (int c = 0; c < NC; ++c) { es[c] = row16_sum_(es[c]); ess[c] = row16_sum_(ess[c]); }
    block_stats_flush<NC, 64 * NTW>(es, ess, r16 == 0, wid * 16 * NTW + NC * q4, smem, sdst, nb * 64 * NTW, a.N, tid);
  }
  KS_STAMP(5);
  if (a.stamps && lane == 0) a.stamps[((size_t)blockIdx.x * 4 + wid) * 8 + 7] = __builtin_amdgcn_s_memrealtime();
#undef KS_STAMP
}

// =====================================================================================================================
// Pointwise convolution, activation-stationary: K = 256 / 512 input channels, N a multiple of 256 output channels (the
// conv3 of the layer3 / layer4 Bottlenecks, 256 -> 1024 and 512 -> 2048; reference cnn.py:46).  A workgroup owns 112 rows:
//   * its 112 x K activation block is read ONCE, normalised on the way in (producer's BatchNorm + ReLU -- once per element,
//     not once per output-channel slice as in conv1x1_wreg_kernel) and stays in LDS (padded rows);
//   * the workgroup then walks ALL output channels in chunks of 256 (64 per wave) with NO barrier: filters stream
//     fragment-major through the 6-K-step register ring across chunk boundaries, the chunk's epilogue (stores straight from
//     the accumulators) sits in the same straight-line stream as the next chunk's MFMAs;
//   * per-channel statistics of every chunk are parked in LDS and leave as full-wave atomics once, at the end.
struct AsArgs {
  const bf16_t* x; const bf16_t* w; bf16_t* y;
  float* stats; int srep;
  const float* scale; const float* shift; int relu;
  const float* in_stats; const float* in_gamma; const float* in_beta; float in_count, in_eps; int in_srep;
  int M, N, nq;                  // nq: channel parts (workgroups per row block)
  int Hin, Win, Ho, Wo, stride;  // STRIDED: output row (b, ho, wo) reads input pixel (b, ho * stride, wo * stride)
  unsigned long long* stamps;   // debug (tools/as_stamps.py), normally NULL
  const bf16_t* res;            // RES: [M][N] residual (the block's identity), added after scale / shift, before the ReLU
};

// RES (eval mode, conv3 of a Bottleneck: out = relu(bn3(conv3(x)) + identity) with bn3 folded to scale / shift): the identity's 16-byte
// piece of tile i of chunk c is REQUESTED in K-step i of chunk c and consumed in K-step i of chunk c + 1 (where chunk c's epilogue
// runs), i.e. a full chunk (KS K-steps, 2 KS filter loads) ahead: unconditional, clamped rows -- the counted waits of the filter ring
// stay counted.  No statistics in this form.
// NTW = 4 (64 channels per wave and chunk; statistics only): every 16-row operand read from LDS feeds four MFMAs instead of two.  The filters
// stay in the ntw = 2 packing (what st_conv_c3c1 indexes): only the channel a lane's sums belong to changes (stats_park).
template <int K, int NCH, bool AFFINE, bool STRIDED, bool RES = false, int NTW = 2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv1x1_astat_kernel(AsArgs a) {
  static_assert(!RES || (AFFINE && !STRIDED), "the residual epilogue is the eval-mode conv3 form");
  static_assert(NTW == 2 || (NTW == 4 && !AFFINE && !RES), "NTW = 4 is the statistics-only form");
  // NCH chunks of 128 output channels (32 per wave, NTW = 2).  TWO accumulator sets alternate by chunk: the epilogue of chunk c-1
  // (accumulator reads, statistics, bf16 packing, stores: ~600 VALU instructions) is spread over the K-steps of chunk c, one
  // 16-row tile per K-step, so it runs under that chunk's MFMAs instead of between two MFMA blocks (measured before: 3.5 us per
  // 256-channel chunk of which 1.5 us MFMA).
  constexpr int TM = 7, BM = 16 * TM, CW = 64 * NTW;                // CW channels per chunk
  constexpr int PIX = 2 * K + 32, KS = K / 32, CH8 = K / 8;
  constexpr int RPP = 256 / CH8, NL = BM / RPP;                      // rows per loader pass, loads per thread
  constexpr int WR = 6;
  constexpr int TOT = NCH * KS;                                      // K-steps over the whole channel walk
  static_assert(KS >= TM + 1, "the chunk epilogue needs one K-step per tile and one for the statistics");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sstat = reinterpret_cast<float*>(smem + BM * PIX);          // [2][CW NCH] statistics of every chunk
  float* coef = sstat + 2 * CW * NCH;                                // [2][K] producer's scale / shift
  float* aff = coef + 2 * K;                                         // AFFINE: [scale(N) | shift(N)] of the eval-mode epilogue (a global load
                                                                     // inside the walk would drain the filter prefetch: vmcnt is in-order)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  // a.nq > 1: the output channels are cut into nq parts of NCH chunks each, one workgroup per (row block, part) -- layers with few rows
  // (512 -> 2048 at 7 x 7: 56 row blocks) fill the chip that way; the parts of a row block are neighbours on one XCD (its rows: L2 hits)
  int bm, bq;
  {
    const int nblk = gridDim.x, id = blockIdx.x, xcd = id & 7, q = nblk >> 3, r = nblk & 7;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    bm = lid / a.nq; bq = lid - bm * a.nq;
  }
  const int ch0 = bq * NCH;                                          // first 128-channel chunk of this workgroup
  const int r16 = lane & 15, q4 = lane >> 4;
  const u32x4* wl = reinterpret_cast<const u32x4*>(a.w) + lane;
  // K-step g of the walk: chunk g / KS, k-step g % KS, this wave's tiles (chunk * 4 + wid) * NTW + j
  auto wfrag = [&](int g, int j) { return wl[((size_t)(((ch0 + g / KS) * 4 + wid) * NTW + j) * KS + g % KS) * 64]; };
#define AS_STAMP(i) do { if (a.stamps && lane == 0) a.stamps[((size_t)blockIdx.x * 4 + wid) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
  AS_STAMP(0);
  if (a.stamps && lane == 0) a.stamps[((size_t)blockIdx.x * 4 + wid) * 8 + 6] = __builtin_amdgcn_s_memrealtime();
  constexpr int NSH = K > 256 ? K / 256 : 1;                          // producer's statistics first (stat_head_issue, common.h)
  StatHead shd[NSH];
  if (a.in_stats) {
#pragma unroll
    for (int k = 0; k < NSH; ++k) { const int c = (tid + 256 * k) % K; stat_head_issue(shd[k], a.in_stats, a.in_srep, K, c, a.in_gamma, a.in_beta); }
  }
  u32x4 wq[WR][NTW];
#pragma unroll
  for (int g = 0; g < WR; ++g)
#pragma unroll
    for (int j = 0; j < NTW; ++j) wq[g][j] = wfrag(g, j);

  // ---- fill: 112 rows x K, every load in flight before the first LDS write -------------------------------------------
  {
    const int cch = tid % CH8, lrow = tid / CH8;
    u32x4 v[NL];
    if constexpr (STRIDED) {
      // (image, row, column) of this thread's first output row by division, the next ones by carry (RPP < Wo on every layer
      // routed here; the loop form is right for any RPP)
      int m0 = bm * BM + lrow;
      m0 = m0 < a.M ? m0 : a.M - 1;
      const int hw = a.Ho * a.Wo;
      int b = m0 / hw, rem = m0 - b * hw, ho = rem / a.Wo, wo = rem - ho * a.Wo, m = m0;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const long src = ((long)b * a.Hin + ho * a.stride) * a.Win + wo * a.stride;
        v[i] = *reinterpret_cast<const u32x4*>(a.x + src * K + cch * 8);
        if (m + RPP < a.M) {                                          // rows past the end re-read the last valid row (never stored)
          m += RPP; wo += RPP;
          while (wo >= a.Wo) { wo -= a.Wo; if (++ho == a.Ho) { ho = 0; ++b; } }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        int m = bm * BM + lrow + i * RPP;
        m = m < a.M ? m : a.M - 1;
        v[i] = *reinterpret_cast<const u32x4*>(a.x + (size_t)m * K + cch * 8);
      }
    }
    if constexpr (AFFINE) {
      for (int c = tid; c < CW * NCH; c += 256) { aff[c] = a.scale[ch0 * CW + c]; aff[CW * NCH + c] = a.shift[ch0 * CW + c]; }
    }
    if (a.in_stats) {
      const float inv = 1.0f / a.in_count;
#pragma unroll
      for (int k = 0; k < NSH; ++k) {
        const int c = tid + 256 * k;
        if (c < K) stat_head_finish(shd[k], a.in_stats, a.in_srep, K, c, inv, a.in_eps, coef[c], coef[K + c]);
      }
      __syncthreads();
      float sc[8], sh[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { sc[e] = coef[cch * 8 + e]; sh[e] = coef[K + cch * 8 + e]; }
#pragma unroll
      for (int i = 0; i < NL; ++i) bn_relu_chunk(v[i], sc, sh);
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) *reinterpret_cast<u32x4*>(smem + (lrow + i * RPP) * PIX + cch * 16) = v[i];
  }
  __syncthreads();
  AS_STAMP(1);

  // ---- channel walk: no barrier from here to the statistics flush -------------------------------------------------------
  constexpr int NC = 4 * NTW;
  const char* abase = smem + r16 * PIX + q4 * 16;
  auto read_a = [&](u32x4 (&f)[TM], int ks) {
#pragma unroll
    for (int i = 0; i < TM; ++i) f[i] = *reinterpret_cast<const u32x4*>(abase + i * 16 * PIX + ks * 64);
  };
  f32x4 accA[TM][NTW], accB[TM][NTW];
  float es[NC], ess[NC];
  u32x4 rres[RES ? TM : 1];                                           // RES: the identity pieces of the chunk whose epilogue comes next
  // rows of this lane's accumulator tiles, clamped (rows past M re-read row M - 1: loads stay unconditional)
  auto res_request = [&](int ch, int i) {
    if constexpr (RES) {
      const int cb = (ch * 4 + wid) * NTW * 16 + NC * q4;
      int m = bm * BM + i * 16 + r16;
      m = m < a.M ? m : a.M - 1;
      rres[i] = *reinterpret_cast<const u32x4*>(a.res + (size_t)m * a.N + ch0 * CW + cb);
    }
  };
  // tile i of chunk `ch` from accumulator set `acc`: statistics partials, (eval: scale / shift / ReLU), bf16, one 16-byte store
  auto tile_epilogue = [&](f32x4 (&acc)[TM][NTW], int ch, int i) {
    const int cb = (ch * 4 + wid) * NTW * 16 + NC * q4;             // this lane's 8 consecutive channels (inside the workgroup's part)
    const int m = bm * BM + i * 16 + r16;
    if constexpr (!RES) {
      if (i == 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) { es[c] = 0.f; ess[c] = 0.f; }
      }
    }
    if (m < a.M) {
      float v[NC];
#pragma unroll
      for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * j + e] = acc[i][j][e];
      if constexpr (!RES) {
#pragma unroll
        for (int c = 0; c < NC; ++c) { es[c] += v[c]; ess[c] += v[c] * v[c]; }
      }
      if constexpr (AFFINE) {
#pragma unroll
        for (int c = 0; c < NC; c += 4) {
          const f32x4 s4 = *reinterpret_cast<const f32x4*>(aff + cb + c), h4 = *reinterpret_cast<const f32x4*>(aff + CW * NCH + cb + c);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[c + q] = v[c + q] * s4[q] + h4[q];
        }
        if constexpr (RES) {
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            v[2 * d] += __uint_as_float(rres[i][d] << 16);
            v[2 * d + 1] += __uint_as_float(rres[i][d] & 0xffff0000u);
          }
        }
        if (a.relu) {
#pragma unroll
          for (int c = 0; c < NC; ++c) v[c] = fmaxf(v[c], 0.f);
        }
      }
      if constexpr (NTW == 2)
      if (a.y)   // y == NULL: statistics only (st_conv_c3c1 recomputes the output where it is consumed)
        st_out_store16(a.y, ((long)m * a.N + ch0 * CW + cb) * 2,
                       u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])});
    }
  };
  auto stats_park = [&](int ch) {                                    // per-channel sums of chunk `ch` -> LDS (flushed once at the end)
    if constexpr (RES) return;
    if (a.stats) {
      const int cb = (ch * 4 + wid) * NTW * 16 + NC * q4;
#pragma unroll
      for (int c = 0; c < NC; ++c) { es[c] = row16_sum_(es[c]); ess[c] = row16_sum_(ess[c]); }
      if (r16 == 0) {
        if constexpr (NTW == 2) {
#pragma unroll
          for (int c = 0; c < NC; ++c) { sstat[cb + c] = es[c]; sstat[CW * NCH + cb + c] = ess[c]; }
        } else {   // tile 4 G + j of the ntw = 2 packing, row 4 q4 + e: channel 64 G + 32 (j / 2) + 8 q4 + 4 (j % 2) + e
          const int g64 = (ch * 4 + wid) * 64 + 8 * q4;
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const int o = g64 + 32 * ((c >> 2) >> 1) + 4 * ((c >> 2) & 1) + (c & 3);
            sstat[o] = es[c]; sstat[CW * NCH + o] = ess[c];
          }
        }
      }
    }
  };

  u32x4 fa0[TM], fa1[TM];
  read_a(fa0, 0);
#pragma clang loop unroll(full)
  for (int ch = 0; ch < NCH; ++ch) {
    f32x4 (&acc)[TM][NTW] = (ch & 1) ? accB : accA;
    f32x4 (&prev)[TM][NTW] = (ch & 1) ? accA : accB;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma clang loop unroll(full)
    for (int ks = 0; ks < KS; ++ks) {
      const int g = ch * KS + ks;
      u32x4 (&fa)[TM] = (g & 1) ? fa1 : fa0;
      u32x4 (&fn)[TM] = (g & 1) ? fa0 : fa1;
      if (g + 1 < TOT) read_a(fn, (ks + 1) % KS);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = mfma_bf16(wq[g % WR][j], fa[i], acc[i][j]);
      if (g + WR < TOT) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) wq[g % WR][j] = wfrag(g + WR, j);
      }
      if (ch > 0) {                                                 // the previous chunk's epilogue, one tile per K-step
        if (ks < TM) tile_epilogue(prev, ch - 1, ks);
        else if (ks == TM) stats_park(ch - 1);
      }
      if (ks < TM) res_request(ch, ks);                             // RES: this chunk's identity piece for tile ks (its register is free now)
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ch == 0) AS_STAMP(2);
    if (ch == NCH / 2 - 1) AS_STAMP(3);
  }
  {                                                                  // the last chunk's epilogue has nothing to hide under
    f32x4 (&last)[TM][NTW] = ((NCH - 1) & 1) ? accB : accA;
#pragma unroll
    for (int i = 0; i < TM; ++i) tile_epilogue(last, NCH - 1, i);
    stats_park(NCH - 1);
  }
  AS_STAMP(4);
  if (!RES && a.stats) {
    float* sdst = a.stats + (a.srep > 1 ? (size_t)(bm % a.srep) * 2 * a.N : 0);
    __syncthreads();
    for (int t = tid; t < 2 * CW * NCH; t += 256)                  // sstat = [sum | sumsq] of this part's CW NCH channels
      atomicAdd(sdst + (t < CW * NCH ? ch0 * CW + t : a.N + ch0 * CW + t - CW * NCH), sstat[t]);
  }
  AS_STAMP(5);
  if (a.stamps && lane == 0) a.stamps[((size_t)blockIdx.x * 4 + wid) * 8 + 7] = __builtin_amdgcn_s_memrealtime();
#undef AS_STAMP
}

// =====================================================================================================================
// The statistics-only pass of the 14 x 14 conv3 (256 -> 1024, train mode; st_conv_c3c1 recomputes the output where it is consumed) with
// TWO workgroups per CU: 112 rows x one part of the channels per workgroup, ONE accumulator set (<= 256 registers, 67 KB of LDS).  Two
// waves per SIMD fill each other's stalls (operand reads, filter waits, the chunk epilogue) -- what the second accumulator set and the
// interleaved epilogue of conv1x1_astat_kernel buy with registers, a second wave buys with occupancy, and the fill of one workgroup can run
// under the walk of the other.  The parts of a row block read the same rows (L2 hits).
// Filters in the ntw = 2 packing st_conv_c3c1 indexes.
template <int K, int NCH, bool STRIDED, bool AFFINE>
__global__ __launch_bounds__(256, 2) void conv1x1_cstat_kernel(AsArgs a) {
  constexpr int TM = 7, NTW = 2, BM = 16 * TM, CW = 64 * NTW;
  constexpr int PIX = 2 * K + 32, KS = K / 32, CH8 = K / 8;
  constexpr int RPP = 256 / CH8, NL = BM / RPP;                      // rows per loader pass, loads per thread
  constexpr int WR = 4;
  constexpr int TOT = NCH * KS;
  static_assert(K <= 256, "one producer channel per thread in the prologue");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sstat = reinterpret_cast<float*>(smem + BM * PIX);          // [2][CW NCH] statistics of this part
  float* coef = sstat + 2 * CW * NCH;                                // [2][K] producer's scale / shift

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bm, bq;
  {   // the parts of a row block are neighbours on one XCD (its rows: L2 hits)
    const int nblk = gridDim.x, id = blockIdx.x, xcd = id & 7, q = nblk >> 3, r = nblk & 7;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    bm = lid / a.nq; bq = lid - bm * a.nq;
  }
  const int ch0 = bq * NCH;                                          // first 128-channel chunk of this workgroup
  const int r16 = lane & 15, q4 = lane >> 4;
  const u32x4* wl = reinterpret_cast<const u32x4*>(a.w) + lane;
  auto wfrag = [&](int g, int j) { return wl[((size_t)(((ch0 + g / KS) * 4 + wid) * NTW + j) * KS + g % KS) * 64]; };
#define CS_STAMP(i) do { if (a.stamps && lane == 0) a.stamps[((size_t)blockIdx.x * 4 + wid) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
  CS_STAMP(0);
  if (a.stamps && lane == 0) a.stamps[((size_t)blockIdx.x * 4 + wid) * 8 + 6] = __builtin_amdgcn_s_memrealtime();
  StatHead shd;
  if (a.in_stats) stat_head_issue(shd, a.in_stats, a.in_srep, K, tid % K, a.in_gamma, a.in_beta);
  u32x4 wq[WR][NTW];
#pragma unroll
  for (int g = 0; g < WR; ++g)
#pragma unroll
    for (int j = 0; j < NTW; ++j) wq[g][j] = wfrag(g, j);

  // ---- fill: 112 rows x K, every load in flight before the first LDS write (rows past M re-read row M - 1; never counted) ----
  {
    const int cch = tid % CH8, lrow = tid / CH8;
    u32x4 v[NL];
    if constexpr (STRIDED) {   // output row (b, ho, wo) reads input pixel (b, ho * stride, wo * stride): conv1x1_astat_kernel's walk
      int m0 = bm * BM + lrow;
      m0 = m0 < a.M ? m0 : a.M - 1;
      const int hw = a.Ho * a.Wo;
      int b = m0 / hw, rem = m0 - b * hw, ho = rem / a.Wo, wo = rem - ho * a.Wo, m = m0;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const long src = ((long)b * a.Hin + ho * a.stride) * a.Win + wo * a.stride;
        v[i] = *reinterpret_cast<const u32x4*>(a.x + src * K + cch * 8);
        if (m + RPP < a.M) {
          m += RPP; wo += RPP;
          while (wo >= a.Wo) { wo -= a.Wo; if (++ho == a.Ho) { ho = 0; ++b; } }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        int m = bm * BM + lrow + i * RPP;
        m = m < a.M ? m : a.M - 1;
        v[i] = *reinterpret_cast<const u32x4*>(a.x + (size_t)m * K + cch * 8);
      }
    }
    if (a.in_stats) {
      if (tid < K) stat_head_finish(shd, a.in_stats, a.in_srep, K, tid, 1.0f / a.in_count, a.in_eps, coef[tid], coef[K + tid]);
      __syncthreads();
      float sc[8], sh[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { sc[e] = coef[cch * 8 + e]; sh[e] = coef[K + cch * 8 + e]; }
#pragma unroll
      for (int i = 0; i < NL; ++i) bn_relu_chunk(v[i], sc, sh);
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) *reinterpret_cast<u32x4*>(smem + (lrow + i * RPP) * PIX + cch * 16) = v[i];
  }
  __syncthreads();
  CS_STAMP(1);

  // ---- channel walk: no barrier until the flush ----------------------------------------------------------------------------
  constexpr int NC = 4 * NTW;
  const char* abase = smem + r16 * PIX + q4 * 16;
  auto read_a = [&](u32x4 (&f)[TM], int ks) {
#pragma unroll
    for (int i = 0; i < TM; ++i) f[i] = *reinterpret_cast<const u32x4*>(abase + i * 16 * PIX + ks * 64);
  };
  f32x4 acc[TM][NTW];
  u32x4 fa0[TM], fa1[TM];
  read_a(fa0, 0);
#pragma clang loop unroll(full)
  for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma clang loop unroll(full)
    for (int ks = 0; ks < KS; ++ks) {
      const int g = ch * KS + ks;
      u32x4 (&fa)[TM] = (g & 1) ? fa1 : fa0;
      u32x4 (&fn)[TM] = (g & 1) ? fa0 : fa1;
      if (g + 1 < TOT) read_a(fn, (ks + 1) % KS);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = mfma_bf16(wq[g % WR][j], fa[i], acc[i][j]);
      if (g + WR < TOT) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) wq[g % WR][j] = wfrag(g + WR, j);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ch == 0) CS_STAMP(2);
    // the chunk's sums (rows past M left out), in conv1x1_astat_kernel's order: tiles ascending, then the 16 lanes of a row group;
    // y != NULL: the outputs too (eval: scale / shift / ReLU), straight from the accumulators
    const int cb = (ch * 4 + wid) * NTW * 16 + NC * q4;               // this lane's 8 consecutive channels inside the workgroup's part
    float es[NC], ess[NC], scv[NC], shv[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { es[c] = 0.f; ess[c] = 0.f; scv[c] = 1.f; shv[c] = 0.f; }
    if constexpr (AFFINE) {
#pragma unroll
      for (int c = 0; c < NC; ++c) { scv[c] = a.scale[ch0 * CW + cb + c]; shv[c] = a.shift[ch0 * CW + cb + c]; }
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
        if (a.y)
          st_out_store16(a.y, ((long)m * a.N + ch0 * CW + cb) * 2,
                         u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])});
      }
    }
    if (a.stats) {
#pragma unroll
      for (int c = 0; c < NC; ++c) { es[c] = row16_sum_(es[c]); ess[c] = row16_sum_(ess[c]); }
      if (r16 == 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) { sstat[cb + c] = es[c]; sstat[CW * NCH + cb + c] = ess[c]; }
      }
    }
    if (ch == NCH / 2 - 1) CS_STAMP(3);
  }
  CS_STAMP(4);
  if (a.stats) {
    float* sdst = a.stats + (a.srep > 1 ? (size_t)(bm % a.srep) * 2 * a.N : 0);
    __syncthreads();
    for (int t = tid; t < 2 * CW * NCH; t += 256)
      atomicAdd(sdst + (t < CW * NCH ? ch0 * CW + t : a.N + ch0 * CW + t - CW * NCH), sstat[t]);
  }
  CS_STAMP(5);
  if (a.stamps && lane == 0) a.stamps[((size_t)blockIdx.x * 4 + wid) * 8 + 7] = __builtin_amdgcn_s_memrealtime();
#undef CS_STAMP
}

#ifdef ST_EXPERIMENTAL   // measured slower than st_bn_act + st_conv1x1_kstream: kept for A/B runs, not in the product build
// =====================================================================================================================
// conv1 of a Bottleneck FUSED WITH THE PREVIOUS BLOCK'S END: x = relu(bn3(raw3) + identity) is what conv1 (1024 -> 256)
// reads, and in torchvision's graph it is also the next identity.  Here the K-streaming kernel's loader forms x itself --
// it loads the raw conv3 output and the identity, applies bn3 (batch statistics of the raw tensor, replicated, summed in the
// prologue) + add + ReLU on the way to LDS and writes x back once (every activation row belongs to exactly ONE workgroup at
// N = 256, so nothing is recomputed).  The separate normalise pass (read raw + identity, write x: 154 MB per layer3 block at
// B = 128, ~20 us + a launch) disappears, and so does conv1's own read of x.  The transform of slab s+1 is spread over the four
// K-steps of slab s (two 16-byte chunks per K-step and thread), under that slab's MFMAs.
struct KfArgs {
  const bf16_t* raw; const bf16_t* res; bf16_t* xout; const bf16_t* w; bf16_t* y;
  float* stats; int srep;
  const float* f_stats; const float* f_gamma; const float* f_beta; float f_count, f_eps; int f_srep;
  int M;
};

template <int K>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv1x1_kfuse_kernel(KfArgs a) {
  constexpr int TM = 7, NTW = 4, BM = 16 * TM, N = 256;
  constexpr int SLAB = 128, PIX = 2 * SLAB + 32, KSS = SLAB / 32;
  constexpr int NSLAB = K / SLAB, KS = K / 32;
  constexpr int SLAB_BYTES = BM * PIX;
  constexpr int NL = BM * (SLAB / 8) / 256;                            // 7
  constexpr int WR = 6;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* coef = reinterpret_cast<float*>(smem + 2 * SLAB_BYTES);       // [scale(K) | shift(K)] of the previous block's bn3

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bm;
  {
    const int nblk = gridDim.x, id = blockIdx.x, xcd = id & 7, q = nblk >> 3, r = nblk & 7;
    bm = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int r16 = lane & 15, q4 = lane >> 4;
  const int T0 = wid * NTW;
  const u32x4* wl = reinterpret_cast<const u32x4*>(a.w) + lane;

  const int cch = tid & 15, lrow = tid >> 4;
  long roff[NL]; bool rok[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    int m = bm * BM + lrow + 16 * i;
    rok[i] = m < a.M;
    m = rok[i] ? m : a.M - 1;
    roff[i] = (long)m * K + cch * 8;
  }
  u32x4 rr[NL], rs[NL];
  auto gload1 = [&](int i, int slab) {
    const int sl = slab < NSLAB ? slab : NSLAB - 1;
    rr[i] = *reinterpret_cast<const u32x4*>(a.raw + roff[i] + sl * SLAB);
    rs[i] = *reinterpret_cast<const u32x4*>(a.res + roff[i] + sl * SLAB);
  };
  // x = relu(raw * sc + sh + res) for chunk i of `slab`, to the LDS ring half `buf` and (once) back to memory
  auto xform1 = [&](int i, int slab, int buf) {
    const float* cs = coef + slab * SLAB + cch * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; e += 4) {
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(cs + e), h4 = *reinterpret_cast<const f32x4*>(cs + K + e);
#pragma unroll
      for (int q = 0; q < 4; ++q) { sc[e + q] = s4[q]; sh[e + q] = h4[q]; }
    }
    u32x4 v;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const float lo = fmaxf(__uint_as_float(rr[i][d] << 16) * sc[2 * d] + sh[2 * d] + __uint_as_float(rs[i][d] << 16), 0.f);
      const float hi = fmaxf(__uint_as_float(rr[i][d] & 0xffff0000u) * sc[2 * d + 1] + sh[2 * d + 1] + __uint_as_float(rs[i][d] & 0xffff0000u), 0.f);
      v[d] = pack_bf16x2(lo, hi);
    }
    *reinterpret_cast<u32x4*>(smem + buf * SLAB_BYTES + (lrow + 16 * i) * PIX + cch * 16) = v;
    if (rok[i]) *reinterpret_cast<u32x4*>(a.xout + roff[i] + slab * SLAB) = v;
  };

#pragma unroll
  for (int i = 0; i < NL; ++i) gload1(i, 0);
  u32x4 wq[WR][NTW];
#pragma unroll
  for (int s = 0; s < WR; ++s)
#pragma unroll
    for (int j = 0; j < NTW; ++j) wq[s][j] = wl[((size_t)(T0 + j) * KS + s) * 64];
  {
    const float inv = 1.0f / a.f_count;
    for (int c = tid; c < K; c += 256) {
      float sm, sq;
      stat_sum(a.f_stats, a.f_srep, K, c, sm, sq);
      bn_scale_shift(sm, sq, inv, a.f_gamma[c], a.f_beta[c], a.f_eps, coef[c], coef[K + c]);
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NL; ++i) { xform1(i, 0, 0); gload1(i, 1); }
  __syncthreads();

  f32x4 acc[TM][NTW];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const char* abase = smem + r16 * PIX + q4 * 16;

  u32x4 fa0[TM], fa1[TM];
#pragma unroll
  for (int s = 0; s < NSLAB; ++s) {
    const char* ab = abase + (s & 1) * SLAB_BYTES;
    auto read_a = [&](u32x4 (&f)[TM], int kk) {
#pragma unroll
      for (int i = 0; i < TM; ++i) f[i] = *reinterpret_cast<const u32x4*>(ab + i * 16 * PIX + kk * 64);
    };
    read_a(fa0, 0);
#pragma unroll
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
        for (int j = 0; j < NTW; ++j) wq[ks % WR][j] = wl[((size_t)(T0 + j) * KS + ks + WR) * 64];
      }
      if (s + 1 < NSLAB) {                              // this K-step's share of the next slab: chunks 2kk, 2kk+1 (7 over 4 K-steps)
#pragma unroll
        for (int i = 2 * kk; i < 2 * kk + 2 && i < NL; ++i) { xform1(i, s + 1, (s + 1) & 1); gload1(i, s + 2); }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }

  constexpr int NC = 4 * NTW;
  const int cb = T0 * 16 + NC * q4;
  float es[NC], ess[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) { es[c] = 0.f; ess[c] = 0.f; }
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
      bf16_t* dst = a.y + (size_t)m * N + cb;
#pragma unroll
      for (int h = 0; h < NTW / 2; ++h)
        *reinterpret_cast<u32x4*>(dst + 8 * h) = u32x4{pack_bf16x2(v[8 * h], v[8 * h + 1]), pack_bf16x2(v[8 * h + 2], v[8 * h + 3]),
                                                       pack_bf16x2(v[8 * h + 4], v[8 * h + 5]), pack_bf16x2(v[8 * h + 6], v[8 * h + 7])};
    }
  }
  if (a.stats) {
    float* sdst = a.stats + (a.srep > 1 ? (size_t)(bm % a.srep) * 2 * N : 0);
#pragma unroll
    for (int c = 0; c < NC; ++c) { es[c] = row16_sum_(es[c]); ess[c] = row16_sum_(ess[c]); }
    block_stats_flush<NC, 64 * NTW>(es, ess, r16 == 0, wid * 16 * NTW + NC * q4, smem, sdst, 0, N, tid);
  }
}

#endif  // ST_EXPERIMENTAL

// ---- fragment-major filter bank ---------------------------------------------------------------------------------
// out[((T * KS + ks) * 64 + lane) * 8 + j], T = 16-channel tile, ks = 32-deep K step (k = tap * Cin + c), lane = (q4, r16):
// MFMA row r16 of tile T is output channel ch(T, r16) = (T / NTW) * 16 NTW + 4 NTW (r16 / 4) + 4 (T % NTW) + r16 % 4
// (so that a lane's accumulators of a wave's NTW tiles are 4 NTW consecutive channels), k = 32 ks + 8 q4 + j.
__global__ void pack_frag_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout, int Cin, int KH, int KW, int ntw, int halves) {
  const long total = (long)Cout * KH * KW * Cin;
  const int KS = KH * KW * Cin / 32;
  const int CH = Cin / halves;                      // halves = 2: K runs (channel half, tap, channel inside the half)
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int j = idx & 7, lane = (idx >> 3) & 63;
    const long f = idx >> 9;
    const int ks = (int)(f % KS), T = (int)(f / KS);
    const int r16 = lane & 15, q4 = lane >> 4;
    const int ch = (T / ntw) * 16 * ntw + 4 * ntw * (r16 >> 2) + 4 * (T % ntw) + (r16 & 3);
    const int kk = 32 * ks + 8 * q4 + j;            // position in the kernel's K order
    const int half = kk / (KH * KW * CH), rem = kk - half * KH * KW * CH;
    const int tap = rem / CH, c = half * CH + rem - tap * CH;
    const int kh = tap / KW, kw = tap - kh * KW;
    out[idx] = (bf16_t)w[(((long)ch * Cin + c) * KH + kh) * KW + kw];
  }
}

struct ImgCfg { int tm, ntw; };
// the instantiations: (C, TM, NTW)
inline bool img_cfg(int C, int N, ImgCfg* c) {
  switch (C) {
    case 64:  *c = ImgCfg{16, 2}; return N % 64 == 0;    // 2 x 2 waves (position halves x 32-channel halves): conv3x3_img_body MS = 2
    case 128: *c = ImgCfg{14, 2}; return N % 128 == 0;
    case 256: *c = ImgCfg{14, 2}; return N % 128 == 0;
    case 512: *c = ImgCfg{4, 2};  return N % 128 == 0;
  }
  return false;
}

template <int C, int TM, int NTW, int HALVES, bool AFFINE>
int launch_img_(const ImgArgs& a, int lds, hipStream_t st, double flops) {
  static int attr_set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  constexpr bool OCC2 = C == 64 || (C == 256 && TM == 7);           // two workgroups per CU: the 64-channel form, the half-band 256-channel form
  if (dev >= 0 && dev < 64 && attr_set[dev] < lds) {
    if constexpr (OCC2) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_img_kernel_occ2<C, TM, NTW, HALVES, AFFINE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    else (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_img_kernel<C, TM, NTW, HALVES, AFFINE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set[dev] = 160 * 1024;
  }
  StProfScope prof(C == 64 ? 8 : C == 128 ? 9 : C == 256 ? 10 : 11, flops, st);
  if constexpr (OCC2) hipLaunchKernelGGL((conv3x3_img_kernel_occ2<C, TM, NTW, HALVES, AFFINE>), dim3(a.B * a.bands * a.nbn), dim3(256), lds, st, a);
  else hipLaunchKernelGGL((conv3x3_img_kernel<C, TM, NTW, HALVES, AFFINE>), dim3(a.B * a.bands * a.nbn), dim3(256), lds, st, a);
  prof.end(st);
  ST_LAUNCH_CHECK();
  return 0;
}

template <int C, int TM, int NTW, int HALVES>
int launch_img(const ImgArgs& a, int lds, hipStream_t st, double flops) {
  return a.scale ? launch_img_<C, TM, NTW, HALVES, true>(a, lds, st, flops) : launch_img_<C, TM, NTW, HALVES, false>(a, lds, st, flops);
}
}  // namespace

// > 0: the NTW (16-channel tiles per wave) of the fragment-major weight layout this geometry runs with; 0: not supported
extern "C" int st_conv3x3_img_supported(int H, int W, int C, int N) {
  ImgCfg c;
  if (!img_cfg(C, N, &c)) return 0;
  const int Wp = W + 2;
  if (H < 1 || W < 1 || 16 * c.tm < Wp) return 0;                  // at least one output row per band
  const long lds = (long)(16 * c.tm + 2 * Wp + 2) * (2 * C + 32) + 2 * C * (long)sizeof(float);
  return lds <= 160 * 1024 ? (c.ntw | (C == 256 ? 2 << 8 : 0)) : 0;   // layout code: ntw | K-order halves << 8
}

extern "C" int st_conv3x3_img(const st_conv3x3_img_desc* d, void* stream) {
  ST_CHECK(d && d->x && d->w_frag && d->y, "st_conv3x3_img: null pointer");
  ImgCfg c;
  ST_CHECK(st_conv3x3_img_supported(d->H, d->W, d->C, d->N) > 0 && img_cfg(d->C, d->N, &c),
           "st_conv3x3_img: unsupported geometry H=%d W=%d C=%d N=%d", d->H, d->W, d->C, d->N);
  ST_CHECK(d->B > 0 && (long)d->B * d->H * d->W * (d->C > d->N ? d->C : d->N) < (1L << 31), "st_conv3x3_img: bad batch");
  ST_CHECK((d->scale == nullptr) == (d->shift == nullptr) && (d->scale || !d->relu), "st_conv3x3_img: scale, shift (and relu) go together");
  ST_CHECK(!d->in_stats || (d->in_gamma && d->in_beta && d->in_count > 0.f), "st_conv3x3_img: input transform needs gamma, beta, count");
  ST_CHECK(d->stats_replicas >= 0 && d->stats_replicas <= 1024 && d->in_stats_replicas >= 0 && d->in_stats_replicas <= 1024, "st_conv3x3_img: bad stats_replicas");
  ImgArgs a;
  a.x = reinterpret_cast<const bf16_t*>(d->x); a.w = reinterpret_cast<const bf16_t*>(d->w_frag); a.y = reinterpret_cast<bf16_t*>(d->y);
  a.stats = d->stats; a.srep = d->stats_replicas; a.scale = d->scale; a.shift = d->shift; a.relu = d->relu;
  a.in_stats = d->in_stats; a.in_gamma = d->in_gamma; a.in_beta = d->in_beta; a.in_count = d->in_count; a.in_eps = d->in_eps; a.in_srep = d->in_stats_replicas > 1 ? d->in_stats_replicas : 1;
  a.B = d->B; a.H = d->H; a.W = d->W; a.N = d->N; a.Wp = d->W + 2;
  // C = 256, half-image bands: 7 row tiles per workgroup, 81 KB of LDS and <= 256 registers, TWO workgroups per CU -- one's
  // fill and epilogue run under the other's K loop; each streams the filter slice for half the rows (twice the L2 -> CU filter bytes per CU).
  // Measured (B = 128, 14 x 14): 37.0 -> 33.7 us per launch, wave total 25.2 -> 19.0 us (K loop 17.7 -> 12.4 us per half band with two waves per
  // SIMD filling each other's stalls), train forward 4.19 -> 4.08 ms.  ST_IMG256_HALF=0: the one-workgroup-per-image form (A/B switch)
  static const bool half_env = [] { const char* e = getenv("ST_IMG256_HALF"); return !e || atoi(e) != 0; }();
  const bool half256 = half_env && d->C == 256 && d->W + 2 <= 16 && d->H > 7 * 16 / (d->W + 2);
  if (half256) c.tm = 7;
  a.R = 16 * c.tm / a.Wp; if (a.R > d->H) a.R = d->H;
  a.bands = (d->H + a.R - 1) / a.R;
  a.nbn = d->C == 64 ? d->N / (32 * c.ntw) : d->N / (64 * c.ntw);   // C = 64: two channel waves per workgroup (position split)
  a.stamps = st_debug_stamps_ptr();
  const int lds = (16 * c.tm + 2 * a.Wp + 2) * (2 * d->C + 32) + 2 * d->C * (int)sizeof(float);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const double flops = 2.0 * d->B * d->H * d->W * (double)d->N * 9.0 * d->C;
  switch (d->C) {
    case 64:  return launch_img<64, 16, 2, 1>(a, lds, st, flops);
    case 128: return launch_img<128, 14, 2, 1>(a, lds, st, flops);
    case 256: return half256 ? launch_img<256, 7, 2, 2>(a, lds, st, flops) : launch_img<256, 14, 2, 2>(a, lds, st, flops);
    case 512: return launch_img<512, 4, 2, 1>(a, lds, st, flops);
  }
  st_set_error("st_conv3x3_img: no kernel for C=%d", d->C);
  return 1;
}

namespace {
struct PwCfg { int ntw, tms; };
// instantiations (K, NTW, TMS); the 4-tile-per-wave forms (NTW = 4) halve the number of channel slices (each slice re-reads
// and re-normalises the rows) at twice the registers
inline bool pw_cfg(int K, int N, PwCfg* c) {
  // tuning knob for tools/bench_conv_pw.py: ST_PW_CFG="ntw,tms" forces a form where it is legal
  static const int force = [] { const char* e = getenv("ST_PW_CFG"); int a = 0, b = 0; if (e && sscanf(e, "%d,%d", &a, &b) == 2) return a * 100 + b; return 0; }();
  if (N % 64 || (K != 64 && K != 128 && K != 256 && K != 512)) return false;
  if (force) {
    const int ntw = force / 100, tms = force % 100;
    const bool ok = N % (64 * ntw) == 0 && ((ntw == 1 && (tms == 2 || tms == 4)) || (ntw == 2 && (tms == 2 || tms == 4)) || (ntw == 4 && tms == 4));
    if (ok) { *c = PwCfg{ntw, tms}; return true; }
  }
  // forms picked per layer geometry from tools/sweep_pw.sh on MI355X (B = 128): (channel tiles per wave, row tiles per stage)
  const bool n128 = N % 128 == 0;
  switch (K) {
    case 64:  *c = N == 64 ? PwCfg{1, 4} : PwCfg{2, 4}; return N == 64 || n128;
    case 128: *c = n128 ? PwCfg{2, 2} : PwCfg{1, 2}; return true;
    case 256: *c = n128 ? PwCfg{2, 4} : PwCfg{1, 2}; return true;
    case 512: *c = n128 ? PwCfg{2, 2} : PwCfg{1, 2}; return true;   // (2, 2): half the channel slices (each re-reads the rows): 30.8 -> 28.5, 48.2 -> 44.8 us
  }
  return false;
}

template <int K, int NTW, int TMS, int D, bool STRIDED, bool AFFINE, bool FUSE = false, bool RES = false>
int launch_pw___(PwArgs& a, hipStream_t st, double flops, bool xf) {
  constexpr int SM = 16 * TMS, PIX = 2 * K + 32;
  const int lds = 2 * SM * PIX + (xf ? (FUSE ? 4 : 2) * K * (int)sizeof(float) : 0);
  a.nbn = a.N / (64 * NTW);
  a.nstage = (a.M + SM - 1) / SM;
  static int attr_set[64] = {};
  static int occ_dev[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  if (!attr_set[dev] && lds > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_wreg_kernel<K, NTW, TMS, D, STRIDED, AFFINE, FUSE, RES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set[dev] = 1;
  }
  // workgroups that are resident together (registers AND LDS: the fused loader's two register sets leave 2 per CU where the LDS
  // alone would take 3 -- a grid sized for 3 ran as one full round plus a 36 % one); the row range is cut into that many pieces
  if (!occ_dev[dev]) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&conv1x1_wreg_kernel<K, NTW, TMS, D, STRIDED, AFFINE, FUSE, RES>), 256, 160 * 1024 / 3) != hipSuccess || nb < 1) nb = 1;
    occ_dev[dev] = nb;
  }
  int occ = (160 * 1024) / lds; if (occ > 3) occ = 3; if (occ > occ_dev[dev]) occ = occ_dev[dev]; if (occ < 1) occ = 1;
  int mbs = (256 * occ) / a.nbn; if (mbs < 1) mbs = 1; if (mbs > a.nstage) mbs = a.nstage;
  a.spb = ((a.nstage + mbs - 1) / mbs + D - 1) / D * D;      // a multiple of the prefetch depth (kernel: straight-line unrolled body)
  mbs = (a.nstage + a.spb - 1) / a.spb;
  static const bool interleave = [] { const char* e = getenv("ST_PW_INTERLEAVE"); return !e || atoi(e) != 0; }();   // A/B switch
  a.mbs = interleave ? mbs : 0;
  StProfScope prof(K == 64 ? 12 : K == 128 ? 13 : K == 256 ? 14 : 15, flops, st);
  hipLaunchKernelGGL((conv1x1_wreg_kernel<K, NTW, TMS, D, STRIDED, AFFINE, FUSE, RES>), dim3(mbs * a.nbn), dim3(256), lds, st, a);
  prof.end(st);
  ST_LAUNCH_CHECK();
  return 0;
}
template <int K, int NTW, int TMS, int D, bool STRIDED>
int launch_pw__(PwArgs& a, hipStream_t st, double flops, bool xf) {
  if constexpr (!STRIDED) { if (a.idn) return launch_pw___<K, NTW, TMS, D, false, true, false, true>(a, st, flops, xf); }
  return a.scale ? launch_pw___<K, NTW, TMS, D, STRIDED, true>(a, st, flops, xf) : launch_pw___<K, NTW, TMS, D, STRIDED, false>(a, st, flops, xf);
}

template <int K, int NTW, int TMS, int D>
int launch_pw_(PwArgs& a, hipStream_t st, double flops, bool xf) {
  return a.stride > 1 ? launch_pw__<K, NTW, TMS, D, true>(a, st, flops, xf) : launch_pw__<K, NTW, TMS, D, false>(a, st, flops, xf);
}

template <int K, int NTW, int TMS>
int launch_pw(PwArgs& a, hipStream_t st, double flops, bool xf) {
  static const int depth = [] { const char* e = getenv("ST_PW_DEPTH"); return e ? atoi(e) : 0; }();   // tuning knob: register prefetch depth 2 | 3
  constexpr int NL = 16 * TMS * K / 2048;             // 16-byte loads per thread per stage
  if constexpr (NL <= 8) { if (depth == 3 || (depth == 0 && NL <= 4)) return launch_pw_<K, NTW, TMS, 3>(a, st, flops, xf); }
  return launch_pw_<K, NTW, TMS, 2>(a, st, flops, xf);
}
}  // namespace

// > 0: supported, the value is the `ntw` of the fragment-major weights (st_pack_conv_weight_frag with KH = KW = 1); 0: use st_conv
extern "C" int st_conv1x1_wreg_supported(int K, int N) {
  PwCfg c;
  return pw_cfg(K, N, &c) ? c.ntw : 0;
}

extern "C" int st_conv1x1_wreg(const st_conv1x1_wreg_desc* d, void* stream) {
  ST_CHECK(d && d->x && d->w_frag && (d->y || (d->stats && !d->scale)), "st_conv1x1_wreg: null pointer (y may be NULL only when statistics are asked for)");
  PwCfg c;
  ST_CHECK(pw_cfg(d->C, d->N, &c), "st_conv1x1_wreg: unsupported geometry C=%d N=%d", d->C, d->N);
  ST_CHECK(d->B > 0 && d->Hin > 0 && d->Win > 0 && d->stride >= 1, "st_conv1x1_wreg: bad geometry");
  ST_CHECK((d->scale == nullptr) == (d->shift == nullptr), "st_conv1x1_wreg: scale and shift go together");
  ST_CHECK(d->scale || !d->relu, "st_conv1x1_wreg: relu comes with the eval-mode scale / shift epilogue");
  ST_CHECK(!d->residual || (d->scale && d->stride == 1 && d->y && !d->stats), "st_conv1x1_wreg: the residual epilogue is the eval-mode form (scale / shift given, stride 1, no statistics)");
  ST_CHECK(!d->in_stats || (d->in_gamma && d->in_beta && d->in_count > 0.f), "st_conv1x1_wreg: input transform needs gamma, beta, count");
  ST_CHECK(d->stats_replicas >= 0 && d->stats_replicas <= 1024 && d->in_stats_replicas >= 0 && d->in_stats_replicas <= 1024, "st_conv1x1_wreg: bad stats_replicas");
  PwArgs a{};
  a.x = reinterpret_cast<const bf16_t*>(d->x); a.w = reinterpret_cast<const bf16_t*>(d->w_frag); a.y = reinterpret_cast<bf16_t*>(d->y);
  a.stats = d->stats; a.srep = d->stats_replicas; a.scale = d->scale; a.shift = d->shift; a.relu = d->relu;
  a.in_stats = d->in_stats; a.in_gamma = d->in_gamma; a.in_beta = d->in_beta; a.in_count = d->in_count; a.in_eps = d->in_eps;
  a.in_srep = d->in_stats_replicas > 1 ? d->in_stats_replicas : 1;
  a.Hin = d->Hin; a.Win = d->Win; a.stride = d->stride;
  a.idn = reinterpret_cast<const bf16_t*>(d->residual);
  a.Ho = (d->Hin - 1) / d->stride + 1; a.Wo = (d->Win - 1) / d->stride + 1;
  const long M = (long)d->B * a.Ho * a.Wo;
  ST_CHECK(M * (d->C > d->N ? d->C : d->N) < (1L << 40) && M < (1L << 31) - 4096, "st_conv1x1_wreg: too many rows");
  a.M = (int)M; a.N = d->N;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const double flops = 2.0 * (double)M * d->N * d->C;
  const bool xf = d->in_stats != nullptr;
#define PW_CASE(KK, NT, TS) if (d->C == KK && c.ntw == NT && c.tms == TS) return launch_pw<KK, NT, TS>(a, st, flops, xf)
#define PW_K(KK) PW_CASE(KK, 1, 2); PW_CASE(KK, 1, 4); PW_CASE(KK, 2, 2); PW_CASE(KK, 2, 4); PW_CASE(KK, 4, 4)
  PW_K(64); PW_K(128); PW_K(256); PW_K(512);
  PW_CASE(64, 1, 8);
#undef PW_K
#undef PW_CASE
  st_set_error("st_conv1x1_wreg: no kernel for C=%d ntw=%d", d->C, c.ntw);
  return 1;
}

namespace {
template <int K, bool STRIDED, bool AFFINE, int NTW>
int launch_ks_(KsArgs& a, hipStream_t st, double flops) {
  constexpr int lds = 2 * 112 * (2 * 128 + 32);
  static int attr_set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_kstream_kernel<K, STRIDED, AFFINE, NTW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set[dev] = 1;
  }
  StProfScope prof(K == 1024 ? 16 : 17, flops, st);
  hipLaunchKernelGGL((conv1x1_kstream_kernel<K, STRIDED, AFFINE, NTW>), dim3(((a.M + 111) / 112) * a.nbn), dim3(256), lds, st, a);
  prof.end(st);
  ST_LAUNCH_CHECK();
  return 0;
}
template <int K, int NTW>
int launch_ks(KsArgs& a, hipStream_t st, double flops) {
  if (a.stride > 1) return a.scale ? launch_ks_<K, true, true, NTW>(a, st, flops) : launch_ks_<K, true, false, NTW>(a, st, flops);
  return a.scale ? launch_ks_<K, false, true, NTW>(a, st, flops) : launch_ks_<K, false, false, NTW>(a, st, flops);
}
}  // namespace

namespace {
template <int K, int NCH, bool AFFINE, bool STRIDED, bool RES = false, int NTW = 2>
int launch_as__(AsArgs& a, hipStream_t st, double flops) {
  constexpr int BM = 112;
  constexpr int lds = BM * (2 * K + 32) + 2 * 64 * NTW * NCH * 4 + 2 * K * 4 + (AFFINE ? 2 * 64 * NTW * NCH * 4 : 0);
  static_assert(lds <= 160 * 1024, "activation block does not fit");
  static int attr_set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_astat_kernel<K, NCH, AFFINE, STRIDED, RES, NTW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set[dev] = 1;
  }
  StProfScope prof(K == 256 ? 18 : 19, flops, st);
  hipLaunchKernelGGL((conv1x1_astat_kernel<K, NCH, AFFINE, STRIDED, RES, NTW>), dim3(((a.M + BM - 1) / BM) * a.nq), dim3(256), lds, st, a);
  prof.end(st);
  ST_LAUNCH_CHECK();
  return 0;
}
template <int K, int NCH, bool STRIDED, bool AFFINE>
int launch_cstat_(AsArgs& a, hipStream_t st, double flops) {
  constexpr int lds = 112 * (2 * K + 32) + 2 * 128 * NCH * 4 + 2 * K * 4;
  static_assert(2 * lds <= 160 * 1024, "two workgroups per CU");
  static int attr_set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_cstat_kernel<K, NCH, STRIDED, AFFINE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set[dev] = 1;
  }
  a.nq = a.N / (128 * NCH);
  StProfScope prof(K == 256 ? 18 : 19, flops, st);
  hipLaunchKernelGGL((conv1x1_cstat_kernel<K, NCH, STRIDED, AFFINE>), dim3(((a.M + 111) / 112) * a.nq), dim3(256), lds, st, a);
  prof.end(st);
  ST_LAUNCH_CHECK();
  return 0;
}
template <int K, int NCH>
int launch_cstat(AsArgs& a, hipStream_t st, double flops) {
  if (a.stride > 1) return a.scale ? launch_cstat_<K, NCH, true, true>(a, st, flops) : launch_cstat_<K, NCH, true, false>(a, st, flops);
  return a.scale ? launch_cstat_<K, NCH, false, true>(a, st, flops) : launch_cstat_<K, NCH, false, false>(a, st, flops);
}
template <int K, int NCH, bool STRIDED>
int launch_as_(AsArgs& a, hipStream_t st, double flops) {
  if constexpr (!STRIDED) { if (a.res) return launch_as__<K, NCH, true, false, true>(a, st, flops); }
  return a.scale ? launch_as__<K, NCH, true, STRIDED>(a, st, flops) : launch_as__<K, NCH, false, STRIDED>(a, st, flops);
}
}  // namespace

// 2: supported (ntw of the fragment-major weights); 0: not.  Stride 1: (C, N) in {(256, 1024), (512, 2048)}, conv3 of layer3 / layer4;
// stride 2: (256, 512) and (512, 1024), the downsample convs of layer2 / layer3.
extern "C" int st_conv1x1_astat_supported(int K, int N) {
  return ((K == 256 && (N == 1024 || N == 512)) || (K == 512 && (N == 2048 || N == 1024))) ? 2 : 0;
}

extern "C" int st_conv1x1_astat(const st_conv1x1_wreg_desc* d, void* stream) {
  ST_CHECK(d && d->x && d->w_frag && (d->y || (d->stats && !d->scale)), "st_conv1x1_astat: null pointer (y may be NULL only when statistics are asked for)");
  ST_CHECK(st_conv1x1_astat_supported(d->C, d->N), "st_conv1x1_astat: unsupported geometry C=%d N=%d", d->C, d->N);
  const bool strided = d->stride > 1;
  ST_CHECK(d->B > 0 && d->Hin > 0 && d->Win > 0 && d->stride >= 1, "st_conv1x1_astat: bad geometry");
  ST_CHECK(strided == (d->N == 2 * d->C), "st_conv1x1_astat: C=%d N=%d comes with stride %s", d->C, d->N, d->N == 2 * d->C ? "> 1" : "1");
  ST_CHECK((d->scale == nullptr) == (d->shift == nullptr) && (d->scale || !d->relu), "st_conv1x1_astat: scale, shift (and relu) go together");
  ST_CHECK(!d->residual || (d->scale && !strided && !d->stats), "st_conv1x1_astat: the residual epilogue is the eval-mode form (scale / shift given, stride 1, no statistics)");
  ST_CHECK(!d->in_stats || (d->in_gamma && d->in_beta && d->in_count > 0.f), "st_conv1x1_astat: input transform needs gamma, beta, count");
  ST_CHECK(d->stats_replicas >= 0 && d->stats_replicas <= 1024 && d->in_stats_replicas >= 0 && d->in_stats_replicas <= 1024, "st_conv1x1_astat: bad stats_replicas");
  AsArgs a;
  a.x = reinterpret_cast<const bf16_t*>(d->x); a.w = reinterpret_cast<const bf16_t*>(d->w_frag); a.y = reinterpret_cast<bf16_t*>(d->y);
  a.stats = d->stats; a.srep = d->stats_replicas; a.scale = d->scale; a.shift = d->shift; a.relu = d->relu;
  a.in_stats = d->in_stats; a.in_gamma = d->in_gamma; a.in_beta = d->in_beta; a.in_count = d->in_count; a.in_eps = d->in_eps;
  a.in_srep = d->in_stats_replicas > 1 ? d->in_stats_replicas : 1;
  a.Hin = d->Hin; a.Win = d->Win; a.stride = d->stride;
  a.Ho = (d->Hin - 1) / d->stride + 1; a.Wo = (d->Win - 1) / d->stride + 1;
  const long M = (long)d->B * a.Ho * a.Wo;
  ST_CHECK(M < (1L << 31) - 4096, "st_conv1x1_astat: too many rows");
  a.M = (int)M; a.N = d->N; a.stamps = st_debug_stamps_ptr();
  a.res = reinterpret_cast<const bf16_t*>(d->residual);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const double flops = 2.0 * (double)M * d->N * d->C;
  a.nq = 1;
  // 256 -> 1024, statistics only (train-mode conv3 of the 14 x 14 blocks, recomputed by st_conv_c3c1): 224-row workgroups x two channel parts
  // 256 -> 1024, statistics only (train-mode conv3 of the 14 x 14 blocks, recomputed by st_conv_c3c1): 64 channels per wave and chunk.
  // Measured (tools/as_stamps.py stats): 21.35 -> 20.2 us per launch.  A (224 rows, two channel parts) form that halves the filter bytes
  // per workgroup measured no gain: this pass is not bound by the L2 -> CU path.  ST_ASTAT_NTW4=0: A/B switch
  // (a dedicated kernel with a 224 x 256 tile per workgroup, one accumulator set: inner loop at the MFMA rate (3.4 us per 256-channel chunk),
  // but the fill (4.4 us) and the now unhidden chunk epilogues (2 x 1.6 us) took back what the walk gained: 14.8 against 14.2 us per wave)
  // conv1x1_cstat_kernel: two channel parts per row block, two workgroups per CU (wave total 14.2 -> 11.3 us, train forward 4.16 -> 4.10 ms);
  // four parts (896 workgroups, 1.75 rounds): no gain.  ST_ASTAT_NTW4 = 1: the one-workgroup-per-CU form above, 0: the 32-channel form (A/B)
  static const int ntw4_env = [] { const char* e = getenv("ST_ASTAT_NTW4"); return e ? atoi(e) : 2; }();
  if (d->C == 256 && !strided && !a.y && ntw4_env == 2) return launch_cstat<256, 4>(a, st, flops);
  // the writing forms with 256 input channels and no residual (conv3 of the first 14 x 14 block, the stride-2 downsample conv of layer2): the same
  // kernel with its stores; N = 512: two parts of two chunks.  ST_ASTAT_OCC2=0: A/B switch
  static const bool occ2_env = [] { const char* e = getenv("ST_ASTAT_OCC2"); return !e || atoi(e) != 0; }();
  if (d->C == 256 && a.y && !a.res && occ2_env) return d->N == 1024 ? launch_cstat<256, 4>(a, st, flops) : launch_cstat<256, 2>(a, st, flops);
  if (d->C == 256 && !strided && !a.y && ntw4_env == 1) return launch_as__<256, 4, false, false, false, 4>(a, st, flops);
  if (d->C == 256) return strided ? launch_as_<256, 4, true>(a, st, flops) : launch_as_<256, 8, false>(a, st, flops);
  if (strided) return launch_as_<512, 8, true>(a, st, flops);
  // 512 -> 2048: with <= 128 row blocks the channels are cut in four parts (56 row blocks at 7 x 7, B = 128: 224 workgroups)
  if ((a.M + 111) / 112 <= 128) { a.nq = 4; return launch_as_<512, 4, false>(a, st, flops); }
  return launch_as_<512, 16, false>(a, st, flops);
}

// conv1 fused with the previous block's end; see st_conv1x1_kfuse_desc in the header.  C = 1024 -> 256: the K-streaming form;
// C = 256 / 512: the register-resident-filter kernel with the fused loader (layer1 / layer2, where the pass it removes is HBM time).
namespace {
int kfuse_wreg(const st_conv1x1_kfuse_desc* d, void* stream) {
  PwCfg c;
  ST_CHECK(pw_cfg(d->C, d->N, &c), "st_conv1x1_kfuse: unsupported geometry C=%d N=%d", d->C, d->N);
  PwArgs a{};
  a.x = reinterpret_cast<const bf16_t*>(d->raw); a.res = reinterpret_cast<const bf16_t*>(d->identity); a.xout = reinterpret_cast<bf16_t*>(d->x_out);
  a.w = reinterpret_cast<const bf16_t*>(d->w_frag); a.y = reinterpret_cast<bf16_t*>(d->y);
  a.stats = d->stats; a.srep = d->stats_replicas;
  a.in_stats = d->f_stats; a.in_gamma = d->f_gamma; a.in_beta = d->f_beta; a.in_count = d->f_count; a.in_eps = d->f_eps;
  a.in_srep = d->f_stats_replicas > 1 ? d->f_stats_replicas : 1;
  a.res_stats = d->id_stats; a.res_gamma = d->id_gamma; a.res_beta = d->id_beta; a.res_srep = d->id_stats_replicas > 1 ? d->id_stats_replicas : 1;
  a.M = (int)d->rows; a.N = d->N; a.stride = 1; a.Hin = 1; a.Win = a.M; a.Ho = 1; a.Wo = a.M;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const double flops = 2.0 * (double)d->rows * d->N * d->C;
#define KF_CASE(KK, NT, TS, DD) if (d->C == KK && c.ntw == NT && c.tms == TS) return launch_pw___<KK, NT, TS, DD, false, false, true>(a, st, flops, true)
  KF_CASE(256, 1, 2, 3); KF_CASE(256, 2, 4, 2); KF_CASE(512, 1, 2, 2); KF_CASE(512, 2, 2, 2);
#undef KF_CASE
  st_set_error("st_conv1x1_kfuse: no fused kernel for C=%d N=%d", d->C, d->N);
  return 1;
}
}  // namespace

extern "C" int st_conv1x1_kfuse_supported(int C, int N) {
#ifdef ST_EXPERIMENTAL
  if (C == 1024 && N == 256) return 4;
#endif
  PwCfg c;
  if ((C == 256 || C == 512) && pw_cfg(C, N, &c) && ((C == 256 && ((c.ntw == 1 && c.tms == 2) || (c.ntw == 2 && c.tms == 4))) || (C == 512 && c.ntw <= 2 && c.tms == 2))) return c.ntw;
  return 0;
}

extern "C" int st_conv1x1_kfuse(const st_conv1x1_kfuse_desc* d, void* stream) {
  ST_CHECK(d && d->raw && d->identity && d->x_out && d->w_frag && d->y && d->f_stats && d->f_gamma && d->f_beta, "st_conv1x1_kfuse: null pointer");
  ST_CHECK(d->rows > 0 && d->rows < (1L << 31) - 4096 && d->f_count > 0.f && d->f_stats_replicas >= 0 && d->f_stats_replicas <= 1024 && d->stats_replicas >= 0 && d->stats_replicas <= 1024,
           "st_conv1x1_kfuse: bad rows / count / replicas");
  ST_CHECK(!d->id_stats || (d->id_gamma && d->id_beta && d->id_stats_replicas >= 0 && d->id_stats_replicas <= 1024), "st_conv1x1_kfuse: id_stats comes with id_gamma, id_beta");
  if (d->C == 256 || d->C == 512) return kfuse_wreg(d, stream);
#ifndef ST_EXPERIMENTAL
  st_set_error("st_conv1x1_kfuse: unsupported geometry C=%d N=%d (the 1024-channel form needs a `make EXPERIMENTAL=1` build)", d->C, d->N);
  return 1;
#else
  ST_CHECK(!d->id_stats, "st_conv1x1_kfuse: the 1024-channel form takes a normalised identity");
  ST_CHECK(d->C == 1024 && d->N == 256, "st_conv1x1_kfuse: unsupported geometry C=%d N=%d", d->C, d->N);
  KfArgs a;
  a.raw = reinterpret_cast<const bf16_t*>(d->raw); a.res = reinterpret_cast<const bf16_t*>(d->identity); a.xout = reinterpret_cast<bf16_t*>(d->x_out);
  a.w = reinterpret_cast<const bf16_t*>(d->w_frag); a.y = reinterpret_cast<bf16_t*>(d->y);
  a.stats = d->stats; a.srep = d->stats_replicas;
  a.f_stats = d->f_stats; a.f_gamma = d->f_gamma; a.f_beta = d->f_beta; a.f_count = d->f_count; a.f_eps = d->f_eps;
  a.f_srep = d->f_stats_replicas > 1 ? d->f_stats_replicas : 1;
  a.M = (int)d->rows;
  constexpr int lds = 2 * 112 * (2 * 128 + 32) + 2 * 1024 * 4;
  static int attr_set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_kfuse_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set[dev] = 1;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  StProfScope prof(16, 2.0 * (double)d->rows * 256.0 * 1024.0, st);
  hipLaunchKernelGGL((conv1x1_kfuse_kernel<1024>), dim3((a.M + 111) / 112), dim3(256), lds, st, a);
  prof.end(st);
  ST_LAUNCH_CHECK();
  return 0;
#endif
}

// 4: supported (the `ntw` of the fragment-major weights); 0: use st_conv
// > 0: supported, the value is the `ntw` of the fragment-major weights.  N = 256 (the 14 x 14 conv1s): 4, the packing st_conv_c3c1 indexes; the
// other layers (1024 -> 512 / 2048, 2048 -> 512): 2 -- 128 channels per workgroup, two workgroups per CU.  ST_KSTREAM_NTW2=0: 4 everywhere (A/B)
extern "C" int st_conv1x1_kstream_supported(int K, int N) {
  static const bool ntw2 = [] { const char* e = getenv("ST_KSTREAM_NTW2"); return !e || atoi(e) != 0; }();
  if (!((K == 1024 || K == 2048) && N % 256 == 0)) return 0;
  return (ntw2 && N != 256) ? 2 : 4;
}

extern "C" int st_conv1x1_kstream(const st_conv1x1_wreg_desc* d, void* stream) {
  ST_CHECK(d && d->x && d->w_frag && d->y, "st_conv1x1_kstream: null pointer");
  ST_CHECK(st_conv1x1_kstream_supported(d->C, d->N), "st_conv1x1_kstream: unsupported geometry C=%d N=%d", d->C, d->N);
  ST_CHECK(d->B > 0 && d->Hin > 0 && d->Win > 0 && d->stride >= 1, "st_conv1x1_kstream: bad geometry");
  ST_CHECK((d->scale == nullptr) == (d->shift == nullptr) && (d->scale || !d->relu), "st_conv1x1_kstream: scale, shift (and relu) go together");
  ST_CHECK(!d->residual && !d->in_stats, "st_conv1x1_kstream: no residual / input transform (use st_conv)");
  ST_CHECK(d->stats_replicas >= 0 && d->stats_replicas <= 1024, "st_conv1x1_kstream: bad stats_replicas");
  KsArgs a;
  a.x = reinterpret_cast<const bf16_t*>(d->x); a.w = reinterpret_cast<const bf16_t*>(d->w_frag); a.y = reinterpret_cast<bf16_t*>(d->y);
  a.stats = d->stats; a.srep = d->stats_replicas; a.scale = d->scale; a.shift = d->shift; a.relu = d->relu;
  a.Hin = d->Hin; a.Win = d->Win; a.stride = d->stride;
  a.Ho = (d->Hin - 1) / d->stride + 1; a.Wo = (d->Win - 1) / d->stride + 1;
  const long M = (long)d->B * a.Ho * a.Wo;
  ST_CHECK(M < (1L << 31) - 4096, "st_conv1x1_kstream: too many rows");
  const int ntw = st_conv1x1_kstream_supported(d->C, d->N);
  a.M = (int)M; a.N = d->N; a.nbn = d->N / (64 * ntw); a.stamps = st_debug_stamps_ptr();
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const double flops = 2.0 * (double)M * d->N * d->C;
  if (ntw == 2) return d->C == 1024 ? launch_ks<1024, 2>(a, st, flops) : launch_ks<2048, 2>(a, st, flops);
  return d->C == 1024 ? launch_ks<1024, 4>(a, st, flops) : launch_ks<2048, 4>(a, st, flops);
}

extern "C" int st_pack_conv_weight_frag(const float* w, void* out, int Cout, int Cin, int KH, int KW, int ntw, void* stream) {
  ST_CHECK(w && out, "st_pack_conv_weight_frag: null pointer");
  const int halves = (ntw >> 8) > 0 ? (ntw >> 8) : 1;           // layout code of the *_supported queries: ntw | K-order halves << 8
  ntw &= 0xff;
  ST_CHECK(ntw >= 1 && ntw <= 16 && Cout % (16 * ntw) == 0 && (KH * KW * Cin) % 32 == 0 && Cin % 8 == 0 && (halves == 1 || (halves == 2 && Cin % 64 == 0)),
           "st_pack_conv_weight_frag: Cout=%d must be a multiple of %d and Cin=%d of 8 (K %% 32 == 0)", Cout, 16 * ntw, Cin);
  const long total = (long)Cout * KH * KW * Cin;
  long grid = (total + 255) / 256; if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(pack_frag_kernel, dim3((unsigned)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w,
                     reinterpret_cast<bf16_t*>(out), Cout, Cin, KH, KW, ntw, halves);
  ST_LAUNCH_CHECK();
  return 0;
}
