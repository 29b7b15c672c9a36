// Block boundary of the 56 x 56 Bottlenecks in ONE pass over the wide tensors (gfx950 / MI355X, bf16, train mode):
//
//     raw3  = conv3( relu(bn2(raw2)) )                 1x1, 64 -> 256       (torchvision Bottleneck.forward, reference cnn.py:46)
//     x     = relu( bn3(raw3) + identity )              the block's output, the next identity          -> written once
//     y     = conv1_next( x )                           1x1, 256 -> 64 | 128 (+ its batch statistics)   -> written once
//
// Why.  At 56 x 56 the 256-channel tensors are 205 MB each at B = 128 and every kernel that touches one is HBM time.  Train-mode
// BatchNorm needs the statistics of raw3 before bn3 can be applied, so the two-kernel form writes raw3 (205 MB) and reads it back.
// raw3 itself is cheap to RE-compute (64 -> 256: 13 GFLOP per layer): st_conv1x1_wreg with y == NULL first produces only conv3's
// statistics from the narrow tensor (51 MB read, nothing written); this kernel then recomputes raw3 tile by tile from raw2 in
// registers, rounds it to bf16 exactly as the stored tensor was, applies bn3 + identity + ReLU, writes x, and feeds x to the next
// conv1 through LDS.  Per boundary: 51 (statistics pass) + 51 + 205 read, 205 + 51 written = 563 MB instead of 922 MB.
// Bit-identical to st_conv1x1_wreg -> st_bn_act -> st_conv1x1_wreg (same MFMA order, same rounding points).
//
// Structure: a workgroup (4 waves) walks 32-row stages, interleaved over the grid like st_conv1x1_wreg's.  Both filter banks live in
// registers as MFMA operands (conv3: a wave owns 2 x 32 output channels; conv1: 16 or 32).  Per stage: raw2 (bn2 + ReLU applied) and
// the identity tile come through LDS (two register sets of prefetch, counted waits); GEMM 1 (16 MFMAs per wave) -> epilogue in
// accumulator layout (round, bn3, + identity, ReLU) -> x tile in LDS -> barrier -> GEMM 2 (16 / 32 MFMAs per wave) reads the x
// tile while the same tile leaves for memory as whole 512-byte rows.  Two barriers per stage; the kernel is HBM-bound by design
// (40 KB per stage against 48 MFMAs per wave).
#include "common.h"
#include "prof.h"
#include <stdlib.h>

namespace {

struct B2bArgs {
  const bf16_t* raw2; const bf16_t* w3; const bf16_t* res; bf16_t* xout; const bf16_t* w1; bf16_t* y;
  float* stats; int srep;                                            // conv1's output statistics [srep][2 N3]
  const float* s2; const float* g2; const float* b2; int s2rep;      // bn2: statistics / gamma / beta of raw2 (64 channels)
  const float* s3; const float* g3; const float* b3; int s3rep;      // bn3: of the recomputed conv3 output (256 channels)
  const float* sr; const float* gr; const float* br; int srrep;      // the identity's own BatchNorm (NULL: already normalised)
  float count, eps;
  int M, nstage, spb, mbs;
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

constexpr int SM = 32;                                              // rows per stage
// LDS: Z[2] | ID[2] (the x tile is formed IN PLACE over the identity tile: a lane reads and writes the same 16 bytes) | coefficient tables
// (bn2: 2 x C1, bn3: 2 x C2, identity bn: 2 x C2 floats)
template <int C1, int C2> constexpr int b2b_lds() { return 2 * SM * (2 * C1 + 32) + 2 * SM * (2 * C2 + 32) + (2 * C1 + 4 * C2) * 4; }

// (C1, C2, N3, NW): conv3 C1 -> C2, next conv1 C2 -> N3 (0: none), NW waves per workgroup.  (64, 256, 64 | 128, 4): layer1, two
// workgroups per CU; (128, 512, 0, 8): layer2 -- eight waves share conv3's 128-KB filter bank (64 registers each), so that two
// waves per SIMD overlap each other's phases (as four waves with 128 filter registers each, one wave per SIMD, the same kernel
// took 57 instead of 4x us: every phase of a stage ran alone).
template <int C1, int C2, int N3, int NW>
__global__ __launch_bounds__(64 * NW, (NW == 4 && C2 == 256 && N3 <= 64) ? 2 : 1) void conv_b2b_kernel(B2bArgs a) {
  constexpr int NT = 64 * NW;
  constexpr int ZPIX = 2 * C1 + 32, XPIX = 2 * C2 + 32;              // padded LDS rows (bytes)
  constexpr int Z_BYTES = SM * ZPIX, X_BYTES = SM * XPIX;
  constexpr int KS1 = C1 / 32, KS3 = C2 / 32;
  constexpr int NSL = (C2 / 128) / (NW / 4);                         // conv3 (ntw = 2 packing): 128-channel slices this wave takes a 32-channel part of
  constexpr bool G2 = N3 > 0;                                        // N3 == 0: no second GEMM -- the kernel ends at x (conv3 + bn3 + identity + ReLU)
  constexpr int NTW3 = G2 ? N3 / 64 : 1;                             // conv1: 16-channel tiles per wave (= the ntw of its fragment-major packing)
  static_assert(!G2 || NW == 4, "the second GEMM is split over four waves");
  constexpr int NC3 = 4 * NTW3;
  constexpr int ZCH = C1 / 8, XCH = C2 / 8;                          // 16-byte chunks per row
  constexpr int ZRS = NT / ZCH, IRS = NT / XCH;                      // rows per loader pass
  constexpr int NZ = SM / ZRS, NI = SM / IRS;                        // loads per thread per stage (raw2, identity)
  static_assert(NZ * ZRS == SM && NI * IRS == SM && NSL >= 1 && NTW3 >= 1, "loader / wave split");
  constexpr int D = 2;                                               // register sets of prefetch
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* zt = smem;                                                   // [2][32][ZPIX]  relu(bn2(raw2))
  char* idt = smem + 2 * Z_BYTES;                                    // [2][32][XPIX]  identity, overwritten in place by x = relu(bn3(raw3) + identity)
  float* coef = reinterpret_cast<float*>(idt + 2 * X_BYTES);         // sc2[C1] sh2[C1] | sc3[C2] sh3[C2] | scr[C2] shr[C2]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w4 = wid & 3, whalf = wid >> 2;                          // conv3's packing: four waves share a 128-channel slice
  const int r16 = lane & 15, q4 = lane >> 4;
  int mb;
  {
    const int nblk = gridDim.x, id = blockIdx.x, xcd = id & 7, q = nblk >> 3, r = nblk & 7;
    mb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  auto gs = [&](int k) { return mb + k * a.mbs; };                   // local stage k -> global stage (interleaved over the grid)
  if (gs(0) >= a.nstage) return;
  const bool rbn = a.sr != nullptr;

  // ---- filters -> registers (fragment-major: one coalesced 1-KiB load per MFMA operand) ---------------------------------------
  // conv3 (ntw = 2 packing, C2 / 16 tiles x KS1 K-steps): this wave's tiles (sl * 4 + w4) * 2 + j, sl = whalf NSL + s
  //   -> channels (sl * 4 + w4) * 32 + 8 q4 + 4 j + e
  const u32x4* w3l = reinterpret_cast<const u32x4*>(a.w3) + lane;
  u32x4 w3f[NSL][2][KS1];
#pragma unroll
  for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) w3f[sl][j][ks] = w3l[((size_t)((((whalf * NSL + sl) * 4 + w4) * 2 + j) * KS1 + ks)) * 64];
  // conv1 (ntw = NTW3 packing, N3 / 16 tiles x KS3 K-steps): this wave's tiles wid * NTW3 + j
  const u32x4* w1l = reinterpret_cast<const u32x4*>(a.w1) + lane;
  u32x4 w1f[NTW3][G2 ? KS3 : 1];
  if constexpr (G2) {
#pragma unroll
    for (int j = 0; j < NTW3; ++j)
#pragma unroll
      for (int ks = 0; ks < KS3; ++ks) w1f[j][ks] = w1l[((size_t)((wid * NTW3 + j) * KS3 + ks)) * 64];
  }

  // ---- loader: raw2 chunk (row tid / 8, chunk tid % 8); identity chunks (row tid / 32 + 8 i, chunk tid % 32) ------------------------
  const int zc = tid % ZCH, zr = tid / ZCH, ic = tid % XCH, ir = tid / XCH;
  u32x4 pz[D][NZ], pi[D][NI];
  auto gload = [&](int set, int k) {       // unconditional clamped loads (rows past the end re-read row M - 1; nothing of them is stored)
    const int g = gs(k);
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
      int m = g * SM + zr + ZRS * i;
      m = m < a.M ? m : a.M - 1;
      pz[set][i] = *reinterpret_cast<const u32x4*>(a.raw2 + (size_t)m * C1 + zc * 8);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      int m = g * SM + ir + IRS * i;
      m = m < a.M ? m : a.M - 1;
      pi[set][i] = *reinterpret_cast<const u32x4*>(a.res + (size_t)m * C2 + ic * 8);
    }
  };
  float sc2[8], sh2[8];
  auto lstore = [&](int set, int buf) {
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
      u32x4 v = pz[set][i];
#pragma unroll
      for (int d = 0; d < 4; ++d) {                                  // relu(bn2(.)): st_bn_act's arithmetic, one rounding
        const float lo = fmaxf(__uint_as_float(v[d] << 16) * sc2[2 * d] + sh2[2 * d], 0.f);
        const float hi = fmaxf(__uint_as_float(v[d] & 0xffff0000u) * sc2[2 * d + 1] + sh2[2 * d + 1], 0.f);
        v[d] = pack_bf16x2(lo, hi);
      }
      *reinterpret_cast<u32x4*>(zt + buf * Z_BYTES + (zr + ZRS * i) * ZPIX + zc * 16) = v;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) *reinterpret_cast<u32x4*>(idt + buf * X_BYTES + (ir + IRS * i) * XPIX + ic * 16) = pi[set][i];
  };

  gload(0, 0);
  {   // BatchNorm coefficients (replicated statistics summed here), while the first rows are in flight
    const float inv = 1.0f / a.count;
    for (int c = tid; c < C1; c += NT) {
      float sm = 0.f, sq = 0.f;
      stat_sum(a.s2, a.s2rep, C1, c, sm, sq);
      bn_scale_shift(sm, sq, inv, a.g2[c], a.b2[c], a.eps, coef[c], coef[C1 + c]);
    }
    for (int c = tid; c < C2; c += NT) {
      float sm = 0.f, sq = 0.f;
      stat_sum(a.s3, a.s3rep, C2, c, sm, sq);
      bn_scale_shift(sm, sq, inv, a.g3[c], a.b3[c], a.eps, coef[2 * C1 + c], coef[2 * C1 + C2 + c]);
      float s2_ = 1.f, h2_ = 0.f;
      if (rbn) {
        float rm = 0.f, rq = 0.f;
        stat_sum(a.sr, a.srrep, C2, c, rm, rq);
        bn_scale_shift(rm, rq, inv, a.gr[c], a.br[c], a.eps, s2_, h2_);
      }
      coef[2 * C1 + 2 * C2 + c] = s2_; coef[2 * C1 + 3 * C2 + c] = h2_;
    }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc2[e] = coef[zc * 8 + e]; sh2[e] = coef[C1 + zc * 8 + e]; }
  // (bn3's coefficients of a lane's 2 x 8 accumulator channels are read from the LDS table where they are used: 32 registers fewer,
  // which is what keeps the N = 128 form at two workgroups per CU without spills)
  lstore(0, 0);
  gload(1, 1);
  gload(0, 2);
  __syncthreads();

  float es[NC3], ess[NC3];
#pragma unroll
  for (int c = 0; c < NC3; ++c) { es[c] = 0.f; ess[c] = 0.f; }
  const int cb3 = wid * 16 * NTW3 + NC3 * q4;                        // this lane's conv1 output channels

  // the host makes spb a multiple of D: the D unrolled copies form one straight-line body (counted waits, see st_conv1x1_wreg)
  for (int k0 = 0; k0 < a.spb; k0 += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const int k = k0 + u, buf = u & 1;                             // (D == 2: stage k sits in ring half k & 1 == u)
      const char* zb = zt + buf * Z_BYTES;
      char* ib = idt + buf * X_BYTES;
      char* xt = ib;                                                 // the x tile replaces the identity tile, element for element
      // ---- GEMM 1: raw3 tile = Z (32 x 64) x W3^T -> this wave's 2 x 32 channels --------------------------------------------------
      f32x4 acc1[2][NSL][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc1[i][sl][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) {
        u32x4 zf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) zf[i] = *reinterpret_cast<const u32x4*>(zb + (16 * i + r16) * ZPIX + ks * 64 + q4 * 16);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc1[i][sl][j] = mfma_bf16(w3f[sl][j][ks], zf[i], acc1[i][sl][j]);
      }
      // ---- epilogue 1 (accumulator layout: 8 consecutive channels of row 16 i + r16 per slice): round as the stored tensor was,
      //      bn3, + identity (through its own BatchNorm after a downsample conv), ReLU -> x tile ---------------------------------------
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl) {
          const int chb = ((whalf * NSL + sl) * 4 + w4) * 32 + 8 * q4;
          const u32x4 idv = *reinterpret_cast<const u32x4*>(ib + (16 * i + r16) * XPIX + chb * 2);
          float sc3[8], sh3[8];
          {
            const f32x4* c3 = reinterpret_cast<const f32x4*>(coef + 2 * C1 + chb);
            const f32x4 s0 = c3[0], s1 = c3[1], h0 = c3[C2 / 4], h1 = c3[C2 / 4 + 1];
#pragma unroll
            for (int e = 0; e < 4; ++e) { sc3[e] = s0[e]; sc3[4 + e] = s1[e]; sh3[e] = h0[e]; sh3[4 + e] = h1[e]; }
          }
          u32x4 o;
#pragma unroll
          for (int d = 0; d < 4; ++d) {                              // channels chb + 2 d, + 1: tile j = d / 2, register e = 2 (d % 2), + 1
            const uint32_t rw = pack_bf16x2(acc1[i][sl][d >> 1][2 * (d & 1)], acc1[i][sl][d >> 1][2 * (d & 1) + 1]);
            float rl = __uint_as_float(idv[d] << 16), rh = __uint_as_float(idv[d] & 0xffff0000u);
            if (rbn) {
              const float* cr = coef + 2 * C1 + 2 * C2 + chb + 2 * d;
              rl = __builtin_fmaf(rl, cr[0], cr[C2]); rh = __builtin_fmaf(rh, cr[1], cr[C2 + 1]);
            }
            const float lo = fmaxf(__builtin_fmaf(__uint_as_float(rw << 16), sc3[2 * d], sh3[2 * d]) + rl, 0.f);
            const float hi = fmaxf(__builtin_fmaf(__uint_as_float(rw & 0xffff0000u), sc3[2 * d + 1], sh3[2 * d + 1]) + rh, 0.f);
            o[d] = pack_bf16x2(lo, hi);
          }
          *reinterpret_cast<u32x4*>(xt + (16 * i + r16) * XPIX + chb * 2) = o;
        }
      __syncthreads();                                               // the x tile is complete
      // ---- x leaves for memory as whole rows (the next identity), while GEMM 2 reads the same tile -------------------------------
      {
        const int g = gs(k);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int m = g * SM + ir + IRS * i;
          const u32x4 v = *reinterpret_cast<const u32x4*>(xt + (ir + IRS * i) * XPIX + ic * 16);
          if (m < a.M && k < a.spb) st_out_store16(a.xout, ((long)m * C2 + ic * 8) * 2, v);
        }
      }
      if constexpr (G2) {
      // ---- GEMM 2: y tile = x (32 x 256) x W1^T -> this wave's 16 NTW3 channels ---------------------------------------------------
      f32x4 acc3[2][NTW3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTW3; ++j) acc3[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS3; ++ks) {
        u32x4 xf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) xf[i] = *reinterpret_cast<const u32x4*>(xt + (16 * i + r16) * XPIX + ks * 64 + q4 * 16);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NTW3; ++j) acc3[i][j] = mfma_bf16(w1f[j][ks], xf[i], acc3[i][j]);
      }
      {
        const int g = gs(k);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int m = g * SM + 16 * i + r16;
          if (m < a.M && k < a.spb) {
            float v[NC3];
#pragma unroll
            for (int j = 0; j < NTW3; ++j)
#pragma unroll
              for (int e = 0; e < 4; ++e) v[4 * j + e] = acc3[i][j][e];
#pragma unroll
            for (int c = 0; c < NC3; ++c) { es[c] += v[c]; ess[c] += v[c] * v[c]; }
            const long dst = ((long)m * N3 + cb3) * 2;
            if constexpr (NTW3 == 1) st_out_store8(a.y, dst, u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])});
            else st_out_store16(a.y, dst, u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])});
          }
        }
      }
      }
      // ---- next stage: its register set -> the other ring half; that set then requests stage k + 1 + D --------------------------
      lstore((u + 1) % D, buf ^ 1);
      gload((u + 1) % D, k + 1 + D);
      __syncthreads();
    }
  }

  if (G2 && a.stats) {
    float* sdst = a.stats + (a.srep > 1 ? (size_t)(mb % a.srep) * 2 * N3 : 0);
#pragma unroll
    for (int c = 0; c < NC3; ++c) { es[c] = row16_sum_(es[c]); ess[c] = row16_sum_(ess[c]); }
    float* sred = reinterpret_cast<float*>(smem);                    // [2][N3]; the stage tiles are dead (2 N3 floats <= the Z ring)
    __syncthreads();
    if (r16 == 0) {
#pragma unroll
      for (int c = 0; c < NC3; ++c) { sred[cb3 + c] = es[c]; sred[N3 + cb3 + c] = ess[c]; }
    }
    __syncthreads();
    for (int t = tid; t < 2 * N3; t += NT) atomicAdd(sdst + t, sred[t]);
  }
}

template <int C1, int C2, int N3, int NW>
int launch_b2b(B2bArgs& a, hipStream_t st, double flops) {
  constexpr int lds = b2b_lds<C1, C2>();
  static int attr_set[64] = {}, occ_dev[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_b2b_kernel<C1, C2, N3, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set[dev] = 1;
  }
  if (!occ_dev[dev]) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&conv_b2b_kernel<C1, C2, N3, NW>), 64 * NW, lds) != hipSuccess || nb < 1) nb = 1;
    occ_dev[dev] = nb > 2 ? 2 : nb;
  }
  a.nstage = (a.M + SM - 1) / SM;
  int mbs = 256 * occ_dev[dev]; if (mbs > a.nstage) mbs = a.nstage;
  a.spb = ((a.nstage + mbs - 1) / mbs + 1) / 2 * 2;                  // a multiple of the prefetch depth
  mbs = (a.nstage + a.spb - 1) / a.spb;
  a.mbs = mbs;
  StProfScope prof(21, flops, st);
  hipLaunchKernelGGL((conv_b2b_kernel<C1, C2, N3, NW>), dim3(mbs), dim3(64 * NW), lds, st, a);
  prof.end(st);
  ST_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int st_conv_b2b_supported(int C1, int C2, int N3) {
  return ((C1 == 64 && C2 == 256 && (N3 == 64 || N3 == 128)) || (C1 == 128 && C2 == 512 && N3 == 0)) ? 1 : 0;   // N3 == 0: stops at x_out
}

extern "C" int st_conv_b2b(const st_conv_b2b_desc* d, void* stream) {
  ST_CHECK(d && d->raw2 && d->w3_frag && d->identity && d->x_out && (d->N == 0 || (d->w1_frag && d->y)), "st_conv_b2b: null pointer");
  ST_CHECK(st_conv_b2b_supported(d->C1, d->C2, d->N), "st_conv_b2b: unsupported geometry %d -> %d -> %d", d->C1, d->C2, d->N);
  ST_CHECK(d->bn2_stats && d->bn2_gamma && d->bn2_beta && d->bn3_stats && d->bn3_gamma && d->bn3_beta, "st_conv_b2b: bn2 and bn3 are required");
  ST_CHECK(!d->id_stats || (d->id_gamma && d->id_beta), "st_conv_b2b: id_stats comes with id_gamma, id_beta");
  ST_CHECK(d->rows > 0 && d->rows < (1L << 31) - 4096 && d->count > 0.f, "st_conv_b2b: bad rows / count");
  ST_CHECK(d->x_out != d->identity && d->x_out != d->raw2, "st_conv_b2b: x_out must not alias an input");
  auto rep = [](int r) { return r > 1 ? r : 1; };
  ST_CHECK(d->stats_replicas >= 0 && d->stats_replicas <= 1024 && d->bn2_replicas <= 1024 && d->bn3_replicas <= 1024 && d->id_replicas <= 1024, "st_conv_b2b: bad replicas");
  B2bArgs a{};
  a.raw2 = reinterpret_cast<const bf16_t*>(d->raw2); a.w3 = reinterpret_cast<const bf16_t*>(d->w3_frag); a.res = reinterpret_cast<const bf16_t*>(d->identity);
  a.xout = reinterpret_cast<bf16_t*>(d->x_out); a.w1 = reinterpret_cast<const bf16_t*>(d->w1_frag); a.y = reinterpret_cast<bf16_t*>(d->y);
  a.stats = d->stats; a.srep = d->stats_replicas;
  a.s2 = d->bn2_stats; a.g2 = d->bn2_gamma; a.b2 = d->bn2_beta; a.s2rep = rep(d->bn2_replicas);
  a.s3 = d->bn3_stats; a.g3 = d->bn3_gamma; a.b3 = d->bn3_beta; a.s3rep = rep(d->bn3_replicas);
  a.sr = d->id_stats; a.gr = d->id_gamma; a.br = d->id_beta; a.srrep = rep(d->id_replicas);
  a.count = d->count; a.eps = d->eps; a.M = (int)d->rows;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const double flops = 2.0 * (double)d->rows * ((double)d->C1 * d->C2 + (double)d->C2 * d->N);
  if (d->C1 == 128) return launch_b2b<128, 512, 0, 8>(a, st, flops);
  return d->N == 64 ? launch_b2b<64, 256, 64, 4>(a, st, flops) : launch_b2b<64, 256, 128, 4>(a, st, flops);
}
