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

constexpr int C1 = 64, C2 = 256, SM = 32;                           // conv3 in / out channels, rows per stage
constexpr int ZPIX = 2 * C1 + 32, XPIX = 2 * C2 + 32;               // padded LDS rows (bytes)
constexpr int Z_BYTES = SM * ZPIX, X_BYTES = SM * XPIX;
// LDS: Z[2] | ID[2] | X | coefficient tables (bn2: 2 x 64, bn3: 2 x 256, identity bn: 2 x 256 floats)
constexpr int B2B_LDS = 2 * Z_BYTES + 3 * X_BYTES + (2 * C1 + 4 * C2) * 4;

template <int N3>
__global__ __launch_bounds__(256, N3 == 64 ? 2 : 1) void conv_b2b_kernel(B2bArgs a) {   // N3 = 128: 64 more filter registers, one workgroup per CU
  constexpr int NTW3 = N3 / 64;                                      // conv1: 16-channel tiles per wave (its fragment-major packing)
  constexpr int NC3 = 4 * NTW3;
  constexpr int D = 2;                                               // register sets of prefetch
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* zt = smem;                                                   // [2][32][ZPIX]  relu(bn2(raw2))
  char* idt = smem + 2 * Z_BYTES;                                    // [2][32][XPIX]  identity
  char* xt = idt + 2 * X_BYTES;                                      // [32][XPIX]     x = relu(bn3(raw3) + identity)
  float* coef = reinterpret_cast<float*>(xt + X_BYTES);              // sc2[64] sh2[64] | sc3[256] sh3[256] | scr[256] shr[256]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
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
  // conv3 (ntw = 2 packing, 16 tiles x 2 K-steps): this wave's tiles (sl * 4 + wid) * 2 + j, sl = 0, 1 -> channels (sl * 4 + wid) * 32 + 8 q4 + ..
  const u32x4* w3l = reinterpret_cast<const u32x4*>(a.w3) + lane;
  u32x4 w3f[2][2][2];
#pragma unroll
  for (int sl = 0; sl < 2; ++sl)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) w3f[sl][j][ks] = w3l[((size_t)(((sl * 4 + wid) * 2 + j) * 2 + ks)) * 64];
  // conv1 (ntw = NTW3 packing, N3 / 16 tiles x 8 K-steps): this wave's tiles wid * NTW3 + j
  const u32x4* w1l = reinterpret_cast<const u32x4*>(a.w1) + lane;
  u32x4 w1f[NTW3][8];
#pragma unroll
  for (int j = 0; j < NTW3; ++j)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) w1f[j][ks] = w1l[((size_t)((wid * NTW3 + j) * 8 + ks)) * 64];

  // ---- loader: raw2 chunk (row tid / 8, chunk tid % 8); identity chunks (row tid / 32 + 8 i, chunk tid % 32) ------------------------
  const int zc = tid & 7, zr = tid >> 3, ic = tid & 31, ir = tid >> 5;
  u32x4 pz[D], pi[D][4]; bool okz[D], oki[D][4];
  auto gload = [&](int set, int k) {
    const int g = gs(k);
    {
      int m = g * SM + zr;
      okz[set] = m < a.M && k < a.spb;
      m = m < a.M ? m : a.M - 1;
      pz[set] = *reinterpret_cast<const u32x4*>(a.raw2 + (size_t)m * C1 + zc * 8);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int m = g * SM + ir + 8 * i;
      oki[set][i] = m < a.M && k < a.spb;
      m = m < a.M ? m : a.M - 1;
      pi[set][i] = *reinterpret_cast<const u32x4*>(a.res + (size_t)m * C2 + ic * 8);
    }
  };
  float sc2[8], sh2[8];
  auto lstore = [&](int set, int buf) {
    u32x4 v = pz[set];
#pragma unroll
    for (int d = 0; d < 4; ++d) {                                    // relu(bn2(.)): st_bn_act's arithmetic, one rounding
      const float lo = fmaxf(__uint_as_float(v[d] << 16) * sc2[2 * d] + sh2[2 * d], 0.f);
      const float hi = fmaxf(__uint_as_float(v[d] & 0xffff0000u) * sc2[2 * d + 1] + sh2[2 * d + 1], 0.f);
      v[d] = pack_bf16x2(lo, hi);
    }
    *reinterpret_cast<u32x4*>(zt + buf * Z_BYTES + zr * ZPIX + zc * 16) = v;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(idt + buf * X_BYTES + (ir + 8 * i) * XPIX + ic * 16) = pi[set][i];
  };

  gload(0, 0);
  {   // BatchNorm coefficients (replicated statistics summed here), while the first rows are in flight
    const float inv = 1.0f / a.count;
    for (int c = tid; c < C1; c += 256) {
      float sm = 0.f, sq = 0.f;
      for (int r = 0; r < a.s2rep; ++r) { sm += a.s2[(size_t)r * 2 * C1 + c]; sq += a.s2[(size_t)r * 2 * C1 + C1 + c]; }
      bn_scale_shift(sm, sq, inv, a.g2[c], a.b2[c], a.eps, coef[c], coef[C1 + c]);
    }
    for (int c = tid; c < C2; c += 256) {
      float sm = 0.f, sq = 0.f;
      for (int r = 0; r < a.s3rep; ++r) { sm += a.s3[(size_t)r * 2 * C2 + c]; sq += a.s3[(size_t)r * 2 * C2 + C2 + c]; }
      bn_scale_shift(sm, sq, inv, a.g3[c], a.b3[c], a.eps, coef[2 * C1 + c], coef[2 * C1 + C2 + c]);
      float s2_ = 1.f, h2_ = 0.f;
      if (rbn) {
        float rm = 0.f, rq = 0.f;
        for (int r = 0; r < a.srrep; ++r) { rm += a.sr[(size_t)r * 2 * C2 + c]; rq += a.sr[(size_t)r * 2 * C2 + C2 + c]; }
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
      const char* ib = idt + buf * X_BYTES;
      // ---- GEMM 1: raw3 tile = Z (32 x 64) x W3^T -> this wave's 2 x 32 channels --------------------------------------------------
      f32x4 acc1[2][2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc1[i][sl][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 zf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) zf[i] = *reinterpret_cast<const u32x4*>(zb + (16 * i + r16) * ZPIX + ks * 64 + q4 * 16);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc1[i][sl][j] = mfma_bf16(w3f[sl][j][ks], zf[i], acc1[i][sl][j]);
      }
      // ---- epilogue 1 (accumulator layout: 8 consecutive channels of row 16 i + r16 per slice): round as the stored tensor was,
      //      bn3, + identity (through its own BatchNorm after a downsample conv), ReLU -> x tile ---------------------------------------
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          const int chb = (sl * 4 + wid) * 32 + 8 * q4;
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
        for (int i = 0; i < 4; ++i) {
          const int m = g * SM + ir + 8 * i;
          const u32x4 v = *reinterpret_cast<const u32x4*>(xt + (ir + 8 * i) * XPIX + ic * 16);
          if (m < a.M && k < a.spb) *reinterpret_cast<u32x4*>(a.xout + (size_t)m * C2 + ic * 8) = v;
        }
      }
      // ---- GEMM 2: y tile = x (32 x 256) x W1^T -> this wave's 16 NTW3 channels ---------------------------------------------------
      f32x4 acc3[2][NTW3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTW3; ++j) acc3[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
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
            bf16_t* dst = a.y + (size_t)m * N3 + cb3;
            if constexpr (NTW3 == 1) *reinterpret_cast<u32x2*>(dst) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            else *reinterpret_cast<u32x4*>(dst) = u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
          }
        }
      }
      // ---- next stage: its register set -> the other ring half; that set then requests stage k + 1 + D --------------------------
      lstore((u + 1) % D, buf ^ 1);
      gload((u + 1) % D, k + 1 + D);
      __syncthreads();
    }
  }

  if (a.stats) {
    float* sdst = a.stats + (a.srep > 1 ? (size_t)(mb % a.srep) * 2 * N3 : 0);
#pragma unroll
    for (int c = 0; c < NC3; ++c) { es[c] = row16_sum_(es[c]); ess[c] = row16_sum_(ess[c]); }
    float* sred = reinterpret_cast<float*>(smem);                    // [2][N3]; the stage tiles are dead
    __syncthreads();
    if (r16 == 0) {
#pragma unroll
      for (int c = 0; c < NC3; ++c) { sred[cb3 + c] = es[c]; sred[N3 + cb3 + c] = ess[c]; }
    }
    __syncthreads();
    for (int t = tid; t < 2 * N3; t += 256) atomicAdd(sdst + t, sred[t]);
  }
}

template <int N3>
int launch_b2b(B2bArgs& a, hipStream_t st, double flops) {
  static int attr_set[64] = {}, occ_dev[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_b2b_kernel<N3>), hipFuncAttributeMaxDynamicSharedMemorySize, B2B_LDS);
    attr_set[dev] = 1;
  }
  if (!occ_dev[dev]) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&conv_b2b_kernel<N3>), 256, B2B_LDS) != hipSuccess || nb < 1) nb = 1;
    occ_dev[dev] = nb > 2 ? 2 : nb;
  }
  a.nstage = (a.M + SM - 1) / SM;
  int mbs = 256 * occ_dev[dev]; if (mbs > a.nstage) mbs = a.nstage;
  a.spb = ((a.nstage + mbs - 1) / mbs + 1) / 2 * 2;                  // a multiple of the prefetch depth
  mbs = (a.nstage + a.spb - 1) / a.spb;
  a.mbs = mbs;
  StProfScope prof(21, flops, st);
  hipLaunchKernelGGL((conv_b2b_kernel<N3>), dim3(mbs), dim3(256), B2B_LDS, st, a);
  prof.end(st);
  ST_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int st_conv_b2b_supported(int C1_, int C2_, int N3) { return C1_ == C1 && C2_ == C2 && (N3 == 64 || N3 == 128) ? 1 : 0; }

extern "C" int st_conv_b2b(const st_conv_b2b_desc* d, void* stream) {
  ST_CHECK(d && d->raw2 && d->w3_frag && d->identity && d->x_out && d->w1_frag && d->y, "st_conv_b2b: null pointer");
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
  const double flops = 2.0 * (double)d->rows * ((double)C1 * C2 + (double)C2 * d->N);
  return d->N == 64 ? launch_b2b<64>(a, st, flops) : launch_b2b<128>(a, st, flops);
}
