// conv1 of a layer3 Bottleneck (1 x 1, 1024 -> 256) fused with the PREVIOUS block's end, as a PRODUCER / CONSUMER workgroup
// (gfx950 / MI355X, bf16, train mode; torchvision Bottleneck.forward, reference cnn.py:46):
//
//     x = relu(bn3(raw3) + identity)      written once (the next identity)
//     y = conv1(x)                         (+ its batch statistics)
//
// replaces one st_bn_act pass (read raw3 + identity, write x: 154 MB at B = 128) and st_conv1x1_kstream (read x again: 51 MB).
//
// Why two kinds of waves.  The four-wave form of this fusion (conv1x1_kfuse_kernel<1024> in conv_img.hip) measured 57 us against 49 for
// the two kernels it replaces: its waves multiply AND load AND store, and on this ISA every vector-memory operation of a wave -- the
// x stores included -- retires through ONE in-order counter (vmcnt), so each wait for a filter fragment also waited for the stores
// and the activation loads issued before it.  Here the roles are split across the eight waves of a 512-thread workgroup:
//   * waves 0-3, CONSUMERS: exactly the K loop of conv1x1_kstream_kernel -- activation slab from LDS, filters through a register
//     ring, 112 MFMAs per slab; the only vector-memory operations in their stream are the filter loads, so their waits are exact;
//   * waves 4-7, PRODUCERS: load raw3 / identity two slabs ahead, apply bn3 + add + ReLU, write the bf16 slab to the LDS ring and
//     to memory.  Their loads and stores queue behind each other, which costs nothing: they have a whole slab time per slab.
// One raw s_barrier per slab hands a ring slot over (no fence: a __syncthreads() would drain both roles' memory queues); both roles
// execute the same number of barriers by construction.  Two waves per SIMD: the producers' VALU runs in the issue slots the MFMA
// stream leaves free.
//
// MEASURED (tools/chain_bench.py variant 5, B = 128): 106.7 us per bottleneck block against 108.0 with the separate pass + K-streaming
// kernel and 112.7 with the four-wave fusion: the role split removes the four-wave form's penalty, but the kernel still runs at
// ~3.6 TB/s (166 MB in ~46 us), not at the 5.8 TB/s of st_bn_act -- without its x stores it is 8.6 us faster, i.e. exactly their HBM
// time, so nothing is stalling on them any more; what is left is the K-slab order itself: every 2-KB activation row is fetched as eight
// 256-byte pieces ~5 us apart (two tensors), a pattern DRAM serves far below the rate of st_bn_act's 1-KB-per-instruction sweep.
// Bit-identical to the two-kernel form and kept in the library, NOT routed by st_resnet_forward (a wash).
#include "common.h"
#include "prof.h"
#include <stdlib.h>

namespace {

struct Kf8Args {
  const bf16_t* raw; const bf16_t* res; bf16_t* xout; const bf16_t* w; bf16_t* y;
  float* stats; int srep;
  const float* f_stats; const float* f_gamma; const float* f_beta; float f_count, f_eps; int f_srep;
  int M;
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
__device__ __forceinline__ void lds_done_barrier() {      // this wave's LDS operations are complete; then the workgroup barrier, WITHOUT a memory fence
  __builtin_amdgcn_s_waitcnt(0xC07F);                       // lgkmcnt(0); vmcnt / expcnt: no wait
  __builtin_amdgcn_s_barrier();
}

constexpr int K = 1024, N = 256, TM = 7, NTW = 4, BM = 16 * TM;
constexpr int SLAB = 128, PIX = 2 * SLAB + 32, KSS = SLAB / 32, NSLAB = K / SLAB, KS = K / 32;
constexpr int SLAB_BYTES = BM * PIX;
constexpr int NL = BM * (SLAB / 8) / 256;                   // 7: 16-byte chunks per producer thread per slab and tensor
constexpr int WR = 4;                                       // consumers' filter ring (K-steps in flight): 64 registers
constexpr int KF8_LDS = 2 * SLAB_BYTES + 2 * K * 4;

__global__ __launch_bounds__(512, 1) void conv1x1_kfuse8_kernel(Kf8Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* coef = reinterpret_cast<float*>(smem + 2 * SLAB_BYTES);       // [scale(K) | shift(K)] of the previous block's bn3

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bm;
  {
    const int nblk = gridDim.x, id = blockIdx.x, xcd = id & 7, q = nblk >> 3, r = nblk & 7;
    bm = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  {
    const float inv = 1.0f / a.f_count;
    for (int c = tid; c < K; c += 512) {
      float sm = 0.f, sq = 0.f;
      for (int r = 0; r < a.f_srep; ++r) { sm += a.f_stats[(size_t)r * 2 * K + c]; sq += a.f_stats[(size_t)r * 2 * K + K + c]; }
      bn_scale_shift(sm, sq, inv, a.f_gamma[c], a.f_beta[c], a.f_eps, coef[c], coef[K + c]);
    }
  }
  __syncthreads();

  const int r16 = lane & 15, q4 = lane >> 4;
  constexpr int NC = 4 * NTW;
  float es[NC], ess[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) { es[c] = 0.f; ess[c] = 0.f; }
  const int cw = wid & 3;                                              // consumer index (producers: unused)
  const int cb = cw * NTW * 16 + NC * q4;

  if (wid < 4) {
    // ================= CONSUMERS: the K loop of conv1x1_kstream_kernel =========================================================
    const int T0 = cw * NTW;
    const u32x4* wl = reinterpret_cast<const u32x4*>(a.w) + lane;
    u32x4 wq[WR][NTW];
#pragma unroll
    for (int s = 0; s < WR; ++s)
#pragma unroll
      for (int j = 0; j < NTW; ++j) wq[s][j] = wl[((size_t)(T0 + j) * KS + s) * 64];
    f32x4 acc[TM][NTW];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* abase = smem + r16 * PIX + q4 * 16;
    u32x4 fa0[TM], fa1[TM];
#pragma unroll
    for (int s = 0; s < NSLAB; ++s) {
      lds_done_barrier();                                              // barrier s: ring half s & 1 holds slab s
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
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // epilogue: accumulators -> statistics partials, bf16, 32-byte stores (16 consecutive channels per lane)
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
  } else {
    // ================= PRODUCERS: raw3 / identity -> relu(bn3(raw3) + identity) -> LDS ring + memory ================================
    const int ptid = tid - 256, cch = ptid & 15, lrow = ptid >> 4;
    long roff[NL]; bool rok[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      int m = bm * BM + lrow + 16 * i;
      rok[i] = m < a.M;
      m = rok[i] ? m : a.M - 1;                                        // rows past the end re-read the last row (never stored)
      roff[i] = (long)m * K + cch * 8;
    }
    u32x4 rr[2][NL], rs[2][NL];
    auto gload = [&](int set, int slab) {
      const int sl = slab < NSLAB ? slab : NSLAB - 1;                  // (clamped past the end: the loop body stays straight-line)
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        rr[set][i] = *reinterpret_cast<const u32x4*>(a.raw + roff[i] + sl * SLAB);
        rs[set][i] = *reinterpret_cast<const u32x4*>(a.res + roff[i] + sl * SLAB);
      }
    };
    auto xform = [&](int set, int slab) {                              // -> ring half slab & 1 and x_out
      const float* cs = coef + slab * SLAB + cch * 8;
      float sc[8], sh[8];
#pragma unroll
      for (int e = 0; e < 8; e += 4) {
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(cs + e), h4 = *reinterpret_cast<const f32x4*>(cs + K + e);
#pragma unroll
        for (int q = 0; q < 4; ++q) { sc[e + q] = s4[q]; sh[e + q] = h4[q]; }
      }
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        u32x4 v;
#pragma unroll
        for (int d = 0; d < 4; ++d) {                                  // st_bn_act's arithmetic: one fma, one add, one rounding
          const float lo = fmaxf(__builtin_fmaf(__uint_as_float(rr[set][i][d] << 16), sc[2 * d], sh[2 * d]) + __uint_as_float(rs[set][i][d] << 16), 0.f);
          const float hi = fmaxf(__builtin_fmaf(__uint_as_float(rr[set][i][d] & 0xffff0000u), sc[2 * d + 1], sh[2 * d + 1]) + __uint_as_float(rs[set][i][d] & 0xffff0000u), 0.f);
          v[d] = pack_bf16x2(lo, hi);
        }
        *reinterpret_cast<u32x4*>(smem + (slab & 1) * SLAB_BYTES + (lrow + 16 * i) * PIX + cch * 16) = v;
        if (rok[i]) *reinterpret_cast<u32x4*>(a.xout + roff[i] + slab * SLAB) = v;
      }
    };
    gload(0, 0);
    gload(1, 1);
    xform(0, 0);
    gload(0, 2);
#pragma unroll
    for (int s = 0; s < NSLAB; ++s) {
      lds_done_barrier();                                              // barrier s: slab s is published; the consumers are done with slab s - 1's half
      if (s + 1 < NSLAB) {
        xform((s + 1) & 1, s + 1);                                     // slab s+1 -> the half slab s-1 was read from
        gload((s + 1) & 1, s + 3);                                     // that register set now requests slab s+3
      }
    }
  }

  if (a.stats) {   // consumers' per-channel sums -> LDS -> one replica (full-wave atomics over consecutive channels); every wave joins the barriers
    float* sdst = a.stats + (a.srep > 1 ? (size_t)(bm % a.srep) * 2 * N : 0);
    float* sred = reinterpret_cast<float*>(smem);                      // [sum(256) | sumsq(256)]; the ring is dead
    if (wid < 4) {
#pragma unroll
      for (int c = 0; c < NC; ++c) { es[c] = row16_sum_(es[c]); ess[c] = row16_sum_(ess[c]); }
    }
    __syncthreads();
    if (wid < 4 && r16 == 0) {
#pragma unroll
      for (int c = 0; c < NC; ++c) { sred[cb + c] = es[c]; sred[N + cb + c] = ess[c]; }
    }
    __syncthreads();
    if (tid < 2 * N) atomicAdd(sdst + tid, sred[tid]);
  }
}

}  // namespace

// same descriptor as st_conv1x1_kfuse (C = 1024, N = 256, id_stats NULL); see include/showtell_hip.h
extern "C" int st_conv1x1_kfuse8(const st_conv1x1_kfuse_desc* d, void* stream) {
  ST_CHECK(d && d->raw && d->identity && d->x_out && d->w_frag && d->y && d->f_stats && d->f_gamma && d->f_beta, "st_conv1x1_kfuse8: null pointer");
  ST_CHECK(d->C == K && d->N == N && !d->id_stats, "st_conv1x1_kfuse8: C=%d N=%d (1024 -> 256, normalised identity only)", d->C, d->N);
  ST_CHECK(d->rows > 0 && d->rows < (1L << 31) - 4096 && d->f_count > 0.f && d->f_stats_replicas >= 0 && d->f_stats_replicas <= 1024 &&
           d->stats_replicas >= 0 && d->stats_replicas <= 1024, "st_conv1x1_kfuse8: bad rows / count / replicas");
  ST_CHECK(d->x_out != d->raw && d->x_out != d->identity, "st_conv1x1_kfuse8: x_out must not alias an input");
  Kf8Args a;
  a.raw = reinterpret_cast<const bf16_t*>(d->raw); a.res = reinterpret_cast<const bf16_t*>(d->identity); a.xout = reinterpret_cast<bf16_t*>(d->x_out);
  a.w = reinterpret_cast<const bf16_t*>(d->w_frag); a.y = reinterpret_cast<bf16_t*>(d->y);
  a.stats = d->stats; a.srep = d->stats_replicas;
  a.f_stats = d->f_stats; a.f_gamma = d->f_gamma; a.f_beta = d->f_beta; a.f_count = d->f_count; a.f_eps = d->f_eps;
  a.f_srep = d->f_stats_replicas > 1 ? d->f_stats_replicas : 1;
  a.M = (int)d->rows;
  static int attr_set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_kfuse8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, KF8_LDS);
    attr_set[dev] = 1;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  StProfScope prof(16, 2.0 * (double)d->rows * N * (double)K, st);
  hipLaunchKernelGGL(conv1x1_kfuse8_kernel, dim3((a.M + BM - 1) / BM), dim3(512), KF8_LDS, st, a);
  prof.end(st);
  ST_LAUNCH_CHECK();
  return 0;
}
