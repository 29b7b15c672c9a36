// conv3 of a layer3 Bottleneck, the block's end, and conv1 of the NEXT block in ONE kernel (gfx950 / MI355X, bf16):
//
//     raw3 = conv3( a2 )                     1x1, 256 -> 1024     a2 = relu(bn2(raw2)) (train: applied in the fill) | conv2's eval output
//     x    = relu( bn3(raw3) + identity )     the block's output = the next identity                 -> written once (x_out)
//     y    = conv1_next( x )                  1x1, 1024 -> 256     (+ batch statistics | folded bn1 + ReLU)  -> written once
//
// (torchvision Bottleneck.forward twice over, reference cnn.py:46.)  Why one kernel: at 14 x 14 the three launches it replaces --
// st_conv1x1_astat (30 us), the block-end st_bn_act pass (28 us, pure HBM time: 154 MB) and st_conv1x1_kstream (22 us, bound by the
// L2 -> CU path: it re-reads x) -- spend more time around their MFMAs than in them (profiles/r03a_layer_table_train.csv: 2.1 - 2.8 x
// their floors).  The observation that makes the fusion cheap: conv3's OUTPUT channels are conv1's K dimension.  A workgroup that
// owns 112 rows walks conv3's 1024 output channels in chunks of 128 (the activation-stationary walk of conv1x1_astat_kernel); each
// finished chunk -- bn3, + identity, ReLU, rounded to bf16: 112 x 128 values of x -- is at once (a) stored to x_out and (b) written to
// a two-slot LDS slab where it is one 128-deep K-slab of conv1: acc1[112 x 256] += x_chunk . W1[:, chunk].  x is never read back,
// the 154-MB normalise pass rides under MFMAs, conv1's activations never cross the L2 -> CU path, and three launch boundaries
// become one.
//   Train mode needs bn3's batch statistics BEFORE the walk: they come from a statistics-only pass of conv3 (st_conv1x1_astat with
// y == NULL over the same inputs; conv3 is computed twice, 13 GFLOP); the accumulators are rounded to bf16 before bn3 exactly as
// the stored raw tensor was, so x_out and y are BIT-IDENTICAL to st_conv1x1_astat -> st_bn_act -> st_conv1x1_kstream (tests assert it).
//   Eval mode (folded BatchNorms, no statistics): x = relu(raw3 * scale3 + shift3 + identity) as st_conv1x1_astat's residual
// epilogue forms it, y = relu(conv1(x) * scale1 + shift1): a layer3 block is two launches (conv2, this).
//
// Structure (4 waves, one per SIMD, one workgroup per CU; 224 workgroups at B = 128):
//   fill      112 x 256 rows of a2 -> LDS (padded rows), bn3 scale / shift of all 1024 channels -> LDS
//   per chunk c (straight-line, fully unrolled; ONE barrier per chunk):
//     A(c)    conv3: 8 K-steps x 14 MFMAs per wave (32 channels x 112 rows), filters through a 6-K-step register ring
//     B(c-1)  conv1 partial over slab (c-1): 4 K-steps x 28 MFMAs per wave (64 channels x 112 rows), filters through a 4-K-step ring,
//             with E(c) -- the epilogue of A(c): identity (requested a chunk ahead), bn3, ReLU, bf16, x_out store, slab write --
//             spread two 16-row tiles per K-step under B's MFMAs
//   end       acc1 -> statistics | scale / shift / ReLU -> y
//
// The 28 x 28 Bottlenecks (128 -> 512 -> 128) run the same body with 4 chunks and 80-row workgroups, two per CU (template
// parameters below; the separate path there is st_conv1x1_wreg -> st_bn_act | st_conv_b2b -> st_conv1x1_wreg, and the bit-identity
// holds against it).  The gain is small there (62 - 77 us in the forward against 46 + 31 for the two launches it replaces): at
// 257 MB per launch the kernel is HBM time (43 us at 6 TB/s), not boundary time.
#include "common.h"
#include "prof.h"

namespace {

struct C3Args {
  const bf16_t* x2; const bf16_t* w3; const bf16_t* res; bf16_t* xout; const bf16_t* w1; bf16_t* y;
  float* stats; int srep;                                            // train: statistics of y [srep][2 x 256]
  const float* s2; const float* g2; const float* b2; int s2rep;      // train: bn2 of x2 (NULL: x2 is already normalised)
  const float* s3; const float* g3; const float* b3; int s3rep;      // train: bn3 from batch statistics
  const float* sd; const float* gd; const float* bd; int sdrep;      // train, IDBN: the identity is a RAW downsample-conv output; its BatchNorm
  const float* sc3; const float* sh3;                                // eval: folded bn3
  const float* sc1; const float* sh1; int relu1;                     // eval: folded bn1 of the next block (+ ReLU) on y
  float count, eps;
  int M;
  unsigned long long* stamps;   // debug (tools/c3_stamps.py): per-wave s_memtime at the phase boundaries, normally NULL
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
__device__ __forceinline__ void bn_relu_chunk_(u32x4& v, const float* sc, const float* sh) {   // conv_img.hip's bn_relu_chunk: same arithmetic
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const float lo = fmaxf(__uint_as_float(v[d] << 16) * sc[2 * d] + sh[2 * d], 0.f);
    const float hi = fmaxf(__uint_as_float(v[d] & 0xffff0000u) * sc[2 * d + 1] + sh[2 * d + 1], 0.f);
    v[d] = pack_bf16x2(lo, hi);
  }
}

// Every global access of the kernel goes through a buffer resource (uniform base in SGPRs + ONE 32-bit lane offset + a compile-time
// constant): with plain 64-bit pointers the fully unrolled walk kept ~20 row / fragment pointers alive (40 VGPRs, spilled) and spent
// two VALU instructions of address arithmetic per load beside the MFMAs.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ u32x4 bload16(__amdgpu_buffer_rsrc_t rs, int voff, int coff) {
  return __builtin_amdgcn_raw_buffer_load_b128(rs, voff, coff, 0);
}
__device__ __forceinline__ void bstore16(__amdgpu_buffer_rsrc_t rs, int voff, int coff, const u32x4& v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, coff, ST_STORE_POLICY == 1 ? 16 : 0);
}

constexpr int CW = 128, PIXS = 2 * CW + 32;                          // chunk width (conv3 output channels = conv1 K-slab), padded slab row
constexpr int KSB = CW / 32;                                         // conv1 K-steps per slab (4)
constexpr int NTW3 = 2;                                              // conv3: 16-channel tiles per wave (32 of a chunk's 128)
template <int K3, int N3, int N1, int TM, bool IDBN = false> constexpr int c3_lds() {
  return 16 * TM * (2 * K3 + 32) + 2 * 16 * TM * PIXS + 2 * N3 * 4 + 2 * K3 * 4 + (IDBN ? 2 * N3 * 4 : 0);
}

// (K3, N3, N1): conv3 K3 -> N3, next conv1 N3 -> N1.  (256, 1024, 256): the layer3 blocks; (128, 512, 128): the layer2 blocks.
// TM: 16-row tiles per workgroup; WPE: workgroups per CU the register budget is cut for (waves per SIMD)
// IDBN (train): the block follows a downsample conv -- the identity arrives RAW with its batch statistics and x = relu(bn3(raw3) + bn_d(identity)),
// st_bn_act's res_bn form (same arithmetic: bn_d as one FMA, then the add).  Its scale / shift sit in LDS behind bn2's and are read per tile.
template <int K3, int N3, int N1, int TM, int WPE, bool TRAIN, bool IDBN = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void conv_c3c1_kernel(C3Args a) {
  static_assert(!IDBN || TRAIN, "the identity's BatchNorm is a train-mode form");
  constexpr int BM = 16 * TM, SLAB_BYTES = BM * PIXS;                // rows per workgroup
  constexpr int WR3 = WPE == 1 ? 6 : 4, WR1 = 4;                     // filter rings (K-steps in flight)
  constexpr int PIX3 = 2 * K3 + 32;                                  // padded LDS row of the conv3 input (bytes)
  constexpr int NCHUNK = N3 / CW;                                    // 8 | 4
  constexpr int KS3 = K3 / 32, KS1 = N3 / 32;                        // K-steps: conv3 per chunk, conv1 in all
  constexpr int NTW1 = N1 / 64;                                      // conv1: 16-channel tiles per wave (N1 / 4 channels per wave)
  constexpr int A2_BYTES = BM * PIX3;
  constexpr int NB3 = N3 / 256;                                      // bn3 channels per thread in the prologue (4 | 2)
  static_assert(K3 <= 256 && N3 % 256 == 0 && N1 % 128 == 0 && KS3 >= 4, "geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* slab = smem + A2_BYTES;
  float* coef3 = reinterpret_cast<float*>(smem + A2_BYTES + 2 * SLAB_BYTES);     // [scale(N3) | shift(N3)]
  float* coef2 = coef3 + 2 * N3;                                                // [scale(K3) | shift(K3)] (train)
  float* coefd = coef2 + 2 * K3;                                                // IDBN: [scale(N3) | shift(N3)] of the identity's BatchNorm

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bm;
  {   // workgroups that share an XCD (equal blockIdx % 8) take neighbouring row blocks
    const int nblk = gridDim.x, id = blockIdx.x, xcd = id & 7, q = nblk >> 3, r = nblk & 7;
    bm = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int r16 = lane & 15, q4 = lane >> 4;
#define C3_STAMP(i) do { if (a.stamps && lane == 0) a.stamps[((size_t)blockIdx.x * 4 + wid) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
  C3_STAMP(0);
  if (a.stamps && lane == 0) a.stamps[((size_t)blockIdx.x * 4 + wid) * 16 + 14] = __builtin_amdgcn_s_memrealtime();
  const __amdgpu_buffer_rsrc_t rs_w3 = rsrc_of(a.w3), rs_w1 = rsrc_of(a.w1), rs_res = rsrc_of(a.res), rs_x = rsrc_of(a.xout), rs_y = rsrc_of(a.y);
  // conv3 K-step g of the walk (chunk g / KS3, K-step g % KS3): this wave's tiles (chunk * 4 + wid) * NTW3 + j   (conv1x1_astat_kernel's layout);
  // fragment (T, ks) sits at ((T * KS + ks) * 64 + lane) * 16 bytes
  const int voff3 = (wid * NTW3 * KS3 * 64 + lane) * 16, voff1 = (wid * NTW1 * KS1 * 64 + lane) * 16;
  auto wfrag3 = [&](int g, int j) { return bload16(rs_w3, voff3, ((((g / KS3) * 4) * NTW3 + j) * KS3 + g % KS3) * 1024); };
  // conv1 K-step h (0 .. KS1): this wave's tiles wid * NTW1 + j                                                    (conv1x1_kstream_kernel's layout)
  auto wfrag1 = [&](int h, int j) { return bload16(rs_w1, voff1, (j * KS1 + h) * 1024); };

  // ---- coefficient inputs FIRST: they head the in-order load queue, so consuming them (below, after every tile / filter load has been
  // issued) waits for them alone.  Train: the first four replicas of bn3's statistics for this thread's four channels and of bn2's for
  // its one channel (K3 == blockDim), gamma / beta; all unconditional (a NULL bn2 reads bn3's arrays and is ignored).
  float sa[NB3][4], sb[NB3][4], gg[NB3], bb[NB3], s2a[4], s2b[4], g2v = 1.f, b2v = 0.f;
  float da[IDBN ? NB3 : 1][4], db[IDBN ? NB3 : 1][4], gdv[IDBN ? NB3 : 1], bdv[IDBN ? NB3 : 1];
  const int c2i = tid % K3;                                          // this thread's bn2 channel (K3 <= blockDim)
  if constexpr (TRAIN) {
    const float* s2p = a.s2 ? a.s2 : a.s3;
    const float* g2p = a.s2 ? a.g2 : a.g3;
    const float* b2p = a.s2 ? a.b2 : a.b3;
    const int rep2 = a.s2 ? a.s2rep : 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r3 = q < a.s3rep ? q : a.s3rep - 1, r2 = q < rep2 ? q : rep2 - 1;
#pragma unroll
      for (int k = 0; k < NB3; ++k) { sa[k][q] = a.s3[(size_t)r3 * 2 * N3 + tid + 256 * k]; sb[k][q] = a.s3[(size_t)r3 * 2 * N3 + N3 + tid + 256 * k]; }
      s2a[q] = s2p[(size_t)r2 * 2 * K3 + c2i]; s2b[q] = s2p[(size_t)r2 * 2 * K3 + K3 + c2i];
    }
#pragma unroll
    for (int k = 0; k < NB3; ++k) { gg[k] = a.g3[tid + 256 * k]; bb[k] = a.b3[tid + 256 * k]; }
    g2v = g2p[c2i]; b2v = b2p[c2i];
    if constexpr (IDBN) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rd = q < a.sdrep ? q : a.sdrep - 1;
#pragma unroll
        for (int k = 0; k < NB3; ++k) { da[k][q] = a.sd[(size_t)rd * 2 * N3 + tid + 256 * k]; db[k][q] = a.sd[(size_t)rd * 2 * N3 + N3 + tid + 256 * k]; }
      }
#pragma unroll
      for (int k = 0; k < NB3; ++k) { gdv[k] = a.gd[tid + 256 * k]; bdv[k] = a.bd[tid + 256 * k]; }
    }
  } else {
#pragma unroll
    for (int k = 0; k < NB3; ++k) { gg[k] = a.sc3[tid + 256 * k]; bb[k] = a.sh3[tid + 256 * k]; }
  }

  u32x4 wq3[WR3][NTW3], wq1[WR1][NTW1];
#pragma unroll
  for (int g = 0; g < WR3; ++g)
#pragma unroll
    for (int j = 0; j < NTW3; ++j) wq3[g][j] = wfrag3(g, j);

  // the identity pieces of a chunk, in accumulator layout (8 consecutive channels of row i * 16 + r16), requested one chunk ahead;
  // rows past M re-read row M - 1 (never stored): every load of the kernel is unconditional
  constexpr int NC3 = 4 * NTW3;                                       // 8
  int rowoff[TM]; bool mok[TM];                                       // byte offset of (row, this lane's first channel of a chunk) in identity / x_out
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = bm * BM + i * 16 + r16;
    mok[i] = m < a.M;
    rowoff[i] = ((mok[i] ? m : a.M - 1) * N3 + wid * 32 + NC3 * q4) * 2;
  }
  u32x4 rres[TM];
  auto res_request = [&](int ch, int i) { rres[i] = bload16(rs_res, rowoff[i], ch * CW * 2); };

  // ---- fill: 112 x 256 rows of the conv3 input, every load in flight before the first LDS write ------------------------------------
  {
    constexpr int CH8 = K3 / 8, RPP = 256 / CH8, NL = BM / RPP;      // 32 chunks per row, 8 rows per pass, 14 loads per thread
    const int cch = tid % CH8, lrow = tid / CH8;
    u32x4 v[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      int m = bm * BM + lrow + i * RPP;
      m = m < a.M ? m : a.M - 1;
      v[i] = *reinterpret_cast<const u32x4*>(a.x2 + (size_t)m * K3 + cch * 8);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) res_request(0, i);
#pragma unroll
    for (int h = 0; h < WR1; ++h)
#pragma unroll
      for (int j = 0; j < NTW1; ++j) wq1[h][j] = wfrag1(h, j);
    if constexpr (TRAIN) {
      const float inv = 1.0f / a.count;
      {   // bn3 scale / shift of all 1024 channels (four per thread), replicas added in replica order (as every other consumer does)
#pragma unroll
        for (int k = 0; k < NB3; ++k) {
          float sm = 0.f, sq = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) { sm += q < a.s3rep ? sa[k][q] : 0.f; sq += q < a.s3rep ? sb[k][q] : 0.f; }
          for (int r = 4; r < a.s3rep; ++r) { sm += a.s3[(size_t)r * 2 * N3 + tid + 256 * k]; sq += a.s3[(size_t)r * 2 * N3 + N3 + tid + 256 * k]; }   // (the engine uses <= 4)
          bn_scale_shift(sm, sq, inv, gg[k], bb[k], a.eps, coef3[tid + 256 * k], coef3[N3 + tid + 256 * k]);
        }
      }
      if constexpr (IDBN) {
#pragma unroll
        for (int k = 0; k < NB3; ++k) {
          float sm = 0.f, sq = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) { sm += q < a.sdrep ? da[k][q] : 0.f; sq += q < a.sdrep ? db[k][q] : 0.f; }
          for (int r = 4; r < a.sdrep; ++r) { sm += a.sd[(size_t)r * 2 * N3 + tid + 256 * k]; sq += a.sd[(size_t)r * 2 * N3 + N3 + tid + 256 * k]; }
          bn_scale_shift(sm, sq, inv, gdv[k], bdv[k], a.eps, coefd[tid + 256 * k], coefd[N3 + tid + 256 * k]);
        }
      }
      if (a.s2) {
        {
          float sm = 0.f, sq = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) { sm += q < a.s2rep ? s2a[q] : 0.f; sq += q < a.s2rep ? s2b[q] : 0.f; }
          for (int r = 4; r < a.s2rep; ++r) { sm += a.s2[(size_t)r * 2 * K3 + c2i]; sq += a.s2[(size_t)r * 2 * K3 + K3 + c2i]; }
          if (tid < K3) bn_scale_shift(sm, sq, inv, g2v, b2v, a.eps, coef2[c2i], coef2[K3 + c2i]);
        }
        __syncthreads();
        float sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = coef2[cch * 8 + e]; sh[e] = coef2[K3 + cch * 8 + e]; }
#pragma unroll
        for (int i = 0; i < NL; ++i) bn_relu_chunk_(v[i], sc, sh);
      }
    } else {
#pragma unroll
      for (int k = 0; k < NB3; ++k) { coef3[tid + 256 * k] = gg[k]; coef3[N3 + tid + 256 * k] = bb[k]; }
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) *reinterpret_cast<u32x4*>(smem + (lrow + i * RPP) * PIX3 + cch * 16) = v[i];
  }
  C3_STAMP(1);
  __syncthreads();
  C3_STAMP(2);

  // ---- the walk -----------------------------------------------------------------------------------------------------------------
  const char* abase3 = smem + r16 * PIX3 + q4 * 16;
  const char* abases = slab + r16 * PIXS + q4 * 16;
  auto read_a3 = [&](u32x4 (&f)[TM], int ks) {
#pragma unroll
    for (int i = 0; i < TM; ++i) f[i] = *reinterpret_cast<const u32x4*>(abase3 + i * 16 * PIX3 + ks * 64);
  };
  auto read_a1 = [&](u32x4 (&f)[TM], int buf, int kk) {
#pragma unroll
    for (int i = 0; i < TM; ++i) f[i] = *reinterpret_cast<const u32x4*>(abases + buf * SLAB_BYTES + i * 16 * PIXS + kk * 64);
  };
  f32x4 acc3[TM][NTW3], acc1[TM][NTW1];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < NTW1; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float sc[NC3], sh[NC3];                                             // bn3 scale / shift of this lane's 8 channels of the chunk in E
  auto coef_load = [&](int ch) {
    const float* cs = coef3 + ch * CW + wid * 32 + NC3 * q4;
#pragma unroll
    for (int e = 0; e < NC3; e += 4) {
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(cs + e), h4 = *reinterpret_cast<const f32x4*>(cs + N3 + e);
#pragma unroll
      for (int q = 0; q < 4; ++q) { sc[e + q] = s4[q]; sh[e + q] = h4[q]; }
    }
  };
  // E: tile i of chunk ch -> x (bn3 + identity + ReLU, bf16) -> x_out and the slab; then this tile's identity register requests chunk ch + 1
  // (packed fp32 arithmetic -- v_pk_fma_f32 / v_pk_add_f32, IEEE per component like their scalar forms -- and the ReLU as a packed signed
  // 16-bit max on the rounded pair: 10.5 VALU instructions per channel pair instead of 14; E is VALU-bound under B's MFMAs.
  // relu(round(x)) == round(relu(x)): rounding keeps the sign, a negative bf16 is a negative int16, and -0 becomes +0 as v_max_f32 makes it)
  auto tile_e = [&](int ch, int i) {
    uint32_t ow[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      f32x2_ r = f32x2_{acc3[i][d >> 1][2 * (d & 1)], acc3[i][d >> 1][2 * (d & 1) + 1]};
      if constexpr (TRAIN) {   // the raw tensor the separate path stores is bf16: round here too (bit-identical x)
        const uint32_t p = pack_bf16x2(r[0], r[1]);
        r = f32x2_{__uint_as_float(p << 16), __uint_as_float(p & 0xffff0000u)};
      }
      f32x2_ idv = f32x2_{__uint_as_float(rres[i][d] << 16), __uint_as_float(rres[i][d] & 0xffff0000u)};
      if constexpr (IDBN) {
        const float* cd = coefd + ch * CW + wid * 32 + NC3 * q4 + 2 * d;
        idv = __builtin_elementwise_fma(idv, f32x2_{cd[0], cd[1]}, f32x2_{cd[N3], cd[N3 + 1]});
      }
      const f32x2_ t = __builtin_elementwise_fma(r, f32x2_{sc[2 * d], sc[2 * d + 1]}, f32x2_{sh[2 * d], sh[2 * d + 1]}) + idv;
      const uint32_t q = pack_bf16x2(t[0], t[1]);
      const i16x2_ m = __builtin_elementwise_max(*reinterpret_cast<const i16x2_*>(&q), i16x2_{0, 0});
      ow[d] = *reinterpret_cast<const uint32_t*>(&m);
    }
    const u32x4 o = u32x4{ow[0], ow[1], ow[2], ow[3]};
    *reinterpret_cast<u32x4*>(slab + (ch & 1) * SLAB_BYTES + (i * 16 + r16) * PIXS + (wid * 32 + NC3 * q4) * 2) = o;
    if (mok[i]) bstore16(rs_x, rowoff[i], ch * CW * 2, o);
    if (ch + 1 < NCHUNK) res_request(ch + 1, i);
  };

  // operand double buffer: K-step parity picks the buffer; every iteration runs an even number of K-steps (8 | 8 + 4 | 4), so the
  // parity of a step is the parity of its index inside its phase
  u32x4 fa0[TM], fa1[TM];
  read_a3(fa0, 0);
#pragma clang loop unroll(full)
  for (int c = 0; c <= NCHUNK; ++c) {
    if (c < NCHUNK) {
      // ---- A(c): conv3 chunk c ------------------------------------------------------------------------------------------------
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < NTW3; ++j) acc3[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma clang loop unroll(full)
      for (int ks = 0; ks < KS3; ++ks) {
        const int g = c * KS3 + ks;
        u32x4 (&fa)[TM] = (ks & 1) ? fa1 : fa0;
        u32x4 (&fn)[TM] = (ks & 1) ? fa0 : fa1;
        if (ks + 1 < KS3) read_a3(fn, ks + 1);
        else if (c >= 1) read_a1(fn, (c - 1) & 1, 0);               // B(c-1) comes next: its slab was completed at the last barrier
        else read_a3(fn, 0);                                         // c == 0: A(1) comes next (after E(0) and the barrier; the rows are static)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < NTW3; ++j) acc3[i][j] = mfma_bf16(wq3[g % WR3][j], fa[i], acc3[i][j]);
        if (g + WR3 < NCHUNK * KS3) {
#pragma unroll
          for (int j = 0; j < NTW3; ++j) wq3[g % WR3][j] = wfrag3(g + WR3, j);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (c == 0) {                                                     // no conv1 work yet: E(0) stands alone, once per workgroup
      coef_load(0);
#pragma unroll
      for (int i = 0; i < TM; ++i) tile_e(0, i);
    } else {
      // ---- B(c-1): conv1 partial over slab (c-1) & 1, with E(c) spread under its MFMAs ------------------------------------------
      if (c < NCHUNK) coef_load(c);
#pragma clang loop unroll(full)
      for (int kk = 0; kk < KSB; ++kk) {
        const int h = (c - 1) * KSB + kk;
        u32x4 (&fa)[TM] = (kk & 1) ? fa1 : fa0;
        u32x4 (&fn)[TM] = (kk & 1) ? fa0 : fa1;
        if (kk + 1 < KSB) read_a1(fn, (c - 1) & 1, kk + 1);
        else if (c + 1 < NCHUNK) read_a3(fn, 0);                     // A(c+1) comes next (static rows: safe across the barrier)
        // (c + 1 == NCHUNK: B(NCHUNK-1) comes next and its slab is only complete at the barrier: read after it)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < NTW1; ++j) acc1[i][j] = mfma_bf16(wq1[h % WR1][j], fa[i], acc1[i][j]);
        if (h + WR1 < KS1) {
#pragma unroll
          for (int j = 0; j < NTW1; ++j) wq1[h % WR1][j] = wfrag1(h + WR1, j);
        }
        if (c < NCHUNK) {
          if (2 * kk < TM) tile_e(c, 2 * kk);
          if (2 * kk + 1 < TM) tile_e(c, 2 * kk + 1);
          // E's VALU stream (accumulator reads, bn3, ReLU, packing: ~45 instructions per tile) does not depend on B's MFMAs: ask the
          // scheduler to issue it BETWEEN them (an MFMA holds the issue port for 8 of its 16 cycles: ~2 other instructions fit per MFMA
          // for free).  Left to itself the backend kept source order -- 28 MFMAs, then E -- and the two phases simply added up
          // (stamps: B alone 0.98 us, E alone 1.0 us, B + E 2.2 us per chunk).
#pragma unroll
          for (int q = 0; q < TM * NTW1; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);        // up to four VALU
            if (q < TM) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // the next K-step's operand reads, one per group
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (c < NCHUNK) {
      C3_STAMP(3 + c);                                                // (before the barrier: own work of the iteration)
      __syncthreads();                                                // slab c & 1 is complete; everybody has left slab (c-1) & 1
      if (c + 1 == NCHUNK) read_a1(fa0, c & 1, 0);                   // B(NCHUNK-1) starts right behind this barrier
    }
  }

  C3_STAMP(11);
  // ---- conv1 epilogue: accumulators -> (statistics | scale / shift / ReLU) -> bf16 -> two 16-byte stores per row ----------------------
  constexpr int NC1 = 4 * NTW1;                                       // 16 consecutive channels per lane
  const int cb1 = wid * NTW1 * 16 + NC1 * q4;
  float es[NC1], ess[NC1], scv[NC1], shv[NC1];
#pragma unroll
  for (int c = 0; c < NC1; ++c) { es[c] = 0.f; ess[c] = 0.f; scv[c] = 1.f; shv[c] = 0.f; }
  if constexpr (!TRAIN) {
#pragma unroll
    for (int c = 0; c < NC1; ++c) { scv[c] = a.sc1[cb1 + c]; shv[c] = a.sh1[cb1 + c]; }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if (mok[i]) {
      float v[NC1];
#pragma unroll
      for (int j = 0; j < NTW1; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * j + e] = acc1[i][j][e];
      if constexpr (TRAIN) {
#pragma unroll
        for (int c = 0; c < NC1; ++c) { es[c] += v[c]; ess[c] += v[c] * v[c]; }
      } else {
#pragma unroll
        for (int c = 0; c < NC1; ++c) v[c] = v[c] * scv[c] + shv[c];
        if (a.relu1) {
#pragma unroll
          for (int c = 0; c < NC1; ++c) v[c] = fmaxf(v[c], 0.f);
        }
      }
      const int dst = ((bm * BM + i * 16 + r16) * N1 + cb1) * 2;
#pragma unroll
      for (int h = 0; h < NTW1 / 2; ++h)
        bstore16(rs_y, dst, 16 * h, u32x4{pack_bf16x2(v[8 * h], v[8 * h + 1]), pack_bf16x2(v[8 * h + 2], v[8 * h + 3]),
                                          pack_bf16x2(v[8 * h + 4], v[8 * h + 5]), pack_bf16x2(v[8 * h + 6], v[8 * h + 7])});
    }
  }
  if (TRAIN && a.stats) {
    float* sdst = a.stats + (a.srep > 1 ? (size_t)(bm % a.srep) * 2 * N1 : 0);
#pragma unroll
    for (int c = 0; c < NC1; ++c) { es[c] = row16_sum_(es[c]); ess[c] = row16_sum_(ess[c]); }
    float* sred = reinterpret_cast<float*>(smem);                     // [2][N1]; the conv3 rows are dead
    __syncthreads();
    if (r16 == 0) {
#pragma unroll
      for (int c = 0; c < NC1; ++c) { sred[cb1 + c] = es[c]; sred[N1 + cb1 + c] = ess[c]; }
    }
    __syncthreads();
    for (int tt = tid; tt < 2 * N1; tt += 256) atomicAdd(sdst + tt, sred[tt]);
  }
  C3_STAMP(12);
  if (a.stamps && lane == 0) a.stamps[((size_t)blockIdx.x * 4 + wid) * 16 + 15] = __builtin_amdgcn_s_memrealtime();
#undef C3_STAMP
}

template <int K3, int N3, int N1, int TM, int WPE, bool TRAIN, bool IDBN = false>
int launch_c3c1(C3Args& a, hipStream_t st, double flops) {
  constexpr int lds = c3_lds<K3, N3, N1, TM, IDBN>(), BM = 16 * TM;
  static_assert(lds * WPE <= 160 * 1024, "LDS");
  static int attr_set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_c3c1_kernel<K3, N3, N1, TM, WPE, TRAIN, IDBN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set[dev] = 1;
  }
  StProfScope prof(K3 == 256 ? 22 : 24, flops, st);
  hipLaunchKernelGGL((conv_c3c1_kernel<K3, N3, N1, TM, WPE, TRAIN, IDBN>), dim3((a.M + BM - 1) / BM), dim3(256), lds, st, a);
  prof.end(st);
  ST_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int st_conv_c3c1_supported(int C1, int C2, int N) {
  return ((C1 == 256 && C2 == 1024 && N == 256) || (C1 == 128 && C2 == 512 && N == 128)) ? 1 : 0;
}

extern "C" int st_conv_c3c1(const st_conv_c3c1_desc* d, void* stream) {
  ST_CHECK(d && d->x2 && d->w3_frag && d->identity && d->x_out && d->w1_frag && d->y, "st_conv_c3c1: null pointer");
  ST_CHECK(st_conv_c3c1_supported(d->C1, d->C2, d->N), "st_conv_c3c1: unsupported geometry %d -> %d -> %d", d->C1, d->C2, d->N);
  ST_CHECK(d->rows > 0 && d->rows * (long)d->C2 * 2 < (1L << 31), "st_conv_c3c1: bad row count %ld", d->rows);
  ST_CHECK(d->x_out != d->identity && d->x_out != d->x2 && d->y != d->x2, "st_conv_c3c1: outputs must not alias inputs");
  const bool train = d->bn3_stats != nullptr;
  if (train) {
    ST_CHECK(d->bn3_gamma && d->bn3_beta && d->count > 0.f && !d->scale3 && !d->scale1, "st_conv_c3c1: train mode takes bn3 statistics + gamma / beta (and no folded coefficients)");
    ST_CHECK(!d->bn2_stats || (d->bn2_gamma && d->bn2_beta), "st_conv_c3c1: bn2_stats comes with bn2_gamma, bn2_beta");
    ST_CHECK(!d->id_stats || (d->id_gamma && d->id_beta), "st_conv_c3c1: id_stats comes with id_gamma, id_beta");
  } else {
    ST_CHECK(d->scale3 && d->shift3 && d->scale1 && d->shift1 && !d->bn2_stats && !d->stats && !d->id_stats, "st_conv_c3c1: eval mode takes folded scale / shift for bn3 and the next bn1, no statistics");
  }
  auto rep = [](int r) { return r > 1 ? r : 1; };
  ST_CHECK(d->stats_replicas >= 0 && d->stats_replicas <= 1024 && d->bn2_replicas >= 0 && d->bn2_replicas <= 1024 && d->bn3_replicas >= 0 && d->bn3_replicas <= 1024, "st_conv_c3c1: bad replicas");
  C3Args a{};
  a.x2 = reinterpret_cast<const bf16_t*>(d->x2); a.w3 = reinterpret_cast<const bf16_t*>(d->w3_frag); a.res = reinterpret_cast<const bf16_t*>(d->identity);
  a.xout = reinterpret_cast<bf16_t*>(d->x_out); a.w1 = reinterpret_cast<const bf16_t*>(d->w1_frag); a.y = reinterpret_cast<bf16_t*>(d->y);
  a.stats = d->stats; a.srep = d->stats_replicas;
  a.s2 = d->bn2_stats; a.g2 = d->bn2_gamma; a.b2 = d->bn2_beta; a.s2rep = rep(d->bn2_replicas);
  a.s3 = d->bn3_stats; a.g3 = d->bn3_gamma; a.b3 = d->bn3_beta; a.s3rep = rep(d->bn3_replicas);
  a.sd = d->id_stats; a.gd = d->id_gamma; a.bd = d->id_beta; a.sdrep = rep(d->id_replicas);
  a.sc3 = d->scale3; a.sh3 = d->shift3; a.sc1 = d->scale1; a.sh1 = d->shift1; a.relu1 = d->relu1;
  a.count = d->count; a.eps = d->eps; a.M = (int)d->rows; a.stamps = st_debug_stamps_ptr();
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const double flops = 2.0 * (double)d->rows * ((double)d->C1 * d->C2 + (double)d->C2 * d->N);
  if (d->C1 == 256 && a.sd) return launch_c3c1<256, 1024, 256, 7, 1, true, true>(a, st, flops);
  if (d->C1 == 256) return train ? launch_c3c1<256, 1024, 256, 7, 1, true>(a, st, flops) : launch_c3c1<256, 1024, 256, 7, 1, false>(a, st, flops);
  // 28 x 28: 80-row workgroups, two per CU (74 KB of LDS, <= 256 registers): one workgroup's fill / prologue runs under the other's
  // walk, and 100352 rows make 1255 workgroups = 2.45 rounds of 512 slots.  Measured at B = 128 (tools/time_c3c1.py, train | eval):
  // (TM 7, 1 per CU) 80 | 74 us, (4, 2) 80 | 74 (3.06 rounds), (3, 3) 93 | 110 (spills), (5, 2) 67 | 64.  ST_C3C1_L2=7: the (7, 1) form
  static const int l2form = [] { const char* e = getenv("ST_C3C1_L2"); return e ? atoi(e) : 5; }();
  if (a.sd) return launch_c3c1<128, 512, 128, 5, 2, true, true>(a, st, flops);
  if (l2form == 7) return train ? launch_c3c1<128, 512, 128, 7, 1, true>(a, st, flops) : launch_c3c1<128, 512, 128, 7, 1, false>(a, st, flops);
  return train ? launch_c3c1<128, 512, 128, 5, 2, true>(a, st, flops) : launch_c3c1<128, 512, 128, 5, 2, false>(a, st, flops);
}
