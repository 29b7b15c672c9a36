// Implicit-GEMM convolution / dense projection for gfx950 (MI355X) on MFMA.
//
//   y[m][n] = sum_k x_gather[m][k] * w[n][k]      m = (b,ho,wo)   k = (kh,kw,c)
//
// Replaces the cuDNN convolutions behind torchvision's ResNet (reference cnn.py:46,
// cnn_attn.py:46) and the cuBLAS GEMMs behind nn.Linear / nn.GRU input projections
// (cnn.py:49, rnn.py:32-33).  Not a translation of either: NHWC activations, [N][K]
// weights, one kernel template for 1x1 / 3x3 / 7x7 / plain GEMM.
//
// Structure (CDNA4):
//   * tile rows are 128 bytes of K (64 bf16 / 32 f32) = 8 chunks of 16 B; a row's chunk c
//     is stored at chunk slot c ^ (row & 7)  -> ds_read_b128 of a 16x(K=32) MFMA operand
//     is bank-conflict free (checked against the gfx950 ds_read_b128 lane groups).
//   * global -> registers -> LDS staging (zero fill for padding / ragged edges), the
//     next K-tile's global loads are issued before the MFMAs of the current one and
//     written to the other LDS buffer afterwards: one barrier per K-step.
//   * operands are swapped (weights = MFMA "A", pixels = MFMA "B") so every lane ends
//     with 4 consecutive output channels of one pixel: 8-byte (bf16) / 16-byte (f32)
//     stores and a cheap 16-lane reduction for the batch-norm statistics.
//   * bf16: v_mfma_f32_16x16x32_bf16; f32: 4 x v_mfma_f32_16x16x4_f32 per 16-byte chunk
//     (exact fp32 fma chain; k order inside a chunk is permuted identically for both
//     operands, which leaves the sum unchanged up to association).
//   * blockIdx is remapped so that blocks of one XCD walk neighbouring tiles
//     (pixel tile major), sharing the activation rows in that XCD's L2.
#include "common.h"
#include "prof.h"

#include <vector>
#include <mutex>
#include <stdlib.h>
#include <string.h>

namespace {

// ---- optional launch profiler (bench.py roofline): HIP events around every igemm launch, on the
// ---- launch stream.  Off by default; never touched on the normal path beyond one branch.
constexpr int kVariants = kProfVariants;
struct ProfRec { hipEvent_t e0, e1; int variant; double flops; };
bool g_prof_on = false;
unsigned long long* g_stamps = nullptr;   // st_debug_stamps
std::vector<ProfRec> g_prof;
std::mutex g_prof_mu;
constexpr size_t kProfMax = 1 << 16;

struct IgemmArgs {
  const void* x; const void* w; void* y;
  const float* bias; const float* scale; const float* shift; const void* residual; float* stats;
  int M, N, K;
  int Hin, Win, Cin, Ho, Wo, KH, KW, stride, pad;
  int ldx, ldw, ldy;
  int relu, accumulate, out_f32, korder, srep;
  int atomic;   // split-K member: fp32 output accumulated with atomics
  const float* in_stats; const float* in_gamma; const float* in_beta; float in_inv_count, in_eps;   // input BN+ReLU (XF)
  unsigned long long* stamps;   // debug: per-block phase timestamps (tools/conv_stamps.py), normally NULL
  int nbm, nbn;
  double flops;   // algorithmic 2*M*N*K (host side only, profiler)
};

// Up to kGroup independent problems of identical shape in ONE launch (blockIdx.y = member): the decoder's ten per-layer
// weight-gradient GEMMs are 48 tiles each -- alone they leave 80 % of the CUs idle for 30 us apiece.
constexpr int kGroup = 12;
struct IgemmGroup { IgemmArgs g[kGroup]; };
static_assert(sizeof(IgemmGroup) <= 4000, "kernel-argument segment is 4 KiB");

template <typename T> struct Mfma;
template <> struct Mfma<bf16_t> {
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a),
                                                *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
  }
};
template <> struct Mfma<float> {
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[e]), __uint_as_float(b[e]), c, 0, 0, 0);
  }
};

// ---- epilogue shared by both main loops ---------------------------------------------------------
// On entry every wave has passed a barrier after its last LDS read of the K loop and no LDS-DMA is in flight.
// RAWB: barriers are raw s_barrier + lgkmcnt(0) (for main loops that keep LDS-DMA in flight: __syncthreads() would
// drain it); only waves [SW0, SW0+SNW) touch global memory.  The production main loop uses the defaults.
template <bool RAWB> __device__ __forceinline__ void epi_barrier() {
  if (RAWB) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  } else {
    __syncthreads();
  }
}

// sum over the 16 lanes of a DPP row (lanes with equal lane >> 4): v_add_f32 with row_ror 8/4/2/1, every lane gets the total
template <int CTRL> __device__ __forceinline__ float dpp_rot(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_rot<0x128>(v); v += dpp_rot<0x124>(v); v += dpp_rot<0x122>(v); v += dpp_rot<0x121>(v);
  return v;
}

template <typename T, int BM, int BN, int WM, int WN, bool RAWB = false, int SW0 = 0, int SNW = WM * WN>
__device__ __forceinline__ void igemm_epilogue(const IgemmArgs& a, f32x4 (&acc)[BM / WM / 16][BN / WN / 16], char* smem,
                                               int bm, int bn, int tid) {
  constexpr int NT = 64 * SNW;                       // threads that store
  const int stid = tid - 64 * SW0;                   // store-thread index (negative / >= NT: not a storer)
  const bool storer = stid >= 0 && stid < NT;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid - wm * WN;
  const int r16 = lane & 15, q4 = lane >> 4;
  // lane holds, for tile (i,j): pixel m = m0 + i*16 + r16, channels n = n0 + j*16 + 4*q4 + e
  const int m0 = bm * BM + wm * (BM / WM), n0 = bn * BN + wn * (BN / WN);

  // per-wave statistics partials live BEHIND the staging buffer, so the cross-wave sum shares the first staging barrier
  float* red = reinterpret_cast<float*>(smem + 64 * (BN + 4) * 4);  // [2][BN][WM]
  if (a.stats) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float s[4], ss[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = 0.f, v2 = 0.f;
        const int n = n0 + j * 16 + 4 * q4 + e;
        const float bz = (a.bias && n < a.N) ? a.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int m = m0 + i * 16 + r16;
          const float t = (m < a.M) ? acc[i][j][e] + bz : 0.f;
          v += t; v2 += t * t;
        }
        s[e] = row16_sum(v); ss[e] = row16_sum(v2);   // DPP rotates inside the 16-lane row: plain VALU, no LDS crossbar
      }
      if (r16 == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int nl = wn * (BN / WN) + j * 16 + 4 * q4 + e;
          red[nl * WM + wm] = s[e];
          red[(BN + nl) * WM + wm] = ss[e];
        }
      }
    }
  }

  // ---- staged store: registers -> LDS (fp32, 64-row halves) -> 16-byte coalesced global stores --------
  // Every output row leaves as whole 128-byte lines (8 consecutive channels per lane, 16 lanes per row).
  constexpr int SROW = BN + 4;                       // floats per staged row (+16 B: spreads the banks)
  float* stage = reinterpret_cast<float*>(smem);     // [64][SROW] <= 33.8 KB, K-loop buffers are dead
  constexpr int HALVES = BM / 64;
  const int my_half = (wm * (BM / WM)) / 64;
  const int lrow0 = (wm * (BM / WM)) % 64;           // this wave's first row inside its half
#pragma unroll
  for (int h = 0; h < HALVES; ++h) {
    if (h > 0) epi_barrier<RAWB>();                   // previous half has left the buffer (the K loop ended on a barrier)
    if (my_half == h) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int nl = wn * (BN / WN) + j * 16 + 4 * q4;
          const int n = bn * BN + nl;
          f32x4 v = acc[i][j];
          if (a.bias || a.scale) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (n + e < a.N) {
                if (a.bias) v[e] += a.bias[n + e];
                if (a.scale) v[e] = v[e] * a.scale[n + e] + a.shift[n + e];
              }
            }
          }
          *reinterpret_cast<f32x4*>(stage + (lrow0 + i * 16 + r16) * SROW + nl) = v;
        }
    }
    epi_barrier<RAWB>();
    if (h == 0 && a.stats) {                          // the partials of every wave are in `red` now
      float* sdst = a.stats + (a.srep > 1 ? (size_t)(bm % a.srep) * 2 * a.N : 0);   // replica of this pixel tile
      if (storer) for (int t = stid; t < 2 * BN; t += NT) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) v += red[t * WM + w];
        const int nl = t < BN ? t : t - BN;
        const int n = bn * BN + nl;
        if (n < a.N) atomicAdd(sdst + (t < BN ? n : a.N + n), v);
      }
    }
    if (a.atomic) {   // split-K slice: the slices of one output tile meet in memory; 64 consecutive floats per wave-instruction
      if (storer) for (int idx = stid; idx < 64 * BN; idx += NT) {
        const int r = idx / BN, c = idx - r * BN;
        const int m = bm * BM + h * 64 + r, n = bn * BN + c;
        if (m < a.M && n < a.N) atomicAdd(reinterpret_cast<float*>(a.y) + (long)m * a.ldy + n, stage[r * SROW + c]);
      }
      continue;
    }
    constexpr int CPR = BN / 8;                       // 8-channel chunks per row
    constexpr int RPI = NT / CPR;                     // rows per pass
    const int c8 = (stid % CPR) * 8, rr = stid / CPR;
    const int n = bn * BN + c8;
    if (storer)
#pragma unroll
    for (int r = rr; r < 64; r += RPI) {
      const int m = bm * BM + h * 64 + r;
      if (m >= a.M || n >= a.N) continue;
      float v[8];
      const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + r * SROW + c8);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + r * SROW + c8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
      const long off = (long)m * a.ldy + n;
      const bool full = (n + 7 < a.N) && ((a.ldy & 7) == 0);
      if (a.out_f32) {
        float* Y = reinterpret_cast<float*>(a.y) + off;
        const float* R = a.residual ? reinterpret_cast<const float*>(a.residual) + off : nullptr;
        if (full) {
          if (R) { const f32x4 r0 = *reinterpret_cast<const f32x4*>(R), r1 = *reinterpret_cast<const f32x4*>(R + 4);
                   for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; } }
          if (a.relu) for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
          if (a.accumulate) { const f32x4 o0 = *reinterpret_cast<const f32x4*>(Y), o1 = *reinterpret_cast<const f32x4*>(Y + 4);
                              for (int e = 0; e < 4; ++e) { v[e] += o0[e]; v[4 + e] += o1[e]; } }
          *reinterpret_cast<f32x4*>(Y) = f32x4{v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f32x4*>(Y + 4) = f32x4{v[4], v[5], v[6], v[7]};
        } else {
          for (int e = 0; e < 8 && n + e < a.N; ++e) {
            float t = v[e];
            if (R) t += R[e];
            if (a.relu) t = fmaxf(t, 0.f);
            if (a.accumulate) t += Y[e];
            Y[e] = t;
          }
        }
      } else {
        uint16_t* Y = reinterpret_cast<uint16_t*>(a.y) + off;
        const uint16_t* R = a.residual ? reinterpret_cast<const uint16_t*>(a.residual) + off : nullptr;
        if (full) {
          if (R) {
            const u32x4 r4 = *reinterpret_cast<const u32x4*>(R);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] += __uint_as_float(r4[e] << 16); v[2 * e + 1] += __uint_as_float(r4[e] & 0xffff0000u); }
          }
          if (a.relu) for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
          if (a.accumulate) {
            const u32x4 o4 = *reinterpret_cast<const u32x4*>(Y);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] += __uint_as_float(o4[e] << 16); v[2 * e + 1] += __uint_as_float(o4[e] & 0xffff0000u); }
          }
          *reinterpret_cast<u32x4*>(Y) = u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
        } else {
          for (int e = 0; e < 8 && n + e < a.N; ++e) {
            float t = v[e];
            if (R) t += bf16_bits_to_f32(R[e]);
            if (a.relu) t = fmaxf(t, 0.f);
            if (a.accumulate) t += bf16_bits_to_f32(Y[e]);
            Y[e] = f32_to_bf16_bits(t);
          }
        }
      }
    }
  }
}

// chunk slot of (row, chunk) inside a tile row of KC 16-byte chunks; both make the ds_read_b128 of an
// MFMA operand (16 rows x 4 chunks, lane groups of MI355X_MICROARCH 'LDS') bank-conflict free
template <int KC> __device__ __forceinline__ int swz(int row, int c) {
  return KC == 8 ? (c ^ (row & 7)) : (c ^ ((-(row >> 2)) & 3));
}

// MODE 0: generic tap walk; 1: FAST (whole K tiles inside one tap); 2: FAST + input transform: the A operand is
// relu(batchnorm(x)) of the producer's raw output, applied between the global load and the LDS write with
// coefficients derived in the prologue from the producer's statistics -- the consumer conv absorbs the producer's
// normalise pass (one launch and one read+write of the tensor less per fused pair).
template <typename T, int BM, int BN, int WM, int WN, int KC, int MODE>
__global__ __launch_bounds__(64 * WM * WN) void igemm_kernel(IgemmGroup grp) {
  const IgemmArgs& a = grp.g[blockIdx.y];
  if ((int)blockIdx.x >= a.nbm * a.nbn) return;
  constexpr bool FAST = MODE >= 1, XF = MODE == 2;
  constexpr int NT = 64 * WM * WN;
  constexpr int EPC = 16 / (int)sizeof(T);   // elements per 16-byte chunk
  constexpr int BK = KC * EPC;               // K elements per tile row (KC chunks = 128 or 64 bytes)
  constexpr int ROWB = KC * 16;
  constexpr int RPP = NT / KC;               // rows covered per loader pass
  constexpr int PA = BM / RPP, PB = BN / RPP;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile/loader mismatch");
  constexpr int TILE_BYTES = (BM + BN) * ROWB;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  // ---- XCD-aware tile assignment (bijective for any grid size) -------------------
  const int nblk = a.nbm * a.nbn;
  int lid;
  {
    const int id = blockIdx.x, xcd = id & 7, q = nblk >> 3, r = nblk & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int bm = lid / a.nbn, bn = lid - bm * a.nbn;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid - wm * WN;
#define ST_STAMP(i) do { if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
  ST_STAMP(0);
  if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime();   // chip-wide 100 MHz clock at block entry

  // ---- loader state ----------------------------------------------------------------
  // FAST (Cin % BK == 0): every chunk of a K tile lies in ONE filter tap, so the tap walk is block-uniform
  // (scalar registers); a row keeps a base pointer and a bit mask of its in-bounds taps, and a chunk costs
  // one bit test + one pointer add.  Otherwise (the 7x7 stem, Cin = 8) each thread walks (kh, kw, c) itself.
  const int ccol = tid % KC, lrow = tid / KC;
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ W = reinterpret_cast<const T*>(a.w);
  int pixbase[PA], hw0[PA];
  const T* arow[PA];
  unsigned long long amask[PA];
  const int HoWo = a.Ho * a.Wo;
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int m = bm * BM + lrow + i * RPP;
    pixbase[i] = 0; hw0[i] = (int)0x80008000; arow[i] = X; amask[i] = 0ull;   // hi0 = wi0 = -32768: never in range
    if (m < a.M) {
      const int b = m / HoWo, rem = m - b * HoWo, ho = rem / a.Wo, wo = rem - ho * a.Wo;
      const int hi0 = ho * a.stride - a.pad, wi0 = wo * a.stride - a.pad;
      pixbase[i] = b * a.Hin * a.Win;
      hw0[i] = (hi0 << 16) | (wi0 & 0xffff);
      if (FAST) {
        arow[i] = X + ((long)(pixbase[i] + hi0 * a.Win + wi0) * a.ldx + ccol * EPC);
        // in-bounds taps form a rectangle [h0,h1) x [w0,w1): one row mask, shifted per filter row
        const int h0 = hi0 < 0 ? -hi0 : 0, h1 = a.Hin - hi0 < a.KH ? a.Hin - hi0 : a.KH;
        const int w0 = wi0 < 0 ? -wi0 : 0, w1 = a.Win - wi0 < a.KW ? a.Win - wi0 : a.KW;
        if (h1 > h0 && w1 > w0) {
          const unsigned long long wm = ((1ull << w1) - 1ull) & ~((1ull << w0) - 1ull);
          for (int fh = h0; fh < h1; ++fh) amask[i] |= wm << (fh * a.KW);
        }
      }
    }
  }
  const T* wptr[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int n = bn * BN + lrow + i * RPP;
    wptr[i] = n < a.N ? W + ((long)n * a.ldw + ccol * EPC) : nullptr;
  }
  // walk state: FAST -> block-uniform (tap, c0, tapoff); generic -> this thread's (kh, kw, kc)
  int kc = FAST ? 0 : ccol * EPC, kh = 0, kw = 0, tap = 0;
  long tapoff = 0;
  if (!FAST) while (kc >= a.Cin) { kc -= a.Cin; if (++kw == a.KW) { kw = 0; ++kh; } }
  int klin = 0;
  const int ntap = a.KH * a.KW;

  u32x4 ra[PA], rb[PB];
  bool rok[PA];          // XF: row i of the tile in registers was really loaded (padding / ragged rows stay zero)
  int xf_kc = 0;         // XF: channel offset of the tile in registers
  auto gload = [&]() {
    const bool kok = FAST ? (a.korder ? kc < a.Cin : tap < ntap) : kh < a.KH;
    if (XF) xf_kc = kc;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (FAST) {
        const bool ok = kok && ((amask[i] >> tap) & 1ull);
        if (ok) v = *reinterpret_cast<const u32x4*>(arow[i] + (tapoff + kc));
        if (XF) rok[i] = ok;
      } else {
        const int hi = (hw0[i] >> 16) + kh, wi = (int)(short)(hw0[i] & 0xffff) + kw;
        if (kok && (unsigned)hi < (unsigned)a.Hin && (unsigned)wi < (unsigned)a.Win)
          v = *reinterpret_cast<const u32x4*>(X + ((long)(pixbase[i] + hi * a.Win + wi) * a.ldx + kc));
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (kok && wptr[i]) v = *reinterpret_cast<const u32x4*>(wptr[i] + klin);
      rb[i] = v;
    }
    // advance to the next K tile
    klin += BK;
    kc += BK;
    if (FAST) {
      if (a.korder) {   // channel chunk outer, filter tap inner: the taps of one chunk re-touch the same lines (L1)
        kc -= BK;
        ++tap; if (++kw == a.KW) { kw = 0; ++kh; }
        if (tap == ntap) { tap = 0; kh = 0; kw = 0; kc += BK; }
        tapoff = (long)(kh * a.Win + kw) * a.ldx;
      } else if (kc >= a.Cin) { kc = 0; ++tap; if (++kw == a.KW) { kw = 0; ++kh; } tapoff = (long)(kh * a.Win + kw) * a.ldx; }
    } else {
      while (kc >= a.Cin) { kc -= a.Cin; if (++kw == a.KW) { kw = 0; ++kh; } }
    }
  };
  constexpr int KLOOP_BYTES = 2 * TILE_BYTES, STAGE_BYTES = 64 * (BN + 4) * 4 + 2 * BN * WM * 4;
  float* coef = reinterpret_cast<float*>(smem + (KLOOP_BYTES > STAGE_BYTES ? KLOOP_BYTES : STAGE_BYTES));   // XF: [scale(Cin) | shift(Cin)]
  auto lstore = [&](int buf) {
    char* base = smem + buf * TILE_BYTES;
    if (XF) {
      float sc[EPC], sh[EPC];
      const float* cs = coef + xf_kc + ccol * EPC;
#pragma unroll
      for (int e = 0; e < EPC; e += 4) {
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(cs + e), h4 = *reinterpret_cast<const f32x4*>(cs + a.Cin + e);
#pragma unroll
        for (int q = 0; q < 4; ++q) { sc[e + q] = s4[q]; sh[e + q] = h4[q]; }
      }
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        if (!rok[i]) continue;
        if (sizeof(T) == 2) {
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const float lo = fmaxf(__uint_as_float(ra[i][d] << 16) * sc[2 * d] + sh[2 * d], 0.f);
            const float hi = fmaxf(__uint_as_float(ra[i][d] & 0xffff0000u) * sc[2 * d + 1] + sh[2 * d + 1], 0.f);
            ra[i][d] = pack_bf16x2(lo, hi);
          }
        } else {
#pragma unroll
          for (int d = 0; d < 4; ++d) ra[i][d] = __float_as_uint(fmaxf(__uint_as_float(ra[i][d]) * sc[d] + sh[d], 0.f));
        }
      }
    }
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int row = lrow + i * RPP;
      *reinterpret_cast<u32x4*>(base + row * ROWB + (swz<KC>(row, ccol) << 4)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int row = lrow + i * RPP;
      *reinterpret_cast<u32x4*>(base + (BM + row) * ROWB + (swz<KC>(row, ccol) << 4)) = rb[i];
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int r16 = lane & 15, q4 = lane >> 4;
  const int nk = (a.K + BK - 1) / BK;

  ST_STAMP(1);
  gload();
  if (XF) {   // coefficients of the producer's BatchNorm, while the first tile is in flight
    const float inv = 1.0f / a.in_inv_count;   // field carries the row count
    for (int c = tid; c < a.Cin; c += NT) {
      bn_scale_shift(a.in_stats[c], a.in_stats[a.Cin + c], inv, a.in_gamma[c], a.in_beta[c], a.in_eps, coef[c], coef[a.Cin + c]);
    }
    __syncthreads();
  }
  lstore(0);
  __syncthreads();
  ST_STAMP(2);

  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) gload();
    const char* base = smem + (kt & 1) * TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < KC / 4; ++ks) {
      const int so = swz<KC>(r16, ks * 4 + q4) << 4;
      u32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[i] = *reinterpret_cast<const u32x4*>(base + (wm * (BM / WM) + i * 16 + r16) * ROWB + so);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        fb[j] = *reinterpret_cast<const u32x4*>(base + (BM + wn * (BN / WN) + j * 16 + r16) * ROWB + so);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) Mfma<T>::run(fb[j], fa[i], acc[i][j]);
    }
    if (more) lstore((kt + 1) & 1);
    __syncthreads();
  }

  ST_STAMP(3);
  igemm_epilogue<T, BM, BN, WM, WN>(a, acc, smem, bm, bn, tid);
  ST_STAMP(4);
  if (a.stamps && tid == 0) {
    unsigned hw;   // HW_ID: which XCC / SE / CU / SIMD this wave ran on
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    a.stamps[(size_t)blockIdx.x * 8 + 5] = ((unsigned long long)xcc << 32) | hw;
    a.stamps[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memrealtime();   // chip-wide 100 MHz clock at block exit
  }
#undef ST_STAMP
}

// =======================================================================================
// Main loop 2 (bf16, Cin % 64 == 0, no input transform): the "many small blocks" form.  128x128 tile, 4 waves (64x64 per
// wave), ONE 32 KiB K-tile buffer filled by LDS-DMA (global_load_lds, 16 B per lane; lane-linear image, bank swizzle on the
// per-lane source chunk and on the reads; padding taps / ragged rows read a zero word), two barriers per K tile:
//   DMA tile k -> vmcnt(0) -> barrier -> 16 ds_read_b128 + 32 MFMA per wave -> barrier.
// Nothing overlaps inside a block; 3-4 blocks per CU (36 KiB of LDS, no staging registers) overlap each other instead.
// =======================================================================================
__device__ __attribute__((aligned(16))) uint32_t g_zero16[4] = {0u, 0u, 0u, 0u};
typedef const void __attribute__((address_space(1))) * gptr_t;
typedef void __attribute__((address_space(3))) * lptr_t;

#ifdef ST_EXPERIMENTAL   // single-buffer form: measured slower inside the network, `make EXPERIMENTAL=1` only
__global__ __launch_bounds__(256) void igemm_s3_kernel(IgemmGroup grp) {
  typedef bf16_t T;
  const IgemmArgs& a = grp.g[blockIdx.y];
  if ((int)blockIdx.x >= a.nbm * a.nbn) return;
  constexpr int BM = 128, BN = 128, WM = 2, WN = 2, NW = 4;
  constexpr int PA = BM / (8 * NW), PB = BN / (8 * NW);     // 1-KiB pieces (8 rows) per wave per tile: 4 + 4
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;       // 4 x 4 MFMA tiles per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int nblk = a.nbm * a.nbn;
  int lid;
  {
    const int id = blockIdx.x, xcd = id & 7, q = nblk >> 3, r = nblk & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int bm = lid / a.nbn, bn = lid - bm * a.nbn;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid - wm * WN;

  const int prow = lane >> 3, ccol = (lane & 7) ^ prow;     // lane -> (row of an 8-row piece, source chunk)
  const char* Zp = reinterpret_cast<const char*>(g_zero16);
  const char* arow[PA];
  unsigned long long amask[PA];
  const char* pb[PB]; int sb[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int n = bn * BN + (wid + NW * i) * 8 + prow;
    const bool ok = n < a.N;
    pb[i] = ok ? reinterpret_cast<const char*>(a.w) + ((long)n * a.ldw + ccol * 8) * 2 : Zp;
    sb[i] = ok ? 128 : 0;
  }
  const int HoWo = a.Ho * a.Wo;
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int m = bm * BM + (wid + NW * i) * 8 + prow;
    arow[i] = Zp; amask[i] = 0ull;
    if (m < a.M) {
      const int b = m / HoWo, rem = m - b * HoWo, ho = rem / a.Wo, wo = rem - ho * a.Wo;
      const int hi0 = ho * a.stride - a.pad, wi0 = wo * a.stride - a.pad;
      arow[i] = reinterpret_cast<const char*>(a.x) + ((long)(b * a.Hin * a.Win + hi0 * a.Win + wi0) * a.ldx + ccol * 8) * 2;
      const int h0 = hi0 < 0 ? -hi0 : 0, h1 = a.Hin - hi0 < a.KH ? a.Hin - hi0 : a.KH;
      const int w0 = wi0 < 0 ? -wi0 : 0, w1 = a.Win - wi0 < a.KW ? a.Win - wi0 : a.KW;
      if (h1 > h0 && w1 > w0) {
        const unsigned long long wmk = ((1ull << w1) - 1ull) & ~((1ull << w0) - 1ull);
        for (int fh = h0; fh < h1; ++fh) amask[i] |= wmk << (fh * a.KW);
      }
    }
  }
  const int ntap = a.KH * a.KW, nchunk = a.Cin >> 6;
  int tap = 0, kh = 0, kw = 0, chunk = 0;
  const char* pa[PA]; int sa[PA];
  auto retap = [&]() {
    const long off = ((long)(kh * a.Win + kw) * a.ldx + chunk * 64) * 2;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const bool ok = (amask[i] >> tap) & 1ull;
      pa[i] = ok ? arow[i] + off : Zp;
      sa[i] = ok ? 128 : 0;
    }
  };
  retap();
  auto issue = [&]() {
#pragma unroll
    for (int i = 0; i < PA; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)pa[i], (lptr_t)(smem + (wid + NW * i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < PB; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)pb[i], (lptr_t)(smem + BM * 128 + (wid + NW * i) * 1024), 16, 0, 0);
    if (a.korder) {
      ++tap; if (++kw == a.KW) { kw = 0; ++kh; }
      if (tap == ntap) { tap = 0; kh = 0; kw = 0; ++chunk; }
      retap();
    } else if (++chunk == nchunk) {
      chunk = 0; ++tap; if (++kw == a.KW) { kw = 0; ++kh; }
      retap();
    } else {
#pragma unroll
      for (int i = 0; i < PA; ++i) pa[i] += sa[i];
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) pb[i] += sb[i];
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int r16 = lane & 15, q4 = lane >> 4;
  const int nk = a.K >> 6;
  const int arow_off = (wm * (BM / WM) + r16) * 128, brow_off = (BM + wn * (BN / WN) + r16) * 128;

  for (int kt = 0; kt < nk; ++kt) {
    issue();
    __syncthreads();                                   // emits vmcnt(0): the tile has landed, for every wave
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int so = ((ks * 4 + q4) ^ (r16 & 7)) << 4;
      u32x4 fa[TM], fb[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const u32x4*>(smem + brow_off + j * 2048 + so);
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const u32x4*>(smem + arow_off + i * 2048 + so);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) Mfma<T>::run(fb[j], fa[i], acc[i][j]);
    }
    __syncthreads();                                   // everybody is done reading before the next DMA overwrites
  }
  igemm_epilogue<T, BM, BN, WM, WN>(a, acc, smem, bm, bn, tid);
}

#endif  // ST_EXPERIMENTAL

// Same block shape with DMA tiles in flight ACROSS the barrier: three 16 KiB buffers of 32-deep K tiles (A 128x64 B + B 128x64 B),
// tile kt+2 is issued right after the barrier that publishes tile kt, one raw barrier per K tile, counted vmcnt.
// 48 KiB of LDS -> still 3 blocks per CU.  Lane-linear image of 64-byte rows: one DMA instruction = 16 rows x 4 chunks,
// LDS slot s of row r holds source chunk s ^ ((r >> 1) & 3) (conflict-free ds_read_b128 over 16 rows x 1 chunk... see read side).
template <int N> __device__ __forceinline__ void s3_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(256) void igemm_s3b_kernel(IgemmGroup grp) {
  typedef bf16_t T;
  const IgemmArgs& a = grp.g[blockIdx.y];
  if ((int)blockIdx.x >= a.nbm * a.nbn) return;
  constexpr int BM = 128, BN = 128, WM = 2, WN = 2, NW = 4, NB = 3;
  constexpr int PA = BM / (16 * NW), PB = BN / (16 * NW);   // 1-KiB pieces (16 rows x 64 B) per wave per tile: 2 + 2
  constexpr int G = PA + PB;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int TILE_BYTES = (BM + BN) * 64;                // 16 KiB
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int nblk = a.nbm * a.nbn;
  int lid;
  {
    const int id = blockIdx.x, xcd = id & 7, q = nblk >> 3, r = nblk & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int bm = lid / a.nbn, bn = lid - bm * a.nbn;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid - wm * WN;

  // lane -> (row of a 16-row piece, LDS slot); the slot receives source chunk slot ^ swz(row)
  const int prow = lane >> 2, slot = lane & 3;
  const int ccol = slot ^ ((prow >> 1) & 3);
  const char* Zp = reinterpret_cast<const char*>(g_zero16);
  const char* arow[PA];
  unsigned long long amask[PA];
  const char* pb[PB]; int sb[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int n = bn * BN + (wid + NW * i) * 16 + prow;
    const bool ok = n < a.N;
    pb[i] = ok ? reinterpret_cast<const char*>(a.w) + ((long)n * a.ldw + ccol * 8) * 2 : Zp;
    sb[i] = ok ? 64 : 0;
  }
  const int HoWo = a.Ho * a.Wo;
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int m = bm * BM + (wid + NW * i) * 16 + prow;
    arow[i] = Zp; amask[i] = 0ull;
    if (m < a.M) {
      const int b = m / HoWo, rem = m - b * HoWo, ho = rem / a.Wo, wo = rem - ho * a.Wo;
      const int hi0 = ho * a.stride - a.pad, wi0 = wo * a.stride - a.pad;
      arow[i] = reinterpret_cast<const char*>(a.x) + ((long)(b * a.Hin * a.Win + hi0 * a.Win + wi0) * a.ldx + ccol * 8) * 2;
      const int h0 = hi0 < 0 ? -hi0 : 0, h1 = a.Hin - hi0 < a.KH ? a.Hin - hi0 : a.KH;
      const int w0 = wi0 < 0 ? -wi0 : 0, w1 = a.Win - wi0 < a.KW ? a.Win - wi0 : a.KW;
      if (h1 > h0 && w1 > w0) {
        const unsigned long long wmk = ((1ull << w1) - 1ull) & ~((1ull << w0) - 1ull);
        for (int fh = h0; fh < h1; ++fh) amask[i] |= wmk << (fh * a.KW);
      }
    }
  }
  // walk in 32-channel steps; k_order 0: tap outer / channels inner, k_order 1: 64-channel chunk outer, tap inner (two steps per tap visit)
  const int ntap = a.KH * a.KW;
  int tap = 0, kh = 0, kw = 0, c32 = 0;                      // c32: 32-channel step index inside the current (tap | chunk) run
  const int run = a.korder ? 2 : (a.Cin >> 5);               // steps before the tap changes
  int chunk = 0;                                             // k_order 1: current 64-channel chunk
  const char* pa[PA]; int sa[PA];
  auto retap = [&]() {
    const long off = ((long)(kh * a.Win + kw) * a.ldx + (a.korder ? chunk * 64 : 0)) * 2;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const bool ok = (amask[i] >> tap) & 1ull;
      pa[i] = ok ? arow[i] + off : Zp;
      sa[i] = ok ? 64 : 0;
    }
  };
  retap();
  auto issue = [&](char* base) {
#pragma unroll
    for (int i = 0; i < PA; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)pa[i], (lptr_t)(base + (wid + NW * i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < PB; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)pb[i], (lptr_t)(base + BM * 64 + (wid + NW * i) * 1024), 16, 0, 0);
    if (++c32 == run) {
      c32 = 0;
      ++tap; if (++kw == a.KW) { kw = 0; ++kh; }
      if (a.korder && tap == ntap) { tap = 0; kh = 0; kw = 0; ++chunk; }
      retap();
    } else {
#pragma unroll
      for (int i = 0; i < PA; ++i) pa[i] += sa[i];
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) pb[i] += sb[i];
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int r16 = lane & 15, q4 = lane >> 4;
  const int nk = a.K >> 5;                                  // 32-deep K tiles (K % 64 == 0)
  // fragment read: row = base + r16, chunk q4 -> slot q4 ^ ((row >> 1) & 3)   (row bases are multiples of 16)
  const int so = (q4 ^ ((r16 >> 1) & 3)) << 4;
  const int arow_off = (wm * (BM / WM) + r16) * 64 + so, brow_off = (BM + wn * (BN / WN) + r16) * 64 + so;

  issue(smem);
  issue(smem + TILE_BYTES);                                  // nk >= 2
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) s3_wait_vmcnt<G>(); else s3_wait_vmcnt<0>();   // tile kt has landed (this wave's pieces)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave is done reading tile kt-1
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + 2 < nk) issue(smem + ((kt + 2) % NB) * TILE_BYTES);   // refill the buffer of tile kt-1
    const char* cur = smem + (kt % NB) * TILE_BYTES;
    u32x4 fa[TM], fb[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const u32x4*>(cur + brow_off + j * 1024);
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const u32x4*>(cur + arow_off + i * 1024);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) Mfma<T>::run(fb[j], fa[i], acc[i][j]);
  }
  __syncthreads();
  igemm_epilogue<T, BM, BN, WM, WN>(a, acc, smem, bm, bn, tid);
}

int launch_s3(IgemmArgs* arr, int n, hipStream_t st, bool pipelined = false) {
  IgemmArgs& a = arr[0];
  IgemmGroup grp;
  memset(&grp, 0, sizeof(grp));
  for (int i = 0; i < n; ++i) {
    arr[i].nbm = (arr[i].M + 127) / 128;
    arr[i].nbn = (arr[i].N + 127) / 128;
    grp.g[i] = arr[i];
  }
  const int lds = pipelined ? 3 * 256 * 64 : 64 * (128 + 4) * 4 + 2 * 128 * 2 * 4;   // 3 K tiles | the epilogue's staging rows + statistics partials
  ProfRec rec; bool prof = false;
  if (g_prof_on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof.size() < kProfMax && hipEventCreate(&rec.e0) == hipSuccess && hipEventCreate(&rec.e1) == hipSuccess) {
      rec.variant = 0; rec.flops = a.flops * n; prof = true;
      (void)hipEventRecord(rec.e0, st);
    }
  }
  if (pipelined) hipLaunchKernelGGL(igemm_s3b_kernel, dim3(a.nbm * a.nbn, n), dim3(256), lds, st, grp);
#ifdef ST_EXPERIMENTAL
  else hipLaunchKernelGGL(igemm_s3_kernel, dim3(a.nbm * a.nbn, n), dim3(256), lds, st, grp);
#else
  else { st_set_error("the single-buffer small-block kernel (ST_IGEMM_S3=1|3) needs a `make EXPERIMENTAL=1` build"); return 1; }
#endif
  if (prof) {
    (void)hipEventRecord(rec.e1, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(rec);
  }
  ST_LAUNCH_CHECK();
  return 0;
}

template <typename T, int BM, int BN, int WM, int WN, int KC, int MODE>
int launch_(IgemmArgs* arr, int n, hipStream_t st) {
  IgemmArgs& a = arr[0];
  IgemmGroup grp;
  memset(&grp, 0, sizeof(grp));
  for (int i = 0; i < n; ++i) {
    arr[i].nbm = (arr[i].M + BM - 1) / BM;
    arr[i].nbn = (arr[i].N + BN - 1) / BN;
    grp.g[i] = arr[i];
  }
  constexpr int kloop = 2 * (BM + BN) * KC * 16, stage = 64 * (BN + 4) * 4 + 2 * BN * WM * 4;   // staging rows + statistics partials
  const int lds = (kloop > stage ? kloop : stage) + (MODE == 2 ? 2 * a.Cin * (int)sizeof(float) : 0);
  constexpr int variant = (sizeof(T) == 2 ? 0 : 4) + (BN == 64 ? 1 : (BM == 64 ? 2 : (BM == 256 ? 3 : 0)));
  ProfRec rec; bool prof = false;
  if (g_prof_on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof.size() < kProfMax && hipEventCreate(&rec.e0) == hipSuccess && hipEventCreate(&rec.e1) == hipSuccess) {
      rec.variant = variant; rec.flops = a.flops * n; prof = true;
      (void)hipEventRecord(rec.e0, st);
    }
  }
  static int attr_set[64] = {};                     // per (instantiation, device): the attribute is a per-device property
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  if (dev_ >= 0 && dev_ < 64 && attr_set[dev_] < lds && lds > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<T, BM, BN, WM, WN, KC, MODE>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set[dev_] = lds;
  }
  hipLaunchKernelGGL((igemm_kernel<T, BM, BN, WM, WN, KC, MODE>), dim3(a.nbm * a.nbn, n), dim3(64 * WM * WN), lds, st, grp);
  if (prof) {
    (void)hipEventRecord(rec.e1, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(rec);
  }
  ST_LAUNCH_CHECK();
  return 0;
}

template <typename T, int BM, int BN, int WM, int WN, int KC>
int launch(IgemmArgs* arr, int n, hipStream_t st) {
  IgemmArgs& a = arr[0];
  constexpr int BK = KC * (16 / (int)sizeof(T));
  const bool fast = (a.Cin % BK == 0) && a.KH * a.KW <= 64;
  if (a.in_stats) {
    ST_CHECK(fast, "st_conv: the input transform needs Cin to be a multiple of %d", BK);
    return launch_<T, BM, BN, WM, WN, KC, 2>(arr, n, st);
  }
  return fast ? launch_<T, BM, BN, WM, WN, KC, 1>(arr, n, st) : launch_<T, BM, BN, WM, WN, KC, 0>(arr, n, st);
}

int g_tune[3] = {-1, -1, -1};   // (unused), kc, w8 (-1: take the environment default)
int tuning_get(int i, const char* env) {
  if (g_tune[i] < 0) { const char* e = getenv(env); g_tune[i] = e ? atoi(e) : 0; }
  return g_tune[i];
}

int tuning_w8() { if (g_tune[2] < 0) { const char* e = getenv("ST_IGEMM_W8"); g_tune[2] = e ? atoi(e) : 1; } return g_tune[2]; }
int tuning_kc() { return tuning_get(1, "ST_IGEMM_KC"); }

template <typename T, int KC>
int dispatch_kc(IgemmArgs* arr, int n, hipStream_t st) {
  IgemmArgs& a = arr[0];
  // Tile choice: fill >= ~1.5 waves of the 256 CUs where the problem allows it.
  const long t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
  if (a.N <= 64) {
    if constexpr (KC == 8) { if (tuning_w8() && tuning_w8() != 3) return launch<T, 128, 64, 4, 2, KC>(arr, n, st); }   // 8 waves: the loader needs 64 rows per pass
    return launch<T, 128, 64, 4, 1, KC>(arr, n, st);
  }
  if (tuning_w8() == 2 && a.M >= 256 * 64) return launch<T, 256, 128, 4, 2, KC>(arr, n, st);
  if (t128 >= 384) return tuning_w8() ? launch<T, 128, 128, 2, 4, KC>(arr, n, st) : launch<T, 128, 128, 2, 2, KC>(arr, n, st);
  const long t64 = (long)((a.M + 63) / 64) * ((a.N + 127) / 128);
  if (t64 >= 256 || a.M <= 64) return launch<T, 64, 128, 1, 4, KC>(arr, n, st);
  return tuning_w8() ? launch<T, 128, 128, 2, 4, KC>(arr, n, st) : launch<T, 128, 128, 2, 2, KC>(arr, n, st);
}

template <typename T>
int dispatch(IgemmArgs* arr, int n, hipStream_t st) {
  IgemmArgs& a = arr[0];
  // EXPERIMENTAL, off by default (ST_IGEMM_S3=1 / st_tune(1,..) routes every legal problem to it, =3 only the 3x3 layers with
  // >= 256 channels and >= 256 tiles): the many-small-blocks form is 15-18 % faster than the 8-wave kernel on the 3x3 256->256
  // @14x14 layer in isolation (47 vs 55 us back to back, 50 vs 56 behind a bn_act launch) but 6 us SLOWER inside the network
  // (59 vs 53 us per launch in the rocprof trace of a forward; encoder 7.00 vs 6.79 ms): with cold per-layer weights a block
  // that overlaps nothing internally pays every K tile's miss latency in full, see DESIGN.md
  const int s3 = tuning_get(0, "ST_IGEMM_S3");
  const bool s3_legal = sizeof(T) == 2 && a.Cin % 64 == 0 && a.K % 64 == 0 && a.KH * a.KW <= 64 && !a.in_stats && a.N > 64 && a.ldx >= a.Cin;
  const long s3_tiles = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
  if (s3_legal && (s3 == 1 || (s3 == 3 && a.KH * a.KW > 1 && a.Cin >= 256 && s3_tiles >= 256)))
    return launch_s3(arr, n, st);
  // The three-buffer form is the default for long-K layers with enough tiles (K >= 1024, >= 256 input channels, >= 256 tiles:
  // the 14x14 conv1 / conv2 layers of ResNet-101): 145 -> 136 us per bottleneck block in tools/chain_bench.py.  4: everywhere
  // legal, 5: only the 3x3 layers, 2: never.
  if (s3_legal && a.K >= 128 && s3 != 2 &&
      (s3 == 4 || (s3 == 5 && a.KH * a.KW > 1 && a.Cin >= 256 && s3_tiles >= 256) || (s3 == 0 && a.K >= 1024 && a.Cin >= 256 && s3_tiles >= 256)))
    return launch_s3(arr, n, st, true);
  // 64-byte tile rows halve the LDS footprint (3-4 blocks per CU instead of 2): the short-K pointwise layers
  // (K <= 256: 4 K steps or fewer, prologue/epilogue-bound) gain 8-15 % from the extra overlap, long-K layers lose.
  const int kc = tuning_kc();
  const bool short_k = a.KH * a.KW == 1 && a.K <= 256 && a.K % 64 == 0 && a.M >= 4096;
  if ((kc == 4 || (kc == 0 && short_k)) && !a.korder) return dispatch_kc<T, 4>(arr, n, st);
  return dispatch_kc<T, 8>(arr, n, st);
}

}  // namespace

StProfScope::StProfScope(int variant, double flops, hipStream_t st) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec rec;
  if (g_prof.size() < kProfMax && hipEventCreate(&rec.e0) == hipSuccess && hipEventCreate(&rec.e1) == hipSuccess) {
    rec.variant = variant; rec.flops = flops;
    (void)hipEventRecord(rec.e0, st);
    idx = g_prof.size(); on = true;
    g_prof.push_back(rec);
  }
}
unsigned long long* st_debug_stamps_ptr() { return g_stamps; }
void StProfScope::end(hipStream_t st) {
  if (!on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (idx < g_prof.size()) (void)hipEventRecord(g_prof[idx].e1, st);
}

static int fill_args(const st_conv_desc* d, IgemmArgs& a) {
  ST_CHECK(d && d->x && d->w && d->y, "st_conv: null pointer");
  ST_CHECK(d->dtype == ST_F32 || d->dtype == ST_BF16, "st_conv: bad dtype %d", d->dtype);
  const int epc = d->dtype == ST_BF16 ? 8 : 4;
  ST_CHECK(d->Cin > 0 && d->Cin % epc == 0, "st_conv: Cin=%d must be a multiple of %d", d->Cin, epc);
  ST_CHECK(d->ldx % epc == 0 && d->ldw % epc == 0, "st_conv: ldx=%d/ldw=%d must be multiples of %d", d->ldx, d->ldw, epc);
  const bool sliding = d->ldx < d->Cin;   // Cin/ldx neighbouring pixels of a row form one tap (space-to-depth stem)
  ST_CHECK(!sliding || (d->KW == 1 && d->pad == 0 && d->Cin % d->ldx == 0 && !d->k_order &&
                        (d->Wo - 1) * d->stride + d->Cin / d->ldx <= d->Win && (d->Ho - 1) * d->stride + d->KH <= d->Hin),
           "st_conv: sliding-window input needs KW=1, pad=0, Cin %% ldx == 0 and rows padded by the caller");
  ST_CHECK(d->ldw >= d->KH * d->KW * d->Cin, "st_conv: leading dimensions too small");
  ST_CHECK(d->stats_replicas >= 0 && d->stats_replicas <= 1024, "st_conv: bad stats_replicas");
  ST_CHECK(d->ldy % 4 == 0 && d->ldy >= d->N, "st_conv: ldy=%d must be a multiple of 4 and >= N=%d", d->ldy, d->N);
  ST_CHECK(d->B > 0 && d->N > 0 && d->Ho > 0 && d->Wo > 0 && d->KH > 0 && d->KW > 0 && d->stride > 0, "st_conv: bad geometry");
  ST_CHECK(d->Hin < 32768 && d->Win < 32768, "st_conv: spatial size too large");
  ST_CHECK((long)d->B * d->Hin * d->Win * d->ldx < (1L << 40), "st_conv: input too large");
  ST_CHECK((d->scale == nullptr) == (d->shift == nullptr), "st_conv: scale and shift must be given together");
  ST_CHECK(!d->k_order || (d->Cin % (8 * epc) == 0 && d->KH * d->KW <= 64), "st_conv: k_order=1 needs Cin to be a multiple of %d", 8 * epc);
  a.x = d->x; a.w = d->w; a.y = d->y; a.bias = d->bias; a.scale = d->scale; a.shift = d->shift;
  a.residual = d->residual; a.stats = d->stats;
  a.M = d->B * d->Ho * d->Wo; a.N = d->N; a.K = d->KH * d->KW * d->Cin;
  a.Hin = d->Hin; a.Win = d->Win; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
  a.ldx = d->ldx; a.ldw = d->ldw; a.ldy = d->ldy;
  a.relu = d->relu; a.accumulate = d->accumulate; a.out_f32 = d->out_dtype == ST_F32;
  a.korder = d->k_order; a.srep = d->stats_replicas; a.stamps = g_stamps; a.atomic = 0;
  a.in_stats = d->in_stats; a.in_gamma = d->in_gamma; a.in_beta = d->in_beta; a.in_inv_count = d->in_count;   /* the kernel forms 1/count itself, as bn_act does */ a.in_eps = d->in_eps;
  ST_CHECK(!d->in_stats || (d->in_gamma && d->in_beta && d->in_count > 0.f && d->ldx >= d->Cin && d->Cin <= 8192), "st_conv: input transform needs gamma, beta, count");
  a.flops = 2.0 * a.M * a.N * d->KH * d->KW * (d->Cin_logical > 0 ? d->Cin_logical : d->Cin);
  return 0;
}

extern "C" int st_conv(const st_conv_desc* d, void* stream) {
  IgemmArgs a;
  if (fill_args(d, a)) return 1;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->split_k > 1) {
    // Split-K for plain GEMMs whose M x N gives too few tiles (dy = dlogits W_lin: 60 tiles, K = 10240): S members of one
    // grouped launch take K/S each and add their fp32 tiles with atomics; bias goes with slice 0.
    const int S = d->split_k, bk = d->dtype == ST_BF16 ? 64 : 32;
    ST_CHECK(S <= kGroup && d->KH == 1 && d->KW == 1 && d->Hin == 1 && d->Win == 1 && d->out_dtype == ST_F32 && !d->relu && !d->scale &&
             !d->residual && !d->stats && !d->in_stats && d->Cin % (S * bk) == 0,
             "st_conv: split_k=%d needs a plain fp32-output GEMM with K a multiple of %d", S, S * bk);
    if (!d->accumulate && hipMemsetAsync(d->y, 0, (size_t)a.M * d->ldy * sizeof(float), st) != hipSuccess) { st_set_error("memset failed"); return 1; }
    IgemmArgs arr[kGroup];
    const int ks = d->Cin / S;
    const size_t es = d->dtype == ST_BF16 ? 2 : 4;
    for (int i = 0; i < S; ++i) {
      arr[i] = a;
      arr[i].x = reinterpret_cast<const char*>(a.x) + (size_t)i * ks * es;
      arr[i].w = reinterpret_cast<const char*>(a.w) + (size_t)i * ks * es;
      arr[i].K = ks; arr[i].Cin = ks; arr[i].atomic = 1; arr[i].accumulate = 0;
      if (i) arr[i].bias = nullptr;
      arr[i].flops = a.flops / S;
    }
    return d->dtype == ST_BF16 ? dispatch<bf16_t>(arr, S, st) : dispatch<float>(arr, S, st);
  }
  return d->dtype == ST_BF16 ? dispatch<bf16_t>(&a, 1, st) : dispatch<float>(&a, 1, st);
}

// n (<= 12) independent problems of IDENTICAL shape, dtype and options in one launch (pointers differ)
extern "C" int st_conv_batch(const st_conv_desc* d, int n, void* stream) {
  ST_CHECK(d && n >= 1 && n <= kGroup, "st_conv_batch: 1..%d problems per launch", kGroup);
  IgemmArgs arr[kGroup];
  for (int i = 0; i < n; ++i) {
    if (fill_args(d + i, arr[i])) return 1;
    const st_conv_desc &p = d[0], &q = d[i];
    ST_CHECK(p.dtype == q.dtype && p.out_dtype == q.out_dtype && p.B == q.B && p.Hin == q.Hin && p.Win == q.Win && p.Cin == q.Cin &&
             p.Ho == q.Ho && p.Wo == q.Wo && p.N == q.N && p.KH == q.KH && p.KW == q.KW && p.stride == q.stride && p.pad == q.pad &&
             p.ldx == q.ldx && p.ldw == q.ldw && p.ldy == q.ldy && p.k_order == q.k_order && (p.in_stats == nullptr) == (q.in_stats == nullptr),
             "st_conv_batch: problem %d differs in shape or options from problem 0", i);
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  return d->dtype == ST_BF16 ? dispatch<bf16_t>(arr, n, st) : dispatch<float>(arr, n, st);
}



// Benchmarking knob (tools/bench_conv.py): main-loop variant selection; -1 keeps the current value.
extern "C" int st_tune(int ring, int kc, int w8) {
  if (ring >= 0) g_tune[0] = ring;
  if (kc >= 0) g_tune[1] = kc;
  if (w8 >= 0) g_tune[2] = w8;
  return 0;
}

// Debug aid (tools/conv_stamps.py): when set, block b of every st_conv launch writes s_memtime at kernel entry, after the
// address set-up, after the first tile is in LDS, after the K loop and after the epilogue to buf[8*b .. 8*b+4], and its
// (XCC_ID << 32 | HW_ID) to buf[8*b+5].  NULL (the default) switches it off.
extern "C" int st_debug_stamps(unsigned long long* buf) { g_stamps = buf; return 0; }

// ---- profiler control (used by bench.py only) -----------------------------------------------
extern "C" int st_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  g_prof.clear();
  g_prof_on = on != 0;
  return 0;
}

// Sums per kernel variant (0: bf16 128x128, 1: bf16 128x64, 2: bf16 64x128, 4..6: the f32 forms).
// The caller must have synchronised the stream(s).  Arrays must hold 32 entries (prof.h lists the variants).
extern "C" int st_prof_collect(double* ms, double* flops, long* launches) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int v = 0; v < kVariants; ++v) { ms[v] = 0; flops[v] = 0; launches[v] = 0; }
  for (auto& r : g_prof) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) { st_set_error("st_prof_collect: events not complete"); return 1; }
    ms[r.variant] += t; flops[r.variant] += r.flops; launches[r.variant] += 1;
  }
  return 0;
}
