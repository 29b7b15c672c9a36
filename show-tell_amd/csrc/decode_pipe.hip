// Greedy decoding (reference rnn.py:37-58: 25 x [5-layer GRU step -> vocabulary projection -> arg-max -> embedding]) as ONE persistent
// kernel laid out as a LAYER-PER-XCD PIPELINE (gfx950 / MI355X, bf16).
//
// Why.  A token step is a chain of L + 1 dependent stages (L GRU layers, the vocabulary arg-max); as a launch chain every stage pays a
// kernel boundary (43.9 us per step at B = 128, 0.078 of the HBM roofline), and a persistent kernel whose stages synchronise ALL 256
// workgroups pays the same in cross-XCD hand-offs (tools/persistent_probe.hip: 45.3 us).  What does work (tools/xcd_pipeline_probe.hip:
// 18.8 - 19.5 us per step) is to give every stage its own XCD and to cut the batch into independent CHAINS of 32 sequences:
//   * XCD l < L holds GRU (or LSTM: 4 gates, cell state in the workgroup's LDS) layer l: 32 workgroups x 16 hidden units; a workgroup's 48 gate rows of W_ih and W_hh (96 KB) stay in the
//     registers of its four waves (K split four ways, exactly the K partition of rnn_gemm_kernel) for the whole decode;
//   * XCDs L .. 7 hold the vocabulary projection: (8 - L) x 32 workgroups x <= 128 entries, weights in registers (full K per wave,
//     one accumulator chain per 16-entry tile: the summation order of vocab_argmax_lds_kernel);
//   * a stage visit handles one chain of one token step: wait for the producer stage's arrival counter (32 or (8 - L) x 32 arrivals
//     on ONE 128-byte line: an XCD-to-XCD edge, never the chip), read the chain's 32 x 512 activations, multiply, publish (sc1
//     write-through stores into a buffer that is fresh for every (step, layer, chain): no cache can hold a stale line, so consumers
//     need no cache maintenance), arrive.  The recurrent half W_hh h_l(t-1) is computed before the wait: off the token chain.
//   * the chains are independent (sequences are), so while chain c sits in layer 3, chain c + 1 is in layer 2, .. : all XCDs work.
// The arithmetic is the launch chain's, operation for operation (same MFMA shapes and K order, same reduction order, same gate
// function st_gru_unit, same arg-max keys): token ids are bit-identical to the split-step path of st_rnn_greedy (tests assert it).
// Every spin is bounded; on a timeout or when the dispatcher did not deal 32 workgroups to every XCD the kernel raises a flag and the
// host falls back to the launch chain.
#include "common.h"
#include "rnn_kernels.h"
#include <stdlib.h>

namespace {

constexpr int PH = 512;                    // E = H = 512
constexpr int PIXB = 2 * PH + 32;          // padded LDS row (bytes)
constexpr int CR = 32;                     // sequences per chain
constexpr int SPIN_LIMIT = 1 << 20;
constexpr int PIPE_LDS = 96 * 1024;        // > 80 KB: one workgroup per CU (the grid must be co-resident, 32 workgroups per XCD)
constexpr int MAXL = 5;

struct PipeArgs {
  const bf16_t* feat; const bf16_t* emb;
  const bf16_t* w_ih[MAXL]; const bf16_t* w_hh[MAXL]; const float* b_ih[MAXL]; const float* b_hh[MAXL];
  const bf16_t* w_lin; const float* b_lin;
  bf16_t* act;                 // [steps][L][nch * 32][512]: layer outputs, each buffer written once and read afterwards
  unsigned long long* keys;    // [steps][nch * 32]: packed (value, index) maxima, zero-initialised
  unsigned* cnt;               // [steps][L + 1][nch] x 32 dwords: arrival counters, each on its own 128-byte line, zero-initialised
  unsigned* ticket;            // [8] per-XCD tickets, [8] = error flag, zero-initialised
  long* ids_out;               // [B][steps]
  int B, steps, L, V, nch, tpw;   // tpw: 16-entry vocabulary tiles per vocabulary workgroup (<= 8)
};

__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
}

// one wave polls (relaxed agent loads: sc1, served past the L1), everybody leaves through a barrier; false = timeout / abort
__device__ __forceinline__ bool wait_count(const unsigned* c, unsigned want, unsigned* err, int wid, int lane, int* abort_flag) {
  if (wid == 0) {
    int spins = 0;
    while (true) {
      const unsigned v = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v >= want) break;
      if (++spins > SPIN_LIMIT || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        if (lane == 0) { *abort_flag = 1; __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  return *abort_flag == 0;
}

// NG = 3: GRU (gate rows r, z, n); NG = 4: LSTM (i, f, g, o; the cell state of a workgroup's own 16 units never leaves its LDS)
template <int NG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void decode_pipe_kernel(PipeArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS: [0, 32 x PIXB) activation tile | reduction buffer [4 waves][2 tiles][NG gates][64] f32x4 | wbest [4][32] u64 | flags |
  //      LSTM: cell state [8 chains][2 tiles][64 lanes] f32x4 (bf16-rounded values, as the launch chain stores c)
  f32x4* red = reinterpret_cast<f32x4*>(smem + CR * PIXB);
  unsigned long long* wbest = reinterpret_cast<unsigned long long*>(smem + CR * PIXB + 4 * 2 * 4 * 64 * 16);
  int* flags = reinterpret_cast<int*>(smem + CR * PIXB + 4 * 2 * 4 * 64 * 16 + 4 * CR * 8);
  f32x4* cstate = reinterpret_cast<f32x4*>(smem + CR * PIXB + 4 * 2 * 4 * 64 * 16 + 4 * CR * 8 + 64);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, q4 = lane >> 4;
  unsigned* err = a.ticket + 8;
  if (tid == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7;
    flags[1] = (int)(xcc * 64 + atomicAdd(&a.ticket[xcc], 1u));
    flags[0] = 0;
  }
  __syncthreads();
  int* abort_flag = flags;
  const int xcc = flags[1] >> 6, li = flags[1] & 63;
  if (li >= 32) {                                      // the dispatcher did not deal the grid evenly over the XCDs: the host falls back
    if (tid == 0) __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const int L = a.L, nch = a.nch, nvw = (8 - L) * 32;
  const size_t act_layer = (size_t)nch * CR * PH;      // elements per (step, layer)
  auto act_buf = [&](int t, int l, int c) { return a.act + ((size_t)t * L + l) * act_layer + (size_t)c * CR * PH; };
  auto cnt_of = [&](int t, int st, int c) { return a.cnt + (((size_t)t * (L + 1) + st) * nch + c) * 32; };
  // 32 rows x 512 -> LDS tile (padded rows): eight 16-byte chunks per thread, all in flight before the first LDS write
  auto stage_rows = [&](const bf16_t* const (&rowp)[8]) {
    u32x4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const u32x4*>(rowp[i] + (tid & 63) * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(smem + ((tid >> 6) + 4 * i) * PIXB + (tid & 63) * 16) = v[i];
  };

  if (xcc < L) {
    // ================================ GRU layer l = xcc: hidden units [16 li, 16 li + 16) =================================================
    const int l = xcc;
    // weights in registers: wave w holds K-steps 4w .. 4w + 3 of both halves for the 3 gates (rnn_gemm_kernel's K slice `wid`)
    u32x4 wx[NG][4], wh[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const size_t off = (size_t)(g * PH + 16 * li + r16) * PH + ((wid * 4 + kk) * 4 + q4) * 8;
        wx[g][kk] = *reinterpret_cast<const u32x4*>(a.w_ih[l] + off);
        wh[g][kk] = *reinterpret_cast<const u32x4*>(a.w_hh[l] + off);
      }
    // biases of this lane's units 16 li + 4 q4 + e (epilogue lanes: waves 0 and 1, one 16-row tile each)
    float bi_[NG][4], bh_[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) { bi_[g][e] = a.b_ih[l][g * PH + 16 * li + 4 * q4 + e]; bh_[g][e] = a.b_hh[l][g * PH + 16 * li + 4 * q4 + e]; }

    for (int t = 0; t < a.steps; ++t) {
      for (int c = 0; c < nch; ++c) {
        const int r0 = c * CR;
        float gh[NG][4], hp[4] = {0.f, 0.f, 0.f, 0.f};
        // ---- recurrent half, OFF the token chain: gh = W_hh h_l(t-1) + b_hh (t = 0: h = 0) --------------------------------------------
        if (t > 0) {
          if (!wait_count(cnt_of(t - 1, l, c), 32u, err, wid, lane, abort_flag)) return;   // (this layer's own previous step: all 32 slices)
          const bf16_t* hb = act_buf(t - 1, l, c);
          const bf16_t* rowp[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) rowp[i] = hb + (size_t)((tid >> 6) + 4 * i) * PH;
          stage_rows(rowp);
          __syncthreads();
          f32x4 acc[2][NG];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
              for (int i = 0; i < 2; ++i) {
                const u32x4 fa = *reinterpret_cast<const u32x4*>(smem + (i * 16 + r16) * PIXB + ((wid * 4 + kk) * 4 + q4) * 16);
                acc[i][g] = mfma16(wh[g][kk], fa, acc[i][g]);
              }
          if (wid < 2) {                               // previous state of this lane's units (row tile wid)
            const u32x2 p = *reinterpret_cast<const u32x2*>(smem + (wid * 16 + r16) * PIXB + (16 * li + 4 * q4) * 2);
            hp[0] = __uint_as_float(p[0] << 16); hp[1] = __uint_as_float(p[0] & 0xffff0000u);
            hp[2] = __uint_as_float(p[1] << 16); hp[3] = __uint_as_float(p[1] & 0xffff0000u);
          }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < NG; ++g) red[((wid * 2 + i) * NG + g) * 64 + lane] = acc[i][g];
          __syncthreads();
          if (wid < 2) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
              f32x4 s = red[((0 * 2 + wid) * NG + g) * 64 + lane];
#pragma unroll
              for (int w = 1; w < 4; ++w) s += red[((w * 2 + wid) * NG + g) * 64 + lane];
#pragma unroll
              for (int e = 0; e < 4; ++e) gh[g][e] = s[e] + bh_[g][e];
            }
          }
          __syncthreads();                             // the tile and the reduction buffer are free again
        } else {
#pragma unroll
          for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) gh[g][e] = 0.f + bh_[g][e];
        }
        // ---- input half, ON the token chain ------------------------------------------------------------------------------------------
        {
          const bf16_t* rowp[8];
          if (l == 0) {
            if (t == 0) {
#pragma unroll
              for (int i = 0; i < 8; ++i) { int m = r0 + (tid >> 6) + 4 * i; m = m < a.B ? m : a.B - 1; rowp[i] = a.feat + (size_t)m * PH; }
            } else {
              if (!wait_count(cnt_of(t - 1, L, c), (unsigned)nvw, err, wid, lane, abort_flag)) return;
#pragma unroll
              for (int i = 0; i < 8; ++i) {        // token of step t-1 -> embedding row (the keys were merged by memory-side atomics)
                const int row = (tid >> 6) + 4 * i;
                const unsigned long long key = __hip_atomic_load(&a.keys[(size_t)(t - 1) * nch * CR + r0 + row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int tok = (int)(0xffffffffu - (unsigned)(key & 0xffffffffull));
                if (tok < 0 || tok >= a.V) tok = 0;
                if (li == 0 && (tid & 63) == 0 && r0 + row < a.B) a.ids_out[(long)(r0 + row) * a.steps + t - 1] = tok;
                rowp[i] = a.emb + (size_t)tok * PH;
              }
            }
          } else {
            if (!wait_count(cnt_of(t, l - 1, c), 32u, err, wid, lane, abort_flag)) return;
            const bf16_t* xb = act_buf(t, l - 1, c);
#pragma unroll
            for (int i = 0; i < 8; ++i) rowp[i] = xb + (size_t)((tid >> 6) + 4 * i) * PH;
          }
          stage_rows(rowp);
        }
        __syncthreads();
        {
          f32x4 acc[2][NG];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
              for (int i = 0; i < 2; ++i) {
                const u32x4 fa = *reinterpret_cast<const u32x4*>(smem + (i * 16 + r16) * PIXB + ((wid * 4 + kk) * 4 + q4) * 16);
                acc[i][g] = mfma16(wx[g][kk], fa, acc[i][g]);
              }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < NG; ++g) red[((wid * 2 + i) * NG + g) * 64 + lane] = acc[i][g];
        }
        __syncthreads();
        if (wid < 2) {
          float xg[NG][4];
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            f32x4 s = red[((0 * 2 + wid) * NG + g) * 64 + lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) s += red[((w * 2 + wid) * NG + g) * 64 + lane];
#pragma unroll
            for (int e = 0; e < 4; ++e) xg[g][e] = NG == 3 ? s[e] + bi_[g][e] : s[e];      // LSTM: the raw sums (bias added in its own order below)
          }
          float hn[4];
          if constexpr (NG == 3) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { float r_, z_, n_; hn[e] = st_gru_unit(xg[0][e], xg[1][e], xg[2][e], gh[0][e], gh[1][e], gh[2][e], hp[e], r_, z_, n_); }
          } else {
            // rnn_gemm_kernel's LSTM epilogue, split form: pre = ((0 + gh) + input-half sums) + b_ih; c is stored (and read back) in bf16
            f32x4 cp = t > 0 ? cstate[(c * 2 + wid) * 64 + lane] : f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 cnew;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float pre[4], ig, fg, gg, og, cn;
#pragma unroll
              for (int g = 0; g < 4; ++g) pre[g] = ((0.f + gh[g][e]) + xg[g][e]) + bi_[g][e];
              hn[e] = st_lstm_unit(pre[0], pre[1], pre[2], pre[3], cp[e], ig, fg, gg, og, cn);
              cnew[e] = cn;
            }
            const u32x2 cb = u32x2{pack_bf16x2(cnew[0], cnew[1]), pack_bf16x2(cnew[2], cnew[3])};
            cstate[(c * 2 + wid) * 64 + lane] = f32x4{__uint_as_float(cb[0] << 16), __uint_as_float(cb[0] & 0xffff0000u),
                                                     __uint_as_float(cb[1] << 16), __uint_as_float(cb[1] & 0xffff0000u)};
          }
          const u32x2 o = u32x2{pack_bf16x2(hn[0], hn[1]), pack_bf16x2(hn[2], hn[3])};
          bf16_t* dst = act_buf(t, l, c) + (size_t)(wid * 16 + r16) * PH + 16 * li + 4 * q4;
          asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(dst), "v"(o) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(cnt_of(t, l, c), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // the last step's token ids (the steps before were written by layer 0 as it gathered their embeddings)
    if (l == 0 && li == 0) {
      for (int c = 0; c < nch; ++c) {
        if (!wait_count(cnt_of(a.steps - 1, L, c), (unsigned)nvw, err, wid, lane, abort_flag)) return;
        if (tid < CR && c * CR + tid < a.B) {
          const unsigned long long key = __hip_atomic_load(&a.keys[(size_t)(a.steps - 1) * nch * CR + c * CR + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          int tok = (int)(0xffffffffu - (unsigned)(key & 0xffffffffull));
          if (tok < 0 || tok >= a.V) tok = 0;
          a.ids_out[(long)(c * CR + tid) * a.steps + a.steps - 1] = tok;
        }
      }
    }
    return;
  }

  // ==================================== vocabulary stage: workgroup vi of nvw, tiles [vi tpw, vi tpw + tpw) =================================
  const int vi = (xcc - L) * 32 + li;
  const int ntile = (a.V + 15) / 16;
  // wave w owns the workgroup's tiles w and w + 4: full-K fragments in registers (16 K-steps each)
  u32x4 fw[2][16];
  float bz[2][4];
  int n0[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int jt = wid + 4 * jj, tile = vi * a.tpw + jt;
    const bool on = jt < a.tpw && tile < ntile;
    n0[jj] = on ? tile * 16 : -1;
    const int nr = tile * 16 + r16;
    const bool nok = on && nr < a.V;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      fw[jj][u] = u32x4{0u, 0u, 0u, 0u};
      if (nok) fw[jj][u] = *reinterpret_cast<const u32x4*>(a.w_lin + (size_t)nr * PH + (u * 4 + q4) * 8);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { const int n = tile * 16 + 4 * q4 + e; bz[jj][e] = (on && n < a.V) ? a.b_lin[n] : 0.f; }
  }
  for (int t = 0; t < a.steps; ++t) {
    for (int c = 0; c < nch; ++c) {
      if (!wait_count(cnt_of(t, L - 1, c), 32u, err, wid, lane, abort_flag)) return;
      {
        const bf16_t* hb = act_buf(t, L - 1, c);
        const bf16_t* rowp[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) rowp[i] = hb + (size_t)((tid >> 6) + 4 * i) * PH;
        stage_rows(rowp);
      }
      __syncthreads();
      // (value, first index) over this wave's entries, per row of the chain's two 16-row tiles
      float best[2] = {-INFINITY, -INFINITY}; int bidx[2] = {0x7fffffff, 0x7fffffff};
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        if (n0[jj] >= 0) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 16; ++u) {
              const u32x4 fa = *reinterpret_cast<const u32x4*>(smem + (i * 16 + r16) * PIXB + (u * 4 + q4) * 16);
              acc = mfma16(fw[jj][u], fa, acc);
            }
            const int n = n0[jj] + 4 * q4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (n + e < a.V) {
                const float v = acc[e] + bz[jj][e];
                if (v > best[i] || (v == best[i] && n + e < bidx[i])) { best[i] = v; bidx[i] = n + e; }
              }
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
          const float ov = __shfl_xor(best[i], o, 64); const int oi = __shfl_xor(bidx[i], o, 64);
          if (ov > best[i] || (ov == best[i] && oi < bidx[i])) { best[i] = ov; bidx[i] = oi; }
        }
        if (q4 == 0) {
          unsigned long long key = 0ull;
          if (bidx[i] != 0x7fffffff) {
            unsigned u = __float_as_uint(best[i]);
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
            key = ((unsigned long long)u << 32) | (unsigned long long)(0xffffffffu - (unsigned)bidx[i]);
          }
          wbest[wid * CR + i * 16 + r16] = key;
        }
      }
      __syncthreads();
      if (tid < CR) {
        unsigned long long k = wbest[tid];
#pragma unroll
        for (int w = 1; w < 4; ++w) { const unsigned long long o = wbest[w * CR + tid]; k = o > k ? o : k; }
        if (k) atomicMax(a.keys + (size_t)t * nch * CR + c * CR + tid, k);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(cnt_of(t, L, c), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

// bytes the pipelined decoder needs behind the launch chain's workspace (0: this configuration stays on the launch chain)
size_t rnn_greedy_pipe_bytes(const st_rnn_params* p, int B, int steps) {
  if (!p || (p->cell != ST_CELL_GRU && p->cell != ST_CELL_LSTM) || p->dtype != ST_BF16 || p->L < 1 || p->L > MAXL || p->H != PH || p->E != PH || p->in0 != PH) return 0;
  if (B < 1 || B > 256 || steps < 1 || steps > 64) return 0;
  const int nvw = (8 - p->L) * 32, ntile = (p->V + 15) / 16, tpw = (ntile + nvw - 1) / nvw;
  if (tpw > 8) return 0;
  const int nch = (B + CR - 1) / CR;
  return al256((size_t)steps * p->L * nch * CR * PH * 2) + al256((size_t)steps * nch * CR * 8) + al256((size_t)steps * (p->L + 1) * nch * 128) + 256;
}

// 0: ids_out holds the result; 1: error (st_last_error); 2: not run / gave up -- the caller runs the launch chain
int rnn_greedy_pipe(const st_rnn_params* p, const void* feat, int B, int steps, void* ws, size_t ws_bytes, long* ids_out, hipStream_t st) {
  const char* env = getenv("ST_DECODE_PIPE");                     // read per call: ST_DECODE_PIPE=0 keeps the launch chain (A/B runs, tests)
  const bool on = !env || atoi(env) != 0;
  const size_t need = rnn_greedy_pipe_bytes(p, B, steps);
  if (!on || need == 0 || ws_bytes < need) return 2;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return 2;   // the result check below synchronises
  int dev = 0, ncu = 0;
  (void)hipGetDevice(&dev);
  if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu != 256) return 2;
  PipeArgs a{};
  a.feat = reinterpret_cast<const bf16_t*>(feat); a.emb = reinterpret_cast<const bf16_t*>(p->emb);
  for (int l = 0; l < p->L; ++l) {
    a.w_ih[l] = reinterpret_cast<const bf16_t*>(p->w_ih[l]); a.w_hh[l] = reinterpret_cast<const bf16_t*>(p->w_hh[l]);
    a.b_ih[l] = p->b_ih[l]; a.b_hh[l] = p->b_hh[l];
  }
  a.w_lin = reinterpret_cast<const bf16_t*>(p->w_lin); a.b_lin = p->b_lin;
  a.B = B; a.steps = steps; a.L = p->L; a.V = p->V; a.nch = (B + CR - 1) / CR;
  const int nvw = (8 - p->L) * 32, ntile = (p->V + 15) / 16;
  a.tpw = (ntile + nvw - 1) / nvw;
  char* w = reinterpret_cast<char*>(ws);
  a.act = reinterpret_cast<bf16_t*>(w); w += al256((size_t)steps * p->L * a.nch * CR * PH * 2);
  char* zero0 = w;
  a.keys = reinterpret_cast<unsigned long long*>(w); w += al256((size_t)steps * a.nch * CR * 8);
  a.cnt = reinterpret_cast<unsigned*>(w); w += al256((size_t)steps * (p->L + 1) * a.nch * 128);
  a.ticket = reinterpret_cast<unsigned*>(w); w += 256;
  a.ids_out = ids_out;
  if (hipMemsetAsync(zero0, 0, (size_t)(w - zero0), st) != hipSuccess) { st_set_error("rnn_greedy_pipe: memset failed"); return 1; }
  static int attr_set[64] = {};
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_pipe_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, PIPE_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_pipe_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, PIPE_LDS);
    attr_set[dev] = 1;
  }
  if (p->cell == ST_CELL_GRU) hipLaunchKernelGGL(decode_pipe_kernel<3>, dim3(256), dim3(256), PIPE_LDS, st, a);
  else hipLaunchKernelGGL(decode_pipe_kernel<4>, dim3(256), dim3(256), PIPE_LDS, st, a);
  ST_LAUNCH_CHECK();
  unsigned flag = 0;
  if (hipMemcpyAsync(&flag, a.ticket + 8, sizeof(flag), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
    st_set_error("rnn_greedy_pipe: result check failed");
    return 1;
  }
  return flag == 0 ? 0 : 2;     // 1: a bounded wait ran out (the grid was not co-resident), 2: uneven XCD placement
}
