// Greedy decode as a LAYER-PER-XCD PIPELINE: what would a persistent decoder cost per token step on MI355X if no hand-off ever had to
// cross the whole chip?  (the all-to-all structure of tools/persistent_probe.hip measured 45.3 us per step, as much
// as the launch chain.)  Like that probe this is not the decoder: it runs the decoder's DEPENDENCE STRUCTURE with its real data volumes.
//
//   * 256 co-resident workgroups, one per CU.  A workgroup reads its XCD id (HW_REG_XCC_ID) and takes a ticket among the workgroups of
//     that XCD: XCD l < 5 is GRU layer l (32 workgroups, 16 hidden units each: 48 gate rows x K = 1024 [x half | h half] of bf16
//     weights, stationary in registers, K split over the four waves); XCDs 5 - 7 are the vocabulary stage (96 workgroups x 104 entries
//     x K = 512, stationary in registers).
//   * A token step of one CHAIN (an independent sub-batch of `rows` sequences) walks the six stages in order:
//       layer l : [off the chain: h half = W_hh . h_l(t-1), the layer's own previous output]  wait for stage l-1 of this step (32 arrivals;
//                 layer 0: the 96 vocabulary workgroups of the previous step), read the rows x 512 input (layer 0: the arg-max keys, then a
//                 gather of rows x 1 KB from the embedding table), x half MFMAs, K-reduction through LDS, gates, publish rows x 16 units
//                 (sc1 write-through stores into a buffer that is FRESH for every (step, stage, chain): no stale line can exist in any L1 /
//                 L2, so consumers need no cache maintenance), one arrival on the stage's counter;
//       vocab   : wait for layer 4 (32 arrivals), read rows x 512, 104-entry logits, per-row arg-max key -> one atomicMax per row, arrival.
//     The hand-off domain is one XCD-to-XCD edge (32 producers -> 32 or 96 consumers), never the chip.
//   * NCH independent chains (the batch cut into NCH sub-batches of 128 / NCH sequences: sequences ARE independent) keep several XCDs
//     busy at once: every workgroup serves the chains round-robin.
// Prints us per token step of the whole 128-sequence batch for NCH = 1, 2, 4, 8.  Every spin is bounded.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/xcd_pipeline_probe tools/xcd_pipeline_probe.hip && tools/bin/xcd_pipeline_probe [steps]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

constexpr int G = 256, H = 512, V = 9984, B = 128;          // workgroups, hidden size, vocabulary (96 x 104), batch
constexpr int PIXB = 2 * H + 32;                           // padded LDS row (bytes)
constexpr int SPIN_LIMIT = 1 << 21;
constexpr int NSTAGE = 6;

struct Args {
  bf16_t* act;             // [step][stage 0..4][chain] rows x H : layer outputs (fresh per use)
  unsigned long long* keys;// [step][chain][rows] arg-max keys (value | index)
  unsigned* cnt;           // [step][stage][chain] x 32 dwords (own 128-byte line)
  unsigned* ticket;        // [8] per-XCD tickets
  const bf16_t* w;         // per-workgroup weight fragments (any values)
  const bf16_t* emb;       // [V][H] embedding table
  int nsteps, nch, rows;   // rows per chain
  unsigned long long* t_out; int* err;
};

__device__ __forceinline__ uint32_t pack2(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) float f2; typedef __attribute__((ext_vector_type(2))) __bf16 b2;
  const b2 v = __builtin_convertvector(f2{a, b}, b2);
  return *reinterpret_cast<const uint32_t*>(&v);
}
__device__ __forceinline__ f32x4 mfma(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
}
__device__ __forceinline__ void store_sc1_x4(void* p, const u32x4& v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }

// wait until *c >= want (relaxed agent polls by one wave, bounded); returns false on timeout / abort
__device__ __forceinline__ bool wait_count(const unsigned* c, unsigned want, int* err, int wid, int lane, int* abort_flag) {
  if (wid == 0) {
    int spins = 0;
    while (true) {
      const unsigned v = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v >= want) break;
      if (++spins > SPIN_LIMIT || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { if (lane == 0) { *abort_flag = 1; *err = 1; } break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  return *abort_flag == 0;
}

// ROWS: sequences per chain (128 / NCH)
template <int ROWS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void probe(Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int abort_flag, s_role;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r16 = lane & 15, q4 = lane >> 4;
  constexpr int TILES = ROWS / 16;
  if (tid == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7;
    const unsigned li = atomicAdd(&a.ticket[xcc], 1u);
    s_role = (int)(xcc * 64 + li);                         // li < 32 when the dispatcher dealt the grid evenly (checked by the host)
    abort_flag = 0;
  }
  __syncthreads();
  const int xcc = s_role >> 6, li = s_role & 63;
  const int stage = xcc < 5 ? xcc : 5;
  const int vi = (xcc - 5) * 32 + li;                      // vocabulary workgroup index 0..95
  if (li >= 32) { if (tid == 0) *a.err = 2; return; }      // uneven placement: the host reports it

  // ---- weights stationary in registers: K split over the four waves --------------------------------------------------------------
  //   layer: 3 gate tiles (48 gate rows) x 8 K-steps per wave (4 of the x half, 4 of the h half)        -> 96 registers
  //   vocab: 7 entry tiles (104 entries, 112 padded) x 4 K-steps per wave                                -> 112 registers
  u32x4 wq[28];
  const u32x4* wl = reinterpret_cast<const u32x4*>(a.w) + ((size_t)blockIdx.x * 4 + wid) * 28 * 64 + lane;
#pragma unroll
  for (int i = 0; i < 28; ++i) wq[i] = wl[i * 64];

  const size_t act_stage = (size_t)a.nch * ROWS * H;       // elements per (step, stage)
  auto act_buf = [&](int t, int st, int c) { return a.act + ((size_t)t * 5 + st) * act_stage + (size_t)c * ROWS * H; };
  auto cnt_of = [&](int t, int st, int c) { return a.cnt + (((size_t)t * NSTAGE + st) * a.nch + c) * 32; };
  auto load_rows = [&](const bf16_t* src) {                // rows x 512 -> LDS (padded rows), all loads in flight first
    constexpr int NL = ROWS * 64 / 256, NB = NL < 16 ? NL : 16;      // at most 16 loads (64 registers) in flight per thread
#pragma unroll
    for (int i0 = 0; i0 < NL; i0 += NB) {
      u32x4 v[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) { const int q = tid + 256 * (i0 + i); v[i] = *reinterpret_cast<const u32x4*>(src + (size_t)(q >> 6) * H + (q & 63) * 8); }
#pragma unroll
      for (int i = 0; i < NB; ++i) { const int q = tid + 256 * (i0 + i); *reinterpret_cast<u32x4*>(smem + (q >> 6) * PIXB + (q & 63) * 16) = v[i]; }
    }
  };

  unsigned long long t0 = 0;
  if (tid == 0) t0 = __builtin_amdgcn_s_memrealtime();
  bool ok = true;
  for (int t = 0; t < a.nsteps && ok; ++t) {
    for (int c = 0; c < a.nch && ok; ++c) {
      if (stage < 5) {
        // ================= GRU layer `stage`, chain c, step t ==========================================================================
        f32x4 acc[TILES][3];
#pragma unroll
        for (int i = 0; i < TILES; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // h half, OFF the token chain: the layer's own output of the previous step (same XCD)
        if (t > 0) {
          load_rows(act_buf(t - 1, stage, c));
          __syncthreads();
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < TILES; ++i) {
              const u32x4 fa = *reinterpret_cast<const u32x4*>(smem + (i * 16 + r16) * PIXB + (wid * 4 + kk) * 64 + q4 * 16);
#pragma unroll
              for (int j = 0; j < 3; ++j) acc[i][j] = mfma(wq[12 + kk * 3 + j], fa, acc[i][j]);
            }
          __syncthreads();
        }
        // x half, ON the chain: wait for the producer stage
        if (stage == 0) {
          if (t > 0) {
            ok = wait_count(cnt_of(t - 1, 5, c), 96u, a.err, wid, lane, &abort_flag);
            if (!ok) break;
            // arg-max keys of the previous step -> token ids -> gather rows x 1 KB from the embedding table
            constexpr int NL = ROWS * 64 / 256, NB = NL < 16 ? NL : 16;
#pragma unroll
            for (int i0 = 0; i0 < NL; i0 += NB) {
              u32x4 v[NB];
#pragma unroll
              for (int i = 0; i < NB; ++i) {
                const int q = tid + 256 * (i0 + i), row = q >> 6;
                const unsigned long long key = __hip_atomic_load(&a.keys[((size_t)(t - 1) * a.nch + c) * ROWS + row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned id = (unsigned)(key & 0xffffffffu) % (unsigned)V;
                v[i] = *reinterpret_cast<const u32x4*>(a.emb + (size_t)id * H + (q & 63) * 8);
              }
#pragma unroll
              for (int i = 0; i < NB; ++i) { const int q = tid + 256 * (i0 + i); *reinterpret_cast<u32x4*>(smem + (q >> 6) * PIXB + (q & 63) * 16) = v[i]; }
            }
          } else load_rows(a.emb + (size_t)c * ROWS * H);
        } else {
          ok = wait_count(cnt_of(t, stage - 1, c), 32u, a.err, wid, lane, &abort_flag);
          if (!ok) break;
          load_rows(act_buf(t, stage - 1, c));
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int i = 0; i < TILES; ++i) {
            const u32x4 fa = *reinterpret_cast<const u32x4*>(smem + (i * 16 + r16) * PIXB + (wid * 4 + kk) * 64 + q4 * 16);
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = mfma(wq[kk * 3 + j], fa, acc[i][j]);
          }
        __syncthreads();
        // K-reduction of the four waves through LDS: [wave][tile][gate tile][lane] f32x4
        f32x4* red = reinterpret_cast<f32x4*>(smem);
#pragma unroll
        for (int i = 0; i < TILES; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) red[((wid * TILES + i) * 3 + j) * 64 + lane] = acc[i][j];
        __syncthreads();
        // gates + publish: wave w takes row tiles w, w + 4, ..; lane (r16, q4) owns units 4 q4 .. 4 q4 + 3 of row r16 (8 bytes)
        for (int i = wid; i < TILES; i += 4) {
          f32x4 g3[3];
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            g3[j] = red[((0 * TILES + i) * 3 + j) * 64 + lane];
#pragma unroll
            for (int w2 = 1; w2 < 4; ++w2) { const f32x4 p = red[((w2 * TILES + i) * 3 + j) * 64 + lane]; g3[j] += p; }
          }
          float hn[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float r = g3[0][e] * 0.02f, z = g3[1][e] * 0.02f, n = g3[2][e] * 0.02f;
            const float rs = r * __builtin_amdgcn_rcpf(1.f + fabsf(r)), zs = 0.5f + 0.5f * z * __builtin_amdgcn_rcpf(1.f + fabsf(z));
            const float nn = (n + rs) * __builtin_amdgcn_rcpf(1.f + fabsf(n + rs));
            hn[e] = (1.f - zs) * nn + zs * 0.25f;
          }
          const u32x2 o = u32x2{pack2(hn[0], hn[1]), pack2(hn[2], hn[3])};
          bf16_t* dst = act_buf(t, stage, c) + (size_t)(i * 16 + r16) * H + li * 16 + q4 * 4;
          asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(dst), "v"(o) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(cnt_of(t, stage, c), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        // ================= vocabulary stage: 104 entries of this workgroup, chain c, step t ================================================
        ok = wait_count(cnt_of(t, 4, c), 32u, a.err, wid, lane, &abort_flag);
        if (!ok) break;
        constexpr int RP = ROWS < 64 ? ROWS : 64;                             // rows per LDS pass (128 rows + the reduction buffer would not fit)
        f32x4* red = reinterpret_cast<f32x4*>(smem + RP * PIXB);             // [wave][7][lane] f32x4 per row tile
        for (int i = 0; i < TILES; ++i) {                                     // one 16-row tile at a time: 7 accumulators
          if (i % (RP / 16) == 0) {
            constexpr int NL = RP * 64 / 256;
            const bf16_t* src = act_buf(t, 4, c) + (size_t)i * 16 * H;
            u32x4 v[NL];
#pragma unroll
            for (int k = 0; k < NL; ++k) { const int q = tid + 256 * k; v[k] = *reinterpret_cast<const u32x4*>(src + (size_t)(q >> 6) * H + (q & 63) * 8); }
#pragma unroll
            for (int k = 0; k < NL; ++k) { const int q = tid + 256 * k; *reinterpret_cast<u32x4*>(smem + (q >> 6) * PIXB + (q & 63) * 16) = v[k]; }
            __syncthreads();
          }
          const int il = i % (RP / 16);
          f32x4 acc[7];
#pragma unroll
          for (int j = 0; j < 7; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const u32x4 fa = *reinterpret_cast<const u32x4*>(smem + (il * 16 + r16) * PIXB + (wid * 4 + kk) * 64 + q4 * 16);
#pragma unroll
            for (int j = 0; j < 7; ++j) acc[j] = mfma(wq[kk * 7 + j], fa, acc[j]);
          }
#pragma unroll
          for (int j = 0; j < 7; ++j) red[(wid * 7 + j) * 64 + lane] = acc[j];
          __syncthreads();
          if (wid == (i & 3)) {                                                // one wave finishes the tile: K-sum, max over its 112 entries per row
            float best = -3.0e38f; unsigned bidx = 0;
#pragma unroll
            for (int j = 0; j < 7; ++j) {
              f32x4 s = red[(0 * 7 + j) * 64 + lane];
#pragma unroll
              for (int w2 = 1; w2 < 4; ++w2) { const f32x4 p = red[(w2 * 7 + j) * 64 + lane]; s += p; }
#pragma unroll
              for (int e = 0; e < 4; ++e) if (s[e] > best) { best = s[e]; bidx = (unsigned)(vi * 104 + j * 16 + q4 * 4 + e); }
            }
            // lanes with equal r16 (same sequence) hold different entries: reduce over q4 with two shuffles
#pragma unroll
            for (int o = 16; o < 64; o <<= 1) {
              const float ob = __shfl_xor(best, o, 64); const unsigned oi = (unsigned)__shfl_xor((int)bidx, o, 64);
              if (ob > best) { best = ob; bidx = oi; }
            }
            if (q4 == 0) {
              unsigned fb = __float_as_uint(best); fb = (fb & 0x80000000u) ? ~fb : (fb | 0x80000000u);   // order-preserving key
              atomicMax(&a.keys[((size_t)t * a.nch + c) * ROWS + i * 16 + r16], ((unsigned long long)fb << 32) | bidx);
            }
          }
          __syncthreads();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(cnt_of(t, 5, c), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (tid == 0) a.t_out[blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int ROWS>
int run(int nsteps, int nch) {
  Args a{};
  const size_t act_elems = (size_t)nsteps * 5 * nch * ROWS * H;
  CK(hipMalloc(&a.act, act_elems * 2)); CK(hipMemset(a.act, 0, act_elems * 2));
  CK(hipMalloc(&a.keys, (size_t)nsteps * nch * ROWS * 8));
  CK(hipMalloc(&a.cnt, (size_t)nsteps * NSTAGE * nch * 128));
  CK(hipMalloc(&a.ticket, 8 * 4));
  bf16_t *w, *emb;
  CK(hipMalloc(&w, (size_t)G * 4 * 28 * 64 * 16)); CK(hipMalloc(&emb, (size_t)V * H * 2));
  std::vector<uint16_t> hw((size_t)G * 4 * 28 * 64 * 8), he((size_t)V * H);
  srand(1);
  for (auto& x : hw) x = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
  for (auto& x : he) x = 0x3c00 + (rand() & 0x1ff);
  CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(emb, he.data(), he.size() * 2, hipMemcpyHostToDevice));
  a.w = w; a.emb = emb; a.nsteps = nsteps; a.nch = nch; a.rows = ROWS;
  CK(hipMalloc(&a.t_out, G * 8)); CK(hipMalloc(&a.err, 4));
  const int lds_layer = ROWS * PIXB > 4 * (ROWS / 16) * 3 * 64 * 16 ? ROWS * PIXB : 4 * (ROWS / 16) * 3 * 64 * 16;
  const int lds_vocab = (ROWS < 64 ? ROWS : 64) * PIXB + 4 * 7 * 64 * 16;
  const int lds = lds_layer > lds_vocab ? lds_layer : lds_vocab;
  if (lds > 160 * 1024 - 64) { printf("LDS %d too large\n", lds); return 1; }
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<ROWS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemset(a.keys, 0, (size_t)nsteps * nch * ROWS * 8)); CK(hipMemset(a.cnt, 0, (size_t)nsteps * NSTAGE * nch * 128));
    CK(hipMemset(a.ticket, 0, 32)); CK(hipMemset(a.err, 0, 4));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<ROWS>, dim3(G), dim3(256), lds, 0, a);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    int herr = 0; CK(hipMemcpy(&herr, a.err, 4, hipMemcpyDeviceToHost));
    if (herr) { printf("chains %d: %s\n", nch, herr == 2 ? "the dispatcher did not deal 32 workgroups to every XCD" : "SPIN LIMIT HIT (not all workgroups co-resident?)"); return 1; }
    if (rep > 0 && ms < best) best = ms;
  }
  printf("chains %d x %3d sequences: %d token steps in %8.1f us: %6.2f us per step of the 128-sequence batch (%5.2f us per stage visit)\n",
         nch, ROWS, nsteps, best * 1e3, best * 1e3 / nsteps, best * 1e3 / nsteps / (6.0 * nch));
  CK(hipFree(a.act)); CK(hipFree(a.keys)); CK(hipFree(a.cnt)); CK(hipFree(a.ticket)); CK(hipFree(w)); CK(hipFree(emb)); CK(hipFree(a.t_out)); CK(hipFree(a.err));
  return 0;
}

int main(int argc, char** argv) {
  const int nsteps = argc > 1 ? atoi(argv[1]) : 25;
  printf("# layer-per-XCD pipeline probe: 5 GRU layers on XCDs 0-4 (32 workgroups each), vocabulary stage on XCDs 5-7 (96 workgroups), B = 128\n");
  if (run<128>(nsteps, 1)) return 1;
  if (run<64>(nsteps, 2)) return 1;
  if (run<32>(nsteps, 4)) return 1;
  if (run<16>(nsteps, 8)) return 1;
  return 0;
}
