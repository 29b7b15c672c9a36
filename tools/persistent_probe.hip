// What would a PERSISTENT greedy-decode kernel cost per step on MI355X?  (the measured number behind the
// decision not to build it.)  This probe is not the decoder: it runs the decoder's DEPENDENCE STRUCTURE with the real data volumes --
//   * 256 workgroups (one per CU, all co-resident), weights stationary in registers;
//   * a "stage" = every workgroup waits until ALL workgroups have published the previous stage (one agent-scope counter per stage,
//     release add / acquire poll), reads the full 128 x 512 bf16 activation matrix of that stage (128 KB, written 1/256 by each
//     workgroup, so it comes from other XCDs' L2s / the Infinity Cache), multiplies it with its own 16 x 512 filter slice
//     (MFMA, 32 per wave), and publishes its 128 x 2 slice of the next activation matrix;
//   * a greedy step of the 5-layer decoder is SIX such stages (5 cells + vocabulary arg-max, whose per-workgroup volumes are
//     the same order: 40 x 512 vocabulary rows per workgroup).
// It prints the time per stage and per 6-stage step.  Every spin is bounded: a workgroup that is not co-resident makes the probe
// report an error instead of hanging.
//   hipcc --offload-arch=gfx950 -O3 -o persistent_probe tools/persistent_probe.hip && ./persistent_probe [stages] [rows]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

constexpr int G = 256, H = 512, PIXB = 2 * H + 32;        // workgroups, hidden size, padded LDS row (bytes)
constexpr int SPIN_LIMIT = 1 << 22;
constexpr int NREP = 8;                                   // counter shards per stage (same-address atomics serialise at the memory side)

__device__ __forceinline__ uint32_t pack2(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) float f2; typedef __attribute__((ext_vector_type(2))) __bf16 b2;
  const b2 v = __builtin_convertvector(f2{a, b}, b2);
  return *reinterpret_cast<const uint32_t*>(&v);
}

template <int ROWS>
__global__ __launch_bounds__(256) void probe(bf16_t* act0, bf16_t* act1, const bf16_t* w, unsigned* counters, int nstages, int sync,
                                              unsigned long long* t_out, int* err) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r16 = lane & 15, q4 = lane >> 4;
  const int b = blockIdx.x;
  constexpr int TILES = ROWS / 16, TPW = TILES / 4;       // 16-row tiles, per wave
  // filter slice: 16 "gate rows" x 512 K as 16 MFMA operands, in registers for the whole run
  u32x4 wq[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) wq[ks] = reinterpret_cast<const u32x4*>(w)[((size_t)b * 16 + ks) * 64 + lane];
  unsigned long long t0 = 0;
  if (tid == 0) t0 = __builtin_amdgcn_s_memrealtime();
  __shared__ int abort_flag;
  if (tid == 0) abort_flag = 0;
  __syncthreads();
  int bad = 0;
  for (int s = 0; s < nstages; ++s) {
    const bf16_t* src = (s & 1) ? act1 : act0;
    bf16_t* dst = (s & 1) ? act0 : act1;
    if (s > 0 && sync) {
      if (wid == 0) {                                      // NREP sharded counters (one 128-byte line each): lane r polls shard r
        int spins = 0;
        bool wait = true;
        while (wait) {
          unsigned c = (unsigned)(G / NREP);
          if (lane < NREP) c = __hip_atomic_load(&counters[((size_t)(s - 1) * NREP + lane) * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          wait = __ballot(c < (unsigned)(G / NREP)) != 0;
          if (wait) {
            if (++spins > SPIN_LIMIT || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { bad = 1; if (lane == 0) { abort_flag = 1; *err = 1; } break; }
            __builtin_amdgcn_s_sleep(1);
          }
        }
      }
      if (sync == 3 && wid == 0) {                          // the polling wave invalidates once; its wait holds the barrier until that is done
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      if (abort_flag) break;                               // a partner never arrived: leave (every workgroup sees err and leaves too)
      if (sync == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // ONE invalidate per wave after the poll has matched (sync == 2: sc1 loads instead)
    }
    // the whole activation matrix -> LDS (padded rows): ROWS x 64 chunks of 16 bytes, 256 threads
    constexpr int NL = ROWS * 64 / 256;
    u32x4 v[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int q = tid + 256 * i, row = q >> 6, c = q & 63;
      if (sync == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[i]) : "v"(src + (size_t)row * H + c * 8) : "memory");
      else v[i] = *reinterpret_cast<const u32x4*>(src + (size_t)row * H + c * 8);
    }
    if (sync == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int q = tid + 256 * i, row = q >> 6, c = q & 63;
      *reinterpret_cast<u32x4*>(smem + row * PIXB + c * 16) = v[i];
    }
    __syncthreads();
    f32x4 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        const u32x4 fa = *reinterpret_cast<const u32x4*>(smem + ((wid * TPW + i) * 16 + r16) * PIXB + ks * 64 + q4 * 16);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&wq[ks]), *reinterpret_cast<const bf16x8*>(&fa), acc[i], 0, 0, 0);
      }
    // publish this workgroup's 2 columns of the next matrix (rows of D = gate rows 4 q4 + e, columns = matrix rows r16)
    if (q4 == 0) {
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        const float p0 = acc[i][0] * 0.05f + acc[i][2] * 0.01f, p1 = acc[i][1] * 0.05f + acc[i][3] * 0.01f;   // any bounded squashing will do
        const float h0 = p0 * __builtin_amdgcn_rcpf(1.f + fabsf(p0)), h1 = p1 * __builtin_amdgcn_rcpf(1.f + fabsf(p1));
        uint32_t* o = reinterpret_cast<uint32_t*>(dst + (size_t)((wid * TPW + i) * 16 + r16) * H + 2 * b);
        const uint32_t pv = pack2(h0, h1);
        if (sync >= 2) asm volatile("global_store_dword %0, %1, off sc1" :: "v"(o), "v"(pv) : "memory");
        else *o = pv;
      }
    }
    if (sync >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                       // (also: everyone is done with the LDS image)
    if (tid == 0 && sync) {
      if (sync == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // the workgroup's stores (complete at the barrier) leave this XCD's L2 (sync == 2: sc1 stores)
      __hip_atomic_fetch_add(&counters[((size_t)s * NREP + (b % NREP)) * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (tid == 0) {
    t_out[b] = __builtin_amdgcn_s_memrealtime() - t0;
    if (bad) *err = 1;
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int ROWS>
int run(int nstages) {
  bf16_t *a0, *a1, *w; unsigned* cnt; unsigned long long* t; int* err;
  CK(hipMalloc(&a0, (size_t)ROWS * H * 2)); CK(hipMalloc(&a1, (size_t)ROWS * H * 2));
  CK(hipMalloc(&w, (size_t)G * 16 * 64 * 16)); CK(hipMalloc(&cnt, (size_t)nstages * NREP * 128)); CK(hipMalloc(&t, G * 8)); CK(hipMalloc(&err, 4));
  std::vector<uint16_t> h((size_t)ROWS * H), hw((size_t)G * 16 * 64 * 8);
  srand(1);
  for (auto& x : h) x = 0x3c00 + (rand() & 0x1ff);         // bf16 around 0.01
  for (auto& x : hw) x = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
  CK(hipMemcpy(a0, h.data(), h.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(a1, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  const int lds = ROWS * PIXB;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<ROWS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  int nb = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&probe<ROWS>), 256, lds));
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  if (nb * pr.multiProcessorCount < G) { printf("grid of %d workgroups cannot be co-resident (%d x %d)\n", G, nb, pr.multiProcessorCount); return 1; }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int sync = 3; sync >= 0; --sync)
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(cnt, 0, (size_t)nstages * NREP * 128)); CK(hipMemset(err, 0, 4));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(probe<ROWS>, dim3(G), dim3(256), lds, 0, a0, a1, w, cnt, nstages, sync, t, err);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
      std::vector<unsigned long long> ht(G); CK(hipMemcpy(ht.data(), t, G * 8, hipMemcpyDeviceToHost));
      unsigned long long mx = 0; for (auto x : ht) mx = x > mx ? x : mx;
      if (rep == 2)
        printf("rows %d, %s: %d stages in %.1f us (event) / %.1f us (s_memrealtime, slowest workgroup): %.2f us per stage, %.1f us per 6-stage step%s\n",
               ROWS, sync == 3 ? "hand-off by sc1 stores + ONE agent acquire per workgroup (buffer_inv) + plain loads (shared through the XCD L2)" : sync == 2 ? "hand-off by sc1 stores / sc1 loads + a relaxed agent counter (no cache maintenance)" : sync == 1 ? "hand-off by agent release / acquire fences (buffer_wbl2 / buffer_inv)" : "WITHOUT hand-off (each workgroup free-runs: load + MFMA + store only)", nstages, ms * 1e3,
               mx / 100.0, ms * 1e3 / nstages, ms * 1e3 / nstages * 6, herr ? "  [SPIN LIMIT HIT: not all workgroups were co-resident]" : "");
    }
  return 0;
}

int main(int argc, char** argv) {
  const int nstages = argc > 1 ? atoi(argv[1]) : 150;      // 25 steps x 6 stages
  const int rows = argc > 2 ? atoi(argv[2]) : 128;
  if (rows == 128) return run<128>(nstages);
  if (rows == 64) return run<64>(nstages);
  printf("rows must be 64 or 128\n");
  return 1;
}
