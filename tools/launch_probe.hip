// Where do the 8-10 us per launch go that rocprof shows around the one-workgroup-per-CU convolution kernels but no workgroup
// spends (Weak 5: astat 20 us first-entry -> last-exit against 28.5-30.5 us per launch)?  This probe launches EMPTY
// and STORE-ONLY kernels with the resource shapes of those kernels and reports, per configuration:
//   * us per launch (hipEvents around 200 back-to-back launches on one stream) -- what rocprof's tiling durations add up to;
//   * in-kernel span   = last exit - first entry of a launch (s_memrealtime, 100 MHz), median over launches;
//   * inter-kernel gap = first entry of launch k+1 - last exit of launch k, median.
// Configurations: dynamic LDS 0 / 80 / 160 KB x registers small / 256 / 512 per lane (waves_per_eu(1,1) + clobbers), grids of 256
// and 512 workgroups; the same with each workgroup writing 200 KB (51 MB per launch, the 256 -> 1024 conv3 output) with plain,
// nt and sc1 (write-through) 16-byte stores -- does the end-of-kernel L2 write-back show up as launch time, and do write-through
// stores move it under the kernel?  And a straight-line code body of ~100 KB executed once per wave (instruction fetch of the fully
// unrolled K loops).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/launch_probe tools/launch_probe.hip && tools/bin/launch_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args {
  unsigned long long* stamps;   // [launch][2]: min first entry, max last exit (atomics)
  int launch;
  u32x4* out; int bytes_per_wg; int policy;   // 0 plain, 1 nt, 2 sc1
  int spin;                     // extra in-kernel work: s_sleep iterations (a kernel body of known length)
};

template <int POL> __device__ __forceinline__ void store16(u32x4* base, size_t idx, u32x4 v) {
  if constexpr (POL == 0) base[idx] = v;
  else if constexpr (POL == 1) __builtin_nontemporal_store(v, base + idx);
  else {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(idx * 16), 0, 16);   // aux 16 = sc1
  }
}

template <int REGS, int BIGCODE>
__device__ __forceinline__ void body(const Args& a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  unsigned long long t0 = 0;
  if (tid == 0) { t0 = __builtin_amdgcn_s_memrealtime(); atomicMin(&a.stamps[a.launch * 2], t0); }
  if constexpr (REGS >= 256) asm volatile("v_mov_b32 v250, 0" ::: "v250");
  if constexpr (REGS >= 512) asm volatile("v_accvgpr_write_b32 a250, 0" ::: "a250");
  smem[tid * 4] = (char)tid;                       // the LDS allocation is real
  float acc = (float)lane;
  if constexpr (BIGCODE > 0) {                     // BIGCODE x 8 dependent v_fma: 64 bytes of code each, executed once
#pragma unroll
    for (int i = 0; i < BIGCODE; ++i) {
      asm volatile("v_fma_f32 %0, %0, 1.0, %0\n v_fma_f32 %0, %0, 1.0, %0\n v_fma_f32 %0, %0, 1.0, %0\n v_fma_f32 %0, %0, 1.0, %0\n"
                   "v_fma_f32 %0, %0, 1.0, %0\n v_fma_f32 %0, %0, 1.0, %0\n v_fma_f32 %0, %0, 1.0, %0\n v_fma_f32 %0, %0, 1.0, %0" : "+v"(acc));
    }
  }
  for (int i = 0; i < a.spin; ++i) __builtin_amdgcn_s_sleep(64);
  if (a.bytes_per_wg > 0) {
    const int n16 = a.bytes_per_wg / 16;
    u32x4 v = {(uint32_t)tid, (uint32_t)blockIdx.x, __float_as_uint(acc), 7u};
    const size_t base = (size_t)blockIdx.x * n16;
    if (a.policy == 0) for (int i = tid; i < n16; i += 256) store16<0>(a.out, base + i, v);
    else if (a.policy == 1) for (int i = tid; i < n16; i += 256) store16<1>(a.out, base + i, v);
    else for (int i = tid; i < n16; i += 256) store16<2>(a.out, base + i, v);
  }
  __syncthreads();
  if (tid == 0) {
    if (acc == 12345.678f) a.out[0] = u32x4{1, 2, 3, 4};
    atomicMax(&a.stamps[a.launch * 2 + 1], __builtin_amdgcn_s_memrealtime());
  }
}

__global__ __launch_bounds__(256) void k_small(Args a) { body<0, 0>(a); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_256(Args a) { body<256, 0>(a); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_512(Args a) { body<512, 0>(a); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_512_code(Args a) { body<512, 1600>(a); }   // ~100 KB
__global__ __launch_bounds__(256) void k_small_code(Args a) { body<0, 1600>(a); }

typedef void (*kern_t)(Args);

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; }

int main() {
  const int NL = 200;
  unsigned long long* stamps;
  CK(hipMalloc(&stamps, NL * 2 * sizeof(unsigned long long)));
  u32x4* out;
  const size_t out_bytes = 512ull * 256 * 1024;
  CK(hipMalloc(&out, out_bytes));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Cfg { const char* name; kern_t k; int lds, grid, bytes, policy, spin; };
  const int KB = 1024;
  std::vector<Cfg> cfgs = {
    {"small regs, LDS 0, 256 WG", k_small, 0, 256, 0, 0, 0},
    {"small regs, LDS 0, 1024 WG", k_small, 0, 1024, 0, 0, 0},
    {"small regs, LDS 160K, 256 WG", k_small, 160 * KB, 256, 0, 0, 0},
    {"256 regs, LDS 80K, 512 WG", k_256, 80 * KB, 512, 0, 0, 0},
    {"512 regs, LDS 0, 256 WG", k_512, 0, 256, 0, 0, 0},
    {"512 regs, LDS 160K, 256 WG", k_512, 160 * KB, 256, 0, 0, 0},
    {"512 regs, LDS 160K, 224 WG", k_512, 160 * KB, 224, 0, 0, 0},
    {"512 regs, LDS 160K, 256 WG, ~20 us body", k_512, 160 * KB, 256, 0, 0, 24},
    {"512 regs, LDS 160K, 256 WG, 100 KB straight-line code", k_512_code, 160 * KB, 256, 0, 0, 0},
    {"small regs, LDS 0, 256 WG, 100 KB straight-line code", k_small_code, 0, 256, 0, 0, 0},
    {"512 regs, LDS 160K, 256 WG, 200 KB/WG plain stores", k_512, 160 * KB, 256, 200 * KB, 0, 0},
    {"512 regs, LDS 160K, 256 WG, 200 KB/WG nt stores", k_512, 160 * KB, 256, 200 * KB, 1, 0},
    {"512 regs, LDS 160K, 256 WG, 200 KB/WG sc1 stores", k_512, 160 * KB, 256, 200 * KB, 2, 0},
    {"512 regs, LDS 160K, 256 WG, 50 KB/WG plain stores", k_512, 160 * KB, 256, 50 * KB, 0, 0},
    {"512 regs, LDS 160K, 256 WG, 50 KB/WG sc1 stores", k_512, 160 * KB, 256, 50 * KB, 2, 0},
    {"512 regs, LDS 160K, ~20 us body + 200 KB/WG plain", k_512, 160 * KB, 256, 200 * KB, 0, 24},
    {"512 regs, LDS 160K, ~20 us body + 200 KB/WG sc1", k_512, 160 * KB, 256, 200 * KB, 2, 24},
    {"small regs, LDS 0, 1024 WG, 50 KB/WG plain stores", k_small, 0, 1024, 50 * KB, 0, 0},
    {"small regs, LDS 0, 1024 WG, 50 KB/WG sc1 stores", k_small, 0, 1024, 50 * KB, 2, 0},
  };
  for (kern_t k : {k_small, k_256, k_512, k_512_code, k_small_code})
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * KB));
  printf("%-64s %10s %10s %10s\n", "configuration", "us/launch", "span us", "gap us");
  for (const Cfg& c : cfgs) {
    if ((size_t)c.grid * c.bytes > out_bytes) { printf("%s: output too large\n", c.name); continue; }
    for (int rep = 0; rep < 2; ++rep) {              // first pass warms up
      CK(hipMemsetAsync(stamps, 0xff, NL * sizeof(unsigned long long) * 2, st));
      // exits start at 0: per launch slot [2k] = min entry (init all ones), [2k+1] = max exit (init 0)
      std::vector<unsigned long long> init(NL * 2);
      for (int i = 0; i < NL; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0ull; }
      CK(hipMemcpyAsync(stamps, init.data(), NL * 2 * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      for (int l = 0; l < NL; ++l) {
        Args a{stamps, l, out, c.bytes, c.policy, c.spin};
        hipLaunchKernelGGL(c.k, dim3(c.grid), dim3(256), c.lds, st, a);
      }
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      CK(hipGetLastError());
      if (rep == 0) continue;
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<unsigned long long> h(NL * 2);
      CK(hipMemcpy(h.data(), stamps, NL * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      std::vector<double> span, gap;
      for (int l = 1; l < NL; ++l) {
        span.push_back((double)(h[2 * l + 1] - h[2 * l]) * 0.01);
        gap.push_back((double)((long long)h[2 * l] - (long long)h[2 * l - 1]) * 0.01);
      }
      printf("%-64s %10.2f %10.2f %10.2f\n", c.name, ms * 1e3 / NL, median(span), median(gap));
      fflush(stdout);
    }
  }
  return 0;
}
