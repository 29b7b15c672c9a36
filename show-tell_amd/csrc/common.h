// Shared device/host helpers for libshowtell_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/showtell_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

void st_set_error(const char* fmt, ...);

#define ST_CHECK(cond, ...)                                                   \
  do {                                                                        \
    if (!(cond)) { st_set_error(__VA_ARGS__); return 1; }                     \
  } while (0)

#define ST_LAUNCH_CHECK()                                                     \
  do {                                                                        \
    hipError_t e_ = hipGetLastError();                                        \
    if (e_ != hipSuccess) { st_set_error("%s:%d launch: %s", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } \
  } while (0)

static inline size_t st_dtype_size(int dt) { return dt == ST_BF16 ? 2 : 4; }

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
  bf16_t b = (bf16_t)f;  // v_cvt_pk_bf16_f32: RNE, NaN-preserving
  return *reinterpret_cast<uint16_t*>(&b);
}
typedef __attribute__((ext_vector_type(2))) float f32x2_;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
typedef __attribute__((ext_vector_type(2))) short i16x2_;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  // ONE v_cvt_pk_bf16_f32 (RNE, NaN-preserving); two scalar casts + shift + or compiled to four instructions
  const bf16x2_ b = __builtin_convertvector(f32x2_{lo, hi}, bf16x2_);
  return *reinterpret_cast<const uint32_t*>(&b);
}

// Output stores of the streaming / convolution kernels (data the storing kernel never reads again).  ST_STORE_POLICY is a BUILD knob
// (make CXXFLAGS+=-DST_STORE_POLICY=1): 0 = plain stores (lines stay dirty in the XCD's L2 and are written back at the end of the
// kernel, which the next kernel on the stream waits for: tools/launch_probe.hip measured 1.1 - 2.6 us per launch for 13 - 51 MB
// outputs), 1 = `sc1` write-through stores (the bytes leave L2 while the kernel is still computing), 2 = `nt`.  `base` must be
// wave-uniform (it becomes a buffer resource in SGPRs), the byte offset is per lane and must stay below 2^31.
#ifndef ST_STORE_POLICY
#define ST_STORE_POLICY 0
#endif
__device__ __forceinline__ void st_out_store16(void* base, long off_bytes, const u32x4& v) {
#if ST_STORE_POLICY == 1
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)off_bytes, 0, 16);      // aux 16 = sc1
#elif ST_STORE_POLICY == 2
  __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(reinterpret_cast<char*>(base) + off_bytes));
#else
  *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(base) + off_bytes) = v;
#endif
}
__device__ __forceinline__ void st_out_store8(void* base, long off_bytes, const u32x2& v) {
#if ST_STORE_POLICY == 1
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b64(v, rs, (int)off_bytes, 0, 16);
#elif ST_STORE_POLICY == 2
  __builtin_nontemporal_store(v, reinterpret_cast<u32x2*>(reinterpret_cast<char*>(base) + off_bytes));
#else
  *reinterpret_cast<u32x2*>(reinterpret_cast<char*>(base) + off_bytes) = v;
#endif
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// Scale / shift of a train-mode BatchNorm from a channel's [sum, sumsq].  ONE definition for every kernel that normalises (st_bn_act, the
// convolution loaders, the fused block-boundary kernels), written with EXPLICIT fused multiply-adds: under -ffp-contract=fast the
// backend fuses `sumsq * inv - mean * mean` either way round depending on the surrounding code (a `#pragma clang fp contract(off)` does
// not stop it), two kernels then disagree in the last bit of the scale, and that flips a bf16 rounding about once per million elements --
// enough to break the bit-for-bit equivalences the tests assert between the fused and the separate forms.
__device__ __forceinline__ void bn_scale_shift(float sum, float sumsq, float inv_count, float gamma, float beta, float eps, float& sc, float& sh) {
  const float mean = sum * inv_count;
  const float var = fmaxf(__builtin_fmaf(-mean, mean, sumsq * inv_count), 0.f);
  const float s = gamma * rsqrtf(var + eps);
  sc = s;
  sh = __builtin_fmaf(-mean, s, beta);
}

// [sum, sumsq] of channel c over `rep` replicas of a [rep][2C] statistics block, added in replica order (the order every consumer
// uses: the scale / shift of a layer must come out bit-identical in all of them).  Four replicas per round trip: the eight loads of a
// batch are unconditional (replica index clamped, the surplus masked to zero), so a thread pays ONE memory latency per four replicas --
// the rolled `for (r < rep) { sm += ..; sq += ..; }` it replaces paid one per replica (1 - 2 us of every kernel prologue at rep = 4).
__device__ __forceinline__ void stat_sum(const float* stats, int rep, int C, int c, float& sm, float& sq) {
  sm = 0.f; sq = 0.f;
  for (int r0 = 0; r0 < rep; r0 += 4) {
    float a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = r0 + q < rep ? r0 + q : rep - 1;
      a[q] = stats[(size_t)r * 2 * C + c]; b[q] = stats[(size_t)r * 2 * C + C + c];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { sm += r0 + q < rep ? a[q] : 0.f; sq += r0 + q < rep ? b[q] : 0.f; }
  }
}
// Split form for kernel prologues: `stat_head_issue` REQUESTS the first four replicas of one channel (plus gamma / beta) -- call it before
// the kernel's tile / filter loads so that these few dwords head the in-order load queue; `stat_head_finish` adds them up in replica
// order (replicas beyond four are fetched there: the engine uses <= 4) and returns the BatchNorm scale / shift.  Consuming them waits for
// them alone, not for the tile loads issued in between, and the coefficient arithmetic overlaps the tiles' flight.
struct StatHead { float a[4], b[4], g, be; };
__device__ __forceinline__ void stat_head_issue(StatHead& h, const float* stats, int rep, int C, int c, const float* gamma, const float* beta) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = q < rep ? q : rep - 1;
    h.a[q] = stats[(size_t)r * 2 * C + c]; h.b[q] = stats[(size_t)r * 2 * C + C + c];
  }
  h.g = gamma[c]; h.be = beta[c];
}
__device__ __forceinline__ void stat_head_finish(const StatHead& h, const float* stats, int rep, int C, int c, float inv_count, float eps, float& sc, float& sh) {
  float sm = 0.f, sq = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) { sm += q < rep ? h.a[q] : 0.f; sq += q < rep ? h.b[q] : 0.f; }
  for (int r = 4; r < rep; ++r) { sm += stats[(size_t)r * 2 * C + c]; sq += stats[(size_t)r * 2 * C + C + c]; }
  bn_scale_shift(sm, sq, inv_count, h.g, h.be, eps, sc, sh);
}

// the same for NCH channels c0, c0 + cs, .. of one thread at once (all 8 NCH loads of a batch in flight together)
template <int NCH>
__device__ __forceinline__ void stat_sums(const float* stats, int rep, int C, int c0, int cs, float (&sm)[NCH], float (&sq)[NCH]) {
#pragma unroll
  for (int k = 0; k < NCH; ++k) { sm[k] = 0.f; sq[k] = 0.f; }
  for (int r0 = 0; r0 < rep; r0 += 4) {
    float a[NCH][4], b[NCH][4];
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = r0 + q < rep ? r0 + q : rep - 1;
        a[k][q] = stats[(size_t)r * 2 * C + c0 + k * cs]; b[k][q] = stats[(size_t)r * 2 * C + C + c0 + k * cs];
      }
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
      for (int q = 0; q < 4; ++q) { sm[k] += r0 + q < rep ? a[k][q] : 0.f; sq[k] += r0 + q < rep ? b[k][q] : 0.f; }
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
