// Soft-attention step kernels (gfx950) for the reference's Attention_Net (Attention/rnn_attn.py:21-31)
// and the doubly-stochastic regulariser of Attention/main_attn.py:131.
//
//   att1 = W_e feat + b_e     (B,P,A)  -- time-invariant: hoisted out of the time loop (one MFMA GEMM per batch;
//                                          the reference recomputes it every timestep, rnn_attn.py:23)
//   att2 = W_d h + b_d        (B,A)    -- skinny MFMA GEMM (rnn_gemm_kernel)
//   e_p  = w_f . lrelu_0.2(att1_p + att2) + b_f ; alpha = softmax_P(e) ; z = sum_p alpha_p feat_p
//
// One 256-thread block per sample: P = 49 pixels fit one 64-lane wavefront for the softmax, the A-dot and
// the F-wide weighted sum are 16-byte coalesced reads of the (P,A) / (P,F) tiles of that sample.
#include "common.h"
#include "attn_kernels.h"

namespace {

constexpr int kMaxP = 64;

template <typename T> struct V16;
template <> struct V16<bf16_t> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const bf16_t* p, float* f) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(v[i] << 16); f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u); }
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
    *reinterpret_cast<u32x4*>(p) = v;
  }
};
template <> struct V16<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void load(const float* p, float* f) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = v[i];
  }
  static __device__ __forceinline__ void store(float* p, const float* f) { *reinterpret_cast<f32x4*>(p) = f32x4{f[0], f[1], f[2], f[3]}; }
};

__device__ __forceinline__ float lrelu(float x) { return x > 0.f ? x : 0.2f * x; }

// (B,F,P) fp32 [cnn_attn.py:49 layout] -> (B,P,F) dtype
template <typename T>
__global__ __launch_bounds__(256) void ncp_to_pf_kernel(const float* __restrict__ x, T* __restrict__ y, int B, int F, int P) {
  extern __shared__ __attribute__((aligned(16))) float tile[];   // [64][P+1]
  const int nslab = (F + 63) / 64;
  const int b = blockIdx.x / nslab, f0 = (blockIdx.x - b * nslab) * 64;
  for (int i = threadIdx.x; i < 64 * P; i += blockDim.x) {
    const int f = i / P, p = i - f * P;
    tile[f * (P + 1) + p] = (f0 + f < F) ? x[((long)b * F + f0 + f) * P + p] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * P; i += blockDim.x) {
    const int p = i >> 6, f = i & 63;
    if (f0 + f < F) y[((long)b * P + p) * F + f0 + f] = from_f32<T>(tile[f * (P + 1) + p]);
  }
}

template <typename T>
__global__ __launch_bounds__(1024) void attn_fwd_kernel(const T* __restrict__ att1, const float* __restrict__ att2,
                                                        const float* __restrict__ wf, const float* __restrict__ bf,
                                                        const T* __restrict__ feat, float* __restrict__ alpha_out, long alpha_stride,
                                                        T* __restrict__ z, int P, int A, int F) {
  __shared__ float e[kMaxP];
  extern __shared__ float zred[];                      // [G][F] partial context vectors (G > 1 only)
  constexpr int N = V16<T>::N;
  const int b = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const T* a1 = att1 + (long)b * P * A;
  const float* a2 = att2 + (long)b * A;
  for (int p = wid; p < P; p += nw) {
    float s = 0.f;
    for (int c = lane * N; c < A; c += 64 * N) {
      float v[N];
      V16<T>::load(a1 + (long)p * A + c, v);
#pragma unroll
      for (int k = 0; k < N; ++k) s += wf[c + k] * lrelu(v[k] + a2[c + k]);
    }
    s = wave_sum(s);
    if (lane == 0) e[p] = s + bf[0];
  }
  __syncthreads();
  if (wid == 0) {
    const float v = lane < P ? e[lane] : -INFINITY;
    const float m = wave_max(v);
    const float ex = lane < P ? expf(v - m) : 0.f;
    const float sum = wave_sum(ex);
    if (lane < P) {
      const float al = ex / sum;
      e[lane] = al;
      alpha_out[(long)b * alpha_stride + lane] = al;
    }
  }
  __syncthreads();
  const T* fb = feat + (long)b * P * F;
  const int nchunk = F / N;                            // 16-byte channel chunks
  const int G = (int)blockDim.x >= 2 * nchunk ? (int)blockDim.x / nchunk : 1;   // pixel groups per chunk
  if (G == 1) {
    for (int c = threadIdx.x * N; c < F; c += blockDim.x * N) {
      float acc[N];
#pragma unroll
      for (int k = 0; k < N; ++k) acc[k] = 0.f;
      for (int p = 0; p < P; ++p) {
        float v[N];
        V16<T>::load(fb + (long)p * F + c, v);
        const float al = e[p];
#pragma unroll
        for (int k = 0; k < N; ++k) acc[k] += al * v[k];
      }
      V16<T>::store(z + (long)b * F + c, acc);
    }
    return;
  }
  // 16 waves, fewer chunks than threads: (chunk, pixel group) per thread; the groups' partial sums meet in LDS and are added
  // in pixel order (group 0 first), so the sum is associated per group, not per pixel
  const int ci = threadIdx.x % nchunk, g = threadIdx.x / nchunk, c = ci * N;
  if (g < G) {
    const int per = (P + G - 1) / G, p0 = g * per, p1 = p0 + per < P ? p0 + per : P;
    float acc[N];
#pragma unroll
    for (int k = 0; k < N; ++k) acc[k] = 0.f;
    for (int p = p0; p < p1; ++p) {
      float v[N];
      V16<T>::load(fb + (long)p * F + c, v);
      const float al = e[p];
#pragma unroll
      for (int k = 0; k < N; ++k) acc[k] += al * v[k];
    }
#pragma unroll
    for (int k = 0; k < N; ++k) zred[(long)g * F + c + k] = acc[k];
  }
  __syncthreads();
  if (g == 0) {
    float acc[N];
#pragma unroll
    for (int k = 0; k < N; ++k) acc[k] = zred[c + k];
    for (int q = 1; q < G; ++q)
#pragma unroll
      for (int k = 0; k < N; ++k) acc[k] += zred[(long)q * F + c + k];
    V16<T>::store(z + (long)b * F + c, acc);
  }
}

// backward of one attention step for sample b (see attn.cpp for the algebra).  One workgroup per sample, but 16 waves of it
// (1024 threads): at B = 64 only 64 workgroups exist, so the parallelism has to come from inside -- the 49 dz . feat_p dot
// products take 4 rounds of the 16 waves instead of 13 of 4, and the channel pass is split over (channel, pixel group).
template <typename T>
__global__ __launch_bounds__(1024) void attn_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ dalpha_extra, long extra_stride,
                                                        const float* __restrict__ alpha, long alpha_stride,
                                                        const T* __restrict__ att1, const float* __restrict__ att2,
                                                        const float* __restrict__ wf, const T* __restrict__ feat,
                                                        float* __restrict__ datt2, float* __restrict__ datt1_acc,
                                                        float* __restrict__ dwf, float* __restrict__ dbf, int P, int A, int F) {
  __shared__ float da[kMaxP], de[kMaxP];
  extern __shared__ float red[];                       // [2][G][A] partial sums of the channel pass (G > 1 only)
  constexpr int N = V16<T>::N;
  const int b = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const T* fb = feat + (long)b * P * F;
  const float* dzb = dz + (long)b * F;
  // d alpha_p = dz . feat_p + regulariser term
  for (int p = wid; p < P; p += nw) {
    float s = 0.f;
    for (int c = lane * N; c < F; c += 64 * N) {
      float v[N];
      V16<T>::load(fb + (long)p * F + c, v);
#pragma unroll
      for (int k = 0; k < N; ++k) s += v[k] * dzb[c + k];
    }
    s = wave_sum(s);
    if (lane == 0) da[p] = s + (dalpha_extra ? dalpha_extra[(long)b * extra_stride + p] : 0.f);
  }
  __syncthreads();
  if (wid == 0) {   // softmax backward: de_p = alpha_p (dalpha_p - sum_q alpha_q dalpha_q)
    const float al = lane < P ? alpha[(long)b * alpha_stride + lane] : 0.f;
    const float d = lane < P ? da[lane] : 0.f;
    const float dot = wave_sum(al * d);
    const float v = al * (d - dot);
    if (lane < P) de[lane] = v;
    const float sb = wave_sum(lane < P ? v : 0.f);
    if (lane == 0) atomicAdd(dbf, sb);
  }
  __syncthreads();
  const T* a1 = att1 + (long)b * P * A;
  const float* a2 = att2 + (long)b * A;
  float* d1 = datt1_acc + (long)b * P * A;
  const int G = (int)blockDim.x >= 2 * A ? (int)blockDim.x / A : 1;   // pixel groups per channel
  if (G == 1) {
    for (int c = threadIdx.x; c < A; c += blockDim.x) {
      const float w = wf[c], t2 = a2[c];
      float s2 = 0.f, sw = 0.f;
      for (int p = 0; p < P; ++p) {
        const float u = to_f32<T>(a1[(long)p * A + c]) + t2;
        const float du = de[p] * w * (u > 0.f ? 1.f : 0.2f);
        s2 += du;
        sw += de[p] * lrelu(u);
        d1[(long)p * A + c] += du;          // only this block touches (b, :, :): plain read-modify-write
      }
      datt2[(long)b * A + c] = s2;
      atomicAdd(dwf + c, sw);
    }
    return;
  }
  const int c = threadIdx.x % A, g = threadIdx.x / A;
  if (g < G) {
    const int per = (P + G - 1) / G, p0 = g * per, p1 = p0 + per < P ? p0 + per : P;
    const float w = wf[c], t2 = a2[c];
    float s2 = 0.f, sw = 0.f;
    for (int p = p0; p < p1; ++p) {
      const float u = to_f32<T>(a1[(long)p * A + c]) + t2;
      const float du = de[p] * w * (u > 0.f ? 1.f : 0.2f);
      s2 += du;
      sw += de[p] * lrelu(u);
      d1[(long)p * A + c] += du;
    }
    red[g * A + c] = s2; red[(G + g) * A + c] = sw;
  }
  __syncthreads();
  if (g == 0) {
    float s2 = 0.f, sw = 0.f;
    for (int q = 0; q < G; ++q) { s2 += red[q * A + c]; sw += red[(G + q) * A + c]; }
    datt2[(long)b * A + c] = s2;
    atomicAdd(dwf + c, sw);
  }
}

// regulariser: S[b][p] = sum_t alpha[b][t][p];  loss += alpha_c * mean_{b,p} (1-S)^2;
// dalpha[b][p] (same for every valid t) = -2 alpha_c (1 - S) / (B P) * gscale
__global__ __launch_bounds__(256) void attn_reg_kernel(const float* __restrict__ alphas, int B, int T, int P, float alpha_c,
                                                       float* __restrict__ loss, float* __restrict__ dalpha, const float* __restrict__ gscale_dev) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float term = 0.f;
  if (i < B * P) {
    const int b = i / P, p = i - b * P;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += alphas[((long)b * T + t) * P + p];
    const float r = 1.f - s;
    term = alpha_c * r * r / (float)(B * P);
    if (dalpha) dalpha[i] = -2.f * alpha_c * r / (float)(B * P) * (gscale_dev ? *gscale_dev : 1.f);
  }
  term = wave_sum(term);
  if ((threadIdx.x & 63) == 0 && loss && term != 0.f) atomicAdd(loss, term);
}

__global__ __launch_bounds__(256) void add_rows_kernel(float* __restrict__ dst, const float* __restrict__ src, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] += src[i];
}

template <typename T>
__global__ __launch_bounds__(256) void replicate_rows_kernel(const T* __restrict__ src, T* __restrict__ dst, long n, int copies) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const T v = src[i];
    for (int c = 0; c < copies; ++c) dst[(long)c * n + i] = v;
  }
}

// split dx0 [rows][2E] fp32: first half -> embedding scatter-add (token ids), second half -> dez rows (dtype)
template <typename T>
__global__ __launch_bounds__(256) void split_dx0_kernel(const float* __restrict__ dx0, const long* __restrict__ ids, float* __restrict__ demb,
                                                        T* __restrict__ dez, int rows, int E, int V) {
  const long total = (long)rows * 2 * E;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / (2 * E)), c = (int)(i - (long)r * 2 * E);
    const float v = dx0[i];
    if (c < E) {
      long t = ids[r]; t = t < 0 ? 0 : (t >= V ? V - 1 : t);
      atomicAdd(demb + t * E + c, v);
    } else {
      dez[(long)r * E + (c - E)] = from_f32<T>(v);
    }
  }
}

inline int grid1d(long work) { long b = (work + 255) / 256; return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); }

}  // namespace

int ncp_to_pf_launch(const float* x, void* y, int B, int F, int P, int dtype, hipStream_t st) {
  ST_CHECK(P <= kMaxP, "attention: at most %d pixels (got P=%d)", kMaxP, P);
  const size_t lds = (size_t)64 * (P + 1) * sizeof(float);
  const int nslab = (F + 63) / 64;
  if (dtype == ST_BF16) hipLaunchKernelGGL(ncp_to_pf_kernel<bf16_t>, dim3(B * nslab), dim3(256), lds, st, x, (bf16_t*)y, B, F, P);
  else hipLaunchKernelGGL(ncp_to_pf_kernel<float>, dim3(B * nslab), dim3(256), lds, st, x, (float*)y, B, F, P);
  ST_LAUNCH_CHECK();
  return 0;
}

int attn_fwd_launch(const void* att1, const float* att2, const float* wf, const float* bf, const void* feat, float* alpha_out,
                    long alpha_stride, void* z, int n, int P, int A, int F, int dtype, hipStream_t st) {
  if (n <= 0) return 0;
  const int nn = dtype == ST_BF16 ? 8 : 4;
  ST_CHECK(A % nn == 0 && F % nn == 0 && P <= kMaxP, "attention: A=%d and F=%d must be multiples of %d, P=%d <= %d", A, F, nn, P, kMaxP);
  const int threads = n < 512 ? 1024 : 256;           // few samples: the parallelism comes from inside the workgroup (attn_bwd_kernel)
  const int nchunk = F / nn;
  const int G = threads >= 2 * nchunk ? threads / nchunk : 1;
  const size_t lds = G > 1 ? (size_t)G * F * sizeof(float) : 0;
  if (dtype == ST_BF16) hipLaunchKernelGGL(attn_fwd_kernel<bf16_t>, dim3(n), dim3(threads), lds, st, (const bf16_t*)att1, att2, wf, bf, (const bf16_t*)feat, alpha_out, alpha_stride, (bf16_t*)z, P, A, F);
  else hipLaunchKernelGGL(attn_fwd_kernel<float>, dim3(n), dim3(threads), lds, st, (const float*)att1, att2, wf, bf, (const float*)feat, alpha_out, alpha_stride, (float*)z, P, A, F);
  ST_LAUNCH_CHECK();
  return 0;
}

int attn_bwd_launch(const float* dz, const float* dalpha_extra, long extra_stride, const float* alpha, long alpha_stride,
                    const void* att1, const float* att2, const float* wf, const void* feat, float* datt2, float* datt1_acc,
                    float* dwf, float* dbf, int n, int P, int A, int F, int dtype, hipStream_t st) {
  if (n <= 0) return 0;
  // 1024 threads per sample while the launch cannot fill the chip with 256-thread workgroups (BASELINE config 3: 64 samples)
  const int threads = n < 512 ? 1024 : 256;
  const int G = threads >= 2 * A ? threads / A : 1;
  const size_t lds = G > 1 ? (size_t)2 * G * A * sizeof(float) : 0;
  if (dtype == ST_BF16) hipLaunchKernelGGL(attn_bwd_kernel<bf16_t>, dim3(n), dim3(threads), lds, st, dz, dalpha_extra, extra_stride, alpha, alpha_stride, (const bf16_t*)att1, att2, wf, (const bf16_t*)feat, datt2, datt1_acc, dwf, dbf, P, A, F);
  else hipLaunchKernelGGL(attn_bwd_kernel<float>, dim3(n), dim3(threads), lds, st, dz, dalpha_extra, extra_stride, alpha, alpha_stride, (const float*)att1, att2, wf, (const float*)feat, datt2, datt1_acc, dwf, dbf, P, A, F);
  ST_LAUNCH_CHECK();
  return 0;
}

int attn_reg_launch(const float* alphas, int B, int T, int P, float alpha_c, float* loss, float* dalpha, const float* gscale_dev, hipStream_t st) {
  hipLaunchKernelGGL(attn_reg_kernel, dim3((B * P + 255) / 256), dim3(256), 0, st, alphas, B, T, P, alpha_c, loss, dalpha, gscale_dev);
  ST_LAUNCH_CHECK();
  return 0;
}

int add_rows_launch(float* dst, const float* src, long n, hipStream_t st) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(add_rows_kernel, dim3(grid1d(n)), dim3(256), 0, st, dst, src, n);
  ST_LAUNCH_CHECK();
  return 0;
}

int replicate_rows_launch(const void* src, void* dst, long n, int copies, int dtype, hipStream_t st) {
  if (n <= 0) return 0;
  if (dtype == ST_BF16) hipLaunchKernelGGL(replicate_rows_kernel<bf16_t>, dim3(grid1d(n)), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, n, copies);
  else hipLaunchKernelGGL(replicate_rows_kernel<float>, dim3(grid1d(n)), dim3(256), 0, st, (const float*)src, (float*)dst, n, copies);
  ST_LAUNCH_CHECK();
  return 0;
}

int split_dx0_launch(const float* dx0, const long* ids, float* demb, void* dez, int rows, int E, int V, int dtype, hipStream_t st) {
  if (rows <= 0) return 0;
  if (dtype == ST_BF16) hipLaunchKernelGGL(split_dx0_kernel<bf16_t>, dim3(grid1d((long)rows * 2 * E)), dim3(256), 0, st, dx0, ids, demb, (bf16_t*)dez, rows, E, V);
  else hipLaunchKernelGGL(split_dx0_kernel<float>, dim3(grid1d((long)rows * 2 * E)), dim3(256), 0, st, dx0, ids, demb, (float*)dez, rows, E, V);
  ST_LAUNCH_CHECK();
  return 0;
}
