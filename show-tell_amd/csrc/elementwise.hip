// HBM-bound elementwise / layout kernels of the encoder path (gfx950).
// Every kernel moves 16 bytes per lane per access and grid-strides over a capped grid.
#include "common.h"
#include <string.h>

namespace {

constexpr int kMaxBlocks = 2048;
inline int grid_for(long work, int threads) {
  long b = (work + threads - 1) / threads;
  return (int)(b < 1 ? 1 : (b > kMaxBlocks ? kMaxBlocks : b));
}

template <typename T> struct Vec;   // 16-byte vector of T
template <> struct Vec<bf16_t> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const void* p, float* f) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(v[i] << 16); f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u); }
  }
  static __device__ __forceinline__ void store(void* p, const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
    *reinterpret_cast<u32x4*>(p) = v;
  }
};
template <> struct Vec<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void load(const void* p, float* f) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = v[i];
  }
  static __device__ __forceinline__ void store(void* p, const float* f) {
    *reinterpret_cast<f32x4*>(p) = f32x4{f[0], f[1], f[2], f[3]};
  }
};

// ---------------------------------------------------------------------------------------
// batch-norm apply (+ residual [+ its own BN]) (+ ReLU)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void bn_coeff(const float* stats, int reps, const float* gamma, const float* beta,
                                         const float* rmean, const float* rvar, int C, int c,
                                         float inv_count, float eps, float& sc, float& sh) {
  if (stats) {
    float s = stats[c], ss = stats[C + c];
    for (int r = 1; r < reps; ++r) { s += stats[(size_t)r * 2 * C + c]; ss += stats[(size_t)r * 2 * C + C + c]; }
    bn_scale_shift(s, ss, inv_count, gamma[c], beta[c], eps, sc, sh);
    return;
  }
  const float mean = rmean[c], var = rvar[c];
  sc = gamma[c] * rsqrtf(var + eps);
  sh = beta[c] - mean * sc;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_act_kernel(st_bn_act_desc d) {
  extern __shared__ __attribute__((aligned(16))) float coef[];  // [sc | sh | rsc | rsh] x C
  const int C = d.C;
  const float inv = 1.0f / d.count;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float sc, sh;
    bn_coeff(d.stats, d.stats_replicas, d.gamma, d.beta, d.running_mean, d.running_var, C, c, inv, d.eps, sc, sh);
    coef[c] = sc; coef[C + c] = sh;
    if (d.res_bn) {
      bn_coeff(d.res_stats, d.res_stats_replicas, d.res_gamma, d.res_beta, d.res_running_mean, d.res_running_var, C, c, inv, d.eps, sc, sh);
      coef[2 * C + c] = sc; coef[3 * C + c] = sh;
    }
  }
  __syncthreads();
  constexpr int N = Vec<T>::N;
  const long nchunk = d.rows * C / N;
  const T* x = reinterpret_cast<const T*>(d.x);
  const T* r = reinterpret_cast<const T*>(d.res);
  T* y = reinterpret_cast<T*>(d.y);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nchunk; i += (long)gridDim.x * blockDim.x) {
    const long e = i * N;
    const int c = (int)(e % C);
    float v[N];
    Vec<T>::load(x + e, v);
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = v[k] * coef[c + k] + coef[C + c + k];
    if (r) {
      float rv[N];
      Vec<T>::load(r + e, rv);
      if (d.res_bn) {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] += rv[k] * coef[2 * C + c + k] + coef[3 * C + c + k];
      } else {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] += rv[k];
      }
    }
    if (d.relu) {
#pragma unroll
      for (int k = 0; k < N; ++k) v[k] = fmaxf(v[k], 0.f);
    }
    Vec<T>::store(y + e, v);
  }
}

// Register-coefficient form for the common case blockDim*N % C == 0 (C a power of two <= 2048): a thread keeps
// the same N channels for the whole grid-stride loop, so it loads only its own coefficients (four 32-byte reads
// issued together with the first data load) -- no LDS table, no barrier, no per-block pass over all C channels.
// The small 14x14 / 7x7 tensors are latency-bound, and that preamble was most of their time.
template <int N> __device__ __forceinline__ void ldf(const float* p, float* f) {
#pragma unroll
  for (int i = 0; i < N; i += 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p + i);
    f[i] = v[0]; f[i + 1] = v[1]; f[i + 2] = v[2]; f[i + 3] = v[3];
  }
}
template <int N> __device__ __forceinline__ void bn_coeff_vec(const float* stats, int reps, const float* gamma, const float* beta,
                                                               const float* rmean, const float* rvar, int C, int c,
                                                               float inv_count, float eps, float* sc, float* sh) {
  float a[N], b[N], g[N], be[N];
  ldf<N>(stats ? stats + c : rmean + c, a);
  ldf<N>(stats ? stats + C + c : rvar + c, b);
  if (stats) {   // replicated statistics ([reps][2C], st_conv_desc.stats_replicas): summed here -- a thread reads 2 x 32 bytes per
                 // replica for its own channels, cheaper than a reduction launch in front of every normalise pass
    int r = 1;
    for (; r + 3 <= reps; r += 3) {      // three replicas per round trip (the loads of a round are independent)
      float a2[3][N], b2[3][N];
#pragma unroll
      for (int q = 0; q < 3; ++q) { ldf<N>(stats + (size_t)(r + q) * 2 * C + c, a2[q]); ldf<N>(stats + (size_t)(r + q) * 2 * C + C + c, b2[q]); }
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int k = 0; k < N; ++k) { a[k] += a2[q][k]; b[k] += b2[q][k]; }
    }
    for (; r < reps; ++r) {
      float a2[N], b2[N];
      ldf<N>(stats + (size_t)r * 2 * C + c, a2);
      ldf<N>(stats + (size_t)r * 2 * C + C + c, b2);
#pragma unroll
      for (int k = 0; k < N; ++k) { a[k] += a2[k]; b[k] += b2[k]; }
    }
  }
  ldf<N>(gamma + c, g);
  ldf<N>(beta + c, be);
#pragma unroll
  for (int k = 0; k < N; ++k) {
    if (stats) { bn_scale_shift(a[k], b[k], inv_count, g[k], be[k], eps, sc[k], sh[k]); continue; }
    sc[k] = g[k] * rsqrtf(b[k] + eps);
    sh[k] = be[k] - a[k] * sc[k];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_act_reg_kernel(st_bn_act_desc d) {
  constexpr int N = Vec<T>::N;
  const int C = d.C;
  const int c = (threadIdx.x * N) % C;
  const float inv = 1.0f / d.count;
  const long nchunk = d.rows * C / N;
  const T* x = reinterpret_cast<const T*>(d.x);
  const T* r = reinterpret_cast<const T*>(d.res);
  T* y = reinterpret_cast<T*>(d.y);
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float v[N], rv[N];
  if (i < nchunk) { Vec<T>::load(x + i * N, v); if (r) Vec<T>::load(r + i * N, rv); }
  float sc[N], sh[N], rsc[N], rsh[N];
  bn_coeff_vec<N>(d.stats, d.stats_replicas, d.gamma, d.beta, d.running_mean, d.running_var, C, c, inv, d.eps, sc, sh);
  if (d.res_bn) bn_coeff_vec<N>(d.res_stats, d.res_stats_replicas, d.res_gamma, d.res_beta, d.res_running_mean, d.res_running_var, C, c, inv, d.eps, rsc, rsh);
  for (; i < nchunk; i += stride) {
    float nv[N], nrv[N];
    const long nx = i + stride;
    if (nx < nchunk) { Vec<T>::load(x + nx * N, nv); if (r) Vec<T>::load(r + nx * N, nrv); }
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = v[k] * sc[k] + sh[k];
    if (r) {
      if (d.res_bn) {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] += rv[k] * rsc[k] + rsh[k];
      } else {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] += rv[k];
      }
    }
    if (d.relu) {
#pragma unroll
      for (int k = 0; k < N; ++k) v[k] = fmaxf(v[k], 0.f);
    }
    if constexpr (sizeof(T) == 2) {
      u32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = pack_bf16x2(v[2 * k], v[2 * k + 1]);
      if (d.rows * C * 2 < (1L << 31)) st_out_store16(y, i * 16, o);
      else *reinterpret_cast<u32x4*>(y + i * N) = o;
    } else Vec<T>::store(y + i * N, v);
#pragma unroll
    for (int k = 0; k < N; ++k) { v[k] = nv[k]; rv[k] = nrv[k]; }
  }
}

__global__ void bn_update_running_kernel(const float* stats, float* rm, float* rv, int C, float count, float mom) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float mean = stats[c] / count;
  const float var = fmaxf(stats[C + c] / count - mean * mean, 0.f);
  const float unbiased = count > 1.f ? var * count / (count - 1.f) : var;
  rm[c] = (1.f - mom) * rm[c] + mom * mean;
  rv[c] = (1.f - mom) * rv[c] + mom * unbiased;
}

// ---------------------------------------------------------------------------------------
// layout
// ---------------------------------------------------------------------------------------
// fp32 NCHW (3 channels, even H and W) -> 2x2 space-to-depth NHWC [B][H/2+3][W/2+3][16]: blocked pixel (p,q) holds
// input rows 2(p-2)+dy, cols 2(q-2)+dx as channel (dy*2+dx)*3+c; channels 12..15 and the border (2 before, 1 after)
// are zero.  One lane per blocked pixel: 6 coalesced 8-byte reads, 32 (bf16) / 64 (f32) contiguous bytes written.
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_s2d_kernel(const float* __restrict__ x, T* __restrict__ y, int B, int H, int W) {
  const int Hp = H / 2 + 3, Wp = W / 2 + 3;
  const long npix = (long)B * Hp * Wp;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
    const long b = p / ((long)Hp * Wp);
    const int rem = (int)(p - b * Hp * Wp), py = rem / Wp, px = rem - py * Wp;
    const int r = py - 2, q = px - 2;
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = 0.f;
    if (r >= 0 && r < H / 2 && q >= 0 && q < W / 2) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          const float2 t = *reinterpret_cast<const float2*>(x + ((b * 3 + c) * H + 2 * r + dy) * W + 2 * q);
          v[(dy * 2 + 0) * 3 + c] = t.x;
          v[(dy * 2 + 1) * 3 + c] = t.y;
        }
    }
#pragma unroll
    for (int o = 0; o < 16; o += Vec<T>::N) Vec<T>::store(y + p * 16 + o, v + o);
  }
}

// packed stem weights [64][7][7][Cpad] -> [64][4][4][16] for the blocked image: tap (i,j), channel (dy*2+dx)*3+c takes
// the 7x7 weight at (2i+dy-1, 2j+dx-1) when that lies inside the filter, else zero.
template <typename T>
__global__ void stem_weight_s2d_kernel(const T* __restrict__ w, T* __restrict__ out, int Cpad) {
  const int n = blockIdx.x, t = threadIdx.x;        // 256 threads = 4*4*16 outputs of one filter
  const int i = t >> 6, j = (t >> 4) & 3, ch = t & 15;
  T v = from_f32<T>(0.f);
  if (ch < 12) {
    const int blk = ch / 3, c = ch - blk * 3, dy = blk >> 1, dx = blk & 1;
    const int ky = 2 * i + dy - 1, kx = 2 * j + dx - 1;
    if (ky >= 0 && ky < 7 && kx >= 0 && kx < 7) v = w[((n * 7 + ky) * 7 + kx) * Cpad + c];
  }
  out[n * 256 + t] = v;
}

template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y,
                                                            int B, int C, int HW, int Cpad) {
  const long npix = (long)B * HW;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
    const long b = p / HW, hw = p - b * HW;
    for (int c0 = 0; c0 < Cpad; c0 += Vec<T>::N) {
      float v[Vec<T>::N];
#pragma unroll
      for (int k = 0; k < Vec<T>::N; ++k) v[k] = (c0 + k < C) ? x[(b * C + c0 + k) * HW + hw] : 0.f;
      Vec<T>::store(y + p * Cpad + c0, v);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_ncp_kernel(const T* __restrict__ x, float* __restrict__ y,
                                                           int B, int HW, int C) {
  // one block per (b, 64-channel slab): stage [HW][64] through LDS, write [64][HW] rows
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [HW][65]
  const int nslab = C / 64;
  const int b = blockIdx.x / nslab, c0 = (blockIdx.x - b * nslab) * 64;
  for (int i = threadIdx.x; i < HW * 64; i += blockDim.x) {
    const int p = i >> 6, c = i & 63;
    tile[p * 65 + c] = to_f32<T>(x[((long)b * HW + p) * C + c0 + c]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < HW * 64; i += blockDim.x) {
    const int c = i / HW, p = i - c * HW;
    y[((long)b * C + c0 + c) * HW + p] = tile[p * 65 + c];
  }
}

struct BnArgs { int on; const float* stats; const float* gamma; const float* beta; const float* rmean; const float* rvar; float inv_count, eps; };

template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                       int B, int H, int W, int C, int Ho, int Wo, BnArgs bn) {
  // bn.on: y = maxpool(relu(batchnorm(x))) -- the stem's normalise pass folded into the pool (relu and the rounding to
  // T are monotone, so max-then-relu-then-round equals the two-pass result bit for bit).  Needs blockDim*N % C == 0:
  // a thread keeps its N channels, coefficients live in registers.
  constexpr int N = Vec<T>::N;
  const int cv = C / N;
  const long total = (long)B * Ho * Wo * cv;
  float sc[N], sh[N];
  if (bn.on) bn_coeff_vec<N>(bn.stats, 1, bn.gamma, bn.beta, bn.rmean, bn.rvar, C, (int)((threadIdx.x * N) % C), bn.inv_count, bn.eps, sc, sh);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * N;
    long p = i / cv;
    const int wo = (int)(p % Wo); p /= Wo;
    const int ho = (int)(p % Ho);
    const long b = p / Ho;
    float m[N];
#pragma unroll
    for (int k = 0; k < N; ++k) m[k] = -INFINITY;
    for (int dh = 0; dh < 3; ++dh) {
      const int hi = ho * 2 - 1 + dh;
      if ((unsigned)hi >= (unsigned)H) continue;
      for (int dw = 0; dw < 3; ++dw) {
        const int wi = wo * 2 - 1 + dw;
        if ((unsigned)wi >= (unsigned)W) continue;
        float v[N];
        Vec<T>::load(x + ((b * H + hi) * W + wi) * C + c, v);
        if (bn.on) {
#pragma unroll
          for (int k = 0; k < N; ++k) v[k] = v[k] * sc[k] + sh[k];
        }
#pragma unroll
        for (int k = 0; k < N; ++k) m[k] = fmaxf(m[k], v[k]);
      }
    }
    if (bn.on) {
#pragma unroll
      for (int k = 0; k < N; ++k) m[k] = fmaxf(m[k], 0.f);
    }
    Vec<T>::store(y + ((b * Ho + ho) * Wo + wo) * C + c, m);
  }
}

template <typename T, typename TO>
__global__ __launch_bounds__(256) void avgpool_kernel(const T* __restrict__ x, TO* __restrict__ y, int B, int HW, int C) {
  constexpr int N = Vec<T>::N;
  const int cv = C / N;
  const long total = (long)B * cv;
  const float inv = 1.0f / HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / cv;
    const int c = (int)(i - b * cv) * N;
    float s[N];
#pragma unroll
    for (int k = 0; k < N; ++k) s[k] = 0.f;
    // seven loads in flight per thread, added in pixel order (a 7 x 7 map is seven batches)
    constexpr int U = 7;
    int p = 0;
    for (; p + U <= HW; p += U) {
      float v[U][N];
#pragma unroll
      for (int u = 0; u < U; ++u) Vec<T>::load(x + (b * HW + p + u) * C + c, v[u]);
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int k = 0; k < N; ++k) s[k] += v[u][k];
    }
    for (; p < HW; ++p) {
      float v[N];
      Vec<T>::load(x + (b * HW + p) * C + c, v);
#pragma unroll
      for (int k = 0; k < N; ++k) s[k] += v[k];
    }
#pragma unroll
    for (int k = 0; k < N; ++k) y[b * C + c + k] = from_f32<TO>(s[k] * inv);
  }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_kernel(const TI* __restrict__ x, TO* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = from_f32<TO>(to_f32<TI>(x[i]));
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast2d_kernel(const TI* __restrict__ x, TO* __restrict__ y, int rows, int cols, int ldx, int ldy) {
  const long total = (long)rows * ldy;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / ldy; const int c = (int)(i - r * ldy);
    y[i] = from_f32<TO>(c < cols ? to_f32<TI>(x[r * ldx + c]) : 0.f);
  }
}

// y[c][r] = x[r][c] over 64x64 tiles, 16-byte global accesses on both sides (rows >= `rows` read as zero, so the pad
// columns of y up to ldy are zero-filled); optionally colsum[c] += sum_r x[r][c] from the same tile -- the bias
// gradient that always accompanies the K-major copy of a gradient matrix in the decoder backward.
// blockIdx.z selects one of up to 12 matrices of identical shape (one launch for the same operand of every decoder
// layer); with rows_t / prev_row the source row of output row r is prev_row[r] (zero when rows_t[r] == 0): the
// "previous hidden state" matrix of BPTT is transposed straight out of the layer output, never materialised.
struct TransposeBatch { const void* x[12]; void* y[12]; float* colsum[12]; };

template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(TransposeBatch tb, int rows, int cols, int ldx, int ldy,
                                                         const int* __restrict__ rows_t, const int* __restrict__ prev_row) {
  constexpr int N = Vec<T>::N, CPR = 64 / N;       // elements per 16 bytes, chunks per 64-element tile row
  constexpr int LD = 64 + (sizeof(T) == 2 ? 2 : 1);
  __shared__ T tile[64][LD];
  const T* __restrict__ x = reinterpret_cast<const T*>(tb.x[blockIdx.z]);
  T* __restrict__ y = reinterpret_cast<T*>(tb.y[blockIdx.z]);
  float* __restrict__ colsum = tb.colsum[blockIdx.z];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const bool vec_in = (ldx % N == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  for (int i = threadIdx.x; i < 64 * CPR; i += 256) {
    const int rl = i / CPR, cl = (i % CPR) * N;
    const int r = r0 + rl, c = c0 + cl;
    long sr = r;                                     // source row
    bool rok = r < rows;
    if (rows_t && rok) { rok = rows_t[r] > 0; sr = rok ? prev_row[r] : 0; }
    float v[N];
    if (rok && c + N <= cols && vec_in) {
      Vec<T>::load(x + sr * ldx + c, v);
    } else {
#pragma unroll
      for (int k = 0; k < N; ++k) v[k] = (rok && c + k < cols) ? to_f32<T>(x[sr * ldx + c + k]) : 0.f;
    }
#pragma unroll
    for (int k = 0; k < N; ++k) tile[rl][cl + k] = from_f32<T>(v[k]);
  }
  __syncthreads();
  if (colsum && threadIdx.x < 64 && c0 + threadIdx.x < cols) {
    float sum = 0.f;
#pragma unroll 8
    for (int r = 0; r < 64; ++r) sum += to_f32<T>(tile[r][threadIdx.x]);
    atomicAdd(colsum + c0 + threadIdx.x, sum);
  }
  const bool vec_out = (ldy % N == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
  for (int i = threadIdx.x; i < 64 * CPR; i += 256) {
    const int cl = i / CPR, rl = (i % CPR) * N;
    const int c = c0 + cl, r = r0 + rl;
    if (c >= cols || r >= ldy) continue;
    float v[N];
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = to_f32<T>(tile[rl + k][cl]);
    if (r + N <= ldy && vec_out) {
      Vec<T>::store(y + (long)c * ldy + r, v);
    } else {
#pragma unroll
      for (int k = 0; k < N; ++k) if (r + k < ldy) y[(long)c * ldy + r + k] = from_f32<T>(v[k]);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_conv_weight_kernel(const float* __restrict__ w, T* __restrict__ out,
                                                                int Cout, int Cin, int KH, int KW, int Cpad, int CH) {
  const long total = (long)Cout * KH * KW * Cpad;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c, kw, kh; long co;
    if (CH == 0) {            // [Cout][KH][KW][Cpad]
      c = (int)(i % Cpad);
      long t = i / Cpad;
      kw = (int)(t % KW); t /= KW;
      kh = (int)(t % KH);
      co = t / KH;
    } else {                  // [Cout][Cpad/CH][KH][KW][CH]
      const int cl = (int)(i % CH);
      long t = i / CH;
      kw = (int)(t % KW); t /= KW;
      kh = (int)(t % KH); t /= KH;
      const int cc = (int)(t % (Cpad / CH));
      co = t / (Cpad / CH);
      c = cc * CH + cl;
    }
    out[i] = from_f32<T>(c < Cin ? w[((co * Cin + c) * KH + kh) * KW + kw] : 0.f);
  }
}

}  // namespace

#define ST_DT_CHECK(dt, name) ST_CHECK((dt) == ST_F32 || (dt) == ST_BF16, name ": bad dtype %d", (dt))

extern "C" int st_bn_act(const st_bn_act_desc* d, void* stream) {
  ST_CHECK(d && d->x && d->y && d->gamma && d->beta, "st_bn_act: null pointer");
  ST_DT_CHECK(d->dtype, "st_bn_act");
  ST_CHECK(d->stats || (d->running_mean && d->running_var), "st_bn_act: need stats or running buffers");
  const int n = d->dtype == ST_BF16 ? 8 : 4;
  ST_CHECK(d->C % n == 0, "st_bn_act: C=%d must be a multiple of %d", d->C, n);
  if (d->res_bn) {
    ST_CHECK(d->res && d->res_gamma && d->res_beta && (d->res_stats || (d->res_running_mean && d->res_running_var)),
             "st_bn_act: residual BN needs its parameters");
  }
  ST_CHECK(d->C <= 8192, "st_bn_act: C too large");
  const long nchunk = d->rows * d->C / n;
  const int grid = grid_for(nchunk, 256);
  const size_t lds = (size_t)(d->res_bn ? 4 : 2) * d->C * sizeof(float);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  ST_CHECK(d->stats_replicas >= 0 && d->res_stats_replicas >= 0, "st_bn_act: bad replica count");
  const bool reg = d->stats_replicas <= 16 && d->res_stats_replicas <= 16 && (256 * n) % d->C == 0;
  if (reg) {
    if (d->dtype == ST_BF16) hipLaunchKernelGGL(bn_act_reg_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, *d);
    else hipLaunchKernelGGL(bn_act_reg_kernel<float>, dim3(grid), dim3(256), 0, st, *d);
  } else if (d->dtype == ST_BF16) hipLaunchKernelGGL(bn_act_kernel<bf16_t>, dim3(grid), dim3(256), lds, st, *d);
  else hipLaunchKernelGGL(bn_act_kernel<float>, dim3(grid), dim3(256), lds, st, *d);
  ST_LAUNCH_CHECK();
  return 0;
}

extern "C" int st_bn_update_running(const float* stats, float* rm, float* rv, int C, float count, float momentum, void* stream) {
  ST_CHECK(stats && rm && rv && C > 0, "st_bn_update_running: bad arguments");
  hipLaunchKernelGGL(bn_update_running_kernel, dim3((C + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     stats, rm, rv, C, count, momentum);
  ST_LAUNCH_CHECK();
  return 0;
}

extern "C" int st_nchw_to_nhwc(const float* x, void* y, int dtype, int B, int C, int H, int W, int Cpad, void* stream) {
  ST_CHECK(x && y, "st_nchw_to_nhwc: null pointer");
  ST_DT_CHECK(dtype, "st_nchw_to_nhwc");
  ST_CHECK(Cpad >= C && Cpad % (dtype == ST_BF16 ? 8 : 4) == 0, "st_nchw_to_nhwc: bad Cpad=%d", Cpad);
  const int grid = grid_for((long)B * H * W, 256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == ST_BF16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, x, (bf16_t*)y, B, C, H * W, Cpad);
  else hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid), dim3(256), 0, st, x, (float*)y, B, C, H * W, Cpad);
  ST_LAUNCH_CHECK();
  return 0;
}

extern "C" int st_nchw_to_s2d16(const float* x, void* y, int dtype, int B, int H, int W, void* stream) {
  ST_CHECK(x && y, "st_nchw_to_s2d16: null pointer");
  ST_DT_CHECK(dtype, "st_nchw_to_s2d16");
  ST_CHECK(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "st_nchw_to_s2d16: H=%d, W=%d must be even", H, W);
  const int grid = grid_for((long)B * (H / 2 + 3) * (W / 2 + 3), 256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == ST_BF16) hipLaunchKernelGGL(nchw_to_s2d_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, x, (bf16_t*)y, B, H, W);
  else hipLaunchKernelGGL(nchw_to_s2d_kernel<float>, dim3(grid), dim3(256), 0, st, x, (float*)y, B, H, W);
  ST_LAUNCH_CHECK();
  return 0;
}

extern "C" int st_stem_weight_s2d(const void* w, void* out, int dtype, int Cpad, void* stream) {
  ST_CHECK(w && out, "st_stem_weight_s2d: null pointer");
  ST_DT_CHECK(dtype, "st_stem_weight_s2d");
  ST_CHECK(Cpad >= 3, "st_stem_weight_s2d: bad Cpad=%d", Cpad);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == ST_BF16) hipLaunchKernelGGL(stem_weight_s2d_kernel<bf16_t>, dim3(64), dim3(256), 0, st, (const bf16_t*)w, (bf16_t*)out, Cpad);
  else hipLaunchKernelGGL(stem_weight_s2d_kernel<float>, dim3(64), dim3(256), 0, st, (const float*)w, (float*)out, Cpad);
  ST_LAUNCH_CHECK();
  return 0;
}

extern "C" int st_nhwc_to_ncp_f32(const void* x, float* y, int dtype, int B, int HW, int C, void* stream) {
  ST_CHECK(x && y, "st_nhwc_to_ncp_f32: null pointer");
  ST_DT_CHECK(dtype, "st_nhwc_to_ncp_f32");
  ST_CHECK(C % 64 == 0 && HW * 65 * 4 <= 64 * 1024, "st_nhwc_to_ncp_f32: needs C%%64==0 and HW<=252 (got C=%d HW=%d)", C, HW);
  const int grid = B * (C / 64);
  const size_t lds = (size_t)HW * 65 * sizeof(float);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == ST_BF16) hipLaunchKernelGGL(nhwc_to_ncp_kernel<bf16_t>, dim3(grid), dim3(256), lds, st, (const bf16_t*)x, y, B, HW, C);
  else hipLaunchKernelGGL(nhwc_to_ncp_kernel<float>, dim3(grid), dim3(256), lds, st, (const float*)x, y, B, HW, C);
  ST_LAUNCH_CHECK();
  return 0;
}

extern "C" int st_maxpool3x3s2(const void* x, void* y, int dtype, int B, int H, int W, int C, void* stream) {
  ST_CHECK(x && y, "st_maxpool3x3s2: null pointer");
  ST_DT_CHECK(dtype, "st_maxpool3x3s2");
  const int n = dtype == ST_BF16 ? 8 : 4;
  ST_CHECK(C % n == 0, "st_maxpool3x3s2: C=%d must be a multiple of %d", C, n);
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int grid = grid_for((long)B * Ho * Wo * (C / n), 256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  BnArgs bn; memset(&bn, 0, sizeof(bn));
  if (dtype == ST_BF16) hipLaunchKernelGGL(maxpool_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, B, H, W, C, Ho, Wo, bn);
  else hipLaunchKernelGGL(maxpool_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, (float*)y, B, H, W, C, Ho, Wo, bn);
  ST_LAUNCH_CHECK();
  return 0;
}

extern "C" int st_maxpool3x3s2_bn(const void* x, void* y, int dtype, int B, int H, int W, int C,
                                  const float* stats, const float* gamma, const float* beta,
                                  const float* running_mean, const float* running_var, float count, float eps, void* stream) {
  ST_CHECK(x && y && gamma && beta && (stats || (running_mean && running_var)), "st_maxpool3x3s2_bn: null pointer");
  ST_DT_CHECK(dtype, "st_maxpool3x3s2_bn");
  const int n = dtype == ST_BF16 ? 8 : 4;
  ST_CHECK(C % n == 0 && (256 * n) % C == 0, "st_maxpool3x3s2_bn: C=%d must divide %d", C, 256 * n);
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int grid = grid_for((long)B * Ho * Wo * (C / n), 256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  BnArgs bn{1, stats, gamma, beta, running_mean, running_var, 1.0f / count, eps};
  if (dtype == ST_BF16) hipLaunchKernelGGL(maxpool_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, B, H, W, C, Ho, Wo, bn);
  else hipLaunchKernelGGL(maxpool_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, (float*)y, B, H, W, C, Ho, Wo, bn);
  ST_LAUNCH_CHECK();
  return 0;
}

extern "C" int st_global_avgpool(const void* x, void* y, int dtype, int out_dtype, int B, int HW, int C, void* stream) {
  ST_CHECK(x && y, "st_global_avgpool: null pointer");
  ST_DT_CHECK(dtype, "st_global_avgpool");
  ST_DT_CHECK(out_dtype, "st_global_avgpool");
  const int n = dtype == ST_BF16 ? 8 : 4;
  ST_CHECK(C % n == 0, "st_global_avgpool: C=%d must be a multiple of %d", C, n);
  const int grid = grid_for((long)B * (C / n), 64);   // one wave per block: 32 K threads at (128, 2048) reach every CU
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == ST_BF16 && out_dtype == ST_BF16) hipLaunchKernelGGL((avgpool_kernel<bf16_t, bf16_t>), dim3(grid), dim3(64), 0, st, (const bf16_t*)x, (bf16_t*)y, B, HW, C);
  else if (dtype == ST_BF16) hipLaunchKernelGGL((avgpool_kernel<bf16_t, float>), dim3(grid), dim3(64), 0, st, (const bf16_t*)x, (float*)y, B, HW, C);
  else if (out_dtype == ST_BF16) hipLaunchKernelGGL((avgpool_kernel<float, bf16_t>), dim3(grid), dim3(64), 0, st, (const float*)x, (bf16_t*)y, B, HW, C);
  else hipLaunchKernelGGL((avgpool_kernel<float, float>), dim3(grid), dim3(64), 0, st, (const float*)x, (float*)y, B, HW, C);
  ST_LAUNCH_CHECK();
  return 0;
}

extern "C" int st_cast(const void* x, void* y, int from_dtype, int to_dtype, long n, void* stream) {
  ST_CHECK(x && y, "st_cast: null pointer");
  ST_DT_CHECK(from_dtype, "st_cast");
  ST_DT_CHECK(to_dtype, "st_cast");
  if (n <= 0) return 0;
  const int grid = grid_for(n, 256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (from_dtype == ST_F32 && to_dtype == ST_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(grid), dim3(256), 0, st, (const float*)x, (bf16_t*)y, n);
  else if (from_dtype == ST_BF16 && to_dtype == ST_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (float*)y, n);
  else if (from_dtype == ST_F32) hipLaunchKernelGGL((cast_kernel<float, float>), dim3(grid), dim3(256), 0, st, (const float*)x, (float*)y, n);
  else hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, n);
  ST_LAUNCH_CHECK();
  return 0;
}

extern "C" int st_cast2d(const void* x, void* y, int from_dtype, int to_dtype, int rows, int cols, int ldx, int ldy, void* stream) {
  ST_CHECK(x && y, "st_cast2d: null pointer");
  ST_DT_CHECK(from_dtype, "st_cast2d");
  ST_DT_CHECK(to_dtype, "st_cast2d");
  ST_CHECK(ldx >= cols && ldy >= cols, "st_cast2d: leading dimensions too small");
  if (rows <= 0) return 0;
  const int grid = grid_for((long)rows * ldy, 256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (from_dtype == ST_F32 && to_dtype == ST_BF16) hipLaunchKernelGGL((cast2d_kernel<float, bf16_t>), dim3(grid), dim3(256), 0, st, (const float*)x, (bf16_t*)y, rows, cols, ldx, ldy);
  else if (from_dtype == ST_BF16 && to_dtype == ST_F32) hipLaunchKernelGGL((cast2d_kernel<bf16_t, float>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (float*)y, rows, cols, ldx, ldy);
  else if (from_dtype == ST_F32) hipLaunchKernelGGL((cast2d_kernel<float, float>), dim3(grid), dim3(256), 0, st, (const float*)x, (float*)y, rows, cols, ldx, ldy);
  else hipLaunchKernelGGL((cast2d_kernel<bf16_t, bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, rows, cols, ldx, ldy);
  ST_LAUNCH_CHECK();
  return 0;
}

extern "C" int st_transpose_batch(const void* const* x, void* const* y, float* const* colsum, int n, int dtype, int rows, int cols,
                                  int ldx, int ldy, const int* rows_t, const int* prev_row, void* stream) {
  ST_CHECK(x && y && n >= 1 && n <= 12, "st_transpose_batch: 1..12 matrices per launch");
  ST_DT_CHECK(dtype, "st_transpose");
  ST_CHECK(ldx >= cols && ldy >= rows, "st_transpose: leading dimensions too small (rows=%d cols=%d ldx=%d ldy=%d)", rows, cols, ldx, ldy);
  ST_CHECK((rows_t == nullptr) == (prev_row == nullptr), "st_transpose_batch: rows_t and prev_row go together");
  TransposeBatch tb;
  memset(&tb, 0, sizeof(tb));
  for (int i = 0; i < n; ++i) {
    ST_CHECK(x[i] && y[i], "st_transpose: null pointer");
    tb.x[i] = x[i]; tb.y[i] = y[i]; tb.colsum[i] = colsum ? colsum[i] : nullptr;
  }
  const dim3 grid((cols + 63) / 64, (ldy + 63) / 64, n);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == ST_BF16) hipLaunchKernelGGL(transpose_kernel<bf16_t>, grid, dim3(256), 0, st, tb, rows, cols, ldx, ldy, rows_t, prev_row);
  else hipLaunchKernelGGL(transpose_kernel<float>, grid, dim3(256), 0, st, tb, rows, cols, ldx, ldy, rows_t, prev_row);
  ST_LAUNCH_CHECK();
  return 0;
}

extern "C" int st_transpose_colsum(const void* x, void* y, float* colsum, int dtype, int rows, int cols, int ldx, int ldy, void* stream) {
  return st_transpose_batch(&x, &y, &colsum, 1, dtype, rows, cols, ldx, ldy, nullptr, nullptr, stream);
}

extern "C" int st_transpose(const void* x, void* y, int dtype, int rows, int cols, int ldx, int ldy, void* stream) {
  return st_transpose_colsum(x, y, nullptr, dtype, rows, cols, ldx, ldy, stream);
}

extern "C" int st_pack_conv_weight(const float* w, void* out, int dtype, int Cout, int Cin, int KH, int KW, int Cpad, int k_order, void* stream) {
  ST_CHECK(w && out, "st_pack_conv_weight: null pointer");
  ST_DT_CHECK(dtype, "st_pack_conv_weight");
  ST_CHECK(Cpad >= Cin, "st_pack_conv_weight: Cpad < Cin");
  const int CH = k_order ? (dtype == ST_BF16 ? 64 : 32) : 0;
  ST_CHECK(!k_order || Cpad % CH == 0, "st_pack_conv_weight: k_order=1 needs Cpad %% %d == 0", CH);
  const long total = (long)Cout * KH * KW * Cpad;
  const int grid = grid_for(total, 256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == ST_BF16) hipLaunchKernelGGL(pack_conv_weight_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, w, (bf16_t*)out, Cout, Cin, KH, KW, Cpad, CH);
  else hipLaunchKernelGGL(pack_conv_weight_kernel<float>, dim3(grid), dim3(256), 0, st, w, (float*)out, Cout, Cin, KH, KW, Cpad, CH);
  ST_LAUNCH_CHECK();
  return 0;
}
