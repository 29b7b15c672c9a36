// Encoder head (Linear -> BatchNorm1d, reference cnn.py:37-38,49) and the optimizers
// (torch.optim.SGD / Adam at main.py:97-100,152) for gfx950.
//
// The head is tiny (B x 2048 x E) and bandwidth-trivial; its GEMMs go through st_conv, the
// BatchNorm1d statistics / gradients use one thread per feature column (coalesced over
// columns, serial over the <= few hundred batch rows).
// The optimizers are single multi-tensor launches over ONE flat fp32 parameter buffer
// (HBM-bound: SGD 3 reads + 2 writes, Adam 4 reads + 3 writes per element, plus the bf16
// shadow copy the MFMA kernels read), instead of torch's per-tensor loops.
#include "common.h"
#include "rnn_kernels.h"
#include <string.h>

namespace {

// BatchNorm1d over [B][E] fp32: a block owns 32 columns, its 8 row groups stride over the batch and meet in LDS
// (128-byte coalesced row segments, E/32 blocks instead of one serial thread per column).
constexpr int kBnTX = 32, kBnTY = 8;
__device__ __forceinline__ float bn1d_colsum(float v, float (*red)[kBnTX]) {
  const int tx = threadIdx.x, ty = threadIdx.y;
  __syncthreads();                      // previous use of red is over
  red[ty][tx] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < kBnTY; ++k) s += red[k][tx];
  return s;
}

template <typename TO>
__global__ __launch_bounds__(256) void bn1d_fwd_kernel(const float* __restrict__ z, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ rm, float* __restrict__ rv, float* __restrict__ save_mean,
                                                       float* __restrict__ save_rstd, TO* __restrict__ y, float* __restrict__ y_f32,
                                                       int B, int E, int train, float mom, float eps) {
  __shared__ float red[kBnTY][kBnTX];
  const int e = blockIdx.x * kBnTX + threadIdx.x, ty = threadIdx.y;
  const bool ok = e < E;
  float mean = 0.f, rstd = 0.f;
  if (train) {
    float s = 0.f;
    if (ok) for (int b = ty; b < B; b += kBnTY) s += z[(long)b * E + e];
    mean = bn1d_colsum(s, red) / B;
    float v = 0.f;
    if (ok) for (int b = ty; b < B; b += kBnTY) { const float d = z[(long)b * E + e] - mean; v += d * d; }
    v = bn1d_colsum(v, red);
    const float var = v / B;
    rstd = rsqrtf(var + eps);
    if (ok && ty == 0) {
      const float unb = B > 1 ? v / (B - 1) : var;
      rm[e] = (1.f - mom) * rm[e] + mom * mean;
      rv[e] = (1.f - mom) * rv[e] + mom * unb;
    }
  } else if (ok) {
    mean = rm[e];
    rstd = rsqrtf(rv[e] + eps);
  }
  if (!ok) return;
  if (save_mean && ty == 0) { save_mean[e] = mean; save_rstd[e] = rstd; }
  const float g = gamma[e] * rstd, sh = beta[e] - mean * g;
  for (int b = ty; b < B; b += kBnTY) {
    const float o = z[(long)b * E + e] * g + sh;
    if (y) y[(long)b * E + e] = from_f32<TO>(o);
    if (y_f32) y_f32[(long)b * E + e] = o;
  }
}

template <typename TO>
__global__ __launch_bounds__(256) void bn1d_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ z, const float* __restrict__ gamma,
                                                       const float* __restrict__ save_mean, const float* __restrict__ save_rstd,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta, TO* __restrict__ dz,
                                                       int B, int E, int ldz, int train) {
  __shared__ float red[kBnTY][kBnTX];
  const int e = blockIdx.x * kBnTX + threadIdx.x, ty = threadIdx.y;
  const bool ok = e < E;
  const float mean = ok ? save_mean[e] : 0.f, rstd = ok ? save_rstd[e] : 0.f, g = ok ? gamma[e] : 0.f;
  float sdy = 0.f, sdyx = 0.f;
  if (ok) for (int b = ty; b < B; b += kBnTY) {
    const float d = dy[(long)b * E + e], xh = (z[(long)b * E + e] - mean) * rstd;
    sdy += d; sdyx += d * xh;
  }
  sdy = bn1d_colsum(sdy, red);
  sdyx = bn1d_colsum(sdyx, red);
  if (!ok) return;
  if (ty == 0) { dgamma[e] += sdyx; dbeta[e] += sdy; }
  const float m1 = sdy / B, m2 = sdyx / B;
  for (int b = ty; b < B; b += kBnTY) {
    const float d = dy[(long)b * E + e], xh = (z[(long)b * E + e] - mean) * rstd;
    const float o = train ? g * rstd * (d - m1 - xh * m2) : g * rstd * d;
    dz[(long)b * ldz + e] = from_f32<TO>(o);
  }
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                  uint16_t* __restrict__ shadow, long n, float lr, float mom, int first, float gscale) {
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long)gridDim.x * blockDim.x * 4) {
    if (i + 3 < n) {
      f32x4 pv = *reinterpret_cast<f32x4*>(p + i);
      const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i);
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (mom != 0.f && !first) bv = *reinterpret_cast<f32x4*>(buf + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gg = gv[e] * gscale;
        const float b = (mom != 0.f) ? (first ? gg : mom * bv[e] + gg) : gg;
        bv[e] = b; pv[e] -= lr * b;
      }
      *reinterpret_cast<f32x4*>(p + i) = pv;
      if (mom != 0.f) *reinterpret_cast<f32x4*>(buf + i) = bv;
      if (shadow) *reinterpret_cast<u32x2*>(shadow + i) = u32x2{pack_bf16x2(pv[0], pv[1]), pack_bf16x2(pv[2], pv[3])};
    } else {
      for (long j = i; j < n; ++j) {
        const float gg = g[j] * gscale;
        const float b = (mom != 0.f) ? (first ? gg : mom * buf[j] + gg) : gg;
        if (mom != 0.f) buf[j] = b;
        p[j] -= lr * b;
        if (shadow) shadow[j] = f32_to_bf16_bits(p[j]);
      }
    }
  }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   uint16_t* __restrict__ shadow, long n, float lr, float b1, float b2, float eps,
                                                   float bc1, float sqrt_bc2, float gscale) {
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long)gridDim.x * blockDim.x * 4) {
    const int cnt = (i + 3 < n) ? 4 : (int)(n - i);
    for (int e = 0; e < cnt; ++e) {
      const long j = i + e;
      const float gg = g[j] * gscale;
      const float mm = b1 * m[j] + (1.f - b1) * gg;
      const float vv = b2 * v[j] + (1.f - b2) * gg * gg;
      m[j] = mm; v[j] = vv;
      const float denom = sqrtf(vv) / sqrt_bc2 + eps;
      const float np = p[j] - (lr / bc1) * (mm / denom);
      p[j] = np;
      if (shadow) shadow[j] = f32_to_bf16_bits(np);
    }
  }
}

int gemm_nt_(const void* a, int lda, const void* w, int ldw, void* y, int ldy, int M, int N, int K, int dtype, int out_dtype,
             const float* bias, int accumulate, void* stream, int split_k = 0) {
  st_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.split_k = split_k;
  d.x = a; d.w = w; d.y = y; d.bias = bias; d.dtype = dtype; d.out_dtype = out_dtype;
  d.B = M; d.Hin = 1; d.Win = 1; d.Cin = K; d.Ho = 1; d.Wo = 1; d.N = N; d.KH = 1; d.KW = 1; d.stride = 1; d.pad = 0;
  d.ldx = lda; d.ldw = ldw; d.ldy = ldy; d.accumulate = accumulate;
  return st_conv(&d, stream);
}

inline int up8(int v) { return (v + 7) & ~7; }
inline size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" size_t st_head_workspace_bytes(int B, int F, int E, int dtype) {
  const size_t es = st_dtype_size(dtype);
  const int Bp = up8(B);
  return al((size_t)B * up8(E) * es) + al((size_t)E * Bp * es) + al((size_t)F * Bp * es);
}

extern "C" int st_linear_bn1d_forward(const void* x, const void* w, const float* bias, const float* gamma, const float* beta,
                                      float* running_mean, float* running_var, int B, int F, int E, int dtype,
                                      int train, float momentum, float eps,
                                      float* z_out, float* save_mean, float* save_rstd, void* y_dtype, float* y_f32, void* stream) {
  ST_CHECK(x && w && bias && gamma && beta && running_mean && running_var && z_out, "st_linear_bn1d_forward: null pointer");
  ST_CHECK(dtype == ST_F32 || dtype == ST_BF16, "st_linear_bn1d_forward: bad dtype");
  ST_CHECK(B > 0 && F % 8 == 0 && E % 4 == 0, "st_linear_bn1d_forward: need F%%8==0 and E%%4==0 (F=%d E=%d)", F, E);
  ST_CHECK(!train || B > 1, "Expected more than 1 value per channel when training (BatchNorm1d, B=%d)", B);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // B x E output tiles are few (4 at B = 128, E = 512) and K = F is long: 8 K slices in one grouped launch
  const int bk = dtype == ST_BF16 ? 64 : 32;
  const int split = (F % (8 * bk) == 0 && (long)B * E <= 256 * 128 * 64) ? 8 : 0;
  if (gemm_nt_(x, F, w, F, z_out, E, B, E, F, dtype, ST_F32, bias, 0, stream, split)) return 1;
  const dim3 grid((E + kBnTX - 1) / kBnTX), block(kBnTX, kBnTY);
  if (dtype == ST_BF16)
    hipLaunchKernelGGL(bn1d_fwd_kernel<bf16_t>, grid, block, 0, st, z_out, gamma, beta, running_mean, running_var, save_mean, save_rstd,
                       (bf16_t*)y_dtype, y_f32, B, E, train, momentum, eps);
  else
    hipLaunchKernelGGL(bn1d_fwd_kernel<float>, grid, block, 0, st, z_out, gamma, beta, running_mean, running_var, save_mean, save_rstd,
                       (float*)y_dtype, y_f32, B, E, train, momentum, eps);
  ST_LAUNCH_CHECK();
  return 0;
}

extern "C" int st_linear_bn1d_backward(const float* dy, const float* z, const void* x, const float* gamma,
                                       const float* save_mean, const float* save_rstd, int B, int F, int E, int dtype, int train,
                                       float* dw, float* dbias, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                                       void* stream) {
  ST_CHECK(dy && z && x && gamma && save_mean && save_rstd && dw && dbias && dgamma && dbeta && workspace, "st_linear_bn1d_backward: null pointer");
  ST_CHECK(workspace_bytes >= st_head_workspace_bytes(B, F, E, dtype), "st_linear_bn1d_backward: workspace too small");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t es = st_dtype_size(dtype);
  const int Bp = up8(B), Ep = up8(E);
  char* ws = reinterpret_cast<char*>(workspace);
  char* dz = ws; ws += al((size_t)B * Ep * es);
  char* dzT = ws; ws += al((size_t)E * Bp * es);
  char* xT = ws;
  const dim3 grid((E + kBnTX - 1) / kBnTX), block(kBnTX, kBnTY);
  if (dtype == ST_BF16)
    hipLaunchKernelGGL(bn1d_bwd_kernel<bf16_t>, grid, block, 0, st, dy, z, gamma, save_mean, save_rstd, dgamma, dbeta, (bf16_t*)dz, B, E, Ep, train);
  else
    hipLaunchKernelGGL(bn1d_bwd_kernel<float>, grid, block, 0, st, dy, z, gamma, save_mean, save_rstd, dgamma, dbeta, (float*)dz, B, E, Ep, train);
  ST_LAUNCH_CHECK();
  // db += colsum(dz) ; dW += dz^T x
  if (st_transpose_colsum(dz, dzT, dbias, dtype, B, E, Ep, Bp, stream)) return 1;
  if (st_transpose(x, xT, dtype, B, F, F, Bp, stream)) return 1;
  return gemm_nt_(dzT, Bp, xT, Bp, dw, F, E, F, Bp, dtype, ST_F32, nullptr, 1, stream);
}

extern "C" int st_sgd_step(float* param, const float* grad, float* momentum_buf, void* bf16_shadow, long n,
                           float lr, float momentum, int first_step, float grad_scale, void* stream) {
  ST_CHECK(param && grad && (momentum == 0.f || momentum_buf), "st_sgd_step: null pointer");
  if (n <= 0) return 0;
  long blocks = (n / 4 + 255) / 256; if (blocks < 1) blocks = 1; if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(sgd_kernel, dim3((int)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), param, grad, momentum_buf,
                     (uint16_t*)bf16_shadow, n, lr, momentum, first_step, grad_scale);
  ST_LAUNCH_CHECK();
  return 0;
}

extern "C" int st_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* bf16_shadow, long n,
                            float lr, float beta1, float beta2, float eps, int step, float grad_scale, void* stream) {
  ST_CHECK(param && grad && exp_avg && exp_avg_sq && step >= 1, "st_adam_step: bad arguments");
  if (n <= 0) return 0;
  long blocks = (n / 4 + 255) / 256; if (blocks < 1) blocks = 1; if (blocks > 2048) blocks = 2048;
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adam_kernel, dim3((int)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), param, grad, exp_avg, exp_avg_sq,
                     (uint16_t*)bf16_shadow, n, lr, beta1, beta2, eps, bc1, sqrtf(bc2), grad_scale);
  ST_LAUNCH_CHECK();
  return 0;
}
