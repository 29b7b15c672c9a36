// Device-side input transform: the reference's per-image pipeline of utils.py:84-88
//   Resize((224,224)) -> RandomHorizontalFlip -> RandomVerticalFlip -> ToTensor -> Normalize
// for a whole minibatch of ragged-size RGB uint8 images in two launches.
//
// Resize is Pillow's 8-bit BILINEAR resample (libImaging/Resample.c), reproduced bit for bit: per output index the
// triangle-filter taps over [center - support, center + support) are computed in double, normalised, rounded to
// 22-bit fixed point; a horizontal pass writes a uint8 intermediate, a vertical pass reads it.  The coin flips are
// the caller's (one flag word per image); ToTensor + Normalize are a 3x256 float table built by the caller with the
// reference's float32 arithmetic, so the float result is exact by construction.
//
// HBM-bound byte work: one read of the source pixels, one write + read of the [H][out_w] intermediate, one write of
// the fp32 planes.  Tap weights are computed once per block into LDS (tap-major, conflict-free) and shared by the
// block's strip of rows.  This file is compiled with -ffp-contract=off (Makefile): the double arithmetic of the tap
// weights must round exactly like the C original.
#include "common.h"

namespace {

constexpr int kMaxTaps = 40;       // ceil(support) * 2 + 1 with support = max(in / out, 1): scale factors up to 19
constexpr int kMaxOut = 320;       // tap table in LDS: at most kMaxTaps * kMaxOut ints = 50 KB
constexpr int kPrecisionBits = 32 - 8 - 2;
constexpr int kRowsH = 16;         // source rows per block of the horizontal pass
constexpr int kRowsV = 8;          // output rows per block of the vertical pass

// Resample.c precompute_coeffs + normalize_coeffs_8bpc for output index o of an in_size -> out_size axis.
// k is written with stride kstride (tap-major LDS table).  Returns the tap count; *first = first source index.
// Entries n .. taps-1 are zeroed (as Resample.c does).
__device__ int tap_coeffs(int in_size, int out_size, int o, int* k, int kstride, int taps, int* first) {
  const double scale = (double)((float)in_size - 0.0f) / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  const double center = 0.0 + (o + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  int lo = (int)(center - support + 0.5);
  if (lo < 0) lo = 0;
  int hi = (int)(center + support + 0.5);
  if (hi > in_size) hi = in_size;
  const int n = hi - lo;
  double ww = 0.0;
  for (int x = 0; x < n; ++x) {
    double a = (x + lo - center + 0.5) * ss;
    if (a < 0.0) a = -a;
    ww += a < 1.0 ? 1.0 - a : 0.0;
  }
  for (int x = 0; x < n; ++x) {
    double a = (x + lo - center + 0.5) * ss;
    if (a < 0.0) a = -a;
    double w = a < 1.0 ? 1.0 - a : 0.0;
    if (ww != 0.0) w /= ww;
    k[x * kstride] = (int)(0.5 + w * (double)(1 << kPrecisionBits));
  }
  for (int x = n; x < taps; ++x) k[x * kstride] = 0;
  *first = lo;
  return n;
}

__device__ __forceinline__ int clip8(int acc) {
  const int v = acc >> kPrecisionBits;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// tmp[b][y][xo][c] = horizontal resample of row y of image b.  grid (ceil(max_h / kRowsH), B), 256 threads.
// Dynamic LDS: tap table [taps][ow], first[ow], count[ow].  Two taps' channel bytes come from ONE unaligned 8-byte
// load and four pixels of the intermediate leave as three aligned dwords; the pass is bound by its integer VALU work
// (three byte loads per tap ran 1.4x slower, staging the rows in LDS and reading bytes there slower still); the last
// bytes of the buffer fall back to byte loads.
__global__ void __launch_bounds__(256) resize_h_kernel(st_image_batch_desc d, int taps) {
  extern __shared__ int smem[];
  const int ow = d.out_w;
  int* kc = smem;
  int* first = kc + taps * ow;
  int* count = first + ow;
  const int b = blockIdx.y;
  const int H = d.height[b], W = d.width[b];
  const int y0 = blockIdx.x * kRowsH;
  // sizes beyond the declared bounds, or an image that does not lie inside src: nothing is touched
  if (y0 >= H || H > d.max_height || W > d.max_width || d.offset[b] < 0 || d.offset[b] + (int64_t)H * W * 3 > d.src_bytes) return;
  for (int o = threadIdx.x; o < ow; o += 256) count[o] = tap_coeffs(W, ow, o, kc + o, ow, taps, first + o);
  __syncthreads();
  const uint8_t* src = d.src + d.offset[b];
  const uint8_t* src_end = d.src + d.src_bytes;
  uint8_t* tmp = d.tmp + (size_t)b * d.max_height * ow * 3;
  const int rows = min(kRowsH, H - y0);
  const bool packed = (ow & 3) == 0 && (reinterpret_cast<uintptr_t>(d.tmp) & 3) == 0;
  for (int idx = threadIdx.x; idx < rows * ow; idx += 256) {
    const int r = idx / ow, o = idx - r * ow;
    const int y = y0 + r;
    const uint8_t* p = src + ((size_t)y * W + first[o]) * 3;
    int a0 = 1 << (kPrecisionBits - 1), a1 = a0, a2 = a0;
    const int n = count[o];
    if (p + 3 * n + 2 <= src_end) {                              // every load below lies inside the buffer
      int x = 0;
      for (; x + 1 < n; x += 2) {                                // two taps = 6 bytes from one unaligned 8-byte load
        const int k0 = kc[x * ow + o], k1 = kc[(x + 1) * ow + o];
        uint32_t v[2];
        __builtin_memcpy(v, p + x * 3, 8);
        a0 += (int)(v[0] & 255u) * k0 + (int)(v[0] >> 24) * k1;
        a1 += (int)((v[0] >> 8) & 255u) * k0 + (int)(v[1] & 255u) * k1;
        a2 += (int)((v[0] >> 16) & 255u) * k0 + (int)((v[1] >> 8) & 255u) * k1;
      }
      if (x < n) {
        const int k = kc[x * ow + o];
        uint32_t v;
        __builtin_memcpy(&v, p + x * 3, 4);
        a0 += (int)(v & 255u) * k;
        a1 += (int)((v >> 8) & 255u) * k;
        a2 += (int)((v >> 16) & 255u) * k;
      }
    } else {                                                     // the buffer's last pixel: byte loads
      for (int x = 0; x < n; ++x) {
        const int k = kc[x * ow + o];
        a0 += (int)p[x * 3 + 0] * k;
        a1 += (int)p[x * 3 + 1] * k;
        a2 += (int)p[x * 3 + 2] * k;
      }
    }
    const uint32_t px = (uint32_t)clip8(a0) | ((uint32_t)clip8(a1) << 8) | ((uint32_t)clip8(a2) << 16);
    uint8_t* q = tmp + ((size_t)y * ow + o) * 3;
    if (packed) {
      // four neighbouring pixels = 12 bytes = three aligned dwords, written by three of the four lanes (three byte
      // stores per pixel were 41 M partial-dword writes per batch); ow % 4 == 0 keeps a quad inside one row and one loop trip
      const uint32_t nx = __shfl_down(px, 1, 64);
      const int j = o & 3;
      if (j < 3) *reinterpret_cast<uint32_t*>(q + j) = (px >> (8 * j)) | (nx << (24 - 8 * j));
    } else {
      q[0] = (uint8_t)px;
      q[1] = (uint8_t)(px >> 8);
      q[2] = (uint8_t)(px >> 16);
    }
  }
}

// out[b][c][yo'][xo'] = lut[c][vertical resample of tmp at (yo, xo)], (yo', xo') = flipped position.
// grid (ceil(out_h / kRowsV), B), 256 threads.  VEC: a thread owns 4 consecutive bytes of the [ow][3] row (one dword
// load per tap; needs ow % 4 == 0 and a 4-byte aligned tmp), otherwise one pixel (three byte loads per tap).
template <bool VEC>
__global__ void __launch_bounds__(256) resize_v_kernel(st_image_batch_desc d) {
  __shared__ int kc[kMaxTaps * kRowsV];
  __shared__ int first[kRowsV], count[kRowsV];
  __shared__ float lut[3 * 256];
  const int b = blockIdx.y;
  const int H = d.height[b];
  if (H > d.max_height || d.width[b] > d.max_width) return;
  const int oh = d.out_h, ow = d.out_w;
  const int y0 = blockIdx.x * kRowsV;
  const int rows = min(kRowsV, oh - y0);
  if ((int)threadIdx.x < rows) count[threadIdx.x] = tap_coeffs(H, oh, y0 + threadIdx.x, kc + threadIdx.x, kRowsV, 0, first + threadIdx.x);
  for (int i = threadIdx.x; i < 3 * 256; i += 256) lut[i] = d.lut[i];
  __syncthreads();
  const int flip = d.flip ? d.flip[b] : 0;
  const uint8_t* tmp = d.tmp + (size_t)b * d.max_height * ow * 3;
  const size_t plane = (size_t)oh * ow;
  float* outb = d.out + (size_t)b * 3 * plane;
  constexpr int PER = VEC ? 4 : 3;                                // bytes of the row a thread owns
  const int units = ow * 3 / PER;
  for (int idx = threadIdx.x; idx < rows * units; idx += 256) {
    const int r = idx / units, j = idx - r * units;
    const uint8_t* p = tmp + (size_t)first[r] * ow * 3 + j * PER;
    int acc[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) acc[i] = 1 << (kPrecisionBits - 1);
    const int n = count[r];
    for (int y = 0; y < n; ++y) {
      const int k = kc[y * kRowsV + r];
      const uint8_t* s = p + (size_t)y * ow * 3;
      if constexpr (VEC) {
        const uint32_t v = *reinterpret_cast<const uint32_t*>(s);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += (int)((v >> (8 * i)) & 255u) * k;
      } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[i] += (int)s[i] * k;
      }
    }
    const int yo = (flip & 2) ? oh - 1 - (y0 + r) : y0 + r;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int f = j * PER + i;
      const int x = f / 3, c = f - 3 * x;
      const int xo = (flip & 1) ? ow - 1 - x : x;
      const int v = clip8(acc[i]);
      outb[c * plane + (size_t)yo * ow + xo] = lut[c * 256 + v];
      if (d.out_u8) d.out_u8[(((size_t)b * oh + yo) * ow + xo) * 3 + c] = (uint8_t)v;
    }
  }
}

}  // namespace

extern "C" int st_image_transform(const st_image_batch_desc* d, void* stream) {
  ST_CHECK(d, "st_image_transform: null descriptor");
  ST_CHECK(d->src && d->offset && d->height && d->width && d->lut && d->tmp && d->out, "st_image_transform: null pointer");
  ST_CHECK(d->src_bytes > 0, "st_image_transform: src_bytes=%lld", (long long)d->src_bytes);
  ST_CHECK(d->batch > 0 && d->batch <= 65535, "st_image_transform: bad batch=%d", d->batch);
  ST_CHECK(d->out_h > 0 && d->out_w > 0 && d->out_h <= kMaxOut && d->out_w <= kMaxOut,
           "st_image_transform: output %dx%d outside 1..%d", d->out_h, d->out_w, kMaxOut);
  ST_CHECK(d->max_height > 0 && d->max_width > 0, "st_image_transform: bad bounds %dx%d", d->max_height, d->max_width);
  // tap table bound: ksize = ceil(max(in / out, 1)) * 2 + 1 (Resample.c) for the largest image of the batch
  const int kh = ((d->max_height + d->out_h - 1) / d->out_h) * 2 + 1, kw = ((d->max_width + d->out_w - 1) / d->out_w) * 2 + 1;
  ST_CHECK(kh <= kMaxTaps && kw <= kMaxTaps, "st_image_transform: %dx%d -> %dx%d needs %d taps (limit %d)",
           d->max_height, d->max_width, d->out_h, d->out_w, kh > kw ? kh : kw, kMaxTaps);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t lds = (size_t)(kw + 2) * d->out_w * sizeof(int);
  hipLaunchKernelGGL(resize_h_kernel, dim3((d->max_height + kRowsH - 1) / kRowsH, d->batch), dim3(256), lds, st, *d, kw);
  ST_LAUNCH_CHECK();
  const dim3 gv((d->out_h + kRowsV - 1) / kRowsV, d->batch);
  if (d->out_w % 4 == 0 && reinterpret_cast<uintptr_t>(d->tmp) % 4 == 0) hipLaunchKernelGGL(resize_v_kernel<true>, gv, dim3(256), 0, st, *d);
  else hipLaunchKernelGGL(resize_v_kernel<false>, gv, dim3(256), 0, st, *d);
  ST_LAUNCH_CHECK();
  return 0;
}
