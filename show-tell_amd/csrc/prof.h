// Launch profiler shared by the convolution kernels (bench.py's roofline): HIP events around a launch, on its stream.
#pragma once
#include <hip/hip_runtime.h>

constexpr int kProfVariants = 32;   // st_prof_collect array length
// 0: bf16 128x128-tile igemm family   1: bf16 128x64   2: bf16 64x128   3: bf16 256x128   4..7: the f32 forms
// conv_img.hip, one slot per kernel symbol rocprof would list:  8..11: conv3x3_img_kernel<C = 64, 128, 256, 512>
// 12..15: conv1x1_wreg_kernel<K = 64, 128, 256, 512>   16, 17: conv1x1_kstream_kernel<K = 1024, 2048> (+ kfuse)   18, 19: conv1x1_astat_kernel<K = 256, 512>
// 20: stem_pool_kernel   21: conv_b2b_kernel   22: conv_c3c1_kernel<256, 1024, 256>   23: conv3x3s2_kstream_kernel   24: conv_c3c1_kernel<128, 512, 128>
struct StProfScope {
  bool on = false; size_t idx = 0;
  StProfScope(int variant, double flops, hipStream_t st);
  void end(hipStream_t st);
};

unsigned long long* st_debug_stamps_ptr();   // st_debug_stamps' buffer (NULL = off)
