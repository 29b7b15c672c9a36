// Launch profiler shared by the convolution kernels (bench.py's roofline): HIP events around a launch, on its stream.
#pragma once
#include <hip/hip_runtime.h>

constexpr int kProfVariants = 16;   // st_prof_collect array length
// 0: bf16 128x128-tile igemm family   1: bf16 128x64   2: bf16 64x128   3: bf16 256x128   4..7: the f32 forms
// 8: image-resident 3x3 (conv_img.hip)   9: activation-stationary 1x1 (conv_img.hip)   10: streamed-K 1x1 (conv_img.hip)
struct StProfScope {
  bool on = false; size_t idx = 0;
  StProfScope(int variant, double flops, hipStream_t st);
  void end(hipStream_t st);
};

unsigned long long* st_debug_stamps_ptr();   // st_debug_stamps' buffer (NULL = off)
