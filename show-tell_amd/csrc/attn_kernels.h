// Internal launch interface of the attention-step kernels.
#pragma once
#include "common.h"
int ncp_to_pf_launch(const float* x, void* y, int B, int F, int P, int dtype, hipStream_t st);
int attn_fwd_launch(const void* att1, const float* att2, const float* wf, const float* bf, const void* feat, float* alpha_out,
                    long alpha_stride, void* z, int n, int P, int A, int F, int dtype, hipStream_t st);
int attn_bwd_launch(const float* dz, const float* dalpha_extra, long extra_stride, const float* alpha, long alpha_stride,
                    const void* att1, const float* att2, const float* wf, const void* feat, float* datt2, float* datt1_acc,
                    float* dwf, float* dbf, int n, int P, int A, int F, int dtype, hipStream_t st);
int attn_reg_launch(const float* alphas, int B, int T, int P, float alpha_c, float* loss, float* dalpha, const float* gscale_dev, hipStream_t st);
int add_rows_launch(float* dst, const float* src, long n, hipStream_t st);
int replicate_rows_launch(const void* src, void* dst, long n, int copies, int dtype, hipStream_t st);
int split_dx0_launch(const float* dx0, const long* ids, float* demb, void* dez, int rows, int E, int V, int dtype, hipStream_t st);
