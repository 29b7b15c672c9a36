/*
 * showtell_hip.h -- C ABI of libshowtell_hip.so, the MI355X (gfx950) kernels of the
 * show-tell captioning hot path.
 *
 * The reference (guptakhil/show-tell) has no FFI layer: its hot path is the Python
 * nn.Module surface (cnn.py, rnn.py, Attention/rnn_attn.py, LSTM/rnn_lstm.py) whose
 * arithmetic is delegated to torch / cuDNN / cuBLAS.  Each entry point below
 * replaces one such delegated call site (cited as file:line of /root/reference).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (0 = default stream);
 *   - every function returns 0 on success, non-zero on error;
 *     st_last_error() returns a thread-local message;
 *   - nothing is allocated, freed or synchronised inside a call (graph-capturable);
 *     workspaces are caller-provided;
 *   - dtype codes: ST_F32 = 0, ST_BF16 = 1.  bf16 tensors are raw uint16 bit patterns.
 *   - activations are NHWC ("pixel-major"), weights are [Cout][KH][KW][Cin].
 */
#ifndef SHOWTELL_HIP_H
#define SHOWTELL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ST_F32 0
#define ST_BF16 1

#define ST_CELL_GRU 0
#define ST_CELL_LSTM 1

const char* st_last_error(void);
int st_version(void);

/* ------------------------------------------------------------------------------------
 * Implicit-GEMM convolution / dense projection on MFMA.
 *   y[m][n] = sum_k x_gather[m][k] * w[n][k]        m = (b,ho,wo), k = (kh,kw,c)
 * Sliding-window form: KW == 1, pad == 0 and ldx < Cin with Cin % ldx == 0 reads Cin consecutive elements
 * starting at pixel (hi, wi), i.e. Cin/ldx neighbouring pixels of one row as one tap (the space-to-depth
 * stem); the caller's rows must be wide enough: (Wo-1)*stride + Cin/ldx <= Win.
 * Replaces: torchvision conv2d inside cnn.py:46 / cnn_attn.py:46 (cuDNN conv),
 * nn.Linear at cnn.py:49, rnn.py:33 and the GRU input projections inside rnn.py:32.
 * A plain GEMM is the case KH=KW=1, Hin=Win=Ho=Wo=1, B=M, Cin=K.
 * Epilogue (all optional, in this order):  v = acc + bias[n];  stats[n] += v, stats[N+n] += v*v
 * (fp32 atomics, batch-norm statistics; with stats_replicas = R > 1 the buffer is [R][2N] and pixel tile t
 * adds into replica t % R -- same-address float atomics serialise at the memory side, ~25 ns each, so layers
 * with thousands of pixel tiles spread them; the consumer sums the replicas);  v = v*scale[n] + shift[n];  v += residual[m][n];
 * v = max(v,0) if relu;  y = (accumulate ? y : 0) + v.
 * Requirements: Cin, ldx, ldw multiples of 8 (bf16) / 4 (f32); ldy multiple of 4.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const void* x;        /* [B,Hin,Win,ldx] dtype                                    */
  const void* w;        /* [N][ldw] dtype, K = KH*KW*Cin contiguous                 */
  void* y;              /* [M][ldy] out_dtype, M = B*Ho*Wo                          */
  const float* bias;    /* [N] or NULL                                              */
  const float* scale;   /* [N] or NULL                                              */
  const float* shift;   /* [N] or NULL                                              */
  const void* residual; /* [M][ldy] out_dtype or NULL                               */
  float* stats;         /* [2N] or NULL                                             */
  int dtype, out_dtype;
  int B, Hin, Win, Cin, Ho, Wo, N, KH, KW, stride, pad;
  int ldx, ldw, ldy;
  int relu, accumulate;
  int Cin_logical;      /* 0 = Cin; the un-padded channel count (profiler FLOP accounting) */
  int k_order;          /* 0: w is [N][KH][KW][Cin]; 1: w is [N][Cin/CH][KH][KW][CH], CH = 128 bytes of channels */
  int stats_replicas;   /* 0/1: stats is [2N]; R > 1: stats is [R][2N] (see above)               */
  /* Optional input transform (train-mode bottleneck, cnn.py:46): x is the RAW output of the producing conv and the
   * A operand becomes relu(batchnorm(x)) with batch statistics in_stats = [sum | sumsq] over in_count rows -- the
   * producer's normalise pass is absorbed by this conv.  Needs Cin % 64 == 0 (bf16) / 32 (f32).  NULL = off. */
  const float* in_stats; const float* in_gamma; const float* in_beta; float in_count, in_eps;
  int split_k;          /* > 1 (plain GEMM, fp32 output, K % (split_k*64|32) == 0): K is cut into split_k slices that run as
                         * one grouped launch and add their tiles with fp32 atomics -- for products with few output tiles */
} st_conv_desc;

int st_conv(const st_conv_desc* d, void* stream);
/* n <= 12 independent problems of identical shape, dtype and options (only the pointers differ) as ONE launch: the
 * per-layer weight-gradient GEMMs of the decoder backward (autograd of nn.GRU, main.py:151) are 48 tiles each. */
int st_conv_batch(const st_conv_desc* d, int n, void* stream);

/* ------------------------------------------------------------------------------------
 * Image-resident 3x3 stride-1 pad-1 convolution, bf16 (the conv2 of torchvision's Bottleneck / the convs of BasicBlock
 * behind `self.model(x)`, reference cnn.py:46 / cnn_attn.py:46; csrc/conv_img.hip).  A workgroup keeps a band of one
 * image (+ zero halo) in LDS for all nine taps, the filter bank streams fragment-major from HBM/L2 straight into MFMA
 * operand registers, no barrier in the K loop.  x: [B][H][W][C] bf16 (dense), y: [B][H][W][N] bf16 (dense),
 * w_frag: st_pack_conv_weight_frag(.., ntw = st_conv3x3_img_supported(H, W, C, N)).
 *   in_stats != NULL: x is the RAW output of the producer convolution; the fill applies relu(batchnorm(x)) with the
 *                     producer's batch statistics ([sum | sumsq] over in_count rows) once per element (bn1 + ReLU of
 *                     the Bottleneck in train mode, main.py:125) -- the separate normalise pass disappears.
 *   stats != NULL   : per-channel [sum(N) | sumsq(N)] of the fp32 results are added (stats_replicas as in st_conv_desc;
 *                     the replica is (image, band) %% R).
 *   scale/shift/relu: eval-mode epilogue y = relu?(acc * scale + shift).
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const void* x; const void* w_frag; void* y;
  float* stats; int stats_replicas;
  const float* scale; const float* shift; int relu;
  const float* in_stats; const float* in_gamma; const float* in_beta; float in_count; float in_eps;
  int B, H, W, C, N;
  int in_stats_replicas;   /* 0/1: in_stats is [2C]; R > 1: [R][2C], summed by the kernel (no reduction launch in between) */
} st_conv3x3_img_desc;
/* > 0: supported, the value is the LAYOUT CODE the weights must be packed with (st_pack_conv_weight_frag's `ntw` argument:
 * low byte = 16-channel tiles per wave, bits 8.. = channel halves of the K order); 0: use st_conv */
int st_conv3x3_img_supported(int H, int W, int C, int N);
int st_conv3x3_img(const st_conv3x3_img_desc* d, void* stream);
/* The stride-2 siblings (3x3, pad 1, stride 2: conv2 of the first block of layer2 / layer3 / layer4; csrc/conv_s2.hip): a K-streaming implicit
 * GEMM with gathered rows, (C, N) in {(128, 128), (256, 256), (512, 512)}.  Same descriptor (H, W = input size, y: [B][(H-1)/2+1][(W-1)/2+1][N]);
 * w_frag: st_pack_conv_weight_frag(.., KH = KW = 3, ntw = st_conv3x3_s2_supported(C, N)). */
int st_conv3x3_s2_supported(int C, int N);
int st_conv3x3_s2(const st_conv3x3_img_desc* d, void* stream);
/* ------------------------------------------------------------------------------------
 * Pointwise (1x1, stride 1 or 2) convolution with the filter slice held in registers, bf16, C <= 512 input channels
 * (conv3 / downsample of torchvision's Bottleneck and the narrow conv1s, reference cnn.py:46; csrc/conv_img.hip).
 * x: [B][Hin][Win][C] bf16 (dense), y: [B][Ho][Wo][N] bf16 with Ho = (Hin - 1) / stride + 1;
 * w_frag: st_pack_conv_weight_frag(.., KH = KW = 1, ntw = st_conv1x1_wreg_supported(C, N)).
 * in_stats / stats / scale / shift / relu as in st_conv3x3_img_desc.  residual ([B][Ho][Wo][N] bf16, the Bottleneck's identity): the
 * eval-mode conv3 epilogue y = relu(conv(x) * scale + shift + residual) -- needs scale / shift, stride 1, no statistics (st_conv1x1_wreg
 * and the stride-1 forms of st_conv1x1_astat; st_conv1x1_kstream has none).  y may be NULL in train mode (stats given): only the
 * statistics of the output are produced.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const void* x; const void* w_frag; void* y; const void* residual;
  float* stats; int stats_replicas;
  const float* scale; const float* shift; int relu;
  const float* in_stats; const float* in_gamma; const float* in_beta; float in_count; float in_eps; int in_stats_replicas;
  int B, Hin, Win, C, N, stride;
} st_conv1x1_wreg_desc;
int st_conv1x1_wreg_supported(int C, int N);
int st_conv1x1_wreg(const st_conv1x1_wreg_desc* d, void* stream);
/* The long-K sibling (C = 1024 / 2048 input channels, N a multiple of 256: the conv1 of the layer3 / layer4 Bottlenecks): a
 * workgroup owns 112 rows x 256 channels and streams K through a padded LDS ring (activations) and a register ring
 * (fragment-major filters, ntw = 4).  Same descriptor; in_stats and residual must be NULL. */
int st_conv1x1_kstream_supported(int C, int N);
int st_conv1x1_kstream(const st_conv1x1_wreg_desc* d, void* stream);
/* conv1 of a Bottleneck fused with the PREVIOUS block's end (torchvision Bottleneck.forward: out = relu(bn3(conv3(..)) + identity),
 * then the next block's conv1(out); reference cnn.py:46), train mode:
 *   x_out = relu(batchnorm(raw; f_stats, f_gamma, f_beta) + idn)   -- written once, it is the next identity
 *           idn = identity, or batchnorm(identity; id_stats, id_gamma, id_beta) when id_stats is given (the previous block had a
 *           downsample conv whose raw output is the identity; same count / eps as f_*)
 *   y     = x_out (*) w_frag (1x1 fragment-major weights, layout code st_conv1x1_kfuse_supported(C, N)),  stats += [sum | sumsq] of y
 * raw / identity / x_out: [rows][C] bf16 (x_out must not alias raw or identity), y: [rows][N] bf16.  Replaces one st_bn_act pass +
 * the conv1 launch.  C = 256 / 512 (layer1 / layer2 and the first conv1 of layer2 / layer3): the register-resident-filter kernel's
 * loader, same weights as st_conv1x1_wreg; C = 1024 -> N = 256: the K-streaming form (id_stats must be NULL; measured slower than the
 * separate pass, compiled only under ST_EXPERIMENTAL). */
typedef struct {
  const void* raw; const void* identity; void* x_out; const void* w_frag; void* y;
  float* stats; int stats_replicas;
  const float* f_stats; const float* f_gamma; const float* f_beta; float f_count; float f_eps; int f_stats_replicas;
  long rows; int C, N;
  const float* id_stats; const float* id_gamma; const float* id_beta; int id_stats_replicas;
} st_conv1x1_kfuse_desc;
int st_conv1x1_kfuse_supported(int C, int N);
int st_conv1x1_kfuse(const st_conv1x1_kfuse_desc* d, void* stream);
/* The 1024 -> 256 form as a producer / consumer workgroup (csrc/conv_kfuse8.hip): four waves multiply, four load / normalise / store; same
 * descriptor (id_stats must be NULL), same results bit for bit.  Measured a wash against st_bn_act + st_conv1x1_kstream (106.7 vs 108.0 us per
 * layer3 block); st_resnet_forward does not use it. */
#ifdef ST_EXPERIMENTAL   /* `make EXPERIMENTAL=1`: not part of the product library */
int st_conv1x1_kfuse8(const st_conv1x1_kfuse_desc* d, void* stream);
#endif
/* Block boundary of the 56 x 56 Bottlenecks in one pass over the wide tensors (csrc/conv_b2b.hip; train mode, bf16):
 *   x_out = relu(bn3(conv3(relu(bn2(raw2)))) + idn),   y = conv1_next(x_out)   (+ statistics of y)
 * conv3 (C1 = 64 -> C2 = 256) is RE-computed here from the narrow tensor raw2 [rows][64] instead of being written and read back;
 * its batch statistics (bn3_stats) come from a st_conv1x1_wreg call with y == NULL over the same raw2 / w3_frag.  idn = identity, or
 * batchnorm(identity; id_*) after a downsample conv.  w3_frag / w1_frag: the fragment-major copies st_conv1x1_wreg uses for the two
 * layers (C1 -> C2 and C2 -> N, N = 64 | 128).  Bit-identical to st_conv1x1_wreg -> st_bn_act -> st_conv1x1_wreg.  `count` rows went
 * into every statistic; x_out must not alias an input. */
typedef struct {
  const void* raw2; const void* w3_frag; const void* identity; void* x_out; const void* w1_frag; void* y;
  float* stats; int stats_replicas;
  const float* bn2_stats; const float* bn2_gamma; const float* bn2_beta; int bn2_replicas;
  const float* bn3_stats; const float* bn3_gamma; const float* bn3_beta; int bn3_replicas;
  const float* id_stats; const float* id_gamma; const float* id_beta; int id_replicas;
  float count; float eps;
  long rows; int C1, C2, N;
} st_conv_b2b_desc;
/* Layout contract (the kernel indexes the fragments with fixed tile permutations): w3_frag packed with ntw = 2, w1_frag with
 * ntw = N / 64 -- what st_conv1x1_wreg_supported returns for these layers today; a caller that packs another ntw gets wrong channels. */
int st_conv_b2b_supported(int C1, int C2, int N);
int st_conv_b2b(const st_conv_b2b_desc* d, void* stream);
/* conv3 of a 14 x 14 Bottleneck (C1 = 256 -> C2 = 1024) or a 28 x 28 one (128 -> 512), the block's end, and conv1 of the NEXT block
 * (C2 -> N = C1) in one kernel (csrc/conv_c3c1.hip; torchvision Bottleneck.forward twice over, reference cnn.py:46):
 *   x_out = relu(bn3(conv3(a2)) + identity),   y = conv1_next(x_out)
 * conv3's output channels are conv1's K dimension: each finished 128-channel chunk of x is stored once AND is one K-slab of conv1
 * (through LDS); x is never read back.  Replaces st_conv1x1_astat + st_bn_act + st_conv1x1_kstream (28 x 28: st_conv1x1_wreg +
 * st_bn_act + st_conv1x1_wreg).
 *   train (bn3_stats given): a2 = relu(batchnorm(x2; bn2_*)) (bn2_stats NULL: x2 is already normalised); bn3 from the batch statistics of
 *     conv3's output, which a STATISTICS-ONLY st_conv1x1_astat (28 x 28: st_conv1x1_wreg) call (y == NULL) over the same x2 / w3_frag
 *     produces first; `stats` receives
 *     [sum | sumsq] of y.  x_out and y are bit-identical to the three-kernel path.
 *   eval (scale3 / shift3 / scale1 / shift1 given, no statistics): a2 = x2, x_out = relu(conv3 * scale3 + shift3 + identity),
 *     y = conv1(x_out) * scale1 + shift1 (ReLU if relu1).
 * x2: [rows][C1], identity / x_out: [rows][C2], y: [rows][N], all bf16.  w3_frag: st_pack_conv_weight_frag(ntw = 2) (the layout
 * st_conv1x1_astat / st_conv1x1_wreg use for conv3), w1_frag: ntw = N / 64 (st_conv1x1_kstream's | st_conv1x1_wreg's for that conv1). */
typedef struct {
  const void* x2; const void* w3_frag; const void* identity; void* x_out; const void* w1_frag; void* y;
  float* stats; int stats_replicas;
  const float* bn2_stats; const float* bn2_gamma; const float* bn2_beta; int bn2_replicas;
  const float* bn3_stats; const float* bn3_gamma; const float* bn3_beta; int bn3_replicas;
  float count; float eps;
  const float* scale3; const float* shift3; const float* scale1; const float* shift1; int relu1;
  long rows; int C1, C2, N;
  /* train, optional: the identity is the RAW output of a downsample conv with these batch statistics (the block behind it):
   * x_out = relu(bn3(conv3) + batchnorm(identity; id_*)), st_bn_act's res_bn form */
  const float* id_stats; const float* id_gamma; const float* id_beta; int id_replicas;
} st_conv_c3c1_desc;
int st_conv_c3c1_supported(int C1, int C2, int N);
int st_conv_c3c1(const st_conv_c3c1_desc* d, void* stream);
/* The activation-stationary sibling for (C, N) = (256, 1024) / (512, 2048) (conv3 of the layer3 / layer4 Bottlenecks): a
 * workgroup keeps its 112 x C rows in LDS (producer's BatchNorm + ReLU applied once per element) and walks all N output
 * channels barrier-free, the epilogue of one 128-channel chunk under the next chunk's MFMAs.  Same descriptor (weights packed with the
 * layout code st_conv1x1_astat_supported returns), stride 1, residual NULL. */
int st_conv1x1_astat_supported(int C, int N);
int st_conv1x1_astat(const st_conv1x1_wreg_desc* d, void* stream);

/* [Cout][Cin][KH][KW] fp32 (torch layout) -> fragment-major bf16: element ((T * KS + ks) * 64 + lane) * 8 + j is
 * w[ch(T, lane & 15)][k = 32 ks + 8 (lane >> 4) + j] with k = (kh * KW + kw) * Cin + c and
 * ch(T, r) = (T / ntw) * 16 ntw + 4 ntw (r / 4) + 4 (T %% ntw) + r %% 4: one coalesced 1-KiB load is one MFMA operand
 * (16 output channels x 32 K), and a wave that owns ntw tiles ends with 4 ntw consecutive channels per lane. */
int st_pack_conv_weight_frag(const float* w, void* out, int Cout, int Cin, int KH, int KW, int ntw, void* stream);

/* Launch profiler for bench.py's roofline: HIP events around every convolution launch on its stream.
 * st_prof_collect fills 32-entry arrays indexed by kernel variant (0: bf16 128x128 tile family,
 * 1: bf16 128x64, 2: bf16 64x128, 3: 256x128, 4..7 the same for f32, 8..19: the csrc/conv_img.hip kernels, one slot per kernel symbol
 * (csrc/prof.h)); synchronise the device first. */
int st_tune(int reserved, int kc, int w8);   /* main-loop variant knobs for tools/bench_conv.py (row chunk count 4|8, block shape); -1 = keep */
int st_prof_enable(int on);
/* debug aid: per-block phase timestamps of st_conv launches (tools/conv_stamps.py); NULL = off (default) */
int st_debug_stamps(unsigned long long* buf);
int st_prof_collect(double* ms, double* flops, long* launches);

/* ------------------------------------------------------------------------------------
 * Batch-norm apply (+ residual)(+ ReLU), NHWC, fused elementwise pass.
 * Replaces: nn.BatchNorm2d + ReLU + residual add inside torchvision's Bottleneck
 * (cnn.py:46).  Train mode (main.py:125): stats = [sum | sumsq] from st_conv, count = B*H*W.
 *   scale = gamma*rsqrt(var+eps), shift = beta - mean*scale   (biased var)
 *   y = relu?( x*scale+shift + (res ? (res_stats ? res*rscale+rshift : res) : 0) )
 * Eval mode: pass stats = NULL and running_mean/var.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const void* x; void* y; const void* res;      /* [rows][C] dtype                  */
  const float* stats;  const float* gamma; const float* beta;
  const float* running_mean; const float* running_var;      /* eval mode            */
  const float* res_stats; const float* res_gamma; const float* res_beta;
  const float* res_running_mean; const float* res_running_var;
  int res_bn;          /* 1: residual goes through its own BN (downsample branch)   */
  int dtype; long rows; int C; float count; float eps; int relu;
  int stats_replicas, res_stats_replicas;   /* 0/1: [2C]; R > 1: [R][2C], summed here (st_conv_desc.stats_replicas) */
} st_bn_act_desc;

int st_bn_act(const st_bn_act_desc* d, void* stream);

/* running_mean/var momentum update from [sum|sumsq] (unbiased var), nn.BatchNorm semantics */
int st_bn_update_running(const float* stats, float* running_mean, float* running_var,
                         int C, float count, float momentum, void* stream);

/* NCHW fp32 images (utils.py:61-77 layout) -> NHWC dtype with channels zero-padded to Cpad */
int st_nchw_to_nhwc(const float* x, void* y, int dtype, int B, int C, int H, int W, int Cpad, void* stream);
/* NCHW fp32 RGB images (even H, W) -> 2x2 space-to-depth NHWC [B][H/2+3][W/2+3][16] dtype: blocked pixel (p,q) holds
 * rows 2(p-2)+dy, cols 2(q-2)+dx as channel (dy*2+dx)*3+c; channels 12..15 and the border (2 before, 1 after) are zero.
 * With st_stem_weight_s2d the torchvision stem (7x7 s2 p3, cnn.py:46) becomes st_conv's sliding-window form:
 * Hin=H/2+3, Win=W/2+3, Cin=64, ldx=16, KH=4, KW=1, stride=1, pad=0, Ho=H/2, Wo=W/2, ldw=256. */
int st_nchw_to_s2d16(const float* x, void* y, int dtype, int B, int H, int W, void* stream);
/* packed stem weights [64][7][7][Cpad] dtype (st_pack_conv_weight, k_order 0) -> [64][4][4][16] dtype for the blocked image */
int st_stem_weight_s2d(const void* w, void* out, int dtype, int Cpad, void* stream);
/* NHWC dtype -> (B,C,H*W) fp32, the layout cnn_attn.py:49 returns */
int st_nhwc_to_ncp_f32(const void* x, float* y, int dtype, int B, int HW, int C, void* stream);
/* 3x3 stride-2 pad-1 max pool NHWC (torchvision resnet maxpool, cnn.py:46) */
int st_maxpool3x3s2(const void* x, void* y, int dtype, int B, int H, int W, int C, void* stream);
/* y = maxpool3x3s2(relu(batchnorm(x))): the stem's bn1 + relu + maxpool of torchvision's resnet in one pass
 * (train: stats = [sum|sumsq] over `count` rows; eval: stats NULL, running buffers).  C must divide 2048 (bf16) / 1024 (f32). */
int st_maxpool3x3s2_bn(const void* x, void* y, int dtype, int B, int H, int W, int C,
                       const float* stats, const float* gamma, const float* beta,
                       const float* running_mean, const float* running_var, float count, float eps, void* stream);
/* The bf16 stem in one kernel (csrc/conv_stem.hip): conv1 7x7/2 (+ BatchNorm statistics) + maxpool 3x3/2 of torchvision's resnet
 * (the first four modules of the reference's `self.model`, cnn.py:46); replaces st_conv on the blocked image + st_maxpool3x3s2(_bn).
 *   x_s2d : st_nchw_to_s2d16's blocked image [B][H/2+3][W/2+3][16] bf16;  w_frag : st_stem_weight_frag(st_stem_weight_s2d(..)) (64 x 256)
 *   y     : [B][PH][PW][64] bf16, PH = (H/2 - 1)/2 + 1
 * train (scale == NULL): y = pool(raw conv output) with MAX on channels with gamma >= 0 and MIN on the others -- the consumer applies
 *   relu(bn1(.)) to y (monotone per channel, so this IS maxpool(relu(bn1(conv)))); stats[replica][sum(64) | sumsq(64)] += the raw
 *   convolution output's sums over all B x H/2 x W/2 positions (fp32 accumulators), replica = workgroup %% stats_replicas.
 * eval (scale, shift = folded bn1): y = maxpool(relu(conv * scale + shift)). */
typedef struct {
  const void* x_s2d; const void* w_frag; void* y;
  float* stats; int stats_replicas;
  const float* gamma;
  const float* scale; const float* shift;
  int B, H, W;
} st_stem_conv_pool_desc;
int st_stem_conv_pool(const st_stem_conv_pool_desc* d, void* stream);
int st_stem_weight_frag(const void* w_s2d, void* out, void* stream);
/* == st_stem_weight_frag(st_stem_weight_s2d(w_packed, bf16, Cpad)) in one launch (what st_resnet_forward runs per forward) */
int st_stem_weight_frag_packed(const void* w_packed, int Cpad, void* out, void* stream);
/* global average pool NHWC -> [B][C] (adaptive avgpool, cnn.py:34) */
int st_global_avgpool(const void* x, void* y, int dtype, int out_dtype, int B, int HW, int C, void* stream);

/* Device-side input transform of a minibatch (utils.py:84-88: Resize((224,224)) -> RandomHorizontalFlip ->
 * RandomVerticalFlip -> ToTensor -> Normalize, applied per image at utils.py:45-47 and stacked at utils.py:68).
 * Resize is Pillow's 8-bit BILINEAR resample, bit for bit (horizontal pass, uint8 intermediate, vertical pass);
 * the coin flips are the caller's; `lut` = the caller's float32 table ((v / 255) - mean[c]) / std[c].
 * Limits: out_h, out_w <= 320; down-scaling factors up to 19 per axis. */
typedef struct st_image_batch_desc {
  const uint8_t* src;      /* device: the batch's RGB images, HWC uint8, back to back */
  int64_t src_bytes;       /* size of src (offset[b] + 3 * height[b] * width[b] <= src_bytes for every b) */
  const int64_t* offset;   /* device [batch]: byte offset of image b in src */
  const int32_t* height;   /* device [batch] */
  const int32_t* width;    /* device [batch] */
  const int32_t* flip;     /* device [batch]: bit 0 = left-right, bit 1 = top-bottom; NULL = no flips */
  int batch;
  int max_height, max_width; /* bounds of height[] / width[] (sizes the launch and the scratch) */
  int out_h, out_w;
  const float* lut;        /* device [3][256] */
  uint8_t* tmp;            /* device scratch, batch * max_height * out_w * 3 bytes */
  float* out;              /* device (batch, 3, out_h, out_w) fp32: the `images` tensor of utils.py:68 */
  uint8_t* out_u8;         /* optional device (batch, out_h, out_w, 3): the resized + flipped pixels before ToTensor */
} st_image_batch_desc;
int st_image_transform(const st_image_batch_desc* d, void* stream);

/* Generic helpers */
int st_cast(const void* x, void* y, int from_dtype, int to_dtype, long n, void* stream);
/* y[c][r] = x[r][c]; y has leading dimension ldy >= rows, pad columns zero-filled */
int st_transpose(const void* x, void* y, int dtype, int rows, int cols, int ldx, int ldy, void* stream);
/* the same, and colsum[c] += sum_r x[r][c] (fp32 atomics) from the tile already in LDS: the bias gradient that goes with
 * every K-major copy of a gradient matrix in the decoder backward (autograd of nn.Linear / nn.GRU, main.py:151) */
int st_transpose_colsum(const void* x, void* y, float* colsum, int dtype, int rows, int cols, int ldx, int ldy, void* stream);
/* n <= 12 matrices of identical shape in one launch (the same operand of every decoder layer); colsum may be NULL or hold
 * NULL entries.  With rows_t / prev_row (packed-sequence tables, st_packed_seq) output row r is read from source row
 * prev_row[r], zero where rows_t[r] == 0: h_{t-1} of every token transposed straight out of the layer output. */
int st_transpose_batch(const void* const* x, void* const* y, float* const* colsum, int n, int dtype, int rows, int cols,
                       int ldx, int ldy, const int* rows_t, const int* prev_row, void* stream);
/* conv weight repack: [Cout][Cin][KH][KW] fp32 (torch layout) -> [Cout][KH][KW][Cpad] dtype (k_order 0)
 * or [Cout][Cpad/CH][KH][KW][CH] with CH = 64 (bf16) / 32 (f32) channels (k_order 1, see st_conv_desc) */
int st_pack_conv_weight(const float* w, void* out, int dtype, int Cout, int Cin, int KH, int KW, int Cpad, int k_order, void* stream);

/* ------------------------------------------------------------------------------------
 * ResNet-{18,34,50,101,152} backbone forward, one call (torchvision children()[:-1] /
 * [:-2]; reference cnn.py:23-34,46 and cnn_attn.py:23-34,46-49).
 * Layers are indexed in forward order: stem conv, then per block conv1, conv2, [conv3],
 * [downsample]; st_resnet_conv_info gives geometry and the element offsets of layer i in
 * the packed weight buffer ([Cout][KH][KW][Cin_padded], dtype) and in the concatenated
 * BatchNorm arrays (gamma, beta, running_mean, running_var: fp32, one entry per channel).
 * st_resnet_create returns 2 with the reference's ValueError text (cnn.py:33) for an
 * unknown version.  Outputs (each optional): feat_nhwc_out (B,h,w,F) dtype;
 * pooled_out (B,F) pooled_dtype (cnn.py:48); ncp_out (B,F,h*w) fp32 (cnn_attn.py:49).
 * train != 0: batch statistics + running-buffer update (main.py:125); else eval mode.
 * train == 2: batch statistics, but the running buffers are left for st_resnet_update_running (same workspace) -- for
 * forwards of successive minibatches that run concurrently on different streams and must apply their momentum updates in order.
 * ---------------------------------------------------------------------------------- */
typedef struct st_resnet st_resnet;
int st_resnet_create(int version, int dtype, st_resnet** out);
void st_resnet_destroy(st_resnet* r);
int st_resnet_num_convs(const st_resnet* r);
int st_resnet_feat_dim(const st_resnet* r);
size_t st_resnet_weight_elems(const st_resnet* r);
size_t st_resnet_bn_channels(const st_resnet* r);
int st_resnet_conv_info(const st_resnet* r, int i, int* cin, int* cout, int* k, int* stride, int* pad,
                        int* cin_padded, size_t* weight_offset, size_t* bn_offset, int* k_order,
                        size_t* frag_weight_offset, int* frag_ntw);
/* frag_ntw > 0: layer i also needs a fragment-major copy of its filters at element offset frag_weight_offset
 * (st_pack_conv_weight_frag(.., ntw = frag_ntw)) for the image-resident kernel; 0: none. */
size_t st_resnet_workspace_bytes(const st_resnet* r, int B, int H, int W);
/* Test / diagnosis aid: with a buffer set, every st_resnet_forward on this handle also copies each residual block's output
 * ([B][h][w][C] in the compute dtype, NHWC) into it, block after block in network order (ResNet-101 @224: 33 blocks, 17.9 MB per
 * image in bf16); buf = NULL switches it off.  The block-by-block parity test feeds these to the oracle one block at a time. */
int st_resnet_set_taps(st_resnet* r, void* buf, size_t bytes);
int st_resnet_forward(const st_resnet* r, const float* images_nchw, int B, int H, int W,
                      const void* weights, const float* bn_gamma, const float* bn_beta,
                      float* bn_running_mean, float* bn_running_var,
                      int train, float momentum, float eps,
                      void* workspace, size_t workspace_bytes,
                      void* feat_nhwc_out, void* pooled_out, int pooled_dtype, float* ncp_out,
                      void* stream);

int st_resnet_update_running(const st_resnet* r, const void* workspace, float* bn_running_mean, float* bn_running_var,
                             float momentum, void* stream);

int st_cast2d(const void* x, void* y, int from_dtype, int to_dtype, int rows, int cols, int ldx, int ldy, void* stream);

/* ------------------------------------------------------------------------------------
 * Teacher-forced recurrent decoder over a packed sequence.
 * Replaces RNN.forward (rnn.py:27-35; LSTM/rnn_lstm.py:25-33): nn.Embedding + cat +
 * pack_padded_sequence + nn.GRU/nn.LSTM (cuDNN fused RNN) + nn.Linear, and the autograd
 * backward of the same graph (main.py:151).
 * Rows are time-major packed as pack_padded_sequence orders them: row(t,b) = off[t] + b.
 *   rows_b/rows_t: (b, t) of every packed row; prev_row: index of row(t-1,b) (unused for t=0).
 *   batch_sizes_host: HOST array, non-increasing (captions sorted by length, utils.py:66).
 * Weights are [G*H][in] / [G*H][H] in `dtype` (gate order r,z,n / i,f,g,o as torch), biases fp32.
 * The workspace carries the saved activations from st_rnn_forward to st_rnn_backward.
 * st_rnn_forward: logits[ntok][ldl] (optional) and targets[ntok] = caption[b][t] (main.py:145).
 * st_rnn_backward: gradients are ACCUMULATED (+=) into the fp32 buffers of st_rnn_grads;
 *   dlogits is [ntok][ldd] in `dtype` with ldd a multiple of 8 and zero pad columns;
 *   dfeat[B][E] (fp32) receives the gradient w.r.t. the image feature rows.
 * ---------------------------------------------------------------------------------- */
#define ST_MAX_LAYERS 8
typedef struct {
  int cell, dtype, L, in0, H, V, E;
  const void* emb;                                   /* [V][E] dtype                  */
  const void* w_ih[ST_MAX_LAYERS]; const void* w_hh[ST_MAX_LAYERS];
  const float* b_ih[ST_MAX_LAYERS]; const float* b_hh[ST_MAX_LAYERS];
  const void* w_lin; const float* b_lin;             /* [V][H] dtype, [V] fp32        */
} st_rnn_params;

typedef struct {
  float* emb;
  float* w_ih[ST_MAX_LAYERS]; float* w_hh[ST_MAX_LAYERS]; float* b_ih[ST_MAX_LAYERS]; float* b_hh[ST_MAX_LAYERS];
  float* w_lin; float* b_lin;
} st_rnn_grads;

typedef struct {
  int B, T, ntok, Tcap;
  const int* batch_sizes_host;
  const int* rows_b; const int* rows_t; const int* prev_row;   /* device int32 [ntok] */
  const long* caption;                                          /* device int64 [B][Tcap] */
} st_packed_seq;

/* leading dimension (elements) for the logits / dlogits rows of a V-entry vocabulary: up8(V) below 2048 entries, else V
 * rounded up to 512 (the backward product over K = V is then split into 8 even slices); pad columns are zero-filled */
int st_rnn_vocab_ld(int V);
size_t st_rnn_workspace_bytes(const st_rnn_params* p, const st_packed_seq* s);
int st_rnn_forward(const st_rnn_params* p, const st_packed_seq* s, const void* x0_override, const void* feat,
                   void* workspace, size_t workspace_bytes, void* logits, int logits_dtype, int ldl,
                   long* targets, int save_for_backward, void* stream);
int st_rnn_backward(const st_rnn_params* p, const st_rnn_grads* g, const st_packed_seq* s,
                    const void* x0_override, const void* dlogits, int ldd, const float* dy_top,
                    void* workspace, size_t workspace_bytes, float* dfeat, float* dx0_out, void* stream);

/* Vocabulary projection (rnn.py:33) + nn.CrossEntropyLoss() (main.py:94,149) WITHOUT a logits tensor (csrc/vocab_ce.hip; bf16, H = 512):
 * after st_rnn_forward(.., logits = NULL, .., targets, save_for_backward = 1):
 *   st_rnn_fused_loss:    *loss_accum += mean_r( logsumexp(x_r) - x_r[target_r] ), x = y_top W_lin^T + b_lin computed tile by tile;
 *                         `scratch` (st_rnn_fused_loss_bytes) keeps the rows' logsumexp for the backward call;
 *   st_rnn_fused_dlogits: dlogits[ntok][ldd] (bf16) = (softmax - onehot) / ntok * *grad_scale_dev from the same tile products, pad columns
 *                         [V, ldd) zero: the operand st_rnn_backward takes.
 * Replaces st_rnn_forward's logits + two st_cross_entropy passes (the logits are never written, dlogits once). */
int st_rnn_fused_loss_supported(const st_rnn_params* p);
size_t st_rnn_fused_loss_bytes(const st_rnn_params* p, const st_packed_seq* s);
int st_rnn_fused_loss(const st_rnn_params* p, const st_packed_seq* s, const void* workspace, size_t workspace_bytes,
                      const long* targets, float* scratch, size_t scratch_bytes, float* loss_accum, void* stream);
int st_rnn_fused_dlogits(const st_rnn_params* p, const st_packed_seq* s, const void* workspace, size_t workspace_bytes,
                         const long* targets, const float* scratch, const float* grad_scale_dev, void* dlogits, int ldd, void* stream);

/* nn.CrossEntropyLoss() (mean) forward + backward (main.py:94,149):
 *   *loss_accum += mean_r( logsumexp(x_r) - x_r[target_r] );  dlogits = (softmax - onehot) * grad_scale / rows
 * dlogits may alias logits when the dtypes match; pad columns [V, ldd) are zero-filled. */
int st_cross_entropy(const void* logits, int logits_dtype, const long* target, int rows, int V, int ldl,
                     float* loss_accum, void* dlogits, int dlogits_dtype, int ldd, float grad_scale,
                     const float* grad_scale_dev /* optional device scalar multiplied in */, void* stream);

/* Encoder head: y = BatchNorm1d(x W^T + b) (cnn.py:37-38,49; momentum 0.01) and its backward
 * (dx is not needed: the backbone output is detached, cnn.py:47).  Gradients are accumulated. */
size_t st_head_workspace_bytes(int B, int F, int E, int dtype);
int st_linear_bn1d_forward(const void* x, const void* w, const float* bias, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, int B, int F, int E, int dtype,
                           int train, float momentum, float eps,
                           float* z_out, float* save_mean, float* save_rstd, void* y_dtype, float* y_f32, void* stream);
int st_linear_bn1d_backward(const float* dy, const float* z, const void* x, const float* gamma,
                            const float* save_mean, const float* save_rstd, int B, int F, int E, int dtype, int train,
                            float* dw, float* dbias, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                            void* stream);

/* torch.optim.SGD(lr, momentum) / torch.optim.Adam(lr) (main.py:97-100,152) over one flat fp32
 * buffer; bf16_shadow (optional) receives the updated parameters rounded to bf16. */
int st_sgd_step(float* param, const float* grad, float* momentum_buf, void* bf16_shadow, long n,
                float lr, float momentum, int first_step, float grad_scale, void* stream);
int st_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* bf16_shadow, long n,
                 float lr, float beta1, float beta2, float eps, int step, float grad_scale, void* stream);

/* Greedy decoding, exactly `steps` iterations with no early stop (rnn.py:37-58, rnn_lstm.py:35-57):
 * ids_out[B][steps] int64 (first-maximum tie rule of torch.max).  logits_out (optional, tests):
 * [steps][B][Vp] fp32 with Vp = V rounded up to 8. */
size_t st_rnn_greedy_workspace_bytes(const st_rnn_params* p, int B);
int st_rnn_greedy(const st_rnn_params* p, const void* feat, int B, int steps, void* workspace, size_t workspace_bytes,
                  long* ids_out, float* logits_out, void* stream);

/* Building blocks of the beam decoders (rnn.py:60-108 and beam_search.py:45-97):
 * st_rnn_step: one timestep of the L-layer cell for n independent rows, states [L][n][H] in `dtype`
 *   (h_in/c_in NULL = zero state, as nn.GRU(x, None)), optional fp32 logits [n][ldl] = linear(h_top);
 * st_embedding_rows: x[r] = embeddings[ids[r]] (rnn.py:53,85);
 * st_gather_state: dst[l][r] = src[l][idx[r]] (a child beam inherits its parent's recurrent state);
 * st_softmax_topk: per row the k largest entries in DESCENDING order with their indices; values are
 *   softmax probabilities (beam_search.py:83-88) or, with raw != 0, the raw logits (rnn.py:90-91). */
int st_rnn_step(const st_rnn_params* p, const void* x, int n, const void* h_in, const void* c_in,
                void* h_out, void* c_out, float* logits, int ldl, void* stream);
int st_embedding_rows(const void* emb, const long* ids, void* out, int n, int E, int V, int ldo, int dtype, void* stream);
int st_gather_state(const void* src, const int* idx, void* dst, int L, int n_src, int n_dst, int H, int dtype, void* stream);
int st_softmax_topk(const float* logits, int ldl, int n, int V, int k, float* top_p, long* top_id, int raw, void* stream);
/* One iteration of beam_search.py:69-94 for B images on the device: fringe slots (b, w), w < W, hold (tok, cost) with cost = +inf
 * for an empty slot; top_p/top_id are st_softmax_topk's rows of the B*W slots (k entries, descending).  Writes the new fringe
 * (stable W-best of the node-major, ascending-probability candidate list by float32 cumulative -log p), every new node's parent
 * slot (-1: empty), ended[b][w] = old slot held <end> (to be harvested), gather[b*W+w] = state row of the parent, and updates
 * done[b] (no live node left).  Needs k <= W and W * k <= 64. */
int st_beam_select(const long* tok, const float* cost, uint8_t* done, const float* top_p, const long* top_id, int B, int W, int k,
                   long end_id, long* new_tok, float* new_cost, int* parent, uint8_t* ended, int* gather, void* stream);

/* ------------------------------------------------------------------------------------
 * Soft-attention decoder (Attention/rnn_attn.py, rnn_attn_LSTM.py; train step Attention/main_attn.py:123-134).
 *   features: cnn_feature (B,F,P) fp32 as cnn_attn.py:49 returns it; caption_T: int64 [Tcap][B] (transposed captions).
 *   Per step t (rnn_attn.py:66-74): attention over P pixels keyed on the last layer's h, x = [emb(cap[:,t]) ; embed(z)],
 *   one step of the L-layer cell on the first B_t rows, logits rows written time-major packed (rnn_attn.py:115).
 *   alphas (B,T,P) fp32 must be zero-filled by the caller (the reference zero-pads, rnn_attn.py:65).
 *   The time-invariant encoder_att projection is computed once per batch instead of once per step.
 * st_attn_backward accumulates (+=) every parameter gradient, including the doubly-stochastic term
 *   alpha_c * mean((1 - sum_t alpha)^2) (main_attn.py:131) scaled like the loss (grad_scale_dev, optional).
 * st_attn_reg_loss adds that term to *loss_accum.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  st_rnn_params rnn;                               /* rnn.in0 = 2E */
  int F, A, P;
  const void* w_enc; const float* b_enc;           /* attn.encoder_att [A][F] dtype, [A] */
  const void* w_dec; const float* b_dec;           /* attn.decoder_att [A][H] dtype, [A] */
  const float* w_full; const float* b_full;        /* attn.full_att    [A] fp32, [1]     */
  const void* w_init_h; const float* b_init_h;     /* init_h [H][F] dtype                */
  const void* w_init_c; const float* b_init_c;     /* init_c (LSTM only)                 */
  const void* w_embed; const float* b_embed;       /* embed [E][F] dtype                 */
} st_attn_params;

typedef struct {
  st_rnn_grads rnn;
  float *w_enc, *b_enc, *w_dec, *b_dec, *w_full, *b_full, *w_init_h, *b_init_h, *w_init_c, *b_init_c, *w_embed, *b_embed;
} st_attn_grads;

size_t st_attn_workspace_bytes(const st_attn_params* p, const st_packed_seq* s);
int st_attn_forward(const st_attn_params* p, const st_packed_seq* s, const float* cnn_feature, const long* caption_T,
                    void* workspace, size_t workspace_bytes, void* logits, int logits_dtype, int ldl,
                    float* alphas, int save_for_backward, void* stream);
int st_attn_backward(const st_attn_params* p, const st_attn_grads* g, const st_packed_seq* s, const long* caption_T,
                     const void* dlogits, int ldd, const float* alphas,
                     const float* dalphas /* (B,T,P) gradient w.r.t. alphas from the caller, or NULL: use alpha_c */,
                     float alpha_c, const float* grad_scale_dev, void* workspace, size_t workspace_bytes, void* stream);
int st_attn_reg_loss(const float* alphas, int B, int T, int P, float alpha_c, float* loss_accum, void* stream);
/* rnn_attn.py:120-145 (test branch 77-94): `steps` greedy iterations from <start>; ids_out[B][steps] int64 */
size_t st_attn_greedy_workspace_bytes(const st_attn_params* p, int B);
int st_attn_greedy(const st_attn_params* p, const float* cnn_feature, int B, int steps, long start_id,
                   void* workspace, size_t workspace_bytes, long* ids_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
