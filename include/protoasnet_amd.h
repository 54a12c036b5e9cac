/*
 * protoasnet_amd.h -- C-ABI of the MI355X (gfx950) ProtoASNet hot path.
 *
 * The reference (hooman007/ProtoASNet) has no FFI / operator registry: its hot path is a
 * stack of torch.nn calls inside three nn.Modules.  This header is
 * the boundary a replacement exports *under* that nn.Module surface: one entry point per
 * fused stage, each citing the reference call site(s) it replaces.  The Python host side
 * (the protoasnet_amd Python package) binds these with ctypes and keeps the reference's module surface
 * (forward / push_forward / compute_occurence_map / state_dict keys) on top.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / C++ types.
 *   - Every pointer is DEVICE memory owned by the caller (torch's caching allocator).  The
 *     library never allocates, frees or retains device memory.  Workspaces are passed in.
 *   - Work is enqueued asynchronously on the caller's HIP stream (`stream`, a hipStream_t);
 *     no internal streams, no host synchronisation, safe to capture into a hipGraph.
 *   - Return 0 on success; non-zero = error, message via pasn_last_error() (thread-local).
 *   - dtype: PASN_F32 = 0, PASN_BF16 = 1.  Arithmetic always accumulates in fp32.
 *   - Activations are CHANNELS-LAST: [N][T][H][W][Cp] (T = 1 for images) with the channel
 *     stride Cp a multiple of 8 and channels >= C stored as zeros.  A logical NC(T)HW view
 *     of such a buffer is what the nn.Module surface hands back to reference callers.
 */
#ifndef PROTOASNET_AMD_H
#define PROTOASNET_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PASN_VERSION 100 /* round 1 */

enum { PASN_F32 = 0, PASN_BF16 = 1, PASN_U8 = 2 /* input clips of the *_gray_fwd entry points only */ };
enum { PASN_ACT_NONE = 0, PASN_ACT_RELU = 1, PASN_ACT_SIGMOID = 2, PASN_ACT_SWISH = 3, PASN_ACT_ABS = 4 };
enum { PASN_OK = 0, PASN_ERR_ARG = 1, PASN_ERR_LAUNCH = 2, PASN_ERR_UNSUPPORTED = 3 };

int pasn_version(void);
const char* pasn_last_error(void);

/* Tuning switches (csrc/tuning.h).  The reference has no counterpart: its kernels are cuDNN's, picked by `cudnn.benchmark = True`
 * (src/agents/base.py:20).  Here the routing of a layer onto a kernel is a pure function of its descriptor and of the PASN_* switches
 * that were set in the environment WHEN THE LIBRARY FIRST LOOKED (one snapshot; registered names only).  pasn_tuning_reload() takes a new
 * snapshot (tests and A/B tools call it after changing the environment); pasn_tuning_get() returns a switch's value in the snapshot or
 * NULL; pasn_tuning_report() writes "NAME=VALUE" lines of the switches in force (+ an "unknown: ..." line for PASN_* names the library
 * does not know, + the registry "name<TAB>class<TAB>meaning" when with_registry != 0) and returns the length needed. */
void pasn_tuning_reload(void);
const char* pasn_tuning_get(const char* name);
int pasn_tuning_report(char* buf, int cap, int with_registry);

/* Geometry of one convolution / pooling window over channels-last activations. */
typedef struct pasn_conv_desc {
    int32_t N, Ti, Hi, Wi;    /* input extent                                              */
    int32_t Cin, Cin_p;       /* input channels, input channel stride (elements)           */
    int32_t To, Ho, Wo;       /* output extent                                             */
    int32_t Cout, Cout_p;     /* output channels, output channel stride (elements)         */
    int32_t kt, kh, kw;       /* window                                                    */
    int32_t st, sh, sw;       /* stride                                                    */
    int32_t pt, ph, pw;       /* zero padding (max-pool: -inf padding)                     */
    int32_t act;              /* PASN_ACT_* applied after scale/bias (+ residual)          */
    int32_t in_swish;         /* conv3d: apply x*sigmoid(x) to the INPUT while loading it  */
    int32_t w_kc;             /* conv3d: per-tap K extent of the packed weight (elements)  */
    int32_t w_rows;           /* conv3d: rows of the packed weight / scale / bias arrays   */
    int32_t w_frag;           /* conv3d: 0 = w is [w_rows][taps][w_kc]; 1 = MFMA-fragment-major
                                 [w_rows/32][taps*w_kc/KSTEP][2][32][CH] (KSTEP/CH = 16/8 bf16, 8/4 fp32; K = (tap, channel)):
                                 only where pasn_conv3d_variant() reports 2500..5999 or >= 7000 */
} pasn_conv_desc;

/*
 * First layer: planar (N,3,T,H,W) clip -> channels-last activation, window (1,kh,kw), fused
 * per-channel scale/bias (folded eval-mode BatchNorm) and activation.
 * Replaces: resnet_features.py:203-205 (conv1+bn1+relu), torchvision R2Plus1dStem conv 0-2
 * (call site resnet_features.py:316-320), X3D stem conv_xy.
 *   x      : in_dtype  [N][3][Ti][Hi][Wi]
 *   w      : fp32 [3*kh*kw][Cout_p]  (tap-major, channel-minor; rows ordered ci, r, s)
 *   scale, bias : fp32 [Cout_p]
 *   y      : out_dtype [N][To][Ho][Wo][Cout_p]
 */
int pasn_first_conv_fwd(const void* x, const float* w, const float* scale, const float* bias, void* y,
                        const pasn_conv_desc* d, int in_dtype, int out_dtype, void* stream);

/*
 * Fused X3D stem: (1,3,3) stride-(1,2,2) conv 3->24 -> depthwise (5,1,1) temporal conv -> BN -> ReLU in one launch
 * (a thread marches along T with a 5-frame register ring; the 24-channel tensor between the two convs never reaches
 * HBM).  Bit-identical to pasn_first_conv_fwd (no norm, no activation) followed by pasn_dwconv3d_fwd.
 *   d    : geometry of the (1,3,3) conv as for pasn_first_conv_fwd; pasn_x3d_stem_supported(d) says whether it fits
 *   w_xy : fp32 [27][24] (rows ordered ci, r, s);  w_t : fp32 [5][24];  scale, bias : fp32 [24] (BN after the temporal conv)
 */
int pasn_x3d_stem_supported(const pasn_conv_desc* d);
int pasn_x3d_stem_fwd(const void* x, const float* w_xy, const float* w_t, const float* scale, const float* bias, void* y,
                      const pasn_conv_desc* d, int in_dtype, int out_dtype, void* stream);

/*
 * The same stem on the matrix cores, bf16 activations out, Wi % 4 == 0 (pasn_x3d_stem_mfma_supported says when): conv_xy and conv_t as
 * ONE linear map over 5 frames x Cin x 3 rows x 4-wide window slots, weights multiplied together by the caller:
 *   wq : bf16 [2*ceil(5*Cin*3/4)][32][8], K row R = (kt*Cin + ci)*3 + r:  wq[R/2][co][4*(R%2) + 1 + s] = w_t[co][kt] * w_xy[co][ci][r][s],
 *        zero elsewhere
 * d->Cin = 3 or 1 (grey clip, taps summed over the input channels); x' = x * in_a + in_b while staging (1, 0 = none).  A bf16 tolerance
 * path (no rounding of the 24-channel intermediate): fp32 activations keep pasn_x3d_stem_fwd.
 */
int pasn_x3d_stem_mfma_supported(const pasn_conv_desc* d, int in_dtype, int out_dtype);
int pasn_x3d_stem_mfma_fwd(const void* x, const void* wq, const float* scale, const float* bias, void* y, const pasn_conv_desc* d,
                           int in_dtype, float in_a, float in_b, void* stream);

/*
 * Device side of the input pipeline.  The reference dataloader resizes a single-channel cine on the host,
 * normalises it ((x - 0.099) / 0.171, as_dataloader.py:180-182), repeats it to 3 identical channels (:168-170) and ships fp32
 * (N,3,T,H,W).  These two entry points take the SINGLE channel instead -- planar [N][1][Ti][Hi][Wi], fp32 / bf16 / uint8 -- apply
 * x' = x * in_a + in_b while loading (in_a = 1/std or 1/(255 std), in_b = -mean/std; 1, 0 for an already normalised clip) and use
 * first-conv weights summed over the three input channels: w [kh*kw][Cout_p] / w_xy [9][24].  Zero padding pads the normalised
 * tensor, as in the reference.  Same outputs as the 3-channel entry points fed the expanded clip, up to fp32 summation order.
 * d->Cin must be 1.
 */
/*
 * The same first layer on the matrix cores, bf16 activations out: (1,kh,kw <= 7) windows, stride (1,2,2) -- the 7x7 stems of
 * R(2+1)D-18 and ResNet-18 (resnet_features.py:203-205, :316-320); d->Cin = 3 (the reference's clip) or 1 (grey clip, taps summed,
 * as for pasn_first_conv_gray_fwd).  pasn_first_conv_mfma_slot returns -1 when the layer is not covered (use pasn_first_conv_fwd), else
 * the slot o of tap 0 in the 8-wide window the weights must be laid out for:
 *   wq : bf16 [2*ceil(Cin*kh/2)][32*ceil(Cout_p/32)][8],  wq[ci*kh + r][co][o + s] = w[co][ci][r][s], zero elsewhere
 *   x' = x * in_a + in_b is applied while the clip is staged (1, 0 = none); zero padding pads the normalised tensor.
 */
int pasn_first_conv_mfma_slot(const pasn_conv_desc* d, int in_dtype, int out_dtype);
int pasn_first_conv_mfma_fwd(const void* x, const void* wq, const float* scale, const float* bias, void* y, const pasn_conv_desc* d,
                             int in_dtype, float in_a, float in_b, void* stream);
int pasn_first_conv_gray_fwd(const void* x, const float* w, const float* scale, const float* bias, void* y,
                             const pasn_conv_desc* d, int in_dtype, int out_dtype, float in_a, float in_b, void* stream);
int pasn_x3d_stem_gray_fwd(const void* x, const float* w_xy, const float* w_t, const float* scale, const float* bias, void* y,
                           const pasn_conv_desc* d, int in_dtype, int out_dtype, float in_a, float in_b, void* stream);

/*
 * Dense convolution as an implicit GEMM on the matrix cores (MFMA), channels-last, groups=1, any
 * window/stride/padding, fused epilogue  y = act(acc*scale + bias [+ residual]).
 * Optional fused input transform (X3D project conv): x' = swish(x * gate[n][ci]).
 * Replaces: every nn.Conv2d/Conv3d + BatchNorm (+ReLU, + residual add) of the trunks
 * (resnet_features.py:49-66,202-213; torchvision Conv2Plus1D/BasicBlock at :316-320), the 1x1(x1)
 * add-on convs of PPNet (ProtoPNet.py:117-130) and the X3D expand / project / shortcut convs.
 *   x     : dtype [N][Ti][Hi][Wi][Cin_p]
 *   w     : dtype [w_rows][kt*kh*kw][w_kc] (or fragment-major, see w_frag), zero padded (w_rows multiple of 128, w_kc multiple of
 *           16 (bf16) / 8 (fp32) and >= Cin_p)
 *   scale, bias : fp32 [w_rows]
 *   residual : dtype [N][To][Ho][Wo][Cout_p] or NULL
 *   gate  : fp32 [N][Cin_p] or NULL
 *   y     : dtype [N][To][Ho][Wo][Cout_p]
 */
int pasn_conv3d_fwd(const void* x, const void* w, const float* scale, const float* bias, const void* residual,
                    const float* gate, void* y, const pasn_conv_desc* d, int dtype, void* stream);
/*
 * Two chained 1x1x1 convs on the same positions (bf16): y1 = act1(scale1 * (w1 . x') + bias1 + residual) with the optional
 * fused input transform x' = swish(x * gate), then y2 = act2(scale2 * (w2 . y1) + bias2) computed from y1 while it is
 * still on chip (y1 is written too).  The X3D pattern: a block's project conv (ResBlock.conv_c + BN + residual + ReLU)
 * and the next block's expand conv (conv_a + BN + ReLU).  Both weights fragment-major (w_frag = 1).
 * pasn_conv3d_pair_supported() says whether the geometry is covered; otherwise call pasn_conv3d_fwd twice.
 */
int pasn_conv3d_pair_supported(const pasn_conv_desc* d1, const pasn_conv_desc* d2, int dtype);
/* Which kernel the pair takes: 0 none, 1 pwconv_xpair_kernel (a block per 64-position tile), 2 pwconv_ws_kernel in pair mode (persistent
 * blocks, both weight sets in registers).  flags bit 0: a gate tensor will be passed.  For profilers and benchmarks. */
int pasn_conv3d_pair_variant(const pasn_conv_desc* d1, const pasn_conv_desc* d2, int dtype, int flags);
/* The chained pair whose first conv's squeeze-excite gate is computed in the launch's prologue from the stencil's pool partial rows
 * (arguments as pasn_conv3d_se_fwd + pasn_conv3d_pair_fwd): stencil -> ONE launch -> next stencil for an X3D SE block that is followed by a
 * block without shortcut.  _supported() = 0: pasn_se_gate_fwd + pasn_conv3d_pair_fwd. */
int pasn_conv3d_pair_se_supported(const pasn_conv_desc* d1, const pasn_conv_desc* d2, int dtype, int Cse);
int pasn_conv3d_pair_se_fwd(const void* x, const void* w1, const float* scale1, const float* bias1, const void* residual,
                            const float* pool_partial, int pool_blocks, int positions, const float* fc1_w, const float* fc1_b,
                            const float* fc2_w, const float* fc2_b, int Cse, void* y1, const pasn_conv_desc* d1, const void* w2,
                            const float* scale2, const float* bias2, void* y2, const pasn_conv_desc* d2, int dtype, void* stream);
int pasn_conv3d_pair_fwd(const void* x, const void* w1, const float* scale1, const float* bias1, const void* residual,
                         const float* gate, void* y1, const pasn_conv_desc* d1, const void* w2, const float* scale2,
                         const float* bias2, void* y2, const pasn_conv_desc* d2, int dtype, void* stream);

/*
 * pasn_conv3d_fwd with the squeeze-excite gate of its input transform computed IN THE SAME LAUNCH (bf16, fragment-major weights): every
 * block derives the gate rows of the (at most two) clips its rows touch from the stencil's pool partial rows -- mean over positions,
 * fc1 + ReLU, fc2 + sigmoid, as pasn_se_gate_fwd -- while its first tiles are in flight, then x' = swish(x * gate) as usual.  Replaces the
 * stand-alone pasn_se_gate_fwd launch between the stencil and the project conv of an X3D SE block (and the gate tensor).
 *   pool_partial : fp32 [N][pool_blocks][Cin_p] (pasn_dwconv3d_fwd's partial sums), positions = T*H*W of the pooled tensor
 *   fc1_w : fp32 [Cse][Cin], fc1_b : [Cse], fc2_w : [Cin][Cse], fc2_b : [Cin];  d->in_swish says whether Swish follows the gate
 * _supported() = 0: call pasn_se_gate_fwd + pasn_conv3d_fwd.
 */
int pasn_conv3d_se_supported(const pasn_conv_desc* d, int dtype, int Cse, int has_residual);
int pasn_conv3d_se_fwd(const void* x, const void* w, const float* scale, const float* bias, const void* residual,
                       const float* pool_partial, int pool_blocks, int positions, const float* fc1_w, const float* fc1_b,
                       const float* fc2_w, const float* fc2_b, int Cse, void* y, const pasn_conv_desc* d, int dtype, void* stream);

/*
 * A 1x1x1 stride-1 conv + norm with the block's STRIDED 1x1x1 shortcut conv + norm accumulated in the same launch (bf16; the first
 * project conv of an X3D stage, whose residual is the strided shortcut of the block input):
 *     y = act(scale * (w . x') + scale2 * (w2 . x2[strided position]) + bias),   x' = swish(x * gate) as in pasn_conv3d_fwd,
 * bias = the two norms' shifts, summed by the caller.  d = the stride-1 conv, d2 = the shortcut conv (same N, To, Ho, Wo, Cout_p;
 * kt = kh = kw = 1, st = 1, any sh / sw; row-major weights).  Replaces the pair of launches pasn_conv3d_fwd(shortcut) ->
 * pasn_conv3d_fwd(project, residual = shortcut) and the shortcut tensor between them.  _supported() = 0: use that pair.
 */
int pasn_conv3d_short_supported(const pasn_conv_desc* d, const pasn_conv_desc* d2, int dtype);
int pasn_conv3d_short_fwd(const void* x, const void* w, const float* scale, const float* bias, const float* gate, const void* x2,
                          const void* w2, const float* scale2, void* y, const pasn_conv_desc* d, const pasn_conv_desc* d2, int dtype,
                          void* stream);

/* Which kernel instance pasn_conv3d_fwd picks for this geometry: 1000 + KS*10 + NT = pwconv_persist_kernel<dtype, KS, NT>
 * (1x1x1 stride-1 convs whose weights fit 64 VGPRs per lane); 2500 + 2*KS (+1 with in_swish) = pwconv_xtile_kernel<dtype, KS, ..>
 * (1x1x1 stride-1 convs with Cin_p >= 64: whole-K position tiles in LDS); 2000 / 2001 = gemm_conv_kernel<dtype, pointwise /
 * windowed> (LDS-tiled implicit GEMM); otherwise NT*10 + MT = conv3d_mfma_kernel<dtype, NT, MT> (output-channel /
 * position tiles per wave); 7000 + KS*10 + MT = pwconv_ws_kernel<KS, MT, ..> (bf16 1x1x1 stride-1 convs with Cin_p >= 48:
 * weight-stationary persistent blocks, LDS-DMA stage ring; fragment-major weights like 2500+); 9000 + KSF = tconv_ws_kernel<KSF, ..>
 * (bf16 (3,1,1) stride-1 convs with Cin_p = 16 KSF in {48, 64, 144} and up to 64 output channels -- torchvision's Conv2Plus1D temporal half as
 * resnet_features.py's r2plus1d_18 trunk instantiates it: weight-stationary, T-marching LDS ring; fragment-major weights over K = 3 Cin_p);
 * 0 on a bad descriptor.
 * flags: bit 0 = `gate` will be non-NULL, bit 1 = `residual` will be non-NULL (the choice between the pointwise kernels and
 * their tile sizes depend on both).  For profilers, benchmarks and the weight packing (w_frag). */
int pasn_conv3d_variant(const pasn_conv_desc* d, int dtype, int flags);

/*
 * Depthwise convolution (groups = C), channels-last, fused scale/bias/activation; optionally also
 * emits per-block partial channel sums of the (pre-activation) output for the squeeze-excite pool.
 * Replaces: X3D stem conv_t + BN + ReLU and X3D conv_b + BN (+Swish); HBM-bound stencil.
 *   w     : fp32 [kt*kh*kw][Cp]
 *   pool_partial : fp32 [N][pool_blocks][Cp] or NULL; pool_blocks = pasn_dwconv3d_pool_blocks(d, dtype)
 */
int pasn_dwconv3d_pool_blocks(const pasn_conv_desc* d, int dtype);
/* Kernel instance for this geometry: 3000 + WT*10 + SW = dwconv3d_march_kernel<SW, WT> (bf16, 3x3x3, stride (1,s,s));
 * WT*100 + KW*10 + SW = dwconv3d_strip_kernel<dtype, WT, KW, SW>; 50001 = dwconv3d_mfma_kernel (stride-1 3x3x3 on the matrix cores);
 * 60001 = dwconv3d_tz_kernel (stride-1 3x3x3, planes 9 .. 14 wide: Toeplitz form); 70000 + kt = dwconv_t_kernel<dtype, kt> ((kt,1,1),
 * kt = 3 / 5, stride 1, taken when pool_partial is NULL: the X3D stem's conv_t as a launch of its own); 0 = generic kernel. */
int pasn_dwconv3d_variant(const pasn_conv_desc* d, int dtype);
/* The same stencil with the squeeze-excite gate of the block fused into the launch (X3D conv_b + SE: global average pool ->
 * fc1 + ReLU -> fc2 + sigmoid): every block writes its pool partial row, the clip's last-arriving block reduces them and computes
 * gate[n][:].  Same results as pasn_dwconv3d_fwd followed by pasn_se_gate_fwd up to fp32 summation order (fixed: repeat runs are
 * bitwise equal).  counter: int32 [N], zero before the FIRST launch (the kernel leaves it zero).  Supported only where
 * pasn_dwconv3d_se_supported returns 1 (bf16 3x3x3 layers on the T-marching kernel). */
int pasn_dwconv3d_se_supported(const pasn_conv_desc* d, int dtype, int Cse);
int pasn_dwconv3d_se_pool_blocks(const pasn_conv_desc* d, int dtype); /* rows of pool_partial per clip for pasn_dwconv3d_se_fwd */
int pasn_dwconv3d_se_fwd(const void* x, const float* w, const float* scale, const float* bias, void* y, float* pool_partial,
                         const pasn_conv_desc* d, int dtype, const float* fc1_w, const float* fc1_b, const float* fc2_w,
                         const float* fc2_b, int Cse, float* gate, int32_t* counter, void* stream);
int pasn_dwconv3d_fwd(const void* x, const float* w, const float* scale, const float* bias, void* y,
                      float* pool_partial, const pasn_conv_desc* d, int dtype, void* stream);

/*
 * Squeeze-excite gate: mean over positions (from the partial sums above, fixed summation order),
 * fc1 + ReLU, fc2 + sigmoid.   gate : fp32 [N][Cp].
 *   w1 : fp32 [Cse][C], b1 : fp32 [Cse], w2 : fp32 [C][Cse], b2 : fp32 [C]
 */
int pasn_se_gate_fwd(const float* pool_partial, int pool_blocks, int positions, const float* w1, const float* b1,
                     const float* w2, const float* b2, float* gate, int N, int C, int Cp, int Cse, void* stream);

/* Max pooling, channels-last.  Replaces nn.MaxPool2d(3,2,1) at resnet_features.py:206. */
int pasn_maxpool3d_fwd(const void* x, void* y, const pasn_conv_desc* d, int dtype, void* stream);

/*
 * Head A -- PPNet prototype layer.  Replaces PPNet._l2_convolution (ProtoPNet.py:189-207: the ones-conv
 * for ||x||^2, the prototype conv for x.p, relu), the global min pooling + distance_2_similarity + last_layer
 * of PPNet.forward (ProtoPNet.py:236-241), and the distance map of push_forward (ProtoPNet.py:245-249).
 *   z        : dtype [N][S][Dp]      add-on output (after Sigmoid), channels-last
 *   protos   : fp32 [P][D]
 *   fc_w     : fp32 [K][P]           last_layer.weight
 *   dist     : fp32 [N][P][S] or NULL  (full map, planar like the reference's (N,P,H,W))
 *   min_dist : fp32 [N][P]
 *   argmin   : int32 [N][P] or NULL  first s attaining the minimum
 *   logits   : fp32 [N][K]
 *   activation: 0 = log((d+1)/(d+eps)), 1 = linear (-d)
 */
int pasn_l2_head_fwd(const void* z, const float* protos, const float* fc_w, float* dist, float* min_dist,
                     int32_t* argmin, float* logits, int N, int S, int D, int Dp, int P, int K, int dtype,
                     int activation, float eps, void* stream);

/*
 * Head B -- XProtoNet / Video_XProtoNet ("ProtoASNet") prototype layer: add-on convs, occurrence module + abs,
 * occurrence-weighted pooling WITHOUT materialising the (N,P,D,S) broadcast product of Video_XProtoNet.py:87 /
 * XProtoNet.py:56, cosine similarity (torch semantics: each vector divided by max(norm, 1e-8) first), (s+1)/2,
 * last layer.  Replaces everything after the trunk in Video_XProtoNet.forward / push_forward
 * (Video_XProtoNet.py:82-130), XProtoNet.forward / push_forward (XProtoNet.py:51-106) and, with mode = 1,
 * the head part of compute_occurence_map (Video_XProtoNet.py:100-109, XProtoNet.py:69-85).
 *   x      : dtype [N][S][Cbp]  trunk features, channels-last
 *   a1,a2  : add_on_layers.{0,2}.weight;  o1,o2,o3 : occurrence_module.{0,2,4}.weight -- each packed like a
 *            pasn_conv3d_fwd weight with one tap: dtype [round_up(Cout_p,128)][round_up(Cin_p, 16|8)], zero padded
 *   a1b,a2b,o1b,o2b : fp32 biases [round_up(Cout_p,128)], zero padded (o3 has no bias)
 *   protos : fp32 [P][D];  fc_w : fp32 [K][P]
 *   occ    : fp32 [N][P][S]   planar, i.e. the reference's (N,P,1,[T,]H,W)
 *   feat   : fp32 [N][P][D]   features_extracted;   sim : fp32 [N][P];   logits : fp32 [N][K]
 *   ws     : workspace of pasn_xproto_head_workspace_bytes() bytes, 256-byte aligned
 *   mode   : 0 = full forward, 1 = occurrence map only (feat / sim / logits may be NULL)
 */
typedef struct pasn_xproto_desc {
    int32_t N, S;
    int32_t Cb, Cbp;   /* trunk channels and channel stride        */
    int32_t D, Dp;     /* prototype depth, round_up(D, 8)          */
    int32_t Hd, Hp;    /* D / 2 (hidden width of the occurrence module), round_up(Hd, 8) */
    int32_t P, Pp;     /* prototypes, round_up(P, 8)               */
    int32_t K;         /* classes                                   */
    int32_t mode;
} pasn_xproto_desc;

/*
 * Front half of an X3D stage's first block in ONE launch (bf16): conv_a (1x1x1) + norm_a + ReLU -> conv_b (depthwise 3x3x3, stride
 * (1,2,2), pad 1) + norm_b (+ the descriptor's activation, + squeeze-excite pool partial rows), pytorchvideo's BottleneckTransform as
 * instantiated by the reference's x3d trunks; replaces pasn_conv3d_fwd + pasn_dwconv3d_fwd for that pair -- the expanded activation (2.25x
 * the block width at the input resolution) stays in LDS.
 *   x  : block input, channels-last [N][T][H][W][de->Cin_p];   y : [N][T][Ho][Wo][d->Cout_p]
 *   wa : conv_a weights FRAGMENT-MAJOR (w_frag = 1), scale_a / bias_a: folded norm_a [de->w_rows] (zero beyond the channels).
 *        scale_a == NULL: the caller has folded norm_a's scale into wa (W * scale, rounded to bf16 once); bias_a then initialises the
 *        fp32 accumulators and the expand epilogue is ReLU + rounding only (what the plan compiler passes)
 *   w  : conv_b weights fp32 [27][Cp], scale / bias: folded norm_b [Cp]
 *   pool_partial : NULL or fp32 [N][pasn_x3d_expdw_pool_blocks()][Cp] (sums of the pre-activation output over positions, fixed order)
 * _supported() == 0: issue the two launches.
 */
int pasn_x3d_expdw_supported(const pasn_conv_desc* de, const pasn_conv_desc* d, int dtype);
int pasn_x3d_expdw_pool_blocks(const pasn_conv_desc* de, const pasn_conv_desc* d, int dtype);
/* which kernel pasn_x3d_expdw_fwd runs for the pair: 0 = block-diagonal stencil operands (x3d_expdw.hip), 1 = per-channel Toeplitz operands on a
 * channel-planar image (x3d_expdw_tz.hip, stride 1, round 5), -1 = pair not covered.  Same arguments, same results up to fp32 summation order. */
int pasn_x3d_expdw_variant(const pasn_conv_desc* de, const pasn_conv_desc* d, int dtype);
int pasn_x3d_expdw_fwd(const void* x, const void* wa, const float* scale_a, const float* bias_a, const float* w, const float* scale,
                       const float* bias, void* y, float* pool_partial, const pasn_conv_desc* de, const pasn_conv_desc* d, int dtype,
                       void* stream);

/*
 * A WHOLE X3D residual block without squeeze-excite on 7 x 7 planes (the last stage: block width 192, inner width 432) in ONE launch (bf16):
 * conv_a (1x1x1) + norm_a + ReLU -> conv_b (depthwise 3x3x3, pad 1) + norm_b + Swish -> conv_c (1x1x1) + norm_c + x + ReLU, optionally followed by
 * the NEXT block's conv_a + norm_a + ReLU (pytorchvideo's ResBlock / BottleneckTransform as the x3d trunks instantiate them).  Replaces
 * pasn_conv3d_fwd + pasn_dwconv3d_fwd + pasn_conv3d_fwd (/ pasn_x3d_pe_fwd): the expanded activation and the stencil's output exist only as LDS
 * images of a (clip, two frames) tile; results are bit-identical to those launches.
 *   x        : block input = residual, channels-last [N][T][7][7][d_a->Cin_p];   y : block output, same shape
 *   w_a, w_c : fragment-major (w_frag = 1); d_a->w_kc = Cin_p, d_c->w_kc = 32 * ceil(Cin_p / 32) (K zero-padded to an even number of steps)
 *   w_dw     : conv_b weights as the stencil's matrix-core operands: uint16 [ceil(Cp / 16)][2][64][8], row (16-channel tile, half, lane):
 *              entry e = kt * 5 + j (half e >> 3, slot e & 7; entry 15 unused) = bf16 bits (round-to-nearest-even) of tap
 *              kt * 9 + 2 j + (lane >> 5) of channel 16 tile + (lane & 15), for lanes with ((lane >> 3) & 1) == ((lane >> 4) & 1), a tap index
 *              < 9 and a real channel; 0 otherwise -- the one possibly nonzero element per lane of the block-diagonal A operands that
 *              pasn_dwconv3d_fwd's matrix-core kernel builds in its own prologue; scale / bias: the folded norms
 *   w_n ...  : the next block's conv_a in w_a's form and its output e_next [N][T][7][7][d_n->Cout_p]; all NULL (with d_n == NULL): none
 * _supported() == 0: issue the separate launches.
 */
int pasn_x3d_edp_supported(const pasn_conv_desc* d_a, const pasn_conv_desc* d_dw, const pasn_conv_desc* d_c, const pasn_conv_desc* d_n, int dtype);
int pasn_x3d_edp_fwd(const void* x, const void* w_a, const float* scale_a, const float* bias_a, const void* w_dw, const float* scale_dw,
                     const float* bias_dw, const void* w_c, const float* scale_c, const float* bias_c, void* y, const void* w_n,
                     const float* scale_n, const float* bias_n, void* e_next, const pasn_conv_desc* d_a, const pasn_conv_desc* d_dw,
                     const pasn_conv_desc* d_c, const pasn_conv_desc* d_n, int dtype, void* stream);

/*
 * An X3D block's conv_c (1x1x1) + norm_c + residual + ReLU chained with the NEXT block's conv_a (1x1x1) + norm_a + ReLU in ONE launch for the
 * 432-channel stage (bf16; inner width 432 -> block width 192 -> 432), with the block's squeeze-excite gate -- when it has one -- computed in the
 * launch's prologue from the stencil's pool partial rows: same contract and results (bit-identical) as pasn_conv3d_pair_se_fwd /
 * pasn_conv3d_pair_fwd, which do not cover this width (both weight sets do not fit a wave's registers); the weights are streamed per row tile
 * instead.  Differences in the operands: w1 is fragment-major with K zero-padded to an EVEN number of 16-wide steps (d1->w_kc = 32 *
 * ceil(Cin_p / 32)); pool_partial == NULL (with Cse = 0): no gate, x is the block's final stencil output (d1->in_swish = 0).
 * _supported() == 0: issue the separate launches.
 */
int pasn_x3d_pe_supported(const pasn_conv_desc* d1, const pasn_conv_desc* d2, int dtype, int Cse);
int pasn_x3d_pe_fwd(const void* x, const void* w1, const float* scale1, const float* bias1, const void* residual, const float* pool_partial,
                    int pool_blocks, int positions, const float* fc1_w, const float* fc1_b, const float* fc2_w, const float* fc2_b, int Cse,
                    void* y1, const pasn_conv_desc* d1, const void* w2, const float* scale2, const float* bias2, void* y2,
                    const pasn_conv_desc* d2, int dtype, void* stream);



int pasn_xproto_head_splits(const pasn_xproto_desc* d);
size_t pasn_xproto_head_workspace_bytes(const pasn_xproto_desc* d, int dtype);
int pasn_xproto_head_fwd(const void* x, const void* a1, const float* a1b, const void* a2, const float* a2b,
                         const void* o1, const float* o1b, const void* o2, const float* o2b, const void* o3,
                         const float* protos, const float* fc_w, float* occ, float* feat, float* sim, float* logits,
                         void* ws, const pasn_xproto_desc* d, int dtype, void* stream);

/*
 * Head B in two launches (csrc/head_chain.hip + the finish kernel): same contract and outputs as pasn_xproto_head_fwd (replaces
 * Video_XProtoNet.forward, Video_XProtoNet.py:66-98, after the trunk), for bf16 with D = 256, Hd = 128, P <= 64 and a trunk channel
 * stride <= 256 (the X3D heads of BASELINE configs 2, 3 and 5; the reference's own R(2+1)D-18[:-3] video trunk).  A block keeps a tile of <= 104 positions of one clip in LDS through the
 * five convs and adds the tile's share of the occurrence-weighted pooling; no intermediate map reaches memory.
 *   a1 .. o3 : conv weights FRAGMENT-MAJOR, [rows / 32][kc / 16][64][8] bf16 (the w_frag = 1 layout of pasn_conv3d_fwd)
 *   ws       : pasn_xproto_chain_workspace_bytes() bytes, 256-byte aligned (the pooling slabs [N][tiles][P][D] fp32; mode 1: unused)
 * pasn_xproto_chain_supported() == 0: use pasn_xproto_head_fwd.
 */
int pasn_xproto_chain_supported(const pasn_xproto_desc* d, int dtype);
size_t pasn_xproto_chain_workspace_bytes(const pasn_xproto_desc* d);
int pasn_xproto_chain_fwd(const void* x, const void* a1, const float* a1b, const void* a2, const float* a2b,
                          const void* o1, const float* o1b, const void* o2, const float* o2b, const void* o3,
                          const float* protos, const float* fc_w, float* occ, float* feat, float* sim, float* logits,
                          void* ws, const pasn_xproto_desc* d, int dtype, void* stream);

/*
 * Push sweep, XProtoNet / Video rule (push_abs_revision.py:288-307): per prototype j, over the clips of
 * this batch whose label matches class(j) (or all clips when class_mask[j] == 0): batch min of
 * proto_dist[:, j], first index; accepted when min <= best (a later batch wins ties).  State stays on the device.
 *   proto_dist : fp32 [B][P]   (1 - similarity)
 *   feat       : fp32 [B][P][D]
 *   labels     : int64 [B]
 *   proto_class: int32 [P]     argmax of prototype_class_identity
 *   class_mask : int32 [P]     1 = class specific
 *   best_dist  : fp32 [P]  (init +inf);  best_index : int64 [P] (init -1; global clip index = index_base + b)
 *   best_feat  : fp32 [P][D]
 */
int pasn_push_xproto_update(const float* proto_dist, const float* feat, const int64_t* labels, const int32_t* proto_class,
                            const int32_t* class_mask, float* best_dist, int64_t* best_index, float* best_feat, int B, int P,
                            int D, int64_t index_base, void* stream);

/*
 * Push sweep, PPNet rule (push_ProtoPNet.py:198-235): per prototype j, argmin over the flattened (n_c,h,w) of the
 * distance map restricted to images of class(j) (all images when class_specific == 0); accepted on strict '<'
 * (the first batch wins ties); the winning 1x1 patch of the add-on output is copied.
 *   dist : fp32 [B][P][S];  z : dtype [B][S][Dp] channels-last add-on output
 *   best_dist fp32 [P] (+inf), best_index int64 [P][2] = (global image index, s) (-1), best_patch fp32 [P][D]
 */
int pasn_push_ppnet_update(const float* dist, const void* z, const int64_t* labels, const int32_t* proto_class,
                           int class_specific, float* best_dist, int64_t* best_index, float* best_patch, int B, int P, int S,
                           int D, int Dp, int dtype, int64_t index_base, void* stream);

/* =====================================================================================================================
 * Training path (train-mode forward with batch statistics + backward).  The reference trains through the same modules
 * (`loss.backward()` through forward / compute_occurence_map: Video_XProtoNet_e2e.py:118-141, XProtoNet_e2e.py); these
 * entry points are what an autograd.Function under that nn.Module surface calls.  A conv + norm + activation "unit" is
 *     y = conv(x)            (pasn_conv3d_fwd / pasn_dwconv3d_fwd / pasn_first_conv_fwd with scale = 1, bias = 0, no activation)
 *     stat = batch statistics of y                                                        (pasn_bn_stats_fwd)
 *     a = act((y * sc + sh + residual) * gate)                                            (pasn_affine_act_fwd)
 * and its backward is pasn_unit_bwd_reduce (+ pasn_bn_bwd_apply), the conv's dgrad (the forward conv kernels with the
 * transposed weight, pasn_scatter_strided for strided 1x1x1 convs, pasn_dwconv3d_dgrad) and its wgrad
 * (pasn_conv3d_wgrad / pasn_first_conv_wgrad / pasn_dwconv3d_wgrad).  Tensors are channels-last rows [N][S][Cp] in `dtype`;
 * statistics, reductions and parameter gradients are fp32; every reduction except the split-K of pasn_conv3d_wgrad /
 * pasn_first_conv_wgrad (fp32 atomics) has a fixed summation order.
 * ===================================================================================================================== */

/* Row chunks per clip used by the reduction passes below: their `ws` is fp32 [N][chunks][2][Cp]. */
int pasn_train_chunks(int N, int S, int Cp);

/* torch.nn.BatchNorm{2,3}d, training=True (resnet_features.py:140,180; every norm layer of the trunks): per-channel batch
 * mean / biased variance of y, running-estimate update (unbiased variance, `momentum`), and the affine folded to
 *   stat : fp32 [4][Cp] = (mean, invstd, sc = gamma*invstd, sh = beta - mean*sc)        (channels >= C: zeros)
 *   pool_u : fp32 [N][Cp] or NULL -- per-clip mean over positions of y*sc + sh (the squeeze-excite pool)
 *   gamma / beta / running_* : fp32 [C] or NULL */
int pasn_bn_stats_fwd(const void* y, float* ws, const float* gamma, const float* beta, float* running_mean, float* running_var,
                      float momentum, float eps, float* stat, float* pool_u, int N, int S, int C, int Cp, int dtype, void* stream);

/* Squeeze-excite unit backward in ONE pass over (d, y) -- the analytic form of modes 1 + 2:
 *   pasn_unit_bwd_reduce(mode 4, ...): d <- d' = d * act'((y*sc + sh) * gate); ws fp32 [N][chunks][3][Cp] = per-clip partials of
 *       (sum d', sum d' yhat, sum yhat)            (gate required; coef / dgamma / dbeta unused)
 *   pasn_se_gate_bwd_stat: the gate's gradient sum d' u = gamma sum d' yhat + beta sum d' per clip, the two FCs' backward (add[n][c],
 *       parameter gradients, as pasn_se_gate_bwd) and, because d'' = d' gate + add is affine in d' per clip, the norm's
 *       coef = (sum d'' / R, sum d'' yhat / R), dgamma, dbeta WITHOUT a second pass over the tensor (ws is updated in place)
 *   pasn_bn_bwd_apply_se: dy = sc (d' gate + add - m1 - yhat m2), d'' formed on the fly (dy may alias d).
 * Same arithmetic as modes 1 + 2 + pasn_bn_bwd_apply up to fp32 summation order; fixed order. */
int pasn_se_gate_bwd_stat(float* ws3, const float* pool_u, const float* stat, const float* gate, const float* fc1_w, const float* fc1_b,
                          const float* fc2_w, const float* fc2_b, float* add, float* pn, float* dfc1_w, float* dfc1_b, float* dfc2_w,
                          float* dfc2_b, float* coef, float* dgamma, float* dbeta, int N, int S, int C, int Cp, int Cse, void* stream);
int pasn_bn_bwd_apply_se(const void* d, const void* y, const float* stat, const float* coef, const float* gate, const float* add, void* dy,
                         int N, int S, int C, int Cp, int dtype, void* stream);

/* Depthwise conv (raw output, as pasn_dwconv3d_fwd with pool_partial = NULL) AND the batch statistics of its output in one pass over
 * y: replaces pasn_dwconv3d_fwd + pasn_bn_stats_fwd for the X3D conv_b units (Video_XProtoNet trunks: every block's 3x3x3 depthwise conv
 * is followed by a BatchNorm3d).  `ws` is fp32 [N][rows][2][Cp] with rows = pasn_dwconv3d_stats_rows(d, dtype); rows = 0: the layer is
 * not covered (caller issues the two separate calls).  Statistics are taken from the fp32 outputs before they are rounded to `dtype`;
 * fixed summation order.  scale / bias: per-channel epilogue constants of the stencil (1 and 0 for a BatchNorm'd unit). */
int pasn_dwconv3d_stats_rows(const pasn_conv_desc* d, int dtype);
int pasn_dwconv3d_stats_fwd(const void* x, const float* w, const float* scale, const float* bias, void* y, float* ws, const float* gamma,
                            const float* beta, float* running_mean, float* running_var, float momentum, float eps, float* stat, float* pool_u,
                            const pasn_conv_desc* d, int dtype, void* stream);

/* dx of a stride-1 "same" depthwise conv (dy correlated with the reversed taps `w_flipped`, the forward stencil) AND, from the same fp32
 * outputs, the backward sums of the unit that produced the conv's input x = act_prev(y_prev * sc + sh): coef = (sum d' / R, sum d' yhat / R),
 * dgamma, dbeta with d' = dx * act_prev'(.) -- what pasn_unit_bwd_reduce(mode 3) would compute from (dx, y_prev) in a second pass.  The
 * caller then runs pasn_bn_bwd_apply(dx, y_prev, stat_prev, coef, ..., act_prev) as after mode 3.  Only when dx is that unit's WHOLE output
 * gradient (one consumer).  `ws`: fp32 [N][rows][2][Cp], rows = pasn_dwconv3d_dgrad_reduce_rows(d, dtype) (0: not covered). */
int pasn_dwconv3d_dgrad_reduce_rows(const pasn_conv_desc* d, int dtype);
int pasn_dwconv3d_dgrad_reduce(const void* dy, const float* w_flipped, const float* scale, const float* bias, void* dx, const void* y_prev,
                               const float* stat_prev, int act_prev, float* ws, float* coef, float* dgamma, float* dbeta,
                               const pasn_conv_desc* d, int dtype, void* stream);

/* a = act((y*sc + sh + residual) * gate[n][c]);  residual (dtype [N][S][Cp]) and gate (fp32 [N][Cp]) may be NULL.
 * A unit without a norm layer passes stat = (0, 1, 1, bias). */
int pasn_affine_act_fwd(const void* y, const float* stat, const void* residual, const float* gate, void* a, int N, int S, int C, int Cp,
                        int act, int dtype, void* stream);

/* One backward pass over a unit, IN PLACE on d (the gradient w.r.t. the unit's output a):
 *   mode 0:  d <- d * act'(y*sc + sh + residual);   coef = (sum d / R, sum d*yhat / R), dgamma = sum d*yhat, dbeta = sum d
 *   mode 1:  d <- d * act'((y*sc + sh) * gate);     ws partials of sum_s d*(y*sc + sh) per clip (gradient of the gate)
 *   mode 2:  d <- d * gate + add[n][c];             coef / dgamma / dbeta as mode 0
 *   mode 3:  the sums of mode 0 WITHOUT writing d back (no residual branch needs it): pasn_bn_bwd_apply is then called with
 *            the unit's `act` and differentiates on the fly -- one tensor write less per unit
 *   mode 4:  mode 1 with the per-clip sums of the analytic squeeze-excite backward (pasn_se_gate_bwd_stat below); ws is [N][chunks][3][Cp]
 * (yhat = (y - mean) * invstd, R = N*S; after mode 0 the buffer d is also the gradient of `residual`.) */
int pasn_unit_bwd_reduce(int mode, void* d, const void* y, const float* stat, const void* residual, const float* gate, const float* add,
                         float* ws, float* coef, float* dgamma, float* dbeta, int N, int S, int C, int Cp, int act, int dtype, void* stream);

/* Batch-norm input gradient: dy = sc * (d' - coef[0] - yhat * coef[1]) with d' = d (act = PASN_ACT_NONE: d was differentiated in
 * place by mode 0 / 2) or d' = d * act'(y*sc + sh) (after mode 3);  dy may alias d. */
int pasn_bn_bwd_apply(const void* d, const void* y, const float* stat, const float* coef, void* dy, int N, int S, int C, int Cp, int act,
                      int dtype, void* stream);

/* Statistics GROUPS (round 4).  The `_g` forms of the entry points above take `groups`: the batch is `groups` runs of N / groups consecutive
 * clips, each normalised with its OWN batch statistics -- stat is [groups][4][Cp], coef [groups][2][Cp]; the running estimates are updated
 * group by group in order (num_batches_tracked is the caller's: + groups); dgamma / dbeta are the parameter's (summed over the groups).
 * groups = 2 runs the two trunk passes of the reference's loss recipe (model(x), then model.compute_occurence_map(warp(x)): loss.py:302)
 * as ONE pass over [clips, warped clips] with exactly the statistics two passes would have used.  groups = 1 == the plain entry point.
 * 1 <= groups <= 4, N % groups == 0. */
int pasn_bn_stats_fwd_g(const void* y, float* ws, const float* gamma, const float* beta, float* running_mean, float* running_var,
                        float momentum, float eps, float* stat, float* pool_u, int N, int S, int C, int Cp, int dtype, int groups, void* stream);
int pasn_dwconv3d_stats_fwd_g(const void* x, const float* w, const float* scale, const float* bias, void* y, float* ws, const float* gamma,
                              const float* beta, float* running_mean, float* running_var, float momentum, float eps, float* stat, float* pool_u,
                              const pasn_conv_desc* d, int dtype, int groups, void* stream);
int pasn_affine_act_fwd_g(const void* y, const float* stat, const void* residual, const float* gate, void* a, int N, int S, int C, int Cp,
                          int act, int dtype, int groups, void* stream);
int pasn_unit_bwd_reduce_g(int mode, void* d, const void* y, const float* stat, const void* residual, const float* gate, const float* add,
                           float* ws, float* coef, float* dgamma, float* dbeta, int N, int S, int C, int Cp, int act, int dtype, int groups,
                           void* stream);
int pasn_bn_bwd_apply_g(const void* d, const void* y, const float* stat, const float* coef, void* dy, int N, int S, int C, int Cp, int act,
                        int dtype, int groups, void* stream);
int pasn_se_gate_bwd_stat_g(float* ws3, const float* pool_u, const float* stat, const float* gate, const float* fc1_w, const float* fc1_b,
                            const float* fc2_w, const float* fc2_b, float* add, float* pn, float* dfc1_w, float* dfc1_b, float* dfc2_w,
                            float* dfc2_b, float* coef, float* dgamma, float* dbeta, int N, int S, int C, int Cp, int Cse, int groups,
                            void* stream);
int pasn_bn_bwd_apply_se_g(const void* d, const void* y, const float* stat, const float* coef, const float* gate, const float* add, void* dy,
                           int N, int S, int C, int Cp, int dtype, int groups, void* stream);

/* Squeeze-excite backward: from the mode-1 partials `ws` and the pooled input `pool_u`, through sigmoid / fc2 / ReLU / fc1:
 *   add : fp32 [N][Cp] = dpool / S (the term mode 2 adds);  dw1 [Cse][C], db1 [Cse], dw2 [C][Cse], db2 [C]
 *   pn  : workspace of pasn_se_bwd_workspace_floats(N, C, Cse) floats */
size_t pasn_se_bwd_workspace_floats(int N, int C, int Cse);
int pasn_se_gate_bwd(const float* ws, const float* pool_u, const float* w1, const float* b1, const float* w2, const float* b2, float* add,
                     float* pn, float* dw1, float* db1, float* dw2, float* db2, int N, int S, int C, int Cp, int Cse, void* stream);

/* dst[n][t*st][h*sh][w*sw][:] (+)= src[n][t][h][w][:] with src [N][To][Ho][Wo][Cin_p], dst [N][Ti][Hi][Wi][Cin_p]; without
 * `accumulate` every other element of dst is zeroed (the input gradient of a strided 1x1x1 conv; zero insertion). */
int pasn_scatter_strided(const void* src, void* dst, const pasn_conv_desc* d, int accumulate, int dtype, void* stream);
int pasn_add_inplace(void* a, const void* b, size_t elements, int dtype, void* stream);
/* Max-pool backward (resnet_features.py:206 in training): dx receives dy of every window whose first maximum (scan order t, h, w:
 * the index torch records) the element is; x is the pooling INPUT, d the forward descriptor. */
int pasn_maxpool3d_bwd(const void* x, const void* dy, void* dx, const pasn_conv_desc* d, int dtype, void* stream);

/* Weight gradients.  dw is fp32 in the PARAMETER layout and must be zeroed by the caller for the two MFMA kernels:
 *   pasn_conv3d_wgrad      dw [Cout][Cin][kt*kh*kw]  += sum_rows dy[row][co] * x[in(row, tap)][ci]
 *   pasn_first_conv_wgrad  dw [Cout][3][kh*kw]       (x planar in_dtype [N][3][T][Hi][Wi])
 *   pasn_dwconv3d_wgrad    dw [C][kt*kh*kw]          (kh*kw <= 9; ws of pasn_dwconv3d_wgrad_workspace_floats(d) floats)
 * pasn_dwconv3d_dgrad: dx[n,ti,hi,wi,c] = sum_taps dy[n,to,ho,wo,c] * w[tap][c], w fp32 [taps][Cp] as for pasn_dwconv3d_fwd. */
int pasn_conv3d_wgrad(const void* x, const void* dy, float* dw, const pasn_conv_desc* d, int dtype, void* stream);
/* The same with a workspace of pasn_conv3d_wgrad_workspace_bytes(d, dtype) bytes (0 = this layer has no workspace path: ws may be NULL and
 * the call is pasn_conv3d_wgrad).  Non-zero for the stride-1 "same" (1,3,3) / (3,1,1) convs in bf16 (R(2+1)D-18, ResNet-18): the
 * gradient is then accumulated per row partition into ws and the partitions are summed in index order -- deterministic, no atomics. */
size_t pasn_conv3d_wgrad_workspace_bytes(const pasn_conv_desc* d, int dtype);
int pasn_conv3d_wgrad_ws(const void* x, const void* dy, float* dw, const pasn_conv_desc* d, int dtype, void* ws, void* stream);
/* ws: NULL, or pasn_first_conv_wgrad_workspace_bytes(d, dtype) bytes (non-zero for bf16): the clip's windows are then gathered once
 * into im2col rows and the gradient runs on the LDS-transposed bf16 MFMA kernel instead of the per-element gather. */
size_t pasn_first_conv_wgrad_workspace_bytes(const pasn_conv_desc* d, int dtype);
int pasn_first_conv_wgrad(const void* x, const void* dy, float* dw, const pasn_conv_desc* d, int in_dtype, int dtype, void* ws, void* stream);
size_t pasn_dwconv3d_wgrad_workspace_floats(const pasn_conv_desc* d);
int pasn_dwconv3d_wgrad(const void* x, const void* dy, float* ws, float* dw, const pasn_conv_desc* d, int dtype, void* stream);
int pasn_dwconv3d_dgrad(const void* dy, const float* w, void* dx, const pasn_conv_desc* d, int dtype, void* stream);

/* Head B tail in training (Video_XProtoNet.py:82-98): z = add-on output [N][S][Dp], r = occurrence-module output BEFORE the
 * abs [N][S][Pp] (both dtype) -> occ fp32 [N][P][S], feat fp32 [N][P][D], sim fp32 [N][P], logits fp32 [N][K].
 * Backward: dlogits [N][K], optional dsim [N][P] and docc [N][P][S] (NULL = none) -> dz, dr (dtype), dprotos [P][D],
 * dfc_w [K][P]; dfeat fp32 [N][P][D] is scratch.
 * z == NULL selects the occurrence-map-only mode of compute_occurence_map (Video_XProtoNet.py:100-109): forward writes occ
 * alone, backward is dr = sign(r) * docc (every other pointer but r / occ / docc / dr may be NULL). */
int pasn_xproto_tail_fwd(const void* z, const void* r, const float* protos, const float* fc_w, float* occ, float* feat, float* sim,
                         float* logits, const pasn_xproto_desc* d, int dtype, void* stream);
/* pasn_xproto_tail_fwd with a workspace of pasn_xproto_tail_workspace_bytes(d) bytes (0: no workspace path, ws may be NULL): the pooling
 * then runs on the matrix cores split over S, as in pasn_xproto_head_fwd. */
size_t pasn_xproto_tail_workspace_bytes(const pasn_xproto_desc* d);
int pasn_xproto_tail_fwd_ws(const void* z, const void* r, const float* protos, const float* fc_w, float* occ, float* feat, float* sim,
                            float* logits, const pasn_xproto_desc* d, int dtype, void* ws, void* stream);
int pasn_xproto_tail_bwd(const void* z, const void* r, const float* protos, const float* fc_w, const float* feat, const float* sim,
                         const float* dlogits, const float* dsim, const float* docc, float* dfeat, void* dz, void* dr, float* dprotos,
                         float* dfc_w, const pasn_xproto_desc* d, int dtype, void* stream);

/* Head A (ProtoPNet) backward of pasn_l2_head_fwd (ProtoPNet.py:189-243 under autograd): only the arg-min position of each
 * (image, prototype) carries gradient.  dlogits [N][K], dmin [N][P] or NULL (the cluster / separation costs act on
 * min_distances) -> dz dtype [N][S][Dp] (fully written), dprotos [P][D], dfc_w [K][P]; coef fp32 [N][P] is scratch. */
int pasn_l2_head_bwd(const void* z, const float* protos, const float* fc_w, const float* min_dist, const int32_t* argmin,
                     const float* dlogits, const float* dmin, void* dz, float* coef, float* dprotos, float* dfc_w, int N, int S, int D,
                     int Dp, int P, int K, int dtype, int activation, float eps, void* stream);

/* The affine warp of TransformLoss (loss.py:257-320): torchvision.transforms.functional.affine with
 * translate = 0, shear = 0, bilinear, fill = 0, applied to every H x W plane of a planar tensor ((N,3,T,H,W) clips: N*3*T planes;
 * (N,P,T',H',W') occurrence maps: N*P*T' planes).  bwd is the adjoint on fp32 (dx zeroed by the caller; fp32 atomics). */
int pasn_affine_warp_fwd(const void* x, void* y, long planes, int H, int W, float angle_deg, float scale, int dtype, void* stream);
int pasn_affine_warp_bwd(const float* dy, float* dx, long planes, int H, int W, float angle_deg, float scale, void* stream);

/*
 * Training: all conv weights of a step packed from the live fp32 parameters into the layouts the forward kernels read, in ONE launch
 * (replaces the per-parameter torch expressions of the host side: the reference has no counterpart -- cuDNN reads the parameters as they
 * are, model/XProtoNet.py keeps nn.Conv modules).  `jobs`, `block_job`, `block_chunk` are DEVICE arrays: block b packs destination
 * elements [block_chunk[b], block_chunk[b] + 1) x pasn_pack_chunk() of job block_job[b]; every destination element is written.
 *   mode 0: dense forward   dst[row = co][tap][k = ci]  = src[co][ci][tap]              (src [cout][cin][taps], dst [rows][taps][kc])
 *   mode 1: dense dX        dst[row = ci][tap][k = co]  = src[co][ci][taps - 1 - tap]   (transposed, taps reversed)
 *   mode 2 / 3: depthwise   dst[tap][c] = src[c][tap] / src[c][taps - 1 - tap]          (fp32 dst [taps][kc = Cp])
 *   frag = 1 (taps == 1): dst is fragment-major [rows/32][kc/kstep][2][32][ch] (the x-tile pointwise kernels, w_frag = 1)
 */
typedef struct pasn_pack_job {
    const float* src;
    void* dst;
    long n;  /* destination elements */
    int mode, cout, cin, taps, rows, kc, frag, bf16, kstep, ch;
} pasn_pack_job;
int pasn_pack_chunk(void);
int pasn_pack_weights(const pasn_pack_job* jobs, const int* block_job, const int* block_chunk, int nblocks, void* stream);

/*
 * Data-parallel gradient exchange on RCCL (xGMI), without torch.distributed in the data path: ONE in-place sum all-reduce of the
 * flat fp32 gradient bucket per optimizer step.  The reference trains on one GPU and has no collective
 * (SURVEY section 0); these four calls are what a trainer needs around protoasnet_amd/dp.py.  librccl.so is resolved at run time.
 *   pasn_comm_unique_id : rank 0 fills PASN_COMM_ID_BYTES bytes; the caller ships them to the other ranks (any side channel)
 *   pasn_comm_init      : every rank, with its HIP device current; *comm_out is an opaque handle
 *   pasn_allreduce      : buf[count] (dtype PASN_F32 / PASN_BF16) summed over the ranks in place, asynchronous on `stream`
 */
#define PASN_COMM_ID_BYTES 128
int pasn_comm_unique_id(void* id_out);
int pasn_comm_init(const void* id, int world_size, int rank, void** comm_out);
int pasn_allreduce(void* comm, void* buf, size_t count, int dtype, void* stream);
int pasn_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* PROTOASNET_AMD_H */
