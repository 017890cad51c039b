/* mgdt.h - C ABI of the MI355X-native MGDT-YOLO detection hot path (libmgdt_hip.so).
 *
 * Every entry point: plain pointers/sizes, no torch types, asynchronous on the caller's hipStream_t,
 * never allocates or frees device memory (the caller owns inputs, outputs, packed weights and
 * workspaces), returns 0 on success or a negative mgdt_status; mgdt_last_error() gives the text
 * (thread-local).  No host synchronisation inside, except where noted.
 *
 * Activations are addressed through `mgdt_view`: an NHWC-ordered 4-d view with explicit element
 * strides, so channel slices of a wider tensor (the reference's chunk()/split()/cat() on dim 1) and
 * torch channels_last tensors are passed without copies.
 *
 * Each function cites the reference interface (paths relative to the reference repo root) it replaces.
 */
#ifndef MGDT_H
#define MGDT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mgdt_stream; /* hipStream_t */

typedef enum { MGDT_OK = 0, MGDT_BAD_SHAPE = -1, MGDT_BAD_DTYPE = -2, MGDT_LAUNCH_FAIL = -3, MGDT_BAD_ARG = -4,
               MGDT_WORKSPACE = -5 } mgdt_status;
typedef enum { MGDT_F32 = 0, MGDT_BF16 = 1, MGDT_U8 = 2 /* image input of mgdt_conv2d_direct_fwd only */ } mgdt_dtype;
typedef enum { MGDT_ACT_NONE = 0, MGDT_ACT_SILU = 1, MGDT_ACT_RELU = 2, MGDT_ACT_GELU = 3 } mgdt_act;

/* 4-d activation view; sizes in elements, strides in elements of `dtype`. p may be NULL for "absent". */
typedef struct {
  void* p;
  int32_t n, h, w, c;
  int64_t sn, sh, sw, sc;
} mgdt_view;

const char* mgdt_last_error(void);
const char* mgdt_version(void);

/* ---- weight packing: nn.Conv2d weight (OIHW fp32) [+ BatchNorm2d fold] -> MFMA fragment order ----------
 * Replaces yolo/utils/torch_utils.py:114-135 (fuse_conv_and_bn) + the implicit weight layout of ATen conv2d.
 * bn_* may all be NULL (no BN; conv_bias optional).  With BN: W' = diag(g/sqrt(var+eps)) W,
 * b' = beta - g*mean/sqrt(var+eps) (+ scaled conv bias).  Outputs: packed weights (size from
 * mgdt_conv_packed_bytes) and bias_out[cout_pad] fp32 (cout rounded up to 16).                          */
size_t mgdt_conv_packed_bytes(int cin, int cout, int k, int dtype);
int mgdt_conv_pack(const float* w_oihw, const float* conv_bias, const float* bn_gamma, const float* bn_beta,
                   const float* bn_mean, const float* bn_var, float bn_eps, int cin, int cout, int k, int dtype,
                   void* packed_out, float* bias_out, mgdt_stream s);
/* Weights of the stride-1 data-gradient convolution: dx = mgdt_conv2d_fwd(x = dy, packed, k, stride 1, y = dx) with
 * w'[ci][co][ky][kx] = w[co][ci][k-1-ky][k-1-kx] (what autograd's conv backward computes for stride 1, same padding).  cin / cout are
 * those of the original conv; size the buffer with mgdt_conv_packed_bytes(cout, cin, k, dtype); bias_out: fp32[cin rounded up to 16], zeroed. */
int mgdt_conv_pack_dgrad(const float* w_oihw, int cin, int cout, int k, int phase, int dtype, void* packed_out, float* bias_out, mgdt_stream s);
/* Many packs in a few launches (a training step re-packs every convolution after the optimizer moved the weights; the reference has no counterpart:
 * ATen convolutions read nn.Conv2d.weight in place).  One descriptor = the arguments of one mgdt_conv_pack (mode 0) or mgdt_conv_pack_dgrad
 * (mode 1 = stride-1 data gradient, mode 2 + phase = stride-2 phase) call; cin / cout are those of the ORIGINAL convolution. */
typedef struct mgdt_pack_desc {
  const float* w; const float* conv_bias; const float* bn_gamma; const float* bn_beta; const float* bn_mean; const float* bn_var;
  float bn_eps; int32_t cin, cout, k, dtype, mode;
  void* packed; float* bias_out;
} mgdt_pack_desc;
int mgdt_conv_pack_batch(const mgdt_pack_desc* descs, int n, mgdt_stream s);
/* phase = -1: stride 1 (above).  phase = 2*py + px in 0..3 (k = 3, stride 2, even input size): the data gradient at input pixels
 * (2a + py, 2b + px) is a 3x3 same-padding convolution over dy with the taps w[py + 1 - 2*dy'][px + 1 - 2*dx'] that exist:
 * dx[:, py::2, px::2] = mgdt_conv2d_fwd(x = dy, packed(phase), k = 3, stride 1, y = that strided view). */

/* ---- fused convolution (implicit GEMM on MFMA) ------------------------------------------------------------
 * Replaces nn/modules/conv.py:25-42 Conv.forward/forward_fuse (conv2d + folded BN + act), the Bottleneck
 * shortcut add (nn/modules/block.py:514-526), the MSPA pre-add `sp + spx[i]` (block.py:252-254), the
 * channel-slice reads/writes that stand in for chunk()/cat() (block.py:203-207,250-266), nn.Linear on NHWC
 * tokens (convnextv2.py:59,62) and GRN's per-(image,channel) affine on the input (utils.py:171-182).
 *   y = act( conv(k, stride, pad=k/2)( (x [+ x2]) [* in_scale[n,c] + in_shift[c]] ) + bias ) [+ r1] [+ r2]
 * groups == 1, k in {1,3}, stride in {1,2}, cin % (4 fp32 | 8 bf16) == 0, all views sc == 1.               */
int mgdt_conv2d_fwd(const mgdt_view* x, const mgdt_view* x2, const float* in_scale, const float* in_shift,
                    const void* packed_w, const float* bias, int k, int stride, int act, const mgdt_view* r1,
                    const mgdt_view* r2, const mgdt_view* y, int dtype, mgdt_stream s);
/* ---- fp8 (OCP e4m3fn) variant of the fused convolution - BASELINE configs[4] ("fp8 inference: CDNA4 fp8-MFMA implicit-GEMM convs"; the
 * reference has no fp8 path: yolo/engine/trainer.py:223 is fp16 autocast only, so the numerics are checked against an e4m3 emulation of
 * nn/modules/conv.py:25-42 and, end to end, against the fp32 fixtures with a stated tolerance).
 * W8A8, fp32 accumulation: activations stay bf16 in HBM and are converted in registers, x_q = e4m3(clamp(x * x_qscale, +-448)) (per-tensor
 * x_qscale, calibrated by the caller); weights are packed once as e4m3(w' / w_scale[co]) with w' the BN-folded weight and
 * w_scale[co] = max|w'[co]| / 448; y = act(acc * oscale[co] + bias[co]) [+ r1] [+ r2], oscale[co] = w_scale[co] / x_qscale.
 * mgdt_conv_pack_fp8 writes the panel (mgdt_conv_packed_bytes_fp8), bias_out[cout_pad] and oscale_out[cout_pad] (cout rounded up to 16);
 * mgdt_conv2d_fp8_fwd takes the same views / fused extras as mgdt_conv2d_fwd with dtype MGDT_BF16.  cin % 8 == 0, cout % 4 == 0, k in {1, 3}. */
size_t mgdt_conv_packed_bytes_fp8(int cin, int cout, int k);
int mgdt_conv_pack_fp8(const float* w_oihw, const float* conv_bias, const float* bn_gamma, const float* bn_beta, const float* bn_mean,
                       const float* bn_var, float bn_eps, int cin, int cout, int k, float x_qscale, void* packed_out, float* bias_out,
                       float* oscale_out, mgdt_stream s);
int mgdt_conv2d_fp8_fwd(const mgdt_view* x, const mgdt_view* x2, const float* in_scale, const float* in_shift, const void* packed_w,
                        const float* bias, const float* oscale, float x_qscale, int k, int stride, int act, const mgdt_view* r1,
                        const mgdt_view* r2, const mgdt_view* y, mgdt_stream s);
/* One phase (py, px) = phase >> 1, phase & 1 of the data gradient of a stride-2 3x3 convolution: dx[:, 2i+py, 2j+px] as a stride-1 convolution of dy
 * whose K holds only the taps that phase uses (packed by mgdt_conv_pack_dgrad(phase)); replaces the four 9-tap convolutions over mostly-zero weights.
 * Reference: the autograd of the stride-2 nn.Conv2d layers (backbone rows 0, 1, 3, 5, 7 of models/v8/*.yaml). */
int mgdt_conv2d_phase_fwd(const mgdt_view* x, const void* packed_w, const float* bias, int phase, const mgdt_view* r1, const mgdt_view* r2,
                          const mgdt_view* y, int dtype, mgdt_stream s);

/* ---- ConvNeXtV2 block MLP, hidden map kept on chip (nn/modules/convnextv2.py:62-77: pwconv1 -> GELU -> GRN -> pwconv2 -> + input).
 * t = LayerNorm output, res = the block input, y = res + pwconv2(GRN(gelu(pwconv1(t)))).  Two launches: GRN statistics, then
 * apply (both compute pwconv1 + GELU).  w1 [4c][c], w2 [c][4c] row-major as nn.Linear.weight; gamma/beta [4c] (GRN).
 * Covered: bf16 with c in {32, 64, 96} (both weight panels in LDS); mgdt_cnx_mlp_packed_bytes returns 0 otherwise and the caller
 * keeps the mgdt_conv2d_fwd / mgdt_grn_stats_fwd chain. */
size_t mgdt_cnx_mlp_packed_bytes(int c, int dtype);
int mgdt_cnx_mlp_pack(const float* w1, const float* b1, const float* w2, const float* b2, int c, void* packed, int dtype, mgdt_stream s);
size_t mgdt_cnx_mlp_workspace_bytes(int n, int h, int w, int c);
int mgdt_cnx_mlp_fwd(const mgdt_view* t, const mgdt_view* res, const void* packed, const float* gamma, const float* beta, void* ws,
                     const mgdt_view* y, int dtype, mgdt_stream s);

/* ---- a whole ConvNeXtV2 block in ONE launch, bf16 inference (nn/modules/convnextv2.py:48-77): dw 7x7 + bias + LayerNorm, pwconv1 + GELU with the
 * 4c-wide hidden tile kept in registers, GRN (the workgroups of an image meet at a per-image barrier for sum_hw h^2), pwconv2 + residual.
 * x, y: N x H x W x c NHWC views (y may not alias x: neighbouring tiles read x's halo), dw_w49c [49][c] fp32 (tap-major), packed: the blob of
 * mgdt_cnx_mlp_pack.  ws (mgdt_cnx_block_workspace_bytes, 16-byte aligned): its first 4096 bytes are the per-image barrier words (arrival
 * counter + generation) that MUST BE ZERO at the first call and belong to this function from then on (the counters are back at zero when a
 * call ends, so hipGraph replays and calls with other shapes need no reset; at most 512 images per call).  mgdt_cnx_block_supported: c in {32, 64, 96}, bf16, a tile decomposition with at most as many tiles per image as the chip has
 * compute units (every workgroup of a launch is resident at once; larger batches run as several launches of whole images).
 * tail_w / tail_b / tail_act (NULL / NULL / 0 for none): a 1x1 Conv + BN + act applied to the block's output inside the launch - the IFM's closing
 * Conv (nn/modules/block.py:336-338) - packed with mgdt_conv_pack(c, cout, 1, bf16) over input channels in the accumulator order of
 * mgdt_conv1x1_inject_conv_fwd's note (j < c / 32); y then has cout <= c channels and the block's own output is never stored. */
int mgdt_cnx_block_supported(int n, int h, int w, int c, int dtype);
size_t mgdt_cnx_block_workspace_bytes(int n, int h, int w, int c);
int mgdt_cnx_block_fwd(const mgdt_view* x, const float* dw_w49c, const float* dw_b, const float* ln_w, const float* ln_b, float eps, const void* packed,
                       const float* gamma, const float* beta, const void* tail_w, const float* tail_b, int tail_act, void* ws, size_t ws_bytes,
                       const mgdt_view* y, int dtype, mgdt_stream s);

/* ---- layers 0 and 1 of every YOLOv8 graph (Conv 3->16 k3 s2, Conv 16->32 k3 s2, both + BN + SiLU; models/v8/*.yaml rows 0-1) in one
 * launch, bf16 path: the image patch is staged in LDS with 16-byte row loads, layer 0 runs on MFMA out of LDS, its map never leaves the
 * CU.  x: N x 3 x H x W NCHW image (any strides), x_dtype MGDT_BF16 / MGDT_F32 / MGDT_U8 (u8 is divided by 255 like
 * yolo/engine/predictor.py:129); y: N x H/4 x W/4 x 32 bf16 NHWC.  packed0 = mgdt_stem2_pack(W' of layer 0, fp32 [16][3][3][3] with the
 * BatchNorm scale folded in as fuse_conv_and_bn does), bias0[16]; packed1 / bias1 = mgdt_conv_pack(16, 32, 3, MGDT_BF16). */
size_t mgdt_stem2_packed_bytes(void);
int mgdt_stem2_pack(const float* w_folded, void* packed, mgdt_stream s);
int mgdt_stem2_fwd(const mgdt_view* x, int x_dtype, const void* packed0, const float* bias0, const void* packed1, const float* bias1,
                   const mgdt_view* y, mgdt_stream s);

/* ---- Detect head tail in one launch, bf16 (nn/modules/head.py:150-177): the two final 1x1 convs with bias (box c2 -> 16, cls c3 -> nc), the raw
 * (N, 16+nc, H, W) map and its decode (DFL expectation, dist2bbox, stride, sigmoid) into y[N][4+nc][a_total] at anchor offset a_off.
 * wb/bb, wc/bc: mgdt_conv_pack(c2, 16, 1, bf16) / mgdt_conv_pack(c3, nc, 1, bf16) with the conv biases.  reg_max must be 4 (this fork's
 * Detect); other heads keep mgdt_conv2d_fwd + mgdt_detect_decode_fwd.  best_keys: NULL, or [N][a_total] - per anchor the NMS key of its
 * best class (first maximal score, ops.py:225-226), see mgdt_nms_fwd.  wb3 / bb3: NULL, or mgdt_conv_pack(16, 16, 3, bf16) of the box branch's second
 * Conv (cv2[i][1], head.py:150: 3x3, BN, SiLU) - tb is then THAT conv's input and the conv runs inside this launch; wb must then be packed from the
 * final 1x1's weight zero-padded to 32 input channels (its 16 real ones in the accumulator order of mgdt_conv1x1_inject_conv_fwd's note). */
int mgdt_detect_tail_supported(int c2, int c3, int nc, int reg_max, int dtype);
int mgdt_detect_tail_fwd(const mgdt_view* tb, const mgdt_view* tc, const void* wb, const float* bb, const void* wc, const float* bc, int nc,
                         float stride, int a_off, int a_total, const mgdt_view* feat, float* y, unsigned long long* best_keys,
                         const void* wb3, const float* bb3, mgdt_stream s);

/* ---- a whole CSP block (MSPA_C2f / C2f) in one launch, bf16 inference (nn/modules/block.py:187-287, :514-526):
 *   mode 0 (MSPA_C2f): front = the three chained 1x1 convs of mgdt_pw_chain3_fwd (blob of mgdt_pw_chain_pack), bottleneck input
 *                      sp2 + x[3wd:4wd]; concat = [sp0 | sp1 | sp2 | b_0 .. b_{n-1}]
 *   mode 1 (C2f):      x = cv1's output (2wd channels, computed by mgdt_conv2d_fwd), front / front_bias unused (NULL); bottleneck
 *                      input = its second half; concat = [x | b_0 .. b_{n-1}]
 *   then n bottlenecks (two 3x3 convs wd -> wd each, mgdt_conv_pack panels mid[2n] in execution order, residual add when `shortcut`)
 *   and the 1x1 conv `back` over the concat -> y.  One workgroup computes one spatial tile of one image with the halo (2n pixels)
 *   recomputed; nothing but x is read from and nothing but y written to HBM.  All convs share `act`.
 *   pool: NULL, or fp32 [n][slots][cout]: channel sums of y per tile (slot = tile * (slots / tiles) + k, tiles in row-major order);
 *   slots = mgdt_csp_block_tiles(...), which also reports {TH, TW, RH, RW, lds bytes, workgroups, tiles_x, tiles_y} in geom8 when
 *   non-NULL; feed it to mgdt_spr_attn_scale_fwd(pool, slots, tiles_x, tiles_y, ...).  mode 0 needs even h, w and picks an even
 *   tile grid (every tile lies inside one adaptive_avg_pool2d(2) bin).
 *   mgdt_csp_block_supported: 1 when covered (bf16; wd in {8,16,32,64}; n in {1,2}; mode 0: cin == 4wd; mode 1: cin == 2wd,
 *   wd >= 16); otherwise callers keep the per-conv launches. */
int mgdt_csp_block_supported(int mode, int cin, int cout, int wd, int nbtl, int h, int w, int dtype);
int mgdt_csp_block_tiles(int mode, int n, int cin, int cout, int wd, int nbtl, int h, int w, int* geom8);
int mgdt_csp_block_fwd(int mode, const mgdt_view* x, const void* front, const float* front_bias, const void* const* mid,
                       const float* const* mid_bias, int nbtl, int shortcut, const void* back, const float* back_bias, int wd, int act,
                       const mgdt_view* y, float* pool, int dtype, mgdt_stream s);

/* ---- MSPA_C2f hierarchical point-wise front in one launch (nn/modules/block.py:250-259, scale = 4):
 *   sp0 = act(conv0(x[0:wd])), sp1 = act(conv1(sp0 + x[wd:2wd])), sp2 = act(conv2(sp1 + x[2wd:3wd])), y[i*wd:(i+1)*wd] = sp_i
 * conv_i = 1x1 conv (wd -> wd) with BN folded.  x, y: N x H x W x 3*wd views (slices of the block input / concat buffer).
 * Pack the three convs with idx = 0, 1, 2 into one blob.  Covered: bf16, wd % 4 == 0, wd <= 64 (packed_bytes returns 0 otherwise
 * and the caller keeps three mgdt_conv2d_fwd launches). */
size_t mgdt_pw_chain_packed_bytes(int wd, int dtype);
int mgdt_pw_chain_pack(int idx, const float* w_oi, const float* conv_bias, const float* bn_gamma, const float* bn_beta, const float* bn_mean,
                       const float* bn_var, float bn_eps, int wd, int dtype, void* packed, mgdt_stream s);
int mgdt_pw_chain3_fwd(const mgdt_view* x, const void* packed, int wd, int act, const mgdt_view* y, int dtype, mgdt_stream s);

/* ---- InjectionMultiSum_Auto_pool, up-sampling branch, in one launch (nn/modules/block.py:376-399):
 *   y = local_embedding(x) * bilinear(h_sigmoid(ga)) + bilinear(gf)      (align_corners=False, gate applied before interpolation)
 * packed_w / bias: the 1x1 local_embedding conv packed by mgdt_conv_pack (BN folded, no activation); ga, gf: equally laid out
 * N x Hg x Wg x cout maps with Hg <= H, Wg <= W.  mgdt_conv1x1_inject_supported tells whether the shapes are covered (bf16,
 * cin <= 128, cout in {128, 256}); otherwise the caller runs mgdt_conv2d_fwd + mgdt_inject_fwd. */
int mgdt_conv1x1_inject_supported(int cin, int cout, int h, int w, int hg, int wg, int dtype);
int mgdt_conv1x1_inject_fwd(const mgdt_view* x, const void* packed_w, const float* bias, const mgdt_view* ga, const mgdt_view* gf,
                            const mgdt_view* y, int dtype, mgdt_stream s);

/* ---- injection + the NEXT layer's 1x1 Conv+BN+act in one launch (models/v8/mspa_c2f_gd_yolov8.yaml head rows 4-5: the injection's 256-channel
 * output has one consumer, C2f.cv1, nn/modules/block.py:199-201): y2 = act2(conv1x1(injection(x, ga, gf)) + bias2), the 256-channel map never
 * leaves the chip.  packed_w2 / bias2: mgdt_conv_pack(256, cout2, 1, bf16) of the second conv with its INPUT channels permuted to the first
 * conv's accumulator order: packed input channel (j*4 + g)*8 + e <- channel (2*j + e/4)*16 + 4*g + e%4 (j < 8, g < 4, e < 8).
 * Global maps: either ga / gf (computed by the caller; gsrc NULL) or gsrc = their common 32-channel input (block.py:377-381: x_g.split(...)[flag])
 * with packed_wg / bias_g = mgdt_conv_pack(32, 2 * cmid, 1, bf16) of [global_act | global_embedding] (cmid = 256): both 1x1 convs are then
 * evaluated on each tile's source pixels inside this launch and the 2 x cmid-channel maps never reach HBM either (ga / gf NULL). */
int mgdt_conv1x1_inject_conv_supported(int cin, int cmid, int cout2, int h, int w, int hg, int wg, int dtype);
int mgdt_conv1x1_inject_conv_fwd(const mgdt_view* x, const void* packed_w, const float* bias, const mgdt_view* ga, const mgdt_view* gf,
                                 const mgdt_view* gsrc, const void* packed_wg, const float* bias_g, int cmid,
                                 const void* packed_w2, const float* bias2, int act2, const mgdt_view* y2, int dtype, mgdt_stream s);

/* ---- TOODHead (nn/modules/head.py:466-572; parity unpinned: mmcv's ModulatedDeformConv2d is not shipped with the reference) --------
 * GroupNorm + activation (Conv_GN head.py:67-81, DyDCNv2's norm block.py:427-431): y = act(group_norm(x, groups, gamma, beta, eps)). */
size_t mgdt_groupnorm_workspace_bytes(int n, int c);
int mgdt_groupnorm_fwd(const mgdt_view* x, const float* gamma, const float* beta, int groups, float eps, int act, void* ws, const mgdt_view* y,
                       int dtype, mgdt_stream s);
/* TaskDecomposition layer attention (head.py:107-123): sums[n][c] = sum_hw feat (mgdt_nc_reduce); w1 [hid][c], w2 [stacked][hid];
 * scale[n][k*feat + j] = sigmoid(w2 relu(w1 avg + b1) + b2)[k] - the per-(image, input channel) scale that turns the reduction
 * conv into mgdt_conv2d_fwd(in_scale = scale). */
int mgdt_tood_layer_attn_fwd(const float* sums, int n, int c, int hw, const float* w1, const float* b1, const float* w2, const float* b2, int hid,
                             int stacked, float* scale, mgdt_stream s);
/* DCNv2 3x3, stride 1, pad 1, one deform group (block.py:401-432, mmcv modulated_deform_conv): offset_mask = N x H x W x (>=27):
 * 18 offsets (dy, dx per kernel point) then 9 mask logits (sigmoid applied inside, head.py:525).  w_gemm [9*cin][cout] fp32
 * (mgdt_conv_pack_direct layout), bias fp32[cout] or NULL. */
int mgdt_dcnv2_fwd(const mgdt_view* x, const mgdt_view* offset_mask, const float* w_gemm, const float* bias, const mgdt_view* y, int dtype,
                   mgdt_stream s);
/* The same on the MFMA path (bf16, cin % 8 == 0, cout in {16,32,48,64}, no bias): packed_w = mgdt_conv_pack(w, no BN, cin, cout, k = 3, bf16). */
int mgdt_dcnv2_mfma_fwd(const mgdt_view* x, const mgdt_view* offset_mask, const void* packed_w, const mgdt_view* y, int dtype, mgdt_stream s);
/* y = x * sigmoid(gate[n,h,w]) (cls_feat * cls_prob, head.py:536); gate: N x H x W x 1 logits. */
int mgdt_pixel_gate_fwd(const mgdt_view* x, const mgdt_view* gate, const mgdt_view* y, int dtype, mgdt_stream s);

/* ---- direct convolution (any strides/groups/cin; used for the 3-channel stem and odd shapes) ------------
 * Same math as above without the fused extras; x may be fp32 NCHW (x_dtype) while y is `dtype` NHWC.
 * Stem (k=3, cin<=4, cout%16==0): x_dtype may also be MGDT_U8 - the uint8 image is divided by 255 on the fly exactly as the
 * reference's preprocess does (yolo/engine/predictor.py:129, yolo/v8/detect/val.py:34, train.py:64), in fp32.
 * w_gemm: [k*k*cin/groups][cout] fp32 from mgdt_conv_pack_direct; bias fp32[cout].                         */
int mgdt_conv_pack_direct(const float* w_oihw, const float* conv_bias, const float* bn_gamma, const float* bn_beta,
                          const float* bn_mean, const float* bn_var, float bn_eps, int cin_g, int cout, int k,
                          float* w_out, float* bias_out, mgdt_stream s);
int mgdt_conv2d_direct_fwd(const mgdt_view* x, int x_dtype, const float* w_gemm, const float* bias, int k, int stride,
                           int groups, int act, const mgdt_view* y, int dtype, mgdt_stream s);

/* ---- MSPA attention: SPRModule pooling + MLP + softmax over the 4 groups + scale ------------------------
 * nn/modules/spr_module.py:8-31, nn/modules/block.py:268-287.
 * pool: x (n,h,w,c) -> pooled fp32 [n][MGDT_SPR_SPLITS][c][5]: per row-band partial SUMS of {whole map, the four
 *       adaptive_avg_pool2d(2) bins (row-major)} (fixed reduction order: results are run-to-run identical);
 * attn: finishes the means, runs fc1/ReLU/fc2/sigmoid on each of the `groups` channel groups (shared weights,
 *       fc1_w [cw/4][5cw], fc2_w [cw][cw/4], cw = c/groups), softmax over the groups (softmax = 1; 0 = the bare sigmoid weights of
 *       SPRModule.forward, spr_module.py:20-31) -> attn fp32 [n][c];
 * scale: y = x * attn[n,c].                                                                               */
#define MGDT_SPR_SPLITS 16
int mgdt_spr_pool_fwd(const mgdt_view* x, float* pooled, int dtype, mgdt_stream s);
int mgdt_spr_attn_fwd(const float* pooled, const float* fc1_w, const float* fc1_b, const float* fc2_w,
                      const float* fc2_b, int n, int c, int groups, int h, int w, int softmax, float* attn, mgdt_stream s);
int mgdt_scale_channels_fwd(const mgdt_view* x, const float* attn, const mgdt_view* y, int dtype, mgdt_stream s);
/* spr_attn + scale_channels in one launch: every workgroup recomputes its image's attention (same order, same bits) and scales
 * a share of the pixels: y = x * softmax_groups(SPR(pooled)).  `pooled` as written by mgdt_spr_pool_fwd (nsplit = MGDT_SPR_SPLITS, or 0)
 * with tiles_x = tiles_y = 0: fp32 [n][nsplit][c][5]; or the per-tile sums of mgdt_csp_block_fwd: fp32 [n][nsplit][c], tiles_x x tiles_y
 * its tile grid.  pool_a / pool_b (NULL or NHWC views of the same n, c with h = H / F, w = W / F, F integer): F x F average pools of y, bit-equal
 * to mgdt_adaptive_avgpool_fwd(y) - the SimFusion_4in / SimFusion_3in inputs of the GD neck (nn/modules/block.py:289-329), written by the
 * pass that has the map in hand instead of by pooling launches of their own. */
int mgdt_spr_attn_scale_fwd(const float* pooled, int nsplit, int tiles_x, int tiles_y, const float* fc1_w, const float* fc1_b, const float* fc2_w, const float* fc2_b, int groups,
                            const mgdt_view* x, const mgdt_view* y, const mgdt_view* pool_a, const mgdt_view* pool_b, int dtype, mgdt_stream s);

/* ---- SPPF pooling: y1,y2,y3 = maxpool5(x), maxpool5(y1), maxpool5(y2) (nn/modules/block.py:138-153) ----- */
int mgdt_sppf_pool_fwd(const mgdt_view* x, const mgdt_view* y1, const mgdt_view* y2, const mgdt_view* y3, int dtype,
                       mgdt_stream s);

/* ---- resamplers (nn/modules/block.py:292,304,316,328,393-394; nn.Upsample in models/v8/yolov8.yaml) ---- */
int mgdt_adaptive_avgpool_fwd(const mgdt_view* x, const mgdt_view* y, int dtype, mgdt_stream s);
int mgdt_bilinear_fwd(const mgdt_view* x, const mgdt_view* y, int dtype, mgdt_stream s); /* align_corners=False */
int mgdt_nearest_fwd(const mgdt_view* x, const mgdt_view* y, int dtype, mgdt_stream s);
int mgdt_copy_fwd(const mgdt_view* x, int x_dtype, const mgdt_view* y, int y_dtype, mgdt_stream s);
/* The 1-3 channel image (any strides; MGDT_U8 is divided by 255: detect/train.py:64) as a 4-channel NHWC map whose remaining channels are zero - the input the
 * stem's weight-gradient kernel reads.  Writes all four channels of y. */
int mgdt_image_pad4_fwd(const mgdt_view* x, int x_dtype, const mgdt_view* y, int y_dtype, mgdt_stream s); /* cat / layout / cast */

/* ---- ConvNeXtV2 block pieces (nn/modules/convnextv2.py:48-77, nn/modules/utils.py:145-182) -------------
 * dwln: y = LayerNorm_c(dwconv7x7(x) + b) * ln_w + ln_b   (eps 1e-6), dw_w is [49][c] fp32.
 * grn_stats: t (n,h,w,c) -> scale[n][c] = gamma[c]*Nx[n,c] + 1, with Nx = ||t||_2(h,w) / (mean_c + 1e-6);
 *            ws: fp32 [n][MGDT_GRN_SPLITS][c] scratch (per pixel-band partial sums, fixed reduction order).
 *            (shift[c] = beta[c] is passed to mgdt_conv2d_fwd directly.)                                   */
#define MGDT_GRN_SPLITS 8
int mgdt_dwconv7_ln_fwd(const mgdt_view* x, const float* dw_w, const float* dw_b, const float* ln_w, const float* ln_b,
                        float eps, const mgdt_view* y, int dtype, mgdt_stream s);
int mgdt_grn_stats_fwd(const mgdt_view* t, const float* gamma, float* ws, float* scale, int dtype, mgdt_stream s);

/* ---- InjectionMultiSum_Auto_pool tail (nn/modules/block.py:381-399) --------------------------------------
 * up branch  (local >= global size): y = local * bilinear(relu6(ga+3)/6) + bilinear(gf)
 * pool branch (local <  global size): y = local * avgpool(ga) + avgpool(gf)   (no h_sigmoid, as written)  */
int mgdt_inject_fwd(const mgdt_view* local, const mgdt_view* ga, const mgdt_view* gf, const mgdt_view* y, int dtype,
                    mgdt_stream s);

/* ---- Detect eval tail (nn/modules/head.py:165-177; DFL block.py:36-54; tal.py:476-500) ------------------
 * feat: one level (n,h,w,4R+nc) raw head map; writes y[n][4+nc][a_total] at anchor offset a_off:
 * xywh = dist2bbox(softmax_R . arange(R), anchor(+0.5)) * stride, cls = sigmoid.  y is fp32.              */
int mgdt_detect_decode_fwd(const mgdt_view* feat, int reg_max, int nc, float stride, int a_off, int a_total, float* y,
                           int dtype, mgdt_stream s);

/* ---- batched NMS (yolo/utils/ops.py:136-266 incl. the torchvision.ops.nms call at :249) -----------------
 * pred fp32 [n][4+nc][a].  Outputs per image: out[n][max_det][6] fp32 rows (x1,y1,x2,y2,conf,cls),
 * kept_anchor[n][max_det] int32 (anchor index of each kept row), counts[n] int32.  classes: optional int32
 * list (NULL = all).  Ties in score: lower candidate index first.  ws from mgdt_nms_workspace_bytes.
 * best_keys: NULL, or [n][a] 64-bit keys ((0xFFFFFFFF - bits(best score)) << 32 | anchor * nc + first best class) of EVERY anchor as
 * mgdt_detect_tail_fwd writes them next to pred: the best-class scan over pred (ops.py:225-226) is then skipped (ignored for multi_label). */
size_t mgdt_nms_workspace_bytes(int n, int nc, int a, int multi_label, int max_nms);
int mgdt_nms_fwd(const float* pred, int n, int nc, int a, float conf_thres, float iou_thres, const int32_t* classes,
                 int n_classes, int agnostic, int multi_label, int max_det, int max_nms, float max_wh, float* out,
                 int32_t* kept_anchor, int32_t* counts, const unsigned long long* best_keys, void* ws, size_t ws_bytes, mgdt_stream s);

/* ---- validator matching: DetectionValidator._process_batch, yolo/v8/detect/val.py:152-175, for a batch ----------
 * det [n][max_det][6] (x1,y1,x2,y2,conf,cls; the layout mgdt_nms_fwd writes) with ndet[n] valid rows, labels [n][max_lab][5]
 * (cls,x1,y1,x2,y2 in the same pixel frame) with nlab[n] valid rows, iouv[n_iou] ascending IoU levels (n_iou <= 16).
 * correct [n][max_det][n_iou] uint8: 1 where the detection is a true positive at that level (rows past ndet are 0). */
int mgdt_val_match_fwd(const float* det, const int32_t* ndet, int n, int max_det, const float* labels, const int32_t* nlab, int max_lab,
                       const float* iouv, int n_iou, uint8_t* correct, mgdt_stream s);

/* ---- v8DetectionLoss: assigner + BCE/CIoU/DFL + gradient w.r.t. the head maps -----------------------------------------
 * yolo/utils/loss.py:108-208 (v8DetectionLoss.__call__, BboxLoss :56-89), yolo/utils/tal.py:56-353
 * (HeuristicPositiveSampleAssigner_v1 -> TaskAlignedAssigner, topk 10, alpha = 0.5*(100 - call_count/161)/100, beta 8),
 * yolo/utils/metrics.py:75-128 (CIoU).  feats[l]: raw head map of level l, NHWC view (B, 4R+nc, H, W); gt: fp32
 * [B][n_gt][5] = (cls, x1, y1, x2, y2) in pixels, zero rows = padding (what loss.py:132-148 `preprocess` builds).
 * out5 (device) = {loss*B, box, cls, dfl (gains applied), target_scores_sum}.  Optional device outputs (may be NULL):
 * fg_out uint8 [B][A], gt_idx_out int32 [B][A], tscore_out fp32 [B][A] (the normalised target score of the assigned class).
 * Top-k ties: lower anchor index first.  n_gt == 0: all-background targets (the reference raises here, tal.py:102-108).
 * bwd: grads[l] = gscale * d(loss*B)/d feats[l]; call after fwd with the same ws / out5 / arguments.                    */
size_t mgdt_detect_loss_workspace_bytes(int b, int a_total, int n_gt);
int mgdt_detect_loss_fwd(const mgdt_view* const* feats, const float* strides, int n_levels, int reg_max, int nc,
                         const float* gt, int n_gt, int call_count, float gain_box, float gain_cls, float gain_dfl,
                         float* out5, unsigned char* fg_out, int32_t* gt_idx_out, float* tscore_out, void* ws,
                         size_t ws_bytes, int dtype, mgdt_stream s);
/* The same with the per-call counter in device memory (a training step captured in a hipGraph; the host advances the counter between
 * replays).  Reference: the `epoch` argument of TaskAlignedAssigner.forward, yolo/utils/loss.py:195-206, tal.py:110,266-267. */
int mgdt_detect_loss_fwd_dev(const mgdt_view* const* feats, const float* strides, int n_levels, int reg_max, int nc,
                             const float* gt, int n_gt, const int32_t* call_count_dev, float gain_box, float gain_cls, float gain_dfl,
                             float* out5, unsigned char* fg_out, int32_t* gt_idx_out, float* tscore_out, void* ws,
                             size_t ws_bytes, int dtype, mgdt_stream s);
int mgdt_detect_loss_bwd(const mgdt_view* const* feats, const mgdt_view* const* grads, const float* strides, int n_levels,
                         int reg_max, int nc, const float* gt, int n_gt, float gain_box, float gain_cls, float gain_dfl,
                         float gscale, const float* out5, void* ws, size_t ws_bytes, int dtype, mgdt_stream s);

/* ---- training side: BatchNorm2d (batch statistics) + activation, its backward, conv dgrad / wgrad, adjoints ----------------
 * Conv.forward in training mode (nn/modules/conv.py:36-38: act(bn(conv(x))) with nn.BatchNorm2d eps 1e-3 / momentum 0.03,
 * yolo/utils/torch_utils.py:254-256) and what torch.autograd derives from it in trainer.py:343 (`scaler.scale(loss).backward()`).
 * ws: mgdt_reduce_workspace_bytes(c) / mgdt_conv_wgrad_workspace_bytes(cin, cout, k) bytes; all reductions have a fixed order.  */
size_t mgdt_reduce_workspace_bytes(int c);
int mgdt_bn_stats_fwd(const mgdt_view* y, float eps, float momentum, float* mean, float* rstd, float* running_mean,
                      float* running_var, void* ws, int dtype, mgdt_stream s);
int mgdt_bn_act_fwd(const mgdt_view* y, const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                    const mgdt_view* r1, const mgdt_view* r2, const mgdt_view* z, int dtype, mgdt_stream s);
int mgdt_bn_act_bwd(const mgdt_view* gz, const mgdt_view* y, const float* mean, const float* rstd, const float* gamma,
                    const float* beta, int act, float* dgamma, float* dbeta, const mgdt_view* dy, void* ws, int dtype, mgdt_stream s);
int mgdt_conv_dgrad(const mgdt_view* dy, const float* w_oihw, int k, int stride, const mgdt_view* dx, int accumulate, int dtype,
                    mgdt_stream s);
/* Grouped / depth-wise convolution (the reference's DWConv, nn/modules/conv.py:82-86; Conv with g > 1): data and weight gradients.
   w / dw: [cout][cin/groups][k][k] fp32 (nn.Conv2d's layout); NHWC views; plain VALU kernels (no target YAML instantiates a grouped conv). */
int mgdt_gconv_dgrad(const mgdt_view* dy, const float* w, int k, int stride, int groups, const mgdt_view* dx, int accumulate, int dtype,
                     mgdt_stream s);
size_t mgdt_gconv_wgrad_workspace_bytes(int cin, int cout, int k, int groups);
int mgdt_gconv_wgrad(const mgdt_view* x, const mgdt_view* dy, int k, int stride, int groups, float* dw, int accumulate, void* ws, int dtype,
                     mgdt_stream s);
size_t mgdt_conv_wgrad_workspace_bytes(int cin, int cout, int k);
int mgdt_conv_wgrad(const mgdt_view* x, const mgdt_view* x2, const mgdt_view* dy, int k, int stride, float* dw_oihw, float* dbias,
                    int accumulate, void* ws, int dtype, mgdt_stream s);
/* Deferred final sums: mgdt_conv_wgrad with dw_oihw == NULL (and dbias == NULL) leaves its per-split partial sums in ws
 * ([mgdt_conv_wgrad_splits(cin, cout, k)][cout][cin][k*k] fp32); mgdt_wgrad_final_batch then finishes many convolutions in one launch, each in the
 * same fixed order as the immediate form (the results are bit-identical).  The caller keeps every ws alive until the batch has run. */
int mgdt_conv_wgrad_splits(int cin, int cout, int k);
typedef struct mgdt_wgrad_final_desc { const float* partial; float* dw; long n; int32_t nsplit; int32_t accumulate; } mgdt_wgrad_final_desc;
int mgdt_wgrad_final_batch(const mgdt_wgrad_final_desc* descs, int n, mgdt_stream s);
int mgdt_add_fwd(const mgdt_view* a, const mgdt_view* b, const mgdt_view* o, int dtype, mgdt_stream s);
int mgdt_maxpool5_bwd(const mgdt_view* x, const mgdt_view* gy, float* gx_f32, int dtype, mgdt_stream s);
int mgdt_nearest_bwd(const mgdt_view* gy, const mgdt_view* gx, int dtype, mgdt_stream s);

/* adjoints of the MSPA pooling attention and GD-neck ops (block.py:209-399, spr_module.py, convnextv2.py:48-77, utils.py:145-182) */
int mgdt_ew_binary(const mgdt_view* a, const mgdt_view* b, const mgdt_view* o, int mode, int dtype, mgdt_stream s);
int mgdt_channel_affine(const mgdt_view* x, const float* scale, const float* shift, const mgdt_view* y, int dtype, mgdt_stream s);
size_t mgdt_nc_reduce_workspace_bytes(int n, int c);
int mgdt_nc_reduce(const mgdt_view* a, const mgdt_view* b, float* out, void* ws, int dtype, mgdt_stream s);
int mgdt_adaptive_avgpool_bwd(const mgdt_view* gy, const mgdt_view* gx, int accumulate, int dtype, mgdt_stream s);
int mgdt_bilinear_bwd(const mgdt_view* gy, const mgdt_view* gx, int accumulate, int dtype, mgdt_stream s);
size_t mgdt_spr_bwd_workspace_bytes(int n, int c, int groups);
int mgdt_spr_bwd(const mgdt_view* gy, const float* pooled_partial, int splits, const float* attn, const float* dattn,
                 const float* fc1_w, const float* fc1_b, const float* fc2_w, const float* fc2_b, int groups,
                 const mgdt_view* gx, float* param_grads, void* ws, int dtype, mgdt_stream s);
int mgdt_dwconv7_ln_train_fwd(const mgdt_view* x, const float* dw_w, const float* dw_b, const float* ln_w, const float* ln_b,
                              float eps, const mgdt_view* y, const mgdt_view* u_out, int dtype, mgdt_stream s);
size_t mgdt_dwconv7_ln_bwd_workspace_bytes(int c);
int mgdt_dwconv7_ln_bwd(const mgdt_view* x, const mgdt_view* u, const mgdt_view* gy, const float* dw_w49c, const float* ln_w, float eps,
                        const mgdt_view* du_tmp, const mgdt_view* dx, int accumulate_dx, float* d_dw_w, float* d_dw_b,
                        float* d_ln_w, float* d_ln_b, void* ws, int dtype, mgdt_stream s);
int mgdt_grn_bwd(const mgdt_view* g, const mgdt_view* t, const float* S, const float* A, const float* B, const float* gamma,
                 const mgdt_view* dt, float* dgamma, float* dbeta, void* ws, int dtype, mgdt_stream s);

/* ---- optimizer step on the flat parameter buffer (yolo/engine/trainer.py:462-470,633-664; torch_utils.py:335-367) ----------
 * clip: out2 = {total grad norm, min(1, max_norm/(norm+1e-6))} on device (no host sync); sgd: torch.optim.SGD(nesterov) with a
 * per-element weight-decay vector (0 for norm weights; a negative entry marks the bias group of trainer.py:644: no decay, stepped
 * with `lr_bias`, the warm-up bias learning rate of trainer.py:323) and the clip coefficient read from device memory.        */
size_t mgdt_grad_norm_workspace_bytes(void);
int mgdt_grad_clip_coef(const float* g, long n, float max_norm, float* out2, void* ws, mgdt_stream s);
int mgdt_sgd_step(float* p, const float* g, float* buf, const float* wd, long n, float lr, float lr_bias, float momentum, int nesterov,
                  int first, const float* clip2, mgdt_stream s);
int mgdt_ema_update(float* ema, const float* p, long n, float decay, mgdt_stream s);
/* SGD + EMA in one launch, scalars from device memory: hyper4 = {lr, lr_bias, momentum, ema_decay}; [0, n_param) parameters, [n_param, n_total)
 * buffers (EMA only; ema may be NULL).  Reference: yolo/engine/trainer.py:317-326 (warm-up values), :462-470, yolo/utils/torch_utils.py:342. */
int mgdt_sgd_ema_step_dev(float* p, const float* g, float* buf, const float* wd, long n_param, float* ema, long n_total, const float* hyper4,
                          int nesterov, int first, const float* clip2, mgdt_stream s);

/* ---- box helpers, validator reductions, predictor preprocess; fp32 boxes, rows of `row` >= 4 floats ------------
 * box_convert: mode 0 = xywh2xyxy (yolo/utils/ops.py:362-377), 1 = xyxy2xywh (:345-359); columns >= 4 are copied.
 * box_iou: pairwise IoU (n,4) x (m,4) -> (n,m), eps in the union (yolo/utils/metrics.py:52-72).
 * bbox_iou: row i of box1 (stride1 floats; 0 broadcasts one box) vs row i of box2; mode 0 IoU, 1 GIoU, 2 DIoU, 3 CIoU (metrics.py:75-128).
 * scale_boxes: in place (b - pad) / gain, then clip to [0, w0] x [0, h0] (ops.py:90-117 + clip_boxes :269-285).
 * letterbox: one uint8 HWC BGR image -> uint8 CHW RGB planes of the letter-boxed batch tensor (augment.py:538-593 geometry computed by the
 *   caller; resize = cv2 INTER_LINEAR 8-bit rule, border 114; predictor.py:123-125 BGR->RGB, HWC->CHW).
 * ap_per_class: the per-class part of metrics.py:410-497 (cumulative TP/FP, recall / precision, compute_ap's envelope + 101-point
 *   interpolation in numpy's own arithmetic order, the 1000-point P/R-vs-confidence curves); inputs grouped by class, descending confidence. */
int mgdt_box_convert(const float* in, float* out, long n, int row, int mode, mgdt_stream s);
int mgdt_box_iou(const float* b1, int n, const float* b2, int m, float eps, float* out, mgdt_stream s);
int mgdt_bbox_iou(const float* b1, int stride1, const float* b2, int stride2, long n, int xywh, int mode, float eps, float* out, mgdt_stream s);
int mgdt_scale_boxes(float* boxes, long n, int row, float gain, float padx, float pady, float h0, float w0, mgdt_stream s);
int mgdt_letterbox_fwd(const void* src_hwc_bgr, int sh, int sw, long src_pitch, void* dst_chw_rgb, int dh, int dw, int new_h, int new_w, int top,
                       int left, mgdt_stream s);
size_t mgdt_ap_workspace_bytes(int n_det, int n_cls);
int mgdt_ap_per_class(const void* tp, const float* conf, const int32_t* seg, const int32_t* nlab, int n_det, int n_cls, int T, const double* x101,
                      const double* px, double eps, void* ws, double* ap, double* pcur, double* rcur, mgdt_stream s);

/* ---- TOODHead training (reference nn/modules/head.py:466-572 under autograd; mmcv ModulatedDeformConv2d backward) -----------------------
 * GroupNorm(16)+act backward of Conv_GN / DyDCNv2.norm (head.py:67-81, block.py:401-432), composed with mgdt_nc_reduce:
 *   gn_affine:          Sy = sum_hw y, Syy = sum_hw y*y  ->  mean / rstd per (image, group), A = gamma*rstd, B = beta - mean*rstd*gamma  (u = y*A + B)
 *   nc_affine_act_bwd:  gu = g * act'(y*A + B)        (A = B = NULL: u = y, i.e. with act = ReLU and y the layer output the ReLU mask)
 *   gn_bwd_coef:        S1 = sum_hw gu, S2 = sum_hw gu*y  ->  P, Q, R with dy = gu*P + y*Q + R, and dgamma / dbeta (written or accumulated: the head is
 *                       shared by all levels); ws: mgdt_gn_bwd_workspace_bytes
 *   nc_axpby:           out = a*sa[n][c] (+ b*sb[n][c]) (+ shift[n][c])     (sa NULL: 1)
 * pixel_gate_bwd: adjoint of mgdt_pixel_gate_fwd (cls_feat * sigmoid(prob logit), head.py:530-537): gx = g*s, glogit = s(1-s) * sum_c g*x.
 * tood_layer_attn_bwd: adjoint of mgdt_tood_layer_attn_fwd (TaskDecomposition.forward head.py:107-116): dscale [n][c] -> dsums [n][c] (gradient
 *   w.r.t. sum_hw feat) and dW1 (hid*c), db1 (hid), dW2 (stacked*hid), db2 (stacked), written or accumulated.
 * dcn_im2col: col[n,y,x,c*9 + tap] = sigmoid(mask logit) * bilinear(x at p + offset) (mmcv modulated_deformable_im2col): the DCN output is the
 *   1x1 convolution of col with weight.view(cout, cin*9), so its weight / column gradients are mgdt_conv_wgrad / dgrad.
 * dcn_col2im_bwd: column gradient -> gx_f32 (dense fp32 NHWC, ZERO on entry; float atomics, the order-dependent scatter of mmcv's col2im) and
 *   gom (offset gradients 0..17, mask-LOGIT gradients 18..26, further channels zeroed) (mmcv modulated_deformable_col2im / col2im_coord). */
int mgdt_gn_affine(const float* Sy, const float* Syy, int n, int c, int hw, const float* gamma, const float* beta, int groups, float eps,
                   float* mean_ng, float* rstd_ng, float* A, float* B, mgdt_stream s);
int mgdt_nc_affine_act_bwd(const mgdt_view* g, const mgdt_view* y, const float* A, const float* B, int act, const mgdt_view* gu, int dtype, mgdt_stream s);
size_t mgdt_gn_bwd_workspace_bytes(int n, int c);
int mgdt_gn_bwd_coef(const float* S1, const float* S2, const float* mean_ng, const float* rstd_ng, const float* gamma, int n, int c, int hw, int groups,
                     float* P, float* Q, float* R, float* dgamma, float* dbeta, int accumulate, void* ws, mgdt_stream s);
int mgdt_nc_axpby(const mgdt_view* a, const float* sa, const mgdt_view* b, const float* sb, const float* shift, const mgdt_view* out, int dtype,
                  mgdt_stream s);
int mgdt_pixel_gate_bwd(const mgdt_view* g, const mgdt_view* x, const mgdt_view* logit, const mgdt_view* gx, const mgdt_view* glogit, int dtype, mgdt_stream s);
size_t mgdt_tood_layer_attn_bwd_workspace_bytes(int n, int c, int hid, int stacked);
int mgdt_tood_layer_attn_bwd(const float* sums, const float* dscale, int n, int c, int hw, const float* w1, const float* b1, const float* w2,
                             const float* b2, int hid, int stacked, float* dsums, float* dw1, float* db1, float* dw2, float* db2, int accumulate,
                             void* ws, mgdt_stream s);
int mgdt_dcn_im2col(const mgdt_view* x, const mgdt_view* om, const mgdt_view* col, int dtype, mgdt_stream s);
int mgdt_dcn_col2im_bwd(const mgdt_view* gcol, const mgdt_view* x, const mgdt_view* om, float* gx_f32, const mgdt_view* gom, int dtype, mgdt_stream s);

#ifdef __cplusplus
}
#endif
#endif /* MGDT_H */
