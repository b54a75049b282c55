/*
 * ideepcolor.h -- C ABI of libideepcolor_hip.so: the MI355X (gfx950) implementation of the
 * Local-Hints colorization forward pass.
 *
 * The reference has NO FFI/plugin interface for this path: the boundary is the
 * duck-typed Python class in data/colorize_image.py whose net_forward() calls
 *     self.net.forward(self.img_l_mc, self.input_ab_mc, self.input_mask_mult, self.mask_cent)
 * at data/colorize_image.py:263 (torch backend) / :427-428 (caffe backend).  Everything below
 * replaces exactly that call and the module it lands in, models/pytorch/model.py:134-175
 * (SIGGRAPHGenerator.forward).  Each entry point cites what it stands in for.
 *
 * Conventions: extern "C", plain pointers and sizes, no torch/C++ types.  Every function returns
 * IDC_OK (0) or a negative idc_status; idc_last_error() gives the text.  Host tensors are NCHW
 * fp32, C-contiguous, caller-owned (the layout of the reference's torch tensors).  A handle owns
 * one device, one stream, all device memory; it is not thread-safe; distinct handles are
 * independent.  There is no CPU fallback: without a gfx950 device idc_create fails.
 */
#ifndef IDEEPCOLOR_H
#define IDEEPCOLOR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IDC_VERSION 2   /* (round 6 adds precisions 2 / 3 without a bump: older blobs stay valid) 2: blob header flags may carry IDC_FLAG_THROUGHPUT_BLOB; round-3 blobs (Winograd images added without a bump) were 1 */

typedef struct idc_context* idc_handle;

typedef enum idc_status {
    IDC_OK = 0,
    IDC_ERR_INVALID_ARG = -1,
    IDC_ERR_NO_DEVICE = -2,     /* no HIP device / not gfx950 */
    IDC_ERR_HIP = -3,           /* a HIP runtime call failed */
    IDC_ERR_NO_WEIGHTS = -4,    /* forward before weights (reference: "I need to have a net!", colorize_image.py:88-90) */
    IDC_ERR_MISSING_KEY = -5,   /* a state_dict key of SURVEY.md Appendix B is absent or mis-shaped */
    IDC_ERR_BATCH = -6,         /* n > max_batch or n <= 0 */
    IDC_ERR_UNSUPPORTED = -7,
    IDC_ERR_INTERNAL = -8       /* a kernel variant was selected for a launch it does not cover: a bug in the library, reported instead of
                                 * falling back to another kernel with the wrong weight image */
} idc_status;

/* Arithmetic type of the conv stack.  BF16: bf16 activations+weights, fp32 MFMA accumulation,
 * fp32 bias/BN/shortcut sums.  FP32: exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) end to end.      */
typedef enum idc_precision { IDC_FP32 = 0, IDC_BF16 = 1, IDC_BF16X3 = 2, IDC_BF16X6 = 3, IDC_FP16X3 = 4, IDC_FP16 = 5 } idc_precision;
/* BF16X3 / BF16X6 (round 6): the fp32 contract of colorize_image.py:263 carried on the bf16 matrix pipe.  Every fp32 operand travels as a sum of
 * bf16 values -- x = hi + lo (X3) or hi + mid + lo (X6: all 24 mantissa bits) -- and a product x*w is the sum of the bf16 products that matter:
 * hi.hi + lo.hi + hi.lo (three v_mfma_f32_16x16x32_bf16 per fragment pair, 2^-16 relative), or those + mid.hi + hi.mid + mid.mid (six, 2^-24:
 * fp32-equivalent), all into ONE fp32 accumulator set.  Activations between layers are stored split (2 or 3 bf16 planes per pixel), bias / BN /
 * shortcut sums / the tanh head stay fp32, model1 (4 -> 64 -> 64 channels, 3 % of the MACs) runs on the exact-fp32 kernels.  Throughput path:
 * every layer runs the large-tile kernels whatever the batch (the batch-1 click path keeps IDC_FP32 / IDC_BF16).  Error against the float64
 * oracle: X6 at or below IDC_FP32's, X3 <= 1e-3 on the +-110 ab map with torch-default-init weights (tests/bounds.py).
 * FP16X3 (round 6): BF16X3's planes, segments and kernels with FP16 parts -- x = hi + lo with 11-bit hi and lo (22 bits, 2^-22 relative per
 * operand instead of 2^-16), three v_mfma_f32_16x16x32_f16 per fragment pair.  A layer's weight parts hold w * 2^s, s the power of two that brings
 * the layer's largest weight into [8192, 16384) (exact; the accumulators are multiplied by 2^-s when the bias joins), so that the lo part of a small
 * weight is a normal fp16 number: at the IDC_FP32 path's distance from the float64 oracle on BOTH weight styles (N = 32: 2.2e-5 torch-default-init,
 * 1.9e-3 full-range, against 2.3e-5 / 1.7e-3; without the scale full-range weights ~0.02 had subnormal lo parts and 3.9e-3), at BF16X3's rate.
 * The price is fp16's range for ACTIVATIONS: beyond +-65504 they saturate (the conversions clamp; nothing becomes inf), below 6e-5 they keep 6e-8 absolute.
 * FP16 (round 6): FP16X3's graph with ONE part and ONE product -- plain fp16 operands (11 significant bits against bf16's 8; unscaled weights), fp32
 * accumulation, every layer on the throughput tiles, the launches on the fp16 twins of the bf16 kernels: 1 / 9 ... 1 / 7 of IDC_BF16's rounding error at
 * 0.96 of its N = 32 rate.  Not a 1e-3 path.
 * The layers of this net are BatchNorm-ed / ReLU-ed activations of O(1..100); a checkpoint with larger activations wants BF16X6. */

/* idc_create flags */
#define IDC_FLAG_DIST_HEAD   0x1u  /* also build model_class (529-bin) head: SIGGRAPHGenerator(dist=True), model.py:105,159-160 */
/* (0x2 reserved: hipGraph replay was left out of the batch-1 path on measurement -- its 51 launches leave < 1 us of gaps
 *  in a 615 us forward, profiles/r02a_click_bf16_trace.txt) */
#define IDC_FLAG_DIST313     0x8u  /* also build the 313-bin distribution / soft-decode head of
                                      models/reference_model/deploy_nopred.prototxt:650-850 (needs the pred.* tensors, see idc_forward_dist313) */
#define IDC_FLAG_GLOBAL_HINTS 0x4u /* also build the Global-Hints branch of models/global_model/deploy_nodist.prototxt:37-172,
                                      501-518 (needs the glob.* tensors, see idc_set_global_hints) */
#define IDC_FLAG_THROUGHPUT_BLOB 0x10u /* weight blob WITHOUT the Winograd images (U = G g G^T of the 3x3 layers and the deconvs).  Since
                                        * round 5 only the fp32 blob carries them (384 MB -> 136 MB with the flag; the direct fp32 kernels then run:
                                        * same results within tolerance, slower); a bf16 blob is 68 MB with or without the flag (round 6; 136 MB in the -DIDC_AB_PARTNERS build, which also carries the layout-2 images) -- the bf16 click
                                        * path's kernels (conv_kwave_*) read the same layout-1 images as the throughput kernels. */

/* ---- library ------------------------------------------------------------------------------- */
int idc_version(void);
/* Process-wide tile-shape policy of the conv kernels (speed only -- every policy computes the same
 * function): 0 automatic (default), 1 small tiles only (conv_igemm), 2 the large-tile bf16 kernel
 * (conv_igemm_v2) wherever it applies.  Exists so that the parity tests can drive every kernel
 * variant at small sizes.  No reference counterpart. */
int idc_set_tile_policy(int policy);
/* Process-wide switches for the parity tests and A/B measurements (speed / kernel choice only: every setting computes the same
 * function).  Eleven names (round 6: three former environment switches became options; the library reads NO tuning knob from the environment):
 *   "fuse_conv1"  (1)  model1 = conv1_1 + conv1_2 as one launch on the bf16 path; 0 keeps the two launches apart, so that conv1_1's own
 *                      output exists and can be read with idc_get_activation.
 *   "click"       (-1 = on)  small launches (the batch-1 click path, fp32 and what "kwave" does not cover) run conv_click
 *                      (weight tiles by LDS-DMA, fragments prefetched across steps); 0 keeps them on conv_igemm.
 *   "winograd"    (1)  fp32 path: 3x3 stride-1 layers as Winograd F(2x2,3x3), small deconv launches as F(2x2,2x2).  0 = direct kernels,
 *                      1 = automatic, 2 = every deconv too (tests), 12 / 21 / 22 = automatic with the 3x3 form <TB,CB> forced (tests).
 *   "mfma16"      (1)  bf16 throughput tile from v_mfma_f32_16x16x32_bf16 (conv_igemm_v2m / v2p).  0 = conv_igemm_v2 (32x32x16): A/B partner,
 *                      exists only in the -DIDC_AB_PARTNERS build; the default library answers IDC_ERR_UNSUPPORTED.
 *   "v2p"         (1)  the 3x3 convs among them as conv_igemm_v2p (column-swizzled halo tile, unrolled taps; bit-identical to conv_igemm_v2m).
 *   "ds_mfma16"   (1)  deconv + shortcut launches as conv_ds_fused_m, grids with fewer 128-cout workgroups than CUs (model10up of ONE 256x256 image)
 *                      in its 64-cout 4-wave form; 2 = 8-wave workgroups on every grid (A/B, tests); 0 = conv_ds_fused: partner build only.
 *   "split_ds_fuse" (1)  operand-split precisions: each ConvTranspose + the 3x3 shortcut conv it is summed with as ONE launch (conv_ds_fused_ms / _msh:
 *                      both K loops walked per segment into one fp32 accumulator set); 0 = shortcut conv (fp32 sums through HBM) + deconv launch.
 *   "conv1_1_split" (1)  operand-split precisions: conv1_1 (their exact-fp32 island) on conv1_1_split_kernel where the grid is throughput-sized; 0 = conv_igemm<float>.
 *   "conv1_2_split" (1)  ... and conv1_2 (64 -> 64 at full resolution) on conv1_2_split_kernel (conv1_block_fused_t's conv1_2 tile walked per segment); 0 = the generic
 *                      64-cout tile conv_igemm_v2ps<1,4,1>.
 *   "fp16_fast"   (1)  IDC_FP16: launches the bf16 throughput kernels cover run their fp16 twins (conv_igemm_v2ph, conv_ds_fused_mh); 0 = the one-segment
 *                      operand-split kernels everywhere.
 *   "kwave"       (1)  bf16 batch-1 click path: 3x3 stride-1 layers and ConvTranspose launches as conv_kwave_bf16 / conv_kwave_deconv_bf16
 *                      (direct form, K split over the waves of a workgroup); 0 = conv_click + split-K (round 2's kernels).
 *   "kwave_chain" (2)  ... and runs of consecutive same-shape 512-channel layers of that path (conv4_2 .. conv7_3 at batch 1) as ONE
 *                      persistent launch with a grid barrier between layers (conv_kwave_chain_bf16): 0 = one launch per layer,
 *                      1 = through hipLaunchCooperativeKernel, 2 = plain launch after an occupancy check; a workgroup that never sees the
 *                      others gives up after ~0.3 s, that forward fails with IDC_ERR_INTERNAL and the handle goes back to one launch per layer.
 *   "spin_sync"   (1)  calls that serve one or two images wait by polling the stream (bounded) instead of parking on an interrupt; 0 = blocking wait.
 *   "pcie_kernel" (1)  their host <-> device transfers (<= 2 MiB, pinned) run as a copy kernel on the forward's stream; 0 = hipMemcpyAsync.
 *   "kw_force_abort" (0)  TEST HOOK: 1 makes the persistent trunk launch's first grid barrier unreachable (plays "workgroups never co-resident").
 * Retired with their kernels, or folded into the above: "fuse_conv1_small", "winograd_bf16", "winograd_form", "winograd_deconv", "conv1_lw",
 * "code_warm", "kwave_deconv" (IDC_ERR_INVALID_ARG).
 * Take effect on the next forward; unknown names return IDC_ERR_INVALID_ARG. */
int idc_set_option(const char* name, int value);
/* How a call waits for the device (no reference counterpart): calls that serve ONE OR TWO images (the click path) poll the stream instead of
 * blocking in hipStreamSynchronize -- the interrupt wake-up after the copy back costs 15-20 us of a 0.4 ms click -- for a bounded time (a few
 * milliseconds, then they block); batches always block.  IDC_SPIN_SYNC=0 in the environment (read once) restores the blocking wait.
 * The same calls move their transfers of at most 2 MiB between PINNED host memory and the device with a copy kernel on the handle's stream
 * instead of a copy engine (no cross-queue hand-over; IDC_PCIE_KERNEL=0 = hipMemcpyAsync everywhere). */
/* Split-K policy of the small-tile kernels (speed only): 0 automatic (launches too small to fill the chip: the
 * batch-1 click path), 1 never, 2 always (tests).  The slice sums are added in a fixed order: results stay
 * deterministic and independent of how many images a call carries. */
int idc_set_splitk_policy(int policy);
/* Number of visible HIP devices (0 when none; never fails). */
int idc_device_count(void);
/* Text of the last error on this handle (h may be NULL: last error of a failed idc_create or of a
 * handle-less call on the calling thread). */
const char* idc_last_error(idc_handle h);

/* ---- lifetime: replaces SIGGRAPHGenerator.__init__ (model.py:6-132) + net.eval()/.cuda()
 *      in ColorizeImageTorch.prep_net (colorize_image.py:216-233) ---------------------------- */
int idc_create(int device_id, int height, int width, int max_batch, int precision, unsigned flags,
               idc_handle* out);
int idc_destroy(idc_handle h);

/* Input/output scaling: {l_div, ab_div, mask_mul, out_mul}.  Default = torch backend
 * {100, 110, 1, 110} (model.py:148,175).  The Caffe twin feeds raw values and scales by 100:
 * {1, 1, 1, 100} (colorize_image.py:379-383,425; deploy_nodist.prototxt:812-821).              */
int idc_set_io_scales(idc_handle h, float l_div, float ab_div, float mask_mul, float out_mul);

/* ---- weights: replaces torch.load + load_state_dict (colorize_image.py:222-229) ------------ *
 * Tensors are given under the reference state_dict key names in torch
 * layouts: Conv2d (Cout,Cin,kh,kw); ConvTranspose2d (Cin,Cout,4,4); BatchNorm weight/bias/
 * running_mean/running_var.  Unknown keys (num_batches_tracked, model_class.* without the dist
 * flag) are ignored; a missing key is IDC_ERR_MISSING_KEY.                                      */
typedef struct idc_tensor_desc {
    const char* name;      /* e.g. "model1.0.weight" */
    const float* data;     /* host fp32, C-contiguous */
    int ndim;
    int64_t dims[4];
} idc_tensor_desc;

/* Size of the packed, device-ready weight blob (MFMA-tiled, LDS-swizzled). */
size_t idc_weights_blob_bytes(int precision, unsigned flags);
/* Host-only: pack a state_dict into `blob` (no device needed; used by rank 0 before the RCCL
 * broadcast and by the CPU-side tests). */
int idc_pack_weights(int precision, unsigned flags, const idc_tensor_desc* tensors, int n_tensors,
                     void* blob, size_t blob_bytes);
/* Upload a packed blob from host memory (H2D copy into handle-owned memory). */
int idc_set_weights_host(idc_handle h, const void* blob, size_t blob_bytes);
/* Use a packed blob that already lives in device memory (e.g. a torch uint8 tensor that just
 * received the RCCL broadcast).  copy=0 adopts the pointer (caller keeps it alive), copy=1 does
 * a D2D copy into handle-owned memory. */
int idc_set_weights_device(idc_handle h, const void* dev_blob, size_t blob_bytes, int copy);
/* Convenience = idc_pack_weights + idc_set_weights_host. */
int idc_load_weights(idc_handle h, const idc_tensor_desc* tensors, int n_tensors);
/* Device pointer of the blob in use (NULL before weights are set). */
const void* idc_weights_device_ptr(idc_handle h);

/* ---- forward: replaces SIGGRAPHGenerator.forward (model.py:134-175) ------------------------ *
 * L_mc [n,1,H,W] = L-50; ab [n,2,H,W] raw Lab ab hints (0 where none); mask [n,1,H,W] in {0,1};
 * maskcent = the mask_cent argument (0 or .5, colorize_image.py:210,263).
 * out_ab [n,2,H,W] = 110*tanh(.) -- the tensor the reference returns at colorize_image.py:263.
 * Host-pointer form: blocking; returns when out_ab is valid.  Each pointer that is pinned host memory (idc_alloc_host,
 * hipHostMalloc, hipHostRegister) is transferred in place; pageable ones go through the handle's pinned staging.      */
int idc_forward(idc_handle h, int n, const float* L_mc, const float* ab, const float* mask,
                float maskcent, float* out_ab);
/* Device-pointer form (same layouts, device memory); enqueued on the handle's stream (see STREAM ORDERING at
 * idc_stream: inputs produced / outputs consumed on another stream need idc_stream_wait / idc_stream_signal).
 * sync!=0 waits for completion. */
int idc_forward_device(idc_handle h, int n, const float* d_L_mc, const float* d_ab,
                       const float* d_mask, float maskcent, float* d_out_ab, int sync);
/* dist=True variant (needs IDC_FLAG_DIST_HEAD): additionally writes the 529-bin distribution
 * softmax(0.2*logits) at quarter resolution, dist_q [n,529,H/4,W/4] (the reference's out_cl is
 * its nearest x4 upsample, model.py:160: out_cl[:, :, y, x] == dist_q[:, :, y/4, x/4]).
 * dist_q may be NULL: the distribution then stays resident on the device for idc_dist_at /
 * idc_suggest_colors / idc_get_dist (no 8.7 MB-per-image copy-out on the click path).           */
int idc_forward_dist(idc_handle h, int n, const float* L_mc, const float* ab, const float* mask,
                     float maskcent, float* out_ab, float* dist_q);
/* ---- Global Hints: replaces the glob_ab_313_mask / s_avg_mask inputs of the Caffe global net
 *      (models/global_model/deploy_nodist.prototxt:7-18) as filled by ColorizeImageCaffeGlobDist.net_forward
 *      (colorize_image.py:451-459).  Needs IDC_FLAG_GLOBAL_HINTS.  glob_ab_313_mask [n,314] = 313-bin ab
 *      histogram + 0/1 flag; s_avg_mask [n,2] = mean saturation + flag, or NULL for zeros (the reference
 *      wrapper never fills it).  The values stay in effect for every later forward (images beyond n and a
 *      handle that was never given hints see all-zero inputs -- which is what the reference feeds for
 *      glob_dist == -1: the branch still runs and adds BN(relu(bias)) chains to conv4_3norm).
 *      Weights: 1x1 convs "glob.s_conv1", "glob.glob_conv1" .. "glob.glob_conv4" (.weight (512,Cin,1,1), .bias)
 *      and BatchNorms "glob.bn1" .. "glob.bn4" (weight, bias, running_mean, running_var) -- the Caffe layer
 *      names of the prototxt, in the key style of the converted caffemodel.pth. ----------------------- */
int idc_set_global_hints(idc_handle h, int n, const float* glob_ab_313_mask, const float* s_avg_mask);
int idc_clear_global_hints(idc_handle h);
/* ---- 313-bin distribution head + annealed-mean soft-decode: the Caffe distribution net
 *      models/reference_model/deploy_nopred.prototxt:650-850 as driven by ColorizeImageCaffeDist
 *      (colorize_image.py:466-507).  Needs IDC_FLAG_DIST313.  Hyper-column sum conv3_pred + conv4..7_pred
 *      (ConvTranspose 4x4 s2) + conv8_pred -> ReLU -> pred_313 (1x1, 313 logits at H/4) -> the two grouped
 *      bilinear x2 deconvs (kernel set at colorize_image.py:410-413) -> per full-resolution pixel
 *          dist_S  = softmax(S * logits)                 [n,313,H,W]  (S = 0.2, colorize_image.py:482-485), may be NULL
 *          pred_ab = W_ab . softmax(2.6 * logits) + b_ab [n,2,H,W]    (W_ab = pts_in_hull.T, colorize_image.py:405-407)
 *      out_ab (may be NULL) receives the regression head as in idc_forward.
 *      Weights: "pred.conv3_pred", "pred.conv8_pred" (384,256,3,3); "pred.conv4_pred" .. "pred.conv7_pred"
 *      ConvTranspose (512,384,4,4); "pred.pred_313" (313,384,1,1); "pred.pred_ab" (2,313,1,1) -- .weight/.bias. */
int idc_forward_dist313(idc_handle h, int n, const float* L_mc, const float* ab, const float* mask,
                        float maskcent, float* out_ab, float* pred_ab, float* dist_S);
/* Temperature of dist_S (the reference's scale_S parameter, colorize_image.py:482-485).  Default 0.2. */
int idc_set_dist_temperature(idc_handle h, float S);
/* ---- colour post-processing on the device: replaces lab2rgb_transpose + _set_out_ab_
 *      (colorize_image.py:20-28,196-198,264-267), i.e. skimage.color.lab2rgb / rgb2lab (sRGB, D65, 2 degree
 *      observer), which the reference runs on the host inside every net_forward.
 *      rgb [n,H,W,3] uint8 = (clip(lab2rgb(L, ab), 0, 1) * 255) truncated, exactly the reference expression;
 *      lab_q [n,3,H,W] float64 (may be NULL) = rgb2lab(rgb / 255): the "refreshed" output_lab / output_ab the
 *      reference keeps.  Arithmetic in float64 like skimage.
 *      idc_forward_rgb = idc_forward followed by the post step on the device-resident result, with
 *      L = L_mc + l_cent (l_cent = 50, colorize_image.py:205); out_ab may be NULL. ------------------------- */
int idc_lab2rgb(idc_handle h, int n, const float* L, const float* ab, uint8_t* rgb, double* lab_q);
int idc_forward_rgb(idc_handle h, int n, const float* L_mc, const float* ab, const float* mask, float maskcent,
                    float l_cent, float* out_ab, uint8_t* rgb, double* lab_q);
/*      Round 5 (the reference-API call's host time): of the 2.2 MB idc_forward_rgb sends back per 256x256 click, the value
 *      net_forward RETURNS is the 0.2 MB uint8 image (colorize_image.py:264-268); the ab map (:263) and the refreshed Lab
 *      (_set_out_ab_, :196-198) are attributes a caller may or may not read.  idc_forward_rgb_lazy computes all three on the
 *      device and copies only rgb; idc_fetch_outputs copies the resident ab map (out_ab [n,2,H,W] f32) and / or refreshed Lab
 *      (lab_q [n,3,H,W] f64) of the LAST forward when asked (either may be NULL).  Same values as idc_forward_rgb.
 *      idc_forward_rgb_lazy accepts L_mc == NULL = "the L planes idc_set_image_l left in the handle": the image's L plane does not change
 *      between the clicks on it (colorize_image.py:161-191 sets it in set_image), so the wrapper uploads it once per image. */
int idc_forward_rgb_lazy(idc_handle h, int n, const float* L_mc, const float* ab, const float* mask, float maskcent,
                         float l_cent, uint8_t* rgb);
int idc_fetch_outputs(idc_handle h, int n, float* out_ab, double* lab_q);
/* ---- Global statistics of a reference image: replaces the global_stats.prototxt net the
 *      notebook runs to obtain glob_dist (DemoGlobalHistogramTransfer.ipynb:176-186; models/global_model/
 *      global_stats.prototxt:10-31,101-111,214-244; caffe_traininglayers.py:53-119,161-196; color_quantization.py:7-33):
 *      RGB uint8 -> Lab (skimage rgb2lab) -> 4x4 average pool of ab -> hard 1-nearest-neighbour assignment to the
 *      313 bin centres (NN = 1: weight 1) -> global average = histogram [n,313] (rows sum to 1); s_avg [n] (may be
 *      NULL) = global mean of the HSV saturation (skimage rgb2hsv), the value s_avg_mask would carry.
 *      rgb [n,H,W,3] with the handle's H, W; centres [313,2] = pts_in_hull (a, b). ------------------------------ */
int idc_global_histogram(idc_handle h, int n, const uint8_t* rgb, const float* centres, float* hist, float* s_avg);

/* ---- Click session: the image's L plane and the hint planes stay on the device, a click
 *      sends its rectangle list only.  Replaces the host work in front of every net_forward of the GUI:
 *      UIControl.get_input (ui/ui_control.py:177-187, PointEdit.updateInput :52-63: cv2.rectangle, filled, corners
 *      inclusive, later edits paint over earlier ones, canvas starts black / mask 0) followed by rgb2lab of the
 *      canvas and mask > 0 in gui_draw.compute_result (ui/gui_draw.py:273-277) -- mode IDC_HINT_RGB, c0..c2 = the
 *      uint8 RGB of the edit; or the notebook's put_point (DemoInteractiveColorization.ipynb:131-139) -- mode
 *      IDC_HINT_AB, (c0, c1) = the (a, b) value.  Rectangles are clipped to the image; uncovered pixels get
 *      ab = 0, mask = 0.  mask_value = what a covered mask pixel holds (1; 110 for the Caffe nets' mask_mult).
 *      idc_set_image_l: L_mc [H,W] = L - 50 of image slot img (colorize_image.py:68-77 img_l_mc), uploaded once.
 *      idc_forward_resident: forward of slots 0..n-1 from the resident planes; out_ab [n,2,H,W], rgb [n,H,W,3] u8,
 *      lab_q [n,3,H,W] f64 as in idc_forward_rgb -- each may be NULL.  Handles with a distribution head leave the
 *      distribution resident (idc_keep_dist for the 313 head).  idc_get_hint_planes reads the planes back
 *      (tests; the wrapper's input_ab / input_mask attributes). -------------------------------------------- */
enum { IDC_HINT_AB = 0, IDC_HINT_RGB = 1 };
typedef struct idc_hint { int32_t y0, x0, y1, x1; float c0, c1, c2; } idc_hint;
int idc_set_image_l(idc_handle h, int img, const float* L_mc);
int idc_set_hints(idc_handle h, int img, int n_hints, const idc_hint* hints, int mode, float mask_value);
int idc_get_hint_planes(idc_handle h, int img, float* ab, float* mask);
int idc_forward_resident(idc_handle h, int n, float maskcent, float l_cent, float* out_ab, uint8_t* rgb, double* lab_q);

/* ---- Colour suggestions: ColorizeImageTorchDist.get_ab_reccs (colorize_image.py:322-354)
 *      on the device-resident distribution of the last forward, instead of copying 529 x H x W probabilities to
 *      the host per click.  idc_dist_bins: 529 / 313 / 0.  idc_dist_at: the B probabilities of pixel (y, x) of
 *      image img (the 529 head is stored at H/4: pixel (y/4, x/4), model.py:131,160).  idc_get_dist: the whole
 *      resident tensor ([n,529,H/4,W/4] or [n,313,H,W]).  idc_keep_dist: 313 handles write dist_S on every
 *      forward (82 MB per 256x256 image of device traffic) so that suggestions can follow.
 *      idc_suggest_colors: cmf = cumsum(pdf)/sum (fp32, sequential); N inverse-CDF draws (np.digitize) from the
 *      counter-based generator u_i = lowbias32(i*0x9E3779B9 + seed*0x85EBCA6B + 0x165667B1) >> 8 / 2^24; k-means
 *      with K clusters over the drawn colours centres[bin] ([B,2] = pts_grid / pts_in_hull) -- greedy k-means++
 *      seeding (most-drawn bin, then argmax count x D^2), Lloyd to a fixed point, float64; out_centres [K,2] and
 *      out_conf [K] (= cluster share of the N draws) ordered by occupancy, descending; out_counts [B] (may be
 *      NULL) = draws per bin.  The reference draws from numpy's global RNG and uses sklearn's randomly seeded
 *      KMeans, so its output is not reproducible run to run; this one is a deterministic function of seed. --- */
int idc_dist_bins(idc_handle h);
int idc_keep_dist(idc_handle h, int on);
int idc_dist_at(idc_handle h, int img, int y, int x, float* pdf);
int idc_get_dist(idc_handle h, int n, float* dist);
int idc_suggest_colors(idc_handle h, int img, int y, int x, int K, int N, unsigned seed, const float* centres,
                       double* out_centres, double* out_conf, unsigned* out_counts);
int idc_sync(idc_handle h);
/* The hipStream_t all work of this handle is enqueued on (as void*): a stream of its own, created non-blocking.
 * STREAM ORDERING: idc_forward_device / idc_forward_resident only ENQUEUE on that stream.  A caller that writes the
 * device inputs or reads the device outputs on another stream (torch's current stream, say) must order the two --
 * either synchronise fully (sync != 0 / idc_sync on this side, a stream/device synchronize on the producer's side),
 * or use the two event helpers below:
 *   idc_stream_wait(h, s)   : work enqueued on the handle AFTER this call waits for everything enqueued on s BEFORE it
 *                             (call it after producing the inputs on s, before idc_forward_device);
 *   idc_stream_signal(h, s) : work enqueued on s AFTER this call waits for everything the handle enqueued BEFORE it
 *                             (call it after idc_forward_device, before consuming d_out_ab on s).
 * The host-pointer entry points (idc_forward, ...) block and need none of this. */
void* idc_stream(idc_handle h);
int idc_stream_wait(idc_handle h, void* caller_stream);
int idc_stream_signal(idc_handle h, void* caller_stream);

/* ---- end-to-end batches: overlapped transfers.  The reference has no counterpart (it
 *      runs one image per call on the host); this is the serving form of idc_forward.  Two slots (0, 1): while slot k
 *      computes, slot 1-k's inputs travel host -> device on a copy stream and its previous result travels back on a third.
 *          idc_forward_async(h, 0, ...batch 0...); idc_forward_async(h, 1, ...batch 1...);
 *          idc_wait(h, 0) -> out of batch 0 valid; idc_forward_async(h, 0, ...batch 2...); idc_wait(h, 1); ...
 *      Host buffers stay caller-owned and must stay untouched until the slot's idc_wait.  Pinned memory (idc_alloc_host,
 *      or the caller's own hipHostMalloc / hipHostRegister) is transferred in place; pageable memory is staged through
 *      pinned buffers with a host memcpy on the calling thread (correct, but the memcpy then bounds the rate).
 *      The blocking entry points drain both slots first.  Each slot owns its device planes: a pipelined batch neither
 *      reads nor replaces the handle's resident L / hint planes or its resident result (idc_forward_resident,
 *      idc_upsample_lab2rgb keep referring to the last BLOCKING forward). */
void* idc_alloc_host(size_t bytes);
int idc_free_host(void* p);
int idc_forward_async(idc_handle h, int slot, int n, const float* L_mc, const float* ab, const float* mask, float maskcent,
                      float* out_ab);
int idc_wait(idc_handle h, int slot);
/* Where the stages of the slot's last completed batch sat on the device clock: ms6 = milliseconds since the pipeline's
 * first use of {H2D start, H2D end, compute start, compute end, D2H start, D2H end} (HIP events on the three streams).
 * Call after idc_wait(slot).  bench.py's end_to_end leg prints the spans and the overlap achieved, so that a box whose
 * copies do not run under the other slot's kernels shows it (SURVEY.md 8d config 3 "end-to-end").
 * ZERO-COPY alternative (no copy engine involved at all): idc_forward_device accepts PINNED HOST pointers
 * (idc_alloc_host / hipHostMalloc memory is mapped into the device's address space at the same address): conv1's
 * operand staging then reads L / ab / mask over PCIe and the tanh head writes out_ab straight into host memory; the
 * result is valid after idc_sync (or an idc_stream_signal'ed consumer).  Bit-identical to idc_forward. */
int idc_pipeline_times(idc_handle h, int slot, float* ms6);

/* ---- multi-GPU weight distribution (SURVEY.md 8b export list, 8e): one RCCL broadcast of the packed blob over xGMI.
 *      Every rank (one process per GPU) creates its handle; the root also loads the weights.  idc_comm_unique_id
 *      (root only) fills 128 bytes that the caller ships to the other ranks by any side channel (a file, a socket,
 *      torch.distributed's store); then EVERY rank calls idc_broadcast_weights(h, id, rank, world, root).  Non-root
 *      handles receive into their own device allocation, verify header + checksum and become ready.  librccl is
 *      opened on first use -- the copy that ships beside the libamdhip64 this library is bound to (RCCL launches on the
 *      handle's stream, so both must belong to ONE HIP runtime; a process can hold torch's bundled runtime as well);
 *      IDC_ERR_UNSUPPORTED if it cannot be found.  EXPERIMENTAL at world > 1: exercised on hardware with one rank
 *      only (no multi-GPU box was available to the builder); sharded.py's default transport is torch.distributed.  No data-path collective exists: images are
 *      independent (colorize_image.py:232 eval-mode BN), each rank runs its own shard. */
#define IDC_UNIQUE_ID_BYTES 128
int idc_comm_unique_id(void* id128);
int idc_broadcast_weights(idc_handle h, const void* unique_id, int rank, int world, int root);

/* ---- display / full-resolution step on the device: what follows every net_forward in the GUI,
 *          ab_win = cv2.resize(output_ab, (win_w, win_h), interpolation=cv2.INTER_CUBIC)
 *          pred_rgb = (clip(lab2rgb(concat(l_win, ab_win)), 0, 1) * 255).astype('uint8')          ui/gui_draw.py:280-283
 *      and the full-resolution getters get_img_fullres / get_input_img_fullres / get_sup_fullres
 *      (scipy.ndimage.zoom(ab, order=1 | 0) + lab2rgb with img_l_fullres, colorize_image.py:123-158).
 *      source: which resident [2,H,W] ab planes of image slot img -- IDC_SRC_OUTPUT_AB the refreshed (uint8-quantised)
 *      output_ab of the last idc_forward_rgb / idc_forward_resident (float64, what the GUI resizes), IDC_SRC_OUTPUT_AB_RAW
 *      the network's own output, IDC_SRC_INPUT_AB the resident hint planes.  interp: IDC_INTERP_CUBIC = cv2 INTER_CUBIC
 *      (a = -0.75, half-pixel centres, replicated border), IDC_INTERP_LINEAR / IDC_INTERP_NEAREST = scipy zoom order 1 / 0
 *      (corner-aligned).  L [out_h,out_w] float64 = the L channel at the output size (l_win / img_l_fullres);
 *      rgb [out_h,out_w,3] uint8. */
enum { IDC_INTERP_CUBIC = 0, IDC_INTERP_LINEAR = 1, IDC_INTERP_NEAREST = 2 };
enum { IDC_SRC_OUTPUT_AB = 0, IDC_SRC_OUTPUT_AB_RAW = 1, IDC_SRC_INPUT_AB = 2 };
int idc_upsample_lab2rgb(idc_handle h, int img, int source, int interp, int out_h, int out_w, const double* L, uint8_t* rgb);

/* ---- introspection for parity tests and roofline accounting -------------------------------- */
int idc_num_layers(idc_handle h);
typedef struct idc_layer_info {
    char name[32];        /* caffe-style layer name: conv1_1 ... conv10_2, conv3_3_short, head ... */
    char kernel[48];      /* kernel family launched */
    double flops;         /* ALGORITHMIC flops per image (2*MACs), 0 for non-conv */
    double min_bytes;     /* algorithmic HBM bytes per image: input + output + residual (+weights once) */
    int launches;         /* kernel launches per forward */
} idc_layer_info;
int idc_layer_info_get(idc_handle h, int layer, idc_layer_info* out);
/* Per-layer timing with hipEvents on the handle's stream.  While on, every forward records an
 * event pair around each layer (kept for the last 32 forwards); idc_layer_times_ms() syncs and
 * returns, for layers [0, idc_num_layers), the mean duration (ms) over the forwards recorded
 * since profiling was switched on (at most the last 32). */
/* on = 2: a single event pair around the whole forward instead (ms[0] of idc_layer_times_ms = mean forward duration,
 * the other entries 0) -- the per-launch pairs of mode 1 slow a batch-32 forward by about 4 %. */
int idc_set_profiling(idc_handle h, int on);
int idc_layer_times_ms(idc_handle h, float* ms, int capacity);
/* The same recorded forwards as per-layer min / median / max (ms) instead of the mean: tells a layer that is slow on every forward
 * from one that was slow once (first launch of a kernel variant, a clock step) -- bench.py's `layers_ms` is the median of 20. */
int idc_layer_times_stats(idc_handle h, float* ms_min, float* ms_median, float* ms_max, int capacity);
/* Copy an intermediate activation of the LAST forward to the host as NCHW fp32.
 * name = idc_layer_info.name; out must hold n*C*H*W floats; *C,*H,*W are returned.            */
int idc_get_activation(idc_handle h, const char* name, int n, float* out, size_t capacity_floats,
                       int* C, int* H, int* W);

/* ---- single operators (parity tests drive the same kernels the network uses) --------------- *
 * x NCHW fp32 [n,cin,h,w]; w/b in torch layouts; optional post-activation affine (eval-BN):
 * y = act(conv(x)+b [+ resid]) * bn_scale + bn_shift.  act: 0 none, 1 ReLU, 2 LeakyReLU(0.2).
 * in_stride=2 reads x[:, :, ::2, ::2] (model.py:149-151).  Output NCHW fp32.                  */
int idc_op_conv2d(int device_id, int precision, int n, int cin, int h, int w, const float* x,
                  int cout, int ksize, int dilation, int in_stride, const float* weight,
                  const float* bias, int act, const float* bn_scale, const float* bn_shift,
                  const float* resid, float* y);
/* ConvTranspose2d(k=4, s=2, p=1) (+ residual) (+act): weight (cin,cout,4,4); y [n,cout,2h,2w].  */
int idc_op_deconv4x4s2(int device_id, int precision, int n, int cin, int h, int w, const float* x,
                       int cout, const float* weight, const float* bias, int act,
                       const float* resid, float* y);

#ifdef __cplusplus
}
#endif
#endif /* IDEEPCOLOR_H */
