// idc_kernels.h -- launch interface of the gfx950 kernels (internal; the public ABI is
// include/ideepcolor.h).
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace idc {

// One implicit-GEMM convolution launch.  The output is computed over a grid of "sites":
// site (sy,sx) reads input pixels (sy*si + tap offsets) and writes output pixel
// (sy*so + ro[phase], sx*so + co[phase]).
//   conv3x3 (dilation d, reading x[::si, ::si]) : so=1, 1 phase, 9 taps, dy,dx in {-d,0,d}
//   ConvTranspose 4x4 s2 p1                     : si=1, so=2, 4 phases x 4 taps (2x2)
//   1x1 conv                                    : 1 tap, halo 0
struct ConvArgs {
    const void* in;         // NHWC [N][Hs*si][Ws*si][Cin]          (T)
    const void* wgt;        // packed [tap][kc][cout group][64][128B] (T), see idc_layout.h
    void* out;              // NHWC [N][Hs*so][Ws*so][CoutPad]       (T, or fp32 if out_f32)
    const void* resid;      // optional NHWC (fp32, or bf16 if resid_bf16), geometry of `out`: added before the activation
    const float* bias;      // [CoutPad]
    const float* bn_scale;  // optional [CoutPad]: y = act(.)*scale + shift  (eval-BN after ReLU)
    const float* bn_shift;
    const float* img_shift; // optional fp32 [N][CoutPad]: per-image vector added after the BN affine (global hints:
                            // glob_conv4norm_rep + conv4_3norm, models/global_model/deploy_nodist.prototxt:501-518)
    int N, Hs, Ws;
    int si, so;
    int nkc;                // Cin / KC   (KC = 128 B of channels)
    int ncg;                // CoutPad / 64
    int nphase, ntaps;
    int tiles_x, tiles_y;   // tiles per image
    int act;                // 0 none, 1 ReLU, 2 LeakyReLU(0.2)
    int out_f32;
    int resid_bf16;
    // split-K (conv_igemm only; small launches = the batch-1 click path): workgroup slice `sid` of ksplit
    // runs cin chunks [sid*kc_per, min(nkc, (sid+1)*kc_per)) and stores its raw fp32 accumulators to
    // partial[sid][output pixel][CoutPad]; splitk_epilogue then sums the slices in fixed order and applies
    // bias / shortcut / activation / BN.  ksplit <= 1: off.
    int ksplit, kc_per;
    float* partial;
    // fused input pack (conv_igemm, HALO = 0, the conv1_1 launch): when pk_L != nullptr the "halo" rows are not
    // loaded from `in` but built from the reference's input planes (NCHW fp32: L [N,1,H,W], ab [N,2,H,W],
    // mask [N,1,H,W]) -- model.py:139-148 (cast, mask - maskcent, cat(L/100, ab/110, mask)) fused with the im2col
    // of conv1_1: channel tap*4 + c of a site = normalised input c at 3x3 neighbour `tap`, zero outside the image
    // and for K >= 36.
    const float* pk_L;
    const float* pk_ab;
    const float* pk_mask;
    float pk_ldiv, pk_abdiv, pk_mmul, pk_mcent;
    // fused shortcut conv (conv_igemm_v2, deconv launches only): a 3x3 conv (pad 1, bias folded into
    // `bias`) of in2 = NHWC [N][2*Hs][2*Ws][nkc2 * 64] bf16 accumulated into the same output pixels;
    // wgt2 = its layout-2 weight image (same couts).  model.py:156,170,172.
    const void* in2;
    const void* wgt2;
    int nkc2;
    // fused regression head (conv_igemm_v2<2,*> only, when one workgroup owns all 128 couts):
    // head_out[n][c][y][x] = head_mul * tanh(head_b[c] + sum_k head_w[c][k] * y[k]), y = this layer's
    // fp32 result (not stored at all).  model_out + tanh + *110, model.py:108-109,174-175.
    const float* head_w;    // [2][128] or nullptr
    const float* head_b;    // [2]
    float* head_out;        // NCHW fp32 [N][2][Hs][Ws]
    float head_mul;
    // operand-split precisions (IDC_BF16X3 / IDC_BF16X6, conv_igemm_v2s / conv_igemm_v2ps in idc_v2m.hip): an fp32 value x travels as
    // `parts` bf16 values x = hi (+ mid) + lo; a pixel of a split tensor is [part][Cpad] bf16.  The K loop runs `nseg` segments of nkc
    // chunks each; segment s multiplies input part (seg_x >> 4s & 15) with weight part (seg_w >> 4s & 15) -- 3 segments (hi.hi, lo.hi,
    // hi.lo) or 6 (+ mid.hi, hi.mid, mid.mid), all into ONE fp32 accumulator set.  Weight part p starts w_part_bytes * p after a.wgt.
    // out_parts: parts written (0 with out_f32 / a fused head).  The shortcut sum (a.resid) is fp32 in these launches.
    int in_parts, out_parts, nseg;
    unsigned seg_x, seg_w;
    unsigned long long w_part_bytes;
    unsigned long long w_part_bytes2;   // ... of the fused shortcut conv's images (a.wgt2; conv_ds_fused_m's split form)
    const float* acc_scale; // IDC_FP16X3: one fp32 in device memory, 2^-s -- the layer's weight parts hold w * 2^s (a power of two: exact) so that the lo parts
                            // of small weights are NORMAL fp16 numbers (he-style weights ~0.02: lo ~1e-5 is subnormal, 6e-8 absolute = 2^-18 of the weight);
                            // the accumulators are multiplied by it when the bias joins.  nullptr = 1.0
    int split_f16;          // IDC_FP16X3: the parts are fp16 (11-bit) instead of bf16 (8-bit) values; same planes, same segments as IDC_BF16X3
    int warm;               // != 0: the throughput kernels pull their own code into L2 at entry (idc_warm_own_code below)
    const void* zeros;      // >= 16 zero bytes in device memory: LDS-DMA source of out-of-image halo rows (conv_click)
    int dy[36], dx[36], tw[36];   // [phase*9 + t]: tap offset in sites, packed-weight tap index
    int ro[4], co[4];
};

#ifdef __HIPCC__
// First-use cost of a kernel: on some boxes the first launch of a kernel after other kernels have run is
// 25-35 % (up to 36 us) slower on EVERY forward -- the launch fetches 17-37 KB of instructions it has not executed for ~1 GB of
// traffic, line after line (or page after page) as the wave reaches them.  One wave per workgroup therefore touches every 128-byte
// line of the kernel's own code at entry -- as DATA, by LDS-DMA into `lds_scratch` (256 bytes nobody reads; no destination register
// to keep alive) -- so that the lines are on their way into this XCD's L2, and their pages walked, all at once and under the prologue's
// own memory latency.  `lines` x 128 bytes from the current PC: callers pass LESS than their kernel's code size (codeLenInByte of the
// build, profiles/r04_kernel_resources.txt lists the kernels) unless other kernels follow it in the same code object.  The requests
// retire with the prologue's first vmcnt(0).
__device__ __forceinline__ void idc_warm_own_code(char* lds_scratch, int lane, int lines) {
    const char* const pc = (const char*)(__builtin_amdgcn_s_getpc() & ~(unsigned long long)127);
    for (int j = 0; j * 64 < lines; ++j)
        if (j * 64 + lane < lines)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pc + (size_t)(j * 64 + lane) * 128),
                                             (__attribute__((address_space(3))) void*)lds_scratch, 4, 0, 0);
}
#endif

// Tuning / A-B knobs read from the environment exist only in the -DIDC_AB_PARTNERS build (make EXTRA=-DIDC_AB_PARTNERS: tools/ab_*.sh, tools/click_sweep.py,
// tools/probe_firstuse.sh); the default library reads ONE environment variable, IDC_RCCL_PATH (idc_engine.hip), and is steered through idc_set_option only.
#ifdef IDC_AB_PARTNERS
static inline int idc_env_int(const char* name, int dflt) { const char* v = getenv(name); return (v && *v) ? atoi(v) : dflt; }
#else
static inline int idc_env_int(const char*, int dflt) { return dflt; }
#endif

struct ConvConfig { int wm, wp; };     // waves along cout (x64) and along pixel rows (x4 rows of 16)

// precision: 0 fp32, 1 bf16.  halo: 0,1,2.  Returns hipSuccess or the launch error.
hipError_t launch_conv(int precision, ConvConfig cfg, int halo, const ConvArgs& a, hipStream_t s);
// Large-tile bf16 throughput kernel (32x32x16 MFMA, 8 waves, LDS-DMA weights).  cfg = {WCO, WPX}:
// {4,2} = 256 couts x (32x8 sites), {2,4} = 128 couts x (32x16 sites).  a.wgt must point at the
// layer's layout-2 weight image (idc_layout.h); tiles_x/tiles_y count 32 x 4*WPX tiles.
hipError_t launch_conv_v2(ConvConfig cfg, int halo, const ConvArgs& a, hipStream_t s);
// The same tile from v_mfma_f32_16x16x32_bf16 (idc_v2m.hip: the more energy-efficient MFMA shape on a power-capped chip).  a.wgt must
// point at the layer's LAYOUT-1 image; bf16-output launches without shortcut sum / per-image shift only (conv_v2m_applies), else
// hipErrorInvalidConfiguration.
hipError_t launch_conv_v2m(ConvConfig cfg, int halo, const ConvArgs& a, hipStream_t s);
bool conv_v2m_applies(const ConvArgs& a);
// ... and its 3x3 form without address arithmetic in the K loop (conv_igemm_v2p: padded halo rows, unrolled taps, buffer loads)
hipError_t launch_conv_v2p(ConvConfig cfg, int halo, const ConvArgs& a, hipStream_t s);
bool conv_v2p_applies(ConvConfig cfg, int halo, const ConvArgs& a);
hipError_t init_kernels_v2m();
// Operand-split form of the same tiles (conv_igemm_v2s: every geometry conv_igemm_v2m covers; conv_igemm_v2ps: the 3x3 forms of
// conv_igemm_v2p): a.in / a.out are split tensors (in_parts / out_parts), a.wgt the layout-1 images of the weight parts, shortcut sum
// fp32, fp32 output / per-image shift / LeakyReLU / fused head supported.  hipErrorInvalidConfiguration if the launch does not qualify.
hipError_t launch_conv_v2s(ConvConfig cfg, int halo, const ConvArgs& a, hipStream_t s);
hipError_t launch_conv_v2ps(ConvConfig cfg, int halo, const ConvArgs& a, hipStream_t s);
bool conv_v2s_applies(const ConvArgs& a);
bool conv_v2ps_applies(ConvConfig cfg, int halo, const ConvArgs& a);

// ConvTranspose 4x4 s2 + the 3x3 shortcut conv it is summed with, one K loop (conv_ds_fused); a.in2 / wgt2 / nkc2 = the
// shortcut's input, layout-2 weights and channel chunks, a.bias = the two biases added.  hipErrorInvalidConfiguration if
// the launch does not qualify.
hipError_t launch_conv_ds(const ConvArgs& a, hipStream_t s);
// The same launch from v_mfma_f32_16x16x32_bf16 (idc_dsm.hip: conv_ds_fused_m); a.wgt / a.wgt2 = the LAYOUT-1 images
hipError_t launch_conv_ds_m(const ConvArgs& a, hipStream_t s);
// ... and its operand-split form (conv_ds_fused_ms / _msh: split tensors in and out, fp32 accumulation of both K loops, split epilogue)
hipError_t launch_conv_ds_ms(const ConvArgs& a, hipStream_t s);
bool conv_ds_m_fits(int Hs, int Ws, int nkc, int nkc2);   // (else conv_ds_fused, which addresses with 64-bit pointers)
hipError_t init_kernels_dsm();
void set_ds_half(int v);              // 1 (default): grids with fewer 128-cout workgroups than CUs run the 64-cout, 4-wave form of conv_ds_fused_m
// model1 = conv1_1 + conv1_2 of a 32x32 tile in one workgroup (conv1_block_fused): `a` = conv1_1's arguments with conv1_2's
// riding in (wgt2 = its layout-1 weights, head_b = its bias, bn_scale/bn_shift, out = its output)
hipError_t launch_conv1_block(const ConvArgs& a, hipStream_t s);
// conv1_1 as the exact-fp32 island of an operand-split handle (conv1_1_split_kernel: fp32 MFMA straight from the input patch, a.out_parts planes out)
hipError_t launch_conv1_1_split(const ConvArgs& a, hipStream_t s);
// conv1_2 of an operand-split handle on conv1_block_fused_t's conv1_2 tile (conv1_2_split_kernel: 32 x 12 pixels, two workgroups per CU, segments over a static halo)
hipError_t launch_conv1_2_split(const ConvArgs& a, hipStream_t s);
// conv1_1 (4 -> 64, input pack fused) as one 32x32 tile per workgroup, bf16; hipErrorInvalidConfiguration if the
// launch does not qualify (the caller then uses launch_conv)
hipError_t launch_conv1_1_bf16(const ConvArgs& a, hipStream_t s);
// Batch-1 click-path kernel (conv_click): tile 16 x 4*wp sites x 64 couts, the workgroup's whole K slice (kc_per chunks x
// all taps) requested by LDS-DMA at entry; a.kc_per <= conv_click_max_chunks(wp, halo, ntaps), a.ksplit = ceil(nkc / kc_per),
// a.tiles_x / tiles_y count 16 x 4*wp tiles, a.zeros set.  Layout-1 weights.
hipError_t launch_conv_click(int precision, int wp, int halo, const ConvArgs& a, hipStream_t s);
int conv_click_max_chunks(int wp, int halo, int ntaps);
// Second half of a split-K launch: out = act(sum_s partial[s] + bias [+ resid]) [* bn_scale + bn_shift] [+ img_shift]
// over all N*Hout*Wout*CoutPad outputs (geometry and epilogue fields taken from the same ConvArgs).
hipError_t launch_splitk_epilogue(int precision, const ConvArgs& a, hipStream_t s);
// fp32 Winograd F(2x2,3x3) form of a 3x3 stride-1 conv (idc_wino.hip): a.wgt = the layer's U image, a.out fp32,
// a.dy[8] = dilation (1 | 2); hipErrorInvalidConfiguration if the launch does not qualify
hipError_t launch_conv_wino(int precision, const ConvArgs& a, hipStream_t s);     // precision 1: conv_wino_bf16 (batch-1 click path only)
// fp32 ConvTranspose 4x4 s2 as Winograd F(2x2,2x2) over its four phases: a.wgt = the 36-position U image, a.Hs / a.Ws = input size
hipError_t launch_deconv_wino(int precision, const ConvArgs& a, hipStream_t s);    // precision 1: the bf16 twin (click path)
// the launch guards of the two entry points above as a predicate (args filled in: in / wgt / zeros / resid / out_f32 set), and the
// 32-bit source-offset bound alone (known before the pointers are: set_geometry)
bool conv_wino_applies(int precision, const ConvArgs& a, bool deconv);
// bf16 click path (idc_kw.hip): a 3x3 stride-1 conv with K split over the waves of a workgroup, no reduction launch; a.wgt = the
// layer's LAYOUT-1 image, a.dy[8] = dilation, a.zeros set; hipErrorInvalidConfiguration if the launch does not qualify
hipError_t launch_conv_kwave(const ConvArgs& a, hipStream_t s);
bool conv_kwave_applies(const ConvArgs& a);
hipError_t init_kernels_kw();
// conv_kwave_chain_bf16 (idc_kw.hip, round 5): a run of consecutive same-shape 8-chunk conv_kwave_bf16 layers (the 512 -> 512 trunk of
// the bf16 click path: conv4_2 .. conv7_3 at batch 1) as ONE persistent launch, a grid barrier between layers instead of a launch floor.
struct KwChainLayer {
    const void* in; void* out; const void* wgt; const float* bias; const float* bn_scale; const float* bn_shift;
    int act;                // 0 none, 1 ReLU, 2 LeakyReLU(0.2)
    int d;                  // dilation 1 | 2
};
constexpr int kKwChainMax = 12;
struct KwChainArgs {
    int nlayers, H, W, N, ncg;
    unsigned spin_limit;            // polls of the barrier counters before a workgroup gives up (sets *abort_flag, leaves)
    unsigned long long* bar;        // eight device counters 128 bytes apart (one per blockIdx & 7), monotone: + its arrivals per barrier
    unsigned long long bar_base;    // barriers done by every earlier launch of the handle (counter c then holds bar_base x arrivals_of(c))
    int* abort_flag;                // host-visible (pinned) int: set to 1 by a workgroup that gave up
    long long* stamps;              // diagnostic (IDC_KW_STAMPS=1), else null: [workgroup][layer][8] cycle stamps of the phases of a layer
    KwChainLayer layer[kKwChainMax];
};
// workgroups of one layer (the same for every layer of a chain: 8x8-pixel tiles x parities x images x 32-cout groups), 0 if the shape is not covered
int conv_kwave_chain_blocks(int H, int W, int N, int ncg, int d);
// workgroups of the chain kernel the device holds at once (one per CU: 104 KiB of LDS each); 0 on error
int conv_kwave_chain_capacity(int device);
// mode 1: hipLaunchCooperativeKernel (the runtime guarantees co-residency or fails cleanly); mode 2: plain launch (caller checked the capacity)
hipError_t launch_conv_kwave_chain(const KwChainArgs& c, int blocks, int mode, hipStream_t s);
bool wino_offsets_fit(int Hs, int Ws, int si, int nkc);
hipError_t init_kernels_wino();
void set_wino_form(int form);      // 0 = by grid size, 12 / 21 / 22 = force conv_wino_f32<TB,CB> (speed only)
// One-time: raise the dynamic-LDS limit of every conv instantiation.
hipError_t init_kernels();
size_t conv_lds_bytes(ConvConfig cfg, int halo);

// conv1x1(128->2) + tanh + *out_mul, NHWC (T) -> NCHW fp32
hipError_t launch_head(int precision, const void* x, const float* w, const float* b, float* out,
                       int N, int H, int W, float out_mul, hipStream_t s);
// softmax over the first `nclass` of `cstride` fp32 logits per pixel, * temperature first;
// NHWC fp32 logits [npix][cstride] -> NCHW fp32 probabilities [N][nclass][H][W]
hipError_t launch_softmax_nchw(const float* logits, float* out, int N, int H, int W, int nclass,
                               int cstride, float temperature, hipStream_t s);
// 313-bin head tail (models/reference_model/deploy_nopred.prototxt:776-850): logits fp32 NHWC [N][H/4][W/4][cstride]
// (313 valid) -> the two grouped bilinear x2 deconvs (fixed kernel, colorize_image.py:410-413; per axis
// out[4m+j] = ((4-j) in[m] + j in[m+1]) / 4 with in[M] = 0) -> per full-resolution pixel
//   dist_S [N][313][H][W] = softmax(S * l)  (optional, may be nullptr)
//   pred_ab [N][2][H][W]  = sum_q softmax(2.6 * l)_q * w_ab[c][q] + b_ab[c]
// w_ab: fp32 [2][313] then [2] bias.
hipError_t launch_dist313(const float* logits, const float* w_ab, float* pred_ab, float* dist_S, int N, int H, int W,
                          int cstride, float S, float T, hipStream_t s);

// Global-hints branch (models/global_model/deploy_nodist.prototxt:37-172): per image
//   y1 = BN1(relu(Wg g + bg + Ws s + bs)),  y_{i+1} = BN_{i+1}(relu(W_{i+1} y_i + b_{i+1})), i = 1..3  ->  out [N][512]
// in [N][316] = 314 histogram+flag values then the 2 saturation values; params = the packed fp32 section
// described at glob_param_floats() (transposed weights [k][512], then bias / BN scale / BN shift).
hipError_t launch_glob_branch(const float* in, const float* params, float* out, int N, hipStream_t s);
constexpr int kGlobIn = 316, kGlobC = 512;
constexpr size_t glob_param_floats() { return (size_t)kGlobIn * kGlobC + 3 * kGlobC + 3 * ((size_t)kGlobC * kGlobC + 3 * kGlobC); }

// Lab -> sRGB uint8 (+ optional rgb -> Lab refresh), float64 like skimage (colorize_image.py:20-28,31-36):
// L [N,1,H,W] fp32 (+ l_add), ab [N,2,H,W] fp32 -> rgb [N,H,W,3] u8, lab_q [N,3,H,W] f64 (or nullptr)
hipError_t launch_pcie_copy(void* dst, const void* src, size_t bytes, hipStream_t s);   // small host <-> device copies as a kernel on the forward's stream
hipError_t launch_lab_post(const float* L, float l_add, const float* ab, unsigned char* rgb, double* lab_q, int N,
                           int H, int W, hipStream_t s);

// Display / full-resolution step (ui/gui_draw.py:280-283, colorize_image.py:123-158): (a, b) planes [H,W] (fp32 or fp64)
// resized to [oh,ow] -- interp 0 cv2 INTER_CUBIC, 1 scipy zoom order 1, 2 scipy zoom order 0 -- then Lab -> sRGB uint8
// with L_out [oh,ow] (float64) -> rgb [oh,ow,3]
hipError_t launch_upsample_lab2rgb(const void* a_plane, const void* b_plane, int src_f64, int H, int W, int interp, const double* L_out,
                                   int oh, int ow, unsigned char* rgb, hipStream_t s);

// Global statistics (global_stats.prototxt): rgb u8 [N,H,W,3] -> counts [N][313] (uint32, zeroed by the caller) of
// the 4x4-pooled ab values' nearest centre, and sat_sum [N] (float64, zeroed) = sum of HSV saturation over pixels.
hipError_t launch_global_stats(const unsigned char* rgb, const float* centres, unsigned* counts, double* sat_sum, int N,
                               int H, int W, hipStream_t s);

// Click session (idc_session.hip).  HintRect = idc_hint of include/ideepcolor.h after clipping: inclusive rectangle,
// (c0,c1) = ab (mode 0) or (c0,c1,c2) = RGB 0..255 (mode 1).  raster_hints fills ab [2,H,W] and mask [1,H,W] of ONE
// image: the last covering hint wins; uncovered pixels get ab = 0, mask = 0 (a black canvas is Lab (0,0,0)).
struct HintRect { int y0, x0, y1, x1; float c0, c1, c2; };
hipError_t launch_raster_hints(const HintRect* hints, int n_hints, int mode, float mask_value, float* ab, float* mask,
                               int H, int W, hipStream_t s);

// Colour suggestions at one pixel (get_ab_reccs, colorize_image.py:322-354): see idc_session.hip for the algorithm.
// pdf bin b at pdf[b*stride]; centres [B][2]; out_centres [K][2], out_conf [K] (f64); out_counts [B] or nullptr.
constexpr int kSuggestMaxBins = 1024, kSuggestMaxK = 16;
hipError_t launch_suggest(const float* pdf, long long stride, int B, const float* centres, int K, int N, unsigned seed,
                          double* out_centres, double* out_conf, unsigned* out_counts, hipStream_t s);

// layout converters for the single-operator test entry points and idc_get_activation
hipError_t launch_nchw_to_nhwc(int precision, const float* src, void* dst, int N, int C, int H, int W,
                               int Cpad, hipStream_t s);
hipError_t launch_nhwc_to_nchw(int src_is_bf16, const void* src, float* dst, int N, int C, int H,
                               int W, int Cstride, hipStream_t s);
// ... of a split tensor: dst = sum over the `parts` bf16 planes of a pixel (fp32 sum, hi first)
hipError_t launch_split_to_nchw(const void* src, float* dst, int N, int C, int H, int W, int Cpad, int parts, int f16, hipStream_t s);
// fp32 NCHW -> split NHWC (single-operator test entry points)
hipError_t launch_nchw_to_split(const float* src, void* dst, int N, int C, int H, int W, int Cpad, int parts, int f16, hipStream_t s);

}  // namespace idc
