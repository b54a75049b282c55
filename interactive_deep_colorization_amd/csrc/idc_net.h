// idc_net.h -- the static description of the SIGGRAPHGenerator graph (models/pytorch/model.py)
// and of the packed weight blob.  Host-only; shared by the packer and the executor.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace idc {

enum LayerKind { kConv3x3 = 0, kDeconv4x4 = 1, kConv1x1 = 2, kConvIm2col = 3 };

// One conv-like layer of the reference graph.  `name` follows the Caffe prototxt naming
// (models/reference_model/deploy_nodist.prototxt) so both backends of the reference map onto it.
struct LayerSpec {
    const char* name;       // tensor it produces
    const char* wkey;       // state_dict prefix, e.g. "model1.0"   
    const char* bnkey;      // eval-BN applied AFTER the activation (model.py:17 ...), or nullptr
    LayerKind kind;
    int cin, cout;          // real channel counts (conv1_1: cin = 4)
    int dilation;           // 1 or 2 (model5/model6: model.py:48-63)
    int in_stride;          // 2 = reads src[:, :, ::2, ::2] (model.py:149-151)
    int act;                // 0 none, 1 ReLU, 2 LeakyReLU(0.2)
    const char* src;        // input tensor name
    const char* resid;      // fp32 tensor summed before the activation (model.py:156,170,172) or nullptr
    int out_f32;            // keep the output in fp32 (shortcut branches, class logits)
    int level;              // output resolution = H / level
    int dist_only;          // 1: only built with IDC_FLAG_DIST_HEAD; 2: only with IDC_FLAG_DIST313
};

inline const std::vector<LayerSpec>& layer_specs() {
    static const std::vector<LayerSpec> specs = {
        // name            wkey              bnkey        kind         cin  cout d si act src              resid            f32 lvl dist
        {"conv1_1",       "model1.0",        nullptr,     kConvIm2col,   4,  64, 1, 1, 1, "data_l_ab_mask", nullptr,         0, 1, 0},
        {"conv1_2",       "model1.2",        "model1.4",  kConv3x3,     64,  64, 1, 1, 1, "conv1_1",        nullptr,         0, 1, 0},
        {"conv2_1",       "model2.0",        nullptr,     kConv3x3,     64, 128, 1, 2, 1, "conv1_2",        nullptr,         0, 2, 0},
        {"conv2_2",       "model2.2",        "model2.4",  kConv3x3,    128, 128, 1, 1, 1, "conv2_1",        nullptr,         0, 2, 0},
        {"conv3_1",       "model3.0",        nullptr,     kConv3x3,    128, 256, 1, 2, 1, "conv2_2",        nullptr,         0, 4, 0},
        {"conv3_2",       "model3.2",        nullptr,     kConv3x3,    256, 256, 1, 1, 1, "conv3_1",        nullptr,         0, 4, 0},
        {"conv3_3",       "model3.4",        "model3.6",  kConv3x3,    256, 256, 1, 1, 1, "conv3_2",        nullptr,         0, 4, 0},
        {"conv4_1",       "model4.0",        nullptr,     kConv3x3,    256, 512, 1, 2, 1, "conv3_3",        nullptr,         0, 8, 0},
        {"conv4_2",       "model4.2",        nullptr,     kConv3x3,    512, 512, 1, 1, 1, "conv4_1",        nullptr,         0, 8, 0},
        {"conv4_3",       "model4.4",        "model4.6",  kConv3x3,    512, 512, 1, 1, 1, "conv4_2",        nullptr,         0, 8, 0},
        {"conv5_1",       "model5.0",        nullptr,     kConv3x3,    512, 512, 2, 1, 1, "conv4_3",        nullptr,         0, 8, 0},
        {"conv5_2",       "model5.2",        nullptr,     kConv3x3,    512, 512, 2, 1, 1, "conv5_1",        nullptr,         0, 8, 0},
        {"conv5_3",       "model5.4",        "model5.6",  kConv3x3,    512, 512, 2, 1, 1, "conv5_2",        nullptr,         0, 8, 0},
        {"conv6_1",       "model6.0",        nullptr,     kConv3x3,    512, 512, 2, 1, 1, "conv5_3",        nullptr,         0, 8, 0},
        {"conv6_2",       "model6.2",        nullptr,     kConv3x3,    512, 512, 2, 1, 1, "conv6_1",        nullptr,         0, 8, 0},
        {"conv6_3",       "model6.4",        "model6.6",  kConv3x3,    512, 512, 2, 1, 1, "conv6_2",        nullptr,         0, 8, 0},
        {"conv7_1",       "model7.0",        nullptr,     kConv3x3,    512, 512, 1, 1, 1, "conv6_3",        nullptr,         0, 8, 0},
        {"conv7_2",       "model7.2",        nullptr,     kConv3x3,    512, 512, 1, 1, 1, "conv7_1",        nullptr,         0, 8, 0},
        {"conv7_3",       "model7.4",        "model7.6",  kConv3x3,    512, 512, 1, 1, 1, "conv7_2",        nullptr,         0, 8, 0},
        {"conv3_3_short", "model3short8.0",  nullptr,     kConv3x3,    256, 256, 1, 1, 0, "conv3_3",        nullptr,         0, 4, 0},
        {"conv8_1",       "model8up.0",      nullptr,     kDeconv4x4,  512, 256, 1, 1, 1, "conv7_3",        "conv3_3_short", 0, 4, 0},
        {"conv8_2",       "model8.1",        nullptr,     kConv3x3,    256, 256, 1, 1, 1, "conv8_1",        nullptr,         0, 4, 0},
        {"conv8_3",       "model8.3",        "model8.5",  kConv3x3,    256, 256, 1, 1, 1, "conv8_2",        nullptr,         0, 4, 0},
        {"class_logits",  "model_class.0",   nullptr,     kConv1x1,    256, 529, 1, 1, 0, "conv8_3",        nullptr,         1, 4, 1},
        {"conv2_2_short", "model2short9.0",  nullptr,     kConv3x3,    128, 128, 1, 1, 0, "conv2_2",        nullptr,         0, 2, 0},
        {"conv9_1",       "model9up.0",      nullptr,     kDeconv4x4,  256, 128, 1, 1, 1, "conv8_3",        "conv2_2_short", 0, 2, 0},
        {"conv9_2",       "model9.1",        "model9.3",  kConv3x3,    128, 128, 1, 1, 1, "conv9_1",        nullptr,         0, 2, 0},
        {"conv1_2_short", "model1short10.0", nullptr,     kConv3x3,     64, 128, 1, 1, 0, "conv1_2",        nullptr,         0, 1, 0},
        {"conv10_1",      "model10up.0",     nullptr,     kDeconv4x4,  128, 128, 1, 1, 1, "conv9_2",        "conv1_2_short", 0, 1, 0},
        {"conv10_2",      "model10.1",       nullptr,     kConv3x3,    128, 128, 1, 1, 2, "conv10_1",       nullptr,         0, 1, 0},
        // ---- 313-bin distribution head (models/reference_model/deploy_nopred.prototxt:650-775; IDC_FLAG_DIST313) ----
        // hyper-column sum conv3_pred + conv4..7_pred + conv8_pred (Eltwise SUM :747-757) kept in fp32 and chained
        // through the shortcut-sum input of each launch; the ReLU (:758-763) rides on the last one
        {"conv3_pred",    "pred.conv3_pred", nullptr,     kConv3x3,    256, 384, 1, 1, 0, "conv3_3",        nullptr,         1, 4, 2},
        {"conv34_pred",   "pred.conv4_pred", nullptr,     kDeconv4x4,  512, 384, 1, 1, 0, "conv4_3",        "conv3_pred",    1, 4, 2},
        {"conv345_pred",  "pred.conv5_pred", nullptr,     kDeconv4x4,  512, 384, 1, 1, 0, "conv5_3",        "conv34_pred",   1, 4, 2},
        {"conv3456_pred", "pred.conv6_pred", nullptr,     kDeconv4x4,  512, 384, 1, 1, 0, "conv6_3",        "conv345_pred",  1, 4, 2},
        {"conv34567_pred","pred.conv7_pred", nullptr,     kDeconv4x4,  512, 384, 1, 1, 0, "conv7_3",        "conv3456_pred", 1, 4, 2},
        {"conv345678_pred","pred.conv8_pred",nullptr,     kConv3x3,    256, 384, 1, 1, 1, "conv8_3",        "conv34567_pred",0, 4, 2},
        {"pred_313",      "pred.pred_313",   nullptr,     kConv1x1,    384, 313, 1, 1, 0, "conv345678_pred", nullptr,        1, 4, 2},
    };
    return specs;
}

inline int cout_pad(int cout) { return cout <= 64 ? 64 : ((cout + 127) / 128) * 128; }
inline int weight_taps(LayerKind k) { return k == kConv3x3 ? 9 : (k == kDeconv4x4 ? 16 : 1); }
// channels of the GEMM K dimension per tap, before padding to the 128-byte chunk
inline int k_channels(const LayerSpec& s) { return s.kind == kConvIm2col ? 36 : s.cin; }
// layers the Winograd kernels (idc_wino.hip) can run: 3x3 (the conv itself always has stride 1; in_stride 2 = it reads a strided
// view of its source), no shortcut sum, 32 | Cin
inline bool wino_eligible(const LayerSpec& s) { return s.kind == kConv3x3 && s.resid == nullptr && s.cin % 32 == 0; }
// fp32 deconv layers run as Winograd F(2x2,2x2) over their four phases (conv_wino_deconv_f32): 36 transformed values per (cin, cout)
inline bool wino_deconv_eligible(const LayerSpec& s) { return s.kind == kDeconv4x4 && s.cin % 32 == 0; }
// layers the bf16 large-tile kernel (conv_igemm_v2, >= 128 couts per workgroup) can run
inline bool v2_eligible(const LayerSpec& s) { return s.kind != kConvIm2col && cout_pad(s.cout) >= 128; }
// operand-split precisions: conv1_1 (K = 36 im2col of the four input planes, 0.2 % of the MACs) is an exact-fp32 "island" -- fp32 weights, the
// fp32 small-tile kernel -- whose epilogue writes its result as split planes; every other layer runs the large tile (conv1_2: its 64-cout form)
inline bool split_island(const LayerSpec& s) { return s.kind == kConvIm2col; }

// Where one layer's parameters live inside the packed blob (byte offsets).
struct LayerBlob {
    size_t w_off, w_bytes;        // [tap][kc][cg][64][128B]
    size_t w2_off;                // same size, layout 2 (bf16 large-tile kernel) or (size_t)-1
    size_t w3_off, w3_bytes;      // fp32 only: Winograd F(2x2,3x3) image U = G g G^T, [chunk][pos 16][cout/16][ks 2][64 lanes][16 B]
                                  // (idc_wino.hip), or (size_t)-1 / 0
    size_t bias_off;              // fp32 [cout_pad]
    size_t bn_scale_off, bn_shift_off;   // fp32 [cout_pad] or (size_t)-1
    size_t fbias_off;             // layers with a shortcut sum: fp32 [cout_pad] = bias + the shortcut conv's bias, or (size_t)-1
    int nkc, ncg;
    size_t wscale_off = (size_t)-1;   // operand-split precisions (not the fp32 island): ONE fp32 = 2^-s, the factor that takes this layer's accumulators back
                                  // from weights packed as w * 2^s (IDC_FP16X3: s chosen so that max|w| * 2^s is in [8192, 16384); bf16 parts: s = 0, 1.0)
    int parts = 1;                // operand-split precisions: weight parts (hi, [mid,] lo), each w_bytes long, contiguous from w_off
    int f32 = 0;                  // operand-split precisions: this layer belongs to the fp32 island (fp32 images, fp32 chunk size)
};

struct BlobPlan {
    int precision;
    unsigned flags;
    std::vector<LayerBlob> layers;       // parallel to the ACTIVE layer list
    std::vector<int> active;             // indices into layer_specs()
    size_t head_w_off, head_b_off;       // model_out.0: fp32 [2][128], [2]
    size_t pred_ab_off;                  // pred_ab 1x1 decode: fp32 [2][313] weights then [2] bias, or (size_t)-1
    size_t glob_off;                     // global-hints branch parameters (fp32, glob_param_floats()) or (size_t)-1
    size_t total_bytes;
};

constexpr uint32_t kBlobMagic = 0x43444931u;   // "1IDC"
struct BlobHeader {
    uint32_t magic, version, precision, flags;
    uint64_t total_bytes;
    uint64_t checksum;                   // FNV-1a over the payload after the header
    uint8_t pad[32];
};
static_assert(sizeof(BlobHeader) == 64, "blob header is 64 bytes");

BlobPlan make_blob_plan(int precision, unsigned flags);

}  // namespace idc
