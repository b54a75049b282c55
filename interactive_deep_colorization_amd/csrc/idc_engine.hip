// idc_engine.hip -- host side of libideepcolor_hip.so: weight packer, static-schedule executor of
// the SIGGRAPHGenerator graph, and the C ABI declared in include/ideepcolor.h.
//
// Replaces: SIGGRAPHGenerator.__init__/forward (models/pytorch/model.py:6-175) and the
// load_state_dict/eval part of ColorizeImageTorch.prep_net (data/colorize_image.py:216-233).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <algorithm>
#include <vector>
#include <atomic>
#include <functional>
#include <thread>

#include "../../include/ideepcolor.h"
#include "idc_kernels.h"
#include "idc_layout.h"
#include "idc_net.h"

namespace idc {

static thread_local std::string g_last_error;

// The 32x32x16-MFMA partners of the throughput kernels (conv_igemm_v2, conv_ds_fused, conv1_1_bf16_kernel: idc_v2.hip, idc_conv1.hip) exist only in the
// -DIDC_AB_PARTNERS build (round 6).  The default library plans every launch on conv_igemm_v2p / conv_igemm_v2m / conv_ds_fused_m / conv1_block_fused_t or
// the small-tile kernels, and refuses the option values that ask for a partner.
#ifdef IDC_AB_PARTNERS
static constexpr bool kAbPartners = true;
#else
static constexpr bool kAbPartners = false;
#endif

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

BlobPlan make_blob_plan(int precision, unsigned flags) {
    BlobPlan p;
    p.precision = precision;
    p.flags = flags & (IDC_FLAG_DIST_HEAD | IDC_FLAG_GLOBAL_HINTS | IDC_FLAG_DIST313 | IDC_FLAG_THROUGHPUT_BLOB);
    // Winograd U images: the fp32 path only (round 5: the bf16 click path's Winograd kernels were retired -- conv_kwave_* read the layout-1
    // images -- so a bf16 blob is 70 MB (136 MB with the partner build's layout-2 images) whatever the flag says; fp32: 384 MB, 136 MB with
    // IDC_FLAG_THROUGHPUT_BLOB)
    const bool wino_images = !(flags & IDC_FLAG_THROUGHPUT_BLOB) && precision == IDC_FP32;
    const auto& specs = layer_specs();
    size_t off = sizeof(BlobHeader);
    for (int i = 0; i < (int)specs.size(); ++i) {
        const LayerSpec& s = specs[i];
        if (s.dist_only == 1 && !(flags & IDC_FLAG_DIST_HEAD)) continue;
        if (s.dist_only == 2 && !(flags & IDC_FLAG_DIST313)) continue;
        LayerBlob lb;
        // operand-split precisions: model1 is an fp32 island (fp32 images incl. conv1_2's Winograd image), every other layer carries
        // split_parts() layout-1 bf16 images
        lb.f32 = is_split(precision) && split_island(s);
        lb.parts = (is_split(precision) && !lb.f32) ? split_parts(precision) : 1;
        const int lprec = lb.f32 ? (int)IDC_FP32 : precision;
        const int kc = kc_elems(lprec);
        const bool wino_l = lb.f32 ? true : wino_images;
        const int kch = k_channels(s);
        lb.nkc = (s.kind == kConvIm2col) ? (64 / kc) : (kch + kc - 1) / kc;   // conv1_1 operand is 64 wide
        lb.ncg = cout_pad(s.cout) / kCoutGroup;
        lb.w_bytes = (size_t)weight_taps(s.kind) * lb.nkc * lb.ncg * kWBlockBytes;
        off = align_up(off, 256); lb.w_off = off; off += lb.w_bytes * lb.parts;
        lb.w2_off = (size_t)-1;
        // (layout 2 = the 32x32x16-MFMA kernels' image: partner build only; the default library's bf16 blob is 70 MB instead of 136)
        if (kAbPartners && precision == IDC_BF16 && v2_eligible(s)) { off = align_up(off, 256); lb.w2_off = off; off += lb.w_bytes; }
        // IDC_FP16: conv1_1 also as ONE fp16 layout-1 block (K = 36 in a 64-wide chunk) -- what conv1_block_fused_th reads; the fp32 island image above stays
        // for the launches the block does not take
        if (precision == IDC_FP16 && s.kind == kConvIm2col) { off = align_up(off, 256); lb.w2_off = off; off += kWBlockBytes; }
        lb.w3_off = (size_t)-1; lb.w3_bytes = 0;
        if (wino_l && wino_eligible(s) && s.cin % kc == 0) {                            // fp32: every batch size; bf16: the batch-1 click path
            lb.w3_bytes = (size_t)s.cin * cout_pad(s.cout) * 16 * elem_bytes(lprec);   // 16 transformed values per (cin, cout)
            off = align_up(off, 256); lb.w3_off = off; off += lb.w3_bytes;
        }
        if (wino_l && wino_deconv_eligible(s) && s.cin % kc == 0) {                      // deconvs: F(2x2,2x2) over the four phases (click path)
            lb.w3_bytes = (size_t)s.cin * cout_pad(s.cout) * 36 * elem_bytes(lprec);
            off = align_up(off, 256); lb.w3_off = off; off += lb.w3_bytes;
        }
        off = align_up(off, 256); lb.bias_off = off; off += (size_t)cout_pad(s.cout) * 4;
        if (s.bnkey) {
            off = align_up(off, 256); lb.bn_scale_off = off; off += (size_t)cout_pad(s.cout) * 4;
            off = align_up(off, 256); lb.bn_shift_off = off; off += (size_t)cout_pad(s.cout) * 4;
        } else {
            lb.bn_scale_off = lb.bn_shift_off = (size_t)-1;
        }
        lb.fbias_off = (size_t)-1;
        if (s.resid) { off = align_up(off, 256); lb.fbias_off = off; off += (size_t)cout_pad(s.cout) * 4; }
        if (is_split(precision) && !lb.f32) { off = align_up(off, 256); lb.wscale_off = off; off += 4; }     // 2^-s of weights packed as w * 2^s (IDC_FP16X3; else 1.0)
        p.layers.push_back(lb);
        p.active.push_back(i);
    }
    off = align_up(off, 256); p.head_w_off = off; off += 2 * 128 * 4;
    off = align_up(off, 256); p.head_b_off = off; off += 2 * 4;
    p.pred_ab_off = (size_t)-1;
    if (flags & IDC_FLAG_DIST313) { off = align_up(off, 256); p.pred_ab_off = off; off += (2 * 313 + 2) * 4; }
    p.glob_off = (size_t)-1;
    if (flags & IDC_FLAG_GLOBAL_HINTS) { off = align_up(off, 256); p.glob_off = off; off += glob_param_floats() * 4; }
    p.total_bytes = align_up(off, 256);
    return p;
}

static uint64_t fnv1a(const uint8_t* p, size_t n) {
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

struct TensorView {
    const float* data = nullptr;
    int ndim = 0;
    int64_t dims[4] = {0, 0, 0, 0};
};

static bool dims_are(const TensorView& t, std::initializer_list<int64_t> d) {
    if (t.ndim != (int)d.size()) return false;
    int i = 0;
    for (int64_t v : d) if (t.dims[i++] != v) return false;
    return true;
}

// Write one element of the packed weight image (layout 1: small-tile kernels, layout 2: conv_igemm_v2).
// part: operand-split precisions -- which part of the weight is stored (0 = hi: rne(v); 1: rne(v - hi); 2: rne(v - hi - mid))
static inline void put_w(uint8_t* wimg, int precision, int layout, int nkc, int ncg, int tw, int co, int k, float v, int part, float wmul = 1.f) {
    v *= wmul;                                                     // (a power of two: exact)
    const int kc_e = kc_elems(precision), eb = elem_bytes(precision), eps = kSlotBytes / eb;
    const int kc = k / kc_e, kin = k % kc_e;
    const int s = kin / eps, e = kin % eps;
    const int cg = co / kCoutGroup, col = co % kCoutGroup;
    int lam, sig;
    if (layout == 2) {
        lam = cg_cout_to_row2(col);
        sig = s ^ swz2(lam);
    } else {
        // inverse of cg_row_to_cout: col = g*16 + ci*4 + reg  ->  lam = ci*16 + g*4 + reg
        const int gq = col >> 4, ci = (col >> 2) & 3, reg = col & 3;
        lam = ci * 16 + gq * 4 + reg;
        sig = s ^ swz(lam);
    }
    const size_t off = ((size_t)(tw * nkc + kc) * ncg + cg) * kWBlockBytes + (size_t)lam * kRowBytes +
                       (size_t)sig * kSlotBytes + (size_t)e * eb;
    if (split_is_f16(precision)) {                                 // IDC_FP16X3: fp16 parts (RNE; weights beyond the fp16 range saturate)
        auto to_h = [](float x) { return (_Float16)(x > 65504.f ? 65504.f : (x < -65504.f ? -65504.f : x)); };
        _Float16 b = to_h(v);
        for (int q = 0; q < part; ++q) { v -= (float)b; b = to_h(v); }
        memcpy(wimg + off, &b, 2);
    } else if (precision != IDC_FP32) {
        uint16_t b = f32_to_bf16_rne(v);
        for (int q = 0; q < part; ++q) {                           // (exact: the remainder of a round-to-nearest is representable)
            uint32_t u = (uint32_t)b << 16; float hi; memcpy(&hi, &u, 4);
            v -= hi;
            b = f32_to_bf16_rne(v);
        }
        memcpy(wimg + off, &b, 2);
    } else {
        memcpy(wimg + off, &v, 4);
    }
}

// Pack one conv-like layer: weights in torch layout -> MFMA-tiled, swizzled image.
// IDC_FP16X3: the power of two s that brings max|w| into [8192, 16384) -- hi = rne16(w 2^s) uses fp16's top binades, lo = rne16(w 2^s - hi) is a NORMAL fp16
// number down to weights 2^-17 of the largest; unscaled, he-style weights (~0.02) have lo parts ~1e-5, below fp16's smallest normal 6.1e-5, and keep only
// 6e-8 absolute = 2^-18 of the weight (measured, oracle/emulate.py + tools/split_study.py: N = 1 he-style 2.7e-3 -> 9.5e-4 on the ab map, fp32 arithmetic 1.2e-3)
static int f16_weight_exponent(const float* w, size_t n) {
    float mx = 0.f;
    for (size_t i = 0; i < n; ++i) { const float a = fabsf(w[i]); if (a > mx && a < INFINITY) mx = a; }
    if (mx == 0.f) return 0;
    int e; (void)frexpf(mx, &e);                                   // mx = m 2^e, m in [0.5, 1)
    int s = 14 - e;                                                // mx 2^s in [8192, 16384)
    return s < -10 ? -10 : (s > 40 ? 40 : s);
}

static void pack_layer_weights(uint8_t* wimg, int precision, int layout, const LayerSpec& s, const LayerBlob& lb,
                               const float* w, int part = 0, float wmul = 1.f) {
    memset(wimg, 0, lb.w_bytes);
    const int cin = s.cin, cout = s.cout;
    if (s.kind == kConv3x3) {
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci)
                for (int t = 0; t < 9; ++t)
                    put_w(wimg, precision, layout, lb.nkc, lb.ncg, t, co, ci, w[((size_t)co * cin + ci) * 9 + t], part, wmul);
    } else if (s.kind == kConvIm2col) {          // K index = tap*4 + c  (the order conv1_1's fused input pack builds)
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci)
                for (int t = 0; t < 9; ++t)
                    put_w(wimg, precision, layout, lb.nkc, lb.ncg, 0, co, t * 4 + ci, w[((size_t)co * cin + ci) * 9 + t], part, wmul);
    } else if (s.kind == kConv1x1) {
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci)
                put_w(wimg, precision, layout, lb.nkc, lb.ncg, 0, co, ci, w[(size_t)co * cin + ci], part, wmul);
    } else {                                      // ConvTranspose2d weight is (Cin, Cout, 4, 4)
        for (int ci = 0; ci < cin; ++ci)
            for (int co = 0; co < cout; ++co)
                for (int t = 0; t < 16; ++t)
                    put_w(wimg, precision, layout, lb.nkc, lb.ncg, t, co, ci, w[((size_t)ci * cout + co) * 16 + t], part, wmul);
    }
}

// Winograd F(2x2,3x3) weight image of one 3x3 layer (fp32 path, idc_wino.hip): U = G g G^T per (cout, cin) in float64,
// G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], stored in MFMA A-operand order
//   [chunk = ci/32][pos = i*4+j][cout block = co/16][ks][lane = g*16 + co%16][e],   ci%32 = (ks*4 + g)*4 + e
// so that one wave-wide 16-byte load is the fragment of (pos, 16 couts, 16 cin).
// bf16: chunk = ci/64, ci%64 = (ks*4 + g)*8 + e, 8 bf16 per lane; U rounded to bf16 once, from the float64 transform.
static void pack_wino_weights(uint8_t* img, int precision, const LayerSpec& s, const LayerBlob& lb, const float* w) {
    memset(img, 0, lb.w3_bytes);
    static const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
    const int ncb = cout_pad(s.cout) / 16;
    const int kc = kc_elems(precision), eps = kSlotBytes / elem_bytes(precision);     // channels per chunk, elements per 16-byte slot
    float* const out = (float*)img;
    uint16_t* const out16 = (uint16_t*)img;
    for (int co = 0; co < s.cout; ++co)
        for (int ci = 0; ci < s.cin; ++ci) {
            const float* g = w + ((size_t)co * s.cin + ci) * 9;
            double t[4][3];
            for (int i = 0; i < 4; ++i)
                for (int kx = 0; kx < 3; ++kx) t[i][kx] = G[i][0] * g[0 * 3 + kx] + G[i][1] * g[1 * 3 + kx] + G[i][2] * g[2 * 3 + kx];
            const int c = ci / kc, within = ci % kc, slot = within / eps, e = within % eps, ks = slot / 4, gq = slot % 4;
            const int cbg = co / 16, m = co % 16, lane = gq * 16 + m;
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    const double u = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
                    const size_t idx = (((((size_t)c * 16 + (i * 4 + j)) * ncb + cbg) * 2 + ks) * 64 + lane) * eps + e;
                    if (precision == IDC_BF16) out16[idx] = f32_to_bf16_rne((float)u);
                    else out[idx] = (float)u;
                }
        }
}

// Winograd F(2x2,2x2) image of a ConvTranspose 4x4 s2 p1 layer (fp32, conv_wino_deconv_f32): per output phase (r,s) the 2x2 sub-kernel
// g[a][b] = W[ci][co][KY[r][a]][KY[s][b]], KY = {{3,1},{2,0}} (taps in ascending input offset: SURVEY.md Appendix C), U = G g G^T
// with G = [[1,0],[1,1],[0,1]]; position p = ((r*2+s)*3 + i)*3 + j; same fragment order as pack_wino_weights with 36 positions.
static void pack_wino_deconv_weights(uint8_t* img, int precision, const LayerSpec& s, const LayerBlob& lb, const float* w) {
    memset(img, 0, lb.w3_bytes);
    static const int KY[2][2] = {{3, 1}, {2, 0}};
    const int ncb = cout_pad(s.cout) / 16;
    const int kc = kc_elems(precision), eps = kSlotBytes / elem_bytes(precision);
    float* const out = (float*)img;
    uint16_t* const out16 = (uint16_t*)img;
    for (int ci = 0; ci < s.cin; ++ci)
        for (int co = 0; co < s.cout; ++co) {
            const float* g16 = w + ((size_t)ci * s.cout + co) * 16;             // (Cin, Cout, 4, 4)
            const int c = ci / kc, within = ci % kc, slot = within / eps, e = within % eps, ks = slot / 4, gq = slot % 4;
            const int cbg = co / 16, m = co % 16, lane = gq * 16 + m;
            for (int r = 0; r < 2; ++r)
                for (int sx = 0; sx < 2; ++sx) {
                    double g[2][2];
                    for (int a = 0; a < 2; ++a)
                        for (int b = 0; b < 2; ++b) g[a][b] = g16[KY[r][a] * 4 + KY[sx][b]];
                    const double t[3][2] = {{g[0][0], g[0][1]}, {g[0][0] + g[1][0], g[0][1] + g[1][1]}, {g[1][0], g[1][1]}};   // G g
                    for (int i = 0; i < 3; ++i) {
                        const double u3[3] = {t[i][0], t[i][0] + t[i][1], t[i][1]};                                       // (G g) G^T
                        for (int j = 0; j < 3; ++j) {
                            const int p = ((r * 2 + sx) * 3 + i) * 3 + j;
                            const size_t idx = (((((size_t)c * 36 + p) * ncb + cbg) * 2 + ks) * 64 + lane) * eps + e;
                            if (precision == IDC_BF16) out16[idx] = f32_to_bf16_rne((float)u3[j]);
                            else out[idx] = (float)u3[j];
                        }
                    }
                }
        }
}

static int fail(std::string* err, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    if (err) *err = buf;
    return code;
}

// the queued image packers on up to 16 host threads (each writes its own image: no sharing)
static void run_pack_tasks(std::vector<std::function<void()>>& t) {
    if (t.empty()) return;
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt == 0 ? 4 : (nt > 16 ? 16 : nt);
    if (nt > t.size()) nt = (unsigned)t.size();
    std::atomic<size_t> next{0};
    auto work = [&]() { for (size_t i; (i = next.fetch_add(1)) < t.size();) t[i](); };
    std::vector<std::thread> th;
    for (unsigned k = 1; k < nt; ++k) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
    t.clear();
}

static int pack_weights_impl(int precision, unsigned flags, const idc_tensor_desc* tensors, int n_tensors,
                             void* blob, size_t blob_bytes, std::string* err) {
    if (precision < IDC_FP32 || precision > IDC_FP16) return fail(err, IDC_ERR_INVALID_ARG, "bad precision %d", precision);
    if (!tensors || n_tensors <= 0 || !blob) return fail(err, IDC_ERR_INVALID_ARG, "null tensors/blob");
    const BlobPlan plan = make_blob_plan(precision, flags);
    if (blob_bytes < plan.total_bytes)
        return fail(err, IDC_ERR_INVALID_ARG, "blob too small: %zu < %zu", blob_bytes, plan.total_bytes);
    std::map<std::string, TensorView> sd;
    for (int i = 0; i < n_tensors; ++i) {
        if (!tensors[i].name || !tensors[i].data) return fail(err, IDC_ERR_INVALID_ARG, "tensor %d has null name/data", i);
        TensorView v;
        v.data = tensors[i].data;
        v.ndim = tensors[i].ndim;
        for (int d = 0; d < 4 && d < v.ndim; ++d) v.dims[d] = tensors[i].dims[d];
        sd[tensors[i].name] = v;
    }
    auto need = [&](const std::string& key, const TensorView** out) -> bool {
        auto it = sd.find(key);
        if (it == sd.end()) return false;
        *out = &it->second;
        return true;
    };
    uint8_t* const base = (uint8_t*)blob;
    memset(base, 0, plan.total_bytes);
    const auto& specs = layer_specs();
    std::vector<std::function<void()>> tasks;
    // IDC_FP16X3: per-layer power-of-two weight scale (f16_weight_exponent); a deconv and the shortcut conv it is summed with share ONE (the smaller
    // exponent): conv_ds_fused_ms accumulates both K loops into one accumulator set.  Every other precision: exponent 0.
    std::vector<int> wexp(plan.active.size(), 0);
    if (split_is_f16(precision) && split_parts(precision) > 1) {       // (IDC_FP16, one part: a weight keeps its 11 bits down to 6e-5 unscaled)
        for (size_t li = 0; li < plan.active.size(); ++li) {
            const LayerSpec& s = specs[plan.active[li]];
            if (plan.layers[li].f32) continue;
            const TensorView* w = nullptr;
            if (!need(std::string(s.wkey) + ".weight", &w)) continue;      // (reported by the loop below)
            size_t cnt = 1;
            for (int d = 0; d < w->ndim; ++d) cnt *= (size_t)w->dims[d];
            wexp[li] = f16_weight_exponent(w->data, cnt);
        }
        for (size_t li = 0; li < plan.active.size(); ++li) {
            const LayerSpec& s = specs[plan.active[li]];
            if (!s.resid) continue;
            for (size_t lj = 0; lj < plan.active.size(); ++lj)
                if (strcmp(specs[plan.active[lj]].name, s.resid) == 0 && !plan.layers[lj].f32 && !plan.layers[li].f32)
                    wexp[li] = wexp[lj] = std::min(wexp[li], wexp[lj]);
        }
    }
    for (size_t li = 0; li < plan.active.size(); ++li) {
        const LayerSpec& s = specs[plan.active[li]];
        const LayerBlob& lb = plan.layers[li];
        const TensorView *w = nullptr, *b = nullptr;
        const std::string wk = std::string(s.wkey) + ".weight", bk = std::string(s.wkey) + ".bias";
        if (!need(wk, &w)) return fail(err, IDC_ERR_MISSING_KEY, "missing state_dict key '%s'", wk.c_str());
        if (!need(bk, &b)) return fail(err, IDC_ERR_MISSING_KEY, "missing state_dict key '%s'", bk.c_str());
        bool ok;
        if (s.kind == kDeconv4x4) ok = dims_are(*w, {s.cin, s.cout, 4, 4});
        else if (s.kind == kConv1x1) ok = dims_are(*w, {s.cout, s.cin, 1, 1});
        else ok = dims_are(*w, {s.cout, s.cin, 3, 3});
        if (!ok) return fail(err, IDC_ERR_MISSING_KEY, "key '%s' has the wrong shape", wk.c_str());
        if (!dims_are(*b, {s.cout})) return fail(err, IDC_ERR_MISSING_KEY, "key '%s' has the wrong shape", bk.c_str());
        const int lprec = lb.f32 ? (int)IDC_FP32 : precision;        // (operand-split precisions: model1's fp32 island)
        // the weight images are independent of each other: queued here, packed by the worker threads below (round 6: 1.4 s -> 0.2 s for a bf16 blob)
        const LayerSpec* sp = &s; const LayerBlob* lbp = &lb; const float* wd = w->data;
        const float wmul = ldexpf(1.f, wexp[li]);
        if (lb.wscale_off != (size_t)-1) *(float*)(base + lb.wscale_off) = ldexpf(1.f, -wexp[li]);
        for (int part = 0; part < lb.parts; ++part)
            tasks.push_back([=]() { pack_layer_weights(base + lbp->w_off + (size_t)part * lbp->w_bytes, lprec, 1, *sp, *lbp, wd, part, wmul); });
        if (precision == IDC_FP16 && s.kind == kConvIm2col && lb.w2_off != (size_t)-1) {
            LayerBlob lbc = lb; lbc.nkc = 1; lbc.w_bytes = kWBlockBytes;           // one 64-wide fp16 chunk
            tasks.push_back([=]() { pack_layer_weights(base + lbc.w2_off, (int)IDC_FP16, 1, *sp, lbc, wd, 0, 1.f); });
        } else if (lb.w2_off != (size_t)-1) tasks.push_back([=]() { pack_layer_weights(base + lbp->w2_off, lprec, 2, *sp, *lbp, wd); });
        if (lb.w3_off != (size_t)-1) {
            if (s.kind == kDeconv4x4) tasks.push_back([=]() { pack_wino_deconv_weights(base + lbp->w3_off, lprec, *sp, *lbp, wd); });
            else tasks.push_back([=]() { pack_wino_weights(base + lbp->w3_off, lprec, *sp, *lbp, wd); });
        }
        float* bias = (float*)(base + lb.bias_off);
        for (int c = 0; c < s.cout; ++c) bias[c] = b->data[c];
        if (s.bnkey) {
            const TensorView *g = nullptr, *be = nullptr, *mu = nullptr, *var = nullptr;
            const std::string p = s.bnkey;
            if (!need(p + ".weight", &g) || !need(p + ".bias", &be) || !need(p + ".running_mean", &mu) ||
                !need(p + ".running_var", &var))
                return fail(err, IDC_ERR_MISSING_KEY, "missing BatchNorm keys under '%s'", s.bnkey);
            if (!dims_are(*g, {s.cout}) || !dims_are(*be, {s.cout}) || !dims_are(*mu, {s.cout}) || !dims_are(*var, {s.cout}))
                return fail(err, IDC_ERR_MISSING_KEY, "BatchNorm '%s' has the wrong shape", s.bnkey);
            float* sc = (float*)(base + lb.bn_scale_off);
            float* sh = (float*)(base + lb.bn_shift_off);
            for (int c = 0; c < cout_pad(s.cout); ++c) { sc[c] = 1.f; sh[c] = 0.f; }
            for (int c = 0; c < s.cout; ++c) {      // eval-BN folded in fp64: y = x*s + t  (eps 1e-5)
                const double sd_ = (double)g->data[c] / sqrt((double)var->data[c] + 1e-5);
                sc[c] = (float)sd_;
                sh[c] = (float)((double)be->data[c] - (double)mu->data[c] * sd_);
            }
        }
    }
    run_pack_tasks(tasks);
    // layers that sum a shortcut branch: bias of the fused launch = own bias + the shortcut conv's bias
    for (size_t li = 0; li < plan.active.size(); ++li) {
        const LayerSpec& s = specs[plan.active[li]];
        if (plan.layers[li].fbias_off == (size_t)-1) continue;
        float* fb = (float*)(base + plan.layers[li].fbias_off);
        const float* own = (const float*)(base + plan.layers[li].bias_off);
        for (int ch = 0; ch < cout_pad(s.cout); ++ch) fb[ch] = own[ch];
        for (size_t lj = 0; lj < plan.active.size(); ++lj)
            if (strcmp(specs[plan.active[lj]].name, s.resid) == 0) {
                const float* sb = (const float*)(base + plan.layers[lj].bias_off);
                for (int ch = 0; ch < cout_pad(s.cout); ++ch) fb[ch] += sb[ch];
            }
    }
    {
        const TensorView *w = nullptr, *b = nullptr;
        if (!need("model_out.0.weight", &w) || !dims_are(*w, {2, 128, 1, 1}))
            return fail(err, IDC_ERR_MISSING_KEY, "missing or mis-shaped key 'model_out.0.weight'");
        if (!need("model_out.0.bias", &b) || !dims_are(*b, {2}))
            return fail(err, IDC_ERR_MISSING_KEY, "missing or mis-shaped key 'model_out.0.bias'");
        memcpy(base + plan.head_w_off, w->data, 2 * 128 * 4);
        memcpy(base + plan.head_b_off, b->data, 2 * 4);
    }
    if (plan.pred_ab_off != (size_t)-1) {      // pred_ab: 1x1 conv 313 -> 2 (deploy_nopred.prototxt:842-850; weight = pts_in_hull.T)
        const TensorView *w = nullptr, *b = nullptr;
        if (!need("pred.pred_ab.weight", &w) || !dims_are(*w, {2, 313, 1, 1}))
            return fail(err, IDC_ERR_MISSING_KEY, "missing or mis-shaped key 'pred.pred_ab.weight' (2,313,1,1: the ab bin centres)");
        if (!need("pred.pred_ab.bias", &b) || !dims_are(*b, {2}))
            return fail(err, IDC_ERR_MISSING_KEY, "missing or mis-shaped key 'pred.pred_ab.bias'");
        memcpy(base + plan.pred_ab_off, w->data, 2 * 313 * 4);
        memcpy(base + plan.pred_ab_off + 2 * 313 * 4, b->data, 2 * 4);
    }
    if (plan.glob_off != (size_t)-1) {
        // Global-hints branch (deploy_nodist.prototxt:37-172): stage 1 = glob_conv1 (314 in) + s_conv1 (2 in) summed
        // before the ReLU (Eltwise :66-72), stages 2..4 = glob_conv2..4; each followed by ReLU then BatchNorm.
        // Stored transposed [k][512] + (bias, bn scale, bn shift) per stage, all fp32.
        float* gp = (float*)(base + plan.glob_off);
        auto conv1x1 = [&](const char* key, int cin, const TensorView** w, const TensorView** b) -> bool {
            const std::string wk = std::string(key) + ".weight", bk = std::string(key) + ".bias";
            return need(wk, w) && need(bk, b) && dims_are(**w, {kGlobC, cin, 1, 1}) && dims_are(**b, {kGlobC});
        };
        auto bn_fold = [&](const char* key, float* sc, float* sh) -> bool {
            const TensorView *g = nullptr, *be = nullptr, *mu = nullptr, *var = nullptr;
            const std::string p = key;
            if (!need(p + ".weight", &g) || !need(p + ".bias", &be) || !need(p + ".running_mean", &mu) ||
                !need(p + ".running_var", &var)) return false;
            if (!dims_are(*g, {kGlobC}) || !dims_are(*be, {kGlobC}) || !dims_are(*mu, {kGlobC}) || !dims_are(*var, {kGlobC})) return false;
            for (int c = 0; c < kGlobC; ++c) {
                const double sd_ = (double)g->data[c] / sqrt((double)var->data[c] + 1e-5);
                sc[c] = (float)sd_; sh[c] = (float)((double)be->data[c] - (double)mu->data[c] * sd_);
            }
            return true;
        };
        const TensorView *wg = nullptr, *bg = nullptr, *ws = nullptr, *bs = nullptr;
        if (!conv1x1("glob.glob_conv1", 314, &wg, &bg) || !conv1x1("glob.s_conv1", 2, &ws, &bs))
            return fail(err, IDC_ERR_MISSING_KEY, "missing or mis-shaped global-hints keys 'glob.glob_conv1' / 'glob.s_conv1'");
        for (int k = 0; k < 314; ++k) for (int c = 0; c < kGlobC; ++c) gp[(size_t)k * kGlobC + c] = wg->data[(size_t)c * 314 + k];
        for (int k = 0; k < 2; ++k) for (int c = 0; c < kGlobC; ++c) gp[(size_t)(314 + k) * kGlobC + c] = ws->data[(size_t)c * 2 + k];
        float* q = gp + (size_t)kGlobIn * kGlobC;
        for (int c = 0; c < kGlobC; ++c) q[c] = bg->data[c] + bs->data[c];
        if (!bn_fold("glob.bn1", q + kGlobC, q + 2 * kGlobC))
            return fail(err, IDC_ERR_MISSING_KEY, "missing or mis-shaped BatchNorm keys under 'glob.bn1'");
        q += 3 * kGlobC;
        for (int st = 2; st <= 4; ++st) {
            char ck[32], bk[32];
            snprintf(ck, sizeof(ck), "glob.glob_conv%d", st); snprintf(bk, sizeof(bk), "glob.bn%d", st);
            const TensorView *w = nullptr, *b = nullptr;
            if (!conv1x1(ck, kGlobC, &w, &b)) return fail(err, IDC_ERR_MISSING_KEY, "missing or mis-shaped key '%s'", ck);
            for (int k = 0; k < kGlobC; ++k) for (int c = 0; c < kGlobC; ++c) q[(size_t)k * kGlobC + c] = w->data[(size_t)c * kGlobC + k];
            float* r = q + (size_t)kGlobC * kGlobC;
            for (int c = 0; c < kGlobC; ++c) r[c] = b->data[c];
            if (!bn_fold(bk, r + kGlobC, r + 2 * kGlobC)) return fail(err, IDC_ERR_MISSING_KEY, "missing or mis-shaped BatchNorm keys under '%s'", bk);
            q = r + 3 * kGlobC;
        }
    }
    BlobHeader h;
    memset(&h, 0, sizeof(h));
    h.magic = kBlobMagic; h.version = IDC_VERSION; h.precision = (uint32_t)precision; h.flags = plan.flags;
    h.total_bytes = plan.total_bytes;
    h.checksum = fnv1a(base + sizeof(BlobHeader), plan.total_bytes - sizeof(BlobHeader));
    memcpy(base, &h, sizeof(h));
    return IDC_OK;
}

// ================================================================================================
struct Tensor {
    std::string name;
    void* ptr = nullptr;
    int C = 0, Cpad = 0, H = 0, W = 0;
    int is_f32 = 0;                      // fp32 storage (else the context's element type)
    int parts = 1;                       // operand-split precisions: bf16 planes per pixel ([part][Cpad]); 1 otherwise
    size_t bytes = 0;
};

struct Layer {
    const LayerSpec* spec = nullptr;
    LayerBlob blob;
    int src = -1, dst = -1, resid = -1;
    int halo = 0;
    ConvConfig cfg{2, 2};
    bool v2 = false;                     // bf16 large-tile kernel (layout-2 weights)
    bool m16 = false;                    // ... its 16x16x32-MFMA build (conv_igemm_v2m, layout-1 weights): set per launch in run_graph
    bool v2p = false;                    // ... conv_igemm_v2p (padded halo rows, unrolled taps): set per launch in run_graph
    bool f16fast = false;                // IDC_FP16: this launch runs the bf16 throughput kernel's fp16 twin (conv_igemm_v2ph / conv_ds_fused_mh): set per launch in run_graph
    bool click = false;                  // batch-1 click-path kernel (conv_click: whole K slice by LDS-DMA)
    bool wino = false;                   // fp32 Winograd F(2x2,3x3) kernel (conv_wino_f32, idc_wino.hip)
    bool kw = false;                     // bf16 click path: conv_kwave_bf16 (idc_kw.hip: K split over the waves of a workgroup, layout-1 weights)
    int chain_len = 0;                   // > 0: this layer and the chain_len - 1 after it ran as ONE conv_kwave_chain_bf16 launch (last forward)
    int chained_into = -1;               // >= 0: ran inside the chain launch headed by that layer (last forward)
    bool fused_head = false;             // conv10_2 only: model_out + tanh run in this layer's epilogue
    int fused_short = -1;                // deconv layers: index of the shortcut conv layer riding in this launch's K loop
    bool skip = false;                   // layer fused into another launch: not launched itself
    int fused_next = -1;                 // conv1_1 only: index of conv1_2 when model1 runs as one launch (conv1_block_fused)
    int lprec = 0;                       // the precision this layer's kernels run in (the handle's; IDC_FP32 on the fp32 island of a split handle)
    bool split = false;                  // operand-split launch (conv_igemm_v2s / conv_igemm_v2ps)
    ConvArgs args{};                     // zero-initialised; pointers patched per forward where they depend on weights
    double flops = 0, min_bytes = 0;
};

}  // namespace idc

using namespace idc;

struct idc_context {
    int device = 0, H = 0, W = 0, max_batch = 0, precision = 0;
    unsigned flags = 0;
    hipStream_t stream = nullptr;
    std::string err;
    float l_div = 100.f, ab_div = 110.f, mask_mul = 1.f, out_mul = 110.f;
    BlobPlan plan;
    uint8_t* d_blob = nullptr;
    bool own_blob = false, weights_set = false;
    std::vector<Tensor> tensors;
    std::vector<Layer> layers;
    int t_input = -1, t_conv10_2 = -1, t_logits = -1;
    // staging for the host-pointer forward
    float *h_in = nullptr, *h_out = nullptr, *h_dist = nullptr;
    float *d_L = nullptr, *d_ab = nullptr, *d_mask = nullptr, *d_out = nullptr, *d_dist = nullptr;
    float* d_scratch = nullptr; size_t scratch_bytes = 0;
    float* d_partial = nullptr; size_t partial_bytes = 0;    // split-K slice sums (grown on demand)
    void* d_zeros = nullptr;             // 256 zero bytes: LDS-DMA source of out-of-image halo rows (conv_click)
    unsigned long long* d_kw_bar = nullptr;   // conv_kwave_chain_bf16's grid-barrier counter (monotone) ...
    unsigned long long kw_bar_count = 0;      // ... grid barriers done by every launch so far (each adds its arrivals to its counter)
    int kw_bar_blocks = 0;                    // ... workgroups per launch those counts are for (a different grid resets the counters)
    long long* d_kw_stamps = nullptr;         // IDC_KW_STAMPS=1: per-phase cycle stamps of the last chain launch, printed when the handle is destroyed
    int kw_stamp_layers = 0, kw_stamp_blocks = 0;
    int* h_kw_abort = nullptr;                // pinned, device-visible: a chain workgroup that gave up waiting sets it
    bool kw_chain_off = false;                // set after a refused / aborted chain launch: the handle falls back to one launch per layer
    int kw_chain_fits = -1;                   // workgroups of the chain kernel this device holds at once (-1: not asked yet)
    float *d_glob_in = nullptr, *d_glob_vec = nullptr;   // global hints: [max_batch][316] inputs, [max_batch][512] branch output
    int t_conv4_3 = -1, t_pred313 = -1;
    float *d_pred_ab = nullptr, *d_dist313 = nullptr, *h_pred_ab = nullptr, *h_dist313 = nullptr;   // 313 head outputs
    float dist_S = 0.2f;
    unsigned char *d_rgb = nullptr, *h_rgb = nullptr;   // colour post-processing (allocated on first use)
    double *d_labq = nullptr, *h_labq = nullptr;
    float* d_post_in = nullptr;          // idc_lab2rgb staging (L + ab planes): the resident L / hint planes are left alone
    bool want_dist313 = false;           // the next forward also writes the full-resolution dist_S
    bool keep_dist313 = false;           // idc_keep_dist: every forward leaves dist_S resident (colour suggestions)
    int dist_n = 0;                      // images whose distribution is resident from the last forward (0 = none)
    HintRect *d_hints = nullptr, *h_hints = nullptr; int hints_cap = 0;   // click session: hint list staging
    float* d_centres = nullptr; double* d_sugg = nullptr; unsigned* d_sugg_counts = nullptr;   // colour suggestions
    std::vector<char> l_set;             // per image slot: d_L holds an uploaded L plane (idc_forward_resident refuses otherwise)
    hipEvent_t ev_sync = nullptr;        // idc_stream_wait / idc_stream_signal
    // two-slot transfer pipeline (idc_forward_async / idc_wait): each slot owns its device planes
    struct PipeSlot {
        float *d_L = nullptr, *d_ab = nullptr, *d_mask = nullptr, *d_out = nullptr;   // device I/O planes
        float *h_in = nullptr, *h_out = nullptr;                                       // pinned staging (pageable callers)
        hipEvent_t ev_in = nullptr, ev_comp = nullptr, ev_out = nullptr;
        hipEvent_t ev_in0 = nullptr, ev_comp0 = nullptr, ev_out0 = nullptr;             // stage starts (idc_pipeline_times)
        bool pending = false, staged_out = false, timed = false;
        float* user_out = nullptr; int n = 0;
    } pipe[2];
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_pipe_base = nullptr;
    bool pipe_ready = false;
    unsigned char* d_up_rgb = nullptr; double* d_up_L = nullptr; size_t up_cap = 0;    // idc_upsample_lab2rgb staging
    unsigned char* h_up_rgb = nullptr; double* h_up_L = nullptr;
    bool out_copy_pending = false;       // forward_host(finish = false): the ab map still has to be copied from h_out to the caller
    bool out_resident = false;           // d_out / d_labq hold the last forward's ab map / refreshed Lab
    bool labq_resident = false;
    int profiling = 0;                   // 0 off, 1 = an event pair around every launch, 2 = one pair around the whole forward
    void* d_arena = nullptr;             // all activation tensors (alloc_graph), or nullptr with IDC_ARENA=0
    std::vector<hipEvent_t> ev;          // kProfRing slots x 2 per timed step: [pack, layers..., head, softmax]
    int n_timed = 0;
    long long prof_count = 0;            // forwards recorded since profiling was switched on
    int last_n = 0;
};

#define HIPCHK(ctx, expr)                                                                                  \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) (void)hipGetLastError();   /* reported through our own status: do not leave it sticky for the caller's runtime */ \
        if (e_ != hipSuccess)                                                                              \
            return fail((ctx) ? &(ctx)->err : nullptr, IDC_ERR_HIP, "%s failed: %s (%s:%d)", #expr,        \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                        \
    } while (0)

static constexpr int kProfRing = 32;    // per-layer event pairs are kept for the last 32 forwards

static int find_tensor(idc_context* c, const char* name) {
    for (size_t i = 0; i < c->tensors.size(); ++i)
        if (c->tensors[i].name == name) return (int)i;
    return -1;
}

// Tile policy (speed only; every choice computes the same result): 0 = automatic, 1 = always the
// small-tile kernels (conv_igemm), 2 = the large-tile bf16 kernel (conv_igemm_v2) wherever it applies.
static int g_tile_policy = 0;
static int g_fuse_conv1 = idc_env_int("IDC_FUSE_CONV1", 1) != 0;   // model1 (conv1_1 + conv1_2) as one launch on the bf16 throughput path (idc_set_option / env IDC_FUSE_CONV1=0 for A/B)
// Split-K policy of the small-tile kernels (speed only): 0 = automatic (launches that would leave most CUs idle,
// i.e. the batch-1 click path), 1 = never, 2 = always split as far as the cin chunks allow (tests).
static int g_splitk_policy = 0;
// fp32 path: 3x3 stride-1 layers as Winograd F(2x2,3x3), small deconv launches as F(2x2,2x2) (idc_set_option "winograd"): 0 = off (direct kernels),
// 1 = automatic (default), 2 = wherever the forms are implemented (deconvs at every size: tests), 12 / 21 / 22 = automatic with the 3x3 form
// <TB,CB> forced (tests, tuning).  The bf16 twins of round 3 were retired in round 5 (docs/experiments/conv_wino_bf16_round3.hip.txt).
static int g_wino = 1;
static int g_wino_deconv = 1;            // 1 = small launches with Cin >= 256 only; 2 = every deconv ("winograd" = 2)
// conv_igemm_v2 launches that qualify run as conv_igemm_v2m (16x16x32 MFMA: fewer joules per FLOP at the power cap; idc_set_option "mfma16")
static int g_mfma16 = idc_env_int("IDC_MFMA16", 1);
// ... and so do the three deconv + shortcut launches (conv_ds_fused_m, idc_dsm.hip; idc_set_option "ds_mfma16" / env IDC_DS_M16=0 for A/B)
static int g_ds_m16 = idc_env_int("IDC_DS_M16", 1);
// operand-split precisions: the deconv + shortcut pairs as ONE launch (conv_ds_fused_ms / _msh) instead of shortcut conv (fp32 sums to HBM) + deconv
// (idc_set_option "split_ds_fuse", 0 for A/B)
static int g_split_ds_fuse = 1;
// ... and conv1_1, their exact-fp32 island, on conv1_1_split_kernel where the grid is throughput-sized (>= 128 tiles of 32 x 16; "conv1_1_split", 0 = conv_igemm<float>)
static int g_conv1_1_split = 1;
// ... and conv1_2 (64 -> 64 at full resolution) on conv1_2_split_kernel instead of the generic 64-cout tile ("conv1_2_split", 0 = conv_igemm_v2ps<1,4,1>)
static int g_conv1_2_split = 1;
// IDC_FP16 (one fp16 part, one segment): launches the bf16 throughput kernels cover run their fp16 twins (conv_igemm_v2ph, conv_ds_fused_mh: bias in the
// accumulators, packed-pair epilogue) instead of the one-segment split kernels ("fp16_fast", 0 = the split kernels everywhere)
static int g_fp16_fast = 1;
static bool conv1_2_split_layer(const LayerSpec& s) {
    return s.kind == kConv3x3 && s.cin == 64 && s.cout == 64 && s.dilation == 1 && s.in_stride == 1 && s.act == 1 && !s.resid;
}
// ... and the 3x3 convs among them as conv_igemm_v2p (no address arithmetic in the K loop; idc_set_option "v2p" / env IDC_V2P=0 for A/B)
static int g_v2p = idc_env_int("IDC_V2P", 1);
// throughput kernels touch their own code at entry (idc_warm_own_code, idc_kernels.h; env IDC_CODE_WARM=0 for the A/B of profiles/r04_firstuse.txt)
static int g_code_warm = idc_env_int("IDC_CODE_WARM", 1);
// bf16 click path: the 3x3 stride-1 layers and the deconvs as conv_kwave_bf16 / conv_kwave_deconv_bf16 ("kwave"; 0 = conv_click + split-K, round 2's kernels)
static int g_kwave = 1;
// ... and runs of consecutive same-shape 8-chunk conv_kwave_bf16 layers (the 512 -> 512 trunk at batch 1) as ONE persistent launch with a
// grid barrier between layers ("kwave_chain" / IDC_KWAVE_CHAIN): 0 = off, 1 = hipLaunchCooperativeKernel (+24 us per launch on this runtime),
// 2 = plain launch after an occupancy check (default; a workgroup that never sees the others gives up after ~0.3 s and the handle falls back)
static int g_kwave_chain = idc_env_int("IDC_KWAVE_CHAIN", 2);
static int g_spin_sync = 1;              // one- and two-image calls poll the stream instead of parking on an interrupt ("spin_sync"; 0 = the blocking wait of rounds 1-4)
static int g_pcie_kernel = 1;            // their host <-> device transfers as copy kernels on the forward's stream ("pcie_kernel"; 0 = hipMemcpyAsync / the copy engines)
static int g_kw_force_abort = 0;         // test hook ("kw_force_abort"): the persistent trunk launch's first grid barrier is unreachable and its give-up counter tiny
static int g_click = -1;                 // conv_click for small launches: -1 = environment default (on), 0 off, 1 on (idc_set_option "click")
// Shortcut fusion (conv_igemm_v2<.,.,1,true>) is correct (parity-tested under tile policy 2) but measured slower
// than two launches on MI355X (4x re-reads of the skip tensor by the four phase workgroups, VGPR spills around
// the staged K loop): 0.87 ms -> 1.25 ms at level 1.  Off unless the tile policy forces every variant on.
// large-tile deconv launches also run the shortcut conv they are summed with (conv_ds_fused); IDC_FUSE_SHORTCUT=0 keeps
// the two launches apart (A/B)
static bool fuse_shortcut_enabled() {
    static const bool off = idc_env_int("IDC_FUSE_SHORTCUT", 1) == 0;
    return !off;
}

// Small-tile kernel shape.  cout<=64 layers can only use one 64-wide cout group per wave column;
// otherwise prefer 128 couts x 128 pixels and fall back to smaller pixel tiles when the launch would
// not fill the 256 CUs.
// Tuning knobs of the small-tile path (speed only), read once from the environment: tools/click_sweep.py walks them on
// the GPU box and the defaults below are what it found best (profiles/r02_click_sweep.txt).
static int env_int(const char* name, int dflt) { return idc_env_int(name, dflt); }      // (defaults only, unless built with -DIDC_AB_PARTNERS)
struct SmallTileTuning {
    int tiles_goal = env_int("IDC_ST_TILES_GOAL", 512);         // shrink the pixel tile until this many workgroups exist
    int force_wp = env_int("IDC_ST_FORCE_WP", 0);               // 1/2/4: fixed rows-of-4 per workgroup (0 = automatic)
    int force_wm = env_int("IDC_ST_FORCE_WM", 0);               // 1/2: cout groups per workgroup (0 = automatic)
    int sk_below[2] = {env_int("IDC_SK_BELOW_FP32", 256), env_int("IDC_SK_BELOW_BF16", 128)};   // split K when fewer tiles than this
    int sk_goal[2] = {env_int("IDC_SK_GOAL_FP32", 512), env_int("IDC_SK_GOAL_BF16", 256)};      // ... until about this many workgroups
    int v2_min_blocks = env_int("IDC_V2_MIN_BLOCKS", 128);      // large-tile bf16 kernel from this many workgroups on
    int v2_half_tiles = env_int("IDC_V2_HALF_TILES", 1);        // 4-wave large tiles for grids of 128..255 workgroups
    int v2_force22 = env_int("IDC_V2_FORCE22", 0);
    int v2_22_nkc = env_int("IDC_V2_22_NKC", 2);
    int click = env_int("IDC_CLICK", 1);                        // conv_click for small launches (the batch-1 click path)
    int click_max_wgs = env_int("IDC_CLICK_MAX_WGS", 1024);     // ... when its grid has at most this many workgroups
    int click_goal = env_int("IDC_CLICK_GOAL", 512);            // ... split K over the cin chunks until about this many exist (same-box A/B: 512 beats 256 by 1.5 % bf16 / 3 % fp32)
    int click_wp = env_int("IDC_CLICK_WP", 4);                  // rows-of-4 per workgroup (4 = 16x16 sites, 256 threads)
};
static const SmallTileTuning& tuning() { static const SmallTileTuning t; return t; }

static ConvConfig choose_config(int n, int Hs, int Ws, int coutpad, int nphase) {
    const SmallTileTuning& tn = tuning();
    int wm = coutpad >= 128 ? 2 : 1;
    if (tn.force_wm && coutpad >= 64 * tn.force_wm && (coutpad / 64) % tn.force_wm == 0) wm = tn.force_wm;
    if (tn.force_wp == 1 || tn.force_wp == 2 || tn.force_wp == 4) return ConvConfig{wm, tn.force_wp};
    const int cand_wp[3] = {wm == 1 ? 4 : 2, 2, 1};
    ConvConfig best{wm, cand_wp[0]};
    for (int i = 0; i < 3; ++i) {
        const int wp = cand_wp[i];
        const long long tiles = (long long)((Ws + 15) / 16) * ((Hs + 4 * wp - 1) / (4 * wp)) * n * (coutpad / (64 * wm)) * nphase;
        best = ConvConfig{wm, wp};
        if (tiles >= tn.tiles_goal) break;
    }
    return best;
}

static void fill_taps(Layer& L) {
    ConvArgs& a = L.args;
    const LayerSpec& s = *L.spec;
    memset(a.dy, 0, sizeof(a.dy)); memset(a.dx, 0, sizeof(a.dx)); memset(a.tw, 0, sizeof(a.tw));
    memset(a.ro, 0, sizeof(a.ro)); memset(a.co, 0, sizeof(a.co));
    if (s.kind == kConv3x3) {
        a.nphase = 1; a.ntaps = 9; a.so = 1; a.si = s.in_stride;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                const int t = ky * 3 + kx;
                a.dy[t] = (ky - 1) * s.dilation; a.dx[t] = (kx - 1) * s.dilation; a.tw[t] = t;
            }
        L.halo = s.dilation;
    } else if (s.kind == kDeconv4x4) {
        // out[co,2m+r,2n+s] = b + sum over (ky,dy) in T(r), (kx,dx) in T(s) of in[m+dy,n+dx]*W[ci,co,ky,kx]
        // T(0) = {(1,0),(3,-1)}, T(1) = {(0,+1),(2,0)}     
        static const int T_k[2][2] = {{1, 3}, {0, 2}};
        static const int T_d[2][2] = {{0, -1}, {1, 0}};
        a.nphase = 4; a.ntaps = 4; a.so = 2; a.si = 1;
        for (int r = 0; r < 2; ++r)
            for (int c = 0; c < 2; ++c) {
                const int ph = r * 2 + c;
                a.ro[ph] = r; a.co[ph] = c;
                for (int i = 0; i < 2; ++i)
                    for (int j = 0; j < 2; ++j) {
                        const int t = ph * 9 + i * 2 + j;
                        a.dy[t] = T_d[r][i]; a.dx[t] = T_d[c][j];
                        a.tw[t] = T_k[r][i] * 4 + T_k[c][j];
                    }
            }
        L.halo = 1;
    } else {
        a.nphase = 1; a.ntaps = 1; a.so = 1; a.si = 1;
        L.halo = 0;
    }
}

// n_policy: the batch the kernel variant is chosen for (the handle's max_batch, so that a handle's
// numerics do not depend on how many images a call carries); n: the batch actually launched.
static void set_geometry(Layer& L, int precision, int n, int n_policy, int Hs, int Ws, bool allow_v2 = true) {
    ConvArgs& a = L.args;
    a.N = n; a.Hs = Hs; a.Ws = Ws;
    a.nkc = L.blob.nkc; a.ncg = L.blob.ncg;
    a.act = L.spec->act;
    a.out_f32 = L.spec->out_f32;
    // fp32 path: every 3x3 stride-1 layer whose U image is in the blob runs as Winograd F(2x2,3x3): 2.25x fewer multiplies
    // on the exact-fp32 matrix pipe, no split-K at batch 1 (one workgroup per 16 tiles x 32 couts)
    L.wino = precision == IDC_FP32 && g_wino && g_tile_policy != 1 && L.blob.w3_off != (size_t)-1 &&
             (L.spec->kind == kDeconv4x4
                  ? (g_wino_deconv == 2 || (g_wino_deconv == 1 && a.nkc >= 8 &&
                                            (long long)((Ws + 7) / 8) * ((Hs + 7) / 8) * n_policy * (a.ncg * 4) <= 1024))
                  : L.spec->resid == nullptr);
    // (deconvs: only the small launches of the click path with Cin >= 256 -- the form is transform-bound (one 16-cout block per
    //  workgroup at the 168-register budget of its 12 waves): model10up (4 chunks) 123 us vs 96 us direct at batch 1, and at N = 32
    //  model8up / model9up 1.30 / 1.54 ms vs 1.23 / 1.31 ms direct; "winograd_deconv" = 2 forces it everywhere for the tests)
    const bool wino_fits = wino_offsets_fit(Hs, Ws, L.spec->kind == kDeconv4x4 ? 1 : L.spec->in_stride, a.nkc);   // 32-bit patch offsets
    L.wino = L.wino && wino_fits;
    L.kw = false;
    if (L.wino) { L.v2 = false; L.click = false; a.ksplit = 1; a.kc_per = a.nkc; a.tiles_x = a.tiles_y = 0; return; }
    // large-tile bf16 kernel: 256 couts x (32x8 sites) when the cout groups divide by 4, else
    // 128 couts x (32x16 sites); used when its grid covers at least half of the 256 CUs
    L.v2 = false;
    const bool split = is_split(precision);      // operand-split precisions: a throughput path, the large tile on every layer whatever the grid
    // (default library: fp32-output launches -- class_logits -- stay on the small tile: conv_igemm_v2m / v2p write bf16 only)
    const bool v2_covers = allow_v2 && (kAbPartners || !L.spec->out_f32);
    if (split ? !split_island(*L.spec) : (precision == IDC_BF16 && v2_eligible(*L.spec) && g_tile_policy != 1 && v2_covers)) {
        ConvConfig c2 = (a.ncg % 4 == 0) ? ConvConfig{4, 2} : (a.ncg % 2 == 0) ? ConvConfig{2, 4} : ConvConfig{1, 4};   // ({1,*}: split precisions only, conv1_2)
        int tx = (Ws + 31) / 32, ty = (Hs + 4 * c2.wp - 1) / (4 * c2.wp);
        long long blocks = (long long)tx * ty * n_policy * (a.ncg / c2.wm) * a.nphase;
        // between one half and one full wave of workgroups (batch-1 conv10_2: 128 tiles on 256 CUs): the 4-wave tile
        // 128 couts x (32 x 8 sites) doubles the grid
        // ... and the same 4-wave tile wherever a CU walks several tiles with a short K loop each (128-cout layers, and
        // layers with <= v2_22_nkc cin chunks): two 4-wave workgroups share a CU (75.5 KiB of LDS each) and run out of
        // step, so one's prologue / epilogue hides under the other's MFMAs.  Same-box A/B at N = 32 (profiles/
        // r02_tile22_ab.txt): conv10_2 -9.5 %, conv2_1 -8..14 %, conv2_2 -6 %, conv9_2 -6..12 %, conv3_1 -22 %; the
        // 512->512 trunk (one tile per CU, long K) is 1-2 % (dilated: 15 %) slower with it and keeps the 8-wave tile.
        const int force22 = tuning().v2_force22;     // experiment switch: 0 = rule above, 1 = never, 2 = everywhere
        // (the 64-cout tile of the split precisions keeps 32 x 16 sites: conv1_2 0.84 ms as {1,4}, 1.01 ms as {1,2} at N = 32, bf16x3)
        const bool rule22 = force22 == 2 || (force22 == 0 && (c2.wm == 2 || (c2.wm == 4 && a.nkc <= tuning().v2_22_nkc)) && blocks >= 256);
        // operand-split precisions have no batch-1 kernels: a grid that leaves most of the chip idle (fp16x3 at 1..8 images: the 512 -> 512 trunk is 8 x N
        // workgroups of the 8-wave tile) at least takes the 4-wave 128-cout tile -- twice the workgroups, half the K-loop time each (round 6, tools/batch_sweep.py)
        const bool split_small = split && blocks < 128;         // (at 128 workgroups -- the trunk at N = 16 -- the 8-wave tile is still the faster one: 2546 against 2475 img/s)
        if (a.nphase == 1 && c2.wm != 1 && ((blocks >= tuning().v2_min_blocks && blocks < 256 && tuning().v2_half_tiles) || rule22 || split_small)) {
            c2 = ConvConfig{2, 2};
            ty = (Hs + 4 * c2.wp - 1) / (4 * c2.wp);
            blocks = (long long)tx * ty * n_policy * (a.ncg / c2.wm) * a.nphase;
        }
        if (split || g_tile_policy == 2 || blocks >= tuning().v2_min_blocks) {
            L.v2 = true; L.cfg = c2; a.tiles_x = tx; a.tiles_y = ty;
            a.ksplit = 1; a.kc_per = a.nkc;
            return;
        }
    }
    // batch-1 click path: small launches are chains of exposed memory round trips in conv_igemm's K loop; conv_click
    // requests a workgroup's whole K slice at entry (as many cin chunks as fit in LDS next to their halo tiles)
    L.click = false;
    // ... "small" = fewer than v2_min_blocks big tiles (the criterion that hands a layer to the throughput kernels, applied to
    // every precision and cout width): conv_igemm's ring loop is the better kernel once the K loop is long and the chip full
    const int wm_big = a.ncg % 4 == 0 ? 4 : (a.ncg % 2 == 0 ? 2 : 1), rows_big = wm_big == 4 ? 8 : 16;
    const long long big_tiles = (long long)((Ws + 31) / 32) * ((Hs + rows_big - 1) / rows_big) * n_policy * (a.ncg / wm_big) * a.nphase;
    // bf16 click path, deconvs: the direct form with K split over the waves of a workgroup (conv_kwave_deconv_bf16, idc_kw.hip)
    if (precision == IDC_BF16 && g_kwave && g_tile_policy != 1 && big_tiles < tuning().v2_min_blocks && wino_fits &&
        L.spec->kind == kDeconv4x4 && (a.nkc == 2 || a.nkc == 4 || a.nkc == 8)) {
        L.kw = true; a.ksplit = 1; a.kc_per = a.nkc; a.tiles_x = a.tiles_y = 0;
        return;
    }
    // bf16 click path, 3x3 stride-1 layers: the direct form with K split over the waves of a workgroup (idc_kw.hip) -- 9/16 of the Winograd
    // form's weight stream, no reduction launch either
    L.kw = false;
    if (precision == IDC_BF16 && g_kwave && g_tile_policy != 1 && big_tiles < tuning().v2_min_blocks && wino_fits && L.spec->resid == nullptr &&
        L.spec->kind == kConv3x3 && (a.nkc == 1 || a.nkc == 2 || a.nkc == 4 || a.nkc == 8)) {
        L.kw = true; a.ksplit = 1; a.kc_per = a.nkc; a.tiles_x = a.tiles_y = 0;
        return;
    }
    if ((g_click < 0 ? tuning().click : g_click) && g_tile_policy != 1 && big_tiles < tuning().v2_min_blocks &&
        (L.spec->kind == kConv3x3 || L.spec->kind == kDeconv4x4)) {
        int wp = tuning().click_wp;
        while (wp > 1 && 4 * wp > Hs * 2) wp >>= 1;               // tiny images: do not launch mostly-empty tiles
        const int maxc = conv_click_max_chunks(wp, L.halo, a.ntaps);
        const int tx = (Ws + 15) / 16, ty = (Hs + 4 * wp - 1) / (4 * wp);
        const long long tiles = (long long)tx * ty * n_policy * a.ncg * a.nphase;
        if (maxc >= 1) {
            int kc_per = maxc < a.nkc ? maxc : a.nkc;                         // split-K policy "never": one workgroup walks all chunks
            if (g_splitk_policy == 0)
                while (kc_per > 1 && tiles * ((a.nkc + kc_per - 1) / kc_per) < tuning().click_goal) --kc_per;
            if (g_splitk_policy == 2) kc_per = 1;
            if (kc_per >= 1) {
                const int ks = (a.nkc + kc_per - 1) / kc_per;
                if (tiles * ks <= tuning().click_max_wgs) {
                    L.click = true; L.cfg = ConvConfig{1, wp};
                    a.tiles_x = tx; a.tiles_y = ty; a.kc_per = kc_per; a.ksplit = ks;
                    return;
                }
            }
        }
    }
    L.cfg = choose_config(n_policy, Hs, Ws, L.blob.ncg * kCoutGroup, a.nphase);
    a.tiles_x = (Ws + 15) / 16;
    a.tiles_y = (Hs + 4 * L.cfg.wp - 1) / (4 * L.cfg.wp);
    // split-K: a launch of < 256 workgroups (batch-1: 32..128) cannot fill 256 CUs and each workgroup walks all
    // 9*Cin/64 tap-steps alone; cut the cin chunks into slices until ~512 workgroups exist
    a.ksplit = 1; a.kc_per = a.nkc;
    const long long tiles = (long long)a.tiles_x * a.tiles_y * n_policy * (a.ncg / L.cfg.wm) * a.nphase;
    const int pi = precision == IDC_FP32 ? 0 : 1;
    if (g_splitk_policy != 1 && a.nkc >= 2 && (g_splitk_policy == 2 || tiles < tuning().sk_below[pi])) {
        // fp32 tap-steps are 16x longer than bf16 ones: worth twice as many slices (measured: 3.0 -> 2.0 ms at N=1)
        long long want = g_splitk_policy == 2 ? a.nkc : (tuning().sk_goal[pi] + tiles - 1) / tiles;
        if (want > a.nkc) want = a.nkc;
        if (want >= 2) {
            a.kc_per = (int)((a.nkc + want - 1) / want);
            a.ksplit = (a.nkc + a.kc_per - 1) / a.kc_per;          // every slice is non-empty
            if (a.ksplit < 2) { a.ksplit = 1; a.kc_per = a.nkc; }
        }
    }
}

static double layer_flops(const LayerSpec& s, int H, int W) {       // per image, SURVEY.md Appendix A
    const double ho = (double)(H / s.level), wo = (double)(W / s.level);
    switch (s.kind) {
        case kConv3x3: case kConvIm2col: return 2.0 * s.cin * s.cout * 9.0 * ho * wo;
        case kConv1x1: return 2.0 * s.cin * s.cout * ho * wo;
        case kDeconv4x4: return 2.0 * s.cin * s.cout * 4.0 * ho * wo;   // 4 taps per OUTPUT pixel
    }
    return 0;
}

static int build_graph(idc_context* c) {
    const auto& specs = layer_specs();
    const int eb = elem_bytes(c->precision);
    const bool split = is_split(c->precision);
    auto add_tensor = [&](const char* name, int C, int Cpad, int level, int f32) -> int {
        Tensor t;
        t.name = name; t.C = C; t.Cpad = Cpad; t.H = c->H / level; t.W = c->W / level; t.is_f32 = f32;
        t.parts = (split && !f32) ? split_parts(c->precision) : 1;
        t.bytes = (size_t)c->max_batch * t.H * t.W * Cpad * (f32 ? 4 : eb * t.parts);
        c->tensors.push_back(t);
        return (int)c->tensors.size() - 1;
    };
    // operand-split precisions: is tensor `name` read by a layer outside the fp32 island / summed into another layer as a shortcut?
    auto read_by_split_layer = [&](const char* name) {
        for (int ai : c->plan.active) { const LayerSpec& q = specs[ai]; if (!split_island(q) && strcmp(q.src, name) == 0) return true; }
        return false;
    };
    auto used_as_shortcut = [&](const char* name) {
        for (int ai : c->plan.active) { const LayerSpec& q = specs[ai]; if (q.resid && strcmp(q.resid, name) == 0) return true; }
        return false;
    };
    c->t_input = add_tensor("data_l_ab_mask", 36, 64, 1, 0);      // never materialised: built inside conv1_1's operand staging
    c->tensors[c->t_input].bytes = 256;
    for (size_t li = 0; li < c->plan.active.size(); ++li) {
        const LayerSpec& s = specs[c->plan.active[li]];
        Layer L;
        L.spec = &s; L.blob = c->plan.layers[li];
        L.src = find_tensor(c, s.src);
        if (L.src < 0) return fail(&c->err, IDC_ERR_INVALID_ARG, "graph: unknown source '%s'", s.src);
        if (s.resid) {
            L.resid = find_tensor(c, s.resid);
            if (L.resid < 0) return fail(&c->err, IDC_ERR_INVALID_ARG, "graph: unknown residual '%s'", s.resid);
        }
        // fp32 storage: everything on the fp32 path; on the bf16 path the class / 313 logits and the hyper-column
        // partial sums of the 313 head (LayerSpec.out_f32)
        L.lprec = c->precision;
        if (split && split_island(s)) {
            // fp32 island (conv1_1): exact-fp32 kernel; its epilogue writes the result as split planes where the split stack reads it
            L.lprec = IDC_FP32;
            L.dst = add_tensor(s.name, s.cout, cout_pad(s.cout), s.level, read_by_split_layer(s.name) ? 0 : 1);
        } else {
            // fp32 storage: everything on the fp32 path; class / 313 logits and the hyper-column partial sums (LayerSpec.out_f32); on the
            // operand-split path also every shortcut branch (summed in fp32 in its consumer's epilogue)
            L.split = split;
            L.dst = add_tensor(s.name, s.cout, cout_pad(s.cout), s.level, c->precision == IDC_FP32 || s.out_f32 || (split && used_as_shortcut(s.name)));
        }
        fill_taps(L);
        L.flops = layer_flops(s, c->H, c->W);
        const Tensor& ti = c->tensors[L.src];
        const Tensor& to = c->tensors[L.dst];
        const double in_px = (double)(ti.H / s.in_stride) * (ti.W / s.in_stride);
        L.min_bytes = in_px * ti.Cpad * (ti.is_f32 ? 4 : eb * ti.parts) + (double)to.H * to.W * to.Cpad * (to.is_f32 ? 4 : eb * to.parts) +
                      (L.resid >= 0 ? (double)to.H * to.W * to.Cpad * (c->tensors[L.resid].is_f32 ? 4 : eb) : 0.0);
        c->layers.push_back(L);
    }
    c->t_conv10_2 = find_tensor(c, "conv10_2");
    c->t_logits = find_tensor(c, "class_logits");
    c->t_conv4_3 = find_tensor(c, "conv4_3");
    c->t_pred313 = find_tensor(c, "pred_313");
    return IDC_OK;
}

static int alloc_graph(idc_context* c) {
    // ONE arena for all activation tensors (3.7 GB at N = 32): a single large allocation is 2 MiB-aligned and physically as
    // contiguous as the driver can make it, whatever the process allocated before -- with one hipMalloc per tensor, a process whose
    // other runtime (torch) had already carved up the address space got some tensors on small pages, and the layers READING those
    // ran 25-35 % slower on every forward (bench.py's conv3_2 / conv5_1 / conv9_2 / conv8_1 against tools/quick_layers.py, which
    // creates the engine first).  IDC_ARENA=0 restores the per-tensor allocations for A/B.
    static const bool use_arena = idc_env_int("IDC_ARENA", 1) != 0;
    if (use_arena) {
        const size_t al = (size_t)2 << 20;
        size_t total = 0;
        for (auto& t : c->tensors) total += align_up(t.bytes, al);
        HIPCHK(c, hipMalloc(&c->d_arena, total));
        size_t off = 0;
        for (auto& t : c->tensors) { t.ptr = (char*)c->d_arena + off; off += align_up(t.bytes, al); }
    } else {
        for (auto& t : c->tensors) HIPCHK(c, hipMalloc(&t.ptr, t.bytes));
    }
    const size_t hw = (size_t)c->H * c->W, nb = (size_t)c->max_batch;
    HIPCHK(c, hipMalloc((void**)&c->d_L, nb * hw * 4));
    HIPCHK(c, hipMalloc((void**)&c->d_ab, nb * hw * 2 * 4));
    HIPCHK(c, hipMalloc((void**)&c->d_mask, nb * hw * 4));
    HIPCHK(c, hipMalloc((void**)&c->d_out, nb * hw * 2 * 4));
    // resident planes start defined: no hints (ab = 0, mask = 0); an L plane has to be uploaded before a resident forward
    HIPCHK(c, hipMemset(c->d_L, 0, nb * hw * 4));
    HIPCHK(c, hipMemset(c->d_ab, 0, nb * hw * 2 * 4));
    HIPCHK(c, hipMemset(c->d_mask, 0, nb * hw * 4));
    HIPCHK(c, hipMemset(c->d_out, 0, nb * hw * 2 * 4));
    c->l_set.assign(nb, 0);
    HIPCHK(c, hipMalloc(&c->d_zeros, 256));
    HIPCHK(c, hipMemset(c->d_zeros, 0, 256));
    if (c->precision == IDC_BF16) {              // conv_kwave_chain_bf16's grid-barrier counter and its host-visible abort flag
        HIPCHK(c, hipMalloc((void**)&c->d_kw_bar, 1024));
        HIPCHK(c, hipMemset(c->d_kw_bar, 0, 1024));
        HIPCHK(c, hipHostMalloc((void**)&c->h_kw_abort, 64, hipHostMallocMapped));
        *c->h_kw_abort = 0;
    }
    // the memsets above run on the NULL stream, the kernels on the handle's non-blocking stream: every conv launch reads the
    // zero page (out-of-image halo rows), so make the fills complete before the handle can launch anything
    HIPCHK(c, hipStreamSynchronize(nullptr));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_sync, hipEventDisableTiming));
    HIPCHK(c, hipHostMalloc((void**)&c->h_in, nb * hw * 4 * 4, hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc((void**)&c->h_out, nb * hw * 2 * 4, hipHostMallocDefault));
    if (c->flags & IDC_FLAG_DIST_HEAD) {
        const size_t dq = nb * 529 * (hw / 16) * 4;
        HIPCHK(c, hipMalloc((void**)&c->d_dist, dq));
        HIPCHK(c, hipHostMalloc((void**)&c->h_dist, dq, hipHostMallocDefault));
    }
    if (c->flags & IDC_FLAG_DIST313) {
        HIPCHK(c, hipMalloc((void**)&c->d_pred_ab, nb * hw * 2 * 4));
        HIPCHK(c, hipHostMalloc((void**)&c->h_pred_ab, nb * hw * 2 * 4, hipHostMallocDefault));
        HIPCHK(c, hipMalloc((void**)&c->d_dist313, nb * hw * 313 * 4));      // 82 MB per 256x256 image: sized for 288 GB parts
    }
    if (c->flags & IDC_FLAG_GLOBAL_HINTS) {
        HIPCHK(c, hipMalloc((void**)&c->d_glob_in, nb * kGlobIn * 4));
        HIPCHK(c, hipMalloc((void**)&c->d_glob_vec, nb * kGlobC * 4));
        HIPCHK(c, hipMemset(c->d_glob_in, 0, nb * kGlobIn * 4));
    }
    c->n_timed = (int)c->layers.size() + 3;           // profiling events (2240 per handle) are created by idc_set_profiling, not here:
    return IDC_OK;                                    // a process holding many handles must not exhaust the runtime's signal pool
}

static int check_chain_abort(idc_context* c);

static int run_graph(idc_context* c, int n, const float* dL, const float* dab, const float* dmask, float maskcent,
                     float* dout, float* ddist) {
    hipStream_t s = c->stream;
    int step = 0;
    {   // a workgroup of an EARLIER conv_kwave_chain_bf16 launch gave up at its grid barrier and nobody has waited on that forward since
        const int arc = check_chain_abort(c);
        if (arc) return arc;
    }
    const size_t ring = (size_t)(c->prof_count % kProfRing) * c->n_timed * 2;
    auto tic = [&]() { if (c->profiling == 1) (void)hipEventRecord(c->ev[ring + step * 2], s); };
    auto toc = [&]() { if (c->profiling == 1) (void)hipEventRecord(c->ev[ring + step * 2 + 1], s); ++step; };
    if (c->profiling == 2) (void)hipEventRecord(c->ev[ring], s);          // whole-forward pair: slot 0
    tic();   // (slot 0: the input pack is fused into conv1_1's operand staging; only the global-hints branch runs here)
    if (c->flags & IDC_FLAG_GLOBAL_HINTS)      // four GEMVs per image; its output is consumed by conv4_3's epilogue
        HIPCHK(c, launch_glob_branch(c->d_glob_in, (const float*)(c->d_blob + c->plan.glob_off), c->d_glob_vec, n, s));
    toc();
    // pass 1: kernel variant per layer, then which shortcut convs ride in their consumer's launch
    for (auto& L : c->layers) {
        const Tensor& ti = c->tensors[L.src];
        const Tensor& to = c->tensors[L.dst];
        const int Hs = L.spec->kind == kDeconv4x4 ? ti.H : to.H;
        const int Ws = L.spec->kind == kDeconv4x4 ? ti.W : to.W;
        set_geometry(L, L.lprec, n, c->max_batch, Hs, Ws);
        L.fused_short = -1; L.skip = false; L.fused_next = -1;
    }
    // model1 in one launch: conv1_1 (input pack fused) followed by conv1_2 on the small-tile bf16 path, >= 128 big tiles
    for (size_t i = 0; i + 1 < c->layers.size(); ++i) {
        Layer& L = c->layers[i];
        const bool f16blk = c->precision == IDC_FP16 && g_fp16_fast && L.blob.w2_off != (size_t)-1;      // IDC_FP16: conv1_block_fused_th
        if (L.spec->kind != kConvIm2col || (c->precision != IDC_BF16 && !f16blk) || !g_fuse_conv1 || L.spec->act != 1 || L.spec->bnkey) continue;
        // 32x32 tiles when there are >= 128 of them (N = 32); else the 32x8 tile (conv1_block_fused_t<4,2>) when THAT gives >= 128
        // workgroups -- the batch-1 click path: one launch instead of conv1_1 + conv1_2 and no 8 MB intermediate
        const long long t32 = (long long)((c->W + 31) / 32) * ((c->H + 31) / 32) * c->max_batch;
        const long long t8 = (long long)((c->W + 31) / 32) * ((c->H + 7) / 8) * c->max_batch;
        if (t32 < 128 && t8 < 128) continue;
        const int tile_req = t32 >= 128 ? 32 : 8;              // tile height request of launch_conv1_block
        for (size_t j = 0; j < c->layers.size(); ++j) {
            Layer& P = c->layers[j];
            const LayerSpec& ps = *P.spec;
            if (P.src != L.dst || (P.v2 && !f16blk) || ps.kind != kConv3x3 || ps.cin != 64 || ps.cout != 64 || ps.dilation != 1 ||
                ps.in_stride != 1 || ps.act != 1 || ps.resid || c->tensors[P.dst].is_f32 || P.args.ksplit > 1 || P.click || P.wino) continue;
            bool only_consumer = true;
            for (const Layer& Q : c->layers) if (&Q != &P && (Q.src == L.dst || Q.resid == L.dst)) only_consumer = false;
            if (only_consumer) { L.fused_next = (int)j; P.skip = true; L.args.tiles_y = tile_req; }
        }
    }
    for (auto& L : c->layers) {
        if (L.spec->kind != kDeconv4x4 || L.resid < 0 || !L.v2 || !fuse_shortcut_enabled() || (L.split && !g_split_ds_fuse)) continue;
        if (L.spec->cout % 128 != 0 || L.spec->bnkey || L.spec->act == 2 || c->tensors[L.dst].is_f32) continue;   // conv_ds_fused's domain
        for (size_t j = 0; j < c->layers.size(); ++j) {
            Layer& P = c->layers[j];
            const LayerSpec& ps = *P.spec;
            if (P.dst != L.resid) continue;
            const Tensor& pin = c->tensors[P.src];
            const Tensor& to = c->tensors[L.dst];
            if (L.split) {      // operand-split: conv_ds_fused_ms (split tensors in and out, both of the same part count) or two launches
                if (ps.kind == kConv3x3 && ps.dilation == 1 && ps.in_stride == 1 && ps.act == 0 && !ps.bnkey && !ps.resid && P.split &&
                    ps.cout == L.spec->cout && pin.H == to.H && pin.W == to.W && !pin.is_f32 && pin.parts == to.parts &&
                    c->tensors[L.src].parts == to.parts && !c->tensors[L.src].is_f32 &&
                    conv_ds_m_fits(L.args.Hs, L.args.Ws, L.blob.nkc * to.parts, P.blob.nkc * to.parts)) {
                    L.fused_short = (int)j; P.skip = true;
                }
                continue;
            }
            if (ps.kind == kConv3x3 && ps.dilation == 1 && ps.in_stride == 1 && ps.act == 0 && !ps.bnkey && !ps.resid &&
                ps.cout == L.spec->cout && pin.H == to.H && pin.W == to.W && !pin.is_f32 && (!kAbPartners || P.blob.w2_off != (size_t)-1) &&
                (kAbPartners || (g_ds_m16 != 0 && conv_ds_m_fits(L.args.Hs, L.args.Ws, L.blob.nkc, P.blob.nkc)))) {      // (default library: conv_ds_fused_m or two launches)
                L.fused_short = (int)j; P.skip = true;
            }
        }
    }
    if (!kAbPartners)            // a large-tile deconv that keeps its shortcut SUM (not fused: images beyond 32-bit offsets) has no 16x16x32 kernel: small tile
        for (auto& L : c->layers)
            if (L.v2 && !L.split && L.resid >= 0 && L.fused_short < 0) {
                const Tensor& ti = c->tensors[L.src];
                const Tensor& to = c->tensors[L.dst];
                set_geometry(L, L.lprec, n, c->max_batch, L.spec->kind == kDeconv4x4 ? ti.H : to.H, L.spec->kind == kDeconv4x4 ? ti.W : to.W, false);
            }
    bool head_done = false;
    int chain_until = -1;                      // layers up to this index ran inside the conv_kwave_chain_bf16 launch of an earlier layer
    for (auto& L : c->layers) {
        const Tensor& ti = c->tensors[L.src];
        const Tensor& to = c->tensors[L.dst];
        const int li_ = (int)(&L - &c->layers[0]);
        L.chain_len = 0;
        if (li_ > chain_until) L.chained_into = -1;
        if (L.skip || li_ <= chain_until) { tic(); toc(); continue; }
        ConvArgs& a = L.args;
        a.in = ti.ptr; a.out = to.ptr;
        a.zeros = c->d_zeros;
        a.warm = g_code_warm;
        a.split_f16 = split_is_f16(c->precision) ? 1 : 0;
        a.acc_scale = (L.split && a.split_f16 && L.blob.wscale_off != (size_t)-1) ? (const float*)(c->d_blob + L.blob.wscale_off) : nullptr;
        a.out_f32 = to.is_f32;
        a.img_shift = ((c->flags & IDC_FLAG_GLOBAL_HINTS) && L.dst == c->t_conv4_3) ? c->d_glob_vec : nullptr;
        if (L.spec->kind == kConvIm2col) {          // model.py:139-148 input pack, fused into the operand staging
            a.pk_L = dL; a.pk_ab = dab; a.pk_mask = dmask;
            a.pk_ldiv = c->l_div; a.pk_abdiv = c->ab_div; a.pk_mmul = c->mask_mul; a.pk_mcent = maskcent;
        } else {
            a.pk_L = nullptr;
        }
        a.wgt = c->d_blob + (L.wino ? L.blob.w3_off : (L.v2 && L.blob.w2_off != (size_t)-1) ? L.blob.w2_off : L.blob.w_off);
        a.bias = (const float*)(c->d_blob + L.blob.bias_off);
        a.bn_scale = L.blob.bn_scale_off != (size_t)-1 ? (const float*)(c->d_blob + L.blob.bn_scale_off) : nullptr;
        a.bn_shift = L.blob.bn_shift_off != (size_t)-1 ? (const float*)(c->d_blob + L.blob.bn_shift_off) : nullptr;
        if (L.fused_short >= 0) {            // model8up(.) + model3short8(.) in one K loop (model.py:156,170,172)
            const Layer& P = c->layers[L.fused_short];
            a.resid = nullptr; a.resid_bf16 = 0;
            a.in2 = c->tensors[P.src].ptr; a.wgt2 = c->d_blob + (P.blob.w2_off != (size_t)-1 ? P.blob.w2_off : P.blob.w_off); a.nkc2 = P.blob.nkc;
            a.bias = (const float*)(c->d_blob + L.blob.fbias_off);
            L.m16 = g_ds_m16 != 0 && conv_ds_m_fits(a.Hs, a.Ws, a.nkc, P.blob.nkc);   // (huge images: conv_ds_fused, 64-bit addressing)
            if (L.m16) { a.wgt = c->d_blob + L.blob.w_off; a.wgt2 = c->d_blob + P.blob.w_off; }   // conv_ds_fused_m reads the layout-1 images
            if (L.split) { L.m16 = true; a.wgt = c->d_blob + L.blob.w_off; a.wgt2 = c->d_blob + P.blob.w_off; a.w_part_bytes2 = P.blob.w_bytes; }
        } else {
            a.resid = L.resid >= 0 ? c->tensors[L.resid].ptr : nullptr;
            a.resid_bf16 = (L.resid >= 0 && !c->tensors[L.resid].is_f32) ? 1 : 0;
            a.in2 = nullptr; a.wgt2 = nullptr; a.nkc2 = 0;
        }
        // the regression head rides in conv10_2's epilogue when one workgroup owns all 128 channels
        L.fused_head = L.dst == c->t_conv10_2 && L.v2 && L.cfg.wm == 2 && a.ncg == 2 && L.spec->bnkey == nullptr;
        a.head_w = L.fused_head ? (const float*)(c->d_blob + c->plan.head_w_off) : nullptr;
        a.head_b = (const float*)(c->d_blob + c->plan.head_b_off);
        a.head_out = dout; a.head_mul = c->out_mul;
        head_done = head_done || L.fused_head;
        if (L.fused_next >= 0) {             // conv1_2 rides in conv1_1's launch (launch_conv1_block's argument convention)
            const Layer& P = c->layers[L.fused_next];
            a.wgt2 = c->d_blob + P.blob.w_off;
            a.head_b = (const float*)(c->d_blob + P.blob.bias_off);
            a.bn_scale = P.blob.bn_scale_off != (size_t)-1 ? (const float*)(c->d_blob + P.blob.bn_scale_off) : nullptr;
            a.bn_shift = P.blob.bn_shift_off != (size_t)-1 ? (const float*)(c->d_blob + P.blob.bn_shift_off) : nullptr;
            a.out = c->tensors[P.dst].ptr;
            a.head_w = nullptr;
        }
        if (a.ksplit > 1) {
            const size_t need = (size_t)a.ksplit * n * to.H * to.W * to.Cpad * 4;
            if (c->partial_bytes < need) {
                HIPCHK(c, hipStreamSynchronize(s));
                if (c->d_partial) (void)hipFree(c->d_partial);
                c->d_partial = nullptr; c->partial_bytes = 0;
                HIPCHK(c, hipMalloc((void**)&c->d_partial, need));
                c->partial_bytes = need;
            }
            a.partial = c->d_partial;
        }
        if (L.fused_next >= 0 && c->precision == IDC_FP16) {
            a.wgt = c->d_blob + L.blob.w2_off;               // conv1_1's fp16 block (conv1_block_fused_th)
            a.out_parts = 0; a.ksplit = 1; a.kc_per = a.nkc;
        } else
        if (is_split(c->precision) && !L.split) {        // fp32 island: conv_igemm<f32> with a split store (no split-K: its epilogue kernel writes fp32)
            a.out_parts = to.is_f32 ? 0 : to.parts;
            a.out_f32 = 1; a.ksplit = 1; a.kc_per = a.nkc;
            if (L.wino || L.click || L.v2 || L.kw) return fail(&c->err, IDC_ERR_INTERNAL, "layer %s: fp32 island outside conv_igemm", L.spec->name);
        }
        if (L.split) {
            // operand-split launch: nseg passes of the K loop (input part x weight part) into one accumulator set; split in / out tensors
            a.wgt = c->d_blob + L.blob.w_off;
            a.in_parts = ti.parts; a.out_parts = (to.is_f32 || L.fused_head) ? 0 : to.parts;
            a.nseg = split_segments(c->precision); a.seg_x = split_seg_x(c->precision); a.seg_w = split_seg_w(c->precision);
            a.w_part_bytes = L.blob.w_bytes;
            if (L.fused_short < 0 && (!L.v2 || !conv_v2s_applies(a)))
                return fail(&c->err, IDC_ERR_INTERNAL, "layer %s: no operand-split kernel covers this launch", L.spec->name);
            L.m16 = true;
            L.v2p = L.fused_short < 0 && g_v2p && conv_v2ps_applies(L.cfg, L.halo, a);
            // IDC_FP16: where the bf16 forward's own kernels cover the launch, their fp16 twins (same tile, bias in the accumulators, packed-pair epilogue)
            L.f16fast = c->precision == IDC_FP16 && g_fp16_fast && !to.is_f32 &&
                        (L.fused_short >= 0 ? (L.spec->act != 2 && !L.spec->bnkey && (a.ncg & 1) == 0) : (L.v2 && g_v2p && conv_v2p_applies(L.cfg, L.halo, a)));
        } else if (L.fused_short < 0) {
            L.m16 = L.v2 && g_mfma16 && L.fused_next < 0 && !L.wino && !L.click && conv_v2m_applies(a);
            if (L.m16) a.wgt = c->d_blob + L.blob.w_off;            // the layout-1 image (the one conv_igemm / conv_click read)
            L.v2p = L.m16 && g_v2p && conv_v2p_applies(L.cfg, L.halo, a);
        }
        // diagnostic (IDC_DOUBLE_LAUNCH=1): every launch issued twice, the event pair around the SECOND -- a layer that is slow only as
        // the first launch of its kernel after other kernels (cold instruction cache / first touch) shows its warm time here
        static const bool double_launch = idc_env_int("IDC_DOUBLE_LAUNCH", 0) != 0;
        for (int rep = double_launch ? 0 : 1; rep < 2; ++rep) {
        if (rep == 1) tic();
        {
            hipError_t le = hipErrorInvalidConfiguration;
            // conv1_1 with >= 128 big tiles: the 32x32-tile form (the small-tile kernel keeps the batch-1 click path).
            // Chosen by the handle's max_batch like every other kernel variant, so a result never depends on how many
            // images share the call.
            if (L.fused_next >= 0) le = launch_conv1_block(a, s);
            else if (L.spec->kind == kConvIm2col && L.lprec == IDC_BF16 && a.ksplit <= 1 &&
                (long long)((a.Ws + 31) / 32) * ((a.Hs + 31) / 32) * c->max_batch >= 128)
                le = launch_conv1_1_bf16(a, s);
            if (is_split(c->precision) && !L.split && L.fused_next < 0 && L.spec->kind == kConvIm2col && g_conv1_1_split && a.out_parts >= 1 &&
                (long long)((a.Ws + 31) / 32) * ((a.Hs + 15) / 16) * c->max_batch >= 128)
                le = launch_conv1_1_split(a, s);
            if (L.split && L.f16fast) le = L.fused_short >= 0 ? launch_conv_ds_m(a, s) : launch_conv_v2p(L.cfg, L.halo, a, s);
            else
            if (L.split && L.fused_short < 0 && g_conv1_2_split && conv1_2_split_layer(*L.spec) && !to.is_f32 &&
                (long long)((a.Ws + 31) / 32) * ((a.Hs + 11) / 12) * c->max_batch >= 256)
                le = launch_conv1_2_split(a, s);
            if (L.fused_short >= 0 && !(L.split && L.f16fast)) le = L.split ? launch_conv_ds_ms(a, s) : L.m16 ? launch_conv_ds_m(a, s) : launch_conv_ds(a, s);
            if (L.fused_short >= 0 && L.split && le == hipErrorInvalidConfiguration)
                return fail(&c->err, IDC_ERR_INTERNAL, "layer %s: conv_ds_fused_ms planned for a launch it does not cover", L.spec->name);
    // deconv + its shortcut conv in one K loop
            if (L.wino) {
                // a.wgt points at the Winograd U image and L.cfg / tiles were never set for this layer: a refused launch must not fall
                // through to the direct kernels below -- it is a variant-selection bug and says so
                if (!conv_wino_applies(L.lprec, a, L.spec->kind == kDeconv4x4))
                    return fail(&c->err, IDC_ERR_INTERNAL, "layer %s: Winograd variant selected for a launch it does not cover", L.spec->name);
                le = L.spec->kind == kDeconv4x4 ? launch_deconv_wino(L.lprec, a, s) : launch_conv_wino(L.lprec, a, s);
                HIPCHK(c, le);
            }
            if (L.kw) {
                if (!conv_kwave_applies(a))
                    return fail(&c->err, IDC_ERR_INTERNAL, "layer %s: conv_kwave_bf16 selected for a launch it does not cover", L.spec->name);
                le = hipErrorInvalidConfiguration;
                // the run of same-shape 8-chunk layers that starts here (conv4_2 .. conv7_3 at batch 1) as ONE persistent launch
                if (g_kwave_chain && !c->kw_chain_off && c->profiling != 1 && !double_launch && c->d_kw_bar && L.spec->kind == kConv3x3 && a.nkc == 8 &&
                    a.si == 1 && a.img_shift == nullptr && !a.out_f32) {
                    const int blocks = conv_kwave_chain_blocks(a.Hs, a.Ws, a.N, a.ncg, a.dy[8]);
                    KwChainArgs ch{};
                    ch.H = a.Hs; ch.W = a.Ws; ch.N = a.N; ch.ncg = a.ncg;
                    ch.spin_limit = 200000u;                   // x ~1.5 us per poll: a third of a second, then the workgroup gives up
                    int last = li_, prev_dst = L.src;
                    for (int j = li_; j < (int)c->layers.size() && ch.nlayers < kKwChainMax; ++j) {
                        Layer& Q = c->layers[j];
                        const bool shifted = (c->flags & IDC_FLAG_GLOBAL_HINTS) && Q.dst == c->t_conv4_3;
                        if (Q.skip || !Q.kw || Q.spec->kind != kConv3x3 || Q.blob.nkc != 8 || Q.spec->in_stride != 1 || Q.resid >= 0 || shifted ||
                            c->tensors[Q.dst].is_f32 || Q.src != prev_dst || Q.args.Hs != a.Hs || Q.args.Ws != a.Ws || Q.args.ncg != a.ncg ||
                            conv_kwave_chain_blocks(a.Hs, a.Ws, a.N, a.ncg, Q.args.dy[8]) != blocks)
                            break;
                        KwChainLayer& y = ch.layer[ch.nlayers++];
                        y.in = c->tensors[Q.src].ptr; y.out = c->tensors[Q.dst].ptr;
                        y.wgt = c->d_blob + Q.blob.w_off;
                        y.bias = (const float*)(c->d_blob + Q.blob.bias_off);
                        y.bn_scale = Q.blob.bn_scale_off != (size_t)-1 ? (const float*)(c->d_blob + Q.blob.bn_scale_off) : nullptr;
                        y.bn_shift = Q.blob.bn_shift_off != (size_t)-1 ? (const float*)(c->d_blob + Q.blob.bn_shift_off) : nullptr;
                        y.act = Q.spec->act; y.d = Q.args.dy[8];
                        prev_dst = Q.dst; last = j;
                    }
                    if (ch.nlayers >= 2 && blocks > 0) {
                        if (c->kw_chain_fits < 0) c->kw_chain_fits = conv_kwave_chain_capacity(c->device);
                        if (c->kw_chain_fits >= blocks) {
                            if (blocks != c->kw_bar_blocks) {      // counters hold (barriers so far) x (arrivals per barrier of THIS grid)
                                HIPCHK(c, hipMemsetAsync(c->d_kw_bar, 0, 1024, s));
                                c->kw_bar_count = 0; c->kw_bar_blocks = blocks;
                            }
                            ch.bar = c->d_kw_bar; ch.bar_base = c->kw_bar_count; ch.abort_flag = c->h_kw_abort;
                            // test hook (tests/test_round5_gpu.py): IDC_KW_FORCE_ABORT=1 makes the first grid barrier unreachable and the give-up
                            // counter tiny, i.e. it plays "the workgroups never become co-resident" on a healthy device
                            if (g_kw_force_abort) { ch.bar_base += 1000; ch.spin_limit = 20u; }
                            static const bool want_stamps = idc_env_int("IDC_KW_STAMPS", 0) != 0;
                            if (want_stamps && !c->d_kw_stamps) HIPCHK(c, hipMalloc((void**)&c->d_kw_stamps, (size_t)4096 * kKwChainMax * 8 * 8));
                            ch.stamps = (want_stamps && blocks <= 4096) ? c->d_kw_stamps : nullptr;
                            c->kw_stamp_layers = ch.nlayers; c->kw_stamp_blocks = blocks;
                            const hipError_t ce = launch_conv_kwave_chain(ch, blocks, g_kwave_chain, s);
                            if (ce == hipSuccess) {
                                c->kw_bar_count += (unsigned long long)(ch.nlayers - 1);
                                L.chain_len = ch.nlayers;
                                for (int j = li_ + 1; j <= last; ++j) c->layers[j].chained_into = li_;
                                chain_until = last;
                                le = hipSuccess;
                            } else {
                                (void)hipGetLastError();          // refused (e.g. hipErrorCooperativeLaunchTooLarge): one launch per layer from now on
                                c->kw_chain_off = true;
                            }
                        }                                          // (else: more workgroups than the chip holds at once -- this launch goes layer by layer)
                    }
                }
                if (le != hipSuccess) le = launch_conv_kwave(a, s);
                HIPCHK(c, le);
            }
            if (!kAbPartners && le == hipErrorInvalidConfiguration && !L.split && !L.click && L.v2 && !L.m16)
                return fail(&c->err, IDC_ERR_INTERNAL, "layer %s: planned on the large tile but no 16x16x32 kernel covers this launch (default library: no "
                            "conv_igemm_v2 / conv_ds_fused; build with -DIDC_AB_PARTNERS)", L.spec->name);
            if (le == hipErrorInvalidConfiguration)
                le = L.split ? (L.v2p ? launch_conv_v2ps(L.cfg, L.halo, a, s) : launch_conv_v2s(L.cfg, L.halo, a, s))
                   : L.click ? launch_conv_click(L.lprec, L.cfg.wp, L.halo, a, s)
                   : L.v2 ? (L.v2p ? launch_conv_v2p(L.cfg, L.halo, a, s) : L.m16 ? launch_conv_v2m(L.cfg, L.halo, a, s) : launch_conv_v2(L.cfg, L.halo, a, s))
                          : launch_conv(L.lprec, L.cfg, L.halo, a, s);
            HIPCHK(c, le);
        }
        if (a.ksplit > 1) HIPCHK(c, launch_splitk_epilogue(L.lprec, a, s));
        }
        toc();
    }
    tic();
    if (!head_done && is_split(c->precision))
        return fail(&c->err, IDC_ERR_INTERNAL, "operand-split forward: the regression head did not ride in conv10_2's launch");
    if (!head_done)
        HIPCHK(c, launch_head(c->precision, c->tensors[c->t_conv10_2].ptr, (const float*)(c->d_blob + c->plan.head_w_off),
                              (const float*)(c->d_blob + c->plan.head_b_off), dout, n, c->H, c->W, c->out_mul, s));
    toc();
    tic();
    if (ddist) {
        const Tensor& tl = c->tensors[c->t_logits];
        HIPCHK(c, launch_softmax_nchw((const float*)tl.ptr, ddist, n, tl.H, tl.W, 529, tl.Cpad, 0.2f, s));
    }
    if (c->flags & IDC_FLAG_DIST313) {         // bilinear x4 + softmax(S.) + softmax(2.6 .) -> pred_ab decode
        const Tensor& tp = c->tensors[c->t_pred313];
        HIPCHK(c, launch_dist313((const float*)tp.ptr, (const float*)(c->d_blob + c->plan.pred_ab_off), c->d_pred_ab,
                                 (c->want_dist313 || c->keep_dist313) ? c->d_dist313 : nullptr, n, c->H, c->W, tp.Cpad,
                                 c->dist_S, 2.6f, s));
    }
    toc();
    c->dist_n = (ddist || ((c->flags & IDC_FLAG_DIST313) && (c->want_dist313 || c->keep_dist313))) ? n : 0;
    if (dout == c->d_out) c->last_n = n;       // images whose ab map sits in the handle's own d_out (display step source)
    if (c->profiling == 2) (void)hipEventRecord(c->ev[ring + 1], s);
    if (c->profiling) ++c->prof_count;
    return IDC_OK;
}

// A conv_kwave_chain_bf16 workgroup that gave up at its grid barrier (not all workgroups co-resident: a partitioned or shared device) set the pinned
// flag: the results of the forward that has just been waited for are invalid.  Called after the synchronising points of the blocking entry points.
static int check_chain_abort(idc_context* c) {
    if (c->h_kw_abort && *c->h_kw_abort) {
        *c->h_kw_abort = 0;
        c->kw_chain_off = true;
        return fail(&c->err, IDC_ERR_INTERNAL, "the persistent trunk launch (conv_kwave_chain_bf16) did not get all its workgroups co-resident and timed "
                    "out: this forward's result is invalid; the handle now runs one launch per layer -- call again");
    }
    return IDC_OK;
}

static int check_forward_args(idc_context* c, int n) {
    if (!c) return fail(nullptr, IDC_ERR_INVALID_ARG, "null handle");
    if (!c->weights_set) return fail(&c->err, IDC_ERR_NO_WEIGHTS, "I need to have a net! (no weights loaded)");
    if (n <= 0 || n > c->max_batch) return fail(&c->err, IDC_ERR_BATCH, "batch %d outside 1..%d", n, c->max_batch);
    return IDC_OK;
}

static int drain_pipeline(idc_context* c);
static bool is_pinned(const void* p);

// The end of a click: hipStreamSynchronize parks the thread on an interrupt (tens of microseconds to come back, against a 0.3-1.1 ms forward); the calls
// that serve ONE OR TWO images poll the stream instead (bounded: after ~4 ms of polling -- several forwards -- it blocks like everybody else).  Batches
// keep the blocking wait: there the CPU is better spent elsewhere.  IDC_SPIN_SYNC=0 restores the blocking wait everywhere (A/B).
static hipError_t wait_stream(idc_context* c, int n) {
    const bool spin = g_spin_sync != 0;
    if (spin && n <= 2) {
        for (int i = 0; i < 4000; ++i) {
            const hipError_t e = hipStreamQuery(c->stream);
            if (e != hipErrorNotReady) return e;
#if defined(__x86_64__) || defined(__i386__)
            for (int k = 0; k < 40; ++k) __builtin_ia32_pause();
#else
            for (volatile int k = 0; k < 40; ++k) {}
#endif
        }
    }
    return hipStreamSynchronize(c->stream);
}

// A click's transfers (<= 2 MiB between pinned host memory and HBM) run as a kernel on the forward's stream (pcie_copy_kernel, idc_kernels.hip) instead of
// going to a copy engine: no cross-queue hand-over on either side of them.  `host` must be pinned (ours, or the caller's idc_alloc_host / hipHostMalloc /
// mapped hipHostRegister memory); anything the device cannot address, bigger, or not 16-byte shaped takes hipMemcpyAsync.  IDC_PCIE_KERNEL=0: always (A/B).
static hipError_t copy_h2d_or_d2h(idc_context* c, void* dev, void* host, size_t bytes, bool to_device) {
    const bool by_kernel = g_pcie_kernel != 0;
    if (by_kernel && bytes <= ((size_t)2 << 20) && bytes % 16 == 0 && (((uintptr_t)dev | (uintptr_t)host) & 15) == 0) {
        void* hv = nullptr;
        if (hipHostGetDevicePointer(&hv, host, 0) == hipSuccess && hv)
            return to_device ? launch_pcie_copy(dev, hv, bytes, c->stream) : launch_pcie_copy(hv, dev, bytes, c->stream);
        (void)hipGetLastError();
    }
    return to_device ? hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, c->stream) : hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream);
}

// finish = false: everything is enqueued (the D2H of the ab map into h_out included) but the stream is NOT synchronised and
// nothing is copied to out_ab -- idc_forward_rgb appends the colour step and synchronises once for both
static int forward_host(idc_context* c, int n, const float* L_mc, const float* ab, const float* mask, float maskcent,
                        float* out_ab, float* dist_q, bool keep_dist = false, bool finish = true, bool copy_out = true) {
    int rc = check_forward_args(c, n);
    if (rc) return rc;
    if (!ab || !mask || (copy_out && !out_ab)) return fail(&c->err, IDC_ERR_INVALID_ARG, "null tensor pointer");
    if (!L_mc) {                             // L_mc == NULL: the L planes idc_set_image_l left in the handle (constant between the clicks on one image)
        if (copy_out) return fail(&c->err, IDC_ERR_INVALID_ARG, "null tensor pointer");      // (only idc_forward_rgb_lazy offers this form)
        for (int i = 0; i < n; ++i)
            if (!c->l_set[i]) return fail(&c->err, IDC_ERR_INVALID_ARG, "L_mc is NULL and image %d has no resident L plane (idc_set_image_l)", i);
    }
    if ((dist_q || keep_dist) && !(c->flags & IDC_FLAG_DIST_HEAD))
        return fail(&c->err, IDC_ERR_UNSUPPORTED, "handle was created without IDC_FLAG_DIST_HEAD");
    HIPCHK(c, hipSetDevice(c->device));
    rc = drain_pipeline(c);
    if (rc) return rc;
    const size_t hw = (size_t)c->H * c->W;
    if (L_mc) for (int i = 0; i < n; ++i) c->l_set[i] = 1;
    c->out_resident = true; c->labq_resident = false;
    // Pinned caller buffers (idc_alloc_host / hipHostMalloc / hipHostRegister) are transferred in place; pageable ones go through
    // the handle's pinned staging with a host memcpy (2.2 MB per click through the reference API: most of its host-side time).
    // Staged in pieces (>= 256 KiB, at most four per tensor), each piece's DMA issued as soon as it is in the staging buffer: the copy engine moves piece k
    // while the CPU copies piece k + 1, so a click pays the host memcpy plus ONE piece of DMA instead of memcpy + all of it.
    float* hL = c->h_in; float* hab = hL + (size_t)n * hw; float* hm = hab + (size_t)n * hw * 2;
    auto upload = [&](void* dst, const float* src, float* staging, size_t bytes) -> hipError_t {
        if (is_pinned(src)) return copy_h2d_or_d2h(c, dst, (void*)src, bytes, true);
        size_t piece = (bytes + 3) / 4;
        if (piece < (size_t)256 * 1024) piece = (size_t)256 * 1024;
        piece = (piece + 4095) & ~(size_t)4095;
        for (size_t o = 0; o < bytes; o += piece) {
            const size_t m = bytes - o < piece ? bytes - o : piece;
            memcpy((char*)staging + o, (const char*)src + o, m);
            const hipError_t e = copy_h2d_or_d2h(c, (char*)dst + o, (char*)staging + o, m, true);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    };
    if (L_mc) HIPCHK(c, upload(c->d_L, L_mc, hL, (size_t)n * hw * 4));
    HIPCHK(c, upload(c->d_ab, ab, hab, (size_t)n * hw * 2 * 4));
    HIPCHK(c, upload(c->d_mask, mask, hm, (size_t)n * hw * 4));
    rc = run_graph(c, n, c->d_L, c->d_ab, c->d_mask, maskcent, c->d_out, (dist_q || keep_dist) ? c->d_dist : nullptr);
    if (rc) return rc;
    const bool out_direct = copy_out && out_ab != c->h_out && is_pinned(out_ab);
    c->out_copy_pending = copy_out && out_ab != c->h_out && !out_direct;
    if (copy_out)          // (copy_out = false: the ab map stays in d_out -- idc_fetch_outputs brings it over when somebody asks)
        HIPCHK(c, copy_h2d_or_d2h(c, c->d_out, out_direct ? out_ab : c->h_out, (size_t)n * hw * 2 * 4, false));
    const size_t dq = (size_t)n * 529 * (hw / 16) * 4;
    if (dist_q) HIPCHK(c, hipMemcpyAsync(c->h_dist, c->d_dist, dq, hipMemcpyDeviceToHost, c->stream));
    if (!finish) return IDC_OK;
    HIPCHK(c, wait_stream(c, n));
    rc = check_chain_abort(c);
    if (rc) return rc;
    if (c->out_copy_pending) memcpy(out_ab, c->h_out, (size_t)n * hw * 2 * 4);
    c->out_copy_pending = false;
    if (dist_q) memcpy(dist_q, c->h_dist, dq);
    return IDC_OK;
}

static void destroy_ctx(idc_context* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->d_arena) (void)hipFree(c->d_arena);
    else for (auto& t : c->tensors) if (t.ptr) (void)hipFree(t.ptr);
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->own_blob && c->d_blob) (void)hipFree(c->d_blob);
    for (auto& sl : c->pipe) {
        void* dv[] = {sl.d_L, sl.d_ab, sl.d_mask, sl.d_out};
        for (void* p : dv) if (p) (void)hipFree(p);
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        if (sl.h_out) (void)hipHostFree(sl.h_out);
        hipEvent_t evs[] = {sl.ev_in, sl.ev_comp, sl.ev_out, sl.ev_in0, sl.ev_comp0, sl.ev_out0};
        for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
    }
    if (c->ev_pipe_base) (void)hipEventDestroy(c->ev_pipe_base);
    if (c->s_in) (void)hipStreamDestroy(c->s_in);
    if (c->s_out) (void)hipStreamDestroy(c->s_out);
    if (c->ev_sync) (void)hipEventDestroy(c->ev_sync);
    if (c->d_zeros) (void)hipFree(c->d_zeros);
    if (c->d_kw_stamps && c->kw_stamp_blocks > 0) {      // diagnostic: where a layer of the last chain launch spent its cycles (mean over workgroups)
        std::vector<long long> st((size_t)c->kw_stamp_blocks * kKwChainMax * 8);
        if (hipMemcpy(st.data(), c->d_kw_stamps, st.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            static const char* names[] = {"prefetch+barrier wait", "halo issue", "halo landed", "taps", "reduce+epilogue", "stores acked+wg barrier"};
            static const int seq[] = {0, 1, 6, 2, 3, 4, 5};
            for (int li = 0; li < c->kw_stamp_layers; ++li) {
                double d[6] = {0, 0, 0, 0, 0, 0};
                for (int b = 0; b < c->kw_stamp_blocks; ++b) {
                    const long long* p = &st[((size_t)b * kKwChainMax + li) * 8];
                    for (int k = 0; k < 6; ++k) {
                        if (li + 1 == c->kw_stamp_layers && k == 5) continue;
                        d[k] += (double)(p[seq[k + 1]] - p[seq[k]]);
                    }
                }
                fprintf(stderr, "kw_chain stamps layer %2d:", li);
                for (int k = 0; k < 6; ++k) fprintf(stderr, " %s %.0f |", names[k], d[k] / c->kw_stamp_blocks);
                fprintf(stderr, "\n");
            }
        }
        (void)hipFree(c->d_kw_stamps);
    }
    if (c->d_kw_bar) (void)hipFree(c->d_kw_bar);
    if (c->h_kw_abort) (void)hipHostFree(c->h_kw_abort);
    if (c->d_up_rgb) (void)hipFree(c->d_up_rgb);
    if (c->d_up_L) (void)hipFree(c->d_up_L);
    if (c->h_up_rgb) (void)hipHostFree(c->h_up_rgb);
    if (c->h_up_L) (void)hipHostFree(c->h_up_L);
    void* dev[] = {c->d_L, c->d_ab, c->d_mask, c->d_out, c->d_dist, c->d_scratch, c->d_glob_in, c->d_glob_vec, c->d_pred_ab, c->d_dist313, c->d_partial, c->d_rgb, c->d_labq, c->d_hints, c->d_centres, c->d_sugg, c->d_sugg_counts, c->d_post_in};
    for (void* p : dev) if (p) (void)hipFree(p);
    void* host[] = {c->h_in, c->h_out, c->h_dist, c->h_pred_ab, c->h_rgb, c->h_labq, c->h_hints};
    for (void* p : host) if (p) (void)hipHostFree(p);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

static int check_device(int device_id, std::string* err) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(err, IDC_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (device_id < 0 || device_id >= count) return fail(err, IDC_ERR_NO_DEVICE, "device %d not in 0..%d", device_id, count - 1);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return fail(err, IDC_ERR_HIP, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(err, IDC_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device_id, prop.gcnArchName);
    return IDC_OK;
}

// ================================================================================================
extern "C" {

int idc_version(void) { return IDC_VERSION; }

int idc_set_splitk_policy(int policy) {
    if (policy < 0 || policy > 2) return fail(nullptr, IDC_ERR_INVALID_ARG, "split-K policy %d not in 0..2", policy);
    g_splitk_policy = policy;
    return IDC_OK;
}

int idc_set_tile_policy(int policy) {
    if (policy < 0 || policy > 2) return fail(nullptr, IDC_ERR_INVALID_ARG, "tile policy %d not in 0..2", policy);
    g_tile_policy = policy;
    return IDC_OK;
}

int idc_set_option(const char* name, int value) {
    if (!name) return fail(nullptr, IDC_ERR_INVALID_ARG, "null option name");
    if (strcmp(name, "fuse_conv1") == 0) { g_fuse_conv1 = value != 0; return IDC_OK; }
    if (strcmp(name, "click") == 0) { g_click = value; return IDC_OK; }
    if (strcmp(name, "winograd") == 0) {         // 0 off | 1 automatic | 2 everywhere implemented | 12 / 21 / 22 automatic, 3x3 form <TB,CB> forced
        if (value != 0 && value != 1 && value != 2 && value != 12 && value != 21 && value != 22)
            return fail(nullptr, IDC_ERR_INVALID_ARG, "option 'winograd': %d not in {0, 1, 2, 12, 21, 22}", value);
        g_wino = value != 0;
        g_wino_deconv = value == 2 ? 2 : 1;
        set_wino_form(value >= 12 ? value : 0);
        return IDC_OK;
    }
    if (!kAbPartners && ((strcmp(name, "mfma16") == 0 && value == 0) || (strcmp(name, "ds_mfma16") == 0 && value == 0)))
        return fail(nullptr, IDC_ERR_UNSUPPORTED, "option '%s' = 0 selects a 32x32x16-MFMA partner kernel: not in the default library (build with make EXTRA=-DIDC_AB_PARTNERS)", name);
    if (strcmp(name, "mfma16") == 0) { g_mfma16 = value != 0; return IDC_OK; }
    if (strcmp(name, "v2p") == 0) { g_v2p = value != 0; return IDC_OK; }
    if (strcmp(name, "ds_mfma16") == 0) { g_ds_m16 = value != 0; set_ds_half(value != 2); return IDC_OK; }     // 2: conv_ds_fused_m, 8-wave workgroups on every grid
    if (strcmp(name, "split_ds_fuse") == 0) { g_split_ds_fuse = value != 0; return IDC_OK; }
    if (strcmp(name, "conv1_1_split") == 0) { g_conv1_1_split = value != 0; return IDC_OK; }
    if (strcmp(name, "conv1_2_split") == 0) { g_conv1_2_split = value != 0; return IDC_OK; }
    if (strcmp(name, "fp16_fast") == 0) { g_fp16_fast = value != 0; return IDC_OK; }
    if (strcmp(name, "kwave") == 0) { g_kwave = value != 0; return IDC_OK; }
    if (strcmp(name, "spin_sync") == 0) { g_spin_sync = value != 0; return IDC_OK; }
    if (strcmp(name, "pcie_kernel") == 0) { g_pcie_kernel = value != 0; return IDC_OK; }
    if (strcmp(name, "kw_force_abort") == 0) { g_kw_force_abort = value != 0; return IDC_OK; }
    if (strcmp(name, "kwave_chain") == 0) { g_kwave_chain = value < 0 ? 0 : (value > 2 ? 2 : value); return IDC_OK; }
    return fail(nullptr, IDC_ERR_INVALID_ARG, "unknown option '%s'", name);
}

int idc_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

const char* idc_last_error(idc_handle h) { return h ? h->err.c_str() : g_last_error.c_str(); }

int idc_create(int device_id, int height, int width, int max_batch, int precision, unsigned flags, idc_handle* out) {
    if (!out) return fail(nullptr, IDC_ERR_INVALID_ARG, "null out handle");
    *out = nullptr;
    if (height <= 0 || width <= 0 || height % 8 || width % 8)
        return fail(nullptr, IDC_ERR_INVALID_ARG, "H and W must be positive multiples of 8 (got %dx%d)", height, width);
    if (max_batch <= 0) return fail(nullptr, IDC_ERR_INVALID_ARG, "max_batch must be positive");
    if (precision < IDC_FP32 || precision > IDC_FP16) return fail(nullptr, IDC_ERR_INVALID_ARG, "bad precision %d", precision);
    int rc = check_device(device_id, nullptr);
    if (rc) return rc;
    if (hipSetDevice(device_id) != hipSuccess) return fail(nullptr, IDC_ERR_HIP, "hipSetDevice(%d) failed", device_id);
    idc_context* c = new idc_context();
    c->device = device_id; c->H = height; c->W = width; c->max_batch = max_batch; c->precision = precision; c->flags = flags;
    c->plan = make_blob_plan(precision, flags);
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = init_kernels();
    if (e != hipSuccess) {
        rc = fail(nullptr, IDC_ERR_HIP, "stream/kernel init failed: %s", hipGetErrorString(e));
        destroy_ctx(c);
        return rc;
    }
    rc = build_graph(c);
    if (rc == IDC_OK) rc = alloc_graph(c);
    if (rc != IDC_OK) { g_last_error = c->err; destroy_ctx(c); return rc; }
    *out = c;
    return IDC_OK;
}

int idc_destroy(idc_handle h) {
    if (!h) return fail(nullptr, IDC_ERR_INVALID_ARG, "null handle");
    destroy_ctx(h);
    return IDC_OK;
}

int idc_set_io_scales(idc_handle h, float l_div, float ab_div, float mask_mul, float out_mul) {
    if (!h) return fail(nullptr, IDC_ERR_INVALID_ARG, "null handle");
    if (l_div == 0.f || ab_div == 0.f) return fail(&h->err, IDC_ERR_INVALID_ARG, "zero divisor");
    h->l_div = l_div; h->ab_div = ab_div; h->mask_mul = mask_mul; h->out_mul = out_mul;
    return IDC_OK;
}

size_t idc_weights_blob_bytes(int precision, unsigned flags) {
    if (precision < IDC_FP32 || precision > IDC_FP16) return 0;
    return make_blob_plan(precision, flags).total_bytes;
}

int idc_pack_weights(int precision, unsigned flags, const idc_tensor_desc* tensors, int n_tensors, void* blob,
                     size_t blob_bytes) {
    return pack_weights_impl(precision, flags, tensors, n_tensors, blob, blob_bytes, nullptr);
}

static int validate_header(idc_context* h, const BlobHeader& hd, size_t blob_bytes) {
    if (hd.magic != kBlobMagic || hd.version != IDC_VERSION)
        return fail(&h->err, IDC_ERR_INVALID_ARG, "not an ideepcolor weight blob (bad magic/version)");
    if ((int)hd.precision != h->precision || hd.flags != h->plan.flags)
        return fail(&h->err, IDC_ERR_INVALID_ARG, "blob was packed for precision %u flags %u, handle needs %d/%u",
                    hd.precision, hd.flags, h->precision, h->plan.flags);
    if (hd.total_bytes != h->plan.total_bytes || blob_bytes < h->plan.total_bytes)
        return fail(&h->err, IDC_ERR_INVALID_ARG, "blob size mismatch");
    return IDC_OK;
}

int idc_set_weights_host(idc_handle h, const void* blob, size_t blob_bytes) {
    if (!h || !blob) return fail(h ? &h->err : nullptr, IDC_ERR_INVALID_ARG, "null handle/blob");
    BlobHeader hd;
    if (blob_bytes < sizeof(hd)) return fail(&h->err, IDC_ERR_INVALID_ARG, "blob too small");
    memcpy(&hd, blob, sizeof(hd));
    int rc = validate_header(h, hd, blob_bytes);
    if (rc) return rc;
    if (fnv1a((const uint8_t*)blob + sizeof(hd), h->plan.total_bytes - sizeof(hd)) != hd.checksum)
        return fail(&h->err, IDC_ERR_INVALID_ARG, "blob checksum mismatch");
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->own_blob || !h->d_blob) {
        h->d_blob = nullptr;
        HIPCHK(h, hipMalloc((void**)&h->d_blob, h->plan.total_bytes));
        h->own_blob = true;
    }
    HIPCHK(h, hipMemcpy(h->d_blob, blob, h->plan.total_bytes, hipMemcpyHostToDevice));
    h->weights_set = true;
    return IDC_OK;
}

// header + payload checksum of a packed blob in device memory: one D2H copy at load time (136 MB, a few ms) -- a
// truncated or stale broadcast must not become silent garbage weights
static int verify_device_blob(idc_context* h, const void* dev_blob, size_t blob_bytes) {
    BlobHeader hd;
    if (blob_bytes < sizeof(hd)) return fail(&h->err, IDC_ERR_INVALID_ARG, "blob too small");
    HIPCHK(h, hipMemcpy(&hd, dev_blob, sizeof(hd), hipMemcpyDeviceToHost));
    int rc = validate_header(h, hd, blob_bytes);
    if (rc) return rc;
    std::vector<uint8_t> tmp(h->plan.total_bytes);
    HIPCHK(h, hipMemcpy(tmp.data(), dev_blob, tmp.size(), hipMemcpyDeviceToHost));
    if (fnv1a(tmp.data() + sizeof(hd), tmp.size() - sizeof(hd)) != hd.checksum)
        return fail(&h->err, IDC_ERR_INVALID_ARG, "device blob checksum mismatch");
    return IDC_OK;
}

int idc_set_weights_device(idc_handle h, const void* dev_blob, size_t blob_bytes, int copy) {
    if (!h || !dev_blob) return fail(h ? &h->err : nullptr, IDC_ERR_INVALID_ARG, "null handle/blob");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = verify_device_blob(h, dev_blob, blob_bytes);
    if (rc) return rc;
    if (copy) {
        if (!h->own_blob || !h->d_blob) {
            h->d_blob = nullptr;
            HIPCHK(h, hipMalloc((void**)&h->d_blob, h->plan.total_bytes));
            h->own_blob = true;
        }
        HIPCHK(h, hipMemcpy(h->d_blob, dev_blob, h->plan.total_bytes, hipMemcpyDeviceToDevice));
    } else {
        if (h->own_blob && h->d_blob) (void)hipFree(h->d_blob);
        h->d_blob = (uint8_t*)dev_blob;
        h->own_blob = false;
    }
    h->weights_set = true;
    return IDC_OK;
}

int idc_load_weights(idc_handle h, const idc_tensor_desc* tensors, int n_tensors) {
    if (!h) return fail(nullptr, IDC_ERR_INVALID_ARG, "null handle");
    std::vector<uint8_t> blob(h->plan.total_bytes);
    int rc = pack_weights_impl(h->precision, h->flags, tensors, n_tensors, blob.data(), blob.size(), &h->err);
    if (rc) return rc;
    return idc_set_weights_host(h, blob.data(), blob.size());
}

const void* idc_weights_device_ptr(idc_handle h) { return (h && h->weights_set) ? h->d_blob : nullptr; }

int idc_forward(idc_handle h, int n, const float* L_mc, const float* ab, const float* mask, float maskcent,
                float* out_ab) {
    return forward_host(h, n, L_mc, ab, mask, maskcent, out_ab, nullptr);
}

int idc_forward_dist(idc_handle h, int n, const float* L_mc, const float* ab, const float* mask, float maskcent,
                     float* out_ab, float* dist_q) {
    return forward_host(h, n, L_mc, ab, mask, maskcent, out_ab, dist_q, /*keep_dist=*/true);   // NULL dist_q: resident only
}

int idc_forward_device(idc_handle h, int n, const float* d_L_mc, const float* d_ab, const float* d_mask, float maskcent,
                       float* d_out_ab, int sync) {
    int rc = check_forward_args(h, n);
    if (rc) return rc;
    if (!d_L_mc || !d_ab || !d_mask || !d_out_ab) return fail(&h->err, IDC_ERR_INVALID_ARG, "null tensor pointer");
    HIPCHK(h, hipSetDevice(h->device));
    // the result lands in the CALLER's buffer: whatever an earlier forward left in d_out / d_labq is no longer "the last
    // forward's map" (idc_upsample_lab2rgb must not serve it), unless the caller handed the handle's own planes back
    h->out_resident = d_out_ab == h->d_out;
    h->labq_resident = false;
    rc = run_graph(h, n, d_L_mc, d_ab, d_mask, maskcent, d_out_ab, (h->flags & IDC_FLAG_DIST_HEAD) ? h->d_dist : nullptr);
    if (rc) return rc;
    if (sync) {
        HIPCHK(h, wait_stream(h, n));
        return check_chain_abort(h);           // (every blocking wait that follows run_graph reports a timed-out chain launch itself)
    }
    return IDC_OK;
}

int idc_set_global_hints(idc_handle h, int n, const float* glob_ab_313_mask, const float* s_avg_mask) {
    if (!h) return fail(nullptr, IDC_ERR_INVALID_ARG, "null handle");
    if (!(h->flags & IDC_FLAG_GLOBAL_HINTS)) return fail(&h->err, IDC_ERR_UNSUPPORTED, "handle was created without IDC_FLAG_GLOBAL_HINTS");
    if (n <= 0 || n > h->max_batch) return fail(&h->err, IDC_ERR_BATCH, "batch %d outside 1..%d", n, h->max_batch);
    if (!glob_ab_313_mask) return fail(&h->err, IDC_ERR_INVALID_ARG, "null glob_ab_313_mask");
    HIPCHK(h, hipSetDevice(h->device));
    std::vector<float> host((size_t)h->max_batch * kGlobIn, 0.f);
    for (int i = 0; i < n; ++i) {
        memcpy(&host[(size_t)i * kGlobIn], glob_ab_313_mask + (size_t)i * 314, 314 * 4);
        if (s_avg_mask) memcpy(&host[(size_t)i * kGlobIn + 314], s_avg_mask + (size_t)i * 2, 2 * 4);
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(h->d_glob_in, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    return IDC_OK;
}

int idc_clear_global_hints(idc_handle h) {
    if (!h) return fail(nullptr, IDC_ERR_INVALID_ARG, "null handle");
    if (!(h->flags & IDC_FLAG_GLOBAL_HINTS)) return fail(&h->err, IDC_ERR_UNSUPPORTED, "handle was created without IDC_FLAG_GLOBAL_HINTS");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemset(h->d_glob_in, 0, (size_t)h->max_batch * kGlobIn * 4));
    return IDC_OK;
}

int idc_forward_dist313(idc_handle h, int n, const float* L_mc, const float* ab, const float* mask, float maskcent,
                        float* out_ab, float* pred_ab, float* dist_S) {
    int rc = check_forward_args(h, n);
    if (rc) return rc;
    if (!(h->flags & IDC_FLAG_DIST313)) return fail(&h->err, IDC_ERR_UNSUPPORTED, "handle was created without IDC_FLAG_DIST313");
    if (!pred_ab) return fail(&h->err, IDC_ERR_INVALID_ARG, "null pred_ab");
    const size_t hw = (size_t)h->H * h->W;
    h->want_dist313 = dist_S != nullptr;
    rc = forward_host(h, n, L_mc, ab, mask, maskcent, out_ab ? out_ab : h->h_out, nullptr);
    h->want_dist313 = false;
    if (rc) return rc;
    HIPCHK(h, hipMemcpy(pred_ab, h->d_pred_ab, (size_t)n * hw * 2 * 4, hipMemcpyDeviceToHost));
    if (dist_S) HIPCHK(h, hipMemcpy(dist_S, h->d_dist313, (size_t)n * hw * 313 * 4, hipMemcpyDeviceToHost));
    return IDC_OK;
}

int idc_set_dist_temperature(idc_handle h, float S) {
    if (!h) return fail(nullptr, IDC_ERR_INVALID_ARG, "null handle");
    if (!(S > 0.f)) return fail(&h->err, IDC_ERR_INVALID_ARG, "temperature must be positive");
    h->dist_S = S;
    return IDC_OK;
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 16); }
};

static int ensure_post_buffers(idc_context* h) {
    if (h->d_rgb) return IDC_OK;
    const size_t hw = (size_t)h->H * h->W, nb = (size_t)h->max_batch;
    HIPCHK(h, hipMalloc((void**)&h->d_rgb, nb * hw * 3));
    HIPCHK(h, hipMalloc((void**)&h->d_labq, nb * hw * 3 * 8));
    HIPCHK(h, hipMalloc((void**)&h->d_post_in, nb * hw * 3 * 4));
    HIPCHK(h, hipHostMalloc((void**)&h->h_rgb, nb * hw * 3, hipHostMallocDefault));
    HIPCHK(h, hipHostMalloc((void**)&h->h_labq, nb * hw * 3 * 8, hipHostMallocDefault));
    return IDC_OK;
}

// post step on device-resident planes: d_Lp [n,1,H,W] (+ l_add), d_abp [n,2,H,W] -> host rgb / lab_q
static int run_lab_post(idc_context* h, int n, const float* d_Lp, float l_add, const float* d_abp, uint8_t* rgb, double* lab_q) {
    const size_t hw = (size_t)h->H * h->W;
    int rc = ensure_post_buffers(h);
    if (rc) return rc;
    HIPCHK(h, launch_lab_post(d_Lp, l_add, d_abp, h->d_rgb, lab_q ? h->d_labq : nullptr, n, h->H, h->W, h->stream));
    const bool rgb_direct = is_pinned(rgb), lab_direct = lab_q && is_pinned(lab_q);      // pinned caller buffers: no staging copy
    HIPCHK(h, copy_h2d_or_d2h(h, h->d_rgb, rgb_direct ? (void*)rgb : (void*)h->h_rgb, (size_t)n * hw * 3, false));
    if (lab_q) HIPCHK(h, copy_h2d_or_d2h(h, h->d_labq, lab_direct ? (void*)lab_q : (void*)h->h_labq, (size_t)n * hw * 3 * 8, false));
    HIPCHK(h, wait_stream(h, n));
    rc = check_chain_abort(h);
    if (rc) return rc;
    if (!rgb_direct) memcpy(rgb, h->h_rgb, (size_t)n * hw * 3);
    if (lab_q && !lab_direct) memcpy(lab_q, h->h_labq, (size_t)n * hw * 3 * 8);
    return IDC_OK;
}

int idc_lab2rgb(idc_handle h, int n, const float* L, const float* ab, uint8_t* rgb, double* lab_q) {
    if (!h) return fail(nullptr, IDC_ERR_INVALID_ARG, "null handle");
    if (n <= 0 || n > h->max_batch) return fail(&h->err, IDC_ERR_BATCH, "batch %d outside 1..%d", n, h->max_batch);
    if (!L || !ab || !rgb) return fail(&h->err, IDC_ERR_INVALID_ARG, "null tensor pointer");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = ensure_post_buffers(h);
    if (rc) return rc;
    const size_t hw = (size_t)h->H * h->W;
    memcpy(h->h_in, L, (size_t)n * hw * 4);
    memcpy(h->h_in + (size_t)n * hw, ab, (size_t)n * hw * 2 * 4);
    HIPCHK(h, hipMemcpyAsync(h->d_post_in, h->h_in, (size_t)n * hw * 3 * 4, hipMemcpyHostToDevice, h->stream));
    rc = run_lab_post(h, n, h->d_post_in, 0.f, h->d_post_in + (size_t)n * hw, rgb, lab_q);
    h->labq_resident = rc == IDC_OK && lab_q != nullptr;        // d_labq = rgb2lab of exactly what was passed in
    if (h->labq_resident && h->last_n < n) h->last_n = n;
    return rc;
}

int idc_forward_rgb(idc_handle h, int n, const float* L_mc, const float* ab, const float* mask, float maskcent,
                    float l_cent, float* out_ab, uint8_t* rgb, double* lab_q) {
    int rc = check_forward_args(h, n);
    if (rc) return rc;
    if (!rgb) return fail(&h->err, IDC_ERR_INVALID_ARG, "null rgb");
    rc = ensure_post_buffers(h);
    if (rc) return rc;
    // forward (leaves L_mc in d_L and the ab map in d_out), then the colour step on the same stream: ONE synchronisation for
    // both (the ab map travels to the host under the colour kernel instead of behind a sync of its own)
    rc = forward_host(h, n, L_mc, ab, mask, maskcent, out_ab ? out_ab : h->h_out, nullptr, false, /*finish=*/false);
    if (rc) return rc;
    rc = run_lab_post(h, n, h->d_L, l_cent, h->d_out, rgb, lab_q);          // synchronises the stream
    if (rc == IDC_OK && out_ab && h->out_copy_pending) memcpy(out_ab, h->h_out, (size_t)n * h->H * h->W * 2 * 4);
    h->out_copy_pending = false;
    h->labq_resident = rc == IDC_OK && lab_q != nullptr;
    return rc;
}

// The click as the reference API needs it on the critical path: the colourised image.  The ab map and the refreshed Lab (2.0 MB of
// the 2.2 MB idc_forward_rgb sends back per 256x256 click) stay on the device until idc_fetch_outputs asks for them.
int idc_forward_rgb_lazy(idc_handle h, int n, const float* L_mc, const float* ab, const float* mask, float maskcent, float l_cent, uint8_t* rgb) {
    int rc = check_forward_args(h, n);
    if (rc) return rc;
    if (!rgb) return fail(&h->err, IDC_ERR_INVALID_ARG, "null rgb");
    rc = ensure_post_buffers(h);
    if (rc) return rc;
    rc = forward_host(h, n, L_mc, ab, mask, maskcent, nullptr, nullptr, false, /*finish=*/false, /*copy_out=*/false);
    if (rc) return rc;
    const size_t hw = (size_t)h->H * h->W;
    HIPCHK(h, launch_lab_post(h->d_L, l_cent, h->d_out, h->d_rgb, h->d_labq, n, h->H, h->W, h->stream));
    const bool rgb_direct = is_pinned(rgb);
    HIPCHK(h, copy_h2d_or_d2h(h, h->d_rgb, rgb_direct ? (void*)rgb : (void*)h->h_rgb, (size_t)n * hw * 3, false));
    HIPCHK(h, wait_stream(h, n));
    rc = check_chain_abort(h);
    if (rc) return rc;
    if (!rgb_direct) memcpy(rgb, h->h_rgb, (size_t)n * hw * 3);
    h->out_copy_pending = false;
    h->labq_resident = true;
    return IDC_OK;
}

int idc_fetch_outputs(idc_handle h, int n, float* out_ab, double* lab_q) {
    if (!h) return fail(nullptr, IDC_ERR_INVALID_ARG, "null handle");
    if (n <= 0 || n > h->max_batch) return fail(&h->err, IDC_ERR_BATCH, "batch %d outside 1..%d", n, h->max_batch);
    if (n > h->last_n) return fail(&h->err, IDC_ERR_UNSUPPORTED, "only %d image(s) of the last forward are resident", h->last_n);
    if (out_ab && !h->out_resident) return fail(&h->err, IDC_ERR_UNSUPPORTED, "no forward result is resident");
    if (lab_q && !h->labq_resident) return fail(&h->err, IDC_ERR_UNSUPPORTED, "no refreshed Lab is resident (run idc_forward_rgb_lazy / idc_forward_rgb with lab_q first)");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t hw = (size_t)h->H * h->W;
    const bool ab_direct = out_ab && is_pinned(out_ab), lab_direct = lab_q && is_pinned(lab_q);
    if (out_ab) HIPCHK(h, copy_h2d_or_d2h(h, h->d_out, ab_direct ? out_ab : h->h_out, (size_t)n * hw * 2 * 4, false));
    if (lab_q) {
        int rc = ensure_post_buffers(h);
        if (rc) return rc;
        HIPCHK(h, copy_h2d_or_d2h(h, h->d_labq, lab_direct ? (void*)lab_q : (void*)h->h_labq, (size_t)n * hw * 3 * 8, false));
    }
    HIPCHK(h, wait_stream(h, n));
    if (out_ab && !ab_direct) memcpy(out_ab, h->h_out, (size_t)n * hw * 2 * 4);
    if (lab_q && !lab_direct) memcpy(lab_q, h->h_labq, (size_t)n * hw * 3 * 8);
    return IDC_OK;
}

// ---------------------------------------------------------------------------------------------- click session
static int check_img(idc_context* h, int img) {
    if (!h) return fail(nullptr, IDC_ERR_INVALID_ARG, "null handle");
    if (img < 0 || img >= h->max_batch) return fail(&h->err, IDC_ERR_BATCH, "image %d outside 0..%d", img, h->max_batch - 1);
    return IDC_OK;
}

int idc_set_image_l(idc_handle h, int img, const float* L_mc) {
    int rc = check_img(h, img);
    if (rc) return rc;
    if (!L_mc) return fail(&h->err, IDC_ERR_INVALID_ARG, "null L_mc");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t hw = (size_t)h->H * h->W;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(h->d_L + (size_t)img * hw, L_mc, hw * 4, hipMemcpyHostToDevice));
    h->l_set[img] = 1;
    return IDC_OK;
}

int idc_set_hints(idc_handle h, int img, int n_hints, const idc_hint* hints, int mode, float mask_value) {
    int rc = check_img(h, img);
    if (rc) return rc;
    if (n_hints < 0 || (n_hints > 0 && !hints)) return fail(&h->err, IDC_ERR_INVALID_ARG, "bad hint list");
    if (mode != IDC_HINT_AB && mode != IDC_HINT_RGB) return fail(&h->err, IDC_ERR_INVALID_ARG, "hint mode %d", mode);
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));            // the pinned list of the previous call may still be in flight
    if (n_hints > h->hints_cap) {
        const int cap = n_hints < 256 ? 256 : 2 * n_hints;
        if (h->d_hints) (void)hipFree(h->d_hints);
        if (h->h_hints) (void)hipHostFree(h->h_hints);
        h->d_hints = nullptr; h->h_hints = nullptr; h->hints_cap = 0;
        HIPCHK(h, hipMalloc((void**)&h->d_hints, (size_t)cap * sizeof(HintRect)));
        HIPCHK(h, hipHostMalloc((void**)&h->h_hints, (size_t)cap * sizeof(HintRect), hipHostMallocDefault));
        h->hints_cap = cap;
    }
    int kept = 0;
    for (int i = 0; i < n_hints; ++i) {                    // cv2.rectangle: corners in either order, inclusive, clipped
        HintRect r;
        r.y0 = hints[i].y0 < hints[i].y1 ? hints[i].y0 : hints[i].y1; r.y1 = hints[i].y0 < hints[i].y1 ? hints[i].y1 : hints[i].y0;
        r.x0 = hints[i].x0 < hints[i].x1 ? hints[i].x0 : hints[i].x1; r.x1 = hints[i].x0 < hints[i].x1 ? hints[i].x1 : hints[i].x0;
        if (r.y0 < 0) r.y0 = 0;
        if (r.x0 < 0) r.x0 = 0;
        if (r.y1 > h->H - 1) r.y1 = h->H - 1;
        if (r.x1 > h->W - 1) r.x1 = h->W - 1;
        if (r.y0 > r.y1 || r.x0 > r.x1) continue;          // entirely outside
        r.c0 = hints[i].c0; r.c1 = hints[i].c1; r.c2 = hints[i].c2;
        if (mode == IDC_HINT_RGB)
            if (!(r.c0 >= 0.f && r.c0 <= 255.f && r.c1 >= 0.f && r.c1 <= 255.f && r.c2 >= 0.f && r.c2 <= 255.f))
                return fail(&h->err, IDC_ERR_INVALID_ARG, "hint %d: RGB outside 0..255", i);
        h->h_hints[kept++] = r;
    }
    const size_t hw = (size_t)h->H * h->W;
    if (kept) HIPCHK(h, hipMemcpyAsync(h->d_hints, h->h_hints, (size_t)kept * sizeof(HintRect), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, launch_raster_hints(h->d_hints, kept, mode, mask_value, h->d_ab + (size_t)img * hw * 2, h->d_mask + (size_t)img * hw,
                                  h->H, h->W, h->stream));
    return IDC_OK;
}

int idc_get_hint_planes(idc_handle h, int img, float* ab, float* mask) {
    int rc = check_img(h, img);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    const size_t hw = (size_t)h->H * h->W;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (ab) HIPCHK(h, hipMemcpy(ab, h->d_ab + (size_t)img * hw * 2, hw * 2 * 4, hipMemcpyDeviceToHost));
    if (mask) HIPCHK(h, hipMemcpy(mask, h->d_mask + (size_t)img * hw, hw * 4, hipMemcpyDeviceToHost));
    return IDC_OK;
}

int idc_forward_resident(idc_handle h, int n, float maskcent, float l_cent, float* out_ab, uint8_t* rgb, double* lab_q) {
    int rc = check_forward_args(h, n);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    for (int i = 0; i < n; ++i)
        if (!h->l_set[i]) return fail(&h->err, IDC_ERR_INVALID_ARG, "I need to have an image! (slot %d has no L plane: idc_set_image_l)", i);
    rc = drain_pipeline(h);
    if (rc) return rc;
    const size_t hw = (size_t)h->H * h->W;
    rc = run_graph(h, n, h->d_L, h->d_ab, h->d_mask, maskcent, h->d_out, (h->flags & IDC_FLAG_DIST_HEAD) ? h->d_dist : nullptr);
    if (rc) return rc;
    h->out_resident = true; h->labq_resident = rgb != nullptr && lab_q != nullptr;
    if (out_ab) HIPCHK(h, hipMemcpyAsync(h->h_out, h->d_out, (size_t)n * hw * 2 * 4, hipMemcpyDeviceToHost, h->stream));
    if (rgb) {
        rc = run_lab_post(h, n, h->d_L, l_cent, h->d_out, rgb, lab_q);      // synchronises the stream
        if (rc) return rc;
    } else {
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    rc = check_chain_abort(h);
    if (rc) return rc;
    if (out_ab) memcpy(out_ab, h->h_out, (size_t)n * hw * 2 * 4);
    return IDC_OK;
}

// ---------------------------------------------------------------------------------------------- colour suggestions
int idc_keep_dist(idc_handle h, int on) {
    if (!h) return fail(nullptr, IDC_ERR_INVALID_ARG, "null handle");
    if (!(h->flags & IDC_FLAG_DIST313)) return fail(&h->err, IDC_ERR_UNSUPPORTED, "handle was created without IDC_FLAG_DIST313");
    h->keep_dist313 = on != 0;
    return IDC_OK;
}

// where the resident distribution of image `img` lives: bins, element stride between bins, pointer to bin 0 at (y, x)
static int dist_locate(idc_context* h, int img, int y, int x, int* B, long long* stride, const float** p) {
    if (!h) return fail(nullptr, IDC_ERR_INVALID_ARG, "null handle");
    if (h->dist_n <= 0) return fail(&h->err, IDC_ERR_UNSUPPORTED, "Need to set prediction first (no resident distribution)");
    if (img < 0 || img >= h->dist_n) return fail(&h->err, IDC_ERR_BATCH, "image %d outside 0..%d", img, h->dist_n - 1);
    if (y < 0 || y >= h->H || x < 0 || x >= h->W) return fail(&h->err, IDC_ERR_INVALID_ARG, "pixel (%d,%d) outside the image", y, x);
    if (h->flags & IDC_FLAG_DIST313) {
        *B = 313; *stride = (long long)h->H * h->W;
        *p = h->d_dist313 + (size_t)img * 313 * (*stride) + (size_t)y * h->W + x;
    } else {                                               // 529 bins at H/4 x W/4; out_cl is its nearest x4 upsample (model.py:131)
        const int h4 = h->H / 4, w4 = h->W / 4;
        *B = 529; *stride = (long long)h4 * w4;
        *p = h->d_dist + (size_t)img * 529 * (*stride) + (size_t)(y / 4) * w4 + (x / 4);
    }
    return IDC_OK;
}

int idc_dist_bins(idc_handle h) { return !h ? 0 : (h->flags & IDC_FLAG_DIST313) ? 313 : (h->flags & IDC_FLAG_DIST_HEAD) ? 529 : 0; }

int idc_dist_at(idc_handle h, int img, int y, int x, float* pdf) {
    int B; long long stride; const float* p;
    int rc = dist_locate(h, img, y, x, &B, &stride, &p);
    if (rc) return rc;
    if (!pdf) return fail(&h->err, IDC_ERR_INVALID_ARG, "null pdf");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy2D(pdf, 4, p, (size_t)stride * 4, 4, (size_t)B, hipMemcpyDeviceToHost));
    return IDC_OK;
}

int idc_get_dist(idc_handle h, int n, float* dist) {
    if (!h) return fail(nullptr, IDC_ERR_INVALID_ARG, "null handle");
    if (h->dist_n <= 0) return fail(&h->err, IDC_ERR_UNSUPPORTED, "Need to set prediction first (no resident distribution)");
    if (n <= 0 || n > h->dist_n) return fail(&h->err, IDC_ERR_BATCH, "batch %d outside 1..%d", n, h->dist_n);
    if (!dist) return fail(&h->err, IDC_ERR_INVALID_ARG, "null dist");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t hw = (size_t)h->H * h->W;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->flags & IDC_FLAG_DIST313) HIPCHK(h, hipMemcpy(dist, h->d_dist313, (size_t)n * 313 * hw * 4, hipMemcpyDeviceToHost));
    else HIPCHK(h, hipMemcpy(dist, h->d_dist, (size_t)n * 529 * (hw / 16) * 4, hipMemcpyDeviceToHost));
    return IDC_OK;
}

int idc_suggest_colors(idc_handle h, int img, int y, int x, int K, int N, unsigned seed, const float* centres,
                       double* out_centres, double* out_conf, unsigned* out_counts) {
    int B; long long stride; const float* p;
    int rc = dist_locate(h, img, y, x, &B, &stride, &p);
    if (rc) return rc;
    if (!centres || !out_centres || !out_conf) return fail(&h->err, IDC_ERR_INVALID_ARG, "null pointer");
    if (K < 1 || K > kSuggestMaxK) return fail(&h->err, IDC_ERR_INVALID_ARG, "K %d outside 1..%d", K, kSuggestMaxK);
    if (N < 1) return fail(&h->err, IDC_ERR_INVALID_ARG, "N must be positive");
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->d_centres) {
        HIPCHK(h, hipMalloc((void**)&h->d_centres, (size_t)kSuggestMaxBins * 2 * 4));
        HIPCHK(h, hipMalloc((void**)&h->d_sugg, (size_t)kSuggestMaxK * 3 * 8));
        HIPCHK(h, hipMalloc((void**)&h->d_sugg_counts, (size_t)kSuggestMaxBins * 4));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(h->d_centres, centres, (size_t)B * 2 * 4, hipMemcpyHostToDevice));
    HIPCHK(h, launch_suggest(p, stride, B, h->d_centres, K, N, seed, h->d_sugg, h->d_sugg + 2 * kSuggestMaxK, h->d_sugg_counts, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(out_centres, h->d_sugg, (size_t)K * 2 * 8, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(out_conf, h->d_sugg + 2 * kSuggestMaxK, (size_t)K * 8, hipMemcpyDeviceToHost));
    if (out_counts) HIPCHK(h, hipMemcpy(out_counts, h->d_sugg_counts, (size_t)B * 4, hipMemcpyDeviceToHost));
    return IDC_OK;
}

int idc_global_histogram(idc_handle h, int n, const uint8_t* rgb, const float* centres, float* hist, float* s_avg) {
    if (!h) return fail(nullptr, IDC_ERR_INVALID_ARG, "null handle");
    if (n <= 0 || n > h->max_batch) return fail(&h->err, IDC_ERR_BATCH, "batch %d outside 1..%d", n, h->max_batch);
    if (!rgb || !centres || !hist) return fail(&h->err, IDC_ERR_INVALID_ARG, "null pointer");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = ensure_post_buffers(h);                      // d_rgb doubles as the upload buffer of the reference image
    if (rc) return rc;
    const size_t hw = (size_t)h->H * h->W;
    DevBuf d_c, d_counts, d_sat;
    HIPCHK(h, d_c.alloc(626 * 4)); HIPCHK(h, d_counts.alloc((size_t)n * 313 * 4)); HIPCHK(h, d_sat.alloc((size_t)n * 8));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(h->d_rgb, rgb, (size_t)n * hw * 3, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(d_c.p, centres, 626 * 4, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemset(d_counts.p, 0, (size_t)n * 313 * 4));
    HIPCHK(h, hipMemset(d_sat.p, 0, (size_t)n * 8));
    HIPCHK(h, launch_global_stats(h->d_rgb, (const float*)d_c.p, (unsigned*)d_counts.p, (double*)d_sat.p, n, h->H, h->W, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    std::vector<unsigned> cnt((size_t)n * 313);
    std::vector<double> sat(n);
    HIPCHK(h, hipMemcpy(cnt.data(), d_counts.p, cnt.size() * 4, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(sat.data(), d_sat.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    const double nblk = (double)(h->H / 4) * (h->W / 4);
    for (int i = 0; i < n; ++i) {
        for (int k = 0; k < 313; ++k) hist[(size_t)i * 313 + k] = (float)(cnt[(size_t)i * 313 + k] / nblk);
        if (s_avg) s_avg[i] = (float)(sat[i] / (double)hw);
    }
    return IDC_OK;
}

// ---------------------------------------------------------------------------------------------- stream ordering
// The handle's work runs on its own non-blocking stream.  A caller that produces device inputs or consumes device
// outputs on ANOTHER stream orders the two with these (or synchronises fully): wait = "the handle's stream waits for
// everything enqueued so far on caller_stream"; signal = "caller_stream waits for everything the handle enqueued so far".
int idc_stream_wait(idc_handle h, void* caller_stream) {
    if (!h) return fail(nullptr, IDC_ERR_INVALID_ARG, "null handle");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipEventRecord(h->ev_sync, (hipStream_t)caller_stream));
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_sync, 0));
    return IDC_OK;
}

int idc_stream_signal(idc_handle h, void* caller_stream) {
    if (!h) return fail(nullptr, IDC_ERR_INVALID_ARG, "null handle");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipEventRecord(h->ev_sync, h->stream));
    HIPCHK(h, hipStreamWaitEvent((hipStream_t)caller_stream, h->ev_sync, 0));
    return IDC_OK;
}

// ---------------------------------------------------------------------------------------------- transfer pipeline
// End-to-end batches: two slots, three streams.  Slot k's H2D copies run on the copy-in
// stream while the other slot computes; its D2H runs on the copy-out stream while the next batch computes.  Callers that
// pass pinned host memory (idc_alloc_host, or their own hipHostMalloc / hipHostRegister) are copied from / to directly;
// pageable pointers go through the slot's pinned staging with a host memcpy on the calling thread.
void* idc_alloc_host(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

int idc_free_host(void* p) {
    if (!p) return IDC_OK;
    return hipHostFree(p) == hipSuccess ? IDC_OK : fail(nullptr, IDC_ERR_HIP, "hipHostFree failed");
}

static bool is_pinned(const void* p) {
    hipPointerAttribute_t at;
    const hipError_t e = hipPointerGetAttributes(&at, p);
    (void)hipGetLastError();                      // pageable memory is "invalid value" to this query: not an error of ours, never sticky
    return e == hipSuccess && at.type == hipMemoryTypeHost;
}

static int ensure_pipeline(idc_context* h) {
    if (h->pipe_ready) return IDC_OK;
    const size_t hw = (size_t)h->H * h->W, nb = (size_t)h->max_batch;
    HIPCHK(h, hipStreamCreateWithFlags(&h->s_in, hipStreamNonBlocking));
    HIPCHK(h, hipStreamCreateWithFlags(&h->s_out, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) {
        auto& sl = h->pipe[k];
        // both slots own their planes (50 MB each at N = 32): nothing here aliases the handle's resident L / hint / output
        // planes, so the copy-in stream never has to be ordered against the compute stream (ordering slot 0's copies behind
        // "everything enqueued so far" serialises them behind the OTHER slot's kernels: measured 0.89 instead of 0.97 of
        // the device-resident rate, profiles/r03a_bench.json)
        HIPCHK(h, hipMalloc((void**)&sl.d_L, nb * hw * 4));
        HIPCHK(h, hipMalloc((void**)&sl.d_ab, nb * hw * 2 * 4));
        HIPCHK(h, hipMalloc((void**)&sl.d_mask, nb * hw * 4));
        HIPCHK(h, hipMalloc((void**)&sl.d_out, nb * hw * 2 * 4));
        // timing-capable events: idc_pipeline_times reports where each stage of a batch sat on the device's clock
        hipEvent_t* evs[] = {&sl.ev_in, &sl.ev_comp, &sl.ev_out, &sl.ev_in0, &sl.ev_comp0, &sl.ev_out0};
        for (hipEvent_t* e : evs) HIPCHK(h, hipEventCreate(e));
    }
    HIPCHK(h, hipEventCreate(&h->ev_pipe_base));
    HIPCHK(h, hipEventRecord(h->ev_pipe_base, h->stream));
    h->pipe_ready = true;
    return IDC_OK;
}

static int wait_slot(idc_context* h, int slot) {
    auto& sl = h->pipe[slot];
    if (!sl.pending) return IDC_OK;
    HIPCHK(h, hipEventSynchronize(sl.ev_out));
    sl.pending = false;
    const int arc = check_chain_abort(h);      // the forward this slot carried ran a chain launch that gave up: its result is invalid
    if (arc) return arc;
    if (sl.staged_out) memcpy(sl.user_out, sl.h_out, (size_t)sl.n * h->H * h->W * 2 * 4);
    return IDC_OK;
}

static int drain_pipeline(idc_context* c) {
    if (!c->pipe_ready) return IDC_OK;
    for (int k = 0; k < 2; ++k) { int rc = wait_slot(c, k); if (rc) return rc; }
    return IDC_OK;
}

int idc_forward_async(idc_handle h, int slot, int n, const float* L_mc, const float* ab, const float* mask, float maskcent,
                      float* out_ab) {
    int rc = check_forward_args(h, n);
    if (rc) return rc;
    if (slot < 0 || slot > 1) return fail(&h->err, IDC_ERR_INVALID_ARG, "slot %d not in 0..1", slot);
    if (!L_mc || !ab || !mask || !out_ab) return fail(&h->err, IDC_ERR_INVALID_ARG, "null tensor pointer");
    HIPCHK(h, hipSetDevice(h->device));
    rc = ensure_pipeline(h);
    if (rc) return rc;
    auto& sl = h->pipe[slot];
    if (sl.pending) return fail(&h->err, IDC_ERR_INVALID_ARG, "slot %d is still in flight: idc_wait it first", slot);
    const size_t hw = (size_t)h->H * h->W, nb = (size_t)h->max_batch;
    const float *sL = L_mc, *sab = ab, *sm = mask;
    if (!(is_pinned(L_mc) && is_pinned(ab) && is_pinned(mask))) {
        if (!sl.h_in) HIPCHK(h, hipHostMalloc((void**)&sl.h_in, nb * hw * 4 * 4, hipHostMallocDefault));
        float* hL = sl.h_in; float* hab = hL + (size_t)n * hw; float* hm = hab + (size_t)n * hw * 2;
        memcpy(hL, L_mc, (size_t)n * hw * 4); memcpy(hab, ab, (size_t)n * hw * 2 * 4); memcpy(hm, mask, (size_t)n * hw * 4);
        sL = hL; sab = hab; sm = hm;
    }
    sl.staged_out = !is_pinned(out_ab);
    if (sl.staged_out && !sl.h_out) HIPCHK(h, hipHostMalloc((void**)&sl.h_out, nb * hw * 2 * 4, hipHostMallocDefault));
    // copy-in stream: the slot's previous inputs were consumed (its previous forward finished: idc_wait was called)
    HIPCHK(h, hipEventRecord(sl.ev_in0, h->s_in));
    HIPCHK(h, hipMemcpyAsync(sl.d_L, sL, (size_t)n * hw * 4, hipMemcpyHostToDevice, h->s_in));
    HIPCHK(h, hipMemcpyAsync(sl.d_ab, sab, (size_t)n * hw * 2 * 4, hipMemcpyHostToDevice, h->s_in));
    HIPCHK(h, hipMemcpyAsync(sl.d_mask, sm, (size_t)n * hw * 4, hipMemcpyHostToDevice, h->s_in));
    HIPCHK(h, hipEventRecord(sl.ev_in, h->s_in));
    HIPCHK(h, hipStreamWaitEvent(h->stream, sl.ev_in, 0));
    HIPCHK(h, hipEventRecord(sl.ev_comp0, h->stream));
    rc = run_graph(h, n, sl.d_L, sl.d_ab, sl.d_mask, maskcent, sl.d_out, nullptr);
    if (rc) return rc;
    HIPCHK(h, hipEventRecord(sl.ev_comp, h->stream));
    HIPCHK(h, hipStreamWaitEvent(h->s_out, sl.ev_comp, 0));
    HIPCHK(h, hipEventRecord(sl.ev_out0, h->s_out));
    HIPCHK(h, hipMemcpyAsync(sl.staged_out ? sl.h_out : out_ab, sl.d_out, (size_t)n * hw * 2 * 4, hipMemcpyDeviceToHost, h->s_out));
    HIPCHK(h, hipEventRecord(sl.ev_out, h->s_out));
    sl.pending = true; sl.timed = true; sl.user_out = out_ab; sl.n = n;
    return IDC_OK;
}

int idc_wait(idc_handle h, int slot) {
    if (!h) return fail(nullptr, IDC_ERR_INVALID_ARG, "null handle");
    if (slot < 0 || slot > 1) return fail(&h->err, IDC_ERR_INVALID_ARG, "slot %d not in 0..1", slot);
    if (!h->pipe_ready) return IDC_OK;
    HIPCHK(h, hipSetDevice(h->device));
    return wait_slot(h, slot);
}

int idc_pipeline_times(idc_handle h, int slot, float* ms6) {
    if (!h || !ms6) return fail(h ? &h->err : nullptr, IDC_ERR_INVALID_ARG, "null argument");
    if (slot < 0 || slot > 1) return fail(&h->err, IDC_ERR_INVALID_ARG, "slot %d not in 0..1", slot);
    if (!h->pipe_ready || !h->pipe[slot].timed) return fail(&h->err, IDC_ERR_UNSUPPORTED, "slot %d has not run a batch yet", slot);
    auto& sl = h->pipe[slot];
    if (sl.pending) return fail(&h->err, IDC_ERR_INVALID_ARG, "slot %d is still in flight: idc_wait it first", slot);
    HIPCHK(h, hipSetDevice(h->device));
    hipEvent_t evs[6] = {sl.ev_in0, sl.ev_in, sl.ev_comp0, sl.ev_comp, sl.ev_out0, sl.ev_out};
    for (int i = 0; i < 6; ++i) HIPCHK(h, hipEventElapsedTime(&ms6[i], h->ev_pipe_base, evs[i]));
    return IDC_OK;
}

// ---------------------------------------------------------------------------------------------- RCCL weight broadcast
// SURVEY.md 8b's export list / 8e: one broadcast of the packed blob from `root` over xGMI, called by every rank's
// process with its own handle.  librccl is opened at the first call (the copy torch already loaded, if any, else
// /opt/rocm's) -- the library has no link-time dependency on it, single-GPU users never touch it.
struct IdcNcclId { char internal[128]; };                  // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value
struct Rccl {
    void* lib = nullptr;
    std::string path;
    std::string tried;                   // every candidate that failed, for the error string
    bool beside_runtime = false;         // opened from the directory of the libamdhip64 this library is bound to
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, IdcNcclId, int) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
static Rccl* rccl() {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        // RCCL launches its kernels on OUR stream handle, so it has to be the copy bound to the SAME libamdhip64 this
        // library resolved (a process that imported torch after this library holds two HIP runtimes: torch's bundled
        // librccl belongs to the other one, and a stream handle of one runtime means nothing to the other).  The HIP
        // runtime we call is found with dladdr; the librccl that ships beside it is the one to open.
        Dl_info di;
        std::string dir;
        if (dladdr((const void*)&hipStreamSynchronize, &di) && di.dli_fname) {
            dir = di.dli_fname;
            const size_t sl = dir.rfind('/');
            dir = sl == std::string::npos ? std::string() : dir.substr(0, sl + 1);
        }
        auto try_open = [&](const std::string& nm, int flags) {
            if (r.lib) return;
            r.lib = dlopen(nm.c_str(), flags);
            if (r.lib) r.path = nm; else r.tried += (r.tried.empty() ? "" : ", ") + nm;
        };
        if (const char* forced = getenv("IDC_RCCL_PATH")) try_open(forced, RTLD_NOW | RTLD_LOCAL);      // explicit override wins
        if (!dir.empty()) {
            try_open(dir + "librccl.so.1", RTLD_NOW | RTLD_LOCAL);
            try_open(dir + "librccl.so", RTLD_NOW | RTLD_LOCAL);
            r.beside_runtime = r.lib != nullptr && r.path.compare(0, dir.size(), dir) == 0;
        }
        // Not beside the runtime (split packages, LD_LIBRARY_PATH installs -- ADVICE r3: this search had been dropped): the loader's
        // own resolution.  A librccl found this way binds to whichever libamdhip64 the loader gives IT; when that is the runtime this
        // library uses (one ROCm install on the library path: the normal case) everything is as above, and the path travels in every
        // error string so a mismatch is diagnosable.
        if (!r.lib) {
            const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
            for (const char* nm : names) try_open(nm, RTLD_NOW | RTLD_LOCAL);
        }
        if (r.lib) {
            r.GetUniqueId = (int (*)(void*))dlsym(r.lib, "ncclGetUniqueId");
            r.CommInitRank = (int (*)(void**, int, IdcNcclId, int))dlsym(r.lib, "ncclCommInitRank");
            r.Broadcast = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(r.lib, "ncclBroadcast");
            r.CommDestroy = (int (*)(void*))dlsym(r.lib, "ncclCommDestroy");
            r.GetErrorString = (const char* (*)(int))dlsym(r.lib, "ncclGetErrorString");
            if (!r.GetUniqueId || !r.CommInitRank || !r.Broadcast || !r.CommDestroy) { r.tried += " (" + r.path + ": nccl symbols missing)"; r.lib = nullptr; }
        }
    }
    return r.lib ? &r : nullptr;
}
static std::string rccl_where() {            // for error strings: which librccl, and whether it is the runtime's sibling
    Rccl* r = rccl();
    if (!r) {
        static Rccl dummy;
        (void)dummy;
        return "librccl.so could not be opened (IDC_RCCL_PATH overrides the search)";
    }
    return r->path + (r->beside_runtime ? "" : " [not beside the libamdhip64 this library is bound to]");
}

int idc_comm_unique_id(void* id128) {
    if (!id128) return fail(nullptr, IDC_ERR_INVALID_ARG, "null id buffer");
    Rccl* r = rccl();
    if (!r) return fail(nullptr, IDC_ERR_UNSUPPORTED, "%s", rccl_where().c_str());
    const int e = r->GetUniqueId(id128);
    if (e != 0) return fail(nullptr, IDC_ERR_HIP, "ncclGetUniqueId failed: %s (%s)", r->GetErrorString ? r->GetErrorString(e) : "?", rccl_where().c_str());
    return IDC_OK;
}

int idc_broadcast_weights(idc_handle h, const void* unique_id, int rank, int world, int root) {
    if (!h || !unique_id) return fail(h ? &h->err : nullptr, IDC_ERR_INVALID_ARG, "null handle / id");
    if (world < 1 || rank < 0 || rank >= world || root < 0 || root >= world)
        return fail(&h->err, IDC_ERR_INVALID_ARG, "rank %d / root %d outside world %d", rank, root, world);
    if (rank == root && !h->weights_set) return fail(&h->err, IDC_ERR_NO_WEIGHTS, "the root rank has no weights to broadcast");
    Rccl* r = rccl();
    if (!r) return fail(&h->err, IDC_ERR_UNSUPPORTED, "%s", rccl_where().c_str());
    HIPCHK(h, hipSetDevice(h->device));
    if (rank != root && (!h->own_blob || !h->d_blob)) {
        h->d_blob = nullptr;
        HIPCHK(h, hipMalloc((void**)&h->d_blob, h->plan.total_bytes));
        h->own_blob = true;
    }
    IdcNcclId id;
    memcpy(&id, unique_id, sizeof(id));
    void* comm = nullptr;
    int e = r->CommInitRank(&comm, world, id, rank);
    if (e != 0) return fail(&h->err, IDC_ERR_HIP, "ncclCommInitRank failed: %s (%s)", r->GetErrorString ? r->GetErrorString(e) : "?", rccl_where().c_str());
    e = r->Broadcast(h->d_blob, h->d_blob, h->plan.total_bytes, /*ncclUint8*/ 1, root, comm, h->stream);
    hipError_t he = hipStreamSynchronize(h->stream);
    (void)r->CommDestroy(comm);
    if (e != 0) return fail(&h->err, IDC_ERR_HIP, "ncclBroadcast failed: %s (%s)", r->GetErrorString ? r->GetErrorString(e) : "?", rccl_where().c_str());
    if (he != hipSuccess) return fail(&h->err, IDC_ERR_HIP, "stream sync after ncclBroadcast: %s", hipGetErrorString(he));
    if (rank != root) {                     // the received bytes are validated like any other device blob
        h->weights_set = false;
        int rc = verify_device_blob(h, h->d_blob, h->plan.total_bytes);
        if (rc) return rc;
        h->weights_set = true;
    }
    return IDC_OK;
}

// ---------------------------------------------------------------------------------------------- display step
int idc_upsample_lab2rgb(idc_handle h, int img, int source, int interp, int out_h, int out_w, const double* L, uint8_t* rgb) {
    int rc = check_img(h, img);
    if (rc) return rc;
    if (!L || !rgb || out_h <= 0 || out_w <= 0) return fail(&h->err, IDC_ERR_INVALID_ARG, "bad output geometry / null pointer");
    if (interp < 0 || interp > 2) return fail(&h->err, IDC_ERR_INVALID_ARG, "interp %d not in 0..2", interp);
    HIPCHK(h, hipSetDevice(h->device));
    const size_t hw = (size_t)h->H * h->W;
    const void *pa = nullptr, *pb = nullptr; int f64 = 0;
    if (source == IDC_SRC_OUTPUT_AB) {
        if (!h->labq_resident || img >= h->last_n) return fail(&h->err, IDC_ERR_UNSUPPORTED, "no refreshed output_ab is resident (run idc_forward_rgb / idc_forward_resident with lab_q first)");
        pa = h->d_labq + ((size_t)img * 3 + 1) * hw; pb = h->d_labq + ((size_t)img * 3 + 2) * hw; f64 = 1;
    } else if (source == IDC_SRC_OUTPUT_AB_RAW) {
        if (!h->out_resident || img >= h->last_n) return fail(&h->err, IDC_ERR_UNSUPPORTED, "no forward result is resident");
        pa = h->d_out + (size_t)img * 2 * hw; pb = h->d_out + ((size_t)img * 2 + 1) * hw;
    } else if (source == IDC_SRC_INPUT_AB) {
        pa = h->d_ab + (size_t)img * 2 * hw; pb = h->d_ab + ((size_t)img * 2 + 1) * hw;
    } else {
        return fail(&h->err, IDC_ERR_INVALID_ARG, "source %d not in 0..2", source);
    }
    rc = drain_pipeline(h);
    if (rc) return rc;
    const size_t np = (size_t)out_h * out_w;
    if (h->up_cap < np) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->d_up_rgb) (void)hipFree(h->d_up_rgb);
        if (h->d_up_L) (void)hipFree(h->d_up_L);
        if (h->h_up_rgb) (void)hipHostFree(h->h_up_rgb);
        if (h->h_up_L) (void)hipHostFree(h->h_up_L);
        h->d_up_rgb = nullptr; h->d_up_L = nullptr; h->h_up_rgb = nullptr; h->h_up_L = nullptr; h->up_cap = 0;
        HIPCHK(h, hipMalloc((void**)&h->d_up_rgb, np * 3));
        HIPCHK(h, hipMalloc((void**)&h->d_up_L, np * 8));
        HIPCHK(h, hipHostMalloc((void**)&h->h_up_rgb, np * 3, hipHostMallocDefault));
        HIPCHK(h, hipHostMalloc((void**)&h->h_up_L, np * 8, hipHostMallocDefault));
        h->up_cap = np;
    }
    memcpy(h->h_up_L, L, np * 8);
    HIPCHK(h, hipMemcpyAsync(h->d_up_L, h->h_up_L, np * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, launch_upsample_lab2rgb(pa, pb, f64, h->H, h->W, interp, (const double*)h->d_up_L, out_h, out_w, h->d_up_rgb, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->h_up_rgb, h->d_up_rgb, np * 3, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    memcpy(rgb, h->h_up_rgb, np * 3);
    return IDC_OK;
}

int idc_sync(idc_handle h) {
    if (!h) return fail(nullptr, IDC_ERR_INVALID_ARG, "null handle");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return check_chain_abort(h);
}

void* idc_stream(idc_handle h) { return h ? (void*)h->stream : nullptr; }

int idc_num_layers(idc_handle h) { return h ? h->n_timed : 0; }

int idc_layer_info_get(idc_handle h, int layer, idc_layer_info* out) {
    if (!h || !out) return fail(h ? &h->err : nullptr, IDC_ERR_INVALID_ARG, "null argument");
    if (layer < 0 || layer >= h->n_timed) return fail(&h->err, IDC_ERR_INVALID_ARG, "layer %d out of range", layer);
    memset(out, 0, sizeof(*out));
    const int nl = (int)h->layers.size();
    const double hw = (double)h->H * h->W;
    const int eb = elem_bytes(h->precision);
    if (layer == 0) {
        snprintf(out->name, sizeof(out->name), "glob_branch");
        snprintf(out->kernel, sizeof(out->kernel), (h->flags & IDC_FLAG_GLOBAL_HINTS) ? "glob_branch_kernel" : "(input pack fused into conv1_1)");
        out->min_bytes = 0; out->launches = (h->flags & IDC_FLAG_GLOBAL_HINTS) ? 1 : 0;
    } else if (layer <= nl) {
        const Layer& L = h->layers[layer - 1];
        snprintf(out->name, sizeof(out->name), "%s", L.spec->name);
        if (L.skip) {                        // its MACs and bytes are accounted to the launch that runs them
            for (const Layer& C : h->layers)
                if (C.fused_short == layer - 1 || C.fused_next == layer - 1) snprintf(out->kernel, sizeof(out->kernel), "fused into %s", C.spec->name);
            out->flops = 0; out->min_bytes = 0; out->launches = 0;
        } else if (L.chained_into >= 0) {    // ran inside the persistent trunk launch headed by another layer (its own MACs stay its own)
            snprintf(out->kernel, sizeof(out->kernel), "chained into %s", h->layers[L.chained_into].spec->name);
            out->flops = L.flops; out->min_bytes = L.min_bytes; out->launches = 0;
        } else if (L.chain_len > 0) {
            snprintf(out->kernel, sizeof(out->kernel), "conv_kwave_chain_bf16 x%d", L.chain_len);
            out->flops = L.flops; out->min_bytes = L.min_bytes; out->launches = 1;
        } else {
            snprintf(out->kernel, sizeof(out->kernel), L.kw ? (L.spec->kind == kDeconv4x4 ? "conv_kwave_deconv_bf16" : "conv_kwave_bf16") : L.wino ? (L.spec->kind == kDeconv4x4 ? "conv_wino_deconv_f32" : "conv_wino_f32") : L.click ? (L.lprec == IDC_BF16 ? "conv_click<bf16,%d,%d>" : "conv_click<f32,%d,%d>")
                     : L.v2 ? "conv_igemm_v2<%d,%d>" : (L.lprec == IDC_BF16 ? "conv_igemm<bf16,%d,%d>" : "conv_igemm<f32,%d,%d>"),
                     L.cfg.wm, L.cfg.wp);
            if (L.split) snprintf(out->kernel, sizeof(out->kernel), split_is_f16(h->precision) ? (L.v2p ? "conv_igemm_v2psh<%d,%d>x%d" : "conv_igemm_v2sh<%d,%d>x%d")
                                                                                                    : (L.v2p ? "conv_igemm_v2ps<%d,%d>x%d" : "conv_igemm_v2s<%d,%d>x%d"),
                                  L.cfg.wm, L.cfg.wp, split_segments(h->precision));
            else if (L.m16) strncat(out->kernel, L.v2p ? "+m16p" : "+m16", sizeof(out->kernel) - strlen(out->kernel) - 1);
            if (L.split && L.f16fast) snprintf(out->kernel, sizeof(out->kernel), "conv_igemm_v2ph<%d,%d>", L.cfg.wm, L.cfg.wp);
            if (is_split(h->precision) && !L.split && L.spec->kind == kConvIm2col && g_conv1_1_split && !h->tensors[L.dst].is_f32 &&
                (long long)((h->W + 31) / 32) * ((h->H + 15) / 16) * h->max_batch >= 128)
                snprintf(out->kernel, sizeof(out->kernel), "conv1_1_split_kernel");
            if (L.split && L.fused_short < 0 && g_conv1_2_split && conv1_2_split_layer(*L.spec) && !h->tensors[L.dst].is_f32 &&
                (long long)((h->W + 31) / 32) * ((h->H + 11) / 12) * h->max_batch >= 256)
                snprintf(out->kernel, sizeof(out->kernel), "conv1_2_split_kernel x%d", split_segments(h->precision));
            if (L.fused_head) strncat(out->kernel, "+head", sizeof(out->kernel) - strlen(out->kernel) - 1);
            if (L.args.ksplit > 1) {
                char sk[16]; snprintf(sk, sizeof(sk), " splitK%d", L.args.ksplit);
                strncat(out->kernel, sk, sizeof(out->kernel) - strlen(out->kernel) - 1);
            }
            out->flops = L.flops; out->min_bytes = L.min_bytes; out->launches = 1;
            if (L.fused_next >= 0) {
                const Layer& P = h->layers[L.fused_next];
                snprintf(out->kernel, sizeof(out->kernel), "conv1_block_fused");
                out->flops += P.flops;
                out->min_bytes += P.min_bytes - 2.0 * (double)h->tensors[L.dst].H * h->tensors[L.dst].W * h->tensors[L.dst].Cpad * eb;
            }
            if (L.fused_short >= 0) {
                const Layer& P = h->layers[L.fused_short];
                // the name rocprofv3 shows for this launch (the deconv and its 3x3 shortcut conv in one K loop)
                if (L.split && L.f16fast) snprintf(out->kernel, sizeof(out->kernel), "conv_ds_fused_mh+shortcut");
                else if (L.split) snprintf(out->kernel, sizeof(out->kernel), split_is_f16(h->precision) ? "conv_ds_fused_msh+shortcut x%d" : "conv_ds_fused_ms+shortcut x%d", L.args.nseg);
                else snprintf(out->kernel, sizeof(out->kernel), L.m16 ? "conv_ds_fused_m+shortcut" : "conv_ds_fused+shortcut");
                out->flops += P.flops;
                // (the shortcut sums are neither written nor read: fp32 in the operand-split graph, bf16 otherwise)
                out->min_bytes += P.min_bytes - 2.0 * (double)h->tensors[P.dst].H * h->tensors[P.dst].W * h->tensors[P.dst].Cpad * (h->tensors[P.dst].is_f32 ? 4 : eb);
            }
        }
    } else if (layer == nl + 1) {
        snprintf(out->name, sizeof(out->name), "head");
        snprintf(out->kernel, sizeof(out->kernel), "head_kernel");
        out->flops = 2.0 * 128 * 2 * hw; out->min_bytes = hw * 128 * eb + hw * 2 * 4; out->launches = 1;
    } else {
        snprintf(out->name, sizeof(out->name), "dist_softmax");
        snprintf(out->kernel, sizeof(out->kernel), "softmax_nchw_kernel");
        out->launches = (h->flags & IDC_FLAG_DIST_HEAD) ? 1 : 0;
        out->min_bytes = out->launches ? (hw / 16) * (640 + 529) * 4 : 0;
    }
    return IDC_OK;
}

int idc_set_profiling(idc_handle h, int on) {
    if (!h) return fail(nullptr, IDC_ERR_INVALID_ARG, "null handle");
    if (on != 0 && h->ev.empty()) {
        HIPCHK(h, hipSetDevice(h->device));
        h->ev.assign((size_t)h->n_timed * 2 * kProfRing, nullptr);
        for (auto& e : h->ev) HIPCHK(h, hipEventCreate(&e));
    }
    h->profiling = on == 2 ? 2 : (on != 0 ? 1 : 0);
    h->prof_count = 0;
    return IDC_OK;
}

int idc_layer_times_ms(idc_handle h, float* ms, int capacity) {
    if (!h || !ms) return fail(h ? &h->err : nullptr, IDC_ERR_INVALID_ARG, "null argument");
    if (h->ev.empty()) return fail(&h->err, IDC_ERR_INVALID_ARG, "no forward was recorded with profiling on");
    if (capacity < h->n_timed) return fail(&h->err, IDC_ERR_INVALID_ARG, "capacity %d < %d", capacity, h->n_timed);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int slots = (int)(h->prof_count < kProfRing ? h->prof_count : kProfRing);
    if (slots == 0) return fail(&h->err, IDC_ERR_INVALID_ARG, "no forward was recorded with profiling on");
    for (int i = 0; i < h->n_timed; ++i) {
        if (h->profiling == 2 && i > 0) { ms[i] = 0.f; continue; }       // mode 2: ms[0] = the whole forward
        double sum = 0;
        for (int sl = 0; sl < slots; ++sl) {
            float t = 0.f;
            const size_t base = (size_t)sl * h->n_timed * 2;
            if (hipEventElapsedTime(&t, h->ev[base + i * 2], h->ev[base + i * 2 + 1]) != hipSuccess) { t = 0.f; (void)hipGetLastError(); }
            sum += t;
        }
        ms[i] = (float)(sum / slots);
    }
    return IDC_OK;
}

int idc_layer_times_stats(idc_handle h, float* ms_min, float* ms_median, float* ms_max, int capacity) {
    if (!h || !ms_min || !ms_median || !ms_max) return fail(h ? &h->err : nullptr, IDC_ERR_INVALID_ARG, "null argument");
    if (h->ev.empty()) return fail(&h->err, IDC_ERR_INVALID_ARG, "no forward was recorded with profiling on");
    if (capacity < h->n_timed) return fail(&h->err, IDC_ERR_INVALID_ARG, "capacity %d < %d", capacity, h->n_timed);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int slots = (int)(h->prof_count < kProfRing ? h->prof_count : kProfRing);
    if (slots == 0) return fail(&h->err, IDC_ERR_INVALID_ARG, "no forward was recorded with profiling on");
    std::vector<float> v((size_t)slots);
    for (int i = 0; i < h->n_timed; ++i) {
        if (h->profiling == 2 && i > 0) { ms_min[i] = ms_median[i] = ms_max[i] = 0.f; continue; }
        for (int sl = 0; sl < slots; ++sl) {
            float t = 0.f;
            const size_t base = (size_t)sl * h->n_timed * 2;
            if (hipEventElapsedTime(&t, h->ev[base + i * 2], h->ev[base + i * 2 + 1]) != hipSuccess) { t = 0.f; (void)hipGetLastError(); }
            v[(size_t)sl] = t;
        }
        std::sort(v.begin(), v.end());
        ms_min[i] = v.front(); ms_max[i] = v.back();
        ms_median[i] = (slots & 1) ? v[(size_t)slots / 2] : 0.5f * (v[(size_t)slots / 2 - 1] + v[(size_t)slots / 2]);
    }
    return IDC_OK;
}

int idc_get_activation(idc_handle h, const char* name, int n, float* out, size_t capacity_floats, int* C, int* H,
                       int* W) {
    if (!h || !name || !out) return fail(h ? &h->err : nullptr, IDC_ERR_INVALID_ARG, "null argument");
    const int ti = find_tensor(h, name);
    if (ti < 0) return fail(&h->err, IDC_ERR_INVALID_ARG, "no activation named '%s'", name);
    const Tensor& t = h->tensors[ti];
    if (n <= 0 || n > h->max_batch) return fail(&h->err, IDC_ERR_BATCH, "bad n");
    for (size_t li = 0; li < h->layers.size(); ++li) {      // a tensor the last forward never wrote: say so, do not return stale data
        const Layer& L = h->layers[li];
        if (L.dst != ti) continue;
        bool fused_away = L.fused_next >= 0 || L.fused_head; // conv1_1 inside conv1_block_fused; conv10_2 consumed by the head in its own epilogue
        for (const Layer& C : h->layers) fused_away = fused_away || C.fused_short == (int)li;   // shortcut conv inside conv_ds_fused
        if (fused_away)
            return fail(&h->err, IDC_ERR_UNSUPPORTED, "activation '%s' is not materialised: its layer runs fused inside another "
                        "launch (idc_set_option(\"fuse_conv1\", 0) / IDC_FUSE_SHORTCUT=0 keep the launches apart)", name);
    }
    const size_t need = (size_t)n * t.C * t.H * t.W;
    if (capacity_floats < need) return fail(&h->err, IDC_ERR_INVALID_ARG, "need %zu floats", need);
    HIPCHK(h, hipSetDevice(h->device));
    if (h->scratch_bytes < need * 4) {
        if (h->d_scratch) (void)hipFree(h->d_scratch);
        h->d_scratch = nullptr; h->scratch_bytes = 0;
        HIPCHK(h, hipMalloc((void**)&h->d_scratch, need * 4));
        h->scratch_bytes = need * 4;
    }
    const int src_bf16 = (!t.is_f32 && h->precision != IDC_FP32) ? 1 : 0;
    if (t.parts > 1 || (is_split(h->precision) && !t.is_f32)) HIPCHK(h, launch_split_to_nchw(      // (IDC_FP16: one fp16 plane)
        t.ptr, h->d_scratch, n, t.C, t.H, t.W, t.Cpad, t.parts, split_is_f16(h->precision) ? 1 : 0, h->stream));
    else HIPCHK(h, launch_nhwc_to_nchw(src_bf16, t.ptr, h->d_scratch, n, t.C, t.H, t.W, t.Cpad, h->stream));
    HIPCHK(h, hipMemcpyAsync(out, h->d_scratch, need * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (C) *C = t.C;
    if (H) *H = t.H;
    if (W) *W = t.W;
    return IDC_OK;
}

// ---- single operators -----------------------------------------------------------------------------

static int run_single_op(int device_id, int precision, LayerSpec spec, int n, int h, int w, const float* x,
                         const float* weight, const float* bias, const float* bn_scale, const float* bn_shift,
                         const float* resid, float* y) {
    int rc = check_device(device_id, nullptr);
    if (rc) return rc;
    if (precision < IDC_FP32 || precision > IDC_FP16) return fail(nullptr, IDC_ERR_INVALID_ARG, "bad precision");
    if (!x || !weight || !bias || !y || n <= 0 || h <= 0 || w <= 0) return fail(nullptr, IDC_ERR_INVALID_ARG, "bad argument");
    const bool split = is_split(precision);
    const int parts = split_parts(precision);
    if (split && !v2_eligible(spec)) return fail(nullptr, IDC_ERR_UNSUPPORTED, "operand-split precisions run the large tile only: cout >= 65");
    const int kc = kc_elems(precision), eb = elem_bytes(precision);
    if (spec.cin % kc) return fail(nullptr, IDC_ERR_UNSUPPORTED, "cin must be a multiple of %d in this precision", kc);
    if (spec.in_stride != 1 && spec.in_stride != 2) return fail(nullptr, IDC_ERR_INVALID_ARG, "in_stride must be 1 or 2");
    if (h % spec.in_stride || w % spec.in_stride) return fail(nullptr, IDC_ERR_INVALID_ARG, "H, W must divide by in_stride");
    idc_context* nullctx = nullptr;
    HIPCHK(nullctx, hipSetDevice(device_id));
    HIPCHK(nullctx, init_kernels());
    Layer L;
    // the shortcut sum arrives through the `resid` pointer; every eligibility gate of set_geometry reads spec.resid (a 3x3 op with
    // a residual must not be handed to conv_kwave_bf16 / the Winograd forms, which do not take one -- ADVICE r4)
    spec.resid = resid ? "resid" : nullptr;
    L.spec = &spec;
    L.blob.nkc = spec.cin / kc; L.blob.ncg = cout_pad(spec.cout) / kCoutGroup;
    L.blob.w_bytes = (size_t)weight_taps(spec.kind) * L.blob.nkc * L.blob.ncg * kWBlockBytes;
    L.blob.w_off = 0; L.blob.w2_off = (size_t)-1; L.blob.bias_off = L.blob.bn_scale_off = L.blob.bn_shift_off = L.blob.fbias_off = (size_t)-1;
    const int cpad = cout_pad(spec.cout);
    // fp32 3x3 stride-1 ops without a shortcut sum take the Winograd kernel exactly as inside the network
    const bool wino_dc = !split && wino_deconv_eligible(spec) && spec.cin % kc == 0;
    const bool wino_ok = !split && ((wino_eligible(spec) && spec.cin % kc == 0 && resid == nullptr) || wino_dc);
    L.blob.w3_off = wino_ok ? 0 : (size_t)-1;
    L.blob.w3_bytes = wino_ok ? (size_t)spec.cin * cpad * (wino_dc ? 36 : 16) * eb : 0;
    if (wino_ok && L.blob.w3_bytes > L.blob.w_bytes) L.blob.w_bytes = L.blob.w3_bytes;      // one staging buffer serves either image
    std::vector<uint8_t> wimg(L.blob.w_bytes * parts);
    L.lprec = precision; L.split = split;
    const int Hs = h / spec.in_stride, Ws = w / spec.in_stride;
    const int so = spec.kind == kDeconv4x4 ? 2 : 1;
    const int Ho = Hs * so, Wo = Ws * so;
    fill_taps(L);
    // (default library: the large tile only where conv_igemm_v2m / v2p cover the launch -- no shortcut sum, no LeakyReLU without the fused head)
    set_geometry(L, precision, n, n, Hs, Ws, kAbPartners || is_split(precision) || (resid == nullptr && spec.act != 2));
    int op_wexp = 0;
    if (L.wino && wino_dc) pack_wino_deconv_weights(wimg.data(), precision, spec, L.blob, weight);
    else if (L.wino) pack_wino_weights(wimg.data(), precision, spec, L.blob, weight);
    else {
        if (wino_ok) L.blob.w_bytes = (size_t)weight_taps(spec.kind) * L.blob.nkc * L.blob.ncg * kWBlockBytes;
        L.m16 = split || (L.v2 && g_mfma16 && resid == nullptr && spec.act != 2);       // as in the network: conv_igemm_v2m where it applies
        const size_t wcount = (size_t)spec.cin * spec.cout * (spec.kind == kDeconv4x4 ? 16 : spec.kind == kConv1x1 ? 1 : 9);
        op_wexp = (split_is_f16(precision) && parts > 1) ? f16_weight_exponent(weight, wcount) : 0;      // as the blob packer does per layer
        for (int part = 0; part < parts; ++part)
            pack_layer_weights(wimg.data() + (size_t)part * L.blob.w_bytes, precision, (L.v2 && !L.m16) ? 2 : 1, spec, L.blob, weight, part, ldexpf(1.f, op_wexp));
    }
    std::vector<float> hb(cpad, 0.f), hs(cpad, 1.f), ht(cpad, 0.f);
    for (int c = 0; c < spec.cout; ++c) {
        hb[c] = bias[c];
        if (bn_scale) { hs[c] = bn_scale[c]; ht[c] = bn_shift ? bn_shift[c] : 0.f; }
    }
    DevBuf d_x, d_xn, d_w, d_b, d_s, d_t, d_r, d_rn, d_yn, d_y;
    const size_t xin = (size_t)n * spec.cin * h * w, yout = (size_t)n * spec.cout * Ho * Wo;
    HIPCHK(nullctx, d_x.alloc(xin * 4));
    HIPCHK(nullctx, d_xn.alloc(xin * eb * parts));
    HIPCHK(nullctx, d_w.alloc(L.blob.w_bytes * parts));
    HIPCHK(nullctx, d_b.alloc(cpad * 4)); HIPCHK(nullctx, d_s.alloc(cpad * 4)); HIPCHK(nullctx, d_t.alloc(cpad * 4));
    HIPCHK(nullctx, d_yn.alloc((size_t)n * Ho * Wo * cpad * (split ? 2 * parts : 4)));
    HIPCHK(nullctx, d_y.alloc(yout * 4));
    HIPCHK(nullctx, hipMemcpy(d_x.p, x, xin * 4, hipMemcpyHostToDevice));
    HIPCHK(nullctx, hipMemcpy(d_w.p, wimg.data(), L.blob.w_bytes * parts, hipMemcpyHostToDevice));
    HIPCHK(nullctx, hipMemcpy(d_b.p, hb.data(), cpad * 4, hipMemcpyHostToDevice));
    HIPCHK(nullctx, hipMemcpy(d_s.p, hs.data(), cpad * 4, hipMemcpyHostToDevice));
    HIPCHK(nullctx, hipMemcpy(d_t.p, ht.data(), cpad * 4, hipMemcpyHostToDevice));
    if (split) HIPCHK(nullctx, launch_nchw_to_split((const float*)d_x.p, d_xn.p, n, spec.cin, h, w, spec.cin, parts, split_is_f16(precision) ? 1 : 0, nullptr));
    else HIPCHK(nullctx, launch_nchw_to_nhwc(precision, (const float*)d_x.p, d_xn.p, n, spec.cin, h, w, spec.cin, nullptr));
    // bf16 precision: the residual arrives and the output leaves in bf16, as inside the network (operand-split: fp32 residual, split output)
    const int io_bf16 = precision == IDC_BF16 ? 1 : 0;
    if (resid) {
        HIPCHK(nullctx, d_r.alloc(yout * 4));
        HIPCHK(nullctx, d_rn.alloc((size_t)n * Ho * Wo * cpad * 4));
        HIPCHK(nullctx, hipMemcpy(d_r.p, resid, yout * 4, hipMemcpyHostToDevice));
        HIPCHK(nullctx, launch_nchw_to_nhwc(io_bf16, (const float*)d_r.p, d_rn.p, n, spec.cout, Ho, Wo, cpad, nullptr));
    }
    ConvArgs& a = L.args;
    a.in = d_xn.p; a.out = d_yn.p; a.wgt = d_w.p; a.bias = (const float*)d_b.p;
    a.bn_scale = bn_scale ? (const float*)d_s.p : nullptr;
    a.bn_shift = bn_scale ? (const float*)d_t.p : nullptr;
    a.resid = resid ? d_rn.p : nullptr;
    a.resid_bf16 = io_bf16;
    a.head_w = nullptr; a.head_b = nullptr; a.head_out = nullptr; a.head_mul = 0.f;
    a.in2 = nullptr; a.wgt2 = nullptr; a.nkc2 = 0;
    a.warm = g_code_warm;
    a.split_f16 = split_is_f16(precision) ? 1 : 0;
    a.out_f32 = (io_bf16 || split) ? 0 : 1;
    if (split) {
        a.in_parts = parts; a.out_parts = parts; a.nseg = split_segments(precision); a.seg_x = split_seg_x(precision); a.seg_w = split_seg_w(precision);
        a.w_part_bytes = L.blob.w_bytes;
    }
    DevBuf d_part, d_zero, d_wsc;
    a.acc_scale = nullptr;
    if (split && split_is_f16(precision)) {
        const float sc = ldexpf(1.f, -op_wexp);
        HIPCHK(nullctx, d_wsc.alloc(256));
        HIPCHK(nullctx, hipMemcpy(d_wsc.p, &sc, 4, hipMemcpyHostToDevice));
        a.acc_scale = (const float*)d_wsc.p;
    }
    if (a.ksplit > 1) {
        HIPCHK(nullctx, d_part.alloc((size_t)a.ksplit * n * Ho * Wo * cpad * 4));
        a.partial = (float*)d_part.p;
    }
    HIPCHK(nullctx, d_zero.alloc(256));
    HIPCHK(nullctx, hipMemset(d_zero.p, 0, 256));
    a.zeros = d_zero.p;
    if (L.wino && !conv_wino_applies(precision, a, wino_dc))
        return fail(nullptr, IDC_ERR_INTERNAL, "single op: Winograd variant selected for a launch it does not cover");
    if (split && (!L.v2 || !conv_v2s_applies(a)))
        return fail(nullptr, IDC_ERR_INTERNAL, "single op: no operand-split kernel covers this launch");
    if (!split && L.m16 && !conv_v2m_applies(a))
        return fail(nullptr, IDC_ERR_INTERNAL, "single op: conv_igemm_v2m selected for a launch it does not cover");
    L.v2p = split ? (g_v2p && conv_v2ps_applies(L.cfg, L.halo, a)) : (L.m16 && g_v2p && conv_v2p_applies(L.cfg, L.halo, a));
    if (L.kw && !conv_kwave_applies(a))
        return fail(nullptr, IDC_ERR_INTERNAL, "single op: conv_kwave_bf16 selected for a launch it does not cover");
    HIPCHK(nullctx, split ? (L.v2p ? launch_conv_v2ps(L.cfg, L.halo, a, nullptr) : launch_conv_v2s(L.cfg, L.halo, a, nullptr)) : L.kw ? launch_conv_kwave(a, nullptr) : L.wino ? (wino_dc ? launch_deconv_wino(precision, a, nullptr) : launch_conv_wino(precision, a, nullptr)) : L.click ? launch_conv_click(precision, L.cfg.wp, L.halo, a, nullptr)
                    : L.v2 ? (L.v2p ? launch_conv_v2p(L.cfg, L.halo, a, nullptr) : L.m16 ? launch_conv_v2m(L.cfg, L.halo, a, nullptr) : launch_conv_v2(L.cfg, L.halo, a, nullptr))
                           : launch_conv(precision, L.cfg, L.halo, a, nullptr));
    if (a.ksplit > 1) HIPCHK(nullctx, launch_splitk_epilogue(precision, a, nullptr));
    if (split) HIPCHK(nullctx, launch_split_to_nchw(d_yn.p, (float*)d_y.p, n, spec.cout, Ho, Wo, cpad, parts, split_is_f16(precision) ? 1 : 0, nullptr));
    else HIPCHK(nullctx, launch_nhwc_to_nchw(io_bf16, d_yn.p, (float*)d_y.p, n, spec.cout, Ho, Wo, cpad, nullptr));
    HIPCHK(nullctx, hipMemcpy(y, d_y.p, yout * 4, hipMemcpyDeviceToHost));
    HIPCHK(nullctx, hipDeviceSynchronize());
    return IDC_OK;
}

int idc_op_conv2d(int device_id, int precision, int n, int cin, int h, int w, const float* x, int cout, int ksize,
                  int dilation, int in_stride, const float* weight, const float* bias, int act, const float* bn_scale,
                  const float* bn_shift, const float* resid, float* y) {
    if (ksize != 3 && ksize != 1) return fail(nullptr, IDC_ERR_UNSUPPORTED, "ksize must be 1 or 3");
    if (dilation != 1 && dilation != 2) return fail(nullptr, IDC_ERR_UNSUPPORTED, "dilation must be 1 or 2");
    if (ksize == 1 && (dilation != 1 || in_stride != 1)) return fail(nullptr, IDC_ERR_UNSUPPORTED, "1x1 conv: d=1, stride 1 only");
    LayerSpec s{"op", "op", nullptr, ksize == 3 ? kConv3x3 : kConv1x1, cin, cout, dilation, in_stride, act,
                "x", nullptr, 1, 1, 0};
    return run_single_op(device_id, precision, s, n, h, w, x, weight, bias, bn_scale, bn_shift, resid, y);
}

int idc_op_deconv4x4s2(int device_id, int precision, int n, int cin, int h, int w, const float* x, int cout,
                       const float* weight, const float* bias, int act, const float* resid, float* y) {
    LayerSpec s{"op", "op", nullptr, kDeconv4x4, cin, cout, 1, 1, act, "x", nullptr, 1, 1, 0};
    return run_single_op(device_id, precision, s, n, h, w, x, weight, bias, nullptr, nullptr, resid, y);
}

}  // extern "C"
