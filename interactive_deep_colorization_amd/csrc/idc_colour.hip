// idc_colour.hip -- colour space, display and layout helpers around the forward pass: Lab -> RGB, the fused upsample + Lab -> RGB of the display step,
// the global-statistics extractor, the PCIe copy kernel and the NCHW <-> NHWC (split) converters.
#include <stdlib.h>
#include <type_traits>

#include "idc_kernels.h"

#include "idc_layout.h"

#include "idc_common.hip.h"

namespace idc {

// ------------------------------------------------------------------------------------------------
// lab_post: skimage.color.lab2rgb -> uint8 -> skimage.color.rgb2lab, per pixel, in float64 (the reference
// computes this on the host in float64 inside every net_forward: colorize_image.py:20-36,196-198,264-267).
// Same constants and operation order as oracle/colorspace.py.  Elementwise, one thread
// per pixel; 65536 pixels per 256x256 image -- latency-, not bandwidth-relevant (it removes ~10 ms of host
// numpy from the per-click path).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lab_post_kernel(const float* __restrict__ Lp, float l_add, const float* __restrict__ ab,
                                                       unsigned char* __restrict__ rgb, double* __restrict__ lab_q,
                                                       long long npix, int HW) {
    const double M[3][3] = {{0.412453, 0.357580, 0.180423}, {0.212671, 0.715160, 0.072169}, {0.019334, 0.119193, 0.950227}};
    // inverse of M (numpy.linalg.inv of the matrix above, float64)
    const double Mi[3][3] = {{3.240481343200526, -1.5371515162713185, -0.4985363261688878},
                             {-0.9692549499965682, 1.8759900014898907, 0.04155592655829284},
                             {0.05564663913517716, -0.20404133836651123, 1.0573110696453443}};
    const double white[3] = {0.95047, 1.0, 1.08883};
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
        const long long n = p / HW, r = p - n * HW;
        const double L = (double)Lp[p] + (double)l_add;
        const double a = (double)ab[(n * 2 + 0) * HW + r], b = (double)ab[(n * 2 + 1) * HW + r];
        double f[3];
        f[1] = (L + 16.0) / 116.0;
        f[0] = a / 500.0 + f[1];
        f[2] = fmax(f[1] - b / 200.0, 0.0);                                  // skimage zeroes negative z
        double xyz[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) xyz[i] = (f[i] > 0.2068966 ? f[i] * f[i] * f[i] : (f[i] - 16.0 / 116.0) / 7.787) * white[i];
        unsigned char q[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double lin = xyz[0] * Mi[c][0] + xyz[1] * Mi[c][1] + xyz[2] * Mi[c][2];
            double s = lin > 0.0031308 ? 1.055 * pow(fmax(lin, 0.0), 1.0 / 2.4) - 0.055 : 12.92 * lin;
            s = fmin(fmax(s, 0.0), 1.0);
            q[c] = (unsigned char)(s * 255.0);                               // astype('uint8'): truncation
            rgb[p * 3 + c] = q[c];
        }
        if (lab_q != nullptr) {
            double lin[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double v = (double)q[c] / 255.0;
                lin[c] = v > 0.04045 ? pow((v + 0.055) / 1.055, 2.4) : v / 12.92;
            }
            double g[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double t = (lin[0] * M[i][0] + lin[1] * M[i][1] + lin[2] * M[i][2]) / white[i];
                g[i] = t > 0.008856 ? cbrt(t) : 7.787 * t + 16.0 / 116.0;
            }
            lab_q[(n * 3 + 0) * HW + r] = 116.0 * g[1] - 16.0;
            lab_q[(n * 3 + 1) * HW + r] = 500.0 * (g[0] - g[1]);
            lab_q[(n * 3 + 2) * HW + r] = 200.0 * (g[1] - g[2]);
        }
    }
}

hipError_t launch_lab_post(const float* L, float l_add, const float* ab, unsigned char* rgb, double* lab_q, int N,
                           int H, int W, hipStream_t s) {
    const long long npix = (long long)N * H * W;
    const int blocks = (int)((npix + 255) / 256 < 4096 ? (npix + 255) / 256 : 4096);
    hipLaunchKernelGGL(lab_post_kernel, dim3(blocks), dim3(256), 0, s, L, l_add, ab, rgb, lab_q, npix, H * W);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// pcie_copy: a click's host <-> device transfers as a KERNEL on the forward's own stream (round 5).  One side of (dst, src) is pinned host
// memory mapped into the device's address space, the other is HBM; 16 bytes per lane, one pass.  hipMemcpyAsync hands the same bytes to a copy
// engine on another queue: two cross-queue hand-overs per copy, which at 0.2-0.8 MB weigh more than the bytes (tools/click_host_breakdown.py:
// 768 KB in, 37 us through the copy engine).  Batches keep the copy engines: there the bytes dominate and the compute units have better things to do.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pcie_copy_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, unsigned n16) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) dst[i] = src[i];
}

hipError_t launch_pcie_copy(void* dst, const void* src, size_t bytes, hipStream_t s) {       // bytes % 16 == 0, both 16-byte aligned
    const unsigned n16 = (unsigned)(bytes / 16);
    if (n16 == 0) return hipSuccess;
    const unsigned blocks = (n16 + 255) / 256 < 1024 ? (n16 + 255) / 256 : 1024;
    hipLaunchKernelGGL(pcie_copy_kernel, dim3(blocks), dim3(256), 0, s, (uint4*)dst, (const uint4*)src, n16);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// upsample_lab2rgb: the display step that follows every net_forward in the GUI (ui/gui_draw.py:280-283):
//     ab_win = cv2.resize(output_ab, (win_w, win_h), interpolation=cv2.INTER_CUBIC); lab2rgb(concat(l_win, ab_win)) -> uint8
// and the full-resolution getters (data/colorize_image.py:123-158): scipy.ndimage.zoom(ab, order=1 | 0) + lab2rgb with
// the full-resolution L.  One thread per OUTPUT pixel: interpolate (a, b) from the resident planes, then the float64
// Lab -> sRGB -> uint8 of lab_post_kernel.
//   interp 0: cv2 INTER_CUBIC as resize.cpp computes it for 64F data -- source coordinate fx = (float)((dx + .5) * scale
//             - .5), taps sx-1 .. sx+2 clamped to the image, float32 Keys coefficients with A = -0.75
//             (interpolateCubic), rows first (four horizontal sums in double, left to right), then the vertical sum;
//   interp 1: scipy.ndimage.zoom(order=1): coordinate = dst * (in - 1) / (out - 1), linear, double;
//   interp 2: scipy.ndimage.zoom(order=0): nearest of the same coordinate (floor(c + .5)).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lab_to_rgb_u8(double L, double a, double b, unsigned char* q) {
    const double Mi[3][3] = {{3.240481343200526, -1.5371515162713185, -0.4985363261688878},
                             {-0.9692549499965682, 1.8759900014898907, 0.04155592655829284},
                             {0.05564663913517716, -0.20404133836651123, 1.0573110696453443}};
    const double white[3] = {0.95047, 1.0, 1.08883};
    double f[3];
    f[1] = (L + 16.0) / 116.0;
    f[0] = a / 500.0 + f[1];
    f[2] = fmax(f[1] - b / 200.0, 0.0);
    double xyz[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) xyz[i] = (f[i] > 0.2068966 ? f[i] * f[i] * f[i] : (f[i] - 16.0 / 116.0) / 7.787) * white[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double lin = xyz[0] * Mi[c][0] + xyz[1] * Mi[c][1] + xyz[2] * Mi[c][2];
        double s = lin > 0.0031308 ? 1.055 * pow(fmax(lin, 0.0), 1.0 / 2.4) - 0.055 : 12.92 * lin;
        s = fmin(fmax(s, 0.0), 1.0);
        q[c] = (unsigned char)(s * 255.0);
    }
}

__device__ __forceinline__ void cubic_coeffs(float x, float* c) {       // cv2 interpolateCubic
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

template <typename S>
__global__ __launch_bounds__(256) void upsample_lab2rgb_kernel(const S* __restrict__ pa, const S* __restrict__ pb, int H, int W,
                                                               int interp, const double* __restrict__ Lout, int oh, int ow,
                                                               unsigned char* __restrict__ rgb) {
    const long long npix = (long long)oh * ow;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
        const int dy = (int)(p / ow), dx = (int)(p - (long long)dy * ow);
        double ab[2];
        if (interp == 0) {
            const double sc_x = (double)W / ow, sc_y = (double)H / oh;
            float fx = (float)((dx + 0.5) * sc_x - 0.5), fy = (float)((dy + 0.5) * sc_y - 0.5);
            const int sx = (int)floorf(fx), sy = (int)floorf(fy);
            fx -= sx; fy -= sy;
            float cx[4], cy[4];
            cubic_coeffs(fx, cx); cubic_coeffs(fy, cy);
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const S* src = ch ? pb : pa;
                double rows[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int yy = sy - 1 + k; yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
                    double v = 0.0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int xx = sx - 1 + j; xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
                        v += (double)src[(size_t)yy * W + xx] * (double)cx[j];
                    }
                    rows[k] = v;
                }
                ab[ch] = rows[0] * (double)cy[0] + rows[1] * (double)cy[1] + rows[2] * (double)cy[2] + rows[3] * (double)cy[3];
            }
        } else {
            const double zy = oh > 1 ? (double)(H - 1) / (double)(oh - 1) : 0.0, zx = ow > 1 ? (double)(W - 1) / (double)(ow - 1) : 0.0;
            const double cyy = dy * zy, cxx = dx * zx;
            if (interp == 2) {
                int yy = (int)floor(cyy + 0.5), xx = (int)floor(cxx + 0.5);
                yy = yy > H - 1 ? H - 1 : yy; xx = xx > W - 1 ? W - 1 : xx;
                ab[0] = (double)pa[(size_t)yy * W + xx]; ab[1] = (double)pb[(size_t)yy * W + xx];
            } else {
                const int y0 = (int)floor(cyy), x0 = (int)floor(cxx);
                const double ty = cyy - y0, tx = cxx - x0;
                const int y1 = y0 + 1 > H - 1 ? H - 1 : y0 + 1, x1 = x0 + 1 > W - 1 ? W - 1 : x0 + 1;   // weight 0 there
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    const S* src = ch ? pb : pa;
                    const double v00 = (double)src[(size_t)y0 * W + x0], v01 = (double)src[(size_t)y0 * W + x1];
                    const double v10 = (double)src[(size_t)y1 * W + x0], v11 = (double)src[(size_t)y1 * W + x1];
                    ab[ch] = v00 * ((1.0 - ty) * (1.0 - tx)) + v01 * ((1.0 - ty) * tx) + v10 * (ty * (1.0 - tx)) + v11 * (ty * tx);
                }
            }
        }
        unsigned char q[3];
        lab_to_rgb_u8(Lout[p], ab[0], ab[1], q);
        rgb[p * 3 + 0] = q[0]; rgb[p * 3 + 1] = q[1]; rgb[p * 3 + 2] = q[2];
    }
}

hipError_t launch_upsample_lab2rgb(const void* a_plane, const void* b_plane, int src_f64, int H, int W, int interp, const double* L_out,
                                   int oh, int ow, unsigned char* rgb, hipStream_t s) {
    const long long npix = (long long)oh * ow;
    if (npix <= 0 || interp < 0 || interp > 2) return hipErrorInvalidValue;
    const int blocks = (int)((npix + 255) / 256 < 8192 ? (npix + 255) / 256 : 8192);
    if (src_f64)
        hipLaunchKernelGGL(upsample_lab2rgb_kernel<double>, dim3(blocks), dim3(256), 0, s, (const double*)a_plane, (const double*)b_plane, H, W,
                           interp, L_out, oh, ow, rgb);
    else
        hipLaunchKernelGGL(upsample_lab2rgb_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)a_plane, (const float*)b_plane, H, W,
                           interp, L_out, oh, ow, rgb);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// global_stats: the reference's global_stats.prototxt on one reference image -- rgb2lab per pixel (float64, the
// skimage formulas of lab_post_kernel), 4x4 average pool of ab (Pooling AVE k4 s4, :101-111), hard assignment of
// each pooled value to its nearest of the 313 centres (NNEncLayer with NN = 1, caffe_traininglayers.py:161-196),
// counted with integer atomics (deterministic); plus the sum of the HSV saturation (BGR2HSVLayer :53-85).
// One thread per 4x4 block.  A 256x256 image is 4096 blocks: latency-, not bandwidth-relevant.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void rgb8_to_lab(const unsigned char* q, double& L, double& a, double& b) {
    const double M[3][3] = {{0.412453, 0.357580, 0.180423}, {0.212671, 0.715160, 0.072169}, {0.019334, 0.119193, 0.950227}};
    const double white[3] = {0.95047, 1.0, 1.08883};
    double lin[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double v = (double)q[c] / 255.0;
        lin[c] = v > 0.04045 ? pow((v + 0.055) / 1.055, 2.4) : v / 12.92;
    }
    double g[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double t = (lin[0] * M[i][0] + lin[1] * M[i][1] + lin[2] * M[i][2]) / white[i];
        g[i] = t > 0.008856 ? cbrt(t) : 7.787 * t + 16.0 / 116.0;
    }
    L = 116.0 * g[1] - 16.0; a = 500.0 * (g[0] - g[1]); b = 200.0 * (g[1] - g[2]);
}

__global__ __launch_bounds__(256) void global_stats_kernel(const unsigned char* __restrict__ rgb, const float* __restrict__ centres,
                                                           unsigned* __restrict__ counts, double* __restrict__ sat_sum,
                                                           int N, int H, int W) {
    __shared__ float cc[313 * 2];
    for (int i = threadIdx.x; i < 626; i += blockDim.x) cc[i] = centres[i];
    __syncthreads();
    const int h4 = H >> 2, w4 = W >> 2;
    const long long nblk = (long long)N * h4 * w4;
    for (long long blk = (long long)blockIdx.x * blockDim.x + threadIdx.x; blk < nblk; blk += (long long)gridDim.x * blockDim.x) {
        const int bx = (int)(blk % w4), by = (int)((blk / w4) % h4), n = (int)(blk / ((long long)w4 * h4));
        double sa = 0.0, sb = 0.0, ssat = 0.0;
        for (int dy = 0; dy < 4; ++dy)
            for (int dx = 0; dx < 4; ++dx) {
                const unsigned char* q = rgb + (((size_t)n * H + by * 4 + dy) * W + bx * 4 + dx) * 3;
                double L, a, b;
                rgb8_to_lab(q, L, a, b);
                sa += a; sb += b;
                const double r = q[0] / 255.0, g = q[1] / 255.0, bl = q[2] / 255.0;
                const double mx = fmax(r, fmax(g, bl)), mn = fmin(r, fmin(g, bl));
                ssat += mx > 0.0 ? (mx - mn) / mx : 0.0;                     // skimage rgb2hsv saturation
            }
        const float pa = (float)(sa / 16.0), pb = (float)(sb / 16.0);      // Caffe blobs are fp32
        int best = 0;
        float bd = 3.0e38f;
        for (int k = 0; k < 313; ++k) {
            const float da = pa - cc[2 * k], db = pb - cc[2 * k + 1];
            const float d = da * da + db * db;
            if (d < bd) { bd = d; best = k; }
        }
        atomicAdd(&counts[(size_t)n * 313 + best], 1u);
        atomicAdd(&sat_sum[n], ssat);
    }
}

hipError_t launch_global_stats(const unsigned char* rgb, const float* centres, unsigned* counts, double* sat_sum, int N,
                               int H, int W, hipStream_t s) {
    const long long nblk = (long long)N * (H / 4) * (W / 4);
    const int blocks = (int)((nblk + 255) / 256 < 1024 ? (nblk + 255) / 256 : 1024);
    hipLaunchKernelGGL(global_stats_kernel, dim3(blocks), dim3(256), 0, s, rgb, centres, counts, sat_sum, N, H, W);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// layout converters (test entry points / activation dumps only -- not on the hot path)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int N, int C, int H, int W,
                                    int Cpad) {
    const long long total = (long long)N * H * W * Cpad;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cpad);
        const long long pix = i / Cpad;
        const long long hw = (long long)H * W;
        const long long n = pix / hw, r = pix - n * hw;
        const float v = c < C ? src[(n * C + c) * hw + r] : 0.f;
        if (sizeof(T) == 4) ((float*)dst)[i] = v;
        else ((__bf16*)dst)[i] = (__bf16)v;
    }
}

__global__ void nhwc_to_nchw_kernel(const void* __restrict__ src, float* __restrict__ dst, int N, int C, int H, int W,
                                    int Cstride, int src_is_bf16) {
    const long long total = (long long)N * C * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long hw = (long long)H * W;
        const long long r = i % hw;
        const int c = (int)((i / hw) % C);
        const long long n = i / (hw * C);
        const long long sidx = (n * hw + r) * Cstride + c;
        float v;
        if (src_is_bf16) v = __uint_as_float((unsigned)((const unsigned short*)src)[sidx] << 16);
        else v = ((const float*)src)[sidx];
        dst[i] = v;
    }
}

hipError_t launch_nchw_to_nhwc(int precision, const float* src, void* dst, int N, int C, int H, int W, int Cpad,
                               hipStream_t s) {
    const long long total = (long long)N * H * W * Cpad;
    const int blocks = (int)((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535);
    if (precision == 1)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, src, (__bf16*)dst, N, C, H, W, Cpad);
    else
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(blocks), dim3(256), 0, s, src, (float*)dst, N, C, H, W, Cpad);
    return hipGetLastError();
}

hipError_t launch_nhwc_to_nchw(int src_is_bf16, const void* src, float* dst, int N, int C, int H, int W, int Cstride,
                               hipStream_t s) {
    const long long total = (long long)N * C * H * W;
    const int blocks = (int)((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(blocks), dim3(256), 0, s, src, dst, N, C, H, W, Cstride, src_is_bf16);
    return hipGetLastError();
}

// ---- operand-split tensors (IDC_BF16X3 / IDC_BF16X6 / IDC_FP16X3): a pixel is [parts][Cpad] bf16 (fp16), x = part 0 + part 1 (+ part 2) ----
__device__ __forceinline__ unsigned short bf16_rne_bits(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }

// test entry points / activation dumps only
__global__ void split_to_nchw_kernel(const unsigned short* __restrict__ src, float* __restrict__ dst, int N, int C, int H, int W, int Cpad, int parts, int f16) {
    const long long total = (long long)N * C * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long hw = (long long)H * W;
        const long long r = i % hw;
        const int c = (int)((i / hw) % C);
        const long long n = i / (hw * C);
        float v = 0.f;
        for (int p = 0; p < parts; ++p) {
            const unsigned short q = src[((n * hw + r) * parts + p) * Cpad + c];
            v += f16 ? (float)__builtin_bit_cast(_Float16, q) : __uint_as_float((unsigned)q << 16);
        }
        dst[i] = v;
    }
}

hipError_t launch_split_to_nchw(const void* src, float* dst, int N, int C, int H, int W, int Cpad, int parts, int f16, hipStream_t s) {
    const long long total = (long long)N * C * H * W;
    const int blocks = (int)((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535);
    hipLaunchKernelGGL(split_to_nchw_kernel, dim3(blocks), dim3(256), 0, s, (const unsigned short*)src, dst, N, C, H, W, Cpad, parts, f16);
    return hipGetLastError();
}

__global__ void nchw_to_split_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int N, int C, int H, int W, int Cpad, int parts, int f16) {
    const long long total = (long long)N * H * W * Cpad;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cpad);
        const long long pix = i / Cpad;
        const long long hw = (long long)H * W;
        const long long n = pix / hw, r = pix - n * hw;
        float v = c < C ? src[(n * C + c) * hw + r] : 0.f;
        for (int p = 0; p < parts; ++p) {
            unsigned short h;
            if (f16) { const _Float16 q = (_Float16)fminf(fmaxf(v, -65504.f), 65504.f); h = __builtin_bit_cast(unsigned short, q); v -= (float)q; }
            else { h = bf16_rne_bits(v); v -= __uint_as_float((unsigned)h << 16); }
            dst[(pix * parts + p) * Cpad + c] = h;
        }
    }
}

hipError_t launch_nchw_to_split(const float* src, void* dst, int N, int C, int H, int W, int Cpad, int parts, int f16, hipStream_t s) {
    const long long total = (long long)N * H * W * Cpad;
    const int blocks = (int)((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535);
    hipLaunchKernelGGL(nchw_to_split_kernel, dim3(blocks), dim3(256), 0, s, src, (unsigned short*)dst, N, C, H, W, Cpad, parts, f16);
    return hipGetLastError();
}


}  // namespace idc
