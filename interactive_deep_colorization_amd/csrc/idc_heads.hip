// idc_heads.hip -- what follows the conv stack: the regression head, the classification softmax, the Caffe 313-bin decode and the
// Global-Hints branch.  models/pytorch/model.py:108-113,174-175; models/reference_model/deploy_nopred.prototxt; models/global_model/deploy_nodist.prototxt.
#include <stdlib.h>
#include <type_traits>

#include "idc_kernels.h"

#include "idc_layout.h"

#include "idc_common.hip.h"

namespace idc {

// ------------------------------------------------------------------------------------------------
// head: model_out = Conv1x1(128->2) -> Tanh, then *110 (model.py:108-109,174-175).
// 16 lanes per pixel, 8 channels each, xor-shuffle reduction inside the 16-lane group.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void head_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ b, float* __restrict__ out,
                                                   long long npix, int HW, float out_mul) {
    const int sub = threadIdx.x & 15;
    float w0[8], w1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { w0[i] = w[sub * 8 + i]; w1[i] = w[128 + sub * 8 + i]; }
    const float b0 = b[0], b1 = b[1];
    const long long stride = (long long)gridDim.x * (blockDim.x >> 4);
    for (long long p = (long long)blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4); p < npix; p += stride) {
        float xv[8];
        if (sizeof(T) == 4) {
            const float4* xp = (const float4*)((const float*)x + p * 128 + sub * 8);
            const float4 a0 = xp[0], a1 = xp[1];
            xv[0] = a0.x; xv[1] = a0.y; xv[2] = a0.z; xv[3] = a0.w;
            xv[4] = a1.x; xv[5] = a1.y; xv[6] = a1.z; xv[7] = a1.w;
        } else {
            const uint4 u = *(const uint4*)((const unsigned short*)x + p * 128 + sub * 8);
            const unsigned uu[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xv[2 * i] = __uint_as_float(uu[i] << 16);
                xv[2 * i + 1] = __uint_as_float(uu[i] & 0xffff0000u);
            }
        }
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { s0 = fmaf(xv[i], w0[i], s0); s1 = fmaf(xv[i], w1[i], s1); }
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) {
            s0 += __shfl_xor(s0, m, 16);
            s1 += __shfl_xor(s1, m, 16);
        }
        if (sub == 0) {
            const long long n = p / HW, r = p - n * HW;
            out[(n * 2 + 0) * HW + r] = tanhf(s0 + b0) * out_mul;
            out[(n * 2 + 1) * HW + r] = tanhf(s1 + b1) * out_mul;
        }
    }
}

hipError_t launch_head(int precision, const void* x, const float* w, const float* b, float* out, int N, int H, int W,
                       float out_mul, hipStream_t s) {
    const long long npix = (long long)N * H * W;
    const long long want = (npix + 15) / 16;
    const int blocks = (int)(want < 8192 ? want : 8192);
    if (precision == 1)
        hipLaunchKernelGGL(head_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, (const __bf16*)x, w, b, out, npix, H * W, out_mul);
    else
        hipLaunchKernelGGL(head_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)x, w, b, out, npix, H * W, out_mul);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// softmax over channels (model.py:160: softmax(model_class(conv8_3) * .2)), one wave per pixel,
// 64-lane shuffle reductions; writes NCHW.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_nchw_kernel(const float* __restrict__ logits, float* __restrict__ out,
                                                           long long npix, int HW, int nclass, int cstride,
                                                           float temperature) {
    const int lane = threadIdx.x & 63;
    const long long stride = (long long)gridDim.x * (blockDim.x >> 6);
    for (long long p = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); p < npix; p += stride) {
        const float* row = logits + p * cstride;
        float v[16];
        float m = -3.0e38f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = lane + i * 64;
            v[i] = c < nclass ? row[c] * temperature : -3.0e38f;
            m = fmaxf(m, v[i]);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = lane + i * 64;
            v[i] = c < nclass ? expf(v[i] - m) : 0.f;
            sum += v[i];
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float inv = 1.0f / sum;
        const long long n = p / HW, r = p - n * HW;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = lane + i * 64;
            if (c < nclass) out[(n * nclass + c) * HW + r] = v[i] * inv;
        }
    }
}

hipError_t launch_softmax_nchw(const float* logits, float* out, int N, int H, int W, int nclass, int cstride,
                               float temperature, hipStream_t s) {
    if (nclass > 1024) return hipErrorInvalidValue;
    const long long npix = (long long)N * H * W;
    const long long want = (npix + 3) / 4;
    const int blocks = (int)(want < 8192 ? want : 8192);
    hipLaunchKernelGGL(softmax_nchw_kernel, dim3(blocks), dim3(256), 0, s, logits, out, npix, H * W, nclass, cstride,
                       temperature);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// dist313: bilinear x4 upsample of the 313 logits + two channel softmaxes + annealed-mean decode.
// One wave per 4x4 block of output pixels (they share the same four quarter-resolution neighbours, read once:
// 4 x 1252 B coalesced); lane l owns bins l, l+64, ... (5 per lane); 64-lane xor-shuffle reductions.
// HBM-bound only when dist_S is requested (313 floats per output pixel); otherwise L2-resident.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dist313_kernel(const float* __restrict__ logits, const float* __restrict__ w_ab,
                                                      float* __restrict__ pred_ab, float* __restrict__ dist_S, int N,
                                                      int H, int W, int cstride, float S, float T) {
    constexpr int NB = 313, PER = 5;
    const int lane = threadIdx.x & 63;
    const int h4 = H >> 2, w4 = W >> 2;
    const long long nblk = (long long)N * h4 * w4;
    float wa[PER], wb[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int q = lane + i * 64;
        wa[i] = q < NB ? w_ab[q] : 0.f;
        wb[i] = q < NB ? w_ab[NB + q] : 0.f;
    }
    const float ba = w_ab[2 * NB], bb = w_ab[2 * NB + 1];
    const long long stride = (long long)gridDim.x * (blockDim.x >> 6);
    for (long long blk = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); blk < nblk; blk += stride) {
        const int n0 = (int)(blk % w4), m0 = (int)((blk / w4) % h4), n = (int)(blk / ((long long)w4 * h4));
        float l00[PER], l01[PER], l10[PER], l11[PER];
        const float* base = logits + ((size_t)n * h4 * w4) * cstride;
        const bool has_r = n0 + 1 < w4, has_d = m0 + 1 < h4;          // beyond the far border the deconv sees zeros
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int q = lane + i * 64;
            const bool ok = q < NB;
            l00[i] = ok ? base[((size_t)m0 * w4 + n0) * cstride + q] : 0.f;
            l01[i] = (ok && has_r) ? base[((size_t)m0 * w4 + n0 + 1) * cstride + q] : 0.f;
            l10[i] = (ok && has_d) ? base[((size_t)(m0 + 1) * w4 + n0) * cstride + q] : 0.f;
            l11[i] = (ok && has_r && has_d) ? base[((size_t)(m0 + 1) * w4 + n0 + 1) * cstride + q] : 0.f;
        }
        for (int jy = 0; jy < 4; ++jy) {
            const float wy1 = 0.25f * jy, wy0 = 1.f - wy1;
            for (int jx = 0; jx < 4; ++jx) {
                const float wx1 = 0.25f * jx, wx0 = 1.f - wx1;
                float v[PER];
                float mx = -3.0e38f;
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    // second x2 stage applied to the first (exactly the composition of the two Caffe layers)
                    v[i] = wy0 * (wx0 * l00[i] + wx1 * l01[i]) + wy1 * (wx0 * l10[i] + wx1 * l11[i]);
                    mx = fmaxf(mx, (lane + i * 64) < NB ? v[i] : -3.0e38f);
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
                // softmax is shift-invariant; S, T > 0 so the same max serves both temperatures
                float es[PER], sumS = 0.f, sumT = 0.f, accA = 0.f, accB = 0.f;
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const bool ok = (lane + i * 64) < NB;
                    es[i] = ok ? expf(S * (v[i] - mx)) : 0.f;
                    const float et = ok ? expf(T * (v[i] - mx)) : 0.f;
                    sumS += es[i]; sumT += et;
                    accA = fmaf(et, wa[i], accA); accB = fmaf(et, wb[i], accB);
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) {
                    sumS += __shfl_xor(sumS, o, 64); sumT += __shfl_xor(sumT, o, 64);
                    accA += __shfl_xor(accA, o, 64); accB += __shfl_xor(accB, o, 64);
                }
                const int y = m0 * 4 + jy, x = n0 * 4 + jx;
                const size_t hw = (size_t)H * W, pix = (size_t)y * W + x;
                if (lane == 0) {
                    pred_ab[((size_t)n * 2 + 0) * hw + pix] = accA / sumT + ba;
                    pred_ab[((size_t)n * 2 + 1) * hw + pix] = accB / sumT + bb;
                }
                if (dist_S != nullptr) {
                    const float inv = 1.0f / sumS;
#pragma unroll
                    for (int i = 0; i < PER; ++i) {
                        const int q = lane + i * 64;
                        if (q < NB) dist_S[((size_t)n * NB + q) * hw + pix] = es[i] * inv;
                    }
                }
            }
        }
    }
}

hipError_t launch_dist313(const float* logits, const float* w_ab, float* pred_ab, float* dist_S, int N, int H, int W,
                          int cstride, float S, float T, hipStream_t s) {
    const long long nblk = (long long)N * (H / 4) * (W / 4);
    const long long want = (nblk + 3) / 4;
    const int blocks = (int)(want < 16384 ? want : 16384);
    hipLaunchKernelGGL(dist313_kernel, dim3(blocks), dim3(256), 0, s, logits, w_ab, pred_ab, dist_S, N, H, W, cstride, S, T);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Global-hints branch: four 1x1 conv + ReLU + BN stages on a 1x1 "image" = four GEMVs per image
// (models/global_model/deploy_nodist.prototxt:37-172).  One workgroup per image, thread c owns output
// channel c; weights are stored transposed [k][512] so that a wave reads 256 contiguous bytes per k.
// ~1 MMAC per image: latency-bound, runs once per forward ahead of the conv stack.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void glob_branch_kernel(const float* __restrict__ in, const float* __restrict__ p,
                                                          float* __restrict__ out) {
    __shared__ float x[kGlobC];
    const int c = threadIdx.x, n = blockIdx.x;
    const float* g = in + (size_t)n * kGlobIn;
    if (c < kGlobIn) x[c] = g[c];
    __syncthreads();
    const float* w = p;                                   // stage 1: [316][512], rows 0..313 glob_conv1, 314..315 s_conv1
    float acc = 0.f;
    for (int k = 0; k < kGlobIn; ++k) acc = fmaf(w[(size_t)k * kGlobC + c], x[k], acc);
    const float* q = p + (size_t)kGlobIn * kGlobC;        // bias (bg + bs), bn scale, bn shift
    float y = fmaf(fmaxf(acc + q[c], 0.f), q[kGlobC + c], q[2 * kGlobC + c]);
    q += 3 * kGlobC;
    for (int stage = 0; stage < 3; ++stage) {
        __syncthreads();
        x[c] = y;
        __syncthreads();
        acc = 0.f;
        for (int k = 0; k < kGlobC; ++k) acc = fmaf(q[(size_t)k * kGlobC + c], x[k], acc);
        const float* r = q + (size_t)kGlobC * kGlobC;
        y = fmaf(fmaxf(acc + r[c], 0.f), r[kGlobC + c], r[2 * kGlobC + c]);
        q = r + 3 * kGlobC;
    }
    out[(size_t)n * kGlobC + c] = y;
}

hipError_t launch_glob_branch(const float* in, const float* params, float* out, int N, hipStream_t s) {
    hipLaunchKernelGGL(glob_branch_kernel, dim3(N), dim3(512), 0, s, in, params, out);
    return hipGetLastError();
}


}  // namespace idc
