// idc_common.hip.h -- device helpers shared by the conv kernel families (idc_igemm.hip, idc_v2.hip, idc_conv1.hip): vector typedefs, the tuning
// harness's cycle stamps, MFMA wrappers on 16-byte fragments, bf16 packing, the XCD-aware block remap and the small-tile epilogue.
// (Round 6: idc_kernels.hip, 3029 lines, split by family -- igemm / v2 / conv1 / heads / colour.)
#pragma once
#include "idc_kernels.h"
#include "idc_layout.h"

namespace idc {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;   // 16-byte slot held in registers
typedef __attribute__((ext_vector_type(16))) float f32x16;    // one 32x32 MFMA accumulator tile

// in-kernel cycle stamps for the tuning harness (tools/ablate): compiled out of the library
#ifdef IDC_TIMING
extern __device__ long long* g_idc_dbg;      // defined in idc_igemm.hip
#define IDC_STAMP(i) do { if (tid == 0) g_idc_dbg[(size_t)blockIdx.x * 16 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define IDC_STAMP(i) do {} while (0)
#endif
#ifdef IDC_TIMING_FINE
#define IDC_STAMP_FINE(i) IDC_STAMP(i)
#else
#define IDC_STAMP_FINE(i) do {} while (0)
#endif

// ------------------------------------------------------------------------------------------------
// MFMA wrappers on 16-byte fragments.  A lane (row/col = lane&15, group g = lane>>4) holds the
// 16-byte slot (ks*4+g) of its row; the K index it stands for is the same permutation for both
// operands, so the contraction is exact whatever the order.
// ------------------------------------------------------------------------------------------------
template <typename T> struct Mma;
template <> struct Mma<__bf16> {
    static __device__ __forceinline__ void run(f32x4& acc, const u32x4& w, const u32x4& x) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w),
                                                      __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static __device__ __forceinline__ void run(f32x4& acc, const u32x4& w, const u32x4& x) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.x), __uint_as_float(x.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.y), __uint_as_float(x.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.z), __uint_as_float(x.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.w), __uint_as_float(x.w), acc, 0, 0, 0);
    }
};

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    // one v_cvt_pk_bf16_f32 (RNE) as a VECTOR conversion: from `(__bf16)lo | (__bf16)hi << 16` the vectoriser pairs the conversions of NEIGHBOURING packs
    // and un-shuffles them with and / shift / two SDWA ors -- six instructions for two dwords instead of two (round 5: the epilogues are VALU-bound).
    // (Not inline asm: the hazard recogniser does not see an asm's reads of MFMA results, and the scheduler may move it next to the MFMAs.)
    typedef float f32x2_pk __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_pk __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_pk){lo, hi}, bf16x2_pk));
}

// XCD-aware, bijective block remap: hardware places block b on XCD b%8; give each XCD a
// contiguous range of the logical order so neighbouring tiles (same weights, shared halo) share
// one L2.  Speed only -- any placement is correct.
__device__ __forceinline__ int xcd_remap(int b, int nb) {
    const int xcd = b & 7, q = nb >> 3, r = nb & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (b >> 3);
}


// Fused epilogue arithmetic for 16 consecutive output channels of one pixel (in place):
// v = act(v + bias [+ resid]) [* bn_scale + bn_shift]
__device__ __forceinline__ void epilogue_values16(const ConvArgs& a, float (&v)[16], size_t oidx, const float* bias,
                                                  const float* bsc, const float* bsh, bool has_bn,
                                                  const float* ishift = nullptr) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] += bias[i];
    if (a.resid != nullptr) {
        if (a.resid_bf16) {
            const uint4* rp = (const uint4*)((const unsigned short*)a.resid + oidx);
            const uint4 r0 = rp[0], r1 = rp[1];
            const unsigned rr[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                v[2 * q] += __uint_as_float(rr[q] << 16);
                v[2 * q + 1] += __uint_as_float(rr[q] & 0xffff0000u);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 rv = *(const float4*)((const float*)a.resid + oidx + q * 4);
                v[q * 4 + 0] += rv.x; v[q * 4 + 1] += rv.y; v[q * 4 + 2] += rv.z; v[q * 4 + 3] += rv.w;
            }
        }
    }
    if (a.act == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = fmaxf(v[i], 0.f);
    } else if (a.act == 2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = v[i] > 0.f ? v[i] : 0.2f * v[i];
    }
    if (has_bn) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = fmaf(v[i], bsc[i], bsh[i]);
    }
    if (ishift != nullptr) {                     // per-image vector (16 consecutive channels) after the BN affine
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 g = *(const float4*)(ishift + q * 4);
            v[q * 4 + 0] += g.x; v[q * 4 + 1] += g.y; v[q * 4 + 2] += g.z; v[q * 4 + 3] += g.w;
        }
    }
}

__device__ __forceinline__ void pack16_bf16(const float (&v)[16], uint4& p0, uint4& p1) {
    p0.x = pack_bf16x2(v[0], v[1]);   p0.y = pack_bf16x2(v[2], v[3]);
    p0.z = pack_bf16x2(v[4], v[5]);   p0.w = pack_bf16x2(v[6], v[7]);
    p1.x = pack_bf16x2(v[8], v[9]);   p1.y = pack_bf16x2(v[10], v[11]);
    p1.z = pack_bf16x2(v[12], v[13]); p1.w = pack_bf16x2(v[14], v[15]);
}

// epilogue_values16 + a direct store from the MFMA layout: 32 B (bf16) or 64 B (fp32) per lane.
template <bool OUT_BF16>
__device__ __forceinline__ void epilogue16(const ConvArgs& a, float (&v)[16], size_t oidx, const float* bias,
                                           const float* bsc, const float* bsh, bool has_bn,
                                           const float* ishift = nullptr) {
    epilogue_values16(a, v, oidx, bias, bsc, bsh, has_bn, ishift);
    if (!OUT_BF16 && a.out_parts > 0) {
        // fp32 island of an operand-split handle (conv1_1): the result enters the split stack as out_parts bf16 planes per pixel,
        // hi = rne(v), next = rne(v - hi), ... (each remainder exact in fp32); a pixel of the split tensor is [part][CoutPad]
        const int CoutPad = a.ncg * kCoutGroup, np = a.out_parts;
        const size_t co0 = oidx % (size_t)CoutPad;
        unsigned short* o = (unsigned short*)a.out + (oidx - co0) * np + co0;
        for (int p = 0; p < np; ++p) {
            uint4 p0, p1;
            if (a.split_f16) {                                 // IDC_FP16X3: fp16 parts (values clamped to the fp16 range first)
                unsigned w[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const _Float16 h0 = (_Float16)fminf(fmaxf(v[2 * e], -65504.f), 65504.f), h1 = (_Float16)fminf(fmaxf(v[2 * e + 1], -65504.f), 65504.f);
                    w[e] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
                    v[2 * e] -= (float)h0; v[2 * e + 1] -= (float)h1;
                }
                *(uint4*)(o + (size_t)p * CoutPad) = uint4{w[0], w[1], w[2], w[3]};
                *(uint4*)(o + (size_t)p * CoutPad + 8) = uint4{w[4], w[5], w[6], w[7]};
                continue;
            }
            pack16_bf16(v, p0, p1);
            *(uint4*)(o + (size_t)p * CoutPad) = p0;
            *(uint4*)(o + (size_t)p * CoutPad + 8) = p1;
            const unsigned w[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[2 * e] -= __uint_as_float(w[e] << 16); v[2 * e + 1] -= __uint_as_float(w[e] & 0xffff0000u); }
        }
    } else if (!OUT_BF16 || a.out_f32) {
        float* o = (float*)a.out + oidx;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *(float4*)(o + q * 4) = float4{v[q * 4 + 0], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3]};
    } else {
        unsigned short* o = (unsigned short*)a.out + oidx;
        uint4 p0, p1;
        pack16_bf16(v, p0, p1);
        *(uint4*)(o) = p0;
        *(uint4*)(o + 8) = p1;
    }
}

__device__ __forceinline__ void load16(float (&dst)[16], const float* src) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = *(const float4*)(src + q * 4);
        dst[q * 4 + 0] = v.x; dst[q * 4 + 1] = v.y; dst[q * 4 + 2] = v.z; dst[q * 4 + 3] = v.w;
    }
}


template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

}  // namespace idc
