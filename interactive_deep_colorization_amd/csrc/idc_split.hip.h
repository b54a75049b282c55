// idc_split.hip.h -- device helpers of the operand-split precisions (IDC_BF16X3 / IDC_BF16X6 / IDC_FP16X3) shared by the kernels that walk a K loop in
// segments (conv_igemm_v2s / conv_igemm_v2ps in idc_v2m.hip, conv_ds_fused_m's split form in idc_dsm.hip): the MFMA step on bf16 or fp16 parts, the
// conversions, the bias that joins after the K loop and the split epilogue (fp32 value -> 2 / 3 planes).
#pragma once
#include <type_traits>

#include "idc_kernels.h"
#include "idc_layout.h"

namespace idc {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_m;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ unsigned pack_bf16x2_m(float lo, float hi) {
    // one v_cvt_pk_bf16_f32 (RNE) as a VECTOR conversion: from `(__bf16)lo | (__bf16)hi << 16` the vectoriser pairs the conversions of NEIGHBOURING packs
    // and un-shuffles them with and / shift / two SDWA ors -- six instructions for two dwords instead of two (round 5: the epilogues are VALU-bound).
    // (Not inline asm: the hazard recogniser does not see an asm's reads of MFMA results, and the scheduler may move it next to the MFMAs.)
    typedef float f32x2_pk __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_pk __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_pk){lo, hi}, bf16x2_pk));
}

// One 16x16x32 MFMA step on two 16-byte fragments: bf16 (every precision but IDC_FP16X3) or fp16 operands, fp32 accumulate.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_m;
template <bool F16>
__device__ __forceinline__ f32x4 mma_16x16x32(const u32x4& a, const u32x4& b, const f32x4& c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_m, a), __builtin_bit_cast(f16x8_m, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_m, a), __builtin_bit_cast(bf16x8_m, b), c, 0, 0, 0);
}
// IDC_FP16X3: two fp16 values (RNE) in one dword; inputs are clamped to the fp16 range first (a value beyond +-65504 saturates instead of becoming inf)
__device__ __forceinline__ unsigned pack_f16x2_m(float lo, float hi) {
    typedef float f32x2_pk __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2_pk __attribute__((ext_vector_type(2)));
    const float a = __builtin_fminf(__builtin_fmaxf(lo, -65504.f), 65504.f), b = __builtin_fminf(__builtin_fmaxf(hi, -65504.f), 65504.f);
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_pk){a, b}, f16x2_pk));
}
// two fp32 -> one dword of 16-bit values: bf16 (RNE) or fp16 (RNE, clamped)
template <bool F16> __device__ __forceinline__ unsigned pack16x2_m(float lo, float hi) {
    if constexpr (F16) return pack_f16x2_m(lo, hi); else return pack_bf16x2_m(lo, hi);
}
__device__ __forceinline__ float f16_lo_to_f32(unsigned q) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(q & 0xffffu)); }
__device__ __forceinline__ float f16_hi_to_f32(unsigned q) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(q >> 16)); }

// acc = acc * sc + bias: sc = *acc_scale (2^-s of weights packed as w * 2^s, IDC_FP16X3) or 1 -- fma(x, 1, b) is x + b rounded once, the bf16 parts' sum
__device__ __forceinline__ void add_bias_after_k(const float* bp, f32x4 (&acc)[4][8], const float* acc_scale = nullptr) {
    const float sc = acc_scale != nullptr ? *acc_scale : 1.f;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const float4 bq = *(const float4*)(bp + mi * 4);
#pragma unroll
        for (int pt = 0; pt < 8; ++pt) {
            acc[mi][pt][0] = fmaf(acc[mi][pt][0], sc, bq.x); acc[mi][pt][1] = fmaf(acc[mi][pt][1], sc, bq.y);
            acc[mi][pt][2] = fmaf(acc[mi][pt][2], sc, bq.z); acc[mi][pt][3] = fmaf(acc[mi][pt][3], sc, bq.w);
        }
    }
}

// Epilogue of the operand-split kernels: lane (site r16, group g16) holds couts g16*16 + mi*4 + j of its wave's 64 at site (pixel row pt >> 1,
// column (pt & 1)*16 + r16) in acc[mi][pt][j].  value = BN(act(acc + fp32 shortcut sum)) + per-image shift, all fp32; then either an fp32 NHWC store
// straight from the MFMA layout (out_parts = 0) or out_parts bf16 planes hi = rne(v), next = rne(v - hi), ... (each remainder is exact in fp32), every
// plane through the wave-private [32 sites][64 couts] bf16 transpose tile so that stores cover whole 128-byte lines.
template <int WCO, bool F16 = false>
__device__ __forceinline__ void split_epilogue(const ConvArgs& a, f32x4 (&acc)[4][8], char* smem, int n, int ty0, int tx0, int wpx, int cow, int ro, int cof) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g16 = lane >> 4;
    const int Hs = a.Hs, Ws = a.Ws, so = a.so, Wout = Ws * so, Hout = Hs * so;
    const int CoutPad = a.ncg * kCoutGroup;
    const int np = a.out_parts;
    const float* const resid = (const float*)a.resid;
    const bool has_bn = a.bn_scale != nullptr, has_shift = a.img_shift != nullptr;
    f32x4 bsc[4], bsh[4], ish[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        bsc[mi] = f32x4{1.f, 1.f, 1.f, 1.f}; bsh[mi] = f32x4{0.f, 0.f, 0.f, 0.f}; ish[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (has_bn) {
            const float4 s4 = *(const float4*)(a.bn_scale + cow + g16 * 16 + mi * 4);
            const float4 t4 = *(const float4*)(a.bn_shift + cow + g16 * 16 + mi * 4);
            bsc[mi] = f32x4{s4.x, s4.y, s4.z, s4.w}; bsh[mi] = f32x4{t4.x, t4.y, t4.z, t4.w};
        }
        if (has_shift) {
            const float4 u4 = *(const float4*)(a.img_shift + (size_t)n * CoutPad + cow + g16 * 16 + mi * 4);
            ish[mi] = f32x4{u4.x, u4.y, u4.z, u4.w};
        }
    }
    char* const tb16 = smem + wave * 4096;
    const int rr = lane >> 3, cc = lane & 7;
    const int co8 = cow + cc * 8;
    auto rows = [&](auto act_c) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_c)::value;
#pragma unroll
        for (int pj = 0; pj < 4; ++pj) {
            const int sy = ty0 + wpx * 4 + pj;
            f32x4 v[2][4];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int pt = pj * 2 + hf, sx = tx0 + hf * 16 + r16;
                const bool inb = sy < Hs && sx < Ws;
                const size_t opix = ((size_t)n * Hout + (sy * so + ro)) * Wout + (sx * so + cof);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    f32x4 x = acc[mi][pt];
                    if (resid != nullptr && inb) {
                        const float4 q = *(const float4*)(resid + opix * CoutPad + cow + g16 * 16 + mi * 4);
                        x += f32x4{q.x, q.y, q.z, q.w};
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float e = x[j];
                        if constexpr (ACT == 1) e = fmaxf(e, 0.f);
                        else if constexpr (ACT == 2) e = fmaxf(e, 0.2f * e);
                        x[j] = fmaf(e, bsc[mi][j], bsh[mi][j]) + ish[mi][j];
                    }
                    v[hf][mi] = x;
                }
                if (np == 0 && inb) {
                    float* const op = (float*)a.out + opix * CoutPad + cow + g16 * 16;
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) *(float4*)(op + mi * 4) = float4{v[hf][mi][0], v[hf][mi][1], v[hf][mi][2], v[hf][mi][3]};
                }
            }
            for (int p = 0; p < np; ++p) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int site = hf * 16 + r16;
                    unsigned pk[8];
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            if constexpr (F16) {
                                const unsigned q = pack_f16x2_m(v[hf][mi][2 * e], v[hf][mi][2 * e + 1]);
                                pk[mi * 2 + e] = q;
                                v[hf][mi][2 * e] -= f16_lo_to_f32(q);                      // exact, as below (11-bit parts)
                                v[hf][mi][2 * e + 1] -= f16_hi_to_f32(q);
                            } else {
                            const unsigned q = pack_bf16x2_m(v[hf][mi][2 * e], v[hf][mi][2 * e + 1]);
                            pk[mi * 2 + e] = q;
                            v[hf][mi][2 * e] -= __uint_as_float(q << 16);              // exact: the remainder of a round-to-nearest fits fp32
                            v[hf][mi][2 * e + 1] -= __uint_as_float(q & 0xffff0000u);
                            }
                        }
                    const int s0 = g16 * 2;
                    *(uint4*)(tb16 + site * 128 + ((s0 ^ (site & 7)) * 16)) = uint4{pk[0], pk[1], pk[2], pk[3]};
                    *(uint4*)(tb16 + site * 128 + (((s0 + 1) ^ (site & 7)) * 16)) = uint4{pk[4], pk[5], pk[6], pk[7]};
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                auto line = [&](int i) { const int row = i * 8 + rr; return *(const uint4*)(tb16 + row * 128 + ((cc ^ (row & 7)) * 16)); };
                const uint4 o0 = line(0), o1 = line(1), o2 = line(2), o3 = line(3);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                auto put = [&](int i, const uint4& o) {
                    const int sx = tx0 + i * 8 + rr;
                    if (sy < Hs && sx < Ws) {
                        const size_t oidx = ((((size_t)n * Hout + (sy * so + ro)) * Wout + (sx * so + cof)) * np + p) * CoutPad + co8;
                        *(uint4*)((unsigned short*)a.out + oidx) = o;
                    }
                };
                put(0, o0); put(1, o1); put(2, o2); put(3, o3);
            }
        }
    };
    if (a.act == 1) rows(std::integral_constant<int, 1>{});
    else if (a.act == 2) rows(std::integral_constant<int, 2>{});
    else rows(std::integral_constant<int, 0>{});
}

}  // namespace idc
