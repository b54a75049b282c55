// idc_layout.h -- data layout shared by the host-side weight packer and the gfx950 kernels.
//
// Everything on the hot path moves in 16-byte slots and 128-byte rows:
//   * activations are NHWC; one pixel's channels are cut into 128-byte chunks
//     (64 bf16 or 32 fp32 channels = "KC" channels), i.e. one chunk of one pixel = one row;
//   * weights are packed per layer as [tap][cin-chunk][cout-group of 64][row 0..63][slot 0..7]:
//     one 8 KiB block is the exact LDS image a workgroup needs for (tap, chunk, 64 couts), so
//     staging it is a straight coalesced copy;
//   * inside a row the eight 16-byte slots are XOR-swizzled by swz(row) = row&7, so that the
//     16 consecutive rows one MFMA operand fragment touches land on 16 different 16-byte bank
//     groups of the 256-byte LDS bank row (conflict-free ds_read_b128);
//   * the 64 rows of a cout group are permuted (cg_row_to_cout) so that after the MFMA each lane
//     owns 16 CONSECUTIVE output channels of one pixel -> 32-byte (bf16) / 64-byte (fp32) stores
//     and one full 128-byte line per pixel per wave.
#pragma once
#include <stdint.h>

namespace idc {

constexpr int kRowBytes = 128;
constexpr int kSlotBytes = 16;
constexpr int kSlots = 8;
constexpr int kCoutGroup = 64;                               // MFMA rows owned by one wave
constexpr int kWBlockBytes = kCoutGroup * kRowBytes;         // 8192
constexpr int kMaxTaps = 9;
constexpr int kMaxPhases = 4;

// swizzle: physical slot = logical slot ^ swz(row)
__host__ __device__ inline int swz(int row) { return row & 7; }

// Row `lam` (0..63) of a cout-group block holds this output channel (relative to the group):
// fragment ci = lam>>4 (MFMA row block), r = lam&15 (MFMA row).  D-layout of the 16x16 MFMA puts
// row r = 4*g + reg in lane group g, register reg; choosing cout = g*16 + ci*4 + reg makes the
// 16 accumulators (ci, reg) of a lane consecutive channels.
__host__ __device__ inline int cg_row_to_cout(int lam) {
    const int ci = lam >> 4, r = lam & 15;
    return (r >> 2) * 16 + ci * 4 + (r & 3);
}

// ---- layout 2 (bf16 large-tile kernel, 32x32x16 MFMA) -------------------------------------------
// Same [tap][cin-chunk][cout group][64 rows][8 slots] blocks, but
//   * swizzle swz2(row) = (row>>1)&7: an operand fragment is 32 rows x 2 k-groups per ds_read_b128;
//   * row permutation for the 32x32 D layout (reg r of lane (col, hgrp) is row (r/4)*8 + hgrp*4 + r%4):
//     row mi*32 + (r>>2)*8 + hh*4 + (r&3) holds cout hh*32 + mi*16 + r, so a lane ends up with 16
//     consecutive couts per accumulator and 32 consecutive ones over its two accumulators.
__host__ __device__ inline int swz2(int row) { return (row >> 1) & 7; }
__host__ __device__ inline int cg_cout_to_row2(int col) {
    const int hh = col >> 5, mi = (col >> 4) & 1, r = col & 15;
    return mi * 32 + (r >> 2) * 8 + hh * 4 + (r & 3);
}

inline int elem_bytes(int precision) { return precision == 0 ? 4 : 2; }       // IDC_FP32 == 0; IDC_BF16 and the operand-split precisions store bf16
// operand-split precisions (IDC_BF16X3 = 2: x = hi + lo, three products; IDC_BF16X6 = 3: hi + mid + lo, six products)
inline bool is_split(int precision) { return precision >= 2 && precision <= 5; }
inline bool split_is_f16(int precision) { return precision == 4 || precision == 5; }   // IDC_FP16X3: bf16x3's planes and segments with fp16 (11-bit) parts
// IDC_FP16 (= 5, round 6): the operand-split machinery with ONE fp16 part and ONE segment -- plain fp16 operands (11 significant bits against bf16's 8),
// fp32 accumulation, per-layer power-of-two weight scale, conv1_1 exact fp32: the throughput tiles at 1 / 8 of the bf16 path's rounding error
// (measured and dropped: a fourth segment lo.lo -- N = 32 he-style 3.99e-3 against 3.86e-3 without it.  What that error was: fp16 lo parts of ~0.02
//  weights are SUBNORMAL (6e-8 absolute = 2^-18 of the weight); with the weight parts holding w * 2^s per layer -- LayerBlob::wscale_off, ConvArgs::acc_scale --
//  it is 1.9e-3, the fp32 arithmetic's own distance; oracle/emulate.py 'splitf2_fp32' / 'splitf2s_fp32' reproduce both figures on the CPU)
inline int split_parts(int precision) { return (precision == 2 || precision == 4) ? 2 : precision == 3 ? 3 : 1; }      // (IDC_FP16: 1)
// K segments (input part, weight part), 4 bits each, segment 0 in the low nibble.  SMALLEST PRODUCTS FIRST, hi.hi last, and the bias after the
// K loop: every v_mfma rounds its result to the accumulator's magnitude, so each of the 9 x nkc x 2 accumulations of a segment costs one rounding
// at the size the accumulator has at that moment.  hi.hi first made all 3 (6) segments round at full magnitude -- N = 32 he-style weights, bf16x6:
// 7.1e-3 on the ab map against the exact-fp32 kernels' 1.7e-3; smallest first leaves only the hi.hi pass rounding at full size.
inline int split_segments(int precision) { return (precision == 2 || precision == 4) ? 3 : precision == 3 ? 6 : 1; }      // (IDC_FP16: 1 = hi.hi)
inline unsigned split_seg_x(int precision) { return (precision == 2 || precision == 4) ? 0x001u : precision == 3 ? 0x001120u : 0u; }   // X3: lo, hi, hi      X6: hi, lo, mid, mid, hi, hi
inline unsigned split_seg_w(int precision) { return (precision == 2 || precision == 4) ? 0x010u : precision == 3 ? 0x010102u : 0u; }   // X3: hi, lo, hi      X6: lo, hi, mid, hi,  mid, hi
inline int kc_elems(int precision) { return kRowBytes / elem_bytes(precision); }  // 64 or 32

// fp32 -> bf16, round to nearest even (matches v_cvt_pk_bf16_f32 for finite values)
inline uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

}  // namespace idc
