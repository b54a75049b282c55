// idc_v2m.hip -- conv_igemm_v2m<WCO, WPX, HALO>: the bf16 throughput tile of conv_igemm_v2 (idc_kernels.hip) built from
// v_mfma_f32_16x16x32_bf16 instead of v_mfma_f32_32x32x16_bf16.
//
// Why a second MFMA shape: the forward runs at the package power cap, and the two instructions do not cost the same energy.  A pure
// MFMA loop on uniform random bf16 operands sustains 1.78 PFLOP/s with the 32x32x16 instruction and 2.02 with the 16x16x32 one
// (tools/ubench/mfma_peak, profiles/r03_mfma_peak_by_shape.txt: the K = 32 form updates an fp32 accumulator once per 32 MACs
// instead of once per 16); the kernel's own K-loop mix -- the same 24 fragment reads, barrier, LDS-DMA, chunk change and address
// update per 64 x 128 x 64 x 2 FLOP wave-step -- reaches 0.65 of nominal with it against 0.60 (tools/ubench/mix_probe,
// profiles/r03_mix_probe_mfma16.txt), although it needs more cycles per step (twice the MFMA instructions to issue).
//
// What is the same as conv_igemm_v2: workgroup = (32 x 4*WPX) sites x 64*WCO couts, wave = 64 couts x 128 sites (4 pixel rows of
// 32) on 128 accumulator registers, halo tile staged once per 128-byte channel chunk (register prefetch under the previous chunk's
// last tap, zero page for out-of-image rows), weight tiles by LDS-DMA into a 2-deep ring one tap ahead, one vmcnt(0) + barrier per
// tap, XCD-aware tile order, bf16-transposed stores of whole 128-byte lines, conv10_2's fused tanh head in the MFMA layout.
// What differs: the LAYOUT-1 weight image (idc_layout.h: slot ^ (row & 7), cg_row_to_cout row order -- the image conv_igemm and
// conv_click read), halo slots swizzled by row & 7; accumulators are 4 x 8 tiles of 16 x 16: lane (site r = lane & 15, group
// g = lane >> 4) register j of acc[mi][pt] is cout g*16 + mi*4 + j of the wave's 64 at site (pixel row pt >> 1, column
// (pt & 1)*16 + r); one k32 step reads 4 A + 8 B fragments (the same 12 KiB per 64 x 128 x 32 MACs as two k16 steps of the 32x32
// form) in two stages of 16 MFMAs.
// Scope: bf16-output launches without a shortcut sum, per-image shift or fp32 output (every conv_igemm_v2 launch of the N = 32
// bench forward); the engine keeps conv_igemm_v2 for the rest.  Same sums in a different order: results differ from
// conv_igemm_v2's in the last bf16 bit here and there, never between batch sizes (the variant is chosen per handle).
#include <stdlib.h>

#include <type_traits>

#include "idc_kernels.h"

#include "idc_layout.h"
#include "idc_split.hip.h"

namespace idc {

// in-kernel cycle stamps for the tuning harness (tools/ablate, -DIDC_TIMING): compiled out of the library
#ifdef IDC_TIMING
extern __device__ long long* g_idc_dbg;
#define IDC_MSTAMP(i) do { if (tid == 0) g_idc_dbg[(size_t)blockIdx.x * 16 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define IDC_MSTAMP(i) do {} while (0)
#endif



__device__ __forceinline__ int xcd_remap_m(int b, int nb) {
    const int xcd = b & 7, q = nb >> 3, r = nb & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (b >> 3);
}


// SPLIT = false: conv_igemm_v2m as described above.  SPLIT = true: conv_igemm_v2s, the operand-split form (IDC_BF16X3 / IDC_BF16X6) -- the same
// tile and K-loop body walked over a.nseg segments of nkc chunks (input part x weight part per segment, ConvArgs), and its own epilogue.
template <int WCO, int WPX, int HALO, int SPLIT>
__device__ __forceinline__ void conv_v2m_body(const ConvArgs& a) {
    constexpr int NT = WCO * WPX * 64;
    constexpr int TW = 32, TH = 4 * WPX;
    constexpr int HWP = TW + 2 * HALO, HHP = TH + 2 * HALO, HROWS = HWP * HHP;
    constexpr int BN = 64 * WCO;
    constexpr int W_BYTES = BN * kRowBytes;
    constexpr int N_HITEMS = (HROWS * kSlots + NT - 1) / NT;
    constexpr int HALO_BYTES = N_HITEMS * NT * kSlotBytes;
    constexpr int N_WITEMS = (W_BYTES / kSlotBytes) / NT;
    static_assert((W_BYTES / kSlotBytes) % NT == 0, "weight tile must split evenly");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const halo = smem;
    char* const wbuf = smem + HALO_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave % WCO, wpx = wave / WCO;
    const int r16 = lane & 15, g16 = lane >> 4;

    // tile order as conv_igemm_v2: (deconv phase, cout tile) fastest, contiguous ranges per XCD
    int b = xcd_remap_m(blockIdx.x, gridDim.x);
    const int nct = a.ncg / WCO;
    const int phase = b % a.nphase; b /= a.nphase;
    const int ct = b % nct; b /= nct;
    const int txi = b % a.tiles_x; b /= a.tiles_x;
    const int tyi = b % a.tiles_y;
    const int n = b / a.tiles_y;
    const int ty0 = tyi * TH, tx0 = txi * TW;
    const int Hs = a.Hs, Ws = a.Ws;
    const int ro = a.ro[phase], cof = a.co[phase];
    const int* const tap_dy = a.dy + phase * 9;
    const int* const tap_dx = a.dx + phase * 9;
    const int* const tap_tw = a.tw + phase * 9;
    const int nkc = a.nkc, ntaps = a.ntaps, si = a.si;
    const size_t w_kc_stride = (size_t)a.ncg * kWBlockBytes;
    const char* const wb = (const char*)a.wgt + (size_t)(ct * WCO) * kWBlockBytes + (size_t)tid * kSlotBytes;
    const int pix_chunks = SPLIT ? a.in_parts * nkc : nkc;     // 128-byte chunks per input pixel (split tensors: [part][chunk])
    const char* const img = (const char*)a.in + (size_t)n * (size_t)(Hs * si) * (Ws * si) * ((size_t)pix_chunks * kRowBytes);

    // accumulators start at the bias: lane (site r16, group g16) register j of acc[mi][.] is cout g16*16 + mi*4 + j of the wave's 64
    f32x4 acc[4][8];
    {
        const float* const bp = a.bias + (ct * WCO + wco) * kCoutGroup + g16 * 16;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const float4 bq = *(const float4*)(bp + mi * 4);
            const f32x4 b4 = SPLIT ? f32x4{0.f, 0.f, 0.f, 0.f} : f32x4{bq.x, bq.y, bq.z, bq.w};     // (SPLIT: the bias joins after the K loop, see idc_layout.h)
#pragma unroll
            for (int pt = 0; pt < 8; ++pt) acc[mi][pt] = b4;
        }
    }

    size_t wpart = 0;                          // SPLIT: byte offset of the current segment's weight part
    auto dma_w = [&](int tw, int kc, int buf) {
        const char* src = wb + ((size_t)tw * nkc + kc) * w_kc_stride + (SPLIT ? wpart : (size_t)0);
        char* dst = wbuf + buf * W_BYTES + wave * 64 * kSlotBytes;
#pragma unroll
        for (int j = 0; j < N_WITEMS; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)j * NT * kSlotBytes),
                                             (__attribute__((address_space(3))) void*)(dst + j * NT * kSlotBytes), 16, 0, 0);
    };
    u32x4 hreg[N_HITEMS];
    auto load_halo = [&](int kc) {
        const int Win = Ws * si, pix_bytes = pix_chunks * kRowBytes;
        int tid_ = tid;
        if constexpr (NT == 256 && HALO == 2) asm volatile("" : "+v"(tid_));   // as conv_igemm_v2: recompute the 14 item addresses per chunk
#pragma unroll
        for (int j = 0; j < N_HITEMS; ++j) {
            const int item = tid_ + j * NT;
            const int hr = item >> 3, sig = item & 7;
            const int hy = hr / HWP, hx = hr - hy * HWP;
            const int sy = ty0 - HALO + hy, sx = tx0 - HALO + hx;
            const bool inside = (unsigned)sy < (unsigned)Hs && (unsigned)sx < (unsigned)Ws && item < HROWS * kSlots;
            const int off = ((sy * si) * Win + sx * si) * pix_bytes + ((sig ^ swz(hr)) + kc * kSlots) * kSlotBytes;
            hreg[j] = *(const u32x4*)(inside ? img + off : (const char*)a.zeros);
        }
    };

    // SPLIT: the K loop below runs once per segment; xoff = first chunk of the segment's input part inside a pixel
    int seg = 0, xoff = 0;
    if constexpr (SPLIT) { xoff = (int)(a.seg_x & 15u) * nkc; wpart = (size_t)(a.seg_w & 15u) * (size_t)a.w_part_bytes; }
    load_halo(xoff);
    dma_w(tap_tw[0], 0, 0);

    const int wrow16 = (wco * 64 + r16) * kRowBytes;           // + mi*16 rows; swz(row) = r16 & 7 for all of them
    const int wslot16 = (g16 ^ swz(r16)) * kSlotBytes;          // slot kk*4 + g16: kk*4 flips bit 2 only
    int buf = 0;
    // LDS byte address of the lane's B row (first 16 sites) for each of the wave's 4 pixel rows; sites 16..31 of the row are 16 rows
    // = 2 KiB further with the same swizzle term
    int xb[4];
    auto set_xb = [&](int dy, int dx) {
#pragma unroll
        for (int pj = 0; pj < 4; ++pj) {
            const int xr = (wpx * 4 + pj + HALO + dy) * HWP + (r16 + HALO + dx);
            xb[pj] = xr * kRowBytes + ((g16 ^ swz(xr)) * kSlotBytes);
        }
    };
    set_xb(tap_dy[0], tap_dx[0]);
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);              // second-dispatched half of an 8-wave workgroup (as conv_igemm_v2)
    int tw_dma = ntaps > 1 ? tap_tw[1] : tap_tw[0];

    const int nseg = SPLIT ? a.nseg : 1;
    for (;;) {
    for (int kc = 0; kc < nkc; ++kc) {
        __syncthreads();                       // previous chunk's halo reads are done
#pragma unroll
        for (int j = 0; j < N_HITEMS; ++j) *(u32x4*)(halo + (tid + j * NT) * kSlotBytes) = hreg[j];
        bool last_kc = kc + 1 == nkc;
        // SPLIT: the chunk after a segment's last one is the next segment's first (its input part, its weight part)
        int kc_next = kc + 1, xoff_next = xoff;
        size_t wpart_next = wpart;
        if constexpr (SPLIT) {
            if (last_kc && seg + 1 < nseg) {
                last_kc = false; kc_next = 0;
                xoff_next = (int)((a.seg_x >> (4 * (seg + 1))) & 15u) * nkc;
                wpart_next = (size_t)((a.seg_w >> (4 * (seg + 1))) & 15u) * (size_t)a.w_part_bytes;
            }
        }
        auto tap_body = [&](int t, auto last_tag) {
            constexpr bool LAST = decltype(last_tag)::value;
            const char* const wcur = wbuf + buf * W_BYTES;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // my pieces of this tap's weight tile landed
            __syncthreads();                                        // everybody's landed; everybody left the other buffer
            const int xaddr[4] = {xb[0], xb[1], xb[2], xb[3]};
            u32x4 wf[4], xlo[4], xhi[4];
            auto read_a1 = [&](int kk, int mi) {
                wf[mi] = *(const u32x4*)(wcur + ((wrow16 + mi * 16 * kRowBytes + wslot16) ^ (kk * 4 * kSlotBytes)));
            };
            auto read_b = [&](int kk, int half, u32x4 (&xf)[4]) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    xf[q] = *(const u32x4*)(halo + (xaddr[half * 2 + (q >> 1)] ^ (kk * 4 * kSlotBytes)) + (q & 1) * 16 * kRowBytes);
            };
            auto mma4 = [&](int mi, int half, const u32x4 (&xf)[4]) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[mi][half * 4 + q] = mma_16x16x32<SPLIT == 2>(wf[mi], xf[q], acc[mi][half * 4 + q]);
            };
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) read_a1(0, mi);
            read_b(0, 0, xlo);
            __builtin_amdgcn_sched_barrier(0);
            // the NEXT step's loads behind this tap's first fragment reads (as conv_igemm_v2)
            if constexpr (!LAST) {
                dma_w(tw_dma, kc, buf ^ 1);
            } else if (!last_kc) {
                if constexpr (SPLIT) wpart = wpart_next;
                dma_w(tw_dma, kc_next, buf ^ 1);
                load_halo(xoff_next + kc_next);
            }
            __builtin_amdgcn_sched_barrier(0);
            // stage (k32 step 0, pixel rows 0-1): 16 MFMAs over the reads of rows 2-3
            read_b(0, 1, xhi);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) mma4(mi, 0, xlo);
#pragma unroll
            for (int q_ = 0; q_ < 4; ++q_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            // stage (0, rows 2-3): cout block by cout block, so that a block's A registers are free for step 1's fragment as soon
            // as its four MFMAs have issued; rows 0-1 of step 1 first
            read_b(1, 0, xlo);
            mma4(0, 1, xhi); read_a1(1, 0);
            mma4(1, 1, xhi); read_a1(1, 1);
            mma4(2, 1, xhi); read_a1(1, 2);
            mma4(3, 1, xhi); read_a1(1, 3);
#pragma unroll
            for (int q_ = 0; q_ < 4; ++q_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int q_ = 0; q_ < 3; ++q_) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            // stage (1, rows 0-1)
            read_b(1, 1, xhi);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) mma4(mi, 0, xlo);
#pragma unroll
            for (int q_ = 0; q_ < 4; ++q_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            // stage (1, rows 2-3): the next tap's addresses and weight-tile index under it
            {
                const int tn = LAST ? 0 : t + 1;
                set_xb(tap_dy[tn], tap_dx[tn]);
                tw_dma = tn + 1 < ntaps ? tap_tw[tn + 1] : tap_tw[0];
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) mma4(mi, 1, xhi);
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            buf ^= 1;
        };
        for (int t = 0; t + 1 < ntaps; ++t) tap_body(t, std::false_type{});
        tap_body(ntaps - 1, std::true_type{});
        if constexpr (SPLIT) xoff = xoff_next;
    }
        if (!SPLIT || ++seg >= nseg) break;
    }

    // ---- epilogue: lane (site r16, group g16) owns couts g16*16 + mi*4 + j of its wave's 64 -------------------------------------
    if constexpr (SPLIT) add_bias_after_k(a.bias + (ct * WCO + wco) * kCoutGroup + g16 * 16, acc, a.acc_scale);
    const int CoutPad = a.ncg * kCoutGroup;
    const bool has_bn = a.bn_scale != nullptr;
    const int so = a.so, Wout = Ws * so, Hout = Hs * so;
    const int cow = (ct * WCO + wco) * kCoutGroup;
    __syncthreads();                                           // every wave left the halo / weight tiles
    if (WCO == 2 && a.head_w != nullptr) {
        // conv10_2 -> model_out (model.py:101-109): activation and the 128 -> 2 dot product in the MFMA layout (16 couts of one
        // site per lane); the eight partial sums of a pixel (4 lane groups x 2 cout waves) meet in LDS.  Nothing stored but the ab map.
        f32x4 w0[4], w1[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const float4 u = *(const float4*)(a.head_w + cow + g16 * 16 + mi * 4);
            const float4 v = *(const float4*)(a.head_w + 128 + cow + g16 * 16 + mi * 4);
            w0[mi] = f32x4{u.x, u.y, u.z, u.w};
            w1[mi] = f32x4{v.x, v.y, v.z, v.w};
        }
        float* const hp = (float*)smem;                        // [wave][pt 8][group 4][16 sites][2]
        auto partial = [&](auto act_c) __attribute__((always_inline)) {        // (activation chosen once, as in conv_igemm_v2p)
            constexpr int ACT = decltype(act_c)::value;
#pragma unroll
            for (int pt = 0; pt < 8; ++pt) {
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float v = acc[mi][pt][j];
                        if constexpr (ACT == 1) v = fmaxf(v, 0.f);
                        else if constexpr (ACT == 2) v = fmaxf(v, 0.2f * v);
                        s0 = fmaf(v, w0[mi][j], s0);
                        s1 = fmaf(v, w1[mi][j], s1);
                    }
                *(float2*)(hp + ((((wave * 8 + pt) * 4 + g16) * 16 + r16) * 2)) = float2{s0, s1};
            }
        };
        if (a.act == 1) partial(std::integral_constant<int, 1>{});
        else if (a.act == 2) partial(std::integral_constant<int, 2>{});
        else partial(std::integral_constant<int, 0>{});
        __syncthreads();
        if (wco == 0) {                                        // waves wave, wave + 1 hold the two cout halves of these pixels
            const int px = lane & 31, ch = lane >> 5;
            const float hb = a.head_b[ch];
#pragma unroll
            for (int pj = 0; pj < 4; ++pj) {
                float p = hb;
#pragma unroll
                for (int w2 = 0; w2 < 2; ++w2)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        p += hp[(((((wave + w2) * 8 + pj * 2 + (px >> 4)) * 4 + g) * 16 + (px & 15)) * 2) + ch];
                const int sy = ty0 + wpx * 4 + pj, sx = tx0 + px;
                if (sy < Hs && sx < Ws) a.head_out[(((size_t)n * 2 + ch) * Hs + sy) * Ws + sx] = tanhf(p) * a.head_mul;
            }
        }
        return;
    }
    if constexpr (SPLIT) {
        split_epilogue<WCO, SPLIT == 2>(a, acc, smem, n, ty0, tx0, wpx, cow, ro, cof);
        return;
    }
    // bf16 outputs, activation (+ eval-BN) and rounding in the MFMA layout, then a wave-private [32 sites][64 couts] bf16 tile
    // (128-byte rows, slot ^ (site & 7)) read back as lane = (site l >> 3, 8 couts l & 7): every store covers whole 128-byte lines
    char* const tb16 = smem + wave * 4096;
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    const int rr = lane >> 3, cc = lane & 7;
    const int co8 = cow + cc * 8;
    f32x4 bsc[4], bsh[4];
    if (has_bn) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const float4 s4 = *(const float4*)(a.bn_scale + cow + g16 * 16 + mi * 4);
            const float4 t4 = *(const float4*)(a.bn_shift + cow + g16 * 16 + mi * 4);
            bsc[mi] = f32x4{s4.x, s4.y, s4.z, s4.w};
            bsh[mi] = f32x4{t4.x, t4.y, t4.z, t4.w};
            if (a.img_shift != nullptr) {                      // Global Hints: the per-image vector added after the BN affine (conv4_3; launches with BN only)
                const float4 u4 = *(const float4*)(a.img_shift + (size_t)n * CoutPad + cow + g16 * 16 + mi * 4);
                bsh[mi] += f32x4{u4.x, u4.y, u4.z, u4.w};
            }
        }
    }
    // (one body per (BN, ReLU) combination chosen once, a row's four lines read before the first bounds check: conv_igemm_v2p's epilogue says why)
    auto rows = [&](auto bn_c, auto relu_c) __attribute__((always_inline)) {
        constexpr bool BN = decltype(bn_c)::value, RELU = decltype(relu_c)::value;
#pragma unroll
        for (int pj = 0; pj < 4; ++pj) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int pt = pj * 2 + hf, site = hf * 16 + r16;
                unsigned pk[8];
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        float v0 = acc[mi][pt][2 * e], v1 = acc[mi][pt][2 * e + 1];
                        if constexpr (BN) {
                            if constexpr (RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                            pk[mi * 2 + e] = pack_bf16x2_m(fmaf(v0, bsc[mi][2 * e], bsh[mi][2 * e]), fmaf(v1, bsc[mi][2 * e + 1], bsh[mi][2 * e + 1]));
                        } else {
                            unsigned p = pack_bf16x2_m(v0, v1);
                            if constexpr (RELU) p = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p), s16x2{0, 0}));
                            pk[mi * 2 + e] = p;
                        }
                    }
                const int s0 = g16 * 2;                         // the lane's 16 couts = slots 2g, 2g+1 of the site's 128-byte row
                *(uint4*)(tb16 + site * 128 + ((s0 ^ (site & 7)) * 16)) = uint4{pk[0], pk[1], pk[2], pk[3]};
                *(uint4*)(tb16 + site * 128 + (((s0 + 1) ^ (site & 7)) * 16)) = uint4{pk[4], pk[5], pk[6], pk[7]};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // same-wave LDS ops are in order: the row tile is complete
            const int sy = ty0 + wpx * 4 + pj;
            auto line = [&](int i) { const int row = i * 8 + rr; return *(const uint4*)(tb16 + row * 128 + ((cc ^ (row & 7)) * 16)); };
            const uint4 o0 = line(0), o1 = line(1), o2 = line(2), o3 = line(3);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // reads retired before the tile is rewritten
            auto put = [&](int i, const uint4& o) {
                const int sx = tx0 + i * 8 + rr;
                if (sy < Hs && sx < Ws) {
                    const size_t oidx = (((size_t)n * Hout + (sy * so + ro)) * Wout + (sx * so + cof)) * CoutPad + co8;
                    *(uint4*)((unsigned short*)a.out + oidx) = o;
                }
            };
            put(0, o0); put(1, o1); put(2, o2); put(3, o3);
        }
    };
    if (has_bn) {
        if (a.act == 1) rows(std::true_type{}, std::true_type{}); else rows(std::true_type{}, std::false_type{});
    } else {
        if (a.act == 1) rows(std::false_type{}, std::true_type{}); else rows(std::false_type{}, std::false_type{});
    }
}

template <int WCO, int WPX, int HALO>
__global__ __launch_bounds__(WCO* WPX * 64, 2) void conv_igemm_v2m(const ConvArgs a) { conv_v2m_body<WCO, WPX, HALO, 0>(a); }
template <int WCO, int WPX, int HALO>
__global__ __launch_bounds__(WCO* WPX * 64, 2) void conv_igemm_v2s(const ConvArgs a) { conv_v2m_body<WCO, WPX, HALO, 1>(a); }
template <int WCO, int WPX, int HALO>      // IDC_FP16X3: the same walk on fp16 parts (v_mfma_f32_16x16x32_f16)
__global__ __launch_bounds__(WCO* WPX * 64, 2) void conv_igemm_v2sh(const ConvArgs a) { conv_v2m_body<WCO, WPX, HALO, 2>(a); }

// ================================================================================================
// conv_igemm_v2p<WCO, WPX, D> -- conv_igemm_v2m for the 3x3 convolutions (dilation D = 1 | 2) with NO address arithmetic in the K loop
// (the per-tap address update and re-swizzle priced at 3 % of the loop at the power cap, profiles/r03_mix_probe_mfma16.txt).
//   * halo rows (128 bytes of channels) sit at a 144-BYTE PITCH instead of being XOR-swizzled: 144 = 9 x 16 with 9 odd, so the 16
//     consecutive rows of a B fragment fall on 16 different 16-byte bank groups whatever row they start at -- conflict-free like the
//     swizzled tile, but the address of (tap, pixel row, 16-site half, k32 half) is ONE lane base plus a compile-time offset
//     (ds_read_b128's immediate field): the nine taps are unrolled and set_xb / xaddr ^ kk disappear;
//   * halo rows come through buffer loads: 32-bit offsets, the hardware bounds check returns zeros for out-of-image rows (offset
//     2^31) -- no zero page select, no 64-bit address pairs, no branch per row;
//   * A fragments: two lane constants (k32 halves) plus the ring slot's base.
// Everything else is conv_igemm_v2m: tile, LDS-DMA weight ring (2 slots, one tap ahead), stage plan, epilogues.  LDS: +12.5 % for the
// halo tile (<2,2>, D = 1: 79.8 KiB -- still two workgroups per CU).
// ================================================================================================

// SPLIT = true: conv_igemm_v2ps, the operand-split form (as conv_igemm_v2s is conv_igemm_v2m's): a.nseg passes over the nkc chunks, split epilogue.
// MODE: 0 = conv_igemm_v2p (bf16), 1 / 2 = the operand-split forms on bf16 / fp16 parts, 4 = conv_igemm_v2ph: MODE 0's body on fp16 operands (IDC_FP16's
// fast path: v_mfma_f32_16x16x32_f16, fp16 stores clamped to the fp16 range; everything else textually MODE 0)
template <int WCO, int WPX, int D, int MODE>
__device__ __forceinline__ void conv_v2p_body(const ConvArgs& a) {
    constexpr int SPLIT = MODE == 4 ? 0 : MODE;
    constexpr bool F16 = MODE == 2 || MODE == 4;
    constexpr int NT = WCO * WPX * 64;
    constexpr int TW = 32, TH = 4 * WPX, HALO = D;
    constexpr int HWP = TW + 2 * HALO, HHP = TH + 2 * HALO, HROWS = HWP * HHP, HP = kRowBytes;
    constexpr int BN = 64 * WCO;
    constexpr int W_BYTES = BN * kRowBytes;
    constexpr int N_HITEMS = (HROWS * kSlots + NT - 1) / NT;
    constexpr int HALO_BYTES = N_HITEMS * NT * kSlotBytes;
    constexpr int N_WITEMS = (W_BYTES / kSlotBytes) / NT;
    static_assert((W_BYTES / kSlotBytes) % NT == 0, "weight tile must split evenly");
    static_assert(NT % 8 == 0, "halo items: 8 slots per row");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const halo = smem;
    char* const wbuf = smem + HALO_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave % WCO, wpx = wave / WCO;
    const int r16 = lane & 15, g16 = lane >> 4;

    int b = xcd_remap_m(blockIdx.x, gridDim.x);
    const int nct = a.ncg / WCO;
    const int ct = b % nct; b /= nct;
    const int txi = b % a.tiles_x; b /= a.tiles_x;
    const int tyi = b % a.tiles_y;
    const int n = b / a.tiles_y;
    const int ty0 = tyi * TH, tx0 = txi * TW;
    const int Hs = a.Hs, Ws = a.Ws;
    const int nkc = a.nkc, si = a.si;
    const size_t w_kc_stride = (size_t)a.ncg * kWBlockBytes;
    const size_t w_tap_stride = (size_t)nkc * w_kc_stride;
    const char* const wb = (const char*)a.wgt + (size_t)(ct * WCO) * kWBlockBytes + (size_t)tid * kSlotBytes;
    const int pix_bytes = (SPLIT ? a.in_parts * nkc : nkc) * kRowBytes, Win = Ws * si;    // split tensors: a pixel is [part][chunk] x 128 bytes
    const char* const img = (const char*)a.in + (size_t)n * (size_t)(Hs * si) * Win * (size_t)pix_bytes;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)img, 0, (Hs * si) * Win * pix_bytes, 0x00020000);

    IDC_MSTAMP(0);
    f32x4 acc[4][8];
    {
        const float* const bp = a.bias + (ct * WCO + wco) * kCoutGroup + g16 * 16;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const float4 bq = *(const float4*)(bp + mi * 4);
            const f32x4 b4 = SPLIT ? f32x4{0.f, 0.f, 0.f, 0.f} : f32x4{bq.x, bq.y, bq.z, bq.w};     // (SPLIT: the bias joins after the K loop, see idc_layout.h)
#pragma unroll
            for (int pt = 0; pt < 8; ++pt) acc[mi][pt] = b4;
        }
    }

    size_t wpart = 0;                                          // SPLIT: byte offset of the current segment's weight part
    auto dma_w = [&](int t, int kc, int slot_byte) {           // slot_byte: 0 | W_BYTES
        const char* src = wb + (size_t)t * w_tap_stride + (size_t)kc * w_kc_stride + (SPLIT ? wpart : (size_t)0);
        char* dst = wbuf + slot_byte + wave * 64 * kSlotBytes;
#pragma unroll
        for (int j = 0; j < N_WITEMS; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)j * NT * kSlotBytes),
                                             (__attribute__((address_space(3))) void*)(dst + j * NT * kSlotBytes), 16, 0, 0);
    };
    u32x4 hreg[N_HITEMS];
    auto load_halo = [&](int kc) {                             // row (tid >> 3) + j*NT/8 of the halo tile, 16-byte slot tid & 7 of chunk kc
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));
#pragma unroll
        for (int j = 0; j < N_HITEMS; ++j) {
            const int hr = (tid_ >> 3) + j * (NT / 8), sig = tid_ & 7;
            const int hy = hr / HWP, hx = hr - hy * HWP;
            const int sy = ty0 - HALO + hy, sx = tx0 - HALO + hx;
            const bool inside = (unsigned)sy < (unsigned)Hs && (unsigned)sx < (unsigned)Ws && hr < HROWS;
            const int off = ((sy * si) * Win + sx * si) * pix_bytes + ((sig ^ (hx & 7)) + kc * kSlots) * kSlotBytes;   // LDS slot sig holds logical slot sig ^ (hx & 7)
            hreg[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, inside ? off : (int)0x80000000, 0, 0));
        }
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int j = 0; j < N_HITEMS; ++j) *(u32x4*)(halo + (tid + j * NT) * kSlotBytes) = hreg[j];
    };

    int seg = 0, xoff = 0;                                     // SPLIT: current segment, first chunk of its input part inside a pixel
    if constexpr (SPLIT) { xoff = (int)(a.seg_x & 15u) * nkc; wpart = (size_t)(a.seg_w & 15u) * (size_t)a.w_part_bytes; }
    load_halo(xoff);
    dma_w(0, 0, 0);
    // own code -> L2 (idc_kernels.h): 23-37 KB; the scratch is the tail of the halo area that no fragment read reaches
    if (a.warm && wave == 0) idc_warm_own_code(halo + HROWS * HP, lane, (WCO == 2 ? 288 : 180));   // 36.9 / 23.1-23.4 KB; conv_igemm_v2m's kernels follow in this code object
    static_assert(HALO_BYTES - HROWS * HP >= 256, "scratch for the code warm-up");

    // lane bases: B rows of the wave's first pixel row for each column shift dx and k32 half (the swizzle term depends on the halo COLUMN
    // only, so pixel-row / kernel-row / 16-site offsets are whole rows: compile-time immediates), A rows of the two k32 halves
    int xb[3][2];
#pragma unroll
    for (int dxi = 0; dxi < 3; ++dxi)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int hx = r16 + HALO + (dxi - 1) * D;
            xb[dxi][kk] = ((wpx * 4) * HWP + hx) * HP + (((kk * 4 + g16) ^ (hx & 7)) * kSlotBytes);
        }
    const int wa0 = (wco * 64 + r16) * kRowBytes + ((g16 ^ swz(r16)) * kSlotBytes);
    const int wa1 = wa0 ^ (4 * kSlotBytes);
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);              // second-dispatched half of an 8-wave workgroup (as conv_igemm_v2)
    int buf_off = 0;                                           // byte offset of the ring slot holding the current tap's tile

    IDC_MSTAMP(1);
    const int nseg = SPLIT ? a.nseg : 1;
    for (;;) {
    for (int kc = 0; kc < nkc; ++kc) {
        __syncthreads();                                       // previous chunk's halo reads are done
        store_halo();
        bool last_kc = kc + 1 == nkc;
        int kc_next = kc + 1, xoff_next = xoff;                // SPLIT: after a segment's last chunk comes the next segment's first
        size_t wpart_next = wpart;
        if constexpr (SPLIT) {
            if (last_kc && seg + 1 < nseg) {
                last_kc = false; kc_next = 0;
                xoff_next = (int)((a.seg_x >> (4 * (seg + 1))) & 15u) * nkc;
                wpart_next = (size_t)((a.seg_w >> (4 * (seg + 1))) & 15u) * (size_t)a.w_part_bytes;
            }
        }
        auto tap_body = [&](auto t_tag) {
            constexpr int t = decltype(t_tag)::value;
            constexpr bool LAST = t == 8;
            constexpr int dy = (t / 3 - 1) * D;
            constexpr int tn = LAST ? 0 : t + 1;
            const char* const wcur = wbuf + buf_off;
            if (t == 4 && kc == 0) IDC_MSTAMP(9);                   // (tools/ablate: where a tap's time goes)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // my pieces of this tap's weight tile landed
            if (t == 4 && kc == 0) IDC_MSTAMP(10);
            __syncthreads();                                        // everybody's landed; everybody left the other buffer
            if (t == 4 && kc == 0) IDC_MSTAMP(11);
            if (t == 5 && kc == 0) IDC_MSTAMP(12);
            const char* const a0 = wcur + wa0;
            const char* const a1 = wcur + wa1;
            u32x4 wf[4], xlo[4], xhi[4];
            auto read_b = [&](int kk, int half, u32x4 (&xf)[4]) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    xf[q] = *(const u32x4*)(halo + xb[t % 3][kk] + ((half * 2 + (q >> 1) + HALO + dy) * HWP + (q & 1) * 16) * HP);
            };
            auto mma4 = [&](int mi, int half, const u32x4 (&xf)[4]) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[mi][half * 4 + q] = mma_16x16x32<F16>(wf[mi], xf[q], acc[mi][half * 4 + q]);
            };
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) wf[mi] = *(const u32x4*)(a0 + mi * 16 * kRowBytes);
            read_b(0, 0, xlo);
            __builtin_amdgcn_sched_barrier(0);
            // the NEXT step's loads behind this tap's first fragment reads (as conv_igemm_v2m)
            if constexpr (!LAST) {
                dma_w(tn, kc, buf_off ^ W_BYTES);
            } else {
                if constexpr (SPLIT) wpart = wpart_next;
                if (!last_kc) dma_w(0, kc_next, buf_off ^ W_BYTES);
                load_halo(last_kc ? xoff + kc : xoff_next + kc_next);   // (unconditional: every halo register has one definition per trip -- a value
            }                                                  //  that might survive "in case" would stay live through all nine taps)
            __builtin_amdgcn_sched_barrier(0);
            // stage (k32 step 0, pixel rows 0-1): 16 MFMAs over the reads of rows 2-3
            read_b(0, 1, xhi);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) mma4(mi, 0, xlo);
#pragma unroll
            for (int q_ = 0; q_ < 4; ++q_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            // stage (0, rows 2-3): cout block by cout block, its A registers reloaded for step 1 behind its last four MFMAs
            read_b(1, 0, xlo);
            mma4(0, 1, xhi); wf[0] = *(const u32x4*)(a1);
            mma4(1, 1, xhi); wf[1] = *(const u32x4*)(a1 + 16 * kRowBytes);
            mma4(2, 1, xhi); wf[2] = *(const u32x4*)(a1 + 32 * kRowBytes);
            mma4(3, 1, xhi); wf[3] = *(const u32x4*)(a1 + 48 * kRowBytes);
#pragma unroll
            for (int q_ = 0; q_ < 4; ++q_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int q_ = 0; q_ < 3; ++q_) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            // stage (1, rows 0-1)
            read_b(1, 1, xhi);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) mma4(mi, 0, xlo);
#pragma unroll
            for (int q_ = 0; q_ < 4; ++q_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            // stage (1, rows 2-3)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) mma4(mi, 1, xhi);
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            buf_off ^= W_BYTES;
        };
        tap_body(std::integral_constant<int, 0>{}); tap_body(std::integral_constant<int, 1>{}); tap_body(std::integral_constant<int, 2>{});
        tap_body(std::integral_constant<int, 3>{}); tap_body(std::integral_constant<int, 4>{}); tap_body(std::integral_constant<int, 5>{});
        tap_body(std::integral_constant<int, 6>{}); tap_body(std::integral_constant<int, 7>{}); tap_body(std::integral_constant<int, 8>{});
        if constexpr (SPLIT) xoff = xoff_next;
    }
        if (!SPLIT || ++seg >= nseg) break;
    }

    // ---- epilogue (conv_igemm_v2m's): lane (site r16, group g16) owns couts g16*16 + mi*4 + j of its wave's 64 ----------------------
    IDC_MSTAMP(2);
    if constexpr (SPLIT) add_bias_after_k(a.bias + (ct * WCO + wco) * kCoutGroup + g16 * 16, acc, a.acc_scale);
    const int CoutPad = a.ncg * kCoutGroup;
    const bool has_bn = a.bn_scale != nullptr;
    const int cow = (ct * WCO + wco) * kCoutGroup;
    __syncthreads();                                           // every wave left the halo / weight tiles
    IDC_MSTAMP(5);
    if (WCO == 2 && a.head_w != nullptr) {
        f32x4 w0[4], w1[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const float4 u = *(const float4*)(a.head_w + cow + g16 * 16 + mi * 4);
            const float4 v = *(const float4*)(a.head_w + 128 + cow + g16 * 16 + mi * 4);
            w0[mi] = f32x4{u.x, u.y, u.z, u.w};
            w1[mi] = f32x4{v.x, v.y, v.z, v.w};
        }
        float* const hp = (float*)smem;                        // [wave][pt 8][group 4][16 sites][2]
        // (the activation is chosen ONCE: as run-time `if`s per element it cost two uniform branches per accumulator register, 256 per wave)
        auto partial = [&](auto act_c) __attribute__((always_inline)) {
            constexpr int ACT = decltype(act_c)::value;
#pragma unroll
            for (int pt = 0; pt < 8; ++pt) {
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float v = acc[mi][pt][j];
                        if constexpr (ACT == 1) v = fmaxf(v, 0.f);
                        else if constexpr (ACT == 2) v = fmaxf(v, 0.2f * v);
                        s0 = fmaf(v, w0[mi][j], s0);
                        s1 = fmaf(v, w1[mi][j], s1);
                    }
                *(float2*)(hp + ((((wave * 8 + pt) * 4 + g16) * 16 + r16) * 2)) = float2{s0, s1};
            }
        };
        if (a.act == 1) partial(std::integral_constant<int, 1>{});
        else if (a.act == 2) partial(std::integral_constant<int, 2>{});
        else partial(std::integral_constant<int, 0>{});
        IDC_MSTAMP(6);
        __syncthreads();
        IDC_MSTAMP(7);
        if (wco == 0) {
            const int px = lane & 31, ch = lane >> 5;
            const float hb = a.head_b[ch];
#pragma unroll
            for (int pj = 0; pj < 4; ++pj) {
                float p = hb;
#pragma unroll
                for (int w2 = 0; w2 < 2; ++w2)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        p += hp[(((((wave + w2) * 8 + pj * 2 + (px >> 4)) * 4 + g) * 16 + (px & 15)) * 2) + ch];
                const int sy = ty0 + wpx * 4 + pj, sx = tx0 + px;
                if (sy < Hs && sx < Ws) a.head_out[(((size_t)n * 2 + ch) * Hs + sy) * Ws + sx] = tanhf(p) * a.head_mul;
            }
        }
        IDC_MSTAMP(3);
#ifdef IDC_TIMING
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        IDC_MSTAMP(4);
#endif
        return;
    }
    if constexpr (SPLIT) {
        split_epilogue<WCO, SPLIT == 2>(a, acc, smem, n, ty0, tx0, wpx, cow, 0, 0);
        return;
    }
    char* const tb16 = smem + wave * 4096;
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    const int rr = lane >> 3, cc = lane & 7;
    const int co8 = cow + cc * 8;
    f32x4 bsc[4], bsh[4];
    if (has_bn) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const float4 s4 = *(const float4*)(a.bn_scale + cow + g16 * 16 + mi * 4);
            const float4 t4 = *(const float4*)(a.bn_shift + cow + g16 * 16 + mi * 4);
            bsc[mi] = f32x4{s4.x, s4.y, s4.z, s4.w};
            bsh[mi] = f32x4{t4.x, t4.y, t4.z, t4.w};
            if (a.img_shift != nullptr) {                      // Global Hints: the per-image vector added after the BN affine (conv4_3; launches with BN only)
                const float4 u4 = *(const float4*)(a.img_shift + (size_t)n * CoutPad + cow + g16 * 16 + mi * 4);
                bsh[mi] += f32x4{u4.x, u4.y, u4.z, u4.w};
            }
        }
    }
    // One body per (BN, ReLU) combination, chosen ONCE: written as run-time `if`s inside the element loops the compiler kept a uniform branch per packed
    // pair (32 taken branches per pixel row) and sank each transposed LDS read under its store's bounds check (read - wait - store, four times in a
    // row): 2.6 k cycles per pixel row, 10.9 k per tile -- 17 % of a conv10_2-shaped tile (tools/ablate stamps, profiles/r05_v2p_tap_stamps.txt).
    unsigned short* const out00 = (unsigned short*)a.out + (((size_t)n * Hs + ty0 + wpx * 4) * Ws + tx0 + rr) * CoutPad + co8;   // lane's line of the wave's first pixel row
    auto rows = [&](auto bn_c, auto relu_c) __attribute__((always_inline)) {
        constexpr bool BN = decltype(bn_c)::value, RELU = decltype(relu_c)::value;
#pragma unroll
        for (int pj = 0; pj < 4; ++pj) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int pt = pj * 2 + hf, site = hf * 16 + r16;
                unsigned pk[8];
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        float v0 = acc[mi][pt][2 * e], v1 = acc[mi][pt][2 * e + 1];
                        if constexpr (BN) {
                            if constexpr (RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                            pk[mi * 2 + e] = pack16x2_m<F16>(fmaf(v0, bsc[mi][2 * e], bsh[mi][2 * e]), fmaf(v1, bsc[mi][2 * e + 1], bsh[mi][2 * e + 1]));
                        } else {
                            unsigned p = pack16x2_m<F16>(v0, v1);
                            if constexpr (RELU) p = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p), s16x2{0, 0}));
                            pk[mi * 2 + e] = p;
                        }
                    }
                const int s0 = g16 * 2;
                *(uint4*)(tb16 + site * 128 + ((s0 ^ (site & 7)) * 16)) = uint4{pk[0], pk[1], pk[2], pk[3]};
                *(uint4*)(tb16 + site * 128 + (((s0 + 1) ^ (site & 7)) * 16)) = uint4{pk[4], pk[5], pk[6], pk[7]};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int sy = ty0 + wpx * 4 + pj;
            auto line = [&](int i) { const int row = i * 8 + rr; return *(const uint4*)(tb16 + row * 128 + ((cc ^ (row & 7)) * 16)); };
            const uint4 o0 = line(0), o1 = line(1), o2 = line(2), o3 = line(3);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // all four lines read (and the tile free for the next row) before the first store's bounds check
            auto put = [&](int i, const uint4& o) {
                const int sx = tx0 + i * 8 + rr;
                if (sy < Hs && sx < Ws) *(uint4*)(out00 + (pj * Ws + i * 8) * CoutPad) = o;      // (one 64-bit base per lane, 32-bit strides)
            };
            put(0, o0); put(1, o1); put(2, o2); put(3, o3);
            if (pj == 0) IDC_MSTAMP(6);
            if (pj == 1) IDC_MSTAMP(7);
        }
    };
    if (has_bn) {
        if (a.act == 1) rows(std::true_type{}, std::true_type{}); else rows(std::true_type{}, std::false_type{});
    } else {
        if (a.act == 1) rows(std::false_type{}, std::true_type{}); else rows(std::false_type{}, std::false_type{});
    }
    IDC_MSTAMP(3);
#ifdef IDC_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    IDC_MSTAMP(4);
#endif
}


template <int WCO, int WPX, int D>
__global__ __launch_bounds__(WCO* WPX * 64, 2) void conv_igemm_v2p(const ConvArgs a) { conv_v2p_body<WCO, WPX, D, 0>(a); }
template <int WCO, int WPX, int D>      // IDC_FP16: conv_igemm_v2p on fp16 operands
__global__ __launch_bounds__(WCO* WPX * 64, 2) void conv_igemm_v2ph(const ConvArgs a) { conv_v2p_body<WCO, WPX, D, 4>(a); }
template <int WCO, int WPX, int D>
__global__ __launch_bounds__(WCO* WPX * 64, 2) void conv_igemm_v2ps(const ConvArgs a) { conv_v2p_body<WCO, WPX, D, 1>(a); }
template <int WCO, int WPX, int D>
__global__ __launch_bounds__(WCO* WPX * 64, 2) void conv_igemm_v2psh(const ConvArgs a) { conv_v2p_body<WCO, WPX, D, 2>(a); }

static constexpr size_t conv_v2p_lds_bytes_c(int wco, int wpx, int d) {            // = conv_igemm_v2m's
    const int nt = wco * wpx * 64;
    const int hrows = (32 + 2 * d) * (4 * wpx + 2 * d);
    const int items = (hrows * kSlots + nt - 1) / nt;
    return (size_t)items * nt * kSlotBytes + 2 * (size_t)(64 * wco) * kRowBytes;
}

#define IDC_FOR_EACH_CONV_V2P(X) X(4, 2, 1) X(4, 2, 2) X(2, 2, 1)
#define IDC_FOR_EACH_CONV_V2PS(X) IDC_FOR_EACH_CONV_V2P(X) X(1, 4, 1) X(1, 2, 1)      // (+ the 64-cout tiles of conv1_2)

// 3x3 convs (so = 1, one phase, nine taps in ky*3 + kx order with offsets (ky-1, kx-1) * D) that conv_igemm_v2m covers
bool conv_v2p_applies(ConvConfig cfg, int halo, const ConvArgs& a) {
    if (!conv_v2m_applies(a) || a.nphase != 1 || a.ntaps != 9 || a.so != 1 || (halo != 1 && halo != 2)) return false;
    if (!((cfg.wm == 4 && cfg.wp == 2) || (cfg.wm == 2 && cfg.wp == 2 && halo == 1))) return false;
    for (int t = 0; t < 9; ++t)
        if (a.dy[t] != (t / 3 - 1) * halo || a.dx[t] != (t % 3 - 1) * halo || a.tw[t] != t) return false;
    // buffer loads address one image with 32-bit offsets; out-of-image rows use offset 2^31
    return (long long)a.Hs * a.si * (long long)a.Ws * a.si * ((long long)a.nkc * kRowBytes) < 0x7fffffffLL;
}

hipError_t launch_conv_v2p(ConvConfig cfg, int halo, const ConvArgs& a, hipStream_t s) {
    if (!conv_v2p_applies(cfg, halo, a)) return hipErrorInvalidConfiguration;
    const int nct = a.ncg / cfg.wm;
    const long long blocks = (long long)a.tiles_x * a.tiles_y * a.N * nct;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
#define X(WCO, WPX, DD)                                                                                                          \
    if (cfg.wm == WCO && cfg.wp == WPX && halo == DD) {                                                                          \
        if (a.split_f16) hipLaunchKernelGGL((conv_igemm_v2ph<WCO, WPX, DD>), dim3((unsigned)blocks), dim3(WCO * WPX * 64),       \
                                            conv_v2p_lds_bytes_c(WCO, WPX, DD), s, a);          /* IDC_FP16's fast path */           \
        else hipLaunchKernelGGL((conv_igemm_v2p<WCO, WPX, DD>), dim3((unsigned)blocks), dim3(WCO * WPX * 64),                    \
                                conv_v2p_lds_bytes_c(WCO, WPX, DD), s, a);                                                       \
        return hipGetLastError();                                                                                                \
    }
    IDC_FOR_EACH_CONV_V2P(X)
#undef X
    return hipErrorInvalidConfiguration;
}

static constexpr size_t conv_v2m_lds_bytes_c(int wco, int wpx, int halo) {
    const int nt = wco * wpx * 64;
    const int hrows = (32 + 2 * halo) * (4 * wpx + 2 * halo);
    const int items = (hrows * kSlots + nt - 1) / nt;
    return (size_t)items * nt * kSlotBytes + 2 * (size_t)(64 * wco) * kRowBytes;
}

#define IDC_FOR_EACH_CONV_V2M(X) X(4, 2, 0) X(4, 2, 1) X(4, 2, 2) X(2, 4, 0) X(2, 4, 1) X(2, 4, 2) X(2, 2, 0) X(2, 2, 1) X(2, 2, 2)
#define IDC_FOR_EACH_CONV_V2S(X) IDC_FOR_EACH_CONV_V2M(X) X(1, 4, 1) X(1, 2, 1)

bool conv_v2m_applies(const ConvArgs& a) {
    return a.resid == nullptr && a.in2 == nullptr && !a.out_f32 && (a.img_shift == nullptr || a.bn_scale != nullptr) && a.pk_L == nullptr && a.ksplit <= 1 &&
           (a.act != 2 || a.head_w != nullptr) && (a.head_w == nullptr || a.bn_scale == nullptr) && a.zeros != nullptr;
}

hipError_t launch_conv_v2m(ConvConfig cfg, int halo, const ConvArgs& a, hipStream_t s) {
    if (!conv_v2m_applies(a)) return hipErrorInvalidConfiguration;
    const int nct = a.ncg / cfg.wm;
    const long long blocks = (long long)a.tiles_x * a.tiles_y * a.N * nct * a.nphase;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
#define X(WCO, WPX, HL)                                                                                                          \
    if (cfg.wm == WCO && cfg.wp == WPX && halo == HL) {                                                                          \
        hipLaunchKernelGGL((conv_igemm_v2m<WCO, WPX, HL>), dim3((unsigned)blocks), dim3(WCO * WPX * 64),                         \
                           conv_v2m_lds_bytes_c(WCO, WPX, HL), s, a);                                                            \
        return hipGetLastError();                                                                                                \
    }
    IDC_FOR_EACH_CONV_V2M(X)
#undef X
    return hipErrorInvalidConfiguration;
}

// operand-split launches (IDC_BF16X3 / IDC_BF16X6): everything conv_igemm_v2m's geometry covers, plus fp32 shortcut sums, fp32 outputs, per-image shifts
bool conv_v2s_applies(const ConvArgs& a) {
    if (a.in2 != nullptr || a.pk_L != nullptr || a.ksplit > 1 || a.zeros == nullptr || a.resid_bf16) return false;
    if (a.in_parts < 1 || a.in_parts > 3 || a.nseg < 1 || a.nseg > 6 || a.w_part_bytes == 0) return false;
    if (a.head_w != nullptr) return a.bn_scale == nullptr && a.resid == nullptr && a.img_shift == nullptr;
    return a.out_f32 ? a.out_parts == 0 : (a.out_parts >= 1 && a.out_parts <= 3);
}

hipError_t launch_conv_v2s(ConvConfig cfg, int halo, const ConvArgs& a, hipStream_t s) {
    if (!conv_v2s_applies(a)) return hipErrorInvalidConfiguration;
    const int nct = a.ncg / cfg.wm;
    const long long blocks = (long long)a.tiles_x * a.tiles_y * a.N * nct * a.nphase;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
#define X(WCO, WPX, HL)                                                                                                          \
    if (cfg.wm == WCO && cfg.wp == WPX && halo == HL) {                                                                          \
        if (a.split_f16) hipLaunchKernelGGL((conv_igemm_v2sh<WCO, WPX, HL>), dim3((unsigned)blocks), dim3(WCO * WPX * 64),       \
                                            conv_v2m_lds_bytes_c(WCO, WPX, HL), s, a);                                           \
        else hipLaunchKernelGGL((conv_igemm_v2s<WCO, WPX, HL>), dim3((unsigned)blocks), dim3(WCO * WPX * 64),                    \
                                conv_v2m_lds_bytes_c(WCO, WPX, HL), s, a);                                                       \
        return hipGetLastError();                                                                                                \
    }
    IDC_FOR_EACH_CONV_V2S(X)
#undef X
    return hipErrorInvalidConfiguration;
}

// conv_igemm_v2ps: the 3x3 launches among them on conv_igemm_v2p's body (column-swizzled halo tile, unrolled taps, buffer loads)
bool conv_v2ps_applies(ConvConfig cfg, int halo, const ConvArgs& a) {
    if (!conv_v2s_applies(a) || a.nphase != 1 || a.ntaps != 9 || a.so != 1 || (halo != 1 && halo != 2)) return false;
    if (!((cfg.wm == 4 && cfg.wp == 2) || (cfg.wm == 2 && cfg.wp == 2 && halo == 1) || (cfg.wm == 1 && (cfg.wp == 2 || cfg.wp == 4) && halo == 1))) return false;
    for (int t = 0; t < 9; ++t)
        if (a.dy[t] != (t / 3 - 1) * halo || a.dx[t] != (t % 3 - 1) * halo || a.tw[t] != t) return false;
    return (long long)a.Hs * a.si * (long long)a.Ws * a.si * ((long long)a.nkc * a.in_parts * kRowBytes) < 0x7fffffffLL;
}

hipError_t launch_conv_v2ps(ConvConfig cfg, int halo, const ConvArgs& a, hipStream_t s) {
    if (!conv_v2ps_applies(cfg, halo, a)) return hipErrorInvalidConfiguration;
    const int nct = a.ncg / cfg.wm;
    const long long blocks = (long long)a.tiles_x * a.tiles_y * a.N * nct;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
#define X(WCO, WPX, DD)                                                                                                          \
    if (cfg.wm == WCO && cfg.wp == WPX && halo == DD) {                                                                          \
        if (a.split_f16) hipLaunchKernelGGL((conv_igemm_v2psh<WCO, WPX, DD>), dim3((unsigned)blocks), dim3(WCO * WPX * 64),      \
                                            conv_v2p_lds_bytes_c(WCO, WPX, DD), s, a);                                           \
        else hipLaunchKernelGGL((conv_igemm_v2ps<WCO, WPX, DD>), dim3((unsigned)blocks), dim3(WCO * WPX * 64),                   \
                                conv_v2p_lds_bytes_c(WCO, WPX, DD), s, a);                                                       \
        return hipGetLastError();                                                                                                \
    }
    IDC_FOR_EACH_CONV_V2PS(X)
#undef X
    return hipErrorInvalidConfiguration;
}

hipError_t init_kernels_v2m() {
    hipError_t e;
#define X(WCO, WPX, DD)                                                                                                          \
    e = hipFuncSetAttribute((const void*)conv_igemm_v2ps<WCO, WPX, DD>, hipFuncAttributeMaxDynamicSharedMemorySize,              \
                            (int)conv_v2p_lds_bytes_c(WCO, WPX, DD));                                                            \
    if (e != hipSuccess) return e;                                                                                               \
    e = hipFuncSetAttribute((const void*)conv_igemm_v2psh<WCO, WPX, DD>, hipFuncAttributeMaxDynamicSharedMemorySize,             \
                            (int)conv_v2p_lds_bytes_c(WCO, WPX, DD));                                                            \
    if (e != hipSuccess) return e;
    IDC_FOR_EACH_CONV_V2PS(X)
#undef X
#define X(WCO, WPX, HL)                                                                                                          \
    e = hipFuncSetAttribute((const void*)conv_igemm_v2s<WCO, WPX, HL>, hipFuncAttributeMaxDynamicSharedMemorySize,               \
                            (int)conv_v2m_lds_bytes_c(WCO, WPX, HL));                                                            \
    if (e != hipSuccess) return e;                                                                                               \
    e = hipFuncSetAttribute((const void*)conv_igemm_v2sh<WCO, WPX, HL>, hipFuncAttributeMaxDynamicSharedMemorySize,              \
                            (int)conv_v2m_lds_bytes_c(WCO, WPX, HL));                                                            \
    if (e != hipSuccess) return e;
    IDC_FOR_EACH_CONV_V2S(X)
#undef X
#define X(WCO, WPX, DD)                                                                                                          \
    e = hipFuncSetAttribute((const void*)conv_igemm_v2p<WCO, WPX, DD>, hipFuncAttributeMaxDynamicSharedMemorySize,               \
                            (int)conv_v2p_lds_bytes_c(WCO, WPX, DD));                                                            \
    if (e != hipSuccess) return e;                                                                                               \
    e = hipFuncSetAttribute((const void*)conv_igemm_v2ph<WCO, WPX, DD>, hipFuncAttributeMaxDynamicSharedMemorySize,              \
                            (int)conv_v2p_lds_bytes_c(WCO, WPX, DD));                                                            \
    if (e != hipSuccess) return e;
    IDC_FOR_EACH_CONV_V2P(X)
#undef X
#define X(WCO, WPX, HL)                                                                                                          \
    e = hipFuncSetAttribute((const void*)conv_igemm_v2m<WCO, WPX, HL>, hipFuncAttributeMaxDynamicSharedMemorySize,               \
                            (int)conv_v2m_lds_bytes_c(WCO, WPX, HL));                                                            \
    if (e != hipSuccess) return e;
    IDC_FOR_EACH_CONV_V2M(X)
#undef X
    return hipSuccess;
}

}  // namespace idc
