// idc_dsm.hip -- conv_ds_fused_m: conv_ds_fused (idc_kernels.hip: ConvTranspose 4x4 s2 + the 3x3 shortcut conv it is summed with, one
// K loop -- model8up + model3short8, model9up + model2short9, model10up + model1short10; model.py:156,170,172) rebuilt on
// v_mfma_f32_16x16x32_bf16, the MFMA shape that costs fewer joules per FLOP at the package power cap (idc_v2m.hip's header;
// profiles/r03_mfma_peak_by_shape.txt: 2.02 PFLOP/s sustained on random operands against 1.78 for the 32x32x16 shape).  These three
// launches are 26 % of the N = 32 bf16 forward.
//
// What is the same as conv_ds_fused: the workgroup (64 x 8 OUTPUT pixels x 128 couts = 2 cout waves x 4 PHASE waves, wave (wco, ph)
// owns the 32 x 4 sites whose output pixel is (2y + ro, 2x + cof)), the S part (shortcut conv over the skip tensor's 66 x 10 halo stored
// de-interleaved by x parity, shared weight tiles on a 3-slot LDS-DMA ring with the barrier one step early), the D part (wave-private
// 8 KiB weight tiles on a 2-deep ring, no workgroup barrier inside a halo chunk, tap table in lanes), register-prefetched halo chunks
// with the zero page for out-of-image rows, the 160 KiB LDS plan, the bf16-transposed whole-line stores.
// What differs (as conv_igemm_v2m differs from conv_igemm_v2): both weight images are the LAYOUT-1 ones (idc_layout.h: slot ^ (row & 7),
// cg_row_to_cout row order), halo slots are swizzled by row & 7, accumulators are 4 x 8 tiles of 16 x 16 -- lane (site r = lane & 15,
// group g = lane >> 4) register j of acc[mi][pt] is cout g*16 + mi*4 + j of the wave's 64 at site (pixel row pt >> 1, column
// (pt & 1)*16 + r) -- and one 64-channel step is two k32 sub-steps of 4 A + 8 B fragment reads and 32 MFMAs each, run as four stages of
// 16 MFMAs: a stage's fragment reads sit under the previous stage's MFMAs, the A registers of a cout block are reloaded for the next
// sub-step as soon as its last four MFMAs have issued, and the last stage of a step already reads the first fragments of the NEXT step's
// tile (published a step early by the ring).
// Same sums in a different order: results differ from conv_ds_fused's in the last bf16 bit here and there, never between batch sizes.
#include <stdlib.h>

#include <type_traits>

#include "idc_kernels.h"

#include "idc_layout.h"
#include "idc_split.hip.h"

namespace idc {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_d;

#ifdef IDC_TIMING
extern __device__ long long* g_idc_dbg;
#define IDC_DSTAMP(i) do { if (tid == 0) g_idc_dbg[(size_t)blockIdx.x * 16 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define IDC_DSTAMP(i) do {} while (0)
#endif

__device__ __forceinline__ int xcd_remap_d(int b, int nb) {
    const int xcd = b & 7, q = nb >> 3, r = nb & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (b >> 3);
}

__device__ __forceinline__ unsigned pack_bf16x2_d(float lo, float hi) {
    // one v_cvt_pk_bf16_f32 (RNE) as a VECTOR conversion: from `(__bf16)lo | (__bf16)hi << 16` the vectoriser pairs the conversions of NEIGHBOURING packs
    // and un-shuffles them with and / shift / two SDWA ors -- six instructions for two dwords instead of two (round 5: the epilogues are VALU-bound).
    // (Not inline asm: the hazard recogniser does not see an asm's reads of MFMA results, and the scheduler may move it next to the MFMAs.)
    typedef float f32x2_pk __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_pk __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_pk){lo, hi}, bf16x2_pk));
}

// NCW = cout waves per workgroup.  2: the tile of the N = 32 forward (128 couts, 8 waves).  1 (round 5, the batch-1 click path): 64 couts, four waves =
// the four phases -- model10up + shortcut at 256x256 is 128 tiles of 64 x 8 output pixels, i.e. HALF the chip with 128-cout workgroups; with 64 couts per
// workgroup it is 256 workgroups.  Same MFMAs in the same order per accumulator: bit-identical results.  Its 256 threads would need 21 halo registers each
// for the skip tensor's chunk (84 KiB), so that chunk goes to LDS by LDS-DMA (buffer loads: bounds check = zero padding; de-interleave and swizzle in the
// per-lane SOURCE address) -- once, in the prologue: the form is only launched for ONE shortcut chunk (64 skip channels: model1short10), which is the
// only deconv + shortcut launch the click path hands to this kernel.
//
// SPLIT (round 6, the operand-split precisions IDC_BF16X3 / IDC_BF16X6 / IDC_FP16X3; 1 = bf16 parts, 2 = fp16 parts): both inputs are split tensors (a pixel
// is [part][chunk] x 128 bytes), a.wgt / a.wgt2 the layout-1 images of the weight parts (w_part_bytes / w_part_bytes2 apart).  The S part and then the D part
// are walked a.nseg times -- segment s multiplies input part seg_x[s] with weight part seg_w[s], smallest products first, as conv_igemm_v2ps does -- into the
// ONE fp32 accumulator set, which starts at zero; the summed bias joins after the K loops and the epilogue is split_epilogue (fp32 activation, 2 / 3 planes).
// A segment is to the loops what a further halo chunk is: the flat chunk index q = seg * nkc + kc selects the pixel's chunk slot seg_x[seg] * nkc + kc and the
// weight image seg_w[seg]; a chunk change whose slot does not change (one-chunk tensors: hi.lo -> hi.hi) keeps the halo in LDS.  The fp32 shortcut sums of
// the two-launch form (1.07 GB written and read at conv10_1's shape) never exist.
// MODE: 0 = conv_ds_fused_m (bf16), 1 / 2 = the split forms above, 4 = MODE 0's body on fp16 operands (IDC_FP16's fast path)
template <int NCW, int MODE>
__device__ __forceinline__ void conv_ds_fused_m_body(const ConvArgs& a) {
    constexpr int SPLIT = MODE == 4 ? 0 : MODE;
    constexpr bool F16 = MODE == 2 || MODE == 4;
    constexpr int NT = NCW * 256;
    constexpr int SW = 66, SROWS = 10 * SW, S_ITEMS = (SROWS * kSlots + NT - 1) / NT, S_HALO_BYTES = S_ITEMS * NT * kSlotBytes;
    constexpr int DW = 34, DROWS = 6 * DW, D_ITEMS = (DROWS * kSlots + NT - 1) / NT, D_HALO_BYTES = D_ITEMS * NT * kSlotBytes;
    constexpr int S_WB = NCW * kWBlockBytes, D_WB = kWBlockBytes;
    constexpr int H_ITEMS = NCW == 2 ? S_ITEMS : D_ITEMS;        // halo registers: the 4-wave form stages only deconv chunks through registers
    static_assert(S_HALO_BYTES + 3 * S_WB <= 160 * 1024 && D_HALO_BYTES + NCW * 8 * D_WB <= 160 * 1024, "LDS budget");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const halo = smem;
    char* const ringS = smem + S_HALO_BYTES;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = NCW == 2 ? (wave & 1) : 0, ph = NCW == 2 ? (wave >> 1) : wave;
    char* const ringD = smem + D_HALO_BYTES + wave * 2 * D_WB;
    const int r16 = lane & 15, g16 = lane >> 4;
    const int Hs = a.Hs, Ws = a.Ws;                            // deconv input (= site) resolution; output is 2x
    const int ntx = (Ws + 31) >> 5, nty = (Hs + 3) >> 2, nct = a.ncg / NCW;
    int b = xcd_remap_d(blockIdx.x, gridDim.x);
    const int ct = b % nct; b /= nct;
    const int txi = b % ntx; b /= ntx;
    const int tyi = b % nty;
    const int n = b / nty;
    const int y0 = tyi * 4, x0 = txi * 32;
    const int ro = a.ro[ph], cof = a.co[ph];
    const int nkc = a.nkc, nkc2 = a.nkc2, ncg = a.ncg;
    const int nseg = SPLIT ? a.nseg : 1;
    const int pixD = (SPLIT ? a.in_parts * nkc : nkc) * kRowBytes, pixS = (SPLIT ? a.in_parts * nkc2 : nkc2) * kRowBytes;
    // SPLIT: input part / weight part of segment s
    auto seg_xp = [&](int sg) { return SPLIT ? (int)((a.seg_x >> (4 * sg)) & 15u) : 0; };
    auto seg_wp = [&](int sg) { return SPLIT ? (size_t)((a.seg_w >> (4 * sg)) & 15u) : (size_t)0; };
    const char* const imgD = (const char*)a.in + (size_t)n * Hs * Ws * pixD;
    const char* const imgS = (const char*)a.in2 + (size_t)n * (4 * (size_t)Hs * Ws) * pixS;
    const int cg0 = ct * NCW;
    // halo rows come through buffer loads: 32-bit offsets into one image, the bounds check returns zeros for out-of-image rows (offset 2^31)
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)imgS, 0, 4 * Hs * Ws * pixS, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)imgD, 0, Hs * Ws * pixD, 0x00020000);
    IDC_DSTAMP(0);

    // accumulators start at the (summed) bias: lane (site r16, group g16) register j of acc[mi][.] is cout g16*16 + mi*4 + j
    f32x4 acc[4][8];
    {
        const float* const bp = a.bias + (cg0 + wco) * kCoutGroup + g16 * 16;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const float4 bq = *(const float4*)(bp + mi * 4);
            const f32x4 b4 = SPLIT ? f32x4{0.f, 0.f, 0.f, 0.f} : f32x4{bq.x, bq.y, bq.z, bq.w};     // (SPLIT: the bias joins after the K loops)
#pragma unroll
            for (int pt = 0; pt < 8; ++pt) acc[mi][pt] = b4;
        }
    }

    u32x4 hreg[H_ITEMS];
    auto load_halo_S = [&](int kc2) {
      if constexpr (NCW == 2) {
        // (item -> address arithmetic recomputed per chunk on purpose, as in conv_ds_fused: hoisted, its 64-bit addresses spill)
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));
#pragma unroll
        for (int j = 0; j < S_ITEMS; ++j) {
            const int item = tid_ + j * NT;
            const int hr = item >> 3, sig = item & 7;          // LDS row (de-interleaved order) and physical slot
            const int hy = hr / SW, rem = hr - hy * SW;
            const int par = rem >= 33 ? 1 : 0, hx = 2 * (rem - par * 33) + par;
            const int Y = 2 * y0 - 1 + hy, X = 2 * x0 - 1 + hx;
            const bool inside = (unsigned)Y < (unsigned)(2 * Hs) && (unsigned)X < (unsigned)(2 * Ws) && hr < SROWS;
            const int off = (Y * (2 * Ws) + X) * pixS + ((sig ^ swz(hr)) + kc2 * kSlots) * kSlotBytes;
            hreg[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsS, inside ? off : (int)0x80000000, 0, 0));
        }
      } else {
        // 4-wave form: straight to LDS (the same item -> (LDS row, slot) map: thread tid's item j lands at (tid + j * NT) * 16)
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));
#pragma unroll
        for (int j = 0; j < S_ITEMS; ++j) {
            const int item = tid_ + j * NT;
            const int hr = item >> 3, sig = item & 7;
            const int hy = hr / SW, rem = hr - hy * SW;
            const int par = rem >= 33 ? 1 : 0, hx = 2 * (rem - par * 33) + par;
            const int Y = 2 * y0 - 1 + hy, X = 2 * x0 - 1 + hx;
            const bool inside = (unsigned)Y < (unsigned)(2 * Hs) && (unsigned)X < (unsigned)(2 * Ws) && hr < SROWS;
            const int off = (Y * (2 * Ws) + X) * pixS + ((sig ^ swz(hr)) + kc2 * kSlots) * kSlotBytes;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, (__attribute__((address_space(3))) void*)(halo + (j * NT + wave * 64) * kSlotBytes), 16,
                                                     inside ? off : (int)0x80000000, 0, 0, 0);
        }
      }
    };
    auto load_halo_D = [&](int kc) {
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));
#pragma unroll
        for (int j = 0; j < D_ITEMS; ++j) {
            const int item = tid_ + j * NT;
            const int hr = item >> 3, sig = item & 7;
            const int hy = hr / DW, hx = hr - hy * DW;
            const int Y = y0 - 1 + hy, X = x0 - 1 + hx;
            const bool inside = (unsigned)Y < (unsigned)Hs && (unsigned)X < (unsigned)Ws && hr < DROWS;
            const int off = (Y * Ws + X) * pixD + ((sig ^ swz(hr)) + kc * kSlots) * kSlotBytes;
            hreg[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsD, inside ? off : (int)0x80000000, 0, 0));
        }
#pragma unroll
        for (int j = D_ITEMS; j < H_ITEMS; ++j) hreg[j] = u32x4{0u, 0u, 0u, 0u};   // (one definition per path for every halo register)
    };
    auto dma_S = [&](int tap, int kc2, int buf) {              // 128 couts x 64 cin, shared: every wave brings 2 KiB
        const char* src = (const char*)a.wgt2 + seg_wp(0) * (size_t)a.w_part_bytes2 + (((size_t)tap * nkc2 + kc2) * ncg + cg0) * kWBlockBytes + (size_t)tid * kSlotBytes;
        char* dst = ringS + buf * S_WB + wave * 64 * kSlotBytes;
#pragma unroll
        for (int j = 0; j < 2; ++j)                            // S_WB = 2 x NT x 16 bytes in both forms
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)j * NT * kSlotBytes),
                                             (__attribute__((address_space(3))) void*)(dst + j * NT * kSlotBytes), 16, 0, 0);
    };
    static_assert(S_WB == 2 * NT * kSlotBytes, "a shortcut weight tile = two 16-byte pieces per thread");
    auto dma_D = [&](int tw, int kc, int buf) {                // this wave's 64 couts x 64 cin of its phase's tap
        int lane_ = lane;
        asm volatile("" : "+v"(lane_));
        const char* src = (const char*)a.wgt + seg_wp(0) * (size_t)a.w_part_bytes + (((size_t)tw * nkc + kc) * ncg + cg0 + wco) * kWBlockBytes + (size_t)lane_ * kSlotBytes;
        char* dst = ringD + buf * D_WB;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * 64 * kSlotBytes),
                                             (__attribute__((address_space(3))) void*)(dst + j * 64 * kSlotBytes), 16, 0, 0);
    };

    // fragments: wf[mi] = A (16 couts x 32 cin of cout block mi), xlo / xhi = B of pixel rows 0-1 / 2-3 (2 rows x 2 halves of 16 sites)
    const int wslot16 = (g16 ^ swz(r16)) * kSlotBytes;          // logical slot kk*4 + g16: kk*4 flips bit 2 of the physical slot only
    u32x4 wf[4], xlo[4], xhi[4];
    auto read_a1 = [&](const char* const wcur, const int wrow_byte, int kk, int mi) {
        wf[mi] = *(const u32x4*)(wcur + ((wrow_byte + mi * 16 * kRowBytes + wslot16) ^ (kk * 4 * kSlotBytes)));
    };
    auto read_b = [&](const int (&xaddr)[4], int kk, int half, u32x4 (&xf)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            xf[q] = *(const u32x4*)(halo + (xaddr[half * 2 + (q >> 1)] ^ (kk * 4 * kSlotBytes)) + (q & 1) * 16 * kRowBytes);
    };
    auto mma4 = [&](int mi, int half, const u32x4 (&xf)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            acc[mi][half * 4 + q] = mma_16x16x32<F16>(wf[mi], xf[q], acc[mi][half * 4 + q]);
    };
    // stage A: 16 MFMAs (all cout blocks x pixel rows 0-1) over the 4 reads of rows 2-3
#define IDC_DSM_STAGE_A()                                                             \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                            \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                            \
    }
    // stage B: 16 MFMAs (rows 2-3) cout block by cout block over the next sub-step's 4 B reads (rows 0-1) and 4 A reloads
#define IDC_DSM_STAGE_B()                                                             \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                            \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                            \
    }                                                                                 \
    _Pragma("unroll") for (int q_ = 0; q_ < 3; ++q_) {                               \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                            \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                            \
    }                                                                                 \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);

    // ---------------------------------------------------------------- S part: 3x3 conv of the skip tensor
    load_halo_S(seg_xp(0) * nkc2);
    dma_S(0, 0, 0);
    dma_S(1, 0, 1);
    if (a.warm && wave == 0) idc_warm_own_code(halo + SROWS * kRowBytes, lane, NCW == 2 ? 100 : 96);   // 12.5 of this kernel's 13.8 KB (<2> is the last kernel of its code object: stay inside -- tools/check_code_warm.py holds the count against the build); scratch: the halo area's unread tail
    static_assert(S_HALO_BYTES - SROWS * kRowBytes >= 256, "scratch for the code warm-up");
    int rt = 2, rkc = 0, rseg = 0;                             // request cursor: (tap, chunk, segment) of tile s+2
    size_t rwp = seg_wp(0) * (size_t)a.w_part_bytes2;          // ... and its weight part's offset
    auto dma_S_req = [&](int slot_off) {
        const bool real = SPLIT ? rseg < nseg : rkc < nkc2;
        const char* src = real ? (const char*)a.wgt2 + rwp + (((size_t)rt * nkc2 + rkc) * ncg + cg0) * kWBlockBytes + (size_t)tid * kSlotBytes
                               : (const char*)a.zeros + (tid & 15) * kSlotBytes;
        const size_t jstep = real ? (size_t)NT * kSlotBytes : 0;
        char* dst = ringS + slot_off + wave * 64 * kSlotBytes;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)j * jstep),
                                             (__attribute__((address_space(3))) void*)(dst + j * NT * kSlotBytes), 16, 0, 0);
    };
    int xs[4];                                                 // LDS byte address of the lane's B row (first 16 sites) per pixel row
    auto set_xs = [&](int t, int pj_lo, int pj_hi) {
        const int ky = t / 3, kx = t - ky * 3;                 // 0..2 (= tap offset + 1)
        const int c = cof + kx, par = c & 1, sh = c >> 1;      // output x = 2*xs + cof reads skip x + kx - 1: halo col 2*xs + c
#pragma unroll
        for (int pj = 0; pj < 4; ++pj) {
            if (pj < pj_lo || pj >= pj_hi) continue;
            const int xr = (2 * pj + ro + ky) * SW + par * 33 + r16 + sh;
            xs[pj] = xr * kRowBytes + ((g16 ^ swz(xr)) * kSlotBytes);
        }
    };
    const int wrowS = (wco * 64 + r16) * kRowBytes;          // (4-wave form: wco = 0)
    int off_cur = 0, off_next = S_WB, off_free = 2 * S_WB;
    if constexpr (NCW == 2) {
#pragma unroll
        for (int j = 0; j < S_ITEMS; ++j) *(u32x4*)(halo + (tid + j * NT) * kSlotBytes) = hreg[j];
    }
    set_xs(0, 0, 4);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");           // my pieces of tile 0 (tile 1 may still be in flight); 4-wave form: the halo pieces are older still
    __syncthreads();                                           // halo chunk 0 and tile 0 are visible
    IDC_DSTAMP(1);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) read_a1(ringS, wrowS, 0, mi);
    read_b(xs, 0, 0, xlo);
    int sseg = 0, skc = 0, sslot = seg_xp(0) * nkc2;           // the S loop's segment, chunk and the chunk's slot inside a pixel
    for (int q = 0, nq = nseg * nkc2; q < nq; ++q) {
        const bool last_kc = q + 1 == nq;
        int sseg_n = sseg, skc_n = skc + 1;
        if (skc_n == nkc2) { skc_n = 0; ++sseg_n; }
        const int sslot_n = SPLIT ? (last_kc ? sslot : seg_xp(sseg_n) * nkc2 + skc_n) : q + 1;
        const bool reloadS = SPLIT ? (!last_kc && sslot_n != sslot) : !last_kc;      // (a segment change on a one-chunk tensor may keep the halo: hi.lo -> hi.hi)
        auto tap_body = [&](int t, auto last_tag) {
            constexpr bool LAST = decltype(last_tag)::value;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // my pieces of the next step's tile
            __syncthreads();                                    // ... everybody's: published; everybody left slot off_free
            dma_S_req(off_free);
            if constexpr (LAST) {
                if (NCW == 2 && !last_kc) { if (reloadS && SPLIT == 0) load_halo_S(sslot_n); }   // (SPLIT: after the barrier below -- a chunk is nseg x longer there, its 44 halo registers would cost spills in every step)
                else load_halo_D(seg_xp(0) * nkc);              // the deconv input's first chunk: rows wait in registers (4-wave form: ONE shortcut chunk)
            }
            __builtin_amdgcn_sched_barrier(0);
            const char* const wcur = ringS + off_cur;
            const char* const wnxt = ringS + off_next;
            // sub-step k32 = 0
            read_b(xs, 0, 1, xhi);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) mma4(mi, 0, xlo);
            IDC_DSM_STAGE_A()
            read_b(xs, 1, 0, xlo);
            mma4(0, 1, xhi); read_a1(wcur, wrowS, 1, 0);
            mma4(1, 1, xhi); read_a1(wcur, wrowS, 1, 1);
            mma4(2, 1, xhi); read_a1(wcur, wrowS, 1, 2);
            mma4(3, 1, xhi); read_a1(wcur, wrowS, 1, 3);
            IDC_DSM_STAGE_B()
            // sub-step k32 = 1; its second stage reads the first fragments of the next step (next tap's rows, tile off_next)
            read_b(xs, 1, 1, xhi);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) mma4(mi, 0, xlo);
            IDC_DSM_STAGE_A()
            set_xs(LAST ? 0 : t + 1, 0, 4);
            if (++rt == 9) {
                rt = 0; ++rkc;
                if constexpr (SPLIT != 0) {
                    if (rkc == nkc2) { rkc = 0; ++rseg; rwp = seg_wp(rseg < nseg ? rseg : 0) * (size_t)a.w_part_bytes2; }
                }
            }
            read_b(xs, 0, 0, xlo);
            mma4(0, 1, xhi); read_a1(wnxt, wrowS, 0, 0);
            mma4(1, 1, xhi); read_a1(wnxt, wrowS, 0, 1);
            mma4(2, 1, xhi); read_a1(wnxt, wrowS, 0, 2);
            mma4(3, 1, xhi); read_a1(wnxt, wrowS, 0, 3);
            IDC_DSM_STAGE_B()
            if constexpr (LAST) {
                if constexpr (NCW == 2) {
                    if (reloadS) {
                        __syncthreads();                        // everybody is done with halo chunk kc2
                        if constexpr (SPLIT != 0) load_halo_S(sslot_n);
#pragma unroll
                        for (int j = 0; j < S_ITEMS; ++j) *(u32x4*)(halo + (tid + j * NT) * kSlotBytes) = hreg[j];
                        __syncthreads();
                        read_b(xs, 0, 0, xlo);                  // the B half of the prefetch crossed the chunk change: read it again
                    }
                }
            }
            const int o_ = off_cur; off_cur = off_next; off_next = off_free; off_free = o_;
        };
        for (int t = 0; t < 8; ++t) tap_body(t, std::false_type{});
        tap_body(8, std::true_type{});
        if constexpr (SPLIT != 0) { sseg = sseg_n; skc = skc_n; sslot = sslot_n; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the trailing zero-page requests target LDS the D part reuses
    // ---------------------------------------------------------------- hand-over: the D part reuses the whole LDS
    IDC_DSTAMP(8);
    const int* const tdy = a.dy + ph * 9;
    const int* const tdx = a.dx + ph * 9;
    const int* const ttw = a.tw + ph * 9;
    int v_xoff = 0, v_tw = 0;                                  // the phase's 2x2 taps as a table in lanes 0..3 (v_readlane: no scalar loads in the loop)
#pragma unroll
    for (int t = 0; t < 4; ++t)
        if (lane == t) { v_xoff = (1 + tdy[t]) * DW + 1 + tdx[t]; v_tw = ttw[t]; }
    __syncthreads();                                           // every wave left the S halo and ring
#pragma unroll
    for (int j = 0; j < D_ITEMS; ++j) *(u32x4*)(halo + (tid + j * NT) * kSlotBytes) = hreg[j];
    dma_D(__builtin_amdgcn_readlane(v_tw, 0), 0, 0);
    dma_D(__builtin_amdgcn_readlane(v_tw, 1), 0, 1);
    // ---------------------------------------------------------------- D part: the wave's deconv phase, 2x2 taps, wave-private weight ring
    const int wrowD = r16 * kRowBytes;
    const int nsteps = 4 * nkc * nseg;
    int xa[4];
    auto set_xa = [&](int xo) {
#pragma unroll
        for (int pj = 0; pj < 4; ++pj) {
            const int xr = pj * DW + r16 + xo;
            xa[pj] = xr * kRowBytes + ((g16 ^ swz(xr)) * kSlotBytes);
        }
    };
    set_xa(__builtin_amdgcn_readlane(v_xoff, 0));
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");           // tile 0 landed (tile 1 may still be in flight)
    __syncthreads();                                           // halo chunk 0 visible
    IDC_DSTAMP(9);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) read_a1(ringD, wrowD, 0, mi);
    read_b(xa, 0, 0, xlo);
    int dkc = 0, dseg = 0, dslot = seg_xp(0) * nkc;             // the D loop's chunk, segment and the chunk's slot inside a pixel
    int qkc = 0, qseg = 0;                                     // request cursor: (chunk, segment) of step st + 2 (steps 0 and 1 are in flight)
    size_t qwp = seg_wp(0) * (size_t)a.w_part_bytes;
    for (int st = 0; st < nsteps; ++st) {
        const int t = st & 3;
        const char* const wcur = ringD + (st & 1) * D_WB;
        const char* const wnext = ringD + ((st + 1) & 1) * D_WB;
        bool swap_ = t == 3 && st + 1 < nsteps;
        if constexpr (SPLIT != 0) {
            if (t == 3) {
                if (++dkc == nkc) { dkc = 0; ++dseg; }
                const int dslot_n = swap_ ? seg_xp(dseg) * nkc + dkc : dslot;
                swap_ = swap_ && dslot_n != dslot;
                dslot = dslot_n;
            }
        } else {
            dslot = (st >> 2) + 1;
        }
        const bool swap = swap_;
        if (swap) load_halo_D(dslot);                          // next chunk's rows wait in registers
        read_b(xa, 0, 1, xhi);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) mma4(mi, 0, xlo);
        IDC_DSM_STAGE_A()
        read_b(xa, 1, 0, xlo);
        mma4(0, 1, xhi); read_a1(wcur, wrowD, 1, 0);
        mma4(1, 1, xhi); read_a1(wcur, wrowD, 1, 1);
        mma4(2, 1, xhi); read_a1(wcur, wrowD, 1, 2);
        mma4(3, 1, xhi); read_a1(wcur, wrowD, 1, 3);
        IDC_DSM_STAGE_B()
        read_b(xa, 1, 1, xhi);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) mma4(mi, 0, xlo);
        IDC_DSM_STAGE_A()
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // tile st+1 landed; every fragment read of tile st is back
        {
            const int s2 = st + 2;
            const bool real = s2 < nsteps;
            if constexpr (SPLIT != 0) {
                if ((s2 & 3) == 0) {                            // step s2 opens a chunk: advance the request cursor
                    if (++qkc == nkc) { qkc = 0; ++qseg; qwp = seg_wp(qseg < nseg ? qseg : 0) * (size_t)a.w_part_bytes; }
                }
            } else {
                qkc = s2 >> 2;
            }
            const int tw2 = __builtin_amdgcn_readlane(v_tw, s2 & 3);
            int lane_ = lane;
            asm volatile("" : "+v"(lane_));
            const char* src = real ? (const char*)a.wgt + qwp + (((size_t)tw2 * nkc + qkc) * ncg + cg0 + wco) * kWBlockBytes + (size_t)lane_ * kSlotBytes
                                   : (const char*)a.zeros + (lane_ & 15) * kSlotBytes;
            const int jstep = real ? 64 * kSlotBytes : 0;
            char* dst = ringD + (st & 1) * D_WB;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)j * jstep),
                                                 (__attribute__((address_space(3))) void*)(dst + j * 64 * kSlotBytes), 16, 0, 0);
        }
        set_xa(__builtin_amdgcn_readlane(v_xoff, (st + 1) & 3));
        read_b(xa, 0, 0, xlo);
        mma4(0, 1, xhi); read_a1(wnext, wrowD, 0, 0);
        mma4(1, 1, xhi); read_a1(wnext, wrowD, 0, 1);
        mma4(2, 1, xhi); read_a1(wnext, wrowD, 0, 2);
        mma4(3, 1, xhi); read_a1(wnext, wrowD, 0, 3);
        IDC_DSM_STAGE_B()
        if (swap) {
            __syncthreads();                                   // every wave is done with halo chunk kc
#pragma unroll
            for (int j = 0; j < D_ITEMS; ++j) *(u32x4*)(halo + (tid + j * NT) * kSlotBytes) = hreg[j];
            __syncthreads();
            read_b(xa, 0, 0, xlo);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (the zero-page requests of the last two steps target this ring)
#undef IDC_DSM_STAGE_A
#undef IDC_DSM_STAGE_B
    // ---------------------------------------------------------------- epilogue: (ReLU,) round, transpose, whole-line stores
    IDC_DSTAMP(2);
    __syncthreads();
    if constexpr (SPLIT != 0) {
        add_bias_after_k(a.bias + (cg0 + wco) * kCoutGroup + g16 * 16, acc, a.acc_scale);
        split_epilogue<NCW, SPLIT == 2>(a, acc, smem, n, y0, x0, 0, (cg0 + wco) * kCoutGroup, ro, cof);
        IDC_DSTAMP(3);
        return;
    }
    char* const tb16 = smem + wave * 4096;                     // wave-private [32 sites][64 couts] bf16, 128-byte rows, slot ^ (site & 7)
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    const int rr = lane >> 3, cc = lane & 7;
    const int CoutPad = ncg * kCoutGroup;
    const int co8 = (cg0 + wco) * kCoutGroup + cc * 8;
    const int Wout = 2 * Ws, Hout = 2 * Hs;
    // (as conv_igemm_v2p's epilogue, round 5: one body per ReLU setting chosen once, the row's four transposed lines read BEFORE the first store's
    //  bounds check -- the compiler sank each read under its store -- and one 64-bit base per lane with 32-bit strides)
    unsigned short* const out00 = (unsigned short*)a.out + (((size_t)n * Hout + (2 * y0 + ro)) * Wout + (2 * (x0 + rr) + cof)) * CoutPad + co8;
    auto rows = [&](auto relu_c) __attribute__((always_inline)) {
        constexpr bool RELU = decltype(relu_c)::value;
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
                        unsigned p = pack16x2_m<F16>(acc[mi][pt][2 * e], acc[mi][pt][2 * e + 1]);
                        if constexpr (RELU) p = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p), s16x2{0, 0}));
                        pk[mi * 2 + e] = p;
                    }
                const int s0 = g16 * 2;                         // the lane's 16 couts = slots 2g, 2g+1 of the site's 128-byte row
                *(uint4*)(tb16 + site * 128 + ((s0 ^ (site & 7)) * 16)) = uint4{pk[0], pk[1], pk[2], pk[3]};
                *(uint4*)(tb16 + site * 128 + (((s0 + 1) ^ (site & 7)) * 16)) = uint4{pk[4], pk[5], pk[6], pk[7]};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // same-wave LDS ops are in order: the row tile is complete
            const int sy = y0 + pj;
            auto line = [&](int i) { const int row = i * 8 + rr; return *(const uint4*)(tb16 + row * 128 + ((cc ^ (row & 7)) * 16)); };
            const uint4 o0 = line(0), o1 = line(1), o2 = line(2), o3 = line(3);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // reads retired before the tile is rewritten (and before the first bounds check)
            auto put = [&](int i, const uint4& o) {
                const int sx = x0 + i * 8 + rr;
                if (sy < Hs && sx < Ws) *(uint4*)(out00 + (2 * pj * Wout + 2 * i * 8) * CoutPad) = o;
            };
            put(0, o0); put(1, o1); put(2, o2); put(3, o3);
        }
    };
    if (a.act == 1) rows(std::true_type{}); else rows(std::false_type{});
    IDC_DSTAMP(3);
#ifdef IDC_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    IDC_DSTAMP(4);
#endif
}

template <int NCW>
__global__ __launch_bounds__(NCW * 256, 2) void conv_ds_fused_m(const ConvArgs a) { conv_ds_fused_m_body<NCW, 0>(a); }
template <int NCW>      // IDC_FP16: conv_ds_fused_m on fp16 operands
__global__ __launch_bounds__(NCW * 256, 2) void conv_ds_fused_mh(const ConvArgs a) { conv_ds_fused_m_body<NCW, 4>(a); }
// the operand-split forms: bf16 parts (IDC_BF16X3 / IDC_BF16X6) and fp16 parts (IDC_FP16X3); 8-wave workgroups only
__global__ __launch_bounds__(512, 2) void conv_ds_fused_ms(const ConvArgs a) { conv_ds_fused_m_body<2, 1>(a); }
__global__ __launch_bounds__(512, 2) void conv_ds_fused_msh(const ConvArgs a) { conv_ds_fused_m_body<2, 2>(a); }

// deconv 4x4 s2 + its 3x3 shortcut conv in one launch, 16x16x32 MFMA: bf16, Cout a multiple of 128, (ReLU | none), no BN.
// a.wgt / a.wgt2 = the LAYOUT-1 images of the deconv / the shortcut conv.
// buffer loads address ONE image with 32-bit offsets (out-of-image rows: offset 2^31)
static int g_ds_half = 1;                 // 0 ("ds_mfma16" = 2): keep the 128-cout workgroups on small grids (A/B, tests)
void set_ds_half(int v) { g_ds_half = v != 0; }

bool conv_ds_m_fits(int Hs, int Ws, int nkc, int nkc2) {
    return (long long)4 * Hs * Ws * ((long long)(nkc2 > nkc ? nkc2 : nkc) * kRowBytes) < 0x7fffffffLL;
}

hipError_t launch_conv_ds_m(const ConvArgs& a, hipStream_t s) {
    if (a.in2 == nullptr || a.wgt2 == nullptr || a.zeros == nullptr || a.nphase != 4 || a.so != 2 || a.si != 1 || (a.ncg & 1) || a.out_f32 ||
        a.bn_scale != nullptr || a.act == 2 || a.img_shift != nullptr || a.resid != nullptr || a.head_w != nullptr ||
        !conv_ds_m_fits(a.Hs, a.Ws, a.nkc, a.nkc2))
        return hipErrorInvalidConfiguration;
    const long long blocks = (long long)((a.Ws + 31) / 32) * ((a.Hs + 3) / 4) * a.N * (a.ncg / 2);
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    // fewer 128-cout workgroups than CUs (model10up + shortcut of ONE 256x256 image: 128) and a single shortcut chunk: the 64-cout, 4-wave form
    static int n_cu = 0;
    if (n_cu == 0) { int dev = 0; hipDeviceProp_t pr; n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : -1; }
    const bool half = g_ds_half && a.nkc2 == 1 && n_cu > 0 && blocks < n_cu;
    if (a.split_f16) {                                           // IDC_FP16's fast path
        if (half) hipLaunchKernelGGL(conv_ds_fused_mh<1>, dim3((unsigned)(2 * blocks)), dim3(256), 160 * 1024, s, a);
        else hipLaunchKernelGGL(conv_ds_fused_mh<2>, dim3((unsigned)blocks), dim3(512), 160 * 1024, s, a);
    } else if (half)
        hipLaunchKernelGGL(conv_ds_fused_m<1>, dim3((unsigned)(2 * blocks)), dim3(256), 160 * 1024, s, a);
    else
        hipLaunchKernelGGL(conv_ds_fused_m<2>, dim3((unsigned)blocks), dim3(512), 160 * 1024, s, a);
    return hipGetLastError();
}

// ... its operand-split form: a.in / a.in2 / a.out split tensors of a.in_parts (= a.out_parts) planes, a.wgt / a.wgt2 the weight parts' layout-1 images
hipError_t launch_conv_ds_ms(const ConvArgs& a, hipStream_t s) {
    if (a.in2 == nullptr || a.wgt2 == nullptr || a.zeros == nullptr || a.nphase != 4 || a.so != 2 || a.si != 1 || (a.ncg & 1) || a.out_f32 ||
        a.img_shift != nullptr || a.resid != nullptr || a.head_w != nullptr || a.in_parts < 1 || a.in_parts > 3 || a.out_parts != a.in_parts ||
        a.nseg < 1 || a.nseg > 6 || a.w_part_bytes == 0 || a.w_part_bytes2 == 0 ||
        !conv_ds_m_fits(a.Hs, a.Ws, a.nkc * a.in_parts, a.nkc2 * a.in_parts))
        return hipErrorInvalidConfiguration;
    const long long blocks = (long long)((a.Ws + 31) / 32) * ((a.Hs + 3) / 4) * a.N * (a.ncg / 2);
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    if (a.split_f16) hipLaunchKernelGGL(conv_ds_fused_msh, dim3((unsigned)blocks), dim3(512), 160 * 1024, s, a);
    else hipLaunchKernelGGL(conv_ds_fused_ms, dim3((unsigned)blocks), dim3(512), 160 * 1024, s, a);
    return hipGetLastError();
}

hipError_t init_kernels_dsm() {
    hipError_t e = hipFuncSetAttribute((const void*)conv_ds_fused_m<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)conv_ds_fused_mh<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)conv_ds_fused_mh<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)conv_ds_fused_ms, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)conv_ds_fused_msh, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)conv_ds_fused_m<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace idc
