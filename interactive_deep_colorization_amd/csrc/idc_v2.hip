// idc_v2.hip -- the 32x32x16-MFMA large-tile family: conv_igemm_v2 and conv_ds_fused (the deconv + shortcut launch on that MFMA shape).
// Since round 3 the throughput path runs their 16x16x32 twins (idc_v2m.hip, idc_dsm.hip); these serve the launches the twins do not cover
// (fp32 shortcut sums, fp32 outputs, images beyond 32-bit addressing) and the "mfma16" / "ds_mfma16" = 0 A/B.
// Round 6: compiled ONLY with -DIDC_AB_PARTNERS (make EXTRA=-DIDC_AB_PARTNERS: the A/B tools and the partner tests).  The default library plans those
// launches on the small-tile kernels (the census of tools/kernel_census.py: class_logits at batch >= 8; per-image shifts moved into conv_igemm_v2p).
#include <stdlib.h>
#include <type_traits>

#include "idc_kernels.h"

#include "idc_layout.h"

#include "idc_common.hip.h"

namespace idc {

#ifdef IDC_AB_PARTNERS


// ================================================================================================
// conv_igemm_v2<WCO, WPX, HALO> -- the throughput kernel (bf16): 8 waves, 32x32x16 MFMA.
//   * workgroup = (32 x 4*WPX) output sites of one image  x  64*WCO output channels;
//     wave = 64 couts x 128 pixels (4 spatial rows of 32) = 2x4 accumulator tiles of 32x32
//     (128 accumulator registers): 6 ds_read_b128 per 8 MFMA (v1: 8 per 16 half-size MFMA), i.e.
//     the LDS array runs ~40 % busy at full MFMA rate instead of ~80 %;
//   * weight tiles (pre-swizzled LDS images, layout 2 of idc_layout.h) go global -> LDS by
//     LDS-DMA (global_load_lds_dwordx4, no VGPR round trip, no ds_write), 2-deep ring, issued one
//     tap ahead right after the barrier; one vmcnt(0) + barrier per tap (= per 1024 MFMA cycles);
//   * halo rows: register-prefetched under the last tap of the previous chunk (as v1);
//   * swizzle (row>>1)&7: conflict-free for 32-row x 2-k-group fragments (tools/bank model).
// ================================================================================================

template <int WCO, int WPX, int HALO>
__global__ __launch_bounds__(WCO* WPX * 64, 2) void conv_igemm_v2(const ConvArgs a) {
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
    const int px = lane & 31, h = lane >> 5;
    IDC_STAMP(0);

    // tile order: (deconv phase, cout tile) vary fastest, so the workgroups that share an input halo run
    // back to back on one XCD (xcd_remap) and read it from that XCD's L2
    int b = xcd_remap(blockIdx.x, gridDim.x);
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
    const size_t w_kc_stride = (size_t)a.ncg * kWBlockBytes;   // next cin chunk (both sources: same couts)
    const size_t w_lane = (size_t)(ct * WCO) * kWBlockBytes + (size_t)tid * kSlotBytes;

    // ---- K loop: one source (a.in / a.wgt / tap tables: 3x3 conv, 1x1 conv or one deconv phase) --------------
    struct Stage { const char* img; const char* wb; int nkc, ntaps, si, oy, ox; };
    auto make_stage = [&](int) -> Stage {
        Stage st;
        const size_t pix = (size_t)a.nkc * kRowBytes;
        st.img = (const char*)a.in + (size_t)n * (size_t)(Hs * a.si) * (Ws * a.si) * pix;
        st.wb = (const char*)a.wgt + w_lane;
        st.nkc = a.nkc; st.ntaps = a.ntaps; st.si = a.si; st.oy = 0; st.ox = 0;
        return st;
    };
    auto tap_of = [&](int, int t, int& dy, int& dx, int& tw) { dy = tap_dy[t]; dx = tap_dx[t]; tw = tap_tw[t]; };
    constexpr int nstage = 1;

    // accumulators start at the bias (bf16-output launches; the fp32-output epilogue adds it itself): lane
    // (pixel px, half h) register r of acc[mi][.] is cout h*32 + mi*16 + r of the wave's 64 (idc_layout.h layout 2)
    f32x16 acc[2][4];
    {
        const float* const bp = a.bias + (ct * WCO + wco) * kCoutGroup + h * 32;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x16 b16;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = a.out_f32 ? float4{0.f, 0.f, 0.f, 0.f} : *(const float4*)(bp + i * 16 + q * 4);
                b16[q * 4 + 0] = bq.x; b16[q * 4 + 1] = bq.y; b16[q * 4 + 2] = bq.z; b16[q * 4 + 3] = bq.w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = b16;
        }
    }

    // LDS-DMA of one weight tile: lane-linear destination (wave-uniform base + lane*16)
    auto dma_w = [&](const Stage& st, int tw, int kc, int buf) {
        const char* src = st.wb + ((size_t)tw * st.nkc + kc) * w_kc_stride;
        char* dst = wbuf + buf * W_BYTES + wave * 64 * kSlotBytes;
#pragma unroll
        for (int j = 0; j < N_WITEMS; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)j * NT * kSlotBytes),
                                             (__attribute__((address_space(3))) void*)(dst + j * NT * kSlotBytes), 16, 0, 0);
    };
    // halo staging of one 128-byte channel chunk: item = (halo row, physical slot), zero outside the image;
    // the rows go to registers now and to LDS after the chunk-end barrier (issue early / write late)
    u32x4 hreg[N_HITEMS];
    auto load_halo = [&](const Stage& st, int kc) {
        const int Win = Ws * st.si, pix_bytes = st.nkc * kRowBytes;
        int tid_ = tid;
        if constexpr (NT == 256 && HALO == 2) asm volatile("" : "+v"(tid_));   // 14 items: recompute their addresses per chunk
                                                                  // (hoisted, the 64-bit selects cost hipcc 9 spilled registers)
#pragma unroll
        for (int j = 0; j < N_HITEMS; ++j) {
            const int item = tid_ + j * NT;
            const int hr = item >> 3, sig = item & 7;
            const int hy = hr / HWP, hx = hr - hy * HWP;
            const int sy = ty0 - HALO + hy, sx = tx0 - HALO + hx;
            const bool inside = (unsigned)sy < (unsigned)Hs && (unsigned)sx < (unsigned)Ws && item < HROWS * kSlots;
            const int off = ((sy * st.si + st.oy) * Win + sx * st.si + st.ox) * pix_bytes + ((sig ^ swz2(hr)) + kc * kSlots) * kSlotBytes;
            // out-of-image rows read the zero page: a select on the ADDRESS, none on the loaded value (a select on the value
            // sits in front of the step's MFMAs and makes hipcc wait there for the loads it has just issued)
            hreg[j] = *(const u32x4*)(inside ? st.img + off : (const char*)a.zeros);
        }
    };

    Stage cur = make_stage(0);
    IDC_STAMP_FINE(5);
    load_halo(cur, 0);
    {
        int dy0, dx0, tw0;
        tap_of(0, 0, dy0, dx0, tw0);
        dma_w(cur, tw0, 0, 0);
    }
    IDC_STAMP_FINE(6);
    // bf16 shortcut partial sums (model.py:156,170,172) are added into the accumulators here, in the MFMA layout
    // (2 x 32 B per lane and pixel row): their latency hides behind the halo fetch that is already in flight, and the
    // epilogue of a deconv + shortcut launch becomes the plain one
    const bool resid_in_acc = !a.out_f32 && a.resid != nullptr && a.resid_bf16;
    if (resid_in_acc) {
        const int so_ = a.so, Wout_ = Ws * so_, Hout_ = Hs * so_, cpad_ = a.ncg * kCoutGroup;
        const int sx = tx0 + px;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            uint4 rv[2][2][2];
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int sy = ty0 + wpx * 4 + half * 2 + pp;
                const bool inside = sy < Hs && sx < Ws;
                const size_t ridx = (((size_t)n * Hout_ + (sy * so_ + ro)) * Wout_ + (sx * so_ + cof)) * cpad_ +
                                    (ct * WCO + wco) * kCoutGroup + h * 32;
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        rv[pp][mi][q] = inside ? *(const uint4*)((const unsigned short*)a.resid + ridx + mi * 16 + q * 8)
                                               : uint4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const unsigned rw[4] = {rv[pp][mi][q].x, rv[pp][mi][q].y, rv[pp][mi][q].z, rv[pp][mi][q].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc[mi][half * 2 + pp][q * 8 + 2 * e] += __uint_as_float(rw[e] << 16);
                            acc[mi][half * 2 + pp][q * 8 + 2 * e + 1] += __uint_as_float(rw[e] & 0xffff0000u);
                        }
                    }
        }
    }

    const int wrow_byte = (wco * 64 + px) * kRowBytes;         // + mi*32 rows
    // swz2(row) = (row>>1)&7 is the same for rows px and px+32 (and +64*wco): one slot term serves both
    const int wslot0 = (h ^ swz2(px)) * kSlotBytes;
    int buf = 0;
    bool first = true;
    // LDS byte address of the lane's B-operand row for each of the wave's 4 pixel rows, for the tap about to run.  It is
    // computed under the previous tap's last MFMAs, so that a tap starts with its fragment reads, not with ~35 VALU.
    int xa[4];
    auto set_xa = [&](int dy, int dx) {
#pragma unroll
        for (int pj = 0; pj < 4; ++pj) {
            const int xr = (wpx * 4 + pj + HALO + dy) * HWP + (px + HALO + dx);
            xa[pj] = xr * kRowBytes + ((h ^ swz2(xr)) * kSlotBytes);              // slot = kk*2 + h: kk*2 flips bits 1,2 only
        }
    };
    set_xa(tap_dy[0], tap_dx[0]);
    // static priority for the second-dispatched half of the workgroup: on every SIMD it is the arbitration loser
    // (MI355X_MICROARCH.md, two waves per SIMD); same-box A/B -0.5 % per forward
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    // weight-tile index the NEXT tap will request right after its barrier (tap t+1 requests tap t+2's tile, the chunk's
    // last tap the first tile of the next chunk): read from the tap table one tap early, so that no scalar load sits
    // between the barrier and the first fragment reads
    int tw_dma = a.ntaps > 1 ? tap_tw[1] : tap_tw[0];
#ifdef IDC_STEP_PROBE
    // intra-step probe of the tuning harness: cycle counter at seven points of ONE steady-state step (chunk 2, tap 4), kept in
    // scalar registers (selected, not branched) and stored after the K loop
    long long pr_[7] = {0, 0, 0, 0, 0, 0, 0};
#define IDC_PROBE(i) const long long pn##i = (long long)__builtin_readcyclecounter();
#define IDC_PROBE_KEEP() { const bool on_ = kc == 2 && t == 4; pr_[0] = on_ ? pn0 : pr_[0]; pr_[1] = on_ ? pn1 : pr_[1]; pr_[2] = on_ ? pn2 : pr_[2]; \
    pr_[3] = on_ ? pn3 : pr_[3]; pr_[4] = on_ ? pn4 : pr_[4]; pr_[5] = on_ ? pn5 : pr_[5]; pr_[6] = on_ ? pn6 : pr_[6]; }
#else
#define IDC_PROBE(i)
#define IDC_PROBE_KEEP()
#endif
    for (int q = 0; q < nstage; ++q) {
        for (int kc = 0; kc < cur.nkc; ++kc) {
            __syncthreads();                   // previous chunk's halo reads are done
#pragma unroll
            for (int j = 0; j < N_HITEMS; ++j) *(u32x4*)(halo + (tid + j * NT) * kSlotBytes) = hreg[j];
            if (first) IDC_STAMP_FINE(7);
            const bool last_kc = kc + 1 == cur.nkc;
            // the tap body exists twice: taps 0 .. ntaps-2 only stream the next weight tile; the chunk's last tap also
            // fetches the next chunk's halo rows.  (As one loop with a branch, hipcc merges the 24-40 halo registers of the
            // two paths with v_mov_b64 copies on EVERY tap.)
            auto tap_body = [&](int t, auto last_tag) {
                constexpr bool LAST = decltype(last_tag)::value;
                const char* const wcur = wbuf + buf * W_BYTES;
                IDC_PROBE(0)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // my pieces of this tap's weight tile landed
                __syncthreads();               // everybody's landed; everybody left the other buffer
                IDC_PROBE(1)
                if (first) { IDC_STAMP(1); first = false; }
                const int xaddr[4] = {xa[0], xa[1], xa[2], xa[3]};
                // explicit 2-stage software pipeline over the four k16 steps of the chunk: fragments of
                // step kk+1 are in flight while the 8 MFMAs of step kk issue
                u32x4 wfA[2], xfA[4], wfB[2], xfB[4];
                auto read_frags = [&](int kk, u32x4 (&wf)[2], u32x4 (&xf)[4]) {
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
                        wf[mi] = *(const u32x4*)(wcur + ((wrow_byte + mi * 32 * kRowBytes + wslot0) ^ (kk * 2 * kSlotBytes)));
#pragma unroll
                    for (int pj = 0; pj < 4; ++pj)
                        xf[pj] = *(const u32x4*)(halo + (xaddr[pj] ^ (kk * 2 * kSlotBytes)));
                };
                auto mma8 = [&](const u32x4 (&wf)[2], const u32x4 (&xf)[4]) {
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int pj = 0; pj < 4; ++pj)
                            acc[mi][pj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[mi]),
                                                                                  __builtin_bit_cast(bf16x8, xf[pj]),
                                                                                  acc[mi][pj], 0, 0, 0);
                };
                // Pin the issue order (hipcc's scheduler otherwise collapses the pipeline to save
                // registers): 6 reads up front, then per stage 1 MFMA : 1 ds_read interleaved.
#define IDC_STAGE_INTERLEAVE()                                                        \
    _Pragma("unroll") for (int q_ = 0; q_ < 6; ++q_) {                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                            \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                            \
    }                                                                                 \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                read_frags(0, wfA, xfA);
                __builtin_amdgcn_sched_barrier(0);
                // the NEXT step's loads, behind this tap's first fragment reads: its weight tile (LDS-DMA into the other
                // buffer) and, when it opens a new chunk, that chunk's halo rows (to registers, written after the
                // chunk-end barrier) -- they land under this tap's 32 MFMAs per wave
                if constexpr (!LAST) {
                    dma_w(cur, tw_dma, kc, buf ^ 1);
                } else if (!last_kc) {
                    dma_w(cur, tw_dma, kc + 1, buf ^ 1);
                    load_halo(cur, kc + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
                IDC_PROBE(2)
                read_frags(1, wfB, xfB);
                mma8(wfA, xfA);
                IDC_STAGE_INTERLEAVE()
#ifdef IDC_STEP_PROBE
                __builtin_amdgcn_sched_barrier(0);
#endif
                IDC_PROBE(3)
                read_frags(2, wfA, xfA);
                mma8(wfB, xfB);
                IDC_STAGE_INTERLEAVE()
#ifdef IDC_STEP_PROBE
                __builtin_amdgcn_sched_barrier(0);
#endif
                IDC_PROBE(4)
                read_frags(3, wfB, xfB);
                mma8(wfA, xfA);
                IDC_STAGE_INTERLEAVE()
#ifdef IDC_STEP_PROBE
                __builtin_amdgcn_sched_barrier(0);
#endif
                IDC_PROBE(5)
                {
                    const int tn = LAST ? 0 : t + 1;            // the tap that runs next
                    set_xa(tap_dy[tn], tap_dx[tn]);
                    tw_dma = tn + 1 < cur.ntaps ? tap_tw[tn + 1] : tap_tw[0];    // what tap tn requests: tap tn+1's tile, or the next chunk's first
                }
                mma8(wfB, xfB);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
#undef IDC_STAGE_INTERLEAVE
#ifdef IDC_STEP_PROBE
                __builtin_amdgcn_sched_barrier(0);
#endif
                IDC_PROBE(6)
                IDC_PROBE_KEEP()
                buf ^= 1;
            };
            for (int t = 0; t + 1 < cur.ntaps; ++t) tap_body(t, std::false_type{});
            tap_body(cur.ntaps - 1, std::true_type{});
        }
        IDC_STAMP(8 + q);
        if (q + 1 < nstage) cur = make_stage(q + 1);
    }

    IDC_STAMP(2);
#ifdef IDC_STEP_PROBE
    if (tid == 0) for (int i = 0; i < 7; ++i) g_idc_dbg[(size_t)blockIdx.x * 16 + 9 + i] = pr_[i];
#endif
    // ---- epilogue: lane (pixel px, half h) owns couts h*32 + mi*16 + reg of its wave's 64 ----------
    const int CoutPad = a.ncg * kCoutGroup;
    const bool has_bn = a.bn_scale != nullptr;
    const int so = a.so, Wout = Ws * so, Hout = Hs * so;
    const int cow = (ct * WCO + wco) * kCoutGroup;             // first cout of this wave
    if (a.out_f32) {
        // fp32 outputs (class logits): straight from the MFMA layout, 64 B per lane
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int co0 = cow + h * 32 + mi * 16;
            float bias[16], bsc[16], bsh[16];
            load16(bias, a.bias + co0);
            if (has_bn) { load16(bsc, a.bn_scale + co0); load16(bsh, a.bn_shift + co0); }
#pragma unroll
            for (int pj = 0; pj < 4; ++pj) {
                const int sy = ty0 + wpx * 4 + pj, sx = tx0 + px;
                if (sy < Hs && sx < Ws) {
                    const size_t opix = ((size_t)n * Hout + (sy * so + ro)) * Wout + (sx * so + cof);
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = acc[mi][pj][r];
                    epilogue16<true>(a, v, opix * CoutPad + co0, bias, bsc, bsh, has_bn,
                                     a.img_shift ? a.img_shift + (size_t)n * CoutPad + co0 : nullptr);
                }
            }
        }
    } else {
        // bf16 outputs.  The MFMA layout gives a lane 2 x 16 couts of one pixel, so a direct store
        // instruction would touch 64 pieces of 32 different 128-B lines (measured: +13..20 % kernel
        // time).  Instead the raw fp32 accumulators of one pixel row (32 pixels x the wave's 64 couts)
        // go through a wave-private 8 KiB LDS tile ([32][64] fp32, 16-B slot ^ (row&7): conflict-free
        // both ways) and come back with lane = (pixel l>>3, 8 consecutive couts l&7): bias, shortcut
        // sum, activation and eval-BN run on that layout with 8-wide per-lane constants, and every
        // global load/store instruction covers 8 whole 128-B lines.
        const bool fuse_head = WCO == 2 && a.head_w != nullptr;
        __syncthreads();                                       // every wave left the halo / weight tiles
        float* const tb = (float*)(smem + wave * 8192);
        float* const part = (float*)(smem + (NT / 64) * 8192);  // fused head: [wave][pj][32 px][2]
        // (lane index recomputed from the hardware counter: held across the K loop it costs hipcc a spilled register)
        const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const int rr = lane_e >> 3, cc = lane_e & 7;
        const int co8 = cow + cc * 8;
        float cs[8], ct[8], hw0[8], hw1[8];
        if (has_bn) {
            const float4 s0 = *(const float4*)(a.bn_scale + co8), s1 = *(const float4*)(a.bn_scale + co8 + 4);
            const float4 t0 = *(const float4*)(a.bn_shift + co8), t1 = *(const float4*)(a.bn_shift + co8 + 4);
            cs[0] = s0.x; cs[1] = s0.y; cs[2] = s0.z; cs[3] = s0.w; cs[4] = s1.x; cs[5] = s1.y; cs[6] = s1.z; cs[7] = s1.w;
            ct[0] = t0.x; ct[1] = t0.y; ct[2] = t0.z; ct[3] = t0.w; ct[4] = t1.x; ct[5] = t1.y; ct[6] = t1.z; ct[7] = t1.w;
        }
        if (fuse_head) {
            const float4 u0 = *(const float4*)(a.head_w + co8), u1 = *(const float4*)(a.head_w + co8 + 4);
            const float4 q0 = *(const float4*)(a.head_w + 128 + co8), q1 = *(const float4*)(a.head_w + 128 + co8 + 4);
            hw0[0] = u0.x; hw0[1] = u0.y; hw0[2] = u0.z; hw0[3] = u0.w; hw0[4] = u1.x; hw0[5] = u1.y; hw0[6] = u1.z; hw0[7] = u1.w;
            hw1[0] = q0.x; hw1[1] = q0.y; hw1[2] = q0.z; hw1[3] = q0.w; hw1[4] = q1.x; hw1[5] = q1.y; hw1[6] = q1.z; hw1[7] = q1.w;
        }
        // bf16 shortcut partials: all 16 loads of the lane go out before the first store (gfx9 counts loads and
        // stores in one in-order vmcnt, so a load issued after a store cannot be waited for without also waiting
        // for that store's L2 acknowledgement -- once per pixel row otherwise)
        // Layers whose epilogue is only (ReLU +) rounding -- no BN, shortcut sum, LeakyReLU, per-image shift or head
        // (15 of the 27 large-tile launches) -- round in the MFMA layout and transpose bf16 instead of fp32: half the
        // LDS traffic, ReLU as one v_pk_max_i16 per pair (a bf16 is negative iff its int16 pattern is), no per-lane
        // constants.  [32 px][64 couts] bf16 = 128-B rows, 16-B slot ^ (px & 7): conflict-free both ways.
        // Fused tanh head (conv10_2 -> model_out, model.py:101-109): the activation and the 128 -> 2 dot product run in
        // the MFMA layout (32 couts of one pixel per lane); the four partial sums of a pixel (2 lane halves x 2 cout
        // waves) meet in LDS.  conv10_2 itself is never rounded or stored, and nothing is transposed.
        if (fuse_head && !has_bn && a.img_shift == nullptr && (a.resid == nullptr || resid_in_acc)) {
            f32x16 w0[2], w1[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 u = *(const float4*)(a.head_w + cow + h * 32 + mi * 16 + q * 4);
                    const float4 v = *(const float4*)(a.head_w + 128 + cow + h * 32 + mi * 16 + q * 4);
                    w0[mi][q * 4 + 0] = u.x; w0[mi][q * 4 + 1] = u.y; w0[mi][q * 4 + 2] = u.z; w0[mi][q * 4 + 3] = u.w;
                    w1[mi][q * 4 + 0] = v.x; w1[mi][q * 4 + 1] = v.y; w1[mi][q * 4 + 2] = v.z; w1[mi][q * 4 + 3] = v.w;
                }
            float* const hp = (float*)smem;                     // [wave][pj][half][32 px][2]
#pragma unroll
            for (int pj = 0; pj < 4; ++pj) {
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[mi][pj][r];
                        if (a.act == 1) v = fmaxf(v, 0.f);
                        else if (a.act == 2) v = fmaxf(v, 0.2f * v);
                        s0 = fmaf(v, w0[mi][r], s0);
                        s1 = fmaf(v, w1[mi][r], s1);
                    }
                *(float2*)(hp + ((((wave * 4 + pj) * 2 + h) * 32 + px) * 2)) = float2{s0, s1};
            }
            __syncthreads();
            if (wco == 0) {                                     // waves wave, wave+1 hold the two cout halves of these pixels
                const float hb = a.head_b[h];
#pragma unroll
                for (int pj = 0; pj < 4; ++pj) {
                    float p = hb;
#pragma unroll
                    for (int w2 = 0; w2 < 2; ++w2)
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) p += hp[((((wave + w2) * 4 + pj) * 2 + hh) * 32 + px) * 2 + h];
                    const int sy = ty0 + wpx * 4 + pj, sx = tx0 + px;
                    if (sy < Hs && sx < Ws) a.head_out[(((size_t)n * 2 + h) * Hs + sy) * Ws + sx] = tanhf(p) * a.head_mul;
                }
            }
            IDC_STAMP(3);
#ifdef IDC_TIMING
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            IDC_STAMP(4);
#endif
            return;
        }
        const bool cheap = (a.resid == nullptr || resid_in_acc) && a.act != 2 && a.img_shift == nullptr && !fuse_head;
        if (cheap) {
            char* const tb16 = smem + wave * 4096;
            typedef short s16x2 __attribute__((ext_vector_type(2)));
            f32x16 bsc[2], bsh[2];                              // eval-BN affine of the lane's 32 couts (after the ReLU)
            if (has_bn) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 s4 = *(const float4*)(a.bn_scale + cow + h * 32 + mi * 16 + q * 4);
                        const float4 t4 = *(const float4*)(a.bn_shift + cow + h * 32 + mi * 16 + q * 4);
                        bsc[mi][q * 4 + 0] = s4.x; bsc[mi][q * 4 + 1] = s4.y; bsc[mi][q * 4 + 2] = s4.z; bsc[mi][q * 4 + 3] = s4.w;
                        bsh[mi][q * 4 + 0] = t4.x; bsh[mi][q * 4 + 1] = t4.y; bsh[mi][q * 4 + 2] = t4.z; bsh[mi][q * 4 + 3] = t4.w;
                    }
            }
#pragma unroll
            for (int pj = 0; pj < 4; ++pj) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    unsigned pk[8];
                    if (has_bn) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float v0 = acc[mi][pj][2 * e], v1 = acc[mi][pj][2 * e + 1];
                            if (a.act == 1) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                            pk[e] = pack_bf16x2(fmaf(v0, bsc[mi][2 * e], bsh[mi][2 * e]), fmaf(v1, bsc[mi][2 * e + 1], bsh[mi][2 * e + 1]));
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            pk[e] = pack_bf16x2(acc[mi][pj][2 * e], acc[mi][pj][2 * e + 1]);
                            if (a.act == 1)
                                pk[e] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pk[e]), s16x2{0, 0}));
                        }
                    }
                    const int s0 = h * 4 + mi * 2;
                    *(uint4*)(tb16 + px * 128 + ((s0 ^ (px & 7)) * 16)) = uint4{pk[0], pk[1], pk[2], pk[3]};
                    *(uint4*)(tb16 + px * 128 + (((s0 + 1) ^ (px & 7)) * 16)) = uint4{pk[4], pk[5], pk[6], pk[7]};
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int sy = ty0 + wpx * 4 + pj;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = i * 8 + rr;
                    const uint4 o = *(const uint4*)(tb16 + row * 128 + ((cc ^ (row & 7)) * 16));
                    const int sx = tx0 + row;
                    if (sy < Hs && sx < Ws) {
                        const size_t oidx = (((size_t)n * Hout + (sy * so + ro)) * Wout + (sx * so + cof)) * CoutPad + co8;
                        *(uint4*)((unsigned short*)a.out + oidx) = o;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            IDC_STAMP(3);
#ifdef IDC_TIMING
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            IDC_STAMP(4);
#endif
            return;
        }
        // transposed fp32 path: LeakyReLU + fused head (conv10_2), per-image shift (global hints), fp32 partial sums
        {
        const bool fh = fuse_head;
#pragma unroll
        for (int pj = 0; pj < 4; ++pj) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int slot = h * 8 + mi * 4 + q;
                    *(f32x4*)(tb + px * 64 + ((slot ^ (px & 7)) * 4)) =
                        f32x4{acc[mi][pj][q * 4 + 0], acc[mi][pj][q * 4 + 1], acc[mi][pj][q * 4 + 2], acc[mi][pj][q * 4 + 3]};
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same-wave LDS ops are in order: the row tile is complete
            const int sy = ty0 + wpx * 4 + pj;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 8 + rr;
                const f32x4 x0 = *(const f32x4*)(tb + row * 64 + (((2 * cc) ^ (row & 7)) * 4));
                const f32x4 x1 = *(const f32x4*)(tb + row * 64 + (((2 * cc + 1) ^ (row & 7)) * 4));
                float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                const int sx = tx0 + row;
                const bool inside = sy < Hs && sx < Ws;
                const size_t oidx = (((size_t)n * Hout + (sy * so + ro)) * Wout + (sx * so + cof)) * CoutPad + co8;
                if (a.resid != nullptr && !resid_in_acc && inside) {   // fp32 partial sums (313 head hyper-column)
                    const float4 r0 = *(const float4*)((const float*)a.resid + oidx);
                    const float4 r1 = *(const float4*)((const float*)a.resid + oidx + 4);
                    v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
                    v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
                }
                if (a.act == 1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                } else if (a.act == 2) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
                }
                if (has_bn) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], cs[e], ct[e]);
                }
                if (a.img_shift != nullptr) {            // global hints: per-image vector after the BN affine
                    const float4 g0 = *(const float4*)(a.img_shift + (size_t)n * CoutPad + co8);
                    const float4 g1 = *(const float4*)(a.img_shift + (size_t)n * CoutPad + co8 + 4);
                    v[0] += g0.x; v[1] += g0.y; v[2] += g0.z; v[3] += g0.w;
                    v[4] += g1.x; v[5] += g1.y; v[6] += g1.z; v[7] += g1.w;
                }
                if (fh) {
                    // model_out (1x1, 128 -> 2): 8 couts per lane, the pixel's other 56 in the 7 neighbour lanes
                    float s0 = 0.f, s1 = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { s0 = fmaf(v[e], hw0[e], s0); s1 = fmaf(v[e], hw1[e], s1); }
#pragma unroll
                    for (int m = 1; m <= 4; m <<= 1) { s0 += __shfl_xor(s0, m, 64); s1 += __shfl_xor(s1, m, 64); }
                    if (cc == 0) *(float2*)(part + ((wave * 4 + pj) * 32 + row) * 2) = float2{s0, s1};
                } else if (inside) {
                    uint4 o;
                    o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
                    o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
                    *(uint4*)((unsigned short*)a.out + oidx) = o;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads retired before the tile is rewritten
        }
        }
        if (fuse_head) {
            // the two cout waves of a pixel row meet in LDS; wave wco == 0 finishes: lane (px, h) = channel h
            __syncthreads();
            if (wco == 0) {
                const float hb = a.head_b[h];
#pragma unroll
                for (int pj = 0; pj < 4; ++pj) {
                    const float p = part[((wave * 4 + pj) * 32 + px) * 2 + h] + part[(((wave + 1) * 4 + pj) * 32 + px) * 2 + h];
                    const int sy = ty0 + wpx * 4 + pj, sx = tx0 + px;
                    if (sy < Hs && sx < Ws)
                        a.head_out[(((size_t)n * 2 + h) * Hs + sy) * Ws + sx] = tanhf(p + hb) * a.head_mul;
                }
            }
        }
    }
    IDC_STAMP(3);
#ifdef IDC_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    IDC_STAMP(4);
#endif
}

static constexpr size_t conv_v2_lds_bytes_c(int wco, int wpx, int halo) {
    const int nt = wco * wpx * 64;
    const int hrows = (32 + 2 * halo) * (4 * wpx + 2 * halo);
    const int items = (hrows * kSlots + nt - 1) / nt;
    return (size_t)items * nt * kSlotBytes + 2 * (size_t)(64 * wco) * kRowBytes;
}

template <int WCO, int WPX, int HALO>
static hipError_t launch_conv_v2_t(const ConvArgs& a, hipStream_t s) {
    constexpr size_t lds = conv_v2_lds_bytes_c(WCO, WPX, HALO);
    const int nct = a.ncg / WCO;
    const long long blocks = (long long)a.tiles_x * a.tiles_y * a.N * nct * a.nphase;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    if (a.in2 != nullptr) return hipErrorInvalidConfiguration;      // fused shortcut launches are conv_ds_fused's
    if (a.zeros == nullptr) return hipErrorInvalidValue;            // out-of-image halo rows and trailing tile requests read the zero page
    hipLaunchKernelGGL((conv_igemm_v2<WCO, WPX, HALO>), dim3((unsigned)blocks), dim3(WCO * WPX * 64), lds, s, a);
    return hipGetLastError();
}

#define IDC_FOR_EACH_CONV_V2(X) X(4, 2, 0) X(4, 2, 1) X(4, 2, 2) X(2, 4, 0) X(2, 4, 1) X(2, 4, 2) X(2, 2, 0) X(2, 2, 1) X(2, 2, 2)

__global__ void conv_ds_fused(const ConvArgs a);

hipError_t init_kernels_v2() {
    hipError_t e;
#define X(WCO, WPX, HL)                                                                                     \
    e = hipFuncSetAttribute((const void*)conv_igemm_v2<WCO, WPX, HL>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                            (int)conv_v2_lds_bytes_c(WCO, WPX, HL));                                        \
    if (e != hipSuccess) return e;
    IDC_FOR_EACH_CONV_V2(X)
#undef X
    // the two fused kernels use more than the default 64 KiB of dynamic LDS (set per device: this runs for every handle)
    return hipFuncSetAttribute((const void*)conv_ds_fused, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// v2 tile = 32 sites wide, 4*wpx rows; cfg.wm = WCO (x64 couts), cfg.wp = WPX.
hipError_t launch_conv_v2(ConvConfig cfg, int halo, const ConvArgs& a, hipStream_t s) {
#define X(WCO, WPX, HL) \
    if (cfg.wm == WCO && cfg.wp == WPX && halo == HL) return launch_conv_v2_t<WCO, WPX, HL>(a, s);
    IDC_FOR_EACH_CONV_V2(X)
#undef X
    return hipErrorInvalidConfiguration;
}


// ================================================================================================
// conv_ds_fused -- ConvTranspose2d 4x4 s2 (model8up / model9up / model10up) and the 3x3 shortcut conv it is summed
// with (model3short8 / model2short9 / model1short10; model.py:156,170,172) in ONE K loop: the shortcut's 128-channel
// partial sums never go to HBM (537 MB written + read back per forward at 256^2, N = 32, for model10 alone).
//   * workgroup = 64 x 8 OUTPUT pixels x 128 couts: 2 cout waves x 4 PHASE waves.  Wave (wco, ph) owns the 32 x 4 sites
//     whose output pixel is (2y + ro, 2x + cof) -- so the four deconv phases of a site tile share one workgroup, and the
//     skip tensor's (10 x 66)-pixel halo is fetched once instead of once per phase launch;
//   * S part (shortcut): K = 9 taps x Cs.  The halo tile is stored de-interleaved by x parity (LDS row = y*66 +
//     (x&1)*33 + x/2): a phase wave reads pixels of one parity, i.e. 32 consecutive rows -> the same conflict-free
//     ds_read_b128 pattern as conv_igemm_v2.  Weight tiles (128 couts) are shared, 3-slot LDS-DMA ring (barrier one step early);
//   * D part (deconv): K = 4 taps x Cd, taps and weight tiles depend on the phase, so every wave streams its own
//     8 KiB tile (64 couts x 64 cin) through a wave-private 2-deep ring, no workgroup barrier inside a halo chunk;
//   * epilogue = the bf16-transpose one (bias in the accumulators, ReLU on packed pairs), per-wave output phase.
// LDS: S part 90 KiB halo + 48 KiB ring (3 slots); D part 32 KiB halo + 128 KiB rings = 160 KiB (the two parts reuse the space,
// one drained hand-over in between).
// ================================================================================================
__global__ __launch_bounds__(512, 2) void conv_ds_fused(const ConvArgs a) {
    constexpr int NT = 512;
    constexpr int SW = 66, SROWS = 10 * SW, S_ITEMS = (SROWS * kSlots + NT - 1) / NT, S_HALO_BYTES = S_ITEMS * NT * kSlotBytes;
    constexpr int DW = 34, DROWS = 6 * DW, D_ITEMS = (DROWS * kSlots + NT - 1) / NT, D_HALO_BYTES = D_ITEMS * NT * kSlotBytes;
    constexpr int S_WB = 2 * kWBlockBytes, D_WB = kWBlockBytes;
    static_assert(S_HALO_BYTES + 3 * S_WB <= 160 * 1024 && D_HALO_BYTES + 16 * D_WB <= 160 * 1024, "LDS budget");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const halo = smem;
    char* const ringS = smem + S_HALO_BYTES;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave & 1, ph = wave >> 1;
    char* const ringD = smem + D_HALO_BYTES + wave * 2 * D_WB;
    const int px = lane & 31, h = lane >> 5;
    const int Hs = a.Hs, Ws = a.Ws;                            // deconv input (= site) resolution; output is 2x
    const int ntx = (Ws + 31) >> 5, nty = (Hs + 3) >> 2, nct = a.ncg >> 1;
    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int ct = b % nct; b /= nct;
    const int txi = b % ntx; b /= ntx;
    const int tyi = b % nty;
    const int n = b / nty;
    const int y0 = tyi * 4, x0 = txi * 32;
    const int ro = a.ro[ph], cof = a.co[ph];
    const int nkc = a.nkc, nkc2 = a.nkc2, ncg = a.ncg;
    const int pixD = nkc * kRowBytes, pixS = nkc2 * kRowBytes;
    const char* const imgD = (const char*)a.in + (size_t)n * Hs * Ws * pixD;
    const char* const imgS = (const char*)a.in2 + (size_t)n * (4 * (size_t)Hs * Ws) * pixS;
    const int cg0 = ct * 2;
    IDC_STAMP(0);

    f32x16 acc[2][4];
    {
        const float* const bp = a.bias + (cg0 + wco) * kCoutGroup + h * 32;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x16 b16;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *(const float4*)(bp + i * 16 + q * 4);
                b16[q * 4 + 0] = bq.x; b16[q * 4 + 1] = bq.y; b16[q * 4 + 2] = bq.z; b16[q * 4 + 3] = bq.w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = b16;
        }
    }

    u32x4 hreg[S_ITEMS];
    auto load_halo_S = [&](int kc2) {
        // (the item -> address arithmetic is recomputed per chunk on purpose: hoisted out of the chunk loop its 64-bit
        //  addresses cost hipcc two spilled register pairs at the 256-VGPR cap)
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
            const int off = (Y * (2 * Ws) + X) * pixS + ((sig ^ swz2(hr)) + kc2 * kSlots) * kSlotBytes;
            hreg[j] = *(const u32x4*)(inside ? imgS + off : (const char*)a.zeros);   // (zero page: no select on the loaded value)
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
            const int off = (Y * Ws + X) * pixD + ((sig ^ swz2(hr)) + kc * kSlots) * kSlotBytes;
            hreg[j] = *(const u32x4*)(inside ? imgD + off : (const char*)a.zeros);
        }
    };
    auto dma_S = [&](int tap, int kc2, int buf) {              // 128 couts x 64 cin, shared: every wave brings 2 KiB
        const char* src = (const char*)a.wgt2 + (((size_t)tap * nkc2 + kc2) * ncg + cg0) * kWBlockBytes + (size_t)tid * kSlotBytes;
        char* dst = ringS + buf * S_WB + wave * 64 * kSlotBytes;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)j * NT * kSlotBytes),
                                             (__attribute__((address_space(3))) void*)(dst + j * NT * kSlotBytes), 16, 0, 0);
    };
    auto dma_D = [&](int tw, int kc, int buf) {                // this wave's 64 couts x 64 cin of its phase's tap
        int lane_ = lane;
        asm volatile("" : "+v"(lane_));                        // (keeps the per-tap 64-bit addresses out of the loop-invariant set)
        const char* src = (const char*)a.wgt + (((size_t)tw * nkc + kc) * ncg + cg0 + wco) * kWBlockBytes + (size_t)lane_ * kSlotBytes;
        char* dst = ringD + buf * D_WB;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * 64 * kSlotBytes),
                                             (__attribute__((address_space(3))) void*)(dst + j * 64 * kSlotBytes), 16, 0, 0);
    };
    // one K step of 64 channels: 4 k16 steps x 8 MFMAs, fragments of step kk+1 in flight under step kk (as conv_igemm_v2)
    const int wslot0 = (h ^ swz2(px)) * kSlotBytes;
    u32x4 wfA[2], xfA[4], wfB[2], xfB[4];
    auto read_frags = [&](const char* const wcur, const int wrow_byte, const int (&xaddr)[4], int kk, u32x4 (&wf)[2], u32x4 (&xf)[4]) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
            wf[mi] = *(const u32x4*)(wcur + ((wrow_byte + mi * 32 * kRowBytes + wslot0) ^ (kk * 2 * kSlotBytes)));
#pragma unroll
        for (int pj = 0; pj < 4; ++pj)
            xf[pj] = *(const u32x4*)(halo + (xaddr[pj] ^ (kk * 2 * kSlotBytes)));
    };
    auto mma8 = [&](const u32x4 (&wf)[2], const u32x4 (&xf)[4]) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int pj = 0; pj < 4; ++pj)
                acc[mi][pj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[mi]),
                                                                      __builtin_bit_cast(bf16x8, xf[pj]),
                                                                      acc[mi][pj], 0, 0, 0);
    };
#define IDC_STAGE_INTERLEAVE()                                                        \
    _Pragma("unroll") for (int q_ = 0; q_ < 6; ++q_) {                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                            \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                            \
    }                                                                                 \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);

    // ---------------------------------------------------------------- S part: 3x3 conv of the skip tensor
    // Three weight slots, as conv_igemm_v2's 8-wave K loop: the barrier at the top of step s publishes tile s+1 (requested
    // a step earlier), tile s+2 is requested behind it, and the first fragments of step s+1 are read under the last 8
    // MFMAs of step s.  Requests past the last tile re-read the zero page (branch-free tail).
    load_halo_S(0);
    dma_S(0, 0, 0);
    IDC_STAMP_FINE(5);
    dma_S(1, 0, 1);
    int rt = 2, rkc = 0;                                       // request cursor: (tap, chunk) of tile s+2
    auto dma_S_req = [&](int slot_off) {
        const bool real = rkc < nkc2;
        const char* src = real ? (const char*)a.wgt2 + (((size_t)rt * nkc2 + rkc) * ncg + cg0) * kWBlockBytes + (size_t)tid * kSlotBytes
                               : (const char*)a.zeros + (tid & 15) * kSlotBytes;
        const size_t jstep = real ? (size_t)NT * kSlotBytes : 0;
        char* dst = ringS + slot_off + wave * 64 * kSlotBytes;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)j * jstep),
                                             (__attribute__((address_space(3))) void*)(dst + j * NT * kSlotBytes), 16, 0, 0);
    };
    int xs[4];
    auto set_xs = [&](int t) {
        const int ky = t / 3, kx = t - ky * 3;                 // 0..2 (= tap offset + 1)
        const int c = cof + kx, par = c & 1, sh = c >> 1;      // output x = 2*xs + cof reads skip x + kx - 1: halo col 2*xs + c
#pragma unroll
        for (int pj = 0; pj < 4; ++pj) {
            const int xr = (2 * pj + ro + ky) * SW + par * 33 + px + sh;
            xs[pj] = xr * kRowBytes + ((h ^ swz2(xr)) * kSlotBytes);
        }
    };
    const int wrowS = (wco * 64 + px) * kRowBytes;
    int off_cur = 0, off_next = S_WB, off_free = 2 * S_WB;
#pragma unroll
    for (int j = 0; j < S_ITEMS; ++j) *(u32x4*)(halo + (tid + j * NT) * kSlotBytes) = hreg[j];
    set_xs(0);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");           // my pieces of tile 0 (tile 1 may still be in flight)
    __syncthreads();                                           // halo chunk 0 and tile 0 are visible
    IDC_STAMP(1);
    read_frags(ringS, wrowS, xs, 0, wfA, xfA);
    for (int kc2 = 0; kc2 < nkc2; ++kc2) {
        const bool last_kc = kc2 + 1 == nkc2;
        auto tap_body = [&](int t, auto last_tag) {
            constexpr bool LAST = decltype(last_tag)::value;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // my pieces of the next step's tile
            __syncthreads();                                    // ... everybody's: published; everybody left slot off_free
            dma_S_req(off_free);
            if constexpr (LAST) {
                if (!last_kc) load_halo_S(kc2 + 1);
                else load_halo_D(0);                            // the deconv input's first chunk: rows wait in registers
            }
            __builtin_amdgcn_sched_barrier(0);
            const char* const wcur = ringS + off_cur;
            read_frags(wcur, wrowS, xs, 1, wfB, xfB);
            mma8(wfA, xfA);
            IDC_STAGE_INTERLEAVE()
            read_frags(wcur, wrowS, xs, 2, wfA, xfA);
            mma8(wfB, xfB);
            IDC_STAGE_INTERLEAVE()
            read_frags(wcur, wrowS, xs, 3, wfB, xfB);
            mma8(wfA, xfA);
            IDC_STAGE_INTERLEAVE()
            set_xs(LAST ? 0 : t + 1);
            if (++rt == 9) { rt = 0; ++rkc; }
            read_frags(ringS + off_next, wrowS, xs, 0, wfA, xfA);
            mma8(wfB, xfB);
            IDC_STAGE_INTERLEAVE()
            if constexpr (LAST) {
                if (!last_kc) {
                    __syncthreads();                            // everybody is done with halo chunk kc2
#pragma unroll
                    for (int j = 0; j < S_ITEMS; ++j) *(u32x4*)(halo + (tid + j * NT) * kSlotBytes) = hreg[j];
                    __syncthreads();
#pragma unroll
                    for (int pj = 0; pj < 4; ++pj) xfA[pj] = *(const u32x4*)(halo + xs[pj]);
                }
            }
            const int o_ = off_cur; off_cur = off_next; off_next = off_free; off_free = o_;
        };
        for (int t = 0; t < 8; ++t) tap_body(t, std::false_type{});
        tap_body(8, std::true_type{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the trailing zero-page requests target LDS the D part reuses
    // ---------------------------------------------------------------- hand-over: the D part reuses the whole LDS
    IDC_STAMP(8);
    const int* const tdy = a.dy + ph * 9;
    const int* const tdx = a.dx + ph * 9;
    const int* const ttw = a.tw + ph * 9;
    // the phase's 2x2 taps as a table in lanes 0..3 (halo row offset, weight tap), read back with v_readlane: no scalar
    // loads inside the loop (hipcc drains lgkmcnt to 0 for them, which would also wait for the prefetched fragments)
    int v_xoff = 0, v_tw = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t)
        if (lane == t) { v_xoff = (1 + tdy[t]) * DW + 1 + tdx[t]; v_tw = ttw[t]; }
    __syncthreads();                                           // every wave left the S halo and ring
#pragma unroll
    for (int j = 0; j < D_ITEMS; ++j) *(u32x4*)(halo + (tid + j * NT) * kSlotBytes) = hreg[j];
    dma_D(__builtin_amdgcn_readlane(v_tw, 0), 0, 0);
    dma_D(__builtin_amdgcn_readlane(v_tw, 1), 0, 1);
    // ---------------------------------------------------------------- D part: the wave's deconv phase, 2x2 taps.
    // Weight tiles are wave-private (own ring, own vmcnt), so a step needs no workgroup barrier: the two waves of a SIMD
    // drift apart and fill each other's bubbles; only the halo chunk change synchronises.  Step s = (kc, t) = (s >> 2,
    // s & 3) uses ring slot s & 1; tile s+2 is requested when the last fragments of tile s have been consumed, and the
    // first fragments of step s+1 are read under the last 8 MFMAs of step s.  The tail is branch-free (a join would make
    // hipcc wait for the prefetched fragments): past the last tile the request re-reads 1 KiB of the zero page, and the
    // halo half of a prefetch that crosses a chunk change is simply read again after the change.
    const int wrowD = px * kRowBytes;
    const int nsteps = 4 * nkc;
    int xa[4];
    {
        const int xo = __builtin_amdgcn_readlane(v_xoff, 0);
#pragma unroll
        for (int pj = 0; pj < 4; ++pj) {
            const int xr = pj * DW + px + xo;
            xa[pj] = xr * kRowBytes + ((h ^ swz2(xr)) * kSlotBytes);
        }
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");           // tile 0 landed (tile 1 may still be in flight)
    __syncthreads();                                           // halo chunk 0 visible
    IDC_STAMP(9);
    read_frags(ringD, wrowD, xa, 0, wfA, xfA);
    for (int st = 0; st < nsteps; ++st) {
        const int t = st & 3, kc = st >> 2;
        const char* const wcur = ringD + (st & 1) * D_WB;
        const char* const wnext = ringD + ((st + 1) & 1) * D_WB;
        const bool swap = t == 3 && st + 1 < nsteps;
        if (swap) load_halo_D(kc + 1);                         // next chunk's rows wait in registers
        read_frags(wcur, wrowD, xa, 1, wfB, xfB);
        mma8(wfA, xfA);
        IDC_STAGE_INTERLEAVE()
        read_frags(wcur, wrowD, xa, 2, wfA, xfA);
        mma8(wfB, xfB);
        IDC_STAGE_INTERLEAVE()
        read_frags(wcur, wrowD, xa, 3, wfB, xfB);
        mma8(wfA, xfA);
        IDC_STAGE_INTERLEAVE()
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // tile st+1 landed; tile st's fragment reads are all back
        {
            const int s2 = st + 2;
            const bool real = s2 < nsteps;
            const int tw2 = __builtin_amdgcn_readlane(v_tw, s2 & 3);
            int lane_ = lane;
            asm volatile("" : "+v"(lane_));
            const char* src = real ? (const char*)a.wgt + (((size_t)tw2 * nkc + (s2 >> 2)) * ncg + cg0 + wco) * kWBlockBytes + (size_t)lane_ * kSlotBytes
                                   : (const char*)a.zeros + (lane_ & 15) * kSlotBytes;
            const int jstep = real ? 64 * kSlotBytes : 0;
            char* dst = ringD + (st & 1) * D_WB;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)j * jstep),
                                                 (__attribute__((address_space(3))) void*)(dst + j * 64 * kSlotBytes), 16, 0, 0);
        }
        {
            const int xo = __builtin_amdgcn_readlane(v_xoff, (st + 1) & 3);
#pragma unroll
            for (int pj = 0; pj < 4; ++pj) {
                const int xr = pj * DW + px + xo;
                xa[pj] = xr * kRowBytes + ((h ^ swz2(xr)) * kSlotBytes);
            }
        }
        read_frags(wnext, wrowD, xa, 0, wfA, xfA);
        mma8(wfB, xfB);
        IDC_STAGE_INTERLEAVE()
        if (swap) {
            __syncthreads();                                   // every wave is done with halo chunk kc
#pragma unroll
            for (int j = 0; j < D_ITEMS; ++j) *(u32x4*)(halo + (tid + j * NT) * kSlotBytes) = hreg[j];
            __syncthreads();
#pragma unroll
            for (int pj = 0; pj < 4; ++pj) xfA[pj] = *(const u32x4*)(halo + xa[pj]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (the zero-page requests of the last two steps target this ring)
#undef IDC_STAGE_INTERLEAVE
    // ---------------------------------------------------------------- epilogue: (ReLU,) round, transpose, whole-line stores
    IDC_STAMP(2);
    __syncthreads();
    char* const tb16 = smem + wave * 4096;
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    const int rr = lane >> 3, cc = lane & 7;
    const int CoutPad = ncg * kCoutGroup;
    const int co8 = (cg0 + wco) * kCoutGroup + cc * 8;
    const int Wout = 2 * Ws, Hout = 2 * Hs;
#pragma unroll
    for (int pj = 0; pj < 4; ++pj) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            unsigned pk[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                pk[e] = pack_bf16x2(acc[mi][pj][2 * e], acc[mi][pj][2 * e + 1]);
                if (a.act == 1)
                    pk[e] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pk[e]), s16x2{0, 0}));
            }
            const int s0 = h * 4 + mi * 2;
            *(uint4*)(tb16 + px * 128 + ((s0 ^ (px & 7)) * 16)) = uint4{pk[0], pk[1], pk[2], pk[3]};
            *(uint4*)(tb16 + px * 128 + (((s0 + 1) ^ (px & 7)) * 16)) = uint4{pk[4], pk[5], pk[6], pk[7]};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int sy = y0 + pj;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 8 + rr;
            const uint4 o = *(const uint4*)(tb16 + row * 128 + ((cc ^ (row & 7)) * 16));
            const int sx = x0 + row;
            if (sy < Hs && sx < Ws)
                *(uint4*)((unsigned short*)a.out + (((size_t)n * Hout + (2 * sy + ro)) * Wout + (2 * sx + cof)) * CoutPad + co8) = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    IDC_STAMP(3);
#ifdef IDC_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    IDC_STAMP(4);
#endif
}

// deconv 4x4 s2 + its 3x3 shortcut conv in one launch: bf16, Cout a multiple of 128, (ReLU | none), no BN
hipError_t launch_conv_ds(const ConvArgs& a, hipStream_t s) {
    if (a.in2 == nullptr || a.wgt2 == nullptr || a.zeros == nullptr || a.nphase != 4 || a.so != 2 || a.si != 1 || (a.ncg & 1) || a.out_f32 ||
        a.bn_scale != nullptr || a.act == 2 || a.img_shift != nullptr || a.resid != nullptr || a.head_w != nullptr)
        return hipErrorInvalidConfiguration;
    const long long blocks = (long long)((a.Ws + 31) / 32) * ((a.Hs + 3) / 4) * a.N * (a.ncg / 2);
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(conv_ds_fused, dim3((unsigned)blocks), dim3(512), 160 * 1024, s, a);
    return hipGetLastError();
}



#else   // default library: the 16x16x32 twins only (idc_v2m.hip, idc_dsm.hip); the engine never plans these launches (idc_engine.hip, kAbPartners)
hipError_t init_kernels_v2() { return hipSuccess; }
hipError_t launch_conv_v2(ConvConfig, int, const ConvArgs&, hipStream_t) { return hipErrorInvalidConfiguration; }
hipError_t launch_conv_ds(const ConvArgs&, hipStream_t) { return hipErrorInvalidConfiguration; }
#endif
}  // namespace idc
