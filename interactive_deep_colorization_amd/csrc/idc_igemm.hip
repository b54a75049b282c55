// idc_igemm.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the Local-Hints forward pass.
//
// conv_igemm<T, WM, WP, HALO>: im2col-free implicit GEMM for every conv / deconv of
// models/pytorch/model.py:13-109, D[cout][pixel] += W_tap[cout][cin] * X[pixel + tap][cin].
//   * one workgroup = (16 x 4*WP) output sites of ONE image x 64*WM output channels;
//     one wave = 64 couts x 4 spatial rows of 16 pixels = 4x4 MFMA 16x16 accumulator tiles;
//   * the input tile WITH ITS HALO is staged once per 128-byte channel chunk into LDS and reused
//     by all taps (9x fewer L2->LDS bytes than per-tap gathers; zero padding = zero-filled rows);
//   * weights arrive as pre-swizzled 8 KiB LDS images (idc_layout.h): a straight 16-B/lane copy,
//     register-prefetched one tap ahead (issue-early / write-late) into a 2-deep LDS ring,
//     one barrier per tap;
//   * MFMA operands are 16-byte ds_read_b128 fragments, conflict-free under the row&7 XOR
//     swizzle; bf16 uses v_mfma_f32_16x16x32_bf16, fp32 uses 4x v_mfma_f32_16x16x4_f32 (exact
//     fp32 = an fmaf chain) on the same 16-byte fragments;
//   * fused epilogue: +bias, +fp32 shortcut sum, ReLU/LeakyReLU, eval-BN affine AFTER the
//     activation (model.py:13-17 order), 32/64-byte stores of 16 consecutive channels per lane.
#include <stdlib.h>
#include <type_traits>

#include "idc_kernels.h"

#include "idc_layout.h"

#include "idc_common.hip.h"

namespace idc {

#ifdef IDC_TIMING
__device__ long long* g_idc_dbg;
#endif

template <typename T, int WM, int WP, int HALO>
__global__ __launch_bounds__(WM* WP * 64) void conv_igemm(const ConvArgs a) {
    constexpr int NT = WM * WP * 64;
    constexpr int TW = 16, TH = 4 * WP;
    constexpr int HWP = TW + 2 * HALO, HHP = TH + 2 * HALO, HROWS = HWP * HHP;
    constexpr int BN = 64 * WM;
    constexpr int W_BYTES = BN * kRowBytes;
    constexpr int N_HITEMS = (HROWS * kSlots + NT - 1) / NT;
    constexpr int HALO_BYTES = N_HITEMS * NT * kSlotBytes;       // >= HROWS*128: every thread always writes
    constexpr int N_WITEMS = (W_BYTES / kSlotBytes) / NT;        // = 8 / WP
    static_assert((W_BYTES / kSlotBytes) % NT == 0, "weight tile must split evenly");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const halo = smem;
    char* const wbuf = smem + HALO_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wp = wave / WM;
    const int px = lane & 15, g = lane >> 4;
    IDC_STAMP(0);

    // ---- which tile -------------------------------------------------------------------------
    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int nct = a.ncg / WM;
    const int ksplit = a.ksplit > 1 ? a.ksplit : 1;
    const int sid = b % ksplit; b /= ksplit;                      // split-K slice (fastest: slices of a tile share its halo pixels)
    const int txi = b % a.tiles_x; b /= a.tiles_x;
    const int tyi = b % a.tiles_y; b /= a.tiles_y;
    const int n = b % a.N; b /= a.N;
    const int ct = b % nct;
    const int phase = b / nct;
    const int ty0 = tyi * TH, tx0 = txi * TW;
    const int Hs = a.Hs, Ws = a.Ws, si = a.si;
    const int Win = Ws * si;
    const int pix_bytes = a.nkc * kRowBytes;                      // Cin * sizeof(T)
    const char* const in_img = (const char*)a.in + (size_t)n * (size_t)(Hs * si) * Win * pix_bytes;

    // ---- halo staging plan: item = (halo row, physical slot); fixed for the whole K loop -----
    int hoff[N_HITEMS];
#pragma unroll
    for (int j = 0; j < N_HITEMS; ++j) {
        const int item = tid + j * NT;
        const int hr = item >> 3, sig = item & 7;
        const int hy = hr / HWP, hx = hr - hy * HWP;
        const int sy = ty0 - HALO + hy, sx = tx0 - HALO + hx;
        const bool inside = (unsigned)sy < (unsigned)Hs && (unsigned)sx < (unsigned)Ws;
        const int s = sig ^ swz(hr);                              // logical slot stored at `sig`
        // rows past the tile (item >= HROWS*8) land in the padding of the LDS halo area: store zeros
        hoff[j] = (inside && item < HROWS * kSlots) ? ((sy * si) * Win + sx * si) * pix_bytes + s * kSlotBytes : -1;
    }

    // ---- weight tile source ------------------------------------------------------------------
    const char* const wbase = (const char*)a.wgt + (size_t)(ct * WM) * kWBlockBytes + (size_t)tid * kSlotBytes;
    const size_t w_kc_stride = (size_t)a.ncg * kWBlockBytes;      // next cin chunk
    const size_t w_tap_stride = w_kc_stride * a.nkc;              // next tap
    const int* const tap_dy = a.dy + phase * 9;
    const int* const tap_dx = a.dx + phase * 9;
    const int* const tap_tw = a.tw + phase * 9;
    const int ntaps = a.ntaps;
    const int kc0 = ksplit > 1 ? sid * a.kc_per : 0;
    const int nkc = ksplit > 1 ? (kc0 + a.kc_per < a.nkc ? kc0 + a.kc_per : a.nkc) : a.nkc;   // this slice: chunks [kc0, nkc)

    // fp32 path: the MFMA is an exact sequential fmaf chain, so one accumulator over K = 9*Cin
    // (up to 4608 terms) would carry ~4x the rounding noise of a blocked sum.  Accumulate each
    // 128-byte channel chunk (<= 288 terms) separately and add it to the running total.
    constexpr bool kBlockedAcc = sizeof(T) == 4;
    f32x4 acc[4][4], tot[kBlockedAcc ? 4 : 1][kBlockedAcc ? 4 : 1];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (kBlockedAcc) tot[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

    u32x4 wreg[N_WITEMS];
    {
        const char* src = wbase + (size_t)tap_tw[0] * w_tap_stride + (size_t)kc0 * w_kc_stride;
#pragma unroll
        for (int j = 0; j < N_WITEMS; ++j) wreg[j] = *(const u32x4*)(src + (size_t)j * NT * kSlotBytes);
    }

    const int wrow_byte = (wm * 64 + px) * kRowBytes;             // + ci*16 rows
    const int wsw = px & 7;                                       // swz of rows wm*64+ci*16+px
    int cur = 0;

    // Halo rows of chunk 0 go to registers now; chunk kc+1 is fetched under the last tap of chunk kc
    // (issue-early / write-late), so HBM/L2 latency hides behind 32 MFMAs per wave.
    u32x4 hreg[N_HITEMS];
    auto load_halo = [&](int kc) {
        if constexpr (HALO == 0) {
            if (a.pk_L != nullptr) {               // conv1_1: build the im2col rows from the L / ab / mask planes
                // step 1 (first call only): the tile's (TH+2)x(TW+2) input patch, normalised once per pixel, goes to
                // LDS with coalesced plane reads (zero outside the image = conv1_1's own zero padding)
                constexpr int PW = TW + 2, PH = TH + 2;
                float4* const patch = (float4*)(wbuf + 2 * W_BYTES);
                if (kc == kc0) {
                    const size_t hw = (size_t)Hs * Ws;
                    const float* const pL = a.pk_L + (size_t)n * hw;
                    const float* const pA = a.pk_ab + (size_t)n * 2 * hw;
                    const float* const pM = a.pk_mask + (size_t)n * hw;
                    for (int idx = tid; idx < PW * PH; idx += NT) {
                        const int py = idx / PW, pxx = idx - py * PW;
                        const int yy = ty0 - 1 + py, xx = tx0 - 1 + pxx;
                        float4 c = float4{0.f, 0.f, 0.f, 0.f};
                        if ((unsigned)yy < (unsigned)Hs && (unsigned)xx < (unsigned)Ws) {
                            const size_t p = (size_t)yy * Ws + xx;
                            c = float4{pL[p] / a.pk_ldiv, pA[p] / a.pk_abdiv, pA[hw + p] / a.pk_abdiv, pM[p] * a.pk_mmul - a.pk_mcent};
                        }
                        patch[idx] = c;
                    }
                    __syncthreads();
                }
                // step 2: this thread's 16-byte pieces of the im2col rows (K index = tap*4 + channel)
#pragma unroll
                for (int j = 0; j < N_HITEMS; ++j) {
                    const int item = tid + j * NT;
                    const int hr = item >> 3, sl = (item & 7) ^ swz(hr);          // logical slot of this piece
                    const int hy = hr / HWP, hx = hr - hy * HWP;
                    const bool live = ty0 + hy < Hs && tx0 + hx < Ws && item < HROWS * kSlots;
                    if constexpr (sizeof(T) == 2) {                               // 8 bf16 = taps 2*sl, 2*sl+1
                        const int t0 = sl * 2, t1 = sl * 2 + 1;
                        const float4 c0 = (live && t0 < 9) ? patch[(hy + t0 / 3) * PW + hx + t0 % 3] : float4{0.f, 0.f, 0.f, 0.f};
                        const float4 c1 = (live && t1 < 9) ? patch[(hy + t1 / 3) * PW + hx + t1 % 3] : float4{0.f, 0.f, 0.f, 0.f};
                        hreg[j] = u32x4{pack_bf16x2(c0.x, c0.y), pack_bf16x2(c0.z, c0.w), pack_bf16x2(c1.x, c1.y), pack_bf16x2(c1.z, c1.w)};
                    } else {                                                      // 4 fp32 = tap kc*8 + sl
                        const int t0 = kc * 8 + sl;
                        const float4 c0 = (live && t0 < 9) ? patch[(hy + t0 / 3) * PW + hx + t0 % 3] : float4{0.f, 0.f, 0.f, 0.f};
                        hreg[j] = u32x4{__float_as_uint(c0.x), __float_as_uint(c0.y), __float_as_uint(c0.z), __float_as_uint(c0.w)};
                    }
                }
                return;
            }
        }
#pragma unroll
        for (int j = 0; j < N_HITEMS; ++j) {
            const int off = hoff[j];                              // zero padding / rows past the tile read the zero page
            hreg[j] = *(const u32x4*)(off >= 0 ? in_img + off + kc * kRowBytes : (const char*)a.zeros);
        }
    };
    load_halo(kc0);
    bool first_ = true;

    for (int kc = kc0; kc < nkc; ++kc) {
        __syncthreads();                       // every wave is done reading the previous halo
#pragma unroll
        for (int j = 0; j < N_HITEMS; ++j) *(u32x4*)(halo + (tid + j * NT) * kSlotBytes) = hreg[j];
        for (int t = 0; t < ntaps; ++t) {
            char* const wcur = wbuf + cur * W_BYTES;
#pragma unroll
            for (int j = 0; j < N_WITEMS; ++j)
                *(u32x4*)(wcur + (tid + j * NT) * kSlotBytes) = wreg[j];
            __syncthreads();
            if (first_) { IDC_STAMP(1); first_ = false; }
            // prefetch the next (tap, chunk) weight tile; it lands in registers under the MFMAs
            {
                int t2 = t + 1, kc2 = kc;
                if (t2 == ntaps) { t2 = 0; kc2 = kc + 1; }
                if (kc2 == nkc) { t2 = t; kc2 = kc; }     // last step: harmless reload, keeps the loop branch-free
                const char* src = wbase + (size_t)tap_tw[t2] * w_tap_stride + (size_t)kc2 * w_kc_stride;
#pragma unroll
                for (int j = 0; j < N_WITEMS; ++j)
                    wreg[j] = *(const u32x4*)(src + (size_t)j * NT * kSlotBytes);
            }
            if (t == ntaps - 1 && kc + 1 < nkc) load_halo(kc + 1);
            // keep the prefetch loads ABOVE the MFMA cluster (hipcc otherwise sinks them below it to
            // save registers, which exposes the L2 latency at the next ds_write)
            __builtin_amdgcn_sched_barrier(0);
            const int dy = tap_dy[t], dx = tap_dx[t];
            int xrow[4];
#pragma unroll
            for (int pj = 0; pj < 4; ++pj)
                xrow[pj] = (wp * 4 + pj + HALO + dy) * HWP + (px + HALO + dx);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int slot = ks * 4 + g;
                u32x4 wf[4], xf[4];
#pragma unroll
                for (int ci = 0; ci < 4; ++ci)
                    wf[ci] = *(const u32x4*)(wcur + wrow_byte + ci * 16 * kRowBytes + ((slot ^ wsw) * kSlotBytes));
#pragma unroll
                for (int pj = 0; pj < 4; ++pj)
                    xf[pj] = *(const u32x4*)(halo + xrow[pj] * kRowBytes + ((slot ^ swz(xrow[pj])) * kSlotBytes));
#pragma unroll
                for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                    for (int pj = 0; pj < 4; ++pj) Mma<T>::run(acc[ci][pj], wf[ci], xf[pj]);
            }
            cur ^= 1;
        }
        if (kBlockedAcc) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    tot[i][j] += acc[i][j];
                    acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
        }
    }
    if (kBlockedAcc) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = tot[i][j];
    }

    IDC_STAMP(2);
    // ---- epilogue: lane owns couts co0..co0+15 of pixel px in each of its 4 rows ----------------
    const int CoutPad = a.ncg * kCoutGroup;
    const int co0 = (ct * WM + wm) * kCoutGroup + g * 16;
    if (ksplit > 1) {                          // raw fp32 slice sums; splitk_epilogue finishes the layer
        const int so = a.so, Wout = Ws * so, Hout = Hs * so;
        const int ro = a.ro[phase], cof = a.co[phase];
        float* const slab = a.partial + (size_t)sid * a.N * Hout * Wout * CoutPad;
#pragma unroll
        for (int pj = 0; pj < 4; ++pj) {
            const int sy = ty0 + wp * 4 + pj, sx = tx0 + px;
            if (sy < Hs && sx < Ws) {
                float* o = slab + (((size_t)n * Hout + (sy * so + ro)) * Wout + (sx * so + cof)) * CoutPad + co0;
#pragma unroll
                for (int ci = 0; ci < 4; ++ci)
                    *(float4*)(o + ci * 4) = float4{acc[ci][pj][0], acc[ci][pj][1], acc[ci][pj][2], acc[ci][pj][3]};
            }
        }
        IDC_STAMP(3);
#ifdef IDC_TIMING
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        IDC_STAMP(4);
#endif
        return;
    }
    const bool has_bn = a.bn_scale != nullptr;
    const int so = a.so, Wout = Ws * so, Hout = Hs * so;
    const int ro = a.ro[phase], cof = a.co[phase];
    float bias[16], bsc[16], bsh[16];
    load16(bias, a.bias + co0);
    if (has_bn) { load16(bsc, a.bn_scale + co0); load16(bsh, a.bn_shift + co0); }
#pragma unroll
    for (int pj = 0; pj < 4; ++pj) {
        const int sy = ty0 + wp * 4 + pj, sx = tx0 + px;
        if (sy < Hs && sx < Ws) {
            const size_t opix = ((size_t)n * Hout + (sy * so + ro)) * Wout + (sx * so + cof);
            float v[16];
#pragma unroll
            for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[ci * 4 + r] = acc[ci][pj][r];
            epilogue16<sizeof(T) == 2>(a, v, opix * CoutPad + co0, bias, bsc, bsh, has_bn,
                                       a.img_shift ? a.img_shift + (size_t)n * CoutPad + co0 : nullptr);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// conv_click<T, WP, HALO> -- the batch-1 click path (ui/gui_draw.py:272-286 fires it on every
// drag pixel).  At N = 1 a layer has one wave per SIMD and a short K loop, so nothing hides a latency unless the kernel
// does it itself.  In-kernel stamps of conv_igemm on a 512->512 layer at 32x32 (tools/ablate, profiles/r02_click_anatomy.txt):
// 7.5 k cycles of prologue, 13 k of main loop for 4.9 k of MFMA issue (every tap-step pays the LDS round trip of its
// fragments and a barrier with nothing else in flight), 12 k of epilogue (fp32 slice sums stored as 16-byte pieces of 32
// different lines per instruction).  This kernel is the same GEMM with the three phases rebuilt around one wave per SIMD:
//   * workgroup = (16 x 4*WP) sites x 64 couts x the cin chunks [kc0, kc1) of one split-K slice x all taps; a "step" is one
//     (chunk, tap) pair = 32 MFMAs per wave (bf16);
//   * operands travel global -> LDS by LDS-DMA only (no VGPR round trip, nothing to wait for until the data is needed):
//     the chunk's halo tile (source-side XOR swizzle; out-of-image rows read a zero page) and one 8 KiB weight tile per
//     step through a 4-deep ring, requested THREE steps ahead, counted vmcnt waits (never 0 in the loop);
//   * the fragments of step s+1 are read (16 ds_read_b128 into a second register set) while the MFMAs of step s issue, and
//     the barrier that publishes step s+1's tile sits at the top of step s: a step's MFMAs never wait for LDS;
//   * ring slot reuse: the tile of step s+3 lands in the slot of step s-1, whose fragment reads were consumed by MFMAs every
//     wave issued before it reached the barrier at the top of step s (program order) -- no read can be in flight;
//   * next chunk's halo tile: second halo buffer, requested at the chunk's tap 1;
//   * split-K epilogue: the wave's 64 px x 64 couts fp32 tile goes through LDS once, so that every store instruction
//     writes four whole 256-byte runs; non-split launches use the fused epilogue of conv_igemm.
// Same fragments, MFMA wrappers, fp32 blocked accumulation and epilogue arithmetic as conv_igemm.
// ------------------------------------------------------------------------------------------------

template <typename T, int WP, int HALO>
__global__ __launch_bounds__(WP * 64, 2) void conv_click(const ConvArgs a) {
    constexpr int NT = WP * 64;
    // bf16: cross-step fragment prefetch (two register sets).  fp32 steps are 16x longer in MFMA time (exact-fp32 MFMA runs
    // at 1/16 of the bf16 rate) and need the registers for the blocked accumulators: one fragment set, the step's own
    // tile is published at its top, and two workgroups per CU (<= 256 registers, <= 80 KiB LDS) cover each other's waits.
    constexpr bool PF = sizeof(T) == 2;
    constexpr int TW = 16, TH = 4 * WP;
    constexpr int HWP = TW + 2 * HALO, HHP = TH + 2 * HALO, HROWS = HWP * HHP;
    constexpr int W_BYTES = kWBlockBytes;                          // 64 couts x 128 B
    constexpr int NH = (HROWS * kSlots + NT - 1) / NT;             // halo DMA pieces per wave
    constexpr int HALO_BYTES = NH * NT * kSlotBytes;
    constexpr int NW = (W_BYTES / kSlotBytes) / NT;                // weight DMA pieces per wave and step (8 / WP)
    constexpr int RING = 4;
    static_assert(NW + NH <= 63 && 3 * NW + NH <= 63, "vmcnt field");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wp = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, g = lane >> 4;
    IDC_STAMP(0);

    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int ksplit = a.ksplit > 1 ? a.ksplit : 1;
    const int sid = b % ksplit; b /= ksplit;
    const int txi = b % a.tiles_x; b /= a.tiles_x;
    const int tyi = b % a.tiles_y; b /= a.tiles_y;
    const int n = b % a.N; b /= a.N;
    const int ct = b % a.ncg;
    const int phase = b / a.ncg;
    const int ty0 = tyi * TH, tx0 = txi * TW;
    const int Hs = a.Hs, Ws = a.Ws, si = a.si;
    const int Win = Ws * si;
    const int pix_bytes = a.nkc * kRowBytes;
    const char* const in_img = (const char*)a.in + (size_t)n * (size_t)(Hs * si) * Win * pix_bytes;
    const int ntaps = a.ntaps;
    const int kc0 = ksplit > 1 ? sid * a.kc_per : 0;
    const int kc1 = ksplit > 1 ? (kc0 + a.kc_per < a.nkc ? kc0 + a.kc_per : a.nkc) : a.nkc;
    const int nch = kc1 - kc0;
    const int total = nch * ntaps;
    // tap tables live in lanes 0..8 of two VGPRs and are read with v_readlane: a scalar load inside the step loop would
    // share lgkmcnt with the fragment reads (SMEM returns out of order: every use costs an lgkmcnt(0))
    int v_roff = 0, v_tw = 0;
#pragma unroll
    for (int t = 0; t < kMaxTaps; ++t) {                           // all 27 scalar loads in one round trip (entries past ntaps are 0)
        const int ro_ = a.dy[phase * 9 + t] * HWP + a.dx[phase * 9 + t], tw_ = a.tw[phase * 9 + t];
        v_roff = lane == t ? ro_ : v_roff;
        v_tw = lane == t ? tw_ : v_tw;
    }
    const int nhalo = a.kc_per > 1 ? 2 : 1;
    char* const halo0 = smem;                                      // [nhalo][HALO_BYTES]
    char* const ring = smem + nhalo * HALO_BYTES;                  // [RING][W_BYTES]

    const char* const wbase = (const char*)a.wgt + (size_t)ct * kWBlockBytes + (size_t)tid * kSlotBytes;
    const size_t w_kc_stride = (size_t)a.ncg * kWBlockBytes;
    const size_t w_tap_stride = w_kc_stride * a.nkc;
    // weight tiles are requested in step order; (ic, it) = chunk / tap of the next step to request
    int ic = 0, it = 0, is = 0;
    auto dma_w_next = [&]() {
        if (is < total) {
            const char* src = wbase + (size_t)__builtin_amdgcn_readlane(v_tw, it) * w_tap_stride + (size_t)(kc0 + ic) * w_kc_stride;
            char* const dst = ring + (is & (RING - 1)) * W_BYTES + wp * 64 * kSlotBytes;
#pragma unroll
            for (int j = 0; j < NW; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)j * NT * kSlotBytes),
                                                 (__attribute__((address_space(3))) void*)(dst + j * NT * kSlotBytes), 16, 0, 0);
        }
        ++is;
        if (++it == ntaps) { it = 0; ++ic; }
    };
    // the first three weight tiles are requested before the halo plan is even computed (their addresses need nothing but
    // the tile indices): the 100+ VALU of the plan run under their flight
    IDC_STAMP_FINE(5);
    dma_w_next(); dma_w_next(); dma_w_next();
    // halo plan: this lane's source offset per piece (fixed for the kernel), -1 = zero row
    int hoff[NH];
#pragma unroll
    for (int j = 0; j < NH; ++j) {
        const int item = tid + j * NT;
        const int hr = item >> 3, sig = item & 7;
        const int hy = hr / HWP, hx = hr - hy * HWP;
        const int sy = ty0 - HALO + hy, sx = tx0 - HALO + hx;
        const bool inside = (unsigned)sy < (unsigned)Hs && (unsigned)sx < (unsigned)Ws && item < HROWS * kSlots;
        hoff[j] = inside ? ((sy * si) * Win + sx * si) * pix_bytes + ((sig ^ swz(hr)) * kSlotBytes) : -1;
    }
    auto dma_halo = [&](int c) {
        const char* const base = in_img + (size_t)(kc0 + c) * kRowBytes;
        char* const dst = halo0 + (c & (nhalo - 1)) * HALO_BYTES + wp * 64 * kSlotBytes;
#pragma unroll
        for (int j = 0; j < NH; ++j) {
            const char* src = hoff[j] >= 0 ? base + hoff[j] : (const char*)a.zeros;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + j * NT * kSlotBytes), 16, 0, 0);
        }
    };
    dma_halo(0);
    IDC_STAMP_FINE(6);
    IDC_STAMP_FINE(7);

    constexpr bool kBlockedAcc = sizeof(T) == 4;
    f32x4 acc[4][4], tot[kBlockedAcc ? 4 : 1][kBlockedAcc ? 4 : 1];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (kBlockedAcc) tot[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    const int wfrag = px * kRowBytes + ((g ^ (px & 7)) * kSlotBytes);   // weight row px (+ ci*16 rows), logical slot g (ks flips bit 2)

    struct Frags { u32x4 w[2][4], x[2][4]; };
    // fragments of step (c, t) from LDS: ring slot `slot`, halo buffer of chunk c
    auto read_frags = [&](Frags& f, int c, int t, int slot) {
        const int roff = __builtin_amdgcn_readlane(v_roff, t);
        const char* const wcur = ring + slot * W_BYTES;
        const char* const halo = halo0 + (c & (nhalo - 1)) * HALO_BYTES;
        int xa[4];
#pragma unroll
        for (int pj = 0; pj < 4; ++pj) {
            const int xr = (wp * 4 + pj + HALO) * HWP + (px + HALO) + roff;
            xa[pj] = xr * kRowBytes + ((g ^ swz(xr)) * kSlotBytes);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) f.w[ks][ci] = *(const u32x4*)(wcur + ((wfrag + ci * 16 * kRowBytes) ^ (ks * 4 * kSlotBytes)));
#pragma unroll
            for (int pj = 0; pj < 4; ++pj) f.x[ks][pj] = *(const u32x4*)(halo + (xa[pj] ^ (ks * 4 * kSlotBytes)));
        }
    };
    auto mma_step = [&](const Frags& f) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                for (int pj = 0; pj < 4; ++pj) Mma<T>::run(acc[ci][pj], f.w[ks][ci], f.x[ks][pj]);
    };

    int sc = 0, st = 0;                                            // chunk / tap of step s
    auto fold_chunk = [&]() {
        if (kBlockedAcc) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    tot[i][j] += acc[i][j];
                    acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
        }
    };
    if constexpr (PF) {
        // step 0 published: its tile and the halo (requested last) have landed, so has everything else requested so far
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        IDC_STAMP(1);
        Frags fa, fb;
        read_frags(fa, 0, 0, 0);
        // one step: publish step s+1 (counted wait + barrier), request step s+3 (+ the next chunk's halo at tap 1), issue
        // the MFMAs of step s from `cur` while the fragments of step s+1 load into `nxt`
        auto step = [&](int s, const Frags& cur, Frags& nxt) {
            const bool more = s + 1 < total;
            int nc = sc, nt = st + 1;
            if (nt == ntaps) { nt = 0; ++nc; }
            if (more) {
                // pieces that may still be in flight once the tile of step s+1 (and, for a chunk's first step, its halo) is
                // in: the tile of step s+2, plus the next chunk's halo when it was requested after the tile of step s+1
                const bool w2 = s + 2 < total;
                const bool halo_behind = nhalo == 2 && sc + 1 < nch && (st == 2 || st == 3) && nt != 0;
                if (w2 && halo_behind) wait_vmcnt<NW + NH>();
                else if (w2) wait_vmcnt<NW>();
                else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
            }
            dma_w_next();                                          // step s+3 -> the slot of step s-1 (reads consumed before the barrier)
            if (st == 1 && sc + 1 < nch) dma_halo(sc + 1);         // -> the buffer of chunk c-1 (last read for step (c-1, last))
            // (after the last step this reads a ring slot / halo buffer nobody needs: harmless, and branch-free -- behind a
            //  branch hipcc joins the paths with an lgkmcnt(0) in front of the MFMAs)
            read_frags(nxt, nc, nt, (s + 1) & (RING - 1));
            mma_step(cur);
            if (nt == 0) fold_chunk();
            sc = nc; st = nt;
        };
        int s = 0;
        for (; s + 1 < total; s += 2) {
            step(s, fa, fb);
            step(s + 1, fb, fa);
        }
        if (s < total) step(s, fa, fb);
    } else {
        bool first = true;
        Frags f;
        for (int s = 0; s < total; ++s) {
            // publish step s: the tiles of steps s+1, s+2 (and a halo requested after the tile of step s) may stay in flight
            const int ahead = total - 1 - s < 2 ? total - 1 - s : 2;
            const bool halo_behind = nhalo == 2 && sc + 1 < nch && st >= 2 && st <= 4;
            if (s == 0) wait_vmcnt<0>();                          // the first halo tile was requested last
            else if (ahead == 2 && halo_behind) wait_vmcnt<2 * NW + NH>();
            else if (ahead == 2) wait_vmcnt<2 * NW>();
            else if (ahead == 1) wait_vmcnt<NW>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (first) { IDC_STAMP(1); first = false; }
            dma_w_next();                                          // step s+3 -> the slot of step s-1
            if (st == 1 && sc + 1 < nch) dma_halo(sc + 1);
            read_frags(f, sc, st, s & (RING - 1));
            mma_step(f);
            if (++st == ntaps) { st = 0; ++sc; fold_chunk(); }
        }
    }
    if (kBlockedAcc) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = tot[i][j];
    }

    IDC_STAMP(2);
    // ---- epilogue: lane owns couts co0..co0+15 of pixel px in each of its 4 rows ----
    const int CoutPad = a.ncg * kCoutGroup;
    const int co0 = ct * kCoutGroup + g * 16;
    const int so = a.so, Wout = Ws * so, Hout = Hs * so;
    const int ro = a.ro[phase], cof = a.co[phase];
    if (ksplit > 1) {
        // raw fp32 slice sums: per pixel row, [16 px][64 couts] fp32 through a wave-private 4 KiB LDS tile so that a store
        // instruction covers four whole 256-byte runs (lane l: pixel l/16, 16-byte piece l%16) instead of 64 scattered pieces
        float* const slab = a.partial + (size_t)sid * a.N * Hout * Wout * CoutPad;
        __builtin_amdgcn_s_barrier();                              // every wave is done with the halo / ring
        char* const tb = smem + wp * 4096;
#pragma unroll
        for (int pj = 0; pj < 4; ++pj) {
#pragma unroll
            for (int ci = 0; ci < 4; ++ci)                         // slot (g*4 + ci) of row px, XOR-swizzled by the row
                *(f32x4*)(tb + px * 256 + (((g * 4 + ci) ^ px) * 16)) = acc[ci][pj];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int sy = ty0 + wp * 4 + pj;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 4 + (lane >> 4), piece = lane & 15;
                const f32x4 v = *(const f32x4*)(tb + row * 256 + ((piece ^ row) * 16));
                const int sx = tx0 + row;
                if (sy < Hs && sx < Ws) {
                    float* o = slab + (((size_t)n * Hout + (sy * so + ro)) * Wout + (sx * so + cof)) * CoutPad + ct * kCoutGroup + piece * 4;
                    *(f32x4*)o = v;
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
    const bool has_bn = a.bn_scale != nullptr;
    float bias[16], bsc[16], bsh[16];
    load16(bias, a.bias + co0);
    if (has_bn) { load16(bsc, a.bn_scale + co0); load16(bsh, a.bn_shift + co0); }
#pragma unroll
    for (int pj = 0; pj < 4; ++pj) {
        const int sy = ty0 + wp * 4 + pj, sx = tx0 + px;
        if (sy < Hs && sx < Ws) {
            const size_t opix = ((size_t)n * Hout + (sy * so + ro)) * Wout + (sx * so + cof);
            float v[16];
#pragma unroll
            for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[ci * 4 + r] = acc[ci][pj][r];
            epilogue16<sizeof(T) == 2>(a, v, opix * CoutPad + co0, bias, bsc, bsh, has_bn,
                                       a.img_shift ? a.img_shift + (size_t)n * CoutPad + co0 : nullptr);
        }
    }
    IDC_STAMP(3);
#ifdef IDC_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    IDC_STAMP(4);
#endif
}

static constexpr int click_halo_bytes(int wp, int halo) {
    const int nt = wp * 64;
    const int hrows = (16 + 2 * halo) * (4 * wp + 2 * halo);
    return ((hrows * kSlots + nt - 1) / nt) * nt * kSlotBytes;
}
static size_t click_lds_bytes(int wp, int halo, int kc_per) {
    const size_t main_loop = (size_t)(kc_per > 1 ? 2 : 1) * click_halo_bytes(wp, halo) + 4 * (size_t)kWBlockBytes;
    const size_t epilogue = (size_t)wp * 4096;
    return main_loop > epilogue ? main_loop : epilogue;
}
// cin chunks one workgroup may walk: unbounded (the weight tiles stream through a ring, halo tiles alternate between two
// buffers); kept as a function so that the engine's split policy has one place to ask
int conv_click_max_chunks(int wp, int halo, int ntaps) {
    (void)wp; (void)halo; (void)ntaps;
    return 1 << 20;
}

template <typename T, int WP, int HALO>
static hipError_t launch_click_t(const ConvArgs& a, hipStream_t s) {
    const size_t lds = click_lds_bytes(WP, HALO, a.kc_per);
    const long long blocks = (long long)a.tiles_x * a.tiles_y * a.N * a.ncg * a.nphase * (a.ksplit > 1 ? a.ksplit : 1);
    if (blocks <= 0 || blocks > 0x7fffffffLL || lds > 160 * 1024 || a.zeros == nullptr) return hipErrorInvalidValue;
    if (a.ksplit > 1 && a.partial == nullptr) return hipErrorInvalidValue;
    if (a.ntaps < 4 || a.kc_per < 1) return hipErrorInvalidConfiguration;     // the halo-prefetch schedule assumes >= 4 taps per chunk
    hipLaunchKernelGGL((conv_click<T, WP, HALO>), dim3((unsigned)blocks), dim3(WP * 64), lds, s, a);
    return hipGetLastError();
}

#define IDC_FOR_EACH_CLICK(X) X(4, 1) X(4, 2) X(2, 1) X(2, 2) X(1, 1) X(1, 2)

hipError_t launch_conv_click(int precision, int wp, int halo, const ConvArgs& a, hipStream_t s) {
#define X(WP, HL) \
    if (wp == WP && halo == HL) return precision == 1 ? launch_click_t<__bf16, WP, HL>(a, s) : launch_click_t<float, WP, HL>(a, s);
    IDC_FOR_EACH_CLICK(X)
#undef X
    return hipErrorInvalidConfiguration;
}

static hipError_t init_kernels_click() {
    hipError_t e;
#define X(WP, HL)                                                                                                          \
    e = hipFuncSetAttribute((const void*)conv_click<__bf16, WP, HL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    if (e != hipSuccess) return e;                                                                                         \
    e = hipFuncSetAttribute((const void*)conv_click<float, WP, HL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  \
    if (e != hipSuccess) return e;
    IDC_FOR_EACH_CLICK(X)
#undef X
    return hipSuccess;
}

// ------------------------------------------------------------------------------------------------
static constexpr size_t conv_lds_bytes_c(int wm, int wp, int halo) {
    const int nt = wm * wp * 64;
    const int hrows = (16 + 2 * halo) * (4 * wp + 2 * halo);
    const int items = (hrows * kSlots + nt - 1) / nt;
    // HALO == 0 instantiations also serve conv1_1's fused input pack: + the (TH+2)x(TW+2) float4 input patch
    return (size_t)items * nt * kSlotBytes + 2 * (size_t)(64 * wm) * kRowBytes + (halo == 0 ? (size_t)18 * (4 * wp + 2) * 16 : 0);
}

template <typename T, int WM, int WP, int HALO>
static hipError_t launch_conv_t(const ConvArgs& a, hipStream_t s) {
    constexpr int NT = WM * WP * 64;
    constexpr size_t lds = conv_lds_bytes_c(WM, WP, HALO);
    const int nct = a.ncg / WM;
    const long long blocks = (long long)a.tiles_x * a.tiles_y * a.N * nct * a.nphase * (a.ksplit > 1 ? a.ksplit : 1);
    if (blocks <= 0 || blocks > 0x7fffffffLL || a.zeros == nullptr) return hipErrorInvalidValue;
    if (a.ksplit > 1 && (a.partial == nullptr || a.kc_per <= 0)) return hipErrorInvalidValue;
    hipLaunchKernelGGL((conv_igemm<T, WM, WP, HALO>), dim3((unsigned)blocks), dim3(NT), lds, s, a);
    return hipGetLastError();
}

template <typename T, int WM, int WP, int HALO>
static hipError_t set_lds_attr() {
    constexpr size_t lds = conv_lds_bytes_c(WM, WP, HALO);
    return hipFuncSetAttribute((const void*)conv_igemm<T, WM, WP, HALO>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

hipError_t init_kernels_v2();
hipError_t init_kernels_conv1();
size_t conv_lds_bytes(ConvConfig cfg, int halo) { return conv_lds_bytes_c(cfg.wm, cfg.wp, halo); }

#define IDC_FOR_EACH_CONV(X)                                                             \
    X(2, 2, 0) X(2, 2, 1) X(2, 2, 2) X(1, 4, 0) X(1, 4, 1) X(1, 4, 2) X(2, 4, 0) X(2, 4, 1) \
    X(2, 4, 2) X(1, 2, 0) X(1, 2, 1) X(1, 2, 2) X(1, 1, 0) X(1, 1, 1) X(1, 1, 2) X(2, 1, 0) X(2, 1, 1) \
    X(2, 1, 2)

hipError_t init_kernels() {
    hipError_t e;
#define X(WM, WP, HL)                                              \
    e = set_lds_attr<float, WM, WP, HL>();  if (e != hipSuccess) return e; \
    e = set_lds_attr<__bf16, WM, WP, HL>(); if (e != hipSuccess) return e;
    IDC_FOR_EACH_CONV(X)
#undef X
    e = init_kernels_click();
    if (e != hipSuccess) return e;
    e = init_kernels_wino();
    if (e != hipSuccess) return e;
    e = init_kernels_v2m();
    if (e != hipSuccess) return e;
    e = init_kernels_dsm();
    if (e != hipSuccess) return e;
    e = init_kernels_kw();
    if (e != hipSuccess) return e;
    e = init_kernels_conv1();
    if (e != hipSuccess) return e;
    return init_kernels_v2();
}

hipError_t launch_conv(int precision, ConvConfig cfg, int halo, const ConvArgs& a, hipStream_t s) {
#define X(WM, WP, HL)                                                                   \
    if (cfg.wm == WM && cfg.wp == WP && halo == HL)                                     \
        return precision == 1 ? launch_conv_t<__bf16, WM, WP, HL>(a, s) : launch_conv_t<float, WM, WP, HL>(a, s);
    IDC_FOR_EACH_CONV(X)
#undef X
    return hipErrorInvalidConfiguration;
}


// ------------------------------------------------------------------------------------------------
// splitk_epilogue: sum the split-K slices in fixed order (deterministic) and finish the layer.
// One thread = 8 consecutive channels of one output pixel (32-byte slab reads, 16/32-byte stores).
// ------------------------------------------------------------------------------------------------
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const ConvArgs a) {
    const int CoutPad = a.ncg * kCoutGroup;
    const int c8 = CoutPad >> 3;
    const int Hout = a.Hs * a.so, Wout = a.Ws * a.so;
    const long long npix = (long long)a.N * Hout * Wout;
    const long long total = npix * c8;
    const size_t slab = (size_t)npix * CoutPad;
    const bool has_bn = a.bn_scale != nullptr;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cq = (int)(i % c8);
        const long long pix = i / c8;
        const int co = cq * 8;
        const size_t oidx = (size_t)pix * CoutPad + co;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < a.ksplit; ++s) {
            const float4 p0 = *(const float4*)(a.partial + (size_t)s * slab + oidx);
            const float4 p1 = *(const float4*)(a.partial + (size_t)s * slab + oidx + 4);
            v[0] += p0.x; v[1] += p0.y; v[2] += p0.z; v[3] += p0.w; v[4] += p1.x; v[5] += p1.y; v[6] += p1.z; v[7] += p1.w;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += a.bias[co + e];
        if (a.resid != nullptr) {
            if (a.resid_bf16) {
                const uint4 r4 = *(const uint4*)((const unsigned short*)a.resid + oidx);
                const unsigned rw[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[2 * e] += __uint_as_float(rw[e] << 16); v[2 * e + 1] += __uint_as_float(rw[e] & 0xffff0000u); }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += ((const float*)a.resid)[oidx + e];
            }
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
            for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], a.bn_scale[co + e], a.bn_shift[co + e]);
        }
        if (a.img_shift != nullptr) {
            const long long n = pix / ((long long)Hout * Wout);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += a.img_shift[(size_t)n * CoutPad + co + e];
        }
        if (!OUT_BF16 || a.out_f32) {
            float* o = (float*)a.out + oidx;
            *(float4*)o = float4{v[0], v[1], v[2], v[3]};
            *(float4*)(o + 4) = float4{v[4], v[5], v[6], v[7]};
        } else {
            uint4 o;
            o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
            o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
            *(uint4*)((unsigned short*)a.out + oidx) = o;
        }
    }
}

hipError_t launch_splitk_epilogue(int precision, const ConvArgs& a, hipStream_t s) {
    const long long total = (long long)a.N * a.Hs * a.so * a.Ws * a.so * (a.ncg * kCoutGroup / 8);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (precision == 1) hipLaunchKernelGGL(splitk_epilogue_kernel<true>, dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(splitk_epilogue_kernel<false>, dim3(blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}


}  // namespace idc
