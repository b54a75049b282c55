// idc_kw.hip -- conv_kwave_bf16: the 3x3 stride-1 convolutions of the bf16 BATCH-1 CLICK PATH (BASELINE configs[1]; ui/gui_draw.py:272-286
// fires the forward on every drag pixel; model.py:13-102), K split over the WAVES of a workgroup (gfx950).
//
// What a batch-1 layer costs (profiles/r04_kwave.txt; DESIGN.md section 4, "Round 4: the click path"): 256 workgroups, one per CU.  An empty
// kernel of that grid takes 6.1 us under per-launch events; a 512 -> 512 layer at 32 x 32 took 16.9 us as Winograd (conv_wino_bf16: 16 positions
// x Cin x 32 couts x 2 B = 512 KiB of U per workgroup + the patch, eight chunks x (barrier, transform, U one chunk ahead) = a chain of eight
// memory round trips; touching U early for L2 residency made it slower) and takes 11.7 us here: 6.1 launch floor + 2.4 MFMAs / reduction /
// epilogue + ~1.5 weight stream + ~1.5 halo stream.  The direct form needs 9/16 of the weight bytes; what it lacked at batch 1 was
// parallelism without a split-K reduction LAUNCH.  Here the K split lives inside the workgroup:
//   * workgroup = 8 x (8*TWB) pixels of one dilation sub-grid (d = 2: four independent d = 1 problems on the parity grids, as in the
//     Winograd kernels) x 32 couts x the WHOLE K; NW waves, wave w = (cin chunk w % NKC, tap range w / NKC): NKC = 8 -> a wave owns
//     one 64-channel chunk and all nine taps; NKC = 4 -> two waves per chunk split the taps 4 : 5; ...
//   * the whole-K halo tile (NKC x 10 x (8*TWB+2) pixels x 128 B) goes global -> LDS by LDS-DMA once (source-side XOR swizzle, zero
//     page outside the image), ONE barrier; after it the waves run independently -- no barrier in the K loop at all;
//   * weights: the layout-1 blocks conv_click / conv_igemm_v2m read ([tap][chunk][64 couts][128 B], slot ^ (row & 7)), streamed
//     global -> registers as MFMA A fragments (a wave's 16-byte pieces of a tap = one 4 KiB run), four taps ahead;
//   * v_mfma_f32_16x16x32_bf16, fp32 accumulation; the NW partial sums of a (pixel, cout) meet in LDS in a fixed order (deterministic),
//     then bias, activation, eval-BN, per-image shift, bf16 (or fp32) store -- the epilogue arithmetic of conv_wino_bf16.
// Per workgroup at Cin = 512, 32 x 32 pixels: 288 KiB of weights + 100 KiB of halo instead of 512 + 100.
// Numerics: plain bf16 products, fp32 sums (no transform): the error of the direct kernels (conv_click), below conv_wino_bf16's.
#include <stdlib.h>

#include <type_traits>

#include "idc_kernels.h"

#include "idc_layout.h"

namespace idc {

typedef __attribute__((ext_vector_type(4))) float f32x4_k;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_k;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_k;

__device__ __forceinline__ int xcd_remap_k(int b, int nb) {
    const int xcd = b & 7, q = nb >> 3, r = nb & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (b >> 3);
}

// weight fragments global -> registers by inline asm with counted waits: as plain loads hipcc sinks the look-ahead loads down to their first use
// (a tap's weights then arrive one memory latency after they are asked for, tap after tap) and joins the first use with a vmcnt(0).  Loads return in
// order; kw_wait*<N> = "these registers have landed once at most N younger loads are in flight", the operands tie the MFMAs that read them behind it.
__device__ __forceinline__ void kw_gload(u32x4_k& dst, const char* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p)); }
template <int N> __device__ __forceinline__ void kw_wait4(u32x4_k& a, u32x4_k& b, u32x4_k& c, u32x4_k& d) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}

// agent-coherent accesses of the persistent trunk launch (conv_kwave_chain_bf16): sc1 = coherent across the XCDs' L2s
__device__ __forceinline__ unsigned long long kw_ld64_sc1(const unsigned long long* p) {
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void kw_st64_sc1(void* p, uint2 v) { asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }

constexpr int kw_halo_items(int nkc, int twb) { return nkc * 10 * (8 * twb + 2) * 8; }
constexpr int kw_lds_bytes(int nkc, int twb, int nw) {
    const int nt = nw * 64;
    const int halo = ((kw_halo_items(nkc, twb) + nt - 1) / nt) * nt * kSlotBytes;
    const int red = nw * 8192;                                 // one round of partial sums: [wave][64 pixels][32 couts] fp32
    return (halo > red ? halo : red) + 256;                    // + the scratch line of idc_warm_own_code
}

constexpr int kwd_lds_bytes(int nkc) {
    const int halo = ((nkc * 800 + 511) / 512) * 512 * kSlotBytes;
    const int red = 8 * (4 / (8 / nkc)) * 4096;                // [wave][local phase][4 KiB]
    return (halo > red ? halo : red) + 256;                    // + the scratch line of idc_warm_own_code
}

template <int NKC, int TWB, int NW>
__global__ __launch_bounds__(NW * 64, 2) void conv_kwave_bf16(const ConvArgs a) {
    static_assert(NW % NKC == 0, "waves = chunks x tap ranges");
    constexpr int NT = NW * 64, NS = NW / NKC;
    constexpr int MAXT = (9 + NS - 1) / NS;                    // taps of the longest range
    constexpr int PWD = 8 * TWB + 2, HR = 10 * PWD;            // halo pixels of one chunk
    constexpr int ITEMS = kw_halo_items(NKC, TWB);
    constexpr int NH = (ITEMS + NT - 1) / NT;
    constexpr int PB = 4 * TWB;                                // 16-pixel blocks of the tile
    constexpr int PD = MAXT < 4 ? MAXT : 4;                    // taps of weights in flight
    static_assert(NH + PD * 4 <= 63, "vmcnt field");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, g = lane >> 4;

    // first launch of this kernel after others (eight kernel changes per click forward): its code comes in as data, all lines at once
    // (idc_kernels.h; 7.5-10.3 KB per instantiation, the <1,2,4> form is the last kernel of its code object: stay inside)
    if (a.warm && wave == NW - 1) idc_warm_own_code(smem + kw_lds_bytes(NKC, TWB, NW) - 256, lane, NKC == 8 ? 64 : NKC == 4 ? 68 : NKC == 2 ? 76 : 48);
    int b = xcd_remap_k(blockIdx.x, gridDim.x);               // tile blocks fastest, the cout group slowest: an XCD's L2 sees few weight slices
    const int d = a.dy[8];
    const int bx = b % a.tiles_x; b /= a.tiles_x;
    const int by = b % a.tiles_y; b /= a.tiles_y;
    const int par = b % (d * d); b /= d * d;
    const int n = b % a.N;
    const int cg = b / a.N;                                    // group of 32 couts
    const int Y0 = par / d + d * 8 * by, X0 = par % d + d * 8 * TWB * bx;      // the tile's pixels are (Y0 + d*r, X0 + d*c)
    const int H = a.Hs, W = a.Ws, si = a.si;
    const int pix_bytes = NKC * kRowBytes;
    const char* const img = (const char*)a.in + (size_t)n * (H * si) * (W * si) * pix_bytes;

    // ---- the whole-K halo tile: global -> LDS, 16-byte pieces, item = (chunk, halo pixel, slot) in LDS order ----------------
#pragma unroll
    for (int j = 0; j < NH; ++j) {
        const int item = tid + j * NT;
        const int kc = item / (HR * 8), rem = item - kc * (HR * 8);
        const int hr = rem >> 3, sig = rem & 7;
        const int hy = hr / PWD, hx = hr - hy * PWD;
        const int Y = Y0 + d * (hy - 1), X = X0 + d * (hx - 1);
        const bool inside = item < ITEMS && (unsigned)Y < (unsigned)H && (unsigned)X < (unsigned)W;
        const char* const src = inside ? img + (((Y * si) * (W * si) + X * si) * pix_bytes + kc * kRowBytes + ((sig ^ (hr & 7)) * kSlotBytes))
                                       : (const char*)a.zeros;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem + (j * NT + wave * 64) * kSlotBytes), 16, 0, 0);
    }

    // ---- this wave's share of K: chunk kcw, taps [t0, t1) -------------------------------------------------------------------
    const int kcw = wave % NKC, split = wave / NKC;
    const int t0 = (split * 9) / NS, t1 = ((split + 1) * 9) / NS;
    const size_t w_kc_stride = (size_t)a.ncg * kWBlockBytes;
    const size_t w_tap_stride = w_kc_stride * NKC;
    // the workgroup's 32 rows of the 64-row block: rows (cg & 1) * 32 + cb * 16 + m, m = the MFMA row = this lane's px; row lam holds
    // cout cg_row_to_cout(lam) (idc_layout.h): accumulator register r of lane group g in block cb = cout g*16 + ((cg&1)*2 + cb)*4 + r
    const char* const wl = (const char*)a.wgt + (size_t)kcw * w_kc_stride + (size_t)(cg >> 1) * kWBlockBytes + (cg & 1) * 32 * kRowBytes;
    int wo[2][2];                                              // logical slots g (ks 0) and 4 + g (ks 1) of row px (+ 16 for the second block)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            wo[cb][ks] = (cb * 16 + px) * kRowBytes + (((ks * 4 + g) ^ (px & 7)) * kSlotBytes);
    u32x4_k areg[PD][2][2];
    auto load_A = [&](auto slotc, int t) {
        constexpr int S = decltype(slotc)::value;
        const char* const src = wl + (size_t)(t < 8 ? t : 8) * w_tap_stride;         // past the range: a harmless re-read (same count on every wave)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            kw_gload(areg[S][cb][0], src + wo[cb][0]);
            kw_gload(areg[S][cb][1], src + wo[cb][1]);
        }
    };
    auto wait_A = [&](auto slotc, auto nc) {
        constexpr int S = decltype(slotc)::value, N = decltype(nc)::value;
        kw_wait4<N>(areg[S][0][0], areg[S][0][1], areg[S][1][0], areg[S][1][1]);
    };
    if constexpr (PD > 0) load_A(std::integral_constant<int, 0>{}, t0);
    if constexpr (PD > 1) load_A(std::integral_constant<int, 1>{}, t0 + 1);
    if constexpr (PD > 2) load_A(std::integral_constant<int, 2>{}, t0 + 2);
    if constexpr (PD > 3) load_A(std::integral_constant<int, 3>{}, t0 + 3);

    f32x4_k acc[2][PB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < PB; ++j) acc[i][j] = f32x4_k{0.f, 0.f, 0.f, 0.f};
    // the lane's pixel in each 16-pixel block, as a halo pixel index (tap (1,1))
    int hbase[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        const int row = TWB == 1 ? pb * 2 + (px >> 3) : pb, col = TWB == 1 ? (px & 7) : px;
        hbase[pb] = (row + 1) * PWD + col + 1;
    }

    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PD * 4) : "memory");      // the halo pieces are older than the weight loads
    __builtin_amdgcn_s_barrier();

    const char* const hchunk = smem + kcw * (HR * kRowBytes);
    auto tap = [&](auto ic) {
        constexpr int I = decltype(ic)::value;
        const int t = t0 + I;
        constexpr int YOUNGER = (PD - 1 < MAXT - 1 - I ? PD - 1 : MAXT - 1 - I) * 4;
        wait_A(std::integral_constant<int, I % PD>{}, std::integral_constant<int, YOUNGER>{});
        if (t < t1) {                                          // wave-uniform
            const int ty = t / 3, tx = t - ty * 3;
            const int toff = (ty - 1) * PWD + (tx - 1);
            u32x4_k bf[PB][2];
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                const int hr = hbase[pb] + toff;
                const int o0 = hr * kRowBytes + ((g ^ (hr & 7)) * kSlotBytes);
                bf[pb][0] = *(const u32x4_k*)(hchunk + o0);
                bf[pb][1] = *(const u32x4_k*)(hchunk + (o0 ^ (4 * kSlotBytes)));
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb)
                        acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_k, areg[I % PD][cb][ks]),
                                                                              __builtin_bit_cast(bf16x8_k, bf[pb][ks]), acc[cb][pb], 0, 0, 0);
        }
        if constexpr (I + PD < MAXT) load_A(std::integral_constant<int, I % PD>{}, t0 + I + PD);
    };
    tap(std::integral_constant<int, 0>{});
    if constexpr (MAXT > 1) tap(std::integral_constant<int, 1>{});
    if constexpr (MAXT > 2) tap(std::integral_constant<int, 2>{});
    if constexpr (MAXT > 3) tap(std::integral_constant<int, 3>{});
    if constexpr (MAXT > 4) tap(std::integral_constant<int, 4>{});
    if constexpr (MAXT > 5) tap(std::integral_constant<int, 5>{});
    if constexpr (MAXT > 6) tap(std::integral_constant<int, 6>{});
    if constexpr (MAXT > 7) tap(std::integral_constant<int, 7>{});
    if constexpr (MAXT > 8) tap(std::integral_constant<int, 8>{});

    // ---- the NW partial sums of a (pixel, cout) meet in LDS, 64 pixels per round; fixed summation order --------------------
    const int CoutPad = a.ncg * kCoutGroup;
    const bool has_bn = a.bn_scale != nullptr;
#pragma unroll
    for (int h = 0; h < TWB; ++h) {
        __syncthreads();                                       // every wave is done with the halo (h = 0) / the previous round's sums are read
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const int p64 = q * 16 + px;
                *(f32x4_k*)(smem + wave * 8192 + p64 * kRowBytes + (((g * 2 + cb) ^ (p64 & 7)) * kSlotBytes)) = acc[cb][h * 4 + q];
            }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < (64 * 8) / NT; ++k) {
            const int idx = tid + k * NT;
            const int p64 = idx >> 3, slot = idx & 7;
            f32x4_k s = *(const f32x4_k*)(smem + p64 * kRowBytes + ((slot ^ (p64 & 7)) * kSlotBytes));
#pragma unroll
            for (int w = 1; w < NW; ++w) s += *(const f32x4_k*)(smem + w * 8192 + p64 * kRowBytes + ((slot ^ (p64 & 7)) * kSlotBytes));
            const int P = h * 64 + p64, pb = P >> 4, pp = P & 15;
            const int row = TWB == 1 ? pb * 2 + (pp >> 3) : pb, col = TWB == 1 ? (pp & 7) : pp;
            const int yy = Y0 + d * row, xx = X0 + d * col;
            const int co = (cg >> 1) * kCoutGroup + (slot >> 1) * 16 + (cg & 1) * 8 + (slot & 1) * 4;      // slot = g*2 + cb (see the weight rows above)
            const f32x4_k bias = *(const f32x4_k*)(a.bias + co);
            f32x4_k v = s + bias;
            if (a.act == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            else if (a.act == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.2f * v[r];
            }
            if (has_bn) {
                const f32x4_k sc = *(const f32x4_k*)(a.bn_scale + co), sh = *(const f32x4_k*)(a.bn_shift + co);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaf(v[r], sc[r], sh[r]);
            }
            if (a.img_shift) v += *(const f32x4_k*)(a.img_shift + (size_t)n * CoutPad + co);
            if (yy < H && xx < W) {
                const size_t o = (((size_t)n * H + yy) * W + xx) * CoutPad + co;
                if (a.out_f32) *(f32x4_k*)((float*)a.out + o) = v;
                else {
                    const __bf16 q0 = (__bf16)v[0], q1 = (__bf16)v[1], q2 = (__bf16)v[2], q3 = (__bf16)v[3];
                    uint2 pk;
                    pk.x = (unsigned)__builtin_bit_cast(unsigned short, q0) | ((unsigned)__builtin_bit_cast(unsigned short, q1) << 16);
                    pk.y = (unsigned)__builtin_bit_cast(unsigned short, q2) | ((unsigned)__builtin_bit_cast(unsigned short, q3) << 16);
                    *(uint2*)((__bf16*)a.out + o) = pk;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// conv_kwave_deconv_bf16<NKC>: ConvTranspose2d 4x4 stride 2 pad 1 (model8up / model9up / model10up, model.py:75,87,97) of the bf16
// click path in the same form.  out[co, 2m+r, 2n+s] = b + sum over (ky,dy) in T(r), (kx,dx) in T(s), ci of in[ci, m+dy, n+dx] W[ci,co,ky,kx],
// T(0) = {(1,0),(3,-1)}, T(1) = {(0,+1),(2,0)}: four phases (r,s) x four taps (i,j), ky = (1-r) + 2i, dy = r - i.
// Workgroup = 8 x 8 input sites (16 x 16 output pixels) x 16 couts (one MFMA row block of the layout-1 weight block: couts g*16 + q*4 + reg)
// x the whole K; 8 waves, wave w = (cin chunk w % NKC, phases [(w / NKC) * PHW, +PHW)): every (phase, tap) weight tile (16 rows x 128 B)
// is read by exactly one wave, global -> registers four items ahead; the halo tile is the 3x3 kernel's (10 x 10 sites x NKC chunks, one barrier).
// The NKC partial sums of a (phase, site, cout) meet in LDS in chunk order; + bias + the shortcut sum (model.py:156,170), activation,
// store at (2m+r, 2n+s).  The Winograd F(2x2,2x2) form it replaces on model8up / model9up streams 36/16 of these weight bytes.
template <int NKC>
__global__ __launch_bounds__(512, NKC == 8 ? 2 : 4) void conv_kwave_deconv_bf16(const ConvArgs a) {
    constexpr int NW = 8, NT = NW * 64, NS = NW / NKC, PHW = 4 / NS;     // phases per wave
    static_assert(NS == 1 || NS == 2 || NS == 4, "8 waves = chunks x phase groups");
    constexpr int MAXI = PHW * 4;                              // (phase, tap) items of a wave
    constexpr int PWD = 10, HR = 100;
    constexpr int ITEMS = NKC * HR * 8;
    constexpr int NH = (ITEMS + NT - 1) / NT;
    constexpr int PD = MAXI < 4 ? MAXI : 4;
    static_assert(NH + PD * 2 <= 63, "vmcnt field");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, g = lane >> 4;

    if (a.warm && wave == NW - 1) idc_warm_own_code(smem + kwd_lds_bytes(NKC) - 256, lane, NKC == 8 ? 70 : NKC == 4 ? 46 : 32);   // 9.5 / 6.3 / 4.5 KB
    int b = xcd_remap_k(blockIdx.x, gridDim.x);
    const int bx = b % a.tiles_x; b /= a.tiles_x;
    const int by = b % a.tiles_y; b /= a.tiles_y;
    const int n = b % a.N;
    const int cq = b / a.N;                                    // group of 16 couts: row block cq & 3 of weight block cq >> 2
    const int Y0 = 8 * by, X0 = 8 * bx;
    const int H = a.Hs, W = a.Ws;                              // INPUT sites; the output is 2H x 2W
    const int pix_bytes = NKC * kRowBytes;
    const char* const img = (const char*)a.in + (size_t)n * H * W * pix_bytes;

#pragma unroll
    for (int j = 0; j < NH; ++j) {
        const int item = tid + j * NT;
        const int kc = item / (HR * 8), rem = item - kc * (HR * 8);
        const int hr = rem >> 3, sig = rem & 7;
        const int hy = hr / PWD, hx = hr - hy * PWD;
        const int Y = Y0 + hy - 1, X = X0 + hx - 1;
        const bool inside = item < ITEMS && (unsigned)Y < (unsigned)H && (unsigned)X < (unsigned)W;
        const char* const src = inside ? img + ((Y * W + X) * pix_bytes + kc * kRowBytes + ((sig ^ (hr & 7)) * kSlotBytes)) : (const char*)a.zeros;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem + (j * NT + wave * 64) * kSlotBytes), 16, 0, 0);
    }

    const int kcw = wave % NKC, split = wave / NKC;
    const size_t w_kc_stride = (size_t)a.ncg * kWBlockBytes;
    const size_t w_tap_stride = w_kc_stride * NKC;
    const char* const wl = (const char*)a.wgt + (size_t)kcw * w_kc_stride + (size_t)(cq >> 2) * kWBlockBytes + ((cq & 3) * 16 + px) * kRowBytes;
    const int wo0 = (g ^ (px & 7)) * kSlotBytes, wo1 = wo0 ^ (4 * kSlotBytes);
    // item I of this wave: phase ph = split * PHW + I / 4 = (r, s), tap (i, j) = ((I >> 1) & 1, I & 1)
    auto item_tw = [&](int I) {
        const int ph = split * PHW + (I >> 2), r = ph >> 1, sx = ph & 1, i = (I >> 1) & 1, j = I & 1;
        return ((1 - r) + 2 * i) * 4 + (1 - sx) + 2 * j;
    };
    u32x4_k areg[PD][2];
    auto load_A = [&](auto slotc, int I) {
        constexpr int S = decltype(slotc)::value;
        const char* const src = wl + (size_t)item_tw(I) * w_tap_stride;
        areg[S][0] = *(const u32x4_k*)(src + wo0);              // (plain loads here: with two loads per item hipcc's own schedule measured 0.3 us
        areg[S][1] = *(const u32x4_k*)(src + wo1);              //  per launch better than the counted-wait form of conv_kwave_bf16)
    };
    if constexpr (PD > 0) load_A(std::integral_constant<int, 0>{}, 0);
    if constexpr (PD > 1) load_A(std::integral_constant<int, 1>{}, 1);
    if constexpr (PD > 2) load_A(std::integral_constant<int, 2>{}, 2);
    if constexpr (PD > 3) load_A(std::integral_constant<int, 3>{}, 3);

    f32x4_k acc[PHW][4];
#pragma unroll
    for (int i = 0; i < PHW; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_k{0.f, 0.f, 0.f, 0.f};
    int hbase[4];
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) hbase[pb] = (pb * 2 + (px >> 3) + 1) * PWD + (px & 7) + 1;

    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PD * 2) : "memory");
    __builtin_amdgcn_s_barrier();

    const char* const hchunk = smem + kcw * (HR * kRowBytes);
    auto item = [&](auto ic) {
        constexpr int I = decltype(ic)::value;
        const int ph = split * PHW + (I >> 2), r = ph >> 1, sx = ph & 1;
        constexpr int i = (I >> 1) & 1, j = I & 1;
        const int toff = (r - i) * PWD + (sx - j);
        u32x4_k bf[4][2];
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
            const int hr = hbase[pb] + toff;
            const int o0 = hr * kRowBytes + ((g ^ (hr & 7)) * kSlotBytes);
            bf[pb][0] = *(const u32x4_k*)(hchunk + o0);
            bf[pb][1] = *(const u32x4_k*)(hchunk + (o0 ^ (4 * kSlotBytes)));
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int pb = 0; pb < 4; ++pb)
                acc[I >> 2][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_k, areg[I % PD][ks]),
                                                                          __builtin_bit_cast(bf16x8_k, bf[pb][ks]), acc[I >> 2][pb], 0, 0, 0);
        if constexpr (I + PD < MAXI) load_A(std::integral_constant<int, I % PD>{}, I + PD);
    };
    item(std::integral_constant<int, 0>{}); item(std::integral_constant<int, 1>{});
    item(std::integral_constant<int, 2>{}); item(std::integral_constant<int, 3>{});
    if constexpr (MAXI > 4) {
        item(std::integral_constant<int, 4>{}); item(std::integral_constant<int, 5>{});
        item(std::integral_constant<int, 6>{}); item(std::integral_constant<int, 7>{});
    }
    if constexpr (MAXI > 8) {
        item(std::integral_constant<int, 8>{}); item(std::integral_constant<int, 9>{});
        item(std::integral_constant<int, 10>{}); item(std::integral_constant<int, 11>{});
        item(std::integral_constant<int, 12>{}); item(std::integral_constant<int, 13>{});
        item(std::integral_constant<int, 14>{}); item(std::integral_constant<int, 15>{});
    }

    // ---- partial sums: [wave][local phase][site 64][16 couts] fp32, 64-byte rows, slot g ^ ((site >> 1) & 3) ---------------------
    __syncthreads();                                           // every wave is done with the halo
#pragma unroll
    for (int pl = 0; pl < PHW; ++pl)
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
            const int site = pb * 16 + px;
            *(f32x4_k*)(smem + (wave * PHW + pl) * 4096 + site * 64 + ((g ^ ((site >> 1) & 3)) * kSlotBytes)) = acc[pl][pb];
        }
    __syncthreads();
    const int CoutPad = a.ncg * kCoutGroup;
    const bool has_bn = a.bn_scale != nullptr;
    const int Ho = 2 * H, Wo = 2 * W;
#pragma unroll
    for (int k = 0; k < (4 * 64 * 4) / NT; ++k) {
        const int idx = tid + k * NT;
        const int slot = idx & 3, site = (idx >> 2) & 63, ph = idx >> 8;
        const int sp = ph / PHW, pl = ph - sp * PHW;
        const char* const base = smem + ((sp * NKC) * PHW + pl) * 4096 + site * 64 + ((slot ^ ((site >> 1) & 3)) * kSlotBytes);
        f32x4_k v = *(const f32x4_k*)base;
#pragma unroll
        for (int kc = 1; kc < NKC; ++kc) v += *(const f32x4_k*)(base + kc * PHW * 4096);
        const int pb = site >> 4, pp = site & 15;
        const int sy = Y0 + pb * 2 + (pp >> 3), sxx = X0 + (pp & 7);
        const int co = (cq >> 2) * kCoutGroup + slot * 16 + (cq & 3) * 4;
        v += *(const f32x4_k*)(a.bias + co);
        if (sy < H && sxx < W) {
            const size_t o = (((size_t)n * Ho + (2 * sy + (ph >> 1))) * Wo + (2 * sxx + (ph & 1))) * CoutPad + co;
            if (a.resid) {
                if (a.resid_bf16) {
                    const uint2 rr = *(const uint2*)((const __bf16*)a.resid + o);
                    v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
                    v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
                } else {
                    v += *(const f32x4_k*)((const float*)a.resid + o);
                }
            }
            if (a.act == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            else if (a.act == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.2f * v[r];
            }
            if (has_bn) {
                const f32x4_k sc = *(const f32x4_k*)(a.bn_scale + co), sh = *(const f32x4_k*)(a.bn_shift + co);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaf(v[r], sc[r], sh[r]);
            }
            if (a.out_f32) *(f32x4_k*)((float*)a.out + o) = v;
            else {
                const __bf16 q0 = (__bf16)v[0], q1 = (__bf16)v[1], q2 = (__bf16)v[2], q3 = (__bf16)v[3];
                uint2 pk;
                pk.x = (unsigned)__builtin_bit_cast(unsigned short, q0) | ((unsigned)__builtin_bit_cast(unsigned short, q1) << 16);
                pk.y = (unsigned)__builtin_bit_cast(unsigned short, q2) | ((unsigned)__builtin_bit_cast(unsigned short, q3) << 16);
                *(uint2*)((__bf16*)a.out + o) = pk;
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// conv_kwave_chain_bf16 (round 5): a run of consecutive conv_kwave_bf16<8,1,8> layers -- the 512 -> 512 trunk of the
// bf16 click forward at batch 1: conv4_2, conv4_3, conv5_1 .. conv7_3, eleven launches of 9-12 us each of which 4.5-6 are launch floor -- as
// ONE persistent launch.  Every layer has the same grid (8x8-pixel tiles x dilation parities x 32-cout groups = 256 workgroups at 32 x 32,
// one per CU: 104 KiB of LDS each), so workgroup b runs its tile of layer 0, then of layer 1, ...; between two layers a GRID BARRIER:
//   arrive : every wave waits for its stores (vmcnt 0), workgroup barrier, thread 0: one agent-scope atomic add on the counter of blockIdx & 7
//            (eight counters, 32 arrivals each: 256 arrivals on ONE address serialise at ~17 ns apiece);
//   wait   : threads 0..7 poll one counter each (sc1 loads, s_sleep between polls), workgroup barrier.
// Measured (tools/ubench/grid_barrier.hip, profiles/r05_grid_barrier.txt): the textbook form -- agent-scope release fence (buffer_wbl2 sc1:
// the XCD's whole L2 written back), one counter, acquire fence (buffer_inv sc1) -- costs 12.0 us per barrier, TWICE a launch floor; one counter
// without cache maintenance 4.5 us; per-XCD counters 1.9 us.  So there is no cache maintenance at the barrier: the activations are STORED
// with sc1 (agent-coherent write-through: in memory once vmcnt says so) and LOADED with sc1 (every halo piece comes from the memory side).
// Invalidating instead (buffer_inv sc1 after the barrier, or before arriving -- nobody reads the next input until all have arrived -- and
// plain halo loads that share lines in the XCD's L2) was measured too: the halo lands 1.2 k cycles earlier, the barrier completes 3 k cycles
// later (profiles/r05_kwave_chain.txt).  Weights, biases and BN vectors are read-only for the whole launch: plain loads.
// What a layer boundary costs is then the barrier round trip instead of a kernel boundary, and the part of the next layer that does NOT
// depend on the previous one -- its first four taps of weight fragments and its bias / BN vectors, global -> registers -- is requested BEFORE
// the wait, so that latency runs under the barrier.  Against conv_kwave_bf16<8,1,8> the body also drops two workgroup barriers: wave w owns
// cin chunk w for all nine taps, so it brings in ITS OWN halo chunk (buffer loads to LDS: 32-bit offsets, bounds check = zero padding) and
// starts its taps when its own pieces have landed; and it parks its partial sums in its own chunk's LDS when its own taps are done.
// Same arithmetic in the same order as the eleven launches (tests/test_round5_gpu.py: array_equal on the ab map and every trunk activation).
// Co-residency: all workgroups must be on the chip at once.  launch mode 2 = a plain launch after an occupancy check (default); mode 1 =
// hipLaunchCooperativeKernel (the runtime checks and fails cleanly; it costs ~24 us more per launch on this runtime: a separate queue and
// cross-queue ordering).  Either way a workgroup that polls `spin_limit` times without seeing the others gives up: it sets *abort_flag
// (host-visible) and leaves; the engine then reports the forward as failed and turns the chain off for the handle -- a partitioned or shared
// device costs one forward, never a hang.
__global__ __launch_bounds__(512, 2) void conv_kwave_chain_bf16(const KwChainArgs c) {
    constexpr int NKC = 8, NW = 8;
    constexpr int PWD = 10, HR = 100;                          // 10 x 10 halo pixels of one chunk (8 x 8 tile + 1)
    constexpr int CHUNK = HR * kRowBytes;                      // 12800 bytes of LDS per cin chunk (= per wave)
    constexpr int NHJ = (HR * 8 + 63) / 64;                    // 13 LDS-DMA instructions per wave (the last one: 32 lanes)
    constexpr int PB = 4, PD = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_abort;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, g = lane >> 4;
    const int H = c.H, W = c.W;
    const int pix_bytes = NKC * kRowBytes;
    const int bid = xcd_remap_k(blockIdx.x, gridDim.x);
    const unsigned long long nwg = gridDim.x;
    const size_t w_kc_stride = (size_t)c.ncg * kWBlockBytes;
    const size_t w_tap_stride = w_kc_stride * NKC;
    const int CoutPad = c.ncg * kCoutGroup;
    if (tid == 0) s_abort = 0;

    int wo[2][2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            wo[cb][ks] = (cb * 16 + px) * kRowBytes + (((ks * 4 + g) ^ (px & 7)) * kSlotBytes);
    int hbase[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) hbase[pb] = (pb * 2 + (px >> 3) + 1) * PWD + (px & 7) + 1;
    char* const hchunk = smem + wave * CHUNK;                   // wave w owns cin chunk w: halo, all nine taps, then its partial sums
    // halo item of (lane, j): halo pixel hr = 8 j + (lane >> 3), LDS slot lane & 7 <- logical slot (lane & 7) ^ (hr & 7) = (lane & 7) ^ (lane >> 3)
    const int l8 = lane >> 3;
    const int hsrc_lane = wave * kRowBytes + (((lane & 7) ^ l8) * kSlotBytes);
    // the reduction's (pixel, 4 couts) of this thread
    const int r_p64 = tid >> 3, r_slot = tid & 7;
    const int r_off = r_p64 * kRowBytes + ((r_slot ^ (r_p64 & 7)) * kSlotBytes);
    const int r_row = (r_p64 >> 4) * 2 + ((r_p64 & 15) >> 3), r_col = r_p64 & 7;

#define IDC_KW_STAMP(k) do { if (c.stamps && tid == 0) c.stamps[((size_t)blockIdx.x * kKwChainMax + li) * 8 + (k)] = (long long)__builtin_readcyclecounter(); } while (0)
    for (int li = 0; li < c.nlayers; ++li) {
        const KwChainLayer& Ly = c.layer[li];
        const int d = Ly.d;
        IDC_KW_STAMP(0);
        const int tiles_x = ((W + d - 1) / d + 7) / 8, tiles_y = ((H + d - 1) / d + 7) / 8;
        int b = bid;
        const int bx = b % tiles_x; b /= tiles_x;
        const int by = b % tiles_y; b /= tiles_y;
        const int par = b % (d * d); b /= d * d;
        const int n = b % c.N;
        const int cg = b / c.N;
        const int Y0 = par / d + d * 8 * by, X0 = par % d + d * 8 * bx;
        const char* const wl = (const char*)Ly.wgt + (size_t)wave * w_kc_stride + (size_t)(cg >> 1) * kWBlockBytes + (cg & 1) * 32 * kRowBytes;

        // ---- weights, bias, BN do not depend on the previous layer: on their way before the grid barrier is waited for
        u32x4_k areg[PD][2][2];
        auto load_A = [&](auto slotc, int t) {
            constexpr int S = decltype(slotc)::value;
            const char* const src = wl + (size_t)(t < 8 ? t : 8) * w_tap_stride;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                kw_gload(areg[S][cb][0], src + wo[cb][0]);
                kw_gload(areg[S][cb][1], src + wo[cb][1]);
            }
        };
        auto wait_A = [&](auto slotc, auto nc) {
            constexpr int S = decltype(slotc)::value, N = decltype(nc)::value;
            kw_wait4<N>(areg[S][0][0], areg[S][0][1], areg[S][1][0], areg[S][1][1]);
        };
        load_A(std::integral_constant<int, 0>{}, 0);
        load_A(std::integral_constant<int, 1>{}, 1);
        load_A(std::integral_constant<int, 2>{}, 2);
        load_A(std::integral_constant<int, 3>{}, 3);
        const int co = (cg >> 1) * kCoutGroup + (r_slot >> 1) * 16 + (cg & 1) * 8 + (r_slot & 1) * 4;
        const bool has_bn = Ly.bn_scale != nullptr;
        const f32x4_k e_bias = *(const f32x4_k*)(Ly.bias + co);
        f32x4_k e_sc = f32x4_k{1.f, 1.f, 1.f, 1.f}, e_sh = f32x4_k{0.f, 0.f, 0.f, 0.f};
        if (has_bn) { e_sc = *(const f32x4_k*)(Ly.bn_scale + co); e_sh = *(const f32x4_k*)(Ly.bn_shift + co); }

        if (li > 0) {
            // ---- grid barrier, wait side: everybody's layer li-1 output is in memory
            if (tid < 8) {
                const unsigned long long target = (c.bar_base + (unsigned long long)li) * ((nwg + 7 - tid) >> 3);  // arrivals at counter `tid` so far
                const unsigned long long* const ctr = c.bar + tid * 16;                                        // 128 bytes apart
                unsigned spins = 0;
                while (kw_ld64_sc1(ctr) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > c.spin_limit) {
                        s_abort = 1;
                        __hip_atomic_store(c.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                }
            }
            __syncthreads();
            if (s_abort) return;                               // (uniform: every thread reads the same LDS word after the barrier)
        }
        IDC_KW_STAMP(1);

        // ---- this wave's halo chunk of this layer: global -> LDS, sc1 (agent-coherent), out-of-image pixels = the buffer's bounds check
        {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)Ly.in + (size_t)n * H * W * pix_bytes), 0,
                                                                                 H * W * pix_bytes, 0x00020000);
#pragma unroll
            for (int j = 0; j < NHJ; ++j) {
                const int hr = 8 * j + l8;
                const int hy = (hr * 205) >> 11, hx = hr - hy * PWD;       // hr / 10, hr % 10 for hr < 1029
                const int Y = Y0 + d * (hy - 1), X = X0 + d * (hx - 1);
                const bool inside = (unsigned)Y < (unsigned)H && (unsigned)X < (unsigned)W;
                const int off = inside ? (Y * W + X) * pix_bytes + hsrc_lane : (int)0x80000000;
                if (j < NHJ - 1 || lane < 32)                               // 100 pixels x 8 slots = 12.5 instructions of 64 lanes
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(hchunk + j * 1024), 16, off, 0, 0, 16);
            }
        }
        IDC_KW_STAMP(6);
        f32x4_k acc[2][PB];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < PB; ++j) acc[i][j] = f32x4_k{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // MY halo chunk has landed (the youngest loads; the weight taps and vectors were asked for first)
        IDC_KW_STAMP(2);

        auto tap = [&](auto ic) {
            constexpr int I = decltype(ic)::value;
            constexpr int YOUNGER = (PD - 1 < 8 - I ? PD - 1 : 8 - I) * 4;
            wait_A(std::integral_constant<int, I % PD>{}, std::integral_constant<int, YOUNGER>{});
            {
                constexpr int ty = I / 3, tx = I - ty * 3;
                constexpr int toff = (ty - 1) * PWD + (tx - 1);
                u32x4_k bf[PB][2];
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) {
                    const int hr = hbase[pb] + toff;
                    const int o0 = hr * kRowBytes + ((g ^ (hr & 7)) * kSlotBytes);
                    bf[pb][0] = *(const u32x4_k*)(hchunk + o0);
                    bf[pb][1] = *(const u32x4_k*)(hchunk + (o0 ^ (4 * kSlotBytes)));
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int pb = 0; pb < PB; ++pb)
                            acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_k, areg[I % PD][cb][ks]),
                                                                                  __builtin_bit_cast(bf16x8_k, bf[pb][ks]), acc[cb][pb], 0, 0, 0);
            }
            if constexpr (I + PD < 9) load_A(std::integral_constant<int, I % PD>{}, I + PD);
        };
        tap(std::integral_constant<int, 0>{}); tap(std::integral_constant<int, 1>{}); tap(std::integral_constant<int, 2>{});
        tap(std::integral_constant<int, 3>{}); tap(std::integral_constant<int, 4>{}); tap(std::integral_constant<int, 5>{});
        tap(std::integral_constant<int, 6>{}); tap(std::integral_constant<int, 7>{}); tap(std::integral_constant<int, 8>{});
        IDC_KW_STAMP(3);

        // ---- the eight partial sums of a (pixel, cout) meet in LDS in wave order: each wave parks its 8 KiB in its OWN halo chunk (its taps
        //      are done, nobody else reads that chunk), one workgroup barrier, then bias, activation, eval-BN, bf16 store
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const int p64 = q * 16 + px;
                *(f32x4_k*)(hchunk + p64 * kRowBytes + (((g * 2 + cb) ^ (p64 & 7)) * kSlotBytes)) = acc[cb][q];
            }
        __syncthreads();
        {
            f32x4_k s = *(const f32x4_k*)(smem + r_off);
#pragma unroll
            for (int w = 1; w < NW; ++w) s += *(const f32x4_k*)(smem + w * CHUNK + r_off);
            const int yy = Y0 + d * r_row, xx = X0 + d * r_col;
            f32x4_k v = s + e_bias;
            if (Ly.act == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            else if (Ly.act == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.2f * v[r];
            }
            if (has_bn) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaf(v[r], e_sc[r], e_sh[r]);
            }
            if (yy < H && xx < W) {
                const size_t o = (((size_t)n * H + yy) * W + xx) * CoutPad + co;
                const __bf16 q0 = (__bf16)v[0], q1 = (__bf16)v[1], q2 = (__bf16)v[2], q3 = (__bf16)v[3];
                uint2 pk;
                pk.x = (unsigned)__builtin_bit_cast(unsigned short, q0) | ((unsigned)__builtin_bit_cast(unsigned short, q1) << 16);
                pk.y = (unsigned)__builtin_bit_cast(unsigned short, q2) | ((unsigned)__builtin_bit_cast(unsigned short, q3) << 16);
                kw_st64_sc1((__bf16*)Ly.out + o, pk);           // agent-coherent: in memory when vmcnt says it is done
            }
        }
        IDC_KW_STAMP(4);
        if (li + 1 < c.nlayers) {
            // ---- grid barrier, arrive side: this workgroup's tile of layer li is written
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my stores are acknowledged
            __syncthreads();                                   // ... everybody's (and every wave has read the partial sums: the chunks are free)
            IDC_KW_STAMP(5);
            if (tid == 0) __hip_atomic_fetch_add(c.bar + (blockIdx.x & 7) * 16, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

#undef IDC_KW_STAMP

int conv_kwave_chain_blocks(int H, int W, int N, int ncg, int d) {
    if (H <= 0 || W <= 0 || N <= 0 || ncg <= 0 || (d != 1 && d != 2) || !wino_offsets_fit(H, W, 1, 8)) return 0;
    const long long tx = ((W + d - 1) / d + 7) / 8, ty = ((H + d - 1) / d + 7) / 8;
    const long long blocks = tx * ty * d * d * N * (ncg * 2);
    return blocks > 0 && blocks <= 0x7fffffffLL ? (int)blocks : 0;
}

int conv_kwave_chain_capacity(int device) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)conv_kwave_chain_bf16, 512, kw_lds_bytes(8, 1, 8)) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    const long long cap = (long long)per_cu * prop.multiProcessorCount;
    return cap > 0x7fffffffLL ? 0x7fffffff : (int)cap;
}

hipError_t launch_conv_kwave_chain(const KwChainArgs& c, int blocks, int mode, hipStream_t s) {
    if (c.nlayers < 2 || c.nlayers > kKwChainMax || blocks <= 0 || c.bar == nullptr || c.abort_flag == nullptr)
        return hipErrorInvalidValue;
    if (mode == 1) {
        KwChainArgs args = c;
        void* params[] = {(void*)&args};
        return hipLaunchCooperativeKernel((const void*)conv_kwave_chain_bf16, dim3((unsigned)blocks), dim3(512), params,
                                          (unsigned)kw_lds_bytes(8, 1, 8), s);
    }
    hipLaunchKernelGGL(conv_kwave_chain_bf16, dim3((unsigned)blocks), dim3(512), kw_lds_bytes(8, 1, 8), s, c);
    return hipGetLastError();
}

// the launches this kernel takes: a bf16 3x3 conv (pad = dilation 1 | 2, reading x or x[::2, ::2]) with 64 / 128 / 256 / 512 input
// channels, no shortcut sum; 32-bit source offsets (as the Winograd kernels)
bool conv_kwave_applies(const ConvArgs& a) {
    if (a.zeros == nullptr || a.wgt == nullptr) return false;
    if (a.nphase == 4)                                         // ConvTranspose 4x4 s2: conv_kwave_deconv_bf16 (shortcut sum and activation in its epilogue)
        return (a.nkc == 2 || a.nkc == 4 || a.nkc == 8) && a.ntaps == 4 && a.so == 2 && a.si == 1 && a.img_shift == nullptr &&
               a.head_w == nullptr && a.in2 == nullptr && a.pk_L == nullptr && wino_offsets_fit(a.Hs, a.Ws, 1, a.nkc);
    const int d = a.dy[8];
    return (a.nkc == 1 || a.nkc == 2 || a.nkc == 4 || a.nkc == 8) && (d == 1 || d == 2) && (a.si == 1 || a.si == 2) && a.so == 1 &&
           a.nphase == 1 && a.ntaps == 9 && a.resid == nullptr && a.head_w == nullptr && a.in2 == nullptr && a.pk_L == nullptr &&
           wino_offsets_fit(a.Hs, a.Ws, a.si, a.nkc);
}

template <int NKC, int TWB, int NW>
static hipError_t launch_kw_t(ConvArgs& a, int d, hipStream_t s) {
    a.tiles_x = ((a.Ws + d - 1) / d + 8 * TWB - 1) / (8 * TWB);
    a.tiles_y = ((a.Hs + d - 1) / d + 7) / 8;
    const long long blocks = (long long)a.tiles_x * a.tiles_y * d * d * a.N * (a.ncg * 2);
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((conv_kwave_bf16<NKC, TWB, NW>), dim3((unsigned)blocks), dim3(NW * 64), kw_lds_bytes(NKC, TWB, NW), s, a);
    return hipGetLastError();
}

hipError_t launch_conv_kwave(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    if (!conv_kwave_applies(a)) return hipErrorInvalidConfiguration;
    if (a.nphase == 4) {
        a.tiles_x = (a.Ws + 7) / 8;
        a.tiles_y = (a.Hs + 7) / 8;
        const long long blocks = (long long)a.tiles_x * a.tiles_y * a.N * (a.ncg * 4);
        if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
        if (a.nkc == 8) hipLaunchKernelGGL((conv_kwave_deconv_bf16<8>), dim3((unsigned)blocks), dim3(512), kwd_lds_bytes(8), s, a);
        else if (a.nkc == 4) hipLaunchKernelGGL((conv_kwave_deconv_bf16<4>), dim3((unsigned)blocks), dim3(512), kwd_lds_bytes(4), s, a);
        else hipLaunchKernelGGL((conv_kwave_deconv_bf16<2>), dim3((unsigned)blocks), dim3(512), kwd_lds_bytes(2), s, a);
        return hipGetLastError();
    }
    const int d = a.dy[8];
    if (a.nkc == 8) return launch_kw_t<8, 1, 8>(a, d, s);
    if (a.nkc == 4) return launch_kw_t<4, 2, 8>(a, d, s);
    if (a.nkc == 2) return launch_kw_t<2, 2, 4>(a, d, s);
    return launch_kw_t<1, 2, 4>(a, d, s);
}

hipError_t init_kernels_kw() {
    hipError_t e = hipFuncSetAttribute((const void*)conv_kwave_deconv_bf16<8>, hipFuncAttributeMaxDynamicSharedMemorySize, kwd_lds_bytes(8));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)conv_kwave_deconv_bf16<4>, hipFuncAttributeMaxDynamicSharedMemorySize, kwd_lds_bytes(4));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)conv_kwave_deconv_bf16<2>, hipFuncAttributeMaxDynamicSharedMemorySize, kwd_lds_bytes(2));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)conv_kwave_bf16<8, 1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, kw_lds_bytes(8, 1, 8));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)conv_kwave_chain_bf16, hipFuncAttributeMaxDynamicSharedMemorySize, kw_lds_bytes(8, 1, 8));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)conv_kwave_bf16<4, 2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, kw_lds_bytes(4, 2, 8));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)conv_kwave_bf16<2, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, kw_lds_bytes(2, 2, 4));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)conv_kwave_bf16<1, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, kw_lds_bytes(1, 2, 4));
}

}  // namespace idc
