// idc_conv1.hip -- model1 on the bf16 throughput path: conv1_1 with the input pack fused (conv1_1_bf16_kernel) and the whole block
// conv1_1 + conv1_2 in one launch (conv1_block_fused_t).  models/pytorch/model.py:13-17,139-148.
#include <stdlib.h>
#include <type_traits>

#include "idc_kernels.h"

#include "idc_layout.h"

#include "idc_common.hip.h"
#include "idc_split.hip.h"

namespace idc {

// one 32x32x16 MFMA step on two 16-byte fragments: bf16 or fp16 operands, fp32 accumulate
template <bool F16>
__device__ __forceinline__ f32x16 mma_32x32x16(const u32x4& w, const u32x4& x, const f32x16& c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_m, w), __builtin_bit_cast(f16x8_m, x), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), c, 0, 0, 0);
}

constexpr int conv1_block_lds(int nw, int rpw, bool lw) { return 34 * (nw * rpw + 2) * 128 + 36 * (nw * rpw + 4) * 8 + (lw ? 2 * kWBlockBytes : 0); }

#ifdef IDC_AB_PARTNERS      // conv1_1 alone at throughput size: "fuse_conv1" = 0 A/B only (the default library keeps it on conv_igemm)
// ------------------------------------------------------------------------------------------------
// conv1_1_bf16_kernel: model1.0 (4 -> 64 channels, 3x3, model.py:13) with the input pack (model.py:139-148) fused,
// throughput form.  One workgroup = a 32x32 tile of one image x all 64 output channels; 8 waves x 4 pixel rows.
// The (32+2)^2 input patch is normalised once into LDS as float4 (L, a, b, mask); every lane builds its own MFMA B
// fragments from it (K index = tap*4 + channel, 36 of 64 used: three k16 steps), the A fragments come straight from
// the packed weights (layout 1 rows, read in the row order of the 32x32 D layout so that a lane ends up with 32
// consecutive couts), so there is no im2col buffer, one barrier, and the 268 MB output is the only HBM stream that
// matters.  Epilogue = the bf16-transpose one of conv_igemm_v2.  (The small-tile conv_igemm path keeps the batch-1 case.)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void conv1_1_bf16_kernel(const ConvArgs a) {
    constexpr int PW = 34;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* const patch = (float4*)smem;                       // [34][34]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 31, h = lane >> 5;
    const int Hs = a.Hs, Ws = a.Ws;
    const int ntx = (Ws + 31) >> 5, nty = (Hs + 31) >> 5;
    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int txi = b % ntx; b /= ntx;
    const int tyi = b % nty;
    const int n = b / nty;
    const int ty0 = tyi * 32, tx0 = txi * 32;
    {
        const size_t hw = (size_t)Hs * Ws;
        const float* const pL = a.pk_L + (size_t)n * hw;
        const float* const pA = a.pk_ab + (size_t)n * 2 * hw;
        const float* const pM = a.pk_mask + (size_t)n * hw;
        const float rl = 1.0f / a.pk_ldiv, ra = 1.0f / a.pk_abdiv;
        for (int idx = tid; idx < PW * PW; idx += 512) {
            const int py = idx / PW, pxx = idx - py * PW;
            const int yy = ty0 - 1 + py, xx = tx0 - 1 + pxx;
            float4 c = float4{0.f, 0.f, 0.f, 0.f};
            if ((unsigned)yy < (unsigned)Hs && (unsigned)xx < (unsigned)Ws) {
                const size_t p = (size_t)yy * Ws + xx;
                c = float4{pL[p] / a.pk_ldiv, pA[p] / a.pk_abdiv, pA[hw + p] / a.pk_abdiv, pM[p] * a.pk_mmul - a.pk_mcent};
            }
            patch[idx] = c;
        }
        (void)rl; (void)ra;
    }
    // A fragments: MFMA row rho = px of block mi is cout hh*32 + mi*16 + r with r = (rho>>3)*4 + (rho&3), hh = (rho>>2)&1
    u32x4 wf[3][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int c = ((px >> 2) & 1) * 32 + mi * 16 + (px >> 3) * 4 + (px & 3);
        const int lam = ((c >> 2) & 3) * 16 + (c >> 4) * 4 + (c & 3);          // layout-1 row of cout c (idc_layout.h)
#pragma unroll
        for (int kk = 0; kk < 3; ++kk)
            wf[kk][mi] = *(const u32x4*)((const char*)a.wgt + lam * kRowBytes + (((kk * 2 + h) ^ swz(lam)) * kSlotBytes));
    }
    f32x16 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        f32x16 b16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bq = *(const float4*)(a.bias + h * 32 + mi * 16 + q * 4);
            b16[q * 4 + 0] = bq.x; b16[q * 4 + 1] = bq.y; b16[q * 4 + 2] = bq.z; b16[q * 4 + 3] = bq.w;
        }
#pragma unroll
        for (int pj = 0; pj < 4; ++pj) acc[mi][pj] = b16;
    }
    __syncthreads();
#pragma unroll
    for (int pj = 0; pj < 4; ++pj) {
        const int base = (wave * 4 + pj) * PW + px;            // patch index of tap (ky=0, kx=0) for this site
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            // lane half h holds K = kk*16 + h*8 .. +7 = taps 4kk+2h, 4kk+2h+1 (x 4 channels); taps >= 9 are zero padding
            const int t0a = 4 * kk, t0b = 4 * kk + 2;          // first tap for h = 0 / h = 1
            const int o0 = h ? (t0b / 3) * PW + t0b % 3 : (t0a / 3) * PW + t0a % 3;
            const int o1 = h ? ((t0b + 1) / 3) * PW + (t0b + 1) % 3 : ((t0a + 1) / 3) * PW + (t0a + 1) % 3;
            const bool z0 = h ? t0b >= 9 : t0a >= 9, z1 = h ? t0b + 1 >= 9 : t0a + 1 >= 9;
            const float4 c0 = z0 ? float4{0.f, 0.f, 0.f, 0.f} : patch[base + o0];
            const float4 c1 = z1 ? float4{0.f, 0.f, 0.f, 0.f} : patch[base + o1];
            const u32x4 xf = u32x4{pack_bf16x2(c0.x, c0.y), pack_bf16x2(c0.z, c0.w), pack_bf16x2(c1.x, c1.y), pack_bf16x2(c1.z, c1.w)};
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                acc[mi][pj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[kk][mi]),
                                                                      __builtin_bit_cast(bf16x8, xf), acc[mi][pj], 0, 0, 0);
        }
    }
    // epilogue: (ReLU,) round, transpose [32 px][64 couts] bf16 through a wave-private LDS tile, whole-line stores
    char* const tb16 = smem + PW * PW * 16 + wave * 4096;
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    const int rr = lane >> 3, cc = lane & 7;
    const int CoutPad = a.ncg * kCoutGroup;
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
        const int sy = ty0 + wave * 4 + pj;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 8 + rr;
            const uint4 o = *(const uint4*)(tb16 + row * 128 + ((cc ^ (row & 7)) * 16));
            const int sx = tx0 + row;
            if (sy < Hs && sx < Ws)
                *(uint4*)((unsigned short*)a.out + (((size_t)n * Hs + sy) * Ws + sx) * CoutPad + cc * 8) = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// conv1_1 (kConvIm2col) in its throughput form; needs bf16, the fused-pack planes, ReLU/none and no BN / shortcut
hipError_t launch_conv1_1_bf16(const ConvArgs& a, hipStream_t s) {
    if (a.pk_L == nullptr || a.bn_scale != nullptr || a.resid != nullptr || a.out_f32 || a.act == 2 || a.ncg != 1 || a.ksplit > 1)
        return hipErrorInvalidConfiguration;
    const long long blocks = (long long)((a.Ws + 31) / 32) * ((a.Hs + 31) / 32) * a.N;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(conv1_1_bf16_kernel, dim3((unsigned)blocks), dim3(512), 34 * 34 * 16 + 8 * 4096, s, a);
    return hipGetLastError();
}
#else
hipError_t launch_conv1_1_bf16(const ConvArgs&, hipStream_t) { return hipErrorInvalidConfiguration; }
#endif

// ------------------------------------------------------------------------------------------------
// conv1_block_fused: model1 = conv1_1 (4 -> 64, ReLU) -> conv1_2 (64 -> 64, ReLU, eval-BN) of one 32x32 tile in one
// workgroup (model.py:13-17): conv1_1's 268 MB output never exists in HBM.
//   phase 0: the 36x36 input patch, normalised (model.py:139-148) and rounded to bf16, 8 B per pixel, into LDS;
//   phase 1: conv1_1 on the 34x34 halo sites conv1_2 needs: a wave takes 32 consecutive halo sites per MFMA column
//            block, builds its B fragments from the patch (K = tap*4 + channel), and writes ReLU(.) as bf16 straight
//            into the halo tile in the layout conv_igemm_v2 reads (128-B rows, slot ^ swz2(row)); sites outside the
//            image are conv1_2's zero padding and are written as zeros;
//   phase 2: conv1_2 = 9 taps x 4 k16 steps x 8 MFMAs per wave over the static halo tile, A fragments straight from
//            the packed weights (global -> registers, one tap ahead): no weights in LDS, no barrier in the K loop;
//   phase 3: ReLU + eval-BN in the MFMA layout, bf16 LDS transpose, whole-line stores.
// LDS: 34*34*128 B halo + 36*36*8 B patch = 154.6 KiB, one workgroup (8 waves) per CU.
// ------------------------------------------------------------------------------------------------
// NW waves x RPW pixel rows each: <8,4> = the 32x32 tile (one workgroup per CU), <4,2> = a 32x8 tile whose 46 KiB of LDS and
// <= 168 registers let THREE workgroups share a CU, so that one's patch / conv1_1 / store phases run under another's conv1_2 MFMAs
// (conv1_1 is recomputed on 34x10 sites per 32x8 outputs: 33 % extra instead of 13 %, of a conv that is 2 % of the block's MACs).
// LW (round 4): conv1_2's weight tiles go through a 2-slot LDS ring by LDS-DMA (one 8 KiB tile per tap, shared by the workgroup's
// waves, requested one tap ahead, one barrier per tap) instead of global -> registers per wave: the L2 -> CU stream of the phase drops
// by the number of waves, which is what kept the small tiles from paying (profiles/r03_conv1_tile8.txt); <4,3,true> = 32x12 tile at
// exactly 80 KiB and <4,2,true> = 32x8 at 62 KiB: two workgroups per CU.
// F16 (round 6, IDC_FP16's fast path): the same block on fp16 operands -- v_mfma_f32_32x32x16_f16, the patch / the conv1_1 tile / the output as fp16 (clamped
// to the fp16 range); a.wgt / a.wgt2 = the fp16 layout-1 images.  Everything else textually the bf16 block.
template <int NW, int RPW, bool LW, bool F16>
__device__ __forceinline__ void conv1_block_body(const ConvArgs& a) {
    constexpr int NT = NW * 64, TH = NW * RPW;
    constexpr int HW_ = 34, HH_ = TH + 2, PW = 36, PH = TH + 4, NSITE = HW_ * HH_;
    constexpr int HALO_BYTES = NSITE * kRowBytes;              // 147,968
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const halo = smem;
    uint2* const patch = (uint2*)(smem + HALO_BYTES);          // [TH + 4][36] x 4 bf16
    char* const wring = smem + HALO_BYTES + PW * PH * 8;       // LW: 2 x 8 KiB weight tiles of conv1_2
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 31, h = lane >> 5;
    const int Hs = a.Hs, Ws = a.Ws;
    const int ntx = (Ws + 31) >> 5, nty = (Hs + TH - 1) / TH;
    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int txi = b % ntx; b /= ntx;
    const int tyi = b % nty;
    const int n = b / nty;
    const int ty0 = tyi * TH, tx0 = txi * 32;
    // conv1_2's tap-t tile (64 couts x 64 cin) -> ring slot, RE-LAID on the way: the blob holds the layout-1 image (rows in the 16x16
    // MFMA's order, slot ^ (row & 7)), whose rows a 32x32 A fragment reads 8-way bank-conflicted (50 % of this kernel's LDS cycles in its
    // first form, profiles/r04b_pmc_sq_summary.txt); the LDS-DMA's per-lane source addresses gather it into conv_igemm_v2's layout instead --
    // LDS row rho = the MFMA row (mi*32 + px), slot ^ ((rho >> 1) & 7) -- still one whole 128-byte line per 8 lanes.
    constexpr int W2_ITEMS = kWBlockBytes / (NT * kSlotBytes);
    static_assert(!LW || kWBlockBytes % (NT * kSlotBytes) == 0, "tile must split evenly");
    int w2_src[W2_ITEMS];
#pragma unroll
    for (int j = 0; j < W2_ITEMS; ++j) {
        const int i = tid + j * NT, rho = i >> 3, sphys = i & 7, px_ = rho & 31, mi_ = rho >> 5;
        const int c = ((px_ >> 2) & 1) * 32 + mi_ * 16 + (px_ >> 3) * 4 + (px_ & 3);         // cout of MFMA row rho (as lam[] below)
        const int lr = ((c >> 2) & 3) * 16 + (c >> 4) * 4 + (c & 3);                          // its row in the layout-1 block
        w2_src[j] = lr * kRowBytes + (((sphys ^ swz2(rho)) ^ swz(lr)) * kSlotBytes);
    }
    auto dma_w2 = [&](int t, int slot) {
        const char* const src = (const char*)a.wgt2 + (size_t)t * kWBlockBytes;
        char* dst = wring + slot * kWBlockBytes + wave * 64 * kSlotBytes;
#pragma unroll
        for (int j = 0; j < W2_ITEMS; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + w2_src[j]),
                                             (__attribute__((address_space(3))) void*)(dst + j * NT * kSlotBytes), 16, 0, 0);
    };
    if constexpr (LW) {
        dma_w2(0, 0);                                          // lands under phases 0 and 1
        if (a.warm && wave == 0) idc_warm_own_code(wring + kWBlockBytes, lane, 64);   // 8 of this kernel's 7.6-9.7 KB (other kernels follow in this code object); scratch: ring slot 1 (rewritten by tap 1's tile)
    }
    IDC_STAMP(0);
    // ---- phase 0 -------------------------------------------------------------------------------
    {
        const size_t hw = (size_t)Hs * Ws;
        const float* const pL = a.pk_L + (size_t)n * hw;
        const float* const pA = a.pk_ab + (size_t)n * 2 * hw;
        const float* const pM = a.pk_mask + (size_t)n * hw;
        // every lane's (up to P_ITEMS) pixels: all twelve plane reads in flight before the first is used (round 5: the loop form waited for
        // each pixel's four loads in turn -- three HBM round trips per tile, 5.2 k of a tile's 33 k ticks -> 3.6 k; tools/ablate v2 = 7)
        constexpr int P_ITEMS = (PW * PH + NT - 1) / NT;
        float vl[P_ITEMS], va[P_ITEMS], vb[P_ITEMS], vm[P_ITEMS];
        bool ok[P_ITEMS];
#pragma unroll
        for (int j = 0; j < P_ITEMS; ++j) {
            const int idx = tid + j * NT;
            const int py = idx / PW, pxx = idx - py * PW;
            const int yy = ty0 - 2 + py, xx = tx0 - 2 + pxx;
            ok[j] = idx < PW * PH && (unsigned)yy < (unsigned)Hs && (unsigned)xx < (unsigned)Ws;
            const size_t p = ok[j] ? (size_t)yy * Ws + xx : 0;
            vl[j] = pL[p]; va[j] = pA[p]; vb[j] = pA[hw + p]; vm[j] = pM[p];
        }
#pragma unroll
        for (int j = 0; j < P_ITEMS; ++j) {
            const int idx = tid + j * NT;
            uint2 c = uint2{0u, 0u};
            if (ok[j])
                c = uint2{pack16x2_m<F16>(vl[j] / a.pk_ldiv, va[j] / a.pk_abdiv), pack16x2_m<F16>(vb[j] / a.pk_abdiv, vm[j] * a.pk_mmul - a.pk_mcent)};
            if (idx < PW * PH) patch[idx] = c;
        }
    }
    IDC_STAMP(1);
    // A-fragment row of this lane: MFMA row rho = px of block mi is cout hh*32 + mi*16 + r (see conv1_1_bf16_kernel)
    int lam[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int c = ((px >> 2) & 1) * 32 + mi * 16 + (px >> 3) * 4 + (px & 3);
        lam[mi] = ((c >> 2) & 3) * 16 + (c >> 4) * 4 + (c & 3);
    }
    // ---- phase 1 -------------------------------------------------------------------------------
    {
        u32x4 wf[3][2];
        f32x16 b1[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int kk = 0; kk < 3; ++kk)
                wf[kk][mi] = *(const u32x4*)((const char*)a.wgt + lam[mi] * kRowBytes + (((kk * 2 + h) ^ swz(lam[mi])) * kSlotBytes));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *(const float4*)(a.bias + h * 32 + mi * 16 + q * 4);
                b1[mi][q * 4 + 0] = bq.x; b1[mi][q * 4 + 1] = bq.y; b1[mi][q * 4 + 2] = bq.z; b1[mi][q * 4 + 3] = bq.w;
            }
        }
        __syncthreads();                                       // patch complete
        IDC_STAMP(5);
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        for (int g = wave; g * 32 < NSITE; g += NW) {
            const int sidx = g * 32 + px;                      // halo site = halo row of the tile
            const int hy = sidx / HW_, hx = sidx - hy * HW_;
            const bool live = sidx < NSITE;
            const int yy = ty0 - 1 + hy, xx = tx0 - 1 + hx;
            const bool inimg = live && (unsigned)yy < (unsigned)Hs && (unsigned)xx < (unsigned)Ws;
            const int pbase = live ? hy * PW + hx : 0;         // patch index of tap (0,0)
            f32x16 c1[2] = {b1[0], b1[1]};
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                const int t0a = 4 * kk, t0b = 4 * kk + 2;
                const int o0 = h ? (t0b / 3) * PW + t0b % 3 : (t0a / 3) * PW + t0a % 3;
                const int o1 = h ? ((t0b + 1) / 3) * PW + (t0b + 1) % 3 : ((t0a + 1) / 3) * PW + (t0a + 1) % 3;
                const bool z0 = h ? t0b >= 9 : t0a >= 9, z1 = h ? t0b + 1 >= 9 : t0a + 1 >= 9;
                const uint2 q0 = z0 ? uint2{0u, 0u} : patch[pbase + o0];
                const uint2 q1 = z1 ? uint2{0u, 0u} : patch[pbase + o1];
                const u32x4 xf = u32x4{q0.x, q0.y, q1.x, q1.y};
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    c1[mi] = mma_32x32x16<F16>(wf[kk][mi], xf, c1[mi]);
            }
            if (live) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    unsigned pk[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        pk[e] = pack16x2_m<F16>(c1[mi][2 * e], c1[mi][2 * e + 1]);
                        if (a.act == 1)
                            pk[e] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pk[e]), s16x2{0, 0}));
                        if (!inimg) pk[e] = 0u;
                    }
                    const int s0 = h * 4 + mi * 2;
                    *(uint4*)(halo + sidx * kRowBytes + ((s0 ^ swz2(sidx)) * kSlotBytes)) = uint4{pk[0], pk[1], pk[2], pk[3]};
                    *(uint4*)(halo + sidx * kRowBytes + (((s0 + 1) ^ swz2(sidx)) * kSlotBytes)) = uint4{pk[4], pk[5], pk[6], pk[7]};
                }
            }
        }
    }
    IDC_STAMP(2);
    // ---- phase 2 -------------------------------------------------------------------------------
    f32x16 acc[2][RPW];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        f32x16 b16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bq = *(const float4*)(a.head_b + h * 32 + mi * 16 + q * 4);      // conv1_2's bias rides in head_b
            b16[q * 4 + 0] = bq.x; b16[q * 4 + 1] = bq.y; b16[q * 4 + 2] = bq.z; b16[q * 4 + 3] = bq.w;
        }
#pragma unroll
        for (int pj = 0; pj < RPW; ++pj) acc[mi][pj] = b16;
    }
  if constexpr (!LW) {
    u32x4 wcur[4][2], wnxt[4][2];
    auto load_w = [&](int t, u32x4 (&w)[4][2]) {
        const char* const base = (const char*)a.wgt2 + (size_t)t * kWBlockBytes;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                w[kk][mi] = *(const u32x4*)(base + lam[mi] * kRowBytes + (((kk * 2 + h) ^ swz(lam[mi])) * kSlotBytes));
    };
    load_w(0, wcur);
    __syncthreads();                                           // conv1_1 tile complete
#pragma unroll 1
    for (int t = 0; t < 9; ++t) {
        if (t + 1 < 9) load_w(t + 1, wnxt);
        const int dy = t / 3 - 1, dx = t - (t / 3) * 3 - 1;
        int xaddr[RPW];
#pragma unroll
        for (int pj = 0; pj < RPW; ++pj) {
            const int xr = (wave * RPW + pj + 1 + dy) * HW_ + (px + 1 + dx);
            xaddr[pj] = xr * kRowBytes + ((h ^ swz2(xr)) * kSlotBytes);
        }
        // 2-stage software pipeline over the four k16 steps, issue order pinned as in conv_igemm_v2 (only B comes from LDS)
        u32x4 xfA[RPW], xfB[RPW];
        auto read_x = [&](int kk, u32x4 (&xf)[RPW]) {
#pragma unroll
            for (int pj = 0; pj < RPW; ++pj) xf[pj] = *(const u32x4*)(halo + (xaddr[pj] ^ (kk * 2 * kSlotBytes)));
        };
        auto mma8 = [&](int kk, const u32x4 (&xf)[RPW]) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int pj = 0; pj < RPW; ++pj)
                    acc[mi][pj] = mma_32x32x16<F16>(wcur[kk][mi], xf[pj], acc[mi][pj]);
        };
#define IDC_C1_INTERLEAVE()                                                           \
    _Pragma("unroll") for (int q_ = 0; q_ < RPW; ++q_) {                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                            \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                            \
    }                                                                                 \
    __builtin_amdgcn_sched_group_barrier(0x008, RPW, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_x(0, xfA);
        __builtin_amdgcn_sched_group_barrier(0x100, RPW, 0);
        read_x(1, xfB);
        mma8(0, xfA);
        IDC_C1_INTERLEAVE()
        read_x(2, xfA);
        mma8(1, xfB);
        IDC_C1_INTERLEAVE()
        read_x(3, xfB);
        mma8(2, xfA);
        IDC_C1_INTERLEAVE()
        mma8(3, xfB);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * RPW, 0);
#undef IDC_C1_INTERLEAVE
        if (t + 1 < 9) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) wcur[kk][mi] = wnxt[kk][mi];
        }
    }
  } else {
    // LW: per tap one vmcnt(0) + barrier publishes the tile requested a tap ago; A fragments (2 per k16 step) and B fragments (RPW) are
    // read a step ahead of their MFMAs, as above
    const int wlam0 = px * kRowBytes + ((h ^ swz2(px)) * kSlotBytes), wlam1 = wlam0 + 32 * kRowBytes;   // rows mi*32 + px of the re-laid tile
#pragma unroll 1
    for (int t = 0; t < 9; ++t) {
        const char* const wcur_ = wring + (t & 1) * kWBlockBytes;
        if (t == 4) IDC_STAMP(9);                               // (tools/ablate v2 = 7: where a tap's time goes)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // my pieces of this tap's tile
        if (t == 4) IDC_STAMP(10);
        __syncthreads();                                        // everybody's (t = 0: also the conv1_1 tile); everybody left the other slot
        if (t == 4) IDC_STAMP(11);
        if (t == 5) IDC_STAMP(12);
        if (t + 1 < 9) dma_w2(t + 1, (t + 1) & 1);
        const int dy = t / 3 - 1, dx = t - (t / 3) * 3 - 1;
        int xaddr[RPW];
#pragma unroll
        for (int pj = 0; pj < RPW; ++pj) {
            const int xr = (wave * RPW + pj + 1 + dy) * HW_ + (px + 1 + dx);
            xaddr[pj] = xr * kRowBytes + ((h ^ swz2(xr)) * kSlotBytes);
        }
        u32x4 xfA[RPW], xfB[RPW], wA[2], wB[2];
        auto read_f = [&](int kk, u32x4 (&xf)[RPW], u32x4 (&wf)[2]) {
            wf[0] = *(const u32x4*)(wcur_ + (wlam0 ^ (kk * 2 * kSlotBytes)));
            wf[1] = *(const u32x4*)(wcur_ + (wlam1 ^ (kk * 2 * kSlotBytes)));
#pragma unroll
            for (int pj = 0; pj < RPW; ++pj) xf[pj] = *(const u32x4*)(halo + (xaddr[pj] ^ (kk * 2 * kSlotBytes)));
        };
        auto mmaL = [&](const u32x4 (&wf)[2], const u32x4 (&xf)[RPW]) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int pj = 0; pj < RPW; ++pj)
                    acc[mi][pj] = mma_32x32x16<F16>(wf[mi], xf[pj], acc[mi][pj]);
        };
#define IDC_C1L_INTERLEAVE()                                                          \
    _Pragma("unroll") for (int q_ = 0; q_ < RPW + 2; ++q_) {                         \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                            \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                            \
    }                                                                                 \
    if constexpr (RPW > 2) __builtin_amdgcn_sched_group_barrier(0x008, RPW - 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_f(0, xfA, wA);
        __builtin_amdgcn_sched_group_barrier(0x100, RPW + 2, 0);
        read_f(1, xfB, wB);
        mmaL(wA, xfA);
        IDC_C1L_INTERLEAVE()
        read_f(2, xfA, wA);
        mmaL(wB, xfB);
        IDC_C1L_INTERLEAVE()
        read_f(3, xfB, wB);
        mmaL(wA, xfA);
        IDC_C1L_INTERLEAVE()
        mmaL(wB, xfB);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * RPW, 0);
#undef IDC_C1L_INTERLEAVE
    }
  }
    // ---- phase 3 -------------------------------------------------------------------------------
    IDC_STAMP(3);
    __syncthreads();                                           // every wave left the halo tile
    IDC_STAMP(6);
    char* const tb16 = smem + wave * 4096;
    const int rr = lane >> 3, cc = lane & 7;
    const int CoutPad = a.ncg * kCoutGroup;
    const bool has_bn = a.bn_scale != nullptr;
    f32x16 bsc[2], bsh[2];
    if (has_bn) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 s4 = *(const float4*)(a.bn_scale + h * 32 + mi * 16 + q * 4);
                const float4 t4 = *(const float4*)(a.bn_shift + h * 32 + mi * 16 + q * 4);
                bsc[mi][q * 4 + 0] = s4.x; bsc[mi][q * 4 + 1] = s4.y; bsc[mi][q * 4 + 2] = s4.z; bsc[mi][q * 4 + 3] = s4.w;
                bsh[mi][q * 4 + 0] = t4.x; bsh[mi][q * 4 + 1] = t4.y; bsh[mi][q * 4 + 2] = t4.z; bsh[mi][q * 4 + 3] = t4.w;
            }
    }
    // (as conv_igemm_v2p's epilogue, round 5: one body per BN setting chosen once -- a run-time `if` per element was a uniform branch per packed pair --,
    //  a row's four transposed lines read BEFORE the first store's bounds check, one 64-bit base per lane with 32-bit strides)
    unsigned short* const out00 = (unsigned short*)a.out + (((size_t)n * Hs + ty0 + wave * RPW) * Ws + tx0 + rr) * CoutPad + cc * 8;
    auto rows = [&](auto bn_c) __attribute__((always_inline)) {
        constexpr bool BN = decltype(bn_c)::value;
#pragma unroll
        for (int pj = 0; pj < RPW; ++pj) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                unsigned pk[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v0 = fmaxf(acc[mi][pj][2 * e], 0.f), v1 = fmaxf(acc[mi][pj][2 * e + 1], 0.f);
                    if constexpr (BN) { v0 = fmaf(v0, bsc[mi][2 * e], bsh[mi][2 * e]); v1 = fmaf(v1, bsc[mi][2 * e + 1], bsh[mi][2 * e + 1]); }
                    pk[e] = pack16x2_m<F16>(v0, v1);
                }
                const int s0 = h * 4 + mi * 2;
                *(uint4*)(tb16 + px * 128 + ((s0 ^ (px & 7)) * 16)) = uint4{pk[0], pk[1], pk[2], pk[3]};
                *(uint4*)(tb16 + px * 128 + (((s0 + 1) ^ (px & 7)) * 16)) = uint4{pk[4], pk[5], pk[6], pk[7]};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int sy = ty0 + wave * RPW + pj;
            auto line = [&](int i) { const int row = i * 8 + rr; return *(const uint4*)(tb16 + row * 128 + ((cc ^ (row & 7)) * 16)); };
            const uint4 o0 = line(0), o1 = line(1), o2 = line(2), o3 = line(3);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            auto put = [&](int i, const uint4& o) {
                const int sx = tx0 + i * 8 + rr;
                if (sy < Hs && sx < Ws) *(uint4*)(out00 + (pj * Ws + i * 8) * CoutPad) = o;
            };
            put(0, o0); put(1, o1); put(2, o2); put(3, o3);
        }
    };
    if (has_bn) rows(std::true_type{}); else rows(std::false_type{});
    IDC_STAMP(4);
#ifdef IDC_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    IDC_STAMP(7);
    if (tid == 0) g_idc_dbg[(size_t)blockIdx.x * 16 + 8] = (long long)__builtin_amdgcn_s_getreg(6 | (0 << 6) | (31 << 11));   // HW_REG_LDS_ALLOC: which of its CU's two LDS slots the workgroup got
#endif
}

template <int NW, int RPW, bool LW>
__global__ __launch_bounds__(NW * 64, (NW == 8 || LW) ? 2 : 3) void conv1_block_fused_t(const ConvArgs a) { conv1_block_body<NW, RPW, LW, false>(a); }
template <int NW, int RPW, bool LW>      // IDC_FP16: the block on fp16 operands
__global__ __launch_bounds__(NW * 64, (NW == 8 || LW) ? 2 : 3) void conv1_block_fused_th(const ConvArgs a) { conv1_block_body<NW, RPW, LW, true>(a); }

// model1 (conv1_1 + conv1_2) in one launch.  `a` = conv1_1's arguments (fused-pack planes, layout-1 weights, bias, act)
// with conv1_2's riding in: wgt2 = its layout-1 weights (9 taps x 8 KiB), head_b = its bias, bn_scale/bn_shift = its
// eval-BN affine, out = its output.  conv1_2 is ReLU + (optional) BN, 64 -> 64.
// Tile (round 4, profiles/r04d_*): conv1_block_fused_t<4,3,true> -- 32x12 pixels, conv1_2's weight tiles through an LDS ring, two workgroups per CU --
// at N = 32 (same-box conv1 block 0.2315 -> 0.2121 ms against the 32x32 tile); the click path's too-few-tiles case takes the 32x8 form
// <4,2,true>.  IDC_C1_LW=0 keeps the round-2 32x32 tile (<8,4,false>: weights global -> registers per wave) as the A/B partner; the
// ring-less 32x8 tile <4,2,false> and the "conv1_lw" option were retired in round 5 (bit-identical results in every form).
static const int g_c1_lw = idc_env_int("IDC_C1_LW", 3);

// the fused block uses more than the default 64 KiB of dynamic LDS (set per device: this runs for every handle)
static hipError_t init_kernels_conv1_split();      // (conv1_2_split_kernel, defined below)
hipError_t init_kernels_conv1() {
    hipError_t e = init_kernels_conv1_split();
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)conv1_block_fused_t<4, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, conv1_block_lds(4, 2, true));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)conv1_block_fused_t<4, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, conv1_block_lds(4, 3, true));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)conv1_block_fused_th<4, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, conv1_block_lds(4, 2, true));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)conv1_block_fused_th<4, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, conv1_block_lds(4, 3, true));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)conv1_block_fused_t<8, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, conv1_block_lds(8, 4, false));
}

hipError_t launch_conv1_block(const ConvArgs& a, hipStream_t s) {
    if (a.pk_L == nullptr || a.wgt2 == nullptr || a.head_b == nullptr || a.ncg != 1 || a.out_f32 || a.resid != nullptr)
        return hipErrorInvalidConfiguration;
    const bool big = a.tiles_y != 8;                        // a.tiles_y = the engine's request: 8 on the batch-1 click path (too few 32x32 tiles)
    const int th = !big ? 8 : g_c1_lw ? 12 : 32;
    const long long blocks = (long long)((a.Ws + 31) / 32) * ((a.Hs + th - 1) / th) * a.N;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    if (a.split_f16) {                                       // IDC_FP16's fast path: the ring forms only
        if (!big) hipLaunchKernelGGL((conv1_block_fused_th<4, 2, true>), dim3((unsigned)blocks), dim3(256), conv1_block_lds(4, 2, true), s, a);
        else hipLaunchKernelGGL((conv1_block_fused_th<4, 3, true>), dim3((unsigned)((long long)((a.Ws + 31) / 32) * ((a.Hs + 11) / 12) * a.N)), dim3(256),
                                conv1_block_lds(4, 3, true), s, a);
        return hipGetLastError();
    }
    if (!big) hipLaunchKernelGGL((conv1_block_fused_t<4, 2, true>), dim3((unsigned)blocks), dim3(256), conv1_block_lds(4, 2, true), s, a);
    else if (g_c1_lw) hipLaunchKernelGGL((conv1_block_fused_t<4, 3, true>), dim3((unsigned)blocks), dim3(256), conv1_block_lds(4, 3, true), s, a);
    else hipLaunchKernelGGL((conv1_block_fused_t<8, 4, false>), dim3((unsigned)blocks), dim3(512), conv1_block_lds(8, 4, false), s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// conv1_1_split_kernel (round 6): model1.0 (4 -> 64, 3x3, ReLU; model.py:13,139-148) as the EXACT-FP32 ISLAND of the operand-split precisions at throughput
// size.  conv1_1 is 0.2 % of the network's MACs and stays fp32 (K = 36 products of normalised inputs; splitting them would cost more than it saves), but its
// output is the largest tensor of the forward: conv_igemm<float> -- a generic small tile that pads K to 64, stages im2col rows through LDS and stores 32-byte
// pieces -- took 0.48 ms at N = 32 for 9.7 GFLOP and 537 MB.  Here: workgroup = 32 x 16 pixels x 64 couts, 4 waves x 4 pixel rows; the 34 x 18 input patch is
// normalised once into LDS as float4 (L, a, b, mask); v_mfma_f32_16x16x4_f32 with k = the four channels of ONE tap (K index = tap*4 + channel, as the packed
// image orders it): the B operand of a (site tile, tap) is one ds_read_b32 per lane straight from the patch -- no im2col rows -- reused by the four cout
// blocks, the A operands (36 floats per lane) sit in registers for the whole tile; accumulators in conv_igemm_v2m's layout (lane (site r16, group g16)
// register j of acc[mi][pt] = cout g16*16 + mi*4 + j), so the epilogue IS split_epilogue: bias, ReLU, hi = rne(v), next = rne(v - hi) ..., every plane
// through the wave-private transpose tile, whole 128-byte line stores.  Nine taps in ascending order into zero accumulators, bias after: fp32 throughout.
template <bool F16>
__global__ __launch_bounds__(256, 2) void conv1_1_split_kernel(const ConvArgs a) {
    constexpr int PW = 34, PH = 18;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [4 x 4096 B transpose tiles][34 x 18 float4 patch]
    float* const patch = (float*)(smem + 4 * 4096);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g16 = lane >> 4;
    const int Hs = a.Hs, Ws = a.Ws;
    const int ntx = (Ws + 31) >> 5, nty = (Hs + 15) >> 4;
    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int txi = b % ntx; b /= ntx;
    const int tyi = b % nty;
    const int n = b / nty;
    const int ty0 = tyi * 16, tx0 = txi * 32;
    {
        const size_t hw = (size_t)Hs * Ws;
        const float* const pL = a.pk_L + (size_t)n * hw;
        const float* const pA = a.pk_ab + (size_t)n * 2 * hw;
        const float* const pM = a.pk_mask + (size_t)n * hw;
        constexpr int P_ITEMS = (PW * PH + 255) / 256;
        float vl[P_ITEMS], va[P_ITEMS], vb[P_ITEMS], vm[P_ITEMS];
        bool ok[P_ITEMS];
#pragma unroll
        for (int j = 0; j < P_ITEMS; ++j) {                          // all plane reads in flight before the first is used
            const int idx = tid + j * 256;
            const int py = idx / PW, pxx = idx - py * PW;
            const int yy = ty0 - 1 + py, xx = tx0 - 1 + pxx;
            ok[j] = idx < PW * PH && (unsigned)yy < (unsigned)Hs && (unsigned)xx < (unsigned)Ws;
            const size_t p = ok[j] ? (size_t)yy * Ws + xx : 0;
            vl[j] = pL[p]; va[j] = pA[p]; vb[j] = pA[hw + p]; vm[j] = pM[p];
        }
#pragma unroll
        for (int j = 0; j < P_ITEMS; ++j) {
            const int idx = tid + j * 256;
            float4 c = float4{0.f, 0.f, 0.f, 0.f};                  // outside the image: conv1_1's zero padding
            if (ok[j]) c = float4{vl[j] / a.pk_ldiv, va[j] / a.pk_abdiv, vb[j] / a.pk_abdiv, vm[j] * a.pk_mmul - a.pk_mcent};
            if (idx < PW * PH) ((float4*)patch)[idx] = c;
        }
    }
    // A operands from the packed fp32 layout-1 image (idc_layout.h; two chunks of 32 K values, K = tap*4 + channel): MFMA row m of block mi is image row
    // lam = mi*16 + m (= cout (m>>2)*16 + mi*4 + (m&3)), tap t is slot t & 7 of chunk t >> 3, channel g16 its element
    float wa[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int lam = mi * 16 + r16;
            wa[t][mi] = *(const float*)((const char*)a.wgt + (size_t)(t >> 3) * kWBlockBytes + lam * kRowBytes + (((t & 7) ^ swz(lam)) * kSlotBytes) + g16 * 4);
        }
    f32x4 acc[4][8];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int pt = 0; pt < 8; ++pt) acc[mi][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();                                                 // patch complete
#pragma unroll
    for (int pt = 0; pt < 8; ++pt) {
        const int pbase = ((wave * 4 + (pt >> 1)) * PW + (pt & 1) * 16 + r16) * 4 + g16;      // float index of tap (0,0)'s channel g16 for this lane's site
        float bv[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) bv[t] = patch[pbase + ((t / 3) * PW + t % 3) * 4];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                acc[mi][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t][mi], bv[t], acc[mi][pt], 0, 0, 0);
    }
    add_bias_after_k(a.bias + g16 * 16, acc);
    split_epilogue<1, F16>(a, acc, smem, n, ty0, tx0, wave, 0, 0, 0);
}

// conv1_1 of an operand-split handle at throughput size: fp32 planes in, a.out_parts (2 | 3) planes out; hipErrorInvalidConfiguration if the launch does not
// qualify (the caller keeps conv_igemm<float>)
hipError_t launch_conv1_1_split(const ConvArgs& a, hipStream_t s) {
    if (a.pk_L == nullptr || a.bn_scale != nullptr || a.resid != nullptr || a.img_shift != nullptr || a.ncg != 1 || a.nkc != 2 || a.so != 1 ||
        a.out_parts < 1 || a.out_parts > 3 || a.in2 != nullptr)
        return hipErrorInvalidConfiguration;
    const long long blocks = (long long)((a.Ws + 31) / 32) * ((a.Hs + 15) / 16) * a.N;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    const int lds = 4 * 4096 + 34 * 18 * 16;
    if (a.split_f16) hipLaunchKernelGGL(conv1_1_split_kernel<true>, dim3((unsigned)blocks), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(conv1_1_split_kernel<false>, dim3((unsigned)blocks), dim3(256), lds, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// conv1_2_split_kernel (round 6): model1.2 (64 -> 64, 3x3, ReLU, eval-BN; model.py:15-17) of the operand-split precisions.  The generic 64-cout tile
// (conv_igemm_v2ps<1,4,1>: 98 KiB of LDS per 4-wave workgroup) leaves ONE wave per SIMD -- every tap's barrier and LDS-DMA wait is exposed: MFMA-busy 0.25,
// 0.86 ms for fp16x3 at N = 32.  This is conv1_block_fused_t<4,3,true>'s conv1_2 phase (32 x 12 pixel tile, 4 waves x 3 rows, v_mfma_f32_32x32x16, the
// tap's 8 KiB weight tile re-laid for the 32x32 A operand on its way into a two-slot LDS ring, static halo tile: 76 KiB, TWO workgroups per CU) with the
// halo coming from the split tensor instead of an in-kernel conv1_1, walked a.nseg times: segment s stages input part seg_x[s] (buffer loads straight to
// LDS, bounds check = zero padding, the source address carries the swizzle; a segment that keeps the part keeps the halo) and streams weight part
// seg_w[s]; the accumulators start at zero, and the epilogue is fp32 throughout: x 2^-s + bias, ReLU, BN, hi = rne(v), next = rne(v - hi), ... every plane
// through the wave-private transpose tile, whole 128-byte lines.
template <bool F16>
__global__ __launch_bounds__(256, 2) void conv1_2_split_kernel(const ConvArgs a) {
    constexpr int NW = 4, RPW = 3, NT = NW * 64, TH = NW * RPW;
    constexpr int HW_ = 34, HH_ = TH + 2, NSITE = HW_ * HH_;
    constexpr int H_ITEMS = (NSITE * kSlots + NT - 1) / NT;            // 15 LDS-DMA instructions per thread and halo
    constexpr int HALO_BYTES = H_ITEMS * NT * kSlotBytes;              // 61,440
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const halo = smem;
    char* const wring = smem + HALO_BYTES;                             // 2 x 8 KiB
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 31, h = lane >> 5;
    const int Hs = a.Hs, Ws = a.Ws;
    const int ntx = (Ws + 31) >> 5, nty = (Hs + TH - 1) / TH;
    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int txi = b % ntx; b /= ntx;
    const int tyi = b % nty;
    const int n = b / nty;
    const int ty0 = tyi * TH, tx0 = txi * 32;
    const int np = a.in_parts, pix_bytes = np * kRowBytes;             // a pixel of the split input: [part][64 channels]
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in + (size_t)n * Hs * Ws * pix_bytes), 0, Hs * Ws * pix_bytes, 0x00020000);
    auto seg_xp = [&](int sg) { return (int)((a.seg_x >> (4 * sg)) & 15u); };
    auto seg_wp = [&](int sg) { return (size_t)((a.seg_w >> (4 * sg)) & 15u) * (size_t)a.w_part_bytes; };
    // weight tile -> ring slot, re-laid for the 32x32 A fragment (as conv1_block_fused_t: LDS row rho = the MFMA row, slot ^ swz2(rho))
    constexpr int W2_ITEMS = kWBlockBytes / (NT * kSlotBytes);
    int w2_src[W2_ITEMS];
#pragma unroll
    for (int j = 0; j < W2_ITEMS; ++j) {
        const int i = tid + j * NT, rho = i >> 3, sphys = i & 7, px_ = rho & 31, mi_ = rho >> 5;
        const int c = ((px_ >> 2) & 1) * 32 + mi_ * 16 + (px_ >> 3) * 4 + (px_ & 3);
        const int lr = ((c >> 2) & 3) * 16 + (c >> 4) * 4 + (c & 3);
        w2_src[j] = lr * kRowBytes + (((sphys ^ swz2(rho)) ^ swz(lr)) * kSlotBytes);
    }
    auto dma_w = [&](int sg, int t, int slot) {
        const char* const src = (const char*)a.wgt + seg_wp(sg) + (size_t)t * kWBlockBytes;
        char* dst = wring + slot * kWBlockBytes + wave * 64 * kSlotBytes;
#pragma unroll
        for (int j = 0; j < W2_ITEMS; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + w2_src[j]),
                                             (__attribute__((address_space(3))) void*)(dst + j * NT * kSlotBytes), 16, 0, 0);
    };
    auto load_halo = [&](int xp) {
#pragma unroll
        for (int j = 0; j < H_ITEMS; ++j) {
            const int item = tid + j * NT;
            const int row = item >> 3, phys = item & 7;                // halo site, physical slot
            const int hy = row / HW_, hx = row - hy * HW_;
            const int Y = ty0 - 1 + hy, X = tx0 - 1 + hx;
            const bool inside = row < NSITE && (unsigned)Y < (unsigned)Hs && (unsigned)X < (unsigned)Ws;
            const int off = (Y * Ws + X) * pix_bytes + xp * kRowBytes + ((phys ^ swz2(row)) * kSlotBytes);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(halo + (j * NT + wave * 64) * kSlotBytes), 16,
                                                     inside ? off : (int)0x80000000, 0, 0, 0);
        }
    };
    const int nseg = a.nseg, total = nseg * 9;
    dma_w(0, 0, 0);
    load_halo(seg_xp(0));

    f32x16 acc[2][RPW];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int pj = 0; pj < RPW; ++pj)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][pj][e] = 0.f;
    const int wlam0 = px * kRowBytes + ((h ^ swz2(px)) * kSlotBytes), wlam1 = wlam0 + 32 * kRowBytes;
    int sg = 0, t = 0;
#pragma unroll 1
    for (int s = 0; s < total; ++s) {
        const char* const wcur_ = wring + (s & 1) * kWBlockBytes;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // my pieces of this step's weight tile (and of a halo just asked for)
        __syncthreads();                                        // everybody's; everybody left the other ring slot
        int sg_n = sg, t_n = t + 1;
        if (t_n == 9) { t_n = 0; ++sg_n; }
        if (s + 1 < total) dma_w(sg_n, t_n, (s + 1) & 1);
        const int dy = t / 3 - 1, dx = t - (t / 3) * 3 - 1;
        int xaddr[RPW];
#pragma unroll
        for (int pj = 0; pj < RPW; ++pj) {
            const int xr = (wave * RPW + pj + 1 + dy) * HW_ + (px + 1 + dx);
            xaddr[pj] = xr * kRowBytes + ((h ^ swz2(xr)) * kSlotBytes);
        }
        u32x4 xfA[RPW], xfB[RPW], wA[2], wB[2];
        auto read_f = [&](int kk, u32x4 (&xf)[RPW], u32x4 (&wf)[2]) {
            wf[0] = *(const u32x4*)(wcur_ + (wlam0 ^ (kk * 2 * kSlotBytes)));
            wf[1] = *(const u32x4*)(wcur_ + (wlam1 ^ (kk * 2 * kSlotBytes)));
#pragma unroll
            for (int pj = 0; pj < RPW; ++pj) xf[pj] = *(const u32x4*)(halo + (xaddr[pj] ^ (kk * 2 * kSlotBytes)));
        };
        auto mmaL = [&](const u32x4 (&wf)[2], const u32x4 (&xf)[RPW]) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int pj = 0; pj < RPW; ++pj) {
                    if constexpr (F16) acc[mi][pj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_m, wf[mi]), __builtin_bit_cast(f16x8_m, xf[pj]), acc[mi][pj], 0, 0, 0);
                    else acc[mi][pj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[mi]), __builtin_bit_cast(bf16x8, xf[pj]), acc[mi][pj], 0, 0, 0);
                }
        };
#define IDC_C12_INTERLEAVE()                                                          \
    _Pragma("unroll") for (int q_ = 0; q_ < RPW + 2; ++q_) {                         \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                            \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                            \
    }                                                                                 \
    __builtin_amdgcn_sched_group_barrier(0x008, RPW - 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_f(0, xfA, wA);
        __builtin_amdgcn_sched_group_barrier(0x100, RPW + 2, 0);
        read_f(1, xfB, wB);
        mmaL(wA, xfA);
        IDC_C12_INTERLEAVE()
        read_f(2, xfA, wA);
        mmaL(wB, xfB);
        IDC_C12_INTERLEAVE()
        read_f(3, xfB, wB);
        mmaL(wA, xfA);
        IDC_C12_INTERLEAVE()
        mmaL(wB, xfB);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * RPW, 0);
#undef IDC_C12_INTERLEAVE
        if (t == 8 && s + 1 < total && seg_xp(sg_n) != seg_xp(sg)) {   // the next segment reads another input part (uniform)
            __syncthreads();                                    // everybody is done with this halo
            load_halo(seg_xp(sg_n));                            // lands before the next step's vmcnt(0) + barrier
        }
        sg = sg_n; t = t_n;
    }
    __syncthreads();                                            // every wave left the halo tile: the transpose tiles live there
    // ---- epilogue: v = BN(ReLU(acc * 2^-s + bias)) in fp32, then the planes -------------------------------------------------------------
    const float sc = a.acc_scale != nullptr ? *a.acc_scale : 1.f;
    const bool has_bn = a.bn_scale != nullptr;
    f32x16 bia[2], bsc[2], bsh[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bq = *(const float4*)(a.bias + h * 32 + mi * 16 + q * 4);
            bia[mi][q * 4 + 0] = bq.x; bia[mi][q * 4 + 1] = bq.y; bia[mi][q * 4 + 2] = bq.z; bia[mi][q * 4 + 3] = bq.w;
            float4 s4 = float4{1.f, 1.f, 1.f, 1.f}, t4 = float4{0.f, 0.f, 0.f, 0.f};
            if (has_bn) { s4 = *(const float4*)(a.bn_scale + h * 32 + mi * 16 + q * 4); t4 = *(const float4*)(a.bn_shift + h * 32 + mi * 16 + q * 4); }
            bsc[mi][q * 4 + 0] = s4.x; bsc[mi][q * 4 + 1] = s4.y; bsc[mi][q * 4 + 2] = s4.z; bsc[mi][q * 4 + 3] = s4.w;
            bsh[mi][q * 4 + 0] = t4.x; bsh[mi][q * 4 + 1] = t4.y; bsh[mi][q * 4 + 2] = t4.z; bsh[mi][q * 4 + 3] = t4.w;
        }
    char* const tb16 = smem + wave * 4096;
    const int rr = lane >> 3, cc = lane & 7;
    const int onp = a.out_parts, CoutPad = a.ncg * kCoutGroup;
    const bool relu = a.act == 1;
#pragma unroll
    for (int pj = 0; pj < RPW; ++pj) {
        const int sy = ty0 + wave * RPW + pj;
        float v[2][16];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float x = fmaf(acc[mi][pj][e], sc, bia[mi][e]);
                if (relu) x = fmaxf(x, 0.f);
                v[mi][e] = fmaf(x, bsc[mi][e], bsh[mi][e]);
            }
        for (int p = 0; p < onp; ++p) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                unsigned pk[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if constexpr (F16) {
                        const unsigned q = pack_f16x2_m(v[mi][2 * e], v[mi][2 * e + 1]);
                        pk[e] = q;
                        v[mi][2 * e] -= f16_lo_to_f32(q); v[mi][2 * e + 1] -= f16_hi_to_f32(q);
                    } else {
                        const unsigned q = pack_bf16x2(v[mi][2 * e], v[mi][2 * e + 1]);
                        pk[e] = q;
                        v[mi][2 * e] -= __uint_as_float(q << 16); v[mi][2 * e + 1] -= __uint_as_float(q & 0xffff0000u);
                    }
                }
                const int s0 = h * 4 + mi * 2;
                *(uint4*)(tb16 + px * 128 + ((s0 ^ (px & 7)) * 16)) = uint4{pk[0], pk[1], pk[2], pk[3]};
                *(uint4*)(tb16 + px * 128 + (((s0 + 1) ^ (px & 7)) * 16)) = uint4{pk[4], pk[5], pk[6], pk[7]};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            auto line = [&](int i) { const int row = i * 8 + rr; return *(const uint4*)(tb16 + row * 128 + ((cc ^ (row & 7)) * 16)); };
            const uint4 o0 = line(0), o1 = line(1), o2 = line(2), o3 = line(3);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            auto put = [&](int i, const uint4& o) {
                const int sx = tx0 + i * 8 + rr;
                if (sy < Hs && sx < Ws)
                    *(uint4*)((unsigned short*)a.out + ((((size_t)n * Hs + sy) * Ws + sx) * onp + p) * CoutPad + cc * 8) = o;
            };
            put(0, o0); put(1, o1); put(2, o2); put(3, o3);
        }
    }
}

static hipError_t init_kernels_conv1_split() {
    hipError_t e = hipFuncSetAttribute((const void*)conv1_2_split_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)conv1_2_split_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
}

constexpr int kConv12SplitLds = ((34 * 14 * kSlots + 255) / 256) * 256 * kSlotBytes + 2 * kWBlockBytes;      // 77,824

// conv1_2 of an operand-split handle (64 -> 64, 3x3 stride 1, ReLU, optional BN; split tensors in and out); hipErrorInvalidConfiguration otherwise
hipError_t launch_conv1_2_split(const ConvArgs& a, hipStream_t s) {
    if (a.ncg != 1 || a.nkc != 1 || a.nphase != 1 || a.ntaps != 9 || a.si != 1 || a.so != 1 || a.dy[8] != 1 || a.act != 1 || a.resid != nullptr ||
        a.img_shift != nullptr || a.in2 != nullptr || a.pk_L != nullptr || a.head_w != nullptr || a.out_f32 || a.in_parts < 1 || a.in_parts > 3 ||
        a.out_parts != a.in_parts || a.nseg < 1 || a.nseg > 6 || a.w_part_bytes == 0 || (long long)a.Hs * a.Ws * a.in_parts * kRowBytes >= 0x7fffffffLL)
        return hipErrorInvalidConfiguration;
    const long long blocks = (long long)((a.Ws + 31) / 32) * ((a.Hs + 11) / 12) * a.N;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    if (a.split_f16) hipLaunchKernelGGL(conv1_2_split_kernel<true>, dim3((unsigned)blocks), dim3(256), kConv12SplitLds, s, a);
    else hipLaunchKernelGGL(conv1_2_split_kernel<false>, dim3((unsigned)blocks), dim3(256), kConv12SplitLds, s, a);
    return hipGetLastError();
}


}  // namespace idc
