// idc_wino.hip -- Winograd F(2x2,3x3) form of the 3x3 stride-1 convolutions on the fp32 path (gfx950).
//
// Why only fp32 ("Winograd study"): a fused Winograd kernel keeps 16 position-accumulators per output tile, so a
// workgroup's per-position GEMM tile shrinks 4x and its operand traffic per MFMA grows 4x.  With bf16 MFMAs (32 cycles for
// 32768 flops) that is ~1 KB of operands per MFMA -- L2-bound; with the exact-fp32 v_mfma_f32_16x16x4_f32 (32 cycles for 2048
// flops) the same bytes are spread over 16x the matrix-pipe time and the 2.25x fewer multiplies are nearly pure gain.  The
// transforms are exact in fp32 up to rounding (CPU emulation, oracle/emulate.py: error vs float64 unchanged on torch-init
// weights, +20 % mean on full-range weights -- inside the reference's own fp32 noise).
//
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A            models/pytorch/model.py:13-102 (every 3x3, stride 1, pad = dilation)
//
// conv_wino_f32: one workgroup = 16 tiles of 2x2 outputs (an 8x8 block of one dilation sub-grid) x 32 output channels x
// all 16 transform positions, K = Cin in 128-byte (32-channel) chunks; 8 waves, wave w owns positions 2w, 2w+1:
//   * dilation d = 2 is four independent d = 1 problems on the parity sub-grids: a workgroup's pixels are Y0 + d*k;
//   * per chunk the (4+4+2)^2 = 100-pixel input patch goes global -> registers -> LDS (one chunk ahead), every thread
//     transforms one (tile, 16-byte channel slot, row i of B^T d B) from it: 8 ds_read_b128, 8 float4 add/sub, 4
//     ds_write_b128 into the position planes V[pos][tile][32 ch] (slot ^ (tile & 7): conflict-free both ways);
//   * the transformed weights U = G g G^T are packed on the host (float64 transform, idc_engine.hip) in the exact order the
//     MFMA A operand wants them: [chunk][pos][16 couts][ks][lane][4 floats] -- each wave streams ITS positions' fragments
//     global -> registers with fully coalesced 1 KiB loads, one chunk ahead; no weights in LDS at all;
//   * MFMA: v_mfma_f32_16x16x4_f32 on 16-byte fragments (4 MFMAs per fragment pair, the same K permutation on both operands),
//     per chunk 2 positions x 2 cout blocks x 8 = 32 MFMAs per wave, blocked accumulation per chunk like conv_igemm<float>;
//   * output transform: the 16 position sums of a (tile, cout) meet in LDS (32 KiB), one thread per (tile, cout) forms the
//     2x2 outputs, adds the bias, applies activation / eval-BN / per-image shift and stores 128-byte runs of couts.
// No split-K at batch 1: 256 tiles x 16 cout groups = 256 workgroups for a 512->512 layer at 32x32.
#include <stdlib.h>

#include <type_traits>

#include "idc_kernels.h"

#include "idc_layout.h"

#ifndef IDC_STAMP            // in-kernel cycle stamps exist only in the tuning harness (tools/ablate includes idc_kernels.hip first)
#define IDC_STAMP(i) do {} while (0)
#endif

namespace idc {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ int xcd_remap_w(int b, int nb) {
    const int xcd = b & 7, q = nb >> 3, r = nb & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (b >> 3);
}

__device__ __forceinline__ f32x4 as_f(const u32x4& v) { return __builtin_bit_cast(f32x4, v); }

constexpr int kWinoNT = 512;
constexpr int wino_v_bytes(int tb) { return 16 * 16 * tb * kRowBytes; }                  // one V buffer: [pos 16][tile 16*TB][128 B]
constexpr int wino_p_items(int tb) { return (10 * (8 * tb + 2) * 8 + kWinoNT - 1) / kWinoNT; }   // 16-byte patch pieces per thread
constexpr int wino_p_bytes(int tb) { return wino_p_items(tb) * kWinoNT * kSlotBytes; }
// two patch buffers; TB = 1: two V buffers (one barrier per chunk), TB = 2: ONE V buffer of 64 KiB (a chunk's B fragments are
// read into registers up front, then the next chunk's transform overwrites it under the chunk's MFMAs -- see the K loops)
constexpr int wino_lds(int tb) { return (tb == 1 ? 2 : 1) * wino_v_bytes(tb) + 2 * wino_p_bytes(tb); }

// TB = tile blocks of 16 per workgroup (tiles: 4 rows x 4*TB columns = an 8 x 8*TB pixel block of one sub-grid),
// CB = 16-cout blocks per workgroup.  A weight fragment (1 KiB per wave) feeds 4*TB MFMAs, a V fragment 4*CB: <1,2> is the
// small-grid form, <2,1> halves the weight stream for the same number of workgroups (batch 1: the kernel is bound by the
// L2 -> CU stream of U, 64 KiB per chunk and workgroup in the <1,2> form), <2,2> is the throughput form.
template <int TB, int CB>
__global__ __launch_bounds__(kWinoNT, 2) void conv_wino_f32(const ConvArgs a) {
    constexpr int NT = kWinoNT, TXL = 4 * TB, PW = 8 * TB + 2, NTILE = 16 * TB, VB = wino_v_bytes(TB), PI = wino_p_items(TB);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Vb = smem;
    char* const Pb = smem + (TB == 1 ? 2 : 1) * VB;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    IDC_STAMP(0);

    int b = xcd_remap_w(blockIdx.x, gridDim.x);
    // block order: tile blocks fastest, the cout group slowest -- each XCD (a contiguous range of the logical order) then
    // works on few cout groups whose U slices stay in that XCD's 4 MiB L2 for all tile blocks
    const int d = a.dy[8];                                     // dilation (tap (2,2) sits at +d)
    const int bx = b % a.tiles_x; b /= a.tiles_x;
    const int by = b % a.tiles_y; b /= a.tiles_y;
    const int par = b % (d * d); b /= d * d;
    const int n = b % a.N;
    const int cg = b / a.N;                                    // group of 16*CB couts
    const int Y0 = par / d + d * 8 * by, X0 = par % d + d * 8 * TB * bx;   // first output pixel; the block's pixels are Y0 + d*k
    const int H = a.Hs, W = a.Ws;
    const int nkc = a.nkc;
    const int pix_bytes = nkc * kRowBytes;
    const int si = a.si;                                       // 2: the layer reads x[::2, ::2] (model.py:149-151) -- a stride-1 conv on a strided view
    const char* const img = (const char*)a.in + (size_t)n * (H * si) * (W * si) * pix_bytes;

    // ---- patch staging plan (fixed over the K loop): item k = (patch pixel k>>3, slot k&7) ---------------------------
    int poff[PI];
#pragma unroll
    for (int j = 0; j < PI; ++j) {
        const int k = tid + j * NT;
        const int p = k >> 3, s = k & 7;
        const int py = p / PW, px = p - py * PW;
        const int Y = Y0 + d * (py - 1), X = X0 + d * (px - 1);
        const bool inside = k < 10 * PW * 8 && (unsigned)Y < (unsigned)H && (unsigned)X < (unsigned)W;
        poff[j] = inside ? ((Y * si) * (W * si) + X * si) * pix_bytes + s * kSlotBytes : -1;
    }
    u32x4 xr[PI];
    auto load_patch = [&](int c) {
#pragma unroll
        for (int j = 0; j < PI; ++j)
            xr[j] = *(const u32x4*)((poff[j] >= 0 && c < nkc) ? img + poff[j] + c * kRowBytes : (const char*)a.zeros);
    };
    constexpr int PB = wino_p_bytes(TB);
    auto store_patch = [&](int pbuf) {
#pragma unroll
        for (int j = 0; j < PI; ++j) *(u32x4*)(Pb + pbuf * PB + (tid + j * NT) * kSlotBytes) = xr[j];
    };

    // ---- input transform: item = (tile tt, row i of B^T d B, slot ts); TB items per thread --------------------------------
    const int ts = tid & 7, ti = (tid >> 3) & 3;
    const int rA = ti == 0 ? 0 : (ti == 2 ? 2 : 1), rB = ti == 3 ? 3 : (ti == 2 ? 1 : 2);
    const float sgn = ti == 1 ? 1.f : -1.f;                    // rows: d0-d2 | d1+d2 | d2-d1 | d1-d3
    auto transform = [&](int buf, int pbuf) {
#pragma unroll
        for (int q = 0; q < TB; ++q) {
            const int tt = (tid >> 5) + q * 16;
            const int pbase = pbuf * PB + ((2 * (tt / TXL)) * PW + 2 * (tt % TXL)) * kRowBytes + ts * kSlotBytes;
            f32x4 t[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 u = *(const f32x4*)(Pb + pbase + (rA * PW + c) * kRowBytes);
                const f32x4 v = *(const f32x4*)(Pb + pbase + (rB * PW + c) * kRowBytes);
                t[c] = u + sgn * v;
            }
            char* const dst = Vb + buf * VB + ((ti * 4) * NTILE + tt) * kRowBytes + ((ts ^ (tt & 7)) * kSlotBytes);
            *(f32x4*)(dst) = t[0] - t[2];
            *(f32x4*)(dst + NTILE * kRowBytes) = t[1] + t[2];
            *(f32x4*)(dst + 2 * NTILE * kRowBytes) = t[2] - t[1];
            *(f32x4*)(dst + 3 * NTILE * kRowBytes) = t[1] - t[3];
        }
    };

    // ---- weight fragments: this wave's positions p0, p0+1; [chunk][pos][cout block of 16][ks][lane][16 B] -------------
    const int p0 = wave * 2;
    const int ncb = a.ncg * 4;                                 // 16-cout blocks in the layer
    const char* const ubase = (const char*)a.wgt + ((size_t)(cg * CB) * 2 * 64 + lane) * kSlotBytes;
    const size_t u_pos_stride = (size_t)ncb * 2 * 64 * kSlotBytes;
    u32x4 areg[TB == 1 ? 3 : 2][2][CB][2];                     // [buffer][pos][cout block][ks]; TB = 1: fragments are requested TWO chunks ahead
    auto load_A = [&](int c, auto bufc) {
        constexpr int B = decltype(bufc)::value;
        const bool real = c < nkc;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const char* src = real ? ubase + ((size_t)c * 16 + p0 + pp) * u_pos_stride : (const char*)a.zeros;
            const int step = real ? 64 * kSlotBytes : 0;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    areg[B][pp][cb][ks] = *(const u32x4*)(src + (cb * 2 + ks) * step);
        }
    };

    const int fn = lane & 15, fg = lane >> 4;
    f32x4 tot[2][CB][TB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
            for (int k = 0; k < TB; ++k) tot[i][j][k] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one chunk's matrix work: B fragments from V[CUR], 2 x CB x TB x 8 MFMAs, blocked accumulation into tot
    [[maybe_unused]] auto mma_chunk = [&](auto curc) {
        constexpr int CUR = decltype(curc)::value;
        f32x4 acc[2][CB][TB];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < CB; ++j)
#pragma unroll
                for (int k = 0; k < TB; ++k) acc[i][j][k] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* const vcur = Vb + CUR * VB;
        f32x4 bf[2][TB][2];
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int tb = 0; tb < TB; ++tb)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    bf[pp][tb][ks] = *(const f32x4*)(vcur + ((p0 + pp) * NTILE + tb * 16 + fn) * kRowBytes + (((ks * 4 + fg) ^ (fn & 7)) * kSlotBytes));
#ifndef IDC_WINO_ABL_NO_MFMA
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                        for (int tb = 0; tb < TB; ++tb)
                            acc[pp][cb][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(as_f(areg[CUR][pp][cb][ks])[e], bf[pp][tb][ks][e], acc[pp][cb][tb], 0, 0, 0);
#else       // timing ablation (tools/ablate): keep the operands live, drop the matrix work
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int tb = 0; tb < TB; ++tb)
                    acc[pp][cb][tb] += as_f(areg[CUR][pp][cb][0]) * bf[pp][tb][0] + as_f(areg[CUR][pp][cb][1]) * bf[pp][tb][1];
#endif
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < CB; ++j)
#pragma unroll
                for (int k = 0; k < TB; ++k) tot[i][j][k] += acc[i][j][k];
    };

    if constexpr (TB == 1) {
        // ---- one barrier per chunk: V and the patch are both double-buffered.  In chunk c (after the barrier that publishes
        // V[c&1] and patch c+1 in P[(c+1)&1]) a wave stores the patch of chunk c+2 (registers, fetched a chunk ago) into
        // P[c&1], requests patch c+3 and the fragments of chunk c+1, transforms patch c+1 into V[(c+1)&1] and runs chunk c's
        // MFMAs -- nothing in the chunk waits for another wave, so the compiler interleaves the transform's LDS traffic and
        // the loads with the matrix work.
        load_patch(0);
        load_A(0, std::integral_constant<int, 0>{});
        load_A(nkc > 1 ? 1 : 0, std::integral_constant<int, 1>{});
        store_patch(0);
        load_patch(1);
        __syncthreads();
        transform(0, 0);
        store_patch(1);
        load_patch(2);
        __syncthreads();
        IDC_STAMP(1);
        // Instruction order inside a chunk is pinned (sched_barrier between groups): a wave that issues its ten 1 KiB loads
        // back to back sits in the memory-instruction queue for ~1 k cycles before it reaches its first MFMA (measured: the
        // matrix work then ADDS to the load time instead of hiding it), so one load follows every group of four MFMAs.
        // The weight fragments are requested TWO chunks ahead (three register sets): with one chunk (64 KiB per CU) in flight the
        // L2 -> CU stream is latency-bound at ~29 B/clk, i.e. a chunk's fragments arrive a chunk and a bit after their request.
        auto chunk1 = [&](int c, auto aidx) {
            constexpr int AI = decltype(aidx)::value, AN = (AI + 2) % 3;
            const int CUR = c & 1;
            static_assert(CB == 2, "group plan below assumes 8 A fragments per chunk");
            const int cn = c + 2 < nkc ? c + 2 : nkc - 1;       // past the end: a harmless re-read of the last chunk (no branch, no select)
            const char* const asrc = ubase + ((size_t)cn * 16 + p0) * u_pos_stride;
            const char* const vcur = Vb + CUR * VB;
            f32x4 bf[2][2];
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    bf[pp][ks] = *(const f32x4*)(vcur + ((p0 + pp) * NTILE + fn) * kRowBytes + (((ks * 4 + fg) ^ (fn & 7)) * kSlotBytes));
            store_patch(CUR);                                  // patch c+2 -> P[c&1] (its reader, the transform of chunk c-1, is behind the barrier)
            f32x4 acc[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto mma4 = [&](int ks, int e) {
#ifndef IDC_WINO_ABL_NO_MFMA
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
                        acc[pp][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(as_f(areg[AI][pp][cb][ks])[e], bf[pp][ks][e], acc[pp][cb], 0, 0, 0);
#else
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) acc[pp][cb][e] += as_f(areg[AI][pp][cb][ks])[e] * bf[pp][ks][e];
#endif
            };
            auto loadA1 = [&](int pp, int cb, int ks) {
                areg[AN][pp][cb][ks] = *(const u32x4*)(asrc + (size_t)pp * u_pos_stride + (cb * 2 + ks) * 64 * kSlotBytes);
            };
            // transform of patch c+1 (P[(c+1)&1]) -> V[(c+1)&1], cut in three pieces
            const int tt = tid >> 5;
            const int pbase = (CUR ^ 1) * PB + ((2 * (tt / TXL)) * PW + 2 * (tt % TXL)) * kRowBytes + ts * kSlotBytes;
            f32x4 tu[4], tv[4];
            const bool do_tr = c + 1 < nkc;
            mma4(0, 0);
            __builtin_amdgcn_sched_barrier(0);
            loadA1(0, 0, 0); loadA1(0, 0, 1);
            mma4(0, 1);
            __builtin_amdgcn_sched_barrier(0);
            loadA1(0, 1, 0); loadA1(0, 1, 1);
            mma4(0, 2);
            __builtin_amdgcn_sched_barrier(0);
            loadA1(1, 0, 0); loadA1(1, 0, 1);
            mma4(0, 3);
            __builtin_amdgcn_sched_barrier(0);
            loadA1(1, 1, 0); loadA1(1, 1, 1);
            mma4(1, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_patch(c + 3);
#ifndef IDC_WINO_ABL_NO_TRANSFORM
#pragma unroll
            for (int q = 0; q < 4; ++q) tu[q] = *(const f32x4*)(Pb + pbase + (rA * PW + q) * kRowBytes);
#endif
            mma4(1, 1);
            __builtin_amdgcn_sched_barrier(0);
#ifndef IDC_WINO_ABL_NO_TRANSFORM
#pragma unroll
            for (int q = 0; q < 4; ++q) tv[q] = *(const f32x4*)(Pb + pbase + (rB * PW + q) * kRowBytes);
#endif
            mma4(1, 2);
            __builtin_amdgcn_sched_barrier(0);
#ifndef IDC_WINO_ABL_NO_TRANSFORM
            if (do_tr) {
                f32x4 t[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) t[q] = tu[q] + sgn * tv[q];
                char* const dst = Vb + (CUR ^ 1) * VB + ((ti * 4) * NTILE + tt) * kRowBytes + ((ts ^ (tt & 7)) * kSlotBytes);
                *(f32x4*)(dst) = t[0] - t[2];
                *(f32x4*)(dst + NTILE * kRowBytes) = t[1] + t[2];
                *(f32x4*)(dst + 2 * NTILE * kRowBytes) = t[2] - t[1];
                *(f32x4*)(dst + 3 * NTILE * kRowBytes) = t[1] - t[3];
            }
#endif
            mma4(1, 3);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) tot[i][j][0] += acc[i][j];
            __syncthreads();
        };
        int c = 0;
        for (; c + 2 < nkc; c += 3) {
            chunk1(c, std::integral_constant<int, 0>{});
            chunk1(c + 1, std::integral_constant<int, 1>{});
            chunk1(c + 2, std::integral_constant<int, 2>{});
        }
        if (c < nkc) chunk1(c, std::integral_constant<int, 0>{});
        if (c + 1 < nkc) chunk1(c + 1, std::integral_constant<int, 1>{});
    } else {
        // ---- TB = 2: ONE V buffer.  Chunk c: barrier A (V(c) and patch c+1 complete) -> every wave reads its 8 B fragments into
        // registers -> barrier B (V is free) -> patch c+2 to P[c&1], fragment loads of chunk c+1, patch c+3, and the transform of
        // patch c+1 INTO THE SAME V, all interleaved with the chunk's 32*CB MFMAs.  The transform writes 64 KiB of LDS per chunk
        // (ds_write_b128 runs at ~80 B/clk per CU: ~800 cycles): between two barriers it was 35 % of the kernel, here it hides
        // under the matrix work.
        load_patch(0);
        load_A(0, std::integral_constant<int, 0>{});
        store_patch(0);
        load_patch(1);
        __syncthreads();
        transform(0, 0);
        store_patch(1);
        load_patch(2);
        IDC_STAMP(1);
        auto chunk2 = [&](int c, auto curc) {
            constexpr int CUR = decltype(curc)::value;
            const int cn = c + 1 < nkc ? c + 1 : nkc - 1;
            const char* const asrc = ubase + ((size_t)cn * 16 + p0) * u_pos_stride;
            __syncthreads();                                       // A: V(c) and the patch of chunk c+1 are complete
            f32x4 bf[2][TB][2];
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int tb = 0; tb < TB; ++tb)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        bf[pp][tb][ks] = *(const f32x4*)(Vb + ((p0 + pp) * NTILE + tb * 16 + fn) * kRowBytes + (((ks * 4 + fg) ^ (fn & 7)) * kSlotBytes));
            __syncthreads();                                       // B: every wave holds its fragments, V may be overwritten
            store_patch(c & 1);                                    // patch c+2 -> P[c&1] (read by the transform of chunk c-1, long done)
            f32x4 acc[2][CB][TB];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < CB; ++j)
#pragma unroll
                    for (int k = 0; k < TB; ++k) acc[i][j][k] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto mmag = [&](int ks, int e) {
#ifndef IDC_WINO_ABL_NO_MFMA
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                        for (int tb = 0; tb < TB; ++tb)
                            acc[pp][cb][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(as_f(areg[CUR][pp][cb][ks])[e], bf[pp][tb][ks][e], acc[pp][cb][tb], 0, 0, 0);
#else
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                        for (int tb = 0; tb < TB; ++tb) acc[pp][cb][tb][e] += as_f(areg[CUR][pp][cb][ks])[e] * bf[pp][tb][ks][e];
#endif
            };
            auto loadA1 = [&](int pp, int cb, int ks) {
                areg[CUR ^ 1][pp][cb][ks] = *(const u32x4*)(asrc + (size_t)pp * u_pos_stride + (cb * 2 + ks) * 64 * kSlotBytes);
            };
            const int pbuf = (c + 1) & 1;
            const bool do_tr = c + 1 < nkc;
            f32x4 tu[4], tv[4];
            auto tr_read = [&](int q) {
                const int tt = (tid >> 5) + q * 16;
                const int pbase = pbuf * PB + ((2 * (tt / TXL)) * PW + 2 * (tt % TXL)) * kRowBytes + ts * kSlotBytes;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    tu[k] = *(const f32x4*)(Pb + pbase + (rA * PW + k) * kRowBytes);
                    tv[k] = *(const f32x4*)(Pb + pbase + (rB * PW + k) * kRowBytes);
                }
            };
            auto tr_write = [&](int q) {
                const int tt = (tid >> 5) + q * 16;
                f32x4 t[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) t[k] = tu[k] + sgn * tv[k];
                char* const dst = Vb + ((ti * 4) * NTILE + tt) * kRowBytes + ((ts ^ (tt & 7)) * kSlotBytes);
                if (do_tr) {
                    *(f32x4*)(dst) = t[0] - t[2];
                    *(f32x4*)(dst + NTILE * kRowBytes) = t[1] + t[2];
                    *(f32x4*)(dst + 2 * NTILE * kRowBytes) = t[2] - t[1];
                    *(f32x4*)(dst + 3 * NTILE * kRowBytes) = t[1] - t[3];
                }
            };
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                mmag(g >> 2, g & 3);
                __builtin_amdgcn_sched_barrier(0);
                if (g < 4) {                                       // the fragments (pos g>>1, ks g&1) of every cout block
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) loadA1(g >> 1, cb, g & 1);
                }
#ifndef IDC_WINO_ABL_NO_TRANSFORM
                if (g == 1) tr_read(0);
                if (g == 3) tr_write(0);
                if (g == 4) tr_read(1);
                if (g == 6) tr_write(1);
#endif
                if (g == 5) load_patch(c + 3);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < CB; ++j)
#pragma unroll
                    for (int k = 0; k < TB; ++k) tot[i][j][k] += acc[i][j][k];
        };
        int c = 0;
        for (; c + 1 < nkc; c += 2) {
            chunk2(c, std::integral_constant<int, 0>{});
            chunk2(c + 1, std::integral_constant<int, 1>{});
        }
        if (c < nkc) chunk2(c, std::integral_constant<int, 0>{});
        __syncthreads();                                           // the epilogue reuses V: every wave's last fragment reads are done
    }

    IDC_STAMP(2);
    // ---- output transform: the 16 position sums of every (tile, cout) meet in LDS -----------------------------------
    // (the last barrier of the K loop is behind us: nobody reads V any more)
    char* const Mx = smem;                                     // [pos 16][tile 16*TB][128-B row of couts], 16-B slot ^ (tile & 7)
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int tb = 0; tb < TB; ++tb)
                *(f32x4*)(Mx + ((p0 + pp) * NTILE + tb * 16 + fn) * kRowBytes + (((cb * 4 + fg) ^ (fn & 7)) * kSlotBytes)) = tot[pp][cb][tb];
    __syncthreads();
    const int CoutPad = a.ncg * kCoutGroup;
    const bool has_bn = a.bn_scale != nullptr;
    constexpr int NC = 16 * CB;
#pragma unroll
    for (int q = 0; q < (NTILE * NC) / NT; ++q) {
        const int idx = tid + q * NT;
        const int oc = idx % NC, ot = idx / NC;
        float m[16];
#pragma unroll
        for (int p = 0; p < 16; ++p)
            m[p] = *(const float*)(Mx + (p * NTILE + ot) * kRowBytes + (((oc >> 2) ^ (ot & 7)) * kSlotBytes) + (oc & 3) * 4);
        float s0[4], s1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s0[j] = m[0 * 4 + j] + m[1 * 4 + j] + m[2 * 4 + j];
            s1[j] = m[1 * 4 + j] - m[2 * 4 + j] - m[3 * 4 + j];
        }
        const float y[2][2] = {{s0[0] + s0[1] + s0[2], s0[1] - s0[2] - s0[3]}, {s1[0] + s1[1] + s1[2], s1[1] - s1[2] - s1[3]}};
        const int co = cg * NC + oc;
        const float bias = a.bias[co];
        const float bsc = has_bn ? a.bn_scale[co] : 1.f, bsh = has_bn ? a.bn_shift[co] : 0.f;
        const float ish = a.img_shift ? a.img_shift[(size_t)n * CoutPad + co] : 0.f;
        float* const out = (float*)a.out + (size_t)n * H * W * CoutPad + co;
        const int oy = Y0 + d * 2 * (ot / TXL), ox = X0 + d * 2 * (ot % TXL);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int yy = oy + d * i, xx = ox + d * j;
                float v = y[i][j] + bias;
                if (a.act == 1) v = fmaxf(v, 0.f);
                else if (a.act == 2) v = v > 0.f ? v : 0.2f * v;
                if (has_bn) v = fmaf(v, bsc, bsh);
                v += ish;
                if (yy < H && xx < W) out[((size_t)yy * W + xx) * CoutPad] = v;
            }
    }
    IDC_STAMP(3);
#ifdef IDC_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    IDC_STAMP(4);
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// conv_wino_deconv_f32<CB>: ConvTranspose2d 4x4 stride 2 pad 1 (model8up / 9up / 10up, model.py:75,87,97) on the fp32 path as
// Winograd F(2x2,2x2).  Each of the four output phases (r,s) of the deconv is a 2x2-tap correlation over the input grid
// (out[2m+r,2n+s] uses taps (ky,dy) in T(r) = {(3,-1),(1,0)} | {(2,0),(0,+1)}); for a tile of 2x2 sites:
//     m1 = (d0-d1) ga, m2 = d1 (ga+gb), m3 = (d2-d1) gb,  y0 = m1+m2, y1 = m2+m3     (3 multiplies per 2 outputs instead of 4)
// in 2-D 9 multiplies per phase and tile instead of 16, 36 "positions" (phase r,s x i,j) per tile over ONE 4x4 site patch (the
// phase (r,s) reads its 3x3 sub-patch at offset (r,s)).  Same machinery as conv_wino_f32: workgroup = 16 tiles (an 8x8 block of
// sites = 16x16 output pixels) x 16*CB couts x 36 positions, 12 waves, wave w = (r, s, i) owns positions 3w .. 3w+2; U image
// [chunk][pos 36][16 couts][ks][lane][16 B] streamed global -> registers; patch -> LDS -> transform (one item per thread:
// tile, 16-byte slot, (r,i)) -> V[pos][tile][32 ch]; ONE V buffer of 72 KiB (fragments in registers, next transform under the
// MFMAs), two patch buffers; the 36 sums of a (tile, cout) meet in LDS, one thread per (tile, phase, cout) forms the 2x2 outputs
// of its phase, adds bias and the fp32 shortcut sum (model.py:156,170,172), applies the activation and stores.
// No split-K at batch 1 (the deconvs were the last layers of the fp32 click path that had reduction launches).
constexpr int kWinoDNT = 768;
constexpr int kWinoDVBytes = 36 * 16 * kRowBytes;           // 73 728
constexpr int kWinoDPBytes = 2 * 1024 * kSlotBytes;          // per patch buffer: 100 pixels x 8 slots <= 2 items x 768 threads (first 800 used)
constexpr int kWinoDLds = kWinoDVBytes + 2 * kWinoDPBytes;

template <int CB>
__global__ __launch_bounds__(kWinoDNT, 3) void conv_wino_deconv_f32(const ConvArgs a) {
    constexpr int NT = kWinoDNT, PW = 10, NTILE = 16, PBY = kWinoDPBytes;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Vb = smem;
    char* const Pb = smem + kWinoDVBytes;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    int b = xcd_remap_w(blockIdx.x, gridDim.x);
    const int bx = b % a.tiles_x; b /= a.tiles_x;
    const int by = b % a.tiles_y; b /= a.tiles_y;
    const int n = b % a.N;
    const int cg = b / a.N;                                    // group of 16*CB couts (slowest: its U slice stays in the XCD's L2)
    const int Y0 = 8 * by, X0 = 8 * bx;                        // first SITE (input pixel) of the block
    const int H = a.Hs, W = a.Ws;                              // input (site) resolution; the output is 2H x 2W
    const int nkc = a.nkc;
    const int pix_bytes = nkc * kRowBytes;
    const char* const img = (const char*)a.in + (size_t)n * H * W * pix_bytes;

    int poff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = tid + j * NT;
        const int p = k >> 3, s = k & 7;
        const int py = p / PW, px = p - py * PW;
        const int Y = Y0 - 1 + py, X = X0 - 1 + px;
        const bool inside = k < PW * PW * 8 && (unsigned)Y < (unsigned)H && (unsigned)X < (unsigned)W;
        poff[j] = inside ? (Y * W + X) * pix_bytes + s * kSlotBytes : -1;
    }
    u32x4 xr[2];
    auto load_patch = [&](int c) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            xr[j] = *(const u32x4*)((poff[j] >= 0 && c < nkc) ? img + poff[j] + c * kRowBytes : (const char*)a.zeros);
    };
    auto store_patch = [&](int pbuf) {
        *(u32x4*)(Pb + pbuf * PBY + tid * kSlotBytes) = xr[0];
        if (tid < PW * PW * 8 - NT) *(u32x4*)(Pb + pbuf * PBY + (tid + NT) * kSlotBytes) = xr[1];
    };

    // transform item = (tile tt, slot ts, (r, i)): rows of B^T d for the phase-row r -- i = 0: P[r] - P[r+1], 1: P[r+1], 2: P[r+2] - P[r+1]
    const int ts = tid & 7, tq = tid >> 3;                     // tq 0..95
    const int tt = tq / 6, tri = tq - tt * 6, tr = tri / 3, ti = tri - tr * 3;
    const int rowA = tr + (ti == 2 ? 2 : (ti == 1 ? 1 : 0)), rowB = tr + 1;
    const float wB = ti == 1 ? 0.f : -1.f;                    // t = P[rowA] + wB * P[rowB]
    const int pbase = ((2 * (tt >> 2)) * PW + 2 * (tt & 3)) * kRowBytes + ts * kSlotBytes;
    auto transform = [&](int pbuf, bool on) {
        f32x4 t[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 u = *(const f32x4*)(Pb + pbuf * PBY + pbase + (rowA * PW + c) * kRowBytes);
            const f32x4 v = *(const f32x4*)(Pb + pbuf * PBY + pbase + (rowB * PW + c) * kRowBytes);
            t[c] = u + wB * v;
        }
        if (on) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {                  // pos = ((r*2 + s)*3 + i)*3 + j
                char* const dst = Vb + ((((tr * 2 + s2) * 3 + ti) * 3) * NTILE + tt) * kRowBytes + ((ts ^ (tt & 7)) * kSlotBytes);
                *(f32x4*)(dst) = t[s2] - t[s2 + 1];
                *(f32x4*)(dst + NTILE * kRowBytes) = t[s2 + 1];
                *(f32x4*)(dst + 2 * NTILE * kRowBytes) = t[s2 + 2] - t[s2 + 1];
            }
        }
    };

    const int p0 = wave * 3;                                   // this wave's positions (r, s, i) x j = 0..2
    const int ncb = a.ncg * 4;
    const char* const ubase = (const char*)a.wgt + ((size_t)(cg * CB) * 2 * 64 + lane) * kSlotBytes;
    const size_t u_pos_stride = (size_t)ncb * 2 * 64 * kSlotBytes;
    u32x4 areg[2][3][CB][2];
    auto load_A = [&](int c, auto bufc) {
        constexpr int B = decltype(bufc)::value;
        const int cn = c < nkc ? c : nkc - 1;
        const char* const src0 = ubase + ((size_t)cn * 36 + p0) * u_pos_stride;
#pragma unroll
        for (int pp = 0; pp < 3; ++pp)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    areg[B][pp][cb][ks] = *(const u32x4*)(src0 + (size_t)pp * u_pos_stride + (cb * 2 + ks) * 64 * kSlotBytes);
    };
    const int fn = lane & 15, fg = lane >> 4;
    f32x4 tot[3][CB];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j) tot[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_patch(0);
    load_A(0, std::integral_constant<int, 0>{});
    store_patch(0);
    load_patch(1);
    __syncthreads();
    transform(0, true);
    store_patch(1);
    load_patch(2);
    auto chunk = [&](int c, auto curc) {
        constexpr int CUR = decltype(curc)::value;
        __syncthreads();                                       // A: V(c) and the patch of chunk c+1 are complete
        f32x4 bf[3][2];
#pragma unroll
        for (int pp = 0; pp < 3; ++pp)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                bf[pp][ks] = *(const f32x4*)(Vb + ((p0 + pp) * NTILE + fn) * kRowBytes + (((ks * 4 + fg) ^ (fn & 7)) * kSlotBytes));
        __syncthreads();                                       // B: every wave holds its fragments, V may be overwritten
        load_A(c + 1, std::integral_constant<int, CUR ^ 1>{});
        store_patch(c & 1);
        load_patch(c + 3);
        transform((c + 1) & 1, c + 1 < nkc);
        f32x4 acc[3][CB];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < CB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int pp = 0; pp < 3; ++pp)
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
                        acc[pp][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(as_f(areg[CUR][pp][cb][ks])[e], bf[pp][ks][e], acc[pp][cb], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < CB; ++j) tot[i][j] += acc[i][j];
    };
    int c = 0;
    for (; c + 1 < nkc; c += 2) {
        chunk(c, std::integral_constant<int, 0>{});
        chunk(c + 1, std::integral_constant<int, 1>{});
    }
    if (c < nkc) chunk(c, std::integral_constant<int, 0>{});
    __syncthreads();

    char* const Mx = smem;                                     // [pos 36][tile 16][128-B row of couts]
#pragma unroll
    for (int pp = 0; pp < 3; ++pp)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
            *(f32x4*)(Mx + ((p0 + pp) * NTILE + fn) * kRowBytes + (((cb * 4 + fg) ^ (fn & 7)) * kSlotBytes)) = tot[pp][cb];
    __syncthreads();
    const int CoutPad = a.ncg * kCoutGroup;
    const bool has_bn = a.bn_scale != nullptr;
    constexpr int NC = 16 * CB, NITEM = NTILE * 4 * NC;
    const int Ho = 2 * H, Wo = 2 * W;
    for (int idx = tid; idx < NITEM; idx += NT) {
        const int oc = idx % NC, rem = idx / NC, ph = rem & 3, ot = rem >> 2;
        float m[9];
#pragma unroll
        for (int p = 0; p < 9; ++p)
            m[p] = *(const float*)(Mx + ((ph * 9 + p) * NTILE + ot) * kRowBytes + (((oc >> 2) ^ (ot & 7)) * kSlotBytes) + (oc & 3) * 4);
        float sa[2][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { sa[0][j] = m[0 * 3 + j] + m[1 * 3 + j]; sa[1][j] = m[1 * 3 + j] + m[2 * 3 + j]; }
        const int co = cg * NC + oc;
        const float bias = a.bias[co];
        const float bsc = has_bn ? a.bn_scale[co] : 1.f, bsh = has_bn ? a.bn_shift[co] : 0.f;
        const int r = ph >> 1, s2 = ph & 1;
        const int sy = Y0 + 2 * (ot >> 2), sx = X0 + 2 * (ot & 3);
#pragma unroll
        for (int ya = 0; ya < 2; ++ya)
#pragma unroll
            for (int xb = 0; xb < 2; ++xb) {
                const int my = sy + ya, mx = sx + xb;
                if (my < H && mx < W) {
                    const size_t o = (((size_t)n * Ho + (2 * my + r)) * Wo + (2 * mx + s2)) * CoutPad + co;
                    float v = sa[ya][xb] + sa[ya][xb + 1] + bias;
                    if (a.resid != nullptr) v += ((const float*)a.resid)[o];
                    if (a.act == 1) v = fmaxf(v, 0.f);
                    else if (a.act == 2) v = v > 0.f ? v : 0.2f * v;
                    if (has_bn) v = fmaf(v, bsc, bsh);
                    ((float*)a.out)[o] = v;
                }
            }
    }
}
// ConvTranspose 4x4 s2 p1, fp32, Winograd F(2x2,2x2).  a.wgt = the layer's 36-position U image, a.Hs / a.Ws = INPUT size,
// a.resid = optional fp32 shortcut sum at the output resolution.
// The kernels of this file address a source image with 32-bit byte offsets (poff[] = ((Y*si)*(W*si) + X*si) * pix_bytes + ..., -1 =
// outside): the strided source of ONE image must stay below 2 GiB (fp32 conv2_1 reads conv1_2 at 256 B per pixel through its
// stride-2 view: 2896 x 2896).  Larger geometries take the direct kernels, whose offsets are size_t.
bool wino_offsets_fit(int Hs, int Ws, int si, int nkc) {
    return (long long)Hs * si * (long long)Ws * si * ((long long)nkc * kRowBytes) < 0x7fffffffLL;
}

// ONE predicate for "this launch can run as Winograd" -- the engine's variant choice (set_geometry), the single-operator entry points and
// the two launchers below all ask it (the launch guards used to be wider than the eligibility tests).
bool conv_wino_applies(int precision, const ConvArgs& a, bool deconv) {
    if (precision != 0 || a.zeros == nullptr || a.nkc < 1 || a.wgt == nullptr) return false;       // the Winograd forms are the fp32 path's
    if (deconv)
        return a.nphase == 4 && a.so == 2 && a.si == 1 && a.img_shift == nullptr && wino_offsets_fit(a.Hs, a.Ws, 1, a.nkc) &&
               !(precision == 0 && (!a.out_f32 || (a.resid != nullptr && a.resid_bf16)));
    const int d = a.dy[8];
    return (d == 1 || d == 2) && (a.si == 1 || a.si == 2) && a.so == 1 && a.nphase == 1 && a.ntaps == 9 && a.resid == nullptr &&
           !(precision == 0 && !a.out_f32) && wino_offsets_fit(a.Hs, a.Ws, a.si, a.nkc);
}

hipError_t launch_deconv_wino(int precision, const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    if (!conv_wino_applies(precision, a, true)) return hipErrorInvalidConfiguration;
    a.tiles_x = (a.Ws + 7) / 8;
    a.tiles_y = (a.Hs + 7) / 8;
    const long long tb = (long long)a.tiles_x * a.tiles_y * a.N;
    const int cb = 1;      // (a <2> form -- 32 couts per workgroup -- spills 57-96 registers at the 168-register budget of 12 waves: not instantiated)
    (void)tb;
    const long long blocks = tb * (a.ncg * 4 / cb);
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    if (precision != 0) return hipErrorInvalidConfiguration;      // fp32 only (the bf16 twin was retired in round 5: conv_kwave_deconv_bf16)
    hipLaunchKernelGGL((conv_wino_deconv_f32<1>), dim3((unsigned)blocks), dim3(kWinoDNT), kWinoDLds, s, a);
    return hipGetLastError();
}

int g_wino_form = idc_env_int("IDC_WINO_FORM", 0);      // tuning: 0 automatic, 12 / 21 / 22 = force <TB,CB>

void set_wino_form(int form) { g_wino_form = form; }

template <int TB, int CB>
static hipError_t launch_wino_t(ConvArgs& a, int d, int precision, hipStream_t s) {
    a.tiles_x = ((a.Ws + d - 1) / d + 8 * TB - 1) / (8 * TB);
    a.tiles_y = ((a.Hs + d - 1) / d + 7) / 8;
    const long long blocks = (long long)a.tiles_x * a.tiles_y * d * d * a.N * (a.ncg * 4 / CB);
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    if (precision != 0) return hipErrorInvalidConfiguration;      // fp32 only (the bf16 twin was retired in round 5: conv_kwave_bf16)
    hipLaunchKernelGGL((conv_wino_f32<TB, CB>), dim3((unsigned)blocks), dim3(kWinoNT), wino_lds(TB), s, a);
    return hipGetLastError();
}

// 3x3 stride-1 conv (dilation 1 or 2), fp32, Winograd F(2x2,3x3).  a.wgt = the layer's U image (idc_engine.hip packs it),
// a.nkc = Cin / 32, a.ncg = CoutPad / 64, a.Hs / a.Ws = image size, a.dy[8] = dilation; tiles_x / tiles_y are set here.
// Form by grid size (speed only: every form computes the same sums in the same order): the throughput form <2,2> when it
// still gives every CU a workgroup, else <1,2>.
hipError_t launch_conv_wino(int precision, const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    const int d = a.dy[8];
    if (!conv_wino_applies(precision, a, false)) return hipErrorInvalidConfiguration;
    const long long t2 = (long long)(((a.Ws + d - 1) / d + 15) / 16) * (((a.Hs + d - 1) / d + 7) / 8) * d * d * a.N;   // 8x16-pixel blocks
    int form = g_wino_form;
    // (measured, profiles/r03_wino_harness.txt: <2,1> loses to <1,2> on every shape -- its transform work per workgroup doubles --
    //  and is kept for the tests and the tuning switch only)
    if (form != 12 && form != 21 && form != 22) form = t2 * (a.ncg * 2) >= 256 ? 22 : 12;
    if (form == 22) return launch_wino_t<2, 2>(a, d, precision, s);
    if (form == 21) return launch_wino_t<2, 1>(a, d, precision, s);
    return launch_wino_t<1, 2>(a, d, precision, s);
}

hipError_t init_kernels_wino() {
    hipError_t e = hipFuncSetAttribute((const void*)conv_wino_f32<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, wino_lds(1));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)conv_wino_f32<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, wino_lds(2));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)conv_wino_f32<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, wino_lds(2));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)conv_wino_deconv_f32<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kWinoDLds);
}

}  // namespace idc
