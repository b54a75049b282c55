// does a bigger REGISTER tile pay for the trunk kernel?  conv_igemm_v2p runs 8 waves (2 per SIMD) with a wave tile of
// 64 couts x 128 sites = 128 accumulator registers, 128 of the 256 registers a wave may hold at 2 waves/SIMD.  This probe runs the kernel's
// K-loop MIX -- per 64-channel step: one workgroup barrier, 4 x 1 KiB LDS-DMA per wave (the 32 KiB weight tile), two k32 sub-steps of
// MA A-fragment + NB B-fragment ds_read_b128 from static conflict-free LDS images and MA x NB v_mfma_f32_16x16x32_bf16 -- for several wave
// tiles, same code for all of them (B fragments in groups of GB, a group's MFMAs over the next group's reads, A blocks reloaded behind their
// last use, hipcc's own interleave), one workgroup per CU:
//     <4, 8, 4>   64 couts x 128 sites   128 accumulators   12 reads / 32 MFMA   (the shipped tile)
//     <4,12, 4>   64 couts x 192 sites   192 accumulators   16 reads / 48 MFMA   LDS fragment bytes per MAC x 0.89
//     <8, 6, 3>  128 couts x  96 sites   192 accumulators   14 reads / 48 MFMA   x 0.78
//     <6, 8, 4>   96 couts x 128 sites   192 accumulators   14 reads / 48 MFMA   x 0.78  (does not divide 512 couts: for the number only)
// on zero operands (cycles: the chip holds its clock) and on uniform random bf16 operands (the power cap: joules).
//   tile_probe [steps] [seed]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int kLds = 152 * 1024;

template <int MA, int NB, int GB>
__global__ __launch_bounds__(512, 2) void tile(const char* __restrict__ gsrc, float* out, long long* clk, int steps, unsigned seed) {
    static_assert(NB % GB == 0, "B groups");
    constexpr int NG = NB / GB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < kLds / 16; i += 512) {
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (seed) {
            unsigned x = (unsigned)i * 2654435761u + seed + blockIdx.x * 40503u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x ^= x << 13; x ^= x >> 17; x ^= x << 5;
                const unsigned lo = (x & 0x80ffu) | 0x3f00u, hi = ((x >> 16) & 0x80ffu) | 0x3f00u;
                v[e] = lo | (hi << 16);
            }
        }
        ((u32x4*)smem)[i] = v;
    }
    __syncthreads();
    char* const halo = smem;                       // up to 34 x 14 rows of 128 B (60 KiB)
    char* const wbuf = smem + 64 * 1024;           // two 32 KiB weight tiles
    char* const ring = smem + 128 * 1024;          // LDS-DMA landing area (24 KiB used)
    const int r16 = lane & 15, g16 = lane >> 4;
    constexpr int WCO = 256 / (MA * 16) > 0 ? 256 / (MA * 16) : 1;       // cout waves of a 256-cout workgroup tile
    const int wco = wave % WCO, wpx = wave / WCO;
    // A rows: cout block mi of this wave = rows (wco*MA + mi)*16 + r16 of the 256-row tile; slot ^ (row & 7)
    const int wrow16 = ((wco * MA * 16 + r16) & 255) * 128, wslot16 = (g16 ^ (r16 & 7)) * 16;
    // B rows: site block q of this wave: halo row ((wpx*NB + q) * 16 + r16 + 35) -- 16 consecutive rows, conflict-free under slot ^ (row & 7)
    int xb[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) { const int xr = ((wpx * NB + q) * 16 + r16 + 35) % 470; xb[q] = xr * 128 + ((g16 ^ (xr & 7)) * 16); }
    f32x4 acc[MA][NB];
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 wf[MA], xf[2][GB];
    auto read_a1 = [&](const char* wcur, int kk, int mi) { wf[mi] = *(const u32x4*)(wcur + ((wrow16 + mi * 16 * 128 + wslot16) ^ (kk * 64))); };
    auto read_bg = [&](int kk, int grp, u32x4 (&x)[GB]) {
#pragma unroll
        for (int q = 0; q < GB; ++q) x[q] = *(const u32x4*)(halo + (xb[grp * GB + q] ^ (kk * 64)));
    };
    auto mma_g = [&](int mi, int grp, const u32x4 (&x)[GB]) {
#pragma unroll
        for (int q = 0; q < GB; ++q)
            acc[mi][grp * GB + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[mi]), __builtin_bit_cast(bf16x8, x[q]), acc[mi][grp * GB + q], 0, 0, 0);
    };
#pragma unroll
    for (int mi = 0; mi < MA; ++mi) read_a1(wbuf, 0, mi);
    read_bg(0, 0, xf[0]);
    __syncthreads();
    const long long c0 = clock64(), w0 = wall_clock64();
    int buf = 0;
    for (int s = 0; s < steps; ++s) {
        const char* const wcur = wbuf + buf * 32768;
        const char* const wnxt = wbuf + (buf ^ 1) * 32768;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char* src = gsrc + ((size_t)(s % 72) * 2 + (blockIdx.x & 1)) * 32768;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)j * 8192 + (unsigned)tid * 16),
                                             (__attribute__((address_space(3))) void*)(ring + (j % 3) * 8192 + wave * 1024), 16, 0, 0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int grp = 0; grp < NG; ++grp) {
                const bool last = grp == NG - 1;
                // the next group's B fragments (next sub-step's first group behind the last one) under this group's MFMAs
                if (!last) read_bg(kk, grp + 1, xf[(grp + 1) & 1]);
                else read_bg(kk ^ 1, 0, xf[(grp + 1) & 1]);
#pragma unroll
                for (int mi = 0; mi < MA; ++mi) {
                    mma_g(mi, grp, xf[grp & 1]);
                    if (last) read_a1(kk == 0 ? wcur : wnxt, kk ^ 1, mi);      // A block mi is free: reload it for the next sub-step / step
                }
            }
        }
        buf ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long c1 = clock64(), w1 = wall_clock64();
    float sum = 0;
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) sum += acc[i][j][r];
    out[(size_t)blockIdx.x * 512 + tid] = sum;
    if (tid == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

static unsigned g_seed = 0;
template <int MA, int NB, int GB>
static void run(const char* gsrc, float* out, long long* clk, int steps, const char* what) {
    CK(hipFuncSetAttribute((const void*)tile<MA, NB, GB>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((tile<MA, NB, GB>), dim3(256), dim3(512), kLds, 0, gsrc, out, clk, steps, g_seed);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((tile<MA, NB, GB>), dim3(256), dim3(512), kLds, 0, gsrc, out, clk, steps, g_seed);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        const double mfma_per_step = 2.0 * MA * NB;                     // per wave
        const double tflops = 256.0 * 8 * mfma_per_step * 16384.0 * steps / (ms * 1e-3) / 1e12;
        long long h[512];
        CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
        double cyc = 0, wall = 0;
        for (int b = 0; b < 256; ++b) { cyc += (double)h[b * 2]; wall += (double)h[b * 2 + 1]; }
        cyc /= 256; wall /= 256;
        const double bound = mfma_per_step * 2 * 16;                    // 2 waves per SIMD x 16 cycles per MFMA
        printf("%-44s acc %3d  reads/MFMA %.3f  %7.0f ticks/step (MFMA bound %5.0f: %.3f)  %7.1f TFLOP/s = %.3f of 2500  clock %.2f GHz\n", what, MA * NB * 4,
               (double)(MA + NB) / (MA * NB), cyc / steps, bound, bound / (cyc / steps), tflops, tflops / 2500.0, cyc / wall * 0.1);
    }
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 3000;
    g_seed = argc > 2 ? (unsigned)atoi(argv[2]) : 0;
    char* gsrc; float* out; long long* clk;
    CK(hipMalloc(&gsrc, 8 << 20)); CK(hipMemset(gsrc, g_seed ? 0x3e : 0, 8 << 20));
    CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&clk, 512 * 8));
    printf("# operands: %s; one step = 64 channels: barrier, 4 LDS-DMA per wave, 2 x (MA + NB) ds_read_b128, 2 x MA x NB MFMA 16x16x32 per wave\n", g_seed ? "uniform random bf16" : "zero");
    run<4, 8, 4>(gsrc, out, clk, steps, "64 couts x 128 sites (shipped tile)");
    run<4, 12, 4>(gsrc, out, clk, steps, "64 couts x 192 sites");
    run<8, 6, 3>(gsrc, out, clk, steps, "128 couts x 96 sites");
    run<6, 8, 4>(gsrc, out, clk, steps, "96 couts x 128 sites");
    run<4, 8, 4>(gsrc, out, clk, steps, "64 couts x 128 sites (again)");
    return 0;
}
