// Instruction-mix microbenchmark for MI355X: what fraction of the MFMA rate does the K-loop MIX of conv_igemm_v2 reach,
// with nothing else in the way?  One 8-wave workgroup per CU (2 waves per SIMD), every wave loops over "steps" of
// 4 x (8 x v_mfma_f32_32x32x16_bf16) with READS ds_read_b128 per group of 8 MFMAs from a static, zero-filled LDS image
// (conflict-free addresses of the kernel's own form, two-stage register pipeline, the same sched_group_barrier
// interleave), optionally a workgroup barrier per step and DMA LDS-DMA requests of 1 KiB per wave per step.
// All operands are zero, so the chip holds its full clock: the printed fraction is cycles, not power.
//
//   mix_probe [steps]      prints one line per variant: MFMA rate as a fraction of 1024 FLOP/cycle/SIMD at the measured clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kLds = 144 * 1024;
__constant__ int tapy[16] = {-1, -1, -1, 0, 0, 0, 1, 1, 1, -1, -1, -1, 0, 0, 0, 1};
__constant__ int tapx[16] = {-1, 0, 1, -1, 0, 1, -1, 0, 1, -1, 0, 1, -1, 0, 1, -1};
__constant__ int tapd[16] = {144, 144, 4608, 144, 144, 4608, 144, 144, -10080, 144, 144, 4608, 144, 144, 4608, -9792};   // (sums to 0: the addresses cycle)

template <int READS, bool BARRIER, int DMA, int STREAM = 0, bool SWAP = false, int XA = 0, bool PAD = false>
__global__ __launch_bounds__(512, 2) void mix(const char* __restrict__ gsrc, float* out, long long* clk, int steps, unsigned seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 31, h = lane >> 5, wco = wave & 3, wpx = wave >> 2;
    for (int i = tid; i < kLds / 16; i += 512) {
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (seed) {                                            // uniform random bf16 pairs in [-1, 1): sign + exponent 0x3f00..0x3f7f + mantissa
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
    char* const halo = smem;                       // 34 x 10 rows of 128 B
    char* const wbuf = smem + 48 * 1024;           // two 32 KiB tiles
    char* const ring = smem + 112 * 1024;          // LDS-DMA landing area (32 KiB)
    const int wrow = (wco * 64 + px) * 128, wslot0 = (h ^ ((px >> 1) & 7)) * 16;
    int xa[4];
#pragma unroll
    for (int pj = 0; pj < 4; ++pj) { const int xr = (wpx * 4 + pj + 1) * 34 + px + 1; xa[pj] = PAD ? xr * 144 + h * 16 : xr * 128 + ((h ^ ((xr >> 1) & 7)) * 16); }
    const int v_tapy = tapy[lane & 15], v_tapx = tapx[lane & 15], v_tapd = tapd[lane & 15];
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    u32x4 wfA[2] = {}, xfA[4] = {}, wfB[2] = {}, xfB[4] = {};
    auto read_frags = [&](const char* wcur, int kk, u32x4 (&wf)[2], u32x4 (&xf)[4]) {
        int n = 0;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) if (n++ < READS) wf[mi] = *(const u32x4*)(wcur + ((wrow + mi * 32 * 128 + wslot0) ^ (kk * 32)));
#pragma unroll
        for (int pj = 0; pj < 4; ++pj) if (n++ < READS) xf[pj] = *(const u32x4*)(halo + (PAD ? xa[pj] + kk * 32 : (xa[pj] ^ (kk * 32))));
    };
    auto mma8 = [&](const u32x4 (&wf)[2], const u32x4 (&xf)[4]) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int pj = 0; pj < 4; ++pj)
                acc[mi][pj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[mi]), __builtin_bit_cast(bf16x8, xf[pj]), acc[mi][pj], 0, 0, 0);
    };
#define IL()                                                                          \
    _Pragma("unroll") for (int q_ = 0; q_ < 6; ++q_) {                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                            \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                            \
    }                                                                                 \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) { wfA[mi] = *(const u32x4*)(wbuf + wrow + mi * 4096 + wslot0); wfB[mi] = *(const u32x4*)(wbuf + 32768 + wrow + mi * 4096 + wslot0); }
#pragma unroll
    for (int pj = 0; pj < 4; ++pj) { xfA[pj] = *(const u32x4*)(halo + xa[pj]); xfB[pj] = *(const u32x4*)(halo + (xa[pj] ^ 32)); }
    __syncthreads();
    const long long c0 = clock64(), w0 = wall_clock64();
    int buf = 0;
    for (int s = 0; s < steps; ++s) {
        const char* const wcur = wbuf + buf * 32768;
        if (BARRIER) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
        read_frags(wcur, 0, wfA, xfA);
        __builtin_amdgcn_sched_barrier(0);
        const char* src = gsrc;
        if (STREAM == 1) src += ((size_t)(s % 72) * 2 + (blockIdx.x & 1)) * 32768;            // the trunk layer's walk: 72 steps x 2 cout tiles x 32 KiB
        if (STREAM == 2) src += ((size_t)(s % 72) * 2 + (blockIdx.x & 1)) * 32768 + (size_t)(blockIdx.x >> 1) * 0;   // (same; placeholder for per-XCD variants)
#pragma unroll
        for (int j = 0; j < DMA; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)j * 8192 + (unsigned)tid * 16),
                                             (__attribute__((address_space(3))) void*)(ring + j * 8192 + wave * 1024), 16, 0, 0);
        if (SWAP && s % 9 == 8) {                              // the halo chunk change: barrier, 6 x 16 B per thread into the halo tile, (next step's barrier)
            u32x4 keep[6];                                     // (rewrites the tile with its own content: the operand statistics stay what they were)
#pragma unroll
            for (int j = 0; j < 6; ++j) keep[j] = *(const u32x4*)(halo + (tid + j * 512) * 16);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 6; ++j) *(u32x4*)(halo + (tid + j * 512) * 16) = keep[j];
        }
        __builtin_amdgcn_sched_barrier(0);
        read_frags(wcur, 1, wfB, xfB); mma8(wfA, xfA); IL()
        read_frags(wcur, 2, wfA, xfA); mma8(wfB, xfB); IL()
        read_frags(wcur, 3, wfB, xfB); mma8(wfA, xfA); IL()
        if (XA) {                                              // the per-tap address update, where the kernel has it: under the step's last MFMAs
            int dy, dx, delta;
            if (XA == 2) {                                     // tap table in lanes, read back with v_readlane: no scalar load in the loop
                dy = __builtin_amdgcn_readlane(v_tapy, s & 15); dx = __builtin_amdgcn_readlane(v_tapx, s & 15); delta = __builtin_amdgcn_readlane(v_tapd, s & 15);
            } else { dy = tapy[s & 15]; dx = tapx[s & 15]; delta = tapd[s & 15]; }
            if (PAD) {
#pragma unroll
                for (int pj = 0; pj < 4; ++pj) xa[pj] += delta;
            } else {
#pragma unroll
                for (int pj = 0; pj < 4; ++pj) { const int xr = (wpx * 4 + pj + 1 + dy) * 34 + px + 1 + dx; xa[pj] = xr * 128 + ((h ^ ((xr >> 1) & 7)) * 16); }
            }
        }
        mma8(wfB, xfB);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        buf ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long c1 = clock64(), w1 = wall_clock64();
    float sum = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    out[(size_t)blockIdx.x * 512 + tid] = sum;
    if (tid == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int READS, bool BARRIER, int DMA, int STREAM = 0, bool SWAP = false, int XA = 0, bool PAD = false>
__global__ __launch_bounds__(512, 2) void mix16(const char* __restrict__ gsrc, float* out, long long* clk, int steps, unsigned seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 31, h = lane >> 5, wco = wave & 3, wpx = wave >> 2;
    for (int i = tid; i < kLds / 16; i += 512) {
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (seed) {                                            // uniform random bf16 pairs in [-1, 1): sign + exponent 0x3f00..0x3f7f + mantissa
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
    char* const halo = smem;                       // 34 x 10 rows of 128 B
    char* const wbuf = smem + 48 * 1024;           // two 32 KiB tiles
    char* const ring = smem + 112 * 1024;          // LDS-DMA landing area (32 KiB)
    const int wrow = (wco * 64 + px) * 128, wslot0 = (h ^ ((px >> 1) & 7)) * 16;
    int xa[4];
#pragma unroll
    for (int pj = 0; pj < 4; ++pj) { const int xr = (wpx * 4 + pj + 1) * 34 + px + 1; xa[pj] = PAD ? xr * 144 + h * 16 : xr * 128 + ((h ^ ((xr >> 1) & 7)) * 16); }
    const int v_tapy = tapy[lane & 15], v_tapx = tapx[lane & 15], v_tapd = tapd[lane & 15];
    // the same wave tile (64 couts x 128 sites, 128 accumulator registers) from v_mfma_f32_16x16x32_bf16: 4 x 8 tiles of 16 x 16;
    // one k32 step = 4 A + 8 B fragments (the same 12 KiB per 64 x 128 x 32 MACs as two k16 steps of the 32x32 form), read in two
    // stages (4 A + 4 B, then 4 B) of 16 MFMAs each
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r16 = lane & 15, g16 = lane >> 4;
    const int wrow16 = (wco * 64 + r16) * 128, wslot16 = (g16 ^ (r16 & 7)) * 16;
    int xb[4];                                              // one address per pixel row; its second 16 sites are 2 KiB further (same swizzle)
#pragma unroll
    for (int pj = 0; pj < 4; ++pj) { const int xr = (wpx * 4 + pj + 1) * 34 + r16 + 1; xb[pj] = xr * 128 + ((g16 ^ (xr & 7)) * 16); }
    u32x4 wf[4] = {}, xlo[4] = {}, xhi[4] = {};
    auto read_a = [&](const char* wcur, int kk) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) wf[mi] = *(const u32x4*)(wcur + ((wrow16 + mi * 16 * 128 + wslot16) ^ (kk * 64)));
    };
    auto read_b = [&](int kk, int half, u32x4 (&xf)[4]) {
#pragma unroll
        for (int pj = 0; pj < 4; ++pj) xf[pj] = *(const u32x4*)(halo + (xb[half * 2 + (pj >> 1)] ^ (kk * 64)) + (pj & 1) * 2048);
    };
    auto mma16 = [&](int half, const u32x4 (&xf)[4]) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int pj = 0; pj < 4; ++pj)
                acc[mi][half * 4 + pj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[mi]), __builtin_bit_cast(bf16x8, xf[pj]), acc[mi][half * 4 + pj], 0, 0, 0);
    };
#define IL16(NR)                                                                      \
    _Pragma("unroll") for (int q_ = 0; q_ < NR; ++q_) {                              \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                            \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                            \
    }                                                                                 \
    __builtin_amdgcn_sched_group_barrier(0x008, 16 - 2 * NR, 0);
    read_a(wbuf, 0); read_b(0, 0, xlo); read_b(0, 1, xhi);
    __syncthreads();
    const long long c0 = clock64(), w0 = wall_clock64();
    int buf = 0;
    for (int s = 0; s < steps; ++s) {
        const char* const wcur = wbuf + buf * 32768;
        if (BARRIER) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
        read_a(wcur, 0); read_b(0, 0, xlo);
        __builtin_amdgcn_sched_barrier(0);
        const char* src = gsrc;
        if (STREAM == 1) src += ((size_t)(s % 72) * 2 + (blockIdx.x & 1)) * 32768;            // the trunk layer's walk: 72 steps x 2 cout tiles x 32 KiB
        if (STREAM == 2) src += ((size_t)(s % 72) * 2 + (blockIdx.x & 1)) * 32768 + (size_t)(blockIdx.x >> 1) * 0;   // (same; placeholder for per-XCD variants)
#pragma unroll
        for (int j = 0; j < DMA; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)j * 8192 + (unsigned)tid * 16),
                                             (__attribute__((address_space(3))) void*)(ring + j * 8192 + wave * 1024), 16, 0, 0);
        if (SWAP && s % 9 == 8) {                              // the halo chunk change: barrier, 6 x 16 B per thread into the halo tile, (next step's barrier)
            u32x4 keep[6];                                     // (rewrites the tile with its own content: the operand statistics stay what they were)
#pragma unroll
            for (int j = 0; j < 6; ++j) keep[j] = *(const u32x4*)(halo + (tid + j * 512) * 16);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 6; ++j) *(u32x4*)(halo + (tid + j * 512) * 16) = keep[j];
        }
        __builtin_amdgcn_sched_barrier(0);
        read_b(0, 1, xhi); mma16(0, xlo); IL16(4)                  // k32 step 0, sites 0..63 under the reads of sites 64..127
        mma16(1, xhi);                                             // (A of step 1 cannot be read before these have issued: one A set)
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        read_a(wcur, 1); read_b(1, 0, xlo);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        read_b(1, 1, xhi); mma16(0, xlo); IL16(4)
        if (XA) {                                              // the per-tap address update, where the kernel has it: under the step's last MFMAs
            int dy, dx, delta;
            if (XA == 2) {                                     // tap table in lanes, read back with v_readlane: no scalar load in the loop
                dy = __builtin_amdgcn_readlane(v_tapy, s & 15); dx = __builtin_amdgcn_readlane(v_tapx, s & 15); delta = __builtin_amdgcn_readlane(v_tapd, s & 15);
            } else { dy = tapy[s & 15]; dx = tapx[s & 15]; delta = tapd[s & 15]; }
            if (PAD) {
#pragma unroll
                for (int pj = 0; pj < 4; ++pj) xb[pj] += delta;
            } else {
#pragma unroll
                for (int pj = 0; pj < 4; ++pj) { const int xr = (wpx * 4 + pj + 1 + dy) * 34 + r16 + 1 + dx; xb[pj] = xr * 128 + ((g16 ^ (xr & 7)) * 16); }
            }
        }
        mma16(1, xhi);
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        buf ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long c1 = clock64(), w1 = wall_clock64();
    float sum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) sum += acc[i][j][r];
    out[(size_t)blockIdx.x * 512 + tid] = sum;
    if (tid == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

static unsigned g_seed = 0;
template <int READS, bool BARRIER, int DMA, int STREAM = 0, bool SWAP = false, int XA = 0, bool PAD = false>
static void run(const char* gsrc, float* out, long long* clk, int steps, const char* what) {
    CK(hipFuncSetAttribute((const void*)mix<READS, BARRIER, DMA, STREAM, SWAP, XA, PAD>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((mix<READS, BARRIER, DMA, STREAM, SWAP, XA, PAD>), dim3(256), dim3(512), kLds, 0, gsrc, out, clk, steps, g_seed);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((mix<READS, BARRIER, DMA, STREAM, SWAP, XA, PAD>), dim3(256), dim3(512), kLds, 0, gsrc, out, clk, steps, g_seed);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double tflops = 256.0 * 8 * 32 * 32768.0 * steps / (ms * 1e-3) / 1e12;
    long long h[512];
    CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
    double cyc = 0, wall = 0;
    for (int b = 0; b < 256; ++b) { cyc += (double)h[b * 2]; wall += (double)h[b * 2 + 1]; }
    cyc /= 256; wall /= 256;
    const double per_step = cyc / steps;                    // shader cycles per step of one wave (= of the workgroup)
    printf("%-58s %7.0f ticks/step  %7.1f TFLOP/s by host events = %.3f of 2500  (tick rate %.2f GHz)\n", what, per_step, tflops, tflops / 2500.0, cyc / wall * 0.1);
}

template <int READS, bool BARRIER, int DMA, int STREAM = 0, bool SWAP = false, int XA = 0, bool PAD = false>
static void run16(const char* gsrc, float* out, long long* clk, int steps, const char* what) {
    CK(hipFuncSetAttribute((const void*)mix16<READS, BARRIER, DMA, STREAM, SWAP, XA, PAD>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((mix16<READS, BARRIER, DMA, STREAM, SWAP, XA, PAD>), dim3(256), dim3(512), kLds, 0, gsrc, out, clk, steps, g_seed);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((mix16<READS, BARRIER, DMA, STREAM, SWAP, XA, PAD>), dim3(256), dim3(512), kLds, 0, gsrc, out, clk, steps, g_seed);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double tflops = 256.0 * 8 * 32 * 32768.0 * steps / (ms * 1e-3) / 1e12;
    long long h[512];
    CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
    double cyc = 0, wall = 0;
    for (int b = 0; b < 256; ++b) { cyc += (double)h[b * 2]; wall += (double)h[b * 2 + 1]; }
    cyc /= 256; wall /= 256;
    printf("%-58s %7.0f ticks/step  %7.1f TFLOP/s by host events = %.3f of 2500  (tick rate %.2f GHz)\n", what, cyc / steps, tflops, tflops / 2500.0, cyc / wall * 0.1);
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 4000;
    g_seed = argc > 2 ? (unsigned)atoi(argv[2]) : 0;      // 0: zero operands (cycles); else: uniform random bf16 operands (the power cap)
    char* gsrc; float* out; long long* clk;
    CK(hipMalloc(&gsrc, 8 << 20)); CK(hipMemset(gsrc, g_seed ? 0x3e : 0, 8 << 20));
    CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&clk, 512 * 8));
    printf("# one step = 32 MFMA per wave, 2 waves per SIMD: 2048 MFMA cycles per step at 100 %%; operands: %s\n", g_seed ? "uniform random bf16" : "zero");
    run<0, false, 0>(gsrc, out, clk, steps, "MFMA only");
    run<6, false, 0>(gsrc, out, clk, steps, "+ 6 ds_read_b128 per 8 MFMA (the kernel's 0.75)");
    run<4, false, 0>(gsrc, out, clk, steps, "+ 4 ds_read_b128 per 8 MFMA (0.50)");
    run<3, false, 0>(gsrc, out, clk, steps, "+ 3 ds_read_b128 per 8 MFMA");
    run<6, true, 0>(gsrc, out, clk, steps, "+ 6 reads, barrier per step");
    run<6, false, 4>(gsrc, out, clk, steps, "+ 6 reads, 4 LDS-DMA requests per wave-step");
    run<6, true, 4>(gsrc, out, clk, steps, "+ 6 reads, barrier, 4 LDS-DMA (the kernel's step)");
    run<4, true, 4>(gsrc, out, clk, steps, "+ 4 reads, barrier, 4 LDS-DMA");
    run<0, true, 4>(gsrc, out, clk, steps, "+ 0 reads, barrier, 4 LDS-DMA");
    run<6, true, 4, 1>(gsrc, out, clk, steps, "kernel's step, LDS-DMA streaming 4.5 MiB of tiles");
    run<6, true, 4, 1, true>(gsrc, out, clk, steps, "... + halo chunk change every 9 steps");
    run<6, true, 4, 1, true, 1>(gsrc, out, clk, steps, "... + per-step address update (XOR-swizzled rows)");
    run<6, true, 4, 1, true, 1, true>(gsrc, out, clk, steps, "... same with 144-byte padded rows (linear addresses)");
    run<6, true, 4, 1, false, 1, true>(gsrc, out, clk, steps, "... padded rows, no halo chunk change");
    run<6, true, 4, 1, true, 2>(gsrc, out, clk, steps, "XOR rows, tap table in lanes (v_readlane, no s_load)");
    run<6, true, 4, 1, true, 2, true>(gsrc, out, clk, steps, "padded rows, tap table in lanes");
    run<6, true, 4, 1, false, 2, true>(gsrc, out, clk, steps, "padded rows, lanes table, no halo chunk change");
    printf("# the same step from v_mfma_f32_16x16x32_bf16 (64 MFMAs of half the size per wave-step, the same 24 fragment reads)\n");
    run16<6, false, 0>(gsrc, out, clk, steps, "16x16x32: MFMA + 24 ds_read_b128 per step");
    run16<6, true, 4>(gsrc, out, clk, steps, "16x16x32: + barrier, 4 LDS-DMA (the kernel's step)");
    run16<6, true, 4, 1>(gsrc, out, clk, steps, "16x16x32: kernel's step, LDS-DMA streaming");
    run16<6, true, 4, 1, true>(gsrc, out, clk, steps, "16x16x32: ... + halo chunk change every 9 steps");
    run16<6, true, 4, 1, true, 1>(gsrc, out, clk, steps, "16x16x32: ... + per-step address update (XOR rows)");
    run<6, true, 4, 1, true, 1>(gsrc, out, clk, steps, "32x32x16 again: ... + per-step address update");
    return 0;
}
