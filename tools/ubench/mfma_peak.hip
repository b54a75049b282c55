// MFMA issue-rate microbenchmark for MI355X: what v_mfma_f32_32x32x16_bf16 sustains on this box, by operand data.
//
//   mfma_peak [seconds-per-arm] [zeros|ones|random|all] [32|16]     (default 10 s, all three arms, the 32x32x16 instruction)
//
// Arms: operand fill = zeros | constant 1.0 | uniform random [-1,1) bf16 (per lane, per register; the accumulators
// see a random walk), each held for `seconds` of back-to-back launches of 256 CUs x 8 waves x 8 independent
// accumulators (the issue-bound form: 2 waves/SIMD, no memory traffic at all).  Per second of wall time it prints
// TFLOP/s and the shader clock the kernel saw (s_memtime ticks = shader cycles, against the 100 MHz wall_clock64),
// so the DVFS give-back (MI355X_MICROARCH.md "DVFS give-back") is visible as a number: the nominal 2.5 PFLOP/s is
// 256 CU x 4 SIMD x 1024 FLOP/cycle x 2.4 GHz; what a kernel can reach on REAL data is this figure at the clock
// the chip sustains under that data.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int NACC>
__global__ __launch_bounds__(512) void k32(const u32x4* __restrict__ ops, float* out, long long* clk, int iters) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 a[2], b[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = ops[t * 6 + i];
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = ops[t * 6 + 2 + i];
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i & 1]),
                                                             __builtin_bit_cast(bf16x8, b[i & 3]), acc[i], 0, 0, 0);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[t] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

// the same loop on v_mfma_f32_16x16x32_bf16 (half the FLOPs per instruction, twice the operand bytes per FLOP): argv[3] = 16
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int NACC>
__global__ __launch_bounds__(512) void k16(const u32x4* __restrict__ ops, float* out, long long* clk, int iters) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 a[2], b[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = ops[t * 6 + i];
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = ops[t * 6 + 2 + i];
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i & 1]),
                                                             __builtin_bit_cast(bf16x8, b[i & 3]), acc[i], 0, 0, 0);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[t] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

static unsigned short bf16_of(float f) {
    unsigned u; __builtin_memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 10.0;
    const int only = argc > 2 ? (argv[2][0] == 'z' ? 0 : argv[2][0] == 'o' ? 1 : argv[2][0] == 'a' ? -1 : 2) : -1;
    const bool v16 = argc > 3 && atoi(argv[3]) == 16;
    const int blocks = 256, threads = 512, iters = v16 ? 40000 : 20000;
    auto launch = [&](u32x4* ops, float* out, long long* clk) {
        if (v16) hipLaunchKernelGGL(k16<8>, dim3(blocks), dim3(threads), 0, 0, ops, out, clk, iters);
        else hipLaunchKernelGGL(k32<8>, dim3(blocks), dim3(threads), 0, 0, ops, out, clk, iters);
    };
    const size_t nthr = (size_t)blocks * threads;
    u32x4* d_ops; float* d_out; long long* d_clk;
    hipMalloc(&d_ops, nthr * 6 * sizeof(u32x4)); hipMalloc(&d_out, nthr * 4); hipMalloc(&d_clk, blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flop_per_launch = 8.0 * 2 * (v16 ? 16 * 16 * 32 : 32 * 32 * 16) * (double)iters * (threads / 64) * blocks;
    const char* names[3] = {"zeros", "constant 1.0", "uniform random [-1,1)"};
    for (int fill = 0; fill < 3; ++fill) {
        if (only >= 0 && fill != only) continue;
        std::vector<unsigned short> h(nthr * 6 * 8);
        unsigned long long st = 0x9e3779b97f4a7c15ull;
        for (auto& v : h) {
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            const float r = (float)((st >> 40) & 0xffffff) / 8388608.0f - 1.0f;       // [-1, 1)
            v = fill == 0 ? 0 : fill == 1 ? 0x3f80 : bf16_of(r);
        }
        hipMemcpy(d_ops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        launch(d_ops, d_out, d_clk);
        hipDeviceSynchronize();
        printf("== %s, operands: %s, %.0f s sustained, %d x %d threads, 8 accumulators per wave\n", v16 ? "v_mfma_f32_16x16x32_bf16" : "v_mfma_f32_32x32x16_bf16", names[fill], seconds, blocks, threads);
        double elapsed = 0, best = 0, sum_tf = 0; int nwin = 0;
        while (elapsed < seconds) {
            // one ~1 s window of back-to-back launches
            int n = 0; float ms = 0;
            hipEventRecord(e0, 0);
            do { for (int k = 0; k < 8; ++k) launch(d_ops, d_out, d_clk);
                 n += 8; hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); } while (ms < 1000.f);
            long long hc[512]; hipMemcpy(hc, d_clk, blocks * 16, hipMemcpyDeviceToHost);
            double cyc = 0, wall = 0; for (int b = 0; b < blocks; ++b) { cyc += hc[2 * b]; wall += hc[2 * b + 1]; }
            const double ghz = cyc / wall * 0.1;                                        // wall_clock64 = 100 MHz
            const double tf = flop_per_launch * n / ms / 1e9;
            const double interval = ghz * 1e9 * 1024.0 * (v16 ? 16384.0 : 32768.0) / (tf * 1e12);          // shader cycles between MFMAs on one SIMD (32 = issue-bound)
            printf("  t=%5.1fs  %7.1f TFLOP/s  shader clock %.3f GHz  MFMA issue interval %.1f cycles per SIMD\n", elapsed + ms / 1e3, tf, ghz, interval);
            elapsed += ms / 1e3; sum_tf += tf; ++nwin; if (tf > best) best = tf;
        }
        printf("  mean %.1f TFLOP/s (best window %.1f) = %.3f of the nominal 2500\n", sum_tf / nwin, best, sum_tf / nwin / 2500.0);
    }
    return 0;
}
