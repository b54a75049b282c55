// MFMA issue-rate microbenchmark: waves/WG and accumulators per wave as parameters.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int NACC>
__global__ __launch_bounds__(512) void k32(float* out, int iters, unsigned seed) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4 a = {0x3f803f80u ^ (threadIdx.x * 2654435761u & 0x007f007fu), 0x3e803f00u, 0x3f003e80u, 0x3f803f00u ^ seed};
    u32x4 b = {0x3f003f80u, 0x3e803f80u ^ (threadIdx.x * 40503u & 0x007f007fu), 0x3f803e80u, 0x3f003f00u};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(512) void k16(float* out, int iters, unsigned seed) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    u32x4 a = {0x3f803f80u ^ (threadIdx.x * 2654435761u & 0x007f007fu), 0x3e803f00u, 0x3f003e80u, 0x3f803f00u ^ seed};
    u32x4 b = {0x3f003f80u, 0x3e803f80u ^ (threadIdx.x * 40503u & 0x007f007fu), 0x3f803e80u, 0x3f003f00u};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main(int argc, char** argv) {
    float* out; hipMalloc(&out, 4096 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int threads : {256, 512}) for (int blocks : {256, 512, 1024}) {
        for (int variant = 0; variant < 2; ++variant) {
            auto launch = [&]() { if (variant == 0) hipLaunchKernelGGL(k32<8>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1u);
                                  else hipLaunchKernelGGL(k16<16>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1u); };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0, 0); for (int r = 0; r < 5; ++r) launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            double flops = (variant == 0 ? 8.0 * 2 * 32 * 32 * 16 : 16.0 * 2 * 16 * 16 * 32) * iters * (threads / 64) * blocks;
            printf("%s threads=%d blocks=%d : %.3f ms  %.0f TFLOP/s\n", variant == 0 ? "32x32x16 x8acc " : "16x16x32 x16acc", threads, blocks, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
