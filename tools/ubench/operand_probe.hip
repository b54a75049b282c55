// Does the power the MFMA draws depend on WHICH operand carries the zeros?  v_mfma_f32_32x32x16_bf16 in the issue-bound form
// of mfma_peak (256 CUs x 8 waves x 8 accumulators, no memory traffic), operands uniform random bf16 in [-1, 1) with a
// fraction of the ELEMENTS of A and/or B forced to zero (post-ReLU activations are ~half zeros; the conv kernels feed the
// weights as A and the pixels as B).  Under the package power cap the sustained TFLOP/s is the energy per MFMA.
//   operand_probe [seconds per arm]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(512) void k(const u32x4* __restrict__ ops, float* out, int iters) {
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 a[2], b[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = ops[t * 6 + i];
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = ops[t * 6 + 2 + i];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i >> 2]), __builtin_bit_cast(bf16x8, b[i & 3]), acc[i], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[t] = s;
}

static unsigned short bf16_of(float f) { unsigned u; __builtin_memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    const int blocks = 256, threads = 512, iters = 20000;
    const size_t nthr = (size_t)blocks * threads;
    u32x4* d_ops; float* d_out;
    CK(hipMalloc(&d_ops, nthr * 6 * 16)); CK(hipMalloc(&d_out, nthr * 4));
    std::vector<unsigned short> h(nthr * 6 * 8);
    const char* names[5] = {"A random, B random", "A random, B half zeros (the kernels: pixels are B)", "A half zeros, B random (roles swapped)", "A half zeros, B half zeros", "A random, B three quarters zeros"};
    const double za[5] = {0, 0, 0.5, 0.5, 0}, zb[5] = {0, 0.5, 0, 0.5, 0.75};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int arm = 0; arm < 5; ++arm) {
        srand(1);
        for (size_t t = 0; t < nthr; ++t)
            for (int r = 0; r < 6; ++r)
                for (int e = 0; e < 8; ++e) {
                    const double z = r < 2 ? za[arm] : zb[arm];
                    const float v = (float)rand() / RAND_MAX * 2.f - 1.f;
                    h[(t * 6 + r) * 8 + e] = ((double)rand() / RAND_MAX < z) ? 0 : bf16_of(v);
                }
        CK(hipMemcpy(d_ops, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d_ops, d_out, iters);
        CK(hipDeviceSynchronize());
        double tot_ms = 0; long launches = 0;
        while (tot_ms < seconds * 1e3) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d_ops, d_out, iters);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot_ms += ms; launches += 10;
        }
        const double flops = (double)launches * nthr / 64 * iters * 8 * 32768.0;
        printf("%-52s %8.1f TFLOP/s\n", names[arm], flops / (tot_ms * 1e-3) / 1e12);
    }
    return 0;
}
