// Standalone timing harness for conv_igemm variants (tuning only; not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DIDC_ABL_xxx] ablate.hip -o ablate_xxx
// Usage: ablate [N=32] [HW=32] [C=512] [halo=1] [wm=2] [wp=2] [prec=1]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../interactive_deep_colorization_amd/csrc/idc_igemm.hip"
#include "../../interactive_deep_colorization_amd/csrc/idc_v2.hip"
#include "../../interactive_deep_colorization_amd/csrc/idc_conv1.hip"
#include "../../interactive_deep_colorization_amd/csrc/idc_wino.hip"
#include "../../interactive_deep_colorization_amd/csrc/idc_v2m.hip"
#include "../../interactive_deep_colorization_amd/csrc/idc_dsm.hip"
#include "../../interactive_deep_colorization_amd/csrc/idc_kw.hip"

__global__ void fill_bf16(unsigned short* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        float f = ((x & 0xffff) / 32768.0f - 1.0f) * 0.5f;
        __bf16 h = (__bf16)f;
        p[i] = __builtin_bit_cast(unsigned short, h);
    }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    int N = argc > 1 ? atoi(argv[1]) : 32, HW = argc > 2 ? atoi(argv[2]) : 32, C = argc > 3 ? atoi(argv[3]) : 512;
    int halo = argc > 4 ? atoi(argv[4]) : 1, wm = argc > 5 ? atoi(argv[5]) : 2, wp = argc > 6 ? atoi(argv[6]) : 2;
    int prec = argc > 7 ? atoi(argv[7]) : 1;
    int v2 = argc > 8 ? atoi(argv[8]) : 0;
    int ntaps = argc > 9 ? atoi(argv[9]) : 9;
    const int eb = prec ? 2 : 4, kc = 128 / eb;
    idc::ConvArgs a; memset(&a, 0, sizeof(a));
    size_t act = (size_t)N * HW * HW * C * eb, wbytes = (size_t)9 * (C / kc) * (C / 64) * 8192;
    void *in, *out, *w; float *bias;
    CK(hipMalloc(&in, act)); CK(hipMalloc(&out, act)); CK(hipMalloc(&w, wbytes)); CK(hipMalloc(&bias, C * 4));
    hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, 0, (unsigned short*)in, act / 2, 1u);
    hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, 0, (unsigned short*)w, wbytes / 2, 7u);
    CK(hipMemset(bias, 0, C * 4));
    CK(idc::init_kernels());
    a.in = in; a.out = out; a.wgt = w; a.bias = bias; a.N = N; a.Hs = HW; a.Ws = HW; a.si = 1; a.so = 1;
    a.nkc = C / kc; a.ncg = C / 64; a.nphase = 1; a.ntaps = ntaps; a.tiles_x = v2 ? (HW + 31) / 32 : (HW + 15) / 16; a.tiles_y = (HW + 4 * wp - 1) / (4 * wp);
    a.act = 1; a.out_f32 = 0;
    for (int t = 0; t < 9; ++t) { a.dy[t] = (t / 3 - 1) * halo; a.dx[t] = (t % 3 - 1) * halo; a.tw[t] = t; }
    if (v2 == 2) {      // fused shortcut: 4-phase deconv of `in` (C ch @ HW) + 3x3 conv of in2 (C2 ch @ 2HW)
        int C2 = argc > 10 ? atoi(argv[10]) : C;
        void *in2, *w2;
        size_t act2 = (size_t)N * 4 * HW * HW * C2 * 2, wb2 = (size_t)9 * (C2 / 64) * (C / 64) * 8192;
        CK(hipMalloc(&in2, act2)); CK(hipMalloc(&w2, wb2));
        hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, 0, (unsigned short*)in2, act2 / 2, 3u);
        hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, 0, (unsigned short*)w2, wb2 / 2, 9u);
        CK(hipFree(out)); CK(hipMalloc(&out, act * 4));
        CK(hipFree(w)); CK(hipMalloc(&w, (size_t)16 * (C / 64) * (C / 64) * 8192));
        hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, 0, (unsigned short*)w, (size_t)16 * (C / 64) * (C / 64) * 8192 / 2, 7u);
        a.out = out; a.wgt = w; a.in2 = in2; a.wgt2 = w2; a.nkc2 = C2 / 64; a.nphase = 4; a.ntaps = 4; a.so = 2;
        static const int T_k[2][2] = {{1, 3}, {0, 2}}, T_d[2][2] = {{0, -1}, {1, 0}};
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c) { int ph = r * 2 + c; a.ro[ph] = r; a.co[ph] = c;
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) { int t = ph * 9 + i * 2 + j; a.dy[t] = T_d[r][i]; a.dx[t] = T_d[c][j]; a.tw[t] = T_k[r][i] * 4 + T_k[c][j]; } }
    }
    if (v2 == 3) {      // 4-phase deconv of `in` (C ch @ HW) -> C ch @ 2HW, + bf16 shortcut partial sums (conv8_1/9_1/10_1)
        void* resid;
        CK(hipFree(out)); CK(hipMalloc(&out, act * 4)); CK(hipMalloc(&resid, act * 4));
        hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, 0, (unsigned short*)resid, act * 4 / 2, 5u);
        CK(hipFree(w)); CK(hipMalloc(&w, (size_t)16 * (C / 64) * (C / 64) * 8192));
        hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, 0, (unsigned short*)w, (size_t)16 * (C / 64) * (C / 64) * 8192 / 2, 7u);
        a.out = out; a.wgt = w; a.resid = resid; a.resid_bf16 = 1; a.nphase = 4; a.ntaps = 4; a.so = 2; ntaps = 4;
        static const int T_k[2][2] = {{1, 3}, {0, 2}}, T_d[2][2] = {{0, -1}, {1, 0}};
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c) { int ph = r * 2 + c; a.ro[ph] = r; a.co[ph] = c;
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) { int t = ph * 9 + i * 2 + j; a.dy[t] = T_d[r][i]; a.dx[t] = T_d[c][j]; a.tw[t] = T_k[r][i] * 4 + T_k[c][j]; } }
    }
    void* zeros = nullptr; float* partial = nullptr;
    CK(hipMalloc(&zeros, 256)); CK(hipMemset(zeros, 0, 256)); a.zeros = zeros;
    if (v2 == 4 || v2 == 5) {   // batch-1 small-tile comparison: 4 = conv_click (whole K slice by LDS-DMA), 5 = conv_igemm with split-K
        const int ksplit = argc > 10 ? atoi(argv[10]) : a.nkc;
        a.kc_per = (a.nkc + ksplit - 1) / ksplit; a.ksplit = (a.nkc + a.kc_per - 1) / a.kc_per;
        if (a.ksplit < 2) { a.ksplit = 1; a.kc_per = a.nkc; }
        CK(hipMalloc((void**)&partial, (size_t)a.ksplit * N * HW * HW * C * 4)); a.partial = partial;
        if (v2 == 4) wm = 1;
        a.tiles_x = (HW + 15) / 16; a.tiles_y = (HW + 4 * wp - 1) / (4 * wp);
        printf("small-tile mode %s: ksplit %d, kc_per %d, %d workgroups\n", v2 == 4 ? "conv_click" : "conv_igemm", a.ksplit, a.kc_per,
               a.tiles_x * a.tiles_y * N * (a.ncg / wm) * a.ksplit);
    }
    if (v2 == 7) {      // model1 block (conv1_1 + conv1_2 fused, conv1_block_fused_t): N images of HW x HW, fp32 input planes; per-phase stamps with IDC_TIMING
        float *pl, *pab, *pm, *hb; void *w1, *w2, *o2;
        const size_t hw = (size_t)HW * HW;
        CK(hipMalloc(&pl, N * hw * 4)); CK(hipMalloc(&pab, N * hw * 8)); CK(hipMalloc(&pm, N * hw * 4)); CK(hipMalloc(&hb, 64 * 4));
        CK(hipMalloc(&w1, 8192)); CK(hipMalloc(&w2, 9 * 8192)); CK(hipMalloc(&o2, N * hw * 64 * 2));
        hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, 0, (unsigned short*)pl, N * hw * 2, 3u);      // (bit patterns: finite floats made of two bf16 halves)
        hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, 0, (unsigned short*)pab, N * hw * 4, 5u);
        hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, 0, (unsigned short*)pm, N * hw * 2, 9u);
        hipLaunchKernelGGL(fill_bf16, dim3(64), dim3(256), 0, 0, (unsigned short*)w1, 4096, 11u);
        hipLaunchKernelGGL(fill_bf16, dim3(64), dim3(256), 0, 0, (unsigned short*)w2, 9 * 4096, 13u);
        CK(hipMemset(hb, 0, 256));
        a.pk_L = pl; a.pk_ab = pab; a.pk_mask = pm; a.pk_ldiv = 1.f; a.pk_abdiv = 1.f; a.pk_mmul = 1.f; a.pk_mcent = 0.f;
        a.wgt = w1; a.wgt2 = w2; a.head_b = hb; a.out = o2; a.ncg = 1; a.nkc = 1; a.tiles_y = 0; a.act = 1;
    }
    if (v2 == 6) {      // fp32 Winograd F(2x2,3x3): U image = 16 floats per (cin, cout); timing only (random U)
        CK(hipFree(w)); CK(hipMalloc(&w, (size_t)C * C * 64));
        hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, 0, (unsigned short*)w, (size_t)C * C * 64 / 2, 7u);
        a.wgt = w; a.out_f32 = prec ? 0 : 1;
    }
    idc::ConvConfig cfg{wm, wp};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define LAUNCH() (v2 == 7 ? idc::launch_conv1_block(a, 0) : v2 == 6 ? idc::launch_conv_wino(prec, a, 0) : v2 == 4 ? idc::launch_conv_click(prec, wp, halo, a, 0) : v2 == 5 ? idc::launch_conv(prec, cfg, halo, a, 0) : v2 == 2 ? (dsm ? idc::launch_conv_ds_m(a, 0) : idc::launch_conv_ds(a, 0)) : v2 ? (ablv == 1 ? idc::launch_conv_v2p(cfg, halo, a, 0) : ablv == 2 ? idc::launch_conv_v2m(cfg, halo, a, 0) : idc::launch_conv_v2(cfg, halo, a, 0)) : idc::launch_conv(prec, cfg, halo, a, 0))
    a.warm = getenv("IDC_CODE_WARM") ? atoi(getenv("IDC_CODE_WARM")) : 1;
    const int dsm = getenv("IDC_DS_M16") ? atoi(getenv("IDC_DS_M16")) : 0;
    const int ablv = getenv("IDC_ABL_V2") ? atoi(getenv("IDC_ABL_V2")) : 0;      // 0: conv_igemm_v2 (32x32 MFMA), 1: conv_igemm_v2p, 2: conv_igemm_v2m
    if (v2 == 1 && ablv) { void* z; CK(hipMalloc(&z, 256)); CK(hipMemset(z, 0, 256)); a.zeros = z; }
    if (v2 == 1 && getenv("IDC_ABL_HEAD") && atoi(getenv("IDC_ABL_HEAD"))) {      // conv10_2's real epilogue: LeakyReLU + the 128 -> 2 head + tanh (C must be 128, cfg <2,2>)
        float *hw_, *hb_, *ho_; CK(hipMalloc(&hw_, 256 * 4)); CK(hipMalloc(&hb_, 8)); CK(hipMalloc(&ho_, (size_t)N * 2 * HW * HW * 4));
        std::vector<float> hh(256); for (int i = 0; i < 256; ++i) hh[i] = 0.01f * (float)((i * 37) % 19 - 9);
        CK(hipMemcpy(hw_, hh.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemset(hb_, 0, 8));
        a.head_w = hw_; a.head_b = hb_; a.head_out = ho_; a.head_mul = 110.f; a.act = 2;
    }
#ifdef IDC_TIMING
    int nb = v2 == 7 ? ((HW + 31) / 32) * ((HW + (getenv("IDC_C1_LW") && atoi(getenv("IDC_C1_LW")) == 0 ? 31 : 11)) / (getenv("IDC_C1_LW") && atoi(getenv("IDC_C1_LW")) == 0 ? 32 : 12)) * N : v2 == 6 ? (int)(((HW + halo - 1) / halo + 7) / 8) * (((HW + halo - 1) / halo + 7) / 8) * halo * halo * N * (C / 32) : v2 == 2 ? ((HW + 31) / 32) * ((HW + 3) / 4) * N * (a.ncg / 2) : a.tiles_x * a.tiles_y * N * (a.ncg / wm) * a.nphase * (a.ksplit > 1 ? a.ksplit : 1);
    long long* dbg; CK(hipMalloc(&dbg, (size_t)nb * 128)); CK(hipMemset(dbg, 0, (size_t)nb * 128));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(idc::g_idc_dbg), &dbg, sizeof(dbg)));
#endif
    for (int i = 0; i < 5; ++i) CK(LAUNCH());
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) CK(LAUNCH());
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    double flops = v2 == 3 ? 2.0 * N * 4 * HW * HW * (double)C * C * 4 : v2 == 2 ? 2.0 * N * 4 * HW * HW * (double)C * (4.0 * C + 9.0 * (argc > 10 ? atoi(argv[10]) : C)) : 2.0 * N * HW * HW * (double)C * C * ntaps;
#ifdef IDC_TIMING
    {
        CK(LAUNCH()); CK(hipDeviceSynchronize());
        std::vector<long long> h((size_t)nb * 16); CK(hipMemcpy(h.data(), dbg, (size_t)nb * 128, hipMemcpyDeviceToHost));
        long long t0 = h[0]; for (int b = 0; b < nb; ++b) if (h[b * 16] < t0) t0 = h[b * 16];
        double s[5] = {0, 0, 0, 0, 0}; long long tend = 0;
        for (int b = 0; b < nb; ++b) { for (int i = 0; i < 5; ++i) s[i] += (double)(h[b * 16 + i] - (i ? h[b * 16 + i - 1] : t0)); if (h[b * 16 + 4] > tend) tend = h[b * 16 + 4]; }
        if (v2 == 2) { double st[3] = {0, 0, 0}; for (int b = 0; b < nb; ++b) { st[0] += (double)(h[b * 16 + 8] - h[b * 16 + 1]); st[1] += (double)(h[b * 16 + 9] - h[b * 16 + 8]); st[2] += (double)(h[b * 16 + 2] - h[b * 16 + 9]); }
            printf("  conv_ds_fused main loop split (mean ticks): S part after its first barrier %.0f | hand-over to first D barrier %.0f | D part %.0f\n", st[0] / nb, st[1] / nb, st[2] / nb); }
#ifdef IDC_STEP_PROBE
        { double d[6] = {0, 0, 0, 0, 0, 0}; int cnt = 0; for (int b = 0; b < nb; ++b) { if (h[b * 16 + 15] == 0) continue; ++cnt; for (int i = 0; i < 6; ++i) d[i] += (double)(h[b * 16 + 10 + i] - h[b * 16 + 9 + i]); }
          if (cnt) printf("  one steady-state step of wave 0 (mean cycles over %d blocks): vmcnt+barrier %.0f | first reads + request issue %.0f | MFMA group 1 %.0f | group 2 %.0f | group 3 %.0f | tap-table update + group 4 %.0f | sum %.0f\n",
                          cnt, d[0] / cnt, d[1] / cnt, d[2] / cnt, d[3] / cnt, d[4] / cnt, d[5] / cnt, (d[0] + d[1] + d[2] + d[3] + d[4] + d[5]) / cnt); }
#endif
        if (v2 == 1 && ablv == 1) { double u[3] = {0, 0, 0}; for (int b = 0; b < nb; ++b) { const long long* q = &h[(size_t)b * 16]; u[0] += (double)(q[10] - q[9]); u[1] += (double)(q[11] - q[10]); u[2] += (double)(q[12] - q[11]); }
            { double e[4] = {0, 0, 0, 0}; for (int b = 0; b < nb; ++b) { const long long* q = &h[(size_t)b * 16]; e[0] += (double)(q[5] - q[2]); e[1] += (double)(q[6] - q[5]); e[2] += (double)(q[7] - q[6]); e[3] += (double)(q[3] - q[7]); }
              printf("  conv_igemm_v2p epilogue (mean ticks): barrier after the K loop %.0f | pixel row 0 (pack, LDS transpose, stores issued) %.0f | row 1 %.0f | rows 2-3 %.0f\n", e[0] / nb, e[1] / nb, e[2] / nb, e[3] / nb); }
            printf("  conv_igemm_v2p tap 4 of chunk 0 (mean ticks): wait for my LDS-DMA pieces %.0f | workgroup barrier %.0f | reads + DMA issue + 64 MFMAs + tap 5's wait and barrier %.0f\n", u[0] / nb, u[1] / nb, u[2] / nb); }
        if (v2 == 7) { double d[7] = {0, 0, 0, 0, 0, 0, 0}; for (int b = 0; b < nb; ++b) { const long long* q = &h[(size_t)b * 16];
              d[0] += (double)(q[1] - q[0]); d[1] += (double)(q[5] - q[1]); d[2] += (double)(q[2] - q[5]); d[3] += (double)(q[3] - q[2]); d[4] += (double)(q[6] - q[3]); d[5] += (double)(q[4] - q[6]); d[6] += (double)(q[7] - q[4]); }
            { double u[4] = {0, 0, 0, 0}; for (int b = 0; b < nb; ++b) { const long long* q = &h[(size_t)b * 16]; u[0] += (double)(q[10] - q[9]); u[1] += (double)(q[11] - q[10]); u[2] += (double)(q[12] - q[11]); }
              printf("  tap 4 of phase 2 (mean ticks): wait for my LDS-DMA pieces %.0f | workgroup barrier %.0f | DMA issue + reads + 24 MFMAs + tap 5's wait and barrier %.0f\n", u[0] / nb, u[1] / nb, u[2] / nb); }
            { int nz = 0, nz_first = 0; for (int b = 0; b < nb; ++b) { const unsigned r = (unsigned)h[(size_t)b * 16 + 8]; nz += (r & 0xfff) != 0; if (b < 512) nz_first += (r & 0xfff) != 0; }
              printf("  HW_REG_LDS_ALLOC: block 0 %#x, block 1 %#x, block 256 %#x, block 300 %#x; LDS_BASE != 0 in %d of %d tiles, %d of the first 512\n",
                     (unsigned)h[8], (unsigned)h[16 + 8], (unsigned)h[(size_t)256 * 16 + 8], (unsigned)h[(size_t)300 * 16 + 8], nz, nb, nz_first); }
            printf("  conv1 block per tile (mean ticks over %d tiles): phase 0 patch issue+write %.0f | wait patch (barrier) %.0f | phase 1 conv1_1 %.0f | phase 2 conv1_2 %.0f | barrier %.0f | phase 3 epilogue %.0f | store drain %.0f | sum %.0f\n",
                   nb, d[0] / nb, d[1] / nb, d[2] / nb, d[3] / nb, d[4] / nb, d[5] / nb, d[6] / nb, (d[0] + d[1] + d[2] + d[3] + d[4] + d[5] + d[6]) / nb); }
        printf("  timing (ticks, mean over %d blocks): start-offset %.0f | prologue %.0f | mainloop %.0f | epilogue-issue %.0f | store-drain %.0f | kernel span %lld\n",
               nb, s[0] / nb, s[1] / nb, s[2] / nb, s[3] / nb, s[4] / nb, tend - t0);
#ifdef IDC_TIMING_FINE
        { double u[3] = {0, 0, 0}; for (int b = 0; b < nb; ++b) { u[0] += (double)(h[b * 16 + 5] - h[b * 16]); u[1] += (double)(h[b * 16 + 6] - h[b * 16 + 5]); u[2] += (double)(h[b * 16 + 7] - h[b * 16 + 6]); }
          printf("  prologue split: setup(to first load issue) %.0f | issue halo+dma %.0f | wait loads + LDS write %.0f | to first tap barrier %.0f\n", u[0] / nb, u[1] / nb, u[2] / nb, s[1] / nb - (u[0] + u[1] + u[2]) / nb); }
#endif
        for (int b : {0, 1, 8, nb / 2, nb - 1}) printf("   block %5d: start %lld pro %lld main %lld epi %lld drain %lld\n", b, h[b * 16] - t0, h[b * 16 + 1] - h[b * 16], h[b * 16 + 2] - h[b * 16 + 1], h[b * 16 + 3] - h[b * 16 + 2], h[b * 16 + 4] - h[b * 16 + 3]);
    }
#endif
    printf("%-28s N=%d HW=%d C=%d halo=%d cfg=<%d,%d> prec=%d v2=%d ntaps=%d : %.4f ms  %.1f TFLOP/s\n", ABL_NAME, N, HW, C, halo, wm, wp, prec, v2, ntaps, ms, flops / ms / 1e9);
    return 0;
}
