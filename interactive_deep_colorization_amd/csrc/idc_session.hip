// Click-session kernels: the two pieces of per-click host work that surround the
// network in the reference GUI, moved next to the resident planes so that a click sends a few dozen bytes.
//
//  * hint rasterisation -- UIControl.get_input (ui/ui_control.py:177-187: filled rectangles on a black canvas,
//    later edits over earlier ones) + the rgb2lab of that canvas in gui_draw.compute_result (ui/gui_draw.py:273-277),
//    or the notebook's put_point (DemoInteractiveColorization.ipynb:131-139) when the colours are given as ab.
//  * colour suggestions -- ColorizeImageTorchDist.get_ab_reccs (data/colorize_image.py:322-354): inverse-CDF
//    samples of one pixel's predicted distribution, k-means, clusters ordered by occupancy.
//
// Both are latency kernels (one pixel per thread / one workgroup); neither touches HBM beyond the planes it writes.
#include <hip/hip_runtime.h>

#include "idc_kernels.h"

namespace idc {

// skimage rgb2lab of one uint8 colour, float64 (same constants / order as lab_post_kernel and oracle/colorspace.py)
__device__ __forceinline__ void hint_rgb8_to_lab(const unsigned char* q, double& L, double& a, double& b) {
    const double M[3][3] = {{0.412453, 0.357580, 0.180423}, {0.212671, 0.715160, 0.072169}, {0.019334, 0.119193, 0.950227}};
    const double white[3] = {0.95047, 1.0, 1.08883};
    double lin[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double v = (double)q[c] / 255.0;
        lin[c] = v > 0.04045 ? pow((v + 0.055) / 1.055, 2.4) : v / 12.92;
    }
    double g[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double t = (lin[0] * M[i][0] + lin[1] * M[i][1] + lin[2] * M[i][2]) / white[i];
        g[i] = t > 0.008856 ? cbrt(t) : 7.787 * t + 16.0 / 116.0;
    }
    L = 116.0 * g[1] - 16.0; a = 500.0 * (g[0] - g[1]); b = 200.0 * (g[1] - g[2]);
}

// One thread per pixel of one image: the LAST hint whose (inclusive, already clipped) rectangle covers the pixel wins,
// as successive cv2.rectangle / slice assignments do.  All lanes walk the same hint list (scalar loads).
__global__ __launch_bounds__(256) void raster_hints_kernel(const HintRect* __restrict__ hints, int n_hints, int mode,
                                                           float mask_value, float* __restrict__ ab,
                                                           float* __restrict__ mask, int H, int W) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, x = p - y * W;
    int hit = -1;
    for (int i = n_hints - 1; i >= 0; --i) {
        const HintRect r = hints[i];
        if (y >= r.y0 && y <= r.y1 && x >= r.x0 && x <= r.x1) { hit = i; break; }
    }
    float a = 0.f, b = 0.f, m = 0.f;
    if (hit >= 0) {
        const HintRect r = hints[hit];
        m = mask_value;
        if (mode == 0) {
            a = r.c0; b = r.c1;
        } else {
            const unsigned char q[3] = {(unsigned char)r.c0, (unsigned char)r.c1, (unsigned char)r.c2};
            double L, da, db;
            hint_rgb8_to_lab(q, L, da, db);
            a = (float)da; b = (float)db;
        }
    }
    ab[p] = a;
    ab[(size_t)H * W + p] = b;
    mask[p] = m;
}

hipError_t launch_raster_hints(const HintRect* hints, int n_hints, int mode, float mask_value, float* ab, float* mask,
                               int H, int W, hipStream_t s) {
    hipLaunchKernelGGL(raster_hints_kernel, dim3((H * W + 255) / 256), dim3(256), 0, s, hints, n_hints, mode, mask_value,
                       ab, mask, H, W);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Colour suggestions for one pixel.  pdf: B probabilities, element b at pdf[b * stride].
//   1. cmf = cumsum(pdf) (sequential fp32, as numpy's cumsum of the fp32 distribution), / cmf[B-1]
//   2. N draws u_i = lowbias32(seed, i) >> 8 scaled by 2^-24; bin = #{cmf <= u} (numpy.digitize); counts per bin
//   3. weighted k-means over the B bin centres with the counts as weights (== k-means over the N samples):
//      deterministic greedy k-means++ seeding (first centre = most-sampled bin, next = argmax count * D^2),
//      Lloyd until the assignment is stable (max 100 sweeps), ties -> lowest index, float64
//   4. clusters ordered by occupancy (descending, ties -> lower cluster index first); conf = occupancy / N
// One workgroup of 256 threads; B <= kSuggestMaxBins, K <= kSuggestMaxK.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned lowbias32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(256) void suggest_kernel(const float* __restrict__ pdf, long long stride, int B,
                                                      const float* __restrict__ centres, int K, int N, unsigned seed,
                                                      double* __restrict__ out_centres, double* __restrict__ out_conf,
                                                      unsigned* __restrict__ out_counts) {
#pragma clang fp contract(off)
    __shared__ float cmf[kSuggestMaxBins];
    __shared__ unsigned cnt[kSuggestMaxBins];
    __shared__ float cx[kSuggestMaxBins], cy[kSuggestMaxBins];
    __shared__ int asg[kSuggestMaxBins];
    __shared__ double mx[kSuggestMaxK], my[kSuggestMaxK];
    __shared__ unsigned long long occ[kSuggestMaxK];
    __shared__ int changed, pick;
    __shared__ double red_v[256];
    __shared__ int red_i[256];
    const int t = threadIdx.x;
    for (int b = t; b < B; b += 256) {
        cmf[b] = pdf[(long long)b * stride];
        cnt[b] = 0u;
        cx[b] = centres[2 * b]; cy[b] = centres[2 * b + 1];
        asg[b] = -1;
    }
    __syncthreads();
    if (t == 0) {
        float run = 0.f;
        for (int b = 0; b < B; ++b) { run += cmf[b]; cmf[b] = run; }
        const float tot = run;
        for (int b = 0; b < B; ++b) cmf[b] = cmf[b] / tot;
    }
    __syncthreads();
    for (int i = t; i < N; i += 256) {
        const unsigned h = lowbias32((unsigned)i * 0x9E3779B9u + seed * 0x85EBCA6Bu + 0x165667B1u);
        const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
        int lo = 0, hi = B;                       // first index with cmf > u  ==  #{cmf <= u}
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cmf[mid] <= u) lo = mid + 1; else hi = mid;
        }
        if (lo > B - 1) lo = B - 1;               // u < 1 = cmf[B-1] always; guard against a NaN pdf
        atomicAdd(&cnt[lo], 1u);
    }
    __syncthreads();
    if (out_counts) for (int b = t; b < B; b += 256) out_counts[b] = cnt[b];

    // --- seeding: k-th centre = argmax over bins of count * (squared distance to the nearest chosen centre) ---
    for (int k = 0; k < K; ++k) {
        double best = -1.0; int bi = 0x7fffffff;
        for (int b = t; b < B; b += 256) {
            double d2 = 1.0;
            if (k > 0) {
                d2 = 1.0e300;
                for (int j = 0; j < k; ++j) {
                    const double dx = (double)cx[b] - mx[j], dy = (double)cy[b] - my[j];
                    const double d = dx * dx + dy * dy;
                    if (d < d2) d2 = d;
                }
            }
            const double v = (double)cnt[b] * d2;
            if (v > best) { best = v; bi = b; }    // ascending b within a thread: first maximum kept
        }
        red_v[t] = best; red_i[t] = bi;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (t < s) {
                const double v2 = red_v[t + s]; const int i2 = red_i[t + s];
                if (v2 > red_v[t] || (v2 == red_v[t] && i2 < red_i[t])) { red_v[t] = v2; red_i[t] = i2; }
            }
            __syncthreads();
        }
        if (t == 0) { pick = red_i[0]; mx[k] = (double)cx[pick]; my[k] = (double)cy[pick]; }
        __syncthreads();
    }

    // --- Lloyd ---
    for (int it = 0; it < 100; ++it) {
        if (t == 0) changed = 0;
        __syncthreads();
        for (int b = t; b < B; b += 256) {
            int best = 0; double bd = 1.0e300;
            for (int j = 0; j < K; ++j) {
                const double dx = (double)cx[b] - mx[j], dy = (double)cy[b] - my[j];
                const double d = dx * dx + dy * dy;
                if (d < bd) { bd = d; best = j; }
            }
            if (best != asg[b]) { asg[b] = best; if (cnt[b]) changed = 1; }
        }
        __syncthreads();
        if (t < K) {                              // fixed-order (ascending bin) weighted mean of cluster t
            double sx = 0.0, sy = 0.0; unsigned long long w = 0ull;
            for (int b = 0; b < B; ++b)
                if (asg[b] == t && cnt[b]) { sx += (double)cnt[b] * (double)cx[b]; sy += (double)cnt[b] * (double)cy[b]; w += cnt[b]; }
            occ[t] = w;
            if (w) { mx[t] = sx / (double)w; my[t] = sy / (double)w; }     // an empty cluster keeps its centre
        }
        __syncthreads();
        if (!changed) break;
    }
    if (t == 0) {                                 // order by occupancy, descending, stable
        unsigned used = 0u;
        for (int r = 0; r < K; ++r) {
            int bk = -1;
            for (int k = 0; k < K; ++k)
                if (!((used >> k) & 1u) && (bk < 0 || occ[k] > occ[bk])) bk = k;
            used |= 1u << bk;
            out_centres[2 * r] = mx[bk]; out_centres[2 * r + 1] = my[bk];
            out_conf[r] = (double)occ[bk] / (double)N;
        }
    }
}

hipError_t launch_suggest(const float* pdf, long long stride, int B, const float* centres, int K, int N, unsigned seed,
                          double* out_centres, double* out_conf, unsigned* out_counts, hipStream_t s) {
    hipLaunchKernelGGL(suggest_kernel, dim3(1), dim3(256), 0, s, pdf, stride, B, centres, K, N, seed, out_centres, out_conf,
                       out_counts);
    return hipGetLastError();
}

}  // namespace idc
