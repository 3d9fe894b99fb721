// Validation metrics on the device (/root/reference/eval/metrics.py, used by train.py:129-137 once per validation frame):
//   _prepare_data (:20-25)   gt = gt > 128;  pred = pred / 255;  min-max normalise when max != min
//   MAE.cal_mae   (:100-102) mean |pred - gt|
//   Smeasure      (:120-213) object term (means / ddof-1 stds of pred over gt and of 1-pred over ~gt), region term
//                            (gt centroid split into 4 quadrants, an SSIM-like score per quadrant from first / second moments)
// Everything is a sum over pixels, so a frame costs three small reduction launches and 40 doubles of D2H instead of a
// full-resolution f32 map + numpy on the host.  (WeightedFmeasure needs an exact Euclidean distance transform and stays on
// the host.)  acc layout (f64[40]): 0 n_gt, 1 sum p, 2 sum |p-g|, 3 sum_{g} p, 4 sum_{g} p^2, 5 sum_{~g} (1-p),
// 6 sum_{~g} (1-p)^2, 7 sum g*col, 8 sum g*row, 9 unused; 10+6q.. per quadrant q (LT, RT, LB, RB): N, sum p, sum g,
// sum p^2, sum g^2, sum p*g;  34 pmin, 35 pmax of the incoming map.
#include "common.h"

namespace {

inline int grid_for(long n, int threads) {
    long b = (n + threads - 1) / threads;
    if (b > 1024) b = 1024;
    return (int)(b < 1 ? 1 : b);
}

__device__ __forceinline__ int f2ord(float f) {
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

__global__ void eval_init_kernel(double* __restrict__ acc, int* __restrict__ ws) {
    if (threadIdx.x < 40) acc[threadIdx.x] = 0.0;
    if (threadIdx.x == 0) {
        ws[0] = f2ord(INFINITY);
        ws[1] = f2ord(-INFINITY);
    }
}

__global__ __launch_bounds__(256) void eval_minmax_kernel(const float* __restrict__ pred, int* __restrict__ ws, long n) {
    float mn = INFINITY, mx = -INFINITY;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = __fdiv_rn(pred[i], 255.f);          // _prepare_data: pred / 255 (float32)
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o));
        mx = fmaxf(mx, __shfl_xor(mx, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(ws, f2ord(mn));
        atomicMax(ws + 1, f2ord(mx));
    }
}

__device__ __forceinline__ float prepared(float raw, float mn, float mx) {
    const float v = __fdiv_rn(raw, 255.f);
    return mx != mn ? __fdiv_rn(__fsub_rn(v, mn), __fsub_rn(mx, mn)) : v;
}

template <int NA>
__device__ __forceinline__ void flush(double (&a)[NA], double* __restrict__ acc, int base) {
    __shared__ double sh[4][NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        double v = a[j];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][j] = v;
    }
    __syncthreads();
    if (threadIdx.x < NA) atomicAdd(acc + base + threadIdx.x, sh[0][threadIdx.x] + sh[1][threadIdx.x] +
                                                                 sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// gt: float [H][W] with the reference's 0..255 convention (gt > 128 is foreground)
__global__ __launch_bounds__(256) void eval_pass1_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                         const int* __restrict__ ws, double* __restrict__ acc, int H,
                                                         int W) {
    const float mn = ord2f(ws[0]), mx = ord2f(ws[1]);
    double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const long n = (long)H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float p = prepared(pred[i], mn, mx);
        const bool g = gt[i] > 128.f;
        const double pd = (double)p;
        a[1] += pd;
        a[2] += fabs(pd - (g ? 1.0 : 0.0));
        if (g) {
            a[0] += 1.0;
            a[3] += pd;
            a[4] += pd * pd;
            a[7] += (double)(i % W);
            a[8] += (double)(i / W);
        } else {
            const double q = (double)(1.0f - p);           // (1 - pred) in float32 like the reference's array arithmetic
            a[5] += q;
            a[6] += q * q;
        }
    }
    flush<9>(a, acc, 0);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        acc[34] = (double)mn;
        acc[35] = (double)mx;
    }
}

__global__ __launch_bounds__(256) void eval_pass2_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                         const int* __restrict__ ws, double* __restrict__ acc, int H,
                                                         int W) {
    const float mn = ord2f(ws[0]), mx = ord2f(ws[1]);
    // Smeasure.centroid (:157-168): rounded centre of mass of gt (+1), image centre when gt is empty
    const double area = acc[0];
    int cx, cy;
    if (area == 0.0) {
        cx = (int)rint((double)W / 2.0) + 1;
        cy = (int)rint((double)H / 2.0) + 1;
    } else {
        cx = (int)rint(acc[7] / area) + 1;
        cy = (int)rint(acc[8] / area) + 1;
    }
    double a[24];
#pragma unroll
    for (int j = 0; j < 24; ++j) a[j] = 0.0;
    const long n = (long)H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)(i / W);
        const double p = (double)prepared(pred[i], mn, mx);
        const double g = gt[i] > 128.f ? 1.0 : 0.0;
        const int q = (y < cy ? 0 : 2) + (x < cx ? 0 : 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {                          // branch-free scatter into the quadrant's accumulators
            const double s = (k == q) ? 1.0 : 0.0;
            a[6 * k] += s;
            a[6 * k + 1] += s * p;
            a[6 * k + 2] += s * g;
            a[6 * k + 3] += s * p * p;
            a[6 * k + 4] += s * g * g;
            a[6 * k + 5] += s * p * g;
        }
    }
    flush<24>(a, acc, 10);
}

}  // namespace

// pred: f32 [H][W] the map the reference hands to `step(pred=res, ...)` (train.py:125-131); gt: f32 [H][W], 0..255.
// acc: f64 [40] (zeroed inside), ws: int [2].  Host-side finalisation: emip_amd/eval_metrics.py.
extern "C" int emip_eval_frame(const float* pred, const float* gt, double* acc, int* ws, int H, int W, void* stream) {
    EMIP_REQUIRE(pred && gt && acc && ws && H > 0 && W > 0);
    hipStream_t s = (hipStream_t)stream;
    const long n = (long)H * W;
    hipLaunchKernelGGL(eval_init_kernel, dim3(1), dim3(64), 0, s, acc, ws);
    hipLaunchKernelGGL(eval_minmax_kernel, dim3(grid_for(n, 256 * 4)), dim3(256), 0, s, pred, ws, n);
    hipLaunchKernelGGL(eval_pass1_kernel, dim3(grid_for(n, 256 * 4)), dim3(256), 0, s, pred, gt, ws, acc, H, W);
    hipLaunchKernelGGL(eval_pass2_kernel, dim3(grid_for(n, 256 * 4)), dim3(256), 0, s, pred, gt, ws, acc, H, W);
    return emip_launch_status();
}
