// Validation metrics on the device (/root/reference/eval/metrics.py, used by train.py:129-137 once per validation frame):
//   _prepare_data (:20-25)   gt = gt > 128;  pred = pred / 255;  min-max normalise when max != min
//   MAE.cal_mae   (:100-102) mean |pred - gt|
//   Smeasure      (:120-213) object term (means / ddof-1 stds of pred over gt and of 1-pred over ~gt), region term
//                            (gt centroid split into 4 quadrants, an SSIM-like score per quadrant from first / second moments)
// Everything is a sum over pixels, so a frame costs three small reduction launches and 40 doubles of D2H instead of a
// full-resolution f32 map + numpy on the host.  WeightedFmeasure (:333-383) follows further down (exact Euclidean feature
// transform with scipy's nearest-index choice, 7x7 Gaussian, weighted sums).  acc layout (f64[40]): 0 n_gt, 1 sum p, 2 sum |p-g|, 3 sum_{g} p, 4 sum_{g} p^2, 5 sum_{~g} (1-p),
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


// ---- WeightedFmeasure.cal_wfm (eval/metrics.py:347-383) ----------------------------------------------------------
// bwdist(gt == 0, return_indices=True) is scipy's exact Euclidean feature transform (ni_morphology.c: one nearest-in-line
// pass along axis 0, then per row a lower-envelope build + monotone walk along axis 1).  Where a background pixel has
// several equidistant foreground pixels the chosen index decides Et, so both passes below repeat scipy's comparisons
// literally (ties -> the smaller row in pass 1; `<= 0 keeps`, `delta1 <= delta2 stays` in pass 2).  All quantities are
// small integers, so int64 replaces scipy's doubles exactly.  One thread owns one image line; the line buffers are laid
// out [x][y] so that the row pass (the long sequential one) reads and writes coalesced across its threads.

__global__ __launch_bounds__(64) void wfm_cols_kernel(const float* __restrict__ gt, int* __restrict__ fyT, int H, int W) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= W) return;
    int* col = fyT + (long)x * H;
    int last = -1;
    for (int y = 0; y < H; ++y) {
        if (gt[(long)y * W + x] > 128.f) last = y;
        col[y] = last;                                       // nearest foreground row at or above y
    }
    int next = -1;
    for (int y = H - 1; y >= 0; --y) {
        if (gt[(long)y * W + x] > 128.f) next = y;
        const int a = col[y];
        col[y] = a < 0 ? next : (next < 0 ? a : ((y - a) <= (next - y) ? a : next));
    }
}

__global__ __launch_bounds__(64) void wfm_rows_kernel(const int* __restrict__ fyT, int* __restrict__ gT,
                                                      int* __restrict__ idxT, int H, int W) {
    const int y = blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= H) return;
    auto FY = [&](int x) { return fyT[(long)x * H + y]; };
    auto G = [&](int l) -> int& { return gT[(long)l * H + y]; };
    int l = -1;
    for (int ii = 0; ii < W; ++ii) {
        const int fy = FY(ii);
        if (fy < 0) continue;
        const long wR = (long)(fy - y) * (fy - y);
        while (l >= 1) {
            const int i1 = G(l), i2 = G(l - 1);
            const long a = i1 - i2, b = ii - i1, c = a + b;
            const long f1 = FY(i1) - y, f2 = FY(i2) - y;
            if (c * (f1 * f1) - b * (f2 * f2) - a * wR - a * b * c <= 0) break;
            --l;
        }
        ++l;
        G(l) = ii;
    }
    const int maxl = l;
    if (maxl < 0) {                                          // no foreground anywhere (the host returns 0 before using this)
        for (int ii = 0; ii < W; ++ii) idxT[(long)ii * H + y] = -1;
        return;
    }
    l = 0;
    int gx = G(0), gy = FY(gx);
    for (int ii = 0; ii < W; ++ii) {
        long d1 = (long)(gx - ii) * (gx - ii) + (long)(gy - y) * (gy - y);
        while (l < maxl) {
            const int nx = G(l + 1), ny = FY(nx);
            const long d2 = (long)(nx - ii) * (nx - ii) + (long)(ny - y) * (ny - y);
            if (d1 <= d2) break;
            d1 = d2;
            ++l;
            gx = nx;
            gy = ny;
        }
        idxT[(long)ii * H + y] = gy * W + gx;               // linear index of the nearest foreground pixel
    }
}

__global__ void wfm_init_kernel(double* __restrict__ out, int* __restrict__ mm) {
    if (threadIdx.x < 4) out[threadIdx.x] = 0.0;
    if (threadIdx.x == 0) {
        mm[0] = f2ord(INFINITY);
        mm[1] = f2ord(-INFINITY);
    }
}

__global__ __launch_bounds__(256) void wfm_unpack_kernel(const int* __restrict__ idxT, int* __restrict__ idx, int H, int W) {
    const long n = (long)H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)(i / W);
        const int j = idxT[(long)x * H + y];
        idx[i] = j < 0 ? -1 : j / W;
        idx[n + i] = j < 0 ? -1 : j % W;
    }
}

// E = |pred - gt| (float32 like the reference's arrays); Et = E with every background pixel taking E of its nearest
// foreground pixel; d2 = squared distance to it (0 on the foreground)
__global__ __launch_bounds__(256) void wfm_et_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                     const int* __restrict__ ws, const int* __restrict__ idxT,
                                                     float* __restrict__ Et, int* __restrict__ d2, int H, int W) {
    const float mn = ord2f(ws[0]), mx = ord2f(ws[1]);
    const long n = (long)H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)(i / W);
        const bool g = gt[i] > 128.f;
        const long j = g ? i : (long)idxT[(long)x * H + y];   // nearest foreground pixel (-1: the frame has none)
        int dd = 0;
        float et;
        if (g || j >= 0) {
            et = fabsf(__fsub_rn(prepared(pred[j], mn, mx), 1.f));
            if (!g) {
                const int fy = (int)(j / W), fx = (int)(j % W);
                dd = (fy - y) * (fy - y) + (fx - x) * (fx - x);
            }
        } else {
            et = fabsf(prepared(pred[i], mn, mx));
        }
        Et[i] = et;
        d2[i] = dd;
    }
}

// EA = 7x7 Gaussian of Et (zero outside the frame, accumulated in double, stored as float32 like scipy's output array);
// MIN_E_EA, the distance weighting B, Ew and the three sums.  kc: 49 kernel taps + log(0.5)/5.
// out: 0 n_gt, 1 sum of Ew over gt, 2 sum of Ew over ~gt
__global__ __launch_bounds__(256) void wfm_sum_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                      const int* __restrict__ ws, const float* __restrict__ Et,
                                                      const int* __restrict__ d2, const double* __restrict__ kc,
                                                      double* __restrict__ out, int H, int W) {
    __shared__ double ks[50];
    if (threadIdx.x < 50) ks[threadIdx.x] = kc[threadIdx.x];
    __syncthreads();
    const float mn = ord2f(ws[0]), mx = ord2f(ws[1]);
    double a[3] = {0, 0, 0};
    const long n = (long)H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)(i / W);
        const bool g = gt[i] > 128.f;
        const float p = prepared(pred[i], mn, mx);
        const float e = g ? fabsf(__fsub_rn(p, 1.f)) : fabsf(p);
        if (g) {
            double acc = 0.0;
            for (int dy = -3; dy <= 3; ++dy) {
                const int yy = y + dy;
                if (yy < 0 || yy >= H) continue;
                for (int dx = -3; dx <= 3; ++dx) {
                    const int xx = x + dx;
                    if (xx < 0 || xx >= W) continue;
                    acc += ks[(dy + 3) * 7 + dx + 3] * (double)Et[(long)yy * W + xx];
                }
            }
            const float ea = (float)acc;
            a[0] += 1.0;
            a[1] += (double)(ea < e ? ea : e);                // B = 1 on the foreground
        } else {
            const double b = 2.0 - exp(ks[49] * sqrt((double)d2[i]));
            a[2] += (double)e * b;                            // EA only replaces E where gt is set
        }
    }
    flush<3>(a, out, 0);
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

// WeightedFmeasure.cal_wfm (eval/metrics.py:347-383) for one frame.  pred / gt as for emip_eval_frame; kc f64 [50] on the
// device: the 49 taps of matlab_style_gauss2D((7,7), 5) followed by log(0.5)/5, both computed by the host in float64
// exactly as the reference does.  out f64 [4]: n_gt, sum Ew[gt], sum Ew[~gt], unused.  ws: 256 + 16*H*W bytes of
// scratch (16-byte aligned).  When out[0] == 0 the other sums are meaningless (the reference
// returns 0 for an empty gt before calling cal_wfm).
extern "C" int emip_eval_wfm(const float* pred, const float* gt, const double* kc, double* out, void* ws, int H, int W,
                             void* stream) {
    EMIP_REQUIRE(pred && gt && kc && out && ws && H > 0 && W > 0 && (long)H * W < (1L << 30));
    EMIP_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15) == 0);
    hipStream_t s = (hipStream_t)stream;
    const long n = (long)H * W;
    int* mm = reinterpret_cast<int*>(ws);
    int* fyT = reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + 256);
    int* gT = fyT + n;
    int* idxT = gT + n;
    float* Et = reinterpret_cast<float*>(idxT + n);
    int* d2 = gT;                                            // the row pass is done with its stack by then
    hipLaunchKernelGGL(wfm_init_kernel, dim3(1), dim3(64), 0, s, out, mm);
    hipLaunchKernelGGL(eval_minmax_kernel, dim3(grid_for(n, 256 * 4)), dim3(256), 0, s, pred, mm, n);
    hipLaunchKernelGGL(wfm_cols_kernel, dim3((W + 63) / 64), dim3(64), 0, s, gt, fyT, H, W);
    hipLaunchKernelGGL(wfm_rows_kernel, dim3((H + 63) / 64), dim3(64), 0, s, fyT, gT, idxT, H, W);
    hipLaunchKernelGGL(wfm_et_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, pred, gt, mm, idxT, Et, d2, H, W);
    hipLaunchKernelGGL(wfm_sum_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, pred, gt, mm, Et, d2, kc, out, H, W);
    return emip_launch_status();
}

// the feature transform alone (tests): idx int [2][H][W] = scipy's distance_transform_edt(gt == 0, return_indices=True)[1]
extern "C" int emip_eval_edt_indices(const float* gt, int* idx, void* ws, int H, int W, void* stream) {
    EMIP_REQUIRE(gt && idx && ws && H > 0 && W > 0 && (long)H * W < (1L << 30));
    hipStream_t s = (hipStream_t)stream;
    const long n = (long)H * W;
    int* fyT = reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + 256);
    int* gT = fyT + n;
    int* idxT = gT + n;
    hipLaunchKernelGGL(wfm_cols_kernel, dim3((W + 63) / 64), dim3(64), 0, s, gt, fyT, H, W);
    hipLaunchKernelGGL(wfm_rows_kernel, dim3((H + 63) / 64), dim3(64), 0, s, fyT, gT, idxT, H, W);
    hipLaunchKernelGGL(wfm_unpack_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, idxT, idx, H, W);
    return emip_launch_status();
}
