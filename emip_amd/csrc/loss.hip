// Flow-side loss kernels (planar f32 tensors, as the reference's loss code sees them):
//   bilinear flow warp with border padding  (/root/reference/loss/warp_utils.py:16-23,83-93)
//   backward-occlusion map: corner indices, bilinear weights, scatter-add, threshold
//                                           (/root/reference/loss/warp_utils.py:26-80,106-112)
// HBM-bound gathers/scatters.  The corner index arithmetic is evaluated in f32 and
// converted to int64 exactly as the reference writes it (x + y*W with W = 352 stays
// below 2^24, so every intermediate is an exactly representable integer).
#include "common.h"

namespace {

inline int grid_for(long n, int threads) {
    long b = (n + threads - 1) / threads;
    if (b > 4096) b = 4096;
    return (int)(b < 1 ? 1 : b);
}

__global__ __launch_bounds__(256) void flow_warp_kernel(const float* __restrict__ X, const float* __restrict__ F,
                                                        float* __restrict__ Y, int B, int C, int H, int W) {
    const long hw = (long)H * W;
    const long total = (long)B * hw;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % W), y = (int)((idx / W) % H);
        const long b = idx / hw;
        const float vx = (float)x + F[(b * 2) * hw + (long)y * W + x];
        const float vy = (float)y + F[(b * 2 + 1) * hw + (long)y * W + x];
        // norm_grid, then ATen's grid_sampler unnormalize (align_corners) + border clip
        const float gx = 2.0f * vx / (float)(W - 1) - 1.0f;
        const float gy = 2.0f * vy / (float)(H - 1) - 1.0f;
        float ix = ((gx + 1.f) / 2.f) * (float)(W - 1);
        float iy = ((gy + 1.f) / 2.f) * (float)(H - 1);
        ix = fminf((float)(W - 1), fmaxf(ix, 0.f));
        iy = fminf((float)(H - 1), fmaxf(iy, 0.f));
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
        const float wx1 = ix - fx0, wy1 = iy - fy0, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        const float nw = wx0 * wy0, ne = wx1 * wy0, sw = wx0 * wy1, se = wx1 * wy1;
        const bool xin = x1 < W, yin = y1 < H;
        for (int c = 0; c < C; ++c) {
            const float* p = X + (b * C + c) * hw;
            float o = p[(long)y0 * W + x0] * nw;
            if (xin) o += p[(long)y0 * W + x1] * ne;
            if (yin) o += p[(long)y1 * W + x0] * sw;
            if (xin && yin) o += p[(long)y1 * W + x1] * se;
            Y[(b * C + c) * hw + (long)y * W + x] = o;
        }
    }
}

// indices [B][4N] int64 and weights [B][4N] f32 (either may be null); cmap [B][N] f32 (may be null,
// must be zero-filled by the caller before the launch): scatter-add of the weights.
__global__ __launch_bounds__(256) void occ_corner_kernel(const float* __restrict__ F, long long* __restrict__ indices,
                                                         float* __restrict__ weights, float* __restrict__ cmap, int B,
                                                         int H, int W) {
    const long n = (long)H * W;
    const long total = (long)B * n;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long pix = idx % n;
        const long b = idx / n;
        const int px = (int)(pix % W), py = (int)(pix / W);
        const float x = (float)px + F[(b * 2) * n + pix];
        const float y = (float)py + F[(b * 2 + 1) * n + pix];
        const float x1 = floorf(x), y1 = floorf(y);
        const float wm = (float)(W - 1), hm = (float)(H - 1);
        const float xf = fminf(fmaxf(x1, 0.f), wm), yf = fminf(fmaxf(y1, 0.f), hm);
        const float x0 = x1 + 1.f, y0 = y1 + 1.f;
        const float xc = fminf(fmaxf(x0, 0.f), wm), yc = fminf(fmaxf(y0, 0.f), hm);
        const bool xco = x0 != xc, yco = y0 != yc, xfo = x1 != xf, yfo = y1 != yf;
        const float fw = (float)W;
        const float id[4] = {xc + yc * fw, xc + yf * fw, xf + yc * fw, xf + yf * fw};
        const float wxc = 1.f - fabsf(x - xc), wxf = 1.f - fabsf(x - xf);
        const float wyc = 1.f - fabsf(y - yc), wyf = 1.f - fabsf(y - yf);
        float wv[4] = {wxc * wyc, wxc * wyf, wxf * wyc, wxf * wyf};
        const bool inv[4] = {xco || yco, xco || yfo, xfo || yco, xfo || yfo};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (inv[k]) wv[k] = 0.f;
            const long long ii = (long long)id[k];
            if (indices) indices[(b * 4 + k) * n + pix] = ii;
            if (weights) weights[(b * 4 + k) * n + pix] = wv[k];
            if (cmap) atomicAdd(cmap + b * n + ii, wv[k]);
        }
    }
}

__global__ __launch_bounds__(256) void occ_threshold_kernel(const float* __restrict__ cmap, float* __restrict__ occ,
                                                            long total, float th) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const float v = fminf(fmaxf(cmap[idx], 0.f), 1.f);
        occ[idx] = v < th ? 1.f : 0.f;
    }
}

}  // namespace

extern "C" int emip_flow_warp(const float* X, const float* flow, float* Y, int B, int C, int H, int W, void* stream) {
    EMIP_REQUIRE(X && flow && Y && B > 0 && C > 0 && H > 1 && W > 1);
    hipLaunchKernelGGL(flow_warp_kernel, dim3(grid_for((long)B * H * W, 256)), dim3(256), 0, (hipStream_t)stream, X,
                       flow, Y, B, C, H, W);
    return emip_launch_status();
}

extern "C" int emip_occ_corners(const float* flow, long long* indices, float* weights, int B, int H, int W,
                                void* stream) {
    EMIP_REQUIRE(flow && (indices || weights) && B > 0 && H > 1 && W > 1 && (long)H * W < (1L << 24));
    hipLaunchKernelGGL(occ_corner_kernel, dim3(grid_for((long)B * H * W, 256)), dim3(256), 0, (hipStream_t)stream,
                       flow, indices, weights, (float*)nullptr, B, H, W);
    return emip_launch_status();
}

// occ = (clamp(scatter_add(weights at indices), 0, 1) < th); cmap_ws: f32 [B][H*W] workspace
extern "C" int emip_occ_mask_backward(const float* flow, float* cmap_ws, float* occ, int B, int H, int W, float th,
                                      void* stream) {
    EMIP_REQUIRE(flow && cmap_ws && occ && B > 0 && H > 1 && W > 1 && (long)H * W < (1L << 24));
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * H * W;
    if (hipMemsetAsync(cmap_ws, 0, sizeof(float) * total, s) != hipSuccess) return EMIP_E_LAUNCH;
    hipLaunchKernelGGL(occ_corner_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, flow, (long long*)nullptr,
                       (float*)nullptr, cmap_ws, B, H, W);
    hipLaunchKernelGGL(occ_threshold_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, cmap_ws, occ, total, th);
    return emip_launch_status();
}
