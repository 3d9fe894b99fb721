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
    const IdxDiv D(total);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long row = D.div(idx, W), b = D.div(row, H);
        const int x = (int)(idx - row * W), y = (int)(row - b * H);
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
            if (cmap && wv[k] != 0.f) atomicAdd(cmap + b * n + ii, wv[k]);   // clamped (out-of-frame) corners carry weight 0:
                                                                              // skipping them avoids piling atomics on the border
        }
    }
}

// The occlusion map with the scatter INSIDE the CU: a workgroup owns a band of TH target rows of one map as f32 in LDS,
// scans every source pixel of that map (the flow is 1 MB: it stays in L2), adds the corner weights that land in its band with
// LDS atomics and writes the thresholded band.  The global scatter above is bound by L2 atomic throughput on a field that
// scatters to unrelated cells (32 M atomics at 352 x 352 x 64 maps: 0.89 ms per launch, and every untrained GMFlow produces
// such a field); here the same adds run at LDS rate, the clear / scatter / threshold passes and the workspace are gone.
__global__ __launch_bounds__(512) void occ_band_kernel(const float* __restrict__ F, float* __restrict__ occ, int H, int W, int TH,
                                                       float th, int complement) {
    extern __shared__ float band[];                    // [TH][W]
    const int b = blockIdx.y;
    const int r0 = blockIdx.x * TH, r1 = min(H, r0 + TH);
    const int n = H * W, nb = (r1 - r0) * W;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) band[i] = 0.f;
    __syncthreads();
    const float* fx = F + ((long)b * 2) * n;
    const float* fy = fx + n;
    const float wm = (float)(W - 1), hm = (float)(H - 1);
    // four consecutive pixels per thread and sweep (16-byte loads of both components: the scan is bound by load latency);
    // n is a multiple of 4 (host check)
    for (int p4 = threadIdx.x * 4; p4 < n; p4 += blockDim.x * 4) {
        const float4 vx = *reinterpret_cast<const float4*>(fx + p4), vy = *reinterpret_cast<const float4*>(fy + p4);
        const float ax[4] = {vx.x, vx.y, vx.z, vx.w}, ay[4] = {vy.x, vy.y, vy.z, vy.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pix = p4 + j;
            const int py = pix / W, px = pix - py * W;
            const float y = (float)py + ay[j];
            const float y1 = floorf(y);
            // rows this pixel can touch: y1 and y1 + 1 after clamping -- skip when neither is in the band
            const float yf = fminf(fmaxf(y1, 0.f), hm), y0 = y1 + 1.f, yc = fminf(fmaxf(y0, 0.f), hm);
            const int iyf = (int)yf, iyc = (int)yc;
            const bool inf = iyf >= r0 && iyf < r1, inc = iyc >= r0 && iyc < r1;
            if (!inf && !inc) continue;
            const float x = (float)px + ax[j];
            const float x1 = floorf(x);
            const float xf = fminf(fmaxf(x1, 0.f), wm), x0 = x1 + 1.f, xc = fminf(fmaxf(x0, 0.f), wm);
            const bool xco = x0 != xc, yco = y0 != yc, xfo = x1 != xf, yfo = y1 != yf;
            const float wxc = 1.f - fabsf(x - xc), wxf = 1.f - fabsf(x - xf);
            const float wyc = 1.f - fabsf(y - yc), wyf = 1.f - fabsf(y - yf);
            const int ixc = (int)xc, ixf = (int)xf;
            // the four corners of occ_corner_kernel; clamped (out-of-frame) corners carry weight 0
            if (inc) {
                if (!(xco || yco)) { const float v = wxc * wyc; if (v != 0.f) atomicAdd(&band[(iyc - r0) * W + ixc], v); }
                if (!(xfo || yco)) { const float v = wxf * wyc; if (v != 0.f) atomicAdd(&band[(iyc - r0) * W + ixf], v); }
            }
            if (inf) {
                if (!(xco || yfo)) { const float v = wxc * wyf; if (v != 0.f) atomicAdd(&band[(iyf - r0) * W + ixc], v); }
                if (!(xfo || yfo)) { const float v = wxf * wyf; if (v != 0.f) atomicAdd(&band[(iyf - r0) * W + ixf], v); }
            }
        }
    }
    __syncthreads();
    float* o = occ + (long)b * n + (long)r0 * W;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) {
        const float v = fminf(fmaxf(band[i], 0.f), 1.f);
        const float m = v < th ? 1.f : 0.f;
        o[i] = complement ? 1.f - m : m;
    }
}

__global__ __launch_bounds__(256) void occ_threshold_kernel(const float* __restrict__ cmap, float* __restrict__ occ,
                                                            long total, float th, int complement) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const float v = fminf(fmaxf(cmap[idx], 0.f), 1.f);
        const float o = v < th ? 1.f : 0.f;
        occ[idx] = complement ? 1.f - o : o;
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
                                      int complement, void* stream) {
    EMIP_REQUIRE(flow && occ && B > 0 && H > 1 && W > 1 && (long)H * W < (1L << 24));
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * H * W;
    // bands of 32 rows (45 KB of LDS at W = 352: three workgroups per CU keep enough loads in flight; four bands of 88 rows,
    // one workgroup per CU, took 487 us against 305)
    const int TH = 32;
    if ((size_t)TH * W * sizeof(float) <= 64 * 1024 && B < 65536 && ((long)H * W) % 4 == 0 && aligned16(flow)) {
        hipLaunchKernelGGL(occ_band_kernel, dim3((H + TH - 1) / TH, B), dim3(512), (size_t)TH * W * sizeof(float), s, flow, occ,
                           H, W, TH, th, complement);
        return emip_launch_status();
    }
    EMIP_REQUIRE(cmap_ws);
    if (emip_zero_async(cmap_ws, sizeof(float) * total, s) != EMIP_OK) return EMIP_E_LAUNCH;
    hipLaunchKernelGGL(occ_corner_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, flow, (long long*)nullptr,
                       (float*)nullptr, cmap_ws, B, H, W);
    hipLaunchKernelGGL(occ_threshold_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, cmap_ws, occ, total, th, complement);
    return emip_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Loss reductions (forward).  Planar f32 inputs; partial sums are combined with f64 atomics.
//   hybrid_e_loss   /root/reference/loss/loss_pred.py:4-22
//   loss_photomatric + SSIM   /root/reference/loss/loss_flow.py:35-49, loss/loss_blocks.py:46-65
namespace {

__device__ __forceinline__ double block_sum(double v, double* sh) {
    // 256 threads -> one value in thread 0
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    double r = 0;
    if (threadIdx.x == 0) r = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return r;
}

// acc[b][0..3] += sum sigmoid(p), sum m, sum sigmoid(p)*m, sum bce(p, m)      (per image b)
__global__ __launch_bounds__(256) void hybrid_pass1_kernel(const float* __restrict__ P, const float* __restrict__ M,
                                                           double* __restrict__ acc, int HW) {
    __shared__ double sh[4];
    const int b = blockIdx.y;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        const float x = P[(long)b * HW + i], z = M[(long)b * HW + i];
        const float sg = 1.f / (1.f + expf(-x));
        s0 += sg; s1 += z; s2 += sg * z;
        s3 += fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x)));
    }
    const double r0 = block_sum(s0, sh), r1 = block_sum(s1, sh), r2 = block_sum(s2, sh), r3 = block_sum(s3, sh);
    if (threadIdx.x == 0) {
        atomicAdd(acc + b * 8 + 0, r0); atomicAdd(acc + b * 8 + 1, r1);
        atomicAdd(acc + b * 8 + 2, r2); atomicAdd(acc + b * 8 + 3, r3);
    }
}

// acc[b][4] += sum (1+EFM)^2/4 with the per-image means from pass 1
__global__ __launch_bounds__(256) void hybrid_pass2_kernel(const float* __restrict__ P, const float* __restrict__ M,
                                                           double* __restrict__ acc, int HW) {
    __shared__ double sh[4];
    const int b = blockIdx.y;
    const float mp = (float)(acc[b * 8 + 0] / HW), mm = (float)(acc[b * 8 + 1] / HW);
    double s = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        const float sg = 1.f / (1.f + expf(-P[(long)b * HW + i]));
        const float f = sg - mp, g = M[(long)b * HW + i] - mm;
        const float e = (2.0f * f * g + 1e-8f) / (f * f + g * g + 1e-8f);
        s += (1.f + e) * (1.f + e) / 4.0f;
    }
    const double r = block_sum(s, sh);
    if (threadIdx.x == 0) atomicAdd(acc + b * 8 + 4, r);
}

__global__ void hybrid_final_kernel(const double* __restrict__ acc, float* __restrict__ out, int B, int HW) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double bce = 0;
    for (int b = 0; b < B; ++b) bce += acc[b * 8 + 3];
    bce /= (double)B * HW;
    double tot = 0;
    for (int b = 0; b < B; ++b) {
        const double eloss = 1.0 - acc[b * 8 + 4] / HW;
        const double inter = acc[b * 8 + 2], uni = acc[b * 8 + 0] + acc[b * 8 + 1];
        const double wiou = 1.0 - (inter + 1 + 1e-8) / (uni - inter + 1 + 1e-8);
        tot += bce + eloss + wiou;
    }
    out[0] = (float)(tot / B);
}

// sums[0] += sum_{b,c,y,x} |im - rec| * m ; sums[1] += sum over the (H-2)x(W-2) interior of ssim_dist(rec*m, im*m);
// sums[2] += sum m  (counted once per pixel)
__global__ __launch_bounds__(256) void photometric_kernel(const float* __restrict__ IM, const float* __restrict__ REC,
                                                          const float* __restrict__ MK, double* __restrict__ sums,
                                                          int B, int C, int H, int W) {
    __shared__ double sh[4];
    const long hw = (long)H * W;
    const long total = (long)B * C * hw;
    double l1 = 0, ss = 0, ms = 0;
    const IdxDiv D(total);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long row = D.div(idx, W), bc = D.div(row, H), b = D.div(bc, C);
        const int x = (int)(idx - row * W), y = (int)(row - bc * H);
        const int c = (int)(bc - b * C);
        const float* im = IM + (b * C + c) * hw;
        const float* rc = REC + (b * C + c) * hw;
        const float* mk = MK + b * hw;
        const float m0 = mk[(long)y * W + x];
        l1 += fabsf(im[(long)y * W + x] - rc[(long)y * W + x]) * m0;
        if (c == 0) ms += m0;
        if (y >= 1 && y < H - 1 && x >= 1 && x < W - 1) {
            float sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const long o = (long)(y + dy) * W + x + dx;
                    const float mm = mk[o];
                    const float a = rc[o] * mm, bb = im[o] * mm;     // SSIM(x = rec*m, y = im*m)
                    sx += a; sy += bb; sxx += a * a; syy += bb * bb; sxy += a * bb;
                }
            const float mux = sx / 9.f, muy = sy / 9.f;
            const float sgx = sxx / 9.f - mux * mux, sgy = syy / 9.f - muy * muy, sgxy = sxy / 9.f - mux * muy;
            const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
            const float n = (2 * mux * muy + C1) * (2 * sgxy + C2);
            const float d = (mux * mux + muy * muy + C1) * (sgx + sgy + C2);
            ss += fminf(fmaxf((1.f - n / d) / 2.f, 0.f), 1.f);
        }
    }
    const double r0 = block_sum(l1, sh), r1 = block_sum(ss, sh), r2 = block_sum(ms, sh);
    if (threadIdx.x == 0) {
        atomicAdd(sums + 0, r0); atomicAdd(sums + 1, r1); atomicAdd(sums + 2, r2);
    }
}

// out[0] += weight * (0.15*L1mean + 0.85*SSIMmean) / maskmean  from the three sums
__global__ void photometric_final_kernel(const double* __restrict__ sums, float* __restrict__ out, double n_l1,
                                         double n_ssim, double n_mask, float weight, int accumulate) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double v = (0.15 * sums[0] / n_l1 + 0.85 * sums[1] / n_ssim) / (sums[2] / n_mask);
    out[0] = (accumulate ? out[0] : 0.f) + weight * (float)v;
}

}  // namespace

// loss = mean_b( bce_mean + eloss_b + wiou_b ).  ws: f64 [B][8] scratch.  pred/mask: [B][1][H][W] f32.
extern "C" int emip_hybrid_e_loss(const float* pred, const float* mask, double* ws, float* out, int B, int H, int W,
                                  void* stream) {
    EMIP_REQUIRE(pred && mask && ws && out && B > 0 && B < 65536 && H > 0 && W > 0);
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W;
    if (emip_zero_async(ws, sizeof(double) * 8 * B, s) != EMIP_OK) return EMIP_E_LAUNCH;
    dim3 grid((HW + 256 * 8 - 1) / (256 * 8), B);
    hipLaunchKernelGGL(hybrid_pass1_kernel, grid, dim3(256), 0, s, pred, mask, ws, HW);
    hipLaunchKernelGGL(hybrid_pass2_kernel, grid, dim3(256), 0, s, pred, mask, ws, HW);
    hipLaunchKernelGGL(hybrid_final_kernel, dim3(1), dim3(64), 0, s, ws, out, B, HW);
    return emip_launch_status();
}

// out[0] (+)= weight * loss_photomatric(im, rec, mask).  ws: f64 [4] scratch.  im/rec [B][C][H][W], mask [B][1][H][W].
extern "C" int emip_photometric_loss(const float* im, const float* rec, const float* mask, double* ws, float* out,
                                     int B, int C, int H, int W, float weight, int accumulate, void* stream) {
    EMIP_REQUIRE(im && rec && mask && ws && out && B > 0 && C > 0 && H > 2 && W > 2);
    hipStream_t s = (hipStream_t)stream;
    if (emip_zero_async(ws, sizeof(double) * 4, s) != EMIP_OK) return EMIP_E_LAUNCH;
    const long total = (long)B * C * H * W;
    hipLaunchKernelGGL(photometric_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, im, rec, mask, ws, B, C, H, W);
    hipLaunchKernelGGL(photometric_final_kernel, dim3(1), dim3(64), 0, s, ws, out, (double)total,
                       (double)B * C * (H - 2) * (W - 2), (double)B * H * W, weight, accumulate);
    return emip_launch_status();
}
