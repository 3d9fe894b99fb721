// Prediction post-processing on the device (/root/reference/test.py:28-31, the step right after CoUpdater.forward):
//   out = F.upsample(mask_logits, size=shape, mode='bilinear', align_corners=False).sigmoid()
//   out = (out - out.min()) / (out.max() - out.min() + 1e-8)           (per image)
//   png = Image.fromarray(out * 255).convert('L')                       (PIL F -> L: clip to [0, 255], truncate)
// Two passes over the OUTPUT grid, nothing intermediate in HBM: pass 1 finds min / max of the resized logits (sigmoid is
// monotonic, so they map to the min / max of the probabilities), pass 2 recomputes the resize and writes the bytes.
#include "common.h"

namespace {

inline int grid_for(long n, int threads) {
    long b = (n + threads - 1) / threads;
    if (b > 4096) b = 4096;
    return (int)(b < 1 ? 1 : b);
}

__device__ __forceinline__ float resized_logit(const float* __restrict__ src, int H, int W, int Ho, int Wo, int oy,
                                               int ox) {
    // ATen upsample_bilinear2d, align_corners=False, scale = in / out
    float sy = ((float)H / (float)Ho) * ((float)oy + 0.5f) - 0.5f;
    float sx = ((float)W / (float)Wo) * ((float)ox + 0.5f) - 0.5f;
    sy = sy < 0.f ? 0.f : sy;
    sx = sx < 0.f ? 0.f : sx;
    int y0 = (int)sy, x0 = (int)sx;
    y0 = min(y0, H - 1);
    x0 = min(x0, W - 1);
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    return hy * (hx * src[(long)y0 * W + x0] + lx * src[(long)y0 * W + x1]) +
           ly * (hx * src[(long)y1 * W + x0] + lx * src[(long)y1 * W + x1]);
}

// order-preserving float <-> int encoding for atomicMin / atomicMax
__device__ __forceinline__ int f2ord(float f) {
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

__global__ void post_init_kernel(int* __restrict__ ws, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        ws[2 * b] = f2ord(INFINITY);
        ws[2 * b + 1] = f2ord(-INFINITY);
    }
}

__global__ __launch_bounds__(256) void post_minmax_kernel(const float* __restrict__ logits, int* __restrict__ ws, int H,
                                                          int W, int Ho, int Wo) {
    const int b = blockIdx.y;
    const float* src = logits + (long)b * H * W;
    float mn = INFINITY, mx = -INFINITY;
    const long n = (long)Ho * Wo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = resized_logit(src, H, W, Ho, Wo, (int)(i / Wo), (int)(i % Wo));
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o));
        mx = fmaxf(mx, __shfl_xor(mx, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(ws + 2 * b, f2ord(mn));
        atomicMax(ws + 2 * b + 1, f2ord(mx));
    }
}

__device__ __forceinline__ float sigmoid_f32(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void post_write_kernel(const float* __restrict__ logits, const int* __restrict__ ws,
                                                         unsigned char* __restrict__ out, float* __restrict__ outf, int H,
                                                         int W, int Ho, int Wo) {
    const int b = blockIdx.y;
    const float* src = logits + (long)b * H * W;
    const float pmin = sigmoid_f32(ord2f(ws[2 * b])), pmax = sigmoid_f32(ord2f(ws[2 * b + 1]));
    const float den = pmax - pmin + 1e-8f;
    const long n = (long)Ho * Wo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float p = sigmoid_f32(resized_logit(src, H, W, Ho, Wo, (int)(i / Wo), (int)(i % Wo)));
        const float r = (p - pmin) / den;
        if (outf) outf[(long)b * n + i] = r;                     // the float map train.py:125-127 feeds to the metrics
        if (out) {
            const float v = r * 255.f;
            out[(long)b * n + i] = (unsigned char)(v <= 0.f ? 0.f : (v >= 255.f ? 255.f : v));     // clip, truncate
        }
    }
}

}  // namespace

// logits f32 [B][1][H][W] -> out u8 [B][Ho][Wo]; ws: int [2*B] scratch.
extern "C" int emip_postprocess_mask(const float* logits, unsigned char* out, int* ws, int B, int H, int W, int Ho, int Wo,
                                     void* stream) {
    EMIP_REQUIRE(logits && out && ws && B > 0 && B < 65536 && H > 0 && W > 0 && Ho > 0 && Wo > 0);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(post_init_kernel, dim3((B + 63) / 64), dim3(64), 0, s, ws, B);
    dim3 grid(grid_for((long)Ho * Wo, 256 * 4), B);
    hipLaunchKernelGGL(post_minmax_kernel, grid, dim3(256), 0, s, logits, ws, H, W, Ho, Wo);
    hipLaunchKernelGGL(post_write_kernel, grid, dim3(256), 0, s, logits, ws, out, (float*)nullptr, H, W, Ho, Wo);
    return emip_launch_status();
}

// same resize + sigmoid + min-max, written as the f32 map in [0, 1] (train.py:125-127: the input of the validation metrics)
extern "C" int emip_postprocess_mask_f32(const float* logits, float* out, int* ws, int B, int H, int W, int Ho, int Wo,
                                         void* stream) {
    EMIP_REQUIRE(logits && out && ws && B > 0 && B < 65536 && H > 0 && W > 0 && Ho > 0 && Wo > 0);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(post_init_kernel, dim3((B + 63) / 64), dim3(64), 0, s, ws, B);
    dim3 grid(grid_for((long)Ho * Wo, 256 * 4), B);
    hipLaunchKernelGGL(post_minmax_kernel, grid, dim3(256), 0, s, logits, ws, H, W, Ho, Wo);
    hipLaunchKernelGGL(post_write_kernel, grid, dim3(256), 0, s, logits, ws, (unsigned char*)nullptr, out, H, W, Ho, Wo);
    return emip_launch_status();
}
