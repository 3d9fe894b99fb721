// Prediction post-processing on the device (/root/reference/test.py:28-31, the step right after CoUpdater.forward):
//   out = F.upsample(mask_logits, size=shape, mode='bilinear', align_corners=False).sigmoid()
//   out = (out - out.min()) / (out.max() - out.min() + 1e-8)           (per image)
//   png = Image.fromarray(out * 255).convert('L')                       (PIL F -> L: clip to [0, 255], truncate)
// Two passes over the OUTPUT grid, nothing intermediate in HBM: pass 1 finds min / max of the probabilities, pass 2
// recomputes them and writes the bytes.
//
// Round 4: BIT FOR BIT what the reference's statements produce on a CPU (tests/golden/postprocess.npz, made by executing
// test.py:29-31,35-36; oracle/make_golden_postprocess.py).  Byte work leaves no tolerance, so every f32 operation is the one
// PyTorch-CPU / numpy / PIL perform, in their order (this file is compiled with -ffp-contract=off; the FMAs below are the ones
// the reference's build contracts):
//   * ATen upsample_bilinear2d (UpSampleKernel.cpp, the generic 2-D interpolation loop): source index
//     fma(in / out, i + 0.5, -0.5) clamped at 0, lambda1 = index - floor, lambda0 = 1 - lambda1; a row
//     t = fma(x0, l0x, x1 * l1x), the pixel fma(t0, l0y, t1 * l1y); equal sizes copy;
//   * ATen sigmoid (UnaryOpsKernel.cpp, AVX2 / AVX512 build): 1 / (1 + exp(0 - x)) with Sleef's expf_u10 -- q = rint(x log2 e),
//     two-constant Cody-Waite reduction by FMA, degree-6 Horner chain by FMA, 1 + fma(s s, u, s), scaling by two exact powers
//     of two -- for all but the last numel % 32 elements of an image, which the vectorised loop leaves to the scalar
//     1 / (1 + std::exp(-x)) (correctly rounded here: exp in f64, rounded once);
//   * numpy: min, max, (max - min) + 1e-8 in f32, (p - min) / that, * 255, each rounded to f32; PIL: <= 0 -> 0, >= 255 -> 255,
//     else truncation.
#include "common.h"

namespace {

inline int grid_for(long n, int threads) {
    long b = (n + threads - 1) / threads;
    if (b > 4096) b = 4096;
    return (int)(b < 1 ? 1 : b);
}

// index and weights of output position o along an axis of n_in -> n_out (area_pixel_compute_source_index + guard_index_and_lambda)
__device__ __forceinline__ void src_index(int o, int n_in, int n_out, int& i0, int& i1, float& l0, float& l1) {
    if (n_in == n_out) {
        i0 = i1 = o;
        l0 = 1.f;
        l1 = 0.f;
        return;
    }
    const float scale = (float)n_in / (float)n_out;
    float real = fmaf(scale, (float)o + 0.5f, -0.5f);
    real = real < 0.f ? 0.f : real;
    i0 = min((int)real, n_in - 1);
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    l1 = fminf(fmaxf(real - (float)i0, 0.f), 1.f);
    l0 = 1.f - l1;
}

__device__ __forceinline__ float resized_logit(const float* __restrict__ src, int H, int W, int Ho, int Wo, int oy, int ox) {
    int y0, y1, x0, x1;
    float hy, ly, hx, lx;
    src_index(oy, H, Ho, y0, y1, hy, ly);
    src_index(ox, W, Wo, x0, x1, hx, lx);
    const float t0 = fmaf(src[(long)y0 * W + x0], hx, src[(long)y0 * W + x1] * lx);
    const float t1 = fmaf(src[(long)y1 * W + x0], hx, src[(long)y1 * W + x1] * lx);
    return fmaf(t0, hy, t1 * ly);
}

// Sleef_expf_u10 (sleefsimdsp.c: xexpf), the exp of ATen's vectorised float kernels
__device__ __forceinline__ float sleef_expf_u10(float d) {
    const float qf = rintf(d * 1.442695040888963407359924681001892137426645954152985934135449406931f);
    const int q = (int)qf;
    float s = fmaf(qf, -0.693145751953125f, d);
    s = fmaf(qf, -1.428606765330187045e-06f, s);
    float u = 0.000198527617612853646278381f;
    u = fmaf(u, s, 0.00139304355252534151077271f);
    u = fmaf(u, s, 0.00833336077630519866943359f);
    u = fmaf(u, s, 0.0416664853692054748535156f);
    u = fmaf(u, s, 0.166666671633720397949219f);
    u = fmaf(u, s, 0.5f);
    u = 1.0f + fmaf(s * s, u, s);
    const int qh = q >> 1;
    u = (u * __int_as_float((qh + 0x7f) << 23)) * __int_as_float((q - qh + 0x7f) << 23);
    return d < -104.f ? 0.f : u;
}

// ATen's sigmoid of element `idx` of an image of `n` elements
__device__ __forceinline__ float sigmoid_aten(float x, long idx, long n) {
    if (idx < n - (n & 31)) return __fdiv_rn(1.f, 1.f + sleef_expf_u10(0.f - x));
    return __fdiv_rn(1.f, 1.f + (float)exp((double)(-x)));
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

// min / max of the PROBABILITIES (an approximated sigmoid need not be monotonic in the last bit: not those of the logits)
__global__ __launch_bounds__(256) void post_minmax_kernel(const float* __restrict__ logits, int* __restrict__ ws, int H,
                                                          int W, int Ho, int Wo) {
    const int b = blockIdx.y;
    const float* src = logits + (long)b * H * W;
    float mn = INFINITY, mx = -INFINITY;
    const long n = (long)Ho * Wo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = sigmoid_aten(resized_logit(src, H, W, Ho, Wo, (int)(i / Wo), (int)(i % Wo)), i, n);
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

__global__ __launch_bounds__(256) void post_write_kernel(const float* __restrict__ logits, const int* __restrict__ ws,
                                                         unsigned char* __restrict__ out, float* __restrict__ outf, int H,
                                                         int W, int Ho, int Wo) {
    const int b = blockIdx.y;
    const float* src = logits + (long)b * H * W;
    const float pmin = ord2f(ws[2 * b]), pmax = ord2f(ws[2 * b + 1]);
    const float den = (pmax - pmin) + 1e-8f;
    const long n = (long)Ho * Wo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float p = sigmoid_aten(resized_logit(src, H, W, Ho, Wo, (int)(i / Wo), (int)(i % Wo)), i, n);
        const float r = __fdiv_rn(p - pmin, den);
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
