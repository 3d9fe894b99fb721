// Input preparation on the device (/root/reference/dataset/dataset.py:257-260 and :76-79): the transform the drivers apply
// to every decoded RGB frame,
//     transforms.Resize((352, 352))  ->  transforms.ToTensor()  ->  transforms.Normalize(mean, std)
// Resize on a PIL image is Pillow's two-pass separable resampling on 8-bit data (Resample.c): triangle filter widened by
// the scale factor when shrinking, coefficients normalised in double precision and quantised to 22 fractional bits,
// each pass rounding to u8 with (acc + 2^21) >> 22 and clipping.  The quantised coefficient tables are built on the host
// (emip_amd/preprocess.py, a few hundred integers per frame size); the kernels below are the two integer passes, so the
// result is BIT-EXACT with Pillow, followed by ToTensor (/255) and Normalize ((x - mean) / std) in IEEE f32.
#include "common.h"

namespace {

inline int grid_for(long n, int threads) {
    long b = (n + threads - 1) / threads;
    if (b > 65535) b = 65535;
    return (int)(b < 1 ? 1 : b);
}

// horizontal pass: tmp[b][y][xx][c] from img[b][y][xmin..xmin+xmax)[c]   (C = 3: RGB frames, C = 1: 'L' ground truth)
template <int C>
__global__ __launch_bounds__(256) void resize_h_kernel(const unsigned char* __restrict__ img, long img_bs, long img_rs,
                                                       const int* __restrict__ kk, const int* __restrict__ bounds,
                                                       int ksize, unsigned char* __restrict__ tmp, int B, int H0, int Wo) {
    const long total = (long)B * H0 * Wo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xx = (int)(i % Wo), y = (int)((i / Wo) % H0);
        const long b = i / ((long)Wo * H0);
        const int xmin = bounds[2 * xx], xmax = bounds[2 * xx + 1];
        const int* k = kk + (long)xx * ksize;
        const unsigned char* row = img + b * img_bs + (long)y * img_rs + (long)xmin * C;
        int acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 1 << 21;
        for (int x = 0; x < xmax; ++x) {
            const int w = k[x];
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] += (int)row[C * x + c] * w;
        }
        unsigned char* o = tmp + i * C;
#pragma unroll
        for (int c = 0; c < C; ++c) o[c] = (unsigned char)min(max(acc[c] >> 22, 0), 255);
    }
}

// vertical pass + ToTensor + Normalize: out[b][c][yy][xx] (planar f32); u8 copy [b][yy][xx][c] optional
template <int C>
__global__ __launch_bounds__(256) void resize_v_norm_kernel(const unsigned char* __restrict__ tmp,
                                                            const int* __restrict__ kk, const int* __restrict__ bounds,
                                                            int ksize, float* __restrict__ out,
                                                            unsigned char* __restrict__ out_u8, int B, int H0, int Ho,
                                                            int Wo, float m0, float m1, float m2, float d0, float d1,
                                                            float d2) {
    const long total = (long)B * Ho * Wo;
    const long plane = (long)Ho * Wo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xx = (int)(i % Wo), yy = (int)((i / Wo) % Ho);
        const long b = i / plane;
        const int ymin = bounds[2 * yy], ymax = bounds[2 * yy + 1];
        const int* k = kk + (long)yy * ksize;
        const unsigned char* col = tmp + ((b * H0 + ymin) * (long)Wo + xx) * C;
        int acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 1 << 21;
        for (int y = 0; y < ymax; ++y) {
            const int w = k[y];
            const unsigned char* p = col + (long)y * Wo * C;
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] += (int)p[c] * w;
        }
        const float mean[3] = {m0, m1, m2}, sd[3] = {d0, d1, d2};
        float* op = out + b * C * plane + (long)yy * Wo + xx;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int v = min(max(acc[c] >> 22, 0), 255);
            if (out_u8) out_u8[i * C + c] = (unsigned char)v;
            op[c * plane] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v, 255.f), mean[c]), sd[c]);
        }
    }
}

}  // namespace

// img u8 [B][H0][W0][3] (decoded RGB; batch stride img_bs and row stride img_rs in bytes) -> out f32 [B][3][Ho][Wo]
// normalised model input; out_u8 (optional) u8 [B][Ho][Wo][3] the resized pixels; tmp u8 [B][H0][Wo][3] scratch.
// kh/bh: quantised horizontal coefficients [Wo][ksh] and bounds [Wo][2] (xmin, count); kv/bv likewise for rows.
extern "C" int emip_preprocess_rgb(const unsigned char* img, long img_bs, long img_rs, int B, int H0, int W0, const int* kh,
                                   const int* bh, int ksh, const int* kv, const int* bv, int ksv, unsigned char* tmp,
                                   float* out, unsigned char* out_u8, int Ho, int Wo, const float* mean3,
                                   const float* std3, void* stream) {
    EMIP_REQUIRE(img && kh && bh && kv && bv && tmp && out && mean3 && std3 && B > 0 && H0 > 0 && W0 > 0 && Ho > 0 && Wo > 0);
    EMIP_REQUIRE(ksh > 0 && ksv > 0 && img_rs >= (long)W0 * 3 && (B == 1 || img_bs >= (long)H0 * img_rs));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(resize_h_kernel<3>, dim3(grid_for((long)B * H0 * Wo, 256)), dim3(256), 0, s, img, img_bs, img_rs, kh, bh,
                       ksh, tmp, B, H0, Wo);
    hipLaunchKernelGGL(resize_v_norm_kernel<3>, dim3(grid_for((long)B * Ho * Wo, 256)), dim3(256), 0, s, tmp, kv, bv, ksv, out,
                       out_u8, B, H0, Ho, Wo, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    return emip_launch_status();
}

// The ground-truth transform of dataset/dataset.py:80-82 (Resize((Ho, Wo)) on the 'L' mask, ToTensor): the same two integer
// passes on one channel, then v / 255.  img u8 [B][H0][W0] -> out f32 [B][1][Ho][Wo]; tmp u8 [B][H0][Wo].
extern "C" int emip_preprocess_gray(const unsigned char* img, long img_bs, long img_rs, int B, int H0, int W0, const int* kh,
                                    const int* bh, int ksh, const int* kv, const int* bv, int ksv, unsigned char* tmp,
                                    float* out, unsigned char* out_u8, int Ho, int Wo, void* stream) {
    EMIP_REQUIRE(img && kh && bh && kv && bv && tmp && out && B > 0 && H0 > 0 && W0 > 0 && Ho > 0 && Wo > 0);
    EMIP_REQUIRE(ksh > 0 && ksv > 0 && img_rs >= (long)W0 && (B == 1 || img_bs >= (long)H0 * img_rs));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(resize_h_kernel<1>, dim3(grid_for((long)B * H0 * Wo, 256)), dim3(256), 0, s, img, img_bs, img_rs, kh,
                       bh, ksh, tmp, B, H0, Wo);
    hipLaunchKernelGGL(resize_v_norm_kernel<1>, dim3(grid_for((long)B * Ho * Wo, 256)), dim3(256), 0, s, tmp, kv, bv, ksv, out,
                       out_u8, B, H0, Ho, Wo, 0.f, 0.f, 0.f, 1.f, 1.f, 1.f);
    return emip_launch_status();
}
