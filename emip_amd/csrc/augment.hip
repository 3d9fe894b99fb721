// Training-time augmentation on the device (/root/reference/dataset/data_augment.py:12-45, applied per sample at
// dataset/dataset.py:95-98 before the Resize / ToTensor / Normalize of csrc/preprocess.hip).  The reference runs these on
// PIL images, so "the same result" means Pillow's 8-bit arithmetic, reproduced operation by operation:
//   colorEnhance  (:22-31)  ImageEnhance.Brightness -> Contrast -> Color -> Sharpness, each Image.blend(degenerate, image, f)
//                           = (UINT8)(d + f * (c - d)) in float32 (clipped to [0,255] when f is outside [0,1]); degenerates:
//                           black / the rounded mean of the 'L' conversion / the 'L' conversion / ImageFilter.SMOOTH
//   randomRotation (:12-19) Image.rotate(angle, BICUBIC): inverse affine map of the pixel centre in float64 and Pillow's
//                           4x4 cubic interpolation with edge clamping, zero outside the source
//   randomPeper    (:34-45) 0 / 255 written at random pixels of the ground truth (positions drawn by the host)
// The random draws stay on the host (emip_amd/data_augment.py consumes Python's `random` / numpy's generator in the
// reference's order); only their values cross the boundary.  Pillow's x86-64 build rounds every multiplication and
// addition separately, so this file is compiled with -ffp-contract=off (csrc/Makefile): HIP's __fmul_rn / __fadd_rn are
// plain operators and would be fused into FMAs under the default -ffp-contract=fast.
#include "common.h"

namespace {

inline int grid_for(long n, int threads) {
    long b = (n + threads - 1) / threads;
    if (b > 65535) b = 65535;
    return (int)(b < 1 ? 1 : b);
}

// Pillow Blend.c: interpolation truncates, extrapolation clips then truncates
__device__ __forceinline__ int blend8(int d, int c, float f, bool inside) {
    const float t = __fadd_rn((float)d, __fmul_rn(f, (float)(c - d)));
    if (inside) return (int)(unsigned char)t;
    if (t <= 0.f) return 0;
    if (t >= 255.f) return 255;
    return (int)(unsigned char)t;
}

// Pillow Convert.c rgb2l
__device__ __forceinline__ int luma(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

// sum of the 'L' conversion of the brightness-adjusted frame (Contrast's mean)
__global__ __launch_bounds__(256) void aug_lsum_kernel(const unsigned char* __restrict__ img, long n, float fb, bool in_b,
                                                       unsigned long long* __restrict__ lsum) {
    unsigned long long acc = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const unsigned char* p = img + 3 * i;
        acc += (unsigned)luma(blend8(0, p[0], fb, in_b), blend8(0, p[1], fb, in_b), blend8(0, p[2], fb, in_b));
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(lsum, acc);
}

// brightness -> contrast -> colour for one pixel each
__global__ __launch_bounds__(256) void aug_point_kernel(const unsigned char* __restrict__ img, unsigned char* __restrict__ out,
                                                        long n, float fb, bool in_b, float fc, bool in_c, float fk,
                                                        bool in_k, const unsigned long long* __restrict__ lsum) {
    // int(ImageStat.Stat(L).mean[0] + 0.5): the mean is an exact integer ratio, so the rounding is done in integers
    const unsigned long long s = *lsum;
    const int mean = (int)((2 * s + (unsigned long long)n) / (2 * (unsigned long long)n));
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const unsigned char* p = img + 3 * i;
        int c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) c[k] = blend8(mean, blend8(0, p[k], fb, in_b), fc, in_c);
        const int l = luma(c[0], c[1], c[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) out[3 * i + k] = (unsigned char)blend8(l, c[k], fk, in_k);
    }
}

// Sharpness: blend(SMOOTH(image), image, f).  Filter.c ImagingFilter3x3: float32 accumulation row by row starting from
// offset + 0.5, border pixels copied.
__global__ __launch_bounds__(256) void aug_sharp_kernel(const unsigned char* __restrict__ img, unsigned char* __restrict__ out,
                                                        int H, int W, float fs, bool in_s) {
    const float k1 = __fdiv_rn(1.f, 13.f), k5 = __fdiv_rn(5.f, 13.f);
    const long n = (long)H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)(i / W);
        const bool border = x == 0 || y == 0 || x == W - 1 || y == H - 1;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int c = img[3 * i + k];
            int d = c;
            if (!border) {
                float ss = 0.5f;
                const unsigned char* r1 = img + 3 * (i + W) + k;        // Pillow starts with the row below
                const unsigned char* r0 = img + 3 * i + k;
                const unsigned char* r_1 = img + 3 * (i - W) + k;
                float t = __fadd_rn(__fadd_rn(__fmul_rn((float)r1[-3], k1), __fmul_rn((float)r1[0], k1)), __fmul_rn((float)r1[3], k1));
                ss = __fadd_rn(ss, t);
                t = __fadd_rn(__fadd_rn(__fmul_rn((float)r0[-3], k1), __fmul_rn((float)r0[0], k5)), __fmul_rn((float)r0[3], k1));
                ss = __fadd_rn(ss, t);
                t = __fadd_rn(__fadd_rn(__fmul_rn((float)r_1[-3], k1), __fmul_rn((float)r_1[0], k1)), __fmul_rn((float)r_1[3], k1));
                ss = __fadd_rn(ss, t);
                d = ss <= 0.f ? 0 : (ss >= 255.f ? 255 : (int)(unsigned char)ss);
            }
            out[3 * i + k] = (unsigned char)blend8(d, c, fs, in_s);
        }
    }
}

// Geometry.c BICUBIC macro, double precision, evaluated in Pillow's order
__device__ __forceinline__ double cubic(double v1, double v2, double v3, double v4, double d) {
    const double p1 = v2;
    const double p2 = __dadd_rn(-v1, v3);
    const double p3 = __dsub_rn(__dadd_rn(__dmul_rn(2.0, __dsub_rn(v1, v2)), v3), v4);
    const double p4 = __dadd_rn(__dsub_rn(__dadd_rn(-v1, v2), v3), v4);
    return __dadd_rn(p1, __dmul_rn(d, __dadd_rn(p2, __dmul_rn(d, __dadd_rn(p3, __dmul_rn(d, p4))))));
}

// Image.rotate(angle, BICUBIC) = transform(AFFINE, matrix, BICUBIC, fill = 0): Geometry.c affine_transform +
// bicubic_filter8 / bicubic_filter32RGB
template <int C>
__global__ __launch_bounds__(256) void aug_rotate_kernel(const unsigned char* __restrict__ img, unsigned char* __restrict__ out,
                                                         int H, int W, double a0, double a1, double a2, double a3, double a4,
                                                         double a5) {
    const long n = (long)H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % W), oy = (int)(i / W);
        const double xc = (double)ox + 0.5, yc = (double)oy + 0.5;
        double xin = __dadd_rn(__dadd_rn(__dmul_rn(a0, xc), __dmul_rn(a1, yc)), a2);
        double yin = __dadd_rn(__dadd_rn(__dmul_rn(a3, xc), __dmul_rn(a4, yc)), a5);
        unsigned char* o = out + (long)C * i;
        if (xin < 0.0 || xin >= (double)W || yin < 0.0 || yin >= (double)H) {
#pragma unroll
            for (int k = 0; k < C; ++k) o[k] = 0;
            continue;
        }
        xin -= 0.5;
        yin -= 0.5;
        int x = xin < 0.0 ? (int)floor(xin) : (int)xin;
        int y = yin < 0.0 ? (int)floor(yin) : (int)yin;
        const double dx = xin - (double)x, dy = yin - (double)y;
        --x;
        --y;
        int xs[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) xs[j] = min(max(x + j, 0), W - 1);
        const int y0 = min(max(y, 0), H - 1);
#pragma unroll
        for (int k = 0; k < C; ++k) {
            double v[4];
            const unsigned char* r = img + ((long)y0 * W) * C + k;
            v[0] = cubic((double)r[xs[0] * C], (double)r[xs[1] * C], (double)r[xs[2] * C], (double)r[xs[3] * C], dx);
#pragma unroll
            for (int j = 1; j < 4; ++j) {
                if (y + j >= 0 && y + j < H) {
                    r = img + ((long)(y + j) * W) * C + k;
                    v[j] = cubic((double)r[xs[0] * C], (double)r[xs[1] * C], (double)r[xs[2] * C], (double)r[xs[3] * C], dx);
                } else {
                    v[j] = v[j - 1];
                }
            }
            const double r1 = cubic(v[0], v[1], v[2], v[3], dy);
            o[k] = r1 <= 0.0 ? 0 : (r1 >= 255.0 ? 255 : (unsigned char)r1);     // Pillow truncates here
        }
    }
}

__global__ void aug_scatter_kernel(unsigned char* __restrict__ img, const int* __restrict__ offs,
                                   const unsigned char* __restrict__ vals, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) img[offs[i]] = vals[i];
}

inline bool inside01(float f) { return f >= 0.f && f <= 1.f; }

}  // namespace

// colorEnhance (data_augment.py:22-31) with the four factors the host drew.  img / out u8 [H][W][3] (packed RGB, out may
// not alias img), tmp u8 [H][W][3] scratch, lsum: 8 bytes of scratch.
extern "C" int emip_color_enhance(const unsigned char* img, unsigned char* out, unsigned char* tmp, void* lsum, int H, int W,
                                  float f_bright, float f_contrast, float f_color, float f_sharp, void* stream) {
    EMIP_REQUIRE(img && out && tmp && lsum && H >= 3 && W >= 3 && img != out && tmp != out && tmp != img);
    hipStream_t s = (hipStream_t)stream;
    const long n = (long)H * W;
    if (hipMemsetAsync(lsum, 0, 8, s) != hipSuccess) return EMIP_E_LAUNCH;
    hipLaunchKernelGGL(aug_lsum_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, s, img, n, f_bright, inside01(f_bright),
                       reinterpret_cast<unsigned long long*>(lsum));
    hipLaunchKernelGGL(aug_point_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, img, tmp, n, f_bright, inside01(f_bright),
                       f_contrast, inside01(f_contrast), f_color, inside01(f_color),
                       reinterpret_cast<const unsigned long long*>(lsum));
    hipLaunchKernelGGL(aug_sharp_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, tmp, out, H, W, f_sharp, inside01(f_sharp));
    return emip_launch_status();
}

// Image.rotate(angle, Image.BICUBIC) (data_augment.py:13-18) given the inverse affine matrix Image.rotate computes
// (a HOST pointer to 6 doubles: xin = a0 (x + .5) + a1 (y + .5) + a2, yin = a3 .. a5).  img / out u8 [H][W][C], C = 3 (RGB)
// or 1 ('L' ground truth); out may not alias img.
extern "C" int emip_rotate_bicubic(const unsigned char* img, unsigned char* out, int H, int W, int C, const double* matrix6,
                                   void* stream) {
    EMIP_REQUIRE(img && out && matrix6 && img != out && H > 0 && W > 0 && (C == 1 || C == 3));
    hipStream_t s = (hipStream_t)stream;
    const long n = (long)H * W;
    const double* a = matrix6;
    if (C == 3) {
        hipLaunchKernelGGL(aug_rotate_kernel<3>, dim3(grid_for(n, 256)), dim3(256), 0, s, img, out, H, W, a[0], a[1], a[2], a[3],
                           a[4], a[5]);
    } else {
        hipLaunchKernelGGL(aug_rotate_kernel<1>, dim3(grid_for(n, 256)), dim3(256), 0, s, img, out, H, W, a[0], a[1], a[2], a[3],
                           a[4], a[5]);
    }
    return emip_launch_status();
}

// randomPeper (data_augment.py:34-45): img[offs[i]] = vals[i]; the host passes each pixel at most once (its LAST draw).
extern "C" int emip_scatter_u8(unsigned char* img, const int* offs, const unsigned char* vals, int n, void* stream) {
    EMIP_REQUIRE(img && n >= 0 && (n == 0 || (offs && vals)));
    if (n == 0) return EMIP_OK;
    hipLaunchKernelGGL(aug_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, img, offs, vals, n);
    return emip_launch_status();
}
