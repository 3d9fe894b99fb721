// Backward building blocks of the EMIP training step (row T of SURVEY.md section 8): the HBM-bound pieces that sit
// between the MFMA contractions (emip_gemm for dgrad, emip_gemm_tn / emip_conv2d_wgrad for weight gradients).
//   softmax over rows (+ additive group mask, + valid length) and its backward   -> unfused attention backward
//   2-D transpose with zero padding                                              -> K^T operand of dQ = dS K
//   GELU backward, depthwise-3x3 weight gradient                                 -> PVT Mlp / MDTA depthwise
//   per-channel (dy, dy*xhat) sums + apply                                       -> train-mode BatchNorm backward
//   adjoint of bilinear resampling (channels-last and planar)                    -> NCD decoder, mask / flow x8
// All channels-last, 8-16 B per lane; reductions combine per-workgroup partials with f32 atomics.
#include "common.h"

namespace {

inline int grid_for(long n, int threads) {
    long b = (n + threads - 1) / threads;
    if (b > 4096) b = 4096;
    return (int)(b < 1 ? 1 : b);
}

// ---- softmax over the first L columns of rows of width ld (columns >= L are written as 0) --------------------
// one wave per row; s = x*scale (+ mask2 when gid_q[row % period] != gid_k[col]).  L <= 2048.
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const T* __restrict__ X, T* __restrict__ Y, long rows,
                                                           int L, long ld, float scale, const int* __restrict__ gq,
                                                           const int* __restrict__ gk, long period, long win_stride) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nw = (long)gridDim.x * 4;
    for (long r = wave; r < rows; r += nw) {
        const T* x = X + r * ld;
        T* y = Y + r * ld;
        int qg = 0;
        const int* gkw = nullptr;
        if (gq) {
            const long local = r % period;               // row inside its (batch, window) block
            const long win = (r / period) % (win_stride > 0 ? win_stride : 1);
            qg = gq[win * period + local];
            gkw = gk + win * (long)L;
        }
        float v[32];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int c = lane + 64 * i;
            v[i] = -INFINITY;
            if (c < L) {
                float s = to_f32<T>(x[c]) * scale;
                if (gq && gkw[c] != qg) s += -100.0f;
                v[i] = s;
                mx = fmaxf(mx, s);
            }
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int c = lane + 64 * i;
            if (c < L) {
                v[i] = expf(v[i] - mx);
                sum += v[i];
            }
        }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int c = lane + 64 * i;
            if (c < ld) y[c] = from_f32<T>(c < L ? v[i] * inv : 0.f);
        }
    }
}

// bf16 rows with 16 bytes per lane (ld % 8 == 0, ld <= 512 NIT): the element-per-lane form above reads 2 bytes per lane --
// the 484-key window softmaxes of the GMFlow training forward ran at 0.9 TB/s.
template <int NIT>
__global__ __launch_bounds__(256) void softmax_rows_vec_kernel(const bf16_t* __restrict__ X, bf16_t* __restrict__ Y, long rows,
                                                               int L, long ld, float scale, const int* __restrict__ gq,
                                                               const int* __restrict__ gk, long period, long win_stride) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nw = (long)gridDim.x * 4;
    for (long r = wave; r < rows; r += nw) {
        const bf16_t* x = X + r * ld;
        bf16_t* y = Y + r * ld;
        int qg = 0;
        const int* gkw = nullptr;
        if (gq) {
            const long local = r % period;
            const long win = (r / period) % (win_stride > 0 ? win_stride : 1);
            qg = gq[win * period + local];
            gkw = gk + win * (long)L;
        }
        float v[NIT][8];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int c0 = 8 * (lane + 64 * i);
            uint4 u = make_uint4(0u, 0u, 0u, 0u);
            if (c0 < ld) u = *reinterpret_cast<const uint4*>(x + c0);
            const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = c0 + j;
                float sv = -INFINITY;
                if (c < L) {
                    sv = __uint_as_float((j & 1) ? (w[j >> 1] & 0xFFFF0000u) : (w[j >> 1] << 16)) * scale;
                    if (gq && gkw[c] != qg) sv += -100.0f;
                    mx = fmaxf(mx, sv);
                }
                v[i][j] = sv;
            }
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NIT; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float e = (8 * (lane + 64 * i) + j < L) ? expf(v[i][j] - mx) : 0.f;
                v[i][j] = e;
                sum += e;
            }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int c0 = 8 * (lane + 64 * i);
            if (c0 < ld) {
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16_t)(v[i][j] * inv);      // columns >= L hold 0
                *reinterpret_cast<bf16x8*>(y + c0) = o;
            }
        }
    }
}

template <int NIT>
__global__ __launch_bounds__(256) void softmax_bwd_rows_vec_kernel(const bf16_t* __restrict__ P, const bf16_t* __restrict__ DP,
                                                                   bf16_t* __restrict__ DS, long rows, int L, long ld, float scale) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nw = (long)gridDim.x * 4;
    for (long r = wave; r < rows; r += nw) {
        float pv[NIT][8], dv[NIT][8];
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int c0 = 8 * (lane + 64 * i);
            uint4 a = make_uint4(0u, 0u, 0u, 0u), b = a;
            if (c0 < ld) {
                a = *reinterpret_cast<const uint4*>(P + r * ld + c0);
                b = *reinterpret_cast<const uint4*>(DP + r * ld + c0);
            }
            const unsigned wa[4] = {a.x, a.y, a.z, a.w}, wb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool in = c0 + j < L;
                pv[i][j] = in ? __uint_as_float((j & 1) ? (wa[j >> 1] & 0xFFFF0000u) : (wa[j >> 1] << 16)) : 0.f;
                dv[i][j] = in ? __uint_as_float((j & 1) ? (wb[j >> 1] & 0xFFFF0000u) : (wb[j >> 1] << 16)) : 0.f;
                dot += pv[i][j] * dv[i][j];
            }
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int c0 = 8 * (lane + 64 * i);
            if (c0 < ld) {
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16_t)(pv[i][j] * (dv[i][j] - dot) * scale);   // columns >= L: p = 0
                *reinterpret_cast<bf16x8*>(DS + r * ld + c0) = o;
            }
        }
    }
}

// dS = P * (dP - rowsum(P*dP)) * scale   (columns >= L -> 0)
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const T* __restrict__ P, const T* __restrict__ DP,
                                                               T* __restrict__ DS, long rows, int L, long ld,
                                                               float scale) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nw = (long)gridDim.x * 4;
    for (long r = wave; r < rows; r += nw) {
        const T* p = P + r * ld;
        const T* dp = DP + r * ld;
        float pv[32], dv[32];
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int c = lane + 64 * i;
            pv[i] = dv[i] = 0.f;
            if (c < L) {
                pv[i] = to_f32<T>(p[c]);
                dv[i] = to_f32<T>(dp[c]);
                dot += pv[i] * dv[i];
            }
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int c = lane + 64 * i;
            if (c < ld) DS[r * ld + c] = from_f32<T>(c < L ? pv[i] * (dv[i] - dot) * scale : 0.f);
        }
    }
}

// ---- 128-column rows (the 121-key SRA attention backward, padded to 128): 16 lanes per row, 8 columns per lane -------
// One wave covers 4 rows per pass with 16-byte accesses (bf16) instead of one row with 2-byte accesses.
template <typename T>
__device__ __forceinline__ void load8(const T* p, float (&v)[8]) {
    if (sizeof(T) == 2) {
        const uint4 u = *reinterpret_cast<const uint4*>(p);
        const T* t = reinterpret_cast<const T*>(&u);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = to_f32<T>(t[j]);
    } else {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
}
template <typename T>
__device__ __forceinline__ void store8(T* p, const float (&v)[8]) {
    if (sizeof(T) == 2) {
        uint4 u;
        T* t = reinterpret_cast<T*>(&u);
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = from_f32<T>(v[j]);
        *reinterpret_cast<uint4*>(p) = u;
    } else {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float group16_max(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_rows128_kernel(const T* __restrict__ X, T* __restrict__ Y, long rows,
                                                              int L, float scale) {
    const int sub = threadIdx.x & 15;
    const long g0 = ((long)blockIdx.x * 256 + threadIdx.x) >> 4, ng = ((long)gridDim.x * 256) >> 4;
    const long iters = (rows + ng - 1) / ng;                       // uniform trip count: shuffles stay convergent
    for (long it = 0; it < iters; ++it) {
        const long r = g0 + it * ng;
        const long rc = r < rows ? r : rows - 1;
        float v[8];
        load8<T>(X + rc * 128 + sub * 8, v);
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = (sub * 8 + j < L) ? v[j] * scale : -INFINITY;
            mx = fmaxf(mx, v[j]);
        }
        mx = group16_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = (sub * 8 + j < L) ? expf(v[j] - mx) : 0.f;
            sum += v[j];
        }
        const float inv = 1.f / group16_sum(sum);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= inv;
        if (r < rows) store8<T>(Y + r * 128 + sub * 8, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_rows128_kernel(const T* __restrict__ P, const T* __restrict__ DP,
                                                                  T* __restrict__ DS, long rows, int L, float scale) {
    const int sub = threadIdx.x & 15;
    const long g0 = ((long)blockIdx.x * 256 + threadIdx.x) >> 4, ng = ((long)gridDim.x * 256) >> 4;
    const long iters = (rows + ng - 1) / ng;
    for (long it = 0; it < iters; ++it) {
        const long r = g0 + it * ng;
        const long rc = r < rows ? r : rows - 1;
        float pv[8], dv[8];
        load8<T>(P + rc * 128 + sub * 8, pv);
        load8<T>(DP + rc * 128 + sub * 8, dv);
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (sub * 8 + j >= L) pv[j] = dv[j] = 0.f;
            dot += pv[j] * dv[j];
        }
        dot = group16_sum(dot);
#pragma unroll
        for (int j = 0; j < 8; ++j) pv[j] = pv[j] * (dv[j] - dot) * scale;
        if (r < rows) store8<T>(DS + r * 128 + sub * 8, pv);
    }
}

// ---- long rows (the EMIP-long memory read: up to 5 x 1936 = 9680 keys): one workgroup per row, three passes ----------
__device__ __forceinline__ float block_reduce_max(float v, float* sh) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    v = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
    return v;
}
__device__ __forceinline__ float block_reduce_sum(float v, float* sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    v = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return v;
}
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_long_kernel(const T* __restrict__ X, T* __restrict__ Y, long rows,
                                                                int L, long ld, float scale) {
    __shared__ float sh[4];
    for (long r = blockIdx.x; r < rows; r += gridDim.x) {
        const T* x = X + r * ld;
        T* y = Y + r * ld;
        float mx = -INFINITY;
        for (int c = threadIdx.x; c < L; c += 256) mx = fmaxf(mx, to_f32<T>(x[c]) * scale);
        mx = block_reduce_max(mx, sh);
        float sum = 0.f;
        for (int c = threadIdx.x; c < L; c += 256) sum += expf(to_f32<T>(x[c]) * scale - mx);
        const float inv = 1.f / block_reduce_sum(sum, sh);
        for (int c = threadIdx.x; c < ld; c += 256)
            y[c] = from_f32<T>(c < L ? expf(to_f32<T>(x[c]) * scale - mx) * inv : 0.f);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_rows_long_kernel(const T* __restrict__ P, const T* __restrict__ DP,
                                                                    T* __restrict__ DS, long rows, int L, long ld,
                                                                    float scale) {
    __shared__ float sh[4];
    for (long r = blockIdx.x; r < rows; r += gridDim.x) {
        const T* p = P + r * ld;
        const T* dp = DP + r * ld;
        float dot = 0.f;
        for (int c = threadIdx.x; c < L; c += 256) dot += to_f32<T>(p[c]) * to_f32<T>(dp[c]);
        dot = block_reduce_sum(dot, sh);
        for (int c = threadIdx.x; c < ld; c += 256)
            DS[r * ld + c] = from_f32<T>(c < L ? to_f32<T>(p[c]) * (to_f32<T>(dp[c]) - dot) * scale : 0.f);
    }
}

// Y[z][c][r] = r < R ? X[z][r][c] : 0   for c < C, r < Rpad   (X rows of stride ldx, Y rows of stride Rpad)
template <typename T>
__global__ __launch_bounds__(256) void transpose_pad_kernel(const T* __restrict__ X, long ldx, long bsx,
                                                            T* __restrict__ Y, long bsy, int R, int C, int Rpad,
                                                            int heads, long hsx) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const long z = blockIdx.z;
    X += heads > 1 ? (z / heads) * bsx + (z % heads) * hsx - z * bsx : 0;      // input of (b, h): b * bsx + h * hsx
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? to_f32<T>(X[z * bsx + (long)r * ldx + c]) : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < Rpad) Y[z * bsy + (long)c * Rpad + r] = from_f32<T>(tile[tx][i]);
    }
}

// ---- GELU backward: dz = dy * (Phi(z) + z * phi(z)) --------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ Z, long ldz, const T* __restrict__ DY,
                                                       long lddy, T* __restrict__ DZ, long lddz, long M, int C) {
    const int nv = C >> 2;
    const long total = M * nv;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int vi = (int)(idx % nv);
        const long row = idx / nv;
        float z[4], dy[4], o[4];
        Vec4<T>::load(Z + row * ldz + vi * 4, z);
        Vec4<T>::load(DY + row * lddy + vi * 4, dy);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] = dy[j] * gelu_grad_t<T>(z[j]);
        }
        Vec4<T>::store(DZ + row * lddz + vi * 4, o);
    }
}

// the same with 16-byte accesses (C a multiple of 16 / sizeof(T), 16-byte aligned rows)
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_wide_kernel(const T* __restrict__ Z, long ldz, const T* __restrict__ DY,
                                                            long lddy, T* __restrict__ DZ, long lddz, long M, int C) {
    constexpr int VEC = 16 / sizeof(T);
    const int nv = C / VEC;
    const long total = M * nv;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int vi = (int)(idx % nv);
        const long row = idx / nv;
        const uint4 zv = *reinterpret_cast<const uint4*>(Z + row * ldz + vi * VEC);
        const uint4 dv = *reinterpret_cast<const uint4*>(DY + row * lddy + vi * VEC);
        uint4 ov;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float z = to_f32<T>(reinterpret_cast<const T*>(&zv)[j]);
            const float dy = to_f32<T>(reinterpret_cast<const T*>(&dv)[j]);
            reinterpret_cast<T*>(&ov)[j] = from_f32<T>(dy * gelu_grad_t<T>(z));
        }
        *reinterpret_cast<uint4*>(DZ + row * lddz + vi * VEC) = ov;
    }
}

// ---- depthwise 3x3 weight / bias gradient: dW[tap][c] += sum_pix dY[pix][c] * X[pix+tap][c];  db[c] += sum dY -----
template <typename T>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const T* __restrict__ X, long ldx, const T* __restrict__ DY,
                                                           long lddy, float* __restrict__ dW, float* __restrict__ db,
                                                           int B, int H, int Wd, int C, int rows_per_block) {
    // grid: (channel groups of 64 x row chunks, B); thread = (16-byte channel vector, pixel lane).  All loads are
    // unconditional 16-byte loads on clamped addresses, masked afterwards (a load inside a branch is waited for on the spot).
    constexpr int VEC = 16 / sizeof(T);          // channels per thread: 4 (f32) or 8 (bf16)
    constexpr int NCV = 64 / VEC;                // channel vectors per 64-channel group
    constexpr int NPL = 256 / NCV;               // pixel lanes
    const int cv = threadIdx.x % NCV, pl = threadIdx.x / NCV;
    const int cgroups = (C + 63) / 64;
    const int cg = blockIdx.x % cgroups, chunk = blockIdx.x / cgroups;
    const long b = blockIdx.y;
    const int c = cg * 64 + cv * VEC;
    const bool cok = c + VEC <= C;               // C % VEC == 0 is required by the entry point
    const int cc = cok ? c : 0;
    float aw[9][VEC], ab[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) ab[j] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < VEC; ++j) aw[t][j] = 0.f;
    const int y0 = chunk * rows_per_block, y1 = min(H, y0 + rows_per_block);
    const int npix = (y1 - y0) * Wd;                 // the chunk's pixels, flattened so that every pixel lane has work
    {
        for (int p = pl; p < npix; p += NPL) {
            const int y = y0 + p / Wd, x = p % Wd;
            const uint4 dv = mask4(*reinterpret_cast<const uint4*>(DY + ((b * H + y) * (long)Wd + x) * lddy + cc), cok);
            float dy[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                dy[j] = to_f32<T>(reinterpret_cast<const T*>(&dv)[j]);
                ab[j] += dy[j];
            }
            uint4 xv[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
                const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)Wd;
                const int iyc = min(max(iy, 0), H - 1), ixc = min(max(ix, 0), Wd - 1);
                xv[t] = mask4(*reinterpret_cast<const uint4*>(X + ((b * H + iyc) * (long)Wd + ixc) * ldx + cc), ok);
            }
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    aw[t][j] = fmaf(dy[j], to_f32<T>(reinterpret_cast<const T*>(&xv[t])[j]), aw[t][j]);
        }
    }
    // reduce over the pixel lanes: inside a wave the lanes with the same cv sit NCV apart (xor-shuffle), then 4 waves
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < VEC; ++j)
#pragma unroll
            for (int o = NCV; o < 64; o <<= 1) aw[t][j] += __shfl_xor(aw[t][j], o);
#pragma unroll
    for (int j = 0; j < VEC; ++j)
#pragma unroll
        for (int o = NCV; o < 64; o <<= 1) ab[j] += __shfl_xor(ab[j], o);
    __shared__ float red[10][64][4];
    if (lane < NCV) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < VEC; ++j) red[t][lane * VEC + j][wave] = aw[t][j];
#pragma unroll
        for (int j = 0; j < VEC; ++j) red[9][lane * VEC + j][wave] = ab[j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 10 * 64; i += 256) {
        const int t = i / 64, ch64 = i % 64;
        const float s = red[t][ch64][0] + red[t][ch64][1] + red[t][ch64][2] + red[t][ch64][3];
        const int ch = cg * 64 + ch64;
        if (ch < C) {
            if (t < 9) atomicAdd(dW + t * C + ch, s);
            else if (db) atomicAdd(db + ch, s);
        }
    }
}

// ---- train-mode BatchNorm backward ------------------------------------------------------------------------------
// pass 1: per channel s1 = sum dy', s2 = sum dy' * xhat   (dy' = dy masked by the ReLU of the forward output)
// pass 2: dx = gamma * rstd * (dy' - s1/n - xhat * s2/n);  dgamma = s2, dbeta = s1.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const T* __restrict__ X, long ldx, const T* __restrict__ DY,
                                                           long lddy, const T* __restrict__ OUT, long ldo,
                                                           const double* __restrict__ fsums, float eps,
                                                           float* __restrict__ s12, long rows, int C,
                                                           int rows_per_block) {
    __shared__ float red[256 * 8];
    const int nv = C >> 2;
    const int plan = 256 / nv;
    const int vi = threadIdx.x % nv, rl = threadIdx.x / nv;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (rl < plan) {
        float mean[4], rstd[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = vi * 4 + j;
            const double m = fsums[c * 2] / (double)rows;
            const double var = fmax(fsums[c * 2 + 1] / (double)rows - m * m, 0.0);
            mean[j] = (float)m;
            rstd[j] = (float)(1.0 / sqrt(var + (double)eps));
        }
        for (long r = r0 + rl; r < r1; r += plan) {
            float x[4], dy[4], o[4];
            Vec4<T>::load(X + r * ldx + vi * 4, x);
            Vec4<T>::load(DY + r * lddy + vi * 4, dy);
            if (OUT) Vec4<T>::load(OUT + r * ldo + vi * 4, o);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float g = (OUT && o[j] <= 0.f) ? 0.f : dy[j];
                s1[j] += g;
                s2[j] = fmaf(g, (x[j] - mean[j]) * rstd[j], s2[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[threadIdx.x * 8 + j] = s1[j];
        red[threadIdx.x * 8 + 4 + j] = s2[j];
    }
    __syncthreads();
    if (threadIdx.x < nv) {
        float ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int l = 0; l < plan; ++l)
#pragma unroll
            for (int j = 0; j < 8; ++j) ts[j] += red[(l * nv + threadIdx.x) * 8 + j];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            atomicAdd(s12 + (threadIdx.x * 4 + j) * 2, ts[j]);
            atomicAdd(s12 + (threadIdx.x * 4 + j) * 2 + 1, ts[4 + j]);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ X, long ldx, const T* __restrict__ DY,
                                                           long lddy, const T* __restrict__ OUT, long ldo,
                                                           T* __restrict__ DX, long lddx,
                                                           const double* __restrict__ fsums, float eps,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ s12, long rows, int C) {
    const int nv = C >> 2;
    const long total = rows * nv;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int vi = (int)(idx % nv);
        const long r = idx / nv;
        float x[4], dy[4], o[4], dx[4];
        Vec4<T>::load(X + r * ldx + vi * 4, x);
        Vec4<T>::load(DY + r * lddy + vi * 4, dy);
        if (OUT) Vec4<T>::load(OUT + r * ldo + vi * 4, o);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = vi * 4 + j;
            const double m = fsums[c * 2] / (double)rows;
            const double var = fmax(fsums[c * 2 + 1] / (double)rows - m * m, 0.0);
            const float rstd = (float)(1.0 / sqrt(var + (double)eps));
            const float xh = (x[j] - (float)m) * rstd;
            const float g = (OUT && o[j] <= 0.f) ? 0.f : dy[j];
            dx[j] = gamma[c] * rstd * (g - s12[c * 2] / (float)rows - xh * s12[c * 2 + 1] / (float)rows);
        }
        Vec4<T>::store(DX + r * lddx + vi * 4, dx);
    }
}

// dgamma += s2, dbeta += s1
__global__ void bn_bwd_acc_kernel(const float* __restrict__ ws, float* __restrict__ dg, float* __restrict__ db, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        dg[c] += ws[c * 2 + 1];
        db[c] += ws[c * 2];
    }
}

// ---- adjoint of bilinear resampling --------------------------------------------------------------------------------
__device__ __forceinline__ void bl_src(int dst, int in, int out, int align, int& i0, int& i1, float& l1) {
    float src;
    if (align) {
        const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
        src = scale * (float)dst;
    } else {
        const float scale = (float)in / (float)out;
        src = scale * ((float)dst + 0.5f) - 0.5f;
        if (src < 0.f) src = 0.f;
    }
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = src - (float)i0;
}

// channels-last: DX[b][y][x][c] (f32 accumulation buffer, zero-filled by the caller) += w * DY[b][oy][ox][c]
template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const T* __restrict__ DY, long lddy, float* __restrict__ DX,
                                                           int B, int H, int Wd, int C, int Ho, int Wo, int align,
                                                           float mul) {
    const int nv = C >> 2;
    const long total = (long)B * Ho * Wo * nv;
    const IdxDiv D(total);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long pix = D.div(idx, nv), orow = D.div(pix, Wo), b = D.div(orow, Ho);
        const int vi = (int)(idx - pix * nv);
        const int ox = (int)(pix - orow * Wo), oy = (int)(orow - b * Ho);
        int y0, y1, x0, x1;
        float ly, lx;
        bl_src(oy, H, Ho, align, y0, y1, ly);
        bl_src(ox, Wd, Wo, align, x0, x1, lx);
        float g[4];
        Vec4<T>::load(DY + pix * lddy + vi * 4, g);
        float* base = DX + b * H * Wd * (long)C + vi * 4;
        const float w00 = (1.f - ly) * (1.f - lx) * mul, w01 = (1.f - ly) * lx * mul;
        const float w10 = ly * (1.f - lx) * mul, w11 = ly * lx * mul;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            atomicAdd(base + ((long)y0 * Wd + x0) * C + j, w00 * g[j]);
            atomicAdd(base + ((long)y0 * Wd + x1) * C + j, w01 * g[j]);
            atomicAdd(base + ((long)y1 * Wd + x0) * C + j, w10 * g[j]);
            atomicAdd(base + ((long)y1 * Wd + x1) * C + j, w11 * g[j]);
        }
    }
}

// planar f32 DY [B][C][Ho][Wo] -> channels-last f32 accumulation DX [B][H][W][ldx] channels xc..xc+C-1.
// Gather form: one wave per input element, lanes split the window of output pixels whose bilinear footprint touches it
// (x8 upsampling: ~28 x 28 candidates); no atomics (the scatter form serialised on 64-fold address collisions).
__global__ __launch_bounds__(256) void bilinear_planar_bwd_kernel(const float* __restrict__ DY, float* __restrict__ DX,
                                                                  long ldx, int xc, int B, int H, int Wd, int C, int Ho,
                                                                  int Wo, int align, float mul) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
    const long total = (long)B * C * H * Wd;
    // conservative window: the lower bound with the smaller of the two inverse scales, the upper bound with the larger
    const float ay = (float)Ho / (float)H, by = H > 1 ? (float)(Ho - 1) / (float)(H - 1) : ay;
    const float ax = (float)Wo / (float)Wd, bx = Wd > 1 ? (float)(Wo - 1) / (float)(Wd - 1) : ax;
    const float invy_lo = fminf(ay, by), invy = fmaxf(ay, by), invx_lo = fminf(ax, bx), invx = fmaxf(ax, bx);
    const IdxDiv D(total);
    for (long e = wave; e < total; e += nw) {
        const long row = D.div(e, Wd), bc = D.div(row, H), b = D.div(bc, C);
        const int x = (int)(e - row * Wd), y = (int)(row - bc * H);
        const int c = (int)(bc - b * C);
        const int oy_lo = max(0, (int)floorf((float)(y - 1) * invy_lo) - 1);
        const int oy_hi = min(Ho - 1, (int)ceilf((float)(y + 2) * invy) + 1);
        const int ox_lo = max(0, (int)floorf((float)(x - 1) * invx_lo) - 1);
        const int ox_hi = min(Wo - 1, (int)ceilf((float)(x + 2) * invx) + 1);
        const int ny = oy_hi - oy_lo + 1, nx = ox_hi - ox_lo + 1;
        const float* src = DY + ((b * C + c) * (long)Ho) * Wo;
        float acc = 0.f;
        for (int i = lane; i < ny * nx; i += 64) {
            const int oy = oy_lo + i / nx, ox = ox_lo + i % nx;
            int y0, y1, x0, x1;
            float ly, lx;
            bl_src(oy, H, Ho, align, y0, y1, ly);
            bl_src(ox, Wd, Wo, align, x0, x1, lx);
            const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
            const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
            const float w = wy * wx;
            if (w != 0.f) acc = fmaf(w, src[(long)oy * Wo + ox], acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) DX[((b * H + y) * (long)Wd + x) * ldx + xc + c] += acc * mul;
    }
}

}  // namespace

#define DISPATCH_T(dtype, ...)                          \
    do {                                                \
        if ((dtype) == EMIP_F32) {                      \
            typedef float T;                            \
            __VA_ARGS__;                                \
        } else {                                        \
            typedef bf16_t T;                           \
            __VA_ARGS__;                                \
        }                                               \
    } while (0)
#define REQ_DT(dtype) EMIP_REQUIRE((dtype) == EMIP_F32 || (dtype) == EMIP_BF16)

extern "C" int emip_softmax_rows(const void* X, void* Y, long rows, int L, long ld, float scale, const int* gid_q,
                                 const int* gid_k, long period, long nwin, int dtype, void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(X && Y && rows > 0 && L > 0 && L <= 65536 && ld >= L && ld <= 65536);
    EMIP_REQUIRE((gid_q == nullptr) == (gid_k == nullptr));
    if (gid_q) EMIP_REQUIRE(period > 0 && nwin > 0 && ld <= 2048);
    if (ld > 2048) {        // long rows: one workgroup per row
        DISPATCH_T(dtype, hipLaunchKernelGGL(softmax_rows_long_kernel<T>, dim3((unsigned)(rows < 65535 ? rows : 65535)),
                                             dim3(256), 0, (hipStream_t)stream, (const T*)X, (T*)Y, rows, L, ld, scale));
        return emip_launch_status();
    }
    if (!gid_q && ld == 128 && (((uintptr_t)X | (uintptr_t)Y) & 15) == 0) {
        DISPATCH_T(dtype, hipLaunchKernelGGL(softmax_rows128_kernel<T>, dim3(grid_for(rows, 16)), dim3(256), 0,
                                             (hipStream_t)stream, (const T*)X, (T*)Y, rows, L, scale));
        return emip_launch_status();
    }
    if (dtype == EMIP_BF16 && (ld & 7) == 0 && (((uintptr_t)X | (uintptr_t)Y) & 15) == 0) {
        const int nit = (int)((ld + 511) / 512);
#define EMIP_SMX(N)                                                                                                        \
    hipLaunchKernelGGL((softmax_rows_vec_kernel<N>), dim3(grid_for(rows, 4)), dim3(256), 0, (hipStream_t)stream,             \
                       (const bf16_t*)X, (bf16_t*)Y, rows, L, ld, scale, gid_q, gid_k, period, nwin)
        if (nit == 1) EMIP_SMX(1);
        else if (nit == 2) EMIP_SMX(2);
        else if (nit == 3) EMIP_SMX(3);
        else EMIP_SMX(4);
#undef EMIP_SMX
        return emip_launch_status();
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL(softmax_rows_kernel<T>, dim3(grid_for(rows, 4)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)X, (T*)Y, rows, L, ld, scale, gid_q, gid_k,
                                         period, nwin));
    return emip_launch_status();
}

extern "C" int emip_softmax_bwd_rows(const void* P, const void* DP, void* DS, long rows, int L, long ld, float scale,
                                     int dtype, void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(P && DP && DS && rows > 0 && L > 0 && L <= 65536 && ld >= L && ld <= 65536);
    if (ld > 2048) {
        DISPATCH_T(dtype, hipLaunchKernelGGL(softmax_bwd_rows_long_kernel<T>,
                                             dim3((unsigned)(rows < 65535 ? rows : 65535)), dim3(256), 0,
                                             (hipStream_t)stream, (const T*)P, (const T*)DP, (T*)DS, rows, L, ld, scale));
        return emip_launch_status();
    }
    if (ld == 128 && (((uintptr_t)P | (uintptr_t)DP | (uintptr_t)DS) & 15) == 0) {
        DISPATCH_T(dtype, hipLaunchKernelGGL(softmax_bwd_rows128_kernel<T>, dim3(grid_for(rows, 16)), dim3(256), 0,
                                             (hipStream_t)stream, (const T*)P, (const T*)DP, (T*)DS, rows, L, scale));
        return emip_launch_status();
    }
    if (dtype == EMIP_BF16 && (ld & 7) == 0 && (((uintptr_t)P | (uintptr_t)DP | (uintptr_t)DS) & 15) == 0) {
        const int nit = (int)((ld + 511) / 512);
#define EMIP_SMXB(N)                                                                                                       \
    hipLaunchKernelGGL((softmax_bwd_rows_vec_kernel<N>), dim3(grid_for(rows, 4)), dim3(256), 0, (hipStream_t)stream,         \
                       (const bf16_t*)P, (const bf16_t*)DP, (bf16_t*)DS, rows, L, ld, scale)
        if (nit == 1) EMIP_SMXB(1);
        else if (nit == 2) EMIP_SMXB(2);
        else if (nit == 3) EMIP_SMXB(3);
        else EMIP_SMXB(4);
#undef EMIP_SMXB
        return emip_launch_status();
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL(softmax_bwd_rows_kernel<T>, dim3(grid_for(rows, 4)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)P, (const T*)DP, (T*)DS, rows, L, ld, scale));
    return emip_launch_status();
}

extern "C" int emip_transpose_pad(const void* X, long ldx, long bsx, void* Y, long bsy, int batch, int R, int C,
                                  int Rpad, int dtype, void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(X && Y && batch > 0 && batch < 65536 && R > 0 && C > 0 && Rpad >= R && ldx >= C);
    dim3 grid((Rpad + 31) / 32, (C + 31) / 32, batch);
    EMIP_REQUIRE(grid.y < 65536);
    DISPATCH_T(dtype, hipLaunchKernelGGL(transpose_pad_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)X,
                                         ldx, bsx, (T*)Y, bsy, R, C, Rpad, 1, 0L));
    return emip_launch_status();
}

// batch = B * heads slices: input of (b, h) at b * bsx + h * hsx (C columns each), output [batch][C][Rpad] contiguous
extern "C" int emip_transpose_pad_heads(const void* X, long ldx, long bsx, long hsx, void* Y, int batch, int heads, int R,
                                        int C, int Rpad, int dtype, void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(X && Y && batch > 0 && batch < 65536 && heads >= 1 && batch % heads == 0 && R > 0 && C > 0 && Rpad >= R);
    dim3 grid((Rpad + 31) / 32, (C + 31) / 32, batch);
    EMIP_REQUIRE(grid.y < 65536);
    DISPATCH_T(dtype, hipLaunchKernelGGL(transpose_pad_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)X,
                                         ldx, bsx, (T*)Y, (long)C * Rpad, R, C, Rpad, heads, hsx));
    return emip_launch_status();
}

extern "C" int emip_gelu_bwd(const void* Z, long ldz, const void* DY, long lddy, void* DZ, long lddz, long M, int C,
                             int dtype, void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(Z && DY && DZ && M > 0 && C >= 4 && (C & 3) == 0 && (ldz & 3) == 0 && (lddy & 3) == 0 &&
                 (lddz & 3) == 0 && ldz >= C && lddy >= C && lddz >= C);
    const int vec = dtype == EMIP_F32 ? 4 : 8;
    if (C % vec == 0 && ldz % vec == 0 && lddy % vec == 0 && lddz % vec == 0 && ((((uintptr_t)Z) | ((uintptr_t)DY) |
                                                                                   ((uintptr_t)DZ)) & 15) == 0) {
        DISPATCH_T(dtype, hipLaunchKernelGGL(gelu_bwd_wide_kernel<T>, dim3(grid_for(M * (C / vec), 256)), dim3(256), 0,
                                             (hipStream_t)stream, (const T*)Z, ldz, (const T*)DY, lddy, (T*)DZ, lddz, M,
                                             C));
        return emip_launch_status();
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL(gelu_bwd_kernel<T>, dim3(grid_for(M * (C >> 2), 256)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)Z, ldz, (const T*)DY, lddy, (T*)DZ, lddz, M, C));
    return emip_launch_status();
}

// dW f32 [9][C] and db f32 [C] (may be NULL) are ACCUMULATED into
static int g_dww_chunks = 0;
#ifdef EMIP_TUNING
extern "C" int emip_debug_set_dww(int chunks) {
    g_dww_chunks = chunks;
    return EMIP_OK;
}
#endif

extern "C" int emip_dwconv3x3_wgrad(const void* X, long ldx, const void* DY, long lddy, float* dW, float* db, int B,
                                    int H, int Wd, int C, int dtype, void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(X && DY && dW && B > 0 && B < 65536 && H > 0 && Wd > 0 && C >= 4);
    const int vec = dtype == EMIP_F32 ? 4 : 8;
    EMIP_REQUIRE(C % vec == 0 && ldx % vec == 0 && lddy % vec == 0 && ldx >= C && lddy >= C);
    EMIP_REQUIRE((((uintptr_t)X) & 15) == 0 && (((uintptr_t)DY) & 15) == 0);
    // rows per workgroup: about 2048 workgroups in all (8 per CU), so that the per-workgroup tail (80 shuffle-reduced sums,
    // 640 atomics) is spread over as many pixels as the grid allows
    const int cgroups = (C + 63) / 64;
    int chunks = (int)((2048 + (long)cgroups * B - 1) / ((long)cgroups * B));
    if (g_dww_chunks > 0) chunks = g_dww_chunks;
    chunks = chunks < 1 ? 1 : (chunks > H ? H : chunks);
    const int rpb = (H + chunks - 1) / chunks;
    dim3 grid(cgroups * ((H + rpb - 1) / rpb), B);
    DISPATCH_T(dtype, hipLaunchKernelGGL(dwconv_wgrad_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)X,
                                         ldx, (const T*)DY, lddy, dW, db, B, H, Wd, C, rpb));
    return emip_launch_status();
}

// x: conv output (pre-BN) [rows][C]; out: the ReLU'd forward output (NULL when no ReLU followed); fsums: the forward's
// emip_chan_stats sums (groups = 1).  dgamma/dbeta (f32 [C]) are ACCUMULATED into.  ws: f32 [2*C] scratch.
extern "C" int emip_bn_train_bwd(const void* X, long ldx, const void* DY, long lddy, const void* OUT, long ldo, void* DX,
                                 long lddx, const double* fsums, const float* gamma, float* dgamma, float* dbeta,
                                 float* ws, long rows, int C, float eps, int dtype, void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(X && DY && DX && fsums && gamma && dgamma && dbeta && ws && rows > 0 && C >= 4 && C <= 1024 &&
                 (C & 3) == 0);
    EMIP_REQUIRE((ldx & 3) == 0 && (lddy & 3) == 0 && (lddx & 3) == 0 && ldx >= C && lddy >= C && lddx >= C);
    if (OUT) EMIP_REQUIRE((ldo & 3) == 0 && ldo >= C);
    hipStream_t s = (hipStream_t)stream;
    if (emip_zero_async(ws, sizeof(float) * 2 * C, s) != EMIP_OK) return EMIP_E_LAUNCH;
    const int rpb = 512;
    DISPATCH_T(dtype, hipLaunchKernelGGL(bn_bwd_stats_kernel<T>, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0,
                                         s, (const T*)X, ldx, (const T*)DY, lddy, (const T*)OUT, ldo, fsums, eps, ws,
                                         rows, C, rpb));
    DISPATCH_T(dtype, hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3(grid_for(rows * (C >> 2), 256)), dim3(256), 0, s,
                                         (const T*)X, ldx, (const T*)DY, lddy, (const T*)OUT, ldo, (T*)DX, lddx, fsums,
                                         eps, gamma, ws, rows, C));
    hipLaunchKernelGGL(bn_bwd_acc_kernel, dim3((C + 255) / 256), dim3(256), 0, s, ws, dgamma, dbeta, C);
    return emip_launch_status();
}

// DX: f32 [B][H][W][C] accumulation buffer (zero-filled by the caller or carrying other contributions)
extern "C" int emip_bilinear_bwd(const void* DY, long lddy, float* DX, int B, int H, int Wd, int C, int Ho, int Wo,
                                 int align_corners, float mul, int dtype, void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(DY && DX && B > 0 && H > 0 && Wd > 0 && Ho > 0 && Wo > 0 && C >= 4 && (C & 3) == 0 &&
                 (lddy & 3) == 0 && lddy >= C);
    const long total = (long)B * Ho * Wo * (C >> 2);
    DISPATCH_T(dtype, hipLaunchKernelGGL(bilinear_bwd_kernel<T>, dim3(grid_for(total, 256)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)DY, lddy, DX, B, H, Wd, C, Ho, Wo,
                                         align_corners, mul));
    return emip_launch_status();
}

extern "C" int emip_bilinear_planar_bwd(const float* DY, float* DX, long ldx, int xc, int B, int H, int Wd, int C,
                                        int Ho, int Wo, int align_corners, float mul, void* stream) {
    EMIP_REQUIRE(DY && DX && B > 0 && H > 0 && Wd > 0 && Ho > 0 && Wo > 0 && C >= 1 && xc >= 0 && ldx >= xc + C);
    const long total = (long)B * C * H * Wd;          // one wave per input element
    hipLaunchKernelGGL(bilinear_planar_bwd_kernel, dim3(grid_for(total, 4)), dim3(256), 0, (hipStream_t)stream, DY,
                       DX, ldx, xc, B, H, Wd, C, Ho, Wo, align_corners, mul);
    return emip_launch_status();
}

// ---- helpers for strided-conv input gradients ----------------------------------------------------------------------
namespace {
// Z[b][oy*s][ox*s][:] = DY[b][oy][ox][:], every other element of Z (size H x W) zero: the transposed conv of a
// strided conv is then the stride-1 conv of Z with the flipped, transposed weights.
template <typename T>
__global__ __launch_bounds__(256) void zero_insert_kernel(const T* __restrict__ DY, long lddy, T* __restrict__ Z, int B,
                                                          int Ho, int Wo, int H, int Wd, int C, int s) {
    const int nv = C >> 2;
    const long total = (long)B * H * Wd * nv;
    const IdxDiv D(total);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long pix = D.div(idx, nv), row = D.div(pix, Wd), b = D.div(row, H);
        const int vi = (int)(idx - pix * nv);
        const int x = (int)(pix - row * Wd), y = (int)(row - b * H);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (y % s == 0 && x % s == 0 && y / s < Ho && x / s < Wo)
            Vec4<T>::load(DY + ((b * Ho + y / s) * (long)Wo + x / s) * lddy + vi * 4, v);
        Vec4<T>::store(Z + pix * C + vi * 4, v);
    }
}
// non-overlapping patches (k == stride, pad 0): DX[b][oy*k+ky][ox*k+kx][ci] = P[(b,oy,ox)][(ky,kx,ci)]
template <typename T>
__global__ __launch_bounds__(256) void depatchify_kernel(const T* __restrict__ P, T* __restrict__ DX, int B, int Ho,
                                                         int Wo, int k, int C) {
    const int nv = C >> 2;
    const int H = Ho * k, Wd = Wo * k;
    const long total = (long)B * H * Wd * nv;
    const IdxDiv D(total);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long pix = D.div(idx, nv), row = D.div(pix, Wd), b = D.div(row, H);
        const int vi = (int)(idx - pix * nv);
        const int x = (int)(pix - row * Wd), y = (int)(row - b * H);
        const long m = (b * Ho + y / k) * Wo + x / k;
        const int tap = (y % k) * k + (x % k);
        float v[4];
        Vec4<T>::load(P + m * ((long)k * k * C) + (long)tap * C + vi * 4, v);
        Vec4<T>::store(DX + pix * C + vi * 4, v);
    }
}
}  // namespace

extern "C" int emip_zero_insert(const void* DY, long lddy, void* Z, int B, int Ho, int Wo, int H, int Wd, int C,
                                int stride, int dtype, void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(DY && Z && B > 0 && Ho > 0 && Wo > 0 && H >= (Ho - 1) * stride + 1 && Wd >= (Wo - 1) * stride + 1 &&
                 C >= 4 && (C & 3) == 0 && (lddy & 3) == 0 && lddy >= C && stride >= 1);
    const long total = (long)B * H * Wd * (C >> 2);
    DISPATCH_T(dtype, hipLaunchKernelGGL(zero_insert_kernel<T>, dim3(grid_for(total, 256)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)DY, lddy, (T*)Z, B, Ho, Wo, H, Wd, C, stride));
    return emip_launch_status();
}

extern "C" int emip_depatchify(const void* P, void* DX, int B, int Ho, int Wo, int k, int C, int dtype, void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(P && DX && B > 0 && Ho > 0 && Wo > 0 && k >= 1 && C >= 4 && (C & 3) == 0);
    const long total = (long)B * Ho * k * Wo * k * (C >> 2);
    DISPATCH_T(dtype, hipLaunchKernelGGL(depatchify_kernel<T>, dim3(grid_for(total, 256)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)P, (T*)DX, B, Ho, Wo, k, C));
    return emip_launch_status();
}

// ---- column sums (bias gradients): out[c] += sum_rows X[row][c], any C (multiple of 4) ----------------------------
namespace {
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, long ldx, float* __restrict__ out,
                                                     long rows, int C, int rows_per_block) {
    __shared__ float red[16][4][17];
    const int cq = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int cgroups = (C + 63) / 64;
    const int cg = blockIdx.x % cgroups;
    const long chunk = blockIdx.x / cgroups;
    const int c = cg * 64 + cq * 4;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        const long r0 = chunk * rows_per_block, r1 = min(rows, r0 + rows_per_block);
        for (long r = r0 + pl; r < r1; r += 16) {
            float v[4];
            Vec4<T>::load(X + r * ldx + c, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] += v[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[cq][j][pl] = s[j];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int q = threadIdx.x >> 2, j = threadIdx.x & 3;
        float t = 0.f;
#pragma unroll
        for (int l = 0; l < 16; ++l) t += red[q][j][l];
        const int ch = cg * 64 + threadIdx.x;
        if (ch < C) atomicAdd(out + ch, t);
    }
}
}  // namespace

// out f32 [C] is ACCUMULATED into
extern "C" int emip_colsum(const void* X, long ldx, float* out, long rows, int C, int dtype, void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(X && out && rows > 0 && C >= 4 && (C & 3) == 0 && (ldx & 3) == 0 && ldx >= C);
    const int rpb = 256;
    const long chunks = (rows + rpb - 1) / rpb;
    const long blocks = chunks * ((C + 63) / 64);
    EMIP_REQUIRE(blocks < 2147483647L);
    DISPATCH_T(dtype, hipLaunchKernelGGL(colsum_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                                         (const T*)X, ldx, out, rows, C, rpb));
    return emip_launch_status();
}

// ---- gated GELU (GDFN of the MDTA block): y = gelu(z[:, :Ch]) * z[:, Ch:2Ch] ----------------------------------------
namespace {
template <typename T>
__global__ __launch_bounds__(256) void gate_fwd_kernel(const T* __restrict__ Z, long ldz, T* __restrict__ Y, long ldy,
                                                       long M, int Ch, int Cpad) {
    const int nv = Cpad >> 2;
    const long total = M * nv;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % nv) * 4;
        const long r = idx / nv;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (c < Ch) {
            float a[4], b[4];
            Vec4<T>::load(Z + r * ldz + c, a);
            Vec4<T>::load(Z + r * ldz + Ch + c, b);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = gelu_erf(a[j]) * b[j];
        }
        Vec4<T>::store(Y + r * ldy + c, o);
    }
}
// dz[:, :Ch] = dy * z2 * gelu'(z1);  dz[:, Ch:] = dy * gelu(z1)
template <typename T>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const T* __restrict__ Z, long ldz, const T* __restrict__ DY,
                                                       long lddy, T* __restrict__ DZ, long lddz, long M, int Ch) {
    const int nv = Ch >> 2;
    const long total = M * nv;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % nv) * 4;
        const long r = idx / nv;
        float a[4], b[4], g[4], d1[4], d2[4];
        Vec4<T>::load(Z + r * ldz + c, a);
        Vec4<T>::load(Z + r * ldz + Ch + c, b);
        Vec4<T>::load(DY + r * lddy + c, g);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float cdf = 0.5f * (1.f + erff(a[j] * 0.70710678118654752440f));
            const float pdf = 0.3989422804014327f * expf(-0.5f * a[j] * a[j]);
            d1[j] = g[j] * b[j] * (cdf + a[j] * pdf);
            d2[j] = g[j] * a[j] * cdf;
        }
        Vec4<T>::store(DZ + r * lddz + c, d1);
        Vec4<T>::store(DZ + r * lddz + Ch + c, d2);
    }
}
// Y[r][c] = A[r][c] + s[(r / rows_per_group)][c] * B[r][c]      (column-scaled add, s f32)
template <typename T>
__global__ __launch_bounds__(256) void colscale_add_kernel(const T* __restrict__ A, long lda, const T* __restrict__ Bp,
                                                           long ldb, const float* __restrict__ S, long lds,
                                                           T* __restrict__ Y, long ldy, long M, int C,
                                                           long rows_per_group) {
    const int nv = C >> 2;
    const long total = M * nv;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % nv) * 4;
        const long r = idx / nv;
        const float* s = S + (r / rows_per_group) * lds + c;
        float a[4], b[4], o[4];
        Vec4<T>::load(A + r * lda + c, a);
        Vec4<T>::load(Bp + r * ldb + c, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = fmaf(s[j], b[j], a[j]);
        Vec4<T>::store(Y + r * ldy + c, o);
    }
}
}  // namespace

extern "C" int emip_gate_fwd(const void* Z, long ldz, void* Y, long ldy, long M, int Ch, int Cpad, int dtype,
                             void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(Z && Y && M > 0 && Ch >= 4 && (Ch & 3) == 0 && (Cpad & 3) == 0 && Cpad >= Ch && (ldz & 3) == 0 &&
                 (ldy & 3) == 0 && ldz >= 2 * Ch && ldy >= Cpad);
    DISPATCH_T(dtype, hipLaunchKernelGGL(gate_fwd_kernel<T>, dim3(grid_for(M * (Cpad >> 2), 256)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)Z, ldz, (T*)Y, ldy, M, Ch, Cpad));
    return emip_launch_status();
}

extern "C" int emip_gate_bwd(const void* Z, long ldz, const void* DY, long lddy, void* DZ, long lddz, long M, int Ch,
                             int dtype, void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(Z && DY && DZ && M > 0 && Ch >= 4 && (Ch & 3) == 0 && (ldz & 3) == 0 && (lddy & 3) == 0 &&
                 (lddz & 3) == 0 && ldz >= 2 * Ch && lddz >= 2 * Ch && lddy >= Ch);
    DISPATCH_T(dtype, hipLaunchKernelGGL(gate_bwd_kernel<T>, dim3(grid_for(M * (Ch >> 2), 256)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)Z, ldz, (const T*)DY, lddy, (T*)DZ, lddz, M, Ch));
    return emip_launch_status();
}

extern "C" int emip_colscale_add(const void* A, long lda, const void* Bp, long ldb, const float* S, long lds, void* Y,
                                 long ldy, long M, int C, long rows_per_group, int dtype, void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(A && Bp && S && Y && M > 0 && C >= 4 && (C & 3) == 0 && rows_per_group > 0 && (lda & 3) == 0 &&
                 (ldb & 3) == 0 && (ldy & 3) == 0 && lda >= C && ldb >= C && ldy >= C);
    DISPATCH_T(dtype, hipLaunchKernelGGL(colscale_add_kernel<T>, dim3(grid_for(M * (C >> 2), 256)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)A, lda, (const T*)Bp, ldb, S, lds, (T*)Y, ldy, M,
                                         C, rows_per_group));
    return emip_launch_status();
}

// ---- MDTA channel attention backward, small-matrix part ----------------------------------------------------------------
// per (b, head): inputs G (raw Gram f32 [64][64]), nq, nk (sums of squares f32 [64]), temperature, A (softmax, T),
// dA (f32 [64][64]).  outputs: dGraw (T [64][64]) = dGhat * tau / (|q||k|), dGrawT (its transpose),
// sq[c] = dnq[c] / |q_c| and sk[c] (f32, the column scales of the normalisation terms), dtau[head] += sum dGhat*Ghat/tau.
namespace {
template <typename T>
__global__ __launch_bounds__(64) void mdta_bwd_small_kernel(const float* __restrict__ G, const float* __restrict__ nq2,
                                                            const float* __restrict__ nk2,
                                                            const float* __restrict__ temperature,
                                                            const T* __restrict__ A, const float* __restrict__ dA,
                                                            T* __restrict__ dG, T* __restrict__ dGT,
                                                            float* __restrict__ sq, float* __restrict__ sk,
                                                            float* __restrict__ dtau, int heads) {
    __shared__ float sdg[64][65];    // dGhat * Ghat  (for the column sums) then dGraw
    const long bh = blockIdx.x;
    const int head = (int)(bh % heads);
    const int c1 = threadIdx.x;
    const float tau = temperature[head];
    const float qn = fmaxf(sqrtf(nq2[bh * 64 + c1]), 1e-12f);
    float a[64], dgh[64];
    float dot = 0.f;
#pragma unroll
    for (int c2 = 0; c2 < 64; ++c2) {
        a[c2] = to_f32<T>(A[(bh * 64 + c1) * 64 + c2]);
        dot += a[c2] * dA[(bh * 64 + c1) * 64 + c2];
    }
    float rowsum = 0.f, tsum = 0.f;
#pragma unroll
    for (int c2 = 0; c2 < 64; ++c2) {
        const float kn = fmaxf(sqrtf(nk2[bh * 64 + c2]), 1e-12f);
        dgh[c2] = a[c2] * (dA[(bh * 64 + c1) * 64 + c2] - dot);                 // d Ghat
        const float gn = G[(bh * 64 + c1) * 64 + c2] / (qn * kn);               // Ghat / tau
        const float ghat = gn * tau;
        rowsum += dgh[c2] * ghat;
        tsum += dgh[c2] * gn;
        sdg[c1][c2] = dgh[c2] * ghat;
        const float draw = dgh[c2] * tau / (qn * kn);
        dG[(bh * 64 + c1) * 64 + c2] = from_f32<T>(draw);
        dGT[(bh * 64 + c2) * 64 + c1] = from_f32<T>(draw);
    }
    // d|q_c1| = -rowsum / |q_c1|; the Q update is dnq * Q / |q| -> column scale sq = -rowsum / |q|^2 (0 when clamped)
    sq[bh * 64 + c1] = sqrtf(nq2[bh * 64 + c1]) > 1e-12f ? -rowsum / (qn * qn) : 0.f;
    atomicAdd(dtau + head, tsum);
    __syncthreads();
    float colsum = 0.f;
#pragma unroll
    for (int r = 0; r < 64; ++r) colsum += sdg[r][c1];
    const float kn1 = fmaxf(sqrtf(nk2[bh * 64 + c1]), 1e-12f);
    sk[bh * 64 + c1] = sqrtf(nk2[bh * 64 + c1]) > 1e-12f ? -colsum / (kn1 * kn1) : 0.f;
}
}  // namespace

extern "C" int emip_mdta_bwd_small(const float* G, const float* nq2, const float* nk2, const float* temperature,
                                   const void* A, const float* dA, void* dG, void* dGT, float* sq, float* sk,
                                   float* dtau, int B, int heads, int dtype, void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(G && nq2 && nk2 && temperature && A && dA && dG && dGT && sq && sk && dtau && B > 0 && heads > 0);
    const unsigned nbh = (unsigned)B * heads;
    DISPATCH_T(dtype, hipLaunchKernelGGL(mdta_bwd_small_kernel<T>, dim3(nbh), dim3(64), 0, (hipStream_t)stream, G, nq2,
                                         nk2, temperature, (const T*)A, dA, (T*)dG, (T*)dGT, sq, sk, dtau, heads));
    return emip_launch_status();
}
