// More backward pieces of the EMIP training step: row gather/scatter (window attention on dense batches), standalone
// activations, convex-upsampling backward, loss backward kernels (hybrid_e_loss, photometric SSIM + L1, flow warp).
#include "common.h"

namespace {

inline int grid_for(long n, int threads) {
    long b = (n + threads - 1) / threads;
    if (b > 4096) b = 4096;
    return (int)(b < 1 ? 1 : b);
}

// dst[(b*nwin + win)][t][:] = src[b][table[win][t]][:]   (scatter: the inverse copy)
template <typename T, bool SCATTER>
__global__ __launch_bounds__(256) void window_rows_kernel(const T* __restrict__ src, T* __restrict__ dst,
                                                          const int* __restrict__ table, int B, int nwin, int L,
                                                          int Lp, long n, int C, long ld_full, long ld_win) {
    const int nv = C >> 2;
    const long total = (long)B * nwin * L * nv;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int vi = (int)(idx % nv);
        long r = idx / nv;
        const int t = (int)(r % L);
        r /= L;
        const int win = (int)(r % nwin);
        const long b = r / nwin;
        const long full = (b * n + table[win * L + t]) * ld_full + vi * 4;
        const long wrow = ((b * nwin + win) * (long)Lp + t) * ld_win + vi * 4;
        float v[4];
        if (SCATTER) {
            Vec4<T>::load(src + wrow, v);
            Vec4<T>::store(dst + full, v);
        } else {
            Vec4<T>::load(src + full, v);
            Vec4<T>::store(dst + wrow, v);
        }
    }
}

// y = alpha * a + beta * b
template <typename T>
__global__ __launch_bounds__(256) void axpby_kernel(const T* __restrict__ A, long lda, const T* __restrict__ Bm, long ldb,
                                                    T* __restrict__ Y, long ldy, long M, int C, float alpha, float beta) {
    const int nv = C >> 2;
    const long total = M * nv;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % nv) * 4;
        const long r = idx / nv;
        float a[4], b[4];
        Vec4<T>::load(A + r * lda + c, a);
        Vec4<T>::load(Bm + r * ldb + c, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = alpha * a[j] + beta * b[j];
        Vec4<T>::store(Y + r * ldy + c, a);
    }
}

// y = act(x) (1 relu, 2 gelu)
template <typename T>
__global__ __launch_bounds__(256) void act_fwd_kernel(const T* __restrict__ X, long ldx, T* __restrict__ Y, long ldy,
                                                      long M, int C, int act) {
    const int nv = C >> 2;
    const long total = M * nv;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % nv) * 4;
        const long r = idx / nv;
        float v[4];
        Vec4<T>::load(X + r * ldx + c, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = act == EMIP_ACT_GELU ? gelu_erf(v[j]) : fmaxf(v[j], 0.f);
        Vec4<T>::store(Y + r * ldy + c, v);
    }
}
// dx = dy where y > 0
template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_kernel(const T* __restrict__ Yo, long ldy, const T* __restrict__ DY,
                                                       long lddy, T* __restrict__ DX, long lddx, long M, int C) {
    const int nv = C >> 2;
    const long total = M * nv;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % nv) * 4;
        const long r = idx / nv;
        float y[4], g[4];
        Vec4<T>::load(Yo + r * ldy + c, y);
        Vec4<T>::load(DY + r * lddy + c, g);
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = y[j] > 0.f ? g[j] : 0.f;
        Vec4<T>::store(DX + r * lddx + c, g);
    }
}

// convex upsampling backward (gmflow.py:64-77).  DY planar f32 [N][2][8H][8W]; logits T [N][H][W][ldl>=576];
// dlogits T same layout (576 channels written), dflow f32 [N][H][W][2] accumulated with atomics (zero-filled by caller)
template <typename T>
__global__ __launch_bounds__(256) void convex_up_bwd_kernel(const T* __restrict__ L, long ldl,
                                                            const float* __restrict__ F, const float* __restrict__ DY,
                                                            T* __restrict__ DL, long lddl, float* __restrict__ DF, int N,
                                                            int H, int Wd) {
    const long total = (long)N * H * Wd * 64;
    const IdxDiv D(total);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ij = (int)(idx & 63);
        const long pix = idx >> 6;
        const long row = D.div(pix, Wd), n = D.div(row, H);
        const int x = (int)(pix - row * Wd), y = (int)(row - n * H);
        const int i = ij >> 3, j = ij & 7;
        const long Ho = 8L * H, Wo = 8L * Wd;
        const long o = (n * 2 * Ho + (8L * y + i)) * Wo + 8L * x + j;
        const float gx = DY[o], gy = DY[o + Ho * Wo];
        float lg[9], p[9], u[9];
        float mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            lg[k] = to_f32<T>(L[pix * ldl + k * 64 + ij]);
            mx = fmaxf(mx, lg[k]);
        }
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            p[k] = expf(lg[k] - mx);
            den += p[k];
        }
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            p[k] /= den;
            const int iy = y + k / 3 - 1, ix = x + k % 3 - 1;
            u[k] = 0.f;
            const bool in = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)Wd;      // wave-uniform
            const long fo = ((n * H + iy) * Wd + ix) * 2;
            if (in) u[k] = 8.f * (gx * F[fo] + gy * F[fo + 1]);      // dL/dp_k
            // the 64 lanes of a wave are the 64 sub-pixels of ONE coarse pixel: reduce, then one atomic per tap
            const float sx = wave_sum(8.f * p[k] * gx), sy = wave_sum(8.f * p[k] * gy);
            if (in && ij == 0) {
                atomicAdd(DF + fo, sx);
                atomicAdd(DF + fo + 1, sy);
            }
            dot += p[k] * u[k];
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) DL[pix * lddl + k * 64 + ij] = from_f32<T>(p[k] * (u[k] - dot));
    }
}

// ---- hybrid_e_loss backward (loss/loss_pred.py:4-22).  acc: the forward's per-image sums [B][8]
// (0 sum sigma, 1 sum z, 2 sum sigma*z); slot 5 receives sum_i t_i of this pass.
__device__ __forceinline__ float hyb_t(float f, float g) {
    const float den = f * f + g * g + 1e-8f;
    const float E = (2.f * f * g + 1e-8f) / den;
    const float dE = (2.f * g * den - (2.f * f * g + 1e-8f) * 2.f * f) / (den * den);
    return 0.5f * (1.f + E) * dE;                                       // dQ/df
}
__global__ void hybrid_bwd_zero_kernel(double* __restrict__ acc, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) acc[b * 8 + 5] = 0.0;
}
__global__ __launch_bounds__(256) void hybrid_bwd_pass1_kernel(const float* __restrict__ P, const float* __restrict__ M,
                                                               double* __restrict__ acc, int HW) {
    __shared__ double sh[4];
    const int b = blockIdx.y;
    const float mp = (float)(acc[b * 8 + 0] / HW), mm = (float)(acc[b * 8 + 1] / HW);
    double s = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        const float sg = 1.f / (1.f + expf(-P[(long)b * HW + i]));
        s += hyb_t(sg - mp, M[(long)b * HW + i] - mm);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc + b * 8 + 5, sh[0] + sh[1] + sh[2] + sh[3]);
}
__global__ __launch_bounds__(256) void hybrid_bwd_pass2_kernel(const float* __restrict__ P, const float* __restrict__ M,
                                                               const double* __restrict__ acc,
                                                               const float* __restrict__ gout, float* __restrict__ DX,
                                                               int B, int HW) {
    const int b = blockIdx.y;
    const float mp = (float)(acc[b * 8 + 0] / HW), mm = (float)(acc[b * 8 + 1] / HW);
    const float tmean = (float)(acc[b * 8 + 5] / HW);
    const double I = acc[b * 8 + 2], U = acc[b * 8 + 0] + acc[b * 8 + 1];
    const float D = (float)(U - I + 1 + 1e-8), Nn = (float)(I + 1 + 1e-8);
    const float go = gout[0];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        const float x = P[(long)b * HW + i], z = M[(long)b * HW + i];
        const float sg = 1.f / (1.f + expf(-x));
        const float dbce = (sg - z) / ((float)B * HW);
        const float de = -(hyb_t(sg - mp, z - mm) - tmean) / HW;
        const float di = -(z * D - Nn * (1.f - z)) / (D * D);
        DX[(long)b * HW + i] = go * (dbce + (de + di) / (float)B * sg * (1.f - sg));
    }
}

// ---- photometric loss backward w.r.t. rec (loss_flow.py:35-49, loss_blocks.py:46-65) ---------------------------------
// pass 1: per 3x3 window (interior pixel) the partial derivatives of ssim_dist w.r.t. (mu_x, E[x^2], E[xy]) -> ABC planes
// pass 2: drec_q = m_q * sum over windows containing q of (A + 2 a_q B + b_q C) / 9 * cs + L1 term
__global__ __launch_bounds__(256) void photometric_bwd1_kernel(const float* __restrict__ IM, const float* __restrict__ REC,
                                                               const float* __restrict__ MK, float* __restrict__ ABC,
                                                               int B, int C, int H, int W) {
    const long hw = (long)H * W;
    const long total = (long)B * C * hw;
    const IdxDiv D(total);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long row = D.div(idx, W), bc = D.div(row, H), b = D.div(bc, C);
        const int x = (int)(idx - row * W), y = (int)(row - bc * H);
        float A = 0.f, Bq = 0.f, Cq = 0.f;
        if (y >= 1 && y < H - 1 && x >= 1 && x < W - 1) {
            const float* im = IM + bc * hw;
            const float* rc = REC + bc * hw;
            const float* mk = MK + b * hw;
            float sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const long o = (long)(y + dy) * W + x + dx;
                    const float mm = mk[o];
                    const float a = rc[o] * mm, bb = im[o] * mm;
                    sx += a; sy += bb; sxx += a * a; syy += bb * bb; sxy += a * bb;
                }
            const float mux = sx / 9.f, muy = sy / 9.f, ex2 = sxx / 9.f, ey2 = syy / 9.f, exy = sxy / 9.f;
            const float C1 = 1e-4f, C2 = 9e-4f;
            const float sgx = ex2 - mux * mux, sgy = ey2 - muy * muy, sgxy = exy - mux * muy;
            const float n1 = 2 * mux * muy + C1, n2 = 2 * sgxy + C2;
            const float d1 = mux * mux + muy * muy + C1, d2 = sgx + sgy + C2;
            const float S = (n1 * n2) / (d1 * d2);
            const float dist = (1.f - S) * 0.5f;
            if (dist > 0.f && dist < 1.f) {                       // inside the clamp
                // dS/dmux (through n1, n2 (sgxy), d1, d2 (sgx)), dS/dex2 (d2), dS/dexy (n2)
                const float dS_dn1 = n2 / (d1 * d2), dS_dn2 = n1 / (d1 * d2);
                const float dS_dd1 = -S / d1, dS_dd2 = -S / d2;
                const float dS_dmux = dS_dn1 * 2 * muy + dS_dn2 * (-2 * muy) + dS_dd1 * 2 * mux + dS_dd2 * (-2 * mux);
                A = -0.5f * dS_dmux;
                Bq = -0.5f * dS_dd2;                               // d sgx / d ex2 = 1
                Cq = -0.5f * dS_dn2 * 2.f;                         // d n2 / d exy = 2
            }
        }
        ABC[idx * 3] = A;
        ABC[idx * 3 + 1] = Bq;
        ABC[idx * 3 + 2] = Cq;
    }
}
__global__ __launch_bounds__(256) void photometric_bwd2_kernel(const float* __restrict__ IM, const float* __restrict__ REC,
                                                               const float* __restrict__ MK, const float* __restrict__ ABC,
                                                               const double* __restrict__ sums,
                                                               const float* __restrict__ gout, float weight,
                                                               float* __restrict__ DREC, int B, int C, int H, int W,
                                                               int accumulate) {
    const long hw = (long)H * W;
    const long total = (long)B * C * hw;
    const float inv_mmean = (float)((double)B * hw / sums[2]);
    const float cl1 = 0.15f / (float)total * inv_mmean * weight * gout[0];
    const float cs = 0.85f / (float)((double)B * C * (H - 2) * (W - 2)) * inv_mmean * weight * gout[0];
    const IdxDiv D(total);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long row = D.div(idx, W), bc = D.div(row, H), b = D.div(bc, C);
        const int x = (int)(idx - row * W), y = (int)(row - bc * H);
        const long o = (long)y * W + x;
        const float m = MK[b * hw + o];
        const float a = REC[bc * hw + o] * m, bb = IM[bc * hw + o] * m;
        float g = 0.f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = y + dy, xx = x + dx;                 // window centres that contain this pixel
                if (yy >= 1 && yy < H - 1 && xx >= 1 && xx < W - 1) {
                    const float* abc = ABC + (bc * hw + (long)yy * W + xx) * 3;
                    g += abc[0] + 2.f * a * abc[1] + bb * abc[2];
                }
            }
        const float d = REC[bc * hw + o] - IM[bc * hw + o];
        const float l1 = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * m * cl1;
        const float v = g / 9.f * m * cs + l1;
        DREC[idx] = accumulate ? DREC[idx] + v : v;
    }
}

// ---- flow warp backward w.r.t. the flow (loss/warp_utils.py:83-93; images need no gradient) -----------------------
__global__ __launch_bounds__(256) void flow_warp_bwd_kernel(const float* __restrict__ X, const float* __restrict__ F,
                                                            const float* __restrict__ DY, float* __restrict__ DF, int B,
                                                            int C, int H, int W) {
    const long hw = (long)H * W;
    const long total = (long)B * hw;
    const IdxDiv D(total);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long row = D.div(idx, W), b = D.div(row, H);
        const int x = (int)(idx - row * W), y = (int)(row - b * H);
        const float vx = (float)x + F[(b * 2) * hw + (long)y * W + x];
        const float vy = (float)y + F[(b * 2 + 1) * hw + (long)y * W + x];
        const float gx = 2.0f * vx / (float)(W - 1) - 1.0f, gy = 2.0f * vy / (float)(H - 1) - 1.0f;
        float ix = ((gx + 1.f) / 2.f) * (float)(W - 1), iy = ((gy + 1.f) / 2.f) * (float)(H - 1);
        // border clip: zero gradient outside [0, size-1] (ATen clip_coordinates_set_grad)
        const float mulx = (ix <= 0.f || ix >= (float)(W - 1)) ? 0.f : 1.f;
        const float muly = (iy <= 0.f || iy >= (float)(H - 1)) ? 0.f : 1.f;
        ix = fminf((float)(W - 1), fmaxf(ix, 0.f));
        iy = fminf((float)(H - 1), fmaxf(iy, 0.f));
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
        const float wx1 = ix - fx0, wy1 = iy - fy0, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        const bool xin = x1 < W, yin = y1 < H;
        float dix = 0.f, diy = 0.f;
        for (int c = 0; c < C; ++c) {
            const float* p = X + (b * C + c) * hw;
            const float g = DY[(b * C + c) * hw + (long)y * W + x];
            const float nw = p[(long)y0 * W + x0];
            const float ne = xin ? p[(long)y0 * W + x1] : 0.f;
            const float sw = yin ? p[(long)y1 * W + x0] : 0.f;
            const float se = (xin && yin) ? p[(long)y1 * W + x1] : 0.f;
            dix += g * ((ne - nw) * wy0 + (se - sw) * wy1);
            diy += g * ((sw - nw) * wx0 + (se - ne) * wx1);
        }
        DF[(b * 2) * hw + (long)y * W + x] = dix * mulx;
        DF[(b * 2 + 1) * hw + (long)y * W + x] = diy * muly;
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

extern "C" int emip_window_rows(const void* src, void* dst, const int* table, int B, int nwin, int L, int Lp, long n,
                                int C, long ld_full, long ld_win, int scatter, int dtype, void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(src && dst && table && B > 0 && nwin > 0 && L > 0 && Lp >= L && n >= L && C >= 4 && (C & 3) == 0 &&
                 (ld_full & 3) == 0 && (ld_win & 3) == 0 && ld_full >= C && ld_win >= C);
    const long total = (long)B * nwin * L * (C >> 2);
    if (scatter)
        DISPATCH_T(dtype, hipLaunchKernelGGL((window_rows_kernel<T, true>), dim3(grid_for(total, 256)), dim3(256), 0,
                                             (hipStream_t)stream, (const T*)src, (T*)dst, table, B, nwin, L, Lp, n, C,
                                             ld_full, ld_win));
    else
        DISPATCH_T(dtype, hipLaunchKernelGGL((window_rows_kernel<T, false>), dim3(grid_for(total, 256)), dim3(256), 0,
                                             (hipStream_t)stream, (const T*)src, (T*)dst, table, B, nwin, L, Lp, n, C,
                                             ld_full, ld_win));
    return emip_launch_status();
}

extern "C" int emip_axpby(const void* A, long lda, const void* B, long ldb, void* Y, long ldy, long M, int C, float alpha,
                          float beta, int dtype, void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(A && B && Y && M > 0 && C >= 4 && (C & 3) == 0 && (lda & 3) == 0 && (ldb & 3) == 0 && (ldy & 3) == 0 &&
                 lda >= C && ldb >= C && ldy >= C);
    DISPATCH_T(dtype, hipLaunchKernelGGL(axpby_kernel<T>, dim3(grid_for(M * (C >> 2), 256)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)A, lda, (const T*)B, ldb, (T*)Y, ldy, M, C, alpha,
                                         beta));
    return emip_launch_status();
}

extern "C" int emip_act_fwd(const void* X, long ldx, void* Y, long ldy, long M, int C, int act, int dtype, void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(X && Y && M > 0 && C >= 4 && (C & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0 && ldx >= C &&
                 ldy >= C && (act == EMIP_ACT_RELU || act == EMIP_ACT_GELU));
    DISPATCH_T(dtype, hipLaunchKernelGGL(act_fwd_kernel<T>, dim3(grid_for(M * (C >> 2), 256)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)X, ldx, (T*)Y, ldy, M, C, act));
    return emip_launch_status();
}

extern "C" int emip_relu_bwd(const void* Yo, long ldy, const void* DY, long lddy, void* DX, long lddx, long M, int C,
                             int dtype, void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(Yo && DY && DX && M > 0 && C >= 4 && (C & 3) == 0 && (ldy & 3) == 0 && (lddy & 3) == 0 &&
                 (lddx & 3) == 0 && ldy >= C && lddy >= C && lddx >= C);
    DISPATCH_T(dtype, hipLaunchKernelGGL(relu_bwd_kernel<T>, dim3(grid_for(M * (C >> 2), 256)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)Yo, ldy, (const T*)DY, lddy, (T*)DX, lddx, M, C));
    return emip_launch_status();
}

// dflow (f32 [N][H][W][2]) is zero-filled here
extern "C" int emip_convex_upsample_bwd(const void* logits, long ldl, const float* flow, const float* dY, void* dlogits,
                                        long lddl, float* dflow, int N, int H, int Wd, int dtype, void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(logits && flow && dY && dlogits && dflow && N > 0 && H > 0 && Wd > 0 && ldl >= 576 && lddl >= 576);
    hipStream_t s = (hipStream_t)stream;
    if (emip_zero_async(dflow, sizeof(float) * 2 * (size_t)N * H * Wd, s) != EMIP_OK) return EMIP_E_LAUNCH;
    const long total = (long)N * H * Wd * 64;
    DISPATCH_T(dtype, hipLaunchKernelGGL(convex_up_bwd_kernel<T>, dim3(grid_for(total, 256)), dim3(256), 0, s,
                                         (const T*)logits, ldl, flow, dY, (T*)dlogits, lddl, dflow, N, H, Wd));
    return emip_launch_status();
}

// ws: the f64 [B][8] scratch FILLED BY emip_hybrid_e_loss on the same inputs; gout: f32 [1] upstream gradient
extern "C" int emip_hybrid_e_loss_bwd(const float* pred, const float* mask, double* ws, const float* gout, float* dpred,
                                      int B, int H, int W, void* stream) {
    EMIP_REQUIRE(pred && mask && ws && gout && dpred && B > 0 && B < 65536 && H > 0 && W > 0);
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W;
    dim3 grid((HW + 256 * 8 - 1) / (256 * 8), B);
    hipLaunchKernelGGL(hybrid_bwd_zero_kernel, dim3((B + 63) / 64), dim3(64), 0, s, ws, B);   // repeated backward passes
    hipLaunchKernelGGL(hybrid_bwd_pass1_kernel, grid, dim3(256), 0, s, pred, mask, ws, HW);
    hipLaunchKernelGGL(hybrid_bwd_pass2_kernel, grid, dim3(256), 0, s, pred, mask, ws, gout, dpred, B, HW);
    return emip_launch_status();
}

// sums: the f64 [4] scratch FILLED BY emip_photometric_loss on the same inputs; abc: f32 [B*C*H*W*3] scratch;
// drec (+)= gout * weight * d loss / d rec
extern "C" int emip_photometric_loss_bwd(const float* im, const float* rec, const float* mask, const double* sums,
                                         float* abc, const float* gout, float* drec, int B, int C, int H, int W,
                                         float weight, int accumulate, void* stream) {
    EMIP_REQUIRE(im && rec && mask && sums && abc && gout && drec && B > 0 && C > 0 && H > 2 && W > 2);
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * C * H * W;
    hipLaunchKernelGGL(photometric_bwd1_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, im, rec, mask, abc, B, C, H,
                       W);
    hipLaunchKernelGGL(photometric_bwd2_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, im, rec, mask, abc, sums,
                       gout, weight, drec, B, C, H, W, accumulate);
    return emip_launch_status();
}

extern "C" int emip_flow_warp_bwd(const float* X, const float* flow, const float* dY, float* dflow, int B, int C, int H,
                                  int W, void* stream) {
    EMIP_REQUIRE(X && flow && dY && dflow && B > 0 && C > 0 && H > 1 && W > 1);
    hipLaunchKernelGGL(flow_warp_bwd_kernel, dim3(grid_for((long)B * H * W, 256)), dim3(256), 0, (hipStream_t)stream, X,
                       flow, dY, dflow, B, C, H, W);
    return emip_launch_status();
}
