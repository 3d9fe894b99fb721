// HBM-bound kernels of the EMIP path: normalisations, depthwise 3x3 convs,
// bilinear resampling, elementwise products/adds, layout conversion, convex
// flow upsampling.  All of them move 8-16 bytes per lane per access over
// channels-last tensors (channels are the fastest dimension, so a wavefront
// touches whole contiguous rows).
#include "common.h"

extern "C" int emip_dwconv3x3_dual(const void*, long, void*, long, void*, long, const float*, const float*, int, int, int, int,
                                   int, int, void*);

namespace {

constexpr int kMaxBlocks = 256 * 16;  // memory-bound grids: cap and grid-stride

inline int grid_for(long work_items, int threads) {
    long b = (work_items + threads - 1) / threads;
    if (b > kMaxBlocks) b = kMaxBlocks;
    if (b < 1) b = 1;
    return (int)b;
}

// ----------------------------------------------------------------------------
// LayerNorm over the channel dimension (rows of C <= 1024, C % 4 == 0).
// A row is spread over LPR lanes (power of two <= 64), 64/LPR rows per wave.
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ X, long ldx, T* __restrict__ Y, long ldy,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const T* R, long ldr,
                                                        float* __restrict__ out_stats, long M, int C, float eps,
                                                        int lpr_log2) {
    const int LPR = 1 << lpr_log2;
    const int lane = threadIdx.x & 63;
    const int sub = lane & (LPR - 1);
    const int rows_per_wave = 64 >> lpr_log2;
    const long wave_global = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;
    const int nv = C >> 2;
    for (long row0 = wave_global * rows_per_wave; row0 < M; row0 += nwaves * rows_per_wave) {
        const long row = row0 + (lane >> lpr_log2);
        const bool ok = row < M;
        float v[4][4];
        float s = 0.f;
        const long rowc = ok ? row : M - 1;   // clamped: loads stay unconditional, results are masked
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int vi = sub + i * LPR;
            if (i * LPR < nv) {               // wave-uniform
                Vec4<T>::load(X + rowc * ldx + min(vi, nv - 1) * 4, v[i]);
                if (!(ok && vi < nv)) v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f;
                s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
            } else {
                v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f;
            }
        }
        for (int o = LPR >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int vi = sub + i * LPR;
            if (vi < nv) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = v[i][j] - mean;
                    q += d * d;
                }
            }
        }
        for (int o = LPR >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = rsqrtf(q / (float)C + eps);
        float os1 = 0.f, os2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int vi = sub + i * LPR;
            if (ok && vi < nv) {
                float o4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o4[j] = (v[i][j] - mean) * rstd * gamma[vi * 4 + j] + beta[vi * 4 + j];
                if (R) {                      // wave-uniform: fused residual add (R may alias Y)
                    float r4[4];
                    Vec4<T>::load(R + row * ldr + vi * 4, r4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) o4[j] += r4[j];
                }
                Vec4<T>::store(Y + row * ldy + vi * 4, o4);
                if (out_stats) {              // sums of the STORED (rounded) row: input of a LayerNorm folded into its consumer
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float q = to_f32<T>(from_f32<T>(o4[j]));
                        os1 += q;
                        os2 += q * q;
                    }
                }
            }
        }
        if (out_stats) {                      // wave-uniform
            for (int o = LPR >> 1; o > 0; o >>= 1) {
                os1 += __shfl_xor(os1, o);
                os2 += __shfl_xor(os2, o);
            }
            if (ok && sub == 0) {
                out_stats[2 * row] = os1;
                out_stats[2 * row + 1] = os2;
            }
        }
    }
}

// ----------------------------------------------------------------------------
// Rows of an f32 accumulation buffer (split-K partial sums, complete) -> storage type, plus the (sum, sum of squares) of
// the stored rows for a LayerNorm folded into the consumer.  Same row / lane layout as layernorm_kernel.
template <typename T>
__global__ __launch_bounds__(256) void rows_finalize_kernel(const float* __restrict__ A, long lda, T* __restrict__ Y,
                                                            long ldy, float* __restrict__ out_stats, long M, int C,
                                                            int lpr_log2) {
    const int LPR = 1 << lpr_log2;
    const int lane = threadIdx.x & 63;
    const int sub = lane & (LPR - 1);
    const int rows_per_wave = 64 >> lpr_log2;
    const long wave_global = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;
    const int nv = C >> 2;
    for (long row0 = wave_global * rows_per_wave; row0 < M; row0 += nwaves * rows_per_wave) {
        const long row = row0 + (lane >> lpr_log2);
        const bool ok = row < M;
        float s1 = 0.f, s2 = 0.f;
        for (int vi = sub; vi < nv; vi += LPR) {
            if (ok) {
                const float4 t = *reinterpret_cast<const float4*>(A + row * lda + vi * 4);
                float v[4] = {t.x, t.y, t.z, t.w};
                Vec4<T>::store(Y + row * ldy + vi * 4, v);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float q = to_f32<T>(from_f32<T>(v[j]));
                    s1 += q;
                    s2 += q * q;
                }
            }
        }
        if (out_stats) {
            for (int o = LPR >> 1; o > 0; o >>= 1) {
                s1 += __shfl_xor(s1, o);
                s2 += __shfl_xor(s2, o);
            }
            if (ok && sub == 0) {
                out_stats[2 * row] = s1;
                out_stats[2 * row + 1] = s2;
            }
        }
    }
}

// ----------------------------------------------------------------------------
// LayerNorm backward (rows of C channels).  mean / rstd are recomputed from x (cheaper than saving them):
//   xh = (x - mean) * rstd,  g = dy * gamma,  dx = rstd * (g - mean_c(g) - xh * mean_c(g * xh))
//   dgamma[c] += sum_rows dy * xh,  dbeta[c] += sum_rows dy      (f32 atomics, one add per channel per workgroup)
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ X, long ldx,
                                                            const T* __restrict__ DY, long lddy,
                                                            T* __restrict__ DX, long lddx,
                                                            const float* __restrict__ gamma,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            long M, int C, float eps, int lpr_log2, int nparts,
                                                            long part_stride, const T* __restrict__ DR, long lddr) {
    extern __shared__ float red[];   // [2][C]
    const int LPR = 1 << lpr_log2;
    const int lane = threadIdx.x & 63;
    const int sub = lane & (LPR - 1);
    const int rows_per_wave = 64 >> lpr_log2;
    const long wave_global = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;
    const int nv = C >> 2;
    for (int i = threadIdx.x; i < 2 * C; i += 256) red[i] = 0.f;
    __syncthreads();
    float ag[4][4], ab[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) ag[i][j] = ab[i][j] = 0.f;
    for (long row0 = wave_global * rows_per_wave; row0 < M; row0 += nwaves * rows_per_wave) {
        const long row = row0 + (lane >> lpr_log2);
        const bool ok = row < M;
        const long rowc = ok ? row : M - 1;
        float x[4][4], dy[4][4];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int vi = sub + i * LPR;
            if (i * LPR < nv) {
                const int vc = min(vi, nv - 1);
                Vec4<T>::load(X + rowc * ldx + vc * 4, x[i]);
                Vec4<T>::load(DY + rowc * lddy + vc * 4, dy[i]);
                if (!(ok && vi < nv)) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[i][j] = dy[i][j] = 0.f;
                }
                s += x[i][0] + x[i][1] + x[i][2] + x[i][3];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) x[i][j] = dy[i][j] = 0.f;
            }
        }
        for (int o = LPR >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s / (float)C;
        float qv = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int vi = sub + i * LPR;
            if (vi < nv) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = x[i][j] - mean;
                    qv += d * d;
                }
            }
        }
        for (int o = LPR >> 1; o > 0; o >>= 1) qv += __shfl_xor(qv, o);
        const float rstd = rsqrtf(qv / (float)C + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int vi = sub + i * LPR;
            if (vi < nv) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = (x[i][j] - mean) * rstd;
                    const float g = dy[i][j] * gamma[vi * 4 + j];
                    x[i][j] = xh;          // keep xh
                    sg += g;
                    sgx += g * xh;
                    ag[i][j] += dy[i][j] * xh;
                    ab[i][j] += dy[i][j];
                }
            }
        }
        for (int o = LPR >> 1; o > 0; o >>= 1) {
            sg += __shfl_xor(sg, o);
            sgx += __shfl_xor(sgx, o);
        }
        const float mg = sg / (float)C, mgx = sgx / (float)C;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int vi = sub + i * LPR;
            if (ok && vi < nv) {
                float o4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    o4[j] = rstd * (dy[i][j] * gamma[vi * 4 + j] - mg - x[i][j] * mgx);
                if (DR) {                     // + the gradient arriving over the skip path of the residual block
                    float r4[4];
                    Vec4<T>::load(DR + row * lddr + vi * 4, r4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) o4[j] += r4[j];
                }
                Vec4<T>::store(DX + row * lddx + vi * 4, o4);
            }
        }
    }
    // combine the workgroup's partial dgamma / dbeta in LDS, then one atomic per channel
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int vi = sub + i * LPR;
        if (vi < nv) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                atomicAdd(&red[vi * 4 + j], ag[i][j]);
                atomicAdd(&red[C + vi * 4 + j], ab[i][j]);
            }
        }
    }
    __syncthreads();
    // workgroups are spread over `nparts` partial accumulators (the caller sums them): 1/nparts of the contention
    const long po = (long)(blockIdx.x % nparts) * part_stride;
    for (int i = threadIdx.x; i < C; i += 256) {
        atomicAdd(dgamma + po + i, red[i]);
        atomicAdd(dbeta + po + i, red[C + i]);
    }
}

// LayerNorm backward, the wide variant used whenever C % 8 == 0: 16-byte loads, U row groups in flight per wave, all four
// row sums (x, x^2, g, g x with g = dy gamma) from ONE pass and one joint shuffle reduction
//   mean = Sx / C,  var = Sx2 / C - mean^2,  mean_c(g xh) = rstd (Sgx - mean Sg) / C
// and 1024-thread workgroups, at most one per CU: measured on MI355X, same-address f32 atomics retire at ~25 ns each
// whatever the channel count (tools/atomic_bench.hip), so the dgamma / dbeta tail costs 25 ns x the number of workgroups;
// 256 large workgroups keep it at ~7 us where 1024 small ones paid 26 us.
template <typename T>
__device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <>
__device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <>
__device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&v)[8]) {
    bf16x4 a, b;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        a[k] = (bf16_t)v[k];
        b[k] = (bf16_t)v[4 + k];
    }
    *reinterpret_cast<bf16x4*>(p) = a;
    *reinterpret_cast<bf16x4*>(p + 4) = b;
}

// eight channels as loaded (bf16 stays packed in 4 registers until the row's turn comes)
template <typename T>
struct Raw8;
template <>
struct Raw8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) {
        a = *reinterpret_cast<const float4*>(p);
        b = *reinterpret_cast<const float4*>(p + 4);
    }
    __device__ __forceinline__ void unpack(float (&v)[8], float keep) const {
        v[0] = a.x * keep; v[1] = a.y * keep; v[2] = a.z * keep; v[3] = a.w * keep;
        v[4] = b.x * keep; v[5] = b.y * keep; v[6] = b.z * keep; v[7] = b.w * keep;
    }
};
template <>
struct Raw8<bf16_t> {
    uint4 t;
    __device__ __forceinline__ void load(const bf16_t* p) { t = *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ void unpack(float (&v)[8], float keep) const {
        const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[2 * k] = __uint_as_float(w[k] << 16) * keep;
            v[2 * k + 1] = __uint_as_float(w[k] & 0xFFFF0000u) * keep;
        }
    }
};

// Forward LayerNorm for bf16 rows, eight lanes per row and VPL 16-byte vectors per lane (vector v of a row in lane v % 8):
// every lane works at C = 64 / 128 / 320 / 512, a row statistic is three shuffle levels, a wave instruction reads whole
// 128-byte segments of 8 rows.  Same arithmetic as layernorm_kernel (mean, then the sum of squared deviations).
template <int VPL>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const bf16_t* __restrict__ X, long ldx, bf16_t* __restrict__ Y, long ldy,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const bf16_t* R, long ldr, float* __restrict__ out_stats, long M,
                                                             int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int sub = lane & 7, rsub = lane >> 3;
    const int nv8 = C >> 3;
    const long wave_global = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long stride = (long)gridDim.x * 32;
    bool act[VPL];
    int vcl[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        act[i] = sub + 8 * i < nv8;
        vcl[i] = min(sub + 8 * i, nv8 - 1) * 8;
    }
    const float invC = 1.f / (float)C;
    for (long row0 = wave_global * 8; row0 < M; row0 += stride) {
        const long row = row0 + rsub;
        const bool ok = row < M;
        const long rowc = ok ? row : M - 1;
        Raw8<bf16_t> rx[VPL], rr[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            rx[i].load(X + rowc * ldx + vcl[i]);
            if (R) rr[i].load(R + rowc * ldr + vcl[i]);
        }
        float x[VPL][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            rx[i].unpack(x[i], (ok && act[i]) ? 1.f : 0.f);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += x[i][j];
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s * invC;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i)
            if (act[i]) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = x[i][j] - mean;
                    q += d * d;
                }
            }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = rsqrtf(q * invC + eps);
        float os1 = 0.f, os2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            float o8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o8[j] = (x[i][j] - mean) * rstd * gamma[vcl[i] + j] + beta[vcl[i] + j];
            if (R) {
                float r8[8];
                rr[i].unpack(r8, 1.f);
#pragma unroll
                for (int j = 0; j < 8; ++j) o8[j] += r8[j];
            }
            if (ok && act[i]) {
                store8<bf16_t>(Y + row * ldy + vcl[i], o8);
                if (out_stats) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float v = (float)(bf16_t)o8[j];
                        os1 += v;
                        os2 += v * v;
                    }
                }
            }
        }
        if (out_stats) {
#pragma unroll
            for (int o = 4; o > 0; o >>= 1) {
                os1 += __shfl_xor(os1, o);
                os2 += __shfl_xor(os2, o);
            }
            if (ok && sub == 0) {
                out_stats[2 * row] = os1;
                out_stats[2 * row + 1] = os2;
            }
        }
    }
}

template <typename T, int U>
__global__ __launch_bounds__(1024) void layernorm_bwd_wide_kernel(const T* __restrict__ X, long ldx,
                                                                  const T* __restrict__ DY, long lddy,
                                                                  T* __restrict__ DX, long lddx,
                                                                  const float* __restrict__ gamma,
                                                                  float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                  long M, int C, float eps, int lpr_log2, int nparts,
                                                                  long part_stride, const T* __restrict__ DR, long lddr) {
    extern __shared__ float red[];   // [2][C]
    const int LPR = 1 << lpr_log2;                       // lanes per row, two 8-channel vectors per lane
    const int lane = threadIdx.x & 63;
    const int sub = lane & (LPR - 1), rsub = lane >> lpr_log2;
    const int rpw = 64 >> lpr_log2;                      // rows per wave and group
    const int nv8 = C >> 3;
    const int nwave = blockDim.x >> 6;
    const long wave_global = (long)blockIdx.x * nwave + (threadIdx.x >> 6);
    const long stride = (long)gridDim.x * nwave * rpw * U;
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) red[i] = 0.f;
    __syncthreads();
    float gam[2][8], ag[2][8], ab[2][8];
    bool act[2];
    int vcl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int vi = sub + i * LPR;
        act[i] = vi < nv8;
        vcl[i] = min(vi, nv8 - 1) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            gam[i][j] = act[i] ? gamma[vcl[i] + j] : 0.f;
            ag[i][j] = ab[i][j] = 0.f;
        }
    }
    const float invC = 1.f / (float)C;
    for (long row0 = wave_global * rpw * U; row0 < M; row0 += stride) {
        Raw8<T> rx[U][2], rdy[U][2], rdr[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long row = row0 + u * rpw + rsub;
            const long rowc = row < M ? row : M - 1;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                rx[u][i].load(X + rowc * ldx + vcl[i]);
                rdy[u][i].load(DY + rowc * lddy + vcl[i]);
                if (DR) rdr[u][i].load(DR + rowc * lddr + vcl[i]);        // workgroup-uniform
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long row = row0 + u * rpw + rsub;
            float x[2][8], dy[2][8];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float keep = (row < M && act[i]) ? 1.f : 0.f;       // rows past the end / lanes past C contribute zeros
                rx[u][i].unpack(x[i], keep);
                rdy[u][i].unpack(dy[i], keep);
            }
            float s = 0.f, s2 = 0.f, sg = 0.f, sgx = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xv = x[i][j], g = dy[i][j] * gam[i][j];
                    s += xv;
                    s2 += xv * xv;
                    sg += g;
                    sgx += g * xv;
                }
            for (int o = LPR >> 1; o > 0; o >>= 1) {
                s += __shfl_xor(s, o);
                s2 += __shfl_xor(s2, o);
                sg += __shfl_xor(sg, o);
                sgx += __shfl_xor(sgx, o);
            }
            const float mean = s * invC;
            const float rstd = rsqrtf(fmaxf(s2 * invC - mean * mean, 0.f) + eps);
            const float mg = sg * invC, mgx = rstd * (sgx - mean * sg) * invC;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float o8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = (x[i][j] - mean) * rstd;       // padded lanes: dy = 0, so nothing is accumulated
                    ag[i][j] += dy[i][j] * xh;
                    ab[i][j] += dy[i][j];
                    o8[j] = rstd * (dy[i][j] * gam[i][j] - mg - xh * mgx);
                }
                if (DR) {                     // + the gradient arriving over the skip path of the residual block
                    float r8[8];
                    rdr[u][i].unpack(r8, 1.f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) o8[j] += r8[j];
                }
                if (row < M && act[i]) store8<T>(DX + row * lddx + vcl[i], o8);
            }
        }
    }
    // the wave's row groups hold partial dgamma / dbeta for the same channels: combine them with shuffles, then one LDS
    // atomic per channel and wave, then one global atomic per channel and workgroup
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = ag[i][j], b = ab[i][j];
            for (int o = 32; o >= LPR; o >>= 1) {
                a += __shfl_xor(a, o);
                b += __shfl_xor(b, o);
            }
            if (rsub == 0 && act[i]) {
                atomicAdd(&red[vcl[i] + j], a);
                atomicAdd(&red[C + vcl[i] + j], b);
            }
        }
    __syncthreads();
    const long po = (long)(blockIdx.x % nparts) * part_stride;
    for (int i = threadIdx.x; i < C; i += blockDim.x) {
        atomicAdd(dgamma + po + i, red[i]);
        atomicAdd(dbeta + po + i, red[C + i]);
    }
}

// ----------------------------------------------------------------------------
// Depthwise 3x3, stride 1, zero pad 1, channels-last.  Weights [9][C] f32.
// GATED: X has 2*Ch channels, Y[c] = gelu(dw(X)[c]) * dw(X)[Ch + c] for c < Ch and 0 for
// Ch <= c < ldy-padding (so that a following GEMM can read a 8-aligned K).
template <typename T, bool GATED>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const T* __restrict__ X, long ldx, T* __restrict__ Y, long ldy,
                                                        const float* __restrict__ Wt, const float* __restrict__ bias,
                                                        int B, int H, int Wd, int C, int Cout_pad, int act) {
    const int nvo = (GATED ? Cout_pad : C) >> 2;
    const long total = (long)B * H * Wd * nvo;
    const int Ch = GATED ? (C >> 1) : C;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int vi = (int)(idx % nvo);
        const long pix = idx / nvo;
        const int x = (int)(pix % Wd);
        const int y = (int)((pix / Wd) % H);
        const long b = pix / ((long)Wd * H);
        const int c = vi * 4;
        float o[4];
        if (GATED && c >= Ch) {
            o[0] = o[1] = o[2] = o[3] = 0.f;
            Vec4<T>::store(Y + pix * ldy + c, o);
            continue;
        }
        float a[4], g[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[j] = bias ? bias[c + j] : 0.f;
            g[j] = (GATED && bias) ? bias[Ch + c + j] : 0.f;
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = y + ky - 1;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = x + kx - 1;
                if ((unsigned)ix >= (unsigned)Wd) continue;
                const T* px = X + ((b * H + iy) * Wd + ix) * ldx;
                const float* w = Wt + (ky * 3 + kx) * C;
                float v[4];
                Vec4<T>::load(px + c, v);
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] = fmaf(v[j], w[c + j], a[j]);
                if (GATED) {
                    Vec4<T>::load(px + Ch + c, v);
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[j] = fmaf(v[j], w[Ch + c + j], g[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (GATED) o[j] = gelu_t<T>(a[j]) * g[j];
            else o[j] = act == EMIP_ACT_GELU ? gelu_t<T>(a[j]) : (act == EMIP_ACT_RELU ? fmaxf(a[j], 0.f) : a[j]);
        }
        Vec4<T>::store(Y + pix * ldy + c, o);
    }
}

// The gated form with XT consecutive output pixels of one row per thread (bf16, 4 channels = 8 B per lane: the second half
// starts at channel Ch, which need only be a multiple of 4): the two 3 x (XT + 2) windows are loaded once and reused from
// registers -- 36 loads per 16 outputs instead of 72 (the one-pixel form above ran the 680-channel GDFN at 1.2 TB/s).
template <int XT>
__global__ __launch_bounds__(256) void dwconv3x3_gated_rows_kernel(const bf16_t* __restrict__ X, long ldx, bf16_t* __restrict__ Y,
                                                                   long ldy, const float* __restrict__ Wt,
                                                                   const float* __restrict__ bias, int B, int H, int Wd, int C,
                                                                   int Cout_pad) {
    const int Ch = C >> 1, ncg = Cout_pad >> 2;
    const int nxg = (Wd + XT - 1) / XT;
    const long total = (long)B * H * nxg * ncg;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int cgi = (int)(idx % ncg);
        long r = idx / ncg;
        const int xg = (int)(r % nxg);
        r /= nxg;
        const int y = (int)(r % H);
        const long b = r / H;
        const int c = cgi * 4, x0 = xg * XT;
        bf16_t* yp = Y + ((b * H + y) * (long)Wd + x0) * ldy + c;
        if (c >= Ch) {                      // zero padding behind the gated channels
#pragma unroll
            for (int o = 0; o < XT; ++o)
                if (x0 + o < Wd) *reinterpret_cast<uint2*>(yp + (long)o * ldy) = make_uint2(0u, 0u);
            continue;
        }
        float acc[2][XT][4];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int o = 0; o < XT; ++o)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[hf][o][j] = bias ? bias[hf * Ch + c + j] : 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = y + ky - 1;
            const float rmask = (unsigned)iy < (unsigned)H ? 1.f : 0.f;
            const int iyc = min(max(iy, 0), H - 1);
            const bf16_t* rowp = X + ((b * H + iyc) * (long)Wd) * ldx + c;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                float w[3][4];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float4 t = *reinterpret_cast<const float4*>(Wt + (ky * 3 + kx) * C + hf * Ch + c);
                    w[kx][0] = t.x * rmask; w[kx][1] = t.y * rmask; w[kx][2] = t.z * rmask; w[kx][3] = t.w * rmask;
                }
                uint2 raw[XT + 2];
#pragma unroll
                for (int dx = 0; dx < XT + 2; ++dx) {
                    const int ix = x0 - 1 + dx;
                    const int ixc = min(max(ix, 0), Wd - 1);
                    const uint2 v = *reinterpret_cast<const uint2*>(rowp + (long)ixc * ldx + hf * Ch);
                    const unsigned m = (unsigned)ix < (unsigned)Wd ? 0xFFFFFFFFu : 0u;
                    raw[dx] = make_uint2(v.x & m, v.y & m);
                }
#pragma unroll
                for (int dx = 0; dx < XT + 2; ++dx) {
                    const float v[4] = {__uint_as_float(raw[dx].x << 16), __uint_as_float(raw[dx].x & 0xFFFF0000u),
                                        __uint_as_float(raw[dx].y << 16), __uint_as_float(raw[dx].y & 0xFFFF0000u)};
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int o = dx - kx;
                        if (o >= 0 && o < XT) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[hf][o][j] = fmaf(v[j], w[kx][j], acc[hf][o][j]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int o = 0; o < XT; ++o) {
            if (x0 + o < Wd) {
                bf16x4 ov;
#pragma unroll
                for (int j = 0; j < 4; ++j) ov[j] = (bf16_t)(gelu_t<bf16_t>(acc[0][o][j]) * acc[1][o][j]);
                *reinterpret_cast<bf16x4*>(yp + (long)o * ldy) = ov;
            }
        }
    }
}

// Depthwise 3x3 (+bias, +activation), 16 B of channels per thread and XT consecutive output pixels of
// one row per thread: the 3 x (XT+2) input window is loaded once and reused from registers, so every
// input element is fetched ~1.5x instead of 9x, with 16-B coalesced accesses along the channel axis.
template <typename T, int XT>
__global__ __launch_bounds__(256) void dwconv3x3_rows_kernel(const T* __restrict__ X, long ldx, T* __restrict__ Y,
                                                             long ldy, const float* __restrict__ Wt,
                                                             const float* __restrict__ bias, int B, int H, int Wd,
                                                             int C, int act, T* __restrict__ Z, long ldz) {
    constexpr int NV = 16 / sizeof(T);
    const int ncg = C / NV;
    const int nxg = (Wd + XT - 1) / XT;
    const long total = (long)B * H * nxg * ncg;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int cgi = (int)(idx % ncg);
        long r = idx / ncg;
        const int xg = (int)(r % nxg);
        r /= nxg;
        const int y = (int)(r % H);
        const long b = r / H;
        const int c = cgi * NV, x0 = xg * XT;
        float acc[XT][NV];
        {
            float bv[NV];
#pragma unroll
            for (int j = 0; j < NV; ++j) bv[j] = 0.f;
            if (bias) {
#pragma unroll
                for (int j = 0; j < NV; j += 4) {
                    const float4 t = *reinterpret_cast<const float4*>(bias + c + j);
                    bv[j] = t.x; bv[j + 1] = t.y; bv[j + 2] = t.z; bv[j + 3] = t.w;
                }
            }
#pragma unroll
            for (int o = 0; o < XT; ++o)
#pragma unroll
                for (int j = 0; j < NV; ++j) acc[o][j] = bv[j];
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = y + ky - 1;
            const float rmask = (unsigned)iy < (unsigned)H ? 1.f : 0.f;   // out-of-image rows: zero the weights,
            const int iyc = min(max(iy, 0), H - 1);                        // keep the (clamped) loads unconditional
            const T* rowp = X + ((b * H + iyc) * (long)Wd) * ldx + c;
            float w[3][NV];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int j = 0; j < NV; j += 4) {
                    const float4 t = *reinterpret_cast<const float4*>(Wt + (ky * 3 + kx) * C + c + j);
                    w[kx][j] = t.x * rmask; w[kx][j + 1] = t.y * rmask; w[kx][j + 2] = t.z * rmask;
                    w[kx][j + 3] = t.w * rmask;
                }
            uint4 raw[XT + 2];
#pragma unroll
            for (int dx = 0; dx < XT + 2; ++dx) {
                const int ix = x0 - 1 + dx;
                const int ixc = min(max(ix, 0), Wd - 1);
                raw[dx] = mask4(*reinterpret_cast<const uint4*>(rowp + (long)ixc * ldx), (unsigned)ix < (unsigned)Wd);
            }
#pragma unroll
            for (int dx = 0; dx < XT + 2; ++dx) {
                const T* tv = reinterpret_cast<const T*>(&raw[dx]);
                float v[NV];
#pragma unroll
                for (int j = 0; j < NV; ++j) v[j] = to_f32<T>(tv[j]);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int o = dx - kx;                    // output pixel this column feeds with tap kx
                    if (o >= 0 && o < XT) {
#pragma unroll
                        for (int j = 0; j < NV; ++j) acc[o][j] = fmaf(v[j], w[kx][j], acc[o][j]);
                    }
                }
            }
        }
        if (Z) {                         // pre-activation copy for the backward pass (training): same launch, one more store
#pragma unroll
            for (int o = 0; o < XT; ++o) {
                const int x = x0 + o;
                if (x < Wd) {
                    uint4 zv;
                    T* zp = reinterpret_cast<T*>(&zv);
#pragma unroll
                    for (int j = 0; j < NV; ++j) zp[j] = from_f32<T>(acc[o][j]);
                    *reinterpret_cast<uint4*>(Z + ((b * H + y) * (long)Wd + x) * ldz + c) = zv;
                }
            }
        }
        if (act == EMIP_ACT_GELU) {      // hoisted: a per-element test of the runtime `act` costs a branch per element
#pragma unroll
            for (int o = 0; o < XT; ++o)
#pragma unroll
                for (int j = 0; j < NV; ++j) acc[o][j] = gelu_t<T>(acc[o][j]);
        } else if (act == EMIP_ACT_RELU) {
#pragma unroll
            for (int o = 0; o < XT; ++o)
#pragma unroll
                for (int j = 0; j < NV; ++j) acc[o][j] = fmaxf(acc[o][j], 0.f);
        }
#pragma unroll
        for (int o = 0; o < XT; ++o) {
            const int x = x0 + o;
            if (x < Wd) {
                uint4 ov;
                T* op = reinterpret_cast<T*>(&ov);
#pragma unroll
                for (int j = 0; j < NV; ++j) op[j] = from_f32<T>(acc[o][j]);
                *reinterpret_cast<uint4*>(Y + ((b * H + y) * (long)Wd + x) * ldy + c) = ov;
            }
        }
    }
}

// ----------------------------------------------------------------------------
// Per-(group, channel) sum and sum of squares over `rows` rows: InstanceNorm2d statistics
// (group = image) and train-mode BatchNorm statistics (one group).  f64 atomics combine the
// block partials so that E[x^2]-E[x]^2 is evaluated without cancellation trouble.
// Round 4: every block partial is rounded to a multiple of 2^-24 before it is added.  Sums of such
// multiples below 2^29 are EXACT in f64, exact additions commute and associate, so the totals no
// longer depend on the order in which the blocks arrive (reproducible bit for bit); the rounding
// moves a partial by at most 3e-8, eleven orders below the sums it joins.
__device__ __forceinline__ double chan_q24(float t) { return rint((double)t * 16777216.0) * (1.0 / 16777216.0); }
template <typename T>
__global__ __launch_bounds__(256) void chan_stats_kernel(const T* __restrict__ X, long ldx, double* __restrict__ sums,
                                                         long rows, int C, int rows_per_block) {
    __shared__ float red[256 * 8];
    const int nv = C >> 2;
    const int plan = 256 / nv;  // row lanes
    const int vi = threadIdx.x % nv, rl = threadIdx.x / nv;
    const long g = blockIdx.y;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(rows, r0 + rows_per_block);
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (rl < plan) {
        for (long r = r0 + rl; r < r1; r += plan) {
            float v[4];
            Vec4<T>::load(X + (g * rows + r) * ldx + vi * 4, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s[j] += v[j];
                q[j] = fmaf(v[j], v[j], q[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[threadIdx.x * 8 + j] = s[j];
        red[threadIdx.x * 8 + 4 + j] = q[j];
    }
    __syncthreads();
    if (threadIdx.x < nv) {
        float ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int l = 0; l < plan; ++l) {
#pragma unroll
            for (int j = 0; j < 8; ++j) ts[j] += red[(l * nv + threadIdx.x) * 8 + j];
        }
        double* dst = sums + (g * C + threadIdx.x * 4) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            atomicAdd(dst + j * 2, chan_q24(ts[j]));
            atomicAdd(dst + j * 2 + 1, chan_q24(ts[4 + j]));
        }
    }
}

// The same for bf16 rows read 16 bytes per lane with four rows in flight per thread: the form above (8-byte loads, one
// dependent row after the other) streamed the GMFlow encoder's 63-MB feature maps at 1.2-1.5 TB/s.
__global__ __launch_bounds__(256) void chan_stats_wide_kernel(const bf16_t* __restrict__ X, long ldx, double* __restrict__ sums,
                                                              long rows, int C, int rows_per_block) {
    __shared__ float red[256 * 16];
    const int nv = C >> 3;
    const int plan = 256 / nv;  // row lanes
    const int vi = threadIdx.x % nv, rl = threadIdx.x / nv;
    const long g = blockIdx.y;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(rows, r0 + rows_per_block);
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
    if (rl < plan) {
        const bf16_t* base = X + g * rows * ldx + vi * 8;
        long r = r0 + rl;
        for (; r + 3L * plan < r1; r += 4L * plan) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const uint4*>(base + (r + (long)u * plan) * ldx);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float a = __uint_as_float(w[k] << 16), b = __uint_as_float(w[k] & 0xFFFF0000u);
                    s[2 * k] += a;
                    q[2 * k] = fmaf(a, a, q[2 * k]);
                    s[2 * k + 1] += b;
                    q[2 * k + 1] = fmaf(b, b, q[2 * k + 1]);
                }
            }
        }
        for (; r < r1; r += plan) {
            const uint4 v = *reinterpret_cast<const uint4*>(base + r * ldx);
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float a = __uint_as_float(w[k] << 16), b = __uint_as_float(w[k] & 0xFFFF0000u);
                s[2 * k] += a;
                q[2 * k] = fmaf(a, a, q[2 * k]);
                s[2 * k + 1] += b;
                q[2 * k + 1] = fmaf(b, b, q[2 * k + 1]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[threadIdx.x * 16 + j] = s[j];
        red[threadIdx.x * 16 + 8 + j] = q[j];
    }
    __syncthreads();
    // one (channel, statistic) pair per thread: 2 C <= 256 sums over the row lanes, then one f64 atomic each
    if ((int)threadIdx.x < 2 * C) {
        const int c = threadIdx.x >> 1, which = threadIdx.x & 1;
        const int slot = (c >> 3) * 16 + which * 8 + (c & 7);
        float t = 0.f;
        for (int l = 0; l < plan; ++l) t += red[l * nv * 16 + slot];
        atomicAdd(sums + (g * C + c) * 2 + which, chan_q24(t));
    }
}

// y = [relu]( R + [relu]( (x - mean) * rstd * gamma + beta ) ), statistics from chan_stats_kernel.
template <typename T>
__global__ __launch_bounds__(256) void chan_norm_apply_kernel(const T* __restrict__ X, long ldx, T* __restrict__ Y,
                                                              long ldy, const T* __restrict__ R, long ldr,
                                                              const double* __restrict__ sums,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, long groups, long rows,
                                                              int C, float eps, int relu_inner, int relu_outer) {
    const int nv = C >> 2;
    const long total = groups * rows * nv;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int vi = (int)(idx % nv);
        const long row = idx / nv;
        const long g = row / rows;
        float v[4], o[4];
        Vec4<T>::load(X + row * ldx + vi * 4, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = vi * 4 + j;
            const double m = sums[(g * C + c) * 2] / (double)rows;
            const double var = fmax(sums[(g * C + c) * 2 + 1] / (double)rows - m * m, 0.0);
            float t = (v[j] - (float)m) * (float)(1.0 / sqrt(var + (double)eps));
            if (gamma) t = t * gamma[c] + beta[c];
            if (relu_inner) t = fmaxf(t, 0.f);
            o[j] = t;
        }
        if (R) {
            float r[4];
            Vec4<T>::load(R + row * ldr + vi * 4, r);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] += r[j];
        }
        if (relu_outer) {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], 0.f);
        }
        Vec4<T>::store(Y + row * ldy + vi * 4, o);
    }
}

// The same with one channel vector per thread for the whole launch: mean / rstd (float64 division and square root) are
// evaluated once per thread instead of once per element -- the element-wise form above spent its time in f64 VALU work --
// and rows are read 16 bytes per lane.  grid (row chunks, groups); same operation order per element as above.
template <typename T>
__global__ __launch_bounds__(256) void chan_norm_apply_wide_kernel(const T* __restrict__ X, long ldx, T* __restrict__ Y,
                                                                   long ldy, const T* __restrict__ R, long ldr,
                                                                   const double* __restrict__ sums,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, long rows, int C,
                                                                   float eps, int relu_inner, int relu_outer,
                                                                   const double* __restrict__ rsums, int r_relu) {
    constexpr int VEC = 16 / sizeof(T);
    __shared__ float s_mean[1024], s_rstd[1024];
    // rsums: the residual R is itself a RAW tensor whose InstanceNorm (+ ReLU with r_relu) was never stored: it is normalised
    // here from its own sums and rounded to T as the stored tensor would have been (same bits as the two-pass form)
    __shared__ float s_rmean[1024], s_rrstd[1024];
    const int nv = C / VEC, rl = 256 / nv;
    const long g = blockIdx.y;
    for (int c = threadIdx.x; c < C; c += 256) {               // one channel per thread: the float64 part, once per workgroup
        const double m = sums[(g * C + c) * 2] / (double)rows;
        const double var = fmax(sums[(g * C + c) * 2 + 1] / (double)rows - m * m, 0.0);
        s_mean[c] = (float)m;
        s_rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (rsums) {
            const double rm = rsums[(g * C + c) * 2] / (double)rows;
            const double rvar = fmax(rsums[(g * C + c) * 2 + 1] / (double)rows - rm * rm, 0.0);
            s_rmean[c] = (float)rm;
            s_rrstd[c] = (float)(1.0 / sqrt(rvar + (double)eps));
        }
    }
    __syncthreads();
    if ((int)threadIdx.x >= nv * rl) return;
    const int vi = threadIdx.x % nv, lr = threadIdx.x / nv;
    float mean[VEC], rstd[VEC], ga[VEC], be[VEC], rmean[VEC], rrstd[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int c = vi * VEC + j;
        mean[j] = s_mean[c];
        rstd[j] = s_rstd[c];
        rmean[j] = rsums ? s_rmean[c] : 0.f;
        rrstd[j] = rsums ? s_rrstd[c] : 1.f;
        ga[j] = gamma ? gamma[c] : 1.f;
        be[j] = gamma ? beta[c] : 0.f;
    }
    const long rpb = (rows + gridDim.x - 1) / gridDim.x;
    const long r0 = (long)blockIdx.x * rpb, r1 = min(rows, r0 + rpb);
    for (long r = r0 + lr; r < r1; r += rl) {
        const long row = g * rows + r;
        const uint4 xv = *reinterpret_cast<const uint4*>(X + row * ldx + vi * VEC);
        uint4 rv = xv;
        if (R) rv = *reinterpret_cast<const uint4*>(R + row * ldr + vi * VEC);
        uint4 ov;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float t = (to_f32<T>(reinterpret_cast<const T*>(&xv)[j]) - mean[j]) * rstd[j];
            if (gamma) t = t * ga[j] + be[j];
            if (relu_inner) t = fmaxf(t, 0.f);
            if (R) {
                float rr = to_f32<T>(reinterpret_cast<const T*>(&rv)[j]);
                if (rsums) {
                    rr = (rr - rmean[j]) * rrstd[j];
                    if (r_relu) rr = fmaxf(rr, 0.f);
                    rr = to_f32<T>(from_f32<T>(rr));
                }
                t += rr;
            }
            if (relu_outer) t = fmaxf(t, 0.f);
            reinterpret_cast<T*>(&ov)[j] = from_f32<T>(t);
        }
        *reinterpret_cast<uint4*>(Y + row * ldy + vi * VEC) = ov;
    }
}

// ----------------------------------------------------------------------------
// Bilinear resize of a channels-last tensor.  align_corners=1: src = dst*(in-1)/(out-1);
// align_corners=0: src = max((dst+0.5)*in/out-0.5, 0).  Index/weight arithmetic in f32
// exactly as ATen's upsample_bilinear2d evaluates it.
__device__ __forceinline__ void bilin_src(int dst, int in, int out, int align, int& i0, int& i1, float& l1) {
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

template <typename T>
__global__ __launch_bounds__(256) void bilinear_kernel(const T* __restrict__ X, long ldx, T* __restrict__ Y, long ldy,
                                                       int B, int H, int Wd, int C, int Ho, int Wo, int align,
                                                       float mul) {
    const int nv = C >> 2;
    const long total = (long)B * Ho * Wo * nv;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int vi = (int)(idx % nv);
        const long pix = idx / nv;
        const int ox = (int)(pix % Wo);
        const int oy = (int)((pix / Wo) % Ho);
        const long b = pix / ((long)Wo * Ho);
        int y0, y1, x0, x1;
        float ly, lx;
        bilin_src(oy, H, Ho, align, y0, y1, ly);
        bilin_src(ox, Wd, Wo, align, x0, x1, lx);
        const T* base = X + b * H * Wd * ldx + vi * 4;
        float a[4], bq[4], c[4], d[4], o[4];
        Vec4<T>::load(base + ((long)y0 * Wd + x0) * ldx, a);
        Vec4<T>::load(base + ((long)y0 * Wd + x1) * ldx, bq);
        Vec4<T>::load(base + ((long)y1 * Wd + x0) * ldx, c);
        Vec4<T>::load(base + ((long)y1 * Wd + x1) * ldx, d);
        const float hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = mul * (hy * (hx * a[j] + lx * bq[j]) + ly * (hx * c[j] + lx * d[j]));
        Vec4<T>::store(Y + pix * ldy + vi * 4, o);
    }
}

// single channel, T in -> f32 out (the x8 mask logits and the x8 train-mode flow)
template <typename T>
__global__ __launch_bounds__(256) void bilinear_plane_kernel(const T* __restrict__ X, long ldx, int xc,
                                                             float* __restrict__ Y, int B, int H, int Wd, int Cout,
                                                             int Ho, int Wo, int align, float mul) {
    // X: [B][H][W][ldx] channels-last, reads channels xc..xc+Cout-1; Y: [B][Cout][Ho][Wo] planar f32
    const long total = (long)B * Cout * Ho * Wo;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % Wo);
        const int oy = (int)((idx / Wo) % Ho);
        const int c = (int)((idx / ((long)Wo * Ho)) % Cout);
        const long b = idx / ((long)Wo * Ho * Cout);
        int y0, y1, x0, x1;
        float ly, lx;
        bilin_src(oy, H, Ho, align, y0, y1, ly);
        bilin_src(ox, Wd, Wo, align, x0, x1, lx);
        const T* base = X + b * H * Wd * ldx + xc + c;
        const float a = to_f32<T>(base[((long)y0 * Wd + x0) * ldx]);
        const float bq = to_f32<T>(base[((long)y0 * Wd + x1) * ldx]);
        const float cc = to_f32<T>(base[((long)y1 * Wd + x0) * ldx]);
        const float d = to_f32<T>(base[((long)y1 * Wd + x1) * ldx]);
        Y[idx] = mul * ((1.f - ly) * ((1.f - lx) * a + lx * bq) + ly * ((1.f - lx) * cc + lx * d));
    }
}

// ----------------------------------------------------------------------------
// Y = A op B (op C): elementwise over [M][C] with independent row strides.
// mode 0: A*B   1: A*B*C3   2: A+B   3: A + bcast(B[row % period])
template <typename T>
__global__ __launch_bounds__(256) void eltwise_kernel(const T* __restrict__ A, long lda, const T* __restrict__ Bp,
                                                      long ldb, const T* __restrict__ C3, long ldc3,
                                                      T* __restrict__ Y, long ldy, long M, int C, int mode,
                                                      long period) {
    const int nv = C >> 2;
    const long total = M * nv;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int vi = (int)(idx % nv);
        const long row = idx / nv;
        float a[4], b[4], o[4];
        Vec4<T>::load(A + row * lda + vi * 4, a);
        const long brow = mode == 3 ? row % period : row;
        Vec4<T>::load(Bp + brow * ldb + vi * 4, b);
        if (mode == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = a[j] * b[j];
        } else if (mode == 1) {
            float c[4];
            Vec4<T>::load(C3 + row * ldc3 + vi * 4, c);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = a[j] * b[j] * c[j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = a[j] + b[j];
        }
        Vec4<T>::store(Y + row * ldy + vi * 4, o);
    }
}

// ----------------------------------------------------------------------------
// planar f32 [B][C][P]  ->  channels-last T [B][P][ldy], zero-filling channels C..Cpad-1
template <typename T>
__global__ __launch_bounds__(256) void planar_to_cl_kernel(const float* __restrict__ X, T* __restrict__ Y, long ldy,
                                                           int B, int C, long P, int Cpad) {
    __shared__ float tile[32][33];
    const long p0 = (long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const long b = blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i;
        const long p = p0 + tx;
        tile[i][tx] = (c < C && p < P) ? X[(b * C + c) * P + p] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const long p = p0 + i;
        const int c = c0 + tx;
        if (p < P && c < Cpad) Y[(b * P + p) * ldy + c] = from_f32<T>(tile[tx][i]);
    }
}

// ... for images (C <= 8 planes -> 8 bf16 channels): one pixel per thread, plane reads and 16-byte row writes both coalesced
// (the 32 x 32 transpose above uses 3 of its 32 tile rows on an RGB frame and writes 2 bytes per lane)
__global__ __launch_bounds__(256) void planar_to_cl8_kernel(const float* __restrict__ X, bf16_t* __restrict__ Y, int B, int C, long P) {
    const long total = (long)B * P;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / P, p = i - b * P;
        bf16x8 o;
#pragma unroll
        for (int c = 0; c < 8; ++c) o[c] = (bf16_t)(c < C ? X[(b * C + c) * P + p] : 0.f);
        *reinterpret_cast<bf16x8*>(Y + i * 8) = o;
    }
}

// channels-last T [B][P][ldx] (channels xc..xc+C-1)  ->  planar f32 [B][C][P]
template <typename T>
__global__ __launch_bounds__(256) void cl_to_planar_kernel(const T* __restrict__ X, long ldx, int xc,
                                                           float* __restrict__ Y, int B, int C, long P) {
    __shared__ float tile[32][33];
    const long p0 = (long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const long b = blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const long p = p0 + i;
        const int c = c0 + tx;
        tile[i][tx] = (p < P && c < C) ? to_f32<T>(X[(b * P + p) * ldx + xc + c]) : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i;
        const long p = p0 + tx;
        if (c < C && p < P) Y[(b * C + c) * P + p] = tile[tx][i];
    }
}

// strided 2-D copy with dtype conversion and zero padding: Y[m][yc + c] = c < C ? X[m][xc + c] : 0, c < Cpad
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void copy_cols_kernel(const TI* __restrict__ X, long ldx, int xc, TO* __restrict__ Y,
                                                        long ldy, int yc, long M, int C, int Cpad) {
    const long total = M * Cpad;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % Cpad);
        const long m = idx / Cpad;
        const float v = c < C ? to_f32<TI>(X[m * ldx + xc + c]) : 0.f;
        Y[m * ldy + yc + c] = from_f32<TO>(v);
    }
}

// ----------------------------------------------------------------------------
// GMFlow convex upsampling (gmflow.py:64-77): logits [N][h][w][576] with channel = (k*8+i)*8+j,
// flow f32 [N][h][w][2]; out f32 planar [N][2][8h][8w].
template <typename T>
__global__ __launch_bounds__(256) void convex_up_kernel(const T* __restrict__ L, long ldl,
                                                        const float* __restrict__ F, float* __restrict__ Y, int N,
                                                        int H, int Wd) {
    const long total = (long)N * H * Wd * 64;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ij = (int)(idx & 63);
        const long pix = idx >> 6;
        const int x = (int)(pix % Wd);
        const int y = (int)((pix / Wd) % H);
        const long n = pix / ((long)Wd * H);
        float lg[9];
        float mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            lg[k] = to_f32<T>(L[pix * ldl + k * 64 + ij]);
            mx = fmaxf(mx, lg[k]);
        }
        float den = 0.f, ax = 0.f, ay = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float e = expf(lg[k] - mx);
            den += e;
            const int iy = y + k / 3 - 1, ix = x + k % 3 - 1;
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)Wd) {
                const float* f = F + ((n * H + iy) * Wd + ix) * 2;
                ax = fmaf(e, 8.f * f[0], ax);
                ay = fmaf(e, 8.f * f[1], ay);
            }
        }
        const int i = ij >> 3, j = ij & 7;
        const long Ho = 8L * H, Wo = 8L * Wd;
        const long o = (n * 2 * Ho + (8L * y + i)) * Wo + 8L * x + j;
        Y[o] = ax / den;
        Y[o + Ho * Wo] = ay / den;
    }
}

// correspondence (first two columns of O, f32) minus pixel grid -> flow f32 [N][hw][2]
__global__ __launch_bounds__(256) void corresp_to_flow_kernel(const float* __restrict__ O, long ldo,
                                                              float* __restrict__ F, long N, int hw, int Wd,
                                                              int sub_grid) {
    const long total = N * hw;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int p = (int)(idx % hw);
        const float gx = sub_grid ? (float)(p % Wd) : 0.f, gy = sub_grid ? (float)(p / Wd) : 0.f;
        F[idx * 2] = O[idx * ldo] - gx;
        F[idx * 2 + 1] = O[idx * ldo + 1] - gy;
    }
}

template <typename T>
struct TypeOf { typedef T type; };

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

extern "C" int emip_version(void) { return 100; }

extern "C" int emip_layernorm(const void* X, long ldx, void* Y, long ldy, const float* gamma, const float* beta,
                              const void* R, long ldr, float* out_stats, long M, int C, float eps, int dtype,
                              void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(X && Y && gamma && beta && M > 0 && C >= 4 && C <= 1024 && (C & 3) == 0);
    EMIP_REQUIRE((ldx & 3) == 0 && (ldy & 3) == 0 && ldx >= C && ldy >= C);
    if (R) EMIP_REQUIRE((ldr & 3) == 0 && ldr >= C);
    if (dtype == EMIP_BF16 && (C & 7) == 0 && C <= 512 && (ldx & 7) == 0 && (ldy & 7) == 0 && aligned16(X) && aligned16(Y) &&
        (!R || ((ldr & 7) == 0 && aligned16(R)))) {
        const int vpl = ((C >> 3) + 7) / 8;
        const int grid = grid_for((M + 7) / 8, 4);
#define EMIP_LN_ROWS(V)                                                                                                   \
    hipLaunchKernelGGL((layernorm_rows_kernel<V>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)X, ldx,    \
                       (bf16_t*)Y, ldy, gamma, beta, (const bf16_t*)R, ldr, out_stats, M, C, eps)
        switch (vpl) {
            case 1: EMIP_LN_ROWS(1); break;
            case 2: EMIP_LN_ROWS(2); break;
            case 3: EMIP_LN_ROWS(3); break;
            case 4: EMIP_LN_ROWS(4); break;
            case 5: EMIP_LN_ROWS(5); break;
            case 6: EMIP_LN_ROWS(6); break;
            case 7: EMIP_LN_ROWS(7); break;
            default: EMIP_LN_ROWS(8); break;
        }
#undef EMIP_LN_ROWS
        return emip_launch_status();
    }
    const int nv = C >> 2;
    int lg = 0;
    while ((1 << lg) < nv && lg < 6) ++lg;
    const int rows_per_wave = 64 >> lg;
    const long waves = (M + rows_per_wave - 1) / rows_per_wave;
    const int grid = grid_for(waves, 4);
    DISPATCH_T(dtype, hipLaunchKernelGGL(layernorm_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                         (const T*)X, ldx, (T*)Y, ldy, gamma, beta, (const T*)R, ldr, out_stats, M, C, eps,
                                         lg));
    return emip_launch_status();
}

// A f32 [M][lda] (complete split-K sums) -> Y [M][ldy] in the storage type; out_stats (may be NULL) f32 [M][2].
extern "C" int emip_rows_finalize(const float* A, long lda, void* Y, long ldy, float* out_stats, long M, int C, int dtype,
                                  void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(A && Y && M > 0 && C >= 4 && C <= 4096 && (C & 3) == 0 && (lda & 3) == 0 && (ldy & 3) == 0 && lda >= C &&
                 ldy >= C && aligned16(A));
    const int nv = C >> 2;
    int lg = 0;
    while ((1 << lg) < nv && lg < 6) ++lg;
    const int rows_per_wave = 64 >> lg;
    const long waves = (M + rows_per_wave - 1) / rows_per_wave;
    DISPATCH_T(dtype, hipLaunchKernelGGL(rows_finalize_kernel<T>, dim3(grid_for(waves, 4)), dim3(256), 0,
                                         (hipStream_t)stream, A, lda, (T*)Y, ldy, out_stats, M, C, lg));
    return emip_launch_status();
}

static int g_lnb_wide = 1, g_lnb_rows = 1;
static long g_lnb_blocks = 256;
#ifdef EMIP_TUNING
extern "C" int emip_debug_set_lnb(int wide) {          // 0: the 4-channel form, 1: two 8-channel vectors per lane, 2 (default): + rows
    g_lnb_wide = wide != 0;
    g_lnb_rows = wide >= 2 || wide < 0;
    return EMIP_OK;
}
#endif

// dgamma / dbeta are ACCUMULATED into (the caller zero-fills them, or keeps accumulating across micro-batches)
extern "C" int emip_layernorm_bwd(const void* X, long ldx, const void* DY, long lddy, void* DX, long lddx,
                                  const float* gamma, float* dgamma, float* dbeta, int nparts, long part_stride, long M,
                                  int C, float eps, int dtype, void* stream) {
    return emip_layernorm_bwd_res(X, ldx, DY, lddy, DX, lddx, nullptr, 0, gamma, dgamma, dbeta, nparts, part_stride, M, C,
                                  eps, dtype, stream);
}

// bf16 rows of C = 8 * 8 * VPL / ... channels with EIGHT lanes per row and VPL 16-byte vectors per lane (vector v of the row
// sits in lane v % 8): every lane works (the two-vectors-per-lane form above keeps 40 of 64 vector slots busy at C = 320 and
// reduces over 32 lanes: it ran the 15 488 x 320 launches of the training step at 1 TB/s), a row statistic is three shuffle
// levels, a wave instruction reads whole 128-byte segments of 8 rows.
template <int VPL>
__global__ __launch_bounds__(256) void layernorm_bwd_rows_kernel(const bf16_t* __restrict__ X, long ldx,
                                                                 const bf16_t* __restrict__ DY, long lddy,
                                                                 bf16_t* __restrict__ DX, long lddx,
                                                                 const float* __restrict__ gamma, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta, long M, int C, float eps, int nparts,
                                                                 long part_stride, const bf16_t* __restrict__ DR, long lddr) {
    extern __shared__ float red[];   // [2][C]
    const int lane = threadIdx.x & 63;
    const int sub = lane & 7, rsub = lane >> 3;
    const int nv8 = C >> 3;
    const int nwave = blockDim.x >> 6;
    const long wave_global = (long)blockIdx.x * nwave + (threadIdx.x >> 6);
    const long stride = (long)gridDim.x * nwave * 8;
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) red[i] = 0.f;
    __syncthreads();
    float gam[VPL][8], ag[VPL][8], ab[VPL][8];
    bool act[VPL];
    int vcl[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int vi = sub + 8 * i;
        act[i] = vi < nv8;
        vcl[i] = min(vi, nv8 - 1) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            gam[i][j] = act[i] ? gamma[vcl[i] + j] : 0.f;
            ag[i][j] = ab[i][j] = 0.f;
        }
    }
    const float invC = 1.f / (float)C;
    for (long row0 = wave_global * 8; row0 < M; row0 += stride) {
        const long row = row0 + rsub;
        const long rowc = row < M ? row : M - 1;
        Raw8<bf16_t> rx[VPL], rdy[VPL], rdr[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            rx[i].load(X + rowc * ldx + vcl[i]);
            rdy[i].load(DY + rowc * lddy + vcl[i]);
            if (DR) rdr[i].load(DR + rowc * lddr + vcl[i]);
        }
        float x[VPL][8], dy[VPL][8];
        float s = 0.f, s2 = 0.f, sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const float keep = (row < M && act[i]) ? 1.f : 0.f;
            rx[i].unpack(x[i], keep);
            rdy[i].unpack(dy[i], keep);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xv = x[i][j], g = dy[i][j] * gam[i][j];
                s += xv;
                s2 += xv * xv;
                sg += g;
                sgx += g * xv;
            }
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) {
            s += __shfl_xor(s, o);
            s2 += __shfl_xor(s2, o);
            sg += __shfl_xor(sg, o);
            sgx += __shfl_xor(sgx, o);
        }
        const float mean = s * invC;
        const float rstd = rsqrtf(fmaxf(s2 * invC - mean * mean, 0.f) + eps);
        const float mg = sg * invC, mgx = rstd * (sgx - mean * sg) * invC;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            float o8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = (x[i][j] - mean) * rstd;           // padded lanes / rows: dy = 0, nothing is accumulated
                ag[i][j] += dy[i][j] * xh;
                ab[i][j] += dy[i][j];
                o8[j] = rstd * (dy[i][j] * gam[i][j] - mg - xh * mgx);
            }
            if (DR) {
                float r8[8];
                rdr[i].unpack(r8, 1.f);
#pragma unroll
                for (int j = 0; j < 8; ++j) o8[j] += r8[j];
            }
            if (row < M && act[i]) store8<bf16_t>(DX + row * lddx + vcl[i], o8);
        }
    }
    // the 8 row groups of a wave hold partial dgamma / dbeta of the same channels: shuffles, one LDS atomic per channel and
    // wave, one global atomic per channel and workgroup
#pragma unroll
    for (int i = 0; i < VPL; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = ag[i][j], b = ab[i][j];
#pragma unroll
            for (int o = 32; o >= 8; o >>= 1) {
                a += __shfl_xor(a, o);
                b += __shfl_xor(b, o);
            }
            if (rsub == 0 && act[i]) {
                atomicAdd(&red[vcl[i] + j], a);
                atomicAdd(&red[C + vcl[i] + j], b);
            }
        }
    __syncthreads();
    const long po = (long)(blockIdx.x % nparts) * part_stride;
    for (int i = threadIdx.x; i < C; i += blockDim.x) {
        atomicAdd(dgamma + po + i, red[i]);
        atomicAdd(dbeta + po + i, red[C + i]);
    }
}

// The same with DX = LayerNorm backward + DR: in a pre-norm residual block x feeds the LayerNorm AND the skip connection, so
// the gradient of the skip path (DR, same dtype as DX, may alias DX) is added here instead of by a separate add launch.
extern "C" int emip_layernorm_bwd_res(const void* X, long ldx, const void* DY, long lddy, void* DX, long lddx, const void* DR,
                                      long lddr, const float* gamma, float* dgamma, float* dbeta, int nparts,
                                      long part_stride, long M, int C, float eps, int dtype, void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(X && DY && DX && gamma && dgamma && dbeta && M > 0 && C >= 4 && C <= 1024 && (C & 3) == 0);
    EMIP_REQUIRE(nparts >= 1 && (nparts == 1 || part_stride >= C));
    EMIP_REQUIRE((ldx & 3) == 0 && (lddy & 3) == 0 && (lddx & 3) == 0 && ldx >= C && lddy >= C && lddx >= C);
    if (DR) EMIP_REQUIRE((lddr & 3) == 0 && lddr >= C);
    if (g_lnb_wide && (C & 7) == 0 && (ldx & 7) == 0 && (lddy & 7) == 0 && (lddx & 7) == 0 && aligned16(X) && aligned16(DY) &&
        aligned16(DX) && (!DR || ((lddr & 7) == 0 && aligned16(DR)))) {
        const int nv8 = C >> 3;
        if (dtype == EMIP_BF16 && g_lnb_rows && nv8 <= 64) {
            // eight lanes per row: 256-thread workgroups of 32 rows per sweep, at most 256 workgroups = one per CU (the kernel
            // holds ~350 registers at C = 320: one wave per SIMD, so 512 workgroups were two rounds anyway).  Every workgroup
            // ends in one global atomic per channel and statistic, ~25 ns each on the same address: 6.4 us instead of 12.8
            const int vpl = (nv8 + 7) / 8;
            long blocks = (M + 63) / 64;
            if (blocks > g_lnb_blocks) blocks = g_lnb_blocks;
            if (blocks < 1) blocks = 1;
#define EMIP_LNB_ROWS(V)                                                                                                       \
    hipLaunchKernelGGL((layernorm_bwd_rows_kernel<V>), dim3((unsigned)blocks), dim3(256), 2 * C * sizeof(float),               \
                       (hipStream_t)stream, (const bf16_t*)X, ldx, (const bf16_t*)DY, lddy, (bf16_t*)DX, lddx, gamma, dgamma,  \
                       dbeta, M, C, eps, nparts, part_stride, (const bf16_t*)DR, lddr)
            switch (vpl) {
                case 1: EMIP_LNB_ROWS(1); break;
                case 2: EMIP_LNB_ROWS(2); break;
                case 3: EMIP_LNB_ROWS(3); break;
                case 4: EMIP_LNB_ROWS(4); break;
                case 5: EMIP_LNB_ROWS(5); break;
                case 6: EMIP_LNB_ROWS(6); break;
                case 7: EMIP_LNB_ROWS(7); break;
                default: EMIP_LNB_ROWS(8); break;
            }
#undef EMIP_LNB_ROWS
            return emip_launch_status();
        }
        const int U = 1;
        int lg = 0;
        while ((2 << lg) < nv8 && lg < 6) ++lg;
        const long rows_per_wave = (64 >> lg) * U;
        const long waves = (M + rows_per_wave - 1) / rows_per_wave;
        const int threads = waves >= 256 * 16 ? 1024 : (waves >= 256 * 8 ? 512 : 256);
        long blocks = (waves + threads / 64 - 1) / (threads / 64);
        if (blocks > 256) blocks = 256;
        if (dtype == EMIP_F32) {
            hipLaunchKernelGGL((layernorm_bwd_wide_kernel<float, 1>), dim3((unsigned)blocks), dim3(threads),
                               2 * C * sizeof(float), (hipStream_t)stream, (const float*)X, ldx, (const float*)DY, lddy,
                               (float*)DX, lddx, gamma, dgamma, dbeta, M, C, eps, lg, nparts, part_stride, (const float*)DR,
                               lddr);
        } else {
            hipLaunchKernelGGL((layernorm_bwd_wide_kernel<bf16_t, 1>), dim3((unsigned)blocks), dim3(threads),
                               2 * C * sizeof(float), (hipStream_t)stream, (const bf16_t*)X, ldx, (const bf16_t*)DY, lddy,
                               (bf16_t*)DX, lddx, gamma, dgamma, dbeta, M, C, eps, lg, nparts, part_stride,
                               (const bf16_t*)DR, lddr);
        }
        return emip_launch_status();
    }
    const int nv = C >> 2;
    int lg = 0;
    while ((1 << lg) < nv && lg < 6) ++lg;
    const int rows_per_wave = 64 >> lg;
    const long waves = (M + rows_per_wave - 1) / rows_per_wave;
    long blocks = (waves + 3) / 4;
    const long cap = nparts >= 16 ? 4096 : 1024;     // many workgroups only pay off when the atomics are spread out
    if (blocks > cap) blocks = cap;
    DISPATCH_T(dtype, hipLaunchKernelGGL(layernorm_bwd_kernel<T>, dim3((unsigned)blocks), dim3(256),
                                         2 * C * sizeof(float), (hipStream_t)stream, (const T*)X, ldx, (const T*)DY,
                                         lddy, (T*)DX, lddx, gamma, dgamma, dbeta, M, C, eps, lg, nparts,
                                         part_stride, (const T*)DR, lddr));
    return emip_launch_status();
}

extern "C" int emip_dwconv3x3(const void* X, long ldx, void* Y, long ldy, const float* Wt, const float* bias, int B,
                              int H, int Wd, int C, int act, int dtype, void* stream) {
    return emip_dwconv3x3_dual(X, ldx, Y, ldy, nullptr, 0, Wt, bias, B, H, Wd, C, act, dtype, stream);
}

// emip_dwconv3x3 that also stores the pre-activation values Z (what the GELU backward needs) from the same pass
extern "C" int emip_dwconv3x3_dual(const void* X, long ldx, void* Y, long ldy, void* Z, long ldz, const float* Wt,
                                   const float* bias, int B, int H, int Wd, int C, int act, int dtype, void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(X && Y && Wt && B > 0 && H > 0 && Wd > 0 && C >= 4 && (C & 3) == 0);
    EMIP_REQUIRE((ldx & 3) == 0 && (ldy & 3) == 0 && ldx >= C && ldy >= C);
    const int nv = dtype == EMIP_F32 ? 4 : 8;
    EMIP_REQUIRE(act >= EMIP_ACT_NONE && act <= EMIP_ACT_GELU);
    if (C % nv == 0 && ldx % nv == 0 && ldy % nv == 0 && aligned16(X) && aligned16(Y)) {
        const long total = (long)B * H * ((Wd + 3) / 4) * (C / nv);
        if (Z) EMIP_REQUIRE(ldz % nv == 0 && ldz >= C && aligned16(Z));
        DISPATCH_T(dtype, hipLaunchKernelGGL((dwconv3x3_rows_kernel<T, 4>), dim3(grid_for(total, 256)), dim3(256), 0,
                                             (hipStream_t)stream, (const T*)X, ldx, (T*)Y, ldy, Wt, bias, B, H, Wd, C,
                                             act, (T*)Z, ldz));
        return emip_launch_status();
    }
    EMIP_REQUIRE(Z == nullptr);          // the dual store lives in the vectorised kernel only
    const long total = (long)B * H * Wd * (C >> 2);
    DISPATCH_T(dtype, hipLaunchKernelGGL((dwconv3x3_kernel<T, false>), dim3(grid_for(total, 256)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)X, ldx, (T*)Y, ldy, Wt, bias, B, H, Wd, C, C,
                                         act));
    return emip_launch_status();
}

// The 3 x 3 patch matrix of a channels-last map: Y[(b, y, x)][(ky * 3 + kx) * C + c] = X[b][y + ky - 1][x + kx - 1][c], zero outside
// the image.  The A operand of a convolution whose WEIGHTS differ per image (CoUpdater.run_conv_corr_factored: the correlation
// volume in front of conv_corr, model/EMIP_short/model.py:59,96, is a rank-128 product, so its 3 x 3 convolution is a
// per-image GEMM against this matrix).
namespace {
template <typename T>
__global__ __launch_bounds__(256) void im2col3x3_kernel(const T* __restrict__ X, long ldx, T* __restrict__ Y, long ldy, int B,
                                                        int H, int Wd, int C) {
    constexpr int NV = 16 / sizeof(T);
    const int ncg = C / NV;
    const long total = (long)B * H * Wd * 9 * ncg;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(idx % ncg);
        long r = idx / ncg;
        const int tap = (int)(r % 9);
        r /= 9;
        const int x = (int)(r % Wd);
        r /= Wd;
        const int y = (int)(r % H);
        const long b = r / H;
        const int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)Wd)
            v = *reinterpret_cast<const uint4*>(X + ((b * H + iy) * (long)Wd + ix) * ldx + cg * NV);
        *reinterpret_cast<uint4*>(Y + ((b * H + y) * (long)Wd + x) * ldy + (long)tap * C + cg * NV) = v;
    }
}
}  // namespace

// ... and its adjoint: DX[b][y][x][c] = sum_taps DY[(b, y - (ky - 1), x - (kx - 1))][(ky * 3 + kx) * C + c] over the taps whose
// source pixel lies inside the image (a gather: every output element is written once, f32 sums, one rounding)
namespace {
template <typename T>
__global__ __launch_bounds__(256) void col2im3x3_kernel(const T* __restrict__ DY, long lddy, T* __restrict__ DX, long lddx, int B,
                                                        int H, int Wd, int C) {
    constexpr int NV = 16 / sizeof(T);
    const int ncg = C / NV;
    const long total = (long)B * H * Wd * ncg;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(idx % ncg);
        long r = idx / ncg;
        const int x = (int)(r % Wd);
        r /= Wd;
        const int y = (int)(r % H);
        const long b = r / H;
        float acc[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) acc[j] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int py = y - (tap / 3 - 1), px = x - (tap % 3 - 1);
            if ((unsigned)py < (unsigned)H && (unsigned)px < (unsigned)Wd) {
                const uint4 raw = *reinterpret_cast<const uint4*>(DY + ((b * H + py) * (long)Wd + px) * lddy + (long)tap * C + cg * NV);
                const T* tv = reinterpret_cast<const T*>(&raw);
#pragma unroll
                for (int j = 0; j < NV; ++j) acc[j] += to_f32<T>(tv[j]);
            }
        }
        uint4 out;
        T* ov = reinterpret_cast<T*>(&out);
#pragma unroll
        for (int j = 0; j < NV; ++j) ov[j] = from_f32<T>(acc[j]);
        *reinterpret_cast<uint4*>(DX + ((b * H + y) * (long)Wd + x) * lddx + cg * NV) = out;
    }
}
}  // namespace

extern "C" int emip_col2im3x3(const void* DY, long lddy, void* DX, long lddx, int B, int H, int Wd, int C, int dtype, void* stream) {
    REQ_DT(dtype);
    const int nv = dtype == EMIP_F32 ? 4 : 8;
    EMIP_REQUIRE(DY && DX && B > 0 && H > 0 && Wd > 0 && C >= nv && C % nv == 0 && lddx % nv == 0 && lddx >= C && lddy % nv == 0 &&
                 lddy >= 9L * C && aligned16(DY) && aligned16(DX));
    const long total = (long)B * H * Wd * (C / nv);
    DISPATCH_T(dtype, hipLaunchKernelGGL(col2im3x3_kernel<T>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                                         (const T*)DY, lddy, (T*)DX, lddx, B, H, Wd, C));
    return emip_launch_status();
}

extern "C" int emip_im2col3x3(const void* X, long ldx, void* Y, long ldy, int B, int H, int Wd, int C, int dtype, void* stream) {
    REQ_DT(dtype);
    const int nv = dtype == EMIP_F32 ? 4 : 8;
    EMIP_REQUIRE(X && Y && B > 0 && H > 0 && Wd > 0 && C >= nv && C % nv == 0 && ldx % nv == 0 && ldx >= C && ldy % nv == 0 &&
                 ldy >= 9L * C && aligned16(X) && aligned16(Y));
    const long total = (long)B * H * Wd * 9 * (C / nv);
    DISPATCH_T(dtype, hipLaunchKernelGGL(im2col3x3_kernel<T>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                                         (const T*)X, ldx, (T*)Y, ldy, B, H, Wd, C));
    return emip_launch_status();
}

extern "C" int emip_dwconv3x3_gated(const void* X, long ldx, void* Y, long ldy, const float* Wt, const float* bias,
                                    int B, int H, int Wd, int C2, int Cout_pad, int dtype, void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(X && Y && Wt && B > 0 && H > 0 && Wd > 0 && C2 >= 8 && (C2 & 7) == 0);
    EMIP_REQUIRE((ldx & 3) == 0 && (ldy & 3) == 0 && ldx >= C2 && (Cout_pad & 3) == 0 && Cout_pad >= C2 / 2 &&
                 ldy >= Cout_pad);
    if (dtype == EMIP_BF16 && ((C2 >> 1) & 3) == 0 && (reinterpret_cast<uintptr_t>(X) & 7u) == 0 &&
        (reinterpret_cast<uintptr_t>(Y) & 7u) == 0 && aligned16(Wt) && (!bias || aligned16(bias))) {
        const long work = (long)B * H * ((Wd + 3) / 4) * (Cout_pad >> 2);
        hipLaunchKernelGGL((dwconv3x3_gated_rows_kernel<4>), dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)X, ldx, (bf16_t*)Y, ldy, Wt, bias, B, H, Wd, C2, Cout_pad);
        return emip_launch_status();
    }
    const long total = (long)B * H * Wd * (Cout_pad >> 2);
    DISPATCH_T(dtype, hipLaunchKernelGGL((dwconv3x3_kernel<T, true>), dim3(grid_for(total, 256)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)X, ldx, (T*)Y, ldy, Wt, bias, B, H, Wd, C2,
                                         Cout_pad, 0));
    return emip_launch_status();
}

extern "C" int emip_chan_stats(const void* X, long ldx, double* sums, long groups, long rows, int C, int prezeroed,
                               int dtype, void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(X && sums && groups > 0 && groups < 65536 && rows > 0 && C >= 4 && C <= 1024 && (C & 3) == 0);
    EMIP_REQUIRE((ldx & 3) == 0 && ldx >= C);
    hipStream_t s = (hipStream_t)stream;
    if (!prezeroed && emip_zero_async(sums, sizeof(double) * 2 * groups * C, s) != EMIP_OK) return EMIP_E_LAUNCH;
    const int rpb = 512;
    dim3 grid((unsigned)((rows + rpb - 1) / rpb), (unsigned)groups);
    if (dtype == EMIP_BF16 && (C & 7) == 0 && C <= 128 && (ldx & 7) == 0 && aligned16(X)) {
        // at least ~512 workgroups: the 44 x 44 maps (1936 rows x 16 images) had 64 of them with 512 rows each
        int wrpb = 512;
        while (wrpb > 64 && ((rows + wrpb - 1) / wrpb) * groups < 512) wrpb >>= 1;
        hipLaunchKernelGGL(chan_stats_wide_kernel, dim3((unsigned)((rows + wrpb - 1) / wrpb), (unsigned)groups), dim3(256), 0, s,
                           (const bf16_t*)X, ldx, sums, rows, C, wrpb);
        return emip_launch_status();
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL(chan_stats_kernel<T>, grid, dim3(256), 0, s, (const T*)X, ldx, sums, rows, C,
                                         rpb));
    return emip_launch_status();
}

extern "C" int emip_chan_norm_apply(const void* X, long ldx, void* Y, long ldy, const void* R, long ldr,
                                    const double* sums, const float* gamma, const float* beta, long groups, long rows,
                                    int C, float eps, int relu_inner, int relu_outer, int dtype, void* stream) {
    return emip_chan_norm_apply_res(X, ldx, Y, ldy, R, ldr, sums, gamma, beta, groups, rows, C, eps, relu_inner, relu_outer, nullptr,
                                    0, dtype, stream);
}

extern "C" int emip_chan_norm_apply_res(const void* X, long ldx, void* Y, long ldy, const void* R, long ldr,
                                        const double* sums, const float* gamma, const float* beta, long groups, long rows,
                                        int C, float eps, int relu_inner, int relu_outer, const double* res_sums, int res_relu,
                                        int dtype, void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(!res_sums || R);
    EMIP_REQUIRE(X && Y && sums && groups > 0 && rows > 0 && C >= 4 && (C & 3) == 0);
    EMIP_REQUIRE((ldx & 3) == 0 && (ldy & 3) == 0 && ldx >= C && ldy >= C);
    EMIP_REQUIRE((gamma == nullptr) == (beta == nullptr));
    if (R) EMIP_REQUIRE((ldr & 3) == 0 && ldr >= C);
    const int vec = dtype == EMIP_F32 ? 4 : 8;
    if (C % vec == 0 && C / vec <= 256 && C <= 1024 && ldx % vec == 0 && ldy % vec == 0 && (!R || ldr % vec == 0) && aligned16(X) &&
        aligned16(Y) && (!R || aligned16(R)) && groups < 65536) {
        const int rl = 256 / (C / vec);
        long chunks = (2048 + groups - 1) / groups;                    // about 2048 workgroups, at least 4 rows per thread
        const long max_chunks = (rows + 4L * rl - 1) / (4L * rl);
        if (chunks > max_chunks) chunks = max_chunks;
        if (chunks < 1) chunks = 1;
        DISPATCH_T(dtype, hipLaunchKernelGGL(chan_norm_apply_wide_kernel<T>, dim3((unsigned)chunks, (unsigned)groups),
                                             dim3(256), 0, (hipStream_t)stream, (const T*)X, ldx, (T*)Y, ldy, (const T*)R,
                                             ldr, sums, gamma, beta, rows, C, eps, relu_inner, relu_outer, res_sums, res_relu));
        return emip_launch_status();
    }
    EMIP_REQUIRE(!res_sums);               // the raw-residual form exists on the wide kernel only
    const long total = groups * rows * (C >> 2);
    DISPATCH_T(dtype, hipLaunchKernelGGL(chan_norm_apply_kernel<T>, dim3(grid_for(total, 256)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)X, ldx, (T*)Y, ldy, (const T*)R, ldr, sums,
                                         gamma, beta, groups, rows, C, eps, relu_inner, relu_outer));
    return emip_launch_status();
}

extern "C" int emip_bilinear(const void* X, long ldx, void* Y, long ldy, int B, int H, int Wd, int C, int Ho, int Wo,
                             int align_corners, float mul, int dtype, void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(X && Y && B > 0 && H > 0 && Wd > 0 && Ho > 0 && Wo > 0 && C >= 4 && (C & 3) == 0);
    EMIP_REQUIRE((ldx & 3) == 0 && (ldy & 3) == 0 && ldx >= C && ldy >= C);
    const long total = (long)B * Ho * Wo * (C >> 2);
    DISPATCH_T(dtype, hipLaunchKernelGGL(bilinear_kernel<T>, dim3(grid_for(total, 256)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)X, ldx, (T*)Y, ldy, B, H, Wd, C, Ho, Wo,
                                         align_corners, mul));
    return emip_launch_status();
}

extern "C" int emip_bilinear_planar(const void* X, long ldx, int xc, float* Y, int B, int H, int Wd, int C, int Ho,
                                    int Wo, int align_corners, float mul, int dtype, void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(X && Y && B > 0 && H > 0 && Wd > 0 && Ho > 0 && Wo > 0 && C >= 1 && xc >= 0 && ldx >= xc + C);
    const long total = (long)B * C * Ho * Wo;
    DISPATCH_T(dtype, hipLaunchKernelGGL(bilinear_plane_kernel<T>, dim3(grid_for(total, 256)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)X, ldx, xc, Y, B, H, Wd, C, Ho, Wo,
                                         align_corners, mul));
    return emip_launch_status();
}

extern "C" int emip_eltwise(const void* A, long lda, const void* Bp, long ldb, const void* C3, long ldc3, void* Y,
                            long ldy, long M, int C, int mode, long period, int dtype, void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(A && Bp && Y && M > 0 && C >= 4 && (C & 3) == 0 && mode >= 0 && mode <= 3);
    EMIP_REQUIRE((lda & 3) == 0 && (ldb & 3) == 0 && (ldy & 3) == 0 && lda >= C && ldb >= C && ldy >= C);
    if (mode == 1) EMIP_REQUIRE(C3 && (ldc3 & 3) == 0 && ldc3 >= C);
    if (mode == 3) EMIP_REQUIRE(period > 0);
    const long total = M * (C >> 2);
    DISPATCH_T(dtype, hipLaunchKernelGGL(eltwise_kernel<T>, dim3(grid_for(total, 256)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)A, lda, (const T*)Bp, ldb, (const T*)C3, ldc3,
                                         (T*)Y, ldy, M, C, mode, period));
    return emip_launch_status();
}

extern "C" int emip_planar_to_cl(const float* X, void* Y, long ldy, int B, int C, long P, int Cpad, int dtype,
                                 void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(X && Y && B > 0 && B < 65536 && C > 0 && P > 0 && Cpad >= C && ldy >= Cpad);
    dim3 grid((unsigned)((P + 31) / 32), (unsigned)((Cpad + 31) / 32), (unsigned)B);
    EMIP_REQUIRE(grid.y < 65536);
    if (dtype == EMIP_BF16 && C <= 8 && Cpad == 8 && ldy == 8 && aligned16(Y)) {
        hipLaunchKernelGGL(planar_to_cl8_kernel, dim3(grid_for((long)B * P, 256)), dim3(256), 0, (hipStream_t)stream, X,
                           (bf16_t*)Y, B, C, P);
        return emip_launch_status();
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL(planar_to_cl_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, X, (T*)Y,
                                         ldy, B, C, P, Cpad));
    return emip_launch_status();
}

extern "C" int emip_cl_to_planar(const void* X, long ldx, int xc, float* Y, int B, int C, long P, int dtype,
                                 void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(X && Y && B > 0 && B < 65536 && C > 0 && P > 0 && xc >= 0 && ldx >= xc + C);
    dim3 grid((unsigned)((P + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)B);
    EMIP_REQUIRE(grid.y < 65536);
    DISPATCH_T(dtype, hipLaunchKernelGGL(cl_to_planar_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)X,
                                         ldx, xc, Y, B, C, P));
    return emip_launch_status();
}

extern "C" int emip_copy_cols(const void* X, long ldx, int xc, int x_dtype, void* Y, long ldy, int yc, int y_dtype,
                              long M, int C, int Cpad, void* stream) {
    REQ_DT(x_dtype);
    REQ_DT(y_dtype);
    EMIP_REQUIRE(X && Y && M > 0 && C > 0 && Cpad >= C && xc >= 0 && yc >= 0 && ldx >= xc + C && ldy >= yc + Cpad);
    const long total = M * Cpad;
    const dim3 g(grid_for(total, 256)), b(256);
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == EMIP_F32 && y_dtype == EMIP_F32)
        hipLaunchKernelGGL((copy_cols_kernel<float, float>), g, b, 0, s, (const float*)X, ldx, xc, (float*)Y, ldy, yc,
                           M, C, Cpad);
    else if (x_dtype == EMIP_F32)
        hipLaunchKernelGGL((copy_cols_kernel<float, bf16_t>), g, b, 0, s, (const float*)X, ldx, xc, (bf16_t*)Y, ldy,
                           yc, M, C, Cpad);
    else if (y_dtype == EMIP_F32)
        hipLaunchKernelGGL((copy_cols_kernel<bf16_t, float>), g, b, 0, s, (const bf16_t*)X, ldx, xc, (float*)Y, ldy,
                           yc, M, C, Cpad);
    else
        hipLaunchKernelGGL((copy_cols_kernel<bf16_t, bf16_t>), g, b, 0, s, (const bf16_t*)X, ldx, xc, (bf16_t*)Y, ldy,
                           yc, M, C, Cpad);
    return emip_launch_status();
}

extern "C" int emip_convex_upsample(const void* logits, long ldl, const float* flow, float* out, int N, int H, int Wd,
                                    int dtype, void* stream) {
    REQ_DT(dtype);
    EMIP_REQUIRE(logits && flow && out && N > 0 && H > 0 && Wd > 0 && ldl >= 576);
    const long total = (long)N * H * Wd * 64;
    DISPATCH_T(dtype, hipLaunchKernelGGL(convex_up_kernel<T>, dim3(grid_for(total, 256)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)logits, ldl, flow, out, N, H, Wd));
    return emip_launch_status();
}

extern "C" int emip_corresp_to_flow(const float* O, long ldo, float* flow, long N, int H, int Wd, int sub_grid,
                                    void* stream) {
    EMIP_REQUIRE(O && flow && N > 0 && H > 0 && Wd > 0 && ldo >= 2);
    const long total = N * H * Wd;
    hipLaunchKernelGGL(corresp_to_flow_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, O, ldo,
                       flow, N, H * Wd, Wd, sub_grid);
    return emip_launch_status();
}
