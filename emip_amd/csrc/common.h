// Shared device/host helpers for libemip_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/emip_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define EMIP_WAVE 64

// Launch-time argument checks: a kernel that faults can reset the whole host,
// so every shape/alignment assumption is verified here and refused with a code.
#define EMIP_REQUIRE(cond)                 \
    do {                                   \
        if (!(cond)) return EMIP_E_INVALID; \
    } while (0)

static inline int emip_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? EMIP_OK : EMIP_E_LAUNCH;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <typename T>
__device__ __forceinline__ float to_f32(T v);
template <>
__device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }

template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7): one rcp, one exp and a degree-5 Horner chain instead of
// libm's branchy erff (~60 VALU ops), which made the GELU-carrying kernels VALU-bound.  Used where the
// activation is stored as bf16 (its rounding step is 4e-3 relative); f32 parity mode keeps erff.
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));   // raw v_rcp_f32 (1 ulp)
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);   // raw v_exp_f32
    const float r = fmaf(-p * t, e, 1.0f);
    return copysignf(r, x);
}
template <typename T>
__device__ __forceinline__ float gelu_t(float x);
template <>
__device__ __forceinline__ float gelu_t<float>(float x) { return gelu_erf(x); }
// bf16 storage: gelu(x) = x Phi(x) with Phi(x) ~ 0.5 + t P(t^2), t = clamp(x, -4, 4), P of degree 6 fitted (minimax on
// x dPhi) under the constraint Phi(+-4) = 1 / 0, so the tails are exact (x and 0).  |error| <= 1.9e-4 absolute, below the
// bf16 rounding step of every |gelu| > 0.05; 10 full-rate VALU slots per PAIR of elements (hipcc packs the chain into
// v_pk_fma_f32) against 30 for the exp + rcp form, which made the depthwise + GELU kernel VALU-bound.
__device__ __forceinline__ float gelu_poly(float x) {
    const float t = __builtin_amdgcn_fmed3f(x, -4.0f, 4.0f);
    const float u = t * t;
    float p = fmaf(2.258824939e-08f, u, -1.588829207e-06f);
    p = fmaf(p, u, 4.776392641e-05f);
    p = fmaf(p, u, -8.121878305e-04f);
    p = fmaf(p, u, 8.763692481e-03f);
    p = fmaf(p, u, -6.455441459e-02f);
    p = fmaf(p, u, 3.978702669e-01f);
    return x * fmaf(t, p, 0.5f);
}
template <>
__device__ __forceinline__ float gelu_t<bf16_t>(float x) { return gelu_poly(x); }

// d gelu(z) / dz = Phi(z) + z phi(z).  f32 parity mode keeps libm's erff / expf.
template <typename T>
__device__ __forceinline__ float gelu_grad_t(float z);
template <>
__device__ __forceinline__ float gelu_grad_t<float>(float z) {
    const float cdf = 0.5f * (1.f + erff(z * 0.70710678118654752440f));
    const float pdf = 0.3989422804014327f * expf(-0.5f * z * z);
    return cdf + z * pdf;
}
// bf16 storage: Phi(z) - 0.5 and z phi(z) are both odd, so gelu'(z) = 0.5 + t Q(t^2) with t = clamp(z, -4, 4) and Q of degree 8
// (weighted least squares towards minimax on [0, 4]; |error| <= 1e-4 on a function of range [-0.13, 1.13], 5.6e-4 in the
// clamped tails), 11 full-rate VALU slots per element that hipcc packs two by two -- the exp + rcp form (one shared exponential
// between the Abramowitz-Stegun erf and the density) took ~25 slots plus two quarter-rate transcendentals and made the
// backward of the depthwise + GELU pair VALU-bound.
__device__ __forceinline__ float gelu_grad_poly(float z) {
    const float t = __builtin_amdgcn_fmed3f(z, -4.0f, 4.0f);
    const float u = t * t;
    float p = fmaf(9.796147385e-10f, u, -8.218867416e-08f);
    p = fmaf(p, u, 3.028366673e-06f);
    p = fmaf(p, u, -6.495786215e-05f);
    p = fmaf(p, u, 9.073290484e-04f);
    p = fmaf(p, u, -8.716332020e-03f);
    p = fmaf(p, u, 5.845612905e-02f);
    p = fmaf(p, u, -2.648265643e-01f);
    p = fmaf(p, u, 7.976095594e-01f);
    return fmaf(t, p, 0.5f);
}
template <>
__device__ __forceinline__ float gelu_grad_t<bf16_t>(float z) { return gelu_grad_poly(z); }

// 4 consecutive elements as a vector (16 B for f32, 8 B for bf16)
template <typename T>
struct Vec4;
template <>
struct Vec4<float> {
    typedef float4 type;
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
        float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <>
struct Vec4<bf16_t> {
    static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[4]) {
        bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
        v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[4]) {
        bf16x4 t;
        t[0] = (bf16_t)v[0]; t[1] = (bf16_t)v[1]; t[2] = (bf16_t)v[2]; t[3] = (bf16_t)v[3];
        *reinterpret_cast<bf16x4*>(p) = t;
    }
};

// v if ok else 0, component-wise AND with a lane mask (a ternary on the 16-byte struct makes hipcc build an
// indexed select through scratch memory).
__device__ __forceinline__ uint4 mask4(uint4 v, bool ok) {
    const unsigned m = ok ? 0xFFFFFFFFu : 0u;
    v.x &= m; v.y &= m; v.z &= m; v.w &= m;
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Bijective XCD-aware remap of a linear workgroup id: blocks are dealt round-robin
// over the 8 XCDs, so give each XCD a contiguous chunk of the tile space (speed only).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// Zero-fill as a KERNEL node.  hipMemsetAsync captured into a hipGraph was observed to mis-order against the
// following kernels when two graphs replay concurrently (GMFlow CNN statistics came out wrong under multi-stream
// replay while eager runs were exact), so every scratch buffer is cleared by this kernel instead.
static __global__ __launch_bounds__(256) void emip_zero_kernel(unsigned* __restrict__ p, size_t nwords) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x)
        p[i] = 0u;
}
static inline int emip_zero_async(void* ptr, size_t bytes, hipStream_t s) {
    if (bytes == 0) return EMIP_OK;
    if ((reinterpret_cast<uintptr_t>(ptr) & 3u) || (bytes & 3u)) return EMIP_E_INVALID;
    const size_t nwords = bytes / 4;
    size_t blocks = (nwords + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(emip_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<unsigned*>(ptr), nwords);
    return emip_launch_status();
}

// Index arithmetic of the grid-stride loops over planar / channels-last tensors.  A 64-bit division costs ~150 VALU
// instructions on gfx950 (no hardware divider, wave64 on 16-lane SIMDs): `x = idx % W, y = idx / W % H, b = idx / (H * W)`
// written with `long` made the loss kernels VALU-bound (flow_warp: 181 us for 127 MB at batch 32).  IdxDiv divides in 32 bits
// whenever the loop's extent allows it (a wave-uniform choice made once per kernel) and the callers chain quotients
// (idx / W, then that / H) instead of dividing idx three times.
struct IdxDiv {
    bool small;
    __device__ __forceinline__ explicit IdxDiv(long total) : small(total < (1L << 31)) {}
    __device__ __forceinline__ long div(long n, long d) const {
        return small ? (long)((unsigned)n / (unsigned)d) : n / d;
    }
};
