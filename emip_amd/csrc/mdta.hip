// Channel ("transposed") attention pieces of the MDTA injector block
// (/root/reference/model/EMIP_short/motion/PromptInteract.py:407-432):
//   gram:    G[b,h,c1,c2] = sum_p q[b,p,h*64+c1] * k[b,p,h*64+c2]   (contraction over 1936 pixels)
//            nq[b,c] = sum_p q^2,  nk[b,c] = sum_p k^2              (for F.normalize over pixels)
//   softmax: attn = softmax_c2( G / (max(|q_c1|,eps) max(|k_c2|,eps)) * temperature[h] )
// attn @ v and project_out are plain GEMMs (emip_gemm).  The Gram matrix is tiny
// (64x64 per head); the work is the HBM pass over q and k, so pixels are split over
// up to MDTA_PARTS workgroups per matrix.  Round 4: every workgroup STORES its partial
// matrix / norms into a slot of its own and the softmax kernel adds the slots in slot
// order (rounds 1-3 combined them with f32 atomics: the sums -- and through a bf16
// rounding now and then the whole forward -- differed from run to run).
#include "common.h"

namespace {

constexpr int PC = 64;  // pixels per tile
constexpr int MDTA_PARTS = 8;   // pixel ranges (= workgroups, = partial-sum slots) per Gram matrix at most
constexpr int MDTA_REC = 4096 + 128;   // floats of one record: G [64][64] | nq [64] | nk [64]

template <typename T>
__global__ __launch_bounds__(256) void mdta_gram_kernel(const T* __restrict__ Q, long ldq, long q_bs,
                                                        const T* __restrict__ K, long ldk, long k_bs,
                                                        float* __restrict__ G, float* __restrict__ nq,
                                                        float* __restrict__ nk, int P, int heads, int chunks) {
    __shared__ __attribute__((aligned(16))) float sq[PC][64];
    __shared__ __attribute__((aligned(16))) float sk[PC][64];
    const int head = blockIdx.y;
    const long b = blockIdx.z;
    const T* q = Q + b * q_bs + head * 64;
    const T* k = K + b * k_bs + head * 64;
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    float sn = 0.f;
    // `chunks` tiles of 64 pixels per workgroup, one round of atomics at the end: with one tile per workgroup the 31 x 4096
    // atomics per Gram matrix (same 32 cache lines) were the launch -- 50 us for 0.5 GFLOP
    for (int ch = 0; ch < chunks; ++ch) {
        const int p0 = (blockIdx.x * chunks + ch) * PC;
        if (p0 >= P) break;
        if (ch) __syncthreads();
        for (int i = threadIdx.x; i < PC * 16; i += 256) {
            const int r = i >> 4, c = (i & 15) * 4;
            float a[4] = {0, 0, 0, 0}, d[4] = {0, 0, 0, 0};
            if (p0 + r < P) {
                Vec4<T>::load(q + (long)(p0 + r) * ldq + c, a);
                Vec4<T>::load(k + (long)(p0 + r) * ldk + c, d);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sq[r][c + j] = a[j];
                sk[r][c + j] = d[j];
            }
        }
        __syncthreads();
        for (int r = 0; r < PC; ++r) {
            const float4 a = *reinterpret_cast<const float4*>(&sq[r][ty * 4]);
            const float4 d = *reinterpret_cast<const float4*>(&sk[r][tx * 4]);
            const float av[4] = {a.x, a.y, a.z, a.w}, dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], dv[j], acc[i][j]);
        }
        if (threadIdx.x < 128) {
            const int c = threadIdx.x & 63;
            const bool isq = threadIdx.x < 64;
            for (int r = 0; r < PC; ++r) {
                const float v = isq ? sq[r][c] : sk[r][c];
                sn = fmaf(v, v, sn);
            }
        }
    }
    // this workgroup's slot: parts[(b * heads + head) * gridDim.x + blockIdx.x] = [G | nq | nk]  (G, nq, nk arrive as the
    // base of the slot array, its + 4096 and its + 4096 + 64)
    const long slot = ((b * heads + head) * (long)gridDim.x + blockIdx.x) * MDTA_REC;
    float* g = G + slot;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        *reinterpret_cast<float4*>(g + (ty * 4 + i) * 64 + tx * 4) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
    if (threadIdx.x < 128) (threadIdx.x < 64 ? nq : nk)[slot + (threadIdx.x & 63)] = sn;
}

// parts: [nbh][nparts][G | nq | nk] partial sums; G / nq / nk: the complete sums, written here (the backward reads them)
template <typename T>
__global__ __launch_bounds__(64) void mdta_softmax_kernel(const float* __restrict__ parts, int nparts, float* __restrict__ G,
                                                          float* __restrict__ nq, float* __restrict__ nk,
                                                          const float* __restrict__ temperature,
                                                          T* __restrict__ attn, int heads) {
    __shared__ float skn[64];
    const long bh = blockIdx.x;
    const int head = (int)(bh % heads);
    const int c1 = threadIdx.x;
    const float* pp = parts + bh * nparts * MDTA_REC;
    float v[64];
#pragma unroll
    for (int c2 = 0; c2 < 64; ++c2) v[c2] = 0.f;
    float sq = 0.f, sk = 0.f;
    for (int s = 0; s < nparts; ++s) {                     // slot order: the same sums in every run
        const float* rec = pp + (long)s * MDTA_REC;
#pragma unroll
        for (int c2 = 0; c2 < 64; c2 += 4) {
            const float4 t = *reinterpret_cast<const float4*>(rec + c1 * 64 + c2);
            v[c2] += t.x; v[c2 + 1] += t.y; v[c2 + 2] += t.z; v[c2 + 3] += t.w;
        }
        sq += rec[4096 + c1];
        sk += rec[4096 + 64 + c1];
    }
    float* g = G + (bh * 64 + c1) * 64;
#pragma unroll
    for (int c2 = 0; c2 < 64; c2 += 4) *reinterpret_cast<float4*>(g + c2) = make_float4(v[c2], v[c2 + 1], v[c2 + 2], v[c2 + 3]);
    nq[bh * 64 + c1] = sq;
    nk[bh * 64 + c1] = sk;
    skn[c1] = fmaxf(sqrtf(sk), 1e-12f);
    __syncthreads();
    const float qn = fmaxf(sqrtf(sq), 1e-12f);
    const float temp = temperature[head];
    float mx = -INFINITY;
#pragma unroll
    for (int c2 = 0; c2 < 64; ++c2) {
        v[c2] = v[c2] / (qn * skn[c2]) * temp;
        mx = fmaxf(mx, v[c2]);
    }
    float den = 0.f;
#pragma unroll
    for (int c2 = 0; c2 < 64; ++c2) {
        v[c2] = expf(v[c2] - mx);
        den += v[c2];
    }
    const float inv = 1.f / den;
    T* o = attn + (bh * 64 + c1) * 64;
#pragma unroll
    for (int c2 = 0; c2 < 64; ++c2) o[c2] = from_f32<T>(v[c2] * inv);
}

}  // namespace

// floats of the workspace of emip_mdta_attn: the complete sums [G | nq | nk] of every (image, head), then their partial-sum slots
extern "C" int emip_mdta_ws_floats(int B, int heads) {
    const long n = (long)B * heads * MDTA_REC * (1 + MDTA_PARTS);
    return n < 0x7FFFFFFFL ? (int)n : -1;
}

// ws: f32 workspace of emip_mdta_ws_floats(B, heads) floats; its first B*heads*(64*64 + 128) hold [G | nq | nk] afterwards (the
// layout emip_mdta_bwd_small reads); attn: T [B][heads][64][64]
extern "C" int emip_mdta_attn(const void* Q, long ldq, long q_bs, const void* K, long ldk, long k_bs,
                              const float* temperature, float* ws, void* attn, int B, int heads, int P, int dtype,
                              void* stream) {
    EMIP_REQUIRE(Q && K && temperature && ws && attn && B > 0 && B < 65536 && heads > 0 && heads < 65536 && P > 0);
    EMIP_REQUIRE(dtype == EMIP_F32 || dtype == EMIP_BF16);
    EMIP_REQUIRE((ldq & 3) == 0 && (ldk & 3) == 0 && ldq >= heads * 64 && ldk >= heads * 64 && (q_bs & 3) == 0 &&
                 (k_bs & 3) == 0);
    hipStream_t s = (hipStream_t)stream;
    const long nbh = (long)B * heads;
    EMIP_REQUIRE(emip_mdta_ws_floats(B, heads) > 0);
    float* G = ws;
    float* nq = ws + nbh * 4096;
    float* nk = nq + nbh * 64;
    float* parts = ws + nbh * MDTA_REC;
    // pixel tiles per workgroup: as many as leave ~256 workgroups, and at most MDTA_PARTS workgroups per matrix
    const int tiles = (P + PC - 1) / PC;
    int chunks = (int)((long)tiles * nbh / 256);
    if (chunks < (tiles + MDTA_PARTS - 1) / MDTA_PARTS) chunks = (tiles + MDTA_PARTS - 1) / MDTA_PARTS;
    dim3 grid((tiles + chunks - 1) / chunks, heads, B);
    const int nparts = (int)grid.x;
    if (dtype == EMIP_F32) {
        hipLaunchKernelGGL(mdta_gram_kernel<float>, grid, dim3(256), 0, s, (const float*)Q, ldq, q_bs,
                           (const float*)K, ldk, k_bs, parts, parts + 4096, parts + 4096 + 64, P, heads, chunks);
        hipLaunchKernelGGL(mdta_softmax_kernel<float>, dim3((unsigned)nbh), dim3(64), 0, s, parts, nparts, G, nq, nk, temperature,
                           (float*)attn, heads);
    } else {
        hipLaunchKernelGGL(mdta_gram_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)Q, ldq, q_bs,
                           (const bf16_t*)K, ldk, k_bs, parts, parts + 4096, parts + 4096 + 64, P, heads, chunks);
        hipLaunchKernelGGL(mdta_softmax_kernel<bf16_t>, dim3((unsigned)nbh), dim3(64), 0, s, parts, nparts, G, nq, nk, temperature,
                           (bf16_t*)attn, heads);
    }
    return emip_launch_status();
}
