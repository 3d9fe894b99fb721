// Fused head of the PVTv2 Mlp for gfx950:   G = GELU(dwconv3x3(X W1^T + b1) + b_dw)
// (/root/reference/lib/pvt_v2.py:45-54,316-327: fc1 -> DWConv -> GELU; fc2 then reads G through emip_gemm).
//
// The fc1 output H [tokens][4C] is the largest activation of the network and unfused it crosses HBM twice between fc1 and
// the depthwise kernel (written, then read back with its 3x3 halo).  Here a workgroup owns a BAND of image rows and one
// 128-byte slab of hidden channels (64 bf16 / 32 f32): it computes H for the band plus one halo row above and below
// (recomputed by the neighbouring band: +22 % fc1 FLOPs at 44x44, none when the whole image is one band), keeps it in LDS,
// runs the depthwise 3x3 + GELU out of LDS and writes only G.  H never reaches HBM.
//
//  * GEMM phase: no operand staging at all -- waves split the band's tokens, so an activation fragment is used by exactly
//    one wave and is loaded straight from global memory in MFMA layout (16-byte loads); the 64 x K weight slab is tiny and
//    comes from L2.  Swapped operands as in gemm.hip: a lane ends up with 4 consecutive hidden channels of one token.
//  * H tile in LDS: [token][128 B], 16-byte chunks XOR-swizzled by ((token >> 1) & 7) so that the accumulator-layout writes
//    and the row-major depthwise reads are both conflict-free.
//  * depthwise phase: a thread keeps ONE 16-byte channel chunk for the whole band, so its 9 x VEC weights and bias are
//    loaded once into registers; neighbours are ds_read_b128.
#include "common.h"

namespace {

template <typename T>
struct MmaH;
template <>
struct MmaH<bf16_t> {
    static __device__ __forceinline__ f32x4 run(const uint4& a, const uint4& b, f32x4 acc) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                       acc, 0, 0, 0);
    }
};
template <>
struct MmaH<float> {
    static __device__ __forceinline__ f32x4 run(const uint4& a, const uint4& b, f32x4 acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
        return acc;
    }
};

struct HeadArgs {
    const void* X;      // [B][H][W][ldx]  (the LayerNorm-ed tokens)
    const void* W1;     // [Ch][C]
    const float* b1;    // [Ch]
    const float* Wt;    // depthwise weights [9][Ch]
    const float* bdw;   // [Ch]
    void* G;            // [B][H][W][ldg]
    long ldx, ldg;
    int B, H, W, C, Ch, R;
};

constexpr int HEAD_MAXM = 512;   // tokens per band tile (64 KB of LDS)
constexpr int HEAD_RT = 8;       // 16-token MFMA row tiles per wave (4 waves x 8 x 16 = 512)

template <typename T>
__global__ __launch_bounds__(256) void mlp_head_kernel(const HeadArgs p) {
    constexpr int ES = sizeof(T);
    constexpr int VEC = 16 / ES;          // elements per 16-byte chunk
    constexpr int NC = 128 / ES;          // hidden channels per workgroup (one 128-byte slab)
    constexpr int NT = NC / 16;           // 16-channel MFMA tiles
    extern __shared__ __attribute__((aligned(16))) char Hs[];    // [HEAD_MAXM][128 B]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * NC;
    const int y0 = blockIdx.y * p.R;
    const long b = blockIdx.z;
    const int rows_out = min(p.R, p.H - y0);
    const int ty_lo = max(y0 - 1, 0), ty_hi = min(y0 + rows_out, p.H - 1);      // image rows held in the tile
    const int m_tile = (ty_hi - ty_lo + 1) * p.W;
    const int n_rt = (m_tile + 15) >> 4;

    const T* __restrict__ X = reinterpret_cast<const T*>(p.X) + (b * p.H + ty_lo) * (long)p.W * p.ldx;
    const T* __restrict__ W1 = reinterpret_cast<const T*>(p.W1);

    // ---- GEMM phase: H[token][n0..n0+NC) = X[token][:] . W1[n][:]
    f32x4 acc[HEAD_RT][NT];
#pragma unroll
    for (int r = 0; r < HEAD_RT; ++r)
#pragma unroll
        for (int a = 0; a < NT; ++a) acc[r][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
    long arow[HEAD_RT];      // element offset of this lane's token row (clamped), per row tile
    bool aok[HEAD_RT];
#pragma unroll
    for (int r = 0; r < HEAD_RT; ++r) {
        const int t = 16 * (wave + 4 * r) + fr;
        aok[r] = t < m_tile;
        arow[r] = (long)(aok[r] ? t : 0) * p.ldx;
    }
    const T* wrow[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) wrow[a] = W1 + (long)(n0 + 16 * a + fr) * p.C;
    for (int k0 = 0; k0 < p.C; k0 += NC) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int kc = k0 + (4 * g + fq) * VEC;
            uint4 fw[NT], fa[HEAD_RT];
#pragma unroll
            for (int a = 0; a < NT; ++a) fw[a] = *reinterpret_cast<const uint4*>(wrow[a] + kc);
#pragma unroll
            for (int r = 0; r < HEAD_RT; ++r)
                fa[r] = mask4(*reinterpret_cast<const uint4*>(X + arow[r] + kc), aok[r]);
#pragma unroll
            for (int r = 0; r < HEAD_RT; ++r) {
                if (wave + 4 * r < n_rt) {          // wave-uniform
#pragma unroll
                    for (int a = 0; a < NT; ++a) acc[r][a] = MmaH<T>::run(fw[a], fa[r], acc[r][a]);
                }
            }
        }
    }

    // ---- depthwise weights of this thread's channel chunk (issued now, consumed after the barrier)
    const int dc = tid & 7;                      // 16-byte channel chunk owned in the depthwise phase
    const int cch = n0 + dc * VEC;
    float wv[9][VEC], bv[VEC];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < VEC; j += 4) {
            const float4 v = *reinterpret_cast<const float4*>(p.Wt + (long)t * p.Ch + cch + j);
            wv[t][j] = v.x; wv[t][j + 1] = v.y; wv[t][j + 2] = v.z; wv[t][j + 3] = v.w;
        }
#pragma unroll
    for (int j = 0; j < VEC; j += 4) {
        const float4 v = *reinterpret_cast<const float4*>(p.bdw + cch + j);
        bv[j] = v.x; bv[j + 1] = v.y; bv[j + 2] = v.z; bv[j + 3] = v.w;
    }

    // ---- H (+ fc1 bias) -> LDS, rounded to the storage type exactly like the unfused fc1 output
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        const float4 bb = *reinterpret_cast<const float4*>(p.b1 + n0 + 16 * a + 4 * fq);
#pragma unroll
        for (int r = 0; r < HEAD_RT; ++r) {
            const int t = 16 * (wave + 4 * r) + fr;
            if (wave + 4 * r < n_rt && t < m_tile) {
                const float v[4] = {acc[r][a][0] + bb.x, acc[r][a][1] + bb.y, acc[r][a][2] + bb.z, acc[r][a][3] + bb.w};
                const int col = 16 * a + 4 * fq;                  // channel inside the slab
                const int chunk = col / VEC, within = col % VEC;
                char* dst = Hs + t * 128 + ((chunk ^ ((t >> 1) & 7)) * 16) + within * ES;
                Vec4<T>::store(reinterpret_cast<T*>(dst), v);
            }
        }
    }
    __syncthreads();

    // ---- depthwise 3x3 + GELU out of LDS
    T* __restrict__ G = reinterpret_cast<T*>(p.G) + (b * p.H) * (long)p.W * p.ldg;
    const int n_out = rows_out * p.W;
    for (int to = tid >> 3; to < n_out; to += 32) {
        const int oy = y0 + to / p.W, ox = to % p.W;
        float g[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) g[j] = bv[j];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const int tt = ok ? (iy - ty_lo) * p.W + ix : 0;      // clamped: the read stays unconditional
            const uint4 hv = mask4(*reinterpret_cast<const uint4*>(Hs + tt * 128 + ((dc ^ ((tt >> 1) & 7)) * 16)), ok);
            const T* hp = reinterpret_cast<const T*>(&hv);
#pragma unroll
            for (int j = 0; j < VEC; ++j) g[j] = fmaf(to_f32<T>(hp[j]), wv[t][j], g[j]);
        }
        uint4 ov;
        T* o = reinterpret_cast<T*>(&ov);
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = from_f32<T>(gelu_t<T>(g[j]));
        *reinterpret_cast<uint4*>(G + ((long)oy * p.W + ox) * p.ldg + cch) = ov;
    }
}


// ---- variant 2: 2-D patches, few accumulators, many co-resident workgroups ---------------------------------------------
// The band kernel above needs 350 registers per lane (one wave per SIMD) and its depthwise + GELU phase then runs with
// nothing to hide latencies.  Here a workgroup owns a PY x PX patch of output pixels (+1 halo ring it recomputes) and one
// 128-byte channel slab: RT 16-token MFMA row tiles per wave (RT = 2: 8x8 patch, 100 tokens, 32 accumulators), depthwise
// weights staged once in LDS, so 4-5 workgroups share a CU.
template <typename T, int RT>
__global__ __launch_bounds__(256) void mlp_head_patch_kernel(const HeadArgs p, int PY, int PX, int tiles_x) {
    constexpr int ES = sizeof(T);
    constexpr int VEC = 16 / ES;
    constexpr int NC = 128 / ES;
    constexpr int NT = NC / 16;
    constexpr int MAXM = 64 * RT;
    extern __shared__ __attribute__((aligned(16))) char smem2[];
    char* Hs = smem2;                                               // [MAXM][128 B]
    float* Wd = reinterpret_cast<float*>(smem2 + MAXM * 128);       // [10][NC]: 9 taps + bias

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * NC;
    const int py0 = (blockIdx.y / tiles_x) * PY, px0 = (blockIdx.y % tiles_x) * PX;
    const long b = blockIdx.z;
    const int TW = PX + 2;                                          // tile width incl. halo
    const int rows_out = min(PY, p.H - py0), cols_out = min(PX, p.W - px0);
    const int m_tile = (rows_out + 2) * TW;
    const int n_rt = (m_tile + 15) >> 4;

    const T* __restrict__ X = reinterpret_cast<const T*>(p.X) + (b * p.H) * (long)p.W * p.ldx;
    const T* __restrict__ W1 = reinterpret_cast<const T*>(p.W1);

    // depthwise weights + bias of this channel slab -> LDS (read back as broadcasts in the depthwise phase)
    for (int i = tid; i < 10 * NC; i += 256) {
        const int t = i / NC, c = i - t * NC;
        Wd[i] = t < 9 ? p.Wt[(long)t * p.Ch + n0 + c] : p.bdw[n0 + c];
    }

    f32x4 acc[RT][NT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int a = 0; a < NT; ++a) acc[r][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
    long arow[RT];
    bool aok[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int t = 16 * (wave + 4 * r) + fr;
        const int ly = t / TW, lx = t - ly * TW;
        const int gy = py0 - 1 + ly, gx = px0 - 1 + lx;
        aok[r] = t < m_tile && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        arow[r] = aok[r] ? ((long)gy * p.W + gx) * p.ldx : 0;
    }
    const T* wrow[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) wrow[a] = W1 + (long)(n0 + 16 * a + fr) * p.C;
    for (int k0 = 0; k0 < p.C; k0 += NC) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int kc = k0 + (4 * g + fq) * VEC;
            uint4 fw[NT], fa[RT];
#pragma unroll
            for (int a = 0; a < NT; ++a) fw[a] = *reinterpret_cast<const uint4*>(wrow[a] + kc);
#pragma unroll
            for (int r = 0; r < RT; ++r) fa[r] = mask4(*reinterpret_cast<const uint4*>(X + arow[r] + kc), aok[r]);
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                if (wave + 4 * r < n_rt) {
#pragma unroll
                    for (int a = 0; a < NT; ++a) acc[r][a] = MmaH<T>::run(fw[a], fa[r], acc[r][a]);
                }
            }
        }
    }
    // H (+ fc1 bias; ZERO outside the image: the depthwise conv pads H, not X) -> LDS
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        const float4 bb = *reinterpret_cast<const float4*>(p.b1 + n0 + 16 * a + 4 * fq);
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int t = 16 * (wave + 4 * r) + fr;
            if (wave + 4 * r < n_rt && t < m_tile) {
                const float m = aok[r] ? 1.f : 0.f;
                const float v[4] = {(acc[r][a][0] + bb.x) * m, (acc[r][a][1] + bb.y) * m, (acc[r][a][2] + bb.z) * m,
                                    (acc[r][a][3] + bb.w) * m};
                const int col = 16 * a + 4 * fq;
                const int chunk = col / VEC, within = col % VEC;
                Vec4<T>::store(reinterpret_cast<T*>(Hs + t * 128 + ((chunk ^ ((t >> 1) & 7)) * 16) + within * ES), v);
            }
        }
    }
    __syncthreads();

    // depthwise 3x3 + GELU out of LDS; thread = (output pixel, 16-byte channel chunk)
    T* __restrict__ G = reinterpret_cast<T*>(p.G) + (b * p.H) * (long)p.W * p.ldg;
    const int dc = tid & 7;
    const int n_out = rows_out * cols_out;
    for (int to = tid >> 3; to < n_out; to += 32) {
        const int oy = to / cols_out, ox = to - oy * cols_out;         // inside the patch
        float g[VEC];
#pragma unroll
        for (int j = 0; j < VEC; j += 4) {
            const float4 v = *reinterpret_cast<const float4*>(Wd + 9 * NC + dc * VEC + j);
            g[j] = v.x; g[j + 1] = v.y; g[j + 2] = v.z; g[j + 3] = v.w;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int tt = (oy + t / 3) * TW + ox + t % 3;             // halo ring is part of the tile: always in range
            const uint4 hv = *reinterpret_cast<const uint4*>(Hs + tt * 128 + ((dc ^ ((tt >> 1) & 7)) * 16));
            const T* hp = reinterpret_cast<const T*>(&hv);
#pragma unroll
            for (int j = 0; j < VEC; j += 4) {
                const float4 w = *reinterpret_cast<const float4*>(Wd + t * NC + dc * VEC + j);
                g[j] = fmaf(to_f32<T>(hp[j]), w.x, g[j]);
                g[j + 1] = fmaf(to_f32<T>(hp[j + 1]), w.y, g[j + 1]);
                g[j + 2] = fmaf(to_f32<T>(hp[j + 2]), w.z, g[j + 2]);
                g[j + 3] = fmaf(to_f32<T>(hp[j + 3]), w.w, g[j + 3]);
            }
        }
        uint4 ov;
        T* o = reinterpret_cast<T*>(&ov);
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = from_f32<T>(gelu_t<T>(g[j]));
        *reinterpret_cast<uint4*>(G + ((long)(py0 + oy) * p.W + px0 + ox) * p.ldg + n0 + dc * VEC) = ov;
    }
}

}  // namespace

int g_head_variant = 0;   // 0: row bands (512 tokens), 2: 8x8 patches, 4: 14x14 patches (both measured slower than unfused)
extern "C" int emip_debug_set_head(int variant) { g_head_variant = variant; return EMIP_OK; }

// G[b][y][x][n] = GELU(bdw[n] + sum_taps Wt[tap][n] * H[b][y+dy][x+dx][n]),  H = X W1^T + b1 (zero outside the image).
// X [B][H][W][ldx] (C channels), W1 [Ch][C], G [B][H][W][ldg] (Ch channels).  C, Ch multiples of the 128-byte slab
// (64 bf16 / 32 f32 channels); W <= 170 so that three image rows fit the 512-token band tile.
extern "C" int emip_mlp_head(const void* X, long ldx, const void* W1, const float* b1, const float* Wt, const float* bdw,
                             void* G, long ldg, int B, int H, int W, int C, int Ch, int dtype, void* stream) {
    EMIP_REQUIRE(X && W1 && b1 && Wt && bdw && G && B > 0 && B < 65536 && H > 0 && W > 0);
    EMIP_REQUIRE(dtype == EMIP_F32 || dtype == EMIP_BF16);
    const int nc = dtype == EMIP_F32 ? 32 : 64, vec = dtype == EMIP_F32 ? 4 : 8;
    EMIP_REQUIRE(C > 0 && Ch > 0 && C % nc == 0 && Ch % nc == 0 && ldx % vec == 0 && ldg % vec == 0 && ldx >= C && ldg >= Ch);
    EMIP_REQUIRE(aligned16(X) && aligned16(W1) && aligned16(G) && aligned16(b1) && aligned16(Wt) && aligned16(bdw));
    EMIP_REQUIRE(3 * W <= HEAD_MAXM);
    HeadArgs a{};
    a.X = X; a.W1 = W1; a.b1 = b1; a.Wt = Wt; a.bdw = bdw; a.G = G;
    a.ldx = ldx; a.ldg = ldg; a.B = B; a.H = H; a.W = W; a.C = C; a.Ch = Ch;
    a.R = ((long)H * W <= HEAD_MAXM) ? H : HEAD_MAXM / W - 2;     // whole image in one band when it fits
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (g_head_variant != 0) {
        // 2-D patches: RT = 2 -> 8x8 outputs (10x10 = 100 tokens), RT = 4 -> 14x14 outputs (16x16 = 256 tokens)
        const int rt = g_head_variant == 4 ? 4 : 2;
        const int PY = rt == 4 ? 14 : 8, PX = PY;
        const int tx = (W + PX - 1) / PX, ty = (H + PY - 1) / PY;
        dim3 grid2((unsigned)(Ch / nc), (unsigned)(tx * ty), (unsigned)B);
        const size_t lds = (size_t)64 * rt * 128 + 10 * nc * sizeof(float);
        if (dtype == EMIP_F32) {
            if (rt == 4) hipLaunchKernelGGL((mlp_head_patch_kernel<float, 4>), grid2, dim3(256), lds, s, a, PY, PX, tx);
            else hipLaunchKernelGGL((mlp_head_patch_kernel<float, 2>), grid2, dim3(256), lds, s, a, PY, PX, tx);
        } else {
            if (rt == 4) hipLaunchKernelGGL((mlp_head_patch_kernel<bf16_t, 4>), grid2, dim3(256), lds, s, a, PY, PX, tx);
            else hipLaunchKernelGGL((mlp_head_patch_kernel<bf16_t, 2>), grid2, dim3(256), lds, s, a, PY, PX, tx);
        }
        return emip_launch_status();
    }
    dim3 grid((unsigned)(Ch / nc), (unsigned)((H + a.R - 1) / a.R), (unsigned)B);
    if (dtype == EMIP_F32)
        hipLaunchKernelGGL(mlp_head_kernel<float>, grid, dim3(256), HEAD_MAXM * 128, s, a);
    else
        hipLaunchKernelGGL(mlp_head_kernel<bf16_t>, grid, dim3(256), HEAD_MAXM * 128, s, a);
    return emip_launch_status();
}
