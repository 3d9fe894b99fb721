// Weight-gradient contraction for gfx950:   C[n][k] (+)= sum_m A[m][n] * B[m][k]      (f32 output)
//
// A = dY [M][N] and B = X [M][K] are both row-major with the CONTRACTION index m as the row (the layouts the
// forward pass leaves in HBM), i.e. a "TN" GEMM: nn.Linear / 1x1-conv wgrad dW = dY^T X
// (backward of /root/reference/lib/pvt_v2.py:45-54,103,110,126 and the other Linear call sites of emip_gemm).
// M is huge (up to 247 808 tokens) and N x K small, so M is split over blockIdx.y and partial tiles are
// combined with f32 atomics into a zero-initialised C.
//
// Tiles are staged row-major ([m][n], [m][k]: coalesced 16-B global loads) and the MFMA operands, which need 8
// consecutive m for a fixed n / k, are produced by the LDS transpose read ds_read_b64_tr_b16 (bf16) or plain
// ds_read_b32 (f32).  16-B chunks are XOR-swizzled by a function of the row so that the 8 rows a half-wave's
// transposed read touches land in distinct bank groups.
#include "common.h"

extern "C" int emip_gemm_tn_bias(const void*, const void*, float*, float*, long, int, int, long, long, long, int, long, long,
                                 long, int, void*);

namespace {

struct TnArgs {
    const void* A;
    const void* B;
    float* C;
    long M;
    int N, K;
    long lda, ldb, ldc;
    long m_per_split;
    int tiles_n, tiles_k;
    // CONV: B[m][k] is the im2col view of an NHWC tensor X (m = output pixel, k = (ky, kx, ci), ci fastest)
    int H, Wd, Cin, Ho, Wo, KW, stride, pad;
    // batching over blockIdx.z (element strides); heads > 1: z = zb * heads + zh, operand at zb * bs + zh * hs
    long bsA, bsB, bsC;
    long hsA, hsB, hsC;
    int heads;
    int prezeroed;
    float* db;     // optional: column sums of A (the bias gradient sum_m dY[m][n]), accumulated with atomics (zero beforehand)
};

__device__ __forceinline__ int tn_f(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

template <typename T, bool CONV>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const TnArgs p) {
    constexpr int ES = sizeof(T);
    constexpr bool BF = ES == 2;
    constexpr int BMS = BF ? 64 : 32;          // rows (m) per stage
    constexpr int RB = 128 * ES;               // bytes per tile row (128 columns)
    constexpr int CPR = RB / 16;               // 16-B chunks per row
    constexpr int NS = BMS * CPR / 256;        // staged chunks per thread per operand
    constexpr int TILE = BMS * RB;
    __shared__ __attribute__((aligned(16))) char smem[2 * TILE];
    char* ta = smem;
    char* tb = smem + TILE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 1, wk = wave >> 1;
    const int tile_n = blockIdx.x / p.tiles_k, tile_k = blockIdx.x - tile_n * p.tiles_k;
    const int n0 = tile_n * 128, k0 = tile_k * 128;
    const long m_lo = (long)blockIdx.y * p.m_per_split;
    const long m_hi = min(p.M, m_lo + p.m_per_split);
    const long zz = blockIdx.z;
    const long zb = p.heads > 1 ? zz / p.heads : zz, zh = p.heads > 1 ? zz - zb * p.heads : 0;
    const T* __restrict__ A = reinterpret_cast<const T*>(p.A) + zb * p.bsA + zh * p.hsA;
    const T* __restrict__ Bp = reinterpret_cast<const T*>(p.B) + zb * p.bsB + zh * p.hsB;
    float* __restrict__ Cp = p.C + zb * p.bsC + zh * p.hsC;

    uint4 ra[NS], rb[NS];
    // bias gradient: a thread always stages the same 16-byte column chunk of A (256 % CPR == 0), so it can keep the column
    // sums of everything it stages; only the k-tile-0 workgroups do it, so that every element of A is counted once
    constexpr int VEC = 16 / ES;
    const bool do_db = p.db != nullptr && tile_k == 0;
    float cs[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) cs[j] = 0.f;
    auto load_stage = [&](long m0) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int id = tid + 256 * i;
            const int r = id / CPR, c = id - r * CPR;
            const long m = m0 + r;
            const long mc = min(m, p.M - 1);
            const int na = n0 + c * (16 / ES), kb = k0 + c * (16 / ES);
            const int nac = min(na, max(p.N - 16 / ES, 0)), kbc = min(kb, max(p.K - 16 / ES, 0));
            ra[i] = mask4(*reinterpret_cast<const uint4*>(A + mc * p.lda + nac), m < m_hi && na < p.N);
            if (CONV) {
                const int hw = p.Ho * p.Wo;
                const int bimg = (int)(mc / hw);
                const int rem = (int)(mc - (long)bimg * hw);
                const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                const int tap = kbc / p.Cin, ci = kbc - tap * p.Cin;
                const int ky = tap / p.KW, kx = tap - ky * p.KW;
                const int iy = oy * p.stride - p.pad + ky, ix = ox * p.stride - p.pad + kx;
                const bool ok = m < m_hi && kb < p.K && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.Wd;
                const int iyc = min(max(iy, 0), p.H - 1), ixc = min(max(ix, 0), p.Wd - 1);
                rb[i] = mask4(*reinterpret_cast<const uint4*>(Bp + (((long)bimg * p.H + iyc) * p.Wd + ixc) * p.ldb + ci), ok);
            } else {
                rb[i] = mask4(*reinterpret_cast<const uint4*>(Bp + mc * p.ldb + kbc), m < m_hi && kb < p.K);
            }
        }
    };
    auto store_stage = [&]() {
        if (do_db) {
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const T* e = reinterpret_cast<const T*>(&ra[i]);
#pragma unroll
                for (int j = 0; j < VEC; ++j) cs[j] += to_f32<T>(e[j]);
            }
        }
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int id = tid + 256 * i;
            const int r = id / CPR, c = id - r * CPR;
            const int cs = BF ? (c ^ (tn_f(r) << 1)) : c;
            *reinterpret_cast<uint4*>(ta + r * RB + cs * 16) = ra[i];
            *reinterpret_cast<uint4*>(tb + r * RB + cs * 16) = rb[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int q = lane >> 4, i16 = lane & 15;
    if (m_lo < m_hi) {
        load_stage(m_lo);
        store_stage();
        __syncthreads();
        for (long m0 = m_lo; m0 < m_hi; m0 += BMS) {
            if (m0 + BMS < m_hi) load_stage(m0 + BMS);
            if (BF) {
                typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
                const int r = i16 >> 2, pp = i16 & 3;
#pragma unroll
                for (int ks = 0; ks < BMS / 32; ++ks) {
                    const int row0 = 32 * ks + 8 * q + r, row1 = row0 + 4;
                    bf16x8 fa[4], fb[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int cola = wn * 64 + 16 * t + 4 * pp, colb = wk * 64 + 16 * t + 4 * pp;
                        const int ca = cola >> 3, ha = (cola >> 2) & 1, cb = colb >> 3, hb = (colb >> 2) & 1;
                        const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (lds_s16x4*)(ta + row0 * RB + ((ca ^ (tn_f(row0) << 1)) * 16) + 8 * ha));
                        const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (lds_s16x4*)(ta + row1 * RB + ((ca ^ (tn_f(row1) << 1)) * 16) + 8 * ha));
                        const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (lds_s16x4*)(tb + row0 * RB + ((cb ^ (tn_f(row0) << 1)) * 16) + 8 * hb));
                        const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (lds_s16x4*)(tb + row1 * RB + ((cb ^ (tn_f(row1) << 1)) * 16) + 8 * hb));
                        const bf16x4 xa0 = __builtin_bit_cast(bf16x4, a0), xa1 = __builtin_bit_cast(bf16x4, a1);
                        const bf16x4 xb0 = __builtin_bit_cast(bf16x4, b0), xb1 = __builtin_bit_cast(bf16x4, b1);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            fa[t][j] = xa0[j]; fa[t][4 + j] = xa1[j];
                            fb[t][j] = xb0[j]; fb[t][4 + j] = xb1[j];
                        }
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
                }
            } else {
#pragma unroll 2
                for (int ks = 0; ks < BMS / 4; ++ks) {
                    const int row = 4 * ks + q;
                    float fa[4], fb[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        fa[t] = *reinterpret_cast<const float*>(ta + row * RB + (wn * 64 + 16 * t + i16) * 4);
                        fb[t] = *reinterpret_cast<const float*>(tb + row * RB + (wk * 64 + 16 * t + i16) * 4);
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
                }
            }
            __syncthreads();
            if (m0 + BMS < m_hi) store_stage();
            __syncthreads();
        }
    }

    if (do_db) {      // workgroup-uniform
        __syncthreads();                                   // the staging tiles are free now
        float* red = reinterpret_cast<float*>(smem);        // [256 / CPR][CPR * VEC]
        const int c = tid % CPR, g = tid / CPR;
#pragma unroll
        for (int j = 0; j < VEC; ++j) red[g * (CPR * VEC) + c * VEC + j] = cs[j];
        __syncthreads();
        if (tid < CPR * VEC) {
            float t = 0.f;
            for (int gg = 0; gg < 256 / CPR; ++gg) t += red[gg * (CPR * VEC) + tid];
            const int n = n0 + tid;
            if (n < p.N) atomicAdd(p.db + (long)blockIdx.z * p.N + n, t);
        }
    }
    // lane holds C[n = .. + 4q + reg][k = .. + i16]
    const bool atomic = gridDim.y > 1 || p.prezeroed;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int k = k0 + wk * 64 + 16 * b + i16;
            if (k >= p.K) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + 16 * a + 4 * q + j;
                if (n >= p.N) continue;
                float* dst = Cp + (long)n * p.ldc + k;
                if (atomic) atomicAdd(dst, acc[a][b][j]);
                else *dst = acc[a][b][j];
            }
        }
}

}  // namespace

int g_tn_target = 0;   // 0 = heuristic; tools/gemm_tn_bench.py overrides it
int g_tn_ring = 1;     // eligible bf16 launches go to the LDS-DMA ring body (gemm_tn8.hip)
#ifdef EMIP_TUNING
extern "C" int emip_debug_set_tn(int target) {        // target < 0: the register-staged body for every launch
    g_tn_ring = target >= 0;
    g_tn_target = target > 0 ? target : 0;
    return EMIP_OK;
}
#endif

namespace {
template <bool CONV>
int launch_tn(TnArgs& a, int batch, int dtype, hipStream_t s) {
    a.tiles_n = (a.N + 127) / 128;
    a.tiles_k = (a.K + 127) / 128;
    const int bms = dtype == EMIP_F32 ? 32 : 64;
    const long tiles = (long)a.tiles_n * a.tiles_k * batch;
    // workgroups to aim at: every split adds a 128x128 f32 tile of atomics, so few output tiles take ~512 workgroups
    // and many tiles ~256 (measured on the PVTv2-b5 shapes, tools/gemm_tn_bench.py)
    const long target = g_tn_target > 0 ? g_tn_target : (tiles <= 4 ? 512 : 256);
    long splits = (target + tiles - 1) / tiles;
    const long max_splits = (a.M + 8 * bms - 1) / (8 * bms);        // but >= 8 stages of work per workgroup (123 904 x 128 x 128:
                                                                    // 512 splits of 4 stages 52 us, 242 splits of 8 stages 39 us)
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    a.m_per_split = ((a.M + splits - 1) / splits + bms - 1) / bms * bms;
    splits = (a.M + a.m_per_split - 1) / a.m_per_split;
    const size_t c_floats = (size_t)a.N * a.K * batch, db_floats = a.db ? (size_t)a.N * batch : 0;
    bool db_cleared = a.db == nullptr || a.prezeroed;
    if (splits > 1 && !a.prezeroed) {   // partial tiles are combined with atomics: clear the output first (dense rows only)
        if (a.ldc != a.K || (batch > 1 && a.bsC != (long)a.N * a.K)) return EMIP_E_INVALID;
        // a bias gradient placed right behind the weight gradient is cleared by the same launch
        const bool joint = !db_cleared && a.db == a.C + c_floats;
        if (emip_zero_async(a.C, sizeof(float) * (c_floats + (joint ? db_floats : 0)), s) != EMIP_OK) return EMIP_E_LAUNCH;
        db_cleared = db_cleared || joint;
    }
    if (!db_cleared && emip_zero_async(a.db, sizeof(float) * db_floats, s) != EMIP_OK) return EMIP_E_LAUNCH;
    dim3 grid((unsigned)(a.tiles_n * a.tiles_k), (unsigned)splits, (unsigned)batch);
    if (dtype == EMIP_F32) hipLaunchKernelGGL((gemm_tn_kernel<float, CONV>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((gemm_tn_kernel<bf16_t, CONV>), grid, dim3(256), 0, s, a);
    return emip_launch_status();
}
}  // namespace

extern "C" int emip_gemm_tn(const void* A, const void* B, float* C, long M, int N, int K, long lda, long ldb, long ldc,
                            int batch, long bsA, long bsB, long bsC, int dtype, void* stream) {
    return emip_gemm_tn_bias(A, B, C, nullptr, M, N, K, lda, ldb, ldc, batch, bsA, bsB, bsC, dtype, stream);
}

// emip_gemm_tn that also produces db[n] = sum_m A[m][n] (the bias gradient of the same Linear; db f32 [batch][N], cleared by
// this call -- in the same zero launch as C when it sits right behind it): the dY tile is already being staged, so the
// separate column-sum pass over dY disappears.
extern "C" int emip_gemm_tn_bias(const void* A, const void* B, float* C, float* db, long M, int N, int K, long lda,
                                 long ldb, long ldc, int batch, long bsA, long bsB, long bsC, int dtype, void* stream) {
    EMIP_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && batch > 0 && batch < 65536);
    EMIP_REQUIRE(dtype == EMIP_F32 || dtype == EMIP_BF16);
    const int vec = dtype == EMIP_F32 ? 4 : 8;
    EMIP_REQUIRE(N % vec == 0 && K % vec == 0 && lda % vec == 0 && ldb % vec == 0 && lda >= N && ldb >= K && ldc >= K);
    EMIP_REQUIRE(bsA % vec == 0 && bsB % vec == 0 && aligned16(A) && aligned16(B));
    if (g_tn_ring && dtype == EMIP_BF16 && batch == 1 && emip_gemm_tn8_eligible(M, N, K, lda, ldb))
        return emip_gemm_tn8(A, B, C, db, M, N, K, lda, ldb, ldc, 0, stream);
    TnArgs a{};
    a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.bsA = bsA; a.bsB = bsB; a.bsC = bsC;
    a.heads = 1;
    a.db = db;
    return launch_tn<false>(a, batch, dtype, reinterpret_cast<hipStream_t>(stream));
}

// The same contraction ADDED into C (and db) that the caller has already cleared: the training step clears one arena holding
// every weight gradient with a single launch instead of one zero launch per weight (emip_amd/ops.py GradArena).
extern "C" int emip_gemm_tn_into(const void* A, const void* B, float* C, float* db, long M, int N, int K, long lda,
                                 long ldb, long ldc, int dtype, void* stream) {
    EMIP_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0);
    EMIP_REQUIRE(dtype == EMIP_F32 || dtype == EMIP_BF16);
    const int vec = dtype == EMIP_F32 ? 4 : 8;
    EMIP_REQUIRE(N % vec == 0 && K % vec == 0 && lda % vec == 0 && ldb % vec == 0 && lda >= N && ldb >= K && ldc >= K);
    EMIP_REQUIRE(aligned16(A) && aligned16(B));
    if (g_tn_ring && dtype == EMIP_BF16 && emip_gemm_tn8_eligible(M, N, K, lda, ldb))
        return emip_gemm_tn8(A, B, C, db, M, N, K, lda, ldb, ldc, 1, stream);
    TnArgs a{};
    a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.heads = 1;
    a.db = db;
    a.prezeroed = 1;
    return launch_tn<false>(a, 1, dtype, reinterpret_cast<hipStream_t>(stream));
}

// Two-level batch (batch = B * heads, operand of (b, h) at b * bs + h * hs) and an output that may be a column slice of a
// wider PRE-ZEROED f32 buffer (ldc > K): results are ADDED with atomics.  dK / dV of all attention heads in one launch,
// written straight into the [B][keys][2C] gradient of the kv projection.
extern "C" int emip_gemm_tn_heads(const void* A, const void* B, float* C, long M, int N, int K, long lda, long ldb,
                                  long ldc, int batch, int heads, long bsA, long hsA, long bsB, long hsB, long bsC,
                                  long hsC, int dtype, void* stream) {
    EMIP_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && batch > 0 && batch < 65536 && heads >= 1 && batch % heads == 0);
    EMIP_REQUIRE(dtype == EMIP_F32 || dtype == EMIP_BF16);
    const int vec = dtype == EMIP_F32 ? 4 : 8;
    EMIP_REQUIRE(N % vec == 0 && K % vec == 0 && lda % vec == 0 && ldb % vec == 0 && lda >= N && ldb >= K && ldc >= K);
    EMIP_REQUIRE(bsA % vec == 0 && bsB % vec == 0 && hsA % vec == 0 && hsB % vec == 0 && aligned16(A) && aligned16(B));
    TnArgs a{};
    a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.bsA = bsA; a.bsB = bsB; a.bsC = bsC; a.hsA = hsA; a.hsB = hsB; a.hsC = hsC; a.heads = heads; a.prezeroed = 1;
    return launch_tn<false>(a, batch, dtype, reinterpret_cast<hipStream_t>(stream));
}

// dW[co][ky][kx][ci] = sum_{b,oy,ox} dY[b,oy,ox,co] * X[b, oy*s-p+ky, ox*s-p+kx, ci]   (f32, packed like the forward weights)
namespace {
int conv_wgrad(const void* dY, const void* X, float* dW, int B, int H, int Wd, int Cin, long ldx, int Cout, long lddy, int KH,
               int KW, int stride, int pad, int dtype, void* stream, int prezeroed) {
    EMIP_REQUIRE(dY && X && dW && B > 0 && H > 0 && Wd > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 &&
                 pad >= 0);
    EMIP_REQUIRE(dtype == EMIP_F32 || dtype == EMIP_BF16);
    const int vec = dtype == EMIP_F32 ? 4 : 8;
    EMIP_REQUIRE(Cin % vec == 0 && Cout % vec == 0 && ldx % vec == 0 && lddy % vec == 0 && ldx >= Cin && lddy >= Cout);
    EMIP_REQUIRE(aligned16(dY) && aligned16(X));
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (Wd + 2 * pad - KW) / stride + 1;
    EMIP_REQUIRE(Ho > 0 && Wo > 0 && (long)B * Ho * Wo < 2147483647L && (long)KH * KW * Cin < 2147483647L);
    if (g_tn_ring && dtype == EMIP_BF16 && emip_conv_wgrad8_eligible(B, H, Wd, Cin, ldx, Cout, lddy, KH, KW, stride, pad))
        return emip_conv_wgrad8(dY, X, dW, B, H, Wd, Cin, ldx, Cout, lddy, KH, KW, stride, pad, prezeroed, stream);
    TnArgs a{};
    a.A = dY; a.B = X; a.C = dW; a.M = (long)B * Ho * Wo; a.N = Cout; a.K = KH * KW * Cin;
    a.lda = lddy; a.ldb = ldx; a.ldc = a.K;
    a.H = H; a.Wd = Wd; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.KW = KW; a.stride = stride; a.pad = pad;
    a.heads = 1;
    a.prezeroed = prezeroed;
    return launch_tn<true>(a, 1, dtype, reinterpret_cast<hipStream_t>(stream));
}
}  // namespace

extern "C" int emip_conv2d_wgrad(const void* dY, const void* X, float* dW, int B, int H, int Wd, int Cin, long ldx,
                                 int Cout, long lddy, int KH, int KW, int stride, int pad, int dtype, void* stream) {
    return conv_wgrad(dY, X, dW, B, H, Wd, Cin, ldx, Cout, lddy, KH, KW, stride, pad, dtype, stream, 0);
}
// ADDED into a dW the caller has cleared (see emip_gemm_tn_into)
extern "C" int emip_conv2d_wgrad_into(const void* dY, const void* X, float* dW, int B, int H, int Wd, int Cin, long ldx,
                                      int Cout, long lddy, int KH, int KW, int stride, int pad, int dtype, void* stream) {
    return conv_wgrad(dY, X, dW, B, H, Wd, Cin, ldx, Cout, lddy, KH, KW, stride, pad, dtype, stream, 1);
}
