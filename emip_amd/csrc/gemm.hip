// MFMA GEMM / implicit-GEMM convolution for gfx950.
//
//   C[m, n] = act( sum_k A[m, k] * W[n, k] + bias[n] ) + R[m, n]
//
// A is either a dense row-major matrix (optionally the K-concatenation of two
// matrices) or the implicit im2col view of an NHWC tensor (conv mode, K index =
// (ky, kx, ci) with ci fastest, weights packed [Cout][KH][KW][Cin]).
//
// Tiling: 256 threads = 4 waves (2 x 2), block tile BM x BN, K tile = 128 bytes
// per row (64 bf16 / 32 f32), register-staged global->LDS with two LDS buffers
// and one barrier per K tile.  LDS rows are 128 B with the 16-byte chunk index
// XOR-swizzled by ((row >> 1) & 7) so that the ds_read_b128 fragment reads of a
// 16-lane group hit 16 distinct 16-B slots.  The MFMA "A" operand is the weight
// tile and the "B" operand the activation tile, so each lane ends up with 4
// consecutive output channels of one output row: bias / GELU / residual and the
// store are all 4-wide vectors.
//
// bf16: v_mfma_f32_16x16x32_bf16.  f32 (parity mode): v_mfma_f32_16x16x4_f32 fed
// from the same 16-byte fragments (4 k-steps per fragment; the k order inside a
// fragment is permuted identically for both operands).
#include "common.h"

namespace {

struct GemmArgs {
    const void* A;
    const void* A2;
    const void* W;
    void* C;
    const float* bias;
    const void* R;
    int M, N, K, K1;
    long lda, lda2, ldw, ldc, ldr;
    int act;
    // conv view of A
    int H, Wd, Cin, Ho, Wo, KH, KW, stride, pad;
    // batching over blockIdx.z
    long bsA, bsW, bsC, bsR;
    int tiles_m, tiles_n;
};

template <typename T>
struct Mma;

template <>
struct Mma<bf16_t> {
    static __device__ __forceinline__ f32x4 run(const uint4& a, const uint4& b, f32x4 acc) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                       acc, 0, 0, 0);
    }
};
template <>
struct Mma<float> {
    static __device__ __forceinline__ f32x4 run(const uint4& a, const uint4& b, f32x4 acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
        return acc;
    }
};

template <typename T, int BM, int BN, bool CONV>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs p) {
    constexpr int VEC = 16 / sizeof(T);   // elements per 16-B chunk
    constexpr int BK = 128 / sizeof(T);   // K elements per tile
    constexpr int TM = BM / 32, TN = BN / 32;  // 16x16 sub-tiles per wave
    constexpr int CA = BM / 32, CW = BN / 32;  // 16-B chunks staged per thread
    constexpr int TILE_BYTES = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;

    const int ntile = p.tiles_m * p.tiles_n;
    const int swz = xcd_remap(blockIdx.x, ntile);
    const int tile_m = swz / p.tiles_n, tile_n = swz - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const long z = blockIdx.z;

    const T* __restrict__ A = reinterpret_cast<const T*>(p.A) + z * p.bsA;
    const T* __restrict__ A2 = p.A2 ? reinterpret_cast<const T*>(p.A2) + z * p.bsA : nullptr;
    const T* __restrict__ Wp = reinterpret_cast<const T*>(p.W) + z * p.bsW;

    // ---- staging bookkeeping: thread owns chunk column sc of rows srow + 32*i
    const int sc = tid & 7;
    const int srow = tid >> 3;
    const int swz_c = (sc ^ ((srow >> 1) & 7)) * 16;  // (row>>1)&7 is the same for row + 32*i

    long a_off[CA];   // GEMM: row offset (elements); CONV: pixel base of the image (b*H*W)
    int a_iy[CA], a_ix[CA];
    bool a_ok[CA];
#pragma unroll
    for (int i = 0; i < CA; ++i) {
        const int m = m0 + srow + 32 * i;
        a_ok[i] = m < p.M;
        if (CONV) {
            const int hw = p.Ho * p.Wo;
            const int mm = a_ok[i] ? m : 0;
            const int b = mm / hw;
            const int r = mm - b * hw;
            const int oy = r / p.Wo, ox = r - oy * p.Wo;
            a_off[i] = (long)b * p.H * p.Wd;
            a_iy[i] = oy * p.stride - p.pad;
            a_ix[i] = ox * p.stride - p.pad;
        } else {
            a_off[i] = (long)m;
            a_iy[i] = a_ix[i] = 0;
        }
    }
    bool w_ok[CW];
    long w_off[CW];
#pragma unroll
    for (int i = 0; i < CW; ++i) {
        const int n = n0 + srow + 32 * i;
        w_ok[i] = n < p.N;
        w_off[i] = (long)n * p.ldw;
    }

    uint4 ra[CA], rw[CW];
    auto load_tile = [&](int k0) {
        const int kk = k0 + sc * VEC;
        const bool kok = kk < p.K;
        if (CONV) {
            const int tap = kk / p.Cin;
            const int ci = kk - tap * p.Cin;
            const int ky = tap / p.KW, kx = tap - ky * p.KW;
#pragma unroll
            for (int i = 0; i < CA; ++i) {
                const int iy = a_iy[i] + ky, ix = a_ix[i] + kx;
                const bool ok = kok && a_ok[i] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.Wd;
                ra[i] = make_uint4(0, 0, 0, 0);
                if (ok) ra[i] = *reinterpret_cast<const uint4*>(A + (a_off[i] + (long)iy * p.Wd + ix) * p.lda + ci);
            }
        } else {
            const bool second = kk >= p.K1;
            const T* base = second ? A2 : A;
            const long ld = second ? p.lda2 : p.lda;
            const int kc = second ? kk - p.K1 : kk;
#pragma unroll
            for (int i = 0; i < CA; ++i) {
                ra[i] = make_uint4(0, 0, 0, 0);
                if (kok && a_ok[i]) ra[i] = *reinterpret_cast<const uint4*>(base + a_off[i] * ld + kc);
            }
        }
#pragma unroll
        for (int i = 0; i < CW; ++i) {
            rw[i] = make_uint4(0, 0, 0, 0);
            if (kok && w_ok[i]) rw[i] = *reinterpret_cast<const uint4*>(Wp + w_off[i] + kk);
        }
    };
    auto store_tile = [&](int buf) {
        char* ta = smem + buf * TILE_BYTES;
        char* tw = ta + BM * 128;
#pragma unroll
        for (int i = 0; i < CA; ++i) *reinterpret_cast<uint4*>(ta + (srow + 32 * i) * 128 + swz_c) = ra[i];
#pragma unroll
        for (int i = 0; i < CW; ++i) *reinterpret_cast<uint4*>(tw + (srow + 32 * i) * 128 + swz_c) = rw[i];
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    const int fsw = (fr >> 1) & 7;  // sub-tile bases are multiples of 16 -> (row>>1)&7 == (fr>>1)&7

    const int nk = (p.K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile((kt + 1) * BK);
        const char* ta = smem + cur * TILE_BYTES;
        const char* tw = ta + BM * 128;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int coff = ((4 * g + fq) ^ fsw) * 16;
            uint4 fa[TM], fw[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = *reinterpret_cast<const uint4*>(ta + (wm * (BM / 2) + 16 * i + fr) * 128 + coff);
#pragma unroll
            for (int i = 0; i < TN; ++i)
                fw[i] = *reinterpret_cast<const uint4*>(tw + (wn * (BN / 2) + 16 * i + fr) * 128 + coff);
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b) acc[a][b] = Mma<T>::run(fw[a], fa[b], acc[a][b]);
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds rows m = .. + fr, channels n = .. + 4*fq + {0..3}
    T* __restrict__ C = reinterpret_cast<T*>(p.C) + z * p.bsC;
    const T* __restrict__ R = p.R ? reinterpret_cast<const T*>(p.R) + z * p.bsR : nullptr;
    const bool vec_ok = ((p.ldc & 3) == 0) && (R == nullptr || (p.ldr & 3) == 0);
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        const int m = m0 + wm * (BM / 2) + 16 * b + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            const int n = n0 + wn * (BN / 2) + 16 * a + 4 * fq;
            if (n >= p.N) continue;
            float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
            const bool full = (n + 3 < p.N) && vec_ok;
            if (p.bias) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (n + j < p.N) v[j] += p.bias[n + j];
            }
            if (p.act == EMIP_ACT_RELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
            } else if (p.act == EMIP_ACT_GELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
            }
            if (full) {
                if (R) {
                    float r[4];
                    Vec4<T>::load(R + (long)m * p.ldr + n, r);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += r[j];
                }
                Vec4<T>::store(C + (long)m * p.ldc + n, v);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (n + j < p.N) {
                        float o = v[j];
                        if (R) o += to_f32<T>(R[(long)m * p.ldr + n + j]);
                        C[(long)m * p.ldc + n + j] = from_f32<T>(o);
                    }
                }
            }
        }
    }
}

template <typename T, int BM, int BN, bool CONV>
int launch(GemmArgs& a, int batch, hipStream_t s) {
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.N + BN - 1) / BN;
    const size_t lds = 2 * (BM + BN) * 128;
    dim3 grid(a.tiles_m * a.tiles_n, 1, batch);
    hipLaunchKernelGGL((gemm_kernel<T, BM, BN, CONV>), grid, dim3(256), lds, s, a);
    return emip_launch_status();
}

template <typename T, bool CONV>
int dispatch(GemmArgs& a, int batch, hipStream_t s) {
    // tile choice: least padded work, then the larger tile; small problems take the
    // smaller tile so that the grid still covers the 256 CUs.
    auto waste = [&](int bm, int bn) {
        const long tm = (a.M + bm - 1) / bm, tn = (a.N + bn - 1) / bn;
        return tm * bm * tn * bn;
    };
    auto blocks = [&](int bm, int bn) { return (long)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn) * batch; };
    int bn = (waste(128, 64) < waste(128, 128)) ? 64 : 128;
    int bm = 128;
    if (blocks(bm, bn) < 512 || waste(64, bn) * 10 < waste(128, bn) * 9) bm = 64;
    if (bn == 128 && blocks(bm, bn) < 256) bn = 64;
    if (bm == 128 && bn == 128) return launch<T, 128, 128, CONV>(a, batch, s);
    if (bm == 128 && bn == 64) return launch<T, 128, 64, CONV>(a, batch, s);
    if (bm == 64 && bn == 128) return launch<T, 64, 128, CONV>(a, batch, s);
    return launch<T, 64, 64, CONV>(a, batch, s);
}

}  // namespace

extern "C" int emip_gemm(const void* A, const void* A2, const void* W, void* C, const float* bias, const void* R,
                         int M, int N, int K, int K1, long lda, long lda2, long ldw, long ldc, long ldr, int act,
                         int batch, long bsA, long bsW, long bsC, long bsR, int dtype, void* stream) {
    EMIP_REQUIRE(A && W && C && M > 0 && N > 0 && K > 0 && batch > 0);
    EMIP_REQUIRE(dtype == EMIP_F32 || dtype == EMIP_BF16);
    const int vec = dtype == EMIP_F32 ? 4 : 8;
    const int bk = dtype == EMIP_F32 ? 32 : 64;
    EMIP_REQUIRE(K % vec == 0 && lda % vec == 0 && ldw % vec == 0 && bsA % vec == 0 && bsW % vec == 0);
    EMIP_REQUIRE(lda >= (A2 ? K1 : K) && ldw >= K && ldc >= N);
    EMIP_REQUIRE(aligned16(A) && aligned16(W));
    if (A2) {
        EMIP_REQUIRE(K1 > 0 && K1 < K && K1 % bk == 0 && lda2 % vec == 0 && lda2 >= K - K1 && aligned16(A2));
    } else {
        K1 = K;
    }
    if (R) EMIP_REQUIRE(ldr >= N);
    EMIP_REQUIRE(act >= EMIP_ACT_NONE && act <= EMIP_ACT_GELU);
    GemmArgs a{};
    a.A = A; a.A2 = A2; a.W = W; a.C = C; a.bias = bias; a.R = R;
    a.M = M; a.N = N; a.K = K; a.K1 = K1;
    a.lda = lda; a.lda2 = lda2; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr; a.act = act;
    a.bsA = bsA; a.bsW = bsW; a.bsC = bsC; a.bsR = bsR;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return dtype == EMIP_F32 ? dispatch<float, false>(a, batch, s) : dispatch<bf16_t, false>(a, batch, s);
}

extern "C" int emip_conv2d(const void* X, const void* W, void* Y, const float* bias, const void* R, int B, int H,
                           int Wd, int Cin, long ldx, int Cout, int KH, int KW, int stride, int pad, long ldy,
                           long ldr, int act, int dtype, void* stream) {
    EMIP_REQUIRE(X && W && Y && B > 0 && H > 0 && Wd > 0 && Cin > 0 && Cout > 0);
    EMIP_REQUIRE(dtype == EMIP_F32 || dtype == EMIP_BF16);
    EMIP_REQUIRE(KH > 0 && KW > 0 && stride > 0 && pad >= 0);
    const int vec = dtype == EMIP_F32 ? 4 : 8;
    EMIP_REQUIRE(Cin % vec == 0 && ldx % vec == 0 && ldx >= Cin && ldy >= Cout);
    EMIP_REQUIRE(aligned16(X) && aligned16(W));
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (Wd + 2 * pad - KW) / stride + 1;
    EMIP_REQUIRE(Ho > 0 && Wo > 0);
    EMIP_REQUIRE((long)B * Ho * Wo < 2147483647L && (long)KH * KW * Cin < 2147483647L);
    if (R) EMIP_REQUIRE(ldr >= Cout);
    EMIP_REQUIRE(act >= EMIP_ACT_NONE && act <= EMIP_ACT_GELU);
    GemmArgs a{};
    a.A = X; a.W = W; a.C = Y; a.bias = bias; a.R = R;
    a.M = B * Ho * Wo; a.N = Cout; a.K = KH * KW * Cin; a.K1 = a.K;
    a.lda = ldx; a.ldw = a.K; a.ldc = ldy; a.ldr = ldr; a.act = act;
    a.H = H; a.Wd = Wd; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return dtype == EMIP_F32 ? dispatch<float, true>(a, 1, s) : dispatch<bf16_t, true>(a, 1, s);
}
