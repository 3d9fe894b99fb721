// Weight-gradient contraction on 256 x 320 output tiles (bf16 in, f32 out) -- the grouped launch behind the training step's
// deferred Linear AND convolution weight gradients whose output has a 320-friendly side.
//
//   C[n, k] (+)= sum_m A[m, n] * B[m, k]        A = dY [M, lda], B = X [M, ldb] (or the im2col view of an NHWC tensor)
//
// Row T of SURVEY.md section 8 (/root/reference/train.py:52-58 loss.backward() through lib/pvt_v2.py:45-54,101-129 and
// model/EMIP_short/model.py conv_corr).  gemm_tn8.hip walks 128 x 128 tiles with four waves of 64 x 64: every 64-row stage
// moves 32 KB from L2 for 2 MFLOP (64 FLOP/B) and every wave reads 16 KB of LDS for 512 MFMA cycles, so the kernel sits on
// the L2 -> LDS path and on the LDS read port at ~500 TFLOP/s.  PVTv2-b5's third stage (40 of 52 blocks) has C = 320: its
// weight gradients are 320 x 320, 1280 x 320, 320 x 1280, 640 x 320 and (spatial reduction conv) 320 x 1280.  Here:
//   * output tile 256 x 320 (orientation 0: 2 x 4 waves of 128 x 80) or 320 x 256 (orientation 1: 4 x 2 waves of 80 x 128),
//     chosen per problem so that the 320 side is covered exactly: 142 FLOP per L2 byte, 13 fragment reads per 40 MFMAs;
//   * a stage = 32 rows (m) of both operands as 128-column slabs [32][128] bf16 (8 KB, row-major as in HBM), five slabs per
//     stage, moved L2 -> LDS by LDS-DMA with the transposed-read swizzle on the per-lane SOURCE offset (as gemm_tn8.hip);
//     three stages in a ring (120 KB), ONE raw s_barrier per stage, counted s_waitcnt vmcnt;
//   * MFMA 16x16x32 bf16, both fragments read TRANSPOSED out of the row-major slabs (ds_read_b64_tr_b16);
//   * the bias gradient (column sums of A) is one more MFMA against a register of ones, spread over the waves of a row;
//   * CONV problems (B = im2col view: m = output pixel, k = (ky, kx, ci)) track their pixel incrementally, one per lane;
//   * ONE persistent launch walks the work items (tile, m range) of every problem of the flush: m ranges come in multiples
//     of 8 and item i runs on XCD i % 8, so the tiles of one m range share their operand panels in one XCD's L2.
// Partial tiles meet in the PRE-CLEARED C by f32 atomics.
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 tn16_rsrc(const void* ptr, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
    return i32x4{(int)(unsigned)a, (int)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
// 64 lanes x 16 B from descriptor rs at (voff + soff) into LDS at lds_dst + 16 lane (asm: see gemm8.hip)
__device__ __forceinline__ void tn16_dma16(unsigned lds_dst, unsigned voff, i32x4 rs, unsigned soff) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_dst), "v"(voff), "s"(rs), "s"(soff)
        : "memory");
}
template <int N>
__device__ __forceinline__ void tn16_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct Tn16Args {
    const bf16_t* A;
    const bf16_t* B;
    float* C;
    float* db;            // optional: sum_m A[m][n], accumulated with atomics (pre-cleared)
    long M;
    int N, K;
    long lda, ldb, ldc;
    long m_per_split;     // multiple of 32
    int tiles_k, tiles, splits;
    unsigned a_bytes, b_bytes;
    int item0;            // first work item of this problem in the grouped launch (a multiple of 8)
    int orient;           // 0: 256 (n) x 320 (k) tiles, 1: 320 x 256
    // CONV (KW > 0): B[m][k] is the im2col view of an NHWC tensor X (m = output pixel, k = (ky, kx, ci), ci fastest), ldb = pixel stride
    int H, Wd, Cin, Ho, Wo, KW, stride, pad;
};

constexpr unsigned TN16_OOB = 0x80000000u;
constexpr int TN16_SLAB = 32 * 256;            // [32 rows][128 bf16]
constexpr int TN16_NST = 3;
constexpr int TN16_STAGE = 5 * TN16_SLAB;
constexpr int TN16_LDS = TN16_NST * TN16_STAGE;
__device__ __forceinline__ int tn16_f(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

// one work item = one (output tile, m range) of problem p; bid = its index inside the problem
template <int SA, int SB, int WGN, int WGK, int FA, int FB, bool CONV>
__device__ __forceinline__ void tn16_body(const Tn16Args& p, const int bid, char* smem) {
    static_assert(SA + SB == 5 && WGN * WGK == 8 && WGN * FA * 16 <= SA * 128 && WGK * FB * 16 <= SB * 128, "tile plan");
    constexpr int TN = WGN * FA * 16, TK = WGK * FB * 16;
    constexpr int LPT = SA + SB;               // LDS-DMA instructions per wave and stage
    constexpr int ND = (FA + WGK - 1) / WGK;   // bias-gradient fragments of one wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WGN, wk = wave / WGN;
    // split = id % 8 (+ 8 per block of `tiles` ids): all output tiles of one m range run on ONE XCD at the same time
    const int j8 = bid >> 3;
    const int tile = j8 % p.tiles;
    const int split = (bid & 7) + 8 * (j8 / p.tiles);
    const int tile_n = tile / p.tiles_k, tile_k = tile - tile_n * p.tiles_k;
    const int n0 = tile_n * TN, k0 = tile_k * TK;
    const long m_lo = (long)split * p.m_per_split;
    const long m_hi = min(p.M, m_lo + p.m_per_split);
    const int nstage = m_lo < m_hi ? (int)((m_hi - m_lo + 31) >> 5) : 0;
    if (nstage == 0) return;                   // workgroup-uniform: an empty tail range adds nothing

    const i32x4 rsA = tn16_rsrc(p.A, p.a_bytes);
    const i32x4 rsB = tn16_rsrc(p.B, p.b_bytes);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)smem;

    // ---- staging plan: this wave moves rows 4 wave .. + 3 of every slab; lane l sits at row l >> 4, slot l & 15 of that 1-KB
    // piece and fetches source chunk slot ^ 2 f(row) of its row
    const int rloc = 4 * wave + (lane >> 4);
    const int cch = (lane & 15) ^ (tn16_f(rloc) << 1);
    unsigned aoff[SA], boff[SB];
#pragma unroll
    for (int i = 0; i < SA; ++i) {
        const int cl = 128 * i + 8 * cch;
        aoff[i] = (cl < TN && n0 + cl < p.N) ? (unsigned)((long)rloc * p.lda * 2) + (unsigned)(n0 + cl) * 2u : TN16_OOB;
    }
#pragma unroll
    for (int i = 0; i < SB; ++i) {
        const int cl = 128 * i + 8 * cch;
        boff[i] = (cl < TK && k0 + cl < p.K) ? (unsigned)((long)rloc * p.ldb * 2) + (unsigned)(k0 + cl) * 2u : TN16_OOB;
    }
    // CONV: this lane's chunk of slab i is the same (tap, channel) in every stage; its pixel moves on by 32 output pixels per
    // stage and is tracked incrementally (b, oy, ox): no division inside the ring
    int t_ky[SB], t_kx[SB], t_ci[SB], px_b = 0, px_y = 0, px_x = 0;
    if (CONV) {
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            const int cl = 128 * i + 8 * cch;
            const int k = k0 + cl;
            const int tap = k / p.Cin;
            t_ci[i] = (cl < TK && k < p.K) ? k - tap * p.Cin : -1;
            t_ky[i] = tap / p.KW;
            t_kx[i] = tap - t_ky[i] * p.KW;
        }
        const long m = m_lo + rloc;
        const int hw = p.Ho * p.Wo;
        px_b = (int)(m / hw);
        const int rem = (int)(m - (long)px_b * hw);
        px_y = rem / p.Wo;
        px_x = rem - px_y * p.Wo;
    }
    auto issue = [&](int s, int buf) {
        const long m0 = m_lo + 32L * s;
        const unsigned sa = (unsigned)(m0 * p.lda * 2), sb = (unsigned)(m0 * p.ldb * 2);
        const unsigned base = lds0 + buf * TN16_STAGE + wave * 1024;
        const bool ok = m0 + rloc < m_hi;
#pragma unroll
        for (int i = 0; i < SA; ++i) tn16_dma16(base + i * TN16_SLAB, ok ? aoff[i] : TN16_OOB, rsA, sa);
        if (CONV) {
            const int by = px_y * p.stride - p.pad, bx = px_x * p.stride - p.pad;
#pragma unroll
            for (int i = 0; i < SB; ++i) {
                const int iy = by + t_ky[i], ix = bx + t_kx[i];
                const bool in = ok && t_ci[i] >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.Wd;
                const unsigned off = (unsigned)((((long)px_b * p.H + iy) * p.Wd + ix) * p.ldb + t_ci[i]) * 2u;
                tn16_dma16(base + (SA + i) * TN16_SLAB, in ? off : TN16_OOB, rsB, 0u);
            }
            px_x += 32;                                         // stages are issued in order: the next one is 32 pixels on
            while (px_x >= p.Wo) {
                px_x -= p.Wo;
                if (++px_y == p.Ho) { px_y = 0; ++px_b; }
            }
        } else {
#pragma unroll
            for (int i = 0; i < SB; ++i) tn16_dma16(base + (SA + i) * TN16_SLAB, ok ? boff[i] : TN16_OOB, rsB, sb);
        }
    };

    f32x4 acc[FA][FB];
#pragma unroll
    for (int a = 0; a < FA; ++a)
#pragma unroll
        for (int b = 0; b < FB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_db = p.db != nullptr && tile_k == 0;                  // workgroup-uniform
    f32x4 accd[ND];
#pragma unroll
    for (int a = 0; a < ND; ++a) accd[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16_t)1.0f;

    const int q = lane >> 4, i16 = lane & 15;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const int r4 = i16 >> 2, pp = i16 & 3;
    const int row0 = 8 * q + r4, row1 = row0 + 4;
    const int f0 = tn16_f(row0) << 1, f1 = tn16_f(row1) << 1;
    const int lo0 = row0 * 256 + (pp >> 1) * 16 + (pp & 1) * 8, lo1 = row1 * 256 + (pp >> 1) * 16 + (pp & 1) * 8;
    const int acol = wn * FA * 16, bcol = wk * FB * 16;
    auto compute = [&](int buf) {
        const char* ta = smem + buf * TN16_STAGE;
        const char* tb = ta + SA * TN16_SLAB;
        bf16x8 fb[FB];
#pragma unroll
        for (int t = 0; t < FB; ++t) {
            const int col = bcol + 16 * t;                     // wave-uniform; + 4 pp lives in lo0 / lo1
            const char* sl = tb + (col >> 7) * TN16_SLAB;
            const int cb = (col & 127) >> 3;                   // even
            const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sl + lo0 + ((cb ^ f0) << 4)));
            const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sl + lo1 + ((cb ^ f1) << 4)));
            const bf16x4 x0 = __builtin_bit_cast(bf16x4, b0), x1 = __builtin_bit_cast(bf16x4, b1);
#pragma unroll
            for (int j = 0; j < 4; ++j) { fb[t][j] = x0[j]; fb[t][4 + j] = x1[j]; }
        }
        // one A fragment at a time (the compiler keeps a few reads ahead of the MFMAs): 13 fragments never live together
#pragma unroll
        for (int a = 0; a < FA; ++a) {
            const int col = acol + 16 * a;
            const char* sl = ta + (col >> 7) * TN16_SLAB;
            const int cb = (col & 127) >> 3;
            const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sl + lo0 + ((cb ^ f0) << 4)));
            const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sl + lo1 + ((cb ^ f1) << 4)));
            const bf16x4 x0 = __builtin_bit_cast(bf16x4, a0), x1 = __builtin_bit_cast(bf16x4, a1);
            bf16x8 fa;
#pragma unroll
            for (int j = 0; j < 4; ++j) { fa[j] = x0[j]; fa[4 + j] = x1[j]; }
#pragma unroll
            for (int b = 0; b < FB; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[b], acc[a][b], 0, 0, 0);
            if (do_db && a % WGK == wk)
                accd[a / WGK] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, ones, accd[a / WGK], 0, 0, 0);
        }
    };

    // ---- stage ring ----------------------------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < TN16_NST - 1; ++i)
        if (i < nstage) issue(i, i);
    int buf = 0;
    for (int s = 0; s < nstage; ++s) {
        if (s + 1 < nstage) tn16_wait<LPT>();  // stage s + 1 may stay in flight
        else tn16_wait<0>();
        __builtin_amdgcn_s_barrier();          // stage s has landed for every wave; everyone has left stage s - 1's buffer
        if (s + TN16_NST - 1 < nstage) {
            int nb = buf + TN16_NST - 1;
            if (nb >= TN16_NST) nb -= TN16_NST;
            issue(s + TN16_NST - 1, nb);
        }
        compute(buf);
        if (++buf == TN16_NST) buf = 0;
    }

    // ---- epilogue: lane holds C[n = .. + 4 q + j][k = .. + i16] -------------------------------------------------------------
    if (do_db) {
#pragma unroll
        for (int a = 0; a < FA; ++a) {
            if (a % WGK == wk && i16 == 0) {   // every column of the ones-product holds the same sums
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + acol + 16 * a + 4 * q + j;
                    if (n < p.N) atomicAdd(p.db + n, accd[a / WGK][j]);
                }
            }
        }
    }
#pragma unroll
    for (int a = 0; a < FA; ++a)
#pragma unroll
        for (int b = 0; b < FB; ++b) {
            const int k = k0 + bcol + 16 * b + i16;
            if (k >= p.K) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + acol + 16 * a + 4 * q + j;
                if (n >= p.N) continue;
                atomicAdd(p.C + (long)n * p.ldc + k, acc[a][b][j]);
            }
        }
}

// Item i runs on workgroup i % grid, i.e. on XCD i % 8; a problem's items start at a multiple of 8.
__global__ __launch_bounds__(512) void gemm_tn16_group_kernel(const Tn16Args* __restrict__ probs, int nprob, int total) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    for (int item = blockIdx.x; item < total; item += gridDim.x) {
        int lo = 0, hi = nprob - 1;
        while (lo < hi) {                      // last problem whose first item is <= item (workgroup-uniform)
            const int mid = (lo + hi + 1) >> 1;
            if (probs[mid].item0 <= item) lo = mid;
            else hi = mid - 1;
        }
        const Tn16Args p = probs[lo];
        const int bid = item - p.item0;
        if (p.KW > 0) {
            if (p.orient) tn16_body<3, 2, 4, 2, 5, 8, true>(p, bid, smem);
            else tn16_body<2, 3, 2, 4, 8, 5, true>(p, bid, smem);
        } else {
            if (p.orient) tn16_body<3, 2, 4, 2, 5, 8, false>(p, bid, smem);
            else tn16_body<2, 3, 2, 4, 8, 5, false>(p, bid, smem);
        }
        __syncthreads();                       // the ring is free again (all of this item's stages were waited for)
    }
}

// share of the padded tile area that is real output
double tn16_eff(int N, int K, int orient) {
    const int tn = orient ? 320 : 256, tk = orient ? 256 : 320;
    const double pn = (double)((N + tn - 1) / tn) * tn, pk = (double)((K + tk - 1) / tk) * tk;
    return (double)N * K / (pn * pk);
}

int tn16_finish(Tn16Args& a, int item0, int splits_hint) {
    a.orient = tn16_eff(a.N, a.K, 1) > tn16_eff(a.N, a.K, 0) ? 1 : 0;
    const int tn = a.orient ? 320 : 256, tk = a.orient ? 256 : 320;
    a.tiles_k = (a.K + tk - 1) / tk;
    a.tiles = ((a.N + tn - 1) / tn) * a.tiles_k;
    const long stages = (a.M + 31) / 32;
    // items of ~128 stages in a multiple of 8 m ranges; a problem with many output tiles (conv_corr: 220 tiles of 320 KB) has
    // its parallelism already, and every further m range is one more pass of f32 atomics over its whole output
    long splits = splits_hint > 0 ? splits_hint : (stages / 128 + 4) / 8 * 8;
    if (splits_hint <= 0 && ((a.N + tn - 1) / tn) * (long)a.tiles_k >= 32) splits = 8;
    if (splits < 8) splits = 8;
    splits = (splits + 7) / 8 * 8;
    a.m_per_split = ((a.M + splits - 1) / splits + 31) / 32 * 32;
    a.splits = (int)splits;
    a.item0 = item0;
    return a.tiles * a.splits;
}

}  // namespace

// 0: leave the contraction to gemm_tn8; else 1 + orientation.  The wide tiles pay where they are mostly real output.
extern "C" int emip_gemm_tn16_eligible(long M, int N, int K, long lda, long ldb) {
    if (!(M >= 2048 && (N & 7) == 0 && (K & 7) == 0 && (lda & 7) == 0 && (ldb & 7) == 0 && lda >= N && ldb >= K &&
          M * lda * 2 < 0x7FFF0000L && M * ldb * 2 < 0x7FFF0000L))
        return 0;
    const double e0 = tn16_eff(N, K, 0), e1 = tn16_eff(N, K, 1);
    const double e = e1 > e0 ? e1 : e0;
    // against gemm_tn8's 128 x 128 tiles at ~0.6 of this kernel's rate
    const double e8 = (double)N * K / ((double)((N + 127) / 128 * 128) * ((K + 127) / 128 * 128));
    if (e < 0.6 * e8) return 0;
    return e1 > e0 ? 2 : 1;
}

extern "C" int emip_conv_wgrad16_eligible(int B, int H, int Wd, int Cin, long ldx, int Cout, long lddy, int KH, int KW, int stride,
                                          int pad) {
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (Wd + 2 * pad - KW) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return 0;
    const long M = (long)B * Ho * Wo, K = (long)KH * KW * Cin;
    if (!(M >= 2048 && (Cin & 7) == 0 && (Cout & 7) == 0 && (ldx & 7) == 0 && (lddy & 7) == 0 && ldx >= Cin && lddy >= Cout &&
          K < (1L << 30) && M * lddy * 2 < 0x7FFF0000L && (long)B * H * Wd * ldx * 2 < 0x7FFF0000L))
        return 0;
    const double e0 = tn16_eff(Cout, (int)K, 0), e1 = tn16_eff(Cout, (int)K, 1);
    const double e = e1 > e0 ? e1 : e0;
    const double e8 = (double)Cout * K / ((double)((Cout + 127) / 128 * 128) * ((K + 127) / 128 * 128));
    return e >= 0.6 * e8;
}

extern "C" int emip_gemm_tn16_recsize(void) { return (int)sizeof(Tn16Args); }

// Fill one record of a grouped launch (HOST memory, emip_gemm_tn16_recsize() bytes) for C += A^T B into a PRE-CLEARED C (and
// db): returns the number of work items of the problem (a multiple of 8), or a negative error code.  item0 = the sum of the
// items of the records before it.  splits: m ranges (0 = by length; rounded up to a multiple of 8).
extern "C" int emip_gemm_tn16_plan(void* rec, const void* A, const void* B, float* C, float* db, long M, int N, int K, long lda,
                                   long ldb, long ldc, int item0, int splits) {
    if (!(rec && A && B && C && M > 0 && N > 0 && K > 0 && ldc >= K && (item0 & 7) == 0)) return EMIP_E_INVALID;
    if (!(emip_gemm_tn16_eligible(M, N, K, lda, ldb) && aligned16(A) && aligned16(B))) return EMIP_E_INVALID;
    Tn16Args a{};
    a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = C; a.db = db; a.M = M; a.N = N; a.K = K;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.a_bytes = (unsigned)(((M - 1) * lda + N) * 2);
    a.b_bytes = (unsigned)(((M - 1) * ldb + K) * 2);
    const int n = tn16_finish(a, item0, splits);
    *reinterpret_cast<Tn16Args*>(rec) = a;
    return n;
}

// ... the same for a convolution weight gradient: dW[co][ky][kx][ci] += sum_pixels dY[pix][co] X[pix shifted by the tap][ci]
extern "C" int emip_conv_wgrad16_plan(void* rec, const void* dY, const void* X, float* dW, float* db, int B, int H, int Wd, int Cin,
                                      long ldx, int Cout, long lddy, int KH, int KW, int stride, int pad, int item0, int splits) {
    if (!(rec && dY && X && dW && (item0 & 7) == 0)) return EMIP_E_INVALID;
    if (!(emip_conv_wgrad16_eligible(B, H, Wd, Cin, ldx, Cout, lddy, KH, KW, stride, pad) && aligned16(dY) && aligned16(X)))
        return EMIP_E_INVALID;
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (Wd + 2 * pad - KW) / stride + 1;
    Tn16Args a{};
    a.A = (const bf16_t*)dY; a.B = (const bf16_t*)X; a.C = dW; a.db = db;      // db (optional): the bias gradient, column sums of dY
    a.M = (long)B * Ho * Wo; a.N = Cout; a.K = KH * KW * Cin; a.lda = lddy; a.ldb = ldx; a.ldc = a.K;
    a.H = H; a.Wd = Wd; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.KW = KW; a.stride = stride; a.pad = pad;
    a.a_bytes = (unsigned)(((a.M - 1) * lddy + Cout) * 2);
    a.b_bytes = (unsigned)((((long)B * H * Wd - 1) * ldx + Cin) * 2);
    const int n = tn16_finish(a, item0, splits);
    *reinterpret_cast<Tn16Args*>(rec) = a;
    return n;
}

// probs: DEVICE array of nprob records (as the plan functions wrote them), total = the sum of their item counts
extern "C" int emip_gemm_tn16_group(const void* probs, int nprob, int total, void* stream) {
    EMIP_REQUIRE(probs && nprob > 0 && total > 0 && (reinterpret_cast<uintptr_t>(probs) & 7u) == 0);
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)gemm_tn16_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TN16_LDS) != hipSuccess)
            return EMIP_E_LAUNCH;
        attr = true;
    }
    const int grid = total < 256 ? total : 256;            // 120 KB of LDS: one workgroup (8 waves) per CU
    hipLaunchKernelGGL(gemm_tn16_group_kernel, dim3(grid), dim3(512), (size_t)TN16_LDS, (hipStream_t)stream,
                       (const Tn16Args*)probs, nprob, total);
    return emip_launch_status();
}
