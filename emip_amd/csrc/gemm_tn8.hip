// Weight-gradient contraction on an LDS-DMA stage ring (bf16, dense operands) -- the large-launch body behind emip_gemm_tn.
//
//   C[n, k] (+)= sum_m A[m, n] * B[m, k]        A = dY [M, lda], B = X [M, ldb] as the forward leaves them in HBM, f32 out
//
// Replaces gemm_tn_kernel (gemm_tn.hip) for the Linear weight gradients of the training step (row T of SURVEY.md section 8:
// /root/reference/train.py:52-58 loss.backward() through lib/pvt_v2.py:45-54,101-129), where that kernel spent 1.29 us per
// 64-row stage with 0.21 us of MFMA in it: register-staged operands, one LDS buffer, two barriers per stage, four waves.
// Here:
//   * a stage = 64 rows (m) of a 128-column slab of A and of B, 16 KB each, row-major as in HBM; both go L2 -> LDS by LDS-DMA
//     (buffer_load_dwordx4 ... lds, inline asm as in gemm8.hip), the chunk swizzle of the transposed fragment reads
//     (c ^ 2 f(row)) applied on the per-lane SOURCE offset; rows beyond the workgroup's m range and columns beyond N / K are
//     fetched from beyond the descriptor's extent, i.e. arrive as zeros;
//   * NST stages in a ring, ONE raw s_barrier per stage and a counted s_waitcnt vmcnt: NST - 2 stages stay in flight across
//     the barrier while stage s is multiplied and stage s + NST - 1 is issued into the buffer stage s - 1 has just left;
//   * MFMA 16x16x32 bf16 with both fragments read TRANSPOSED out of the row-major stage (ds_read_b64_tr_b16), 2 x 2 waves of
//     64 x 64 outputs -- the fragment code of gemm_tn.hip unchanged;
//   * the bias gradient (column sums of A) is one more MFMA per A fragment against a register of ones, on the k-tile-0
//     workgroups only: the staged tile never passes through registers, so there is nothing to sum on the way.
// M is split over workgroups; partial tiles meet in C by f32 atomics (or are stored when one workgroup owns all of M).
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 tn8_rsrc(const void* ptr, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
    return i32x4{(int)(unsigned)a, (int)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
// 64 lanes x 16 B from descriptor rs at (voff + soff) into LDS at lds_dst + 16 lane; asm so that hipcc does not drain
// vmcnt(0) in front of every LDS read (see gemm8.hip)
__device__ __forceinline__ void tn8_dma16(unsigned lds_dst, unsigned voff, i32x4 rs, unsigned soff) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_dst), "v"(voff), "s"(rs), "s"(soff)
        : "memory");
}
template <int N>
__device__ __forceinline__ void tn8_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct Tn8Args {
    const bf16_t* A;
    const bf16_t* B;
    float* C;
    float* db;            // optional: sum_m A[m][n], accumulated with atomics (cleared beforehand)
    long M;
    int N, K;
    long lda, ldb, ldc;
    long m_per_split;     // multiple of 64
    int tiles_k, tiles, splits;
    unsigned a_bytes, b_bytes;
    int atomic;           // 1: add into C with atomics (several m splits, or a pre-cleared C that others add into as well)
    int item0;            // grouped launch: first work item of this problem
    // CONV: B[m][k] is the im2col view of an NHWC tensor X (m = output pixel, k = (ky, kx, ci), ci fastest), ldb = pixel stride
    int H, Wd, Cin, Ho, Wo, KW, stride, pad;
};

constexpr unsigned TN8_OOB = 0x80000000u;
__device__ __forceinline__ int tn8_f(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

// one work item = one (output tile, m split) of problem p; bid = its index inside the problem
template <int NST, bool CONV = false>
__device__ __forceinline__ void tn8_body(const Tn8Args& p, const int bid, char* smem) {
    constexpr int RB = 256;                    // bytes per stage row (128 bf16)
    constexpr int TILE = 64 * RB;              // one operand of one stage
    constexpr int STAGE = 2 * TILE;
    constexpr int LPT = 8;                     // LDS-DMA instructions per wave and stage (4 per operand)
    static_assert(NST >= 2 && NST <= 4 && NST * STAGE <= 160 * 1024, "ring depth");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wk = wave >> 1;
    // Workgroup ids are dealt round-robin over the 8 XCDs.  With a multiple of 8 m-splits, split = id % 8 (+ 8 per block of
    // `tiles` ids): all output tiles of one m range run on ONE XCD at the same time and share its 64-row operand panels in
    // that XCD's L2 -- every panel leaves HBM / Infinity Cache once instead of once per tile that multiplies it.
    int tile, split;
    if ((p.splits & 7) == 0) {
        const int j = bid >> 3;
        tile = j % p.tiles;
        split = (bid & 7) + 8 * (j / p.tiles);
    } else {                                   // many tiles, few splits: every XCD takes a contiguous run of tiles (they
        const int rb = xcd_remap(bid, p.tiles * p.splits);      // share their dY panel and walk neighbouring X panels)
        tile = rb % p.tiles;
        split = rb / p.tiles;
    }
    const int tile_n = tile / p.tiles_k, tile_k = tile - tile_n * p.tiles_k;
    const int n0 = tile_n * 128, k0 = tile_k * 128;
    const long m_lo = (long)split * p.m_per_split;
    const long m_hi = min(p.M, m_lo + p.m_per_split);
    const int nstage = m_lo < m_hi ? (int)((m_hi - m_lo + 63) >> 6) : 0;

    const i32x4 rsA = tn8_rsrc(p.A, p.a_bytes);
    const i32x4 rsB = tn8_rsrc(p.B, p.b_bytes);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)smem;

    // ---- staging plan: instruction j (0..3) of this wave moves stage rows 16 j + 4 wave .. + 3; lane l sits at row l >> 4,
    // slot l & 15 of that 1-KB piece and fetches source chunk slot ^ 2 f(row) of its row
    unsigned aoff[4], boff[4];
    int rloc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 16 * j + 4 * wave + (lane >> 4);
        const int c = (lane & 15) ^ (tn8_f(r) << 1);
        rloc[j] = r;
        aoff[j] = n0 + 8 * c < p.N ? (unsigned)((long)r * p.lda * 2) + (unsigned)(n0 + 8 * c) * 2u : TN8_OOB;
        boff[j] = k0 + 8 * c < p.K ? (unsigned)((long)r * p.ldb * 2) + (unsigned)(k0 + 8 * c) * 2u : TN8_OOB;
    }
    // CONV: this lane's chunk of instruction j is the same (tap, channel) in every stage; the pixel it belongs to moves on
    // by 64 output pixels per stage and is tracked incrementally (b, oy, ox): no division inside the ring
    int t_ky[4], t_kx[4], t_ci[4], px_b[4], px_y[4], px_x[4];
    if (CONV) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = (lane & 15) ^ (tn8_f(rloc[j]) << 1);
            const int k = k0 + 8 * c;
            const int tap = k / p.Cin;
            t_ci[j] = k < p.K ? k - tap * p.Cin : -1;
            t_ky[j] = tap / p.KW;
            t_kx[j] = tap - t_ky[j] * p.KW;
            const long m = m_lo + rloc[j];
            const int hw = p.Ho * p.Wo;
            px_b[j] = (int)(m / hw);
            const int rem = (int)(m - (long)px_b[j] * hw);
            px_y[j] = rem / p.Wo;
            px_x[j] = rem - px_y[j] * p.Wo;
        }
    }
    auto issue = [&](int s, int buf) {
        const long m0 = m_lo + 64L * s;
        const unsigned sa = (unsigned)(m0 * p.lda * 2), sb = (unsigned)(m0 * p.ldb * 2);
        const unsigned base = lds0 + buf * STAGE + wave * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = m0 + rloc[j] < m_hi;
            tn8_dma16(base + j * 4096, ok ? aoff[j] : TN8_OOB, rsA, sa);
            if (CONV) {
                const int iy = px_y[j] * p.stride - p.pad + t_ky[j], ix = px_x[j] * p.stride - p.pad + t_kx[j];
                const bool in = ok && t_ci[j] >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.Wd;
                const unsigned off = (unsigned)((((long)px_b[j] * p.H + iy) * p.Wd + ix) * p.ldb + t_ci[j]) * 2u;
                tn8_dma16(base + TILE + j * 4096, in ? off : TN8_OOB, rsB, 0u);
                px_x[j] += 64;                                  // stages are issued in order: the next one is 64 pixels on
                while (px_x[j] >= p.Wo) {
                    px_x[j] -= p.Wo;
                    if (++px_y[j] == p.Ho) { px_y[j] = 0; ++px_b[j]; }
                }
            } else {
                tn8_dma16(base + TILE + j * 4096, ok ? boff[j] : TN8_OOB, rsB, sb);
            }
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_db = p.db != nullptr && tile_k == 0 && wk == 0;       // wave-uniform
    f32x4 accd[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) accd[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16_t)1.0f;

    const int q = lane >> 4, i16 = lane & 15;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    auto compute = [&](int buf) {
        const char* ta = smem + buf * STAGE;
        const char* tb = ta + TILE;
        const int r = i16 >> 2, pp = i16 & 3;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int row0 = 32 * ks + 8 * q + r, row1 = row0 + 4;
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int cola = wn * 64 + 16 * t + 4 * pp, colb = wk * 64 + 16 * t + 4 * pp;
                const int ca = cola >> 3, ha = (cola >> 2) & 1, cb = colb >> 3, hb = (colb >> 2) & 1;
                const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (lds_s16x4*)(ta + row0 * RB + ((ca ^ (tn8_f(row0) << 1)) * 16) + 8 * ha));
                const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (lds_s16x4*)(ta + row1 * RB + ((ca ^ (tn8_f(row1) << 1)) * 16) + 8 * ha));
                const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (lds_s16x4*)(tb + row0 * RB + ((cb ^ (tn8_f(row0) << 1)) * 16) + 8 * hb));
                const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (lds_s16x4*)(tb + row1 * RB + ((cb ^ (tn8_f(row1) << 1)) * 16) + 8 * hb));
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
            if (do_db) {
#pragma unroll
                for (int a = 0; a < 4; ++a) accd[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], ones, accd[a], 0, 0, 0);
            }
        }
    };

    // ---- stage ring ----------------------------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < NST - 1; ++i)
        if (i < nstage) issue(i, i);
    int buf = 0;
    for (int s = 0; s < nstage; ++s) {
        const int ahead = min(NST - 2, nstage - 1 - s);       // stages behind s whose loads may stay in flight
        if (ahead >= 2) tn8_wait<2 * LPT>();
        else if (ahead == 1) tn8_wait<LPT>();
        else tn8_wait<0>();
        __builtin_amdgcn_s_barrier();          // stage s has landed for every wave; everyone has left stage s - 1's buffer
        if (s + NST - 1 < nstage) {
            int nb = buf + NST - 1;
            if (nb >= NST) nb -= NST;
            issue(s + NST - 1, nb);
        }
        compute(buf);
        if (++buf == NST) buf = 0;
    }

    // ---- epilogue: lane holds C[n = .. + 4 q + j][k = .. + i16] -------------------------------------------------------------
    if (do_db) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (i16 == 0) {                    // every column of the ones-product holds the same sums
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + wn * 64 + 16 * a + 4 * q + j;
                    if (n < p.N) atomicAdd(p.db + n, accd[a][j]);
                }
            }
        }
    }
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
                float* dst = p.C + (long)n * p.ldc + k;
                if (p.atomic) atomicAdd(dst, acc[a][b][j]);
                else *dst = acc[a][b][j];
            }
        }
}

template <int NST, bool CONV = false>
__global__ __launch_bounds__(256) void gemm_tn8_kernel(const Tn8Args p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    tn8_body<NST, CONV>(p, (int)blockIdx.x, smem);
}

// Grouped launch: the weight gradients of MANY layers in one persistent grid.  The training step defers its Linear weight
// gradients (their operands dY and X stay alive: the chip has 288 GB) and flushes them together, so that the launch holds
// thousands of work items: no launch is short of tiles any more, m splits exist only to bound the item length (fewer
// atomics), and every CU walks a similar mix of items.  Item i runs on workgroup i % grid, i.e. on XCD i % 8; a problem's
// items start at a multiple of 8, so the id decode of tn8_body keeps the tiles of one m range on one XCD.
template <int NST>
__global__ __launch_bounds__(256) void gemm_tn8_group_kernel(const Tn8Args* __restrict__ probs, int nprob, int total) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    for (int item = blockIdx.x; item < total; item += gridDim.x) {
        int lo = 0, hi = nprob - 1;
        while (lo < hi) {                      // last problem whose first item is <= item (workgroup-uniform)
            const int mid = (lo + hi + 1) >> 1;
            if (probs[mid].item0 <= item) lo = mid;
            else hi = mid - 1;
        }
        const Tn8Args p = probs[lo];
        tn8_body<NST>(p, item - p.item0, smem);
        __syncthreads();                       // the ring is free again (all of this item's stages were waited for)
    }
}

int g_tn8_nst = 0;        // ring depth; 0 = heuristic (tools/gemm_tn_bench.py sweeps it through the tuning library)
int g_tn8_target = 0;     // workgroups to aim at; 0 = heuristic
int g_tn8_xcd = 1;        // m splits in multiples of 8, one m range per XCD

}  // namespace

#ifdef EMIP_TUNING
extern "C" int emip_debug_set_tn8(int nst, int target) {       // nst < 0: |nst| stages without the XCD-aware split map
    g_tn8_xcd = nst >= 0;
    g_tn8_nst = nst < 0 ? -nst : nst;
    g_tn8_target = target;
    return EMIP_OK;
}
#endif

// 1 if the ring body takes this contraction (bf16, dense, offsets within the 32-bit descriptor range)
extern "C" int emip_gemm_tn8_eligible(long M, int N, int K, long lda, long ldb) {
    return M >= 2048 && (N & 7) == 0 && (K & 7) == 0 && (lda & 7) == 0 && (ldb & 7) == 0 && lda >= N && ldb >= K &&
           M * lda * 2 < 0x7FFF0000L && M * ldb * 2 < 0x7FFF0000L;
}

// C (f32 [N][ldc]) = or += A^T B.  prezeroed: C / db are already clear and are ADDED into; otherwise this call clears them
// when it has to (several m splits).  db may be NULL.
extern "C" int emip_gemm_tn8(const void* A, const void* B, float* C, float* db, long M, int N, int K, long lda, long ldb,
                             long ldc, int prezeroed, void* stream) {
    EMIP_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && ldc >= K);
    EMIP_REQUIRE(emip_gemm_tn8_eligible(M, N, K, lda, ldb) && aligned16(A) && aligned16(B));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    Tn8Args a{};
    a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = C; a.db = db; a.M = M; a.N = N; a.K = K;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.a_bytes = (unsigned)(((M - 1) * lda + N) * 2);
    a.b_bytes = (unsigned)(((M - 1) * ldb + K) * 2);
    const int tiles_n = (N + 127) / 128;
    a.tiles_k = (K + 127) / 128;
    const long tiles = (long)tiles_n * a.tiles_k;
    // Measured on the PVTv2-b5 shapes at 64 images (tools/gemm_tn_bench.py): launches with >= 8 output tiles run best on a
    // 2-deep ring (64 KB: two workgroups per CU; 512 workgroups from 16 tiles on, e.g. 30976 x 1280 x 320: 97.8 us on the
    // register-staged body, 50.4 here), launches with few tiles -- all parallelism from m splits -- on a 3-deep ring with
    // one workgroup per CU.  Every split adds a 128 x 128 tile of f32 atomics (1.3 TB/s chip-wide), hence >= 8 stages each.
    const int nst = g_tn8_nst > 0 ? g_tn8_nst : (tiles >= 8 ? 2 : 3);
    const long target = g_tn8_target > 0 ? g_tn8_target : (tiles >= 16 ? 512 : 256);
    long splits = (target + tiles - 1) / tiles;
    const long max_splits = (M + 8 * 64 - 1) / (8 * 64);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits >= 6 && g_tn8_xcd && max_splits >= 8) {      // a multiple of 8: one m range per XCD (the kernel's id decode)
        splits = (splits + 4) / 8 * 8;
        if (splits > max_splits) splits = max_splits / 8 * 8;
    }
    a.m_per_split = ((M + splits - 1) / splits + 63) / 64 * 64;
    if ((splits & 7) != 0) splits = (M + a.m_per_split - 1) / a.m_per_split;     // (a multiple of 8 keeps its map; a tail
                                                                                 // split left without rows adds zeros)
    a.tiles = (int)tiles;
    a.splits = (int)splits;
    a.atomic = splits > 1 || prezeroed;
    if (!prezeroed) {
        if (splits > 1) {
            if (ldc != K) return EMIP_E_INVALID;
            const bool joint = db != nullptr && db == C + (size_t)N * K;
            if (emip_zero_async(C, sizeof(float) * ((size_t)N * K + (joint ? N : 0)), s) != EMIP_OK) return EMIP_E_LAUNCH;
            if (db && !joint && emip_zero_async(db, sizeof(float) * N, s) != EMIP_OK) return EMIP_E_LAUNCH;
        } else if (db && emip_zero_async(db, sizeof(float) * N, s) != EMIP_OK) {
            return EMIP_E_LAUNCH;
        }
    }
    dim3 grid((unsigned)(tiles * splits));
    const size_t lds = (size_t)nst * 2 * 64 * 256;
    if (nst == 2) {
        hipLaunchKernelGGL(gemm_tn8_kernel<2>, grid, dim3(256), lds, s, a);
    } else if (nst == 4) {
        static bool attr4 = false;
        if (!attr4) { (void)hipFuncSetAttribute((const void*)gemm_tn8_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr4 = true; }
        hipLaunchKernelGGL(gemm_tn8_kernel<4>, grid, dim3(256), lds, s, a);
    } else {
        static bool attr3 = false;
        if (!attr3) { (void)hipFuncSetAttribute((const void*)gemm_tn8_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr3 = true; }
        hipLaunchKernelGGL(gemm_tn8_kernel<3>, grid, dim3(256), lds, s, a);
    }
    return emip_launch_status();
}

// ---- conv weight gradient on the ring: dW[co][ky][kx][ci] (+)= sum_pixels dY[pix][co] X[pix shifted by the tap][ci] ----------------
extern "C" int emip_conv_wgrad8_eligible(int B, int H, int Wd, int Cin, long ldx, int Cout, long lddy, int KH, int KW, int stride,
                                         int pad) {
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (Wd + 2 * pad - KW) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return 0;
    const long M = (long)B * Ho * Wo, K = (long)KH * KW * Cin;
    return M >= 2048 && (Cin & 7) == 0 && (Cout & 7) == 0 && (ldx & 7) == 0 && (lddy & 7) == 0 && ldx >= Cin && lddy >= Cout &&
           K < (1L << 30) && M * lddy * 2 < 0x7FFF0000L && (long)B * H * Wd * ldx * 2 < 0x7FFF0000L;
}

extern "C" int emip_conv_wgrad8(const void* dY, const void* X, float* dW, int B, int H, int Wd, int Cin, long ldx, int Cout,
                                long lddy, int KH, int KW, int stride, int pad, int prezeroed, void* stream) {
    EMIP_REQUIRE(dY && X && dW && emip_conv_wgrad8_eligible(B, H, Wd, Cin, ldx, Cout, lddy, KH, KW, stride, pad));
    EMIP_REQUIRE(aligned16(dY) && aligned16(X));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (Wd + 2 * pad - KW) / stride + 1;
    Tn8Args a{};
    a.A = (const bf16_t*)dY; a.B = (const bf16_t*)X; a.C = dW; a.db = nullptr;
    a.M = (long)B * Ho * Wo; a.N = Cout; a.K = KH * KW * Cin; a.lda = lddy; a.ldb = ldx; a.ldc = a.K;
    a.H = H; a.Wd = Wd; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.KW = KW; a.stride = stride; a.pad = pad;
    a.a_bytes = (unsigned)(((a.M - 1) * lddy + Cout) * 2);
    a.b_bytes = (unsigned)((((long)B * H * Wd - 1) * ldx + Cin) * 2);
    a.tiles_k = (a.K + 127) / 128;
    a.tiles = ((a.N + 127) / 128) * a.tiles_k;
    // few tiles: m splits in multiples of 8 (one m range per XCD); many tiles (conv_corr: 1096): two splits keep the last
    // round of the 512 resident workgroups from running half empty
    const long max_splits = (a.M + 8 * 64 - 1) / (8 * 64);
    long splits = (512 + a.tiles - 1) / a.tiles;
    if (a.tiles >= 512) splits = 2;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits >= 6 && max_splits >= 8) {
        splits = (splits + 4) / 8 * 8;
        if (splits > max_splits) splits = max_splits / 8 * 8;
    }
    a.m_per_split = ((a.M + splits - 1) / splits + 63) / 64 * 64;
    if ((splits & 7) != 0) splits = (a.M + a.m_per_split - 1) / a.m_per_split;
    a.splits = (int)splits;
    a.atomic = splits > 1 || prezeroed;
    if (!prezeroed && splits > 1 && emip_zero_async(dW, sizeof(float) * (size_t)a.N * a.K, s) != EMIP_OK) return EMIP_E_LAUNCH;
    hipLaunchKernelGGL((gemm_tn8_kernel<2, true>), dim3((unsigned)(a.tiles * splits)), dim3(256), (size_t)2 * 2 * 64 * 256, s, a);
    return emip_launch_status();
}

// ---- grouped launch ------------------------------------------------------------------------------------------------------
extern "C" int emip_gemm_tn8_group_recsize(void) { return (int)sizeof(Tn8Args); }

// Fill one record of a grouped launch (HOST memory, emip_gemm_tn8_group_recsize() bytes) for C += A^T B into a PRE-CLEARED C
// (and db): returns the number of work items of the problem (a multiple of 8), or a negative error code.  item0 = the sum
// of the items of the records before it.
extern "C" int emip_gemm_tn8_group_plan(void* rec, const void* A, const void* B, float* C, float* db, long M, int N, int K,
                                        long lda, long ldb, long ldc, int item0) {
    if (!(rec && A && B && C && M > 0 && N > 0 && K > 0 && ldc >= K && (item0 & 7) == 0)) return EMIP_E_INVALID;
    if (!(emip_gemm_tn8_eligible(M, N, K, lda, ldb) && aligned16(A) && aligned16(B))) return EMIP_E_INVALID;
    Tn8Args a{};
    a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = C; a.db = db; a.M = M; a.N = N; a.K = K;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.a_bytes = (unsigned)(((M - 1) * lda + N) * 2);
    a.b_bytes = (unsigned)(((M - 1) * ldb + K) * 2);
    a.tiles_k = (K + 127) / 128;
    a.tiles = ((N + 127) / 128) * a.tiles_k;
    const long stages = (M + 63) / 64;
    long splits = (stages / 60 + 4) / 8 * 8;               // items of ~60 stages, a multiple of 8 m ranges (one per XCD)
    if (splits < 8) splits = 8;
    a.m_per_split = ((M + splits - 1) / splits + 63) / 64 * 64;
    a.splits = (int)splits;
    a.atomic = 1;
    a.item0 = item0;
    *reinterpret_cast<Tn8Args*>(rec) = a;
    return a.tiles * a.splits;
}

// probs: DEVICE array of nprob records (as emip_gemm_tn8_group_plan wrote them), total = the sum of their item counts
extern "C" int emip_gemm_tn8_group(const void* probs, int nprob, int total, void* stream) {
    EMIP_REQUIRE(probs && nprob > 0 && total > 0 && (reinterpret_cast<uintptr_t>(probs) & 7u) == 0);
    const int grid = total < 512 ? total : 512;            // 2-deep ring (64 KB): two workgroups per CU
    hipLaunchKernelGGL(gemm_tn8_group_kernel<2>, dim3(grid), dim3(256), (size_t)2 * 2 * 64 * 256, (hipStream_t)stream,
                       (const Tn8Args*)probs, nprob, total);
    return emip_launch_status();
}
