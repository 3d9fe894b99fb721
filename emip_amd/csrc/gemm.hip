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
    // batching over blockIdx.z; with heads > 1 the batch index z = zb * heads + zh and an operand sits at zb * bs + zh * hs
    // (per-head column slices of [B][tokens][heads*64] tensors)
    long bsA, bsW, bsC, bsR;
    long hsA, hsW, hsC;
    int heads;
    int tiles_m, tiles_n;
    // LayerNorm elimination: ln_stats [rows of A][2] = (sum x, sum x^2) over ln_C channels of every A
    // row / input pixel -> the operand loader feeds (x - mean) * rstd to the MFMA (gamma / beta are folded into W / bias by
    // the host); out_stats [M][2]: the epilogue accumulates the same sums of the rows it writes (zeroed by an earlier kernel)
    const float* ln_stats;
    float* out_stats;
    int ln_C;
    float ln_eps;
    // The same LayerNorm applied on the OUTPUT side of a dense GEMM: the product runs on the raw rows (any main loop, in
    // particular the LDS-DMA one) and the epilogue uses  xhat W^T = rstd (x W^T) - rstd mean colsum(W):
    // lne_stats [M][2] like ln_stats, lne_colsum [N] = sum_k W[n][k] of the packed (rounded) weights.  Two FMAs per
    // output element instead of a normalisation of every staged operand element in every N tile.
    const float* lne_stats;
    const float* lne_colsum;
    // split-K (small-M, long-K launches): blockIdx.y owns a range of K tiles and adds its partial tile into acc_out (f32
    // [M][ldacc], zero beforehand) with atomics; bias is added by split 0; the row cast / statistics happen in
    // emip_rows_finalize
    float* acc_out;
    long ldacc;
    int ksplit;
    // fused split-K: with `ticket` (one zeroed counter per output tile) the LAST split workgroup of a tile to arrive takes the
    // summed tile back out of acc_out (atomic exchange with 0, so accumulator and counter are clean for the next launch)
    // and runs the normal epilogue: no finalize launch, no zero-fill
    unsigned* ticket;
    unsigned* zero_ptr;   // optional scratch the first workgroup clears (saves the consumer's zero-fill launch)
    long zero_words;
    int dbg;   // experiment flags (emip_debug_set key 2): 1 = skip epilogue stores, 2 = skip global loads in the K loop
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

// NBUF = 2: two LDS buffers, one barrier per K tile (64 KB at 128x128 -> 2 workgroups per CU).
// NBUF = 1: one LDS buffer, two barriers per K tile (32 KB -> 4-5 workgroups per CU): more tiles
//           in flight per CU, which is what the short-K GEMMs of this network need (K = 64..1280,
//           each workgroup has only 1..20 K tiles to hide HBM/L2 latency behind).
// NBUF = 3: LDS-DMA variant (dense GEMM with K a multiple of the K tile): global_load_lds_dwordx4 writes the
//           tiles straight into two LDS buffers (no staging VGPRs, no ds_write traffic -- the register-staged
//           loop is bound by the ds_write_b128 rate); the XOR swizzle moves to the per-lane SOURCE address.
template <typename T>
__device__ __forceinline__ uint4 ln_apply(uint4 v, float mu, float rs) {
    T* e = reinterpret_cast<T*>(&v);
#pragma unroll
    for (int j = 0; j < (int)(16 / sizeof(T)); ++j) e[j] = from_f32<T>((to_f32<T>(e[j]) - mu) * rs);
    return v;
}
// bf16: the normalisation sits on the K loop's critical path (loaded chunk -> LDS store), so it is written as one packed
// FMA per channel pair, x * rstd + (-mean * rstd), on values unpacked with a shift / mask
template <>
__device__ __forceinline__ uint4 ln_apply<bf16_t>(uint4 v, float mu, float rs) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const float nmr = -mu * rs;
    unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f32x2 x = {__uint_as_float(w[k] << 16), __uint_as_float(w[k] & 0xFFFF0000u)};
        x = x * rs + nmr;
        bf16x2 o;
        o[0] = (bf16_t)x.x;
        o[1] = (bf16_t)x.y;
        w[k] = __builtin_bit_cast(unsigned, o);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// The kernel body is a device function over an explicit block id so that one launch can host several problems
// (gemm_pair_kernel below): bx = tile index, by = split-K index, bz = batch index, gx = number of tile blocks.
// NOPAD (CONV only): the caller guarantees pad == 0, so every tap of a valid output pixel lies inside the image and the
// loader skips the bounds tests and clamps (as a template flag: a run-time branch around the loads would serialise them)
template <typename T, int BM, int BN, bool CONV, int NBUF, bool LNA, bool NOPAD = false>
__device__ __forceinline__ void gemm_body(const GemmArgs& p, const int bx, const int by, const int bz, const int gx) {
    constexpr int VEC = 16 / sizeof(T);   // elements per 16-B chunk
    constexpr int BK = 128 / sizeof(T);   // K elements per tile
    constexpr int TM = BM / 32, TN = BN / 32;  // 16x16 sub-tiles per wave
    constexpr int CA = BM / 32, CW = BN / 32;  // 16-B chunks staged per thread
    constexpr int TILE_BYTES = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;

    if (p.zero_ptr && bz == 0 && by == 0) {
        // scratch of LATER kernels in stream order (statistics, split-K accumulators): cleared here by all workgroups
        for (long i = (long)bx * 256 + tid; i < p.zero_words; i += (long)gx * 256) p.zero_ptr[i] = 0u;
    }
    const int ntile = p.tiles_m * p.tiles_n;
    const int swz = xcd_remap(bx, ntile);
    const int tile_m = swz / p.tiles_n, tile_n = swz - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const long z = bz;

    const long zb = p.heads > 1 ? z / p.heads : z, zh = p.heads > 1 ? z - zb * p.heads : 0;
    const T* __restrict__ A = reinterpret_cast<const T*>(p.A) + zb * p.bsA + zh * p.hsA;
    const T* __restrict__ A2 = p.A2 ? reinterpret_cast<const T*>(p.A2) + z * p.bsA : nullptr;
    const T* __restrict__ Wp = reinterpret_cast<const T*>(p.W) + zb * p.bsW + zh * p.hsW;

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
            a_off[i] = (long)(a_ok[i] ? m : p.M - 1);
            a_iy[i] = a_ix[i] = 0;
        }
    }
    bool w_ok[CW];
    long w_off[CW];
#pragma unroll
    for (int i = 0; i < CW; ++i) {
        const int n = n0 + srow + 32 * i;
        w_ok[i] = n < p.N;
        w_off[i] = (long)(w_ok[i] ? n : p.N - 1) * p.ldw;
    }

    // CONV: per-chunk image base pointers (the pixel offset inside an image fits 32 bits: H * W * lda < 2^31 is checked by
    // the entry points) and the wave-uniform tap state of load_tile_ln
    const T* a_base[CA];
#pragma unroll
    for (int i = 0; i < CA; ++i) a_base[i] = CONV ? A + a_off[i] * p.lda : A;
    const bool cin_tiled = CONV && (p.Cin % BK) == 0;
    int tap_next_k0 = -1, tap_ci0 = 0, tap_ky = 0, tap_kx = 0;
    (void)cin_tiled; (void)tap_next_k0; (void)tap_ci0; (void)tap_ky; (void)tap_kx;

    // Loads are UNCONDITIONAL on clamped (always valid) addresses and masked afterwards: a load inside an
    // `if` makes hipcc branch around it and wait for it on the spot, which serialises the tile's loads.
    // LNA: per-row (sum, sum of squares) of the A operand's source rows; the normalisation itself happens in store_tile, when
    // the data has arrived, so that the global loads of a tile stay back to back
    struct LnStage {
        float2 st[LNA ? CA : 1];
        unsigned ok;
    };
    float2 ln_row[LNA ? CA : 1];          // dense GEMM: the statistics of this thread's rows never change
    if (LNA && !CONV) {
#pragma unroll
        for (int i = 0; i < CA; ++i) ln_row[LNA ? i : 0] = *reinterpret_cast<const float2*>(p.ln_stats + 2 * a_off[i]);
    }
    LnStage ln0, ln1, ln2;                // one per register stage (only ln0 is live outside the 3-stage loop)
    (void)ln1; (void)ln2;
    auto load_tile_ln = [&](int k0, uint4 (&ra)[CA], uint4 (&rw)[CW], LnStage& ln) {
        const int kk = k0 + sc * VEC;
        const bool kok = kk < p.K;
        const int kkc = kok ? kk : 0;
        if (LNA) ln.ok = 0;
        if (CONV) {
            // tap decode.  When Cin is a multiple of the K tile, a tile lies inside one tap and consecutive calls walk the
            // taps in order: the (ky, kx, channel base) state is carried in wave-uniform registers instead of two integer
            // divisions per thread and tile (the loader's address arithmetic, not the MFMA, is what a 64x64 conv tile
            // spends its issue slots on)
            int ci, ky, kx;
            if (cin_tiled) {
                if (k0 != tap_next_k0) {                 // first call of this workgroup (or a jump): decode once
                    const int tap = k0 / p.Cin;
                    tap_ci0 = k0 - tap * p.Cin;
                    tap_ky = tap / p.KW;
                    tap_kx = tap - tap_ky * p.KW;
                }
                ci = tap_ci0 + sc * VEC; ky = tap_ky; kx = tap_kx;
                tap_next_k0 = k0 + BK;
                tap_ci0 += BK;
                if (tap_ci0 >= p.Cin) {
                    tap_ci0 = 0;
                    if (++tap_kx == p.KW) { tap_kx = 0; ++tap_ky; }
                }
            } else {
                const int tap = kkc / p.Cin;
                ci = kkc - tap * p.Cin;
                ky = tap / p.KW; kx = tap - ky * p.KW;
            }
            {
#pragma unroll
                for (int i = 0; i < CA; ++i) {
                    const int iy = a_iy[i] + ky, ix = a_ix[i] + kx;
                    const bool ok = NOPAD ? (kok && a_ok[i])
                                          : (kok && a_ok[i] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.Wd);
                    const int iyc = NOPAD ? iy : min(max(iy, 0), p.H - 1), ixc = NOPAD ? ix : min(max(ix, 0), p.Wd - 1);
                    const int po = iyc * p.Wd + ixc;
                    const uint4 v = *reinterpret_cast<const uint4*>(a_base[i] + (long)(po * (int)p.lda + ci));
                    ra[i] = mask4(v, ok);
                    if (LNA) {
                        ln.st[LNA ? i : 0] = *reinterpret_cast<const float2*>(p.ln_stats + 2 * (a_off[i] + po));
                        ln.ok |= ok ? (1u << i) : 0u;
                    }
                }
            }
        } else {
            const bool second = kkc >= p.K1;
            const T* base = second ? A2 : A;
            const long ld = second ? p.lda2 : p.lda;
            const int kc = second ? kkc - p.K1 : kkc;
#pragma unroll
            for (int i = 0; i < CA; ++i) {
                const uint4 v = *reinterpret_cast<const uint4*>(base + a_off[i] * ld + kc);
                ra[i] = mask4(v, kok && a_ok[i]);
                if (LNA) {
                    ln.st[LNA ? i : 0] = ln_row[LNA ? i : 0];
                    ln.ok |= (kok && a_ok[i]) ? (1u << i) : 0u;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < CW; ++i) {
            const uint4 v = *reinterpret_cast<const uint4*>(Wp + w_off[i] + kkc);
            rw[i] = mask4(v, kok && w_ok[i]);
        }
    };
    auto load_tile = [&](int k0, uint4 (&ra)[CA], uint4 (&rw)[CW]) { load_tile_ln(k0, ra, rw, ln0); };
    auto store_tile_ln = [&](int buf, const uint4 (&ra)[CA], const uint4 (&rw)[CW], const LnStage& ln) {
        char* ta = smem + buf * TILE_BYTES;
        char* tw = ta + BM * 128;
#pragma unroll
        for (int i = 0; i < CA; ++i) {
            uint4 v = ra[i];
            if (LNA) {
                const float2 st = ln.st[LNA ? i : 0];
                const float mu = st.x / (float)p.ln_C;
                const float rs = rsqrtf(fmaxf(st.y / (float)p.ln_C - mu * mu, 0.f) + p.ln_eps);
                v = mask4(ln_apply<T>(v, mu, rs), (ln.ok >> i) & 1u);      // padding / tail stays exactly zero
            }
            *reinterpret_cast<uint4*>(ta + (srow + 32 * i) * 128 + swz_c) = v;
        }
#pragma unroll
        for (int i = 0; i < CW; ++i) *reinterpret_cast<uint4*>(tw + (srow + 32 * i) * 128 + swz_c) = rw[i];
    };
    auto store_tile = [&](int buf, const uint4 (&ra)[CA], const uint4 (&rw)[CW]) { store_tile_ln(buf, ra, rw, ln0); };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    const int fsw = (fr >> 1) & 7;  // sub-tile bases are multiples of 16 -> (row>>1)&7 == (fr>>1)&7

    auto compute_tile = [&](int buf) {
        const char* ta = smem + buf * TILE_BYTES;
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
    };

    const int nk_all = (p.K + BK - 1) / BK;
    // split-K: this workgroup's range of K tiles (register-staged loops only)
    const int kt_per = p.ksplit > 1 ? (nk_all + p.ksplit - 1) / p.ksplit : nk_all;
    const int kt_lo = p.ksplit > 1 ? by * kt_per : 0;
    const int nk = min(nk_all, kt_lo + kt_per);
    if (NBUF == 3) {
        // One wave instruction moves 64 x 16 B = 8 tile rows (LDS destination = wave-uniform base + lane*16).
        // Wave w stages rows [w*BM/4, (w+1)*BM/4) of the A tile and [w*BN/4, ...) of the W tile.
        typedef __attribute__((address_space(3))) void lds_void;
        typedef const __attribute__((address_space(1))) void glb_void;
        const int lrow = lane >> 3, lslot = lane & 7;
        const T* ga[BM / 32];
        const T* gw[BN / 32];
        const T* ga2[BM / 32];
#pragma unroll
        for (int i = 0; i < BM / 32; ++i) {
            const int r = wave * (BM / 4) + 8 * i + lrow;                 // tile row
            const int c = lslot ^ ((r >> 1) & 7);                          // source chunk that belongs in this slot
            const long m = min((long)(m0 + r), (long)p.M - 1);             // clamped: rows >= M are never stored
            ga[i] = A + m * p.lda + c * VEC;
            ga2[i] = A2 ? A2 + m * p.lda2 + c * VEC : ga[i];
        }
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) {
            const int r = wave * (BN / 4) + 8 * i + lrow;
            const int c = lslot ^ ((r >> 1) & 7);
            const long n = min((long)(n0 + r), (long)p.N - 1);
            gw[i] = Wp + n * p.ldw + c * VEC;
        }
        auto issue = [&](int kt, int buf) {
            char* ta = smem + buf * TILE_BYTES + wave * (BM / 4) * 128;
            char* tw = smem + buf * TILE_BYTES + BM * 128 + wave * (BN / 4) * 128;
            const int k0 = kt * BK;
            const bool second = k0 >= p.K1;
#pragma unroll
            for (int i = 0; i < BM / 32; ++i) {
                const T* src = second ? ga2[i] + (k0 - p.K1) : ga[i] + k0;
                __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(ta + i * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < BN / 32; ++i)
                __builtin_amdgcn_global_load_lds((glb_void*)(gw[i] + k0), (lds_void*)(tw + i * 1024), 16, 0, 0);
        };
        issue(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) issue(kt + 1, cur ^ 1);
            compute_tile(cur);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else if (NBUF == 2) {
        uint4 ra[CA], rw[CW];
        load_tile(0, ra, rw);
        store_tile(0, ra, rw);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) load_tile((kt + 1) * BK, ra, rw);
            compute_tile(cur);
            if (kt + 1 < nk) store_tile(cur ^ 1, ra, rw);
            __syncthreads();
        }
    } else if (NBUF == 6) {
        // KPB = 2 K tiles per barrier interval (as many LDS buffers and register stages): a 64x64 tile's K step is a
        // latency chain (LDS store -> barrier -> ds_read -> 8 MFMAs -> barrier, ~0.65 us with one workgroup per CU whatever
        // the prefetch depth), so the long-K small-grid convs cut the number of chains instead of trying to hide them.  The
        // next group's global loads are issued before the multiplications.
        constexpr int KPB = 2;       // 4 (64 KB of LDS for the whole pair launch) measured slower: 965 vs 984 pairs/s
        uint4 ra[KPB][CA], rw[KPB][CW];
        LnStage lns[KPB];
#pragma unroll
        for (int j = 0; j < KPB; ++j)
            if (kt_lo + j < nk) load_tile_ln((kt_lo + j) * BK, ra[j], rw[j], lns[j]);
        for (int kt = kt_lo; kt < nk; kt += KPB) {
#pragma unroll
            for (int j = 0; j < KPB; ++j)
                if (kt + j < nk) store_tile_ln(j, ra[j], rw[j], lns[j]);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < KPB; ++j)
                if (kt + KPB + j < nk) load_tile_ln((kt + KPB + j) * BK, ra[j], rw[j], lns[j]);
#pragma unroll
            for (int j = 0; j < KPB; ++j)
                if (kt + j < nk) compute_tile(j);
            __syncthreads();
        }
    } else if (NBUF == 5) {
        // one LDS buffer, THREE register stages: the global loads of K tiles kt+1 and kt+2 are in flight while tile kt is
        // multiplied.  For grids of about one workgroup per CU (the 121-token spatial-reduction convs: 16-80 workgroups
        // walking K = 1280-4096) nothing else hides the load latency; the staging registers are cheap at 64x64 tiles.
        uint4 ra0[CA], rw0[CW], ra1[CA], rw1[CW], ra2[CA], rw2[CW];
        if (kt_lo < nk) load_tile_ln(kt_lo * BK, ra0, rw0, ln0);
        if (kt_lo + 1 < nk) load_tile_ln((kt_lo + 1) * BK, ra1, rw1, ln1);
        for (int kt = kt_lo; kt < nk; kt += 3) {
            if (kt + 2 < nk) load_tile_ln((kt + 2) * BK, ra2, rw2, ln2);
            store_tile_ln(0, ra0, rw0, ln0);
            __syncthreads();
            compute_tile(0);
            __syncthreads();
            if (kt + 1 < nk) {
                if (kt + 3 < nk) load_tile_ln((kt + 3) * BK, ra0, rw0, ln0);
                store_tile_ln(0, ra1, rw1, ln1);
                __syncthreads();
                compute_tile(0);
                __syncthreads();
            }
            if (kt + 2 < nk) {
                if (kt + 4 < nk) load_tile_ln((kt + 4) * BK, ra1, rw1, ln1);
                store_tile_ln(0, ra2, rw2, ln2);
                __syncthreads();
                compute_tile(0);
                __syncthreads();
            }
        }
    } else {
        // one LDS buffer, two barriers per K tile: half the LDS, more workgroups per CU
        uint4 ra[CA], rw[CW];
        load_tile(kt_lo * BK, ra, rw);
        store_tile(0, ra, rw);
        __syncthreads();
        for (int kt = kt_lo; kt < nk; ++kt) {
#ifdef EMIP_TUNING
            if (kt + 1 < nk && !(p.dbg & 2)) load_tile((kt + 1) * BK, ra, rw);
#else
            if (kt + 1 < nk) load_tile((kt + 1) * BK, ra, rw);
#endif
            compute_tile(0);
            __syncthreads();
            if (kt + 1 < nk) store_tile(0, ra, rw);
            __syncthreads();
        }
    }

#ifdef EMIP_TUNING
    if ((p.dbg & 1) && acc[0][0][0] != 123456.f) return;   // tuning library only: no epilogue (the test keeps acc live)
#endif
    if (p.acc_out) {
        // split-K partial tile: lane holds channels n .. n+3 of row m (swapped operands), straight to f32 atomics
        const bool first = by == 0;
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            const int n = n0 + wn * (BN / 2) + 16 * a + 4 * fq;
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                const int m = m0 + wm * (BM / 2) + 16 * b + fr;
                if (m < p.M) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (n + j < p.N) {
                            const float v = acc[a][b][j] + ((first && p.bias && !p.ticket) ? p.bias[n + j] : 0.f);
                            atomicAdd(p.acc_out + (long)m * p.ldacc + n + j, v);
                        }
                    }
                }
            }
        }
        if (!p.ticket) return;
        // Order this workgroup's accumulator adds before its ticket: no-return atomics are only known to have landed
        // once the issuing wave has waited for vmcnt(0) -- a workgroup barrier alone emits no such wait on gfx950 -- so
        // every wave drains, the workgroup meets, and one lane draws the ticket.  Every access to the accumulator and the
        // ticket, here and in the read-back below, is a device-scope atomic performed at the memory side: no fence is needed,
        // and an agent-scope fence is NOT free on this part -- release writes back, acquire invalidates the XCD's whole L2
        // (round 4: the same pair of fences in gemm8's statistics combine took the step from 2230 to 1310 pairs/s).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* flag = reinterpret_cast<unsigned*>(smem);
        if (tid == 0) *flag = __hip_atomic_fetch_add(p.ticket + swz, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*flag != (unsigned)p.ksplit - 1u) return;
        if (tid == 0) atomicExch(p.ticket + swz, 0u);
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            const int n = n0 + wn * (BN / 2) + 16 * a + 4 * fq;
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                const int m = m0 + wm * (BM / 2) + 16 * b + fr;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[a][b][j] = (m < p.M && n + j < p.N) ? atomicExch(p.acc_out + (long)m * p.ldacc + n + j, 0.f) : 0.f;
            }
        }
    }
    T* C = reinterpret_cast<T*>(p.C) + zb * p.bsC + zh * p.hsC;   // may alias R (in-place residual update)
    const T* R = p.R ? reinterpret_cast<const T*>(p.R) + z * p.bsR : nullptr;
    constexpr int WM = BM / 2, WN = BN / 2, EP_LD = WN + 4;
    constexpr int CG = WN / VEC;        // 16-B column groups per row of the wave block
    constexpr int RPI = 64 / CG;        // rows covered by one wave instruction
    float* ep = reinterpret_cast<float*>(smem) + wave * 16 * EP_LD;
    const int cg = lane % CG, er0 = lane / CG;
    const bool fast = ((p.ldc % VEC) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
                      (R == nullptr || (((p.ldr % VEC) == 0) && ((reinterpret_cast<uintptr_t>(R) & 15) == 0)));
    const int nb = n0 + wn * WN;
    const bool act_gelu = p.act == EMIP_ACT_GELU, act_relu = p.act == EMIP_ACT_RELU;
    float4 bias_v[TN];      // this lane's 4 output channels of each sub-tile column, loaded once
    float4 csum_v[TN];      // the same channels of lne_colsum (output-side LayerNorm)
#pragma unroll
    for (int a = 0; a < TN; ++a) {
        csum_v[a] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.lne_stats) {
            const int n = nb + 16 * a + 4 * fq;
            csum_v[a] = make_float4(p.lne_colsum[min(n + 0, p.N - 1)], p.lne_colsum[min(n + 1, p.N - 1)],
                                    p.lne_colsum[min(n + 2, p.N - 1)], p.lne_colsum[min(n + 3, p.N - 1)]);
        }
    }
#pragma unroll
    for (int a = 0; a < TN; ++a) {
        bias_v[a] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) {
            const int n = nb + 16 * a + 4 * fq;
            const int nc = min(n, max(p.N - 4, 0));     // clamped 4-wide window, masked per element below
            const float b0 = p.bias[min(n + 0, p.N - 1)], b1 = p.bias[min(n + 1, p.N - 1)];
            const float b2 = p.bias[min(n + 2, p.N - 1)], b3 = p.bias[min(n + 3, p.N - 1)];
            (void)nc;
            bias_v[a] = make_float4(b0, b1, b2, b3);    // columns >= N are never stored
        }
    }
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        uint4 rres[16 / RPI];   // residual rows of this slab: issued first, consumed after the LDS bounce
#pragma unroll
        for (int i = 0; i < 16 / RPI; ++i) {
            const int m = m0 + wm * WM + 16 * b + er0 + RPI * i;
            const int n = nb + cg * VEC;
            rres[i] = make_uint4(0, 0, 0, 0);
            if (R && fast) {   // wave-uniform condition; the address is clamped instead of branching per lane
                const int mc = min(m, p.M - 1), nc = (n + VEC <= p.N) ? n : 0;
                rres[i] = *reinterpret_cast<const uint4*>(R + (long)mc * p.ldr + nc);
            }
        }
        float rs = 1.f, mrs = 0.f;       // output-side LayerNorm of this lane's row: rstd and mean * rstd
        if (p.lne_stats) {
            const int mrow = min(m0 + wm * WM + 16 * b + fr, p.M - 1);
            const float2 st = *reinterpret_cast<const float2*>(p.lne_stats + 2 * ((long)z * p.M + mrow));
            const float mu = st.x / (float)p.ln_C;
            rs = rsqrtf(fmaxf(st.y / (float)p.ln_C - mu * mu, 0.f) + p.ln_eps);
            mrs = mu * rs;
        }
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            float4 v = make_float4(fmaf(acc[a][b][0], rs, fmaf(-mrs, csum_v[a].x, bias_v[a].x)),
                                   fmaf(acc[a][b][1], rs, fmaf(-mrs, csum_v[a].y, bias_v[a].y)),
                                   fmaf(acc[a][b][2], rs, fmaf(-mrs, csum_v[a].z, bias_v[a].z)),
                                   fmaf(acc[a][b][3], rs, fmaf(-mrs, csum_v[a].w, bias_v[a].w)));
            if (act_gelu) {          // wave-uniform flags, tested once per sub-tile (not per element)
                v.x = gelu_t<T>(v.x); v.y = gelu_t<T>(v.y); v.z = gelu_t<T>(v.z); v.w = gelu_t<T>(v.w);
            } else if (act_relu) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            *reinterpret_cast<float4*>(ep + fr * EP_LD + 16 * a + 4 * fq) = v;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 16 / RPI; ++i) {
            const int row = er0 + RPI * i;
            const int m = m0 + wm * WM + 16 * b + row;
            const int n = nb + cg * VEC;
            float st1 = 0.f, st2 = 0.f;
            if (m < p.M && n < p.N) {
                float v[VEC];
#pragma unroll
                for (int j = 0; j < VEC; j += 4) {
                    const float4 t = *reinterpret_cast<const float4*>(ep + row * EP_LD + cg * VEC + j);
                    v[j] = t.x; v[j + 1] = t.y; v[j + 2] = t.z; v[j + 3] = t.w;
                }
                if (fast && n + VEC <= p.N) {
                    if (R) {
                        const T* rv = reinterpret_cast<const T*>(&rres[i]);
#pragma unroll
                        for (int j = 0; j < VEC; ++j) v[j] += to_f32<T>(rv[j]);
                    }
                    uint4 ov;
                    T* o = reinterpret_cast<T*>(&ov);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) o[j] = from_f32<T>(v[j]);
                    *reinterpret_cast<uint4*>(C + (long)m * p.ldc + n) = ov;
                    if (p.out_stats) {       // statistics of the STORED (rounded) values, like a LayerNorm that reads them back
#pragma unroll
                        for (int j = 0; j < VEC; ++j) {
                            const float q = to_f32<T>(o[j]);
                            st1 += q;
                            st2 += q * q;
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        if (n + j < p.N) {
                            float o = v[j];
                            if (R) o += to_f32<T>(R[(long)m * p.ldr + n + j]);
                            C[(long)m * p.ldc + n + j] = from_f32<T>(o);
                        }
                    }
                }
            }
            if (p.out_stats) {
                // the CG lanes that hold one row's column groups are adjacent: reduce, then one atomic pair per row and
                // 64-column wave block (every lane takes part in the shuffles, rows >= M contribute zeros)
#pragma unroll
                for (int o = CG >> 1; o > 0; o >>= 1) {
                    st1 += __shfl_xor(st1, o);
                    st2 += __shfl_xor(st2, o);
                }
                if (cg == 0 && m < p.M) {
                    atomicAdd(p.out_stats + 2 * ((long)z * p.M + m), st1);
                    atomicAdd(p.out_stats + 2 * ((long)z * p.M + m) + 1, st2);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <typename T, int BM, int BN, bool CONV, int NBUF, bool LNA = false>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs p) {
    gemm_body<T, BM, BN, CONV, NBUF, LNA>(p, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x);
}

// Two implicit-GEMM problems in ONE launch (64x64 tiles, normalising loader): the first `na` workgroups work on `a`, the
// rest on `b`.  Built for the PVT block, where the q projection (a 1x1 conv over the tokens) and the spatial-reduction conv
// read the same normalised tokens and are independent: the sr conv is 16-80 workgroups walking a long K, alone it leaves the
// chip idle for its whole duration, next to the q tiles it is hidden.  Neither problem may clear scratch the other one
// accumulates into (the statistics scratch of a stage is cleared by the stage's patch-embed conv).
// ADENSE: problem a is a plain GEMM over the raw rows with the LayerNorm on the output side (lne_stats / lne_colsum) on the
// LDS-DMA main loop -- the q projection then neither normalises its operand once per N tile nor stages it through registers.
template <typename T, int NBUF, bool ADENSE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void gemm_pair_kernel(const GemmArgs a, const GemmArgs b, const int na) {
    if ((int)blockIdx.x < na) {
        if constexpr (ADENSE) gemm_body<T, 64, 64, false, 3, false>(a, blockIdx.x, 0, 0, na);
        else gemm_body<T, 64, 64, true, NBUF, true, true>(a, blockIdx.x, 0, 0, na);
    } else {                                            // problem b may be split along K: block = split * tiles + tile
        const int idx = (int)blockIdx.x - na, tiles = b.tiles_m * b.tiles_n;
        const int split = idx / tiles;
        gemm_body<T, 64, 64, true, NBUF, true, true>(b, idx - split * tiles, split, 0, tiles);
    }
}

int g_gemm_nbuf = 1;  // debug/tuning knob (emip_debug_set)
int g_gemm_tile = 0;  // 0 = heuristic, else BM*1000+BN
int g_gemm_dbg = 0;
int g_gemm_glds = 1;  // LDS-DMA main loop for dense GEMMs
int g_pair_two = 1;   // pair launch: the sr conv walks two K tiles per barrier interval (emip_debug_set key 8)
int g_gemm_deep = 512;    // 64x64 tiles: 3-stage register prefetch for grids of at most this many workgroups (0 = off)
int g_gemm_share = 1; // concurrent streams sharing the GPU (tile choice assumes 256 / share CUs); measured: no gain

template <typename T, int BM, int BN, bool CONV>
int launch(GemmArgs& a, int batch, hipStream_t s) {
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.N + BN - 1) / BN;
    a.dbg = g_gemm_dbg;
    dim3 grid(a.tiles_m * a.tiles_n, a.ksplit > 1 ? a.ksplit : 1, batch);
    if (a.ksplit > 1 && !a.ln_stats) {       // split-K lives in the register-staged loops
        if constexpr (BM == 64 && BN == 64) {
            hipLaunchKernelGGL((gemm_kernel<T, BM, BN, CONV, 5>), grid, dim3(256), (BM + BN) * 128, s, a);
        } else {
            hipLaunchKernelGGL((gemm_kernel<T, BM, BN, CONV, 1>), grid, dim3(256), (BM + BN) * 128, s, a);
        }
        return emip_launch_status();
    }
    if (a.ln_stats) {          // normalising operand loader: the register-staged loops (the LDS-DMA path cannot touch the data)
        if constexpr (BM == 64 && BN == 64) {
            if (g_gemm_deep && (long)a.tiles_m * a.tiles_n * batch <= g_gemm_deep && a.K >= 8 * (int)(128 / sizeof(T))) {
                hipLaunchKernelGGL((gemm_kernel<T, BM, BN, CONV, 5, true>), grid, dim3(256), (BM + BN) * 128, s, a);
                return emip_launch_status();
            }
        }
        hipLaunchKernelGGL((gemm_kernel<T, BM, BN, CONV, 1, true>), grid, dim3(256), (BM + BN) * 128, s, a);
        return emip_launch_status();
    }
    if constexpr (!CONV) {
        constexpr int BKE = 128 / sizeof(T);
        if (g_gemm_glds && a.K % BKE == 0 && a.K1 % BKE == 0) {
            hipLaunchKernelGGL((gemm_kernel<T, BM, BN, CONV, 3>), grid, dim3(256), 2 * (BM + BN) * 128, s, a);
            return emip_launch_status();
        }
    }
    if constexpr (BM == 64 && BN == 64) {
        // small grids with a long K walk: deep register prefetch (see NBUF == 5)
        if (g_gemm_deep && (long)a.tiles_m * a.tiles_n * batch <= g_gemm_deep && a.K >= 8 * (int)(128 / sizeof(T))) {
            hipLaunchKernelGGL((gemm_kernel<T, BM, BN, CONV, 5>), grid, dim3(256), (BM + BN) * 128, s, a);
            return emip_launch_status();
        }
    }
    if (g_gemm_nbuf == 2)
        hipLaunchKernelGGL((gemm_kernel<T, BM, BN, CONV, 2>), grid, dim3(256), 2 * (BM + BN) * 128, s, a);
    else
        hipLaunchKernelGGL((gemm_kernel<T, BM, BN, CONV, 1>), grid, dim3(256), (BM + BN) * 128, s, a);
    return emip_launch_status();
}

// tile choice: least padded work, then the larger tile; small problems take the smaller tile so
// that the grid still covers the 256 CUs.
void pick_tile(long M, long N, long batch, int& bm, int& bn, long K = 0) {
    auto waste = [&](int tm_, int tn_) {
        const long tm = (M + tm_ - 1) / tm_, tn = (N + tn_ - 1) / tn_;
        return tm * tm_ * tn * tn_;
    };
    auto blocks = [&](int tm_, int tn_) { return ((M + tm_ - 1) / tm_) * ((N + tn_ - 1) / tn_) * batch; };
    bn = (waste(128, 64) < waste(128, 128)) ? 64 : 128;
    bm = 128;
    // short K: workgroups are latency-bound, prefer many small ones; long K (>= 1024): each workgroup streams
    // many K tiles, so keep the big tile (half the operand traffic per flop) as long as every CU gets one.
    const long min_blocks = (K >= 1024 ? 256 : 1024) / g_gemm_share;
    if (blocks(bm, bn) < min_blocks || waste(64, bn) * 10 < waste(128, bn) * 9) bm = 64;
    if (bn == 128 && blocks(bm, bn) < 256 / g_gemm_share) bn = 64;
}

template <typename T, bool CONV>
int dispatch(GemmArgs& a, int batch, hipStream_t s) {
    int bm, bn;
    pick_tile(a.M, a.N, batch, bm, bn, a.K);
    if (g_gemm_tile) { bm = g_gemm_tile / 1000; bn = g_gemm_tile % 1000; }
    if (bm == 128 && bn == 128) return launch<T, 128, 128, CONV>(a, batch, s);
    if (bm == 128 && bn == 64) return launch<T, 128, 64, CONV>(a, batch, s);
    if (bm == 64 && bn == 128) return launch<T, 64, 128, CONV>(a, batch, s);
    return launch<T, 64, 64, CONV>(a, batch, s);
}

}  // namespace

extern "C" int emip_gemm_ln(const void*, const void*, const void*, void*, const float*, const void*, int, int, int, int, long,
                            long, long, long, long, int, int, long, long, long, long, const float*, int, float, float*, void*,
                            long, int, void*);
extern "C" int emip_gemm_lne(const void*, const void*, void*, const float*, const void*, int, int, int, long, long, long, long,
                             int, const float*, const float*, float, float*, void*, long, int, void*);
extern "C" int emip_conv2d_ln(const void*, const void*, void*, const float*, const void*, int, int, int, int, long, int, int, int,
                              int, int, long, long, int, void*, long, const float*, float, float*, int, void*);
extern "C" int emip_conv2d_splitk(const void*, const void*, void*, const float*, const void*, int, int, int, int, long, int, int,
                                  int, int, int, long, long, int, void*, long, const float*, float, float*, float*, long, int,
                                  int, void*);

// the 8-wave LDS-DMA body (gemm8.hip) takes the large bf16 launches; 0 = not eligible
extern "C" int emip_gemm8(const void*, const void*, const void*, void*, const float*, const void*, int, int, int, int, long,
                          long, long, long, long, int, const float*, const float*, float, float*, void*, long, int, void*);
extern "C" int emip_conv8(const void*, const void*, void*, const float*, const void*, int, int, int, int, long, int, int, int,
                          int, int, long, long, int, const float*, const float*, float, float*, void*, long, int, void*);
namespace emip_internal {
int gemm8_choice(int M, int N, long K, long lda, long ldw, int K1, bool has_a2, long lda2);
int conv8_choice(int M, int Cout, int Cin, int KH, int KW, long a_elems);
int row_stats(const void* C, long ldc, float* out_stats, int M, int N, void* stream);   // gemm8.hip: fixed-order row sums of a bf16 matrix
}

// head strides handed from emip_gemm_heads to the shared entry body below (host-side, set and cleared around the call)
static thread_local long t_hsA = 0, t_hsW = 0, t_hsC = 0;
static thread_local int t_heads = 1;

static int check_ln(const GemmArgs& a, int batch, bool fast_epilogue) {
    if (a.ln_stats) EMIP_REQUIRE(a.ln_C > 0 && a.ln_eps > 0.f && a.A2 == nullptr && batch == 1 &&
                                 (reinterpret_cast<uintptr_t>(a.ln_stats) & 7) == 0);
    if (a.out_stats) EMIP_REQUIRE(fast_epilogue && (reinterpret_cast<uintptr_t>(a.out_stats) & 3) == 0);
    return EMIP_OK;
}

extern "C" int emip_gemm(const void* A, const void* A2, const void* W, void* C, const float* bias, const void* R,
                         int M, int N, int K, int K1, long lda, long lda2, long ldw, long ldc, long ldr, int act,
                         int batch, long bsA, long bsW, long bsC, long bsR, int dtype, void* stream) {
    return emip_gemm_ln(A, A2, W, C, bias, R, M, N, K, K1, lda, lda2, ldw, ldc, ldr, act, batch, bsA, bsW, bsC, bsR,
                        nullptr, 0, 0.f, nullptr, nullptr, 0, dtype, stream);
}

// Strided-batched GEMM with a second (head) level: batch = B * heads, operand of (b, h) at b * bs + h * hs.  The per-head
// products of the attention backward (lib/pvt_v2.py:113-121) in one launch instead of one per head.
extern "C" int emip_gemm_heads(const void* A, const void* W, void* C, int M, int N, int K, long lda, long ldw, long ldc,
                               int batch, int heads, long bsA, long hsA, long bsW, long hsW, long bsC, long hsC, int dtype,
                               void* stream) {
    EMIP_REQUIRE(heads >= 1 && batch >= heads);
    t_heads = heads; t_hsA = hsA; t_hsW = hsW; t_hsC = hsC;
    const int rc = emip_gemm_ln(A, nullptr, W, C, nullptr, nullptr, M, N, K, K, lda, 0, ldw, ldc, 0, EMIP_ACT_NONE, batch,
                                bsA, bsW, bsC, 0, nullptr, 0, 0.f, nullptr, nullptr, 0, dtype, stream);
    t_heads = 1; t_hsA = t_hsW = t_hsC = 0;
    return rc;
}

// emip_gemm plus the LayerNorm-elimination hooks: ln_stats [M][2] (sum, sum of squares over ln_C = K channels of every A
// row): the loader feeds (x - mean) * rstd; out_stats [M][2]: row sums of the stored output, accumulated with atomics
// (cleared beforehand, e.g. through an earlier launch's zero_ptr); zero_ptr / zero_bytes: scratch this launch clears.
// output-side parameters handed from emip_gemm_lne to the shared entry body (host-side, set and cleared around the call)
static thread_local const float* t_lne_stats = nullptr;
static thread_local const float* t_lne_colsum = nullptr;

// emip_gemm_ln with the LayerNorm applied on the output side (see GemmArgs::lne_stats): same result as ln_stats up to f32
// rounding, but the main loop stages the raw rows (LDS-DMA path) and the normalisation costs two FMAs per OUTPUT element.
// colsum f32 [N] = sum_k W[n][k] over the packed weights (the values the MFMA sees).
extern "C" int emip_gemm_lne(const void* A, const void* W, void* C, const float* bias, const void* R, int M, int N, int K,
                             long lda, long ldw, long ldc, long ldr, int act, const float* ln_stats, const float* colsum,
                             float ln_eps, float* out_stats, void* zero_ptr, long zero_bytes, int dtype, void* stream) {
    EMIP_REQUIRE(ln_stats && colsum && ln_eps > 0.f && (reinterpret_cast<uintptr_t>(ln_stats) & 7) == 0);
    t_lne_stats = ln_stats; t_lne_colsum = colsum;
    const int rc = emip_gemm_ln(A, nullptr, W, C, bias, R, M, N, K, K, lda, 0, ldw, ldc, ldr, act, 1, 0, 0, 0, 0, nullptr, K,
                                ln_eps, out_stats, zero_ptr, zero_bytes, dtype, stream);
    t_lne_stats = t_lne_colsum = nullptr;
    return rc;
}

extern "C" int emip_gemm_ln(const void* A, const void* A2, const void* W, void* C, const float* bias, const void* R,
                            int M, int N, int K, int K1, long lda, long lda2, long ldw, long ldc, long ldr, int act,
                            int batch, long bsA, long bsW, long bsC, long bsR, const float* ln_stats, int ln_C,
                            float ln_eps, float* out_stats, void* zero_ptr, long zero_bytes, int dtype, void* stream) {
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
    if (dtype == EMIP_BF16 && batch == 1 && t_heads == 1 && ln_stats == nullptr && aligned16(C) && (R == nullptr || aligned16(R))) {
        const int cfg = emip_internal::gemm8_choice(M, N, K, lda, ldw, K1, A2 != nullptr, lda2);
        if (cfg > 0) {
            if (zero_ptr) EMIP_REQUIRE(zero_bytes > 0 && (zero_bytes & 3) == 0);
            if (t_lne_stats) EMIP_REQUIRE(ln_eps > 0.f);
            return emip_gemm8(A, A2, W, C, bias, R, M, N, K, K1, lda, lda2, ldw, ldc, ldr, act, t_lne_stats, t_lne_colsum, ln_eps,
                              out_stats, zero_ptr, zero_bytes, cfg, stream);
        }
    }
    GemmArgs a{};
    a.A = A; a.A2 = A2; a.W = W; a.C = C; a.bias = bias; a.R = R;
    a.M = M; a.N = N; a.K = K; a.K1 = K1;
    a.lda = lda; a.lda2 = lda2; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr; a.act = act;
    a.bsA = bsA; a.bsW = bsW; a.bsC = bsC; a.bsR = bsR;
    a.heads = t_heads; a.hsA = t_hsA; a.hsW = t_hsW; a.hsC = t_hsC;
    if (t_heads > 1) EMIP_REQUIRE(A2 == nullptr && R == nullptr && batch % t_heads == 0 && t_hsA % vec == 0 && t_hsW % vec == 0);
    a.ln_stats = ln_stats; a.ln_C = ln_C; a.ln_eps = ln_eps; a.out_stats = out_stats;
    a.lne_stats = t_lne_stats; a.lne_colsum = t_lne_colsum;
    if (zero_ptr) EMIP_REQUIRE(zero_bytes > 0 && (zero_bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(zero_ptr) & 3) == 0);
    a.zero_ptr = static_cast<unsigned*>(zero_ptr);
    a.zero_words = zero_ptr ? zero_bytes / 4 : 0;
    {
        const bool fast = (ldc % vec) == 0 && aligned16(C) && N % vec == 0 &&
                          (R == nullptr || ((ldr % vec) == 0 && aligned16(R)));
        if (check_ln(a, batch, fast) != EMIP_OK) return EMIP_E_INVALID;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == EMIP_BF16 && a.out_stats && batch == 1) {
        // round 4: the 4-wave body adds a row's statistics with one f32 atomic pair per wave and column tile, in arrival order;
        // in the bf16 mode they come from a fixed-order pass over the stored rows instead (reproducible bit for bit)
        a.out_stats = nullptr;
        const int rc = dispatch<bf16_t, false>(a, batch, s);
        return rc != EMIP_OK ? rc : emip_internal::row_stats(C, ldc, out_stats, M, N, stream);
    }
    return dtype == EMIP_F32 ? dispatch<float, false>(a, batch, s) : dispatch<bf16_t, false>(a, batch, s);
}

namespace emip_internal {
extern thread_local void* t_stats_ws;
extern thread_local long t_stats_ws_bytes;
}  // namespace emip_internal

// emip_gemm_ln with a caller-owned workspace (emip_gemm_stats_ws_bytes(M, N) bytes, its ticket block zero before the first
// use) through which a bf16 launch whose rows span more than two column tiles combines its row statistics in a fixed order
// INSIDE the launch (the last column tile of a row tile to finish adds the partials) instead of a row_stats pass behind it.
// Launches that do not need it ignore it.
extern "C" int emip_gemm_ln_ws(const void* A, const void* A2, const void* W, void* C, const float* bias, const void* R,
                               int M, int N, int K, int K1, long lda, long lda2, long ldw, long ldc, long ldr, int act,
                               int batch, long bsA, long bsW, long bsC, long bsR, const float* ln_stats, int ln_C,
                               float ln_eps, float* out_stats, void* zero_ptr, long zero_bytes, int dtype, void* stats_ws,
                               long stats_ws_bytes, void* stream) {
    EMIP_REQUIRE(!stats_ws || (stats_ws_bytes > 0 && (reinterpret_cast<uintptr_t>(stats_ws) & 63u) == 0));
    emip_internal::t_stats_ws = stats_ws;
    emip_internal::t_stats_ws_bytes = stats_ws ? stats_ws_bytes : 0;
    const int rc = emip_gemm_ln(A, A2, W, C, bias, R, M, N, K, K1, lda, lda2, ldw, ldc, ldr, act, batch, bsA, bsW, bsC, bsR,
                                ln_stats, ln_C, ln_eps, out_stats, zero_ptr, zero_bytes, dtype, stream);
    emip_internal::t_stats_ws = nullptr;
    emip_internal::t_stats_ws_bytes = 0;
    return rc;
}

extern "C" int emip_conv2d(const void* X, const void* W, void* Y, const float* bias, const void* R, int B, int H,
                           int Wd, int Cin, long ldx, int Cout, int KH, int KW, int stride, int pad, long ldy,
                           long ldr, int act, void* zero_ptr, long zero_bytes, int dtype, void* stream) {
    return emip_conv2d_ln(X, W, Y, bias, R, B, H, Wd, Cin, ldx, Cout, KH, KW, stride, pad, ldy, ldr, act, zero_ptr,
                          zero_bytes, nullptr, 0.f, nullptr, dtype, stream);
}

// emip_conv2d plus the LayerNorm-elimination hooks: ln_stats [B*H*W][2] (sum, sum of squares over the Cin channels of
// every INPUT pixel): the im2col loader feeds (x - mean) * rstd (zero padding stays zero); out_stats [B*Ho*Wo][2]: row sums
// of the stored output pixels, accumulated with atomics.
extern "C" int emip_conv2d_ln(const void* X, const void* W, void* Y, const float* bias, const void* R, int B, int H,
                              int Wd, int Cin, long ldx, int Cout, int KH, int KW, int stride, int pad, long ldy,
                              long ldr, int act, void* zero_ptr, long zero_bytes, const float* ln_stats, float ln_eps,
                              float* out_stats, int dtype, void* stream) {
    return emip_conv2d_splitk(X, W, Y, bias, R, B, H, Wd, Cin, ldx, Cout, KH, KW, stride, pad, ldy, ldr, act, zero_ptr,
                              zero_bytes, ln_stats, ln_eps, out_stats, nullptr, 0, 1, dtype, stream);
}

// emip_conv2d_ln with split-K for small-M / long-K launches (the 121-token spatial-reduction convs): ksplit > 1 workgroups
// share every output tile, each walks 1/ksplit of the K tiles and adds its partial sums (bias with split 0) into
// acc_out f32 [B*Ho*Wo][ldacc] (zero beforehand) with atomics; Y, R, act and out_stats are then unused -- the cast to the
// storage type and the row statistics come from emip_rows_finalize.
extern "C" int emip_conv2d_splitk(const void* X, const void* W, void* Y, const float* bias, const void* R, int B, int H,
                                  int Wd, int Cin, long ldx, int Cout, int KH, int KW, int stride, int pad, long ldy,
                                  long ldr, int act, void* zero_ptr, long zero_bytes, const float* ln_stats, float ln_eps,
                                  float* out_stats, float* acc_out, long ldacc, int ksplit, int dtype, void* stream) {
    EMIP_REQUIRE(X && W && (Y || acc_out) && B > 0 && H > 0 && Wd > 0 && Cin > 0 && Cout > 0);
    EMIP_REQUIRE(ksplit >= 1 && ksplit <= 64);
    if (ksplit > 1) EMIP_REQUIRE(acc_out && ldacc >= Cout && R == nullptr && act == EMIP_ACT_NONE && out_stats == nullptr);
    else EMIP_REQUIRE(Y != nullptr);
    if (!Y) Y = acc_out;            // split-K: Y is not written, the checks below only look at its alignment
    if (zero_ptr) EMIP_REQUIRE(zero_bytes > 0 && (zero_bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(zero_ptr) & 3) == 0);
    EMIP_REQUIRE(dtype == EMIP_F32 || dtype == EMIP_BF16);
    EMIP_REQUIRE(KH > 0 && KW > 0 && stride > 0 && pad >= 0);
    const int vec = dtype == EMIP_F32 ? 4 : 8;
    EMIP_REQUIRE(Cin % vec == 0 && ldx % vec == 0 && ldx >= Cin && ldy >= Cout);
    EMIP_REQUIRE(aligned16(X) && aligned16(W));
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (Wd + 2 * pad - KW) / stride + 1;
    EMIP_REQUIRE(Ho > 0 && Wo > 0);
    EMIP_REQUIRE((long)B * Ho * Wo < 2147483647L && (long)KH * KW * Cin < 2147483647L && (long)H * Wd * ldx < 2147483647L);
    if (R) EMIP_REQUIRE(ldr >= Cout);
    EMIP_REQUIRE(act >= EMIP_ACT_NONE && act <= EMIP_ACT_GELU);
    if (dtype == EMIP_BF16 && ksplit == 1 && ln_stats == nullptr && aligned16(Y) && (R == nullptr || aligned16(R))) {
        const int cfg = emip_internal::conv8_choice(B * Ho * Wo, Cout, Cin, KH, KW, ((long)B * H * Wd - 1) * ldx + Cin);
        if (cfg > 0)
            return emip_conv8(X, W, Y, bias, R, B, H, Wd, Cin, ldx, Cout, KH, KW, stride, pad, ldy, ldr, act, nullptr, nullptr,
                              0.f, out_stats, zero_ptr, zero_bytes, cfg, stream);
    }
    GemmArgs a{};
    a.A = X; a.W = W; a.C = Y; a.bias = bias; a.R = R;
    a.M = B * Ho * Wo; a.N = Cout; a.K = KH * KW * Cin; a.K1 = a.K;
    a.lda = ldx; a.ldw = a.K; a.ldc = ldy; a.ldr = ldr; a.act = act;
    a.H = H; a.Wd = Wd; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
    a.zero_ptr = static_cast<unsigned*>(zero_ptr);
    a.zero_words = zero_ptr ? zero_bytes / 4 : 0;
    a.ln_stats = ln_stats; a.ln_C = Cin; a.ln_eps = ln_eps; a.out_stats = out_stats;
    a.acc_out = ksplit > 1 ? acc_out : nullptr; a.ldacc = ldacc; a.ksplit = ksplit;
    a.heads = 1;
    {
        const bool fast = (ldy % vec) == 0 && aligned16(Y) && Cout % vec == 0 &&
                          (R == nullptr || ((ldr % vec) == 0 && aligned16(R)));
        if (check_ln(a, 1, fast) != EMIP_OK) return EMIP_E_INVALID;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == EMIP_BF16 && a.out_stats) {      // as in emip_gemm_ln: the statistics by a fixed-order pass over the stored rows
        a.out_stats = nullptr;
        const int rc = dispatch<bf16_t, true>(a, 1, s);
        return rc != EMIP_OK ? rc : emip_internal::row_stats(Y, ldy, out_stats, a.M, a.N, stream);
    }
    return dtype == EMIP_F32 ? dispatch<float, true>(a, 1, s) : dispatch<bf16_t, true>(a, 1, s);
}

// Split-K with the reduction INSIDE the launch: ksplit workgroups share every output tile, each walks 1/ksplit of the K tiles
// and adds its partial tile into acc (f32 [B*Ho*Wo][Cout], ZERO on entry) with atomics; the last of them to arrive at the
// tile's ticket (u32 per 64x64 tile, ZERO on entry) takes the sums back out (leaving acc and ticket zero again) and runs the
// normal epilogue: bias, activation, storage type, row statistics.  For the launches of this network with few output tiles
// and a long K walk -- the 3x3 reductions to 32 channels in front of the decoder (create_backbone.py:199-208: 11 x 11 x 512
// -> 32 is 31 tiles walking 72 K tiles, 73 us) and the 8 x 8 / 4 x 4 spatial-reduction convs (lib/pvt_v2.py:106-108).
extern "C" int emip_conv2d_ksplit(const void* X, const void* W, void* Y, const float* bias, int B, int H, int Wd, int Cin,
                                  long ldx, int Cout, int KH, int KW, int stride, int pad, long ldy, int act,
                                  const float* ln_stats, float ln_eps, float* out_stats, float* acc, void* ticket, int ksplit,
                                  int dtype, void* stream) {
    EMIP_REQUIRE(X && W && Y && acc && ticket && B > 0 && H > 0 && Wd > 0 && Cin > 0 && Cout > 0);
    EMIP_REQUIRE(ksplit >= 2 && ksplit <= 64 && (reinterpret_cast<uintptr_t>(acc) & 3) == 0 &&
                 (reinterpret_cast<uintptr_t>(ticket) & 3) == 0);
    EMIP_REQUIRE(dtype == EMIP_F32 || dtype == EMIP_BF16);
    EMIP_REQUIRE(KH > 0 && KW > 0 && stride > 0 && pad >= 0);
    const int vec = dtype == EMIP_F32 ? 4 : 8;
    EMIP_REQUIRE(Cin % vec == 0 && ldx % vec == 0 && ldx >= Cin && ldy >= Cout);
    EMIP_REQUIRE(aligned16(X) && aligned16(W));
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (Wd + 2 * pad - KW) / stride + 1;
    EMIP_REQUIRE(Ho > 0 && Wo > 0);
    EMIP_REQUIRE((long)B * Ho * Wo < 2147483647L && (long)KH * KW * Cin < 2147483647L && (long)H * Wd * ldx < 2147483647L);
    EMIP_REQUIRE(act >= EMIP_ACT_NONE && act <= EMIP_ACT_GELU);
    GemmArgs a{};
    a.A = X; a.W = W; a.C = Y; a.bias = bias; a.R = nullptr;
    a.M = B * Ho * Wo; a.N = Cout; a.K = KH * KW * Cin; a.K1 = a.K;
    a.lda = ldx; a.ldw = a.K; a.ldc = ldy; a.ldr = 0; a.act = act;
    a.H = H; a.Wd = Wd; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
    a.ln_stats = ln_stats; a.ln_C = Cin; a.ln_eps = ln_eps; a.out_stats = out_stats;
    a.acc_out = acc; a.ldacc = Cout; a.ksplit = ksplit; a.ticket = static_cast<unsigned*>(ticket);
    a.heads = 1;
    {
        const bool fast = (ldy % vec) == 0 && aligned16(Y) && Cout % vec == 0;
        if (check_ln(a, 1, fast) != EMIP_OK) return EMIP_E_INVALID;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // the ticket protocol is written for 64 x 64 tiles (one counter per tile of that grid)
    return dtype == EMIP_F32 ? launch<float, 64, 64, true>(a, 1, s) : launch<bf16_t, 64, 64, true>(a, 1, s);
}

// ---- two convs in one launch -------------------------------------------------------------------------------------------
struct emip_conv_desc_t {      // mirrors emip_conv_desc of include/emip_hip.h
    const void* X; const void* W; void* Y; const float* bias; const void* R;
    int B, H, Wd, Cin; long ldx; int Cout, KH, KW, stride, pad; long ldy, ldr; int act;
    const float* ln_stats; float ln_eps; float* out_stats;
    float* acc; unsigned* ticket; int ksplit;
    const float* colsum;
};

static int fill_pair_args(const emip_conv_desc_t& d, int dtype, GemmArgs& a) {
    EMIP_REQUIRE(d.X && d.W && d.Y && d.ln_stats && d.B > 0 && d.H > 0 && d.Wd > 0 && d.Cin > 0 && d.Cout > 0);
    EMIP_REQUIRE(d.KH > 0 && d.KW > 0 && d.stride > 0 && d.pad == 0 && d.ln_eps > 0.f);     // the pair kernel is the NOPAD loader
    const int vec = dtype == EMIP_F32 ? 4 : 8;
    EMIP_REQUIRE(d.Cin % vec == 0 && d.ldx % vec == 0 && d.ldx >= d.Cin && d.ldy >= d.Cout && d.Cout % vec == 0);
    EMIP_REQUIRE(aligned16(d.X) && aligned16(d.W) && aligned16(d.Y) && d.ldy % vec == 0);
    EMIP_REQUIRE((reinterpret_cast<uintptr_t>(d.ln_stats) & 7) == 0);
    if (d.R) EMIP_REQUIRE(d.ldr >= d.Cout && d.ldr % vec == 0 && aligned16(d.R));
    if (d.out_stats) EMIP_REQUIRE((reinterpret_cast<uintptr_t>(d.out_stats) & 3) == 0);
    EMIP_REQUIRE(d.act >= EMIP_ACT_NONE && d.act <= EMIP_ACT_GELU);
    const int Ho = (d.H + 2 * d.pad - d.KH) / d.stride + 1, Wo = (d.Wd + 2 * d.pad - d.KW) / d.stride + 1;
    EMIP_REQUIRE(Ho > 0 && Wo > 0 && (long)d.B * Ho * Wo < 2147483647L && (long)d.KH * d.KW * d.Cin < 2147483647L &&
                 (long)d.H * d.Wd * d.ldx < 2147483647L);
    a = GemmArgs{};
    a.A = d.X; a.W = d.W; a.C = d.Y; a.bias = d.bias; a.R = d.R;
    a.M = d.B * Ho * Wo; a.N = d.Cout; a.K = d.KH * d.KW * d.Cin; a.K1 = a.K;
    a.lda = d.ldx; a.ldw = a.K; a.ldc = d.ldy; a.ldr = d.ldr; a.act = d.act;
    a.H = d.H; a.Wd = d.Wd; a.Cin = d.Cin; a.Ho = Ho; a.Wo = Wo; a.KH = d.KH; a.KW = d.KW; a.stride = d.stride; a.pad = d.pad;
    a.ln_stats = d.ln_stats; a.ln_C = d.Cin; a.ln_eps = d.ln_eps; a.out_stats = d.out_stats;
    a.ksplit = 1;
    if (d.ksplit > 1) {      // fused split-K (second problem of a pair only, checked by the caller)
        EMIP_REQUIRE(d.acc && d.ticket && d.ksplit <= 64 && (reinterpret_cast<uintptr_t>(d.acc) & 3) == 0 && d.R == nullptr);
        a.ksplit = d.ksplit; a.acc_out = d.acc; a.ldacc = d.Cout; a.ticket = d.ticket;
    }
    a.heads = 1;
    a.tiles_m = (a.M + 63) / 64;
    a.tiles_n = (a.N + 63) / 64;
    a.dbg = 0;
    if (d.colsum) {          // a 1x1 conv handed over as a dense GEMM with the output-side LayerNorm (first problem only)
        const int bk = dtype == EMIP_F32 ? 32 : 64;
        EMIP_REQUIRE(d.KH == 1 && d.KW == 1 && d.stride == 1 && d.ksplit <= 1 && d.Cin % bk == 0);
        a.lne_stats = d.ln_stats; a.lne_colsum = d.colsum; a.ln_stats = nullptr;
        a.bsA = a.bsW = a.bsC = a.bsR = 0;
    }
    return EMIP_OK;
}

// Two NHWC convs (either may be a 1x1 conv, i.e. a Linear over the tokens) with the normalising loader, in ONE launch.
extern "C" int emip_conv2d_pair(const void* da, const void* db, int dtype, void* stream) {
    EMIP_REQUIRE(da && db && (dtype == EMIP_F32 || dtype == EMIP_BF16));
    GemmArgs a, b;
    if (fill_pair_args(*static_cast<const emip_conv_desc_t*>(da), dtype, a) != EMIP_OK) return EMIP_E_INVALID;
    if (fill_pair_args(*static_cast<const emip_conv_desc_t*>(db), dtype, b) != EMIP_OK) return EMIP_E_INVALID;
    EMIP_REQUIRE(a.ksplit == 1);
    const int na = a.tiles_m * a.tiles_n, nb = b.tiles_m * b.tiles_n * b.ksplit;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    EMIP_REQUIRE(b.lne_stats == nullptr);
    if (a.lne_stats) {       // dense first problem: two LDS-DMA buffers of a 64x64 tile pair
        if (dtype == EMIP_F32)
            hipLaunchKernelGGL((gemm_pair_kernel<float, 5, true>), dim3(na + nb), dim3(256), 2 * 128 * 128, s, a, b, na);
        else if (g_pair_two)
            hipLaunchKernelGGL((gemm_pair_kernel<bf16_t, 6, true>), dim3(na + nb), dim3(256), 2 * 128 * 128, s, a, b, na);
        else
            hipLaunchKernelGGL((gemm_pair_kernel<bf16_t, 5, true>), dim3(na + nb), dim3(256), 2 * 128 * 128, s, a, b, na);
        return emip_launch_status();
    }
    if (dtype == EMIP_F32)
        hipLaunchKernelGGL((gemm_pair_kernel<float, 5, false>), dim3(na + nb), dim3(256), 128 * 128, s, a, b, na);
    else
        hipLaunchKernelGGL((gemm_pair_kernel<bf16_t, 5, false>), dim3(na + nb), dim3(256), 128 * 128, s, a, b, na);
    return emip_launch_status();
}

// the block tile the dispatcher picks for an (M, N, batch) problem, as BM*1000 + BN (introspection for bench.py)
extern "C" int emip_gemm_tile(long M, long N, long batch, long K) {
    int bm, bn;
    pick_tile(M, N, batch, bm, bn, K);
    return bm * 1000 + bn;
}

// tuning knobs for experiments (not part of the product contract): key 0 = LDS buffers of the GEMM (1|2)
#ifdef EMIP_TUNING
extern "C" int emip_debug_set(int key, int value) {
    if (key == 0 && (value == 1 || value == 2)) {
        g_gemm_nbuf = value;
        return EMIP_OK;
    }
    if (key == 1 && (value == 0 || value == 128128 || value == 128064 || value == 64128 || value == 64064)) {
        g_gemm_tile = value;
        return EMIP_OK;
    }
    if (key == 2) {
        g_gemm_dbg = value;
        return EMIP_OK;
    }
    if (key == 3) {
        g_gemm_glds = value ? 1 : 0;
        return EMIP_OK;
    }
    if (key == 6 && value >= 0) {
        g_gemm_deep = value;
        return EMIP_OK;
    }
    if (key == 8) {
        g_pair_two = value ? 1 : 0;
        return EMIP_OK;
    }
    if (key == 4 && (value == 1 || value == 2 || value == 4 || value == 8)) {
        g_gemm_share = value;      // launches are issued for one of `value` streams that share the GPU concurrently
        return EMIP_OK;
    }
    return EMIP_E_INVALID;
}
#endif
