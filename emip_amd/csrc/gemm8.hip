// 8-wave bf16 MFMA GEMM / implicit-GEMM conv for gfx950 -- the large-launch body of libemip_hip.so.
//
//   C[m, n] = act( LNout( sum_k A[m, k] W[n, k] ) + bias[n] ) + R[m, n]          (bf16 in / out, f32 accumulation)
//
// Used for every Linear / 1x1 conv / dense conv of the path whose launch is big enough to fill the chip with
// 128..256-row tiles (PVTv2 q / proj / fc1 / fc2, /root/reference/lib/pvt_v2.py:45-54,101-129; GMFlow q/k/v/merge/FFN,
// gmflow/transformer.py:128-196; conv_corr, model/EMIP_short/model.py:59-62; GMFlow CNN, gmflow/backbone.py:154-192).
//
// Structure (one workgroup = 8 waves = one BM x BN output tile, one workgroup per CU):
//   * operands go HBM/L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no staging registers, no ds_write pass.
//     K tile = 64 bf16 = one 128-B LDS row per tile row; the XOR swizzle (16-B chunk ^ row&7) is applied on the per-lane
//     SOURCE offset, the LDS image is lane-linear.  Out-of-range rows / channels / padding taps are loaded from an
//     offset beyond the buffer descriptor's extent, which the hardware returns as zeros.
//   * NST LDS stages (3 where they fit), ONE raw s_barrier per K tile and a counted s_waitcnt vmcnt(LPT): the loads of
//     tile kt+1 stay in flight across the barrier while tile kt is multiplied and tile kt+2 is issued.
//   * the weight tile is the MFMA A operand, so a lane owns 4 consecutive output channels of one row; the W rows of a
//     32-column block are PERMUTED when they are staged so that the two accumulators of the block give every lane 8
//     consecutive channels: the epilogue stores 16 B per lane straight from registers (no LDS bounce), with bias /
//     output-side LayerNorm / GELU / ReLU / residual / row statistics fused.
#include <algorithm>
#include "common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef int i32x4 __attribute__((ext_vector_type(4)));

// raw buffer descriptor (stride 0, `bytes` records, 32-bit data format): offsets >= bytes read as zero
__device__ __forceinline__ i32x4 make_rsrc(const void* ptr, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
    return i32x4{(int)(unsigned)a, (int)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}

// One LDS-DMA wave instruction: 64 lanes x 16 B from descriptor `rs` at (voff + soff) into LDS at lds_dst + 16 lane.
// Written in asm so that hipcc does not know an LDS write is pending: with the builtin it drains vmcnt(0) in front of the
// first ds_read of every K tile, which serialises the stage ring; the counted waits of the main loop order the data.
__device__ __forceinline__ void dma16(unsigned lds_dst, unsigned voff, i32x4 rs, int soff) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_dst), "v"(voff), "s"(rs), "s"(soff)
        : "memory");
}

struct G8Args {
    const bf16_t* A;      // dense: [M, lda]; conv: X [B, H, Wd, ldx = lda]
    const bf16_t* A2;     // optional second K source (columns K1..K-1 of the logical A), dense only
    const bf16_t* W;      // [N, ldw]
    bf16_t* C;            // [M, ldc]
    const float* bias;    // [N] or null
    const bf16_t* R;      // [M, ldr] or null (may alias C)
    int M, N, K, K1;
    long lda, lda2, ldw, ldc, ldr;
    int act;
    const float* lne_stats;    // [M, 2] (sum, sum of squares) of the rows of A: LayerNorm applied on the output side
    const float* lne_colsum;   // [N] column sums of W
    float lne_eps;
    float* out_stats;          // [M, 2]: accumulates (sum, sum of squares) of the stored rows
    int stats_lds;             // > 0: byte offset of [BM][WGN][2] f32 in LDS through which the WGN waves of a tile row combine
                               // their partial sums in FIXED order (one writer per row and column tile); 0: one atomic pair per
                               // wave and row -- order-independent only while a row receives at most two contributions
    float* stats_part;         // with stats_cnt: [tiles_m][tiles_n][BM][2] f32 row partials of the column tiles and
    unsigned* stats_cnt;       // [tiles_m] tickets (zero at launch, zero again at exit): the LAST column tile of a row tile to
                               // arrive adds the partials in column order and stores out_stats (rows of more than two tiles)
    const float* rowscale;     // [M / rs_rows] or null: act(...) of row m is multiplied by rowscale[m / rs_rows] BEFORE the
    int rs_rows;               // residual is added (stochastic depth: residual + scale[sample] * branch)
    const float* lno_gamma;    // LNO instances: C = R + LayerNorm_over_N(A W^T + bias) * gamma + beta, one output tile per row
    const float* lno_beta;     // block (N <= BN): post-norm Linear layers (GMFlow transformer.py:87-113: merge -> norm1, mlp -> norm2)
    float lno_eps;
    unsigned* zero_ptr;
    long zero_words;
    int H, Wd, Cin, KH, KW, stride, pad, Ho, Wo;   // conv geometry (CONV instances)
    // LNT instances (patch convs behind a folded LayerNorm, lib/pvt_v2.py:106-108): ln_stats f32 [B*H*Wd][2] = (sum, sum of
    // squares) over the Cin channels of every INPUT pixel, tapsum f32 [KH*KW][N] = sum over the channels of a tap of W.
    //   y = sum_tap rstd_tap (W_tap . x_tap - mean_tap tapsum_tap) + bias      (pad == 0: every tap lies inside the image)
    const float* ln_stats;
    const float* tapsum;
    float ln_eps;
    unsigned s_bytes;
    unsigned a_bytes, a2_bytes, w_bytes;            // descriptor extents (< 2^31)
    int tiles_m, tiles_n;
    int tpb, ntiles;        // tiles per batch entry, tiles of the launch (batch * tpb)
    long bsA, bsW, bsC;     // strided-batched dense launches: element strides between batch entries (bias shared; no R, no statistics)
#ifdef EMIP_TUNING
    int dbg;      // ablations (tuning library only): 1 = no epilogue stores, 2 = no MFMA, 4 = no operand loads, 8 = return at entry
#endif
};

#ifdef EMIP_TUNING
int g8_dbg = 0;
#define G8_DBG(p, bit) ((p).dbg & (bit))
#else
#define G8_DBG(p, bit) false
#endif

constexpr unsigned OOB = 0x80000000u;   // beyond every descriptor extent: the load returns zeros

// the 4-byte form: lane l lands at lds_dst + 4 l
__device__ __forceinline__ void dma4(unsigned lds_dst, unsigned voff, i32x4 rs, int soff) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_dst), "v"(voff), "s"(rs), "s"(soff)
        : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// RS: the per-sample scale of the epilogue (emip_gemm8_rs) is its own instance -- compiled into every instance it cost the
// inference step 1.2 % (1133 -> 1120 pairs/s in-call: one more live register per accumulator row in every epilogue)
// LNO: LayerNorm over the N columns of the output row in the epilogue (its own instances too): the waves of a tile row meet
// through 2 KB of LDS behind the ring
template <int BM, int BN, int WGM, int WGN, int NST, bool CONV, bool LNT = false, bool RS = false, bool LNO = false>
__global__ __launch_bounds__(512) void gemm8_kernel(const G8Args p) {
    constexpr int WTM = BM / WGM, WTN = BN / WGN;      // wave tile
    constexpr int TM = WTM / 16, TN = WTN / 16;        // 16x16 accumulators per wave
    constexpr int NPAIR = TN / 2;                      // 32-column blocks stored 16 B per lane
    constexpr int R = BM + BN;                         // tile rows per stage (128 B each)
    constexpr int LPR = R / 64;                        // 16-byte LDS-DMA instructions per wave per stage
    constexpr int LPT = LPR + (LNT ? 1 : 0);           // ... plus one 4-byte one for the row statistics of the tap (LNT)
    constexpr int LA = BM / 64;                        // ... of which the first LA move A rows
    constexpr int STAGE = R * 128 + (LNT ? 2048 : 0);  // LNT: [BM][2] f32 statistics behind the operand rows
    static_assert(WGM * WGN == 8 && WTM % 16 == 0 && WTN % 16 == 0 && BM % 64 == 0 && BN % 64 == 0, "tile shape");
    static_assert(NST * STAGE <= 160 * 1024 && NST >= 2 && NST <= 5 && 3 * LPT < 64, "LDS / ring depth");
    static_assert(!LNT || (CONV && BM <= 256), "per-tap LayerNorm is a conv mode");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    if (G8_DBG(p, 8)) return;                          // tuning build: the bare launch (dispatch + kernel boundary)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;

    // Persistent over output tiles: workgroup b works on tiles {round * G + remap(b)}; inside a round every XCD owns a
    // contiguous chunk of tile ids (tiles that share an A panel sit in one L2).  The stage ring runs on ACROSS tile
    // boundaries: the first K tiles of the next output tile are in flight while the current one is finished and stored.
    const int ntiles = p.ntiles;
    const int G = gridDim.x;
    const int rounds = (ntiles - (int)blockIdx.x + G - 1) / G;
    auto tile_of = [&](int i) {
        const int base = i * G;
        return base + xcd_remap(blockIdx.x, min(G, ntiles - base));
    };

    if (p.zero_ptr && blockIdx.x == 0)
        for (long i = tid; i < p.zero_words; i += 512) p.zero_ptr[i] = 0u;

    const i32x4 rsA = make_rsrc(p.A, p.a_bytes);
    const i32x4 rsA2 = make_rsrc(p.A2 ? p.A2 : p.A, p.A2 ? p.a2_bytes : p.a_bytes);
    const i32x4 rsW = make_rsrc(p.W, p.w_bytes);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)smem;

    // ---- staging plan of the tile being ISSUED: instruction j of this wave moves the 8 tile rows of group 8 j + wave ----
    const int lrow = lane >> 3, lslot = lane & 7;
    const unsigned chunk = (unsigned)(lslot ^ lrow);          // source chunk of this lane's slot (row & 7 == lrow)
    unsigned arow[LA];        // dense: byte offset of the row in A (or OOB); conv: 0 = valid output pixel, OOB = beyond M
    unsigned arow2[LA];
    int a_oy[LA], a_ox[LA], a_img[LA];
    unsigned woff[LPR - LA];
    const int sidx = wave * 64 + lane;                 // LNT: this lane moves component sidx & 1 of tile row sidx >> 1
    int s_oy = 0, s_ox = 0, s_img = 0;
    bool s_ok = false;
    const i32x4 rsS = make_rsrc(LNT ? (const void*)p.ln_stats : (const void*)p.A, LNT ? p.s_bytes : 0u);
    auto setup = [&](int t) {
        const int bz = t / p.tpb, tr = t - bz * p.tpb;           // batch entry (0 unless strided-batched), tile inside it
        const int m0 = (tr / p.tiles_n) * BM, n0 = (tr % p.tiles_n) * BN;
        const unsigned boffA = (unsigned)(bz * p.bsA * 2), boffW = (unsigned)(bz * p.bsW * 2);
        if (LNT) {
            const int m = m0 + (sidx >> 1), hw = p.Ho * p.Wo;
            const int b = m / hw, rem = m - b * hw;
            s_oy = (rem / p.Wo) * p.stride;
            s_ox = (rem % p.Wo) * p.stride;
            s_img = b * p.H;
            s_ok = (sidx >> 1) < BM && m < p.M;
        }
#pragma unroll
        for (int j = 0; j < LA; ++j) {
            const int m = m0 + 8 * (8 * j + wave) + lrow;
            if (CONV) {
                const int hw = p.Ho * p.Wo;
                const int b = m / hw, rem = m - b * hw;
                a_oy[j] = (rem / p.Wo) * p.stride - p.pad;
                a_ox[j] = (rem % p.Wo) * p.stride - p.pad;
                a_img[j] = b * p.H;
                arow[j] = m < p.M ? 0u : OOB;
                arow2[j] = 0u;
            } else {
                arow[j] = m < p.M ? boffA + (unsigned)((long)m * p.lda * 2) + chunk * 16u : OOB;
                arow2[j] = m < p.M ? (unsigned)((long)m * p.lda2 * 2) + chunk * 16u : OOB;
                a_oy[j] = a_ox[j] = a_img[j] = 0;
            }
        }
#pragma unroll
        for (int j = 0; j < LPR - LA; ++j) {
            const int pr = 8 * (8 * j + wave) + lrow;              // W tile row
            const int wv = pr / WTN, wi = pr % WTN;
            int nl = wi;
            if (wi < 32 * NPAIR) {                                  // paired 16-blocks: i -> 8 (i >> 2) + 4 half + (i & 3)
                const int i = wi & 15;
                nl = (wi & ~31) + 8 * (i >> 2) + 4 * ((wi >> 4) & 1) + (i & 3);
            }
            const int n = n0 + wv * WTN + nl;
            woff[j] = n < p.N ? boffW + (unsigned)((long)n * p.ldw * 2) + chunk * 16u : OOB;
        }
    };

    const int ctiles = CONV ? (p.Cin + 63) / 64 : 0;
    const int nk = CONV ? p.KH * p.KW * ctiles : p.K / 64;

    auto issue = [&](int kt, int stage) {
        if (G8_DBG(p, 4)) return;
        const unsigned base = lds0 + stage * STAGE + wave * 1024;
        if (CONV) {
            const int tap = kt / ctiles, c0 = (kt - tap * ctiles) * 64;
            const int ky = tap / p.KW, kx = tap - ky * p.KW;
            const bool cok = c0 + (int)chunk * 8 < p.Cin;
#pragma unroll
            for (int j = 0; j < LA; ++j) {
                const int iy = a_oy[j] + ky, ix = a_ox[j] + kx;
                const bool ok = cok && arow[j] == 0u && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.Wd;
                const unsigned off = ok ? (unsigned)(((long)(a_img[j] + iy) * p.Wd + ix) * p.lda * 2) + (c0 + chunk * 8) * 2 : OOB;
                dma16(base + j * 8192, off, rsA, 0);
            }
            const int kw0 = (tap * p.Cin + c0) * 2;
#pragma unroll
            for (int j = 0; j < LPR - LA; ++j) dma16(base + (LA + j) * 8192, cok ? woff[j] : OOB, rsW, kw0);
            if (LNT) {      // (sum, sum of squares) of the tap's input pixel of tile row sidx >> 1, component sidx & 1
                const unsigned off = s_ok ? (unsigned)((((long)(s_img + s_oy + ky) * p.Wd + s_ox + kx) * 2 + (sidx & 1)) * 4) : OOB;
                dma4(lds0 + stage * STAGE + R * 128 + wave * 256, off, rsS, 0);
            }
        } else {
            const int k0 = kt * 64;
            if (k0 < p.K1) {
#pragma unroll
                for (int j = 0; j < LA; ++j) dma16(base + j * 8192, arow[j], rsA, k0 * 2);
            } else {
#pragma unroll
                for (int j = 0; j < LA; ++j) dma16(base + j * 8192, arow2[j], rsA2, (k0 - p.K1) * 2);
            }
#pragma unroll
            for (int j = 0; j < LPR - LA; ++j) dma16(base + (LA + j) * 8192, woff[j], rsW, k0 * 2);
        }
    };
    int it_i = 0, it_k = 0;          // (tile, K tile) of the next step to issue
    auto issue_next = [&](int stage) {
        issue(it_k, stage);
        if (++it_k == nk) {
            it_k = 0;
            if (++it_i < rounds) setup(tile_of(it_i));
        }
    };

    // ---- main loop ------------------------------------------------------------------------------------------------
    f32x4 acc[TM][TN];
    f32x4 part[LNT ? TM : 1][LNT ? TN : 1];            // LNT: the running tap's partial products
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < (LNT ? TM : 1); ++a)
#pragma unroll
        for (int b = 0; b < (LNT ? TN : 1); ++b) part[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    int loff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) loff[ks] = fr * 128 + (((ks * 4 + fq) ^ (lane & 7)) * 16);
    const int a_base = wm * WTM * 128, w_base = (BM + wn * WTN) * 128;

    auto compute = [&](int stage) {
        if (G8_DBG(p, 2)) return;
        const char* sb = smem + stage * STAGE;
        constexpr int AH = TM > 4 ? 4 : TM;          // A fragments live at a time (register budget of the 128-row wave tiles)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 wf[TN];
#pragma unroll
            for (int b = 0; b < TN; ++b) wf[b] = *reinterpret_cast<const uint4*>(sb + w_base + b * 2048 + loff[ks]);
#pragma unroll
            for (int a0 = 0; a0 < TM; a0 += AH) {
                uint4 af[AH];
#pragma unroll
                for (int a = 0; a < AH; ++a) af[a] = *reinterpret_cast<const uint4*>(sb + a_base + (a0 + a) * 2048 + loff[ks]);
#pragma unroll
                for (int a = 0; a < AH; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        f32x4& d = LNT ? part[LNT ? a0 + a : 0][LNT ? b : 0] : acc[a0 + a][b];
                        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[b]),
                                                                    __builtin_bit_cast(bf16x8, af[a]), d, 0, 0, 0);
                    }
            }
        }
    };

    // LNT: the tile's slice of tapsum in LDS, [tap][BN]; such launches give every workgroup exactly one tile
    float* ts_lds = reinterpret_cast<float*>(smem + NST * STAGE);
    if (LNT) {
        const int t0 = tile_of(0), n0 = (t0 % p.tiles_n) * BN, taps = p.KH * p.KW;
        for (int i = tid; i < taps * BN; i += 512) {
            const int tp = i / BN, c = i - tp * BN;
            ts_lds[i] = n0 + c < p.N ? p.tapsum[(long)tp * p.N + n0 + c] : 0.f;
        }
        __syncthreads();
    }
    auto merge_tap = [&](int stage, int tap) {
        const float invC = 1.f / (float)p.Cin;
        const float* sp = reinterpret_cast<const float*>(smem + stage * STAGE + R * 128);
#pragma unroll
        for (int a = 0; a < (LNT ? TM : 0); ++a) {
            const float2 s2 = *reinterpret_cast<const float2*>(sp + 2 * (wm * WTM + 16 * a + fr));
            const float mu = s2.x * invC;
            const float rs = rsqrtf(fmaxf(s2.y * invC - mu * mu, 0.f) + p.ln_eps);
#pragma unroll
            for (int b = 0; b < (LNT ? TN : 0); ++b) {
                // channels of accumulator b, element j: the epilogue's map (paired blocks hold 8 q + 4 (b & 1) + j)
                const int c = wn * WTN + (b < 2 * NPAIR ? 32 * (b >> 1) + 8 * fq + 4 * (b & 1) : 16 * b + 4 * fq);
                const float4 t4 = *reinterpret_cast<const float4*>(ts_lds + tap * BN + c);
                f32x4& pp = part[LNT ? a : 0][LNT ? b : 0];
                acc[a][b][0] = fmaf(rs, fmaf(-mu, t4.x, pp[0]), acc[a][b][0]);
                acc[a][b][1] = fmaf(rs, fmaf(-mu, t4.y, pp[1]), acc[a][b][1]);
                acc[a][b][2] = fmaf(rs, fmaf(-mu, t4.z, pp[2]), acc[a][b][2]);
                acc[a][b][3] = fmaf(rs, fmaf(-mu, t4.w, pp[3]), acc[a][b][3]);
                pp = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };

    const int total = rounds * nk;
    setup(tile_of(0));
    // ring of NST stages: NST - 1 tiles are issued ahead (NST == 2: one, behind the barrier of the previous step)
#pragma unroll
    for (int i = 0; i < (NST >= 3 ? NST - 1 : 1); ++i)
        if (i < total) issue_next(i);
    int st = 0, kt = 0, ci = 0;
    for (int s = 0; s < total; ++s) {
        if (NST >= 3) {
            // the counted wait leaves the loads of the NST - 2 tiles behind tile s in flight (fewer at the tail) -- or, right
            // behind an epilogue, its stores, which only makes the wait more conservative (never less)
            const int ahead = min(NST - 2, total - 1 - s);
            if (ahead >= 3) wait_vmcnt<3 * LPT>();
            else if (ahead == 2) wait_vmcnt<2 * LPT>();
            else if (ahead == 1) wait_vmcnt<LPT>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (s + NST - 1 < total) issue_next(st == 0 ? NST - 1 : st - 1);   // stage (st - 1) mod NST: read in step s-1
            compute(st);
            if (LNT && (kt + 1) % ctiles == 0) merge_tap(st, kt / ctiles);
            st = st == NST - 1 ? 0 : st + 1;
        } else {
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < total) issue_next(st ^ 1);     // stage st^1 was read in step s-1: every wave is past it
            compute(st);
            if (LNT && (kt + 1) % ctiles == 0) merge_tap(st, kt / ctiles);
            st ^= 1;
        }
        if (++kt < nk) continue;
        kt = 0;
        const int t = tile_of(ci++);
        const int bz = t / p.tpb, tr = t - bz * p.tpb;
        const int m0 = (tr / p.tiles_n) * BM, n0 = (tr % p.tiles_n) * BN;
        bf16_t* const Cb = p.C + bz * p.bsC;

        // ---- epilogue: registers -> HBM, 16 B per lane (32-column blocks outermost: 16 live bias / colsum registers) ----
        if (G8_DBG(p, 1) && acc[0][0][0] != 123456.f) {          // tuning build: skip the stores, keep the accumulators live
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            continue;
        }
        const int nb = n0 + wn * WTN;
        const bool gelu = p.act == EMIP_ACT_GELU, relu = p.act == EMIP_ACT_RELU;
        const float invK = 1.f / (float)p.K;
        const bool vec_ok = ((p.ldc & 7) == 0) && (!p.R || (p.ldr & 7) == 0);
        float rsv[TM], mrsv[TM], st1[TM], st2[TM], rsc[TM];
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            rsv[a] = 1.f;
            mrsv[a] = st1[a] = st2[a] = 0.f;
            rsc[a] = RS ? p.rowscale[min(m0 + wm * WTM + 16 * a + fr, p.M - 1) / p.rs_rows] : 1.f;
            if (p.lne_stats) {
                const int m = min(m0 + wm * WTM + 16 * a + fr, p.M - 1);
                const float2 s2 = *reinterpret_cast<const float2*>(p.lne_stats + 2 * (long)m);
                const float mu = s2.x * invK;
                rsv[a] = rsqrtf(fmaxf(s2.y * invK - mu * mu, 0.f) + p.lne_eps);
                mrsv[a] = mu * rsv[a];
            }
        }
        float lmean[TM], lrstd[TM];
        if (LNO) {
            // pass 1: sums of (acc + bias) and of its square over this wave's columns, the 4 lanes of a row, the WGN waves
            static_assert(!LNO || (TN % 2 == 0), "LNO: paired column blocks only");
            float* red = reinterpret_cast<float*>(smem + NST * STAGE);      // [BM][WGN][2]
            float q1[TM], q2[TM];
#pragma unroll
            for (int a = 0; a < TM; ++a) q1[a] = q2[a] = 0.f;
#pragma unroll
            for (int pb = 0; pb < NPAIR; ++pb) {
                const int n = nb + 32 * pb + 8 * fq;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool in = n + j < p.N;
                    const float b = (p.bias && in) ? p.bias[n + j] : 0.f;
#pragma unroll
                    for (int a = 0; a < TM; ++a) {
                        const float x = in ? acc[a][2 * pb + (j >> 2)][j & 3] + b : 0.f;
                        q1[a] += x;
                        q2[a] = fmaf(x, x, q2[a]);
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                q1[a] += __shfl_xor(q1[a], 16);
                q2[a] += __shfl_xor(q2[a], 16);
                q1[a] += __shfl_xor(q1[a], 32);
                q2[a] += __shfl_xor(q2[a], 32);
                if (fq == 0) *reinterpret_cast<float2*>(red + ((wm * WTM + 16 * a + fr) * WGN + wn) * 2) = make_float2(q1[a], q2[a]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const float invN = 1.f / (float)p.N;
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int w = 0; w < WGN; ++w) {
                    const float2 t = *reinterpret_cast<const float2*>(red + ((wm * WTM + 16 * a + fr) * WGN + w) * 2);
                    t1 += t.x;
                    t2 += t.y;
                }
                lmean[a] = t1 * invN;
                lrstd[a] = rsqrtf(fmaxf(t2 * invN - lmean[a] * lmean[a], 0.f) + p.lno_eps);
            }
        }
#pragma unroll
        for (int pb = 0; pb < NPAIR; ++pb) {
            const int n = nb + 32 * pb + 8 * fq;
            float bv[8], cs[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int nc = min(n + j, p.N - 1);
                bv[j] = p.bias ? p.bias[nc] : 0.f;
                cs[j] = p.lne_stats ? p.lne_colsum[nc] : 0.f;
            }
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int m = m0 + wm * WTM + 16 * a + fr;
                float v[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = acc[a][2 * pb][j];
                    v[4 + j] = acc[a][2 * pb + 1][j];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float x = fmaf(v[j], rsv[a], fmaf(-mrsv[a], cs[j], bv[j]));
                    if (gelu) x = gelu_t<bf16_t>(x);
                    else if (relu) x = fmaxf(x, 0.f);
                    v[j] = x;
                }
                if (LNO) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int nc = min(n + j, p.N - 1);
                        v[j] = (v[j] - lmean[a]) * lrstd[a] * p.lno_gamma[nc] + p.lno_beta[nc];
                    }
                }
                if (RS) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] *= rsc[a];
                }
                if (m < p.M && n < p.N) {
                    bf16_t* cp = Cb + (long)m * p.ldc + n;
                    if (vec_ok && n + 8 <= p.N) {
                        if (p.R) {
                            const uint4 rr = *reinterpret_cast<const uint4*>(p.R + (long)m * p.ldr + n);
                            const bf16_t* rv = reinterpret_cast<const bf16_t*>(&rr);
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] += (float)rv[j];
                        }
                        uint4 ov;
                        bf16_t* o = reinterpret_cast<bf16_t*>(&ov);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            o[j] = (bf16_t)v[j];
                            const float q = (float)o[j];
                            st1[a] += q;
                            st2[a] += q * q;
                        }
                        *reinterpret_cast<uint4*>(cp) = ov;
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            if (n + j < p.N) {
                                float x = v[j];
                                if (p.R) x += (float)p.R[(long)m * p.ldr + n + j];
                                const bf16_t o = (bf16_t)x;
                                cp[j] = o;
                                st1[a] += (float)o;
                                st2[a] += (float)o * (float)o;
                            }
                        }
                    }
                }
            }
        }
        if (TN & 1) {          // unpaired last 16-block: 4 channels (8 B) per lane
            const int n = nb + 32 * NPAIR + 4 * fq;
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int m = m0 + wm * WTM + 16 * a + fr;
                if (n < p.N && m < p.M) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (n + j < p.N) {
                            float x = fmaf(acc[a][TN - 1][j], rsv[a], fmaf(-mrsv[a], p.lne_stats ? p.lne_colsum[n + j] : 0.f,
                                                                            p.bias ? p.bias[n + j] : 0.f));
                            if (gelu) x = gelu_t<bf16_t>(x);
                            else if (relu) x = fmaxf(x, 0.f);
                            if (RS) x *= rsc[a];
                            if (p.R) x += (float)p.R[(long)m * p.ldr + n + j];
                            const bf16_t o = (bf16_t)x;
                            Cb[(long)m * p.ldc + n + j] = o;
                            st1[a] += (float)o;
                            st2[a] += (float)o * (float)o;
                        }
                    }
                }
            }
        }
        if (p.out_stats) {      // the 4 lanes fr, fr+16, fr+32, fr+48 hold the same row
            // Reproducibility (round 4): f32 atomics from the WGN waves (and the column tiles) of a row arrive in any order, the
            // sums differ in the last bit from run to run, and a bf16 rounding of a normalised value flips now and then
            // (round 3: two runs of the same batch 0.3 apart on logits spanning 8).  With stats_lds the waves of a tile row
            // meet through LDS and ONE lane adds their partial sums in wave order; a single column tile stores the result, two
            // column tiles add theirs atomically (two addends commute).  The launcher sends rows of more than two column
            // tiles through a separate statistics pass (emip_internal::row_stats).
            float* red = reinterpret_cast<float*>(smem + p.stats_lds);      // [BM][WGN][2]
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int m = m0 + wm * WTM + 16 * a + fr;
                float s1 = st1[a], s2 = st2[a];
                s1 += __shfl_xor(s1, 16);
                s2 += __shfl_xor(s2, 16);
                s1 += __shfl_xor(s1, 32);
                s2 += __shfl_xor(s2, 32);
                if (p.stats_lds) {
                    if (fq == 0) *reinterpret_cast<float2*>(red + ((wm * WTM + 16 * a + fr) * WGN + wn) * 2) = make_float2(s1, s2);
                } else if (fq == 0 && m < p.M) {
                    atomicAdd(p.out_stats + 2 * (long)m, s1);
                    atomicAdd(p.out_stats + 2 * (long)m + 1, s2);
                }
            }
            if (p.stats_lds && p.stats_cnt) {
                // rows of more than two column tiles: no atomics on the sums (three addends do not commute).  Every column
                // tile leaves its row partials in memory and draws a ticket; the tile that draws the last one adds the
                // partials in column order.  No waiting on other workgroups anywhere: whoever comes last does the work.
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                const int tm = t / p.tiles_n;
                if (wn == 0 && fq == 0) {
#pragma unroll
                    for (int a = 0; a < TM; ++a) {
                        const int r = wm * WTM + 16 * a + fr;
                        float t1 = 0.f, t2 = 0.f;
#pragma unroll
                        for (int w = 0; w < WGN; ++w) {
                            const float2 v = *reinterpret_cast<const float2*>(red + (r * WGN + w) * 2);
                            t1 += v.x;
                            t2 += v.y;
                        }
                        // device-scope relaxed atomics, NOT plain stores behind a __threadfence(): on this multi-XCD part an
                        // agent-scope fence writes back and invalidates the whole L2 of the XCD (measured: the step went from
                        // 2230 to 1310 pairs/s with four steps in flight); a device-scope access goes past the L2 by itself
                        __hip_atomic_store(reinterpret_cast<unsigned long long*>(p.stats_part + ((long)t * BM + r) * 2),
                                           ((unsigned long long)__float_as_uint(t2) << 32) | __float_as_uint(t1), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the partials have been performed ...
                __syncthreads();                                   // ... for every row of the tile, and red is free
                unsigned* tick = reinterpret_cast<unsigned*>(red);
                if (tid == 0) *tick = atomicAdd(p.stats_cnt + tm, 1u);
                __syncthreads();
                if (*tick == (unsigned)(p.tiles_n - 1)) {
                    if (tid < BM && m0 + tid < p.M) {
                        float t1 = 0.f, t2 = 0.f;
                        for (int c = 0; c < p.tiles_n; ++c) {
                            const unsigned long long v = __hip_atomic_load(
                                reinterpret_cast<const unsigned long long*>(p.stats_part + (((long)tm * p.tiles_n + c) * BM + tid) * 2),
                                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            t1 += __uint_as_float((unsigned)v);
                            t2 += __uint_as_float((unsigned)(v >> 32));
                        }
                        *reinterpret_cast<float2*>(p.out_stats + 2 * (long)(m0 + tid)) = make_float2(t1, t2);
                    }
                    // the next launch finds the ticket at zero again
                    if (tid == 0) __hip_atomic_store(p.stats_cnt + tm, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
            } else if (p.stats_lds) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (wn == 0 && fq == 0) {
#pragma unroll
                    for (int a = 0; a < TM; ++a) {
                        const int m = m0 + wm * WTM + 16 * a + fr;
                        float t1 = 0.f, t2 = 0.f;
#pragma unroll
                        for (int w = 0; w < WGN; ++w) {
                            const float2 t = *reinterpret_cast<const float2*>(red + ((wm * WTM + 16 * a + fr) * WGN + w) * 2);
                            t1 += t.x;
                            t2 += t.y;
                        }
                        if (m < p.M) {
                            if (p.tiles_n == 1) {
                                *reinterpret_cast<float2*>(p.out_stats + 2 * (long)m) = make_float2(t1, t2);
                            } else {
                                atomicAdd(p.out_stats + 2 * (long)m, t1);
                                atomicAdd(p.out_stats + 2 * (long)m + 1, t2);
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }   // steps
}

// see emip_internal::row_stats below: row r = 32 blockIdx + tid / 8, lane j of its eight takes the 8-element chunks j, j + 8, ..
__global__ __launch_bounds__(256) void row_stats_kernel(const bf16_t* __restrict__ C, long ldc, float* __restrict__ out, int M, int N) {
    const int row = blockIdx.x * 32 + (threadIdx.x >> 3), gl = threadIdx.x & 7;
    const bf16_t* cp = C + (long)min(row, M - 1) * ldc;
    const bool vec = ((ldc & 7) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15u) == 0);
    float s1 = 0.f, s2 = 0.f;
    for (int c0 = 8 * gl; c0 < N; c0 += 64) {
        if (vec && c0 + 8 <= N) {
            const uint4 v = *reinterpret_cast<const uint4*>(cp + c0);
            const unsigned vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = __uint_as_float(vw[j] << 16), b = __uint_as_float(vw[j] & 0xFFFF0000u);
                s1 += a + b;
                s2 = fmaf(a, a, fmaf(b, b, s2));
            }
        } else {
            for (int j = 0; j < 8 && c0 + j < N; ++j) {
                const float a = (float)cp[c0 + j];
                s1 += a;
                s2 = fmaf(a, a, s2);
            }
        }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if (gl == 0 && row < M) *reinterpret_cast<float2*>(out + 2 * (long)row) = make_float2(s1, s2);
}

}  // namespace
namespace emip_internal {
// set by emip_gemm_ln_ws (gemm.hip) around ITS launch: a caller-owned workspace for the in-launch combine of row statistics
thread_local void* t_stats_ws = nullptr;
thread_local long t_stats_ws_bytes = 0;
}  // namespace emip_internal
namespace {

struct Cfg {
    int bm, bn, nst, nst_lnt;
    void (*dense)(const G8Args);
    void (*conv)(const G8Args);
    void (*lnt)(const G8Args);      // conv with the per-tap LayerNorm (small tiles only)
    void (*rs)(const G8Args);       // dense with the per-sample epilogue scale (emip_gemm8_rs)
    void (*lno)(const G8Args);      // dense with the LayerNorm over the output row (emip_gemm8_lno; BN = 128 tiles only)
};

#define G8_CFG(BM, BN, WGM, WGN, NST)                                                                            \
    {BM, BN, NST, 0, gemm8_kernel<BM, BN, WGM, WGN, NST, false>, gemm8_kernel<BM, BN, WGM, WGN, NST, true>, nullptr, \
     gemm8_kernel<BM, BN, WGM, WGN, NST, false, false, true>, nullptr}
#define G8_CFG_LNO(BM, BN, WGM, WGN, NST)                                                                        \
    {BM, BN, NST, 0, gemm8_kernel<BM, BN, WGM, WGN, NST, false>, gemm8_kernel<BM, BN, WGM, WGN, NST, true>, nullptr, \
     gemm8_kernel<BM, BN, WGM, WGN, NST, false, false, true>, gemm8_kernel<BM, BN, WGM, WGN, NST, false, false, false, true>}
// the per-tap LayerNorm instances serve 31..183 workgroups walking 20..64 K tiles each: a 5-deep ring (4 tiles in flight)
constexpr int NST_LNT = 5;
#define G8_CFG_LNT(BM, BN, WGM, WGN, NST)                                                             \
    {BM, BN, NST, NST_LNT, gemm8_kernel<BM, BN, WGM, WGN, NST, false>, gemm8_kernel<BM, BN, WGM, WGN, NST, true>, \
     gemm8_kernel<BM, BN, WGM, WGN, NST_LNT, true, true>, gemm8_kernel<BM, BN, WGM, WGN, NST, false, false, true>, \
     gemm8_kernel<BM, BN, WGM, WGN, NST, false, false, false, true>}
// N = 320 in ONE tile for the per-tap LayerNorm conv of the 22 x 22 stage: the token panel is read once instead of once per
// 128-column tile, 61 workgroups instead of 183 at 32 images; 3 ring slots of 50 KB
#define G8_CFG_LNTW(BM, BN, WGM, WGN, NST, NSTL)                                                          \
    {BM, BN, NST, NSTL, gemm8_kernel<BM, BN, WGM, WGN, NST, false>, gemm8_kernel<BM, BN, WGM, WGN, NST, true>, \
     gemm8_kernel<BM, BN, WGM, WGN, NSTL, true, true>, gemm8_kernel<BM, BN, WGM, WGN, NST, false, false, true>, nullptr}

const Cfg g_cfg[] = {
    G8_CFG(256, 128, 4, 2, 3),       // 1: wave 64 x 64, 144 KB
    G8_CFG(128, 256, 2, 4, 3),       // 2: wave 64 x 64, 144 KB
    G8_CFG_LNO(128, 128, 2, 4, 2),   // 3: wave 64 x 32, 64 KB (2 workgroups per CU)
    G8_CFG(128, 320, 2, 4, 2),       // 4: wave 64 x 80 (N = 320 in one tile), 112 KB
    G8_CFG(64, 320, 2, 4, 3),        // 5: wave 32 x 80, 144 KB
    G8_CFG(256, 64, 4, 2, 2),        // 6: wave 64 x 32, 80 KB (2 per CU)
    G8_CFG(256, 256, 2, 4, 2),       // 7: wave 128 x 64, 128 KB
    G8_CFG_LNT(128, 64, 4, 2, 3),    // 8: wave 32 x 32, 72 KB (2 per CU)
    G8_CFG_LNT(64, 128, 2, 4, 3),    // 9: wave 32 x 32, 72 KB (2 per CU)
    G8_CFG_LNTW(64, 320, 2, 4, 3, 3),  // 10: wave 32 x 80, 144 KB; per-tap LayerNorm 155 KB
    G8_CFG_LNTW(64, 192, 2, 4, 3, 4),  // 11: wave 32 x 48; N = 320 in two tiles (192 + 128), per-tap LayerNorm ring 4 deep (139 KB)
};
constexpr int NCFG = sizeof(g_cfg) / sizeof(g_cfg[0]);

// Tile choice, from the per-shape sweep of tools/gemm8_bench.py on MI355X (profiles/r02_gemm8_sweep.txt):
//   * long K x wide N (conv_corr, big GEMMs): 256 x 256, the only tile whose L2 -> LDS bytes per MAC keep the MFMA fed;
//   * fewer than ~0.75 tiles of 128 x 128 per CU (round 2c; first ~1.5): 64 x 128 / 128 x 64 (two workgroups per CU overlap their fill / drain);
//   * N = 64 (mod 128): 256 x 64; else 128 x 128 with two workgroups per CU.
int pick_cfg(int M, int N, long K) {
    if ((double)K * N >= 4.0e6 && M >= 8192) return 7;
    const long tiles3 = (long)((M + 127) / 128) * ((N + 127) / 128);
    if (tiles3 < 192) return N <= 64 ? 8 : 9;   // round 2c: was 384 (8-pair sub-batches: the 242-tile GEMMs prefer 128 x 128)
    // the training step's 64-image shapes (tools/gemm8_train_sweep.py; the inference shapes end at M = 15 488 for N = 320):
    // 30 976 x 320 x 1280 (fc2 forward, fc1 input gradient): N = 320 in one 128 x 320 tile 36.5 us, 256 x 64 42.0;
    // 30 976 x 320 x 320 (q / proj forward and input gradient): 128 x 128 16.3 us, 256 x 64 17.6
    if (N == 320 && M >= 24576) return K >= 1024 ? 4 : 3;
    if (N == 320 && K >= 1024) return 3;      // 15 488 x 320 x 1280 (fc2 at 32 images): 128 x 128 27.1 us, 256 x 64 30.6
    if (N % 128 == 64 || N <= 64) return 6;
    return 3;
}

int launch(const G8Args& a0, int cfg, bool conv, hipStream_t s) {
    G8Args a = a0;
#ifdef EMIP_TUNING
    a.dbg = g8_dbg;
#endif
    const bool lnt = a.ln_stats != nullptr;
    if (cfg <= 0 || cfg > NCFG) cfg = pick_cfg(a.M, a.N, conv ? (long)a.KH * a.KW * a.Cin : a.K);
    if (lnt && !g_cfg[cfg - 1].lnt) cfg = a.N <= 64 ? 8 : 9;
    const Cfg& g = g_cfg[cfg - 1];
    a.tiles_m = (a.M + g.bm - 1) / g.bm;
    a.tiles_n = (a.N + g.bn - 1) / g.bn;
    const int nbatch = a.ntiles > 0 ? a.ntiles : 1;      // (the batched entry point passes its batch count here)
    a.tpb = a.tiles_m * a.tiles_n;
    a.ntiles = a.tpb * nbatch;
    if (nbatch > 1 && (conv || a.R || a.out_stats || a.rowscale || a.lno_gamma || a.A2)) return EMIP_E_INVALID;
    size_t lds = (size_t)g.nst * (g.bm + g.bn) * 128;
    if (lnt) lds = (size_t)g.nst_lnt * ((g.bm + g.bn) * 128 + 2048) + (size_t)a.KH * a.KW * g.bn * 4;
    if (lds > 160 * 1024) return EMIP_E_INVALID;
    const bool rs = a.rowscale != nullptr;
    if (rs && (lnt || conv)) return EMIP_E_INVALID;
    const bool lno = a.lno_gamma != nullptr;
    if (lno && (rs || lnt || conv || !g.lno || a.tiles_n != 1 || a.act != EMIP_ACT_NONE)) return EMIP_E_INVALID;
    if (lno) lds += (size_t)g.bm * 8 * 4;                      // [BM][WGN <= 4][2] f32 behind the ring
    // row statistics in a reproducible order (see the epilogue).  WGN = 2 tiles with one column tile: two addends per row,
    // the atomics commute.  More than two column tiles: the GEMM without statistics, then a statistics pass over its output.
    float* late_stats = nullptr;
    if (a.out_stats) {
        const int wgn = (g.bm == 256 && g.bn <= 128) || (g.bm == 128 && g.bn == 64) ? 2 : 4;       // configurations 1, 6, 8: 4 x 2 waves
        // workspace for the in-launch combine: a ticket block of FIXED size (launches of any shape may use one workspace in
        // turn: a block sized by M let one launch's partials overwrite another's tickets -- NaNs in stage 3, round 4), then
        // the partials
        constexpr size_t cnt_bytes = 4 * 4096;
        const size_t ws_need = cnt_bytes + (size_t)a.tiles_m * a.tiles_n * g.bm * 8;
        void* ws = emip_internal::t_stats_ws;
        const bool ws_ok = ws && a.tiles_m <= 4096 && (size_t)emip_internal::t_stats_ws_bytes >= ws_need && !lno &&
                           lds + (size_t)g.bm * wgn * 8 <= 160 * 1024;
        if (a.tiles_n > 2 && ws_ok) {
            a.stats_cnt = (unsigned*)ws;
            a.stats_part = (float*)((char*)ws + cnt_bytes);
            a.stats_lds = (int)lds;
            lds += (size_t)g.bm * wgn * 8;
        } else if (a.tiles_n > 2) {
            late_stats = a.out_stats;
            a.out_stats = nullptr;
        } else if (wgn * a.tiles_n > 2) {
            if (lno) return EMIP_E_INVALID;
            a.stats_lds = (int)lds;
            lds += (size_t)g.bm * wgn * 8;
            if (lds > 160 * 1024) return EMIP_E_INVALID;
        }
    }
    auto fn = lno ? g.lno : (rs ? g.rs : (lnt ? g.lnt : (conv ? g.conv : g.dense)));
    const int which = lno ? 4 : (rs ? 3 : (lnt ? 2 : (conv ? 1 : 0)));
    static size_t attr_done[NCFG][5];
    if (attr_done[cfg - 1][which] < lds) {
        if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return EMIP_E_LAUNCH;
        attr_done[cfg - 1][which] = lds;
    }
    // persistent grid: one workgroup per CU, two where LDS (<= 80 KB) and registers (<= 128, true of those tiles) allow
    const int per_cu = lds <= 80 * 1024 ? 2 : 1;
    const int tiles = a.ntiles;
    const int grid = tiles < 256 * per_cu ? tiles : 256 * per_cu;
    if (lnt && grid != tiles) return EMIP_E_INVALID;          // the tapsum slice in LDS belongs to ONE tile
    hipLaunchKernelGGL(fn, dim3(grid), dim3(512), lds, s, a);
    if (late_stats) hipLaunchKernelGGL(row_stats_kernel, dim3((unsigned)((a.M + 31) / 32)), dim3(256), 0, s, a.C, a.ldc, late_stats, a.M, a.N);
    return emip_launch_status();
}

}  // namespace

// (sum, sum of squares) of the N stored values of every row of a bf16 matrix, eight lanes per row, fixed order: the reproducible
// stand-in for epilogue statistics where a row is spread over more column tiles than atomics can add in a fixed order
namespace emip_internal {
int row_stats(const void* C, long ldc, float* out_stats, int M, int N, void* stream) {
    EMIP_REQUIRE(C && out_stats && M > 0 && N > 0 && ldc >= N);
    hipLaunchKernelGGL(row_stats_kernel, dim3((unsigned)((M + 31) / 32)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)C, ldc,
                       out_stats, M, N);
    return emip_launch_status();
}
}  // namespace emip_internal

extern "C" int emip_gemm8_auto_cfg(int M, int N, int K) { return pick_cfg(M, N, K); }
// bytes of the workspace emip_gemm_ln_ws wants for an M x N output, whatever tile it picks: 16 KB of tickets (ZERO at the first
// launch; launches leave them zero) + row partials of every column tile
extern "C" long emip_gemm_stats_ws_bytes(int M, int N) {
    const long tn = (N + 63) / 64;
    return 4 * 4096 + ((long)M + 256) * tn * 8;
}
#ifdef EMIP_TUNING
extern "C" int emip_tuning_gemm8_dbg(int v) { g8_dbg = v; return 0; }
#endif

// introspection for bench.py: the tile (BM * 1000 + BN) and ring depth of a configuration, 0 for an unknown one
extern "C" int emip_gemm8_cfg_tile(int cfg) { return cfg >= 1 && cfg <= NCFG ? g_cfg[cfg - 1].bm * 1000 + g_cfg[cfg - 1].bn : 0; }
extern "C" int emip_gemm8_cfg_stages(int cfg, int lnt) {
    return cfg >= 1 && cfg <= NCFG ? (lnt ? g_cfg[cfg - 1].nst_lnt : g_cfg[cfg - 1].nst) : 0;
}

// ---- dispatch hooks of emip_gemm_ln / emip_conv2d_splitk (gemm.hip): > 0 = not eligible, the 4-wave body runs -----------
namespace emip_internal {
int gemm8_choice(int M, int N, long K, long lda, long ldw, int K1, bool has_a2, long lda2) {
    if (M < 1024 || N < 64 || (K % 64) != 0 || (lda % 8) != 0 || (ldw % 8) != 0) return 0;
    if (has_a2 && ((K1 % 64) != 0 || (lda2 % 8) != 0)) return 0;
    if (((long)(M - 1) * lda + K) * 2 >= (1L << 31) || ((long)(N - 1) * ldw + K) * 2 >= (1L << 31)) return 0;
    return pick_cfg(M, N, K);
}
int conv8_choice(int M, int Cout, int Cin, int KH, int KW, long a_elems) {
    // (round 4: 1024, was 2048 -- with the deep PVT stages on one frame the 11 x 11-stage convolutions have 1 936 rows)
    if (M < 1024 || Cout < 64 || Cin < 64 || (Cin % 8) != 0 || a_elems * 2 >= (1L << 31) ||
        (long)Cout * KH * KW * Cin * 2 >= (1L << 31))
        return 0;
    return pick_cfg(M, Cout, (long)KH * KW * Cin);
}
}  // namespace emip_internal

// introspection for bench.py: the configuration emip_gemm / emip_gemm_ln(e) / emip_conv2d hand a bf16 launch to (0 = the
// 4-wave body of gemm.hip runs it)
extern "C" int emip_gemm8_dispatch(int M, int N, int K, long lda, long ldw, int K1, int has_a2, long lda2) {
    return emip_internal::gemm8_choice(M, N, K, lda, ldw, K1, has_a2 != 0, lda2);
}
extern "C" int emip_conv8_dispatch(int M, int Cout, int Cin, int KH, int KW, long a_elems) {
    return emip_internal::conv8_choice(M, Cout, Cin, KH, KW, a_elems);
}

// strided-batched dense GEMM on the 8-wave body: C[b] = act(A[b] W[b]^T + bias), b < batch, element strides bsA / bsW / bsC (0 = shared)
extern "C" int emip_gemm8_batched(const void* A, const void* W, void* C, const float* bias, int M, int N, int K, long lda, long ldw,
                                  long ldc, int act, int batch, long bsA, long bsW, long bsC, int cfg, void* stream) {
    EMIP_REQUIRE(A && W && C && M > 0 && N > 0 && K > 0 && (K % 64) == 0 && batch > 0 && bsA >= 0 && bsW >= 0 && bsC >= 0);
    EMIP_REQUIRE(aligned16(A) && aligned16(W) && aligned16(C) && (lda % 8) == 0 && (ldw % 8) == 0 && lda >= K && ldw >= K && ldc >= N &&
                 (bsA % 8) == 0 && (bsW % 8) == 0 && (bsC % 8) == 0);
    const long ab = ((long)(batch - 1) * bsA + (long)(M - 1) * lda + K) * 2, wb = ((long)(batch - 1) * bsW + (long)(N - 1) * ldw + K) * 2;
    EMIP_REQUIRE(ab < (1L << 31) && wb < (1L << 31));
    G8Args a = {};
    a.A = (const bf16_t*)A; a.W = (const bf16_t*)W; a.C = (bf16_t*)C; a.bias = bias; a.M = M; a.N = N; a.K = K; a.K1 = K; a.lda = lda;
    a.ldw = ldw; a.ldc = ldc; a.act = act; a.a_bytes = (unsigned)ab; a.a2_bytes = 0; a.w_bytes = (unsigned)wb;
    a.bsA = bsA; a.bsW = bsW; a.bsC = bsC; a.ntiles = batch;
    // (the tile choice sees the whole batch's rows: 16 x 1 936 rows fill the chip with 128 x 128 tiles, one image would not)
    if (cfg <= 0) cfg = pick_cfg((int)std::min<long>((long)M * batch, 1L << 30), N, K);
    return launch(a, cfg, false, (hipStream_t)stream);
}

extern "C" int emip_gemm8(const void* A, const void* A2, const void* W, void* C, const float* bias, const void* R, int M,
                          int N, int K, int K1, long lda, long lda2, long ldw, long ldc, long ldr, int act,
                          const float* lne_stats, const float* lne_colsum, float lne_eps, float* out_stats, void* zero_ptr,
                          long zero_bytes, int cfg, void* stream) {
    EMIP_REQUIRE(A && W && C && M > 0 && N > 0 && K > 0 && (K % 64) == 0);
    EMIP_REQUIRE(aligned16(A) && aligned16(W) && aligned16(C) && (lda % 8) == 0 && (ldw % 8) == 0 && lda >= (A2 ? K1 : K) &&
                 ldw >= K && ldc >= N);
    EMIP_REQUIRE(!R || (aligned16(R) && ldr >= N));
    EMIP_REQUIRE(!A2 || (aligned16(A2) && K1 > 0 && K1 < K && (K1 % 64) == 0 && (lda2 % 8) == 0 && lda2 >= K - K1));
    EMIP_REQUIRE(!lne_stats || lne_colsum);
    EMIP_REQUIRE((zero_bytes % 4) == 0 && (!zero_ptr || (reinterpret_cast<uintptr_t>(zero_ptr) & 3u) == 0));
    const long ab = ((long)(M - 1) * lda + (A2 ? K1 : K)) * 2, wb = ((long)(N - 1) * ldw + K) * 2;
    const long a2b = A2 ? ((long)(M - 1) * lda2 + (K - K1)) * 2 : 0;
    EMIP_REQUIRE(ab < (1L << 31) && wb < (1L << 31) && a2b < (1L << 31));
    G8Args a = {};
    a.A = (const bf16_t*)A; a.A2 = (const bf16_t*)A2; a.W = (const bf16_t*)W; a.C = (bf16_t*)C; a.bias = bias;
    a.R = (const bf16_t*)R; a.M = M; a.N = N; a.K = K; a.K1 = A2 ? K1 : K; a.lda = lda; a.lda2 = A2 ? lda2 : 0; a.ldw = ldw;
    a.ldc = ldc; a.ldr = ldr; a.act = act; a.lne_stats = lne_stats; a.lne_colsum = lne_colsum; a.lne_eps = lne_eps;
    a.out_stats = out_stats; a.zero_ptr = (unsigned*)zero_ptr; a.zero_words = zero_ptr ? zero_bytes / 4 : 0;
    a.a_bytes = (unsigned)ab; a.a2_bytes = (unsigned)a2b; a.w_bytes = (unsigned)wb;
    return launch(a, cfg, false, (hipStream_t)stream);
}

// C = R + LayerNorm(A W^T + bias) * gamma + beta, the normalisation over the N <= 128 output columns of a row in the epilogue
// (post-norm Linear layers: GMFlow transformer.py:87-113 merge -> norm1 and mlp -> norm2, the residual of :113 as R)
extern "C" int emip_gemm8_lno(const void* A, const void* W, void* C, const float* bias, const void* R, const float* gamma,
                              const float* beta, float eps, int M, int N, int K, long lda, long ldw, long ldc, long ldr,
                              void* stream) {
    EMIP_REQUIRE(A && W && C && gamma && beta && M > 0 && N > 0 && N <= 128 && (N % 8) == 0 && K > 0 && (K % 64) == 0);
    EMIP_REQUIRE(aligned16(A) && aligned16(W) && aligned16(C) && (lda % 8) == 0 && (ldw % 8) == 0 && lda >= K && ldw >= K &&
                 ldc >= N && (ldc % 8) == 0);
    EMIP_REQUIRE(!R || (aligned16(R) && ldr >= N && (ldr % 8) == 0));
    const long ab = ((long)(M - 1) * lda + K) * 2, wb = ((long)(N - 1) * ldw + K) * 2;
    EMIP_REQUIRE(ab < (1L << 31) && wb < (1L << 31));
    G8Args a = {};
    a.A = (const bf16_t*)A; a.W = (const bf16_t*)W; a.C = (bf16_t*)C; a.bias = bias; a.R = (const bf16_t*)R;
    a.M = M; a.N = N; a.K = K; a.K1 = K; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr; a.act = EMIP_ACT_NONE;
    a.lno_gamma = gamma; a.lno_beta = beta; a.lno_eps = eps;
    a.a_bytes = (unsigned)ab; a.w_bytes = (unsigned)wb;
    // 64 x 128 tiles unless there are more than two 128-row tiles per CU
    return launch(a, (long)((M + 127) / 128) >= 512 ? 3 : 9, false, (hipStream_t)stream);
}

// emip_gemm8 with a per-sample scale on the branch: C = R + rowscale[m / rs_rows] * act(A W^T + bias) -- stochastic depth
// (timm DropPath as used by lib/pvt_v2.py:167-169) applied in the epilogue of the proj / fc2 GEMM of a training forward
extern "C" int emip_gemm8_rs(const void* A, const void* A2, const void* W, void* C, const float* bias, const void* R, int M,
                          int N, int K, int K1, long lda, long lda2, long ldw, long ldc, long ldr, int act,
                          const float* lne_stats, const float* lne_colsum, float lne_eps, float* out_stats, void* zero_ptr,
                          long zero_bytes, const float* rowscale, int rs_rows, int cfg, void* stream) {
    EMIP_REQUIRE(!rowscale || rs_rows > 0);
    EMIP_REQUIRE(A && W && C && M > 0 && N > 0 && K > 0 && (K % 64) == 0);
    EMIP_REQUIRE(aligned16(A) && aligned16(W) && aligned16(C) && (lda % 8) == 0 && (ldw % 8) == 0 && lda >= (A2 ? K1 : K) &&
                 ldw >= K && ldc >= N);
    EMIP_REQUIRE(!R || (aligned16(R) && ldr >= N));
    EMIP_REQUIRE(!A2 || (aligned16(A2) && K1 > 0 && K1 < K && (K1 % 64) == 0 && (lda2 % 8) == 0 && lda2 >= K - K1));
    EMIP_REQUIRE(!lne_stats || lne_colsum);
    EMIP_REQUIRE((zero_bytes % 4) == 0 && (!zero_ptr || (reinterpret_cast<uintptr_t>(zero_ptr) & 3u) == 0));
    const long ab = ((long)(M - 1) * lda + (A2 ? K1 : K)) * 2, wb = ((long)(N - 1) * ldw + K) * 2;
    const long a2b = A2 ? ((long)(M - 1) * lda2 + (K - K1)) * 2 : 0;
    EMIP_REQUIRE(ab < (1L << 31) && wb < (1L << 31) && a2b < (1L << 31));
    G8Args a = {};
    a.A = (const bf16_t*)A; a.A2 = (const bf16_t*)A2; a.W = (const bf16_t*)W; a.C = (bf16_t*)C; a.bias = bias;
    a.R = (const bf16_t*)R; a.M = M; a.N = N; a.K = K; a.K1 = A2 ? K1 : K; a.lda = lda; a.lda2 = A2 ? lda2 : 0; a.ldw = ldw;
    a.ldc = ldc; a.ldr = ldr; a.act = act; a.lne_stats = lne_stats; a.lne_colsum = lne_colsum; a.lne_eps = lne_eps;
    a.out_stats = out_stats; a.zero_ptr = (unsigned*)zero_ptr; a.zero_words = zero_ptr ? zero_bytes / 4 : 0;
    a.a_bytes = (unsigned)ab; a.a2_bytes = (unsigned)a2b; a.w_bytes = (unsigned)wb;
    a.rowscale = rowscale; a.rs_rows = rs_rows;
    return launch(a, cfg, false, (hipStream_t)stream);
}

extern "C" int emip_conv8(const void* X, const void* W, void* Y, const float* bias, const void* R, int B, int H, int Wd,
                          int Cin, long ldx, int Cout, int KH, int KW, int stride, int pad, long ldy, long ldr, int act,
                          const float* ln_stats, const float* tapsum, float ln_eps, float* out_stats, void* zero_ptr,
                          long zero_bytes, int cfg, void* stream) {
    EMIP_REQUIRE(X && W && Y && B > 0 && H > 0 && Wd > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0);
    EMIP_REQUIRE((Cin % 8) == 0 && (ldx % 8) == 0 && ldx >= Cin && aligned16(X) && aligned16(W) && aligned16(Y) && ldy >= Cout);
    EMIP_REQUIRE(!R || (aligned16(R) && ldr >= Cout));
    EMIP_REQUIRE((zero_bytes % 4) == 0 && (!zero_ptr || (reinterpret_cast<uintptr_t>(zero_ptr) & 3u) == 0));
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (Wd + 2 * pad - KW) / stride + 1;
    EMIP_REQUIRE(Ho > 0 && Wo > 0);
    const long ab = (((long)B * H * Wd - 1) * ldx + Cin) * 2, wb = (long)Cout * KH * KW * Cin * 2;
    EMIP_REQUIRE(ab < (1L << 31) && wb < (1L << 31) && (long)B * Ho * Wo < (1L << 31));
    if (ln_stats) {     // per-tap LayerNorm: every tap inside the image, 16-byte aligned tapsum rows, tiles of one round
        EMIP_REQUIRE(tapsum && pad == 0 && ln_eps > 0.f && (Cout % 4) == 0 && (long)B * H * Wd * 8 < (1L << 31) &&
                     (reinterpret_cast<uintptr_t>(tapsum) & 15u) == 0 && (reinterpret_cast<uintptr_t>(ln_stats) & 7u) == 0 &&
                     (Ho - 1) * stride + KH <= H && (Wo - 1) * stride + KW <= Wd);
    }
    G8Args a = {};
    a.A = (const bf16_t*)X; a.W = (const bf16_t*)W; a.C = (bf16_t*)Y; a.bias = bias; a.R = (const bf16_t*)R;
    a.M = B * Ho * Wo; a.N = Cout; a.K = KH * KW * Cin; a.K1 = a.K; a.lda = ldx; a.ldw = (long)KH * KW * Cin; a.ldc = ldy;
    a.ldr = ldr; a.act = act; a.out_stats = out_stats; a.zero_ptr = (unsigned*)zero_ptr;
    a.zero_words = zero_ptr ? zero_bytes / 4 : 0;
    a.H = H; a.Wd = Wd; a.Cin = Cin; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.Ho = Ho; a.Wo = Wo;
    a.a_bytes = (unsigned)ab; a.w_bytes = (unsigned)wb;
    a.ln_stats = ln_stats; a.tapsum = tapsum; a.ln_eps = ln_eps; a.s_bytes = ln_stats ? (unsigned)((long)B * H * Wd * 8) : 0u;
    return launch(a, cfg, true, (hipStream_t)stream);
}

// emip_conv8 with the statistics workspace of emip_gemm_ln_ws (row statistics of an output whose rows span more than two column
// tiles combined inside the launch; emip_gemm_stats_ws_bytes(B Ho Wo, Cout) bytes, ticket block zero)
extern "C" int emip_conv8_ws(const void* X, const void* W, void* Y, const float* bias, const void* R, int B, int H, int Wd,
                             int Cin, long ldx, int Cout, int KH, int KW, int stride, int pad, long ldy, long ldr, int act,
                             const float* ln_stats, const float* tapsum, float ln_eps, float* out_stats, void* zero_ptr,
                             long zero_bytes, int cfg, void* stats_ws, long stats_ws_bytes, void* stream) {
    EMIP_REQUIRE(!stats_ws || (stats_ws_bytes > 0 && (reinterpret_cast<uintptr_t>(stats_ws) & 63u) == 0));
    emip_internal::t_stats_ws = stats_ws;
    emip_internal::t_stats_ws_bytes = stats_ws ? stats_ws_bytes : 0;
    const int rc = emip_conv8(X, W, Y, bias, R, B, H, Wd, Cin, ldx, Cout, KH, KW, stride, pad, ldy, ldr, act, ln_stats, tapsum, ln_eps,
                              out_stats, zero_ptr, zero_bytes, cfg, stream);
    emip_internal::t_stats_ws = nullptr;
    emip_internal::t_stats_ws_bytes = 0;
    return rc;
}
