// The Mlp half of a PVTv2 stage-3 block (22 x 22 tokens, C = 320, hidden 1280) per BAND of an image, one launch, the hidden
// tensor on the CU only (bf16 inference):
//
//     out = x + fc2( GELU( dwconv3x3( LN(x) W1^T + b1 ) + bd ) ) + b2          and the row statistics of `out`
//
// (/root/reference/lib/pvt_v2.py:45-54 Mlp.forward with DWConv :316-327, inside Block.forward :165-169; norm2 folded into
// W1 / b1 and applied on the output side from the row statistics that travel with the residual stream, like emip_gemm_lne.)
//
// Round 3's emip_mlp_block did the same in one launch and lost to the two launches it replaced (119 against 79 us): its fc1,
// depthwise and fc2 phases ran one after the other on all ten waves, three barriers per 64-channel chunk, single-buffered
// weights.  This kernel is built around what that ablation showed:
//   * a workgroup = one QUARTER of an image's tokens (121, token-contiguous; + 23 halo tokens on either side for the 3 x 3
//     taps: <= 167 fc1 rows), 8 waves, the whole LDS of a CU.  64 workgroups for a 16-image step: with several steps in
//     flight the chip fills with the bands of different steps, and what a launch costs the others is its CU-time;
//   * the hidden dimension is walked in 40 chunks of 32 channels as a THREE-STAGE software pipeline with ONE s_barrier per
//     iteration: iteration t runs fc1 of chunk t (MFMA; -> H[t & 1] in LDS), the depthwise 3 x 3 + GELU of chunk t - 1 (VALU;
//     H[(t-1) & 1] -> G[(t-1) & 1]) and fc2 of chunk t - 2 (MFMA; G[t & 1] -> accumulators).  The three touch different
//     buffers, so waves 0-3 run them as fc1, fc2, depthwise and waves 4-7 as depthwise, fc1, fc2: the two waves of a SIMD are
//     on different pipes most of the time (the phases of round 3's kernel ran in lockstep);
//   * fc1: waves 0..5 own one 32-token tile of the band + halo each, their raw tokens as MFMA B fragments in 80 registers for
//     the whole launch; fc2 (40 tiles of 32 tokens x 32 output channels): waves 6 and 7, which hold no tokens, own ALL ten
//     channel tiles of token tiles 0 and 1 (160 accumulator registers), waves 0-2 / 3-5 share token tile 2 / 3 as 3 + 3 + 4
//     channel tiles -- 26-28 MFMAs per wave and chunk for the token-holding waves, 20 for the other two, and no wave holds more
//     than 160 long-lived registers (tokens and accumulators live in ONE register array: the roles are wave-uniform, the
//     compiler cannot know they exclude each other); the depthwise pass: wave w owns 8 channels (w & 3) x half of the
//     band's tokens, ONE token per lane -- the
//     nine neighbour addresses are loop invariants (taps outside the image point at a zero row: no masks) and the wave's 72 tap
//     weights + 8 biases are wave-uniform scalars (s_load, no LDS traffic);
//   * both weight matrices are packed on the host in MFMA-fragment order (one 1-KB piece = the A operand of all 64 lanes), a
//     pipeline stage t = [W1 chunk t | W2 chunk t - 2 | fc1 bias + column sums of chunk t] = 41 contiguous KB that go through
//     a 3-slot LDS-DMA ring, two stages in flight, one counted s_waitcnt per iteration; fragment reads are conflict-free
//     ds_read_b128 at lane * 16;
//   * epilogue: + b2 + x, one rounding, through an LDS image of the band to whole 128-byte row segments; row sums / sums of
//     squares by fixed-order lane reductions (no atomics: two runs give the same bits).
// The halo rows are recomputed by the neighbouring band (fc1 work x 1.38 for the inner bands), the price of no exchange
// between workgroups inside the launch.
#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 md_rsrc(const void* ptr, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
    return i32x4{(int)(unsigned)a, (int)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void md_dma16(unsigned lds_dst, unsigned voff, i32x4 rs) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_dst), "v"(voff), "s"(rs)
        : "memory");
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// gelu_poly (common.h) on two values at once: the same operations in the same order, as packed instructions
__device__ __forceinline__ f32x2 gelu_poly2(f32x2 x) {
    const f32x2 t = {__builtin_amdgcn_fmed3f(x[0], -4.0f, 4.0f), __builtin_amdgcn_fmed3f(x[1], -4.0f, 4.0f)};
    const f32x2 u = t * t;
    f32x2 p = __builtin_elementwise_fma(f32x2{2.258824939e-08f, 2.258824939e-08f}, u, f32x2{-1.588829207e-06f, -1.588829207e-06f});
    p = __builtin_elementwise_fma(p, u, f32x2{4.776392641e-05f, 4.776392641e-05f});
    p = __builtin_elementwise_fma(p, u, f32x2{-8.121878305e-04f, -8.121878305e-04f});
    p = __builtin_elementwise_fma(p, u, f32x2{8.763692481e-03f, 8.763692481e-03f});
    p = __builtin_elementwise_fma(p, u, f32x2{-6.455441459e-02f, -6.455441459e-02f});
    p = __builtin_elementwise_fma(p, u, f32x2{3.978702669e-01f, 3.978702669e-01f});
    return x * __builtin_elementwise_fma(t, p, f32x2{0.5f, 0.5f});
}
template <int N>
__device__ __forceinline__ void md_wait() {
    // ... and every LDS write of this wave (H, G) has landed before the barrier that follows: a raw s_barrier waits for no counter
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

struct MdArgs {
    const bf16_t* X;        // [B * 484, ldx] raw residual stream
    bf16_t* Out;            // [B * 484, ldo] (must not alias X: a band's halo tokens are another band's outputs)
    const char* Wst;        // [42 stages][41 984 B]: W1 chunk t | W2 chunk t - 2 | b1, column sums of chunk t (ops.mlp_band_packs)
    const float* taps;      // [40 chunks][10][32]: the 9 depthwise taps and the depthwise bias of the chunk's channels
    const float* b2;        // [320]
    const float* ln_stats;  // [B * 484, 2] (sum, sum of squares) of the rows of X
    float* out_stats;       // [B * 484, 2] of the rows of Out (stored, not accumulated), or null
    long ldx, ldo;
    int B, xcd_map;
    float eps;
    unsigned wst_bytes;
#ifdef EMIP_TUNING
    unsigned long long* prof;   // calibration build only: per (workgroup, wave) cycles in [barrier wait, DMA issue, fc1, fc2, depthwise, total]
    int skip;               // calibration build only: 1 no fc1 MFMAs, 2 no depthwise pass, 4 no fc2 MFMAs, 8 no weight DMA, 16 no H stores, 32 constant taps, 64 no G stores
#endif
};
#ifdef EMIP_TUNING
#define MD_SKIP(bit) (p.skip & (bit))
#define MD_SKIPV p.skip
#define MD_STAMP(i) do { if (p.prof) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_readcyclecounter(); pacc[i] += now_ - last_; last_ = now_; } } while (0)
#else
#define MD_SKIP(bit) false
#define MD_SKIPV 0
#define MD_STAMP(i) do {} while (0)
#endif

constexpr int MD_C = 320, MD_N = 1280, MD_CH = 32, MD_NCH = MD_N / MD_CH, MD_NST = MD_NCH + 2;
constexpr int MD_HW = 22, MD_TOK = MD_HW * MD_HW, MD_BAND = MD_TOK / 4, MD_HALO = MD_HW + 1, MD_HROWS = MD_BAND + 2 * MD_HALO;   // sizes of the 4-band form (the 8-band form needs less)
constexpr int MD_STAGE = 41 * 1024, MD_W2OFF = 20 * 1024, MD_COFF = 40 * 1024;
constexpr int MD_HB = 64 + MD_HROWS * 64, MD_GB = MD_BAND * 64;                 // H buffer: zero row + 167 rows; G buffer: 121 rows
constexpr int OFF_DUMP = 160 * 1024 - 64;                                       // 64 bytes nobody reads: where masked-out lanes store
constexpr int OFF_H = 3 * MD_STAGE, OFF_G = OFF_H + 2 * MD_HB, MD_LDS = 160 * 1024;
constexpr int MD_OROW = 2 * MD_C + 16;                                          // bytes of a row of the epilogue's output image
static_assert(OFF_G + 2 * MD_GB + 7 * 64 <= MD_LDS, "fc2 reads token slots 121..127 of the second G buffer");
static_assert(MD_BAND * MD_OROW <= 3 * MD_STAGE, "the output image lies in the ring");
static_assert(MD_BAND * 4 == MD_TOK && MD_HROWS <= 6 * 32 && MD_BAND <= 4 * 32 && (MD_TOK + 7) / 8 <= 2 * 32 && (MD_TOK + 7) / 8 + 2 * MD_HALO <= 4 * 32, "tile counts");

// One pipeline iteration of one wave, HAND-SCHEDULED: the three chains of an iteration are independent --
//   chain A: W1 fragments of the stage -> 20 MFMAs -> output-side LayerNorm + bias -> H (bf16 [row][32 ch], 64-byte rows, 16-byte
//            chunk c of row r at c ^ ((r >> 2) & 3))
//   chain B: nine rows of the previous chunk's H -> 3 x 3 taps (wave-uniform scalars) + bias -> GELU -> G (same layout)
//   chain C: G of the chunk before that x W2 fragments of the stage -> accumulators
// -- and what makes the launch fast is B's ~190 vector instructions sitting in the shadow of A's and C's matrix instructions
// (an MFMA occupies the matrix pipe for 32 cycles and the wave's issue port for 8).  hipcc does not produce that order: left
// alone it emits the chains one after the other, and sched_group_barrier patterns did not survive its other passes here.  So
// the order is written down: behind every MFMA one SLICE of chain B (a tap = 4 dwords: 8 unpack + 4 v_pk_fma_f32; later the
// GELU of one channel pair), the A-operand fragment reads three MFMAs ahead in a rotating register set, and a
// sched_barrier(0) behind every step, which nothing crosses.  The LDS regions arrive as __restrict__ pointers (through one char
// array the compiler must assume that H written and H read alias, and orders every load behind every store).
template <bool FC1, bool F1, bool DW, bool F2, int NACC, bool PRE>
__device__ __forceinline__ void md_body(const char* __restrict__ st, char* __restrict__ hw, const char* __restrict__ hr,
                                        char* __restrict__ gw, const char* __restrict__ gr, char* __restrict__ dump,
                                        const float* __restrict__ tpn, float (&tap)[80], const u32x4 (&xf)[FC1 ? 20 : 1], f32x16 (&oacc)[NACC],
                                        const unsigned (&hoff)[9], float rs, float mrs, int hw_off, int hw_sw, int gw_off, int cg, int h,
                                        int gr_off, int gr_sw, int d0, int lane) {
    constexpr int N1 = (FC1 && F1) ? 20 : 0, N2 = F2 ? 2 * NACC : 0, NM = N1 + N2;      // matrix instructions of this iteration
    // ---- loads first: the nine H rows of this lane's token (taps outside the image: the zero row at offset 0), its G column
    uint4 hraw[9];
    if (DW)
#pragma unroll
        for (int k = 0; k < 9; ++k) hraw[k] = *reinterpret_cast<const uint4*>(hr + hoff[k]);
    uint4 g0 = make_uint4(0u, 0u, 0u, 0u), g1 = g0;
    if (F2) {
        g0 = *reinterpret_cast<const uint4*>(gr + gr_off + ((h ^ gr_sw) * 16));
        g1 = *reinterpret_cast<const uint4*>(gr + gr_off + (((2 + h) ^ gr_sw) * 16));
    }
    // A-operand fragment m of the iteration: W1 piece m (m < N1), then W2 piece (d0 + d) * 2 + sp with m - N1 = 2 d + sp
    const char* w1b = st + lane * 16;
    const char* w2b = st + MD_W2OFF + d0 * 2048 + lane * 16;
    auto frag = [&](int m) -> uint4 {
        return *reinterpret_cast<const uint4*>(m < N1 ? w1b + m * 1024 : w2b + (m - N1) * 1024);
    };
    constexpr int AHEAD = 3;
    uint4 wf[AHEAD + 1];
#pragma unroll
    for (int m = 0; m < AHEAD && m < NM; ++m) wf[m] = frag(m);
    __builtin_amdgcn_sched_barrier(0);

    f32x16 hacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) hacc[r] = 0.f;
    f32x2 o[4];                                            // chain B: four channel pairs of this lane's token
    uint4 ov = make_uint4(0u, 0u, 0u, 0u);
    // slice s of chain B: 0..8 = tap s, 9..12 = GELU + rounding of channel pair s - 9, 13 = the store
    auto slice = [&](int sl) {
        if (!DW) return;
        if (sl == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = f32x2{tap[72 + 2 * j], tap[72 + 2 * j + 1]};
        }
        if (sl < 9) {
            const unsigned rw[4] = {hraw[sl].x, hraw[sl].y, hraw[sl].z, hraw[sl].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2 v = {__uint_as_float(rw[j] << 16), __uint_as_float(rw[j] & 0xFFFF0000u)};
                const f32x2 w = {tap[8 * sl + 2 * j], tap[8 * sl + 2 * j + 1]};
                o[j] = __builtin_elementwise_fma(v, w, o[j]);
            }
        } else if (sl < 13) {
            const int j = sl - 9;
            const f32x2 g = gelu_poly2(o[j]);
            bf16x2 b2_;
            b2_[0] = (bf16_t)g[0];
            b2_[1] = (bf16_t)g[1];
            reinterpret_cast<unsigned*>(&ov)[j] = __builtin_bit_cast(unsigned, b2_);
        } else if (sl == 13) {
            // (lanes without a token store to the dump row: no branch)
            *reinterpret_cast<uint4*>(gw_off >= 0 ? gw + gw_off : dump + 16 * cg) = ov;
        }
    };
    constexpr int NSL = 14;
    // chain A's tail in four slices (one per 8 hidden channels), placed behind the LAST four matrix instructions of the iteration
    auto tail = [&](int g) {
        if (!(FC1 && F1)) return;
        const float* cs = reinterpret_cast<const float*>(st + MD_COFF);       // [b1 (32) | column sums of W1 (32)]
        char* hb = hw_off >= 0 ? hw + hw_off : dump + 8 * h;
        const float4 b4 = *reinterpret_cast<const float4*>(cs + 8 * g + 4 * h);
        const float4 c4 = *reinterpret_cast<const float4*>(cs + 32 + 8 * g + 4 * h);
        bf16x4 hv;
        hv[0] = (bf16_t)fmaf(hacc[4 * g + 0], rs, fmaf(-mrs, c4.x, b4.x));
        hv[1] = (bf16_t)fmaf(hacc[4 * g + 1], rs, fmaf(-mrs, c4.y, b4.y));
        hv[2] = (bf16_t)fmaf(hacc[4 * g + 2], rs, fmaf(-mrs, c4.z, b4.z));
        hv[3] = (bf16_t)fmaf(hacc[4 * g + 3], rs, fmaf(-mrs, c4.w, b4.w));
        *reinterpret_cast<bf16x4*>(hb + ((g ^ hw_sw) * 16)) = hv;
    };

    // ---- the schedule: step m = matrix instruction m, the fragment read of m + AHEAD, one slice of chain B (steps without an
    // MFMA -- iterations 0, 1 and the tile-less rows of an outer band -- still run their slices)
    constexpr int NSTEP = NM > NSL ? NM : NSL;
#pragma unroll
    for (int m = 0; m < NSTEP; ++m) {
        if (m < NM) {
            const bf16x8 a = __builtin_bit_cast(bf16x8, wf[m % (AHEAD + 1)]);
            if (m < N1) {
                hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, xf[m < 20 ? m : 0]), hacc, 0, 0, 0);
            } else {
                const int k = m - N1;
                oacc[k >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, (k & 1) ? g1 : g0), oacc[k >> 1], 0, 0, 0);
            }
            if (m + AHEAD < NM) wf[(m + AHEAD) % (AHEAD + 1)] = frag(m + AHEAD);
        }
        if (m < NSL) slice(m);
        // chain A's tail needs all of fc1's MFMAs: behind the last four matrix instructions when those are fc2's
        if (N2 >= 4 && m >= NSTEP - 4) tail(m - (NSTEP - 4));
        __builtin_amdgcn_sched_barrier(0);
    }
    if (N2 < 4)
#pragma unroll
        for (int g = 0; g < 4; ++g) tail(g);
    // ---- the NEXT iteration's taps + bias (8 channels x 10, wave-uniform: scalar loads), issued behind this iteration's last LDS
    // read: a scalar load in flight turns every LDS wait of the block into lgkmcnt(0) (scalar loads return out of order).
    // (the last iterations load chunk 0's again: no branch; waves without a share of the depthwise pass load nothing)
    if (PRE)
#pragma unroll
        for (int k = 0; k < 10; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) tap[8 * k + j] = tpn[k * MD_CH + j];
}

// one wave's share of the launch.  NB: bands per image (4: 121 tokens per workgroup; 8: 60 / 61, for launches that would leave
// most of the chip idle with four); FC1: the wave holds a 32-row tile of the band + halo; DWA: it has a share of the depthwise
// pass; NACC: its fc2 accumulator tiles (0: none).  Each instantiation is entered through a wave-uniform branch: every wave
// executes the same sequence of s_barriers, and each role gets a register allocation of its own.
//   NB = 4: waves 0..5 <FC1, DWA, 4> (fc2: token tile 2 + w / 3, channel tiles 3 (w % 3) .. + 3), waves 6, 7 <-, DWA, 10> (token
//           tile w - 6, all ten channel tiles);
//   NB = 8: waves 0..3 <FC1, -, 0> (the four fc1 tiles), waves 4..7 <-, DWA, 5> (the depthwise pass, one token per lane, and fc2:
//           token tile (w - 4) >> 1, channel tiles 5 ((w - 4) & 1) .. + 4)
template <int NB, bool FC1, bool DWA, int NACC>
__device__ __forceinline__ void md_run(const MdArgs& p, const float* __restrict__ taps, char* smem, int wave, int lane, int img, int band) {
    const int lq = lane & 31, h = lane >> 5;
    const int t0 = MD_TOK * band / NB, nb = MD_TOK * (band + 1) / NB - t0;      // first own token (inside the image), own tokens
    const int h0 = max(t0 - MD_HALO, 0), h1 = min(t0 + nb + MD_HALO, MD_TOK);
    const int nh = h1 - h0;                                          // fc1 rows: own + one or two halos
    const long rowbase = (long)img * MD_TOK;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)smem;
    const i32x4 rsW = md_rsrc(p.Wst, p.wst_bytes);

    // ---- weight stream: stage s -> slot s % 3; wave w moves pieces 5 w .. 5 w + 4, wave 0 also piece 40 (the constants)
    auto issue = [&](int s) {
        const unsigned base = lds0 + (unsigned)((s % 3) * MD_STAGE);
        const unsigned src = (unsigned)(s * MD_STAGE + lane * 16);
        if (MD_SKIP(8)) return;
#pragma unroll
        for (int j = 0; j < 5; ++j) md_dma16(base + (5 * wave + j) * 1024, src + (5 * wave + j) * 1024, rsW);
        if (wave == 0) md_dma16(base + 40 * 1024, src + 40 * 1024, rsW);
    };

    // ---- fc1 operands: this lane's token of the band + halo, raw, as B fragments; its LayerNorm statistics
    issue(0);
    issue(1);
    u32x4 xf[FC1 ? 20 : 1];
    float rs = 0.f, mrs = 0.f;
    const int fr = 32 * wave + lq;                                   // fc1 row (index into band + halo) of this lane
    if (FC1) {
        const long tr = rowbase + h0 + min(fr, nh - 1);              // rows beyond the band: clamped (computed, never stored)
        const bf16_t* xr = p.X + tr * p.ldx + 8 * h;
#pragma unroll
        for (int i = 0; i < 20; ++i) xf[i] = *reinterpret_cast<const u32x4*>(xr + 16 * i);
        const float2 s2 = *reinterpret_cast<const float2*>(p.ln_stats + 2 * tr);
        const float mu = s2.x * (1.f / MD_C);
        rs = rsqrtf(fmaxf(s2.y * (1.f / MD_C) - mu * mu, 0.f) + p.eps);
        mrs = mu * rs;
        // the compiler's wait for these loads belongs HERE: left to the first use it sits inside the loop, where a vmcnt(0) per
        // iteration would drain the weight stream the compiler cannot see (the LDS-DMAs are inline asm)
#pragma unroll
        for (int i = 0; i < 20; ++i) asm volatile("" : "+v"(xf[i]));
        asm volatile("" : "+v"(rs), "+v"(mrs));
    }
    if (wave == 7 && lane < 32) {                                    // the zero rows in front of both H buffers
        reinterpret_cast<unsigned*>(smem + OFF_H)[lane & 15] = 0u;
        reinterpret_cast<unsigned*>(smem + OFF_H + MD_HB)[lane & 15] = 0u;
    }

    // ---- depthwise pass geometry: channel group cg = w & 3 (8 channels), one own token per lane: NB = 4: token half w >> 2 of the
    // band on every wave; NB = 8: the band's 60 / 61 tokens on each of the waves 4..7
    const int cg = wave & 3;
    const int q = (NB == 4 ? 64 * (wave >> 2) : 0) + lane;           // own-token index of this lane
    const bool q_ok = DWA && q < nb;
    unsigned hoff[9];                                                // byte offsets into an H buffer of the nine taps' rows
    {
        const int pq = t0 + min(q, nb - 1);                          // token inside the image
        const int py = pq / MD_HW, px = pq - py * MD_HW;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int yy = py + ky - 1, xx = px + kx - 1;
                const bool in = (unsigned)yy < (unsigned)MD_HW && (unsigned)xx < (unsigned)MD_HW;
                const int r = yy * MD_HW + xx - h0;                  // row of the H buffer
                hoff[3 * ky + kx] = in ? (unsigned)(64 + r * 64 + ((cg ^ ((r >> 2) & 3)) * 16)) : 0u;
            }
    }
    const int qs = min(q, nb - 1);
    const unsigned goff = (unsigned)(qs * 64 + ((cg ^ ((qs >> 2) & 3)) * 16));

    // ---- fc2 geometry (see the table above)
    constexpr int NA = NACC > 0 ? NACC : 1;
    const int w3 = (NB == 4 && FC1) ? wave % 3 : 0;
    const int tw = NB == 4 ? (FC1 ? 2 + wave / 3 : wave - 6) : ((wave - 4) >> 1);
    const int d0 = NB == 4 ? (FC1 ? 3 * w3 : 0) : 5 * ((wave - 4) & 1);
    // (NB = 4: waves 0..5 with three own tiles run a fourth -- their neighbour's first -- and drop it: a branch in the loop body
    // would cut the block the schedule is written for, and two MFMAs per iteration are cheaper than that)
    const bool four = !(NB == 4 && FC1) || w3 == 2;
    const int n2 = 32 * tw + lq;                                     // own-token index of this lane's fc2 column
    f32x16 oacc[NA];
#pragma unroll
    for (int d = 0; d < NA; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;

#ifdef EMIP_TUNING
    unsigned long long pacc[5] = {0, 0, 0, 0, 0}, last_ = __builtin_readcyclecounter();
    const unsigned long long start_ = last_;
#endif
    float tap[80];                                                   // the depthwise taps + bias of the NEXT depthwise pass (scalars)
#pragma unroll
    for (int k = 0; k < 80; ++k) tap[k] = 0.f;
    // one iteration (md_body below); which of the three phases exist is a compile-time matter -- the first two and the last two
    // iterations lack some -- so that the steady state, iterations 2 .. 39, is ONE basic block
    auto iteration = [&](int t, auto f1_, auto dw_, auto f2_) {
        constexpr bool F1 = decltype(f1_)::value, DW = decltype(dw_)::value, F2 = decltype(f2_)::value;
        // stage t has landed once all but this wave's pieces of stage t + 1 (issued one iteration ago) are done
        if (t + 1 < MD_NST) {
            if (wave == 0) md_wait<6>();
            else md_wait<5>();
        } else {
            md_wait<0>();
        }
        __builtin_amdgcn_s_barrier();                      // ... for every wave; H / G of the previous iteration are complete, and
        __builtin_amdgcn_sched_barrier(0);                 // everyone has left stage t - 1's slot and the buffers written next
        MD_STAMP(0);
        if (t + 2 < MD_NST) issue(t + 2);
        MD_STAMP(1);
        const int b1_ = t & 1, b0_ = b1_ ^ 1;               // H[t & 1] is written, H[(t - 1) & 1] read; G[(t - 1) & 1] written, G[t & 1] read
        md_body<FC1, F1, DW && DWA, F2 && (NACC > 0), NA, DWA>(
            smem + (t % 3) * MD_STAGE, smem + OFF_H + b1_ * MD_HB, smem + OFF_H + b0_ * MD_HB, smem + OFF_G + b0_ * MD_GB,
            smem + OFF_G + b1_ * MD_GB, smem + OFF_DUMP, taps + (long)((MD_SKIP(32) || t >= MD_NCH) ? 0 : t) * (10 * MD_CH) + 8 * cg,
            tap, xf, oacc, hoff,
            rs, mrs, (FC1 && fr < nh) ? 64 + fr * 64 + 8 * h : -1, (FC1 && fr < nh) ? (fr >> 2) & 3 : 0, q_ok ? (int)goff : -1, cg, h,
            n2 * 64, (n2 >> 2) & 3, d0, lane);
        MD_STAMP(2);
    };
    typedef std::true_type Y_;
    typedef std::false_type N_;
    iteration(0, Y_{}, N_{}, N_{});
    iteration(1, Y_{}, Y_{}, N_{});
#pragma unroll 1
    for (int t = 2; t < MD_NCH; ++t) iteration(t, Y_{}, Y_{}, Y_{});
    iteration(MD_NCH, N_{}, Y_{}, Y_{});
    iteration(MD_NCH + 1, N_{}, N_{}, Y_{});
#ifdef EMIP_TUNING
    if (p.prof && lane == 0) {
        unsigned long long* pr = p.prof + ((long)blockIdx.x * 8 + wave) * 6;
        for (int i = 0; i < 5; ++i) pr[i] = pacc[i];
        pr[5] = __builtin_readcyclecounter() - start_;
    }
#endif

    // ---- epilogue: + b2 + x in the accumulator layout, one rounding, -> an LDS image of the band (in the idle ring).
    // register 4 g + j of accumulator tile d = output channel 32 (d0 + d) + 8 g + 4 h + j of token n2
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (NACC > 0) {
        const bool ok = n2 < nb;
        const long tr = rowbase + t0 + min(n2, nb - 1);
        const bf16_t* xr = p.X + tr * p.ldx;
        char* ob = smem + min(n2, nb - 1) * MD_OROW;
#pragma unroll
        for (int d = 0; d < NACC; ++d) {
            if (d == 3 && !four) break;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = 32 * (d0 + d) + 8 * g + 4 * h;
                const float4 bv = *reinterpret_cast<const float4*>(p.b2 + ch);
                const bf16x4 xv = *reinterpret_cast<const bf16x4*>(xr + ch);
                bf16x4 ov;
                ov[0] = (bf16_t)(oacc[d][4 * g + 0] + bv.x + (float)xv[0]);
                ov[1] = (bf16_t)(oacc[d][4 * g + 1] + bv.y + (float)xv[1]);
                ov[2] = (bf16_t)(oacc[d][4 * g + 2] + bv.z + (float)xv[2]);
                ov[3] = (bf16_t)(oacc[d][4 * g + 3] + bv.w + (float)xv[3]);
                if (ok) *reinterpret_cast<bf16x4*>(ob + ch * 2) = ov;
            }
        }
    }
}

template <int NB>
__global__ __launch_bounds__(512) void mlp_band_kernel(const MdArgs p, const float* __restrict__ taps) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int img, band;
    if (p.xcd_map) {       // the bands of an image on one XCD: they share its token rows in that L2
        const int j = blockIdx.x >> 3;
        band = j % NB;
        img = (blockIdx.x & 7) + 8 * (j / NB);
    } else {
        band = blockIdx.x % NB;
        img = blockIdx.x / NB;
    }
    if (NB == 4) {
        if (wave < 6) md_run<4, true, true, 4>(p, taps, smem, wave, lane, img, band);
        else md_run<4, false, true, 10>(p, taps, smem, wave, lane, img, band);
    } else {
        if (wave < 4) md_run<8, true, false, 0>(p, taps, smem, wave, lane, img, band);
        else md_run<8, false, true, 5>(p, taps, smem, wave, lane, img, band);
    }

    // ---- the band's rows out of the LDS image, whole 128-byte segments, and their statistics
    __syncthreads();
    const int t0 = MD_TOK * band / NB, nb = MD_TOK * (band + 1) / NB - t0;
    const long rowbase = (long)img * MD_TOK;
    // 8 lanes per row: lane j of the group moves the 16-byte chunks j, j + 8, .. j + 32 (128-byte segments per group and step)
    const int grp = tid >> 3, gl = tid & 7;
#pragma unroll
    for (int pass = 0; pass < (NB == 4 ? 2 : 1); ++pass) {
        const int row = grp + 64 * pass;
        const bool ok = row < nb;
        const char* ob = smem + min(row, nb - 1) * MD_OROW;
        bf16_t* op = p.Out + (rowbase + t0 + min(row, nb - 1)) * p.ldo;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const uint4 v = *reinterpret_cast<const uint4*>(ob + (gl + 8 * k) * 16);
            if (ok) *reinterpret_cast<uint4*>(op + (gl + 8 * k) * 8) = v;
            const unsigned vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = __uint_as_float(vw[j] << 16), b = __uint_as_float(vw[j] & 0xFFFF0000u);
                s1 += a + b;
                s2 = fmaf(a, a, fmaf(b, b, s2));
            }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            s1 += __shfl_xor(s1, o);
            s2 += __shfl_xor(s2, o);
        }
        if (p.out_stats && ok && gl == 0) *reinterpret_cast<float2*>(p.out_stats + 2 * (rowbase + t0 + row)) = make_float2(s1, s2);
    }
}

}  // namespace

#ifdef EMIP_TUNING
static int g_md_skip = 0;
static unsigned long long* g_md_prof = nullptr;
extern "C" int emip_debug_set_md(int flags) { g_md_skip = flags; return 0; }
extern "C" int emip_debug_set_md_prof(void* buf) { g_md_prof = (unsigned long long*)buf; return 0; }   // [B * 4 * 8][6] u64 or NULL
#endif

extern "C" int emip_mlp_band_eligible(int B, int H, int W, int C, int N) {
    return B > 0 && H == MD_HW && W == MD_HW && C == MD_C && N == MD_N;
}
extern "C" int emip_mlp_band_stage_bytes(void) { return MD_NST * MD_STAGE; }

// Out = X + fc2(GELU(dwconv3x3(LN(X) W1^T + b1) + bd)) + b2 per image, out_stats = (sum, sum of squares) of the rows of Out (may
// be NULL).  X, Out: bf16 [B, 22, 22, 320] with row strides ldx / ldo, Out must not overlap X.  Wst: the 42 pipeline stages of
// ops.mlp_band_packs (fragment-order W1 with the LayerNorm scale folded in, fragment-order W2, fc1 bias + W1 beta, row sums of the
// packed W1), taps: f32 [40][10][32] (9 depthwise taps + bias per hidden channel, chunk-major), b2: f32 [320], ln_stats: f32
// [B 484][2] (sum, sum of squares) of the rows of X.  bands: workgroups per image, 4 or 8 (0: 8 while four would leave CUs
// idle, else 4; the result does not depend on it: every output element is one accumulator chain over the same 40 chunks).
extern "C" int emip_mlp_band(const void* X, long ldx, const void* Wst, const float* taps, const float* b2, const float* ln_stats,
                             float eps, void* Out, long ldo, float* out_stats, int B, int H, int W, int C, int N, int bands,
                             void* stream) {
    EMIP_REQUIRE(X && Wst && taps && b2 && ln_stats && Out && emip_mlp_band_eligible(B, H, W, C, N));
    EMIP_REQUIRE(bands == 0 || bands == 4 || bands == 8);
    if (bands == 0) bands = B * 4 < 256 ? 8 : 4;
    EMIP_REQUIRE(ldx >= C && (ldx & 7) == 0 && ldo >= C && (ldo & 7) == 0);
    EMIP_REQUIRE(aligned16(X) && aligned16(Wst) && aligned16(taps) && aligned16(b2) && aligned16(Out) &&
                 (reinterpret_cast<uintptr_t>(ln_stats) & 7u) == 0 && (reinterpret_cast<uintptr_t>(out_stats) & 7u) == 0);
    const long rows = (long)B * H * W;
    {   // no overlap of the two token tensors
        const char* x0 = (const char*)X; const char* x1 = x0 + ((rows - 1) * ldx + C) * 2;
        const char* o0 = (const char*)Out; const char* o1 = o0 + ((rows - 1) * ldo + C) * 2;
        EMIP_REQUIRE(x1 <= o0 || o1 <= x0);
    }
    MdArgs a{};
    a.X = (const bf16_t*)X; a.Out = (bf16_t*)Out; a.Wst = (const char*)Wst; a.taps = taps; a.b2 = b2;
    a.ln_stats = ln_stats; a.out_stats = out_stats; a.ldx = ldx; a.ldo = ldo; a.B = B; a.eps = eps;
    a.xcd_map = (B % 8) == 0;
    a.wst_bytes = (unsigned)(MD_NST * MD_STAGE);
#ifdef EMIP_TUNING
    a.skip = g_md_skip;
    a.prof = g_md_prof;
#endif
    static bool attr = false;
    if (!attr) {
        attr = true;
        if (hipFuncSetAttribute((const void*)mlp_band_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, MD_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)mlp_band_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, MD_LDS) != hipSuccess)
            return EMIP_E_LAUNCH;
    }
    if (bands == 4) hipLaunchKernelGGL(mlp_band_kernel<4>, dim3((unsigned)(B * 4)), dim3(512), MD_LDS, (hipStream_t)stream, a, taps);
    else hipLaunchKernelGGL(mlp_band_kernel<8>, dim3((unsigned)(B * 8)), dim3(512), MD_LDS, (hipStream_t)stream, a, taps);
    return emip_launch_status();
}
