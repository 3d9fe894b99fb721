// The attention half of a PVTv2 block in ONE launch (bf16 inference):
//     x <- x + proj( softmax( (LN(x) Wq^T) k^T * scale ) v )
// (/root/reference/lib/pvt_v2.py:95-127 Attention.forward for sr_ratio > 1, and the residual add of Block.forward :165-168;
// norm1 is folded into Wq / bq and applied on the output side from the row statistics that travel with the residual stream,
// like emip_gemm_lne.  k / v come from the spatial-reduction conv -> LayerNorm -> kv GEMM chain as before.)
//
// The three launches this replaces (q GEMM, emip_sra_attention, proj GEMM) each stream the whole token tensor through HBM /
// L2 once in and once out and pay a launch floor around ~5 us of work.  Here a workgroup owns 128 query rows
// of one image for ALL heads and Q, the scores and the attention output never leave the CU:
//   * a wave owns 32 query rows for the whole kernel.  Their raw tokens become MFMA B-operand fragments once (20 x 16 B per
//     lane for C = 320) and stay in registers; Q^T = Wq x^T comes out of v_mfma_f32_32x32x16_bf16 with the QUERY on the lane,
//     so the LayerNorm statistics are one pair of scalars per lane and the converted accumulators ARE the B operand of
//     S^T = K Q^T -- no LDS round trip.  For that the rows of Wq are stored with bits 2 and 3 of their index swapped inside
//     every 16 (pack time): accumulator registers 8 e .. 8 e + 7 then hold 8 consecutive head channels.
//   * per head: K (A operand, read row-wise) and V (ds_read_tr16_b64) tiles arrive by LDS-DMA one head ahead; one-pass
//     softmax over the <= 128 key slots with the scores of a query in one lane pair; O^T = V^T P leaves the head's output
//     again as B-operand registers, in the place of the head's Q.
//   * out^T = Wp O^T: the columns of Wp carry the same bit swap (the order the O registers have), its rows too (so the
//     output registers hold 8 consecutive channels); + bias, through a wave-private f32 window in LDS to whole 128-byte
//     row segments, + residual, one rounding, row sums / sums of squares of what was stored for the next LayerNorm.
//   * both weight matrices stream through ONE ring of 32-row tiles (32 x 2C bytes, contiguous in memory; 16 B of padding per
//     LDS row keeps the fragment reads conflict-free): LDS-DMA, three slots, two tiles in flight, one s_barrier and one
//     counted s_waitcnt per tile.  Every vector-memory instruction of the kernel that is in flight across such a wait is
//     issued unconditionally (inline asm, out-of-range lanes through the buffer range check), so the counts are exact.
// Workgroup ids are dealt round-robin over the XCDs: image = id % 8 + 8 (id / 8 / tiles), so an image's workgroups share its
// K / V rows in one L2.
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 sb_rsrc(const void* ptr, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
    return i32x4{(int)(unsigned)a, (int)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void sb_dma16(unsigned lds_dst, unsigned voff, i32x4 rs, unsigned soff) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_dst), "v"(voff), "s"(rs), "s"(soff)
        : "memory");
}
// loads / stores hipcc's wait insertion does not see (it would drain the ring in front of their first use)
__device__ __forceinline__ u32x4 sb_load16(i32x4 rs, unsigned voff) {
    u32x4 v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(rs) : "memory");
    return v;
}
__device__ __forceinline__ void sb_store16(u32x4 v, i32x4 rs, unsigned voff) {
    // the s_nop: a store of more than 8 bytes reads its data registers for two more cycles, and hipcc's hazard recogniser
    // does not look into inline asm (it placed a VALU write to the first data register right behind the store)
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rs) : "memory");
}
template <int N>
__device__ __forceinline__ void sb_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct SbArgs {
    const bf16_t* X;        // [B * N, ldx] raw residual stream
    bf16_t* Out;            // [B * N, ldo] (may be X: a wave reads only the rows it writes)
    const float* stats;     // [B * N, 2] (sum, sum of squares) of the rows of X
    float* out_stats;       // [B * N, 2] of the rows of Out, or null
    const bf16_t* Wq;       // [C, C] gamma folded in, rows bit-swapped
    const float* bq;        // [C] bias + Wq beta (channel order)
    const float* csq;       // [C] row sums of the packed Wq (channel order)
    const bf16_t* KV;       // [B, Lk, 2C]
    const bf16_t* Wp;       // [C, C] rows and columns bit-swapped
    const float* bp;        // [C]
    long ldx, ldo;
    int B, N, Lk, tpi, xcd_map;
    float eps, scale;
    unsigned x_bytes, o_bytes;
};

constexpr unsigned SB_OOB = 0x80000000u;

// 16-B chunk c of row `row` of a [rows][128 B] image read row-wise (K as the MFMA A operand, token rows as B)
__device__ __forceinline__ int sb_r128(int row, int c) { return row * 128 + ((c ^ (row & 7)) * 16); }
// ... of the V tile, conflict-free for the transposed reads (as in sra.hip)
__device__ __forceinline__ int sb_v128(int row, int c) { return row * 128 + ((c ^ (((row >> 1) & 1) << 2)) * 16); }
// 16-B chunk c (4 floats) of row `row` of a wave's [32][64] f32 output window
__device__ __forceinline__ int sb_w256(int row, int c) { return row * 256 + ((c ^ (row & 15)) * 16); }

template <int C>
struct SbCfg {
    static constexpr int NT = C / 32, KS = C / 16, HEADS = C / 64, CH = C / 64;
    static constexpr int SB = 2 * C + 16;                                        // bytes of an LDS row of a weight tile
    static constexpr int PP = (((32 * SB + 1023) / 1024) + 3) / 4 * 4;           // 1-KB DMA pieces per tile, 4 waves
    static constexpr int PPW = PP / 4;
    static constexpr int SLOT = PP * 1024, RING = 3 * SLOT;
    static constexpr int KVB = 2 * 128 * 128;                                    // K tile + V tile of one head
    static constexpr int OFF_KV = RING, OFF_TAB = OFF_KV + 2 * KVB;
    static constexpr int LDS = OFF_TAB + 3 * C * 4;
};

template <int C>
__global__ __launch_bounds__(256) void sra_block_kernel(const SbArgs p) {
    typedef SbCfg<C> G;
    constexpr int NT = G::NT, KS = G::KS, HEADS = G::HEADS, CH = G::CH, SB = G::SB, PPW = G::PPW, SLOT = G::SLOT;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, h = lane >> 5, srow = lane >> 3, sch = lane & 7;
    int img, qt;
    if (p.xcd_map) {
        const int j = blockIdx.x >> 3;
        qt = j % p.tpi;
        img = (blockIdx.x & 7) + 8 * (j / p.tpi);
    } else {
        qt = blockIdx.x % p.tpi;
        img = blockIdx.x / p.tpi;
    }
    const long rowbase = (long)img * p.N;
    const int q0 = qt * 128 + wave * 32;                 // first query row (inside the image) of this wave

    const i32x4 rsQ = sb_rsrc(p.Wq, (unsigned)(C * C * 2));
    const i32x4 rsP = sb_rsrc(p.Wp, (unsigned)(C * C * 2));
    const i32x4 rsKV = sb_rsrc(p.KV + (long)img * p.Lk * 2 * C, (unsigned)(p.Lk * 4 * C));
    const i32x4 rsX = sb_rsrc(p.X, p.x_bytes);
    const i32x4 rsO = sb_rsrc(p.Out, p.o_bytes);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)smem;

    // ---- staging plans ------------------------------------------------------------------------------------------------
    // weight tile: LDS byte P = 1024 piece + 16 lane sits in row P / SB; its last 16 bytes are padding (nothing fetched)
    unsigned ringoff[PPW];
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
        const int P = 1024 * (wave + 4 * k) + 16 * lane;
        const int row = P / SB, within = P - row * SB;
        ringoff[k] = (row < 32 && within < 2 * C) ? (unsigned)(row * 2 * C + within) : SB_OOB;
    }
    // K / V tile: piece = 8 rows of 128 B, lane at row lane >> 3, slot lane & 7 fetches the chunk the swizzle puts there
    unsigned koff[4], voff[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = 8 * (wave + 4 * k) + srow;
        const int ck = sch ^ (row & 7), cv = sch ^ (((row >> 1) & 1) << 2);
        koff[k] = row < p.Lk ? (unsigned)(row * 4 * C + 16 * ck) : SB_OOB;
        voff[k] = row < p.Lk ? (unsigned)(row * 4 * C + 2 * C + 16 * cv) : SB_OOB;
    }
    auto issue_kv = [&](int hd, int buf) {
        const unsigned base = lds0 + G::OFF_KV + buf * G::KVB;
        const unsigned so = (unsigned)(hd * 128);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            sb_dma16(base + (wave + 4 * k) * 1024, koff[k], rsKV, so);
            sb_dma16(base + 16384 + (wave + 4 * k) * 1024, voff[k], rsKV, so);
        }
    };
    auto issue_ring = [&](int s, int slot) {             // stage s: tile s of Wq, then tile s - NT of Wp
        const bool isq = s < NT;
        const i32x4 rs = isq ? rsQ : rsP;
        const unsigned so = (unsigned)((isq ? s : s - NT) * 32 * 2 * C);
        const unsigned base = lds0 + slot * SLOT;
#pragma unroll
        for (int k = 0; k < PPW; ++k) sb_dma16(base + (wave + 4 * k) * 1024, ringoff[k], rs, so);
    };

    // ---- prologue: epilogue vectors -> LDS, row statistics, this wave's 32 token rows (full 128-byte lines) --------------
    float* tab = reinterpret_cast<float*>(smem + G::OFF_TAB);          // [bq | column sums of Wq | bp]
    for (int i = tid; i < 3 * C; i += 256) tab[i] = i < C ? p.bq[i] : (i < 2 * C ? p.csq[i - C] : p.bp[i - 2 * C]);
    float rs, mrs;
    {
        const float2 s2 = *reinterpret_cast<const float2*>(p.stats + 2 * (rowbase + min(q0 + lq, p.N - 1)));
        const float mu = s2.x * (1.f / (float)C);
        rs = rsqrtf(fmaxf(s2.y * (1.f / (float)C) - mu * mu, 0.f) + p.eps);
        mrs = mu * rs;
    }
    u32x4 xs[CH][4];
#pragma unroll
    for (int cc = 0; cc < CH; ++cc)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long r = rowbase + min(q0 + srow + 8 * j, p.N - 1);          // rows beyond N: clamped (computed, never stored)
            xs[cc][j] = *reinterpret_cast<const u32x4*>(p.X + r * p.ldx + 64 * cc + 8 * sch);
        }
    issue_kv(0, 0);
    issue_ring(0, 0);
    issue_ring(1, 1);
    // token rows -> B-operand fragments through a wave-private window (it lies in K/V buffer 1, idle until head 1 is fetched)
    u32x4 xf[KS];
    {
        char* win = smem + G::OFF_KV + G::KVB + wave * 4096;
#pragma unroll
        for (int cc = 0; cc < CH; ++cc) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(win + sb_r128(srow + 8 * j, sch)) = xs[cc][j];
#pragma unroll
            for (int i = 0; i < 4; ++i) xf[4 * cc + i] = *reinterpret_cast<const u32x4*>(win + sb_r128(lq, 2 * i + h));
        }
    }

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the epilogue vectors are in LDS before this wave's first barrier

    // ---- phase A: Q^T = Wq x^T, tile by tile; the converted accumulators are the B operands of S^T ------------------------
    bf16x8 qo[HEADS][4];                                  // per head: 4 k-steps of 16 channels; later the head's output
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        sb_wait<PPW>();                                   // all but the newest tile (t + 1) have landed
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        issue_ring(t + 2, (t + 2) % 3);
        const char* sa = smem + (t % 3) * SLOT + lq * SB + 16 * h;
        f32x16 a0, a1;
#pragma unroll
        for (int r = 0; r < 16; ++r) a0[r] = a1[r] = 0.f;
#pragma unroll
        for (int i = 0; i < KS; i += 2) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sa + 32 * i)),
                                                         __builtin_bit_cast(bf16x8, xf[i]), a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sa + 32 * i + 32)),
                                                         __builtin_bit_cast(bf16x8, xf[i + 1]), a1, 0, 0, 0);
        }
        // register 4 g + j = channel 32 t + 16 (g >> 1) + 8 h + 4 (g & 1) + j of this lane's query
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = 32 * t + 16 * (g >> 1) + 8 * h + 4 * (g & 1);
            const float4 b4 = *reinterpret_cast<const float4*>(tab + d0);
            const float4 c4 = *reinterpret_cast<const float4*>(tab + C + d0);
            const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, cs[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                qo[t >> 1][2 * (t & 1) + (g >> 1)][4 * (g & 1) + j] =
                    (bf16_t)fmaf(a0[4 * g + j] + a1[4 * g + j], rs, fmaf(-mrs, cs[j], bb[j]));
        }
    }

    // ---- phase B: per head  S^T = K Q^T, one-pass softmax, O^T = V^T P  (the loop body of sra.hip) -----------------------
    const float sc2 = p.scale * 1.4426950408889634f;
    const int i16 = lane & 15, g16 = (lane >> 4) & 1;
#pragma unroll
    for (int hd = 0; hd < HEADS; ++hd) {
        if (hd == 0) sb_wait<2 * PPW>();                  // K / V of head 0 are older than the two weight tiles in flight
        else sb_wait<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (hd + 1 < HEADS) issue_kv(hd + 1, (hd + 1) & 1);
        const char* kt_ = smem + G::OFF_KV + (hd & 1) * G::KVB;
        const char* vt_ = kt_ + 16384;
        f32x16 s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(kt_ + sb_r128(32 * kt + lq, 2 * i + h))), qo[hd][i],
                    s[kt], 0, 0, 0);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const bool edge = 32 * kt + 31 >= p.Lk;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = s[kt][r] * sc2;
                if (edge && 32 * kt + 8 * (r >> 2) + 4 * h + (r & 3) >= p.Lk) v = -INFINITY;
                s[kt][r] = v;
                mx = fmaxf(mx, v);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float psum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(s[kt][r] - mx);
                s[kt][r] = e;
                psum += e;
            }
        psum += __shfl_xor(psum, 32);
        f32x16 oacc[2];
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                bf16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)s[kt][8 * sp + j];
                const int base0 = 32 * kt + 16 * sp + 4 * h + (i16 >> 2);
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const int col = 32 * d + 16 * g16 + 4 * (i16 & 3);
                    const int c = col >> 3, half = (col >> 2) & 1;
                    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
                    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vt_ + sb_v128(base0, c) + 8 * half));
                    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vt_ + sb_v128(base0 + 8, c) + 8 * half));
                    bf16x8 vf;
                    const bf16x4 b0 = __builtin_bit_cast(bf16x4, v0), b1 = __builtin_bit_cast(bf16x4, v1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        vf[j] = b0[j];
                        vf[4 + j] = b1[j];
                    }
                    oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[d], 0, 0, 0);
                }
            }
        }
        // registers 8 e .. 8 e + 7 of oacc[d] = head channels 32 d + 16 e + {4 h + j, 8 + 4 h + j}: the column order of Wp
        const float inv = 1.0f / psum;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int j = 0; j < 8; ++j) qo[hd][2 * d + e][j] = (bf16_t)(oacc[d][8 * e + j] * inv);
    }

    // ---- phase C: out^T = Wp O^T + bp, two tiles (64 channels) at a time through the f32 window, + residual, store --------
    float* win = reinterpret_cast<float*>(smem + G::OFF_KV + wave * 8192);      // K / V buffers are idle now
    char* winb = reinterpret_cast<char*>(win);
    int slot = NT % 3;
    float st1[4] = {0.f, 0.f, 0.f, 0.f}, st2[4] = {0.f, 0.f, 0.f, 0.f};
    constexpr int NP = NT / 2;
#pragma unroll 1
    for (int P = 0; P < NP; ++P) {
        u32x4 rr[4];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int s = NT + 2 * P + half;
            // vector-memory instructions in issue order:  R(4) D(PPW) | D(PPW) S(4)  per pair (R residual loads, D tile DMA,
            // S stores); the tile needed now was issued two tiles ago
            if (half == 0) {
                if (P == 0) sb_wait<PPW>();
                else sb_wait<PPW + 4>();
            } else {
                if (P == NP - 1) sb_wait<0>();
                else if (P == 0) sb_wait<PPW + 4>();
                else sb_wait<PPW + 8>();
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (half == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const long r = rowbase + min(q0 + srow + 8 * j, p.N - 1);
                    rr[j] = sb_load16(rsX, (unsigned)((r * p.ldx + 64 * P + 8 * sch) * 2));
                }
            }
            if (s + 2 < 2 * NT) issue_ring(s + 2, slot >= 1 ? slot - 1 : 2);       // (slot + 2) % 3
            const char* sa = smem + slot * SLOT + lq * SB + 16 * h;
            f32x16 a0, a1;
#pragma unroll
            for (int r = 0; r < 16; ++r) a0[r] = a1[r] = 0.f;
#pragma unroll
            for (int i = 0; i < KS; i += 2) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sa + 32 * i)),
                                                             qo[i >> 2][i & 3], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sa + 32 * i + 32)), qo[(i + 1) >> 2][(i + 1) & 3], a1, 0, 0, 0);
            }
            // registers 8 e .. 8 e + 7 = channels 32 t + 16 e + 8 h + (0..7) of this lane's query (t = 2 P + half)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int cw = 32 * half + 16 * e + 8 * h;                       // channel inside the 64-channel window
                const float* bp_ = tab + 2 * C + 64 * P + cw;
#pragma unroll
                for (int q4 = 0; q4 < 2; ++q4) {
                    const float4 b4 = *reinterpret_cast<const float4*>(bp_ + 4 * q4);
                    float4 o;
                    o.x = a0[8 * e + 4 * q4 + 0] + a1[8 * e + 4 * q4 + 0] + b4.x;
                    o.y = a0[8 * e + 4 * q4 + 1] + a1[8 * e + 4 * q4 + 1] + b4.y;
                    o.z = a0[8 * e + 4 * q4 + 2] + a1[8 * e + 4 * q4 + 2] + b4.z;
                    o.w = a0[8 * e + 4 * q4 + 3] + a1[8 * e + 4 * q4 + 3] + b4.w;
                    *reinterpret_cast<float4*>(winb + sb_w256(lq, (cw >> 2) + q4)) = o;
                }
            }
            slot = slot == 2 ? 0 : slot + 1;
        }
        // ---- the pair's 64 channels: rows out of the window, + residual, one rounding, statistics, whole row segments -------
        if (P == NP - 1) sb_wait<0>();
        else sb_wait<2 * PPW>();                          // the residual rows are older than the two tiles issued since
        asm volatile("" : "+v"(rr[0]), "+v"(rr[1]), "+v"(rr[2]), "+v"(rr[3]));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = srow + 8 * j, q = q0 + row;
            const float4 lo = *reinterpret_cast<const float4*>(winb + sb_w256(row, 2 * sch));
            const float4 hi = *reinterpret_cast<const float4*>(winb + sb_w256(row, 2 * sch + 1));
            const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            const unsigned rw[4] = {rr[j][0], rr[j][1], rr[j][2], rr[j][3]};
            bf16x8 ov;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ov[2 * k] = (bf16_t)(v[2 * k] + __uint_as_float(rw[k] << 16));
                ov[2 * k + 1] = (bf16_t)(v[2 * k + 1] + __uint_as_float(rw[k] & 0xFFFF0000u));
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float o = (float)ov[k];
                st1[j] += o;
                st2[j] += o * o;
            }
            sb_store16(__builtin_bit_cast(u32x4, ov), rsO,
                       q < p.N ? (unsigned)(((rowbase + q) * p.ldo + 64 * P + 8 * sch) * 2) : SB_OOB);
        }
    }
    if (p.out_stats) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = st1[j], b = st2[j];
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
                a += __shfl_xor(a, o);
                b += __shfl_xor(b, o);
            }
            const int q = q0 + srow + 8 * j;
            if (sch == 0 && q < p.N) *reinterpret_cast<float2*>(p.out_stats + 2 * (rowbase + q)) = make_float2(a, b);
        }
    }
}

// ---- q projection + attention of ONE head per workgroup (the 22 x 22 stage) ------------------------------------------------
// With C = 320 the kernel above leaves three quarters of the SIMDs idle (16 images x 484 tokens are 242 blocks of 32 queries)
// and every wave walks 20 weight tiles and 5 heads on its own: 42 us against 31 for the three launches.  Here the grid is
// (image, 128 queries, HEAD): a wave computes the 64 channels of its head's Q (two weight tiles, both resident: no ring) and
// runs the head -- the work of emip_sra_attention plus 40 MFMAs per wave, without Q ever reaching memory.  The attention
// output goes out as bf16 rows; the proj GEMM (residual, statistics) stays the launch it was.
template <int C>
__global__ __launch_bounds__(256) void sra_q_kernel(const SbArgs p) {
    typedef SbCfg<C> G;
    constexpr int KS = G::KS, HEADS = G::HEADS, CH = G::CH, SB = G::SB, PPW = G::PPW, SLOT = G::SLOT;
    constexpr int OFF_K = 2 * SLOT, OFF_V = OFF_K + 16384;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, h = lane >> 5, srow = lane >> 3, sch = lane & 7;
    const int per = p.tpi * HEADS;
    int img, r;
    if (p.xcd_map) {
        const int j = blockIdx.x >> 3;
        r = j % per;
        img = (blockIdx.x & 7) + 8 * (j / per);
    } else {
        r = blockIdx.x % per;
        img = blockIdx.x / per;
    }
    const int hd = r % HEADS, qt = r / HEADS;
    const long rowbase = (long)img * p.N;
    const int q0 = qt * 128 + wave * 32;

    const i32x4 rsQ = sb_rsrc(p.Wq, (unsigned)(C * C * 2));
    const i32x4 rsKV = sb_rsrc(p.KV + (long)img * p.Lk * 2 * C, (unsigned)(p.Lk * 4 * C));
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)smem;

    float rs, mrs;
    {
        const float2 s2 = *reinterpret_cast<const float2*>(p.stats + 2 * (rowbase + min(q0 + lq, p.N - 1)));
        const float mu = s2.x * (1.f / (float)C);
        rs = rsqrtf(fmaxf(s2.y * (1.f / (float)C) - mu * mu, 0.f) + p.eps);
        mrs = mu * rs;
    }
    // register 4 g + j of tile tt = head channel 32 tt + 16 (g >> 1) + 8 h + 4 (g & 1) + j
    float4 bqv[2][4], csv[2][4];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = 64 * hd + 32 * tt + 16 * (g >> 1) + 8 * h + 4 * (g & 1);
            bqv[tt][g] = *reinterpret_cast<const float4*>(p.bq + d0);
            csv[tt][g] = *reinterpret_cast<const float4*>(p.csq + d0);
        }
    u32x4 xs[CH][4];
#pragma unroll
    for (int cc = 0; cc < CH; ++cc)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long rr = rowbase + min(q0 + srow + 8 * j, p.N - 1);
            xs[cc][j] = *reinterpret_cast<const u32x4*>(p.X + rr * p.ldx + 64 * cc + 8 * sch);
        }
    // K tile and the head's two weight tiles by LDS-DMA (V follows once the staging windows, which lie in its place, are done)
    const unsigned kvso = (unsigned)(hd * 128);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = 8 * (wave + 4 * k) + srow;
        sb_dma16(lds0 + OFF_K + (wave + 4 * k) * 1024, row < p.Lk ? (unsigned)(row * 4 * C + 16 * (sch ^ (row & 7))) : SB_OOB, rsKV,
                 kvso);
    }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int P = 1024 * (wave + 4 * k) + 16 * lane;
            const int row = P / SB, within = P - row * SB;
            sb_dma16(lds0 + tt * SLOT + (wave + 4 * k) * 1024,
                     (row < 32 && within < 2 * C) ? (unsigned)(row * 2 * C + within) : SB_OOB, rsQ,
                     (unsigned)((2 * hd + tt) * 32 * 2 * C));
        }
    u32x4 xf[KS];
    {
        char* win = smem + OFF_V + wave * 4096;
#pragma unroll
        for (int cc = 0; cc < CH; ++cc) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(win + sb_r128(srow + 8 * j, sch)) = xs[cc][j];
#pragma unroll
            for (int i = 0; i < 4; ++i) xf[4 * cc + i] = *reinterpret_cast<const u32x4*>(win + sb_r128(lq, 2 * i + h));
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // every wave has its token fragments: the V tile may land
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = 8 * (wave + 4 * k) + srow;
        sb_dma16(lds0 + OFF_V + (wave + 4 * k) * 1024,
                 row < p.Lk ? (unsigned)(row * 4 * C + 2 * C + 16 * (sch ^ (((row >> 1) & 1) << 2))) : SB_OOB, rsKV, kvso);
    }
    sb_wait<4>();                                         // all but the V pieces
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // ---- Q^T of the head: two tiles ---------------------------------------------------------------------------------------
    bf16x8 qf[4];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const char* sa = smem + tt * SLOT + lq * SB + 16 * h;
        f32x16 a0, a1;
#pragma unroll
        for (int r16 = 0; r16 < 16; ++r16) a0[r16] = a1[r16] = 0.f;
#pragma unroll
        for (int i = 0; i < KS; i += 2) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sa + 32 * i)),
                                                         __builtin_bit_cast(bf16x8, xf[i]), a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sa + 32 * i + 32)),
                                                         __builtin_bit_cast(bf16x8, xf[i + 1]), a1, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float bb[4] = {bqv[tt][g].x, bqv[tt][g].y, bqv[tt][g].z, bqv[tt][g].w};
            const float cs[4] = {csv[tt][g].x, csv[tt][g].y, csv[tt][g].z, csv[tt][g].w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                qf[2 * tt + (g >> 1)][4 * (g & 1) + j] = (bf16_t)fmaf(a0[4 * g + j] + a1[4 * g + j], rs, fmaf(-mrs, cs[j], bb[j]));
        }
    }

    // ---- the head: S^T = K Q^T, one-pass softmax, O^T = V^T P ---------------------------------------------------------------
    const float sc2 = p.scale * 1.4426950408889634f;
    const int i16 = lane & 15, g16 = (lane >> 4) & 1;
    const char* kt_ = smem + OFF_K;
    const char* vt_ = smem + OFF_V;
    f32x16 s[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
        for (int r16 = 0; r16 < 16; ++r16) s[kt][r16] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(kt_ + sb_r128(32 * kt + lq, 2 * i + h))), qf[i], s[kt], 0, 0, 0);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        const bool edge = 32 * kt + 31 >= p.Lk;
#pragma unroll
        for (int r16 = 0; r16 < 16; ++r16) {
            float v = s[kt][r16] * sc2;
            if (edge && 32 * kt + 8 * (r16 >> 2) + 4 * h + (r16 & 3) >= p.Lk) v = -INFINITY;
            s[kt][r16] = v;
            mx = fmaxf(mx, v);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float psum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r16 = 0; r16 < 16; ++r16) {
            const float e = __builtin_amdgcn_exp2f(s[kt][r16] - mx);
            s[kt][r16] = e;
            psum += e;
        }
    psum += __shfl_xor(psum, 32);
    sb_wait<0>();                                         // the V tile
    __builtin_amdgcn_s_barrier();                          // ... of every wave; and every wave has left the weight tiles
    __builtin_amdgcn_sched_barrier(0);
    f32x16 oacc[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r16 = 0; r16 < 16; ++r16) oacc[d][r16] = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            bf16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)s[kt][8 * sp + j];
            const int base0 = 32 * kt + 16 * sp + 4 * h + (i16 >> 2);
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int col = 32 * d + 16 * g16 + 4 * (i16 & 3);
                const int c = col >> 3, half = (col >> 2) & 1;
                typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
                const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vt_ + sb_v128(base0, c) + 8 * half));
                const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vt_ + sb_v128(base0 + 8, c) + 8 * half));
                bf16x8 vf;
                const bf16x4 b0 = __builtin_bit_cast(bf16x4, v0), b1 = __builtin_bit_cast(bf16x4, v1);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    vf[j] = b0[j];
                    vf[4 + j] = b1[j];
                }
                oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[d], 0, 0, 0);
            }
        }
    }
    // ---- normalise, pack, through the wave's staging image (in the first weight tile's place), whole 128-byte rows out ------
    char* st_ = smem + wave * 4096;
    const float inv = 1.0f / psum;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {     // registers 4 g .. 4 g + 3 = channels 32 d + 8 g + 4 h + (0..3) of query lq
            bf16x4 t;
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = (bf16_t)(oacc[d][4 * g + j] * inv);
            *reinterpret_cast<uint2*>(st_ + sb_r128(lq, 4 * d + g) + 8 * h) = __builtin_bit_cast(uint2, t);
        }
    bf16_t* Op = p.Out + rowbase * p.ldo + hd * 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = q0 + srow + 8 * j;
        const uint4 v = *reinterpret_cast<const uint4*>(st_ + sb_r128(srow + 8 * j, sch));
        if (q < p.N) *reinterpret_cast<uint4*>(Op + (long)q * p.ldo + sch * 8) = v;
    }
}

template <int C>
int sb_launch(const SbArgs& a, hipStream_t stream) {
    static bool attr = false;
    if (!attr) {
        attr = true;
        if (hipFuncSetAttribute((const void*)sra_block_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, SbCfg<C>::LDS) !=
            hipSuccess)
            return EMIP_E_LAUNCH;
    }
    hipLaunchKernelGGL(sra_block_kernel<C>, dim3((unsigned)(a.B * a.tpi)), dim3(256), SbCfg<C>::LDS, stream, a);
    return emip_launch_status();
}

}  // namespace

// O = softmax((LN(X) Wq^T) k^T scale) v, the q projection computed on the fly (one head per workgroup); Wq is the row-swapped
// pack of emip_sra_block.  O [B*N, ldo] must not alias X.
extern "C" int emip_sra_qattn(const void* X, long ldx, const float* stats, float eps, const void* Wq, const float* bq,
                              const float* colsum_q, const void* KV, void* O, long ldo, int B, int N, int Lk, int C, float scale,
                              void* stream) {
    EMIP_REQUIRE(X && stats && Wq && bq && colsum_q && KV && O && O != X && B > 0 && N > 0);
    EMIP_REQUIRE(C == 320 && Lk > 0 && Lk <= 128 && ldx >= C && (ldx & 7) == 0 && ldo >= C && (ldo & 7) == 0);
    EMIP_REQUIRE(aligned16(X) && aligned16(Wq) && aligned16(KV) && aligned16(O) && aligned16(bq) && aligned16(colsum_q) &&
                 (reinterpret_cast<uintptr_t>(stats) & 7u) == 0);
    SbArgs a{};
    a.X = (const bf16_t*)X; a.Out = (bf16_t*)O; a.stats = stats; a.Wq = (const bf16_t*)Wq; a.bq = bq; a.csq = colsum_q;
    a.KV = (const bf16_t*)KV; a.ldx = ldx; a.ldo = ldo; a.B = B; a.N = N; a.Lk = Lk; a.tpi = (N + 127) / 128;
    a.xcd_map = (B % 8) == 0; a.eps = eps; a.scale = scale;
    constexpr int LDS = 2 * SbCfg<320>::SLOT + 32768;
    static bool attr = false;
    if (!attr) {
        attr = true;
        if (hipFuncSetAttribute((const void*)sra_q_kernel<320>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
            return EMIP_E_LAUNCH;
    }
    hipLaunchKernelGGL(sra_q_kernel<320>, dim3((unsigned)(B * a.tpi * 5)), dim3(256), LDS, (hipStream_t)stream, a);
    return emip_launch_status();
}

extern "C" int emip_sra_block_eligible(int C, int Lk) { return (C == 64 || C == 128 || C == 320) && Lk > 0 && Lk <= 128; }

// Out = X + proj(attention(LN(X) Wq^T, KV)); see the head of this file.  Wq / Wp are the bit-swapped packs
// (emip_amd/lib/pvt_v2.py: Block._folded), bq / colsum_q / bp in channel order; out_stats (may be null) receives the row sums
// and sums of squares of Out (stored, not accumulated).  Out may alias X.
extern "C" int emip_sra_block(const void* X, long ldx, const float* stats, float eps, const void* Wq, const float* bq,
                              const float* colsum_q, const void* KV, const void* Wp, const float* bp, void* Out, long ldo,
                              float* out_stats, int B, int N, int Lk, int C, float scale, void* stream) {
    EMIP_REQUIRE(X && stats && Wq && bq && colsum_q && KV && Wp && bp && Out && B > 0 && N > 0);
    EMIP_REQUIRE(emip_sra_block_eligible(C, Lk) && ldx >= C && (ldx & 7) == 0 && ldo >= C && (ldo & 7) == 0);
    EMIP_REQUIRE(aligned16(X) && aligned16(Wq) && aligned16(KV) && aligned16(Wp) && aligned16(Out) && aligned16(bq) &&
                 aligned16(colsum_q) && aligned16(bp) && (reinterpret_cast<uintptr_t>(stats) & 7u) == 0 &&
                 (reinterpret_cast<uintptr_t>(out_stats) & 7u) == 0);
    const long rows = (long)B * N;
    EMIP_REQUIRE(((rows - 1) * ldx + C) * 2 < 0x7FFF0000L && ((rows - 1) * ldo + C) * 2 < 0x7FFF0000L);
    SbArgs a{};
    a.X = (const bf16_t*)X; a.Out = (bf16_t*)Out; a.stats = stats; a.out_stats = out_stats;
    a.Wq = (const bf16_t*)Wq; a.bq = bq; a.csq = colsum_q; a.KV = (const bf16_t*)KV; a.Wp = (const bf16_t*)Wp; a.bp = bp;
    a.ldx = ldx; a.ldo = ldo; a.B = B; a.N = N; a.Lk = Lk; a.tpi = (N + 127) / 128; a.xcd_map = (B % 8) == 0;
    a.eps = eps; a.scale = scale;
    a.x_bytes = (unsigned)(((rows - 1) * ldx + C) * 2);
    a.o_bytes = (unsigned)(((rows - 1) * ldo + C) * 2);
    hipStream_t s = (hipStream_t)stream;
    if (C == 64) return sb_launch<64>(a, s);
    if (C == 128) return sb_launch<128>(a, s);
    return sb_launch<320>(a, s);
}
