// GMFlow global matching and flow propagation for gfx950 (bf16 features, D = 128):
//
//   out[z][q] = sum_k softmax_k(scale * <Q[z][q], K[zk][k]>) * v[k]  (- pixel(q))        v[k] in R^2
//
//   * global matching, /root/reference/model/EMIP_short/motion/gmflow/matching.py:8-41: Q / K = the two frames' features
//     (both directions in ONE launch: batch z reads its keys from batch (z + rot) mod Z), v[k] = the pixel grid, the result
//     minus the query's own pixel is the flow, and the raw correlation  scale * <Q, K>  of the forward direction is written
//     once as [z][src q][tgt k] -- the tensor matching.py:18-20 returns permuted and model.py:96 feeds to conv_corr;
//   * flow propagation, gmflow/transformer.py:503-533: v[k] = the flow at key k (f32 [Z][n][2]), no grid subtraction.
//
// The generic attention kernel spent 86 us on the two matching launches of an 8-pair sub-batch: register-staged K / V tiles
// behind index tables, a 32-column V of which 2 columns are used, the score matrix stored 8 bytes per lane from a kernel
// that waits for all of its memory traffic once per tile.  This kernel is built for the shape:
//   * a workgroup = 128 queries x 8 waves: wave w and wave w + 4 hold the SAME 32 query rows (MFMA B-operand fragments, kept
//     for the whole launch) and split the key tiles between them (even / odd), so every SIMD has two waves whose MFMA and
//     softmax phases overlap -- the problem has fewer than 1024 32-query blocks, one wave per SIMD ran at half this speed --
//     and their (max, sum p v, sum p) partials meet in LDS at the end;
//   * the keys stream through a 3-slot LDS ring by LDS-DMA (buffer_load ... lds, 16 B per lane, the XOR swizzle of the
//     ds_read_b128 row reads applied on the per-lane SOURCE chunk, rows beyond n read as zeros through the range check), ONE
//     s_barrier and ONE counted s_waitcnt vmcnt per 64-key tile: the score stores of the two previous tiles and the next
//     tile's DMA stay in flight across it (every store is an unconditional 16-byte buffer store, so the counts are exact);
//   * S^T = K Q^T with the key on the MFMA row: a lane holds scores of ONE query, the row maximum needs one lane exchange,
//     and the exponentiated accumulators are the B operand of  O^T += V^T P  as they stand;
//   * V^T has three live rows -- v_x, v_y and ONES, so the softmax denominator comes out of the same MFMA -- kept for all keys
//     in 12 KB of LDS ([row][key] bf16: a lane's A fragment is two 8-byte reads);
//   * the running maximum is only raised (and the three live accumulator rows rescaled) when some query of the wave needs it.
// Bound by: the forward direction's score stream (7.5 MB per pair) -- HBM writes.
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 mt_rsrc(const void* ptr, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
    return i32x4{(int)(unsigned)a, (int)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void mt_dma16(unsigned lds_dst, unsigned voff, i32x4 rs) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_dst), "v"(voff), "s"(rs)
        : "memory");
}
// 16 bytes per lane, unconditional: lanes whose offset lies outside the descriptor's range store nothing
__device__ __forceinline__ void mt_store16(u32x4 v, unsigned voff, i32x4 rs) {
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(v), "v"(voff), "s"(rs) : "memory");
}

struct MatchArgs {
    const bf16_t* Q;      // [Z][n][ldq]
    const bf16_t* K;      // [Z][n][ldk]
    const float* V;       // [Z][n][2] or null (= the pixel grid)
    bf16_t* S;            // [Zs][n][n] raw scores (scale * q.k) of batches z < Zs, or null
    float* Out;           // [Z][n][2]
    float* lse;           // [Z][n] log2-sum-exp of the scaled scores (training forward) or null
    long ldq, ldk, q_bs, k_bs;
    int Z, Zs, n, W, rot, sub, xcd_map, qtiles;
    float scale;
    unsigned k_bytes, s_bytes;
};

constexpr unsigned MT_OOB = 0x80000000u;
constexpr int MT_BK = 64, MT_NST = 6, MT_TILE = MT_BK * 256, MT_NPAD = 2048;
constexpr int MT_RING = MT_NST * MT_TILE;                 // 3 rounds of 2 tiles: 98 304 B
constexpr int MT_LDS = MT_RING + 4 * MT_NPAD * 2;         // + V^T rows (x, y, 1, 0) for every key: 114 688 B

template <bool SC>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void match_kernel(const MatchArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qw = wave & 3, half = wave >> 2;             // query block of the wave; which tile of a round it multiplies
    const int lq = lane & 31, h = lane >> 5;
    int z, qt;
    if (p.xcd_map) {        // ids are dealt round-robin over the 8 XCDs: the query tiles of a batch element share its keys in one L2
        const int j = blockIdx.x >> 3;
        qt = j % p.qtiles;
        z = (blockIdx.x & 7) + 8 * (j / p.qtiles);
    } else {
        qt = blockIdx.x % p.qtiles;
        z = blockIdx.x / p.qtiles;
    }
    int zk = z + p.rot;
    if (zk >= p.Z) zk -= p.Z;
    const bf16_t* __restrict__ Qp = p.Q + (long)z * p.q_bs;
    const i32x4 rsK = mt_rsrc(p.K + (long)zk * p.k_bs, p.k_bytes);
    const bool sc_on = SC && z < p.Zs;                     // workgroup-uniform
    const i32x4 rsS = mt_rsrc(sc_on ? p.S + (long)z * p.n * p.n : p.S, sc_on ? p.s_bytes : 0u);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)smem;
    bf16_t* vt = reinterpret_cast<bf16_t*>(smem + MT_RING);

    // ---- this lane's query row as B-operand fragments (k-step i: channels 16 i + 8 h .. + 7)
    const int q = qt * 128 + qw * 32 + lq;
    const bool q_ok = q < p.n;
    uint4 qf[8];
    {
        const long qr = q_ok ? q : p.n - 1;
#pragma unroll
        for (int i = 0; i < 8; ++i) qf[i] = *reinterpret_cast<const uint4*>(Qp + qr * p.ldq + (2 * i + h) * 8);
    }

    // ---- key tiles, two per round: a 1-KB DMA piece = 4 key rows x 16 chunks; lane l sits at row l >> 4, slot l & 15 and
    // fetches source chunk slot ^ (row & 15); wave w moves pieces w, w + 8, w + 16, w + 24 of a round's 32 (128 key rows)
    unsigned koff[4];
    int krow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        krow[j] = 4 * (wave + 8 * j) + (lane >> 4);        // key row inside the round
        koff[j] = (unsigned)((krow[j] * p.ldk + 8 * ((lane & 15) ^ (krow[j] & 15))) * 2);
    }
    const unsigned round_stride = (unsigned)(2 * MT_BK * p.ldk * 2);
    auto issue = [&](int r) {                              // round r = tiles 2 r, 2 r + 1 -> ring slots (2 r) % 6, + 1
        const unsigned base = lds0 + ((2 * r) % MT_NST) * MT_TILE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned off = (r * 2 * MT_BK + krow[j] < p.n) ? koff[j] + (unsigned)r * round_stride : MT_OOB;
            mt_dma16(base + (wave + 8 * j) * 1024, off, rsK);
        }
    };
    const int ntile = (p.n + MT_BK - 1) / MT_BK, nround = (ntile + 1) >> 1;
    issue(0);
    if (nround > 1) issue(1);

    // ---- V^T for every key: rows x, y, ones (zeros beyond n)
    for (int k = tid; k < MT_NPAD; k += 512) {
        float x = 0.f, y = 0.f, o = 0.f;
        if (k < p.n) {
            if (p.V) {
                const float2 f = *reinterpret_cast<const float2*>(p.V + ((long)zk * p.n + k) * 2);
                x = f.x; y = f.y;
            } else {
                y = (float)(k / p.W);
                x = (float)(k - (k / p.W) * p.W);
            }
            o = 1.f;
        }
        vt[k] = (bf16_t)x;
        vt[MT_NPAD + k] = (bf16_t)y;
        vt[2 * MT_NPAD + k] = (bf16_t)o;
        vt[3 * MT_NPAD + k] = (bf16_t)0.f;                 // the row every lane >= 3 reads: V^T rows 3..31 are zero
    }
    __syncthreads();                                       // the table is read by every wave from tile 0 on

    f32x16 oacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
    float m_run = -INFINITY;
    const float sc2 = p.scale * 1.4426950408889634f;       // scores in log2 units
    const int vrow = lq < 3 ? lq : 3;
    const char* vt_row = smem + MT_RING + (vrow * MT_NPAD + 4 * h) * 2;
    const unsigned s_q = q_ok ? (unsigned)q * (unsigned)p.n * 2u : MT_OOB;    // byte offset of this query's score row

    for (int r = 0; r < nround; ++r) {
        // round r has landed for this wave once all but the younger operations are done: in issue order behind its 4 DMA
        // pieces sit [4 score stores of round r - 2] 4 pieces of round r + 1 [4 score stores of round r - 1]
        if (r + 1 < nround) {
            if (SC) {
                if (r == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if (r == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // last round: nothing younger was issued that may stay
        }
        __builtin_amdgcn_s_barrier();                      // ... and for every wave; everyone has left round r - 1's slots
        __builtin_amdgcn_sched_barrier(0);
        if (r + 2 < nround) issue(r + 2);                  // into the slots round r - 1 has just left
        const int t = 2 * r + half;                        // this wave's tile of the round
        if (t >= ntile) {                                  // (odd tile count: the last round has one tile)
            if (SC) {                                      // keep the per-round store count of the waits above
#pragma unroll
                for (int i = 0; i < 4; ++i) mt_store16(u32x4{0u, 0u, 0u, 0u}, MT_OOB, rsS);
            }
            continue;
        }

        const char* sb = smem + (t % MT_NST) * MT_TILE;
        // ---- S^T = K Q^T for the tile's two 32-key blocks
        f32x16 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
            const int row = 32 * kt + lq;
            const char* rp = sb + row * 256;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint4 kf = *reinterpret_cast<const uint4*>(rp + (((2 * i + h) ^ (row & 15)) * 16));
                s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[i]),
                                                                s[kt], 0, 0, 0);
            }
        }
        // ---- raw correlation out: register 4 g + j of block kt = key t 64 + 32 kt + 8 g + 4 h + j of query q.  The two lanes of a
        // query trade halves (v_permlane32_swap) so that each stores 16 contiguous bytes: lane half 0 the 8 keys of group g,
        // lane half 1 those of group g + 1
        if (SC) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    bf16x4 va, vb;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        va[j] = (bf16_t)(s[kt][4 * g + j] * p.scale);
                        vb[j] = (bf16_t)(s[kt][4 * g + 4 + j] * p.scale);
                    }
                    u32x2 a2 = __builtin_bit_cast(u32x2, va), b2 = __builtin_bit_cast(u32x2, vb);
                    const auto r0 = __builtin_amdgcn_permlane32_swap(a2.x, b2.x, false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(a2.y, b2.y, false, false);
                    const int kb = t * MT_BK + 32 * kt + 8 * (g + h);
                    const unsigned off = (kb + 7 < p.n) ? s_q + (unsigned)kb * 2u : MT_OOB;
                    mt_store16(u32x4{r0[0], r1[0], r0[1], r1[1]}, off, rsS);
                }
        }
        // ---- softmax: keys beyond n (last tile only) out, running maximum raised only when some query needs it
        if (t == ntile - 1 && (p.n & (MT_BK - 1))) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t * MT_BK + 32 * kt + 8 * (r >> 2) + 4 * h + (r & 3) >= p.n) s[kt][r] = -INFINITY;
        }
        float tm = s[0][0];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) tm = fmaxf(tm, s[kt][r]);
        tm = fmaxf(tm, __shfl_xor(tm, 32)) * sc2;
        if (!__all(tm <= m_run)) {
            const float m_new = fmaxf(m_run, tm);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            oacc[0] *= alpha; oacc[1] *= alpha; oacc[2] *= alpha;      // the three live V^T rows (registers 0..2 of lane half 0)
            m_run = m_new;
        }
        // ---- O^T += V^T P
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                bf16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)__builtin_amdgcn_exp2f(fmaf(s[kt][8 * sp + j], sc2, -m_run));
                // A fragment: element j = V^T[row lq][key t 64 + 32 kt + 16 sp + 8 (j >> 2) + 4 h + (j & 3)]
                const char* vp = vt_row + (t * MT_BK + 32 * kt + 16 * sp) * 2;
                const uint2 a0 = *reinterpret_cast<const uint2*>(vp), a1 = *reinterpret_cast<const uint2*>(vp + 16);
                const uint4 vf = make_uint4(a0.x, a0.y, a1.x, a1.y);
                oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), pf, oacc, 0, 0, 0);
            }
        }
    }

    // ---- lanes of lane half 0 hold (sum p v_x, sum p v_y, sum p) of their query in registers 0..2, relative to m_run; the odd
    // tiles' wave hands its four numbers to the even tiles' wave through the (drained) ring
    __syncthreads();
    float4* part = reinterpret_cast<float4*>(smem);
    if (half == 1 && h == 0) part[qw * 32 + lq] = make_float4(oacc[0], oacc[1], oacc[2], m_run);
    __syncthreads();
    if (half == 0 && h == 0 && q_ok) {
        const float4 o2 = part[qw * 32 + lq];
        const float m = fmaxf(m_run, o2.w);                // (o2.w = -inf with an all-zero partial when there was a single tile)
        const float fa = __builtin_amdgcn_exp2f(m_run - m), fb = __builtin_amdgcn_exp2f(o2.w - m);
        const float l_tot = oacc[2] * fa + o2.z * fb;
        const float inv = 1.0f / l_tot;
        if (p.lse) p.lse[(long)z * p.n + q] = m + __log2f(l_tot);
        float ox = (oacc[0] * fa + o2.x * fb) * inv, oy = (oacc[1] * fa + o2.y * fb) * inv;
        if (p.sub) {
            const int qy = q / p.W;
            ox -= (float)(q - qy * p.W);
            oy -= (float)qy;
        }
        *reinterpret_cast<float2*>(p.Out + ((long)z * p.n + q) * 2) = make_float2(ox, oy);
    }
}

}  // namespace

// Q, K: bf16 [Z][n][128] (row strides ldq / ldk, batch strides q_bs / k_bs, in elements); the keys of batch z are those of
// batch (z + kv_rot) mod Z.  V: f32 [Z][n][2] (indexed like the keys), or NULL = the pixel grid (x = k mod W, y = k / W).
// S: bf16 [Zs][n][n] receives scale * q.k for batches z < Zs (may be NULL with Zs = 0).  Out: f32 [Z][n][2] =
// softmax-weighted mean of V, minus the query's own pixel when sub_grid.
extern "C" int emip_match(const void* Q, const void* K, const float* V, void* S, float* Out, int Z, int Zs, int n, int W,
                          long ldq, long ldk, long q_bs, long k_bs, int kv_rot, float scale, int sub_grid, float* lse, void* stream) {
    EMIP_REQUIRE(Q && K && Out && Z > 0 && Zs >= 0 && Zs <= Z && (Zs == 0 || S));
    EMIP_REQUIRE(n >= 2 * MT_BK && n <= MT_NPAD && (n & 7) == 0 && W > 0 && kv_rot >= 0 && kv_rot < Z);
    EMIP_REQUIRE(ldq >= 128 && ldk >= 128 && (ldq & 7) == 0 && (ldk & 7) == 0 && (q_bs & 7) == 0 && (k_bs & 7) == 0);
    EMIP_REQUIRE(aligned16(Q) && aligned16(K) && (reinterpret_cast<uintptr_t>(Out) & 7u) == 0 &&
                 (reinterpret_cast<uintptr_t>(V) & 7u) == 0 && aligned16(S));
    EMIP_REQUIRE(((long)(n - 1) * ldk + 128) * 2 < 0x7FFF0000L && (long)n * n * 2 < 0x7FFF0000L);
    EMIP_REQUIRE(q_bs >= (long)(n - 1) * ldq + 128 || Z == 1);
    MatchArgs a{};
    a.Q = (const bf16_t*)Q; a.K = (const bf16_t*)K; a.V = V; a.S = (bf16_t*)S; a.Out = Out;
    a.ldq = ldq; a.ldk = ldk; a.q_bs = q_bs; a.k_bs = k_bs;
    a.lse = lse; a.Z = Z; a.Zs = Zs; a.n = n; a.W = W; a.rot = kv_rot; a.sub = sub_grid; a.scale = scale;
    a.qtiles = (n + 127) / 128;
    a.xcd_map = (Z % 8) == 0;
    a.k_bytes = (unsigned)(((long)(n - 1) * ldk + 128) * 2);
    a.s_bytes = (unsigned)((long)n * n * 2);
    static bool attr = false;
    if (!attr) {
        attr = true;
        if (hipFuncSetAttribute((const void*)match_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, MT_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)match_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, MT_LDS) != hipSuccess)
            return EMIP_E_LAUNCH;
    }
    const dim3 grid((unsigned)(Z * a.qtiles));
    if (Zs > 0)
        hipLaunchKernelGGL(match_kernel<true>, grid, dim3(512), MT_LDS, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(match_kernel<false>, grid, dim3(512), MT_LDS, (hipStream_t)stream, a);
    return emip_launch_status();
}
