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
//   * a workgroup = 4 waves x 32 queries; a wave keeps its 32 query rows as MFMA B-operand fragments for the whole launch;
//   * the keys stream through a 3-slot LDS ring by LDS-DMA (buffer_load ... lds, 16 B per lane, the XOR swizzle of the
//     ds_read_b128 row reads applied on the per-lane SOURCE chunk, rows beyond n read as zeros through the range check), ONE
//     s_barrier and ONE counted s_waitcnt vmcnt per 64-key tile: the score stores of the two previous tiles and the next
//     tile's DMA stay in flight across it (every store is an unconditional buffer store, so the counts are exact);
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
// 8 bytes per lane, unconditional: lanes whose offset lies outside the descriptor's range store nothing
__device__ __forceinline__ void mt_store8(u32x2 v, unsigned voff, i32x4 rs) {
    asm volatile("buffer_store_dwordx2 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(v), "v"(voff), "s"(rs) : "memory");
}

struct MatchArgs {
    const bf16_t* Q;      // [Z][n][ldq]
    const bf16_t* K;      // [Z][n][ldk]
    const float* V;       // [Z][n][2] or null (= the pixel grid)
    bf16_t* S;            // [Zs][n][n] raw scores (scale * q.k) of batches z < Zs, or null
    float* Out;           // [Z][n][2]
    long ldq, ldk, q_bs, k_bs;
    int Z, Zs, n, W, rot, sub, xcd_map, qtiles;
    float scale;
    unsigned k_bytes, s_bytes;
};

constexpr unsigned MT_OOB = 0x80000000u;
constexpr int MT_BK = 64, MT_NST = 3, MT_TILE = MT_BK * 256, MT_NPAD = 2048;
constexpr int MT_RING = MT_NST * MT_TILE;                 // 49 152 B
constexpr int MT_LDS = MT_RING + 3 * MT_NPAD * 2;         // + V^T rows (x, y, 1) for every key: 61 440 B

template <bool SC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void match_kernel(const MatchArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const int q = qt * 128 + wave * 32 + lq;
    const bool q_ok = q < p.n;
    uint4 qf[8];
    {
        const long qr = q_ok ? q : p.n - 1;
#pragma unroll
        for (int i = 0; i < 8; ++i) qf[i] = *reinterpret_cast<const uint4*>(Qp + qr * p.ldq + (2 * i + h) * 8);
    }

    // ---- key tiles: a 1-KB DMA piece = 4 key rows x 16 chunks; lane l sits at row l >> 4, slot l & 15 and fetches source
    // chunk slot ^ (row & 15); wave w moves pieces w, w + 4, w + 8, w + 12 of a tile
    unsigned koff[4];
    int krow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        krow[j] = 4 * (wave + 4 * j) + (lane >> 4);
        koff[j] = (unsigned)((krow[j] * p.ldk + 8 * ((lane & 15) ^ (krow[j] & 15))) * 2);
    }
    const unsigned tile_stride = (unsigned)(MT_BK * p.ldk * 2);
    auto issue = [&](int t) {
        const unsigned base = lds0 + (t % MT_NST) * MT_TILE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned off = (t * MT_BK + krow[j] < p.n) ? koff[j] + (unsigned)t * tile_stride : MT_OOB;
            mt_dma16(base + (wave + 4 * j) * 1024, off, rsK);
        }
    };
    const int ntile = (p.n + MT_BK - 1) / MT_BK;
    issue(0);
    issue(1);

    // ---- V^T for every key: rows x, y, ones (zeros beyond n)
    for (int k = tid; k < MT_NPAD; k += 256) {
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
    }
    __syncthreads();                                       // the table is read by every wave from tile 0 on

    f32x16 oacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
    float m_run = -INFINITY;
    const float sc2 = p.scale * 1.4426950408889634f;       // scores in log2 units
    const unsigned vmask = lq < 3 ? 0xFFFFFFFFu : 0u;      // V^T rows >= 3 are zero
    const int vrow = lq < 3 ? lq : 0;
    const char* vt_row = smem + MT_RING + (vrow * MT_NPAD + 4 * h) * 2;
    const unsigned s_q = q_ok ? (unsigned)q * (unsigned)p.n * 2u : MT_OOB;    // byte offset of this query's score row

    for (int t = 0; t < ntile; ++t) {
        // tile t has landed for this wave once all but the younger operations are done: in issue order behind its 4 DMA
        // pieces sit [8 score stores of tile t - 2] 4 pieces of tile t + 1 [8 score stores of tile t - 1]
        if (t + 1 < ntile) {
            if (SC) {
                if (t == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if (t == 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // last tile: nothing younger was issued that may stay
        }
        __builtin_amdgcn_s_barrier();                      // ... and for every wave; everyone has left tile t - 1's slot
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < ntile) issue(t + 2);                   // into the slot tile t - 1 has just left

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
        // ---- raw correlation out: register 4 g + j of block kt = key t 64 + 32 kt + 8 g + 4 h + j of query q
        if (SC) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int kb = t * MT_BK + 32 * kt + 8 * g + 4 * h;
                    bf16x4 v4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v4[j] = (bf16_t)(s[kt][4 * g + j] * p.scale);
                    const unsigned off = (kb + 3 < p.n) ? s_q + (unsigned)kb * 2u : MT_OOB;
                    mt_store8(__builtin_bit_cast(u32x2, v4), off, rsS);
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
                uint2 a0 = *reinterpret_cast<const uint2*>(vp), a1 = *reinterpret_cast<const uint2*>(vp + 16);
                uint4 vf = make_uint4(a0.x & vmask, a0.y & vmask, a1.x & vmask, a1.y & vmask);
                oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), pf, oacc, 0, 0, 0);
            }
        }
    }

    // ---- lanes of half 0 hold (sum p v_x, sum p v_y, sum p) of their query in registers 0..2
    if (h == 0 && q_ok) {
        const float inv = 1.0f / oacc[2];
        float ox = oacc[0] * inv, oy = oacc[1] * inv;
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
                          long ldq, long ldk, long q_bs, long k_bs, int kv_rot, float scale, int sub_grid, void* stream) {
    EMIP_REQUIRE(Q && K && Out && Z > 0 && Zs >= 0 && Zs <= Z && (Zs == 0 || S));
    EMIP_REQUIRE(n >= 2 * MT_BK && n <= MT_NPAD && (n & 3) == 0 && W > 0 && kv_rot >= 0 && kv_rot < Z);
    EMIP_REQUIRE(ldq >= 128 && ldk >= 128 && (ldq & 7) == 0 && (ldk & 7) == 0 && (q_bs & 7) == 0 && (k_bs & 7) == 0);
    EMIP_REQUIRE(aligned16(Q) && aligned16(K) && (reinterpret_cast<uintptr_t>(Out) & 7u) == 0 &&
                 (reinterpret_cast<uintptr_t>(V) & 7u) == 0 && (reinterpret_cast<uintptr_t>(S) & 7u) == 0);
    EMIP_REQUIRE(((long)(n - 1) * ldk + 128) * 2 < 0x7FFF0000L && (long)n * n * 2 < 0x7FFF0000L);
    EMIP_REQUIRE(q_bs >= (long)(n - 1) * ldq + 128 || Z == 1);
    MatchArgs a{};
    a.Q = (const bf16_t*)Q; a.K = (const bf16_t*)K; a.V = V; a.S = (bf16_t*)S; a.Out = Out;
    a.ldq = ldq; a.ldk = ldk; a.q_bs = q_bs; a.k_bs = k_bs;
    a.Z = Z; a.Zs = Zs; a.n = n; a.W = W; a.rot = kv_rot; a.sub = sub_grid; a.scale = scale;
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
        hipLaunchKernelGGL(match_kernel<true>, grid, dim3(256), MT_LDS, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(match_kernel<false>, grid, dim3(256), MT_LDS, (hipStream_t)stream, a);
    return emip_launch_status();
}
