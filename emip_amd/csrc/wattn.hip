// GMFlow split-window attention for gfx950 (bf16, single head, D = DV = 128):
//
//     O[rows[win][q]] = softmax_k( scale <Q[rows[win][q]], K[rows[win][k]]> - 100 [gid[win][q] != gid[win][k]] ) V[rows[win][k]]
//
// /root/reference/model/EMIP_short/motion/gmflow/transformer.py:46-105 (single_head_split_window_attention): the 44 x 44
// feature map is cut into 2 x 2 windows of 484 tokens, every second layer after a roll by half a window with the additive
// -100 mask of :19-43; self attention reads K / V of the same frame, cross attention those of the other frame of the pair
// (batch element (b + kv_rot) mod B).  Roll, split, merge and roll-back are the index table `rows` (window-local token ->
// token of the frame), the mask is the region-id table `gid`.
//
// The generic attention kernel took 61 us per launch at 32 images (register-staged K / V tiles behind per-tile table look-ups,
// four waves per workgroup = one per SIMD: its load -> S -> softmax -> PV chain ran un-overlapped).  Here:
//   * a workgroup = 256 queries of one window = 8 waves x 32 queries (two waves per SIMD: one's MFMA phases under the
//     other's softmax), query rows kept as MFMA B-operand fragments for the whole launch;
//   * the window's index and region tables go to LDS once; K and V tiles of 64 keys stream through a 3-slot LDS ring by
//     LDS-DMA with per-lane source rows from that table (the ds_read_b128 / ds_read_tr16_b64 swizzles applied on the per-lane
//     SOURCE chunk), ONE s_barrier and one counted s_waitcnt vmcnt per tile;
//   * S^T = K Q^T with the key on the MFMA row (a lane holds scores of one query; one lane exchange per row statistic), the
//     exponentiated accumulators are the B operand of O^T += V^T P as they stand, V^T fragments by ds_read_tr16_b64;
//   * the running maximum is raised -- and the 64 output accumulators rescaled -- only when some query of the wave needs it.
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 wa_rsrc(const void* ptr, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
    return i32x4{(int)(unsigned)a, (int)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void wa_dma16(unsigned lds_dst, unsigned voff, i32x4 rs) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_dst), "v"(voff), "s"(rs)
        : "memory");
}

struct WaArgs {
    const bf16_t* Q;
    const bf16_t* K;
    const bf16_t* V;
    bf16_t* O;
    float* lse;            // [B][tokens] log2-sum-exp of the scaled, masked scores (training forward) or null
    const bf16_t* Wm;      // MERGE: merge.weight [128][128] in fragment order (ops.wattn_merge_pack)
    const bf16_t* Wq;      // QP: q_proj.weight [128][128] in fragment order, rows bit-swapped (ops.wattn_q_pack); Q = the token rows
    const float* gamma;    // MERGE: norm1 weight / bias, f32 [128]
    const float* beta;
    const bf16_t* Res;     // MERGE: residual tokens (may alias O) or null
    long ldr, r_bs;
    float eps;
    const int* rows;       // [nwin][L] token of the frame for every window-local token
    const int* gid;        // [nwin][L] region ids (shifted windows) or null
    long ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs;
    int B, nwin, L, rot, qblocks, tokens;
    float scale;
    unsigned k_bytes, v_bytes;
#ifdef EMIP_TUNING
    int skip;      // calibration build only: 1 no S MFMAs, 2 no softmax, 4 no PV, 8 no DMA, 16 no V reads (PV MFMAs on stale registers)
#endif
};
#ifdef EMIP_TUNING
#define WA_SKIP(bit) (p.skip & (bit))
#else
#define WA_SKIP(bit) false
#endif

constexpr unsigned WA_OOB = 0x80000000u;
constexpr int WA_BK = 64, WA_NST = 3, WA_KT = WA_BK * 256, WA_STAGE = 2 * WA_KT, WA_LMAX = 512;
constexpr int WA_RING = WA_NST * WA_STAGE;                 // 98 304 B
constexpr int WA_LDS = WA_RING + 2 * WA_LMAX * 4;          // + the window's row and region tables: 102 400 B
constexpr int WA_WM = 32 * 1024, WA_LDS_MERGE = WA_LDS + WA_WM + 2 * 128 * 4;      // + the merge weight (32 fragment pieces), gamma, beta

__device__ __forceinline__ int wa_voff(int row, int c) { return row * 256 + ((c ^ ((row & 3) << 2)) * 16); }

template <bool MASK, bool MERGE = false, bool QP = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void wattn_kernel(const WaArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, h = lane >> 5;
    const int qb = blockIdx.x, win = blockIdx.y;
    const long b = blockIdx.z;
    long bk = b + p.rot;
    if (bk >= p.B) bk -= p.B;
    const i32x4 rsK = wa_rsrc(p.K + bk * p.k_bs, p.k_bytes), rsV = wa_rsrc(p.V + bk * p.v_bs, p.v_bytes);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)smem;
    int* trow = reinterpret_cast<int*>(smem + WA_RING);
    int* tgid = trow + WA_LMAX;
    const int* rows = p.rows + (long)win * p.L;
    for (int i = tid; i < WA_LMAX; i += 512) {
        trow[i] = i < p.L ? rows[i] : 0;
        tgid[i] = (MASK && i < p.L) ? p.gid[(long)win * p.L + i] : 0;
    }

    if (MERGE) {
        // the merge weight (32 KB, fragment order) and the norm vectors wait behind the tables for the epilogue; issued first, so
        // every later counted wait covers these pieces too
        const i32x4 rsM = wa_rsrc(p.Wm, 32768u);
#pragma unroll
        for (int j = 0; j < 4; ++j) wa_dma16(lds0 + WA_LDS + (4 * wave + j) * 1024, (unsigned)((4 * wave + j) * 1024 + lane * 16), rsM);
        float* tg = reinterpret_cast<float*>(smem + WA_LDS + WA_WM);
        if (tid < 256) tg[tid] = tid < 128 ? p.gamma[tid] : p.beta[tid - 128];
    }
    // ---- this lane's query: its row of the frame, its region, its fragments (k-step i: channels 16 i + 8 h .. + 7)
    const int q = qb * 256 + wave * 32 + lq;
    const bool q_ok = q < p.L;
    const int qrow = rows[q_ok ? q : 0];
    const int q_g = MASK ? p.gid[(long)win * p.L + (q_ok ? q : 0)] : 0;
    uint4 qf[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) qf[i] = *reinterpret_cast<const uint4*>(p.Q + b * p.q_bs + (long)qrow * p.ldq + (2 * i + h) * 8);
    __syncthreads();                                        // the tables are in LDS

    // ---- K / V tiles: a 1-KB DMA piece = 4 key rows x 16 chunks; lane l sits at row l >> 4, slot l & 15; wave w moves pieces
    // 2 w, 2 w + 1 of the K tile and of the V tile.  K: source chunk slot ^ (row & 15); V: slot ^ ((row & 3) << 2)
    auto issue = [&](int t) {
        const unsigned base = lds0 + (t % WA_NST) * WA_STAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = 4 * (2 * wave + j) + (lane >> 4);             // key row inside the tile
            const int key = t * WA_BK + r;
            const bool ok = key < p.L;
            const int grow = trow[ok ? key : 0];
            const unsigned ko = ok ? (unsigned)((grow * p.ldk + 8 * ((lane & 15) ^ (r & 15))) * 2) : WA_OOB;
            const unsigned vo = ok ? (unsigned)((grow * p.ldv + 8 * ((lane & 15) ^ ((r & 3) << 2))) * 2) : WA_OOB;
            wa_dma16(base + (2 * wave + j) * 1024, ko, rsK);
            wa_dma16(base + WA_KT + (2 * wave + j) * 1024, vo, rsV);
        }
    };
    const int ntile = (p.L + WA_BK - 1) / WA_BK;
    if (QP) {
        // the q projection weight rides in ring slot 2 (free until the first loop barrier has passed, i.e. until every wave has
        // left this prologue): 32 fragment pieces, 4 per wave, issued ahead of tiles 0 and 1
        const i32x4 rsQ = wa_rsrc(p.Wq, 32768u);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            wa_dma16(lds0 + 2 * WA_STAGE + (4 * wave + j) * 1024, (unsigned)((4 * wave + j) * 1024 + lane * 16), rsQ);
    }
    issue(0);
    if (ntile > 1) issue(1);
    if (QP) {
        // Q^T = Wq X^T with the query on the lane; the weight rows are stored with bits 2 and 3 of their index swapped inside every
        // 16, so registers 8 sp .. 8 sp + 7 of tile dd hold channels 32 dd + 16 sp + 8 h .. + 7: the B fragment 2 dd + sp of S^T
        if (ntile > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // all but the pieces of tiles 0 and 1
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const char* wq = smem + 2 * WA_STAGE;
        uint4 xq[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) xq[i] = qf[i];
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
            f32x16 qa;
#pragma unroll
            for (int r = 0; r < 16; ++r) qa[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const uint4 wf = *reinterpret_cast<const uint4*>(wq + (dd * 8 + ks) * 1024 + lane * 16);
                qa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, xq[ks]), qa, 0, 0, 0);
            }
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                bf16x8 t;
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = (bf16_t)qa[8 * sp + j];
                qf[2 * dd + sp] = __builtin_bit_cast(uint4, t);
            }
        }
    }

    f32x16 oacc[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sc2 = p.scale * 1.4426950408889634f;       // scores in log2 units
    const float mask2 = -100.0f * 1.4426950408889634f;
    const int i16 = lane & 15, g16 = (lane >> 4) & 1;

    for (int t = 0; t < ntile; ++t) {
        // tile t has landed once all but the 4 pieces of tile t + 1 are done
        if (t + 1 < ntile) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // ... for every wave; everyone has left tile t - 1's slot
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < ntile && !WA_SKIP(8)) issue(t + 2);

        const char* kt_ = smem + (t % WA_NST) * WA_STAGE;
        const char* vt_ = kt_ + WA_KT;
        // ---- S^T = K Q^T
        f32x16 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
            const int row = 32 * kt + lq;
            const char* rp = kt_ + row * 256;
            if (!WA_SKIP(1))
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint4 kf = *reinterpret_cast<const uint4*>(rp + (((2 * i + h) ^ (row & 15)) * 16));
                s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[i]),
                                                                s[kt], 0, 0, 0);
            }
        }
        // ---- scale, mask (register 4 g + j of block kt = key t 64 + 32 kt + 8 g + 4 h + j), keys beyond L out
        if (!WA_SKIP(2)) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int kb = t * WA_BK + 32 * kt + 8 * g + 4 * h;
                int4 kg = make_int4(0, 0, 0, 0);
                if (MASK) kg = *reinterpret_cast<const int4*>(tgid + kb);
                const int kgv[4] = {kg.x, kg.y, kg.z, kg.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = s[kt][4 * g + j] * sc2;
                    if (MASK) v += (kgv[j] != q_g) ? mask2 : 0.f;
                    s[kt][4 * g + j] = v;
                }
            }
        if (t == ntile - 1 && (p.L & (WA_BK - 1))) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t * WA_BK + 32 * kt + 8 * (r >> 2) + 4 * h + (r & 3) >= p.L) s[kt][r] = -INFINITY;
        }
        float tm = s[0][0];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) tm = fmaxf(tm, s[kt][r]);
        tm = fmaxf(tm, __shfl_xor(tm, 32));
        if (!__all(tm <= m_run)) {
            const float m_new = fmaxf(m_run, tm);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
            m_run = m_new;
        }
        float psum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(s[kt][r] - m_run);
                s[kt][r] = e;
                psum += e;
            }
        l_run += psum;
        }
        // ---- O^T += V^T P
        if (!WA_SKIP(4))
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                bf16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)s[kt][8 * sp + j];
                const int base0 = 32 * kt + 16 * sp + 4 * h + (i16 >> 2);
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int col = 32 * d + 16 * g16 + 4 * (i16 & 3);
                    const int c = col >> 3, half = (col >> 2) & 1;
                    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
                    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vt_ + wa_voff(base0, c) + 8 * half));
                    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vt_ + wa_voff(base0 + 8, c) + 8 * half));
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

    // ---- normalise and store: registers 4 g .. 4 g + 3 of block d = channels 32 d + 8 g + 4 h + (0..3) of this lane's query
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    if (MERGE) {
        // ---- out = res + LayerNorm(merge(attention)) * gamma + beta (transformer.py:330-338) without the attention output in
        // memory: the normalised O accumulators, rounded to bf16, are the B operand of M^T = Wm O^T (registers 8 sp .. 8 sp + 7 of
        // tile d = channels 32 d + 16 sp + 8 (j >> 2) + 4 h + (j & 3): the order the weight pack follows)
        bf16x8 of[8];
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp)
#pragma unroll
                for (int j = 0; j < 8; ++j) of[2 * d + sp][j] = (bf16_t)(oacc[d][8 * sp + j] * inv);
        const char* wm = smem + WA_LDS;
        f32x16 macc[4];
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
#pragma unroll
            for (int r = 0; r < 16; ++r) macc[dd][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const uint4 wf = *reinterpret_cast<const uint4*>(wm + (dd * 8 + ks) * 1024 + lane * 16);
                macc[dd] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf), of[ks], macc[dd], 0, 0, 0);
            }
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int dd = 0; dd < 4; ++dd)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s1 += macc[dd][r];
                s2 = fmaf(macc[dd][r], macc[dd][r], s2);
            }
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        const float mean = s1 * (1.0f / 128), rstd = rsqrtf(fmaxf(s2 * (1.0f / 128) - mean * mean, 0.f) + p.eps);
        if (q_ok) {
            const float* tg = reinterpret_cast<const float*>(smem + WA_LDS + WA_WM);
            const bf16_t* rp = p.Res ? p.Res + b * p.r_bs + (long)qrow * p.ldr : nullptr;
            bf16_t* Op = p.O + b * p.o_bs + (long)qrow * p.ldo;
#pragma unroll
            for (int dd = 0; dd < 4; ++dd)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = 32 * dd + 8 * g + 4 * h;
                    const float4 gm = *reinterpret_cast<const float4*>(tg + ch), bt = *reinterpret_cast<const float4*>(tg + 128 + ch);
                    const float gv[4] = {gm.x, gm.y, gm.z, gm.w}, bv[4] = {bt.x, bt.y, bt.z, bt.w};
                    float rv[4] = {0.f, 0.f, 0.f, 0.f};
                    if (rp) {
                        const bf16x4 r4 = *reinterpret_cast<const bf16x4*>(rp + ch);
#pragma unroll
                        for (int j = 0; j < 4; ++j) rv[j] = (float)r4[j];
                    }
                    bf16x4 ov;
#pragma unroll
                    for (int j = 0; j < 4; ++j) ov[j] = (bf16_t)(fmaf((macc[dd][4 * g + j] - mean) * rstd, gv[j], bv[j]) + rv[j]);
                    *reinterpret_cast<bf16x4*>(Op + ch) = ov;
                }
        }
        return;
    }
    if (q_ok) {
        if (p.lse && h == 0) p.lse[b * p.tokens + qrow] = m_run + __log2f(l_tot);      // what emip_window_attention_bwd rebuilds P from
        bf16_t* Op = p.O + b * p.o_bs + (long)qrow * p.ldo;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 ov;
#pragma unroll
                for (int j = 0; j < 4; ++j) ov[j] = (bf16_t)(oacc[d][4 * g + j] * inv);
                *reinterpret_cast<bf16x4*>(Op + 32 * d + 8 * g + 4 * h) = ov;
            }
    }
}

}  // namespace

#ifdef EMIP_TUNING
static int g_wa_skip = 0;
extern "C" int emip_debug_set_wa(int flags) { g_wa_skip = flags; return 0; }
#endif

// Q, K, V, O: bf16 token matrices of B frames (batch strides q_bs ... o_bs, row strides ldq ... ldo, in elements; 128
// channels at the pointer); rows: int [nwin][L] frame token of every window-local token, gid: int [nwin][L] region ids of the
// shifted layout or NULL (no mask); keys / values of frame b are read from frame (b + kv_rot) mod B.  L <= 512.  lse: NULL or
// f32 [B][tokens], receives log2(sum_k exp2(score_k log2 e)) of every query (the training forward; emip_window_attention_bwd).
extern "C" int emip_window_attention(const void* Q, const void* K, const void* V, void* O, int B, int nwin, int L, long ldq,
                                     long ldk, long ldv, long ldo, long q_bs, long k_bs, long v_bs, long o_bs, const int* rows,
                                     const int* gid, int tokens, int kv_rot, float scale, float* lse, void* stream) {
    EMIP_REQUIRE(Q && K && V && O && rows && B > 0 && nwin > 0 && L >= WA_BK && L <= WA_LMAX && tokens >= L);
    EMIP_REQUIRE(B < 65536 && nwin < 65536 && kv_rot >= 0 && kv_rot < B);
    EMIP_REQUIRE(ldq >= 128 && ldk >= 128 && ldv >= 128 && ldo >= 128 && ((ldq | ldk | ldv) & 7) == 0 && (ldo & 3) == 0);
    EMIP_REQUIRE(((q_bs | k_bs | v_bs) & 7) == 0 && (o_bs & 3) == 0);
    EMIP_REQUIRE(aligned16(Q) && aligned16(K) && aligned16(V) && (reinterpret_cast<uintptr_t>(O) & 7u) == 0);
    EMIP_REQUIRE(((long)(tokens - 1) * ldk + 128) * 2 < 0x7FFF0000L && ((long)(tokens - 1) * ldv + 128) * 2 < 0x7FFF0000L);
    WaArgs a{};
    a.Q = (const bf16_t*)Q; a.K = (const bf16_t*)K; a.V = (const bf16_t*)V; a.O = (bf16_t*)O; a.lse = lse; a.tokens = tokens; a.rows = rows; a.gid = gid;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.q_bs = q_bs; a.k_bs = k_bs; a.v_bs = v_bs; a.o_bs = o_bs;
    a.B = B; a.nwin = nwin; a.L = L; a.rot = kv_rot; a.scale = scale;
    a.qblocks = (L + 255) / 256;
    a.k_bytes = (unsigned)(((long)(tokens - 1) * ldk + 128) * 2);
    a.v_bytes = (unsigned)(((long)(tokens - 1) * ldv + 128) * 2);
#ifdef EMIP_TUNING
    a.skip = g_wa_skip;
#endif
    static bool attr = false;
    if (!attr) {
        attr = true;
        if (hipFuncSetAttribute((const void*)wattn_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, WA_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)wattn_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, WA_LDS) != hipSuccess)
            return EMIP_E_LAUNCH;
    }
    const dim3 grid((unsigned)a.qblocks, (unsigned)nwin, (unsigned)B);
    if (gid)
        hipLaunchKernelGGL(wattn_kernel<true>, grid, dim3(512), WA_LDS, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(wattn_kernel<false>, grid, dim3(512), WA_LDS, (hipStream_t)stream, a);
    return emip_launch_status();
}

// The same followed, in the launch, by the layer's merge Linear, norm1 and the residual (gmflow/transformer.py:330-338: message =
// norm1(merge(attention)), then `source + message` for the layer without FFN): O = Res + LayerNorm(attention Wm^T) * gamma + beta.
// Wm: merge.weight [128][128] in fragment order (emip_amd.ops.wattn_merge_pack); Res: NULL or the residual tokens (row stride ldr,
// batch stride r_bs; may alias O when Q is not a view of O's rows).  Wq: NULL, or q_proj.weight [128][128] in fragment order with
// bit-swapped rows (emip_amd.ops.wattn_q_pack): Q then points at the TOKEN rows and the q projection runs in the launch's prologue
// (a wave reads the rows of its own 32 queries only, so Q may be O's buffer).
extern "C" int emip_window_attention_merge(const void* Q, const void* K, const void* V, void* O, int B, int nwin, int L, long ldq,
                                           long ldk, long ldv, long ldo, long q_bs, long k_bs, long v_bs, long o_bs,
                                           const int* rows, const int* gid, int tokens, int kv_rot, float scale, const void* Wm,
                                           const float* gamma, const float* beta, float eps, const void* Res, long ldr, long r_bs,
                                           const void* Wq, void* stream) {
    EMIP_REQUIRE(Q && K && V && O && rows && Wm && gamma && beta && eps > 0.f && B > 0 && nwin > 0 && L >= WA_BK && L <= WA_LMAX && tokens >= L);
    EMIP_REQUIRE(B < 65536 && nwin < 65536 && kv_rot >= 0 && kv_rot < B);
    EMIP_REQUIRE(ldq >= 128 && ldk >= 128 && ldv >= 128 && ldo >= 128 && ((ldq | ldk | ldv) & 7) == 0 && (ldo & 3) == 0);
    EMIP_REQUIRE(((q_bs | k_bs | v_bs) & 7) == 0 && (o_bs & 3) == 0 && (!Res || (ldr >= 128 && ((ldr | r_bs) & 3) == 0)));
    EMIP_REQUIRE(aligned16(Q) && aligned16(K) && aligned16(V) && aligned16(Wm) && (reinterpret_cast<uintptr_t>(O) & 7u) == 0 &&
                 (!Res || (reinterpret_cast<uintptr_t>(Res) & 7u) == 0));
    EMIP_REQUIRE(((long)(tokens - 1) * ldk + 128) * 2 < 0x7FFF0000L && ((long)(tokens - 1) * ldv + 128) * 2 < 0x7FFF0000L);
    WaArgs a{};
    a.Q = (const bf16_t*)Q; a.K = (const bf16_t*)K; a.V = (const bf16_t*)V; a.O = (bf16_t*)O; a.lse = nullptr; a.tokens = tokens;
    a.rows = rows; a.gid = gid;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.q_bs = q_bs; a.k_bs = k_bs; a.v_bs = v_bs; a.o_bs = o_bs;
    a.B = B; a.nwin = nwin; a.L = L; a.rot = kv_rot; a.scale = scale;
    a.qblocks = (L + 255) / 256;
    a.k_bytes = (unsigned)(((long)(tokens - 1) * ldk + 128) * 2);
    a.v_bytes = (unsigned)(((long)(tokens - 1) * ldv + 128) * 2);
    a.Wq = (const bf16_t*)Wq; a.Wm = (const bf16_t*)Wm; a.gamma = gamma; a.beta = beta; a.eps = eps; a.Res = (const bf16_t*)Res; a.ldr = ldr; a.r_bs = r_bs;
#ifdef EMIP_TUNING
    a.skip = 0;
#endif
    static bool attr = false;
    if (!attr) {
        attr = true;
        if (hipFuncSetAttribute((const void*)wattn_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, WA_LDS_MERGE) != hipSuccess ||
            hipFuncSetAttribute((const void*)wattn_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, WA_LDS_MERGE) != hipSuccess ||
            hipFuncSetAttribute((const void*)wattn_kernel<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, WA_LDS_MERGE) != hipSuccess ||
            hipFuncSetAttribute((const void*)wattn_kernel<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, WA_LDS_MERGE) != hipSuccess)
            return EMIP_E_LAUNCH;
    }
    EMIP_REQUIRE(!Wq || (aligned16(Wq) && L > 2 * WA_BK));      // (the weight rides in ring slot 2: at least three key tiles)
    const dim3 grid((unsigned)a.qblocks, (unsigned)nwin, (unsigned)B);
    if (Wq) {
        if (gid) hipLaunchKernelGGL((wattn_kernel<true, true, true>), grid, dim3(512), WA_LDS_MERGE, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((wattn_kernel<false, true, true>), grid, dim3(512), WA_LDS_MERGE, (hipStream_t)stream, a);
    } else {
        if (gid) hipLaunchKernelGGL((wattn_kernel<true, true>), grid, dim3(512), WA_LDS_MERGE, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((wattn_kernel<false, true>), grid, dim3(512), WA_LDS_MERGE, (hipStream_t)stream, a);
    }
    return emip_launch_status();
}
