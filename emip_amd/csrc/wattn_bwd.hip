// Backward of the GMFlow split-window attention (wattn.hip) for gfx950: bf16, single head, D = DV = 128, windows of L <= 512
// tokens (/root/reference/model/EMIP_short/motion/gmflow/transformer.py:46-105 under loss.backward() of train.py:52-58; GMFlow's
// weights are frozen, train.py:340-342, but the camouflage feeder's gradient flows back through its attention).
//
//     score = scale q.k - 100 [gid_q != gid_k]      P = softmax_k(score)      O = P V
//     dV = P^T dO      dP = dO V^T      dS = P o (dP - delta) * scale, delta_q = <dO_q, O_q>      dQ = dS K      dK = dS^T Q
//
// The training step ran this unfused on gathered windows: 4 gathers, QK^T, row softmax, dO V^T, softmax backward, three more
// batched GEMMs, a transpose and 3 scatters per attention -- the [256 windows][484][488] score matrix crossed HBM nine times,
// ~11 ms of a 102-ms step.  Here P is recomputed from the forward's log-sum-exp (emip_window_attention writes it) and never
// leaves registers; dQ needs a reduction over keys and dK / dV one over queries, so there are two kernels, each keeping ITS side
// stationary on the lanes (no atomics, no partial buffers) and streaming the other side through an LDS ring by LDS-DMA with the
// window's row table applied to the per-lane source address (no gathered copies, outputs land at their frame rows):
//
//   wattn_bwd_dq_kernel   8 waves x 32 queries; Q and dO rows are MFMA B fragments for the whole launch; K (row-read image for
//                         S^T = K Q^T, transposed-read image for dQ^T += K^T dS^T) and V (row-read, dP^T = V dO^T) stream in
//                         64-key tiles; the key is on the MFMA row, so the dS accumulators are the next B operand as they stand.
//   wattn_bwd_dkv_kernel  4 waves x 32 keys; K and V rows are the B fragments; Q and dO stream in 64-query tiles (row-read
//                         images for S = Q K^T, dP = dO V^T; transposed-read images for dK^T += Q^T dS, dV^T += dO^T P); the
//                         per-query log-sum-exp / delta / region id come from LDS tables, four consecutive queries per read.
//   rowdot_kernel         delta = rowsum(dO o O), 16 lanes per token.
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 wb_rsrc(const void* ptr, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
    return i32x4{(int)(unsigned)a, (int)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void wb_dma16(unsigned lds_dst, unsigned voff, i32x4 rs) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_dst), "v"(voff), "s"(rs)
        : "memory");
}

struct WbArgs {
    const bf16_t* Q;
    const bf16_t* K;
    const bf16_t* V;
    const bf16_t* dO;
    const float* lse;      // [B][tokens] log2-sum-exp of the scaled, masked scores (emip_window_attention)
    const float* delta;    // [B][tokens] <dO, O>
    bf16_t* dQ;
    bf16_t* dK;
    bf16_t* dV;            // [B][tokens][128]-like: row stride ldg, batch stride g_bs
    const int* rows;
    const int* gid;
    long ldq, ldk, ldv, lddo, q_bs, k_bs, v_bs, do_bs, ldg, g_bs;
    int B, nwin, L, rot, tokens;
    float scale;
    unsigned q_bytes, k_bytes, v_bytes, do_bytes;
};

constexpr unsigned WB_OOB = 0x80000000u;
constexpr int WB_BK = 64, WB_IMG = WB_BK * 256, WB_LMAX = 512;
// dq kernel: 3 ring slots of (K rows | K transposed-read | V rows) + row / region tables
constexpr int WQ_NST = 3, WQ_STAGE = 3 * WB_IMG, WQ_RING = WQ_NST * WQ_STAGE, WQ_LDS = WQ_RING + 2 * WB_LMAX * 4;      // 151 552 B
// dkv kernel: 2 ring slots of (Q rows | Q transposed | dO rows | dO transposed) + row / region / lse / delta tables
constexpr int WK_NST = 2, WK_STAGE = 4 * WB_IMG, WK_RING = WK_NST * WK_STAGE, WK_LDS = WK_RING + 4 * WB_LMAX * 4;      // 139 264 B

__device__ __forceinline__ int wb_voff(int row, int c) { return row * 256 + ((c ^ ((row & 3) << 2)) * 16); }

// MFMA A operand = 32 columns (32 d ..) of a transposed-read image, k = its rows base0 .. (the V^T fragment of wattn.hip)
__device__ __forceinline__ bf16x8 wb_tr_frag(const char* img, int base0, int d, int lane) {
    const int i16 = lane & 15, g16 = (lane >> 4) & 1;
    const int col = 32 * d + 16 * g16 + 4 * (i16 & 3);
    const int c = col >> 3, half = (col >> 2) & 1;
    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + wb_voff(base0, c) + 8 * half));
    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + wb_voff(base0 + 8, c) + 8 * half));
    const bf16x4 b0 = __builtin_bit_cast(bf16x4, v0), b1 = __builtin_bit_cast(bf16x4, v1);
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[j] = b0[j];
        f[4 + j] = b1[j];
    }
    return f;
}

// ------------------------------------------------------------------------------------------------------------------------
template <bool MASK>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void wattn_bwd_dq_kernel(const WbArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, h = lane >> 5;
    const int qb = blockIdx.x, win = blockIdx.y;
    const long b = blockIdx.z;
    long bk = b + p.rot;
    if (bk >= p.B) bk -= p.B;
    const i32x4 rsK = wb_rsrc(p.K + bk * p.k_bs, p.k_bytes), rsV = wb_rsrc(p.V + bk * p.v_bs, p.v_bytes);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)smem;
    int* trow = reinterpret_cast<int*>(smem + WQ_RING);
    int* tgid = trow + WB_LMAX;
    const int* rows = p.rows + (long)win * p.L;
    for (int i = tid; i < WB_LMAX; i += 512) {
        trow[i] = i < p.L ? rows[i] : 0;
        tgid[i] = (MASK && i < p.L) ? p.gid[(long)win * p.L + i] : 0;
    }

    // ---- this lane's query: fragments of its Q and dO rows (k-step i: channels 16 i + 8 h .. + 7), its statistics
    const int q = qb * 256 + wave * 32 + lq;
    const bool q_ok = q < p.L;
    const int qrow = rows[q_ok ? q : 0];
    const int q_g = MASK ? p.gid[(long)win * p.L + (q_ok ? q : 0)] : 0;
    uint4 qf[8], dof[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        qf[i] = *reinterpret_cast<const uint4*>(p.Q + b * p.q_bs + (long)qrow * p.ldq + (2 * i + h) * 8);
        dof[i] = *reinterpret_cast<const uint4*>(p.dO + b * p.do_bs + (long)qrow * p.lddo + (2 * i + h) * 8);
    }
    const float lse = p.lse[b * p.tokens + qrow] - __log2f(p.scale);          // exp2(s - lse) = scale P
    const float del = p.delta[b * p.tokens + qrow];
    __syncthreads();                                        // the tables are in LDS

    // ---- a tile = three 16-KB images of 64 keys; a 1-KB DMA piece = 4 key rows x 16 chunks (lane l: row l >> 4, slot l & 15);
    // wave w moves pieces 2 w, 2 w + 1 of every image.  Row-read images: source chunk slot ^ (row & 15); transposed-read:
    // slot ^ ((row & 3) << 2)
    auto issue = [&](int t) {
        const unsigned base = lds0 + (t % WQ_NST) * WQ_STAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = 4 * (2 * wave + j) + (lane >> 4);
            const int key = t * WB_BK + r;
            const bool ok = key < p.L;
            const int grow = trow[ok ? key : 0];
            const int cr = (lane & 15) ^ (r & 15), ct = (lane & 15) ^ ((r & 3) << 2);
            const unsigned kr = ok ? (unsigned)((grow * p.ldk + 8 * cr) * 2) : WB_OOB;
            const unsigned kt = ok ? (unsigned)((grow * p.ldk + 8 * ct) * 2) : WB_OOB;
            const unsigned vr = ok ? (unsigned)((grow * p.ldv + 8 * cr) * 2) : WB_OOB;
            wb_dma16(base + (2 * wave + j) * 1024, kr, rsK);
            wb_dma16(base + WB_IMG + (2 * wave + j) * 1024, kt, rsK);
            wb_dma16(base + 2 * WB_IMG + (2 * wave + j) * 1024, vr, rsV);
        }
    };
    const int ntile = (p.L + WB_BK - 1) / WB_BK;
    issue(0);
    if (ntile > 1) issue(1);

    f32x16 dq[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;
    const float sc2 = p.scale * 1.4426950408889634f;
    const float mask2 = -100.0f * 1.4426950408889634f;
    const int i16 = lane & 15;

    for (int t = 0; t < ntile; ++t) {
        if (t + 1 < ntile) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // all but the 6 pieces of tile t + 1
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < ntile) issue(t + 2);

        const char* kr_ = smem + (t % WQ_NST) * WQ_STAGE;
        const char* kt_ = kr_ + WB_IMG;
        const char* vr_ = kr_ + 2 * WB_IMG;
        // ---- S^T = K Q^T, dP^T = V dO^T (key on the MFMA row, query on the lane)
        f32x16 s[2], dp[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[kt][r] = 0.f;
                dp[kt][r] = 0.f;
            }
            const int row = 32 * kt + lq;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int off = row * 256 + (((2 * i + h) ^ (row & 15)) * 16);
                const uint4 kf = *reinterpret_cast<const uint4*>(kr_ + off);
                const uint4 vf = *reinterpret_cast<const uint4*>(vr_ + off);
                s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[i]),
                                                                s[kt], 0, 0, 0);
                dp[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, dof[i]),
                                                                 dp[kt], 0, 0, 0);
            }
        }
        // ---- P from the log-sum-exp, dS = P (dP - delta) scale; register 4 g + j of block kt = key 64 t + 32 kt + 8 g + 4 h + j
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int kb = t * WB_BK + 32 * kt + 8 * g + 4 * h;
                int4 kg = make_int4(0, 0, 0, 0);
                if (MASK) kg = *reinterpret_cast<const int4*>(tgid + kb);
                const int kgv[4] = {kg.x, kg.y, kg.z, kg.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = fmaf(s[kt][4 * g + j], sc2, -lse);
                    if (MASK) x += (kgv[j] != q_g) ? mask2 : 0.f;
                    float pr = __builtin_amdgcn_exp2f(x);
                    s[kt][4 * g + j] = pr * (dp[kt][4 * g + j] - del);      // pr = scale P; keys beyond L: zero K rows, no effect
                }
            }
        // ---- dQ^T += K^T dS^T
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                bf16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)s[kt][8 * sp + j];
                const int base0 = 32 * kt + 16 * sp + 4 * h + (i16 >> 2);
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    dq[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb_tr_frag(kt_, base0, d, lane), pf, dq[d], 0, 0, 0);
            }
    }

    // ---- store: registers 4 g .. 4 g + 3 of block d = channels 32 d + 8 g + 4 h + (0..3) of this lane's query
    if (q_ok) {
        bf16_t* Gp = p.dQ + b * p.g_bs + (long)qrow * p.ldg;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 ov;
#pragma unroll
                for (int j = 0; j < 4; ++j) ov[j] = (bf16_t)dq[d][4 * g + j];
                *reinterpret_cast<bf16x4*>(Gp + 32 * d + 8 * g + 4 * h) = ov;
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
template <bool MASK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wattn_bwd_dkv_kernel(const WbArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, h = lane >> 5;
    const int kblk = blockIdx.x, win = blockIdx.y;
    const long b = blockIdx.z;                               // frame of the queries; keys / values live in frame bk
    long bk = b + p.rot;
    if (bk >= p.B) bk -= p.B;
    const i32x4 rsQ = wb_rsrc(p.Q + b * p.q_bs, p.q_bytes), rsD = wb_rsrc(p.dO + b * p.do_bs, p.do_bytes);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)smem;
    int* trow = reinterpret_cast<int*>(smem + WK_RING);
    int* tgid = trow + WB_LMAX;
    float* tlse = reinterpret_cast<float*>(tgid + WB_LMAX);
    float* tdel = tlse + WB_LMAX;
    const int* rows = p.rows + (long)win * p.L;
    for (int i = tid; i < WB_LMAX; i += 256) {
        const bool ok = i < p.L;
        const int r = ok ? rows[i] : 0;
        trow[i] = r;
        tgid[i] = (MASK && ok) ? p.gid[(long)win * p.L + i] : 0;
        tlse[i] = ok ? p.lse[b * p.tokens + r] : 1e30f;      // P of a padding query = exp2(-1e30) = 0
        tdel[i] = ok ? p.delta[b * p.tokens + r] * p.scale : 0.f;
    }

    // ---- this lane's key: fragments of its K and V rows
    const int key = kblk * 128 + wave * 32 + lq;
    const bool k_ok = key < p.L;
    const int krow = rows[k_ok ? key : 0];
    const int k_g = MASK ? p.gid[(long)win * p.L + (k_ok ? key : 0)] : 0;
    uint4 kf[8], vf[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        kf[i] = *reinterpret_cast<const uint4*>(p.K + bk * p.k_bs + (long)krow * p.ldk + (2 * i + h) * 8);
        vf[i] = *reinterpret_cast<const uint4*>(p.V + bk * p.v_bs + (long)krow * p.ldv + (2 * i + h) * 8);
    }
    __syncthreads();

    // ---- a tile = four 16-KB images of 64 queries (Q rows | Q transposed-read | dO rows | dO transposed-read); wave w moves
    // pieces 4 w .. 4 w + 3 of every image
    auto issue = [&](int t) {
        const unsigned base = lds0 + (t % WK_NST) * WK_STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pc = 4 * wave + j;
            const int r = 4 * pc + (lane >> 4);
            const int qi = t * WB_BK + r;
            const bool ok = qi < p.L;
            const int grow = trow[ok ? qi : 0];
            const int cr = (lane & 15) ^ (r & 15), ct = (lane & 15) ^ ((r & 3) << 2);
            const unsigned qr = ok ? (unsigned)((grow * p.ldq + 8 * cr) * 2) : WB_OOB;
            const unsigned qt = ok ? (unsigned)((grow * p.ldq + 8 * ct) * 2) : WB_OOB;
            const unsigned dr = ok ? (unsigned)((grow * p.lddo + 8 * cr) * 2) : WB_OOB;
            const unsigned dt = ok ? (unsigned)((grow * p.lddo + 8 * ct) * 2) : WB_OOB;
            wb_dma16(base + pc * 1024, qr, rsQ);
            wb_dma16(base + WB_IMG + pc * 1024, qt, rsQ);
            wb_dma16(base + 2 * WB_IMG + pc * 1024, dr, rsD);
            wb_dma16(base + 3 * WB_IMG + pc * 1024, dt, rsD);
        }
    };
    const int ntile = (p.L + WB_BK - 1) / WB_BK;
    issue(0);

    f32x16 dk[4], dv[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dk[d][r] = 0.f;
            dv[d][r] = 0.f;
        }
    const float sc2 = p.scale * 1.4426950408889634f;
    const float mask2 = -100.0f * 1.4426950408889634f;
    const int i16 = lane & 15;

    for (int t = 0; t < ntile; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // tile t (the only one in flight)
        __builtin_amdgcn_s_barrier();                        // ... for every wave; everyone has left tile t - 1's slot
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < ntile) issue(t + 1);

        const char* qr_ = smem + (t % WK_NST) * WK_STAGE;
        const char* qt_ = qr_ + WB_IMG;
        const char* dr_ = qr_ + 2 * WB_IMG;
        const char* dt_ = qr_ + 3 * WB_IMG;
        // ---- S = Q K^T, dP = dO V^T (query on the MFMA row, key on the lane)
        f32x16 s[2], dp[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[qt][r] = 0.f;
                dp[qt][r] = 0.f;
            }
            const int row = 32 * qt + lq;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int off = row * 256 + (((2 * i + h) ^ (row & 15)) * 16);
                const uint4 qa = *reinterpret_cast<const uint4*>(qr_ + off);
                const uint4 da = *reinterpret_cast<const uint4*>(dr_ + off);
                s[qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qa), __builtin_bit_cast(bf16x8, kf[i]),
                                                                s[qt], 0, 0, 0);
                dp[qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, da), __builtin_bit_cast(bf16x8, vf[i]),
                                                                 dp[qt], 0, 0, 0);
            }
        }
        // ---- P and dS; register 4 g + j of block qt = query 64 t + 32 qt + 8 g + 4 h + j (statistics from the LDS tables)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int qb0 = t * WB_BK + 32 * qt + 8 * g + 4 * h;
                const float4 l4 = *reinterpret_cast<const float4*>(tlse + qb0);
                const float4 d4 = *reinterpret_cast<const float4*>(tdel + qb0);
                int4 g4 = make_int4(0, 0, 0, 0);
                if (MASK) g4 = *reinterpret_cast<const int4*>(tgid + qb0);
                const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
                const int gv[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = fmaf(s[qt][4 * g + j], sc2, -lv[j]);
                    if (MASK) x += (gv[j] != k_g) ? mask2 : 0.f;
                    const float pr = __builtin_amdgcn_exp2f(x);
                    s[qt][4 * g + j] = pr;
                    dp[qt][4 * g + j] = pr * fmaf(dp[qt][4 * g + j], p.scale, -dl[j]);      // the table holds scale delta
                }
            }
        // ---- dV^T += dO^T P, dK^T += Q^T dS
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                bf16x8 pf, sf;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    pf[j] = (bf16_t)s[qt][8 * sp + j];
                    sf[j] = (bf16_t)dp[qt][8 * sp + j];
                }
                const int base0 = 32 * qt + 16 * sp + 4 * h + (i16 >> 2);
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    dv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb_tr_frag(dt_, base0, d, lane), pf, dv[d], 0, 0, 0);
                    dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb_tr_frag(qt_, base0, d, lane), sf, dk[d], 0, 0, 0);
                }
            }
    }

    if (k_ok) {
        bf16_t* Kp = p.dK + bk * p.g_bs + (long)krow * p.ldg;
        bf16_t* Vp = p.dV + bk * p.g_bs + (long)krow * p.ldg;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 ok4, ov4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ok4[j] = (bf16_t)dk[d][4 * g + j];
                    ov4[j] = (bf16_t)dv[d][4 * g + j];
                }
                *reinterpret_cast<bf16x4*>(Kp + 32 * d + 8 * g + 4 * h) = ok4;
                *reinterpret_cast<bf16x4*>(Vp + 32 * d + 8 * g + 4 * h) = ov4;
            }
    }
}

// delta[row] = sum_c A[row][c] * Bm[row][c] over 128 channels: 16 lanes per row, 16 bytes per lane
__global__ __launch_bounds__(256) void rowdot128_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ Bm,
                                                        long ldb, float* __restrict__ out, long rows) {
    const int sub = threadIdx.x & 15;
    for (long r = (long)blockIdx.x * 16 + (threadIdx.x >> 4); r < rows; r += (long)gridDim.x * 16) {
        const uint4 a = *reinterpret_cast<const uint4*>(A + r * lda + sub * 8);
        const uint4 c = *reinterpret_cast<const uint4*>(Bm + r * ldb + sub * 8);
        const unsigned aw[4] = {a.x, a.y, a.z, a.w}, cw[4] = {c.x, c.y, c.z, c.w};
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s = fmaf(__uint_as_float(aw[k] << 16), __uint_as_float(cw[k] << 16), s);
            s = fmaf(__uint_as_float(aw[k] & 0xffff0000u), __uint_as_float(cw[k] & 0xffff0000u), s);
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 8);
        if (sub == 0) out[r] = s;
    }
}

}  // namespace

// Q, K, V, O, dO: bf16 token matrices of B frames of `tokens` tokens, 128 channels at the pointer, CONTIGUOUS frames for O / dO /
// the gradients (row stride 128, batch stride tokens * 128) so that one row index addresses lse / delta; Q, K, V may be column
// slices (row strides ldq / ldk / ldv, batch strides *_bs).  lse: f32 [B][tokens] from emip_window_attention; delta: f32
// [B][tokens] workspace; dQ, dK, dV: bf16 [B][tokens][128], every row written exactly once (rows is a bijection per frame).
extern "C" int emip_window_attention_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO,
                                         const float* lse, float* delta, void* dQ, void* dK, void* dV, int B, int nwin, int L,
                                         long ldq, long ldk, long ldv, long q_bs, long k_bs, long v_bs, const int* rows,
                                         const int* gid, int tokens, int kv_rot, float scale, void* stream) {
    EMIP_REQUIRE(Q && K && V && O && dO && lse && delta && dQ && dK && dV && rows);
    EMIP_REQUIRE(B > 0 && B < 65536 && nwin > 0 && nwin < 65536 && L >= WB_BK && L <= WB_LMAX && tokens >= L && (long)nwin * L == tokens);
    EMIP_REQUIRE(kv_rot >= 0 && kv_rot < B);
    EMIP_REQUIRE(ldq >= 128 && ldk >= 128 && ldv >= 128 && ((ldq | ldk | ldv | q_bs | k_bs | v_bs) & 7) == 0);
    EMIP_REQUIRE(aligned16(Q) && aligned16(K) && aligned16(V) && aligned16(O) && aligned16(dO) && aligned16(dQ) && aligned16(dK) &&
                 aligned16(dV));
    const long span = ((long)(tokens - 1) * 128 + 128) * 2;
    EMIP_REQUIRE(((long)(tokens - 1) * ldq + 128) * 2 < 0x7FFF0000L && ((long)(tokens - 1) * ldk + 128) * 2 < 0x7FFF0000L &&
                 ((long)(tokens - 1) * ldv + 128) * 2 < 0x7FFF0000L && span < 0x7FFF0000L);
    WbArgs a{};
    a.Q = (const bf16_t*)Q; a.K = (const bf16_t*)K; a.V = (const bf16_t*)V; a.dO = (const bf16_t*)dO;
    a.lse = lse; a.delta = delta; a.dQ = (bf16_t*)dQ; a.dK = (bf16_t*)dK; a.dV = (bf16_t*)dV; a.rows = rows; a.gid = gid;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.lddo = 128; a.q_bs = q_bs; a.k_bs = k_bs; a.v_bs = v_bs;
    a.do_bs = (long)tokens * 128; a.ldg = 128; a.g_bs = (long)tokens * 128;
    a.B = B; a.nwin = nwin; a.L = L; a.rot = kv_rot; a.tokens = tokens; a.scale = scale;
    a.q_bytes = (unsigned)(((long)(tokens - 1) * ldq + 128) * 2);
    a.k_bytes = (unsigned)(((long)(tokens - 1) * ldk + 128) * 2);
    a.v_bytes = (unsigned)(((long)(tokens - 1) * ldv + 128) * 2);
    a.do_bytes = (unsigned)span;
    static bool attr = false;
    if (!attr) {
        attr = true;
        if (hipFuncSetAttribute((const void*)wattn_bwd_dq_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, WQ_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)wattn_bwd_dq_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, WQ_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)wattn_bwd_dkv_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, WK_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)wattn_bwd_dkv_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, WK_LDS) != hipSuccess)
            return EMIP_E_LAUNCH;
    }
    hipStream_t st = (hipStream_t)stream;
    const long nrows = (long)B * tokens;
    long nb = (nrows + 15) / 16;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(rowdot128_kernel, dim3((unsigned)nb), dim3(256), 0, st, (const bf16_t*)dO, 128L, (const bf16_t*)O, 128L,
                       delta, nrows);
    const dim3 gq((unsigned)((L + 255) / 256), (unsigned)nwin, (unsigned)B);
    const dim3 gk((unsigned)((L + 127) / 128), (unsigned)nwin, (unsigned)B);
    if (gid) {
        hipLaunchKernelGGL(wattn_bwd_dq_kernel<true>, gq, dim3(512), WQ_LDS, st, a);
        hipLaunchKernelGGL(wattn_bwd_dkv_kernel<true>, gk, dim3(256), WK_LDS, st, a);
    } else {
        hipLaunchKernelGGL(wattn_bwd_dq_kernel<false>, gq, dim3(512), WQ_LDS, st, a);
        hipLaunchKernelGGL(wattn_bwd_dkv_kernel<false>, gk, dim3(256), WK_LDS, st, a);
    }
    return emip_launch_status();
}
