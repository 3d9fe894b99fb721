// Backward of emip_match (match.hip) for gfx950: GMFlow global matching in both directions with the gradient of the raw
// correlation volume coming back from conv_corr, and the flow-propagation attention (bf16, D = 128, n <= 2048 keys):
//
//     score = scale q.k      P = softmax_k(score)      O = P V   (V: 2 values per key -- the pixel grid or the flow)
//     dP_qk = <dO_q, V_k>    dscore = P o (dP - delta) + dS_up,  delta_q = <dO_q, O_q>      dQ = scale dscore K      dK = scale dscore^T Q
//
// /root/reference/model/EMIP_short/motion/gmflow/matching.py:8-41 and transformer.py:485-533 under loss.backward()
// (train.py:52-58); the matching layer has no parameters and GMFlow is frozen (train.py:340-342): token gradients only.
// The training step ran this as QK^T, a 1936-wide row softmax, softmax backward, an axpby with the upstream score gradient, a
// transpose and two more batched GEMMs per direction: the [32][1936][1936] score matrix crossed HBM nine times per call.
// Here, as in wattn_bwd.hip, P is recomputed from the forward's log-sum-exp and stays in registers; two kernels, each with its
// side stationary on the lanes:
//   match_bwd_dq_kernel  8 waves x 32 queries, K streams (row-read image for S^T = K Q^T, transposed-read image for
//                        dQ^T += K^T dS^T); V of the key batch sits in LDS (16 KB); the upstream gradient's [256 queries][64 keys]
//                        tile rides the ring too (8-byte loads per lane from the volume ran the kernel 2.8x slower);
//   match_bwd_dk_kernel  4 waves x 32 keys, Q streams (row-read + transposed-read images) with one 1-KB piece of per-query
//                        statistics (lse, delta, dO) per tile; the upstream gradient is read key-contiguous.
//   match_stat_kernel    packs (lse - log2 scale, delta, dO_x, dO_y) per query.
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 mb_rsrc(const void* ptr, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
    return i32x4{(int)(unsigned)a, (int)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void mb_dma16(unsigned lds_dst, unsigned voff, i32x4 rs) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_dst), "v"(voff), "s"(rs)
        : "memory");
}

struct MbArgs {
    const bf16_t* Q;      // [Z][n][ldq]
    const bf16_t* K;      // [Z][n][ldk]
    const float* V;       // [Z][n][2] (indexed like the keys) or null (= the pixel grid)
    const float4* stat;   // [Z][n] (lse - log2 scale, delta, dO_x, dO_y)
    const bf16_t* dS;     // [Zs][n][n] upstream gradient w.r.t. the scaled scores of batches z < Zs, or null
    bf16_t* dQ;           // [Z][n][128]
    bf16_t* dK;           // [Z][n][128]; rows of batch (z + rot) mod Z; may be dQ's buffer with accum = 1
    long ldq, ldk, q_bs, k_bs;
    int Z, Zs, n, W, rot, accum;
    float scale;
    unsigned q_bytes, k_bytes, st_bytes;
};

constexpr unsigned MB_OOB = 0x80000000u;
constexpr int MB_BK = 64, MB_IMG = MB_BK * 256, MB_NPAD = 2048;
// dq kernel: 3 ring slots of (K rows | K transposed-read); with an upstream gradient 2 slots of (K rows | K transposed-read | the
// [256 queries][64 keys] tile of the upstream gradient) -- plus V of every key
constexpr int MQ_IMGS = 2 * MB_IMG, MQ_UPT = 256 * 128;
constexpr int MQ_LDS_PLAIN = 3 * MQ_IMGS + MB_NPAD * 8;                 // 114 688 B
constexpr int MQ_LDS_UP = 2 * (MQ_IMGS + MQ_UPT) + MB_NPAD * 8;         // 147 456 B
constexpr int MK_NST = 3, MK_STAGE = 2 * MB_IMG + 1024, MK_RING = MK_NST * MK_STAGE, MK_LDS = MK_RING;              // 101 376 B

__device__ __forceinline__ int mb_voff(int row, int c) { return row * 256 + ((c ^ ((row & 3) << 2)) * 16); }

__device__ __forceinline__ bf16x8 mb_tr_frag(const char* img, int base0, int d, int lane) {
    const int i16 = lane & 15, g16 = (lane >> 4) & 1;
    const int col = 32 * d + 16 * g16 + 4 * (i16 & 3);
    const int c = col >> 3, half = (col >> 2) & 1;
    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + mb_voff(base0, c) + 8 * half));
    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + mb_voff(base0 + 8, c) + 8 * half));
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
template <bool UP>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void match_bwd_dq_kernel(const MbArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, h = lane >> 5;
    const int qb = blockIdx.x;
    const long z = blockIdx.y;
    long zk = z + p.rot;
    if (zk >= p.Z) zk -= p.Z;
    constexpr int NST = UP ? 2 : 3, STAGE = UP ? MQ_IMGS + MQ_UPT : MQ_IMGS, RING = NST * STAGE;
    const i32x4 rsK = mb_rsrc(p.K + zk * p.k_bs, p.k_bytes);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)smem;
    float2* tv = reinterpret_cast<float2*>(smem + RING);                   // V of every key of batch zk
    for (int i = tid; i < MB_NPAD; i += 512) {
        float2 v = make_float2(0.f, 0.f);
        if (i < p.n) {
            if (p.V) v = *reinterpret_cast<const float2*>(p.V + (zk * p.n + i) * 2);
            else {
                const int ky = i / p.W;
                v = make_float2((float)(i - ky * p.W), (float)ky);
            }
        }
        tv[i] = v;
    }
    const int q = qb * 256 + wave * 32 + lq;
    const bool q_ok = q < p.n;
    const int qc = q_ok ? q : 0;
    uint4 qf[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) qf[i] = *reinterpret_cast<const uint4*>(p.Q + z * p.q_bs + (long)qc * p.ldq + (2 * i + h) * 8);
    const float4 st = p.stat[z * p.n + qc];                                   // (lse, delta, dO_x, dO_y)
    const bool up = UP && z < p.Zs;
    // upstream gradient d score[z][q][k]: the tile [this workgroup's 256 queries][64 keys] rides the ring as 32 1-KB pieces (8
    // query rows of 128 B each), 4 per wave = the wave's own 32 queries; chunk ^ (row & 7) on the source side
    const i32x4 rsU = mb_rsrc(UP ? p.dS + (long)(up ? z : 0) * p.n * p.n : nullptr, (unsigned)((long)p.n * p.n * 2));
    __syncthreads();

    auto issue = [&](int t) {
        const unsigned base = lds0 + (t % NST) * STAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = 4 * (2 * wave + j) + (lane >> 4);
            const int key = t * MB_BK + r;
            const bool ok = key < p.n;
            const int cr = (lane & 15) ^ (r & 15), ct = (lane & 15) ^ ((r & 3) << 2);
            const unsigned kr = ok ? (unsigned)((key * p.ldk + 8 * cr) * 2) : MB_OOB;
            const unsigned kt = ok ? (unsigned)((key * p.ldk + 8 * ct) * 2) : MB_OOB;
            mb_dma16(base + (2 * wave + j) * 1024, kr, rsK);
            mb_dma16(base + MB_IMG + (2 * wave + j) * 1024, kt, rsK);
        }
        if (UP) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rl = 8 * j + (lane >> 3);                            // query row inside the wave's 32
                const int qr = qb * 256 + wave * 32 + rl;
                const int key = t * MB_BK + 8 * ((lane & 7) ^ (rl & 7));
                const unsigned uo = (up && qr < p.n && key < p.n) ? (unsigned)(((long)qr * p.n + key) * 2) : MB_OOB;
                mb_dma16(base + MQ_IMGS + wave * 4096 + j * 1024, uo, rsU);
            }
        }
    };
    const int ntile = (p.n + MB_BK - 1) / MB_BK;
    issue(0);
    if (!UP && ntile > 1) issue(1);

    f32x16 dq[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;
    const float sc2 = p.scale * 1.4426950408889634f;
    const int i16 = lane & 15;

    for (int t = 0; t < ntile; ++t) {
        if (UP) {                                            // 2 slots: tile t is the only one in flight
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (t + 1 < ntile) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (UP) {
            if (t + 1 < ntile) issue(t + 1);
        } else if (t + 2 < ntile) {
            issue(t + 2);
        }

        const char* kr_ = smem + (t % NST) * STAGE;
        const char* up_ = kr_ + MQ_IMGS + wave * 4096 + lq * 128 + h * 8;
        const char* kt_ = kr_ + MB_IMG;
        f32x16 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
            const int row = 32 * kt + lq;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint4 kf = *reinterpret_cast<const uint4*>(kr_ + row * 256 + (((2 * i + h) ^ (row & 15)) * 16));
                s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[i]),
                                                                s[kt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int kb = t * MB_BK + 32 * kt + 8 * g + 4 * h;
                const float4 va = *reinterpret_cast<const float4*>(tv + kb), vb = *reinterpret_cast<const float4*>(tv + kb + 2);
                const float vx[4] = {va.x, va.z, vb.x, vb.z}, vy[4] = {va.y, va.w, vb.y, vb.w};
                float u[4] = {0.f, 0.f, 0.f, 0.f};
                if (UP) {                                    // zeros where there is no upstream gradient (out-of-range DMA)
                    const uint2 ug = *reinterpret_cast<const uint2*>(up_ + (((4 * kt + g) ^ (lq & 7)) * 16));
                    u[0] = __uint_as_float(ug.x << 16); u[1] = __uint_as_float(ug.x & 0xffff0000u);
                    u[2] = __uint_as_float(ug.y << 16); u[3] = __uint_as_float(ug.y & 0xffff0000u);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // st.x carries -log2(scale): pr = scale P.  Keys beyond n: their K rows read as zeros, so whatever d score
                    // comes out for them adds nothing to dQ
                    const float pr = __builtin_amdgcn_exp2f(fmaf(s[kt][4 * g + j], sc2, -st.x));
                    const float w = fmaf(st.z, vx[j], fmaf(st.w, vy[j], -st.y));
                    s[kt][4 * g + j] = UP ? fmaf(u[j], p.scale, pr * w) : pr * w;
                }
            }
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
                    dq[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mb_tr_frag(kt_, base0, d, lane), pf, dq[d], 0, 0, 0);
            }
    }
    if (q_ok) {
        bf16_t* Gp = p.dQ + (z * p.n + q) * 128;
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
template <bool UP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void match_bwd_dk_kernel(const MbArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, h = lane >> 5;
    const int kblk = blockIdx.x;
    const long z = blockIdx.y;                               // batch of the queries; the keys live in batch zk
    long zk = z + p.rot;
    if (zk >= p.Z) zk -= p.Z;
    const i32x4 rsQ = mb_rsrc(p.Q + z * p.q_bs, p.q_bytes), rsS = mb_rsrc(p.stat + z * p.n, p.st_bytes);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)smem;

    const int key = kblk * 128 + wave * 32 + lq;
    const bool k_ok = key < p.n;
    const int kc = k_ok ? key : 0;
    uint4 kf[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) kf[i] = *reinterpret_cast<const uint4*>(p.K + zk * p.k_bs + (long)kc * p.ldk + (2 * i + h) * 8);
    float vx, vy;
    if (p.V) {
        const float2 v = *reinterpret_cast<const float2*>(p.V + (zk * p.n + kc) * 2);
        vx = v.x; vy = v.y;
    } else {
        const int ky = kc / p.W;
        vx = (float)(kc - ky * p.W); vy = (float)ky;
    }
    const bool up = UP && z < p.Zs;
    const bf16_t* ups = UP ? p.dS + (long)(up ? z : 0) * p.n * p.n + kc : nullptr;

    // a tile = Q rows | Q transposed-read | 1 KB of statistics of its 64 queries; wave w moves pieces 4 w .. 4 w + 3 of both
    // images, wave 0 also the statistics piece
    auto issue = [&](int t) {
        const unsigned base = lds0 + (t % MK_NST) * MK_STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pc = 4 * wave + j;
            const int r = 4 * pc + (lane >> 4);
            const int qi = t * MB_BK + r;
            const bool ok = qi < p.n;
            const int cr = (lane & 15) ^ (r & 15), ct = (lane & 15) ^ ((r & 3) << 2);
            const unsigned qr = ok ? (unsigned)((qi * p.ldq + 8 * cr) * 2) : MB_OOB;
            const unsigned qt = ok ? (unsigned)((qi * p.ldq + 8 * ct) * 2) : MB_OOB;
            mb_dma16(base + pc * 1024, qr, rsQ);
            mb_dma16(base + MB_IMG + pc * 1024, qt, rsQ);
        }
        if (wave == 0) {
            const int qi = t * MB_BK + lane;
            mb_dma16(base + 2 * MB_IMG, qi < p.n ? (unsigned)(qi * 16) : MB_OOB, rsS);      // zeros beyond n: P = exp2(s) (dO = 0)
        }
    };
    const int ntile = (p.n + MB_BK - 1) / MB_BK;
    issue(0);
    if (ntile > 1) issue(1);

    f32x16 dk[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[d][r] = 0.f;
    const float sc2 = p.scale * 1.4426950408889634f;
    const int i16 = lane & 15;
    const int npc = wave == 0 ? 9 : 8;                       // DMA pieces of a tile issued by this wave

    for (int t = 0; t < ntile; ++t) {
        // upstream gradient d score[q][key] of this lane's key for the tile's 64 queries, issued BEFORE the next DMA pieces (a
        // load behind them would be waited for in order, i.e. behind two tiles of DMA); unconditional: the waits count them
        unsigned short ur[2][16];
        if (UP) {
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qi = min(t * MB_BK + 32 * qt + 8 * (r >> 2) + 4 * h + (r & 3), p.n - 1);
                    ur[qt][r] = *reinterpret_cast<const unsigned short*>(ups + (long)qi * p.n);
                }
            if (t + 1 < ntile) {
                if (npc == 9) asm volatile("s_waitcnt vmcnt(41)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            }
        } else if (t + 1 < ntile) {
            if (npc == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < ntile) issue(t + 2);

        const char* qr_ = smem + (t % MK_NST) * MK_STAGE;
        const char* qt_ = qr_ + MB_IMG;
        const float4* stq = reinterpret_cast<const float4*>(qr_ + 2 * MB_IMG);
        f32x16 s[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[qt][r] = 0.f;
            const int row = 32 * qt + lq;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint4 qa = *reinterpret_cast<const uint4*>(qr_ + row * 256 + (((2 * i + h) ^ (row & 15)) * 16));
                s[qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qa), __builtin_bit_cast(bf16x8, kf[i]),
                                                                s[qt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ql = 32 * qt + 8 * g + 4 * h + j;              // query inside the tile
                    const float4 st = stq[ql];
                    const int qi = t * MB_BK + ql;
                    // pr = scale P (st.x carries -log2(scale)); queries beyond n: zero statistics, so w = 0
                    const float pr = __builtin_amdgcn_exp2f(fmaf(s[qt][4 * g + j], sc2, -st.x));
                    float u = 0.f;
                    if (UP && up && qi < p.n) u = __uint_as_float((unsigned)ur[qt][4 * g + j] << 16);
                    const float w = fmaf(st.z, vx, fmaf(st.w, vy, -st.y));
                    s[qt][4 * g + j] = UP ? fmaf(u, p.scale, pr * w) : pr * w;
                }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                bf16x8 sf;
#pragma unroll
                for (int j = 0; j < 8; ++j) sf[j] = (bf16_t)s[qt][8 * sp + j];
                const int base0 = 32 * qt + 16 * sp + 4 * h + (i16 >> 2);
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mb_tr_frag(qt_, base0, d, lane), sf, dk[d], 0, 0, 0);
            }
    }
    if (k_ok) {
        bf16_t* Gp = p.dK + (zk * p.n + key) * 128;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16_t* o = Gp + 32 * d + 8 * g + 4 * h;
                bf16x4 ov;
                if (p.accum) {
                    const bf16x4 old = *reinterpret_cast<const bf16x4*>(o);
#pragma unroll
                    for (int j = 0; j < 4; ++j) ov[j] = (bf16_t)(dk[d][4 * g + j] + (float)old[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) ov[j] = (bf16_t)dk[d][4 * g + j];
                }
                *reinterpret_cast<bf16x4*>(o) = ov;
            }
    }
}

// stat[z][q] = (lse, <dO, O>, dO_x, dO_y); O = Out (+ the query's own pixel when the forward subtracted it)
__global__ __launch_bounds__(256) void match_stat_kernel(const float* __restrict__ lse, const float* __restrict__ Out,
                                                         const float* __restrict__ dOut, float4* __restrict__ stat, long total,
                                                         int n, int W, int sub, float lscale) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const float2 o = *reinterpret_cast<const float2*>(Out + 2 * i), d = *reinterpret_cast<const float2*>(dOut + 2 * i);
        float ox = o.x, oy = o.y;
        if (sub) {
            const int q = (int)(i % n), qy = q / W;
            ox += (float)(q - qy * W);
            oy += (float)qy;
        }
        stat[i] = make_float4(lse[i] - lscale, d.x * ox + d.y * oy, d.x, d.y);      // exp2(s - stat.x) = scale P
    }
}

}  // namespace

// Q, K as in emip_match; V: f32 [Z][n][2] or NULL (pixel grid of width W); Out, dOut: f32 [Z][n][2] (the forward's output and
// its gradient); lse: f32 [Z][n] from emip_match; dS: bf16 [Zs][n][n] gradient w.r.t. the scaled scores the forward returned
// for batches z < Zs, or NULL; stat: f32 [Z][n][4] workspace; dQ, dK: bf16 [Z][n][128].  dK of the keys batch z read lands in
// batch (z + kv_rot) mod Z; accum_dk != 0: dK is ADDED to what dK's rows hold (pass dK == dQ for Q == K: the token gradient).
extern "C" int emip_match_bwd(const void* Q, const void* K, const float* V, const float* Out, const float* dOut, const float* lse,
                              const void* dS, float* stat, void* dQ, void* dK, int Z, int Zs, int n, int W, long ldq, long ldk,
                              long q_bs, long k_bs, int kv_rot, float scale, int sub_grid, int accum_dk, void* stream) {
    EMIP_REQUIRE(Q && K && Out && dOut && lse && stat && dQ && dK && Z > 0 && Z < 65536 && n >= 128 && n <= MB_NPAD && (n & 7) == 0);
    EMIP_REQUIRE(W > 0 && kv_rot >= 0 && kv_rot < Z && Zs >= 0 && Zs <= Z && (Zs == 0 || dS));
    EMIP_REQUIRE(ldq >= 128 && ldk >= 128 && ((ldq | ldk | q_bs | k_bs) & 7) == 0);
    EMIP_REQUIRE(aligned16(Q) && aligned16(K) && aligned16(stat) && aligned16(dQ) && aligned16(dK) && (!V || aligned16(V)) &&
                 (!dS || aligned16(dS)) && ((uintptr_t)Out & 7) == 0 && ((uintptr_t)dOut & 7) == 0);
    EMIP_REQUIRE(((long)(n - 1) * ldq + 128) * 2 < 0x7FFF0000L && ((long)(n - 1) * ldk + 128) * 2 < 0x7FFF0000L);
    MbArgs a{};
    a.Q = (const bf16_t*)Q; a.K = (const bf16_t*)K; a.V = V; a.stat = (const float4*)stat; a.dS = (const bf16_t*)dS;
    a.dQ = (bf16_t*)dQ; a.dK = (bf16_t*)dK;
    a.ldq = ldq; a.ldk = ldk; a.q_bs = q_bs; a.k_bs = k_bs;
    a.Z = Z; a.Zs = dS ? Zs : 0; a.n = n; a.W = W; a.rot = kv_rot; a.accum = accum_dk ? 1 : 0; a.scale = scale;
    a.q_bytes = (unsigned)(((long)(n - 1) * ldq + 128) * 2);
    a.k_bytes = (unsigned)(((long)(n - 1) * ldk + 128) * 2);
    a.st_bytes = (unsigned)(n * 16);
    static bool attr = false;
    if (!attr) {
        attr = true;
        if (hipFuncSetAttribute((const void*)match_bwd_dq_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, MQ_LDS_UP) != hipSuccess ||
            hipFuncSetAttribute((const void*)match_bwd_dq_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, MQ_LDS_PLAIN) != hipSuccess ||
            hipFuncSetAttribute((const void*)match_bwd_dk_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, MK_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)match_bwd_dk_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, MK_LDS) != hipSuccess)
            return EMIP_E_LAUNCH;
    }
    hipStream_t st = (hipStream_t)stream;
    const long total = (long)Z * n;
    long nb = (total + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(match_stat_kernel, dim3((unsigned)nb), dim3(256), 0, st, lse, Out, dOut, (float4*)stat, total, n, W,
                       sub_grid ? 1 : 0, log2f(scale));
    const dim3 gq((unsigned)((n + 255) / 256), (unsigned)Z), gk((unsigned)((n + 127) / 128), (unsigned)Z);
    if (a.Zs > 0) {
        hipLaunchKernelGGL(match_bwd_dq_kernel<true>, gq, dim3(512), MQ_LDS_UP, st, a);
        hipLaunchKernelGGL(match_bwd_dk_kernel<true>, gk, dim3(256), MK_LDS, st, a);
    } else {
        hipLaunchKernelGGL(match_bwd_dq_kernel<false>, gq, dim3(512), MQ_LDS_PLAIN, st, a);
        hipLaunchKernelGGL(match_bwd_dk_kernel<false>, gk, dim3(256), MK_LDS, st, a);
    }
    return emip_launch_status();
}
