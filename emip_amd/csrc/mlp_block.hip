// The Mlp half of a PVTv2 block in ONE launch (bf16 inference, the 22 x 22 stage: C = 320, hidden 1280):
//
//     out = x + fc2( GELU( dwconv3x3( LN(x) W1^T + b1 ) + bd ) ) + b2          and the row statistics of `out`
//
// (/root/reference/lib/pvt_v2.py:45-54 Mlp.forward with DWConv :316-327, inside Block.forward :165-169; norm2 folded into
// W1 / b1 and applied on the output side from the row statistics that travel with the residual stream, like emip_gemm_lne).
//
// Two launches did this before (emip_mlp_fc1dw + the fc2 GEMM): the activated hidden tensor -- 4 x the token tensor, the
// largest stream of the forward -- went out to HBM and came back, and both launches filled the whole chip for 49 + 36 us
// per block at 32 images.  Here a workgroup owns a BAND of image rows (<= 5 rows = 110 tokens, plus one halo row above and
// below for the depthwise taps) for ALL 1280 hidden channels, 64 at a time, and nothing of the hidden tensor leaves the CU:
//   * 10 waves; wave w keeps the 16 tokens of fc1 row tile w (band + halo: <= 160 tokens) as MFMA B-operand fragments in
//     40 registers for the whole launch, and the fc2 accumulators of output channels 32 w .. 32 w + 31 for all (<= 112) own
//     tokens in 56 -- tokens are never re-read, the residual stream costs one read and one write;
//   * per 64-channel chunk: fc1 (W1 chunk k-blocked in LDS as the A operand) -> LayerNorm on the output side + bias -> H
//     (bf16 [token][64]) in LDS -> depthwise 3 x 3 + bias + GELU out of LDS (register windows, as emip_mlp_fc1dw) -> G in
//     LDS -> fc2 partial sums (W2 chunk in LDS as the A operand, G as B).  Three s_barriers per chunk;
//   * the weights stream by LDS-DMA one phase ahead into SINGLE buffers: W2's chunk is fetched while fc1 and the depthwise
//     pass of the same chunk run, W1's next chunk (and the next chunk's taps / biases, one prepacked 3-KB block) while the
//     depthwise pass and fc2 run; two counted waits per chunk;
//   * epilogue: + b2 + x in registers, one rounding, row sums / sums of squares of the stored values through LDS atomics.
// The halo rows are recomputed by the neighbouring band (fc1 work x 1.36), the price of no exchange between workgroups.
// 160 workgroups at 32 images, one per CU: the launch leaves 96 CUs to the other steps in flight (bench.py replays three).
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
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
    const bf16_t* X;        // [B * H * W, ldx] raw residual stream
    bf16_t* Out;            // [B * H * W, ldo] (must not alias X: the halo rows of a band are another band's output rows)
    const bf16_t* W1;       // [N, C] fc1 weights with the LayerNorm scale folded in
    const bf16_t* W2;       // [C, N] fc2 weights
    const float* cst;       // [N / 64][12][64]: per 64-channel chunk the 9 depthwise taps, its bias, fc1's bias, W1's row sums
    const float* b2;        // [C]
    const float* ln_stats;  // [B * H * W, 2] (sum, sum of squares) of the rows of X
    float* out_stats;       // [B * H * W, 2] of the rows of Out (stored, not accumulated)
    long ldx, ldo;
    int B, H, W, nbands, xcd_map;
    float eps;
    unsigned x_bytes, w1_bytes, w2_bytes, cst_bytes;
#ifdef EMIP_TUNING
    int skip;               // calibration build only: 1 no fc1 MFMAs, 2 no depthwise pass, 4 no fc2 MFMAs, 8 no weight DMA, 16 no H store
#endif
};
#ifdef EMIP_TUNING
#define MB_SKIP(bit) (p.skip & (bit))
#else
#define MB_SKIP(bit) false
#endif

constexpr int MB_C = 320, MB_N = 1280, MB_NC = 64, MB_NCHUNK = MB_N / MB_NC;
constexpr int MB_T1 = 10, MB_T2 = 7;                       // 16-token tiles: fc1 rows (band + halo), fc2 rows (band)
// LDS: W1 chunk, k-blocked [10 k-steps][64 ch][64 B] | W2 chunk [320 out ch][128 B] | H [160][128 B] | G [112][128 B] |
//      2 x constants [12][64] f32 | row statistics [112][2] f32
constexpr int MB_W1 = 0, MB_W2 = MB_W1 + 10 * 64 * 64, MB_H = MB_W2 + 320 * 128, MB_G = MB_H + 160 * 128,
              MB_CST = MB_G + 112 * 128, MB_ST = MB_CST + 2 * 12 * 64 * 4, MB_LDS = MB_ST + 112 * 2 * 4;     // 123 776 B

__global__ __launch_bounds__(640) void mlp_block_kernel(const MbArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    int img, band;
    if (p.xcd_map) {       // the bands of an image on one XCD: they share its token rows in that L2
        const int j = blockIdx.x >> 3;
        band = j % p.nbands;
        img = (blockIdx.x & 7) + 8 * (j / p.nbands);
    } else {
        band = blockIdx.x % p.nbands;
        img = blockIdx.x / p.nbands;
    }
    const int y0 = band * p.H / p.nbands, y1 = (band + 1) * p.H / p.nbands;       // own image rows [y0, y1)
    const int yh0 = max(y0 - 1, 0), yh1 = min(y1 + 1, p.H);                        // with the halo
    const int ntok_h = (yh1 - yh0) * p.W, ntok = (y1 - y0) * p.W;                  // <= 160, <= 112
    const long tok_h0 = (long)img * p.H * p.W + (long)yh0 * p.W;                   // global token of halo token 0
    const int own_off = (y0 - yh0) * p.W;                                           // halo index of own token 0
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)smem;
    const i32x4 rsW1 = mb_rsrc(p.W1, p.w1_bytes), rsW2 = mb_rsrc(p.W2, p.w2_bytes), rsC = mb_rsrc(p.cst, p.cst_bytes);

    // ---- weight streams (the per-lane source offsets are rebuilt at every issue from an opaque copy of the lane id: kept
    // live across the loop they were spilled, 8 registers the kernel does not have).
    // W1 chunk: piece (k-step s, row block b) = 16 channel rows x 64 B; lane l sits at row l >> 2, slot l & 3 and fetches
    // source 16-B chunk slot ^ (row >> 2 & 3); wave w moves the 4 row blocks of k-step w.
    // W2 chunk: piece = 8 output-channel rows x 128 B; lane l sits at row l >> 3, slot l & 7, source chunk slot ^ (row & 7);
    // wave w moves pieces w, w + 10, w + 20, w + 30.
    auto issue_w1 = [&](int c) {          // + the chunk's constants: 3 KB, one 1-KB piece from each of the waves 0..2
        int l = lane;
        asm volatile("" : "+v"(l));
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int r = 16 * b + (l >> 2);
            const unsigned off = (unsigned)((r * MB_C + 32 * wave + 8 * ((l & 3) ^ ((r >> 2) & 3))) * 2);
            mb_dma16(lds0 + MB_W1 + wave * 4096 + b * 1024, off + (unsigned)(c * MB_NC * MB_C * 2), rsW1);
        }
        if (wave < 3) mb_dma16(lds0 + MB_CST + (c & 1) * 3072 + wave * 1024, (unsigned)(c * 3072 + wave * 1024 + l * 16), rsC);
    };
    auto issue_w2 = [&](int c) {
        int l = lane;
        asm volatile("" : "+v"(l));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 8 * (wave + 10 * j) + (l >> 3);
            const unsigned off = (unsigned)((r * MB_N + 8 * ((l & 7) ^ (r & 7))) * 2);
            mb_dma16(lds0 + MB_W2 + (wave + 10 * j) * 1024, off + (unsigned)(c * MB_NC * 2), rsW2);
        }
    };
    issue_w1(0);

    // ---- this wave's 16 tokens (fc1 row tile `wave` of the band + halo) as B-operand fragments; their LayerNorm statistics
    const int th = 16 * wave + fr;                          // halo-token index of this lane's token
    const bool t_ok = th < ntok_h;
    uint4 xf[10];
    {
        const bf16_t* xr = p.X + (tok_h0 + (t_ok ? th : 0)) * p.ldx + 8 * fq;
#pragma unroll
        for (int s = 0; s < 10; ++s) xf[s] = mask4(*reinterpret_cast<const uint4*>(xr + 32 * s), t_ok);
    }
    float rs = 0.f, mrs = 0.f;
    {
        const float2 s2 = *reinterpret_cast<const float2*>(p.ln_stats + 2 * (tok_h0 + (t_ok ? th : 0)));
        const float mu = s2.x * (1.f / MB_C);
        rs = rsqrtf(fmaxf(s2.y * (1.f / MB_C) - mu * mu, 0.f) + p.eps);
        mrs = mu * rs;
    }
    for (int i = tid; i < 112 * 2; i += 640) reinterpret_cast<float*>(smem + MB_ST)[i] = 0.f;

    f32x4 acc2[MB_T2][2];
#pragma unroll
    for (int a = 0; a < MB_T2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc2[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // depthwise pass geometry: thread = 8-channel group cg x one pixel pair of the band (rebuilt per chunk, see above)
    const int W2p = (p.W + 1) >> 1;
    const int npair = (y1 - y0) * W2p;

    for (int c = 0; c < MB_NCHUNK; ++c) {
        // ---- (A) W1 chunk c and its constants have landed; every wave has left chunk c - 1 (its fc2 read W2 / G)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (!MB_SKIP(8)) issue_w2(c);
        const float* cs = reinterpret_cast<const float*>(smem + MB_CST + (c & 1) * 3072);    // [12][64]: taps 0..8, bd, b1, colsum

        // ---- fc1: H^T tile = W1c (A: 16 channels x 32 k) x tokens (B), 4 channel tiles x 10 k-steps
        f32x4 acc1[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc1[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!MB_SKIP(1))
#pragma unroll
        for (int s = 0; s < 10; ++s) {
            const char* slab = smem + MB_W1 + s * 4096;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int row = 16 * ct + fr;
                const uint4 wf = *reinterpret_cast<const uint4*>(slab + row * 64 + ((fq ^ ((row >> 2) & 3)) * 16));
                acc1[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, xf[s]),
                                                                   acc1[ct], 0, 0, 0);
            }
        }
        // ---- output-side LayerNorm + bias -> H (bf16 [halo token][64 ch], 128-B rows, chunk ^ (token & 7))
        // acc1[ct][j] = channel 16 ct + 4 fq + j of token th
        if (!MB_SKIP(16))
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const int col = 16 * ct + 4 * fq;
            const float4 b1v = *reinterpret_cast<const float4*>(cs + 10 * 64 + col);
            const float4 csv = *reinterpret_cast<const float4*>(cs + 11 * 64 + col);
            bf16x4 hv;
            hv[0] = (bf16_t)fmaf(acc1[ct][0], rs, fmaf(-mrs, csv.x, b1v.x));
            hv[1] = (bf16_t)fmaf(acc1[ct][1], rs, fmaf(-mrs, csv.y, b1v.y));
            hv[2] = (bf16_t)fmaf(acc1[ct][2], rs, fmaf(-mrs, csv.z, b1v.z));
            hv[3] = (bf16_t)fmaf(acc1[ct][3], rs, fmaf(-mrs, csv.w, b1v.w));
            *reinterpret_cast<bf16x4*>(smem + MB_H + th * 128 + (((col >> 3) ^ (th & 7)) * 16) + ((col >> 2) & 1) * 8) = hv;
        }
        // ---- (C) H is complete (so every wave is through with W1 chunk c): fetch the next one, run the depthwise pass
        __syncthreads();
        if (c + 1 < MB_NCHUNK && !MB_SKIP(8)) issue_w1(c + 1);
        int tt = tid;
        asm volatile("" : "+v"(tt));
        const int cg = tt & 7, pr = tt >> 3;
        const int py = pr < npair ? y0 + pr / W2p : -4;     // image row of the pair (-4: none)
        const int px = 2 * (pr - (pr / W2p) * W2p);
        if (py >= 0 && !MB_SKIP(2)) {
            float o[2][8];
#pragma unroll
            for (int j = 0; j < 8; j += 4) {
                const float4 t4 = *reinterpret_cast<const float4*>(cs + 9 * 64 + 8 * cg + j);
                o[0][j] = o[1][j] = t4.x; o[0][j + 1] = o[1][j + 1] = t4.y;
                o[0][j + 2] = o[1][j + 2] = t4.z; o[0][j + 3] = o[1][j + 3] = t4.w;
            }
            // pixel q = 0 at column px takes window columns px - 1 .. px + 1, pixel q = 1 at px + 1 the columns px .. px + 2:
            // tap kx of a kernel row meets window column kx for q = 0 and kx + 1 for q = 1, so two columns stay live
            auto hload = [&](int yy, int xx, float (&v)[8]) {
                const bool in = (unsigned)xx < (unsigned)p.W;
                const int t = (yy - yh0) * p.W + (in ? xx : 0);
                uint4 raw = *reinterpret_cast<const uint4*>(smem + MB_H + t * 128 + ((cg ^ (t & 7)) * 16));
                raw = mask4(raw, in);
                const unsigned rw[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[2 * k] = __uint_as_float(rw[k] << 16);
                    v[2 * k + 1] = __uint_as_float(rw[k] & 0xFFFF0000u);
                }
            };
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int yy = py + ky - 1;
                if ((unsigned)yy >= (unsigned)p.H) continue;
                float v0[8], v1[8];
                hload(yy, px - 1, v0);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    hload(yy, px + kx, v1);
                    float w[8];
#pragma unroll
                    for (int j = 0; j < 8; j += 4) {
                        const float4 t4 = *reinterpret_cast<const float4*>(cs + (3 * ky + kx) * 64 + 8 * cg + j);
                        w[j] = t4.x; w[j + 1] = t4.y; w[j + 2] = t4.z; w[j + 3] = t4.w;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        o[0][j] = fmaf(v0[j], w[j], o[0][j]);
                        o[1][j] = fmaf(v1[j], w[j], o[1][j]);
                        v0[j] = v1[j];
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int x = px + q;
                if (x < p.W) {
                    const int t = (py - y0) * p.W + x;      // own-token index
                    bf16x8 ov;
#pragma unroll
                    for (int j = 0; j < 8; ++j) ov[j] = (bf16_t)gelu_poly(o[q][j]);
                    *reinterpret_cast<bf16x8*>(smem + MB_G + t * 128 + ((cg ^ (t & 7)) * 16)) = ov;
                }
            }
        }
        // ---- (D) G is complete and W2 chunk c has landed (the next W1 chunk's pieces, issued after it, may stay in flight: 4, or
        // 5 for the waves that also fetch the constants)
        if (c + 1 < MB_NCHUNK) {
            if (wave < 3) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        // ---- fc2: out^T tiles += W2c (A: 16 output channels x 32 k) x G (B: 32 k x 16 own tokens)
        if (!MB_SKIP(4))
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            uint4 af[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int row = 32 * wave + 16 * ct + fr;
                af[ct] = *reinterpret_cast<const uint4*>(smem + MB_W2 + row * 128 + (((4 * s2 + fq) ^ (row & 7)) * 16));
            }
#pragma unroll
            for (int rt = 0; rt < MB_T2; ++rt) {
                const int t = 16 * rt + fr;
                const uint4 gf = *reinterpret_cast<const uint4*>(smem + MB_G + t * 128 + (((4 * s2 + fq) ^ (t & 7)) * 16));
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    acc2[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[ct]),
                                                                           __builtin_bit_cast(bf16x8, gf), acc2[rt][ct], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: acc2[rt][ct][j] = output channel 32 wave + 16 ct + 4 fq + j of own token 16 rt + fr
    float* st = reinterpret_cast<float*>(smem + MB_ST);
    const long tok0 = tok_h0 + own_off;
#pragma unroll
    for (int rt = 0; rt < MB_T2; ++rt) {
        const int t = 16 * rt + fr;
        const bool ok = t < ntok;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int col = 32 * wave + 16 * ct + 4 * fq;
            if (ok) {
                const float4 bv = *reinterpret_cast<const float4*>(p.b2 + col);
                const bf16x4 xv = *reinterpret_cast<const bf16x4*>(p.X + (tok0 + t) * p.ldx + col);
                bf16x4 ov;
                ov[0] = (bf16_t)(acc2[rt][ct][0] + bv.x + (float)xv[0]);
                ov[1] = (bf16_t)(acc2[rt][ct][1] + bv.y + (float)xv[1]);
                ov[2] = (bf16_t)(acc2[rt][ct][2] + bv.z + (float)xv[2]);
                ov[3] = (bf16_t)(acc2[rt][ct][3] + bv.w + (float)xv[3]);
                *reinterpret_cast<bf16x4*>(p.Out + (tok0 + t) * p.ldo + col) = ov;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float f = (float)ov[j];
                    s1 += f;
                    s2 = fmaf(f, f, s2);
                }
            }
        }
        s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        if (ok && fq == 0) {
            atomicAdd(st + 2 * t, s1);
            atomicAdd(st + 2 * t + 1, s2);
        }
    }
    __syncthreads();
    if (p.out_stats && tid < ntok)
        *reinterpret_cast<float2*>(p.out_stats + 2 * (tok0 + tid)) = make_float2(st[2 * tid], st[2 * tid + 1]);
}

}  // namespace

#ifdef EMIP_TUNING
static int g_mb_skip = 0;
extern "C" int emip_debug_set_mb(int flags) { g_mb_skip = flags; return 0; }
#endif

// bands per image so that band + halo <= 160 tokens and band <= 112 tokens; 0 = the shape does not fit this kernel
static int mb_bands(int H, int W) {
    for (int nb = 1; nb <= H; ++nb) {
        int worst_own = 0, worst_halo = 0;
        for (int b = 0; b < nb; ++b) {
            const int y0 = b * H / nb, y1 = (b + 1) * H / nb;
            const int h0 = y0 > 0 ? y0 - 1 : 0, h1 = y1 < H ? y1 + 1 : H;
            if ((y1 - y0) * W > worst_own) worst_own = (y1 - y0) * W;
            if ((h1 - h0) * W > worst_halo) worst_halo = (h1 - h0) * W;
        }
        if (worst_own <= 16 * MB_T2 && worst_halo <= 16 * MB_T1 && worst_own > 0) return nb;
    }
    return 0;
}

extern "C" int emip_mlp_block_eligible(int B, int H, int W, int C, int N) {
    return B > 0 && C == MB_C && N == MB_N && W >= 2 && W <= 32 && H >= 1 && mb_bands(H, W) > 0;
}

// Out = X + fc2(GELU(dwconv3x3(LN(X) W1^T + b1) + bd)) + b2 per image, and out_stats = (sum, sum of squares) of the rows of
// Out (may be NULL).  X, Out: bf16 [B, H, W, 320] with row strides ldx / ldo, Out must not overlap X.  W1: bf16 [1280][320]
// with the LayerNorm scale folded in, W2: bf16 [320][1280], cst: f32 [20][12][64] -- per 64-channel chunk of the hidden
// tensor the 9 depthwise taps, the depthwise bias, fc1's bias (+ W1 beta) and the row sums of the packed W1 -- b2: f32 [320],
// ln_stats: f32 [B H W][2] (sum, sum of squares) of the rows of X.
extern "C" int emip_mlp_block(const void* X, long ldx, const void* W1, const void* W2, const float* cst, const float* b2,
                              const float* ln_stats, float eps, void* Out, long ldo, float* out_stats, int B, int H, int W,
                              int C, int N, void* stream) {
    EMIP_REQUIRE(X && W1 && W2 && cst && b2 && ln_stats && Out && emip_mlp_block_eligible(B, H, W, C, N));
    EMIP_REQUIRE(ldx >= C && (ldx & 7) == 0 && ldo >= C && (ldo & 3) == 0);
    EMIP_REQUIRE(aligned16(X) && aligned16(W1) && aligned16(W2) && aligned16(cst) && aligned16(b2) &&
                 (reinterpret_cast<uintptr_t>(Out) & 7u) == 0 && (reinterpret_cast<uintptr_t>(ln_stats) & 7u) == 0 &&
                 (reinterpret_cast<uintptr_t>(out_stats) & 7u) == 0);
    const long rows = (long)B * H * W;
    {   // no overlap of the two token tensors
        const char* x0 = (const char*)X; const char* x1 = x0 + ((rows - 1) * ldx + C) * 2;
        const char* o0 = (const char*)Out; const char* o1 = o0 + ((rows - 1) * ldo + C) * 2;
        EMIP_REQUIRE(x1 <= o0 || o1 <= x0);
    }
    EMIP_REQUIRE(rows * ldx * 2 < 0x7FFF0000L);
    MbArgs a{};
    a.X = (const bf16_t*)X; a.Out = (bf16_t*)Out; a.W1 = (const bf16_t*)W1; a.W2 = (const bf16_t*)W2; a.cst = cst; a.b2 = b2;
    a.ln_stats = ln_stats; a.out_stats = out_stats; a.ldx = ldx; a.ldo = ldo; a.B = B; a.H = H; a.W = W; a.eps = eps;
    a.nbands = mb_bands(H, W);
    a.xcd_map = (B % 8) == 0;
    a.x_bytes = (unsigned)(((rows - 1) * ldx + C) * 2);
    a.w1_bytes = (unsigned)((long)MB_N * MB_C * 2);
    a.w2_bytes = (unsigned)((long)MB_C * MB_N * 2);
    a.cst_bytes = (unsigned)(MB_NCHUNK * 12 * 64 * 4);
#ifdef EMIP_TUNING
    a.skip = g_mb_skip;
#endif
    static bool attr = false;
    if (!attr) {
        attr = true;
        if (hipFuncSetAttribute((const void*)mlp_block_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MB_LDS) != hipSuccess)
            return EMIP_E_LAUNCH;
    }
    hipLaunchKernelGGL(mlp_block_kernel, dim3((unsigned)(B * a.nbands)), dim3(640), MB_LDS, (hipStream_t)stream, a);
    return emip_launch_status();
}
