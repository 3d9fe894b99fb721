// GMFlow transformer FFN in one launch (bf16 inference):
//     out = res + LayerNorm( GELU([x1 | x2] W0^T) W2^T ) * gamma + beta
// /root/reference/model/EMIP_short/motion/gmflow/transformer.py:316-345 (TransformerLayer.forward with the FFN: mlp =
// Linear(2C, 8C, bias=False) -> GELU -> Linear(8C, C, bias=False), norm2, `source + message`), C = 128.
//
// As two launches (GEMM + GELU, GEMM + LayerNorm + residual) the [tokens][1024] hidden tensor crosses memory twice: 254 MB per
// layer at 32 frames, the largest stream of the GMFlow half.  Here it never exists:
//   * a workgroup = 256 tokens = 8 waves x 32; a wave keeps its tokens' 256 input channels as MFMA B fragments in registers
//     (64 VGPRs) for the whole launch and owns ALL 128 output channels of those tokens (4 accumulator tiles);
//   * the hidden dimension is walked in 32 chunks of 32 channels: H^T chunk = W0[chunk] X^T (16 MFMAs 32x32x16, the token on
//     the lane), GELU on the accumulators, which -- rounded to bf16 -- ARE the B operand of out^T += W2[:, chunk] H^T chunk
//     (8 MFMAs): no LDS round trip between the two contractions;
//   * both weight matrices are packed on the host in FRAGMENT order (one 1-KB piece = one MFMA A operand of all 64 lanes, W2 in
//     the key order the accumulator registers have), so a chunk is 24 contiguous KB: LDS-DMA into a 4-slot ring (3 chunks in
//     flight, one s_barrier + one counted vmcnt per chunk), fragment reads are conflict-free ds_read_b128 at lane * 16;
//   * epilogue: the lane pair (lq, lq + 32) holds the 128 outputs of a token: LayerNorm statistics by one shuffle, gamma / beta
//     from LDS, residual added, the tile leaves through an LDS image in the drained ring: whole 256-byte rows.
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned ff_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned ff_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ i32x4 ff_rsrc(const void* ptr, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
    return i32x4{(int)(unsigned)a, (int)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void ff_dma16(unsigned lds_dst, unsigned voff, i32x4 rs) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_dst), "v"(voff), "s"(rs)
        : "memory");
}

struct FfArgs {
    const bf16_t* X1;      // [M, ld1]  first 128 input channels
    const bf16_t* X2;      // [M, ld2]  second 128 input channels
    const bf16_t* W0p;     // packed [32 chunks][16 k-steps][2 halves][32 rows][8]   (fragment order, see ops.ffn_block_packs)
    const bf16_t* W2p;     // packed [32 chunks][4 row tiles][2 k-steps][2 halves][32 rows][8]
    const float* gamma;
    const float* beta;
    const bf16_t* Res;     // [M, ldr] (may alias Out)
    bf16_t* Out;           // [M, ldo]
    long ld1, ld2, ldr, ldo;
    int M;
    float eps;
};

constexpr int FF_C = 128, FF_HID = 1024, FF_CH = 32, FF_NCH = FF_HID / FF_CH;
constexpr int FF_W0C = 16 * 1024, FF_W2C = 8 * 1024, FF_SLOT = FF_W0C + FF_W2C;      // 24 KB per chunk
constexpr int FF_NST = 4, FF_RING = FF_NST * FF_SLOT;                                // 96 KB
constexpr int FF_LDS = FF_RING + 2 * FF_C * 4;                                       // + gamma, beta

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void ffn_block_kernel(const FfArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, h = lane >> 5;
    const i32x4 rs0 = ff_rsrc(p.W0p, (unsigned)(FF_HID * 2 * FF_C * 2)), rs2 = ff_rsrc(p.W2p, (unsigned)(FF_C * FF_HID * 2));
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)smem;
    float* tg = reinterpret_cast<float*>(smem + FF_RING);
    if (tid < 2 * FF_C) tg[tid] = tid < FF_C ? p.gamma[tid] : p.beta[tid - FF_C];

    // ---- this lane's token: its 256 input channels as B fragments (k-step i: channels 16 i + 8 h .. + 7)
    const long tok = (long)blockIdx.x * 256 + wave * 32 + lq;
    const bool ok = tok < p.M;
    const long tc = ok ? tok : p.M - 1;
    ff_u32x4 xf[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        xf[i] = *reinterpret_cast<const ff_u32x4*>(p.X1 + tc * p.ld1 + (2 * i + h) * 8);
        xf[8 + i] = *reinterpret_cast<const ff_u32x4*>(p.X2 + tc * p.ld2 + (2 * i + h) * 8);
    }
    // round 4: the compiler's wait for these loads belongs HERE.  Left to the first use it sat inside the chunk loop -- vmcnt(15) ..
    // vmcnt(0) between the MFMAs of EVERY iteration -- where it drained the weight ring the compiler cannot see (the LDS-DMAs are
    // inline asm)
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(xf[i]));

    // ---- a chunk = 24 contiguous 1-KB pieces (16 of W0, 8 of W2); wave w moves pieces 3 w .. 3 w + 2
    auto issue = [&](int c) {
        const unsigned base = lds0 + (c % FF_NST) * FF_SLOT;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int pc = 3 * wave + j;
            if (pc < 16) ff_dma16(base + pc * 1024, (unsigned)(c * FF_W0C + pc * 1024 + lane * 16), rs0);
            else ff_dma16(base + pc * 1024, (unsigned)(c * FF_W2C + (pc - 16) * 1024 + lane * 16), rs2);
        }
    };
    issue(0);
    issue(1);
    issue(2);

    f32x16 oacc[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;

    for (int c = 0; c < FF_NCH; ++c) {
        // chunk c has landed once all but the pieces of the (up to two) younger chunks are done
        if (c + 2 < FF_NCH) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (c + 1 < FF_NCH) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // ... for every wave; everyone has left chunk c - 1's slot
        __builtin_amdgcn_sched_barrier(0);
        if (c + 3 < FF_NCH) issue(c + 3);

        const char* w0 = smem + (c % FF_NST) * FF_SLOT;
        const char* w2 = w0 + FF_W0C;
        // ---- H^T chunk = W0[chunk] X^T
        f32x16 hacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) hacc[r] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint4 wf = *reinterpret_cast<const uint4*>(w0 + i * 1024 + lane * 16);
            hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, xf[i]), hacc, 0, 0, 0);
        }
        // ---- GELU; registers 8 sp .. 8 sp + 7 = hidden channels 16 sp + 8 (j >> 2) + 4 h + (j & 3) of the chunk: the B operand
        bf16x8 pf[2];
#pragma unroll
        for (int sp = 0; sp < 2; ++sp)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[sp][j] = (bf16_t)gelu_t<bf16_t>(hacc[8 * sp + j]);
        // ---- out^T += W2[:, chunk] H^T chunk
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const uint4 wf = *reinterpret_cast<const uint4*>(w2 + (d * 2 + sp) * 1024 + lane * 16);
                oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf), pf[sp], oacc[d], 0, 0, 0);
            }
    }

    // ---- LayerNorm over the token's 128 outputs (this lane: channels 32 d + 8 g + 4 h + j; the other half: lane ^ 32)
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s1 += oacc[d][r];
            s2 = fmaf(oacc[d][r], oacc[d][r], s2);
        }
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    const float mean = s1 * (1.0f / FF_C);
    const float rstd = rsqrtf(fmaxf(s2 * (1.0f / FF_C) - mean * mean, 0.f) + p.eps);
    // Round 4: whole lines out.  A lane holds 4 channels (8 bytes) of every 8-channel group and its partner (lane ^ 32) the other 4;
    // stored as they were, every 128-byte line of the output took sixteen 8-byte partial writes (PMC WRITE_SIZE 50.9 MB for 15.9 MB
    // of output); with the pair trading halves (v_permlane32_swap: 16 bytes per lane, second form of the round) still 33.7 MB.  Now
    // the workgroup's 256 x 128 outputs go through an image in the drained weight ring (row pitch 256 B, 16-byte chunk c of row q
    // at c ^ (q & 15)) and leave as whole 256-byte rows, 16 lanes each.  The residual is read by the token's own lane pair before
    // the barrier, so Res may alias Out.
    __builtin_amdgcn_s_barrier();                          // every wave is out of the last chunk's slot
    __builtin_amdgcn_sched_barrier(0);
    {
        const bf16_t* rp = p.Res ? p.Res + tc * p.ldr : nullptr;
        const int q = wave * 32 + lq;
        char* ib = smem + q * 256;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = 32 * d + 8 * g + 4 * h;
                const float4 gm = *reinterpret_cast<const float4*>(tg + ch), bt = *reinterpret_cast<const float4*>(tg + FF_C + ch);
                const float gv[4] = {gm.x, gm.y, gm.z, gm.w}, bv[4] = {bt.x, bt.y, bt.z, bt.w};
                float rv[4] = {0.f, 0.f, 0.f, 0.f};
                if (rp) {
                    const bf16x4 r4 = *reinterpret_cast<const bf16x4*>(rp + ch);
#pragma unroll
                    for (int j = 0; j < 4; ++j) rv[j] = (float)r4[j];
                }
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (bf16_t)(fmaf((oacc[d][4 * g + j] - mean) * rstd, gv[j], bv[j]) + rv[j]);
                *reinterpret_cast<bf16x4*>(ib + (((4 * d + g) ^ (q & 15)) * 16) + 8 * h) = o;
            }
    }
    __syncthreads();
    {
        const int c = tid & 15;
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int q = 32 * pass + (tid >> 4);
            const long t2 = (long)blockIdx.x * 256 + q;
            const ff_u32x4 v = *reinterpret_cast<const ff_u32x4*>(smem + q * 256 + ((c ^ (q & 15)) * 16));
            if (t2 < p.M) *reinterpret_cast<ff_u32x4*>(p.Out + t2 * p.ldo + 8 * c) = v;
        }
    }
}

}  // namespace

// X1, X2: bf16 [M][>= 128] (row strides ld1, ld2); W0p / W2p: the FRAGMENT-ORDER packs of mlp[0].weight [1024][256] and
// mlp[2].weight [128][1024] (ops.ffn_block_packs); gamma, beta: f32 [128]; Res (may be NULL, may alias Out), Out: bf16 [M][>= 128].
extern "C" int emip_ffn_block(const void* X1, long ld1, const void* X2, long ld2, const void* W0p, const void* W2p,
                              const float* gamma, const float* beta, float eps, const void* Res, long ldr, void* Out, long ldo,
                              long M, void* stream) {
    EMIP_REQUIRE(X1 && X2 && W0p && W2p && gamma && beta && Out && M > 0 && M < 2147483647L && eps > 0.f);
    EMIP_REQUIRE(ld1 >= FF_C && ld2 >= FF_C && ldo >= FF_C && ((ld1 | ld2 | ldo) & 7) == 0 && (!Res || (ldr >= FF_C && (ldr & 3) == 0)));
    EMIP_REQUIRE(aligned16(X1) && aligned16(X2) && aligned16(W0p) && aligned16(W2p) && aligned16(gamma) && aligned16(beta) &&
                 aligned16(Out) && (!Res || (reinterpret_cast<uintptr_t>(Res) & 7u) == 0));
    FfArgs a{};
    a.X1 = (const bf16_t*)X1; a.X2 = (const bf16_t*)X2; a.W0p = (const bf16_t*)W0p; a.W2p = (const bf16_t*)W2p;
    a.gamma = gamma; a.beta = beta; a.Res = (const bf16_t*)Res; a.Out = (bf16_t*)Out;
    a.ld1 = ld1; a.ld2 = ld2; a.ldr = ldr; a.ldo = ldo; a.M = (int)M; a.eps = eps;
    static bool attr = false;
    if (!attr) {
        attr = true;
        if (hipFuncSetAttribute((const void*)ffn_block_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS) != hipSuccess)
            return EMIP_E_LAUNCH;
    }
    hipLaunchKernelGGL(ffn_block_kernel, dim3((unsigned)((M + 255) / 256)), dim3(512), FF_LDS, (hipStream_t)stream, a);
    return emip_launch_status();
}
