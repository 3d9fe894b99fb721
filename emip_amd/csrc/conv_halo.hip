// 3 x 3 convolution (stride 1, zero padding 1, 64 -> 64 channels, bf16) of the GMFlow CNN encoder's first level as a DIRECT
// convolution on an LDS-resident halo tile, with the InstanceNorm + ReLU of the producing layer applied while the tile is
// staged and the statistics of the result taken in the epilogue (/root/reference/model/EMIP_short/motion/gmflow/backbone.py:
// 39-69 ResidualBlock: conv1 -> norm1 -> relu -> conv2 -> norm2 -> relu, :154-192 CNNEncoder).
//
// The implicit-GEMM form (gemm8.hip, CONV) moves the input tile of every tap L2 -> LDS: nine passes over the same pixels
// (360 KB per 256 output pixels), and InstanceNorm, which needs the whole image's statistics, costs a statistics pass and a
// normalise pass over the tensor between any two convolutions (profiles/pmc_traffic.json: 3.2 GB of a 21-GB step).  Here:
//   * persistent workgroups (one per CU, 8 waves); all nine taps' weights stay in LDS for the whole launch in MFMA A-fragment
//     order (72 pieces of 1 KB: conflict-free ds_read_b128 at lane * 16);
//   * an output tile is 16 x 16 pixels; its 18 x 18 x 64 input halo (41.5 KB) arrives by LDS-DMA (buffer_load ... lds) into one
//     of two buffers while the previous tile is computed; pixels outside the image come back as zeros from the buffer
//     descriptor's range check; the 16-byte chunk c of halo pixel P lies at chunk c ^ ((P >> 1) & 7) of its 128-byte row, so
//     that the 16 lanes of a b128 read (16 neighbouring pixels, same channel chunk) cover all 64 banks;
//   * NORM: before the tile is used every thread normalises 5 chunks of it in place -- relu((x - mean) * rstd) with the
//     per-(image, channel) mean / rstd from the producer's sums -- except the chunks of pixels outside the image, which stay
//     zero (the reference pads the NORMALISED tensor);
//   * a wave owns 2 rows x 16 pixels x 64 output channels: per tap 4 B fragments (its pixels shifted by the tap, 16 input
//     channels each) and 8 weight fragments feed 8 MFMAs 32x32x16; 72 MFMAs per tile and wave;
//   * epilogue: accumulators -> bf16 -> an LDS image of the tile in the halo buffer just read: whole 128-byte lines leave for
//     memory; STATS: per-channel (sum, sum of squares) of the ROUNDED values from columns of that image (thread = channel x
//     eighth of the tile, fixed order), carried in registers over the workgroup's CONTIGUOUS tiles of an image -> one partial
//     per (image, workgroup) in memory (device-scope store) + a ticket per image; the workgroup that draws an image's last
//     ticket adds the partials in workgroup order in f64 and stores the sums the normalising consumers read.  No
//     atomics on the sums, no waiting on other workgroups, no fences (gemm8.hip's statistics combine has the measurements).
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned ch_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 ch_bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ i32x4 ch_rsrc(const void* ptr, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
    return i32x4{(int)(unsigned)a, (int)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void ch_dma16(unsigned lds_dst, unsigned voff, i32x4 rs) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_dst), "v"(voff), "s"(rs)
        : "memory");
}

constexpr unsigned CH_OOB = 0x80000000u;
constexpr int CH_C = 64, CH_T = 16, CH_HS = CH_T + 2, CH_HPX = CH_HS * CH_HS;          // 18 x 18 = 324 halo pixels
constexpr int CH_HINS = (CH_HPX + 7) / 8;                                              // 41 DMA instructions of 8 pixels
constexpr int CH_HALO = CH_HINS * 1024;                                                // 41 984 B
constexpr int CH_W = 9 * 2 * 4 * 1024;                                                 // 73 728 B
constexpr int CH_OFF_H = CH_W, CH_OFF_S = CH_W + 2 * CH_HALO;                          // statistics scratch behind the halos
constexpr int CH_LDS = CH_OFF_S + 8 * CH_C * 2 * 4 + 64;                               // + [8 waves][64][2] f32 + the ticket
static_assert(CH_LDS <= 160 * 1024, "LDS");

struct ChArgs {
    const bf16_t* X;        // [B, H, W, ldx >= 64]
    const bf16_t* Wp;       // fragment-order weights (ops.conv3x3_halo_pack): [9 taps][2][4][64 lanes][8]
    bf16_t* Y;              // [B, H, W, ldy >= 64]
    long ldx, ldy;
    int B, H, W, tx, ty;    // tiles per row / column
    unsigned x_bytes;
    const double* in_sums;  // NORM: [B][64][2] (sum, sum of squares) of the INPUT tensor's channels per image
    float in_eps;
    float* part;            // STATS: [B * ty * tx][64][2] tile partials
    unsigned* cnt;          // STATS: [B] tickets (zero at launch, zero at exit)
    double* out_sums;       // STATS: [B][64][2]
};

template <bool NORM, bool STATS>
__global__ __launch_bounds__(512) void conv_halo_kernel(const ChArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 31, h = lane >> 5;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)smem;
    const i32x4 rsX = ch_rsrc(p.X, p.x_bytes);
    const i32x4 rsW = ch_rsrc(p.Wp, (unsigned)CH_W);
    const int per_img = p.tx * p.ty, ntiles = p.B * per_img;

    // ---- halo of tile t -> buffer b: instruction i moves halo pixels 8 i .. 8 i + 7 (one 16-byte chunk per lane)
    auto issue = [&](int t, int b) {
        const int img = t / per_img, r = t - img * per_img;
        const int y0 = (r / p.tx) * CH_T - 1, x0 = (r % p.tx) * CH_T - 1;
        for (int i = wave; i < CH_HINS; i += 8) {
            const int P = 8 * i + (lane >> 3);
            const int hy = P / CH_HS, hx = P - hy * CH_HS;
            const int gy = y0 + hy, gx = x0 + hx;
            const bool in = P < CH_HPX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            const unsigned c = (unsigned)((lane & 7) ^ ((P >> 1) & 7));           // the logical chunk this slot holds
            const unsigned off = in ? (unsigned)((((long)img * p.H + gy) * p.W + gx) * p.ldx * 2) + c * 16u : CH_OOB;
            ch_dma16((unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + CH_OFF_H + b * CH_HALO + i * 1024)), off, rsX);
        }
    };
    // ---- prologue: the weights (72 pieces of 1 KB, 9 per wave) and the first halo
    for (int i = wave; i < 72; i += 8)
        ch_dma16((unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + i * 1024)), (unsigned)(i * 1024 + lane * 16), rsW);
    // workgroup b walks the CONTIGUOUS tiles [b N / G, (b + 1) N / G): neighbouring halos share rows in L2, and the tiles of a
    // workgroup lie in one or two images, so the channel sums are carried in registers over tiles and published per image
    const int G = gridDim.x;
    const int t_begin = (int)(((long)blockIdx.x * ntiles) / G), t_end = (int)(((long)(blockIdx.x + 1) * ntiles) / G);
    auto owner = [&](int t) { return (int)((((long)t + 1) * G - 1) / ntiles); };       // the workgroup whose range holds tile t
    if (t_begin < t_end) issue(t_begin, 0);

    // this lane's output pixel inside a tile and its halo pixel for tap (0, 0)
    const int oy = 2 * wave + (px >> 4), ox = px & 15;
    const int P00 = oy * CH_HS + ox;
    float run1 = 0.f, run2 = 0.f;            // STATS: thread (channel tid & 63, eighth tid >> 6 of every tile) over this image's tiles
    int tab_img = -1;                        // NORM: the image whose (mean, rstd) table is in LDS

    int buf = 0;
    for (int t = t_begin; t < t_end; ++t, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                       // the halo (and the weights) have landed; everyone has left the other buffer
        if (t + 1 < t_end) issue(t + 1, buf ^ 1);
        const int img = t / per_img, r = t - img * per_img;
        const int ty0 = (r / p.tx) * CH_T, tx0 = (r % p.tx) * CH_T;
        char* hb = smem + CH_OFF_H + buf * CH_HALO;

        float* red = reinterpret_cast<float*>(smem + CH_OFF_S);                      // [8][64][2] f32 scratch
        if (NORM) {
            // the image's (mean, rstd) per input channel -> LDS, then in place: relu((x - mean) * rstd) on the 16-byte slots of
            // pixels inside the image (2 592 slots; pixels outside stay zero: the reference pads the NORMALISED tensor)
            if (img != tab_img && tid < CH_C) {          // (a workgroup's tiles lie in one or two images: two or three table builds)
                const double* sm = p.in_sums + ((long)img * CH_C + tid) * 2;
                const double n = (double)p.H * (double)p.W;
                const float mu = (float)(sm[0] / n);
                const float var = fmaxf((float)(sm[1] / n) - mu * mu, 0.f);
                *reinterpret_cast<float2*>(red + 2 * tid) = make_float2(mu, rsqrtf(var + p.in_eps));
            }
            if (img != tab_img) __syncthreads();
            tab_img = img;
            {
                // a thread keeps ONE logical chunk (8 channels: their mean / rstd in registers) and walks the pixels
                const int c = tid & 7;
                float mr[16];
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(mr + 4 * j) = *reinterpret_cast<const float4*>(red + 16 * c + 4 * j);
                for (int P = tid >> 3; P < CH_HPX; P += 64) {
                    const int hy = P / CH_HS, hx = P - hy * CH_HS;
                    const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
                    if ((unsigned)gy >= (unsigned)p.H || (unsigned)gx >= (unsigned)p.W) continue;
                    char* sp = hb + P * 128 + ((c ^ ((P >> 1) & 7)) * 16);
                    const ch_u32x4 v = *reinterpret_cast<const ch_u32x4*>(sp);
                    unsigned vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float x0 = fmaxf((__uint_as_float(vw[j] << 16) - mr[4 * j]) * mr[4 * j + 1], 0.f);
                        const float x1 = fmaxf((__uint_as_float(vw[j] & 0xFFFF0000u) - mr[4 * j + 2]) * mr[4 * j + 3], 0.f);
                        ch_bf16x2 o;
                        o[0] = (bf16_t)x0;
                        o[1] = (bf16_t)x1;
                        vw[j] = __builtin_bit_cast(unsigned, o);
                    }
                    *reinterpret_cast<ch_u32x4*>(sp) = ch_u32x4{vw[0], vw[1], vw[2], vw[3]};
                }
            }
            __syncthreads();
        }

        // ---- 9 taps x (4 B fragments, 8 weight fragments, 8 MFMAs)
        f32x16 acc[2];
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) acc[d][r2] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int P = P00 + (tap / 3) * CH_HS + (tap % 3);
            const char* pb = hb + P * 128;
            const int sw = (P >> 1) & 7;
            ch_u32x4 bfr[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) bfr[ks] = *reinterpret_cast<const ch_u32x4*>(pb + (((2 * ks + h) ^ sw) * 16));
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const ch_u32x4 af = *reinterpret_cast<const ch_u32x4*>(smem + ((tap * 2 + d) * 4 + ks) * 1024 + lane * 16);
                    acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bfr[ks]),
                                                                    acc[d], 0, 0, 0);
                }
        }

        // ---- epilogue: register 4 g + j of tile d = output channel 32 d + 8 g + 4 h + j of this lane's pixel.  The tile goes
        // through an LDS image [256 pixels][64 ch] bf16 in the halo buffer just read (same chunk swizzle): whole 128-byte lines
        // leave for memory, and the channel sums read columns of it
        __syncthreads();                                   // every wave is done with this halo buffer
        {
            const int q = 32 * wave + px, sw = (q >> 1) & 7;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (bf16_t)acc[d][4 * g + j];
                    *reinterpret_cast<bf16x4*>(hb + q * 128 + (((4 * d + g) ^ sw) * 16) + 8 * h) = o;
                }
        }
        __syncthreads();
        {
            const int c = tid & 7;
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int q = 64 * pass + (tid >> 3);
                const ch_u32x4 v = *reinterpret_cast<const ch_u32x4*>(hb + q * 128 + ((c ^ ((q >> 1) & 7)) * 16));
                bf16_t* op = p.Y + (((long)img * p.H + ty0 + (q >> 4)) * p.W + tx0 + (q & 15)) * p.ldy;
                *reinterpret_cast<ch_u32x4*>(op + 8 * c) = v;
            }
        }
        if (STATS) {
            // thread (channel, eighth of the tile): 32 pixels of its channel in pixel order, carried over the image's tiles
            const int ch = tid & 63, part = tid >> 6;
#pragma unroll 8
            for (int k = 0; k < 32; ++k) {
                const int q = 32 * part + k;
                const unsigned short u = *reinterpret_cast<const unsigned short*>(hb + q * 128 + (((ch >> 3) ^ ((q >> 1) & 7)) * 16) + 2 * (ch & 7));
                const float x = __uint_as_float((unsigned)u << 16);
                run1 += x;
                run2 = fmaf(x, x, run2);
            }
            if (t + 1 == t_end || (t + 1) / per_img != img) {
                // this workgroup's last tile of the image: eighths in order -> ONE partial per (image, workgroup) in memory
                // (device-scope store) + a ticket; the workgroup that draws the image's last ticket adds the partials in
                // workgroup order, in f64.  How many workgroups touch an image follows from the partition: no communication
                const int lo = img * per_img, b0 = owner(lo), nb = owner(lo + per_img - 1) - b0 + 1;
                *reinterpret_cast<float2*>(red + (part * CH_C + ch) * 2) = make_float2(run1, run2);
                run1 = run2 = 0.f;
                tab_img = -1;                                  // (the scratch held the NORM table)
                __syncthreads();
                float* pp = p.part + ((long)img * per_img) * CH_C * 2;                 // [<= per_img contributions][64][2]
                if (tid < 2 * CH_C) {
                    float s = 0.f;
#pragma unroll
                    for (int w = 0; w < 8; ++w) s += red[w * CH_C * 2 + tid];
                    __hip_atomic_store(reinterpret_cast<unsigned*>(pp + (long)((int)blockIdx.x - b0) * CH_C * 2 + tid), __float_as_uint(s),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __syncthreads();
                unsigned* tick = reinterpret_cast<unsigned*>(smem + CH_OFF_S + 8 * CH_C * 2 * 4);
                if (tid == 0) *tick = atomicAdd(p.cnt + img, 1u);
                __syncthreads();
                if (*tick == (unsigned)(nb - 1)) {
                    if (tid < 2 * CH_C) {
                        double s = 0.0;
                        for (int k = 0; k < nb; k += 8) {          // device-scope loads, eight in flight (a chain of them waits a
                            float v[8];                            // memory round trip EACH)
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float* a = pp + (long)min(k + e, nb - 1) * CH_C * 2 + tid;
                                asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v[e]) : "v"(a) : "memory");
                            }
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                            for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(v[e]));
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                if (k + e < nb) s += (double)v[e];
                        }
                        p.out_sums[(long)img * CH_C * 2 + tid] = s;
                    }
                    if (tid == 0) __hip_atomic_store(p.cnt + img, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}


// ---- the wider levels (96 and 128 channels: 88 x 88 and 44 x 44 maps): nine taps of weights no longer fit beside the halo, so
// they STREAM: one tap (C x C, 18 / 32 KB in fragment order) per ring slot, two slots, tap k + 1 in flight while tap k is
// used, one s_barrier per tap; with 96 channels the NEXT tile's halo rides along in eight pieces (one DMA instruction per
// wave and tap) into the second halo buffer; with 128 channels one halo buffer is all that fits (at 32 images a CU has one
// tile).  Tiles are 11 x 22 pixels (242 of the 256 pixel lanes): 176, 88 and 44 are multiples of both, and 32 images give
// 4 096 / 1 024 / 256 tiles = 16 / 4 / 1 per CU exactly.  Chunk swizzles per pixel pitch: 128 B: c ^ ((P >> 1) & 7),
// 192 B: c ^ ((P >> 2) & 3), 256 B: c ^ (P & 15) -- each makes the 16 lanes of a b128 read cover the 64 banks.
constexpr int CS_TH = 11, CS_TW = 22, CS_TPX = CS_TH * CS_TW, CS_HY = CS_TH + 2, CS_HX = CS_TW + 2, CS_HPX = CS_HY * CS_HX;   // 242, 13 x 24 = 312

template <int C, bool DBLP>
struct CsGeo {
    static constexpr int NCH = C / 8, CB = 2 * C, NKS = C / 16, NCT = C / 32;
    static constexpr int TAPW = C * C * 2, NWI = (TAPW / 1024 + 7) / 8;              // weight DMA instructions per wave and tap
    static constexpr int HINS = (CS_HPX * NCH + 63) / 64, HALO = HINS * 1024;
    static constexpr bool DBL = DBLP;                                               // two halo buffers (the next tile's halo rides along)
    static constexpr int OFF_W = (DBL ? 2 : 1) * HALO, OFF_S = OFF_W + 2 * TAPW;    // scratch: 1 KB table / ticket + 1 KB dump
    static constexpr int LDS = OFF_S + 2048;
    static constexpr int NP = 512 / C, PP = (CS_TPX + NP - 1) / NP;                 // statistics: parts of the tile, pixels per part
    static_assert(LDS <= 160 * 1024 && CS_TPX * CB + NP * C * 8 <= HALO && HINS <= 64 + (DBL ? 0 : 64), "LDS");
    __device__ static __forceinline__ int swz(int P) { return C == 64 ? (P >> 1) & 7 : (C == 96 ? (P >> 2) & 3 : P & 15); }
};

template <int C, bool NORM, bool STATS, bool DBLP>
__global__ __launch_bounds__(512) void conv_halo_stream_kernel(const ChArgs p) {
    typedef CsGeo<C, DBLP> Gm;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 31, h = lane >> 5;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)smem;
    const i32x4 rsX = ch_rsrc(p.X, p.x_bytes);
    const i32x4 rsW = ch_rsrc(p.Wp, (unsigned)(9 * Gm::TAPW));
    const int per_img = p.tx * p.ty, ntiles = p.B * per_img;
    const unsigned dump = lds0 + Gm::OFF_S + 1024;

    // halo instruction i of tile t -> buffer b (slot s = 64 i + lane: pixel s / NCH, physical chunk s % NCH)
    auto issue_h = [&](int t, int b, int i) {
        const bool live = t < ntiles && i < Gm::HINS;
        const int tt = min(t, ntiles - 1);
        const int img = tt / per_img, r = tt - img * per_img;
        const int y0 = (r / p.tx) * CS_TH - 1, x0 = (r % p.tx) * CS_TW - 1;
        const int sl = 64 * i + lane, P = sl / Gm::NCH, cp = sl - P * Gm::NCH;
        const int hy = P / CS_HX, hx = P - hy * CS_HX;
        const int gy = y0 + hy, gx = x0 + hx;
        const bool in = live && P < CS_HPX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        const unsigned c = (unsigned)(cp ^ Gm::swz(P));
        const unsigned off = in ? (unsigned)((((long)img * p.H + gy) * p.W + gx) * p.ldx * 2) + c * 16u : CH_OOB;
        const unsigned dst = live ? lds0 + b * Gm::HALO + i * 1024 : dump;
        ch_dma16((unsigned)__builtin_amdgcn_readfirstlane((int)dst), off, rsX);
    };
    // the weights of tap k -> ring slot: NWI instructions per wave (pieces beyond the tap go to the dump: the counts the
    // vmcnt waits rely on are the same for every wave)
    auto issue_w = [&](int k, int slot) {
#pragma unroll
        for (int m = 0; m < Gm::NWI; ++m) {
            const int j = wave + 8 * m;
            const bool live = j < Gm::TAPW / 1024;
            const unsigned dst = live ? lds0 + Gm::OFF_W + slot * Gm::TAPW + j * 1024 : dump;
            ch_dma16((unsigned)__builtin_amdgcn_readfirstlane((int)dst), live ? (unsigned)(k * Gm::TAPW + j * 1024 + lane * 16) : CH_OOB, rsW);
        }
    };

    const int G = gridDim.x;
    const int t_begin = (int)(((long)blockIdx.x * ntiles) / G), t_end = (int)(((long)(blockIdx.x + 1) * ntiles) / G);
    auto owner = [&](int t) { return (int)((((long)t + 1) * G - 1) / ntiles); };
    if (t_begin < t_end) {
        for (int i = wave; i < Gm::HINS; i += 8) issue_h(t_begin, 0, i);
        issue_w(0, 0);
    }
    // this lane's output pixel (lanes 242 .. 255 of the tile repeat the last pixel and store nothing)
    const int q = min(32 * wave + px, CS_TPX - 1);
    const bool q_ok = 32 * wave + px < CS_TPX;
    const int oy = q / CS_TW, ox = q - oy * CS_TW;
    const int P00 = oy * CS_HX + ox;
    float run1 = 0.f, run2 = 0.f;
    float* tab = reinterpret_cast<float*>(smem + Gm::OFF_S);            // NORM: [C][2] (mean, rstd); publish: the ticket at tab[0]
    int tab_img = -1;

    int buf = 0;
    int wslot = 0;                                                      // ring slot of the tap about to be used
    for (int t = t_begin; t < t_end; ++t, buf ^= (Gm::DBL ? 1 : 0)) {
        const int img = t / per_img, r = t - img * per_img;
        const int ty0 = (r / p.tx) * CS_TH, tx0 = (r % p.tx) * CS_TW;
        char* hb = smem + buf * Gm::HALO;
        if (!Gm::DBL && t > t_begin) {
            // one halo buffer: everyone is out of the image of the previous tile -> load this tile's halo now
            __syncthreads();
            for (int i = wave; i < Gm::HINS; i += 8) issue_h(t, 0, i);
        }
        f32x16 acc[Gm::NCT];
#pragma unroll
        for (int d = 0; d < Gm::NCT; ++d)
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) acc[d][r2] = 0.f;
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            // tap's weights (issued one tap ago) have landed once everything but the halo piece issued behind them is done
            if (tap == 0 || !Gm::DBL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            __syncthreads();
            // next tap's weights (the first tap of the next tile behind the last one) + a piece of the next tile's halo
            if (tap < 8 || t + 1 < t_end) issue_w(tap < 8 ? tap + 1 : 0, wslot ^ 1);
            else issue_w(0, wslot ^ 1);                                  // (nothing follows: harmless, keeps the counts)
            if (Gm::DBL && tap < 8) issue_h(t + 1 < t_end ? t + 1 : ntiles, buf ^ 1, wave + 8 * tap);
            if (NORM && tap == 0) {
                if (img != tab_img && tid < C) {
                    const double* sm = p.in_sums + ((long)img * C + tid) * 2;
                    const double n = (double)p.H * (double)p.W;
                    const float mu = (float)(sm[0] / n);
                    const float var = fmaxf((float)(sm[1] / n) - mu * mu, 0.f);
                    *reinterpret_cast<float2*>(tab + 2 * tid) = make_float2(mu, rsqrtf(var + p.in_eps));
                }
                if (img != tab_img) __syncthreads();
                tab_img = img;
                {
                    // a thread keeps ONE logical chunk (its 8 channels' mean / rstd in registers) and walks the pixels
                    constexpr int PPT = 512 / Gm::NCH;                   // pixels per sweep (threads beyond PPT * NCH idle)
                    const int c = tid % Gm::NCH, pr = tid / Gm::NCH;
                    float mr[16];
#pragma unroll
                    for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(mr + 4 * j) = *reinterpret_cast<const float4*>(tab + 16 * c + 4 * j);
                    for (int P = pr; P < CS_HPX && pr < PPT; P += PPT) {
                        const int hy = P / CS_HX, hx = P - hy * CS_HX;
                        const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
                        if ((unsigned)gy >= (unsigned)p.H || (unsigned)gx >= (unsigned)p.W) continue;
                        char* sp = hb + P * Gm::CB + ((c ^ Gm::swz(P)) * 16);
                        const ch_u32x4 v = *reinterpret_cast<const ch_u32x4*>(sp);
                        unsigned vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float x0 = fmaxf((__uint_as_float(vw[j] << 16) - mr[4 * j]) * mr[4 * j + 1], 0.f);
                            const float x1 = fmaxf((__uint_as_float(vw[j] & 0xFFFF0000u) - mr[4 * j + 2]) * mr[4 * j + 3], 0.f);
                            ch_bf16x2 o;
                            o[0] = (bf16_t)x0;
                            o[1] = (bf16_t)x1;
                            vw[j] = __builtin_bit_cast(unsigned, o);
                        }
                        *reinterpret_cast<ch_u32x4*>(sp) = ch_u32x4{vw[0], vw[1], vw[2], vw[3]};
                    }
                }
                __syncthreads();
            }
            const int P = P00 + (tap / 3) * CS_HX + (tap % 3);
            const char* pb = hb + P * Gm::CB;
            const int sw = Gm::swz(P);
            const char* wb = smem + Gm::OFF_W + wslot * Gm::TAPW + lane * 16;
            ch_u32x4 bfr[Gm::NKS];
#pragma unroll
            for (int ks = 0; ks < Gm::NKS; ++ks) bfr[ks] = *reinterpret_cast<const ch_u32x4*>(pb + (((2 * ks + h) ^ sw) * 16));
#pragma unroll
            for (int d = 0; d < Gm::NCT; ++d)
#pragma unroll
                for (int ks = 0; ks < Gm::NKS; ++ks) {
                    const ch_u32x4 af = *reinterpret_cast<const ch_u32x4*>(wb + (d * Gm::NKS + ks) * 1024);
                    acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bfr[ks]),
                                                                    acc[d], 0, 0, 0);
                }
            wslot ^= 1;
        }

        // ---- epilogue through an LDS image [242 pixels][C] bf16 in the halo buffer just read (same swizzle)
        __syncthreads();
        if (q_ok) {
            const int sw = Gm::swz(q);
#pragma unroll
            for (int d = 0; d < Gm::NCT; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (bf16_t)acc[d][4 * g + j];
                    *reinterpret_cast<bf16x4*>(hb + q * Gm::CB + (((4 * d + g) ^ sw) * 16) + 8 * h) = o;
                }
        }
        __syncthreads();
        for (int s2 = tid; s2 < CS_TPX * Gm::NCH; s2 += 512) {
            const int qq = s2 / Gm::NCH, c = s2 - qq * Gm::NCH;
            const ch_u32x4 v = *reinterpret_cast<const ch_u32x4*>(hb + qq * Gm::CB + ((c ^ Gm::swz(qq)) * 16));
            const int yy = qq / CS_TW, xx = qq - yy * CS_TW;
            bf16_t* op = p.Y + (((long)img * p.H + ty0 + yy) * p.W + tx0 + xx) * p.ldy;
            *reinterpret_cast<ch_u32x4*>(op + 8 * c) = v;
        }
        if (STATS) {
            const int ch = tid % C, part = tid / C;
            if (part < Gm::NP) {
                const int q1 = min(CS_TPX, (part + 1) * Gm::PP);
                for (int qq = part * Gm::PP; qq < q1; ++qq) {
                    const unsigned short u = *reinterpret_cast<const unsigned short*>(hb + qq * Gm::CB + (((ch >> 3) ^ Gm::swz(qq)) * 16) + 2 * (ch & 7));
                    const float x = __uint_as_float((unsigned)u << 16);
                    run1 += x;
                    run2 = fmaf(x, x, run2);
                }
            }
            if (t + 1 == t_end || (t + 1) / per_img != img) {
                const int lo = img * per_img, b0 = owner(lo), nb = owner(lo + per_img - 1) - b0 + 1;
                __syncthreads();                                           // the column reads of the image are done
                float* red = reinterpret_cast<float*>(hb + CS_TPX * Gm::CB);   // [NP][C][2] behind the image
                if (part < Gm::NP) *reinterpret_cast<float2*>(red + (part * C + ch) * 2) = make_float2(run1, run2);
                run1 = run2 = 0.f;
                __syncthreads();
                float* pp = p.part + ((long)img * per_img) * C * 2;
                if (tid < 2 * C) {
                    float sacc = 0.f;
#pragma unroll
                    for (int w = 0; w < Gm::NP; ++w) sacc += red[w * C * 2 + tid];
                    __hip_atomic_store(reinterpret_cast<unsigned*>(pp + (long)((int)blockIdx.x - b0) * C * 2 + tid), __float_as_uint(sacc),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __syncthreads();
                unsigned* tick = reinterpret_cast<unsigned*>(tab);
                tab_img = -1;                                              // (the ticket overwrites the table's first word)
                if (tid == 0) *tick = atomicAdd(p.cnt + img, 1u);
                __syncthreads();
                if (*tick == (unsigned)(nb - 1)) {
                    if (tid < 2 * C) {
                        double sd = 0.0;
                        for (int k = 0; k < nb; k += 8) {
                            float v[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float* a = pp + (long)min(k + e, nb - 1) * C * 2 + tid;
                                asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v[e]) : "v"(a) : "memory");
                            }
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                            for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(v[e]));
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                if (k + e < nb) sd += (double)v[e];
                        }
                        p.out_sums[(long)img * C * 2 + tid] = sd;
                    }
                    if (tid == 0) __hip_atomic_store(p.cnt + img, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
            }
        }
    }
}

template <int C, bool DBLP>
int launch_stream(const ChArgs& a, bool norm, bool stats, int ntiles, hipStream_t s) {
    void (*fn)(const ChArgs) = norm ? (stats ? conv_halo_stream_kernel<C, true, true, DBLP> : conv_halo_stream_kernel<C, true, false, DBLP>)
                                    : (stats ? conv_halo_stream_kernel<C, false, true, DBLP> : conv_halo_stream_kernel<C, false, false, DBLP>);
    static bool attr[4] = {false, false, false, false};
    const int which = (norm ? 2 : 0) + (stats ? 1 : 0);
    constexpr int lds = CsGeo<C, DBLP>::LDS;
    if (!attr[which]) {
        if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return EMIP_E_LAUNCH;
        attr[which] = true;
    }
    const int slots = lds <= 80 * 1024 ? 512 : 256;                  // two workgroups per CU where the LDS allows
    hipLaunchKernelGGL(fn, dim3((unsigned)(ntiles < slots ? ntiles : slots)), dim3(512), lds, s, a);
    return emip_launch_status();
}

#ifdef EMIP_TUNING
int g_halo_mode = 0;     // 1: 64 channels on the streamed form (two halo buffers); 2: the same with ONE halo buffer, two workgroups per CU
#endif

// ---- the encoder's stem: 7 x 7, stride 2, zero padding 3, 3 (stored as 8) -> 64 channels, no bias (backbone.py:84,154-160), with
// the InstanceNorm sums of its output in the epilogue.  A pixel is one 16-byte chunk, so the halo of a 16 x 16 output tile --
// 37 x 37 input pixels -- is 22 KB; it is stored in column-parity planes ([row][column & 1][column >> 1]) so that the 16 lanes of
// a fragment read (16 neighbouring outputs = input columns two apart, same tap) are consecutive chunks.  K = 49 taps x 8 = 392,
// walked as 25 k-steps of two taps (the 50th is zero); the 50 weight fragments (50 KB) stay in LDS.  Epilogue and statistics
// as in conv_halo_kernel (the image buffers are sized for the 32-KB output tile).
constexpr int ST_HS = 37, ST_PL = 19, ST_SLOTS = ST_HS * 2 * ST_PL;                    // 37 rows x 2 parity planes x 19 columns = 1406 chunks
constexpr int ST_HINS = (ST_SLOTS + 63) / 64, ST_BUF = 32 * 1024;                      // 22 DMA instructions; buffer = the 32-KB output image
constexpr int ST_NKS = 25, ST_W = 2 * ST_NKS * 1024;                                   // 51 200 B of weights
constexpr int ST_OFF_H = ST_W, ST_OFF_S = ST_W + 2 * ST_BUF, ST_LDS = ST_OFF_S + 8 * CH_C * 2 * 4 + 64;
static_assert(ST_HINS * 1024 <= ST_BUF && ST_LDS <= 160 * 1024, "LDS");

template <bool STATS>
__global__ __launch_bounds__(512) void conv_stem_kernel(const ChArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 31, h = lane >> 5;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)smem;
    const i32x4 rsX = ch_rsrc(p.X, p.x_bytes);
    const i32x4 rsW = ch_rsrc(p.Wp, (unsigned)ST_W);
    const int Ho = p.H / 2, Wo = p.W / 2;                     // output map (p.H, p.W: the input image)
    const int per_img = p.tx * p.ty, ntiles = p.B * per_img;

    auto issue = [&](int t, int b) {
        const int img = t / per_img, r = t - img * per_img;
        const int y0 = 2 * (r / p.tx) * CH_T - 3, x0 = 2 * (r % p.tx) * CH_T - 3;
        for (int i = wave; i < ST_HINS; i += 8) {
            const int sl = 64 * i + lane;                     // chunk (row hy, parity pr, column pair cc) = input pixel (hy, 2 cc + pr)
            const int hy = sl / (2 * ST_PL), rem = sl - hy * 2 * ST_PL;
            const int pr = rem / ST_PL, cc = rem - pr * ST_PL, hx = 2 * cc + pr;
            const int gy = y0 + hy, gx = x0 + hx;
            const bool in = sl < ST_SLOTS && hx < ST_HS && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            const unsigned off = in ? (unsigned)((((long)img * p.H + gy) * p.W + gx) * p.ldx * 2) : CH_OOB;
            ch_dma16((unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + ST_OFF_H + b * ST_BUF + i * 1024)), off, rsX);
        }
    };
    for (int i = wave; i < 2 * ST_NKS; i += 8)
        ch_dma16((unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + i * 1024)), (unsigned)(i * 1024 + lane * 16), rsW);
    const int G = gridDim.x;
    const int t_begin = (int)(((long)blockIdx.x * ntiles) / G), t_end = (int)(((long)(blockIdx.x + 1) * ntiles) / G);
    auto owner = [&](int t) { return (int)((((long)t + 1) * G - 1) / ntiles); };
    if (t_begin < t_end) issue(t_begin, 0);

    const int oy = 2 * wave + (px >> 4), ox = px & 15;
    // this lane's two taps of k-step ks: tap = 2 ks + h -> byte offset of its input pixel's chunk in a halo buffer
    float run1 = 0.f, run2 = 0.f;
    float* red = reinterpret_cast<float*>(smem + ST_OFF_S);

    int buf = 0;
    for (int t = t_begin; t < t_end; ++t, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < t_end) issue(t + 1, buf ^ 1);
        const int img = t / per_img, r = t - img * per_img;
        const int ty0 = (r / p.tx) * CH_T, tx0 = (r % p.tx) * CH_T;
        char* hb = smem + ST_OFF_H + buf * ST_BUF;

        f32x16 acc[2];
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) acc[d][r2] = 0.f;
#pragma unroll
        for (int ks = 0; ks < ST_NKS; ++ks) {
            const int tap = min(2 * ks + h, 48);              // (tap 49 does not exist: its weights are zero, any pixel will do)
            const int ky = tap / 7, kx = tap - ky * 7;
            const int hy = 2 * oy + ky, hx = 2 * ox + kx;
            const ch_u32x4 bf = *reinterpret_cast<const ch_u32x4*>(hb + ((hy * 2 + (hx & 1)) * ST_PL + (hx >> 1)) * 16);
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const ch_u32x4 af = *reinterpret_cast<const ch_u32x4*>(smem + (d * ST_NKS + ks) * 1024 + lane * 16);
                acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bf), acc[d], 0, 0, 0);
            }
        }

        // ---- epilogue through an LDS image of the tile (as conv_halo_kernel)
        __syncthreads();
        {
            const int q = 32 * wave + px, sw = (q >> 1) & 7;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (bf16_t)acc[d][4 * g + j];
                    *reinterpret_cast<bf16x4*>(hb + q * 128 + (((4 * d + g) ^ sw) * 16) + 8 * h) = o;
                }
        }
        __syncthreads();
        {
            const int c = tid & 7;
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int q = 64 * pass + (tid >> 3);
                const ch_u32x4 v = *reinterpret_cast<const ch_u32x4*>(hb + q * 128 + ((c ^ ((q >> 1) & 7)) * 16));
                bf16_t* op = p.Y + (((long)img * Ho + ty0 + (q >> 4)) * Wo + tx0 + (q & 15)) * p.ldy;
                *reinterpret_cast<ch_u32x4*>(op + 8 * c) = v;
            }
        }
        if (STATS) {
            const int ch = tid & 63, part = tid >> 6;
#pragma unroll 8
            for (int k = 0; k < 32; ++k) {
                const int q = 32 * part + k;
                const unsigned short u = *reinterpret_cast<const unsigned short*>(hb + q * 128 + (((ch >> 3) ^ ((q >> 1) & 7)) * 16) + 2 * (ch & 7));
                const float x = __uint_as_float((unsigned)u << 16);
                run1 += x;
                run2 = fmaf(x, x, run2);
            }
            if (t + 1 == t_end || (t + 1) / per_img != img) {
                const int lo = img * per_img, b0 = owner(lo), nb = owner(lo + per_img - 1) - b0 + 1;
                *reinterpret_cast<float2*>(red + (part * CH_C + ch) * 2) = make_float2(run1, run2);
                run1 = run2 = 0.f;
                __syncthreads();
                float* pp = p.part + ((long)img * per_img) * CH_C * 2;
                if (tid < 2 * CH_C) {
                    float sacc = 0.f;
#pragma unroll
                    for (int w = 0; w < 8; ++w) sacc += red[w * CH_C * 2 + tid];
                    __hip_atomic_store(reinterpret_cast<unsigned*>(pp + (long)((int)blockIdx.x - b0) * CH_C * 2 + tid), __float_as_uint(sacc),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __syncthreads();
                unsigned* tick = reinterpret_cast<unsigned*>(smem + ST_OFF_S + 8 * CH_C * 2 * 4);
                if (tid == 0) *tick = atomicAdd(p.cnt + img, 1u);
                __syncthreads();
                if (*tick == (unsigned)(nb - 1)) {
                    if (tid < 2 * CH_C) {
                        double sd = 0.0;
                        for (int k = 0; k < nb; k += 8) {
                            float v[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float* a = pp + (long)min(k + e, nb - 1) * CH_C * 2 + tid;
                                asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v[e]) : "v"(a) : "memory");
                            }
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                            for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(v[e]));
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                if (k + e < nb) sd += (double)v[e];
                        }
                        p.out_sums[(long)img * CH_C * 2 + tid] = sd;
                    }
                    if (tid == 0) __hip_atomic_store(p.cnt + img, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

}  // namespace

static bool ch_resident(int H, int W, int C) { return C == CH_C && (H % CH_T) == 0 && (W % CH_T) == 0; }
static bool ch_stream(int H, int W, int C) { return (C == 64 || C == 96 || C == 128) && (H % CS_TH) == 0 && (W % CS_TW) == 0; }

extern "C" int emip_conv3x3_halo_eligible(int B, int H, int W, int Cin, int Cout) {
    return B > 0 && Cin == Cout && H > 0 && W > 0 && (ch_resident(H, W, Cin) || ch_stream(H, W, Cin));
}
extern "C" int emip_conv3x3_halo_pack_bytes(int C) { return 9 * C * C * 2; }
/* bytes of the statistics workspace: [B] tickets (64-byte block, ZERO before the first use) then the partials, one per
 * (image, workgroup that holds tiles of it): never more than the image's tiles */
extern "C" long emip_conv3x3_halo_ws_bytes(int B, int H, int W, int C) {
    const long t16 = (long)((H + CH_T - 1) / CH_T) * ((W + CH_T - 1) / CH_T), t11 = (long)((H + CS_TH - 1) / CS_TH) * ((W + CS_TW - 1) / CS_TW);
    return (((long)B * 4 + 63) & ~63L) + (long)B * (t16 > t11 ? t16 : t11) * C * 2 * 4;
}

// Y = conv3x3(f(X)), stride 1, zero padding 1, no bias, C -> C channels (64: H, W multiples of 16, or 64 / 96 / 128: H a multiple
// of 11 and W of 22), bf16 channels-last (row strides ldx / ldy in elements); f = identity (in_sums NULL) or
// relu(InstanceNorm(X)) from in_sums f64 [B][C][2] = (sum, sum of squares) of X per image and channel (biased variance, eps);
// out_sums (may be NULL): the same sums of the stored Y, through ws (emip_conv3x3_halo_ws_bytes, ticket block zero).
// Wp: ops.conv3x3_halo_pack.
extern "C" int emip_conv3x3_halo(const void* X, long ldx, const void* Wp, void* Y, long ldy, int B, int H, int W, int Cin, int Cout,
                                 const double* in_sums, float in_eps, double* out_sums, void* ws, long ws_bytes, void* stream) {
    EMIP_REQUIRE(X && Wp && Y && emip_conv3x3_halo_eligible(B, H, W, Cin, Cout));
    const int C = Cin;
    EMIP_REQUIRE(ldx >= C && (ldx & 7) == 0 && ldy >= C && (ldy & 7) == 0 && aligned16(X) && aligned16(Wp) && aligned16(Y));
    const long xb = (((long)B * H * W - 1) * ldx + C) * 2;
    EMIP_REQUIRE(xb < (1L << 31));
    EMIP_REQUIRE(!out_sums || (ws && ws_bytes >= emip_conv3x3_halo_ws_bytes(B, H, W, C) && (reinterpret_cast<uintptr_t>(ws) & 63u) == 0));
    EMIP_REQUIRE(!in_sums || in_eps > 0.f);
    ChArgs a{};
    a.X = (const bf16_t*)X; a.Wp = (const bf16_t*)Wp; a.Y = (bf16_t*)Y; a.ldx = ldx; a.ldy = ldy; a.B = B; a.H = H; a.W = W;
    a.x_bytes = (unsigned)xb; a.in_sums = in_sums; a.in_eps = in_eps; a.out_sums = out_sums;
    if (out_sums) {
        a.cnt = (unsigned*)ws;
        a.part = (float*)((char*)ws + (((long)B * 4 + 63) & ~63L));
    }
    bool resident = ch_resident(H, W, C);
#ifdef EMIP_TUNING
    if (g_halo_mode && ch_stream(H, W, C)) resident = false;
#endif
    if (!resident) {
        a.tx = W / CS_TW; a.ty = H / CS_TH;
        const int nt = B * a.tx * a.ty;
        const bool nm = in_sums != nullptr, st = out_sums != nullptr;
#ifdef EMIP_TUNING
        if (C == 64 && g_halo_mode == 2) return launch_stream<64, false>(a, nm, st, nt, (hipStream_t)stream);
#endif
        if (C == 64) return launch_stream<64, true>(a, nm, st, nt, (hipStream_t)stream);
        if (C == 96) return launch_stream<96, true>(a, nm, st, nt, (hipStream_t)stream);
        return launch_stream<128, false>(a, nm, st, nt, (hipStream_t)stream);
    }
    a.tx = W / CH_T; a.ty = H / CH_T;
    const int ntiles = B * a.tx * a.ty;
    const int grid = ntiles < 256 ? ntiles : 256;
    void (*fn)(const ChArgs) = in_sums ? (out_sums ? conv_halo_kernel<true, true> : conv_halo_kernel<true, false>)
                                       : (out_sums ? conv_halo_kernel<false, true> : conv_halo_kernel<false, false>);
    static bool attr[4] = {false, false, false, false};
    const int which = (in_sums ? 2 : 0) + (out_sums ? 1 : 0);
    if (!attr[which]) {
        if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS) != hipSuccess) return EMIP_E_LAUNCH;
        attr[which] = true;
    }
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(512), CH_LDS, (hipStream_t)stream, a);
    return emip_launch_status();
}

#ifdef EMIP_TUNING
extern "C" int emip_debug_set_halo(int mode) { g_halo_mode = mode; return 0; }
#endif

// The encoder's stem (see conv_stem_kernel): Y [B, H / 2, W / 2, 64] = conv7x7(X [B, H, W, 8], stride 2, zero padding 3), bf16,
// H and W multiples of 32; Wp: 51 200 bytes in fragment order (ops.conv_stem_pack); out_sums / ws as in emip_conv3x3_halo
// (workspace of the OUTPUT map: emip_conv3x3_halo_ws_bytes(B, H / 2, W / 2, 64)).
extern "C" int emip_conv_stem_eligible(int B, int H, int W, int Cin, int Cout) {
    return B > 0 && Cin == 8 && Cout == CH_C && H >= 32 && W >= 32 && (H % 32) == 0 && (W % 32) == 0;
}
extern "C" int emip_conv_stem(const void* X, long ldx, const void* Wp, void* Y, long ldy, int B, int H, int W, int Cin, int Cout,
                              double* out_sums, void* ws, long ws_bytes, void* stream) {
    EMIP_REQUIRE(X && Wp && Y && emip_conv_stem_eligible(B, H, W, Cin, Cout));
    EMIP_REQUIRE(ldx >= 8 && (ldx & 7) == 0 && ldy >= CH_C && (ldy & 7) == 0 && aligned16(X) && aligned16(Wp) && aligned16(Y));
    const long xb = (((long)B * H * W - 1) * ldx + 8) * 2;
    EMIP_REQUIRE(xb < (1L << 31));
    EMIP_REQUIRE(!out_sums || (ws && ws_bytes >= emip_conv3x3_halo_ws_bytes(B, H / 2, W / 2, CH_C) && (reinterpret_cast<uintptr_t>(ws) & 63u) == 0));
    ChArgs a{};
    a.X = (const bf16_t*)X; a.Wp = (const bf16_t*)Wp; a.Y = (bf16_t*)Y; a.ldx = ldx; a.ldy = ldy; a.B = B; a.H = H; a.W = W;
    a.tx = W / 32; a.ty = H / 32; a.x_bytes = (unsigned)xb; a.out_sums = out_sums;
    if (out_sums) {
        a.cnt = (unsigned*)ws;
        a.part = (float*)((char*)ws + (((long)B * 4 + 63) & ~63L));
    }
    const int ntiles = B * a.tx * a.ty;
    void (*fn)(const ChArgs) = out_sums ? conv_stem_kernel<true> : conv_stem_kernel<false>;
    static bool attr[2] = {false, false};
    if (!attr[out_sums ? 1 : 0]) {
        if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS) != hipSuccess) return EMIP_E_LAUNCH;
        attr[out_sums ? 1 : 0] = true;
    }
    hipLaunchKernelGGL(fn, dim3((unsigned)(ntiles < 256 ? ntiles : 256)), dim3(512), ST_LDS, (hipStream_t)stream, a);
    return emip_launch_status();
}
