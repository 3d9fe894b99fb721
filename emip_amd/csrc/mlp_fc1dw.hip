// PVTv2 Mlp head in one launch (bf16 inference):  G = GELU( dwconv3x3( LN(x) W1^T + b1 ) + bd )
// (/root/reference/lib/pvt_v2.py:45-54 fc1 -> DWConv -> act, with norm2 of :165-169 folded into W1 / b1 and applied on the
// output side from the row statistics that travel with the residual stream, like emip_gemm_lne).
//
// The hidden tensor of the Mlp is the largest stream of the forward: fc1 writes it, the depthwise kernel reads and rewrites
// it, fc2 reads it -- 4 passes over 40 MB per stage-3 block at 16 pairs.  Here the fc1 output never reaches HBM:
//   * one workgroup (8 waves) owns ONE WHOLE IMAGE (H x W <= 512 tokens: the 22 x 22 and 11 x 11 stages) x a slab of 64 hidden
//     channels, so the depthwise 3 x 3 needs no halo and nothing is recomputed;
//   * the fc1 tile [512 x 64] is an LDS-DMA GEMM over K in steps of 32 channels (64-byte LDS rows, chunk ^ (row >> 2 & 3) on
//     the source side, two stages of 36 KB, one s_barrier per step), 4 x 2 waves of 128 x 32 outputs, MFMA 16x16x32 with the
//     weight fragment as the A operand (a lane owns 4 consecutive channels of one token);
//   * the tile is then written to LDS as bf16 [token][64 ch] over the drained ring (72 KB per workgroup: TWO workgroups per
//     CU, so one's pointwise phase runs under the other's MFMA loop), and every thread produces 2 x 8 outputs at a time
//     from a 3 x 4 window of 16-byte LDS reads, adds the bias, applies the bf16 GELU polynomial and stores whole 128-byte
//     row segments of G;
//   * workgroup ids are dealt round-robin over the XCDs: image = id % 8 + 8 (id / 8 / slabs), so the 20 slabs of an image
//     run on one XCD and share its 310-KB token panel in that L2.
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 mh_rsrc(const void* ptr, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
    return i32x4{(int)(unsigned)a, (int)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void mh_dma16(unsigned lds_dst, unsigned voff, i32x4 rs, unsigned soff) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_dst), "v"(voff), "s"(rs), "s"(soff)
        : "memory");
}

struct MhArgs {
    const bf16_t* X;        // [B * H * W, ldx] tokens (raw residual stream)
    const bf16_t* W1;       // [N, K] fc1 weights, LayerNorm gamma folded in
    const float* b1;        // [N] fc1 bias + W1 beta
    const float* colsum;    // [N] row sums of the packed W1 (output-side LayerNorm)
    const float* ln_stats;  // [B * H * W, 2] (sum, sum of squares) of the token rows
    const float* Wd;        // [9, N] depthwise taps
    const float* bd;        // [N]
    bf16_t* G;              // [B * H * W, ldg]
    long ldx, ldg;
    int B, H, Wd_, K, N;
    int bands, band_rows;   // row bands per image (1: the whole image) and output rows per band
    float eps;
    unsigned x_bytes, w_bytes;
    int xcd_map;
};

constexpr unsigned MH_OOB = 0x80000000u;
constexpr int MH_ROWS = 512, MH_BN = 64, MH_BK = 32;
constexpr int MH_A = MH_ROWS * 64;                  // bytes of the token slab of one stage
constexpr int MH_STAGE = MH_A + MH_BN * 64;         // + the weight slab
constexpr int MH_RING = 2 * MH_STAGE;               // 73 728 B; the bf16 fc1 tile (<= 512 x 128 B) reuses it
constexpr int MH_LDS = MH_RING + 10 * MH_BN * 4;    // + the slab's depthwise taps [9][64] and bias [64] (f32)

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4))) void mlp_fc1dw_kernel(const MhArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int slabs = p.N / MH_BN;
    int unit, slab;                                    // unit = (image, row band)
    if (p.xcd_map) {
        const int j = blockIdx.x >> 3;
        slab = j % slabs;
        unit = (blockIdx.x & 7) + 8 * (j / slabs);
    } else {
        slab = blockIdx.x % slabs;
        unit = blockIdx.x / slabs;
    }
    const int img = unit / p.bands, band = unit - img * p.bands;
    // the band's output rows y0 .. y1 - 1 need the fc1 result of rows ylo .. yhi - 1 (one halo row on each side inside the image:
    // recomputed by the neighbouring band too -- the price of keeping the fc1 output out of memory on maps of > 512 tokens)
    const int y0 = band * p.band_rows, y1 = min(p.H, y0 + p.band_rows);
    const int ylo = max(0, y0 - 1), yhi = min(p.H, y1 + 1);
    const int T = (yhi - ylo) * p.Wd_;                // tokens staged by this workgroup
    const long img0 = (long)img * p.H * p.Wd_;
    const long row0 = img0 + (long)ylo * p.Wd_;
    const int n0 = slab * MH_BN;

    const i32x4 rsX = mh_rsrc(p.X, p.x_bytes);
    const i32x4 rsW = mh_rsrc(p.W1, p.w_bytes);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)smem;

    // ---- staging plan: a 1-KB LDS-DMA piece = 16 rows of 64 B; lane l sits at row l >> 2, slot l & 3 and fetches source
    // chunk slot ^ (row >> 2 & 3).  Token pieces 4 j + ... : wave w moves pieces 8 j + w (j = 0..3); weight pieces: waves 0..3.
    unsigned xoff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 16 * (8 * j + wave) + (lane >> 2);
        const int c = (lane & 3) ^ ((r >> 2) & 3);
        xoff[j] = r < T ? (unsigned)(((row0 + r) * p.ldx + 8 * c) * 2) : MH_OOB;
    }
    unsigned woff;
    {
        const int r = 16 * (wave & 3) + (lane >> 2);
        const int c = (lane & 3) ^ ((r >> 2) & 3);
        woff = (unsigned)(((long)(n0 + r) * p.K + 8 * c) * 2);
    }
    auto issue = [&](int ks, int buf) {
        const unsigned base = lds0 + buf * MH_STAGE;
        const unsigned so = (unsigned)(ks * MH_BK * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) mh_dma16(base + (8 * j + wave) * 1024, xoff[j], rsX, so);
        if (wave < 4) mh_dma16(base + MH_A + wave * 1024, woff, rsW, so);
    };

    // the slab's depthwise taps and bias wait in LDS behind the ring: loading them in the pointwise phase, with two waves per
    // SIMD and nothing to hide a global round trip per kernel row, cost half of that phase
    float* wl = reinterpret_cast<float*>(smem + MH_RING);
    for (int i = tid; i < 10 * MH_BN; i += 512) {
        const int tap = i >> 6, c = i & 63;
        wl[i] = tap < 9 ? p.Wd[(long)tap * p.N + n0 + c] : p.bd[n0 + c];
    }

    f32x4 acc[8][2];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    auto frag = [&](const char* base, int row) {
        return *reinterpret_cast<const uint4*>(base + row * 64 + ((fq ^ ((row >> 2) & 3)) * 16));
    };
    const int nk = p.K / MH_BK;
    issue(0, 0);
    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                   // step ks has landed for every wave; everyone has left the other buffer
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 1 < nk) issue(ks + 1, buf ^ 1);
        const char* sa = smem + buf * MH_STAGE;
        const char* sw = sa + MH_A;
        uint4 wf[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) wf[b] = frag(sw, wn * 32 + 16 * b + fr);
#pragma unroll
        for (int a0 = 0; a0 < 8; a0 += 4) {
            uint4 af[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) af[a] = frag(sa, wm * 128 + 16 * (a0 + a) + fr);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a0 + a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[b]),
                                                                             __builtin_bit_cast(bf16x8, af[a]), acc[a0 + a][b], 0, 0, 0);
        }
    }
    __syncthreads();                                    // the ring is drained: its LDS becomes the fc1 tile

    // ---- epilogue 1: output-side LayerNorm + bias, bf16, into LDS as [token][64 ch] (128-B rows, chunk ^ (token & 7)) ----
    // lane holds acc[a][b][j] = C[token = wm 128 + 16 a + fr][channel = n0 + wn 32 + 16 b + 4 fq + j]
    {
        const float invK = 1.f / (float)p.K;
        float bv[2][4], cs[2][4];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 32 + 16 * b + 4 * fq + j;
                bv[b][j] = p.b1[n];
                cs[b][j] = p.colsum[n];
            }
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const int t = wm * 128 + 16 * a + fr;
            if (t < T) {
                const float2 s2 = *reinterpret_cast<const float2*>(p.ln_stats + 2 * (row0 + t));
                const float mu = s2.x * invK;
                const float rs = rsqrtf(fmaxf(s2.y * invK - mu * mu, 0.f) + p.eps);
                const float mrs = mu * rs;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    bf16x4 h;
#pragma unroll
                    for (int j = 0; j < 4; ++j) h[j] = (bf16_t)fmaf(acc[a][b][j], rs, fmaf(-mrs, cs[b][j], bv[b][j]));
                    const int col = wn * 32 + 16 * b + 4 * fq;            // channel inside the slab
                    *reinterpret_cast<bf16x4*>(smem + t * 128 + (((col >> 3) ^ (t & 7)) * 16) + ((col >> 2) & 1) * 8) = h;
                }
            }
        }
    }
    __syncthreads();

    // ---- epilogue 2: depthwise 3 x 3 + bias + GELU out of LDS; thread = channel group cg (8 channels) x pixel pairs --------
    {
        const int cg = tid & 7, c0 = n0 + 8 * cg;
        const int W2 = (p.Wd_ + 1) >> 1;                  // pixel pairs per image row
        const int npair = (y1 - y0) * W2;
        float bdv[8];
#pragma unroll
        for (int j = 0; j < 8; j += 4) {
            const float4 t4 = *reinterpret_cast<const float4*>(wl + 9 * MH_BN + 8 * cg + j);
            bdv[j] = t4.x; bdv[j + 1] = t4.y; bdv[j + 2] = t4.z; bdv[j + 3] = t4.w;
        }
        // a thread's (up to) NI pixel pairs share its channel group: the taps of one kernel row are loaded once per thread
        constexpr int NI = (MH_ROWS / 2 + 63) / 64;        // 4
        int py[NI], px[NI];
        float o[NI][2][8];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int pr = (tid >> 3) + 64 * i;
            py[i] = pr < npair ? y0 + pr / W2 : -4;       // image row; -4: no row of the window is inside the image
            px[i] = 2 * (pr - (pr / W2) * W2);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) o[i][q][j] = bdv[j];
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            float w[3][8];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int j = 0; j < 8; j += 4) {
                    const float4 t4 = *reinterpret_cast<const float4*>(wl + (3 * ky + kx) * MH_BN + 8 * cg + j);
                    w[kx][j] = t4.x; w[kx][j + 1] = t4.y; w[kx][j + 2] = t4.z; w[kx][j + 3] = t4.w;
                }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int yy = py[i] + ky - 1;
                if ((unsigned)yy >= (unsigned)p.H) continue;
#pragma unroll
                for (int dx = 0; dx < 4; ++dx) {
                    const int xx = px[i] - 1 + dx;
                    if ((unsigned)xx >= (unsigned)p.Wd_) continue;
                    const int t = (yy - ylo) * p.Wd_ + xx;  // token inside the staged rows
                    const uint4 raw = *reinterpret_cast<const uint4*>(smem + t * 128 + ((cg ^ (t & 7)) * 16));
                    const unsigned rw[4] = {raw.x, raw.y, raw.z, raw.w};
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[2 * k] = __uint_as_float(rw[k] << 16);
                        v[2 * k + 1] = __uint_as_float(rw[k] & 0xFFFF0000u);
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int kx = dx - q;                // output pixel q takes this column with tap kx
                        if (kx >= 0 && kx < 3) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) o[i][q][j] = fmaf(v[j], w[kx][j], o[i][q][j]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (py[i] < 0) continue;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int x = px[i] + q;
                if (x < p.Wd_) {
                    bf16x8 ov;
#pragma unroll
                    for (int j = 0; j < 8; ++j) ov[j] = (bf16_t)gelu_poly(o[i][q][j]);
                    *reinterpret_cast<bf16x8*>(p.G + (img0 + py[i] * p.Wd_ + x) * p.ldg + c0) = ov;
                }
            }
        }
    }
}

}  // namespace

// output rows per band: the whole image when it fits the 512-row tile, else the most rows whose halo-extended band does
static int mh_band_rows(int H, int Wd) {
    if (H * Wd <= MH_ROWS) return H;
    const int r = MH_ROWS / Wd - 2;
    return r >= 1 ? r : 0;
}

extern "C" int emip_mlp_fc1dw_eligible(int B, int H, int Wd, int K, int N) {
    // at least half of the 512-row tile must be tokens (the 11 x 11 stage pads 121 to 512 and loses: 46 vs 34 us)
    return B > 0 && H * Wd <= MH_ROWS && 2 * H * Wd > MH_ROWS && Wd >= 2 && K % MH_BK == 0 && K >= MH_BK && N % MH_BN == 0;
}

// maps of more than 512 tokens: row bands with a halo row on each side (the 44 x 44 stage: bands of 9 rows, 11 staged = 1.22x the
// fc1 work; the 88 x 88 stage: 3 rows, 5 staged = 1.67x).  Returns the output rows per band, 0 = not eligible.
extern "C" int emip_mlp_fc1dw_band_rows(int B, int H, int Wd, int K, int N) {
    if (!(B > 0 && Wd >= 2 && K % MH_BK == 0 && K >= MH_BK && N % MH_BN == 0) || H * Wd <= MH_ROWS) return 0;
    const int r = mh_band_rows(H, Wd);
    return (r >= 1 && 2 * (r + 2) * Wd > MH_ROWS) ? r : 0;
}

// G = GELU(dwconv3x3(LN(X) W1^T + b1) + bd) per image; see the head of this file.  bf16 activations / weights, f32 vectors.
extern "C" int emip_mlp_fc1dw(const void* X, long ldx, const void* W1, const float* b1, const float* colsum,
                              const float* ln_stats, float eps, const float* Wd, const float* bd, void* G, long ldg, int B,
                              int H, int Wdt, int K, int N, void* stream) {
    EMIP_REQUIRE(X && W1 && b1 && colsum && ln_stats && Wd && bd && G);
    EMIP_REQUIRE((emip_mlp_fc1dw_eligible(B, H, Wdt, K, N) || emip_mlp_fc1dw_band_rows(B, H, Wdt, K, N) > 0) && ldx >= K &&
                 (ldx & 7) == 0 && ldg >= N && (ldg & 7) == 0);
    EMIP_REQUIRE(aligned16(X) && aligned16(W1) && aligned16(G) && aligned16(Wd) && aligned16(bd));
    const long rows = (long)B * H * Wdt;
    EMIP_REQUIRE(rows * ldx * 2 < 0x7FFF0000L && (long)N * K * 2 < 0x7FFF0000L);
    MhArgs a{};
    a.X = (const bf16_t*)X; a.W1 = (const bf16_t*)W1; a.b1 = b1; a.colsum = colsum; a.ln_stats = ln_stats;
    a.Wd = Wd; a.bd = bd; a.G = (bf16_t*)G; a.ldx = ldx; a.ldg = ldg;
    a.B = B; a.H = H; a.Wd_ = Wdt; a.K = K; a.N = N; a.eps = eps;
    a.x_bytes = (unsigned)(((rows - 1) * ldx + K) * 2);
    a.w_bytes = (unsigned)((long)N * K * 2);
    a.band_rows = mh_band_rows(H, Wdt);
    a.bands = (H + a.band_rows - 1) / a.band_rows;
    a.xcd_map = ((B * a.bands) % 8) == 0;
    EMIP_REQUIRE((long)B * a.bands * (N / MH_BN) < 2147483647L);
    static bool attr = false;
    if (!attr) {
        attr = true;
        if (hipFuncSetAttribute((const void*)mlp_fc1dw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MH_LDS) != hipSuccess)
            return EMIP_E_LAUNCH;
    }
    hipLaunchKernelGGL(mlp_fc1dw_kernel, dim3((unsigned)(B * a.bands * (N / MH_BN))), dim3(512), MH_LDS, (hipStream_t)stream, a);
    return emip_launch_status();
}
