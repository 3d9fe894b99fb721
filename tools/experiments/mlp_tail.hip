// Fused tail of the PVTv2 Mlp for gfx950:   Y = R + bias2 + GELU(dwconv3x3(H) + bias_dw) * W2^T
// (/root/reference/lib/pvt_v2.py:45-54: fc1 -> DWConv -> GELU -> fc2; H is the fc1 output, channels-last).
//
// Why: the hidden tensor H [tokens][4C] is the largest activation of the network.  Unfused it crosses HBM four times per
// block (fc1 writes it, the depthwise kernel reads it and writes G, fc2 reads G) -- 42 % of all HBM bytes of a forward
// (PMC).  Here G never exists in HBM: a workgroup owns BM token rows and ALL N = C output channels, walks the hidden
// channels in K chunks of 128 bytes, computes the depthwise 3x3 + GELU for its BM x chunk slab on the VALU (the 3x3
// neighbours are ordinary coalesced 16-byte loads from H: rows above / below a tile are re-read by the neighbouring
// workgroup out of L2), drops the slab into LDS as the MFMA "B" operand while the matching W2 chunk arrives by LDS-DMA,
// and accumulates the full BM x N output tile in registers.  Full-N tiles mean every G element is computed exactly once.
//
// Layout notes: 4 waves split N (wave w owns columns [w*N/4, (w+1)*N/4)), every wave covers all BM rows; LDS rows are
// 128 B with the 16-byte chunk XOR-swizzled by ((row >> 1) & 7) like gemm.hip; operands are swapped (weights as MFMA "A")
// so a lane ends up with 4 consecutive output channels of one row.
#include "common.h"

namespace {

template <typename T>
struct MmaT;
template <>
struct MmaT<bf16_t> {
    static __device__ __forceinline__ f32x4 run(const uint4& a, const uint4& b, f32x4 acc) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                       acc, 0, 0, 0);
    }
};
template <>
struct MmaT<float> {
    static __device__ __forceinline__ f32x4 run(const uint4& a, const uint4& b, f32x4 acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
        return acc;
    }
};

struct TailArgs {
    const void* H;      // fc1 output [B][Hh][Ww][ldh]
    const float* Wt;    // depthwise weights [9][Ch]
    const float* bdw;   // depthwise bias [Ch]
    const void* W2;     // fc2 weight [N][ldw]
    const float* b2;    // fc2 bias [N] (may be null)
    const void* R;      // residual [M][ldr] (may be null; may alias Y)
    void* Y;            // [M][ldy]
    long ldh, ldw, ldr, ldy;
    int B, Hh, Ww, Ch, N;
    long M;
};

// TN: 16-column MFMA tiles per wave  (N = 64 * TN);  BM: token rows per workgroup (32 | 64)
template <typename T, int BM, int TN>
__global__ __launch_bounds__(256) void mlp_tail_kernel(const TailArgs p) {
    constexpr int VEC = 16 / sizeof(T);
    constexpr int BK = 128 / sizeof(T);
    constexpr int N = 64 * TN;
    constexpr int TM = BM / 16;
    constexpr int RA = BM / 32;                 // G rows staged per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                            // [BM][128 B]
    char* Ws = smem + BM * 128;                 // [N][128 B]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long m0 = (long)blockIdx.x * BM;
    const T* __restrict__ H = reinterpret_cast<const T*>(p.H);
    const T* __restrict__ W2 = reinterpret_cast<const T*>(p.W2);

    // ---- G staging: thread owns chunk column sc of rows srow + 32 i
    const int sc = tid & 7, srow = tid >> 3;
    const int swz_c = (sc ^ ((srow >> 1) & 7)) * 16;
    long pix[RA];            // centre pixel (row of H) of this thread's rows
    unsigned tapmask[RA];    // bit t set: tap t is inside the image
    bool rok[RA];
    const int hw = p.Hh * p.Ww;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const long m = m0 + srow + 32 * i;
        rok[i] = m < p.M;
        const long mc = rok[i] ? m : p.M - 1;
        const int rem = (int)(mc % hw);
        const int y = rem / p.Ww, x = rem - y * p.Ww;
        pix[i] = mc;
        unsigned mk = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
            if ((unsigned)iy < (unsigned)p.Hh && (unsigned)ix < (unsigned)p.Ww) mk |= 1u << t;
        }
        tapmask[i] = rok[i] ? mk : 0u;
    }

    // ---- W2 chunk by LDS-DMA: wave w stages rows [w*N/4, (w+1)*N/4); one instruction = 8 rows
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    const int lrow = lane >> 3, lslot = lane & 7;
    constexpr int WI = N / 32;                  // DMA instructions per wave per chunk
    const T* gw[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int r = wave * (N / 4) + 8 * i + lrow;
        const int c = lslot ^ ((r >> 1) & 7);
        gw[i] = W2 + (long)r * p.ldw + c * VEC;
    }

    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    const int fsw = (fr >> 1) & 7;
    const int nk = p.Ch / BK;
    // neighbour loads of one K chunk: unconditional 16-byte loads on clamped addresses, masked when consumed
    uint4 raw[RA][9];
    auto load_raw = [&](int k0) {
        const int c0 = k0 + sc * VEC;
#pragma unroll
        for (int i = 0; i < RA; ++i)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const bool ok = (tapmask[i] >> t) & 1u;
                const long off = ok ? (long)((t / 3 - 1) * p.Ww + (t % 3 - 1)) : 0L;     // clamped to the centre pixel
                raw[i][t] = *reinterpret_cast<const uint4*>(H + (pix[i] + off) * p.ldh + c0);
            }
    };
    load_raw(0);
    for (int kt = 0; kt < nk; ++kt) {
        const int k0 = kt * BK;
        // (1) W2 chunk -> LDS (asynchronous)
#pragma unroll
        for (int i = 0; i < WI; ++i)
            __builtin_amdgcn_global_load_lds((glb_void*)(gw[i] + k0), (lds_void*)(Ws + (wave * (N / 4) + 8 * i) * 128),
                                             16, 0, 0);
        // (2) depthwise 3x3 of this thread's RA x VEC slab from the prefetched neighbours
        const int c0 = k0 + sc * VEC;
        float g[RA][VEC];
        {
            float bv[VEC];
#pragma unroll
            for (int j = 0; j < VEC; j += 4) {
                const float4 t = *reinterpret_cast<const float4*>(p.bdw + c0 + j);
                bv[j] = t.x; bv[j + 1] = t.y; bv[j + 2] = t.z; bv[j + 3] = t.w;
            }
#pragma unroll
            for (int i = 0; i < RA; ++i)
#pragma unroll
                for (int j = 0; j < VEC; ++j) g[i][j] = bv[j];
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float w[VEC];
#pragma unroll
            for (int j = 0; j < VEC; j += 4) {
                const float4 tv = *reinterpret_cast<const float4*>(p.Wt + (long)t * p.Ch + c0 + j);
                w[j] = tv.x; w[j + 1] = tv.y; w[j + 2] = tv.z; w[j + 3] = tv.w;
            }
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const uint4 mv = mask4(raw[i][t], (tapmask[i] >> t) & 1u);
                const T* tv = reinterpret_cast<const T*>(&mv);
#pragma unroll
                for (int j = 0; j < VEC; ++j) g[i][j] = fmaf(to_f32<T>(tv[j]), w[j], g[i][j]);
            }
        }
        // the neighbour registers are free again: fetch the next chunk now, it lands during GELU + MFMA
        if (kt + 1 < nk) load_raw(k0 + BK);
        // GELU, round, drop into the LDS activation tile
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            uint4 ov;
            T* o = reinterpret_cast<T*>(&ov);
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] = from_f32<T>(gelu_t<T>(g[i][j]));
            *reinterpret_cast<uint4*>(As + (srow + 32 * i) * 128 + swz_c) = mask4(ov, rok[i]);
        }
        // loads retire in order: the W2 chunk has landed once at most the RA*9 prefetch loads are outstanding
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RA * 9) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // (3) MFMA: acc[n-tile][m-tile] += W2 chunk x G chunk
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
            const int coff = ((4 * gg + fq) ^ fsw) * 16;
            uint4 fa[TM], fw[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const uint4*>(As + (16 * i + fr) * 128 + coff);
#pragma unroll
            for (int a = 0; a < TN; ++a)
                fw[a] = *reinterpret_cast<const uint4*>(Ws + (wave * (N / 4) + 16 * a + fr) * 128 + coff);
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b) acc[a][b] = MmaT<T>::run(fw[a], fa[b], acc[a][b]);
        }
        __syncthreads();
    }

    // ---- epilogue: lane holds channels n0 + 4 fq .. +3 of row 16 b + fr
    T* Y = reinterpret_cast<T*>(p.Y);
    const T* R = reinterpret_cast<const T*>(p.R);
#pragma unroll
    for (int a = 0; a < TN; ++a) {
        const int n = wave * (N / 4) + 16 * a + 4 * fq;
        float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.b2) bias = *reinterpret_cast<const float4*>(p.b2 + n);
#pragma unroll
        for (int b = 0; b < TM; ++b) {
            const long m = m0 + 16 * b + fr;
            if (m >= p.M) continue;
            float v[4] = {acc[a][b][0] + bias.x, acc[a][b][1] + bias.y, acc[a][b][2] + bias.z, acc[a][b][3] + bias.w};
            if (R) {
                float r[4];
                Vec4<T>::load(R + m * p.ldr + n, r);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += r[j];
            }
            Vec4<T>::store(Y + m * p.ldy + n, v);
        }
    }
}

template <typename T, int BM, int TN>
int launch_tail(const TailArgs& a, hipStream_t s) {
    auto kfn = mlp_tail_kernel<T, BM, TN>;
    constexpr int lds = (BM + 64 * TN) * 128;
    static bool attr_set = false;
    if (lds > 65536 && !attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
            hipSuccess)
            return EMIP_E_LAUNCH;
        attr_set = true;
    }
    const unsigned grid = (unsigned)((a.M + BM - 1) / BM);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, s, a);
    return emip_launch_status();
}

template <typename T>
int dispatch_tail(const TailArgs& a, hipStream_t s) {
    // BM = 64 when that still gives every CU a couple of workgroups, else 32; N = 512 always takes 32 (accumulators)
    const bool big = (a.M + 63) / 64 >= 512;
    switch (a.N) {
        case 64: return big ? launch_tail<T, 64, 1>(a, s) : launch_tail<T, 32, 1>(a, s);
        case 128: return big ? launch_tail<T, 64, 2>(a, s) : launch_tail<T, 32, 2>(a, s);
        case 320: return big ? launch_tail<T, 64, 5>(a, s) : launch_tail<T, 32, 5>(a, s);
        case 512: return launch_tail<T, 32, 8>(a, s);
    }
    return EMIP_E_INVALID;
}

}  // namespace

// Y[m][n] = R[m][n] + b2[n] + sum_k GELU(dwconv3x3(H)[m][k] + bdw[k]) * W2[n][k];  m = (b, y, x) over [B][Hh][Ww].
// N in {64, 128, 320, 512} (the PVTv2-b5 embedding widths); Ch a multiple of the 128-byte K chunk.
extern "C" int emip_mlp_tail(const void* H, long ldh, const float* Wt, const float* bdw, const void* W2, long ldw,
                             const float* b2, const void* R, long ldr, void* Y, long ldy, int B, int Hh, int Ww, int Ch,
                             int N, int dtype, void* stream) {
    EMIP_REQUIRE(H && Wt && bdw && W2 && Y && B > 0 && Hh > 0 && Ww > 0 && Ch > 0);
    EMIP_REQUIRE(dtype == EMIP_F32 || dtype == EMIP_BF16);
    EMIP_REQUIRE(N == 64 || N == 128 || N == 320 || N == 512);
    const int bk = dtype == EMIP_F32 ? 32 : 64, vec = dtype == EMIP_F32 ? 4 : 8;
    EMIP_REQUIRE(Ch % bk == 0 && ldh % vec == 0 && ldw % vec == 0 && ldh >= Ch && ldw >= Ch && ldy >= N && (ldy & 3) == 0);
    EMIP_REQUIRE(aligned16(H) && aligned16(W2) && aligned16(Wt) && aligned16(bdw) && (((uintptr_t)Y) & 15) == 0);
    if (R) EMIP_REQUIRE(ldr >= N && (ldr & 3) == 0 && (((uintptr_t)R) & 15) == 0);
    if (b2) EMIP_REQUIRE(aligned16(b2));
    TailArgs a{};
    a.H = H; a.Wt = Wt; a.bdw = bdw; a.W2 = W2; a.b2 = b2; a.R = R; a.Y = Y;
    a.ldh = ldh; a.ldw = ldw; a.ldr = ldr; a.ldy = ldy;
    a.B = B; a.Hh = Hh; a.Ww = Ww; a.Ch = Ch; a.N = N;
    a.M = (long)B * Hh * Ww;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return dtype == EMIP_F32 ? dispatch_tail<float>(a, s) : dispatch_tail<bf16_t>(a, s);
}
