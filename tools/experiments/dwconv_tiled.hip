// LDS-tiled depthwise 3x3 (+bias, +GELU / ReLU) for bf16 channels-last tensors -- the Mlp's DWConv of PVTv2
// (/root/reference/lib/pvt_v2.py:316-327: dwconv(x.transpose(1, 2).view(B, C, H, W)), then GELU at :50-51) and the
// MDTA q / kv depthwise convs (model/EMIP_short/motion/PromptInteract.py:413-415).
//
// HBM-bound: 2 B in + 2 B out per element.  A workgroup owns a band of TY output rows x the full width x a 64-channel slab
// (128 B per pixel = one cache line) of one image:
//   * the (TY+2) x (W+2) input tile goes to LDS once, 16 B per lane, whole lines, zero padding written as zeros -- every
//     input element is read from HBM / Infinity Cache (TY+2)/TY times (exactly once when a band is a whole 22x22 image)
//     instead of the ~4.5 L2 reads per element of the register-window kernel;
//   * a thread keeps ITS 8 channels' 9 taps and bias in registers for the whole band (thread t always works on chunk t & 7),
//     reads the 9 neighbours with ds_read_b128 (8 pixels x 8 chunks per wave: conflict-free) and stores 16 B.
#include "common.h"
#include <stdlib.h>

namespace {

template <int CS>      // channels per slab: 64 (128-B pixels) or 32
__global__ __launch_bounds__(256) void dwconv_tiled_kernel(const bf16_t* __restrict__ X, long ldx, bf16_t* __restrict__ Y,
                                                           long ldy, const float* __restrict__ Wt,
                                                           const float* __restrict__ bias, int H, int Wd, int C, int TY,
                                                           int act) {
    constexpr int CPP = CS / 8;            // 16-B chunks per pixel
    constexpr int PB = CS * 2;             // bytes per pixel in the tile
    extern __shared__ __attribute__((aligned(16))) char tile[];
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * CS, y0 = blockIdx.y * TY;
    const long img = blockIdx.z;
    const int rows = min(TY, H - y0);
    const int TW = Wd + 2, P = (rows + 2) * TW;
    const bf16_t* Xi = X + img * H * Wd * ldx + c0;
    bf16_t* Yi = Y + img * H * Wd * ldy + c0;

    // ---- tile -> LDS (loads first, all in flight; then the stores)
    for (int base = 0; base < P * CPP; base += 256 * 4) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * 256 + tid;
            const int px = idx / CPP, ch = idx - px * CPP;
            const int ty = px / TW, tx = px - ty * TW;
            const int iy = y0 + ty - 1, ix = tx - 1;
            const bool ok = idx < P * CPP && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)Wd;
            const long off = ((long)min(max(iy, 0), H - 1) * Wd + min(max(ix, 0), Wd - 1)) * ldx + ch * 8;
            v[u] = mask4(*reinterpret_cast<const uint4*>(Xi + off), ok);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * 256 + tid;
            if (idx < P * CPP) *reinterpret_cast<uint4*>(tile + (long)idx * 16) = v[u];
        }
    }
    // ---- this thread's channels: taps and bias
    const int ch = tid % CPP;
    float w[9][8], bv[8];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float4 a = *reinterpret_cast<const float4*>(Wt + (long)t * C + c0 + ch * 8);
        const float4 b = *reinterpret_cast<const float4*>(Wt + (long)t * C + c0 + ch * 8 + 4);
        w[t][0] = a.x; w[t][1] = a.y; w[t][2] = a.z; w[t][3] = a.w;
        w[t][4] = b.x; w[t][5] = b.y; w[t][6] = b.z; w[t][7] = b.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[j] = bias ? bias[c0 + ch * 8 + j] : 0.f;
    __syncthreads();

    const int npix = rows * Wd;
    for (int o = tid / CPP; o < npix; o += 256 / CPP) {
        const int oy = o / Wd, ox = o - oy * Wd;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = bv[j];
        const char* tp = tile + ((long)oy * TW + ox) * PB + ch * 16;       // tile pixel (oy, ox) = input pixel (y0+oy-1, ox-1)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const uint4 r = *reinterpret_cast<const uint4*>(tp + ((long)ky * TW + kx) * PB);
                const unsigned u[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    acc[2 * k] = fmaf(__uint_as_float(u[k] << 16), w[ky * 3 + kx][2 * k], acc[2 * k]);
                    acc[2 * k + 1] = fmaf(__uint_as_float(u[k] & 0xFFFF0000u), w[ky * 3 + kx][2 * k + 1], acc[2 * k + 1]);
                }
            }
        if (act == EMIP_ACT_GELU) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = gelu_t<bf16_t>(acc[j]);
        } else if (act == EMIP_ACT_RELU) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaxf(acc[j], 0.f);
        }
        uint4 ov;
        bf16_t* op = reinterpret_cast<bf16_t*>(&ov);
#pragma unroll
        for (int j = 0; j < 8; ++j) op[j] = (bf16_t)acc[j];
        *reinterpret_cast<uint4*>(Yi + ((long)(y0 + oy) * Wd + ox) * ldy + ch * 8) = ov;
    }
}

}  // namespace

namespace emip_internal {
// > 0: not eligible (the register-window kernel of pointwise.hip runs), EMIP_OK: launched
int dwconv_tiled_try(const void* X, long ldx, void* Y, long ldy, const float* Wt, const float* bias, int B, int H, int Wd,
                     int C, int act, hipStream_t s) {
    static int enabled = -1;
    if (enabled < 0) {
        const char* e = getenv("EMIP_DW_TILED");
        enabled = (e && e[0] == '0') ? 0 : 1;
    }
    if (!enabled || (C % 64) != 0 || (ldx % 8) != 0 || (ldy % 8) != 0 || !aligned16(X) || !aligned16(Y) || B >= 65536 ||
        (reinterpret_cast<uintptr_t>(Wt) & 15) != 0)
        return 1;
    // 64-channel slabs unless the row is so wide that a band of >= 6 rows would not fit ~72 KB of LDS
    const int budget = 72 * 1024;
    int cs = 64;
    int ty = budget / ((Wd + 2) * 128) - 2;
    if (ty < 6) {
        cs = 32;
        ty = budget / ((Wd + 2) * 64) - 2;
    }
    if (ty < 2) return 1;
    if (ty > H) ty = H;
    // equal bands (the last one is not a sliver)
    const int bands = (H + ty - 1) / ty;
    ty = (H + bands - 1) / bands;
    const size_t lds = (size_t)(ty + 2) * (Wd + 2) * cs * 2;
    auto k64 = dwconv_tiled_kernel<64>;
    auto k32 = dwconv_tiled_kernel<32>;
    static size_t attr64 = 0, attr32 = 0;
    size_t& done = cs == 64 ? attr64 : attr32;
    if (done < lds) {
        if (hipFuncSetAttribute((const void*)(cs == 64 ? k64 : k32), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
            return EMIP_E_LAUNCH;
        done = lds;
    }
    dim3 grid(C / cs, bands, B);
    if (cs == 64)
        hipLaunchKernelGGL(k64, grid, dim3(256), lds, s, (const bf16_t*)X, ldx, (bf16_t*)Y, ldy, Wt, bias, H, Wd, C, ty, act);
    else
        hipLaunchKernelGGL(k32, grid, dim3(256), lds, s, (const bf16_t*)X, ldx, (bf16_t*)Y, ldy, Wt, bias, H, Wd, C, ty, act);
    return emip_launch_status();
}
}  // namespace emip_internal
