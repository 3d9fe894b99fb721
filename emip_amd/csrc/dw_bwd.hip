// Backward of the PVT Mlp's depthwise 3x3 + GELU in ONE pass over the hidden gradient (training step, SURVEY.md section 8
// row T; lib/pvt_v2.py:45-54,316-327):
//     dPre = dY * gelu'(Z)                      Z = pre-activation the forward stored (emip_dwconv3x3_dual)
//     dX[q]   = sum_t w[t] dPre[q - off(t)]     input gradient (what emip_dwconv3x3 with flipped taps computed)
//     dW[c][t] += sum_q X[q] dPre[q - off(t)]   weight gradient, written in the parameter's own [C][1][3][3] order
//     db[c]   += sum_q dPre[q]
// The three launches this replaces (emip_gelu_bwd, emip_dwconv3x3 on flipped taps, emip_dwconv3x3_wgrad) read or write the
// hidden-size tensor 7 times (dY, Z -> dPre; dPre -> dX; X, dPre -> dW); this one 4 times (dY, Z, X in, dX out).
//
// A workgroup owns a TS x TS pixel tile of one image and 64 channels.  Phase 1 evaluates dPre ONCE per element for the
// tile plus a one-pixel halo (zero outside the image) and parks it in LDS as bf16 -- the rounding the three-launch form
// applied when it stored dPre.  Phase 2: a thread owns 8 channels (16 B) and every 32nd pixel of the tile; per pixel the
// nine shifted dPre vectors come from LDS (conflict-free: a wave reads 8 whole 128-byte pixel rows per instruction) and
// feed both the input gradient (x w[t]) and the weight-gradient accumulators (x X[q]); X is read and dX written once, as
// whole 128-byte row segments.  Tail: 80 sums per thread are shuffle-reduced over the 8 pixel lanes of a wave, combined
// over the 4 waves through LDS and added with 640 f32 atomics per workgroup (same-address chains: one per tile of the
// image batch, 25 ns each -- DESIGN.md 5b).
#include "common.h"

namespace {

typedef unsigned int u32;

struct DwbArgs {
    const bf16_t* X;
    const bf16_t* Z;
    const bf16_t* DY;
    bf16_t* DX;
    const float* wt;   // [9][C]
    float* dW;         // [C][9]
    float* db;         // [C] or null
    long ldx, ldz, lddy, lddx;
    int H, W, C, TS, tiles_x, tiles_y, cgroups;
};


__device__ __forceinline__ u32 pack2(float lo, float hi) {
    const bf16_t a = (bf16_t)lo, b = (bf16_t)hi;
    return (u32)__builtin_bit_cast(unsigned short, a) | ((u32)__builtin_bit_cast(unsigned short, b) << 16);
}

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}

// Variants measured on MI355X at 64 x 22 x 22 x 1280 (this form: 109 us; the three launches: 152 us): 4 channels per thread
// with 512-thread workgroups at four waves per SIMD, taps in LDS (8-byte global accesses: 155 us); a rolled phase-2 loop
// that keeps four X vectors in flight with the channels in register-pair order (153 us: 256 registers, 12 spilled).
template <bool GELU>
__global__ __launch_bounds__(256, 2) void dw_bwd_kernel(DwbArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* tile = reinterpret_cast<uint4*>(smem);            // [(TS+2)][(TS+2)][8] vectors of 8 bf16
    const int tid = threadIdx.x, cv = tid & 7, pl = tid >> 3;
    const int cg = blockIdx.x % a.cgroups, tl = blockIdx.x / a.cgroups;
    const int ty0 = (tl / a.tiles_x) * a.TS, tx0 = (tl % a.tiles_x) * a.TS;
    const long b = blockIdx.y;
    const int c = cg * 64 + cv * 8;
    const bool cok = c < a.C;                                // C % 8 == 0 (entry point)
    const int cc = cok ? c : 0;
    const int TP = a.TS + 2;
    const long img = b * (long)a.H * a.W;

    // ---- phase 1: dPre of the tile + halo -> LDS
    const int nhalo = TP * TP;
    for (int p0 = 0; p0 < nhalo; p0 += 96) {
        uint4 zv[3], dv[3];
        bool ok[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int p = p0 + 32 * u + pl;
            const int py = p / TP, px = p - py * TP;
            const int y = ty0 - 1 + py, x = tx0 - 1 + px;
            ok[u] = cok && p < nhalo && y >= 0 && y < a.H && x >= 0 && x < a.W;
            const int yc = min(max(y, 0), a.H - 1), xc = min(max(x, 0), a.W - 1);
            const long row = img + (long)yc * a.W + xc;
            dv[u] = *reinterpret_cast<const uint4*>(a.DY + row * a.lddy + cc);
            if (GELU) zv[u] = *reinterpret_cast<const uint4*>(a.Z + row * a.ldz + cc);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int p = p0 + 32 * u + pl;
            if (p < nhalo) {
                uint4 o = make_uint4(0u, 0u, 0u, 0u);
                if (ok[u]) {
                    if (GELU) {
                        float z[8], d[8];
                        unpack8(zv[u], z);
                        unpack8(dv[u], d);
#pragma unroll
                        for (int j = 0; j < 8; ++j) d[j] *= gelu_grad_t<bf16_t>(z[j]);
                        o = make_uint4(pack2(d[0], d[1]), pack2(d[2], d[3]), pack2(d[4], d[5]), pack2(d[6], d[7]));
                    } else {
                        o = dv[u];
                    }
                }
                tile[p * 8 + cv] = o;
            }
        }
    }
    float w[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float4 w0 = *reinterpret_cast<const float4*>(a.wt + (long)t * a.C + cc);
        const float4 w1 = *reinterpret_cast<const float4*>(a.wt + (long)t * a.C + cc + 4);
        w[t][0] = w0.x; w[t][1] = w0.y; w[t][2] = w0.z; w[t][3] = w0.w;
        w[t][4] = w1.x; w[t][5] = w1.y; w[t][6] = w1.z; w[t][7] = w1.w;
    }
    float aw[9][8], ab[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        ab[j] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) aw[t][j] = 0.f;
    }
    __syncthreads();

    // ---- phase 2: input gradient + weight-gradient partial sums
    const int npix = a.TS * a.TS;
    for (int q = pl; q < npix; q += 32) {
        const int ly = q / a.TS, lx = q - ly * a.TS;
        const int y = ty0 + ly, x = tx0 + lx;
        const bool ok = cok && y < a.H && x < a.W;
        const int yc = min(y, a.H - 1), xc = min(x, a.W - 1);
        const long row = img + (long)yc * a.W + xc;
        const uint4 xv = *reinterpret_cast<const uint4*>(a.X + row * a.ldx + cc);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        float xf[8];
        unpack8(xv, xf);
        if (!ok) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xf[j] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int py = ly + 2 - t / 3, px = lx + 2 - t % 3;
            float v[8];
            unpack8(tile[(py * TP + px) * 8 + cv], v);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[j] = fmaf(w[t][j], v[j], acc[j]);
                aw[t][j] = fmaf(xf[j], v[j], aw[t][j]);
            }
            if (t == 4 && ok) {
#pragma unroll
                for (int j = 0; j < 8; ++j) ab[j] += v[j];
            }
        }
        if (ok)
            *reinterpret_cast<uint4*>(a.DX + row * a.lddx + c) =
                make_uint4(pack2(acc[0], acc[1]), pack2(acc[2], acc[3]), pack2(acc[4], acc[5]), pack2(acc[6], acc[7]));
    }

    // ---- tail: 80 sums per thread -> 8 pixel lanes of the wave -> 4 waves -> atomics
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float s = aw[t][j];
            s += __shfl_xor(s, 8);
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            aw[t][j] = s;
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float s = ab[j];
        s += __shfl_xor(s, 8);
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        ab[j] = s;
    }
    __syncthreads();                                         // the tile is dead: reuse its LDS
    float* red = reinterpret_cast<float*>(smem);             // [4 waves][10][64]
    const int wave = tid >> 6, lane = tid & 63;
    if (lane < 8) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) red[(wave * 10 + t) * 64 + lane * 8 + j] = aw[t][j];
#pragma unroll
        for (int j = 0; j < 8; ++j) red[(wave * 10 + 9) * 64 + lane * 8 + j] = ab[j];
    }
    __syncthreads();
    for (int i = tid; i < 640; i += 256) {
        const int t = i >> 6, ch = i & 63;
        const int cch = cg * 64 + ch;
        if (cch < a.C) {
            const float s = red[i] + red[640 + i] + red[1280 + i] + red[1920 + i];
            if (t < 9)
                atomicAdd(a.dW + (long)cch * 9 + t, s);
            else if (a.db)
                atomicAdd(a.db + cch, s);
        }
    }
}

}  // namespace

extern "C" int emip_dwconv3x3_bwd_fused(const void* X, long ldx, const void* Z, long ldz, const void* DY, long lddy,
                                        void* DX, long lddx, const float* wt, float* dW, float* db, int B, int H, int Wd,
                                        int C, int gelu, void* stream) {
    EMIP_REQUIRE(X && DY && DX && wt && dW && (Z || !gelu));
    EMIP_REQUIRE(B > 0 && B < 65536 && H > 0 && Wd > 0 && C >= 8 && C % 8 == 0);
    EMIP_REQUIRE(ldx >= C && lddy >= C && lddx >= C && ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0);
    EMIP_REQUIRE(!gelu || (ldz >= C && ldz % 8 == 0 && aligned16(Z)));
    EMIP_REQUIRE(aligned16(X) && aligned16(DY) && aligned16(DX) && aligned16(wt));
    // tile edge: 22 fits the PVT maps (88, 44, 22) without a remainder; 11 for the 11 x 11 stage; otherwise 16 with masking
    int TS = (H % 22 == 0 && Wd % 22 == 0) ? 22 : ((H % 11 == 0 && Wd % 11 == 0) ? 11 : 16);
    DwbArgs a;
    a.X = (const bf16_t*)X; a.Z = (const bf16_t*)Z; a.DY = (const bf16_t*)DY; a.DX = (bf16_t*)DX;
    a.wt = wt; a.dW = dW; a.db = db;
    a.ldx = ldx; a.ldz = ldz; a.lddy = lddy; a.lddx = lddx;
    a.H = H; a.W = Wd; a.C = C; a.TS = TS;
    a.tiles_x = (Wd + TS - 1) / TS; a.tiles_y = (H + TS - 1) / TS; a.cgroups = (C + 63) / 64;
    const long gx = (long)a.tiles_x * a.tiles_y * a.cgroups;
    EMIP_REQUIRE(gx < 2147483647L);
    size_t lds = (size_t)(TS + 2) * (TS + 2) * 128;
    if (lds < 4 * 640 * sizeof(float)) lds = 4 * 640 * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)dw_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 24 * 24 * 128) != hipSuccess ||
            hipFuncSetAttribute((const void*)dw_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 24 * 24 * 128) != hipSuccess)
            return EMIP_E_LAUNCH;
        attr_done = true;
    }
    if (gelu)
        hipLaunchKernelGGL(dw_bwd_kernel<true>, dim3((unsigned)gx, (unsigned)B), dim3(256), lds, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(dw_bwd_kernel<false>, dim3((unsigned)gx, (unsigned)B), dim3(256), lds, (hipStream_t)stream, a);
    return emip_launch_status();
}
