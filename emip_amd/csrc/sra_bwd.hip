// Backward of the PVTv2 spatial-reduction attention for gfx950 (bf16, head_dim 64, at most 128 keys), in one launch:
//   P = softmax(q k^T * scale),  O = P v      (/root/reference/lib/pvt_v2.py:113-125, under loss.backward() of train.py:52-58)
//   dV = P^T dO,  dP = dO V^T,  dS = P o (dP - rowsum(dO o O)) * scale,  dQ = dS K,  dK = dS^T Q
//
// The first rounds ran this as an unfused chain per block (three batched GEMMs, two softmax passes, two transposed GEMMs,
// two copies), materialising the [B, heads, Lq, 128] score matrix four times: ~15 ms of the 122-ms training step.  Here the
// key side is resident and the queries stream, like in the forward kernel (sra.hip), but with the KEYS split over the four
// waves of a workgroup (wave w owns keys 32 w .. 32 w + 31) so that each wave's dK / dV accumulators are 32 x 64:
//   * K_w, V_w (MFMA A operand, rows = keys) and K_w^T (rows = head channels) live in registers for the whole kernel;
//   * per 32-query block, all four waves read the same Q / dO / O rows from LDS images (row-read and transposed-read copies);
//     P is recomputed from the saved log-sum-exp L (no row reduction across waves), D = rowsum(dO o O) per lane pair;
//   * S^T, dP^T come out with the query on the lane, so P and dS are the B operand of dQ^T += K_w^T dS^T as they stand;
//     for dV += P^T dO and dK += dS^T Q they take A-operand shape through a wave-private 32 x 32 LDS tile;
//   * the four waves' dQ partials (one per key block) meet in LDS and leave as whole 128-byte rows;
//   * dK / dV are ADDED (f32 atomics, 128-byte row segments) into a pre-cleared [B, Lk, 2C] buffer: the query range of an
//     (image, head) pair may be split over workgroups.
#include "common.h"

namespace {

struct SraBwdArgs {
    const bf16_t* Q;     // [B, Lq, C]
    const bf16_t* KV;    // [B, Lk, 2C]
    const bf16_t* O;     // [B, Lq, C]   forward output
    const bf16_t* dO;    // [B, Lq, C]
    const float* L;      // [B, heads, Lq]  log2-sum-exp of the scaled scores
    bf16_t* dQ;          // [B, Lq, C]
    float* dKV;          // [B, Lk, 2C]  accumulated into
    bf16_t* dKVb;        // or: [B, Lk, 2C] bf16, STORED (one workgroup per (image, head): splits == 1)
    int Lq, Lk, C, heads, splits;
    float scale;
};

__device__ __forceinline__ int bv_off(int row, int c) { return row * 128 + ((c ^ (((row >> 1) & 1) << 2)) * 16); }   // transposed reads
__device__ __forceinline__ int br_off(int row, int c) { return row * 128 + ((c ^ (row & 7)) * 16); }                   // row reads

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// MFMA 32x32x16 operand with rows (A) / columns (B) = the COLUMNS 32 itile .. of a [k rows][128 B] image (bv_off layout) and
// k = its rows kbase .. kbase + 15: the transposed-read pattern of sra.hip's V^T fragments
__device__ __forceinline__ bf16x8 tr_frag(const char* img, int kbase, int itile, int lane) {
    const int i16 = lane & 15, g16 = (lane >> 4) & 1, h = lane >> 5;
    const int base0 = kbase + 4 * h + (i16 >> 2);
    const int col = 32 * itile + 16 * g16 + 4 * (i16 & 3);
    const int c = col >> 3, half = (col >> 2) & 1;
    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + bv_off(base0, c) + 8 * half));
    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + bv_off(base0 + 8, c) + 8 * half));
    const bf16x4 b0 = __builtin_bit_cast(bf16x4, v0), b1 = __builtin_bit_cast(bf16x4, v1);
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[j] = b0[j];
        f[4 + j] = b1[j];
    }
    return f;
}

constexpr int PSTRIDE = 272;                                  // bytes per query row of a wave's f32 dQ partial (256 + 16)
constexpr int REG_A = 4 * 32 * PSTRIDE;                       // 34 816 B: K row image + K transposed image at start, dQ partials after
constexpr int IMG = 32 * 128;                                 // one 32-row image
constexpr int SRA_BWD_LDS = REG_A + 5 * IMG + 4 * IMG;        // + Q row / Q tr / dO row / dO tr / O row + one tile per wave

__global__ __launch_bounds__(256) void sra_bwd_kernel(const SraBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* krow = smem;                     // [128][128 B] row-read layout (prologue only)
    char* ktr = smem + 128 * 128;          // [128][128 B] transposed-read layout (prologue only)
    char* part = smem;                     // [4 waves][32 q][PSTRIDE] f32 (after the prologue)
    char* qrow = smem + REG_A;
    char* qtr = qrow + IMG;
    char* dorow = qtr + IMG;
    char* dotr = dorow + IMG;
    char* orow = dotr + IMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char* tile = orow + IMG + wave * IMG;  // wave-private [32 q][128 B], transposed-read layout
    const int lq = lane & 31, h = lane >> 5;
    const int split = blockIdx.x, head = blockIdx.y;
    const long batch = blockIdx.z;
    const long qbase = batch * p.Lq * p.C + head * 64;
    const bf16_t* __restrict__ Qp = p.Q + qbase;
    const bf16_t* __restrict__ Op = p.O + qbase;
    const bf16_t* __restrict__ dOp = p.dO + qbase;
    bf16_t* __restrict__ dQp = p.dQ + qbase;
    const bf16_t* __restrict__ Kp = p.KV + batch * p.Lk * 2 * p.C + head * 64;
    const bf16_t* __restrict__ Vp = Kp + p.C;
    const float* __restrict__ Lp = p.L + (batch * p.heads + head) * p.Lq;
    const long ldk = 2 * p.C;

    const int nblk = (p.Lq + 31) >> 5;
    const int b_lo = (int)((long)split * nblk / p.splits), b_hi = (int)((long)(split + 1) * nblk / p.splits);
    const int srow = tid >> 3, sch = tid & 7;             // image staging: thread moves chunk sch of row srow
    uint4 nq, ndo, no_;
    auto load_blk = [&](int blk) {
        const int q = blk * 32 + srow;
        const long qr = min(q, p.Lq - 1);
        const bool ok = q < p.Lq;
        nq = mask4(*reinterpret_cast<const uint4*>(Qp + qr * p.C + sch * 8), ok);
        ndo = mask4(*reinterpret_cast<const uint4*>(dOp + qr * p.C + sch * 8), ok);
        no_ = mask4(*reinterpret_cast<const uint4*>(Op + qr * p.C + sch * 8), ok);
    };
    if (b_lo < b_hi) load_blk(b_lo);

    // ---- prologue: K -> both images -> K_w, K_w^T fragments; then V over the row image -> V_w fragments ---------------------
    uint4 kf[4], vf[4];
    bf16x8 ktrf[2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = tid + 256 * i, r = id >> 3, c = id & 7;
        const long kr = min(r, p.Lk - 1);
        const uint4 v = mask4(*reinterpret_cast<const uint4*>(Kp + kr * ldk + c * 8), r < p.Lk);
        *reinterpret_cast<uint4*>(krow + br_off(r, c)) = v;
        *reinterpret_cast<uint4*>(ktr + bv_off(r, c)) = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) kf[i] = *reinterpret_cast<const uint4*>(krow + br_off(32 * wave + lq, 2 * i + h));
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) ktrf[dt][sp] = tr_frag(ktr, 32 * wave + 16 * sp, dt, lane);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = tid + 256 * i, r = id >> 3, c = id & 7;
        const long kr = min(r, p.Lk - 1);
        *reinterpret_cast<uint4*>(krow + br_off(r, c)) = mask4(*reinterpret_cast<const uint4*>(Vp + kr * ldk + c * 8), r < p.Lk);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) vf[i] = *reinterpret_cast<const uint4*>(krow + br_off(32 * wave + lq, 2 * i + h));

    f32x16 dk[2], dv[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[d][r] = dv[d][r] = 0.f;
    const float sc2 = p.scale * 1.4426950408889634f;

    for (int blk = b_lo; blk < b_hi; ++blk) {
        __syncthreads();                    // everyone has left the images and the dQ partials of the previous block
        *reinterpret_cast<uint4*>(qrow + br_off(srow, sch)) = nq;
        *reinterpret_cast<uint4*>(qtr + bv_off(srow, sch)) = nq;
        *reinterpret_cast<uint4*>(dorow + br_off(srow, sch)) = ndo;
        *reinterpret_cast<uint4*>(dotr + bv_off(srow, sch)) = ndo;
        *reinterpret_cast<uint4*>(orow + br_off(srow, sch)) = no_;
        if (blk + 1 < b_hi) load_blk(blk + 1);
        const int q = blk * 32 + lq;
        const float lse = q < p.Lq ? Lp[q] : INFINITY;      // queries past the end: P = exp2(-inf) = 0
        __syncthreads();

        // ---- fragments of this lane's query: Q and dO as B operands (k = head channels 16 i + 8 h ..), D = rowsum(dO o O)
        uint4 qf[4], dof[4];
        float dsum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            qf[i] = *reinterpret_cast<const uint4*>(qrow + br_off(lq, 2 * i + h));
            dof[i] = *reinterpret_cast<const uint4*>(dorow + br_off(lq, 2 * i + h));
            const uint4 of = *reinterpret_cast<const uint4*>(orow + br_off(lq, 2 * i + h));
            const bf16_t* a = reinterpret_cast<const bf16_t*>(&dof[i]);
            const bf16_t* b = reinterpret_cast<const bf16_t*>(&of);
#pragma unroll
            for (int j = 0; j < 8; ++j) dsum = fmaf((float)a[j], (float)b[j], dsum);
        }
        dsum += __shfl_xor(dsum, 32);

        // ---- S^T_w = K_w Q^T, dP^T_w = V_w dO^T   (register r = key 32 w + 8 (r >> 2) + 4 h + (r & 3), lane = query lq)
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[i]), __builtin_bit_cast(bf16x8, qf[i]), s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf[i]), __builtin_bit_cast(bf16x8, dof[i]), dp, 0, 0, 0);
        }
        // ---- P = exp2(S sc2 - L), dS = P (dP - D) scale
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = 32 * wave + 8 * (r >> 2) + 4 * h + (r & 3);
            const float pv = key < p.Lk ? __builtin_amdgcn_exp2f(fmaf(s[r], sc2, -lse)) : 0.f;
            s[r] = pv;
            dp[r] = pv * (dp[r] - dsum) * p.scale;
        }
        bf16x8 pf[2], dsf[2];
#pragma unroll
        for (int sp = 0; sp < 2; ++sp)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                pf[sp][j] = (bf16_t)s[8 * sp + j];
                dsf[sp][j] = (bf16_t)dp[8 * sp + j];
            }

        // ---- dQ^T (this wave's key block) = K_w^T dS^T_w: lane = query, register = head channel 32 dt + 8 (r >> 2) + 4 h + (r & 3)
        f32x16 dq[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktrf[dt][sp], dsf[sp], dq[dt], 0, 0, 0);
        }
        {
            char* pw = part + wave * (32 * PSTRIDE) + lq * PSTRIDE;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(pw + (32 * dt + 8 * g + 4 * h) * 4) =
                        make_float4(dq[dt][4 * g], dq[dt][4 * g + 1], dq[dt][4 * g + 2], dq[dt][4 * g + 3]);
        }

        // ---- dV_w += P^T_w dO,  dK_w += dS^T_w Q: P / dS take A-operand shape (rows = keys, k = queries) through the wave's tile
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x4 t;
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = pf[g >> 1][4 * (g & 1) + j];
            *reinterpret_cast<uint2*>(tile + bv_off(lq, g) + 8 * h) = __builtin_bit_cast(uint2, t);
        }
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            const bf16x8 a = tr_frag(tile, 16 * sp, 0, lane);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, tr_frag(dotr, 16 * sp, dt, lane), dv[dt], 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x4 t;
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = dsf[g >> 1][4 * (g & 1) + j];
            *reinterpret_cast<uint2*>(tile + bv_off(lq, g) + 8 * h) = __builtin_bit_cast(uint2, t);
        }
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            const bf16x8 a = tr_frag(tile, 16 * sp, 0, lane);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, tr_frag(qtr, 16 * sp, dt, lane), dk[dt], 0, 0, 0);
        }

        // ---- dQ rows: sum of the four key blocks' partials, bf16, whole 128-byte rows
        __syncthreads();
        {
            const int qq = tid >> 3, c8 = (tid & 7) * 8;          // query row of the block, 8 head channels
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float* pr = reinterpret_cast<const float*>(part + w * (32 * PSTRIDE) + qq * PSTRIDE) + c8;
                const float4 a = *reinterpret_cast<const float4*>(pr), b = *reinterpret_cast<const float4*>(pr + 4);
                acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
                acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
            }
            const int qg = blk * 32 + qq;
            if (qg < p.Lq) {
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16_t)acc[j];
                *reinterpret_cast<bf16x8*>(dQp + (long)qg * p.C + c8) = o;
            }
        }
    }

    // ---- dK_w, dV_w: lane = head channel 32 dt + lq, register = key 32 w + 8 (r >> 2) + 4 h + (r & 3) ----------------------
    if (p.dKVb) {                                            // this workgroup saw every query of its (image, head): final values
        bf16_t* __restrict__ dkb = p.dKVb + batch * p.Lk * 2 * p.C + head * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = 32 * wave + 8 * (r >> 2) + 4 * h + (r & 3);
                if (key < p.Lk) {
                    bf16_t* row = dkb + (long)key * 2 * p.C + 32 * dt + lq;
                    row[0] = (bf16_t)dk[dt][r];
                    row[p.C] = (bf16_t)dv[dt][r];
                }
            }
        return;
    }
    float* __restrict__ dkp = p.dKV + batch * p.Lk * 2 * p.C + head * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = 32 * wave + 8 * (r >> 2) + 4 * h + (r & 3);
            if (key < p.Lk) {
                float* row = dkp + (long)key * 2 * p.C + 32 * dt + lq;
                atomicAdd(row, dk[dt][r]);
                atomicAdd(row + p.C, dv[dt][r]);
            }
        }
}

}  // namespace

static int sra_bwd_launch(const void* Q, const void* KV, const void* O, const void* dO, const float* L, void* dQ, float* dKV,
                          void* dKVb, int batch, int heads, int Lq, int Lk, int C, float scale, void* stream);

extern "C" int emip_sra_attention_bwd(const void* Q, const void* KV, const void* O, const void* dO, const float* L, void* dQ,
                                      float* dKV, int batch, int heads, int Lq, int Lk, int C, float scale, void* stream) {
    EMIP_REQUIRE(dKV);
    return sra_bwd_launch(Q, KV, O, dO, L, dQ, dKV, nullptr, batch, heads, Lq, Lk, C, scale, stream);
}

// the same with dK | dV STORED as bf16 [B, Lk, 2C] (no pre-cleared f32 accumulator, no conversion pass): one workgroup per
// (image, head) walks all queries -- the form for batch * heads >= 256 (the 22 x 22 and 11 x 11 stages at batch 32)
extern "C" int emip_sra_attention_bwd_bf16(const void* Q, const void* KV, const void* O, const void* dO, const float* L, void* dQ,
                                           void* dKV, int batch, int heads, int Lq, int Lk, int C, float scale, void* stream) {
    EMIP_REQUIRE(dKV && aligned16(dKV));
    return sra_bwd_launch(Q, KV, O, dO, L, dQ, nullptr, dKV, batch, heads, Lq, Lk, C, scale, stream);
}

static int sra_bwd_launch(const void* Q, const void* KV, const void* O, const void* dO, const float* L, void* dQ, float* dKV,
                          void* dKVb, int batch, int heads, int Lq, int Lk, int C, float scale, void* stream) {
    EMIP_REQUIRE(Q && KV && O && dO && L && dQ && batch > 0 && heads > 0 && Lq > 0 && Lk > 0 && Lk <= 128 && C == heads * 64);
    EMIP_REQUIRE(batch < 65536 && heads < 65536 && aligned16(Q) && aligned16(KV) && aligned16(O) && aligned16(dO) && aligned16(dQ));
    SraBwdArgs a{};
    a.Q = (const bf16_t*)Q; a.KV = (const bf16_t*)KV; a.O = (const bf16_t*)O; a.dO = (const bf16_t*)dO; a.L = L;
    a.dQ = (bf16_t*)dQ; a.dKV = dKV; a.dKVb = (bf16_t*)dKVb; a.Lq = Lq; a.Lk = Lk; a.C = C; a.heads = heads; a.scale = scale;
    // two workgroups per CU (72 KB each); the query blocks of an (image, head) pair are split when there are fewer pairs than
    // that, but every workgroup keeps >= 8 blocks: its prologue (K, V, K^T into registers) and its 64 KB of atomics are fixed
    const int nblk = (Lq + 31) / 32, pairs = batch * heads;
    int splits = (512 + pairs - 1) / pairs;
    if (splits > nblk / 8) splits = nblk / 8;
    if (splits < 1 || dKVb) splits = 1;
    a.splits = splits;
    static bool attr = false;
    if (!attr) {
        attr = true;
        if (hipFuncSetAttribute((const void*)sra_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SRA_BWD_LDS) != hipSuccess)
            return EMIP_E_LAUNCH;
    }
    hipLaunchKernelGGL(sra_bwd_kernel, dim3(splits, heads, batch), dim3(256), SRA_BWD_LDS, (hipStream_t)stream, a);
    return emip_launch_status();
}
