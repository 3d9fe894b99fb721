// Fused softmax attention for gfx950:  O = softmax(Q K^T * scale + mask) V
//
// One kernel template serves every attention-shaped op of the EMIP path:
//   PVTv2 spatial-reduction attention (D = DV = 64, 121 keys),
//   GMFlow split-window attention with optional shift (D = DV = 128, rows gathered
//     through index tables so that roll + window split + merge + roll-back are pure
//     addressing, additive -100 mask from region ids),
//   GMFlow global matching: row softmax of the all-pairs correlation times the pixel
//     grid (DV = 32, only 2 columns used) with the raw correlation written out,
//   flow propagation (DV = 32) and the EMIP-long memory read (D = DV = 128).
//
// Structure: 256 threads = 4 waves, each wave owns 32 queries; K/V tiles of
// 64 (bf16) / 32 (f32) keys are register-staged into two LDS buffers (one barrier
// per tile).  S^T = K Q^T is computed with the key on the MFMA row and the query on
// the lane (32x32 MFMA), so a lane holds the scores of ONE query: the online-softmax
// row statistics need a single cross-lane exchange (lane ^ 32) and the exponentiated
// accumulator registers are directly the B operand of O^T += V^T P (no LDS round trip
// for P).  V^T fragments come from the row-major V tile through ds_read_b64_tr_b16
// (bf16) or plain ds_read_b32 (f32).  K rows are XOR-swizzled for conflict-free
// ds_read_b128; the bf16 V tile is swizzled for conflict-free transposed reads.
#include "common.h"

extern "C" int emip_attention_rot(const void*, const void*, const void*, void*, void*, int, int, int, int, int, int, int, long,
                                  long, long, long, long, long, long, long, long, long, long, long, long, long, const int*,
                                  const int*, const int*, const int*, float, int, int, float*, int, int, void*);
extern "C" int emip_attention_splitkv(const void*, const void*, const void*, void*, void*, int, int, int, int, int, int, int, long,
                                      long, long, long, long, long, long, long, long, long, long, long, long, long, const int*,
                                      const int*, const int*, const int*, float, int, int, float*, int, void*);

namespace {

struct AttnArgs {
    const void* Q;
    const void* K;
    const void* V;
    void* O;
    void* S;
    int Lq, Lk, nwin, heads;
    long q_bs, k_bs, v_bs, o_bs, s_bs;
    long ldq, ldk, ldv, ldo, lds;
    long q_hs, k_hs, v_hs, o_hs;
    const int* q_rows;
    const int* k_rows;
    const int* q_gid;
    const int* k_gid;
    float scale;
    int o_f32;
    // KV split (long key sets on small grids): blockIdx.y = head * ksplit + split, every split walks 1/ksplit of the key
    // tiles and leaves (unnormalised O, running max, running sum) in ws [z][head][split][Lq][DV + 2] (f32); attn_merge_kernel
    // combines the splits and writes O
    int ksplit;
    float* ws;
    // keys / values of batch element b are read from element (b + kv_rot) mod nbatch (cross attention between the two
    // halves of a batch without a copy: kv_rot = nbatch / 2)
    int kv_rot, nbatch;
};

template <int RB>
__device__ __forceinline__ int kswz(int row) {
    return RB == 128 ? ((row >> 1) & 7) : (row & 15);
}

// byte offset of 16-B chunk `c` of row `row` inside a V tile
template <typename T, int RBV>
__device__ __forceinline__ int v_off(int row, int c) {
    if (sizeof(T) == 4) return row * RBV + c * 16;  // f32: plain (b32 row reads are conflict-free)
    if (RBV == 256) return row * RBV + ((c ^ ((row & 3) << 2)) * 16);
    if (RBV == 128) return row * RBV + ((c ^ (((row >> 1) & 1) << 2)) * 16);
    return row * RBV + c * 16;
}

// BKV = keys per K/V tile: 64 (bf16 streaming), 32 (f32), or 128 (bf16, whole K/V of the 121-key SRA
// attention resident in one tile: a single pass, no online rescale, one barrier).
template <typename T, int D, int DV, int BKV>
// amdgpu_waves_per_eu(2): left alone, hipcc gives the D = DV = 128 bf16 instance 256 VGPRs + 81 AGPRs (one wave per SIMD);
// asked for two waves it fits 246 VGPRs without a spill.  A wave that owns the whole register file of its SIMD also
// keeps every other stream's waves off it: 983 -> 1012 pairs/s in-call.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void attn_kernel(const AttnArgs p) {
    constexpr int ES = sizeof(T);
    constexpr bool BF = ES == 2;
    constexpr int NKT = BKV / 32;
    constexpr int RBK = D * ES, RBV = DV * ES;
    constexpr int NQ = RBK / 32;   // 16-B fragments per lane per Q/K row
    constexpr int NDT = DV / 32;
    constexpr int CPRK = RBK / 16, CPRV = RBV / 16;          // chunks per row
    constexpr int NSK = (BKV * CPRK + 255) / 256;            // staged chunks per thread
    constexpr int NSV = (BKV * CPRV + 255) / 256;
    constexpr int KT_BYTES = BKV * RBK, VT_BYTES = BKV * RBV;
    constexpr int STAGE_BYTES = KT_BYTES + VT_BYTES + BKV * 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, h = lane >> 5;
    const int z = blockIdx.z;
    const int win = z % p.nwin;
    const long batch = z / p.nwin;
    const int head = p.ksplit > 1 ? (int)blockIdx.y / p.ksplit : (int)blockIdx.y;
    const int split = p.ksplit > 1 ? (int)blockIdx.y - head * p.ksplit : 0;

    const T* __restrict__ Qp = reinterpret_cast<const T*>(p.Q) + batch * p.q_bs + head * p.q_hs;
    long kvb = batch + p.kv_rot;
    if (kvb >= p.nbatch) kvb -= p.nbatch;
    const T* __restrict__ Kp = reinterpret_cast<const T*>(p.K) + kvb * p.k_bs + head * p.k_hs;
    const T* __restrict__ Vp = reinterpret_cast<const T*>(p.V) + kvb * p.v_bs + head * p.v_hs;
    const int* qrows = p.q_rows ? p.q_rows + (long)win * p.Lq : nullptr;
    const int* krows = p.k_rows ? p.k_rows + (long)win * p.Lk : nullptr;
    const int* kgid = p.k_gid ? p.k_gid + (long)win * p.Lk : nullptr;

    // ---- this lane's query
    const int q = blockIdx.x * 128 + wave * 32 + lq;
    const bool q_ok = q < p.Lq;
    const long q_row = q_ok ? (qrows ? qrows[q] : q) : 0;
    const int q_g = (p.q_gid && q_ok) ? p.q_gid[(long)win * p.Lq + q] : 0;
    uint4 qf[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {   // unconditional load (q_row is 0 for out-of-range queries), masked afterwards
        const uint4 v = *reinterpret_cast<const uint4*>(Qp + q_row * p.ldq + (2 * i + h) * (16 / ES));
        qf[i] = mask4(v, q_ok);
    }

    // ---- staging bookkeeping
    uint4 rk[NSK], rv[NSV];
    int gid_reg = 0;
    auto load_tile = [&](int t) {
        const int k0 = t * BKV;
#pragma unroll
        for (int i = 0; i < NSK; ++i) {
            const int id = tid + 256 * i;
            const int r = id / CPRK, c = id - r * CPRK;
            const bool ok = r < BKV && k0 + r < p.Lk;
            const int kr = min(k0 + r, p.Lk - 1);          // clamped: loads stay unconditional (no per-load branch+wait)
            const long gr = krows ? krows[kr] : kr;
            const uint4 v = *reinterpret_cast<const uint4*>(Kp + gr * p.ldk + c * (16 / ES));
            rk[i] = mask4(v, ok);
        }
#pragma unroll
        for (int i = 0; i < NSV; ++i) {
            const int id = tid + 256 * i;
            const int r = id / CPRV, c = id - r * CPRV;
            const bool ok = r < BKV && k0 + r < p.Lk;
            const int kr = min(k0 + r, p.Lk - 1);
            const long gr = krows ? krows[kr] : kr;
            const uint4 v = *reinterpret_cast<const uint4*>(Vp + gr * p.ldv + c * (16 / ES));
            rv[i] = mask4(v, ok);
        }
        if (kgid) gid_reg = kgid[min(k0 + (tid & (BKV - 1)), p.Lk - 1)];
    };
    auto store_tile = [&](int buf) {
        char* kt_ = smem + buf * STAGE_BYTES;
        char* vt_ = kt_ + KT_BYTES;
#pragma unroll
        for (int i = 0; i < NSK; ++i) {
            const int id = tid + 256 * i;
            const int r = id / CPRK, c = id - r * CPRK;
            if (r < BKV) *reinterpret_cast<uint4*>(kt_ + r * RBK + ((c ^ kswz<RBK>(r)) * 16)) = rk[i];
        }
#pragma unroll
        for (int i = 0; i < NSV; ++i) {
            const int id = tid + 256 * i;
            const int r = id / CPRV, c = id - r * CPRV;
            if (r < BKV) *reinterpret_cast<uint4*>(vt_ + v_off<T, RBV>(r, c)) = rv[i];
        }
        if (kgid && tid < BKV) reinterpret_cast<int*>(vt_ + VT_BYTES)[tid] = gid_reg;
    };

    f32x16 oacc[NDT];
#pragma unroll
    for (int d = 0; d < NDT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sc2 = p.scale * 1.4426950408889634f;  // scores in log2 units
    const float mask2 = -100.0f * 1.4426950408889634f;

    T* __restrict__ Sp = p.S ? reinterpret_cast<T*>(p.S) + (long)z * p.s_bs + (long)head * 0 : nullptr;

    const int ntile_all = (p.Lk + BKV - 1) / BKV;
    const int tps = p.ksplit > 1 ? (ntile_all + p.ksplit - 1) / p.ksplit : ntile_all;
    const int t_lo = split * tps;
    const int ntile = min(ntile_all, t_lo + tps);          // this workgroup walks tiles [t_lo, ntile)
    if (t_lo < ntile) {
        load_tile(t_lo);
        store_tile(t_lo & 1);
    }
    __syncthreads();
    for (int t = t_lo; t < ntile; ++t) {
        const int cur = t & 1;
        if (t + 1 < ntile) load_tile(t + 1);
        const char* kt_ = smem + cur * STAGE_BYTES;
        const char* vt_ = kt_ + KT_BYTES;
        const int* gl = reinterpret_cast<const int*>(vt_ + VT_BYTES);

        // ---- S^T = K Q^T
        f32x16 s[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
            const int row = 32 * kt + lq;
            const char* rp = kt_ + row * RBK;
            const int sw = kswz<RBK>(row);
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const uint4 kf = *reinterpret_cast<const uint4*>(rp + (((2 * i + h) ^ sw) * 16));
                if (BF) {
                    s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf),
                                                                    __builtin_bit_cast(bf16x8, qf[i]), s[kt], 0, 0, 0);
                } else {
                    s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(kf.x), __uint_as_float(qf[i].x), s[kt], 0, 0, 0);
                    s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(kf.y), __uint_as_float(qf[i].y), s[kt], 0, 0, 0);
                    s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(kf.z), __uint_as_float(qf[i].z), s[kt], 0, 0, 0);
                    s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(kf.w), __uint_as_float(qf[i].w), s[kt], 0, 0, 0);
                }
            }
        }

        // ---- raw scores out (the correlation volume), masks, online softmax
        float tmax = -INFINITY;
        const bool last_partial = (t == ntile_all - 1) && (p.Lk % BKV != 0);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int kbase = t * BKV + 32 * kt + 8 * g + 4 * h;  // 4 consecutive keys
                if (Sp && q_ok) {
                    float v4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v4[j] = s[kt][4 * g + j] * p.scale;
                    if (kbase + 3 < p.Lk && (p.lds & 3) == 0) {
                        Vec4<T>::store(Sp + (long)q * p.lds + kbase, v4);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (kbase + j < p.Lk) Sp[(long)q * p.lds + kbase + j] = from_f32<T>(v4[j]);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = s[kt][4 * g + j] * sc2;
                    if (kgid) {        // shifted-window mask only where a table was passed
                        const int kg = gl[32 * kt + 8 * g + 4 * h + j];
                        v += (kg != q_g) ? mask2 : 0.f;
                    }
                    s[kt][4 * g + j] = v;
                }
                if (last_partial) {    // wave-uniform: only the last K/V tile can hold keys >= Lk
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (kbase + j >= p.Lk) s[kt][4 * g + j] = -INFINITY;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) tmax = fmaxf(tmax, s[kt][4 * g + j]);
            }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(s[kt][r] - m_new);
                s[kt][r] = e;
                psum += e;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int d = 0; d < NDT; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;

        // ---- O^T += V^T P
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (BF) {
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    bf16x8 pf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)s[kt][8 * sp + j];
                    const int i16 = lane & 15, g16 = (lane >> 4) & 1;
                    const int base0 = 32 * kt + 16 * sp + 4 * h + (i16 >> 2);
#pragma unroll
                    for (int d = 0; d < NDT; ++d) {
                        // column (bf16 index) 32d + 16*g16 + 4*(i16&3): chunk + 8-byte half
                        const int col = 32 * d + 16 * g16 + 4 * (i16 & 3);
                        const int c = col >> 3, half = (col >> 2) & 1;
                        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
                        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (lds_s16x4*)(vt_ + v_off<T, RBV>(base0, c) + 8 * half));
                        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (lds_s16x4*)(vt_ + v_off<T, RBV>(base0 + 8, c) + 8 * half));
                        bf16x8 vf;
                        const bf16x4 b0 = __builtin_bit_cast(bf16x4, v0), b1 = __builtin_bit_cast(bf16x4, v1);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            vf[j] = b0[j];
                            vf[4 + j] = b1[j];
                        }
                        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[d], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
                    for (int d = 0; d < NDT; ++d) {
                        const float vv = *reinterpret_cast<const float*>(vt_ + key * RBV + (32 * d + lq) * 4);
                        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, s[kt][r], oacc[d], 0, 0, 0);
                    }
                }
            }
        }
        if (t + 1 < ntile) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    if (p.ksplit > 1) {
        // partial result of this key range: unnormalised O (f32), running max (log2 units) and running sum
        if (q_ok) {
            float* wp = p.ws + ((((long)z * p.heads + head) * p.ksplit + split) * p.Lq + q) * (DV + 2);
#pragma unroll
            for (int d = 0; d < NDT; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v4[j] = oacc[d][4 * g + j];
                    *reinterpret_cast<float2*>(wp + 32 * d + 8 * g + 4 * h) = make_float2(v4[0], v4[1]);
                    *reinterpret_cast<float2*>(wp + 32 * d + 8 * g + 4 * h + 2) = make_float2(v4[2], v4[3]);
                }
            if (h == 0) {
                wp[DV] = m_run;
                wp[DV + 1] = l_tot;
            }
        }
        return;
    }
    const float inv = 1.0f / l_tot;
    if (q_ok) {
        const long orow = q_row;
        if (p.o_f32) {
            float* Op = reinterpret_cast<float*>(p.O) + batch * p.o_bs + head * p.o_hs + orow * p.ldo;
#pragma unroll
            for (int d = 0; d < NDT; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v4[j] = oacc[d][4 * g + j] * inv;
                    Vec4<float>::store(Op + 32 * d + 8 * g + 4 * h, v4);
                }
        } else {
            T* Op = reinterpret_cast<T*>(p.O) + batch * p.o_bs + head * p.o_hs + orow * p.ldo;
#pragma unroll
            for (int d = 0; d < NDT; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v4[j] = oacc[d][4 * g + j] * inv;
                    Vec4<T>::store(Op + 32 * d + 8 * g + 4 * h, v4);
                }
        }
    }
}

// combine the key-range partials: O = sum_s O_s 2^(m_s - M) / sum_s l_s 2^(m_s - M)
template <typename T, int DV>
__global__ __launch_bounds__(256) void attn_merge_kernel(const AttnArgs p, int zcount) {
    const long total = (long)zcount * p.heads * p.Lq * (DV / 4);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % (DV / 4));
        long r = idx / (DV / 4);
        const int q = (int)(r % p.Lq);
        r /= p.Lq;
        const int head = (int)(r % p.heads);
        const long z = r / p.heads;
        const float* wp = p.ws + (((z * p.heads + head) * p.ksplit) * (long)p.Lq + q) * (DV + 2);
        const long sstride = (long)p.Lq * (DV + 2);
        float M = -INFINITY;
        for (int s = 0; s < p.ksplit; ++s) M = fmaxf(M, wp[s * sstride + DV]);
        float L = 0.f, o[4] = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < p.ksplit; ++s) {
            const float m = wp[s * sstride + DV];
            const float f = m == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m - M);      // an empty key range contributes nothing
            L += wp[s * sstride + DV + 1] * f;
            const float4 v = *reinterpret_cast<const float4*>(wp + s * sstride + 4 * c4);
            o[0] += v.x * f; o[1] += v.y * f; o[2] += v.z * f; o[3] += v.w * f;
        }
        const float inv = 1.f / L;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] *= inv;
        const int win = (int)(z % p.nwin);
        const long batch = z / p.nwin;
        const long orow = p.q_rows ? p.q_rows[(long)win * p.Lq + q] : q;
        if (p.o_f32) {
            Vec4<float>::store(reinterpret_cast<float*>(p.O) + batch * p.o_bs + head * p.o_hs + orow * p.ldo + 4 * c4, o);
        } else {
            Vec4<T>::store(reinterpret_cast<T*>(p.O) + batch * p.o_bs + head * p.o_hs + orow * p.ldo + 4 * c4, o);
        }
    }
}

template <typename T, int D, int DV, int BKV>
int launch_bkv(const AttnArgs& a, int batch, hipStream_t s) {
    constexpr int ES = sizeof(T);
    const int ntile = (a.Lk + BKV - 1) / BKV;
    const size_t lds = (ntile > 1 ? 2 : 1) * (size_t)(BKV * D * ES + BKV * DV * ES + BKV * 4);
    auto kfn = attn_kernel<T, D, DV, BKV>;
    static bool attr_done = false;  // per instantiation; set on the first (warm-up) launch, before any capture
    if (lds > 64 * 1024 && !attr_done) {
        attr_done = true;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return EMIP_E_LAUNCH;
    }
    dim3 grid((a.Lq + 127) / 128, a.heads * (a.ksplit > 1 ? a.ksplit : 1), batch * a.nwin);
    hipLaunchKernelGGL(kfn, grid, dim3(256), lds, s, a);
    if (a.ksplit > 1) {
        const long items = (long)batch * a.nwin * a.heads * a.Lq * (DV / 4);
        long blocks = (items + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL((attn_merge_kernel<T, DV>), dim3((unsigned)blocks), dim3(256), 0, s, a, batch * a.nwin);
    }
    return emip_launch_status();
}

template <typename T, int D, int DV>
int launch(const AttnArgs& a, int batch, hipStream_t s) {
    // BKV = 128 (whole 121-key SRA K/V in one tile) measured slower than two 64-key tiles on MI355X
    // (240 vs 166 VGPRs -> one fewer wave per SIMD), so bf16 always streams 64-key tiles.
    if constexpr (sizeof(T) == 4) {
        return launch_bkv<T, D, DV, 32>(a, batch, s);
    } else {
        return launch_bkv<T, D, DV, 64>(a, batch, s);
    }
}

}  // namespace

extern "C" int emip_attention(const void* Q, const void* K, const void* V, void* O, void* S, int batch, int heads,
                              int nwin, int Lq, int Lk, int D, int DV, long q_bs, long k_bs, long v_bs, long o_bs,
                              long s_bs, long ldq, long ldk, long ldv, long ldo, long lds, long q_hs, long k_hs,
                              long v_hs, long o_hs, const int* q_rows, const int* k_rows, const int* q_gid,
                              const int* k_gid, float scale, int o_f32, int dtype, void* stream) {
    return emip_attention_splitkv(Q, K, V, O, S, batch, heads, nwin, Lq, Lk, D, DV, q_bs, k_bs, v_bs, o_bs, s_bs, ldq, ldk,
                                  ldv, ldo, lds, q_hs, k_hs, v_hs, o_hs, q_rows, k_rows, q_gid, k_gid, scale, o_f32, 1,
                                  nullptr, dtype, stream);
}

// emip_attention with the key range split over ksplit workgroups per query tile (flash-decoding style): for long key sets on
// small grids (global matching / flow propagation: 1936 keys, the EMIP-long memory read: up to 9680) a workgroup otherwise
// walks all key tiles serially while most of the chip idles.  ws: f32 [batch*nwin][heads][ksplit][Lq][DV + 2] scratch; a
// second small launch merges the partial softmaxes and writes O.
extern "C" int emip_attention_splitkv(const void* Q, const void* K, const void* V, void* O, void* S, int batch, int heads,
                                      int nwin, int Lq, int Lk, int D, int DV, long q_bs, long k_bs, long v_bs, long o_bs,
                                      long s_bs, long ldq, long ldk, long ldv, long ldo, long lds, long q_hs, long k_hs,
                                      long v_hs, long o_hs, const int* q_rows, const int* k_rows, const int* q_gid,
                                      const int* k_gid, float scale, int o_f32, int ksplit, float* ws, int dtype,
                                      void* stream) {
    return emip_attention_rot(Q, K, V, O, S, batch, heads, nwin, Lq, Lk, D, DV, q_bs, k_bs, v_bs, o_bs, s_bs, ldq, ldk, ldv,
                              ldo, lds, q_hs, k_hs, v_hs, o_hs, q_rows, k_rows, q_gid, k_gid, scale, o_f32, ksplit, ws, 0,
                              dtype, stream);
}

// ... with the keys / values of batch element b taken from element (b + kv_batch_rot) mod batch: GMFlow's cross attention
// between the two frames of a pair (transformer.py:281-301: source = the OTHER frame's tokens) reads the other half of the
// batch in place, so the k / v projections of both frames stay ONE GEMM over the whole batch.
extern "C" int emip_attention_rot(const void* Q, const void* K, const void* V, void* O, void* S, int batch, int heads,
                                  int nwin, int Lq, int Lk, int D, int DV, long q_bs, long k_bs, long v_bs, long o_bs,
                                  long s_bs, long ldq, long ldk, long ldv, long ldo, long lds, long q_hs, long k_hs,
                                  long v_hs, long o_hs, const int* q_rows, const int* k_rows, const int* q_gid,
                                  const int* k_gid, float scale, int o_f32, int ksplit, float* ws, int kv_batch_rot,
                                  int dtype, void* stream) {
    EMIP_REQUIRE(kv_batch_rot >= 0 && kv_batch_rot < batch);
    EMIP_REQUIRE(ksplit >= 1 && ksplit <= 32 && (ksplit == 1 || (ws && (reinterpret_cast<uintptr_t>(ws) & 15) == 0)));
    EMIP_REQUIRE((long)heads * ksplit < 65536);
    EMIP_REQUIRE(Q && K && V && O && batch > 0 && heads > 0 && nwin > 0 && Lq > 0 && Lk > 0);
    EMIP_REQUIRE(dtype == EMIP_F32 || dtype == EMIP_BF16);
    EMIP_REQUIRE((long)batch * nwin < 65536 && heads < 65536);
    const int vec = dtype == EMIP_F32 ? 4 : 8;
    EMIP_REQUIRE(ldq % vec == 0 && ldk % vec == 0 && ldv % vec == 0 && q_bs % vec == 0 && k_bs % vec == 0 &&
                 v_bs % vec == 0 && q_hs % vec == 0 && k_hs % vec == 0 && v_hs % vec == 0);
    EMIP_REQUIRE(ldq >= D && ldk >= D && ldv >= DV && ldo >= DV && (ldo & 3) == 0 && (o_hs & 3) == 0 &&
                 (o_bs & 3) == 0);
    EMIP_REQUIRE(aligned16(Q) && aligned16(K) && aligned16(V) && aligned16(O));
    EMIP_REQUIRE((q_gid == nullptr) == (k_gid == nullptr));
    if (S) EMIP_REQUIRE(heads == 1 && lds >= Lk && s_bs >= (long)Lq * lds - (lds - Lk) && q_rows == nullptr);
    AttnArgs a{};
    a.Q = Q; a.K = K; a.V = V; a.O = O; a.S = S;
    a.Lq = Lq; a.Lk = Lk; a.nwin = nwin; a.heads = heads;
    a.q_bs = q_bs; a.k_bs = k_bs; a.v_bs = v_bs; a.o_bs = o_bs; a.s_bs = s_bs;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.lds = lds;
    a.q_hs = q_hs; a.k_hs = k_hs; a.v_hs = v_hs; a.o_hs = o_hs;
    a.q_rows = q_rows; a.k_rows = k_rows; a.q_gid = q_gid; a.k_gid = k_gid;
    a.scale = scale; a.o_f32 = o_f32;
    a.ksplit = ksplit; a.ws = ws;
    a.kv_rot = kv_batch_rot; a.nbatch = batch;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define ATTN_CASE(d, dv)                                                                     \
    if (D == d && DV == dv)                                                                  \
        return dtype == EMIP_F32 ? launch<float, d, dv>(a, batch, s) : launch<bf16_t, d, dv>(a, batch, s);
    ATTN_CASE(64, 64)
    ATTN_CASE(128, 128)
    ATTN_CASE(128, 32)
#undef ATTN_CASE
    return EMIP_E_INVALID;
}
