// Fused multi-tensor optimizer step of the EMIP training loop:
//   clip_gradient (element-wise clamp to +-clip, /root/reference/utils/utils.py:1-11)  +
//   AdamW (torch.optim.AdamW semantics; /root/reference/train.py:380: lr 1e-5, wd 1e-7, betas (0.9, 0.999), eps 1e-8)
// over all trainable tensors in ONE launch.  HBM-bound: 16 B/element read (p, g, m, v) + 12 B written.
// The host builds, once, a table of tensors and a block->(tensor, offset) map; each workgroup updates
// one CHUNK-element slice of one tensor with 16-byte vector accesses.
#include "common.h"

namespace {

constexpr int CHUNK = 2048;  // elements per workgroup (256 threads x 2 x float4)

struct TensorRec {
    float* p;
    const float* g;     // used when the launch passes no separate gradient-pointer array
    float* m;
    float* v;
    long n;
};

__global__ __launch_bounds__(256) void clamp_adamw_kernel(const TensorRec* __restrict__ recs,
                                                          const int2* __restrict__ blockmap, float lr, float beta1,
                                                          float beta2, float eps, float wd, float clip, float bc1,
                                                          float bc2_sqrt, const float* const* __restrict__ gptrs) {
    const int2 bm = blockmap[blockIdx.x];
    TensorRec r = recs[bm.x];
    if (gptrs) r.g = gptrs[bm.x];
    const long base = (long)bm.y * CHUNK;
    const float step = lr / bc1;
#pragma unroll
    for (int it = 0; it < CHUNK / 1024; ++it) {
        const long i = base + it * 1024 + threadIdx.x * 4;
        if (i + 3 < r.n && ((reinterpret_cast<uintptr_t>(r.p + i) & 15) == 0)) {
            float4 p = *reinterpret_cast<float4*>(r.p + i);
            const float4 g4 = *reinterpret_cast<const float4*>(r.g + i);
            float4 m = *reinterpret_cast<float4*>(r.m + i), v = *reinterpret_cast<float4*>(r.v + i);
            float* pp = &p.x; const float* gg = &g4.x; float* mm = &m.x; float* vv = &v.x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float g = clip > 0.f ? fminf(fmaxf(gg[j], -clip), clip) : gg[j];
                pp[j] *= (1.f - lr * wd);
                mm[j] = beta1 * mm[j] + (1.f - beta1) * g;
                vv[j] = beta2 * vv[j] + (1.f - beta2) * g * g;
                pp[j] -= step * mm[j] / (sqrtf(vv[j]) / bc2_sqrt + eps);
            }
            *reinterpret_cast<float4*>(r.p + i) = p;
            *reinterpret_cast<float4*>(r.m + i) = m;
            *reinterpret_cast<float4*>(r.v + i) = v;
        } else {
            for (long k = i; k < i + 4 && k < r.n; ++k) {
                const float g0 = r.g[k];
                const float g = clip > 0.f ? fminf(fmaxf(g0, -clip), clip) : g0;
                float p = r.p[k] * (1.f - lr * wd);
                const float m = beta1 * r.m[k] + (1.f - beta1) * g;
                const float v = beta2 * r.v[k] + (1.f - beta2) * g * g;
                p -= step * m / (sqrtf(v) / bc2_sqrt + eps);
                r.p[k] = p; r.m[k] = m; r.v[k] = v;
            }
        }
    }
}

// ---- packed-weight refresh ------------------------------------------------------------------------------------------------
// After an optimizer step every kernel-ready copy of a trainable weight (bf16 [N][K], its transpose for the input gradient,
// conv packs [Cout][KH][KW][Cin_pad], flipped / transposed conv packs, depthwise [9][C]) has to be rebuilt from the f32
// master.  Done with torch ops that is ~1 300 tiny launches per EMIP-short step (7 ms); here it is ONE launch over a table
// of generalised permutations: dst (contiguous, up to 4 dims) <- src[base + i0 s0 + i1 s1 + i2 s2 + i3 s3] (strides may be
// negative: kernel flips), zero where i3 >= valid3 (channel padding), cast to bf16 or kept f32.
struct RepackRec {
    const float* src;
    void* dst;
    long n;               // dst elements = d0 d1 d2 d3
    long s0, s1, s2, s3;  // source strides in elements
    long base;
    int d1, d2, d3, valid3;
    int dst_bf16;
    float scale;          // the copy is multiplied by it; 0 (the all-zero record tail of older tables) = 1
};

constexpr int RCHUNK = 2048;  // dst elements per workgroup: 256 threads x 8

__global__ __launch_bounds__(256) void repack_kernel(const RepackRec* __restrict__ recs, const int2* __restrict__ blockmap) {
    const int2 bm = blockmap[blockIdx.x];
    const RepackRec r = recs[bm.x];
    const long i = (long)bm.y * RCHUNK + threadIdx.x * 8;
    if (i >= r.n) return;
    float v[8];
    const float sc = r.scale != 0.f ? r.scale : 1.f;
    long q = i / r.d3;
    int i3 = (int)(i - q * r.d3);
    long q2 = q / r.d2;
    int i2 = (int)(q - q2 * r.d2);
    long i0 = q2 / r.d1;
    int i1 = (int)(q2 - i0 * r.d1);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const bool in = i + j < r.n;
        v[j] = (in && i3 < r.valid3) ? sc * r.src[r.base + i0 * r.s0 + i1 * r.s1 + i2 * r.s2 + i3 * r.s3] : 0.f;
        if (++i3 == r.d3) {
            i3 = 0;
            if (++i2 == r.d2) {
                i2 = 0;
                if (++i1 == r.d1) { i1 = 0; ++i0; }
            }
        }
    }
    if (r.dst_bf16) {
        bf16_t* d = reinterpret_cast<bf16_t*>(r.dst) + i;
        if (i + 7 < r.n && (reinterpret_cast<uintptr_t>(d) & 15) == 0) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16_t)v[j];
            *reinterpret_cast<bf16x8*>(d) = o;
        } else {
            for (int j = 0; j < 8 && i + j < r.n; ++j) d[j] = (bf16_t)v[j];
        }
    } else {
        float* d = reinterpret_cast<float*>(r.dst) + i;
        for (int j = 0; j < 8 && i + j < r.n; ++j) d[j] = v[j];
    }
}

// Train-mode BatchNorm bookkeeping (nn.BatchNorm2d.forward in training: running statistics with the unbiased variance,
// num_batches_tracked += 1) from the f64 column sums the normalisation itself used: replaces 14 torch launches per layer.
__global__ __launch_bounds__(256) void bn_running_kernel(const double* __restrict__ sums, float* __restrict__ rmean,
                                                         float* __restrict__ rvar, long long* __restrict__ tracked, long n,
                                                         float momentum, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && tracked) *tracked += 1;
    if (c >= C) return;
    const double mean = sums[2 * c] / (double)n;
    double var = sums[2 * c + 1] / (double)n - mean * mean;
    if (var < 0.0) var = 0.0;
    const double unb = var * (double)n / (double)(n > 1 ? n - 1 : 1);
    rmean[c] = rmean[c] * (1.f - momentum) + (float)mean * momentum;
    rvar[c] = rvar[c] * (1.f - momentum) + (float)unb * momentum;
}

// ---- data-parallel gradient buckets ----------------------------------------------------------------------------------------
// The gradient all-reduce of the training step (/root/reference/train.py:279: DDP's flat buckets) moves CONTIGUOUS buckets;
// the gradients are ~1 300 separate f32 tensors.  One launch per bucket gathers them into its slice of the flat transport
// buffer (f32, or bf16 on the wire), one launch per step scatters the reduced values back into the .grad tensors times
// 1 / world.  Tables as above: GradRec per tensor, blockmap (record, chunk), the gradient pointers in their own array.
struct GradRec {
    long off;             // element offset of the tensor inside the flat buffer
    long n;
};

template <bool PACK, bool BF>
__global__ __launch_bounds__(256) void grad_bucket_kernel(const GradRec* __restrict__ recs, const int2* __restrict__ blockmap,
                                                          float* const* __restrict__ gptrs, void* __restrict__ flat,
                                                          float scale) {
    const int2 bm = blockmap[blockIdx.x];
    const GradRec r = recs[bm.x];
    float* g = gptrs[bm.x];
    if (!PACK && g == nullptr) return;                    // a parameter without a gradient this step keeps none
    const long base = (long)bm.y * CHUNK;
#pragma unroll
    for (int it = 0; it < CHUNK / 1024; ++it) {
        const long i = base + it * 1024 + threadIdx.x * 4;
        if (i >= r.n) return;
        const long f = r.off + i;
        float v[4];
        // whole, aligned quads: 16-byte accesses on the f32 side (8 on a bf16 flat buffer)
        if (g != nullptr && i + 3 < r.n && (reinterpret_cast<uintptr_t>(g + i) & 15) == 0 && (f & 3) == 0) {
            if (PACK) {
                const float4 t4 = *reinterpret_cast<const float4*>(g + i);
                if (BF) {
                    bf16x4 o;
                    o[0] = (bf16_t)t4.x; o[1] = (bf16_t)t4.y; o[2] = (bf16_t)t4.z; o[3] = (bf16_t)t4.w;
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(flat) + f) = o;
                } else {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(flat) + f) = t4;
                }
            } else {
                float4 t4;
                if (BF) {
                    const bf16x4 o = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(flat) + f);
                    t4 = make_float4((float)o[0], (float)o[1], (float)o[2], (float)o[3]);
                } else {
                    t4 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(flat) + f);
                }
                *reinterpret_cast<float4*>(g + i) = make_float4(scale * t4.x, scale * t4.y, scale * t4.z, scale * t4.w);
            }
            continue;
        }
        if (PACK) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (g != nullptr && i + j < r.n) ? g[i + j] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i + j < r.n) {
                    if (BF) reinterpret_cast<bf16_t*>(flat)[f + j] = (bf16_t)v[j];
                    else reinterpret_cast<float*>(flat)[f + j] = v[j];
                }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i + j < r.n)
                    g[i + j] = scale * (BF ? (float)reinterpret_cast<const bf16_t*>(flat)[f + j]
                                           : reinterpret_cast<const float*>(flat)[f + j]);
        }
    }
}

// out[i] = sum_w in[w][i] with f32 accumulation (the reduce step of the direct reduce-scatter: shard i of every rank)
template <bool BF>
__global__ __launch_bounds__(256) void shard_sum_kernel(const void* __restrict__ in, void* __restrict__ out, int world,
                                                        long chunk) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < chunk; i += (long)gridDim.x * blockDim.x) {
        float a = 0.f;
        for (int w = 0; w < world; ++w)
            a += BF ? (float)reinterpret_cast<const bf16_t*>(in)[w * chunk + i] : reinterpret_cast<const float*>(in)[w * chunk + i];
        if (BF) reinterpret_cast<bf16_t*>(out)[i] = (bf16_t)a;
        else reinterpret_cast<float*>(out)[i] = a;
    }
}

}  // namespace

// recs: device array of {element offset inside flat, element count} (2 x 8 bytes) per gradient tensor, blockmap: device int2
// [nblocks] = (record, chunk of emip_adamw_chunk() elements), gptrs: device array of the tensors' f32 pointers (null = no
// gradient: packs zeros / is skipped on the way back).  flat: the transport buffer, f32 or (flat_bf16) bf16.
extern "C" int emip_grad_pack(const void* recs, const void* blockmap, const void* gptrs, int nblocks, void* flat,
                              int flat_bf16, void* stream) {
    EMIP_REQUIRE(recs && blockmap && gptrs && flat && nblocks > 0);
    static_assert(sizeof(GradRec) == 16, "host-side table layout");
    if (flat_bf16)
        hipLaunchKernelGGL((grad_bucket_kernel<true, true>), dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const GradRec*)recs,
                           (const int2*)blockmap, (float* const*)gptrs, flat, 1.f);
    else
        hipLaunchKernelGGL((grad_bucket_kernel<true, false>), dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const GradRec*)recs,
                           (const int2*)blockmap, (float* const*)gptrs, flat, 1.f);
    return emip_launch_status();
}

// the way back: gradient tensor <- scale * flat slice (scale = 1 / world: the mean of the replicas' gradients)
extern "C" int emip_grad_unpack(const void* recs, const void* blockmap, const void* gptrs, int nblocks, const void* flat,
                                int flat_bf16, float scale, void* stream) {
    EMIP_REQUIRE(recs && blockmap && gptrs && flat && nblocks > 0);
    if (flat_bf16)
        hipLaunchKernelGGL((grad_bucket_kernel<false, true>), dim3(nblocks), dim3(256), 0, (hipStream_t)stream,
                           (const GradRec*)recs, (const int2*)blockmap, (float* const*)gptrs, const_cast<void*>(flat), scale);
    else
        hipLaunchKernelGGL((grad_bucket_kernel<false, false>), dim3(nblocks), dim3(256), 0, (hipStream_t)stream,
                           (const GradRec*)recs, (const int2*)blockmap, (float* const*)gptrs, const_cast<void*>(flat), scale);
    return emip_launch_status();
}

// out [chunk] = sum over the `world` rows of in [world, chunk], accumulated in f32; both in f32 or both in bf16
extern "C" int emip_shard_sum(const void* in, void* out, int world, long chunk, int is_bf16, void* stream) {
    EMIP_REQUIRE(in && out && world > 0 && chunk > 0);
    long blocks = (chunk + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (is_bf16)
        hipLaunchKernelGGL((shard_sum_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, out, world, chunk);
    else
        hipLaunchKernelGGL((shard_sum_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, out, world, chunk);
    return emip_launch_status();
}

// recs: device array of RepackRec (see above; 88 bytes each), blockmap: device int2 [nblocks] = (record, chunk) with chunks
// of emip_repack_chunk() dst elements.
extern "C" int emip_repack(const void* recs, const void* blockmap, int nblocks, void* stream) {
    EMIP_REQUIRE(recs && blockmap && nblocks > 0);
    static_assert(sizeof(RepackRec) == 88, "host-side table layout");
    hipLaunchKernelGGL(repack_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const RepackRec*)recs,
                       (const int2*)blockmap);
    return emip_launch_status();
}
extern "C" int emip_repack_chunk(void) { return RCHUNK; }

// sums: f64 [C][2] (sum, sum of squares over n values per channel, as emip_chan_stats leaves them for one group)
extern "C" int emip_bn_running_update(const double* sums, float* running_mean, float* running_var, long long* tracked,
                                      long n, float momentum, int C, void* stream) {
    EMIP_REQUIRE(sums && running_mean && running_var && n > 0 && C > 0 && momentum >= 0.f && momentum <= 1.f);
    hipLaunchKernelGGL(bn_running_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, running_mean,
                       running_var, tracked, n, momentum, C);
    return emip_launch_status();
}

// recs: device array of {p, g, m, v (f32 device pointers), n} records (5 x 8 bytes each); blockmap: device int2
// [nblocks] = (record index, chunk index) with chunks of 2048 elements.  step >= 1 is the AdamW step count.
extern "C" int emip_clamp_adamw(const void* recs, const void* blockmap, int nblocks, float lr, float beta1,
                                float beta2, float eps, float weight_decay, float clip, int step, void* stream) {
    return emip_clamp_adamw_g(recs, blockmap, nullptr, nblocks, lr, beta1, beta2, eps, weight_decay, clip, step, stream);
}

// The same with the gradient pointers in their own device array (const float* [records]): parameters and moments stay where
// they are from step to step, gradients are fresh tensors every step, so only this small array changes between launches.
extern "C" int emip_clamp_adamw_g(const void* recs, const void* blockmap, const void* gptrs, int nblocks, float lr, float beta1,
                                  float beta2, float eps, float weight_decay, float clip, int step, void* stream) {
    EMIP_REQUIRE(recs && blockmap && nblocks > 0 && step >= 1 && lr >= 0.f && beta1 >= 0.f && beta1 < 1.f &&
                 beta2 >= 0.f && beta2 < 1.f);
    // bias corrections in double on the host, as torch.optim.AdamW evaluates them (float powf is ~6e-5 relative off at small t)
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    hipLaunchKernelGGL(clamp_adamw_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream,
                       (const TensorRec*)recs, (const int2*)blockmap, lr, beta1, beta2, eps, weight_decay, clip, bc1,
                       bc2s, (const float* const*)gptrs);
    return emip_launch_status();
}

extern "C" int emip_adamw_chunk(void) { return CHUNK; }
