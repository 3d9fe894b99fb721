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
    const float* g;
    float* m;
    float* v;
    long n;
};

__global__ __launch_bounds__(256) void clamp_adamw_kernel(const TensorRec* __restrict__ recs,
                                                          const int2* __restrict__ blockmap, float lr, float beta1,
                                                          float beta2, float eps, float wd, float clip, float bc1,
                                                          float bc2_sqrt) {
    const int2 bm = blockmap[blockIdx.x];
    const TensorRec r = recs[bm.x];
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

}  // namespace

// recs: device array of {p, g, m, v (f32 device pointers), n} records (5 x 8 bytes each); blockmap: device int2
// [nblocks] = (record index, chunk index) with chunks of 2048 elements.  step >= 1 is the AdamW step count.
extern "C" int emip_clamp_adamw(const void* recs, const void* blockmap, int nblocks, float lr, float beta1,
                                float beta2, float eps, float weight_decay, float clip, int step, void* stream) {
    EMIP_REQUIRE(recs && blockmap && nblocks > 0 && step >= 1 && lr >= 0.f && beta1 >= 0.f && beta1 < 1.f &&
                 beta2 >= 0.f && beta2 < 1.f);
    // bias corrections in double on the host, as torch.optim.AdamW evaluates them (float powf is ~6e-5 relative off at small t)
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    hipLaunchKernelGGL(clamp_adamw_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream,
                       (const TensorRec*)recs, (const int2*)blockmap, lr, beta1, beta2, eps, weight_decay, clip, bc1,
                       bc2s);
    return emip_launch_status();
}

extern "C" int emip_adamw_chunk(void) { return CHUNK; }
