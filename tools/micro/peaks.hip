// Peak confirmation on the box ("nominal peaks to be confirmed by a microbenchmark before use"):
//   * dense bf16 MFMA rate: every SIMD of the chip issues independent v_mfma_f32_32x32x16_bf16 on register operands
//     (4 accumulator chains per wave, 2 and 4 waves per SIMD), nothing else in the loop;
//   * HBM: streaming read (sum into a register, one store per thread), streaming write, and copy over 8 GB -- far beyond the
//     256 MB Infinity Cache -- with 16 bytes per lane.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/peaks.hip -o tools/bin/peaks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(256) void mfma_kernel(float* out, int iters) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)(0.001f * (threadIdx.x + j));
        b[j] = (__bf16)(0.002f * (threadIdx.x + 2 * j));
    }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += c0[j] + c1[j] + c2[j] + c3[j];
    if (s == 12345.678f) out[0] = s;                      // keeps the chains alive without a store per thread
}

__global__ __launch_bounds__(256) void read_kernel(const uint4* __restrict__ p, uint4* __restrict__ sink, size_t n) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void write_kernel(uint4* __restrict__ p, size_t n) {
    const uint4 v = make_uint4(1, 2, 3, 4);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ __launch_bounds__(256) void copy_kernel(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

template <typename F>
static float time_us(F f, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1e3f;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("%s: %d CUs, clock %d MHz, memory clock %d MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000,
           prop.memoryClockRate / 1000);
    float* out;
    hipMalloc(&out, 256);
    const int iters = 20000;
    for (int wg_per_cu : {2, 4}) {                        // 256 threads = 4 waves = one per SIMD per workgroup
        const int blocks = prop.multiProcessorCount * wg_per_cu;
        const float us = time_us([&] { hipLaunchKernelGGL(mfma_kernel, dim3(blocks), dim3(256), 0, 0, out, iters); }, 3);
        const double flop = (double)blocks * 4 * iters * 4 * 2.0 * 32 * 32 * 16;
        printf("bf16 MFMA 32x32x16, %d waves per SIMD: %.1f TFLOP/s\n", wg_per_cu, flop / us / 1e6);
    }
    const size_t bytes = (size_t)8 << 30, n = bytes / 16;
    uint4 *a, *b;
    hipMalloc(&a, bytes);
    hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes);
    hipMemset(b, 2, bytes);
    for (int blocks : {2048, 8192}) {
        const float tr = time_us([&] { hipLaunchKernelGGL(read_kernel, dim3(blocks), dim3(256), 0, 0, a, b, n); }, 5);
        const float tw = time_us([&] { hipLaunchKernelGGL(write_kernel, dim3(blocks), dim3(256), 0, 0, b, n); }, 5);
        const float tc = time_us([&] { hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, 0, a, b, n); }, 5);
        printf("HBM over 8 GB, %d workgroups: read %.2f TB/s, write %.2f TB/s, copy %.2f TB/s (read + write bytes)\n", blocks,
               bytes / tr / 1e6, bytes / tw / 1e6, 2.0 * bytes / tc / 1e6);
    }
    return 0;
}
