// Calibration: throughput of device-scope f32 atomicAdd on MI355X as a function of the address footprint, against plain
// stores.  Build: hipcc --offload-arch=gfx950 -O3 tools/atomic_bench.hip -o tools/bin/atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void atom_kernel(float* buf, long span, int per_thread) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = 0; k < per_thread; ++k) atomicAdd(buf + (tid + (long)k * gridDim.x * blockDim.x) % span, 1.0f);
}
__global__ void atom_wg_kernel(float* buf, int C) {          // the backward-kernel pattern: every workgroup adds C values
    for (int i = threadIdx.x; i < C; i += blockDim.x) atomicAdd(buf + i, 1.0f);
}
__global__ void store_kernel(float* buf, long span, int per_thread) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = 0; k < per_thread; ++k) buf[(tid + (long)k * gridDim.x * blockDim.x) % span] = 1.0f;
}

template <typename F>
float timeit(F f) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / 20 * 1e3f;
}

int main() {
    float* buf;
    hipMalloc(&buf, 64 << 20);
    hipMemset(buf, 0, 64 << 20);
    const int blocks = 1024, per = 4;
    const long total = (long)blocks * 256 * per;
    for (long span : {640L, 4096L, 65536L, 1048576L, 16777216L}) {
        float ta = timeit([&] { hipLaunchKernelGGL(atom_kernel, dim3(blocks), dim3(256), 0, 0, buf, span, per); });
        float ts = timeit([&] { hipLaunchKernelGGL(store_kernel, dim3(blocks), dim3(256), 0, 0, buf, span, per); });
        printf("span %9ld floats: %ld atomics %.1f us (%.1f /ns) | stores %.1f us\n", span, total, ta, total / ta / 1e3, ts);
    }
    for (int wgs : {256, 1024, 4096})
        for (int C : {128, 640, 1024}) {
            float t = timeit([&] { hipLaunchKernelGGL(atom_wg_kernel, dim3(wgs), dim3(256), 0, 0, buf, C); });
            printf("%d workgroups x %d channel atomics: %.1f us (%.1f /ns)\n", wgs, C, t, (double)wgs * C / t / 1e3);
        }
    return 0;
}
