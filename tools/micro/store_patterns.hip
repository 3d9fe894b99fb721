// Microbenchmark: how fast does a tile-shaped bf16 output leave the chip, by store pattern?
//   hipcc --offload-arch=gfx950 -O3 -o store_patterns store_patterns.hip && ./store_patterns
// Output [M][N] bf16 written as 128 x 128 tiles by 512-thread workgroups (2 per CU, persistent over tiles), 16 B per lane:
//   pattern 0: one store instruction = 16 rows x 64 B   (the MFMA accumulator layout of gemm8: 4 lanes per row)
//   pattern 1: one store instruction =  8 rows x 128 B  (8 lanes per row: whole cache lines)
//   pattern 2: one store instruction =  4 rows x 256 B
//   variants: plain / nontemporal / sc1 (write-through) stores
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int PAT, int MODE>
__global__ __launch_bounds__(512) void k(unsigned short* C, int M, int N, int tiles_m, int tiles_n) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)C, 0, (int)((long)M * N * 2), 0x00020000);
    for (int t = blockIdx.x; t < tiles_m * tiles_n; t += gridDim.x) {
        const int m0 = (t / tiles_n) * 128, n0 = (t % tiles_n) * 128;
        // every wave writes 2 KB x 2 = 4 x 16 B per lane; wave (wm 0..1, wn 0..3) owns rows 64 wm .., columns 32 wn ..
        const int wm = wave >> 2, wn = wave & 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r, cbyte;
            if (PAT == 0) { r = 64 * wm + 16 * i + (lane & 15); cbyte = 64 * wn + 16 * (lane >> 4); }
            else if (PAT == 1) { r = 16 * wave + 4 * i + (lane >> 4) * 1 + 0; r = 16 * wave + 2 * i * 1 + 0; r = wave * 16 + i * 4 + (lane >> 4); cbyte = 16 * (lane & 15); }
            else { r = wave * 16 + i * 4 + (lane >> 4); cbyte = 16 * (lane & 15); }
            if (PAT == 1) { r = wave * 16 + i * 4 + (lane >> 4); cbyte = 0; /* placeholder, fixed below */ }
            // PAT 1: 8 lanes per row (128 B), 8 rows per instruction; the tile is 256 B wide, so two column halves
            if (PAT == 1) { const int half = i & 1, rr = (i >> 1) * 8 + (lane >> 3); r = wave * 16 + rr; cbyte = 128 * half + 16 * (lane & 7); }
            const u32x4 v = {(unsigned)t, (unsigned)r, (unsigned)lane, 7u};
            const unsigned off = (unsigned)(((long)(m0 + r) * N) * 2 + n0 * 2 + cbyte);
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, MODE == 1 ? 2 : (MODE == 2 ? 17 : 0));   // aux: 2 = nt, 17 = sc0|sc1
        }
    }
}

template <int PAT, int MODE>
float run(unsigned short* C, int M, int N, int grid) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int tm = M / 128, tn = N / 128;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<PAT, MODE>), dim3(grid), dim3(512), 0, 0, C, M, N, tm, tn);
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<PAT, MODE>), dim3(grid), dim3(512), 0, 0, C, M, N, tm, tn);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / 20;
}

int main() {
    const int M = 15488 / 128 * 128, N = 1280;
    unsigned short* C;
    hipMalloc(&C, (size_t)M * N * 2);
    const double mb = (double)M * N * 2 / 1e6;
    for (int grid : {256, 512, 1024}) {
        printf("grid %4d (%.1f MB): ", grid, mb);
        printf("16x64B plain %.1f nt %.1f sc1 %.1f | ", run<0, 0>(C, M, N, grid), run<0, 1>(C, M, N, grid), run<0, 2>(C, M, N, grid));
        printf("8x128B plain %.1f nt %.1f sc1 %.1f | ", run<1, 0>(C, M, N, grid), run<1, 1>(C, M, N, grid), run<1, 2>(C, M, N, grid));
        printf("4x256B plain %.1f nt %.1f sc1 %.1f us\n", run<2, 0>(C, M, N, grid), run<2, 1>(C, M, N, grid), run<2, 2>(C, M, N, grid));
    }
    return 0;
}
