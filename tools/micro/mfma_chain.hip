// Microbenchmark: v_mfma_f32_32x32x2_f32 throughput per SIMD as a function of the number of INDEPENDENT accumulators a wave
// cycles through (NACC = 1: every MFMA reads the previous one's result as SrcC) and of the waves per SIMD.  A 64x64 tile kernel
// (one 32x32 accumulator per wave) lives in the NACC = 1 column.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float af = 1.f + threadIdx.x * 1e-3f, bf = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            acc[s % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[s % NACC], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks_per_cu) {
    float* out;
    const int grid = 256 * blocks_per_cu, iters = 10000;
    hipMalloc(&out, grid * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC>), dim3(grid), dim3(256), 0, 0, out, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)blocks_per_cu * iters * 8;       // one wave of each block per SIMD
    printf("accumulators %d  waves/SIMD %d : %7.1f ns per MFMA per SIMD  (%6.1f TFLOP/s)\n", NACC, blocks_per_cu,
           ms * 1e6 / mfma_per_simd, mfma_per_simd * 1024 * 4096.0 / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int b = 1; b <= 4; ++b) run<1>(b);
    for (int b = 1; b <= 4; ++b) run<2>(b);
    for (int b = 1; b <= 4; ++b) run<4>(b);
    return 0;
}
