// Microbenchmark: does fp32 VALU work overlap fp32 MFMA work on a gfx950 SIMD?
// A loop of 4 v_mfma_f32_32x32x2_f32 (4 independent accumulators) with NV independent fp32 vector ops per MFMA, scalar
// (v_fma_f32) or packed (v_pk_fma_f32), at 1 / 2 / 3 waves per SIMD.  Reported: time per MFMA in cycles (at the measured rate
// of the MFMA-only loop = 64 cycles) -- if VALU work hides in the MFMA's shadow the number stays at 64 until the issue port fills.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int PK>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float af = 1.f + threadIdx.x * 1e-3f, bf = 0.5f;
    f32x2 x[8];
    for (int i = 0; i < 8; ++i) x[i] = f32x2{seed + i, seed - i};
    f32x2 m = {1.0001f, 0.9999f}, c = {1e-7f, -1e-7f};
    asm volatile("" : "+v"(m), "+v"(c));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[s], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (PK) {
                    x[v & 7] = __builtin_elementwise_fma(x[v & 7], m, c);
                    asm volatile("" : "+v"(x[v & 7]));            // keep it a v_pk_fma_f32 (see the ISA dump)
                } else {
                    float t = x[v & 7][0];
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(t) : "v"(m[0]), "v"(c[0]));
                    x[v & 7][0] = t;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 8; ++i) s += x[i][0] + x[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, int PK>
void run(int blocks_per_cu) {
    float* out;
    const int grid = 256 * blocks_per_cu, iters = 20000;
    hipMalloc(&out, grid * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, PK>), dim3(grid), dim3(256), 0, 0, out, 2000, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, PK>), dim3(grid), dim3(256), 0, 0, out, iters, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)blocks_per_cu * iters * 4;       // one wave of each block per SIMD
    printf("NV=%2d %s waves/SIMD %d : %8.2f ms  %7.1f ns per MFMA per SIMD  (%6.1f TFLOP/s MFMA)\n", NV, PK ? "pk " : "f32", blocks_per_cu, ms,
           ms * 1e6 / mfma_per_simd, mfma_per_simd * 1024 * 4096.0 / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int b = 1; b <= 3; ++b) run<0, 0>(b);
    for (int b = 1; b <= 3; ++b) run<4, 0>(b);
    for (int b = 1; b <= 3; ++b) run<8, 0>(b);
    for (int b = 1; b <= 3; ++b) run<12, 0>(b);
    for (int b = 1; b <= 3; ++b) run<16, 0>(b);
    for (int b = 1; b <= 3; ++b) run<4, 1>(b);
    for (int b = 1; b <= 3; ++b) run<8, 1>(b);
    return 0;
}
