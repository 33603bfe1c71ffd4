// Microbenchmark: what rate does v_mfma_f32_32x32x2_f32 sustain per SIMD with 1/2/3 waves per SIMD,
// with and without LDS fragment reads in the loop?  (ceiling check for conv_igemm)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int LDS, int SHAPE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float sm[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = (float)(i & 7) * 0.001f;
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    f32x4 af[2], bf[2];
    af[0] = af[1] = bf[0] = bf[1] = f32x4{1.f, 0.5f, 0.25f, 0.125f};
    const float* base = sm + (threadIdx.x & 63) * 20;
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
            af[0] = *(const f32x4*)(base + ((it * 8) & 1023));
            af[1] = *(const f32x4*)(base + 2560 + ((it * 8) & 1023));
            bf[0] = *(const f32x4*)(base + 4096 + ((it * 8) & 1023));
            bf[1] = *(const f32x4*)(base + 5120 + ((it * 8) & 1023));
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (SHAPE == 32) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][s], bf[0][s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][s], bf[1][s], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][s], bf[0][s], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][s], bf[1][s], acc[3], 0, 0, 0);
            } else {
                // 16 independent 16x16 accumulators live in the 4 f32x16 (4 regs each): same FLOPs per s
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    f32x4 c = {acc[t >> 2][(t & 3) * 4], acc[t >> 2][(t & 3) * 4 + 1], acc[t >> 2][(t & 3) * 4 + 2], acc[t >> 2][(t & 3) * 4 + 3]};
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t & 1][s], bf[(t >> 1) & 1][s], c, 0, 0, 0);
                    acc[t >> 2][(t & 3) * 4] = c[0]; acc[t >> 2][(t & 3) * 4 + 1] = c[1];
                    acc[t >> 2][(t & 3) * 4 + 2] = c[2]; acc[t >> 2][(t & 3) * 4 + 3] = c[3];
                }
            }
        }
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int LDS, int SHAPE>
void run(const char* name, int blocks_per_cu) {
    float* out;
    const int grid = 256 * blocks_per_cu, iters = 20000;
    hipMalloc(&out, grid * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<LDS, SHAPE>), dim3(grid), dim3(256), 0, 0, out, 2000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<LDS, SHAPE>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per iteration per wave: SHAPE32: 16 MFMAs x 4096 flop ; SHAPE16: 64 MFMAs x 2048 flop... = 65536 flop
    const double flops = (double)grid * 4 * iters * 65536.0 * (SHAPE == 32 ? 1.0 : 2.0);
    printf("%-28s blocks/CU %d : %8.2f ms  %7.1f TFLOP/s\n", name, blocks_per_cu, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int b = 1; b <= 3; ++b) run<0, 32>("32x32x2 regs only", b);
    for (int b = 1; b <= 3; ++b) run<1, 32>("32x32x2 + 4 ds_read_b128/16", b);
    for (int b = 1; b <= 3; ++b) run<0, 16>("16x16x4 regs only", b);
    for (int b = 1; b <= 3; ++b) run<1, 16>("16x16x4 + ds_read", b);
    return 0;
}
