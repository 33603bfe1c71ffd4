// Which 16-bit MFMA shape sustains more FLOP/s under the power limit on random operands?  MI355X_MICROARCH.md ("DVFS give-back", item 7)
// reports v_mfma_f32_16x16x32 at ~1.15x the FLOP/s of 32x32x16 at equal cycles per FLOP in bare loops; this measures it for IEEE-half
// operands in the shape the SPLIT conv kernel would use (a wave tile of 64 x 64 outputs: 2 x 2 accumulators of 32x32 or 4 x 4 of 16x16,
// different random operands on every instruction), long launches, the two shapes interleaved A/B/A/B.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shape.hip -o build/micro/mfma_shape
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(256) void rate(const float* __restrict__ src, float* out, int iters) {
    f16x8 av[8], bv[8];
    for (int s = 0; s < 8; ++s)
        for (int j = 0; j < 8; ++j) {
            av[s][j] = (_Float16)src[((s * 8 + j) * 256 + threadIdx.x) % 16384];
            bv[s][j] = (_Float16)src[((s * 8 + j) * 256 + threadIdx.x + 8192) % 16384];
        }
    float s = 0.f;
    if (SHAPE == 32) {          // 4 MFMAs of 32x32x16 = one K = 16 step of a 64 x 64 wave tile
        f32x16 acc[4];
        for (int a = 0; a < 4; ++a)
            for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        for (int it = 0; it < iters; it += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[h * 4 + a], bv[(h * 4 + a + 3) & 7], acc[a], 0, 0, 0);
        }
        for (int a = 0; a < 4; ++a) s += acc[a][0] + acc[a][15];
    } else {                    // 16 MFMAs of 16x16x32 = one K = 32 step of the same tile: the same FLOPs per `it` pair
        f32x4 acc[16];
        for (int a = 0; a < 16; ++a)
            for (int r = 0; r < 4; ++r) acc[a][r] = 0.f;
        for (int it = 0; it < iters; it += 2) {
#pragma unroll
            for (int a = 0; a < 16; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[a & 7], bv[(a + 3 + (a >> 3)) & 7], acc[a], 0, 0, 0);
        }
        for (int a = 0; a < 16; ++a) s += acc[a][0] + acc[a][3];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static double urand() { return (rand() + 0.5) / (RAND_MAX + 1.0); }
static double nrand() { return sqrt(-2.0 * log(urand())) * cos(6.283185307179586 * urand()); }

int main() {
    std::vector<float> h(16384);
    for (auto& v : h) v = (float)nrand();
    float *src, *out;
    hipMalloc(&src, h.size() * 4);
    hipMalloc(&out, 3 * 256 * 256 * 4);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int iters = 320000;
    for (int w = 1; w <= 2; ++w)
        for (int rep = 0; rep < 3; ++rep)
            for (int shape = 0; shape < 2; ++shape) {
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                if (shape == 0) hipLaunchKernelGGL(rate<32>, dim3(256 * w), dim3(256), 0, 0, src, out, iters);
                else hipLaunchKernelGGL(rate<16>, dim3(256 * w), dim3(256), 0, 0, src, out, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double flop = (double)iters * 4 * w * 1024 * 32.0 * 32.0 * 2.0 * 16;      // (both shapes: 64 x 64 x 16 MACs per it)
                printf("%s  random half operands  %d wave(s)/SIMD  rep %d: %7.2f ms  %7.1f TFLOP/s\n", shape ? "v_mfma_f32_16x16x32_f16" : "v_mfma_f32_32x32x16_f16", w, rep, ms,
                       flop / ms / 1e9);
            }
    return 0;
}
