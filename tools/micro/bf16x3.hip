// Microbenchmark for the next design step: an fp32-accurate GEMM inner product on the bf16 MFMA.
//   x = hi + lo with hi = bf16(x), lo = bf16(x - hi): 16 significant bits;  a * b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi, fp32 accumulate.
// Part 1 (accuracy): C[32][32] = A[32][K] * B[K][32], K = 2304 (256 channels x 9 taps), one wave: fp32 MFMA, bf16 x1, bf16 x3 and
//   bf16 x6 (all products of three 8-bit pieces down to 2^-24) against an fp64 host reference, on unit-variance operands and on
//   operands with a 1e-3 .. 1e3 dynamic range.
// Part 2 (rate): independent MFMA streams, v_mfma_f32_32x32x2_f32 against v_mfma_f32_32x32x16_bf16 (x3 per K = 16), 1-3 waves / SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/micro/bf16x3.hip -o build/micro/bf16x3
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ __bf16 to_bf16(float x) { return (__bf16)x; }      // round to nearest even

// mode 0: fp32 MFMA; 1: bf16 hi only; 3: three products; 6: six products (hi, mid, lo pieces)
template <int MODE>
__global__ void gemm_wave(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int K) {
    const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (MODE == 0) {
        for (int k = 0; k < K; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[li * K + k + lh], B[(k + lh) * 32 + li], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            bf16x8 a[3], b[3];
            for (int j = 0; j < 8; ++j) {
                const float av = A[li * K + k + 8 * lh + j], bv = B[(k + 8 * lh + j) * 32 + li];
                const __bf16 a0 = to_bf16(av), b0 = to_bf16(bv);
                const float ar = av - (float)a0, br = bv - (float)b0;
                const __bf16 a1 = to_bf16(ar), b1 = to_bf16(br);
                a[0][j] = a0; a[1][j] = a1; a[2][j] = to_bf16(ar - (float)a1);
                b[0][j] = b0; b[1][j] = b1; b[2][j] = to_bf16(br - (float)b1);
            }
            if (MODE == 6) {        // smallest terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
            }
            if (MODE >= 3) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
        }
    }
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = acc[r];
}

template <int BF>
__global__ __launch_bounds__(256) void rate(float* out, int iters) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 ab, bb;
    for (int j = 0; j < 8; ++j) { ab[j] = (__bf16)(0.5f + threadIdx.x * 1e-3f); bb[j] = (__bf16)0.25f; }
    const float af = 0.5f + threadIdx.x * 1e-3f, bf = 0.25f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (BF) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[a], 0, 0, 0);
            else acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[a], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) s += acc[a][0] + acc[a][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// Part 3: the same MFMA stream with DIFFERENT random operands on every instruction (eight register sets in rotation, unit-variance values
// as half): what the 16-bit pipe sustains on real data -- the clock under the power limit, not the issue rate, sets it.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void rate_random(const float* __restrict__ src, float* out, int iters) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    f16x8 av[8], bv[8];
    for (int s = 0; s < 8; ++s)
        for (int j = 0; j < 8; ++j) {
            av[s][j] = (_Float16)src[((s * 8 + j) * 256 + threadIdx.x) % 16384];
            bv[s][j] = (_Float16)src[((s * 8 + j) * 256 + threadIdx.x + 8192) % 16384];
        }
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[h * 4 + a], bv[(h * 4 + a + 3) & 7], acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) s += acc[a][0] + acc[a][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static double urand() { return (rand() + 0.5) / (RAND_MAX + 1.0); }
static double nrand() { return sqrt(-2.0 * log(urand())) * cos(6.283185307179586 * urand()); }

int main() {
    const int K = 2304;
    for (int wide = 0; wide < 2; ++wide) {
        std::vector<float> A(32 * K), B(K * 32), C(32 * 32);
        srand(7 + wide);
        for (auto& v : A) v = (float)(nrand() * (wide ? pow(10.0, 6.0 * urand() - 3.0) : 1.0));
        for (auto& v : B) v = (float)(nrand() / sqrt((double)K) * (wide ? pow(10.0, 2.0 * urand() - 1.0) : 1.0));
        std::vector<double> ref(32 * 32, 0.0), mag(32 * 32, 0.0);
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j)
                for (int k = 0; k < K; ++k) {
                    ref[i * 32 + j] += (double)A[i * K + k] * (double)B[k * 32 + j];
                    mag[i * 32 + j] += fabs((double)A[i * K + k] * (double)B[k * 32 + j]);
                }
        float *dA, *dB, *dC;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        printf("%s operands, K = %d: error relative to sum |a b| (max, rms)\n", wide ? "wide-range (1e-3..1e3)" : "unit-variance", K);
        const char* names[4] = {"fp32 MFMA 32x32x2", "bf16 x1", "bf16 x3", "bf16 x6"};
        for (int m = 0; m < 4; ++m) {
            if (m == 0) hipLaunchKernelGGL(gemm_wave<0>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
            if (m == 1) hipLaunchKernelGGL(gemm_wave<1>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
            if (m == 2) hipLaunchKernelGGL(gemm_wave<3>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
            if (m == 3) hipLaunchKernelGGL(gemm_wave<6>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
            hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
            double mx = 0, rms = 0, mxv = 0;
            for (int i = 0; i < 1024; ++i) {
                const double e = fabs(C[i] - ref[i]) / mag[i];
                mx = e > mx ? e : mx;
                rms += e * e;
                mxv = fabs(ref[i]) > mxv ? fabs(ref[i]) : mxv;
            }
            double mxo = 0;
            for (int i = 0; i < 1024; ++i) mxo = fmax(mxo, fabs(C[i] - ref[i]) / mxv);
            printf("  %-18s %.2e  %.2e   (max error / max |c|: %.2e)\n", names[m], mx, sqrt(rms / 1024), mxo);
        }
        hipFree(dA); hipFree(dB); hipFree(dC);
    }
    float* out;
    hipMalloc(&out, 3 * 256 * 256 * 4);
    const int iters = 20000;
    for (int bf = 0; bf < 2; ++bf)
        for (int w = 1; w <= 3; ++w) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (bf) hipLaunchKernelGGL(rate<1>, dim3(256 * w), dim3(256), 0, 0, out, iters);
                else hipLaunchKernelGGL(rate<0>, dim3(256 * w), dim3(256), 0, 0, out, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double n = (double)iters * 4 * w * 1024;                       // MFMAs in the machine (w waves on each of 1024 SIMDs)
            const double flop = n * 32.0 * 32.0 * 2.0 * (bf ? 16 : 2);
            printf("%s  %d wave(s)/SIMD: %7.2f ms  %6.1f ns per MFMA per SIMD  %7.1f TFLOP/s%s\n", bf ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_32x32x2_f32  ",
                   w, ms, ms * 1e6 / (iters * 4.0 * w), flop / ms / 1e9, bf ? "  (/3 for the three-product fp32 emulation)" : "");
        }
    {
        std::vector<float> h(16384);
        for (auto& v : h) v = (float)nrand();
        float* src;
        hipMalloc(&src, h.size() * 4);
        hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int w = 1; w <= 2; ++w)
            for (int len = 0; len < 2; ++len) {                     // a short launch (the clock has not settled) and a long one
                const int it2 = len ? 8 * iters : iters;
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(rate_random, dim3(256 * w), dim3(256), 0, 0, src, out, it2);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                }
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double flop = (double)it2 * 4 * w * 1024 * 32.0 * 32.0 * 2.0 * 16;
                printf("v_mfma_f32_32x32x16_f16, random operands  %d wave(s)/SIMD, %6d MFMAs per wave: %7.2f ms  %7.1f TFLOP/s\n", w, it2 * 4, ms, flop / ms / 1e9);
            }
    }
    return 0;
}
