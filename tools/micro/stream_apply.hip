// Micro-benchmark: what a BatchNorm-apply-shaped streaming kernel (y = relu(x * s[c] + t[c]), fp32 NHWC, 16 bytes per lane) reaches on
// MI355X as a function of its launch shape -- grid size, loads in flight per thread, non-temporal stores.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/micro/stream_apply.hip -o /tmp/stream_apply && /tmp/stream_apply
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void apply_kernel(const float* __restrict__ x, const float* __restrict__ sc, const float* __restrict__ sh,
                                                    float* __restrict__ y, long long total4, int C) {
    const int cg = C / 4;
    const long long stride = (long long)gridDim.x * 256;
    const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c = (int)(i0 % cg) * 4;
    const f32x4 s = *reinterpret_cast<const f32x4*>(sc + c), t = *reinterpret_cast<const f32x4*>(sh + c);
    long long i = i0;
    for (; i + (UNROLL - 1) * stride < total4; i += UNROLL * stride) {
        f32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x) + i + u * stride) : reinterpret_cast<const f32x4*>(x)[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            f32x4 o = v[u] * s + t;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], 0.f);
            if (NT) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(y) + i + u * stride);
            else reinterpret_cast<f32x4*>(y)[i + u * stride] = o;
        }
    }
    for (; i < total4; i += stride) {
        f32x4 o = reinterpret_cast<const f32x4*>(x)[i] * s + t;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], 0.f);
        reinterpret_cast<f32x4*>(y)[i] = o;
    }
}

// contiguous-chunk form: a workgroup owns ONE contiguous range of the tensor (consecutive iterations touch consecutive 4 KiB)
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void apply_chunk_kernel(const float* __restrict__ x, const float* __restrict__ sc, const float* __restrict__ sh,
                                                          float* __restrict__ y, long long total4, int C) {
    const int cg = C / 4;
    const long long per = (total4 + gridDim.x - 1) / gridDim.x / 256 * 256 + 256;
    const long long b0 = (long long)blockIdx.x * per, b1 = b0 + per < total4 ? b0 + per : total4;
    for (long long i = b0 + threadIdx.x; i < b1; i += 256 * UNROLL) {
        f32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const long long k = i + 256 * u < b1 ? i + 256 * u : b1 - 1;
            v[u] = reinterpret_cast<const f32x4*>(x)[k];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const long long k = i + 256 * u;
            if (k >= b1) break;
            const int c = (int)(k % cg) * 4;
            f32x4 o = v[u] * *reinterpret_cast<const f32x4*>(sc + c) + *reinterpret_cast<const f32x4*>(sh + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], 0.f);
            if (NT) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(y) + k);
            else reinterpret_cast<f32x4*>(y)[k] = o;
        }
    }
}

template <typename F>
float time_us(F f, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}

int main() {
    const int shapes[][2] = {{200704, 256}, {200704, 64}, {50176, 512}, {12544, 1024}};
    for (auto& sh_ : shapes) {
        const long long M = sh_[0];
        const int C = sh_[1];
        const long long n = M * C, total4 = n / 4;
        float *x, *y, *s, *t, *big;
        hipMalloc(&x, n * 4);
        hipMalloc(&y, n * 4);
        hipMalloc(&s, C * 4);
        hipMalloc(&t, C * 4);
        hipMalloc(&big, 1ll << 30);
        hipMemset(x, 0, n * 4);
        hipMemset(s, 0, C * 4);
        hipMemset(t, 0, C * 4);
        printf("M=%lld C=%d (%.1f MB read + %.1f MB written)\n", M, C, n * 4 / 1e6, n * 4 / 1e6);
        for (int flush = 0; flush < 2; ++flush) {
            auto run = [&](const char* name, auto launch) {
                auto f = [&]() {
                    if (flush) hipMemsetAsync(big, 1, 1ll << 30, 0);      // push x / y out of the 256 MB MALL between launches
                    launch();
                };
                float us = time_us(f, 20);
                if (flush) {
                    auto g = [&]() { hipMemsetAsync(big, 1, 1ll << 30, 0); };
                    us -= time_us(g, 20);
                }
                printf("  %-34s %s %7.1f us  %5.2f TB/s\n", name, flush ? "cold" : "warm", us, 2.0 * n * 4 / us / 1e6);
            };
            for (int grid : {512, 1024, 2048, 4096, 8192}) {
                char nm[64];
                snprintf(nm, sizeof nm, "stride u4 grid %d", grid);
                run(nm, [&]() { hipLaunchKernelGGL((apply_kernel<4, false>), dim3(grid), dim3(256), 0, 0, x, s, t, y, total4, C); });
            }
            run("stride u8 grid 2048", [&]() { hipLaunchKernelGGL((apply_kernel<8, false>), dim3(2048), dim3(256), 0, 0, x, s, t, y, total4, C); });
            run("stride u2 grid 4096", [&]() { hipLaunchKernelGGL((apply_kernel<2, false>), dim3(4096), dim3(256), 0, 0, x, s, t, y, total4, C); });
            run("stride u4 grid 2048 nontemporal", [&]() { hipLaunchKernelGGL((apply_kernel<4, true>), dim3(2048), dim3(256), 0, 0, x, s, t, y, total4, C); });
            for (int grid : {1024, 2048, 4096}) {
                char nm[64];
                snprintf(nm, sizeof nm, "chunk u4 grid %d", grid);
                run(nm, [&]() { hipLaunchKernelGGL((apply_chunk_kernel<4, false>), dim3(grid), dim3(256), 0, 0, x, s, t, y, total4, C); });
            }
            run("chunk u4 grid 2048 nontemporal", [&]() { hipLaunchKernelGGL((apply_chunk_kernel<4, true>), dim3(2048), dim3(256), 0, 0, x, s, t, y, total4, C); });
            run("hipMemcpyDtoD", [&]() { hipMemcpyAsync(y, x, n * 4, hipMemcpyDeviceToDevice, 0); });
        }
        hipFree(x); hipFree(y); hipFree(s); hipFree(t); hipFree(big);
    }
    return 0;
}
