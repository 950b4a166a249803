// fp32 MFMA peak micro-benchmark: waves-per-SIMD sweep, 32x32x2 and 16x16x4, to calibrate the achievable roof.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k32(float* out, int iters) {
    f16v acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void k16(float* out, int iters) {
    f4v acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
double run(F launch, double flops) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return flops / (ms * 1e-3) / 1e12;
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 1024 * 4);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = 256 * wps;   // 256-thread blocks: 1 wave per SIMD each
        double f32 = 2.0 * 32 * 32 * 2 * 4 * (double)iters * blocks * 4;
        printf("32x32x2  4acc  %d waves/SIMD: %.1f TFLOP/s\n", wps, run([&] { hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(256), 0, 0, out, iters); }, f32));
        double f16 = 2.0 * 16 * 16 * 4 * 8 * (double)iters * blocks * 4;
        printf("16x16x4  8acc  %d waves/SIMD: %.1f TFLOP/s\n", wps, run([&] { hipLaunchKernelGGL(k16<8>, dim3(blocks), dim3(256), 0, 0, out, iters); }, f16));
    }
    double f1 = 2.0 * 32 * 32 * 2 * 1 * (double)iters * 256 * 4;
    printf("32x32x2  1acc  1 wave/SIMD (dependent chain): %.1f TFLOP/s\n", run([&] { hipLaunchKernelGGL(k32<1>, dim3(256), dim3(256), 0, 0, out, iters); }, f1));
    return 0;
}
