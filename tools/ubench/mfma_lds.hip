// Inner-loop ceiling: per k-pair {A: 2 x ds_read_b32, B: 2 x ds_read_b32} -> 4 MFMA 32x32x2 (the conv GEMM's 64x64 wave tile),
// and the 16x16x4 variant {A: 4 reads, B: 4 reads} -> 16 MFMA (64x64 wave tile). Sweeps waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

__global__ void k32(float* out, int iters) {
    __shared__ float Ws[16 * 128], Xs[16 * 136];
    for (int i = threadIdx.x; i < 16 * 128; i += blockDim.x) { Ws[i] = i * 1e-4f; }
    for (int i = threadIdx.x; i < 16 * 136; i += blockDim.x) { Xs[i] = 1.f + i * 1e-5f; }
    __syncthreads();
    const int lane = threadIdx.x & 63, l31 = lane & 31, lhi = lane >> 5, wave = threadIdx.x >> 6;
    f16v acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float* wsb = Ws + (wave >> 1) * 64 + l31;
    const float* xsb = Xs + (wave & 1) * 64 + l31;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int krow = kk * 2 + lhi;
            float a0 = wsb[krow * 128], a1 = wsb[krow * 128 + 32], b0 = xsb[krow * 136], b1 = xsb[krow * 136 + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 16x16x4: wave tile 64x64 = 4x4 tiles; per 4-deep k-step: 4 A reads + 4 B reads, 16 MFMAs (32 cycles each)
__global__ void k16(float* out, int iters) {
    __shared__ float Ws[16 * 128], Xs[16 * 144];
    for (int i = threadIdx.x; i < 16 * 128; i += blockDim.x) { Ws[i] = i * 1e-4f; }
    for (int i = threadIdx.x; i < 16 * 144; i += blockDim.x) { Xs[i] = 1.f + i * 1e-5f; }
    __syncthreads();
    const int lane = threadIdx.x & 63, l15 = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    f4v acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    const float* wsb = Ws + (wave >> 1) * 64 + l15;
    const float* xsb = Xs + (wave & 1) * 64 + l15;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int krow = kk * 4 + g;
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = wsb[krow * 128 + i * 16]; b[i] = xsb[krow * 144 + i * 16]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
double run(F launch, double flops) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return flops / (ms * 1e-3) / 1e12;
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 8 * 1024 * 4);
    const int iters = 2000;
    for (int wps = 1; wps <= 4; ++wps) {
        const int blocks = 256 * wps;
        double fl = 2.0 * 64 * 64 * 16 * (double)iters * blocks * 4;
        printf("32x32x2 loop (2A+2B ds_read_b32 / 4 MFMA)  %d waves/SIMD: %.1f TFLOP/s\n", wps, run([&] { hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, out, iters); }, fl));
        printf("16x16x4 loop (4A+4B ds_read_b32 / 16 MFMA) %d waves/SIMD: %.1f TFLOP/s\n", wps, run([&] { hipLaunchKernelGGL(k16, dim3(blocks), dim3(256), 0, 0, out, iters); }, fl));
    }
    return 0;
}
