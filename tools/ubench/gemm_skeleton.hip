// Skeleton of the conv GEMM K-loop, features switched on one at a time, to find where MFMA utilisation is lost.
// 256 threads = 4 waves (2x2), wave tile 64x64 (2x2 MFMA 32x32x2), BK=16, LDS tiles Ws[2][16][128], Xs[2][16][130].
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int BM = 128, BK = 16, XWP = 130;

template <int F>   // bit0: barrier per k-step, bit1: LDS-DMA W loads, bit2: register-staged X loads+stores, bit3: epilogue store
__global__ __launch_bounds__(256) void skel(const float* __restrict__ W, const float* __restrict__ X, float* __restrict__ Y, int ksteps, int T) {
    extern __shared__ float smem[];
    float* Ws = smem;
    float* Xs = smem + 2 * BK * BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    for (int i = tid; i < 2 * BK * BM; i += 256) Ws[i] = i * 1e-5f;
    for (int i = tid; i < 2 * BK * XWP; i += 256) Xs[i] = 1.f + i * 1e-6f;
    __syncthreads();
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    f16v acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * 128;
    float xreg[8];
    for (int s = 0; s < ksteps; ++s) {
        if (F & 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = tid + i * 256, row = idx / 32, c4 = idx - row * 32;
                const float* g = W + ((long long)(s * BK + row)) * 768 + m0 + c4 * 4;
                float* l = Ws + ((s + 1) & 1) * BK * BM + (wave * 64 + i * 256) * 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
            }
        }
        if (F & 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) xreg[r * 2 + c] = X[((long long)(s * BK + wave * 4 + r)) * T + n0 + lane + 64 * c];
        }
        const float* wq = Ws + (s & 1) * BK * BM + lhi * BM + wm0 + l31;
        const float* xq = Xs + (s & 1) * BK * XWP + lhi * XWP + wn0 + l31;
        float af[8][2], bf[8][2];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            af[kk][0] = wq[kk * 2 * BM]; af[kk][1] = wq[kk * 2 * BM + 32];
            bf[kk][0] = xq[kk * 2 * XWP]; bf[kk][1] = xq[kk * 2 * XWP + 32];
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][i], bf[kk][j], acc[i][j], 0, 0, 0);
        if (F & 4) {
            float* dst = Xs + ((s + 1) & 1) * BK * XWP + (wave * 4) * XWP + lane;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) dst[r * XWP + 64 * c] = xreg[r * 2 + c] * 1.0001f;
        }
        if (F & 1) __syncthreads();
    }
    if (F & 8) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi, n = n0 + wn0 + j * 32 + l31;
                    Y[((long long)blockIdx.z * 768 + row) * T + n] = acc[i][j][r];
                }
    } else {
        float sum = 0.f;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
        if (sum == 12345.f) Y[tid] = sum;
    }
}
template <int F>
void run(const float* W, const float* X, float* Y, const char* what) {
    const int ksteps = 48, T = 1024;
    dim3 grid(6, 8, 16);
    const size_t lds = sizeof(float) * (2 * BK * BM + 2 * BK * XWP);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(skel<F>, grid, dim3(256), lds, 0, W, X, Y, ksteps, T);
    (void)hipEventRecord(a);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(skel<F>, grid, dim3(256), lds, 0, W, X, Y, ksteps, T);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double fl = 2.0 * 768 * 768 * 16 * 1024 * reps;
    printf("%-58s %7.1f us/launch  %6.1f TFLOP/s\n", what, ms / reps * 1e3, fl / (ms * 1e-3) / 1e12);
}
int main() {
    float *W, *X, *Y;
    (void)hipMalloc(&W, 768 * 768 * 4); (void)hipMalloc(&X, (size_t)768 * 1024 * 16 * 4); (void)hipMalloc(&Y, (size_t)16 * 768 * 1024 * 4);
    (void)hipMemset(W, 0, 768 * 768 * 4); (void)hipMemset(X, 0, (size_t)768 * 1024 * 16 * 4);
    run<0>(W, X, Y, "loop only (LDS reads + MFMA)");
    run<1>(W, X, Y, "+ barrier per k-step");
    run<3>(W, X, Y, "+ barrier + LDS-DMA W tile");
    run<5>(W, X, Y, "+ barrier + register-staged X tile");
    run<7>(W, X, Y, "+ barrier + W DMA + X staged");
    run<15>(W, X, Y, "+ all + epilogue store");
    return 0;
}
