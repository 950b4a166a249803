// Dense 16-bit MFMA peak on gfx950: v_mfma_f32_32x32x16_{f16,bf16} and v_mfma_f32_16x16x32_f16, register operands only (no memory, no
// LDS), waves-per-SIMD sweep, operands ZERO vs RANDOM (the matrix pipe's power draw, hence the sustained clock, depends on the data),
// and long (50 - 200 ms) vs short (0.1 - 0.4 ms) runs (a short burst finishes before the clock settles).  The number the conv / attention
// rooflines should be read against is the random-operand sustained rate, not the 2.5 PFLOP/s nominal figure.
//   hipcc --offload-arch=gfx950 -O3 -o bin/mfma_peak16 mfma_peak16.hip && bin/mfma_peak16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef _Float16 hf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

// KIND 0: 32x32x16 f16, 1: 32x32x16 bf16, 2: 16x16x32 f16.  NACC independent accumulators, A/B fragments loaded once from memory.
template <int KIND, int NACC>
__global__ __launch_bounds__(256) void peak(const uint4* __restrict__ ops, float* out, int iters) {
    const uint4 ua = ops[threadIdx.x], ub = ops[256 + threadIdx.x];
    f16v acc[NACC];
    f4v acc4[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf8, ua), __builtin_bit_cast(hf8, ub), acc[i], 0, 0, 0);
            else if (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, ua), __builtin_bit_cast(bf8, ub), acc[i], 0, 0, 0);
            else acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(hf8, ua), __builtin_bit_cast(hf8, ub), acc4[i], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc4[i][r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// Do the matrix and the vector pipe overlap, or does the chip run at constant power?  Workgroups of 512 threads: waves 0-3 (one per SIMD)
// run the 32x32x16 f16 MFMA loop, waves 4-7 (MODE 1) a loop of independent v_fma_f32 (MODE 2: v_exp_f32), MODE 0: they exit at once.
template <int MODE>
__global__ __launch_bounds__(512) void mixed(const uint4* __restrict__ ops, float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    const uint4 ua = ops[threadIdx.x & 255], ub = ops[256 + (threadIdx.x & 255)];
    if (wave < 4) {
        f16v acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf8, ua), __builtin_bit_cast(hf8, ub), acc[i], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][r];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else if (MODE) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_bit_cast(float, ua.x) * (i + 1) * 1e-3f;
        const float c = 1.0001f, d = 1e-7f;
        // 4 MFMAs of 8 passes = 128 cycles per outer iteration of the MFMA waves; 32 VALU ops (4 cycles each) fill the same time
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < (MODE == 3 ? 1 : 2); ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = MODE == 2 ? __builtin_amdgcn_exp2f(v[i]) : __builtin_fmaf(v[i], c, d);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += v[i];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
}

template <int MODE>
void run_mixed(const uint4* ops, float* out, const char* name) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int iters = 400000;
    hipLaunchKernelGGL(mixed<MODE>, dim3(256), dim3(512), 0, 0, ops, out, 200);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(mixed<MODE>, dim3(256), dim3(512), 0, 0, ops, out, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double fl = 2.0 * 32 * 32 * 16 * 4 * (double)iters * 256 * 4;
    printf("mixed: 1 MFMA wave/SIMD + %-28s %7.1f ms   MFMA rate %7.1f TFLOP/s\n", name, ms, fl / (ms * 1e-3) / 1e12);
}

template <int KIND, int NACC>
void sweep(const char* name, const uint4* ops, float* out, const char* data) {
    const double flop_per = KIND == 2 ? 2.0 * 16 * 16 * 32 : 2.0 * 32 * 32 * 16;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = 256 * wps;            // 256-thread blocks = one wave per SIMD each
        for (int iters : {2000, 1000000}) {
            hipLaunchKernelGGL((peak<KIND, NACC>), dim3(blocks), dim3(256), 0, 0, ops, out, 200);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(a);
            hipLaunchKernelGGL((peak<KIND, NACC>), dim3(blocks), dim3(256), 0, 0, ops, out, iters);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            const double fl = flop_per * NACC * (double)iters * blocks * 4;
            printf("%-18s %-7s %d acc  %d waves/SIMD  %7.1f ms  %7.1f TFLOP/s  (%.3f of 2500)\n", name, data, NACC, wps, ms, fl / (ms * 1e-3) / 1e12,
                   fl / (ms * 1e-3) / 1e12 / 2500.0);
        }
    }
}

int main() {
    uint4* ops; float* out;
    (void)hipMalloc(&ops, 512 * 16); (void)hipMalloc(&out, 256 * 4 * 1024 * 4 * 2);
    for (int pass = 0; pass < 2; ++pass) {
        std::vector<unsigned short> h(512 * 8);
        srand(3);
        for (auto& v : h) {
            if (!pass) { v = 0; continue; }
            // random fp16 / bf16 bit patterns of moderate magnitude: sign + exponent near 1 + random mantissa (valid in both formats)
            v = (unsigned short)(((rand() & 1) << 15) | (0x3800 + ((rand() & 3) << 8)) | (rand() & 0xff));
        }
        (void)hipMemcpy(ops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        const char* data = pass ? "random" : "zeros";
        sweep<0, 4>("32x32x16 f16", ops, out, data);
        sweep<1, 4>("32x32x16 bf16", ops, out, data);
        sweep<2, 8>("16x16x32 f16", ops, out, data);
        printf("operands: %s\n", data);
        run_mixed<0>(ops, out, "nothing");
        run_mixed<1>(ops, out, "1 v_fma_f32 wave/SIMD");
        run_mixed<3>(ops, out, "1 v_fma_f32 wave/SIMD, half");
        run_mixed<2>(ops, out, "1 v_exp_f32 wave/SIMD");
    }
    return 0;
}
