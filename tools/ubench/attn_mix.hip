// What bounds the split-precision attention loop?  One "step" of csrc/attention_x3b.hip is 21 MFMAs (32x32x16 f16: a 9-long dependent
// chain for QK^T, then 4 groups of 3 dependent ones alternating between two accumulators for PV), ~70 vector instructions (16 fma, 16
// exp, 8 max3, 24 cvt / mix, ...) and 14 ds_read_b128.  This benchmark runs that instruction mix from registers (no global memory, no
// barriers, LDS reads from a fixed region), with each ingredient switchable, at 1 .. 3 waves per SIMD on every CU:
//   chain:  0 = 21 independent accumulators-ish (7 accumulators round robin), 1 = the kernel's dependency structure
//   valu:   number of independent vector instructions per step (0 / 35 / 70 / 140), spread between the MFMAs
//   lds:    number of ds_read_b128 per step (0 / 14 / 28)
// Output: time per step per wave in SIMD cycles (at the measured clock) and the MFMA-pipe utilisation.
//   hipcc --offload-arch=gfx950 -O3 -o bin/attn_mix attn_mix.hip && bin/attn_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 hf8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int CHAIN, int VALU, int LDS>
__global__ __launch_bounds__(256) void step_kernel(const uint4* __restrict__ ops, float* out, int iters) {
    __shared__ uint4 lds[2048];                                   // 32 KiB
    const int tid = threadIdx.x;
    for (int i = tid; i < 2048; i += 256) lds[i] = ops[i & 511];
    __syncthreads();
    hf8 a = __builtin_bit_cast(hf8, ops[tid]), b = __builtin_bit_cast(hf8, ops[256 + tid]);
    f16v acc[7];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = __builtin_bit_cast(float, ops[tid].x) * (i + 1) * 1e-3f;
    const float c = 1.0001f, d = 1e-7f;
    unsigned laddr = (tid & 63) * 16;
    for (int it = 0; it < iters; ++it) {
        hf8 fa = a, fb = b;
        if (LDS) {
            // the first fragments of the step come from LDS: the MFMAs depend on them as in the kernel
            fa = __builtin_bit_cast(hf8, lds[(laddr >> 4) + (it & 1) * 64]);
            fb = __builtin_bit_cast(hf8, lds[(laddr >> 4) + 128 + (it & 1) * 64]);
#pragma unroll
            for (int l = 2; l < LDS; ++l) {
                const uint4 t = lds[(laddr >> 4) + 64 * (l & 15) + (it & 1) * 1024];
                v[l & 15] += __builtin_bit_cast(float, t.x);      // keep the read alive (one add per read)
            }
        }
#pragma unroll
        for (int m = 0; m < 21; ++m) {
            // dependency structure: CHAIN 1: MFMAs 0..8 on acc[0]; 9..11 acc[1]; 12..14 acc[2]; 15..17 acc[1]; 18..20 acc[2]
            const int k = CHAIN ? (m < 9 ? 0 : (((m - 9) / 3) & 1) + 1) : m % 7;
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[k], 0, 0, 0);
            // VALU / 21 vector instructions after each MFMA (independent of the MFMAs)
#pragma unroll
            for (int u = 0; u < (VALU * (m + 1)) / 21 - (VALU * m) / 21; ++u) {
                const int i = (m * 7 + u) & 15;
                v[i] = (u & 3) == 3 ? __builtin_amdgcn_exp2f(v[i]) : __builtin_fmaf(v[i], c, d);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * 256 + tid] = s;
}

template <int CHAIN, int VALU, int LDS>
static void run(const uint4* ops, float* out, int wps, double clk_ghz) {
    const int iters = 20000, ncu = 256;
    const dim3 grid(ncu * wps);                                   // 4 waves per workgroup = one per SIMD; wps workgroups per CU
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((step_kernel<CHAIN, VALU, LDS>), grid, dim3(256), 0, 0, ops, out, 200);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((step_kernel<CHAIN, VALU, LDS>), grid, dim3(256), 0, 0, ops, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double cyc_per_step_simd = ms * 1e-3 * clk_ghz * 1e9 / iters;             // SIMD cycles per (step of ALL its waves)
    const double mfma_cyc = 21.0 * 32.0 * wps;
    printf("chain %d  valu %3d  lds %2d  waves/SIMD %d : %7.1f ms  %8.1f cycles per step-round per SIMD (%6.1f per wave-step)  MFMA util %5.1f %%  %7.1f TFLOP/s\n",
           CHAIN, VALU, LDS, wps, ms, cyc_per_step_simd, cyc_per_step_simd / wps, 100.0 * mfma_cyc / cyc_per_step_simd,
           21.0 * 32768.0 * 4 * ncu * wps * iters / (ms * 1e-3) / 1e12);
}

int main() {
    std::vector<uint4> h(2048);
    srand(1);
    for (auto& q : h) {
        _Float16 t[8];
        for (int i = 0; i < 8; ++i) t[i] = (_Float16)((rand() % 2001 - 1000) * 1e-3f);
        memcpy(&q, t, 16);
    }
    uint4* ops;
    float* out;
    CHECK(hipMalloc(&ops, h.size() * 16));
    CHECK(hipMemcpy(ops, h.data(), h.size() * 16, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&out, 256 * 256 * 4 * sizeof(float) * 4));
    const double clk = 2.4;
    printf("attention step mix from registers (21 MFMA 32x32x16 f16 per step; cycles quoted at %.1f GHz)\n", clk);
    for (int wps = 1; wps <= 3; ++wps) {
        run<0, 0, 0>(ops, out, wps, clk);
        run<1, 0, 0>(ops, out, wps, clk);
        run<1, 35, 0>(ops, out, wps, clk);
        run<1, 70, 0>(ops, out, wps, clk);
        run<1, 140, 0>(ops, out, wps, clk);
        run<0, 70, 0>(ops, out, wps, clk);
        run<1, 0, 14>(ops, out, wps, clk);
        run<1, 70, 14>(ops, out, wps, clk);
        run<1, 70, 28>(ops, out, wps, clk);
        run<0, 70, 14>(ops, out, wps, clk);
    }
    return 0;
}
