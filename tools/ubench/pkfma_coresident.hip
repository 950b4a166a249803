// Does packed fp32 math go wrong when its wave shares a SIMD with another kernel's waves?  (csrc/gpt_token.hip: the persistent decode
// kernel gave wrong results next to the split-precision conv kernels, and only when built WITH v_pk_fma_f32.)
//   hipcc --offload-arch=gfx950 -O3 -o bin/pkfma_coresident pkfma_coresident.hip
// Victim: `wgs` workgroups of 256 threads; every thread runs the column-GEMV inner loop of gpt_token.hip on constant data (x pairs from
// LDS, a weight float4 from registers, 16 accumulators: (row, row + 1) pairs times a broadcast weight = v_pk_fma_f32 op_sel_hi:[1,0,1] /
// op_sel:[0,1,0]) `iters` times and compares with the value the SAME thread computed in its first pass.  Run it alone, then next to a
// load (e.g. `python tools/bench_layer.py` in another process), with `lds_kb` small enough to share CUs and large (160) to own them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int KP = 768;
template <bool PACKED>
__global__ __launch_bounds__(256) void victim(int iters, unsigned* bad, float* sink) {
    extern __shared__ float4 xs[];                 // [2][KP]
    const int tid = threadIdx.x;
    for (int k = tid; k < 2 * KP; k += 256) {
        const float f = 0.001f * (float)((k * 37 + 11) % 1000) - 0.5f;
        xs[k] = make_float4(f, f * 0.5f + 0.1f, -f, f * 0.25f - 0.2f);
    }
    __syncthreads();
    float4 w[14];
    for (int i = 0; i < 14; ++i) {
        const float g = 0.01f * (float)((tid * 13 + i * 7) % 97) - 0.4f;
        w[i] = make_float4(g, -g * 0.5f, g * 0.3f + 0.05f, 0.7f * g - 0.1f);
    }
    const int kl = tid / 9;
    float ref[16];
    unsigned nbad = 0;
    for (int it = 0; it <= iters; ++it) {
        float acc[8][2];
        if (PACKED) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 av[4][2];                       // (row 2 p, row 2 p + 1) x column c
#pragma unroll
            for (int p = 0; p < 4; ++p) av[p][0] = av[p][1] = f2{0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 14; ++i) {
                const int k0 = kl + 28 * 2 * i, k1 = k0 + 28;
                const float4 xa = xs[k0], xb = xs[KP + k0], ya = xs[k1], yb = xs[KP + k1];
                const f2 xp[4] = {f2{xa.x, xa.y}, f2{xa.z, xa.w}, f2{xb.x, xb.y}, f2{xb.z, xb.w}};
                const f2 yp[4] = {f2{ya.x, ya.y}, f2{ya.z, ya.w}, f2{yb.x, yb.y}, f2{yb.z, yb.w}};
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    av[p][0] = __builtin_elementwise_fma(xp[p], f2{w[i].x, w[i].x}, av[p][0]);      // v_pk_fma_f32 ... op_sel_hi:[1,0,1]
                    av[p][1] = __builtin_elementwise_fma(xp[p], f2{w[i].y, w[i].y}, av[p][1]);      // v_pk_fma_f32 ... op_sel:[0,1,0]
                    av[p][0] = __builtin_elementwise_fma(yp[p], f2{w[i].z, w[i].z}, av[p][0]);
                    av[p][1] = __builtin_elementwise_fma(yp[p], f2{w[i].w, w[i].w}, av[p][1]);
                }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    acc[2 * p][c] = av[p][c].x;
                    acc[2 * p + 1][c] = av[p][c].y;
                }
        } else {
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[b][0] = acc[b][1] = 0.f;
#pragma unroll
            for (int i = 0; i < 14; ++i) {
                const int k0 = kl + 28 * 2 * i, k1 = k0 + 28;
                const float4 xa = xs[k0], xb = xs[KP + k0], ya = xs[k1], yb = xs[KP + k1];
                const float x[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
                const float y[8] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w};
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    acc[b][0] = __builtin_fmaf(x[b], w[i].x, acc[b][0]);
                    acc[b][1] = __builtin_fmaf(x[b], w[i].y, acc[b][1]);
                    acc[b][0] = __builtin_fmaf(y[b], w[i].z, acc[b][0]);
                    acc[b][1] = __builtin_fmaf(y[b], w[i].w, acc[b][1]);
                }
            }
        }
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (it == 0) ref[b * 2 + c] = acc[b][c];
                else if (acc[b][c] != ref[b * 2 + c]) ++nbad;
            }
        asm volatile("" ::: "memory");
    }
    if (nbad) atomicAdd(bad, nbad);
    if (ref[0] == 12345.f) sink[0] = ref[1];
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 128, lds_kb = argc > 2 ? atoi(argv[2]) : 56, iters = argc > 3 ? atoi(argv[3]) : 20000, rounds = argc > 4 ? atoi(argv[4]) : 20;
    unsigned* bad; float* sink;
    (void)hipMalloc(&bad, 4); (void)hipMalloc(&sink, 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(victim<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(victim<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int packed = 1; packed >= 0; --packed) {
        unsigned total = 0;
        for (int r = 0; r < rounds; ++r) {
            (void)hipMemset(bad, 0, 4);
            if (packed) hipLaunchKernelGGL(victim<true>, dim3(wgs), dim3(256), lds_kb * 1024, 0, iters, bad, sink);
            else hipLaunchKernelGGL(victim<false>, dim3(wgs), dim3(256), lds_kb * 1024, 0, iters, bad, sink);
            unsigned h = 0;
            (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
            total += h;
        }
        printf("%s math, %d workgroups x %d KB LDS, %d x %d passes: %u mismatching accumulators\n", packed ? "packed" : "scalar", wgs, lds_kb, rounds, iters, total);
    }
    return 0;
}
