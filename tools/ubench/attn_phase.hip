// Does the vector work of the attention step hide under the matrix pipe?  Same per-step mix as csrc/attention_x3b.hip (21 MFMAs 32x32x16
// f16, ~72 vector instructions: 16 fma + 16 exp + 8 cvt_pk + 16 residual fma + 8 max3 + ..., TRUE dependencies), two schedules:
//   MODE 0 "phases"    : QK^T chain (9 MFMA) -> softmax of ITS result (72 VALU) -> PV (12 MFMA that consume the P planes): inside a wave the
//                        matrix and the vector work alternate; overlap can only come from the other waves of the SIMD
//   MODE 1 "pipelined" : step b issues QK^T(b + 1) and PV(b - 1) - 21 MFMAs that depend on nothing computed in this step - interleaved
//                        one MFMA : ~3.5 vector instructions (sched_group_barrier) with the softmax of block b
// at 1 .. 3 waves per SIMD on every CU, operands random.   hipcc --offload-arch=gfx950 -O3 -o bin/attn_phase attn_phase.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 hf8 __attribute__((ext_vector_type(8)));
typedef _Float16 hf2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// softmax-like vector work on 16 scores -> two P planes (2 x hf8 x 2 key steps); ~72 instructions
__device__ __forceinline__ void softmax16(const f16v& s, float c0, float& mx_out, hf8 (&pf)[2][2]) {
    float mx = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s[r]), s[r + 1]);
    mx_out = fmaxf(mx, s[15]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        unsigned w0[4], w1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float x = __builtin_amdgcn_exp2f(fmaf(s[8 * j + 2 * k], 0.00390625f, c0)), y = __builtin_amdgcn_exp2f(fmaf(s[8 * j + 2 * k + 1], 0.00390625f, c0));
            const f2 v = {x, y};
            const hf2 h = __builtin_convertvector(v, hf2);
            w0[k] = __builtin_bit_cast(unsigned, h);
            const f2 r = v - __builtin_convertvector(h, f2);
            w1[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, hf2));
        }
        pf[j][0] = __builtin_bit_cast(hf8, make_uint4(w0[0], w0[1], w0[2], w0[3]));
        pf[j][1] = __builtin_bit_cast(hf8, make_uint4(w1[0], w1[1], w1[2], w1[3]));
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ ops, float* out, int iters) {
    const int tid = threadIdx.x;
    const hf8 a0 = __builtin_bit_cast(hf8, ops[tid]), a1 = __builtin_bit_cast(hf8, ops[256 + tid]), q0 = __builtin_bit_cast(hf8, ops[512 + tid]),
              q1 = __builtin_bit_cast(hf8, ops[768 + tid]);
    f16v o0, o1, s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = o1[r] = s0[r] = s1[r] = 0.f;
    hf8 pf[2][2] = {{a0, a1}, {a1, a0}};
    float c0 = -1.f, mxa = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            f16v s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int m = 0; m < 9; ++m) s = __builtin_amdgcn_mfma_f32_32x32x16_f16((m % 3) ? a0 : a1, (m % 3 == 1) ? q1 : q0, s, 0, 0, 0);
            float mx;
            softmax16(s, c0, mx, pf);
            mxa += mx;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, pf[j][0], o0, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, pf[j][1], o0, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, pf[j][0], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, pf[j][0], o1, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, pf[j][1], o1, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, pf[j][0], o1, 0, 0, 0);
            }
        } else {
            // s0 holds block b (finished last step), pf the planes of block b - 1: QK(b + 1) -> s1, PV(b - 1), softmax(b) -> pfn
            hf8 pfn[2][2];
#pragma unroll
            for (int r = 0; r < 16; ++r) s1[r] = 0.f;
#pragma unroll
            for (int m = 0; m < 9; ++m) s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16((m % 3) ? a0 : a1, (m % 3 == 1) ? q1 : q0, s1, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, pf[j][0], o0, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, pf[j][1], o0, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, pf[j][0], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, pf[j][0], o1, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, pf[j][1], o1, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, pf[j][0], o1, 0, 0, 0);
            }
            float mx;
            softmax16(s0, c0, mx, pfn);
            mxa += mx;
#define SGB2 __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 3, 0); \
             __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 4, 0);
            SGB2 SGB2 SGB2 SGB2 SGB2 SGB2 SGB2 SGB2 SGB2 SGB2
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x402, 3, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) pf[j][pl] = pfn[j][pl];
            s0 = s1;          // (a real kernel alternates the two sets by unrolling; the copy is 16 moves here)
        }
    }
    float s = mxa;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += o0[r] + o1[r] + s0[r] + s1[r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
static void run(const uint4* ops, float* out, int wps) {
    const int iters = 20000, ncu = 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE>), dim3(ncu * wps), dim3(256), 0, 0, ops, out, 200);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE>), dim3(ncu * wps), dim3(256), 0, 0, ops, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double cyc = ms * 1e-3 * 2.4e9 / iters;
    printf("%-10s waves/SIMD %d : %7.1f ms  %7.1f nominal cycles per wave-step  MFMA util %5.1f %% of nominal  %7.1f TFLOP/s\n", MODE ? "pipelined" : "phases", wps, ms,
           cyc / wps, 100.0 * 21 * 32 * wps / cyc, 21.0 * 32768.0 * 4 * ncu * wps * iters / (ms * 1e-3) / 1e12);
}

int main() {
    std::vector<uint4> h(1024);
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
    CHECK(hipMalloc(&out, 256 * 3 * 256 * sizeof(float)));
    for (int wps = 1; wps <= 3; ++wps) {
        run<0>(ops, out, wps);
        run<1>(ops, out, wps);
    }
    return 0;
}
