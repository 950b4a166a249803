// Co-resident loads for tools/diag_token_pk.py: which ingredient of the split-precision kernels makes packed fp32 math of ANOTHER
// kernel's waves on the same CU go wrong?  Each kernel keeps <= 64 VGPRs and <= 48 KiB of LDS so that its workgroups share CUs (and
// SIMDs) with the token kernel's, runs ~`iters` loop trips, and touches nothing the token kernel uses.
//   mfma_f16_load   back-to-back v_mfma_f32_32x32x16_f16 on register operands (no LDS, no memory)
//   lds_dma_load    global_load_lds_dwordx4 streams into its own LDS allocation (no MFMA)
//   mfma_f32_load   v_mfma_f32_32x32x2_f32 (the exact-fp32 conv's instruction), register operands
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/diag/loads.so tools/diag/loads.hip
#include <hip/hip_runtime.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 hf8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void mfma_f16_kernel(float* out, int iters) {
    f16v acc = {};
    hf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x ^ i)); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    if (acc[0] == 12345.678f) out[0] = acc[1];
}
__global__ __launch_bounds__(256) void mfma_f32_kernel(float* out, int iters) {
    f16v acc = {};
    float a = 0.001f * threadIdx.x, b = 0.002f * (threadIdx.x ^ 5);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (acc[0] == 12345.678f) out[0] = acc[1];
}
__global__ __launch_bounds__(256) void lds_dma_kernel(const unsigned char* src, float* out, int iters, unsigned mask) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned off = (blockIdx.x * 65536u + wave * 1024u) & mask;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off + lane * 16),
                                             (__attribute__((address_space(3))) void*)(smem + (wave * 8 + u) * 1024), 16, 0, 0);
            off = (off + 4096u) & mask;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (reinterpret_cast<float*>(smem)[threadIdx.x] == 12345.678f) out[0] = 1.f;
}
extern "C" {
int diag_mfma_f16(float* out, int iters, void* stream) {
    hipLaunchKernelGGL(mfma_f16_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, out, iters);
    return (int)hipGetLastError();
}
int diag_mfma_f32(float* out, int iters, void* stream) {
    hipLaunchKernelGGL(mfma_f32_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, out, iters);
    return (int)hipGetLastError();
}
int diag_lds_dma(const void* src, float* out, int iters, unsigned mask, void* stream) {
    hipLaunchKernelGGL(lds_dma_kernel, dim3(1024), dim3(256), 32 * 1024, (hipStream_t)stream, (const unsigned char*)src, out, iters, mask);
    return (int)hipGetLastError();
}
}
