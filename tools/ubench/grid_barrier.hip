// How fast is a device-wide barrier inside ONE persistent kernel on gfx950 (8 XCDs, one L2 each)?  Decides whether the GPT decode step
// (53 dependent launches of 5-10 us each) is better served by a persistent kernel with grid barriers.
//   hipcc --offload-arch=gfx950 -O3 -o bin/grid_barrier grid_barrier.hip
// Every workgroup publishes a value before barrier k and checks a distant workgroup's value after it (visibility across XCDs), optionally
// streaming `bytes` of "weights" per phase (issued BEFORE the barrier wait, as the real kernel would prefetch them).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

template <int MODE>   // 0: barrier only; 1: + publish/check a value; 2: + stream weights (float4 per thread per iteration) before the wait
__global__ __launch_bounds__(256) void persist(unsigned* counter, float* slots, const float4* __restrict__ w, long long w_per_phase, int phases, int* errors,
                                               float* sink) {
    const int nwg = gridDim.x, wg = blockIdx.x;
    float4 acc = {0, 0, 0, 0};
    for (int k = 0; k < phases; ++k) {
        if (MODE >= 2) {
            const float4* wp = w + (long long)(k & 7) * w_per_phase;
            for (long long i = (long long)wg * 256 + threadIdx.x; i < w_per_phase; i += (long long)nwg * 256) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(wp + i));
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        if (MODE >= 1 && threadIdx.x < 32) slots[(size_t)wg * 32 + threadIdx.x] = (float)(k + 1) + acc.x * 0.f;
        grid_barrier(counter, (unsigned)(k + 1) * nwg);
        if (MODE >= 1 && threadIdx.x < 32) {
            const float v = __builtin_nontemporal_load(slots + (size_t)((wg + 97) % nwg) * 32 + threadIdx.x);
            if (v != (float)(k + 1)) atomicAdd(errors, 1);
        }
        if (MODE >= 1) {
            // second barrier: nobody may overwrite its slot before everyone has read (the real kernel double-buffers instead)
            grid_barrier(counter + 32, (unsigned)(k + 1) * nwg);
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) sink[0] = acc.x;
}

template <int MODE>
void run(int nwg, int phases, long long bytes_per_phase, const char* name) {
    unsigned* counter; float* slots; float4* w; int* errors; float* sink;
    (void)hipMalloc(&counter, 256 * 4); (void)hipMalloc(&slots, (size_t)nwg * 32 * 4); (void)hipMalloc(&errors, 4); (void)hipMalloc(&sink, 4);
    const long long wpp = bytes_per_phase / 16;
    (void)hipMalloc(&w, (size_t)(wpp ? wpp : 1) * 16 * 8);
    (void)hipMemset(w, 0, (size_t)(wpp ? wpp : 1) * 16 * 8);
    (void)hipMemset(errors, 0, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        (void)hipMemset(counter, 0, 256 * 4);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(persist<MODE>, dim3(nwg), dim3(256), 0, 0, counter, slots, w, wpp, phases, errors, sink);
        (void)hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess) { printf("%s: failed\n", name); return; }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    int herr = 0; (void)hipMemcpy(&herr, errors, 4, hipMemcpyDeviceToHost);
    const int nb = MODE >= 1 ? 2 : 1;
    printf("%-34s %4d WGs  %5d phases (%d barriers each)  %8.2f us / phase  %s", name, nwg, phases, nb, best * 1e3 / phases, herr ? "VISIBILITY ERRORS " : "");
    if (bytes_per_phase) printf(" %6.2f TB/s", bytes_per_phase / (best * 1e-3 / phases) / 1e12);
    printf("\n");
    (void)hipFree(counter); (void)hipFree(slots); (void)hipFree(w); (void)hipFree(errors); (void)hipFree(sink);
}

int main() {
    for (int nwg : {64, 128, 256, 512}) {
        run<0>(nwg, 2000, 0, "barrier only");
        run<1>(nwg, 2000, 0, "publish + barrier + check + barrier");
        run<2>(nwg, 1000, 4ll << 20, "4 MiB stream + publish/check");
        run<2>(nwg, 1000, 16ll << 20, "16 MiB stream + publish/check");
    }
    return 0;
}
