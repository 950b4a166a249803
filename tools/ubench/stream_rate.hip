// How fast can ONE workgroup per CU stream weights (global -> registers, 16 B per lane, all loads issued up front)?  Sizes the persistent
// GPT token kernel: per-CU bandwidth is latency x outstanding-request bound, so the number of resident workgroups sets the chip-level rate.
//   hipcc --offload-arch=gfx950 -O3 -o bin/stream_rate stream_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int NV, int MODE>      // NV float4 per thread per round; MODE 0: plain loads, 1: nontemporal, 2: agent-scope (sc1) 8-byte atomics
__global__ __launch_bounds__(1024) void stream(const float4* __restrict__ w, long long per_wg_vec, int rounds, float* sink) {
    const float4* base = w + (long long)blockIdx.x * per_wg_vec;
    float4 acc = {0, 0, 0, 0};
    const int nt = blockDim.x;
    for (int r = 0; r < rounds; ++r) {
        const float4* p = base + (long long)r * NV * nt + threadIdx.x;
        float4 v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (MODE == 0) v[i] = p[(long long)i * nt];
            else if (MODE == 1) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p + (long long)i * nt));
                v[i] = make_float4(t.x, t.y, t.z, t.w);
            } else {
                const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p + (long long)i * nt);
                const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v[i] = make_float4(__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned)b), __uint_as_float((unsigned)(b >> 32)));
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) sink[0] = acc.x;
}

template <int NV, int MODE>
void run(int nwg, int nt, int rounds, float4* w, size_t total_vec, float* sink, const char* name) {
    const long long per = (long long)NV * nt * rounds;
    if ((size_t)per * nwg > total_vec) { printf("skip\n"); return; }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int it = 0; it < 5; ++it) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((stream<NV, MODE>), dim3(nwg), dim3(nt), 0, 0, w, per, rounds, sink);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes = (double)per * 16 * nwg;
    printf("%-10s wgs %4d x %4d thr, %3d f4/thread/round x %3d rounds: %7.1f KB/wg  %8.1f us  %7.1f GB/s total  %6.1f GB/s per wg\n", name, nwg, nt, NV, rounds,
           per * 16 / 1024.0, best * 1e3, bytes / best * 1e-6, bytes / best * 1e-6 / nwg);
}

int main() {
    const size_t total_vec = (size_t)1 << 26;      // 1 GiB
    float4* w; float* sink;
    (void)hipMalloc(&w, total_vec * 16); (void)hipMalloc(&sink, 4);
    (void)hipMemset(w, 0, total_vec * 16);
    for (int nwg : {64, 128, 256, 512}) {
        run<36, 0>(nwg, 256, 1, w, total_vec, sink, "plain");      // one 147 KB slice, like a c_fc prefetch
        run<36, 0>(nwg, 256, 8, w, total_vec, sink, "plain");
        run<9, 0>(nwg, 256, 32, w, total_vec, sink, "plain");
        run<36, 0>(nwg, 1024, 8, w, total_vec, sink, "plain");
        run<36, 1>(nwg, 256, 8, w, total_vec, sink, "nontemp");
        run<36, 2>(nwg, 256, 8, w, total_vec, sink, "sc1 8B");
    }
    return 0;
}
