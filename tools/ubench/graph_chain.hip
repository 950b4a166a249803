// Does hipGraph shorten a dependent chain of small kernels (the GPT decode step: ~72 launches of 5-12 us) on this platform?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void small(float* x, int n, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = x[i];
    for (int k = 0; k < iters; ++k) v = v * 1.0001f + 0.5f;
    x[i] = v;
}
int main() {
    float* x; (void)hipMalloc(&x, 1 << 22);
    hipStream_t s; (void)hipStreamCreate(&s);
    const int n = 1 << 18, chain = 72, reps = 50;
    for (int iters : {8, 256, 2048}) {
        auto launch_chain = [&]() { for (int k = 0; k < chain; ++k) hipLaunchKernelGGL(small, dim3(n / 256), dim3(256), 0, s, x, n, iters); };
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        launch_chain(); (void)hipStreamSynchronize(s);
        (void)hipEventRecord(a, s);
        for (int r = 0; r < reps; ++r) launch_chain();
        (void)hipEventRecord(b, s); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        const float plain = ms / reps / chain * 1e3f;
        hipGraph_t g; hipGraphExec_t ge;
        (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        launch_chain();
        (void)hipStreamEndCapture(s, &g);
        (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s);
        (void)hipEventRecord(a, s);
        for (int r = 0; r < reps; ++r) (void)hipGraphLaunch(ge, s);
        (void)hipEventRecord(b, s); (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&ms, a, b);
        printf("iters %5d: plain launches %.2f us per kernel, graph %.2f us per kernel\n", iters, plain, ms / reps / chain * 1e3f);
    }
    return 0;
}
