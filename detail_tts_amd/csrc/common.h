// Common definitions for libdetail_hip.so (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>

namespace dtts {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define DTTS_CHECK_HIP(expr)                                                                  \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            throw ::dtts::Error(-2, std::string(#expr) + " failed: " + hipGetErrorString(_e) + \
                                        " at " + __FILE__ + ":" + std::to_string(__LINE__)); \
    } while (0)

#define DTTS_REQUIRE(cond, msg)                                                      \
    do {                                                                             \
        if (!(cond))                                                                 \
            throw ::dtts::Error(-1, std::string("invalid argument: ") + (msg) +      \
                                        " [" #cond "] at " + __FILE__ + ":" + std::to_string(__LINE__)); \
    } while (0)

// Opt-in LDS sizes (hipFuncSetAttribute(MaxDynamicSharedMemorySize)) are a per-DEVICE property of a kernel: applied once per
// (device, kernel), under a lock, so that handles on several GPUs of one process and the stage threads of infer_stream are all served
// (device.hip).  lds_optin throws on failure; device_fits reports whether the current device can hold `wgs` co-resident workgroups
// of `lds_bytes` each, one per CU (the persistent kernels' requirement).
void lds_optin(const void* kernel, int bytes);
bool device_fits(int wgs, int lds_bytes);

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return cdiv(a, b) * b; }

// noise streams of the Philox spec (oracle/philox.py)
enum NoiseStage : int { STAGE_GPT_SAMPLE = 1, STAGE_DIFF_INIT = 2, STAGE_DIFF_STEP = 3, STAGE_FLOW_PRIOR = 4 };

// activation ids shared by prologues / epilogues
enum Act : int {
    ACT_NONE = 0,
    ACT_SILU = 1,
    ACT_LRELU = 2,     // slope given separately
    ACT_RELU = 3,
    ACT_GELU_NEW = 4,
    ACT_TANH = 5,
    ACT_MISH = 6,
    ACT_LOG_CLAMP = 7, // log(max(v, 1e-5))   (dynamic_range_compression_torch, vqvae/utils/data_utils.py:21-27)
};

// epilogue pairing modes: packed weight rows (2r, 2r+1) hold the two halves of a gated pair
enum Gate : int {
    GATE_NONE = 0,
    GATE_TANH_SIGMOID = 1,   // tanh(a_r) * sigmoid(b_r)            (WN, modules.py:15-22)
    GATE_GLU = 2,            // a_r * sigmoid(b_r)                   (Conv1dGLU, modules.py:517-523)
};

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
    switch (act) {
        case ACT_SILU: return v / (1.f + __expf(-v));
        case ACT_LRELU: return v >= 0.f ? v : v * slope;
        case ACT_RELU: return fmaxf(v, 0.f);
        case ACT_GELU_NEW: {
            float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
            return 0.5f * v * (1.f + tanhf(u));
        }
        case ACT_TANH: return tanhf(v);
        case ACT_MISH: {
            float sp = v > 20.f ? v : log1pf(expf(v));
            return v * tanhf(sp);
        }
        case ACT_LOG_CLAMP: return logf(fmaxf(v, 1e-5f));
        default: return v;
    }
}

// XCD-aware block remap (block i runs on XCD i % 8 on MI355X; used for L2 locality only, never for correctness):
// returns a logical block id such that each XCD owns one contiguous range of logical ids.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int i, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = i & 7, j = i >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + __expf(-v)); }

// sums of 8 values per lane over the wave in 10 shuffles (halving exchange): every lane gets the total of value (lane >> 3) & 7
__device__ __forceinline__ float wave_sum8(const float (&s)[8], int lane) {
    const bool h32 = lane & 32, h16 = lane & 16, h8 = lane & 8;
    float t[4], u[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = (h32 ? s[4 + i] : s[i]) + __shfl_xor(h32 ? s[i] : s[4 + i], 32);
#pragma unroll
    for (int i = 0; i < 2; ++i) u[i] = (h16 ? t[2 + i] : t[i]) + __shfl_xor(h16 ? t[i] : t[2 + i], 16);
    float w = (h8 ? u[1] : u[0]) + __shfl_xor(h8 ? u[0] : u[1], 8);
    w += __shfl_xor(w, 4);
    w += __shfl_xor(w, 2);
    w += __shfl_xor(w, 1);
    return w;
}

}  // namespace dtts
