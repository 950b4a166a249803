// Device twin of oracle/philox.py: Philox4x32-10 keyed noise.
//   key = (seed lo, seed hi), counter = (block, sample, (stage<<16)|step, 0x44545453)
//   u = ((x >> 8) + 0.5) * 2^-24 ; normals by Box-Muller on (u0,u1) and (u2,u3).
#pragma once
#include <hip/hip_runtime.h>
#include "common.h"

namespace dtts {


__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned out[4]) {
    const unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        const unsigned hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float philox_u01(unsigned x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// 4 uniforms of block `blk`
__device__ __forceinline__ void philox_uniform4(unsigned long long seed, unsigned sample, int stage, int step, unsigned blk,
                                                float u[4]) {
    unsigned o[4];
    philox4x32_10(blk, sample, ((unsigned)stage << 16) | (unsigned)step, 0x44545453u, (unsigned)(seed & 0xffffffffull),
                  (unsigned)(seed >> 32), o);
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = philox_u01(o[i]);
}

// 4 normals of block `blk` (elements 4*blk .. 4*blk+3)
__device__ __forceinline__ void philox_normal4(unsigned long long seed, unsigned sample, int stage, int step, unsigned blk,
                                               float z[4]) {
    float u[4];
    philox_uniform4(seed, sample, stage, step, blk, u);
    const float two_pi = 6.283185307179586f;
    const float r0 = sqrtf(-2.0f * logf(u[0])), r1 = sqrtf(-2.0f * logf(u[2]));
    const float t0 = two_pi * u[1], t1 = two_pi * u[3];
    z[0] = r0 * cosf(t0); z[1] = r0 * sinf(t0);
    z[2] = r1 * cosf(t1); z[3] = r1 * sinf(t1);
}

__device__ __forceinline__ float philox_normal1(unsigned long long seed, unsigned sample, int stage, int step, unsigned elem) {
    float z[4];
    philox_normal4(seed, sample, stage, step, elem >> 2, z);
    return z[elem & 3];
}

}  // namespace dtts
