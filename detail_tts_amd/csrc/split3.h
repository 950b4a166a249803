// fp32 -> three bf16 planes (v = p0 + p1 + p2 to 24 bits): device helpers shared by the split-precision conv and attention.
#pragma once
#include <hip/hip_runtime.h>

namespace dtts {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// two elements at a time on the hardware converter (v_cvt_pk_bf16_f32, round-to-nearest-even): 3 converts, 4 unpacks, 2 packed subs
__device__ __forceinline__ void split_pair(float x, float y, unsigned& w0, unsigned& w1, unsigned& w2) {
    const f32x2 v = {x, y};
    const bf16x2 p0 = __builtin_convertvector(v, bf16x2);
    const f32x2 r1 = v - __builtin_convertvector(p0, f32x2);
    const bf16x2 p1 = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(p1, f32x2);
    const bf16x2 p2 = __builtin_convertvector(r2, bf16x2);
    w0 = __builtin_bit_cast(unsigned, p0);
    w1 = __builtin_bit_cast(unsigned, p1);
    w2 = __builtin_bit_cast(unsigned, p2);
}
__device__ __forceinline__ void split8(const float* v, uint4& q0, uint4& q1, uint4& q2) {
    split_pair(v[0], v[1], q0.x, q1.x, q2.x);
    split_pair(v[2], v[3], q0.y, q1.y, q2.y);
    split_pair(v[4], v[5], q0.z, q1.z, q2.z);
    split_pair(v[6], v[7], q0.w, q1.w, q2.w);
}


}  // namespace dtts
