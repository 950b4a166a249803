// fp32 -> two fp16 planes (v s = h0 + h1 to 22 bits, s a power of two): device helpers shared by the split-precision conv and
// attention.  See conv_x3.h for the arithmetic.
#pragma once
#include <hip/hip_runtime.h>

namespace dtts {

typedef _Float16 hf8 __attribute__((ext_vector_type(8)));
typedef _Float16 hf2 __attribute__((ext_vector_type(2)));
typedef _Float16 hf4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int XS_PLANES = 2;
// Power-of-two operand scales (exact; undone by one multiply of the fp32 accumulator): they keep the LOW plane of typical weights
// (|w| ~ 0.03) and activations (|x| ~ 1) in fp16's normal range.  Elements so small that their low plane is subnormal keep an
// ABSOLUTE error <= 2^-25 / scale; elements beyond +-65504 / scale saturate (never inf).
constexpr float XS_SCALE_W = 64.f, XS_SCALE_X = 16.f;
constexpr float XS_ACC_SCALE = 1.f / (XS_SCALE_W * XS_SCALE_X);

// two elements at a time: clamp, round to fp16 (v_cvt_pk_f16_f32, nearest-even), residual (exact in fp32), round again
__device__ __forceinline__ void split_pair(float x, float y, unsigned& w0, unsigned& w1) {
    const f32x2 v = {__builtin_amdgcn_fmed3f(x, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(y, -65504.f, 65504.f)};
    const hf2 p0 = __builtin_convertvector(v, hf2);
    const f32x2 r1 = v - __builtin_convertvector(p0, f32x2);
    const hf2 p1 = __builtin_convertvector(r1, hf2);
    w0 = __builtin_bit_cast(unsigned, p0);
    w1 = __builtin_bit_cast(unsigned, p1);
}
// the same for values known to lie inside fp16's range (softmax numerators): no clamp, and the residual as ONE mixed-precision fma
// per element (v_fma_mix_f32 reads the fp16 half in place: x - float(h) without converting back; the compiler does not select it)
__device__ __forceinline__ void split_pair_inrange(float x, float y, unsigned& w0, unsigned& w1) {
    const f32x2 v = {x, y};
    w0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, hf2));
    f32x2 r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r1[0]) : "v"(w0), "v"(x));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1[1]) : "v"(w0), "v"(y));
    w1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, hf2));
}
__device__ __forceinline__ void split8_inrange(const float* v, uint4& q0, uint4& q1) {
    split_pair_inrange(v[0], v[1], q0.x, q1.x);
    split_pair_inrange(v[2], v[3], q0.y, q1.y);
    split_pair_inrange(v[4], v[5], q0.z, q1.z);
    split_pair_inrange(v[6], v[7], q0.w, q1.w);
}
// v[0..7] (already scaled) -> one 16-byte chunk per plane
__device__ __forceinline__ void split8(const float* v, uint4& q0, uint4& q1) {
    split_pair(v[0], v[1], q0.x, q1.x);
    split_pair(v[2], v[3], q0.y, q1.y);
    split_pair(v[4], v[5], q0.z, q1.z);
    split_pair(v[6], v[7], q0.w, q1.w);
}

}  // namespace dtts
