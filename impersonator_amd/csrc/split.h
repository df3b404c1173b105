// Device helpers of the split-bf16 (bf16x3) format, shared by conv.hip and train.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace lwg {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2w_t __attribute__((ext_vector_type(2)));

// two fp32 values -> their bf16 hi terms and lo terms (hi = bf16(v), lo = bf16(v - hi): conv.h), each pair packed in a
// dword with the first value in the low half
__device__ __forceinline__ void split_pair(float a, float b, unsigned &hi, unsigned &lo)
{
    bf16x2w_t h;
    h[0] = (__bf16)a;
    h[1] = (__bf16)b;
    hi = __builtin_bit_cast(unsigned, h);
    bf16x2w_t l;
    l[0] = (__bf16)(a - __builtin_bit_cast(float, hi << 16));
    l[1] = (__bf16)(b - __builtin_bit_cast(float, hi & 0xffff0000u));
    lo = __builtin_bit_cast(unsigned, l);
}

}  // namespace lwg
