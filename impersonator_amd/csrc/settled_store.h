// Multi-dword global stores with room behind them (device code, header only).
//
// on a CU shared with the LDS-read + MFMA loop of the conv kernels, a VALU instruction that
// rewrites a data register of a 2/3/4-dword global store a few wait states after it can lose the last 16-lane pass of
// its write (tools/coresidency_repro.hip; profiles/r03_coresidency.md).  hipcc leaves 1..3 wait states there.  24 wait
// states, single-dword stores, or no VALU work behind the stores were clean in every experiment.  Every multi-dword
// store of liblwg that is followed by more work goes through one of these: the store(s) and 24 wait states of s_nop in
// ONE asm statement, so that nothing the compiler schedules can land in the window (an instruction that rewrites a data
// register cannot move ahead of the statement that reads it).  tests/test_store_hazard_lint.py compiles the library to
// assembly and fails on any store that escaped.  Cost: 24 cycles of one wave per group of stores.
#pragma once
#include <hip/hip_runtime.h>

namespace lwg {

#define LWG_SETTLE "\n\ts_nop 15\n\ts_nop 7"
typedef float settled_f4 __attribute__((ext_vector_type(4)));
typedef float settled_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ settled_f4 settled_v(const float4 &v) { return settled_f4{v.x, v.y, v.z, v.w}; }

__device__ __forceinline__ void store_x4_settled(float *p, const float4 &v)
{
    asm volatile("global_store_dwordx4 %0, %1, off" LWG_SETTLE ::"v"(p), "v"(settled_v(v)) : "memory");
}

// two 16-byte stores to p and p + 4 floats
__device__ __forceinline__ void store_2x4_settled(float *p, const float4 &v0, const float4 &v1)
{
    asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dwordx4 %0, %2, off offset:16" LWG_SETTLE ::"v"(p), "v"(settled_v(v0)), "v"(settled_v(v1))
                 : "memory");
}

__device__ __forceinline__ void store_4x4_settled(float *p0, float *p1, float *p2, float *p3, const float4 &v0, const float4 &v1,
                                                  const float4 &v2, const float4 &v3)
{
    asm volatile("global_store_dwordx4 %0, %4, off\n\tglobal_store_dwordx4 %1, %5, off\n\tglobal_store_dwordx4 %2, %6, off\n\t"
                 "global_store_dwordx4 %3, %7, off" LWG_SETTLE ::"v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(settled_v(v0)), "v"(settled_v(v1)),
                 "v"(settled_v(v2)), "v"(settled_v(v3))
                 : "memory");
}

typedef float settled_f3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void store_x3_settled(float *p, float a, float b, float c)
{
    asm volatile("global_store_dwordx3 %0, %1, off" LWG_SETTLE ::"v"(p), "v"(settled_f3{a, b, c}) : "memory");
}

__device__ __forceinline__ void store_x2_settled(float2 *p, const float2 &v)
{
    asm volatile("global_store_dwordx2 %0, %1, off" LWG_SETTLE ::"v"(p), "v"(settled_f2{v.x, v.y}) : "memory");
}

// eight 64-bit words (the conv kernels' trace records: measurement hook only)
typedef unsigned long long settled_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_u64x8_settled(unsigned long long *p, unsigned long long r0, unsigned long long r1,
                                                    unsigned long long r2, unsigned long long r3, unsigned long long r4,
                                                    unsigned long long r5, unsigned long long r6, unsigned long long r7)
{
    asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dwordx4 %0, %2, off offset:16\n\t"
                 "global_store_dwordx4 %0, %3, off offset:32\n\tglobal_store_dwordx4 %0, %4, off offset:48" LWG_SETTLE ::"v"(p),
                 "v"(settled_u2{r0, r1}), "v"(settled_u2{r2, r3}), "v"(settled_u2{r4, r5}), "v"(settled_u2{r6, r7})
                 : "memory");
}

#undef LWG_SETTLE

}  // namespace lwg
