// tools/mfma_peak.hip -- what a wave can get out of v_mfma_f32_32x32x16_bf16 (development probe, not part of liblwg).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/_build/mfma_peak
// One wave per SIMD (256 threads per CU-sized workgroup), NACC accumulators walked round-robin so that a dependent
// MFMA follows its producer after NACC-1 independent ones; operands random (power) or zero.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(const float4 *in, float *out, int iters)
{
    const float4 av = in[threadIdx.x], bv = in[256 + threadIdx.x];
    const bf16x8 a = __builtin_bit_cast(bf16x8, av), b = __builtin_bit_cast(bf16x8, bv);
    f32x16 acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32 / NACC; ++u)
#pragma unroll
            for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[k][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
static void run(const char *name, const float4 *in, float *out, int blocks)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4000;
    for (int w = 0; w < 3; ++w) mfma_loop<NACC><<<blocks, 256>>>(in, out, iters);   // ~100 ms: settle the clocks
    hipEventRecord(e0);
    for (int w = 0; w < 3; ++w) mfma_loop<NACC><<<blocks, 256>>>(in, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 3.0 * blocks * 4 * (double)iters * 32 * 32768.0;
    printf("%-28s %8.1f TFLOP/s  (%.1f%% of 2500)\n", name, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 2.5e15 * 100);
}

int main()
{
    float4 *in;
    float *out;
    hipMalloc(&in, 512 * 16);
    hipMalloc(&out, 1024 * 256 * 4);
    std::vector<unsigned short> h(512 * 8);
    for (int pass = 0; pass < 2; ++pass) {
        for (auto &v : h) v = pass ? (unsigned short)(0x3c00 + (rand() & 0x3ff)) : 0;   // bf16 in [0.0078, 0.03) / zeros
        hipMemcpy(in, h.data(), 512 * 16, hipMemcpyHostToDevice);
        printf("-- %s operands, 256 workgroups (one wave per SIMD)\n", pass ? "random" : "zero");
        run<2>("2 accumulators", in, out, 256);
        run<4>("4 accumulators", in, out, 256);
        run<8>("8 accumulators", in, out, 256);
        printf("-- %s operands, 512 workgroups (two waves per SIMD)\n", pass ? "random" : "zero");
        run<4>("4 accumulators", in, out, 512);
    }
    return 0;
}
