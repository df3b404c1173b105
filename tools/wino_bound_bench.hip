// tools/wino_bound_bench.hip -- go/no-go bound for a Winograd F(2x2,3x3) trunk convolution (development tool, not part of liblwg).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DLWG_IGEMM_BENCH tools/wino_bound_bench.hip impersonator_amd/csrc/capi.hip -o tools/_build/wino_bound_bench
//
// F(2x2,3x3) turns the 512->512 3x3 convolution over P pixels into 16 GEMMs of (P/4 tiles) x 512 x 512 -- 2.25x fewer multiply-adds.
// With the output transform in registers (the only form that avoids a 4x round trip of the results through memory) the 16 transform
// positions of a tile set are walked INSIDE a workgroup: 16 x (512/32) = 256 stages of the same 24 MFMAs per wave as the direct
// kernel's 144, one (tiles x 32) activation slice and one (32 x Cout-tile) weight slice per stage.  That GEMM -- without the input
// transform (assumed precomputed by the producer), without the per-position accumulator hand-over and without the output transform --
// IS a 1x1 convolution with Cin = 16 x 512 over P/4 "pixels", i.e. something the production ring kernel already runs.  This tool
// times exactly that (an UPPER bound on the Winograd kernel's speed) next to the production halo kernel on the real layer.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../impersonator_amd/csrc/conv.hip"

using namespace lwg;

struct Shape { const char *name; int N, H, Cin, Cout, k, bn, dbg; double direct_flop; };

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    const double F16 = 2.0 * 16 * 1024 * 512.0 * 4608, F32 = 2 * F16;
    const Shape shapes[] = {
        {"direct 3x3 512->512 @32, 16 frames: halo kernel", 16, 32, 512, 512, 3, 128, 300, F16},
        {"direct 3x3 512->512 @32, 32 frames: halo kernel", 32, 32, 512, 512, 3, 128, 300, F32},
        {"wino bound, 16 frames: 4096 x 512 x 8192, 128x128 tiles (128 WGs)", 16, 16, 8192, 512, 1, 128, 240, F16},
        {"wino bound, 16 frames: 4096 x 512 x 8192, 128x64 tiles (256 WGs)", 16, 16, 8192, 512, 1, 64, 200, F16},
        {"wino bound, 32 frames: 8192 x 512 x 8192, 128x128 tiles (256 WGs)", 32, 16, 8192, 512, 1, 128, 240, F32},
        {"wino bound, 32 frames: 8192 x 512 x 8192, 128x128, 3-slot ring", 32, 16, 8192, 512, 1, 128, 200, F32},
        {"wino bound, 64 frames: 16384 x 512 x 8192, 256x128 tiles (256 WGs)", 64, 16, 8192, 512, 1, 128, 300, 2 * F32},
    };
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (const Shape &s : shapes) {
        const int pad = s.k / 2, Ho = s.H;
        const size_t xin = (size_t)s.N * s.H * s.H * s.Cin, yout = (size_t)s.N * Ho * Ho * s.Cout;
        const int K = s.k * s.k * s.Cin;
        float *y, *xs, *ws;
        float2 *part;
        hipMalloc(&y, yout * 4);
        const int mtiles = s.N * Ho * Ho / kConvBM;
        hipMalloc(&part, (size_t)mtiles * s.Cout * 8);
        std::vector<float> hx(xin), hw((size_t)s.Cout * K), t;
        for (auto &v : hx) v = (float)rand() / RAND_MAX * 2.f - 1.f;   // random data: realistic clocks (DVFS)
        for (auto &v : hw) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.02f;
        hipMalloc(&xs, xin * 4 + 4096);
        hipMemset(xs + xin, 0, 4096);
        hipMalloc(&ws, hw.size() * 4);
        t.resize(xin);
        split_bf16_groups(hx.data(), xin, t.data());
        hipMemcpy(xs, t.data(), xin * 4, hipMemcpyHostToDevice);
        t.resize(hw.size());
        split_bf16_groups(hw.data(), hw.size(), t.data());
        hipMemcpy(ws, t.data(), hw.size() * 4, hipMemcpyHostToDevice);
        ConvArgs a = {};
        a.w_split = ws;
        a.x = xs; a.zeros = xs + xin; a.ldx = s.Cin; a.N = s.N; a.H = s.H; a.W = s.H; a.Cin = s.Cin;
        a.cin_log2 = 0; while ((1 << a.cin_log2) < s.Cin) ++a.cin_log2;
        a.w = ws; a.y = y; a.ldy = s.Cout; a.Ho = Ho; a.Wo = Ho; a.Cout = s.Cout;
        a.Hm = Ho; a.Wm = Ho; a.stride = 1; a.pad = pad; a.os = 1; a.dil = 1;
        a.partials = part; a.mtiles = mtiles; a.nphase = 1; a.tap_inner = 1;
        a.ph[0].KH = a.ph[0].KW = s.k; a.ph[0].ntaps = s.k * s.k; a.ph[0].Kpad = K; a.ph[0].w_off = 0;
        for (int i = 0; i < 40; ++i) launch_conv_igemm_dbg(a, s.bn, s.dbg, st);   // let the clocks settle
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        for (int i = 0; i < reps; ++i) launch_conv_igemm_dbg(a, s.bn, s.dbg, st);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / reps, mfma_flop = 2.0 * s.N * Ho * Ho * (double)s.Cout * K;
        printf("%-72s %8.1f us per launch = %6.1f us per 8 frames; executed %6.1f TFLOP/s x3, direct-conv equivalent %6.1f TFLOP/s\n", s.name, us,
               us * 8 / s.N * (s.k == 1 ? 1 : 1), mfma_flop / (us * 1e-6) / 1e12, s.direct_flop / (us * 1e-6) / 1e12);
        hipFree(xs); hipFree(ws); hipFree(y); hipFree(part);
    }
    // ---- proxy for a 1-D Winograd F(2,3)-along-x kernel in the halo kernel's own structure (see profiles/r04_winograd_bound.md):
    // four accumulator sets M0..M3 (one per transform position), twelve (position, kernel-row) stage bodies per 32-channel slice,
    // 128 GEMM rows = 256 output pixels per workgroup.  The one-launch transposed convolution (CT = 1) IS that structure with nine
    // (phase, tap) bodies per slice and four accumulator sets: its time on 512 -> 512 at 8 frames (256 workgroups, one round),
    // scaled by 12 / 9, estimates the Winograd-x kernel on 16 frames (256 workgroups, one round) -- four epilogues instead of two on
    // the conservative side, the in-register output transform (two passes of 64 v_add per wave) not included.
    {
        const int N = 8, H = 32, Cin = 512, Cout = 512;
        const size_t xin = (size_t)N * H * H * Cin, yout = (size_t)N * 4 * H * H * Cout, wn = (size_t)Cout * 9 * Cin;
        float *y, *xs, *ws;
        float2 *part;
        hipMalloc(&y, yout * 4);
        const int mtiles = N * H * H / kConvBM;
        hipMalloc(&part, (size_t)4 * mtiles * Cout * 8);
        std::vector<float> hx(xin), hw(wn), t;
        for (auto &v : hx) v = (float)rand() / RAND_MAX * 2.f - 1.f;
        for (auto &v : hw) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.02f;
        hipMalloc(&xs, xin * 4 + 4096);
        hipMemset(xs + xin, 0, 4096);
        hipMalloc(&ws, wn * 4);
        t.resize(xin);
        split_bf16_groups(hx.data(), xin, t.data());
        hipMemcpy(xs, t.data(), xin * 4, hipMemcpyHostToDevice);
        t.resize(wn);
        split_bf16_groups(hw.data(), wn, t.data());
        hipMemcpy(ws, t.data(), wn * 4, hipMemcpyHostToDevice);
        ConvArgs a = {};
        a.w_split = ws; a.w = ws; a.precision = 1;
        a.x = xs; a.zeros = xs + xin; a.ldx = Cin; a.N = N; a.H = H; a.W = H; a.Cin = Cin; a.cin_log2 = 9;
        a.y = y; a.ldy = Cout; a.Ho = 2 * H; a.Wo = 2 * H; a.Cout = Cout;
        a.Hm = H; a.Wm = H; a.stride = 1; a.pad = 0; a.os = 2; a.dil = 1;
        a.partials = part; a.mtiles = mtiles; a.nphase = 4; a.tap_inner = 1;
        const int kh[4] = {1, 1, 2, 2}, kw[4] = {1, 2, 1, 2};
        long off = 0;
        for (int p = 0; p < 4; ++p) {
            a.ph[p].KH = kh[p]; a.ph[p].KW = kw[p]; a.ph[p].ntaps = kh[p] * kw[p]; a.ph[p].Kpad = kh[p] * kw[p] * Cin;
            a.ph[p].w_off = off; a.ph[p].oy0 = p >> 1; a.ph[p].ox0 = p & 1;
            off += (long)Cout * a.ph[p].Kpad;
        }
        int variant = -1;
        for (int i = 0; i < 40; ++i) launch_conv_igemm(a, 128, st, &variant);
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        for (int i = 0; i < reps; ++i) launch_conv_igemm(a, 128, st, &variant);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / reps;
        printf("%-72s %8.1f us per launch (variant %s); x 12/9 = %6.1f us per 16 frames of a Winograd-x trunk conv = %6.1f us per 8 frames\n",
               "transposed 3x3 512->512 @32->64, 8 frames: halo kernel CT=1 (256 WGs)", us, variant >= 0 ? kIgemmVariantNames[variant] : "?",
               us * 12 / 9, us * 12 / 9 / 2);
    }
    return 0;
}
