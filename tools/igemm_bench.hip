// tools/igemm_bench.hip -- development micro-benchmark of conv_igemm_f32 (not part of liblwg).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DLWG_IGEMM_BENCH tools/igemm_bench.hip impersonator_amd/csrc/capi.hip -o tools/_build/igemm_bench
// Times the production kernel and its ablation variants (see the DBG template parameter in conv.hip) on the layer
// shapes of the tsf stream at batch 8, so that a change to the main loop is judged in one GPU call.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../impersonator_amd/csrc/conv.hip"

using namespace lwg;

struct Shape { const char *name; int N, H, Cin, Cout, k, stride, bn; };

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    const Shape shapes[] = {
        {"res 512->512 @32 (bn128)", 8, 32, 512, 512, 3, 1, 128},
        {"res 512->512 @32 N=16 (bn128)", 16, 32, 512, 512, 3, 1, 128},
        {"skip0 512->256 @64 (bn128)", 8, 64, 512, 256, 3, 1, 128},
        {"skip1 256->128 @128 (bn128)", 8, 128, 256, 128, 3, 1, 128},
        {"skip2 128->64 @256 (bn64)", 8, 256, 128, 64, 3, 1, 64},
        {"enc3 256->512 s2 @64 (bn128)", 8, 64, 256, 512, 3, 2, 128},
        {"enc3 256->512 s2 @64 N=16 (bn128)", 16, 64, 256, 512, 3, 2, 128},
        {"enc2 128->256 s2 @128 (bn128)", 8, 128, 128, 256, 3, 2, 128},
        {"enc1 64->128 s2 @256 (bn128)", 8, 256, 64, 128, 3, 2, 128},
    };
    // 100: fp32 DMA kernel; 200 / 240: bf16x3 DMA ring with 3 / 4 slots; 300: the product's choice (halo kernel on 3x3 s1);
    // 1240: ring kernel with a quarter of the activation DMAs (timing only); 241 / 245: no steady-state DMA / MFMAs only
    const int dbgs[] = {100, 200, 240, 300, 1240, 241, 300, 240};
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (const Shape &s : shapes) {
        const int pad = s.k / 2, Ho = (s.H + 2 * pad - s.k) / s.stride + 1;
        const size_t xin = (size_t)s.N * s.H * s.H * s.Cin, yout = (size_t)s.N * Ho * Ho * s.Cout;
        const int K = s.k * s.k * s.Cin;
        float *x, *w, *y;
        float2 *part;
        float *zeros;
        hipMalloc(&zeros, 256);
        hipMemset(zeros, 0, 256);
        hipMalloc(&x, xin * 4);
        hipMalloc(&w, (size_t)s.Cout * K * 4);
        hipMalloc(&y, yout * 4);
        const int mtiles = s.N * Ho * Ho / kConvBM;
        hipMalloc(&part, (size_t)mtiles * s.Cout * 8);
        std::vector<float> hx(xin), hw((size_t)s.Cout * K);
        for (auto &v : hx) v = (float)rand() / RAND_MAX * 2.f - 1.f;   // random data: realistic clocks (DVFS)
        for (auto &v : hw) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.02f;
        hipMemcpy(x, hx.data(), xin * 4, hipMemcpyHostToDevice);
        hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
        // split-bf16 copies of both operands for the bf16x3 variants (dbg >= 200)
        float *xs, *ws;
        hipMalloc(&xs, xin * 4 + 4096);   // + the zero run the bf16x3 kernel wants behind its input
        hipMemset(xs + xin, 0, 4096);
        hipMalloc(&ws, hw.size() * 4);
        {
            std::vector<float> t(xin);
            split_bf16_groups(hx.data(), xin, t.data());
            hipMemcpy(xs, t.data(), xin * 4, hipMemcpyHostToDevice);
            t.resize(hw.size());
            split_bf16_groups(hw.data(), hw.size(), t.data());
            hipMemcpy(ws, t.data(), hw.size() * 4, hipMemcpyHostToDevice);
        }
        ConvArgs a = {};
        a.w_split = ws;
        a.x = x; a.ldx = s.Cin; a.N = s.N; a.H = s.H; a.W = s.H; a.Cin = s.Cin;
        a.cin_log2 = 0; while ((1 << a.cin_log2) < s.Cin) ++a.cin_log2;
        a.w = w; a.zeros = zeros; a.y = y; a.ldy = s.Cout; a.Ho = Ho; a.Wo = Ho; a.Cout = s.Cout;
        a.Hm = Ho; a.Wm = Ho; a.stride = s.stride; a.pad = pad; a.os = 1; a.dil = 1;
        a.partials = part; a.mtiles = mtiles; a.nphase = 1;
        {   // LWG_K_ORDER=channel: the round-2 walk of the reduction (default: taps innermost)
            const char *ko = getenv("LWG_K_ORDER");
            a.tap_inner = (ko && ko[0] == 'c') ? 0 : 1;
        }
        a.ph[0].KH = a.ph[0].KW = s.k; a.ph[0].ntaps = s.k * s.k; a.ph[0].Kpad = K; a.ph[0].w_off = 0;
        const double flop = 2.0 * s.N * Ho * Ho * (double)s.Cout * K;
        printf("%-30s", s.name);
        for (int i = 0; i < 300; ++i) launch_conv_igemm_dbg(a, s.bn, 100, st);   // ~100 ms: let the clocks settle
        hipStreamSynchronize(st);
        for (int dbg : dbgs) {
            a.x = dbg >= 200 ? xs : x;
            a.zeros = dbg >= 200 ? xs + xin : zeros;
            for (int i = 0; i < 3; ++i) launch_conv_igemm_dbg(a, s.bn, dbg, st);
            hipEventRecord(e0, st);
            for (int i = 0; i < reps; ++i) launch_conv_igemm_dbg(a, s.bn, dbg, st);
            hipEventRecord(e1, st);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            printf("  d%-3d %6.1f", dbg, flop * reps / (ms * 1e-3) / 1e12);
        }
        {   // sanity: bf16x3 result against the fp32 kernel's
            std::vector<float> y0(yout), y1(yout);
            a.x = x;
            a.zeros = zeros;
            launch_conv_igemm_dbg(a, s.bn, 100, st);
            hipStreamSynchronize(st);
            hipMemcpy(y0.data(), y, yout * 4, hipMemcpyDeviceToHost);
            a.x = xs;
            a.zeros = xs + xin;
            hipMemset(y, 0, yout * 4);
            launch_conv_igemm_dbg(a, s.bn, 300, st);   // the halo kernel walks the reduction in the ring kernel's order: bit-identical
            hipStreamSynchronize(st);
            hipMemcpy(y1.data(), y, yout * 4, hipMemcpyDeviceToHost);
            std::vector<float> y2(yout);
            hipMemset(y, 0, yout * 4);
            launch_conv_igemm_dbg(a, s.bn, 200, st);
            hipStreamSynchronize(st);
            hipMemcpy(y2.data(), y, yout * 4, hipMemcpyDeviceToHost);
            size_t nbad = 0;
            for (size_t i = 0; i < yout; ++i) nbad += y1[i] != y2[i];
            printf("  halo!=ring: %zu", nbad);
            double md = 0, mx = 0;
            for (size_t i = 0; i < yout; ++i) {
                md = fmax(md, fabs((double)y0[i] - y1[i]));
                mx = fmax(mx, fabs((double)y0[i]));
            }
            printf("  |d|max %.2e of %.2f", md, mx);
        }
        printf("\n");
        hipFree(xs); hipFree(ws); hipFree(x); hipFree(w); hipFree(y); hipFree(part); hipFree(zeros);
    }
    return 0;
}
