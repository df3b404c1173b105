// tools/bf16x3_bench.hip -- PROTOTYPE (not part of liblwg): fp32 convolution as three bf16 MFMA products
// (hi*hi + hi*lo + lo*hi, fp32 accumulate) against the exact-fp32 kernel, on the tsf-stream layer shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DLWG_IGEMM_BENCH tools/bf16x3_bench.hip impersonator_amd/csrc/capi.hip -o tools/_build/bf16x3_bench
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../impersonator_amd/csrc/conv.hip"

using namespace lwg;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int SP = 40;  // LDS row pitch in bf16 (80 B): conflict-free 16-byte fragment reads

template <int BN>
__global__ __launch_bounds__(256) void conv_bf16x3(const ConvArgs a, const __bf16 *__restrict__ w_hi,
                                                   const __bf16 *__restrict__ w_lo)
{
    constexpr int WM = 2, WN = BN / 64;        // wave tile 64 x (32*WN); waves 2 x 2
    constexpr int PLANE_A = BM * SP, PLANE_B = BN * SP;
    constexpr int STAGE = 2 * PLANE_A + 2 * PLANE_B;
    extern __shared__ __attribute__((aligned(16))) __bf16 sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const ConvPhase ph = a.ph[0];
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int hw_m = a.Hm * a.Wm, img = m0 / hw_m, rem0 = m0 - img * hw_m;

    const int lrow = tid >> 3, kq = tid & 7;
    int aoff[4];
    unsigned amask[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rem = rem0 + lrow + 32 * j;
        const int hm = rem / a.Wm, wm = rem - hm * a.Wm;
        const int hi0 = hm * a.stride - a.pad, wi0 = wm * a.stride - a.pad;
        aoff[j] = (hi0 * a.W + wi0) * a.ldx + kq * 4;
        unsigned m = 0;
        for (int t = 0, kh = 0, kw = 0; t < ph.ntaps; ++t) {
            m |= (unsigned)((unsigned)(hi0 + kh) < (unsigned)a.H && (unsigned)(wi0 + kw) < (unsigned)a.W) << t;
            if (++kw == ph.KW) { kw = 0; ++kh; }
        }
        amask[j] = m;
    }
    const float *xin = a.x + (size_t)img * a.H * a.W * a.ldx;
    const int brow = tid >> 2, bq = tid & 3;   // weight tile: BN rows x 4 chunks of 8 bf16
    const __bf16 *whi = w_hi + (size_t)(n0 + brow) * ph.Kpad + bq * 8;
    const __bf16 *wlo = w_lo + (size_t)(n0 + brow) * ph.Kpad + bq * 8;

    float4 ra[4];
    bf16x8 rbh[BN / 64], rbl[BN / 64];
    unsigned rvalid = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) ra[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    int s_tap = 0, s_kh = 0, s_kw = 0, s_ci0 = 0;
    auto load_stage = [&](int kt) {
        const int toff = (s_kh * a.W + s_kw) * a.ldx + s_ci0;
        rvalid = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) rvalid |= ((amask[j] >> s_tap) & 1u) << j;
        s_ci0 += BK;
        if (s_ci0 == a.Cin) { s_ci0 = 0; ++s_tap; if (++s_kw == ph.KW) { s_kw = 0; ++s_kh; } }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            ra[j] = *reinterpret_cast<const float4 *>(xin + (((rvalid >> j) & 1u) ? aoff[j] + toff : kq * 4));
#pragma unroll
        for (int j = 0; j < BN / 64; ++j) {
            rbh[j] = *reinterpret_cast<const bf16x8 *>(whi + (size_t)(64 * j) * ph.Kpad + kt * BK);
            rbl[j] = *reinterpret_cast<const bf16x8 *>(wlo + (size_t)(64 * j) * ph.Kpad + kt * BK);
        }
    };
    auto store_stage = [&](int buf) {
        __bf16 *s = sm + buf * STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 v = ((rvalid >> j) & 1u) ? ra[j] : make_float4(0.f, 0.f, 0.f, 0.f);
            bf16x4 h, l;
            h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
            l[0] = (__bf16)(v.x - (float)h[0]); l[1] = (__bf16)(v.y - (float)h[1]);
            l[2] = (__bf16)(v.z - (float)h[2]); l[3] = (__bf16)(v.w - (float)h[3]);
            const int o = (lrow + 32 * j) * SP + kq * 4;
            *reinterpret_cast<bf16x4 *>(s + o) = h;
            *reinterpret_cast<bf16x4 *>(s + PLANE_A + o) = l;
        }
#pragma unroll
        for (int j = 0; j < BN / 64; ++j) {
            const int o = 2 * PLANE_A + (brow + 64 * j) * SP + bq * 8;
            *reinterpret_cast<bf16x8 *>(s + o) = rbh[j];
            *reinterpret_cast<bf16x8 *>(s + PLANE_B + o) = rbl[j];
        }
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fr = (lane & 31) * SP + (lane >> 5) * 8;
    const int a_fr = (wave_m * 64) * SP + fr;
    const int b_fr = 2 * PLANE_A + (wave_n * 32 * WN) * SP + fr;

    auto body = [&](int kt, auto do_store, auto do_load) {
        const __bf16 *s = sm + (kt & 1) * STAGE;
#pragma unroll
        for (int kb = 0; kb < BK / 16; ++kb) {
            bf16x8 ah[WM], al[WM], bh[WN], bl[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8 *>(s + a_fr + i * 32 * SP + kb * 16);
                al[i] = *reinterpret_cast<const bf16x8 *>(s + PLANE_A + a_fr + i * 32 * SP + kb * 16);
            }
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                bh[j] = *reinterpret_cast<const bf16x8 *>(s + b_fr + j * 32 * SP + kb * 16);
                bl[j] = *reinterpret_cast<const bf16x8 *>(s + PLANE_B + b_fr + j * 32 * SP + kb * 16);
            }
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
            if (decltype(do_store)::value && kb == 0) store_stage((kt & 1) ^ 1);
            if (decltype(do_load)::value && kb == 1) load_stage(kt + 2);
        }
        __syncthreads();
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    const int nk = ph.Kpad / BK;
    load_stage(0);
    store_stage(0);
    if (nk > 1) load_stage(1);
    __syncthreads();
    int kt = 0;
    for (; kt + 2 < nk; ++kt) body(kt, yes{}, yes{});
    if (kt + 1 < nk) body(kt++, yes{}, no{});
    body(kt, no{}, no{});

    const int col = lane & 31, rsel = 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave_m * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + rsel;
            float *yo = a.y + (size_t)(m0 + row) * a.ldy + n0 + wave_n * 32 * WN + col;
#pragma unroll
            for (int j = 0; j < WN; ++j) yo[j * 32] = acc[i][j][r];
        }
}

struct Shape { const char *name; int N, H, Cin, Cout, k, stride, bn; };

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    const Shape shapes[] = {
        {"res 512->512 @32", 8, 32, 512, 512, 3, 1, 128},
        {"skip0 512->256 @64", 8, 64, 512, 256, 3, 1, 128},
        {"skip1 256->128 @128", 8, 128, 256, 128, 3, 1, 128},
        {"skip2 128->64 @256", 8, 256, 128, 64, 3, 1, 64},
    };
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (const Shape &s : shapes) {
        const int pad = s.k / 2, Ho = (s.H + 2 * pad - s.k) / s.stride + 1;
        const size_t xin = (size_t)s.N * s.H * s.H * s.Cin, yout = (size_t)s.N * Ho * Ho * s.Cout;
        const int K = s.k * s.k * s.Cin;
        float *x, *w, *y, *y2, *zeros;
        __bf16 *wh, *wl;
        hipMalloc(&x, xin * 4); hipMalloc(&w, (size_t)s.Cout * K * 4); hipMalloc(&y, yout * 4); hipMalloc(&y2, yout * 4);
        hipMalloc(&wh, (size_t)s.Cout * K * 2); hipMalloc(&wl, (size_t)s.Cout * K * 2);
        hipMalloc(&zeros, 256); hipMemset(zeros, 0, 256);
        std::vector<float> hx(xin), hw((size_t)s.Cout * K);
        for (auto &v : hx) v = (float)rand() / RAND_MAX * 4.f - 2.f;
        for (auto &v : hw) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.03f;
        std::vector<__bf16> hh(hw.size()), hl(hw.size());
        for (size_t i = 0; i < hw.size(); ++i) { hh[i] = (__bf16)hw[i]; hl[i] = (__bf16)(hw[i] - (float)hh[i]); }
        hipMemcpy(x, hx.data(), xin * 4, hipMemcpyHostToDevice);
        hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(wh, hh.data(), hh.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(wl, hl.data(), hl.size() * 2, hipMemcpyHostToDevice);
        ConvArgs a = {};
        a.x = x; a.ldx = s.Cin; a.N = s.N; a.H = s.H; a.W = s.H; a.Cin = s.Cin;
        a.cin_log2 = 0; while ((1 << a.cin_log2) < s.Cin) ++a.cin_log2;
        a.w = w; a.zeros = zeros; a.y = y; a.ldy = s.Cout; a.Ho = Ho; a.Wo = Ho; a.Cout = s.Cout;
        a.Hm = Ho; a.Wm = Ho; a.stride = s.stride; a.pad = pad; a.os = 1; a.dil = 1;
        a.partials = nullptr; a.mtiles = s.N * Ho * Ho / kConvBM; a.nphase = 1;
        a.ph[0].KH = a.ph[0].KW = s.k; a.ph[0].ntaps = s.k * s.k; a.ph[0].Kpad = K; a.ph[0].w_off = 0;
        const double flop = 2.0 * s.N * Ho * Ho * (double)s.Cout * K;
        for (int i = 0; i < 200; ++i) launch_conv_igemm(a, s.bn, st);   // warm clocks
        hipEventRecord(e0, st);
        for (int i = 0; i < reps; ++i) launch_conv_igemm(a, s.bn, st);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms32 = 0; hipEventElapsedTime(&ms32, e0, e1);
        ConvArgs b = a; b.y = y2;
        const dim3 grid(a.mtiles, s.Cout / s.bn);
        const size_t lds = (size_t)2 * (2 * BM * SP + 2 * s.bn * SP) * 2;
        auto launch = [&]() {
            if (s.bn == 128) {
                hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_bf16x3<128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                conv_bf16x3<128><<<grid, 256, lds, st>>>(b, wh, wl);
            } else {
                hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_bf16x3<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                conv_bf16x3<64><<<grid, 256, lds, st>>>(b, wh, wl);
            }
        };
        for (int i = 0; i < 50; ++i) launch();
        hipEventRecord(e0, st);
        for (int i = 0; i < reps; ++i) launch();
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float msb = 0; hipEventElapsedTime(&msb, e0, e1);
        std::vector<float> r1(yout), r2(yout);
        hipMemcpy(r1.data(), y, yout * 4, hipMemcpyDeviceToHost);
        hipMemcpy(r2.data(), y2, yout * 4, hipMemcpyDeviceToHost);
        double maxd = 0, maxv = 0;
        for (size_t i = 0; i < yout; ++i) { maxd = fmax(maxd, fabs((double)r1[i] - r2[i])); maxv = fmax(maxv, fabs((double)r1[i])); }
        printf("%-22s fp32 %6.1f TF   bf16x3 %6.1f TF-equiv (%.2fx)   max|d| %.3g (max|y| %.3g)  err %s\n", s.name,
               flop * reps / (ms32 * 1e-3) / 1e12, flop * reps / (msb * 1e-3) / 1e12, ms32 / msb, maxd, maxv,
               hipGetErrorString(hipGetLastError()));
        hipFree(x); hipFree(w); hipFree(y); hipFree(y2); hipFree(wh); hipFree(wl); hipFree(zeros);
    }
    return 0;
}
