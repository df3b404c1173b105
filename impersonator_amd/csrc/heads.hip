// heads.hip -- the generator's 7x7 regression heads on the bf16x3 matrix path (gfx950).
//
// Replaces, on the default (split-bf16) precision, what PyTorch runs for ResUnetGenerator.regress
// (networks/generator.py:180-184: img_reg = Conv2d(64, 3, k7, p3) + Tanh, attetion_reg = Conv2d(64, 1, k7, p3) +
// Sigmoid) together with the InstanceNorm + ReLU of the layer in front of it and the blend of Imitator.forward
// (models/imitator.py:331).  conv.hip's heads_kernel does the same on the vector ALU (exact fp32, 59 TFLOP/s); this
// kernel puts the 13 GFLOP of a batch on v_mfma_f32_32x32x16_bf16.
//
// N = 4 outputs cannot fill a 32-wide MFMA tile, (kw, output) = 7 x 4 = 28 columns can:
//     U[y][x'][(kw, o)] = sum_{kh, c} Xn[y + kh - 3][x'][c] * W[kh][kw][c][o]        K = 7 x 64 = 448, N = 28 (of 32)
//     out[y][x][o]      = sum_{kw} U[y][x + kw - 3][(kw, o)]                            seven shifted adds
// i.e. a 7x1 (vertical) convolution with 28 output columns as the GEMM, then a horizontal sum across the tile.
//
// One wave (a 64-lane workgroup, all 512 registers of its SIMD lane slots) owns a strip of 32 staged columns
// (26 output columns + 3 + 3 halo) and a band of output rows, and marches down the input rows:
//   * the hi half of the B operand -- 7 x 4 k-step fragments, 112 registers -- stays in registers for the wave's life,
//     the lo half (used by one product in three) in 28 KiB of wave-private LDS;
//   * an input row is loaded straight from global memory in MFMA A-fragment shape (lane = staged column x 8 channels),
//     normalised (scale/shift of the InstanceNorm in front, ReLU), split into bf16 hi/lo in registers: no LDS staging,
//     no barrier anywhere in the loop; loads run TWO rows ahead of the MFMAs (a wave is alone on its SIMD, so nothing
//     else hides HBM latency: with one row of look-ahead the kernel spent 60 % of its time waiting);
//   * row r feeds the seven output rows r-3 .. r+3 (kh = 6 .. 0), whose accumulators sit in a ring of EIGHT 32x32 tiles
//     (one spare, so that the row loop unrolls eight times: ring positions and the two alternating load buffers are
//     all compile-time);
//   * a finished output row leaves through 4 KiB of wave-private LDS for the horizontal sum, then tanh / sigmoid /
//     blend and plain stores to the NCHW planes;
//   * each step is one basic block in which the MFMAs of row r, the loads of row r + 2, the conversion of row r + 1
//     and the reduction of the previous output row are interleaved (sched_group_barrier): with the phases one after
//     the other the matrix pipe idled 65 % of the time.
// Products are evaluated as lo*hi + hi*lo + hi*hi with fp32 accumulation, as in conv_igemm_bf16x3 (conv.h).
#include <type_traits>

#include "conv.h"

namespace lwg {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

constexpr int HS_COLS = 32;            // staged input columns per wave (one MFMA row tile)
constexpr int HS_OUT = HS_COLS - 6;    // output columns per wave
constexpr int HS_KS = 4;               // k-steps of 16 channels (64 input channels)
constexpr int HS_NF = 7 * HS_KS;       // B fragments per plane: (kh, ks)
constexpr int HS_UP = 33;              // pitch of the horizontal-sum scratch (floats)

// wfrag layout: [(kh * 4 + ks) * 2 + plane][lane] x 16 bytes; lane = 32 * (k half) + column n, n = kw * 4 + output
__global__ __launch_bounds__(256) void heads_pack_kernel(const float *__restrict__ wh, uint4 *__restrict__ wfrag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HS_NF * 2 * 64) return;
    const int lane = i & 63, plane = (i >> 6) & 1, f = i >> 7;
    const int kh = f / HS_KS, ks = f - kh * HS_KS;
    const int n = lane & 31, k8 = (lane >> 5) * 8;
    const int kw = n >> 2, o = n & 3;
    bf16x8_t out;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = ks * 16 + k8 + e;
        const float v = n < 28 ? wh[((size_t)(kh * 7 + kw) * 64 + c) * 4 + o] : 0.f;
        const __bf16 hi = (__bf16)v;
        out[e] = plane == 0 ? hi : (__bf16)(v - (float)hi);
    }
    wfrag[i] = __builtin_bit_cast(uint4, out);
}

struct HeadsGeom {
    int strips, bands, band_rows;
};

struct RowRegs {
    float4 raw[2 * HS_KS];   // the lane's 8 channels of every k-step of one input row (fp32, as stored)
};

__global__ __launch_bounds__(64) void heads_bf16x3_kernel(const HeadsArgs a, const uint4 *__restrict__ wfrag,
                                                          const HeadsGeom g)
{
    __shared__ float2 s_ss[64];
    __shared__ float s_u[HS_COLS * HS_UP];
    __shared__ uint4 s_bl[HS_NF * 64];   // lo fragments of B, [fragment][lane]
    const int lane = threadIdx.x;
    const int strip = blockIdx.x % g.strips;
    const int band = (blockIdx.x / g.strips) % g.bands;
    const int n = blockIdx.x / (g.strips * g.bands);
    const int xs0 = strip * HS_OUT;
    const int y0 = band * g.band_rows;
    const int y1 = min(y0 + g.band_rows, a.H);
    if (y0 >= y1) return;

    s_ss[lane] = a.scale_shift[(size_t)n * 64 + lane];
    bf16x8_t bh[HS_NF];
#pragma unroll
    for (int f = 0; f < HS_NF; ++f) {
        bh[f] = __builtin_bit_cast(bf16x8_t, wfrag[(f * 2 + 0) * 64 + lane]);
        s_bl[f * 64 + lane] = wfrag[(f * 2 + 1) * 64 + lane];
    }
    __syncthreads();

    const int m = lane & 31, kh2 = lane >> 5;
    const int xp = xs0 - 3 + m;                                // staged column of this lane's A rows
    const bool col_ok = xp >= 0 && xp < a.W;
    const float *xin = a.x + ((size_t)n * a.H * a.W + (col_ok ? xp : 0)) * 64 + kh2 * 8;
    const size_t row_stride = (size_t)a.W * 64;

    constexpr int RING = 8;
    f32x16 acc[RING];
#pragma unroll
    for (int s = 0; s < RING; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;

    // Software pipeline (a wave is alone on its SIMD: only its own instruction stream can overlap the matrix pipe, and
    // every s_waitcnt is dead time -- a first version that loaded the background value right before the blend and read
    // each B-lo fragment right before its MFMA sat in s_waitcnt 47 % of the time).  Step `it` (input row r = y0 - 3 + it):
    //     at the top: loads of input row r + 2 (into the raw buffer row r has left) and of the background pixels the
    //                 step will blend at its end; accumulator tile of the output row finished by step it - 1 -> LDS;
    //     four k-step blocks: 7 B-lo fragment reads (LDS) issued ahead of the 21 MFMAs that cover their latency
    //                 (lo*hi, hi*hi, then hi*lo), then the same k-step of row r + 1 is converted IN PLACE (raw fp32 ->
    //                 normalise, ReLU, bf16 hi / lo), its fragment registers being dead by then;
    //     between them: horizontal sum + activations of the finished output row; plain stores at the end.
    // Rows outside the image take the same path with their fragments forced to zero (no branch splits the block);
    // loads clamp their address instead of being predicated.
    struct Frag { bf16x8_t hi[HS_KS], lo[HS_KS]; };
    RowRegs raw[2];
    Frag frag;
    auto load_row = [&](int r, RowRegs &dst) {
        const int rc = min(max(r, 0), a.H - 1);
        const float *p = xin + (size_t)rc * row_stride;
#pragma unroll
        for (int ks = 0; ks < HS_KS; ++ks) {
            dst.raw[2 * ks] = *reinterpret_cast<const float4 *>(p + ks * 16);
            dst.raw[2 * ks + 1] = *reinterpret_cast<const float4 *>(p + ks * 16 + 4);
        }
    };
    auto convert = [&](int r, const RowRegs &src, int ks, Frag &f) {
        const bool ok = col_ok && r >= 0 && r < a.H;   // zero padding applies to the NORMALISED activation
        const float4 q0 = src.raw[2 * ks], q1 = src.raw[2 * ks + 1];
        const float raw8[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        const float4 *ssp = reinterpret_cast<const float4 *>(s_ss + ks * 16 + kh2 * 8);
        bf16x8_t h, l;
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
            const float4 ss2 = ssp[e2];   // (scale, shift) of channels 2*e2, 2*e2 + 1
            const float v0 = fmaxf(fmaf(raw8[2 * e2], ss2.x, ss2.y), 0.f);
            const float v1 = fmaxf(fmaf(raw8[2 * e2 + 1], ss2.z, ss2.w), 0.f);
            const __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1;
            h[2 * e2] = h0;
            h[2 * e2 + 1] = h1;
            l[2 * e2] = (__bf16)(v0 - (float)h0);
            l[2 * e2 + 1] = (__bf16)(v1 - (float)h1);
        }
        const uint4 hz = __builtin_bit_cast(uint4, h), lz = __builtin_bit_cast(uint4, l);
        f.hi[ks] = __builtin_bit_cast(bf16x8_t, ok ? hz : make_uint4(0u, 0u, 0u, 0u));
        f.lo[ks] = __builtin_bit_cast(bf16x8_t, ok ? lz : make_uint4(0u, 0u, 0u, 0u));
    };

    const size_t hw = (size_t)a.H * a.W;
    const int xl = lane & 31, half = lane >> 5;               // half 0: colour 0, 1;  half 1: colour 2, mask
    const int oxc = min(xs0 + min(xl, HS_OUT - 1), a.W - 1);  // clamped output column (address of the background prefetch)
    // background pixels of one output row for this lane's two channels (channel 3 is the mask: reads channel 2's, unused)
    const float *bgp = a.pred ? a.bg + (size_t)(a.bg_bs > 1 ? n : 0) * 3 * hw : a.x;
    struct Bg { float v[2]; };
    auto load_bg = [&](int y) {
        const size_t p = (size_t)min(max(y, 0), a.H - 1) * a.W + oxc;
        Bg b;
        b.v[0] = bgp[(size_t)(half * 2) * hw + p];
        b.v[1] = bgp[(size_t)(half ? 2 : 1) * hw + p];
        return b;
    };
    struct Out { float c0, c1, msk; };
    auto tile_to_lds = [&](const f32x16 &u) {
        const int col = lane & 31, rsel = 4 * (lane >> 5);   // C/D layout: column n = lane & 31, row as below
#pragma unroll
        for (int r = 0; r < 16; ++r) s_u[((r & 3) + 8 * (r >> 2) + rsel) * HS_UP + col] = u[r];
    };
    auto reduce_row = [&]() {
        const int xr = min(xl, HS_OUT - 1);
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int kw = 0; kw < 7; ++kw) {
            s0 += s_u[(xr + kw) * HS_UP + kw * 4 + half * 2];
            s1 += s_u[(xr + kw) * HS_UP + kw * 4 + half * 2 + 1];
        }
        // tanh(x) = 1 - 2 / (1 + e^{2x}), sigmoid(x) = 1 / (1 + e^{-x}) on the hardware exp2 / rcp (~1e-6 absolute)
        const float e0 = __expf(2.f * s0);
        const float e1 = __expf(half ? -s1 : 2.f * s1);
        const float r0 = __builtin_amdgcn_rcpf(1.f + e0), r1 = __builtin_amdgcn_rcpf(1.f + e1);
        Out o;
        o.c0 = fmaf(-2.f, r0, 1.f);
        o.c1 = half ? r1 : fmaf(-2.f, r1, 1.f);
        o.msk = __shfl(o.c1, 32 + xl);                        // the mask sits in the upper half's c1
        return o;
    };
    auto store_row = [&](const Out &o, const Bg &b, int y) {
        const int ox = xs0 + xl;
        if (y < y0 || y >= y1 || xl >= HS_OUT || ox >= a.W) return;
        const size_t p = (size_t)y * a.W + ox;
        const float cv[2] = {o.c0, o.c1};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int c = half * 2 + k;
            if (c == 3) {
                if (a.mask) a.mask[(size_t)n * hw + p] = o.msk;
                continue;
            }
            if (a.color) a.color[((size_t)n * 3 + c) * hw + p] = cv[k];
            if (a.pred) a.pred[((size_t)n * 3 + c) * hw + p] = o.msk * b.v[k] + (1.f - o.msk) * cv[k];
        }
    };

    const int niter = y1 - y0 + 6;   // input rows y0-3 .. y1+2
    load_row(y0 - 3, raw[0]);
    load_row(y0 - 2, raw[1]);
#pragma unroll
    for (int ks = 0; ks < HS_KS; ++ks) convert(y0 - 3, raw[0], ks, frag);

    // PH = (iteration index) mod 8 fixes every ring position and the buffer parity at compile time
    auto row_step = [&](int it, auto ph_c) {
        constexpr int PH = decltype(ph_c)::value;
        constexpr int PREV = (PH + RING - 1) % RING;
        const int r = y0 - 3 + it;
        const Bg bgv = load_bg(r - 4);
        load_row(r + 2, raw[PH & 1]);
        // the output row the PREVIOUS step completed (y = r - 4): its ring slot is not touched by this step's MFMAs
        tile_to_lds(acc[PREV]);
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[PREV][q] = 0.f;
        Out o;
#pragma unroll
        for (int ks = 0; ks < HS_KS; ++ks) {
            bf16x8_t bl[7];
#pragma unroll
            for (int kh = 0; kh < 7; ++kh) bl[kh] = __builtin_bit_cast(bf16x8_t, s_bl[(kh * HS_KS + ks) * 64 + lane]);
#pragma unroll
            for (int t = 0; t < 3; ++t)   // lo*hi, hi*hi, hi*lo: the LDS-fed product last
#pragma unroll
                for (int kh = 0; kh < 7; ++kh) {
                    const int slot = (PH - kh + 6 + RING) % RING;   // output row r - kh + 3
                    const bf16x8_t av = t == 0 ? frag.lo[ks] : frag.hi[ks];
                    const bf16x8_t bv = t == 2 ? bl[kh] : bh[kh * HS_KS + ks];
                    acc[slot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[slot], 0, 0, 0);
                }
            if (ks == 1) o = reduce_row();
            convert(r + 1, raw[(PH + 1) & 1], ks, frag);
            // one MFMA, then a few of the independent vector / LDS instructions of this block
#pragma unroll
            for (int g2 = 0; g2 < 21; ++g2) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        store_row(o, bgv, r - 4);
    };

    for (int it = 0; it < niter; it += RING) {
        row_step(it, std::integral_constant<int, 0>{});
        if (it + 1 < niter) row_step(it + 1, std::integral_constant<int, 1>{});
        if (it + 2 < niter) row_step(it + 2, std::integral_constant<int, 2>{});
        if (it + 3 < niter) row_step(it + 3, std::integral_constant<int, 3>{});
        if (it + 4 < niter) row_step(it + 4, std::integral_constant<int, 4>{});
        if (it + 5 < niter) row_step(it + 5, std::integral_constant<int, 5>{});
        if (it + 6 < niter) row_step(it + 6, std::integral_constant<int, 6>{});
        if (it + 7 < niter) row_step(it + 7, std::integral_constant<int, 7>{});
    }
    // the last step's output row (y1 - 1) is still in its ring slot
    {
        const int last = niter - 1, y = y0 - 3 + last - 3;
        const Bg bgv = load_bg(y);
        switch (last % RING) {
            case 0: tile_to_lds(acc[0]); break;
            case 1: tile_to_lds(acc[1]); break;
            case 2: tile_to_lds(acc[2]); break;
            case 3: tile_to_lds(acc[3]); break;
            case 4: tile_to_lds(acc[4]); break;
            case 5: tile_to_lds(acc[5]); break;
            case 6: tile_to_lds(acc[6]); break;
            default: tile_to_lds(acc[7]); break;
        }
        const Out o = reduce_row();
        store_row(o, bgv, y);
    }
}

}  // namespace

size_t heads_bf16x3_frag_bytes() { return (size_t)HS_NF * 2 * 64 * 16; }

int launch_heads_pack(const float *wh, void *wfrag, hipStream_t st)
{
    heads_pack_kernel<<<ceil_div(HS_NF * 2 * 64, 256), 256, 0, st>>>(wh, static_cast<uint4 *>(wfrag));
    LWG_LAUNCH_CHECK("heads_pack_kernel");
    return LWG_OK;
}

int launch_heads_bf16x3(const HeadsArgs &a, const void *wfrag, hipStream_t st)
{
    if (a.pred && !a.bg) LWG_FAIL(LWG_ERR_INVALID_ARG, "heads: pred requested without a background image");
    if (!wfrag) LWG_FAIL(LWG_ERR_STATE, "heads: packed weight fragments missing");
    HeadsGeom g;
    g.strips = ceil_div(a.W, HS_OUT);
    // one wave per SIMD slot: aim at 4 waves per CU over the whole grid, bands of at least 8 output rows
    const int target = 4 * device_cu_count();
    int bands = target / (g.strips * a.N);
    if (bands < 1) bands = 1;
    if (bands > ceil_div(a.H, 8)) bands = ceil_div(a.H, 8);
    g.band_rows = ceil_div(a.H, bands);
    g.bands = ceil_div(a.H, g.band_rows);
    heads_bf16x3_kernel<<<g.strips * g.bands * a.N, 64, 0, st>>>(a, static_cast<const uint4 *>(wfrag), g);
    LWG_LAUNCH_CHECK("heads_bf16x3_kernel");
    return LWG_OK;
}

}  // namespace lwg
