// train.hip -- first slice of the training step: the PatchGAN discriminator update.
//
// Replaces, for the discriminator, what PyTorch autograd + cuDNN + torch.optim.Adam execute in the reference's
// ImpersonatorTrainer._optimize_D (models/impersonator_trainer.py:396-411) on a PatchDiscriminator
// (networks/discriminator.py:8-57: conv4x4 s2 + LeakyReLU, n-1 x [conv4x4 s2 + InstanceNorm + LeakyReLU],
// conv4x4 s1 + InstanceNorm + LeakyReLU, conv4x4 s1 -> 1 channel) with the LSGAN targets of
// _compute_loss_D (:413-414):  loss = mean((D(real) - 1)^2) + mean((D(fake) + 1)^2).
//
// Real and fake images run as ONE batch of 2N (InstanceNorm has no cross-sample statistics, so this is the same
// arithmetic as two forward calls).  Forward convs and data gradients run on the fp32 MFMA implicit GEMM of conv.hip in
// its general mode (a stride-2 conv's data gradient is a 4-phase transposed conv, a stride-1 conv's a conv with the
// flipped kernel; both read weight matrices derived on the device from the master copy after every optimiser step);
// the weight gradient is its own MFMA kernel below.  Parameters, gradients and Adam moments live in flat device
// buffers in the forward-matrix layout [Cout][tap][Cin]; the gradient buffer is what a data-parallel job all-reduces
// (27.8 MB for the reference's n_layers = 4, ndf = 64), through torch.distributed/RCCL on the Python side.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "conv.h"
#include "split.h"
#include "sample.h"

namespace lwg {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr float kLeaky = 0.2f;
constexpr float kEps = 1e-5f;

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// ---- InstanceNorm statistics: (mean, 1/sqrt(var + eps)) per (image, channel) of an NHWC tensor, biased variance.
// grid (C/16, N), 256 threads = 16 pixel slices x 16 channels (64-byte rows); double accumulation, fixed reduction order.
constexpr int RS_CH = 16, RS_SL = 16;
__global__ __launch_bounds__(256) void in_stats_kernel(const float *__restrict__ x, int HW, int C, float2 *__restrict__ out)
{
    __shared__ double sh[2][RS_SL][RS_CH];
    const int cl = threadIdx.x & (RS_CH - 1), c = blockIdx.x * RS_CH + cl, slice = threadIdx.x / RS_CH, n = blockIdx.y;
    const float *p = x + (size_t)n * HW * C + c;
    double s = 0., q = 0.;
    for (int i = slice; i < HW; i += RS_SL) {
        const double v = p[(size_t)i * C];
        s += v;
        q += v * v;
    }
    sh[0][slice][cl] = s;
    sh[1][slice][cl] = q;
    __syncthreads();
    if (slice == 0) {
        s = q = 0.;
        for (int k = 0; k < RS_SL; ++k) {
            s += sh[0][k][cl];
            q += sh[1][k][cl];
        }
        const double mean = s / HW, var = fmax(q / HW - mean * mean, 0.);
        out[(size_t)n * C + c] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)kEps)));
    }
}

// ---- activation: y = leaky((x - mean) * rstd) (stats != null) or leaky(x); float4 over channels
__global__ __launch_bounds__(256) void d_act_kernel(const float *__restrict__ raw, const float2 *__restrict__ stats, int HW,
                                                    int C, long total4, float *__restrict__ y)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const long e = i * 4;
    const int c = (int)(e % C);
    const int n = (int)(e / ((long)HW * C));
    float4 v = ld4(raw + e);
    float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (stats) {
            const float2 s = stats[(size_t)n * C + c + k];
            f[k] = (f[k] - s.x) * s.y;
        }
        f[k] = f[k] > 0.f ? f[k] : kLeaky * f[k];
    }
    *reinterpret_cast<float4 *>(y + e) = make_float4(f[0], f[1], f[2], f[3]);
}

// ---- LSGAN loss and its gradient on the 1-channel patch map (channel 0 of a Cp-channel NHWC tensor).
// Images [0, N) are real (target +1), [N, 2N) fake (target -1).  One workgroup; fixed-order reduction.
// halves = 2: images [0,N) target t0, [N,2N) target t1, loss = mean over each half, summed; halves = 1: N images, target t0
__global__ __launch_bounds__(256) void lsgan_kernel(const float *__restrict__ out, int N, int HW, int Cp,
                                                    float *__restrict__ dout, float *__restrict__ loss, int halves, float t0,
                                                    float t1)
{
    __shared__ double sh[256];
    const int per_half = N * HW;
    double acc = 0.;
    for (int i = threadIdx.x; i < halves * per_half; i += 256) {
        const float y = i < per_half ? t0 : t1;
        const float d = out[(size_t)i * Cp] - y;
        acc += (double)d * d;
        float *g = dout + (size_t)i * Cp;
        g[0] = 2.f * d / (float)per_half;
        for (int c = 1; c < Cp; ++c) g[c] = 0.f;
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(sh[0] / per_half);
}

// ---- backward of [InstanceNorm] + LeakyReLU.  g = dy * leaky'(y);  with norm:
//      dx = rstd * (g - mean_hw(g) - xhat * mean_hw(g * xhat)),  xhat = (x - mean) * rstd.
// pass 1: (sum g, sum g*xhat) per (image, channel); grid (C/16, N), 16 pixel slices x 16 channels
__global__ __launch_bounds__(256) void in_bwd_reduce_kernel(const float *__restrict__ raw, const float *__restrict__ act,
                                                            const float *__restrict__ dact, const float2 *__restrict__ stats,
                                                            int HW, int C, float2 *__restrict__ sums)
{
    __shared__ double sh[2][RS_SL][RS_CH];
    const int cl = threadIdx.x & (RS_CH - 1), c = blockIdx.x * RS_CH + cl, slice = threadIdx.x / RS_CH, n = blockIdx.y;
    const size_t base = (size_t)n * HW * C + c;
    const float2 st = stats[(size_t)n * C + c];
    double s1 = 0., s2 = 0.;
    for (int i = slice; i < HW; i += RS_SL) {
        const size_t o = base + (size_t)i * C;
        const float g = dact[o] * (act[o] > 0.f ? 1.f : kLeaky);
        s1 += g;
        s2 += (double)g * ((raw[o] - st.x) * st.y);
    }
    sh[0][slice][cl] = s1;
    sh[1][slice][cl] = s2;
    __syncthreads();
    if (slice == 0) {
        s1 = s2 = 0.;
        for (int k = 0; k < RS_SL; ++k) {
            s1 += sh[0][k][cl];
            s2 += sh[1][k][cl];
        }
        sums[(size_t)n * C + c] = make_float2((float)(s1 / HW), (float)(s2 / HW));
    }
}
// pass 2 (elementwise); stats == null: no norm, dx = g; act == null: no activation either (dx = dy)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float *__restrict__ raw, const float *__restrict__ act,
                                                      const float *__restrict__ dact, const float2 *__restrict__ stats,
                                                      const float2 *__restrict__ sums, int HW, int C, long total,
                                                      float *__restrict__ draw)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    float g = dact[e];
    if (act) g *= act[e] > 0.f ? 1.f : kLeaky;
    if (stats) {
        const int c = (int)(e % C);
        const int n = (int)(e / ((long)HW * C));
        const float2 st = stats[(size_t)n * C + c], sm = sums[(size_t)n * C + c];
        g = st.y * (g - sm.x - (raw[e] - st.x) * st.y * sm.y);
    }
    draw[e] = g;
}

// ---- column sums (bias gradient): out[c] = sum over P pixels of x[p][c].  Two deterministic stages.
constexpr int CS_SLICES = 64;
__global__ __launch_bounds__(256) void col_sum_partial_kernel(const float *__restrict__ x, long P, int C, float *__restrict__ part)
{
    __shared__ double sh[4][64];
    const int cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, sub = threadIdx.x >> 6, slice = blockIdx.y;
    const long per = (P + CS_SLICES - 1) / CS_SLICES, p0 = slice * per, p1 = p0 + per < P ? p0 + per : P;
    double s = 0.;
    for (long p = p0 + sub; p < p1; p += 4) s += x[(size_t)p * C + c];
    sh[sub][cl] = s;
    __syncthreads();
    if (sub == 0) part[(size_t)slice * C + c] = (float)(sh[0][cl] + sh[1][cl] + sh[2][cl] + sh[3][cl]);
}
__global__ __launch_bounds__(256) void reduce_slices_kernel(const float *__restrict__ part, int S, long n, float *__restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += part[(size_t)k * n + i];
    out[i] = s;
}

__global__ __launch_bounds__(256) void reduce_slices4_kernel(const float4 *__restrict__ part, int S, long n4, float4 *__restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);   // the same sums in the same order as reduce_slices_kernel, four at a time
    for (int k = 0; k < S; ++k) {
        const float4 v = part[(size_t)k * n4 + i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    out[i] = s;
}

// partials [slice][tap][co][ci] (wgrad_row3_bf16x3_kernel) -> out in the (co, ci, tap) or (co, tap, ci) layout; slices summed in
// order.  A block owns 64 consecutive (co, ci) pairs: thread (pair, tl) sums taps tl, tl + 4, tl + 8 -- coalesced reads along the
// pairs -- and the 64 x 9 sums leave through LDS as one contiguous run of the (co, ci, tap) layout.
constexpr int kReduceTapsMax = 49;
__global__ __launch_bounds__(256) void reduce_taps_kernel(const float *__restrict__ part, int S, int Cout, int Cin, int ntaps, int oihw,
                                                          float *__restrict__ out)
{
    __shared__ float sm[64 * kReduceTapsMax];
    const long n = (long)Cout * Cin, pair0 = (long)blockIdx.x * 64;
    const int pl = threadIdx.x & 63, tl = threadIdx.x >> 6;
    const long pair = pair0 + pl;
    for (int t = tl; t < ntaps; t += 4) {
        float sum = 0.f;
        if (pair < n)
            for (int k = 0; k < S; ++k) sum += part[((size_t)k * ntaps + t) * n + pair];
        if (oihw) sm[pl * ntaps + t] = sum;
        else if (pair < n) out[((size_t)(pair / Cin) * ntaps + t) * Cin + pair % Cin] = sum;
    }
    if (!oihw) return;
    __syncthreads();
    const long total = n * ntaps;
    for (int i = threadIdx.x; i < 64 * ntaps; i += 256)
        if (pair0 * ntaps + i < total) out[pair0 * ntaps + i] = sm[i];
}

// the same for small filters (the stem: 64 x 8 pairs, 49 taps, ~150 slices): one thread per (tap, pair), four partial sums
// (slices k = 0, 4, 8, ... / 1, 5, ... / ...) added in a fixed order
__global__ __launch_bounds__(256) void reduce_taps_direct_kernel(const float *__restrict__ part, int S, int Cout, int Cin, int ntaps,
                                                                 int oihw, float *__restrict__ out)
{
    const long n = (long)Cout * Cin, e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * ntaps) return;
    const int t = (int)(e / n);
    const long pair = e - (long)t * n;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const float *src = part + (size_t)t * n + pair;
    const size_t pitch = (size_t)ntaps * n;
    int k = 0;
    for (; k + 3 < S; k += 4) {
        s0 += src[(size_t)k * pitch];
        s1 += src[(size_t)(k + 1) * pitch];
        s2 += src[(size_t)(k + 2) * pitch];
        s3 += src[(size_t)(k + 3) * pitch];
    }
    for (; k < S; ++k) s0 += src[(size_t)k * pitch];
    const float sum = (s0 + s1) + (s2 + s3);
    out[oihw ? pair * ntaps + t : ((size_t)(pair / Cin) * ntaps + t) * Cin + pair % Cin] = sum;
}

// ---- weight gradient on the fp32 matrix cores:  dW[co][tap][ci] = sum_p dY[p][co] * X[pix(p, tap)][ci].
// Both operands are "reduction index (pixel) x contiguous channels" in memory -- exactly what v_mfma_f32_32x32x2_f32
// wants from one ds_read_b32 per lane (A: lane -> channel co, k = pixel pair; B: lane -> channel ci), so the tiles go
// to LDS as loaded, no transpose.  Workgroup = 64 co x 64 ci of one tap over a slice of the pixels, four waves of
// 32 x 32; the slices' partial results are summed in a fixed order by reduce_slices_kernel (deterministic).
constexpr int WG_PX = 32;
constexpr long kWgradMaxSlices = 256;   // split-K bound (the partial buffer may hold fewer)
// KWd x (ntaps / KWd) filter taps; oihw = 0: out[slice][Cout][ntaps][Cin] (matrix layout), 1: out[slice][Cout][Cin][ntaps].
// T = 64: four waves of one 32x32 tile; T = 128 (both channel counts multiples of 128): four waves of 2x2 tiles -- four
// times the MFMA work per byte staged through LDS.
template <int T>
__global__ __launch_bounds__(256) void wgrad_kernel(const float *__restrict__ dy, int Cout, const float *__restrict__ x,
                                                    int Cin, int N, int H, int W, int Ho, int Wo, int stride, int pad,
                                                    long px_per_slice, float *__restrict__ out, int KWd, int ntaps, int oihw)
{
    constexpr int PITCH = T + 4, TW = T / 64, NV = T / 32;   // LDS row pitch; MFMA tiles per wave and side; float4 per thread and operand
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float *sA = wsm, *sB = wsm + 2 * WG_PX * PITCH;          // [2][WG_PX][PITCH] each
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ci0 = blockIdx.x * T, co0 = blockIdx.y * T;
    const int tap = blockIdx.z % ntaps, slice = blockIdx.z / ntaps;
    const int kh = tap / KWd, kw = tap - kh * KWd;
    const long P = (long)N * Ho * Wo;
    const long p0 = slice * px_per_slice, p1 = p0 + px_per_slice < P ? p0 + px_per_slice : P;

    const int lrow = tid >> 3, lcol = (tid & 7) * 4;     // loader: 32 pixel rows x 8 float4, NV column groups of 32
    float4 ra[NV], rb[NV];
    auto load = [&](long pc) {
        const long p = pc + lrow;
#pragma unroll
        for (int v = 0; v < NV; ++v) ra[v] = rb[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p < p1) {
            const float *d = dy + (size_t)p * Cout + co0 + lcol;
#pragma unroll
            for (int v = 0; v < NV; ++v) ra[v] = ld4(d + 32 * v);
            const int n = (int)(p / ((long)Ho * Wo));
            const int rem = (int)(p - (long)n * Ho * Wo);
            const int oh = rem / Wo, ow = rem - oh * Wo;
            const int ih = oh * stride - pad + kh, iw = ow * stride - pad + kw;
            if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
                const float *s = x + (((size_t)n * H + ih) * W + iw) * Cin + ci0 + lcol;
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    if (ci0 + lcol + 32 * v < Cin) rb[v] = ld4(s + 32 * v);
            }
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            *reinterpret_cast<float4 *>(sA + (buf * WG_PX + lrow) * PITCH + lcol + 32 * v) = ra[v];
            *reinterpret_cast<float4 *>(sB + (buf * WG_PX + lrow) * PITCH + lcol + 32 * v) = rb[v];
        }
    };
    f32x16 acc[TW][TW];
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int buf = 0;
    if (p0 < p1) {
        load(p0);
        store(0);
    }
    __syncthreads();
    for (long pc = p0; pc < p1; pc += WG_PX) {
        const bool more = pc + WG_PX < p1;
        if (more) load(pc + WG_PX);
#pragma unroll
        for (int kk = 0; kk < WG_PX / 2; ++kk) {
            const float *ar = sA + (buf * WG_PX + 2 * kk + (lane >> 5)) * PITCH + wm * 32 * TW + (lane & 31);
            const float *br = sB + (buf * WG_PX + 2 * kk + (lane >> 5)) * PITCH + wn * 32 * TW + (lane & 31);
            float av[TW], bv[TW];
#pragma unroll
            for (int i = 0; i < TW; ++i) {
                av[i] = ar[32 * i];
                bv[i] = br[32 * i];
            }
#pragma unroll
            for (int i = 0; i < TW; ++i)
#pragma unroll
                for (int j = 0; j < TW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (more) store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // C/D layout: col = lane&31 -> ci, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -> co
    float *o = out + (size_t)slice * Cout * ntaps * Cin;
#pragma unroll
    for (int j = 0; j < TW; ++j) {
        const int ci = ci0 + (wn * TW + j) * 32 + (lane & 31);
        if (ci >= Cin) continue;
#pragma unroll
        for (int i = 0; i < TW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + (wm * TW + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (co < Cout) o[oihw ? ((size_t)co * Cin + ci) * ntaps + tap : ((size_t)co * ntaps + tap) * Cin + ci] = acc[i][j][r];
            }
    }
}

// ---- the same on the bf16 matrix cores (bf16x3: operands carried as hi + lo bf16, products lo*hi + hi*lo + hi*hi with
// fp32 accumulation, conv.h).  v_mfma_f32_32x32x16_bf16 wants 8 consecutive reduction entries -- pixels -- per lane,
// the tensors are channel-contiguous; the transposition happens in registers on the way to LDS: a loader thread
// fetches one channel quad of PXT consecutive pixels (PXT float4 loads, coalesced across the wave's 32 quads), splits
// the values and writes, per channel, the PXT pixels' hi terms and lo terms as one 16-byte (8-byte) LDS store.  The LDS
// image of an operand tile is the inference kernel's: row = channel, 128 bytes = [hi: k 0-7 | 8-15 | 16-23 | 24-31 | lo ...],
// 16-byte slots XOR-swizzled by the row so that fragment reads (32 rows x one slot) and loader writes (rows 4 apart)
// spread over the banks.  Tile, grid, split-K slices and epilogue as wgrad_kernel.
typedef __bf16 bf16x4w_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int wg_swz(int row) { return ((row >> 1) ^ (row >> 4)) & 7; }

template <int T>
__global__ __launch_bounds__(256, 2) void wgrad_bf16x3_kernel(const float *__restrict__ dy, int Cout, const float *__restrict__ x,
                                                           int Cin, int N, int H, int W, int Ho, int Wo, int stride, int pad,
                                                           long px_per_slice, float *__restrict__ out, int KWd, int ntaps, int oihw)
{
    constexpr int TW = T / 64, PXT = T / 16, QN = T / 4;   // tiles per wave and side; pixels per loader item; channel quads
    constexpr int OPB = T * 128;                           // bytes of one operand tile (T rows x 32 pixels x (hi, lo))
    extern __shared__ __attribute__((aligned(16))) char wsb[];   // [2 buffers][2 operands][T][128]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ci0 = blockIdx.x * T, co0 = blockIdx.y * T;
    const int tap = blockIdx.z % ntaps, slice = blockIdx.z / ntaps;
    const int kh = tap / KWd, kw = tap - kh * KWd;
    const long P = (long)N * Ho * Wo;
    const long p0 = slice * px_per_slice, p1 = p0 + px_per_slice < P ? p0 + px_per_slice : P;

    // loader item: operand (waves 0-1: dY, waves 2-3: X), channel quad q, pixel block blk of the 32-pixel step.  Both
    // operands run the same code: dY is "a conv operand with one tap, stride 1, no padding".  The item's first pixel
    // (pl -> image ln, output position loh, low) walks ahead of the MFMAs by two steps; element offsets are 32-bit.
    const int op = tid >> 7, idx = tid & 127, q = idx % QN, blk = idx / QN;
    const float *lbase = op ? x : dy;
    const unsigned lC = op ? (unsigned)Cin : (unsigned)Cout, lcoff = (op ? ci0 : co0) + 4 * q;
    const int ls = op ? stride : 1, ldh = op ? kh - pad : 0, ldw = op ? kw - pad : 0;
    const int lH = op ? H : Ho, lW = op ? W : Wo;
    const bool lchan = lcoff < lC;
    const unsigned lsC = (unsigned)ls * lC;
    long pl = p0 + blk * PXT;
    int ln = (int)(pl / ((long)Ho * Wo));
    int loh, low;
    {
        const int rem = (int)(pl - (long)ln * Ho * Wo);
        loh = rem / Wo;
        low = rem - loh * Wo;
    }
    auto load = [&](float4 (&r)[PXT]) {
        int tn = ln, toh = loh, tow = low;
        int ih = toh * ls + ldh, iw = tow * ls + ldw;
        unsigned off = ((unsigned)(tn * lH + ih) * (unsigned)lW + (unsigned)iw) * lC + lcoff;
#pragma unroll
        for (int i = 0; i < PXT; ++i) {
            const bool ok = lchan && pl + i < p1 && (unsigned)ih < (unsigned)lH && (unsigned)iw < (unsigned)lW;
            const float4 v = ld4(lbase + (ok ? off : 0u));
            r[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            ++tow;
            iw += ls;
            off += lsC;
            if (tow == Wo) {
                tow = 0;
                iw = ldw;
                ++toh;
                ih += ls;
                if (toh == Ho) {
                    toh = 0;
                    ih = ldh;
                    ++tn;
                }
                off = ((unsigned)(tn * lH + ih) * (unsigned)lW + (unsigned)iw) * lC + lcoff;
            }
        }
        pl += WG_PX;
        low += WG_PX;
        while (low >= Wo) {
            low -= Wo;
            if (++loh == Ho) {
                loh = 0;
                ++ln;
            }
        }
    };
    auto store = [&](const float4 (&r)[PXT], int buf) {
        char *base = wsb + (buf * 2 + op) * OPB;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const int row = 4 * q + ch, sw = wg_swz(row);
            __bf16 h[PXT], l[PXT];
#pragma unroll
            for (int i = 0; i < PXT; ++i) {
                const float v = ch == 0 ? r[i].x : ch == 1 ? r[i].y : ch == 2 ? r[i].z : r[i].w;
                h[i] = (__bf16)v;
                l[i] = (__bf16)(v - (float)h[i]);
            }
            if constexpr (PXT == 8) {
                bf16x8_t hv, lv;
#pragma unroll
                for (int i = 0; i < 8; ++i) { hv[i] = h[i]; lv[i] = l[i]; }
                *reinterpret_cast<bf16x8_t *>(base + row * 128 + ((blk ^ sw) * 16)) = hv;
                *reinterpret_cast<bf16x8_t *>(base + row * 128 + (((4 + blk) ^ sw) * 16)) = lv;
            } else {
                bf16x4w_t hv, lv;
#pragma unroll
                for (int i = 0; i < 4; ++i) { hv[i] = h[i]; lv[i] = l[i]; }
                *reinterpret_cast<bf16x4w_t *>(base + row * 128 + (((blk >> 1) ^ sw) * 16) + (blk & 1) * 8) = hv;
                *reinterpret_cast<bf16x4w_t *>(base + row * 128 + (((4 + (blk >> 1)) ^ sw) * 16) + (blk & 1) * 8) = lv;
            }
        }
    };
    f32x16 acc[TW][TW];
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TW; ++j)
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) acc[i][j][r2] = 0.f;
    // fragment rows of this lane and their swizzles
    int arow[TW], brow[TW];
#pragma unroll
    for (int i = 0; i < TW; ++i) {
        arow[i] = (wm * TW + i) * 32 + (lane & 31);
        brow[i] = (wn * TW + i) * 32 + (lane & 31);
    }
    auto frag = [&](int buf, int opi, int row, int col) {
        return *reinterpret_cast<const bf16x8_t *>(wsb + (buf * 2 + opi) * OPB + row * 128 + ((col ^ wg_swz(row)) * 16));
    };

    auto compute = [&](int buf) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int ch = 2 * kb + (lane >> 5);
            bf16x8_t ah[TW], al[TW], bh[TW], bl[TW];
#pragma unroll
            for (int i = 0; i < TW; ++i) {
                ah[i] = frag(buf, 0, arow[i], ch);
                al[i] = frag(buf, 0, arow[i], 4 + ch);
                bh[i] = frag(buf, 1, brow[i], ch);
                bl[i] = frag(buf, 1, brow[i], 4 + ch);
            }
            // small terms first; tiles inside a product, so consecutive MFMAs do not share an accumulator
#pragma unroll
            for (int i = 0; i < TW; ++i)
#pragma unroll
                for (int j = 0; j < TW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TW; ++i)
#pragma unroll
                for (int j = 0; j < TW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TW; ++i)
#pragma unroll
                for (int j = 0; j < TW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    };
    // step k multiplies LDS buffer k&1 while the loads of step k+2 fly into one register set and step k+1, already
    // in the other, is split and written to the other buffer behind the MFMAs
    const int K = (int)((p1 - p0 + WG_PX - 1) / WG_PX);
    float4 r0[PXT], r1[PXT];
    if (K > 0) {
        load(r0);
        store(r0, 0);
        if (K > 1) load(r1);
    }
    __syncthreads();
    auto step = [&](float4 (&rload)[PXT], const float4 (&rnext)[PXT], int k) {
        if (k + 2 < K) load(rload);
        compute(k & 1);
        if (k + 1 < K) store(rnext, (k + 1) & 1);
        __syncthreads();
    };
    for (int k = 0; k < K; k += 2) {
        step(r0, r1, k);
        if (k + 1 < K) step(r1, r0, k + 1);
    }
    float *o = out + (size_t)slice * Cout * ntaps * Cin;
#pragma unroll
    for (int j = 0; j < TW; ++j) {
        const int ci = ci0 + (wn * TW + j) * 32 + (lane & 31);
        if (ci >= Cin) continue;
#pragma unroll
        for (int i = 0; i < TW; ++i)
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) {
                const int co = co0 + (wm * TW + i) * 32 + (r2 & 3) + 8 * (r2 >> 2) + 4 * (lane >> 5);
                if (co < Cout) o[oihw ? ((size_t)co * Cin + ci) * ntaps + tap : ((size_t)co * ntaps + tap) * Cin + ci] = acc[i][j][r2];
            }
    }
}

__device__ __forceinline__ float quad_ch(const float4 &v, int ch) { return ch == 0 ? v.x : ch == 1 ? v.y : ch == 2 ? v.z : v.w; }

// ---- 3x3 / stride 1 / pad 1 layers (the trunk and the skippers: nine tenths of the generator's weight-gradient arithmetic):
// ONE KERNEL ROW -- three taps -- per workgroup.  wgrad_bf16x3_kernel streams both operand tiles once per tap, every tile
// of the other operand and every slice: a 128x128 tile gets 32 flop per byte it loads, the working set (X and dY, 16 MiB on
// the trunk at batch 4) does not fit an XCD's 4 MiB L2, and the launch runs at the fabric's rate (594 MB in 121 us) with
// the matrix pipe a fifth busy.  Here
//   * a step is 32 pixels of ONE output row (Wo % 32 == 0): dY's tile is loaded and split once for the three taps, X's
//     once with a one-pixel apron -- a loader thread fetches pixels 8*blk-1 .. 8*blk+8 of its channels and writes the
//     three 8-pixel windows the taps need (the MFMA's reduction index must be the same pixel in both operands, so each
//     tap gets its own shifted copy in LDS); a 128 x 128 tile loads 36 KB per 3 x 1.05 Mflop instead of 96 KB (87 flop per byte);
//   * work items are ordered (slice, co tile, ci tile, kernel row) and dealt to the XCDs in contiguous runs (a 1-D grid is
//     dispatched round-robin over the eight XCDs), so an XCD's workgroups share a pixel slice and its L2 holds what
//     they re-read;
//   * the LDS image is wgrad_bf16x3_kernel's, with bit 0 of the row index XORed by the loader's channel-group parity: a
//     loader's 16 lanes write rows C apart -- all even or all odd, i.e. one half of the banks -- and the flip sends every
//     other one to the other half;
//   * loader items are sized per operand so that every SIMD converts: CA / CB channels x 8 (+2) pixels per thread,
//     (TA / CA + TB / CB) * 4 items;
//   * the big tiles run EIGHT waves (128 x 128 x 3 taps needs 128 KiB of LDS, one workgroup per CU): two waves per SIMD,
//     one splitting dY and one splitting X, each other's conversions under each other's MFMAs.  (Four waves of a
//     128 x 64 tile, 80 KiB, do not get a second workgroup beside them: 1.8 us per step alone against 0.64 of MFMAs.)
//     Measured (profiles/r04_wgrad_bench.md, r04_wgrad_pmc_lds.md): trunk 148 -> 79 us, no LDS bank conflicts, LDS 0.3 active;
//     a step still takes 2.1 us for 1.3 us of MFMAs -- the barrier per 32 pixels with one workgroup per CU.
// TA x TB = co x ci tile on WGM x WGN waves; accumulators 3 x (TA / 32 WGM) x (TB / 32 WGN) x 16 per lane.

template <int TA, int TB, int CA, int CB, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN, (2 * (TA + 3 * TB) * 256 <= 160 * 1024) ? 2 : 1)
void wgrad_row3_bf16x3_kernel(const float *__restrict__ dy, int Cout, const float *__restrict__ x, int Cin, int N, int H, int W,
                              long px_per_slice, int S, float *__restrict__ out, int oihw)
{
    constexpr int NT = 64 * WGM * WGN;                    // WGM x WGN waves of (TA / WGM) x (TB / WGN) outputs per tap
    constexpr int TWA = TA / (32 * WGM), TWB = TB / (32 * WGN);
    constexpr int QA = TA / CA, QB = TB / CB;             // channel groups per operand tile
    constexpr int SA = CA == 4 ? 2 : 1, SB = CB == 4 ? 2 : 1;   // log2 of the group sizes
    constexpr int OPA = TA * 128, OPB = TB * 128;         // bytes of an operand tile in LDS
    constexpr int BUF = OPA + 3 * OPB;
    static_assert((CA == 2 || CA == 4) && (CB == 2 || CB == 4) && (QA + QB) * 4 <= NT && (QA * 4) % 64 == 0, "loader items");
    static_assert(TWA >= 1 && TWB >= 1 && 2 * BUF <= 160 * 1024, "tile");
    extern __shared__ __attribute__((aligned(16))) char wsb[];   // [2 buffers][dY | X tap 0 | X tap 1 | X tap 2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;

    // work item of this workgroup: XCD c = blockIdx.x % 8 owns items [c * chunk, (c + 1) * chunk)
    const int tiles_ci = Cin / TB, tiles_co = Cout / TA;
    const int units = tiles_ci * tiles_co * 3 * S, chunk = (units + 7) >> 3;
    const int item = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= chunk || item >= units) return;
    const int kh = item % 3, ci0 = (item / 3) % tiles_ci * TB, co0 = (item / (3 * tiles_ci)) % tiles_co * TA;
    const int slice = item / (3 * tiles_ci * tiles_co);
    const long P = (long)N * H * W;
    const long p0 = slice * px_per_slice, p1 = p0 + px_per_slice < P ? p0 + px_per_slice : P;
    const int K = (int)((p1 - p0) / WG_PX);               // whole steps: P and the slices are multiples of 32 pixels

    // loader roles (wave-uniform): threads [0, 4 QA) one dY item each, [4 QA, 4 QA + 4 QB) one X item, the rest none
    const int role = tid < 4 * QA ? 0 : (tid < 4 * (QA + QB) ? 1 : 2);
    const int idx = role == 1 ? tid - 4 * QA : tid;
    const int q = role == 1 ? idx % QB : idx % QA, blk = role == 1 ? idx / QB : idx / QA;
    // position of the next step to load: image ln, row loh, first column low (a multiple of 32)
    int ln = (int)(p0 / ((long)H * W)), loh, low;
    {
        const int rem = (int)(p0 - (long)ln * H * W);
        loh = rem / W;
        low = rem - loh * W;
    }
    auto advance = [&]() {   // branch-free: a step's loads, MFMAs and conversions stay one scheduling region
        low += WG_PX;
        const bool row_end = low == W;
        low = row_end ? 0 : low;
        loh += row_end ? 1 : 0;
        const bool img_end = loh == H;
        loh = img_end ? 0 : loh;
        ln += img_end ? 1 : 0;
    };
    // r[0..7] (dY: pixels 8*blk .. +7) or r[0..9] (X: pixels 8*blk-1 .. 8*blk+8 of input row loh + kh - 1).
    // Out-of-image pixels (X only: the apron's first / last pixel at the ends of a row, a whole row above or below the image)
    // are fetched from the tensor's first element and zeroed AFTER the split, by masks on the packed pairs (m.x: pixels 0-1,
    // m.y: pixels 2-7, m.z: pixels 8-9) -- nothing between a load and its use one step later touches the registers, so
    // the loads stay in flight across the MFMAs.  The loader role is a template argument of the whole main loop (run_role
    // below), not a branch inside it: with a branch the values meet in phi copies right behind the loads and the in-order
    // memory counter has to be waited down to zero there.
    auto fetch = [&](auto c_tag, const float *src) -> float4 {
        if constexpr (decltype(c_tag)::value == 4) {
            return ld4(src);
        } else {
            const float2 v2 = *reinterpret_cast<const float2 *>(src);
            return make_float4(v2.x, v2.y, 0.f, 0.f);
        }
    };
    auto load = [&](auto role_tag, float4 (&r)[10], uint3 &m) {
        constexpr int ROLE = decltype(role_tag)::value;
        if constexpr (ROLE == 0) {
            unsigned off = ((unsigned)(ln * H + loh) * (unsigned)W + (unsigned)(low + 8 * blk)) * (unsigned)Cout + (unsigned)(co0 + CA * q);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                r[i] = fetch(std::integral_constant<int, CA>{}, dy + off);
                off += (unsigned)Cout;
            }
        } else if constexpr (ROLE == 1) {
            const int ih = loh + kh - 1, iw0 = low + 8 * blk - 1;
            const bool row_ok = (unsigned)ih < (unsigned)H;
            const unsigned base = ((unsigned)(ln * H + (row_ok ? ih : 0)) * (unsigned)W) * (unsigned)Cin + (unsigned)(ci0 + CB * q);
            m.y = row_ok ? 0xffffffffu : 0u;
            m.x = iw0 < 0 ? (m.y & 0xffff0000u) : m.y;
            m.z = iw0 + 9 >= W ? (m.y & 0x0000ffffu) : m.y;
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const bool ok = row_ok && (unsigned)(iw0 + i) < (unsigned)W;
                r[i] = fetch(std::integral_constant<int, CB>{}, x + (ok ? base + (unsigned)(iw0 + i) * (unsigned)Cin : 0u));
            }
        }
        advance();
    };
    // physical LDS row of logical row `row` of a tile written in groups of 1 << cs channels (see above) and the 16-byte slot
    // swizzle of wgrad_bf16x3_kernel on it
    auto lds_at = [&](int buf, int tile_off, int row, int col, int cs) -> char * {
        const int pr = row ^ ((row >> cs) & 1);
        return wsb + buf * BUF + tile_off + pr * 128 + ((col ^ wg_swz(pr)) * 16);
    };
    auto store = [&](auto role_tag, const float4 (&r)[10], const uint3 &m, int buf) {
        constexpr int ROLE = decltype(role_tag)::value;
        if constexpr (ROLE == 0) {
#pragma unroll
            for (int ch = 0; ch < CA; ++ch) {
                uint4 hv, lv;
                split_pair(quad_ch(r[0], ch), quad_ch(r[1], ch), hv.x, lv.x);
                split_pair(quad_ch(r[2], ch), quad_ch(r[3], ch), hv.y, lv.y);
                split_pair(quad_ch(r[4], ch), quad_ch(r[5], ch), hv.z, lv.z);
                split_pair(quad_ch(r[6], ch), quad_ch(r[7], ch), hv.w, lv.w);
                *reinterpret_cast<uint4 *>(lds_at(buf, 0, CA * q + ch, blk, SA)) = hv;
                *reinterpret_cast<uint4 *>(lds_at(buf, 0, CA * q + ch, 4 + blk, SA)) = lv;
            }
        } else if constexpr (ROLE == 1) {
#pragma unroll
            for (int ch = 0; ch < CB; ++ch) {
                // pixels (0,1) (2,3) (4,5) (6,7) (8,9) as packed pairs; the odd-aligned pairs (1,2) .. (7,8) by funnel shifts
                unsigned h0, h1, h2, h3, h4, l0, l1, l2, l3, l4;
                split_pair(quad_ch(r[0], ch), quad_ch(r[1], ch), h0, l0);
                split_pair(quad_ch(r[2], ch), quad_ch(r[3], ch), h1, l1);
                split_pair(quad_ch(r[4], ch), quad_ch(r[5], ch), h2, l2);
                split_pair(quad_ch(r[6], ch), quad_ch(r[7], ch), h3, l3);
                split_pair(quad_ch(r[8], ch), quad_ch(r[9], ch), h4, l4);
                h0 &= m.x; l0 &= m.x; h1 &= m.y; l1 &= m.y; h2 &= m.y; l2 &= m.y; h3 &= m.y; l3 &= m.y; h4 &= m.z; l4 &= m.z;
                // tap kw pairs output pixel j with input pixel j + kw - 1 = r[j + kw]
                const uint4 hv0 = make_uint4(h0, h1, h2, h3), lv0 = make_uint4(l0, l1, l2, l3);
                const uint4 hv1 = make_uint4(__builtin_amdgcn_alignbit(h1, h0, 16), __builtin_amdgcn_alignbit(h2, h1, 16),
                                             __builtin_amdgcn_alignbit(h3, h2, 16), __builtin_amdgcn_alignbit(h4, h3, 16));
                const uint4 lv1 = make_uint4(__builtin_amdgcn_alignbit(l1, l0, 16), __builtin_amdgcn_alignbit(l2, l1, 16),
                                             __builtin_amdgcn_alignbit(l3, l2, 16), __builtin_amdgcn_alignbit(l4, l3, 16));
                const uint4 hv2 = make_uint4(h1, h2, h3, h4), lv2 = make_uint4(l1, l2, l3, l4);
                *reinterpret_cast<uint4 *>(lds_at(buf, OPA, CB * q + ch, blk, SB)) = hv0;
                *reinterpret_cast<uint4 *>(lds_at(buf, OPA, CB * q + ch, 4 + blk, SB)) = lv0;
                *reinterpret_cast<uint4 *>(lds_at(buf, OPA + OPB, CB * q + ch, blk, SB)) = hv1;
                *reinterpret_cast<uint4 *>(lds_at(buf, OPA + OPB, CB * q + ch, 4 + blk, SB)) = lv1;
                *reinterpret_cast<uint4 *>(lds_at(buf, OPA + 2 * OPB, CB * q + ch, blk, SB)) = hv2;
                *reinterpret_cast<uint4 *>(lds_at(buf, OPA + 2 * OPB, CB * q + ch, 4 + blk, SB)) = lv2;
            }
        }
    };
    f32x16 acc[3][TWA][TWB];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < TWA; ++i)
#pragma unroll
            for (int j = 0; j < TWB; ++j)
#pragma unroll
                for (int r2 = 0; r2 < 16; ++r2) acc[t][i][j][r2] = 0.f;
    auto compute = [&](int buf) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int col = 2 * kb + (lane >> 5);
            bf16x8_t ah[TWA], al[TWA];
#pragma unroll
            for (int i = 0; i < TWA; ++i) {
                const int row = (wm * TWA + i) * 32 + (lane & 31);
                ah[i] = *reinterpret_cast<const bf16x8_t *>(lds_at(buf, 0, row, col, SA));
                al[i] = *reinterpret_cast<const bf16x8_t *>(lds_at(buf, 0, row, 4 + col, SA));
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                bf16x8_t bh[TWB], bl[TWB];
#pragma unroll
                for (int j = 0; j < TWB; ++j) {
                    const int row = (wn * TWB + j) * 32 + (lane & 31);
                    bh[j] = *reinterpret_cast<const bf16x8_t *>(lds_at(buf, OPA + t * OPB, row, col, SB));
                    bl[j] = *reinterpret_cast<const bf16x8_t *>(lds_at(buf, OPA + t * OPB, row, 4 + col, SB));
                }
                // small terms first; tiles inside a product, so consecutive MFMAs do not share an accumulator
#pragma unroll
                for (int i = 0; i < TWA; ++i)
#pragma unroll
                    for (int j = 0; j < TWB; ++j) acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[t][i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TWA; ++i)
#pragma unroll
                    for (int j = 0; j < TWB; ++j) acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[t][i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TWA; ++i)
#pragma unroll
                    for (int j = 0; j < TWB; ++j) acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[t][i][j], 0, 0, 0);
            }
        }
    };
    // pipeline as wgrad_bf16x3_kernel: step k multiplies buffer k & 1, the loads of step k + 2 fly, step k + 1 is split and
    // written to the other buffer behind the MFMAs
    // pipeline: step k multiplies LDS buffer k & 1 while the loads of step k + 2 fly into one register set and step k + 1,
    // already in the other, is split and written to the other buffer.  Every wave leaves the barrier in the same phase, so
    // "MFMAs, then conversions" in all of them leaves the matrix pipe idle while a SIMD's waves convert (measured: 3 us per
    // step of 1.3 us of MFMAs).  The X loaders therefore split FIRST and multiply second, the others the other way round:
    // a SIMD of the eight-wave tiles holds one wave of each kind, each converting under the other's MFMAs.
    auto run_role = [&](auto role_tag) {
        constexpr bool SPLIT_FIRST = decltype(role_tag)::value == 1;
        float4 r0[10], r1[10];
        uint3 m0 = make_uint3(0u, 0u, 0u), m1 = m0;
        auto work = [&](const float4 (&rnext)[10], const uint3 &mnext, int kk, bool do_store) {
            if (SPLIT_FIRST) {
                if (do_store) store(role_tag, rnext, mnext, (kk + 1) & 1);
                compute(kk & 1);
            } else {
                compute(kk & 1);
                if (do_store) store(role_tag, rnext, mnext, (kk + 1) & 1);
            }
        };
        int k = 0;
        if (K >= 4) {   // steady state without a condition between a load and its use
            load(role_tag, r0, m0);
            store(role_tag, r0, m0, 0);
            load(role_tag, r1, m1);
            __syncthreads();
            for (; k + 3 < K; k += 2) {
                load(role_tag, r0, m0);
                work(r1, m1, 0, true);
                __syncthreads();
                load(role_tag, r1, m1);
                work(r0, m0, 1, true);
                __syncthreads();
            }
        } else {
            if (K > 0) {
                load(role_tag, r0, m0);
                store(role_tag, r0, m0, 0);
                if (K > 1) load(role_tag, r1, m1);
            }
            __syncthreads();
        }
        auto step = [&](float4 (&rload)[10], uint3 &mload, const float4 (&rnext)[10], const uint3 &mnext, int kk) {
            if (kk + 2 < K) load(role_tag, rload, mload);
            work(rnext, mnext, kk, kk + 1 < K);
            __syncthreads();
        };
        for (; k < K; k += 2) {   // the last two or three steps (k is even: buffer 0 holds step k, r1 step k + 1)
            step(r0, m0, r1, m1, k);
            if (k + 1 < K) step(r1, m1, r0, m0, k + 1);
        }
    };
    // (wave-uniform: every wave meets the same number of barriers whichever copy of the loop it runs)
    if (role == 0) run_role(std::integral_constant<int, 0>{});
    else if (role == 1) run_role(std::integral_constant<int, 1>{});
    else run_role(std::integral_constant<int, 2>{});
    // S == 1: the gradient itself, in the caller's layout.  S > 1: this slice's partial as [tap][co][ci] -- a wave's 32
    // ci lanes store 128 contiguous bytes (in the (co, ci, tap) layout of the result every lane would touch its own cache
    // line: a quarter of this kernel's time went there) -- and reduce_taps_kernel writes the caller's layout.
    float *o = out + (size_t)slice * Cout * 9 * Cin;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int tap = kh * 3 + t;
#pragma unroll
        for (int j = 0; j < TWB; ++j) {
            const int ci = ci0 + (wn * TWB + j) * 32 + (lane & 31);
#pragma unroll
            for (int i = 0; i < TWA; ++i)
#pragma unroll
                for (int r2 = 0; r2 < 16; ++r2) {
                    const int co = co0 + (wm * TWA + i) * 32 + (r2 & 3) + 8 * (r2 >> 2) + 4 * (lane >> 5);
                    const size_t at = S > 1 ? ((size_t)tap * Cout + co) * Cin + ci
                                            : (oihw ? ((size_t)co * Cin + ci) * 9 + tap : ((size_t)co * 9 + tap) * Cin + ci);
                    o[at] = acc[t][i][j][r2];
                }
        }
    }
}

// ---- the 7x7 stem (8 input channels, 6 of them real): wgrad_bf16x3_kernel gives every tap its own workgroup with a 64 x 64
// tile of which 8 columns are channels -- 49 passes over dY, seven eighths of the MFMAs on zeros, 0.55 ms per launch at
// 256 x 256, batch 4.  Here a workgroup owns ONE KERNEL ROW and the tile's 64 columns are (kw, ci): 7 x 8 = 56 used.  As
// wgrad_row3_bf16x3_kernel otherwise (32-pixel steps of one output row, XCD-local work order, partials [tap][co][ci]);
// loader items: waves 0-1 split dY (2 channels x 8 pixels), wave 2 splits X (one channel, 4 pixels + a 6-pixel apron,
// the seven 4-pixel windows of the taps as 8-byte LDS stores), wave 3 only multiplies.  KW x KW taps, pad (KW - 1) / 2.
template <int KW>
__global__ __launch_bounds__(256, 4) void wgrad_rowk_c8_bf16x3_kernel(const float *__restrict__ dy, int Cout, const float *__restrict__ x,
                                                                    int N, int H, int W, long px_per_slice, int S,
                                                                    float *__restrict__ out, int oihw)
{
    constexpr int PADW = (KW - 1) / 2, NPX = 4 + KW - 1, NPAIR = (NPX + 1) / 2;   // pixels an X item loads, as packed pairs
    constexpr int OPT = 64 * 128, BUF = 2 * OPT;   // one operand tile (64 rows x 128 B), one buffer (dY | X)
    static_assert(KW * 8 <= 64 && (KW & 1) && NPAIR * 2 >= NPX + 1 - 1, "tile");
    extern __shared__ __attribute__((aligned(16))) char wsb[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_co = Cout / 64;
    const int units = tiles_co * KW * S, chunk = (units + 7) >> 3;
    const int item = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= chunk || item >= units) return;
    const int kh = item % KW, co0 = (item / KW) % tiles_co * 64, slice = item / (KW * tiles_co);
    const long P = (long)N * H * W;
    const long p0 = slice * px_per_slice, p1 = p0 + px_per_slice < P ? p0 + px_per_slice : P;
    const int K = (int)((p1 - p0) / WG_PX);

    int ln = (int)(p0 / ((long)H * W)), loh, low;
    {
        const int rem = (int)(p0 - (long)ln * H * W);
        loh = rem / W;
        low = rem - loh * W;
    }
    auto advance = [&]() {
        low += WG_PX;
        const bool row_end = low == W;
        low = row_end ? 0 : low;
        loh += row_end ? 1 : 0;
        const bool img_end = loh == H;
        loh = img_end ? 0 : loh;
        ln += img_end ? 1 : 0;
    };
    auto lds_at = [&](int buf, int tile_off, int row, int col, int cs) -> char * {
        const int pr = row ^ ((row >> cs) & 1);
        return wsb + buf * BUF + tile_off + pr * 128 + ((col ^ wg_swz(pr)) * 16);
    };
    // role 0 (waves 0-1): dY item = channel pair q of 32, pixel block blk of 4 (8 pixels)
    // role 1 (wave 2):    X item = channel ci of 8, pixel block b8 of 8 (4 pixels, NPX loaded)
    const int q = tid & 31, blk = (tid >> 5) & 3, ci = tid & 7, b8 = (tid >> 3) & 7;
    struct Regs { float v[NPX > 16 ? NPX : 16]; unsigned m[NPAIR]; };
    auto load = [&](auto role_tag, Regs &r) {
        constexpr int ROLE = decltype(role_tag)::value;
        if constexpr (ROLE == 0) {
            unsigned off = ((unsigned)(ln * H + loh) * (unsigned)W + (unsigned)(low + 8 * blk)) * (unsigned)Cout + (unsigned)(co0 + 2 * q);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float2 v2 = *reinterpret_cast<const float2 *>(dy + off);
                r.v[2 * i] = v2.x;
                r.v[2 * i + 1] = v2.y;
                off += (unsigned)Cout;
            }
        } else if constexpr (ROLE == 1) {
            const int ih = loh + kh - PADW, iw0 = low + 4 * b8 - PADW;
            const bool row_ok = (unsigned)ih < (unsigned)H;
            const unsigned base = ((unsigned)(ln * H + (row_ok ? ih : 0)) * (unsigned)W) * 8u + (unsigned)ci;
#pragma unroll
            for (int i = 0; i < NPX; ++i) {
                const bool ok = row_ok && (unsigned)(iw0 + i) < (unsigned)W;
                r.v[i] = x[ok ? base + (unsigned)(iw0 + i) * 8u : 0u];
            }
#pragma unroll
            for (int pp = 0; pp < NPAIR; ++pp) {
                const bool ok0 = row_ok && (unsigned)(iw0 + 2 * pp) < (unsigned)W;
                const bool ok1 = 2 * pp + 1 < NPX && row_ok && (unsigned)(iw0 + 2 * pp + 1) < (unsigned)W;
                r.m[pp] = (ok0 ? 0x0000ffffu : 0u) | (ok1 ? 0xffff0000u : 0u);
            }
        }
        advance();
    };
    auto store = [&](auto role_tag, const Regs &r, int buf) {
        constexpr int ROLE = decltype(role_tag)::value;
        if constexpr (ROLE == 0) {
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                uint4 hv, lv;
                split_pair(r.v[0 + ch], r.v[2 + ch], hv.x, lv.x);
                split_pair(r.v[4 + ch], r.v[6 + ch], hv.y, lv.y);
                split_pair(r.v[8 + ch], r.v[10 + ch], hv.z, lv.z);
                split_pair(r.v[12 + ch], r.v[14 + ch], hv.w, lv.w);
                *reinterpret_cast<uint4 *>(lds_at(buf, 0, 2 * q + ch, blk, 1)) = hv;
                *reinterpret_cast<uint4 *>(lds_at(buf, 0, 2 * q + ch, 4 + blk, 1)) = lv;
            }
        } else if constexpr (ROLE == 1) {
            unsigned h[NPAIR], l[NPAIR];
#pragma unroll
            for (int pp = 0; pp < NPAIR; ++pp) {
                split_pair(r.v[2 * pp], 2 * pp + 1 < NPX ? r.v[2 * pp + 1] : 0.f, h[pp], l[pp]);
                h[pp] &= r.m[pp];
                l[pp] &= r.m[pp];
            }
#pragma unroll
            for (int kw = 0; kw < KW; ++kw) {   // tap kw pairs output pixel j with loaded pixel j + kw
                uint2 hv, lv;
                if (kw & 1) {
                    hv = make_uint2(__builtin_amdgcn_alignbit(h[kw / 2 + 1], h[kw / 2], 16), __builtin_amdgcn_alignbit(h[kw / 2 + 2], h[kw / 2 + 1], 16));
                    lv = make_uint2(__builtin_amdgcn_alignbit(l[kw / 2 + 1], l[kw / 2], 16), __builtin_amdgcn_alignbit(l[kw / 2 + 2], l[kw / 2 + 1], 16));
                } else {
                    hv = make_uint2(h[kw / 2], h[kw / 2 + 1]);
                    lv = make_uint2(l[kw / 2], l[kw / 2 + 1]);
                }
                // rows are written one channel apart (cs = 0 would flip every row: no flip, the eight lanes of a pixel block cover
                // eight consecutive rows = both bank halves already)
                const int row = kw * 8 + ci;
                char *dst = wsb + buf * BUF + OPT + row * 128 + (b8 & 1) * 8;
                *reinterpret_cast<uint2 *>(dst + (((b8 >> 1) ^ wg_swz(row)) * 16)) = hv;
                *reinterpret_cast<uint2 *>(dst + (((4 + (b8 >> 1)) ^ wg_swz(row)) * 16)) = lv;
            }
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r2 = 0; r2 < 16; ++r2) acc[r2] = 0.f;
    auto compute = [&](int buf) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int col = 2 * kb + (lane >> 5);
            const int arow = wm * 32 + (lane & 31), brow = wn * 32 + (lane & 31);
            const bf16x8_t ah = *reinterpret_cast<const bf16x8_t *>(lds_at(buf, 0, arow, col, 1));
            const bf16x8_t al = *reinterpret_cast<const bf16x8_t *>(lds_at(buf, 0, arow, 4 + col, 1));
            const char *bb = wsb + buf * BUF + OPT + brow * 128;
            const bf16x8_t bh = *reinterpret_cast<const bf16x8_t *>(bb + ((col ^ wg_swz(brow)) * 16));
            const bf16x8_t bl = *reinterpret_cast<const bf16x8_t *>(bb + (((4 + col) ^ wg_swz(brow)) * 16));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        }
    };
    auto run_role = [&](auto role_tag) {
        Regs r0, r1;
        auto step = [&](Regs &rload, const Regs &rnext, int kk) {
            if (kk + 2 < K) load(role_tag, rload);
            compute(kk & 1);
            if (kk + 1 < K) store(role_tag, rnext, (kk + 1) & 1);
            __syncthreads();
        };
        int k = 0;
        if (K >= 4) {
            load(role_tag, r0);
            store(role_tag, r0, 0);
            load(role_tag, r1);
            __syncthreads();
            for (; k + 3 < K; k += 2) {
                load(role_tag, r0);
                compute(0);
                store(role_tag, r1, 1);
                __syncthreads();
                load(role_tag, r1);
                compute(1);
                store(role_tag, r0, 0);
                __syncthreads();
            }
        } else {
            if (K > 0) {
                load(role_tag, r0);
                store(role_tag, r0, 0);
                if (K > 1) load(role_tag, r1);
            }
            __syncthreads();
        }
        for (; k < K; k += 2) {
            step(r0, r1, k);
            if (k + 1 < K) step(r1, r0, k + 1);
        }
    };
    if (wave < 2) run_role(std::integral_constant<int, 0>{});
    else if (wave == 2) run_role(std::integral_constant<int, 1>{});
    else run_role(std::integral_constant<int, 2>{});

    // column n of the tile = (kw, ci); the columns behind the last tap multiplied whatever the LDS held and are dropped
    const int n = wn * 32 + (lane & 31), kw = n >> 3, cin = n & 7;
    if (kw >= KW) return;
    const int tap = kh * KW + kw, ntaps = KW * KW;
    float *o = out + (size_t)slice * Cout * ntaps * 8;
#pragma unroll
    for (int r2 = 0; r2 < 16; ++r2) {
        const int co = co0 + wm * 32 + (r2 & 3) + 8 * (r2 >> 2) + 4 * (lane >> 5);
        const size_t at = S > 1 ? ((size_t)tap * Cout + co) * 8 + cin
                                : (oihw ? ((size_t)co * 8 + cin) * ntaps + tap : ((size_t)co * ntaps + tap) * 8 + cin);
        o[at] = acc[r2];
    }
}

// launcher: picks the tile, the pixel slices (split-K) and the partial buffer.  `part` must hold 32 * Cout*ntaps*Cin floats
// (at most 32 slices); the result lands in `out`.
int launch_wgrad(const float *go, int O, const float *in, int I, int N, int Hin, int Win, int Hg, int Wg, int stride, int pad,
                 int KWd, int ntaps, int oihw, float *out, float *part, size_t part_floats, hipStream_t st, int precision = 0)
{
    const long P = (long)N * Hg * Wg;
    const size_t w_floats = (size_t)O * ntaps * I;
    // the bf16x3 loaders address both tensors with 32-bit element offsets
    if (precision == 1 && ((size_t)N * Hin * Win * I >= ((size_t)1 << 32) || (size_t)P * O >= ((size_t)1 << 32)))
        LWG_FAIL(LWG_ERR_UNSUPPORTED, "wgrad: tensors of 2^32 elements or more (%d x %d x %d x %d)", N, Hin, Win, I > O ? I : O);
    auto reduce = [&](long S) -> int {
        if (S > 1) {
            if (w_floats % 4 == 0) reduce_slices4_kernel<<<ceil_div((long)(w_floats / 4), 256), 256, 0, st>>>(
                reinterpret_cast<const float4 *>(part), (int)S, (long)(w_floats / 4), reinterpret_cast<float4 *>(out));
            else reduce_slices_kernel<<<ceil_div((long)w_floats, 256), 256, 0, st>>>(part, (int)S, (long)w_floats, out);
            LWG_LAUNCH_CHECK("reduce_slices_kernel");
        }
        return LWG_OK;
    };
    // 3x3 / stride 1 / pad 1 on whole 32-pixel row segments: one kernel row per workgroup (wgrad_row3_bf16x3_kernel)
    static const char *row3_env = getenv("LWG_WGRAD_ROW3");   // "0": the per-tap kernel everywhere (A/B switch)
    if (precision == 1 && stride == 1 && pad == 1 && KWd == 3 && ntaps == 9 && Hin == Hg && Win == Wg && Wg % WG_PX == 0 &&
        O % 64 == 0 && I % 64 == 0 && !(row3_env && row3_env[0] == '0')) {
        static const char *s_env = getenv("LWG_WGRAD_ROW3_SLICES");  // force the slice count (measurement)
        const int TA = O % 128 == 0 ? 128 : 64, TB = I % 128 == 0 ? 128 : 64;
        const int nwaves = TB == 128 ? 8 : 4;
        const size_t lds = (size_t)2 * (TA + 3 * TB) * 128;
        const long units = (long)(O / TA) * (I / TB) * 3, slots = (long)device_cu_count() * (lds <= 80 * 1024 - 4096 ? 2 : 1);
        // a step is 3 x (TA x TB / 1024 / waves) x 6 MFMAs of 32 cycles per wave, the waves (and workgroups) of a SIMD
        // share its pipe; a slice costs its partial's trip through reduce_slices
        const double step_us = (slots / device_cu_count()) * (nwaves / 4) * 3.0 * (TA * TB / 1024 / nwaves) * 6 * 32 / 1800.0;
        const double slice_us = (double)w_floats * 4 / 5e6;
        long cap = (long)(part_floats / w_floats);
        if (cap > kWgradMaxSlices) cap = kWgradMaxSlices;
        if (cap > P / WG_PX) cap = P / WG_PX;
        if (cap < 1) cap = 1;
        long S = 1, per = P;
        double best = -1.;
        for (long s2 = 1; s2 <= cap; ++s2) {
            const long ps = ceil_div(ceil_div(P, s2), WG_PX) * (long)WG_PX;
            const long se = ceil_div(P, ps);
            const double cost = (double)ceil_div(units * se, slots) * (ps / WG_PX + 3) * step_us + (se > 1 ? 3. + (se + 1) * slice_us : 0.);
            if (best < 0 || cost < best) { best = cost; S = se; per = ps; }
        }
        if (s_env && atol(s_env) >= 1 && atol(s_env) <= cap) {
            per = ceil_div(ceil_div(P, atol(s_env)), WG_PX) * (long)WG_PX;
            S = ceil_div(P, per);
        }
        float *wout = S == 1 ? out : part;
        const unsigned grid = 8u * (unsigned)ceil_div(units * S, 8);
        auto run = [&](auto kern, DeviceOnce &once) -> int {
            if (!once.done()) {
                LWG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                once.mark();
            }
            kern<<<grid, 64 * nwaves, lds, st>>>(go, O, in, I, N, Hg, Wg, per, (int)S, wout, oihw);
            return LWG_OK;
        };
        static DeviceOnce once[4];
        int rc;
        if (TA == 128 && TB == 128) rc = run(&wgrad_row3_bf16x3_kernel<128, 128, 2, 2, 2, 4>, once[0]);
        else if (TB == 128) rc = run(&wgrad_row3_bf16x3_kernel<64, 128, 2, 2, 2, 4>, once[1]);
        else if (TA == 128) rc = run(&wgrad_row3_bf16x3_kernel<128, 64, 4, 2, 2, 2>, once[2]);
        else rc = run(&wgrad_row3_bf16x3_kernel<64, 64, 2, 2, 2, 2>, once[3]);
        if (rc != LWG_OK) return rc;
        LWG_LAUNCH_CHECK("wgrad_row3_bf16x3_kernel");
        if (S > 1) {   // small filters with many slices: a thread per element; big ones: coalesced through LDS
            if ((long)O * I * 9 <= 512 * 1024)
                reduce_taps_direct_kernel<<<ceil_div((long)O * I * 9, 256), 256, 0, st>>>(part, (int)S, O, I, 9, oihw, out);
            else
                reduce_taps_kernel<<<ceil_div((long)O * I, 64), 256, 0, st>>>(part, (int)S, O, I, 9, oihw, out);
            LWG_LAUNCH_CHECK("reduce_taps_kernel");
        }
        return LWG_OK;
    }
    // the 7x7 stem: eight input channels, one kernel row per workgroup (wgrad_rowk_c8_bf16x3_kernel)
    if (precision == 1 && stride == 1 && KWd == 7 && ntaps == 49 && pad == 3 && I == 8 && Hin == Hg && Win == Wg && Wg % WG_PX == 0 &&
        O % 64 == 0 && !(row3_env && row3_env[0] == '0')) {
        const long units = (long)(O / 64) * 7, slots = (long)device_cu_count() * 4;
        long cap = (long)(part_floats / w_floats);
        if (cap > kWgradMaxSlices) cap = kWgradMaxSlices;
        if (cap > P / WG_PX) cap = P / WG_PX;
        if (cap < 1) cap = 1;
        long S = ceil_div(slots, units);   // one round of workgroups, no fewer than 8 steps each
        if (S > cap) S = cap;
        while (S > 1 && ceil_div(P, S) < 8 * WG_PX) --S;
        const long per = ceil_div(ceil_div(P, S), WG_PX) * (long)WG_PX;
        S = ceil_div(P, per);
        float *wout = S == 1 ? out : part;
        wgrad_rowk_c8_bf16x3_kernel<7><<<8u * (unsigned)ceil_div(units * S, 8), 256, 2 * 2 * 64 * 128, st>>>(go, O, in, N, Hg, Wg, per, (int)S,
                                                                                                      wout, oihw);
        LWG_LAUNCH_CHECK("wgrad_rowk_c8_bf16x3_kernel");
        if (S > 1) {
            reduce_taps_direct_kernel<<<ceil_div((long)O * I * 49, 256), 256, 0, st>>>(part, (int)S, O, I, 49, oihw, out);
            LWG_LAUNCH_CHECK("reduce_taps_direct_kernel");
        }
        return LWG_OK;
    }
    const int T = (O % 128 == 0 && I % 128 == 0) ? 128 : 64;
    const long tiles = (long)(O / T) * ceil_div(I, T) * ntaps;
    // Pixel slices (split-K).  The grid runs in rounds of `slots` resident workgroups (LDS: two 128-wide or four
    // 64-wide ones per CU), each walking its slice in steps of WG_PX pixels: take the slice count with the fewest
    // steps end to end (a 9-tap 512x512 layer is 144 tiles: 4 slices = 576 workgroups is two rounds of 32 steps,
    // 3 slices one round of 43), within what the partial buffer holds.
    const long slots = (long)device_cu_count() * (T == 128 ? 2 : 4);
    long cap = (long)(part_floats / w_floats);
    if (cap > kWgradMaxSlices) cap = kWgradMaxSlices;
    if (cap > ceil_div(P, WG_PX)) cap = ceil_div(P, WG_PX);
    if (cap < 1) cap = 1;
    long S = 1, per = 0, best = -1;
    for (long s = 1; s <= cap; ++s) {
        const long ps = ceil_div(ceil_div(P, s), WG_PX) * (long)WG_PX;
        const long se = ceil_div(P, ps);
        const long cost = ceil_div(tiles * se, slots) * (ps / WG_PX + 2) + (se > 1 ? se / 8 : 0);   // + prologue/epilogue, + reduce
        if (best < 0 || cost < best) { best = cost; S = se; per = ps; }
    }
    if (S > 1 && (size_t)S * w_floats > part_floats) LWG_FAIL(LWG_ERR_STATE, "wgrad: partial buffer too small");
    float *wout = S == 1 ? out : part;
    const dim3 grid(ceil_div(I, T), O / T, (unsigned)(ntaps * S));
    if (precision == 1) {
        const size_t lds16 = (size_t)4 * T * 128;
        if (T == 128) {
            static DeviceOnce opt_in16;
            if (!opt_in16.done()) {
                LWG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_bf16x3_kernel<128>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16));
                opt_in16.mark();
            }
            wgrad_bf16x3_kernel<128><<<grid, 256, lds16, st>>>(go, O, in, I, N, Hin, Win, Hg, Wg, stride, pad, per, wout, KWd, ntaps, oihw);
        } else {
            wgrad_bf16x3_kernel<64><<<grid, 256, lds16, st>>>(go, O, in, I, N, Hin, Win, Hg, Wg, stride, pad, per, wout, KWd, ntaps, oihw);
        }
        LWG_LAUNCH_CHECK("wgrad_bf16x3_kernel");
        return reduce(S);
    }
    const size_t lds = (size_t)4 * WG_PX * (T + 4) * sizeof(float);
    if (T == 128) {
        static DeviceOnce opt_in;
        if (!opt_in.done()) {
            LWG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds));
            opt_in.mark();
        }
        wgrad_kernel<128><<<grid, 256, lds, st>>>(go, O, in, I, N, Hin, Win, Hg, Wg, stride, pad, per, wout, KWd, ntaps, oihw);
    } else {
        wgrad_kernel<64><<<grid, 256, lds, st>>>(go, O, in, I, N, Hin, Win, Hg, Wg, stride, pad, per, wout, KWd, ntaps, oihw);
    }
    LWG_LAUNCH_CHECK("wgrad_kernel");
    return reduce(S);
}

// ---- data-gradient weight matrices from the master copy W[co][kh*4+kw][ci] (run after every optimiser step).
// stride 1: one matrix [ci][kh'*4+kw'][co] = W[co][(3-kh')*4 + (3-kw')][ci]
// stride 2: four phase matrices [ci][th*2+tw][co], phase (py,px): kh = py ? (th ? 0 : 2) : (th ? 1 : 3), same for kw
// `rows` >= Cin input-channel rows are produced (rows beyond Cin are zero: the 64-wide matrix of the first layer)
__global__ __launch_bounds__(256) void dgrad_weights_kernel(const float *__restrict__ w, int Cout, int Cin, int stride,
                                                            float *__restrict__ wd, int rows)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)Cout * 16 * rows;
    if (i >= total) return;
    if (stride == 1) {
        const int co = (int)(i % Cout);
        const int t = (int)((i / Cout) % 16);
        const int ci = (int)(i / ((long)Cout * 16));
        const int kh = 3 - (t >> 2), kw = 3 - (t & 3);
        wd[i] = ci < Cin ? w[((size_t)co * 16 + kh * 4 + kw) * Cin + ci] : 0.f;
    } else {
        const long per_phase = (long)rows * 4 * Cout;
        const int phase = (int)(i / per_phase);
        const long j = i - phase * per_phase;
        const int co = (int)(j % Cout);
        const int t = (int)((j / Cout) % 4);
        const int ci = (int)(j / ((long)Cout * 4));
        const int py = phase >> 1, px = phase & 1, th = t >> 1, tw = t & 1;
        const int kh = py ? (th ? 0 : 2) : (th ? 1 : 3);
        const int kw = px ? (tw ? 0 : 2) : (tw ? 1 : 3);
        wd[i] = ci < Cin ? w[((size_t)co * 16 + kh * 4 + kw) * Cin + ci] : 0.f;
    }
}

// ---- Adam (torch.optim.Adam, no weight decay, no amsgrad).  step_dev != null: the step count lives in device memory
// (advanced by step_inc_kernel in front of this launch) and the bias corrections come from it -- the form a captured graph
// can replay: a host-side count would be frozen into the launch arguments.  The state carries beta1^t and beta2^t as running
// products in double (a powf in the kernel would do, but its expansion contains packed-fp32 instructions of the form
// tests/test_pk_opsel_lint.py bans from this library).
struct AdamStep { long step; double p1, p2; };
__global__ void step_inc_kernel(AdamStep *s, float b1, float b2)
{
    s->step += 1;
    s->p1 *= (double)b1;
    s->p2 *= (double)b2;
}
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                   float bc1, float bc2_sqrt, const AdamStep *__restrict__ step_dev)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (step_dev) {
        bc1 = 1.f - (float)step_dev->p1;
        bc2_sqrt = sqrtf(1.f - (float)step_dev->p2);
    }
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
}

// channel 0 of an NHWC tensor -> dense (N,1,H,W)
__global__ __launch_bounds__(256) void take_channel0_kernel(const float *__restrict__ x, long P, int C, float *__restrict__ y)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P) y[i] = x[(size_t)i * C];
}

// ---- op-level InstanceNorm2d(affine) [+ ReLU] and its gradient (NHWC fp32)
// y = act(gamma * (x - mean) * rstd + beta); stats = (mean, rstd) per (image, channel) from in_stats_kernel
__global__ __launch_bounds__(256) void in_apply_kernel(const float *__restrict__ x, const float2 *__restrict__ stats,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta,
                                                       int relu, int HW, int C, long total, float *__restrict__ y)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % C), n = (int)(e / ((long)HW * C));
    const float2 st = stats[(size_t)n * C + c];
    float v = gamma[c] * ((x[e] - st.x) * st.y) + beta[c];
    if (relu && v < 0.f) v = 0.f;
    y[e] = v;
}
// pass 1: per (image, channel) sums of g and g * xhat, g = dy masked by the ReLU (y > 0) when y is given
__global__ __launch_bounds__(256) void in_affine_bwd_reduce_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                                   const float *__restrict__ dy, const float2 *__restrict__ stats,
                                                                   int HW, int C, float2 *__restrict__ sums)
{
    __shared__ double sh[2][RS_SL][RS_CH];
    const int cl = threadIdx.x & (RS_CH - 1), c = blockIdx.x * RS_CH + cl, slice = threadIdx.x / RS_CH, n = blockIdx.y;
    const size_t base = (size_t)n * HW * C + c;
    const float2 st = stats[(size_t)n * C + c];
    double s1 = 0., s2 = 0.;
    for (int i = slice; i < HW; i += RS_SL) {
        const size_t o = base + (size_t)i * C;
        const float g = (y && !(y[o] > 0.f)) ? 0.f : dy[o];
        s1 += g;
        s2 += (double)g * ((x[o] - st.x) * st.y);
    }
    sh[0][slice][cl] = s1;
    sh[1][slice][cl] = s2;
    __syncthreads();
    if (slice == 0) {
        s1 = s2 = 0.;
        for (int k = 0; k < RS_SL; ++k) {
            s1 += sh[0][k][cl];
            s2 += sh[1][k][cl];
        }
        sums[(size_t)n * C + c] = make_float2((float)s1, (float)s2);   // plain sums: dbeta / dgamma need them over n too
    }
}
// pass 2: dx = gamma * rstd * (g - S1/HW - xhat * S2/HW)
__global__ __launch_bounds__(256) void in_affine_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                                  const float *__restrict__ dy, const float2 *__restrict__ stats,
                                                                  const float2 *__restrict__ sums, const float *__restrict__ gamma,
                                                                  int HW, int C, long total, float *__restrict__ dx)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % C), n = (int)(e / ((long)HW * C));
    const float2 st = stats[(size_t)n * C + c], sm = sums[(size_t)n * C + c];
    const float g = (y && !(y[e] > 0.f)) ? 0.f : dy[e];
    const float inv = 1.f / (float)HW;
    dx[e] = gamma[c] * st.y * (g - sm.x * inv - (x[e] - st.x) * st.y * (sm.y * inv));
}
// dbeta[c] = sum_n S1, dgamma[c] = sum_n S2 (fixed order)
__global__ __launch_bounds__(256) void in_affine_bwd_params_kernel(const float2 *__restrict__ sums, int N, int C,
                                                                   float *__restrict__ dgamma, float *__restrict__ dbeta)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double b = 0., g = 0.;
    for (int n = 0; n < N; ++n) {
        b += sums[(size_t)n * C + c].x;
        g += sums[(size_t)n * C + c].y;
    }
    dbeta[c] = (float)b;
    dgamma[c] = (float)g;
}

// ---- the same two reductions for large maps: the pixels of an (image, channel) are split into S slabs handled by
// different workgroups (grid (C/16, N, S)), partial sums go to a double scratch [N][S][C][2] and are combined in slab
// order by the *_final kernels -- deterministic, and 16 x S times more workgroups than channels / 16 x images.
__global__ __launch_bounds__(256) void in_stats_partial_kernel(const float *__restrict__ x, int HW, int C, int S,
                                                               double *__restrict__ part)
{
    __shared__ double sh[2][RS_SL][RS_CH];
    const int cl = threadIdx.x & (RS_CH - 1), c = blockIdx.x * RS_CH + cl, slice = threadIdx.x / RS_CH, n = blockIdx.y, sb = blockIdx.z;
    const int per = (HW + S - 1) / S, p0 = sb * per, p1 = p0 + per < HW ? p0 + per : HW;
    const float *p = x + (size_t)n * HW * C + c;
    double s = 0., q = 0.;
    for (int i = p0 + slice; i < p1; i += RS_SL) {
        const double v = p[(size_t)i * C];
        s += v;
        q += v * v;
    }
    sh[0][slice][cl] = s;
    sh[1][slice][cl] = q;
    __syncthreads();
    if (slice == 0) {
        s = q = 0.;
        for (int k = 0; k < RS_SL; ++k) {
            s += sh[0][k][cl];
            q += sh[1][k][cl];
        }
        double *o = part + (((size_t)n * S + sb) * C + c) * 2;
        o[0] = s;
        o[1] = q;
    }
}
__global__ __launch_bounds__(256) void in_stats_final_kernel(const double *__restrict__ part, int N, int C, int S, int HW,
                                                             float2 *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    double s = 0., q = 0.;
    for (int k = 0; k < S; ++k) {
        const double *o = part + (((size_t)n * S + k) * C + c) * 2;
        s += o[0];
        q += o[1];
    }
    const double mean = s / HW, var = fmax(q / HW - mean * mean, 0.);
    out[i] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)kEps)));
}
__global__ __launch_bounds__(256) void in_affine_bwd_partial_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                                    const float *__restrict__ dy, const float2 *__restrict__ stats,
                                                                    int HW, int C, int S, double *__restrict__ part)
{
    __shared__ double sh[2][RS_SL][RS_CH];
    const int cl = threadIdx.x & (RS_CH - 1), c = blockIdx.x * RS_CH + cl, slice = threadIdx.x / RS_CH, n = blockIdx.y, sb = blockIdx.z;
    const int per = (HW + S - 1) / S, p0 = sb * per, p1 = p0 + per < HW ? p0 + per : HW;
    const size_t base = (size_t)n * HW * C + c;
    const float2 st = stats[(size_t)n * C + c];
    double s1 = 0., s2 = 0.;
    for (int i = p0 + slice; i < p1; i += RS_SL) {
        const size_t o = base + (size_t)i * C;
        const float g = (y && !(y[o] > 0.f)) ? 0.f : dy[o];
        s1 += g;
        s2 += (double)g * ((x[o] - st.x) * st.y);
    }
    sh[0][slice][cl] = s1;
    sh[1][slice][cl] = s2;
    __syncthreads();
    if (slice == 0) {
        s1 = s2 = 0.;
        for (int k = 0; k < RS_SL; ++k) {
            s1 += sh[0][k][cl];
            s2 += sh[1][k][cl];
        }
        double *o = part + (((size_t)n * S + sb) * C + c) * 2;
        o[0] = s1;
        o[1] = s2;
    }
}
// The same sums for C % 64 == 0 (every layer of the generator): a thread owns four channels (float4 loads: a pixel row of
// the workgroup is 256 contiguous bytes, not 64) and every sixteenth pixel of slab sb, four pixels in flight.
constexpr int RS4_CH = 64;
__global__ __launch_bounds__(256) void in_affine_bwd_sums4_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                                  const float *__restrict__ dy, const float2 *__restrict__ stats,
                                                                  int HW, int C, int S, double *__restrict__ part)
{
    __shared__ double sh[2][RS_SL][RS4_CH];
    const int cq = threadIdx.x & 15, c = blockIdx.x * RS4_CH + 4 * cq, slice = threadIdx.x >> 4, n = blockIdx.y, sb = blockIdx.z;
    const int per = (HW + S - 1) / S, p0 = sb * per, p1 = p0 + per < HW ? p0 + per : HW;
    const size_t base = (size_t)n * HW * C + c;
    const float4 sa = ld4(reinterpret_cast<const float *>(stats + (size_t)n * C + c));       // (mean, rstd) of c, c+1
    const float4 sb4 = ld4(reinterpret_cast<const float *>(stats + (size_t)n * C + c + 2));  // c+2, c+3
    const float mean[4] = {sa.x, sa.z, sb4.x, sb4.z}, rstd[4] = {sa.y, sa.w, sb4.y, sb4.w};
    double s1[4] = {0., 0., 0., 0.}, s2[4] = {0., 0., 0., 0.};
    auto add = [&](const float4 xv, const float4 yv, const float4 gv) {
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ys[4] = {yv.x, yv.y, yv.z, yv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float g = (y && !(ys[k] > 0.f)) ? 0.f : gs[k];
            s1[k] += g;
            s2[k] += (double)g * ((xs[k] - mean[k]) * rstd[k]);
        }
    };
    int i = p0 + slice;
    for (; i + 3 * RS_SL < p1; i += 4 * RS_SL) {
        float4 xv[4], yv[4], gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t o = base + (size_t)(i + u * RS_SL) * C;
            xv[u] = ld4(x + o);
            yv[u] = y ? ld4(y + o) : make_float4(1.f, 1.f, 1.f, 1.f);
            gv[u] = ld4(dy + o);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) add(xv[u], yv[u], gv[u]);
    }
    for (; i < p1; i += RS_SL) {
        const size_t o = base + (size_t)i * C;
        add(ld4(x + o), y ? ld4(y + o) : make_float4(1.f, 1.f, 1.f, 1.f), ld4(dy + o));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sh[0][slice][4 * cq + k] = s1[k];
        sh[1][slice][4 * cq + k] = s2[k];
    }
    __syncthreads();
    if (threadIdx.x < RS4_CH) {
        const int cl = threadIdx.x;
        double a = 0., b = 0.;
        for (int k = 0; k < RS_SL; ++k) {
            a += sh[0][k][cl];
            b += sh[1][k][cl];
        }
        double *o = part + (((size_t)n * S + sb) * C + blockIdx.x * RS4_CH + cl) * 2;
        o[0] = a;
        o[1] = b;
    }
}
// grid ceil(N*C / 64): 64 (image, channel) pairs per workgroup, four threads per pair each summing every fourth slab
__global__ __launch_bounds__(256) void in_sums_final_kernel(const double *__restrict__ part, int N, int C, int S,
                                                            float2 *__restrict__ sums)
{
    __shared__ double sh[2][4][64];
    const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + cl;
    double s1 = 0., s2 = 0.;
    if (i < N * C) {
        const int n = i / C, c = i - n * C;
        for (int k = grp; k < S; k += 4) {
            const double *o = part + (((size_t)n * S + k) * C + c) * 2;
            s1 += o[0];
            s2 += o[1];
        }
    }
    sh[0][grp][cl] = s1;
    sh[1][grp][cl] = s2;
    __syncthreads();
    if (grp == 0 && i < N * C)
        sums[i] = make_float2((float)(sh[0][0][cl] + sh[0][1][cl] + sh[0][2][cl] + sh[0][3][cl]),
                              (float)(sh[1][0][cl] + sh[1][1][cl] + sh[1][2][cl] + sh[1][3][cl]));
}
inline int in_slabs(int N, int C, int HW)
{
    // enough workgroups to fill the chip (~1024), at least 1024 pixels per slab, at most 64 slabs
    long S = 1024 / ((long)(C / RS_CH) * N);
    if (S > HW / 1024) S = HW / 1024;
    if (S > 64) S = 64;
    return S < 1 ? 1 : (int)S;
}

// ---- float4 forms of the InstanceNorm passes for C % 64 == 0 (every layer of the generator)
// statistics partials: slab sb of image n, 64 channels per workgroup, a thread owns four channels and every sixteenth pixel
__global__ __launch_bounds__(256) void in_stats4_kernel(const float *__restrict__ x, int HW, int C, int S, double *__restrict__ part)
{
    __shared__ double sh[2][RS_SL][RS4_CH];
    const int cq = threadIdx.x & 15, c = blockIdx.x * RS4_CH + 4 * cq, slice = threadIdx.x >> 4, n = blockIdx.y, sb = blockIdx.z;
    const int per = (HW + S - 1) / S, p0 = sb * per, p1 = p0 + per < HW ? p0 + per : HW;
    const float *p = x + (size_t)n * HW * C + c;
    double s[4] = {0., 0., 0., 0.}, q[4] = {0., 0., 0., 0.};
    auto add = [&](const float4 v) {
        const double d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s[k] += d[k];
            q[k] += d[k] * d[k];
        }
    };
    int i = p0 + slice;
    for (; i + 3 * RS_SL < p1; i += 4 * RS_SL) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ld4(p + (size_t)(i + u * RS_SL) * C);
#pragma unroll
        for (int u = 0; u < 4; ++u) add(v[u]);
    }
    for (; i < p1; i += RS_SL) add(ld4(p + (size_t)i * C));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sh[0][slice][4 * cq + k] = s[k];
        sh[1][slice][4 * cq + k] = q[k];
    }
    __syncthreads();
    if (threadIdx.x < RS4_CH) {
        const int cl = threadIdx.x;
        double a = 0., b = 0.;
        for (int k = 0; k < RS_SL; ++k) {
            a += sh[0][k][cl];
            b += sh[1][k][cl];
        }
        double *o = part + (((size_t)n * S + sb) * C + blockIdx.x * RS4_CH + cl) * 2;
        o[0] = a;
        o[1] = b;
    }
}
// grid ceil(N*C / 64): four threads per (image, channel), each summing every fourth slab
__global__ __launch_bounds__(256) void in_stats_final4_kernel(const double *__restrict__ part, int N, int C, int S, int HW,
                                                              float2 *__restrict__ out)
{
    __shared__ double sh[2][4][64];
    const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + cl;
    double s = 0., q = 0.;
    if (i < N * C) {
        const int n = i / C, c = i - n * C;
        for (int k = grp; k < S; k += 4) {
            const double *o = part + (((size_t)n * S + k) * C + c) * 2;
            s += o[0];
            q += o[1];
        }
    }
    sh[0][grp][cl] = s;
    sh[1][grp][cl] = q;
    __syncthreads();
    if (grp == 0 && i < N * C) {
        s = sh[0][0][cl] + sh[0][1][cl] + sh[0][2][cl] + sh[0][3][cl];
        q = sh[1][0][cl] + sh[1][1][cl] + sh[1][2][cl] + sh[1][3][cl];
        const double mean = s / HW, var = fmax(q / HW - mean * mean, 0.);
        out[i] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)kEps)));
    }
}
__global__ __launch_bounds__(256) void in_apply4_kernel(const float *__restrict__ x, const float2 *__restrict__ stats,
                                                        const float *__restrict__ gamma, const float *__restrict__ beta,
                                                        int relu, int HW, int C, long total4, float *__restrict__ y)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const long e = i * 4;
    const int c = (int)(e % C), n = (int)(e / ((long)HW * C));
    const float4 v = ld4(x + e), g = ld4(gamma + c), b = ld4(beta + c);
    const float4 sa = ld4(reinterpret_cast<const float *>(stats + (size_t)n * C + c));
    const float4 sb = ld4(reinterpret_cast<const float *>(stats + (size_t)n * C + c + 2));
    float4 o;
    o.x = g.x * ((v.x - sa.x) * sa.y) + b.x;
    o.y = g.y * ((v.y - sa.z) * sa.w) + b.y;
    o.z = g.z * ((v.z - sb.x) * sb.y) + b.z;
    o.w = g.w * ((v.w - sb.z) * sb.w) + b.w;
    if (relu) {
        o.x = o.x < 0.f ? 0.f : o.x; o.y = o.y < 0.f ? 0.f : o.y; o.z = o.z < 0.f ? 0.f : o.z; o.w = o.w < 0.f ? 0.f : o.w;
    }
    *reinterpret_cast<float4 *>(y + e) = o;
}
__global__ __launch_bounds__(256) void in_affine_bwd_apply4_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                                   const float *__restrict__ dy, const float2 *__restrict__ stats,
                                                                   const float2 *__restrict__ sums, const float *__restrict__ gamma,
                                                                   int HW, int C, long total4, float *__restrict__ dx)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const long e = i * 4;
    const int c = (int)(e % C), n = (int)(e / ((long)HW * C));
    const float4 xv = ld4(x + e), gv = ld4(dy + e), gm = ld4(gamma + c);
    const float4 yv = y ? ld4(y + e) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 sa = ld4(reinterpret_cast<const float *>(stats + (size_t)n * C + c));
    const float4 sb = ld4(reinterpret_cast<const float *>(stats + (size_t)n * C + c + 2));
    const float4 ma = ld4(reinterpret_cast<const float *>(sums + (size_t)n * C + c));
    const float4 mb = ld4(reinterpret_cast<const float *>(sums + (size_t)n * C + c + 2));
    const float inv = 1.f / (float)HW;
    auto one = [&](float xx, float yy, float gg, float gam, float mean, float rstd, float s1, float s2) {
        const float g = (y && !(yy > 0.f)) ? 0.f : gg;
        return gam * rstd * (g - s1 * inv - (xx - mean) * rstd * (s2 * inv));
    };
    float4 o;
    o.x = one(xv.x, yv.x, gv.x, gm.x, sa.x, sa.y, ma.x, ma.y);
    o.y = one(xv.y, yv.y, gv.y, gm.y, sa.z, sa.w, ma.z, ma.w);
    o.z = one(xv.z, yv.z, gv.z, gm.z, sb.x, sb.y, mb.x, mb.y);
    o.w = one(xv.w, yv.w, gv.w, gm.w, sb.z, sb.w, mb.z, mb.w);
    *reinterpret_cast<float4 *>(dx + e) = o;
}

// slabs of in_affine_bwd_sums4_kernel: ~4096 workgroups, at least 256 pixels each, at most 256 slabs
inline int in_bwd_slabs4(int N, int C, int HW)
{
    long S = 4096 / ((long)(C / RS4_CH) * N);
    if (S > HW / 256) S = HW / 256;
    if (S > 256) S = 256;
    return S < 1 ? 1 : (int)S;
}

// ---- bilinear grid_sample (zeros padding), NHWC: y[n][p][c] = sum_tap w_tap * x[n or 0][tap][c]
__global__ __launch_bounds__(256) void grid_sample_nhwc_kernel(const float *__restrict__ x, const float *__restrict__ grid, int xn,
                                                               int C, int H, int W, int Ho, int Wo, int align_corners, long total,
                                                               float *__restrict__ y)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int c4n = C >> 2;
    const long pix = e / c4n;
    const int c = (int)(e - pix * c4n) * 4;
    const int n = (int)(pix / ((long)Ho * Wo));
    const float2 gq = *reinterpret_cast<const float2 *>(grid + pix * 2);
    const GridTaps t = grid_taps(gq.x, gq.y, W, H, align_corners);
    const float *base = x + (size_t)(xn > 1 ? n : 0) * H * W * C + c;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    auto tap = [&](int yy, int xx, float w) {
        const float4 v = ld4(base + ((size_t)yy * W + xx) * C);
        acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
    };
    if (t.vnw) tap(t.y0, t.x0, t.wnw);
    if (t.vne) tap(t.y0, t.x0 + 1, t.wne);
    if (t.vsw) tap(t.y0 + 1, t.x0, t.wsw);
    if (t.vse) tap(t.y0 + 1, t.x0 + 1, t.wse);
    *reinterpret_cast<float4 *>(y + pix * C + c) = acc;
}

// ---- gradient of bilinear grid_sample (zeros padding) wrt its INPUT, NHWC: dx[n][tap][c] += w_tap * dy[n][p][c].
// Several output pixels may sample one source texel, so the accumulation is atomic (fp32 add: the summation order,
// hence the last bits, can differ from run to run -- the same property torch's CUDA grid_sampler backward has).
__global__ __launch_bounds__(256) void grid_sample_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ grid,
                                                              int xn, int C, int H, int W, int Ho, int Wo, int align_corners,
                                                              long total, float *__restrict__ dx)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (output pixel, 4 channels)
    if (e >= total) return;
    const int c4n = C >> 2;
    const long pix = e / c4n;
    const int c = (int)(e - pix * c4n) * 4;
    const int n = (int)(pix / ((long)Ho * Wo));
    const float2 gq = *reinterpret_cast<const float2 *>(grid + pix * 2);
    const GridTaps t = grid_taps(gq.x, gq.y, W, H, align_corners);
    const float4 g = ld4(dy + pix * C + c);
    float *base = dx + (size_t)(xn > 1 ? n : 0) * H * W * C + c;
    auto add = [&](int yy, int xx, float w) {
        float *p = base + ((size_t)yy * W + xx) * C;
        unsafeAtomicAdd(p + 0, w * g.x);
        unsafeAtomicAdd(p + 1, w * g.y);
        unsafeAtomicAdd(p + 2, w * g.z);
        unsafeAtomicAdd(p + 3, w * g.w);
    };
    if (t.vnw) add(t.y0, t.x0, t.wnw);
    if (t.vne) add(t.y0, t.x0 + 1, t.wne);
    if (t.vsw) add(t.y0 + 1, t.x0, t.wsw);
    if (t.vse) add(t.y0 + 1, t.x0 + 1, t.wse);
}

// ---- op-level convolution API: weight re-layouts from PyTorch tensors (device to device)
// mode 0: conv OIHW (Cout,Cin,k,k) -> forward matrix [Cout][tap][Cin]
// mode 1: conv OIHW -> data-gradient matrix of a stride-1 conv [Cin][k*k-1-tap'][Cout]: dst[ci][t][co] = w[co][ci][T-1-t]
// mode 2: "transposed-conv forward" phase matrices from a tensor laid out (A, B, 3, 3): dst phase (py,px), [b][t][a]
//         = w[a][b][ky][kx], ky = py ? (ty ? 0 : 2) : 1 (generator.hip upload_convT).  Serves ConvTranspose2d forward
//         (A = Cin, B = Cout) and the data gradient of Conv2d(k3,s2,p1) (its OIHW tensor read as A = Cout, B = Cin).
// split = 1: the matrix is written in the split-bf16 format of conv.h ([hi x32 | lo x32] per 32 entries, same offsets)
__global__ __launch_bounds__(256) void weight_layout_kernel(const float *__restrict__ w, float *__restrict__ dst, int mode,
                                                            int A, int B, int taps, long total, int pitch, int split)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    auto put = [&](size_t d, float v) {
        if (!split) {
            dst[d] = v;
            return;
        }
        __bf16 *o = reinterpret_cast<__bf16 *>(dst) + (d >> 5) * 64 + (d & 31);
        const __bf16 hi = (__bf16)v;
        o[0] = hi;
        o[32] = (__bf16)(v - (float)hi);
    };
    if (mode == 0) {          // A = Cout, B = Cin: dst[a*pitch + t*B + b]
        const int b = (int)(i % B), t = (int)((i / B) % taps), a = (int)(i / ((long)B * taps));
        put((size_t)a * pitch + (size_t)t * B + b, w[((size_t)a * B + b) * taps + t]);
    } else if (mode == 1) {   // A = Cout, B = Cin: dst[b*pitch + t*A + a]
        const int a = (int)(i % A), t = (int)((i / A) % taps), b = (int)(i / ((long)A * taps));
        put((size_t)b * pitch + (size_t)t * A + a, w[((size_t)a * B + b) * taps + (taps - 1 - t)]);
    } else {                  // four phases with 1, 2, 2, 4 taps, rows of B outputs x (t, a)
        long j = i;
        int phase = 0, nt = 1;
        for (; phase < 4; ++phase) {
            nt = (1 + (phase >> 1)) * (1 + (phase & 1));
            const long sz = (long)B * nt * A;
            if (j < sz) break;
            j -= sz;
        }
        const int py = phase >> 1, px = phase & 1, KWp = 1 + px;
        const int a = (int)(j % A), t = (int)((j / A) % nt), b = (int)(j / ((long)A * nt));
        const int ty = t / KWp, tx = t - ty * KWp;
        const int ky = py == 0 ? 1 : (ty == 0 ? 2 : 0), kx = px == 0 ? 1 : (tx == 0 ? 2 : 0);
        put((size_t)i, w[(((size_t)a * B + b) * 3 + ky) * 3 + kx]);
    }
}

// sums the `KS` reduction-split slices of a general-mode conv (ConvArgs::ksplit) in order and adds the bias: y[i] = bias[i % C] + sum_k part[k][i]
__global__ __launch_bounds__(256) void ksplit_reduce_kernel(const float4 *__restrict__ part, int KS, size_t stride4, long n4, int C,
                                                            const float *__restrict__ bias, float4 *__restrict__ y)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 s = part[i];
    for (int k = 1; k < KS; ++k) {
        const float4 v = part[(size_t)k * stride4 + i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (bias) {
        const float4 b = ld4(bias + (int)((i * 4) % C));
        s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
    }
    y[i] = s;
}

struct DLayer {
    int cin = 0, cin_pad = 0, cout = 0, cout_pad = 0, stride = 2;
    bool norm = false, act = false;
    int Hin = 0, Ho = 0;
    size_t w_off = 0, b_off = 0, w_floats = 0;   // into the flat parameter / gradient / moment buffers
    float *wd = nullptr;                         // data-gradient matrices (layers 1..)
    float *raw = nullptr, *actv = nullptr;       // (2N, Ho, Ho, cout_pad): conv + bias; after norm/activation
    float *draw = nullptr, *dact = nullptr;      // gradients wrt raw / wrt this layer's OUTPUT (dact of the last = d loss)
    float2 *stats = nullptr, *sums = nullptr;    // [2N][cout_pad]
    bool got_w = false, got_b = false;
};

}  // namespace
}  // namespace lwg

struct lwg_discriminator {
    int input_nc = 0, ndf = 0, n_layers = 0, is = 0, max_batch = 0;
    std::vector<lwg::DLayer> L;
    size_t nparams = 0;
    float *params = nullptr, *grads = nullptr, *m = nullptr, *v = nullptr;
    float *x0 = nullptr;        // packed input (2N, is, is, cin_pad0)
    float *wd0 = nullptr, *dx0 = nullptr;   // input-gradient path (allocated on first use): 64-row matrices, (N,is,is,64)
    float *part = nullptr;      // split-K / column-sum partials
    size_t part_floats = 0;
    float *loss = nullptr;      // device scalar
    long step = 0;
    bool wd_stale = true;
    int precision = 0;          // 0: fp32 MFMA; 1: bf16x3 (operands split inside the conv / weight-gradient kernels)
    lwg::AdamStep *step_dev = nullptr;   // device copy of `step` (lwg_discriminator_use_device_step): what a captured graph replays
    bool device_step = false;
};

namespace lwg {
namespace {

int d_alloc(float **p, size_t floats)
{
    LWG_HIP(hipMalloc(reinterpret_cast<void **>(p), floats * sizeof(float)));
    LWG_HIP(hipMemset(*p, 0, floats * sizeof(float)));
    return LWG_OK;
}

ConvArgs base_args(const float *x, int ldx, int N, int H, int Cin, const float *w, float *y, int Cout)
{
    ConvArgs a = {};
    a.x = x; a.ldx = ldx; a.N = N; a.H = H; a.W = H; a.Cin = Cin;
    while ((1 << a.cin_log2) < Cin) ++a.cin_log2;
    a.w = w; a.y = y; a.ldy = Cout; a.Cout = Cout;
    a.dil = 1; a.general = 1;
    return a;
}

// 128-channel tiles once they give every CU a workgroup, else twice as many 64-channel ones (the 15x15 and 14x14 maps of the
// last layers are 15 row tiles: 60 workgroups of 128 channels on 256 CUs)
int d_tile_width(int mtiles, int cout)
{
    return (cout % 128 == 0 && (long)mtiles * (cout / 128) >= device_cu_count()) ? 128 : 64;
}

// launch_conv_igemm with a reduction split where a general-mode conv has few output tiles and many stages (ConvArgs::ksplit);
// the split's partials go through d->part (free between the weight gradients that use it on the same stream)
int d_launch_conv(lwg_discriminator *d, ConvArgs &a, int bn, hipStream_t st)
{
    static const char *ks_env = getenv("LWG_D_KSPLIT");   // "0": never split (A/B switch)
    const long tiles = (long)a.mtiles * (a.Cout / bn) * a.nphase;
    const size_t y_floats = (size_t)a.N * a.Ho * a.Wo * a.ldy;
    int nk = a.ph[0].Kpad / kConvBK, ks = 1;
    for (int p = 1; p < a.nphase; ++p) nk = a.ph[p].Kpad / kConvBK < nk ? a.ph[p].Kpad / kConvBK : nk;
    if (!(ks_env && ks_env[0] == '0') && a.ldy == a.Cout && a.Cout % 4 == 0)
        for (int k = 4; k >= 2; k >>= 1)
            if (tiles * k <= device_cu_count() && nk >= 16 * k && (size_t)k * y_floats <= d->part_floats) { ks = k; break; }
    if (ks == 1) return launch_conv_igemm(a, bn, st);
    const float *bias = a.bias;
    a.ksplit = ks; a.kpart = d->part; a.kpart_stride = y_floats; a.bias = nullptr;
    const int rc = launch_conv_igemm(a, bn, st);
    if (rc != LWG_OK) return rc;
    ksplit_reduce_kernel<<<ceil_div((long)(y_floats / 4), 256), 256, 0, st>>>(reinterpret_cast<const float4 *>(d->part), ks, y_floats / 4,
                                                                           (long)(y_floats / 4), a.Cout, bias, reinterpret_cast<float4 *>(a.y));
    LWG_LAUNCH_CHECK("ksplit_reduce_kernel");
    return LWG_OK;
}

// forward conv of layer l on `x` (2N images): raw = conv(x) + bias
int d_conv_forward(lwg_discriminator *d, int l, const float *x, int B, hipStream_t st)
{
    const DLayer &L = d->L[l];
    ConvArgs a = base_args(x, L.cin_pad, B, L.Hin, L.cin_pad, d->params + L.w_off, L.raw, L.cout_pad);
    a.Hm = a.Wm = a.Ho = a.Wo = L.Ho;
    a.stride = L.stride; a.pad = 1; a.os = 1;
    a.bias = d->params + L.b_off;
    a.nphase = 1;
    a.ph[0] = ConvPhase{4, 4, 16, 16 * L.cin_pad, 0, 0, 0, 0, 0};
    a.mtiles = ceil_div((long)B * L.Ho * L.Ho, kConvBM);
    a.precision = d->precision;
    return d_launch_conv(d, a, d_tile_width(a.mtiles, L.cout_pad), st);
}

// data gradient of layer l: dact[l-1] = conv^T(draw[l])
int d_conv_dgrad(lwg_discriminator *d, int l, int B, hipStream_t st)
{
    const DLayer &L = d->L[l];
    float *dx = d->L[l - 1].dact;
    ConvArgs a = base_args(L.draw, L.cout_pad, B, L.Ho, L.cout_pad, L.wd, dx, L.cin_pad);
    if (L.stride == 1) {
        a.Hm = a.Wm = a.Ho = a.Wo = L.Hin;
        a.stride = 1; a.pad = 2; a.os = 1;
        a.nphase = 1;
        a.ph[0] = ConvPhase{4, 4, 16, 16 * L.cout_pad, 0, 0, 0, 0, 0};
    } else {
        a.Hm = a.Wm = L.Hin / 2; a.Ho = a.Wo = L.Hin;
        a.stride = 1; a.pad = 0; a.os = 2;
        a.nphase = 4;
        for (int p = 0; p < 4; ++p) {
            const int py = p >> 1, px = p & 1;
            a.ph[p] = ConvPhase{2, 2, 4, 4 * L.cout_pad, (long)p * L.cin_pad * 4 * L.cout_pad, py, px, py ? 0 : -1, px ? 0 : -1};
        }
    }
    a.mtiles = ceil_div((long)B * a.Hm * a.Wm, kConvBM);
    a.precision = d->precision;
    return d_launch_conv(d, a, d_tile_width(a.mtiles, L.cin_pad), st);
}

int d_refresh_dgrad_weights(lwg_discriminator *d, hipStream_t st)
{
    for (size_t l = 1; l < d->L.size(); ++l) {
        const DLayer &L = d->L[l];
        const long total = (long)L.cout_pad * 16 * L.cin_pad;
        dgrad_weights_kernel<<<ceil_div(total, 256), 256, 0, st>>>(d->params + L.w_off, L.cout_pad, L.cin_pad, L.stride, L.wd,
                                                                   L.cin_pad);
        LWG_LAUNCH_CHECK("dgrad_weights_kernel");
    }
    if (d->wd0) {   // first layer, 64 output rows (only needed for the gradient wrt the input image)
        const DLayer &L = d->L[0];
        const long total = (long)L.cout_pad * 16 * 64;
        dgrad_weights_kernel<<<ceil_div(total, 256), 256, 0, st>>>(d->params + L.w_off, L.cout_pad, L.cin_pad, L.stride, d->wd0, 64);
        LWG_LAUNCH_CHECK("dgrad_weights_kernel");
    }
    d->wd_stale = false;
    return LWG_OK;
}

int d_forward(lwg_discriminator *d, const float *x_nchw_a, const float *x_nchw_b, int n_each, hipStream_t st)
{
    const int B = x_nchw_b ? 2 * n_each : n_each;
    const DLayer &L0 = d->L[0];
    int rc = lwg_pack_nhwc(x_nchw_a, n_each, d->input_nc, d->is, d->is, L0.cin_pad, d->x0, st);
    if (rc != LWG_OK) return rc;
    if (x_nchw_b) {
        rc = lwg_pack_nhwc(x_nchw_b, n_each, d->input_nc, d->is, d->is, L0.cin_pad,
                           d->x0 + (size_t)n_each * d->is * d->is * L0.cin_pad, st);
        if (rc != LWG_OK) return rc;
    }
    const float *x = d->x0;
    for (size_t l = 0; l < d->L.size(); ++l) {
        DLayer &L = d->L[l];
        if ((rc = d_conv_forward(d, (int)l, x, B, st)) != LWG_OK) return rc;
        const int HW = L.Ho * L.Ho;
        if (L.norm) {
            in_stats_kernel<<<dim3(L.cout_pad / RS_CH, B), 256, 0, st>>>(L.raw, HW, L.cout_pad, L.stats);
            LWG_LAUNCH_CHECK("in_stats_kernel");
        }
        if (L.act) {
            const long total4 = (long)B * HW * L.cout_pad / 4;
            d_act_kernel<<<ceil_div(total4, 256), 256, 0, st>>>(L.raw, L.norm ? L.stats : nullptr, HW, L.cout_pad, total4, L.actv);
            LWG_LAUNCH_CHECK("d_act_kernel");
            x = L.actv;
        } else {
            x = L.raw;
        }
    }
    return LWG_OK;
}

int d_check(const lwg_discriminator *d, int bs)
{
    if (!d) LWG_FAIL(LWG_ERR_INVALID_ARG, "NULL discriminator handle");
    if (bs <= 0 || bs > d->max_batch) LWG_FAIL(LWG_ERR_STATE, "batch %d outside 1..max_batch=%d", bs, d->max_batch);
    for (size_t l = 0; l < d->L.size(); ++l)
        if (!d->L[l].got_w || !d->L[l].got_b) LWG_FAIL(LWG_ERR_STATE, "discriminator layer %zu has no weights yet", l);
    return LWG_OK;
}

// "model.<idx>.weight|bias" -> layer; sequence indices of the convs: 0, 2, 5, 8, ... (discriminator.py:29-49)
int d_find(const lwg_discriminator *d, const char *key, int *layer, bool *is_bias)
{
    int idx = -1;
    char what[16] = {0};
    if (sscanf(key, "model.%d.%15s", &idx, what) != 2) return -1;
    *is_bias = strcmp(what, "bias") == 0;
    if (!*is_bias && strcmp(what, "weight") != 0) return -1;
    int seq = 0;
    for (size_t l = 0; l < d->L.size(); ++l) {
        if (seq == idx) {
            *layer = (int)l;
            return 0;
        }
        seq += l == 0 ? 2 : 3;
    }
    return -1;
}

}  // namespace
}  // namespace lwg

using namespace lwg;

extern "C" {

int lwg_discriminator_create(lwg_discriminator **out, int input_nc, int ndf, int n_layers, int image_size, int max_batch)
{
    LWG_REQUIRE(out, "discriminator_create: NULL out");
    LWG_REQUIRE(input_nc >= 1 && input_nc <= 8 && ndf == 64 && n_layers >= 1 && n_layers <= 5 && max_batch >= 1,
                "discriminator_create: supported: input_nc <= 8, ndf = 64, 1 <= n_layers <= 5");
    LWG_REQUIRE(image_size >= 32 && (image_size >> n_layers) >= 4 && image_size % (1 << n_layers) == 0,
                "discriminator_create: image_size %d too small for %d stride-2 layers", image_size, n_layers);
    lwg_discriminator *d = new lwg_discriminator();
    d->input_nc = input_nc; d->ndf = ndf; d->n_layers = n_layers; d->is = image_size; d->max_batch = max_batch;
    // discriminator.py:29-49
    int H = image_size, cin = input_nc, mult = 1;
    auto add = [&](int cout, int stride, bool norm, bool act) {
        DLayer L;
        L.cin = cin; L.cin_pad = cin < 8 ? 8 : cin; L.cout = cout; L.cout_pad = cout < 64 ? 64 : cout;
        L.stride = stride; L.norm = norm; L.act = act;
        L.Hin = H; L.Ho = stride == 2 ? H / 2 : H - 1;
        d->L.push_back(L);
        H = L.Ho;
        cin = cout;
    };
    add(ndf, 2, false, true);
    for (int n = 1; n < n_layers; ++n) {
        mult = 1 << n; if (mult > 8) mult = 8;
        add(ndf * mult, 2, true, true);
    }
    mult = 1 << n_layers; if (mult > 8) mult = 8;
    add(ndf * mult, 1, true, true);
    add(1, 1, false, false);
    // a layer's input channel padding is the previous layer's output padding
    for (size_t l = 1; l < d->L.size(); ++l) d->L[l].cin_pad = d->L[l - 1].cout_pad;

    const size_t B = 2 * (size_t)max_batch;
    size_t off = 0, part = 0;
    int rc = LWG_OK;
    for (size_t l = 0; l < d->L.size() && rc == LWG_OK; ++l) {
        DLayer &L = d->L[l];
        L.w_floats = (size_t)L.cout_pad * 16 * L.cin_pad;
        L.w_off = off; off += L.w_floats;
        L.b_off = off; off += L.cout_pad;
        const size_t act = B * L.Ho * L.Ho * L.cout_pad;
        rc = d_alloc(&L.raw, act);
        if (rc == LWG_OK) rc = d_alloc(&L.draw, act);
        if (rc == LWG_OK) rc = d_alloc(&L.dact, act);
        if (rc == LWG_OK && L.act) rc = d_alloc(&L.actv, act);
        if (rc == LWG_OK && L.norm) {
            float *p = nullptr;
            rc = d_alloc(&p, B * L.cout_pad * 2);
            L.stats = reinterpret_cast<float2 *>(p);
            if (rc == LWG_OK) rc = d_alloc(&p, B * L.cout_pad * 2);
            L.sums = reinterpret_cast<float2 *>(p);
        }
        if (rc == LWG_OK && l > 0) rc = d_alloc(&L.wd, L.w_floats);
        const size_t need = 32 * L.w_floats > (size_t)CS_SLICES * L.cout_pad ? 32 * L.w_floats : (size_t)CS_SLICES * L.cout_pad;
        if (need > part) part = need;
    }
    d->nparams = off;
    // split-K partials: a layer never needs more than 32 slices, and only the small layers get that many
    size_t part_cap = 0;
    for (const DLayer &L : d->L) {
        const long P = (long)B * L.Ho * L.Ho;
        const int T = (L.cout_pad % 128 == 0 && L.cin_pad % 128 == 0) ? 128 : 64;   // as launch_wgrad
        const long tiles = (long)(L.cout_pad / T) * ceil_div(L.cin_pad, T) * 16;
        long S = ceil_div(512, tiles);
        if (S > 32) S = 32;
        if (S > ceil_div(P, WG_PX)) S = ceil_div(P, WG_PX);
        const size_t need = (size_t)S * L.w_floats > (size_t)CS_SLICES * L.cout_pad ? (size_t)S * L.w_floats : (size_t)CS_SLICES * L.cout_pad;
        if (need > part_cap) part_cap = need;
    }
    (void)part;
    d->part_floats = part_cap;
    if (rc == LWG_OK) rc = d_alloc(&d->params, off);
    if (rc == LWG_OK) rc = d_alloc(&d->grads, off);
    if (rc == LWG_OK) rc = d_alloc(&d->m, off);
    if (rc == LWG_OK) rc = d_alloc(&d->v, off);
    if (rc == LWG_OK) rc = d_alloc(&d->part, part_cap);
    if (rc == LWG_OK) rc = d_alloc(&d->loss, 4);
    if (rc == LWG_OK) rc = d_alloc(&d->x0, B * (size_t)image_size * image_size * d->L[0].cin_pad);
    if (rc != LWG_OK) {
        lwg_discriminator_destroy(d);
        return rc;
    }
    *out = d;
    return LWG_OK;
}

void lwg_discriminator_destroy(lwg_discriminator *d)
{
    if (!d) return;
    for (DLayer &L : d->L) {
        float *ptrs[] = {L.wd, L.raw, L.actv, L.draw, L.dact, reinterpret_cast<float *>(L.stats), reinterpret_cast<float *>(L.sums)};
        for (float *p : ptrs)
            if (p) (void)hipFree(p);
    }
    if (d->step_dev) (void)hipFree(d->step_dev);
    float *ptrs[] = {d->params, d->grads, d->m, d->v, d->x0, d->part, d->loss, d->wd0, d->dx0};
    for (float *p : ptrs)
        if (p) (void)hipFree(p);
    delete d;
}

int lwg_discriminator_num_params(const lwg_discriminator *d, size_t *n_floats)
{
    LWG_REQUIRE(d && n_floats, "num_params: NULL argument");
    *n_floats = d->nparams;
    return LWG_OK;
}

int lwg_discriminator_load_weight(lwg_discriminator *d, const char *key, const float *data_host, const int64_t *shape, int ndim)
{
    LWG_REQUIRE(d && key && data_host && shape, "discriminator load_weight: NULL argument");
    int l = 0;
    bool is_bias = false;
    if (d_find(d, key, &l, &is_bias) != 0) LWG_FAIL(LWG_ERR_INVALID_ARG, "discriminator: unknown state_dict key '%s'", key);
    DLayer &L = d->L[l];
    if (is_bias) {
        if (ndim != 1 || shape[0] != L.cout) LWG_FAIL(LWG_ERR_INVALID_ARG, "%s: expected shape (%d,)", key, L.cout);
        LWG_HIP(hipMemcpy(d->params + L.b_off, data_host, L.cout * sizeof(float), hipMemcpyHostToDevice));
        L.got_b = true;
    } else {
        if (ndim != 4 || shape[0] != L.cout || shape[1] != L.cin || shape[2] != 4 || shape[3] != 4)
            LWG_FAIL(LWG_ERR_INVALID_ARG, "%s: expected shape (%d,%d,4,4)", key, L.cout, L.cin);
        std::vector<float> h(L.w_floats, 0.f);
        for (int co = 0; co < L.cout; ++co)
            for (int ci = 0; ci < L.cin; ++ci)
                for (int t = 0; t < 16; ++t) h[((size_t)co * 16 + t) * L.cin_pad + ci] = data_host[((size_t)co * L.cin + ci) * 16 + t];
        LWG_HIP(hipMemcpy(d->params + L.w_off, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
        L.got_w = true;
    }
    d->wd_stale = true;
    return LWG_OK;
}

int lwg_discriminator_read_weight(lwg_discriminator *d, const char *key, int from_grads, float *data_host, size_t n_floats)
{
    LWG_REQUIRE(d && key && data_host, "discriminator read_weight: NULL argument");
    int l = 0;
    bool is_bias = false;
    if (d_find(d, key, &l, &is_bias) != 0) LWG_FAIL(LWG_ERR_INVALID_ARG, "discriminator: unknown state_dict key '%s'", key);
    const DLayer &L = d->L[l];
    const float *src = from_grads ? d->grads : d->params;
    LWG_HIP(hipDeviceSynchronize());
    if (is_bias) {
        if (n_floats != (size_t)L.cout) LWG_FAIL(LWG_ERR_INVALID_ARG, "%s: expected %d floats", key, L.cout);
        LWG_HIP(hipMemcpy(data_host, src + L.b_off, L.cout * sizeof(float), hipMemcpyDeviceToHost));
    } else {
        if (n_floats != (size_t)L.cout * L.cin * 16) LWG_FAIL(LWG_ERR_INVALID_ARG, "%s: expected %d floats", key, L.cout * L.cin * 16);
        std::vector<float> h(L.w_floats);
        LWG_HIP(hipMemcpy(h.data(), src + L.w_off, h.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (int co = 0; co < L.cout; ++co)
            for (int ci = 0; ci < L.cin; ++ci)
                for (int t = 0; t < 16; ++t) data_host[((size_t)co * L.cin + ci) * 16 + t] = h[((size_t)co * 16 + t) * L.cin_pad + ci];
    }
    return LWG_OK;
}

int lwg_discriminator_set_precision(lwg_discriminator *d, int mode)
{
    LWG_REQUIRE(d, "discriminator_set_precision: NULL handle");
    if (mode != 0 && mode != 1) LWG_FAIL(LWG_ERR_INVALID_ARG, "discriminator_set_precision: mode %d (0: fp32, 1: bf16x3)", mode);
    d->precision = mode;
    return LWG_OK;
}

int lwg_discriminator_output_size(const lwg_discriminator *d, int *h)
{
    LWG_REQUIRE(d && h, "output_size: NULL argument");
    *h = d->L.back().Ho;
    return LWG_OK;
}

int lwg_discriminator_forward(lwg_discriminator *d, const float *x_nchw, int bs, float *out, lwg_stream_t stream)
{
    int rc = d_check(d, bs);
    if (rc != LWG_OK) return rc;
    LWG_REQUIRE(x_nchw && out, "discriminator forward: NULL argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if ((rc = d_forward(d, x_nchw, nullptr, bs, st)) != LWG_OK) return rc;
    const DLayer &L = d->L.back();
    const long P = (long)bs * L.Ho * L.Ho;
    take_channel0_kernel<<<ceil_div(P, 256), 256, 0, st>>>(L.raw, P, L.cout_pad, out);
    LWG_LAUNCH_CHECK("take_channel0_kernel");
    return LWG_OK;
}

int lwg_discriminator_backward(lwg_discriminator *d, const float *real_nchw, const float *fake_nchw, int bs, float *loss_device,
                               lwg_stream_t stream)
{
    int rc = d_check(d, bs);
    if (rc != LWG_OK) return rc;
    LWG_REQUIRE(real_nchw && fake_nchw, "discriminator backward: NULL argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d->wd_stale && (rc = d_refresh_dgrad_weights(d, st)) != LWG_OK) return rc;
    if ((rc = d_forward(d, real_nchw, fake_nchw, bs, st)) != LWG_OK) return rc;
    const int B = 2 * bs, nl = (int)d->L.size();
    {
        DLayer &L = d->L[nl - 1];
        lsgan_kernel<<<1, 256, 0, st>>>(L.raw, bs, L.Ho * L.Ho, L.cout_pad, L.dact, d->loss, 2, 1.f, -1.f);
        LWG_LAUNCH_CHECK("lsgan_kernel");
        if (loss_device) LWG_HIP(hipMemcpyAsync(loss_device, d->loss, sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    for (int l = nl - 1; l >= 0; --l) {
        DLayer &L = d->L[l];
        const int HW = L.Ho * L.Ho;
        const long total = (long)B * HW * L.cout_pad, P = (long)B * HW;
        if (L.norm) {
            in_bwd_reduce_kernel<<<dim3(L.cout_pad / RS_CH, B), 256, 0, st>>>(L.raw, L.actv, L.dact, L.stats, HW, L.cout_pad, L.sums);
            LWG_LAUNCH_CHECK("in_bwd_reduce_kernel");
        }
        act_bwd_kernel<<<ceil_div(total, 256), 256, 0, st>>>(L.raw, L.act ? L.actv : nullptr, L.dact, L.norm ? L.stats : nullptr,
                                                             L.sums, HW, L.cout_pad, total, L.draw);
        LWG_LAUNCH_CHECK("act_bwd_kernel");
        // bias gradient
        col_sum_partial_kernel<<<dim3(L.cout_pad / 64, CS_SLICES), 256, 0, st>>>(L.draw, P, L.cout_pad, d->part);
        LWG_LAUNCH_CHECK("col_sum_partial_kernel");
        reduce_slices_kernel<<<ceil_div(L.cout_pad, 256), 256, 0, st>>>(d->part, CS_SLICES, L.cout_pad, d->grads + L.b_off);
        LWG_LAUNCH_CHECK("reduce_slices_kernel");
        // weight gradient
        const float *xin = l == 0 ? d->x0 : (d->L[l - 1].act ? d->L[l - 1].actv : d->L[l - 1].raw);
        if ((rc = launch_wgrad(L.draw, L.cout_pad, xin, L.cin_pad, B, L.Hin, L.Hin, L.Ho, L.Ho, L.stride, 1, 4, 16, 0,
                               d->grads + L.w_off, d->part, d->part_floats, st, d->precision)) != LWG_OK)
            return rc;
        if (l > 0 && (rc = d_conv_dgrad(d, l, B, st)) != LWG_OK) return rc;
    }
    return LWG_OK;
}

int lwg_discriminator_buffers(lwg_discriminator *d, float **params, float **grads, size_t *n_floats)
{
    LWG_REQUIRE(d, "discriminator buffers: NULL handle");
    if (params) *params = d->params;
    if (grads) *grads = d->grads;
    if (n_floats) *n_floats = d->nparams;
    return LWG_OK;
}

int lwg_discriminator_adam_step(lwg_discriminator *d, float lr, float beta1, float beta2, float eps, lwg_stream_t stream)
{
    LWG_REQUIRE(d, "adam_step: NULL handle");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    d->step += 1;
    const float bc1 = 1.f - powf(beta1, (float)d->step), bc2 = 1.f - powf(beta2, (float)d->step);
    if (d->device_step) {
        step_inc_kernel<<<1, 1, 0, st>>>(d->step_dev, beta1, beta2);
        LWG_LAUNCH_CHECK("step_inc_kernel");
    }
    adam_kernel<<<ceil_div((long)d->nparams, 256), 256, 0, st>>>(d->params, d->grads, d->m, d->v, (long)d->nparams, lr, beta1,
                                                                 beta2, eps, bc1, sqrtf(bc2), d->device_step ? d->step_dev : nullptr);
    LWG_LAUNCH_CHECK("adam_kernel");
    return d_refresh_dgrad_weights(d, st);
}

int lwg_discriminator_use_device_step(lwg_discriminator *d, int on, float beta1, float beta2)
{
    LWG_REQUIRE(d, "use_device_step: NULL handle");
    if (on) {
        if (!d->step_dev) LWG_HIP(hipMalloc(reinterpret_cast<void **>(&d->step_dev), sizeof(AdamStep)));
        const AdamStep h = {d->step, pow((double)beta1, (double)d->step), pow((double)beta2, (double)d->step)};
        LWG_HIP(hipMemcpy(d->step_dev, &h, sizeof(h), hipMemcpyHostToDevice));
    } else if (d->device_step) {
        AdamStep h;
        LWG_HIP(hipMemcpy(&h, d->step_dev, sizeof(h), hipMemcpyDeviceToHost));   // replays advanced only the device copy
        d->step = h.step;
    }
    d->device_step = on != 0;
    return LWG_OK;
}

// ------------------------------------------------------------------------------------------------
// Op-level convolution gradients (building blocks of the generator-side training step).
namespace {

struct ConvGeom { int Ho, Wo; size_t w_floats; };

int conv_geom(const lwg_conv2d_desc *d, ConvGeom *g)
{
    LWG_REQUIRE(d && g, "conv2d: NULL descriptor");
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    if (!pow2(d->Cin) || !pow2(d->Cout) || d->Cin < 8 || d->Cout < 8)
        LWG_FAIL(LWG_ERR_UNSUPPORTED, "conv2d: channel counts must be powers of two >= 8 (Cin=%d, Cout=%d)", d->Cin, d->Cout);
    if (d->transposed) {
        if (d->k != 3 || d->stride != 2 || d->pad != 1)
            LWG_FAIL(LWG_ERR_UNSUPPORTED, "conv2d: transposed convs are ConvTranspose2d(k3, s2, p1, output_padding 1)");
        g->Ho = 2 * d->H; g->Wo = 2 * d->W;
    } else {
        if (d->stride != 1 && d->stride != 2) LWG_FAIL(LWG_ERR_UNSUPPORTED, "conv2d: stride must be 1 or 2");
        if (d->k < 1 || d->k > 7) LWG_FAIL(LWG_ERR_UNSUPPORTED, "conv2d: kernel size 1..7");
        g->Ho = (d->H + 2 * d->pad - d->k) / d->stride + 1;
        g->Wo = (d->W + 2 * d->pad - d->k) / d->stride + 1;
    }
    if (d->N < 1 || g->Ho < 1 || g->Wo < 1) LWG_FAIL(LWG_ERR_INVALID_ARG, "conv2d: empty tensor");
    g->w_floats = (size_t)d->Cin * d->Cout * d->k * d->k;
    return LWG_OK;
}

// Scratch of the bf16x3 route (precision 1): the operand tensor and the weight matrices are re-written in the
// split-bf16 format of conv.h ([hi x32 | lo x32] per 32 values, same offsets as fp32) and the DMA-fed kernel of the
// inference path runs on them.  The zero run the kernel reads out-of-image taps from sits right behind the operand.
struct SplitWs {
    float *w = nullptr;       // split weight matrices (as many floats as the fp32 ones)
    float *x = nullptr;       // split operand tensor
    float *zeros = nullptr;   // one pixel's worth of channels (the kernel reads the zero run at the channel offset it is at)
    size_t zero_floats = 0;
};

// (the threads behind the tensor clear the zero run the conv kernel reads out-of-image taps from: one launch instead of a
// kernel and a memset per convolution)
__global__ __launch_bounds__(256) void split_pack_kernel(const float4 *__restrict__ src, float *__restrict__ dst, size_t n4,
                                                         float4 *__restrict__ zeros, size_t z4)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) {
        if (i - n4 < z4) zeros[i - n4] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const float4 v = src[i];
    const size_t e = i * 4;
    const size_t soff = (e >> 5) * 32 + ((e & 31) >> 1);   // 4-byte units: 8 bytes of hi, 8 of lo 64 B further
    bf16x4_t h, l;
    h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
    l[0] = (__bf16)(v.x - (float)h[0]); l[1] = (__bf16)(v.y - (float)h[1]);
    l[2] = (__bf16)(v.z - (float)h[2]); l[3] = (__bf16)(v.w - (float)h[3]);
    *reinterpret_cast<bf16x4_t *>(dst + soff) = h;
    *reinterpret_cast<bf16x4_t *>(dst + soff + 16) = l;
}

int split_pack(const float *src, float *dst, size_t n, hipStream_t st, float *zeros = nullptr, size_t zero_floats = 0)
{
    if (n % 32 || zero_floats % 4) LWG_FAIL(LWG_ERR_INVALID_ARG, "split: %zu floats is not a whole number of 32-value groups", n);
    const size_t z4 = zeros ? zero_floats / 4 : 0;
    split_pack_kernel<<<(unsigned)ceil_div((long)(n / 4 + z4), 256), 256, 0, st>>>(reinterpret_cast<const float4 *>(src), dst, n / 4,
                                                                                  reinterpret_cast<float4 *>(zeros), z4);
    LWG_LAUNCH_CHECK("split_pack_kernel");
    return LWG_OK;
}

// the DMA-fed bf16x3 kernel wants: 32-channel granularity on the reduction side, whole 128-pixel tiles per image,
// at most 32 taps, no bias
bool split_route_ok(int precision, int Cin, int Hm, int Wm, int taps, const float *bias)
{
    return precision == 1 && !bias && Cin % 32 == 0 && (Hm * Wm) % kConvBM == 0 && taps <= 32;
}

// operand + weights -> split copies; fills in the ConvArgs fields of the bf16x3 route
// (the weight matrices were written split, into sw.w, by relayout)
int to_split_route(ConvArgs &a, const float *x, size_t x_floats, const SplitWs &sw, hipStream_t st)
{
    int rc;
    if ((rc = split_pack(x, sw.x, x_floats, st, sw.zeros, sw.zero_floats)) != LWG_OK) return rc;
    a.x = sw.x;
    a.w_split = sw.w;
    a.zeros = sw.zeros;
    a.precision = 1;
    a.general = 0;
    a.mtiles = a.N * a.Hm * a.Wm / kConvBM;
    return LWG_OK;
}

// plain conv through the implicit GEMM: x (N,H,W,Cin) * wmat [Cout][k*k][Cin] -> y (N,Ho,Wo,Cout)
int op_conv(const float *x, int N, int H, int W, int Cin, const float *wmat, const float *bias, int Cout, int k, int stride,
            int pad, float *y, hipStream_t st, int precision = 0, const SplitWs *sw = nullptr)
{
    if (Cout % 64) LWG_FAIL(LWG_ERR_UNSUPPORTED, "conv2d: this direction needs a multiple of 64 output channels, got %d", Cout);
    ConvArgs a = base_args(x, Cin, N, H, Cin, wmat, y, Cout);
    a.W = W;
    a.Hm = a.Ho = (H + 2 * pad - k) / stride + 1;
    a.Wm = a.Wo = (W + 2 * pad - k) / stride + 1;
    a.stride = stride; a.pad = pad; a.os = 1;
    a.bias = bias;
    a.nphase = 1;
    a.ph[0] = ConvPhase{k, k, k * k, (int)align_up((size_t)k * k * Cin, kConvBK), 0, 0, 0, 0, 0};   // wmat rows have this pitch
    a.mtiles = ceil_div((long)N * a.Hm * a.Wm, kConvBM);
    if (sw) {   // the caller checked split_route_ok and had the weights written split
        if (!split_route_ok(precision, Cin, a.Hm, a.Wm, k * k, bias) || wmat != sw->w) LWG_FAIL(LWG_ERR_STATE, "conv2d: bf16x3 route misuse");
        const int rc = to_split_route(a, x, (size_t)N * H * W * Cin, *sw, st);
        if (rc != LWG_OK) return rc;
        // 128-channel tiles once they still give every CU a workgroup, else twice as many 64-channel ones (the 32x32
        // trunk at batch 4 is 128 tiles of 128)
        const bool wide = Cout % 128 == 0 && (long)a.mtiles * (Cout / 128) >= device_cu_count();
        return launch_conv_igemm(a, wide ? 128 : 64, st);
    }
    // general mode, plain fp32 tensors: bf16x3 by splitting inside the kernel (the 8-channel stem and the heads' data gradient).
    // Biased convolutions -- the loss networks -- stay fp32 in both modes.
    a.precision = (precision == 1 && !bias) ? 1 : 0;
    return launch_conv_igemm(a, (Cout % 128 == 0 && Cin >= kConvBK) ? 128 : 64, st);
}

// "transposed conv forward" through the 4-phase decomposition: x (N,H,W,Cin) * phase matrices -> y (N,2H,2W,Cout)
int op_convT(const float *x, int N, int H, int W, int Cin, const float *wph, int Cout, float *y, hipStream_t st,
             int precision = 0, const SplitWs *sw = nullptr)
{
    if (Cout % 64 || Cin < kConvBK) LWG_FAIL(LWG_ERR_UNSUPPORTED, "conv2d: this direction needs Cout %% 64 == 0 and Cin >= 32");
    ConvArgs a = base_args(x, Cin, N, H, Cin, wph, y, Cout);
    a.W = W;
    a.Hm = H; a.Wm = W; a.Ho = 2 * H; a.Wo = 2 * W;
    a.stride = 1; a.pad = 0; a.os = 2;
    a.nphase = 4;
    long off = 0;
    for (int p = 0; p < 4; ++p) {
        const int py = p >> 1, px = p & 1, nt = (1 + py) * (1 + px);
        a.ph[p] = ConvPhase{1 + py, 1 + px, nt, nt * Cin, off, py, px, 0, 0};
        off += (long)Cout * nt * Cin;
    }
    a.mtiles = ceil_div((long)N * H * W, kConvBM);
    if (sw) {
        if (!split_route_ok(precision, Cin, H, W, 4, nullptr) || wph != sw->w) LWG_FAIL(LWG_ERR_STATE, "conv2d: bf16x3 route misuse");
        const int rc = to_split_route(a, x, (size_t)N * H * W * Cin, *sw, st);
        if (rc != LWG_OK) return rc;
        a.fuse_phases = 0;
        return launch_conv_igemm(a, Cout % 128 == 0 ? 128 : 64, st);
    }
    return launch_conv_igemm(a, Cout % 128 == 0 ? 128 : 64, st);
}

// `pitch` (floats, 0 = dense): row pitch of the matrix -- rows are output channels: A of them with taps*B entries in
// mode 0, B of them with taps*A entries in mode 1
int relayout(const float *w, float *dst, int mode, int A, int B, int taps, long total, hipStream_t st, int pitch = 0,
             bool split = false)
{
    const int rows = mode == 1 ? B : A, dense = taps * (mode == 1 ? A : B);
    if (split && pitch && pitch != dense) LWG_FAIL(LWG_ERR_STATE, "relayout: split matrices have no row padding");
    if (pitch && pitch != dense) LWG_HIP(hipMemsetAsync(dst, 0, (size_t)rows * pitch * sizeof(float), st));
    // (an LDS-tiled transpose -- 32 x 32 x taps tiles, contiguous reads, 128-byte writes -- measured SLOWER in the training iteration:
    // 25.3 vs 24.3 ms at 256x256 batch 4, profiles/r05_train_ab.md: 256 workgroups of 36 serial passes against 9216 independent ones)
    weight_layout_kernel<<<ceil_div(total, 256), 256, 0, st>>>(w, dst, mode, A, B, taps, total, pitch ? pitch : dense, split ? 1 : 0);
    LWG_LAUNCH_CHECK("weight_layout_kernel");
    return LWG_OK;
}

}  // namespace

namespace {
// floats of the fp32 part of the workspace: one re-laid-out weight matrix, or up to 32 split-K partial gradients + the
// column-sum scratch (+ row padding of the matrix)
// split-K partial gradients: 32 slices always, up to 256 for small filters (the 64-channel 256x256 layers have a quarter
// of a million pixels to slice and 36 k weights), within 32 MiB
size_t wgrad_part_floats(const ConvGeom &g)
{
    const size_t want = (size_t)kWgradMaxSlices * g.w_floats, lim = (size_t)8 << 20;
    const size_t least = 32 * g.w_floats;
    return want < lim ? want : (least > lim ? least : lim);
}

size_t base_ws_floats(const lwg_conv2d_desc *d, const ConvGeom &g)
{
    const size_t cmax = (size_t)(d->Cout > d->Cin ? d->Cout : d->Cin);
    return align_up(wgrad_part_floats(g) + g.w_floats + (32 + kConvBK) * cmax + (size_t)CS_SLICES * cmax, (size_t)32);
}

// the larger of the two tensors a conv of this geometry touches (the forward's operand is x, the data gradient's is dy)
size_t operand_floats(const lwg_conv2d_desc *d, const ConvGeom &g)
{
    const size_t xin = (size_t)d->N * d->H * d->W * d->Cin, yout = (size_t)d->N * g.Ho * g.Wo * d->Cout;
    return align_up(xin > yout ? xin : yout, (size_t)32);
}

size_t zero_run_floats(const lwg_conv2d_desc *d) { return (size_t)(d->Cout > d->Cin ? d->Cout : d->Cin) + 32; }

SplitWs carve_split(const lwg_conv2d_desc *d, const ConvGeom &g, void *ws)
{
    SplitWs sw;
    // 128-byte aligned: the split format is addressed in 128-byte groups
    float *p = reinterpret_cast<float *>(align_up((size_t)(uintptr_t)ws, (size_t)128)) + base_ws_floats(d, g);
    sw.w = p;
    sw.x = p + align_up(g.w_floats, (size_t)32);
    sw.zeros = sw.x + operand_floats(d, g);
    sw.zero_floats = zero_run_floats(d);
    return sw;
}
}  // namespace

size_t lwg_conv2d_workspace_bytes(const lwg_conv2d_desc *d)
{
    ConvGeom g;
    if (conv_geom(d, &g) != LWG_OK) return 0;
    size_t floats = base_ws_floats(d, g);
    if (d->precision == 1) floats += align_up(g.w_floats, (size_t)32) + operand_floats(d, g) + zero_run_floats(d) + 32;   // + alignment slack
    return floats * sizeof(float);
}

int lwg_conv2d_forward(const lwg_conv2d_desc *d, const float *x, const float *w, const float *bias, float *y, void *ws,
                       size_t ws_bytes, lwg_stream_t stream)
{
    ConvGeom g;
    int rc = conv_geom(d, &g);
    if (rc != LWG_OK) return rc;
    LWG_REQUIRE(x && w && y && ws, "conv2d_forward: NULL argument");
    if (ws_bytes < lwg_conv2d_workspace_bytes(d)) LWG_FAIL(LWG_ERR_INVALID_ARG, "conv2d_forward: workspace too small");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float *wm = static_cast<float *>(ws);
    // bf16x3 route: the weight matrices are written split straight away, into the split part of the workspace
    SplitWs sw_, *sw = nullptr;
    const bool split = d->transposed ? split_route_ok(d->precision, d->Cin, d->H, d->W, 4, nullptr)
                                     : split_route_ok(d->precision, d->Cin, g.Ho, g.Wo, d->k * d->k, bias);
    if (split) { sw_ = carve_split(d, g, ws); sw = &sw_; wm = sw_.w; }
    if (d->transposed) {
        if (bias) LWG_FAIL(LWG_ERR_UNSUPPORTED, "conv2d_forward: bias on a transposed conv");
        if ((rc = relayout(w, wm, 2, d->Cin, d->Cout, 9, (long)g.w_floats, st, 0, split)) != LWG_OK) return rc;
        return op_convT(x, d->N, d->H, d->W, d->Cin, wm, d->Cout, y, st, d->precision, sw);
    }
    const int pitch = (int)align_up((size_t)d->k * d->k * d->Cin, kConvBK);   // 7x7x8 = 392 -> 416
    if ((rc = relayout(w, wm, 0, d->Cout, d->Cin, d->k * d->k, (long)g.w_floats, st, pitch, split)) != LWG_OK) return rc;
    return op_conv(x, d->N, d->H, d->W, d->Cin, wm, bias, d->Cout, d->k, d->stride, d->pad, y, st, d->precision, sw);
}

int lwg_conv2d_backward_data(const lwg_conv2d_desc *d, const float *dy, const float *w, float *dx, void *ws, size_t ws_bytes,
                             lwg_stream_t stream)
{
    ConvGeom g;
    int rc = conv_geom(d, &g);
    if (rc != LWG_OK) return rc;
    LWG_REQUIRE(dy && w && dx && ws, "conv2d_backward_data: NULL argument");
    if (ws_bytes < lwg_conv2d_workspace_bytes(d)) LWG_FAIL(LWG_ERR_INVALID_ARG, "conv2d_backward_data: workspace too small");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float *wm = static_cast<float *>(ws);
    // the conv that computes dx: its reduction side is Cout, its output grid the input's (the phase grid for stride 2)
    SplitWs sw_, *sw = nullptr;
    const bool via_convT = !d->transposed && d->stride == 2;
    const bool split = via_convT ? split_route_ok(d->precision, d->Cout, g.Ho, g.Wo, 4, nullptr)
                                 : split_route_ok(d->precision, d->Cout, d->H, d->W, d->k * d->k, nullptr);
    if (split) { sw_ = carve_split(d, g, ws); sw = &sw_; wm = sw_.w; }
    if (d->transposed) {
        // gradient of ConvTranspose2d(k3,s2,p1,op1) = Conv2d(k3,s2,p1) of dy whose OIHW tensor is the (Cin,Cout,3,3) one
        if ((rc = relayout(w, wm, 0, d->Cin, d->Cout, 9, (long)g.w_floats, st, 0, split)) != LWG_OK) return rc;
        return op_conv(dy, d->N, g.Ho, g.Wo, d->Cout, wm, nullptr, d->Cin, 3, 2, 1, dx, st, d->precision, sw);
    }
    if (d->stride == 1) {
        if (2 * d->pad != d->k - 1) LWG_FAIL(LWG_ERR_UNSUPPORTED, "conv2d_backward_data: stride-1 convs need 'same' padding");
        // rows of k*k*Cout entries, padded to the reduction slice (the heads' data gradient: 49 x 8 = 392 -> 416)
        const int pitch = (int)align_up((size_t)d->k * d->k * d->Cout, kConvBK);
        if ((rc = relayout(w, wm, 1, d->Cout, d->Cin, d->k * d->k, (long)g.w_floats, st, pitch, split)) != LWG_OK) return rc;
        return op_conv(dy, d->N, g.Ho, g.Wo, d->Cout, wm, nullptr, d->Cin, d->k, 1, d->k - 1 - d->pad, dx, st, d->precision, sw);
    }
    if (d->k != 3 || d->pad != 1 || (d->H & 1) || (d->W & 1))
        LWG_FAIL(LWG_ERR_UNSUPPORTED, "conv2d_backward_data: stride-2 convs are k3 p1 on even sizes (the discriminator's k4 has its own path)");
    // gradient of Conv2d(k3,s2,p1) = ConvTranspose2d(k3,s2,p1,op1) forward of dy with the same tensor read as (Cout,Cin,3,3)
    if ((rc = relayout(w, wm, 2, d->Cout, d->Cin, 9, (long)g.w_floats, st, 0, split)) != LWG_OK) return rc;
    return op_convT(dy, d->N, g.Ho, g.Wo, d->Cout, wm, d->Cin, dx, st, d->precision, sw);
}

int lwg_conv2d_backward_weight(const lwg_conv2d_desc *d, const float *x, const float *dy, float *dw, float *dbias, void *ws,
                               size_t ws_bytes, lwg_stream_t stream)
{
    ConvGeom g;
    int rc = conv_geom(d, &g);
    if (rc != LWG_OK) return rc;
    LWG_REQUIRE(x && dy && dw && ws, "conv2d_backward_weight: NULL argument");
    if (ws_bytes < lwg_conv2d_workspace_bytes(d)) LWG_FAIL(LWG_ERR_INVALID_ARG, "conv2d_backward_weight: workspace too small");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float *part = static_cast<float *>(ws);
    // the conv whose weight gradient this is: out (O channels, grad `go`) = conv(in (I channels), stride, pad)
    // plain conv: in = x, go = dy.  ConvTranspose2d: the roles swap (dW[ci][co] pairs x[ci] with dy[co] taken 2i-1+kh).
    const float *in = d->transposed ? dy : x, *go = d->transposed ? x : dy;
    const int I = d->transposed ? d->Cout : d->Cin, O = d->transposed ? d->Cin : d->Cout;
    const int Hin = d->transposed ? g.Ho : d->H, Win = d->transposed ? g.Wo : d->W;
    const int Hg = d->transposed ? d->H : g.Ho, Wg = d->transposed ? d->W : g.Wo;
    const int stride = d->transposed ? 2 : d->stride, pad = d->pad, taps = d->k * d->k;
    if (O % 64) LWG_FAIL(LWG_ERR_UNSUPPORTED, "conv2d_backward_weight: needs a multiple of 64 channels on the gradient side, got %d", O);
    const long P = (long)d->N * Hg * Wg;
    if ((rc = launch_wgrad(go, O, in, I, d->N, Hin, Win, Hg, Wg, stride, pad, d->k, taps, 1, dw, part, wgrad_part_floats(g), st, d->precision)) != LWG_OK) return rc;
    if (dbias) {
        if (d->transposed) LWG_FAIL(LWG_ERR_UNSUPPORTED, "conv2d_backward_weight: bias gradient of a transposed conv");
        float *cs = part + wgrad_part_floats(g);
        col_sum_partial_kernel<<<dim3(d->Cout / 64, CS_SLICES), 256, 0, st>>>(dy, P, d->Cout, cs);
        LWG_LAUNCH_CHECK("col_sum_partial_kernel");
        reduce_slices_kernel<<<ceil_div(d->Cout, 256), 256, 0, st>>>(cs, CS_SLICES, d->Cout, dbias);
        LWG_LAUNCH_CHECK("reduce_slices_kernel");
    }
    return LWG_OK;
}

// ------------------------------------------------------------------------------------------------
// The regression heads in the training step (generator.py:142-152: 7x7 conv of the 64-channel decoder output to 3
// colour channels + 1 mask channel).  As 64-output convolutions (their first form here) they cost 16 x their arithmetic.
//   forward        : the inference path's fp32 heads kernel (conv.hip) behind an identity InstanceNorm
//   data gradient  : lwg_conv2d_backward_data with Cout = 8 (a 7x7 conv of an 8-channel tensor: the stem's kernel)
//   weight gradient: below
namespace {

// wh[tap][c][o] (o < 4) from w (R >= 4 rows, 64, 7, 7)
__global__ __launch_bounds__(256) void heads_weight_kernel(const float *__restrict__ w, float *__restrict__ wh)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 49 * 64 * 4) return;
    const int o = i & 3, c = (i >> 2) & 63, tap = i >> 8;
    wh[i] = w[((size_t)o * 64 + c) * 49 + tap];
}
__global__ __launch_bounds__(256) void fill_identity_norm_kernel(float2 *ss, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ss[i] = make_float2(1.f, 0.f);
}

// dW[o][c][ky][kx] = sum_p dY[p][o] * X[p + (ky-3, kx-3)][c]  with dY (N,H,W,8) and X (N,H,W,64), as the GEMM
//   OUT[m = (ky', kx', o)][c] = sum_q dY[q + (ky'-3, kx'-3)][o] * X[q][c]      (q = p + tap - 3, so tap' = 6 - tap)
// M = 49 x 8 = 392 (13 tiles of 32), N = 64, K = the pixels.  A workgroup walks a slice of the pixels in steps of 32
// consecutive pixels of one image row: the A operand of a step is a 7 x 38-pixel band of dY in LDS (8.5 KiB) read in
// place -- row m of the im2col matrix is the band at offset ky'*304 + kx'*8 + o, stride 8 per pixel -- so nothing is
// gathered; v_mfma_f32_32x32x2_f32 takes one ds_read_b32 per lane from it.  Four waves: wave (wm, wn) owns column tile wn
// and the row tiles wm, wm+2, ...  The slices' partial results are summed in a fixed order by heads_wgrad_reduce_kernel,
// which also undoes the tap flip and writes PyTorch's (8,64,7,7) layout.
constexpr int HG_PX = 32, HG_BAND_ROW = (HG_PX + 6) * 8, HG_BAND = 7 * HG_BAND_ROW, HG_XP = 64 + 4, HG_M = 392, HG_MT = 13;
constexpr int HG_BUF = HG_BAND + HG_PX * HG_XP;   // floats per LDS buffer

__global__ __launch_bounds__(256) void wgrad_heads_kernel(const float *__restrict__ x, const float *__restrict__ dy, int N, int H,
                                                          int W, long steps_per_slice, float *__restrict__ part)
{
    __shared__ __attribute__((aligned(16))) float sm[2 * HG_BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 1, wm = wave >> 1;
    const int row_steps = (W + HG_PX - 1) / HG_PX;
    const long total = (long)N * H * row_steps;
    const long g0 = (long)blockIdx.x * steps_per_slice, g1 = g0 + steps_per_slice < total ? g0 + steps_per_slice : total;

    float4 rd[3], rx[2];
    auto load = [&](long g) {
        const int n = (int)(g / ((long)H * row_steps));
        const int r = (int)(g - (long)n * H * row_steps);
        const int qy = r / row_steps, qx0 = (r - qy * row_steps) * HG_PX;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int idx = tid + 256 * k;   // band: 7 rows x 76 float4 (38 pixels x 8 channels)
            rd[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < 7 * 76) {
                const int br = idx / 76, f4 = idx - br * 76;
                const int sy = qy + br - 3, sx = qx0 - 3 + (f4 >> 1);
                if ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W)
                    rd[k] = ld4(dy + (((size_t)n * H + sy) * W + sx) * 8 + (f4 & 1) * 4);
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = tid + 256 * k;   // x tile: 32 pixels x 16 float4
            const int px = idx >> 4, f4 = idx & 15;
            rx[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qx0 + px < W) rx[k] = ld4(x + (((size_t)n * H + qy) * W + qx0 + px) * 64 + f4 * 4);
        }
    };
    auto store = [&](int buf) {
        float *b = sm + buf * HG_BUF;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int idx = tid + 256 * k;
            if (idx < 7 * 76) *reinterpret_cast<float4 *>(b + idx * 4) = rd[k];
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = tid + 256 * k;
            *reinterpret_cast<float4 *>(b + HG_BAND + (idx >> 4) * HG_XP + (idx & 15) * 4) = rx[k];
        }
    };
    // LDS offsets of this lane's A rows: m -> (tap' = m >> 3, o = m & 7); rows past 391 repeat the last one (never stored)
    int abase[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        int m = (wm + 2 * i) * 32 + (lane & 31);
        if (m > HG_M - 1) m = HG_M - 1;
        const int tap = m >> 3, o = m & 7, ky = tap / 7, kx = tap - ky * 7;
        abase[i] = ky * HG_BAND_ROW + kx * 8 + o + 8 * (lane >> 5);
    }
    const int bbase = HG_BAND + (lane >> 5) * HG_XP + wn * 32 + (lane & 31);
    f32x16 acc[7];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    int buf = 0;
    if (g0 < g1) {
        load(g0);
        store(0);
    }
    __syncthreads();
    for (long g = g0; g < g1; ++g) {
        const bool more = g + 1 < g1;
        if (more) load(g + 1);
        const float *b = sm + buf * HG_BUF;
#pragma unroll
        for (int kk = 0; kk < HG_PX / 2; ++kk) {
            const float bv = b[bbase + 2 * kk * HG_XP];
            float av[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) av[i] = b[abase[i] + 16 * kk];
#pragma unroll
            for (int i = 0; i < 7; ++i)
                if (wm + 2 * i < HG_MT) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv, acc[i], 0, 0, 0);
        }
        if (more) store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // C/D layout: col = lane&31 -> channel, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -> m inside the tile
    float *o = part + (size_t)blockIdx.x * HG_M * 64;
    const int c = wn * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        if (wm + 2 * i >= HG_MT) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (wm + 2 * i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m < HG_M) o[(size_t)m * 64 + c] = acc[i][r];
        }
    }
}

__global__ __launch_bounds__(256) void heads_wgrad_reduce_kernel(const float *__restrict__ part, int S, float *__restrict__ dw)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HG_M * 64) return;
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += part[(size_t)k * HG_M * 64 + i];
    const int c = i & 63, m = i >> 6, o = m & 7, tap = m >> 3;
    const int ky = 6 - tap / 7, kx = 6 - tap % 7;
    dw[(((size_t)o * 64 + c) * 7 + ky) * 7 + kx] = s;
}

constexpr long kHeadsMaxSlices = 512;   // two workgroups per CU of a 256-CU part
long heads_wgrad_slices(int N, int H, int W, long *steps_per_slice)
{
    const long total = (long)N * H * ceil_div(W, HG_PX);
    long S = kHeadsMaxSlices;
    if (S > total) S = total;
    *steps_per_slice = ceil_div(total, S);
    return ceil_div(total, *steps_per_slice);
}
constexpr size_t kHeadsWFloats = 49 * 64 * 4;

}  // namespace

size_t lwg_heads_workspace_bytes(int N, int H, int W)
{
    if (N < 1 || H < 1 || W < 1) return 0;
    // the larger of: forward (re-laid-out weights + identity scale/shift) and weight gradient (slice partials)
    const size_t fwd = kHeadsWFloats + (size_t)N * 64 * 2, wg = (size_t)kHeadsMaxSlices * HG_M * 64;
    return (fwd > wg ? fwd : wg) * sizeof(float);
}

int lwg_heads_forward(const float *x, int N, int H, int W, const float *w, int w_rows, float *color, float *mask, void *ws,
                      size_t ws_bytes, lwg_stream_t stream)
{
    LWG_REQUIRE(x && w && ws && (color || mask), "heads_forward: NULL argument");
    if (w_rows < 4) LWG_FAIL(LWG_ERR_INVALID_ARG, "heads_forward: the weight tensor needs >= 4 rows (3 colour + 1 mask), got %d", w_rows);
    if (ws_bytes < lwg_heads_workspace_bytes(N, H, W)) LWG_FAIL(LWG_ERR_INVALID_ARG, "heads_forward: workspace too small");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float *wh = static_cast<float *>(ws);
    float2 *ss = reinterpret_cast<float2 *>(wh + kHeadsWFloats);
    heads_weight_kernel<<<ceil_div((long)kHeadsWFloats, 256), 256, 0, st>>>(w, wh);
    LWG_LAUNCH_CHECK("heads_weight_kernel");
    fill_identity_norm_kernel<<<ceil_div((long)N * 64, 256), 256, 0, st>>>(ss, N * 64);
    LWG_LAUNCH_CHECK("fill_identity_norm_kernel");
    HeadsArgs h = {};
    h.x = x; h.N = N; h.H = H; h.W = W;
    h.scale_shift = ss;
    h.wh = wh;
    h.color = color;
    h.mask = mask;
    return launch_heads(h, st);
}

int lwg_heads_backward_weight(const float *x, const float *dy8, int N, int H, int W, float *dw, void *ws, size_t ws_bytes,
                              lwg_stream_t stream)
{
    LWG_REQUIRE(x && dy8 && dw && ws, "heads_backward_weight: NULL argument");
    if (N < 1 || H < 1 || W < 1) LWG_FAIL(LWG_ERR_INVALID_ARG, "heads_backward_weight: empty tensor");
    if (ws_bytes < lwg_heads_workspace_bytes(N, H, W)) LWG_FAIL(LWG_ERR_INVALID_ARG, "heads_backward_weight: workspace too small");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    long per = 0;
    const long S = heads_wgrad_slices(N, H, W, &per);
    float *part = static_cast<float *>(ws);
    wgrad_heads_kernel<<<(unsigned)S, 256, 0, st>>>(x, dy8, N, H, W, per, part);
    LWG_LAUNCH_CHECK("wgrad_heads_kernel");
    heads_wgrad_reduce_kernel<<<ceil_div((long)HG_M * 64, 256), 256, 0, st>>>(part, (int)S, dw);
    LWG_LAUNCH_CHECK("heads_wgrad_reduce_kernel");
    return LWG_OK;
}

/* InstanceNorm2d(affine=True, eps 1e-5, biased variance) [+ ReLU], NHWC fp32.  stats: (N, C) float2 (mean, rstd), written by
 * forward and read by backward. */
size_t lwg_instance_norm_scratch_bytes(int N, int HW, int C)
{
    if (N < 1 || HW < 1 || C < RS_CH) return 0;
    const int S4 = C % RS4_CH == 0 ? in_bwd_slabs4(N, C, HW) : 1, S1 = in_slabs(N, C, HW);
    return (size_t)N * (S4 > S1 ? S4 : S1) * C * 2 * sizeof(double) + (size_t)N * C * 2 * sizeof(float);
}

int lwg_instance_norm_forward(const float *x, int N, int HW, int C, const float *gamma, const float *beta, int relu, float *y,
                              float *stats, void *scratch, lwg_stream_t stream)
{
    LWG_REQUIRE(x && gamma && beta && y && stats && scratch, "instance_norm_forward: NULL argument");
    if (C % RS_CH) LWG_FAIL(LWG_ERR_UNSUPPORTED, "instance_norm: C=%d must be a multiple of %d", C, RS_CH);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float2 *s2 = reinterpret_cast<float2 *>(stats);
    const long total = (long)N * HW * C;
    if (C % RS4_CH == 0) {
        const int S4 = in_bwd_slabs4(N, C, HW);
        double *part = static_cast<double *>(scratch);
        in_stats4_kernel<<<dim3(C / RS4_CH, N, S4), 256, 0, st>>>(x, HW, C, S4, part);
        LWG_LAUNCH_CHECK("in_stats4_kernel");
        in_stats_final4_kernel<<<ceil_div((long)N * C, 64), 256, 0, st>>>(part, N, C, S4, HW, s2);
        LWG_LAUNCH_CHECK("in_stats_final4_kernel");
        in_apply4_kernel<<<ceil_div(total / 4, 256), 256, 0, st>>>(x, s2, gamma, beta, relu, HW, C, total / 4, y);
        LWG_LAUNCH_CHECK("in_apply4_kernel");
        return LWG_OK;
    }
    const int S = in_slabs(N, C, HW);
    if (S == 1) {
        in_stats_kernel<<<dim3(C / RS_CH, N), 256, 0, st>>>(x, HW, C, s2);
        LWG_LAUNCH_CHECK("in_stats_kernel");
    } else {
        double *part = static_cast<double *>(scratch);
        in_stats_partial_kernel<<<dim3(C / RS_CH, N, S), 256, 0, st>>>(x, HW, C, S, part);
        LWG_LAUNCH_CHECK("in_stats_partial_kernel");
        in_stats_final_kernel<<<ceil_div((long)N * C, 256), 256, 0, st>>>(part, N, C, S, HW, s2);
        LWG_LAUNCH_CHECK("in_stats_final_kernel");
    }
    in_apply_kernel<<<ceil_div(total, 256), 256, 0, st>>>(x, s2, gamma, beta, relu, HW, C, total, y);
    LWG_LAUNCH_CHECK("in_apply_kernel");
    return LWG_OK;
}

/* y: the forward output when it went through the ReLU (its sign is the mask), NULL without activation.
 * scratch: (N, C) float2.  dgamma / dbeta: (C,), overwritten. */
int lwg_instance_norm_backward(const float *x, const float *y, const float *dy, const float *stats, const float *gamma, int N,
                               int HW, int C, float *dx, float *dgamma, float *dbeta, void *scratch, lwg_stream_t stream)
{
    LWG_REQUIRE(x && dy && stats && gamma && dx && dgamma && dbeta && scratch, "instance_norm_backward: NULL argument");
    if (C % RS_CH) LWG_FAIL(LWG_ERR_UNSUPPORTED, "instance_norm: C=%d must be a multiple of %d", C, RS_CH);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const float2 *s2 = reinterpret_cast<const float2 *>(stats);
    const bool vec4 = C % RS4_CH == 0;
    const int S = vec4 ? in_bwd_slabs4(N, C, HW) : in_slabs(N, C, HW);
    double *part = static_cast<double *>(scratch);
    float2 *sums = reinterpret_cast<float2 *>(part + (size_t)N * S * C * 2);
    if (vec4) {
        in_affine_bwd_sums4_kernel<<<dim3(C / RS4_CH, N, S), 256, 0, st>>>(x, y, dy, s2, HW, C, S, part);
        LWG_LAUNCH_CHECK("in_affine_bwd_sums4_kernel");
        in_sums_final_kernel<<<ceil_div((long)N * C, 64), 256, 0, st>>>(part, N, C, S, sums);
        LWG_LAUNCH_CHECK("in_sums_final_kernel");
    } else if (S == 1) {
        in_affine_bwd_reduce_kernel<<<dim3(C / RS_CH, N), 256, 0, st>>>(x, y, dy, s2, HW, C, sums);
        LWG_LAUNCH_CHECK("in_affine_bwd_reduce_kernel");
    } else {
        in_affine_bwd_partial_kernel<<<dim3(C / RS_CH, N, S), 256, 0, st>>>(x, y, dy, s2, HW, C, S, part);
        LWG_LAUNCH_CHECK("in_affine_bwd_partial_kernel");
        in_sums_final_kernel<<<ceil_div((long)N * C, 64), 256, 0, st>>>(part, N, C, S, sums);
        LWG_LAUNCH_CHECK("in_sums_final_kernel");
    }
    const long total = (long)N * HW * C;
    if (vec4) {
        in_affine_bwd_apply4_kernel<<<ceil_div(total / 4, 256), 256, 0, st>>>(x, y, dy, s2, sums, gamma, HW, C, total / 4, dx);
        LWG_LAUNCH_CHECK("in_affine_bwd_apply4_kernel");
    } else {
        in_affine_bwd_apply_kernel<<<ceil_div(total, 256), 256, 0, st>>>(x, y, dy, s2, sums, gamma, HW, C, total, dx);
        LWG_LAUNCH_CHECK("in_affine_bwd_apply_kernel");
    }
    in_affine_bwd_params_kernel<<<ceil_div(C, 256), 256, 0, st>>>(sums, N, C, dgamma, dbeta);
    LWG_LAUNCH_CHECK("in_affine_bwd_params_kernel");
    return LWG_OK;
}

/* Gradient of lwg_grid_sample wrt its input (NHWC here): dy (n,Ho,Wo,C), grid (n,Ho,Wo,2) -> dx (xn,H,W,C), xn in {1, n};
 * dx is ACCUMULATED into (zero it first).  Atomic fp32 adds: not bit-reproducible. */
int lwg_grid_sample_backward(const float *dy, const float *grid, int xn, int C, int H, int W, int n, int Ho, int Wo,
                             int align_corners, float *dx, lwg_stream_t stream)
{
    LWG_REQUIRE(dy && grid && dx, "grid_sample_backward: NULL argument");
    if (C % 4) LWG_FAIL(LWG_ERR_UNSUPPORTED, "grid_sample_backward: C=%d must be a multiple of 4", C);
    if (xn != 1 && xn != n) LWG_FAIL(LWG_ERR_INVALID_ARG, "grid_sample_backward: input batch must be 1 or %d", n);
    const long total = (long)n * Ho * Wo * (C / 4);
    grid_sample_bwd_kernel<<<ceil_div(total, 256), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(dy, grid, xn, C, H, W, Ho, Wo,
                                                                                                  align_corners, total, dx);
    LWG_LAUNCH_CHECK("grid_sample_bwd_kernel");
    return LWG_OK;
}

/* torch.optim.Adam step (no weight decay) on any flat fp32 device tensor; `step` counts from 1 */
int lwg_adam_update(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n, long step, float lr, float beta1,
                    float beta2, float eps, lwg_stream_t stream)
{
    LWG_REQUIRE(param && grad && exp_avg && exp_avg_sq && step >= 1, "adam_update: bad argument");
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    adam_kernel<<<ceil_div((long)n, 256), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(param, grad, exp_avg, exp_avg_sq, (long)n,
                                                                                          lr, beta1, beta2, eps, bc1, sqrtf(bc2), nullptr);
    LWG_LAUNCH_CHECK("adam_kernel");
    return LWG_OK;
}

int lwg_adam_update_device_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n, void *step_state,
                                float lr, float beta1, float beta2, float eps, lwg_stream_t stream)
{
    LWG_REQUIRE(param && grad && exp_avg && exp_avg_sq && step_state, "adam_update_device_step: bad argument");
    static_assert(sizeof(AdamStep) == 24, "three 8-byte words: include/lwg.h");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    AdamStep *sd = static_cast<AdamStep *>(step_state);
    step_inc_kernel<<<1, 1, 0, st>>>(sd, beta1, beta2);
    LWG_LAUNCH_CHECK("step_inc_kernel");
    adam_kernel<<<ceil_div((long)n, 256), 256, 0, st>>>(param, grad, exp_avg, exp_avg_sq, (long)n, lr, beta1, beta2, eps, 1.f, 1.f, sd);
    LWG_LAUNCH_CHECK("adam_kernel");
    return LWG_OK;
}

/* Generator-side adversarial term (impersonator_trainer.py:369-371): loss = mean((D(x) - target)^2) for x (bs,input_nc,is,is)
 * and its gradient wrt x (same shape, NCHW); the discriminator's parameters get no gradient from this call. */
int lwg_discriminator_input_grad(lwg_discriminator *d, const float *x_nchw, int bs, float target, float *loss_device,
                                 float *dx_nchw, lwg_stream_t stream)
{
    int rc = d_check(d, bs);
    if (rc != LWG_OK) return rc;
    LWG_REQUIRE(x_nchw && dx_nchw, "discriminator input_grad: NULL argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!d->wd0) {
        if ((rc = d_alloc(&d->wd0, (size_t)d->L[0].cout_pad * 16 * 64)) != LWG_OK) return rc;
        if ((rc = d_alloc(&d->dx0, (size_t)d->max_batch * d->is * d->is * 64)) != LWG_OK) return rc;
        d->wd_stale = true;
    }
    if (d->wd_stale && (rc = d_refresh_dgrad_weights(d, st)) != LWG_OK) return rc;
    if ((rc = d_forward(d, x_nchw, nullptr, bs, st)) != LWG_OK) return rc;
    const int nl = (int)d->L.size();
    {
        DLayer &L = d->L[nl - 1];
        lsgan_kernel<<<1, 256, 0, st>>>(L.raw, bs, L.Ho * L.Ho, L.cout_pad, L.dact, d->loss, 1, target, 0.f);
        LWG_LAUNCH_CHECK("lsgan_kernel");
        if (loss_device) LWG_HIP(hipMemcpyAsync(loss_device, d->loss, sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    for (int l = nl - 1; l >= 0; --l) {
        DLayer &L = d->L[l];
        const int HW = L.Ho * L.Ho;
        const long total = (long)bs * HW * L.cout_pad;
        if (L.norm) {
            in_bwd_reduce_kernel<<<dim3(L.cout_pad / RS_CH, bs), 256, 0, st>>>(L.raw, L.actv, L.dact, L.stats, HW, L.cout_pad, L.sums);
            LWG_LAUNCH_CHECK("in_bwd_reduce_kernel");
        }
        act_bwd_kernel<<<ceil_div(total, 256), 256, 0, st>>>(L.raw, L.act ? L.actv : nullptr, L.dact, L.norm ? L.stats : nullptr,
                                                             L.sums, HW, L.cout_pad, total, L.draw);
        LWG_LAUNCH_CHECK("act_bwd_kernel");
        if (l > 0) {
            if ((rc = d_conv_dgrad(d, l, bs, st)) != LWG_OK) return rc;
        } else {
            // first layer: 4-phase transposed conv onto a 64-channel image (the real input channels are the first few)
            ConvArgs a = base_args(L.draw, L.cout_pad, bs, L.Ho, L.cout_pad, d->wd0, d->dx0, 64);
            a.Hm = a.Wm = L.Hin / 2; a.Ho = a.Wo = L.Hin;
            a.stride = 1; a.pad = 0; a.os = 2;
            a.nphase = 4;
            for (int p = 0; p < 4; ++p) {
                const int py = p >> 1, px = p & 1;
                a.ph[p] = ConvPhase{2, 2, 4, 4 * L.cout_pad, (long)p * 64 * 4 * L.cout_pad, py, px, py ? 0 : -1, px ? 0 : -1};
            }
            a.mtiles = ceil_div((long)bs * a.Hm * a.Wm, kConvBM);
            a.precision = d->precision;
            if ((rc = launch_conv_igemm(a, 64, st)) != LWG_OK) return rc;
            if ((rc = lwg_unpack_nchw(d->dx0, bs, d->input_nc, d->is, d->is, 64, dx_nchw, stream)) != LWG_OK) return rc;
        }
    }
    return LWG_OK;
}

/* bilinear grid_sample (zeros padding), NHWC: x (xn,H,W,C), xn in {1, n}; grid (n,Ho,Wo,2) -> y (n,Ho,Wo,C) */
int lwg_grid_sample_nhwc(const float *x, int xn, int C, int H, int W, const float *grid, int n, int Ho, int Wo, int align_corners,
                         float *y, lwg_stream_t stream)
{
    LWG_REQUIRE(x && grid && y, "grid_sample_nhwc: NULL argument");
    if (C % 4) LWG_FAIL(LWG_ERR_UNSUPPORTED, "grid_sample_nhwc: C=%d must be a multiple of 4", C);
    if (xn != 1 && xn != n) LWG_FAIL(LWG_ERR_INVALID_ARG, "grid_sample_nhwc: input batch must be 1 or %d", n);
    const long total = (long)n * Ho * Wo * (C / 4);
    grid_sample_nhwc_kernel<<<ceil_div(total, 256), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(x, grid, xn, C, H, W, Ho, Wo,
                                                                                                   align_corners, total, y);
    LWG_LAUNCH_CHECK("grid_sample_nhwc_kernel");
    return LWG_OK;
}

}  // extern "C"
