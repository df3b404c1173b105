// inpaint.hip -- background inpaintor (InpaintSANet, DeepFill-v2 style) for gfx950: once per source image.
//
// Replaces what PyTorch executes for networks/inpaintor.py:110-202 in eval mode:
//   35 gated convolutions  y = BN( LeakyReLU_0.2(conv_f(x) + b_f) * sigmoid(conv_g(x) + b_g) )   (inpaintor.py:12-48)
//   (5x5, 4x4 stride 2, 3x3 with dilation 1/2/4/8/16; nearest x2 upsampling in front of four of them, :51-68),
//   one 4096-token self-attention (:71-107), the mask compositing and clamps of InpaintSANet.forward (:178-202).
// A gated layer is ONE implicit GEMM on the exact-fp32 matrix cores (conv.hip) whose N dimension holds the feature
// filters followed by the gate filters, then one elementwise pass (bias, activation, gate, folded BatchNorm, optional
// x2 replication, channel padding for the next layer).  The attention is a streaming (online-softmax) kernel.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "conv.h"
#include "split.h"

namespace lwg {
namespace {

struct GatedArgs {
    const float *raw; int C2;        // conv output (npix, C2): features [0,Cp) then gates [Cp,2Cp)
    int Cp, Cout;                    // padded / real channel count of each half
    const float *bias;               // (C2)
    const float *bn_scale, *bn_shift;  // (Cout) folded eval-mode BatchNorm
    int act;                         // 1: LeakyReLU(0.2) on the feature half
    int H, W;                        // raw resolution
    int up;                          // 1: replicate every pixel 2x2 (nearest upsampling feeding the next conv)
    float *dst; int Cdst;            // NHWC output, Cdst >= Cout channels (extra ones zeroed)
    int split;                       // 1: dst in the split-bf16 format of conv.h (Cdst % 32 == 0): the consumer is a bf16x3 conv
};

__device__ __forceinline__ float gated_value(const GatedArgs &a, const float *px, int c)
{
    float f = px[c] + a.bias[c];
    const float g = px[a.Cp + c] + a.bias[a.Cp + c];
    if (a.act) f = f > 0.f ? f : 0.2f * f;
    return (f * (1.f / (1.f + expf(-g)))) * a.bn_scale[c] + a.bn_shift[c];
}

__global__ __launch_bounds__(256) void gated_apply_kernel(const GatedArgs a)
{
    const int c4n = a.Cdst >> 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)a.H * a.W * c4n) return;
    const int pix = (int)(i / c4n), c = (int)(i - (long)pix * c4n) * 4;
    const float *px = a.raw + (size_t)pix * a.C2;
    float4 y;
    y.x = c + 0 < a.Cout ? gated_value(a, px, c + 0) : 0.f;
    y.y = c + 1 < a.Cout ? gated_value(a, px, c + 1) : 0.f;
    y.z = c + 2 < a.Cout ? gated_value(a, px, c + 2) : 0.f;
    y.w = c + 3 < a.Cout ? gated_value(a, px, c + 3) : 0.f;
    // split-bf16 format: channels c..c+3 of a 32-channel group are 8 bytes of hi at 2*(c&31) and 8 of lo 64 B further
    const int soff = (c >> 5) * 32 + ((c & 31) >> 1);   // float (4-byte) units
    bf16x4_t h, l;
    h[0] = (__bf16)y.x; h[1] = (__bf16)y.y; h[2] = (__bf16)y.z; h[3] = (__bf16)y.w;
    l[0] = (__bf16)(y.x - (float)h[0]); l[1] = (__bf16)(y.y - (float)h[1]);
    l[2] = (__bf16)(y.z - (float)h[2]); l[3] = (__bf16)(y.w - (float)h[3]);
    auto put = [&](size_t o) {
        if (a.split) {
            *reinterpret_cast<bf16x4_t *>(a.dst + o * a.Cdst + soff) = h;
            *reinterpret_cast<bf16x4_t *>(a.dst + o * a.Cdst + soff + 16) = l;
        } else {
            *reinterpret_cast<float4 *>(a.dst + o * a.Cdst + c) = y;
        }
    };
    if (!a.up) {
        put((size_t)pix);
    } else {
        const int yy = pix / a.W, xx = pix - yy * a.W;
        const int W2 = 2 * a.W;
#pragma unroll
        for (int d = 0; d < 4; ++d) put((size_t)(2 * yy + (d >> 1)) * W2 + 2 * xx + (d & 1));
    }
}

// network input: cat([img*(1-m) + fill*m, m]) as NHWC8; fill = 1 (coarse stage, inpaintor.py:180-181) or the
// clamped coarse result (refine stage, :187-188)
__global__ __launch_bounds__(256) void inpaint_input_kernel(const float *__restrict__ img, const float *__restrict__ mask,
                                                            const float *__restrict__ fill, int npix,
                                                            float *__restrict__ out8)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const float m = mask[p];
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = img[c * npix + p] * (1.f - m) + (fill ? fill[c * npix + p] : 1.f) * m;
    float4 *o = reinterpret_cast<float4 *>(out8 + (size_t)p * 8);
    o[0] = make_float4(v[0], v[1], v[2], m);
    o[1] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// last gated layer of a stage (3 channels, no activation): x = clamp(y,-1,1) (NCHW) and, optionally,
// comp = x*m + img*(1-m) (inpaintor.py:183,194-196)
__global__ __launch_bounds__(256) void inpaint_output_kernel(const GatedArgs a, const float *__restrict__ img,
                                                             const float *__restrict__ mask, float *__restrict__ x_out,
                                                             float *__restrict__ comp)
{
    const int npix = a.H * a.W;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const float *px = a.raw + (size_t)p * a.C2;
    const float m = mask[p];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float y = fminf(fmaxf(gated_value(a, px, c), -1.f), 1.f);
        x_out[c * npix + p] = y;
        if (comp) comp[c * npix + p] = y * m + img[c * npix + p] * (1.f - m);
    }
}

// Self attention over N tokens (inpaintor.py:85-103): out = gamma * softmax(Q K^T) V + x.
// qkv (N, ld) raw 1x1-conv output [q(16) | k(16) | v(C)], bias (ld).  64 queries per workgroup, 256 threads =
// 64 queries x 4 channel groups of C/4; keys/values stream through LDS in tiles of 64 with an online softmax.
constexpr int AT_Q = 64, AT_K = 64, AT_D = 16;
template <int C>
__global__ __launch_bounds__(256) void attention_kernel(const float *__restrict__ qkv, int ld, const float *__restrict__ bias,
                                                        const float *__restrict__ x, float gamma, int N,
                                                        float *__restrict__ out)
{
    constexpr int CG = C / 4;
    __shared__ __attribute__((aligned(16))) float Ks[AT_K][AT_D];
    __shared__ __attribute__((aligned(16))) float Vs[AT_K][C];
    const int tid = threadIdx.x, ql = tid & 63, cg = tid >> 6;
    const int qi = blockIdx.x * AT_Q + ql;
    float q[AT_D];
#pragma unroll
    for (int d = 0; d < AT_D; ++d) q[d] = qkv[(size_t)qi * ld + d] + bias[d];
    float acc[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) acc[c] = 0.f;
    float m = -INFINITY, l = 0.f;
    for (int k0 = 0; k0 < N; k0 += AT_K) {
        __syncthreads();
        {   // K tile: 64 x 16 = one float4 per thread
            const int r = tid >> 2, c4 = (tid & 3) * 4;
            float4 v = *reinterpret_cast<const float4 *>(qkv + (size_t)(k0 + r) * ld + AT_D + c4);
            const float4 b = *reinterpret_cast<const float4 *>(bias + AT_D + c4);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
            *reinterpret_cast<float4 *>(&Ks[r][c4]) = v;
        }
        for (int i = tid; i < AT_K * C / 4; i += 256) {
            const int r = i / (C / 4), c4 = (i - r * (C / 4)) * 4;
            float4 v = *reinterpret_cast<const float4 *>(qkv + (size_t)(k0 + r) * ld + 2 * AT_D + c4);
            const float4 b = *reinterpret_cast<const float4 *>(bias + 2 * AT_D + c4);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
            *reinterpret_cast<float4 *>(&Vs[r][c4]) = v;
        }
        __syncthreads();
        float s[AT_K];
        float tmax = -INFINITY;
#pragma unroll
        for (int j = 0; j < AT_K; ++j) {
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < AT_D; ++e) d += q[e] * Ks[j][e];
            s[j] = d;
            tmax = fmaxf(tmax, d);
        }
        const float mn = fmaxf(m, tmax);
        const float f = expf(m - mn);   // exp(-inf) = 0 on the first tile
        l *= f;
#pragma unroll
        for (int c = 0; c < CG; ++c) acc[c] *= f;
        m = mn;
#pragma unroll
        for (int j = 0; j < AT_K; ++j) {   // fully unrolled: s[] must stay in registers
            const float p = expf(s[j] - m);
            l += p;
            const float *vr = &Vs[j][cg * CG];
#pragma unroll
            for (int c = 0; c < CG; ++c) acc[c] += p * vr[c];
        }
    }
    const float inv = 1.f / l;
    const size_t o = (size_t)qi * C + cg * CG;
#pragma unroll
    for (int c = 0; c < CG; ++c) out[o + c] = gamma * (acc[c] * inv) + x[o + c];
}

// ------------------------------------------------------------------------------------------------
// The same attention on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32, bit-for-bit fmaf chains): flash-style, transposed.
//   workgroup = 8 waves x 32 queries = 256 queries, one of KS key chunks (N / KS keys, 32 per tile); grid (N / 256, KS).
//   S^T tile = K_tile Q^T  (32 keys x 32 queries, 8 MFMAs over d = 16): lane (q = lane & 31, h = lane >> 5) ends up holding, for
//       ITS query, the scores of keys (r & 3) + 8 (r >> 2) + 4 h, r = 0..15 -- a whole column: the running max / sum of the online
//       softmax are per-lane scalars, completed with one exchange between the two half-waves;
//   O^T += V^T P^T  (128 channels x 32 queries, 4 x 16 MFMAs over the 32 keys): the B operand of step r is exactly the lane's own
//       p[r] (keys paired (k, k + 4) across the half-waves, as the accumulator layout has them) -- probabilities never leave their
//       registers; the A operand V[key][channel] is read from an LDS tile the eight waves share (32 consecutive channels per
//       half-wave: conflict-free 4-byte reads), double-buffered, staged through registers one tile ahead.  K rows come straight
//       from L1/L2 (2 KiB per tile).
//   Each workgroup leaves un-normalised O, the running max m and sum l of its key chunk; attention_combine_kernel merges the KS
//   chunks (softmax is associative under (m, l, O) pairs) and applies out = gamma * (O / l + b_v) + x.
// 4096 tokens: 4.8 GFLOP on 256 workgroups of 8 tiles instead of a VALU loop at 5 TFLOP/s.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int AM_Q = 256, AM_T = 32, AM_C = 128;   // queries per workgroup, keys per tile, value channels

__global__ __launch_bounds__(512, 2) void attention_mfma_kernel(const float *__restrict__ qkv, int ld, const float *__restrict__ bias,
                                                             int N, int keys_per_chunk, float *__restrict__ o_part,
                                                             float2 *__restrict__ ml_part)
{
    __shared__ __attribute__((aligned(16))) float Vs[2][AM_T][AM_C];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qlane = lane & 31, half = lane >> 5;
    const int q = blockIdx.x * AM_Q + wave * 32 + qlane;
    const int chunk = blockIdx.y, key_lo = chunk * keys_per_chunk, ntiles = keys_per_chunk / AM_T;

    // B operand of the score MFMAs: Q[q][2 s + half] (+ bias), s = 0..7
    float qv[8];
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) qv[s2] = qkv[(size_t)q * ld + 2 * s2 + half] + bias[2 * s2 + half];
    float bk[8];
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) bk[s2] = bias[AT_D + 2 * s2 + half];

    f32x16 acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
    float m = -INFINITY, l = 0.f;

    // staging geometry of a V tile: 32 x 128 floats = 1024 float4, two per thread (rows tid / 32 and 16 + tid / 32); the K rows of a
    // tile: the lane's own key row, 16 floats (the even or odd components are picked when the tile is used).  Named registers, not
    // arrays: everything fetched for tile t + 1 must stay in flight underneath tile t's MFMAs.
    const int vrow = tid >> 5, vc4 = (tid & 31) * 4;
    const float *vsrc = qkv + ((size_t)key_lo + vrow) * ld + 2 * AT_D + vc4;
    const float *ksrc = qkv + ((size_t)key_lo + qlane) * ld + AT_D;
    const size_t tile_stride = (size_t)AM_T * ld;
    float4 vp0 = *reinterpret_cast<const float4 *>(vsrc), vp1 = *reinterpret_cast<const float4 *>(vsrc + (size_t)16 * ld);
    float4 k0 = *reinterpret_cast<const float4 *>(ksrc), k1 = *reinterpret_cast<const float4 *>(ksrc + 4);
    float4 k2 = *reinterpret_cast<const float4 *>(ksrc + 8), k3 = *reinterpret_cast<const float4 *>(ksrc + 12);
    *reinterpret_cast<float4 *>(&Vs[0][vrow][vc4]) = vp0;
    *reinterpret_cast<float4 *>(&Vs[0][vrow + 16][vc4]) = vp1;
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        // A operand of the score MFMAs: K[key0 + qlane][2 s + half] (+ bias)
        float kc[8];
        kc[0] = (half ? k0.y : k0.x) + bk[0]; kc[1] = (half ? k0.w : k0.z) + bk[1];
        kc[2] = (half ? k1.y : k1.x) + bk[2]; kc[3] = (half ? k1.w : k1.z) + bk[3];
        kc[4] = (half ? k2.y : k2.x) + bk[4]; kc[5] = (half ? k2.w : k2.z) + bk[5];
        kc[6] = (half ? k3.y : k3.x) + bk[6]; kc[7] = (half ? k3.w : k3.z) + bk[7];
        if (t + 1 < ntiles) {        // next tile on its way while this one computes
            vsrc += tile_stride;
            ksrc += tile_stride;
            vp0 = *reinterpret_cast<const float4 *>(vsrc);
            vp1 = *reinterpret_cast<const float4 *>(vsrc + (size_t)16 * ld);
            k0 = *reinterpret_cast<const float4 *>(ksrc);
            k1 = *reinterpret_cast<const float4 *>(ksrc + 4);
            k2 = *reinterpret_cast<const float4 *>(ksrc + 8);
            k3 = *reinterpret_cast<const float4 *>(ksrc + 12);
        }
        // ---- scores: S^T = K_tile Q^T
        f32x16 sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kc[s2], qv[s2], sc, 0, 0, 0);
        // ---- online softmax of this lane's query over the tile's 32 keys (16 here, 16 in the other half-wave)
        float tmax = sc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, sc[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float mn = fmaxf(m, tmax);
        const float f = expf(m - mn);      // exp(-inf) = 0 on the first tile
        float p[16], psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = expf(sc[r] - mn);
            psum += p[r];
        }
        psum += __shfl_xor(psum, 32);
        l = l * f + psum;
        m = mn;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][r] *= f;
        // ---- O^T += V^T P^T: step r pairs keys (r & 3) + 8 (r >> 2) [half 0] and + 4 [half 1]: the lane's own p[r]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float *vr = &Vs[buf][(r & 3) + 8 * (r >> 2) + 4 * half][qlane];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[cb * 32], p[r], acc[cb], 0, 0, 0);
        }
        if (t + 1 < ntiles) {
            *reinterpret_cast<float4 *>(&Vs[buf ^ 1][vrow][vc4]) = vp0;
            *reinterpret_cast<float4 *>(&Vs[buf ^ 1][vrow + 16][vc4]) = vp1;
        }
        __syncthreads();
    }
    // ---- un-normalised partial output of this key chunk: lane (q, half) holds channels cb*32 + 8 g + 4 half + 0..3 in acc[cb][4g..4g+3]
    float *op = o_part + ((size_t)chunk * N + q) * AM_C;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
            *reinterpret_cast<float4 *>(op + cb * 32 + 8 * g4 + 4 * half) =
                make_float4(acc[cb][4 * g4], acc[cb][4 * g4 + 1], acc[cb][4 * g4 + 2], acc[cb][4 * g4 + 3]);
    if (half == 0) ml_part[(size_t)chunk * N + q] = make_float2(m, l);
}

// merges the key chunks of a query: m = max m_s, O = sum O_s exp(m_s - m), l = sum l_s exp(m_s - m); out = gamma (O / l + b_v) + x
__global__ __launch_bounds__(256) void attention_combine_kernel(const float *__restrict__ o_part, const float2 *__restrict__ ml_part,
                                                                int KS, int N, const float *__restrict__ bias_v,
                                                                const float *__restrict__ x, float gamma, float *__restrict__ out,
                                                                int split_out)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;   // (query, 4 channels)
    if (e >= (long)N * (AM_C / 4)) return;
    const int q = (int)(e / (AM_C / 4)), c = (int)(e - (long)q * (AM_C / 4)) * 4;
    float m = -INFINITY;
    for (int s2 = 0; s2 < KS; ++s2) m = fmaxf(m, ml_part[(size_t)s2 * N + q].x);
    float l = 0.f;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s2 = 0; s2 < KS; ++s2) {
        const float2 ml = ml_part[(size_t)s2 * N + q];
        const float w = expf(ml.x - m);
        const float4 v = *reinterpret_cast<const float4 *>(o_part + ((size_t)s2 * N + q) * AM_C + c);
        l += ml.y * w;
        o.x += v.x * w; o.y += v.y * w; o.z += v.z * w; o.w += v.w * w;
    }
    const float inv = 1.f / l;
    const float4 b = *reinterpret_cast<const float4 *>(bias_v + c);
    const float4 xr = *reinterpret_cast<const float4 *>(x + (size_t)q * AM_C + c);
    float4 r;
    r.x = gamma * (o.x * inv + b.x) + xr.x;
    r.y = gamma * (o.y * inv + b.y) + xr.y;
    r.z = gamma * (o.z * inv + b.z) + xr.z;
    r.w = gamma * (o.w * inv + b.w) + xr.w;
    if (!split_out) {
        *reinterpret_cast<float4 *>(out + (size_t)q * AM_C + c) = r;
        return;
    }
    const int soff = (c >> 5) * 32 + ((c & 31) >> 1);
    bf16x4_t hi4, lo4;
    hi4[0] = (__bf16)r.x; hi4[1] = (__bf16)r.y; hi4[2] = (__bf16)r.z; hi4[3] = (__bf16)r.w;
    lo4[0] = (__bf16)(r.x - (float)hi4[0]); lo4[1] = (__bf16)(r.y - (float)hi4[1]);
    lo4[2] = (__bf16)(r.z - (float)hi4[2]); lo4[3] = (__bf16)(r.w - (float)hi4[3]);
    *reinterpret_cast<bf16x4_t *>(out + (size_t)q * AM_C + soff) = hi4;
    *reinterpret_cast<bf16x4_t *>(out + (size_t)q * AM_C + soff + 16) = lo4;
}

struct GLayer {
    int cin, cin_pad, cout, cp, k, stride, dil, pad, up, act;
    float *w = nullptr, *bias = nullptr, *bn_scale = nullptr, *bn_shift = nullptr;
    float *w_split = nullptr;        // w in the split-bf16 format (layers whose input has whole 32-channel groups), else null
    int kpad = 0;
    // host staging of BatchNorm parameters until all four arrived
    std::vector<float> bn[4];   // weight, bias, running_mean, running_var
    bool got_wf = false, got_wg = false, got_bf = false, got_bg = false, got_bn = false;
};

}  // namespace
}  // namespace lwg

using namespace lwg;

struct lwg_inpaint {
    int c_dim, is;
    std::vector<GLayer> net[3];   // coarse_net, refine_conv_net, refine_upsample_net
    // attention
    float *wqkv = nullptr, *bqkv = nullptr;   // [192][128], [192]
    float gamma = 0.f;
    bool got_q[2] = {false, false}, got_k[2] = {false, false}, got_v[2] = {false, false}, got_gamma = false;
    // scratch
    float *in8 = nullptr, *act[2] = {nullptr, nullptr}, *raw = nullptr, *coarse = nullptr, *zeros = nullptr;
    // attention on the matrix cores: key chunks per query block, their partial (O, m, l) results
    int attn_ks = 0;
    float *attn_o = nullptr, *attn_ml = nullptr;
    int precision = 1;               // 1: gated convs with >= 32 input channels on the bf16x3 kernels (split operands); 0: exact fp32
    size_t act_floats = 0;           // size of act[0] / act[1] (a zero run for the DMA kernels' padding taps sits behind each)
};

namespace lwg {
namespace {

constexpr int kAttnC = 128, kQkvN = 192;   // q(16) + k(16) + v(128) padded to a multiple of 64
const char *const kNetNames[3] = {"coarse_net", "refine_conv_net", "refine_upsample_net"};

// (cin, cout, k, stride, dilation, up2x, activation): networks/inpaintor.py:115-176 with cnum = 32
void build_spec(std::vector<GLayer> net[3], int c_dim)
{
    const int c = 32;
    const int coarse[][7] = {{c_dim, c, 5, 1, 1, 0, 1}, {c, 2 * c, 4, 2, 1, 0, 1}, {2 * c, 2 * c, 3, 1, 1, 0, 1},
                             {2 * c, 4 * c, 4, 2, 1, 0, 1}, {4 * c, 4 * c, 3, 1, 1, 0, 1}, {4 * c, 4 * c, 3, 1, 1, 0, 1},
                             {4 * c, 4 * c, 3, 1, 2, 0, 1}, {4 * c, 4 * c, 3, 1, 4, 0, 1}, {4 * c, 4 * c, 3, 1, 8, 0, 1},
                             {4 * c, 4 * c, 3, 1, 16, 0, 1}, {4 * c, 4 * c, 3, 1, 1, 0, 1}, {4 * c, 4 * c, 3, 1, 1, 0, 1},
                             {4 * c, 2 * c, 3, 1, 1, 1, 1}, {2 * c, 2 * c, 3, 1, 1, 0, 1}, {2 * c, c, 3, 1, 1, 1, 1},
                             {c, c / 2, 3, 1, 1, 0, 1}, {c / 2, 3, 3, 1, 1, 0, 0}};
    const int rconv[][7] = {{c_dim, c, 5, 1, 1, 0, 1}, {c, c, 4, 2, 1, 0, 1}, {c, 2 * c, 3, 1, 1, 0, 1},
                            {2 * c, 2 * c, 4, 2, 1, 0, 1}, {2 * c, 4 * c, 3, 1, 1, 0, 1}, {4 * c, 4 * c, 3, 1, 1, 0, 1},
                            {4 * c, 4 * c, 3, 1, 1, 0, 1}, {4 * c, 4 * c, 3, 1, 2, 0, 1}, {4 * c, 4 * c, 3, 1, 4, 0, 1},
                            {4 * c, 4 * c, 3, 1, 8, 0, 1}, {4 * c, 4 * c, 3, 1, 16, 0, 1}};
    const int rup[][7] = {{4 * c, 4 * c, 3, 1, 1, 0, 1}, {4 * c, 4 * c, 3, 1, 1, 0, 1}, {4 * c, 2 * c, 3, 1, 1, 1, 1},
                          {2 * c, 2 * c, 3, 1, 1, 0, 1}, {2 * c, c, 3, 1, 1, 1, 1}, {c, c / 2, 3, 1, 1, 0, 1},
                          {c / 2, 3, 3, 1, 1, 0, 0}};
    auto add = [](std::vector<GLayer> &v, const int s[7]) {
        GLayer L;
        L.cin = s[0]; L.cout = s[1]; L.k = s[2]; L.stride = s[3]; L.dil = s[4]; L.up = s[5]; L.act = s[6];
        L.cin_pad = L.cin < 8 ? 8 : L.cin;
        L.cp = (L.cout + 31) / 32 * 32;
        // get_pad() (inpaintor.py:7-9): 'same' padding; 1 for the 4x4 stride-2 layers
        L.pad = (L.k == 4 && L.stride == 2) ? 1 : L.dil * (L.k - 1) / 2;
        L.kpad = (int)align_up((size_t)L.k * L.k * L.cin_pad, kConvBK);
        v.push_back(L);
    };
    for (auto &s : coarse) add(net[0], s);
    for (auto &s : rconv) add(net[1], s);
    for (auto &s : rup) add(net[2], s);
}

int dalloc(float **p, size_t n, bool zero = true)
{
    LWG_HIP(hipMalloc(reinterpret_cast<void **>(p), n * sizeof(float)));
    if (zero) LWG_HIP(hipMemset(*p, 0, n * sizeof(float)));
    return LWG_OK;
}

// conv weight (cout, cin, k, k) -> rows [row0, row0+cout) of the [2cp][kpad] matrix, K index = tap*cin_pad + ci
int upload_half(GLayer &L, int row0, const float *w, const int64_t *shape, int ndim, const char *key)
{
    if (ndim != 4 || shape[0] != L.cout || shape[1] != L.cin || shape[2] != L.k || shape[3] != L.k)
        LWG_FAIL(LWG_ERR_INVALID_ARG, "%s: expected shape (%d,%d,%d,%d)", key, L.cout, L.cin, L.k, L.k);
    std::vector<float> h((size_t)L.cout * L.kpad, 0.f);
    const int kk = L.k * L.k;
    for (int co = 0; co < L.cout; ++co)
        for (int ci = 0; ci < L.cin; ++ci)
            for (int t = 0; t < kk; ++t) h[(size_t)co * L.kpad + (size_t)t * L.cin_pad + ci] = w[((size_t)co * L.cin + ci) * kk + t];
    LWG_HIP(hipMemcpy(L.w + (size_t)row0 * L.kpad, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    if (L.w_split) {   // the same rows as [hi x32 | lo x32] bf16 per 32 reduction entries (conv.h): same offsets
        std::vector<float> sp(h.size());
        split_bf16_groups(h.data(), h.size(), sp.data());
        LWG_HIP(hipMemcpy(L.w_split + (size_t)row0 * L.kpad, sp.data(), sp.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return LWG_OK;
}

int upload_bias(GLayer &L, int off, const float *b, const int64_t *shape, int ndim, const char *key)
{
    if (ndim != 1 || shape[0] != L.cout) LWG_FAIL(LWG_ERR_INVALID_ARG, "%s: expected shape (%d,)", key, L.cout);
    LWG_HIP(hipMemcpy(L.bias + off, b, (size_t)L.cout * sizeof(float), hipMemcpyHostToDevice));
    return LWG_OK;
}

int fold_bn(GLayer &L)
{
    for (auto &v : L.bn)
        if ((int)v.size() != L.cout) return LWG_OK;   // not complete yet
    std::vector<float> sc(L.cout), sh(L.cout);
    for (int c = 0; c < L.cout; ++c) {   // F.batch_norm(eval): (x - mean) / sqrt(var + 1e-5) * weight + bias
        const double inv = 1.0 / std::sqrt((double)L.bn[3][c] + 1e-5);
        sc[c] = (float)(L.bn[0][c] * inv);
        sh[c] = (float)(L.bn[1][c] - L.bn[2][c] * L.bn[0][c] * inv);
    }
    LWG_HIP(hipMemcpy(L.bn_scale, sc.data(), sc.size() * sizeof(float), hipMemcpyHostToDevice));
    LWG_HIP(hipMemcpy(L.bn_shift, sh.data(), sh.size() * sizeof(float), hipMemcpyHostToDevice));
    L.got_bn = true;
    return LWG_OK;
}

// does layer L take its input in the split-bf16 format (and run on the bf16x3 kernels)?
bool layer_split(const lwg_inpaint *g, const GLayer &L) { return g->precision == 1 && L.w_split != nullptr; }

int run_gated_conv(lwg_inpaint *g, const GLayer &L, const float *x, bool x_split, int H, hipStream_t st, int *Ho)
{
    ConvArgs a = {};
    a.x = x; a.ldx = L.cin_pad; a.N = 1; a.H = H; a.W = H; a.Cin = L.cin_pad;
    a.cin_log2 = 0;
    while ((1 << a.cin_log2) < L.cin_pad) ++a.cin_log2;
    a.w = L.w; a.zeros = g->zeros; a.y = g->raw; a.ldy = 2 * L.cp; a.Cout = 2 * L.cp;
    if (x_split) {
        // bf16x3: three bf16 MFMA products per multiply-add on split operands, fp32 accumulate (conv.h); the padding taps read the
        // zero run behind the activation buffer x lives in
        if (!L.w_split) LWG_FAIL(LWG_ERR_STATE, "inpaint: split input for a layer without split weights");
        a.precision = 1;
        a.w_split = L.w_split;
        a.tap_inner = 1;
        a.zeros = (x >= g->act[0] && x < g->act[0] + g->act_floats) ? g->act[0] + g->act_floats
                  : (x >= g->act[1] && x < g->act[1] + g->act_floats) ? g->act[1] + g->act_floats : nullptr;
        if (!a.zeros) LWG_FAIL(LWG_ERR_STATE, "inpaint: bf16x3 conv input is not one of the handle's activation buffers");
    }
    a.Hm = (H + 2 * L.pad - L.dil * (L.k - 1) - 1) / L.stride + 1;
    a.Wm = a.Hm; a.Ho = a.Hm; a.Wo = a.Hm;
    a.stride = L.stride; a.pad = L.pad; a.os = 1; a.dil = L.dil;
    a.partials = nullptr;
    a.mtiles = a.Hm * a.Wm / kConvBM;
    a.nphase = 1;
    a.ph[0].KH = a.ph[0].KW = L.k; a.ph[0].ntaps = L.k * L.k; a.ph[0].Kpad = L.kpad; a.ph[0].w_off = 0;
    a.ph[0].oy0 = a.ph[0].ox0 = 0;
    *Ho = a.Hm;
    const int bn = (a.Cout % 128 == 0 && (long)a.mtiles * (a.Cout / 128) >= 256) ? 128 : 64;
    return launch_conv_igemm(a, bn, st);
}

GatedArgs gated_args(const lwg_inpaint *g, const GLayer &L, int H, float *dst, int cdst)
{
    GatedArgs a = {};
    a.raw = g->raw; a.C2 = 2 * L.cp; a.Cp = L.cp; a.Cout = L.cout; a.bias = L.bias;
    a.bn_scale = L.bn_scale; a.bn_shift = L.bn_shift; a.act = L.act; a.H = H; a.W = H; a.up = 0;
    a.dst = dst; a.Cdst = cdst; a.split = 0;
    return a;
}

// runs net[n] starting from `x` (NHWC, net's first cin_pad channels) at resolution H; layers alternate between the two
// activation buffers; the last layer is left in g->raw for the caller when `keep_last_raw`
int run_net(lwg_inpaint *g, int n, const float *x, bool x_split, int *H, bool keep_last_raw, const float **out, hipStream_t st)
{
    std::vector<GLayer> &net = g->net[n];
    int cur = 0;
    for (size_t i = 0; i < net.size(); ++i) {
        const GLayer &L = net[i];
        int Ho = 0;
        int rc = run_gated_conv(g, L, x, x_split, *H, st, &Ho);
        if (rc != LWG_OK) return rc;
        *H = Ho;
        if (i + 1 == net.size() && keep_last_raw) break;
        // channel count the consumer reads: the next layer of this net, else the layer's own (padded to 4)
        const bool next_up = i + 1 < net.size() && net[i + 1].up;
        const int cdst = i + 1 < net.size() ? net[i + 1].cin_pad : (int)align_up((size_t)L.cout, 4);
        GatedArgs a = gated_args(g, L, Ho, g->act[cur], cdst);
        a.up = next_up ? 1 : 0;
        // the next layer of this net decides the format of what is written (the last layer's consumers read fp32)
        x_split = i + 1 < net.size() && layer_split(g, net[i + 1]);
        a.split = x_split ? 1 : 0;
        gated_apply_kernel<<<ceil_div((long)Ho * Ho * (cdst >> 2), 256), 256, 0, st>>>(a);
        LWG_LAUNCH_CHECK("gated_apply_kernel");
        if (next_up) *H = 2 * Ho;
        x = g->act[cur];
        cur ^= 1;
    }
    if (out) *out = x;
    return LWG_OK;
}

int missing(const lwg_inpaint *g)
{
    int m = 0;
    for (auto &net : g->net)
        for (auto &L : net) m += !L.got_wf + !L.got_wg + !L.got_bf + !L.got_bg + !L.got_bn;
    for (int i = 0; i < 2; ++i) m += !g->got_q[i] + !g->got_k[i] + !g->got_v[i];
    return m + !g->got_gamma;
}

}  // namespace
}  // namespace lwg

extern "C" {

int lwg_inpaint_create(lwg_inpaint **out, int c_dim, int image_size)
{
    LWG_REQUIRE(out, "inpaint_create: NULL out");
    *out = nullptr;
    if (c_dim != 4) LWG_FAIL(LWG_ERR_UNSUPPORTED, "inpaint_create: c_dim=%d (the Imitator path uses 4: rgb + mask)", c_dim);
    const int hq = image_size / 4;
    if (image_size <= 0 || (image_size & 3) || (hq * hq) % kConvBM != 0)
        LWG_FAIL(LWG_ERR_UNSUPPORTED, "inpaint_create: image_size=%d; (image_size/4)^2 must be a multiple of %d", image_size,
                 kConvBM);
    int ndev = 0;
    LWG_HIP(hipGetDeviceCount(&ndev));
    lwg_inpaint *g = new lwg_inpaint();
    g->c_dim = c_dim;
    g->is = image_size;
    build_spec(g->net, c_dim);
    int rc = LWG_OK;
    for (auto &net : g->net)
        for (auto &L : net) {
            if (rc == LWG_OK) rc = dalloc(&L.w, (size_t)2 * L.cp * L.kpad);
            // layers whose input has whole 32-channel groups can run on the bf16x3 kernels (split operands)
            if (rc == LWG_OK && L.cin_pad % 32 == 0 && L.kpad == L.k * L.k * L.cin_pad && L.k * L.k <= 32)
                rc = dalloc(&L.w_split, (size_t)2 * L.cp * L.kpad);
            if (rc == LWG_OK) rc = dalloc(&L.bias, (size_t)2 * L.cp);
            if (rc == LWG_OK) rc = dalloc(&L.bn_scale, L.cout);
            if (rc == LWG_OK) rc = dalloc(&L.bn_shift, L.cout);
        }
    const size_t P = (size_t)image_size * image_size;
    if (rc == LWG_OK) rc = dalloc(&g->wqkv, (size_t)kQkvN * kAttnC);
    if (rc == LWG_OK) rc = dalloc(&g->bqkv, kQkvN);
    if (rc == LWG_OK) rc = dalloc(&g->in8, P * 8);
    // largest activation: 64 channels replicated to full resolution in front of the last up-sampling conv
    // (+ 1024 zeros behind each: where the bf16x3 kernels' out-of-image taps point, never written)
    g->act_floats = P * 64;
    if (rc == LWG_OK) rc = dalloc(&g->act[0], P * 64 + 1024);
    if (rc == LWG_OK) rc = dalloc(&g->act[1], P * 64 + 1024);
    if (rc == LWG_OK) rc = dalloc(&g->raw, P * 64);
    if (rc == LWG_OK) rc = dalloc(&g->coarse, P * 3);
    if (rc == LWG_OK) rc = dalloc(&g->zeros, 64);
    {
        // attention_mfma_kernel: 256 queries per workgroup, KS key chunks so that every CU gets a workgroup; a chunk is whole
        // 32-key tiles.  Token counts it does not divide stay on attention_kernel (attn_ks = 0).
        const int N = hq * hq;
        int ks = 0;
        if (N % AM_Q == 0) {
            ks = 256 / (N / AM_Q);
            ks = ks < 1 ? 1 : (ks > 16 ? 16 : ks);
            while (ks > 1 && N % (ks * AM_T) != 0) --ks;
            if (N % (ks * AM_T) != 0) ks = 0;
        }
        g->attn_ks = ks;
        if (ks && rc == LWG_OK) rc = dalloc(&g->attn_o, (size_t)ks * N * AM_C, false);
        if (ks && rc == LWG_OK) rc = dalloc(&g->attn_ml, (size_t)ks * N * 2, false);
    }
    if (rc != LWG_OK) {
        lwg_inpaint_destroy(g);
        return rc;
    }
    *out = g;
    return LWG_OK;
}

void lwg_inpaint_destroy(lwg_inpaint *g)
{
    if (!g) return;
    auto fr = [](void *p) { if (p) (void)hipFree(p); };
    for (auto &net : g->net)
        for (auto &L : net) { fr(L.w); fr(L.w_split); fr(L.bias); fr(L.bn_scale); fr(L.bn_shift); }
    fr(g->wqkv); fr(g->bqkv); fr(g->in8); fr(g->act[0]); fr(g->act[1]); fr(g->raw); fr(g->coarse); fr(g->zeros);
    fr(g->attn_o); fr(g->attn_ml);
    delete g;
}

int lwg_inpaint_load_weight(lwg_inpaint *g, const char *key, const float *data_host, const int64_t *shape, int ndim)
{
    LWG_REQUIRE(g && key && data_host, "inpaint load_weight: NULL argument");
    std::string k(key);
    if (k.rfind("module.", 0) == 0) k = k.substr(7);
    if (k.size() > 19 && k.compare(k.size() - 19, 19, "num_batches_tracked") == 0) return LWG_OK;
    if (k.rfind("refine_attn.", 0) == 0) {
        const std::string t = k.substr(12);
        if (t == "gamma") {
            g->gamma = data_host[0];
            g->got_gamma = true;
            return LWG_OK;
        }
        // query/key (16,128,1,1) -> rows 0..15 / 16..31, value (128,128,1,1) -> rows 32..159 of [192][128]
        const bool isq = t.rfind("query_conv.", 0) == 0, isk = t.rfind("key_conv.", 0) == 0, isv = t.rfind("value_conv.", 0) == 0;
        if (!(isq || isk || isv)) LWG_FAIL(LWG_ERR_INVALID_ARG, "inpaint load_weight: unknown key '%s'", key);
        const int rows = isv ? kAttnC : 16, row0 = isq ? 0 : (isk ? 16 : 32);
        const bool is_w = t.size() > 7 && t.compare(t.size() - 7, 7, ".weight") == 0;
        bool *flag = isq ? g->got_q : (isk ? g->got_k : g->got_v);
        if (is_w) {
            if (ndim != 4 || shape[0] != rows || shape[1] != kAttnC || shape[2] != 1 || shape[3] != 1)
                LWG_FAIL(LWG_ERR_INVALID_ARG, "%s: expected shape (%d,%d,1,1)", key, rows, kAttnC);
            LWG_HIP(hipMemcpy(g->wqkv + (size_t)row0 * kAttnC, data_host, (size_t)rows * kAttnC * sizeof(float), hipMemcpyHostToDevice));
            flag[0] = true;
        } else {
            if (ndim != 1 || shape[0] != rows) LWG_FAIL(LWG_ERR_INVALID_ARG, "%s: expected shape (%d,)", key, rows);
            LWG_HIP(hipMemcpy(g->bqkv + row0, data_host, (size_t)rows * sizeof(float), hipMemcpyHostToDevice));
            flag[1] = true;
        }
        return LWG_OK;
    }
    int n = -1;
    for (int i = 0; i < 3; ++i)
        if (k.rfind(std::string(kNetNames[i]) + ".", 0) == 0) n = i;
    if (n < 0) LWG_FAIL(LWG_ERR_INVALID_ARG, "inpaint load_weight: unknown key '%s'", key);
    std::string rest = k.substr(strlen(kNetNames[n]) + 1);
    const size_t dot = rest.find('.');
    const int idx = atoi(rest.substr(0, dot).c_str());
    if (dot == std::string::npos || idx < 0 || idx >= (int)g->net[n].size())
        LWG_FAIL(LWG_ERR_INVALID_ARG, "inpaint load_weight: unknown key '%s'", key);
    GLayer &L = g->net[n][idx];
    rest = rest.substr(dot + 1);
    if (L.up) {   // GatedDeConv2dWithActivation wraps the gated conv in `.conv2d` (inpaintor.py:60-62)
        if (rest.rfind("conv2d.", 0) != 0) LWG_FAIL(LWG_ERR_INVALID_ARG, "inpaint load_weight: unknown key '%s'", key);
        rest = rest.substr(7);
    }
    int rc = LWG_OK;
    if (rest == "conv2d.weight") { rc = upload_half(L, 0, data_host, shape, ndim, key); L.got_wf = rc == LWG_OK; }
    else if (rest == "mask_conv2d.weight") { rc = upload_half(L, L.cp, data_host, shape, ndim, key); L.got_wg = rc == LWG_OK; }
    else if (rest == "conv2d.bias") { rc = upload_bias(L, 0, data_host, shape, ndim, key); L.got_bf = rc == LWG_OK; }
    else if (rest == "mask_conv2d.bias") { rc = upload_bias(L, L.cp, data_host, shape, ndim, key); L.got_bg = rc == LWG_OK; }
    else {
        static const char *const names[4] = {"batch_norm2d.weight", "batch_norm2d.bias", "batch_norm2d.running_mean",
                                            "batch_norm2d.running_var"};
        int which = -1;
        for (int i = 0; i < 4; ++i)
            if (rest == names[i]) which = i;
        if (which < 0) LWG_FAIL(LWG_ERR_INVALID_ARG, "inpaint load_weight: unknown key '%s'", key);
        if (ndim != 1 || shape[0] != L.cout) LWG_FAIL(LWG_ERR_INVALID_ARG, "%s: expected shape (%d,)", key, L.cout);
        L.bn[which].assign(data_host, data_host + L.cout);
        rc = fold_bn(L);
    }
    return rc;
}

int lwg_inpaint_missing_weights(const lwg_inpaint *g) { return g ? missing(g) : -1; }

int lwg_inpaint_set_precision(lwg_inpaint *g, int precision)
{
    LWG_REQUIRE(g, "inpaint_set_precision: NULL handle");
    if (precision != 0 && precision != 1) LWG_FAIL(LWG_ERR_INVALID_ARG, "inpaint_set_precision: 0 (fp32) or 1 (bf16x3)");
    g->precision = precision;
    return LWG_OK;
}

int lwg_inpaint_forward(lwg_inpaint *g, const float *imgs, const float *masks, float *coarse_x, float *x, float *comp,
                        lwg_stream_t stream)
{
    LWG_REQUIRE(g && imgs && masks, "inpaint_forward: NULL argument");
    LWG_REQUIRE(x, "inpaint_forward: the refined output x is required");
    const int m = missing(g);
    if (m) LWG_FAIL(LWG_ERR_STATE, "inpaint: %d weight tensors have not been loaded", m);
    hipStream_t st = as_stream(stream);
    const int S = g->is, P = S * S;
    float *coarse = coarse_x ? coarse_x : g->coarse;

    // ---- coarse stage (inpaintor.py:180-184)
    inpaint_input_kernel<<<ceil_div(P, 256), 256, 0, st>>>(imgs, masks, nullptr, P, g->in8);
    LWG_LAUNCH_CHECK("inpaint_input_kernel");
    int H = S;
    int rc = run_net(g, 0, g->in8, false, &H, true, nullptr, st);
    if (rc != LWG_OK) return rc;
    {
        const GatedArgs a = gated_args(g, g->net[0].back(), H, nullptr, 4);
        inpaint_output_kernel<<<ceil_div(P, 256), 256, 0, st>>>(a, imgs, masks, coarse, nullptr);
        LWG_LAUNCH_CHECK("inpaint_output_kernel");
    }
    // ---- refine stage (inpaintor.py:186-194)
    inpaint_input_kernel<<<ceil_div(P, 256), 256, 0, st>>>(imgs, masks, coarse, P, g->in8);
    LWG_LAUNCH_CHECK("inpaint_input_kernel");
    H = S;
    const float *feat = nullptr;
    if ((rc = run_net(g, 1, g->in8, false, &H, false, &feat, st)) != LWG_OK) return rc;
    bool feat_split = false;   // format of what the attention leaves for refine_upsample_net
    {   // self attention on (H*H) tokens of 128 channels: q/k/v as one 1x1 implicit GEMM, then the streaming softmax
        ConvArgs a = {};
        a.x = feat; a.ldx = kAttnC; a.N = 1; a.H = H; a.W = H; a.Cin = kAttnC; a.cin_log2 = 7;
        a.w = g->wqkv; a.zeros = g->zeros; a.y = g->raw; a.ldy = kQkvN; a.Cout = kQkvN;
        a.Hm = H; a.Wm = H; a.Ho = H; a.Wo = H; a.stride = 1; a.pad = 0; a.os = 1; a.dil = 1;
        a.mtiles = H * H / kConvBM; a.nphase = 1;
        a.ph[0].KH = a.ph[0].KW = 1; a.ph[0].ntaps = 1; a.ph[0].Kpad = kAttnC; a.ph[0].w_off = 0;
        if ((rc = launch_conv_igemm(a, 64, st)) != LWG_OK) return rc;
        float *dst = const_cast<float *>(feat) == g->act[0] ? g->act[1] : g->act[0];
        static const char *attn_env = getenv("LWG_ATTN");   // "valu": the streaming vector-ALU kernel (A/B switch)
        const int N = H * H;
        if (g->attn_ks && !(attn_env && attn_env[0] == 'v')) {
            attention_mfma_kernel<<<dim3(N / AM_Q, g->attn_ks), 512, 0, st>>>(g->raw, kQkvN, g->bqkv, N, N / g->attn_ks, g->attn_o,
                                                                           reinterpret_cast<float2 *>(g->attn_ml));
            LWG_LAUNCH_CHECK("attention_mfma_kernel");
            feat_split = layer_split(g, g->net[2][0]);   // written in the format refine_upsample_net's first layer reads
            attention_combine_kernel<<<ceil_div((long)N * (AM_C / 4), 256), 256, 0, st>>>(
                g->attn_o, reinterpret_cast<const float2 *>(g->attn_ml), g->attn_ks, N, g->bqkv + 2 * AT_D, feat, g->gamma, dst,
                feat_split ? 1 : 0);
            LWG_LAUNCH_CHECK("attention_combine_kernel");
        } else {
            attention_kernel<kAttnC><<<N / AT_Q, 256, 0, st>>>(g->raw, kQkvN, g->bqkv, feat, g->gamma, N, dst);
            LWG_LAUNCH_CHECK("attention_kernel");
        }
        feat = dst;
    }
    // run_net alternates act[0]/act[1] starting with act[0]: keep its first output away from `feat`
    if (feat == g->act[0]) {
        LWG_HIP(hipMemcpyAsync(g->act[1], feat, (size_t)H * H * kAttnC * sizeof(float), hipMemcpyDeviceToDevice, st));
        feat = g->act[1];
    }
    if ((rc = run_net(g, 2, feat, feat_split, &H, true, nullptr, st)) != LWG_OK) return rc;
    {
        const GatedArgs a = gated_args(g, g->net[2].back(), H, nullptr, 4);
        inpaint_output_kernel<<<ceil_div(P, 256), 256, 0, st>>>(a, imgs, masks, x, comp);
        LWG_LAUNCH_CHECK("inpaint_output_kernel");
    }
    return LWG_OK;
}

}  // extern "C"
