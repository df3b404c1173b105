// Launch interface of the generator's device kernels (conv.hip). Host-side only.
#pragma once
#include "common.h"

namespace lwg {

constexpr int kConvBM = 128;  // output pixels per workgroup tile
constexpr int kConvBK = 32;   // reduction slice per LDS stage

struct ConvPhase {
    int KH, KW;        // taps of this phase
    int ntaps;         // KH*KW
    int Kpad;          // padded reduction length (multiple of kConvBK)
    long w_off;        // float offset of this phase's [Cout][Kpad] matrix inside the layer's weights
    int oy0, ox0;      // output pixel = (hm*os + oy0, wm*os + ox0)
};

// One convolution expressed as (up to 4) implicit GEMMs:  Y[m][co] = sum_k A[m][k] * Wt[co][k],
//   m -> (image, hm, wm) over an Hm x Wm grid per image, k -> (tap, ci),
//   A[m][k] = X[image][hm*stride - pad + kh*dil][wm*stride - pad + kw*dil][ci]   (0 outside the image).
// A stride-2 transposed conv is 4 such GEMMs (one per output parity) with 1/2/2/4 taps and pad 0.
struct ConvArgs {
    const float *x; int ldx;        // input NHWC, pixel stride in floats (>= Cin: reads a channel slice of a wider buffer)
    int N, H, W, Cin, cin_log2;
    const float *w;
    const float *zeros;             // >= 16 bytes of zeros (source of out-of-image taps on the DMA path), may be null
    float *y; int ldy;              // raw (pre-norm) output NHWC
    int Ho, Wo, Cout;
    int Hm, Wm, stride, pad, os;
    int dil;                        // dilation (>= 1): tap (kh,kw) reads hm*stride - pad + kh*dil
    float2 *partials;               // [nphase][mtiles][Cout] per-tile (mean, M2) of the raw output, or null
    int mtiles;                     // N*Hm*Wm / kConvBM
    int nphase;
    int fuse_phases;                // 1: one workgroup per tile walks all phases (balanced transposed conv)
    ConvPhase ph[4];
};

// bn = 64 or 128 output channels per workgroup tile.  *variant (optional) receives which kernel instantiation ran:
enum { kIgemmReg64 = 0, kIgemmReg128 = 1, kIgemmSmallCin = 2, kIgemmDma64 = 3, kIgemmDma128 = 4, kIgemmVariants = 5 };
extern const char *const kIgemmVariantNames[kIgemmVariants];
int launch_conv_igemm(const ConvArgs &a, int bn, hipStream_t st, int *variant = nullptr);

// (mean, M2) partials -> per (image, channel) scale/shift of InstanceNorm2d(affine, eps) (biased variance)
int launch_in_finalize(const float2 *partials, int nphase, int mtiles, int N, int C, const float *gamma,
                       const float *beta, float eps, float2 *scale_shift, hipStream_t st);

struct ApplyArgs {
    const float *raw; int C;        // raw conv output, dense NHWC (N,H,W,C)
    int N, H, W;
    const float2 *scale_shift;      // [N][C]
    int relu;
    float *dst; int ld_dst;         // destination NHWC slice (channel offset already applied)
    const float *res; int ld_res;   // optional residual added after the norm (ResidualBlock)
    int nwarp;                      // 0..2 Liquid-Warping-Block terms added after the activation
    const float *warp_src[2];       // source features NHWC (warp_n, H, W, C)
    int warp_n[2];                  // 1 (shared source) or N
    const float *warp_T[2];         // flow resized to (N,H,W,2)
    int align_corners;
};
int launch_apply(const ApplyArgs &a, hipStream_t st);

// 7x7 regression heads on the 64-channel decoder output: color = tanh(conv), mask = sigmoid(conv),
// pred = mask*bg + (1-mask)*color (models/imitator.py:331). x is the RAW skipper output with its
// InstanceNorm+ReLU folded into the halo load (scale_shift), wh = [49][64][4].
struct HeadsArgs {
    const float *x; int N, H, W;
    const float2 *scale_shift;      // [N][64]
    const float *wh;
    float *color, *mask;            // NCHW, optional
    const float *bg; int bg_bs;     // NCHW, optional
    float *pred;                    // NCHW, optional
};
int launch_heads(const HeadsArgs &a, hipStream_t st);

}  // namespace lwg
