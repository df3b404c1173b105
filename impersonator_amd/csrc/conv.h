// Launch interface of the generator's device kernels (conv.hip). Host-side only.
#pragma once
#include <vector>

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
    int iy0, ix0;      // general mode only: extra input offset of the phase's first tap (may be negative)
};

// Split-bf16 format (precision 1).  An fp32 value v is carried as two bf16 terms hi = bf16(v), lo = bf16(v - hi)
// (16 significand bits, |v - hi - lo| <= 2^-17 |v|).  A run of 32 consecutive values (32 channels of a pixel / 32
// reduction entries of a weight row) occupies the same 128 bytes it would as fp32: [hi x32 | lo x32].  Pixel strides,
// channel-slice offsets (multiples of 32) and weight offsets are therefore identical in both formats, the async
// global->LDS copy moves the same 16-byte chunks, and a product is evaluated as hi*hi + hi*lo + lo*hi on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation (bf16 x bf16 is exact in fp32; only lo*lo, ~2^-18, is dropped).
inline void split_bf16_groups(const float *src, size_t n, float *dst_opaque)   // host helper, n % 32 == 0
{
    __bf16 *d = reinterpret_cast<__bf16 *>(dst_opaque);
    for (size_t g = 0; g < n; g += 32)
        for (int e = 0; e < 32; ++e) {
            const float v = src[g + e];
            const __bf16 hi = (__bf16)v;
            d[2 * g + e] = hi;
            d[2 * g + 32 + e] = (__bf16)(v - (float)hi);
        }
}

// One convolution expressed as (up to 4) implicit GEMMs:  Y[m][co] = sum_k A[m][k] * Wt[co][k],
//   m -> (image, hm, wm) over an Hm x Wm grid per image, k -> (tap, ci),
//   A[m][k] = X[image][hm*stride - pad + kh*dil][wm*stride - pad + kw*dil][ci]   (0 outside the image).
// A stride-2 transposed conv is 4 such GEMMs (one per output parity) with 1/2/2/4 taps and pad 0.
struct ConvArgs {
    const float *x; int ldx;        // input NHWC, pixel stride in floats (>= Cin: reads a channel slice of a wider buffer)
    int N, H, W, Cin, cin_log2;
    const float *w;
    const float *w_split;           // w in the split-bf16 format below (same offsets, opaque 4-byte units), or null
    int precision;                  // 0: exact fp32 MFMA on x / w;  1: bf16x3 on SPLIT x / w_split
    const float *zeros;             // source of out-of-image taps on the DMA paths, may be null: >= 16 bytes of zeros
                                    // (fp32); Cin floats behind the input tensor (bf16x3: read at the stage's channel offset)
    float *y; int ldy;              // raw (pre-norm) output NHWC
    int Ho, Wo, Cout;
    int Hm, Wm, stride, pad, os;
    int dil;                        // dilation (>= 1): tap (kh,kw) reads hm*stride - pad + kh*dil
    float2 *partials;               // [nphase][mtiles][Cout] per-tile (mean, M2) of the raw output, or null
    int mtiles;                     // N*Hm*Wm / kConvBM
    int nphase;
    int fuse_phases;                // 1: one workgroup per tile walks all phases (balanced transposed conv)
    // General mode (the training path's layers: PatchGAN 4x4 convs and their data gradients).  Lifts the "Hm*Wm is a
    // multiple of 128" rule: M = N*Hm*Wm rows are tiled across images with a masked tail, phases may start at a
    // negative input offset (ph.iy0/ix0), `bias` (per output channel, or null) is added in the epilogue, and
    // `accumulate` adds into y instead of storing.  Register-staged fp32 kernel only; no fused statistics
    // (partials must be null), mtiles = ceil(M / 128).
    int natural_order;              // 1: keep the natural tile order (default 0: XCD bands, see conv_igemm_bf16x3)
    int tap_inner;                  // bf16x3 kernel: 1 = walk the reduction (32-channel slice, tap) with the taps innermost
    // Fused InstanceNorm-apply of the INPUT (conv3x3_halo_bf16x3 only; conv_raw_input_supported() says whether a launch takes
    // it): channels >= raw_from of x hold the producer's RAW fp32 conv output instead of a split-bf16 activation, and the
    // kernel normalises them itself after the halo lands in LDS -- relu(x * scale + shift) with the producer's
    // (scale, shift) = in_ss[image * in_ss_ld + (channel - raw_from)], then the hi/lo split -- the same operations, bit for
    // bit, that apply_kernel would have written to memory and this kernel read back.  raw_in = 0: off.
    int raw_in, raw_from;
    const float2 *in_ss; int in_ss_ld;
    unsigned long long *trace;      // measurement builds only (LWG_CONV_TRACE): per-wave cycle accounting, see conv_igemm_bf16x3
    int general;
    const float *bias;
    // General mode, few output tiles and a long reduction (the PatchGAN's 512-channel layers on 16 x 16 maps: 64-128 workgroups
    // of 256 stages): ksplit > 1 workgroups share a tile's reduction; split s writes its raw sums (no bias) to
    // kpart + s * kpart_stride at y's offsets, and the caller adds the slices in order (+ bias) -- deterministic.
    int ksplit;
    float *kpart;
    size_t kpart_stride;
    ConvPhase ph[4];
};

// measurement hook behind lwg_conv_trace / lwg_conv_trace_launch (include/lwg.h)
int conv_trace_set(void *device_buffer, size_t bytes);
int conv_trace_launch(int idx, long long *v10);

// bn = 64 or 128 output channels per workgroup tile (exact-fp32 DMA path also 32: twice the workgroups for launches that leave
// most of the chip idle -- one source, one frame).  *variant (optional) receives which kernel instantiation ran:
enum { kIgemmReg64 = 0, kIgemmReg128 = 1, kIgemmSmallCin = 2, kIgemmDma64 = 3, kIgemmDma128 = 4, kIgemmBf16x3_64 = 5,
       kIgemmBf16x3_128 = 6, kDirectStemBf16x3 = 7, kHaloBf16x3_128 = 8, kHaloBf16x3_64 = 9, kIgemmDma32 = 10, kIgemmVariants = 11 };
extern const char *const kIgemmVariantNames[kIgemmVariants];
int launch_conv_igemm(const ConvArgs &a, int bn, hipStream_t st, int *variant = nullptr);
// true when launch_conv_igemm(a, bn) would run a kernel that honours a.raw_in (the halo-resident 3x3 / transposed kernels)
bool conv_raw_input_supported(const ConvArgs &a, int bn);

// The 7x7 stem (Cin <= 6 in an NHWC8 fp32 tensor -> 64 channels) as an LDS-resident direct convolution on the bf16x3
// path (direct.hip).  `w` is the filter bank packed by stem_pack_weights; output and partials as launch_conv_igemm.
constexpr int kStemKRow = 48;                      // reduction entries per kernel row: 7 taps x 6 channels = 42, padded to three k-steps
constexpr int kStemWPitch = 7 * kStemKRow * 2 + 16; // bytes per output channel and plane: 7 rows x 96 B + 16 pad (43 x 16: odd)
constexpr int kStemWBytes = 2 * 64 * kStemWPitch;  // hi plane, lo plane
struct StemArgs {
    const float *x; int N, H, W;    // NHWC8 fp32
    const void *w;                  // device, kStemWBytes
    float *y;                       // raw output NHWC (N,H,W,64)
    float2 *partials;               // [N*H*W/128][64] or null
    int dbg;                        // measurement builds only (LWG_STEM_DBG)
};
bool stem_bf16x3_supported(int H, int W, int cin, int cin_pad, int cout, int k, int stride, int pad);
void stem_pack_weights(const float *w_oihw, int cin, std::vector<unsigned char> &out);
int launch_stem_bf16x3(const StemArgs &a, hipStream_t st);

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
    int split;                      // 1: dst is written, and res read, in the split-bf16 format (C, offsets % 32 == 0)
    int nwarp;                      // 0..2 Liquid-Warping-Block terms added after the activation
    const float *warp_src[2];       // source features NHWC (warp_n, H, W, C)
    int warp_n[2];                  // 1 (shared source) or N
    const float *warp_T[2];         // flow resized to (N,H,W,2)
    int align_corners;
};
int launch_apply(const ApplyArgs &a, hipStream_t st);

// in-place split-bf16 -> fp32 of n floats (n % 32 == 0): the test hook's view of a split activation buffer
int launch_unsplit(float *buf, size_t n, hipStream_t st);

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
// the same on the bf16x3 matrix path (heads.hip): `wfrag` = the weights packed into MFMA B fragments by
// launch_heads_pack(wh [49][64][4], wfrag) (heads_bf16x3_frag_bytes() bytes, device)
size_t heads_bf16x3_frag_bytes();
int launch_heads_pack(const float *wh, void *wfrag, hipStream_t st);
int launch_heads_bf16x3(const HeadsArgs &a, const void *wfrag, hipStream_t st);

}  // namespace lwg
