// personalize.hip -- the once-per-source glue of Imitator.personalize (models/imitator.py:82-155) as liblwg kernels, so that
// `personalize` launches no framework (ATen) compute kernel (gfx950).
//
//   morph_kernel            utils/util.py:73-89 `morph`: erode / dilate of a {0,1} mask with a ks x ks box = pad with 1 / 0, count
//                           the ones under the box, compare with ks^2 / 1.  The reference takes the counts with a ones-kernel
//                           conv2d; here a 16 x 64-pixel tile sits in LDS with its halo and the box sum is separable (row sums,
//                           then column sums of row sums).  Counts are integers <= ks^2 <= 961, exact in fp32 in any order, so the
//                           comparison -- the only thing that leaves the kernel -- is exact.  `complement` writes 1 - result
//                           (body_mask = 1 - bg_mask, ft_mask = 1 - erode: imitator.py:117,134).
//   mask_compose_kernel     torch.cat([img * mask, tail], dim=1) (imitator.py:127-128, 135: BGNet input, source-stream input), the
//                           mask optionally complemented on the fly.
//   source_p2verts_kernel   imitator.py:105-107 (hazard H9): p2verts is a view of f2verts[..., 0:2] and `p2verts[..., 1] *= -1`
//                           mutates f2verts through it: negate y of f2verts IN PLACE and hand out the contiguous (bs,nf,3,2) copy
//                           the per-frame flow kernel reads.
//   vis_mark / vis_select   SMPLRenderer.get_vis_f2pts (utils/nmr.py:506-546, --only_vis, hazard H10): faces that are not among
//                           `fim.unique()[1:]` become -2.  unique() sorts, so [1:] drops the SMALLEST value present -- the
//                           background's -1 whenever a background pixel exists, otherwise the lowest visible face id (reproduced,
//                           not fixed): one flag per value, an atomicMin for the dropped one, a select pass.
// Latency-bound bookkeeping (a 256 x 256 mask, 13776 faces): no tuning beyond coalesced rows.
#include <climits>

#include "common.h"

namespace lwg {
namespace {

constexpr int MT_H = 16, MT_W = 64, MAX_KS = 31;

__global__ __launch_bounds__(256) void morph_kernel(const float *__restrict__ mask, long batch_stride, int H, int W, int ks, int dilate,
                                                    int complement, float *__restrict__ out)
{
    constexpr int PW = MT_W + MAX_KS - 1, PH = MT_H + MAX_KS - 1;
    __shared__ float tile[PH][PW + 1];
    __shared__ float rows[PH][MT_W + 1];
    const int pad = ks / 2, b = blockIdx.z, y0 = blockIdx.y * MT_H, x0 = blockIdx.x * MT_W;
    const int hh = MT_H + ks - 1, ww = MT_W + ks - 1;
    const float border = dilate ? 0.f : 1.f;
    const float *m = mask + (size_t)b * batch_stride;
    for (int i = threadIdx.x; i < hh * ww; i += 256) {
        const int r = i / ww, c = i - r * ww, y = y0 + r - pad, x = x0 + c - pad;
        tile[r][c] = ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? m[(size_t)y * W + x] : border;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < hh * MT_W; i += 256) {
        const int r = i / MT_W, c = i - r * MT_W;
        float s = 0.f;
        for (int k = 0; k < ks; ++k) s += tile[r][c + k];
        rows[r][c] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < MT_H * MT_W; i += 256) {
        const int r = i / MT_W, c = i - r * MT_W, y = y0 + r, x = x0 + c;
        if (y >= H || x >= W) continue;
        float s = 0.f;
        for (int k = 0; k < ks; ++k) s += rows[r + k][c];
        const bool on = dilate ? s >= 1.f : s == (float)(ks * ks);
        out[((size_t)b * H + y) * W + x] = (on != (complement != 0)) ? 1.f : 0.f;
    }
}

__global__ __launch_bounds__(256) void mask_compose_kernel(const float *__restrict__ img, const float *__restrict__ mask, int invert,
                                                           const float *__restrict__ tail, int ct, int n, int HW,
                                                           float *__restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int C = 3 + ct;
    if (i >= (long)n * C * HW) return;
    const int p = (int)(i % HW), c = (int)((i / HW) % C), b = (int)(i / ((long)HW * C));
    float v;
    if (c < 3) {
        const float m = mask[(size_t)b * HW + p];
        v = img[((size_t)b * 3 + c) * HW + p] * (invert ? 1.f - m : m);
    } else {
        v = tail[((size_t)b * ct + (c - 3)) * HW + p];
    }
    out[i] = v;
}

__global__ __launch_bounds__(256) void source_p2verts_kernel(float *__restrict__ f2verts, long total_verts, float *__restrict__ p2verts)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // one face vertex (x, y, z)
    if (i >= total_verts) return;
    const float x = f2verts[i * 3], y = f2verts[i * 3 + 1] * -1.f;
    f2verts[i * 3 + 1] = y;
    p2verts[i * 2] = x;
    p2verts[i * 2 + 1] = y;
}

// flags: (bs, nf + 1) ints, entry 0 = the value -1 (background); lowest: (bs) ints preset to INT_MAX
__global__ __launch_bounds__(256) void vis_mark_kernel(const int *__restrict__ fim, int HW, int nf, int *__restrict__ flags,
                                                       int *__restrict__ lowest)
{
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    const int f = fim[(size_t)b * HW + i];
    if (f < -1 || f >= nf) return;   // cannot happen for a rasteriser output; never index out of bounds
    flags[(size_t)b * (nf + 1) + f + 1] = 1;
    atomicMin(lowest + b, f);
}

// max |a[i] - b[i]| over n floats into *out (zeroed on the stream ahead of the launch): non-negative floats order like their bit
// patterns, so the block maxima meet in one atomicMax on the bits; a NaN difference reports +inf.  Behind ImpersonatorGenerator's
// `precision="auto"` probe (one read-back of four bytes per weight set).
__global__ __launch_bounds__(256) void max_abs_diff_kernel(const float *__restrict__ a, const float *__restrict__ b, long n, unsigned *out)
{
    __shared__ float red[4];
    float m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float d = fabsf(a[i] - b[i]);
        if (d != d) d = __builtin_inff();
        m = fmaxf(m, d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, __builtin_bit_cast(unsigned, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}

__global__ __launch_bounds__(256) void vis_select_kernel(const float *__restrict__ f2pts, int nf, int per_face,
                                                         const int *__restrict__ flags, const int *__restrict__ lowest,
                                                         float *__restrict__ out)
{
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf * per_face) return;
    const int f = i / per_face;
    const bool vis = flags[(size_t)b * (nf + 1) + f + 1] != 0 && f != lowest[b];   // unique()[1:] drops the smallest value present
    const size_t o = (size_t)b * nf * per_face + i;
    out[o] = vis ? f2pts[o] : -2.f;
}

// ---- appearance transfer glue (models/swapper.py:198-253): the mask bookkeeping of Swapper.swap as kernels, so that a swap launches
// no framework kernel, copies no index list to the device and never reads the device back (the boolean-mask assignment of the
// reference's calculate_trans does) -- which also makes the whole swap capturable as one HIP graph.
//   swap_masks_kernel   part map (nparts, H, W) of the source -> part_mask = (sum of the selected parts != 0), left_mask = (sum of the
//                       kept parts != 0) as {0,1} floats (:206-207, the sums taken in channel order), and T11 = the identity grid where
//                       left_mask, else -2 (:243-245)
//   mask_faces_kernel   tsf_f2p = p2verts.clone(); tsf_f2p[0, left_faces] = -2 (:246-247) with the face set as a byte per face
//   swap_compose_kernel cat([tsf21 * part_mask + tsf11 * left_mask, cond]) (:213-216): products rounded separately, then added
//   clamp_kernel        T21.clamp_(-2, 2) (:249)
__global__ __launch_bounds__(256) void swap_masks_kernel(const float *__restrict__ part, int nparts, int HW, unsigned sel_bits,
                                                         unsigned left_bits, const float *__restrict__ grid,
                                                         float *__restrict__ part_mask, float *__restrict__ left_mask,
                                                         float *__restrict__ T11)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    float s = 0.f, l = 0.f;
    for (int c = 0; c < nparts; ++c) {
        const float v = part[(size_t)c * HW + p];
        if ((sel_bits >> c) & 1u) s += v;
        if ((left_bits >> c) & 1u) l += v;
    }
    const bool keep = l != 0.f;
    part_mask[p] = s != 0.f ? 1.f : 0.f;
    left_mask[p] = keep ? 1.f : 0.f;
    const float2 g = *reinterpret_cast<const float2 *>(grid + (size_t)p * 2);
    *reinterpret_cast<float2 *>(T11 + (size_t)p * 2) = keep ? g : make_float2(-2.f, -2.f);
}

__global__ __launch_bounds__(256) void mask_faces_kernel(const float *__restrict__ f2pts, const unsigned char *__restrict__ drop, int nf,
                                                         int per_face, float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf * per_face) return;
    out[i] = drop[i / per_face] ? -2.f : f2pts[i];
}

__global__ __launch_bounds__(256) void swap_compose_kernel(const float *__restrict__ tsf21, const float *__restrict__ tsf11,
                                                           const float *__restrict__ part_mask, const float *__restrict__ left_mask,
                                                           const float *__restrict__ cond, int nc, int HW, float *__restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)(3 + nc) * HW) return;
    const int c = (int)(i / HW), p = (int)(i - (long)c * HW);
    if (c < 3) {
        const float a = tsf21[i] * part_mask[p], b = tsf11[i] * left_mask[p];   // (built with -ffp-contract=off: two roundings, then the sum)
        out[i] = a + b;
    } else {
        out[i] = cond[(size_t)(c - 3) * HW + p];
    }
}

__global__ __launch_bounds__(256) void clamp_kernel(float *__restrict__ x, long n, float lo, float hi)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = fminf(fmaxf(x[i], lo), hi);
}

}  // namespace
}  // namespace lwg

using namespace lwg;

extern "C" {

int lwg_swap_masks(const float *part, int nparts, int H, int W, unsigned selected_bits, unsigned left_bits, const float *grid,
                   float *part_mask, float *left_mask, float *T11, lwg_stream_t stream)
{
    LWG_REQUIRE(part && grid && part_mask && left_mask && T11, "swap_masks: NULL argument");
    LWG_REQUIRE(nparts > 0 && nparts <= 32 && H > 0 && W > 0, "swap_masks: 1..32 parts, positive image size");
    swap_masks_kernel<<<ceil_div((long)H * W, 256), 256, 0, as_stream(stream)>>>(part, nparts, H * W, selected_bits, left_bits, grid,
                                                                                part_mask, left_mask, T11);
    LWG_LAUNCH_CHECK("swap_masks_kernel");
    return LWG_OK;
}

int lwg_mask_faces(const float *f2pts, const unsigned char *drop, int nf, int per_face, float *out, lwg_stream_t stream)
{
    LWG_REQUIRE(f2pts && drop && out && nf > 0 && per_face > 0, "mask_faces: bad arguments");
    mask_faces_kernel<<<ceil_div((long)nf * per_face, 256), 256, 0, as_stream(stream)>>>(f2pts, drop, nf, per_face, out);
    LWG_LAUNCH_CHECK("mask_faces_kernel");
    return LWG_OK;
}

int lwg_swap_compose(const float *tsf21, const float *tsf11, const float *part_mask, const float *left_mask, const float *cond, int nc,
                     int H, int W, float *out, lwg_stream_t stream)
{
    LWG_REQUIRE(tsf21 && tsf11 && part_mask && left_mask && out && (cond || nc == 0), "swap_compose: NULL argument");
    LWG_REQUIRE(nc >= 0 && H > 0 && W > 0, "swap_compose: bad sizes");
    swap_compose_kernel<<<ceil_div((long)(3 + nc) * H * W, 256), 256, 0, as_stream(stream)>>>(tsf21, tsf11, part_mask, left_mask, cond, nc,
                                                                                             H * W, out);
    LWG_LAUNCH_CHECK("swap_compose_kernel");
    return LWG_OK;
}

int lwg_max_abs_diff(const float *a, const float *b, size_t n, float *out, lwg_stream_t stream)
{
    LWG_REQUIRE(a && b && out, "max_abs_diff: NULL argument");
    LWG_HIP(hipMemsetAsync(out, 0, sizeof(float), as_stream(stream)));
    if (!n) return LWG_OK;
    const long blocks = ceil_div((long)n, 256 * 8);
    max_abs_diff_kernel<<<blocks < 2048 ? (int)blocks : 2048, 256, 0, as_stream(stream)>>>(a, b, (long)n, reinterpret_cast<unsigned *>(out));
    LWG_LAUNCH_CHECK("max_abs_diff_kernel");
    return LWG_OK;
}

int lwg_clamp(float *x, size_t n, float lo, float hi, lwg_stream_t stream)
{
    LWG_REQUIRE(x && lo <= hi, "clamp: bad arguments");
    if (!n) return LWG_OK;
    clamp_kernel<<<ceil_div((long)n, 256), 256, 0, as_stream(stream)>>>(x, (long)n, lo, hi);
    LWG_LAUNCH_CHECK("clamp_kernel");
    return LWG_OK;
}

int lwg_morph(const float *mask, int n, int H, int W, long batch_stride, int ks, int mode, int complement, float *out,
              lwg_stream_t stream)
{
    LWG_REQUIRE(mask && out, "morph: NULL argument");
    LWG_REQUIRE(n > 0 && H > 0 && W > 0 && batch_stride >= (long)H * W, "morph: bad sizes");
    LWG_REQUIRE(ks >= 1 && ks <= MAX_KS && (ks & 1), "morph: box size %d must be odd and <= %d", ks, MAX_KS);
    LWG_REQUIRE(mode == 0 || mode == 1, "morph: mode 0 = erode, 1 = dilate");
    morph_kernel<<<dim3(ceil_div(W, MT_W), ceil_div(H, MT_H), n), 256, 0, as_stream(stream)>>>(mask, batch_stride, H, W, ks, mode,
                                                                                                 complement, out);
    LWG_LAUNCH_CHECK("morph_kernel");
    return LWG_OK;
}

int lwg_mask_compose(const float *img, const float *mask, int invert, const float *tail, int tail_channels, int n, int H, int W,
                     float *out, lwg_stream_t stream)
{
    LWG_REQUIRE(img && mask && out && (tail || tail_channels == 0), "mask_compose: NULL argument");
    LWG_REQUIRE(n > 0 && H > 0 && W > 0 && tail_channels >= 0, "mask_compose: bad sizes");
    const long total = (long)n * (3 + tail_channels) * H * W;
    mask_compose_kernel<<<ceil_div(total, 256), 256, 0, as_stream(stream)>>>(img, mask, invert, tail, tail_channels, n, H * W, out);
    LWG_LAUNCH_CHECK("mask_compose_kernel");
    return LWG_OK;
}

int lwg_source_p2verts(float *f2verts, int bs, int nf, float *p2verts, lwg_stream_t stream)
{
    LWG_REQUIRE(f2verts && p2verts && bs > 0 && nf > 0, "source_p2verts: bad arguments");
    const long total = (long)bs * nf * 3;
    source_p2verts_kernel<<<ceil_div(total, 256), 256, 0, as_stream(stream)>>>(f2verts, total, p2verts);
    LWG_LAUNCH_CHECK("source_p2verts_kernel");
    return LWG_OK;
}

size_t lwg_vis_f2pts_workspace_bytes(int bs, int nf) { return bs > 0 && nf > 0 ? ((size_t)bs * (nf + 1) + bs) * sizeof(int) : 0; }

int lwg_vis_f2pts(const float *f2pts, int bs, int nf, int per_face, const int32_t *fim, int H, int W, float *out, void *workspace,
                  size_t workspace_bytes, lwg_stream_t stream)
{
    LWG_REQUIRE(f2pts && fim && out, "vis_f2pts: NULL argument");
    LWG_REQUIRE(bs > 0 && nf > 0 && per_face > 0 && H > 0 && W > 0, "vis_f2pts: bad sizes");
    if (!workspace || workspace_bytes < lwg_vis_f2pts_workspace_bytes(bs, nf))
        LWG_FAIL(LWG_ERR_WORKSPACE, "vis_f2pts: workspace needs %zu bytes", lwg_vis_f2pts_workspace_bytes(bs, nf));
    hipStream_t st = as_stream(stream);
    int *flags = static_cast<int *>(workspace), *lowest = flags + (size_t)bs * (nf + 1);
    LWG_HIP(hipMemsetAsync(flags, 0, (size_t)bs * (nf + 1) * sizeof(int), st));
    LWG_HIP(hipMemsetAsync(lowest, 0x7f, (size_t)bs * sizeof(int), st));   // 0x7f7f7f7f: above every face id
    vis_mark_kernel<<<dim3(ceil_div((long)H * W, 256), bs), 256, 0, st>>>(fim, H * W, nf, flags, lowest);
    LWG_LAUNCH_CHECK("vis_mark_kernel");
    vis_select_kernel<<<dim3(ceil_div((long)nf * per_face, 256), bs), 256, 0, st>>>(f2pts, nf, per_face, flags, lowest, out);
    LWG_LAUNCH_CHECK("vis_select_kernel");
    return LWG_OK;
}

}  // extern "C"
