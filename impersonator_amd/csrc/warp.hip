// warp.hip -- boundary-layout kernels: NCHW grid_sample, flow resize, NCHW<->NHWC packing (gfx950).
//
// These serve the reference's python-visible operators on NCHW tensors
//   F.grid_sample            models/imitator.py:259, networks/generator.py:312-315 (stn)
//   F.interpolate(T, ...)    networks/generator.py:303-310 (resize_trans)
// The generator's internal Liquid Warping Block does NOT go through them: it samples NHWC features
// inside the normalise/activate pass (conv.hip, apply kernel), channel-contiguous and coalesced.
#include "common.h"
#include "sample.h"

namespace lwg {
namespace {

// one lane per output location; channel planes are walked with the same four taps
__global__ __launch_bounds__(256) void grid_sample_nchw_kernel(const float *__restrict__ x, int xn, int C, int H,
                                                               int W, const float *__restrict__ grid, int n, int Ho,
                                                               int Wo, int align_corners, float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int opix = Ho * Wo;
    if (i >= n * opix) return;
    const int b = i / opix, pn = i - b * opix;
    const float2 g = *reinterpret_cast<const float2 *>(grid + (size_t)i * 2);
    const GridTaps t = grid_taps(g.x, g.y, W, H, align_corners);
    const float *xb = x + (size_t)(xn > 1 ? b : 0) * C * H * W;
    float *ob = out + (size_t)b * C * opix + pn;
    for (int c = 0; c < C; ++c) {
        const float *pl = xb + (size_t)c * H * W;
        float acc = 0.f;
        if (t.vnw) acc += pl[t.y0 * W + t.x0] * t.wnw;
        if (t.vne) acc += pl[t.y0 * W + t.x0 + 1] * t.wne;
        if (t.vsw) acc += pl[(t.y0 + 1) * W + t.x0] * t.wsw;
        if (t.vse) acc += pl[(t.y0 + 1) * W + t.x0 + 1] * t.wse;
        ob[(size_t)c * opix] = acc;
    }
}

__global__ __launch_bounds__(256) void resize_flow_kernel(const float *__restrict__ T, int bs, int H, int W, int h,
                                                          int w, float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bs * h * w) return;
    const int b = i / (h * w), pn = i - b * h * w;
    const int y = pn / w, x = pn - y * w;
    const float2 r = resize_flow_at(T + (size_t)b * H * W * 2, H, W, h, w, y, x);
    *reinterpret_cast<float2 *>(out + (size_t)i * 2) = r;
}

// NCHW -> NHWC(cpad): a 32(pixels) x 32(channels) LDS transpose tile keeps both sides coalesced
__global__ __launch_bounds__(256) void pack_nhwc_kernel(const float *__restrict__ x, int C, int HW, int cpad,
                                                        float *__restrict__ out)
{
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        tile[r][tx] = (c < C && p < HW) ? x[((size_t)b * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        if (p < HW && c < cpad) out[((size_t)b * HW + p) * cpad + c] = tile[tx][r];
    }
}

__global__ __launch_bounds__(256) void unpack_nchw_kernel(const float *__restrict__ x, int C, int HW, int cpad,
                                                          float *__restrict__ out)
{
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        tile[r][tx] = (p < HW && c < C) ? x[((size_t)b * HW + p) * cpad + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        if (c < C && p < HW) out[((size_t)b * C + c) * HW + p] = tile[tx][r];
    }
}

}  // namespace
}  // namespace lwg

using namespace lwg;

extern "C" {

int lwg_grid_sample(const float *x, int xn, int C, int H, int W, const float *grid, int n, int Ho, int Wo,
                    int align_corners, float *out, lwg_stream_t stream)
{
    LWG_REQUIRE(x && grid && out, "grid_sample: NULL argument");
    LWG_REQUIRE(C > 0 && H > 0 && W > 0 && n > 0 && Ho > 0 && Wo > 0, "grid_sample: sizes must be positive");
    LWG_REQUIRE(xn == 1 || xn == n, "grid_sample: input batch must be 1 or %d (got %d)", n, xn);
    grid_sample_nchw_kernel<<<ceil_div((long)n * Ho * Wo, 256), 256, 0, as_stream(stream)>>>(x, xn, C, H, W, grid, n,
                                                                                             Ho, Wo, align_corners,
                                                                                             out);
    LWG_LAUNCH_CHECK("grid_sample_nchw_kernel");
    return LWG_OK;
}

int lwg_resize_flow(const float *T, int bs, int H, int W, int h, int w, float *out, lwg_stream_t stream)
{
    LWG_REQUIRE(T && out, "resize_flow: NULL argument");
    LWG_REQUIRE(bs > 0 && H > 0 && W > 0 && h > 0 && w > 0, "resize_flow: sizes must be positive");
    resize_flow_kernel<<<ceil_div((long)bs * h * w, 256), 256, 0, as_stream(stream)>>>(T, bs, H, W, h, w, out);
    LWG_LAUNCH_CHECK("resize_flow_kernel");
    return LWG_OK;
}

int lwg_pack_nhwc(const float *x_nchw, int n, int C, int H, int W, int cpad, float *out_nhwc, lwg_stream_t stream)
{
    LWG_REQUIRE(x_nchw && out_nhwc, "pack_nhwc: NULL argument");
    LWG_REQUIRE(n > 0 && C > 0 && H > 0 && W > 0 && cpad >= C, "pack_nhwc: bad sizes (C=%d cpad=%d)", C, cpad);
    const dim3 grid(ceil_div((long)H * W, 32), ceil_div(cpad, 32), n);
    pack_nhwc_kernel<<<grid, 256, 0, as_stream(stream)>>>(x_nchw, C, H * W, cpad, out_nhwc);
    LWG_LAUNCH_CHECK("pack_nhwc_kernel");
    return LWG_OK;
}

int lwg_unpack_nchw(const float *x_nhwc, int n, int C, int H, int W, int cpad, float *out_nchw, lwg_stream_t stream)
{
    LWG_REQUIRE(x_nhwc && out_nchw, "unpack_nchw: NULL argument");
    LWG_REQUIRE(n > 0 && C > 0 && H > 0 && W > 0 && cpad >= C, "unpack_nchw: bad sizes (C=%d cpad=%d)", C, cpad);
    const dim3 grid(ceil_div((long)H * W, 32), ceil_div(C, 32), n);
    unpack_nchw_kernel<<<grid, 256, 0, as_stream(stream)>>>(x_nhwc, C, H * W, cpad, out_nchw);
    LWG_LAUNCH_CHECK("unpack_nchw_kernel");
    return LWG_OK;
}

}  // extern "C"
