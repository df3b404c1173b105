// Bilinear sampling arithmetic shared by the flow / warp kernels (device side).
//
// Restates torch's grid_sample(bilinear, zeros) and interpolate(bilinear, align_corners=True) index and
// weight formulas, the operators the reference calls at models/imitator.py:259 and
// networks/generator.py:307,313.
#pragma once
#include <hip/hip_runtime.h>

namespace lwg {

// Source-pixel taps of one grid_sample location. Taps outside the image carry weight 0 (zeros padding).
struct GridTaps {
    int x0, y0;            // north-west tap (may be -1 .. W-1 / -1 .. H-1 or further out)
    float wnw, wne, wsw, wse;
    bool vnw, vne, vsw, vse;  // tap inside the image
};

__device__ __forceinline__ float unnormalize(float g, int size, int align_corners)
{
    // align_corners=True: -1/+1 are the centres of the corner pixels; False: the image edges
    return align_corners ? ((g + 1.f) / 2.f) * (float)(size - 1) : ((g + 1.f) * (float)size - 1.f) / 2.f;
}

__device__ __forceinline__ GridTaps grid_taps(float gx, float gy, int W, int H, int align_corners)
{
    GridTaps t;
    float ix = unnormalize(gx, W, align_corners);
    float iy = unnormalize(gy, H, align_corners);
    // keep the float->int conversion defined for wild coordinates (they sample nothing anyway)
    if (!(ix > -4.f)) ix = -4.f;
    if (!(iy > -4.f)) iy = -4.f;
    if (ix > (float)(W + 4)) ix = (float)(W + 4);
    if (iy > (float)(H + 4)) iy = (float)(H + 4);
    const float fx = floorf(ix), fy = floorf(iy);
    const float ex = fx + 1.f, ey = fy + 1.f;
    t.wnw = (ex - ix) * (ey - iy);
    t.wne = (ix - fx) * (ey - iy);
    t.wsw = (ex - ix) * (iy - fy);
    t.wse = (ix - fx) * (iy - fy);
    t.x0 = (int)fx;
    t.y0 = (int)fy;
    const bool xl = t.x0 >= 0 && t.x0 < W, xr = t.x0 + 1 >= 0 && t.x0 + 1 < W;
    const bool yt = t.y0 >= 0 && t.y0 < H, yb = t.y0 + 1 >= 0 && t.y0 + 1 < H;
    t.vnw = xl && yt;
    t.vne = xr && yt;
    t.vsw = xl && yb;
    t.vse = xr && yb;
    return t;
}

// One output sample of F.interpolate(bilinear, align_corners=True) on a 2-channel (H,W,2) field.
__device__ __forceinline__ float2 resize_flow_at(const float *__restrict__ T, int H, int W, int h, int w, int y, int x)
{
    const float sy = h > 1 ? (float)(H - 1) / (float)(h - 1) : 0.f;
    const float sx = w > 1 ? (float)(W - 1) / (float)(w - 1) : 0.f;
    const float fy = sy * (float)y, fx = sx * (float)x;
    int y0 = (int)fy, x0 = (int)fx;
    if (y0 > H - 1) y0 = H - 1;
    if (x0 > W - 1) x0 = W - 1;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float2 a = *reinterpret_cast<const float2 *>(T + ((size_t)y0 * W + x0) * 2);
    const float2 b = *reinterpret_cast<const float2 *>(T + ((size_t)y0 * W + x1) * 2);
    const float2 c = *reinterpret_cast<const float2 *>(T + ((size_t)y1 * W + x0) * 2);
    const float2 d = *reinterpret_cast<const float2 *>(T + ((size_t)y1 * W + x1) * 2);
    float2 r;
    r.x = hy * (hx * a.x + lx * b.x) + ly * (hx * c.x + lx * d.x);
    r.y = hy * (hx * a.y + lx * b.y) + ly * (hx * c.y + lx * d.y);
    return r;
}

}  // namespace lwg
