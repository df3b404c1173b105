// scatter.hip -- DETERMINISTIC gradient of bilinear grid_sample wrt its input (gfx950).
//
// Replaces, in the training step (models/impersonator_trainer.py:355-356 `loss_G.backward()` through the Liquid Warping Block's
// F.grid_sample, networks/generator.py:312-315), what torch's grid_sampler_2d_backward does with float atomics: several output
// pixels may sample one source texel, and the order in which their contributions are added decides the last bits -- two runs of
// one seeded job drift apart, a captured graph cannot be compared with eager iterations, a test can only bound the drift.
//
// Here the scatter is turned into a gather with a fixed summation order:
//   plan  (depends on the flow field only; one per pyramid level and iteration, shared by every warp of that level)
//     1. count : one thread per output pixel, integer atomics -> contributions per source texel (counts are order-independent)
//     2. place : a block scan hands every texel a slot range of the contribution list (one atomic per block; WHERE a list lives
//                does not matter, only the order inside it); texels with more than kHeavy contributions are queued
//     3. fill  : every pixel appends key = pixel * 4 + tap to its texels' lists (order: whatever the atomics give)
//     4. sort  : one thread per texel sorts its (short) list by key and stores the bilinear weight beside every key
//     5. heavy : a queued texel's list of up to 4096 keys is sorted in LDS by its workgroup (bitonic); a longer one is rebuilt IN KEY
//                ORDER by a dense scan over the pixels of its image with a block-wide ordered compaction (a degenerate flow may
//                send a whole image to one texel: no quadratic sort, and at most 4P / 4096 texels can be that long)
//   apply (per warp) : one thread per (texel, 4 channels) walks the list in key order: acc = fma(w_k, dy[pixel_k], acc); dx += acc.
// Same contributions as the atomic kernel (w * dy per tap, zeros padding), summed in a fixed order: bit-reproducible.
#include "common.h"
#include "sample.h"

namespace lwg {
namespace {

constexpr int kHeavy = 32;   // longest list the per-thread insertion sort takes

struct PlanDims {
    int xn, H, W, n, Ho, Wo, align;
    long T, P;                                   // source texels, output pixels
};

// plan memory (4-byte words): count[T] fill[T] cursor heavy_count | offset[T] heavy[HC] keys[4P] weights[4P]
struct PlanView {
    int *count, *fill, *cursor, *heavy_count, *offset, *heavy, *keys;
    float *weights;
    long heavy_cap;
};
__host__ __device__ inline long heavy_capacity(long P) { return 4 * P / (kHeavy + 1) + 1; }
__host__ __device__ inline size_t plan_words(long T, long P) { return (size_t)(3 * T + 2 + heavy_capacity(P) + 8 * P); }
__host__ __device__ inline PlanView plan_view(void *mem, long T, long P)
{
    PlanView v;
    int *w = static_cast<int *>(mem);
    v.count = w; w += T;
    v.fill = w; w += T;
    v.cursor = w; w += 1;
    v.heavy_count = w; w += 1;
    v.offset = w; w += T;
    v.heavy_cap = heavy_capacity(P);
    v.heavy = w; w += v.heavy_cap;
    v.keys = w; w += 4 * P;
    v.weights = reinterpret_cast<float *>(w);
    return v;
}

// the four taps of output pixel p: texel index (or -1) and weight, tap order nw, ne, sw, se (the key's low two bits)
struct Taps4 { long tex[4]; float w[4]; };
__device__ __forceinline__ Taps4 pixel_taps(const float *__restrict__ grid, long p, const PlanDims &d)
{
    const long hw_o = (long)d.Ho * d.Wo;
    const int n = (int)(p / hw_o);
    const float2 g = *reinterpret_cast<const float2 *>(grid + p * 2);
    const GridTaps t = grid_taps(g.x, g.y, d.W, d.H, d.align);
    const long base = (long)(d.xn > 1 ? n : 0) * d.H * d.W;
    Taps4 o;
    o.tex[0] = t.vnw ? base + (long)t.y0 * d.W + t.x0 : -1;
    o.tex[1] = t.vne ? base + (long)t.y0 * d.W + t.x0 + 1 : -1;
    o.tex[2] = t.vsw ? base + (long)(t.y0 + 1) * d.W + t.x0 : -1;
    o.tex[3] = t.vse ? base + (long)(t.y0 + 1) * d.W + t.x0 + 1 : -1;
    o.w[0] = t.wnw; o.w[1] = t.wne; o.w[2] = t.wsw; o.w[3] = t.wse;
    return o;
}

// count, fill, cursor, heavy_count <- 0 (a kernel, not hipMemsetAsync: the plan is built inside captured training iterations, and a
// runtime fill kernel has no place in a step whose every launch is liblwg's)
__global__ __launch_bounds__(256) void gs_plan_zero_kernel(int *__restrict__ p, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}

__global__ __launch_bounds__(256) void gs_plan_count_kernel(const float *__restrict__ grid, PlanDims d, PlanView v)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= d.P) return;
    const Taps4 t = pixel_taps(grid, p, d);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (t.tex[k] >= 0) atomicAdd(v.count + t.tex[k], 1);
}

// 1024 texels per block (4 per thread): block-exclusive scan of the counts, one atomic on the list cursor per block
__global__ __launch_bounds__(256) void gs_plan_place_kernel(PlanDims d, PlanView v)
{
    __shared__ int wave_sum[4];
    __shared__ int block_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long t0 = ((long)blockIdx.x * 256 + tid) * 4;
    int c[4], mine = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        c[k] = t0 + k < d.T ? v.count[t0 + k] : 0;
        mine += c[k];
    }
    int incl = mine;   // inclusive scan over the wave
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const int o = __shfl_up(incl, s, 64);
        if (lane >= s) incl += o;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    if (tid == 0) {
        const int total = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
        block_base = total ? atomicAdd(v.cursor, total) : 0;
    }
    __syncthreads();
    int off = block_base + incl - mine;
    for (int w = 0; w < wave; ++w) off += wave_sum[w];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (t0 + k < d.T) {
            v.offset[t0 + k] = off;
            if (c[k] > kHeavy) {
                const int slot = atomicAdd(v.heavy_count, 1);
                if (slot < v.heavy_cap) v.heavy[slot] = (int)(t0 + k);
            }
        }
        off += c[k];
    }
}

__global__ __launch_bounds__(256) void gs_plan_fill_kernel(const float *__restrict__ grid, PlanDims d, PlanView v)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= d.P) return;
    const Taps4 t = pixel_taps(grid, p, d);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (t.tex[k] >= 0) {
            const int slot = v.offset[t.tex[k]] + atomicAdd(v.fill + t.tex[k], 1);
            v.keys[slot] = (int)(p * 4 + k);
        }
}

// one thread per texel: insertion sort of its keys (<= kHeavy), then the weight of every entry
__global__ __launch_bounds__(256) void gs_plan_sort_kernel(const float *__restrict__ grid, PlanDims d, PlanView v)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= d.T) return;
    const int c = v.count[t];
    if (c == 0 || c > kHeavy) return;
    int *keys = v.keys + v.offset[t];
    for (int i = 1; i < c; ++i) {
        const int key = keys[i];
        int j = i - 1;
        while (j >= 0 && keys[j] > key) {
            keys[j + 1] = keys[j];
            --j;
        }
        keys[j + 1] = key;
    }
    float *wts = v.weights + v.offset[t];
    for (int i = 0; i < c; ++i) {
        const int key = keys[i];
        const Taps4 tp = pixel_taps(grid, key >> 2, d);
        wts[i] = tp.w[key & 3];
    }
}

// one workgroup per queued texel (strided over the queue).
//   * up to kSortMax contributions (a minifying flow: every texel of the sampled region collects (Ho/H)^2-ish pixels -- up to
//     4P/33 texels can be queued): the keys the fill pass left in arrival order are sorted in LDS (bitonic, padded with INT_MAX)
//     and the weights looked up from the sorted keys: O(c log^2 c) per texel, O(P log^2) per plan whatever the flow;
//   * longer lists (a degenerate flow sending a whole image to a few texels; at most 4P / kSortMax of them): the list rebuilt in key
//     order by scanning the pixels of the texel's image in order, 256 at a time, with a block-wide ordered compaction.  A pixel's
//     four taps are four different texels, so a pixel contributes at most once.
// (Round 5 rescanned the image for EVERY queued texel: O(#heavy x pixels), ~1e9-1e10 tap evaluations for a strongly minifying flow.)
constexpr int kSortMax = 4096;
__global__ __launch_bounds__(256) void gs_plan_heavy_kernel(const float *__restrict__ grid, PlanDims d, PlanView v)
{
    __shared__ int wave_cnt[4];
    __shared__ int skeys[kSortMax];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int nheavy = *v.heavy_count;
    if (nheavy > v.heavy_cap) nheavy = (int)v.heavy_cap;
    const long hw_o = (long)d.Ho * d.Wo, hw_i = (long)d.H * d.W;
    for (int q = blockIdx.x; q < nheavy; q += gridDim.x) {
        const long tex = v.heavy[q];
        int *keys = v.keys + v.offset[tex];
        float *wts = v.weights + v.offset[tex];
        const int cnt = v.count[tex];
        if (cnt <= kSortMax) {
            int n2 = 64;
            while (n2 < cnt) n2 <<= 1;
            for (int i = tid; i < n2; i += 256) skeys[i] = i < cnt ? keys[i] : 0x7fffffff;
            __syncthreads();
            for (int k = 2; k <= n2; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int i = tid; i < n2; i += 256) {
                        const int l = i ^ j;
                        if (l > i) {
                            const int a0 = skeys[i], b0 = skeys[l];
                            const bool up = (i & k) == 0;
                            if ((a0 > b0) == up) { skeys[i] = b0; skeys[l] = a0; }
                        }
                    }
                    __syncthreads();
                }
            for (int i = tid; i < cnt; i += 256) {
                const int key = skeys[i];
                keys[i] = key;
                wts[i] = pixel_taps(grid, key >> 2, d).w[key & 3];
            }
            __syncthreads();   // skeys is reused by this workgroup's next texel
            continue;
        }
        // pixels that can reach this texel: those of its own image (xn == n) or of every image (xn == 1: one shared source)
        const long p0 = d.xn > 1 ? (tex / hw_i) * hw_o : 0, p1 = d.xn > 1 ? p0 + hw_o : d.P;
        int written = 0;
        for (long base = p0; base < p1; base += 256) {
            const long p = base + tid;
            int hit = -1;
            float w = 0.f;
            if (p < p1) {
                const Taps4 tp = pixel_taps(grid, p, d);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (tp.tex[k] == tex) { hit = k; w = tp.w[k]; }
            }
            const unsigned long long ball = __ballot(hit >= 0);
            const int before = __popcll(ball & ((1ull << lane) - 1ull));
            if (lane == 0) wave_cnt[wave] = __popcll(ball);
            __syncthreads();
            int off = written + before;
            for (int ww = 0; ww < wave; ++ww) off += wave_cnt[ww];
            if (hit >= 0) {
                keys[off] = (int)(p * 4 + hit);
                wts[off] = w;
            }
            written += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
            __syncthreads();
        }
    }
}

// one thread per (texel, 4 channels): dx[texel] += sum_k w_k * dy[pixel_k] in key order
__global__ __launch_bounds__(256) void gs_backward_gather_kernel(const float *__restrict__ dy, int C, PlanDims d, PlanView v, long total,
                                                                 float *__restrict__ dx)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int c4n = C >> 2;
    const long tex = e / c4n;
    const int c = (int)(e - tex * c4n) * 4;
    const int cnt = v.count[tex];
    if (cnt == 0) return;
    const int *keys = v.keys + v.offset[tex];
    const float *wts = v.weights + v.offset[tex];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < cnt; ++k) {
        const long p = keys[k] >> 2;
        const float w = wts[k];
        const float4 g = *reinterpret_cast<const float4 *>(dy + p * C + c);
        acc.x = fmaf(w, g.x, acc.x);
        acc.y = fmaf(w, g.y, acc.y);
        acc.z = fmaf(w, g.z, acc.z);
        acc.w = fmaf(w, g.w, acc.w);
    }
    float4 *o = reinterpret_cast<float4 *>(dx + tex * C + c);
    float4 cur = *o;
    cur.x += acc.x; cur.y += acc.y; cur.z += acc.z; cur.w += acc.w;
    *o = cur;
}

int make_dims(int xn, int H, int W, int n, int Ho, int Wo, int align, PlanDims *d)
{
    if (xn != 1 && xn != n) LWG_FAIL(LWG_ERR_INVALID_ARG, "grid_sample plan: input batch must be 1 or %d", n);
    if (H < 1 || W < 1 || n < 1 || Ho < 1 || Wo < 1) LWG_FAIL(LWG_ERR_INVALID_ARG, "grid_sample plan: empty tensor");
    d->xn = xn; d->H = H; d->W = W; d->n = n; d->Ho = Ho; d->Wo = Wo; d->align = align;
    d->T = (long)xn * H * W;
    d->P = (long)n * Ho * Wo;
    if (d->P * 4 > 0x7fffffffL || d->T > 0x7fffffffL) LWG_FAIL(LWG_ERR_UNSUPPORTED, "grid_sample plan: more than 2^29 output pixels");
    return LWG_OK;
}

}  // namespace
}  // namespace lwg

using namespace lwg;

size_t lwg_grid_sample_plan_bytes(int xn, int H, int W, int n, int Ho, int Wo)
{
    PlanDims d;
    if (make_dims(xn, H, W, n, Ho, Wo, 0, &d) != LWG_OK) return 0;
    return plan_words(d.T, d.P) * sizeof(int);
}

int lwg_grid_sample_plan(const float *grid, int xn, int H, int W, int n, int Ho, int Wo, int align_corners, void *plan,
                         size_t plan_bytes, lwg_stream_t stream)
{
    LWG_REQUIRE(grid && plan, "grid_sample_plan: NULL argument");
    PlanDims d;
    const int rc = make_dims(xn, H, W, n, Ho, Wo, align_corners, &d);
    if (rc != LWG_OK) return rc;
    if (plan_bytes < plan_words(d.T, d.P) * sizeof(int) || ((uintptr_t)plan & 3))
        LWG_FAIL(LWG_ERR_INVALID_ARG, "grid_sample_plan: plan buffer too small (%zu < %zu bytes) or misaligned", plan_bytes,
                 plan_words(d.T, d.P) * sizeof(int));
    const PlanView v = plan_view(plan, d.T, d.P);
    hipStream_t st = as_stream(stream);
    const long nz = 2 * d.T + 2;   // count, fill, cursor, heavy_count
    gs_plan_zero_kernel<<<(unsigned)((nz + 255) / 256), 256, 0, st>>>(v.count, nz);
    LWG_LAUNCH_CHECK("gs_plan_zero_kernel");
    const unsigned pb = (unsigned)((d.P + 255) / 256), tb = (unsigned)((d.T + 255) / 256);
    gs_plan_count_kernel<<<pb, 256, 0, st>>>(grid, d, v);
    LWG_LAUNCH_CHECK("gs_plan_count_kernel");
    gs_plan_place_kernel<<<(unsigned)((d.T + 1023) / 1024), 256, 0, st>>>(d, v);
    LWG_LAUNCH_CHECK("gs_plan_place_kernel");
    gs_plan_fill_kernel<<<pb, 256, 0, st>>>(grid, d, v);
    LWG_LAUNCH_CHECK("gs_plan_fill_kernel");
    gs_plan_sort_kernel<<<tb, 256, 0, st>>>(grid, d, v);
    LWG_LAUNCH_CHECK("gs_plan_sort_kernel");
    gs_plan_heavy_kernel<<<1024, 256, 0, st>>>(grid, d, v);   // strided over the queue; workgroups without a texel leave at once
    LWG_LAUNCH_CHECK("gs_plan_heavy_kernel");
    return LWG_OK;
}

int lwg_grid_sample_backward_planned(const float *dy, int C, int xn, int H, int W, int n, int Ho, int Wo, const void *plan,
                                     size_t plan_bytes, float *dx, lwg_stream_t stream)
{
    LWG_REQUIRE(dy && plan && dx, "grid_sample_backward_planned: NULL argument");
    if (C % 4) LWG_FAIL(LWG_ERR_UNSUPPORTED, "grid_sample_backward_planned: C=%d must be a multiple of 4", C);
    PlanDims d;
    const int rc = make_dims(xn, H, W, n, Ho, Wo, 0, &d);
    if (rc != LWG_OK) return rc;
    if (plan_bytes < plan_words(d.T, d.P) * sizeof(int))
        LWG_FAIL(LWG_ERR_INVALID_ARG, "grid_sample_backward_planned: plan buffer smaller than the plan of these dimensions");
    const PlanView v = plan_view(const_cast<void *>(plan), d.T, d.P);
    const long total = d.T * (C / 4);
    gs_backward_gather_kernel<<<(unsigned)((total + 255) / 256), 256, 0, as_stream(stream)>>>(dy, C, d, v, total, dx);
    LWG_LAUNCH_CHECK("gs_backward_gather_kernel");
    return LWG_OK;
}
