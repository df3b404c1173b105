// raster.hip -- SMPL mesh rasteriser + flow/condition epilogue for gfx950 (MI355X): tile-owned z-buffer in LDS.
//
// Replaces the brute-force CUDA rasteriser of the reference
//   thirdparty/neural_renderer/neural_renderer/cuda/rasterize_cuda_kernel.cu:40-84   (per-face inverse)
//   thirdparty/neural_renderer/neural_renderer/cuda/rasterize_cuda_kernel.cu:86-186  (every pixel loops all faces)
// and the python glue around it (utils/nmr.py:263-278,328-341,617-659; rasterize.py:50-52,334-338), with results
// bit-identical to the uncontracted (no fused multiply-add) evaluation of the .cu file's float expressions.
//
// Design (MI355X-first, not a translation).  The reference tests 65536 x 13776 pixel/face pairs per frame; SMPL faces
// cover ~2 pixels each.  Two launches (three from lwg_transfer_frame), no global atomics, no depth buffer in memory, no
// clear pass:
//   0. (lwg_transfer_frame) projection kernel: vertex gather + orthographic projection + y flip + look_at shift -> f2verts;
//   1. setup kernel, one lane per (frame, face): back-face cull, inverse matrix (.cu:40-84), conservative pixel box, and
//      that box in units of tiles, packed in 4 bytes.  Culled / off-screen faces get an empty box.
//   2. tile kernel, one workgroup per 32x8-pixel tile (a tile row is one 128-byte line of every per-pixel plane):
//        a. the workgroup streams the frame's packed tile boxes (4 B per face, 16 B per lane and load, L2-resident:
//           55 KB for SMPL) and appends the faces that touch its tile to a list in LDS;
//        b. one lane per listed face clips the face's pixel box to the tile and scan-converts it, resolving
//           visibility with a 64-bit LDS atomic min per covered pixel on the key (orderable(zp) << 32 | face_id):
//           the lexicographic minimum is exactly the reference's "strictly smaller depth, lowest face index wins
//           ties" rule (hazard H6), whatever the order faces are visited in.  The z-buffer (2 KB) lives and dies
//           in the workgroup's LDS; faces whose clipped box is large (slivers: the whole tile) are swept by all 256
//           lanes, one pixel each, instead;
//        c. one lane per pixel decodes the winner, recomputes its barycentrics with the same float expression
//           sequence and writes fim/wim/depth (vertically flipped, rasterize.py:334-338) and, in the fused per-frame
//           path, cond = map_fn[fim], the flow T, the warped source image and the generator's NHWC8 input --
//           five reference passes -- with plain stores.
//      The list holds 4096 faces; a tile that more faces touch (a mesh shrunk to a few pixels, a frame full of
//      slivers) flushes it through step b and keeps scanning, so every face count is handled without a fallback.
// Nothing survives between launches except f2verts and the per-face records (plain stores in one launch, plain loads
// in the next, ordinary stream order): the entry points are re-entrant.  Two code shapes of earlier versions -- the
// projection fused into the setup kernel, and a wave-per-face pixel loop for the large faces -- computed wrong values
// in single quarter-waves whenever conv_igemm_bf16x3 workgroups shared their CUs (DESIGN.md section 5.1 has the
// bisection); the present shapes have run 6000 batches beside those kernels without a differing pixel.
// Work per frame drops from 9.0e8 pair tests to ~2e5 plus 3.5e6 four-byte box tests.
//
// Exactness: this file is compiled with -ffp-contract=off and evaluates the .cu file's expressions in the same
// order.  The .cu file's double literals (hazard H5) promote four sub-expressions to double -- 0.5 * (...),
// (2. * yi + 1 - is) / is, min(max(w, 0.), 1.), 1. / (...) -- each of which takes float (or small-integer) inputs
// and is narrowed to float at once.  Every one has a float form with the identical result: halving and clamping
// to [0, 1] are exact; and for a quotient a / b of two floats (or integers < 2^24), narrowing the correctly rounded
// double quotient cannot differ from the correctly rounded float quotient, because a quotient that is not itself a
// float rounding boundary stays at least 2^-49 (relative) away from every such boundary, far more than the 2^-53
// the double rounding moves it (hipcc's float division is correctly rounded).  So the kernels use float
// instructions only -- the CPU restatement the tests compare with keeps the doubles, and the bit-exact tests against
// it over thousands of frames are the check of this paragraph (tests/test_raster_float_identities.py tries the
// identities themselves on millions of values on the CPU).  Conservative boxes are safe because a pixel can only
// pass the three float edge tests if it lies within float rounding (<< 1 px) of the triangle, except for degenerate /
// sliver faces whose edge functions are ill-conditioned; those (and non-finite or huge coordinates) get the whole
// image as their box, i.e. every tile sweeps them over all its pixels, as brute force would.
#include "common.h"
#include "sample.h"

namespace lwg {
namespace {

constexpr unsigned long long kKeyEmpty = ~0ull;
constexpr int kTileW = 32, kTileH = 8;     // pixels owned by one workgroup (one lane per pixel in the resolve step)
constexpr int kTilePix = kTileW * kTileH;
constexpr int kThreads = 256;
constexpr int kListCap = 4096;             // faces a tile can hold before it flushes its list
constexpr int kScanStep = 4 * kThreads;    // faces examined per scan step: 4 packed boxes (16 bytes) per lane
constexpr int kInlineBoxMax = 64;          // clipped boxes up to this many pixels are scan-converted by the face's lane
constexpr float kSliverRatio = 1e-4f;      // |2*area| / extent^2 below this => ill-conditioned => whole-image box
constexpr float kHugeCoord = 1.0e6f;       // pixel coordinates beyond this => whole-image box
constexpr unsigned kTileBoxEmpty = 0x0000ffffu;   // tx0 = ty0 = 255 > tx1 = ty1 = 0: touches no tile
static_assert(kTilePix == kThreads, "the resolve step maps one lane to one pixel of the tile");

struct Box {
    unsigned short x0, y0, x1, y1;
};

__device__ __forceinline__ bool backside(const float v[9])
{
    return (v[7] - v[1]) * (v[3] - v[0]) < (v[4] - v[1]) * (v[6] - v[0]);
}

// monotone map float -> uint (handles negative depths too, should a caller pass near < 0)
__device__ __forceinline__ unsigned order_bits(float z)
{
    const unsigned u = __float_as_uint(z);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unorder_bits(unsigned o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// .cu:64-81; px/py are the pixel-space vertex positions
__device__ __forceinline__ void face_inverse(const float v[9], int is, float px[3], float py[3], float inv[9],
                                             float &det)
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        px[k] = 0.5f * (v[3 * k + 0] * is + is - 1);   // .cu: 0.5 * (double)(float expr), narrowed: halving is exact
        py[k] = 0.5f * (v[3 * k + 1] * is + is - 1);
    }
    float m[9];
    m[0] = py[1] - py[2];
    m[1] = px[2] - px[1];
    m[2] = px[1] * py[2] - px[2] * py[1];
    m[3] = py[2] - py[0];
    m[4] = px[0] - px[2];
    m[5] = px[2] * py[0] - px[0] * py[2];
    m[6] = py[0] - py[1];
    m[7] = px[1] - px[0];
    m[8] = px[0] * py[1] - px[1] * py[0];
    det = px[2] * (py[0] - py[1]) + px[0] * (py[1] - py[2]) + px[1] * (py[2] - py[0]);
#pragma unroll
    for (int k = 0; k < 9; ++k) inv[k] = m[k] / det;
}

// .cu:113-114: pixel centre in normalised coordinates.  The .cu file evaluates (2. * i + 1 - is) / is in double and
// narrows; for integers below 2^13 the correctly rounded float quotient is the same number (see the note on float
// arithmetic in the header), so no double-precision instruction is needed.
__device__ __forceinline__ float pixel_centre(int i, int is) { return (float)(2 * i + 1 - is) / (float)is; }

// .cu:132-134
__device__ __forceinline__ bool inside(const float v[9], float xp, float yp)
{
    return !(((yp - v[1]) * (v[3] - v[0]) < (xp - v[0]) * (v[4] - v[1])) ||
             ((yp - v[4]) * (v[6] - v[3]) < (xp - v[3]) * (v[7] - v[4])) ||
             ((yp - v[7]) * (v[0] - v[6]) < (xp - v[6]) * (v[1] - v[7])));
}

// .cu:137-153: clamped, renormalised barycentrics and perspective-correct depth
__device__ __forceinline__ float bary_depth(const float v[9], const float inv[9], int xi, int yi, float w[3])
{
    float w_sum = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        w[k] = inv[3 * k + 0] * xi + inv[3 * k + 1] * yi + inv[3 * k + 2];
        w[k] = fminf(fmaxf(w[k], 0.f), 1.f);   // .cu clamps in double and narrows: same value
        w_sum += w[k];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) w[k] /= w_sum;
    return 1.f / (w[0] / v[2] + w[1] / v[5] + w[2] / v[8]);   // .cu: 1. / (double)(float sum), narrowed: same value
}

__device__ __forceinline__ void load_face(const float *fv, const float *fi, int fn, float v[9], float inv[9])
{
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        v[k] = fv[(size_t)fn * 9 + k];
        inv[k] = fi[(size_t)fn * 9 + k];
    }
}

// ------------------------------------------------------------------------------------------------ launch 1: faces
struct Tiling {
    int tiles_x, tiles_y;   // tiles of kTileW x kTileH pixels covering the image
    int shx, shy;           // packed tile boxes count in units of (tile << sh): at most 256 units per axis
};

Tiling tiling_for(int is)
{
    Tiling t;
    t.tiles_x = (is + kTileW - 1) / kTileW;
    t.tiles_y = (is + kTileH - 1) / kTileH;
    t.shx = t.shy = 0;
    while (((t.tiles_x - 1) >> t.shx) > 255) ++t.shx;
    while (((t.tiles_y - 1) >> t.shy) > 255) ++t.shy;
    return t;
}

// One lane per (frame, face slot); slots nf .. nfp-1 of a frame are padding (nfp = nf rounded up to 4, so that the
// tile kernel can fetch four packed boxes with one aligned 16-byte load) and only receive an empty box.  `faces` = f2verts:
// given by the caller, or written just before by project_faces_kernel.  (Round 2 first fused that projection in here,
// gather + stores ahead of the record arithmetic.  That form computed wrong records for single quarter-waves while
// conv_igemm_bf16x3 workgroups ran on the same CUs -- never alone, never beside other kernels, cause not found; the
// two-kernel form has shown none in 6000 overlapped batches: DESIGN.md section 5.1.)
__global__ __launch_bounds__(256) void raster_setup_kernel(const float *__restrict__ faces, int bs, int nf, int nfp, int is,
                                                           Tiling tl, float *__restrict__ faces_inv,
                                                           Box *__restrict__ pbox, unsigned *__restrict__ tbox)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bs * nfp) return;
    const int b = i / nfp, fn = i - b * nfp;
    unsigned packed = kTileBoxEmpty;
    // All arithmetic first, every store last, one asm block per predicate.  (The order dates from a refuted reading of the
    // co-residency miscompute of DESIGN.md section 5.1 -- stores followed by arithmetic; the cause is a packed-fp32
    // instruction form, which this file is built without: impersonator_amd/build.py, tests/test_pk_opsel_lint.py.  The
    // order is kept because it is the measured, validated one.)
    bool keep = false, has_box = false;
    float inv[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    unsigned box_lo = 0, box_hi = 0;
    const size_t t = (size_t)b * nf + (fn < nf ? fn : 0);
    if (fn < nf) {
        float v[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = faces[t * 9 + k];
        if (!backside(v)) {
            float px[3], py[3], det;
            face_inverse(v, is, px, py, inv, det);
            keep = true;

            const float xmn = fminf(fminf(px[0], px[1]), px[2]), xmx = fmaxf(fmaxf(px[0], px[1]), px[2]);
            const float ymn = fminf(fminf(py[0], py[1]), py[2]), ymx = fmaxf(fmaxf(py[0], py[1]), py[2]);
            const float ext = fmaxf(xmx - xmn, ymx - ymn);
            bool finite = true;
#pragma unroll
            for (int k = 0; k < 3; ++k) finite = finite && (px[k] - px[k] == 0.f) && (py[k] - py[k] == 0.f);
            const bool sweep_all = !finite || !(fabsf(det) > kSliverRatio * ext * ext) ||
                                   fmaxf(fmaxf(fabsf(xmn), fabsf(xmx)), fmaxf(fabsf(ymn), fabsf(ymx))) > kHugeCoord;
            int x0 = 0, y0 = 0, x1 = is - 1, y1 = is - 1;
            if (!sweep_all) {
                // pixel xi sits at pixel-space coordinate xi exactly.  A pixel outside the triangle's hull by more
                // than float rounding cannot pass all three edge tests: the products of an edge test carry ~2^-22
                // relative error, which moves an edge of a face with |2*area| > kSliverRatio * ext^2 by less than
                // ext * 2.4e-3 pixels; the margin is ten times that (at least 1/50 pixel)
                const float m = fmaxf(0.02f, ext * 0.025f);
                x0 = max(0, (int)ceilf(xmn - m));
                y0 = max(0, (int)ceilf(ymn - m));
                x1 = min(is - 1, (int)floorf(xmx + m));
                y1 = min(is - 1, (int)floorf(ymx + m));
            }
            if (x0 <= x1 && y0 <= y1) {
                has_box = true;
                box_lo = (unsigned)x0 | (unsigned)y0 << 16;      // Box {x0, y0, x1, y1} as two dwords
                box_hi = (unsigned)x1 | (unsigned)y1 << 16;
                packed = (unsigned)((x0 / kTileW) >> tl.shx) | (unsigned)((y0 / kTileH) >> tl.shy) << 8 |
                         (unsigned)((x1 / kTileW) >> tl.shx) << 16 | (unsigned)((y1 / kTileH) >> tl.shy) << 24;
            }
        }
    }
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const f4 i0 = {inv[0], inv[1], inv[2], inv[3]}, i1 = {inv[4], inv[5], inv[6], inv[7]};
    const u2 bxw = {box_lo, box_hi};
    float *pinv = faces_inv + t * 9;
    Box *pb = pbox + t;
    unsigned *pt = tbox + i;
    // all operands of all three blocks are inputs of the first one: nothing is computed between the stores
    asm volatile("; stores last" ::"v"(pinv), "v"(i0), "v"(i1), "v"(inv[8]), "v"(pb), "v"(bxw), "v"(pt), "v"(packed));
    if (keep)
        asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dwordx4 %0, %2, off offset:16\n\tglobal_store_dword %0, %3, off offset:32"
                     ::"v"(pinv), "v"(i0), "v"(i1), "v"(inv[8]) : "memory");
    if (has_box) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(pb), "v"(bxw) : "memory");
    asm volatile("global_store_dword %0, %1, off" ::"v"(pt), "v"(packed) : "memory");
}

// ------------------------------------------------------------------------------------------------ launch 2: tiles
struct ResolveOut {
    int32_t *fim;      // (bs,is,is)
    float *wim;        // (bs,is,is,3)
    float *depth;      // (bs,is,is) or null
    // fused per-frame epilogue (all optional)
    const float *map_fn; int nrows, nc; float *cond;       // cond (bs,nc,is,is)
    const float *src_p2verts; float *T;                    // (nf,3,2) -> (bs,is,is,2)
    const float *src_img; int align_corners; float *tsf_img;  // (3,is,is) -> (bs,3,is,is)
    float *x0;                                             // (bs,is,is,8)
};

// depth test of one (face, pixel) pair against the tile's z-buffer in LDS; xp / yp = pixel centre (.cu:113-114)
__device__ __forceinline__ void shade_pixel(const float v[9], const float inv[9], int fn, int xi, int yi, float xp,
                                            float yp, float near_z, float far_z, unsigned long long *key)
{
    if (!inside(v, xp, yp)) return;
    float w[3];
    const float zp = bary_depth(v, inv, xi, yi, w);
    // .cu:154-159: reject zp <= near, far <= zp; depth_min starts at far, so NaN never wins either
    if (!(zp > near_z && zp < far_z)) return;
    atomicMin(key, ((unsigned long long)order_bits(zp) << 32) | (unsigned)fn);
}

constexpr int kBigCap = 256;   // large-footprint faces of a tile kept in their own list (more: tagged in the main list)

__global__ __launch_bounds__(kThreads) void raster_tile_kernel(const float *__restrict__ faces,
                                                               const float *__restrict__ faces_inv,
                                                               const Box *__restrict__ pbox,
                                                               const unsigned *__restrict__ tbox, int nf, int nfp, int is,
                                                               Tiling tl, float near_z, float far_z, ResolveOut o)
{
    __shared__ unsigned long long sh_key[kTilePix];
    __shared__ int sh_list[kListCap];
    __shared__ int sh_big[kBigCap];
    __shared__ float sh_xp[kTileW], sh_yp[kTileH];
    __shared__ int sh_n, sh_nbig;

    const int tid = threadIdx.x;
    const int tile = blockIdx.x % (tl.tiles_x * tl.tiles_y), b = blockIdx.x / (tl.tiles_x * tl.tiles_y);
    const int ty = tile / tl.tiles_x, tx = tile - ty * tl.tiles_x;
    const int ox = tx * kTileW, oy = ty * kTileH;             // tile origin, pre-flip orientation (as the .cu tests)
    const int ex = min(ox + kTileW, is) - 1, ey = min(oy + kTileH, is) - 1;
    const unsigned cx = (unsigned)(tx >> tl.shx), cy = (unsigned)(ty >> tl.shy);

    sh_key[tid] = kKeyEmpty;
    if (tid == 0) sh_n = sh_nbig = 0;
    if (tid < kTileW) sh_xp[tid] = pixel_centre(ox + tid, is);
    else if (tid < kTileW + kTileH) sh_yp[tid - kTileW] = pixel_centre(oy + tid - kTileW, is);

    const float *fv = faces + (size_t)b * nf * 9;
    const float *fi = faces_inv + (size_t)b * nf * 9;
    const Box *fb = pbox + (size_t)b * nf;
    const uint4 *tb = reinterpret_cast<const uint4 *>(tbox + (size_t)b * nfp);

    // step b of the header: every listed face against the tile's pixels
    auto flush = [&](int n) {
        for (int e = tid; e < n; e += kThreads) {
            const int fn = sh_list[e];
            const Box bx = fb[fn];
            const int x0 = max((int)bx.x0, ox), x1 = min((int)bx.x1, ex);
            const int y0 = max((int)bx.y0, oy), y1 = min((int)bx.y1, ey);
            if (x0 > x1 || y0 > y1) continue;      // the packed box is coarser than the pixel box
            if ((x1 - x0 + 1) * (y1 - y0 + 1) > kInlineBoxMax) {
                const int slot = atomicAdd(&sh_nbig, 1);
                if (slot < kBigCap) sh_big[slot] = fn;            // swept by all lanes after the barrier
                else sh_list[e] = fn | (int)0x80000000;           // left for the all-lane sweep
                continue;
            }
            float v[9], inv[9];
            load_face(fv, fi, fn, v, inv);
            for (int yi = y0; yi <= y1; ++yi) {
                const float yp = sh_yp[yi - oy];
                for (int xi = x0; xi <= x1; ++xi)
                    shade_pixel(v, inv, fn, xi, yi, sh_xp[xi - ox], yp, near_z, far_z,
                                &sh_key[(yi - oy) * kTileW + (xi - ox)]);
            }
        }
        __syncthreads();
        const int nbig = sh_nbig;
        if (!nbig) return;
        // Faces with a large footprint in this tile (the whole-image boxes of slivers, in practice): all lanes, one pixel
        // each, face by face -- first the ones that fitted sh_big, then (a frame full of slivers) the tagged list entries.
        // (Round 2 swept them a wave per face with a pixel loop per lane.  That loop produced wrong hits in single
        // quarter-waves while conv_igemm_bf16x3 workgroups ran beside it on the CU -- never alone, never with other
        // neighbours, cause not found: DESIGN.md section 5.1.  The straight-line form below has shown none.)
        const int lx = tid & (kTileW - 1), ly = tid / kTileW;
        const int xi = ox + lx, yi = oy + ly;
        auto sweep = [&](int fn) {
            const Box bx = fb[fn];
            if (xi < (int)bx.x0 || xi > (int)bx.x1 || yi < (int)bx.y0 || yi > (int)bx.y1 || xi > ex || yi > ey) return;
            float v[9], inv[9];
            load_face(fv, fi, fn, v, inv);
            shade_pixel(v, inv, fn, xi, yi, sh_xp[lx], sh_yp[ly], near_z, far_z, &sh_key[tid]);
        };
        for (int e = 0; e < min(nbig, kBigCap); ++e) sweep(__builtin_amdgcn_readfirstlane(sh_big[e]));
        if (nbig <= kBigCap) return;
        for (int e = 0; e < n; ++e) {
            const int tagged = sh_list[e];
            if (tagged < 0) sweep(__builtin_amdgcn_readfirstlane(tagged & 0x7fffffff));
        }
    };

    // step a: stream the packed boxes.  The list length is read between two barriers (every wave sees the same value);
    // `room` scan steps are then certain to fit the list, and their loads are issued together.
    constexpr int kMaxRoom = kListCap / kScanStep;
    int f0 = 0;
    while (f0 < nfp) {
        __syncthreads();
        int n = sh_n;
        __syncthreads();
        if (kListCap - n < kScanStep) {
            flush(n);
            __syncthreads();
            if (tid == 0) sh_n = sh_nbig = 0;
            __syncthreads();
            n = 0;
        }
        const int room = (kListCap - n) / kScanStep;
        uint4 q[kMaxRoom];
#pragma unroll
        for (int s = 0; s < kMaxRoom; ++s) {
            const int f = f0 + s * kScanStep + tid * 4;
            q[s] = (s < room && f < nfp) ? tb[f >> 2] : make_uint4(kTileBoxEmpty, kTileBoxEmpty, kTileBoxEmpty, kTileBoxEmpty);
        }
#pragma unroll
        for (int s = 0; s < kMaxRoom; ++s) {
            const int f = f0 + s * kScanStep + tid * 4;
            const unsigned qq[4] = {q[s].x, q[s].y, q[s].z, q[s].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned tb0 = qq[k];
                const bool hit = cx >= (tb0 & 255u) && cy >= ((tb0 >> 8) & 255u) && cx <= ((tb0 >> 16) & 255u) && cy <= (tb0 >> 24);
                // one list allocation per wave: lane 0 reserves the wave's hits, the lanes place theirs by rank
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(hit);
                if (bal) {
                    int base = 0;
                    if ((tid & 63) == 0) base = atomicAdd(&sh_n, __builtin_popcountll(bal));
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (hit) sh_list[base + __builtin_popcountll(bal & ((1ull << (tid & 63)) - 1ull))] = f + k;
                }
            }
        }
        f0 += room * kScanStep;
    }
    __syncthreads();
    flush(sh_n);
    __syncthreads();

    // step c: one lane per pixel
    const int xi = ox + (tid & (kTileW - 1)), yi = oy + tid / kTileW;
    if (xi >= is || yi >= is) return;
    const int npix = is * is;
    const int yo = is - 1 - yi;  // the maps are flipped vertically on the way out (rasterize.py:334-338)
    const int pn = yo * is + xi;
    const size_t i = (size_t)b * npix + pn;

    const unsigned long long key = sh_key[tid];
    int fn = -1;
    float w[3] = {0.f, 0.f, 0.f};
    float zp = far_z;
    if (key != kKeyEmpty) {
        fn = (int)(unsigned)(key & 0xffffffffull);
        zp = unorder_bits((unsigned)(key >> 32));
        float v[9], inv[9];
        load_face(fv, fi, fn, v, inv);
        (void)bary_depth(v, inv, xi, yi, w);
    }
    o.fim[i] = fn;
    o.wim[i * 3 + 0] = w[0];
    o.wim[i * 3 + 1] = w[1];
    o.wim[i * 3 + 2] = w[2];
    if (o.depth) o.depth[i] = zp;

    float cnd[4] = {0.f, 0.f, 0.f, 0.f};
    if (o.cond || o.x0) {
        // utils/nmr.py:336: python negative indexing sends fim == -1 to the last (background) row
        const int row = fn >= 0 ? fn : o.nrows - 1;
        for (int c = 0; c < o.nc; ++c) {
            const float m = o.map_fn[(size_t)row * o.nc + c];
            if (c < 4) cnd[c] = m;
            if (o.cond) o.cond[((size_t)b * o.nc + c) * npix + pn] = m;
        }
    }
    if (!o.T) return;
    float tx_ = -2.f, ty_ = -2.f;  // utils/nmr.py:626
    if (fn >= 0) {
        const float *p = o.src_p2verts + (size_t)fn * 6;
        tx_ = (p[0] * w[0] + p[2] * w[1]) + p[4] * w[2];
        ty_ = (p[1] * w[0] + p[3] * w[1]) + p[5] * w[2];
    }
    *reinterpret_cast<float2 *>(o.T + i * 2) = make_float2(tx_, ty_);

    if (!(o.tsf_img || o.x0)) return;
    float rgb[3] = {0.f, 0.f, 0.f};
    const GridTaps g = grid_taps(tx_, ty_, is, is, o.align_corners);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float *pl = o.src_img + (size_t)c * npix;
        float acc = 0.f;
        if (g.vnw) acc += pl[g.y0 * is + g.x0] * g.wnw;
        if (g.vne) acc += pl[g.y0 * is + g.x0 + 1] * g.wne;
        if (g.vsw) acc += pl[(g.y0 + 1) * is + g.x0] * g.wsw;
        if (g.vse) acc += pl[(g.y0 + 1) * is + g.x0 + 1] * g.wse;
        rgb[c] = acc;
        if (o.tsf_img) o.tsf_img[((size_t)b * 3 + c) * npix + pn] = acc;
    }
    if (o.x0) {
        // the pixel's eight floats as two 16-byte stores from eight registers that are ready beforehand (left to the
        // compiler this became dwordx3 + dwordx3 + a zero pair materialised in between)
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 lo = {rgb[0], rgb[1], rgb[2], cnd[0]}, hi = {cnd[1], cnd[2], 0.f, 0.f};
        float *dst = o.x0 + i * 8;
        asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dwordx4 %0, %2, off offset:16" ::"v"(dst), "v"(lo), "v"(hi) : "memory");
    }
}

__global__ __launch_bounds__(256) void project_faces_kernel(const float *__restrict__ verts,
                                                            const float *__restrict__ cam,
                                                            const int32_t *__restrict__ faces_idx, int bs, int nv,
                                                            int nf, float eye_z, float *__restrict__ f2verts)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // (b, f, k)
    if (i >= bs * nf * 3) return;
    const int b = i / (nf * 3), fk = i - b * nf * 3;
    const int vi = faces_idx[fk];
    const float *p = verts + ((size_t)b * nv + vi) * 3;
    const float s = cam[b * 3 + 0], tx = cam[b * 3 + 1], ty = cam[b * 3 + 2];
    float *o = f2verts + (size_t)i * 3;
    o[0] = s * (p[0] + tx);          // utils/nmr.py:24
    o[1] = -(s * (p[1] + ty));       // utils/nmr.py:24 then :271
    o[2] = p[2] - eye_z;             // look_at.py:58-60 with the identity rotation of nmr.py:177
}

__global__ __launch_bounds__(256) void encode_fim_kernel(const int32_t *__restrict__ fim,
                                                         const float *__restrict__ map_fn, int bs, int npix,
                                                         int nrows, int nc, int transpose, float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bs * npix) return;
    const int b = i / npix, pn = i - b * npix;
    int row = fim[i];
    if (row < 0) row += nrows;  // python negative indexing (utils/nmr.py:336)
    for (int c = 0; c < nc; ++c) {
        const float m = map_fn[(size_t)row * nc + c];
        if (transpose) out[((size_t)b * nc + c) * npix + pn] = m;
        else out[(size_t)i * nc + c] = m;
    }
}

__global__ __launch_bounds__(256) void bc_transform_kernel(const float *__restrict__ src_f2pts, int src_bs,
                                                           const int32_t *__restrict__ fim,
                                                           const float *__restrict__ wim, int bs, int nf, int npix,
                                                           float *__restrict__ T)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bs * npix) return;
    const int b = i / npix;
    const int fn = fim[i];
    float tx = -2.f, ty = -2.f;
    if (fn >= 0) {
        const float *p = src_f2pts + ((size_t)(src_bs > 1 ? b : 0) * nf + fn) * 6;
        const float w0 = wim[(size_t)i * 3], w1 = wim[(size_t)i * 3 + 1], w2 = wim[(size_t)i * 3 + 2];
        tx = (p[0] * w0 + p[2] * w1) + p[4] * w2;
        ty = (p[1] * w0 + p[3] * w1) + p[5] * w2;
    }
    *reinterpret_cast<float2 *>(T + (size_t)i * 2) = make_float2(tx, ty);
}

struct RasterWs {
    float *faces_inv;   // (bs,nf,9)
    Box *pbox;          // (bs,nf) pixel boxes
    unsigned *tbox;     // (bs,nfp) packed tile boxes
};

inline int padded_faces(int nf) { return (nf + 3) & ~3; }

size_t raster_ws_bytes(int bs, int nf, int is)
{
    (void)is;
    size_t n = 0;
    n += align_up((size_t)bs * nf * 9 * sizeof(float), 256);
    n += align_up((size_t)bs * nf * sizeof(Box), 256);
    n += align_up((size_t)bs * padded_faces(nf) * sizeof(unsigned), 256);
    return n;
}

RasterWs carve(void *ws, int bs, int nf)
{
    char *p = static_cast<char *>(ws);
    RasterWs r;
    r.faces_inv = reinterpret_cast<float *>(p);
    p += align_up((size_t)bs * nf * 9 * sizeof(float), 256);
    r.pbox = reinterpret_cast<Box *>(p);
    p += align_up((size_t)bs * nf * sizeof(Box), 256);
    r.tbox = reinterpret_cast<unsigned *>(p);
    return r;
}

int check_raster_args(const void *faces, int bs, int nf, int is, const void *fim, const void *wim, const void *ws,
                      size_t ws_bytes)
{
    LWG_REQUIRE(faces && fim && wim, "rasterize: NULL faces/fim/wim");
    LWG_REQUIRE(bs > 0 && nf > 0 && is > 0, "rasterize: bs/nf/image_size must be positive");
    if (is > 8192 || (long)bs * nf > (1l << 30) || (long)bs * is * is > (1l << 30))
        LWG_FAIL(LWG_ERR_UNSUPPORTED, "rasterize: problem too large (bs=%d nf=%d is=%d)", bs, nf, is);
    if (!ws || ws_bytes < raster_ws_bytes(bs, nf, is) || (reinterpret_cast<uintptr_t>(ws) & 255))
        LWG_FAIL(LWG_ERR_WORKSPACE, "rasterize: workspace must be 256-byte aligned and >= %zu bytes (got %zu)",
                 raster_ws_bytes(bs, nf, is), ws_bytes);
    return LWG_OK;
}

// verts != nullptr: faces (= f2verts) is produced from the posed mesh by the setup kernel; otherwise it is the input
int run_raster(const float *verts, const float *cam, const int32_t *faces_idx, int nv, float eye_z, float *faces,
               int bs, int nf, int is, float near_z, float far_z, const ResolveOut &out, void *ws, hipStream_t st)
{
    const RasterWs w = carve(ws, bs, nf);
    const Tiling tl = tiling_for(is);
    const int nfp = padded_faces(nf);
    const int setup_blocks = ceil_div((long)bs * nfp, 256);
    if (verts) {
        project_faces_kernel<<<ceil_div((long)bs * nf * 3, 256), 256, 0, st>>>(verts, cam, faces_idx, bs, nv, nf, eye_z, faces);
        LWG_LAUNCH_CHECK("project_faces_kernel");
    }
    raster_setup_kernel<<<setup_blocks, 256, 0, st>>>(faces, bs, nf, nfp, is, tl, w.faces_inv, w.pbox, w.tbox);
    LWG_LAUNCH_CHECK("raster_setup_kernel");
    raster_tile_kernel<<<bs * tl.tiles_x * tl.tiles_y, kThreads, 0, st>>>(faces, w.faces_inv, w.pbox, w.tbox, nf, nfp, is,
                                                                           tl, near_z, far_z, out);
    LWG_LAUNCH_CHECK("raster_tile_kernel");
    return LWG_OK;
}

}  // namespace
}  // namespace lwg

using namespace lwg;

extern "C" {

int lwg_project_faces(const float *verts, const float *cam, const int32_t *faces_idx, int bs, int nv, int nf,
                      float eye_z, float *f2verts, lwg_stream_t stream)
{
    LWG_REQUIRE(verts && cam && faces_idx && f2verts, "project_faces: NULL argument");
    LWG_REQUIRE(bs > 0 && nv > 0 && nf > 0, "project_faces: bs/nv/nf must be positive");
    project_faces_kernel<<<ceil_div((long)bs * nf * 3, 256), 256, 0, as_stream(stream)>>>(verts, cam, faces_idx, bs,
                                                                                          nv, nf, eye_z, f2verts);
    LWG_LAUNCH_CHECK("project_faces_kernel");
    return LWG_OK;
}

size_t lwg_rasterize_workspace_bytes(int bs, int nf, int image_size)
{
    if (bs <= 0 || nf <= 0 || image_size <= 0) return 0;
    return raster_ws_bytes(bs, nf, image_size);
}

int lwg_rasterize_fim_wim(const float *faces, int bs, int nf, int image_size, float near_z, float far_z, int32_t *fim,
                          float *wim, float *depth, void *workspace, size_t workspace_bytes, lwg_stream_t stream)
{
    const int rc = check_raster_args(faces, bs, nf, image_size, fim, wim, workspace, workspace_bytes);
    if (rc != LWG_OK) return rc;
    ResolveOut o = {};
    o.fim = fim;
    o.wim = wim;
    o.depth = depth;
    // the setup kernel only reads `faces` on this path
    return run_raster(nullptr, nullptr, nullptr, 0, 0.f, const_cast<float *>(faces), bs, nf, image_size, near_z, far_z, o,
                      workspace, as_stream(stream));
}

int lwg_encode_fim(const int32_t *fim, const float *map_fn, int bs, int npix, int nrows, int nc, int transpose,
                   float *out, lwg_stream_t stream)
{
    LWG_REQUIRE(fim && map_fn && out, "encode_fim: NULL argument");
    LWG_REQUIRE(bs > 0 && npix > 0 && nrows > 0 && nc > 0, "encode_fim: sizes must be positive");
    encode_fim_kernel<<<ceil_div((long)bs * npix, 256), 256, 0, as_stream(stream)>>>(fim, map_fn, bs, npix, nrows, nc,
                                                                                     transpose, out);
    LWG_LAUNCH_CHECK("encode_fim_kernel");
    return LWG_OK;
}

int lwg_cal_bc_transform(const float *src_f2pts, int src_bs, const int32_t *fim, const float *wim, int bs, int nf,
                         int image_size, float *T, lwg_stream_t stream)
{
    LWG_REQUIRE(src_f2pts && fim && wim && T, "cal_bc_transform: NULL argument");
    LWG_REQUIRE(bs > 0 && nf > 0 && image_size > 0, "cal_bc_transform: sizes must be positive");
    LWG_REQUIRE(src_bs == 1 || src_bs == bs, "cal_bc_transform: src_bs must be 1 or bs (got %d vs %d)", src_bs, bs);
    const int npix = image_size * image_size;
    bc_transform_kernel<<<ceil_div((long)bs * npix, 256), 256, 0, as_stream(stream)>>>(src_f2pts, src_bs, fim, wim, bs,
                                                                                       nf, npix, T);
    LWG_LAUNCH_CHECK("bc_transform_kernel");
    return LWG_OK;
}

size_t lwg_transfer_workspace_bytes(int bs, int nf, int image_size)
{
    return lwg_rasterize_workspace_bytes(bs, nf, image_size);
}

int lwg_transfer_frame(const float *verts, const float *cam, const int32_t *faces_idx, int bs, int nv, int nf,
                       int image_size, float eye_z, float near_z, float far_z, const float *map_fn, int nc,
                       const float *src_p2verts, const float *src_img, int align_corners, float *f2verts,
                       int32_t *fim, float *wim, float *cond, float *T, float *tsf_img, float *tsf_inputs_nhwc8,
                       void *workspace, size_t workspace_bytes, lwg_stream_t stream)
{
    LWG_REQUIRE(verts && cam && faces_idx && map_fn && src_p2verts && src_img && f2verts && T,
                "transfer_frame: NULL argument");
    LWG_REQUIRE(nv > 0 && nc > 0, "transfer_frame: nv/nc must be positive");
    if (tsf_inputs_nhwc8 && nc != 3)
        LWG_FAIL(LWG_ERR_UNSUPPORTED, "transfer_frame: the NHWC8 generator input needs nc == 3 (got %d)", nc);
    int rc = check_raster_args(verts, bs, nf, image_size, fim, wim, workspace, workspace_bytes);
    if (rc != LWG_OK) return rc;
    ResolveOut o = {};
    o.fim = fim;
    o.wim = wim;
    o.map_fn = map_fn;
    o.nrows = nf + 1;
    o.nc = nc;
    o.cond = cond;
    o.src_p2verts = src_p2verts;
    o.T = T;
    o.src_img = src_img;
    o.align_corners = align_corners;
    o.tsf_img = tsf_img;
    o.x0 = tsf_inputs_nhwc8;
    return run_raster(verts, cam, faces_idx, nv, eye_z, f2verts, bs, nf, image_size, near_z, far_z, o, workspace,
                      as_stream(stream));
}

}  // extern "C"
