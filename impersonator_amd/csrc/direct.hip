// direct.hip -- LDS-resident direct convolutions on the bf16x3 matrix path (gfx950), for the layers whose im2col row
// is too ragged for the implicit-GEMM DMA ring.
//
// stem_bf16x3_kernel: the generator's first layer, Conv2d(6 -> 64, k7, s1, p3) on the NHWC8 input
// (networks/generator.py:80-84, ResUnetGenerator.encoders[0]).  Its im2col row is 49 runs of 24 bytes, so instead of
// gathering it the kernel keeps, per workgroup, the whole filter bank and the input halo of its tile in LDS as
// split-bf16 planes (conv.h) and feeds v_mfma_f32_32x32x16_bf16 straight from them:
//   * the halo is DENSE: 6 channels = 12 bytes per pixel and plane, so the 7 taps x 6 channels of one kernel row are ONE run
//     of 42 consecutive bf16 values starting at the pixel's own entry; padded to 48 they are three k-steps of 16 (K = 7 x 48
//     = 336; round 5 padded channels to 8 and taps to 8: K = 448, a third of the MFMAs on zeros).  The six padding values of a
//     run are the next halo pixel's channels (finite) against zero weights;
//   * a lane's 16-byte fragment read starts at (pixel * 12 + k-step * 32 + half * 16): dword-aligned only.  The compiler lowers
//     such a load to two ds_read2_b32 (12-byte lane pitch: every bank once per dword, conflict-free); a single ds_read_b128 at
//     such an address is legal on gfx950 but was measured 1.5x SLOWER than the padded layout (profiles/r06_stem.md);
//   * weight rows: [cout][kh][48] bf16 with a 43 x 16-byte pitch (odd: conflict-free b128 reads across 32 output channels);
//   * software pipeline: the twelve LDS instructions of k-step n+1 ride one behind each of the twelve MFMAs of k-step n;
//   * a tile = 2 image rows x 128 columns x 64 channels on four waves of 64 px x 64 ch.  A workgroup is TWO such four-wave
//     groups (512 threads) that share the 86 KiB filter bank (staged once: workgroups are persistent and walk tiles with a grid
//     stride) but own their halo and walk their own tiles, synchronised by a per-group LDS counter instead of s_barrier:
//     a tile is halo conversion -> MFMAs (half of its time) -> stores + statistics, all serial inside a wave, and with two
//     independent groups one group's MFMAs run under the other's memory phases (333 -> 233 us per 32 frames together with K = 336);
//   * epilogue identical in meaning to the implicit GEMM's: raw fp32 output + per-128-pixel (mean, M2) partials
//     reduced per 32-pixel MFMA tile and combined in a fixed order.
#include <type_traits>

#include "conv.h"

namespace lwg {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // a fragment at a dword-aligned LDS address

constexpr int ST_COLS = 128, ST_ROWS = 2;                  // output tile
constexpr int ST_HW = ST_COLS + 8, ST_HH = ST_ROWS + 6;    // halo: 3 px left, 3 + 2 right (the padded tail of a run reads one further)
constexpr int ST_PIX = 12;                                 // bytes per halo pixel and plane (6 bf16 channels)
constexpr int ST_HPLANE = ST_HH * ST_HW * ST_PIX;          // bytes per halo plane
constexpr int ST_WROW = kStemKRow * 2;                     // bytes per (output channel, kernel row) of a weight plane
constexpr int ST_WPITCH = kStemWPitch;                     // bytes per output channel of a weight plane
constexpr int ST_WPLANE = 64 * ST_WPITCH;
constexpr int ST_RED = 2 * 4 * 64 * 8;                     // statistics scratch: [row][32-px subtile][channel] float2
constexpr int ST_GROUPS = 2;                               // independent four-wave groups per workgroup (see the header)
constexpr int ST_GLDS = 2 * ST_HPLANE + ST_RED;            // LDS of one group: its halo planes and statistics scratch
constexpr int ST_LDS = 2 * ST_WPLANE + ST_GROUPS * ST_GLDS + 16;
static_assert(2 * ST_WPLANE == kStemWBytes, "host and device agree on the staged filter bank");
static_assert(ST_HPLANE % 16 == 0 && ST_LDS <= 160 * 1024, "stem tile must fit one CU's LDS");
static_assert((ST_WPITCH / 16) % 2 == 1, "odd 16-byte pitch: conflict-free weight fragment reads");

// Barrier of ONE four-wave group (the hardware barrier would couple the two groups): a monotonic LDS counter, one add per wave, then
// a poll.  LDS operations execute in issue order, so a wave's earlier LDS writes are performed before its add and a wave that has
// seen the count sees them; `target` is the count after this barrier (4 more per barrier).
__device__ __forceinline__ void stem_group_barrier(unsigned *ctr, unsigned &target, int lane)
{
    target += 4;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

__global__ __launch_bounds__(256 * ST_GROUPS) void stem_bf16x3_kernel(const StemArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *wts = lds;                        // [2 planes][64][ST_WPITCH], shared by the groups
    const int group = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
    unsigned char *halo = lds + 2 * ST_WPLANE + group * ST_GLDS;   // [2 planes][ST_HH][ST_HW] x 12 B
    float2 *red = reinterpret_cast<float2 *>(halo + 2 * ST_HPLANE);
    unsigned *gctr_all = reinterpret_cast<unsigned *>(lds + 2 * ST_WPLANE + ST_GROUPS * ST_GLDS);   // one barrier counter per group
    unsigned *gctr = gctr_all + group;
    unsigned gtarget = 0;

    const int tid = threadIdx.x & 255, lane = tid & 63;   // thread within its group
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef LWG_EXPERIMENTS   // timing only, WRONG RESULTS: 1 no output stores, 2 no statistics, 4 no MFMA loop, 8 halo converted once per workgroup
    const int dbg = a.dbg;
#else
    constexpr int dbg = 0;
#endif
    const int wrow = wave >> 1, whalf = wave & 1;    // this wave: image row wrow of the tile, columns whalf*64 .. +63

    // filter bank: staged once per workgroup, already in LDS layout
    for (int i = threadIdx.x; i < 2 * ST_WPLANE / 16; i += 256 * ST_GROUPS)
        reinterpret_cast<float4 *>(wts)[i] = reinterpret_cast<const float4 *>(a.w)[i];
    if (threadIdx.x < ST_GROUPS) gctr_all[threadIdx.x] = 0;

    // lane bases (bytes).  A: run of pixel (wrow, whalf*64 + i*32 + (lane&31)), k half lane>>5 (relative to a halo plane);
    // B: weight row of channel j*32 + (lane&31), k half lane>>5 (relative to a weight plane)
    const int a_base = ((wrow * ST_HW) + whalf * 64 + (lane & 31)) * ST_PIX + (lane >> 5) * 16;
    const int b_base = (lane & 31) * ST_WPITCH + (lane >> 5) * 16;

    const int tiles_x = a.W / ST_COLS, tiles_y = a.H / ST_ROWS;
    const int ntiles = a.N * tiles_y * tiles_x;
    // The halo of the NEXT tile is fetched into registers before the MFMAs of the current one (its global-load latency
    // rides under the matrix work) and converted / written to LDS once every wave has left the current halo.
    constexpr int HE = (ST_HH * ST_HW + 255) / 256;   // halo entries (6 channels, 24 B of fp32) per thread
    float4 pre4[HE];
    float2 pre2[HE];
    auto fetch_halo = [&](int tile) {
        const int img = tile / (tiles_y * tiles_x);
        const int trem = tile - img * (tiles_y * tiles_x);
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        const int h0 = ty * ST_ROWS, c0 = tx * ST_COLS;
        const float *xin = a.x + (size_t)img * a.H * a.W * 8;
#pragma unroll
        for (int k = 0; k < HE; ++k) {
            const int e = tid + k * 256;
            const int hy = e / ST_HW, hx = e - hy * ST_HW;
            const int gy = h0 - 3 + hy, gx = c0 - 3 + hx;
            pre4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            pre2[k] = make_float2(0.f, 0.f);
            if (e < ST_HH * ST_HW && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W) {
                const float *p = xin + ((size_t)gy * a.W + gx) * 8;
                pre4[k] = *reinterpret_cast<const float4 *>(p);
                pre2[k] = *reinterpret_cast<const float2 *>(p + 4);
            }
        }
    };
    auto store_halo = [&]() {   // fp32 NHWC8 (channels 0..5) -> two dense bf16 planes (zeros outside the image)
#pragma unroll
        for (int k = 0; k < HE; ++k) {
            const int e = tid + k * 256;
            if (e >= ST_HH * ST_HW) break;
            const float f[6] = {pre4[k].x, pre4[k].y, pre4[k].z, pre4[k].w, pre2[k].x, pre2[k].y};
            unsigned hw[3], lw[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const __bf16 h0 = (__bf16)f[2 * c], h1 = (__bf16)f[2 * c + 1];
                const __bf16 l0 = (__bf16)(f[2 * c] - (float)h0), l1 = (__bf16)(f[2 * c + 1] - (float)h1);
                hw[c] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
                lw[c] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
            }
            unsigned *ph = reinterpret_cast<unsigned *>(halo + e * ST_PIX);
            unsigned *pl = reinterpret_cast<unsigned *>(halo + ST_HPLANE + e * ST_PIX);
            ph[0] = hw[0]; ph[1] = hw[1]; ph[2] = hw[2];
            pl[0] = lw[0]; pl[1] = lw[1]; pl[2] = lw[2];
        }
    };
    const int tile0 = blockIdx.x * ST_GROUPS + group, tstride = gridDim.x * ST_GROUPS;
    if (tile0 < ntiles) fetch_halo(tile0);
    __syncthreads();   // the filter bank and the group counters are in place (the only workgroup-wide barrier)
    for (int tile = tile0; tile < ntiles; tile += tstride) {
        const int img = tile / (tiles_y * tiles_x);
        const int trem = tile - img * (tiles_y * tiles_x);
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        const int h0 = ty * ST_ROWS, c0 = tx * ST_COLS;

        if (!(dbg & 8) || tile == tile0) store_halo();
        stem_group_barrier(gctr, gtarget, lane);
        if (!(dbg & 8) && tile + tstride < ntiles) fetch_halo(tile + tstride);

        // ---- 7 kernel rows x 3 k-steps, 12 MFMAs each (2 x 2 tiles x {lo*hi, hi*lo, hi*hi})
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        // software pipeline: the twelve LDS instructions of k-step n+1 (eight ds_read2_b32 for the dword-aligned halo
        // fragments, four ds_read_b128 of weights) ride one behind each of the twelve MFMAs of k-step n
        struct Frags { f32x4 ah[2], al[2], bh[2], bl[2]; };
        auto load_frags = [&](int n, Frags &f) {
            const int kh = n / 3, s3 = n - 3 * kh;
            const int ao = a_base + kh * ST_HW * ST_PIX + s3 * 32, bo = b_base + kh * ST_WROW + s3 * 32;
            // in the order the products consume them: lo(A) x hi(B), hi(A) x lo(B), hi(A) x hi(B)
            f.al[0] = *reinterpret_cast<const f32x4u *>(halo + ST_HPLANE + ao);
            f.bh[0] = *reinterpret_cast<const f32x4 *>(wts + bo);
            f.bh[1] = *reinterpret_cast<const f32x4 *>(wts + bo + 32 * ST_WPITCH);
            f.al[1] = *reinterpret_cast<const f32x4u *>(halo + ST_HPLANE + ao + 32 * ST_PIX);
            f.ah[0] = *reinterpret_cast<const f32x4u *>(halo + ao);
            f.bl[0] = *reinterpret_cast<const f32x4 *>(wts + ST_WPLANE + bo);
            f.bl[1] = *reinterpret_cast<const f32x4 *>(wts + ST_WPLANE + bo + 32 * ST_WPITCH);
            f.ah[1] = *reinterpret_cast<const f32x4u *>(halo + ao + 32 * ST_PIX);
        };
        Frags fr[2];
        load_frags(0, fr[0]);
        if (!(dbg & 4))
#pragma unroll
        for (int n = 0; n < 21; ++n) {
            const Frags &f = fr[n & 1];
            if (n + 1 < 21) load_frags(n + 1, fr[(n + 1) & 1]);
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const f32x4 a4 = t == 0 ? f.al[i] : f.ah[i];
                        const f32x4 b4 = t == 1 ? f.bl[j] : f.bh[j];
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8_t, a4), __builtin_bit_cast(bf16x8_t, b4), acc[i][j], 0, 0, 0);
                    }
            if (n + 1 < 21) {
#pragma unroll
                for (int g2 = 0; g2 < 12; ++g2) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- raw output: C/D layout of the 32x32 MFMA: col = lane&31 (channel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel)
        // (16-byte stores through an LDS staging tile measured no faster here: the other group's MFMAs cover the store issue)
        const int col = lane & 31, rsel = 4 * (lane >> 5);
        const size_t prow = ((size_t)img * a.H + h0 + wrow) * a.W + c0 + whalf * 64;
        if (!(dbg & 1)) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int px = i * 32 + (r & 3) + 8 * (r >> 2) + rsel;
                    float *yo = a.y + (prow + px) * 64 + col;
                    yo[0] = acc[i][0][r];
                    yo[32] = acc[i][1][r];
                }
        }
        // ---- InstanceNorm partials, as igemm_epilogue: (mean, M2) per 32-pixel tile, four of them combined in order
        if (a.partials && !(dbg & 2)) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float s = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) s += acc[i][j][r];
                    s += __shfl_xor(s, 32);
                    const float mu = s * (1.f / 32.f);
                    float q = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float d = acc[i][j][r] - mu;
                        q += d * d;
                    }
                    q += __shfl_xor(q, 32);
                    if (lane < 32) red[(wrow * 4 + whalf * 2 + i) * 64 + j * 32 + col] = make_float2(mu, q);
                }
        }
        stem_group_barrier(gctr, gtarget, lane);   // statistics visible; every wave of the group is done reading the halo
        if (a.partials && tid < 128) {
            const int row = tid >> 6, c = tid & 63;
            float mean = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) mean += red[(row * 4 + w) * 64 + c].x;
            mean *= 0.25f;
            float m2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float2 pr = red[(row * 4 + w) * 64 + c];
                const float d = pr.x - mean;
                m2 += pr.y + 32.f * d * d;
            }
            const size_t mtile = (((size_t)img * a.H + h0 + row) * a.W + c0) / kConvBM;
            a.partials[mtile * 64 + c] = make_float2(mean, m2);
        }
        // the next tile's statistics cannot overtake these `red` reads: the group's next barrier comes first for the writers
    }
}

}  // namespace

bool stem_bf16x3_supported(int H, int W, int cin, int cin_pad, int cout, int k, int stride, int pad)
{
    return cin <= 6 && cin_pad == 8 && cout == 64 && k == 7 && stride == 1 && pad == 3 && W % ST_COLS == 0 && H % ST_ROWS == 0 &&
           kConvBM == 128;
}

void stem_pack_weights(const float *w, int cin, std::vector<unsigned char> &out)
{
    // PyTorch (64, cin, 7, 7) -> [plane hi|lo][cout][kh][kw * 6 + ch, 48 entries (42 used)] bf16, kStemWPitch bytes per channel
    out.assign(kStemWBytes, 0);
    __bf16 *hi = reinterpret_cast<__bf16 *>(out.data());
    __bf16 *lo = reinterpret_cast<__bf16 *>(out.data() + kStemWBytes / 2);
    for (int co = 0; co < 64; ++co)
        for (int kh = 0; kh < 7; ++kh)
            for (int kw = 0; kw < 7; ++kw)
                for (int ci = 0; ci < cin && ci < 6; ++ci) {
                    const float v = w[(((size_t)co * cin + ci) * 7 + kh) * 7 + kw];
                    const size_t idx = (size_t)co * (kStemWPitch / 2) + (size_t)kh * kStemKRow + (size_t)kw * 6 + ci;
                    const __bf16 h = (__bf16)v;
                    hi[idx] = h;
                    lo[idx] = (__bf16)(v - (float)h);
                }
}

int launch_stem_bf16x3(const StemArgs &a, hipStream_t st)
{
    if (!a.x || !a.w || !a.y) LWG_FAIL(LWG_ERR_INVALID_ARG, "stem: NULL argument");
    if (a.W % ST_COLS || a.H % ST_ROWS) LWG_FAIL(LWG_ERR_UNSUPPORTED, "stem: %dx%d is not a multiple of the 2x128 tile", a.H, a.W);
    static DeviceOnce opt_in;
    if (!opt_in.done()) {
        LWG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&stem_bf16x3_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS));
        opt_in.mark();
    }
    const int ncu = device_cu_count();
    const int ntiles = a.N * (a.H / ST_ROWS) * (a.W / ST_COLS);
    const int want = ceil_div(ntiles, ST_GROUPS), nwg = want < ncu ? want : ncu;
#ifdef LWG_EXPERIMENTS
    static const int dbg = getenv("LWG_STEM_DBG") ? atoi(getenv("LWG_STEM_DBG")) : 0;
    StemArgs b = a;
    b.dbg = dbg;
    stem_bf16x3_kernel<<<nwg, 256 * ST_GROUPS, ST_LDS, st>>>(b);
#else
    stem_bf16x3_kernel<<<nwg, 256 * ST_GROUPS, ST_LDS, st>>>(a);
#endif
    LWG_LAUNCH_CHECK("stem_bf16x3_kernel");
    return LWG_OK;
}

}  // namespace lwg
