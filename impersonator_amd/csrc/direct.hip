// direct.hip -- LDS-resident direct convolutions on the bf16x3 matrix path (gfx950), for the layers whose im2col row
// is too ragged for the implicit-GEMM DMA ring.
//
// stem_bf16x3_kernel: the generator's first layer, Conv2d(6 -> 64, k7, s1, p3) on the NHWC8 input
// (networks/generator.py:80-84, ResUnetGenerator.encoders[0]).  Its im2col row is 49 runs of 32 bytes, so instead of
// gathering it the kernel keeps, per workgroup, the whole filter bank and the input halo of its tile in LDS as
// split-bf16 planes (conv.h) and feeds v_mfma_f32_32x32x16_bf16 straight from them:
//   * one k-step (16) = two horizontally adjacent taps x 8 channels: lanes 0-31 read the 16-byte halo entry of tap
//     (kh, 2p), lanes 32-63 that of tap (kh, 2p+1); a kernel row is padded to 8 taps (zero weights), so every LDS
//     address is "lane base + immediate" and the 28 k-steps unroll without any address arithmetic;
//   * halo reads: 32 consecutive 16-byte entries per half-wave (conflict-free); weight rows have a 57-entry pitch
//     (odd: conflict-free b128 reads across 32 output channels);
//   * workgroup tile = 2 image rows x 128 columns x 64 channels, four waves of 64 px x 64 ch; workgroups are persistent
//     (the 114 KiB filter bank is staged once, then tiles are walked with a grid stride);
//   * epilogue identical in meaning to the implicit GEMM's: raw fp32 output + per-128-pixel (mean, M2) partials
//     reduced per 32-pixel MFMA tile and combined in a fixed order.
#include "conv.h"

namespace lwg {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

constexpr int ST_COLS = 128, ST_ROWS = 2;                  // output tile
constexpr int ST_HW = ST_COLS + 8, ST_HH = ST_ROWS + 6;    // halo: 3 px left, 3 + 2 right (the padded 8th tap reads one further)
constexpr int ST_HPLANE = ST_HH * ST_HW * 16;              // bytes per halo plane (16 B = 8 bf16 channels per pixel)
constexpr int ST_WPITCH = kStemWPitch;                     // bytes per output channel of a weight plane
constexpr int ST_WPLANE = 64 * ST_WPITCH;
constexpr int ST_RED = 2 * 4 * 64 * 8;                     // statistics scratch: [row][32-px subtile][channel] float2
constexpr int ST_LDS = 2 * ST_HPLANE + 2 * ST_WPLANE + ST_RED;
static_assert(2 * ST_WPLANE == kStemWBytes, "host and device agree on the staged filter bank");
static_assert(ST_LDS <= 160 * 1024, "stem tile must fit one CU's LDS");

__global__ __launch_bounds__(256) void stem_bf16x3_kernel(const StemArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *halo = lds;                       // [2 planes][ST_HH][ST_HW] x 16 B
    unsigned char *wts = lds + 2 * ST_HPLANE;        // [2 planes][64][ST_WPITCH]
    float2 *red = reinterpret_cast<float2 *>(lds + 2 * ST_HPLANE + 2 * ST_WPLANE);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wrow = wave >> 1, whalf = wave & 1;    // this wave: image row wrow of the tile, columns whalf*64 .. +63

    // filter bank: staged once per workgroup, already in LDS layout
    for (int i = tid; i < 2 * ST_WPLANE / 16; i += 256)
        reinterpret_cast<float4 *>(wts)[i] = reinterpret_cast<const float4 *>(a.w)[i];

    // lane bases (bytes).  A: halo entry of pixel (wrow, whalf*64 + i*32 + (lane&31)) at tap (0, lane>>5);
    // B: weight entry of channel j*32 + (lane&31) at tap (0, lane>>5)
    const int a_base = ((wrow * ST_HW) + whalf * 64 + (lane & 31) + (lane >> 5)) * 16;
    const int b_base = (lane & 31) * ST_WPITCH + (lane >> 5) * 16;

    const int tiles_x = a.W / ST_COLS, tiles_y = a.H / ST_ROWS;
    const int ntiles = a.N * tiles_y * tiles_x;
    // The halo of the NEXT tile is fetched into registers before the MFMAs of the current one (its global-load latency
    // rides under ~4.5 us of matrix work) and converted / written to LDS once every wave has left the current halo.
    constexpr int HE = (ST_HH * ST_HW + 255) / 256;   // halo entries (8 channels, 32 B of fp32) per thread
    float4 pre[HE][2];
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
            pre[k][0] = pre[k][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < ST_HH * ST_HW && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W) {
                const float *p = xin + ((size_t)gy * a.W + gx) * 8;
                pre[k][0] = *reinterpret_cast<const float4 *>(p);
                pre[k][1] = *reinterpret_cast<const float4 *>(p + 4);
            }
        }
    };
    auto store_halo = [&]() {   // fp32 NHWC8 -> two bf16 planes (zeros outside the image)
#pragma unroll
        for (int k = 0; k < HE; ++k) {
            const int e = tid + k * 256;
            if (e >= ST_HH * ST_HW) break;
            const float f[8] = {pre[k][0].x, pre[k][0].y, pre[k][0].z, pre[k][0].w, pre[k][1].x, pre[k][1].y, pre[k][1].z, pre[k][1].w};
            bf16x8_t hi, lo;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                hi[c] = (__bf16)f[c];
                lo[c] = (__bf16)(f[c] - (float)hi[c]);
            }
            *reinterpret_cast<bf16x8_t *>(halo + e * 16) = hi;
            *reinterpret_cast<bf16x8_t *>(halo + ST_HPLANE + e * 16) = lo;
        }
    };
    if ((int)blockIdx.x < ntiles) fetch_halo(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int img = tile / (tiles_y * tiles_x);
        const int trem = tile - img * (tiles_y * tiles_x);
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        const int h0 = ty * ST_ROWS, c0 = tx * ST_COLS;

        store_halo();
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) fetch_halo(tile + gridDim.x);

        // ---- 7 kernel rows x 4 tap pairs, 12 MFMAs each (2 x 2 tiles x {lo*hi, hi*lo, hi*hi})
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
        for (int kh = 0; kh < 7; ++kh)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int off = a_base + (i * 32 + kh * ST_HW + 2 * p) * 16;
                    ah[i] = *reinterpret_cast<const float4 *>(halo + off);
                    al[i] = *reinterpret_cast<const float4 *>(halo + ST_HPLANE + off);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int off = b_base + j * 32 * ST_WPITCH + (kh * 8 + 2 * p) * 16;
                    bh[j] = *reinterpret_cast<const float4 *>(wts + off);
                    bl[j] = *reinterpret_cast<const float4 *>(wts + ST_WPLANE + off);
                }
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const float4 a4 = t == 0 ? al[i] : ah[i];
                            const float4 b4 = t == 1 ? bl[j] : bh[j];
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(bf16x8_t, a4), __builtin_bit_cast(bf16x8_t, b4), acc[i][j], 0, 0, 0);
                        }
            }

        // ---- raw output: C/D layout of the 32x32 MFMA: col = lane&31 (channel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel)
        const int col = lane & 31, rsel = 4 * (lane >> 5);
        const size_t prow = ((size_t)img * a.H + h0 + wrow) * a.W + c0 + whalf * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int px = i * 32 + (r & 3) + 8 * (r >> 2) + rsel;
                float *yo = a.y + (prow + px) * 64 + col;
                yo[0] = acc[i][0][r];
                yo[32] = acc[i][1][r];
            }
        // ---- InstanceNorm partials, as igemm_epilogue: (mean, M2) per 32-pixel tile, four of them combined in order
        if (a.partials) {
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
        __syncthreads();   // statistics visible; every wave is done reading the halo
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
        // the next tile's halo stores cannot overtake these `red` reads: its __syncthreads() comes first for the writers
    }
}

}  // namespace

bool stem_bf16x3_supported(int H, int W, int cin_pad, int cout, int k, int stride, int pad)
{
    return cin_pad == 8 && cout == 64 && k == 7 && stride == 1 && pad == 3 && W % ST_COLS == 0 && H % ST_ROWS == 0 &&
           kConvBM == 128;
}

void stem_pack_weights(const float *w, int cin, std::vector<unsigned char> &out)
{
    // PyTorch (64, cin, 7, 7) -> [plane hi|lo][cout][kh][kw 0..7][ch 0..7] bf16, 57 16-byte entries per channel
    out.assign(kStemWBytes, 0);
    __bf16 *hi = reinterpret_cast<__bf16 *>(out.data());
    __bf16 *lo = reinterpret_cast<__bf16 *>(out.data() + kStemWBytes / 2);
    for (int co = 0; co < 64; ++co)
        for (int kh = 0; kh < 7; ++kh)
            for (int kw = 0; kw < 7; ++kw)
                for (int ci = 0; ci < cin; ++ci) {
                    const float v = w[(((size_t)co * cin + ci) * 7 + kh) * 7 + kw];
                    const size_t idx = (size_t)co * (kStemWPitch / 2) + (size_t)(kh * 8 + kw) * 8 + ci;
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
    stem_bf16x3_kernel<<<ntiles < ncu ? ntiles : ncu, 256, ST_LDS, st>>>(a);
    LWG_LAUNCH_CHECK("stem_bf16x3_kernel");
    return LWG_OK;
}

}  // namespace lwg
