// conv.hip -- the generator's device kernels for gfx950 (MI355X, CDNA4).
//
// Replaces what PyTorch/cuDNN executes for the reference's ResUnetGenerator (networks/generator.py:8-20,
// 68-184): Conv2d 7x7/3x3 (stride 1/2), ConvTranspose2d 3x3 s2, InstanceNorm2d(affine), ReLU, the residual
// add, the Liquid Warping Block add (generator.py:283-295,303-320), torch.cat and the tanh/sigmoid heads.
//
// Layout: activations NHWC (channel-contiguous => the im2col row of a tap is one contiguous run), fp32 or the
// split-bf16 format of conv.h.
//
// conv_igemm_bf16x3 (default per-frame path): the implicit GEMM with every product evaluated as three bf16 MFMAs on
//   split operands, DMA-fed; see the comment above the kernel.  direct.hip holds the 7x7 stem of that path.
// conv_igemm_f32: implicit GEMM on the exact-fp32 matrix cores, v_mfma_f32_32x32x2_f32
//   (64 FLOP/clk/SIMD, 157.3 TFLOP/s chip peak; results are an fp32 fmaf chain, no reduced precision).
//   Workgroup tile 128 pixels x BN channels x 32 reduction, 4 waves, each wave a 32*WM x 32*WN block of
//   32x32 MFMA tiles.  Both operands are staged in LDS row-major with a 36-float row pitch and read with
//   ds_read_b128: lanes 0-31 fetch k..k+3 and lanes 32-63 fetch k+4..k+7 of their row, which feeds four
//   back-to-back MFMAs (k-pairs {k+j, k+4+j}); pitch 36 makes the 16-lane b128 groups hit 16 distinct
//   16-byte slots (bank-conflict free).  Global->register prefetch of stage t+1 is issued before the MFMAs
//   of stage t and written to the other LDS buffer afterwards: one barrier per stage.
//   Epilogue: raw output store + per-tile InstanceNorm statistics (mean, M2 over the tile's 128 pixels,
//   Chan-combinable, deterministic -- no atomics).
// in_finalize : combines tile statistics per (image, channel) -> scale/shift.
// apply       : y = act(x*scale+shift) (+ residual) (+ bilinear-warped source features), float4 over
//               channels, writes straight into channel slices of the decoder's concat buffers (torch.cat is free).
// heads       : direct 7x7 conv 64->3+1 on the vector ALU (N=4 outputs cannot feed a 32-wide MFMA tile), four pixels
//               per thread, InstanceNorm+ReLU of its input folded into the halo load, tanh/sigmoid/blend fused.
#include <type_traits>

#include "conv.h"
#include "split.h"
#include "sample.h"

namespace lwg {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#pragma clang diagnostic ignored "-Winline-asm"   // "m0 is reserved": the DMA asm below sets m0 itself before each use

constexpr int BM = kConvBM;
constexpr int BK = kConvBK;
constexpr int LDK = BK + 4;  // LDS row pitch in floats (144 B: keeps 16-B alignment, spreads banks)

// Shared epilogue: raw output store + per-tile InstanceNorm statistics.
// WIDE: the raw output goes through LDS (one 32x32 tile per wave at a time, 36-float pitch) so that it leaves as
// 16-byte stores, 8 per tile, instead of 32 four-byte ones: the epilogue is store-ISSUE bound.  Callers must have a
// barrier between their last LDS reads and this call; the statistics scratch sits behind the staging area.
// TC2 > 0: the tile's rows are a 2-D block of (BMT / TC2) image rows x TC2 columns whose origin pixel is `rem0` (row r of the
// tile = pixel rem0 + (r / TC2) * Wm + r % TC2) instead of BMT consecutive pixels.
template <int BN, int WM, int WN, bool WIDE = false, int NWAVES = 4, int BMT = BM, int TC2 = 0>
__device__ __forceinline__ void igemm_epilogue(const ConvArgs &a, const ConvPhase &ph, int phase, f32x16 (&acc)[WM][WN],
                                               float *smem, int tid, int lane, int wave_m, int wave_n, int img, int rem0, int n0,
                                               int mtile, int sub_stride = 0)
{
    // Everything below is loop-invariant in the callers: without this the compiler computes output addresses ahead of the
    // main loop and carries them through it in registers (or spills them: the eight-wave ring kernel sits at its 256).
    // The lane and thread ids are re-derived here for the same reason (nothing of the epilogue stays live across the loop).
    unsigned ones = ~0u;
    asm volatile("" : "+s"(ones));
    lane = (int)__builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u));
    tid = (wave_m * (BN / (32 * WN)) + wave_n) * 64 + lane;
    // ---- epilogue 1: raw output.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int col = lane & 31, rsel = 4 * (lane >> 5);
    if constexpr (WIDE) {
        constexpr int TP = 36;                                  // staging pitch (floats)
        float *stage = smem + (tid >> 6) * 32 * TP;              // one 32x32 tile per wave
        const int rrow = lane >> 3, rcol = (lane & 7) * 4;       // read-back: 8 rows x 8 float4 per instruction
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + rsel) * TP + col] = acc[i][j][r];
                // same wave wrote and reads: no barrier needed, only the LDS ordering of one wave
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = q * 8 + rrow;
                    const float4 v = *reinterpret_cast<const float4 *>(stage + row * TP + rcol);
                    const int trow = wave_m * 32 * WM + i * 32 + row;
                    const int rem = rem0 + (TC2 > 0 ? (trow / (TC2 > 0 ? TC2 : 1)) * a.Wm + trow % (TC2 > 0 ? TC2 : 1) : trow);
                    const int hm = rem / a.Wm, wm = rem - hm * a.Wm;
                    const size_t opix = ((size_t)img * a.Ho + hm * a.os + ph.oy0) * a.Wo + wm * a.os + ph.ox0;
                    *reinterpret_cast<float4 *>(a.y + opix * a.ldy + n0 + wave_n * 32 * WN + j * 32 + rcol) = v;
                }
            }
        smem += NWAVES * 32 * TP;   // statistics scratch behind the staging tiles
    } else {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave_m * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + rsel;
            const int rem = rem0 + (TC2 > 0 ? (row / (TC2 > 0 ? TC2 : 1)) * a.Wm + row % (TC2 > 0 ? TC2 : 1) : row);
            const int hm = rem / a.Wm, wm = rem - hm * a.Wm;
            const size_t opix = ((size_t)img * a.Ho + hm * a.os + ph.oy0) * a.Wo + wm * a.os + ph.ox0;
            float *yo = a.y + opix * a.ldy + n0 + wave_n * 32 * WN + col;
#pragma unroll
            for (int j = 0; j < WN; ++j) yo[j * 32] = acc[i][j][r];
        }
    }

    // lane ^ 32's value (__shfl_xor derives the lane id on its own: one more loop-invariant value for the compiler to hoist)
    auto other_half = [&](float v) {
        return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, v)));
    };
    // ---- epilogue 2: per-tile InstanceNorm statistics (mean, M2) per channel over the tile's 128 pixels.
    // Reduced per 32-row MFMA tile first and combined over the four row tiles in a fixed order, so the numbers do not
    // depend on which wave layout (BN/WM/WN variant) produced them: results stay bit-identical across batch sizes.
    if (a.partials) {
        float2 *red = reinterpret_cast<float2 *>(smem);  // [BM/32][BN], LDS is free again after the last barrier
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[i][j][r];
                s += other_half(s);
                const float mu = s * (1.f / 32.f);
                float q = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float d = acc[i][j][r] - mu;
                    q += d * d;
                }
                q += other_half(q);
                if (lane < 32) red[(wave_m * WM + i) * BN + wave_n * 32 * WN + j * 32 + col] = make_float2(mu, q);
            }
        __syncthreads();
        // one (mean, M2) per 128 output pixels, whatever the workgroup's tile height: a 256-row tile writes two
        if (tid < BN * (BMT / BM)) {
            constexpr int RT = BM / 32;
            const int sub = tid / BN, ch = tid - sub * BN;
            float mean = 0.f;
#pragma unroll
            for (int w = 0; w < RT; ++w) mean += red[(sub * RT + w) * BN + ch].x;
            mean *= 1.f / RT;
            float m2 = 0.f;
#pragma unroll
            for (int w = 0; w < RT; ++w) {
                const float2 p = red[(sub * RT + w) * BN + ch];
                const float d = p.x - mean;
                m2 += p.y + 32.f * d * d;
            }
            // TC2 == 0: `mtile` counts BMT-row tiles; TC2 > 0: `mtile` is the index of the tile's first 128-row block and
            // the following blocks sit `sub_stride` entries apart
            const size_t pidx = TC2 > 0 ? (size_t)mtile + (size_t)sub * sub_stride : (size_t)mtile * (BMT / BM) + sub;
            a.partials[((size_t)phase * a.mtiles + pidx) * a.Cout + n0 + ch] = make_float2(mean, m2);
        }
    }
}

// DBG is 0 in the product; tools/igemm_bench.hip instantiates ablation variants (timing only, results invalid):
//   1 no global loads in the loop, 2 no LDS staging stores, 4 no barrier, 8 no fragment re-reads, 16 s_setprio around MFMAs
// GEN = true is the general mode of ConvArgs (rows tiled across images with a masked tail, per-phase input offsets,
// bias epilogue, no statistics).
// X3 = true (general mode with ConvArgs.precision 1): same fp32 operands in memory, same loads and masks; a staged float4 is
// split into its bf16 hi and lo terms on the way to LDS (row = [hi x32 | lo x32], the layout of conv.h in a 144-byte pitch)
// and the stage is multiplied as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16: 6 MFMAs of 32 cycles per tile and
// stage instead of 16 of 64 -- the PatchGAN's 4x4 convolutions and their data gradients in the bf16x3 training mode.
template <int BN, int WM, int WN, bool SMALL_CIN, int DBG = 0, bool GEN = false, bool X3 = false>
__global__ __launch_bounds__(256) void conv_igemm_f32(const ConvArgs a)
{
    constexpr int WAVES_N = BN / (32 * WN);
    constexpr int WAVES_M = BM / (32 * WM);
    static_assert(WAVES_M * WAVES_N == 4, "four waves per workgroup");
    constexpr int B_ROWS = BN / 32;  // float4 loads per thread for the weight tile

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                  // [2][BM][LDK]
    float *Bs = smem + 2 * BM * LDK;   // [2][BN][LDK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;

    const int ksn = GEN && a.ksplit > 1 ? a.ksplit : 1;           // reduction split (general mode only): z = split * nphase + phase
    const int kss = GEN ? (int)blockIdx.z / a.nphase : 0;
    const ConvPhase ph = a.ph[GEN ? (int)blockIdx.z % a.nphase : (int)blockIdx.z];
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int hw_m = a.Hm * a.Wm;
    const int img = GEN ? 0 : m0 / hw_m;   // !GEN: a tile never straddles two images (hw_m % BM == 0)
    const int rem0 = m0 - img * hw_m;
    const int m_total = a.N * hw_m;

    // ---- loader geometry: thread -> (row = tid/8 + 32*j, 16-byte column kq = tid%8)
    // Everything a stage load needs is reduced to: one int offset per row (pixel origin of the filter window),
    // a per-row bit mask of the taps that fall inside the image, and a tap offset that is wave-uniform for
    // Cin >= 32 (tracked incrementally in scalar registers: no division, no branch in the loop).
    const int lrow = tid >> 3, kq = tid & 7;
    int aoff[4];                   // float offset of (hi0, wi0) from the image base, may be negative
    int hi0[4], wi0[4];            // SMALL_CIN: window origin, taps are checked per lane
    unsigned long long amask[4];   // !SMALL_CIN: bit t = tap t is inside the image for this row
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int rem = rem0 + lrow + 32 * j, im = 0;
        bool inb = true;
        if (GEN) {   // row -> (image, pixel); rows past the end of the batch read nothing
            inb = rem < m_total;
            rem = inb ? rem : 0;
            im = rem / hw_m;
            rem -= im * hw_m;
        }
        const int hm = rem / a.Wm, wm = rem - hm * a.Wm;
        hi0[j] = hm * a.stride - a.pad + (GEN ? ph.iy0 : 0);
        wi0[j] = wm * a.stride - a.pad + (GEN ? ph.ix0 : 0);
        aoff[j] = ((im * a.H + hi0[j]) * a.W + wi0[j]) * a.ldx + kq * 4;
        if (GEN && !inb) hi0[j] = -(1 << 28);   // every tap of this row tests as outside the image
        amask[j] = 0ull;
        if (!SMALL_CIN) {
            for (int t = 0, kh = 0, kw = 0; t < ph.ntaps; ++t) {
                const bool ok = (unsigned)(hi0[j] + kh * a.dil) < (unsigned)a.H && (unsigned)(wi0[j] + kw * a.dil) < (unsigned)a.W;
                amask[j] |= (unsigned long long)ok << t;
                if (++kw == ph.KW) { kw = 0; ++kh; }
            }
        }
    }
    const float *xin = a.x + (size_t)img * a.H * a.W * a.ldx;
    const float *wt = a.w + ph.w_off + (size_t)(n0 + lrow) * ph.Kpad + kq * 4;
    const int cin_mask = a.Cin - 1;

    float4 ra[4], rb[B_ROWS];
    unsigned rvalid = 0;  // bit j: ra[j] holds real data (else it is zero padding and is cleared when staged)
#pragma unroll
    for (int j = 0; j < 4; ++j) ra[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < B_ROWS; ++j) rb[j] = make_float4(0.f, 0.f, 0.f, 0.f);

    // stages [kb, kb + nk) of the phase's Kpad / BK are this workgroup's (all of them without a reduction split)
    const int nk_all = ph.Kpad / BK;
    const int kb = (int)((long)kss * nk_all / ksn), nk = (int)((long)(kss + 1) * nk_all / ksn) - kb;
    // scalar walk over (tap, ci0) for the wave-uniform case, started at stage kb
    int s_tap = SMALL_CIN ? 0 : (kb * BK) >> a.cin_log2, s_kh = s_tap / ph.KW, s_kw = s_tap - s_kh * ph.KW,
        s_ci0 = SMALL_CIN ? 0 : (kb * BK) & (a.Cin - 1);
    // stage fetch, split in two halves so that they can be slotted between MFMA groups
    int toff = 0;
    auto load_a = [&](int kt) {
        rvalid = 0;
        if (SMALL_CIN) {
            const int kg = kt * BK + kq * 4;
            const int tap = kg >> a.cin_log2;
            const int kh = tap / ph.KW, kw = tap - kh * ph.KW;
            toff = (kh * a.W + kw) * a.dil * a.ldx + (kg & cin_mask) - kq * 4;  // aoff already carries the lane's kq*4
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = tap < ph.ntaps && (unsigned)(hi0[j] + kh * a.dil) < (unsigned)a.H &&
                                (unsigned)(wi0[j] + kw * a.dil) < (unsigned)a.W;
                rvalid |= (unsigned)ok << j;
            }
        } else {
            toff = (s_kh * a.W + s_kw) * a.dil * a.ldx + s_ci0;
#pragma unroll
            for (int j = 0; j < 4; ++j) rvalid |= (unsigned)((amask[j] >> s_tap) & 1ull) << j;
            s_ci0 += BK;
            if (s_ci0 == a.Cin) {
                s_ci0 = 0;
                ++s_tap;
                if (++s_kw == ph.KW) { s_kw = 0; ++s_kh; }
            }
        }
        // branch-free: an out-of-image tap reads a harmless in-range address and is zeroed when staged
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int off = ((rvalid >> j) & 1u) ? aoff[j] + toff : kq * 4;
            ra[j] = *reinterpret_cast<const float4 *>(xin + off);
        }
    };
    auto load_b = [&](int kt) {
#pragma unroll
        for (int j = 0; j < B_ROWS; ++j)
            rb[j] = *reinterpret_cast<const float4 *>(wt + (size_t)(32 * j) * ph.Kpad + kt * BK);
    };
    // X3: k = 4 kq .. 4 kq + 3 of the row's 32 -> 8 bytes of hi terms at byte 8 kq, 8 bytes of lo terms 64 bytes further
    auto put_split = [&](float *row, const float4 v) {
        uint2 h, l;
        split_pair(v.x, v.y, h.x, l.x);
        split_pair(v.z, v.w, h.y, l.y);
        char *dst = reinterpret_cast<char *>(row) + kq * 8;
        *reinterpret_cast<uint2 *>(dst) = h;
        *reinterpret_cast<uint2 *>(dst + 64) = l;
    };
    auto store_a = [&](int buf) {
        float *ad = As + buf * BM * LDK + lrow * LDK;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 v = ((rvalid >> j) & 1u) ? ra[j] : make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (X3) put_split(ad + 32 * j * LDK, v);
            else *reinterpret_cast<float4 *>(ad + 32 * j * LDK + kq * 4) = v;
        }
    };
    auto store_b = [&](int buf) {
        float *bd = Bs + buf * BN * LDK + lrow * LDK;
#pragma unroll
        for (int j = 0; j < B_ROWS; ++j) {
            if constexpr (X3) put_split(bd + 32 * j * LDK, rb[j]);
            else *reinterpret_cast<float4 *>(bd + 32 * j * LDK + kq * 4) = rb[j];
        }
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frag = (lane & 31) * LDK + 4 * (lane >> 5);
    const int a_frag = (wave_m * 32 * WM) * LDK + frag;
    const int b_frag = (wave_n * 32 * WN) * LDK + frag;

    // ---- main loop.  Registers hold stage kt+1 (fetched during iteration kt-1); iteration kt stages them into the
    // idle LDS buffer, re-issues the same registers for stage kt+2 and runs the MFMAs of stage kt.  Global latency
    // is covered by a full iteration, and the staging instructions are slotted BETWEEN this wave's own 64-cycle
    // MFMAs (straight-line body, no branches) instead of in front of them.  One barrier per stage.
    auto stage_body = [&](int kt, auto do_store, auto do_load) {
        const int buf = kt & 1;
        if constexpr (X3) {
            // lane -> row lane & 31, k chunk (lane >> 5) * 8 + 16 ks of the stage's 32: 16 bytes of hi terms, 16 of lo terms
            const char *Ab = reinterpret_cast<const char *>(As + buf * BM * LDK + (wave_m * 32 * WM + (lane & 31)) * LDK) + (lane >> 5) * 16;
            const char *Bb = reinterpret_cast<const char *>(Bs + buf * BN * LDK + (wave_n * 32 * WN + (lane & 31)) * LDK) + (lane >> 5) * 16;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8_t ah[WM], al[WM], bh[WN], bl[WN];
#pragma unroll
                for (int i = 0; i < WM; ++i) {
                    ah[i] = *reinterpret_cast<const bf16x8_t *>(Ab + i * 32 * LDK * 4 + ks * 32);
                    al[i] = *reinterpret_cast<const bf16x8_t *>(Ab + i * 32 * LDK * 4 + ks * 32 + 64);
                }
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    bh[j] = *reinterpret_cast<const bf16x8_t *>(Bb + j * 32 * LDK * 4 + ks * 32);
                    bl[j] = *reinterpret_cast<const bf16x8_t *>(Bb + j * 32 * LDK * 4 + ks * 32 + 64);
                }
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                if (decltype(do_store)::value && ks == 0) {
                    store_a(buf ^ 1);
                    store_b(buf ^ 1);
                }
                if (decltype(do_load)::value && ks == 1) {
                    load_a(kb + kt + 2);
                    load_b(kb + kt + 2);
                }
            }
            __syncthreads();
            return;
        }
        const float *Ab = As + buf * BM * LDK + a_frag;
        const float *Bb = Bs + buf * BN * LDK + b_frag;
        float4 af[WM], bf[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) af[i] = *reinterpret_cast<const float4 *>(Ab + i * 32 * LDK);
#pragma unroll
        for (int j = 0; j < WN; ++j) bf[j] = *reinterpret_cast<const float4 *>(Bb + j * 32 * LDK);
#pragma unroll
        for (int k8 = 0; k8 < BK / 8; ++k8) {
            float4 an[WM], bn4[WN];
            if (k8 + 1 < BK / 8 && !(DBG & 8)) {
#pragma unroll
                for (int i = 0; i < WM; ++i) an[i] = *reinterpret_cast<const float4 *>(Ab + i * 32 * LDK + (k8 + 1) * 8);
#pragma unroll
                for (int j = 0; j < WN; ++j) bn4[j] = *reinterpret_cast<const float4 *>(Bb + j * 32 * LDK + (k8 + 1) * 8);
            }
            if (DBG & 16) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
            if (DBG & 16) __builtin_amdgcn_s_setprio(0);
            // stage kt+1 -> LDS right behind the first MFMA group, then immediately re-issue the registers for
            // stage kt+2: its loads get ~3/4 of an iteration (1.5-3k cycles) plus the barrier to land
            if (decltype(do_store)::value && k8 == 0 && !(DBG & 2)) {
                store_a(buf ^ 1);
                store_b(buf ^ 1);
            }
            if (decltype(do_load)::value && k8 == 1 && !(DBG & 1)) {
                load_a(kb + kt + 2);
                load_b(kb + kt + 2);
            }
            if (k8 + 1 < BK / 8 && !(DBG & 8)) {
#pragma unroll
                for (int i = 0; i < WM; ++i) af[i] = an[i];
#pragma unroll
                for (int j = 0; j < WN; ++j) bf[j] = bn4[j];
            }
        }
        if (!(DBG & 4)) __syncthreads();
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;

    load_a(kb);
    load_b(kb);
    store_a(0);
    store_b(0);
    if (nk > 1) {
        load_a(kb + 1);
        load_b(kb + 1);
    }
    __syncthreads();
    int kt = 0;
    for (; kt + 2 < nk; ++kt) stage_body(kt, yes{}, yes{});
    if (kt + 1 < nk) stage_body(kt++, yes{}, no{});
    stage_body(kt, no{}, no{});

    if constexpr (GEN) {
        // raw output (+ bias), rows decoded one by one; C/D layout as in igemm_epilogue
        const int col = lane & 31, rsel = 4 * (lane >> 5);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wave_m * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + rsel;
                if (m >= m_total) continue;
                const int im = m / hw_m, rem = m - im * hw_m;
                const int hm = rem / a.Wm, wm = rem - hm * a.Wm;
                const size_t opix = ((size_t)im * a.Ho + hm * a.os + ph.oy0) * a.Wo + wm * a.os + ph.ox0;
                float *yo = (ksn > 1 ? a.kpart + (size_t)kss * a.kpart_stride : a.y) + opix * a.ldy + n0 + wave_n * 32 * WN + col;
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const int co = n0 + wave_n * 32 * WN + j * 32 + col;
                    yo[j * 32] = acc[i][j][r] + (a.bias && ksn == 1 ? a.bias[co] : 0.f);
                }
            }
    } else {
        igemm_epilogue<BN, WM, WN>(a, ph, blockIdx.z, acc, smem, tid, lane, wave_m, wave_n, img, rem0, n0, blockIdx.x);
    }
}

// ------------------------------------------------------------------------------------------------
// conv_igemm_dma_f32: the same implicit GEMM with the operands streamed global -> LDS by the DMA path
// (global_load_lds_dwordx4, 1 KiB per wave instruction) into a 3-stage ring, two stages in flight.
//   * no staging VGPRs, no ds_write pass, no select/zero VALU in the loop: what remains between the MFMAs of a
//     stage is 4 + BN/32 address adds and DMA issues per wave;
//   * a DMA writes wave-uniform base + lane*16 B, so the LDS image of a tile is row-major with a 128-B pitch;
//     the b128 fragment reads stay conflict-free because each lane fetches the 16-B column (lane&7) ^ f(row),
//     f(row) = (row>>1)&7, of its row (swizzle on the SOURCE address, linear destination) and the fragment read
//     applies the same involution;
//   * zero padding: a tap outside the image makes the lane read from a 16-byte zero buffer instead;
//   * synchronisation is a raw s_barrier plus counted vmcnt: at the end of iteration t every wave waits until only
//     its DMAs of stage t+2 are outstanding (=> stage t+1 has landed), then the barrier publishes it.

template <int BN, int WM, int WN, int DBG, int NS>
__device__ __forceinline__ void igemm_dma_body(const ConvArgs &a)
{
    constexpr int WAVES_N = BN / (32 * WN);
    constexpr int WAVES_M = BM / (32 * WM);
    static_assert(WAVES_M * WAVES_N == 4, "four waves per workgroup");
    constexpr int B_ROWS = BN / 32;            // DMA instructions per wave for the weight tile
    // NS = LDS ring depth: NS-1 stages are in flight while one is being consumed
    constexpr int STAGE = (BM + BN) * BK;      // floats per stage: A [BM][32] then B [BN][32], 128-B rows
    constexpr int LPS = 4 + B_ROWS;            // DMA instructions per wave per stage

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;

    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int hw_m = a.Hm * a.Wm;
    const int img = m0 / hw_m;
    const int rem0 = m0 - img * hw_m;

    // A transposed conv is four sub-pixel phases with 1/2/2/4 taps.  Launched with gridDim.z == 1 (a.fuse_phases) one
    // workgroup walks all of them for its tile, so every workgroup carries the same 9 taps of work instead of the
    // 4-tap phase finishing long after the 1-tap one.
    // Unfused, the phases are separate workgroups; blockIdx.z counts down the phase index because the last phase has
    // the most taps and the dispatcher hands out z = 0 first (longest job first).
    const int pz0 = a.fuse_phases ? 0 : a.nphase - 1 - (int)blockIdx.z;
    const int pz1 = a.fuse_phases ? a.nphase : pz0 + 1;
    for (int pz = pz0; pz < pz1; ++pz) {
    const ConvPhase ph = a.ph[pz];

    // ---- DMA geometry: wave w, instruction j moves tile rows (w*4 + j)*8 .. +7; lane -> (row = lane>>3, slot = lane&7)
    const int lr = lane >> 3, ls = lane & 7;
    int aoff[4];                   // float offset of the lane's source chunk at tap 0 / channel 0 (may be negative)
    unsigned amask[4];             // bit t: tap t of that row is inside the image (<= 32 taps on this path)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wave * 4 + j) * 8 + lr;
        const int rem = rem0 + row;
        const int hm = rem / a.Wm, wm = rem - hm * a.Wm;
        const int hi0 = hm * a.stride - a.pad, wi0 = wm * a.stride - a.pad;
        const int src_col = ls ^ ((row >> 1) & 7);
        aoff[j] = (hi0 * a.W + wi0) * a.ldx + src_col * 4;
        unsigned m = 0;
        for (int t = 0, kh = 0, kw = 0; t < ph.ntaps; ++t) {
            const bool ok = (unsigned)(hi0 + kh * a.dil) < (unsigned)a.H && (unsigned)(wi0 + kw * a.dil) < (unsigned)a.W;
            m |= (unsigned)ok << t;
            if (++kw == ph.KW) { kw = 0; ++kh; }
        }
        amask[j] = m;
    }
    const float *xin = a.x + (size_t)img * a.H * a.W * a.ldx;
    const float *wsrc[B_ROWS];
#pragma unroll
    for (int j = 0; j < B_ROWS; ++j) {
        const int row = (wave * B_ROWS + j) * 8 + lr;
        wsrc[j] = a.w + ph.w_off + (size_t)(n0 + row) * ph.Kpad + (ls ^ ((row >> 1) & 7)) * 4;
    }

    // The DMA is issued from inline asm on purpose: hipcc treats a compiler-visible global_load_lds as a pending LDS
    // write that may alias every later ds_read and drains it with s_waitcnt vmcnt(0) -- which would serialise the
    // ring.  Hidden in asm, its completion is tracked by hand (counted vmcnt before the barrier, see below).
    // m0 carries the wave-uniform LDS byte address of the 1 KiB chunk; it is saved/restored around the instruction.
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)smem;
    auto dma16 = [&](const float *gsrc, unsigned lds_byte) {
        asm volatile("s_mov_b32 m0, %1\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, off"
                     :
                     : "v"(gsrc), "s"(lds_byte)
                     : "memory", "m0");
    };
    // wave-uniform LDS byte offsets of this wave's first A / weight chunk inside a stage
    const unsigned wave_a = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(wave * 4 * 8 * BK * 4));
    const unsigned wave_b = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((BM * BK + wave * B_ROWS * 8 * BK) * 4));
    int s_tap = 0, s_kh = 0, s_kw = 0, s_ci0 = 0;   // scalar walk over (tap, channel slice)
    // One DMA instruction occupies the wave's issue port for ~60 cycles; a stage needs LPS of them.  They are issued
    // ONE AT A TIME, each behind a chain of four MFMAs (256 cycles of matrix-pipe work already queued), never
    // back to back: piece p < 4 moves A chunk p, piece p >= 4 moves weight chunk p-4 of stage `kt` into ring slot `slot`.
    auto dma_piece = [&](int p, int kt, int slot) {
        const unsigned slot_byte = (unsigned)(slot * STAGE * 4);
        if (p < 4) {
            const unsigned sa = wave_a + slot_byte + (unsigned)(p * 8 * BK * 4);
            const int toff = (s_kh * a.W + s_kw) * a.dil * a.ldx + s_ci0;
            const float *src = ((amask[p] >> s_tap) & 1u) ? xin + (aoff[p] + toff) : a.zeros;
            dma16(src, sa);
        } else {
            const unsigned sb = wave_b + slot_byte + (unsigned)((p - 4) * 8 * BK * 4);
            dma16(wsrc[p - 4] + kt * BK, sb);
        }
        if (p == LPS - 1) {
            s_ci0 += BK;
            if (s_ci0 == a.Cin) {
                s_ci0 = 0;
                ++s_tap;
                if (++s_kw == ph.KW) { s_kw = 0; ++s_kh; }
            }
        }
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addressing: row = base + (lane&31); 16-B column (2*k8 + (lane>>5)) ^ f(row); f only depends on lane
    const int frow = lane & 31;
    const int fsw = (frow >> 1) & 7;
    const int a_row = (wave_m * 32 * WM + frow) * BK;
    const int b_row = BM * BK + (wave_n * 32 * WN + frow) * BK;
    int fcol[BK / 8];
#pragma unroll
    for (int k8 = 0; k8 < BK / 8; ++k8) fcol[k8] = (((2 * k8 + (lane >> 5)) ^ fsw) * 4);

    // iteration kt: MFMAs of stage kt from ring slot `slot`, DMA pieces of stage kt+2 (if any) in their shadow
    auto stage_body = [&](int kt, int slot, auto do_dma) {
        const float *As = smem + slot * STAGE + a_row;
        const float *Bs = smem + slot * STAGE + b_row;
        int slot2 = slot + (NS - 1);
        if (slot2 >= NS) slot2 -= NS;
        float4 af[WM], bf[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) af[i] = *reinterpret_cast<const float4 *>(As + i * 32 * BK + fcol[0]);
#pragma unroll
        for (int j = 0; j < WN; ++j) bf[j] = *reinterpret_cast<const float4 *>(Bs + j * 32 * BK + fcol[0]);
#pragma unroll
        for (int k8 = 0; k8 < BK / 8; ++k8) {
            float4 an[WM], bn4[WN];
            if (k8 + 1 < BK / 8 && !(DBG & 8)) {
#pragma unroll
                for (int i = 0; i < WM; ++i) an[i] = *reinterpret_cast<const float4 *>(As + i * 32 * BK + fcol[k8 + 1]);
#pragma unroll
                for (int j = 0; j < WN; ++j) bn4[j] = *reinterpret_cast<const float4 *>(Bs + j * 32 * BK + fcol[k8 + 1]);
            }
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                    // piece p rides behind chain p; where a stage has more pieces than chains (the 32-channel tile: 5 pieces, 4
                    // chains) the last chain takes the rest -- in order: the last piece advances the (tap, channel) walk
                    constexpr int CHAINS = (BK / 8) * WM * WN;
                    const int piece = k8 * (WM * WN) + i * WN + j;
                    if (decltype(do_dma)::value && !(DBG & 1)) {
#pragma unroll
                        for (int pp = piece; pp < LPS; ++pp) {
                            if (pp != piece && piece != CHAINS - 1) break;
                            __builtin_amdgcn_sched_barrier(0);   // keep the piece where it is: behind this MFMA chain
                            dma_piece(pp, kt + (NS - 1), slot2);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            if (k8 + 1 < BK / 8 && !(DBG & 8)) {
#pragma unroll
                for (int i = 0; i < WM; ++i) af[i] = an[i];
#pragma unroll
                for (int j = 0; j < WN; ++j) bf[j] = bn4[j];
            }
        }
        // stage kt+1 must have landed (for every wave) before anyone reads it: only this iteration's own pieces
        // (stage kt+2) may still be in flight
        if (decltype(do_dma)::value) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LPS) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tail: drain (a few iterations, not worth counting)
        }
        if (!(DBG & 4)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    static_assert(LPS <= 2 * (BK / 8) * WM * WN, "not enough MFMA chains to hide the DMA pieces of a stage");

    const int nk = ph.Kpad / BK;
    // prologue: NS-1 stages in flight, the first one must have landed
#pragma unroll
    for (int st = 0; st < NS - 1; ++st)
        if (st < nk) {
#pragma unroll
            for (int p = 0; p < LPS; ++p) dma_piece(p, st, st);
        }
    if (nk >= NS - 1) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LPS) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int slot = 0, kt = 0;
    for (; kt + (NS - 1) < nk; ++kt) {
        stage_body(kt, slot, yes{});
        if (++slot == NS) slot = 0;
    }
    for (; kt < nk; ++kt) {
        stage_body(kt, slot, no{});
        if (++slot == NS) slot = 0;
    }
    igemm_epilogue<BN, WM, WN, true>(a, ph, pz, acc, smem, tid, lane, wave_m, wave_n, img, rem0, n0, blockIdx.x);
    __syncthreads();   // the staging / statistics scratch aliases the ring: finish reading it before the next phase's DMA
    }  // phase loop
}

template <int BN, int WM, int WN, int DBG = 0, int NS = 3>
__global__ __launch_bounds__(256) void conv_igemm_dma_f32(const ConvArgs a)
{
    igemm_dma_body<BN, WM, WN, DBG, NS>(a);
}

// ------------------------------------------------------------------------------------------------
// conv_igemm_bf16x3: the bf16x3 implicit GEMM (conv.h, "split-bf16 format") -- DMA-fed ring like conv_igemm_dma_f32,
// same 128-B rows, swizzle, ring and epilogue -- built around what a 32-cycle MFMA leaves room for (one wave per SIMD
// hides about five other instructions per MFMA):
//   * fragments: 16-B column 2*kb + (lane>>5) of a row is the hi operand of k-block kb (8 bf16 per lane = one
//     v_mfma_f32_32x32x16_bf16 operand), column 4 + 2*kb + (lane>>5) the lo operand; a stage is 2 k-blocks x 3
//     products (lo*hi, hi*lo, hi*hi) per 32x32 tile, tiles walked inside each product so that consecutive MFMAs never
//     share an accumulator;
//   * every DMA is "SGPR base + 32-bit lane offset": the base walks the channel slice / weight row by 128 B per stage,
//     the lane offsets change only with the tap (a real, wave-uniform branch), padding taps point at the zero run
//     behind the input tensor (a.zeros).  No address VALU in the steady state.
//   * ring slots are compile-time (whole turns of the ring unrolled): every LDS address is an immediate.
//   * the stage barrier sits 3/4 into the MFMA stream; behind it the next stage's first fragments are fetched under
//     the remaining MFMAs.
// Measured dead ends (tools/igemm_bench.hip): eight waves splitting the two k-blocks of a stage (two per SIMD, partial
// tiles added through LDS) 343 vs 409 TFLOP/s; 4-slot ring =; barrier at 1/2 instead of 3/4 =; re-dealing the tiles so
// that an XCD (own L2) covers 2 n-tiles x 16 m-tiles instead of 4 x 8 (fabric reads 107 -> ~71 MB per trunk launch) -2%.  rocprofv3: LDS 18% busy,
// no bank conflicts; the matrix pipe is 49% busy at a power-limited 2.1-2.2 GHz (a pure MFMA loop on random operands
// reaches 80% of the 2.5 PFLOP/s dense peak on this part, tools/mfma_peak.hip).
template <int BN, int WM, int WN, int NS = 3, int DBG = 0, int BMT = BM>
__global__ __launch_bounds__(64 * (BMT / (32 * WM)) * (BN / (32 * WN))) void conv_igemm_bf16x3(const ConvArgs a)
{
    // BMT = output pixels per workgroup tile: 128, or 256 for the 64-channel layers (256x64: the 2x2 wave tile of the
    // 128x128 kernel, i.e. one B fragment read per two MFMAs instead of one per MFMA on the 128x64 tile)
    constexpr int WAVES_N = BN / (32 * WN), WAVES_M = BMT / (32 * WM), NW = WAVES_M * WAVES_N;
    // 4 waves: one per SIMD.  8 waves: two per SIMD, each owning half as many 32x32 tiles -- while one of a SIMD's
    // two waves sits in a DMA issue (~60 cycles), a counted wait or the stage barrier, the other one's MFMAs keep the
    // matrix pipe busy.
    static_assert((NW == 4 || NW == 8) && NS >= 3, "wave layout");
    // DBG & 1024 (timing experiment only, results are garbage): one activation chunk per wave and stage instead of four --
    // the DMA count a halo-resident activation operand would have
    constexpr int A_CH = (DBG & 1024) ? 1 : BMT / 8 / NW, B_CH = BN / 8 / NW;   // 1-KiB DMA chunks (8 rows x 128 B) per wave per stage
    constexpr int LPS = A_CH + B_CH;
    constexpr int STAGE = (BMT + BN) * BK;                // floats (4-byte units) per ring slot
    constexpr int TILES = WM * WN;
    constexpr int Q = 6 * TILES;                          // MFMAs of a wave per stage
    static_assert(Q >= LPS, "at most one DMA piece per MFMA");
    constexpr int QB = Q * 3 / 4;                         // stage barrier before this MFMA (2/3 measured no better)
    // piece p is issued behind MFMA (p + 1) * Q / LPS - 1: evenly spread, the last one behind the last MFMA
    constexpr int ISSUED = (QB * LPS) / Q;                // pieces of the stage already issued before the barrier

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    // Workgroups go to the 8 XCDs (each with its own 4 MiB L2) round-robin by linear id.  In natural order the three
    // image rows a 3x3 tile needs belong to tiles on three other XCDs, so every L2 fetches them again: 1.2 GB of
    // fabric reads for the 268 MB input of the 256x256 skipper.  Re-dealt, XCD k owns the contiguous band of m-tiles
    // [k, k+1) * gridDim.x / 8 and walks it in order: neighbouring rows meet in one L2.
    int bx = blockIdx.x;
    const int by = blockIdx.y;
    if (!(DBG & 64) && !a.natural_order && (gridDim.x & 7) == 0) bx = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int m0 = bx * BMT, n0 = by * BN;
    const int hw_m = a.Hm * a.Wm, img = __builtin_amdgcn_readfirstlane(m0 / hw_m), rem0 = m0 - img * hw_m;   // the division runs on the VALU
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)smem;
    auto uniform_ptr = [](const void *p) {   // wave-uniform pointer pinned to SGPRs (the "s" asm operand below)
        const unsigned long long v = (unsigned long long)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const char *>(((unsigned long long)hi << 32) | lo);
    };
    // compiler-invisible on purpose (see conv_igemm_dma_f32): completion is tracked with counted vmcnt
    auto dma16 = [&](unsigned voff, const char *sbase, unsigned lds_byte) {
        asm volatile("s_mov_b32 m0, %2\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, %1"
                     :
                     : "v"(voff), "s"(sbase), "s"(lds_byte)
                     : "memory", "m0");
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    // DBG & 512 (tools/conv_trace.py): per-wave cycle accounting with s_memtime around the stage's two blocking sites.
    // The three reads of a stage are consumed at the NEXT stage's site, behind the lgkmcnt(0) that is there anyway, so
    // the instrumented kernel has no wait the production kernel does not have.
    unsigned long long tr_a = 0, tr_b = 0, tr_c = 0, tr_wait = 0, tr_bar = 0, tr_t0 = 0, tr_loop0 = 0, tr_loop1 = 0;
    unsigned long long tr_d0 = 0, tr_d1 = 0, tr_d2 = 0, tr_d3 = 0, tr_dma = 0, tr_rt0 = 0;
    unsigned tr_stages = 0;
    if (DBG & 512) {
        tr_t0 = __builtin_amdgcn_s_memtime();
        tr_rt0 = __builtin_amdgcn_s_memrealtime();   // constant 100 MHz: gives the shader clock the kernel really ran at
    }

    // unfused phases: blockIdx.z counts the phase index down (most taps first, see conv_igemm_dma_f32)
    const int pz0 = a.fuse_phases ? 0 : a.nphase - 1 - (int)blockIdx.z;
    const int pz1 = a.fuse_phases ? a.nphase : pz0 + 1;
    for (int pz = pz0; pz < pz1; ++pz) {
    const ConvPhase ph = a.ph[pz];

    // ---- DMA geometry: chunk c of a stage = tile rows 8c .. 8c+7; lane -> (row = lane>>3, 16-B slot = lane&7);
    //      the source column is swizzled, (lane&7) ^ ((row>>1)&7), the LDS image is linear (conflict-free b128 reads)
    const int lr = lane >> 3, ls = lane & 7;
    int aoff[A_CH];
    unsigned amask[A_CH];
#pragma unroll
    for (int j = 0; j < A_CH; ++j) {
        const int row = (wave * A_CH + j) * 8 + lr;
        const int rem = rem0 + row;
        const int hm = rem / a.Wm, wm = rem - hm * a.Wm;
        const int hi0 = hm * a.stride - a.pad, wi0 = wm * a.stride - a.pad;
        aoff[j] = (hi0 * a.W + wi0) * a.ldx + (ls ^ ((row >> 1) & 7)) * 4;
        unsigned m = 0;
        for (int t = 0, kh = 0, kw = 0; t < ph.ntaps; ++t) {
            const bool ok = (unsigned)(hi0 + kh * a.dil) < (unsigned)a.H && (unsigned)(wi0 + kw * a.dil) < (unsigned)a.W;
            m |= (unsigned)ok << t;
            if (++kw == ph.KW) { kw = 0; ++kh; }
        }
        amask[j] = m;
    }
    const float *xin = a.x + (size_t)img * a.H * a.W * a.ldx;
    const char *x_base = uniform_ptr(xin);
    const char *w_base = uniform_ptr(a.w_split + ph.w_off);
    const unsigned zoff_b = (unsigned)((const char *)a.zeros - (const char *)xin);
    unsigned wvoff[B_CH], cur[A_CH];
#pragma unroll
    for (int j = 0; j < B_CH; ++j) {
        const int row = (wave * B_CH + j) * 8 + lr;
        wvoff[j] = (unsigned)(((n0 + row) * ph.Kpad + (ls ^ ((row >> 1) & 7)) * 4) * 4);
    }
    int s_tap = 0, s_kh = 0, s_kw = 0;
    unsigned s_ci_b = 0, s_w_b = 0;   // channel-slice byte offset of the activations, K byte offset of the weight rows
    auto retap = [&]() {
        const int tapoff = (s_kh * a.W + s_kw) * a.dil * a.ldx;
#pragma unroll
        for (int p = 0; p < A_CH; ++p) cur[p] = ((amask[p] >> s_tap) & 1u) ? (unsigned)((aoff[p] + tapoff) * 4) : zoff_b;
    };
    retap();
    const unsigned wave_a = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(wave * A_CH * 8 * BK * 4));
    const unsigned wave_b = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((BMT * BK + wave * B_CH * 8 * BK) * 4));
    // piece p of stage kt into ring slot `slot`: p < A_CH activation chunks, then the weight chunks
    auto dma_piece = [&](int p, int kt, int slot) {
        const unsigned slot_byte = (unsigned)(slot * STAGE * 4);
        if (p < A_CH) {
            dma16(cur[p], x_base + s_ci_b, wave_a + slot_byte + (unsigned)(p * 8 * BK * 4));
        } else {
            dma16(wvoff[p - A_CH], w_base + __builtin_amdgcn_readfirstlane(s_w_b), wave_b + slot_byte + (unsigned)((p - A_CH) * 8 * BK * 4));
        }
        if (p == LPS - 1) {
            // next stage's slice of the reduction.  The weight rows are [tap][ci]; the walk order is free:
            if (a.tap_inner) {
                // taps innermost: the nine taps of one 32-channel slice are consecutive stages, so a tile re-reads
                // the 128-byte lines of its own pixel neighbourhood (9 x 16 KiB loaded, ~26 KiB distinct) while they
                // are still in L1/L2.  Channel-innermost, a pixel's line comes back Cin/32 stages later, after the
                // XCD's 32 workgroups have pushed 16 MB through its 4 MiB L2: every tap was a fabric fetch.
                ++s_tap;
                if (++s_kw == ph.KW) { s_kw = 0; ++s_kh; }
                if (s_tap == ph.ntaps) { s_tap = 0; s_kh = 0; s_kw = 0; s_ci_b += BK * 4; }
                s_w_b = (unsigned)(s_tap * a.Cin) * 4u + s_ci_b;
                retap();
            } else {
                s_w_b += BK * 4;
                s_ci_b += BK * 4;
                if (s_ci_b == (unsigned)a.Cin * 4) {
                    asm volatile("; next tap" ::: "memory");   // keeps this rare path a real (wave-uniform) branch
                    s_ci_b = 0;
                    ++s_tap;
                    if (++s_kw == ph.KW) { s_kw = 0; ++s_kh; }
                    retap();
                }
            }
        }
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragments: row = base + (lane&31); hi operand of k-block kb = 16-B column 2*kb + (lane>>5), lo = 4 + that
    const int frow = lane & 31, fsw = (frow >> 1) & 7;
    const int a_row = (wave_m * 32 * WM + frow) * BK;
    const int b_row = BMT * BK + (wave_n * 32 * WN + frow) * BK;
    int fcol[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) fcol[c] = (((2 * c + (lane >> 5)) ^ fsw) * 4);
    struct Frags { float4 ah[WM], al[WM], bh[WN], bl[WN]; };
    auto load_frags = [&](int slot, int kb, Frags &f) {
        const float *As = smem + slot * STAGE + a_row;
        const float *Bs = smem + slot * STAGE + b_row;
        const int ch = fcol[kb], cl = fcol[2 + kb];
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            f.ah[i] = *reinterpret_cast<const float4 *>(As + i * 32 * BK + ch);
            f.al[i] = *reinterpret_cast<const float4 *>(As + i * 32 * BK + cl);
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            f.bh[j] = *reinterpret_cast<const float4 *>(Bs + j * 32 * BK + ch);
            f.bl[j] = *reinterpret_cast<const float4 *>(Bs + j * 32 * BK + cl);
        }
    };
    // one fragment register of load_frags, in the order the products consume them (lo(A), hi(B) first)
    auto load_one = [&](int slot, int kb, Frags &f, int idx) {
        const float *As = smem + slot * STAGE + a_row;
        const float *Bs = smem + slot * STAGE + b_row;
        const int ch = fcol[kb], cl = fcol[2 + kb];
        if (idx < WM) f.al[idx] = *reinterpret_cast<const float4 *>(As + idx * 32 * BK + cl);
        else if (idx < WM + WN) f.bh[idx - WM] = *reinterpret_cast<const float4 *>(Bs + (idx - WM) * 32 * BK + ch);
        else if (idx < 2 * WM + WN) f.ah[idx - WM - WN] = *reinterpret_cast<const float4 *>(As + (idx - WM - WN) * 32 * BK + ch);
        else f.bl[idx - 2 * WM - WN] = *reinterpret_cast<const float4 *>(Bs + (idx - 2 * WM - WN) * 32 * BK + cl);
    };
    constexpr int NL = 2 * (WM + WN);
    constexpr int RPM = (NL + (Q - QB) - 1) / (Q - QB);   // next-stage reads per MFMA after the barrier
    Frags fr;   // k-block-0 fragments of the stage about to run (fetched during the previous stage)

    auto stage_body = [&](int kt, auto slot_c, auto do_dma) {
        constexpr int slot = decltype(slot_c)::value;
        constexpr int slot1 = (slot + 1) % NS, slot2 = (slot + NS - 1) % NS;
        Frags f1, nx;
        if (DBG & 256) load_frags(slot, 1, f1);   // ablation: the reads in two bulk groups ahead of the MFMAs (2-3% slower)
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int kb = q / (3 * TILES), r = q % (3 * TILES);
            const int t = r / TILES, i = (r / WN) % WM, j = r % WN;   // cross terms first, hi*hi last
            if (q == QB) {
                __builtin_amdgcn_sched_barrier(0);
                if (DBG & 512) {
                    if (tr_stages) {   // the previous stage's reads have landed
                        tr_wait += tr_b - tr_a; tr_bar += tr_c - tr_b;
                        tr_dma += (tr_d1 - tr_d0) + (tr_d3 - tr_d2);
                    }
                    ++tr_stages;
                    tr_a = __builtin_amdgcn_s_memtime();
                }
                // stage kt+1 must have landed for every wave (only younger pieces may be in flight) and every wave
                // must be done reading this slot's predecessor before the pieces issued below overwrite it
                if (decltype(do_dma)::value) {
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 3) * LPS + ISSUED) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                }
                if (DBG & 512) tr_b = __builtin_amdgcn_s_memtime();
                if (!(DBG & 4)) __builtin_amdgcn_s_barrier();
                if (DBG & 512) tr_c = __builtin_amdgcn_s_memtime();
                asm volatile("" ::: "memory");
                if (DBG & 256) {
                    load_frags(slot1, 0, nx);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            const Frags &f = kb == 0 ? fr : f1;
            const float4 a4 = t == 0 ? f.al[i] : f.ah[i];
            const float4 b4 = t == 1 ? f.bl[j] : f.bh[j];
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a4),
                                                                __builtin_bit_cast(bf16x8_t, b4), acc[i][j], 0, 0, 0);
            if (!(DBG & 8) && !(DBG & 256)) {
                // fragment reads ride behind the MFMAs, one or two each, so the matrix pipe never waits for a burst of
                // LDS issue slots: k-block 1 of this stage behind the first NL products, k-block 0 of the next stage
                // behind the first ones after the barrier
                if (q < NL) {
                    load_one(slot, 1, f1, q);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                } else if (q >= QB && RPM * (q - QB) < NL) {
#pragma unroll
                    for (int r2 = 0; r2 < RPM; ++r2)
                        if (RPM * (q - QB) + r2 < NL) load_one(slot1, 0, nx, RPM * (q - QB) + r2);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, RPM, 0);
                }
            }
            if (decltype(do_dma)::value && !(DBG & 1)) {
                const int pc = ((q + 1) * LPS) / Q;           // pieces due once MFMA q is issued ...
                const int pb = (q * LPS) / Q;                 // ... and before it: equal unless a piece rides behind q
                if (pc != pb) {
                    __builtin_amdgcn_sched_barrier(0);
                    // traced twin: cycles from before to after the issue of the stage's first activation piece and
                    // first weight piece (both ahead of the stage's blocking site, where the reads are consumed)
                    if ((DBG & 512) && pb == 0) tr_d0 = __builtin_amdgcn_s_memtime();
                    if ((DBG & 512) && pb == A_CH) tr_d2 = __builtin_amdgcn_s_memtime();
                    dma_piece(pb, kt + (NS - 1), slot2);
                    if ((DBG & 512) && pb == 0) tr_d1 = __builtin_amdgcn_s_memtime();
                    if ((DBG & 512) && pb == A_CH) tr_d3 = __builtin_amdgcn_s_memtime();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (!(DBG & 8)) fr = nx;
    };

    const int nk = ph.Kpad / BK;
    // prologue: NS-1 stages in flight, the first one landed
#pragma unroll
    for (int st = 0; st < NS - 1; ++st)
        if (st < nk) {
#pragma unroll
            for (int p = 0; p < LPS; ++p) dma_piece(p, st, st);
        }
    if (nk >= NS - 1) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LPS) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    load_frags(0, 0, fr);

    if (DBG & 512) tr_loop0 = __builtin_amdgcn_s_memtime();
    // whole turns of the ring with compile-time slots, the remaining stages through a slot dispatch
    int kt = 0, slot = 0;
    static_assert(NS >= 3 && NS <= 5, "ring depth");
    for (; kt + 2 * (NS - 1) < nk; kt += NS) {
        stage_body(kt, std::integral_constant<int, 0>{}, yes{});
        stage_body(kt + 1, std::integral_constant<int, 1>{}, yes{});
        stage_body(kt + 2, std::integral_constant<int, 2>{}, yes{});
        if constexpr (NS >= 4) stage_body(kt + 3, std::integral_constant<int, 3>{}, yes{});
        if constexpr (NS >= 5) stage_body(kt + 4, std::integral_constant<int, 4>{}, yes{});
    }
    auto run_stage = [&](int k, int sl, auto do_dma) {
        if (sl == 0) stage_body(k, std::integral_constant<int, 0>{}, do_dma);
        else if (sl == 1) stage_body(k, std::integral_constant<int, 1>{}, do_dma);
        else if (NS == 3 || sl == 2) stage_body(k, std::integral_constant<int, 2>{}, do_dma);
        else if (NS == 4 || sl == 3) stage_body(k, std::integral_constant<int, (NS > 3 ? 3 : 2)>{}, do_dma);
        else stage_body(k, std::integral_constant<int, NS - 1>{}, do_dma);
    };
    for (; kt + (NS - 1) < nk; ++kt) {
        run_stage(kt, slot, yes{});
        if (++slot == NS) slot = 0;
    }
    for (; kt < nk; ++kt) {
        run_stage(kt, slot, no{});
        if (++slot == NS) slot = 0;
    }

    if (DBG & 512) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        tr_wait += tr_b - tr_a;
        tr_bar += tr_c - tr_b;
        tr_loop1 = __builtin_amdgcn_s_memtime();
    }
    if (!(DBG & 32)) igemm_epilogue<BN, WM, WN, !(DBG & 128), NW, BMT>(a, ph, pz, acc, smem, tid, lane, wave_m, wave_n, img, rem0, n0, bx);
    if ((DBG & 512) && a.trace && lane == 0) {
        // [workgroup][wave][8]: start, prologue end, loop end, kernel end (absolute), cycles in the data wait, in the
        // barrier, stages, (xcc_id << 8 | cu_id-ish hw id)
        const unsigned long long tr_end = __builtin_amdgcn_s_memtime();
        unsigned long long *o = a.trace + ((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NW + wave) * 8;
        const unsigned long long tr_rt1 = __builtin_amdgcn_s_memrealtime();
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        o[0] = tr_t0; o[1] = tr_loop0; o[2] = tr_loop1; o[3] = tr_end; o[4] = tr_wait; o[5] = tr_bar;
        o[6] = (unsigned long long)tr_stages | (tr_dma << 16) | ((tr_rt1 - tr_rt0) << 44);   // stages < 2^16, dma cycles < 2^28, 100 MHz ticks
        o[7] = ((unsigned long long)xcc << 32) | hwid;
    }
    if (DBG & 32) {   // bench only: keep every MFMA alive without an epilogue
        float keep = 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) keep += acc[i][j][r];
        if (keep == 123.456f) a.y[0] = 1.f;
    }
    __syncthreads();   // scratch aliases the ring: done with it before the next phase's DMA
    }  // phase loop
}

// ------------------------------------------------------------------------------------------------
// conv3x3_halo_bf16x3 -- 3x3 / stride 1 / pad 1 convolutions (the trunk and the skipper convs: 15 of the 19 launches of
// a step) with the ACTIVATION operand resident in LDS as a halo.
//
// Why (profiles/r03_conv_trace.md, tools/conv_trace.py): inside the pipeline conv_igemm_bf16x3<128> spends 3 % of its
// main loop waiting for data and 2 % in barriers, yet a stage takes ~1250 cycles against the 768 its 24 MFMAs need.
// What it is short of is global->LDS transfer: 32 KiB per stage and CU = 32 DMA instructions, half of them re-loading
// activation pixels the CU already holds -- tap (kh,kw) of a 128-pixel tile is the same pixels shifted by one.  With a
// quarter of the activation DMAs (wrong results, timing only) the same kernel runs 24 % faster in the pipeline.
//
// So: a workgroup's 128 output pixels are a block of TR = 4 rows x TC = 32 columns of one image (for W = 32: 128
// consecutive pixels, the implicit GEMM's tile); for one 32-channel slice it keeps the (TR+2) x (TC+2) = 204-pixel input
// halo in LDS (26 KiB) and runs all nine taps from it -- tap (kh,kw) of MFMA row tile i (= image row i of the block) is the
// 32 consecutive halo pixels starting at (i + kh) * (TC+2) + kw.  Two halo slots: slice s+1 is fetched while slice s
// computes.  Per stage a CU now moves 16 KiB of weights + 26/9 KiB of halo = 19 KiB instead of 32 (64-channel tile: 11 instead
// of 24, and its 76 KiB of LDS still let two workgroups share a CU).
// The weight operand keeps the DMA ring of conv_igemm_bf16x3 (NS slots, NS-1 stages in flight, same fragment layout).
//   * LDS image of a halo pixel = its 128-byte [hi x32 | lo x32] run with the 16-byte slots XOR-swizzled by (p >> 1) & 7,
//     p = halo pixel index: 32 consecutive pixels from ANY start read conflict-free with ds_read_b128 (rows 2k, 2k+1
//     share a swizzle and sit 128 B = 32 banks apart); the four fragments of a row (hi/lo x two k-blocks) are the base
//     address ^ 32 / ^ 64 / ^ 96.
//   * the nine taps of a slice are nine stage bodies (everything tap-dependent is an immediate or one SGPR add); the
//     weight ring slot is a run-time offset.
//   * halo pieces (1 KiB = 8 pixels each, NHW per wave and slice, the last ones may repeat a chunk so that every wave
//     issues the same number) ride behind the MFMAs of taps 0..6, ahead of the stage's weight pieces; counted
//     s_waitcnt: the pieces younger than the next stage's weights are known per tap at compile time.
//   * RAW (ConvArgs::raw_in): channel slices >= raw_from arrive as the producer's raw fp32 conv output and are normalised,
//     ReLU'd and split IN the halo slot by the wave that fetched them (every wave converts the chunks it loaded itself, so no
//     extra barrier: the stage barrier that publishes the halo publishes the converted halo).  The halo pieces are then
//     issued in taps 0..4 instead of 0..6; a piece issued in tap t has landed behind the counted wait of tap t+2, and its
//     chunk is converted under the MFMAs of tap t+3 -- the last one in tap 7, ahead of the barrier in tap 8 after which the
//     next slice's first fragments are read.  Chunks that two waves would fetch (so that all waves issue the same number
//     of pieces) go to a dummy chunk for one of them: a late duplicate must not overwrite a converted chunk.
template <int NHW, int B_CH, int Q, int QB, int NS, int HT = 7>
struct HaloSched {
    static constexpr int nh(int t) { return t < HT ? NHW / HT + (t < NHW % HT ? 1 : 0) : 0; }   // halo pieces issued in tap t's stage
    static constexpr int tap_of(int k) { int t = 0; while (hstart(t + 1) <= k) ++t; return t; }    // tap in which piece k is issued
    // MFMA slots ahead of the barrier that carry no DMA piece (RAW: where the conversion micro-steps ride)
    static constexpr bool dma_at(int t, int q) { for (int p = 0; p < lps(t); ++p) if (pos(t, p) == q) return true; return false; }
    static constexpr int nfree(int t, int q0) { int n = 0; for (int q = q0; q < QB; ++q) n += dma_at(t, q) ? 0 : 1; return n; }
    // RAW: which of a piece's eight conversion micro-steps [first, last) ride in free slot `slot` (piece 0 or 1 of the stage).  The
    // read (step 0) goes first and its first use two slots later: an in-order wave that waits for an LDS read does not issue
    // MFMAs either.
    static constexpr int conv_first(int nconv, int nslot, int slot, int piece) { return conv_range(nconv, nslot, slot, piece) / 16; }
    static constexpr int conv_last(int nconv, int nslot, int slot, int piece) { return conv_range(nconv, nslot, slot, piece) % 16; }
    static constexpr int conv_range(int nconv, int nslot, int slot, int piece) {   // first * 16 + last (no tables: everything folds)
        if (nconv == 1) {
            if (piece != 0) return 0;
            if (nslot >= 7)   // {0} - - {1,2} {3,4} {5,6} {7}
                return slot == 0 ? 0 * 16 + 1 : slot == 3 ? 1 * 16 + 3 : slot == 4 ? 3 * 16 + 5 : slot == 5 ? 5 * 16 + 7 : slot == 6 ? 7 * 16 + 8 : 0;
            return slot == 0 ? 0 * 16 + 1 : slot == 2 ? 1 * 16 + 4 : slot == 3 ? 4 * 16 + 6 : slot == 4 ? 6 * 16 + 8 : 0;   // {0} - {1,2,3} {4,5} {6,7}
        }
        if (nslot >= 8) {     // A{0} B{0} A{1,2} B{1,2} A{3,4} B{3,4} A{5,6,7} B{5,6,7}
            if (slot >= 8 || (slot & 1) != piece) return 0;
            return (slot >> 1) == 0 ? 0 * 16 + 1 : (slot >> 1) == 1 ? 1 * 16 + 3 : (slot >> 1) == 2 ? 3 * 16 + 5 : 5 * 16 + 8;
        }
        if (slot == 0) return 0 * 16 + 1;   // A{0} B{0} | - | A{1,2,3} B{1,2,3} A{4..7} B{4..7}
        if (slot < 2 || slot >= 6 || (slot & 1) != piece) return 0;
        return slot < 4 ? 1 * 16 + 4 : 4 * 16 + 8;
    }
    static constexpr int free_index(int t, int q0, int q) {
        if (q < q0 || q >= QB || dma_at(t, q)) return -1;
        int n = 0;
        for (int u = q0; u < q; ++u) n += dma_at(t, u) ? 0 : 1;
        return n;
    }
    static constexpr int hstart(int t) { int n = 0; for (int u = 0; u < t; ++u) n += nh(u); return n; }
    static constexpr int lps(int t) { return nh(t) + B_CH; }
    static constexpr int pos(int t, int p) { return (p + 1) * Q / lps(t) - 1; }              // piece p rides behind this MFMA
    static constexpr int issued(int t) { int n = 0; for (int p = 0; p < lps(t); ++p) n += pos(t, p) < QB ? 1 : 0; return n; }
    // pieces younger than the LAST weight piece of the stage that loaded stage k+1's weights (= stage k - (NS-2)):
    // everything issued in the NS-3 stages in between plus this stage's pieces ahead of the barrier
    static constexpr int nwait(int t) { int n = issued(t); for (int d = 1; d <= NS - 3; ++d) n += lps((t + 9 - d) % 9); return n; }
};

// BMT = 256: eight waves (two per SIMD, 204 registers each) on an 8 x 32-pixel block -- the two 128-pixel halves share the
// weight stage, so a CU moves 16 KiB of weights + 43/9 KiB of halo per TWO stages' worth of MFMAs.  For launches with at
// least one such tile per CU (the skippers at batch 8, the trunk from batch 16).
//
// CT = 1: ConvTranspose2d(k3, s2, p1, output_padding 1) as ONE launch.  Its four output parities (phases) are stride-1
// correlations with 1 / 2 / 2 / 4 taps reading in(i + ty, j + tx), ty, tx in {0, 1} (generator.hip, init_convT): nine
// (phase, tap) pairs per channel slice, all from the same (TR+1) x (TC+1) input halo.  The nine stage bodies are those
// pairs; each accumulates into its phase's own set of tiles (4 x 64 AGPRs on the 128-channel tile) and the epilogue
// runs once per phase.  The implicit-GEMM path ran the phases as separate workgroups of 4..16 stages each, most of
// their life in prologue and epilogue (profiles/r03_conv_trace_ring.md).
struct HaloCT {
    static constexpr int phase(int t) { return t == 0 ? 0 : t < 3 ? 1 : t < 5 ? 2 : 3; }
    static constexpr int tap(int t) { return t == 0 ? 0 : t < 3 ? t - 1 : t < 5 ? t - 3 : t - 5; }     // tap index inside its phase
    static constexpr int dy(int t) { return (t == 4 || t == 7 || t == 8) ? 1 : 0; }
    static constexpr int dx(int t) { return (t == 2 || t == 6 || t == 8) ? 1 : 0; }
};

template <int BN, int WM, int WN, int NS, int TC, int BMT = BM, int CT = 0, bool RAW = false>
__global__ __launch_bounds__(64 * (BMT / (32 * WM)) * (BN / (32 * WN))) void conv3x3_halo_bf16x3(const ConvArgs a)
{
    constexpr int NPH = CT ? 4 : 1, PADL = CT ? 0 : 1;   // accumulator sets; halo rows / columns above and left of the tile
    constexpr int WAVES_N = BN / (32 * WN), WAVES_M = BMT / (32 * WM), NW = WAVES_M * WAVES_N;
    static_assert((NW == 4 || NW == 8) && (NS == 3 || NS == 4) && BMT % TC == 0 && TC % 32 == 0, "layout");
    constexpr int TR = BMT / TC, HC = TC + 2 - (CT ? 1 : 0), HR = TR + 2 - (CT ? 1 : 0), HP = HR * HC;
    constexpr int NCH = (HP + 7) / 8;              // 1-KiB chunks (8 halo pixels) of one halo
    constexpr int NHW = (NCH + NW - 1) / NW;       // chunks a wave loads per slice
    constexpr int HALO = NCH * 8 * BK;             // floats per halo slot
    constexpr int B_CH = BN / 8 / NW, BSTAGE = BN * BK, BOFF = 2 * HALO;
    static_assert(B_CH >= 1, "a weight chunk per wave");
    constexpr int TILES = WM * WN, Q = 6 * TILES, QB = Q * 3 / 4, NL = 2 * (WM + WN);
    constexpr int RPM = (NL + (Q - QB) - 1) / (Q - QB);
    constexpr int HT = RAW ? 6 : 7;                // taps in whose stages the halo pieces of the next slice are issued
    using S = HaloSched<NHW, B_CH, Q, QB, NS, HT>;
    static_assert(S::hstart(9) == NHW && Q >= NHW / HT + 1 + B_CH, "halo pieces fit their taps, one piece per MFMA at most");
    static_assert(!RAW || (NW % 2 == 0 && S::tap_of(NHW - 1) + 3 <= 8), "RAW: chunk parity per wave, conversions end ahead of the barrier of tap 8");
    // RAW: one dummy chunk behind the ring (duplicate pieces land there), then the raw channels' (scale, shift) of this image
    constexpr int DUMMY = BOFF + NS * BSTAGE;      // float offset
    constexpr int SSOFF = DUMMY + 8 * BK;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    // measurement hook (lwg_conv_trace): four clock reads per wave, taken only when a record buffer is set
    unsigned long long tr_t0 = 0, tr_rt0 = 0, tr_loop0 = 0, tr_loop1 = 0;
    if (a.trace) {
        tr_t0 = __builtin_amdgcn_s_memtime();
        tr_rt0 = __builtin_amdgcn_s_memrealtime();
    }
    int bx = blockIdx.x;
    const int by = blockIdx.y;
    if (!a.natural_order && (gridDim.x & 7) == 0) bx = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // XCD bands
    const int n0 = by * BN;
    // tile bx = TR rows x TC columns of one image, tiles of an image in row-major order
    const int tiles_x = a.Wm / TC, tiles_img = tiles_x * (a.Hm / TR);
    const int img = __builtin_amdgcn_readfirstlane(bx / tiles_img), trem = bx - img * tiles_img;   // the divisions run on the VALU
    const int trow0 = __builtin_amdgcn_readfirstlane(trem / tiles_x);
    const int h0 = trow0 * TR, w0 = (trem - trow0 * tiles_x) * TC;
    const int rem0 = h0 * a.Wm + w0;                         // origin pixel (the epilogue's 2-D row mapping starts here)
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)smem;
    auto uniform_ptr = [](const void *p) {
        const unsigned long long v = (unsigned long long)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const char *>(((unsigned long long)hi << 32) | lo);
    };
    auto dma16 = [&](unsigned voff, const char *sbase, unsigned lds_byte) {
        asm volatile("s_mov_b32 m0, %2\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, %1"
                     :
                     : "v"(voff), "s"(sbase), "s"(lds_byte)
                     : "memory", "m0");
    };
    const ConvPhase ph = a.ph[0];
    const int lr = lane >> 3, ls = lane & 7;
    const float *xin = a.x + (size_t)img * a.H * a.W * a.ldx;
    const char *x_base = uniform_ptr(xin);
    const unsigned zoff_b = (unsigned)((const char *)a.zeros - (const char *)xin);

    // ---- halo chunks of this wave: chunk c = (wave + NW j) mod NCH covers halo pixels 8c .. 8c+7; lane -> (pixel 8c + lane>>3,
    //      16-byte slot lane&7), fetched from the swizzled source slot.  The byte offsets do not depend on the slice.
    unsigned hoff[NHW];
    unsigned inside = 0;   // RAW: bit j = this lane's pixel of piece j is inside the image (padding stays zero: nothing to normalise)
#pragma unroll
    for (int j = 0; j < NHW; ++j) {
        int c = wave + NW * j;
        if (c >= NCH) c -= NCH;
        const int p = c * 8 + lr;
        const int hr = p / HC, hc = p - hr * HC;
        const int gh = h0 - PADL + hr, gw = w0 - PADL + hc;
        const bool ok = p < HP && (unsigned)gh < (unsigned)a.H && (unsigned)gw < (unsigned)a.W;
        hoff[j] = ok ? (unsigned)(((gh * a.W + gw) * a.ldx + (ls ^ ((p >> 1) & 7)) * 4) * 4) : zoff_b;
        if (RAW && ok && wave + NW * j < NCH) inside |= 1u << j;   // (a duplicate piece goes to the dummy chunk and is nobody's to convert)
    }
    unsigned wvoff[NPH][B_CH];   // lane offsets into the phase's [Cout][ntaps * Cin] matrix
#pragma unroll
    for (int pp = 0; pp < NPH; ++pp)
#pragma unroll
        for (int j = 0; j < B_CH; ++j) {
            const int row = (wave * B_CH + j) * 8 + lr;
            wvoff[pp][j] = (unsigned)(((n0 + row) * a.ph[pp].Kpad + (ls ^ ((row >> 1) & 7)) * 4) * 4);
        }
    const char *w_phase[NPH];
#pragma unroll
    for (int pp = 0; pp < NPH; ++pp) w_phase[pp] = uniform_ptr(a.w_split + a.ph[pp].w_off);
    const unsigned cin4 = (unsigned)a.Cin * 4u;
    const int nslices = a.Cin / BK;
    const unsigned wave_b = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((BOFF + wave * B_CH * 8 * BK) * 4));

    // halo piece j of channel slice `sl` into halo slot hs
    auto halo_piece = [&](int j, int sl, int hs) {
        int c = wave + NW * j;
        unsigned dst = (unsigned)((hs * HALO + c * 8 * BK) * 4);
        if (c >= NCH) {
            c -= NCH;
            dst = RAW ? (unsigned)(DUMMY * 4) : (unsigned)((hs * HALO + c * 8 * BK) * 4);
        }
        dma16(hoff[j], x_base + __builtin_amdgcn_readfirstlane((unsigned)sl * (BK * 4u)), __builtin_amdgcn_readfirstlane(lds_base + dst));
    };
    // RAW: piece j of slice `sl` (landed in halo slot hs) from raw fp32 to the normalised split-bf16 form, in place.  Lane
    // (lr, ls) holds the 16 bytes it fetched: raw channels 4g .. 4g+3 of the slice, g = ls ^ swz (the fetch swizzled the SOURCE
    // slot); as a split run those four values are 8 bytes of hi in 16-byte slot g/2 and 8 of lo in slot 4 + g/2, both
    // stored at their swizzled positions.  One ds_read_b128 per lane, then the writes: the wave reads its whole chunk before
    // it writes any of it (LDS operations of one wave execute in order).
    [[maybe_unused]] float4 ssa = {0.f, 0.f, 0.f, 0.f}, ssb = {0.f, 0.f, 0.f, 0.f};   // (scale, shift) of this lane's four channels
    [[maybe_unused]] auto load_ss = [&](int sl) {
        // chunk c = wave + NW j and NW is even: every chunk of this wave has the parity of `wave`, so swz and g are per-lane constants
        const int g = ls ^ ((((wave & 1) << 2) + (lr >> 1)) & 7);
        const float *q = smem + SSOFF + (sl * BK - a.raw_from + 4 * g) * 2;
        ssa = *reinterpret_cast<const float4 *>(q);
        ssb = *reinterpret_cast<const float4 *>(q + 4);
    };
    // per-lane constants of the conversion (see load_ss): byte offsets inside a chunk row of the slot read and of the two slots written
    [[maybe_unused]] const int cv_swz = (((wave & 1) << 2) + (lr >> 1)) & 7, cv_g = ls ^ cv_swz;
    [[maybe_unused]] const int cv_rd = lr * 128 + ls * 16;
    [[maybe_unused]] const int cv_hi = lr * 128 + (((cv_g >> 1) ^ cv_swz) << 4) + ((cv_g & 1) << 3);
    [[maybe_unused]] const int cv_lo = lr * 128 + (((4 + (cv_g >> 1)) ^ cv_swz) << 4) + ((cv_g & 1) << 3);
    [[maybe_unused]] const int cv_dummy = DUMMY * 4 + lane * 16;     // where the writes of lanes with nothing to convert go
    struct Conv { float4 v, y, hf; bf16x4_t h, l; int ahi, alo; };
    [[maybe_unused]] Conv cvA, cvB;   // (two named objects, not an array: an index the front end cannot fold sends an array to scratch)
    // Eight micro-steps per piece, a few instructions each, so that they can ride one by one behind consecutive MFMAs:
    // 0 read the raw 16 bytes; 1, 2 normalise + ReLU; 3 hi terms; 4 hi terms back as floats; 5 lo terms; 6 where to write
    // (`live` = the lane's pixel is inside the image and the slice is a raw one, else the dummy chunk: no branch in the MFMA
    // stream); 7 write
    [[maybe_unused]] auto convert_step = [&](int j, Conv &c, int hs, int step, bool live) {
        const int chunk = (hs * HALO + (wave + NW * j) * 8 * BK) * 4;
        char *base = reinterpret_cast<char *>(smem);
        if (step == 0) {
            c.v = *reinterpret_cast<const float4 *>(base + chunk + cv_rd);
        } else if (step == 1) {
            c.y.x = fmaxf(fmaf(c.v.x, ssa.x, ssa.y), 0.f);
            c.y.y = fmaxf(fmaf(c.v.y, ssa.z, ssa.w), 0.f);
        } else if (step == 2) {
            c.y.z = fmaxf(fmaf(c.v.z, ssb.x, ssb.y), 0.f);
            c.y.w = fmaxf(fmaf(c.v.w, ssb.z, ssb.w), 0.f);
        } else if (step == 3) {
            c.h[0] = (__bf16)c.y.x; c.h[1] = (__bf16)c.y.y; c.h[2] = (__bf16)c.y.z; c.h[3] = (__bf16)c.y.w;
        } else if (step == 4) {
            c.hf = make_float4((float)c.h[0], (float)c.h[1], (float)c.h[2], (float)c.h[3]);
        } else if (step == 5) {
            c.l[0] = (__bf16)(c.y.x - c.hf.x); c.l[1] = (__bf16)(c.y.y - c.hf.y);
            c.l[2] = (__bf16)(c.y.z - c.hf.z); c.l[3] = (__bf16)(c.y.w - c.hf.w);
        } else if (step == 6) {
            c.ahi = live ? chunk + cv_hi : cv_dummy;
            c.alo = live ? chunk + cv_lo : cv_dummy + 8;
        } else {
            *reinterpret_cast<bf16x4_t *>(base + c.ahi) = c.h;
            *reinterpret_cast<bf16x4_t *>(base + c.alo) = c.l;
        }
    };
    // weight piece jb of reduction stage (tap it, slice is) into ring slot `slot`
    auto weight_piece = [&](int jb, int it, int is, int slot) {
        const int pp = CT ? HaloCT::phase(it) : 0, ti = CT ? HaloCT::tap(it) : it;
        dma16(wvoff[pp][jb], w_phase[pp] + __builtin_amdgcn_readfirstlane((unsigned)ti * cin4 + (unsigned)is * (BK * 4u)),
              __builtin_amdgcn_readfirstlane(wave_b + (unsigned)((slot * BSTAGE + jb * 8 * BK) * 4)));
    };

    f32x16 acc[NPH][WM][WN];
#pragma unroll
    for (int pp = 0; pp < NPH; ++pp)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[pp][i][j][r] = 0.f;

    // ---- fragment addressing
    const int frow = lane & 31, half = lane >> 5, fsw = (frow >> 1) & 7;
    int pl[WM];            // halo pixel of this lane's row of MFMA row tile i at tap (0,0)
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int m = wave_m * 32 * WM + i * 32;
        pl[i] = (m / TC) * HC + (m % TC) + frow;
    }
    // byte address (within smem) of the hi fragment of k-block 0; ^32: hi of k-block 1, ^64 / ^96: the lo fragments
    auto a_addr = [&](int i, int tap, int hs) {
        const int p = pl[i] + (CT ? HaloCT::dy(tap) * HC + HaloCT::dx(tap) : (tap / 3) * HC + (tap % 3));
        return hs * (HALO * 4) + p * 128 + ((half ^ ((p >> 1) & 7)) << 4);
    };
    const int b_row = (wave_n * 32 * WN + frow) * BK;
    int fcol[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) fcol[c] = (((2 * c + half) ^ fsw) * 4);
    struct Frags { float4 ah[WM], al[WM], bh[WN], bl[WN]; };
    const char *smem_b = reinterpret_cast<const char *>(smem);
    // fragment register idx of k-block kb, in the order the products consume them: lo(A), hi(B), hi(A), lo(B)
    auto load_one = [&](const int (&aa)[WM], int bslot, int kb, Frags &f, int idx) {
        const float *Bs = smem + BOFF + bslot * BSTAGE + b_row;
        if (idx < WM) f.al[idx] = *reinterpret_cast<const float4 *>(smem_b + (aa[idx] ^ (64 + 32 * kb)));
        else if (idx < WM + WN) f.bh[idx - WM] = *reinterpret_cast<const float4 *>(Bs + (idx - WM) * 32 * BK + fcol[kb]);
        else if (idx < 2 * WM + WN) f.ah[idx - WM - WN] = *reinterpret_cast<const float4 *>(smem_b + (aa[idx - WM - WN] ^ (32 * kb)));
        else f.bl[idx - 2 * WM - WN] = *reinterpret_cast<const float4 *>(Bs + (idx - 2 * WM - WN) * 32 * BK + fcol[2 + kb]);
    };
    Frags fr;          // k-block-0 fragments of the stage about to run
    int cur_a[WM];     // A fragment base addresses of the stage about to run

    // One stage = tap T of channel slice `sl`, weights in ring slot `slot`.  ISSUE: the stage also issues the halo pieces
    // due in tap T (for slice sl+1, into the other halo slot) and the weights of stage k+NS-1; false for the last NS-1
    // stages of the kernel.
    auto stage_body = [&](auto tap_c, auto issue_c, int sl, int slot) {
        constexpr int T = decltype(tap_c)::value;
        constexpr bool ISSUE = decltype(issue_c)::value;
        constexpr int TN = (T + 1) % 9;                       // next stage's tap
        constexpr int IT = (T + NS - 1) % 9;                  // tap of the stage whose weights are issued here
        const int slot1 = slot + 1 == NS ? 0 : slot + 1;      // next stage's weights
        const int slot2 = slot == 0 ? NS - 1 : slot - 1;      // ring slot being refilled (read last by stage k-1)
        const int hs = sl & 1, hs_next = T == 8 ? hs ^ 1 : hs;
        const int is = sl + (T + NS - 1 >= 9 ? 1 : 0);        // slice of the stage whose weights are issued here
        const int sl_halo = sl + 1 < nslices ? sl + 1 : 0;    // the last slice fetches a halo nobody reads
        Frags f1, nx;
        int nxt_a[WM];
        // RAW: the chunks of slice sl+1 whose piece was issued three taps ago have landed (counted wait of tap T-1): normalise them
        // under this tap's MFMAs, ahead of its barrier
        // (a slice that arrives in the split format -- the skip half of a decoder level's input -- runs the same instructions
        // with every lane's result sent to the dummy chunk: separate stage bodies for it cost the compiler its register allocation)
        [[maybe_unused]] const bool raw_next = RAW && sl + 1 < nslices && (sl + 1) * BK >= a.raw_from;
        [[maybe_unused]] const unsigned live_bits = raw_next ? inside : 0u;
        if constexpr (RAW) {
            if (T == S::tap_of(0) + 3) load_ss(sl + 1 < nslices ? sl + 1 : 0);
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int kb = q / (3 * TILES), r = q % (3 * TILES);
            const int t = r / TILES, i = (r / WN) % WM, j = r % WN;   // cross terms first, hi*hi last
            if (q == QB) {
                __builtin_amdgcn_sched_barrier(0);
                if (ISSUE) {
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(S::nwait(T)) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
#pragma unroll
                for (int ii = 0; ii < WM; ++ii) nxt_a[ii] = a_addr(ii, TN, hs_next);
            }
            const Frags &f = kb == 0 ? fr : f1;
            const float4 a4 = t == 0 ? f.al[i] : f.ah[i];
            const float4 b4 = t == 1 ? f.bl[j] : f.bh[j];
            constexpr int PP = CT ? HaloCT::phase(T) : 0;
            acc[PP][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a4),
                                                                    __builtin_bit_cast(bf16x8_t, b4), acc[PP][i][j], 0, 0, 0);
            if (q < NL) {
                load_one(cur_a, slot, 1, f1, q);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            } else if (q >= QB && RPM * (q - QB) < NL) {
#pragma unroll
                for (int r2 = 0; r2 < RPM; ++r2)
                    if (RPM * (q - QB) + r2 < NL) load_one(nxt_a, slot1, 0, nx, RPM * (q - QB) + r2);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, RPM, 0);
            }
            if constexpr (RAW) {
                // the conversion of the pieces issued three taps ago, five steps per piece, one step behind each of the MFMAs ahead
                // of the barrier (four steps where the stage has fewer than ten MFMAs there: the 64-channel tile)
                constexpr int NCONV = T >= 3 ? S::nh(T - 3) : 0, K0 = T >= 3 ? S::hstart(T - 3) : 0;
                // ... behind the MFMAs that follow the pinned fragment reads of k-block 1 (q >= NL) where ten or more are left ahead
                // of the barrier, else from the first one (the 64-channel tile)
                constexpr int Q0 = QB - NL >= 10 ? NL : 0;
                constexpr int NF = S::nfree(T, Q0);   // free MFMA slots
                static_assert(NCONV <= 2 && NF >= (NCONV == 2 ? 6 : 5), "conversion steps fit ahead of the barrier");
                const int fi = S::free_index(T, Q0, q);
                if (NCONV > 0 && fi >= 0) {
                    const int a0 = S::conv_first(NCONV, NF, fi, 0), a1 = S::conv_last(NCONV, NF, fi, 0);
                    const int b0 = S::conv_first(NCONV, NF, fi, 1), b1 = S::conv_last(NCONV, NF, fi, 1);
                    if (a1 > a0 || b1 > b0) {
                        __builtin_amdgcn_sched_barrier(0);   // the micro-steps stay behind THIS MFMA: a handful of instructions per gap
#pragma unroll
                        for (int m = 0; m < 8; ++m)
                            if (m >= a0 && m < a1) convert_step(K0, cvA, hs ^ 1, m, (live_bits >> K0) & 1u);
#pragma unroll
                        for (int m = 0; m < 8; ++m)
                            if (m >= b0 && m < b1) convert_step(K0 + 1, cvB, hs ^ 1, m, (live_bits >> (K0 + 1)) & 1u);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (ISSUE) {
#pragma unroll
                for (int p = 0; p < S::lps(T); ++p)
                    if (q == S::pos(T, p)) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (p < S::nh(T)) halo_piece(S::hstart(T) + p, sl_halo, hs ^ 1);
                        else weight_piece(p - S::nh(T), IT, is, slot2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
            }
        }
        fr = nx;
#pragma unroll
        for (int ii = 0; ii < WM; ++ii) cur_a[ii] = nxt_a[ii];
    };

    // ---- prologue: halo of slice 0, weights of stages 0 .. NS-2 (taps 0 .. NS-2 of slice 0); the first stage landed
    if constexpr (RAW) {
        // this image's (scale, shift) of the raw channels into LDS: plain loads, drained before the first DMA is issued (the
        // main loop counts its outstanding memory operations)
        const float2 *ss = a.in_ss + (size_t)img * a.in_ss_ld;
        const int nraw = a.Cin - a.raw_from;
        for (int i2 = tid; i2 < nraw; i2 += NW * 64) reinterpret_cast<float2 *>(smem + SSOFF)[i2] = ss[i2];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int j = 0; j < NHW; ++j) halo_piece(j, 0, 0);
#pragma unroll
    for (int st = 0; st < NS - 1; ++st)
#pragma unroll
        for (int jb = 0; jb < B_CH; ++jb) weight_piece(jb, st, 0, st);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * B_CH) : "memory");
    if constexpr (RAW) {
        if (a.raw_from == 0) {                 // slice 0 is raw: convert it before anyone reads it
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();      // the (scale, shift) table is complete
            asm volatile("" ::: "memory");
            load_ss(0);
#pragma unroll
            for (int j = 0; j < NHW; ++j)
#pragma unroll
                for (int step = 0; step < 8; ++step) convert_step(j, cvA, 0, step, (inside >> j) & 1u);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int ii = 0; ii < WM; ++ii) cur_a[ii] = a_addr(ii, 0, 0);
#pragma unroll
    for (int idx = 0; idx < NL; ++idx) load_one(cur_a, 0, 0, fr, idx);

    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    if (a.trace) tr_loop0 = __builtin_amdgcn_s_memtime();
    int slot = 0;
    auto bump = [&]() { slot = slot + 1 == NS ? 0 : slot + 1; };
#define LWG_HALO_STAGE(T, ISS) stage_body(std::integral_constant<int, T>{}, ISS{}, sl, slot); bump();
    for (int sl = 0; sl + 1 < nslices; ++sl) {
        LWG_HALO_STAGE(0, yes) LWG_HALO_STAGE(1, yes) LWG_HALO_STAGE(2, yes) LWG_HALO_STAGE(3, yes) LWG_HALO_STAGE(4, yes)
        LWG_HALO_STAGE(5, yes) LWG_HALO_STAGE(6, yes) LWG_HALO_STAGE(7, yes) LWG_HALO_STAGE(8, yes)
    }
    {
        const int sl = nslices - 1;   // last slice: its final NS-1 stages have nothing left to fetch
        LWG_HALO_STAGE(0, yes) LWG_HALO_STAGE(1, yes) LWG_HALO_STAGE(2, yes) LWG_HALO_STAGE(3, yes) LWG_HALO_STAGE(4, yes)
        LWG_HALO_STAGE(5, yes)
        if constexpr (NS == 3) { LWG_HALO_STAGE(6, yes) } else { LWG_HALO_STAGE(6, no) }
        LWG_HALO_STAGE(7, no) LWG_HALO_STAGE(8, no)
    }
#undef LWG_HALO_STAGE

    if (a.trace) tr_loop1 = __builtin_amdgcn_s_memtime();
    __syncthreads();   // the epilogue's staging area aliases the halo
    // statistics: one (mean, M2) per 4 x 32 block, numbered row-major within the image whatever the tile height, so that
    // the combination order in in_finalize (and with it every bit) does not depend on which variant ran
    const int part0 = img * (tiles_x * (a.Hm / 4)) + (h0 / 4) * tiles_x + (w0 / TC);
#pragma unroll
    for (int pp = 0; pp < NPH; ++pp) {
        if (pp) __syncthreads();   // the previous phase's statistics scratch has been read
        igemm_epilogue<BN, WM, WN, true, NW, BMT, TC>(a, a.ph[pp], pp, acc[pp], smem, tid, lane, wave_m, wave_n, img, rem0, n0, part0, tiles_x);
    }
    if (a.trace && lane == 0) {   // record layout of conv_igemm_bf16x3's traced twin; no per-stage wait accounting here
        const unsigned long long tr_end = __builtin_amdgcn_s_memtime(), tr_rt1 = __builtin_amdgcn_s_memrealtime();
        unsigned long long *o = a.trace + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NW + wave) * 8;
        o[0] = tr_t0; o[1] = tr_loop0; o[2] = tr_loop1; o[3] = tr_end; o[4] = 0; o[5] = 0;
        o[6] = (unsigned long long)(9 * nslices) | ((tr_rt1 - tr_rt0) << 44);
        o[7] = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// grid (C/16, N), 256 threads = 16 tile slices x 16 channels (64-byte coalesced rows of float2 partials)
constexpr int FIN_CH = 16, FIN_SL = 16;
__global__ __launch_bounds__(256) void in_finalize_kernel(const float2 *__restrict__ partials, int nphase, int mtiles,
                                                          int tiles_per_img, int C, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float eps,
                                                          float2 *__restrict__ scale_shift)
{
    __shared__ double sh[3][FIN_SL][FIN_CH];
    const int cl = threadIdx.x & (FIN_CH - 1);
    const int c = blockIdx.x * FIN_CH + cl;
    const int slice = threadIdx.x / FIN_CH;
    const int n = blockIdx.y;
    double sm = 0., sq = 0., m2 = 0.;
    if (c < C) {
        for (int p = 0; p < nphase; ++p) {
            const float2 *base = partials + ((size_t)p * mtiles + (size_t)n * tiles_per_img) * C + c;
#pragma unroll 4
            for (int t = slice; t < tiles_per_img; t += FIN_SL) {
                const float2 v = base[(size_t)t * C];
                sm += v.x;
                sq += (double)v.x * v.x;
                m2 += v.y;
            }
        }
    }
    sh[0][slice][cl] = sm;
    sh[1][slice][cl] = sq;
    sh[2][slice][cl] = m2;
    __syncthreads();
    if (slice == 0 && c < C) {
        sm = sq = m2 = 0.;
#pragma unroll
        for (int k = 0; k < FIN_SL; ++k) {
            sm += sh[0][k][cl];
            sq += sh[1][k][cl];
            m2 += sh[2][k][cl];
        }
        const double cnt = (double)nphase * tiles_per_img;
        const double mean = sm / cnt;
        // Chan et al.: M2_total = sum M2_i + n_i * sum (mean_i - mean)^2, every tile holds BM samples
        double between = sq - cnt * mean * mean;
        if (between < 0.) between = 0.;
        const double var = (m2 + BM * between) / (cnt * BM);  // biased, as InstanceNorm2d uses
        const double inv = 1.0 / sqrt(var + (double)eps);
        const float sc = (float)(gamma[c] * inv);
        scale_shift[(size_t)n * C + c] = make_float2(sc, (float)(beta[c] - mean * gamma[c] * inv));
    }
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void fma4(float4 &acc, const float4 v, float w)
{
    acc.x += v.x * w; acc.y += v.y * w; acc.z += v.z * w; acc.w += v.w * w;
}

__global__ __launch_bounds__(256) void apply_kernel(const ApplyArgs a)
{
    const int c4n = a.C >> 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)a.N * a.H * a.W * c4n;
    if (i >= total) return;
    const long pix = i / c4n;
    const int c = (int)(i - pix * c4n) * 4;
    const int hw = a.H * a.W;
    const int n = (int)(pix / hw);
    const int pn = (int)(pix - (long)n * hw);

    const float4 v = ld4(a.raw + pix * a.C + c);
    const float4 s01 = ld4(reinterpret_cast<const float *>(a.scale_shift + (size_t)n * a.C + c));
    const float4 s23 = ld4(reinterpret_cast<const float *>(a.scale_shift + (size_t)n * a.C + c + 2));
    float4 y;
    y.x = v.x * s01.x + s01.y;
    y.y = v.y * s01.z + s01.w;
    y.z = v.z * s23.x + s23.y;
    y.w = v.w * s23.z + s23.w;
    if (a.relu) {
        y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f);
    }
    // split-bf16 format: channels c..c+3 of a 32-channel group are 8 bytes of hi at 2*(c&31) and 8 of lo 64 B further
    const int soff = (c >> 5) * 32 + ((c & 31) >> 1);   // float (4-byte) units
    if (a.res) {
        float4 r;
        if (a.split) {
            const bf16x4_t h = *reinterpret_cast<const bf16x4_t *>(a.res + pix * a.ld_res + soff);
            const bf16x4_t l = *reinterpret_cast<const bf16x4_t *>(a.res + pix * a.ld_res + soff + 16);
            r = make_float4((float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2],
                            (float)h[3] + (float)l[3]);
        } else {
            r = ld4(a.res + pix * a.ld_res + c);
        }
        y.x += r.x; y.y += r.y; y.z += r.z; y.w += r.w;
    }
    for (int k = 0; k < a.nwarp; ++k) {
        const float2 g = *reinterpret_cast<const float2 *>(a.warp_T[k] + pix * 2);
        const GridTaps t = grid_taps(g.x, g.y, a.W, a.H, a.align_corners);
        const float *src = a.warp_src[k] + (size_t)(a.warp_n[k] > 1 ? n : 0) * hw * a.C + c;
        float4 wsum = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t.vnw) fma4(wsum, ld4(src + (size_t)(t.y0 * a.W + t.x0) * a.C), t.wnw);
        if (t.vne) fma4(wsum, ld4(src + (size_t)(t.y0 * a.W + t.x0 + 1) * a.C), t.wne);
        if (t.vsw) fma4(wsum, ld4(src + (size_t)((t.y0 + 1) * a.W + t.x0) * a.C), t.wsw);
        if (t.vse) fma4(wsum, ld4(src + (size_t)((t.y0 + 1) * a.W + t.x0 + 1) * a.C), t.wse);
        y.x += wsum.x; y.y += wsum.y; y.z += wsum.z; y.w += wsum.w;
    }
    (void)pn;
    if (a.split) {
        bf16x4_t h, l;
        h[0] = (__bf16)y.x; h[1] = (__bf16)y.y; h[2] = (__bf16)y.z; h[3] = (__bf16)y.w;
        l[0] = (__bf16)(y.x - (float)h[0]); l[1] = (__bf16)(y.y - (float)h[1]);
        l[2] = (__bf16)(y.z - (float)h[2]); l[3] = (__bf16)(y.w - (float)h[3]);
        *reinterpret_cast<bf16x4_t *>(a.dst + pix * a.ld_dst + soff) = h;
        *reinterpret_cast<bf16x4_t *>(a.dst + pix * a.ld_dst + soff + 16) = l;
    } else {
        *reinterpret_cast<float4 *>(a.dst + pix * a.ld_dst + c) = y;
    }
}

// The same pass with EIGHT channels per thread, for the split-bf16 format (C % 32 == 0): per lane two 16-byte raw loads, 16 + 16
// bytes of residual, two 16-byte loads per Liquid-Warping-Block tap, and the hi / lo terms of the eight results leave as two
// 16-byte stores (apply_kernel: four 8-byte ones per eight channels); the flow sample and its bilinear taps are computed once
// per eight channels instead of once per four.  Every value goes through exactly apply_kernel's operations in apply_kernel's
// order: bit-identical output (tests/test_gpu_generator.py), A/B switch LWG_APPLY8=0.
__global__ __launch_bounds__(256) void apply8_kernel(const ApplyArgs a)
{
    const int c8n = a.C >> 3;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)a.N * a.H * a.W * c8n;
    if (i >= total) return;
    const long pix = i / c8n;
    const int c = (int)(i - pix * c8n) * 8;
    const int hw = a.H * a.W;
    const int n = (int)(pix / hw);

    const float4 v0 = ld4(a.raw + pix * a.C + c), v1 = ld4(a.raw + pix * a.C + c + 4);
    const float *ssp = reinterpret_cast<const float *>(a.scale_shift + (size_t)n * a.C + c);
    const float4 s0 = ld4(ssp), s1 = ld4(ssp + 4), s2 = ld4(ssp + 8), s3 = ld4(ssp + 12);
    float y[8];
    y[0] = v0.x * s0.x + s0.y; y[1] = v0.y * s0.z + s0.w; y[2] = v0.z * s1.x + s1.y; y[3] = v0.w * s1.z + s1.w;
    y[4] = v1.x * s2.x + s2.y; y[5] = v1.y * s2.z + s2.w; y[6] = v1.z * s3.x + s3.y; y[7] = v1.w * s3.z + s3.w;
    if (a.relu) {
#pragma unroll
        for (int k = 0; k < 8; ++k) y[k] = fmaxf(y[k], 0.f);
    }
    // split-bf16 format: channels c..c+7 of a 32-channel group are 16 bytes of hi at 2*(c&31) and 16 of lo 64 B further
    const int soff = (c >> 5) * 32 + ((c & 31) >> 1);   // float (4-byte) units
    if (a.res) {
        const bf16x8_t h = *reinterpret_cast<const bf16x8_t *>(a.res + pix * a.ld_res + soff);
        const bf16x8_t l = *reinterpret_cast<const bf16x8_t *>(a.res + pix * a.ld_res + soff + 16);
#pragma unroll
        for (int k = 0; k < 8; ++k) y[k] += (float)h[k] + (float)l[k];
    }
    for (int k = 0; k < a.nwarp; ++k) {
        const float2 g = *reinterpret_cast<const float2 *>(a.warp_T[k] + pix * 2);
        const GridTaps t = grid_taps(g.x, g.y, a.W, a.H, a.align_corners);
        const float *src = a.warp_src[k] + (size_t)(a.warp_n[k] > 1 ? n : 0) * hw * a.C + c;
        float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0;
        if (t.vnw) { const float *q = src + (size_t)(t.y0 * a.W + t.x0) * a.C; fma4(w0, ld4(q), t.wnw); fma4(w1, ld4(q + 4), t.wnw); }
        if (t.vne) { const float *q = src + (size_t)(t.y0 * a.W + t.x0 + 1) * a.C; fma4(w0, ld4(q), t.wne); fma4(w1, ld4(q + 4), t.wne); }
        if (t.vsw) { const float *q = src + (size_t)((t.y0 + 1) * a.W + t.x0) * a.C; fma4(w0, ld4(q), t.wsw); fma4(w1, ld4(q + 4), t.wsw); }
        if (t.vse) { const float *q = src + (size_t)((t.y0 + 1) * a.W + t.x0 + 1) * a.C; fma4(w0, ld4(q), t.wse); fma4(w1, ld4(q + 4), t.wse); }
        y[0] += w0.x; y[1] += w0.y; y[2] += w0.z; y[3] += w0.w; y[4] += w1.x; y[5] += w1.y; y[6] += w1.z; y[7] += w1.w;
    }
    bf16x8_t h, l;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        h[k] = (__bf16)y[k];
        l[k] = (__bf16)(y[k] - (float)h[k]);
    }
    *reinterpret_cast<bf16x8_t *>(a.dst + pix * a.ld_dst + soff) = h;
    *reinterpret_cast<bf16x8_t *>(a.dst + pix * a.ld_dst + soff + 16) = l;
}

// one thread per 32-value group, in place: [hi x32 | lo x32] bf16 -> 32 floats
__global__ __launch_bounds__(256) void unsplit_kernel(float *buf, size_t ngroups)
{
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    float4 raw[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) raw[i] = ld4(buf + g * 32 + i * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // raw[i] = hi of values 8i..8i+7, raw[4+i] = lo
        const bf16x8_t h = __builtin_bit_cast(bf16x8_t, raw[i]);
        const bf16x8_t l = __builtin_bit_cast(bf16x8_t, raw[4 + i]);
        float4 o0 = make_float4((float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2],
                                (float)h[3] + (float)l[3]);
        float4 o1 = make_float4((float)h[4] + (float)l[4], (float)h[5] + (float)l[5], (float)h[6] + (float)l[6],
                                (float)h[7] + (float)l[7]);
        *reinterpret_cast<float4 *>(buf + g * 32 + i * 8) = o0;
        *reinterpret_cast<float4 *>(buf + g * 32 + i * 8 + 4) = o1;
    }
}

// ------------------------------------------------------------------------------------------------
// heads: thread = one image column x four consecutive rows of a 32x32 tile.  Channels are walked in chunks of 8 through
// an LDS halo (38x38 pixels, 48-byte pitch: conflict-free b128 reads for lanes on adjacent columns).  For a fixed
// (kx, 4-channel group) a thread fetches the ten halo rows its four pixels see once and reuses them over the seven
// ky taps: one b128 activation read per 45 FMAs.  The chunk's weights sit in LDS too and are read as wave-wide
// broadcasts (scalar loads of the weights, tried first, left the two resident waves per SIMD waiting on s_waitcnt:
// the SGPR file cannot hold a prefetched (kx, group) block).  The compiler emits v_pk_fma_f32 for the output pairs.
constexpr int HT = 32, HH = HT + 6;         // tile edge, halo edge
constexpr int HCH = 8;                      // channels per chunk.  16 halves the fabric re-reads of the input (634 -> ~320 MB:
                                            // a 128-B line serves 4 chunk passes) but leaves one workgroup per CU: 301 vs 207 us
constexpr int HPITCH = HCH + 4;             // floats per halo pixel
constexpr int HEADS_W = 49 * HCH * 4;       // weights of one chunk: [tap][channel][output]
constexpr int HEADS_LDS = (HH * HH * HPITCH + HEADS_W) * (int)sizeof(float);

__global__ __launch_bounds__(256) void heads_kernel(const HeadsArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float halo[];
    const int tid = threadIdx.x;
    const int tx = tid & 31, tg = tid >> 5;
    const int n = blockIdx.z;
    const int y0 = blockIdx.y * HT, x0 = blockIdx.x * HT;
    const float *xin = a.x + (size_t)n * a.H * a.W * 64;
    const float2 *ss = a.scale_shift + (size_t)n * 64;

    float acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[r][o] = 0.f;

    for (int chunk = 0; chunk < 64 / HCH; ++chunk) {
        __syncthreads();
        for (int i = tid; i < HH * HH * (HCH / 4); i += 256) {
            const int pix = i / (HCH / 4), half = i % (HCH / 4);
            const int hy = pix / HH, hx = pix - hy * HH;
            const int gy = y0 - 3 + hy, gx = x0 - 3 + hx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W) {
                const int c = chunk * HCH + half * 4;
                v = ld4(xin + ((size_t)gy * a.W + gx) * 64 + c);
                const float4 s01 = ld4(reinterpret_cast<const float *>(ss + c));
                const float4 s23 = ld4(reinterpret_cast<const float *>(ss + c + 2));
                v.x = fmaxf(v.x * s01.x + s01.y, 0.f);
                v.y = fmaxf(v.y * s01.z + s01.w, 0.f);
                v.z = fmaxf(v.z * s23.x + s23.y, 0.f);
                v.w = fmaxf(v.w * s23.z + s23.w, 0.f);
            }
            *reinterpret_cast<float4 *>(halo + pix * HPITCH + half * 4) = v;
        }
        float *wl = halo + HH * HH * HPITCH;
        for (int i = tid; i < 49 * HCH; i += 256) {   // (tap, channel) -> 4 outputs
            const int t = i / HCH, c = i % HCH;
            *reinterpret_cast<float4 *>(wl + i * 4) = ld4(a.wh + ((size_t)t * 64 + chunk * HCH + c) * 4);
        }
        __syncthreads();
        for (int kx = 0; kx < 7; ++kx)
            for (int q = 0; q < HCH / 4; ++q) {
                // all LDS reads of the block up front (10 activation rows + 28 weight quads, ~150 VGPRs): one
                // latency per 448 FMAs instead of one per 16
                float4 hv[10], wq[7][4];
#pragma unroll
                for (int hr = 0; hr < 10; ++hr) hv[hr] = ld4(halo + ((tg * 4 + hr) * HH + tx + kx) * HPITCH + q * 4);
#pragma unroll
                for (int ky = 0; ky < 7; ++ky) {
                    const float *wp = wl + ((ky * 7 + kx) * HCH + q * 4) * 4;   // same address in every lane: broadcast reads
#pragma unroll
                    for (int ci = 0; ci < 4; ++ci) wq[ky][ci] = ld4(wp + ci * 4);
                }
                __builtin_amdgcn_sched_barrier(0);   // keeps the scheduler from sinking the loads to their uses
#pragma unroll
                for (int ky = 0; ky < 7; ++ky)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float4 v = hv[r + ky];
                        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int ci = 0; ci < 4; ++ci) {   // fma chains per output (pack into v_pk_fma_f32 over output pairs)
                            const float4 w = wq[ky][ci];
                            acc[r][0] = fmaf(vv[ci], w.x, acc[r][0]);
                            acc[r][1] = fmaf(vv[ci], w.y, acc[r][1]);
                            acc[r][2] = fmaf(vv[ci], w.z, acc[r][2]);
                            acc[r][3] = fmaf(vv[ci], w.w, acc[r][3]);
                        }
                    }
            }
    }
    const int ox = x0 + tx;
    if (ox >= a.W) return;
    const size_t hw = (size_t)a.H * a.W;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int oy = y0 + tg * 4 + r;
        if (oy >= a.H) break;
        const size_t p = (size_t)oy * a.W + ox;
        const float col[3] = {tanhf(acc[r][0]), tanhf(acc[r][1]), tanhf(acc[r][2])};
        const float m = 1.f / (1.f + expf(-acc[r][3]));
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (a.color) a.color[((size_t)n * 3 + c) * hw + p] = col[c];
            if (a.pred) {
                const float b = a.bg[((size_t)(a.bg_bs > 1 ? n : 0) * 3 + c) * hw + p];
                a.pred[((size_t)n * 3 + c) * hw + p] = m * b + (1.f - m) * col[c];
            }
        }
        if (a.mask) a.mask[(size_t)n * hw + p] = m;
    }
}

}  // namespace

const char *const kIgemmVariantNames[kIgemmVariants] = {
    "conv_igemm_f32<64, 1, 2, false, 0>", "conv_igemm_f32<128, 2, 2, false, 0>", "conv_igemm_f32<64, 1, 2, true, 0>",
    "conv_igemm_dma_f32<64, 1, 2, 0, 3>", "conv_igemm_dma_f32<128, 2, 2, 0, 3>",
    // kernels that run as several instantiations (ring depth / tile height) are named by the prefix rocprofv3 prints
    "conv_igemm_bf16x3<64, 1, 2, 3, 0>", "conv_igemm_bf16x3<128, 2, 2", "stem_bf16x3_kernel",
    "conv3x3_halo_bf16x3<128, 2, 2", "conv3x3_halo_bf16x3<64, 1, 2", "conv_igemm_dma_f32<32, 1, 1, 0, 3>"};

// ---- measurement hook: per-wave cycle accounting of the 128-wide bf16x3 kernel (lwg_conv_trace, tools/conv_trace.py)
constexpr int kTraceMaxLaunches = 4096;
struct TraceLaunch { size_t offset; int gx, gy, gz, waves, stages, cin, cout, hm, n; };
struct TraceState {
    void *buf = nullptr; size_t bytes = 0, used = 0; int n = 0;
    TraceLaunch launch[kTraceMaxLaunches];
};
static TraceState g_trace;

int conv_trace_set(void *device_buffer, size_t bytes)
{
    g_trace.buf = device_buffer; g_trace.bytes = device_buffer ? bytes : 0; g_trace.used = 0; g_trace.n = 0;
    return LWG_OK;
}
int conv_trace_launch(int idx, long long *v)
{
    if (idx < 0 || idx >= g_trace.n) return LWG_ERR_INVALID_ARG;
    const TraceLaunch &L = g_trace.launch[idx];
    v[0] = (long long)L.offset; v[1] = L.gx; v[2] = L.gy; v[3] = L.gz; v[4] = L.waves; v[5] = L.stages; v[6] = L.cin; v[7] = L.cout;
    v[8] = L.hm; v[9] = L.n;
    return LWG_OK;
}

// record block for one traced launch of `waves`-wave workgroups on grid (gx, gy); null when tracing is off or the buffer is full
static unsigned long long *trace_block(const ConvArgs &a, int gx, int gy, int waves, int stages)
{
    if (!g_trace.buf) return nullptr;
    const size_t need = (size_t)gx * gy * waves * 8 * sizeof(unsigned long long);
    if (g_trace.used + need > g_trace.bytes || g_trace.n >= kTraceMaxLaunches) return nullptr;
    unsigned long long *p = reinterpret_cast<unsigned long long *>(static_cast<char *>(g_trace.buf) + g_trace.used);
    TraceLaunch &L = g_trace.launch[g_trace.n++];
    L.offset = g_trace.used; L.gx = gx; L.gy = gy; L.gz = 1; L.waves = waves; L.stages = stages; L.cin = a.Cin; L.cout = a.Cout;
    L.hm = a.Hm; L.n = a.N;
    g_trace.used += need;
    return p;
}

// the shapes conv3x3_halo_bf16x3 takes (3x3 / stride 1 / pad 1, or the four phases of ConvTranspose2d(k3, s2, p1, op1))
static bool halo3x3_shape(const ConvArgs &a)
{
    const ConvPhase &p0 = a.ph[0];
    return a.nphase == 1 && p0.KH == 3 && p0.KW == 3 && a.stride == 1 && a.pad == 1 && a.dil == 1 && a.os == 1 && a.H == a.Hm &&
           a.W == a.Wm && p0.Kpad == 9 * a.Cin && a.Wm % 32 == 0 && a.Hm % 4 == 0;
}
static bool haloCT_shape(const ConvArgs &a)
{
    return a.nphase == 4 && a.os == 2 && a.stride == 1 && a.pad == 0 && a.dil == 1 && a.H == a.Hm && a.W == a.Wm && a.Wm % 32 == 0 &&
           a.Hm % 4 == 0 && a.Cout % 64 == 0 && a.ph[0].KH == 1 && a.ph[0].KW == 1 && a.ph[1].KH == 1 && a.ph[1].KW == 2 &&
           a.ph[2].KH == 2 && a.ph[2].KW == 1 && a.ph[3].KH == 2 && a.ph[3].KW == 2 && a.ph[0].Kpad == a.Cin &&
           a.ph[1].Kpad == 2 * a.Cin && a.ph[2].Kpad == 2 * a.Cin && a.ph[3].Kpad == 4 * a.Cin;
}
bool conv_raw_input_supported(const ConvArgs &a, int bn)
{
    static const char *halo_env = getenv("LWG_HALO"), *fused_env = getenv("LWG_FUSED_APPLY");   // A/B switches ("0": off)
    if ((halo_env && halo_env[0] == '0') || (fused_env && fused_env[0] == '0')) return false;
    if (a.precision != 1 || a.general || !a.w_split || (a.ldx & 31) || (a.Cin & 31) || a.Cin < 32 || (bn != 64 && bn != 128)) return false;
    if (a.raw_from < 0 || (a.raw_from & 31) || a.raw_from >= a.Cin || a.Cin - a.raw_from > 512) return false;
    return halo3x3_shape(a) || haloCT_shape(a);
}

int launch_conv_igemm(const ConvArgs &a_in, int bn, hipStream_t st, int *variant)
{
    ConvArgs a = a_in;   // (the halo launches attach a trace block)
    a.trace = nullptr;   // only this function hands out trace blocks (lwg_conv_trace); never a caller's uninitialised field
    // (32-channel tiles: the exact-fp32 DMA-fed kernel only)
    const bool bn32_ok = bn == 32 && a.precision == 0 && !a.general && a.Cin >= BK && a.zeros;
    if (a.Cout % bn != 0 || (bn != 64 && bn != 128 && !bn32_ok))
        LWG_FAIL(LWG_ERR_UNSUPPORTED, "conv: Cout=%d not a multiple of the %d-channel tile", a.Cout, bn);
    if (!a.general && ((a.Hm * a.Wm) % BM != 0 || a.mtiles * BM != a.N * a.Hm * a.Wm))
        LWG_FAIL(LWG_ERR_UNSUPPORTED, "conv: %dx%d output grid per image is not a multiple of %d pixels", a.Hm, a.Wm, BM);
    if (a.general && (a.partials || a.fuse_phases || (a.precision != 0 && a.precision != 1) ||
                      a.mtiles != ceil_div((long)a.N * a.Hm * a.Wm, BM)))
        LWG_FAIL(LWG_ERR_INVALID_ARG, "conv: general mode is unfused, without statistics, mtiles = ceil(M/128)");
    if (a.ksplit > 1 && (!a.general || !a.kpart || a.kpart_stride < (size_t)a.N * a.Ho * a.Wo * a.ldy))
        LWG_FAIL(LWG_ERR_INVALID_ARG, "conv: a reduction split needs the general mode and a partial buffer of ksplit outputs");
    if (a.dil < 1) LWG_FAIL(LWG_ERR_INVALID_ARG, "conv: dilation must be >= 1");
    if ((1 << a.cin_log2) != a.Cin || a.Cin < 4 || (a.ldx & 3))
        LWG_FAIL(LWG_ERR_UNSUPPORTED, "conv: Cin=%d must be a power of two >= 4 with a 16-byte aligned pixel stride", a.Cin);
    for (int p = 0; p < a.nphase; ++p)
        if (a.ph[p].Kpad % BK != 0 || a.ph[p].Kpad < a.ph[p].ntaps * a.Cin)
            LWG_FAIL(LWG_ERR_UNSUPPORTED, "conv: bad padded K=%d for %d taps x %d channels", a.ph[p].Kpad, a.ph[p].ntaps, a.Cin);
    if (a.precision == 1 && a.tap_inner)
        for (int p = 0; p < a.nphase; ++p)
            if (a.ph[p].Kpad != a.ph[p].ntaps * a.Cin)   // the taps-innermost walk addresses weight column tap*Cin + ci: no K padding
                LWG_FAIL(LWG_ERR_UNSUPPORTED, "conv: taps-innermost walk needs K=%d == %d taps x %d channels", a.ph[p].Kpad,
                         a.ph[p].ntaps, a.Cin);
    if (a.raw_in && (!a.in_ss || !conv_raw_input_supported(a, bn)))
        LWG_FAIL(LWG_ERR_UNSUPPORTED, "conv: raw (un-normalised) input is only taken by the halo-resident bf16x3 kernels");
    const dim3 grid(a.mtiles, a.Cout / bn, a.fuse_phases ? 1 : a.nphase * (a.ksplit > 1 ? a.ksplit : 1));
    const size_t lds = (size_t)2 * (BM + bn) * LDK * sizeof(float);
    // the 128-channel tile needs 72 KiB of LDS: above the 64 KiB default, well inside gfx950's 160 KiB per CU
    static DeviceOnce lds_opt_in;
    if (!lds_opt_in.done()) {
        const int l64 = 2 * (BM + 64) * LDK * (int)sizeof(float), l128 = 2 * (BM + 128) * LDK * (int)sizeof(float);
        LWG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_f32<64, 1, 2, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, l64));
        LWG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_f32<64, 1, 2, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, l64));
        LWG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_f32<128, 2, 2, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, l128));
        lds_opt_in.mark();
    }
    const bool small_cin = a.Cin < BK;
    if (a.general) {
        static DeviceOnce gen_opt_in;
        if (!gen_opt_in.done()) {
            const int l64 = 2 * (BM + 64) * LDK * (int)sizeof(float), l128 = 2 * (BM + 128) * LDK * (int)sizeof(float);
            LWG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_f32<64, 1, 2, false, 0, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, l64));
            LWG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_f32<64, 1, 2, true, 0, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, l64));
            LWG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_f32<128, 2, 2, false, 0, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, l128));
            gen_opt_in.mark();
        }
        if (small_cin && bn != 64) LWG_FAIL(LWG_ERR_UNSUPPORTED, "conv: Cin=%d < %d is only built for the 64-channel tile", a.Cin, BK);
        for (int p = 0; p < a.nphase; ++p)
            if (a.ph[p].ntaps > 64) LWG_FAIL(LWG_ERR_UNSUPPORTED, "conv: more than 64 taps");
        if (a.precision == 1) {   // plain fp32 operands, split in the kernel (conv_igemm_f32<..., X3>)
            static DeviceOnce x3_opt_in;
            if (!x3_opt_in.done()) {
                const int l128 = 2 * (BM + 128) * LDK * (int)sizeof(float);
                LWG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_f32<128, 2, 2, false, 0, true, true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, l128));
                x3_opt_in.mark();
            }
            if (small_cin) conv_igemm_f32<64, 1, 2, true, 0, true, true><<<grid, 256, lds, st>>>(a);
            else if (bn == 64) conv_igemm_f32<64, 1, 2, false, 0, true, true><<<grid, 256, lds, st>>>(a);
            else conv_igemm_f32<128, 2, 2, false, 0, true, true><<<grid, 256, lds, st>>>(a);
        } else if (small_cin) {
            conv_igemm_f32<64, 1, 2, true, 0, true><<<grid, 256, lds, st>>>(a);
        } else if (bn == 64) {
            conv_igemm_f32<64, 1, 2, false, 0, true><<<grid, 256, lds, st>>>(a);
        } else {
            conv_igemm_f32<128, 2, 2, false, 0, true><<<grid, 256, lds, st>>>(a);
        }
        if (variant) *variant = small_cin ? kIgemmSmallCin : (bn == 64 ? kIgemmReg64 : kIgemmReg128);
        LWG_LAUNCH_CHECK("conv_igemm_f32 (general)");
        return LWG_OK;
    }
    if (a.precision == 1) {
        // bf16x3 on split operands: DMA-fed ring of (BM + bn) 128-byte rows per stage
        if (small_cin || !a.w_split || !a.zeros || (a.ldx & 31) || (a.Cin & 31))
            LWG_FAIL(LWG_ERR_UNSUPPORTED, "conv: the bf16x3 path needs split weights and Cin, pixel stride multiples of 32");
        // lane offsets are unsigned 32-bit bytes from the image base: the zero run must sit above the whole input
        // tensor and within reach of its first image (see igemm_dma_body)
        {
            const char *x_end = reinterpret_cast<const char *>(a.x + (size_t)a.N * a.H * a.W * a.ldx);
            const char *z = reinterpret_cast<const char *>(a.zeros);
            if (z < x_end || (size_t)(z - reinterpret_cast<const char *>(a.x)) + (size_t)a.Cin * 4 > 0xffffffffull)
                LWG_FAIL(LWG_ERR_UNSUPPORTED, "conv: the bf16x3 path needs its zero run behind the input tensor, within 4 GiB");
            for (int p = 0; p < a.nphase; ++p)
                if ((size_t)a.Cout * a.ph[p].Kpad * 4 > 0xffffffffull)
                    LWG_FAIL(LWG_ERR_UNSUPPORTED, "conv: weight matrix too large for 32-bit lane offsets");
        }
        const size_t lds3 = (size_t)3 * (BM + bn) * BK * sizeof(float);
        static DeviceOnce opt16;
        if (!opt16.done()) {
            LWG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_bf16x3<64, 1, 2>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (BM + 64) * BK * (int)sizeof(float)));
            LWG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_bf16x3<128, 2, 2>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (BM + 128) * BK * (int)sizeof(float)));
            opt16.mark();
        }
        for (int p = 0; p < a.nphase; ++p)
            if (a.ph[p].ntaps > 32) LWG_FAIL(LWG_ERR_UNSUPPORTED, "conv: more than 32 taps on the bf16x3 path");
        // 3x3 / stride 1 / pad 1 with whole 128-pixel row blocks: the halo-resident kernel (conv3x3_halo_bf16x3)
        static const char *halo_env = getenv("LWG_HALO");   // "0": the DMA-ring kernel everywhere (A/B switch)
        if (!(halo_env && halo_env[0] == '0') && halo3x3_shape(a)) {
            const int tc = 32;
            static DeviceOnce halo_opt[8];
            const size_t raw_bytes = a.raw_in ? (size_t)8 * BK * sizeof(float) + (size_t)(a.Cin - a.raw_from) * sizeof(float2) : 0;
            auto run = [&](auto kern, int ns, int bmt, DeviceOnce &once) -> int {
                const int hp = (bmt / tc + 2) * (tc + 2), nch = (hp + 7) / 8;
                const size_t base = ((size_t)2 * nch * 8 * BK + (size_t)ns * bn * BK) * sizeof(float), bytes = base + raw_bytes;
                if (!once.done()) {   // the raw-input variants size their (scale, shift) table per launch: opt in for up to 512 channels
                    const size_t most = base + (a.raw_in ? (size_t)8 * BK * sizeof(float) + 512 * sizeof(float2) : 0);
                    LWG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)most));
                    once.mark();
                }
                const dim3 g(grid.x / (bmt / BM), grid.y, 1);
                a.trace = trace_block(a, g.x, g.y, bmt == 256 ? 8 : 4, 9 * a.Cin / BK);
                kern<<<g, bmt == 256 ? 512 : 256, bytes, st>>>(a);
                return LWG_OK;
            };
            // 8 x 32-pixel tiles with eight waves once every CU still gets one (the weight stage is shared by twice the MFMAs)
            static const char *tall_env = getenv("LWG_HALO_TALL");   // "0": 4 x 32 tiles only (A/B switch)
            const bool tall = bn == 128 && a.Hm % 8 == 0 && (long)(a.mtiles / 2) * (a.Cout / 128) >= device_cu_count() &&
                              !(tall_env && tall_env[0] == '0');
            // (the 64-channel tile on eight waves of 32 x 64 -- one 110 KiB workgroup instead of two of 76 -- measured slower:
            // 380 -> 348 TFLOP/s on skipper.2, gpurun_out/r03o)
            int rc;
            if (a.raw_in) {   // the input's InstanceNorm + ReLU + split folded into the halo (see HaloSched)
                if (tall) rc = run(&conv3x3_halo_bf16x3<128, 2, 2, 4, 32, 256, 0, true>, 4, 256, halo_opt[6]);
                else if (bn == 128) rc = run(&conv3x3_halo_bf16x3<128, 2, 2, 4, 32, 128, 0, true>, 4, 128, halo_opt[4]);
                else rc = run(&conv3x3_halo_bf16x3<64, 1, 2, 3, 32, 128, 0, true>, 3, 128, halo_opt[5]);
            } else if (tall) rc = run(&conv3x3_halo_bf16x3<128, 2, 2, 4, 32, 256>, 4, 256, halo_opt[2]);
            else if (bn == 128) rc = run(&conv3x3_halo_bf16x3<128, 2, 2, 4, 32>, 4, 128, halo_opt[0]);
            else rc = run(&conv3x3_halo_bf16x3<64, 1, 2, 3, 32>, 3, 128, halo_opt[1]);   // 52 + 24 KiB: two workgroups per CU
            if (rc != LWG_OK) return rc;
            if (variant) *variant = bn == 128 ? kHaloBf16x3_128 : kHaloBf16x3_64;
            LWG_LAUNCH_CHECK("conv3x3_halo_bf16x3");
            return LWG_OK;
        }
        // ConvTranspose2d(k3, s2, p1, op1) given as its four phases: one launch of the halo kernel (CT = 1)
        if (!(halo_env && halo_env[0] == '0') && haloCT_shape(a)) {
            static const char *ct_wide_env = getenv("LWG_CT_WIDE");   // "0": the 64-channel tile (two workgroups per CU) everywhere (A/B switch)
            const bool wide = a.Cout % 128 == 0 && (long)a.mtiles * (a.Cout / 128) >= device_cu_count() &&
                              !(ct_wide_env && ct_wide_env[0] == '0');
            const int cbn = wide ? 128 : 64, ns = wide ? 4 : 3;
            const int nch = ((4 + 1) * (32 + 1) + 7) / 8;
            const size_t bytes = ((size_t)2 * nch * 8 * BK + (size_t)ns * cbn * BK) * sizeof(float) +
                                 (a.raw_in ? (size_t)8 * BK * sizeof(float) + (size_t)(a.Cin - a.raw_from) * sizeof(float2) : 0);
            const dim3 g(a.mtiles, a.Cout / cbn, 1);
            a.trace = trace_block(a, g.x, g.y, 4, -(int)(9 * a.Cin / BK));   // negative stage count marks a transposed conv
            static DeviceOnce ct_opt[4];
            if (a.raw_in && wide) {
                auto kern = &conv3x3_halo_bf16x3<128, 2, 2, 4, 32, 128, 1, true>;
                if (!ct_opt[2].done()) {
                    LWG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    ct_opt[2].mark();
                }
                kern<<<g, 256, bytes, st>>>(a);
            } else if (a.raw_in) {
                auto kern = &conv3x3_halo_bf16x3<64, 1, 2, 3, 32, 128, 1, true>;
                if (!ct_opt[3].done()) {
                    LWG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    ct_opt[3].mark();
                }
                kern<<<g, 256, bytes, st>>>(a);
            } else if (wide) {
                auto kern = &conv3x3_halo_bf16x3<128, 2, 2, 4, 32, 128, 1>;
                if (!ct_opt[0].done()) {
                    LWG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
                    ct_opt[0].mark();
                }
                kern<<<g, 256, bytes, st>>>(a);
            } else {
                auto kern = &conv3x3_halo_bf16x3<64, 1, 2, 3, 32, 128, 1>;
                if (!ct_opt[1].done()) {
                    LWG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
                    ct_opt[1].mark();
                }
                kern<<<g, 256, bytes, st>>>(a);
            }
            if (variant) *variant = wide ? kHaloBf16x3_128 : kHaloBf16x3_64;
            LWG_LAUNCH_CHECK("conv3x3_halo_bf16x3 (transposed)");
            return LWG_OK;
        }
        static const char *ring = getenv("LWG_RING");   // "3": the round-2 3-slot ring on the 128-wide tile (A/B switch)
        const bool ring4 = !(ring && ring[0] == '3');
        if (g_trace.buf && bn == 128 && ring4) {
            // measurement hook (lwg_conv_trace): the same kernel with s_memtime accounting, one record block per launch
            const size_t need = (size_t)grid.x * grid.y * grid.z * 4 * 8 * sizeof(unsigned long long);
            if (g_trace.used + need <= g_trace.bytes && g_trace.n < kTraceMaxLaunches) {
                ConvArgs t = a;
                t.trace = reinterpret_cast<unsigned long long *>(static_cast<char *>(g_trace.buf) + g_trace.used);
                TraceLaunch &L = g_trace.launch[g_trace.n++];
                L.offset = g_trace.used; L.gx = grid.x; L.gy = grid.y; L.gz = grid.z; L.waves = 4;
                L.stages = a.ph[a.fuse_phases ? 0 : 0].Kpad / BK; L.cin = a.Cin; L.cout = a.Cout; L.hm = a.Hm; L.n = a.N;
                g_trace.used += need;
                static DeviceOnce optt;
                if (!optt.done()) {
                    LWG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_bf16x3<128, 2, 2, 4, 512>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (BM + 128) * BK * (int)sizeof(float)));
                    optt.mark();
                }
                conv_igemm_bf16x3<128, 2, 2, 4, 512><<<grid, 256, (size_t)4 * (BM + 128) * BK * sizeof(float), st>>>(t);
                if (variant) *variant = kIgemmBf16x3_128;
                LWG_LAUNCH_CHECK("conv_igemm_bf16x3 (traced)");
                return LWG_OK;
            }
        }
        // 256 x 128 tiles on eight waves (two per SIMD) where every CU still gets one: the two 128-row halves share the
        // weight stage and a workgroup's prologue / epilogue are paid once per 256 rows -- the stride-2 encoders
        static const char *tall_ring_env = getenv("LWG_RING_TALL");   // "0": 128-row tiles only (A/B switch)
        // (a tile's rows are img * Hm*Wm + rem without a wrap: a 256-row tile must not straddle two images)
        if (bn == 128 && !a.fuse_phases && a.nphase == 1 && (a.mtiles & 1) == 0 && (a.Hm * a.Wm) % 256 == 0 && !g_trace.buf &&
            (long)(a.mtiles / 2) * (a.Cout / 128) >= device_cu_count() && !(tall_ring_env && tall_ring_env[0] == '0')) {
            static DeviceOnce opt_tall;
            const size_t lds_t = (size_t)3 * (256 + 128) * BK * sizeof(float);
            if (!opt_tall.done()) {
                LWG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_bf16x3<128, 2, 2, 3, 0, 256>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t));
                opt_tall.mark();
            }
            conv_igemm_bf16x3<128, 2, 2, 3, 0, 256><<<dim3(grid.x / 2, grid.y, 1), 512, lds_t, st>>>(a);
            if (variant) *variant = kIgemmBf16x3_128;
            LWG_LAUNCH_CHECK("conv_igemm_bf16x3 (256-row tiles)");
            return LWG_OK;
        }
        if (bn == 64) {
            conv_igemm_bf16x3<64, 1, 2><<<grid, 256, lds3, st>>>(a);
#ifdef LWG_EXPERIMENTS   // builds for profiles/r03_conv_experiments.md only (hipcc -DLWG_EXPERIMENTS): never in the shipped library
        } else if (ring && ring[0] == 'x') {   // LWG_RING=x: timing experiment, WRONG RESULTS (see DBG & 1024)
            static DeviceOnce optx;
            if (!optx.done()) {
                LWG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_bf16x3<128, 2, 2, 4, 1024>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (BM + 128) * BK * (int)sizeof(float)));
                optx.mark();
            }
            conv_igemm_bf16x3<128, 2, 2, 4, 1024><<<grid, 256, (size_t)4 * (BM + 128) * BK * sizeof(float), st>>>(a);
        } else if (ring && ring[0] == '5') {   // measurement switch: 5 slots = all 160 KiB of a CU's LDS
            static DeviceOnce opt5;
            if (!opt5.done()) {
                LWG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_bf16x3<128, 2, 2, 5>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 5 * (BM + 128) * BK * (int)sizeof(float)));
                opt5.mark();
            }
            conv_igemm_bf16x3<128, 2, 2, 5><<<grid, 256, (size_t)5 * (BM + 128) * BK * sizeof(float), st>>>(a);
#endif
        } else if (ring4) {
            static DeviceOnce opt4;
            if (!opt4.done()) {
                LWG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_bf16x3<128, 2, 2, 4>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (BM + 128) * BK * (int)sizeof(float)));
                opt4.mark();
            }
            conv_igemm_bf16x3<128, 2, 2, 4><<<grid, 256, (size_t)4 * (BM + 128) * BK * sizeof(float), st>>>(a);
        } else {
            conv_igemm_bf16x3<128, 2, 2><<<grid, 256, lds3, st>>>(a);
        }
        if (variant) *variant = bn == 64 ? kIgemmBf16x3_64 : kIgemmBf16x3_128;
        LWG_LAUNCH_CHECK("conv_igemm_bf16x3");
        return LWG_OK;
    }
    if (small_cin && bn != 64) LWG_FAIL(LWG_ERR_UNSUPPORTED, "conv: Cin=%d < %d is only built for the 64-channel tile", a.Cin, BK);
    // Kernel choice (tools/igemm_bench.hip, warm clocks, TFLOP/s): the 64-channel tile runs best DMA-fed (118-122 vs
    // 115-120); the 128-channel tile runs best DMA-fed when the grid is a single round of one workgroup per CU
    // (the 96 KiB ring allows one per CU anyway: 127 vs 122) and register-staged when several rounds are queued
    // (74 KiB: two resident workgroups per CU, 130 vs 126).  Both kernels add the products of an output in the
    // same order, so the choice never changes a result bit.
    const int ncu = device_cu_count();
    const long nblocks = (long)grid.x * grid.y * grid.z;
    bool few_taps = true;   // the DMA ring keeps a 32-bit tap mask per row: a 7x7 stem on 32 padded channels is register-staged
    for (int p = 0; p < a.nphase; ++p) few_taps = few_taps && a.ph[p].ntaps <= 32;
    const bool use_dma = !small_cin && a.zeros && few_taps && (bn <= 64 || nblocks <= ncu);
    if (a.fuse_phases && !use_dma) LWG_FAIL(LWG_ERR_UNSUPPORTED, "conv: fused phases need the DMA-fed kernel (Cin >= 32, 64-channel tile)");
    if (use_dma) {
        // DMA-fed 3-stage ring: (BM + bn) * 32 floats per stage
        const size_t lds_dma = (size_t)3 * (BM + bn) * BK * sizeof(float);
        static DeviceOnce dma_opt_in;
        if (!dma_opt_in.done()) {
            LWG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_dma_f32<64, 1, 2>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (BM + 64) * BK * (int)sizeof(float)));
            LWG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_dma_f32<128, 2, 2>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (BM + 128) * BK * (int)sizeof(float)));
            dma_opt_in.mark();
        }
        for (int p = 0; p < a.nphase; ++p)
            if (a.ph[p].ntaps > 32) LWG_FAIL(LWG_ERR_UNSUPPORTED, "conv: more than 32 taps on the DMA path");
        if (bn == 32) {
            // 128 pixels x 32 channels, one 32 x 32 MFMA tile per wave: a launch of 64 of the 64-channel tiles (one source's
            // 512 -> 512 layer on a 32 x 32 map) becomes 128 workgroups.  Same products in the same order per output: same bits.
            static DeviceOnce dma32_opt_in;
            if (!dma32_opt_in.done()) {
                LWG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_dma_f32<32, 1, 1>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (BM + 32) * BK * (int)sizeof(float)));
                dma32_opt_in.mark();
            }
            conv_igemm_dma_f32<32, 1, 1><<<grid, 256, lds_dma, st>>>(a);
        } else if (bn == 64) {
            conv_igemm_dma_f32<64, 1, 2><<<grid, 256, lds_dma, st>>>(a);
        } else {
            conv_igemm_dma_f32<128, 2, 2><<<grid, 256, lds_dma, st>>>(a);
        }
        if (variant) *variant = bn == 32 ? kIgemmDma32 : (bn == 64 ? kIgemmDma64 : kIgemmDma128);
        LWG_LAUNCH_CHECK("conv_igemm_dma_f32");
        return LWG_OK;
    }
    if (bn != 64 && bn != 128) LWG_FAIL(LWG_ERR_UNSUPPORTED, "conv: the register-staged fp32 kernel has 64- and 128-channel tiles, not %d", bn);
    if (variant) *variant = small_cin ? kIgemmSmallCin : (bn == 64 ? kIgemmReg64 : kIgemmReg128);
    if (small_cin) {
        conv_igemm_f32<64, 1, 2, true><<<grid, 256, lds, st>>>(a);
    } else if (bn == 64) {
        conv_igemm_f32<64, 1, 2, false><<<grid, 256, lds, st>>>(a);
    } else {
        conv_igemm_f32<128, 2, 2, false><<<grid, 256, lds, st>>>(a);
    }
    LWG_LAUNCH_CHECK("conv_igemm_f32");
    return LWG_OK;
}

#ifdef LWG_IGEMM_BENCH
// ablation launcher for tools/igemm_bench.hip
template <int DBG>
static void launch_dbg(const ConvArgs &a, int bn, hipStream_t st)
{
    const dim3 grid(a.mtiles, a.Cout / bn, a.nphase);
    const size_t lds = (size_t)2 * (BM + bn) * LDK * sizeof(float);
    if (bn == 64) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_f32<64, 1, 2, false, DBG>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        conv_igemm_f32<64, 1, 2, false, DBG><<<grid, 256, lds, st>>>(a);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_f32<128, 2, 2, false, DBG>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        conv_igemm_f32<128, 2, 2, false, DBG><<<grid, 256, lds, st>>>(a);
    }
}
template <int DBG>
static void launch_dma_dbg(const ConvArgs &a, int bn, hipStream_t st)
{
    const dim3 grid(a.mtiles, a.Cout / bn, a.nphase);
    const size_t lds = (size_t)3 * (BM + bn) * BK * sizeof(float);
    if (bn == 64) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_dma_f32<64, 1, 2, DBG>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        conv_igemm_dma_f32<64, 1, 2, DBG><<<grid, 256, lds, st>>>(a);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_dma_f32<128, 2, 2, DBG>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        conv_igemm_dma_f32<128, 2, 2, DBG><<<grid, 256, lds, st>>>(a);
    }
}
template <int NS, int DBG>
static void launch_k_dbg(const ConvArgs &a, int bn, hipStream_t st)
{
    const dim3 grid(a.mtiles, a.Cout / bn, a.nphase);
    const size_t lds = (size_t)NS * (BM + bn) * BK * sizeof(float);
    if (bn == 64) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_bf16x3<64, 1, 2, NS, DBG>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        conv_igemm_bf16x3<64, 1, 2, NS, DBG><<<grid, 256, lds, st>>>(a);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_bf16x3<128, 2, 2, NS, DBG>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        conv_igemm_bf16x3<128, 2, 2, NS, DBG><<<grid, 256, lds, st>>>(a);
    }
}
template <int BN, int WM, int WN, int DBG>
static void launch_w_dbg(const ConvArgs &a, hipStream_t st)
{
    const dim3 grid(a.mtiles, a.Cout / BN, a.nphase);
    const size_t lds = (size_t)3 * (BM + BN) * BK * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_bf16x3<BN, WM, WN, 3, DBG>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    conv_igemm_bf16x3<BN, WM, WN, 3, DBG><<<grid, 64 * (BM / (32 * WM)) * (BN / (32 * WN)), lds, st>>>(a);
}
int launch_conv_igemm_dbg(const ConvArgs &a, int bn, int dbg, hipStream_t st)
{
    switch (dbg) {
        // eight waves (two per SIMD) on the same tile: wave tile 64x32 / 32x64 (bn 128), 32x32 (bn 64)
        case 821: if (bn == 128) launch_w_dbg<128, 2, 1, 0>(a, st); else launch_w_dbg<64, 1, 1, 0>(a, st); break;
        case 812: if (bn == 128) launch_w_dbg<128, 1, 2, 0>(a, st); else launch_w_dbg<64, 1, 1, 0>(a, st); break;
        case 831: if (bn == 128) launch_w_dbg<128, 2, 1, 1>(a, st); else launch_w_dbg<64, 1, 1, 1>(a, st); break;   // no DMA
        case 856: {   // 256x64 tile (bn 64 only)
            if (bn != 64 || (a.mtiles & 1) || (a.Hm * a.Wm) % 256 != 0) return LWG_ERR_INVALID_ARG;
            const dim3 grid(a.mtiles / 2, a.Cout / 64, a.nphase);
            const size_t lds = (size_t)3 * (256 + 64) * BK * sizeof(float);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_bf16x3<64, 2, 2, 3, 0, 256>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            conv_igemm_bf16x3<64, 2, 2, 3, 0, 256><<<grid, 256, lds, st>>>(a);
            break;
        }
        case 300: {   // whatever launch_conv_igemm picks for the bf16x3 path (the halo kernel where it applies)
            ConvArgs b = a;
            b.precision = 1;
            return launch_conv_igemm(b, bn, st, nullptr);
        }
        case 200: launch_k_dbg<3, 0>(a, bn, st); break;
        case 201: launch_k_dbg<3, 1>(a, bn, st); break;   // no DMA
        case 204: launch_k_dbg<3, 4>(a, bn, st); break;   // no barrier
        case 213: launch_k_dbg<3, 13>(a, bn, st); break;  // MFMAs only
        case 240: launch_k_dbg<4, 0>(a, bn, st); break;   // 4-slot ring
        case 241: launch_k_dbg<4, 1>(a, bn, st); break;   // 4-slot ring, no steady-state DMA
        case 245: launch_k_dbg<4, 13>(a, bn, st); break;  // 4-slot ring, MFMAs only
        case 1240: launch_k_dbg<4, 1024>(a, bn, st); break;   // 4-slot ring, a quarter of the activation DMAs (timing only)
        case 232: launch_k_dbg<3, 32>(a, bn, st); break;  // no epilogue
        case 264: launch_k_dbg<3, 64>(a, bn, st); break;  // natural tile order (no XCD bands)
        case 328: launch_k_dbg<3, 128>(a, bn, st); break; // 4-byte epilogue stores
        case 456: launch_k_dbg<3, 256>(a, bn, st); break; // fragment reads in two bulk groups
        case 0: launch_dbg<0>(a, bn, st); break;
        case 1: launch_dbg<1>(a, bn, st); break;
        case 3: launch_dbg<3>(a, bn, st); break;
        case 7: launch_dbg<7>(a, bn, st); break;
        case 15: launch_dbg<15>(a, bn, st); break;
        case 16: launch_dbg<16>(a, bn, st); break;
        case 4: launch_dbg<4>(a, bn, st); break;
        case 100: launch_dma_dbg<0>(a, bn, st); break;
        case 116: launch_dma_dbg<16>(a, bn, st); break;
        case 101: launch_dma_dbg<1>(a, bn, st); break;
        case 104: launch_dma_dbg<4>(a, bn, st); break;
        case 140: {   // 4-slot ring (3 stages in flight)
            const dim3 grid(a.mtiles, a.Cout / bn, a.nphase);
            const size_t lds4 = (size_t)4 * (BM + bn) * BK * sizeof(float);
            if (bn == 64) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_dma_f32<64, 1, 2, 0, 4>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
                conv_igemm_dma_f32<64, 1, 2, 0, 4><<<grid, 256, lds4, st>>>(a);
            } else {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_igemm_dma_f32<128, 2, 2, 0, 4>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
                conv_igemm_dma_f32<128, 2, 2, 0, 4><<<grid, 256, lds4, st>>>(a);
            }
            break;
        }
        default: return LWG_ERR_INVALID_ARG;
    }
    return LWG_OK;
}
#endif

int launch_in_finalize(const float2 *partials, int nphase, int mtiles, int N, int C, const float *gamma,
                       const float *beta, float eps, float2 *scale_shift, hipStream_t st)
{
    if (mtiles % N != 0) LWG_FAIL(LWG_ERR_UNSUPPORTED, "in_finalize: %d tiles do not split over %d images", mtiles, N);
    const dim3 grid(ceil_div(C, FIN_CH), N);
    in_finalize_kernel<<<grid, 256, 0, st>>>(partials, nphase, mtiles, mtiles / N, C, gamma, beta, eps, scale_shift);
    LWG_LAUNCH_CHECK("in_finalize_kernel");
    return LWG_OK;
}

int launch_apply(const ApplyArgs &a, hipStream_t st)
{
    if ((a.C & 3) || (a.ld_dst & 3) || (a.res && (a.ld_res & 3)))
        LWG_FAIL(LWG_ERR_UNSUPPORTED, "apply: channel counts/strides must be multiples of 4 (C=%d)", a.C);
    if (a.split && ((a.C & 31) || (a.ld_dst & 31) || ((uintptr_t)a.dst & 127) ||
                    (a.res && ((a.ld_res & 31) || ((uintptr_t)a.res & 127)))))
        LWG_FAIL(LWG_ERR_UNSUPPORTED, "apply: split-bf16 buffers need 32-channel granularity (C=%d)", a.C);
    static const char *a8_env = getenv("LWG_APPLY8");   // "0": four channels per thread everywhere (A/B switch)
    if (a.split && !(a8_env && a8_env[0] == '0')) {
        const long total8 = (long)a.N * a.H * a.W * (a.C >> 3);
        apply8_kernel<<<ceil_div(total8, 256), 256, 0, st>>>(a);
        LWG_LAUNCH_CHECK("apply8_kernel");
        return LWG_OK;
    }
    const long total = (long)a.N * a.H * a.W * (a.C >> 2);
    apply_kernel<<<ceil_div(total, 256), 256, 0, st>>>(a);
    LWG_LAUNCH_CHECK("apply_kernel");
    return LWG_OK;
}

int launch_unsplit(float *buf, size_t n, hipStream_t st)
{
    if (n % 32) LWG_FAIL(LWG_ERR_INVALID_ARG, "unsplit: %zu floats is not a whole number of 32-value groups", n);
    if (!n) return LWG_OK;
    unsplit_kernel<<<(unsigned)ceil_div((long)(n / 32), 256), 256, 0, st>>>(buf, n / 32);
    LWG_LAUNCH_CHECK("unsplit_kernel");
    return LWG_OK;
}

int launch_heads(const HeadsArgs &a, hipStream_t st)
{
    if (a.pred && !a.bg) LWG_FAIL(LWG_ERR_INVALID_ARG, "heads: pred requested without a background image");
    static DeviceOnce opt_in;
    if (!opt_in.done()) {
        LWG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&heads_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    HEADS_LDS));
        opt_in.mark();
    }
    const dim3 grid(ceil_div(a.W, HT), ceil_div(a.H, HT), a.N);
    heads_kernel<<<grid, 256, HEADS_LDS, st>>>(a);
    LWG_LAUNCH_CHECK("heads_kernel");
    return LWG_OK;
}

}  // namespace lwg
