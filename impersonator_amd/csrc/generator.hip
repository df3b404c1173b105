// generator.hip -- host side of the lwg_generator handle: weight re-layout, scratch, layer schedule.
//
// Mirrors the data flow of the reference's ImpersonatorGenerator (networks/generator.py:187-320):
//   encode_src : src_model encoders + residual blocks, every level kept      (generator.py:136-147,213-214)
//   inference  : tsf_model with the Liquid Warping Block add per level        (generator.py:277-301)
//   swap       : the two-source variant                                       (generator.py:245-275)
// Every layer is  conv_igemm (raw output + tile statistics) -> in_finalize -> apply  (see conv.hip).
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include <cstdlib>

#include "conv.h"

namespace lwg {
namespace {

constexpr int kNDown = 3;
constexpr float kInEps = 1e-5f;  // nn.InstanceNorm2d default

int ilog2(int v)
{
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

struct Layer {
    // geometry
    int cin = 0, cin_pad = 0, cout = 0, k = 0, stride = 1, pad = 0;
    bool transposed = false, has_norm = true;
    // device weights: conv [cout][kpad]; transposed: four phase matrices back to back
    float *w = nullptr;
    float *w_split = nullptr;         // the same matrices in the split-bf16 format (conv.h), Cin >= 32 layers only
    void *w_stem = nullptr;           // 7x7 stem only: filter bank packed for the direct bf16x3 kernel (direct.hip)
    ConvPhase ph[4];
    int nphase = 1;
    float *gamma = nullptr, *beta = nullptr;
    bool got_w = false, got_g = false, got_b = false;
    size_t w_floats = 0;
};

struct StreamNet {
    Layer enc[kNDown + 1];
    std::vector<Layer> res;  // 2 per block
    Layer dec[kNDown], skip[kNDown];
    float *heads_w = nullptr;  // [49][64][4]
    void *heads_frag = nullptr;       // heads_w as bf16x3 MFMA B fragments (heads.hip), repacked on the device when stale
    bool heads_frag_stale = true;
    bool got_img = false, got_att = false;
    bool has_skip = true;      // false: BGNet (ResNetGenerator): decoders without skip connections, one 3-channel head
};

}  // namespace
}  // namespace lwg

using namespace lwg;

struct lwg_generator {
    int src_dim, tsf_dim, cd, repeat, is, max_batch;
    StreamNet src, tsf, bg;
    int bg_dim = 0;                   // 0: BGNet not enabled (lwg_generator_enable_bg)
    int ignored_keys = 0;

    // scratch (sized for max_batch)
    float *x0 = nullptr;              // (bs,is,is,x0_c) packed input
    int x0_c = 8;                     // widest padded stem input of the streams (8 for the default 6- / 4-channel inputs)
    float *raw = nullptr;             // largest raw conv output
    float *cat[kNDown] = {};          // cat[l]: (bs, is>>l, is>>l, 2*cd<<l): [skip | decoder]
    float *trunk[3] = {};             // (bs, is/8, is/8, 8cd) ping/pong/mid
    float *sk[kNDown] = {};           // skipper outputs sk[i] at level (2-i): sk[0] 64^2x4cd ... (last one stays raw)
    float *tscale[2][kNDown] = {};    // resized flows per level (two sets for swap)
    float2 *partials = nullptr;
    float2 *ss = nullptr;             // scale/shift [max_batch][8cd]
    float *zeros = nullptr;           // 256 B of zeros: source of out-of-image taps for the DMA-fed conv kernel
    size_t cat_floats[kNDown] = {}, trunk_floats = 0, sk_floats[kNDown] = {};   // payload sizes (zero tails follow)
    int trunk_out = 0;                // which trunk buffer holds the residual trunk's output
    int precision = 1;                // tsf-stream conv arithmetic: 0 exact fp32 MFMA, 1 bf16x3 split (default)
    bool split = false;               // activation format of the pass being enqueued (split-bf16 when true)
    bool last_split = false;          // ... of the last tsf pass (what peek finds in the buffers)

    // profiling of the implicit-GEMM kernel
    bool profile = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    std::vector<int> ev_variant;      // kernel instantiation of each bracketed launch
    double prof_flops[kIgemmVariants] = {};
    int prof_launches[kIgemmVariants] = {};
};

namespace lwg {
namespace {

int dev_alloc(float **p, size_t floats)
{
    LWG_HIP(hipMalloc(reinterpret_cast<void **>(p), floats * sizeof(float)));
    return LWG_OK;
}

// Activation buffer that feeds a conv: kZeroTail floats of zeros sit right behind it.  The bf16x3 kernel addresses
// out-of-image taps with 32-bit lane offsets from the image base and needs a zero run of >= Cin*4 bytes it can reach.
constexpr size_t kZeroTail = 1024;
int act_alloc(float **p, size_t floats)
{
    LWG_HIP(hipMalloc(reinterpret_cast<void **>(p), (floats + kZeroTail) * sizeof(float)));
    LWG_HIP(hipMemset(*p + floats, 0, kZeroTail * sizeof(float)));
    return LWG_OK;
}

void init_conv(Layer &L, int cin, int cout, int k, int stride, int pad)
{
    L.cin = cin;
    // input channels padded to what the kernels address: 8 (NHWC8: the default condition maps), 16 or 32 for the wide condition
    // maps of utils/mesh.py:446-473 ('par': 3 + 11, 'binary': 3 + 15); inner layers have powers of two >= 64
    L.cin_pad = cin <= 8 ? 8 : cin <= 16 ? 16 : cin <= 32 ? 32 : cin;
    L.cout = cout;
    L.k = k;
    L.stride = stride;
    L.pad = pad;
    L.transposed = false;
    L.nphase = 1;
    ConvPhase &p = L.ph[0];
    p.KH = p.KW = k;
    p.ntaps = k * k;
    p.Kpad = (int)align_up((size_t)p.ntaps * L.cin_pad, kConvBK);
    p.w_off = 0;
    p.oy0 = p.ox0 = 0;
    L.w_floats = (size_t)cout * p.Kpad;
}

// ConvTranspose2d(k=3, s=2, p=1, output_padding=1): out(2i+py, 2j+px) only sees taps whose parity matches;
// phase (py,px) is a stride-1 correlation with (1+py) x (1+px) taps reading in(i+ty, j+tx).
void init_convT(Layer &L, int cin, int cout)
{
    L.cin = L.cin_pad = cin;
    L.cout = cout;
    L.k = 3;
    L.stride = 1;
    L.pad = 0;
    L.transposed = true;
    L.nphase = 4;
    long off = 0;
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            ConvPhase &p = L.ph[py * 2 + px];
            p.KH = 1 + py;
            p.KW = 1 + px;
            p.ntaps = p.KH * p.KW;
            p.Kpad = (int)align_up((size_t)p.ntaps * cin, kConvBK);
            p.w_off = off;
            p.oy0 = py;
            p.ox0 = px;
            off += (long)cout * p.Kpad;
        }
    L.w_floats = (size_t)off;
}

int alloc_layer(Layer &L)
{
    int rc = dev_alloc(&L.w, L.w_floats);
    if (rc != LWG_OK) return rc;
    LWG_HIP(hipMemset(L.w, 0, L.w_floats * sizeof(float)));
    // (a 7x7 stem with 32 padded input channels has 49 taps: more than the bf16x3 ring walks -- it stays on the exact-fp32 kernel)
    if (L.cin_pad >= kConvBK && L.k * L.k <= 32) {
        LWG_HIP(hipMalloc(reinterpret_cast<void **>(&L.w_split), L.w_floats * sizeof(float)));
        LWG_HIP(hipMemset(L.w_split, 0, L.w_floats * sizeof(float)));
    } else if (!L.transposed && stem_bf16x3_supported(kConvBM, kConvBM, L.cin, L.cin_pad, L.cout, L.k, L.stride, L.pad)) {
        LWG_HIP(hipMalloc(&L.w_stem, kStemWBytes));
        LWG_HIP(hipMemset(L.w_stem, 0, kStemWBytes));
    }
    if (L.has_norm) {
        if ((rc = dev_alloc(&L.gamma, L.cout)) != LWG_OK) return rc;
        if ((rc = dev_alloc(&L.beta, L.cout)) != LWG_OK) return rc;
    }
    return LWG_OK;
}

void free_layer(Layer &L)
{
    if (L.w) (void)hipFree(L.w);
    if (L.w_split) (void)hipFree(L.w_split);
    if (L.w_stem) (void)hipFree(L.w_stem);
    L.w_split = nullptr;
    L.w_stem = nullptr;
    if (L.gamma) (void)hipFree(L.gamma);
    if (L.beta) (void)hipFree(L.beta);
    L.w = L.gamma = L.beta = nullptr;
}

int build_stream(StreamNet &s, int in_dim, int cd, int repeat, bool with_decoder, bool with_skip = true)
{
    s.has_skip = with_skip;
    init_conv(s.enc[0], in_dim, cd, 7, 1, 3);
    for (int i = 1; i <= kNDown; ++i) init_conv(s.enc[i], cd << (i - 1), cd << i, 3, 2, 1);
    const int ct = cd << kNDown;
    s.res.resize(2 * repeat);
    for (auto &L : s.res) init_conv(L, ct, ct, 3, 1, 1);
    int rc;
    for (int i = 0; i <= kNDown; ++i)
        if ((rc = alloc_layer(s.enc[i])) != LWG_OK) return rc;
    for (auto &L : s.res)
        if ((rc = alloc_layer(L)) != LWG_OK) return rc;
    if (with_decoder) {
        int cur = ct;
        for (int i = 0; i < kNDown; ++i) {
            init_convT(s.dec[i], cur, cur / 2);
            if ((rc = alloc_layer(s.dec[i])) != LWG_OK) return rc;
            if (with_skip) {
                init_conv(s.skip[i], cur, cur / 2, 3, 1, 1);
                if ((rc = alloc_layer(s.skip[i])) != LWG_OK) return rc;
            }
            cur /= 2;
        }
        if ((rc = dev_alloc(&s.heads_w, 49 * 64 * 4)) != LWG_OK) return rc;
        LWG_HIP(hipMemset(s.heads_w, 0, 49 * 64 * 4 * sizeof(float)));
        LWG_HIP(hipMalloc(&s.heads_frag, heads_bf16x3_frag_bytes()));
        s.heads_frag_stale = true;
    }
    return LWG_OK;
}

void free_stream(StreamNet &s)
{
    for (auto &L : s.enc) free_layer(L);
    for (auto &L : s.res) free_layer(L);
    for (auto &L : s.dec) free_layer(L);
    for (auto &L : s.skip) free_layer(L);
    if (s.heads_w) (void)hipFree(s.heads_w);
    s.heads_w = nullptr;
    if (s.heads_frag) (void)hipFree(s.heads_frag);
    s.heads_frag = nullptr;
}

// device copy of a re-laid-out weight matrix, plus the same matrix in the split-bf16 format (every Kpad is a
// multiple of 32, so the 32-value groups never straddle a row)
int upload_matrix(Layer &L, const std::vector<float> &h)
{
    LWG_HIP(hipMemcpy(L.w, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    if (L.w_split) {
        std::vector<float> sp(h.size());
        split_bf16_groups(h.data(), h.size(), sp.data());
        LWG_HIP(hipMemcpy(L.w_split, sp.data(), sp.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    L.got_w = true;
    return LWG_OK;
}

// PyTorch Conv2d weight (cout, cin, k, k) -> [cout][(kh*k+kw)*cin_pad + ci], zero padded to Kpad
int upload_conv(Layer &L, const float *w, const int64_t *shape, int ndim, const char *key)
{
    if (ndim != 4 || shape[0] != L.cout || shape[1] != L.cin || shape[2] != L.k || shape[3] != L.k)
        LWG_FAIL(LWG_ERR_INVALID_ARG, "%s: expected shape (%d,%d,%d,%d)", key, L.cout, L.cin, L.k, L.k);
    const ConvPhase &p = L.ph[0];
    std::vector<float> h((size_t)L.cout * p.Kpad, 0.f);
    for (int co = 0; co < L.cout; ++co)
        for (int ci = 0; ci < L.cin; ++ci)
            for (int t = 0; t < L.k * L.k; ++t)
                h[(size_t)co * p.Kpad + (size_t)t * L.cin_pad + ci] = w[((size_t)co * L.cin + ci) * L.k * L.k + t];
    if (L.w_stem) {
        std::vector<unsigned char> packed;
        stem_pack_weights(w, L.cin, packed);
        LWG_HIP(hipMemcpy(L.w_stem, packed.data(), packed.size(), hipMemcpyHostToDevice));
    }
    return upload_matrix(L, h);
}

// PyTorch ConvTranspose2d weight (cin, cout, 3, 3); out(oy) = sum in(iy) w[ky] with oy = 2*iy - 1 + ky.
// Phase parity 0 uses ky = 1 (iy = i); parity 1 uses ky = 2 (iy = i, tap offset 0) and ky = 0 (iy = i+1, offset 1).
int upload_convT(Layer &L, const float *w, const int64_t *shape, int ndim, const char *key)
{
    if (ndim != 4 || shape[0] != L.cin || shape[1] != L.cout || shape[2] != 3 || shape[3] != 3)
        LWG_FAIL(LWG_ERR_INVALID_ARG, "%s: expected shape (%d,%d,3,3)", key, L.cin, L.cout);
    std::vector<float> h(L.w_floats, 0.f);
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            const ConvPhase &p = L.ph[py * 2 + px];
            for (int ty = 0; ty < p.KH; ++ty)
                for (int tx = 0; tx < p.KW; ++tx) {
                    const int ky = py == 0 ? 1 : (ty == 0 ? 2 : 0);
                    const int kx = px == 0 ? 1 : (tx == 0 ? 2 : 0);
                    const int t = ty * p.KW + tx;
                    for (int co = 0; co < L.cout; ++co)
                        for (int ci = 0; ci < L.cin; ++ci)
                            h[(size_t)p.w_off + (size_t)co * p.Kpad + (size_t)t * L.cin + ci] =
                                w[(((size_t)ci * L.cout + co) * 3 + ky) * 3 + kx];
                }
        }
    return upload_matrix(L, h);
}

int upload_vec(float *dst, int n, bool *flag, const float *src, const int64_t *shape, int ndim, const char *key)
{
    if (ndim != 1 || shape[0] != n) LWG_FAIL(LWG_ERR_INVALID_ARG, "%s: expected shape (%d,)", key, n);
    LWG_HIP(hipMemcpy(dst, src, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    *flag = true;
    return LWG_OK;
}

// heads: img_reg (3,64,7,7) -> channels 0..2, attetion_reg (1,64,7,7) -> channel 3 of [49][64][4]
int upload_head(StreamNet &s, const float *w, const int64_t *shape, int ndim, int c0, int nc, int cd, const char *key)
{
    if (ndim != 4 || shape[0] != nc || shape[1] != cd || shape[2] != 7 || shape[3] != 7)
        LWG_FAIL(LWG_ERR_INVALID_ARG, "%s: expected shape (%d,%d,7,7)", key, nc, cd);
    std::vector<float> h(49 * 64 * 4);
    LWG_HIP(hipMemcpy(h.data(), s.heads_w, h.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int c = 0; c < nc; ++c)
        for (int ci = 0; ci < cd; ++ci)
            for (int t = 0; t < 49; ++t) h[((size_t)t * 64 + ci) * 4 + c0 + c] = w[((size_t)c * cd + ci) * 49 + t];
    LWG_HIP(hipMemcpy(s.heads_w, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    s.heads_frag_stale = true;
    return LWG_OK;
}

int missing_in(const StreamNet &s, bool with_decoder)
{
    int m = 0;
    auto need = [&](const Layer &L) { m += !L.got_w + (L.has_norm ? (!L.got_g + !L.got_b) : 0); };
    for (auto &L : s.enc) need(L);
    for (auto &L : s.res) need(L);
    if (with_decoder) {
        for (auto &L : s.dec) need(L);
        if (s.has_skip) {
            for (auto &L : s.skip) need(L);
            m += !s.got_att;
        }
        m += !s.got_img;
    }
    return m;
}

// Knock-out switches for timing experiments (builds with -DLWG_EXPERIMENTS only: `python -m impersonator_amd.build --experiments`
// writes _C/liblwg_exp.so; never in the shipped library).  LWG_KO = bit mask of launches to SKIP from the third pass of a handle on
// (buffers keep the previous pass's values, so the arithmetic downstream stays representative; RESULTS ARE WRONG): what a launch
// costs inside the two-lane pipeline = the bound on what optimising it can return.
//   1 in_finalize, 2 stem apply, 4 encoder 1-3 applies, 8 trunk applies, 16 heads, 32 stem conv, 64 skipper.2, 128 convT.2,
//   256 all other applies
#ifdef LWG_EXPERIMENTS
static int g_ko_passes = 0;
static bool ko(int bit)
{
    static const int mask = getenv("LWG_KO") ? atoi(getenv("LWG_KO")) : 0;
    return (mask & bit) && g_ko_passes >= 6;
}
#else
static constexpr bool ko(int) { return false; }
#endif

// ---- profiling helpers
int prof_begin(lwg_generator *g, hipStream_t st, hipEvent_t *e1)
{
    if (g->ev_used + 2 > g->ev_pool.size()) {
        for (int i = 0; i < 64; ++i) {
            hipEvent_t e;
            LWG_HIP(hipEventCreate(&e));
            g->ev_pool.push_back(e);
        }
    }
    LWG_HIP(hipEventRecord(g->ev_pool[g->ev_used], st));
    *e1 = g->ev_pool[g->ev_used + 1];
    g->ev_used += 2;
    return LWG_OK;
}

// the zero run behind the handle-owned activation buffer `x` points into (null for anything else)
const float *zero_tail_of(const lwg_generator *g, const float *x)
{
    for (int l = 0; l < kNDown; ++l) {
        if (g->cat[l] && x >= g->cat[l] && x < g->cat[l] + g->cat_floats[l]) return g->cat[l] + g->cat_floats[l];
        if (g->sk[l] && x >= g->sk[l] && x < g->sk[l] + g->sk_floats[l]) return g->sk[l] + g->sk_floats[l];
    }
    for (int i = 0; i < 3; ++i)
        if (g->trunk[i] && x >= g->trunk[i] && x < g->trunk[i] + g->trunk_floats) return g->trunk[i] + g->trunk_floats;
    return nullptr;
}

// launch arguments of one conv layer: `raw` / `ld_raw` = where the raw (pre-norm) output goes and its pixel stride (0: dense);
// raw_from >= 0: the input's channels from there on are the PRODUCER's raw output, normalised by the conv itself with the
// (scale, shift) currently in g->ss (ConvArgs::raw_in)
int conv_args(lwg_generator *g, const Layer &L, const float *x, int ldx, int N, int H, int W, float *raw, int ld_raw, int raw_from,
              ConvArgs &a, int &bn)
{
    a = ConvArgs{};
    a.x = x;
    a.ldx = ldx;
    a.N = N;
    a.H = H;
    a.W = W;
    a.Cin = L.cin_pad;
    a.cin_log2 = ilog2(L.cin_pad);
    a.w = L.w;
    a.w_split = L.w_split;
    a.precision = (g->split && L.w_split) ? 1 : 0;   // the 7x7 stem (Cin 6, fp32 NHWC8 input) stays on the fp32 kernel
    static const bool natural = getenv("LWG_NATURAL_TILE_ORDER") != nullptr;   // A/B switch for measurements
    a.natural_order = natural ? 1 : 0;
    static const char *k_order = getenv("LWG_K_ORDER");                          // "channel": the round-2 walk (A/B switch)
    a.tap_inner = (k_order && k_order[0] == 'c') ? 0 : 1;
    a.zeros = g->zeros;
    if (a.precision == 1) {
        a.zeros = zero_tail_of(g, x);
        if (!a.zeros) LWG_FAIL(LWG_ERR_STATE, "bf16x3 conv input is not one of the handle's activation buffers");
    }
    a.y = raw;
    a.ldy = ld_raw > 0 ? ld_raw : L.cout;
    a.Cout = L.cout;
    if (raw_from >= 0) {
        a.raw_in = 1;
        a.raw_from = raw_from;
        a.in_ss = g->ss;
        a.in_ss_ld = L.cin_pad - raw_from;
    }
    a.nphase = L.nphase;
    a.dil = 1;
    for (int p = 0; p < L.nphase; ++p) a.ph[p] = L.ph[p];
    if (L.transposed) {
        a.Hm = H; a.Wm = W; a.stride = 1; a.pad = 0; a.os = 2; a.Ho = 2 * H; a.Wo = 2 * W;
    } else {
        a.Hm = (H + 2 * L.pad - L.k) / L.stride + 1;
        a.Wm = (W + 2 * L.pad - L.k) / L.stride + 1;
        a.stride = L.stride; a.pad = L.pad; a.os = 1; a.Ho = a.Hm; a.Wo = a.Wm;
    }
    a.mtiles = N * a.Hm * a.Wm / kConvBM;
    a.partials = L.has_norm ? g->partials : nullptr;
    // tile width: 128 channels once that still gives every CU a workgroup (the 32x32 trunk at batch 8 is exactly 256
    // tiles), else 64 for more, smaller tiles (small batches).  Transposed convs: 64-channel tiles with the four
    // phases walked inside the workgroup (equal work per workgroup).
    const long tiles128 = (long)a.mtiles * (L.cout / 128) * L.nphase;
    bn = (L.cout % 128 == 0 && tiles128 >= 256) ? 128 : 64;
    // exact fp32 with few tiles (one source's encoder, one frame per call): 32-channel tiles while the launch stays under one
    // workgroup per CU -- the 512 -> 512 layer on a 32 x 32 map is 64 workgroups of the 64-channel tile (env LWG_F32_BN32=0: off)
    static const char *bn32_env = getenv("LWG_F32_BN32");
    if (bn == 64 && a.precision == 0 && !L.transposed && L.cin_pad >= kConvBK && L.k * L.k <= 32 && L.cout % 32 == 0 &&
        (long)a.mtiles * (L.cout / 64) * L.nphase * 2 <= 256 && !(bn32_env && bn32_env[0] == '0'))
        bn = 32;
    if (L.transposed) {
        if (a.precision == 1 && L.cout % 128 == 0) {
            // bf16x3: the 128-channel tile is ~1.4x faster than the 64-channel one; phases become separate workgroups,
            // dispatched heaviest first (4, 2, 2, 1 taps), which packs 256 CUs nearly as well as fusing them
            bn = 128;
            a.fuse_phases = 0;
        } else {
            bn = 64;
            a.fuse_phases = 1;
        }
    }
    return LWG_OK;
}

// would layer L, reading `x`, take raw input channels from `raw_from` on?  (the halo-resident bf16x3 kernels do)
bool takes_raw_input(lwg_generator *g, const Layer &L, const float *x, int ldx, int N, int H, int W, int raw_from)
{
    if (!g->split || !L.w_split) return false;
    ConvArgs a;
    int bn = 0;
    if (conv_args(g, L, x, ldx, N, H, W, g->raw, 0, raw_from, a, bn) != LWG_OK) return false;
    return conv_raw_input_supported(a, bn);
}

// one normalised conv layer: conv -> statistics -> (caller applies, or the consumer does: see conv_args)
int run_conv(lwg_generator *g, const Layer &L, const float *x, int ldx, int N, int H, int W, float *raw, hipStream_t st,
             int ld_raw = 0, int raw_from = -1)
{
    ConvArgs a;
    int bn = 0;
    {
        const int rc0 = conv_args(g, L, x, ldx, N, H, W, raw, ld_raw, raw_from, a, bn);
        if (rc0 != LWG_OK) return rc0;
    }

    hipEvent_t e1 = nullptr;
    if (g->profile) {
        const int rc = prof_begin(g, st, &e1);
        if (rc != LWG_OK) return rc;
    }
    int variant = 0;
    int rc;
    if (g->split && L.w_stem && ldx == 8 && stem_bf16x3_supported(H, W, L.cin, L.cin_pad, L.cout, L.k, L.stride, L.pad)) {
        StemArgs s = {};
        s.x = x; s.N = N; s.H = H; s.W = W;
        s.w = L.w_stem;
        s.y = raw;
        s.partials = a.partials;
        rc = launch_stem_bf16x3(s, st);
        variant = kDirectStemBf16x3;
    } else {
        rc = launch_conv_igemm(a, bn, st, &variant);
    }
    if (rc != LWG_OK) return rc;
    if (g->profile) {
        LWG_HIP(hipEventRecord(e1, st));
        double k_alg = 0;
        for (int p = 0; p < L.nphase; ++p) k_alg += (double)L.ph[p].ntaps * L.cin;
        g->prof_flops[variant] += 2.0 * N * a.Hm * a.Wm * (double)L.cout * k_alg;
        g->prof_launches[variant] += 1;
        g->ev_variant.push_back(variant);
    }
    if (L.has_norm && !ko(1)) {
        rc = launch_in_finalize(g->partials, L.nphase, a.mtiles, N, L.cout, L.gamma, L.beta, kInEps, g->ss, st);
        if (rc != LWG_OK) return rc;
    }
    return LWG_OK;
}

struct Warp {
    const float *src = nullptr;  // NHWC (n,H,W,C)
    const float *T = nullptr;    // resized flow (N,H,W,2)
    int n = 1;                   // 1: one source shared by the batch (inference); N: a source per sample (infer_front)
};

int run_apply(lwg_generator *g, int N, int H, int W, int C, bool relu, float *dst, int ld_dst, const float *res,
              int ld_res, const Warp *warps, int nwarp, int align, hipStream_t st)
{
    ApplyArgs a = {};
    a.raw = g->raw;
    a.C = C;
    a.N = N;
    a.H = H;
    a.W = W;
    a.scale_shift = g->ss;
    a.relu = relu;
    a.dst = dst;
    a.ld_dst = ld_dst;
    a.res = res;
    a.ld_res = ld_res;
    a.nwarp = nwarp;
    for (int k = 0; k < nwarp; ++k) {
        a.warp_src[k] = warps[k].src;
        a.warp_n[k] = warps[k].n;
        a.warp_T[k] = warps[k].T;
    }
    a.align_corners = align;
    a.split = g->split ? 1 : 0;
    return launch_apply(a, st);
}

// cpad = the consuming stem's padded input channels (Layer::cin_pad)
int pack_input(lwg_generator *g, const float *x, int layout, int N, int C, int cpad, hipStream_t st, const float **out)
{
    if (layout == 1) {
        if (cpad != 8) LWG_FAIL(LWG_ERR_INVALID_ARG, "layout 1 (NHWC8) needs an input of at most 8 channels, this stream has %d", C);
        *out = x;
        return LWG_OK;
    }
    if (layout != 0) LWG_FAIL(LWG_ERR_INVALID_ARG, "layout must be 0 (NCHW) or 1 (NHWC8)");
    if (cpad > g->x0_c) LWG_FAIL(LWG_ERR_STATE, "packed-input buffer holds %d channels, %d needed", g->x0_c, cpad);
    const int rc = lwg_pack_nhwc(x, N, C, g->is, g->is, cpad, g->x0, st);
    *out = g->x0;
    return rc;
}

// encoder level `lvl` of one stream; dst/ld_dst is where the activated output goes
int run_encoder(lwg_generator *g, const StreamNet &s, int lvl, const float *x, int ldx, int N, float *dst, int ld_dst,
                const Warp *warps, int nwarp, int align, hipStream_t st)
{
    const Layer &L = s.enc[lvl];
    const int Hin = lvl == 0 ? g->is : g->is >> (lvl - 1);
    int rc = (lvl == 0 && &s == &g->tsf && ko(32)) ? LWG_OK : run_conv(g, L, x, ldx, N, Hin, Hin, g->raw, st);
    if (rc != LWG_OK) return rc;
    const int Ho = g->is >> lvl;
    if (&s == &g->tsf && ko(lvl == 0 ? 2 : 4)) return LWG_OK;
    return run_apply(g, N, Ho, Ho, L.cout, true, dst, ld_dst, nullptr, 0, warps, nwarp, align, st);
}

// ResidualBlock i (generator.py:8-20): x + IN(conv(ReLU(IN(conv(x))))) [+ warps]; xin -> xout
int run_resblock(lwg_generator *g, const StreamNet &s, int i, const float *xin, float *xout, int N,
                 const Warp *warps, int nwarp, int align, hipStream_t st)
{
    const int h = g->is >> kNDown, C = g->cd << kNDown;
    int rc;
    if (takes_raw_input(g, s.res[2 * i + 1], g->trunk[2], C, N, h, h, 0)) {
        // the first conv's InstanceNorm + ReLU is applied by the second conv as it loads its halo: no pass over memory in between
        if ((rc = run_conv(g, s.res[2 * i], xin, C, N, h, h, g->trunk[2], st, C)) != LWG_OK) return rc;
        if ((rc = run_conv(g, s.res[2 * i + 1], g->trunk[2], C, N, h, h, g->raw, st, 0, 0)) != LWG_OK) return rc;
    } else {
        if ((rc = run_conv(g, s.res[2 * i], xin, C, N, h, h, g->raw, st)) != LWG_OK) return rc;
        if ((rc = run_apply(g, N, h, h, C, true, g->trunk[2], C, nullptr, 0, nullptr, 0, align, st)) != LWG_OK) return rc;
        if ((rc = run_conv(g, s.res[2 * i + 1], g->trunk[2], C, N, h, h, g->raw, st)) != LWG_OK) return rc;
    }
    if (&s == &g->tsf && ko(8)) return LWG_OK;
    return run_apply(g, N, h, h, C, false, xout, C, xin, C, warps, nwarp, align, st);
}

int check_ready(const lwg_generator *g, int bs)
{
    if (!g) LWG_FAIL(LWG_ERR_INVALID_ARG, "NULL generator handle");
    if (bs <= 0 || bs > g->max_batch) LWG_FAIL(LWG_ERR_STATE, "batch %d outside 1..max_batch=%d", bs, g->max_batch);
    const int m = missing_in(g->src, false) + missing_in(g->tsf, true);
    if (m) LWG_FAIL(LWG_ERR_STATE, "%d weight tensors have not been loaded", m);
    return LWG_OK;
}

// the tsf stream shared by inference (one warp set) and swap (two)
int run_tsf(lwg_generator *g, const float *tsf_inputs, int layout, const float *const T[2],
            const float *const *feats[2], int nsets, int bs, int align, float *color, float *mask, const float *bg,
            int bg_bs, float *pred, hipStream_t st, int feats_bs = 1)
{
    const StreamNet &s = g->tsf;
    const int is = g->is, cd = g->cd;
    int rc;
    g->split = g->last_split = g->precision == 1;   // every activation buffer of this pass is in that format
    for (int k = 0; k < nsets; ++k)
        for (int l = 1; l <= kNDown; ++l)
            if ((rc = lwg_resize_flow(T[k], bs, is, is, is >> l, is >> l, g->tscale[k][l - 1], st)) != LWG_OK) return rc;

    const float *x0 = nullptr;
    if ((rc = pack_input(g, tsf_inputs, layout, bs, g->tsf_dim, s.enc[0].cin_pad, st, &x0)) != LWG_OK) return rc;

    // encoders: level l output lives in the first half of cat[l] (the decoder's skip operand), level 3 in the trunk
    if ((rc = run_encoder(g, s, 0, x0, s.enc[0].cin_pad, bs, g->cat[0], 2 * cd, nullptr, 0, align, st)) != LWG_OK) return rc;
    for (int l = 1; l <= kNDown; ++l) {
        Warp w[2];
        for (int k = 0; k < nsets; ++k) {
            w[k].src = feats[k][l];
            w[k].T = g->tscale[k][l - 1];
            w[k].n = feats_bs;
        }
        const float *xin = g->cat[l - 1];
        const int ldx = 2 * (cd << (l - 1));
        float *dst = l < kNDown ? g->cat[l] : g->trunk[0];
        const int ld_dst = l < kNDown ? 2 * (cd << l) : (cd << l);
        if ((rc = run_encoder(g, s, l, xin, ldx, bs, dst, ld_dst, w, nsets, align, st)) != LWG_OK) return rc;
    }
    // residual trunk (generator.py:291-295): the 32x32 flow is shared by all blocks
    int cur = 0;
    for (int i = 0; i < g->repeat; ++i) {
        Warp w[2];
        for (int k = 0; k < nsets; ++k) {
            w[k].src = feats[k][kNDown + 1 + i];
            w[k].T = g->tscale[k][kNDown - 1];
            w[k].n = feats_bs;
        }
        if ((rc = run_resblock(g, s, i, g->trunk[cur], g->trunk[cur ^ 1], bs, w, nsets, align, st)) != LWG_OK) return rc;
        cur ^= 1;
    }
    // decoder (generator.py:173-181): convT -> second half of cat[level]; skipper conv over the whole cat buffer
    g->trunk_out = cur;
    const float *d = g->trunk[cur];
    int dC = cd << kNDown, dH = is >> kNDown;
    bool d_raw = false;   // d holds the previous skipper's RAW output (its InstanceNorm + ReLU pending in g->ss)
    for (int i = 0; i < kNDown; ++i) {
        const int lvl = kNDown - 1 - i, oC = dC / 2, oH = dH * 2;
        const bool last = i + 1 == kNDown;
        // Where the consumer is the halo-resident bf16x3 kernel, a layer's InstanceNorm + ReLU is applied by that consumer as it
        // loads its halo (ConvArgs::raw_in): the producer writes its raw output straight into the consumer's input buffer and
        // no apply pass runs in between.  Transposed conv -> second half of cat[level]:
        const bool skip_takes_raw = takes_raw_input(g, s.skip[i], g->cat[lvl], 2 * oC, bs, oH, oH, oC);
        if (!(last && ko(128)) &&
            (rc = run_conv(g, s.dec[i], d, dC, bs, dH, dH, skip_takes_raw ? g->cat[lvl] + oC : g->raw, st, skip_takes_raw ? 2 * oC : 0,
                           d_raw ? 0 : -1)) != LWG_OK)
            return rc;
        if (!skip_takes_raw &&
            (rc = run_apply(g, bs, oH, oH, oC, true, g->cat[lvl] + oC, 2 * oC, nullptr, 0, nullptr, 0, align, st)) != LWG_OK)
            return rc;
        // skipper conv over the whole cat buffer -> the next level's input (the last one's raw output feeds the heads)
        const bool next_takes_raw = !last && takes_raw_input(g, s.dec[i + 1], g->sk[i], oC, bs, oH, oH, 0);
        if (!(last && ko(64)) &&
            (rc = run_conv(g, s.skip[i], g->cat[lvl], 2 * oC, bs, oH, oH, next_takes_raw ? g->sk[i] : g->raw, st, 0,
                           skip_takes_raw ? oC : -1)) != LWG_OK)
            return rc;
        if (!last) {
            if (!next_takes_raw &&
                (rc = run_apply(g, bs, oH, oH, oC, true, g->sk[i], oC, nullptr, 0, nullptr, 0, align, st)) != LWG_OK)
                return rc;
            d = g->sk[i];
            d_raw = next_takes_raw;
        }
        dC = oC;
        dH = oH;
    }
    // heads read the last skipper's raw output and fold its InstanceNorm+ReLU into their halo load
    HeadsArgs h = {};
    h.x = g->raw;
    h.N = bs;
    h.H = is;
    h.W = is;
    h.scale_shift = g->ss;
    h.wh = s.heads_w;
    h.color = color;
    h.mask = mask;
    h.bg = bg;
    h.bg_bs = bg_bs;
    h.pred = pred;
#ifdef LWG_EXPERIMENTS
    ++g_ko_passes;
#endif
    if (ko(16)) return LWG_OK;
    if (!g->split) return launch_heads(h, st);   // exact fp32 on the vector ALU
    if (g->tsf.heads_frag_stale) {
        if ((rc = launch_heads_pack(s.heads_w, s.heads_frag, st)) != LWG_OK) return rc;
        g->tsf.heads_frag_stale = false;
    }
    return launch_heads_bf16x3(h, s.heads_frag, st);
}

}  // namespace
}  // namespace lwg

extern "C" {

int lwg_generator_create(lwg_generator **out, int src_dim, int tsf_dim, int conv_dim, int repeat_num, int image_size,
                         int max_batch)
{
    LWG_REQUIRE(out, "generator_create: NULL out");
    *out = nullptr;
    LWG_REQUIRE(src_dim > 0 && src_dim <= 32 && tsf_dim > 0 && tsf_dim <= 32, "generator_create: input dims must be 1..32");
    LWG_REQUIRE(repeat_num > 0 && max_batch > 0, "generator_create: repeat_num/max_batch must be positive");
    if (conv_dim != 64)
        LWG_FAIL(LWG_ERR_UNSUPPORTED, "generator_create: conv_dim=%d (the kernels are built for the reference's 64)", conv_dim);
    const int hb = image_size >> kNDown;
    if (image_size <= 0 || (image_size & 7) || (hb * hb) % kConvBM != 0)
        LWG_FAIL(LWG_ERR_UNSUPPORTED, "generator_create: image_size=%d; (image_size/8)^2 must be a multiple of %d",
                 image_size, kConvBM);
    int ndev = 0;
    LWG_HIP(hipGetDeviceCount(&ndev));

    lwg_generator *g = new lwg_generator();
    g->src_dim = src_dim;
    g->tsf_dim = tsf_dim;
    g->cd = conv_dim;
    g->repeat = repeat_num;
    g->is = image_size;
    g->max_batch = max_batch;
    int rc = build_stream(g->src, src_dim, conv_dim, repeat_num, true);   // decoder: infer_front / forward only
    if (rc == LWG_OK) rc = build_stream(g->tsf, tsf_dim, conv_dim, repeat_num, true);

    const size_t B = (size_t)max_batch, P = (size_t)image_size * image_size;
    const int cd = conv_dim;
    if (rc == LWG_OK) {
        g->x0_c = g->src.enc[0].cin_pad > g->tsf.enc[0].cin_pad ? g->src.enc[0].cin_pad : g->tsf.enc[0].cin_pad;
        rc = dev_alloc(&g->x0, B * P * g->x0_c);
    }
    if (rc == LWG_OK) rc = dev_alloc(&g->raw, B * P * cd);
    for (int l = 0; l < kNDown && rc == LWG_OK; ++l) rc = act_alloc(&g->cat[l], g->cat_floats[l] = B * (P >> (2 * l)) * (size_t)(2 * (cd << l)));
    for (int i = 0; i < 3 && rc == LWG_OK; ++i) rc = act_alloc(&g->trunk[i], g->trunk_floats = B * (P >> (2 * kNDown)) * (size_t)(cd << kNDown));
    for (int i = 0; i + 1 < kNDown && rc == LWG_OK; ++i) {
        const int lvl = kNDown - 1 - i;
        rc = act_alloc(&g->sk[i], g->sk_floats[i] = B * (P >> (2 * lvl)) * (size_t)(cd << lvl));
    }
    for (int k = 0; k < 2; ++k)
        for (int l = 1; l <= kNDown && rc == LWG_OK; ++l) rc = dev_alloc(&g->tscale[k][l - 1], B * (P >> (2 * l)) * 2);
    // statistics partials: the widest need is 4 phases x tiles x channels of a decoder conv-transpose
    if (rc == LWG_OK) {
        size_t need = 0;
        for (int l = 0; l <= kNDown; ++l) {
            const size_t tiles = B * (P >> (2 * l)) / kConvBM;
            const size_t c = (size_t)(cd << l);
            need = need > tiles * c ? need : tiles * c;
        }
        float *p = nullptr;
        rc = dev_alloc(&p, need * 2);
        g->partials = reinterpret_cast<float2 *>(p);
    }
    if (rc == LWG_OK) {
        float *p = nullptr;
        rc = dev_alloc(&p, B * (size_t)(cd << kNDown) * 2);
        g->ss = reinterpret_cast<float2 *>(p);
    }
    if (rc == LWG_OK) rc = dev_alloc(&g->zeros, 64);
    if (rc == LWG_OK && hipMemset(g->zeros, 0, 64 * sizeof(float)) != hipSuccess) {
        set_error("hipMemset of the zero page failed");
        rc = LWG_ERR_HIP;
    }
    if (rc != LWG_OK) {
        lwg_generator_destroy(g);
        return rc;
    }
    *out = g;
    return LWG_OK;
}

void lwg_generator_destroy(lwg_generator *g)
{
    if (!g) return;
    free_stream(g->src);
    free_stream(g->tsf);
    free_stream(g->bg);
    auto fr = [](void *p) { if (p) (void)hipFree(p); };
    fr(g->x0);
    fr(g->raw);
    for (auto p : g->cat) fr(p);
    for (auto p : g->trunk) fr(p);
    for (auto p : g->sk) fr(p);
    for (auto &k : g->tscale)
        for (auto p : k) fr(p);
    fr(g->partials);
    fr(g->ss);
    fr(g->zeros);
    for (auto e : g->ev_pool) (void)hipEventDestroy(e);
    delete g;
}

int lwg_generator_load_weight(lwg_generator *g, const char *key, const float *data_host, const int64_t *shape, int ndim)
{
    LWG_REQUIRE(g && key && data_host && shape, "load_weight: NULL argument");
    std::string k(key);
    if (k.rfind("module.", 0) == 0) k = k.substr(7);  // DataParallel prefix (models/models.py:163-171)
    if (k.rfind("bg_model.", 0) == 0) {
        if (!g->bg_dim) {
            g->ignored_keys++;
            return LWG_OK;  // BGNet only runs for --bg_model ORIGINAL (models/imitator.py:30-34): lwg_generator_enable_bg
        }
        // ResNetGenerator.model is one nn.Sequential (generator.py:29-58): [conv7, IN, ReLU], n_down x [conv, IN, ReLU],
        // repeat x ResidualBlock, n_down x [convT, IN, ReLU], conv7, tanh
        StreamNet &b = g->bg;
        int idx = -1, sub = -1;
        char tail[32] = {0};
        const int res0 = 3 + 3 * kNDown, up0 = res0 + g->repeat, fin = up0 + 3 * kNDown;
        auto put = [&](Layer &L, bool norm, const char *what) -> int {
            if (!norm && !strcmp(what, "weight"))
                return L.transposed ? upload_convT(L, data_host, shape, ndim, key) : upload_conv(L, data_host, shape, ndim, key);
            if (norm && !strcmp(what, "weight")) return upload_vec(L.gamma, L.cout, &L.got_g, data_host, shape, ndim, key);
            if (norm && !strcmp(what, "bias")) return upload_vec(L.beta, L.cout, &L.got_b, data_host, shape, ndim, key);
            LWG_FAIL(LWG_ERR_INVALID_ARG, "load_weight: unknown key '%s'", key);
        };
        if (sscanf(k.c_str(), "bg_model.model.%d.main.%d.%31s", &idx, &sub, tail) == 3 && idx >= res0 && idx < up0) {
            const int i = idx - res0;
            if (sub == 0 || sub == 1) return put(b.res[2 * i], sub == 1, tail);
            if (sub == 3 || sub == 4) return put(b.res[2 * i + 1], sub == 4, tail);
            LWG_FAIL(LWG_ERR_INVALID_ARG, "load_weight: unknown key '%s'", key);
        }
        if (sscanf(k.c_str(), "bg_model.model.%d.%31s", &idx, tail) == 2) {
            if (idx == fin && !strcmp(tail, "weight")) {
                const int rc = upload_head(b, data_host, shape, ndim, 0, 3, g->cd, key);
                if (rc == LWG_OK) b.got_img = true;
                return rc;
            }
            if (idx >= 0 && idx < res0) return put(b.enc[idx / 3], idx % 3 == 1, tail);
            if (idx >= up0 && idx < fin) return put(b.dec[(idx - up0) / 3], (idx - up0) % 3 == 1, tail);
        }
        LWG_FAIL(LWG_ERR_INVALID_ARG, "load_weight: unknown key '%s'", key);
    }
    StreamNet *s = nullptr;
    bool is_tsf = false;
    if (k.rfind("src_model.", 0) == 0) s = &g->src;
    else if (k.rfind("tsf_model.", 0) == 0) { s = &g->tsf; is_tsf = true; }
    else LWG_FAIL(LWG_ERR_INVALID_ARG, "load_weight: unknown key '%s'", key);
    k = k.substr(10);

    int i = -1, j = -1;
    char tail[32] = {0};
    auto norm_or_conv = [&](Layer &L, int sub, const char *what) -> int {
        // <block>.<i>.0.weight = conv ; <block>.<i>.1.{weight,bias} = InstanceNorm affine
        if (sub == 0 && !strcmp(what, "weight"))
            return L.transposed ? upload_convT(L, data_host, shape, ndim, key) : upload_conv(L, data_host, shape, ndim, key);
        if (sub == 1 && !strcmp(what, "weight")) return upload_vec(L.gamma, L.cout, &L.got_g, data_host, shape, ndim, key);
        if (sub == 1 && !strcmp(what, "bias")) return upload_vec(L.beta, L.cout, &L.got_b, data_host, shape, ndim, key);
        LWG_FAIL(LWG_ERR_INVALID_ARG, "load_weight: unknown key '%s'", key);
    };
    if (sscanf(k.c_str(), "encoders.%d.%d.%31s", &i, &j, tail) == 3 && i >= 0 && i <= kNDown)
        return norm_or_conv(s->enc[i], j, tail);
    if (sscanf(k.c_str(), "resnets.%d.main.%d.%31s", &i, &j, tail) == 3 && i >= 0 && i < g->repeat) {
        // main = [conv, IN, ReLU, conv, IN] (generator.py:12-17)
        if (j == 0 || j == 1) return norm_or_conv(s->res[2 * i], j, tail);
        if (j == 3 || j == 4) return norm_or_conv(s->res[2 * i + 1], j - 3, tail);
        LWG_FAIL(LWG_ERR_INVALID_ARG, "load_weight: unknown key '%s'", key);
    }
    const bool dec = sscanf(k.c_str(), "decoders.%d.%d.%31s", &i, &j, tail) == 3;
    const bool skp = !dec && sscanf(k.c_str(), "skippers.%d.%d.%31s", &i, &j, tail) == 3;
    if ((dec || skp) && i >= 0 && i < kNDown) {
        return norm_or_conv(dec ? s->dec[i] : s->skip[i], j, tail);
    }
    if (k == "img_reg.0.weight" || k == "attetion_reg.0.weight") {
        const bool img = k[0] == 'i';
        const int rc = upload_head(*s, data_host, shape, ndim, img ? 0 : 3, img ? 3 : 1, g->cd, key);
        if (rc == LWG_OK) (img ? s->got_img : s->got_att) = true;
        return rc;
    }
    LWG_FAIL(LWG_ERR_INVALID_ARG, "load_weight: unknown key '%s'", key);
}

int lwg_generator_missing_weights(const lwg_generator *g)
{
    if (!g) return -1;
    return missing_in(g->src, false) + missing_in(g->tsf, true);
}

int lwg_generator_num_src_features(const lwg_generator *g) { return g ? kNDown + 1 + g->repeat : 0; }

int lwg_generator_src_feature_shape(const lwg_generator *g, int index, int *C, int *H, int *W)
{
    LWG_REQUIRE(g && C && H && W, "src_feature_shape: NULL argument");
    LWG_REQUIRE(index >= 0 && index < kNDown + 1 + g->repeat, "src_feature_shape: index %d out of range", index);
    const int lvl = index <= kNDown ? index : kNDown;
    *C = g->cd << lvl;
    *H = *W = g->is >> lvl;
    return LWG_OK;
}

int lwg_generator_encode_src(lwg_generator *g, const float *src_inputs_nchw, float *const *feats_nhwc,
                             lwg_stream_t stream)
{
    return lwg_generator_encode_src_n(g, src_inputs_nchw, 1, feats_nhwc, stream);
}

int lwg_generator_encode_src_n(lwg_generator *g, const float *src_inputs_nchw, int bs, float *const *feats_nhwc,
                               lwg_stream_t stream)
{
    int rc = check_ready(g, bs);
    if (rc != LWG_OK) return rc;
    LWG_REQUIRE(src_inputs_nchw && feats_nhwc, "encode_src: NULL argument");
    for (int i = 0; i < kNDown + 1 + g->repeat; ++i) LWG_REQUIRE(feats_nhwc[i], "encode_src: feats_nhwc[%d] is NULL", i);
    hipStream_t st = as_stream(stream);
    // once per source, and its outputs are fp32 tensors handed to the caller (the LWB gathers read them): always fp32
    g->split = false;
    const float *x0 = nullptr;
    if ((rc = pack_input(g, src_inputs_nchw, 0, bs, g->src_dim, g->src.enc[0].cin_pad, st, &x0)) != LWG_OK) return rc;
    const float *x = x0;
    int ldx = g->src.enc[0].cin_pad;
    for (int l = 0; l <= kNDown; ++l) {
        const int C = g->cd << l;
        if ((rc = run_encoder(g, g->src, l, x, ldx, bs, feats_nhwc[l], C, nullptr, 0, 0, st)) != LWG_OK) return rc;
        x = feats_nhwc[l];
        ldx = C;
    }
    for (int i = 0; i < g->repeat; ++i) {
        float *o = feats_nhwc[kNDown + 1 + i];
        if ((rc = run_resblock(g, g->src, i, x, o, bs, nullptr, 0, 0, st)) != LWG_OK) return rc;
        x = o;
    }
    return LWG_OK;
}

// src_model.regress(src_model.decode(x, encoder_outs)) of infer_front (generator.py:238; ResUnetGenerator.decode /
// regress :163-184) on features produced by encode_src_n: feats[0..2] are the skip operands, feats[last] the trunk output.
int lwg_generator_decode_src(lwg_generator *g, const float *const *feats_nhwc, int bs, float *color, float *mask,
                             lwg_stream_t stream)
{
    int rc = check_ready(g, bs);
    if (rc != LWG_OK) return rc;
    LWG_REQUIRE(feats_nhwc && color && mask, "decode_src: NULL argument");
    const int miss = missing_in(g->src, true);
    if (miss) LWG_FAIL(LWG_ERR_STATE, "%d source-stream weight tensors (decoder / heads) have not been loaded", miss);
    hipStream_t st = as_stream(stream);
    const StreamNet &s = g->src;
    const int is = g->is, cd = g->cd;
    g->split = g->last_split = false;
    // skip operands into the first halves of the concat buffers (torch.cat([skip, d]) by layout)
    for (int l = 0; l < kNDown; ++l) {
        LWG_REQUIRE(feats_nhwc[l], "decode_src: feats_nhwc[%d] is NULL", l);
        const size_t C = (size_t)cd << l, P = (size_t)bs * (is >> l) * (is >> l);
        LWG_HIP(hipMemcpy2DAsync(g->cat[l], 2 * C * sizeof(float), feats_nhwc[l], C * sizeof(float), C * sizeof(float), P,
                                 hipMemcpyDeviceToDevice, st));
    }
    const float *d = feats_nhwc[kNDown + g->repeat];
    LWG_REQUIRE(d, "decode_src: the last residual feature is NULL");
    int dC = cd << kNDown, dH = is >> kNDown;
    for (int i = 0; i < kNDown; ++i) {
        const int lvl = kNDown - 1 - i, oC = dC / 2, oH = dH * 2;
        if ((rc = run_conv(g, s.dec[i], d, dC, bs, dH, dH, g->raw, st)) != LWG_OK) return rc;
        if ((rc = run_apply(g, bs, oH, oH, oC, true, g->cat[lvl] + oC, 2 * oC, nullptr, 0, nullptr, 0, 0, st)) != LWG_OK) return rc;
        if ((rc = run_conv(g, s.skip[i], g->cat[lvl], 2 * oC, bs, oH, oH, g->raw, st)) != LWG_OK) return rc;
        if (i + 1 < kNDown) {
            if ((rc = run_apply(g, bs, oH, oH, oC, true, g->sk[i], oC, nullptr, 0, nullptr, 0, 0, st)) != LWG_OK) return rc;
            d = g->sk[i];
        }
        dC = oC;
        dH = oH;
    }
    HeadsArgs h = {};
    h.x = g->raw;
    h.N = bs;
    h.H = is;
    h.W = is;
    h.scale_shift = g->ss;
    h.wh = s.heads_w;
    h.color = color;
    h.mask = mask;
    return launch_heads(h, st);
}

int lwg_generator_enable_bg(lwg_generator *g, int bg_dim)
{
    LWG_REQUIRE(g, "enable_bg: NULL handle");
    LWG_REQUIRE(bg_dim >= 1 && bg_dim <= 8, "enable_bg: bg_dim must be 1..8");
    if (g->bg_dim) {
        if (g->bg_dim != bg_dim) LWG_FAIL(LWG_ERR_STATE, "enable_bg: already enabled with bg_dim=%d", g->bg_dim);
        return LWG_OK;
    }
    const int rc = build_stream(g->bg, bg_dim, g->cd, g->repeat, true, false);
    if (rc == LWG_OK) g->bg_dim = bg_dim;
    return rc;
}

// BGNet, ResNetGenerator.forward (generator.py:60-65): bg_inputs (bs, bg_dim, is, is) NCHW -> (bs, 3, is, is), fp32.
// Runs on the tsf stream's scratch buffers (dense, no skip halves), so it must not overlap an inference call.
int lwg_generator_bg_forward(lwg_generator *g, const float *bg_inputs_nchw, int bs, float *out_nchw, lwg_stream_t stream)
{
    LWG_REQUIRE(g && bg_inputs_nchw && out_nchw, "bg_forward: NULL argument");
    if (!g->bg_dim) LWG_FAIL(LWG_ERR_STATE, "bg_forward: BGNet not enabled (lwg_generator_enable_bg)");
    if (bs <= 0 || bs > g->max_batch) LWG_FAIL(LWG_ERR_STATE, "batch %d outside 1..max_batch=%d", bs, g->max_batch);
    const int miss = missing_in(g->bg, true);
    if (miss) LWG_FAIL(LWG_ERR_STATE, "%d BGNet weight tensors have not been loaded", miss);
    hipStream_t st = as_stream(stream);
    const StreamNet &s = g->bg;
    const int is = g->is, cd = g->cd;
    g->split = false;
    int rc;
    const float *x = nullptr;
    if ((rc = pack_input(g, bg_inputs_nchw, 0, bs, g->bg_dim, s.enc[0].cin_pad, st, &x)) != LWG_OK) return rc;
    int ldx = s.enc[0].cin_pad;
    for (int l = 0; l <= kNDown; ++l) {
        float *dst = l < kNDown ? g->cat[l] : g->trunk[0];
        const int C = cd << l;
        if ((rc = run_encoder(g, s, l, x, ldx, bs, dst, C, nullptr, 0, 0, st)) != LWG_OK) return rc;
        x = dst;
        ldx = C;
    }
    int cur = 0;
    for (int i = 0; i < g->repeat; ++i) {
        if ((rc = run_resblock(g, s, i, g->trunk[cur], g->trunk[cur ^ 1], bs, nullptr, 0, 0, st)) != LWG_OK) return rc;
        cur ^= 1;
    }
    const float *d = g->trunk[cur];
    int dC = cd << kNDown, dH = is >> kNDown;
    for (int i = 0; i < kNDown; ++i) {
        const int oC = dC / 2, oH = dH * 2;
        if ((rc = run_conv(g, s.dec[i], d, dC, bs, dH, dH, g->raw, st)) != LWG_OK) return rc;
        if (i + 1 < kNDown) {
            if ((rc = run_apply(g, bs, oH, oH, oC, true, g->sk[i], oC, nullptr, 0, nullptr, 0, 0, st)) != LWG_OK) return rc;
            d = g->sk[i];
        }
        dC = oC;
        dH = oH;
    }
    // final conv7 + tanh on the last convT's raw output (its InstanceNorm + ReLU folded into the halo load)
    HeadsArgs h = {};
    h.x = g->raw;
    h.N = bs;
    h.H = is;
    h.W = is;
    h.scale_shift = g->ss;
    h.wh = s.heads_w;
    h.color = out_nchw;
    g->last_split = false;
    return launch_heads(h, st);
}

int lwg_generator_inference(lwg_generator *g, const float *tsf_inputs, int layout, const float *T, int bs,
                            const float *const *feats_nhwc, int align_corners, float *color, float *mask,
                            const float *bg, int bg_bs, float *pred, lwg_stream_t stream)
{
    int rc = check_ready(g, bs);
    if (rc != LWG_OK) return rc;
    LWG_REQUIRE(tsf_inputs && T && feats_nhwc, "inference: NULL argument");
    LWG_REQUIRE(!pred || (bg && (bg_bs == 1 || bg_bs == bs)), "inference: pred needs bg with batch 1 or %d", bs);
    for (int i = 0; i < kNDown + 1 + g->repeat; ++i) LWG_REQUIRE(feats_nhwc[i], "inference: feats_nhwc[%d] is NULL", i);
    const float *const Ts[2] = {T, nullptr};
    const float *const *fs[2] = {feats_nhwc, nullptr};
    return run_tsf(g, tsf_inputs, layout, Ts, fs, 1, bs, align_corners, color, mask, bg, bg_bs, pred, as_stream(stream));
}

// inference with one source per sample (feats_bs == bs: the tsf half of infer_front, generator.py:216-243) or a shared
// one (feats_bs == 1: lwg_generator_inference)
int lwg_generator_inference_n(lwg_generator *g, const float *tsf_inputs, int layout, const float *T, int bs,
                              const float *const *feats_nhwc, int feats_bs, int align_corners, float *color, float *mask,
                              const float *bg, int bg_bs, float *pred, lwg_stream_t stream)
{
    int rc = check_ready(g, bs);
    if (rc != LWG_OK) return rc;
    LWG_REQUIRE(tsf_inputs && T && feats_nhwc, "inference: NULL argument");
    LWG_REQUIRE(feats_bs == 1 || feats_bs == bs, "inference: source features must have batch 1 or %d", bs);
    LWG_REQUIRE(!pred || (bg && (bg_bs == 1 || bg_bs == bs)), "inference: pred needs bg with batch 1 or %d", bs);
    for (int i = 0; i < kNDown + 1 + g->repeat; ++i) LWG_REQUIRE(feats_nhwc[i], "inference: feats_nhwc[%d] is NULL", i);
    const float *const Ts[2] = {T, nullptr};
    const float *const *fs[2] = {feats_nhwc, nullptr};
    return run_tsf(g, tsf_inputs, layout, Ts, fs, 1, bs, align_corners, color, mask, bg, bg_bs, pred, as_stream(stream), feats_bs);
}

int lwg_generator_swap(lwg_generator *g, const float *tsf_inputs, int layout, const float *T12, const float *T21,
                       int bs, const float *const *feats12_nhwc, const float *const *feats21_nhwc, int align_corners,
                       float *color, float *mask, const float *bg, int bg_bs, float *pred, lwg_stream_t stream)
{
    int rc = check_ready(g, bs);
    if (rc != LWG_OK) return rc;
    LWG_REQUIRE(tsf_inputs && T12 && T21 && feats12_nhwc && feats21_nhwc, "swap: NULL argument");
    LWG_REQUIRE(!pred || (bg && (bg_bs == 1 || bg_bs == bs)), "swap: pred needs bg with batch 1 or %d", bs);
    const float *const Ts[2] = {T12, T21};
    const float *const *fs[2] = {feats12_nhwc, feats21_nhwc};
    return run_tsf(g, tsf_inputs, layout, Ts, fs, 2, bs, align_corners, color, mask, bg, bg_bs, pred, as_stream(stream));
}

int lwg_generator_peek(lwg_generator *g, int which, float *dst, size_t n_floats, lwg_stream_t stream)
{
    LWG_REQUIRE(g && dst, "peek: NULL argument");
    const size_t B = (size_t)g->max_batch, P = (size_t)g->is * g->is;
    const int cd = g->cd;
    const float *src = nullptr;
    size_t cap = 0;
    const bool act = which >= 0 && which <= 5;   // activation buffers: in the split-bf16 format after a bf16x3 pass
    if (which >= 0 && which < kNDown) {
        src = g->cat[which];
        cap = B * (P >> (2 * which)) * (size_t)(2 * (cd << which));
    } else if (which == 3) {
        src = g->trunk[g->trunk_out];
        cap = B * (P >> (2 * kNDown)) * (size_t)(cd << kNDown);
    } else if (which == 4 || which == 5) {
        const int lvl = kNDown - 1 - (which - 4);
        src = g->sk[which - 4];
        cap = B * (P >> (2 * lvl)) * (size_t)(cd << lvl);
    } else if (which == 6) {
        src = g->raw;
        cap = B * P * cd;
    } else if (which >= 7 && which < 7 + kNDown) {
        const int l = which - 6;
        src = g->tscale[0][l - 1];
        cap = B * (P >> (2 * l)) * 2;
    } else {
        LWG_FAIL(LWG_ERR_INVALID_ARG, "peek: unknown buffer %d", which);
    }
    const size_t n = n_floats < cap ? n_floats : cap;
    LWG_HIP(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, as_stream(stream)));
    if (act && g->last_split) return launch_unsplit(dst, n - n % 32, as_stream(stream));
    return LWG_OK;
}

int lwg_generator_set_precision(lwg_generator *g, int mode)
{
    LWG_REQUIRE(g, "set_precision: NULL handle");
    LWG_REQUIRE(mode == 0 || mode == 1, "set_precision: mode must be 0 (fp32) or 1 (bf16x3), got %d", mode);
    g->precision = mode;
    return LWG_OK;
}

int lwg_generator_profile(lwg_generator *g, int enable)
{
    LWG_REQUIRE(g, "profile: NULL handle");
    g->profile = enable != 0;
    g->ev_used = 0;
    g->ev_variant.clear();
    for (int v = 0; v < kIgemmVariants; ++v) {
        g->prof_flops[v] = 0.0;
        g->prof_launches[v] = 0;
    }
    return LWG_OK;
}

int lwg_generator_profile_variants(void) { return kIgemmVariants; }

int lwg_conv_trace(void *device_buffer, size_t bytes) { return conv_trace_set(device_buffer, bytes); }

int lwg_conv_trace_launch(int index, long long *info10)
{
    LWG_REQUIRE(info10, "conv_trace_launch: NULL argument");
    return conv_trace_launch(index, info10);
}

const char *lwg_generator_profile_variant_name(int variant)
{
    return variant >= 0 && variant < kIgemmVariants ? kIgemmVariantNames[variant] : "";
}

int lwg_generator_profile_read(lwg_generator *g, int variant, int *launches, double *total_ms, double *total_flops)
{
    LWG_REQUIRE(g && launches && total_ms && total_flops, "profile_read: NULL argument");
    LWG_REQUIRE(variant >= -1 && variant < kIgemmVariants, "profile_read: variant %d out of range", variant);
    double ms = 0.0;
    for (size_t i = 0; i + 1 < g->ev_used; i += 2) {
        if (variant >= 0 && g->ev_variant[i / 2] != variant) continue;
        LWG_HIP(hipEventSynchronize(g->ev_pool[i + 1]));
        float t = 0.f;
        LWG_HIP(hipEventElapsedTime(&t, g->ev_pool[i], g->ev_pool[i + 1]));
        ms += t;
    }
    *launches = 0;
    *total_flops = 0.0;
    for (int v = 0; v < kIgemmVariants; ++v)
        if (variant < 0 || v == variant) {
            *launches += g->prof_launches[v];
            *total_flops += g->prof_flops[v];
        }
    *total_ms = ms;
    return LWG_OK;
}

}  // extern "C"
