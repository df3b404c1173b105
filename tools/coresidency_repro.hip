// tools/coresidency_repro.hip -- torch-free reproducer of the co-residency miscompute of DESIGN.md section 5.1.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/coresidency_repro.hip -o tools/_build/coresidency_repro
//   hipcc ... -DREAL_NEIGHBOUR -DLWG_IGEMM_BENCH tools/coresidency_repro.hip impersonator_amd/csrc/capi.hip -o tools/_build/coresidency_repro_real
//   coresidency_repro [launches=300] [victim=1] [neighbour=2] [cumask=0]
//
// RESULT (end of round 3; profiles/r03_coresidency.md): a packed-fp32 VALU instruction (v_pk_mul_f32 / v_pk_add_f32 /
// v_pk_fma_f32) with op_sel set for its SECOND source returns wrong values while its CU is shared with the LDS-read +
// bf16-MFMA loop of the conv kernels.  Victims 60-73 (victim_pk) are that one instruction in a loop: 66, 67, 68, 70 (op_sel on
// src1) fail in 300 of 300 launches beside neighbour 300, everything else is clean.  The older victims are the way there and
// are kept because the write-up cites them: a rasteriser kernel that gathers a face's vertices, projects them, stores the nine
// floats and runs ~150 VALU instructions of record arithmetic on the same registers failed or not depending on where and how
// the stores were written -- because that decided whether hipcc used `v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` for two
// differences of the back-face test (tools/coresidency_asm_variants.py edits that one instruction in the assembly).
//
//   victim   1 = fused gather -> stores -> arithmetic (fails: hipcc emits the op_sel[src1] packed add here)
//            9 = the same with its stores + 24 wait states as one asm statement (a mitigation tried and refuted: it fails too)
//            60-66 = the minimal victim: one packed-fp32 instruction with / without op_sel (see victim_pk)
//            15 = 9 with record()'s arithmetic pinned behind the loads' side of the stores (nothing of it ahead of them)
//            16 = 15 without the wait states; 17 = 15 with compiled stores; 18 = asm stores, arithmetic pinned right behind them
//            2 = projection as its own kernel, arithmetic kernel reads what it wrote (the shape the product uses now)
//            3 = fused, the nine stores moved BEHIND the arithmetic
//            4 = fused as 1 + s_waitcnt vmcnt(0) right after the stores
//            5 = fused as 1 with the nine stores issued as single dwords (probe DESIGN.md section 9 lists)
//            6 = fused as 1 + 24 wait states (s_nop) between the stores and everything after them
//            7 = fused, the 36 bytes stored as three (dwordx2 + dword) pairs
//            8 = fused stores, then the back-face test's two products and its result written out as data (no branch)
//            10-14 = asm micro-victim: ONE 96-bit store, K = 0/1/2/4/8 wait states, VALU overwrite of its data registers
//            20 + 10*W + S = asm micro-victim 2: S (1-3) back-to-back stores of W (2-4) dwords, s_nop 1, VALU overwrite;
//            + 100: the overwrite as 64-bit v_pk_mov_b32 on register pairs
//   neighbour 0 = none (control), 1 = MFMA stream only, 2 = ds_read_b128 + MFMA stream (stripped conv main loop:
//            96 KiB of LDS per workgroup = one workgroup per CU, one wave per SIMD, one 16-byte LDS read behind every MFMA),
//            3 = the product's conv_igemm_bf16x3<128> itself (only when built with -DREAL_NEIGHBOUR)
//            4 = neighbour 2 with RANDOM bf16 operand data in LDS instead of values next to 1.0
//   cumask   0 = no masks; 1 = victim stream on CUs 0-127, neighbours on CUs 128-255 (same chip, never the same CU);
//            2 = victim and neighbours both confined to CUs 0-127 (always share CUs)
//
// Every victim launch runs on fixed inputs into zeroed outputs and is compared word for word, on the device, with the
// result of the same launch on an idle device.  Output: launches with at least one differing word, per output array.
#include <hip/hip_runtime.h>
#ifdef REAL_NEIGHBOUR
#include "../impersonator_amd/csrc/conv.hip"
#endif

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                                         \
    do {                                                                                                 \
        hipError_t e_ = (x);                                                                             \
        if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } \
    } while (0)

namespace repro {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kTileW = 32, kTileH = 8;
constexpr float kSliverRatio = 1e-4f, kHugeCoord = 1.0e6f;
constexpr unsigned kTileBoxEmpty = 0x0000ffffu;
struct Box { unsigned short x0, y0, x1, y1; };

__device__ __forceinline__ bool backside(const float v[9]) { return (v[7] - v[1]) * (v[3] - v[0]) < (v[4] - v[1]) * (v[6] - v[0]); }

__device__ __forceinline__ void face_inverse(const float v[9], int is, float px[3], float py[3], float inv[9], float &det)
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        px[k] = 0.5f * (v[3 * k + 0] * is + is - 1);
        py[k] = 0.5f * (v[3 * k + 1] * is + is - 1);
    }
    float m[9];
    m[0] = py[1] - py[2]; m[1] = px[2] - px[1]; m[2] = px[1] * py[2] - px[2] * py[1];
    m[3] = py[2] - py[0]; m[4] = px[0] - px[2]; m[5] = px[2] * py[0] - px[0] * py[2];
    m[6] = py[0] - py[1]; m[7] = px[1] - px[0]; m[8] = px[0] * py[1] - px[1] * py[0];
    det = px[2] * (py[0] - py[1]) + px[0] * (py[1] - py[2]) + px[1] * (py[2] - py[0]);
#pragma unroll
    for (int k = 0; k < 9; ++k) inv[k] = m[k] / det;
}

__device__ __forceinline__ unsigned record(const float v[9], int is, size_t t, float *faces_inv, Box *pbox)
{
    unsigned packed = kTileBoxEmpty;
    if (!backside(v)) {
        float px[3], py[3], inv[9], det;
        face_inverse(v, is, px, py, inv, det);
#pragma unroll
        for (int k = 0; k < 9; ++k) faces_inv[t * 9 + k] = inv[k];
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
            const float m = fmaxf(0.02f, ext * 0.025f);
            x0 = max(0, (int)ceilf(xmn - m));
            y0 = max(0, (int)ceilf(ymn - m));
            x1 = min(is - 1, (int)floorf(xmx + m));
            y1 = min(is - 1, (int)floorf(ymx + m));
        }
        if (x0 <= x1 && y0 <= y1) {
            Box bx;
            bx.x0 = (unsigned short)x0; bx.y0 = (unsigned short)y0; bx.x1 = (unsigned short)x1; bx.y1 = (unsigned short)y1;
            pbox[t] = bx;
            packed = (unsigned)(x0 / kTileW) | (unsigned)(y0 / kTileH) << 8 | (unsigned)(x1 / kTileW) << 16 | (unsigned)(y1 / kTileH) << 24;
        }
    }
    return packed;
}

__device__ __forceinline__ void project(const float *verts, const float *cam, const int *faces_idx, int b, int nv, int fn, float eye_z,
                                        float v[9])
{
    const float s = cam[b * 3 + 0], tx = cam[b * 3 + 1], ty = cam[b * 3 + 2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float *p = verts + ((size_t)b * nv + faces_idx[fn * 3 + k]) * 3;
        v[3 * k + 0] = s * (p[0] + tx);
        v[3 * k + 1] = -(s * (p[1] + ty));
        v[3 * k + 2] = p[2] - eye_z;
    }
}

// the failing shape and its variants
template <int MODE>
__global__ __launch_bounds__(256) void victim_fused(const float *__restrict__ verts, const float *__restrict__ cam,
                                                    const int *__restrict__ faces_idx, int nv, float eye_z, float *__restrict__ faces,
                                                    int bs, int nf, int is, float *__restrict__ faces_inv, Box *__restrict__ pbox,
                                                    unsigned *__restrict__ tbox)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bs * nf) return;
    const int b = i / nf, fn = i - b * nf;
    const size_t t = (size_t)i;
    float v[9];
    project(verts, cam, faces_idx, b, nv, fn, eye_z, v);
    if (MODE == 1 || MODE == 4 || MODE == 6) {
#pragma unroll
        for (int k = 0; k < 9; ++k) faces[t * 9 + k] = v[k];
        if (MODE == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // 6: every later VALU instruction is held back by 24 wait states and a dependency on all nine registers
        if (MODE == 6)
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]),
                         "+v"(v[7]), "+v"(v[8])::"memory");
    }
    if (MODE == 7) {   // the same 36 bytes as 64-bit + 32-bit stores
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            asm volatile("global_store_dwordx2 %0, %1, off\n\tglobal_store_dword %0, %2, off offset:8" ::"v"(faces + t * 9 + 3 * k),
                         "v"(*reinterpret_cast<const double *>(&v[3 * k])), "v"(v[3 * k + 2]) : "memory");
        }
    }
    // 15-18: WHERE the arithmetic may sit relative to the stores, under control.  PIN = an empty asm that redefines all nine
    // coordinates: nothing computed from them can be scheduled on the other side of it.
#define REPRO_PIN asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]))
    if (MODE == 15 || MODE == 16 || MODE == 17) REPRO_PIN;          // no arithmetic of record() ahead of the stores
    if (MODE == 9 || MODE == 15) {   // the two 16-byte stores + 24 wait states as one asm statement
        {
            typedef float f4 __attribute__((ext_vector_type(4)));
            asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dwordx4 %0, %2, off offset:16\n\ts_nop 15\n\ts_nop 7" ::"v"(faces + t * 9),
                         "v"(f4{v[0], v[1], v[2], v[3]}), "v"(f4{v[4], v[5], v[6], v[7]}) : "memory");
        }
        faces[t * 9 + 8] = v[8];
    }
    if (MODE == 16 || MODE == 18) {   // the same two 16-byte stores without the wait states
        typedef float f4 __attribute__((ext_vector_type(4)));
        asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dwordx4 %0, %2, off offset:16" ::"v"(faces + t * 9),
                     "v"(f4{v[0], v[1], v[2], v[3]}), "v"(f4{v[4], v[5], v[6], v[7]}) : "memory");
        faces[t * 9 + 8] = v[8];
    }
    if (MODE == 17) {
#pragma unroll
        for (int k = 0; k < 9; ++k) faces[t * 9 + k] = v[k];
    }
    if (MODE == 18) REPRO_PIN;                                      // all arithmetic of record() behind the stores, right behind
    // 19 / 20: VALU results written just AHEAD of the stores and read behind them -- the back-face differences, computed
    // before the stores (pinned there together with the coordinates), carried across, and written out in place of tbox.
    // 20: 24 wait states between that arithmetic and the stores.
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
    if (MODE == 19 || MODE == 20) {
        REPRO_PIN;
        d0 = v[3] - v[0]; d1 = v[4] - v[1]; d2 = v[6] - v[0]; d3 = v[7] - v[1];
        if (MODE == 19)
            asm volatile("" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]),
                         "+v"(v[6]), "+v"(v[7]), "+v"(v[8]));
        else
            asm volatile("s_nop 15\n\ts_nop 7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]),
                         "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]));
        typedef float f4 __attribute__((ext_vector_type(4)));
        asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dwordx4 %0, %2, off offset:16" ::"v"(faces + t * 9),
                     "v"(f4{v[0], v[1], v[2], v[3]}), "v"(f4{v[4], v[5], v[6], v[7]}) : "memory");
        faces[t * 9 + 8] = v[8];
        asm volatile("" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]),
                     "+v"(v[6]), "+v"(v[7]), "+v"(v[8]));
    }
    // 21-24: the same differences from ONE asm statement together with the two 16-byte stores, so that op and adjacency are
    // exact.  21: two v_pk_add_f32 immediately ahead of the stores; 22: 24 wait states in between; 23: the v_pk_add_f32
    // immediately behind the stores; 24: four v_sub_f32 immediately ahead (the unpacked control).
    if (MODE >= 21 && MODE <= 24) {
        REPRO_PIN;
        typedef float f4 __attribute__((ext_vector_type(4)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
        const f2 a = {v[3], v[4]}, b = {v[6], v[7]}, c = {v[0], v[1]};
        f2 da, db;
        float *dst = faces + t * 9;
#define REPRO_ST "global_store_dwordx4 %2, %3, off\n\tglobal_store_dwordx4 %2, %4, off offset:16\n\t"
#define REPRO_PK "v_pk_add_f32 %0, %5, %7 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %1, %6, %7 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        if (MODE == 21)
            asm volatile(REPRO_PK REPRO_ST : "=&v"(da), "=&v"(db) : "v"(dst), "v"(lo), "v"(hi), "v"(a), "v"(b), "v"(c) : "memory");
        if (MODE == 22)
            asm volatile(REPRO_PK "s_nop 15\n\ts_nop 7\n\t" REPRO_ST : "=&v"(da), "=&v"(db) : "v"(dst), "v"(lo), "v"(hi), "v"(a), "v"(b), "v"(c) : "memory");
        if (MODE == 23)
            asm volatile(REPRO_ST REPRO_PK : "=&v"(da), "=&v"(db) : "v"(dst), "v"(lo), "v"(hi), "v"(a), "v"(b), "v"(c) : "memory");
        if (MODE == 24) {
            float e0, e1, e2, e3;
            asm volatile("v_sub_f32 %0, %7, %11\n\tv_sub_f32 %1, %8, %12\n\tv_sub_f32 %2, %9, %11\n\tv_sub_f32 %3, %10, %12\n\t"
                         "global_store_dwordx4 %4, %5, off\n\tglobal_store_dwordx4 %4, %6, off offset:16"
                         : "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3)
                         : "v"(dst), "v"(lo), "v"(hi), "v"(v[3]), "v"(v[4]), "v"(v[6]), "v"(v[7]), "v"(v[0]), "v"(v[1]) : "memory");
            da = f2{e0, e1}; db = f2{e2, e3};
        }
#undef REPRO_ST
#undef REPRO_PK
        faces[t * 9 + 8] = v[8];
        d0 = da.x; d1 = da.y; d2 = db.x; d3 = db.y;
        asm volatile("" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]),
                     "+v"(v[6]), "+v"(v[7]), "+v"(v[8]));
    }
#undef REPRO_PIN
    if (MODE == 5) {
#pragma unroll
        for (int k = 0; k < 9; ++k) asm volatile("global_store_dword %0, %1, off" ::"v"(faces + t * 9 + k), "v"(v[k]) : "memory");
    }
    tbox[i] = record(v, is, t, faces_inv, pbox);
    if (MODE >= 19 && MODE <= 24) {   // the carried differences instead of the tile box: tbox differing alone = they were damaged
        const unsigned h0 = __float_as_uint(d0), h1 = __float_as_uint(d1), h2 = __float_as_uint(d2), h3 = __float_as_uint(d3);
        tbox[i] = h0 ^ (h1 << 7 | h1 >> 25) ^ (h2 << 13 | h2 >> 19) ^ (h3 << 21 | h3 >> 11);
    }
    if (MODE == 3) {
        float u[9];
        project(verts, cam, faces_idx, b, nv, fn, eye_z, u);
#pragma unroll
        for (int k = 0; k < 9; ++k) faces[t * 9 + k] = u[k];
    }
}

// victims 60-66: the minimal form.  One packed-fp32 VALU instruction per iteration on lane-dependent operands, its two
// results hashed into tbox (compared with the same launch on an idle device like every other output).  OP: 0 = v_pk_mul_f32
// with the half-swap  op_sel:[1,0] op_sel_hi:[0,1]  (low result <- src0.high, high result <- src0.low: what hipcc emits for
// the back-face differences of the failing victims); 1 = src0.high to both results (op_sel:[1,0], op_sel_hi default);
// 2 = src0.low to both results (op_sel_hi:[0,1]: the broadcast hipcc uses everywhere); 3 = no modifiers; 4 = v_pk_fma_f32
// op_sel:[1,0,0]; 5 = v_pk_add_f32 with the swap; 6 = the swap on src1 instead (op_sel:[0,1] op_sel_hi:[1,0]); 7-13: see the code.
template <int OP>
__global__ __launch_bounds__(256) void victim_pk(const float *__restrict__ verts, const float *__restrict__ cam,
                                                 const int *__restrict__ faces_idx, int nv, float eye_z, float *__restrict__ faces,
                                                 int bs, int nf, int is, float *__restrict__ faces_inv, Box *__restrict__ pbox,
                                                 unsigned *__restrict__ tbox)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bs * nf) return;
    typedef float f2 __attribute__((ext_vector_type(2)));
    const int nvals = bs * nv * 3;
    f2 x = {verts[(2 * i) % nvals], verts[(2 * i + 1) % nvals]};
    f2 y = {cam[i % (bs * 3)] + 1.5f, 0.75f + (float)(i & 7)};
    f2 z = {0.5f, -0.25f};
    unsigned h = 0;
    for (int it = 0; it < 32; ++it) {
        f2 r = {0.f, 0.f};
        if (OP == 0) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
        if (OP == 1) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(x), "v"(y));
        if (OP == 2) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
        if (OP == 3) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
        if (OP == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(r) : "v"(x), "v"(y), "v"(z));
        if (OP == 5) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
        if (OP == 6) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(y));
        if (OP == 7) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(x), "v"(y));                       // src1.high to both
        if (OP == 8) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(r) : "v"(x), "v"(y), "v"(z));
        if (OP == 9) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(r) : "v"(x), "v"(y), "v"(z));
        if (OP == 10) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(y));
        if (OP == 11) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(x), "v"(y));                       // D = {x.lo, y.hi}
        if (OP == 12) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(x), "v"(y));                       // D = {x.hi, y.lo}
        if (OP == 13) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(y));                    // src1.low to both
        if (OP == 14) {   // detail: ONE instruction (src1.high to both results), operands and both results written out
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(x), "v"(y));
            tbox[i] = __float_as_uint(r.x);
            float *o = faces_inv + (size_t)i * 9;
            o[0] = r.y; o[1] = x.x; o[2] = x.y; o[3] = y.x; o[4] = y.y;
            return;
        }
        const unsigned a = __float_as_uint(r.x), b = __float_as_uint(r.y);
        h = h * 31u + (a ^ (b << 11 | b >> 21));
        x.x += 0.25f; x.y -= 0.125f;
    }
    tbox[i] = h;
}

// victim 8: the fused shape with the decision turned into DATA.  Stores as victim 1, then the two products of the back-face
// test and the test's result (as 0/1 through v_cndmask, no branch) go to a debug buffer and the record is computed for every
// lane regardless.  If the products are right and the flag is wrong, the compare / VCC is what breaks; if a product is wrong, a
// VGPR write was lost.
__global__ __launch_bounds__(256) void victim_probe(const float *__restrict__ verts, const float *__restrict__ cam,
                                                    const int *__restrict__ faces_idx, int nv, float eye_z, float *__restrict__ faces,
                                                    int bs, int nf, float *__restrict__ dbg)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bs * nf) return;
    const int b = i / nf, fn = i - b * nf;
    float v[9];
    project(verts, cam, faces_idx, b, nv, fn, eye_z, v);
#pragma unroll
    for (int k = 0; k < 9; ++k) faces[(size_t)i * 9 + k] = v[k];
    const float c1 = (v[7] - v[1]) * (v[3] - v[0]), c2 = (v[4] - v[1]) * (v[6] - v[0]);
    const float flag = c1 < c2 ? 1.f : 0.f;
    dbg[(size_t)i * 4 + 0] = c1;
    dbg[(size_t)i * 4 + 1] = c2;
    dbg[(size_t)i * 4 + 2] = flag;
    dbg[(size_t)i * 4 + 3] = v[2] + v[5] + v[8];
}

__global__ __launch_bounds__(256) void victim_project(const float *__restrict__ verts, const float *__restrict__ cam,
                                                      const int *__restrict__ faces_idx, int nv, float eye_z, float *__restrict__ faces,
                                                      int bs, int nf)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bs * nf) return;
    float v[9];
    project(verts, cam, faces_idx, i / nf, nv, i % nf, eye_z, v);
#pragma unroll
    for (int k = 0; k < 9; ++k) faces[(size_t)i * 9 + k] = v[k];
}

__global__ __launch_bounds__(256) void victim_setup(const float *__restrict__ faces, int bs, int nf, int is, float *__restrict__ faces_inv,
                                                    Box *__restrict__ pbox, unsigned *__restrict__ tbox)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bs * nf) return;
    float v[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) v[k] = faces[(size_t)i * 9 + k];
    tbox[i] = record(v, is, (size_t)i, faces_inv, pbox);
}

// Micro-victim (victims 10-14): the suspected mechanism in isolation, in inline asm so that the instruction sequence is
// exactly this: a 96-bit store from v[40:42], K wait states (victim 10 + k: K = 0, 1, 2, 4, 8; the ISA asks for 1 after a
// store of more than 64 bits, hipcc emits s_nop 1 = 2), then VALU writes of NEW values to v40-42, then the registers are
// read back.  counts[0] += lanes whose registers do not hold the new values (a VALU write was lost), and a second
// kernel checks the stored words against the OLD values (counts[1]: the store picked up new data = the documented hazard).
template <int K>
__global__ __launch_bounds__(256) void victim_asm(unsigned *__restrict__ out_old, unsigned n, int iters, unsigned *__restrict__ counts)
{
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n) return;
    unsigned lost = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned a0 = tid * 3u + (unsigned)it * 7919u, a1 = a0 ^ 0x55aa55aau, a2 = a0 + 0x01234567u;
        const unsigned b0 = ~a0, b1 = ~a1, b2 = ~a2;
        unsigned *p = out_old + ((size_t)it * n + tid) * 3;
        unsigned r0, r1, r2;
        asm volatile("v_mov_b32 v40, %4\n\tv_mov_b32 v41, %5\n\tv_mov_b32 v42, %6\n\t"
                     "s_nop 4\n\t"
                     "global_store_dwordx3 %3, v[40:42], off\n\t"
                     "s_nop %10\n\t"
                     "v_mov_b32 v40, %7\n\tv_mov_b32 v41, %8\n\tv_mov_b32 v42, %9\n\t"
                     "s_nop 4\n\t"
                     "v_mov_b32 %0, v40\n\tv_mov_b32 %1, v41\n\tv_mov_b32 %2, v42\n\t"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2)
                     : "v"(p), "v"(a0), "v"(a1), "v"(a2), "v"(b0), "v"(b1), "v"(b2), "n"(K > 0 ? K - 1 : 0)
                     : "v40", "v41", "v42", "memory");
        lost += (r0 != b0) + (r1 != b1) + (r2 != b2);
    }
    if (lost) atomicAdd(counts + 0, lost);
}

// Micro-victim 2 (victims 20 + 10*W + S, W = store width in dwords 2..4, S = 1..3 back-to-back stores): S stores of W
// dwords from consecutive register groups starting at v40, then the two wait states hipcc leaves after a wide store
// (s_nop 1), then VALU writes of new values to ALL the stored registers, then read-back.  What the fused rasteriser setup
// kernel does around its three global_store_dwordx3.
// PK = 1: the overwrite is done by v_pk_mov_b32 on register pairs (a 64-bit VALU write, what the compiled victim's
// v_pk_add_f32 / v_pk_mul_f32 do) instead of twelve 32-bit v_add_u32.
template <int W, int S, int PK = 0>
__global__ __launch_bounds__(256) void victim_asm2(unsigned *__restrict__ out_old, unsigned n, int iters, unsigned *__restrict__ counts)
{
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n) return;
    unsigned lost = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned a = tid * 12u + (unsigned)it * 7919u, b = ~a;
        unsigned *p = out_old + ((size_t)it * n + tid) * 12;     // 48 bytes per thread and iteration: 16-byte aligned
        unsigned bad;
        // v40..v51 = a + k (old); store groups; s_nop 1; v40..v51 = b + k (new); count registers that are not new
#define LWG_OLD(k) "v_add_u32 v" #k ", %2, " #k " - 40\n\t"
#define LWG_NEW(k) "v_add_u32 v" #k ", %3, " #k " - 40\n\t"
#define LWG_PRE(k) "v_add_u32 v" #k ", %3, " #k " - 60\n\t"
#define LWG_PKM(d, s_) "v_pk_mov_b32 v[" #d ":" #d "+1], v[" #s_ ":" #s_ "+1], v[" #s_ ":" #s_ "+1] op_sel:[0,1]\n\t"
#define LWG_CHK(k) "v_add_u32 v52, %3, " #k " - 40\n\tv_cmp_ne_u32 vcc, v52, v" #k "\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
        asm volatile("v_mov_b32 %0, 0\n\t"
                     LWG_OLD(40) LWG_OLD(41) LWG_OLD(42) LWG_OLD(43) LWG_OLD(44) LWG_OLD(45) LWG_OLD(46) LWG_OLD(47) LWG_OLD(48)
                     LWG_OLD(49) LWG_OLD(50) LWG_OLD(51)
                     LWG_PRE(60) LWG_PRE(61) LWG_PRE(62) LWG_PRE(63) LWG_PRE(64) LWG_PRE(65) LWG_PRE(66) LWG_PRE(67) LWG_PRE(68)
                     LWG_PRE(69) LWG_PRE(70) LWG_PRE(71)
                     "s_nop 4\n\t"
                     ".if %4 == 2\n\t"
                     "global_store_dwordx2 %1, v[40:41], off\n\t"
                     ".if %5 > 1\n\tglobal_store_dwordx2 %1, v[42:43], off offset:8\n\t.endif\n\t"
                     ".if %5 > 2\n\tglobal_store_dwordx2 %1, v[44:45], off offset:16\n\t.endif\n\t"
                     ".endif\n\t"
                     ".if %4 == 3\n\t"
                     "global_store_dwordx3 %1, v[40:42], off\n\t"
                     ".if %5 > 1\n\tglobal_store_dwordx3 %1, v[44:46], off offset:12\n\t.endif\n\t"
                     ".if %5 > 2\n\tglobal_store_dwordx3 %1, v[48:50], off offset:24\n\t.endif\n\t"
                     ".endif\n\t"
                     ".if %4 == 4\n\t"
                     "global_store_dwordx4 %1, v[40:43], off\n\t"
                     ".if %5 > 1\n\tglobal_store_dwordx4 %1, v[44:47], off offset:16\n\t.endif\n\t"
                     ".if %5 > 2\n\tglobal_store_dwordx4 %1, v[48:51], off offset:32\n\t.endif\n\t"
                     ".endif\n\t"
                     "s_nop 1\n\t"
                     ".if %6 == 0\n\t"
                     LWG_NEW(40) LWG_NEW(41) LWG_NEW(42) LWG_NEW(43) LWG_NEW(44) LWG_NEW(45) LWG_NEW(46) LWG_NEW(47) LWG_NEW(48)
                     LWG_NEW(49) LWG_NEW(50) LWG_NEW(51)
                     ".else\n\t"
                     LWG_PKM(40, 60) LWG_PKM(42, 62) LWG_PKM(44, 64) LWG_PKM(46, 66) LWG_PKM(48, 68) LWG_PKM(50, 70)
                     ".endif\n\t"
                     "s_nop 4\n\t"
                     LWG_CHK(40) LWG_CHK(41) LWG_CHK(42) LWG_CHK(43) LWG_CHK(44) LWG_CHK(45) LWG_CHK(46) LWG_CHK(47) LWG_CHK(48)
                     LWG_CHK(49) LWG_CHK(50) LWG_CHK(51)
                     : "=&v"(bad)
                     : "v"(p), "v"(a), "v"(b), "n"(W), "n"(S), "n"(PK)
                     : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v60", "v61", "v62", "v63",
                       "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "vcc", "memory");
#undef LWG_OLD
#undef LWG_NEW
#undef LWG_CHK
#undef LWG_PRE
#undef LWG_PKM
        lost += bad;
    }
    if (lost) atomicAdd(counts + 0, lost);
}

template <int W, int S>
__global__ void victim_asm2_check(const unsigned *__restrict__ out_old, unsigned n, int iters, unsigned *__restrict__ counts)
{
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n) return;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned a = tid * 12u + (unsigned)it * 7919u;
        const unsigned *p = out_old + ((size_t)it * n + tid) * 12;
        for (int g = 0; g < S; ++g)
            for (int j = 0; j < W; ++j) bad += p[g * W + j] != a + (W == 3 ? 4 * g : W * g) + j;   // 96-bit groups start at even registers
    }
    if (bad) atomicAdd(counts + 1, bad);
}

__global__ void victim_asm_check(const unsigned *__restrict__ out_old, unsigned n, int iters, unsigned *__restrict__ counts)
{
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n) return;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned a0 = tid * 3u + (unsigned)it * 7919u, a1 = a0 ^ 0x55aa55aau, a2 = a0 + 0x01234567u;
        const unsigned *p = out_old + ((size_t)it * n + tid) * 3;
        bad += (p[0] != a0) + (p[1] != a1) + (p[2] != a2);
    }
    if (bad) atomicAdd(counts + 1, bad);
}

__global__ void compare_words(const unsigned *a, const unsigned *b, size_t n, unsigned *count)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned bad = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) bad += a[i] != b[i];
    if (bad) atomicAdd(count, bad);
}

// Stripped neighbour: the shape of conv_igemm_bf16x3<128>'s steady state without its DMA and barriers -- four waves,
// 96 KiB of LDS (one workgroup per CU), per "stage" 16 ds_read_b128 of swizzled 128-byte rows, each behind one of 24
// v_mfma_f32_32x32x16_bf16 on four accumulator tiles.  LDS = 0: the MFMA stream alone.
template <bool LDS>
__global__ __launch_bounds__(256) void neighbour_kernel(float *sink, int stages, int random_data = 0)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 24576; i += 256) {
        unsigned w = 0x3f803f80u + (unsigned)(i * 2654435761u >> 20);      // two bf16 values next to 1.0: few bits toggle
        if (random_data) {   // neighbour 4: two bf16 values with random sign / mantissa and an exponent of 2^-2 .. 2^0: every bit toggles
            unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            w = (h & 0x807f807fu) | 0x3e803e80u | ((h >> 3) & 0x00800080u);
        }
        smem[i] = __uint_as_float(w);
    }
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int frow = lane & 31, fsw = (frow >> 1) & 7;
    const int a_row = ((wave >> 1) * 64 + frow) * 32, b_row = 128 * 32 + ((wave & 1) * 64 + frow) * 32;
    float4 f[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = make_float4(1.f, 0.5f, 0.25f, 2.f);
    for (int s = 0; s < stages; ++s) {
        const float *base = smem + (s % 3) * 8192;
#pragma unroll
        for (int q = 0; q < 24; ++q) {
            const int t = q & 3;
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[q & 7]), __builtin_bit_cast(bf16x8, f[(q + 3) & 7]),
                                                             acc[t], 0, 0, 0);
            if (LDS && q < 16) {
                // random_data & 2 (neighbour 6): lanes 32-63 read the NEXT 16-byte chunk, as the product's fragments do (64 distinct
                // addresses per instruction: twice the LDS bytes of the default, whose upper half re-reads the lower half's)
                const int col = (random_data & 2) ? ((((2 * (q & 3) + (lane >> 5)) & 7) ^ fsw) * 4) : (((q & 7) ^ fsw) * 4);
                const float *p = base + ((q & 8) ? b_row : a_row) + ((q & 4) ? 32 * 32 : 0) + col;
                f[q & 7] = *reinterpret_cast<const float4 *>(p);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
    }
    float keep = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) keep += acc[t][r];
    if (keep == 123.456f) sink[0] = keep;
}

// Neighbours 901-906: workgroups that do NOTHING but hold registers (s_sleep in a loop), to separate "what the neighbour
// executes" from "where the victim's registers land in the SIMD's register file because of what the neighbour holds".
#define REPRO_OCCUPIER(name, ...)                                                          \
    __global__ __launch_bounds__(256) void name(float *sink, int spins)                    \
    {                                                                                      \
        for (int i = 0; i < spins; ++i) asm volatile("s_sleep 32" ::: __VA_ARGS__);        \
        if (spins < 0) sink[0] = 1.f;                                                      \
    }
REPRO_OCCUPIER(occupier_v64, "v63")
REPRO_OCCUPIER(occupier_v128, "v127")
REPRO_OCCUPIER(occupier_v120_a64, "v119", "a63")      // the four-wave halo kernel's allocation (116 + 64)
REPRO_OCCUPIER(occupier_v184, "v183")                 // the eight-wave halo kernel's (182, no AGPRs)
REPRO_OCCUPIER(occupier_v256, "v255")
REPRO_OCCUPIER(occupier_v192_a64, "v191", "a63")

}  // namespace repro

using namespace repro;

int main(int argc, char **argv)
{
    const int launches = argc > 1 ? atoi(argv[1]) : 300;
    const int victim = argc > 2 ? atoi(argv[2]) : 1;
    const int neigh = argc > 3 ? atoi(argv[3]) : 2;
    const int cumask = argc > 4 ? atoi(argv[4]) : 0;
    const int bs = 8, is = 256, GW = 84, GH = 82, nv = (GW + 1) * (GH + 1), nf = GW * GH * 2;
    const float eye_z = -(1.0f / tanf(30.0f * 3.14159265f / 180.0f) + 1.0f);

    // a body-sized sheet of 13776 small triangles (+ two whole-image slivers per frame, as the pipeline's mesh has), 8 cameras
    std::vector<float> hverts((size_t)bs * nv * 3), hcam(bs * 3);
    std::vector<int> hfaces(nf * 3);
    srand(1);
    for (int b = 0; b < bs; ++b) {
        hcam[b * 3 + 0] = 0.8f + 0.03f * b; hcam[b * 3 + 1] = 0.01f * b; hcam[b * 3 + 2] = -0.02f * b;
        for (int y = 0; y <= GH; ++y)
            for (int x = 0; x <= GW; ++x) {
                float *p = &hverts[((size_t)b * nv + y * (GW + 1) + x) * 3];
                p[0] = -0.3f + 0.6f * x / GW + 0.002f * ((float)rand() / RAND_MAX);
                p[1] = -0.85f + 1.7f * y / GH + 0.002f * ((float)rand() / RAND_MAX);
                p[2] = 0.1f * sinf(0.2f * x + b) * cosf(0.15f * y);
            }
    }
    for (int y = 0, f = 0; y < GH; ++y)
        for (int x = 0; x < GW; ++x) {
            const int v00 = y * (GW + 1) + x, v10 = v00 + 1, v01 = v00 + GW + 1, v11 = v01 + 1;
            hfaces[f * 3 + 0] = v00; hfaces[f * 3 + 1] = v01; hfaces[f * 3 + 2] = v10; ++f;   // orientation: front-facing after the y flip
            hfaces[f * 3 + 0] = v10; hfaces[f * 3 + 1] = v01; hfaces[f * 3 + 2] = v11; ++f;
        }
    hfaces[0] = 0; hfaces[1] = GW / 2; hfaces[2] = GW;                                         // two degenerate (collinear) faces: slivers
    hfaces[3] = 0; hfaces[4] = (GW + 1) * (GH / 2); hfaces[5] = (GW + 1) * GH;

    float *verts, *cam, *faces, *faces_inv, *ref_faces, *ref_inv, *sink;
    int *faces_idx;
    Box *pbox, *ref_pbox;
    unsigned *tbox, *ref_tbox, *counts;
    const size_t nface = (size_t)bs * nf;
    CHECK(hipMalloc(&verts, hverts.size() * 4)); CHECK(hipMalloc(&cam, hcam.size() * 4)); CHECK(hipMalloc(&faces_idx, hfaces.size() * 4));
    CHECK(hipMalloc(&faces, nface * 36)); CHECK(hipMalloc(&faces_inv, nface * 36)); CHECK(hipMalloc(&pbox, nface * 8)); CHECK(hipMalloc(&tbox, nface * 4));
    CHECK(hipMalloc(&ref_faces, nface * 36)); CHECK(hipMalloc(&ref_inv, nface * 36)); CHECK(hipMalloc(&ref_pbox, nface * 8)); CHECK(hipMalloc(&ref_tbox, nface * 4));
    CHECK(hipMalloc(&counts, 4 * 4)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemcpy(verts, hverts.data(), hverts.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(cam, hcam.data(), hcam.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(faces_idx, hfaces.data(), hfaces.size() * 4, hipMemcpyHostToDevice));

    hipStream_t sv, sn[2];
    if (cumask) {
        // 256 CUs = 8 masks of 32 bits; victim on the low half, neighbours on the other (1) or the same (2) half
        unsigned low[8] = {~0u, ~0u, ~0u, ~0u, 0, 0, 0, 0}, high[8] = {0, 0, 0, 0, ~0u, ~0u, ~0u, ~0u};
        CHECK(hipExtStreamCreateWithCUMask(&sv, 8, low));
        for (int k = 0; k < 2; ++k) CHECK(hipExtStreamCreateWithCUMask(&sn[k], 8, cumask == 1 ? high : low));
    } else {
        CHECK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) CHECK(hipStreamCreateWithFlags(&sn[k], hipStreamNonBlocking));
    }
    const int blocks = (int)((nface + 255) / 256);
    // REPRO_CO=<code object> REPRO_KERNEL=<mangled name>: the victim comes from a separately assembled code object with the
    // signature of victim_fused -- hand-edited variants of the compiler's assembly (tools/coresidency_asm_variants.py)
    hipFunction_t co_fn = nullptr;
    if (getenv("REPRO_CO")) {
        hipModule_t mod;
        if (hipModuleLoad(&mod, getenv("REPRO_CO")) != hipSuccess || hipModuleGetFunction(&co_fn, mod, getenv("REPRO_KERNEL")) != hipSuccess) {
            fprintf(stderr, "cannot load %s / %s\n", getenv("REPRO_CO"), getenv("REPRO_KERNEL"));
            return 2;
        }
    }
    auto run_victim = [&](hipStream_t st) {
        if (co_fn) {
            const float *a0 = verts, *a1 = cam; const int *a2 = faces_idx; int a3 = nv; float a4 = eye_z; float *a5 = faces;
            int a6 = bs, a7 = nf, a8 = is; float *a9 = faces_inv; Box *a10 = pbox; unsigned *a11 = tbox;
            void *params[] = {&a0, &a1, &a2, &a3, &a4, &a5, &a6, &a7, &a8, &a9, &a10, &a11};
            const int nblocks = (bs * nf + 255) / 256;
            hipModuleLaunchKernel(co_fn, nblocks, 1, 1, 256, 1, 1, 0, st, params, nullptr);
            return;
        }
        CHECK(hipMemsetAsync(faces, 0, nface * 36, st)); CHECK(hipMemsetAsync(faces_inv, 0, nface * 36, st));
        CHECK(hipMemsetAsync(pbox, 0, nface * 8, st)); CHECK(hipMemsetAsync(tbox, 0, nface * 4, st));
        switch (victim) {
            case 1: victim_fused<1><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 3: victim_fused<3><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 4: victim_fused<4><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 5: victim_fused<5><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 6: victim_fused<6><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 7: victim_fused<7><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 9: victim_fused<9><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 19: victim_fused<19><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 20: victim_fused<20><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 21: victim_fused<21><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 22: victim_fused<22><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 23: victim_fused<23><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 24: victim_fused<24><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 60: victim_pk<0><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 61: victim_pk<1><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 62: victim_pk<2><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 63: victim_pk<3><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 64: victim_pk<4><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 65: victim_pk<5><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 66: victim_pk<6><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 67: victim_pk<7><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 68: victim_pk<8><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 69: victim_pk<9><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 70: victim_pk<10><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 71: victim_pk<11><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 72: victim_pk<12><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 73: victim_pk<13><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 74: victim_pk<14><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 15: victim_fused<15><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 16: victim_fused<16><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 17: victim_fused<17><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            case 18: victim_fused<18><<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, is, faces_inv, pbox, tbox); break;
            default:
                victim_project<<<blocks, 256, 0, st>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf);
                victim_setup<<<blocks, 256, 0, st>>>(faces, bs, nf, is, faces_inv, pbox, tbox);
        }
        CHECK(hipGetLastError());
    };
    // reference: the same launch on an idle device
    run_victim(sv);
    CHECK(hipStreamSynchronize(sv));
    CHECK(hipMemcpy(ref_faces, faces, nface * 36, hipMemcpyDeviceToDevice)); CHECK(hipMemcpy(ref_inv, faces_inv, nface * 36, hipMemcpyDeviceToDevice));
    CHECK(hipMemcpy(ref_pbox, pbox, nface * 8, hipMemcpyDeviceToDevice)); CHECK(hipMemcpy(ref_tbox, tbox, nface * 4, hipMemcpyDeviceToDevice));
    {
        std::vector<unsigned> ht(nface);
        CHECK(hipMemcpy(ht.data(), tbox, nface * 4, hipMemcpyDeviceToHost));
        size_t live = 0;
        for (unsigned v : ht) live += v != kTileBoxEmpty;
        printf("reference: %zu of %zu faces have a non-empty box\n", live, nface);
    }

    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&neighbour_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&neighbour_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
#ifdef REAL_NEIGHBOUR
    lwg::ConvArgs ca = {};
    {
        const int N = 8, H = 32, C = 512, K = 9 * C;
        const size_t xin = (size_t)N * H * H * C;
        float *xs, *ws, *y;
        float2 *part;
        CHECK(hipMalloc(&xs, xin * 4 + 4096)); CHECK(hipMemset(xs, 0x3c, xin * 4)); CHECK(hipMemset(xs + xin, 0, 4096));
        CHECK(hipMalloc(&ws, (size_t)C * K * 4)); CHECK(hipMemset(ws, 0x3c, (size_t)C * K * 4));
        CHECK(hipMalloc(&y, xin * 4)); CHECK(hipMalloc(&part, (size_t)(N * H * H / 128) * C * 8));
        ca.w_split = ws; ca.w = ws; ca.x = xs; ca.ldx = C; ca.N = N; ca.H = H; ca.W = H; ca.Cin = C; ca.cin_log2 = 9; ca.zeros = xs + xin;
        ca.y = y; ca.ldy = C; ca.Ho = H; ca.Wo = H; ca.Cout = C; ca.Hm = H; ca.Wm = H; ca.stride = 1; ca.pad = 1; ca.os = 1; ca.dil = 1;
        ca.partials = part; ca.mtiles = N * H * H / 128; ca.nphase = 1; ca.precision = 1; ca.tap_inner = 1;
        ca.ph[0].KH = ca.ph[0].KW = 3; ca.ph[0].ntaps = 9; ca.ph[0].Kpad = K; ca.ph[0].w_off = 0;
    }
#endif
    auto run_neighbour = [&](hipStream_t st) {
        if (neigh == 1) neighbour_kernel<false><<<256, 256, 98304, st>>>(sink, 150);
        if (neigh == 2) neighbour_kernel<true><<<256, 256, 98304, st>>>(sink, 150);
        if (neigh == 4) neighbour_kernel<true><<<256, 256, 98304, st>>>(sink, 150, 1);    // the same loop on random operand data
        if (neigh == 5) neighbour_kernel<false><<<256, 256, 98304, st>>>(sink, 150, 1);   // MFMAs only (operands stay constant)
        if (neigh == 6) neighbour_kernel<true><<<256, 256, 98304, st>>>(sink, 150, 3);    // random data, 64 distinct 16-byte reads per instruction
        if (neigh >= 4 && neigh <= 6) return;
        // register occupiers: 1024 workgroups of four waves, ~250 us each
        if (neigh == 901) occupier_v64<<<1024, 256, 0, st>>>(sink, 200);
        if (neigh == 902) occupier_v128<<<1024, 256, 0, st>>>(sink, 200);
        if (neigh == 903) occupier_v120_a64<<<1024, 256, 0, st>>>(sink, 200);
        if (neigh == 904) occupier_v184<<<1024, 256, 0, st>>>(sink, 200);
        if (neigh == 905) occupier_v256<<<1024, 256, 0, st>>>(sink, 200);
        if (neigh == 906) occupier_v192_a64<<<1024, 256, 0, st>>>(sink, 200);
        if (neigh >= 901 && neigh <= 906) return;
#ifdef REAL_NEIGHBOUR
        if (neigh == 3) lwg::launch_conv_igemm_dbg(ca, 128, 200, st);
        if (neigh >= 100 && lwg::launch_conv_igemm_dbg(ca, 128, neigh, st) != 0) { fprintf(stderr, "unknown variant %d\n", neigh); exit(2); }
#else
        if (neigh >= 3) { fprintf(stderr, "neighbour %d needs -DREAL_NEIGHBOUR\n", neigh); exit(2); }
#endif
    };

    bool dumped = false;
    if (victim == 8) {
        float *dbg, *ref_dbg;
        CHECK(hipMalloc(&dbg, nface * 16)); CHECK(hipMalloc(&ref_dbg, nface * 16));
        victim_probe<<<blocks, 256, 0, sv>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, ref_dbg);
        CHECK(hipStreamSynchronize(sv));
        std::vector<float> hr(nface * 4), hg(nface * 4);
        CHECK(hipMemcpy(hr.data(), ref_dbg, nface * 16, hipMemcpyDeviceToHost));
        int bad_launch = 0;
        unsigned long long wrong_c = 0, wrong_flag_only = 0, wrong_sum = 0, lanes_hi = 0, lanes_any = 0;
        for (int it = 0; it < launches; ++it) {
            for (int k = 0; k < 6; ++k) run_neighbour(sn[k & 1]);
            CHECK(hipMemsetAsync(dbg, 0, nface * 16, sv));
            CHECK(hipMemsetAsync(counts, 0, 16, sv));
            victim_probe<<<blocks, 256, 0, sv>>>(verts, cam, faces_idx, nv, eye_z, faces, bs, nf, dbg);
            compare_words<<<256, 256, 0, sv>>>((const unsigned *)dbg, (const unsigned *)ref_dbg, nface * 4, counts);
            unsigned h;
            CHECK(hipMemcpyAsync(&h, counts, 4, hipMemcpyDeviceToHost, sv));
            CHECK(hipStreamSynchronize(sv));
            if (h) {
                ++bad_launch;
                CHECK(hipMemcpy(hg.data(), dbg, nface * 16, hipMemcpyDeviceToHost));
                for (size_t f = 0; f < nface; ++f) {
                    const bool c = memcmp(&hg[f * 4], &hr[f * 4], 8) != 0, fl = hg[f * 4 + 2] != hr[f * 4 + 2], sm = hg[f * 4 + 3] != hr[f * 4 + 3];
                    if (c || fl || sm) { ++lanes_any; lanes_hi += (f % 64) >= 48; }
                    wrong_c += c; wrong_flag_only += (fl && !c); wrong_sum += sm;
                }
            }
            if ((it & 7) == 7) { CHECK(hipStreamSynchronize(sn[0])); CHECK(hipStreamSynchronize(sn[1])); }
        }
        CHECK(hipDeviceSynchronize());
        printf("victim 8 (decision as data) neighbour %d cumask %d: %d of %d launches differ; lanes with a wrong product %llu, with right products "
               "but a wrong flag %llu, with a wrong z-sum %llu; wrong lanes in 48-63: %llu of %llu\n", neigh, cumask, bad_launch, launches,
               wrong_c, wrong_flag_only, wrong_sum, lanes_hi, lanes_any);
        return 0;
    }
    if (victim >= 40 && (victim < 60 || victim > 74)) {
        const int pk = victim >= 140 ? 1 : 0;
        const int iters = 32, W = (victim - 100 * pk - 20) / 10, S = (victim - 100 * pk - 20) % 10;
        const unsigned n = (unsigned)nface;
        unsigned *out_old;
        CHECK(hipMalloc(&out_old, (size_t)iters * n * 48));
        int bad_launch[2] = {0, 0};
        unsigned long long tot[2] = {0, 0};
        for (int it = 0; it < launches; ++it) {
            for (int k = 0; k < 6; ++k) run_neighbour(sn[k & 1]);
            CHECK(hipMemsetAsync(counts, 0, 16, sv));
#define LWG_CASE(w, s_)                                                                              \
    if (W == w && S == s_) {                                                                          \
        if (pk) victim_asm2<w, s_, 1><<<blocks, 256, 0, sv>>>(out_old, n, iters, counts);             \
        else victim_asm2<w, s_><<<blocks, 256, 0, sv>>>(out_old, n, iters, counts);                   \
        victim_asm2_check<w, s_><<<blocks, 256, 0, sv>>>(out_old, n, iters, counts);                  \
    }
            LWG_CASE(2, 1) LWG_CASE(2, 2) LWG_CASE(2, 3) LWG_CASE(3, 1) LWG_CASE(3, 2) LWG_CASE(3, 3) LWG_CASE(4, 1) LWG_CASE(4, 2) LWG_CASE(4, 3)
#undef LWG_CASE
            CHECK(hipGetLastError());
            unsigned h[4];
            CHECK(hipMemcpyAsync(h, counts, 16, hipMemcpyDeviceToHost, sv));
            CHECK(hipStreamSynchronize(sv));
            for (int k = 0; k < 2; ++k) { bad_launch[k] += h[k] != 0; tot[k] += h[k]; }
            if ((it & 7) == 7) { CHECK(hipStreamSynchronize(sn[0])); CHECK(hipStreamSynchronize(sn[1])); }
        }
        CHECK(hipDeviceSynchronize());
        printf("asm victim %d (%d back-to-back stores of %d dwords, s_nop 1, VALU overwrite) neighbour %d cumask %d: launches (of %d) with lost "
               "VALU writes %d (%llu registers), with stored words != old values %d (%llu words)\n", victim, S, W, neigh, cumask, launches,
               bad_launch[0], tot[0], bad_launch[1], tot[1]);
        return 0;
    }
    if (victim >= 10 && victim <= 14) {
        const int iters = 32;
        const unsigned n = (unsigned)nface;
        unsigned *out_old;
        CHECK(hipMalloc(&out_old, (size_t)iters * n * 12));
        int bad_launch[2] = {0, 0};
        unsigned long long tot[2] = {0, 0};
        for (int it = 0; it < launches; ++it) {
            for (int k = 0; k < 6; ++k) run_neighbour(sn[k & 1]);
            CHECK(hipMemsetAsync(counts, 0, 16, sv));
            switch (victim) {
                case 10: victim_asm<0><<<blocks, 256, 0, sv>>>(out_old, n, iters, counts); break;
                case 11: victim_asm<1><<<blocks, 256, 0, sv>>>(out_old, n, iters, counts); break;
                case 12: victim_asm<2><<<blocks, 256, 0, sv>>>(out_old, n, iters, counts); break;
                case 13: victim_asm<4><<<blocks, 256, 0, sv>>>(out_old, n, iters, counts); break;
                default: victim_asm<8><<<blocks, 256, 0, sv>>>(out_old, n, iters, counts); break;
            }
            victim_asm_check<<<blocks, 256, 0, sv>>>(out_old, n, iters, counts);
            unsigned h[4];
            CHECK(hipMemcpyAsync(h, counts, 16, hipMemcpyDeviceToHost, sv));
            CHECK(hipStreamSynchronize(sv));
            for (int k = 0; k < 2; ++k) { bad_launch[k] += h[k] != 0; tot[k] += h[k]; }
            if ((it & 7) == 7) { CHECK(hipStreamSynchronize(sn[0])); CHECK(hipStreamSynchronize(sn[1])); }
        }
        CHECK(hipDeviceSynchronize());
        printf("asm victim %d (wait states %d) neighbour %d cumask %d: launches (of %d) with lost VALU writes %d (%llu registers), with "
               "stored words != old values %d (%llu words)\n", victim, victim == 10 ? 0 : victim == 11 ? 1 : victim == 12 ? 2 : victim == 13 ? 4 : 8,
               neigh, cumask, launches, bad_launch[0], tot[0], bad_launch[1], tot[1]);
        return 0;
    }
    int wrong[4] = {0, 0, 0, 0};
    unsigned long long words[4] = {0, 0, 0, 0};
    for (int it = 0; it < launches; ++it) {
        for (int k = 0; k < 6; ++k) run_neighbour(sn[k & 1]);       // ~0.3 ms of neighbours on each stream
        CHECK(hipMemsetAsync(counts, 0, 16, sv));
        run_victim(sv);
        compare_words<<<256, 256, 0, sv>>>((const unsigned *)faces, (const unsigned *)ref_faces, nface * 9, counts + 0);
        compare_words<<<256, 256, 0, sv>>>((const unsigned *)faces_inv, (const unsigned *)ref_inv, nface * 9, counts + 1);
        compare_words<<<256, 256, 0, sv>>>((const unsigned *)pbox, (const unsigned *)ref_pbox, nface * 2, counts + 2);
        compare_words<<<256, 256, 0, sv>>>(tbox, ref_tbox, nface, counts + 3);
        unsigned h[4];
        CHECK(hipMemcpyAsync(h, counts, 16, hipMemcpyDeviceToHost, sv));
        CHECK(hipStreamSynchronize(sv));
        for (int k = 0; k < 4; ++k) { wrong[k] += h[k] != 0; words[k] += h[k]; }
        if (!dumped && (h[1] || h[3]) && getenv("REPRO_DUMP")) {
            // what is wrong: per face (= lane) expected / got, so that the pattern over lanes and the values can be read
            dumped = true;
            std::vector<float> gi(nface * 9), ri(nface * 9), gf(nface * 9);
            std::vector<unsigned> gt(nface), rt(nface);
            CHECK(hipMemcpy(gi.data(), faces_inv, nface * 36, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(ri.data(), ref_inv, nface * 36, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(gf.data(), faces, nface * 36, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(gt.data(), tbox, nface * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(rt.data(), ref_tbox, nface * 4, hipMemcpyDeviceToHost));
            if (victim == 74) {   // which lanes, and what the instruction computed instead of x.lo * y.hi / x.hi * y.hi
                size_t lanes[4] = {0, 0, 0, 0}, nb = 0, lo_as_lolo = 0, lo_as_hilo = 0, lo_zero = 0, hi_bad = 0, hi_as_hilo = 0, hi_as_lolo = 0, other = 0;
                for (size_t i = 0; i < nface; ++i) {
                    const float *g = &gi[i * 9];
                    const float xl = g[1], xh = g[2], yl = g[3], yh = g[4], rx = *(float *)&gt[i], ry = g[0];
                    const bool bx = rx != xl * yh, by = ry != xh * yh;
                    if (!bx && !by) continue;
                    ++nb; ++lanes[(i % 64) / 16];
                    if (bx) { if (rx == xl * yl) ++lo_as_lolo; else if (rx == xh * yl) ++lo_as_hilo; else if (rx == 0.f) ++lo_zero; else ++other; }
                    if (by) { ++hi_bad; if (ry == xh * yl) ++hi_as_hilo; else if (ry == xl * yl) ++hi_as_lolo; }
                    if (nb <= 4) printf("  thread %zu (lane %zu): x = (%g, %g) y = (%g, %g): low result %g (x.lo*y.hi = %g, x.lo*y.lo = %g), high result %g (x.hi*y.hi = %g, x.hi*y.lo = %g)\n",
                                        i, i % 64, xl, xh, yl, yh, rx, xl * yh, xl * yl, ry, xh * yh, xh * yl);
                }
                printf("  detail: %zu wrong threads in this launch; by quarter-wave (lanes 0-15, 16-31, 32-47, 48-63): %zu %zu %zu %zu; low result = x.lo*y.lo "
                       "(read src1.LOW instead of src1.high) %zu, = x.hi*y.lo %zu, = 0 %zu, something else %zu; high result wrong %zu (= x.hi*y.lo %zu, = x.lo*y.lo %zu)\n",
                       nb, lanes[0], lanes[1], lanes[2], lanes[3], lo_as_lolo, lo_as_hilo, lo_zero, other, hi_bad, hi_as_hilo, hi_as_lolo);
            }
            int shown = victim == 74 ? 6 : 0;
            size_t run_start = 0, nbad = 0;
            for (size_t i = 0; i < nface; ++i) {
                const bool bad = gt[i] != rt[i] || memcmp(&gi[i * 9], &ri[i * 9], 36) != 0;
                if (bad) {
                    if (!nbad || i != run_start + nbad) { if (nbad) printf("  run of %zu faces from %zu (wave %zu, lane %zu)\n", nbad, run_start, run_start / 64, run_start % 64); run_start = i; nbad = 0; }
                    ++nbad;
                    if (shown < 6) {
                        ++shown;
                        printf("  face %zu (block %zu wave %zu lane %zu): tbox exp %08x got %08x\n    inv exp", i, i / 256, (i / 64) % 4, i % 64, rt[i], gt[i]);
                        for (int k = 0; k < 9; ++k) printf(" %.6g", ri[i * 9 + k]);
                        printf("\n    inv got");
                        for (int k = 0; k < 9; ++k) printf(" %.6g(%08x)", gi[i * 9 + k], *(unsigned *)&gi[i * 9 + k]);
                        printf("\n    f2v got");
                        for (int k = 0; k < 9; ++k) printf(" %.6g", gf[i * 9 + k]);
                        printf("\n");
                    }
                }
            }
            if (nbad) printf("  run of %zu faces from %zu (wave %zu, lane %zu)\n", nbad, run_start, run_start / 64, run_start % 64);
        }
        if ((it & 7) == 7) { CHECK(hipStreamSynchronize(sn[0])); CHECK(hipStreamSynchronize(sn[1])); }
    }
    CHECK(hipDeviceSynchronize());
    printf("victim %d neighbour %d cumask %d: launches with differing words (of %d): f2verts %d, faces_inv %d, pbox %d, tbox %d; "
           "words: %llu %llu %llu %llu\n", victim, neigh, cumask, launches, wrong[0], wrong[1], wrong[2], wrong[3], words[0], words[1],
           words[2], words[3]);
    return 0;
}
