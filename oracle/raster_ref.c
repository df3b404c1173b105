/*
 * oracle/raster_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, gcc) of the only hot-path piece of the reference that cannot be
 * executed in this container: the two forward kernels of neural_renderer's rasteriser
 *   thirdparty/neural_renderer/neural_renderer/cuda/rasterize_cuda_kernel.cu:40-84   (per-face inverse)
 *   thirdparty/neural_renderer/neural_renderer/cuda/rasterize_cuda_kernel.cu:86-186  (per-pixel z-buffer)
 * plus the pre-fill values of rasterize.py:50-52 and the vertical flips of rasterize.py:334-338.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product path (impersonator_amd/csrc) never links or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_raster.py checks this file against the reference's own
 * known-answer fixtures (teapot_blender.png silhouette: exact; test_depth.png: atol 1e-2, the
 * reference's tolerance) -- see tests/golden/make_golden.py for how they were extracted.
 *
 * Numeric contract: scalar_t is float; the reference mixes in double literals.
 * Every expression below keeps the same operand types as the .cu file; build with
 * -ffp-contract=off so that no multiply-add is fused (x86-64 SSE2 evaluates each op in IEEE
 * binary32/binary64 exactly as written).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* back-face predicate shared by both kernels (.cu:57 and .cu:128) */
static inline int is_backside(const float *v)
{
    return (v[7] - v[1]) * (v[3] - v[0]) < (v[4] - v[1]) * (v[6] - v[0]);
}

/* .cu:40-84 -- one "thread" per (batch, face). faces_inv must be zero-filled by the caller for
 * culled faces to stay zero (rasterize_cuda.cpp allocates it with at::zeros_like). */
ORACLE_API void nmr_face_inverse(const float *faces, float *faces_inv, int batch, int num_faces,
                                 int image_size)
{
    const int is = image_size;
    const long total = (long)batch * num_faces;
    for (long t = 0; t < total; ++t) {
        const float *v = faces + t * 9;
        float *out = faces_inv + t * 9;
        if (is_backside(v))
            continue;

        /* pixel-space vertex positions: float*int -> float, +int -> float, -1 -> float,
         * then 0.5 (double) * float -> double, narrowed on assignment (.cu:64) */
        float px[3], py[3];
        for (int k = 0; k < 3; ++k) {
            px[k] = (float)(0.5 * (v[3 * k + 0] * is + is - 1));
            py[k] = (float)(0.5 * (v[3 * k + 1] * is + is - 1));
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
        const float det = px[2] * (py[0] - py[1]) + px[0] * (py[1] - py[2]) + px[1] * (py[2] - py[0]);
        for (int k = 0; k < 9; ++k)
            out[k] = m[k] / det;
    }
}

/* CUDA's max(double,double)/min(double,double) are fmax/fmin: a NaN operand yields the other one */
static inline float clamp01_like_cuda(float w)
{
    return (float)fmin(fmax((double)w, 0.), 1.);
}

/* .cu:86-186 -- one "thread" per (batch, pixel); loops every face in index order.
 * Outputs are only written for covered pixels (caller pre-fills -1 / 0 / far). */
ORACLE_API void nmr_face_index_map(const float *faces, const float *faces_inv, int32_t *fim,
                                   float *wim, float *depth, int batch, int num_faces,
                                   int image_size, float near, float far)
{
    const int is = image_size;
    const int nf = num_faces;
    const long npix = (long)batch * is * is;
#pragma omp parallel for schedule(dynamic, 256)
    for (long i = 0; i < npix; ++i) {
        const int bn = (int)(i / ((long)is * is));
        const int pn = (int)(i % ((long)is * is));
        const int yi = pn / is;
        const int xi = pn % is;
        /* (2. * yi + 1 - is) / is is evaluated in double and narrowed (.cu:113-114) */
        const float yp = (float)((2. * yi + 1 - is) / is);
        const float xp = (float)((2. * xi + 1 - is) / is);

        float best_z = far;
        int best_f = -1;
        float best_w[3] = {0.f, 0.f, 0.f};

        for (int fn = 0; fn < nf; ++fn) {
            const float *v = faces + ((long)bn * nf + fn) * 9;
            const float *inv = faces_inv + ((long)bn * nf + fn) * 9;
            if (is_backside(v))
                continue;
            /* three edge tests at the pixel centre, pure float (.cu:132-134) */
            if (((yp - v[1]) * (v[3] - v[0]) < (xp - v[0]) * (v[4] - v[1])) ||
                ((yp - v[4]) * (v[6] - v[3]) < (xp - v[3]) * (v[7] - v[4])) ||
                ((yp - v[7]) * (v[0] - v[6]) < (xp - v[6]) * (v[1] - v[7])))
                continue;

            /* barycentrics from the precomputed inverse: float * int -> float (.cu:139-141) */
            float w[3];
            for (int k = 0; k < 3; ++k)
                w[k] = inv[3 * k + 0] * xi + inv[3 * k + 1] * yi + inv[3 * k + 2];

            float w_sum = 0;
            for (int k = 0; k < 3; ++k) {
                w[k] = clamp01_like_cuda(w[k]);
                w_sum += w[k];
            }
            for (int k = 0; k < 3; ++k)
                w[k] /= w_sum;

            /* perspective-correct depth; 1. is a double literal (.cu:153) */
            const float zp = (float)(1. / (w[0] / v[2] + w[1] / v[5] + w[2] / v[8]));
            if (zp <= near || far <= zp)
                continue;

            if (zp < best_z) { /* strict: the lowest face index wins ties (H6) */
                best_z = zp;
                best_f = fn;
                best_w[0] = w[0];
                best_w[1] = w[1];
                best_w[2] = w[2];
            }
        }

        if (best_f >= 0) {
            depth[i] = best_z;
            fim[i] = best_f;
            wim[3 * i + 0] = best_w[0];
            wim[3 * i + 1] = best_w[1];
            wim[3 * i + 2] = best_w[2];
        }
    }
}

/* rasterize.py:22-98 (fills) + rasterize_cuda.cpp:70-95 (zeros_like faces_inv) + the two kernels +
 * rasterize.py:334-338 (torch.flip along the image rows).  depth may be NULL. */
ORACLE_API int nmr_rasterize_fim_wim(const float *faces, int batch, int num_faces, int image_size,
                                     float near, float far, int32_t *fim, float *wim, float *depth)
{
    const int is = image_size;
    const long npix = (long)batch * is * is;
    float *inv = (float *)calloc((size_t)batch * num_faces * 9, sizeof(float));
    int32_t *fim_raw = (int32_t *)malloc(sizeof(int32_t) * npix);
    float *wim_raw = (float *)calloc((size_t)npix * 3, sizeof(float));
    float *dep_raw = (float *)malloc(sizeof(float) * npix);
    if (!inv || !fim_raw || !wim_raw || !dep_raw) {
        free(inv); free(fim_raw); free(wim_raw); free(dep_raw);
        return -1;
    }
    for (long i = 0; i < npix; ++i) {
        fim_raw[i] = -1;
        dep_raw[i] = far;
    }
    nmr_face_inverse(faces, inv, batch, num_faces, is);
    nmr_face_index_map(faces, inv, fim_raw, wim_raw, dep_raw, batch, num_faces, is, near, far);

    for (int b = 0; b < batch; ++b)
        for (int y = 0; y < is; ++y) {
            const long src = ((long)b * is + (is - 1 - y)) * is;
            const long dst = ((long)b * is + y) * is;
            memcpy(fim + dst, fim_raw + src, sizeof(int32_t) * is);
            memcpy(wim + dst * 3, wim_raw + src * 3, sizeof(float) * 3 * is);
            if (depth)
                memcpy(depth + dst, dep_raw + src, sizeof(float) * is);
        }
    free(inv); free(fim_raw); free(wim_raw); free(dep_raw);
    return 0;
}
