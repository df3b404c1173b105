/*
 * lwg.h -- C ABI of liblwg.so, the MI355X (gfx950) native library behind the Imitator.forward()
 * hot path of Liquid Warping GAN (reference: svip-lab/impersonator).
 *
 * This is the drop-in boundary.  The reference's native boundary for this path
 * is the pybind11 module `neural_renderer.cuda.rasterize` (rasterize_cuda.cpp:70-95,194-200) plus the
 * ATen/cuDNN operators PyTorch dispatches for networks/generator.py; liblwg replaces both with plain
 * C entry points: raw device pointers + sizes, no C++ or torch types, never throws, returns 0 or a
 * negative LWG_ERR_* code (text via lwg_last_error()).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter name ends in `_host`;
 *   - the caller owns every input/output buffer (PyTorch tensors in the Python host code);
 *   - all work is enqueued on the caller's stream (`lwg_stream_t` = hipStream_t), no implicit sync;
 *   - fp32 everywhere, int32 face ids; image tensors are NCHW at the boundary like the reference's,
 *     feature maps handed between entry points are NHWC (== torch.channels_last storage);
 *   - a `lwg_generator` handle owns only what the caller cannot see: re-laid-out weights and scratch.
 *     One handle per device, not to be used from two threads at once.
 */
#ifndef LWG_H_
#define LWG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LWG_API __attribute__((visibility("default")))

typedef void *lwg_stream_t; /* hipStream_t */

enum {
    LWG_OK = 0,
    LWG_ERR_INVALID_ARG = -1, /* NULL pointer, non-positive size, ... (reference: AT_CHECK -> RuntimeError) */
    LWG_ERR_UNSUPPORTED = -2, /* shape outside what the kernels are built for */
    LWG_ERR_WORKSPACE = -3,   /* workspace too small / misaligned */
    LWG_ERR_HIP = -4,         /* a HIP runtime call or kernel launch failed (reference only printf'd these) */
    LWG_ERR_STATE = -5        /* handle not ready (weights missing, batch above max_batch, ...) */
};

LWG_API int lwg_version(void);
/* thread-local text of the last failure in this thread; never NULL */
LWG_API const char *lwg_last_error(void);
/* name may be NULL; returns LWG_ERR_HIP when no device is visible */
LWG_API int lwg_device_info(int *cu_count, size_t *hbm_bytes, char *name_host, size_t name_len);

/* ------------------------------------------------------------------------------------------------
 * Geometry: replaces the python glue in front of the rasteriser.
 * ---------------------------------------------------------------------------------------------- */

/* verts (bs,nv,3), cam (bs,3) = [s,tx,ty], faces_idx (nf,3) -> f2verts (bs,nf,3,3).
 * = orthographic_proj_withz_idrot (utils/nmr.py:10-28) + y flip (nmr.py:271) + nr.look_at with the
 * renderer's eye (0,0,eye_z) (nmr.py:177,273; look_at.py:6-62: identity rotation, z -= eye_z)
 * + nr.vertices_to_faces (vertices_to_faces.py:4-22). */
LWG_API int lwg_project_faces(const float *verts, const float *cam, const int32_t *faces_idx, int bs, int nv,
                              int nf, float eye_z, float *f2verts, lwg_stream_t stream);

/* Rasteriser: replaces rasterize_cuda.forward_face_index_map (rasterize_cuda.cpp:70-95; kernels
 * rasterize_cuda_kernel.cu:40-186) together with the fills of rasterize.py:50-52 and the vertical
 * flips of rasterize.py:334-338.  faces (bs,nf,3,3) -> fim int32 (bs,is,is) [-1 = background],
 * wim (bs,is,is,3) [0 where uncovered], depth (bs,is,is) [far where uncovered] or NULL.
 * Results are bit-identical to the UNcontracted (no fused multiply-add) evaluation of the .cu file's float
 * expressions, lowest face index winning depth ties; nvcc contracts a*b+c by default, which moves the barycentric
 * weights of a CUDA build by the amounts tabulated in profiles/r02_fma_sensitivity.md (coverage and face indices: 0).
 * Stateless between calls (the workspace is scratch for the duration of one call): no global atomics, no depth buffer
 * in memory.  workspace: lwg_rasterize_workspace_bytes, 256-byte aligned. */
LWG_API size_t lwg_rasterize_workspace_bytes(int bs, int nf, int image_size);
LWG_API int lwg_rasterize_fim_wim(const float *faces, int bs, int nf, int image_size, float near_z, float far_z,
                                  int32_t *fim, float *wim, float *depth, void *workspace, size_t workspace_bytes,
                                  lwg_stream_t stream);

/* SMPLRenderer.encode_fim (utils/nmr.py:328-341): out = map_fn[fim] with fim == -1 -> last row.
 * map_fn (nrows,nc); fim (bs,npix); out (bs,nc,npix) when transpose != 0 else (bs,npix,nc). */
LWG_API int lwg_encode_fim(const int32_t *fim, const float *map_fn, int bs, int npix, int nrows, int nc,
                           int transpose, float *out, lwg_stream_t stream);

/* SMPLRenderer.cal_bc_transform (utils/nmr.py:617-659): T = sum_k wim[k] * src_f2pts[fim][k] on
 * covered pixels, (-2,-2) elsewhere.  src_f2pts (src_bs,nf,3,2) with src_bs == 1 (shared) or bs. */
LWG_API int lwg_cal_bc_transform(const float *src_f2pts, int src_bs, const int32_t *fim, const float *wim, int bs,
                                 int nf, int image_size, float *T, lwg_stream_t stream);

/* F.grid_sample(x, grid, mode='bilinear', padding_mode='zeros') as called at models/imitator.py:259 and
 * networks/generator.py:313.  x (xn,C,H,W) NCHW with xn == 1 (shared) or n; grid (n,Ho,Wo,2);
 * out (n,C,Ho,Wo).  align_corners: 0 = torch>=1.3 default (parity target), 1 = torch 1.2 behaviour (H1). */
LWG_API int lwg_grid_sample(const float *x, int xn, int C, int H, int W, const float *grid, int n, int Ho, int Wo,
                            int align_corners, float *out, lwg_stream_t stream);

/* ImpersonatorGenerator.resize_trans (networks/generator.py:303-310): bilinear, align_corners=True,
 * of the flow field T (bs,H,W,2) -> (bs,h,w,2). */
LWG_API int lwg_resize_flow(const float *T, int bs, int H, int W, int h, int w, float *out, lwg_stream_t stream);

/* One launch sequence for models/imitator.py:250-260 given posed vertices:
 * project -> rasterise -> cond = map_fn[fim] -> T -> tsf_img = grid_sample(src_img, T)
 * -> tsf_inputs = cat(tsf_img, cond).  Shared source: src_p2verts (nf,3,2), src_img (3,is,is).
 * Outputs (any of cond/tsf_img/tsf_inputs_nhwc8 may be NULL): f2verts (bs,nf,3,3), fim, wim,
 * cond (bs,nc,is,is), T (bs,is,is,2), tsf_img (bs,3,is,is), tsf_inputs_nhwc8 (bs,is,is,8) =
 * [tsf_img(3), cond(nc=3), 0, 0] per pixel -- the layout lwg_generator_* consumes directly. */
LWG_API size_t lwg_transfer_workspace_bytes(int bs, int nf, int image_size);
LWG_API int lwg_transfer_frame(const float *verts, const float *cam, const int32_t *faces_idx, int bs, int nv, int nf,
                               int image_size, float eye_z, float near_z, float far_z, const float *map_fn, int nc,
                               const float *src_p2verts, const float *src_img, int align_corners, float *f2verts,
                               int32_t *fim, float *wim, float *cond, float *T, float *tsf_img,
                               float *tsf_inputs_nhwc8, void *workspace, size_t workspace_bytes,
                               lwg_stream_t stream);

/* SMPL.forward (networks/batch_smpl.py:285-375) for a batch of frames: theta (bs, 3+72+num_betas) =
 * [cam, pose, shape] -> verts (bs,nv,3), joints (bs,num_out_joints,3) [optional], Rs (bs,24,3,3) [optional].
 * Model tensors in the reference's own layouts: v_template (nv,3), shapedirs (num_betas, nv*3),
 * posedirs (207, nv*3), weights (nv,24), joint_regressor (nv,num_out_joints), parents (24).
 * J_template (24,3) and J_shapedirs (num_betas, 72) are the joint regressor applied to v_template / shapedirs
 * (the regression J_regressor^T v_shaped of batch_smpl.py:318-321 is linear in the shape; precomputed once). */
/* The tensor glue around SMPL.forward on the per-frame path, as two launches.
 * smpl_swap: Imitator.swap_smpl (models/imitator.py:216-234) + the slicing of HumanModelRecovery.get_details
 *   (networks/hmr.py:302-330): tgt_smpl (bs, 75+num_betas) -> theta (bs, 75+num_betas) = [cam, tgt pose, src shape] and its
 *   contiguous parts cam (bs,3), pose (bs,72), shape (bs,num_betas).  strategy 1 'smooth': cam = [src_s, src_t + (tgt_t -
 *   first_t)]; 2 'source': src_cam; 3 'copy': the target's own camera; 0: theta = tgt_smpl unchanged (get_details only).
 *   src_cam (3), src_shape (num_betas), first_cam (3): device pointers, NULL where the strategy does not read them.
 * smpl_project_joints: batch_orth_proj_idrot (networks/batch_smpl.py:221-234): j2d (bs,J,2) = cam_s * (j3d_xy + cam_t). */
LWG_API int lwg_smpl_swap(const float *tgt_smpl, int bs, int num_betas, int strategy, const float *src_cam,
                          const float *src_shape, const float *first_cam, float *theta, float *cam, float *pose, float *shape,
                          lwg_stream_t stream);
LWG_API int lwg_smpl_project_joints(const float *j3d, const float *cam, int bs, int num_joints, float *j2d, lwg_stream_t stream);
/* Viewer.rotate_trans (models/viewer.py:240-247), `torch.bmm(X, R) + t` of the novel-view path: out (n,3) = x (n,3) @ R + t.
 * R9 (row-major 3x3) and t3 are HOST pointers (twelve floats, passed to the kernel by value); x / out device pointers. */
LWG_API int lwg_rotate_translate(const float *x, long n, const float *R9, const float *t3, float *out, lwg_stream_t stream);
LWG_API size_t lwg_smpl_workspace_bytes(int bs);
LWG_API int lwg_smpl_forward(const float *theta, int bs, int num_betas, int nv, int num_out_joints,
                             const float *v_template, const float *shapedirs, const float *posedirs,
                             const float *J_template, const float *J_shapedirs, const int32_t *parents,
                             const float *weights, const float *joint_regressor, float *verts, float *joints,
                             float *Rs, void *workspace, size_t workspace_bytes, lwg_stream_t stream);
/* The same function with every intermediate in fp64 and ONE rounding to fp32 at the end ("compensated" mode of
 * networks/batch_smpl.py:285-375): the fp32 model tensors are read as they are, J_template / J_shapedirs are the fp64
 * products of the fp32 regressor with the fp32 template / shape directions.  The result is the correctly rounded value of
 * the function the reference's code defines; any fp32 evaluation of it (the reference's own included) differs from that by
 * its ~1e-6 of summation noise, which the rasteriser downstream amplifies. */
LWG_API int lwg_smpl_forward_f64(const float *theta, int bs, int num_betas, int nv, int num_out_joints,
                                 const float *v_template, const float *shapedirs, const float *posedirs,
                                 const double *J_template, const double *J_shapedirs, const int32_t *parents,
                                 const float *weights, const float *joint_regressor, float *verts, float *joints,
                                 float *Rs, void *workspace, size_t workspace_bytes, lwg_stream_t stream);

/* The mask bookkeeping of appearance transfer, models/swapper.py:198-253 (Swapper.swap / calculate_trans), batch 1 as the reference
 * runs it, NCHW fp32.  With these a swap launches no framework kernel and never reads the device back (the reference's
 * `T11[~src_left_mask[0]] = -2` does), so the whole call can be captured in a HIP graph.
 *   lwg_swap_masks  : part (nparts,H,W) = source part map (encode_fim with the 'par' table); bit c of selected_bits / left_bits = part c
 *                     belongs to the swapped / the kept set.  -> part_mask, left_mask (H,W) in {0,1} = (channel sum != 0), and
 *                     T11 (H,W,2) = grid where left_mask else -2 (grid: the identity sampling grid, utils/nmr.py:490-504).
 *   lwg_mask_faces  : out = f2pts with the faces flagged in `drop` (one byte per face) set to -2 (tsf_f2p[0, left_faces] = -2).
 *   lwg_swap_compose: out (3+nc,H,W) = cat([tsf21 * part_mask + tsf11 * left_mask, cond]).
 *   lwg_clamp       : x = min(max(x, lo), hi) in place (T21.clamp_(-2, 2)). */
LWG_API int lwg_swap_masks(const float *part, int nparts, int H, int W, unsigned selected_bits, unsigned left_bits, const float *grid,
                           float *part_mask, float *left_mask, float *T11, lwg_stream_t stream);
LWG_API int lwg_mask_faces(const float *f2pts, const unsigned char *drop, int nf, int per_face, float *out, lwg_stream_t stream);
LWG_API int lwg_swap_compose(const float *tsf21, const float *tsf11, const float *part_mask, const float *left_mask, const float *cond,
                             int nc, int H, int W, float *out, lwg_stream_t stream);
LWG_API int lwg_clamp(float *x, size_t n, float lo, float hi, lwg_stream_t stream);
/* *out (device) = max |a[i] - b[i]| over n floats; a NaN difference reports +inf.  The comparison behind the generator's
 * `precision="auto"` (impersonator_amd/networks/generator.py): the reference computes networks/generator.py:80-133 in fp32
 * throughout, the library's bf16x3 arithmetic is narrower -- one probe pass in both arithmetics per weight set decides which one
 * serves it, and this is its `(a - b).abs().max()` without a framework kernel. */
LWG_API int lwg_max_abs_diff(const float *a, const float *b, size_t n, float *out, lwg_stream_t stream);
/* ---- Once-per-source glue of Imitator.personalize (models/imitator.py:82-155), so that `personalize` launches no
 * framework kernel.
 * morph: utils/util.py:73-89 -- erode (mode 0: pad with 1, count == ks*ks) / dilate (mode 1: pad with 0, count >= 1) of a
 *   mask with a ks x ks box (ks odd, <= 31); mask (n,1,H,W) fp32 whose images sit batch_stride floats apart (a channel slice of
 *   an NCHW tensor qualifies), out (n,1,H,W) dense.  Exact for {0,1} masks (integer counts).  complement != 0 writes
 *   1 - result (body_mask = 1 - bg_mask, ft_mask = 1 - erode: imitator.py:117,134).
 * mask_compose: torch.cat([img * m, tail], dim=1) with m = mask or 1 - mask (invert): img (n,3,H,W), mask (n,1,H,W), tail
 *   (n,tail_channels,H,W) -> out (n,3+tail_channels,H,W) (imitator.py:127-128,135).
 * source_p2verts: hazard H9 (imitator.py:105-107): negates y of f2verts (bs,nf,3,3) IN PLACE (the reference does it through
 *   the p2verts view) and writes the contiguous p2verts (bs,nf,3,2) = f2verts[..., 0:2].
 * vis_f2pts: SMPLRenderer.get_vis_f2pts (utils/nmr.py:506-546, --only_vis, hazard H10): out = f2pts (bs,nf,per_face floats per
 *   face) where the face id is among fim.unique()[1:] -- the sorted unique values minus the SMALLEST one present, whatever it
 *   is -- and -2 elsewhere. */
LWG_API int lwg_morph(const float *mask, int n, int H, int W, long batch_stride, int ks, int mode, int complement, float *out,
                      lwg_stream_t stream);
LWG_API int lwg_mask_compose(const float *img, const float *mask, int invert, const float *tail, int tail_channels, int n,
                             int H, int W, float *out, lwg_stream_t stream);
LWG_API int lwg_source_p2verts(float *f2verts, int bs, int nf, float *p2verts, lwg_stream_t stream);
LWG_API size_t lwg_vis_f2pts_workspace_bytes(int bs, int nf);
LWG_API int lwg_vis_f2pts(const float *f2pts, int bs, int nf, int per_face, const int32_t *fim, int H, int W, float *out,
                          void *workspace, size_t workspace_bytes, lwg_stream_t stream);

/* NCHW (n,C,H,W) -> NHWC with the channel count padded to cpad (zeros), and back (first C channels). */
LWG_API int lwg_pack_nhwc(const float *x_nchw, int n, int C, int H, int W, int cpad, float *out_nhwc,
                          lwg_stream_t stream);
LWG_API int lwg_unpack_nchw(const float *x_nhwc, int n, int C, int H, int W, int cpad, float *out_nchw,
                            lwg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Generator: replaces what PyTorch dispatches for ImpersonatorGenerator (networks/generator.py:187-320):
 * conv / conv-transpose / InstanceNorm / ReLU / grid_sample / interpolate / cat / tanh / sigmoid.
 * ---------------------------------------------------------------------------------------------- */
typedef struct lwg_generator lwg_generator;

/* ImpersonatorGenerator(bg_dim, src_dim, tsf_dim, conv_dim=64, repeat_num=6), n_down = 3
 * (generator.py:189-202).  Allocates weights + scratch for batches up to max_batch.  src_dim / tsf_dim = 3 + the condition
 * map's channels (models/models.py:85-94, utils/mesh.py:446-473): 1..32, i.e. every map_name of the reference ('uv_seg' 6,
 * 'par' 14, 'binary' 18).  Inputs of at most 8 channels take the NHWC8 path (layout 1 of lwg_generator_inference) and, under
 * precision 1, the LDS-resident bf16x3 stem; wider ones are NCHW only and their 7x7 stem runs on the exact-fp32 kernel. */
LWG_API int lwg_generator_create(lwg_generator **out, int src_dim, int tsf_dim, int conv_dim, int repeat_num,
                                 int image_size, int max_batch);
LWG_API void lwg_generator_destroy(lwg_generator *g);

/* Arithmetic of the convolutions.  0: exact fp32 on v_mfma_f32_32x32x2_f32 (bit-for-bit an fmaf chain).
 * 1 (default): every fp32 operand split into two bf16 terms, hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with
 * fp32 accumulation -- 16 mantissa bits per operand, ~2^-16 relative per product, measured 8e-5 L-inf on the final
 * image against the fp32 reference (budget 1e-3), at ~2.2x the speed.  The 7x7 stem and heads are fp32 in both modes. */
LWG_API int lwg_generator_set_precision(lwg_generator *g, int mode);

/* Batched forms for ImpersonatorGenerator.infer_front / forward (generator.py:204-243: one source per sample, both
 * streams decoded).  encode_src_n: src_inputs (bs,src_dim,is,is) -> the same feature list with batch bs.
 * decode_src: src_model.regress(src_model.decode(...)) on those features -> color (bs,3,is,is), mask (bs,1,is,is).
 * inference_n: lwg_generator_inference with feats_bs in {1, bs} sources. */
LWG_API int lwg_generator_encode_src_n(lwg_generator *g, const float *src_inputs_nchw, int bs, float *const *feats_nhwc,
                                       lwg_stream_t stream);
LWG_API int lwg_generator_decode_src(lwg_generator *g, const float *const *feats_nhwc, int bs, float *color, float *mask,
                                     lwg_stream_t stream);
LWG_API int lwg_generator_inference_n(lwg_generator *g, const float *tsf_inputs, int layout, const float *T, int bs,
                                      const float *const *feats_nhwc, int feats_bs, int align_corners, float *color,
                                      float *mask, const float *bg, int bg_bs, float *pred, lwg_stream_t stream);

/* BGNet = ImpersonatorGenerator.bg_model (ResNetGenerator, networks/generator.py:23-65), used by Imitator.personalize
 * when --bg_model ORIGINAL (models/imitator.py:30-34, 127-132).  enable_bg allocates its weights (call before feeding
 * "bg_model.*" keys, which are ignored otherwise); bg_forward: (bs, bg_dim, is, is) NCHW -> (bs, 3, is, is), fp32. */
LWG_API int lwg_generator_enable_bg(lwg_generator *g, int bg_dim);
LWG_API int lwg_generator_bg_forward(lwg_generator *g, const float *bg_inputs_nchw, int bs, float *out_nchw,
                                     lwg_stream_t stream);

/* Feed one state_dict entry (PyTorch layout, HOST memory), e.g.
 * "tsf_model.encoders.0.0.weight" (64,6,7,7), "tsf_model.resnets.2.main.1.bias" (512,),
 * "tsf_model.decoders.0.0.weight" (512,256,3,3), "tsf_model.attetion_reg.0.weight" (1,64,7,7).
 * Keys of bg_model.* are accepted and ignored (not on this path).  Unknown keys: LWG_ERR_INVALID_ARG. */
LWG_API int lwg_generator_load_weight(lwg_generator *g, const char *key, const float *data_host, const int64_t *shape,
                                      int ndim);
/* number of src_model/tsf_model entries still missing (0 = ready) */
LWG_API int lwg_generator_missing_weights(const lwg_generator *g);

/* Shapes of the 4 + repeat_num cached source feature maps, in the order
 * encoder_outs[0..3], resnet_outs[0..repeat_num-1] (generator.py:136-147). */
LWG_API int lwg_generator_num_src_features(const lwg_generator *g);
LWG_API int lwg_generator_src_feature_shape(const lwg_generator *g, int index, int *C, int *H, int *W);

/* ImpersonatorGenerator.encode_src (generator.py:213-214): src_inputs (1,src_dim,is,is) NCHW ->
 * feats_nhwc[i] (1,H,W,C) for every feature above (caller-allocated). */
LWG_API int lwg_generator_encode_src(lwg_generator *g, const float *src_inputs_nchw, float *const *feats_nhwc,
                                     lwg_stream_t stream);

/* ImpersonatorGenerator.inference (generator.py:277-301) + the blend of Imitator.forward
 * (models/imitator.py:326-336).
 *   tsf_inputs: layout 0 = NCHW (bs,tsf_dim,is,is); 1 = NHWC8 (bs,is,is,8)
 *   T (bs,is,is,2); feats_nhwc as produced by encode_src (batch 1, shared by all frames)
 *   color (bs,3,is,is), mask (bs,1,is,is): may be NULL
 *   bg (bg_bs,3,is,is) with bg_bs in {1,bs} and pred (bs,3,is,is) = mask*bg + (1-mask)*color: both may be NULL
 */
LWG_API int lwg_generator_inference(lwg_generator *g, const float *tsf_inputs, int layout, const float *T, int bs,
                                    const float *const *feats_nhwc, int align_corners, float *color, float *mask,
                                    const float *bg, int bg_bs, float *pred, lwg_stream_t stream);

/* ImpersonatorGenerator.swap (generator.py:245-275): two warped sources per level; optional fused blend
 * pred = mask*bg + (1-mask)*color of Swapper.forward (models/swapper.py:261-271), bg/pred as in inference. */
LWG_API int lwg_generator_swap(lwg_generator *g, const float *tsf_inputs, int layout, const float *T12,
                               const float *T21, int bs, const float *const *feats12_nhwc,
                               const float *const *feats21_nhwc, int align_corners, float *color, float *mask,
                               const float *bg, int bg_bs, float *pred, lwg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Background inpaintor: replaces what PyTorch dispatches for InpaintSANet(c_dim=4) in eval mode
 * (networks/inpaintor.py:110-202): 35 gated convolutions with folded BatchNorm, nearest x2 up-sampling,
 * one self-attention over (image_size/4)^2 tokens, mask compositing and clamps.  Runs once per source image
 * (models/imitator.py:124-125), batch 1.
 * ---------------------------------------------------------------------------------------------- */
typedef struct lwg_inpaint lwg_inpaint;
LWG_API int lwg_inpaint_create(lwg_inpaint **out, int c_dim, int image_size);
LWG_API void lwg_inpaint_destroy(lwg_inpaint *g);
/* state_dict entries in PyTorch layout (HOST memory), e.g. "coarse_net.3.mask_conv2d.weight" (128,64,4,4),
 * "refine_upsample_net.2.conv2d.batch_norm2d.running_var" (64,), "refine_attn.gamma" (1,);
 * "...num_batches_tracked" is accepted and ignored. */
LWG_API int lwg_inpaint_load_weight(lwg_inpaint *g, const char *key, const float *data_host, const int64_t *shape,
                                    int ndim);
LWG_API int lwg_inpaint_missing_weights(const lwg_inpaint *g);
/* Arithmetic of the gated convolutions with >= 32 input channels (31 of the 35; the 5x5 entry layers, the 16 -> 3 exit layers and
 * the attention's projections and products always run in exact fp32): 1 (default) = bf16x3, the split-operand kernels of the
 * generator (three bf16 MFMA products per multiply-add, fp32 accumulate); 0 = exact fp32 MFMA everywhere. */
LWG_API int lwg_inpaint_set_precision(lwg_inpaint *g, int precision);
/* InpaintSANet.forward(imgs, masks) (inpaintor.py:178-202): imgs (1,3,is,is), masks (1,1,is,is) ->
 * coarse_x [optional], x (refined, clamped), comp_imgs = x*masks + imgs*(1-masks) [optional], all (1,3,is,is). */
LWG_API int lwg_inpaint_forward(lwg_inpaint *g, const float *imgs, const float *masks, float *coarse_x, float *x,
                                float *comp_imgs, lwg_stream_t stream);

/* ---- training, first slice: the PatchGAN discriminator update ---------------------------------
 * Replaces PatchDiscriminator.forward (networks/discriminator.py:8-57, norm_type='instance', use_sigmoid=False) and,
 * for the discriminator, ImpersonatorTrainer._optimize_D + loss.backward() + torch.optim.Adam.step()
 * (models/impersonator_trainer.py:229-232, 396-414).  state_dict keys: "model.<i>.weight" (Cout,Cin,4,4) and
 * "model.<i>.bias" (Cout,) with the nn.Sequential indices of the reference (0, 2, 5, 8, ...). */
typedef struct lwg_discriminator lwg_discriminator;
LWG_API int lwg_discriminator_create(lwg_discriminator **out, int input_nc, int ndf, int n_layers, int image_size,
                                     int max_batch);
LWG_API void lwg_discriminator_destroy(lwg_discriminator *d);
LWG_API int lwg_discriminator_load_weight(lwg_discriminator *d, const char *key, const float *data_host,
                                          const int64_t *shape, int ndim);
/* copies a parameter (from_grads = 0) or its gradient (1) back to the host in PyTorch layout; synchronises */
LWG_API int lwg_discriminator_read_weight(lwg_discriminator *d, const char *key, int from_grads, float *data_host,
                                          size_t n_floats);
LWG_API int lwg_discriminator_num_params(const lwg_discriminator *d, size_t *n_floats);
/* Arithmetic of the discriminator's convolutions (forward, data gradient, weight gradient): 0 (default) fp32 MFMA; 1 bf16x3 --
 * the same fp32 tensors in memory, every operand split into two bf16 terms inside the kernels (hi*hi + hi*lo + lo*hi with
 * fp32 accumulation, ~2^-16 relative per product), what `conv_precision='bf16x3'` selects for the generator's streams
 * (reference: networks/discriminator.py:8-57 run in fp32 by cuDNN; the training step's tolerance is a loss curve, not bits). */
LWG_API int lwg_discriminator_set_precision(lwg_discriminator *d, int mode);
LWG_API int lwg_discriminator_output_size(const lwg_discriminator *d, int *h);   /* patch map edge (14 at 256, n_layers 4) */
/* x (bs,input_nc,is,is) NCHW device -> out (bs,1,h,h) */
LWG_API int lwg_discriminator_forward(lwg_discriminator *d, const float *x_nchw, int bs, float *out, lwg_stream_t stream);
/* loss = mean((D(real)-1)^2) + mean((D(fake)+1)^2) and its gradient wrt every parameter, into the handle's flat
 * gradient buffer (real, fake: (bs,input_nc,is,is) NCHW device; loss_device: optional device float). */
LWG_API int lwg_discriminator_backward(lwg_discriminator *d, const float *real_nchw, const float *fake_nchw, int bs,
                                       float *loss_device, lwg_stream_t stream);
/* The flat device buffers (n_floats each, identical layout): what a data-parallel job all-reduces between backward
 * and adam_step (torch.distributed / RCCL on the caller's side; padding entries are always zero). */
LWG_API int lwg_discriminator_buffers(lwg_discriminator *d, float **params, float **grads, size_t *n_floats);
/* torch.optim.Adam(lr, betas=(beta1, beta2), eps) step on the gradient buffer */
LWG_API int lwg_discriminator_adam_step(lwg_discriminator *d, float lr, float beta1, float beta2, float eps,
                                        lwg_stream_t stream);

/* ---- op-level convolution and its gradients (building blocks of the generator-side training step) ----------------
 * Tensors NHWC fp32 on the device; weights in PyTorch layout: (Cout,Cin,k,k) for Conv2d, (Cin,Cout,3,3) for
 * ConvTranspose2d(k3,s2,p1,output_padding 1) (`transposed` = 1; H, W are then the INPUT size, the output is 2H x 2W).
 * Replaces F.conv2d / F.conv_transpose2d and torch.autograd's conv backward (the reference: networks/generator.py:8-20,
 * 80-133 under loss.backward(), models/impersonator_trainer.py:355-357).  fp32 MFMA; channel counts powers of two >= 8,
 * the side that becomes the GEMM's N dimension a multiple of 64 (Cout for forward, Cin for backward_data; for
 * backward_weight Cout, or Cin when transposed).  stride 1: any k <= 7 with 'same' padding for backward_data;
 * stride 2: k3 p1 on even sizes.  workspace: lwg_conv2d_workspace_bytes, scratch only (nothing persists).
 * precision 0: fp32 MFMA.  precision 1 (bf16x3: operands carried as two bf16 terms, three MFMA products, fp32
 * accumulation: ~2^-16 relative per operand): forward and backward_data run the inference path's kernel wherever the
 * layer fits it -- no bias, reduction-side channels a multiple of 32, output grid per image a multiple of 128 pixels, at
 * most 32 taps -- and the fp32 kernel elsewhere; backward_weight runs its own bf16x3 kernel on every layer. */
typedef struct lwg_conv2d_desc {
    int N, H, W, Cin, Cout, k, stride, pad, transposed;
    int precision;
} lwg_conv2d_desc;
LWG_API size_t lwg_conv2d_workspace_bytes(const lwg_conv2d_desc *d);
LWG_API int lwg_conv2d_forward(const lwg_conv2d_desc *d, const float *x, const float *w, const float *bias, float *y,
                               void *workspace, size_t workspace_bytes, lwg_stream_t stream);
LWG_API int lwg_conv2d_backward_data(const lwg_conv2d_desc *d, const float *dy, const float *w, float *dx,
                                     void *workspace, size_t workspace_bytes, lwg_stream_t stream);
LWG_API int lwg_conv2d_backward_weight(const lwg_conv2d_desc *d, const float *x, const float *dy, float *dw, float *dbias,
                                       void *workspace, size_t workspace_bytes, lwg_stream_t stream);

/* The regression heads inside the training step (networks/generator.py:142-152, 163-171: img_reg = Conv2d(64,3,7,1,3) +
 * Tanh, attetion_reg = Conv2d(64,1,7,1,3) + Sigmoid, no bias).  x (N,H,W,64) NHWC; w (w_rows >= 4, 64, 7, 7): rows 0-2 the
 * colour head, row 3 the mask head.  forward: color (N,3,H,W) = tanh(conv), mask (N,1,H,W) = sigmoid(conv), either may
 * be NULL.  backward_weight: dy8 (N,H,W,8) = gradient wrt the PRE-activation outputs (channels 4-7 zero) -> dw (8,64,7,7).
 * The data gradient is lwg_conv2d_backward_data with Cout = 8.  workspace: lwg_heads_workspace_bytes, scratch only.
 * PRECONDITION: x >= 0 (it is the output of the last IN + ReLU block, generator.py:129-133,178-181).  The forward is the
 * inference path's heads kernel, whose operand load folds that ReLU in: a negative entry of x is read as 0, i.e. the
 * entry point computes tanh/sigmoid(conv(max(x, 0))) and neither gradient accounts for the clamp. */
LWG_API size_t lwg_heads_workspace_bytes(int N, int H, int W);
LWG_API int lwg_heads_forward(const float *x, int N, int H, int W, const float *w, int w_rows, float *color, float *mask,
                              void *workspace, size_t workspace_bytes, lwg_stream_t stream);
LWG_API int lwg_heads_backward_weight(const float *x, const float *dy8, int N, int H, int W, float *dw, void *workspace,
                                      size_t workspace_bytes, lwg_stream_t stream);

/* Generator-side adversarial term (models/impersonator_trainer.py:369-371): loss = mean((D(x) - target)^2) on
 * x (bs,input_nc,is,is) NCHW and its gradient wrt x (same shape); the discriminator's parameters get no gradient. */
LWG_API int lwg_discriminator_input_grad(lwg_discriminator *d, const float *x_nchw, int bs, float target, float *loss_device,
                                         float *dx_nchw, lwg_stream_t stream);
/* bilinear grid_sample (zeros padding) on NHWC tensors: x (xn,H,W,C), xn in {1, n}; grid (n,Ho,Wo,2) -> y (n,Ho,Wo,C) */
LWG_API int lwg_grid_sample_nhwc(const float *x, int xn, int C, int H, int W, const float *grid, int n, int Ho, int Wo,
                                 int align_corners, float *y, lwg_stream_t stream);

/* InstanceNorm2d(affine=True, eps 1e-5, biased variance) [+ ReLU] and its gradient, NHWC fp32 (x: (N,HW,C)).
 * stats: (N,C,2) floats (mean, rstd) written by forward, read by backward.  backward: y = the forward output when it
 * went through the ReLU (its sign is the mask) or NULL; dgamma/dbeta (C,) overwritten.  scratch: device memory of
 * lwg_instance_norm_scratch_bytes(N, HW, C) bytes (8-byte aligned; slab partial sums of the two-stage reductions). */
LWG_API size_t lwg_instance_norm_scratch_bytes(int N, int HW, int C);
LWG_API int lwg_instance_norm_forward(const float *x, int N, int HW, int C, const float *gamma, const float *beta, int relu,
                                      float *y, float *stats, void *scratch, lwg_stream_t stream);
LWG_API int lwg_instance_norm_backward(const float *x, const float *y, const float *dy, const float *stats,
                                       const float *gamma, int N, int HW, int C, float *dx, float *dgamma, float *dbeta,
                                       void *scratch, lwg_stream_t stream);
/* Gradient of bilinear grid_sample (zeros padding) wrt its input, NHWC: dy (n,Ho,Wo,C), grid (n,Ho,Wo,2) ->
 * dx (xn,H,W,C) ACCUMULATED (zero it first), xn in {1, n}.  Atomic fp32 adds (not bit-reproducible, as torch's). */
LWG_API int lwg_grid_sample_backward(const float *dy, const float *grid, int xn, int C, int H, int W, int n, int Ho, int Wo,
                                     int align_corners, float *dx, lwg_stream_t stream);
/* The same gradient, DETERMINISTIC (what the training step uses; replaces torch's grid_sampler_2d_backward behind
 * networks/generator.py:312-315 in `loss_G.backward()`, models/impersonator_trainer.py:355-356): the scatter is planned once per
 * flow field -- every source texel gets the list of (output pixel, tap, bilinear weight) that reach it, sorted by pixel -- and
 * applied per tensor as a gather that adds the contributions in that fixed order (bit-reproducible; same zeros-padding taps and
 * weights as lwg_grid_sample_nhwc).  plan: caller-owned device memory of lwg_grid_sample_plan_bytes() bytes (4-byte aligned),
 * written by lwg_grid_sample_plan, read by any number of lwg_grid_sample_backward_planned calls with the SAME dimensions
 * (one plan serves every warp of a pyramid level: the Liquid Warping Block warps seven feature maps with one flow).
 * dx (xn,H,W,C) is ACCUMULATED (zero it first); C a multiple of 4. */
LWG_API size_t lwg_grid_sample_plan_bytes(int xn, int H, int W, int n, int Ho, int Wo);
LWG_API int lwg_grid_sample_plan(const float *grid, int xn, int H, int W, int n, int Ho, int Wo, int align_corners, void *plan,
                                 size_t plan_bytes, lwg_stream_t stream);
LWG_API int lwg_grid_sample_backward_planned(const float *dy, int C, int xn, int H, int W, int n, int Ho, int Wo, const void *plan,
                                             size_t plan_bytes, float *dx, lwg_stream_t stream);
/* torch.optim.Adam step (no weight decay, no amsgrad) on a flat fp32 device tensor; step counts from 1 */
LWG_API int lwg_adam_update(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n, long step, float lr,
                            float beta1, float beta2, float eps, lwg_stream_t stream);
/* The same update with the step count in device memory: `step_state` is three 8-byte words on the device -- the count t
 * (int64) and beta1^t, beta2^t (doubles); {0, 1.0, 1.0} before the first step -- which the call advances on the stream (with the
 * betas it is given: keep them fixed) and then uses for the bias corrections.  Nothing about the step is baked into launch
 * arguments, so a captured HIP graph of a training iteration replays correctly. */
LWG_API int lwg_adam_update_device_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n,
                                        void *step_state, float lr, float beta1, float beta2, float eps, lwg_stream_t stream);
/* lwg_discriminator_adam_step keeps its step count on the device as well (on != 0; synchronous, call outside a capture, with the
 * betas the following steps will use): see lwg_adam_update_device_step.  Switching it off copies the count back. */
LWG_API int lwg_discriminator_use_device_step(lwg_discriminator *d, int on, float beta1, float beta2);

/* Test hook: copies an internal scratch buffer (device to device) after inference/swap/encode_src.
 * which: 0..2 = concat buffers cat[l] (bs, is>>l, is>>l, 2*conv_dim<<l)  [skip half | decoder half],
 *        3 = residual trunk output (bs, is/8, is/8, 8*conv_dim), 4..5 = skipper outputs 0..1,
 *        6 = raw (pre-norm) output of the last conv, 7..9 = resized flows at is/2, is/4, is/8.
 * Copies min(n_floats, buffer size) floats. */
LWG_API int lwg_generator_peek(lwg_generator *g, int which, float *dst, size_t n_floats, lwg_stream_t stream);

/* Kernel timing for bench.py's roofline: when enabled, every launch of an implicit-GEMM conv kernel is bracketed by
 * HIP events on the launch stream.  The conv runs as one of lwg_generator_profile_variants() kernel instantiations
 * (names as rocprofv3 prints them: lwg_generator_profile_variant_name).  read() synchronises the events and returns,
 * for one variant (or all of them with variant = -1), the launch count, the summed duration and the algorithmic
 * FLOPs (2*M*N*K of the convolution with its real taps and channels; zero padding counted as cuDNN would count it,
 * the kernel's own K/channel padding not counted).  profile(enable) clears the record. */
LWG_API int lwg_generator_profile(lwg_generator *g, int enable);
LWG_API int lwg_generator_profile_variants(void);
LWG_API const char *lwg_generator_profile_variant_name(int variant);
LWG_API int lwg_generator_profile_read(lwg_generator *g, int variant, int *launches, double *total_ms,
                                       double *total_flops);

/* Measurement hook (tools/conv_trace.py; no reference counterpart -- rocprofv3's thread trace has no decoder in this
 * image): with a device buffer set, every launch of the dominant 128-channel bf16x3 conv kernel runs its instrumented
 * twin (same code + s_memtime reads around the stage's data wait and barrier) and appends one block of
 * [workgroup][wave][8] uint64 {start, loop start, loop end, end, cycles in the data wait, cycles in the barrier,
 * stages, hw id}.  trace_launch(i): {byte offset, grid x, y, z, waves, stages, Cin, Cout, Hm, N} of traced launch i, or
 * LWG_ERR_INVALID_ARG past the last one.  conv_trace(NULL, 0) switches it off.  Process-wide, not thread-safe. */
LWG_API int lwg_conv_trace(void *device_buffer, size_t bytes);
LWG_API int lwg_conv_trace_launch(int index, long long *info10);

#ifdef __cplusplus
}
#endif
#endif /* LWG_H_ */
