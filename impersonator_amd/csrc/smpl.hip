// smpl.hip -- SMPL body model forward (linear blend skinning) for gfx950.
//
// Replaces the ~10 small GEMMs, the 24x cat/pad/stack traffic and the 23-step Python loop of
// networks/batch_smpl.py:285-375 (helpers batch_rodrigues :64-101, batch_global_rigid_transformation :129-218)
// with three launches per batch of frames:
//   smpl_pose_kernel   one wave per frame: Rodrigues for the 24 joints, pose feature (R - I), rest joints
//                      J = J_template + beta @ J_shapedirs (the joint regressor is linear in the shape, so the
//                      6890-vertex regression is folded into two tiny host-precomputed tables), the kinematic
//                      chain walked with 12 lanes (one per element of the 3x4 affine product) and the relative
//                      bone transforms A (batch_smpl.py:193-216).
//   smpl_verts_kernel  one lane per vertex coordinate and four frames: shape blend (10 terms), pose blend (207 terms,
//                      the only real traffic: posedirs, 17 MB, streamed bs/4 times), skinning T = sum_j W_j A_j,
//                      verts = T [v;1].
//   smpl_joints_kernel 19 keypoints = joint_regressor^T verts (block reduction), part of the reference's
//                      get_details() dictionary (hmr.py:302-330).
// HBM-bound integer-free float work; no GEMM reshaping (M = batch is 1..8).
#include "common.h"

namespace lwg {
namespace {

constexpr int NJ = 24;        // SMPL joints
constexpr int NPF = 207;      // pose feature length = 23 * 9
static_assert(NPF % 23 == 0, "the pose-blend loop walks 23 terms at a time");

// The working type R of the three kernels: float = the reference's own arithmetic (fp32 tensors, networks/batch_smpl.py), every
// multiply-add one rounding; double = the "compensated" mode (lwg_smpl_forward_f64): the same expressions on the same fp32
// model tensors evaluated in fp64 and rounded to fp32 once at the end -- the correctly rounded value of the function the
// reference's code defines, which any fp32 evaluation (the reference's included) misses by its own ~1e-6 of summation
// noise.  That noise matters downstream: the rasteriser's barycentric weights amplify a 1e-6 vertex difference to 1e-3 in
// the flow.  Cost: nothing measurable (the kernels are latency-bound, 0.02 GFLOP per frame).
template <typename R> struct mathx;
template <> struct mathx<float> {
    static __device__ float sqrt_(float x) { return sqrtf(x); }
    static __device__ float cos_(float x) { return cosf(x); }
    static __device__ float sin_(float x) { return sinf(x); }
    static __device__ float fma_(float a, float b, float c) { return fmaf(a, b, c); }
};
template <> struct mathx<double> {
    static __device__ double sqrt_(double x) { return sqrt(x); }
    static __device__ double cos_(double x) { return cos(x); }
    static __device__ double sin_(double x) { return sin(x); }
    static __device__ double fma_(double a, double b, double c) { return fma(a, b, c); }
};

// workspace: pose feature (bs, 207) then A (bs, 24, 12), both of type R
template <typename R, typename JT>
__global__ __launch_bounds__(64) void smpl_pose_kernel(const float *__restrict__ theta, int nb,
                                                       const JT *__restrict__ J_template,
                                                       const JT *__restrict__ J_shapedirs,
                                                       const int *__restrict__ parents, R *__restrict__ pf,
                                                       R *__restrict__ A_out, float *__restrict__ Rs_out)
{
    using M = mathx<R>;
    __shared__ R Rm[NJ][9];
    __shared__ R J[NJ][3];
    __shared__ R G[NJ][12];   // world transforms, row-major 3x4
    const int b = blockIdx.x, l = threadIdx.x;
    const float *th = theta + (size_t)b * (3 + 72 + nb);
    const float *pose = th + 3, *beta = th + 75;

    if (l < NJ) {
        // batch_smpl.py:86-100: angle = ||r + 1e-8||, axis = r / angle
        const R rx = pose[3 * l], ry = pose[3 * l + 1], rz = pose[3 * l + 2];
        const R eps = (R)1e-8f;
        const R ax = rx + eps, ay = ry + eps, az = rz + eps;
        const R angle = M::sqrt_(ax * ax + ay * ay + az * az);
        const R x = rx / angle, y = ry / angle, z = rz / angle;
        const R c = M::cos_(angle), s = M::sin_(angle), t = (R)1 - c;
        R m[9];
        m[0] = c + t * x * x;     m[1] = t * x * y - s * z; m[2] = t * x * z + s * y;
        m[3] = t * y * x + s * z; m[4] = c + t * y * y;     m[5] = t * y * z - s * x;
        m[6] = t * z * x - s * y; m[7] = t * z * y + s * x; m[8] = c + t * z * z;
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            Rm[l][e] = m[e];
            if (Rs_out) Rs_out[((size_t)b * NJ + l) * 9 + e] = (float)m[e];
            if (l >= 1) pf[(size_t)b * NPF + (l - 1) * 9 + e] = m[e] - ((e == 0 || e == 4 || e == 8) ? (R)1 : (R)0);
        }
    }
    for (int i = l; i < NJ * 3; i += 64) {
        R v = J_template[i];
        for (int k = 0; k < nb; ++k) v += (R)beta[k] * (R)J_shapedirs[k * NJ * 3 + i];
        J[i / 3][i % 3] = v;
    }
    __syncthreads();

    // kinematic chain (batch_smpl.py:189-197): G_i = G_parent * [R_i | J_i - J_parent]; 12 lanes, one per element
    const int r = l >> 2, cidx = l & 3;
    if (l < 12) G[0][l] = cidx < 3 ? Rm[0][r * 3 + cidx] : J[0][r];
    __syncthreads();
    for (int i = 1; i < NJ; ++i) {
        const int p = parents[i];
        if (l < 12) {
            R v;
            if (cidx < 3) {
                v = G[p][r * 4 + 0] * Rm[i][0 * 3 + cidx] + G[p][r * 4 + 1] * Rm[i][1 * 3 + cidx] +
                    G[p][r * 4 + 2] * Rm[i][2 * 3 + cidx];
            } else {
                const R dx = J[i][0] - J[p][0], dy = J[i][1] - J[p][1], dz = J[i][2] - J[p][2];
                v = G[p][r * 4 + 0] * dx + G[p][r * 4 + 1] * dy + G[p][r * 4 + 2] * dz + G[p][r * 4 + 3];
            }
            G[i][l] = v;
        }
        __syncthreads();
    }
    // relative transforms (batch_smpl.py:206-216): A = G - [0 | G_rot * J]
    for (int i = l; i < NJ * 12; i += 64) {
        const int j = i / 12, e = i % 12, rr = e >> 2, cc = e & 3;
        R v = G[j][e];
        if (cc == 3) v -= G[j][rr * 4 + 0] * J[j][0] + G[j][rr * 4 + 1] * J[j][1] + G[j][rr * 4 + 2] * J[j][2];
        A_out[((size_t)b * NJ + j) * 12 + e] = v;
    }
}

// One lane per vertex COORDINATE (v*3 + c) and up to VF frames of the batch: the pose-blend tables (posedirs, 207 rows
// of nv*3 floats -- the only real traffic of SMPL) are read with fully coalesced 4-byte-per-lane loads and every
// loaded value feeds VF frames' accumulators, so a batch streams them bs/VF times instead of bs times.  The three
// coordinates of a vertex meet through LDS for the skinning: lane (v, c) builds row c of T = sum_j W_j A_j (24 x 4
// terms instead of 24 x 12) and writes verts[v][c].  Every multiply-add is an explicit fmaf: left to the compiler,
// the four frames of a lane were contracted differently (packed v_pk_fma for some, mul + add for others) and a
// frame's vertices depended on its position in the batch in the last bit.  With fmaf every frame sees one fixed
// operation sequence: results do not depend on the batch size or on the frame's position (tested bit for bit).
constexpr int VB = 64;          // vertices per workgroup (192 lanes)
constexpr int VF = 4;           // frames per lane
template <typename R>
__global__ __launch_bounds__(3 * VB) void smpl_verts_kernel(const float *__restrict__ theta, int nb, int nv, int bs,
                                                            const float *__restrict__ v_template,
                                                            const float *__restrict__ shapedirs,
                                                            const float *__restrict__ posedirs,
                                                            const float *__restrict__ weights,
                                                            const R *__restrict__ pf, const R *__restrict__ A,
                                                            float *__restrict__ verts)
{
    using M = mathx<R>;
    __shared__ R s_pf[VF][NPF];
    __shared__ R s_A[VF][NJ * 12];
    __shared__ R s_beta[VF][16];
    __shared__ R s_p[VF][3 * VB];
    const int b0 = blockIdx.y * VF, tid = threadIdx.x;
    for (int i = tid; i < VF * NPF; i += 3 * VB) {
        const int f = i / NPF, k = i - f * NPF;
        s_pf[f][k] = b0 + f < bs ? pf[(size_t)(b0 + f) * NPF + k] : (R)0;
    }
    for (int i = tid; i < VF * NJ * 12; i += 3 * VB) {
        const int f = i / (NJ * 12), k = i - f * NJ * 12;
        s_A[f][k] = b0 + f < bs ? A[(size_t)(b0 + f) * NJ * 12 + k] : (R)0;
    }
    if (tid < VF * 16) {
        const int f = tid >> 4, k = tid & 15;
        s_beta[f][k] = (b0 + f < bs && k < nb) ? (R)theta[(size_t)(b0 + f) * (75 + nb) + 75 + k] : (R)0;
    }
    __syncthreads();
    const size_t row = (size_t)nv * 3;
    const int e = blockIdx.x * 3 * VB + tid;
    const bool ok = e < nv * 3;
    R p[VF];
    if (ok) {
        const R vt = v_template[e];
#pragma unroll
        for (int f = 0; f < VF; ++f) p[f] = vt;
        for (int k = 0; k < nb; ++k) {
            const R sd = shapedirs[k * row + e];
#pragma unroll
            for (int f = 0; f < VF; ++f) p[f] = M::fma_(s_beta[f][k], sd, p[f]);
        }
        // 207 = 9 x 23: 23 independent loads in flight per lane (the kernel runs at ~3 waves per CU: latency-bound)
        for (int k0 = 0; k0 < NPF; k0 += 23) {
            float pd[23];
#pragma unroll
            for (int j = 0; j < 23; ++j) pd[j] = posedirs[(k0 + j) * row + e];
#pragma unroll
            for (int j = 0; j < 23; ++j)
#pragma unroll
                for (int f = 0; f < VF; ++f) p[f] = M::fma_(s_pf[f][k0 + j], (R)pd[j], p[f]);
        }
#pragma unroll
        for (int f = 0; f < VF; ++f) s_p[f][tid] = p[f];
    }
    __syncthreads();
    if (!ok) return;
    const int v = e / 3, c = e - v * 3, vl = (tid / 3) * 3;
    R T[VF][4];
#pragma unroll
    for (int f = 0; f < VF; ++f)
#pragma unroll
        for (int q = 0; q < 4; ++q) T[f][q] = (R)0;
    const float *w = weights + (size_t)v * NJ;
#pragma unroll 4
    for (int j = 0; j < NJ; ++j) {
        const R wj = w[j];
#pragma unroll
        for (int f = 0; f < VF; ++f)
#pragma unroll
            for (int q = 0; q < 4; ++q) T[f][q] = M::fma_(wj, s_A[f][j * 12 + c * 4 + q], T[f][q]);
    }
#pragma unroll
    for (int f = 0; f < VF; ++f)
        if (b0 + f < bs)
            verts[((size_t)(b0 + f) * nv + v) * 3 + c] =
                (float)(M::fma_(T[f][2], s_p[f][vl + 2], M::fma_(T[f][1], s_p[f][vl + 1], T[f][0] * s_p[f][vl])) + T[f][3]);
}

// grid (n_out_joints, bs): joints[b][j][:] = sum_v verts[b][v][:] * joint_regressor[v][j]
template <typename R>
__global__ __launch_bounds__(256) void smpl_joints_kernel(const float *__restrict__ verts,
                                                          const float *__restrict__ jreg, int nv, int nj,
                                                          float *__restrict__ joints)
{
    __shared__ R red[3][4];
    const int j = blockIdx.x, b = blockIdx.y;
    R a0 = 0, a1 = 0, a2 = 0;
    for (int v = threadIdx.x; v < nv; v += 256) {
        const R w = jreg[(size_t)v * nj + j];
        const float *p = verts + ((size_t)b * nv + v) * 3;
        a0 += w * p[0];
        a1 += w * p[1];
        a2 += w * p[2];
    }
    for (int off = 32; off > 0; off >>= 1) {
        a0 += __shfl_down(a0, off);
        a1 += __shfl_down(a1, off);
        a2 += __shfl_down(a2, off);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[0][wave] = a0;
        red[1][wave] = a1;
        red[2][wave] = a2;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        joints[((size_t)b * nj + j) * 3 + c] = (float)(red[c][0] + red[c][1] + red[c][2] + red[c][3]);
    }
}

// ---- host glue of the per-frame prelude as two elementwise kernels (models/imitator.py:216-234 swap_smpl, the slicing
// of networks/hmr.py:302-330 get_details, networks/batch_smpl.py:221-234 batch_orth_proj_idrot): a dozen tiny torch
// launches per round otherwise.  One thread per output float; operation order as the reference's tensor expressions.
__global__ void smpl_swap_kernel(const float *__restrict__ tgt, int n, int nbeta, int strategy, const float *__restrict__ src_cam,
                                 const float *__restrict__ src_shape, const float *__restrict__ first_cam, float *__restrict__ theta,
                                 float *__restrict__ cam, float *__restrict__ pose, float *__restrict__ shape)
{
    const int D = 75 + nbeta, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * D) return;
    const int f = i / D, c = i - f * D;
    float v;
    if (c < 3) {
        const float t = tgt[(size_t)f * D + c];
        if (strategy == 1) v = c == 0 ? src_cam[0] : src_cam[c] + (t - first_cam[c]);   // 'smooth': cam[:, 1:] += tgt - first
        else if (strategy == 2) v = src_cam[c];                                          // 'source'
        else v = t;                                                                      // 'copy' / theta given as is
        cam[f * 3 + c] = v;
    } else if (c < 75) {
        v = tgt[(size_t)f * D + c];
        pose[f * 72 + c - 3] = v;
    } else {
        v = strategy == 0 ? tgt[(size_t)f * D + c] : src_shape[c - 75];
        shape[f * nbeta + c - 75] = v;
    }
    theta[i] = v;
}

__global__ void smpl_j2d_kernel(const float *__restrict__ j3d, const float *__restrict__ cam, int n, int nj, float *__restrict__ j2d)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * nj * 2) return;
    const int f = i / (nj * 2), r = i - f * nj * 2, j = r >> 1, c = r & 1;
    j2d[i] = cam[f * 3] * (j3d[((size_t)f * nj + j) * 3 + c] + cam[f * 3 + 1 + c]);
}

// Viewer.rotate_trans (models/viewer.py:240-247): out = X @ R + t for every vertex, R row-major (3,3).  The rotation and the
// translation travel as kernel arguments (a dozen floats: no upload, nothing for a graph replay to re-read from the host).
struct RigidArgs { float R[9]; float t[3]; };
__global__ void rotate_translate_kernel(const float *__restrict__ x, long n, RigidArgs a, float *__restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x0 = x[3 * i], x1 = x[3 * i + 1], x2 = x[3 * i + 2];
#pragma unroll
    for (int j = 0; j < 3; ++j) out[3 * i + j] = fmaf(x2, a.R[6 + j], fmaf(x1, a.R[3 + j], x0 * a.R[j])) + a.t[j];
}

}  // namespace
}  // namespace lwg

using namespace lwg;

extern "C" {

int lwg_smpl_swap(const float *tgt_smpl, int bs, int num_betas, int strategy, const float *src_cam, const float *src_shape,
                  const float *first_cam, float *theta, float *cam, float *pose, float *shape, lwg_stream_t stream)
{
    LWG_REQUIRE(tgt_smpl && theta && cam && pose && shape, "smpl_swap: NULL argument");
    LWG_REQUIRE(bs > 0 && num_betas > 0 && strategy >= 0 && strategy <= 3, "smpl_swap: bad arguments");
    LWG_REQUIRE(strategy == 0 || strategy == 3 || src_cam, "smpl_swap: the 'smooth' and 'source' strategies need src_cam");
    LWG_REQUIRE(strategy == 0 || src_shape, "smpl_swap: src_shape missing");
    LWG_REQUIRE(strategy != 1 || first_cam, "smpl_swap: 'smooth' needs first_cam");
    const int total = bs * (75 + num_betas);
    smpl_swap_kernel<<<ceil_div(total, 256), 256, 0, as_stream(stream)>>>(tgt_smpl, bs, num_betas, strategy, src_cam, src_shape, first_cam,
                                                                          theta, cam, pose, shape);
    LWG_LAUNCH_CHECK("smpl_swap_kernel");
    return LWG_OK;
}

int lwg_smpl_project_joints(const float *j3d, const float *cam, int bs, int num_joints, float *j2d, lwg_stream_t stream)
{
    LWG_REQUIRE(j3d && cam && j2d && bs > 0 && num_joints > 0, "smpl_project_joints: bad arguments");
    smpl_j2d_kernel<<<ceil_div((long)bs * num_joints * 2, 256), 256, 0, as_stream(stream)>>>(j3d, cam, bs, num_joints, j2d);
    LWG_LAUNCH_CHECK("smpl_j2d_kernel");
    return LWG_OK;
}

int lwg_rotate_translate(const float *x, long n, const float *R9, const float *t3, float *out, lwg_stream_t stream)
{
    LWG_REQUIRE(x && R9 && t3 && out && n > 0, "rotate_translate: bad arguments");
    RigidArgs a;
    for (int i = 0; i < 9; ++i) a.R[i] = R9[i];
    for (int i = 0; i < 3; ++i) a.t[i] = t3[i];
    rotate_translate_kernel<<<ceil_div(n, 256), 256, 0, as_stream(stream)>>>(x, n, a, out);
    LWG_LAUNCH_CHECK("rotate_translate_kernel");
    return LWG_OK;
}

size_t lwg_smpl_workspace_bytes(int bs) { return bs > 0 ? (size_t)bs * (NPF + NJ * 12) * sizeof(double) : 0; }

}  // extern "C"

namespace {
template <typename R, typename JT>
int smpl_forward_impl(const float *theta, int bs, int num_betas, int nv, int num_out_joints, const float *v_template,
                      const float *shapedirs, const float *posedirs, const JT *J_template, const JT *J_shapedirs,
                      const int32_t *parents, const float *weights, const float *joint_regressor, float *verts, float *joints,
                      float *Rs, void *workspace, size_t workspace_bytes, lwg_stream_t stream)
{
    LWG_REQUIRE(theta && v_template && shapedirs && posedirs && J_template && J_shapedirs && parents && weights && verts,
                "smpl_forward: NULL argument");
    LWG_REQUIRE(bs > 0 && nv > 0 && num_betas > 0 && num_betas <= 16, "smpl_forward: bad sizes (bs=%d nv=%d betas=%d)",
                bs, nv, num_betas);
    LWG_REQUIRE(!joints || (joint_regressor && num_out_joints > 0), "smpl_forward: joints requested without a regressor");
    if (!workspace || workspace_bytes < lwg_smpl_workspace_bytes(bs))
        LWG_FAIL(LWG_ERR_WORKSPACE, "smpl_forward: workspace needs %zu bytes", lwg_smpl_workspace_bytes(bs));
    hipStream_t st = as_stream(stream);
    R *pf = static_cast<R *>(workspace);
    R *A = pf + (size_t)bs * NPF;
    smpl_pose_kernel<R, JT><<<bs, 64, 0, st>>>(theta, num_betas, J_template, J_shapedirs, parents, pf, A, Rs);
    LWG_LAUNCH_CHECK("smpl_pose_kernel");
    smpl_verts_kernel<R><<<dim3(ceil_div((long)nv * 3, 3 * VB), ceil_div(bs, VF)), 3 * VB, 0, st>>>(
        theta, num_betas, nv, bs, v_template, shapedirs, posedirs, weights, pf, A, verts);
    LWG_LAUNCH_CHECK("smpl_verts_kernel");
    if (joints) {
        smpl_joints_kernel<R><<<dim3(num_out_joints, bs), 256, 0, st>>>(verts, joint_regressor, nv, num_out_joints, joints);
        LWG_LAUNCH_CHECK("smpl_joints_kernel");
    }
    return LWG_OK;
}
}  // namespace

extern "C" {

int lwg_smpl_forward(const float *theta, int bs, int num_betas, int nv, int num_out_joints, const float *v_template,
                     const float *shapedirs, const float *posedirs, const float *J_template,
                     const float *J_shapedirs, const int32_t *parents, const float *weights,
                     const float *joint_regressor, float *verts, float *joints, float *Rs, void *workspace,
                     size_t workspace_bytes, lwg_stream_t stream)
{
    return smpl_forward_impl<float, float>(theta, bs, num_betas, nv, num_out_joints, v_template, shapedirs, posedirs, J_template,
                                           J_shapedirs, parents, weights, joint_regressor, verts, joints, Rs, workspace,
                                           workspace_bytes, stream);
}

int lwg_smpl_forward_f64(const float *theta, int bs, int num_betas, int nv, int num_out_joints, const float *v_template,
                         const float *shapedirs, const float *posedirs, const double *J_template,
                         const double *J_shapedirs, const int32_t *parents, const float *weights,
                         const float *joint_regressor, float *verts, float *joints, float *Rs, void *workspace,
                         size_t workspace_bytes, lwg_stream_t stream)
{
    return smpl_forward_impl<double, double>(theta, bs, num_betas, nv, num_out_joints, v_template, shapedirs, posedirs, J_template,
                                             J_shapedirs, parents, weights, joint_regressor, verts, joints, Rs, workspace,
                                             workspace_bytes, stream);
}

}  // extern "C"
