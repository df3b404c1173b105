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
//   smpl_verts_kernel  one lane per (frame, vertex): shape blend (10 terms), pose blend (207 terms, the only
//                      real traffic: posedirs, 17 MB, L2-resident across the batch), skinning T = sum_j W_j A_j,
//                      verts = T [v;1].
//   smpl_joints_kernel 19 keypoints = joint_regressor^T verts (block reduction), part of the reference's
//                      get_details() dictionary (hmr.py:302-330).
// HBM-bound integer-free float work; no GEMM reshaping (M = batch is 1..8).
#include "common.h"

namespace lwg {
namespace {

constexpr int NJ = 24;        // SMPL joints
constexpr int NPF = 207;      // pose feature length = 23 * 9

// workspace: pose feature (bs, 207) then A (bs, 24, 12)
__global__ __launch_bounds__(64) void smpl_pose_kernel(const float *__restrict__ theta, int nb,
                                                       const float *__restrict__ J_template,
                                                       const float *__restrict__ J_shapedirs,
                                                       const int *__restrict__ parents, float *__restrict__ pf,
                                                       float *__restrict__ A_out, float *__restrict__ Rs_out)
{
    __shared__ float R[NJ][9];
    __shared__ float J[NJ][3];
    __shared__ float G[NJ][12];   // world transforms, row-major 3x4
    const int b = blockIdx.x, l = threadIdx.x;
    const float *th = theta + (size_t)b * (3 + 72 + nb);
    const float *pose = th + 3, *beta = th + 75;

    if (l < NJ) {
        // batch_smpl.py:86-100: angle = ||r + 1e-8||, axis = r / angle
        const float rx = pose[3 * l], ry = pose[3 * l + 1], rz = pose[3 * l + 2];
        const float ax = rx + 1e-8f, ay = ry + 1e-8f, az = rz + 1e-8f;
        const float angle = sqrtf(ax * ax + ay * ay + az * az);
        const float x = rx / angle, y = ry / angle, z = rz / angle;
        const float c = cosf(angle), s = sinf(angle), t = 1.f - c;
        float m[9];
        m[0] = c + t * x * x;     m[1] = t * x * y - s * z; m[2] = t * x * z + s * y;
        m[3] = t * y * x + s * z; m[4] = c + t * y * y;     m[5] = t * y * z - s * x;
        m[6] = t * z * x - s * y; m[7] = t * z * y + s * x; m[8] = c + t * z * z;
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            R[l][e] = m[e];
            if (Rs_out) Rs_out[((size_t)b * NJ + l) * 9 + e] = m[e];
            if (l >= 1) pf[(size_t)b * NPF + (l - 1) * 9 + e] = m[e] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
        }
    }
    for (int i = l; i < NJ * 3; i += 64) {
        float v = J_template[i];
        for (int k = 0; k < nb; ++k) v += beta[k] * J_shapedirs[k * NJ * 3 + i];
        J[i / 3][i % 3] = v;
    }
    __syncthreads();

    // kinematic chain (batch_smpl.py:189-197): G_i = G_parent * [R_i | J_i - J_parent]; 12 lanes, one per element
    const int r = l >> 2, cidx = l & 3;
    if (l < 12) G[0][l] = cidx < 3 ? R[0][r * 3 + cidx] : J[0][r];
    __syncthreads();
    for (int i = 1; i < NJ; ++i) {
        const int p = parents[i];
        if (l < 12) {
            float v;
            if (cidx < 3) {
                v = G[p][r * 4 + 0] * R[i][0 * 3 + cidx] + G[p][r * 4 + 1] * R[i][1 * 3 + cidx] +
                    G[p][r * 4 + 2] * R[i][2 * 3 + cidx];
            } else {
                const float dx = J[i][0] - J[p][0], dy = J[i][1] - J[p][1], dz = J[i][2] - J[p][2];
                v = G[p][r * 4 + 0] * dx + G[p][r * 4 + 1] * dy + G[p][r * 4 + 2] * dz + G[p][r * 4 + 3];
            }
            G[i][l] = v;
        }
        __syncthreads();
    }
    // relative transforms (batch_smpl.py:206-216): A = G - [0 | G_rot * J]
    for (int i = l; i < NJ * 12; i += 64) {
        const int j = i / 12, e = i % 12, rr = e >> 2, cc = e & 3;
        float v = G[j][e];
        if (cc == 3) v -= G[j][rr * 4 + 0] * J[j][0] + G[j][rr * 4 + 1] * J[j][1] + G[j][rr * 4 + 2] * J[j][2];
        A_out[((size_t)b * NJ + j) * 12 + e] = v;
    }
}

__global__ __launch_bounds__(256) void smpl_verts_kernel(const float *__restrict__ theta, int nb, int nv,
                                                         const float *__restrict__ v_template,
                                                         const float *__restrict__ shapedirs,
                                                         const float *__restrict__ posedirs,
                                                         const float *__restrict__ weights,
                                                         const float *__restrict__ pf, const float *__restrict__ A,
                                                         float *__restrict__ verts)
{
    __shared__ float s_pf[NPF];
    __shared__ float s_A[NJ * 12];
    __shared__ float s_beta[16];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < NPF; i += 256) s_pf[i] = pf[(size_t)b * NPF + i];
    for (int i = threadIdx.x; i < NJ * 12; i += 256) s_A[i] = A[(size_t)b * NJ * 12 + i];
    if (threadIdx.x < nb && threadIdx.x < 16) s_beta[threadIdx.x] = theta[(size_t)b * (75 + nb) + 75 + threadIdx.x];
    __syncthreads();
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= nv) return;
    const size_t row = (size_t)nv * 3, col = (size_t)v * 3;

    float p0 = v_template[col], p1 = v_template[col + 1], p2 = v_template[col + 2];
    for (int k = 0; k < nb; ++k) {
        const float *sd = shapedirs + k * row + col;
        p0 += s_beta[k] * sd[0];
        p1 += s_beta[k] * sd[1];
        p2 += s_beta[k] * sd[2];
    }
#pragma unroll 4
    for (int k = 0; k < NPF; ++k) {
        const float *pd = posedirs + k * row + col;
        const float f = s_pf[k];
        p0 += f * pd[0];
        p1 += f * pd[1];
        p2 += f * pd[2];
    }
    float T[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) T[e] = 0.f;
    const float *w = weights + (size_t)v * NJ;
#pragma unroll 4
    for (int j = 0; j < NJ; ++j) {
        const float wj = w[j];
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] += wj * s_A[j * 12 + e];
    }
    float *o = verts + ((size_t)b * nv + v) * 3;
    o[0] = T[0] * p0 + T[1] * p1 + T[2] * p2 + T[3];
    o[1] = T[4] * p0 + T[5] * p1 + T[6] * p2 + T[7];
    o[2] = T[8] * p0 + T[9] * p1 + T[10] * p2 + T[11];
}

// grid (n_out_joints, bs): joints[b][j][:] = sum_v verts[b][v][:] * joint_regressor[v][j]
__global__ __launch_bounds__(256) void smpl_joints_kernel(const float *__restrict__ verts,
                                                          const float *__restrict__ jreg, int nv, int nj,
                                                          float *__restrict__ joints)
{
    __shared__ float red[3][4];
    const int j = blockIdx.x, b = blockIdx.y;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int v = threadIdx.x; v < nv; v += 256) {
        const float w = jreg[(size_t)v * nj + j];
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
        joints[((size_t)b * nj + j) * 3 + c] = red[c][0] + red[c][1] + red[c][2] + red[c][3];
    }
}

}  // namespace
}  // namespace lwg

using namespace lwg;

extern "C" {

size_t lwg_smpl_workspace_bytes(int bs) { return bs > 0 ? (size_t)bs * (NPF + NJ * 12) * sizeof(float) : 0; }

int lwg_smpl_forward(const float *theta, int bs, int num_betas, int nv, int num_out_joints, const float *v_template,
                     const float *shapedirs, const float *posedirs, const float *J_template,
                     const float *J_shapedirs, const int32_t *parents, const float *weights,
                     const float *joint_regressor, float *verts, float *joints, float *Rs, void *workspace,
                     size_t workspace_bytes, lwg_stream_t stream)
{
    LWG_REQUIRE(theta && v_template && shapedirs && posedirs && J_template && J_shapedirs && parents && weights && verts,
                "smpl_forward: NULL argument");
    LWG_REQUIRE(bs > 0 && nv > 0 && num_betas > 0 && num_betas <= 16, "smpl_forward: bad sizes (bs=%d nv=%d betas=%d)",
                bs, nv, num_betas);
    LWG_REQUIRE(!joints || (joint_regressor && num_out_joints > 0), "smpl_forward: joints requested without a regressor");
    if (!workspace || workspace_bytes < lwg_smpl_workspace_bytes(bs))
        LWG_FAIL(LWG_ERR_WORKSPACE, "smpl_forward: workspace needs %zu bytes", lwg_smpl_workspace_bytes(bs));
    hipStream_t st = as_stream(stream);
    float *pf = static_cast<float *>(workspace);
    float *A = pf + (size_t)bs * NPF;
    smpl_pose_kernel<<<bs, 64, 0, st>>>(theta, num_betas, J_template, J_shapedirs, parents, pf, A, Rs);
    LWG_LAUNCH_CHECK("smpl_pose_kernel");
    smpl_verts_kernel<<<dim3(ceil_div(nv, 256), bs), 256, 0, st>>>(theta, num_betas, nv, v_template, shapedirs,
                                                                   posedirs, weights, pf, A, verts);
    LWG_LAUNCH_CHECK("smpl_verts_kernel");
    if (joints) {
        smpl_joints_kernel<<<dim3(num_out_joints, bs), 256, 0, st>>>(verts, joint_regressor, nv, num_out_joints, joints);
        LWG_LAUNCH_CHECK("smpl_joints_kernel");
    }
    return LWG_OK;
}

}  // extern "C"
