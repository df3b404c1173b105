"""SMPL body model with the reference's interface (networks/batch_smpl.py:229-375).

Row "f1 / next" of the hot-path scope: per frame it is ~0.02 GFLOP of small GEMMs
in front of the rasteriser, so this round keeps it as PyTorch-ROCm tensor ops (device memory plumbing);
everything downstream of the vertices is liblwg.  The parameters come from `smpl_model.pkl` when present
(README.md:48-68 of the reference: a download) or from a seeded synthetic model with the same tensor shapes.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from ..utils import synthetic
from ..utils.util import load_pickle_file

# SMPL kinematic tree (kintree_table[0] of smpl_model.pkl)
SMPL_PARENTS = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21], np.int32)


def synthetic_smpl_params(seed=0):
    """A stand-in with SMPL's shapes: 6890 vertices, 10 betas, 24 joints, 207 pose features, 19 keypoints."""
    rng = np.random.default_rng(seed + 4242)
    rest, faces = synthetic.body_mesh()
    nv = rest.shape[0]
    # 24 joints spread along the body axis with small lateral offsets
    jy = np.linspace(-0.75, 0.75, 24)
    joints = np.stack([0.08 * np.sin(np.arange(24) * 1.7), jy, 0.03 * np.cos(np.arange(24) * 2.3)], 1)
    d2 = ((rest[:, None, :].astype(np.float64) - joints[None]) ** 2).sum(-1)
    w = np.exp(-d2 / (2 * 0.08 ** 2))
    w /= w.sum(1, keepdims=True)
    jr = np.exp(-d2 / (2 * 0.05 ** 2)).T
    jr /= jr.sum(1, keepdims=True)
    coco = np.exp(-d2[:, :19] / (2 * 0.06 ** 2)).T
    coco /= coco.sum(1, keepdims=True)
    smooth = np.sin(rest.astype(np.float64) @ rng.normal(0, 3.0, (3, 10)))          # (nv,10)
    shapedirs = (smooth[:, None, :] * rng.normal(0, 0.01, (1, 3, 10)))              # (nv,3,10)
    posedirs = rng.normal(0, 1e-3, (nv, 3, 207))
    return dict(v_template=rest.astype(np.float64), f=faces.astype(np.int64), shapedirs=shapedirs,
                J_regressor=jr, posedirs=posedirs, kintree_table=np.stack([SMPL_PARENTS, np.arange(24)]),
                weights=w, cocoplus_regressor=coco)


def batch_rodrigues(theta):
    """networks/batch_smpl.py:64-101: axis-angle (N,3) -> rotation matrices (N,3,3)."""
    n = theta.shape[0]
    angle = torch.norm(theta + 1e-8, p=2, dim=1, keepdim=True)
    r = (theta / angle).unsqueeze(-1)
    angle = angle.unsqueeze(-1)
    cos, sin = torch.cos(angle), torch.sin(angle)
    outer = torch.matmul(r, r.permute(0, 2, 1))
    eye = torch.eye(3, dtype=theta.dtype, device=theta.device).unsqueeze(0).expand(n, 3, 3)
    rx, ry, rz = r[:, 0, 0], r[:, 1, 0], r[:, 2, 0]
    zero = torch.zeros_like(rx)
    skew = torch.stack([zero, -rz, ry, rz, zero, -rx, -ry, rx, zero], dim=1).view(n, 3, 3)
    return cos * eye + (1 - cos) * outer + sin * skew


def batch_global_rigid_transformation(Rs, Js, parent):
    """networks/batch_smpl.py:129-218 (rotate_base=False): world transforms of the 24 joints."""
    n = Rs.shape[0]
    Js = Js.unsqueeze(-1)

    def make_A(R, t):
        top = torch.cat([R, t], dim=2)
        bottom = torch.tensor([0., 0., 0., 1.], dtype=R.dtype, device=R.device).view(1, 1, 4).expand(n, 1, 4)
        return torch.cat([top, bottom], dim=1)

    results = [make_A(Rs[:, 0], Js[:, 0])]
    for i in range(1, parent.shape[0]):
        j_here = Js[:, i] - Js[:, parent[i]]
        results.append(torch.matmul(results[parent[i]], make_A(Rs[:, i], j_here)))
    results = torch.stack(results, dim=1)
    new_J = results[:, :, :3, 3]
    Js_w0 = torch.cat([Js, torch.zeros(n, 24, 1, 1, dtype=Rs.dtype, device=Rs.device)], dim=2)
    init_bone = torch.matmul(results, Js_w0)
    init_bone = torch.nn.functional.pad(init_bone, (3, 0, 0, 0, 0, 0, 0, 0))
    return new_J, results - init_bone


def batch_orth_proj_idrot(X, camera):
    """networks/batch_smpl.py:221-234."""
    return camera[:, None, 0:1] * (X[:, :, :2] + camera[:, None, 1:])


class SMPL(nn.Module):
    # arithmetic of the device path: "fp32" = the reference's own (fp32 tensors, one rounding per multiply-add);
    # "compensated" = every intermediate in fp64, ONE rounding to fp32 at the end (smpl.hip, lwg_smpl_forward_f64).  The
    # compensated vertices are the correctly rounded values of the function the reference's code defines; an fp32 evaluation
    # -- this one or the reference's -- sits ~1e-6 away from them by summation order alone, and the rasteriser amplifies
    # that to 1e-3 in the flow field.  Same speed (latency-bound kernels).  env LWG_SMPL_PRECISION.
    precision = "compensated"

    def __init__(self, pkl_path=None, rotate=False, params=None):
        super().__init__()
        if rotate:
            raise NotImplementedError("rotate_base is never enabled on the Imitator path")
        if params is None:
            if pkl_path is None or not os.path.exists(pkl_path):
                raise FileNotFoundError("SMPL model %s not found; pass params=synthetic_smpl_params()" % pkl_path)
            dd = load_pickle_file(pkl_path)
            params = dict(v_template=np.asarray(dd['v_template']), f=np.asarray(dd['f']),
                          shapedirs=np.asarray(dd['shapedirs']), J_regressor=np.asarray(dd['J_regressor'].todense()),
                          posedirs=np.asarray(dd['posedirs']), kintree_table=np.asarray(dd['kintree_table']),
                          weights=np.asarray(dd['weights']),
                          cocoplus_regressor=np.asarray(dd['cocoplus_regressor'].todense()))
        # liblwg reads these through raw pointers: row-major, whatever strides numpy's transposes had
        f32 = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32).contiguous()
        self.faces = torch.from_numpy(np.asarray(params['f']).astype(np.int32))
        self.register_buffer('v_template', f32(params['v_template']))
        self.size = [self.v_template.shape[0], 3]
        self.num_betas = params['shapedirs'].shape[-1]
        self.register_buffer('shapedirs', f32(np.reshape(params['shapedirs'], [-1, self.num_betas]).T))
        self.register_buffer('J_regressor', f32(np.asarray(params['J_regressor']).T))
        npose = params['posedirs'].shape[-1]
        self.register_buffer('posedirs', f32(np.reshape(params['posedirs'], [-1, npose]).T))
        self.parents = np.asarray(params['kintree_table'][0]).astype(np.int32)
        self.register_buffer('weights', f32(params['weights']))
        self.register_buffer('joint_regressor', f32(np.asarray(params['cocoplus_regressor']).T))
        self.register_buffer('parents_t', torch.from_numpy(self.parents.astype(np.int32)), persistent=False)
        # Derived buffers of the device path (liblwg, smpl.hip) -- the joint regression is linear in the shape and is folded once.
        # All NON-persistent: a state_dict holds exactly the reference's six keys (networks/batch_smpl.py:251-283: v_template,
        # shapedirs, J_regressor, posedirs, weights, joint_regressor), a strict load of a reference checkpoint works, and
        # _refresh_derived() rebuilds them whenever load_state_dict has replaced the tensors they are functions of.
        self._refresh_derived()
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._after_load(incompatible))
        self.precision = os.environ.get("LWG_SMPL_PRECISION", self.precision)
        self._ws = None

    def _refresh_derived(self):
        """J_template (24,3) / J_shapedirs (nb,72): J_regressor^T applied to v_template / shapedirs, folded in float64 and rounded
        (the fp32 kernels' operands); J_template_d / J_shapedirs_d: the same products kept in fp64 (the `compensated` kernels')."""
        dev = self.v_template.device
        jreg = self.J_regressor.detach().double().cpu().t()                                       # (24, nv), the fp32 values
        jt = jreg @ self.v_template.detach().double().cpu()                                        # (24, 3)
        js = torch.einsum('jv,kvc->kjc', jreg, self.shapedirs.detach().double().cpu().reshape(self.num_betas, -1, 3)) \
            .reshape(self.num_betas, -1)                                                           # (nb, 72)
        for name, val in (('J_template', jt.float()), ('J_shapedirs', js.float()), ('J_template_d', jt), ('J_shapedirs_d', js)):
            val = val.contiguous().to(dev)
            if name in self._buffers:
                self._buffers[name] = val
            else:
                self.register_buffer(name, val, persistent=False)

    def _after_load(self, incompatible):
        """load_state_dict post-hook: checkpoints written by earlier rounds of this repo carried the derived buffers as keys --
        they are accepted and ignored (never an `unexpected key` error) -- and the derived buffers follow the loaded tensors."""
        legacy = {'J_template', 'J_shapedirs', 'parents_t', 'J_template_d', 'J_shapedirs_d'}
        incompatible.unexpected_keys[:] = [k for k in incompatible.unexpected_keys if k.split('.')[-1] not in legacy]
        self._refresh_derived()

    def forward(self, beta, theta, get_skin=False):
        """networks/batch_smpl.py:285-375.  CUDA tensors run the fused HIP kernels of liblwg (smpl.hip);
        CPU tensors run the reference's tensor-op formulation (used by the CPU tests)."""
        if beta.is_cuda:
            return self.forward_device(beta, theta, get_skin)
        return self.forward_ops(beta, theta, get_skin)

    @torch.no_grad()
    def forward_device(self, beta, theta, get_skin=False):
        n = beta.shape[0]
        th = torch.cat([torch.zeros(n, 3, device=beta.device), theta.float(), beta.float()], dim=1)
        verts, joints, Rs = self.forward_theta(th)
        if get_skin:
            return verts, joints, Rs
        return joints

    @torch.no_grad()
    def forward_theta(self, theta):
        """theta (n, 3+72+num_betas) = [cam, pose, shape] on the GPU -> (verts, joints, Rs) in one liblwg call."""
        from .. import _lib
        lib = _lib.load()
        th = theta.float().contiguous()
        n, nv, dev = th.shape[0], self.size[0], th.device
        verts = torch.empty((n, nv, 3), device=dev, dtype=torch.float32)
        joints = torch.empty((n, self.joint_regressor.shape[1], 3), device=dev, dtype=torch.float32)
        Rs = torch.empty((n, 24, 3, 3), device=dev, dtype=torch.float32)
        need = lib.lwg_smpl_workspace_bytes(n)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        if self.precision not in ("fp32", "compensated"):
            raise ValueError("SMPL.precision must be 'fp32' or 'compensated', not %r" % (self.precision,))
        f64 = self.precision == "compensated"
        if f64 and (self.J_template_d.dtype != torch.float64 or self.J_shapedirs_d.dtype != torch.float64):
            # nn.Module.float() / .half() / .to(dtype) casts every floating buffer; the kernel reads these two as double
            # through raw pointers.  Restore them (exactly: they are functions of the fp32 buffers) instead of reading garbage.
            jreg = self.J_regressor.double().t()
            self.J_template_d = (jreg @ self.v_template.double()).contiguous()
            self.J_shapedirs_d = torch.einsum('jv,kvc->kjc', jreg, self.shapedirs.double().reshape(self.num_betas, -1, 3)) \
                .reshape(self.num_betas, -1).contiguous()
        for name in ("v_template", "shapedirs", "posedirs", "J_template", "J_shapedirs", "weights", "joint_regressor"):
            if getattr(self, name).dtype != torch.float32:
                raise TypeError("SMPL.%s is %s: the device kernels read fp32 (do not cast the SMPL module)" % (name, getattr(self, name).dtype))
        _lib.check((lib.lwg_smpl_forward_f64 if f64 else lib.lwg_smpl_forward)(
            _lib.ptr(th), n, self.num_betas, nv, joints.shape[1], _lib.ptr(self.v_template), _lib.ptr(self.shapedirs),
            _lib.ptr(self.posedirs), _lib.ptr(self.J_template_d if f64 else self.J_template),
            _lib.ptr(self.J_shapedirs_d if f64 else self.J_shapedirs), _lib.ptr(self.parents_t),
            _lib.ptr(self.weights), _lib.ptr(self.joint_regressor), _lib.ptr(verts), _lib.ptr(joints), _lib.ptr(Rs),
            _lib.ptr(self._ws), self._ws.numel(), _lib.stream_ptr()))
        return verts, joints, Rs

    def forward_ops(self, beta, theta, get_skin=False):
        """networks/batch_smpl.py:285-375 as tensor ops."""
        n = beta.shape[0]
        nv = self.size[0]
        v_shaped = torch.matmul(beta, self.shapedirs).view(-1, nv, 3) + self.v_template
        J = torch.stack([torch.matmul(v_shaped[:, :, k], self.J_regressor) for k in range(3)], dim=2)
        Rs = batch_rodrigues(theta.reshape(-1, 3)).view(-1, 24, 3, 3)
        pose_feature = (Rs[:, 1:] - torch.eye(3, device=beta.device)).view(-1, 207)
        v_posed = torch.matmul(pose_feature, self.posedirs).view(-1, nv, 3) + v_shaped
        _, A = batch_global_rigid_transformation(Rs, J, self.parents)
        W = self.weights.unsqueeze(0).expand(n, -1, -1)
        T = torch.matmul(W, A.view(n, 24, 16)).view(n, -1, 4, 4)
        v_homo = torch.cat([v_posed, torch.ones(n, nv, 1, dtype=torch.float32, device=beta.device)], dim=2)
        verts = torch.matmul(T, v_homo.unsqueeze(-1))[:, :, :3, 0]
        joints = torch.stack([torch.matmul(verts[:, :, k], self.joint_regressor) for k in range(3)], dim=2)
        if get_skin:
            return verts, joints, Rs
        return joints


class HumanModelRecovery(nn.Module):
    """The part of networks/hmr.py on the Imitator path when target SMPL vectors are given:
    `get_details` (hmr.py:302-330).  The image -> theta ResNet-50 regressor is out of scope."""

    def __init__(self, smpl_pkl_path=None, smpl_params=None):
        super().__init__()
        self.smpl = SMPL(smpl_pkl_path, params=smpl_params)

    def forward(self, inputs):
        raise NotImplementedError("the HMR image regressor (networks/hmr.py:200-300) is not part of the "
                                  "Imitator.forward() path; pass src_smpl / tgt_smpls")

    # anything else: the target's own camera (models/imitator.py:216-234); 'as_is': theta = the given vector (get_details only)
    STRATEGIES = {'smooth': 1, 'source': 2, 'as_is': 0}

    @torch.no_grad()
    def get_details_swapped(self, tgt_smpl, src_cam, src_shape, first_cam, cam_strategy='smooth'):
        """`get_details(Imitator.swap_smpl(src_cam, src_shape, tgt_smpl, cam_strategy))` for CUDA tensors as liblwg launches
        only: lwg_smpl_swap (camera policy, theta and its contiguous parts), the SMPL kernels, lwg_smpl_project_joints.  Same
        values bit for bit as the tensor expressions (tests/test_gpu_imitator.py)."""
        from .. import _lib
        lib = _lib.load()
        tgt = tgt_smpl.float().contiguous()
        n, nb = tgt.shape[0], tgt.shape[1] - 75
        dev = tgt.device
        # the kernel reads src_cam[0..2], first_cam[0..2] and src_shape[0..nb-1] through raw pointers: ONE source row each, on
        # the targets' device (Imitator.swap_smpl would broadcast or raise; here a wrong shape would read out of bounds)
        for name, t, width in (("src_cam", src_cam, 3), ("src_shape", src_shape, nb), ("first_cam", first_cam, 3)):
            if t is None:
                continue
            if tuple(t.shape) != (1, width) or t.device != dev:
                raise ValueError("get_details_swapped: %s must be a (1, %d) tensor on %s, got %s on %s"
                                 % (name, width, dev, tuple(t.shape), t.device))
        theta = torch.empty((n, 75 + nb), device=dev, dtype=torch.float32)
        cam = torch.empty((n, 3), device=dev, dtype=torch.float32)
        pose = torch.empty((n, 72), device=dev, dtype=torch.float32)
        shape = torch.empty((n, nb), device=dev, dtype=torch.float32)
        strategy = self.STRATEGIES.get(cam_strategy, 3)
        sc = src_cam.float().contiguous() if strategy in (1, 2) else None
        fc = first_cam.float().contiguous() if strategy == 1 else None
        ss = src_shape.float().contiguous() if strategy != 0 else None
        _lib.check(lib.lwg_smpl_swap(_lib.ptr(tgt), n, nb, strategy, _lib.ptr(sc), _lib.ptr(ss), _lib.ptr(fc),
                                     _lib.ptr(theta), _lib.ptr(cam), _lib.ptr(pose), _lib.ptr(shape), _lib.stream_ptr()))
        verts, j3d, _ = self.smpl.forward_theta(theta)
        j2d = torch.empty((n, j3d.shape[1], 2), device=dev, dtype=torch.float32)
        _lib.check(lib.lwg_smpl_project_joints(_lib.ptr(j3d), _lib.ptr(cam), n, j3d.shape[1], _lib.ptr(j2d), _lib.stream_ptr()))
        return {'theta': theta, 'cam': cam, 'pose': pose, 'shape': shape, 'verts': verts, 'j2d': j2d, 'j3d': j3d}

    def get_details(self, theta):
        if theta.is_cuda:
            # the slicing and the keypoint projection as liblwg launches too (same values as the tensor expressions below)
            return self.get_details_swapped(theta, None, None, None, 'as_is')
        # CPU tensors: the tensor-op statement of the same function (SMPL.forward_ops) -- what the CPU tests of the host logic and
        # tests/test_oracle_vs_reference.py pin to the reference; the product path (CUDA tensors) returned above
        cam = theta[:, 0:3].contiguous()
        pose = theta[:, 3:75].contiguous()
        shape = theta[:, 75:].contiguous()
        verts, j3d, _ = self.smpl(beta=shape, theta=pose, get_skin=True)
        return {'theta': theta, 'cam': cam, 'pose': pose, 'shape': shape, 'verts': verts,
                'j2d': batch_orth_proj_idrot(j3d, cam), 'j3d': j3d}
