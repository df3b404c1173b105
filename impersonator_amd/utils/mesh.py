"""Face -> condition tables from the reference's asset files (utils/mesh.py:156-421 of the reference).

`create_mapping` returns the (nf + 1, C) table that SMPLRenderer.encode_fim indexes with the face-index map
(the extra last row is the background, reached by fim == -1).  The tables are constants built once on the host;
only their use (the gather) is on the hot path.  Asset files (`mapper.txt` = an .obj with texture coordinates,
`smpl_part_info.json`, `front_facial.json`, `head.json`) are downloads of the reference (README.md:48-68).
"""
import json
import os

import numpy as np


def _load_uv_obj(path):
    """Texture coordinates and per-face texture indices of an .obj file (utils/mesh.py:28-77)."""
    vts, faces_vts = [], []
    with open(path, 'r') as fp:
        for line in fp:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == 'vt':
                vts.append([float(tok[1]), float(tok[2])])
            elif tok[0] == 'f':
                faces_vts.append([int(p.split('/')[1]) - 1 for p in tok[1:4]])
    return np.asarray(vts, np.float32), np.asarray(faces_vts, np.int32)


def get_f2vts(uv_mapping_path, fill_back=False):
    """utils/mesh.py:173-196: (F, 3, 3) per-face UV vertices (v flipped, z = 0)."""
    vts, faces = _load_uv_obj(uv_mapping_path)
    vts[:, 1] = 1 - vts[:, 1]
    vts = np.concatenate([vts, np.zeros((vts.shape[0], 1), np.float32)], axis=-1)
    if fill_back:
        faces = np.concatenate((faces, faces[:, ::-1]), axis=0)
    return vts[faces]


def compute_barycenter(f2vts):
    """utils/mesh.py:156-170: centroid as v2 + (v0 - v2)/2 + (v1 - v2)/2 (sic: not the mean of the corners)."""
    v2 = f2vts[:, 2]
    return v2 + 0.5 * (f2vts[:, 0] - v2) + 0.5 * (f2vts[:, 1] - v2)


def _face_list(path, nf, fill_back):
    with open(path, 'r') as reader:
        faces = list(json.load(reader)['face'])
    if fill_back:
        faces = faces + [f + nf // 2 for f in faces]
    return faces


def binary_mapping(nf):
    """utils/mesh.py:271-279: the face index in binary, most significant bit first, len(binary_repr(nf)) channels; background = -1
    everywhere.  Returns (table (nf, width), background row (1, width))."""
    width = len(np.binary_repr(nf))
    idx = np.arange(nf, dtype=np.int64)[:, None]
    map_fn = ((idx >> np.arange(width - 1, -1, -1, dtype=np.int64)[None, :]) & 1).astype(np.float32)
    return map_fn, np.zeros((1, width), np.float32) - 1.0


def create_mapping(map_name, mapping_path='assets/pretrains/mapper.txt',
                   part_info='assets/pretrains/smpl_part_info.json',
                   front_info='assets/pretrains/front_facial.json',
                   head_info='assets/pretrains/head.json', contain_bg=True, fill_back=False):
    """utils/mesh.py:368-421 for the map names the Imitator / Swapper paths use."""
    if not os.path.exists(mapping_path):
        raise FileNotFoundError("UV mapper %s not found (a download of the reference); pass map_fn= to SMPLRenderer, "
                                "e.g. impersonator_amd.utils.synthetic.uv_seg_map_fn" % mapping_path)
    f2vts = get_f2vts(mapping_path, fill_back=fill_back)
    nf = f2vts.shape[0]
    if map_name == 'uv':
        map_fn, bg = compute_barycenter(f2vts)[:, 0:2], np.array([[-1, -1]], np.float32)
    elif map_name == 'seg':
        map_fn, bg = np.ones((nf, 1), np.float32), np.array([[0]], np.float32)
    elif map_name == 'uv_seg':
        map_fn, bg = compute_barycenter(f2vts), np.array([[0, 0, 1]], np.float32)
    elif map_name == 'par':
        with open(part_info, 'r') as reader:
            parts = json.load(reader)
        map_fn = np.zeros((nf, len(parts) + 1), np.float32)
        seen = set()
        for i, name in enumerate(sorted(parts.keys())):
            faces = list(parts[name]['face'])
            if fill_back:
                faces = faces + [f + nf // 2 for f in faces]
            map_fn[faces, i] = 1.0
            seen |= set(faces)
        assert len(seen) == nf, 'part table covers %d of %d faces' % (len(seen), nf)
        bg = np.zeros((1, len(parts) + 1), np.float32)
        bg[0, -1] = 1
    elif map_name in ('front', 'head'):
        map_fn = np.zeros((nf, 1), np.float32)
        map_fn[_face_list(front_info if map_name == 'front' else head_info, nf, fill_back)] = 1.0
        bg = np.zeros((1, 1), np.float32)
    elif map_name == 'back':
        head = set(_face_list(head_info, nf, False))
        front = set(_face_list(front_info, nf, False))
        faces = list(head - front)
        if fill_back:
            faces = faces + [f + nf // 2 for f in faces]
        map_fn = np.zeros((nf, 1), np.float32)
        map_fn[faces] = 1.0
        bg = np.zeros((1, 1), np.float32)
    elif map_name == 'binary':
        map_fn, bg = binary_mapping(nf)
    else:
        # ('ids' cannot be built with contain_bg in the reference either: its table is 1-D, utils/mesh.py:282-285)
        raise ValueError('map name error {}'.format(map_name))
    if contain_bg:
        map_fn = np.concatenate([map_fn, bg], axis=0)
    return map_fn.astype(np.float32)


def get_part_face_ids(part_type, mapping_path='assets/pretrains/mapper.txt',
                      part_info='assets/pretrains/smpl_part_info.json', fill_back=False):
    """utils/mesh.py:424-445 for part_type == 'par': {part name: face ids}, names sorted."""
    if part_type != 'par':
        raise ValueError('part type {} not supported'.format(part_type))
    nf = get_f2vts(mapping_path, fill_back=fill_back).shape[0]
    with open(part_info, 'r') as reader:
        parts = json.load(reader)
    out = {}
    for name in sorted(parts.keys()):
        faces = list(parts[name]['face'])
        if fill_back:
            faces = faces + [f + nf // 2 for f in faces]
        out[name] = faces
    return out
