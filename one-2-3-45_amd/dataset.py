"""Dataset / pose front end of the reconstruction stage: a Zero123 output folder -> the sample dict the
trainer consumes.  Mirror of ``reconstruction/data/One2345_eval_new_data.py:58-377`` (``BlenderPerView``, val / export_mesh split)
and ``data/scene.py:15-101`` (``get_boundingbox``), host side, numpy + PIL only:

    <root>/<name>/pose.json            {"c2ws": {img_id: 4x4 blender pose, ...40 entries}, "intrinsics": 3x3, "near_far": [n, f]}
    <root>/<name>/stage1_8/<img_id>    the 8 first-stage views (entry 0 is the target / query view)
    <root>/<name>/stage2_8/<img_id>    the 32 second-stage views (entries 8..39 are the source views of the cost volume)

Same keys, shapes, dtypes and arithmetic order as the reference (poses relative to the target view, blender -> opencv flip, scale_mat
from the union of the view frusta x 1.1, normalised poses recovered from P = K [R|t] scale_mat, query_near_far = (0.95 (d - 1),
1.05 (d + 1)), RGBA blended onto white).  The reference recovers the normalised pose with ``cv2.decomposeProjectionMatrix``; for a
similarity ``scale_mat`` that decomposition has the closed form R' = R, C' = (C - centre) / r used here (no OpenCV dependency).
Pinned against the reference class itself by tests/test_dataset_vs_reference.py."""
import json
import os

import numpy as np
import torch

from . import synth

BLENDER2OPENCV = np.array([[1, 0, 0, 0], [0, -1, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]])


def read_image(path):
    """PNG -> float32 [3,H,W] in [0,1]; an alpha channel is blended onto white (One2345_eval_new_data.py:206-209)."""
    from PIL import Image
    a = np.asarray(Image.open(path))
    if a.ndim == 2:
        a = a[..., None]
    t = torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1).to(torch.float32).div(255)      # torchvision ToTensor
    if t.shape[0] == 4:
        t = t[:3] * t[-1:] + (1 - t[-1:])
    return t


def bounding_sphere(img_hw, K4, w2cs, near_fars, factor=1.1):
    """cal_scale_mat + get_boundingbox (data/scene.py:49-101): centre / radius of the box around all view frusta -> (scale_mat float32, 1/radius)."""
    lo, hi = np.full(3, np.inf, np.float32), np.full(3, -np.inf, np.float32)
    for w2c, nf in zip(w2cs, near_fars):
        c2w = torch.inverse(torch.tensor(w2c)).numpy()                          # float64 inverse like the reference, cast inside
        p = synth._frustum_bounds(K4, c2w, nf[0], nf[1], img_hw)
        lo, hi = np.minimum(lo, p.min(1)), np.maximum(hi, p.max(1))
    center = ((hi + lo) / 2).astype(np.float32)
    radius = np.float32((hi - lo).max() / 2) * np.float32(factor)
    scale_mat = np.diag([radius, radius, radius, 1.0])
    scale_mat[:3, 3] = center
    return scale_mat.astype(np.float32), np.float32(1.0) / radius


def normalised_pose(K4, w2c, scale_mat):
    """load_K_Rt_from_P(None, (K @ w2c @ scale_mat)[:3, :4])[1] in closed form -> c2w float32 (the reference's pose is float32)."""
    R = w2c[:3, :3]
    C = -R.T @ w2c[:3, 3]
    r, c = np.float64(scale_mat[0, 0]), scale_mat[:3, 3].astype(np.float64)
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = R.T
    pose[:3, 3] = (C - c) / r
    return pose


class SceneFolder:
    """``BlenderPerView(root_dir, split, ..., specific_dataset_name=name)`` for split 'val' / 'export_mesh': one scene per folder.
    Difference from the reference, split 'export_mesh' only: the reference draws N_rays RANDOM rays for every split whose name contains neither
    'val' nor 'test' (One2345_eval_new_data.py:354-372), which export_mesh_step never reads; here every split carries the full val ray grid
    (deterministic; a caller that wants the reference's random subset indexes it)."""

    def __init__(self, root_dir, split="val", img_wh=(256, 256), vol_dims=(128, 128, 128), specific_dataset_name="", clean_image=False, **_unused):
        if split not in ("val", "export_mesh", "test"):
            raise NotImplementedError("o2345 SceneFolder: the inference splits only ('val' / 'export_mesh')")
        self.folder = os.path.join(root_dir, specific_dataset_name)
        self.split, self.name, self.clean_image = split, specific_dataset_name, clean_image
        self.voxel_dims = torch.tensor(vol_dims, dtype=torch.float32)
        self.partial_vol_origin = torch.tensor([-1.0, -1.0, -1.0], dtype=torch.float32)

    def __len__(self):
        return 1

    def __getitem__(self, idx):
        folder = os.path.join(self.folder, "")
        shape_name = os.path.split(folder)[-1]
        meta = json.load(open(os.path.join(folder, "pose.json")))
        img_ids = list(meta["c2ws"].keys())
        poses = np.array(list(meta["c2ws"].values()))
        K4 = np.eye(4)
        K4[:3, :3] = np.array(meta["intrinsics"])
        near_far = np.array(meta["near_far"])
        img_wh = (256, 256)
        c2ws_all = np.stack([p @ BLENDER2OPENCV for p in poses])
        w2cs_all = np.stack([np.linalg.inv(c) for c in c2ws_all])
        w2c_ref_inv = np.linalg.inv(np.linalg.inv(c2ws_all[0]))
        src_views = list(range(8, 8 + 8 * 4))
        last = len(img_ids) - 1            # the reference's loop variable shadows `idx` (:169): view_ids[0] / the meta string carry the LAST index
        imgs = [read_image(os.path.join(folder, "stage1_8", img_ids[0]))] + [read_image(os.path.join(folder, "stage2_8", img_ids[v])) for v in src_views]
        w2cs = [np.linalg.inv(c2ws_all[0]) @ w2c_ref_inv] + [w2cs_all[v] @ w2c_ref_inv for v in src_views]
        near_fars = [near_far] * len(w2cs)
        scale_mat, scale_factor = bounding_sphere([img_wh[1], img_wh[0]], K4, w2cs, near_fars, 1.1)
        new_c2ws = np.stack([normalised_pose(K4, w, scale_mat) for w in w2cs])
        new_w2cs = np.stack([np.linalg.inv(c) for c in new_c2ws])
        aff = np.tile(np.eye(4)[None], (len(w2cs), 1, 1))
        aff[:, :3, :4] = K4[:3, :3] @ new_w2cs[:, :3, :4]
        dist = np.sqrt(np.sum(new_c2ws[:, :3, 3] ** 2, axis=-1))
        new_nf = np.stack([0.95 * (dist - 1), 1.05 * (dist + 1)], -1)
        target_w2cs = np.stack([np.linalg.inv(normalised_pose(K4, w2cs_all[i] @ w2c_ref_inv, scale_mat)) for i in range(8)])
        H, W = imgs[0].shape[1:]
        depths = np.stack([np.full((H, W), -1.0, np.float32)] + [np.ones((H, W), np.float32)] * len(src_views)) * scale_factor
        f32 = lambda a: torch.from_numpy(np.asarray(a).astype(np.float32))
        images = torch.stack(imgs).float()
        s = {"origin_idx": idx, "images": images, "depths_h": f32(depths), "masks_h": torch.ones(len(w2cs), H, W), "w2cs": f32(new_w2cs), "c2ws": f32(new_c2ws),
             "target_candidate_w2cs": f32(target_w2cs), "near_fars": f32(new_nf), "intrinsics": f32(np.stack([K4] * len(w2cs)))[:, :3, :3],
             "view_ids": torch.from_numpy(np.array([last] + src_views)), "affine_mats": f32(aff), "scan": shape_name,
             "scale_factor": torch.tensor(scale_factor), "img_wh": torch.from_numpy(np.array(img_wh)), "render_img_idx": torch.tensor(0),
             "partial_vol_origin": self.partial_vol_origin, "meta": str(self.name) + "_" + str(shape_name) + "_refview" + str(last)}
        s["query_image"], s["query_c2w"], s["query_w2c"] = s["images"][0], s["c2ws"][0], s["w2cs"][0]
        s["query_intrinsic"], s["query_depth"], s["query_mask"], s["query_near_far"] = s["intrinsics"][0], s["depths_h"][0], s["masks_h"][0], s["near_fars"][0]
        for k in ("images", "depths_h", "masks_h", "w2cs", "c2ws", "intrinsics", "view_ids", "affine_mats"):
            s[k] = s[k][1:]                                                      # the source views (start_idx = 1 outside training)
        s["scale_mat"], s["trans_mat"] = torch.from_numpy(scale_mat), torch.from_numpy(w2c_ref_inv)
        ro, rd = synth.gen_rays(s["query_intrinsic"].numpy(), s["query_c2w"].numpy(), img_wh[1], img_wh[0])
        ys, xs = np.meshgrid(np.linspace(0, H - 1, H), np.linspace(0, W - 1, W), indexing="ij")
        uv = np.stack([2 * xs / (W - 1) - 1, 2 * ys / (H - 1) - 1], -1).reshape(-1, 2).astype(np.float32)
        pix = np.stack([xs, ys, np.ones_like(xs)], -1).reshape(-1, 3).astype(np.float32)
        xyz_cam = (np.linalg.inv(s["query_intrinsic"].numpy().astype(np.float64)).astype(np.float32) @ pix.T).T      # rays.py:33-35, before normalisation
        s["rays"] = {"rays_o": torch.from_numpy(ro), "rays_v": torch.from_numpy(rd), "rays_ndc_uv": torch.from_numpy(uv),
                     "rays_norm_XYZ_cam": torch.from_numpy(np.ascontiguousarray(xyz_cam)),
                     "rays_color": s["query_image"].permute(1, 2, 0).reshape(-1, 3), "rays_mask": s["query_mask"].reshape(-1, 1) if self.clean_image else torch.ones(H * W, 1),
                     "rays_depth": s["query_depth"].reshape(-1, 1)}
        return s


def write_synthetic_folder(root, name="", seed=0, hw=(256, 256), polar=60.0):
    """A Zero123-style output folder (pose.json + 8 + 32 RGBA PNGs) with the reference's camera rig (utils/utils.py:80-135) and
    seeded random images -- what ``run.py`` stage 1/2 would have written; for tests and offline runs."""
    from PIL import Image
    folder = os.path.join(root, name)
    os.makedirs(os.path.join(folder, "stage1_8"), exist_ok=True)
    os.makedirs(os.path.join(folder, "stage2_8"), exist_ok=True)
    poses = synth.rig_c2ws(polar)
    h, w = hw
    f = 280.0 * w / 256.0
    ids = [f"view_{i}.png" for i in range(8)] + [f"view_{i}_{j}_10.png" for i in range(8) for j in range(4)]
    rng = np.random.default_rng(seed)
    for k, img_id in enumerate(ids):
        a = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        a[..., 3] = np.where(rng.random((h, w)) < 0.7, 255, rng.integers(0, 256, (h, w)))
        Image.fromarray(a, "RGBA").save(os.path.join(folder, "stage1_8" if k < 8 else "stage2_8", img_id))
    json.dump({"c2ws": {i: p.tolist() for i, p in zip(ids, poses)}, "intrinsics": [[f, 0, w / 2], [0, f, h / 2], [0, 0, 1]], "near_far": [1.2 - 0.7, 1.2 + 0.6]},
              open(os.path.join(folder, "pose.json"), "w"))
    return folder
