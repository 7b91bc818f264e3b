#!/usr/bin/env python
"""Where the drop-in val image spends its time in the whole-image mode (diagnostic; GPU box): first chunk = fused segmented render of all 128 chunks,
later chunks = slices.  Host-side pieces timed separately."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import dropin_bench as DB  # noqa: E402


def main():
    import torch
    dev = torch.device("cuda:0")
    torch.set_num_threads(8)
    tr = DB.TrainerLike(dev)
    sample = DB.make_sample(dev)
    tr.val_step(sample); tr.val_step(sample)
    imgs, fmaps, vol, mask = tr._volume(sample, DB.Stages(False))
    near, far = sample["query_near_far"][0, :1], sample["query_near_far"][0, 1:]
    ro = sample["rays"]["rays_o"][0].reshape(-1, 3).split(512)
    rd = sample["rays"]["rays_v"][0].reshape(-1, 3).split(512)
    kw = dict(background_rgb=None, alpha_inter_ratio=1.0, lod=0, conditional_volume=vol, conditional_valid_mask_volume=mask, feature_maps=fmaps, color_maps=imgs,
              w2cs=sample["w2cs"][0], intrinsics=sample["intrinsics"][0], img_wh=[256, 256], query_c2w=sample["query_c2w"], if_render_with_grad=False)
    ren = tr.sdf_renderer_lod0
    out = {}
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        o0 = ren.render(ro[0], rd[0], near, far, tr.sdf_network_lod0, tr.rendering_network_lod0, **kw)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        for a, b in zip(ro[1:], rd[1:]):
            ren.render(a, b, near, far, tr.sdf_network_lod0, tr.rendering_network_lod0, **kw)
        t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
        out[f"rep{rep}"] = {"first_call_host_ms": (t1 - t0) * 1e3, "first_call_device_tail_ms": (t2 - t1) * 1e3, "127_serves_host_ms": (t3 - t2) * 1e3,
                            "serves_device_tail_ms": (t4 - t3) * 1e3}
    # host RNG cost of one image
    t0 = time.perf_counter()
    st = []
    for k in range(128):
        torch.rand(512, 64); torch.rand([1024, 3]); st.append(torch.get_rng_state())
    out["rng_draws_128_chunks_ms"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    buf = torch.empty(65536, 64, pin_memory=True)
    out["pinned_alloc_ms"] = (time.perf_counter() - t0) * 1e3
    # the trainer's own per-chunk host work on a chunk's outputs
    S = 128
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(128):
        o0["depth"].detach().cpu().numpy(); o0["color_fine"].detach().cpu().numpy()
        (o0["gradients"] * o0["weights"][:, :S, None] * o0["inside_sphere"][..., None]).sum(dim=1).detach().cpu().numpy()
    out["trainer_own_per_chunk_work_128x_ms"] = (time.perf_counter() - t0) * 1e3
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
