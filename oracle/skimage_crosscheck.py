"""Independent marching-cubes cross-check (TEST INFRASTRUCTURE ONLY): runs scikit-image's classic Lorensen implementation
(skimage.measure.marching_cubes(method='lorensen'), its own Cython code and its own copy of the table) on analytic and
random volumes and writes tests/golden/mc_skimage.npz.  tests/test_mc_tables.py compares oracle/mc.c (and, on the GPU,
csrc/mcubes.hip) with these fixtures: same vertex set, same triangles (as vertex-position triples), same Euler characteristic.

    /opt/conda/bin/python3.9 oracle/skimage_crosscheck.py       # the image's conda env ships scikit-image 0.18.3

Conventions bridged here (PyMCubes as the reference calls it, sparse_neus_renderer.py:932-936, vs scikit-image):
  * PyMCubes marks a corner inside when u <= iso, scikit-image when u > level  -> scikit-image runs on -u at level -iso;
  * PyMCubes' corner 1 is +axis0 and corners 4..7 are +axis2, scikit-image's corner 1 is +axis2 and 4..7 are +axis0
    -> scikit-image runs on the transposed volume and its vertex columns are reversed.
Every volume avoids u == iso exactly on grid points (the two inside tests differ only there)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "mc_skimage.npz")


def volumes():
    n = 24
    g = np.stack(np.meshgrid(*[np.linspace(-1, 1, n)] * 3, indexing="ij"), -1)
    x, y, z = g[..., 0], g[..., 1], g[..., 2]
    v = {}
    v["sphere"] = (np.sqrt(x * x + y * y + z * z) - 0.6137, 0.0)
    v["torus"] = (np.sqrt((np.sqrt(x * x + y * y) - 0.55) ** 2 + z * z) - 0.2213, 0.0)
    d1 = np.sqrt((x + 0.3) ** 2 + (y - 0.1) ** 2 + z * z) - 0.413
    d2 = np.sqrt((x - 0.35) ** 2 + y * y + (z + 0.2) ** 2) - 0.337
    v["two_spheres"] = (np.minimum(d1, d2), 0.0)
    v["ellipsoid_iso"] = (np.sqrt((x / 0.8) ** 2 + (y / 0.5) ** 2 + ((z - 0.1) / 0.3) ** 2), 0.873)    # non-zero iso value
    rng = np.random.default_rng(1234)
    v["noise_anisotropic"] = (rng.standard_normal((9, 7, 11)), 0.05)         # every one of the 256 cases, ambiguous ones included
    v["noise_cube"] = (rng.standard_normal((12, 12, 12)), -0.1)
    return {k: (np.ascontiguousarray(u, np.float32), float(iso)) for k, (u, iso) in v.items()}


def canonical(verts, faces):
    """-> (vertices sorted lexicographically, faces as sorted triples of NEW vertex ids, sorted) -- numbering independent."""
    verts = np.asarray(verts, np.float64)
    faces = np.asarray(faces, np.int64)
    order = np.lexsort((verts[:, 2], verts[:, 1], verts[:, 0]))
    inv = np.empty_like(order)
    inv[order] = np.arange(len(order))
    f = np.sort(inv[faces], axis=1)
    f = f[np.lexsort((f[:, 2], f[:, 1], f[:, 0]))]
    return verts[order], f


def main(out=OUT):
    from skimage import measure
    rec = {}
    for name, (u, iso) in volumes().items():
        vol = np.ascontiguousarray(-u.transpose(2, 1, 0))
        verts, faces, _, _ = measure.marching_cubes(vol, -iso, method="lorensen")
        verts = verts[:, ::-1]
        keep = np.array([len(set(f)) == 3 for f in faces], bool)      # unpack_unique_verts can merge coincident vertices
        cv, cf = canonical(verts, faces[keep])
        rec[name + ":u"], rec[name + ":iso"] = u, np.float64(iso)
        rec[name + ":verts"], rec[name + ":faces"] = cv, cf
        rec[name + ":n_degenerate"] = np.int64((~keep).sum())
        print(name, u.shape, "verts", len(cv), "faces", len(cf), "degenerate", int((~keep).sum()))
    np.savez_compressed(out, **rec)
    print(out)


if __name__ == "__main__":
    main(*sys.argv[1:])
