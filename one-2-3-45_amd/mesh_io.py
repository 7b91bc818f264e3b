"""Mesh serialisation: binary little-endian PLY in the layout trimesh's exporter produces for
``trimesh.Trimesh(vertices, faces[, vertex_colors]).export('x.ply')`` (the reference's models/trainer_generic.py:1302-1303,
1377-1382): vertex = 3 x float32 [+ 4 x uint8 rgba], face = uint8 count + 3 x int32.

``export_mesh`` is the device path: marching-cubes index coordinates, the frame transforms and the uint8 colours are turned into
the two record arrays by csrc/mesh_pack.hip, copied to the host once and written with a single ``write``.
``write_ply`` / ``read_ply`` are the host-side (numpy) file layer, also used by the ``trimesh`` shim.
trimesh itself is not vendored with the reference (requirements.txt); the header text follows its published PLY template."""
import numpy as np

_VERTEX = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")]
_RGBA = [("red", "u1"), ("green", "u1"), ("blue", "u1"), ("alpha", "u1")]
_FACE = np.dtype([("n", "u1"), ("idx", "<i4", (3,))])


def ply_header(n_vertices, n_faces, colors):
    lines = ["ply", "format binary_little_endian 1.0", "comment https://github.com/mikedh/trimesh", f"element vertex {n_vertices}",
             "property float x", "property float y", "property float z"]
    if colors:
        lines += ["property uchar red", "property uchar green", "property uchar blue", "property uchar alpha"]
    lines += [f"element face {n_faces}", "property list uchar int vertex_indices", "end_header"]
    return ("\n".join(lines) + "\n").encode("ascii")


_WARNED = False


def write_records(path, vertex_records, face_records, colors):
    """vertex_records / face_records: contiguous uint8 arrays (16 or 12 bytes per vertex, 13 per face)."""
    vertex_records = np.ascontiguousarray(vertex_records, np.uint8).reshape(-1)
    face_records = np.ascontiguousarray(face_records, np.uint8).reshape(-1)
    vs = 16 if colors else 12
    assert vertex_records.size % vs == 0 and face_records.size % 13 == 0
    with open(path, "wb") as f:
        f.write(ply_header(vertex_records.size // vs, face_records.size // 13, colors))
        f.write(vertex_records.data)            # straight from the arrays' buffers: no intermediate bytes objects
        f.write(face_records.data)


def write_ply(path, vertices, faces, vertex_colors=None):
    """Host arrays in (vertices [N,3] float, faces [M,3] int, vertex_colors [N,3|4] uint8 or None).  The two record arrays are packed by the library's
    host-side packer (o2345_ply_records_host: float64 -> float32, int64 -> int32, rgba; a few threads, no device work) -- numpy's 12 -> 13-byte row
    copies took 6 - 8 ms for the 1 M records of a 256^3 mesh, inside the reference's "export mesh time" bracket."""
    import ctypes
    from . import _lib
    v = np.ascontiguousarray(vertices, np.float64).reshape(-1, 3)
    f = np.ascontiguousarray(faces, np.int64).reshape(-1, 3)
    c = None if vertex_colors is None else np.ascontiguousarray(vertex_colors, np.uint8)
    if c is not None and (c.ndim != 2 or c.shape[0] != v.shape[0] or c.shape[1] not in (3, 4)):
        raise ValueError(f"write_ply: vertex_colors must be [N,3] or [N,4] uint8 for N = {v.shape[0]} vertices, got {c.shape}")
    try:
        L = _lib.lib()
    except (RuntimeError, OSError, AttributeError) as e:
        # FILE FORMATTING on the host, not device work: without a loadable library (a machine that only converts meshes) the numpy writer produces the
        # same bytes (tests/test_mesh_io.py), 6 - 8 ms slower per million records.  Every compute op still refuses to run without the library.
        global _WARNED
        if not _WARNED:
            import warnings
            warnings.warn(f"o2345 mesh_io.write_ply: libo2345_hip.so is not loadable ({e}); writing the PLY with the numpy packer (same bytes)")
            _WARNED = True
        return write_ply_numpy(path, v, f, c)
    vrec = np.empty((v.shape[0], 16 if c is not None else 12), np.uint8)
    frec = np.empty((f.shape[0], 13), np.uint8)
    P = lambda a: None if a is None or a.size == 0 else a.ctypes.data_as(ctypes.c_void_p)
    _lib.check(L.o2345_ply_records_host(P(v), v.shape[0], P(c), 0 if c is None else c.shape[1], P(f), f.shape[0], P(vrec), P(frec)), "ply_records_host")
    write_records(path, vrec, frec, c is not None)


def write_ply_numpy(path, vertices, faces, vertex_colors=None):
    """The same file from numpy structured arrays (the definition write_ply is tested against)."""
    vertices = np.asarray(vertices)
    faces = np.asarray(faces)
    dt = np.dtype(_VERTEX + (_RGBA if vertex_colors is not None else []))
    v = np.zeros(vertices.shape[0], dt)
    v["x"], v["y"], v["z"] = vertices[:, 0], vertices[:, 1], vertices[:, 2]
    if vertex_colors is not None:
        c = np.asarray(vertex_colors)
        v["red"], v["green"], v["blue"] = c[:, 0], c[:, 1], c[:, 2]
        v["alpha"] = c[:, 3] if c.shape[1] > 3 else 255
    fr = np.zeros(faces.shape[0], _FACE)
    fr["n"] = 3
    fr["idx"] = faces
    write_records(path, v.view(np.uint8), fr.view(np.uint8), vertex_colors is not None)


def read_ply(path):
    """Parser for the files written above -> (vertices float32 [N,3], faces int32 [M,3], colors uint8 [N,4] or None)."""
    raw = open(path, "rb").read()
    end = raw.index(b"end_header\n") + len(b"end_header\n")
    head = raw[:end].decode("ascii").split("\n")
    assert head[0] == "ply" and head[1] == "format binary_little_endian 1.0"
    nv = int([l for l in head if l.startswith("element vertex")][0].split()[-1])
    nf = int([l for l in head if l.startswith("element face")][0].split()[-1])
    colors = any(l == "property uchar red" for l in head)
    dt = np.dtype(_VERTEX + (_RGBA if colors else []))
    v = np.frombuffer(raw, dt, nv, end)
    f = np.frombuffer(raw, _FACE, nf, end + nv * dt.itemsize)
    assert end + nv * dt.itemsize + nf * 13 == len(raw) and (nf == 0 or (f["n"] == 3).all())
    verts = np.stack([v["x"], v["y"], v["z"]], 1)
    cols = np.stack([v["red"], v["green"], v["blue"], v["alpha"]], 1) if colors else None
    return verts, f["idx"].copy(), cols


def export_mesh(path, verts_idx, tris, grid_R, bound_min=(-1.0, -1.0, -1.0), bound_max=(1.0, 1.0, 1.0), scale_mat=None, trans_mat=None,
                vertex_colors=None):
    """Device path: verts_idx fp64 [N,3] index coordinates and tris [M,3] from ops.marching_cubes, optional fp32 colours in [0,1]
    (ops.color_points) -> PLY file.  scale_mat / trans_mat: 4x4 (numpy or tensor) as in the reference's sample dict."""
    from . import ops
    vrec, frec = ops.mesh_pack(verts_idx, tris, grid_R, bound_min, bound_max, scale_mat, trans_mat, vertex_colors)
    write_records(path, vrec.cpu().numpy(), frec.cpu().numpy(), vertex_colors is not None)
    return int(verts_idx.shape[0]), int(tris.shape[0])
