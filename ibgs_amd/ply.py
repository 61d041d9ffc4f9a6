"""The Gaussian checkpoint format either side of the rasterizer: `point_cloud.ply` as written by
`GaussianModel.save_ply` and read by `load_ply` (reference scene/gaussian_model.py:264-360), without the `plyfile`
dependency (absent from this image).  Binary little-endian, one `vertex` element, every property `float`:

    x y z | nx ny nz | nd | f_dc_0..2 | f_rest_0..(3(K-1)-1) | opacity | scale_0..2 | rot_0..3

`nx,ny,nz` hold the raw `_normal` parameter and `nd` the raw `_offset` (the reference overwrites the conventional
zero normals with them, :283-284); SH blocks are stored channel-major -- `features.transpose(1, 2).flatten(1)` -- i.e.
f_rest_j = coefficient (1 + j % (K-1)) of colour channel j // (K-1).  All values are the RAW (pre-activation) parameters."""
import numpy as np


def attribute_names(n_rest):
    """construct_list_of_attributes (gaussian_model.py:264-277) for 3 DC and `n_rest` higher-order SH values."""
    names = ["x", "y", "z", "nx", "ny", "nz", "nd"]
    names += ["f_dc_%d" % i for i in range(3)] + ["f_rest_%d" % i for i in range(n_rest)]
    return names + ["opacity"] + ["scale_%d" % i for i in range(3)] + ["rot_%d" % i for i in range(4)]


def save_ply(path, xyz, normal, offset, features_dc, features_rest, opacity, scaling, rotation):
    """Arrays as the model holds them: xyz (P,3), normal (P,3), offset (P,1), features_dc (P,1,3), features_rest (P,K-1,3),
    opacity (P,1), scaling (P,3), rotation (P,4).  numpy arrays or anything `np.asarray` accepts (detached CPU tensors)."""
    a = lambda t: np.asarray(t, dtype=np.float32)
    xyz = a(xyz); P = xyz.shape[0]
    f_dc = a(features_dc).transpose(0, 2, 1).reshape(P, -1)
    f_rest = a(features_rest).transpose(0, 2, 1).reshape(P, -1)
    table = np.concatenate([xyz, a(normal).reshape(P, 3), a(offset).reshape(P, 1), f_dc, f_rest, a(opacity).reshape(P, 1),
                            a(scaling).reshape(P, 3), a(rotation).reshape(P, 4)], axis=1).astype("<f4")
    names = attribute_names(f_rest.shape[1])
    assert table.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % P
    header += "".join("property float %s\n" % n for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(table).tobytes())


def load_ply(path, max_sh_degree=None):
    """Returns a dict of float32 arrays in the model's layout (see save_ply); checks the SH count like the reference (:326)."""
    with open(path, "rb") as f:
        raw = f.read()
    end = raw.index(b"end_header\n") + len(b"end_header\n")
    lines = raw[:end].decode("ascii").split("\n")
    if lines[0] != "ply" or not lines[1].startswith("format binary_little_endian"):
        raise ValueError("load_ply reads binary little-endian PLY files")
    P, props, in_vertex = 0, [], False
    types = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1", "int": "<i4", "int32": "<i4"}
    for ln in lines[2:]:
        t = ln.split()
        if t[:2] == ["element", "vertex"]:
            P, in_vertex = int(t[2]), True
        elif t[:1] == ["element"]:
            in_vertex = False
        elif t[:1] == ["property"] and in_vertex:
            props.append((t[2], types[t[1]]))
    rec = np.frombuffer(raw, dtype=np.dtype(props), count=P, offset=end)
    col = lambda n: np.asarray(rec[n], np.float32)
    rest = sorted([n for n, _ in props if n.startswith("f_rest_")], key=lambda s: int(s.split("_")[-1]))
    if max_sh_degree is not None and len(rest) != 3 * (max_sh_degree + 1) ** 2 - 3:
        raise ValueError("PLY holds %d f_rest values, expected %d" % (len(rest), 3 * (max_sh_degree + 1) ** 2 - 3))
    K1 = len(rest) // 3
    out = {"xyz": np.stack([col("x"), col("y"), col("z")], 1), "normal": np.stack([col("nx"), col("ny"), col("nz")], 1),
           "offset": col("nd").reshape(-1, 1), "opacity": col("opacity").reshape(-1, 1),
           "features_dc": np.stack([col("f_dc_0"), col("f_dc_1"), col("f_dc_2")], 1).reshape(P, 3, 1).transpose(0, 2, 1).copy(),
           "features_rest": (np.stack([col(n) for n in rest], 1).reshape(P, 3, K1).transpose(0, 2, 1).copy() if K1 else np.zeros((P, 0, 3), np.float32)),
           "scaling": np.stack([col(n) for n in sorted([n for n, _ in props if n.startswith("scale_")], key=lambda s: int(s.split("_")[-1]))], 1),
           "rotation": np.stack([col(n) for n in sorted([n for n, _ in props if n.startswith("rot")], key=lambda s: int(s.split("_")[-1]))], 1)}
    return out
