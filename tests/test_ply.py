"""Checkpoint format (reference scene/gaussian_model.py:264-360): attribute order, channel-major SH layout, raw normal /
offset in nx,ny,nz / nd, binary little-endian floats; round trip and a hand-built file as a plyfile-style writer lays it out."""
import numpy as np

from ibgs_amd import ply


def _model(P=37, K=9, seed=0):
    r = np.random.default_rng(seed)
    f = lambda *s: r.standard_normal(s).astype(np.float32)
    return dict(xyz=f(P, 3), normal=f(P, 3), offset=f(P, 1), features_dc=f(P, 1, 3), features_rest=f(P, K - 1, 3),
                opacity=f(P, 1), scaling=f(P, 3), rotation=f(P, 4))


def test_attribute_order_matches_construct_list_of_attributes():
    n = ply.attribute_names(24)
    assert n[:7] == ["x", "y", "z", "nx", "ny", "nz", "nd"] and n[7:10] == ["f_dc_0", "f_dc_1", "f_dc_2"]
    assert n[10] == "f_rest_0" and n[33] == "f_rest_23" and n[34] == "opacity"
    assert n[35:38] == ["scale_0", "scale_1", "scale_2"] and n[38:] == ["rot_0", "rot_1", "rot_2", "rot_3"] and len(n) == 42


def test_round_trip_and_byte_layout(tmp_path):
    m = _model()
    path = str(tmp_path / "point_cloud.ply")
    ply.save_ply(path, **m)
    raw = open(path, "rb").read()
    head = raw[:raw.index(b"end_header\n") + 11].decode()
    assert head.startswith("ply\nformat binary_little_endian 1.0\nelement vertex 37\nproperty float x\n")
    assert head.count("property float ") == 7 + 3 + 24 + 1 + 3 + 4
    body = np.frombuffer(raw, "<f4", offset=len(head)).reshape(37, 42)
    assert np.array_equal(body[:, 0:3], m["xyz"]) and np.array_equal(body[:, 3:6], m["normal"]) and np.array_equal(body[:, 6:7], m["offset"])
    # channel-major SH: f_rest_j = coefficient 1 + j % 8 of channel j // 8   (features.transpose(1, 2).flatten(1))
    assert np.array_equal(body[:, 10 + 0], m["features_rest"][:, 0, 0]) and np.array_equal(body[:, 10 + 7], m["features_rest"][:, 7, 0])
    assert np.array_equal(body[:, 10 + 8], m["features_rest"][:, 0, 1]) and np.array_equal(body[:, 10 + 23], m["features_rest"][:, 7, 2])
    assert np.array_equal(body[:, 7:10], m["features_dc"][:, 0, :])
    back = ply.load_ply(path, max_sh_degree=2)
    for k, v in m.items():
        assert back[k].shape == v.shape and np.array_equal(back[k], v), k
    try:
        ply.load_ply(path, max_sh_degree=3)
        raise AssertionError("SH count mismatch must be rejected like the reference's assert")
    except ValueError:
        pass


def test_reads_a_file_with_reordered_and_extra_properties(tmp_path):
    """load_ply goes by property NAME like the reference (plydata.elements[0]["x"] ...), not by position."""
    m = _model(P=5, K=4, seed=2)
    names = ply.attribute_names(9)
    order = names[::-1] + ["extra"]
    path = str(tmp_path / "shuffled.ply")
    ply.save_ply(path, **m)
    ref = np.frombuffer(open(path, "rb").read()[-5 * len(names) * 4:], "<f4").reshape(5, len(names))
    cols = {n: ref[:, i] for i, n in enumerate(names)}; cols["extra"] = np.zeros(5, np.float32)
    head = "ply\nformat binary_little_endian 1.0\ncomment hand made\nelement vertex 5\n" + "".join("property float %s\n" % n for n in order) + "end_header\n"
    with open(path, "wb") as f:
        f.write(head.encode()); f.write(np.stack([cols[n] for n in order], 1).astype("<f4").tobytes())
    back = ply.load_ply(path, max_sh_degree=1)
    for k, v in m.items():
        assert np.array_equal(back[k], v), k
