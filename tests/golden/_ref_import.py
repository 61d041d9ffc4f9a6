"""Generator-side helper (BUILD CONTAINER ONLY; nothing under tests/ that runs in the suites imports this): makes the reference's Python hot-path
glue -- `gaussian_renderer`, `scene.gaussian_model.GaussianModel`, `scene.cameras.Camera`, `scene.Scene` -- importable and runnable from
/root/reference on a machine with no GPU, so that the fixture generators can RUN the reference's own code and freeze what it computes.

Two things stand in the way, and this module removes exactly those:

 1. Third-party modules the image lacks: plyfile, simple_knn, cv2, diff_plane_rasterization (the CUDA extension itself) and pytorch3d.  They are
    registered as empty modules so the `import` lines succeed; none of them is on a call path the generators exercise, with one exception --
    `pytorch3d.transforms.quaternion_to_matrix` (scene/gaussian_model.py:23, 162-163: the smallest-axis normals).  pytorch3d is an UNPINNED git
    dependency of the reference (requirements.txt:7, "git+https://github.com/facebookresearch/pytorch3d.git") and is not vendored; its published
    algorithm (pytorch3d/transforms/rotation_conversions.py, unchanged since v0.3) is restated below and checked against the reference's own
    `utils.general_utils.build_rotation` on unit quaternions, where the two must agree.
 2. `device="cuda"` literals and `.cuda()` calls: allocations are pointed at the CPU for the duration of the generator (as make_golden.py already
    does for `build_scaling_rotation`).  The arithmetic that then runs is the reference's own, on torch-CPU fp32.

`diff_plane_rasterization` is replaced by a RECORDER (the rasterizer is the one thing that cannot run here): it keeps the settings and keyword
arguments every `render()` / `render_depth()` call hands to the rasterizer and returns seeded tensors of the right shapes, so the fixtures pin
everything the glue computes AROUND the op -- its inputs, and what it builds from its outputs."""
import sys
import types

import torch

REF = "/root/reference"


class _Absent(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return None


def quaternion_to_matrix(quaternions):
    """pytorch3d.transforms.quaternion_to_matrix as published (real part first; scales by 2 / |q|^2, so any non-zero quaternion)."""
    r, i, j, k = torch.unbind(quaternions, -1)
    two_s = 2.0 / (quaternions * quaternions).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(quaternions.shape[:-1] + (3, 3))


RECORDED = []          # (settings dict, forward kwargs dict, the 9 tensors handed back) per rasterizer call, in call order


class RecordingSettings:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class RecordingRasterizer:
    seed = 0

    def __init__(self, raster_settings):
        self.raster_settings = raster_settings

    def __call__(self, **kw):
        s = self.raster_settings
        H, W, P = int(s.image_height), int(s.image_width), kw["means3D"].shape[0]
        g = torch.Generator().manual_seed(1000 + RecordingRasterizer.seed)
        RecordingRasterizer.seed += 1
        r = lambda *shape: torch.rand(*shape, generator=g)
        outs = (r(3, H, W), torch.randint(0, 9, (P,), generator=g, dtype=torch.int32), r(3, H, W) - 0.5, 1.0 + 3.0 * r(1, H, W), r(20, H, W), r(15, H, W),
                r(1, H, W), r(3, H, W) - 0.5, torch.randint(0, 2, (1, H, W), generator=g, dtype=torch.int32))
        RECORDED.append((dict(s.__dict__), dict(kw), outs))
        return outs


_PATCHED = {}


def install():
    """Stubs + device redirection; returns nothing.  Call before importing anything of the reference."""
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for name in ("plyfile", "simple_knn", "simple_knn._C", "pytorch3d", "pytorch3d.transforms", "cv2", "diff_plane_rasterization"):
        m = _Absent(name); m.__path__ = []
        sys.modules[name] = m
    sys.modules["pytorch3d.transforms"].quaternion_to_matrix = quaternion_to_matrix
    sys.modules["diff_plane_rasterization"].GaussianRasterizationSettings = RecordingSettings
    sys.modules["diff_plane_rasterization"].GaussianRasterizer = RecordingRasterizer

    def on_cpu(fn):
        def wrapped(*a, **k):
            if "device" in k and str(k["device"]).startswith("cuda"):
                k["device"] = "cpu"
            return fn(*a, **k)
        return wrapped
    for name in ("zeros", "zeros_like", "ones", "ones_like", "tensor", "eye", "empty", "full", "arange", "rand", "randn"):
        _PATCHED[name] = getattr(torch, name)
        setattr(torch, name, on_cpu(_PATCHED[name]))
    _PATCHED["Tensor.cuda"] = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    _PATCHED["cuda.empty_cache"] = torch.cuda.empty_cache
    torch.cuda.empty_cache = lambda: None
    # the stand-in for pytorch3d against the reference's own quaternion -> matrix (utils/general_utils.py:81-102) on unit quaternions
    import utils.general_utils as gu
    q = torch.nn.functional.normalize(torch.randn(64, 4, generator=torch.Generator().manual_seed(7)))
    assert torch.allclose(quaternion_to_matrix(q), gu.build_rotation(q), atol=1e-6)


def save_deduped(path, arrays):
    """np.savez_compressed with every distinct array stored once: `_alias` (a JSON string) maps the other names onto it (tests/golden_glue.py:Fixture resolves them)."""
    import hashlib
    import json
    import numpy as np
    seen, alias, uniq = {}, {}, {}
    for k, v in arrays.items():
        v = np.asarray(v)
        h = (hashlib.sha1(np.ascontiguousarray(v).tobytes()).hexdigest(), str(v.dtype), v.shape)
        if v.size > 8 and h in seen:
            alias[k] = seen[h]
        else:
            seen.setdefault(h, k); uniq[k] = v
    uniq["_alias"] = np.asarray(json.dumps(alias))
    np.savez_compressed(path, **uniq)
