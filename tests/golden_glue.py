"""Reader of tests/golden/glue.npz (made by tests/golden/make_glue_fixture.py from the reference's own render glue) + the objects that let this
repository's glue run on the same inputs: `SimpleGaussians` / `SimpleCamera` / `SimpleScene` built from the fixture's raw parameters and poses, and
a stand-in rasterizer that records what `ibgs_amd.renderer` hands to the op and returns the tensors the reference's recorder returned."""
import json
import os
from types import SimpleNamespace

import numpy as np
import torch

from ibgs_amd import simple_scene, synthetic as syn

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "glue.npz")
SETTING_FIELDS = ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix", "ref_to_src_list", "src_cam_pos",
                  "src_images", "src_rendered_depths", "nb_src_images", "buffer_length", "depth_error_threshold", "sh_degree", "campos", "prefiltered",
                  "render_geo", "render_depth_only", "debug")
FORWARD_ARGS = ("means3D", "means2D", "means2D_abs", "shs", "colors_precomp", "opacities", "scales", "rotations", "all_map", "cov3D_precomp")


class Fixture:
    """An .npz written by tests/golden/_ref_import.py:save_deduped (every distinct array stored once, `_alias` maps the other names onto it)."""

    def __init__(self, path):
        z = np.load(path)
        self.alias = json.loads(str(z["_alias"]))
        self.z = z

    def __contains__(self, k):
        return k in self.alias or k in self.z.files

    def __getitem__(self, k):
        return self.z[self.alias.get(k, k)]

    def keys(self):
        return [k for k in self.z.files if k != "_alias"] + list(self.alias)


class Glue(Fixture):
    def __init__(self, path=PATH):
        super().__init__(path)
        self.cases = [str(c) for c in self["cases"]]

    def none(self, prefix):
        return set(str(s) for s in self[prefix + "_none"])

    # ---- this repository's objects on the fixture's inputs
    def model(self, device="cpu"):
        g = {"means3D": self["raw_xyz"], "shs": np.concatenate([self["raw_f_dc"], self["raw_f_rest"]], axis=1), "scales": np.exp(self["raw_scaling"]),
             "rotations": self["raw_rotation"], "opacities": 1.0 / (1.0 + np.exp(-self["raw_opacity"])), "normal": self["raw_normal"], "offset": self["raw_offset"]}
        pc = simple_scene.SimpleGaussians(g, sh_degree=int(self["DEG"]), device=device)
        with torch.no_grad():          # the raw values themselves, not exp(log(.)) / logit(sigmoid(.)) round trips
            pc._scaling.copy_(torch.as_tensor(self["raw_scaling"])); pc._opacity.copy_(torch.as_tensor(self["raw_opacity"]))
        return pc

    def cameras(self, device="cpu"):
        W, H = int(self["W"]), int(self["H"])
        cams = []
        for k in range(int(self["NV"])):
            c = simple_scene.SimpleCamera(syn.camera_from_pose(W, H, self["R%d" % k], self["T%d" % k], float(self["fovx"]), float(self["fovy"])), uid=k, device=device)
            c.nearest_id = [int(i) for i in self["cam%d_nearest" % k]]
            cams.append(c)
        return cams

    def scene(self, cams, device="cpu"):
        sc = simple_scene.SimpleScene(cams, images=torch.as_tensor(self["src_images"], device=device), device=device)
        sc.rendered_depth_list = torch.as_tensor(self["rendered_depth_list"], device=device)
        return sc

    def pipe_args(self, name):
        pipe = SimpleNamespace(compute_cov3D_python=bool(self["case_%s_pipe_python" % name]), convert_SHs_python=bool(self["case_%s_pipe_python" % name]), debug=False)
        exp = bool(self["case_%s_args_exposure" % name])
        args = SimpleNamespace(depth_error_threshold=float(self["args_depth_error_threshold"]), shuffle_source_frame=False, multi_view_num=3 if exp else 8,
                               multi_view_max_angle=30, multi_view_min_dis=0.01, multi_view_max_dis=1.5, enable_exposure_correction=exp)
        return pipe, args

    def call_kwargs(self, name):
        pre = "case_%s_kw_" % name
        kw = {}
        for k in self.z.files + list(self.alias):
            if k.startswith(pre) and not k.endswith("_none"):
                v = self[k]
                kw[k[len(pre):]] = v.item() if v.shape == () else v
        for k in self.none(pre):
            kw[k] = None
        return kw


def replaying_rasterizer(glue, name, log, device="cpu"):
    """A class to put in place of `renderer.GaussianRasterizer`: logs (settings, kwargs) and returns what the reference's recorder returned for the same call."""
    class Replay:
        n = 0

        def __init__(self, raster_settings):
            self.raster_settings = raster_settings

        def __call__(self, **kw):
            j = Replay.n
            Replay.n += 1
            log.append((self.raster_settings, kw))
            return tuple(torch.as_tensor(glue["case_%s_call%d_ret%d" % (name, j, i)], device=device) for i in range(9))
    return Replay
