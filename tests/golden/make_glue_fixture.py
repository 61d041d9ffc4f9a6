"""Generates tests/golden/glue.npz IN THE BUILD CONTAINER by RUNNING the reference's own Python glue around the rasterizer:

    scene/gaussian_model.py:156-173     GaussianModel.get_normal / get_normal_w_smallest_axis (plane normals facing the camera, offset sign)
    gaussian_renderer/__init__.py:143-365  render(): sinks, tan(FoV/2), SH / covariance switches, source selection, ref_to_src, src_cam_pos,
                                           the (P, 5) plane map `all_map` (:304-316), depth -> normal, appearance affine, returned dictionary
    gaussian_renderer/__init__.py:41-140   render_depth()
    scene/cameras.py:51-134, scene/__init__.py:113-141   Camera matrices, Scene's per-view tables

The CUDA rasterizer in the middle is replaced by a recorder (tests/golden/_ref_import.py): the fixture holds, per call, every setting and tensor the
reference hands TO the op, the (seeded) tensors the recorder hands back, and the dictionary the reference builds from them.  Data only: inputs and
the reference's outputs.  Re-run:  python tests/golden/make_glue_fixture.py"""
import math
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import as ri  # noqa: E402

ri.install()
import gaussian_renderer as gr  # noqa: E402  (the reference's)
from scene import Scene  # noqa: E402
from scene.cameras import Camera  # noqa: E402
from scene.gaussian_model import GaussianModel  # noqa: E402
from utils.graphics_utils import focal2fov, fov2focal  # noqa: E402

rng = np.random.default_rng(20261003)
P, W, H, NV, DEG = 96, 20, 12, 5, 2
f32 = lambda a: torch.tensor(np.asarray(a, dtype=np.float32))

# ---- the model: raw (pre-activation) parameters, as a trained GaussianModel holds them --------------------------------------------------
raw = {"xyz": rng.uniform(-1.3, 1.3, (P, 3)), "f_dc": rng.normal(0, 1.0, (P, 1, 3)), "f_rest": rng.normal(0, 0.1, (P, (DEG + 1) ** 2 - 1, 3)),
       "scaling": rng.normal(-2.5, 0.8, (P, 3)), "rotation": rng.normal(size=(P, 4)), "opacity": rng.normal(0, 2, (P, 1)),
       "normal": rng.normal(size=(P, 3)), "offset": 0.05 * rng.normal(size=(P, 1))}
raw = {k: v.astype(np.float32) for k, v in raw.items()}
pc = GaussianModel(DEG)
pc.active_sh_degree = DEG
for k, attr in (("xyz", "_xyz"), ("f_dc", "_features_dc"), ("f_rest", "_features_rest"), ("scaling", "_scaling"), ("rotation", "_rotation"),
                ("opacity", "_opacity"), ("normal", "_normal"), ("offset", "_offset")):
    setattr(pc, attr, torch.nn.Parameter(f32(raw[k])))

# ---- cameras (scene/cameras.py) and the scene tables (scene/__init__.py:113-141) --------------------------------------------------------
fovx = 0.6911
fovy = focal2fov(fov2focal(fovx, W), H)
cams, cam_in = [], {}
for k in range(NV):
    az = math.radians(14.0 * k); el = math.radians(20.0)
    eye = 4.0 * np.array([math.cos(el) * math.cos(az), math.cos(el) * math.sin(az), math.sin(el)])
    z = -eye / np.linalg.norm(eye); x = np.cross(z, [0, 0, 1.0]); x /= np.linalg.norm(x); y = np.cross(z, x)
    R = np.stack([x, y, z], axis=1); T = -R.T @ eye
    c = Camera(colmap_id=k, R=R, T=T, FoVx=fovx, FoVy=fovy, image_width=W, image_height=H, image_path=None, image_name="v%d" % k, uid=k,
               preload_img=False, data_device="cpu")
    c.original_image = f32(rng.uniform(0, 1, (3, H, W)))
    cams.append(c)
    cam_in.update({"R%d" % k: R, "T%d" % k: T})
scene = SimpleNamespace()
Scene._initialize_train_buffers(scene, cams, SimpleNamespace(data_device="cpu"))
scene.rendered_depth_list = f32(rng.uniform(2.0, 5.0, (NV, 1, H, W)))
scene.getTrainCameras = lambda scale=1.0: cams
nearest = {0: [1, 2, 3, 4], 1: [0, 2], 2: [], 3: [4, 2, 1], 4: [3]}
for k, c in enumerate(cams):
    c.nearest_id = nearest[k]

pipe = SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False, debug=False)
pipe_py = SimpleNamespace(compute_cov3D_python=True, convert_SHs_python=True, debug=False)
args = SimpleNamespace(depth_error_threshold=0.02, shuffle_source_frame=False, multi_view_num=8, multi_view_max_angle=30, multi_view_min_dis=0.01,
                       multi_view_max_dis=1.5, enable_exposure_correction=False)
args_exp = SimpleNamespace(**{**args.__dict__, "enable_exposure_correction": True, "multi_view_num": 3})
bg = f32([0.1, 0.2, 0.3])
app = SimpleNamespace(appear_ab=f32(rng.normal(0, 0.2, (NV, 2))))

out = {"P": P, "W": W, "H": H, "NV": NV, "DEG": DEG, "fovx": fovx, "fovy": np.float64(fovy), "bg": bg.numpy(), "appear_ab": app.appear_ab.numpy(),
       "src_images": scene.original_image_list.numpy(), "rendered_depth_list": scene.rendered_depth_list.numpy()}
out.update({"raw_" + k: v for k, v in raw.items()}); out.update(cam_in)
for k, c in enumerate(cams):          # what Camera / Scene derived (checked against the stand-ins of ibgs_amd/simple_scene.py)
    out.update({"cam%d_wvt" % k: c.world_view_transform.numpy(), "cam%d_full" % k: c.full_proj_transform.numpy(), "cam%d_center" % k: c.camera_center.numpy(),
                "cam%d_nearest" % k: np.asarray(c.nearest_id, np.int64)})
    Kc, Ec = c.get_calib_matrix_nerf()
    out.update({"cam%d_K" % k: Kc.numpy(), "cam%d_E" % k: Ec.numpy()})
out.update(scene_wvts=scene.world_view_transforms.numpy(), scene_centers=scene.camera_centers.numpy(), scene_center_rays=scene.center_rays.numpy())

# ---- the two normal getters alone (gaussian_model.py:156-173) ---------------------------------------------------------------------------
with torch.no_grad():
    for k in (0, 3):
        n, off = pc.get_normal(cams[k])
        out.update({"get_normal_n_cam%d" % k: n.numpy(), "get_normal_off_cam%d" % k: off.numpy(),
                    "smallest_axis_n_cam%d" % k: pc.get_normal_w_smallest_axis(cams[k]).numpy()})

# ---- render() / render_depth() calls: (name, fn, camera, kwargs) -------------------------------------------------------------------------
CASES = [
    ("geo_learnt", "render", 0, dict(pipe=pipe, args=args, learnt_normal=True, nb_src_frames=3, buffer_length=4)),
    ("geo_axis_two_sources_app", "render", 1, dict(pipe=pipe, args=args, learnt_normal=False, nb_src_frames=4, buffer_length=5, depth_error_threshold=0.03,
                                                   app=True, scaling_modifier=0.7)),
    ("geo_no_neighbours", "render", 2, dict(pipe=pipe, args=args, learnt_normal=True, nb_src_frames=3, buffer_length=4, return_depth_normal=False)),
    ("colour_only_python_sh_cov", "render", 3, dict(pipe=pipe_py, args=args, learnt_normal=True, nb_src_frames=3, buffer_length=4, render_geo=False,
                                                    return_depth_normal=False)),
    ("depth_only_through_render", "render", 3, dict(pipe=pipe, args=args, learnt_normal=False, nb_src_frames=3, buffer_length=4, render_geo=False,
                                                    render_depth_only=True)),
    ("geo_find_closest_fresh_depths", "render", 2, dict(pipe=pipe, args=args_exp, learnt_normal=True, nb_src_frames=2, buffer_length=4,
                                                        do_find_closest_frame=True, do_render_src_depth=True)),
    ("render_depth_learnt", "render_depth", 1, dict(pipe=pipe, args=args, learnt_normal=True, nb_src_frames=3, buffer_length=4)),
    ("render_depth_axis_python", "render_depth", 4, dict(pipe=pipe_py, args=args, learnt_normal=False, nb_src_frames=3, buffer_length=6, depth_error_threshold=0.5)),
]


def put(prefix, d):
    none = []
    for k, v in d.items():
        if v is None:
            none.append(k)
        elif isinstance(v, torch.Tensor):
            out[prefix + k] = v.detach().numpy()
        else:
            out[prefix + k] = np.asarray(v)
    out[prefix + "_none"] = np.asarray(none, dtype="U64")


names = []
for name, fn, ci, kw in CASES:
    kw = dict(kw)
    pc.use_app = bool(kw.pop("app", False))
    first = len(ri.RECORDED)
    call = dict(viewpoint_camera=cams[ci], pc=pc, scene=scene, bg_color=bg, **kw)
    if fn == "render":
        res = gr.render(app_model=app if pc.use_app else None, **call)
    else:
        res = {"median_intersected_depth": gr.render_depth(**call)}
    calls = ri.RECORDED[first:]
    names.append(name)
    out["case_%s_fn" % name] = fn; out["case_%s_cam" % name] = ci; out["case_%s_ncalls" % name] = len(calls)
    put("case_%s_kw_" % name, {k: v for k, v in kw.items() if k not in ("pipe", "args")})
    out["case_%s_pipe_python" % name] = kw["pipe"] is pipe_py; out["case_%s_args_exposure" % name] = kw["args"] is args_exp
    out["case_%s_use_app" % name] = pc.use_app
    for j, (st, fkw, outs) in enumerate(calls):          # the LAST call is the main pass; earlier ones are the fresh source-depth passes
        put("case_%s_call%d_set_" % (name, j), st)
        put("case_%s_call%d_arg_" % (name, j), fkw)
        for i, t in enumerate(outs):
            out["case_%s_call%d_ret%d" % (name, j, i)] = t.numpy()
    for k, v in res.items():
        if k in ("viewspace_points", "viewspace_points_abs") and v is not None:
            out["case_%s_res_%s_requires_grad" % (name, k)] = bool(v.requires_grad); out["case_%s_res_%s_is_leaf" % (name, k)] = bool(v.is_leaf)
    put("case_%s_res_" % name, res)
out["cases"] = np.asarray(names, dtype="U64")
out["args_depth_error_threshold"] = args.depth_error_threshold
# the same tensors appear many times (the model's getters in every call, rasterizer outputs handed through to the dictionary): stored once
ri.save_deduped(os.path.join(HERE, "glue.npz"), out)
print("glue.npz written: %d cases, %d rasterizer calls recorded, %d arrays, %.0f KB" % (len(names), len(ri.RECORDED), len(out), os.path.getsize(os.path.join(HERE, "glue.npz")) / 1024))
