"""Synthetic "random-init" scenes and orbit cameras.

Used by bench.py, __graft_entry__.smoke() and the tests.  Mirrors how the reference initialises a
scene from a random point cloud (scene/dataset_readers.py:272-278, scene/gaussian_model.py:185-204)
and how ``Camera`` builds its matrices (scene/cameras.py:102-105 with utils/graphics_utils.py:151-184):
``world_view_transform`` and ``full_proj_transform`` are the TRANSPOSED matrices, flattened row-major.
Pure numpy so that CPU-only tests can use it.
"""
import math

import numpy as np

SH_C0 = 0.28209479177387814


def world_to_view(R, t):
    """getWorld2View2(R, t) with translate=0, scale=1 (utils/graphics_utils.py:151-162):
    R is the camera-to-world rotation, t the world-to-camera translation."""
    Rt = np.zeros((4, 4), dtype=np.float64)
    Rt[:3, :3] = R.T
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    return Rt.astype(np.float32)


def projection_matrix(znear, zfar, fovx, fovy):
    """getProjectionMatrix (utils/graphics_utils.py:164-184)."""
    tan_y = math.tan(fovy / 2); tan_x = math.tan(fovx / 2)
    top = tan_y * znear; bottom = -top; right = tan_x * znear; left = -right
    Pm = np.zeros((4, 4), dtype=np.float32)
    Pm[0, 0] = 2.0 * znear / (right - left)
    Pm[1, 1] = 2.0 * znear / (top - bottom)
    Pm[0, 2] = (right + left) / (right - left)
    Pm[1, 2] = (top + bottom) / (top - bottom)
    Pm[3, 2] = 1.0
    Pm[2, 2] = zfar / (zfar - znear)
    Pm[2, 3] = -(zfar * znear) / (zfar - znear)
    return Pm


def look_at_camera(eye, target=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0)):
    """Camera-to-world rotation R (columns: right, down, forward -- the COLMAP convention the
    reference uses) and world-to-camera translation t for a camera at `eye` looking at `target`."""
    eye = np.asarray(eye, np.float64); target = np.asarray(target, np.float64); up = np.asarray(up, np.float64)
    z = target - eye; z /= np.linalg.norm(z)
    x = np.cross(z, up); x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z], axis=1)
    t = -R.T @ eye
    return R, t


def camera_from_pose(W, H, R, t, fovx, fovy, znear=0.01, zfar=100.0):
    """The matrices `Camera.__init__` derives from a pose (scene/cameras.py:102-105): R camera-to-world rotation, t world-to-camera translation."""
    w2c = world_to_view(np.asarray(R, np.float64), np.asarray(t, np.float64))
    world_view_transform = np.ascontiguousarray(w2c.T)                       # transposed, cameras.py:102
    proj = projection_matrix(znear, zfar, fovx, fovy)
    full_proj_transform = (world_view_transform @ proj.T).astype(np.float32)  # cameras.py:104
    camera_center = np.linalg.inv(world_view_transform.astype(np.float64))[3, :3].astype(np.float32)
    return {
        "W": int(W), "H": int(H), "FoVx": fovx, "FoVy": fovy,
        "tanfovx": math.tan(fovx * 0.5), "tanfovy": math.tan(fovy * 0.5),
        "viewmatrix": world_view_transform, "projmatrix": np.ascontiguousarray(full_proj_transform),
        "campos": camera_center, "R": np.asarray(R, np.float32), "T": np.asarray(t, np.float32),
    }


def make_camera(W, H, fovx=0.6911, azimuth_deg=0.0, elevation_deg=20.0, radius=4.0, znear=0.01, zfar=100.0):
    az = math.radians(azimuth_deg); el = math.radians(elevation_deg)
    eye = radius * np.array([math.cos(el) * math.cos(az), math.cos(el) * math.sin(az), math.sin(el)])
    R, t = look_at_camera(eye)
    focal = W / (2.0 * math.tan(fovx / 2.0))
    fovy = 2.0 * math.atan(H / (2.0 * focal))
    return camera_from_pose(W, H, R, t, fovx, fovy, znear, zfar)


def quat_to_rotmat(q):
    """(P,4) (w,x,y,z) -> (P,3,3), the rotation of utils/general_utils.py:81-102 / forward.cu:172-176."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    Rm = np.empty((q.shape[0], 3, 3), dtype=q.dtype)
    Rm[:, 0, 0] = 1 - 2 * (y * y + z * z); Rm[:, 0, 1] = 2 * (x * y - r * z); Rm[:, 0, 2] = 2 * (x * z + r * y)
    Rm[:, 1, 0] = 2 * (x * y + r * z); Rm[:, 1, 1] = 1 - 2 * (x * x + z * z); Rm[:, 1, 2] = 2 * (y * z - r * x)
    Rm[:, 2, 0] = 2 * (x * z - r * y); Rm[:, 2, 1] = 2 * (y * z + r * x); Rm[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return Rm


def make_gaussians(P, seed, sh_degree=3, max_coeffs=16, opacity="init", extent=1.3, anisotropy=None, scale_sigma=0.0, cluster=0.0):
    """Random-init Gaussians.  Returns fp32 arrays: xyz (P,3), shs (P,Mc,3), scales (P,3) [activated],
    rotations (P,4) [normalised], opacities (P,1) [activated].

    anisotropy: None = the near-isotropic random init above (axis ratios <= ~2.5);
      "plane"  = the shape IBGS / PGSR Gaussians are trained towards (scene/gaussian_model.py:156-173 takes the smallest-scale axis
                 as the plane normal): one random axis shrunk to 10^-2 .. 10^-3 of the others, the disc 1.6x larger so that the
                 footprints stay comparable -- seen edge-on these project to ellipses of aspect up to ~100:1;
      "needle" = one random axis stretched 30x, the other two shrunk 3x (long thin splats);
      "mixed"  = a third of each.
    scale_sigma: > 0 = a heavy tail of sizes, what densification leaves behind (scene/gaussian_model.py:580-604 clones / splits by gradient
      and prunes by size only every few hundred steps): every Gaussian's three scales are multiplied by exp(N(0, sigma^2) - 1.5 sigma^2), a
      log-normal factor -- most Gaussians shrink (median factor e^(-1.5 sigma^2) = 0.22 at sigma = 1; the mean footprint area falls by
      e^(-sigma^2)), a per cent or two grow several-fold and cover hundreds of tiles.  At sigma = 1 the C3-sized plane scene keeps about as
      many (Gaussian, tile) pairs as the near-isotropic one has (10.7 M against 12.4 M), a third of them from 0.8 % of the Gaussians;
    cluster: > 0 = this fraction of the Gaussians (the first ones) moved into one blob of 0.3 x the extent around (0.5, 0.25, 0) -- an uneven image.
    The extra draws come from their own generators, so the other arrays do not depend on the mode."""
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(-extent, extent, size=(P, 3)).astype(np.float32)
    rgb = rng.uniform(0.0, 1.0, size=(P, 3)).astype(np.float32)
    shs = np.zeros((P, max_coeffs, 3), dtype=np.float32)
    shs[:, 0, :] = (rgb - 0.5) / SH_C0
    if sh_degree > 0:
        shs[:, 1:, :] = rng.normal(0.0, 0.05, size=(P, max_coeffs - 1, 3)).astype(np.float32)
    sbar = 0.65 * ((2 * extent) ** 3 / P) ** (1.0 / 3.0)
    scales = (sbar * np.exp(rng.normal(0.0, 0.3, size=(P, 3)))).astype(np.float32)
    if anisotropy is not None:
        arng = np.random.default_rng(1_000_003 + seed)
        axis = arng.integers(0, 3, size=P)
        mode = {"plane": np.zeros(P, int), "needle": np.ones(P, int), "mixed": arng.integers(0, 3, size=P)}[anisotropy]
        flat = 10.0 ** arng.uniform(-3.0, -2.0, size=P)
        f = np.ones((P, 3))
        rows = np.arange(P)
        pl, nd = mode == 0, mode == 1
        f[pl] = 1.6; f[rows[pl], axis[pl]] = flat[pl]
        f[nd] = 1.0 / 3.0; f[rows[nd], axis[nd]] = 30.0
        scales = (scales * f).astype(np.float32)
    if scale_sigma > 0:
        srng = np.random.default_rng(2_000_003 + seed)
        scales = (scales * np.exp(srng.normal(0.0, scale_sigma, size=(P, 1)) - 1.5 * scale_sigma * scale_sigma)).astype(np.float32)
    if cluster > 0:
        k = int(cluster * P)
        xyz[:k] = xyz[:k] * 0.3 + np.array([0.5, 0.25, 0.0], np.float32)
    q = rng.normal(0.0, 1.0, size=(P, 4))
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    if opacity == "init":
        op = np.full((P, 1), 0.1, dtype=np.float32)
    else:
        op = (1.0 / (1.0 + np.exp(-rng.normal(0.0, 2.0, size=(P, 1))))).astype(np.float32)
    return {"means3D": xyz, "shs": shs, "scales": scales, "rotations": q, "opacities": op}


def plane_all_map(xyz, scales, rotations, cam, normal=None, offset=None):
    """all_map (P,5) = [n_cam (3), 1.0, |plane distance in the camera frame|], the construction of
    gaussian_renderer/__init__.py:304-316 with the normal of scene/gaussian_model.py:148-173
    (smallest-scale axis, or a given per-Gaussian normal/offset; flipped to face the camera)."""
    xyz64 = xyz.astype(np.float32)
    if normal is None:
        Rm = quat_to_rotmat(rotations.astype(np.float32))
        idx = np.argmin(scales, axis=1)
        n = Rm[np.arange(xyz.shape[0]), :, idx]
        off = None
    else:
        n = normal / np.linalg.norm(normal, axis=1, keepdims=True)
        off = offset
    to_cam = cam["campos"][None, :] - xyz64
    neg = (n * to_cam).sum(-1) < 0.0
    n = n.copy(); n[neg] = -n[neg]
    V = cam["viewmatrix"]
    n_cam = n @ V[:3, :3]
    d_g = -(n * xyz64).sum(-1)
    if off is not None:
        d_g = d_g + (off.reshape(-1) * np.where(neg, -1.0, 1.0))
    d_cam = np.abs(d_g - (n_cam * V[3:4, :3]).sum(-1))
    am = np.zeros((xyz.shape[0], 5), dtype=np.float32)
    am[:, :3] = n_cam; am[:, 3] = 1.0; am[:, 4] = d_cam
    return am


def ref_to_src(ref_cam, src_cams):
    """(n,16) row-major true matrices W2C_src @ C2W_ref and (n,3) source camera centres
    (gaussian_renderer/__init__.py:258-265)."""
    ref_to_world = np.linalg.inv(ref_cam["viewmatrix"].T.astype(np.float64))
    mats, pos = [], []
    for s in src_cams:
        w2s = s["viewmatrix"].T.astype(np.float64)
        mats.append((w2s @ ref_to_world).astype(np.float32).reshape(16))
        pos.append(np.linalg.inv(w2s)[:3, 3].astype(np.float32))
    return np.stack(mats), np.stack(pos)


CONFIGS = {
    # BASELINE.json configs; seeds fixed there
    "C1": dict(P=10_000, W=400, H=400, sh_degree=3, seed=1),
    "C2": dict(P=100_000, W=800, H=800, sh_degree=0, seed=2),
    "C3": dict(P=1_000_000, W=1920, H=1080, sh_degree=3, seed=3),
    # not a BASELINE config: C3's Gaussians on a 720p frame (3 600 tiles: one resident round of the blend kernels even at 4 waves per SIMD);
    # used by occupancy experiments (bench.py --config C3_720p --wave-shape tile)
    "C3_720p": dict(P=1_000_000, W=1280, H=720, sh_degree=3, seed=3),
    # not a BASELINE config either: C3's Gaussians on the frame of the reference's `garden -r 4` recipe (1297 x 840 = 4 346 tiles, just above the
    # 4 096 tiles from which the blend kernels run one wave per tile) -- where the two wave shapes cross over
    "C3_800": dict(P=1_000_000, W=800, H=800, sh_degree=3, seed=3),          # (C2's frame with C3's Gaussians: 2 500 tiles)
    "C3_garden4": dict(P=1_000_000, W=1297, H=840, sh_degree=3, seed=3),
}


def make_scene(P, W, H, sh_degree=3, seed=1, view=0, opacity="init", with_planes=False, anisotropy=None, scale_sigma=0.0, cluster=0.0):
    """Oracle-style input dict for one view (colour path; see make_geo_inputs for the geo path)."""
    g = make_gaussians(P, seed, sh_degree=sh_degree, opacity=opacity, anisotropy=anisotropy, scale_sigma=scale_sigma, cluster=cluster)
    cam = make_camera(W, H, azimuth_deg=45.0 * view)
    inp = dict(g)
    inp.update({
        "W": W, "H": H, "tanfovx": cam["tanfovx"], "tanfovy": cam["tanfovy"],
        "viewmatrix": cam["viewmatrix"], "projmatrix": cam["projmatrix"], "campos": cam["campos"],
        "bg": np.zeros(3, np.float32), "sh_degree": sh_degree, "scale_modifier": 1.0,
        "render_geo": False, "render_depth_only": False, "n_src": 1, "buffer_length": 4, "depth_thr": 0.01,
    })
    if with_planes:
        inp["all_map"] = plane_all_map(g["means3D"], g["scales"], g["rotations"], cam)
    inp["_cam"] = cam
    return inp
