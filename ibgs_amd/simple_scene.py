"""Minimal stand-ins for the reference's ``Camera`` / ``GaussianModel`` / ``Scene`` objects -- just the
attributes that ``renderer.render()`` reads (reference scene/cameras.py:51-134,
scene/gaussian_model.py:127-178, scene/__init__.py:113-141).  They let bench.py, smoke() and the tests
drive the renderer without the reference's data-loading stack (COLMAP, PLY, PIL ...), which is out of
scope here.  Activations are the reference's: exp for scales, sigmoid for opacity, L2-normalised
quaternions."""
import math
from types import SimpleNamespace

import numpy as np
import torch

from . import synthetic as syn

__all__ = ['SimpleNamespace']


class SimpleCamera:
    def __init__(self, cam, uid=0, device="cpu"):
        self.uid = uid
        self.image_width = cam["W"]; self.image_height = cam["H"]
        self.FoVx = cam["FoVx"]; self.FoVy = cam["FoVy"]
        self.R = cam["R"]; self.T = cam["T"]
        self.world_view_transform = torch.as_tensor(cam["viewmatrix"], device=device)
        self.full_proj_transform = torch.as_tensor(cam["projmatrix"], device=device)
        self.camera_center = torch.as_tensor(cam["campos"], device=device)
        self.Fx = self.image_width / (2 * math.tan(self.FoVx / 2)); self.Fy = self.image_height / (2 * math.tan(self.FoVy / 2))
        self.Cx = 0.5 * self.image_width; self.Cy = 0.5 * self.image_height
        self.nearest_id = []

    def get_calib_matrix_nerf(self, scale=1.0):
        K = torch.tensor([[self.Fx / scale, 0, self.Cx / scale], [0, self.Fy / scale, self.Cy / scale], [0, 0, 1]]).float()
        return K, self.world_view_transform.transpose(0, 1).contiguous()


class SimpleGaussians(torch.nn.Module):
    """Raw (pre-activation) parameters as nn.Parameters; getters apply the reference's activations."""
    standard_activations = True          # exp / F.normalize / sigmoid on `_scaling` / `_rotation` / `_opacity`: the renderer may fuse them (ibgs_amd/activations.py)

    def __init__(self, g, sh_degree=3, device="cpu"):
        super().__init__()
        t = lambda a: torch.nn.Parameter(torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device))
        self.max_sh_degree = int(round(math.sqrt(g["shs"].shape[1]))) - 1
        self.active_sh_degree = sh_degree
        self._xyz = t(g["means3D"])
        self._features_dc = t(g["shs"][:, :1, :]); self._features_rest = t(g["shs"][:, 1:, :])
        self._scaling = t(np.log(g["scales"]))
        self._rotation = t(g["rotations"])
        op = np.clip(g["opacities"], 1e-6, 1 - 1e-6)
        self._opacity = t(np.log(op / (1 - op)))
        self._normal = t(g.get("normal", np.tile(np.array([[0.0, 0.0, 1.0]], np.float32), (g["means3D"].shape[0], 1))))
        self._offset = t(g.get("offset", np.zeros((g["means3D"].shape[0], 1), np.float32)))
        self.use_app = False

    @property
    def get_xyz(self): return self._xyz
    @property
    def get_scaling(self): return torch.exp(self._scaling)
    @property
    def get_rotation(self): return torch.nn.functional.normalize(self._rotation)
    @property
    def get_opacity(self): return torch.sigmoid(self._opacity)
    @property
    def get_features(self): return torch.cat((self._features_dc, self._features_rest), dim=1)

    def rotation_matrices(self):
        q = self.get_rotation
        r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                            2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                            2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1).view(-1, 3, 3)

    def get_covariance(self, scaling_modifier=1):
        L = self.rotation_matrices() * (scaling_modifier * self.get_scaling)[:, None, :]
        S = L @ L.transpose(1, 2)
        return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=-1)

    def _flip(self, n, view_cam):
        to_cam = view_cam.camera_center.to(n.device) - self._xyz
        neg = (n * to_cam).sum(-1) < 0.0
        n = torch.where(neg[:, None], -n, n)
        return n, neg

    def get_normal_w_smallest_axis(self, view_cam):
        Rm = self.rotation_matrices()
        idx = self.get_scaling.min(dim=-1)[1][..., None, None].expand(-1, 3, -1)
        n = Rm.gather(2, idx).squeeze(2)
        return self._flip(n, view_cam)[0]

    def get_normal(self, view_cam):
        n = self._normal / torch.norm(self._normal, dim=1, keepdim=True)
        n, neg = self._flip(n, view_cam)
        off = self._offset * (neg.float() * -2 + 1).unsqueeze(-1)
        return n, off

    def raster_params(self):
        """Leaf tensors whose gradients the view-parallel step all-reduces."""
        return [self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation, self._opacity,
                self._normal, self._offset]


class SimpleScene:
    """Source-image stack, cached depth maps and camera tables like scene/__init__.py:113-141."""

    def __init__(self, cameras, images=None, device="cpu"):
        self.cameras = cameras
        H, W = cameras[0].image_height, cameras[0].image_width
        n = len(cameras)
        self.original_image_list = images if images is not None else torch.zeros(n, 3, H, W, device=device)
        self.rendered_depth_list = torch.zeros(n, 1, H, W, device=device)
        self.world_view_transforms = torch.stack([c.world_view_transform.T for c in cameras]).to(device)
        self.camera_centers = torch.stack([c.camera_center for c in cameras]).to(device)
        rays = [torch.tensor([0.0, 0.0, 1.0]) @ torch.as_tensor(c.R, dtype=torch.float32).T for c in cameras]
        self.center_rays = torch.stack(rays).to(device)

    def getTrainCameras(self):
        return self.cameras


def default_pipe(debug=False):
    return SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False, debug=debug)


def default_args():
    return SimpleNamespace(depth_error_threshold=0.01, shuffle_source_frame=False, multi_view_num=8,
                           multi_view_max_angle=30, multi_view_min_dis=0.01, multi_view_max_dis=1.5,
                           enable_exposure_correction=False)


def orbit_cameras(W, H, n_views=8, device="cpu", nearest=4):
    cams = [SimpleCamera(syn.make_camera(W, H, azimuth_deg=360.0 / n_views * k), uid=k, device=device) for k in range(n_views)]
    for k, c in enumerate(cams):   # nearest other views by camera-centre distance
        d = [(float(torch.norm(c.camera_center - o.camera_center)), j) for j, o in enumerate(cams) if j != k]
        c.nearest_id = [j for _, j in sorted(d)[:nearest]]
    return cams
