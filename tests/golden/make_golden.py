"""Generates tests/golden/*.npz IN THE BUILD CONTAINER by importing the reference's own pure-python
helpers from /root/reference (read-only).  The fixtures are data only (inputs + the reference's
outputs); nothing from the reference is copied.  Re-run:  python tests/golden/make_golden.py

  sh_eval.npz      utils/sh_utils.py: eval_sh for degrees 0..3      -> pins the SH polynomial (A1)
  camera_mats.npz  utils/graphics_utils.py: getWorld2View2, getProjectionMatrix, and the
                   Camera matrix algebra of scene/cameras.py:102-105  -> pins the matrix conventions (A.1)
  metrics.npz      utils/image_utils.py: psnr, utils/loss_utils.py: l1_loss, ssim -> parity metric definitions
  depth_normal.npz utils/graphics_utils.py: normal_from_depth_image  -> glue row G(vii)
"""
import math
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))

from utils.sh_utils import eval_sh  # noqa: E402
from utils.graphics_utils import getWorld2View2, getProjectionMatrix, normal_from_depth_image, fov2focal, focal2fov  # noqa: E402
from utils.image_utils import psnr  # noqa: E402
from utils.loss_utils import l1_loss, ssim  # noqa: E402

rng = np.random.default_rng(20240501)

# ---- SH -------------------------------------------------------------------------------------
dirs = rng.normal(size=(64, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
sh = rng.normal(size=(64, 3, 16))
out = {"dirs": dirs.astype(np.float32), "sh": sh.astype(np.float32)}
for deg in range(4):
    out["deg%d" % deg] = eval_sh(deg, torch.tensor(out["sh"]), torch.tensor(out["dirs"])).numpy()
np.savez_compressed(os.path.join(HERE, "sh_eval.npz"), **out)

# ---- cameras --------------------------------------------------------------------------------
cams = {}
W, H, fovx = 400, 300, 0.6911
for k in range(8):
    az = math.radians(45.0 * k); el = math.radians(20.0)
    eye = 4.0 * np.array([math.cos(el) * math.cos(az), math.cos(el) * math.sin(az), math.sin(el)])
    z = -eye / np.linalg.norm(eye); x = np.cross(z, [0, 0, 1.0]); x /= np.linalg.norm(x); y = np.cross(z, x)
    R = np.stack([x, y, z], axis=1); T = -R.T @ eye
    fovy = focal2fov(fov2focal(fovx, W), H)
    wvt = torch.tensor(getWorld2View2(R, T, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
    proj = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
    full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
    center = wvt.inverse()[3, :3]
    cams["R%d" % k] = R; cams["T%d" % k] = T
    cams["wvt%d" % k] = wvt.numpy(); cams["full%d" % k] = full.numpy(); cams["center%d" % k] = center.numpy()
    cams["fovy%d" % k] = np.float64(fovy)
cams["W"] = W; cams["H"] = H; cams["fovx"] = fovx
np.savez_compressed(os.path.join(HERE, "camera_mats.npz"), **cams)

# ---- metrics --------------------------------------------------------------------------------
a = torch.tensor(rng.uniform(0, 1, size=(4, 3, 24, 32)).astype(np.float32))
b = (a + torch.tensor(rng.normal(0, 0.05, size=a.shape).astype(np.float32))).clamp(0, 1)
np.savez_compressed(os.path.join(HERE, "metrics.npz"), a=a.numpy(), b=b.numpy(),
                    psnr=psnr(a, b).numpy(), l1=np.float32(l1_loss(a, b).item()),
                    # utils/loss_utils.py:34-64: 11x11 Gaussian window (sigma 1.5), zero padding, C1 = 0.01^2, C2 = 0.03^2
                    ssim=np.float32(ssim(a, b).item()), ssim_per_image=ssim(a, b, size_average=False).numpy())

# ---- depth -> normal ------------------------------------------------------------------------
depth = torch.tensor((3.0 + 0.3 * rng.normal(size=(20, 28))).astype(np.float32))
K = torch.tensor([[300.0, 0, 14.0], [0, 310.0, 10.0], [0, 0, 1]])
n = normal_from_depth_image(depth, K, torch.eye(4))
np.savez_compressed(os.path.join(HERE, "depth_normal.npz"), depth=depth.numpy(), K=K.numpy(), normal=n.numpy())
print("golden fixtures written to", HERE)

# ---- 3D covariance (gaussian_model.py:38-42 + utils/general_utils.py:67-120) -------------------------------------
# build_rotation / build_scaling_rotation / strip_symmetric allocate with device="cuda"; there is no GPU in the build
# container, so the allocation calls are pointed at the CPU for the duration of this section -- the arithmetic is the
# reference's own.  Pins the oracle's computeCov3D restatement (forward.cu:118-153 does the same product on the GPU) and the
# renderer's `get_covariance` convention (row-major upper triangle xx, xy, xz, yy, yz, zz).
import utils.general_utils as gu  # noqa: E402

_zeros = torch.zeros
torch.zeros = lambda *a, **k: _zeros(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
try:
    scal = torch.tensor(np.exp(rng.normal(-2.0, 1.0, size=(96, 3))).astype(np.float32))
    quat = torch.tensor(rng.normal(size=(96, 4)).astype(np.float32))          # NOT normalised: build_rotation normalises
    cov = {}
    for mod in (1.0, 0.5):
        L = gu.build_scaling_rotation(mod * scal, quat)
        cov["cov6_mod%g" % mod] = gu.strip_symmetric(L @ L.transpose(1, 2)).numpy()
    cov["rotmat"] = gu.build_rotation(quat).numpy()
finally:
    torch.zeros = _zeros
np.savez_compressed(os.path.join(HERE, "cov3d.npz"), scales=scal.numpy(), quats=quat.numpy(), **cov)
print("cov3d.npz written")
