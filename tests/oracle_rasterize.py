"""The oracle behind the rasterizer's autograd surface (TEST INFRASTRUCTURE): a drop-in for `ibgs_amd.rasterizer.rasterize_gaussians`
whose forward and backward are oracle/ibgs_oracle.c on the host.  Lets a test drive the very same training loop once with the HIP
kernels' gradients and once with the oracle's (tests/test_gpu_example_training.py).  Colour path only (what the reference's warm-up
iterations use, train.py:289-292)."""
import numpy as np
import torch

import oracle


class _OracleRasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, means2D_abs, sh, opacities, scales, rotations, raster_settings):
        st = raster_settings
        if st.render_geo or st.render_depth_only:
            raise NotImplementedError("oracle_rasterize covers the colour path")
        n = lambda t: t.detach().cpu().numpy().astype(np.float32)
        H, W = int(st.image_height), int(st.image_width)
        inp = {"means3D": n(means3D), "shs": n(sh), "opacities": n(opacities).reshape(-1), "scales": n(scales), "rotations": n(rotations),
               "W": W, "H": H, "tanfovx": float(st.tanfovx), "tanfovy": float(st.tanfovy), "viewmatrix": n(st.viewmatrix), "projmatrix": n(st.projmatrix),
               "campos": n(st.campos), "bg": n(st.bg), "sh_degree": int(st.sh_degree), "scale_modifier": float(st.scale_modifier),
               "render_geo": False, "render_depth_only": False, "n_src": 1, "buffer_length": int(st.buffer_length)}
        fwd = oracle.forward(inp, cull=True)
        ctx.inp, ctx.fwd = inp, fwd
        dev = means3D.device
        P = means3D.shape[0]
        z = lambda *s: torch.zeros(*s, device=dev)
        radii = torch.as_tensor(fwd["radii"], device=dev)
        color = torch.as_tensor(fwd["color"], device=dev)
        ctx.mark_non_differentiable(radii)
        ctx.shapes = (P, sh.shape[1])
        return (color, radii, z(3, H, W), z(1, H, W), z(20, H, W), z(15, H, W), z(1, H, W), z(3, H, W), torch.zeros(1, H, W, dtype=torch.int32, device=dev))

    @staticmethod
    def backward(ctx, g_color, *unused):
        gb = oracle.backward(ctx.inp, ctx.fwd, g_color.detach().cpu().numpy().astype(np.float32))
        dev = g_color.device
        t = lambda a, shape: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev).reshape(shape)
        P, M = ctx.shapes
        return (t(gb["dL_dmeans3D"], (P, 3)), t(gb["dL_dmeans2D"], (P, 3)), t(gb["dL_dmeans2D_abs"], (P, 3)), t(gb["dL_dsh"], (P, M, 3)),
                t(gb["dL_dopacity"], (P, 1)), t(gb["dL_dscales"], (P, 3)), t(gb["dL_drotations"], (P, 4)), None)


def oracle_rasterize(means3D, means2D, means2D_abs, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, all_map, raster_settings,
                     plane_normal=None, plane_offset=None, plane_mode=0, sh_rest=None):
    if sh_rest is not None:          # the model's two SH arrays (rasterizer: shs_rest): the oracle takes their concatenation, autograd splits the gradient again
        sh = torch.cat((sh, sh_rest), dim=1)
    assert colors_precomp is None or colors_precomp.numel() == 0
    assert cov3Ds_precomp is None or cov3Ds_precomp.numel() == 0
    return _OracleRasterize.apply(means3D, means2D, means2D_abs, sh, opacities, scales, rotations, raster_settings)
