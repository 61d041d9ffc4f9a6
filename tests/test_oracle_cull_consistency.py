"""The tile cull's decisions are taken twice -- preprocess counts a Gaussian's tiles, the binning emits them -- and every build of the oracle
(plain, fma-contracted twin, float64) has to agree with ITSELF: the twin once did not (gcc contracted A C - B B differently at the two call
sites and `orc_bin` emitted a different number of entries than had been counted).  The decision helpers are compiled uncontracted in every build
now (`ORC_DECISION`, oracle/ibgs_oracle.c); the HIP side's counterpart is tests/test_gpu_trained_scene.py::test_row_runs_of_the_binning_equal_the_count."""
import numpy as np
import pytest

import oracle
from ibgs_amd import synthetic as syn


@pytest.mark.parametrize("seed,anisotropy,sigma", [(21, "plane", 1.0), (25, None, 0.0), (32, "mixed", 1.0), (7, "needle", 0.0)])          # (the first three broke the twin before the fix)
def test_every_build_emits_what_it_counted(seed, anisotropy, sigma):
    inp = syn.make_scene(2500, 640, 400, sh_degree=0, seed=seed, opacity="init", anisotropy=anisotropy, scale_sigma=sigma, cluster=0.3)
    inp["scales"] = (inp["scales"] * (7.0 if sigma == 0.0 else 14.0)).astype(np.float32)
    plain = oracle.forward(inp, cull=True)          # (forward() asserts that the binning emitted exactly the counted entries)
    r = plain["rect4"].astype(np.int64)
    rows_mode = ((r[:, 2] - r[:, 0]) * (r[:, 3] - r[:, 1]) > 256) & (plain["tmask"][:, 0] == 0)
    assert rows_mode.sum() > 20, "no rectangle beyond the mask: the row runs are not recomputed"
    assert int(plain["tiles_touched"].sum()) == plain["num_rendered"]
    for name in ("fma", "f64"):
        with oracle.variant(name):
            v = oracle.forward(inp, cull=True)
        assert int(v["tiles_touched"].sum()) == v["num_rendered"], name
