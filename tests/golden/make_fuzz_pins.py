"""Generator states of the pinned fuzz cases (tests/test_gpu_fuzz_pins.py): the numpy PCG64 state right BEFORE the sweep draws the case, so that the test replays
case k of `tools/fuzz_parity.py N SEED - MODE` without drawing cases 0 .. k - 1 first (20 s of normal deviates for case 117).
    python tests/golden/make_fuzz_pins.py        -> tests/golden/fuzz_pins.json
Data only: seeds, indices and generator states (integers)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import fuzz_cases as fc

PINS = [("parity", 9103, 9, "trained"), ("parity", 9103, 117, "trained")]
out = []
for kind, seed, index, mode in PINS:
    rng = np.random.default_rng(seed)
    for _ in range(index):
        fc.draw_parity_grads(rng, fc.draw_parity(rng, mode))
    st = rng.bit_generator.state
    out.append({"kind": kind, "seed": seed, "index": index, "mode": mode, "bit_generator": st["bit_generator"],
                "state": {"state": str(st["state"]["state"]), "inc": str(st["state"]["inc"])}, "has_uint32": st["has_uint32"], "uinteger": st["uinteger"]})
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_pins.json"), "w"), indent=1)
print("wrote", len(out), "pins")
