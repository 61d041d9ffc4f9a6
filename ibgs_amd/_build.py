"""Builds libibgs_rast.so (the gfx950 HIP library behind include/ibgs_rast.h) in-tree with hipcc.

No torch.utils.cpp_extension here: under ROCm it would hipify the sources, and the library has no
torch types in its ABI anyway.  One hipcc invocation per translation unit (parallel), then a link.
preprocess.hip is compiled with -ffp-contract=off so its integer outputs are bit-identical to the
C oracle (see the header of that file).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libibgs_rast.so")
SOURCES = ["api", "preprocess", "scan_sort", "binning", "render_fwd", "render_bwd", "preprocess_bwd", "knn", "adam", "compact", "deterministic", "loss", "depth_normal", "activate"]
EXTRA = {
    "preprocess": ["-ffp-contract=off"],           # bit-identical to the oracle (see preprocess.hip)
    # no SLP packing: v_pk_*_f32 is not faster than two scalar VALU ops on gfx950 and costs v_mov / s_nop glue
    "render_fwd": ["-fno-slp-vectorize"],
    "render_bwd": ["-fno-slp-vectorize", "-fno-signed-zeros"],
}
ARCH = "gfx950"


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newest_dep():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))]
    deps.append(os.path.join(HERE, "..", "include", "ibgs_rast.h"))
    return max(os.path.getmtime(d) for d in deps)


def _code_only(text):
    """C / HIP source without comments and blank lines (string literals in these sources never hold comment markers)."""
    import re
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return "\n".join(l.rstrip() for l in text.split("\n") if l.strip())


def csrc_sha():
    """Fingerprint of the kernel sources (csrc/*.hip, csrc/*.h, include/ibgs_rast.h), comments and blank lines excluded.
    profiles/summarize.py stamps it into the committed rocprofv3 summaries and bench.py quotes profile-derived numbers only when the
    stamp matches this tree."""
    import hashlib
    h = hashlib.sha1()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(_code_only(open(os.path.join(CSRC, f), "r").read()).encode())
    h.update(_code_only(open(os.path.join(HERE, "..", "include", "ibgs_rast.h"), "r").read()).encode())
    return h.hexdigest()[:12]


# which translation unit a kernel of the step lives in (substring of its name -> TU): profile-derived numbers of a kernel stay valid while ITS
# unit (and the headers) are unchanged -- a host-only edit of api.hip does not make the blend kernels' counters stale
KERNEL_TU = (("render_fwd", "render_fwd"), ("pack_rgba", "render_fwd"), ("render_bwd", "render_bwd"), ("geo_window", "render_bwd"), ("tile_order", "render_bwd"),
             ("preprocess_bwd", "preprocess_bwd"), ("sh_grad", "preprocess_bwd"), ("preprocess_kernel", "preprocess"), ("sh_color", "preprocess"), ("mark_visible", "preprocess"),
             ("onesweep", "scan_sort"), ("radix", "scan_sort"), ("scan_", "scan_sort"), ("cell_", "binning"), ("expand_", "binning"), ("tile_ranges", "binning"),
             ("rendered_note", "api"), ("l1_", "loss"), ("depth_normal", "depth_normal"), ("activate_", "activate"), ("adam", "adam"), ("compact", "compact"), ("det_", "deterministic"), ("knn", "knn"))


def tu_of(kernel_name):
    for sub, tu in KERNEL_TU:
        if sub in kernel_name:
            return tu
    return None


def tu_shas():
    """{translation unit: fingerprint of its .hip + every header (csrc/*.h, include/ibgs_rast.h)}, comments and blank lines excluded."""
    import hashlib
    hdr = hashlib.sha1()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".h"):
            hdr.update(f.encode()); hdr.update(_code_only(open(os.path.join(CSRC, f), "r").read()).encode())
    hdr.update(_code_only(open(os.path.join(HERE, "..", "include", "ibgs_rast.h"), "r").read()).encode())
    out = {}
    for name in SOURCES:
        h = hdr.copy()
        h.update(_code_only(open(os.path.join(CSRC, name + ".hip"), "r").read()).encode())
        h.update(" ".join(EXTRA.get(name, [])).encode())
        out[name] = h.hexdigest()[:12]
    return out


def needs_build():
    if not os.path.exists(LIB):
        return True
    newest = _newest_dep()
    objs = [os.path.join(OBJ, n + ".o") for n in SOURCES]
    return any((not os.path.exists(o)) or os.path.getmtime(o) < newest for o in objs) or os.path.getmtime(LIB) < newest


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdr_time = max([os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".h")]
                   + [os.path.getmtime(os.path.join(HERE, "..", "include", "ibgs_rast.h"))])

    def compile_one(name):
        src = os.path.join(CSRC, name + ".hip")
        obj = os.path.join(OBJ, name + ".o")
        if (not force) and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_time):
            return obj
        cmd = [hipcc, "--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-c", src, "-o", obj] + EXTRA.get(name, [])
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or (not os.path.exists(LIB)) or os.path.getmtime(LIB) < max(os.path.getmtime(o) for o in objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
