"""The op runs on the caller's current stream, like the reference's CUDA extension (rasterize_points.cu launches on the current stream; PyTorch's autograd runs a
node's backward on the stream its forward ran on).  The library keeps one piece of per-call host state outside the caller's arenas -- the pinned words R and the depth
sort's error word come back through, api.hip: RSlot -- and keys it by (device, stream): two streams must not see each other's R, tickets or errors, whichever threads
drive them.  Nothing in the suite left the default stream until round 6 (the slot was per THREAD until then: ADVICE r5).

Checked here in the deterministic backward mode, so that "the same" means bit for bit:
  * two workloads of different sizes on two side streams, forwards and backwards interleaved from ONE thread;
  * the same from TWO threads running concurrently, several steps each (the hinted path: the forward returns before its R is known and polls the stream's ticket);
  * a forward on one stream whose backward is queued while another stream's forward is in flight."""
import threading

import numpy as np
import pytest
import torch

from ibgs_amd import rasterizer
from tests import hipref
from tests.scenes import scene

pytestmark = pytest.mark.gpu

GRADS = ("means3D", "means2D", "opacities", "shs", "scales", "rotations")


def _upstream(inp, seed):
    return torch.as_tensor(np.random.default_rng(seed).standard_normal((3, int(inp["H"]), int(inp["W"]))).astype(np.float32), device="cuda")


def _step(inp, g):
    outs, lv, _ = hipref.run_forward(inp)
    (outs["color"] * g).sum().backward()
    return outs["color"].detach(), {k: lv[k].grad for k in GRADS}


def _same(a, b):
    return torch.equal(a[0], b[0]) and all(torch.equal(a[1][k], b[1][k]) for k in GRADS)


@pytest.fixture()
def deterministic():
    old = rasterizer.DETERMINISTIC
    rasterizer.DETERMINISTIC = True
    try:
        yield
    finally:
        rasterizer.DETERMINISTIC = old


def _workloads():
    a = scene(P=60000, W=480, H=320, deg=2, seed=5, opacity="trained")
    b = scene(P=25000, W=333, H=257, deg=1, seed=6)
    return (a, _upstream(a, 1)), (b, _upstream(b, 2))


def test_two_streams_interleaved_from_one_thread(deterministic):
    (a, ga), (b, gb) = _workloads()
    for _ in range(2):          # (the second call of a shape is the hinted one)
        ref_a, ref_b = _step(a, ga), _step(b, gb)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for rnd in range(3):
        with torch.cuda.stream(s1):
            oa, la, _ = hipref.run_forward(a)
        with torch.cuda.stream(s2):
            ob, lb, _ = hipref.run_forward(b)          # stream 2's forward is queued while stream 1's is in flight: its R, its ticket
        with torch.cuda.stream(s1):
            (oa["color"] * ga).sum().backward()
        with torch.cuda.stream(s2):
            (ob["color"] * gb).sum().backward()
        torch.cuda.synchronize()
        assert _same((oa["color"].detach(), {k: la[k].grad for k in GRADS}), ref_a), "stream 1, round %d" % rnd
        assert _same((ob["color"].detach(), {k: lb[k].grad for k in GRADS}), ref_b), "stream 2, round %d" % rnd
    rasterizer.check_async_errors()


def test_two_threads_two_streams(deterministic):
    (a, ga), (b, gb) = _workloads()
    for _ in range(2):
        ref_a, ref_b = _step(a, ga), _step(b, gb)
    torch.cuda.synchronize()
    results, errors = {}, []

    def worker(name, inp, g, ref, steps):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for i in range(steps):
                    got = _step(inp, g)
                    s.synchronize()
                    if not _same(got, ref):
                        errors.append("%s: step %d differs from the default stream's result" % (name, i))
                        return
            results[name] = steps
        except Exception as e:          # noqa: BLE001 -- reported by the main thread
            errors.append("%s: %r" % (name, e))

    ts = [threading.Thread(target=worker, args=("A", a, ga, ref_a, 12)), threading.Thread(target=worker, args=("B", b, gb, ref_b, 20))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    assert results == {"A": 12, "B": 20}


def test_default_stream_result_does_not_depend_on_a_busy_side_stream(deterministic):
    (a, ga), (b, gb) = _workloads()
    for _ in range(2):
        ref_a = _step(a, ga)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    big = torch.empty(64 << 20, device="cuda")
    with torch.cuda.stream(side):
        for _ in range(8):
            big.normal_()          # something long-running on the other stream
        ob, lb, _ = hipref.run_forward(b)
    got = _step(a, ga)              # default stream, queued while the side stream is still busy
    with torch.cuda.stream(side):
        (ob["color"] * gb).sum().backward()
    torch.cuda.synchronize()
    assert _same(got, ref_a)
