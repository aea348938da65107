"""kl_trace_rays_noalloc: the entry point that allocates nothing (device pointers in and out, intermediate storage in
the library's fixed device array) must return the hits of kl_trace_rays bit for bit, on consecutive calls that share
that array, the second smaller than the first."""
import numpy as np
import pytest

import klrender as kl
from test_gpu_scene import random_rays

pytestmark = pytest.mark.gpu


def _noalloc(r, torch, o, d):
    dev = torch.device("cuda", r.device)
    n = len(o)
    od, dd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    # outputs pre-filled with a pattern no hit can produce: every element must be overwritten
    t = torch.full((n,), -7.0, dtype=torch.float32, device=dev)
    prim = torch.full((n,), 12345, dtype=torch.int32, device=dev)
    uv = torch.full((n, 2), -7.0, dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)
    r.trace_rays_noalloc(od.data_ptr(), dd.data_ptr(), n, t.data_ptr(), prim.data_ptr(), uv.data_ptr())
    r.synchronize()
    return t.cpu().numpy(), prim.cpu().numpy().view(np.uint32), uv.cpu().numpy()


def _same(a, b, what):
    for x, y, name in zip(a, b, ("t", "prim", "uv")):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), "%s: %s differs" % (what, name)


def test_noalloc_trace_equals_trace_rays_on_two_calls_in_a_row():
    import torch
    s, q, t, m = kl.synth_scene(blobs=24, tris_per_blob=160, analytic_spheres=40, seed=3)
    with kl.Renderer(8, 8) as r:
        r.update_params(accel=kl.ACCEL_BVH)
        r.upload_scene(s, q, t, m)
        assert r.bvh_stats().flat_quads > 0                   # flat pass + traversal both run
        big = random_rays(200000, 21)
        small = random_rays(3001, 22)
        want_big, want_small = r.trace_rays(*big), r.trace_rays(*small)
        assert (want_big[1] != kl.MISS).any() and (want_big[2] != 0).any()
        got_big = _noalloc(r, torch, *big)                    # first call
        got_small = _noalloc(r, torch, *small)                # second call, smaller: the array still holds the first one's entries
        _same(got_big, want_big, "first call")
        _same(got_small, want_small, "second, smaller call")
        # the limit is a clear error, and nothing was queued for it
        o, d = random_rays(kl.NOALLOC_MAX_RAYS + 1, 23)
        with pytest.raises(kl.KlError, match="KL_NOALLOC_MAX_RAYS"):
            _noalloc(r, torch, o, d)
        full = random_rays(kl.NOALLOC_MAX_RAYS, 24)
        _same(_noalloc(r, torch, *full), r.trace_rays(*full), "largest call")
    # the built-in scene goes through the flat loop (no tree): same entry point, other kernel
    with kl.Renderer(8, 8) as r:
        r.load_builtin_scene()
        o, d = random_rays(5000, 25)
        _same(_noalloc(r, torch, o, d), r.trace_rays(o, d), "flat loop")
        o, d = random_rays(77, 26)
        _same(_noalloc(r, torch, o, d), r.trace_rays(o, d), "flat loop, smaller")


def test_two_contexts_on_one_gpu_do_not_race_on_the_shared_array():
    """Two contexts (two streams) on one GPU queue calls back to back without waiting in between: the library orders them
    one after the other on that GPU's one array, so both return the hits of kl_trace_rays."""
    import torch
    s, q, t, m = kl.synth_scene(blobs=24, tris_per_blob=160, analytic_spheres=40, seed=3)
    with kl.Renderer(8, 8) as a, kl.Renderer(8, 8) as b:
        assert a.stream() != b.stream() and a.stream() != 0
        for r in (a, b):
            r.update_params(accel=kl.ACCEL_BVH)
            r.upload_scene(s, q, t, m)
        dev = torch.device("cuda", a.device)
        rays = [random_rays(150000, 31), random_rays(9000, 32), random_rays(120000, 33), random_rays(500, 34)]
        want = [a.trace_rays(o, d) for o, d in rays]
        bufs = []
        for o, d in rays:
            n = len(o)
            bufs.append((torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev), torch.full((n,), -7.0, dtype=torch.float32, device=dev),
                         torch.full((n,), 12345, dtype=torch.int32, device=dev), torch.full((n, 2), -7.0, dtype=torch.float32, device=dev)))
        torch.cuda.synchronize(dev)                               # inputs complete before the calls, as the header asks
        for k, (od, dd, tt, pp, uu) in enumerate(bufs):           # a, b, a, b: no wait between the calls
            (a if k % 2 == 0 else b).trace_rays_noalloc(od.data_ptr(), dd.data_ptr(), len(od), tt.data_ptr(), pp.data_ptr(), uu.data_ptr())
        a.synchronize(); b.synchronize()
        for k, (od, dd, tt, pp, uu) in enumerate(bufs):
            _same((tt.cpu().numpy(), pp.cpu().numpy().view(np.uint32), uu.cpu().numpy()), want[k], "overlapped call %d" % k)
