"""N > 1 path on CPU: world_size 2, gloo.  Shards, the single conditioning broadcast, gather."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from cfgpp_amd import dist as D
    r, lr, w = D.init("gloo")
    total = 5
    shapes = [((1, 77, 64), torch.float16), ((total, 77, 64), torch.float16), ((total, 6), torch.float32)]
    payload = [None, None, None]
    if r == 0:
        g = torch.Generator().manual_seed(0)
        payload = [torch.randn(s, generator=g).to(dt) for s, dt in shapes]
    got = D.broadcast_conditioning(payload, shapes, torch.device("cpu"))
    lo, hi = D.shard_range(total, r, w)
    local = got[1][lo:hi].float().sum(dim=(1, 2), keepdim=False).reshape(-1, 1)      # a per-chain "result"
    counts = [D.shard_range(total, i, w)[1] - D.shard_range(total, i, w)[0] for i in range(w)]
    gathered = D.gather_rows(local, counts)
    mx = D.max_over_ranks(float(r + 1), torch.device("cpu"))
    D.barrier()
    q.put((r, [float(x.float().sum()) for x in got], (lo, hi), None if gathered is None else gathered.reshape(-1).tolist(), mx))


def test_world2_broadcast_shard_gather():
    world, port = 2, 29533
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    res.sort()
    assert res[0][1] == res[1][1]                       # every rank holds the same conditioning
    assert res[0][2] == (0, 3) and res[1][2] == (3, 5)  # contiguous shards, sizes differ by <= 1
    g = torch.Generator().manual_seed(0)
    ref = [torch.randn(s, generator=g).to(dt) for s, dt in [((1, 77, 64), torch.float16), ((5, 77, 64), torch.float16), ((5, 6), torch.float32)]]
    want = ref[1].float().sum(dim=(1, 2)).tolist()
    assert res[0][3] is not None and res[1][3] is None
    assert all(abs(a - b) < 1e-3 for a, b in zip(res[0][3], want))
    assert res[0][4] == res[1][4] == 2.0


def test_shard_range_covers_everything():
    from cfgpp_amd.dist import shard_range
    for n in (1, 7, 16, 64):
        for w in (1, 2, 4, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
