"""World-size-2 gloo test of the only collective on the path: the gather of the per-rank
int32 (row, col) results to rank 0.  Runs on CPU."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    import pawsometracker_jl_amd as pt
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = pt.shard_range(n_total, rank, world)
    # stand-in for the rank's detect() output: row = global window id, col = rank
    local = torch.stack([torch.arange(lo, hi, dtype=torch.int32), torch.full((hi - lo,), rank, dtype=torch.int32)], 1)
    out = pt.gather_positions(local, n_total)
    # the asynchronous form (bench.py overlaps it with the next batch): several in flight, same answers
    handles = [pt.gather_positions(local + k, n_total, async_op=True) for k in range(3)]
    outs = [h.wait() for h in handles]
    if rank == 0:
        for k in range(3):
            assert torch.equal(outs[k], out + k)
        q.put(out.tolist())
    else:
        assert out is None and all(o is None for o in outs)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_gather_positions_world2():
    for n_total in (7, 8):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
        for p in procs:
            p.start()
        got = q.get(timeout=120)
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        assert [g[0] for g in got] == list(range(n_total))
        lo, hi = (n_total + 1) // 2, n_total
        assert [g[1] for g in got] == [0] * lo + [1] * (hi - lo)
