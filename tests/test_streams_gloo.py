"""N>1 path on CPU: world_size-2 gloo run of the stream partition + whole-job throughput reduction bench.py uses."""
import os
import sys

import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from infinisst_amd import streams
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = streams.assign_streams(7, rank, world)
    elapsed = 2.0 if rank == 0 else 4.0  # rank 1 is the straggler
    audio_s = 0.96 * 10 * len(mine)
    xrt = streams.whole_job_xrt(audio_s, elapsed)
    lat = streams.gather_floats([0.01 * (rank + 1)] * 3)
    dist.barrier()
    q.put((rank, mine, xrt, lat))
    dist.destroy_process_group()


def test_stream_partition_and_aggregate_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, x0, l0), (r1, s1, x1, l1) = res
    assert s0 == [0, 2, 4, 6] and s1 == [1, 3, 5]  # stream_id mod n_gpu, disjoint cover
    expect = 0.96 * 10 * 7 / 4.0  # all ranks' audio / slowest rank's time
    assert abs(x0 - expect) < 1e-9 and abs(x1 - expect) < 1e-9
    assert l0 == l1 == [0.01] * 3 + [0.02] * 3
