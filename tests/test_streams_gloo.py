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


def _run_bench(*flags, timeout=300):
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=timeout, env=env)


def test_bench_starts_its_own_ranks_and_relays_rank0_json():
    """`python bench.py --gpus 2` typed as is (no launcher): the parent starts torch.distributed.run as a child, the ranks rendezvous
    on 127.0.0.1, deal the global stream ids (stream_id mod n_gpu), time their steps behind barriers, reduce max/sum over ranks,
    and rank 0's JSON line arrives on the parent's stdout.  --dry-run replaces the GPU work by a sleep and RCCL by gloo -- the
    launcher, partition, StreamBatch and reduction code is the code of a real run."""
    import json
    r = _run_bench("--dry-run", "--gpus", "2", "--steps", "12", "--streams", "3")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["dry_run"] is True and j["n_gpus"] == 2 and j["ranks_seen"] == 2
    assert j["streams_of_rank0"] == [0, 2, 4] and j["latencies_gathered"] == 24
    assert j["evictions_all_ranks"] >= 6  # the product's StreamBatch ran on every rank: all 6 streams outgrew the 200-entry budget
    assert j["ms_per_step"] >= 4.0  # the slower rank (4 ms per step) sets the time
    # BASELINE.json configs[3]: at N > 1 the 64-streams-per-GPU leg runs on EVERY rank behind the same barriers (bench.run_streams64_all_ranks)
    s64 = j["streams64"]
    assert s64["ranks_seen"] == 2 and s64["streams_per_gpu"] == 64 and s64["streams_total"] == 128
    assert s64["ms_per_step"] >= 4.0 and s64["xrt"] > 0
    # every rank pinned itself to its own host cores before touching a GPU (streams.pin_rank_to_local_cores): disjoint sets
    assert s64["cores_pinned_all_ranks"] == s64["cores_pinned_distinct"] >= 2
    assert j["host_cores_of_rank0"]


def test_local_cores_follow_the_gpus_numa_node(tmp_path):
    """streams.local_cores_of_rank over a fake sysfs: 4 amdgpu cards, two per NUMA node (PCI order), one non-AMD card and one connector entry that
    must be ignored; ranks on one node split its cores, a visibility mask or too few cards fall back to an even split of the allowed cores."""
    sys.path.insert(0, ROOT)
    from infinisst_amd import streams
    layout = {"card0": ("0x1002", "0-7", "0000:05:00.0", True), "card1": ("0x1002", "0-7", "0000:15:00.0", True),
              "card2": ("0x1a03", "0-15", "0000:03:00.0", False),  # the board's VGA controller
              "card3": ("0x1002", "8-15", "0000:85:00.0", True), "card4": ("0x1002", "8-15", "0000:95:00.0", True)}
    for name, (vendor, cpus, bdf, vram) in layout.items():
        dev = tmp_path / "pci" / bdf
        dev.mkdir(parents=True)
        (dev / "vendor").write_text(vendor + "\n")
        (dev / "local_cpulist").write_text(cpus + "\n")
        if vram:
            (dev / "mem_info_vram_total").write_text("1\n")
        (tmp_path / name).mkdir()
        os.symlink(dev, tmp_path / name / "device")
    (tmp_path / "card0-DP-1").mkdir()
    allowed = list(range(16))
    nokfd = str(tmp_path / "no-kfd")  # no KFD topology: the cards count in PCI address order
    got = [streams.local_cores_of_rank(r, 4, sysfs=str(tmp_path), allowed=allowed, kfd=nokfd) for r in range(4)]
    assert got == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11], [12, 13, 14, 15]]
    assert streams.local_cores_of_rank(1, 2, sysfs=str(tmp_path), allowed=allowed, kfd=nokfd) == [4, 5, 6, 7]      # two ranks: both GPUs sit on node 0
    assert streams.local_cores_of_rank(5, 8, sysfs=str(tmp_path), allowed=allowed, kfd=nokfd) == [10, 11]           # more ranks than cards: even split
    assert streams.local_cores_of_rank(0, 1, sysfs=str(tmp_path / "missing"), allowed=allowed, kfd=nokfd) == allowed
    os.environ["HIP_VISIBLE_DEVICES"] = "3"
    try:
        assert streams.local_cores_of_rank(1, 4, sysfs=str(tmp_path), allowed=allowed, kfd=nokfd) == [4, 5, 6, 7]  # masked: the card order is unknown -> even split
    finally:
        del os.environ["HIP_VISIBLE_DEVICES"]
    # HIP device i = the i-th GPU agent of the KFD topology, which need not be PCI order: here KFD lists the node-1 GPUs first
    _fake_kfd(tmp_path / "kfd", ["0000:85:00.0", "0000:95:00.0", "0000:05:00.0", "0000:15:00.0"])
    got = [streams.local_cores_of_rank(r, 4, sysfs=str(tmp_path), allowed=allowed, kfd=str(tmp_path / "kfd")) for r in range(4)]
    assert got == [[8, 9, 10, 11], [12, 13, 14, 15], [0, 1, 2, 3], [4, 5, 6, 7]]
    # a KFD GPU without an amdgpu card in this tree: the two views disagree -> even split, not a guess
    _fake_kfd(tmp_path / "kfd2", ["0000:85:00.0", "0000:a5:00.0", "0000:05:00.0", "0000:15:00.0"])
    assert streams.local_cores_of_rank(1, 4, sysfs=str(tmp_path), allowed=allowed, kfd=str(tmp_path / "kfd2")) == [4, 5, 6, 7]


def _fake_kfd(root, bdfs):
    """A KFD topology tree: node 0 a CPU agent (no SIMDs), then one GPU agent per PCI address in the given order."""
    def node(i, simd, loc, domain=0):
        d = root / "topology" / "nodes" / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count {0 if simd else 8}\nsimd_count {simd}\ndomain {domain}\nlocation_id {loc}\ndrm_render_minor {128 + i}\n")
    node(0, 0, 0)
    for i, bdf in enumerate(bdfs):
        dom, bus, devfn = bdf.split(":")
        dev, fn = devfn.split(".")
        node(i + 1, 1024, (int(bus, 16) << 8) | (int(dev, 16) << 3) | int(fn), int(dom, 16))


def _fake_drm(root, cards):
    """cards: {name: (pci address, local_cpulist)} of amdgpu cards under a fake /sys/class/drm."""
    for name, (bdf, cpus) in cards.items():
        dev = root / "pci" / bdf
        dev.mkdir(parents=True)
        (dev / "vendor").write_text("0x1002\n")
        (dev / "local_cpulist").write_text(cpus + "\n")
        (dev / "mem_info_vram_total").write_text("1\n")
        (root / name).mkdir()
        os.symlink(dev, root / name / "device")


def test_bench_dry_run_on_eight_ranks(tmp_path):
    """BASELINE.json configs[3] as the launcher will see it on an 8-GPU node (none is in the pool): `bench.py --dry-run --gpus 8` -- eight ranks over gloo,
    512 streams dealt 64 per rank, the `streams64` block present, and every rank pinned to its own cores out of a fake sysfs with eight amdgpu cards on two
    NUMA nodes whose KFD order differs from their PCI order (scripts/infer/infinisst.sh:5-13: the reference runs one GPU per SLURM array task)."""
    import json
    ncpu = len(os.sched_getaffinity(0))
    if ncpu < 8:
        import pytest
        pytest.skip("needs 8 host cores for eight disjoint core sets")
    cpus = sorted(os.sched_getaffinity(0))[:8]
    node = [f"{cpus[0]}-{cpus[3]}" if cpus[3] - cpus[0] == 3 else ",".join(map(str, cpus[:4])),
            f"{cpus[4]}-{cpus[7]}" if cpus[7] - cpus[4] == 3 else ",".join(map(str, cpus[4:]))]
    bdfs = [f"0000:{b:02x}:00.0" for b in (0x05, 0x15, 0x25, 0x35, 0x85, 0x95, 0xa5, 0xb5)]
    _fake_drm(tmp_path / "drm", {f"card{i}": (bdfs[i], node[i // 4]) for i in range(8)})
    _fake_kfd(tmp_path / "kfd", [bdfs[i] for i in (4, 5, 6, 7, 0, 1, 2, 3)])
    os.environ["ISST_SYSFS_DRM"], os.environ["ISST_SYSFS_KFD"] = str(tmp_path / "drm"), str(tmp_path / "kfd")
    try:
        r = _run_bench("--dry-run", "--gpus", "8", "--steps", "6", "--streams", "2", timeout=600)
    finally:
        del os.environ["ISST_SYSFS_DRM"], os.environ["ISST_SYSFS_KFD"]
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["dry_run"] is True and j["n_gpus"] == 8 and j["ranks_seen"] == 8
    assert j["streams_of_rank0"] == [0, 8] and j["latencies_gathered"] == 48
    s64 = j["streams64"]
    assert s64["ranks_seen"] == 8 and s64["streams_per_gpu"] == 64 and s64["streams_total"] == 512
    assert s64["ms_per_step"] >= 16.0 and s64["xrt"] > 0          # the slowest rank (rank 7 sleeps 16 ms per step) sets the time
    assert s64["cores_pinned_all_ranks"] == s64["cores_pinned_distinct"] == 8  # eight disjoint core sets
    assert j["host_cores_of_rank0"] == [cpus[4]]                   # rank 0 = KFD GPU 0 = the first card of NUMA node 1, first of its four sharers


def test_bench_skips_the_64_stream_leg_on_every_rank_when_one_rank_fails_to_set_it_up():
    """a rank that fails while building the configs[3] leg must not strand the others at the timed barrier -- the ranks agree
    (TimingGroup.all_ok) and every one of them skips the leg; the job ends normally and says so."""
    import json
    r = _run_bench("--dry-run", "--gpus", "2", "--steps", "4", "--streams", "1", "--dry-fail-rank", "1", timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert "failed" in j["streams64"] and j["streams64"]["ranks_seen"] == 2
    assert j["ranks_seen"] == 2 and j["latencies_gathered"] == 8  # the headline leg ran to the end on both ranks


def test_bench_ends_non_zero_when_a_rank_fails_inside_the_timed_64_stream_leg():
    """the set-up failure above is agreed on before the timed barrier; a GPU step that raises AFTER that barrier -- inside the timed steps of
    the configs[3] leg -- must not strand the healthy ranks in the closing barrier / reductions either.  Every rank catches, the ranks agree on the host
    before the next collective (TimingGroup.all_ok), rank 0 still prints the line (headline intact, the leg marked failed) and the job exits non-zero
    well inside the launcher's timeout."""
    import json
    import time
    t0 = time.monotonic()
    r = _run_bench("--dry-run", "--gpus", "2", "--steps", "4", "--streams", "1", "--dry-fail-step-rank", "1", timeout=300)
    assert time.monotonic() - t0 < 240, "the healthy rank waited for the failed one"
    assert r.returncode != 0, r.stdout[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["ranks_seen"] == 2 and j["latencies_gathered"] == 8          # the headline leg ran to the end on both ranks
    assert "failed" in j["streams64"] and j["streams64"]["failed_in"] == "timed steps" and j["streams64"]["fatal"] is True


def test_bench_parent_reports_a_failing_rank():
    """Without --dry-run on a box without GPUs every rank fails loudly (no CPU fallback); the parent must exit non-zero and print no JSON line."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without GPU")
    r = _run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--toy", "--no-cpu-baseline")
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_timing_group_probe_cannot_strand_ranks(monkeypatch):
    """streams.TimingGroup decides ONCE whether RCCL works, and no outcome of the probe may leave ranks in different collectives: the
    all-reduce is launched asynchronously, the ranks agree over gloo whether every LAUNCH worked before anybody waits, completion is awaited on the
    host with a deadline, and the outcome is agreed again.  Driven here with stand-ins for torch.distributed (no GPU): a launch that raises, a launch
    that never completes, a healthy one."""
    sys.path.insert(0, ROOT)
    import torch
    from infinisst_amd import streams

    class Work:
        def __init__(self, done):
            self.done, self.waited = done, False

        def is_completed(self):
            return self.done

        def wait(self):
            self.waited = True

    log = []

    def install(mode):
        log.clear()
        monkeypatch.setattr(streams.dist, "is_available", lambda: True)
        monkeypatch.setattr(streams.dist, "is_initialized", lambda: True)
        monkeypatch.setattr(streams.dist, "get_world_size", lambda *a, **k: 2)
        monkeypatch.setattr(streams.dist, "new_group", lambda backend=None: "gloo-group")

        def all_reduce(t, op=None, group=None, async_op=False):
            if group == "gloo-group":
                log.append(("gloo", float(t.item())))
                return None
            log.append(("rccl", async_op))
            if mode == "raise":
                raise RuntimeError("hipIpcGetMemHandle: invalid argument")
            t.fill_(2.0)
            return Work(done=(mode == "ok"))
        monkeypatch.setattr(streams.dist, "all_reduce", all_reduce)

    install("raise")
    tg = streams.TimingGroup(torch.device("cpu"), probe_timeout_s=0.05)
    assert not tg.use_rccl and "hipIpcGetMemHandle" in tg.rccl_error
    assert log == [("rccl", True), ("gloo", 0.0)]            # the failing rank goes STRAIGHT to the agreement; nobody waited on RCCL
    install("hang")
    tg = streams.TimingGroup(torch.device("cpu"), probe_timeout_s=0.05)
    assert not tg.use_rccl and "did not complete" in tg.rccl_error
    assert log == [("rccl", True), ("gloo", 1.0), ("gloo", 0.0)]  # launched everywhere, then timed out on the host, then agreed to fall back
    install("ok")
    tg = streams.TimingGroup(torch.device("cpu"), probe_timeout_s=0.05)
    assert tg.use_rccl and tg.rccl_error is None and tg.describe() == "rccl"
    assert log == [("rccl", True), ("gloo", 1.0), ("gloo", 1.0)]
