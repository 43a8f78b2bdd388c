"""N > 1 path: party-sharded encrypt with one CRS broadcast and no data-path collective.
CPU (gloo, world_size 2): shard plan + broadcast + gather plumbing with the C restatement as the
per-rank compute.  GPU: the same with the HIP path on both ranks."""
import os
import socket
import subprocess
import sys

import pytest

from pvw_rs_amd import dist as D

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return str(port)


def _run(mode, world=2):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), mode, str(r), str(world), port],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for p, out in zip(procs, outs):
        assert p.returncode == 0, out
    assert f"DIST_OK {mode}" in outs[0], outs[0]


def test_shard_ranges_partition():
    for n, k, world in [(4096, 256, 8), (22, 10, 2), (5, 3, 4), (16384, 512, 8), (7, 7, 7)]:
        parts = [D.shard_ranges(n, k, world, r) for r in range(world)]
        assert parts[0][0] == 0 and parts[-1][1] == n and parts[0][2] == 0 and parts[-1][3] == k
        for a, b in zip(parts, parts[1:]):
            assert a[1] == b[0] and a[3] == b[2]
        sizes = [p[1] - p[0] for p in parts]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        D.shard_ranges(4, 4, 2, 2)
    for Dn, world in [(8192, 8), (9, 2), (3, 4)]:
        parts = [D.shard_dealers(Dn, world, r) for r in range(world)]
        assert parts[0][0] == 0 and parts[-1][1] == Dn and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))


def test_world2_gloo_cpu():
    _run("oracle")


@pytest.mark.gpu
def test_world2_hip_on_one_gpu():
    _run("hip")

@pytest.mark.gpu
def test_world1_rccl_collectives_on_the_gpu():
    # backend "nccl" IS RCCL on ROCm: one rank on cuda:0 (a child process, started before anything here touches the
    # GPU) drives dist.broadcast_crs / load_broadcast_crs / all_gather_decrypted on device tensors, barrier and the
    # all_reduce(MAX) of bench.py, and checks the encrypt against the unsharded oracle result
    import torch.distributed as dist
    if not dist.is_nccl_available():
        pytest.skip("torch.distributed has no nccl backend")
    _run("nccl", world=1)


@pytest.mark.gpu
def test_bench_forced_process_group_over_rccl():
    # PVW_BENCH_FORCE_DIST=1: `bench.py --gpus 1` through the N > 1 code path with backend "nccl" (CRS broadcast as a
    # device tensor, barrier, max-reduce): one line, backend recorded, same workload
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, PVW_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port())
    for extra in ([], ["--path", "decrypt", "--config", "c5x16"]):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                            "--no-cpu", "--no-probe", "--sustain-seconds", "0"] + (extra or ["--config", "c2"]), env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        d = json.loads(lines[0])
        assert d["n_gpus"] == 1 and d["config"]["backend"] == "nccl" and d["value"] > 0


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu():
    # `python bench.py --gpus 2` end to end: it launches torch.distributed.run itself (child process, one rank
    # per "GPU"); rehearsed here with both ranks on cuda:0 and gloo for the barrier / max-reduce, because RCCL
    # cannot put two ranks on one device.  Checks the ONE JSON line of rank 0.
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, PVW_BENCH_BACKEND="gloo", PVW_BENCH_SAME_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--config", "c2", "--no-cpu", "--sustain-seconds", "0.2"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["world_size_observed"] == 2 and d["config"]["parties_total"] == 2048
    assert d["value"] > 0 and d["scaling"] == "weak" and d["roofline"]["frac"] > 0
