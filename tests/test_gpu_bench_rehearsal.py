"""Rehearsal of the driver's multi-rank bench run on a ONE-GPU box: bench.py itself (not a helper) is launched
at world size 2 -- through torch.distributed.run AND through its own launcher (`python bench.py --gpus 2` with no
WORLD_SIZE in the environment: the parent starts the ranks itself and never touches the GPU) --,
OHS_BENCH_DIST_BACKEND=gloo (table broadcast and barriers over gloo, both ranks on cuda:0), in weak and in strong
(--total-streams) mode.  Checks the argument / environment /
barrier plumbing, that the rank shards are disjoint and cover the job, and that every rank's outputs equal the
single-rank run of the same stream ids bit for bit (streams share only read-only tables: SURVEY section 8e)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "2", "--warmup", "1", "--frames", "20480", "--no-cpu-baseline", "--no-extras"]


def _run(tmp_path, tag, world, extra, port, launcher="torchrun"):
    dump = str(tmp_path / tag)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OHS_BENCH_DIST_BACKEND="gloo", OHS_BENCH_DUMP=dump)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "OHS_BENCH_LAUNCHER"):
        env.pop(k, None)
    if world == 1 or launcher == "self":
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), *COMMON, *extra]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
               "--gpus", str(world), *COMMON, *extra]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]          # rank 0 prints ONE JSON line
    rec = json.loads(lines[0])
    shards = {}
    for r in range(world):
        a = np.load(dump + f".rank{r}.npy")
        shards[r] = (int(a[0]), int(a[1]), a[2:])
    return rec, shards


@pytest.mark.parametrize("launcher,mode", [("torchrun", "weak"), ("torchrun", "strong"), ("self", "weak"), ("self", "strong")])
def test_bench_two_ranks_on_one_gpu_matches_single_rank(tmp_path, launcher, mode):
    if mode == "weak":
        two, one = ["--streams", "6"], ["--streams", "12"]
    else:
        two, one = ["--total-streams", "12"], ["--total-streams", "12"]
    rec2, sh2 = _run(tmp_path, "w2" + mode, 2, two, 29541 if mode == "weak" else 29543, launcher)
    rec1, sh1 = _run(tmp_path, "w1" + mode, 1, one, 0)
    # placement evidence gathered from every rank (here both ranks share cuda:0: ONE distinct device)
    assert rec2["rccl_world_size"] == 2 and len(rec2["ranks"]) == 2 and rec2["distinct_devices"] == 1
    assert [r["rank"] for r in rec2["ranks"]] == [0, 1] and [r["streams"] for r in rec2["ranks"]] == [6, 6]
    assert rec2["ranks"][0]["pid"] != rec2["ranks"][1]["pid"]
    assert (rec2["launcher"] == "self") == (launcher == "self")
    assert ("expected_strong_scaling" in rec2) == (mode == "strong")
    assert "ranks" not in rec1
    assert rec2["n_gpus"] == 2 and rec1["n_gpus"] == 1
    assert rec2["scaling"] == mode and rec2["config"]["total_streams"] == 12
    assert rec2["config"]["streams_per_gpu"] == 6 and rec2["finite"] and rec2["value"] > 0
    assert rec2["steps"] == 2 and rec2["warmup"] == 1 and rec2["unit"] == "Msamples/s"
    # disjoint, contiguous, covering
    assert (sh2[0][0], sh2[0][1], sh2[1][0], sh2[1][1]) == (0, 6, 6, 12)
    assert (sh1[0][0], sh1[0][1]) == (0, 12)
    full = sh1[0][2].reshape(12, 2, 4096)
    for r in (0, 1):
        lo, hi, data = sh2[r]
        assert np.array_equal(data.reshape(hi - lo, 2, 4096), full[lo:hi]), f"rank {r} differs from the single-rank run"


def _bench(args, env_extra=None, timeout=900):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **(env_extra or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "OHS_BENCH_LAUNCHER"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True,
                         timeout=timeout, cwd=ROOT)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def test_node_batch_mode_is_the_headline_at_one_gpu(tmp_path):
    """`bench.py --gpus 1 --node-batch` -- ONE process through ohs_node_batch_* only (the C-ABI path a Rust host binds) --
    against the plain headline run on the same box: same bits, the same rate within 2 % (the verdict asks for 1 %; the
    measured ratio is printed and recorded in profiles/), and the record's device evidence."""
    flags = ["--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-extras"]
    plain = _bench(["--gpus", "1", *flags], {"OHS_BENCH_DUMP": str(tmp_path / "plain")})
    node = _bench(["--gpus", "1", "--node-batch", *flags], {"OHS_BENCH_DUMP": str(tmp_path / "node")})
    assert node["mode"] == "node_batch" and node["metric"] == plain["metric"] and node["unit"] == "Msamples/s"
    assert node["n_gpus"] == 1 and node["steps"] == 10 and node["warmup"] == 3 and node["finite"]
    assert node["rccl_comm_size"] == 1 and node["rccl_loaded"] is True
    assert len(node["slots"]) == 1 and node["distinct_devices"] == 1
    sl = node["slots"][0]
    assert sl["device"] == 0 and sl["first_stream"] == 0 and sl["n_streams"] == 256 and sl["pci_bus_id"].count(":") == 2
    assert 0 < sl["ms_per_step"] <= node["ms_per_step"] * 1.02
    assert node["config"]["total_streams"] == plain["config"]["total_streams"] == 256
    assert "roofline" in node and node["roofline"]["kernel"] == "k_eq_ring"
    ratio = node["value"] / plain["value"]
    print(f"node-batch / plain headline: {node['value']:.1f} / {plain['value']:.1f} Msamples/s = {ratio:.4f}")
    assert 0.98 <= ratio <= 1.02, ratio
    a, b = np.load(str(tmp_path / "plain") + ".rank0.npy"), np.load(str(tmp_path / "node") + ".slot0.npy")
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_two_rank_record_carries_the_node_batch_leg(tmp_path):
    """At N > 1 rank 0 measures the node API in a child run while the other ranks wait on the rendezvous store (CPU side).
    Rehearsed at world size 2 over gloo on this one-GPU box: the child then drives the one device there is."""
    rec = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--frames", "20480", "--streams", "6", "--no-cpu-baseline"],
                 {"OHS_BENCH_DIST_BACKEND": "gloo"})
    assert rec["n_gpus"] == 2 and rec["rccl_world_size"] == 2 and rec["launcher"] == "self"
    nbr = rec["node_batch"]
    assert "error" not in nbr, nbr
    assert nbr["rccl_comm_size"] == 1 and nbr["finite"] and nbr["value"] > 0 and len(nbr["slots"]) == 1
    assert nbr["slots"][0]["n_streams"] == 6
