"""`python bench.py --gpus N` as a plain command: with no WORLD_SIZE in the environment the parent starts the N
ranks itself (torch.distributed.run on 127.0.0.1) and relays rank 0's JSON line and the exit code.  Rehearsed with two ranks on the
one GPU of the box: fewer devices than ranks puts the process group on gloo; the one-shot record exchange between two
processes that share a device stalls at this shard size, which the bench must notice after its warm-up
and answer with the collective transport - the line comes out either way."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "ESPM_BENCH_BACKEND")}
    env["ESPM_XCHG_SELFTEST"] = "64"
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--no-extras", "--no-cpu"],
                        env=env, capture_output=True, text=True, timeout=900)
    assert pr.returncode == 0, pr.stderr[-3000:]
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, pr.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["warmup"] == 2 and out["value"] > 0 and out["nonfinite"] == 0
    cfg = out["config"]
    assert cfg["parallelism"] == "pixel-row shard x2"
    assert cfg["process_group"]["backend"] == "gloo" and cfg["process_group"]["ranks_per_device"] == 2
    assert len(cfg["per_rank"]) == 2 and all(r["rows"] == 256 and r["lost_peers"] == 0 for r in cfg["per_rank"])
    xr = cfg["record_exchange"]
    assert xr["transport"] in ("p2p", "collective")
    if xr["transport"] == "collective":
        assert xr.get("fell_back_from") == "p2p"
    assert out["loss_last"] < out["loss_first"]
    assert out["roofline"]["frac"] > 0


@pytest.mark.gpu
def test_bench_four_ranks_selflaunch_reports_per_rank_diagnostics():
    """`--gpus 4` self-launched on the one GPU (four processes fit the box's limit; gloo): the line parses with
    n_gpus 4, every rank reports its shard's projected single-rank time and its exchange wait, the transport (with the reason for a
    fallback), and the replicated W has the same checksum on all four ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "ESPM_BENCH_BACKEND")}
    env["ESPM_XCHG_SELFTEST"] = "32"
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "6", "--warmup", "2", "--no-extras", "--no-cpu"],
                        env=env, capture_output=True, text=True, timeout=1200)
    assert pr.returncode == 0, pr.stderr[-3000:]
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, pr.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["value"] > 0 and out["nonfinite"] == 0 and out["loss_last"] < out["loss_first"]
    ranks = out["config"]["per_rank"]
    assert len(ranks) == 4 and [r["rank"] for r in ranks] == [0, 1, 2, 3] and all(r["rows"] == 128 for r in ranks)
    assert len({r["w_crc32"] for r in ranks}) == 1, [r["w_crc32"] for r in ranks]
    for r in ranks:
        assert r["exchange_transport"] in ("p2p", "collective")
        assert (r["exchange_fallback_reason"] is None) == (r["exchange_transport"] == "p2p")
        assert r.get("local_iteration_us", 0) > 0 and "exchange_wait_us" in r, r.get("local_iteration_error")
    log = os.environ.get("ESPM_BENCH_LOG")   # (tools/analysis: the line kept under profiles/)
    if log:
        with open(log, "w") as f:
            f.write(lines[0] + "\n")


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                        capture_output=True, text=True, timeout=300)
    assert pr.returncode != 0 and "WORLD_SIZE=1" in (pr.stderr + pr.stdout)
