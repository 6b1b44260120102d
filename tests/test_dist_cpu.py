"""world_size-2 gloo test of the N>1 harness (batch sharding, barrier, max-over-ranks timing)
that bench.py uses on the GPUs with RCCL.  Runs on CPU."""
import json
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_partitions_exactly():
    from helib_amd.dist import shard
    for total in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            parts = [shard(total, world, r) for r in range(world)]
            assert sum(c for _, c in parts) == total
            pos = 0
            for s, c in parts:
                assert s == pos
                pos += c
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def test_two_rank_gloo_timing_and_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(f"""
        import sys, time, json
        sys.path.insert(0, {ROOT!r})
        from helib_amd.dist import Group, shard
        g = Group(backend="gloo")
        start, count = shard(513, g.world, g.rank)
        g.barrier()
        t0 = time.perf_counter()
        time.sleep(0.05 * (g.rank + 1))          # rank 1 is the slow one
        g.barrier()
        dt = g.max_over_ranks(time.perf_counter() - t0)
        total = g.sum_over_ranks(count)
        print(json.dumps({{"rank": g.rank, "start": start, "count": count, "dt": dt, "total": total}}))
        g.close()
    """))
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(r), LOCAL_RANK=str(r),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=120)
        assert p.returncode == 0, e
        import json
        outs.append(json.loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    assert outs[0]["count"] + outs[1]["count"] == 513 and outs[1]["start"] == outs[0]["count"]
    assert outs[0]["total"] == 513 and outs[1]["total"] == 513
    # both ranks report the slow rank's time
    assert abs(outs[0]["dt"] - outs[1]["dt"]) < 1e-9 and outs[0]["dt"] >= 0.09


def test_bench_launcher_spawns_ranks_and_aggregates():
    """`bench.py --gpus N` started as one process spawns N ranks (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set), shards a global batch with helib_amd.dist.shard and prints ONE line with
    n_gpus = N.  --dry-launch runs exactly that path over gloo with no engine call."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch",
                          "--global-batch", "513", "--steps", "3"], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["dry_launch"] is True and line["scaling"] == "strong"
    assert line["config"]["pairs_all_ranks"] == 513 and line["config"]["last_rank_start"] == 257
    # one key pair: rank 0's key material reaches every rank through Group.broadcast_words
    assert line["config"]["process_group_world_size"] == 2 and line["config"]["key_material_bytes_broadcast"] == 4096 * 8
    assert line["config"]["key_material_same_on_all_ranks"] is True
    # weak scaling: every rank keeps --batch pairs
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch",
                          "--batch", "64", "--steps", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["pairs_all_ranks"] == 128
    # BASELINE configs[3] as the driver's 8-GPU run would start it: the CKKS chain, 512 pairs split over the ranks
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "ckks65536",
                          "--global-batch", "512", "--dry-launch", "--steps", "2"], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["pairs_all_ranks"] == 512
    assert line["config"]["batch_this_rank"] == 256 and line["config"]["workload"] == "ckks65536"


def test_bench_refuses_a_gpus_flag_that_disagrees_with_the_world():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-launch"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stderr + out.stdout)


def test_rccl_selfcheck_reports_instead_of_raising_and_restores_the_environment():
    """helib_amd.dist.rccl_selfcheck (the N = 1 benchmark line's `config.rccl_selfcheck`): on a box without a GPU the
    nccl backend cannot come up -- the function must say so in its result, leave no process group behind and put
    MASTER_ADDR / MASTER_PORT / WORLD_SIZE back as they were."""
    import numpy as np
    from helib_amd import dist as hdist
    import torch.distributed as td
    before = {k: os.environ.get(k) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = hdist.rccl_selfcheck(np.arange(1024, dtype=np.uint64), None)
    assert isinstance(out, dict) and "ok" in out and "torch" in out
    import torch
    if not torch.cuda.is_available():
        assert out["ok"] is False and out.get("error")
    assert not td.is_initialized()
    assert {k: os.environ.get(k) for k in before} == before


def test_launcher_keeps_a_log_per_rank(tmp_path):
    """bench.py --gpus 2 --dry-launch: rank 1's output goes to rank1.log in HX_RANK_LOG_DIR instead of /dev/null, and the
    N > 1 line says what it leaves out."""
    env = dict(os.environ, HX_RANK_LOG_DIR=str(tmp_path))
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch", "--steps", "2",
                        "--batch", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["process_group_world_size"] == 2
    assert (tmp_path / "rank1.log").exists()


def test_launcher_watchdog_stops_the_other_ranks_when_one_dies():
    """A rank > 0 that dies during setup used to leave rank 0 inside its first collective until the backend's own
    timeout (ten minutes of an 8-GPU lease with RCCL; half an hour with gloo): launch_ranks waited on rank 0 first.
    The launcher now polls every rank, ends the others when the first one fails and prints that rank's log."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HX_TEST_FAIL_RANK"] = "1"
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch", "--steps", "2"],
                         env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode != 0
    assert time.time() - t0 < 120
    assert "rank 1 failed first" in out.stderr and "rank dies during setup" in out.stderr, out.stderr[-2000:]
    # ... and rank 0 failing is reported as rank 0's failure
    env["HX_TEST_FAIL_RANK"] = "0"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch", "--steps", "2"],
                         env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode != 0 and "rank 0 failed first" in out.stderr
