"""bench.py --gpus N must start N ranks by itself.  CPU-side: the launcher function is driven with a
gloo probe script in place of the GPU step, and the argument/launcher mismatch guard is checked before any GPU call."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _last_json(text):
    for line in reversed(text.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError("no JSON line in:\n" + text)


def test_spawn_ranks_starts_n_ranks_on_loopback():
    import bench
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    rc, out = bench.spawn_ranks(2, os.path.join(ROOT, "tests", "_rank_probe.py"), [], env=env, capture=True)
    assert rc == 0, out
    got = _last_json(out)
    assert got["n_gpus"] == 2 and got["sum"] == 3.0 and sorted(got["local_ranks"]) == [0, 1] and got["master"] == "127.0.0.1"
    # the diagnostics an N > 1 bench line carries: per-rank step times, all-reduce exposure, overlap probe, launch form
    d = got["dist"]
    assert d["world_size"] == 2 and d["backend"] == "gloo" and d["per_rank_ms"] == [10.0, 11.0] and d["finish_wait_ms"] == [0.0, 0.5]
    assert d["nt_launch_form"]["chosen"] in ("persistent", "per_tile") and "side_overlap_ratio" in d and d["tn_target_wgs"] in (128, 252)
    assert d["buckets"] >= 1


def test_bench_refuses_raised_hw_queues_with_ranks_sharing_a_gpu():
    """several ranks per card + GPU_MAX_HW_QUEUES hung in round 2; bench.py must exit non-zero before touching the GPU
    (here: 2 ranks, at most one - in this container zero - visible devices)."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", GPU_MAX_HW_QUEUES="8", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GPU_MAX_HW_QUEUES" in (r.stderr + r.stdout) and "refusing" in (r.stderr + r.stdout)


def test_bench_refuses_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing to mislabel" in (r.stderr + r.stdout)
