"""`bench.py --gpus N` starts its N ranks itself (the reference's analogue is the voxel pool of
mf.py:978-1009).  CPU-only: gloo backend and MFX_BENCH_STUB=1, i.e. the launcher, the rendezvous, the dictionary
broadcast, the shard sizes and the one-line contract are exercised, no kernel runs (the line says so)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(MFX_BENCH_STUB="1", MFX_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    env.update(kw)
    return env


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_gpus2_launches_two_ranks_and_prints_one_line():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--voxels", "1001",
                        "--scaling", "strong"], env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout
    r = lines[0]
    assert r["n_gpus"] == 2 and r["config"]["ranks_in_group"] == 2
    assert r["scaling"] == "strong" and r["steps"] == 3 and r["warmup"] == 1
    assert sorted(r["config"]["voxels_per_rank"]) == [500, 501] and r["config"]["global_voxels"] == 1001
    assert r["config"]["table_atoms_after_broadcast"] == 64      # rank 1 got the dictionary over the process group
    assert r["data"] == "stub" and r["metric"].startswith("STUB")


def test_weak_scaling_gives_every_rank_its_own_shard():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--voxels", "300"],
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    r = _json_lines(p.stdout)[0]
    assert r["scaling"] == "weak" and r["config"]["voxels_per_rank"] == [300, 300] and r["config"]["global_voxels"] == 600


def test_single_gpu_path_does_not_spawn():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "1", "--warmup", "0", "--voxels", "10"],
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    r = _json_lines(p.stdout)[0]
    assert r["n_gpus"] == 1 and r["config"]["ranks_in_group"] == 1


def test_world_size_mismatch_exits_nonzero():
    # as a launcher would start it, but with --gpus that does not match the ranks: no line, exit code != 0
    p = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert not _json_lines(p.stdout)
    assert "WORLD_SIZE" in p.stderr
