"""Real multi-rank runs (BASELINE configs[4]): two processes, two GPUs, RCCL with a 2-rank communicator.  The reference is one process
(/root/reference/experiment/exp_mapping.cpp:18-59); sharding + the all-gather merge are this repo's.  Collected everywhere, SKIPPED unless the box has
>= 2 HIP devices (the round's GPU box has one: tests/test_gpu_sharding.py emulates ranks with contexts there, tests/test_sharding_gloo.py covers the
N > 1 host logic on CPU).  On an 8-GPU node these are the first N > 1 executions of ncclCommInitRank / the padded in-place all-gather / the id hand-off."""
import json
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "semantic_slam_mapping_amd", "host")


def hip_device_count():
    """hipGetDeviceCount in a CHILD process (this process stays free to start ranks; no torch import)"""
    code = ("import ctypes\n"
            "try:\n"
            "    h = ctypes.CDLL('libamdhip64.so'); n = ctypes.c_int(0)\n"
            "    print(n.value if h.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0)\n"
            "except OSError:\n"
            "    print(0)\n")
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120,
                           env=dict(os.environ, LD_LIBRARY_PATH=os.environ.get("LD_LIBRARY_PATH", "") + ":/opt/rocm/lib"))
        return int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else 0
    except (subprocess.TimeoutExpired, ValueError):
        return 0


_NDEV = None


def need_two_gpus():
    global _NDEV
    if _NDEV is None:
        _NDEV = hip_device_count() if os.path.exists("/dev/kfd") else 0
    if _NDEV < 2:
        pytest.skip(f"needs >= 2 HIP devices (found {_NDEV}): the multi-rank RCCL path runs on the driver's multi-GPU node")


@pytest.mark.gpu
def test_bench_two_ranks_merge_verified():
    """`bench.py --gpus 2`: two ranks of 60 frames each; every rank must export the same merged map (CRC all-gather) and rank 0's own rebuild of the
    whole 120-frame stream must equal it byte for byte"""
    need_two_gpus()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "60", "--batch", "30", "--steps", "1", "--warmup", "0", "--no-cpu"],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, SSM_RANKS_TIMEOUT="600", SSM_BENCH_H2D="0"))
    print(r.stdout[-3000:], r.stderr[-3000:])
    assert r.returncode == 0
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["frames_all_gpus"] == 120
    assert line["merge_verified"] is True
    m = line["merge"]
    assert m["equals_single_gpu_map"] is True and len(m["voxels_per_rank"]) == 2 and all(v > 0 for v in m["voxels_per_rank"])
    assert len(line["per_rank"]["frames_per_s"]) == 2 and all(v > 0 for v in line["per_rank"]["frames_per_s"]) and line["allgather_ms_per_step"] > 0


@pytest.mark.gpu
def test_bench_two_ranks_strong_scaling_split():
    """--total-frames: ONE 90-frame stream cut into two contiguous blocks with the matcher halo (configs[4]'s partitioning)"""
    need_two_gpus()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--total-frames", "90", "--batch", "30", "--steps", "1", "--warmup", "0", "--no-cpu"],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, SSM_RANKS_TIMEOUT="600", SSM_BENCH_H2D="0"))
    print(r.stdout[-3000:], r.stderr[-3000:])
    assert r.returncode == 0
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["scaling"] == "strong" and line["merge_verified"] is True and line["merge"]["equals_single_gpu_map"] is True
    assert line["config"]["matcher_halo_frames"] > 0


@pytest.mark.gpu
def test_exp_mapping_two_ranks_equal_maps(tmp_path):
    """`exp_mapping --ranks 2` (C++ host: fork before HIP, id through a shared directory, ssm_voxel_allgather): both ranks must report the same merged map
    (voxel count + FNV) and it must equal the map of the same frames run as ONE rank"""
    need_two_gpus()
    prm = tmp_path / "p.txt"
    base = open(os.path.join(HOST, "parameters_test.txt")).read()
    out = {}
    for n in (2, 1):
        prm.write_text(base.replace("map_output=/tmp/ssm_test_map.pcd", f"map_output={tmp_path}/merged{n}.pcd") + "\nforce_rank_path=1\n")
        r = subprocess.run([os.path.join(HOST, "exp_mapping"), str(prm), "--ranks", str(n)], capture_output=True, text=True, timeout=600)
        print(r.stdout[-2000:], r.stderr[-2000:])
        assert r.returncode == 0
        stats = []
        for k in range(n):
            line = [l for l in r.stdout.splitlines() if l.startswith(f"rank {k}/{n} ")][-1].split()
            stats.append(dict(zip(line[2::2], line[3::2])))
        out[n] = stats
    a, b = out[2]
    assert a["map_fnv"] == b["map_fnv"] and a["merged_voxels"] == b["merged_voxels"]
    assert a["frames"] != b["frames"] and int(b["halo"]) > 0
    assert a["map_fnv"] == out[1][0]["map_fnv"] and a["merged_voxels"] == out[1][0]["merged_voxels"]
    assert os.path.getsize(tmp_path / "merged2.pcd") == os.path.getsize(tmp_path / "merged1.pcd") > 1000


def test_spawn_ranks_watchdog_stops_hung_ranks(tmp_path):
    """bench.py's launcher: ranks that never finish are stopped by PID after SSM_RANKS_TIMEOUT seconds and the run fails with 124 (CPU test: the 'ranks' are
    this interpreter sleeping -- spawn_ranks starts sys.argv again with RANK set)"""
    script = tmp_path / "hang.py"
    script.write_text("import os, sys, time\n"
                      f"sys.path.insert(0, {ROOT!r})\n"
                      "import bench\n"
                      "if 'RANK' in os.environ:\n"
                      "    time.sleep(600)\n"
                      "bench.__file__ = os.path.abspath(__file__)\n"
                      "sys.exit(bench.spawn_ranks(2))\n")
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=120, env=dict(os.environ, SSM_RANKS_TIMEOUT="2"))
    assert r.returncode == 124, (r.returncode, r.stderr[-500:])
    assert "still running" in r.stderr and time.time() - t0 < 60
