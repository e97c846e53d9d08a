#!/usr/bin/env python3
"""bench.py -- frames/sec of the per-frame semantic-mapping front end on MI355X (BASELINE.json metric).

A step = one pass of the hot path over one batch = the configs[1] stream: 1000 synthetic 640x480 RGB-D frames with
precomputed 12-class masks, ORB 1000 kp/frame, match against the 5 preceding frames, moving mask, gated
back-projection with the frame pose, voxel fusion (leaf 0.1 m) and the export of the fused map.  Inputs are generated
on the device before the timed region (resident in HBM).

N>1 (`--gpus N`): one process per GPU.  Run as `python bench.py --gpus N` this process starts the N ranks itself
(children, started before anything touches the GPU; nothing is exec'd from a process that has).  Under `python -m torch.distributed.run`
the ranks exist already (RANK / WORLD_SIZE in the environment).  Rank r owns the contiguous block [r F, (r+1) F) of ONE N F-frame
stream (weak scaling, F = 1000) -- or its block of a `--total-frames T` stream (strong scaling: BASELINE configs[4] is
`--gpus 8 --total-frames 10000`).  A rank first runs the tracker_ref_frames frames in front of its block through ORB only (the matcher
halo), so that its match tables are those of the single-GPU run (src/track.cpp:150-152), fuses its frames into its own voxel map,
and the maps are merged inside the step by ssm_voxel_allgather: ONE RCCL all-gather behind the C ABI, no torch on the data path
(torch.distributed only ships the ncclUniqueId and provides the barrier / max of the timing contract).

Prints ONE JSON line on rank 0.  cpu_baseline = the CPU oracle (oracle/, kind "port") timed on a bounded sample of
the same stream on the host cores of this box (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 640, 480
CAM = (318.6, 255.3, 517.3, 516.5, 1000.0)
SEED = 0x5EED0000
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
# VALU issue ceilings, measured per instruction class (profiles/r02_valu_rate.md, scripts/ubench/valu_rate.hip): everything these integer kernels are made
# of (bcnt, min / max / med3, perm, alignbyte, sad, dot, pk_*, mul, cmp, cndmask, DPP, cvt, f64) issues one wave64 instruction per 4 cycles per SIMD, and so
# does ANY mixed stream (xor + bcnt alternating: 8.0 cycles per pair): 1024 SIMDs x 16 lanes per clock x 2.4 GHz.  Only pure runs of mov / and / or / xor /
# add / sub / lshr / ashr / f32 add, sub, mul reach 2 cycles (SURVEY.md s.8d's 78.6 T); reported beside it as frac_of_simple_op_peak.
VALU_LANEOPS_PEAK = 39.3e12
VALU_SIMPLE_OP_PEAK = 78.6e12


def algorithmic_bytes(stage, P, nkp):
    """ALGORITHMIC HBM bytes per FRAME of each stage, P = emitted points of the frame"""
    px, pyr = W * H, 950532
    return {
        "gray": 3 * px + px,                       # R bgr + W gray                               1.229 MB
        "pyramid": 926546 + 643332,                # each level from the previous                 1.570 MB
        "fast": pyr + nkp * 8,                     # R pyramid (+ candidate list)                 0.951 MB
        "octree": nkp * 8 * 2,                     # candidate list re-read (tiny)
        "blur": 2 * pyr,                           # R + W                                        1.901 MB
        "describe": nkp * (32 + 28 + 12),          # descriptors + keypoints + 3-D positions      0.072 MB
        "mask": 3 * px + px,                       # R semantic + W mask (1.843 MB with the dilate passes in LDS)
        "backproject": (2 + 3 + 3 + 1) * px + 32 * P,   # R depth,rgb,sem,mask + W points         2.765 MB + 32 P
        "voxel_insert": 32 * P + 32 * P,           # R points + table update                      64 P
        "map_fuse": (2 + 3 + 3) * px,              # fused K10+K11+K12: R depth, rgb, semantic once; points stay on chip
    }.get(stage, 0)


STAGE_KERNELS = {"gray": ["gray_kernel<true>"], "pyramid": ["resize4_kernel", "resize_kernel"], "fast": ["fast_kernel", "fast_need_kernel", "fast_retry_kernel"], "octree": ["octree_kernel"],
                 "blur": ["blur_kernel", "blur_mfma_kernel"], "describe": ["kp_prepare_kernel", "orient_kernel", "angle_kernel", "brief_kernel"],
                 "match": ["match_expand_kernel", "match_mfma_kernel", "match_compact_kernel"],
                 "map_fuse": ["class_bits_kernel", "vdilate_bits_kernel", "map_stream_kernel", "map_stream2_kernel"],
                 "segnet": ["segnet_prep_kernel", "conv3x3_", "unpool2x2_kernel", "label_color_kernel", "argmax_kernel"]}


def traffic_profile(suffix, source):
    """(path, None) of the newest committed profiles/rNN_<suffix> if it was collected for the CURRENT semantic_slam_mapping_amd/csrc/<source> (the file records the
    sha256 of every kernel source it was made from: scripts/pmc_traffic.py), else (None, why) -- a kernel change without a re-collection must not keep the old ratio"""
    import hashlib
    path = latest_profile(suffix)
    if not path:
        return None, "no profiles/rNN_%s" % suffix
    csrc = os.path.join(ROOT, "semantic_slam_mapping_amd", "csrc")
    # a kernel file and the parts it includes (kernels_sgbm.hip: sgbm_*.inc)
    files = [source] + sorted(f for f in os.listdir(csrc) if f.endswith(".inc") and f.startswith(source.replace("kernels_", "").replace(".hip", "") + "_"))
    try:
        rec = json.load(open(path)).get("sources_sha256", {})
        for f in files:
            if rec.get(f) != hashlib.sha256(open(os.path.join(csrc, f), "rb").read()).hexdigest():
                return None, "%s is stale: collected for another %s (re-collect with scripts/collect_profiles.sh + scripts/make_profiles.sh)" % (os.path.relpath(path, ROOT), f)
    except Exception as e:
        return None, "%s: %r" % (os.path.relpath(path, ROOT), e)
    return path, None


TRAFFIC_SOURCE = {"traffic.json": "kernels_map.hip", "stereo_traffic.json": "kernels_sgbm.hip", "segnet_traffic.json": "kernels_segnet.hip"}


def measured_traffic(stage, frames_per_launch, suffix="traffic.json"):
    """HBM bytes per launch of a stage from the committed rocprofv3 PMC passes (profiles/rNN_traffic.json: FETCH_SIZE + WRITE_SIZE, separate passes;
    FETCH_SIZE x 2 on gfx950 for the kernels that read 16 B per lane, raw for the others: scripts/pmc_traffic.py), or None"""
    try:
        path, why = traffic_profile(suffix, TRAFFIC_SOURCE[suffix])
        if path is None:
            return None
        k = json.load(open(path))["kernels"]
        return round(sum(v.get("total_bytes_per_frame", v["total_bytes_per_frame_fetch_x2"]) for name, v in k.items() if any(n in name for n in STAGE_KERNELS[stage])) * frames_per_launch)      # (`in`: the fp16 kernels' names stay mangled in rocprofv3's output)
    except Exception:
        return None


def measured_valu(stage, us_per_frame):
    """VALU issue rate of a stage: wave-instructions per frame from the newest committed rocprofv3 SQ pass (profiles/rNN_sq_counters.json)
    x 64 lanes / the stage time measured now, against the measured integer issue ceiling (VALU_LANEOPS_PEAK), or None"""
    try:
        k = json.load(open(latest_profile("sq_counters.json")))["kernels"]
        insts = sum(v["valu_wave_insts_per_frame"] for name, v in k.items() if any(name.startswith(n) for n in STAGE_KERNELS[stage]))
        ach = insts * 64 / (us_per_frame * 1e-6) / 1e12
        return {"achieved": round(ach, 2), "peak": VALU_LANEOPS_PEAK / 1e12, "unit": "Tlaneop/s", "frac": round(ach / (VALU_LANEOPS_PEAK / 1e12), 3),
                "frac_of_simple_op_peak": round(ach / (VALU_SIMPLE_OP_PEAK / 1e12), 3), "wave_insts_per_frame": round(insts),
                "peak_basis": "4 cycles per wave64 instruction per SIMD: measured for every instruction class of these kernels and for mixed streams (profiles/r02_valu_rate.md)"}
    except Exception:
        return None


def stereo_sequence(n, w, h, seed, planes=((12, None), (30, (0.4, 0.9, 0.2, 0.5)), (60, (0.5, 0.95, 0.6, 0.8))), flow=(3, 1), noise=4):
    """synthetic rectified stereo SEQUENCE (configs[3] shape): a smooth random texture seen through fronto-parallel planes of constant disparity; the
    camera pans by `flow` pixels per frame, so consecutive frames track (quad matcher) and every pair has a dense disparity (SGBM)"""
    import numpy as np
    rng = np.random.default_rng(seed)
    H, W = h + 64, w + 160 + 256
    tex = rng.integers(0, 256, (H, W)).astype(np.float32)
    k = np.array([1, 4, 6, 4, 1], np.float32) / 16.0
    for ax in (0, 1):                                          # separable 5-tap smoothing (wrap-around borders: the canvas is cropped anyway)
        tex = sum(k[i] * np.roll(tex, i - 2, axis=ax) for i in range(5))
    tex = ((tex - tex.min()) / (tex.max() - tex.min()) * 255).astype(np.uint8)
    dmap = np.zeros((h, w), np.int32)
    for d, box in planes:
        if box is None:
            dmap[:] = d
        else:
            y0, y1, x0, x1 = int(box[0] * h), int(box[1] * h), int(box[2] * w), int(box[3] * w)
            dmap[y0:y1, x0:x1] = d
    L = np.empty((n, h, w), np.uint8); R = np.empty((n, h, w), np.uint8)
    for f in range(n):
        ox, oy = 80 + (f * flow[0]) % 256, (f * flow[1]) % 64
        win = tex[oy:oy + h, :]
        R[f] = win[:, ox:ox + w]
        L[f] = np.take_along_axis(win, np.arange(w)[None, :] - dmap + ox, axis=1)
        if noise:
            L[f] = np.clip(L[f].astype(np.int32) + rng.integers(-noise, noise + 1, (h, w)), 0, 255).astype(np.uint8)
    return L, R


KITTI = dict(baseline=0.532331858, cu=607.1928, cv=185.2157, f=718.856, roix=20.0, roiy=5.0, roiz=40.0, scale=1000.0)   # parameters.txt:37-63


def stereo_main(args):
    """configs[3]: KITTI-geometry stereo (1241 x 376, synthetic rectified sequence) through the batched device-resident path: a step = one
    ssm_stereo_seq_process call over F frame pairs resident in HBM -- per frame the quad matcher against the previous frame (GFTT + 4 x pyramidal LK +
    filteringTracks), SGBM depth (80 disparities, SAD 11) + the ROI depth conversion, and the stereo VO (200 RANSAC hypotheses) on the quad matches.
    The roofline object is for the SGBM kernels (algorithmic bytes = the u16 cost volume written once and read once per scan direction)."""
    import numpy as np
    import torch
    import semantic_slam_mapping_amd as ssm
    from semantic_slam_mapping_amd.api import GlibcRand
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    Wd, Hd, D, ITERS = 1241, 376, 80, 200
    F = args.frames if args.frames != 1000 else 256           # frame pairs per step, resident in HBM (replicas on every rank: the sequence of a rank is its own)
    B = max(1, args.stereo_batch)
    ctx = ssm.Context(local_rank, width=640, height=480, max_batch=B)      # the stereo path launches min(max_batch, 128) frame pairs at a time (ssm_stereo_batch)
    L, R = stereo_sequence(F, Wd, Hd, 100 + rank)
    dl = ctx.dev_alloc(L.nbytes); dr = ctx.dev_alloc(R.nbytes); ds = ctx.dev_alloc(F * ITERS * 3 * 4)
    ctx.h2d(dl, L); ctx.h2d(dr, R); ctx.h2d(ds, GlibcRand(0).draws(F * ITERS * 3))
    vo = (KITTI["f"], KITTI["cu"], KITTI["cv"], KITTI["baseline"], 2.0, True)

    def step():
        out = ctx.stereo_seq_process(dl, dr, F, Wd, Hd, vo=vo, ransac_iters=ITERS, rand_stream_dev=ds, **KITTI)
        ctx.sync()
        return out

    def fence():
        torch.cuda.synchronize(); ctx.sync()
        if world > 1: dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.set_profiling(2 if args.serial_only else 1)
    fence()
    t0 = time.perf_counter()
    stage_ovl = {}
    for _ in range(args.steps):
        out = step()
        for k, (ms, ln) in ctx.stage_times().items():
            a = stage_ovl.setdefault(k, [0.0, 0]); a[0] += ms; a[1] += ln
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=torch.device("cuda", local_rank)); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
    # undisturbed stage durations: the same steps with SGBM on the chain's stream (profiling mode 2)
    stage_acc = {k: list(v) for k, v in stage_ovl.items()} if args.serial_only else {}
    ctx.set_profiling(2)
    for _ in range(0 if args.serial_only else max(1, min(args.steps, 3))):
        step()
        for k, (ms, ln) in ctx.stage_times().items():
            a = stage_acc.setdefault(k, [0.0, 0]); a[0] += ms; a[1] += ln
    ctx.set_profiling(0)
    res = ctx.stereo_seq_fetch(out, F, Wd, Hd, 1 | 4)
    if rank != 0:
        ctx.close()
        return None
    nser = args.steps if args.serial_only else max(1, min(args.steps, 3))
    frames = world * F * args.steps
    sg_ms_launch = stage_acc["sgbm"][0] / max(stage_acc["sgbm"][1], 1)
    fpl = F * nser / max(stage_acc["sgbm"][1], 1)            # frame pairs per SGBM launch group
    vol = (Wd - D) * Hd * D
    alg = 6 * vol * 2 * fpl                                 # u16 cost volume: written once, read once by each of the 5 scan directions
    ach = alg / (sg_ms_launch * 1e-3) / 1e9
    cpu = None
    if world == 1 and not args.no_cpu:
        from oracle.binding import Oracle                   # the checker, here only as the timed CPU baseline
        orc = Oracle()
        t1 = time.perf_counter(); ref = orc.sgbm(L[1], R[1], orc.sgbm_params()); orc.disparity_to_depth(ref, **KITTI); tc = time.perf_counter() - t1
        t1 = time.perf_counter(); qm = orc.quad_track(L[1], R[1], L[0], R[0]); tq = time.perf_counter() - t1
        cpu = {"value": round(1.0 / (tc + tq), 3), "unit": "frame pairs/s", "cores": 1, "kind": "port",
               "sample": "one frame pair of the same sequence: oracle/ quad matcher + SGBM + depth conversion (C, 1 thread); the VO is below a millisecond",
               "ms_per_frame": {"sgbm": round(tc * 1e3, 2), "quad_track": round(tq * 1e3, 2)},
               "quad_matches_equal_gpu": bool(int(res["nquad"][1]) == len(qm) and res["quad"][1, :len(qm)].tobytes() == qm.tobytes())}
    traffic = None
    try:        # HBM bytes of the SGBM kernels per launch from the committed PMC passes (profiles/rNN_stereo_traffic.json: FETCH_SIZE x 2 + WRITE_SIZE)
        tpath, twhy = traffic_profile("stereo_traffic.json", "kernels_sgbm.hip")
        kk = json.load(open(tpath))["kernels"]
        traffic = round(sum(v["total_bytes_per_frame"] for name, v in kk.items() if name.startswith("sgbm_")) * fpl)
    except Exception:
        pass
    nq = res["nquad"][1:]
    line = {"metric": "frame pairs/sec stereo front end (quad matcher + SGBM depth + stereo VO), 1241x376 rectified stereo", "value": round(frames / dt, 2), "unit": "frame pairs/s",
            "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "i16", "data": "synthetic",
            "config": {"workload": "configs[3] stages: synthetic rectified stereo sequence 1241x376 (seeded textured planes, 3 px/frame pan), per frame pair: quad matcher "
                                   "against the previous frame (GFTT + 4x LK), SGBM depth (80 disparities, SAD 11) + ROI depth conversion, stereo VO (200 RANSAC "
                                   "hypotheses); %d frame pairs resident in HBM per step, ssm_stereo_seq_process" % F,
                       "frame_pairs_per_gpu": F, "frame_pairs_per_launch": B, "parallelism": "replicas" if world > 1 else "single GPU"},
            "per_frame": {"quad_matches": round(float(nq.mean()), 1), "vo_success_rate": round(float(res["vo_result"][1:, 1].mean()), 3)},
            "roofline": {"bound": "hbm", "kernel": "sgbm (prefilter .. wta, median, speckle, depth: all kernels of the SGBM stage)", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_profile("stereo_traffic.json", "kernels_sgbm.hip")[1] if traffic is None else os.path.relpath(latest_profile("stereo_traffic.json"), ROOT) + " (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, not measured in this run)",
                         "algorithmic_bytes_per_launch": round(alg),
                         "stages_ms_per_frame": {k: round(v[0] / (F * nser), 4) for k, v in stage_acc.items()},
                         "stages_ms_per_frame_overlapped": {k: round(v[0] / (F * args.steps), 4) for k, v in stage_ovl.items()}},
            "cpu_baseline": cpu}
    for p in (dl, dr, ds):
        ctx.dev_free(p)
    ctx.close()
    return line


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children of this process, which has not imported torch or
    touched HIP (a process that has initialised the GPU must neither exec nor fork workers).  Rank 0's stdout (the JSON line) passes
    through; any failing rank fails the run and the others are stopped by PID."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # watchdog: a collective that hangs with every rank alive would otherwise wait for the driver's kill.  SSM_RANKS_TIMEOUT seconds of wall clock for the
    # whole job (default 900), then every rank is stopped by PID (terminate, then kill) and the run fails with 124 -- never a re-exec
    limit = float(os.environ.get("SSM_RANKS_TIMEOUT", "900"))
    t_end = time.monotonic() + limit
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            try:
                code = p.wait(timeout=0.5)
            except subprocess.TimeoutExpired:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:
                    q.terminate()
        if alive and time.monotonic() > t_end:
            sys.stderr.write("bench.py: %d of %d ranks still running after %.0f s (SSM_RANKS_TIMEOUT): stopping them\n" % (len(alive), n, limit))
            for q in alive:
                q.terminate()
            t_kill = time.monotonic() + 10
            for q in alive:
                try:
                    q.wait(timeout=max(0.1, t_kill - time.monotonic()))
                except subprocess.TimeoutExpired:
                    q.kill(); q.wait()
            return 124
    return rc


def latest_profile(suffix):
    """newest committed profiles/rNN_<suffix> (the counter passes are re-collected whenever a kernel changes)"""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    return c[-1] if c else None


def granted_cores():
    """host cores this process can really use: the affinity mask, capped by the container's CPU quota (cgroup v2 cpu.max = "<quota> <period>")"""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            cores = max(1, min(cores, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_all_cores(orc, frames_per_thread, leaf):
    """the oracle pipeline on every host core, frame-sharded like the GPU path (each thread runs its own contiguous
    block of the stream through oracle/pipeline.c; ctypes releases the GIL during the call)"""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    cores = granted_cores()
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        list(ex.map(lambda i: orc.pipeline(i * frames_per_thread, frames_per_thread, nfeatures=1000, leaf=np.float32(leaf)), range(cores)))
    dt = time.perf_counter() - t0
    model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip(); break
    except OSError:
        pass
    return {"value": round(cores * frames_per_thread / dt, 3), "unit": "frames/s", "cores": cores, "cpu": model,
            "sample": f"{cores} threads x {frames_per_thread} consecutive frames each (blocks of the same stream, synth included: {dt:.1f} s wall)"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=1000, help="frames per GPU per step (configs[1]: 1000); weak scaling")
    ap.add_argument("--total-frames", type=int, default=0, help="strong scaling: ONE stream of this many frames split into contiguous blocks over the GPUs "
                    "(BASELINE configs[4]: --gpus 8 --total-frames 10000)")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("SSM_BATCH", "250")), help="frames per batched launch (a tuning knob: 5 MB of workspace per frame; "
                    "with three ORB chains and the map stream the rate is flat (+-2 %%) from 200 to 500 frames and ~3 %% lower at 125)")
    ap.add_argument("--leaf", type=float, default=0.1)
    ap.add_argument("--cpu-frames", type=int, default=int(os.environ.get("SSM_CPU_FRAMES", "150")))
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--first-frame", type=int, default=0, help="stream offset of rank 0 (testing aid: > 0 makes rank 0 run a matcher halo too, like ranks > 0 do)")
    ap.add_argument("--no-halo", action="store_true", help="N>1: skip the matcher halo (the first tracker_ref_frames frames of a block then lack their references)")
    ap.add_argument("--segnet", action="store_true", help="configs[2]: labels from the on-GPU SegNet (fp16 MFMA) instead of precomputed masks")
    ap.add_argument("--serial-only", action="store_true", help="profiling aid: run ONLY serialised steps (every stage on one stream, no PCIe pass), so that a "
                    "rocprofv3 --stats run of this command has the same per-kernel average as roofline's launch duration; `value` is then the serialised rate")
    ap.add_argument("--stereo-batch", type=int, default=128, help="configs[3]: frame pairs per launch of the batched stereo path (0.23 GB of SGBM workspace each, two workspaces; "
                    "round 5: 128 instead of 64 -- the sweep's strips of more frames in flight overlap better: 4.93 k -> 5.18 k pairs/s)")
    ap.add_argument("--solve-poses", action="store_true", help="also time the closed pose loop: ORB + match tables -> ssm_tracker_run (Tracker::updateFrame for every frame: "
                    "the serial PnP chain) -> the solved poses into the map stage; reported as `solve_poses` beside `value` (whose poses are the stream's)")
    ap.add_argument("--pose-frames", type=int, default=200, help="frames of the --solve-poses leg")
    ap.add_argument("--pose-stream", default="rigid", choices=["rigid", "synthetic"], help="--solve-poses: 'rigid' = frame 0 of the stream seen by a panning camera (a "
                    "fronto-parallel plane at 2 m, 2 x 1 px per frame: every frame tracks); 'synthetic' = the configs[1] stream itself (its depth pattern does not move "
                    "with the texture, so PnP loses track often: exercises LOST / lostRecover)")
    ap.add_argument("--pnp-device", type=int, default=0, help="--solve-poses: 1 = the pose chain on the GPU (a cluster of eight blocks; one block per chain with --pose-threads > 1), 0 = on one host core; same bits")
    ap.add_argument("--pose-threads", type=int, default=1, help="--solve-poses with the rigid stream: its independent 20-frame sequences are tracked by this many trackers from as many host threads (device chains on their own streams: one CU each)")
    ap.add_argument("--no-verify-whole", dest="verify_whole", action="store_false", help="N>1: skip rank 0's rebuild of the whole-stream map that the merged "
                    "map is compared with byte for byte (the cross-rank CRC check always runs)")
    ap.add_argument("--stereo", action="store_true", help="configs[3]: the stereo stages on 1241x376 pairs (quad matcher, SGBM depth, stereo VO)")
    ap.add_argument("--no-other-configs", dest="other_configs", action="store_false", help="default run (N = 1, no mode flag): skip the short configs[2] / configs[3] / "
                    "closed-pose-loop legs that are reported as `other_configs` beside the configs[1] headline")
    ap.add_argument("--other-scale", type=float, default=1.0, help="size factor of the `other_configs` legs (tests: 0.1 with a short headline; 1.0 = 256 SegNet frames, 256 stereo pairs, 400 pose frames)")
    args = ap.parse_args(argv)
    args.pose_cpu_sample = 0
    return args


def rgbd_main(args):
    """configs[1] (default), configs[2] (--segnet), configs[4] (--gpus N --total-frames T): returns the JSON line (rank 0) or None"""
    if args.solve_poses and args.pose_threads > 1:
        # HIP multiplexes streams onto 4 hardware queues by default and two chains that share a queue run one after the other (scripts/tracker_concurrency.py):
        # the runtime reads this before its first call, so it is set before torch is imported
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

    import numpy as np
    import torch
    import torch.distributed as dist
    import semantic_slam_mapping_amd as ssm
    from semantic_slam_mapping_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_merge = os.environ.get("SSM_FORCE_MERGE") == "1"      # run the RCCL all-gather + merge path with a 1-rank communicator (tests)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # ---- this rank's block of the stream
    if args.total_frames > 0:
        lo, hi = sharding.frame_block(args.total_frames, rank, world); scaling = "strong"
    else:
        lo, hi = rank * args.frames, (rank + 1) * args.frames; scaling = "weak"
    lo += args.first_frame; hi += args.first_frame
    F = hi - lo
    ctx = ssm.Context(local_rank, orb_features=1000, max_batch=args.batch, voxel_capacity_log2=20,
                      mapper_resolution=args.leaf, camera=CAM)
    R = ctx.R
    if world > 1 or force_merge:                      # the communicator of the data path lives behind the C ABI
        idt = torch.zeros(ssm.api.COMM_ID_BYTES, dtype=torch.uint8, device=dev)
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(ctx.comm_unique_id()), dtype=torch.uint8))
        if world > 1:
            dist.broadcast(idt, 0)
        ctx.comm_init_rank(world, rank, bytes(idt.cpu().numpy().tobytes()))
    # ---- inputs resident in HBM
    bgr = torch.empty(F * H * W * 3, dtype=torch.uint8, device=dev)
    dep = torch.empty(F * H * W, dtype=torch.int16, device=dev)
    sem = torch.empty(F * H * W * 3, dtype=torch.uint8, device=dev)
    pose = torch.empty(F * 16, dtype=torch.float64, device=dev)
    ctx.synth_frames_dev(SEED, lo, F, bgr.data_ptr(), dep.data_ptr(), sem.data_ptr(), pose.data_ptr())
    halo_lo, halo_hi = sharding.halo_block(lo, R) if not args.no_halo else (lo, lo)
    HN = halo_hi - halo_lo                            # the matcher halo: frames [lo - R, lo), ORB only
    if HN:
        hb = [torch.empty(HN * H * W * 3, dtype=torch.uint8, device=dev), torch.empty(HN * H * W, dtype=torch.int16, device=dev),
              torch.empty(HN * H * W * 3, dtype=torch.uint8, device=dev), torch.empty(HN * 16, dtype=torch.float64, device=dev)]
        ctx.synth_frames_dev(SEED, halo_lo, HN, *[t.data_ptr() for t in hb])
    ctx.sync()
    tab_cap = 1 << 20
    tab_buf = torch.empty(tab_cap * sharding.VOXEL_BYTES, dtype=torch.uint8, device=dev)

    stages = 0
    if args.segnet:
        from semantic_slam_mapping_amd import segnet_model
        for l, (wt, sc, sh) in enumerate(segnet_model.make_weights(1234)):
            ctx.segnet_set_layer(l, wt, sc, sh)
        stages = 1 | 2 | 4 | 8

    def step():
        ctx.map_clear()
        if HN:                                        # descriptors of the frames in front of the block (the previous rank owns them)
            ctx.seq_process(hb[0].data_ptr(), None, None, None, HN, stages=ssm.api.STAGE_ORB)
        out = ctx.seq_process(bgr.data_ptr(), dep.data_ptr(), None if args.segnet else sem.data_ptr(), pose.data_ptr(), F, continue_sequence=HN > 0, stages=stages)
        if world > 1 or force_merge:                  # merge the per-GPU voxel maps: ONE RCCL all-gather of the tables (C ABI)
            ctx.voxel_allgather()
        n_vox = ctx.map_export_table_dev(tab_buf.data_ptr(), tab_cap)      # sorted fused map (sync)
        return out, n_vox

    def fence():
        torch.cuda.synchronize()
        ctx.sync()                                    # the context streams are not torch's
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.leaf < 0.1:                                # a small leaf: millions of voxels -- the export buffer is sized from one untimed pass (the context map grows by itself)
        ctx.map_clear()
        ctx.seq_process(bgr.data_ptr(), dep.data_ptr(), None if args.segnet else sem.data_ptr(), pose.data_ptr(), F, stages=stages or ssm.api.STAGE_MAP)
        ctx.sync(); tab_cap = max(tab_cap, 2 * ctx.map_size() * max(world, 1))
        del tab_buf
        tab_buf = torch.empty(tab_cap * sharding.VOXEL_BYTES, dtype=torch.uint8, device=dev)
    for _ in range(args.warmup):
        step()
    # ---- timed region: K steps as a user runs them (map / SegNet stage on a second stream beside the ORB -> match chain);
    # hipEvents per stage are recorded here too, but these stage times overlap
    ctx.set_profiling(2 if args.serial_only else 1)
    fence()
    t0 = time.perf_counter()
    stage_ovl = {}
    for _ in range(args.steps):
        out, n_vox = step()                            # (ends with the map export: that waits for the MAP's stream; the ORB -> match chain of the step's last sub-batch runs on
                                                       # into the next step -- ssm_get_stage_times would wait for it, so the stage brackets are read once, behind the fence)
    fence()
    dt = time.perf_counter() - t0
    for k, (ms, ln) in ctx.stage_times().items():      # the last step's brackets, scaled to the K steps (the steps are identical)
        stage_ovl[k] = [ms * args.steps, ln * args.steps]
    # ---- kernel durations for the roofline: the same K steps with every stage serialised on one stream (profiling mode 2),
    # so that a stage's hipEvent bracket is that kernel alone (this is also what profiles/*_kernel_stats.md lists)
    ctx.set_profiling(2)
    stage_acc = {}
    if args.serial_only:                               # the timed region already ran serialised: every launch of the process is of that kind
        stage_acc = {k: list(v) for k, v in stage_ovl.items()}
    for _ in range(0 if args.serial_only else args.steps):
        step()
        for k, (ms, ln) in ctx.stage_times().items():
            a = stage_acc.setdefault(k, [0.0, 0])
            a[0] += ms; a[1] += ln
    fence()
    ctx.set_profiling(0)
    # ---- PCIe-inclusive rate (reported beside `value`, never as `value`): the same steps with the frames copied from pinned
    # host memory to the device inside the timed region (bgr + depth + labels + pose = 2.46 MB per frame), copy and compute overlapped
    h2d_fps = None
    if world == 1 and not args.segnet and not args.serial_only and os.environ.get("SSM_BENCH_H2D", "1") == "1":
        try:
            hbuf = [t.cpu().pin_memory() for t in (bgr, dep, sem, pose)]
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            # copies of sub-batch b + 1 run on a side stream under the kernels of sub-batch b (the frames of a step stay resident, so
            # every sub-batch has its own region of the device buffers)
            # the copies go to torch's DEFAULT stream (the context's streams are non-blocking ones: no implicit ordering with it).  A torch side stream here --
            # torch creates its pool of 32 streams at the first request -- left the process in a state in which the stereo leg that follows in the default run lost 6 %
            # (4.85 k instead of 5.15 k pairs/s: its four streams no longer on four hardware queues); round 5, scripts: bisected with the pass on / off
            cstream = torch.cuda.current_stream()
            def copy_chunk(a, b):
                for src, dst, per in zip(hbuf, (bgr, dep, sem, pose), (H * W * 3, H * W, H * W * 3, 16)):      # flat tensors: elements per frame
                    dst[a * per:b * per].copy_(src[a * per:b * per], non_blocking=True)
                ev = torch.cuda.Event(); ev.record(cstream)
                return ev
            for _ in range(max(1, min(args.steps, 3))):
                ctx.map_clear()
                cb = min(args.batch, 125)                   # copy granularity: small enough that only the first chunk's copy is exposed
                ev = copy_chunk(0, min(F, cb))
                for a in range(0, F, cb):
                    b = min(F, a + cb)
                    ev.synchronize()                       # host wait: the context stream is not torch's
                    if b < F:
                        ev = copy_chunk(b, min(F, b + cb))
                    ctx.seq_process(bgr[a * H * W * 3:].data_ptr(), dep[a * H * W:].data_ptr(), sem[a * H * W * 3:].data_ptr(), pose[a * 16:].data_ptr(), b - a, continue_sequence=a > 0, stages=stages)
                ctx.map_export_table_dev(tab_buf.data_ptr(), tab_cap)
            torch.cuda.synchronize()
            h2d_fps = F * max(1, min(args.steps, 3)) / (time.perf_counter() - t1)
            del hbuf
        except Exception:
            h2d_fps = None
    # ---- self-validation of the merge (N > 1, or the 1-rank communicator of SSM_FORCE_MERGE): one more, untimed step in which every rank reads its
    # local voxel count before the all-gather, then hashes the merged table it exports; the hashes must agree on every rank.  With --verify-whole
    # (default for strong scaling, where the stream is one) rank 0 also rebuilds the map of the WHOLE stream by itself (map stage only, block by
    # block through its own buffers) and compares the tables byte for byte: merged == single-GPU.
    res = ctx.seq_fetch(out, F)                  # the last timed / serialised step's outputs (the validation below overwrites the device buffers)
    merge_info = None
    if world > 1 or force_merge:
        ctx.map_clear()
        if HN:
            ctx.seq_process(hb[0].data_ptr(), None, None, None, HN, stages=ssm.api.STAGE_ORB)
        ctx.seq_process(bgr.data_ptr(), dep.data_ptr(), None if args.segnet else sem.data_ptr(), pose.data_ptr(), F, continue_sequence=HN > 0, stages=stages)
        ctx.sync()
        n_local = ctx.map_size()
        ctx.voxel_allgather()
        n_merged = ctx.map_export_table_dev(tab_buf.data_ptr(), tab_cap)
        merged_bytes = tab_buf[: n_merged * sharding.VOXEL_BYTES].cpu().numpy().tobytes()
        merge_info = sharding.merge_check(merged_bytes, n_merged, n_local, dist if world > 1 else None, dev)
        if args.verify_whole and not args.segnet:
            whole_ok = None
            if rank == 0:
                ctx.map_clear()
                blocks = [sharding.frame_block(args.total_frames, r, world) if args.total_frames > 0 else (r * args.frames, (r + 1) * args.frames) for r in range(world)]
                for blo, bhi in blocks:
                    blo += args.first_frame; bhi += args.first_frame
                    for a in range(blo, bhi, F):
                        nfr = min(F, bhi - a)
                        ctx.synth_frames_dev(SEED, a, nfr, bgr.data_ptr(), dep.data_ptr(), sem.data_ptr(), pose.data_ptr())
                        ctx.seq_process(bgr.data_ptr(), dep.data_ptr(), sem.data_ptr(), pose.data_ptr(), nfr, stages=ssm.api.STAGE_MAP)
                n_whole = ctx.map_export_table_dev(tab_buf.data_ptr(), tab_cap)
                whole_ok = n_whole == n_merged and tab_buf[: n_whole * sharding.VOXEL_BYTES].cpu().numpy().tobytes() == merged_bytes
                ctx.synth_frames_dev(SEED, lo, F, bgr.data_ptr(), dep.data_ptr(), sem.data_ptr(), pose.data_ptr())      # the resident buffers hold this rank's own block again
            merge_info["equals_single_gpu_map"] = whole_ok
            if rank == 0:
                merge_info["verified"] = bool(merge_info["verified"] and whole_ok)
    # ---- the closed pose loop (reported beside `value`): poses from the pipeline instead of the stream's ground truth
    solve_info = None
    if args.solve_poses and world == 1 and not args.segnet:
        PF = min(args.pose_frames, F)
        trk = ssm.Tracker(ctx, use_device=bool(args.pnp_device))
        if args.pose_stream == "rigid":                # overwrite the first PF frames of the resident stream with the panning view of frame 0
            b0 = bgr[: H * W * 3].view(H, W, 3).clone(); s0 = sem[: H * W * 3].view(H, W, 3).clone()
            for k in range(PF):
                bgr[k * H * W * 3:(k + 1) * H * W * 3] = torch.roll(b0, shifts=(k, 2 * k), dims=(0, 1)).reshape(-1)
                sem[k * H * W * 3:(k + 1) * H * W * 3] = torch.roll(s0, shifts=(k, 2 * k), dims=(0, 1)).reshape(-1)
            dep[: PF * H * W] = 2000
            torch.cuda.synchronize()
        extra = None; NT = 1
        for rep in range(2):                             # pass 0 warms (tracker scratch, first-use costs), pass 1 is reported
            t1 = time.perf_counter()
            ctx.map_clear()
            o2 = ctx.seq_process(bgr.data_ptr(), dep.data_ptr(), None, None, PF, stages=ssm.api.STAGE_ORB | ssm.api.STAGE_MATCH)
            ctx.sync(); t2 = time.perf_counter()
            if args.pose_stream == "rigid":
                # the reference's odometry chain is not stable on a planar scene (depth error feeds back through the reference poses: it diverges after ~25
                # frames, host class and bulk tracker alike), so the rigid stream is tracked as independent 20-frame sequences: views of the call's outputs at
                # a frame offset, the tracker reset in between (its first frame is then an initFirstFrame)
                from semantic_slam_mapping_amd._lib import SeqOutDev
                CH = 20
                def view(a0):
                    cap_, R_ = o2.cap, o2.R
                    return SeqOutDev(o2.kps + a0 * cap_ * 28, o2.desc + a0 * cap_ * 32, o2.pos3d + a0 * cap_ * 12, o2.nkp + a0 * 4, o2.matches + a0 * R_ * cap_ * 16,
                                     o2.nmatch + a0 * R_ * 4, o2.npoints + a0 * 4, cap_, R_)
                starts = list(range(0, PF, CH)); walked = {}
                def walk(tr, mine):
                    for a0 in mine:
                        tr.reset()
                        walked[a0] = tr.run(view(a0), min(CH, PF - a0))
                NT = max(1, min(args.pose_threads, len(starts)))
                if NT == 1:
                    walk(trk, starts)
                else:
                    # independent sequences side by side: a tracker per host thread, each chain (one block = one CU) on a stream of its own
                    import threading
                    if extra is None:
                        extra = [ssm.Tracker(ctx, use_device=bool(args.pnp_device), own_stream=True) for _ in range(NT)]
                    th = [threading.Thread(target=walk, args=(extra[k], starts[k::NT])) for k in range(NT)]
                    for t_ in th: t_.start()
                    for t_ in th: t_.join()
                    pose_stats = [e.stats() for e in extra]
                poses_s, info_s = np.concatenate([walked[a0][0] for a0 in starts]), np.concatenate([walked[a0][1] for a0 in starts])
            else:
                trk.reset()
                poses_s, info_s = trk.run(o2, PF)
            t3 = time.perf_counter()
            pdev = torch.from_numpy(np.ascontiguousarray(poses_s.transpose(0, 2, 1)).reshape(PF * 16)).to(dev)
            ctx.seq_process(bgr.data_ptr(), dep.data_ptr(), sem.data_ptr(), pdev.data_ptr(), PF, stages=ssm.api.STAGE_MAP)
            nv = ctx.map_export_table_dev(tab_buf.data_ptr(), tab_cap)
            t4 = time.perf_counter()
            dvf, hsf = trk.stats()
            if args.pose_stream == "rigid" and NT > 1:
                dvf, hsf = sum(a for a, _ in pose_stats), sum(b for _, b in pose_stats)
            dvf, hsf = dvf // (rep + 1), hsf // (rep + 1)                # the counters run over both passes
            solve_info = {"frames": PF, "stream": args.pose_stream, "frames_per_s": round(PF / (t4 - t1), 1), "frames_on_device_chain": dvf, "frames_on_host_path": hsf, "pnp_on": ("gpu (one block per chain)" if (args.pose_stream == "rigid" and NT > 1) else "gpu (%s blocks per chain)" % os.environ.get("SSM_PNP_BLOCKS", "8")) if args.pnp_device else "host (1 core)", "sequences_in_flight": (NT if args.pose_stream == "rigid" else 1),
                          "ms": {"orb_match": round((t2 - t1) * 1e3, 2), "pose_chain": round((t3 - t2) * 1e3, 2), "map": round((t4 - t3) * 1e3, 2)},
                          "tracked_frames": int(info_s["tracked"].sum()), "lost_events": int((info_s["state"] == 2).sum()), "voxels": int(nv),
                          "note": "Tracker::updateFrame for every frame (src/track.cpp:140-200): the PnP chain is serial by construction (frame f starts from frame f-1's pose)"}
        # roofline of the chain: the f64 operations PnPSolver::solvePnP's passes over the edges need (include/ssm/pnp_core.h, counted from the source: an edge's chi2
        # term = edge_map 18 + edge_error 8 + chi2 3 + Huber 2 = 31; a Levenberg iteration adds the Jacobian 22 + weights 6 + the 40 non-zero accumulations x 2 + 10 = 118)
        # x the passes and level-0 edges the device chain counted (ssm_tracker_work), against the f64 vector rate of the CUs the chain occupies
        if args.pnp_device and NT == 1:
            wk = trk.work(); nblk = int(os.environ.get("SSM_PNP_BLOCKS", "8"))
            fl = (31.0 + 118.0) * wk[2] + 31.0 * wk[3]
            sec = solve_info["ms"]["pose_chain"] * 1e-3 * 2                      # the counters ran over both passes
            peak = 78.6e12 * nblk / 256
            solve_info["work"] = {"levenberg_iterations_per_frame": round(wk[0] / (2.0 * PF), 1), "chi2_passes_per_frame": round(wk[1] / (2.0 * PF), 1),
                                  "edges_per_fused_pass": round(wk[2] / max(wk[0], 1), 1)}
            solve_info["roofline"] = {"bound": "valu_f64", "kernel": "pnp_chain_kernel (%d blocks = %d CUs)" % (nblk, nblk), "achieved": round(fl / sec / 1e9, 2), "peak": round(peak / 1e9, 1), "unit": "GFLOP/s",
                                      "frac": round(fl / sec / peak, 4), "traffic": None,
                                      "note": "a serial chain: the Levenberg iterations and trial rounds of `work` per frame, each a pass over the edges followed by a 6 x 6 solve that the next pass waits for; "
                                              "latency-bound by construction (DESIGN.md s.5.1), the peak is that of the CUs it occupies (78.6 TFLOP/s f64 vector x blocks / 256)"}
        if args.pose_cpu_sample and args.pose_stream == "rigid":
            # cpu_baseline of the pose loop: the same chain (include/ssm/pnp_core.h, the arithmetic oracle/pnp.c pins) on ONE host core over a bounded sample of the sequences
            ns = min(args.pose_cpu_sample, PF)
            o2 = ctx.seq_process(bgr.data_ptr(), dep.data_ptr(), None, None, PF, stages=ssm.api.STAGE_ORB | ssm.api.STAGE_MATCH); ctx.sync()
            ht = ssm.Tracker(ctx, use_device=False)
            th0 = time.perf_counter()
            for a0 in range(0, ns, CH):
                ht.reset(); ht.run(view(a0), min(CH, ns - a0))
            solve_info["host_chain_ms_per_frame"] = round((time.perf_counter() - th0) * 1e3 / ns, 3); solve_info["host_chain_sample_frames"] = ns
            ht.close()
        trk.close()
        for e in (extra or []): e.close()
        ctx.map_clear()
    frames_all = F
    per_rank = None
    if world > 1:
        # every rank's own rate and voxel count (rank order), so that a SCALE record explains itself: a slow rank, an uneven split or a long merge shows in the line
        mine = torch.tensor([dt, float(F), float(stage_ovl.get("allgather", [0.0, 0])[0])], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = {"frames_per_s": [round(float(a[1]) * args.steps / float(a[0]), 1) for a in allr],
                    "allgather_ms_per_step": [round(float(a[2]) / max(args.steps, 1), 3) for a in allr]}
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        t = torch.tensor([F], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        frames_all = int(t.item())

    P_total = int(res["npoints"].sum()); kp_total = int(res["nkp"].sum())
    m = res["nmatch"]; match_total = int(m[m > 0].sum())
    P_all = P_total
    if world > 1:
        t = torch.tensor([P_total], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        P_all = int(t.item())
    if rank == 0:
        frames_total = frames_all * args.steps        # frames of all ranks (rank 0's stage breakdown below is per ITS frames)
        value = frames_total / dt
        P = P_total / F; nkp = kp_total / F
        # dominant kernel and its roofline
        per_stage = {k: (v[0] / max(v[1], 1), v[0] / args.steps / F * 1e3) for k, v in stage_acc.items()}   # ms/launch-group, us/frame
        nb = -(-F // args.batch)
        dom = max(stage_acc, key=lambda k: stage_acc[k][0])
        launches_per_step = stage_acc[dom][1] / args.steps
        ms_per_launch = stage_acc[dom][0] / max(stage_acc[dom][1], 1)
        frames_per_launch = F / launches_per_step
        if dom == "segnet":
            from semantic_slam_mapping_amd import segnet_model
            tf = segnet_model.flops() * frames_per_launch / (ms_per_launch * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": "conv3x3_dma2_kernel + conv3x3_first_kernel (26 conv layers with fused pool / un-pool / ArgMax, prep, colouring: whole SegNet stage)", "achieved": round(tf, 1), "peak": 2500.0,
                    "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4), "traffic": measured_traffic("segnet", frames_per_launch, "segnet_traffic.json")}
            if roof["traffic"] is None:
                roof["traffic_source"] = traffic_profile("segnet_traffic.json", "kernels_segnet.hip")[1]
            else:
                roof["traffic_source"] = os.path.relpath(latest_profile("segnet_traffic.json"), ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --segnet`, scripts/collect_profiles.sh pmc_segnet; per frame x frames per launch group, not measured in this run)"
        elif dom == "match":
            # K6 runs on the matrix cores (v_mfma_scale_f32_32x32x64_f8f6f4 on descriptors expanded to +-1 FP4 elements): 2 x 256 operations per
            # descriptor pair, against the dense FP4 peak (4 x the bf16 rate: MI355X_MICROARCH.md, Matrix cores)
            pairs = sum(int(res["nkp"][max(f - R + r, 0)]) * int(res["nkp"][f]) for f in range(F) for r in range(R) if m[f, r] >= 0) / F
            tops = pairs * 512 * frames_per_launch / (ms_per_launch * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": "match_mfma_kernel (+ match_expand_kernel, match_compact_kernel)", "achieved": round(tops, 1), "peak": 10000.0, "unit": "TOP/s (fp4)",
                    "frac": round(tops / 10000.0, 4), "traffic": measured_traffic(dom, frames_per_launch),
                    "note": "Hamming distance matrix as an exact FP4 matrix product; algorithmic operations only (the padding of 1000 descriptors to 1024 is not counted); "
                            "the kernel's longer side is the VALU top-2 tracking of the distance tiles: DESIGN.md s.4.1",
                    "descriptor_pairs_per_frame": round(pairs)}
        else:
            gb = algorithmic_bytes(dom, P, nkp) * frames_per_launch / 1e9
            ach = gb / (ms_per_launch * 1e-3)
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": measured_traffic(dom, frames_per_launch),
                    "algorithmic_bytes_per_launch": round(gb * 1e9),
                    "note": "integer/byte kernel limited by VALU issue (4 cycles per instruction, profiles/r02_valu_rate.md) and LDS, not HBM: see `valu` and DESIGN.md s.4"}
        if roof.get("bound") == "hbm":
            roof["valu"] = measured_valu(dom, per_stage[dom][1])      # what actually bounds the integer stages
        try:    # the whole pipeline against the VALU issue ceiling: every kernel's wave-instructions per frame (committed SQ pass) x 64 x frames/s
            kk = json.load(open(latest_profile("sq_counters.json")))["kernels"]
            wi = sum(v["valu_wave_insts_per_frame"] for name, v in kk.items() if "synth" not in name)
            if not args.segnet:
                roof["pipeline_valu"] = {"achieved": round(wi * 64 * value / 1e12, 2), "peak": VALU_LANEOPS_PEAK / 1e12, "unit": "Tlaneop/s",
                                         "frac": round(wi * 64 * value / VALU_LANEOPS_PEAK, 3), "frac_of_simple_op_peak": round(wi * 64 * value / VALU_SIMPLE_OP_PEAK, 3),
                                         "wave_insts_per_frame": round(wi),
                                         "note": "all kernels of a frame (ORB, match, map) at the timed rate `value`, two chains overlapped"}
        except Exception:
            pass
        if roof.get("traffic") is None and "traffic_source" not in roof and dom in ("map_fuse", "voxel_insert"):
            roof["traffic_source"] = traffic_profile("traffic.json", "kernels_map.hip")[1]      # why there is no figure (a stale or missing counter pass)
        if roof.get("traffic") is not None and "traffic_source" not in roof:     # `traffic` is NOT measured in this run: it is this kernel's figure from the committed counter passes of the same command
            roof["traffic_source"] = os.path.relpath(latest_profile("traffic.json"), ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, scripts/collect_profiles.sh pmc; per frame x frames per launch)"
        roof["stages_us_per_frame"] = {k: round(v[1], 3) for k, v in sorted(per_stage.items(), key=lambda kv: -kv[1][1])}
        roof["stages_us_per_frame_overlapped"] = {k: round(v[0] / args.steps / F * 1e3, 3) for k, v in sorted(stage_ovl.items(), key=lambda kv: -kv[1][0])}
        roof["timing"] = ("achieved / stages_us_per_frame: hipEvents around each stage in a second pass of the same K steps with all stages "
                          "serialised on one stream; value / ms_per_step: the timed region, where the map (and SegNet) stage runs on a second "
                          "stream beside ORB + match (stages_us_per_frame_overlapped)")
        cpu = None
        if world == 1 and not args.no_cpu and args.cpu_frames > 0:
            from oracle.binding import Oracle, build
            try:
                build(native=True); orc = Oracle(native=True)
            except Exception:
                orc = Oracle()
            st = orc.pipeline(0, args.cpu_frames, nfeatures=1000, leaf=np.float32(args.leaf))
            tcpu = st["t_orb"] + st["t_match"] + st["t_mask"] + st["t_backproject"] + st["t_voxel"]
            cpu = {"value": round(args.cpu_frames / tcpu, 3), "unit": "frames/s", "cores": 1, "kind": "port",
                   "sample": f"first {args.cpu_frames} frames of the same stream, oracle/ (C, -O3 -march=native, 1 thread, like the reference: no -fopenmp), synth excluded",
                   "ms_per_frame": {k[2:]: round(st[k] / args.cpu_frames * 1e3, 3) for k in ("t_orb", "t_match", "t_mask", "t_backproject", "t_voxel")}}
            try:    # the generous baseline, every host core
                cpu["all_cores"] = cpu_all_cores(orc, max(8, args.cpu_frames // 3), args.leaf)
            except Exception as e:
                cpu["all_cores"] = {"error": str(e)}
        line = {
            "metric": "frames/sec semantic-mapping, 640x480 RGB-D", "value": round(value, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f16" if args.segnet else "u8", "data": "synthetic",
            "config": {"workload": ("configs[2]: SegNet driving_webdemo fp16 on-GPU (seeded weights), synthetic 640x480 RGB-D, full pipeline, %d frames per GPU, "
                                    "ORB 1000 kp/frame, 5 ref frames, leaf %.2f m" % (F, args.leaf)) if args.segnet else
                                   ("configs[4]: %d-frame synthetic 640x480 RGB-D stream block-sharded over %d GPUs, precomputed 12-class masks, ORB 1000 kp/frame, "
                                    "5 ref frames (+ matcher halo), leaf %.2f m, RCCL all-gather voxel-map merge" % (args.total_frames, world, args.leaf)) if args.total_frames > 0 else
                                   ("configs[1]: synthetic 640x480 RGB-D + precomputed 12-class masks, 1k frames per GPU, ORB 1000 kp/frame, "
                                    "5 ref frames, leaf %.2f m" % args.leaf),
                       "frames_per_gpu": F, "frames_all_gpus": frames_all, "batch_frames": args.batch, "matcher_halo_frames": HN,
                       "parallelism": "contiguous frame blocks x%d, one RCCL all-gather of the voxel tables per step (ssm_voxel_allgather)" % world if world > 1 else "single GPU"},
            "mpoints_per_s": round(P_all * args.steps / dt / 1e6, 2),
            "frames_per_s_including_h2d": None if h2d_fps is None else round(h2d_fps, 1),
            "per_frame": {"keypoints": round(nkp, 1), "matches": round(match_total / F, 1), "points": round(P, 1), "voxels_in_map": int(n_vox)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if solve_info is not None:
            line["solve_poses"] = solve_info
        if per_rank is not None:
            line["per_rank"] = per_rank
        if "allgather" in stage_ovl:
            line["allgather_ms_per_step"] = round(stage_ovl["allgather"][0] / max(args.steps, 1), 3)      # ssm_voxel_allgather on rank 0: count all-gather + table all-gather + merge kernels (hipEvents on the context stream)
        if merge_info is not None:
            line["merge_verified"] = merge_info["verified"]
            line["merge"] = merge_info
    else:
        line = None
    if world > 1 or force_merge:
        ctx.comm_finalize()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    del bgr, dep, sem, pose, tab_buf
    ctx.close()
    return line


def segnet_cpu_baseline(pipeline_cpu):
    """configs[2] on the host: the reference runs Caffe-SegNet's ForwardPrefilled (src/segnet.cpp:99) before the per-frame path; here one 360 x 480 forward of the same
    topology and seeded weights through PyTorch-CPU fp32 (tests/segnet_ref.py: the checker of the SegNet kernels, timed here as the baseline) on every core the box
    grants, + the configs[1] oracle pipeline's per-frame time"""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import segnet_ref
    from semantic_slam_mapping_amd import segnet_model
    wts = segnet_model.make_weights(1234)
    x = np.random.default_rng(7).integers(0, 256, (3, 360, 480)).astype(np.uint8)
    cores = granted_cores()                                 # (torch would start a thread per visible core: the box shows 128, its quota is 16)
    torch.set_num_threads(cores)
    segnet_ref.forward(x, wts)                              # first call: thread pool, allocator
    t0 = time.perf_counter(); n = 0
    while n < 2 or (time.perf_counter() - t0 < 6.0 and n < 8):
        segnet_ref.forward(x, wts); n += 1
    t_seg = (time.perf_counter() - t0) / n
    t_pipe = 1.0 / pipeline_cpu["value"] if pipeline_cpu else 0.0
    return {"value": round(1.0 / (t_seg + t_pipe), 3), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d SegNet forwards (PyTorch-CPU fp32, %d threads, 213.6 GFLOP each: %.0f GFLOP/s) + the configs[1] oracle pipeline's per-frame time (1 thread)" % (n, cores, segnet_model.flops() / t_seg / 1e9),
            "ms_per_frame": {"segnet": round(t_seg * 1e3, 1), "pipeline": round(t_pipe * 1e3, 2)}}


def exp_mapping_legs(scale=1.0):
    """the PRODUCT -- semantic_slam_mapping_amd/host/exp_mapping, the C++ drop-in of the reference's experiment/exp_mapping.cpp:18-59 on the
    rgbd_tutor classes of include/ssm/ (FrameReader::next -> Tracker::updateFrame -> PoseGraph::tryInsertKeyFrame, Mapper::viewer on its own thread) -- run as CHILD
    processes, one after the other, BEFORE this process imports torch or makes its first HIP call (a process that has initialised the GPU must neither fork nor exec).
    The rates are the ones the binary prints itself, timed where the reference's drivers time (experiment/run_tracker.cpp:35-48 around updateFrame,
    experiment/match_orbfeature_tum.cpp:22-27 around detect + match), the first frames left out (context creation, code-object load).
    Stream: 640 x 480 frames of the rigid form of the configs[1] stream (FrameReader `synthetic_rigid`: frame 0 as a plane at 2 m under a panning camera, independent
    20-frame sequences -- the stream on which every frame tracks, the same one the pose_loop leg uses); stereo: a KITTI-layout directory of 1241 x 376 PNG pairs
    written here (FrameReader::KITTI reads and decodes them like the reference's cv::imread calls, src/rgbdframe.cpp:34-80)."""
    import shutil
    import subprocess
    import tempfile
    host = os.path.join(ROOT, "semantic_slam_mapping_amd", "host"); exe = os.path.join(host, "exp_mapping")
    if not os.path.exists(exe):
        return {"error": "semantic_slam_mapping_amd/host/exp_mapping is not built (python -c 'import __graft_entry__ as g; g.build()')"}
    t_all = time.perf_counter()
    tmp = tempfile.mkdtemp(prefix="ssm_expmap_")
    N = max(60, int(400 * scale)); SKIP = 40
    CH = min(200, N); NB = 3 * CH                       # the chain-less bulk leg: three chunks, the first one untimed
    base = open(os.path.join(host, "parameters_test.txt")).read().replace("end_index=8", "end_index=%d" % N).replace("map_output=/tmp/ssm_test_map.pcd", "")
    # key-frame gate: 0.03 m = every 4th frame of this stream (2 x 2 m / 517 px = 7.7 mm per frame; the reference's 5.5 is for KITTI's metres per frame)
    base = base.replace("keyframe_min_translation=0.005", "keyframe_min_translation=0.03")
    base += "\nsynthetic_rigid=1\nsequence_length=20\ntiming_skip_frames=%d\nfinal_map_fnv=60\nmapper_drain_ms=100\n" % SKIP

    def run(name, text, *flags, timeout=240):
        prm = os.path.join(tmp, name + ".txt")
        open(prm, "w").write(text)
        t0 = time.perf_counter()
        r = subprocess.run([exe, prm, *flags], capture_output=True, text=True, timeout=timeout)
        wall = time.perf_counter() - t0
        rows = [l for l in r.stdout.splitlines() if l.startswith("frames ")]
        if r.returncode != 0 or not rows:
            return {"error": "rc %d: %s" % (r.returncode, (r.stderr or r.stdout)[-400:])}
        w = rows[-1].split()
        st = dict(zip(w[0::2], w[1::2]))
        st["wall_s"] = round(wall, 1); st["lost"] = r.stdout.count("tracker is lost")
        return st

    out = {}
    try:
        legs = {
            "per_frame": run("a", base + "use_stream_pose=1\n"),
            "per_frame_solved": run("b", base + "use_stream_pose=0\n"),
            # the bulk loops: frames preloaded into (page-locked) host memory, so that the loop measured is upload + tracker + key-frame gate, not the synthetic reader
            "batched": run("c", base.replace("end_index=%d" % N, "end_index=%d" % NB) + "use_stream_pose=1\ntracker_batched_chain=0\ntracker_chunk=%d\nssm_max_batch=%d\nreader_preload=1\ntiming_skip_frames=%d\n" % (CH, CH, CH), "--batched"),
            "batched_solved": run("c2", base + "use_stream_pose=0\ntracker_chunk=20\nssm_max_batch=20\nreader_preload=1\n", "--batched"),
        }
        # ---- stereo: a KITTI-layout directory written here (PNG, gray), then `exp_mapping --batched` with tracker_mode = stereo (BatchStereoTracker)
        import numpy as np
        from PIL import Image
        NS = max(24, int(96 * scale)); SCH = 32 if NS >= 96 else 8
        L, R = stereo_sequence(NS + 1, 1241, 376, 100)
        seq = os.path.join(tmp, "kitti"); os.makedirs(os.path.join(seq, "image_2")); os.makedirs(os.path.join(seq, "image_3"))
        for i in range(NS + 1):
            Image.fromarray(L[i], "L").save(os.path.join(seq, "image_2", "%06d.png" % i), compress_level=1)
            Image.fromarray(R[i], "L").save(os.path.join(seq, "image_3", "%06d.png" % i), compress_level=1)
        st_txt = open(os.path.join(host, "parameters_test.txt")).read().replace("end_index=8", "end_index=%d" % NS).replace("dataset=synthetic", "dataset=kitti")
        st_txt = st_txt.replace("map_output=/tmp/ssm_test_map.pcd", "").replace("image_width=640", "image_width=1241").replace("image_height=480", "image_height=376")
        for k, v in (("camera.cx", KITTI["cu"]), ("camera.cy", KITTI["cv"]), ("camera.fx", KITTI["f"]), ("camera.fy", KITTI["f"])):
            st_txt = "\n".join(("%s=%r" % (k, v)) if ln.startswith(k + "=") else ln for ln in st_txt.splitlines())
        st_txt += ("\ndata_source=%s\ntracker_mode=stereo\ncamera.baseline=%r\ncamera.roix=%r\ncamera.roiy=%r\ncamera.roiz=%r\ninlier_threshold=2.0\ntracker_chunk=%d\nssm_max_batch=%d\n"
                   "timing_skip_frames=%d\nmapper_drain_ms=100\nkeyframe_min_translation=0.5\nreader_preload=1\n" % (seq, KITTI["baseline"], KITTI["roix"], KITTI["roiy"], KITTI["roiz"], SCH, SCH, SCH))
        legs["batched_stereo"] = run("d", st_txt, "--batched")
        # the per-frame stereo loop -- what the CHECKED-IN reference runs (src/track.cpp:19 calls estimateVO): FrameReader::next computes the depth by SGBM (preloaded here, so
        # that its time shows as preload_s, not in the loop), Tracker::updateFrame = ORB + QuadFeatureMatch + VisualOdometryStereo::Process per frame
        legs["per_frame_stereo"] = run("e", st_txt.replace("timing_skip_frames=%d" % SCH, "timing_skip_frames=8"))
        out["runs"] = legs
        f = lambda leg, key: (round(float(legs[leg][key]), 1) if key in legs[leg] else None)
        f2 = lambda v: None if v is None else round(float(v), 4)
        out["per_frame_fps"] = f("per_frame", "tracker_fps"); out["per_frame_solved_fps"] = f("per_frame_solved", "tracker_fps")
        out["batched_fps"] = f("batched", "loop_fps"); out["batched_solved_fps"] = f("batched_solved", "loop_fps"); out["batched_stereo_pairs_per_s"] = f("batched_stereo", "loop_fps")
        out["per_frame_stereo_pairs_per_s"] = f("per_frame_stereo", "tracker_fps")
        if "preload_s" in legs["per_frame_stereo"]:           # FrameReader::next of the KITTI reader per frame: five PNG decodes + SGBM depth (ssm_stereo_depth)
            out["per_frame_stereo_reader_ms"] = round(float(legs["per_frame_stereo"]["preload_s"]) * 1e3 / max(int(legs["per_frame_stereo"]["frames"]), 1), 2)
        out["per_frame_loop_fps_reader_included"] = {k: f(k, "loop_fps") for k in ("per_frame", "per_frame_solved")}
        out["per_call_ms"] = {k: f2(legs["per_frame_solved"].get(k)) for k in ("detect_ms", "match_ms", "pnp_ms", "tracker_ms", "reader_ms")}
        ok = all("error" not in v for v in legs.values())
        # stream poses: the per-frame and the bulk loop keep the same poses, pick the same key-frames and build the same map; solved poses: the bulk chain is the per-frame Tracker, bit for bit
        out["map_fnv_equal"] = bool(ok and legs["per_frame"].get("map_fnv") is not None and legs["per_frame"].get("map_fnv") == legs["batched"].get("map_fnv")
                                    and legs["per_frame_solved"].get("map_fnv") == legs["batched_solved"].get("map_fnv"))
        out["pose_fnv_equal"] = bool(ok and legs["per_frame_solved"]["pose_fnv"] == legs["batched_solved"]["pose_fnv"])       # (the chain-less bulk leg runs more frames: its poses are the stream's)
        out["unit"] = "frames/s (stereo: frame pairs/s)"
        out["config"] = {"workload": "exp_mapping (C++ host, include/ssm classes over the C ABI): %d frames 640x480 of the rigid configs[1] stream in 20-frame sequences, ORB 1000 kp, 5 refs, rates over the frames after "
                                     "the first %d; stereo: %d PNG pairs 1241x376 from a KITTI-layout directory, SGBM 80 disparities" % (N, SKIP, NS),
                         "timed": "per_frame*: frames / wall time inside Tracker::updateFrame (where experiment/run_tracker.cpp:35-48 times it): uploads, kernels, downloads, host state machine, Mapper::viewer busy on its own "
                                  "thread and context; per_frame_loop_fps_reader_included adds FrameReader::next (host roll of the base frame) and PoseGraph::tryInsertKeyFrame.  batched*: frames / wall time of the whole loop "
                                  "over frames preloaded into page-locked host memory (reader_preload: BatchTracker / BatchStereoTracker push + flush -- upload, ORB + match tables [+ PnP chain] or quad matcher + SGBM + VO, "
                                  "per-frame depth download for stereo -- then the key-frame gate), chunk %d (stream poses), 20 (solved), %d (stereo); frame buffers from ssm_host_alloc (reader_pinned)" % (CH, SCH)}
    except Exception as e:                      # a failing leg must not take the headline with it
        out["error"] = repr(e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out["wall_s"] = round(time.perf_counter() - t_all, 1)
    return out


def sub_line(line, **extra):
    """what `other_configs` keeps of a full bench line"""
    keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "cpu_baseline", "per_frame")
    d = {k: line[k] for k in keep if k in line}
    d.update(extra)
    return d


def other_configs(args, head):
    """The default run's extra legs: configs[2] (SegNet on the GPU), configs[3] (stereo front end) and the closed pose loop, short enough
    for the whole command to stay within ~90 s, each with its own value / ms_per_step / steps / roofline / cpu_baseline.  The headline `value` is never touched."""
    import copy
    sc = args.other_scale
    out = {}
    t_all = time.perf_counter()
    legs_order = os.environ.get("SSM_BENCH_LEG_ORDER", "stereo,segnet")      # (ablation: which of the two heavy legs runs first; the second one meets a warmer chip)
    def leg_segnet():
        # ---- configs[2]: the SegNet stage in front of the map stage (labels from the network instead of the precomputed masks)
        a = copy.copy(args)
        a.segnet, a.solve_poses, a.frames, a.batch, a.steps, a.warmup = True, False, max(16, int(256 * sc)), max(8, int(128 * sc)), 3, 1
        a.no_cpu = True                                        # its CPU leg is segnet_cpu_baseline below (the pipeline part is the headline's sample)
        t0 = time.perf_counter()
        try:
            ln = rgbd_main(a)
            if not args.no_cpu:
                ln["cpu_baseline"] = segnet_cpu_baseline(head.get("cpu_baseline"))
            out["configs[2]"] = sub_line(ln, wall_s=round(time.perf_counter() - t0, 1))
        except Exception as e:                                  # a failing leg must not take the headline with it; it is reported
            out["configs[2]"] = {"error": repr(e)}
    def leg_stereo():
        # ---- configs[3]: the stereo front end on resident 1241 x 376 pairs
        a = copy.copy(args)
        a.stereo, a.frames, a.steps, a.warmup, a.stereo_batch = True, max(8, int(256 * sc)), 3, 1, max(4, min(args.stereo_batch, int(128 * sc)))
        t0 = time.perf_counter()
        try:
            out["configs[3]"] = sub_line(stereo_main(a), wall_s=round(time.perf_counter() - t0, 1))
        except Exception as e:
            out["configs[3]"] = {"error": repr(e)}
    for leg in legs_order.split(","):
        (leg_segnet if leg.strip() == "segnet" else leg_stereo)()
    # ---- the closed pose loop: measured inside the headline's process on its resident stream (rgbd_main's --solve-poses leg); re-shaped here
    sp = head.pop("solve_poses", None)
    if sp is not None:
        cpu = None
        if "host_chain_ms_per_frame" in sp and head.get("cpu_baseline"):
            c1 = head["cpu_baseline"]["ms_per_frame"]
            t_front = (c1["orb"] + c1["match"] + c1["mask"] + c1["backproject"] + c1["voxel"]) * 1e-3
            cpu = {"value": round(1.0 / (t_front + sp["host_chain_ms_per_frame"] * 1e-3), 3), "unit": "frames/s", "cores": 1, "kind": "port",
                   "sample": "the pose chain of the first %d frames of the same sequences on one host core (include/ssm/pnp_core.h = the arithmetic of oracle/pnp.c) + the configs[1] "
                             "oracle pipeline's per-frame time" % sp["host_chain_sample_frames"],
                   "ms_per_frame": {"pose_chain": sp["host_chain_ms_per_frame"], "front_end": round(t_front * 1e3, 2)}}
        ms = sp["ms"]
        out["pose_loop"] = {"metric": "frames/sec with the poses solved by the pipeline (Tracker::updateFrame for every frame, src/track.cpp:140-200): ORB + match tables -> PnP chain -> map stage with the solved poses",
                            "value": sp["frames_per_s"], "unit": "frames/s", "steps": 1, "warmup": 1, "ms_per_step": round(ms["orb_match"] + ms["pose_chain"] + ms["map"], 2), "dtype": "f64",
                            "config": {"workload": "%d frames of the rigid stream (frame 0 of configs[1] seen by a panning camera: every frame tracks) as independent 20-frame sequences, one sequence in flight" % sp["frames"]},
                            "pose_chain_ms_per_frame": round(ms["pose_chain"] / sp["frames"], 4),
                            "roofline": sp.get("roofline"), "cpu_baseline": cpu, "detail": {k: v for k, v in sp.items() if k not in ("roofline",)}}
    out["wall_s"] = round(time.perf_counter() - t_all, 1)
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))              # nothing below has run: no torch import, no HIP call in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # the default run (the command the driver times: N = 1, no mode flag) also carries configs[2], configs[3] and the closed pose loop as `other_configs`
    default_run = (world == 1 and args.other_configs and (args.frames == 1000 or args.other_scale != 1.0)
                   and not (args.stereo or args.segnet or args.solve_poses or args.serial_only or args.total_frames > 0 or os.environ.get("SSM_FORCE_MERGE") == "1"))
    expmap = None
    if default_run and os.environ.get("SSM_BENCH_EXP_MAPPING", "1") != "0":
        expmap = exp_mapping_legs(args.other_scale)   # child processes, started and finished before this process's first GPU call
    if args.stereo:
        line = stereo_main(args)
    else:
        if default_run:                               # the pose loop rides on the headline's resident stream (after its timed region and its serialised pass)
            args.solve_poses, args.pnp_device, args.pose_frames, args.pose_stream, args.pose_threads = True, 1, max(40, int(400 * args.other_scale)), "rigid", 1
            args.pose_cpu_sample = 0 if args.no_cpu else 100
        line = rgbd_main(args)
        if default_run and line is not None:
            line["other_configs"] = other_configs(args, line)
            if expmap is not None:
                if "error" not in expmap and line.get("cpu_baseline"):
                    cb = line["cpu_baseline"]
                    expmap["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                              "sample": "the configs[1] oracle pipeline (ORB + match + mask + back-projection + voxel filter per frame, no PnP), as in the headline's cpu_baseline"}
                line["other_configs"]["exp_mapping"] = expmap
    if line is not None:
        # RCCL prints a version banner through C stdio when a communicator is created; flush it first so that the JSON line is the LAST line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
