"""Frame sharding across GPUs and the voxel-map merge (BASELINE.json configs[4]).

The reference is single-process; this is the one place a collective exists.  Frames of a sequence are split into
contiguous blocks, one per rank (each rank's matcher then only needs a tracker_ref_frames halo at its block start);
every rank fuses its own frames into its own voxel table; ONE all-gather of the tables (counts first, then tables padded
to the longest) merges them.  Tables hold exact integer sums, so the merged map is bit-identical to the single-GPU map
whatever the rank order.  On the GPU the collective lives behind the C ABI (ssm_voxel_allgather: RCCL inside
libssm_hip.so, no torch on the data path); `allgather_tables` below is the same exchange over torch.distributed, kept for
the world_size-2 gloo tests on CPU tensors, where it checks the plan (blocks, halos, counts, padding, merge).
"""
import numpy as np

VOXEL_BYTES = 112


def frame_block(n_frames, rank, world):
    """contiguous block [lo, hi) of rank; earlier ranks take the remainder"""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def halo_block(lo, ref_frames):
    """the matcher halo of a block that starts at frame lo: the <= ref_frames frames [lo - ref_frames, lo) in front of it.  A rank runs them
    through ORB only (stages = SSM_STAGE_ORB) and then its block with continue_sequence = 1: Tracker::trackRefFrame matches a frame against
    the refFrames deque = the ref_frames preceding frames (src/track.cpp:150-152,192-196), whoever owns them."""
    return max(0, lo - ref_frames), lo


def allgather_tables(local_table_u8, n_local, dist, device):
    """local_table_u8: torch.uint8 tensor [cap*112] on `device` holding n_local voxels.
    Returns (list of (tensor, n) for every rank) after one count all-gather and one padded table all-gather."""
    import torch
    world = dist.get_world_size()
    cnt = torch.tensor([n_local], dtype=torch.int64, device=device)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    counts = [int(c.item()) for c in cnts]
    mx = max(max(counts), 1)
    send = torch.zeros(mx * VOXEL_BYTES, dtype=torch.uint8, device=device)
    send[: n_local * VOXEL_BYTES] = local_table_u8[: n_local * VOXEL_BYTES]
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send)
    if device.type == "cuda":
        torch.cuda.synchronize(device)      # the merge runs on the context's own HIP stream
    return [(recv[r], counts[r]) for r in range(world)]


def merge_tables_numpy(tables):
    """CPU merge of key-sorted voxel tables (numpy structured arrays, VOXEL_DTYPE): used by the gloo tests as the
    checker of the device merge; exact integer sums, so order does not matter."""
    from .api import VOXEL_DTYPE
    allv = np.concatenate([np.asarray(t, VOXEL_DTYPE) for t in tables]) if tables else np.zeros(0, VOXEL_DTYPE)
    if len(allv) == 0:
        return allv
    order = np.argsort(allv["key"], kind="stable")
    allv = allv[order]
    keys, start = np.unique(allv["key"], return_index=True)
    out = np.zeros(len(keys), VOXEL_DTYPE)
    out["key"] = keys
    for f in ("sx", "sy", "sz", "sr", "sg", "sb", "n"):
        out[f] = np.add.reduceat(allv[f], start)
    out["hist"] = np.add.reduceat(allv["hist"], start, axis=0)
    return out


def merge_check(table_bytes, n_voxels, n_local_before_merge, dist, device):
    """Self-validation of the multi-GPU merge (bench.py prints it as `merge_verified`): after ssm_voxel_allgather every rank must hold the
    bit-identical key-sorted table.  Each rank hashes its exported merged table (CRC-32 of the raw 112-byte records); the (crc, voxel count,
    local count before the merge) triples are all-gathered and compared.  Returns a dict: verified (every rank has the same crc and count and the
    merged map is at least as large as the largest local one), crc, voxels_merged, voxels_per_rank (before the merge)."""
    import zlib
    import torch
    crc = zlib.crc32(bytes(table_bytes)) & 0xFFFFFFFF
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    mine = torch.tensor([crc, int(n_voxels), int(n_local_before_merge)], dtype=torch.int64, device=device)
    if world > 1:
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rows = [[int(v) for v in t.cpu().tolist()] for t in every]
    else:
        rows = [[int(v) for v in mine.cpu().tolist()]]
    same = all(r[0] == rows[0][0] and r[1] == rows[0][1] for r in rows)
    sane = rows[0][1] >= max(r[2] for r in rows) and rows[0][1] <= sum(r[2] for r in rows)
    return {"verified": bool(same and sane), "crc": "%08x" % rows[0][0], "voxels_merged": rows[0][1], "voxels_per_rank": [r[2] for r in rows],
            "ranks_agree": bool(same)}
