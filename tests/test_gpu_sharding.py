"""Multi-GPU path on ONE GPU (the round-end box has one): two contexts stand in for two ranks.

* the block plan + matcher halo (sharding.frame_block / halo_block, the calls bench.py makes per rank) must reproduce the single-GPU
  match tables byte for byte: Tracker::trackRefFrame matches against the tracker_ref_frames preceding frames whoever owns them
  (/root/reference/src/track.cpp:150-152,192-196);
* the table a second context exports, merged by ssm_map_merge_table_dev (the kernel ssm_voxel_allgather runs on the remote tables),
  must give the single-GPU map byte for byte;
* ssm_voxel_allgather itself runs through RCCL with a 1-rank communicator (RCCL refuses two ranks on one device).
"""
import numpy as np
import pytest

import semantic_slam_mapping_amd as ssm
from semantic_slam_mapping_amd import sharding
from conftest import CAM, SEED

pytestmark = pytest.mark.gpu
W, H = 640, 480


def _ctx(**kw):
    return ssm.Context(0, orb_features=1000, max_batch=4, voxel_capacity_log2=18, camera=CAM, **kw)


def _frames(c, first, n):
    bufs = [c.dev_alloc(n * W * H * 3), c.dev_alloc(n * W * H * 2), c.dev_alloc(n * W * H * 3), c.dev_alloc(n * 128)]
    c.synth_frames_dev(SEED, first, n, *bufs)
    return bufs


def _run_block(c, lo, hi, R, halo=True):
    """what bench.py does per rank: halo frames through ORB only, then the block with continue_sequence"""
    c.map_clear()
    hlo, hhi = sharding.halo_block(lo, R) if halo else (lo, lo)
    hn = hhi - hlo
    keep = []
    if hn:
        hb = _frames(c, hlo, hn); keep.append(hb)
        c.seq_process(hb[0], None, None, None, hn, stages=ssm.api.STAGE_ORB)
    fb = _frames(c, lo, hi - lo); keep.append(fb)
    out = c.seq_process(*fb, hi - lo, continue_sequence=hn > 0)
    c.sync()
    res = c.seq_fetch(out, hi - lo)
    for b in keep:
        for p in b:
            c.dev_free(p)
    return res


@pytest.mark.parametrize("n_frames,world", [(12, 2), (11, 3), (43, 8)])      # (43, 8): the rank count of BASELINE configs[4] -- eight contexts on one GPU, uneven blocks of 5 and 6 frames (a block as short as the 5-frame halo)
def test_sharded_blocks_equal_single_gpu(n_frames, world):
    single = _ctx()
    ref = _run_block(single, 0, n_frames, single.R)
    ref_map = single.map_export_table()
    R = single.R
    ranks = [_ctx() for _ in range(world)]
    try:
        covered = 0
        for r, c in enumerate(ranks):
            lo, hi = sharding.frame_block(n_frames, r, world)
            assert lo == covered; covered = hi
            res = _run_block(c, lo, hi, R)
            assert np.array_equal(res["nkp"], ref["nkp"][lo:hi])
            for f in range(hi - lo):
                k = int(res["nkp"][f])
                assert res["desc"][f, :k].tobytes() == ref["desc"][lo + f, :k].tobytes() and res["kps"][f, :k].tobytes() == ref["kps"][lo + f, :k].tobytes()
            assert np.array_equal(res["nmatch"], ref["nmatch"][lo:hi]), "halo: the match counts at the block start differ from the single-GPU run"
            for f in range(hi - lo):
                for k in range(R):
                    nm = int(res["nmatch"][f, k])
                    if nm > 0:
                        assert res["matches"][f, k, :nm].tobytes() == ref["matches"][lo + f, k, :nm].tobytes()
            assert np.array_equal(res["npoints"], ref["npoints"][lo:hi])
        assert covered == n_frames
        # merge: every rank's table into rank 0's map through device buffers (the remote-table step of ssm_voxel_allgather)
        c0 = ranks[0]
        for c in ranks[1:]:
            n = c.map_size()
            buf = c.dev_alloc(max(n, 1) * sharding.VOXEL_BYTES)
            assert c.map_export_table_dev(buf, n) == n
            c0.map_merge_table_dev(buf, n)        # same device: the buffer of one context is readable by the other
            c0.sync()
            c.dev_free(buf)
        assert len(ref_map) > 1000
        assert c0.map_export_table().tobytes() == ref_map.tobytes()
        assert c0.map_export().tobytes() == single.map_export().tobytes()
    finally:
        for c in ranks:
            c.close()
        single.close()


def test_block_without_halo_lacks_references():
    """the halo is what makes the shards match the single-GPU tables: without it the first R frames of a block have no references"""
    c = _ctx()
    try:
        res = _run_block(c, 6, 9, c.R, halo=False)
        assert (res["nmatch"][0] == -1).all() and (res["nmatch"][2, :c.R - 2] == -1).all() and (res["nmatch"][2, c.R - 2:] >= 0).all()
    finally:
        c.close()


def test_voxel_allgather_one_rank_communicator():
    """RCCL behind the C ABI: unique id, ncclCommInitRank, the count + table all-gathers on the context stream; with one rank the map
    must come back unchanged"""
    c = _ctx()
    try:
        _run_block(c, 0, 3, c.R)
        before = c.map_export_table()
        with pytest.raises(ssm.SsmError):
            c.voxel_allgather()                   # no communicator yet
        uid = c.comm_unique_id()
        assert len(uid) == ssm.api.COMM_ID_BYTES and any(uid)
        c.comm_init_rank(1, 0, uid)
        assert c.lib.ssm_comm_size(c.h) == 1 and c.lib.ssm_comm_rank(c.h) == 0
        c.voxel_allgather()
        c.voxel_allgather()
        c.sync()
        assert c.map_export_table().tobytes() == before.tobytes() and len(before) > 500
        c.comm_finalize()
    finally:
        c.close()
