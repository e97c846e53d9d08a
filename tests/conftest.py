import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CAM = (318.6, 255.3, 517.3, 516.5, 1000.0)     # TUM fr1 intrinsics, camera.scale 1000 (parameters.txt:63)
SEED = 0x5EED0000


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    """a HIP device is present (no torch import: the kernel driver's node is enough to tell a GPU box from the build container)"""
    return os.environ.get("SSM_FORCE_GPU_TESTS") == "1" or os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device (/dev/kfd missing): gpu-marked tests need a real MI355X")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle.binding import Oracle, build
    build()
    return Oracle()


@pytest.fixture(scope="session")
def ctx():
    """libssm_hip.so context on cuda:0, BASELINE configs[1] parameters (ORB 1000 kp, leaf 0.1)."""
    import semantic_slam_mapping_amd as ssm
    c = ssm.Context(0, orb_features=1000, max_batch=4, voxel_capacity_log2=18, camera=CAM)
    yield c
    c.close()


@pytest.fixture(scope="session")
def frames(oracle):
    """first 8 frames of the synthetic stream (host copies)."""
    return [oracle.synth_frame(SEED, f) for f in range(8)]


def rand_desc(rng, n):
    return rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
