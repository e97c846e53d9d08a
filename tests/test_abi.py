"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol include/ssm_hip.h
declares, fails loudly (no CPU fallback), and its structs have the layouts the reference's types have."""
import ctypes
import os
import re
import subprocess
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "ssm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ssm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import semantic_slam_mapping_amd as ssm
    lib = ssm.load()
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"libssm_hip.so does not export {n}"
    assert sorted(ssm.SYMBOLS) == names        # the Python binding covers exactly the header


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import semantic_slam_mapping_amd as ssm
    with pytest.raises(ssm.SsmError) as e:
        ssm.Context(0)
    assert e.value.code == -7                  # SSM_E_NODEVICE


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "semantic_slam_mapping_amd")
    py = [f for f in os.listdir(pkg) if f.endswith(".py")]
    assert "api.py" in py and "_lib.py" in py
    for f in py:                                # every Python module of the package (the oracle's binding lives in oracle/)
        src = open(os.path.join(pkg, f)).read()
        assert "oracle" not in src.replace("oracle/ ", ""), f
    for f in os.listdir(os.path.join(ROOT, "semantic_slam_mapping_amd", "csrc")):
        if f.endswith((".hip", ".h", ".cpp")):
            src = open(os.path.join(ROOT, "semantic_slam_mapping_amd", "csrc", f)).read()
            code = re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", src, flags=re.S))      # comments may cite the oracle's contract
            assert "ssm_oracle" not in code and "sso_" not in code and "dlopen" not in code, f
    out = subprocess.run(["ldd", os.path.join(ROOT, "semantic_slam_mapping_amd", "libssm_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_struct_layouts_match_reference_types():
    import semantic_slam_mapping_amd as ssm
    k, m, p, v = ssm.KEYPOINT_DTYPE, ssm.DMATCH_DTYPE, ssm.POINT_DTYPE, ssm.VOXEL_DTYPE
    assert k.itemsize == 28 and [k.fields[n][1] for n in ("x", "y", "size", "angle", "response", "octave", "class_id")] == [0, 4, 8, 12, 16, 20, 24]   # cv::KeyPoint
    assert m.itemsize == 16 and [m.fields[n][1] for n in ("queryIdx", "trainIdx", "imgIdx", "distance")] == [0, 4, 8, 12]                           # cv::DMatch
    assert p.itemsize == 32 and [p.fields[n][1] for n in ("x", "y", "z", "b", "g", "r", "a", "label")] == [0, 4, 8, 16, 17, 18, 19, 20]             # pcl::PointXYZRGBL
    assert v.itemsize == 112
    # the C compiler agrees
    code = '#include "ssm_hip.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu %zu %zu", sizeof(ssm_keypoint), sizeof(ssm_dmatch), sizeof(ssm_point), sizeof(ssm_voxel), sizeof(ssm_camera), sizeof(ssm_config));return 0;}'
    exe = "/tmp/ssm_layout_check"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-x", "c", "-", "-o", exe], input=code, text=True, check=True)
    sizes = subprocess.run([exe], capture_output=True, text=True).stdout.split()
    assert sizes[:5] == ["28", "16", "32", "112", "40"]
    from semantic_slam_mapping_amd._lib import Config
    assert int(sizes[5]) == ctypes.sizeof(Config)


def test_config_default_is_parameters_txt():
    import semantic_slam_mapping_amd as ssm
    c = ssm.default_config()
    assert (c.orb_features, c.orb_levels, c.orb_iniThFAST, c.orb_minThFAST, c.tracker_ref_frames) == (2000, 8, 20, 7, 5)   # parameters.txt:66-71,81
    assert abs(c.orb_scale - 1.2) < 1e-6 and c.knn_match_ratio == 0.8 and c.mapper_resolution == 0.1 and c.mapper_max_distance == 40
    assert c.camera.scale == 1000.0
    with pytest.raises(KeyError):
        ssm.default_config(no_such_key=1)


def test_frame_block_partition():
    from semantic_slam_mapping_amd.sharding import frame_block
    for n, w in ((1000, 8), (10000, 8), (7, 3), (5, 8), (0, 2)):
        blocks = [frame_block(n, r, w) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in blocks]
        assert max(sizes) - min(sizes) <= 1
