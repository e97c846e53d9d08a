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


# kernels whose software prefetch a register spill silently turns into a synchronous load (a reload is followed by s_waitcnt vmcnt(0): DESIGN.md s.4.4) -- the hot
# instantiations of the bench's configurations, by mangled-name prefix
_NO_SPILL = {
    "kernels_sgbm.hip": ["_Z11sgbm_sweep8ILi5ELb1E", "_Z10sgbm_rows8ILi12ELb1E", "_Z20sgbm_cost_reg_kernelILi80ELi5ELi32ELi6E"],
    "kernels_orb.hip": ["_Z16blur_mfma_kernel", "_Z11fast_kernel", "_Z13orient_kernel", "_Z12brief_kernel", "_Z14resize4_kernel"],
    "kernels_map.hip": ["_Z18map_stream2_kernelILb1ELb0E"],
}


def _resource_usage(src):
    import re
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc"
    csrc = os.path.join(ROOT, "semantic_slam_mapping_amd", "csrc")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                        "-c", os.path.join(csrc, src), "-o", os.devnull], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out, name = {}, None
    for ln in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            name = m.group(1)
            out[name] = {}
        m = re.search(r"remark:\s+(VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]|VGPRs): (\d+)", ln)
        if m and name:
            out[name][m.group(1)] = int(m.group(2))
    return out


def test_hot_kernels_do_not_spill_registers():
    """hipcc cross-compiles without a GPU; ~2 minutes for the three files (in parallel)"""
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(3) as ex:
        usage = dict(zip(_NO_SPILL, ex.map(_resource_usage, _NO_SPILL)))
    for src, prefixes in _NO_SPILL.items():
        for p in prefixes:
            hit = [k for k in usage[src] if k.startswith(p)]
            assert hit, (src, p, sorted(usage[src])[:5])
            for k in hit:
                assert usage[src][k].get("VGPRs Spill", 0) == 0 and usage[src][k].get("ScratchSize [bytes/lane]", 0) == 0, (k, usage[src][k])
