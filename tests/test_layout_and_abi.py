"""Layout constants, C-ABI exports and host-only behaviour of libqmhip.so (no GPU needed)."""
import ctypes as C
import os
import re
import numpy as np
import pytest
from conftest import ROOT, REFERENCE


def _defines(path):
    d = {}
    for line in open(path):
        m = re.match(r"#define\s+(\w+)\s+(\d+)\b", line)
        if m:
            d[m.group(1)] = int(m.group(2))
    return d


def test_front_layout_matches_header():
    import front
    d = _defines(os.path.join(ROOT, "include", "qmhip_layout.h"))
    for k, v in front.MB.items():
        assert d["MB_" + k] == v, k
    for k, v in front.ST.items():
        assert d["ST_" + k] == v, k
    assert d["QM_NX"] == 30 and d["QM_NU"] == 30 and d["QM_NREF"] == 37 and d["QM_NRBD"] == 55


def test_library_exports_every_declared_symbol():
    from qm_control_amd import api
    hdr = open(os.path.join(ROOT, "include", "qmhip.h")).read()
    declared = sorted(set(re.findall(r"\b(qmhip_\w+)\s*\(", hdr)))
    assert len(declared) >= 20
    lib = api.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(api.EXPORTS) == [n for n in declared if n in api.EXPORTS]
    assert set(api.EXPORTS) == set(declared)


def test_no_cpu_fallback_without_device(blobs):
    """the product path must fail loudly when no HIP device is present"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from qm_control_amd import api
    with pytest.raises(api.QmhipError):
        api.QMInterface(blobs=blobs, max_batch=1, max_nodes=16)


def test_missing_files_raise_like_the_reference():
    from qm_control_amd import api
    with pytest.raises(ValueError, match="Task file not found"):
        api.parse_model("/nonexistent/robot.urdf", "/nonexistent/task.info", "/nonexistent/reference.info")


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference inputs not present")
def test_cpp_parsers_match_numpy_front_end(blobs):
    """product C++ URDF/INFO ingestion vs the independent numpy front-end (committed blobs)"""
    from qm_control_amd import api
    mb, st = api.parse_model(REFERENCE + "/qm_description/urdf/qudraputed_manipulator/robot.urdf", REFERENCE + "/qm_controllers/config/task.info", REFERENCE + "/qm_controllers/config/reference.info")
    assert np.abs(mb - blobs[0]).max() <= 1e-14
    assert np.abs(st - blobs[1]).max() <= 1e-14
    assert abs(mb[654] - 27.371574) < 1e-9                         # total mass
    with pytest.raises(ValueError, match="URDF file not found"):
        api.parse_model("/nonexistent/robot.urdf", REFERENCE + "/qm_controllers/config/task.info", REFERENCE + "/qm_controllers/config/reference.info")


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference inputs not present")
def test_committed_blobs_are_current():
    import front
    from qm_control_amd import scenarios
    mb, _ = front.build_model(REFERENCE + "/qm_description/urdf/qudraputed_manipulator/robot.urdf", REFERENCE + "/qm_controllers/config/reference.info")
    st = front.build_settings(REFERENCE + "/qm_controllers/config/task.info", mb)
    cmb, cst = scenarios.load_blobs()
    assert np.array_equal(mb, cmb) and np.array_equal(st, cst)
    times, modes = front.load_gait(REFERENCE + "/qm_controllers/config/gait.info", "trot")
    g = scenarios.load_gaits()["trot"]
    assert times == g["switchingTimes"] and modes == g["modeSequence"]


def test_create_rejects_bad_sizes(blobs):
    """argument validation happens before any device work (qmhip.h: 3 <= max_nodes <= 512)"""
    from qm_control_amd import api
    for bad in (2, 513):
        with pytest.raises(api.QmhipError, match="bad argument"):
            api.QMInterface(blobs=blobs, max_batch=1, max_nodes=bad)
