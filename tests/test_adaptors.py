"""The C++ adaptors (adaptors/*.h: qm::QMInterface-, ocs2::MPC_BASE-, qm::WbcBase-shaped classes over the C ABI) are shipped as SOURCE for the reference's
catkin workspace; here they are parsed and type-checked against stand-in declarations of the OCS2 / ROS / qm_* names they touch (adaptors/stubs)."""
import os
import re
import subprocess
from conftest import ROOT


def test_adaptors_compile_against_stubs():
    p = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "adaptors"),
                        os.path.join(ROOT, "adaptors", "compile_check.cpp")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]


def test_reference_golden_generator_type_checks_against_stubs():
    """adaptors/tools/dump_reference_goldens.cpp — the program a maintainer compiles in the reference's catkin workspace to produce tests/golden_ref — parses and
    type-checks against the stand-in declarations (QMInterface, SqpMpc, GaitSchedule, HierarchicalWbc / HierarchicalMpcWbc, PrimalSolution ...)"""
    p = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-DQMHIP_ADAPTOR_STUBS", "-I" + os.path.join(ROOT, "adaptors"),
                        os.path.join(ROOT, "adaptors", "tools", "dump_reference_goldens.cpp")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]


def test_adaptors_only_call_declared_entry_points():
    hdr = open(os.path.join(ROOT, "include", "qmhip.h")).read()
    declared = set(re.findall(r"\b(qmhip_\w+)\s*\(", hdr))
    used = set()
    for f in ("QmhipInterface.h", "QmhipMpc.h", "QmhipWbc.h", "QmhipController.h"):
        used |= set(re.findall(r"\b(qmhip_\w+)\s*\(", open(os.path.join(ROOT, "adaptors", f)).read()))
    assert used and used <= declared, used - declared
    assert {"qmhip_create", "qmhip_mpc_upload", "qmhip_mpc_update_references", "qmhip_mpc_solve_resident_warm", "qmhip_mpc_download", "qmhip_wbc_step"} <= used
