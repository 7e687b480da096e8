"""Compiler flags of the device library — ONE list for the product build (__graft_entry__.build) and for every tool that compiles the kernels to look at them
(tools/kernel_resources.py, tools/isa_hist.py, tests/test_kernel_budgets.py through the former).

`-mllvm -amdgpu-load-store-vectorizer=0`: the IR-level vectorizer turns two neighbouring 8-byte LDS accesses into one 16-byte access of 8-byte alignment, which gfx950 executes
as ds_read2_b64 / ds_write2_b64 — 1.6 x the LDS cycles of the two single accesses (tools/probes/lds_width_probe.hip).  Every kernel of this library works on doubles in LDS;
none relies on the pass for its global accesses (the wide ones are written as double2).  Measured on one box (profiles/r06_ab_lds_pairing.log): WBC - 5 %, K3 - 1 %, K1a - 6 %,
nothing slower.  The machine-level pass that forms the same pairs later is switched off per kernel (QM_UNPAIRED_LDS in qm_dev_common.h) where that was measured to pay.
"""
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-Wno-unused-result", "-mllvm", "-amdgpu-load-store-vectorizer=0"]

# Translation units of the device library: (source relative to the repo root, extra flags).  K1b's instances are compiled with the max-ILP scheduling strategy
# (qm_control_amd/csrc/host/qmhip_lq.hip says why); everything else with the default one.
DEVICE_UNITS = [("qm_control_amd/csrc/host/qmhip.hip", []),
                ("qm_control_amd/csrc/host/qmhip_lq.hip", ["-mllvm", "-amdgpu-sched-strategy=max-ilp"])]
HOST_UNITS = ["qm_control_amd/csrc/host/qm_model_io.cpp"]
