"""Register / LDS budgets the schedule of the control step relies on, read from the built library's gfx950 code object (no GPU needed).
These are the numbers behind DESIGN.md's occupancy notes; crossing an allocation granule here has cost more than most kernel changes gained:
  * a WBC wavefront holds ceil(vgpr / 8) * 8 of its SIMD's 512 registers while the next step's grid kernels run on the other stream — they must fit beside it;
  * the one-wave-per-instance kernels and the LQ kernel must not touch the private segment (rocprofv3's ScratchBytesPerLane);
  * the two product instances of the LQ kernel run THREE waves per SIMD (round 4: 168 registers, no scratch — any scratch doubles the time of a launch of 110 k
    one-wave workgroups — and 13.1 KB of LDS: twelve waves per CU since round 6), the thread-per-node kernels four waves per CU (their LDS rows decide that)."""
import os, re, struct, subprocess, tempfile, ctypes as C
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "qm_control_amd", "libqmhip.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def _code_objects():
    """the gfx950 code object of every translation unit of the library (one offload bundle each: qmhip.hip, qmhip_lq.hip — qm_control_amd/build_flags.py)"""
    blob = open(LIB, "rb").read(); cos = []; i = blob.find(b"__CLANG_OFFLOAD_BUNDLE__"); assert i >= 0, "no offload bundle in libqmhip.so"
    while i >= 0:
        n = struct.unpack_from("<Q", blob, i + 24)[0]; p = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p); p += 24; triple = blob[p:p + tl].decode(); p += tl
            if "gfx950" in triple and size > 0: cos.append(blob[i + off:i + off + size])
        i = blob.find(b"__CLANG_OFFLOAD_BUNDLE__", i + 1)
    assert cos, "no gfx950 code object"
    return cos


def _readelf(args):
    out = ""
    for co in _code_objects():
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(co); f.flush(); out += subprocess.run([READELF] + args + [f.name], capture_output=True, text=True, check=True).stdout
    return out


def _code_bytes():
    """machine-code size of every kernel (symbol table of the gfx950 code object)"""
    syms = _readelf(["-s", "--wide"])
    out = {}
    for line in syms.splitlines():
        m = re.search(r"\s(\d+)\s+FUNC\s+\S+\s+\S+\s+\S+\s+_Z\d+(qm_\w+_kernel)\w*$", line)
        if m: out[m.group(2)] = max(int(m.group(1)), out.get(m.group(2), 0))
    return out


def _kernels():
    notes = _readelf(["--notes"])
    out = {}
    for blk in notes.split("  - .agpr_count:")[1:]:
        g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
        name = re.search(r"\.name:\s+_Z\d+(qm_\w+_kernel)", blk)
        if name:      # (template instances of one kernel — qm_ls_eval_kernel_t<true / false> — share a name: the larger figures count; each instance is also kept under its mangled name)
            cur = dict(vgpr=g("vgpr_count"), lds=g("group_segment_fixed_size"), scratch=g("private_segment_fixed_size")); old = out.get(name.group(1), cur)
            out[name.group(1)] = {q: max(cur[q], old[q]) for q in cur}
            out[re.search(r"\.name:\s+(\S+)", blk).group(1)] = cur
    return out


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(READELF)), reason="libqmhip.so / llvm-readelf not available")
def test_register_and_scratch_budgets():
    k = _kernels(); alloc = lambda name: (k[name]["vgpr"] + 7) // 8 * 8
    # (qm_lq_ipm_kernel / qm_lq_dbg_kernel are instances of the K1b body at two waves per SIMD: their 32 B came from ONE private array indexed by a run-time value in the
    # divergent Jacobian columns, qm_dev_kin.h; qm_target_kernel was compiled for 1024-thread workgroups — 128 registers, 212 B — and is launched with 64)
    for name in ("qm_wbc_kernel", "qm_sim_kernel", "qm_riccati_kernel", "qm_lq_kernel", "qm_lq_m18_kernel", "qm_lq_ipm_kernel", "qm_lq_dbg_kernel", "qm_target_kernel", "qm_ilqr_rollout_kernel", "qm_hoqp_kernel"):
        assert k[name]["scratch"] == 0, (name, k[name])
    assert alloc("qm_wbc_kernel") + alloc("qm_grid_nodes_kernel") <= 512 and alloc("qm_wbc_kernel") + alloc("qm_grid_kernel") <= 512, (k["qm_wbc_kernel"], k["qm_grid_nodes_kernel"])
    assert alloc("qm_wbc_kernel") + alloc("qm_policy_kernel") <= 512
    assert 3 * alloc("qm_lq_kernel") <= 512 and 3 * alloc("qm_lq_m18_kernel") <= 512, (k["qm_lq_kernel"], k["qm_lq_m18_kernel"])
    # the thread-per-node kernels are capped at 256 registers (two waves per SIMD: every wavefront of the benchmark launch resident at once).  What does not fit is a handful
    # of spill stores / reloads among ~ 10 k instructions; round 5 (no inlined library sin / cos behind a never-taken branch: K1a 27 k -> 7.8 k instructions, K4 39.8 k -> 12.5 k;
    # the wave's rows moved together through LDS) keeps them at <= 160 / <= 224 B per lane (round 4: 108 / 352 B with 2504 scalar-register spills in K4)
    # Round 6, second half: built without the IR-level load/store vectorizer (qm_control_amd/build_flags.py) K1a no longer spills at all (100 B -> 0: the pass had been turning its
    # register-resident rows into 16-byte values); the line search's PRODUCT instance (block-diagonal R0, SQP) is at 220 B, the dense-R0 instance nobody ships at 328 B
    assert alloc("qm_lq_kin_kernel") <= 256 and alloc("qm_ls_eval_kernel") <= 256
    assert k["qm_lq_kin_kernel"]["scratch"] == 0, k["qm_lq_kin_kernel"]
    assert k["_Z19qm_ls_eval_kernel_tILb1ELb0EEv8QmLsArgs"]["scratch"] <= 224 and k["qm_ls_eval_kernel"]["scratch"] <= 336, {n: v for n, v in k.items() if "ls_eval" in n}


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(READELF)), reason="libqmhip.so / llvm-readelf not available")
def test_code_size_of_the_large_kernels():
    """A wave of the LQ kernel executes nearly all of its code once (few loops), so its code size IS its instruction count: 30.7 KB / 31.8 KB for the two product instances
    since the lane's choice between the two stages' kinematics arrays is one opaque base (QM_LANE_OPAQUE) — written as `fs ? K2 : K1` the compiler selects between two
    literal LDS addresses at each of the 66 accesses behind it (+ 1.8 KB, + 1.2 % of the kernel's time, profiles/r05_ab_lane_base.log).  Round 6: 30.2 / 31.4 KB INCLUDING the dense
    R0 (u - u_nom) product a wave no longer executes with the shipped (block-diagonal) input weight (≈ 1 KB; `tools/isa_hist.py` counts the executed stream: 4.84 k instructions,
    round 5: 5.12 k static + ≈ 0.25 k in the rolled zero-fill loops that are now eight unrolled 16-byte stores).  The bounds leave ~ 3 % of room."""
    b = _code_bytes()
    assert b["qm_lq_kernel"] <= 31200 and b["qm_lq_m18_kernel"] <= 32300, (b["qm_lq_kernel"], b["qm_lq_m18_kernel"])
    assert b["qm_riccati_kernel"] <= 45000 and b["qm_wbc_kernel"] <= 185000, (b["qm_riccati_kernel"], b["qm_wbc_kernel"])     # (the WBC exceeds the 64 KB instruction cache by design: DESIGN.md section 7)


def test_lds_budgets_fit_the_intended_waves_per_cu():
    import emu_harness
    lib = C.CDLL(emu_harness.build())
    cu = 160 * 1024
    lq, ric, kin, ev, wbc, sim = (lib.emu_sizes(i) for i in (3, 4, 6, 7, 8, 9))
    assert 12 * ((lq + 255) // 256 * 256) <= cu and 4 * ric <= cu and 7 * kin <= cu and 8 * ev <= cu and 4 * wbc <= cu and 4 * sim <= cu, (lq, ric, kin, ev, wbc, sim)      # (K1a: seven waves per CU hold the benchmark launch's 6.45 per CU; K1b: TWELVE since round 6 — 13440 B per wave, the register file's three per SIMD — worth 11 % of the kernel against ten)


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(OBJDUMP)), reason="libqmhip.so / llvm-objdump not available")
def test_no_paired_lds_accesses_in_the_lds_heavy_kernels():
    """gfx950 executes ds_read2_b64 / ds_write2_b64 at 1.6 x the LDS cycles of the two single accesses (tools/probes/lds_width_probe.hip, DESIGN.md section 7.0 "LDS pairs"); the
    compiler forms them from neighbouring 8-byte accesses unless the build switches the IR-level vectorizer off (qm_control_amd/build_flags.py) and the kernel carries
    QM_UNPAIRED_LDS.  A library built without either is correct and ~ 4 % slower per step: this is the guard."""
    pairs = {}
    for co in _code_objects():
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(co); f.flush(); dis = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True, check=True).stdout
        cur = None
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <_Z\d+(qm_\w+_kernel)\w*>:", line)
            if m: cur = m.group(1); pairs.setdefault(cur, 0); continue
            if cur and re.search(r"\bds_(read|write)2(st64)?_b64\b", line): pairs[cur] += 1
    for name in ("qm_lq_kernel", "qm_lq_m18_kernel", "qm_lq_kin_kernel", "qm_riccati_kernel", "qm_wbc_kernel"):
        assert name in pairs, (name, sorted(pairs))
        assert pairs[name] == 0, (name, pairs[name])

