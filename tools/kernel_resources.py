"""tools/kernel_resources.py — register / LDS / scratch (spill) budget of every gfx950 kernel of libqmhip, from the compiler's own kernel descriptors
(hipcc --save-temps in a temporary directory; no GPU needed).  `scratch` > 0 means the kernel spills to private memory (rocprofv3's ScratchBytesPerLane).
usage: python tools/kernel_resources.py [out.csv]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qm_control_amd.build_flags import HIPCC_FLAGS, DEVICE_UNITS      # the product build's flags


def main(out=None):
    with tempfile.TemporaryDirectory() as d:
        s = ""
        for src, extra in DEVICE_UNITS:      # every translation unit with its own flags (qm_control_amd/build_flags.py)
            stem = os.path.splitext(os.path.basename(src))[0]
            subprocess.check_call(["/opt/rocm/bin/hipcc"] + HIPCC_FLAGS + extra + ["-I" + os.path.join(ROOT, "include"), "--save-temps", "-c", os.path.join(ROOT, src), "-o", os.path.join(d, stem + ".o")],
                                  cwd=d, stderr=subprocess.DEVNULL)
            s += open(os.path.join(d, stem + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    lines = ["kernel,vgpr_total,accum_offset,sgpr,scratch_bytes_per_lane,waves_per_simd"]
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
        body = m.group(2); g = lambda k: int((re.search(r"\.amdhsa_%s (\d+)" % k, body) or [None, "0"])[1])
        name = (re.match(r"_Z\d+(qm_\w+_kernel)", m.group(1)) or [None, m.group(1)])[1]
        v = g("next_free_vgpr"); lines.append("%s,%d,%d,%d,%d,%d" % (name, v, g("accum_offset"), g("next_free_sgpr"), g("private_segment_fixed_size"), max(1, min(8, 512 // max(1, ((v + 7) // 8) * 8)))))
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(*sys.argv[1:])
