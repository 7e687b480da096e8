"""Named offsets of the MODEL / SETTINGS blobs: the #defines of include/qmhip_layout.h, read from the header itself
(one source of truth; e.g. layout.ST_SQP_ITER, layout.MB_ROBOTMASS, layout.QM_NX)."""
import os
import re

_HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "qmhip_layout.h")
DEFINES = {}
with open(_HDR) as _fh:
    for _line in _fh:
        _m = re.match(r"#define\s+((?:MB|ST|QM)_\w+)\s+(\d+)\s", _line)
        if _m:
            DEFINES[_m.group(1)] = int(_m.group(2))
            continue
        _m = re.match(r"#define\s+(QM_\w+)\s+(\d+\.\d*(?:[eE][-+]?\d+)?)\s", _line)      # floating-point constants (QM_GRID_DT_MIN_*)
        if _m:
            DEFINES[_m.group(1)] = float(_m.group(2))
globals().update(DEFINES)
