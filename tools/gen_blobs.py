"""tools/gen_blobs.py — regenerate qm_control_amd/data/{model_blob,settings_blob}.npy and gaits.json
from the reference's input files (robot.urdf, task.info, reference.info, gait.info).

The PRODUCT's blobs (qm_control_amd/data) are produced by the product's own C++ ingestion — qmhip_parse_model of libqmhip.so
(qm_control_amd/csrc/host/qm_model_io.cpp) — from tests/data/{robot.urdf, task.info, reference.info}.
The independent numpy front-end oracle/front.py writes the ORACLE's blobs (oracle/data) from the same files and is the checker:
tests/test_layout_and_abi.py holds the two within 1e-13 of each other and both bit-equal to a fresh parse.
gaits.json needs gait.info, i.e. /root/reference (this container); it is skipped where that is absent.
"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import front

sys.path.insert(0, ROOT)
REF = os.environ.get("QM_REFERENCE", "/root/reference")
DATA = os.path.join(ROOT, "tests", "data")
URDF = os.path.join(DATA, "robot.urdf")
TASK = os.path.join(DATA, "task.info")
REFI = os.path.join(DATA, "reference.info")
GAIT = os.path.join(REF, "qm_controllers/config/gait.info")

def main():
    from qm_control_amd import api
    pmb, pst = api.parse_model(URDF, TASK, REFI)                 # product parser -> product blobs
    out = os.path.join(ROOT, "qm_control_amd", "data")
    np.save(os.path.join(out, "model_blob.npy"), pmb)
    np.save(os.path.join(out, "settings_blob.npy"), pst)
    mb, names = front.build_model(URDF, REFI)                    # numpy front-end -> oracle blobs
    st = front.build_settings(TASK, mb)
    oout = os.path.join(ROOT, "oracle", "data"); os.makedirs(oout, exist_ok=True)
    np.save(os.path.join(oout, "model_blob.npy"), mb)
    np.save(os.path.join(oout, "settings_blob.npy"), st)
    print("product vs front-end: model %.1e settings %.1e" % (np.abs(pmb - mb).max(), np.abs(pst - st).max()))
    if not os.path.exists(GAIT):
        print("gait.info not present: gaits.json left as committed"); return
    g = front.parse_info(GAIT)
    gaits = {}
    for name in front.info_list(g["list"]):
        times, modes = front.load_gait(GAIT, name)
        gaits[name] = dict(switchingTimes=times, modeSequence=modes)
    json.dump(dict(joint_names=names, gaits=gaits), open(os.path.join(out, "gaits.json"), "w"), indent=1)
    print("wrote blobs:", mb.shape, st.shape, list(gaits))

if __name__ == "__main__":
    main()
