"""tools/gen_blobs.py — regenerate qm_control_amd/data/{model_blob,settings_blob}.npy and gaits.json
from the reference's input files (robot.urdf, task.info, reference.info, gait.info).

Runs only where /root/reference exists (this container); the GPU box uses the committed outputs.
The numbers are produced by the independent numpy front-end oracle/front.py; tests/test_host_parsers.py
checks that the product's own C++ parsers (qm_control_amd/csrc/host) reproduce them from the same files.
"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import front

REF = os.environ.get("QM_REFERENCE", "/root/reference")
URDF = os.path.join(REF, "qm_description/urdf/qudraputed_manipulator/robot.urdf")
TASK = os.path.join(REF, "qm_controllers/config/task.info")
REFI = os.path.join(REF, "qm_controllers/config/reference.info")
GAIT = os.path.join(REF, "qm_controllers/config/gait.info")

def main():
    mb, names = front.build_model(URDF, REFI)
    st = front.build_settings(TASK, mb)
    out = os.path.join(ROOT, "qm_control_amd", "data")
    np.save(os.path.join(out, "model_blob.npy"), mb)
    np.save(os.path.join(out, "settings_blob.npy"), st)
    g = front.parse_info(GAIT)
    gaits = {}
    for name in front.info_list(g["list"]):
        times, modes = front.load_gait(GAIT, name)
        gaits[name] = dict(switchingTimes=times, modeSequence=modes)
    json.dump(dict(joint_names=names, gaits=gaits), open(os.path.join(out, "gaits.json"), "w"), indent=1)
    print("wrote blobs:", mb.shape, st.shape, list(gaits))

if __name__ == "__main__":
    main()
