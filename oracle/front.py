"""oracle/front.py — TEST INFRASTRUCTURE (CPU oracle front-end), not product code.

Independent (numpy + xml.etree) ingestion of the reference's three input files into the flat
MODEL / SETTINGS blobs of include/qmhip_layout.h, plus small numpy kinematics used to pin the
oracle against the known answers of SURVEY.md §8(c).  The product parses the same files with its
own C++ code (qm_control_amd/csrc/host); tests compare the two blobs.

PARITY UNPINNED: the reference ships no tests / golden vectors and none of its dependencies
(OCS2, Pinocchio, CppAD, HPIPM, qpOASES) exist here, so this restates the published algorithms.

Reference anchors:
  model build      qm_interface/src/QMInterface.cpp:408-416 (createPinocchioInterface +
                   createCentroidalModelInfo [upstream ocs2_centroidal_model/FactoryFunctions])
  joint ordering   urdfdom sorts child joints by name; Pinocchio visits depth-first
                   -> LF, LH, RF, RH, arm  (qm_controllers/config/task.info:168-188)
  settings         qm_controllers/config/task.info, reference.info; qm_wbc/cfg/wbcWigeht.cfg:7-47
  R transform      qm_interface/src/QMInterface.cpp:274-299
"""
import re
import xml.etree.ElementTree as ET
import numpy as np

# ---- layout constants (mirror of include/qmhip_layout.h; checked by tests/test_layout.py) ----
NJ, NB, NQ, NX, NU, NF = 18, 19, 24, 30, 30, 5
MB = dict(PARENT=0, JR=18, JP=180, AXIS=234, QLO=288, QHI=306, TAUMAX=324, MASS=342, COM=361,
          INERTIA=418, FPARENT=589, FR=594, FP=639, ROBOTMASS=654, INOM=655, RNOM=664, QNOM=667,
          SIZE=685)
ST = dict(Q=0, R=30, XINIT=930, MU_EE_POS=960, MU_EE_ORI=961, MU_EEF_POS=962, MU_EEF_ORI=963,
          FRIC_COEF=964, FRIC_MU=965, FRIC_DELTA=966, FRIC_REG=967, FRIC_SHIFT=968, JPOS_MU=969,
          JPOS_DELTA=970, JVEL_MU=971, JVEL_DELTA=972, JVEL_LO=973, JVEL_HI=979, POS_ERR_GAIN=985,
          PHASE_TRANS_STANCE=986, LIFTOFF_VEL=987, TOUCHDOWN_VEL=988, SWING_HEIGHT=989,
          SWING_TIME_SCALE=990, SQP_DT=991, SQP_ITER=992, DELTA_TOL=993, G_MAX=994, G_MIN=995,
          TIME_HORIZON=996, WBC_FRIC=997, KP_SWING=998, KD_SWING=999, KP_BASE_H=1000,
          KD_BASE_H=1001, KP_BASE_LIN=1002, KD_BASE_LIN=1003, KP_BASE_ANG=1004, KD_BASE_ANG=1005,
          KP_ARM_J=1006, KD_ARM_J=1012, KP_EE_LIN=1018, KD_EE_LIN=1021, KP_EE_ANG=1024,
          KD_EE_ANG=1027, SOLVER=1030, DDP_MIN_STEP=1031, DDP_MAX_STEP=1032, DDP_PENALTY=1033,
          IPM_DT=1034, IPM_ITER=1035, IPM_DELTA_TOL=1036, IPM_G_MAX=1037, IPM_G_MIN=1038, IPM_MU=1039, GRID_DT_MIN=1040, RICCATI_STRICT=1041,
          IPM_MU_TARGET=1042, IPM_MU_LINEAR=1043, IPM_MU_POWER=1044, IPM_RED_COST_TOL=1045, IPM_RED_CON_TOL=1046, IPM_FTB_MARGIN=1047, IPM_PRIMAL_FOR_DUAL=1048,
          IPM_SLACK_LB=1049, IPM_DUAL_LB=1050, IPM_SLACK_MARGIN=1051, IPM_DUAL_MARGIN=1052, SIZE=1056)

FOOT_FRAMES = ["LF_FOOT", "RF_FOOT", "LH_FOOT", "RH_FOOT"]   # ModelSettings.h:38 (contact order)
MODE_NAMES = {"FLY": 0, "RH": 1, "LH": 2, "LH_RH": 3, "RF": 4, "RF_RH": 5, "RF_LH": 6,
              "RF_LH_RH": 7, "LF": 8, "LF_RH": 9, "LF_LH": 10, "LF_LH_RH": 11, "LF_RF": 12,
              "LF_RF_RH": 13, "LF_RF_LH": 14, "STANCE": 15}


# ------------------------------------------------------------------------------------------------
# Boost-INFO subset parser (nested { } blocks, "key value" lines, ';' comments)
# ------------------------------------------------------------------------------------------------
def parse_info(path):
    """INFO is line based: "key value", "key" followed by "{" (same or next line), "}"."""
    root = {}
    stack = [root]
    pending = None
    with open(path) as fh:
        for raw in fh:
            line = raw.split(';')[0]
            line = re.sub(r'//.*', '', line).strip()
            if not line:
                continue
            parts = re.findall(r'"[^"]*"|[{}]|[^\s{}]+', line)
            j = 0
            while j < len(parts):
                p = parts[j]
                if p == '{':
                    sub = {}
                    stack[-1][pending] = sub
                    stack.append(sub)
                    pending = None
                    j += 1
                elif p == '}':
                    stack.pop()
                    pending = None
                    j += 1
                else:
                    key = p
                    if j + 1 < len(parts) and parts[j + 1] not in '{}':
                        stack[-1][key] = parts[j + 1].strip('"')
                        pending = None
                        j += 2
                    else:
                        pending = key
                        stack[-1].setdefault(key, '')
                        j += 1
    return root


def info_get(tree, dotted):
    node = tree
    for k in dotted.split('.'):
        node = node[k]
    return node


def info_matrix(tree, name, rows, cols):
    """ocs2 loadData::loadEigenMatrix: entries "(i,j) v", optional 'scaling', default 0."""
    node = info_get(tree, name)
    m = np.zeros((rows, cols))
    scaling = float(node.get('scaling', 1.0))
    for k, v in node.items():
        mm = re.match(r'\((\d+),(\d+)\)', k)
        if mm:
            m[int(mm.group(1)), int(mm.group(2))] = float(v)
    return m * scaling


def info_list(node):
    """entries "[i] value" -> list ordered by i"""
    items = []
    for k, v in node.items():
        mm = re.match(r'\[(\d+)\]', k)
        if mm:
            items.append((int(mm.group(1)), v))
    return [v for _, v in sorted(items)]


# ------------------------------------------------------------------------------------------------
# rotations
# ------------------------------------------------------------------------------------------------
def rpy_to_R(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def zyx_to_R(zyx):
    """R = Rz(z) Ry(y) Rx(x) — ocs2 getRotationMatrixFromZyxEulerAngles"""
    return rpy_to_R([zyx[2], zyx[1], zyx[0]])


def axis_angle_R(axis, q):
    a = np.asarray(axis, float)
    K = skew(a)
    return np.eye(3) + np.sin(q) * K + (1 - np.cos(q)) * (K @ K)


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


# ------------------------------------------------------------------------------------------------
# URDF -> model (Pinocchio semantics: fixed joints merged, frames kept)
# ------------------------------------------------------------------------------------------------
def _origin(el):
    o = el.find('origin') if el is not None else None
    if o is None:
        return np.eye(3), np.zeros(3)
    xyz = np.array([float(s) for s in o.get('xyz', '0 0 0').split()])
    rpy = np.array([float(s) for s in o.get('rpy', '0 0 0').split()])
    return rpy_to_R(rpy), xyz


def build_model(urdf_path, reference_info_path, ee_frame="j2n6s300_end_effector"):
    root = ET.parse(urdf_path).getroot()
    links = {l.get('name'): l for l in root.findall('link')}
    joints = {j.get('name'): j for j in root.findall('joint')}
    child_links = {j.find('child').get('link') for j in joints.values()}
    root_link = [n for n in links if n not in child_links]
    assert len(root_link) == 1
    root_link = root_link[0]
    # urdfdom: joints kept in a std::map (sorted by name) -> child order by joint name
    children = {}
    for jn in sorted(joints):
        j = joints[jn]
        children.setdefault(j.find('parent').get('link'), []).append(jn)

    def link_inertia(name):
        ine = links[name].find('inertial')
        if ine is None:
            return 0.0, np.zeros(3), np.zeros((3, 3))
        R, c = _origin(ine)
        m = float(ine.find('mass').get('value'))
        i = ine.find('inertia')
        I = np.array([[float(i.get('ixx')), float(i.get('ixy')), float(i.get('ixz'))],
                      [float(i.get('ixy')), float(i.get('iyy')), float(i.get('iyz'))],
                      [float(i.get('ixz')), float(i.get('iyz')), float(i.get('izz'))]])
        return m, c, R @ I @ R.T

    bodies = []     # dict(mass, mc (first moment), Io (inertia about body origin))
    jinfo = []      # per movable joint
    frames = {}     # name -> (body, R, p)

    def add_inertia(body, m, c, Ic, R, p):
        # link inertia (about its COM c, in link axes) placed at (R,p) in the body frame
        cb = R @ c + p
        Ib = R @ Ic @ R.T
        b = bodies[body]
        b['m'] += m
        b['mc'] += m * cb
        b['Io'] += Ib + m * (cb @ cb * np.eye(3) - np.outer(cb, cb))

    def visit(link, body, R, p):
        """link rigidly attached to `body` with placement (R,p)"""
        frames[link] = (body, R.copy(), p.copy())
        m, c, Ic = link_inertia(link)
        add_inertia(body, m, c, Ic, R, p)
        for jn in children.get(link, []):
            j = joints[jn]
            Rj, pj = _origin(j)
            Rj, pj = R @ Rj, R @ pj + p
            child = j.find('child').get('link')
            if j.get('type') == 'fixed':
                visit(child, body, Rj, pj)
            else:
                assert j.get('type') in ('revolute', 'continuous')
                ax = np.array([float(s) for s in j.find('axis').get('xyz').split()])
                lim = j.find('limit')
                bodies.append(dict(m=0.0, mc=np.zeros(3), Io=np.zeros((3, 3))))
                nb = len(bodies) - 1
                jinfo.append(dict(name=jn, parent=body, R=Rj, p=pj, axis=ax,
                                  lo=float(lim.get('lower')), hi=float(lim.get('upper')),
                                  effort=float(lim.get('effort'))))
                visit(child, nb, np.eye(3), np.zeros(3))

    bodies.append(dict(m=0.0, mc=np.zeros(3), Io=np.zeros((3, 3))))
    visit(root_link, 0, np.eye(3), np.zeros(3))
    assert len(jinfo) == NJ and len(bodies) == NB, (len(jinfo), len(bodies))

    blob = np.zeros(MB['SIZE'])
    for k, j in enumerate(jinfo):
        blob[MB['PARENT'] + k] = j['parent']
        blob[MB['JR'] + 9 * k: MB['JR'] + 9 * k + 9] = j['R'].ravel()
        blob[MB['JP'] + 3 * k: MB['JP'] + 3 * k + 3] = j['p']
        blob[MB['AXIS'] + 3 * k: MB['AXIS'] + 3 * k + 3] = j['axis']
        blob[MB['QLO'] + k] = j['lo']
        blob[MB['QHI'] + k] = j['hi']
        blob[MB['TAUMAX'] + k] = j['effort']
    for b, bd in enumerate(bodies):
        m = bd['m']
        c = bd['mc'] / m
        Ic = bd['Io'] - m * (c @ c * np.eye(3) - np.outer(c, c))
        blob[MB['MASS'] + b] = m
        blob[MB['COM'] + 3 * b: MB['COM'] + 3 * b + 3] = c
        blob[MB['INERTIA'] + 9 * b: MB['INERTIA'] + 9 * b + 9] = Ic.ravel()
    for f, name in enumerate(FOOT_FRAMES + [ee_frame]):
        body, R, p = frames[name]
        blob[MB['FPARENT'] + f] = body
        blob[MB['FR'] + 9 * f: MB['FR'] + 9 * f + 9] = R.ravel()
        blob[MB['FP'] + 3 * f: MB['FP'] + 3 * f + 3] = p

    ref = parse_info(reference_info_path)
    qnom = info_matrix(ref, 'defaultJointState', NJ, 1)[:, 0]
    blob[MB['QNOM']: MB['QNOM'] + NJ] = qnom

    # createCentroidalModelInfo (SRBD): ccrba at q = [0_6; qnom], v = 0
    q = np.concatenate([np.zeros(6), qnom])
    kin = forward_kinematics(blob, q)
    mass = sum(kin['mass'])
    com = sum(m * c for m, c in zip(kin['mass'], kin['com_w'])) / mass
    Ig = np.zeros((3, 3))
    for m, c, Iw in zip(kin['mass'], kin['com_w'], kin['I_w']):
        d = c - com
        Ig += Iw + m * (d @ d * np.eye(3) - np.outer(d, d))
    blob[MB['ROBOTMASS']] = mass
    blob[MB['INOM']: MB['INOM'] + 9] = Ig.ravel()
    blob[MB['RNOM']: MB['RNOM'] + 3] = q[:3] - com
    names = [j['name'] for j in jinfo]
    return blob, names


def forward_kinematics(blob, q):
    """numpy FK of the blob model at generalized coords q(24) = [p, zyx, joints]."""
    Rb, pb = zyx_to_R(q[3:6]), q[:3].copy()
    Rw, pw = [Rb], [pb]
    for k in range(NJ):
        par = int(blob[MB['PARENT'] + k])
        Rj = blob[MB['JR'] + 9 * k: MB['JR'] + 9 * k + 9].reshape(3, 3)
        pj = blob[MB['JP'] + 3 * k: MB['JP'] + 3 * k + 3]
        ax = blob[MB['AXIS'] + 3 * k: MB['AXIS'] + 3 * k + 3]
        R = Rw[par] @ Rj @ axis_angle_R(ax, q[6 + k])
        p = pw[par] + Rw[par] @ pj
        Rw.append(R)
        pw.append(p)
    mass, com_w, I_w = [], [], []
    for b in range(NB):
        mass.append(blob[MB['MASS'] + b])
        com_w.append(pw[b] + Rw[b] @ blob[MB['COM'] + 3 * b: MB['COM'] + 3 * b + 3])
        I = blob[MB['INERTIA'] + 9 * b: MB['INERTIA'] + 9 * b + 9].reshape(3, 3)
        I_w.append(Rw[b] @ I @ Rw[b].T)
    fpos, frot = [], []
    for f in range(NF):
        b = int(blob[MB['FPARENT'] + f])
        fr = blob[MB['FR'] + 9 * f: MB['FR'] + 9 * f + 9].reshape(3, 3)
        fp = blob[MB['FP'] + 3 * f: MB['FP'] + 3 * f + 3]
        fpos.append(pw[b] + Rw[b] @ fp)
        frot.append(Rw[b] @ fr)
    return dict(R=Rw, p=pw, mass=mass, com_w=com_w, I_w=I_w, fpos=fpos, frot=frot)


def frame_jacobian_lin(blob, q, f):
    """3x24 LOCAL_WORLD_ALIGNED linear Jacobian of frame f (geometric construction)."""
    kin = forward_kinematics(blob, q)
    p = kin['fpos'][f]
    J = np.zeros((3, NQ))
    J[:, :3] = np.eye(3)
    z, y = q[3], q[4]
    E = np.array([[0, -np.sin(z), np.cos(y) * np.cos(z)],
                  [0, np.cos(z), np.cos(y) * np.sin(z)],
                  [1, 0, -np.sin(y)]])
    for k in range(3):
        J[:, 3 + k] = np.cross(E[:, k], p - q[:3])
    # chain of joints up to the frame's parent body
    b = int(blob[MB['FPARENT'] + f])
    while b > 0:
        k = b - 1
        ax = kin['R'][b] @ blob[MB['AXIS'] + 3 * k: MB['AXIS'] + 3 * k + 3]
        J[:, 6 + k] = np.cross(ax, p - kin['p'][b])
        b = int(blob[MB['PARENT'] + k])
    return J


def mat_to_quat_xyzw(R):
    """branching rotation-matrix -> quaternion (ocs2 matrixToQuaternion), xyzw"""
    if R[2, 2] < 0:
        if R[0, 0] > R[1, 1]:
            t = 1 + R[0, 0] - R[1, 1] - R[2, 2]
            q = [t, R[1, 0] + R[0, 1], R[0, 2] + R[2, 0], R[2, 1] - R[1, 2]]
        else:
            t = 1 - R[0, 0] + R[1, 1] - R[2, 2]
            q = [R[1, 0] + R[0, 1], t, R[2, 1] + R[1, 2], R[0, 2] - R[2, 0]]
    else:
        if R[0, 0] < -R[1, 1]:
            t = 1 - R[0, 0] - R[1, 1] + R[2, 2]
            q = [R[0, 2] + R[2, 0], R[2, 1] + R[1, 2], t, R[1, 0] - R[0, 1]]
        else:
            t = 1 + R[0, 0] + R[1, 1] + R[2, 2]
            q = [R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1], t]
    return np.array(q) * 0.5 / np.sqrt(t)


# ------------------------------------------------------------------------------------------------
# settings blob
# ------------------------------------------------------------------------------------------------
WBC_GAIN_DEFAULTS = dict(  # qm_wbc/cfg/wbcWigeht.cfg:7-47 (dynamic_reconfigure defaults)
    kp_swing=350.0, kd_swing=37.0, baseHeightKp=400.0, baseHeightKd=140.0,
    kp_base_linear=400.0, kd_base_linear=100.0, kp_base_angular=400.0, kd_base_angular=140.0,
    kp_arm_joint=[4000.0, 4200.0, 4000.0, 4000.0, 4200.0, 6000.0], kd_arm_joint=[75.0] * 6,
    kp_ee_linear=[3000.0] * 3, kd_ee_linear=[75.0] * 3,
    kp_ee_angular=[2000.0] * 3, kd_ee_angular=[75.0] * 3)


def build_settings(task_info_path, model_blob):
    t = parse_info(task_info_path)
    s = np.zeros(ST['SIZE'])
    Q = info_matrix(t, 'Q', NX, NX)
    s[ST['Q']: ST['Q'] + NX] = np.diag(Q)
    xinit = info_matrix(t, 'initialState', NX, 1)[:, 0]
    s[ST['XINIT']: ST['XINIT'] + NX] = xinit
    # R: leg joint-velocity block -> Jᵀ R J, J = feet-position Jacobian wrt leg joints at initialState
    Rt = info_matrix(t, 'R', NU, NU)
    q = xinit[6:30]
    J = np.zeros((12, 12))
    for i in range(4):
        J[3 * i: 3 * i + 3, :] = frame_jacobian_lin(model_blob, q, i)[:, 6:18]
    R = Rt.copy()
    R[12:24, 12:24] = J.T @ Rt[12:24, 12:24] @ J
    s[ST['R']: ST['R'] + NU * NU] = R.ravel()
    g = lambda k: float(info_get(t, k))

    def gopt(k, default):
        try:
            v = info_get(t, k)
        except (KeyError, TypeError):
            return float(default)
        return float(v) if v != '' else float(default)
    s[ST['MU_EE_POS']] = g('endEffector.muPosition')
    s[ST['MU_EE_ORI']] = g('endEffector.muOrientation')
    s[ST['MU_EEF_POS']] = g('finalEndEffector.muPosition')
    s[ST['MU_EEF_ORI']] = g('finalEndEffector.muOrientation')
    s[ST['FRIC_COEF']] = g('frictionConeSoftConstraint.frictionCoefficient')
    s[ST['FRIC_MU']] = g('frictionConeSoftConstraint.mu')
    s[ST['FRIC_DELTA']] = g('frictionConeSoftConstraint.delta')
    s[ST['FRIC_REG']] = 25.0
    s[ST['FRIC_SHIFT']] = 1e-6
    s[ST['JPOS_MU']] = g('jointPositionLimits.mu')
    s[ST['JPOS_DELTA']] = g('jointPositionLimits.delta')
    s[ST['JVEL_MU']] = g('jointVelocityLimits.mu')
    s[ST['JVEL_DELTA']] = g('jointVelocityLimits.delta')
    s[ST['JVEL_LO']: ST['JVEL_LO'] + 6] = info_matrix(t, 'jointVelocityLimits.lowerBound.arm', 6, 1)[:, 0]
    s[ST['JVEL_HI']: ST['JVEL_HI'] + 6] = info_matrix(t, 'jointVelocityLimits.upperBound.arm', 6, 1)[:, 0]
    s[ST['POS_ERR_GAIN']] = g('model_settings.positionErrorGain')
    s[ST['PHASE_TRANS_STANCE']] = g('model_settings.phaseTransitionStanceTime')
    s[ST['LIFTOFF_VEL']] = g('swing_trajectory_config.liftOffVelocity')
    s[ST['TOUCHDOWN_VEL']] = g('swing_trajectory_config.touchDownVelocity')
    s[ST['SWING_HEIGHT']] = g('swing_trajectory_config.swingHeight')
    s[ST['SWING_TIME_SCALE']] = g('swing_trajectory_config.swingTimeScale')
    s[ST['SQP_DT']] = g('sqp.dt')
    s[ST['SQP_ITER']] = g('sqp.sqpIteration')
    s[ST['DELTA_TOL']] = g('sqp.deltaTol')
    s[ST['G_MAX']] = g('sqp.g_max')
    s[ST['G_MIN']] = g('sqp.g_min')
    s[ST['TIME_HORIZON']] = g('mpc.timeHorizon')
    s[ST['WBC_FRIC']] = g('frictionConeTask.frictionCoefficient')
    w = WBC_GAIN_DEFAULTS
    s[ST['KP_SWING']], s[ST['KD_SWING']] = w['kp_swing'], w['kd_swing']
    s[ST['KP_BASE_H']], s[ST['KD_BASE_H']] = w['baseHeightKp'], w['baseHeightKd']
    s[ST['KP_BASE_LIN']], s[ST['KD_BASE_LIN']] = w['kp_base_linear'], w['kd_base_linear']
    s[ST['KP_BASE_ANG']], s[ST['KD_BASE_ANG']] = w['kp_base_angular'], w['kd_base_angular']
    s[ST['KP_ARM_J']: ST['KP_ARM_J'] + 6] = w['kp_arm_joint']
    s[ST['KD_ARM_J']: ST['KD_ARM_J'] + 6] = w['kd_arm_joint']
    s[ST['KP_EE_LIN']: ST['KP_EE_LIN'] + 3] = w['kp_ee_linear']
    s[ST['KD_EE_LIN']: ST['KD_EE_LIN'] + 3] = w['kd_ee_linear']
    s[ST['KP_EE_ANG']: ST['KP_EE_ANG'] + 3] = w['kp_ee_angular']
    s[ST['KD_EE_ANG']: ST['KD_EE_ANG'] + 3] = w['kd_ee_angular']
    # discrete iLQR: the controller instantiates SqpMpc whatever `ddp.algorithm` says (QMController.cpp:287-288) -> solver 0
    s[ST['SOLVER']] = 0.0
    # the `ddp` / `ipm` blocks are optional ([upstream, recalled] loadSettings keeps the struct defaults for missing keys); the default solver reads neither
    s[ST['DDP_MIN_STEP']] = gopt('ddp.lineSearch.minStepLength', 0.05)
    s[ST['DDP_MAX_STEP']] = gopt('ddp.lineSearch.maxStepLength', 1.0)
    s[ST['DDP_PENALTY']] = gopt('ddp.constraintPenaltyInitialValue', 2.0)
    s[ST['GRID_DT_MIN']] = 10.0 * 2.220446049250313e-16      # [upstream] timeDiscretizationWithEvents' default dt_min
    # `ipm` block (task.info:94-125, loaded at QMInterface.cpp:72, never instantiated): the multiple-shooting parameter set of solver 2
    for k, key, dflt in (('IPM_DT', 'ipm.dt', 0.01), ('IPM_ITER', 'ipm.ipmIteration', 10.0), ('IPM_DELTA_TOL', 'ipm.deltaTol', 1e-6), ('IPM_G_MAX', 'ipm.g_max', 1e6), ('IPM_G_MIN', 'ipm.g_min', 1e-6),
                         ('IPM_MU', 'ipm.initialBarrierParameter', 1e-2), ('IPM_MU_TARGET', 'ipm.targetBarrierParameter', 1e-4), ('IPM_MU_LINEAR', 'ipm.barrierLinearDecreaseFactor', 0.2),
                         ('IPM_MU_POWER', 'ipm.barrierSuperlinearDecreasePower', 1.5), ('IPM_RED_COST_TOL', 'ipm.barrierReductionCostTol', 1e-3), ('IPM_RED_CON_TOL', 'ipm.barrierReductionConstraintTol', 1e-3),
                         ('IPM_FTB_MARGIN', 'ipm.fractionToBoundaryMargin', 0.995), ('IPM_SLACK_LB', 'ipm.initialSlackLowerBound', 1e-4), ('IPM_DUAL_LB', 'ipm.initialDualLowerBound', 1e-4),
                         ('IPM_SLACK_MARGIN', 'ipm.initialSlackMarginRate', 1e-2), ('IPM_DUAL_MARGIN', 'ipm.initialDualMarginRate', 1e-2)):
        s[ST[k]] = gopt(key, dflt)
    try:
        v = info_get(t, 'ipm.usePrimalStepSizeForDual')
    except (KeyError, TypeError):
        v = ''
    s[ST['IPM_PRIMAL_FOR_DUAL']] = 1.0 if v == '' else (1.0 if str(v) in ('true', '1') else 0.0)        # a boolean key (hard-inequality IPM, solver 3)
    return s


def load_gait(gait_info_path, name):
    """(switchingTimes, modeSequence ids) of one gait template (qm_controllers/config/gait.info)."""
    g = parse_info(gait_info_path)[name]
    modes = [MODE_NAMES[m] for m in info_list(g['modeSequence'])]
    times = [float(v) for v in info_list(g['switchingTimes'])]
    return times, modes
