// qm_model_io.cpp — host-side ingestion of the reference's input files into the MODEL / SETTINGS blobs
// (include/qmhip_layout.h).  Dependency-free C++17: a small XML DOM for the URDF subset and a Boost-INFO
// subset reader.  Reproduces what the reference obtains from urdfdom + Pinocchio + OCS2 loaders:
//   * createPinocchioInterface(urdf, jointNames) with root composite(Translation, SphericalZYX)
//     [upstream], called at qm_interface/src/QMInterface.cpp:410-411: children visited in joint-name order
//     (urdfdom keeps joints in a std::map), depth first; fixed joints merged into their parent body, frames kept
//   * createCentroidalModelInfo (SRBD): total mass, composite inertia about the COM and COM->base offset at
//     q = [0_6, defaultJointState]  (QMInterface.cpp:413-416, reference.info:6-26)
//   * loadData::loadEigenMatrix / loadPtreeValue semantics for task.info (QMInterface.cpp:64-73,147-259,274-319,384-403)
//   * dynamic_reconfigure defaults of qm_wbc/cfg/wbcWigeht.cfg:7-47 as the effective WBC gains (WbcBase.cpp:61-116)
#include "qm_model_io.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <algorithm>

namespace qmio {

// ------------------------------------------------------------------ XML subset
struct XNode { std::string name; std::map<std::string, std::string> attr; std::vector<std::unique_ptr<XNode>> kids;
  const XNode* child(const std::string& n) const { for (auto& k : kids) if (k->name == n) return k.get(); return nullptr; } };

struct XParser {
  const std::string& s; size_t p = 0; std::string err;
  explicit XParser(const std::string& src) : s(src) {}
  void skipWs() { while (p < s.size() && isspace((unsigned char)s[p])) ++p; }
  bool startsWith(const char* t) const { return s.compare(p, strlen(t), t) == 0; }
  void skipMisc() {
    for (;;) {
      skipWs();
      if (startsWith("<!--")) { size_t e = s.find("-->", p); p = (e == std::string::npos) ? s.size() : e + 3; }
      else if (startsWith("<?")) { size_t e = s.find("?>", p); p = (e == std::string::npos) ? s.size() : e + 2; }
      else if (startsWith("<!")) { size_t e = s.find(">", p); p = (e == std::string::npos) ? s.size() : e + 1; }
      else break;
    }
  }
  std::string readName() { size_t b = p; while (p < s.size() && (isalnum((unsigned char)s[p]) || s[p] == '_' || s[p] == ':' || s[p] == '-' || s[p] == '.')) ++p; return s.substr(b, p - b); }
  std::unique_ptr<XNode> element() {
    skipMisc();
    if (p >= s.size() || s[p] != '<') { err = "expected '<'"; return nullptr; }
    ++p; auto n = std::make_unique<XNode>(); n->name = readName();
    for (;;) {
      skipWs();
      if (p >= s.size()) { err = "unterminated tag"; return nullptr; }
      if (s[p] == '/') { p += 2; return n; }
      if (s[p] == '>') { ++p; break; }
      std::string k = readName(); skipWs();
      if (s[p] != '=') { err = "expected '=' in <" + n->name + ">"; return nullptr; }
      ++p; skipWs(); const char q = s[p]; if (q != '"' && q != '\'') { err = "expected quote"; return nullptr; }
      size_t e = s.find(q, p + 1); if (e == std::string::npos) { err = "unterminated attribute"; return nullptr; }
      n->attr[k] = s.substr(p + 1, e - p - 1); p = e + 1;
    }
    for (;;) {   // content
      size_t lt = s.find('<', p); if (lt == std::string::npos) { err = "unterminated element " + n->name; return nullptr; }
      p = lt; skipMisc();
      if (startsWith("</")) { size_t e = s.find('>', p); p = e + 1; return n; }
      auto k = element(); if (!k) return nullptr; n->kids.push_back(std::move(k));
    }
  }
};

// ------------------------------------------------------------------ INFO subset
struct INode { std::string value; std::vector<std::pair<std::string, std::unique_ptr<INode>>> kids;
  const INode* get(const std::string& dotted) const {
    const INode* n = this; size_t b = 0;
    while (n && b <= dotted.size()) { size_t e = dotted.find('.', b); std::string k = dotted.substr(b, e == std::string::npos ? std::string::npos : e - b);
      const INode* nx = nullptr; for (auto& kv : n->kids) if (kv.first == k) { nx = kv.second.get(); break; } n = nx; if (e == std::string::npos) break; b = e + 1; }
    return n; } };

static bool parseInfo(const std::string& path, INode& root, std::string& err) {
  std::ifstream f(path); if (!f) { err = "cannot open " + path; return false; }
  std::vector<INode*> stack{&root}; INode* pending = nullptr; std::string line;
  while (std::getline(f, line)) {
    size_t c = line.find(';'); if (c != std::string::npos) line.erase(c);
    c = line.find("//"); if (c != std::string::npos) line.erase(c);
    std::vector<std::string> tok; size_t i = 0;
    while (i < line.size()) {
      if (isspace((unsigned char)line[i])) { ++i; continue; }
      if (line[i] == '{' || line[i] == '}') { tok.push_back(std::string(1, line[i])); ++i; continue; }
      if (line[i] == '"') { size_t e = line.find('"', i + 1); if (e == std::string::npos) e = line.size(); tok.push_back(line.substr(i + 1, e - i - 1)); i = e + 1; continue; }
      size_t b = i; while (i < line.size() && !isspace((unsigned char)line[i]) && line[i] != '{' && line[i] != '}') ++i; tok.push_back(line.substr(b, i - b));
    }
    for (size_t j = 0; j < tok.size();) {
      if (tok[j] == "{") { if (!pending) { err = "INFO: '{' without key in " + path; return false; } stack.push_back(pending); pending = nullptr; ++j; }
      else if (tok[j] == "}") { if (stack.size() < 2) { err = "INFO: unbalanced '}' in " + path; return false; } stack.pop_back(); pending = nullptr; ++j; }
      else {
        auto n = std::make_unique<INode>(); INode* raw = n.get();
        if (j + 1 < tok.size() && tok[j + 1] != "{" && tok[j + 1] != "}") { n->value = tok[j + 1]; stack.back()->kids.emplace_back(tok[j], std::move(n)); pending = nullptr; j += 2; }
        else { stack.back()->kids.emplace_back(tok[j], std::move(n)); pending = raw; ++j; }
      }
    }
  }
  return true;
}
static bool infoScalar(const INode& root, const std::string& key, double& v, std::string& err) {
  const INode* n = root.get(key); if (!n || n->value.empty()) { err = "INFO: missing key " + key; return false; } v = atof(n->value.c_str()); return true;
}
// loadEigenMatrix: "(i,j) v" entries, optional "scaling", unspecified = 0
static bool infoMatrix(const INode& root, const std::string& key, int rows, int cols, std::vector<double>& m, std::string& err) {
  const INode* n = root.get(key); if (!n) { err = "INFO: missing block " + key; return false; }
  m.assign((size_t)rows * cols, 0.0); double scaling = 1.0;
  for (auto& kv : n->kids) {
    if (kv.first == "scaling") { scaling = atof(kv.second->value.c_str()); continue; }
    int i, j; if (sscanf(kv.first.c_str(), "(%d,%d)", &i, &j) == 2 && i >= 0 && i < rows && j >= 0 && j < cols) m[(size_t)i * cols + j] = atof(kv.second->value.c_str());
  }
  for (auto& v : m) v *= scaling;
  return true;
}

// ------------------------------------------------------------------ small 3-D helpers
struct V3 { double x[3]; };
struct M3 { double m[9]; };
static M3 eye() { return {{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
static M3 mul(const M3& A, const M3& B) { M3 C; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j]; return C; }
static M3 tr(const M3& A) { M3 T; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T.m[3 * i + j] = A.m[3 * j + i]; return T; }
static V3 mulv(const M3& A, const V3& v) { V3 r; for (int i = 0; i < 3; ++i) r.x[i] = A.m[3 * i] * v.x[0] + A.m[3 * i + 1] * v.x[1] + A.m[3 * i + 2] * v.x[2]; return r; }
static V3 add(const V3& a, const V3& b) { return {{a.x[0] + b.x[0], a.x[1] + b.x[1], a.x[2] + b.x[2]}}; }
static V3 sub(const V3& a, const V3& b) { return {{a.x[0] - b.x[0], a.x[1] - b.x[1], a.x[2] - b.x[2]}}; }
static V3 cross(const V3& a, const V3& b) { return {{a.x[1] * b.x[2] - a.x[2] * b.x[1], a.x[2] * b.x[0] - a.x[0] * b.x[2], a.x[0] * b.x[1] - a.x[1] * b.x[0]}}; }
static double dot(const V3& a, const V3& b) { return a.x[0] * b.x[0] + a.x[1] * b.x[1] + a.x[2] * b.x[2]; }
static M3 rpy(double r, double p, double y) {   // R = Rz(y) Ry(p) Rx(r)
  const double cr = cos(r), sr = sin(r), cp = cos(p), sp = sin(p), cy = cos(y), sy = sin(y);
  M3 Rx{{1, 0, 0, 0, cr, -sr, 0, sr, cr}}, Ry{{cp, 0, sp, 0, 1, 0, -sp, 0, cp}}, Rz{{cy, -sy, 0, sy, cy, 0, 0, 0, 1}};
  return mul(mul(Rz, Ry), Rx);
}
static M3 axisAngle(const V3& a, double q) {
  const double s = sin(q), c = cos(q), oc = 1 - c; const double* v = a.x;
  return {{c + oc * v[0] * v[0], oc * v[0] * v[1] - s * v[2], oc * v[0] * v[2] + s * v[1], oc * v[1] * v[0] + s * v[2], c + oc * v[1] * v[1], oc * v[1] * v[2] - s * v[0],
           oc * v[2] * v[0] - s * v[1], oc * v[2] * v[1] + s * v[0], c + oc * v[2] * v[2]}};
}
// parallel-axis term m (|c|² I − c cᵀ)
static void addPointMass(M3& I, double m, const V3& c) { const double cc = dot(c, c); for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) I.m[3 * i + j] += m * ((i == j ? cc : 0.0) - c.x[i] * c.x[j]); }
static bool parse3(const std::string& s, double* v) { return sscanf(s.c_str(), "%lf %lf %lf", v, v + 1, v + 2) == 3; }
static void origin(const XNode* el, M3& R, V3& p) {
  R = eye(); p = {{0, 0, 0}};
  const XNode* o = el ? el->child("origin") : nullptr; if (!o) return;
  double xyz[3] = {0, 0, 0}, a[3] = {0, 0, 0};
  auto ix = o->attr.find("xyz"); if (ix != o->attr.end()) parse3(ix->second, xyz);
  auto ir = o->attr.find("rpy"); if (ir != o->attr.end()) parse3(ir->second, a);
  R = rpy(a[0], a[1], a[2]); p = {{xyz[0], xyz[1], xyz[2]}};
}

struct Body { double m = 0; V3 mc{{0, 0, 0}}; M3 Io{{0, 0, 0, 0, 0, 0, 0, 0, 0}}; };   // mass, first moment, inertia about the body origin
struct JointRec { std::string name; int parent; M3 R; V3 p; V3 axis; double lo, hi, effort; };
struct FrameRec { int body; M3 R; V3 p; };

struct Builder {
  std::map<std::string, const XNode*> links, joints; std::map<std::string, std::vector<std::string>> children;
  std::vector<Body> bodies; std::vector<JointRec> jrec; std::map<std::string, FrameRec> frames; std::string err;
  void addLinkInertia(int body, const XNode* link, const M3& R, const V3& p) {
    const XNode* in = link->child("inertial"); if (!in) return;
    M3 Ri; V3 c; origin(in, Ri, c);
    const XNode* mass = in->child("mass"); const XNode* ine = in->child("inertia"); if (!mass || !ine) return;
    const double m = atof(mass->attr.at("value").c_str());
    auto g = [&](const char* k) { auto it = ine->attr.find(k); return it == ine->attr.end() ? 0.0 : atof(it->second.c_str()); };
    M3 I{{g("ixx"), g("ixy"), g("ixz"), g("ixy"), g("iyy"), g("iyz"), g("ixz"), g("iyz"), g("izz")}};
    I = mul(mul(Ri, I), tr(Ri));                       // link axes
    const V3 cb = add(mulv(R, c), p); const M3 Ib = mul(mul(R, I), tr(R));
    Body& b = bodies[body]; b.m += m; for (int i = 0; i < 3; ++i) b.mc.x[i] += m * cb.x[i];
    for (int i = 0; i < 9; ++i) b.Io.m[i] += Ib.m[i];
    addPointMass(b.Io, m, cb);
  }
  bool visit(const std::string& link, int body, const M3& R, const V3& p) {
    frames[link] = {body, R, p};
    addLinkInertia(body, links.at(link), R, p);
    auto it = children.find(link); if (it == children.end()) return true;
    for (const std::string& jn : it->second) {
      const XNode* j = joints.at(jn); M3 Rj; V3 pj; origin(j, Rj, pj);
      const M3 Rw = mul(R, Rj); const V3 pw = add(mulv(R, pj), p);
      const std::string child = j->child("child")->attr.at("link"); const std::string type = j->attr.count("type") ? j->attr.at("type") : "";
      if (type == "fixed") { if (!visit(child, body, Rw, pw)) return false; }
      else if (type == "revolute" || type == "continuous") {
        const XNode* ax = j->child("axis"); const XNode* lim = j->child("limit");
        double a[3] = {1, 0, 0}; if (ax) parse3(ax->attr.at("xyz"), a);
        JointRec r; r.name = jn; r.parent = body; r.R = Rw; r.p = pw; r.axis = {{a[0], a[1], a[2]}};
        r.lo = lim && lim->attr.count("lower") ? atof(lim->attr.at("lower").c_str()) : -1e22; r.hi = lim && lim->attr.count("upper") ? atof(lim->attr.at("upper").c_str()) : 1e22;
        r.effort = lim && lim->attr.count("effort") ? atof(lim->attr.at("effort").c_str()) : 0.0;
        bodies.emplace_back(); jrec.push_back(r);
        if (!visit(child, (int)bodies.size() - 1, eye(), {{0, 0, 0}})) return false;
      } else { err = "unsupported joint type '" + type + "' at " + jn; return false; }
    }
    return true;
  }
};

static const char* kFootFrames[4] = {"LF_FOOT", "RF_FOOT", "LH_FOOT", "RH_FOOT"};   // contactNames3DoF, ModelSettings.h:38

bool fileExists(const std::string& p) { std::ifstream f(p); return (bool)f; }

static void fkBlob(const double* mb, const double* q /*24*/, std::vector<M3>& R, std::vector<V3>& p) {
  R.assign(QM_NB, eye()); p.assign(QM_NB, V3{{0, 0, 0}});
  R[0] = rpy(q[5], q[4], q[3]); p[0] = {{q[0], q[1], q[2]}};
  for (int j = 0; j < QM_NJ; ++j) {
    const int par = (int)mb[MB_PARENT + j]; M3 Rj; memcpy(Rj.m, mb + MB_JR + 9 * j, 72); V3 pj; memcpy(pj.x, mb + MB_JP + 3 * j, 24); V3 ax; memcpy(ax.x, mb + MB_AXIS + 3 * j, 24);
    R[j + 1] = mul(mul(R[par], Rj), axisAngle(ax, q[6 + j])); p[j + 1] = add(p[par], mulv(R[par], pj));
  }
}

bool buildModelBlob(const std::string& urdf, const std::string& referenceInfo, const std::string& eeFrame, double* mb, std::vector<std::string>* jointNames, std::string& err) {
  std::ifstream f(urdf); if (!f) { err = "URDF file not found: " + urdf; return false; }
  std::stringstream ss; ss << f.rdbuf(); const std::string src = ss.str();
  XParser xp(src); auto root = xp.element(); if (!root) { err = "URDF parse error: " + xp.err; return false; }
  if (root->name != "robot") { err = "URDF root element is not <robot>"; return false; }
  Builder b; std::map<std::string, bool> isChild;
  for (auto& k : root->kids) { if (k->name == "link") b.links[k->attr["name"]] = k.get(); else if (k->name == "joint" && k->child("parent") && k->child("child")) b.joints[k->attr["name"]] = k.get(); }
  for (auto& kv : b.joints) { b.children[kv.second->child("parent")->attr.at("link")].push_back(kv.first); isChild[kv.second->child("child")->attr.at("link")] = true; }   // std::map iteration = name order
  std::string rootLink; for (auto& kv : b.links) if (!isChild.count(kv.first)) { if (!rootLink.empty()) { err = "URDF has several root links"; return false; } rootLink = kv.first; }
  if (rootLink.empty()) { err = "URDF has no root link"; return false; }
  b.bodies.emplace_back();
  if (!b.visit(rootLink, 0, eye(), {{0, 0, 0}})) { err = b.err; return false; }
  if ((int)b.jrec.size() != QM_NJ) { err = "model must have 18 actuated joints, found " + std::to_string(b.jrec.size()); return false; }
  // topology the device kernels are specialised to: 4 chains of 3 + 1 chain of 6 hanging off the base
  for (int j = 0; j < QM_NJ; ++j) { const bool first = (j < 12) ? (j % 3 == 0) : (j == 12); const int want = first ? 0 : j; if (b.jrec[j].parent != want) { err = "unsupported kinematic topology at joint " + b.jrec[j].name; return false; } }
  std::fill(mb, mb + MB_SIZE, 0.0);
  for (int j = 0; j < QM_NJ; ++j) {
    const JointRec& r = b.jrec[j]; mb[MB_PARENT + j] = r.parent; memcpy(mb + MB_JR + 9 * j, r.R.m, 72); memcpy(mb + MB_JP + 3 * j, r.p.x, 24); memcpy(mb + MB_AXIS + 3 * j, r.axis.x, 24);
    mb[MB_QLO + j] = r.lo; mb[MB_QHI + j] = r.hi; mb[MB_TAUMAX + j] = r.effort; if (jointNames) jointNames->push_back(r.name);
  }
  for (int k = 0; k < QM_NB; ++k) {
    const Body& bd = b.bodies[k]; if (!(bd.m > 0)) { err = "body without mass"; return false; }
    V3 c{{bd.mc.x[0] / bd.m, bd.mc.x[1] / bd.m, bd.mc.x[2] / bd.m}}; M3 Ic = bd.Io; addPointMass(Ic, -bd.m, c);
    mb[MB_MASS + k] = bd.m; memcpy(mb + MB_COM + 3 * k, c.x, 24); memcpy(mb + MB_INERTIA + 9 * k, Ic.m, 72);
  }
  for (int fIdx = 0; fIdx < QM_NF; ++fIdx) {
    const std::string name = fIdx < 4 ? kFootFrames[fIdx] : eeFrame; auto it = b.frames.find(name);
    if (it == b.frames.end()) { err = "frame not found in URDF: " + name; return false; }
    mb[MB_FPARENT + fIdx] = it->second.body; memcpy(mb + MB_FR + 9 * fIdx, it->second.R.m, 72); memcpy(mb + MB_FP + 3 * fIdx, it->second.p.x, 24);
  }
  for (int c = 0; c < 4; ++c) { const int chain = (c == 1) ? 2 : (c == 2) ? 1 : c; if ((int)mb[MB_FPARENT + c] != 3 * chain + 3) { err = std::string("foot frame not at the tip of its leg chain: ") + kFootFrames[c]; return false; } }
  if ((int)mb[MB_FPARENT + 4] != 18) { err = "end-effector frame not at the tip of the arm chain"; return false; }
  // defaultJointState + centroidal info (ccrba at q_nom, v = 0)
  INode ref; if (!parseInfo(referenceInfo, ref, err)) return false;
  std::vector<double> qn; if (!infoMatrix(ref, "defaultJointState", QM_NJ, 1, qn, err)) return false;
  memcpy(mb + MB_QNOM, qn.data(), QM_NJ * 8);
  double q[QM_NQ] = {0}; memcpy(q + 6, qn.data(), QM_NJ * 8);
  std::vector<M3> R; std::vector<V3> p; fkBlob(mb, q, R, p);
  double mass = 0; V3 com{{0, 0, 0}}; std::vector<V3> cw(QM_NB);
  for (int k = 0; k < QM_NB; ++k) { V3 c; memcpy(c.x, mb + MB_COM + 3 * k, 24); cw[k] = add(p[k], mulv(R[k], c)); mass += mb[MB_MASS + k]; for (int i = 0; i < 3; ++i) com.x[i] += mb[MB_MASS + k] * cw[k].x[i]; }
  for (int i = 0; i < 3; ++i) com.x[i] /= mass;
  M3 Ig{{0, 0, 0, 0, 0, 0, 0, 0, 0}};
  for (int k = 0; k < QM_NB; ++k) { M3 I; memcpy(I.m, mb + MB_INERTIA + 9 * k, 72); M3 Iw = mul(mul(R[k], I), tr(R[k])); for (int i = 0; i < 9; ++i) Ig.m[i] += Iw.m[i]; addPointMass(Ig, mb[MB_MASS + k], sub(cw[k], com)); }
  mb[MB_ROBOTMASS] = mass; memcpy(mb + MB_INOM, Ig.m, 72); for (int i = 0; i < 3; ++i) mb[MB_RNOM + i] = q[i] - com.x[i];
  return true;
}

bool buildSettingsBlob(const std::string& taskInfo, const double* mb, double* st, std::string& err) {
  INode t; if (!parseInfo(taskInfo, t, err)) return false;
  std::fill(st, st + ST_SIZE, 0.0);
  std::vector<double> Q, Rt, xi, lo, hi;
  if (!infoMatrix(t, "Q", QM_NX, QM_NX, Q, err) || !infoMatrix(t, "R", QM_NU, QM_NU, Rt, err) || !infoMatrix(t, "initialState", QM_NX, 1, xi, err)) return false;
  for (int i = 0; i < QM_NX; ++i) { st[ST_Q + i] = Q[(size_t)i * QM_NX + i]; st[ST_XINIT + i] = xi[i]; }
  // initializeInputCostWeight: leg block <- Jᵀ R J with J = feet-position Jacobian wrt the 12 leg joints at initialState
  std::vector<M3> R; std::vector<V3> p; fkBlob(mb, xi.data() + 6, R, p);
  double J[12][12] = {{0}};
  for (int c = 0; c < 4; ++c) {
    int b = (int)mb[MB_FPARENT + c]; V3 fp; memcpy(fp.x, mb + MB_FP + 3 * c, 24); const V3 pf = add(p[b], mulv(R[b], fp));
    while (b > 0) { const int j = b - 1; V3 ax; memcpy(ax.x, mb + MB_AXIS + 3 * j, 24); const V3 aw = mulv(R[b], ax); const V3 col = cross(aw, sub(pf, p[b])); if (j < 12) for (int r = 0; r < 3; ++r) J[3 * c + r][j] = col.x[r]; b = (int)mb[MB_PARENT + j]; }
  }
  for (int i = 0; i < QM_NU * QM_NU; ++i) st[ST_R + i] = Rt[i];
  for (int a = 0; a < 12; ++a) for (int b2 = 0; b2 < 12; ++b2) {
    double s = 0; for (int i = 0; i < 12; ++i) { double ri = 0; for (int k = 0; k < 12; ++k) ri += Rt[(size_t)(12 + i) * QM_NU + 12 + k] * J[k][b2]; s += J[i][a] * ri; }
    st[ST_R + (12 + a) * QM_NU + 12 + b2] = s;
  }
  struct KV { const char* key; int idx; };
  const KV kv[] = {{"endEffector.muPosition", ST_MU_EE_POS}, {"endEffector.muOrientation", ST_MU_EE_ORI}, {"finalEndEffector.muPosition", ST_MU_EEF_POS}, {"finalEndEffector.muOrientation", ST_MU_EEF_ORI},
                   {"frictionConeSoftConstraint.frictionCoefficient", ST_FRIC_COEF}, {"frictionConeSoftConstraint.mu", ST_FRIC_MU}, {"frictionConeSoftConstraint.delta", ST_FRIC_DELTA},
                   {"jointPositionLimits.mu", ST_JPOS_MU}, {"jointPositionLimits.delta", ST_JPOS_DELTA}, {"jointVelocityLimits.mu", ST_JVEL_MU}, {"jointVelocityLimits.delta", ST_JVEL_DELTA},
                   {"model_settings.positionErrorGain", ST_POS_ERR_GAIN}, {"model_settings.phaseTransitionStanceTime", ST_PHASE_TRANS_STANCE},
                   {"swing_trajectory_config.liftOffVelocity", ST_LIFTOFF_VEL}, {"swing_trajectory_config.touchDownVelocity", ST_TOUCHDOWN_VEL}, {"swing_trajectory_config.swingHeight", ST_SWING_HEIGHT},
                   {"swing_trajectory_config.swingTimeScale", ST_SWING_TIME_SCALE}, {"sqp.dt", ST_SQP_DT}, {"sqp.sqpIteration", ST_SQP_ITER}, {"sqp.deltaTol", ST_DELTA_TOL}, {"sqp.g_max", ST_G_MAX}, {"sqp.g_min", ST_G_MIN},
                   {"mpc.timeHorizon", ST_TIME_HORIZON}, {"frictionConeTask.frictionCoefficient", ST_WBC_FRIC}};
  for (const KV& e : kv) if (!infoScalar(t, e.key, st[e.idx], err)) return false;
  // The `ddp` and `ipm` blocks are read by solver slots 1 / 2 only, which the reference never instantiates (QMController.cpp:287-288): a task.info without them is
  // accepted and gets the solver structs' defaults ([upstream, recalled] ddp::Settings / ipm::Settings: loadSettings leaves the default where a key is missing).
  struct KVD { const char* key; int idx; double dflt; };
  const KVD opt[] = {{"ddp.lineSearch.minStepLength", ST_DDP_MIN_STEP, 0.05}, {"ddp.lineSearch.maxStepLength", ST_DDP_MAX_STEP, 1.0}, {"ddp.constraintPenaltyInitialValue", ST_DDP_PENALTY, 2.0},
                     {"ipm.dt", ST_IPM_DT, 0.01}, {"ipm.ipmIteration", ST_IPM_ITER, 10.0}, {"ipm.deltaTol", ST_IPM_DELTA_TOL, 1e-6}, {"ipm.g_max", ST_IPM_G_MAX, 1e6}, {"ipm.g_min", ST_IPM_G_MIN, 1e-6},
                     {"ipm.initialBarrierParameter", ST_IPM_MU, 1e-2}, {"ipm.targetBarrierParameter", ST_IPM_MU_TARGET, 1e-4}, {"ipm.barrierLinearDecreaseFactor", ST_IPM_MU_LINEAR, 0.2},
                     {"ipm.barrierSuperlinearDecreasePower", ST_IPM_MU_POWER, 1.5}, {"ipm.barrierReductionCostTol", ST_IPM_RED_COST_TOL, 1e-3}, {"ipm.barrierReductionConstraintTol", ST_IPM_RED_CON_TOL, 1e-3},
                     {"ipm.fractionToBoundaryMargin", ST_IPM_FTB_MARGIN, 0.995}, {"ipm.initialSlackLowerBound", ST_IPM_SLACK_LB, 1e-4}, {"ipm.initialDualLowerBound", ST_IPM_DUAL_LB, 1e-4},
                     {"ipm.initialSlackMarginRate", ST_IPM_SLACK_MARGIN, 1e-2}, {"ipm.initialDualMarginRate", ST_IPM_DUAL_MARGIN, 1e-2}};
  for (const KVD& e : opt) { std::string ignored; if (!infoScalar(t, e.key, st[e.idx], ignored)) st[e.idx] = e.dflt; }
  { st[ST_IPM_PRIMAL_FOR_DUAL] = 1.0; const INode* n = t.get("ipm.usePrimalStepSizeForDual"); if (n && !n->value.empty()) st[ST_IPM_PRIMAL_FOR_DUAL] = (n->value == "true" || n->value == "1") ? 1.0 : 0.0; }      // a boolean key
  if (!(st[ST_SQP_DT] > 0.0) || !(st[ST_SQP_DT] < 1.0e300)) { err = "INFO: sqp.dt must be a positive finite number"; return false; }   // K0 walks t0 + k dt up to the horizon
  // ipm.dt is validated where it is used: qmhip_set_setting(ST_SOLVER, 2) / (ST_IPM_DT, .) and K0's `sane` guard
  st[ST_GRID_DT_MIN] = QM_GRID_DT_MIN_UPSTREAM;         // [upstream] timeDiscretizationWithEvents' default dt_min = 10 * limitEpsilon
  st[ST_FRIC_REG] = 25.0; st[ST_FRIC_SHIFT] = 1e-6;     // [upstream] FrictionConeConstraint::Config defaults
  st[ST_SOLVER] = 0.0;                                  // the controller instantiates SqpMpc whatever `ddp.algorithm` says (QMController.cpp:287-288)
  if (!infoMatrix(t, "jointVelocityLimits.lowerBound.arm", 6, 1, lo, err) || !infoMatrix(t, "jointVelocityLimits.upperBound.arm", 6, 1, hi, err)) return false;
  for (int i = 0; i < 6; ++i) { st[ST_JVEL_LO + i] = lo[i]; st[ST_JVEL_HI + i] = hi[i]; }
  // qm_wbc/cfg/wbcWigeht.cfg defaults
  st[ST_KP_SWING] = 350; st[ST_KD_SWING] = 37; st[ST_KP_BASE_H] = 400; st[ST_KD_BASE_H] = 140; st[ST_KP_BASE_LIN] = 400; st[ST_KD_BASE_LIN] = 100; st[ST_KP_BASE_ANG] = 400; st[ST_KD_BASE_ANG] = 140;
  const double kpArm[6] = {4000, 4200, 4000, 4000, 4200, 6000};
  for (int i = 0; i < 6; ++i) { st[ST_KP_ARM_J + i] = kpArm[i]; st[ST_KD_ARM_J + i] = 75; }
  for (int i = 0; i < 3; ++i) { st[ST_KP_EE_LIN + i] = 3000; st[ST_KD_EE_LIN + i] = 75; st[ST_KP_EE_ANG + i] = 2000; st[ST_KD_EE_ANG + i] = 75; }
  return true;
}

bool loadEeFrameName(const std::string& taskInfo, std::string& name, std::string& err) {
  INode t; if (!parseInfo(taskInfo, t, err)) return false;
  const INode* n = t.get("model_settings.eeFrame"); if (!n || n->value.empty()) { err = "INFO: missing model_settings.eeFrame"; return false; }
  name = n->value; return true;
}

// topology / sanity validation of a MODEL blob given directly
bool validateModelBlob(const double* mb, std::string& err) {
  for (int j = 0; j < QM_NJ; ++j) { const bool first = (j < 12) ? (j % 3 == 0) : (j == 12); const int want = first ? 0 : j; if ((int)mb[MB_PARENT + j] != want) { err = "model blob: unsupported kinematic topology"; return false; } }
  for (int c = 0; c < 4; ++c) { const int chain = (c == 1) ? 2 : (c == 2) ? 1 : c; if ((int)mb[MB_FPARENT + c] != 3 * chain + 3) { err = "model blob: foot frame parent mismatch"; return false; } }
  if ((int)mb[MB_FPARENT + 4] != 18) { err = "model blob: end-effector frame parent mismatch"; return false; }
  if (!(mb[MB_ROBOTMASS] > 0)) { err = "model blob: non-positive mass"; return false; }
  return true;
}

}  // namespace qmio
