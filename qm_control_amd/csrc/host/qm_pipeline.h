// qm_pipeline.h — launch sequence of the MPC+WBC step, independent of how kernels get launched.
//
// `BK` (backend) provides:  template<class K, class A> void launch(K kernel, int grid, int block, size_t lds_bytes, const A& args);
//                           void* alloc(size_t bytes);  void free(void*);  void zero(void* p, size_t bytes);
//                           void to_device(void* dst, const void* src, size_t);  void to_host(void* dst, const void* src, size_t);  void sync();
// The product instantiates it with the HIP backend (qm_control_amd/csrc/host/qmhip.hip); the -m "not gpu" tests
// instantiate it with the host emulator so the very same sequence is exercised without a GPU.
//
// Sequence of one control step (QMController's mpcThread_ + update(), qm_controllers/src/QMController.cpp:128-175,315-332):
//   K0 grid  ->  K1 LQ+projection  ->  baseline performance  ->  K3 Riccati  ->  K4 line-search loop  ->  primal solution
//   ->  policy evaluation at t0 + K5..K7 whole-body controller
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <vector>
#include "../kernels/k_grid.h"
#include "../kernels/k_lq.h"
#include "../kernels/k_riccati.h"
#include "../kernels/k_ls.h"
#include "../kernels/k_ilqr.h"
#include "../kernels/k_ipm.h"

struct QmMpcBuffers {
  int Bmax = 0, nmax = 0, nref = 0, nev = 0;
  // model
  double* mb = nullptr; double* st = nullptr;
  // inputs (device copies)
  double* t0 = nullptr; double* x0 = nullptr; double* ref_t = nullptr; double* ref_x = nullptr; double* ev = nullptr; int* modes = nullptr;
  // grid
  int* n_nodes = nullptr; double* node_t = nullptr; double* node_ts = nullptr; double* node_dt = nullptr; int* node_ev = nullptr; int* node_mode = nullptr;
  double* zvel = nullptr; double* zpos = nullptr; double* xref = nullptr; double* eeref = nullptr; int* status = nullptr;
  // iterate, step, stage data
  double* x = nullptr; double* u = nullptr; double* dx = nullptr; double* du = nullptr; double* stage = nullptr; double* lqdbg = nullptr; double* kin = nullptr;
  double* perf = nullptr; double* base_sum = nullptr; double* perf_sum = nullptr; double* step_info = nullptr;
  double* alpha = nullptr; int* done = nullptr; double* xs = nullptr; double* us = nullptr; double* out_perf = nullptr;
  double* xt = nullptr; double* ut = nullptr;   // iLQR trial rollouts [nmax][B][30] (allocated on first use)
  // interior-point solver (slot 3, k_ipm.h; allocated on first use): slack / dual and their directions [nmax][B][QM_NH], step-limit ratios [nmax][B][2], per-instance info [B][IPM_INFO]
  double* ipm_s = nullptr; double* ipm_l = nullptr; double* ipm_ds = nullptr; double* ipm_dl = nullptr; double* ipm_ratio = nullptr; double* ipm_info = nullptr;
  // grid of the solve that produced (xs, us): the warm start of the next solve interpolates on it
  int* prev_n = nullptr; double* prev_t = nullptr; int* prev_ev = nullptr;
  // line search: instances still searching after trial t (device counters + their host-visible copy)
  int* open_cnt = nullptr; int* tickets = nullptr; int* host_open_dev = nullptr; volatile int* host_open = nullptr;
  // largest node count of the batch, published by K0 (the per-node launches cover only that many nodes per instance)
  int* ncap_dev = nullptr; int* host_ncap_dev = nullptr; volatile int* host_ncap = nullptr;
};

// Status of one instance's solve as the C ABI reports it (qmhip_mpc_download), from K0's status word and K3's step_info[4] = {Armijo metric, |dx|², |du|², pivot flags}.
// Pivot flags (k_riccati.h): bit 0 = non-positive pivots of Huu on a stage of NON-POSITIVE duration (the interval in front of a gait event: zeroed, the solve is valid ->
// the warning QM_MPC_WARN_PIVOT, or -4 with ST_RICCATI_STRICT); bit 1 = a non-positive pivot on a stage of positive duration (Huu genuinely indefinite) or a pivot that is
// not a number.  Bit 1, or a step that is not finite (a NaN in the observation reaches all three sums), is the hard failure -4: [upstream] SqpSolver throws on HPIPM's NaN
// status, the controller stops (QMController.cpp:315-333) — never a policy made of NaNs behind a "valid solution" status.
inline int qm_mpc_status(int k0_status, const double* step_info4, bool strict) {
  if (k0_status != 0) return k0_status;
  const double* s4 = step_info4; const int pv = (s4[3] == s4[3]) ? (int)s4[3] : 2;
  const bool finite = (s4[0] - s4[0] == 0.0) && (s4[1] - s4[1] == 0.0) && (s4[2] - s4[2] == 0.0);
  if (!finite || (pv & 2)) return -4;
  if (pv & 1) return strict ? -4 : QM_MPC_WARN_PIVOT;
  return 0;
}

template <class BK>
struct QmMpcPipeline {
  BK& bk; QmMpcBuffers d;
  int ls_trials_run = 0;          // trials of the last line search (the longest of the batch); with the device-side tail it is known only on the device: read through ls_trials()
  bool ls_trials_pending = false; int ls_trials_cap = 0;
  // policy at t0 from the deciding kernels (k_ls.h): where the step's whole-body controller reads its inputs; enabled per solve by the control-step entry points
  double* p0_x = nullptr; double* p0_u = nullptr; int* p0_mode = nullptr; bool p0_enable = false; bool p0_done = false;      // p0_done: the last sqp_iteration wrote them
  bool defer_apply = false; int pending_apply_B = 0; int pending_apply_threads = 0;      // the caller launches the batch's apply itself (apply_pending) — behind the WBC launch of a control step
  bool device_tail = true;        // SQP: the trials after the first run in ONE launch without the host (qm_ls_tail_kernel, k_ls.h); false = the host-driven trial loop of rounds 1-5 (tests, A/B)
  int riccati_skip = 0;   // profiling only
  int lq_prof = 0;        // profiling only
  int solver = 0;         // 0: multiple-shooting SQP (the reference's SqpMpc), 1: discrete iLQR, 2: the SQP path run on the `ipm` block's parameters (not an interior-point method), 3: interior-point method with hard cones / boxes (k_ipm.h)
                          // (no hard inequality rows in this OCP: include/qmhip_layout.h, ST_IPM_*); settings slot ST_SOLVER
  bool r_blocks = false;           // the input weight R of the settings blob is block diagonal (k_ls.h): the structured instance of the trial-evaluation kernel and the structured R0 (u − u_nom) of K1b run; kept current by note_settings()
  bool r_force_dense = false;      // tests / A-B only (qmhip_debug_set "r_dense"): the dense instances run although R is block diagonal — same bits, by construction
  bool rblk() const { return r_blocks && !r_force_dense; }
  bool speculative_apply = true;   // tests only: false = the first trial's apply waits for the host's decision like every later one (A/B of the invariant below)
  int lq_slices = 1;      // K1a / K1b run the horizon in this many node slices (1: one launch each)
  bool ipm_fresh = true;  // interior-point solver: the next iteration is the first of its solve (K0 ran): slack / dual / barrier parameter are initialised at the initial iterate
  int solved_B = 0;       // batch size of the last completed solve (0: none yet -> a warm start falls back to the cold start)
  int ncap = 0;           // nodes per instance the per-node launches of the current grid cover (0: not read back yet)
  bool ncap_pending = false;   // K0 has been launched and its count not been read yet
  bool has_m18 = true;         // some horizon of the current grid holds an all-stance phase (K0's conservative flag): K1b's second instance is launched
  std::function<void()> before_lq;   // profiling only (co-residency probe): called right before the LQ kernel is launched
  const int* front_status = nullptr; int front_B = 0;   // sticky status of the device-resident GaitSchedule driving batches of front_B instances (null: schedules come from the host)
  explicit QmMpcPipeline(BK& b) : bk(b) {}

  template <class T> T* A(size_t n) { T* p = (T*)bk.alloc(n * sizeof(T)); bk.zero(p, n * sizeof(T)); return p; }

  void allocate(const double* mb_host, const double* st_host, int Bmax, int nmax, int nref, int nev, bool debug_lq) {
    d.Bmax = Bmax; d.nmax = nmax; d.nref = nref; d.nev = nev;
    const size_t NB = (size_t)nmax * Bmax;
    d.mb = A<double>(MB_SIZE); d.st = A<double>(ST_SIZE);
    bk.to_device(d.mb, mb_host, MB_SIZE * 8); bk.to_device(d.st, st_host, ST_SIZE * 8); note_settings(st_host);
    d.t0 = A<double>(Bmax); d.x0 = A<double>((size_t)Bmax * 30); d.ref_t = A<double>((size_t)Bmax * nref); d.ref_x = A<double>((size_t)Bmax * nref * QM_NREF);
    d.ev = A<double>((size_t)Bmax * nev); d.modes = A<int>((size_t)Bmax * (nev + 1));
    d.n_nodes = A<int>(Bmax); d.node_t = A<double>(NB); d.node_ts = A<double>(NB); d.node_dt = A<double>(NB); d.node_ev = A<int>(NB); d.node_mode = A<int>(NB);
    d.zvel = A<double>(NB * 4); d.zpos = A<double>(NB * 4); d.xref = A<double>(NB * 30); d.eeref = A<double>(NB * 7); d.status = A<int>(Bmax);
    d.x = A<double>(NB * 30); d.u = A<double>(NB * 30); d.dx = A<double>(NB * 30); d.du = A<double>(NB * 30);
    d.stage = A<double>(NB * SR_SIZE); d.lqdbg = debug_lq ? A<double>(NB * LQ_DBG_SIZE) : nullptr; d.kin = A<double>((NB + 64) * KR_SIZE);      // (+ 64 records: K1a's waves store whole 64-record blocks, k_lq.h)
    d.perf = A<double>(NB * PF_SIZE); d.base_sum = A<double>((size_t)Bmax * 4); d.perf_sum = A<double>((size_t)Bmax * 4); d.step_info = A<double>((size_t)Bmax * 4);
    d.alpha = A<double>(Bmax); d.done = A<int>(Bmax); d.xs = A<double>(NB * 30); d.us = A<double>(NB * 30); d.out_perf = A<double>((size_t)Bmax * 10);
    d.prev_n = A<int>(Bmax); d.prev_t = A<double>(NB); d.prev_ev = A<int>(NB);
    d.open_cnt = A<int>(QM_LS_MAX_TRIALS); d.tickets = A<int>(QM_LS_MAX_TRIALS);
    d.ncap_dev = A<int>(3); { void* hv = nullptr; d.host_ncap_dev = (int*)bk.alloc_mapped(sizeof(int), &hv); d.host_ncap = (volatile int*)hv; d.host_ncap[0] = 0; }
    { void* hv = nullptr; d.host_open_dev = (int*)bk.alloc_mapped(QM_LS_MAX_TRIALS * sizeof(int), &hv); d.host_open = (volatile int*)hv; for (int i = 0; i < QM_LS_MAX_TRIALS; ++i) d.host_open[i] = 0; }
  }
  void note_settings(const double* st_host) { r_blocks = qm_r_is_block_diagonal(st_host); }      // after every change of the settings blob (create, qmhip_set_setting)
  void release() {
    void* ps[] = {d.mb, d.st, d.t0, d.x0, d.ref_t, d.ref_x, d.ev, d.modes, d.n_nodes, d.node_t, d.node_ts, d.node_dt, d.node_ev, d.node_mode, d.zvel, d.zpos, d.xref, d.eeref, d.status,
                  d.x, d.u, d.dx, d.du, d.xt, d.ut, d.ipm_s, d.ipm_l, d.ipm_ds, d.ipm_dl, d.ipm_ratio, d.ipm_info, d.stage, d.lqdbg, d.kin, d.perf, d.base_sum, d.perf_sum, d.step_info, d.alpha, d.done, d.xs, d.us, d.out_perf, d.prev_n, d.prev_t, d.prev_ev, d.open_cnt, d.tickets, d.ncap_dev};
    for (void* p : ps) if (p) bk.free(p);
    if (d.host_open) bk.free_mapped((void*)d.host_open);
    if (d.host_ncap) bk.free_mapped((void*)d.host_ncap);
    d = QmMpcBuffers();
  }

  // number of line-search trials the last sqp_iteration needed (max over the batch).  After a device-side tail: 1 + the number of leading trials t that left an instance
  // searching (open_cnt[t] > 0), read back on demand — a synchronising copy, not on the hot path
  int ls_trials() {
    if (ls_trials_pending) { int oc[QM_LS_MAX_TRIALS]; bk.to_host(oc, d.open_cnt, sizeof(oc)); int t = 0; while (t < QM_LS_MAX_TRIALS && oc[t] > 0) ++t; ls_trials_run = (1 + t > ls_trials_cap) ? ls_trials_cap : 1 + t; ls_trials_pending = false; }
    return ls_trials_run;
  }

  // the deferred apply of the last sqp_iteration (defer_apply): x + alpha dx on every node -> the primal solution (xs, us)
  void apply_pending() { if (!pending_apply_B) return; QmLsArgs l = ls_args(pending_apply_B); bk.launch(qm_ls_apply_kernel, (pending_apply_threads * 30 + 255) / 256, 256, 0, l); pending_apply_B = 0; }

  // inputs are HOST pointers (instance-major, as the C ABI receives them)
  void upload_inputs(int B, const double* t0, const double* x0, const double* ref_t, const double* ref_x, const double* ev, const int* modes) {
    bk.to_device(d.t0, t0, (size_t)B * 8); bk.to_device(d.x0, x0, (size_t)B * 30 * 8);
    bk.to_device(d.ref_t, ref_t, (size_t)B * d.nref * 8); bk.to_device(d.ref_x, ref_x, (size_t)B * d.nref * QM_NREF * 8);
    bk.to_device(d.ev, ev, (size_t)B * d.nev * 8); bk.to_device(d.modes, modes, (size_t)B * (d.nev + 1) * 4);
  }

  QmLsArgs ls_args(int B) {
    QmLsArgs a; a.mb = d.mb; a.st = d.st; a.B = B; a.nmax = d.nmax; a.n_nodes = d.n_nodes; a.node_ts = d.node_ts; a.node_dt = d.node_dt; a.node_ev = d.node_ev; a.node_mode = d.node_mode;
    a.zvel = d.zvel; a.zpos = d.zpos; a.xref = d.xref; a.eeref = d.eeref; a.x0 = d.x0; a.x = d.x; a.u = d.u; a.dx = d.dx; a.du = d.du; a.alpha = d.alpha; a.done = d.done;
    a.perf = d.perf; a.perf_sum = d.perf_sum; a.base_sum = d.base_sum; a.step_info = d.step_info; a.xs = d.xs; a.us = d.us; a.out_perf = d.out_perf; a.trial = 0; a.max_trials = 0; a.node_t = d.node_t; a.p0_t = d.t0; a.p0_ev = d.ev; a.p0_modes = d.modes; a.p0_nev = d.nev; a.p0_x = nullptr; a.p0_u = nullptr; a.p0_mode = nullptr; a.with_alpha = 0; a.open_cnt = d.open_cnt; a.tickets = d.tickets; a.host_open = (volatile int*)d.host_open_dev; a.xt = nullptr; a.ut = nullptr; a.ilqr = 0; a.ipm_s = nullptr; a.ipm_ds = nullptr; a.ipm_info = nullptr;
    return a;
  }

  // K0: grid + references + initial guess (inputs already resident on the device).  warm: interpolate the previous primal solution
  // ([upstream] SqpSolver keeps primalSolution_ between MPC calls, mpc.coldStart false, task.info:142) — falls back to cold without one.
  void grid(int B, double horizon, bool warm = false) {
    warm = warm && solved_B == B;
    if (warm) { QmSaveGridArgs sg; sg.B = B; sg.nmax = d.nmax; sg.n_nodes = d.n_nodes; sg.node_t = d.node_t; sg.node_ev = d.node_ev; sg.prev_n = d.prev_n; sg.prev_t = d.prev_t; sg.prev_ev = d.prev_ev;
                bk.launch(qm_save_grid_kernel, (d.nmax * B + 63) / 64, 64, 0, sg); }
    QmGridArgs g; g.mb = d.mb; g.st = d.st; g.B = B; g.nmax = d.nmax; g.nref = d.nref; g.nev = d.nev; g.t0 = d.t0; g.x0 = d.x0; g.ref_t = d.ref_t; g.ref_x = d.ref_x; g.ev = d.ev; g.modes = d.modes;
    g.horizon = horizon; g.n_nodes = d.n_nodes; g.node_t = d.node_t; g.node_ts = d.node_ts; g.node_dt = d.node_dt; g.node_ev = d.node_ev; g.node_mode = d.node_mode;
    g.zvel = d.zvel; g.zpos = d.zpos; g.xref = d.xref; g.eeref = d.eeref; g.x = d.x; g.u = d.u; g.status = d.status; g.front_status = (front_status && front_B == B) ? front_status : nullptr;
    if (ncap_pending) bk.wait_flag(d.host_ncap, -1);          // a grid whose count was never read: let it publish before the word is re-armed
    g.ncap_dev = d.ncap_dev; g.host_ncap = (volatile int*)d.host_ncap_dev; d.host_ncap[0] = -1; ncap = 0; ncap_pending = true;
    g.warm = warm ? 1 : 0; g.prev_n = d.prev_n; g.prev_t = d.prev_t; g.prev_ev = d.prev_ev; g.prev_xs = d.xs; g.prev_us = d.us;
    bk.launch(qm_grid_kernel, (B + 63) / 64, 64, 0, g);
    bk.launch(qm_grid_nodes_kernel, (d.nmax * B + 63) / 64, 64, 0, g);
    ipm_fresh = true;
  }
  // closed loop with a perfect-tracking plant: t0 += dt, x0 <- policy state at the new t0 (uses the grid / primal solution of the last solve)
  void advance(int B, double dt) {
    QmAdvanceArgs v; v.B = B; v.nmax = d.nmax; v.n_nodes = d.n_nodes; v.node_t = d.node_t; v.node_ev = d.node_ev; v.xs = d.xs; v.dt = dt; v.t0 = d.t0; v.x0 = d.x0;
    bk.launch(qm_advance_kernel, (B + 63) / 64, 64, 0, v);
  }
  // one SQP iteration on the current iterate (x,u); max_trials bounds the line search (14 reaches alpha_min).  `last`: no further iteration of this solve
  // follows, so the accepted step only has to reach the primal solution (xs, us), not the iterate (x, u) — the next solve starts from xs / us or cold
  void sqp_iteration(int B, int max_trials = 14, bool last = false) {
    const bool ilqr = solver == 1, ipm = solver == 3;
    QmIpmArgs ia;
    if (ipm) {
      const size_t NBm = (size_t)d.nmax * d.Bmax;
      if (!d.ipm_s) { d.ipm_s = A<double>(NBm * QM_NH); d.ipm_l = A<double>(NBm * QM_NH); d.ipm_ds = A<double>(NBm * QM_NH); d.ipm_dl = A<double>(NBm * QM_NH); d.ipm_ratio = A<double>(NBm * 2); d.ipm_info = A<double>((size_t)d.Bmax * IPM_INFO); }
      ia.mb = d.mb; ia.st = d.st; ia.B = B; ia.nmax = d.nmax; ia.n_nodes = d.n_nodes; ia.node_ev = d.node_ev; ia.node_mode = d.node_mode; ia.node_dt = d.node_dt; ia.x = d.x; ia.u = d.u; ia.dx = d.dx; ia.du = d.du;
      ia.s = d.ipm_s; ia.lam = d.ipm_l; ia.ds = d.ipm_ds; ia.dlam = d.ipm_dl; ia.ratio = d.ipm_ratio; ia.info = d.ipm_info; ia.alpha = d.alpha; ia.done = d.done; ia.out_perf = d.out_perf;
      if (ipm_fresh) bk.launch(qm_ipm_init_kernel, (d.nmax * B + 63) / 64, 64, 0, ia);      // slack / dual at the initial iterate, mu = ipm.initialBarrierParameter (every solve starts over: k_ipm.h)
      ipm_fresh = false;
    }
    if (ilqr && !d.xt) { d.xt = A<double>((size_t)d.nmax * d.Bmax * 30); d.ut = A<double>((size_t)d.nmax * d.Bmax * 30); }
    QmRolloutArgs ro; ro.mb = d.mb; ro.st = d.st; ro.B = B; ro.nmax = d.nmax; ro.mode = 0; ro.trial = 0; ro.n_nodes = d.n_nodes; ro.node_dt = d.node_dt; ro.node_ev = d.node_ev; ro.x0 = d.x0;
    ro.x = d.x; ro.u = d.u; ro.stage = d.stage; ro.alpha = d.alpha; ro.done = d.done; ro.xt = d.xt; ro.ut = d.ut;
    if (ilqr) bk.launch(qm_ilqr_rollout_kernel, B, 64, 0, ro);      // single shooting: the nominal states are the rollout of the initial inputs
    if (ncap == 0) { if (ncap_pending) bk.wait_flag(d.host_ncap, -1); const int word = ncap_pending ? d.host_ncap[0] : (d.nmax | (1 << 16)); ncap_pending = false;
                     ncap = word & 0xFFFF; has_m18 = (word >> 16) != 0; if (ncap < 1 || ncap > d.nmax) { ncap = d.nmax; has_m18 = true; } }   // K0 ran first in the stream: published long before K1a is done
    const int nodes_threads = ncap * B;
    if (max_trials > QM_LS_MAX_TRIALS) max_trials = QM_LS_MAX_TRIALS;
    QmLqArgs q; q.mb = d.mb; q.st = d.st; q.B = B; q.nmax = d.nmax; q.n_nodes = d.n_nodes; q.node_ts = d.node_ts; q.node_dt = d.node_dt; q.node_ev = d.node_ev; q.node_mode = d.node_mode;
    q.zvel = d.zvel; q.zpos = d.zpos; q.xref = d.xref; q.eeref = d.eeref; q.x = d.x; q.u = d.u; q.stage = d.stage; q.perf = d.perf; q.dbg = d.lqdbg; q.kin = d.kin; q.prof = lq_prof; q.ncap = ncap;
    q.ipm_s = d.ipm_s; q.ipm_l = d.ipm_l; q.ipm_info = d.ipm_info; q.rb = rblk() ? 1 : 0; q.single_mt = has_m18 ? 0 : 1;
    q.i0 = 0;
    const int nsl = (ipm || d.lqdbg || lq_prof || lq_slices < 1) ? 1 : (lq_slices > ncap ? ncap : lq_slices);
    if (nsl > 1) {
      // node slices: kin records of a slice are written by K1a and read back by K1b before the next slice's records push them out of the memory-side cache
      if (before_lq) before_lq();
      for (int sidx = 0; sidx < nsl; ++sidx) {
        const int a0 = (int)((long long)ncap * sidx / nsl), a1 = (int)((long long)ncap * (sidx + 1) / nsl); if (a1 <= a0) continue;
        q.i0 = a0; q.ncap = a1 - a0;
        bk.launch(qm_lq_kin_kernel, ((a1 - a0) * B + 63) / 64, 64, LQ_KIN_LDS_BYTES, q);
        bk.launch(qm_lq_kernel, B * (a1 - a0), LW_BLOCK, LQ_LDS_BYTES, q);
        if (has_m18) bk.launch(qm_lq_m18_kernel, B * (a1 - a0), LW_BLOCK, LQ_LDS_BYTES, q);
      }
      q.i0 = 0; q.ncap = ncap;
    } else {
    bk.launch(qm_lq_kin_kernel, (nodes_threads + 63) / 64, 64, LQ_KIN_LDS_BYTES, q);
    if (before_lq) before_lq();
    if (ipm) bk.launch(qm_lq_ipm_kernel, B * ncap, LW_BLOCK, LQ_LDS_BYTES, q);      // the interior-point instance: condensed inequality rows instead of the soft barrier costs
    else if (d.lqdbg || lq_prof) bk.launch(qm_lq_dbg_kernel, B * ncap, LW_BLOCK, LQ_LDS_BYTES, q);   // the instance with debug records / phase cycle stamps (parity tests, profiling)
    else { bk.launch(qm_lq_kernel, B * ncap, LW_BLOCK, LQ_LDS_BYTES, q);      // one wavefront per node: the nodes with m <= 16 reduced inputs (any gait phase with a swing leg) ...
           if (has_m18) bk.launch(qm_lq_m18_kernel, B * ncap, LW_BLOCK, LQ_LDS_BYTES, q); }  // ... and the stance nodes (m = 18): two instances of one body, three waves per SIMD each (k_lq.h)
    }
    QmLsArgs l = ls_args(B); if (ilqr) { l.xt = d.xt; l.ut = d.ut; l.ilqr = 1; }
    if (ipm) { l.ipm_s = d.ipm_s; l.ipm_ds = d.ipm_ds; l.ipm_info = d.ipm_info; }
    QmRiccatiArgs r; r.B = B; r.nmax = d.nmax; r.n_nodes = d.n_nodes; r.node_ev = d.node_ev; r.x0 = d.x0; r.x = d.x; r.stage = d.stage; r.dx = d.dx; r.du = d.du; r.step_info = d.step_info; r.skip = riccati_skip;
    r.perf = d.perf; r.base_sum = d.base_sum; r.alpha = d.alpha; r.done = d.done; r.out_perf = d.out_perf; r.open_cnt = d.open_cnt; r.tickets = d.tickets;   // baseline merit + arming of the line search
    if (riccati_skip) bk.launch(qm_riccati_prof_kernel, B, RW_BLOCK, RW_LDS_BYTES, r);   // instrumented instance: phase skip bits, in-kernel cycle counters (profiling / parity tests only)
    else bk.launch(qm_riccati_kernel, B, RW_BLOCK, RW_LDS_BYTES, r);   // one wavefront per instance
    if (ipm) { bk.launch(qm_ipm_dir_kernel, (d.nmax * B + 63) / 64, 64, 0, ia); bk.launch(qm_ipm_alpha_kernel, B, 64, 0, ia); }      // slack / dual directions, fraction to the boundary: the line search starts at alphaP
    ls_trials_run = 0; ls_trials_pending = false; p0_done = false;
    if (device_tail && !ilqr && !ipm && max_trials >= 1) {
      // SQP line search without the host: trial 0 over the batch, its apply (instances still searching keep their iterate for now), then ONE launch in which every
      // instance that is still searching finishes its own search and writes its own primal solution (k_ls.h).  Nothing here waits for the device.
      l.trial = 0; l.max_trials = max_trials;
      if (p0_enable && last && p0_x) { l.p0_x = p0_x; l.p0_u = p0_u; l.p0_mode = p0_mode; p0_done = true; bk.wbc_inputs_next(); }      // (the previous step's WBC has read its inputs before they are rewritten)
      if (rblk()) bk.launch(qm_ls_eval_kernel, (nodes_threads + 63) / 64, 64, LS_EVAL_LDS_BYTES, l); else bk.launch(qm_ls_eval_dense_kernel, (nodes_threads + 63) / 64, 64, LS_EVAL_LDS_BYTES, l);
      { QmLsArgs ls = l; ls.with_alpha = 1; ls.tickets = nullptr; bk.launch(qm_perf_sum_kernel, B, 64, 0, ls); }      // (nobody waits for the published count of the instances still searching: no ticket, no fences)
      if (max_trials > 1) { QmLsArgs lt = l; lt.trial = 1;
        if (rblk()) bk.launch(qm_ls_tail_kernel, B, LS_TAIL_BLOCK, LS_TAIL_LDS_BYTES(d.nmax), lt); else bk.launch(qm_ls_tail_dense_kernel, B, LS_TAIL_BLOCK, LS_TAIL_LDS_BYTES(d.nmax), lt); }
      ls_trials_pending = true; ls_trials_cap = max_trials;
      // the batch's apply reads the FINAL done / alpha of every instance.  A control step launches it behind its WBC (defer_apply): only the policy at t0 — written above by
      // the deciding kernels — is on the WBC's way, the primal solution on all nodes is not
      if (defer_apply && last) { pending_apply_B = B; pending_apply_threads = nodes_threads; solved_B = B; return; }
      bk.launch(qm_ls_apply_kernel, (nodes_threads * 30 + 255) / 256, 256, 0, l);
      if (!last) bk.launch(qm_ls_commit_kernel, (nodes_threads * 30 + 255) / 256, 256, 0, l);
      solved_B = B; return;
    } else {
    for (int t = 0; t < max_trials; ++t) {
      l.trial = t;
      if (ilqr) { ro.mode = 1; ro.trial = t; bk.launch(qm_ilqr_rollout_kernel, B, 64, 0, ro); }      // nonlinear rollout with feedback at the instance's step length
      if (ipm) bk.launch(qm_ls_eval_ipm_kernel, (nodes_threads + 63) / 64, 64, LS_EVAL_LDS_BYTES, l);
      else if (rblk()) bk.launch(qm_ls_eval_kernel, (nodes_threads + 63) / 64, 64, LS_EVAL_LDS_BYTES, l);
      else bk.launch(qm_ls_eval_dense_kernel, (nodes_threads + 63) / 64, 64, LS_EVAL_LDS_BYTES, l);
      d.host_open[t] = -1;                                 // armed: the launch's last block overwrites it with the count of the instances still searching
      { QmLsArgs ls = l; ls.with_alpha = 1; bk.launch(qm_perf_sum_kernel, B, 64, 0, ls); }   // trial merit + filter decision + count of the instances still searching
      ++ls_trials_run;
      // The first trial is accepted by every instance most of the time: its apply is enqueued BEHIND the decision before the host knows the outcome, so the device does not
      // idle through the host's round trip (flag -> launch: 20-30 us per step in the kernel trace).  The apply only reads the iterate and the step and writes the primal
      // solution — instances still searching get alpha = 0 — so it is simply launched again once the remaining trials are through.
      // INVARIANT this rests on: between the speculative apply and the final one NOTHING reads or writes xs / us, and no trial kernel writes what the apply reads of an instance
      // that is done — qm_ls_eval / qm_perf_sum / the iLQR trial rollouts read x, u, dx, du (xt, ut) and write perf / alpha / done / xt, ut only, and skip instances with
      // done != 0, so an accepted instance's xt / ut / alpha stay intact through the later trials.  A new trial kernel that touches xs / us, or a rollout that overwrites a done
      // instance's xt / ut, breaks it (tests/test_emu_kernels.py::test_speculative_apply_is_idempotent runs mixed batches against the non-speculative order, SQP and iLQR).
      if (t == 0 && speculative_apply) { bk.launch(qm_ls_apply_kernel, (nodes_threads * 30 + 255) / 256, 256, 0, l); }
      bk.wait_flag(d.host_open + t, -1);                   // spin on the host-visible word (a stream synchronisation costs 10-30 us of wake-up latency per step)
      if (d.host_open[t] == 0) break;
    }
    if (ls_trials_run != 1 || !speculative_apply) bk.launch(qm_ls_apply_kernel, (nodes_threads * 30 + 255) / 256, 256, 0, l);
    }
    if (!last) bk.launch(qm_ls_commit_kernel, (nodes_threads * 30 + 255) / 256, 256, 0, l);
    if (ipm) { bk.launch(qm_ipm_commit_kernel, (int)(((size_t)d.nmax * B * QM_NH + 255) / 256), 256, 0, ia); bk.launch(qm_ipm_barrier_kernel, (B + 63) / 64, 64, 0, ia); }      // accepted slack / dual step, barrier update
    solved_B = B;
  }
};
