// qm_sim_pipeline.h — launches of the batched rigid-body plant (backend-templated like qm_pipeline.h)
#pragma once
#include "qm_pipeline.h"
#include "qm_wbc_pipeline.h"
#include "../kernels/k_sim.h"
#include "../kernels/k_loop.h"

struct QmSimBuffers {
  int Bmax = 0;
  double* q = nullptr; double* v = nullptr; double* time = nullptr; double* cmd = nullptr; double* ring = nullptr; int* ring_n = nullptr;
  double* rbd = nullptr; int* contact = nullptr; double* force = nullptr; int* status = nullptr;
  double* arm_hold = nullptr; double* arm_last = nullptr;     // QMMpcController: held arm position commands and last_time_ per joint ([B][6])
  // published policy (pipelined loop): what evaluatePolicy reads while the MPC writes its next solution — OCS2 guards this buffer with a mutex, here it is a copy
  int p_nmax = 0, p_nev = 0; double* p_xs = nullptr; double* p_us = nullptr; double* p_node_t = nullptr; int* p_node_ev = nullptr; int* p_n_nodes = nullptr; double* p_ev = nullptr; int* p_modes = nullptr;
  bool p_valid = false;
};

template <class BK>
struct QmSimPipeline {
  BK& bk; QmSimBuffers s; QmSimParams p;
  int controller = 0;       // 0: QMController (HierarchicalWbc, arm torque + PD), 1: QMMpcController (HierarchicalMpcWbc, arm position commands at 100 Hz)
  explicit QmSimPipeline(BK& b) : bk(b) { p.k_n = 4.0e4; p.d_n = 200.0; p.mu = 0.8; p.v_eps = 1.0e-2; p.foot_radius = 0.02; p.delay = 0.009; p.saturate = 1; }
  template <class T> T* A(size_t n) { T* ptr = (T*)bk.alloc(n * sizeof(T)); bk.zero(ptr, n * sizeof(T)); return ptr; }
  void allocate(int Bmax) {
    if (s.Bmax) return;
    s.Bmax = Bmax; s.q = A<double>((size_t)Bmax * 24); s.v = A<double>((size_t)Bmax * 24); s.time = A<double>(Bmax); s.cmd = A<double>((size_t)Bmax * (QM_SIM_CMD - 1));
    s.ring = A<double>((size_t)Bmax * QM_SIM_SLOTS * QM_SIM_CMD); s.ring_n = A<int>((size_t)Bmax * 2); s.rbd = A<double>((size_t)Bmax * QM_NRBD); s.contact = A<int>((size_t)Bmax * 4);
    s.force = A<double>((size_t)Bmax * 12); s.status = A<int>(Bmax);
    s.arm_hold = A<double>((size_t)Bmax * 6); s.arm_last = A<double>((size_t)Bmax * 6);
  }
  void release() { void* ps[] = {s.q, s.v, s.time, s.cmd, s.ring, s.ring_n, s.rbd, s.contact, s.force, s.status, s.arm_hold, s.arm_last, s.p_xs, s.p_us, s.p_node_t, s.p_node_ev, s.p_n_nodes, s.p_ev, s.p_modes}; for (void* ptr : ps) if (ptr) bk.free(ptr); s = QmSimBuffers(); }
  // "Simulation reset" of QMHWSim::writeSim: state set, delay buffer and held command cleared
  void reset(int B, const double* q_host, const double* v_host, const double* time_host) {
    s.p_valid = false;
    bk.to_device(s.q, q_host, (size_t)B * 24 * 8); bk.to_device(s.v, v_host, (size_t)B * 24 * 8); bk.to_device(s.time, time_host, (size_t)B * 8);
    bk.zero(s.ring_n, (size_t)s.Bmax * 2 * sizeof(int)); bk.zero(s.cmd, (size_t)s.Bmax * (QM_SIM_CMD - 1) * 8);
    QmArmResetArgs r; r.B = B; r.q = s.q; r.time = s.time; r.arm_hold = s.arm_hold; r.arm_last = s.arm_last; bk.launch(qm_arm_reset_kernel, (B * 6 + 63) / 64, 64, 0, r);
  }
  void set_command(int B, const double* cmd_host) { bk.to_device(s.cmd, cmd_host, (size_t)B * (QM_SIM_CMD - 1) * 8); }
  // copy of the solver's primal solution + grid + mode schedule into the published-policy buffers
  void publish_policy(const QmMpcBuffers& d) {
    const size_t NB = (size_t)d.nmax * d.Bmax;
    if (!s.p_xs) { s.p_nmax = d.nmax; s.p_nev = d.nev; s.p_xs = A<double>(NB * 30); s.p_us = A<double>(NB * 30); s.p_node_t = A<double>(NB); s.p_node_ev = A<int>(NB); s.p_n_nodes = A<int>(d.Bmax);
                   s.p_ev = A<double>((size_t)d.Bmax * d.nev); s.p_modes = A<int>((size_t)d.Bmax * (d.nev + 1)); }
    bk.copy_dd(s.p_xs, d.xs, NB * 30 * 8); bk.copy_dd(s.p_us, d.us, NB * 30 * 8); bk.copy_dd(s.p_node_t, d.node_t, NB * 8); bk.copy_dd(s.p_node_ev, d.node_ev, NB * 4); bk.copy_dd(s.p_n_nodes, d.n_nodes, (size_t)d.Bmax * 4);
    bk.copy_dd(s.p_ev, d.ev, (size_t)d.Bmax * d.nev * 8); bk.copy_dd(s.p_modes, d.modes, (size_t)d.Bmax * (d.nev + 1) * 4);
    s.p_valid = true;
  }
  // currentObservation_ of the MPC (x0, t0) from the plant state
  void observe(const QmMpcBuffers& d, int B) { QmObserveArgs o; o.mb = d.mb; o.B = B; o.rbd = s.rbd; o.time = s.time; o.x0 = d.x0; o.t0 = d.t0; bk.launch(qm_observe_kernel, (B + 63) / 64, 64, 0, o); }
  // hybrid joint command from the evaluated policy and the WBC torques
  void command(int B, const double* x_des, const double* u_des, const double* wbc_out, double arm_kp, double arm_kd) {
    QmCommandArgs c; c.B = B; c.x_des = x_des; c.u_des = u_des; c.wbc_out = wbc_out; c.time = s.time; c.arm_kp = arm_kp; c.arm_kd = arm_kd; c.cmd = s.cmd;
    c.controller = controller; c.rbd = s.rbd; c.arm_hold = s.arm_hold; c.arm_last = s.arm_last;
    bk.launch(qm_command_kernel, (B * QM_NJ + 63) / 64, 64, 0, c);
  }
  // rbd state / contact flags of the current plant state without advancing it (nsub = 0 is not a step: the delay buffer is left alone)
  void step(const double* mb_dev, int B, double period, int nsub) {
    QmSimArgs a; a.mb = mb_dev; a.B = B; a.nsub = nsub; a.h = nsub > 0 ? period / nsub : 0.0; a.p = p; a.q = s.q; a.v = s.v; a.time = s.time; a.cmd = s.cmd; a.ring = s.ring; a.ring_n = s.ring_n;
    a.rbd = s.rbd; a.contact = s.contact; a.force = s.force; a.status = s.status;
    bk.launch(qm_sim_kernel, B, 64, SIM_LDS_BYTES, a);   // one wavefront per instance
  }
};

// n_ticks of the whole controller around the plant (QMController::update + the body of mpcThread_, qm_controllers/src/QMController.cpp:128-175, 315-332), all on
// resident data: state estimate (the plant's state) -> every mpc_every ticks an MPC call on that observation (pre_mpc: the gait front-end's schedule refresh,
// then a warm-started solve with sqp_iters SQP iterations) -> policy at the plant time -> WBC on the measured state -> hybrid joint command -> simulation step.
// Shared by the product (qmhip_closed_loop_sim) and the host emulator of the tests.
template <class BK, class PreMpc>
void qm_closed_loop_sim_ticks(BK& bk, QmMpcPipeline<BK>& mpc, QmWbcPipeline<BK>& wbc, QmSimPipeline<BK>& sim, long& sim_ticks, int B, int n_ticks, double period, int n_substeps,
                              int mpc_every, double horizon, double arm_kp, double arm_kd, int sqp_iters, PreMpc pre_mpc) {
  for (int k = 0; k < n_ticks; ++k) {
    if ((sim_ticks % mpc_every) == 0) {
      sim.observe(mpc.d, B);
      pre_mpc();
      mpc.grid(B, horizon, true); for (int it = 0; it < sqp_iters; ++it) mpc.sqp_iteration(B, 14, it + 1 == sqp_iters);
    }
    bk.launch(qm_policy_kernel, (B + 63) / 64, 64, 0, wbc.pargs(mpc.d, B, sim.s.time));
    // first tick after a reset: inputLast_ primed with the planned input (the reference's WBC has been running since time 0 when the legs are switched on at
    // time 10): zero joint acceleration
    if (sim_ticks == 0) bk.copy_dd(wbc.w.input_last, wbc.w.u_des, (size_t)B * 30 * 8);
    wbc.step(mpc.d, B, period, sim.controller == 1 ? 1 : 0, sim.s.rbd, sim.s.time);      // QMMpcController::setupWbc installs HierarchicalMpcWbc (QMController.cpp:410-414)
    sim.command(B, wbc.w.x_des, wbc.w.u_des, wbc.w.out, arm_kp, arm_kd);
    sim.step(mpc.d.mb, B, period, n_substeps);
    ++sim_ticks;
  }
}

// The same loop with the MPC beside the control ticks, as the reference's mpcThread_ runs beside QMController::update: the MPC call triggered at a tick observes
// the plant at that tick and computes on its own stream while the next mpc_every ticks run on the other one with the policy published before; its solution is
// published (copied into the policy buffers evaluatePolicy reads) when those ticks are done, i.e. it is used one MPC period after its observation.  The very
// first call is synchronous (there is no policy yet).  n_ticks and the tick counter must be multiples of mpc_every.  BK::stream_select(s) makes stream s (0: MPC,
// 1: ticks) the target of the following launches, BK::stream_order(a, b) orders everything launched so far on a before everything launched later on b; the
// host emulator runs the same sequence on one queue (identical results: the data dependencies are the same).
template <class BK, class PreMpc>
void qm_closed_loop_sim_pipelined(BK& bk, QmMpcPipeline<BK>& mpc, QmWbcPipeline<BK>& wbc, QmSimPipeline<BK>& sim, long& sim_ticks, int B, int n_ticks, double period, int n_substeps,
                                  int mpc_every, double horizon, double arm_kp, double arm_kd, int sqp_iters, PreMpc pre_mpc) {
  auto solve = [&]() { pre_mpc(); mpc.grid(B, horizon, true); for (int it = 0; it < sqp_iters; ++it) mpc.sqp_iteration(B, 14, it + 1 == sqp_iters); };
  auto ticks = [&]() {
    for (int k = 0; k < mpc_every; ++k) {
      QmPolicyArgs pa = wbc.pargs(mpc.d, B, sim.s.time); pa.n_nodes = sim.s.p_n_nodes; pa.node_t = sim.s.p_node_t; pa.node_ev = sim.s.p_node_ev; pa.xs = sim.s.p_xs; pa.us = sim.s.p_us; pa.ev = sim.s.p_ev; pa.modes = sim.s.p_modes;
      bk.launch(qm_policy_kernel, (B + 63) / 64, 64, 0, pa);
      if (sim_ticks == 0) bk.copy_dd(wbc.w.input_last, wbc.w.u_des, (size_t)B * 30 * 8);
      wbc.step(mpc.d, B, period, sim.controller == 1 ? 1 : 0, sim.s.rbd, sim.s.time);
      sim.command(B, wbc.w.x_des, wbc.w.u_des, wbc.w.out, arm_kp, arm_kd);
      sim.step(mpc.d.mb, B, period, n_substeps);
      ++sim_ticks;
    }
  };
  for (int p = 0; p < n_ticks / mpc_every; ++p) {
    bk.stream_select(1); sim.observe(mpc.d, B); bk.stream_order(1, 0);
    if (!sim.s.p_valid) {                       // no policy yet: solve first, then run the ticks of this period on it
      bk.stream_select(0); solve(); bk.stream_order(0, 1);
      bk.stream_select(1); sim.publish_policy(mpc.d); ticks();
    } else {
      bk.stream_select(1); ticks();             // enqueued first: they run while the host waits inside the solve's line search
      bk.stream_select(0); solve(); bk.stream_order(0, 1);
      bk.stream_select(1); sim.publish_policy(mpc.d);
    }
    bk.stream_order(1, 0);                      // the next solve must not overwrite the solution before it has been published
  }
  bk.stream_select(0);
}
