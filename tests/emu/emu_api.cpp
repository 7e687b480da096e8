// tests/emu/emu_api.cpp — TEST INFRASTRUCTURE: runs the product's kernel sequence (qm_pipeline.h) on the host
// emulator and exposes it to pytest through ctypes.  Never linked into the product.
#include "hip_emu.h"
#include "../../qm_control_amd/csrc/host/qm_pipeline.h"
#include "../../qm_control_amd/csrc/host/qm_wbc_pipeline.h"
#include "../../qm_control_amd/csrc/host/qm_sim_pipeline.h"
#include "../../qm_control_amd/csrc/host/qm_front_pipeline.h"
#include "../../qm_control_amd/csrc/host/qm_hoqp_pipeline.h"

struct EmuBackend {
  template <class K, class A> void launch(K kernel, int grid, int block, size_t, const A& args) { emu::launch(dim3(grid), dim3(block), [&]() { kernel(args); }); }
  void* alloc(size_t n) { return malloc(n ? n : 8); }
  void free(void* p) { ::free(p); }
  void zero(void* p, size_t n) { memset(p, 0, n); }
  void to_device(void* d, const void* s, size_t n) { memcpy(d, s, n); }
  void to_host(void* d, const void* s, size_t n) { memcpy(d, s, n); }
  void sync() {}
  void* alloc_mapped(size_t n, void** host_view) { void* p = malloc(n ? n : 8); *host_view = p; return p; }
  void free_mapped(void* p) { ::free(p); }
  void wait_launched() {}
  void wait_flag(volatile int*, int) {}
  void wbc_inputs_next() {}
  void stream_select(int) {}
  void stream_order(int, int) {}
  void copy_dd(void* d, const void* s, size_t n) { memcpy(d, s, n); }
};

struct EmuCtx { bool fused_policy = true; double* lqdbg_stash = nullptr; EmuBackend bk; QmMpcPipeline<EmuBackend> mpc; QmWbcPipeline<EmuBackend> wbc; QmFrontPipeline<EmuBackend> front; QmSimPipeline<EmuBackend> sim; EmuCtx() : mpc(bk), wbc(bk), front(bk), sim(bk) {} };

extern "C" {
void* emu_create(const double* mb, const double* st, int Bmax, int nmax, int nref, int nev) {
  EmuCtx* c = new EmuCtx(); c->mpc.allocate(mb, st, Bmax, nmax, nref, nev, true); c->wbc.allocate(Bmax, true); c->front.allocate(Bmax); c->front.phase_transition_stance_time = st[ST_PHASE_TRANS_STANCE]; return c;
}
void emu_set_solver(void* h, int solver) { ((EmuCtx*)h)->mpc.solver = solver; }      // 0 SQP, 1 discrete iLQR (qmhip_set_setting(ST_SOLVER, .))
// 0: run the PRODUCT instance of the LQ kernel (qm_lq_kernel: no debug records, no cycle stamps); 1 (default): qm_lq_dbg_kernel.  qmhip_debug_set("lq_debug", .)
void emu_set_lq_debug(void* h, int on) { EmuCtx* c = (EmuCtx*)h; if (!c->lqdbg_stash) c->lqdbg_stash = c->mpc.d.lqdbg; c->mpc.d.lqdbg = on ? c->lqdbg_stash : nullptr; }
void emu_set_r_dense(void* h, int on) { ((EmuCtx*)h)->mpc.r_force_dense = on != 0; }      // qmhip_debug_set("r_dense", .)
int emu_r_blocks(void* h) { return ((EmuCtx*)h)->mpc.rblk() ? 1 : 0; }
void emu_set_riccati_skip(void* h, int mask) { ((EmuCtx*)h)->mpc.riccati_skip = mask; }      // qmhip_debug_set("riccati_skip", .): 20 leaves K1b's stage records untouched
void emu_destroy(void* h) { EmuCtx* c = (EmuCtx*)h; if (c->lqdbg_stash) c->mpc.d.lqdbg = c->lqdbg_stash; c->mpc.release(); c->wbc.release(); c->front.release(); c->sim.release(); delete c; }
int emu_mpc_step(void* h, int B, const double* t0, const double* x0, const double* ref_t, const double* ref_x, const double* ev, const int* modes, double horizon, int max_trials) {
  EmuCtx* c = (EmuCtx*)h;
  c->mpc.upload_inputs(B, t0, x0, ref_t, ref_x, ev, modes);
  c->mpc.grid(B, horizon);
  c->mpc.sqp_iteration(B, max_trials);
  return c->mpc.ls_trials();
}
// receding horizon: new observation + warm-started iteration / perfect-tracking advance (same calls as the qmhip_* entry points)
int emu_mpc_step_warm(void* h, int B, const double* t0, const double* x0, double horizon, int max_trials) {
  EmuCtx* c = (EmuCtx*)h; QmMpcBuffers& d = c->mpc.d;
  if (t0) memcpy(d.t0, t0, (size_t)B * 8); if (x0) memcpy(d.x0, x0, (size_t)B * 30 * 8);
  c->mpc.grid(B, horizon, true); c->mpc.sqp_iteration(B, max_trials); return c->mpc.ls_trials();
}
// one more iteration on the committed iterate of the last call (sqp.sqpIteration / ipm.ipmIteration > 1: what qmhip_mpc_solve_resident loops over)
int emu_mpc_iterate(void* h, int B, int max_trials) { EmuCtx* c = (EmuCtx*)h; c->mpc.sqp_iteration(B, max_trials); return c->mpc.ls_trials(); }
void emu_advance(void* h, int B, double dt) { ((EmuCtx*)h)->mpc.advance(B, dt); }
// K0 only: grid, modes, references, initial guess of the uploaded problem
void emu_grid(void* h, int B, double horizon) { ((EmuCtx*)h)->mpc.grid(B, horizon); }
void emu_upload(void* h, int B, const double* t0, const double* x0, const double* ref_t, const double* ref_x, const double* ev, const int* modes) { ((EmuCtx*)h)->mpc.upload_inputs(B, t0, x0, ref_t, ref_x, ev, modes); }
// raw buffer access for parity tests: name -> pointer
void* emu_buffer(void* h, const char* name) {
  QmMpcBuffers& d = ((EmuCtx*)h)->mpc.d;
  { EmuCtx* c = (EmuCtx*)h; if (!strcmp(name, "sim_q")) return (void*)c->sim.s.q; if (!strcmp(name, "sim_v")) return (void*)c->sim.s.v; if (!strcmp(name, "wbc_out")) return (void*)c->wbc.w.out; if (!strcmp(name, "wbc_qp_status")) return (void*)c->wbc.w.qp_status;
    if (const void* p = c->wbc.buffer(name)) return (void*)p; }      // wbc_x_des, wbc_u_des, wbc_mode, wbc_rbd, ...
#define F(n) if (!strcmp(name, #n)) return (void*)d.n;
  F(n_nodes) F(node_t) F(node_ts) F(node_dt) F(node_ev) F(node_mode) F(zvel) F(zpos) F(xref) F(eeref) F(status) F(x) F(u) F(dx) F(du) F(stage) F(lqdbg) F(perf) F(base_sum)
  F(perf_sum) F(step_info) F(alpha) F(done) F(xs) F(us) F(out_perf) F(t0) F(x0) F(ipm_s) F(ipm_l) F(ipm_ds) F(ipm_dl) F(ipm_info)
#undef F
  return nullptr;
}
void emu_policy_eval(void* h, int B, const double* t, double* xd, double* ud, int* mode) {
  EmuCtx* c = (EmuCtx*)h; c->wbc.policy_eval(c->mpc.d, B, t);
  memcpy(xd, c->wbc.w.x_des, (size_t)B * 30 * 8); memcpy(ud, c->wbc.w.u_des, (size_t)B * 30 * 8); memcpy(mode, c->wbc.w.mode, (size_t)B * 4);
}
void emu_wbc_reset(void* h) { ((EmuCtx*)h)->wbc.reset(); }
// batched rigid-body plant (same calls as the qmhip_sim_* entry points)
void emu_sim_params(void* h, const double* p) { QmSimParams& q = ((EmuCtx*)h)->sim.p; q.k_n = p[0]; q.d_n = p[1]; q.mu = p[2]; q.v_eps = p[3]; q.foot_radius = p[4]; q.delay = p[5]; q.saturate = p[6] != 0.0; }
void emu_sim_reset(void* h, int B, const double* q, const double* v, const double* time) { EmuCtx* c = (EmuCtx*)h; c->sim.allocate(c->mpc.d.Bmax); c->sim.reset(B, q, v, time); }
void emu_sim_command(void* h, int B, const double* cmd90) { ((EmuCtx*)h)->sim.set_command(B, cmd90); }
static long g_emu_sim_ticks = 0;
void emu_sim_set_controller(void* h, int kind) { ((EmuCtx*)h)->sim.controller = kind; }
void emu_closed_loop_sim(void* h, int B, int n_ticks, double period, int nsub, int mpc_every, double horizon, double arm_kp, double arm_kd, int restart) {
  EmuCtx* c = (EmuCtx*)h; if (restart) { g_emu_sim_ticks = 0; c->sim.step(c->mpc.d.mb, B, 0.0, 0); }
  qm_closed_loop_sim_ticks(c->bk, c->mpc, c->wbc, c->sim, g_emu_sim_ticks, B, n_ticks, period, nsub, mpc_every, horizon, arm_kp, arm_kd, 1, []() {});
}
void emu_closed_loop_sim_pipelined(void* h, int B, int n_ticks, double period, int nsub, int mpc_every, double horizon, double arm_kp, double arm_kd, int restart) {
  EmuCtx* c = (EmuCtx*)h; if (restart) { g_emu_sim_ticks = 0; c->sim.s.p_valid = false; c->sim.step(c->mpc.d.mb, B, 0.0, 0); }
  qm_closed_loop_sim_pipelined(c->bk, c->mpc, c->wbc, c->sim, g_emu_sim_ticks, B, n_ticks, period, nsub, mpc_every, horizon, arm_kp, arm_kd, 1, []() {});
}
void emu_sim_step(void* h, int B, double period, int nsub, double* rbd, int* contact, double* q, double* v, double* force, int* status) {
  EmuCtx* c = (EmuCtx*)h; c->sim.step(c->mpc.d.mb, B, period, nsub);
  memcpy(rbd, c->sim.s.rbd, (size_t)B * QM_NRBD * 8); memcpy(contact, c->sim.s.contact, (size_t)B * 16); memcpy(q, c->sim.s.q, (size_t)B * 24 * 8); memcpy(v, c->sim.s.v, (size_t)B * 24 * 8);
  memcpy(force, c->sim.s.force, (size_t)B * 12 * 8); memcpy(status, c->sim.s.status, (size_t)B * 4);
}
void emu_wbc_step(void* h, int B, const double* xd, const double* ud, const double* rbd, const int* mode, double period, const double* time, int variant, double* out, int* status, double* dbg) {
  EmuCtx* c = (EmuCtx*)h; c->wbc.upload(B, xd, ud, rbd, mode, time); c->wbc.step(c->mpc.d, B, period, variant);
  memcpy(out, c->wbc.w.out, (size_t)B * QM_NWBC_OUT * 8); memcpy(status, c->wbc.w.qp_status, (size_t)B * 3 * 4); if (dbg) memcpy(dbg, c->wbc.w.dbg, (size_t)B * WBC_DBG_SIZE * 8);
}
// the benchmark's whole control step on resident data (same calls as qmhip_control_step_resident)
void emu_control_step(void* h, int B, double horizon, double period, double time, double* out, int* status, double* rbd_out) {
  // as qmhip_control_step_resident: measured state first, the policy at t0 from the line search's deciding kernels, the WBC, then the batch's apply
  EmuCtx* c = (EmuCtx*)h; c->mpc.grid(B, horizon);
  c->mpc.p0_x = c->wbc.w.x_des; c->mpc.p0_u = c->wbc.w.u_des; c->mpc.p0_mode = c->wbc.w.mode; c->mpc.p0_enable = c->fused_policy; c->mpc.defer_apply = c->fused_policy;
  c->wbc.measured_from_x0(c->mpc.d, B, time);
  c->mpc.sqp_iteration(B, 14, true); c->mpc.p0_enable = false; c->mpc.defer_apply = false;
  if (!c->mpc.p0_done) c->wbc.policy_eval_at_t0(c->mpc.d, B);
  c->wbc.step(c->mpc.d, B, period, 0); c->mpc.apply_pending();
  memcpy(out, c->wbc.w.out, (size_t)B * QM_NWBC_OUT * 8); memcpy(status, c->wbc.w.qp_status, (size_t)B * 3 * 4); if (rbd_out) memcpy(rbd_out, c->wbc.w.rbd, (size_t)B * QM_NRBD * 8);
}
// reference / gait front-end (same calls as the qmhip_gait_* / qmhip_target_* entry points)
void emu_gait_set_templates(void* h, int G, const int* n_phases, const double* times, const int* modes) { ((EmuCtx*)h)->front.set_templates(G, n_phases, times, modes); }
void emu_gait_reset(void* h, int B, int n0, const double* ev0, const int* mode0, int tpl0) { EmuCtx* c = (EmuCtx*)h; c->front.gait_reset(B, n0, ev0, mode0, tpl0); c->mpc.front_status = c->front.f.gs_status; c->mpc.front_B = B; }   // as qmhip_gait_reset does
void emu_gait_insert(void* h, int B, const int* tpl, const double* start, const double* final_t) { ((EmuCtx*)h)->front.gait_insert(B, tpl, start, final_t); }
void emu_gait_update(void* h, int B, const double* t0, double horizon) { EmuCtx* c = (EmuCtx*)h; memcpy(c->mpc.d.t0, t0, (size_t)B * 8); c->front.gait_schedule(c->mpc.d, B, horizon); }
void emu_gait_download(void* h, int B, int* n, double* ev, int* mode, int* tpl, int* status) { ((EmuCtx*)h)->front.gait_download(B, n, ev, mode, tpl, status); }
void emu_schedule_download(void* h, int B, double* ev, int* modes) { EmuCtx* c = (EmuCtx*)h; memcpy(ev, c->mpc.d.ev, (size_t)B * c->mpc.d.nev * 8); memcpy(modes, c->mpc.d.modes, (size_t)B * (c->mpc.d.nev + 1) * 4); }
void emu_target_reset(void* h, int B, const double* last7) { ((EmuCtx*)h)->front.target_reset(B, last7); }
void emu_target_from_command(void* h, int B, const double* t0, const double* x0, const int* kind, const double* cmd, const double* ee, int thru_float, double T, double vd, double vr, double ch) {
  EmuCtx* c = (EmuCtx*)h; memcpy(c->mpc.d.t0, t0, (size_t)B * 8); memcpy(c->mpc.d.x0, x0, (size_t)B * 30 * 8);
  c->front.target_from_command(c->mpc.d, B, kind, cmd, ee, thru_float, T, vd, vr, ch);
}
void emu_target_download(void* h, int B, double* rt, double* rx, double* last) {
  EmuCtx* c = (EmuCtx*)h; memcpy(rt, c->mpc.d.ref_t, (size_t)B * c->mpc.d.nref * 8); memcpy(rx, c->mpc.d.ref_x, (size_t)B * c->mpc.d.nref * QM_NREF * 8); memcpy(last, c->front.f.last_ee, (size_t)B * 7 * 8);
}
// device math helpers on the host (tests/test_device_math.py)
// qmhip_hoqp_solve on the host emulator: the general HoQp kernel (k_hoqp.h)
int emu_hoqp(int B, int n_levels, int n, const int* ma, const int* md, const double* A, const double* b, const double* D, const double* f, double* x, int* status) {
  if (!QmHoqpPipeline<EmuBackend>::shapes_ok(n_levels, n, ma, md)) return -1;
  static EmuBackend bk; static QmHoqpPipeline<EmuBackend> h(bk);      // ONE pipeline for the process, like the context's: successive shapes go through its capacity bookkeeping
  h.solve(B, n_levels, n, ma, md, A, b, D, f, x, status); return 0;
}
void emu_set_speculative_apply(void* h, int on) { ((EmuCtx*)h)->mpc.speculative_apply = on != 0; }
void emu_set_fused_policy(void* h, int on) { ((EmuCtx*)h)->fused_policy = on != 0; }      // 0: qm_policy_kernel behind the apply (rounds 1-5)
void emu_set_device_tail(void* h, int on) { ((EmuCtx*)h)->mpc.device_tail = on != 0; }      // qmhip_debug_set("ls_device_tail", .)
// the C ABI's status of an instance from K0's word and K3's step_info (the mapping qmhip_mpc_download applies)
int emu_mpc_status(int k0_status, const double* step_info4, int strict) { return qm_mpc_status(k0_status, step_info4, strict != 0); }
void emu_sincos(int n, const double* x, double* sn, double* cs) { for (int i = 0; i < n; ++i) qm_sincos(x[i], sn[i], cs[i]); }
void emu_frcp(int n, const double* x, double* r) { for (int i = 0; i < n; ++i) r[i] = qm_frcp(x[i]); }
void emu_log(int n, const double* x, double* r) { for (int i = 0; i < n; ++i) r[i] = qm_log(x[i]); }
int emu_sizes(int which) { int v[] = {SR_SIZE, LQ_DBG_SIZE, PF_SIZE, LQ_LDS_BYTES, RW_LDS_BYTES, WBC_DBG_SIZE, LQ_KIN_LDS_BYTES, LS_EVAL_LDS_BYTES, WBC_LDS_BYTES, SIM_LDS_BYTES}; return v[which]; }
}
