// qm_front_pipeline.h — reference / gait front-end launches (backend-templated like qm_pipeline.h): device-resident
// GaitSchedule of every instance and the command -> TargetTrajectories conversion, both feeding K0's input buffers directly.
#pragma once
#include "qm_pipeline.h"
#include "../kernels/k_front.h"

#define QM_GAIT_EVENT_SLOTS 256       /* per-instance schedule capacity: window [t − T, t + 2T] of the busiest gait at N = 150 is < 64 events */

struct QmFrontBuffers {
  int Bmax = 0, n_gaits = 0;
  int* tpl_n = nullptr; double* tpl_times = nullptr; int* tpl_modes = nullptr;          // template table
  int* gs_n = nullptr; double* gs_ev = nullptr; int* gs_mode = nullptr; int* gs_tpl = nullptr; int* gs_status = nullptr;
  int* req_tpl = nullptr; double* req_start = nullptr; double* req_final = nullptr;       // insert requests
  double* ev0 = nullptr; int* mode0 = nullptr;                                           // initial schedule (reset)
  int* cmd_kind = nullptr; double* cmd = nullptr; double* ee_state = nullptr; double* last_ee = nullptr;
};

template <class BK>
struct QmFrontPipeline {
  BK& bk; QmFrontBuffers f; double phase_transition_stance_time = 0.0;
  explicit QmFrontPipeline(BK& b) : bk(b) {}
  template <class T> T* A(size_t n) { T* p = (T*)bk.alloc(n * sizeof(T)); bk.zero(p, n * sizeof(T)); return p; }
  void allocate(int Bmax) {
    f.Bmax = Bmax;
    f.gs_n = A<int>(Bmax); f.gs_ev = A<double>((size_t)QM_GAIT_EVENT_SLOTS * Bmax); f.gs_mode = A<int>((size_t)(QM_GAIT_EVENT_SLOTS + 1) * Bmax); f.gs_tpl = A<int>(Bmax); f.gs_status = A<int>(Bmax);
    f.req_tpl = A<int>(Bmax); f.req_start = A<double>(Bmax); f.req_final = A<double>(Bmax); f.ev0 = A<double>(QM_GAIT_EVENT_SLOTS); f.mode0 = A<int>(QM_GAIT_EVENT_SLOTS + 1);
    f.cmd_kind = A<int>(Bmax); f.cmd = A<double>((size_t)Bmax * 7); f.ee_state = A<double>((size_t)Bmax * 7); f.last_ee = A<double>((size_t)Bmax * 7);
  }
  void release() {
    void* ps[] = {f.tpl_n, f.tpl_times, f.tpl_modes, f.gs_n, f.gs_ev, f.gs_mode, f.gs_tpl, f.gs_status, f.req_tpl, f.req_start, f.req_final, f.ev0, f.mode0, f.cmd_kind, f.cmd, f.ee_state, f.last_ee};
    for (void* p : ps) if (p) bk.free(p);
    f = QmFrontBuffers();
  }
  // templates: n_phases[G], times[G][QM_GAIT_MAX_PHASES + 1], modes[G][QM_GAIT_MAX_PHASES] (host)
  void set_templates(int G, const int* n_phases, const double* times, const int* modes) {
    if (f.tpl_n) { bk.free(f.tpl_n); bk.free(f.tpl_times); bk.free(f.tpl_modes); }
    f.n_gaits = G; f.tpl_n = A<int>(G); f.tpl_times = A<double>((size_t)G * (QM_GAIT_MAX_PHASES + 1)); f.tpl_modes = A<int>((size_t)G * QM_GAIT_MAX_PHASES);
    bk.to_device(f.tpl_n, n_phases, (size_t)G * 4); bk.to_device(f.tpl_times, times, (size_t)G * (QM_GAIT_MAX_PHASES + 1) * 8); bk.to_device(f.tpl_modes, modes, (size_t)G * QM_GAIT_MAX_PHASES * 4);
  }
  QmGaitTable table() const { QmGaitTable T; T.n_gaits = f.n_gaits; T.n_phases = f.tpl_n; T.times = f.tpl_times; T.modes = f.tpl_modes; return T; }
  QmGaitState state(int B) const { QmGaitState s; s.B = B; s.cap = QM_GAIT_EVENT_SLOTS; s.n = f.gs_n; s.ev = f.gs_ev; s.mode = f.gs_mode; s.tpl = f.gs_tpl; s.status = f.gs_status; return s; }
  // NOTE the [slot][B] arrays are indexed with the B of the call: one batch size per reset
  void gait_reset(int B, int n0, const double* ev0, const int* mode0, int tpl0) {
    bk.to_device(f.ev0, ev0, (size_t)n0 * 8); bk.to_device(f.mode0, mode0, (size_t)(n0 + 1) * 4);
    QmGaitResetArgs a; a.s = state(B); a.n0 = n0; a.ev0 = f.ev0; a.mode0 = f.mode0; a.tpl0 = tpl0;
    bk.launch(qm_gait_reset_kernel, (B + 63) / 64, 64, 0, a);
  }
  void gait_insert(int B, const int* tpl, const double* start, const double* final_t) {
    bk.to_device(f.req_tpl, tpl, (size_t)B * 4); bk.to_device(f.req_start, start, (size_t)B * 8); bk.to_device(f.req_final, final_t, (size_t)B * 8);
    QmGaitInsertArgs a; a.T = table(); a.s = state(B); a.req_tpl = f.req_tpl; a.start = f.req_start; a.final_t = f.req_final; a.phase_transition_stance_time = phase_transition_stance_time;
    bk.launch(qm_gait_insert_kernel, (B + 63) / 64, 64, 0, a);
  }
  // getModeSchedule(t0 − T, t0 + 2T) of every instance -> the solver's ev / modes buffers
  void gait_schedule(QmMpcBuffers& d, int B, double horizon) {
    QmGaitScheduleArgs a; a.T = table(); a.s = state(B); a.t0 = d.t0; a.horizon = horizon; a.nev = d.nev; a.ev_out = d.ev; a.modes_out = d.modes;
    bk.launch(qm_gait_schedule_kernel, (B + 63) / 64, 64, 0, a);
  }
  void target_reset(int B, const double* last_ee7) { std::vector<double> h((size_t)B * 7); for (int b = 0; b < B; ++b) for (int q = 0; q < 7; ++q) h[(size_t)b * 7 + q] = last_ee7[q]; bk.to_device(f.last_ee, h.data(), h.size() * 8); }
  // kind[B], cmd[B][7], ee_state[B][7] or null (host); result goes to the solver's ref_t / ref_x buffers
  void target_from_command(QmMpcBuffers& d, int B, const int* kind, const double* cmd, const double* ee_state, int ee_through_float, double time_to_target, double disp_velocity, double rot_velocity, double com_height) {
    bk.to_device(f.cmd_kind, kind, (size_t)B * 4); bk.to_device(f.cmd, cmd, (size_t)B * 7 * 8);
    if (ee_state) bk.to_device(f.ee_state, ee_state, (size_t)B * 7 * 8);
    QmTargetArgs a; a.mb = d.mb; a.B = B; a.nref = d.nref; a.kind = f.cmd_kind; a.cmd = f.cmd; a.t0 = d.t0; a.x0 = d.x0; a.ee_state = ee_state ? f.ee_state : nullptr; a.ee_through_float = ee_through_float;
    a.time_to_target = time_to_target; a.disp_velocity = disp_velocity; a.rot_velocity = rot_velocity; a.com_height = com_height; a.last_ee = f.last_ee; a.ref_t = d.ref_t; a.ref_x = d.ref_x;
    bk.launch(qm_target_kernel, (B + 63) / 64, 64, 0, a);
  }
  // tests: schedule state of instance-major host arrays n[B], ev[B][cap], mode[B][cap + 1], tpl[B], status[B]
  void gait_download(int B, int* n, double* ev, int* mode, int* tpl, int* status) {
    const int cap = QM_GAIT_EVENT_SLOTS; std::vector<double> e((size_t)cap * B); std::vector<int> m((size_t)(cap + 1) * B);
    bk.to_host(n, f.gs_n, (size_t)B * 4); bk.to_host(tpl, f.gs_tpl, (size_t)B * 4); bk.to_host(status, f.gs_status, (size_t)B * 4);
    bk.to_host(e.data(), f.gs_ev, e.size() * 8); bk.to_host(m.data(), f.gs_mode, m.size() * 4);
    for (int b = 0; b < B; ++b) { for (int k = 0; k < cap; ++k) ev[(size_t)b * cap + k] = e[(size_t)k * B + b]; for (int k = 0; k <= cap; ++k) mode[(size_t)b * (cap + 1) + k] = m[(size_t)k * B + b]; }
  }
};
