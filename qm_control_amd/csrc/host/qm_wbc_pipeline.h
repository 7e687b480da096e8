// qm_wbc_pipeline.h — policy evaluation + whole-body controller launches (backend-templated like qm_pipeline.h)
#pragma once
#include "qm_pipeline.h"
#include "../kernels/k_policy.h"
#include "../kernels/k_wbc.h"

struct QmWbcBuffers {
  int Bmax = 0;
  double* t = nullptr; double* x_des = nullptr; double* u_des = nullptr; double* rbd = nullptr; int* mode = nullptr; double* time = nullptr;
  double* input_last = nullptr; double* out = nullptr; int* qp_status = nullptr; double* scratch = nullptr; double* dbg = nullptr;
};

template <class BK>
struct QmWbcPipeline {
  BK& bk; QmWbcBuffers w;
  explicit QmWbcPipeline(BK& b) : bk(b) {}
  template <class T> T* A(size_t n) { T* p = (T*)bk.alloc(n * sizeof(T)); bk.zero(p, n * sizeof(T)); return p; }
  // The tick inputs [x_des | u_des | rbd | time | mode] and the outputs [out | qp_status] are carved from ONE allocation each, in this order, so that a
  // batch of exactly Bmax instances (the ros_control plugin: 1) moves with one host copy per direction (qmhip_wbc_step)
  static size_t in_bytes(int B) { return (size_t)B * ((30 + 30 + QM_NRBD + 1) * sizeof(double) + sizeof(int)); }
  static size_t out_bytes(int B) { return (size_t)B * (QM_NWBC_OUT * sizeof(double) + 3 * sizeof(int)); }
  void allocate(int Bmax, bool debug = false) {
    w.Bmax = Bmax; w.t = A<double>(Bmax);
    w.x_des = (double*)bk.alloc(in_bytes(Bmax)); bk.zero(w.x_des, in_bytes(Bmax)); w.u_des = w.x_des + (size_t)Bmax * 30; w.rbd = w.u_des + (size_t)Bmax * 30; w.time = w.rbd + (size_t)Bmax * QM_NRBD; w.mode = (int*)(w.time + Bmax);
    w.out = (double*)bk.alloc(out_bytes(Bmax)); bk.zero(w.out, out_bytes(Bmax)); w.qp_status = (int*)(w.out + (size_t)Bmax * QM_NWBC_OUT);
    w.input_last = A<double>((size_t)Bmax * 30);
    w.scratch = A<double>((size_t)Bmax * WBC_SCRATCH); w.dbg = debug ? A<double>((size_t)Bmax * WBC_DBG_SIZE) : nullptr;
  }
  void release() { void* ps[] = {w.t, w.x_des, w.input_last, w.out, w.scratch, w.dbg}; for (void* p : ps) if (p) bk.free(p); w = QmWbcBuffers(); }
  void reset() { bk.zero(w.input_last, (size_t)w.Bmax * 30 * 8); }
  const void* buffer(const char* name) const {
#define F(n) if (!strcmp(name, "wbc_" #n)) return (const void*)w.n;
    F(x_des) F(u_des) F(rbd) F(mode) F(time) F(input_last) F(out) F(qp_status) F(scratch) F(dbg)
#undef F
    return nullptr;
  }
  QmPolicyArgs pargs(const QmMpcBuffers& d, int B, const double* t_dev) {
    QmPolicyArgs p; p.mb = d.mb; p.B = B; p.nmax = d.nmax; p.nev = d.nev; p.n_nodes = d.n_nodes; p.node_t = d.node_t; p.node_ev = d.node_ev; p.xs = d.xs; p.us = d.us; p.ev = d.ev; p.modes = d.modes;
    p.t = t_dev; p.x_des = w.x_des; p.u_des = w.u_des; p.mode = w.mode; return p;
  }
  void policy_eval(const QmMpcBuffers& d, int B, const double* t_host) { bk.to_device(w.t, t_host, (size_t)B * 8); bk.launch(qm_policy_kernel, (B + 63) / 64, 64, 0, pargs(d, B, w.t)); }
  void policy_eval_at_t0(const QmMpcBuffers& d, int B) { bk.launch(qm_policy_kernel, (B + 63) / 64, 64, 0, pargs(d, B, d.t0)); }
  void measured_from_x0(const QmMpcBuffers& d, int B, double time) { QmMeasArgs m; m.mb = d.mb; m.B = B; m.x0 = d.x0; m.time = time; m.rbd = w.rbd; m.time_out = w.time; bk.launch(qm_measured_kernel, (B + 63) / 64, 64, 0, m); }
  void policy_at_t0_and_measured(const QmMpcBuffers& d, int B, double time) {
    QmPolicyMeasArgs a; a.p = pargs(d, B, d.t0); a.m.mb = d.mb; a.m.B = B; a.m.x0 = d.x0; a.m.time = time; a.m.rbd = w.rbd; a.m.time_out = w.time;
    bk.launch(qm_policy_measured_kernel, (2 * B + 63) / 64, 64, 0, a);
  }
  void upload(int B, const double* xd, const double* ud, const double* rbd, const int* mode, const double* time) {
    bk.to_device(w.x_des, xd, (size_t)B * 30 * 8); bk.to_device(w.u_des, ud, (size_t)B * 30 * 8); bk.to_device(w.rbd, rbd, (size_t)B * QM_NRBD * 8); bk.to_device(w.mode, mode, (size_t)B * 4); bk.to_device(w.time, time, (size_t)B * 8);
  }
  int wbc_stop = 0;   // profiling only
  // rbd_dev / time_dev: measured state and time from another resident producer (the plant); null: the uploaded / synthetic ones
  void step(const QmMpcBuffers& d, int B, double period, int variant, const double* rbd_dev = nullptr, const double* time_dev = nullptr) {
    QmWbcArgs a; a.mb = d.mb; a.st = d.st; a.B = B; a.x_des = w.x_des; a.u_des = w.u_des; a.rbd = w.rbd; a.mode = w.mode; a.time = w.time; a.period = period; a.variant = variant;
    a.input_last = w.input_last; a.out = w.out; a.qp_status = w.qp_status; a.scratch = w.scratch; a.dbg = w.dbg; a.stop = wbc_stop; if (rbd_dev) a.rbd = rbd_dev; if (time_dev) a.time = time_dev;
    if (wbc_stop != 0 || w.dbg) bk.launch(qm_wbc_prof_kernel, B, WBC_BLOCK, WBC_LDS_BYTES, a);   // instrumented instance: in-kernel cycle counters, early-return switches, debug records (profiling / parity tests)
    else bk.launch(qm_wbc_kernel, B, WBC_BLOCK, WBC_LDS_BYTES, a);   // one wavefront per instance
  }
};
