// tests/emu/emu_api.cpp — TEST INFRASTRUCTURE: runs the product's kernel sequence (qm_pipeline.h) on the host
// emulator and exposes it to pytest through ctypes.  Never linked into the product.
#include "hip_emu.h"
#include "../../qm_control_amd/csrc/host/qm_pipeline.h"

struct EmuBackend {
  template <class K, class A> void launch(K kernel, int grid, int block, size_t, const A& args) { emu::launch(dim3(grid), dim3(block), [&]() { kernel(args); }); }
  void* alloc(size_t n) { return malloc(n ? n : 8); }
  void free(void* p) { ::free(p); }
  void zero(void* p, size_t n) { memset(p, 0, n); }
  void to_device(void* d, const void* s, size_t n) { memcpy(d, s, n); }
  void to_host(void* d, const void* s, size_t n) { memcpy(d, s, n); }
  void sync() {}
};

struct EmuCtx { EmuBackend bk; QmMpcPipeline<EmuBackend> mpc; EmuCtx() : mpc(bk) {} };

extern "C" {
void* emu_create(const double* mb, const double* st, int Bmax, int nmax, int nref, int nev) {
  EmuCtx* c = new EmuCtx(); c->mpc.allocate(mb, st, Bmax, nmax, nref, nev, true); return c;
}
void emu_destroy(void* h) { EmuCtx* c = (EmuCtx*)h; c->mpc.release(); delete c; }
int emu_mpc_step(void* h, int B, const double* t0, const double* x0, const double* ref_t, const double* ref_x, const double* ev, const int* modes, double horizon, int max_trials) {
  EmuCtx* c = (EmuCtx*)h;
  c->mpc.upload_inputs(B, t0, x0, ref_t, ref_x, ev, modes);
  c->mpc.grid(B, horizon);
  c->mpc.sqp_iteration(B, max_trials);
  return c->mpc.ls_trials_run;
}
// raw buffer access for parity tests: name -> pointer
void* emu_buffer(void* h, const char* name) {
  QmMpcBuffers& d = ((EmuCtx*)h)->mpc.d;
#define F(n) if (!strcmp(name, #n)) return (void*)d.n;
  F(n_nodes) F(node_t) F(node_ts) F(node_dt) F(node_ev) F(node_mode) F(zvel) F(zpos) F(xref) F(eeref) F(status) F(x) F(u) F(dx) F(du) F(stage) F(lqdbg) F(perf) F(base_sum)
  F(perf_sum) F(step_info) F(alpha) F(done) F(xs) F(us) F(out_perf)
#undef F
  return nullptr;
}
int emu_sizes(int which) { int v[] = {SR_SIZE, LQ_DBG_SIZE, PF_SIZE, LQ_LDS_BYTES, RC_LDS_BYTES}; return v[which]; }
}
