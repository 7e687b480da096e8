// qmhip.hip — libqmhip.so: HIP backend of the launch sequence (qm_pipeline.h) + the C ABI of include/qmhip.h.
// gfx950 only.  One context = one device, one HIP stream, all buffers resident in HBM for max_batch instances.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "../../../include/qmhip.h"
#include "qm_model_io.h"
#define QM_LQ_KERNELS_EXTERN 1      /* K1b's instances live in qmhip_lq.hip */
#include "qm_pipeline.h"
#include "qm_wbc_pipeline.h"
#include "qm_sim_pipeline.h"
#include "qm_front_pipeline.h"
#include "qm_hoqp_pipeline.h"

static thread_local std::string g_create_error;      // per calling thread: qmhip_last_error(NULL) is the error of THIS thread's last failed create

// every entry point that takes a context serialises on it: calls from several threads are safe and run one after another (include/qmhip.h, "Threads")
#define QM_GUARD(c) std::unique_lock<std::recursive_mutex> qm_lk_; if (c) qm_lk_ = std::unique_lock<std::recursive_mutex>(const_cast<qmhip_ctx*>(c)->mu)
// entry points that touch horizon / front-end / plant buffers refuse a WBC-only context
#define QM_NEED_MPC(c) do { if ((c) && (c)->wbc_only) { const_cast<qmhip_ctx*>(c)->fail(std::string(__func__) + ": not available on a WBC-only context (qmhip_create_wbc_context)"); return QMHIP_ERR_STATE; } } while (0)
#define HIP_TRY(ctx, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { (ctx)->fail(std::string(#expr) + ": " + hipGetErrorString(e_)); return QMHIP_ERR_HIP; } } while (0)

struct HipBackend {
  hipStream_t stream = nullptr;            // MPC stream (K0..K5); host copies
  hipStream_t stream_b = nullptr;          // WBC stream: the WBC of step k runs beside the MPC kernels of step k + 1 (the reference runs them in two threads too)
  hipStream_t cur = nullptr;               // stream the next launch / memset goes to
  hipEvent_t ev_in = nullptr, ev_wbc = nullptr; bool wbc_pending = false;
  int profiling = 0;   // 0 off, 1 HIP-event span around every launch, 2 only around the modelled kernels (lq, riccati, wbc), 3 only around the LQ kernel (the dominant one: what the bench's timed region carries): two event records cost ≈ a launch
  std::string error;
  struct Span { std::string name; hipEvent_t a, b; };
  std::vector<Span> spans; std::vector<hipEvent_t> pool;
  std::map<std::string, std::pair<double, int>> acc;
  std::map<const void*, int> lds_set;
  std::map<std::string, int> lds_pad;      // profiling only (occupancy sweep, tools/occupancy_sweep.py): extra dynamic LDS per workgroup of a kernel group -> fewer resident workgroups per CU
  hipEvent_t ev() { if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; } hipEvent_t e = nullptr; check(hipEventCreate(&e), "hipEventCreate"); return e; }
  void check(hipError_t e, const char* what) { if (e != hipSuccess && error.empty()) error = std::string(what) + ": " + hipGetErrorString(e); }
  template <class K> const char* name_of(K k) {
    const void* p = (const void*)k;
    if (p == (const void*)qm_grid_kernel || p == (const void*)qm_grid_nodes_kernel || p == (const void*)qm_save_grid_kernel || p == (const void*)qm_advance_kernel) return "grid"; if (p == (const void*)qm_lq_kernel || p == (const void*)qm_lq_dbg_kernel || p == (const void*)qm_lq_ipm_kernel) return "lq"; if (p == (const void*)qm_lq_m18_kernel) return "lq_m18"; if (p == (const void*)qm_lq_kin_kernel) return "lq_kin"; if (p == (const void*)qm_riccati_kernel || p == (const void*)qm_riccati_prof_kernel) return "riccati";
    if (p == (const void*)qm_ls_eval_kernel || p == (const void*)qm_ls_eval_dense_kernel || p == (const void*)qm_ls_eval_ipm_kernel) return "ls_eval";
    if (p == (const void*)qm_ipm_init_kernel || p == (const void*)qm_ipm_dir_kernel || p == (const void*)qm_ipm_alpha_kernel || p == (const void*)qm_ipm_commit_kernel || p == (const void*)qm_ipm_barrier_kernel) return "ipm"; if (p == (const void*)qm_ilqr_rollout_kernel) return "rollout"; if (p == (const void*)qm_sim_kernel) return "sim"; if (p == (const void*)qm_wbc_kernel || p == (const void*)qm_wbc_prof_kernel) return "wbc"; if (p == (const void*)qm_policy_kernel || p == (const void*)qm_policy_measured_kernel) return "policy"; if (p == (const void*)qm_hoqp_kernel) return "hoqp";
    return "ls_misc";
  }
  template <class K, class A> void launch(K kernel, int grid, int block, size_t lds, const A& args) {
    if (grid <= 0) return;
    if (!lds_pad.empty()) { auto it = lds_pad.find(name_of(kernel)); if (it != lds_pad.end()) lds += (size_t)it->second; }
    if (lds > 48 * 1024) {
      const void* p = (const void*)kernel; auto it = lds_set.find(p);
      if (it == lds_set.end() || it->second < (int)lds) { check(hipFuncSetAttribute(p, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute"); lds_set[p] = (int)lds; }
    }
    const void* kp_ = (const void*)kernel;
    const bool is_lq_ = kp_ == (const void*)qm_lq_kernel || kp_ == (const void*)qm_lq_m18_kernel || kp_ == (const void*)qm_lq_ipm_kernel;
    const bool span = profiling == 1 || (profiling == 3 && is_lq_) || (profiling == 2 && (kp_ == (const void*)qm_lq_kernel || kp_ == (const void*)qm_lq_m18_kernel || kp_ == (const void*)qm_lq_ipm_kernel || kp_ == (const void*)qm_riccati_kernel || kp_ == (const void*)qm_wbc_kernel));      // (lq_m18: launched only on workloads with a phase of three or four stance feet; bench.py prices lq + lq_m18 together)
    Span s; if (span) { s.name = name_of(kernel); s.a = ev(); s.b = ev(); check(hipEventRecord(s.a, cur), "hipEventRecord"); }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds, cur, args);
    check(hipGetLastError(), "kernel launch");
    if (span) { check(hipEventRecord(s.b, cur), "hipEventRecord"); spans.push_back(s); }
  }
  void resolve() {
    if (spans.empty()) return;
    hipStreamSynchronize(stream); hipStreamSynchronize(stream_b);
    for (auto& s : spans) { float ms = 0; hipEventElapsedTime(&ms, s.a, s.b); auto& a = acc[s.name]; a.first += ms; a.second += 1; pool.push_back(s.a); pool.push_back(s.b); }
    spans.clear();
  }
  void* alloc(size_t n) { void* p = nullptr; check(hipMalloc(&p, n ? n : 8), "hipMalloc"); return p; }
  void free(void* p) { hipFree(p); }
  void zero(void* p, size_t n) { check(hipMemsetAsync(p, 0, n, cur), "hipMemsetAsync"); }
  // host copies see the results of BOTH streams and never run beside device work of either (they are not on the hot path)
  void to_device(void* d, const void* s, size_t n) { sync(); check(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, stream), "H2D"); check(hipStreamSynchronize(stream), "sync"); }
  void to_host(void* d, const void* s, size_t n) { sync(); check(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, stream), "D2H"); check(hipStreamSynchronize(stream), "sync"); }
  void sync() { check(hipStreamSynchronize(stream), "sync"); check(hipStreamSynchronize(stream_b), "sync"); wbc_pending = false; }
  // host-visible (pinned, mapped) memory for flags a kernel publishes: returns the device-side address, *host_view the host-side one
  hipEvent_t ev_order = nullptr;
  void stream_select(int s) { cur = s ? stream_b : stream; }
  void stream_order(int a, int b) { if (!ev_order) check(hipEventCreateWithFlags(&ev_order, hipEventDisableTiming), "hipEventCreate"); check(hipEventRecord(ev_order, a ? stream_b : stream), "hipEventRecord"); check(hipStreamWaitEvent(b ? stream_b : stream, ev_order, 0), "hipStreamWaitEvent"); }
  void copy_dd(void* d, const void* s, size_t n) { check(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, cur), "D2D"); }
  void* alloc_mapped(size_t n, void** host_view) { void* h = nullptr; void* dv = nullptr; check(hipHostMalloc(&h, n ? n : 8, hipHostMallocMapped), "hipHostMalloc"); check(hipHostGetDevicePointer(&dv, h, 0), "hipHostGetDevicePointer"); *host_view = h; return dv; }
  void free_mapped(void* host_view) { hipHostFree(host_view); }
  void wait_launched() { check(hipStreamSynchronize(cur), "sync"); }     // everything launched so far on the current stream has completed
  // wait for a host-visible word a kernel already launched on `cur` overwrites (anything but `pending`): a BOUNDED spin — a stream synchronisation costs 10-30 us of
  // wake-up latency, which matters for a line-search trial of a few tens of microseconds, but a solve must not hold a host core beside the ros_control thread for its
  // whole length — of at most spin_us microseconds (the trial kernels of a 1024-batch take ≈ 0.25 ms), then the stream synchronisation
  int spin_us = 400;
  void wait_flag(volatile int* word, int pending) {
    if (*word != pending) return;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      for (int k = 0; k < 256; ++k) if (*word != pending) return;
      if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > spin_us) break;
    }
    check(hipStreamSynchronize(stream), "sync"); if (cur != stream) check(hipStreamSynchronize(cur), "sync");
    if (*word == pending && error.empty()) error = "a kernel did not publish its host-visible word";
  }
  // WBC of the current step on stream_b: its inputs were produced on `stream` (ev_in); the next producers on `stream` wait for ev_wbc
  void wbc_inputs_next() { if (wbc_pending) { check(hipStreamWaitEvent(stream, ev_wbc, 0), "hipStreamWaitEvent"); wbc_pending = false; } }
  void wbc_begin() { check(hipEventRecord(ev_in, stream), "hipEventRecord"); check(hipStreamWaitEvent(stream_b, ev_in, 0), "hipStreamWaitEvent"); cur = stream_b; }
  void wbc_end() { check(hipEventRecord(ev_wbc, stream_b), "hipEventRecord"); cur = stream; wbc_pending = true; }
};

struct qmhip_ctx {
  int device = 0, max_batch = 0, max_nodes = 0, max_ref = 0, max_ev = 0;
  double mb[MB_SIZE], st[ST_SIZE];
  HipBackend bk; QmMpcPipeline<HipBackend> mpc; QmWbcPipeline<HipBackend> wbc; QmFrontPipeline<HipBackend> front; QmSimPipeline<HipBackend> sim; QmHoqpPipeline<HipBackend> hoqp;
  std::recursive_mutex mu;      // serialises the entry points of this context
  bool wbc_only = false;        // created by qmhip_create_wbc_context: carries the model + the WBC buffers, no horizon buffers
  void* dl_dev = nullptr; void* dl_pin = nullptr; size_t dl_cap = 0;      // staging of qmhip_mpc_download (device transpose buffer + its pinned host mirror), allocated on first use
  char* tick_pin = nullptr;     // pinned host staging of the control-tick path (qmhip_wbc_step): [inputs of max_batch instances | outputs]
  std::string error; int lastB = 0; bool have_solution = false; int front_B = 0; long sim_ticks = 0;
  int filler_at_lq = 0; int filler_live = 0, filler_lds = 20 * 1024; double* filler_in = nullptr;      // profiling only: footprint of the co-residency stand-in (qmhip_debug_set "filler_live" / "filler_lds")
  int wbc_defer = 0; int wbc_deferred_B = 0; double wbc_deferred_period = 0.0;      // scheduling experiment (qmhip_debug_set "wbc_defer"): the WBC of step k is launched behind K1a of step k + 1 (flush_wbc at the latest)
  double* filler_out = nullptr; int filler_cap = 0;      // output of the profiling-only filler kernel (co-residency probe)
  bool filler_buffer(int waves) { if (filler_cap >= waves) return true; if (filler_out) hipFree(filler_out); filler_out = nullptr; filler_cap = 0;
                                  if (hipMalloc(&filler_out, (size_t)waves * 64 * 8) != hipSuccess) return false; filler_cap = waves; return true; }
  qmhip_ctx() : mpc(bk), wbc(bk), front(bk), sim(bk), hoqp(bk) {}
  void fail(const std::string& m) { error = m; }
  // getModeSchedule on the device GaitSchedule -> the solver's schedule buffers; from here on its sticky status speaks for the schedule of this batch (until the host supplies one)
  void gait_schedule(int B, double horizon) { front.gait_schedule(mpc.d, B, horizon); mpc.front_status = front.f.gs_status; mpc.front_B = B; }
  // sqp.sqpIteration (task.info:79, shipped 1): SQP iterations per MPC call [upstream SqpSolver::runImpl loop]; every instance of the batch runs all of
  // them (an instance whose line search finds no step just keeps its iterate)
  int sqp_iterations() const { const int n = (int)qm_ms_param(st, ST_SQP_ITER); return n < 1 ? 1 : (n > 50 ? 50 : n); }      // ipm.ipmIteration with solver 2
  bool fused_policy = true;      // the policy at t0 comes from the line search's deciding kernels and the batch's apply runs behind the WBC launch (qm_pipeline.h); false: qm_policy_kernel behind the apply (rounds 1-5; qmhip_debug_set "fused_policy")
  // One control step on resident data, everything enqueued, nothing waited for:
  //   WBC stream: the synthetic measured state of this step (a function of x0 only) — early, beside the MPC kernels
  //   MPC stream: K0 .. K4; the kernels that DECIDE the step length also write the policy at t0 (what evaluatePolicy(t0) reads from x + alpha dx)
  //   WBC stream: waits for that decision, then the WBC;   MPC stream: the batch's apply (the primal solution on every node) runs beside it, then the next step
  // Rounds 1-5 had apply -> policy kernel -> WBC on the critical path: 2 launches and ~ 60 us of a 2.7 ms step.
  void control_step(int B, double horizon, double period, double time, bool warm) {
    bk.stream_order(0, 1); bk.cur = bk.stream_b; wbc.measured_from_x0(mpc.d, B, time); bk.cur = bk.stream;      // (x0 is final on the MPC stream at this point; the previous WBC on this stream has read its rbd)
    mpc.p0_x = wbc.w.x_des; mpc.p0_u = wbc.w.u_des; mpc.p0_mode = wbc.w.mode;
    mpc.grid(B, horizon, warm);
    for (int it = 0, ni = sqp_iterations(); it < ni; ++it) { const bool last_it = it + 1 == ni; mpc.p0_enable = fused_policy && last_it; mpc.defer_apply = fused_policy && last_it; mpc.sqp_iteration(B, 14, last_it); }
    mpc.p0_enable = false; mpc.defer_apply = false; lastB = B; have_solution = true;
    if (!mpc.p0_done) { mpc.apply_pending(); bk.wbc_inputs_next(); wbc.policy_eval_at_t0(mpc.d, B); }      // iLQR / interior-point solver / host-driven line search: the policy kernel on the applied primal solution
    if (wbc_defer) { mpc.apply_pending(); wbc_deferred_B = B; wbc_deferred_period = period; qmhip_ctx* cc = this; mpc.before_lq = [cc]() { cc->flush_wbc(); }; return; }      // (scheduling experiment)
    bk.wbc_begin(); wbc.step(mpc.d, B, period, 0); bk.wbc_end();
    mpc.apply_pending();
  }
  // launch a deferred WBC now, ordered behind everything enqueued on the MPC stream so far
  void flush_wbc() { if (!wbc_deferred_B) return; const int B = wbc_deferred_B; wbc_deferred_B = 0; bk.wbc_begin(); wbc.step(mpc.d, B, wbc_deferred_period, 0); bk.wbc_end(); }
  int hipstate() { if (!bk.error.empty()) { error = bk.error; bk.error.clear(); return QMHIP_ERR_HIP; } return QMHIP_OK; }
};

// ---- FP64 micro-benchmarks (roofline denominators) ----
__global__ void qm_bench_fma_kernel(double* out, int iters) {
  double a0 = threadIdx.x * 1e-9, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; const double m = 1.0000001, c = 1e-9;
  for (int i = 0; i < iters; ++i) { a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c); a4 = fma(a4, m, c); a5 = fma(a5, m, c); a6 = fma(a6, m, c); a7 = fma(a7, m, c); }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ void qm_bench_mfma_kernel(double* out, int iters) {
  qm_d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0; const double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

// settings whose value the kernels' loop bounds depend on (K0 walks t0 + k dt up to the horizon)
static bool setting_ok(int idx, double v) {
  if (idx == ST_SQP_DT || idx == ST_IPM_DT) return v > 0.0 && std::isfinite(v);
  if (idx == ST_GRID_DT_MIN) return v >= 0.0 && std::isfinite(v);
  if (idx == ST_RICCATI_STRICT || idx == ST_IPM_PRIMAL_FOR_DUAL) return v == 0.0 || v == 1.0;
  // interior-point solver (slot 3, k_ipm.h): the barrier parameter and the slack / dual floors are divided by and go through log(); the margin and the linear factor are fractions
  if (idx == ST_IPM_MU || idx == ST_IPM_MU_TARGET || idx == ST_IPM_SLACK_LB || idx == ST_IPM_DUAL_LB) return v > 0.0 && std::isfinite(v);
  if (idx == ST_IPM_FTB_MARGIN || idx == ST_IPM_MU_LINEAR) return v > 0.0 && v < 1.0;
  if (idx == ST_IPM_MU_POWER) return v > 1.0 && std::isfinite(v);
  if (idx == ST_IPM_SLACK_MARGIN || idx == ST_IPM_DUAL_MARGIN) return v >= 0.0 && std::isfinite(v);
  return true;
}
// the slots the interior-point solver reads: all of them must hold before solver 3 may run (a blob of the 1048-double layout has zeros there: division by zero in qm_ipm_dir_kernel, log(0) in the merit)
static const int kIpmSlots[] = {ST_IPM_DT, ST_IPM_MU, ST_IPM_MU_TARGET, ST_IPM_MU_LINEAR, ST_IPM_MU_POWER, ST_IPM_FTB_MARGIN, ST_IPM_PRIMAL_FOR_DUAL, ST_IPM_SLACK_LB, ST_IPM_DUAL_LB, ST_IPM_SLACK_MARGIN, ST_IPM_DUAL_MARGIN};
static bool ipm_settings_ok(const double* st) { for (int i : kIpmSlots) if (!setting_ok(i, st[i])) return false; return true; }

// ---- co-residency probe (profiling only): a latency-bound stand-in for a narrow (<= 256 VGPR, <= 20 KB LDS) one-wave-per-instance solver wave — chains of
// dependent f64 MFMAs and FMAs with an LDS round trip per step, ≈ 40 % issue utilisation like qm_riccati_kernel — launched on the second stream beside the
// MPC kernels to measure what sharing SIMDs with the issue-bound LQ kernel is worth before restructuring the real kernel (tools/coresidency_probe.py)
__global__ void __launch_bounds__(64, 2) qm_filler_kernel(double* out, int iters) {
  extern __shared__ double qm_smem[];
  const int l = threadIdx.x & 63;
  qm_d4 acc = {0.0, 0.0, 0.0, 0.0}; double a = 1.0 + l * 1e-9, b = 1e-9, f = 1.0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 6; ++k) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0); a = acc[0] * 1e-30 + 1.0; }      // dependent matrix-core chain
#pragma unroll
    for (int k = 0; k < 12; ++k) f = fma(f, 1.0000001, acc[k & 3] * 1e-30);                                                     // dependent vector chain
    qm_smem[(l * 7 + i) & 2047] = f; __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    f += qm_smem[(l * 13 + i + 1) & 2047];                                                                                     // LDS round trip
  }
  out[blockIdx.x * 64 + l] = f + acc[1];
}

// the same stand-in with a WIDE register footprint (round 6): NLIVE doubles are loaded up front and consumed at the end, so 2 NLIVE vector registers stay allocated for the wave's whole
// life — NLIVE 153 (386 registers) / 40 KB of LDS is the footprint of today's qm_wbc_kernel, NLIVE 144 (339 registers) / 24 KB the footprint the round-5 review asks of it (344 registers:
// one 168-register LQ wave fits beside it on a SIMD, four 15.6 KB LQ workgroups beside four of these on a CU).  What the second footprint is worth is measured BEFORE rewriting the WBC.
template <int NLIVE> __device__ __forceinline__ void qm_filler_wide_body(double* out, const double* in, int iters) {
  extern __shared__ double qm_smem[];
  const int l = threadIdx.x & 63;
  double live[NLIVE];
#pragma unroll
  for (int k = 0; k < NLIVE; ++k) live[k] = in[(size_t)k * 64 + l];
  qm_d4 acc = {0.0, 0.0, 0.0, 0.0}; double a = 1.0 + l * 1e-9, b = 1e-9, f = 1.0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 6; ++k) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0); a = acc[0] * 1e-30 + 1.0; }
#pragma unroll
    for (int k = 0; k < 12; ++k) f = fma(f, 1.0000001, acc[k & 3] * 1e-30);
    qm_smem[(l * 7 + i) & 2047] = f; __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    f += qm_smem[(l * 13 + i + 1) & 2047];
#pragma unroll
    for (int k = 0; k < NLIVE; ++k) asm volatile("" : "+v"(live[k]));      // every value stays in a register through the loop (no rematerialisation, no sinking of the loads)
  }
  double s = f + acc[1];
#pragma unroll
  for (int k = 0; k < NLIVE; ++k) s += live[k];
  out[blockIdx.x * 64 + l] = s;
}

__global__ void __launch_bounds__(64) qm_filler_wide392_kernel(double* out, const double* in, int iters) { qm_filler_wide_body<153>(out, in, iters); }      // 386 registers (the count is the compiler's: found by trial)
__global__ void __launch_bounds__(64) qm_filler_wide344_kernel(double* out, const double* in, int iters) { qm_filler_wide_body<144>(out, in, iters); }      // 339 registers

static int create_common(const double* mb, const double* st, int device, int max_batch, int max_nodes, int max_ref, int max_ev, qmhip_ctx** out, bool wbc_only = false) {
  if (!out || max_batch <= 0 || max_nodes < 3 || max_nodes > RW_MAXNODES || max_ref < 1 || max_ev < 1) { g_create_error = "qmhip_create: bad argument (max_nodes must be in [3, 512])"; return QMHIP_ERR_ARG; }
  std::string err; if (!qmio::validateModelBlob(mb, err)) { g_create_error = err; return QMHIP_ERR_MODEL; }
  if (!setting_ok(ST_SQP_DT, st[ST_SQP_DT])) { g_create_error = "settings blob: sqp.dt must be a positive finite number"; return QMHIP_ERR_MODEL; }
  // a blob of an older layout (no size / version stamp travels with it) would put garbage into the slots added since: check the ones a kernel's control flow depends on
  if (!setting_ok(ST_GRID_DT_MIN, st[ST_GRID_DT_MIN])) { g_create_error = "settings blob: the time grid's minimum step (ST_GRID_DT_MIN) must be a non-negative finite number - is the blob of an older layout (ST_SIZE)?"; return QMHIP_ERR_MODEL; }
  if (st[ST_RICCATI_STRICT] != 0.0 && st[ST_RICCATI_STRICT] != 1.0) { g_create_error = "settings blob: ST_RICCATI_STRICT must be 0 or 1 - is the blob of an older layout (ST_SIZE)?"; return QMHIP_ERR_MODEL; }
  if (st[ST_SOLVER] != 0.0 && st[ST_SOLVER] != 1.0 && st[ST_SOLVER] != 2.0 && st[ST_SOLVER] != 3.0) { g_create_error = "settings blob: ST_SOLVER must be 0, 1, 2 or 3"; return QMHIP_ERR_MODEL; }
  if (st[ST_SOLVER] == 3.0 && !ipm_settings_ok(st)) { g_create_error = "settings blob: ST_SOLVER = 3 needs the ipm block's slots (ST_IPM_*: barrier parameters, slack / dual floors > 0, margin and linear factor in (0, 1), power > 1) - is the blob of an older layout (ST_SIZE)?"; return QMHIP_ERR_MODEL; }
  int ndev = 0; if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_create_error = "no HIP device available (libqmhip has no CPU fallback)"; return QMHIP_ERR_HIP; }
  if (device < 0 || device >= ndev) { g_create_error = "device index out of range"; return QMHIP_ERR_ARG; }
  if (hipSetDevice(device) != hipSuccess) { g_create_error = "hipSetDevice failed"; return QMHIP_ERR_HIP; }
  qmhip_ctx* c = new qmhip_ctx(); c->device = device; c->max_batch = max_batch; c->max_nodes = max_nodes; c->max_ref = max_ref; c->max_ev = max_ev;
  memcpy(c->mb, mb, sizeof(c->mb)); memcpy(c->st, st, sizeof(c->st));
  // (profiling only: QM_WBC_STREAM_PRIORITY = -1 / 0 / 1 creates the WBC stream with the highest / default / lowest priority the device offers; default: no priority given)
  const char* pr_env = getenv("QM_WBC_STREAM_PRIORITY"); bool sb_ok = true;
  if (pr_env) { int lo = 0, hi = 0; hipDeviceGetStreamPriorityRange(&lo, &hi); const int want = atoi(pr_env); sb_ok = hipStreamCreateWithPriority(&c->bk.stream_b, hipStreamDefault, want < 0 ? hi : (want > 0 ? lo : 0)) == hipSuccess; }
  if (!pr_env) { int lo = 0, hi = 0; hipDeviceGetStreamPriorityRange(&lo, &hi); sb_ok = hipStreamCreateWithPriority(&c->bk.stream_b, hipStreamDefault, hi) == hipSuccess; }      // the WBC / control-tick stream at the highest priority: its waves get the SIMDs first (- 0.4 % of the pipelined step, profiles/r06_ab_wbc_schedule.log)
  if (hipStreamCreate(&c->bk.stream) != hipSuccess || !sb_ok) { g_create_error = "hipStreamCreate failed"; delete c; return QMHIP_ERR_HIP; }
  c->bk.cur = c->bk.stream; hipEventCreateWithFlags(&c->bk.ev_in, hipEventDisableTiming); hipEventCreateWithFlags(&c->bk.ev_wbc, hipEventDisableTiming);
  c->wbc_only = wbc_only;
  c->mpc.allocate(c->mb, c->st, wbc_only ? 1 : max_batch, max_nodes, max_ref, max_ev, false); c->mpc.solver = (int)st[ST_SOLVER];
  c->wbc.allocate(max_batch);
  if (!wbc_only) c->front.allocate(max_batch);
  { void* h = nullptr; c->bk.check(hipHostMalloc(&h, QmWbcPipeline<HipBackend>::in_bytes(max_batch) + QmWbcPipeline<HipBackend>::out_bytes(max_batch), hipHostMallocDefault), "hipHostMalloc"); c->tick_pin = (char*)h; }
  c->bk.sync();
  if (!c->bk.error.empty()) { g_create_error = c->bk.error; c->mpc.release(); c->wbc.release(); c->front.release(); if (c->tick_pin) hipHostFree(c->tick_pin); hipStreamDestroy(c->bk.stream); hipStreamDestroy(c->bk.stream_b); delete c; return QMHIP_ERR_HIP; }
  *out = c; return QMHIP_OK;
}

extern "C" {

int qmhip_parse_model(const char* urdf, const char* task, const char* ref, double* mb, double* st) {
  if (!urdf || !task || !ref || !mb || !st) { g_create_error = "qmhip_parse_model: null argument"; return QMHIP_ERR_ARG; }
  // same checks (and order) as QMInterface's constructor, QMInterface.cpp:40-62
  if (!qmio::fileExists(task)) { g_create_error = std::string("[QMInterface] Task file not found: ") + task; return QMHIP_ERR_FILE; }
  if (!qmio::fileExists(urdf)) { g_create_error = std::string("[QMInterface] URDF file not found: ") + urdf; return QMHIP_ERR_FILE; }
  if (!qmio::fileExists(ref)) { g_create_error = std::string("[QMInterface] targetCommand file not found: ") + ref; return QMHIP_ERR_FILE; }
  std::string ee, err;
  if (!qmio::loadEeFrameName(task, ee, err) || !qmio::buildModelBlob(urdf, ref, ee, mb, nullptr, err) || !qmio::buildSettingsBlob(task, mb, st, err)) { g_create_error = err; return QMHIP_ERR_MODEL; }
  return QMHIP_OK;
}
int qmhip_create(const char* urdf, const char* task, const char* ref, int device, int max_batch, int max_nodes, int max_ref, int max_ev, qmhip_ctx** out) {
  std::vector<double> mb(MB_SIZE), st(ST_SIZE);
  const int rc = qmhip_parse_model(urdf, task, ref, mb.data(), st.data()); if (rc != QMHIP_OK) return rc;
  return create_common(mb.data(), st.data(), device, max_batch, max_nodes, max_ref, max_ev, out);
}
int qmhip_create_from_blobs(const double* mb, const double* st, int device, int max_batch, int max_nodes, int max_ref, int max_ev, qmhip_ctx** out) {
  if (!mb || !st) { g_create_error = "qmhip_create_from_blobs: null blob"; return QMHIP_ERR_ARG; }
  return create_common(mb, st, device, max_batch, max_nodes, max_ref, max_ev, out);
}
int qmhip_create_wbc_context(const qmhip_ctx* c, int max_batch, qmhip_ctx** out) { QM_GUARD(c);
  if (!c || !out || max_batch <= 0) { g_create_error = "qmhip_create_wbc_context: bad argument"; return QMHIP_ERR_ARG; }
  return create_common(c->mb, c->st, c->device, max_batch, 3, 1, 1, out, true);      // same model / settings values, own device copies, own streams: nothing mutable is shared
}
void qmhip_destroy(qmhip_ctx* c) {
  if (!c) return; hipSetDevice(c->device); c->bk.sync(); if (c->filler_out) hipFree(c->filler_out); if (c->filler_in) hipFree(c->filler_in); if (c->dl_dev) hipFree(c->dl_dev); if (c->dl_pin) hipHostFree(c->dl_pin); if (c->tick_pin) hipHostFree(c->tick_pin); c->mpc.release(); c->wbc.release(); c->front.release(); c->sim.release(); c->hoqp.release();
  for (auto e : c->bk.pool) hipEventDestroy(e); if (c->bk.ev_order) hipEventDestroy(c->bk.ev_order); hipEventDestroy(c->bk.ev_in); hipEventDestroy(c->bk.ev_wbc); hipStreamDestroy(c->bk.stream); hipStreamDestroy(c->bk.stream_b); delete c;
}
// the text is copied under the context lock into a per-thread buffer: the pointer stays valid (until this THREAD's next qmhip_last_error) even if another thread's
// failing call on the same context replaces the context's message meanwhile
const char* qmhip_last_error(const qmhip_ctx* c) { static thread_local std::string copy; QM_GUARD(c); if (!c) return g_create_error.c_str(); copy = c->error; return copy.c_str(); }
int qmhip_export_blobs(const qmhip_ctx* c, double* mb, double* st) { QM_GUARD(c); if (!c) return QMHIP_ERR_ARG; if (mb) memcpy(mb, c->mb, sizeof(c->mb)); if (st) memcpy(st, c->st, sizeof(c->st)); return QMHIP_OK; }
// name of a field of qm_wbc::WbcWeightConfig (cfg/wbcWigeht.cfg:7-47) -> settings slot of the gain WbcBase::dynamicCallback stores it in (WbcBase.cpp:69-116)
int qmhip_wbc_gain_index(const char* name) {
  if (!name) return -1;
  static const struct { const char* name; int idx; } scalars[] = {{"kp_swing", ST_KP_SWING}, {"kd_swing", ST_KD_SWING}, {"baseHeightKp", ST_KP_BASE_H}, {"baseHeightKd", ST_KD_BASE_H},
    {"kp_base_linear", ST_KP_BASE_LIN}, {"kd_base_linear", ST_KD_BASE_LIN}, {"kp_base_angular", ST_KP_BASE_ANG}, {"kd_base_angular", ST_KD_BASE_ANG}};
  for (const auto& e : scalars) if (!strcmp(name, e.name)) return e.idx;
  static const struct { const char* prefix; int base; } joints[] = {{"kp_arm_joint_", ST_KP_ARM_J}, {"kd_arm_joint_", ST_KD_ARM_J}};
  for (const auto& e : joints) { const size_t n = strlen(e.prefix); if (!strncmp(name, e.prefix, n) && name[n] >= '1' && name[n] <= '6' && name[n + 1] == 0) return e.base + (name[n] - '1'); }
  static const struct { const char* prefix; int base; } axes[] = {{"kp_ee_linear_", ST_KP_EE_LIN}, {"kd_ee_linear_", ST_KD_EE_LIN}, {"kp_ee_angular_", ST_KP_EE_ANG}, {"kd_ee_angular_", ST_KD_EE_ANG}};
  for (const auto& e : axes) { const size_t n = strlen(e.prefix); if (!strncmp(name, e.prefix, n) && name[n] >= 'x' && name[n] <= 'z' && name[n + 1] == 0) return e.base + (name[n] - 'x'); }
  return -1;
}
int qmhip_set_setting(qmhip_ctx* c, int idx, double v) { QM_GUARD(c);
  if (!c || idx < 0 || idx >= ST_SIZE) return QMHIP_ERR_ARG;
  if (!setting_ok(idx, v)) { c->fail("qmhip_set_setting: sqp.dt / ipm.dt must be a positive finite number, the grid's minimum step a non-negative one, ST_RICCATI_STRICT 0 or 1; ipm block: barrier parameters and slack / dual floors > 0, margin and linear decrease factor in (0, 1), superlinear power > 1, margin rates >= 0"); return QMHIP_ERR_ARG; }
  if (idx == ST_SOLVER && v == 3.0 && !ipm_settings_ok(c->st)) { c->fail("qmhip_set_setting: solver 3 needs valid ipm settings (ST_IPM_*) - is the settings blob of an older layout?"); return QMHIP_ERR_ARG; }
  if (idx == ST_SOLVER && v >= 2.0 && !setting_ok(ST_IPM_DT, c->st[ST_IPM_DT])) { c->fail("qmhip_set_setting: solvers 2 / 3 need a positive finite ipm.dt"); return QMHIP_ERR_ARG; }
  if (idx == ST_SOLVER) { if (v != 0.0 && v != 1.0 && v != 2.0 && v != 3.0) { c->fail("qmhip_set_setting: ST_SOLVER is 0 (SQP), 1 (discrete iLQR), 2 (the SQP step on the `ipm` block's parameters) or 3 (interior-point method with hard friction cones / arm boxes)"); return QMHIP_ERR_ARG; } c->mpc.solver = (int)v; c->mpc.solved_B = 0; c->have_solution = false; c->mpc.ipm_fresh = true; }      // (a solver switch starts the interior-point state over: slack / dual / barrier parameter of an earlier solve are not carried across)
  hipSetDevice(c->device); c->st[idx] = v; c->bk.to_device(c->mpc.d.st + idx, &v, 8); c->mpc.note_settings(c->st); return c->hipstate();
}

int qmhip_mpc_upload(qmhip_ctx* c, int B, const double* t0, const double* x0, int n_ref, const double* ref_t, const double* ref_x, int n_ev, const double* ev, const int32_t* modes) { QM_GUARD(c); QM_NEED_MPC(c);
  if (!c) return QMHIP_ERR_ARG;
  if (B <= 0 || B > c->max_batch || n_ref != c->max_ref || n_ev != c->max_ev || !t0 || !x0 || !ref_t || !ref_x || !ev || !modes) { c->fail("qmhip_mpc_upload: bad argument (B <= max_batch, n_ref == max_ref_knots, n_events == max_events required)"); return QMHIP_ERR_ARG; }
  hipSetDevice(c->device); c->mpc.upload_inputs(B, t0, x0, ref_t, ref_x, ev, modes); c->mpc.solved_B = 0; c->lastB = B; c->have_solution = false;
  c->mpc.front_status = nullptr; c->mpc.front_B = 0;      // the schedule now comes from the host: a failed device GaitSchedule update no longer speaks for it (re-attached by the next qmhip_gait_update_resident)
 return c->hipstate();
}
int qmhip_mpc_solve_resident(qmhip_ctx* c, int B, double horizon) { QM_GUARD(c); QM_NEED_MPC(c);
  if (!c || B <= 0 || B > c->max_batch || !(horizon > 0)) { if (c) c->fail("qmhip_mpc_solve_resident: bad argument"); return QMHIP_ERR_ARG; }
  hipSetDevice(c->device); c->mpc.grid(B, horizon); for (int it = 0, ni = c->sqp_iterations(); it < ni; ++it) c->mpc.sqp_iteration(B, 14, it + 1 == ni); c->lastB = B; c->have_solution = true; return c->hipstate();
}
int qmhip_mpc_set_initial(qmhip_ctx* c, int B, const double* t0, const double* x0) { QM_GUARD(c); QM_NEED_MPC(c);
  if (!c || B <= 0 || B > c->max_batch || !t0 || !x0) { if (c) c->fail("qmhip_mpc_set_initial: bad argument"); return QMHIP_ERR_ARG; }
  hipSetDevice(c->device); c->bk.to_device(c->mpc.d.t0, t0, (size_t)B * 8); c->bk.to_device(c->mpc.d.x0, x0, (size_t)B * 30 * 8); return c->hipstate();
}
int qmhip_mpc_update_references(qmhip_ctx* c, int B, int n_ref, const double* ref_t, const double* ref_x, int n_ev, const double* ev, const int32_t* modes) { QM_GUARD(c); QM_NEED_MPC(c);
  if (!c) return QMHIP_ERR_ARG;
  if (B <= 0 || B != c->lastB || (!ref_t != !ref_x) || (!ev != !modes) || (ref_t && n_ref != c->max_ref) || (ev && n_ev != c->max_ev)) {
    c->fail("qmhip_mpc_update_references: bad argument (B == batch of the last upload, n_ref == max_ref_knots, n_events == max_events required; arrays come in pairs)"); return QMHIP_ERR_ARG; }
  hipSetDevice(c->device);
  if (ref_t) { c->bk.to_device(c->mpc.d.ref_t, ref_t, (size_t)B * n_ref * 8); c->bk.to_device(c->mpc.d.ref_x, ref_x, (size_t)B * n_ref * QM_NREF * 8); }
  if (ev) { c->bk.to_device(c->mpc.d.ev, ev, (size_t)B * n_ev * 8); c->bk.to_device(c->mpc.d.modes, modes, (size_t)B * (n_ev + 1) * 4); c->mpc.front_status = nullptr; c->mpc.front_B = 0; }
  return c->hipstate();
}
int qmhip_mpc_solve_resident_warm(qmhip_ctx* c, int B, double horizon) { QM_GUARD(c); QM_NEED_MPC(c);
  if (!c || B <= 0 || B > c->max_batch || !(horizon > 0)) { if (c) c->fail("qmhip_mpc_solve_resident_warm: bad argument"); return QMHIP_ERR_ARG; }
  hipSetDevice(c->device); c->mpc.grid(B, horizon, true); for (int it = 0, ni = c->sqp_iterations(); it < ni; ++it) c->mpc.sqp_iteration(B, 14, it + 1 == ni); c->lastB = B; c->have_solution = true; return c->hipstate();
}
int qmhip_mpc_advance_resident(qmhip_ctx* c, int B, double dt) { QM_GUARD(c); QM_NEED_MPC(c);
  if (!c || B <= 0 || B > c->max_batch) { if (c) c->fail("qmhip_mpc_advance_resident: bad argument"); return QMHIP_ERR_ARG; }
  if (!c->have_solution || c->mpc.solved_B != B) { c->fail("qmhip_mpc_advance_resident: no solution of this batch to advance along"); return QMHIP_ERR_STATE; }
  hipSetDevice(c->device); c->mpc.advance(B, dt); return c->hipstate();
}
int qmhip_closed_loop_resident(qmhip_ctx* c, int B, int n_steps, double mpc_dt, double horizon, double period, double time0) { QM_GUARD(c); QM_NEED_MPC(c);
  if (!c || B <= 0 || B > c->max_batch || n_steps <= 0 || !(horizon > 0) || !(period > 0)) { if (c) c->fail("qmhip_closed_loop_resident: bad argument"); return QMHIP_ERR_ARG; }
  hipSetDevice(c->device);
  for (int k = 0; k < n_steps; ++k) {
    if (k > 0) c->mpc.advance(B, mpc_dt);
    if (c->front_B == B) c->gait_schedule(B, horizon);     // device-resident GaitSchedule active: modifyReferences before every MPC call
    c->control_step(B, horizon, period, time0 + k * mpc_dt, true);
  }
  return c->hipstate();
}
// ---- reference / gait front-end ----
int qmhip_gait_set_templates(qmhip_ctx* c, int n_gaits, const int32_t* n_phases, const double* switching_times, const int32_t* mode_sequence) { QM_GUARD(c); QM_NEED_MPC(c);
  if (!c || n_gaits <= 0 || !n_phases || !switching_times || !mode_sequence) { if (c) c->fail("qmhip_gait_set_templates: bad argument"); return QMHIP_ERR_ARG; }
  for (int g = 0; g < n_gaits; ++g) if (n_phases[g] < 0 || n_phases[g] > QM_GAIT_MAX_PHASES) { c->fail("qmhip_gait_set_templates: a template has more than QMHIP_GAIT_MAX_PHASES phases"); return QMHIP_ERR_ARG; }
  hipSetDevice(c->device); c->front.set_templates(n_gaits, n_phases, switching_times, mode_sequence); return c->hipstate();
}
int qmhip_gait_reset(qmhip_ctx* c, int B, int n_events, const double* event_times, const int32_t* mode_sequence, int default_template) { QM_GUARD(c); QM_NEED_MPC(c);
  if (!c || B <= 0 || B > c->max_batch || n_events < 1 || n_events > QM_GAIT_EVENT_SLOTS || !event_times || !mode_sequence || default_template < 0 || default_template >= c->front.f.n_gaits) {
    if (c) c->fail("qmhip_gait_reset: bad argument (templates must be set first; the initial schedule needs at least one event)"); return QMHIP_ERR_ARG; }
  hipSetDevice(c->device); c->front.phase_transition_stance_time = c->st[ST_PHASE_TRANS_STANCE];
  c->front.gait_reset(B, n_events, event_times, mode_sequence, default_template); c->front_B = B; c->mpc.front_status = c->front.f.gs_status; c->mpc.front_B = B; return c->hipstate();
}
int qmhip_gait_insert_template(qmhip_ctx* c, int B, const int32_t* template_id, const double* start_time, const double* final_time) { QM_GUARD(c); QM_NEED_MPC(c);
  if (!c || B <= 0 || B != c->front_B || !template_id || !start_time || !final_time) { if (c) c->fail("qmhip_gait_insert_template: bad argument (B must be the batch of qmhip_gait_reset)"); return QMHIP_ERR_ARG; }
  hipSetDevice(c->device); c->front.gait_insert(B, template_id, start_time, final_time); return c->hipstate();
}
int qmhip_gait_update_resident(qmhip_ctx* c, int B, double horizon) { QM_GUARD(c); QM_NEED_MPC(c);
  if (!c || B <= 0 || B != c->front_B || !(horizon > 0)) { if (c) c->fail("qmhip_gait_update_resident: bad argument (B must be the batch of qmhip_gait_reset)"); return QMHIP_ERR_ARG; }
  hipSetDevice(c->device); c->gait_schedule(B, horizon); return c->hipstate();
}
int qmhip_gait_download(qmhip_ctx* c, int B, int32_t* n_events, double* event_times, int32_t* mode_sequence, int32_t* template_id, int32_t* status) { QM_GUARD(c); QM_NEED_MPC(c);
  if (!c || B <= 0 || B != c->front_B || !n_events || !event_times || !mode_sequence || !template_id || !status) return QMHIP_ERR_ARG;
  hipSetDevice(c->device); c->front.gait_download(B, n_events, event_times, mode_sequence, template_id, status); return c->hipstate();
}
int qmhip_schedule_download(qmhip_ctx* c, int B, double* ev, int32_t* modes) { QM_GUARD(c); QM_NEED_MPC(c);
  if (!c || B <= 0 || B > c->max_batch || !ev || !modes) return QMHIP_ERR_ARG;
  if (B != c->lastB && B != c->front_B) { c->fail("qmhip_schedule_download: B differs from the batch size the schedule buffers were filled for"); return QMHIP_ERR_STATE; }
  hipSetDevice(c->device); c->bk.to_host(ev, c->mpc.d.ev, (size_t)B * c->max_ev * 8); c->bk.to_host(modes, c->mpc.d.modes, (size_t)B * (c->max_ev + 1) * 4); return c->hipstate();
}
int qmhip_target_reset(qmhip_ctx* c, int B, const double* last_ee_target) { QM_GUARD(c); QM_NEED_MPC(c);
  if (!c || B <= 0 || B > c->max_batch || !last_ee_target) return QMHIP_ERR_ARG;
  hipSetDevice(c->device); c->front.target_reset(B, last_ee_target); return c->hipstate();
}
int qmhip_target_from_command(qmhip_ctx* c, int B, const int32_t* kind, const double* cmd, const double* ee_state, int ee_through_float, const qmhip_target_params* p) { QM_GUARD(c); QM_NEED_MPC(c);
  if (!c || B <= 0 || B > c->max_batch || !kind || !cmd || !p || !(p->target_displacement_velocity > 0) || !(p->target_rotation_velocity > 0) || c->max_ref < 2) {
    if (c) c->fail("qmhip_target_from_command: bad argument (max_ref_knots >= 2 required)"); return QMHIP_ERR_ARG; }
  hipSetDevice(c->device);
  c->front.target_from_command(c->mpc.d, B, kind, cmd, ee_state, ee_through_float, p->time_to_target, p->target_displacement_velocity, p->target_rotation_velocity, p->com_height);
  return c->hipstate();
}
int qmhip_target_download(qmhip_ctx* c, int B, double* ref_t, double* ref_x, double* last_ee_target) { QM_GUARD(c); QM_NEED_MPC(c);
  if (!c || B <= 0 || B > c->max_batch) return QMHIP_ERR_ARG;
  hipSetDevice(c->device);
  if (ref_t) c->bk.to_host(ref_t, c->mpc.d.ref_t, (size_t)B * c->max_ref * 8);
  if (ref_x) c->bk.to_host(ref_x, c->mpc.d.ref_x, (size_t)B * c->max_ref * QM_NREF * 8);
  if (last_ee_target) c->bk.to_host(last_ee_target, c->front.f.last_ee, (size_t)B * 7 * 8);
  return c->hipstate();
}
int qmhip_mpc_download(qmhip_ctx* c, int B, int32_t* nn, double* ot, int32_t* oev, int32_t* omode, double* ox, double* ou, double* operf, int32_t* status) { QM_GUARD(c); QM_NEED_MPC(c);
  if (!c || B <= 0 || B > c->max_batch) return QMHIP_ERR_ARG;
  if (!c->have_solution) { c->fail("qmhip_mpc_download: no solution available"); return QMHIP_ERR_STATE; }
  if (B != c->mpc.solved_B) { c->fail("qmhip_mpc_download: B differs from the batch size of the last solve (the solver buffers are strided by it)"); return QMHIP_ERR_STATE; }
  hipSetDevice(c->device); const int nm = c->max_nodes; const QmMpcBuffers& d = c->mpc.d;
  std::vector<int> n_h(B), st_h(B); c->bk.to_host(n_h.data(), d.n_nodes, (size_t)B * 4); c->bk.to_host(st_h.data(), d.status, (size_t)B * 4);
  std::vector<double> si((size_t)B * 4); c->bk.to_host(si.data(), d.step_info, si.size() * 8);
  for (int b = 0; b < B; ++b) { st_h[b] = qm_mpc_status(st_h[b], si.data() + (size_t)b * 4, c->st[ST_RICCATI_STRICT] != 0.0);      /* zeroed pivots of the degenerate stage: a warning; NaN / indefinite: -4 (qm_pipeline.h) */
    if (nn) nn[b] = n_h[b]; if (status) status[b] = st_h[b]; }
  // node-major [nmax][B][k] -> instance-major [B][nmax][k]: transposed by a kernel into a device staging buffer, copied through pinned memory (one contiguous
  // transfer per array; the per-node host loops this replaces took 20 ms of a 24 ms hand-over at B = 1024, bench.py `pcie_inclusive`)
  const size_t words = (size_t)nm * B * 30;
  if (c->dl_cap < words) { if (c->dl_dev) hipFree(c->dl_dev); if (c->dl_pin) hipHostFree(c->dl_pin); c->dl_dev = nullptr; c->dl_pin = nullptr; c->dl_cap = 0;
    HIP_TRY(c, hipMalloc(&c->dl_dev, words * 8)); HIP_TRY(c, hipHostMalloc(&c->dl_pin, words * 8, hipHostMallocDefault)); c->dl_cap = words; }
  auto gather = [&](const double* dev_d, const int* dev_i, int k, void* out) {
    if (!out) return; const size_t n = (size_t)nm * B * k, bytes = n * (dev_d ? 8 : 4);
    QmGatherArgs g; g.B = B; g.nmax = nm; g.k = k; g.src_d = dev_d; g.src_i = dev_i; g.dst_d = (double*)c->dl_dev; g.dst_i = (int*)c->dl_dev;
    c->bk.sync(); c->bk.launch(qm_gather_kernel, (int)((n + 255) / 256), 256, 0, g);
    c->bk.check(hipMemcpyAsync(c->dl_pin, c->dl_dev, bytes, hipMemcpyDeviceToHost, c->bk.stream), "D2H"); c->bk.check(hipStreamSynchronize(c->bk.stream), "sync");
    memcpy(out, c->dl_pin, bytes);
  };
  auto gather_d = [&](const double* dev, int k, double* out) { gather(dev, nullptr, k, out); };
  auto gather_i = [&](const int* dev, int32_t* out) { gather(nullptr, dev, 1, out); };
  gather_d(d.node_t, 1, ot); gather_i(d.node_ev, oev); gather_i(d.node_mode, omode); gather_d(d.xs, 30, ox); gather_d(d.us, 30, ou);
  if (operf) c->bk.to_host(operf, d.out_perf, (size_t)B * 10 * 8);
  return c->hipstate();
}
int qmhip_mpc_step(qmhip_ctx* c, int B, const double* t0, const double* x0, int n_ref, const double* ref_t, const double* ref_x, int n_ev, const double* ev, const int32_t* modes, double horizon,
                   int32_t* nn, double* ot, int32_t* oev, int32_t* omode, double* ox, double* ou, double* operf, int32_t* status) { QM_GUARD(c);
  int rc = qmhip_mpc_upload(c, B, t0, x0, n_ref, ref_t, ref_x, n_ev, ev, modes); if (rc) return rc;
  rc = qmhip_mpc_solve_resident(c, B, horizon); if (rc) return rc;
  return qmhip_mpc_download(c, B, nn, ot, oev, omode, ox, ou, operf, status);
}

int qmhip_policy_eval(qmhip_ctx* c, int B, const double* t, double* xd, double* ud, int32_t* mode) { QM_GUARD(c); QM_NEED_MPC(c);
  if (!c || B <= 0 || B > c->max_batch || !t) return QMHIP_ERR_ARG;
  if (!c->have_solution) { c->fail("qmhip_policy_eval: no policy received yet"); return QMHIP_ERR_STATE; }
  if (B != c->mpc.solved_B) { c->fail("qmhip_policy_eval: B differs from the batch size of the last solve"); return QMHIP_ERR_STATE; }
  hipSetDevice(c->device); c->wbc.policy_eval(c->mpc.d, B, t);
  if (xd) c->bk.to_host(xd, c->wbc.w.x_des, (size_t)B * 30 * 8); if (ud) c->bk.to_host(ud, c->wbc.w.u_des, (size_t)B * 30 * 8); if (mode) c->bk.to_host(mode, c->wbc.w.mode, (size_t)B * 4);
  return c->hipstate();
}
int qmhip_wbc_reset(qmhip_ctx* c) { QM_GUARD(c); if (!c) return QMHIP_ERR_ARG; hipSetDevice(c->device); c->bk.cur = c->bk.stream_b; c->wbc.reset(); c->bk.cur = c->bk.stream; return c->hipstate(); }   // ordered with the WBC launches
// The control-tick path (WbcBase::update on the ros_control thread, QMController.cpp:145-147).  Everything it enqueues goes to the WBC stream and the host waits for
// THAT stream only: inputs staged in pinned memory -> asynchronous copies -> qm_wbc_kernel -> asynchronous copy of [out | qp_status] -> one stream synchronisation.
// On a context that also runs the MPC the WBC stream first waits (on the device, not the host) for what the MPC stream has enqueued so far, because the resident
// step shares the WBC's input buffers; on a WBC-only context (qmhip_create_wbc_context) there is nothing to wait for and a tick never sees the MPC.
int qmhip_wbc_step(qmhip_ctx* c, int B, const double* xd, const double* ud, const double* rbd, const int32_t* mode, double period, const double* time, int variant, double* out, int32_t* qps) { QM_GUARD(c);
  if (!c || B <= 0 || B > c->max_batch || !xd || !ud || !rbd || !mode || !time || !(period > 0)) { if (c) c->fail("qmhip_wbc_step: bad argument"); return QMHIP_ERR_ARG; }
  hipSetDevice(c->device); HipBackend& bk = c->bk; QmWbcBuffers& w = c->wbc.w; typedef QmWbcPipeline<HipBackend> WP;
  // host staging, sections in the device buffers' order: [x_des | u_des | rbd | time | mode]
  char* hin = c->tick_pin; char* hout = c->tick_pin + WP::in_bytes(c->max_batch);
  double* h_xd = (double*)hin; double* h_ud = h_xd + (size_t)B * 30; double* h_rbd = h_ud + (size_t)B * 30; double* h_time = h_rbd + (size_t)B * QM_NRBD; int* h_mode = (int*)(h_time + B);
  memcpy(h_xd, xd, (size_t)B * 30 * 8); memcpy(h_ud, ud, (size_t)B * 30 * 8); memcpy(h_rbd, rbd, (size_t)B * QM_NRBD * 8); memcpy(h_time, time, (size_t)B * 8); memcpy(h_mode, mode, (size_t)B * 4);
  hipStream_t sb = bk.stream_b;
  if (!c->wbc_only) { bk.check(hipEventRecord(bk.ev_in, bk.stream), "hipEventRecord"); bk.check(hipStreamWaitEvent(sb, bk.ev_in, 0), "hipStreamWaitEvent"); }
  if (B == w.Bmax) bk.check(hipMemcpyAsync(w.x_des, hin, WP::in_bytes(B), hipMemcpyHostToDevice, sb), "H2D");
  else {
    bk.check(hipMemcpyAsync(w.x_des, h_xd, (size_t)B * 30 * 8, hipMemcpyHostToDevice, sb), "H2D"); bk.check(hipMemcpyAsync(w.u_des, h_ud, (size_t)B * 30 * 8, hipMemcpyHostToDevice, sb), "H2D");
    bk.check(hipMemcpyAsync(w.rbd, h_rbd, (size_t)B * QM_NRBD * 8, hipMemcpyHostToDevice, sb), "H2D"); bk.check(hipMemcpyAsync(w.time, h_time, (size_t)B * 8, hipMemcpyHostToDevice, sb), "H2D");
    bk.check(hipMemcpyAsync(w.mode, h_mode, (size_t)B * 4, hipMemcpyHostToDevice, sb), "H2D");
  }
  bk.cur = sb; c->wbc.step(c->mpc.d, B, period, variant); bk.wbc_end();
  double* h_out = (double*)hout; int* h_qps = (int*)(h_out + (size_t)B * QM_NWBC_OUT);
  if (B == w.Bmax) bk.check(hipMemcpyAsync(hout, w.out, WP::out_bytes(B), hipMemcpyDeviceToHost, sb), "D2H");
  else { bk.check(hipMemcpyAsync(h_out, w.out, (size_t)B * QM_NWBC_OUT * 8, hipMemcpyDeviceToHost, sb), "D2H"); bk.check(hipMemcpyAsync(h_qps, w.qp_status, (size_t)B * 3 * 4, hipMemcpyDeviceToHost, sb), "D2H"); }
  bk.check(hipStreamSynchronize(sb), "sync");
  if (out) memcpy(out, h_out, (size_t)B * QM_NWBC_OUT * 8); if (qps) memcpy(qps, h_qps, (size_t)B * 3 * 4);
  return c->hipstate();
}
// qm::HoQp on arbitrary task hierarchies (HoQp.h:17-36): B independent cascades of the same shape, solved by the general kernel (k_hoqp.h)
int qmhip_hoqp_solve(qmhip_ctx* c, int B, int n_levels, int n, const int32_t* ma, const int32_t* md, const double* A, const double* b, const double* D, const double* f, double* x, int32_t* status) { QM_GUARD(c);
  if (!c || B <= 0 || !ma || !md || !x || !status || !QmHoqpPipeline<HipBackend>::shapes_ok(n_levels, n, ma, md)) {
    if (c) c->fail("qmhip_hoqp_solve: bad argument (1 <= n <= 36 variables, <= 8 levels, <= 36 equality and <= 64 inequality rows per level, <= 128 inequality rows in total)"); return QMHIP_ERR_ARG; }
  int sa = 0, sd = 0; for (int k = 0; k < n_levels; ++k) { sa += ma[k]; sd += md[k]; }
  if ((sa && (!A || !b)) || (sd && (!D || !f))) { c->fail("qmhip_hoqp_solve: null task array"); return QMHIP_ERR_ARG; }
  hipSetDevice(c->device); c->hoqp.solve(B, n_levels, n, ma, md, A, b, D, f, x, status); return c->hipstate();
}
int qmhip_wbc_download(qmhip_ctx* c, int B, double* out, int32_t* qps) { QM_GUARD(c);
  if (!c || B <= 0 || B > c->max_batch) return QMHIP_ERR_ARG; hipSetDevice(c->device); c->flush_wbc();
  if (out) c->bk.to_host(out, c->wbc.w.out, (size_t)B * QM_NWBC_OUT * 8); if (qps) c->bk.to_host(qps, c->wbc.w.qp_status, (size_t)B * 3 * 4);
  return c->hipstate();
}
int qmhip_control_step_resident(qmhip_ctx* c, int B, double horizon, double period, double time) { QM_GUARD(c); QM_NEED_MPC(c);
  if (!c || B <= 0 || B > c->max_batch || !(horizon > 0)) { if (c) c->fail("qmhip_control_step_resident: bad argument"); return QMHIP_ERR_ARG; }
  // the WBC goes to its own stream: back-to-back steps overlap WBC(k) — one wave per SIMD whose run time is that of the instance with the most
  // active-set iterations — with the MPC kernels of step k + 1, which fill the SIMDs the finished WBC waves leave behind
  hipSetDevice(c->device); c->flush_wbc(); c->control_step(B, horizon, period, time, false);
  return c->hipstate();
}

// ---- batched rigid-body plant (QMHWSim.cpp:60-116) ----
int qmhip_sim_set_params(qmhip_ctx* c, const double* p, int n) { QM_GUARD(c); QM_NEED_MPC(c);
  if (!c || !p || n < 1 || n > 7) { if (c) c->fail("qmhip_sim_set_params: bad argument"); return QMHIP_ERR_ARG; }
  QmSimParams& q = c->sim.p; double* f[6] = {&q.k_n, &q.d_n, &q.mu, &q.v_eps, &q.foot_radius, &q.delay};
  for (int i = 0; i < n && i < 6; ++i) *f[i] = p[i]; if (n == 7) q.saturate = p[6] != 0.0;
  if (!(q.k_n >= 0) || !(q.d_n >= 0) || !(q.mu >= 0) || !(q.v_eps > 0) || !(q.delay >= 0)) { c->fail("qmhip_sim_set_params: negative parameter"); return QMHIP_ERR_ARG; }
  return QMHIP_OK;
}
int qmhip_sim_set_controller(qmhip_ctx* c, int controller) { QM_GUARD(c); QM_NEED_MPC(c);
  if (!c || (controller != 0 && controller != 1)) { if (c) c->fail("qmhip_sim_set_controller: 0 (QMController) or 1 (QMMpcController)"); return QMHIP_ERR_ARG; }
  c->sim.controller = controller; return QMHIP_OK;
}
int qmhip_sim_reset(qmhip_ctx* c, int B, const double* q, const double* v, const double* time) { QM_GUARD(c); QM_NEED_MPC(c);
  if (!c || B <= 0 || B > c->max_batch || !q || !v || !time) { if (c) c->fail("qmhip_sim_reset: bad argument"); return QMHIP_ERR_ARG; }
  hipSetDevice(c->device); c->sim.allocate(c->max_batch); c->sim.reset(B, q, v, time); c->sim_ticks = 0;
  c->mpc.solved_B = 0; c->have_solution = false;      // a new episode starts cold, like the reference after "Simulation reset" (no warm start from the previous episode's trajectory)
  c->sim.step(c->mpc.d.mb, B, 0.0, 0); return c->hipstate();   // rbd / contact of the reset state
}
int qmhip_sim_set_command(qmhip_ctx* c, int B, const double* pos_des, const double* vel_des, const double* kp, const double* kd, const double* ff) { QM_GUARD(c); QM_NEED_MPC(c);
  if (!c || B <= 0 || B > c->max_batch || !pos_des || !vel_des || !kp || !kd || !ff) { if (c) c->fail("qmhip_sim_set_command: bad argument"); return QMHIP_ERR_ARG; }
  if (!c->sim.s.Bmax) { c->fail("qmhip_sim_set_command: qmhip_sim_reset has not been called"); return QMHIP_ERR_STATE; }
  std::vector<double> cmd((size_t)B * (QM_SIM_CMD - 1)); const double* src[5] = {pos_des, vel_des, kp, kd, ff};
  for (int b = 0; b < B; ++b) for (int k = 0; k < 5; ++k) for (int j = 0; j < 18; ++j) cmd[(size_t)b * (QM_SIM_CMD - 1) + 18 * k + j] = src[k][(size_t)b * 18 + j];
  hipSetDevice(c->device); c->sim.set_command(B, cmd.data()); return c->hipstate();
}
int qmhip_sim_step(qmhip_ctx* c, int B, double period, int n_substeps, double* rbd, int32_t* contact) { QM_GUARD(c); QM_NEED_MPC(c);
  if (!c || B <= 0 || B > c->max_batch || !(period > 0) || n_substeps < 1) { if (c) c->fail("qmhip_sim_step: bad argument"); return QMHIP_ERR_ARG; }
  if (!c->sim.s.Bmax) { c->fail("qmhip_sim_step: qmhip_sim_reset has not been called"); return QMHIP_ERR_STATE; }
  hipSetDevice(c->device); c->sim.step(c->mpc.d.mb, B, period, n_substeps);
  if (rbd) c->bk.to_host(rbd, c->sim.s.rbd, (size_t)B * QM_NRBD * 8); if (contact) c->bk.to_host(contact, c->sim.s.contact, (size_t)B * 4 * 4);
  return c->hipstate();
}
int qmhip_sim_get_state(qmhip_ctx* c, int B, double* q, double* v, double* time, double* force, int32_t* status) { QM_GUARD(c); QM_NEED_MPC(c);
  if (!c || B <= 0 || B > c->max_batch) { if (c) c->fail("qmhip_sim_get_state: bad argument"); return QMHIP_ERR_ARG; }
  if (!c->sim.s.Bmax) { c->fail("qmhip_sim_get_state: qmhip_sim_reset has not been called"); return QMHIP_ERR_STATE; }
  hipSetDevice(c->device);
  if (q) c->bk.to_host(q, c->sim.s.q, (size_t)B * 24 * 8); if (v) c->bk.to_host(v, c->sim.s.v, (size_t)B * 24 * 8); if (time) c->bk.to_host(time, c->sim.s.time, (size_t)B * 8);
  if (force) c->bk.to_host(force, c->sim.s.force, (size_t)B * 12 * 8); if (status) c->bk.to_host(status, c->sim.s.status, (size_t)B * 4);
  return c->hipstate();
}

// device-resident control loop around the plant: per tick [state estimate (ground truth) -> MPC call every mpc_every ticks (warm-started SQP on the observation)
// -> policy at the plant time -> WBC on the measured state -> hybrid joint command -> one simulation step]; QMController::update + mpcThread_
// (qm_controllers/src/QMController.cpp:128-175, 315-332), the MPC synchronous with the tick it is triggered on
int qmhip_closed_loop_sim(qmhip_ctx* c, int B, int n_ticks, double period, int n_substeps, int mpc_every, double horizon, double arm_kp, double arm_kd) { QM_GUARD(c); QM_NEED_MPC(c);
  if (!c || B <= 0 || B > c->max_batch || n_ticks <= 0 || !(period > 0) || n_substeps < 1 || mpc_every < 1 || !(horizon > 0)) { if (c) c->fail("qmhip_closed_loop_sim: bad argument"); return QMHIP_ERR_ARG; }
  if (!c->sim.s.Bmax) { c->fail("qmhip_closed_loop_sim: qmhip_sim_reset has not been called"); return QMHIP_ERR_STATE; }
  hipSetDevice(c->device); c->bk.sync();
  qm_closed_loop_sim_ticks(c->bk, c->mpc, c->wbc, c->sim, c->sim_ticks, B, n_ticks, period, n_substeps, mpc_every, horizon, arm_kp, arm_kd, c->sqp_iterations(),
                           [&]() { if (c->front_B == B) c->gait_schedule(B, horizon); });
  c->lastB = B; c->have_solution = true;
  return c->hipstate();
}

int qmhip_closed_loop_sim_pipelined(qmhip_ctx* c, int B, int n_ticks, double period, int n_substeps, int mpc_every, double horizon, double arm_kp, double arm_kd) { QM_GUARD(c); QM_NEED_MPC(c);
  if (!c || B <= 0 || B > c->max_batch || n_ticks <= 0 || !(period > 0) || n_substeps < 1 || mpc_every < 1 || !(horizon > 0)) { if (c) c->fail("qmhip_closed_loop_sim_pipelined: bad argument"); return QMHIP_ERR_ARG; }
  if (!c->sim.s.Bmax) { c->fail("qmhip_closed_loop_sim_pipelined: qmhip_sim_reset has not been called"); return QMHIP_ERR_STATE; }
  if (n_ticks % mpc_every || c->sim_ticks % mpc_every) { c->fail("qmhip_closed_loop_sim_pipelined: n_ticks and the tick counter must be multiples of mpc_every"); return QMHIP_ERR_ARG; }
  hipSetDevice(c->device); c->bk.sync();
  qm_closed_loop_sim_pipelined(c->bk, c->mpc, c->wbc, c->sim, c->sim_ticks, B, n_ticks, period, n_substeps, mpc_every, horizon, arm_kp, arm_kd, c->sqp_iterations(),
                               [&]() { if (c->front_B == B) c->gait_schedule(B, horizon); });
  c->lastB = B; c->have_solution = true; c->bk.sync();
  return c->hipstate();
}

int qmhip_set_profiling(qmhip_ctx* c, int en) { QM_GUARD(c); if (!c) return QMHIP_ERR_ARG; c->bk.resolve(); c->bk.profiling = (en == 2 || en == 3) ? en : (en != 0); return QMHIP_OK; }
int qmhip_get_kernel_ms(qmhip_ctx* c, const char* name, double* ms, int* launches) { QM_GUARD(c);
  if (!c || !name) return QMHIP_ERR_ARG; hipSetDevice(c->device); c->bk.resolve(); auto it = c->bk.acc.find(name);
  if (ms) *ms = it == c->bk.acc.end() ? 0.0 : it->second.first; if (launches) *launches = it == c->bk.acc.end() ? 0 : it->second.second; return QMHIP_OK;
}
int qmhip_reset_kernel_ms(qmhip_ctx* c) { QM_GUARD(c); if (!c) return QMHIP_ERR_ARG; c->bk.resolve(); c->bk.acc.clear(); return QMHIP_OK; }
int qmhip_synchronize(qmhip_ctx* c) { QM_GUARD(c); if (!c) return QMHIP_ERR_ARG; hipSetDevice(c->device); c->flush_wbc(); c->bk.sync(); return c->hipstate(); }
int qmhip_last_ls_trials(const qmhip_ctx* c) { QM_GUARD(c); if (!c) return -1; hipSetDevice(c->device); return const_cast<qmhip_ctx*>(c)->mpc.ls_trials(); }      // (after a device-side line search: one synchronising read of the trial counters)
int qmhip_debug_set(qmhip_ctx* c, const char* key, int value) { QM_GUARD(c); if (!c || !key) return QMHIP_ERR_ARG; if (!strcmp(key, "lq_slices")) { c->mpc.lq_slices = value < 1 ? 1 : value; return QMHIP_OK; } if (!strcmp(key, "riccati_skip")) { c->mpc.riccati_skip = value; return QMHIP_OK; } if (!strcmp(key, "wbc_stop")) { c->wbc.wbc_stop = value; return QMHIP_OK; } if (!strcmp(key, "lq_prof")) { c->mpc.lq_prof = value; return QMHIP_OK; }
  if (!strcmp(key, "r_dense")) { c->mpc.r_force_dense = value != 0; return QMHIP_OK; }      // 1: the dense instances of the trial evaluation and of K1b's R0 (u − u_nom) although R is block diagonal (tests: same bits)
  if (!strcmp(key, "fused_policy")) { c->fused_policy = value != 0; return QMHIP_OK; }      // 0: apply -> qm_policy_kernel -> WBC as in rounds 1-5 (A/B, tests)
  if (!strcmp(key, "filler_at_lq")) { c->filler_at_lq = value; return QMHIP_OK; }
  if (!strcmp(key, "filler_live")) { c->filler_live = value; return QMHIP_OK; } if (!strcmp(key, "filler_lds")) { if (value < 16 * 1024 || value > 48 * 1024) return QMHIP_ERR_ARG; c->filler_lds = value; return QMHIP_OK; }      // co-residency stand-in: 0 / 160 / 184 live doubles, LDS bytes (>= 16 KB: the kernel indexes 2048 doubles)
  if (!strcmp(key, "wbc_defer")) { c->flush_wbc(); c->wbc_defer = value; if (!value) c->mpc.before_lq = nullptr; return QMHIP_OK; }      // scheduling experiment (profiles/r06_ab_wbc_schedule.log)
  if (!strcmp(key, "ls_device_tail")) { c->mpc.device_tail = value != 0; return QMHIP_OK; }      // 0: the host-driven line-search trial loop of rounds 1-5 (A/B, tests); 1 (default): the trials after the first in one launch (k_ls.h)
  if (!strncmp(key, "lds_pad:", 8)) {   // profiling only: "lds_pad:<kernel group>" (lq, lq_kin, riccati, ls_eval, wbc, ...) = extra dynamic LDS bytes per workgroup; 0 removes it
    if (value < 0 || value > 160 * 1024) return QMHIP_ERR_ARG; if (value) c->bk.lds_pad[key + 8] = value; else c->bk.lds_pad.erase(key + 8); return QMHIP_OK; }
  if (!strcmp(key, "lq_debug")) {      // parity tests: K1b additionally writes the UNPROJECTED LQ model of every interval (buffer "lqdbg", [B][max_nodes][LQ_DBG_SIZE]) and Pu / the zero rows of Px into the stage record
    QM_NEED_MPC(c); hipSetDevice(c->device); c->bk.sync();
    if (value && !c->mpc.d.lqdbg) c->mpc.d.lqdbg = c->mpc.A<double>((size_t)c->max_nodes * c->max_batch * LQ_DBG_SIZE);
    if (!value && c->mpc.d.lqdbg) { c->bk.free(c->mpc.d.lqdbg); c->mpc.d.lqdbg = nullptr; }
    c->bk.sync(); return c->hipstate();
  }
  return QMHIP_ERR_ARG; }
// profiling only: one MPC iteration of the resident batch with the filler started on the second stream right when the LQ kernel starts; ms[0] = LQ kernel,
// ms[1] = filler, ms[2] = Riccati kernel, all from HIP events
int qmhip_debug_lq_with_filler(qmhip_ctx* c, int B, double horizon, int waves, int iters, double* ms) { QM_GUARD(c); QM_NEED_MPC(c);
  if (!c || B <= 0 || B > c->max_batch || !ms) return QMHIP_ERR_ARG; hipSetDevice(c->device);
  if (waves > 0 && !c->filler_buffer(waves)) { c->fail("hipMalloc of the filler buffer failed"); return QMHIP_ERR_HIP; }
  double* out = c->filler_out;
  hipEvent_t a0 = nullptr, b0 = nullptr, b1 = nullptr;
  if (hipEventCreate(&a0) != hipSuccess || hipEventCreate(&b0) != hipSuccess || hipEventCreate(&b1) != hipSuccess) { c->fail("hipEventCreate failed"); return QMHIP_ERR_HIP; }
  c->bk.sync();
  bool fired = false;      // the hook runs once per sqp iteration; the filler is started by the first call only
  c->mpc.before_lq = [&]() {
    if (fired) return; fired = true;
    hipEventRecord(a0, c->bk.stream); hipStreamWaitEvent(c->bk.stream_b, a0, 0);      // the filler starts when everything before the LQ kernel is done
    hipEventRecord(b0, c->bk.stream_b);
    if (waves > 0) hipLaunchKernelGGL(qm_filler_kernel, dim3(waves), dim3(64), 20 * 1024, c->bk.stream_b, out, iters);
    hipEventRecord(b1, c->bk.stream_b);
  };
  c->mpc.grid(B, horizon);
  // one iteration with a span around every launch: the LQ and Riccati kernel times come from those spans
  const int prof = c->bk.profiling; c->bk.resolve(); c->bk.acc.clear(); c->bk.profiling = 1;
  c->mpc.sqp_iteration(B, 14, true);
  c->bk.sync(); c->bk.resolve(); c->bk.profiling = prof; c->mpc.before_lq = nullptr;
  float fb = 0; hipEventElapsedTime(&fb, b0, b1);
  ms[0] = c->bk.acc["lq"].first; ms[1] = fb; ms[2] = c->bk.acc["riccati"].first;
  hipEventDestroy(a0); hipEventDestroy(b0); hipEventDestroy(b1);
  return c->hipstate();
}
// profiling only: `waves` filler waves of `iters` steps on the second stream (see qm_filler_kernel); *ms (may be null) = its duration when waited for
int qmhip_debug_filler(qmhip_ctx* c, int waves, int iters, int wait, double* ms) { QM_GUARD(c);
  if (!c || waves <= 0 || iters <= 0) return QMHIP_ERR_ARG; hipSetDevice(c->device);
  if (!c->filler_buffer(waves)) { c->fail("hipMalloc of the filler buffer failed"); return QMHIP_ERR_HIP; }
  double* out = c->filler_out;
  if (!wait && c->filler_at_lq) {      // "filler_at_lq": the stand-in of step k goes out BEHIND K1a of the next solve (between its kin and LQ launches), i.e. beside the LQ kernel only
    qmhip_ctx* cc = c; c->mpc.before_lq = [cc, waves, iters]() { cc->mpc.before_lq = nullptr; const int keep = cc->filler_at_lq; cc->filler_at_lq = 0; qmhip_debug_filler(cc, waves, iters, 0, nullptr); cc->filler_at_lq = keep; };
    return QMHIP_OK; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  if (!wait) { hipEventRecord(c->bk.ev_in, c->bk.stream); hipStreamWaitEvent(c->bk.stream_b, c->bk.ev_in, 0); }      // like the WBC of a control step: starts when everything enqueued on the MPC stream so far is done
  hipEventRecord(e0, c->bk.stream_b);
  if (c->filler_live) {
    if (!c->filler_in) { if (hipMalloc(&c->filler_in, 192 * 64 * 8) != hipSuccess) { c->fail("hipMalloc failed"); return QMHIP_ERR_HIP; } hipMemset(c->filler_in, 0, 192 * 64 * 8); }
    const int lds = c->filler_lds;
    if (c->filler_live >= 184) hipLaunchKernelGGL(qm_filler_wide392_kernel, dim3(waves), dim3(64), lds, c->bk.stream_b, out, c->filler_in, iters);
    else hipLaunchKernelGGL(qm_filler_wide344_kernel, dim3(waves), dim3(64), lds, c->bk.stream_b, out, c->filler_in, iters);
  } else hipLaunchKernelGGL(qm_filler_kernel, dim3(waves), dim3(64), c->filler_lds, c->bk.stream_b, out, iters);
  hipEventRecord(e1, c->bk.stream_b);
  if (wait) { hipEventSynchronize(e1); float f = 0; hipEventElapsedTime(&f, e0, e1); if (ms) *ms = f; }
  hipEventDestroy(e0); hipEventDestroy(e1);
  return c->hipstate();
}
int qmhip_debug_get(const qmhip_ctx* c, const char* key, int* value) { QM_GUARD(c);
  if (!c || !key || !value) return QMHIP_ERR_ARG;
  if (!strcmp(key, "riccati_skip")) { *value = c->mpc.riccati_skip; return QMHIP_OK; } if (!strcmp(key, "wbc_stop")) { *value = c->wbc.wbc_stop; return QMHIP_OK; } if (!strcmp(key, "lq_prof")) { *value = c->mpc.lq_prof; return QMHIP_OK; }
  if (!strcmp(key, "lq_debug")) { *value = c->mpc.d.lqdbg ? 1 : 0; return QMHIP_OK; }
  if (!strcmp(key, "lds_pad")) { *value = (int)c->bk.lds_pad.size(); return QMHIP_OK; }      // number of kernel groups running with padded LDS
  if (!strcmp(key, "ls_device_tail")) { *value = c->mpc.device_tail ? 1 : 0; return QMHIP_OK; } if (!strcmp(key, "fused_policy")) { *value = c->fused_policy ? 1 : 0; return QMHIP_OK; }      // launch-order switches (A/B, tests): 1 is the product's order
  if (!strcmp(key, "r_dense")) { *value = c->mpc.r_force_dense ? 1 : 0; return QMHIP_OK; } if (!strcmp(key, "r_blocks")) { *value = c->mpc.rblk() ? 1 : 0; return QMHIP_OK; }      // r_blocks: the structured instances are the ones that run
  if (!strcmp(key, "wbc_defer")) { *value = c->wbc_defer; return QMHIP_OK; } if (!strcmp(key, "filler_at_lq")) { *value = c->filler_at_lq; return QMHIP_OK; }
  return QMHIP_ERR_ARG;
}
int qmhip_debug_read(qmhip_ctx* c, const char* name, void* dst, size_t bytes) { QM_GUARD(c);
  if (!c || !name || !dst) return QMHIP_ERR_ARG; hipSetDevice(c->device); const QmMpcBuffers& d = c->mpc.d; const void* p = nullptr;
#define F(n) if (!strcmp(name, #n)) p = d.n;
  if (c->wbc_only) { p = c->wbc.buffer(name); if (!p) { c->fail("qmhip_debug_read: a WBC-only context only has the wbc_* buffers"); return QMHIP_ERR_ARG; } c->bk.to_host(dst, p, bytes); return c->hipstate(); }
  if (!strcmp(name, "sim_rbd")) p = c->sim.s.rbd; if (!strcmp(name, "sim_cmd")) p = c->sim.s.cmd;
  F(lqdbg) F(n_nodes) F(node_t) F(node_ts) F(node_dt) F(node_ev) F(node_mode) F(zvel) F(zpos) F(xref) F(eeref) F(status) F(x) F(u) F(dx) F(du) F(stage) F(perf) F(base_sum) F(perf_sum) F(step_info) F(alpha) F(done) F(xs) F(us) F(out_perf) F(t0) F(x0) F(ipm_s) F(ipm_l) F(ipm_ds) F(ipm_dl) F(ipm_info)
#undef F
  if (!p) p = c->wbc.buffer(name);
  if (!p) { c->fail(std::string("qmhip_debug_read: unknown buffer ") + name); return QMHIP_ERR_ARG; }
  c->bk.to_host(dst, p, bytes); return c->hipstate();
}
int qmhip_microbench_fp64(qmhip_ctx* c, int use_mfma, double* tflops) { QM_GUARD(c);
  if (!c || !tflops) return QMHIP_ERR_ARG; hipSetDevice(c->device);
  const int blocks = 256 * 8, threads = 256, iters = 20000; double* out = (double*)c->bk.alloc((size_t)blocks * threads * 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a, c->bk.stream);
    if (use_mfma) hipLaunchKernelGGL(qm_bench_mfma_kernel, dim3(blocks), dim3(threads), 0, c->bk.stream, out, iters);
    else hipLaunchKernelGGL(qm_bench_fma_kernel, dim3(blocks), dim3(threads), 0, c->bk.stream, out, iters);
    hipEventRecord(b, c->bk.stream); hipEventSynchronize(b);
  }
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  const double flops = use_mfma ? (double)blocks * (threads / 64) * iters * 4.0 * (2.0 * 16 * 16 * 4) : (double)blocks * threads * iters * 8.0 * 2.0;
  *tflops = flops / (ms * 1e-3) / 1e12; hipEventDestroy(a); hipEventDestroy(b); c->bk.free(out); return c->hipstate();
}

}  // extern "C"
