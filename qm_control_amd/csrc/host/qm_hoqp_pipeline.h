// qm_hoqp_pipeline.h — launch of the general HoQp cascade (k_hoqp.h), backend-templated like the other pipelines.
#pragma once
#include <vector>
#include "qm_pipeline.h"
#include "../kernels/k_hoqp.h"

template <class BK>
struct QmHoqpPipeline {
  BK& bk; int cap = 0; double* ws = nullptr; int* wlist = nullptr; double* A = nullptr; double* b = nullptr; double* D = nullptr; double* f = nullptr; double* x = nullptr; int* status = nullptr;
  size_t capA = 0, capD = 0, capB = 0, capF = 0;      // one capacity per task buffer: A / D scale with rows * n, b / f with rows alone (n = 36, ma = 10 then n = 10, ma = 36 must not fit by accident)
  explicit QmHoqpPipeline(BK& k) : bk(k) {}
  void release() { void* ps[] = {ws, wlist, A, b, D, f, x, status}; for (void* p : ps) if (p) bk.free(p); ws = nullptr; wlist = nullptr; A = b = D = f = x = nullptr; status = nullptr; cap = 0; capA = capD = capB = capF = 0; }
  // shapes are validated by the caller (qmhip_hoqp_solve); returns false when they exceed the kernel's compile-time limits
  static bool shapes_ok(int n_levels, int n, const int* ma, const int* md) {
    if (n_levels < 1 || n_levels > HQ_LEVELS || n < 1 || n > HQ_NMAX) return false;
    int mh = 0; for (int k = 0; k < n_levels; ++k) { if (ma[k] < 0 || ma[k] > HQ_MAMAX || md[k] < 0 || md[k] > HQ_MDMAX) return false; mh += md[k]; if (mh > HQ_MHMAX) return false; }
    return true;
  }
  void solve(int B, int n_levels, int n, const int* ma, const int* md, const double* hA, const double* hb, const double* hD, const double* hf, double* hx, int* hstatus) {
    int sa = 0, sd = 0; for (int k = 0; k < n_levels; ++k) { sa += ma[k]; sd += md[k]; }
    const size_t nA = (size_t)B * (sa ? sa : 1) * n, nD = (size_t)B * (sd ? sd : 1) * n, nB = (size_t)B * (sa ? sa : 1), nF = (size_t)B * (sd ? sd : 1);
    if (B > cap || nA > capA || nD > capD || nB > capB || nF > capF) {
      const int ncap = B > cap ? B : cap; const size_t a_ = nA > capA ? nA : capA, d_ = nD > capD ? nD : capD, b_ = nB > capB ? nB : capB, f_ = nF > capF ? nF : capF;
      release(); cap = ncap; capA = a_; capD = d_; capB = b_; capF = f_;
      ws = (double*)bk.alloc((size_t)cap * HQW_SIZE * 8); wlist = (int*)bk.alloc((size_t)cap * HQ_NCMAX * 4);
      A = (double*)bk.alloc(capA * 8); b = (double*)bk.alloc(capB * 8); D = (double*)bk.alloc(capD * 8); f = (double*)bk.alloc(capF * 8);
      x = (double*)bk.alloc((size_t)cap * HQ_NMAX * 8); status = (int*)bk.alloc((size_t)cap * HQ_LEVELS * 4);
    }
    if (sa) { bk.to_device(A, hA, (size_t)B * sa * n * 8); bk.to_device(b, hb, (size_t)B * sa * 8); }
    if (sd) { bk.to_device(D, hD, (size_t)B * sd * n * 8); bk.to_device(f, hf, (size_t)B * sd * 8); }
    QmHoqpArgs a; a.B = B; a.n_levels = n_levels; a.n = n; for (int k = 0; k < HQ_LEVELS; ++k) { a.ma[k] = k < n_levels ? ma[k] : 0; a.md[k] = k < n_levels ? md[k] : 0; }
    a.A = A; a.b = b; a.D = D; a.f = f; a.sum_ma = sa; a.sum_md = sd; a.ws = ws; a.wlist = wlist; a.x = x; a.status = status;
    bk.launch(qm_hoqp_kernel, B, 64, 0, a);
    bk.to_host(hx, x, (size_t)B * n * 8); bk.to_host(hstatus, status, (size_t)B * n_levels * 4);
  }
};
