// tests/emu/hip_emu.h — TEST INFRASTRUCTURE ONLY.
//
// A minimal host-side executor for the device code under qm_control_amd/csrc/kernels/, so the kernels'
// logic can be parity-tested against the oracle in this GPU-less container (-m "not gpu" tests).
// It is NOT a compatibility layer of the product: the product is compiled by hipcc for gfx950 only and
// never sees this file.  Each GPU thread of a block runs as a ucontext fiber; __syncthreads(), wave
// shuffles and __builtin_amdgcn_mfma_f64_16x16x4f64 are implemented with fiber barriers.  Blocks run
// one after another on the calling OS thread.
#pragma once
#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>
#include <algorithm>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__
#define __restrict__
#define __launch_bounds__(...)
#define QM_MAX_VGPRS(n)            /* register caps mean nothing on the host */
#define __builtin_amdgcn_sched_barrier(m) ((void)0)   /* instruction-scheduling fence: no meaning on the host */
#define QM_TABLE_OPAQUE(p)            /* device-only register constraint */
#define QM_UNPAIRED_LDS               /* device-only kernel attribute */
#define QM_LANE_OPAQUE(i)             /* device-only register constraint */
#define QM_PIN4(q) ((void)0)             /* device-only scheduling pin */
#define QM_LOADED(d)                  /* device-only: "this value is loaded here" */
#define QM_SCALARS_READY(a, b, c, d)  /* device-only: "these wave-uniform values are in scalar registers here" */

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };

namespace emu {
struct Fiber { ucontext_t ctx; dim3 tid; int lin, wave, lane; bool done; char* stack; };
struct Block {
  dim3 bid, bdim, gdim; std::vector<Fiber> fibers; ucontext_t sched; int cur;
  int bar_count, bar_gen; std::vector<int> wbar_count, wbar_gen;
  std::vector<double> xa, xb;      // per-wave exchange buffers [nwaves][64]
  std::vector<unsigned long long> xi;
  std::function<void()> body;
};
extern Block* B;
inline Fiber& cur() { return B->fibers[B->cur]; }
inline void yield() { swapcontext(&cur().ctx, &B->sched); }
inline void syncthreads() {
  Block* b = B; const int gen = b->bar_gen; int alive = 0; for (auto& f : b->fibers) alive += !f.done;
  if (++b->bar_count == alive) { b->bar_count = 0; ++b->bar_gen; return; }
  while (b->bar_gen == gen) yield();
}
inline void wavesync() {
  Block* b = B; const int w = cur().wave; const int gen = b->wbar_gen[w];
  int n = 0; for (auto& f : b->fibers) n += (f.wave == w && !f.done);
  if (++b->wbar_count[w] == n) { b->wbar_count[w] = 0; ++b->wbar_gen[w]; return; }
  while (b->wbar_gen[w] == gen) yield();
}
void trampoline();
template <class F> void launch(dim3 grid, dim3 block, F&& body);
extern double* dyn_smem;
}  // namespace emu

#define threadIdx (emu::cur().tid)
#define blockIdx (emu::B->bid)
#define blockDim (emu::B->bdim)
#define gridDim (emu::B->gdim)
inline void __syncthreads() { emu::syncthreads(); }

// dynamic LDS: every kernel declares `extern __shared__ double qm_smem[];`
extern double qm_smem[];

// compiled with ROCm's clang++ for the host, so ext_vector_type is available like in device code
typedef double double4v __attribute__((ext_vector_type(4)));
#define __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, x, y, z) emu_mfma_f64_16x16x4((a), (b), (c))
#define __builtin_amdgcn_wave_barrier() emu::wavesync()
#define __popcll(x) __builtin_popcountll(x)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_s_memrealtime() 0ull          /* 100 MHz reference clock / hardware slot id: profiling stamps of the instrumented instances only */
#define __builtin_amdgcn_s_getreg(x) 0u
// global_load_lds: lane l's `size` bytes land at lds_base + size * l; the immediate offset `off` is added on BOTH sides
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) memcpy((char*)(l) + (off) + (size) * emu::cur().lane, (const char*)(g) + (off), (size))
#define __ffsll(x) __builtin_ffsll(x)
#define __builtin_amdgcn_rsq(x) (1.0 / std::sqrt((double)(x)))
#define __builtin_amdgcn_rcp(x) (1.0 / (double)(x))
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __threadfence() ((void)0)
#define __threadfence_system() ((void)0)
inline int atomicAdd(int* p, int v) { const int o = *p; *p += v; return o; }
inline int atomicMax(int* p, int v) { const int o = *p; if (v > o) *p = v; return o; }   /* blocks run one after the other on the host */
inline int atomicOr(int* p, int v) { const int o = *p; *p |= v; return o; }
struct double2 { double x, y; };
inline double2 make_double2(double a, double b) { double2 r; r.x = a; r.y = b; return r; }

// v_mfma_f64_16x16x4_f64: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[row=(l>>4)+4r][col=l&15]
// (cdna_hip_programming.md §3 "f64 MFMA does NOT use these maps")
inline double4v emu_mfma_f64_16x16x4(double a, double b, double4v c) {
  emu::Block* blk = emu::B; const int w = emu::cur().wave, l = emu::cur().lane;
  blk->xa[w * 64 + l] = a; blk->xb[w * 64 + l] = b;
  emu::wavesync();
  double4v d;
  for (int r = 0; r < 4; ++r) {
    const int row = (l >> 4) + 4 * r, col = l & 15; double s = c[r];
    for (int k = 0; k < 4; ++k) s = std::fma(blk->xa[w * 64 + row + 16 * k], blk->xb[w * 64 + col + 16 * k], s);
    d[r] = s;
  }
  emu::wavesync();
  return d;
}
inline double __shfl(double v, int src, int width = 64) {
  emu::Block* blk = emu::B; const int w = emu::cur().wave, l = emu::cur().lane;
  blk->xa[w * 64 + l] = v; emu::wavesync();
  const int base = (l / width) * width; const double r = blk->xa[w * 64 + base + (src % width)];
  emu::wavesync(); return r;
}
inline double __shfl_xor(double v, int mask, int width = 64) { return __shfl(v, (emu::cur().lane % width) ^ mask, width); }
inline double __shfl_down(double v, int delta, int width = 64) { const int l = emu::cur().lane % width; return __shfl(v, (l + delta < width) ? l + delta : l, width); }
inline int __shfl(int v, int src, int width = 64);
#define __builtin_amdgcn_readlane(v, lane) __shfl((int)(v), (int)(lane), 64)
// DPP: only the controls the kernels use (row_shr:n, row_bcast:15, row_bcast:31), bank mask 0xf
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  emu::Block* blk = emu::B; const int w = emu::cur().wave, l = emu::cur().lane;
  blk->xa[w * 64 + l] = (double)src; emu::wavesync();
  const int row = l >> 4, pos = l & 15; int r = old;
  if ((row_mask >> row) & 1) {
    if (ctrl >= 0x111 && ctrl <= 0x11f) { const int sh = ctrl - 0x110; if (pos >= sh) r = (int)blk->xa[w * 64 + l - sh]; else if (bound_ctrl) r = 0; }
    else if (ctrl == 0x142) { if (row >= 1) r = (int)blk->xa[w * 64 + 16 * row - 1]; }
    else if (ctrl == 0x143) { if (row >= 2) r = (int)blk->xa[w * 64 + 31]; }
    else { fprintf(stderr, "emu: unsupported dpp ctrl %x\n", ctrl); abort(); }
  }
  emu::wavesync(); return r;
}
// v_mov_b32_dpp without an `old` operand: the kernels use it with bound_ctrl and a full row mask only (zeros shifted in)
inline int __builtin_amdgcn_mov_dpp(int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) { return __builtin_amdgcn_update_dpp(0, src, ctrl, row_mask, bank_mask, bound_ctrl); }
inline unsigned long long __ballot(int pred) {
  emu::Block* blk = emu::B; const int w = emu::cur().wave, l = emu::cur().lane;
  blk->xa[w * 64 + l] = pred ? 1.0 : 0.0; emu::wavesync();
  unsigned long long m = 0ull; for (int i = 0; i < 64; ++i) if (blk->xa[w * 64 + i] != 0.0) m |= (1ull << i);
  emu::wavesync(); return m;
}
inline int __shfl(int v, int src, int width) { return (int)__shfl((double)v, src, width); }
inline int __shfl_xor(int v, int mask, int width = 64) { return (int)__shfl_xor((double)v, mask, width); }

using std::sqrt; using std::sin; using std::cos; using std::fabs; using std::fma; using std::acos; using std::log; using std::fmin; using std::fmax; using std::pow;

namespace emu {
template <class F> void launch(dim3 grid, dim3 block, F&& body) {
  const int nthreads = block.x * block.y * block.z; const int nwaves = (nthreads + 63) / 64;
  static std::vector<char*> stacks; const size_t STK = 1 << 20;
  while ((int)stacks.size() < nthreads) stacks.push_back((char*)malloc(STK));
  for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned bx = 0; bx < grid.x; ++bx) {
    Block blk; B = &blk; blk.bid = dim3(bx, by, bz); blk.bdim = block; blk.gdim = grid; blk.bar_count = 0; blk.bar_gen = 0;
    blk.wbar_count.assign(nwaves, 0); blk.wbar_gen.assign(nwaves, 0); blk.xa.assign(nwaves * 64, 0.0); blk.xb.assign(nwaves * 64, 0.0);
    blk.body = body; blk.fibers.resize(nthreads);
    for (int t = 0; t < nthreads; ++t) {
      Fiber& f = blk.fibers[t]; f.lin = t; f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
      f.wave = t / 64; f.lane = t % 64; f.done = false; f.stack = stacks[t];
      getcontext(&f.ctx); f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = STK; f.ctx.uc_link = &blk.sched;
      makecontext(&f.ctx, (void (*)())trampoline, 0);
    }
    int remaining = nthreads;
    while (remaining > 0) {
      for (int t = 0; t < nthreads; ++t) {
        if (blk.fibers[t].done) continue;
        blk.cur = t; swapcontext(&blk.sched, &blk.fibers[t].ctx);
        if (blk.fibers[t].done) --remaining;
      }
    }
    B = nullptr;
  }
}
}  // namespace emu
