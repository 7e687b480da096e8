// qm_dev_common.h — device-side building blocks shared by the gfx950 kernels of the MPC+WBC hot path.
//
//  * LDS "tile": a 32x32 f64 matrix stored row-major with leading dimension QM_LD = 34 doubles
//    (34 keeps the ds_read_b64 operand-fragment reads of v_mfma_f64_16x16x4_f64 conflict free for the
//    [i][k] pattern and at most 2-way for the [k][j] pattern; MI355X_MICROARCH.md §LDS).
//    nx = nu = 30 and every constraint / projected-input dimension is <= 18, so every matrix of the path
//    fits one tile.  Padding rows/cols are kept exactly zero.
//  * wg_gemm: C(16·mt x 16·nt) = op(A)·op(B) on the matrix cores; a 256-thread workgroup's 4 waves take
//    one 16x16 output tile each (64-wide wavefronts, one MFMA accumulator of 4 f64 per lane).
//    Fragment maps (cdna_hip_programming.md §3): A[i=l&15][k=l>>4], B[k=l>>4][j=l&15],
//    D[row=(l>>4)+4r][col=l&15].
#pragma once
#ifndef QM_MAX_VGPRS
#define QM_MAX_VGPRS(n) __attribute__((amdgpu_waves_per_eu(512 / (n), 512 / (n))))      /* cap a kernel's vector registers through the occupancy it asks for (the host emulator defines this away) */
#endif
#include <hip/hip_runtime.h>
#include "../../../include/qmhip_layout.h"

#define QM_LD 34
#define QM_TILE (32 * QM_LD)          /* doubles per LDS tile */
#define QM_BLOCK 256                  /* threads per workgroup for the per-node / per-instance kernels */

typedef double qm_d4 __attribute__((ext_vector_type(4)));

// "this fragment register is complete here": an empty asm that uses the four values keeps the compiler from sinking their arithmetic past this point
// (the host emulator defines it away)
#ifndef QM_PIN4
#define QM_PIN4(q) asm volatile("" :: "v"((q)[0]), "v"((q)[1]), "v"((q)[2]), "v"((q)[3]))
#endif

// Read-only parameter tables (model blob `mb`, settings `st`) are read through the CONSTANT address space: their loads become
// scalar (s_load into SGPRs, one copy per wave) instead of 64-lane vector loads into VGPR pairs.  No kernel writes these tables.
typedef const double __attribute__((address_space(4)))* qm_ctab;
#ifndef QM_TABLE_OPAQUE                 /* keeps the compiler from folding the two casts back into a plain global pointer */
#define QM_TABLE_OPAQUE(p) asm volatile("" : "+s"(p))
#endif
#ifndef QM_LANE_OPAQUE                  /* a per-lane integer the compiler must treat as unknown: a lane-dependent choice between two LDS arrays is then ONE select of the base and
                                          plain `ds_read ... offset:imm` accesses, not a select between two absolute addresses at every access (K1b: 66 accesses, 198 instructions) */
#define QM_LANE_OPAQUE(i) asm volatile("" : "+v"(i))
#endif
#ifndef QM_SCALARS_READY                /* "these wave-uniform values are in scalar registers here": an empty asm with scalar-register inputs (the host emulator defines it away) */
#define QM_SCALARS_READY(a, b, c, d) asm volatile("" :: "s"(a), "s"(b), "s"(c), "s"(d))
#endif
#ifndef QM_LOADED                       /* "this double is loaded HERE, by every lane": without it the compiler sinks a load whose only use is one arm of a select into a lane-conditional
                                          region — s_and_saveexec / s_cbranch_execz / s_or per element, three scalar instructions and a branch around one ds_read (K1b had 300 such regions) */
#define QM_LOADED(d) asm volatile("" : "+v"(d))
#endif
#ifndef QM_UNPAIRED_LDS                 /* kernel attribute: the backend's load/store optimizer does not pair this kernel's 8-byte LDS accesses into ds_read2_b64 / ds_write2_b64 (nor its
                                          global ones into wider ones).  Measured on gfx950 (tools/probes/lds_width_probe.hip, profiles/r06_lds_width_probe.log): 16 bytes per lane cost the LDS 8.0 units
                                          as two ds_read_b64 and 13.4 as one ds_read2_b64; 15.1 as two ds_write_b64, 24.3 as ds_write2_b64.  The IR-level vectorizer that forms the same pairs earlier is
                                          switched off for the whole library by the build (qm_control_amd/build_flags.py) */
#if defined(__HIP_DEVICE_COMPILE__)
#define QM_UNPAIRED_LDS __attribute__((target("no-load-store-opt")))
#else
#define QM_UNPAIRED_LDS      /* (the host pass of the same source: not a feature of the host target) */
#endif
#endif
__device__ __forceinline__ const double* qm_table(const double* p) { qm_ctab c = (qm_ctab)(p); QM_TABLE_OPAQUE(c); return (const double*)c; }

// ---- 3-vector helpers (pointer based so operands may live in LDS, registers or global) ----
__device__ __forceinline__ void v3_cross(const double* a, const double* b, double* c) {
  const double c0 = a[1] * b[2] - a[2] * b[1], c1 = a[2] * b[0] - a[0] * b[2], c2 = a[0] * b[1] - a[1] * b[0];
  c[0] = c0; c[1] = c1; c[2] = c2;
}
__device__ __forceinline__ void m3_mulv(const double* M, const double* v, double* r) {
  const double r0 = M[0] * v[0] + M[1] * v[1] + M[2] * v[2], r1 = M[3] * v[0] + M[4] * v[1] + M[5] * v[2], r2 = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
  r[0] = r0; r[1] = r1; r[2] = r2;
}
__device__ __forceinline__ void m3_mul(const double* A, const double* B, double* C) {   // C may not alias A or B
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
__device__ __forceinline__ void m3_inv(const double* A, double* R) {   // cofactor inverse
  const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[2] * A[7] - A[1] * A[8], c02 = A[1] * A[5] - A[2] * A[4];
  const double c10 = A[5] * A[6] - A[3] * A[8], c11 = A[0] * A[8] - A[2] * A[6], c12 = A[2] * A[3] - A[0] * A[5];
  const double c20 = A[3] * A[7] - A[4] * A[6], c21 = A[1] * A[6] - A[0] * A[7], c22 = A[0] * A[4] - A[1] * A[3];
  const double id = 1.0 / (A[0] * c00 + A[1] * c10 + A[2] * c20);
  R[0] = c00 * id; R[1] = c01 * id; R[2] = c02 * id; R[3] = c10 * id; R[4] = c11 * id; R[5] = c12 * id; R[6] = c20 * id; R[7] = c21 * id; R[8] = c22 * id;
}
// sin and cos of an angle together, ≈ 1 ulp: Cody–Waite reduction x = k π/2 + r, then the minimax kernels on |r| <= π/4 (the standard degree-13 / degree-14
// coefficients).  ≈ 35 instructions for the pair;
// the device library's sin() and cos() carry a full-range Payne–Hanek reduction in double-double arithmetic — ≈ 300 instructions EACH — and the thread-per-node
// kinematics kernels evaluate 36 pairs per node.  Joint and Euler angles never leave a few multiples of π: this pair reduces with two Cody–Waite constants (fdlibm's
// pio2_1 = the first 33 bits of π/2 and pio2_1t = π/2 − pio2_1) and is good for |x| < 2^31: k = rint(2x/π) < 1.4e9 fits an int32, x − k pio2_1 is EXACT under the fma (a
// multiple of 2^-36 below 1 in magnitude), and the second step rounds once, so the reduced argument carries an absolute error below 7e-17 + k 6.7e-27 < 8e-17 over the whole
// range.  Round 5: anything beyond (|x| >= 2^31 rad, infinities, NaN) yields NaN — which the solve reports as a failure (status -4) — instead of taking the library path:
// rounds 1-4 kept `sin(x); cos(x)` behind a branch for |x| >= 1e5, and that never-executed code was INLINED AT EVERY CALL SITE: 19.5 k of the 27 k instructions of
// qm_lq_kin_kernel, 20.5 k of qm_ls_eval_kernel's 39.8 k, 13.2 k of qm_wbc_kernel's 41.5 k — kernels several times the 64 KB instruction cache two CUs share, with the hot
// code scattered between the cold blocks.  Results for |x| < 1e5 are bit-identical to rounds 1-4 (same constants, same operations).
__device__ __forceinline__ void qm_sincos(double xin, double& sn, double& cs) {
  const double x = (fabs(xin) < 2147483648.0) ? xin : __builtin_nan("");
  const double k = rint(x * 6.36619772367581382433e-01);
  double r = fma(-k, 1.57079632673412561417e+00, x); r = fma(-k, 6.07710050650619224932e-11, r);
  const double z = r * r;
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06), -1.98412698298579493134e-04), 8.33333333332248946124e-03), -1.66666666666666324348e-01);
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07), 2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
  const double sr = fma(r * z, ps, r), cr = fma(z, fma(z, pc, -0.5), 1.0);
  const int q = (int)k & 3;                                   // k mod 4 (two's complement: also for negative k)
  const double ss = (q & 1) ? cr : sr, cc = (q & 1) ? sr : cr;
  sn = (q & 2) ? -ss : ss; cs = ((q + 1) & 2) ? -cc : cc;
}
// FAST selects qm_sincos (the MPC's kinematics kernels and, since round 4, the rigid-body passes of the WBC / plant kernels: − 9 k of 117 k cycles per instance; while the WBC sat at
// 437 of the 440 registers that let the next step's grid kernel run beside it the inline pair cost one allocation granule too many — at 390 it has the room); the default is the library's sin / cos
template <bool FAST = false>
__device__ __forceinline__ void rot_axis_angle(const double* a, double q, double* R) {
  double s, c; if (FAST) qm_sincos(q, s, c); else { s = sin(q); c = cos(q); } const double oc = 1.0 - c;
  R[0] = c + oc * (a[0] * a[0]);        R[1] = oc * (a[0] * a[1]) - s * a[2]; R[2] = oc * (a[0] * a[2]) + s * a[1];
  R[3] = oc * (a[1] * a[0]) + s * a[2]; R[4] = c + oc * (a[1] * a[1]);        R[5] = oc * (a[1] * a[2]) - s * a[0];
  R[6] = oc * (a[2] * a[0]) - s * a[1]; R[7] = oc * (a[2] * a[1]) + s * a[0]; R[8] = c + oc * (a[2] * a[2]);
}
// R = Rz(z) Ry(y) Rx(x); E maps zyx rates to world angular velocity
template <bool FAST = false>
__device__ __forceinline__ void rot_zyx(double z, double y, double x, double* R) {
  double sz, cz, sy, cy, sx, cx;
  if (FAST) { qm_sincos(z, sz, cz); qm_sincos(y, sy, cy); qm_sincos(x, sx, cx); } else { sz = sin(z); cz = cos(z); sy = sin(y); cy = cos(y); sx = sin(x); cx = cos(x); }
  R[0] = cz * cy; R[1] = cz * sy * sx - sz * cx; R[2] = cz * sy * cx + sz * sx;
  R[3] = sz * cy; R[4] = sz * sy * sx + cz * cx; R[5] = sz * sy * cx - cz * sx;
  R[6] = -sy;     R[7] = cy * sx;                R[8] = cy * cx;
}
template <bool FAST = false>
__device__ __forceinline__ void euler_E(double z, double y, double* E) {
  double sz, cz, sy, cy;
  if (FAST) { qm_sincos(z, sz, cz); qm_sincos(y, sy, cy); } else { sz = sin(z); cz = cos(z); sy = sin(y); cy = cos(y); }
  E[0] = 0.0; E[1] = -sz; E[2] = cy * cz; E[3] = 0.0; E[4] = cz; E[5] = cy * sz; E[6] = 1.0; E[7] = 0.0; E[8] = -sy;
}
// natural logarithm for the barrier values: frexp + atanh series in s = (m − 1)/(m + 1), |s| <= 0.172, error < 1e-15 relative.
// (≈ 30 instructions; the library's double-double log costs ≈ 200 and a one-thread-per-node kernel evaluates 52 barriers per node)
__device__ __forceinline__ double qm_log(double h) {
  int e; double m = frexp(h, &e);                                   // h = m 2^e, m in [0.5, 1)
  if (m < 0.70710678118654752) { m *= 2.0; e -= 1; }                // m in [1/sqrt 2, sqrt 2)
  const double den = m + 1.0; double r = __builtin_amdgcn_rcp(den);
  r = fma(fma(-den, r, 1.0), r, r); r = fma(fma(-den, r, 1.0), r, r);
  const double s = (m - 1.0) * r, z = s * s;
  double p = 1.0 / 19.0;
  p = fma(p, z, 1.0 / 17.0); p = fma(p, z, 1.0 / 15.0); p = fma(p, z, 1.0 / 13.0); p = fma(p, z, 1.0 / 11.0); p = fma(p, z, 1.0 / 9.0);
  p = fma(p, z, 1.0 / 7.0); p = fma(p, z, 1.0 / 5.0); p = fma(p, z, 1.0 / 3.0); p = fma(p, z, 1.0);
  return fma((double)e, 0.69314718055994531, 2.0 * s * p);
}
// relaxed log barrier (task.info:290-314 -> [upstream RelaxedBarrierPenalty]); one logarithm, no divergent branch
__device__ __forceinline__ double barrier_val(double mu, double delta, double h) {
  const bool in = h > delta; const double L = qm_log(in ? h : delta), t = (h - 2.0 * delta) * (1.0 / delta);   // 1/delta: one division per distinct delta after CSE
  return in ? -mu * L : mu * (-L + 0.5 * t * t - 0.5);
}
// 1 / x to ≈ 1 ulp: the hardware estimate (v_rcp_f64, 2^-24 on gfx950) + one third-order correction — 5 instructions against the ≈ 13 of an IEEE division
// (v_div_scale x 2, v_rcp, four fused steps, v_div_fmas, v_div_fixup); for finite, normal x (every caller divides by a sum of squares or a barrier argument)
__device__ __forceinline__ double qm_frcp(double x) { const double r = __builtin_amdgcn_rcp(x); const double e = fma(-x, r, 1.0); return fma(fma(e, e, e), r, r); }
// first and second derivative of the relaxed log barrier: one reciprocal serves both (select the argument, then divide once)
__device__ __forceinline__ void barrier_d12(double mu, double delta, double h, double& d1, double& d2) {
  const bool in = h > delta; const double inv = qm_frcp(in ? h : delta);
  d1 = in ? -mu * inv : mu * (h - 2.0 * delta) * (inv * inv);
  d2 = mu * (inv * inv);
}
__device__ __forceinline__ double barrier_d1(double mu, double delta, double h) { double a, b; barrier_d12(mu, delta, h, a, b); return a; }
__device__ __forceinline__ double barrier_d2(double mu, double delta, double h) { double a, b; barrier_d12(mu, delta, h, a, b); return b; }

// contact index (LF,RF,LH,RH; ModelSettings.h:38) of leg chain c (joint order LF,LH,RF,RH; task.info:168-188)
__device__ __forceinline__ int chain_to_contact(int c) { return (c == 1) ? 2 : (c == 2) ? 1 : c; }
__device__ __forceinline__ int contact_to_chain(int i) { return (i == 1) ? 2 : (i == 2) ? 1 : i; }
__device__ __forceinline__ bool mode_flag(int mode, int contact) { return (mode >> (3 - contact)) & 1; }   // 8*LF+4*RF+2*LH+RH

// wave-level ordering point for LDS traffic inside ONE wavefront (no s_barrier: a wave's DS ops retire in order)
__device__ __forceinline__ void qm_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// One-thread-per-row kernels: 64 consecutive rows of W doubles, row r of the block <-> lane r.  The global side is accessed
// coalesced (lanes run over consecutive elements) and transposed through LDS [64][W+1]; per-lane arrays keep static indices.
#define QM_ROWS_LDS(W) (64 * ((W) + 1))
template <int W, class F> __device__ __forceinline__ void qm_rows_gather(double* lds, size_t row0, size_t nrows, int l, double* out, F elem) {
#pragma unroll
  for (int t = 0; t < W; ++t) { const int e = t * 64 + l; const int r = e / W, c = e - r * W; lds[r * (W + 1) + c] = (row0 + r < nrows) ? elem(row0 + r, c) : 0.0; }
  qm_wave_sync();
#pragma unroll
  for (int q = 0; q < W; ++q) out[q] = lds[l * (W + 1) + q];
  qm_wave_sync();
}
// rows whose bit in `rowmask` is clear are not written
template <int W> __device__ __forceinline__ void qm_rows_scatter(double* lds, double* dst, size_t stride, size_t row0, unsigned long long rowmask, int l, const double* v) {
#pragma unroll
  for (int q = 0; q < W; ++q) lds[l * (W + 1) + q] = v[q];
  qm_wave_sync();
#pragma unroll
  for (int t = 0; t < W; ++t) { const int e = t * 64 + l; const int r = e / W, c = e - r * W; if ((rowmask >> r) & 1ull) dst[(row0 + r) * stride + c] = lds[r * (W + 1) + c]; }
  qm_wave_sync();
}

// value of lane `src` (wave-uniform index) in every lane: two v_readlane_b32, the result lives in SGPRs
__device__ __forceinline__ double qm_bcast(double v, int src) {
  union { double d; int i[2]; } u; u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], src); u.i[1] = __builtin_amdgcn_readlane(u.i[1], src);
  return u.d;
}

// one DPP step on a double: src taken through the dpp control (row_shr:n = 0x110 + n, row_bcast:15 = 0x142, row_bcast:31 = 0x143);
// lanes outside the row mask or shifted in from outside the row keep `old`
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double qm_dpp(double old, double src) {
  union { double d; int i[2]; } o, v; o.d = old; v.d = src;
  v.i[0] = __builtin_amdgcn_update_dpp(o.i[0], v.i[0], CTRL, ROW_MASK, 0xf, false);
  v.i[1] = __builtin_amdgcn_update_dpp(o.i[1], v.i[1], CTRL, ROW_MASK, 0xf, false);
  return v.d;
}
// row_shr step that shifts ZEROS in (bound_ctrl): no `old` operand, hence no zero initialisation of the destination pair in front of every step
template <int CTRL>
__device__ __forceinline__ double qm_dpp0(double src) {
  union { double d; int i[2]; } v; v.d = src;
  v.i[0] = __builtin_amdgcn_mov_dpp(v.i[0], CTRL, 0xf, 0xf, true);
  v.i[1] = __builtin_amdgcn_mov_dpp(v.i[1], CTRL, 0xf, 0xf, true);
  return v.d;
}
// wave-wide sum / max, result in every lane: four row_shr steps inside the 16-lane rows, two row broadcasts, one v_readlane
// (no LDS crossbar traffic: ds_bpermute-based butterflies cost several times more on a lone wave)
__device__ __forceinline__ double qm_wave_sum(double v) {
  v += qm_dpp0<0x111>(v); v += qm_dpp0<0x112>(v); v += qm_dpp0<0x114>(v); v += qm_dpp0<0x118>(v);
  v += qm_dpp<0x142, 0xa>(0.0, v); v += qm_dpp<0x143, 0xc>(0.0, v);
  return qm_bcast(v, 63);
}
__device__ __forceinline__ double qm_wave_max(double v) {
  v = fmax(v, qm_dpp<0x111, 0xf>(v, v)); v = fmax(v, qm_dpp<0x112, 0xf>(v, v)); v = fmax(v, qm_dpp<0x114, 0xf>(v, v)); v = fmax(v, qm_dpp<0x118, 0xf>(v, v));
  v = fmax(v, qm_dpp<0x142, 0xa>(v, v)); v = fmax(v, qm_dpp<0x143, 0xc>(v, v));
  return qm_bcast(v, 63);
}

// asynchronous global -> LDS copy of 16 bytes per lane (global_load_lds_dwordx4): lane l's 16 bytes land at lds_chunk + 16 l, no VGPR
// is touched; completion is tracked by vmcnt (qm_dma_wait), the data becomes visible to ds_read after that
// (cache policy 2 = nt, non-temporal: the copies stream stage records that are read once per sweep — measured K3 1.12 -> 1.05 ms against the default policy;
// the scope bits sc0 / sc1 on top of it change nothing)
__device__ __forceinline__ void qm_dma16(const double* g, double* lds_chunk) { __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)lds_chunk, 16, 0, 2); }
// The same with the instruction's IMMEDIATE OFFSET (bytes, 0 .. 4095): the hardware adds it to BOTH sides — global address and LDS address — so consecutive 1 KB chunks of a
// segment that is contiguous on both sides share one global base and one LDS base (M0): offsets 0, 1024, 2048, 3072.  The LDS side is an address-space-3 pointer, formed
// ONCE from the kernel's shared array: passing a generic pointer per chunk makes the compiler rebuild M0 through a generic -> LDS cast with a null check every time
// (s_add_u32 / s_addc_u32 / s_cmp_lg_u64 / s_cselect: K3 issued ≈ 6 scalar instructions per 1 KB chunk, 35 chunks per stage, on a lone wave that pays for every one of them)
typedef __attribute__((address_space(3))) char* qm_lds_ptr;
__device__ __forceinline__ qm_lds_ptr qm_lds(double* p) { return (qm_lds_ptr)p; }
template <int OFF> __device__ __forceinline__ void qm_dma16_at(const char* g, qm_lds_ptr lds_chunk) { static_assert(OFF >= 0 && OFF < 4096, "13-bit signed immediate"); __builtin_amdgcn_global_load_lds(g, lds_chunk, 16, OFF, 2); }
__device__ __forceinline__ void qm_dma_wait() { __builtin_amdgcn_s_waitcnt(0x0F70); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }      // vmcnt(0)
__device__ __forceinline__ void qm_lds_drain() { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }     // lgkmcnt(0): every ds_read has returned

// strided view of a per-instance array living in a lane-interleaved HBM workspace: element i of instance b sits at
// base[i * stride + b], so the 64 lanes of a wave (consecutive instances) touch consecutive addresses
struct QmSPtr {
  double* p; int s;
  __device__ __forceinline__ double& operator[](int i) const { return p[(size_t)i * s]; }
  __device__ __forceinline__ QmSPtr operator+(int o) const { QmSPtr r; r.p = p + (size_t)o * s; r.s = s; return r; }
};

// ---- LDS tile helpers (all threads of the workgroup participate; caller places the barriers) ----
__device__ __forceinline__ void tile_zero(double* T, int ndoubles = QM_TILE) { for (int i = threadIdx.x; i < ndoubles; i += blockDim.x) T[i] = 0.0; }
// copy a rows x cols row-major global matrix (leading dim sld) into a tile
__device__ __forceinline__ void tile_load(double* T, const double* src, int rows, int cols, int sld) {
  for (int idx = threadIdx.x; idx < rows * cols; idx += blockDim.x) { const int r = idx / cols, c = idx - r * cols; T[r * QM_LD + c] = src[r * sld + c]; }
}
__device__ __forceinline__ void tile_store(const double* T, double* dst, int rows, int cols, int dld) {
  for (int idx = threadIdx.x; idx < rows * cols; idx += blockDim.x) { const int r = idx / cols, c = idx - r * cols; dst[r * dld + c] = T[r * QM_LD + c]; }
}

// C = op(A) op(B) over k-slabs [ks0, ks1) of 4; op(A) is (16 mt) x K, op(B) is K x (16 nt).
// TA: A holds Aᵀ (A[k][i]); TB: B holds Bᵀ (B[j][k]).  epi(row, col, value) consumes every output element.
template <bool TA, bool TB, class Epi>
__device__ __forceinline__ void wg_gemm(const double* A, const double* B, int mt, int nt, int ks0, int ks1, Epi epi) {
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int li = l & 15, lk = l >> 4;
  for (int t = wave; t < mt * nt; t += nw) {
    const int I = t / nt, J = t - I * nt;
    qm_d4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int kk = ks0; kk < ks1; ++kk) {
      const double a = TA ? A[(4 * kk + lk) * QM_LD + 16 * I + li] : A[(16 * I + li) * QM_LD + 4 * kk + lk];
      const double b = TB ? B[(16 * J + li) * QM_LD + 4 * kk + lk] : B[(4 * kk + lk) * QM_LD + 16 * J + li];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) epi(16 * I + lk + 4 * r, 16 * J + li, acc[r]);
  }
}

// y(rows) = M(rows x cols tile) x   — one thread per row
__device__ __forceinline__ double tile_row_dot(const double* T, int row, const double* x, int cols) {
  double s = 0.0; for (int c = 0; c < cols; ++c) s += T[row * QM_LD + c] * x[c]; return s;
}
__device__ __forceinline__ double tile_col_dot(const double* T, int col, const double* x, int rows) {
  double s = 0.0; for (int r = 0; r < rows; ++r) s += T[r * QM_LD + col] * x[r]; return s;
}

// ---- register fragments of the f64 matrix core (one wavefront) ----
// v_mfma_f64_16x16x4_f64: A[i = l&15][k = l>>4], B[k = l>>4][j = l&15], D[row = (l>>4) + 4r][col = l&15].  A matrix kept as
// D-fragments ("D-layout": tile (I,J), register r <-> element (16I + (l>>4) + 4r, 16J + (l&15))) is directly the B operand of
// k-step kk = 4K + r and — read as an A operand — supplies its TRANSPOSE, so products of the form P = Zᵀ Y chain from MFMA to
// MFMA with no layout conversion.
// P += (neg ? −1 : 1) · Zᵀ Y over k-steps [k0, k1); Z: [KT][IT] tiles, Y: [KT][JT] tiles, P: [IT][JT] tiles
template <int KT, int IT, int JT>
__device__ __forceinline__ void qm_gemm_tn(const qm_d4 (&Z)[KT][IT], const qm_d4 (&Y)[KT][JT], qm_d4 (&P)[IT][JT], int k0, int k1, bool neg) {
#pragma unroll
  for (int kk = 0; kk < 4 * KT; ++kk) if (kk >= k0 && kk < k1) {
#pragma unroll
    for (int I = 0; I < IT; ++I) {
      const double av = neg ? -Z[kk >> 2][I][kk & 3] : Z[kk >> 2][I][kk & 3];
#pragma unroll
      for (int J = 0; J < JT; ++J) P[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Y[kk >> 2][J][kk & 3], P[I][J], 0, 0, 0);
    }
  }
}
template <int IT, int JT>
__device__ __forceinline__ void qm_frag_zero(qm_d4 (&T)[IT][JT]) {
#pragma unroll
  for (int I = 0; I < IT; ++I)
#pragma unroll
    for (int J = 0; J < JT; ++J) T[I][J] = qm_d4{0.0, 0.0, 0.0, 0.0};
}
// D-layout read of a rows x cols row-major matrix (global or LDS, leading dim ld); TR: the source holds the transpose
template <int IT, int JT, bool TR>
__device__ __forceinline__ void qm_frag_load(qm_d4 (&T)[IT][JT], const double* src, int ld, int rows, int cols) {
  const int g = (threadIdx.x & 63) >> 4, c = threadIdx.x & 15;
#pragma unroll
  for (int I = 0; I < IT; ++I)
#pragma unroll
    for (int J = 0; J < JT; ++J)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int row = 16 * I + g + 4 * r, col = 16 * J + c; T[I][J][r] = (row < rows && col < cols) ? (TR ? src[col * ld + row] : src[row * ld + col]) : 0.0; }
}
// the same for a source that is a fully populated, zero-padded 32 x ld LDS tile: no bounds, hence no exec-mask juggling per element
template <int IT, int JT, bool TR>
__device__ __forceinline__ void qm_frag_load_tile(qm_d4 (&T)[IT][JT], const double* src, int ld) {
  const int g = (threadIdx.x & 63) >> 4, c = threadIdx.x & 15;
#pragma unroll
  for (int I = 0; I < IT; ++I)
#pragma unroll
    for (int J = 0; J < JT; ++J)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int row = 16 * I + g + 4 * r, col = 16 * J + c; T[I][J][r] = TR ? src[col * ld + row] : src[row * ld + col]; }
}
// STREAMING data — written once by one kernel, read once by a later one, with gigabytes of other traffic in between (stage records K1b -> K3: 2.8 GB per launch) —
// is stored with the non-temporal hint (global_store ... nt): the lines do not stay in L2 / MALL at the expense of what IS re-read (measured: K3 − 7 %, step − 3 %)
#ifdef QM_NO_NT_STORES      /* experiment only */
#define QM_STREAM_ST(p, v) (*(p) = (double)(v))
#else
#define QM_STREAM_ST(p, v) __builtin_nontemporal_store((double)(v), (p))
#endif
template <int IT, int JT, bool STREAM = false>
__device__ __forceinline__ void qm_frag_store(const qm_d4 (&T)[IT][JT], double* dst, int ld, int rows, int cols) {
  const int g = (threadIdx.x & 63) >> 4, c = threadIdx.x & 15;
#pragma unroll
  for (int I = 0; I < IT; ++I)
#pragma unroll
    for (int J = 0; J < JT; ++J)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int row = 16 * I + g + 4 * r, col = 16 * J + c; if (row < rows && col < cols) { if (STREAM) QM_STREAM_ST(dst + row * ld + col, T[I][J][r]); else dst[row * ld + col] = T[I][J][r]; } }
}

// parameter of the multiple-shooting transcription / filter line search for the selected solver: the `ipm` block's value with ST_SOLVER == 2, the `sqp` block's otherwise
// (the iLQR shares the SQP's grid).  sqp_slot is one of ST_SQP_DT, ST_SQP_ITER, ST_DELTA_TOL, ST_G_MAX, ST_G_MIN — contiguous, in the order of the ipm slots
static_assert(ST_SQP_ITER - ST_SQP_DT == ST_IPM_ITER - ST_IPM_DT && ST_DELTA_TOL - ST_SQP_DT == ST_IPM_DELTA_TOL - ST_IPM_DT && ST_G_MAX - ST_SQP_DT == ST_IPM_G_MAX - ST_IPM_DT &&
              ST_G_MIN - ST_SQP_DT == ST_IPM_G_MIN - ST_IPM_DT, "qm_ms_param maps sqp slots onto ipm slots by offset: the two blocks of include/qmhip_layout.h must keep the same order");
__host__ __device__ __forceinline__ double qm_ms_param(const double* st, int sqp_slot) { return (st[ST_SOLVER] >= 2.0) ? st[ST_IPM_DT + (sqp_slot - ST_SQP_DT)] : st[sqp_slot]; }      // slot 2 (the SQP step on the `ipm` block) and slot 3 (the interior-point method, k_ipm.h)

// ---- per-node stage record written by K1 (LQ + projection) and read by K3 (Riccati); doubles ----
// dimensions: nx = 30, projected input dim m <= 18 (stance 18, trot 16); row-major, fixed strides
#define QM_MMAX 18
// Round 5: COMPACT record — only what is written is laid out (rounds 1-4 reserved full 30-row blocks, an unused Rp and a 540-double gain slot: 58.9 KB per node; now 30 KB):
//   [0, 1094)      what K1b writes row-major and K3's sweeps fetch: Ap rows 0..11 | Bp rows 0..11 | Px rows 12..23 | bp qp rp Pe | k | swing blocks, mode, dt | m, cp
//   [1152, 1692)   the gain K (written by K3's backward sweep, read by its rollout)
//   [1792, 3840)   K3's backward operands in fragment order
// The terminal record's Q_N (30 x 30) overlays the head: a terminal node has no dynamics rows.  Debug-only data (the null-space basis Pu) lives in the debug record (k_lq.h).
#define SR_AP   0                      /* [12][30]  rows 0..11 of A + B Px (joint rows: e_j + dt Px[j], rebuilt by K3) */
#define SR_QP   0                      /* [30][30]  TERMINAL record only (intermediate nodes: SR_FRAG); overlays SR_AP / SR_BP / the Px rows */
#define SR_BP   360                    /* [12][18]  rows 0..11 of B Pu (joint rows: dt Pu[j]) */
#define SR_PX   216                    /* VIRTUAL base of the [30][30] matrix Px (du = Pe + Px dx + Pu ut): only its rows 12..23 exist, at SR_PX + 360 = 576 ... 935 */
#define SR_BPV  936                    /* [30]      b + B Pe            */
#define SR_QPV  966                    /* [30]                          */
#define SR_RPV  996                    /* [18]                          */
#define SR_PE   1014                   /* [30]                          */
#define SR_KFF  1044                   /* [18]  the offset k = −L⁻ᵀ y (written by K3) */
#define SR_SWG   1064                  /* [4][6]    per contact: the 3x2 null-space block of a swing leg's joint velocities (columns of Pu) */
#define SR_MODEF 1088                  /* contact mode of the interval (as double), dt, and the constants 1.0, 0.0: Pu = {identity columns, SR_SWG blocks} is rebuilt from it */
#define SR_SCAL 1092                   /* [0]=m (as double) [1]=cp      */
#define SR_K    1096                   /* [32] profiling stamps of the instrumented instances only */
#define SR_PP   1152                   /* [18][30]  the gain K = −L⁻ᵀ W, written by K3 (Pp itself travels in SR_FRAG) */
static_assert(SR_PX + 360 == SR_BP + 216 && SR_PX + 720 == SR_BPV && SR_MODEF == SR_SWG + 24 && SR_SCAL >= SR_MODEF + 4 && SR_K >= SR_SCAL + 2 && SR_PP >= SR_K + 32 && SR_QP + 900 <= SR_BPV, "stage record fields");
/* K3's backward operands [Qp | qp], [Pp | rp], Rp in FRAGMENT order: register r of tile t is one contiguous 512-byte row, element (t, r, lane l) at (4 t + r) 64 + l, i.e.
   matrix entry (16 I + (l >> 4) + 4 r, 16 J + (l & 15)) of tile (I, J).  K1b stores its fragments as they are (one unconditional 512-byte store per register, padding
   rows / columns of Pp, Rp zeroed by a select) and K3 reads them back with one LDS load per register: no masks, no per-element addresses on either side.
   The vectors ride in column 30.  m <= 16 (one tile row of reduced inputs) reads the first 1536 doubles only. */
#define SR_FRAG   1792
static_assert(SR_FRAG >= SR_PP + 540 && SR_FRAG % 128 == 0, "fragment region behind the gain, on a 1 KB boundary");
#define SR_F_QP    0                    /* tiles (0,0), (0,1), (1,1) of [Qp | qp]: 3 x 256 (the lower-left tile is the mirror image, never formed) */
#define SR_F_PP    768                  /* tile row 0 of [Pp | rp]: tiles (0,0), (0,1)                                                              */
#define SR_F_RP    1280                 /* tile (0,0) of Rp                                                                                           */
#define SR_F_PP1   1536                 /* m > 16: register 0 (rows 16..19) of tiles (1,0), (1,1) of [Pp | rp]                                        */
#define SR_F_RP01  1664                 /* m > 16: tile (0,1) of Rp                                                                                   */
#define SR_F_RP11  1920                 /* m > 16: register 0 of tile (1,1) of Rp                                                                     */
#define SR_F_SIZE  2048
#define SR_SIZE (SR_FRAG + SR_F_SIZE)

// per-node performance terms: cost, dynamics defect SSE (dt weighted), equality SSE (dt weighted)
#define PF_SIZE 4
