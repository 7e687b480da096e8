// tools/probes/lds_width_probe.hip — what 16 bytes per lane cost the LDS as two ds_read_b64, one ds_read2_b64 or one ds_read_b128 (and the stores likewise), measured:
// every wave of a full machine (ten 64-thread workgroups per CU, K1b's residency) issues the same number of bytes per lane in a loop of conflict-free accesses.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/probes/lds_width_probe.hip -o /tmp/lds_probe && /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4096
template <int MODE>
__global__ void __launch_bounds__(64) probe(double* out) {
  extern __shared__ double S[];                    // 16 KB per workgroup
  const int l = threadIdx.x;
  for (int i = l; i < 2048; i += 64) S[i] = (double)i;
  __syncthreads();
  double acc = 0.0;
  const unsigned a8 = (unsigned)(size_t)(__attribute__((address_space(3))) double*)S + l * 8, a16 = (unsigned)(size_t)(__attribute__((address_space(3))) double*)S + l * 16;
  for (int it = 0; it < ITERS; ++it) {
    double v0, v1, v2, v3, v4, v5, v6, v7;
    if (MODE == 0) {          // 8 x ds_read_b64, lane stride 8 B, rows 512 B apart
      asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:512\n ds_read_b64 %2, %8 offset:1024\n ds_read_b64 %3, %8 offset:1536\n"
                   "ds_read_b64 %4, %8 offset:2048\n ds_read_b64 %5, %8 offset:2560\n ds_read_b64 %6, %8 offset:3072\n ds_read_b64 %7, %8 offset:3584\n s_waitcnt lgkmcnt(0)"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(a8) : "memory");
    } else if (MODE == 1) {   // 4 x ds_read2_b64: the same eight doubles
      typedef double d2 __attribute__((ext_vector_type(2))); d2 p0, p1, p2, p3;
      asm volatile("ds_read2_b64 %0, %4 offset0:0 offset1:64\n ds_read2_b64 %1, %4 offset0:128 offset1:192\n s_waitcnt lgkmcnt(0)\n"
                   : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3) : "v"(a8) : "memory");
      asm volatile("ds_read2_b64 %0, %2 offset0:0 offset1:64\n ds_read2_b64 %1, %2 offset0:128 offset1:192\n s_waitcnt lgkmcnt(0)\n"
                   : "=&v"(p2), "=&v"(p3) : "v"(a8 + 2048) : "memory");
      v0 = p0.x; v1 = p0.y; v2 = p1.x; v3 = p1.y; v4 = p2.x; v5 = p2.y; v6 = p3.x; v7 = p3.y;
    } else if (MODE == 2) {   // 4 x ds_read_b128, lane stride 16 B, rows 1 KB apart
      typedef double d2 __attribute__((ext_vector_type(2))); d2 p0, p1, p2, p3;
      asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)\n"
                   : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3) : "v"(a16) : "memory");
      v0 = p0.x; v1 = p0.y; v2 = p1.x; v3 = p1.y; v4 = p2.x; v5 = p2.y; v6 = p3.x; v7 = p3.y;
    } else if (MODE == 3) {   // 4 x ds_read2_b64 with ADJACENT doubles (what the compiler makes of two neighbouring 8-byte loads), lane stride 16 B
      typedef double d2 __attribute__((ext_vector_type(2))); d2 p0, p1, p2, p3;
      asm volatile("ds_read2_b64 %0, %4 offset0:0 offset1:1\n ds_read2_b64 %1, %4 offset0:128 offset1:129\n s_waitcnt lgkmcnt(0)\n"
                   : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3) : "v"(a16) : "memory");
      asm volatile("ds_read2_b64 %0, %2 offset0:0 offset1:1\n ds_read2_b64 %1, %2 offset0:128 offset1:129\n s_waitcnt lgkmcnt(0)\n"
                   : "=&v"(p2), "=&v"(p3) : "v"(a16 + 2048) : "memory");
      v0 = p0.x; v1 = p0.y; v2 = p1.x; v3 = p1.y; v4 = p2.x; v5 = p2.y; v6 = p3.x; v7 = p3.y;
    } else if (MODE == 4) {   // 8 x ds_write_b64
      v0 = v1 = v2 = v3 = v4 = v5 = v6 = v7 = acc;
      asm volatile("ds_write_b64 %1, %0\n ds_write_b64 %1, %0 offset:512\n ds_write_b64 %1, %0 offset:1024\n ds_write_b64 %1, %0 offset:1536\n"
                   "ds_write_b64 %1, %0 offset:2048\n ds_write_b64 %1, %0 offset:2560\n ds_write_b64 %1, %0 offset:3072\n ds_write_b64 %1, %0 offset:3584\n s_waitcnt lgkmcnt(0)" :: "v"(v0), "v"(a8) : "memory");
    } else if (MODE == 5) {   // 4 x ds_write2_b64
      v0 = v1 = v2 = v3 = v4 = v5 = v6 = v7 = acc;
      asm volatile("ds_write2_b64 %1, %0, %0 offset0:0 offset1:64\n ds_write2_b64 %1, %0, %0 offset0:128 offset1:192\n s_waitcnt lgkmcnt(0)\n" :: "v"(v0), "v"(a8) : "memory");
      asm volatile("ds_write2_b64 %1, %0, %0 offset0:0 offset1:64\n ds_write2_b64 %1, %0, %0 offset0:128 offset1:192\n s_waitcnt lgkmcnt(0)\n" :: "v"(v0), "v"(a8 + 2048) : "memory");
    } else {                  // 4 x ds_write_b128
      typedef double d2 __attribute__((ext_vector_type(2))); d2 p = {acc, acc}; v0 = v1 = v2 = v3 = v4 = v5 = v6 = v7 = acc;
      asm volatile("ds_write_b128 %1, %0\n ds_write_b128 %1, %0 offset:1024\n ds_write_b128 %1, %0 offset:2048\n ds_write_b128 %1, %0 offset:3072\n s_waitcnt lgkmcnt(0)\n" :: "v"(p), "v"(a16) : "memory");
    }
    acc += ((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7));
  }
  out[blockIdx.x * 64 + l] = acc;
}
template <int MODE> static void run(const char* name, double* out, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE><<<blocks, 64, 16384>>>(out); hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < 5; ++r) { hipEventRecord(e0); probe<MODE><<<blocks, 64, 16384>>>(out); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
  // per CU: blocks / 256 waves, each ITERS x 64 lanes x 64 B
  const double bytes_per_cu = (double)blocks / 256.0 * ITERS * 64.0 * 64.0, clk = 2.4e9;
  printf("%-44s %8.3f ms  %7.1f B/clk/CU (at 2.4 GHz)  %6.2f LDS cycles per 16 B/lane wave access\n", name, best, bytes_per_cu / (best * 1e-3 * clk), (best * 1e-3 * clk) / ((double)blocks / 256.0 * ITERS * 4.0));
}
int main() {
  const int blocks = 256 * 10; double* out; hipMalloc(&out, (size_t)blocks * 64 * 8);
  run<0>("2 x ds_read_b64 per 16 B", out, blocks); run<1>("ds_read2_b64 (rows 512 B apart)", out, blocks); run<3>("ds_read2_b64 (adjacent doubles)", out, blocks); run<2>("ds_read_b128", out, blocks);
  run<4>("2 x ds_write_b64 per 16 B", out, blocks); run<5>("ds_write2_b64", out, blocks); run<6>("ds_write_b128", out, blocks);
  return 0;
}
