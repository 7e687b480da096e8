// k_wbc.h — K6/K7: whole-body controller (rigid-body quantities, task assembly, hierarchical LSI, torque map).
#pragma once
#include "qm_dev_kin.h"

struct QmWbcArgs {
  const double* mb; const double* st;
  int B;
  const double* x_des; const double* u_des;   // [B][30]
  const double* rbd;                          // [B][55]
  const int* mode;                            // [B]
  const double* time;                         // [B]
  double period; int variant;
  double* input_last;                         // [B][30] state (WbcBase.cpp:212-213)
  double* out;                                // [B][54]
  int* qp_status;                             // [B][3]
  double* scratch;                            // [B][WBC_SCRATCH] global workspace
  double* dbg;                                // optional [B][WBC_DBG_SIZE]
};
#define WBC_SCRATCH 8
#define WBC_DBG_SIZE 8
#define WBC_LDS_BYTES 0
#define WBC_BLOCK 64

__global__ void qm_wbc_kernel(QmWbcArgs a) {
  const int b = blockIdx.x;
  if (threadIdx.x < 3) a.qp_status[b * 3 + threadIdx.x] = -9;   // placeholder until the WBC kernels land
}
