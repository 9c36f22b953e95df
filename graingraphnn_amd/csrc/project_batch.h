// Launch descriptor shared by the two projection kernels (project.hip: native fp32 MFMA;
// project_x6.hip: exact bf16 split).
#pragma once
#include "common.h"

namespace ggnn {

constexpr int PJ_MAX_PROBLEMS = 4;
// Up to four projections in one launch (the node types of a cell and / or both models): every
// workgroup belongs to one problem and is one (96-column tile, row split) of it.
struct ProjectBatch {
  ggnn_project_args a[PJ_MAX_PROBLEMS];
  int wg_off[PJ_MAX_PROBLEMS + 1];  // first workgroup of every problem
  int m_splits[PJ_MAX_PROBLEMS];
  int n;
};

}  // namespace ggnn
