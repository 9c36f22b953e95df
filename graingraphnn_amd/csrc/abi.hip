// Version / error strings / workspace sizing of libggnn.
#include "common.h"

#include <cstdlib>
#include <cstring>

extern "C" int ggnn_version(void) { return GGNN_ABI_VERSION; }

int ggnn::gemm_mode() {
  static const int mode = [] {
    const char* e = std::getenv("GGNN_GEMM");
    return (e && std::strcmp(e, "fp32") == 0) ? GGNN_GEMM_FP32 : GGNN_GEMM_BF16X6;
  }();
  return mode;
}
extern "C" int ggnn_gemm_mode(void) { return ggnn::gemm_mode(); }

int ggnn::num_cu() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;  // MI355X
    return v;
  }();
  return n;
}

extern "C" const char* ggnn_error_string(int code) {
  switch (code) {
    case GGNN_OK: return "ok";
    case GGNN_EINVAL: return "invalid argument (null pointer, size, alignment or unsupported width)";
    case GGNN_ELAUNCH: return "HIP launch / memset failed";
    case GGNN_ETOPOLOGY: return "the edge lists are not a valid grain graph (ggnn_topology_args.error)";
    default: return "unknown ggnn error code";
  }
}

// One model forward (decoder sizing, 4 gates): projections [n_joint, 28*96] + [n_grain, 16*96],
// aggregates [n_joint, 4*196] + [n_grain, 4*100], two (h, c) pairs per node type, head scratch.
extern "C" size_t ggnn_workspace_bytes(int64_t n_grain, int64_t n_joint, int64_t E) {
  if (n_grain < 0 || n_joint < 0 || E < 0) return 0;
  const size_t C = GGNN_C;
  size_t floats = 0;
  floats += (size_t)n_joint * 28 * C + (size_t)n_grain * 16 * C;
  floats += (size_t)n_joint * 4 * 196 + (size_t)n_grain * 4 * 100;
  floats += 4 * ((size_t)n_joint + (size_t)n_grain) * C;
  floats += (size_t)n_joint * 8 + 3 * (size_t)E;
  return floats * sizeof(float);
}
