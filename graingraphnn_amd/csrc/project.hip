// Node-level projection GEMM on the fp32 matrix cores (v_mfma_f32_16x16x4_f32, exact fp32):
//   out[M, ncols] = [X[:, :F] | H] . Wp^T + bias
// One launch per node type and cell produces, for every gate and edge type, the per-node
// key/value (as source), query (as destination) and summed-skip pre-activations that the
// reference computes per EDGE (periodGATconv.py:216-218, :186).  K = F + 96 <= 108 is so
// short that a workgroup keeps its whole 64-node x K input tile and 96-column x K weight
// tile in LDS (70 KB -> two workgroups per CU) and runs one barrier-free MFMA sweep.
//
// Orientation: the WEIGHT tile is the MFMA A operand and the NODE tile the B operand, so
// that each lane ends up with 4 consecutive output columns of one node -> 16-byte stores.
//   A[i][k] = Wp[n0+i][k]   lane l holds i = l&15, k = k0 + (l>>4)
//   B[k][j] = XH[m0+j][k]   lane l holds j = l&15, k = k0 + (l>>4)
//   D[i][j] -> lane l, reg r: i = 4*(l>>4) + r, j = l&15
// LDS rows have stride Kp + 2 floats (= 2 * odd), which makes the ds_read_b32 operand
// reads bank-conflict-free (bank = (row*ld + k) mod 32 covers 0..31 over 16 rows x 2 k).
#include "common.h"

namespace ggnn {

constexpr int PJ_BM = 64;       // nodes per workgroup
constexpr int PJ_BN = 96;       // output columns per workgroup
constexpr int PJ_KP_MAX = 108;  // roundup4(F <= 12) + 96
constexpr int PJ_LD_MAX = PJ_KP_MAX + 2;

__global__ __launch_bounds__(256, 2) void project_kernel(
    const float* __restrict__ X, int64_t ldx, int F, int Fp, const float* __restrict__ H,
    int64_t ldh, int K2, const float* __restrict__ Wp, const float* __restrict__ bias, int64_t M,
    int ncols, float* __restrict__ out, int64_t ldo) {
  __shared__ float s_x[PJ_BM * PJ_LD_MAX];
  __shared__ float s_w[PJ_BN * PJ_LD_MAX];

  const int Kp = Fp + K2;
  const int ld = Kp + 2;
  const int tid = threadIdx.x;
  const int nb_n = ncols / PJ_BN;
  const int bn = blockIdx.x % nb_n;
  const int64_t bm = blockIdx.x / nb_n;
  const int n0 = bn * PJ_BN;
  const int64_t m0 = bm * PJ_BM;

  // ---- stage the node tile: feature columns (scalar, F is 8 or 11) then hidden (16 B) ----
  for (int idx = tid; idx < PJ_BM * Fp; idx += 256) {
    const int r = idx / Fp, k = idx - r * Fp;
    const int64_t m = m0 + r;
    s_x[r * ld + k] = (m < M && k < F) ? X[m * ldx + k] : 0.0f;
  }
  if (K2 > 0) {
    const int nv = K2 >> 2;
    for (int idx = tid; idx < PJ_BM * nv; idx += 256) {
      const int r = idx / nv, c4 = idx - r * nv;
      const int64_t m = m0 + r;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (m < M) v = *reinterpret_cast<const f32x4*>(H + m * ldh + 4 * c4);
      float2* dst = reinterpret_cast<float2*>(&s_x[r * ld + Fp + 4 * c4]);
      dst[0] = make_float2(v.x, v.y);
      dst[1] = make_float2(v.z, v.w);
    }
  }
  // ---- stage the weight tile (rows are Kp floats, 16-byte aligned, zero padded) ----
  {
    const int nv = Kp >> 2;
    for (int idx = tid; idx < PJ_BN * nv; idx += 256) {
      const int r = idx / nv, c4 = idx - r * nv;
      const f32x4 v = *reinterpret_cast<const f32x4*>(Wp + (int64_t)(n0 + r) * Kp + 4 * c4);
      float2* dst = reinterpret_cast<float2*>(&s_w[r * ld + 4 * c4]);
      dst[0] = make_float2(v.x, v.y);
      dst[1] = make_float2(v.z, v.w);
    }
  }
  __syncthreads();

  // ---- MFMA sweep: wave (wm, wn) owns 32 nodes x 48 columns = 2 x 3 tiles of 16x16 ----
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int lr = lane & 15, lq = lane >> 4;
  const float* pw = &s_w[(wn * 48 + lr) * ld + lq];
  const float* px = &s_x[(wm * 32 + lr) * ld + lq];
  f32x4 acc[3][2];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll 2
  for (int k0 = 0; k0 < Kp; k0 += 4) {
    float wf[3], xf[2];
#pragma unroll
    for (int a = 0; a < 3; ++a) wf[a] = pw[a * 16 * ld + k0];
#pragma unroll
    for (int b = 0; b < 2; ++b) xf[b] = px[b * 16 * ld + k0];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[a], xf[b], acc[a][b], 0, 0, 0);
  }

  // ---- epilogue: + bias, 16-byte stores (4 consecutive columns of one node per lane) ----
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int n = n0 + wn * 48 + a * 16 + 4 * lq;
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int64_t m = m0 + wm * 32 + b * 16 + lr;
      if (m < M) *reinterpret_cast<f32x4*>(out + m * ldo + n) = acc[a][b] + bv;
    }
  }
}

}  // namespace ggnn

extern "C" int ggnn_project(const float* X, int64_t ldx, int F, const float* H, int64_t ldh,
                            int k2, const float* Wp, const float* bias, int64_t M, int ncols,
                            float* out, int64_t ldo, ggnn_stream_t stream) {
  using namespace ggnn;
  if (!X || !Wp || !bias || !out || M <= 0) return GGNN_EINVAL;
  if (F < 1 || F > 12 || ldx < F) return GGNN_EINVAL;
  if (k2 != 0 && k2 != C) return GGNN_EINVAL;
  if (k2 != 0 && (!H || ldh < k2 || (ldh & 3) || !aligned16(H))) return GGNN_EINVAL;
  if (ncols <= 0 || ncols % PJ_BN != 0 || ldo < ncols || (ldo & 3)) return GGNN_EINVAL;
  if (!aligned16(Wp) || !aligned16(bias) || !aligned16(out)) return GGNN_EINVAL;
  const int Fp = (F + 3) & ~3;
  const int64_t nblk = (int64_t)(ncols / PJ_BN) * ((M + PJ_BM - 1) / PJ_BM);
  if (nblk >= INT32_MAX) return GGNN_EINVAL;
  hipLaunchKernelGGL(project_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, X,
                     ldx, F, Fp, H, ldh, k2, Wp, bias, M, ncols, out, ldo);
  return launch_status();
}
