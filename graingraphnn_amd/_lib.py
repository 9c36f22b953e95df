"""ctypes binding of libggnn.so (the C ABI declared in include/ggnn.h).

There is no CPU fallback: if the shared library is missing or fails to load, every use of
the HIP path raises `GGNNLibraryError`.  `torch` is imported first so that the library
binds to the HIP runtime PyTorch already loaded (same SONAME `libamdhip64.so.7`).
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

import torch  # noqa: F401  (must precede the dlopen below)

LIB_NAME = "libggnn.so"
LIB_PATH = os.environ.get("GGNN_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)

GGNN_ABI_VERSION = 25
GGNN_UNIT_EDGES = 3
GGNN_EINFO_ROW = 20
GGNN_C = 96
GGNN_EDGE_PARAM_ROWS = 3
GGNN_DC_SLICE_BYTES = 14336
GGNN_PRECISION_BF16 = 1
GGNN_PRECISION_F16X2 = 2
GGNN_OUT_BLOCK_MAJOR = 0x100
GGNN_ETOPOLOGY = -3
GGNN_FLAG_F16_RANGE = 1
GGNN_ADAM_CHUNK, GGNN_ADAM_MAX_TENSORS, GGNN_ADAM_MAX_GROUPS = 4096, 384, 8
GGNN_ROWGEMM_MAX_PACK = 8
GGNN_MSE_MAX_TERMS, GGNN_MSE_BLOCKS = 4, 64
MODE_LSTM, MODE_LSTM_H0, MODE_RAW = 0, 1, 2

# Every symbol include/ggnn.h declares (tests/test_cabi.py checks the library exports them all).
EXPORTED_SYMBOLS = (
    "ggnn_version", "ggnn_error_string", "ggnn_gemm_mode", "ggnn_csr_workspace_bytes", "ggnn_csr_max_units",
    "ggnn_build_csr", "ggnn_build_csr_batch",
    "ggnn_edge_prepare", "ggnn_project", "ggnn_project_batch", "ggnn_period_gat_aggregate",
    "ggnn_period_gat_aggregate_batch", "ggnn_period_gat_aggregate_enc_batch", "ggnn_encoder_cell_batch",
    "ggnn_decoder_cell_batch",
    "ggnn_aggregate_bwd_partials", "ggnn_period_gat_aggregate_backward",
    "ggnn_lstm_epilogue", "ggnn_lstm_epilogue_batch", "ggnn_heads_regressor", "ggnn_heads_regressor_update",
    "ggnn_step_refresh_prepare", "ggnn_lstm_train_forward", "ggnn_lstm_train_backward",
    "ggnn_lstm_train_forward_batch", "ggnn_lstm_train_backward_batch", "ggnn_train_input_rows", "ggnn_sum_rows_batch",
    "ggnn_pack_weights_batch", "ggnn_pack_weights_backward_batch",
    "ggnn_wgrad_splits", "ggnn_wgrad", "ggnn_rowgemm_workspace_bytes", "ggnn_rowgemm_pack", "ggnn_rowgemm", "ggnn_rowgemm_pair", "ggnn_heads_regressor_backward",
    "ggnn_adam_step", "ggnn_masked_mse", "ggnn_sum_rows", "ggnn_pack_weights", "ggnn_pack_weights_backward",
    "ggnn_heads_classifier", "ggnn_heads_classifier_n", "ggnn_step_update", "ggnn_grain_centres", "ggnn_detect_events", "ggnn_detect_events_n", "ggnn_topology_update", "ggnn_topology_open", "ggnn_topology_apply",
    "ggnn_topology_counts", "ggnn_topology_export", "ggnn_topology_close", "ggnn_step_refresh",
    "ggnn_workspace_bytes",
)


class GGNNLibraryError(RuntimeError):
    pass


class GGNNError(RuntimeError):
    pass


class PrepareEdge(Structure):
    """Mirror of `ggnn_prepare_edge` (include/ggnn.h)."""
    _fields_ = [
        ("col", c_void_p), ("perm", c_void_p), ("row", c_void_p), ("edge_attr", c_void_p),
        ("x_src", c_void_p), ("x_dst", c_void_p), ("einfo", c_void_p),
        ("ldx_src", c_int64), ("ldx_dst", c_int64), ("E", c_int64), ("f_src", c_int64), ("E_dev", c_void_p),
    ]


class ProjectArgs(Structure):
    """Mirror of `ggnn_project_args`."""
    _fields_ = [
        ("X", c_void_p), ("H", c_void_p), ("Wp", c_void_p), ("bias", c_void_p), ("out", c_void_p),
        ("ldx", c_int64), ("ldh", c_int64), ("M", c_int64), ("ldo", c_int64),
        ("F", c_int32), ("k2", c_int32), ("ncols", c_int32), ("precision", c_int32),
    ]


class AggregateArgs(Structure):
    """Mirror of `ggnn_aggregate_args`."""
    _fields_ = [
        ("unit_ptr", c_void_p), ("units", c_void_p), ("einfo", c_void_p),
        ("p_src", c_void_p), ("p_dst", c_void_p), ("h_src", c_void_p),
        ("edge_params", c_void_p), ("agg", c_void_p),
        ("ldp_src", c_int64), ("ldp_dst", c_int64), ("ld_agg", c_int64),
        ("ldh_src", c_int64), ("n_src", c_int64), ("n_dst", c_int64), ("E", c_int64),
        ("v_off", c_int32), ("u_off", c_int32), ("u4_off", c_int32), ("a_off", c_int32),
        ("a_gstride", c_int32), ("sc_off", c_int32), ("n_gates", c_int32), ("pad_n", c_int32),
    ]


class AggregateEncArgs(Structure):
    """Mirror of `ggnn_aggregate_enc_args`."""
    _fields_ = [
        ("unit_ptr", c_void_p), ("units", c_void_p), ("einfo", c_void_p), ("p_dst", c_void_p),
        ("wv_frag", c_void_p), ("agg", c_void_p),
        ("ldp_dst", c_int64), ("ld_agg", c_int64), ("n_dst", c_int64), ("E", c_int64),
        ("u4_off", c_int32), ("a_off", c_int32), ("a_gstride", c_int32), ("sc_off", c_int32),
        ("n_gates", c_int32), ("reserved", c_int32),
    ]


class EncCellSweep(Structure):
    """Mirror of `ggnn_enc_cell_sweep`."""
    _fields_ = [("rowptr", c_void_p), ("einfo", c_void_p), ("E", c_int64)]


class EncCellArgs(Structure):
    """Mirror of `ggnn_enc_cell_args`."""
    _fields_ = [
        ("sweeps", EncCellSweep * 2),
        ("x_dst", c_void_p), ("h_out", c_void_p), ("c_out", c_void_p), ("wstream", c_void_p), ("w2_tail", c_void_p),
        ("flags", c_void_p),
        ("n_dst", c_int64), ("ldx", c_int64),
        ("n_in", c_int32), ("f_dst", c_int32),
    ]


class DecCellSweep(Structure):
    """Mirror of `ggnn_dec_cell_sweep`."""
    _fields_ = [
        ("rowptr", c_void_p), ("col", c_void_p), ("einfo", c_void_p), ("h_src", c_void_p), ("v_src", c_void_p),
        ("edge_params", c_void_p),
        ("E", c_int64), ("n_src", c_int64), ("ldh_src", c_int64), ("ldv", c_int64),
        ("v_off", c_int32), ("v_block_major", c_int32),
    ]


class DecCellArgs(Structure):
    """Mirror of `ggnn_dec_cell_args`."""
    _fields_ = [
        ("sweeps", DecCellSweep * 2),
        ("x_dst", c_void_p), ("h_dst", c_void_p), ("c_in", c_void_p), ("h_out", c_void_p), ("c_out", c_void_p),
        ("wstream", c_void_p), ("w2_tail", c_void_p), ("flags", c_void_p),
        ("n_dst", c_int64), ("ldx", c_int64), ("ldh", c_int64),
        ("n_in", c_int32), ("f_dst", c_int32),
    ]


class CsrArgs(Structure):
    """Mirror of `ggnn_csr_args`."""
    _fields_ = [
        ("edge_index", c_void_p), ("E", c_int64), ("n_src", c_int64), ("n_dst", c_int64),
        ("rowptr", c_void_p), ("col", c_void_p), ("perm", c_void_p), ("row", c_void_p), ("unit_ptr", c_void_p),
        ("units", c_void_p), ("flags", c_void_p), ("workspace", c_void_p), ("workspace_bytes", c_size_t),
    ]


class TopologyArgs(Structure):
    """Mirror of `ggnn_topology_args` (host memory)."""
    _fields_ = [
        ("pp", c_void_p), ("pq", c_void_p),
        ("n_pp", c_int64), ("n_pq", c_int64), ("pp_cap", c_int64), ("pq_cap", c_int64),
        ("x_joint", c_void_p), ("y_joint", c_void_p), ("y_grain_area", c_void_p), ("edge_prob", c_void_p),
        ("grain_event", c_void_p), ("mask_grain", c_void_p), ("mask_joint", c_void_p),
        ("active_grain", c_void_p), ("active_joint", c_void_p), ("switching", c_void_p), ("events_extra", c_void_p),
        ("n_joint", c_int64), ("n_grain", c_int64), ("ldx", c_int64), ("ldyg", c_int64), ("n_grain_event", c_int64),
        ("switching_cap", c_int64), ("extra_cap", c_int64), ("n_switching", c_int64), ("n_extra", c_int64),
        ("threshold", ctypes.c_double),
        ("error", ctypes.c_char * 192),
    ]


class WgradArgs(Structure):
    """Mirror of `ggnn_wgrad_args`."""
    _fields_ = [
        ("a", c_void_p), ("b", c_void_p), ("partial", c_void_p), ("out", c_void_p),
        ("lda", c_int64), ("ldb", c_int64), ("a_bstride", c_int64), ("b_bstride", c_int64), ("K", c_int64),
        ("M", c_int32), ("Nc", c_int32), ("batch", c_int32), ("n_split", c_int32),
        ("b_ins", c_void_p), ("ld_ins", c_int64), ("ins_off", c_int32), ("ins_w", c_int32),
    ]


GGNN_LSTM_TRAIN_MAX = 4
GGNN_TRAIN_ROWS_MAX = 4


class LstmTrainProblem(Structure):
    """Mirror of `ggnn_lstm_train_problem`."""
    _fields_ = [("z", c_void_p), ("p_dst", c_void_p), ("c_in", c_void_p), ("h_out", c_void_p), ("c_out", c_void_p),
                ("g_h", c_void_p), ("g_c", c_void_p), ("g_z", c_void_p), ("g_p_dst", c_void_p), ("g_c_in", c_void_p),
                ("ldp", c_int64), ("N", c_int64), ("s_off", c_int32), ("pad_off", c_int32), ("pad_n", c_int32),
                ("reserved", c_int32)]


GGNN_SUM_ROWS_MAX = 8


class SumRowsProblem(Structure):
    """Mirror of `ggnn_sum_rows_problem`."""
    _fields_ = [("in_", c_void_p), ("out", c_void_p), ("n_rows", c_int64), ("n_cols", c_int64), ("batch", c_int32),
                ("reserved", c_int32)]


class TrainRowsProblem(Structure):
    """Mirror of `ggnn_train_rows_problem`."""
    _fields_ = [("x", c_void_p), ("out", c_void_p), ("ldx", c_int64), ("ldo", c_int64), ("N", c_int64), ("F", c_int32),
                ("reserved", c_int32)]


class RowGemmArgs(Structure):
    """Mirror of `ggnn_rowgemm_args`."""
    _fields_ = [
        ("a", c_void_p), ("w", c_void_p), ("c", c_void_p), ("c_in", c_void_p), ("workspace", c_void_p),
        ("workspace_bytes", c_size_t),
        ("M", c_int64), ("lda", c_int64), ("ldc", c_int64), ("a_bstride", c_int64), ("c_bstride", c_int64),
        ("w_bstride", c_int64), ("w_nstride", c_int64), ("w_kstride", c_int64),
        ("K", c_int32), ("n_out", c_int32), ("batch", c_int32), ("precision", c_int32), ("prepacked", c_int32), ("reserved", c_int32),
    ]


class AdamTensor(Structure):
    """Mirror of `ggnn_adam_tensor` (one entry of the DEVICE table of ggnn_adam_step)."""
    _fields_ = [("param", c_void_p), ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p), ("n", c_int64), ("group", c_int32),
                ("reserved", c_int32)]


class AdamArgs(Structure):
    """Mirror of `ggnn_adam_args`."""
    _fields_ = [("table", c_void_p), ("chunk_tensor", c_void_p), ("chunk_index", c_void_p), ("step", c_void_p),
                ("grad", c_void_p * GGNN_ADAM_MAX_TENSORS), ("lr", c_float * GGNN_ADAM_MAX_GROUPS),
                ("weight_decay", c_float * GGNN_ADAM_MAX_GROUPS), ("beta1", c_float), ("beta2", c_float), ("eps", c_float),
                ("n_chunks", c_int32), ("n_tensors", c_int32), ("hyper", c_void_p)]


GGNN_PACK_OUTPUTS = 9
GGNN_PACK_TENSOR_SHIFT = 40
GGNN_PACK_MAX = 2


class PackArgs(Structure):
    """Mirror of `ggnn_pack_args`."""
    _fields_ = [("flat2", c_void_p), ("kq", c_void_p), ("kq_idx", c_void_p), ("idx3", c_void_p), ("packed", c_void_p),
                ("n_flat", c_int64), ("zero", c_int64), ("n_packed", c_int64),
                ("nb", c_int32), ("r", c_int32), ("c", c_int32), ("L", c_int32), ("coef", c_float), ("reserved", c_int32),
                ("params", c_void_p)]


class PackBwdArgs(Structure):
    """Mirror of `ggnn_pack_bwd_args`."""
    _fields_ = [("fwd", PackArgs), ("g_out", c_void_p * GGNN_PACK_OUTPUTS), ("g_off", c_int64 * (GGNN_PACK_OUTPUTS + 1)),
                ("g_w", c_int64 * GGNN_PACK_OUTPUTS), ("g_rs", c_int64 * GGNN_PACK_OUTPUTS), ("g_cs", c_int64 * GGNN_PACK_OUTPUTS),
                ("inv", c_void_p), ("inv_kq", c_void_p), ("g_flat2", c_void_p), ("g_kq", c_void_p), ("g_flat", c_void_p),
                ("n_flat2", c_int64), ("n_kq", c_int64), ("inv_m", c_int32), ("inv_kq_m", c_int32), ("n_tail", c_int64)]


class MseArgs(Structure):
    """Mirror of `ggnn_mse_args`."""
    _fields_ = [("pred", c_void_p * GGNN_MSE_MAX_TERMS), ("target", c_void_p * GGNN_MSE_MAX_TERMS),
                ("mask", c_void_p * GGNN_MSE_MAX_TERMS), ("g_pred", c_void_p * GGNN_MSE_MAX_TERMS),
                ("n", c_int64 * GGNN_MSE_MAX_TERMS), ("mask_div", c_int64 * GGNN_MSE_MAX_TERMS),
                ("workspace", c_void_p), ("loss", c_void_p), ("scale", c_float), ("n_terms", c_int32)]


class AggregateBwdArgs(Structure):
    """Mirror of `ggnn_aggregate_bwd_args`."""
    _fields_ = [
        ("rowptr", c_void_p), ("col", c_void_p), ("einfo", c_void_p),
        ("p_src", c_void_p), ("p_dst", c_void_p), ("h_src", c_void_p),
        ("edge_params", c_void_p), ("agg", c_void_p), ("g_agg", c_void_p),
        ("r_rowptr", c_void_p), ("r_dst", c_void_p), ("r_slot", c_void_p),
        ("edge_alpha", c_void_p), ("edge_ds", c_void_p), ("ep_partial", c_void_p),
        ("g_p_dst", c_void_p), ("g_p_src", c_void_p), ("g_h_src", c_void_p),
        ("ldp_src", c_int64), ("ldp_dst", c_int64), ("ld_agg", c_int64),
        ("ldh_src", c_int64), ("n_src", c_int64), ("n_dst", c_int64), ("E", c_int64),
        ("n_partials", c_int64),
        ("v_off", c_int32), ("u_off", c_int32), ("u4_off", c_int32), ("a_off", c_int32),
        ("a_gstride", c_int32), ("sc_off", c_int32), ("n_gates", c_int32), ("g_h_accumulate", c_int32),
    ]


class EpilogueArgs(Structure):
    """Mirror of `ggnn_epilogue_args`."""
    _fields_ = [
        ("agg", c_void_p), ("w2", c_void_p), ("p_dst", c_void_p), ("c_in", c_void_p),
        ("h_out", c_void_p), ("c_out", c_void_p), ("raw_out", c_void_p),
        ("ldp", c_int64), ("N", c_int64),
        ("Ka", c_int32), ("s_off", c_int32), ("n_gates", c_int32), ("mode", c_int32),
        ("w2_planes", c_void_p), ("ld_agg", c_int64), ("g_stride", c_int32), ("reserved", c_int32),
    ]


class RefreshEdge(Structure):
    """Mirror of `ggnn_refresh_edge`."""
    _fields_ = [
        ("edge_index", c_void_p), ("x_src", c_void_p), ("x_dst", c_void_p), ("edge_attr", c_void_p),
        ("ldx_src", c_int64), ("ldx_dst", c_int64), ("n_src", c_int64), ("n_dst", c_int64),
        ("E", c_int64), ("E_dev", c_void_p),
    ]


_lib = None


def _declare(lib):
    lib.ggnn_version.restype = c_int
    lib.ggnn_version.argtypes = []
    lib.ggnn_gemm_mode.restype = c_int
    lib.ggnn_gemm_mode.argtypes = []
    lib.ggnn_error_string.restype = c_char_p
    lib.ggnn_error_string.argtypes = [c_int]
    lib.ggnn_csr_workspace_bytes.restype = c_size_t
    lib.ggnn_csr_workspace_bytes.argtypes = [c_int64, c_int64]
    lib.ggnn_build_csr.restype = c_int
    lib.ggnn_csr_max_units.restype = c_int64
    lib.ggnn_csr_max_units.argtypes = [c_int64, c_int64]
    lib.ggnn_build_csr.argtypes = [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_size_t, c_void_p]
    lib.ggnn_build_csr_batch.restype = c_int
    lib.ggnn_build_csr_batch.argtypes = [POINTER(CsrArgs), c_int, c_void_p]
    lib.ggnn_edge_prepare.restype = c_int
    lib.ggnn_edge_prepare.argtypes = [POINTER(PrepareEdge), c_int, c_void_p]
    lib.ggnn_project.restype = c_int
    lib.ggnn_project.argtypes = [c_void_p, c_int64, c_int, c_void_p, c_int64, c_int, c_void_p,
                                 c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p]
    lib.ggnn_project_batch.restype = c_int
    lib.ggnn_project_batch.argtypes = [POINTER(ProjectArgs), c_int, c_void_p]
    lib.ggnn_period_gat_aggregate.restype = c_int
    lib.ggnn_period_gat_aggregate.argtypes = [POINTER(AggregateArgs), c_void_p]
    lib.ggnn_period_gat_aggregate_batch.restype = c_int
    lib.ggnn_period_gat_aggregate_batch.argtypes = [POINTER(AggregateArgs), c_int, c_void_p]
    lib.ggnn_period_gat_aggregate_enc_batch.restype = c_int
    lib.ggnn_period_gat_aggregate_enc_batch.argtypes = [POINTER(AggregateEncArgs), c_int, c_void_p]
    lib.ggnn_encoder_cell_batch.restype = c_int
    lib.ggnn_encoder_cell_batch.argtypes = [POINTER(EncCellArgs), c_int, c_void_p]
    lib.ggnn_decoder_cell_batch.restype = c_int
    lib.ggnn_decoder_cell_batch.argtypes = [POINTER(DecCellArgs), c_int, c_void_p]
    lib.ggnn_aggregate_bwd_partials.restype = c_int64
    lib.ggnn_aggregate_bwd_partials.argtypes = [c_int64]
    lib.ggnn_period_gat_aggregate_backward.restype = c_int
    lib.ggnn_period_gat_aggregate_backward.argtypes = [POINTER(AggregateBwdArgs), c_void_p]
    lib.ggnn_lstm_epilogue.restype = c_int
    lib.ggnn_lstm_epilogue.argtypes = [POINTER(EpilogueArgs), c_void_p]
    lib.ggnn_lstm_epilogue_batch.restype = c_int
    lib.ggnn_lstm_epilogue_batch.argtypes = [POINTER(EpilogueArgs), c_int, c_void_p]
    lib.ggnn_heads_regressor.restype = c_int
    lib.ggnn_heads_regressor.argtypes = [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.ggnn_heads_regressor_update.restype = c_int
    lib.ggnn_heads_regressor_update.argtypes = [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                                                c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_float, c_float, c_void_p, c_void_p]
    lib.ggnn_step_refresh_prepare.restype = c_int
    lib.ggnn_step_refresh_prepare.argtypes = [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_float, c_void_p,
                                              POINTER(PrepareEdge), c_int, c_void_p, c_void_p, c_void_p]
    lib.ggnn_lstm_train_forward.restype = c_int
    lib.ggnn_lstm_train_forward.argtypes = [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int64,
                                            c_int, c_void_p]
    lib.ggnn_lstm_train_backward.restype = c_int
    lib.ggnn_lstm_train_backward.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_int64, c_int, c_void_p, c_int64, c_int, c_void_p]
    for fn in (lib.ggnn_lstm_train_forward_batch, lib.ggnn_lstm_train_backward_batch):
        fn.restype = c_int
        fn.argtypes = [POINTER(LstmTrainProblem), c_int, c_int, c_void_p]
    lib.ggnn_sum_rows_batch.restype = c_int
    lib.ggnn_sum_rows_batch.argtypes = [POINTER(SumRowsProblem), c_int, c_void_p]
    lib.ggnn_train_input_rows.restype = c_int
    lib.ggnn_train_input_rows.argtypes = [POINTER(TrainRowsProblem), c_int, c_void_p]
    lib.ggnn_wgrad_splits.restype = c_int
    lib.ggnn_wgrad_splits.argtypes = [c_int64, c_int, c_int, c_int]
    lib.ggnn_wgrad.restype = c_int
    lib.ggnn_wgrad.argtypes = [POINTER(WgradArgs), c_void_p]
    lib.ggnn_rowgemm_workspace_bytes.restype = c_size_t
    lib.ggnn_rowgemm_workspace_bytes.argtypes = [c_int32, c_int32, c_int32]
    lib.ggnn_rowgemm_pack.restype = c_int
    lib.ggnn_rowgemm_pack.argtypes = [POINTER(RowGemmArgs), c_int, c_void_p]
    lib.ggnn_rowgemm.restype = c_int
    lib.ggnn_rowgemm.argtypes = [POINTER(RowGemmArgs), c_void_p]
    lib.ggnn_rowgemm_pair.restype = c_int
    lib.ggnn_rowgemm_pair.argtypes = [POINTER(RowGemmArgs), c_void_p]
    lib.ggnn_adam_step.restype = c_int
    lib.ggnn_adam_step.argtypes = [POINTER(AdamArgs), c_void_p]
    lib.ggnn_sum_rows.restype = c_int
    lib.ggnn_sum_rows.argtypes = [c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p]
    lib.ggnn_masked_mse.restype = c_int
    lib.ggnn_masked_mse.argtypes = [POINTER(MseArgs), c_void_p]
    lib.ggnn_pack_weights.restype = c_int
    lib.ggnn_pack_weights.argtypes = [POINTER(PackArgs), c_void_p]
    lib.ggnn_pack_weights_backward.restype = c_int
    lib.ggnn_pack_weights_backward.argtypes = [POINTER(PackBwdArgs), c_void_p]
    lib.ggnn_pack_weights_batch.restype = c_int
    lib.ggnn_pack_weights_batch.argtypes = [POINTER(PackArgs), c_int, c_void_p]
    lib.ggnn_pack_weights_backward_batch.restype = c_int
    lib.ggnn_pack_weights_backward_batch.argtypes = [POINTER(PackBwdArgs), c_int, c_void_p]
    lib.ggnn_heads_regressor_backward.restype = c_int
    lib.ggnn_heads_regressor_backward.argtypes = [c_int64, c_int64] + [c_void_p] * 11
    lib.ggnn_heads_classifier.restype = c_int
    lib.ggnn_heads_classifier.argtypes = [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.ggnn_heads_classifier_n.restype = c_int
    lib.ggnn_heads_classifier_n.argtypes = [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.ggnn_step_update.restype = c_int
    lib.ggnn_step_update.argtypes = [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int,
                                     c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p]
    lib.ggnn_step_refresh.restype = c_int
    lib.ggnn_step_refresh.argtypes = [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                      c_float, c_void_p, POINTER(RefreshEdge), c_int, c_void_p]
    lib.ggnn_grain_centres.restype = c_int
    lib.ggnn_grain_centres.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p,
                                       c_float, c_void_p, c_int64, c_int64, c_void_p, c_void_p]
    lib.ggnn_detect_events.restype = c_int
    lib.ggnn_detect_events.argtypes = [c_void_p, c_void_p, c_int64, c_float, c_void_p, c_void_p, c_int64,
                                       c_float, c_void_p, c_void_p, c_void_p]
    lib.ggnn_detect_events_n.restype = c_int
    lib.ggnn_detect_events_n.argtypes = [c_void_p, c_void_p, c_int64, c_float, c_void_p, c_void_p, c_int64, c_void_p,
                                         c_float, c_void_p, c_void_p, c_void_p]
    lib.ggnn_topology_update.restype = c_int
    lib.ggnn_topology_update.argtypes = [POINTER(TopologyArgs)]
    lib.ggnn_topology_open.restype = c_int
    lib.ggnn_topology_open.argtypes = [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                       POINTER(c_void_p), ctypes.c_char_p]
    lib.ggnn_topology_apply.restype = c_int
    lib.ggnn_topology_apply.argtypes = [c_void_p, POINTER(TopologyArgs)]
    lib.ggnn_topology_counts.restype = c_int
    lib.ggnn_topology_counts.argtypes = [c_void_p, POINTER(c_int64), POINTER(c_int64)]
    lib.ggnn_topology_export.restype = c_int
    lib.ggnn_topology_export.argtypes = [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64]
    lib.ggnn_topology_close.restype = None
    lib.ggnn_topology_close.argtypes = [c_void_p]
    lib.ggnn_workspace_bytes.restype = c_size_t
    lib.ggnn_workspace_bytes.argtypes = [c_int64, c_int64, c_int64]


def load():
    """Load libggnn.so (once).  Raises GGNNLibraryError if it is not built / not loadable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GGNNLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C graingraphnn_amd/csrc`.  graingraphnn_amd has no CPU fallback.")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:  # pragma: no cover - depends on the environment
        raise GGNNLibraryError(f"cannot load {LIB_PATH}: {exc}") from exc
    missing = [s for s in EXPORTED_SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise GGNNLibraryError(f"{LIB_PATH} lacks symbols {missing}")
    _declare(lib)
    if lib.ggnn_version() != GGNN_ABI_VERSION:
        raise GGNNLibraryError(
            f"{LIB_PATH} has ABI version {lib.ggnn_version()}, expected {GGNN_ABI_VERSION}: rebuild")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().ggnn_error_string(rc).decode()
        raise GGNNError(f"{what} failed: {msg} (code {rc})")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())


def current_stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)
