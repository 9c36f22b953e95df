#!/usr/bin/env python3
"""Rollout throughput of the GrainGNN hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One *step* = Rmodel.forward + Cmodel.forward + Rmodel.update + z advance + edge-length
refresh (test.py:382-407, 562-575) on the synthetic 10 000-grain / 20 000-junction periodic
honeycomb of SURVEY.md 8(d) (BASELINE config "synthetic 10k-grain / 20k-junction periodic
heterograph, 500-step rollout"), static topology, weights RandomState(0) x0.3, inputs resident
in HBM before the timed region.  N > 1 runs one independent replica per GPU (weak scaling; a
single 10k-grain graph does not shard) and all-gathers the final states over
RCCL inside the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from graingraphnn_amd import GrainRollout, synthetic  # noqa: E402
from graingraphnn_amd.backend import default_backend  # noqa: E402
from graingraphnn_amd.dist import gather_states  # noqa: E402
from graingraphnn_amd.models import GrainNN_classifier, GrainNN_regressor  # noqa: E402
from graingraphnn_amd.seeding import load_seeded  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SPAN = 6


def algorithmic_bytes(n_src, n_dst, E, G):
    """each distinct operand row once + each output row once, fp32, int32 CSR."""
    return 4 * (G * 96 * (2 * n_src + 2 * n_dst) + 3 * (n_src + n_dst) + E) + 4 * E + 4 * (n_dst + 1) + 8 * G * n_dst


def build(device, n=100, fold=10, seed=0, scale=0.3):
    x, ei, ea, off = synthetic.honeycomb(n, fold, seed, return_offset=True)
    hp = synthetic.default_hyper(device)
    R = GrainNN_regressor(hp)
    Cm = GrainNN_classifier(hp, R)
    load_seeded(R, seed, scale).eval()
    load_seeded(Cm, seed + 1, scale).eval()
    X, EI, EA = synthetic.to_torch(x, ei, ea, device)
    return R.to(device), Cm.to(device), X, EI, EA, (x, ei, ea, float(fold), off)


def build_generated(device, lxd=368.0, fold=9.0, seed=0, scale=0.3):
    """The reference generator's structure at the headline's size (--lxd=368 -> 9 775 grains), folded by the
    INTEGER factor nearest to lxd / 40 (test.py:29-55 folds by lxd / 40 = 9.2; a non-integer factor leaves edges across the
    global seam with inconsistent min-images), models as in `build`."""
    x, ei, ea = synthetic.generate(lxd, seed)
    x, ea = {k: v.copy() for k, v in x.items()}, {k: v.copy() for k, v in ea.items()}
    off = synthetic.scale_feature_patchs(fold, x, ea)
    hp = synthetic.default_hyper(device)
    R = GrainNN_regressor(hp)
    Cm = GrainNN_classifier(hp, R)
    load_seeded(R, seed, scale).eval()
    load_seeded(Cm, seed + 1, scale).eval()
    X, EI, EA = synthetic.to_torch(x, ei, ea, device)
    return R.to(device), Cm.to(device), X, EI, EA, (x, ei, ea, float(fold), off)


FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 matrix (= vector) peak
FP16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 / fp16 matrix peak


class _HipEvent:
    """A HIP event recorded on the launch stream, created without the system-scope release fence
    (hipEventDisableSystemFence): a time stamp in the stream and nothing else.  (Measured: the fence is not what made
    round 4's first brackets read high -- the Python wrapper inside them was, see EventTimedBackend._timed.)"""
    _hip = None
    DISABLE_SYSTEM_FENCE = 0x20000000

    def __init__(self):
        import ctypes
        if _HipEvent._hip is None:
            _HipEvent._hip = ctypes.CDLL("libamdhip64.so")
        self._c = ctypes
        self.h = ctypes.c_void_p()
        rc = _HipEvent._hip.hipEventCreateWithFlags(ctypes.byref(self.h), ctypes.c_uint(self.DISABLE_SYSTEM_FENCE))
        if rc != 0:
            raise RuntimeError(f"hipEventCreateWithFlags: {rc}")

    def record(self):
        rc = _HipEvent._hip.hipEventRecord(self.h, self._c.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError(f"hipEventRecord: {rc}")

    def elapsed_time(self, other):
        ms = self._c.c_float()
        _HipEvent._hip.hipEventSynchronize(other.h)
        rc = _HipEvent._hip.hipEventElapsedTime(self._c.byref(ms), self.h, other.h)
        if rc != 0:
            raise RuntimeError(f"hipEventElapsedTime: {rc}")
        return float(ms.value)

    def __del__(self):
        try:
            _HipEvent._hip.hipEventDestroy(self.h)
        except Exception:
            pass


class EventTimedBackend:
    """Wraps the HIP backend so that the launches of the four heavy kernel families are bracketed by
    HIP events recorded on the launch stream: the decoder sweep (aggregate_kernel<4, true>, the
    kernel graded against the HBM roofline), the encoder cell (enc_cell_kernel; aggregate_enc_kernel<3> with
    GGNN_ENC=split), the
    decoder projection (project_x6_kernel) and the decoder gate GEMM + LSTM (gates_x6_kernel<4, 0>).
    An event bracket also contains the dispatch/event overhead of the launch (3-6 us in eager mode),
    which rocprofv3's kernel durations do not.  It is calibrated right behind the launch on a kernel
    that is too small to care about cache state: a [1 launch] and a [2 launches] bracket of it
    differ by exactly its duration, what is left of the first bracket is the overhead.
    (Differencing the kernel itself would time a re-run whose operands the first run left in the
    256 MB MALL.)"""

    def __init__(self, inner):
        self.inner = inner
        self.events = {}

    def __getattr__(self, name):
        return getattr(self.inner, name)

    def _timed(self, key, fn, arg, meta):
        """Bracket the C-ABI call itself: the backend's Python wrapper (argument checks, ctypes structs: 50-100 us for
        a cell) runs BEFORE e0 is recorded -- with the encoder cell at 39 us the GPU is no longer behind the host in the
        eager pass, and a bracket around the wrapper counted its preparation as kernel time (136 against 109 us for
        the decoder cell in rocprofv3's trace of the same launches)."""
        e0, e1, c0, c1, c2 = (_HipEvent() for _ in range(5))   # HIP events on the launch stream, no cache-flushing fence
        inner, n_calls = self.inner, [0]
        launch = inner._launch

        def bracketed(cfn, name, *cargs):
            n_calls[0] += 1
            e0.record()
            launch(cfn, name, *cargs)
            e1.record()

        inner._launch = bracketed
        try:
            fn(arg)
        finally:
            del inner._launch            # back to the class's method
        assert n_calls[0] == 1, "one C-ABI launch per timed call"
        c0.record()
        self.null.zero_()
        c1.record()
        self.null.zero_()
        self.null.zero_()
        c2.record()
        self.events.setdefault(key, []).append((e0, e1, c0, c1, c2, meta))

    def aggregate_batch(self, sweeps):
        if sweeps[0][-1] != 4:  # n_gates
            return self.inner.aggregate_batch(sweeps)
        nbytes = sum(algorithmic_bytes(sw[2].size(0), sw[3].size(0), sw[0].E, 4) for sw in sweeps)
        self._timed("dec_sweep", self.inner.aggregate_batch, sweeps, (nbytes, len(sweeps)))

    def aggregate_enc_batch(self, sweeps):
        # (csr, einfo, p_dst, wv_frag, agg, u4_off, a_off, a_gstride, sc_off, n_gates)
        real = 0
        for sw in sweeps:
            E, n_dst, G = sw[0].E, sw[2].size(0), sw[-1]
            n_units = int(sw[0].units.size(0))
            real += 80 * E + 4 * (n_dst + 1) + 32 * n_units + 4 * n_dst * G * 16 + 4 * n_dst * G * 98
        self._timed("enc_sweep", self.inner.aggregate_enc_batch, sweeps, (real, len(sweeps)))

    def encoder_cell_batch(self, problems):
        # (sweeps [(csr, einfo)], x_dst, wstream, w2_tail, h_out, c_out): matrix-pipe cycles as launched
        cycles = 0.0
        for sweeps, x_dst, *_ in problems:
            n = x_dst.size(0)
            n_t = (n + 15) // 16
            per_gate = n_t * 18 * 16                                   # skip: 6 column tiles x 3 products, one k-step
            for csr, _ in sweeps:
                deg = (csr.rowptr[1:] - csr.rowptr[:-1]).long()
                nu = torch.clamp((deg + 2) // 3, min=1)
                units = int(torch.nn.functional.pad(nu, (0, n_t * 16 - n)).view(n_t, 16).max(1).values.sum())
                # per tile: u4 (3 fp16 MFMAs of 16 cycles) + lin_l2 (3 k-steps x 6 column tiles x 3 products) + 6 fp32
                # MFMAs of 32 cycles for the rank-1 columns; per unit of <= 3 in-edges per node: 9 score + 54 value MFMAs
                per_gate += n_t * ((3 + 54) * 16 + 6 * 32) + units * 63 * 16
            cycles += 3 * per_gate
        self._timed("enc_cell", self.inner.encoder_cell_batch, problems, (cycles, len(problems)))

    def project_batch(self, problems):
        if problems[0][2] is None:  # encoder (K <= 12): store-bound, not a GEMM worth grading
            return self.inner.project_batch(problems)
        flops = sum(2.0 * x.size(0) * (F + h.size(1)) * wp.size(0) for x, F, h, wp, bp, out, *_ in problems)
        nbytes = sum(4.0 * (x.size(0) * (F + h.size(1) + wp.size(0)) + wp.numel() + bp.numel())
                     for x, F, h, wp, bp, out, *_ in problems)     # node rows in, weights once, projected rows out
        self._timed("dec_project", self.inner.project_batch, problems, (flops, len(problems), nbytes))

    def lstm_epilogue_batch(self, problems):
        if problems[0][9] != 0:  # GGNN_MODE_LSTM only (decoder)
            return self.inner.lstm_epilogue_batch(problems)
        flops = sum(2.0 * p[0].size(0) * p[8] * 96 * p[1].size(2) for p in problems)
        # aggregates + skip rows + old cell state in, weights once, h and c out
        nbytes = sum(4.0 * (p[0].size(0) * (p[8] * p[1].size(2) + p[8] * 96 + 96 + 2 * 96) + p[1].numel()) for p in problems)
        self._timed("dec_gates", self.inner.lstm_epilogue_batch, problems, (flops, len(problems), nbytes))

    def decoder_cell_batch(self, problems):
        # (sweeps [(csr, einfo, h_src, v_src, v_off, ep)], x_dst, h_dst, c_in, wstream, w2_tail, h_out, c_out)
        flops = nbytes = canon = 0.0
        n_sweeps = 0
        for sweeps, x_dst, h_dst, c_in, wstream, w2_tail, h_out, c_out, *_ in problems:
            n, n_in = x_dst.size(0), len(sweeps)
            flops += 2.0 * n * 4 * (n_in * (128 * 112 + 98 * 96) + 128 * 96)
            nbytes += 4.0 * n * (x_dst.size(1) + 96 + 96 + 2 * 96) + 2.0 * wstream.numel()
            for csr, einfo, h_src, v_src, v_off, ep, *_ in sweeps:   # value rows + hidden rows of the sources, edge records
                nbytes += 4.0 * (h_src.size(0) * (4 * 96 + 96) + csr.E * 20 + csr.E + n + 1)
                canon += algorithmic_bytes(h_src.size(0), n, csr.E, 4)   # what the cell's sweeps are defined to move
                n_sweeps += 1
        self._timed("dec_cell", self.inner.decoder_cell_batch, problems, (flops, len(problems), nbytes, canon, n_sweeps))

    def summary(self, key):
        evs = self.events.get(key)
        if not evs:
            return None
        bracket = np.array([ev[0].elapsed_time(ev[1]) for ev in evs]) * 1e3
        null1 = np.array([ev[2].elapsed_time(ev[3]) for ev in evs]) * 1e3
        null2 = np.array([ev[3].elapsed_time(ev[4]) for ev in evs]) * 1e3
        overhead = float(np.median(null1 - (null2 - null1)))     # bracket minus the null kernel itself
        return {"avg_us": float(np.mean(bracket)) - overhead, "bracket_us": float(np.mean(bracket)),
                "overhead_us": overhead, "work": float(np.mean([ev[5][0] for ev in evs])),
                "work2": float(np.mean([ev[5][2] if len(ev[5]) > 2 else 0.0 for ev in evs])),
                "work3": float(np.mean([ev[5][3] if len(ev[5]) > 3 else 0.0 for ev in evs])),
                "work4": float(np.mean([ev[5][4] if len(ev[5]) > 4 else 0.0 for ev in evs])),
                "per_launch": round(float(np.mean([ev[5][1] for ev in evs])), 2), "n": len(evs)}


def measure_roofline(ro, n_steps):
    """Average durations of the heavy kernels inside real rollout steps (eager launches, one set
    per model on ONE stream, so that no other kernel shares the chip with the launch being timed;
    HIP events on the launch stream, minus the bracket overhead calibrated on a null kernel).
    Returns (roofline of the default plan's dominant decoder kernel, roofline of the encoder cell, GEMM records,
    roofline of the three-kernel plan's sweep when the default plan is the fused cell)."""
    timed = EventTimedBackend(ro.be)
    timed.null = torch.zeros(64, device="cuda")
    inner = timed.inner
    fused_plan = bool(getattr(inner, "fused_decoder", False))

    def timed_steps():
        nonlocal n_steps
        ro.be, side, joint = timed, ro._side, ro.joint_launches
        ro._side, ro.joint_launches = None, False
        try:
            # Keep the GPU behind the host for the whole pass: an event bracket spans from the GPU reaching e0 to the GPU
            # reaching e1, so a launch the host is still preparing when the GPU gets to e0 counts its preparation as
            # kernel time (round 4: with 39 us encoder cells the eager host path -- ~25 us per C-ABI call -- fell behind
            # and every bracket read 13-30 us high against rocprofv3).  A spin kernel in front lets the host queue the
            # pass's ~12 launches per step ahead; the kernels then run back to back as they do in the captured step.
            torch.cuda._sleep(int(12e6) * n_steps)
            for _ in range(n_steps):
                ro._enqueue_step()
            torch.cuda.synchronize()
        finally:
            ro.be, ro._side, ro.joint_launches = inner, side, joint

    timed_steps()
    plans = {"fused" if fused_plan else "split": timed.events}
    if fused_plan:
        # the default plan runs the decoder cell as ONE kernel: the three kernels of the other plan (GGNN_DEC=split),
        # among them the sweep that earlier rounds' `roofline` described, are timed in a second pass of the same steps
        keep_plan, inner.fused_decoder = inner.fused_decoder, False
        try:
            n_keep, n_steps = n_steps, 2
            timed_steps()          # untimed in effect: the other plan's buffers and code objects are touched for the first time
            n_steps = n_keep
            timed.events = {}
            timed_steps()
        finally:
            inner.fused_decoder = keep_plan
        plans["split"] = timed.events

    def summary(plan, key):
        timed.events = plans.get(plan, {})
        return timed.summary(key)

    default_plan = "fused" if fused_plan else "split"
    roof = enc = sweep = None
    gemm = []
    d = summary("split", "dec_sweep")
    if d:
        achieved = d["work"] / d["avg_us"] / 1e3
        sweep = {"bound": "hbm", "kernel": "ggnn::aggregate_kernel<4, true>", "plan": "GGNN_DEC=split",
                 "achieved": round(achieved, 1),
                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                 "traffic": pmc_traffic("ggnn::aggregate_kernel<4, true>"), "traffic_source": "recorded rocprofv3 --pmc passes (profiles/r6_pmc_kernels.json: tools/profile_round.sh on this library, matched by ABI + source hash); null traffic = no matching record", "avg_launch_us": round(d["avg_us"], 2),
                 "event_bracket_us": round(d["bracket_us"], 2), "bracket_overhead_us": round(d["overhead_us"], 2),
                 "algorithmic_bytes_per_launch": int(d["work"]), "launches_timed": d["n"],
                 "sweeps_per_launch": d["per_launch"]}
    f = summary("fused", "dec_cell")
    if f:
        # the fused decoder cell: graded like the sweeps it contains (SURVEY 8(d) bytes of its 2-3 sweeps per launch /
        # its duration) although it also runs the cell's three GEMMs and the LSTM update in that time; what it
        # actually has to move (value / hidden rows of the sources once, edge records, x / h / c in, h / c out, the
        # weight stream once) is `bytes_as_built_per_launch`, what it did move is `traffic`
        achieved = f["work3"] / f["avg_us"] / 1e3
        roof = {"bound": "hbm", "kernel": "ggnn::dec_cell_kernel", "plan": "default (GGNN_DEC=fused)",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic("ggnn::dec_cell_kernel", "fused"), "traffic_source": "recorded rocprofv3 --pmc passes (profiles/r6_pmc_kernels.json: tools/profile_round.sh on this library, matched by ABI + source hash); null traffic = no matching record",
                "avg_launch_us": round(f["avg_us"], 2), "event_bracket_us": round(f["bracket_us"], 2),
                "bracket_overhead_us": round(f["overhead_us"], 2), "algorithmic_bytes_per_launch": int(f["work3"]),
                "bytes_as_built_per_launch": int(f["work2"]), "launches_timed": f["n"],
                "sweeps_per_launch": f["work4"], "problems_per_launch": f["per_launch"],
                "fp32_equivalent_tflops": round(f["work"] / f["avg_us"] / 1e6, 1),
                "note": "one launch = the whole decoder cell of a model (destination-side projections, its 2-3 sweeps, "
                        "lin_l2, skip, LSTM); algorithmic bytes = SURVEY 8(d) for those sweeps, the same unit of work as "
                        "`roofline_split_sweep` (the sweep kernel of the three-kernel plan, timed in the same run)"}
    else:
        roof = sweep
    e = summary(default_plan, "enc_sweep")
    if e:
        achieved = e["work"] / e["avg_us"] / 1e3
        enc = {"bound": "hbm", "kernel": "ggnn::aggregate_enc_kernel<3>", "achieved": round(achieved, 1),
               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
               "avg_launch_us": round(e["avg_us"], 2), "sweeps_per_launch": e["per_launch"],
               "bytes_per_launch": int(e["work"]),
               "bytes": "what the kernel as built must move, per sweep: 80 E (edge records) + 4 (n_dst + 1) + 32 "
                        "n_units (unit table) + 64 G n_dst (score tails) + 392 G n_dst (rows written); the "
                        "SURVEY 8(d) formula does not apply (no K / V / Q rows exist)"}
    c = summary(default_plan, "enc_cell")
    if c:  # the fused encoder cell: matrix-pipe cycles as launched against the chip's 1024 SIMDs at 2.4 GHz
        busy = c["work"] / (1024 * 2400.0 * c["avg_us"])
        enc = {"bound": "mfma", "kernel": "ggnn::enc_cell_kernel (the whole encoder cell of a model: one launch)",
               "achieved": round(busy * FP16_MFMA_PEAK_TFLOPS, 1), "peak": FP16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
               "frac": round(busy, 4), "avg_launch_us": round(c["avg_us"], 2), "problems_per_launch": c["per_launch"],
               "traffic": pmc_traffic("ggnn::enc_cell_kernel", "fused"), "traffic_source": "recorded rocprofv3 --pmc passes (profiles/r6_pmc_kernels.json: tools/profile_round.sh on this library, matched by ABI + source hash); null traffic = no matching record",
               "mfma_cycles_per_launch": int(c["work"]),
               "note": "frac = matrix-pipe cycles of the launch (fp16 MFMAs at 16 cycles: three per fp32 product, half of "
                       "every 16-slot k-step's 32-deep reduction is padding; six fp32 MFMAs of 32 cycles per pass) / "
                       "(1024 SIMDs x 2.4 GHz x duration); achieved = frac x the dense fp16 matrix peak; the kernel is "
                       "bound by vector + matrix instruction issue at two waves per SIMD (profiles/README.md)"}
    for plan, key, kname, name in (
            ("fused", "dec_project", "ggnn::project_x6_kernel", "ggnn::project_x6_kernel (decoder projection of the default plan: value rows only)"),
            ("split", "dec_project", "ggnn::project_x6_kernel", "ggnn::project_x6_kernel (decoder projection, GGNN_DEC=split: both node types of a model)"),
            ("split", "dec_gates", "ggnn::gates_x6_kernel<4, 0>", "ggnn::gates_x6_kernel<4, 0> (decoder gate GEMM + LSTM, GGNN_DEC=split)")):
        g = summary(plan, key)
        if g:
            flops, nbytes = g["work"], g["work2"]
            gbs = nbytes / g["avg_us"] / 1e3
            tf = flops / g["avg_us"] / 1e6
            gemm.append({"bound": "hbm", "kernel": name, "plan": plan, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(kname, plan), "traffic_source": "recorded rocprofv3 --pmc passes (profiles/r6_pmc_kernels.json: tools/profile_round.sh on this library, matched by ABI + source hash); null traffic = no matching record",
                         "algorithmic_bytes_per_launch": int(nbytes), "avg_launch_us": round(g["avg_us"], 2),
                         "problems_per_launch": g["per_launch"],
                         "fp32_equivalent_tflops": round(tf, 1),
                         "frac_of_bf16_pipe": round((3 if plan == "fused" and default_backend().f16_projection() else 6) * tf / 2500.0, 4),
                         "note": "graded against HBM: every operand row read once + every output row written once, fp32, as "
                                 "launched (the matrix work runs as 6 bf16 MFMA products per fp32 product -- 3 fp16 products for the "
                                 "value rows of the default plan, GGNN_PRECISION_F16X2 --: frac_of_bf16_pipe = products x "
                                 "fp32-equivalent rate / 2.5 PFLOP/s dense 16-bit matrix peak)"})
    timed.events = {}
    return roof, enc, gemm, (sweep if fused_plan else None)


def kernel_source_hash():
    """What a PMC record is tied to: the ABI version of the loaded library + the sources of the kernels it
    describes (the sweep, the decoder GEMMs, the fused cells)."""
    import hashlib
    h = hashlib.sha256()
    for name in ("aggregate.hip", "project_x6.hip", "gates_x6.hip", "enc_cell.hip", "dec_cell.hip", "cell_common.h", "common.h"):
        with open(os.path.join(ROOT, "graingraphnn_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return f"abi{default_backend().lib.ggnn_version()}-{h.hexdigest()[:16]}"


def pmc_traffic(kernel, plan="split"):
    """HBM-side bytes per launch of `kernel` from the rocprofv3 --pmc passes recorded in
    profiles/r6_pmc_kernels.json (PMC collection cannot run inside this process; the file says how it was taken
    and corrected, tools/pmc_kernels.py).  The record is stamped with the ABI version and a hash of the kernels'
    sources: None when the file is absent or does not describe the library that is running."""
    try:
        with open(os.path.join(ROOT, "profiles", "r6_pmc_kernels.json")) as f:
            doc = json.load(f)
        if doc["kernel_source_hash"] != kernel_source_hash():
            return None
        rec = doc["plans"][plan][kernel]
        return int(rec["read_bytes"] + rec["written_bytes"])
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(inputs, seed=0, scale=0.3, budget_s=12.0):
    """The oracle (reference formulation, plain PyTorch CPU) on the same workload, bounded.
    PyTorch's default of one thread per hardware thread is far from its best on a 2 x 64-core host
    (measured: 128 threads 8.2 s/step, 16 threads 0.9 s/step, 1 thread 4.0 s/step), so a few
    thread counts are tried first and the baseline is quoted at the fastest one; the 1-thread
    figure, the physical core count and the CPU model are reported beside it."""
    from oracle import grainnn_oracle as oracle
    x, ei, ea, factor, off = inputs
    centres = (factor, torch.from_numpy(off))
    hp = synthetic.default_hyper("cpu")
    R = oracle.GrainNN_regressor(hp)
    Cm = oracle.GrainNN_classifier(hp, R)
    load_seeded(R, seed, scale).eval()
    load_seeded(Cm, seed + 1, scale).eval()
    X, EI, EA = synthetic.to_torch(x, ei, ea, "cpu")
    default_threads = torch.get_num_threads()
    model, physical, hw_threads = host_cpu()
    cands = sorted({1} | {t for t in (8, 16, 32) if t <= hw_threads} | {min(physical, hw_threads)})
    trial = {}

    def timed_step():
        nonlocal EA
        t0 = time.perf_counter()
        _, EA = oracle.rollout_step(R, Cm, X, EI, EA, SPAN, centres=centres)
        return time.perf_counter() - t0

    try:
        for t in cands:     # one warm-up + one timed step per candidate: only to pick the thread count
            torch.set_num_threads(t)
            timed_step()
            trial[t] = timed_step()
        best = min(trial, key=trial.get)
        torch.set_num_threads(best)
        timed_step()
        t_start, steps = time.perf_counter(), []
        while len(steps) < 5 or (time.perf_counter() - t_start < budget_s and len(steps) < 50):
            steps.append(timed_step())
    finally:
        torch.set_num_threads(default_threads)
    med = float(np.median(steps))
    return {"value": round(1.0 / med, 4), "unit": "steps/s", "cores": best, "threads": best,
            "statistic": f"median of {len(steps)} consecutive steps (min {1 / max(steps):.3f}, max {1 / min(steps):.3f} steps/s)",
            "physical_cores": physical, "hardware_threads": hw_threads, "cpu_model": model,
            "all_physical_cores_steps_per_s": round(1 / trial[min(physical, hw_threads)], 4),
            "one_thread_steps_per_s": round(1 / trial[1], 4), "kind": "port",
            "sample": f"{len(steps)} steps of the same 10k-grain workload at the fastest of "
                      f"{{{', '.join(f'{t} threads: {1 / v:.2f} steps/s' for t, v in sorted(trial.items()))}}} "
                      "(one timed step each after a warm-up step; the all-physical-cores entry is SURVEY 8d's figure: "
                      "PyTorch's intra-op threading loses to its own synchronisation beyond ~16 threads on this workload)"}


def host_cpu():
    """(model name, physical cores, hardware threads) of this host from /proc/cpuinfo."""
    model, cores, phys, core = "unknown", set(), None, None
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                k, _, v = ln.partition(":")
                k, v = k.strip(), v.strip()
                if k == "model name":
                    model = v
                elif k == "physical id":
                    phys = v
                elif k == "core id":
                    core = v
                elif not k and phys is not None:
                    cores.add((phys, core))
                    phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    return model, (len(cores) or (os.cpu_count() or 1)), (os.cpu_count() or 1)


def visible_gpus():
    """GPUs this process would see, WITHOUT touching the HIP runtime (the parent of the rank processes must
    stay free of GPU state): the kfd topology's nodes with SIMDs, cut down by a HIP_/ROCR_VISIBLE_DEVICES
    list.  None when the topology cannot be read (the ranks then fail on their own if a device is missing)."""
    import glob
    n = 0
    try:
        for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
            with open(f) as fh:
                props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
    except (OSError, ValueError):
        return None
    if n == 0:
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([t for t in v.split(",") if t.strip() != ""]))
    return n


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` started plainly (no torchrun): start N rank processes ourselves,
    the way the reference does it (dist_train.py:394-395, mp.spawn(nprocs=device_count); :76-82).
    The parent has made NO GPU call at this point and makes none: the ranks are fresh child
    processes (never an exec of a GPU-initialised process); the parent relays rank 0's JSON line
    and exits non-zero when any rank fails."""
    import socket
    import subprocess
    visible = visible_gpus()
    if os.environ.get("GGNN_BENCH_BACKEND", "nccl") == "nccl" and visible is not None and visible < n:
        raise SystemExit(f"--gpus {n} but only {visible} GPU(s) are visible")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out)
    sys.stdout.flush()
    if any(codes):
        raise SystemExit(f"rank exit codes {codes}")
    line = json.loads(out.strip().splitlines()[-1])
    if line.get("n_gpus") != n:
        raise SystemExit(f"--gpus {n} but the job reports n_gpus = {line.get('n_gpus')}")


def train_step_records():
    """BASELINE config 5's kernel path beside the headline: one optimisation step of the regressor (forward, loss,
    backward, Adam) on the HIP training path, for a collated batch of four 40 um graphs (train.py:365-366's batch) and
    for the 10k-grain graph, eager and replayed from a hipGraph, with the reference's optimizer call
    (torch.optim.Adam(...): the foreach implementation, ~40 launches over 284 tensors), with Adam(fused=True) (14
    launches) and with training.FusedAdam (ggnn_adam_step: the same update in two launches); the cfg3 record of the
    fastest configuration carries live per-call roofline records of the weight-gradient GEMM, the sweep backward and
    the row GEMMs (`kernel_rooflines`).  Child processes (tests/bench_train_step.py); a failure leaves a string in
    the record and never touches the headline."""
    import subprocess
    script = os.path.join(ROOT, "tests", "bench_train_step.py")
    out = []
    for extra in (["--graph", "--ggnn-adam"], ["--graph", "--ggnn-adam", "--cfg3", "--roofline"],
                  ["--graph", "--ggnn-adam", "--cfg3", "--bf16"], ["--graph", "--ggnn-adam", "--cfg3", "--classifier"],
                  ["--graph", "--fused", "--cfg3"], ["--graph", "--cfg3"], ["--cfg3"]):
        try:
            r = subprocess.run([sys.executable, script, "--no-cpu", "--json", "--steps", "20"] + extra,
                               capture_output=True, text=True, timeout=300)
            out.append(json.loads(r.stdout.strip().splitlines()[-1]))
        except Exception as e:  # noqa: BLE001 -- informational record only
            out.append({"args": extra, "error": f"{type(e).__name__}: {e}"[:200]})
    return out


def event_mode_record():
    """SURVEY 8 f-2 beside the headline: a quiet step of the event-driven loop against a static step and an eventful step piece
    by piece (host-side rewiring, CSR rebuild, the rest), at the 10k-grain graph (tests/bench_event_step.py, child process;
    a failure leaves a string in the record and never touches the headline)."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_event_step.py")], capture_output=True, text=True,
                           timeout=420)
        try:
            return json.loads(r.stdout.strip().splitlines()[-1])
        except (ValueError, IndexError):
            return {"error": "no record", "stderr_tail": r.stderr[-400:]}
    except Exception as e:  # noqa: BLE001 -- informational record only
        return {"error": f"{type(e).__name__}: {e}"[:200]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile", action="store_true",
                    help="for rocprofv3 runs: only the rollout steps (no roofline timing, forward-only figure, "
                         "CPU baseline or native-fp32 child run), so every traced kernel belongs to a step")
    ap.add_argument("--joint", action="store_true", help="R and C in the same launches whatever the graph size")
    ap.add_argument("--serial", action="store_true", help="one set of launches per model, on one stream")
    ap.add_argument("--two-streams", action="store_true",
                    help="one set of launches per model, regressor and classifier on two streams (round-1 plan)")
    ap.add_argument("--events", action="store_true",
                    help="event-driven mode (SURVEY 8f-2): per-step event detection + host topology update when "
                         "one fires; not the headline metric")
    ap.add_argument("--events-quiet", action="store_true",
                    help="--events with thresholds no prediction reaches: what the event machinery costs on a quiet step")
    ap.add_argument("--workload", default="cfg3", choices=["cfg3", "cfg2", "cfg4", "gen368"],
                    help="cfg3 (default, the BASELINE metric): 10k-grain honeycomb; cfg2: the 120 um fixture "
                         "(1043 grains); cfg4: 64 perturbed 40 um trajectories sharded over the ranks, each "
                         "rank batching its shard as one disjoint-union graph (--steps = steps per trajectory); "
                         "gen368: the reference generator's own 368 um structure (graph_trajectory.py --mode=generate "
                         "--lxd=368: 9 775 grains / 19 550 junctions / 58 650 edges per type, nodes in Qhull order, "
                         "SURVEY 8(d)'s cross-check) -- the headline's size on a node numbering that is not a lattice's")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: ranks and --gpus must agree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # GGNN_BENCH_BACKEND=gloo + fewer GPUs than ranks: single-GPU smoke run of the N > 1 code path
    backend = os.environ.get("GGNN_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)  # RCCL over xGMI on ROCm
        else:
            dist.init_process_group(backend)

    units_per_step = world  # graph-steps completed per ro.step() over the whole job
    if args.workload == "cfg3":
        # every rank owns an independent replica (different seed of the junction noise per rank)
        R, Cm, X, EI, EA, inputs = build(device, seed=0)
        if rank:
            xp = synthetic.perturbed_copy(inputs[0], 1e-4, 1000 + rank)
            X = {k: torch.from_numpy(v).to(device) for k, v in xp.items()}
        workload = ("cfg3: synthetic periodic honeycomb, 10000 grains / 20000 junctions / 60000 edges per "
                    "type, fold 10, static topology, R+C forward + update + grain-centre refresh + edge "
                    "refresh per step, weights "
                    "RandomState(0) x0.3")
    elif args.workload == "gen368":
        R, Cm, X, EI, EA, inputs = build_generated(device)
        workload = ("gen368: synthetic.generate(lxd=368, seed=0) = the reference generator's sample (graph_trajectory.py "
                    f"--mode=generate --lxd=368), {X['grain'].size(0)} grains / {X['joint'].size(0)} junctions / "
                    f"{EI[('joint', 'connect', 'joint')].size(1)} edges per type, nodes and edges in the generator's (Qhull) order, "
                    "x9 patch folding, static topology, weights RandomState(0) x0.3; one replica per rank")
    else:
        from graingraphnn_amd.dist import shard_trajectories
        gold = os.path.join(ROOT, "tests", "golden")
        hp = synthetic.default_hyper(device)
        R = GrainNN_regressor(hp)
        Cm = GrainNN_classifier(hp, R)
        if args.workload == "cfg2":
            x, ei, ea = synthetic.load_fixture(os.path.join(gold, "graph_120.npz"))
            x, ea = {k: v.copy() for k, v in x.items()}, {k: v.copy() for k, v in ea.items()}
            factor, off = 3.0, synthetic.scale_feature_patchs(3.0, x, ea)
            load_seeded(R, 0).eval(), load_seeded(Cm, 1).eval()
            workload = "cfg2: 120 um fixture (1043 grains / 2086 junctions) after x3 patch folding, one replica per rank"
        else:
            factor, off = 1.0, None
            x0, ei0, ea0 = synthetic.load_fixture(os.path.join(gold, "graph_40.npz"))
            mine = shard_trajectories(64, rank, world)
            x, ei, ea, _ = synthetic.disjoint_union(
                [(synthetic.perturbed_copy(x0, 1e-3, 1000 + t), ei0, ea0) for t in mine])
            load_seeded(R, 10020).eval(), load_seeded(Cm, 10021).eval()
            units_per_step = 64
            workload = (f"cfg4: 64 perturbed 40 um trajectories (118 grains each), {len(mine)} per rank batched "
                        "as one disjoint-union graph")
        inputs = (x, ei, ea, factor, off)
        R, Cm = R.to(device), Cm.to(device)
        X, EI, EA = synthetic.to_torch(x, ei, ea, device)
    ro = GrainRollout(R, Cm, X, EI, EA, SPAN, use_graph=not args.no_graph, concurrent=not args.serial,
                      joint_launches=True if args.joint else False if (args.serial or args.two_streams) else None,
                      refresh_centres=True, domain_factor=inputs[3],
                      domain_offset=None if inputs[4] is None else torch.from_numpy(inputs[4]))

    step = ro.step
    args.events = args.events or args.events_quiet
    if args.events:
        n_g, n_j = X["grain"].size(0), X["joint"].size(0)
        ro.enable_events({"grain": np.ones((n_g, 1)), "joint": np.ones((n_j, 1))}, 1e-4, 0.6)
        from graingraphnn_amd.topology import TopologyError
        ev_state = {"stopped": None, "n": 0}
        if args.events_quiet:
            ro.area_threshold, ro.edge_threshold, ro._logit_trigger = -1.0, 2.0, 1e30
            ev_state["stopped"] = "events switched off from the start (--events-quiet)"

        def step():
            # random (untrained) weights eventually drive the graph into states the reference itself
            # asserts on (models.py:673); from then on the run continues with the events switched off
            ev_state["n"] += 1
            if ev_state["stopped"] is None:
                try:
                    return ro.step_events()
                except (TopologyError, IndexError, ValueError) as exc:
                    ev_state["stopped"] = f"step {ev_state['n']}: {type(exc).__name__}: {exc}"
                    ro.area_threshold, ro.edge_threshold, ro._logit_trigger = -1.0, 2.0, 1e30
            return ro.step_events()
    for _ in range(args.warmup):
        step()
    untimed_steps = args.warmup

    def run_ev(n):
        # the speculative event loop (GrainRollout.run_events: no host stall on a quiet step); a rejected event switches the
        # events off for the rest of the run, as the per-step wrapper above does during the warm-up
        target = ro.steps_done + n
        while ro.steps_done < target:
            try:
                ro.run_events(target - ro.steps_done)
            except (TopologyError, IndexError, ValueError) as exc:
                if ev_state["stopped"] is None:
                    ev_state["stopped"] = f"step {ro.steps_done}: {type(exc).__name__}: {exc}"
                ro.area_threshold, ro.edge_threshold, ro._logit_trigger = -1.0, 2.0, 1e30
    if args.events:
        run_ev(4 * ro.EVENTS_UNROLL)   # untimed: the block graphs are captured here
        untimed_steps += 4 * ro.EVENTS_UNROLL
    if not args.events:
        # steps per hipGraph: GrainRollout.RUN_UNROLL, the product's default (10: the timed region of --steps 20 / 500 is whole
        # replays of one captured graph); GGNN_BENCH_UNROLL overrides for A/B runs and is then named in config.launch
        ro.RUN_UNROLL = max(1, min(args.steps, int(os.environ.get("GGNN_BENCH_UNROLL", str(ro.RUN_UNROLL)))))
        # untimed: captures the multi-step graph and replays it a few times -- a freshly instantiated graph and a
        # memory system that has seen 5 steps run the first replays ~4 % slower than the steady state the metric
        # is about (measured with --steps 20: 1 872-1 889 steps/s without, 1 940-1 981 with a longer warm-up);
        # reported as config.untimed_steps
        settle = 1 + int(os.environ.get("GGNN_BENCH_SETTLE", "4"))
        ro.run(ro.RUN_UNROLL * settle)
        untimed_steps = args.warmup + ro.RUN_UNROLL * settle
    gather_states(ro.state(), world)  # warm-up of the collective too (communicator set-up is lazy)
    if not args.events and os.environ.get("GGNN_BENCH_PREROLL", "1") != "0":
        # the first gather sets up host buffers (and the communicator) while the GPU idles for milliseconds; one more
        # untimed replay + gather puts the device where every later region finds it (the first 20-step region measured
        # 4-5 % below its own back-to-back repeats without it, on every box); counted in config.untimed_steps
        ro.run(ro.RUN_UNROLL)
        gather_states(ro.state(), world)
        untimed_steps += ro.RUN_UNROLL
        if world > 1:
            # ... and a rehearsal of the bracket's collectives (barrier, the MAX all-reduce of the times): their first calls
            # set up connections (measured with two gloo ranks on one GPU: the first region took 84 ms, its repeats 15)
            torch.cuda.synchronize()
            dist.barrier()
            t = torch.zeros(1, dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.barrier()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if args.events:
        run_ev(args.steps)
    else:
        ro.run(args.steps)  # exactly args.steps steps; hipGraph replay goes 4 steps per graph launch
    gathered = gather_states(ro.state(), world)  # RCCL all-gather of the rollout results
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    finite = all(bool(torch.isfinite(v).all()) for g in gathered for v in g.values())
    # The same region -- exactly args.steps steps + the gather, bracketed the same way -- repeated back to back: `value` /
    # `ms_per_step` stay the single shot above (what the driver's clock brackets), the median of the repeats says how
    # much of a 7 ms region's number is the region (GGNN_BENCH_REPEATS, default 15; 0 switches it off)
    repeats = []
    for _ in range(0 if args.events else int(os.environ.get("GGNN_BENCH_REPEATS", "15"))):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        tr = time.perf_counter()
        ro.run(args.steps)
        gather_states(ro.state(), world)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dtr = time.perf_counter() - tr
        if world > 1:
            t = torch.tensor([dtr], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtr = float(t.item())
        repeats.append(units_per_step * args.steps / dtr)

    if rank == 0:
        roof, roof_enc, roof_gemm, roof_sweep = (None, None, [], None) if args.profile else measure_roofline(ro, 10)
        # forward-only rate beside the full step (eager launches, R then C on one stream)
        forward_only = 0.0
        with torch.no_grad():
            for _ in range(0 if args.profile else 5):
                R(X, EI, EA), Cm(X, EI, EA)
            torch.cuda.synchronize()
            tf = time.perf_counter()
            for _ in range(0 if args.profile else 100):
                R(X, EI, EA), Cm(X, EI, EA)
            torch.cuda.synchronize()
            forward_only = 100 / (time.perf_counter() - tf)
        line = {
            "metric": "rollout steps/sec (10k-grain heterograph)",
            "value": round(units_per_step * args.steps / dt, 2),
            "unit": "steps/s",
            "n_gpus": world,
            "rccl_ranks": dist.get_world_size() if world > 1 else 1,
            "collective_backend": (backend + (" (RCCL " + ".".join(map(str, torch.cuda.nccl.version())) + ")"
                                              if backend == "nccl" else "")) if world > 1 else None,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "value_median_of_repeats": round(float(np.median(repeats)), 2) if repeats else None,
            "repeats": {"n": len(repeats), "min": round(min(repeats), 2), "max": round(max(repeats), 2),
                        "what": "the timed region (steps + gather, same bracket) run again back to back"} if repeats else None,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload, "replicas": world, "launch": ("eager" if args.no_graph else f"hipGraph replay ({ro.RUN_UNROLL} steps per graph)") + (", R+C in the same launches" if ro.joint_launches else
                                  ", R|C on two streams" if ro.concurrent else ", R then C on one stream"),
                       "decoder_plan": (("one kernel per decoder cell (ggnn_decoder_cell_batch)"
                                         if X["joint"].size(0) >= getattr(default_backend(), "fused_decoder_min_joints", 0) else
                                         f"projection + sweeps + gate GEMM: fewer than {default_backend().fused_decoder_min_joints} junctions "
                                         "(the fused cell is the plan of the larger graphs; GGNN_DEC=fused forces it)")
                                        if getattr(default_backend(), "fused_decoder", False) is True else
                                        f"fused_decoder={getattr(default_backend(), 'fused_decoder', False)!r}: projection + sweeps + gate "
                                        "GEMM where not fused (GGNN_DEC)"),
                       "gemm": ("fp32 operands as exact pieces on the matrix cores, fp32 accumulate: fused decoder cell and encoder "
                                "gate GEMM 2 fp16 pieces / 3 products per k-step (error vs fp64 5e-8 of sum|x||w|, numpy emulation); "
                                "value projection of the default plan the same 3 products (3.4e-7 measured on the GPU); the three-kernel "
                                "plan's projection and gate GEMM 3 bf16 pieces / 6 products (2.7e-7 .. 8e-7 measured; native fp32 "
                                "MFMA chain 7.0e-7)"
                                if default_backend().lib.ggnn_gemm_mode() == 1 else "native fp32 MFMA (GGNN_GEMM=fp32)"),
                       "results_finite": finite, "untimed_steps": untimed_steps,
                       "forward_only": {"steps_per_s_per_gpu": round(forward_only * (units_per_step // world), 2),
                                        "launch": "eager, R then C on one stream (no update / refresh)"},
                       **({"events": {"grains_eliminated": int(sum(len(e) for e in ro.grain_events)),
                                      "edges_switched": int(sum(len(e) for e in ro.switched)),
                                      "edges_left": int(ro.edge_index[("joint", "connect", "joint")].size(1)),
                                      "stopped": ev_state["stopped"]}}
                          if args.events else {})},
            "roofline": roof,
            "roofline_split_sweep": roof_sweep,
            "roofline_encoder_cell": roof_enc,
            "roofline_gemm": roof_gemm,
        }
        if not args.no_cpu_baseline and not args.profile and args.workload == "cfg3" and world == 1:  # N = 1 only: a reported baseline
            line["cpu_baseline"] = cpu_baseline(inputs)
            if default_backend().lib.ggnn_gemm_mode() == 1 and not args.events:
                # the same workload with the decoder GEMMs on the native fp32 matrix path, in a child
                # process (the mode is fixed per process), so both arithmetic paths sit in one line
                import subprocess
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--steps", "200", "--warmup", "10",
                                    "--no-cpu-baseline"], env=dict(os.environ, GGNN_GEMM="fp32"),
                                   capture_output=True, text=True, timeout=600)
                try:
                    line["config"]["native_fp32_gemm_steps_per_s"] = json.loads(r.stdout.strip().splitlines()[-1])["value"]
                except (ValueError, IndexError, KeyError):
                    line["config"]["native_fp32_gemm_steps_per_s"] = None
            if not args.events:
                line["train_step"] = train_step_records()
                line["event_mode"] = event_mode_record()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
