#!/usr/bin/env python3
"""Development aid: time ggnn_lstm_epilogue alone (decoder, G = 4) and print the in-kernel stamps of
a GX_VAR_CLOCK build (tools/build_variant.sh gx_clock -DGX_VAR_CLOCK)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from graingraphnn_amd import _lib
from graingraphnn_amd.backend import default_backend
from graingraphnn_amd.packing import bf16_planes

be = default_backend()
dev = "cuda"
for N, Ka, G in ((20000, 196, 4), (10000, 100, 4), (20000, 196, 3), (10000, 100, 3)):
    gs = int(os.environ.get("GX_PAD", 0)) and ((Ka + 31) // 32 * 32)
    agg = torch.randn(N, G * (gs or Ka), device=dev)
    w2 = torch.randn(G, 96, Ka, device=dev) * 0.1
    pd = torch.randn(N, G * 96, device=dev)
    c_in = torch.randn(N, 96, device=dev)
    h, c = torch.empty(N, 96, device=dev), torch.empty(N, 96, device=dev)
    pl = bf16_planes(w2)
    for _ in range(5):
        be.lstm_epilogue(agg, w2, pd, 0, c_in if G == 4 else None, h, c, None, G, _lib.MODE_LSTM if G == 4 else _lib.MODE_LSTM_H0, pl, gs)
    ts = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); be.lstm_epilogue(agg, w2, pd, 0, c_in if G == 4 else None, h, c, None, G, _lib.MODE_LSTM if G == 4 else _lib.MODE_LSTM_H0, pl, gs); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    extra = ""
    if os.environ.get("GX_CLOCK2"):
        torch.cuda.synchronize()
        v = c.view(-1)[:240 * 8 * 4].view(240, 8, 4).cpu().numpy().astype(np.float64)
        st = v[:, :, 0]
        st = (st - st.min()) % (1 << 24)
        en = st + v[:, :, 2]
        pro = v[:, :, 1]
        extra = (f"\n   wave start: p50 {np.median(st) / 100:.1f} max {st.max() / 100:.1f} us; prologue p50 {np.median(pro) / 100:.1f} "
                 f"max {pro.max() / 100:.1f} us; lifetime p50 {np.median(v[:, :, 2]) / 100:.1f} min {v[:, :, 2].min() / 100:.1f} "
                 f"max {v[:, :, 2].max() / 100:.1f} us; last end {en.max() / 100:.1f} us")
    if os.environ.get("GX_CLOCK"):
        v = h.view(-1)[:15 * 8].view(15, 8).tolist()
        extra = "".join(f"\n   wave {w:2d}: total {t:6.0f} cyc = {r / 100:5.1f} us; prologue {pro:6.0f}, sweep {sw:6.0f}, "
                        f"stage {st:6.0f}, barrier {ba:6.0f}" for w, (t, r, sw, st, ba, pro, _, _) in enumerate(v))
    print(f"N={N} Ka={Ka} G={G}: med {np.median(ts):6.1f} us  min {min(ts):6.1f} us{extra}")
