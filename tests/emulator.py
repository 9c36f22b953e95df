"""Torch (CPU) emulation of the C-ABI calls in include/ggnn.h -- TEST INFRASTRUCTURE.

It restates, call by call, what each `ggnn_*` entry point is documented to compute, so that
the CPU test-suite can push the product's host logic (weight packing, column offsets, launch
plan of graingraphnn_amd/engine.py) end-to-end against the oracle without a GPU.  It is not
shipped, not importable from the package, and never used as a fallback: the product's only
backend is `graingraphnn_amd.backend.HipBackend`.
"""
import torch

C = 96


class TorchEmulatorBackend:
    name = "torch-emulator (tests only)"
    fused_encoder = True  # engine.run_cells: encoder cells through encoder_cell_batch (False: sweep + gate epilogue)
    value_rows_block_major = True   # the fused decoder plan's value rows as [blocks][N][96], like HipBackend's default

    def build_csr(self, edge_index, n_src, n_dst):
        src, dst = edge_index[0], edge_index[1]
        if ((src < 0) | (src >= n_src) | (dst < 0) | (dst >= n_dst)).any():
            raise IndexError("edge_index out of range")
        E = edge_index.size(1)
        # stable sort by destination == ascending original edge id inside each row
        perm = torch.sort(dst * (E + 1) + torch.arange(E), stable=True).indices
        rowptr = torch.zeros(n_dst + 1, dtype=torch.int64)
        rowptr[1:] = torch.bincount(dst, minlength=n_dst).cumsum(0)
        from graingraphnn_amd.backend import CSR
        return CSR(rowptr.int(), src[perm].int(), perm.int(), dst[perm].int(), None, None, E)

    def edge_prepare(self, items):
        for csr, ea, xs, xd, einfo in items:
            col, perm, row = csr.col.long(), csr.perm.long(), csr.row.long()
            E = ea.numel()
            rel = xs[col[:E], :3] - xd[row[:E], :3]
            reloc = torch.where(rel > 0.5, -1.0, torch.where(rel < -0.5, 1.0, 0.0)) + rel
            Fs = xs.size(1)
            einfo[:E] = 0.0
            einfo[:E, 0:3] = einfo[:E, 16:19] = reloc
            einfo[:E, 3:Fs] = xs[col[:E], 3:Fs]
            einfo[:E, 12] = 1.0
            einfo[:E, 13] = einfo[:E, 19] = ea[perm[:E]]
            if xs.size(1) <= 11:
                einfo[:E, 11] = 1.0

    def project(self, x, F, h, wp, bp, out):
        Fp = (F + 3) & ~3
        k2 = 0 if h is None else h.size(1)
        assert wp.size(1) == Fp + k2
        xin = torch.zeros(x.size(0), Fp + k2)
        xin[:, :F] = x[:, :F]
        if k2:
            xin[:, Fp:] = h
        out[:, :wp.size(0)] = xin @ wp.t() + bp

    def aggregate_batch(self, sweeps):
        for sweep in sweeps:
            self.aggregate(*sweep)

    def aggregate_enc_batch(self, sweeps):
        """ggnn_period_gat_aggregate_enc_batch: values relu(Bp^T x4[:12]) from the edge records."""
        for csr, einfo, p_dst, wvf, agg, u4_off, a_off, a_gstride, sc_off, G in sweeps:
            rowptr = csr.rowptr.long()
            n_dst, E = p_dst.size(0), int(rowptr[-1])
            dst = torch.repeat_interleave(torch.arange(n_dst), rowptr[1:] - rowptr[:-1])
            fr = wvf.view(G, 3, 2, 3, 4, 16)                       # g m2 e s kq j
            Bp = fr.permute(3, 4, 0, 1, 5, 2).reshape(12, G * C)   # k = 4 s + kq, column = g*96 + 32 m2 + 2 j + e
            x4, a = einfo[:E, :16], einfo[:E, 19]
            assert E == 0 or (bool((x4[:, 11] == 1).all()) and bool((x4[:, 12] == 1).all()))
            val = torch.relu(x4[:, :12] @ Bp)                    # [E, G * 96]
            for g in range(G):
                s = (p_dst[dst, u4_off + g * 16: u4_off + (g + 1) * 16] * x4).sum(-1)
                smax = torch.full((n_dst,), float("-inf")).scatter_reduce(0, dst, s, "amax")
                p = (s - smax[dst]).exp()
                den = torch.zeros(n_dst).index_add(0, dst, p)
                alpha = p / (den[dst] + 1e-16)
                base = g * a_gstride
                agg[:, base + a_off: base + a_off + C] = torch.zeros(n_dst, C).index_add(
                    0, dst, alpha[:, None] * val[:, g * C:(g + 1) * C])
                agg[:, base + sc_off] = torch.zeros(n_dst).index_add(0, dst, alpha)
                agg[:, base + sc_off + 1] = torch.zeros(n_dst).index_add(0, dst, alpha * a)

    def encoder_cell_batch(self, problems):
        """ggnn_encoder_cell_batch (include/ggnn.h): per gate (i, c~, o) and incoming edge type the score tails
        u4 = T [x | 1], the sweep with values relu(V record) from the edge records, lin_l2 + (b_l2, w_edge) on the
        aggregates; then the skip block and the LSTM update from the zero state.  Everything is computed from the
        DECODED weight stream (16-slot k-steps un-spread, lin_l2's columns un-permuted as the header states), so a
        packing error shows up here."""
        self.calls = getattr(self, "calls", []) + ["encoder_cell_batch"]
        p3 = torch.tensor([32 * (k // 32) + 16 * ((k % 8) // 4) + 4 * ((k % 32) // 8) + k % 4 for k in range(C)])

        def unspread(blk):   # [rows, 32] -> [rows, 16 slots]: k = 8 q + j holds slot 4 q + j for j < 4, zero for j >= 4
            v = blk.view(-1, 4, 8)
            assert not bool(v[:, :, 4:].any())
            return v[:, :, :4].reshape(-1, 16)

        for sweeps, x_dst, wstream, w2_tail, h_out, c_out, *_ in problems:
            n, n_in, F = x_dst.size(0), len(sweeps), x_dst.size(1)
            from graingraphnn_amd.packing import DC_SLICE_I16
            assert wstream.numel() == 3 * (4 * n_in + 1) * DC_SLICE_I16 and tuple(w2_tail.shape) == (3, n_in, 6, 64)
            xs = torch.zeros(n, 16)
            xs[:, :F], xs[:, 12] = x_dst[:, :F], 1.0
            pre, s = {}, 0
            for g in range(3):
                z = torch.zeros(n, C)
                for d, (csr, einfo) in enumerate(sweeps):
                    A = unspread(self._decode_slices(wstream, s, 1, 7))      # [112, 16]: value rows | u4 rows
                    W3p = self._decode_slices(wstream, s + 1, 3, 6)          # [96, 96], columns permuted
                    s += 4
                    W3 = torch.empty_like(W3p)
                    W3[:, p3] = W3p                                          # column k of the block = lin_l2 column of channel p3[k]
                    V, T = A[:C], A[C:]
                    u4 = xs @ T.t()                                           # [n, 16]
                    rowptr = csr.rowptr.long()
                    E = int(rowptr[-1])
                    dst = torch.repeat_interleave(torch.arange(n), rowptr[1:] - rowptr[:-1])
                    x4 = einfo[:E, :16]
                    a = x4[:, 13]                                             # the edge length, where the kernel reads it
                    sc = (u4[dst] * x4).sum(-1)
                    smax = torch.full((n,), float("-inf")).scatter_reduce(0, dst, sc, "amax")
                    p = (sc - smax[dst]).exp()
                    den = torch.zeros(n).index_add(0, dst, p)
                    alpha = p / (den[dst] + 1e-16)
                    val = torch.relu(x4 @ V.t())
                    Ag = torch.zeros(n, C).index_add(0, dst, alpha[:, None] * val)
                    sa = torch.zeros(n).index_add(0, dst, alpha)
                    sae = torch.zeros(n).index_add(0, dst, alpha * a)
                    tail = w2_tail[g, d].view(6, 4, 16)                       # ct k m: k = 0 b_l2, k = 3 w_edge
                    assert not bool(tail[:, 1:3].any())
                    z = z + Ag @ W3.t() + sa[:, None] * tail[:, 0].reshape(1, C) + sae[:, None] * tail[:, 3].reshape(1, C)
                Sg = unspread(self._decode_slices(wstream, s, 1, 6))         # [96, 16]
                s += 1
                pre[g] = z + xs @ Sg.t()
            c = torch.sigmoid(pre[0]) * torch.tanh(pre[1])
            c_out.copy_(c)
            h_out.copy_(torch.sigmoid(pre[2]) * torch.tanh(c))

    @staticmethod
    def _aggregate_values(csr, einfo, p_src, p_dst, h_src, ep, v_off, u_off, u4_off, n_gates):
        """Per gate (values [n_dst, 96], sum alpha [n_dst], sum alpha * a [n_dst]); differentiable."""
        rowptr, col = csr.rowptr.long(), csr.col.long()
        n_dst = p_dst.size(0)
        E = int(rowptr[-1])
        dst = torch.repeat_interleave(torch.arange(n_dst), rowptr[1:] - rowptr[:-1])
        j, reloc, a = col[:E], einfo[:E, 16:19], einfo[:E, 19]
        x4 = einfo[:E, :16]  # the per-edge 16-wide tail: reloc, raw features 3..F-1, zeros, 1 @12, a_e @13
        out = []
        for g in range(n_gates):
            V = p_src[j, v_off + g * C: v_off + (g + 1) * C]
            s = (p_dst[dst, u4_off + g * 16: u4_off + (g + 1) * 16] * x4).sum(-1)
            if h_src is not None:
                s = s + (p_dst[dst, u_off + g * C: u_off + (g + 1) * C] * h_src[j]).sum(-1)
            smax = torch.full((n_dst,), float("-inf")).scatter_reduce(0, dst, s.detach(), "amax")
            p = (s - smax[dst]).exp()
            den = torch.zeros(n_dst).index_add(0, dst, p)
            alpha = p / (den[dst] + 1e-16)
            r = torch.relu(V + reloc @ ep[g])
            out.append((torch.zeros(n_dst, C).index_add(0, dst, alpha[:, None] * r),
                        torch.zeros(n_dst).index_add(0, dst, alpha),
                        torch.zeros(n_dst).index_add(0, dst, alpha * a)))
        return out

    def aggregate(self, csr, einfo, p_src, p_dst, h_src, ep, agg, v_off, u_off, u4_off, a_off,
                  a_gstride, sc_off, n_gates, pad_n=0):
        with torch.no_grad():
            vals = self._aggregate_values(csr, einfo, p_src, p_dst, h_src, ep, v_off, u_off, u4_off, n_gates)
        for g, (val, sa, sae) in enumerate(vals):
            base = g * a_gstride
            agg[:, base + a_off: base + a_off + C] = val
            agg[:, base + sc_off] = sa
            agg[:, base + sc_off + 1] = sae
            agg[:, base + sc_off + 2: base + sc_off + 2 + pad_n] = 0.0   # (ggnn_aggregate_args.pad_n)

    def aggregate_backward(self, csr, rcsr, r_slot, einfo, p_src, p_dst, h_src, ep, agg, g_agg,
                           v_off, u_off, u4_off, a_off, a_gstride, sc_off, n_gates, out_p_dst=None, out_p_src=None,
                           ep_partial_out=None, g_h_into=None):
        """ggnn_period_gat_aggregate_backward by autograd of the emulated forward.  The reverse
        CSR is checked for what the HIP kernel relies on, then not needed."""
        E = csr.E
        if E:
            assert torch.equal(csr.col[:E].long()[r_slot[:E].long()],
                               torch.repeat_interleave(torch.arange(p_src.size(0)),
                                                       (rcsr.rowptr[1:] - rcsr.rowptr[:-1]).long()))
            assert torch.equal(csr.row[:E].long()[r_slot[:E].long()], rcsr.col[:E].long())
        leaves = [t.detach().clone().requires_grad_(True) for t in (p_src, p_dst, ep)]
        hl = None if h_src is None else h_src.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            vals = self._aggregate_values(csr, einfo, leaves[0], leaves[1], hl, leaves[2], v_off, u_off, u4_off, n_gates)
            tot = 0.0
            for g, (val, sa, sae) in enumerate(vals):
                base = g * a_gstride
                tot = tot + (val * g_agg[:, base + a_off: base + a_off + C]).sum() \
                    + (sa * g_agg[:, base + sc_off]).sum() + (sae * g_agg[:, base + sc_off + 1]).sum()
            grads = torch.autograd.grad(tot, leaves + ([hl] if hl is not None else []), allow_unused=True)
        z = lambda g, like: torch.zeros_like(like) if g is None else g
        g_p_dst, g_p_src = z(grads[1], p_dst), z(grads[0], p_src)
        if out_p_dst is not None:   # shared buffers: this sweep WRITES its own columns only, as the kernel does
            cols = [(u4_off, n_gates * 16)] + ([(u_off, n_gates * C)] if hl is not None else [])
            for o, w in cols:
                out_p_dst[:, o:o + w] = g_p_dst[:, o:o + w]
            g_p_dst = out_p_dst
        if out_p_src is not None:
            out_p_src[:, v_off:v_off + n_gates * C] = g_p_src[:, v_off:v_off + n_gates * C]
            g_p_src = out_p_src
        g_ep = z(grads[2], ep)
        if ep_partial_out is not None:  # all of it in the first partial, zeros in the others the kernel would write
            n_part = self.aggregate_bwd_partials(p_dst.size(0))
            ep_partial_out[:n_part] = 0.0
            ep_partial_out[0] = g_ep
            g_ep = None
        g_h = None if hl is None else z(grads[3], h_src)
        if g_h_into is not None:   # (g_h_accumulate: added in place to the earlier sweep's rows)
            g_h = g_h_into.add_(g_h)
        return (g_p_dst, g_p_src, g_h, g_ep)

    @staticmethod
    def aggregate_bwd_partials(n_dst):
        want = (n_dst + 3) // 4
        return min(max(want, 1), 768)

    fused_decoder = True  # engine.run_cells: decoder cells through decoder_cell_batch (False: projection + sweeps + gates)

    @staticmethod
    def _decode_slices(stream, first, n_ks, n_tiles):
        """`n_ks` consecutive slices of ggnn_dec_cell_args.wstream (include/ggnn.h) -> the fp32 block
        [16 n_tiles, 32 n_ks] they hold: hi + lo' / 2^11 of [column tile][plane][lane 16 kq + m][8 fp16]."""
        from graingraphnn_amd.packing import DC_LO_SCALE, DC_PLANES, DC_SLICE_I16
        S, P = DC_SLICE_I16, DC_PLANES
        sl = stream.view(-1, S)[first:first + n_ks, :n_tiles * P * 64 * 8]
        assert not bool(stream.view(-1, S)[first:first + n_ks, n_tiles * P * 64 * 8:].any())
        fr = sl.reshape(n_ks, n_tiles, P, 4, 16, 8).view(torch.float16).float()            # ks nb p kq m j
        fr = fr[:, :, 0] + fr[:, :, 1] / DC_LO_SCALE
        return fr.permute(1, 3, 0, 2, 4).reshape(16 * n_tiles, 32 * n_ks)                  # (nb m) x (ks kq j)

    def decoder_cell_batch(self, problems):
        """ggnn_decoder_cell_batch: per gate (stream order i, c~, f, o) and incoming edge type the score operands
        u_h | u4 = W1 [h | x | 1], the sweep, lin_l2 + (b_l2, w_edge) on the aggregates; then the skip block and the
        LSTM update.  Everything is computed from the DECODED weight stream, so a packing error shows up here."""
        self.calls = getattr(self, "calls", []) + ["decoder_cell_batch"]   # (which plan ran: test_decoder_plan_per_model)
        for sweeps, x_dst, h_dst, c_in, wstream, w2_tail, h_out, c_out, *_ in problems:
            n, n_in, F = x_dst.size(0), len(sweeps), x_dst.size(1)
            from graingraphnn_amd.packing import DC_SLICE_I16
            assert wstream.numel() == 4 * (7 * n_in + 4) * DC_SLICE_I16 and tuple(w2_tail.shape) == (4, n_in, 6, 64)
            xin = torch.zeros(n, 128)
            xin[:, :C], xin[:, C:C + F], xin[:, C + F] = h_dst, x_dst, 1.0
            pre, s = {}, 0
            for gi, g in enumerate((0, 2, 1, 3)):
                z = torch.zeros(n, C)
                order = list(enumerate(sweeps))
                for d, (csr, einfo, h_src, v_src, v_off, ep, *v_bm) in (order[::-1] if gi & 1 else order):   # ggnn.h: backwards for the 2nd and 4th gate
                    W1 = self._decode_slices(wstream, s, 4, 7)        # [112, 128]
                    W3 = self._decode_slices(wstream, s + 4, 3, 6)    # [96, 96]
                    s += 7
                    assert not bool(W1[:, C + F + 1:].any())
                    u = xin @ W1.t()                                   # [n, 112]: u_h | u4
                    rowptr = csr.rowptr.long()
                    E = int(rowptr[-1])
                    dst = torch.repeat_interleave(torch.arange(n), rowptr[1:] - rowptr[:-1])
                    src = csr.col.long()[:E]
                    x4, reloc, a = einfo[:E, :16], einfo[:E, 16:19], einfo[:E, 19]
                    sc = (u[dst, :C] * h_src[src, :C]).sum(-1) + (u[dst, C:] * x4).sum(-1)
                    smax = torch.full((n,), float("-inf")).scatter_reduce(0, dst, sc, "amax")
                    p = (sc - smax[dst]).exp()
                    den = torch.zeros(n).index_add(0, dst, p)
                    alpha = p / (den[dst] + 1e-16)
                    if v_bm and v_bm[0]:   # GGNN_OUT_BLOCK_MAJOR: block v_off / 96 + g of [blocks][n_src][96]
                        n_src = v_src.size(0)
                        b0 = (v_off // C + g) * n_src * C
                        v_g = v_src.reshape(-1)[b0:b0 + n_src * C].view(n_src, C)[src]
                    else:
                        v_g = v_src[src, v_off + g * C: v_off + (g + 1) * C]
                    val = torch.relu(v_g + reloc @ ep[g])
                    A = torch.zeros(n, C).index_add(0, dst, alpha[:, None] * val)
                    sa = torch.zeros(n).index_add(0, dst, alpha)
                    sae = torch.zeros(n).index_add(0, dst, alpha * a)
                    tail = w2_tail[g, d].view(6, 4, 16)                # ct k m
                    assert not bool(tail[:, 2:].any())
                    z = z + A @ W3.t() + sa[:, None] * tail[:, 0].reshape(1, C) + sae[:, None] * tail[:, 1].reshape(1, C)
                W4 = self._decode_slices(wstream, s, 4, 6)             # [96, 128]
                s += 4
                pre[g] = z + xin @ W4.t()
            i, f, t, o = torch.sigmoid(pre[0]), torch.sigmoid(pre[1]), torch.tanh(pre[2]), torch.sigmoid(pre[3])
            c = f * c_in + i * t
            c_out.copy_(c)
            h_out.copy_(o * torch.tanh(c))

    def lstm_epilogue(self, agg, w2, p_dst, s_off, c_in, h_out, c_out, raw_out, n_gates, mode,
                      w2_planes=None, g_stride=0):
        Ka = w2.size(2)
        gs = g_stride or Ka
        if w2_planes is not None:  # the planes must reassemble to w2[:, :, :Ka-4] exactly (ggnn.h layout)
            G, KM = w2.size(0), Ka - 4
            fr = w2_planes.view(torch.float32).view(G, KM // 32, 6, 2, 4, 16, 4)  # g ks ct h kq i j
            back = fr.permute(0, 2, 5, 1, 4, 3, 6).reshape(G, 96, KM)
            assert torch.equal(back, w2[:, :, :KM])
        pre = [agg[:, g * gs:g * gs + Ka] @ w2[g].t() + p_dst[:, s_off + g * C: s_off + (g + 1) * C]
               for g in range(n_gates)]
        if mode == 2:
            raw_out.copy_(torch.cat(pre, 1))
        elif mode == 0:
            i, f, t, o = torch.sigmoid(pre[0]), torch.sigmoid(pre[1]), torch.tanh(pre[2]), torch.sigmoid(pre[3])
            c = f * c_in + i * t
            c_out.copy_(c)
            h_out.copy_(o * torch.tanh(c))
        else:
            i, t, o = torch.sigmoid(pre[0]), torch.tanh(pre[1]), torch.sigmoid(pre[2])
            c = i * t
            c_out.copy_(c)
            h_out.copy_(o * torch.tanh(c))

    def lstm_train_forward(self, z, p_dst, s_off, c_in, h_out, c_out):
        G = z.size(0)
        for g in range(G):
            z[g] += p_dst[:, s_off + g * C: s_off + (g + 1) * C]
        c = torch.sigmoid(z[0]) * torch.tanh(z[G - 2])
        if G == 4:
            c = c + torch.sigmoid(z[1]) * c_in
        c_out.copy_(c)
        h_out.copy_(torch.sigmoid(z[G - 1]) * torch.tanh(c))

    def lstm_train_backward(self, z, c_in, c_out, g_h, g_c, g_z, g_p_dst, s_off, g_c_in):
        G = z.size(0)
        gh = torch.zeros_like(c_out) if g_h is None else g_h
        gc = torch.zeros_like(c_out) if g_c is None else g_c
        o, tc = torch.sigmoid(z[G - 1]), torch.tanh(c_out)
        g_z[G - 1] = gh * tc * o * (1 - o)
        dc = gc + gh * o * (1 - tc * tc)
        i, ct = torch.sigmoid(z[0]), torch.tanh(z[G - 2])
        g_z[0] = dc * ct * i * (1 - i)
        g_z[G - 2] = dc * i * (1 - ct * ct)
        if G == 4:
            f = torch.sigmoid(z[1])
            g_z[1] = dc * c_in * f * (1 - f)
            if g_c_in is not None:
                g_c_in.copy_(dc * f)
        if g_p_dst is not None:
            for g in range(G):
                g_p_dst[:, s_off + g * C: s_off + (g + 1) * C] = g_z[g]

    def lstm_train_forward_batch(self, problems, n_gates):
        for p in problems:
            assert p[0].size(0) == n_gates
            self.lstm_train_forward(*p)

    def lstm_train_backward_batch(self, problems, n_gates):
        for (z, c_in, c_out, g_h, g_c, g_z, g_p_dst, s_off, g_c_in, pad_off, pad_n) in problems:
            assert z.size(0) == n_gates and pad_n % 4 == 0 and pad_off % 4 == 0 and 0 <= pad_n <= C
            self.lstm_train_backward(z, c_in, c_out, g_h, g_c, g_z, g_p_dst, s_off, g_c_in)
            if pad_n:
                g_p_dst[:, pad_off:pad_off + pad_n] = 0.0

    @staticmethod
    def train_input_rows(problems):
        outs = []
        for x, F in problems:
            Fp = (F + 3) & ~3
            out = torch.zeros(x.size(0), Fp + 4)
            out[:, :F] = x[:, :F]
            out[:, Fp] = 1.0
            outs.append(out)
        return outs

    def wgrad(self, a, b, K, M, Nc, lda, ldb, batch=1, a_bstride=0, b_bstride=0, b_ins=None, ins_off=0, defer=None):
        if b_ins is not None:   # (ggnn_wgrad_args.b_ins: the columns of b_ins inserted into b at ins_off)
            assert batch == 1 and ins_off % 4 == 0 and b_ins.size(1) % 4 == 0 and b_ins.is_contiguous()
            bm = torch.as_strided(b.reshape(-1), (K, Nc - b_ins.size(1)), (ldb, 1), b.reshape(-1).storage_offset())
            b = torch.cat([bm[:, :ins_off], b_ins[:K], bm[:, ins_off:]], 1).contiguous()
            ldb = Nc
        fa, fb = a.reshape(-1), b.reshape(-1)   # (as_strided offsets are absolute in the storage)
        oa, ob = fa.storage_offset(), fb.storage_offset()
        return torch.stack([torch.as_strided(fa, (K, M), (lda, 1), oa + k * a_bstride).t()
                            @ torch.as_strided(fb, (K, Nc), (ldb, 1), ob + k * b_bstride) for k in range(batch)])

    @staticmethod
    def sum_rows(t):
        return t.sum(1)

    def rowgemm(self, a, w, out, K, n_out, batch=1, c_in=None, transposed=False, bf16=False, planes=None):
        """ggnn_rowgemm: out[b][:, :n_out] = a[b][:, :K] W[b]^T (+ c_in[b]); W[b] = w[b][:n_out, :K] or, transposed,
        w[b][:K, :n_out]^T.  (bf16: the product of bf16-rounded operands, fp32 accumulation.)"""
        a3 = a if a.dim() == 3 else a.unsqueeze(0)
        w3 = w if w.dim() == 3 else w.unsqueeze(0)
        o3 = out if out.dim() == 3 else out.unsqueeze(0)
        c3 = None if c_in is None else (c_in if c_in.dim() == 3 else c_in.unsqueeze(0))
        for b in range(batch):
            W = w3[b][:K, :n_out].t() if transposed else w3[b][:n_out, :K]
            A = a3[b][:, :K]
            if bf16:
                A, W = A.bfloat16().float(), W.bfloat16().float()
            res = A @ W.t()
            if c3 is not None:
                res = res + c3[b][:, :n_out]
            o3[b][:, :n_out] = res
        return out

    @staticmethod
    def f16_projection():
        return True   # (GGNN_PRECISION_F16X2 is fp32-equivalent: the emulation computes it in fp32)

    def project_batch(self, problems):
        """(x, F, h, wp, bp, out[, precision]); precision GGNN_PRECISION_BF16: both operands rounded to bf16, products
        accumulated in fp32 (what the HIP kernel's single-product mode computes)."""
        for x, F, h, wp, bp, out, *rest in problems:
            prec = rest[0] if rest else 0
            target = out
            if prec & 0x100:   # GGNN_OUT_BLOCK_MAJOR: [ncols / 96][M][96] at the front of out's storage
                target = torch.empty(x.size(0), wp.size(0))
            if prec & 0xff == 1:
                r = lambda t: None if t is None else t.to(torch.bfloat16).float()
                self.project(r(x), F, r(h), r(wp), bp, target)
            else:
                self.project(x, F, h, wp, bp, target)
            if prec & 0x100:
                M, nc = target.shape
                base = out._base if out._base is not None else out
                base.view(-1)[:M * nc] = target.view(M, nc // 96, 96).permute(1, 0, 2).reshape(-1)

    def lstm_epilogue_batch(self, problems):
        for prob in problems:
            self.lstm_epilogue(*prob)

    def heads_regressor(self, h_joint, h_grain, x_grain, w, b, y_joint, y_grain, grain_area):
        y_joint.copy_(torch.tanh(h_joint @ w[0].t() + b[0:2]))
        yg = h_grain @ w[1].t() + b[2:4]
        t0 = torch.tanh(yg[:, 0])
        grain_area.copy_(t0 / 20 + x_grain[:, 3])
        y_grain[:, 0] = t0
        y_grain[:, 1] = torch.relu(yg[:, 1])

    def heads_regressor_backward(self, w, y_joint, y_grain, g_y_joint, g_y_grain, g_grain_area):
        z = lambda g, like: torch.zeros_like(like) if g is None else g
        gyj, gyg = z(g_y_joint, y_joint), z(g_y_grain, y_grain)
        ga = torch.zeros(y_grain.size(0)) if g_grain_area is None else g_grain_area
        gpj, gpg = torch.zeros(y_joint.size(0), 4), torch.zeros(y_grain.size(0), 4)
        gpj[:, :2] = gyj * (1 - y_joint * y_joint)
        gpg[:, 0] = (gyg[:, 0] + ga / 20.0) * (1 - y_grain[:, 0] ** 2)
        gpg[:, 1] = gyg[:, 1] * (y_grain[:, 1] > 0)
        return gpj, gpg, gpj[:, :2] @ w[0], gpg[:, :2] @ w[1]

    def heads_regressor_update(self, h_joint, h_grain, x_joint, x_grain, w, b, y_joint, y_grain, grain_area, dz, zmax,
                               flags):
        self.heads_regressor(h_joint, h_grain, x_grain, w, b, y_joint, y_grain, grain_area)
        self.step_update(x_joint, x_grain, y_joint, y_grain, dz, zmax, flags)

    def step_refresh_prepare(self, x_joint, x_grain, zmax, flags, items, mirror=None):
        if int(flags[1]):
            x_joint[:, 2] = zmax
            x_grain[:, 2] = zmax
        if mirror is not None:
            mirror[0].copy_(x_joint)
            mirror[1].copy_(x_grain)
        for csr, ea, xs, xd, einfo in items:
            col, perm, row = csr.col.long(), csr.perm.long(), csr.row.long()
            E = ea.numel()
            rel = xs[col[:E], :2] - xd[row[:E], :2]
            rel = torch.where(rel > 0.5, -1.0, torch.where(rel < -0.5, 1.0, 0.0)) + rel
            ea[perm[:E]] = torch.sqrt(rel[:, 0] ** 2 + rel[:, 1] ** 2)
        self.edge_prepare(items)

    def heads_classifier(self, h_joint, edge_index_jj, edge_attr_jj, w_node, w_edge, node_tmp,
                         edge_event, edge, E_dev=None):
        assert E_dev is None or int(E_dev[0]) == edge_index_jj.size(1)   # (eager: the count in memory is the tensors' size)
        node_tmp[:, :6] = h_joint @ w_node.t()
        s, d, a = edge_index_jj[0], edge_index_jj[1], edge_attr_jj
        edge[:, 0] = torch.tanh(node_tmp[s, 0] + node_tmp[d, 3] + w_edge[0] * a + w_edge[3])
        edge[:, 1] = torch.tanh(node_tmp[s, 1] + node_tmp[d, 4] + w_edge[1] * a + w_edge[4])
        edge_event.copy_(node_tmp[s, 2] + node_tmp[d, 5] + w_edge[2] * a + w_edge[5])

    def step_update(self, x_joint, x_grain, y_joint, y_grain, dz, zmax, flags):
        dz = torch.tensor(dz, dtype=torch.float32)
        x_joint[:, :2] += y_joint / 5
        x_joint[:, 6:8] = y_joint
        x_joint[:, 2] += dz
        x_grain[:, 3] += y_grain[:, 0] / 20
        x_grain[:, 4] = y_grain[:, 1]
        x_grain[:, -1] = y_grain[:, 0]
        x_grain[:, 2] += dz
        flags[1] = int(x_grain[0, 2] > torch.tensor(zmax, dtype=torch.float32))

    def grain_centres(self, csr_jg, x_joint, x_grain, domain_factor=1.0, domain_offset=None, centres_before=None):
        if centres_before is not None:
            centres_before.copy_(x_grain[:, :2])
        rowptr, col = csr_jg.rowptr.tolist(), csr_jg.col.tolist()
        f = torch.tensor(domain_factor, dtype=torch.float32)
        xy = x_joint[:, :2].clone()
        if domain_factor > 1:
            xy = (xy + (domain_offset if domain_offset is not None else 0)) / f
        for g in range(x_grain.size(0)):
            js = col[rowptr[g]:rowptr[g + 1]]
            if len(js) <= 1:
                continue
            v = [xy[js[0]]]
            for j in js[1:]:
                rel = xy[j] - v[-1]
                v.append(xy[j] + torch.where(rel > 0.5, -1.0, torch.where(rel < -0.5, 1.0, 0.0)))
            v = torch.stack(v)
            m = v.sum(0) / len(js) + (~(v > -1e-12).all(0)).float()
            if domain_factor > 1:
                m = m * f
                m = m - torch.floor(m)
            x_grain[g, :2] = m

    def detect_events(self, grain_area, live_grain, area_threshold, edge_event, edge_index_jj,
                      logit_threshold, flags, range_word=None, E_dev=None):
        if range_word is not None:
            flags[2] = int(range_word[0])
            range_word.zero_()
        flags[0] = int(((live_grain > 0) & (grain_area < area_threshold)).sum())
        flags[1] = int(((edge_event > logit_threshold) & (edge_index_jj[0] < edge_index_jj[1])).sum())

    def step_refresh(self, x_joint, x_grain, zmax, flags, edges):
        if int(flags[1]):
            x_joint[:, 2] = zmax
            x_grain[:, 2] = zmax
        for ei, xs, xd, ea, *_ in edges:
            rel = xs[ei[0], :2] - xd[ei[1], :2]
            rel = torch.where(rel > 0.5, -1.0, torch.where(rel < -0.5, 1.0, 0.0)) + rel
            ea.copy_(torch.sqrt(rel[:, 0] ** 2 + rel[:, 1] ** 2))
