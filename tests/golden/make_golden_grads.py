"""Golden vectors for SURVEY 8f-3 (training path): loss values and parameter gradients of the
UNMODIFIED reference models under the reference's loss definitions (train.py:31-37 regressor,
train.py:40-70 classifier, edge_len off) on the 40 um graph.

Runs only in the build container (needs /root/reference); writes `golden_cfg1_grads.npz`.
    python tests/golden/make_golden_grads.py

Set-up: weights RandomState(10020) x1.0 (regressor) / RandomState(10021) (classifier), targets
drawn from RandomState(77): y_joint U(-1,1) [236,2], y_grain U(-1,1) [118,2], edge labels in
{-1, 0, 1} [708] (-1 = unlabelled, skipped by the loss), masks from the fixture (all ones here)
with every 7th joint and every 5th grain masked out.

A full gradient is 1.2 M floats per model, too big for a fixture; per parameter tensor the file
keeps a digest -- sum, L2 norm, max |g| and the 6 entries at the flat positions
RandomState(5).randint(numel, size=6) -- plus the loss.  The oracle is pinned against these
digests (asserted below and in tests/test_oracle_golden.py), the HIP training path against the
oracle's full gradients and against the digests.
"""
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402  (sets up sys.path for the reference + stubs)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from graingraphnn_amd.training import classifier_loss, regressor_loss  # noqa: E402


def targets(n_j, n_g, E):
    rs = np.random.RandomState(77)
    y = {"joint": rs.uniform(-1, 1, (n_j, 2)).astype(np.float32),
         "grain": rs.uniform(-1, 1, (n_g, 2)).astype(np.float32),
         "edge_event": rs.randint(-1, 2, size=E).astype(np.int64)}
    mask = {"joint": np.ones((n_j, 1), np.float32), "grain": np.ones((n_g, 1), np.float32)}
    mask["joint"][::7] = 0
    mask["grain"][::5] = 0
    return y, mask


def digest(name, g):
    g = g.detach().cpu().numpy().astype(np.float64).ravel()
    idx = np.random.RandomState(5).randint(g.size, size=6)
    return np.concatenate([[g.sum(), np.sqrt((g * g).sum()), np.abs(g).max()], g[idx]])


def reference_losses(R, Cm, X, EI, EA, y, mask):
    """The reference's criterion (train.py:25-70) written out for args.edge_len == False."""
    pr = R(X, EI, EA)
    loss_r = 100 * (torch.mean(mask["joint"] * (y["joint"] - pr["joint"]) ** 2)
                    + torch.mean(mask["grain"] * (y["grain"] - pr["grain"]) ** 2))
    pc = Cm(X, EI, EA)
    z, lab = pc["edge_event"], y["edge_event"]
    keep = torch.where(lab > -1)
    loss_c = torch.nn.BCEWithLogitsLoss(pos_weight=torch.tensor(1.0))(z[keep], lab[keep].float())
    return loss_r, loss_c


def main():
    g40, x, ei, ea = mg.load_graph(os.path.join(mg.REF, "graphs/40_40/seed10020_G1.904_R0.558_span6.pkl"))
    hp = mg.make_hyper(g40)
    R, Cm = mg.build_reference(hp, x, ei, ea, 10020, 1.0)
    R.train(), Cm.train()
    X, EI, EA = mg.tt(x), mg.tt(ei), mg.tt(ea)
    y_np, mask_np = targets(x["joint"].shape[0], x["grain"].shape[0], ei[mg.JJ].shape[1])
    y, mask = mg.tt(y_np), mg.tt(mask_np)
    loss_r, loss_c = reference_losses(R, Cm, X, EI, EA, y, mask)
    R.zero_grad(), Cm.zero_grad()
    loss_r.backward()
    loss_c.backward()
    out = {"loss_regressor": np.float64(loss_r.item()), "loss_classifier": np.float64(loss_c.item())}
    for tag, m in (("R", R), ("C", Cm)):
        for name, p in m.named_parameters():
            gr = p.grad if p.grad is not None else torch.zeros_like(p)
            out[f"{tag}/{name}"] = digest(name, gr)
    # the oracle, same weights, same losses through the product's loss functions
    oR, oC = mg.build_oracle(hp, 10020, 1.0)
    oR.train(), oC.train()
    lr_o = regressor_loss(y, oR(mg.tt(x), mg.tt(ei), mg.tt(ea)), mask)
    lc_o = classifier_loss(y, oC(mg.tt(x), mg.tt(ei), mg.tt(ea)), 1.0)
    lr_o.backward()
    lc_o.backward()
    assert abs(lr_o.item() - loss_r.item()) <= 1e-5 * abs(loss_r.item()), (lr_o.item(), loss_r.item())
    assert abs(lc_o.item() - loss_c.item()) <= 1e-5 * abs(loss_c.item()), (lc_o.item(), loss_c.item())
    worst = 0.0
    ref_params = {"R": dict(R.named_parameters()), "C": dict(Cm.named_parameters())}
    for tag, m in (("R", oR), ("C", oC)):
        for name, p in m.named_parameters():
            gr = p.grad if p.grad is not None else torch.zeros_like(p)
            ref = ref_params[tag][name].grad
            ref = ref if ref is not None else torch.zeros_like(gr)
            scale = max(float(ref.abs().max()), 1e-12)
            worst = max(worst, float((gr - ref).abs().max()) / scale if float(ref.abs().max()) > 1e-9
                        else float((gr - ref).abs().max()))
    print(f"oracle vs reference gradients: worst per-tensor relative error {worst:.2e}")
    assert worst <= 2e-4, worst
    np.savez_compressed(os.path.join(HERE, "golden_cfg1_grads.npz"), **out)
    print("wrote golden_cfg1_grads.npz:", len(out), "arrays; losses", loss_r.item(), loss_c.item())


if __name__ == "__main__":
    main()
