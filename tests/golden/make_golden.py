"""Generate the committed golden fixtures from the REAL reference (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Imports /root/reference through tests/golden/_ref_import.py (in-process stand-ins only for
absent third-party modules), runs its nn.Modules on seeded inputs and stores
inputs + outputs as small .npz files next to this script.  Big tensors (31 M parameters) are
not stored: they are regenerated from the seed (oracle.init_params mirrors the reference's
construction order, which this script asserts), and pinned by per-tensor statistics plus a
strided sample.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from _ref_import import import_reference  # noqa: E402

torch.set_num_threads(8)
torch.use_deterministic_algorithms(True)


def stat(t):
    t = t.detach().double().flatten()
    idx = torch.linspace(0, t.numel() - 1, steps=64).long()
    return np.concatenate([[t.sum().item(), t.abs().sum().item(), (t * t).sum().item()], t[idx].numpy()])


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print("wrote", name, sum(a.nbytes for a in out.values()) // 1024, "KiB")


def g1_blocks(ns):
    L = ns.layers2d
    torch.manual_seed(11)
    dc = L.DoubleConvolution(3, 8)
    x = torch.randn(2, 3, 12, 20, requires_grad=True)
    y = dc(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    d = {"dc_x": x, "dc_y": y, "dc_gy": gy, "dc_gx": x.grad}
    for k, v in dc.state_dict().items():
        d["dc_p_" + k] = v
    for k, v in dc.named_parameters():
        d["dc_g_" + k] = v.grad
    up = L.UpSample(8, 4)
    xu = torch.randn(2, 8, 6, 10, requires_grad=True)
    yu = up(xu)
    gyu = torch.randn_like(yu)
    yu.backward(gyu)
    d.update({"up_x": xu, "up_y": yu, "up_gy": gyu, "up_gx": xu.grad,
              "up_w": up.up.weight, "up_b": up.up.bias, "up_gw": up.up.weight.grad, "up_gb": up.up.bias.grad})
    ds = L.DownSample()
    xd = torch.randn(2, 4, 10, 14, requires_grad=True)
    # plant ties so the "first max wins" rule is pinned
    with torch.no_grad():
        xd[0, 0, 0:2, 0:2] = 1.5
        xd[1, 2, 4:6, 6:8] = -0.25
    yd = ds(xd)
    gyd = torch.randn_like(yd)
    yd.backward(gyd)
    d.update({"ds_x": xd, "ds_y": yd, "ds_gy": gyd, "ds_gx": xd.grad})
    cc = L.CropAndConcat()
    a = torch.randn(1, 2, 8, 8)
    b = torch.randn(1, 3, 12, 12)
    d.update({"cc_a": a, "cc_b": b, "cc_y": cc(a, b)})
    save("g1_blocks2d.npz", **d)


def g2_unet(ns, cin, cout, B, H, W, tag, oracle):
    torch.manual_seed(0)
    net = ns.unet2d.UNet(cin, cout)
    # the oracle's init must reproduce the reference's construction order bit for bit
    po = oracle.init_params(cin, cout, seed=0)
    sd = net.state_dict()
    assert list(sd.keys()) == list(po.keys()), "param order differs"
    for k in sd:
        assert torch.equal(sd[k], po[k]), k
    g = torch.Generator().manual_seed(1234)
    images = torch.randn(B, cin, H, W, generator=g)
    if cout > 1:
        labels = torch.randint(0, cout, (B, H, W), generator=g)
    else:
        labels = (torch.rand(B, 1, H, W, generator=g) > 0.5).float()
    cfg = ns.unet2d.UNetConfig(in_channels=cin, out_channels=cout, unet_type="UNet")
    torch.manual_seed(0)
    model = ns.unet2d.UNetModel(cfg)
    model.unet.load_state_dict(sd)
    out = model(images=images, labels=labels)
    loss, logits = out["loss"], out["logits"]
    loss.backward()
    d = {"images": images, "labels": labels, "logits": logits, "loss": loss,
         "argmax": logits.argmax(1) if cout > 1 else (logits > 0).long()}
    names = [k for k, _ in model.unet.named_parameters()]
    d["names"] = np.array(names)
    d["param_stats"] = np.stack([stat(p) for _, p in model.unet.named_parameters()])
    d["grad_stats"] = np.stack([stat(p.grad) for _, p in model.unet.named_parameters()])
    # small tensors in full
    d["g_final_w"] = model.unet.final_conv.weight.grad
    d["g_final_b"] = model.unet.final_conv.bias.grad
    d["g_down0_first_w"] = model.unet.down_conv[0].first.weight.grad
    d["g_down0_first_b"] = model.unet.down_conv[0].first.bias.grad
    d["g_mid_second_b"] = model.unet.middle_conv.second.bias.grad
    d["g_up0_b"] = model.unet.up_sample[0].up.bias.grad
    # HF Trainer step: clip 1.0 + AdamW with the decay / no-decay split, 3 steps at constant lr
    decay = [p for n, p in model.unet.named_parameters() if not n.endswith("bias")]
    nodecay = [p for n, p in model.unet.named_parameters() if n.endswith("bias")]
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": 1e-3}, {"params": nodecay, "weight_decay": 0.0}],
                            lr=5e-3, betas=(0.9, 0.999), eps=1e-8)
    losses, norms = [loss.item()], []
    for step in range(3):
        if step > 0:
            opt.zero_grad()
            l = model(images=images, labels=labels)["loss"]
            l.backward()
            losses.append(l.item())
        norms.append(torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0).item())
        opt.step()
        d[f"param_stats_step{step + 1}"] = np.stack([stat(p) for _, p in model.unet.named_parameters()])
    d["step_losses"] = np.array(losses)
    d["step_gradnorms"] = np.array(norms)
    save(f"g2_unet_{tag}.npz", **d)


def g3_unet3d(ns, oracle3):
    B3 = ns.bb3d
    # blocks with small channels, stored in full
    torch.manual_seed(5)
    sc = B3.SingleConv(16, 24, order="gcr", num_groups=8)
    x = torch.randn(2, 16, 6, 8, 10, requires_grad=True)
    y = sc(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    d = {"sc_x": x, "sc_y": y, "sc_gy": gy, "sc_gx": x.grad}
    for k, v in sc.named_parameters():
        d["sc_p_" + k] = v
        d["sc_g_" + k] = v.grad
    # decoder with nearest upsampling to a non-2x size
    torch.manual_seed(6)
    dec = B3.Decoder(24 + 16, 16, basic_module=B3.DoubleConv, num_groups=8)
    enc_f = torch.randn(1, 16, 6, 10, 10, requires_grad=True)
    low = torch.randn(1, 24, 3, 5, 5, requires_grad=True)
    yd = dec(enc_f, low)
    gyd = torch.randn_like(yd)
    yd.backward(gyd)
    d.update({"dec_enc": enc_f, "dec_low": low, "dec_y": yd, "dec_gy": gyd, "dec_genc": enc_f.grad,
              "dec_glow": low.grad})
    for k, v in dec.named_parameters():
        d["dec_p_" + k] = v
        d["dec_g_" + k] = v.grad
    save("g3_blocks3d.npz", **d)

    # small full net, everything stored: f_maps [8, 16, 32], 3 levels
    torch.manual_seed(0)
    net = ns.model3d.UNet3D(1, 3, f_maps=[8, 16, 32], num_groups=4)
    po = oracle3.init_params(1, 3, f_maps=[8, 16, 32], seed=0)
    sd = net.state_dict()
    assert list(sd.keys()) == list(po.keys()), (list(sd.keys()), list(po.keys()))
    for k in sd:
        assert torch.equal(sd[k], po[k]), k
    g = torch.Generator().manual_seed(77)
    x = torch.randn(2, 1, 8, 12, 16, generator=g)
    t = (torch.rand(2, 3, 8, 12, 16, generator=g) > 0.5).float()
    net.train()
    logits = net(x)
    crit = ns.losses3d.BCEDiceLoss(1.0, 1.0)
    loss = crit(logits, t)
    loss.backward()
    d = {"x": x, "t": t, "logits": logits, "loss": loss,
         "dice_loss": ns.losses3d.DiceLoss()(logits.detach(), t),
         "quirk_loss": crit(torch.sigmoid(logits.detach()), t),
         "argmax": logits.argmax(1)}
    for k, v in net.named_parameters():
        d["p_" + k] = v
        d["g_" + k] = v.grad
    save("g3_unet3d_small.npz", **d)

    g3_unet3d_default(ns, oracle3)

    # loss-only vectors (N(0,1) logits / Bernoulli targets)
    g = torch.Generator().manual_seed(3)
    lg = torch.randn(2, 3, 8, 8, 8, generator=g, requires_grad=True)
    tg = (torch.rand(2, 3, 8, 8, 8, generator=g) > 0.5).float()
    l = crit(lg, tg)
    l.backward()
    save("g3_loss.npz", logits=lg, target=tg, loss=l, grad=lg.grad,
         dice=ns.losses3d.compute_per_channel_dice(torch.sigmoid(lg.detach()), tg))


MIN_ARGMAX_GAP = 1e-4


def g3_unet3d_default(ns, oracle3):
    """default-width net (64..512, 16.3 M params) on 1x1x16^3: stats only.

    The arg-max of this golden is a BIT-EXACT bar (north star: "seg masks bit-exact at argmax"), so the input must not hold a voxel whose top-2 logit gap is within
    reach of fp32 summation-order noise (two correct fp32 evaluations of this net differ by ~2e-5): input seeds are tried from 78 upwards until the smallest gap
    is >= MIN_ARGMAX_GAP (seed 78, used until round 3, has a 2.4e-5 voxel; seed 79 has 2.3e-4), and the generator asserts it."""
    crit = ns.losses3d.BCEDiceLoss(1.0, 1.0)
    torch.manual_seed(0)
    net = ns.model3d.UNet3D(1, 3)
    po = oracle3.init_params(1, 3, seed=0)
    sd = net.state_dict()
    assert list(sd.keys()) == list(po.keys())
    for k in sd:
        assert torch.equal(sd[k], po[k]), k
    for seed in range(78, 178):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(1, 1, 16, 16, 16, generator=g)
        t = (torch.rand(1, 3, 16, 16, 16, generator=g) > 0.5).float()
        logits = net(x)
        top2 = logits.detach().topk(2, dim=1).values
        gap = float((top2[:, 0] - top2[:, 1]).min())
        print(f"g3_unet3d_default: input seed {seed}: smallest top-2 logit gap {gap:.3g}")
        if gap >= MIN_ARGMAX_GAP:
            break
    assert gap >= MIN_ARGMAX_GAP, gap
    loss = crit(logits, t)
    loss.backward()
    d = {"x": x, "t": t, "logits": logits, "loss": loss, "argmax": logits.argmax(1),
         "input_seed": np.int64(seed), "min_top2_gap": np.float64(gap),
         "names": np.array([k for k, _ in net.named_parameters()]),
         "param_stats": np.stack([stat(p) for _, p in net.named_parameters()]),
         "grad_stats": np.stack([stat(p.grad) for _, p in net.named_parameters()]),
         "g_final_w": net.final_conv.weight.grad, "g_final_b": net.final_conv.bias.grad}
    save("g3_unet3d_default.npz", **d)


def main():
    ns = import_reference()
    from oracle import unet2d_oracle, unet3d_oracle
    if sys.argv[1:] == ["--only", "unet3d_default"]:        # regenerate that one fixture (round 4: near-tie-free input) without touching the others
        g3_unet3d_default(ns, unet3d_oracle)
        return
    g1_blocks(ns)
    g2_unet(ns, 1, 2, 2, 32, 32, "1_2", unet2d_oracle)
    g2_unet(ns, 3, 4, 1, 32, 48, "3_4", unet2d_oracle)
    g2_unet(ns, 1, 1, 2, 16, 16, "1_1", unet2d_oracle)
    g3_unet3d(ns, unet3d_oracle)


if __name__ == "__main__":
    main()
