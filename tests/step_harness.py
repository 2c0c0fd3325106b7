"""One full training step of the PRODUCT (SemiSupervisedEpocher + INFONCEHook + fused RAdam on the
HIP kernels) next to the ORACLE's CPU step on identical weights, inputs, affine geometry and
labels.  Used by tests/test_gpu_step.py and by __graft_entry__.smoke()."""
from __future__ import annotations

import random
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
for _p in (REPO, REPO / "contrast-you_amd"):
    if str(_p) not in sys.path:
        sys.path.insert(0, str(_p))


class _Transforms:
    _total_freedom = False


class _Dataset:
    transforms = _Transforms()


class OneBatchLoader:
    dataset = _Dataset()

    def __init__(self, batch):
        self.batch = batch

    def __len__(self):
        return 1

    def __iter__(self):
        yield self.batch


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


class RawTaps:
    """`with RawTaps(model) as rec:` -- records, per BN layer (reference-named prefix) and evaluation, the raw
    conv output, the BN coefficients and the block output the device routes ReLU / max-pool by
    (cyhip.functions.RAW_TAP); `rec.force` is the `force=` argument of oracle.unet.unet_forward"""

    def __init__(self, model):
        self.names = {m: n for n, m in model.named_modules() if isinstance(m, torch.nn.BatchNorm2d)}
        self.force = {}

    def __enter__(self):
        from cyhip import functions as Fn

        def tap(bn, y, scale, shift, a):
            c = lambda t: None if t is None else t.detach().float().cpu()  # noqa: E731
            self.force.setdefault(self.names[bn], []).append({"y": c(y), "scale": c(scale), "shift": c(shift), "a": c(a)})

        self._fn, self._old = Fn, Fn.RAW_TAP
        Fn.RAW_TAP = tap
        return self

    def __exit__(self, *exc):
        self._fn.RAW_TAP = self._old
        return False


def compare_step_with_oracle(device="cuda:0", n_l=2, n_unl=3, hw=32, max_channel=128, dtype=torch.float32,
                             two_stage=True, lr=1e-3, py_seed=0, pin_routing=False, num_classes=4, return_grads=False):
    """pin_routing: the oracle differentiates the function the device evaluated (the device's ReLU / max-pool
    decisions and activation values, oracle.unet.unet_forward(force=)): deterministic gradient parity"""
    from contrastyou.amp import BF16Scaler
    from contrastyou.arch import UNet
    from contrastyou.hooks.base import TrainerHook
    from contrastyou.losses.kl import KL_div
    from contrastyou.optim import RAdam
    from oracle import losses as ol
    from oracle import step as ostep
    from oracle import unet as ou
    from semi_seg.augment import AffineAugment
    from semi_seg.epochers import SemiSupervisedEpocher
    from semi_seg.hooks import create_infonce_hooks

    dev = torch.device(device)
    sd0 = ou.init_state_dict(1, num_classes, max_channel, seed=7)
    psd0 = ol.init_projector_sd(max_channel, 256, 256, seed=8)
    b = ostep.synthetic_batch(n_l, n_unl, hw, num_classes, seed=99)

    # ---------------- product ----------------
    model = UNet(input_dim=1, num_classes=num_classes, max_channel=max_channel, momentum=0.01)
    model.load_state_dict(sd0, strict=True)
    model.to(dev)
    type(TrainerHook).names.clear()
    hook = create_infonce_hooks(model=model, feature_names="Conv5", weights=1.0, contrast_ons="partition",
                                spatial_size=1, data_name="acdc")
    proj = hook._hooks[0]._projector
    proj.load_state_dict(psd0, strict=True)
    hook.to(dev)
    opt = RAdam([{"params": list(model.parameters())}, {"params": list(hook.parameters())}], lr=lr,
                weight_decay=1e-5)
    if dtype == torch.float16:  # the reference's own mode: fp16 autocast + loss scaling
        scaler = torch.amp.GradScaler("cuda", enabled=True, init_scale=256.0)
    else:
        scaler = BF16Scaler() if dtype == torch.bfloat16 else torch.amp.GradScaler("cuda", enabled=False)
    lab = {"img": [b["labeled_image"], b["labeled_image"]], "gt": [b["labeled_target"], b["labeled_target"]],
           "filename": [[f"l{i}" for i in range(n_l)]] * 2, "partition": [["0"] * n_l] * 2,
           "scan_num": [b["labeled_scan"]] * 2}
    unl = {"img": [b["unlabeled_image"], b["unlabeled_image_cf"]],
           "gt": [torch.zeros(n_unl, 1, hw, hw, dtype=torch.long)] * 2,
           "filename": [[f"u{i}" for i in range(n_unl)]] * 2, "partition": [b["partition"]] * 2,
           "scan_num": [b["scan"]] * 2}
    ep = SemiSupervisedEpocher(model=model, optimizer=opt, labeled_loader=OneBatchLoader(lab),
                               unlabeled_loader=OneBatchLoader(unl), sup_criterion=KL_div(), num_batches=1,
                               cur_epoch=0, device=dev, two_stage=two_stage, disable_bn=False, scaler=scaler,
                               accumulate_iter=1)
    ep.init()
    captured = {}

    class Spy:  # records what the hooks are handed (kwargs contract of regularization())
        name = "spy"

    from contrastyou.hooks.base import EpocherHook

    class SpyHook(EpocherHook):
        def _call_implementation(self, **kw):
            captured.update({k: v for k, v in kw.items() if k in ("unlabeled_tf_logits", "unlabeled_logits_tf", "seed")})
            return torch.zeros((), device=dev)

    random.seed(py_seed)
    taps = RawTaps(model)
    with ep.register_hook(hook(), SpyHook(name="spy")), taps:
        ep.run()
    torch.cuda.synchronize()
    stats = ep.get_metric()

    # ---------------- oracle ----------------
    random.seed(py_seed)
    seed = random.randint(0, int(1e7))
    assert captured["seed"] == seed
    theta_np, gam_np = AffineAugment().sample(n_unl, seed)
    theta = torch.from_numpy(theta_np)
    gam = torch.from_numpy(gam_np)
    # (pinned routing: f64 oracle, so that what is left is the device's own f32 arithmetic error)
    od = torch.float64 if pin_routing else torch.float32
    sd = ou.clone_state_dict(sd0, requires_grad=True, dtype=od)
    psd = {k: v.clone().to(od).requires_grad_(True) for k, v in psd0.items()}
    labels = ol.get_label("partition", "acdc", b["partition"], b["scan"])
    theta = theta.to(od)
    out = ostep.semi_step(sd, psd, labeled_image=b["labeled_image"].to(od), labeled_target=b["labeled_target"],
                          unlabeled_image=b["unlabeled_image"].to(od),
                          unlabeled_image_tf=ol.affine_nearest(b["unlabeled_image_cf"], theta.float(), gam).to(od),
                          theta=theta,
                          labels=labels, momentum=0.01, two_stage=two_stage,
                          round_dtype=dtype if dtype in (torch.bfloat16, torch.float16) else None,
                          force=taps.force if pin_routing else None)
    out["total"].backward()
    names = [k for k, v in sd.items() if v.requires_grad]
    grads = {k: sd[k].grad.clone() for k in names}
    oparams = [sd[k] for k in names]
    oopt = torch.optim.RAdam([{"params": oparams}, {"params": list(psd.values())}], lr=lr, weight_decay=1e-5)
    oopt.step()

    semi = stats["semi"]
    res = {
        "sup": semi["sup_loss"], "reg": semi["reg_loss"],
        "rel_sup": abs(semi["sup_loss"] - out["sup"].item()) / abs(out["sup"].item()),
        "rel_reg": abs(semi["reg_loss"] - out["reg"].item()) / abs(out["reg"].item()),
        "rel_total": abs(semi["sup_loss"] + semi["reg_loss"] - out["total"].item()) / abs(out["total"].item()),
        "rel_logits": rel(captured["unlabeled_tf_logits"], out["unlabeled_tf_logits"]),
        "rel_logits_tf": rel(captured["unlabeled_logits_tf"], out["unlabeled_logits_tf"]),
    }
    pm = dict(model.named_parameters())
    res["rel_param_after_step"] = max(rel(pm[k], sd[k]) for k in names)
    res["rel_proj_after_step"] = max(rel(p, psd[k]) for k, p in proj.named_parameters())
    res["rel_running_mean"] = max(rel(bf, sd[k]) for k, bf in model.named_buffers() if "running_mean" in k)
    # gradients are consumed by the fused optimizer in place; compare through the update instead,
    # plus the raw flat gradient buffer that is still intact after step()
    flat = opt._flat[0]
    off, worst, num, den = 0, 0.0, 0.0, 0.0
    for k, p in model.named_parameters():
        n = p.numel()
        g = flat.grad[off:off + n].view(p.shape)
        worst = max(worst, rel(g, grads[k]))
        num += (g.detach().double().cpu() - grads[k].double()).pow(2).sum().item()
        den += grads[k].double().pow(2).sum().item()
        off += n
    res["rel_grad_worst"] = worst              # worst parameter, max-norm relative to that parameter's own gradient
    res["rel_grad_l2"] = (num / den) ** 0.5    # whole gradient vector (what the optimizer step sees)
    if return_grads:
        off, dev_g = 0, {}
        for k, p in model.named_parameters():
            dev_g[k] = flat.grad[off:off + p.numel()].view(p.shape).detach().double().cpu()
            off += p.numel()
        res["_grads"] = (dev_g, {k: v.detach().double() for k, v in grads.items()})
    res["dice"] = semi["sup_dice"]["DSC_mean"]
    res["scale_after"] = scaler.get_scale() if isinstance(scaler, torch.amp.GradScaler) and scaler.is_enabled() else None
    return res
