"""CPU tests of the host-side mirror of the reference's plugin API (no kernels run here)."""
import random

import numpy as np
import pytest
import torch
from torch import nn


def test_module_base_buffers_and_schema():
    from contrastyou.nn import Buffer, ModuleBase, NoTrackable

    class M(ModuleBase):
        def __init__(self):
            super().__init__()
            self.lin = nn.Linear(2, 2)
            self.epoch = Buffer(3)
            self.ref = NoTrackable(nn.Linear(4, 4))
            self.opt = torch.optim.SGD(self.lin.parameters(), lr=0.1)

    m = M()
    assert m.epoch == 3
    m.epoch = 4
    sd = m.state_dict()
    assert set(sd) == {"module_state", "buffer_state", "other_state"}
    assert sd["buffer_state"] == {"epoch": 4}
    assert "opt" in sd["other_state"] and "ref" not in sd["other_state"]
    assert all(not k.startswith("ref") for k in sd["module_state"])
    m2 = M()
    m2.load_state_dict(sd)
    assert m2.epoch == 4
    with pytest.raises(RuntimeError):
        m2.load_state_dict({"module_state": sd["module_state"], "buffer_state": {"bogus": 1}, "other_state": {}})


def test_hook_names_are_unique_and_combine_sums():
    from contrastyou.hooks.base import (CombineEpochHook, CombineTrainerHook, EpocherHook, HookNameExistError,
                                        TrainerHook)

    class T(TrainerHook):
        def __init__(self, *, hook_name, v):
            super().__init__(hook_name=hook_name)
            self.v = v
            self.lin = nn.Linear(1, 1)

        @property
        def learnable_modules(self):
            return [self.lin]

        def __call__(self):
            return E(name=self._hook_name, v=self.v)

    class E(EpocherHook):
        def __init__(self, *, name, v):
            super().__init__(name=name)
            self.v, self.calls = v, []

        def before_forward_pass(self, **kw):
            self.calls.append("bf")

        def _call_implementation(self, **kw):
            return torch.tensor(float(self.v))

    type(TrainerHook).names.clear()
    a, b = T(hook_name="a", v=1), T(hook_name="b", v=2)
    with pytest.raises(HookNameExistError):
        T(hook_name="a", v=3)
    comb = CombineTrainerHook(a, b)
    assert len(list(comb.parameters())) == 4
    eh = comb()
    assert isinstance(eh, CombineEpochHook)

    from contrastyou.meters import MeterInterface

    class FakeEpocher:
        meters = MeterInterface()

    fe = FakeEpocher()
    eh.epocher = fe
    eh.call_before_forward_pass()
    assert float(eh()) == 3.0
    assert all(h.calls == ["bf"] for h in eh._epocher_hook)


def test_meters_and_dice_match_oracle(golden_dir):
    from contrastyou.meters import AverageValueMeter, MeterInterface, UniversalDice
    g = np.load(golden_dir / "heads_losses.npz")
    m = UniversalDice(4, report_axis=[1, 2, 3])
    for pr, tt, gp in zip(g["dice_preds"], g["dice_targets"], g["dice_groups"]):
        m.add(torch.from_numpy(pr), torch.from_numpy(tt), group_name=list(gp))
    summ = m.summary()
    for k, v in zip(g["dice_keys"], g["dice_vals"]):
        assert abs(summ[str(k)] - float(v)) < 1e-6
    mi = MeterInterface(default_focus="semi")
    mi.register_meter("loss", AverageValueMeter())
    with mi.focus_on("hook"):
        mi.register_meter("loss", AverageValueMeter())
        mi["loss"].add(torch.tensor(2.0))
    mi["loss"].add(1.0)
    mi["loss"].add(torch.tensor(3.0))
    st = dict(mi.statistics())
    assert st["semi"]["loss"] == 2.0 and st["hook"]["loss"] == 2.0


def test_label_generators_match_oracle():
    from oracle import losses as ol
    from semi_seg.hooks.utils import get_label
    part = ["1", "0", "2", "1", "0"]
    scan = ["patient003_01", "patient001_00", "patient003_00", "patient010_01", "patient001_01"]
    for on in ("partition", "patient", "cycle", "self"):
        assert get_label(on, "acdc", part, scan) == ol.get_label(on, "acdc", part, scan)
    assert get_label("patient", "prostate", part, scan) == ol.get_label("patient", "prostate", part, scan)


def test_affine_parameters_are_a_function_of_the_seed_only():
    from semi_seg.augment import AffineAugment
    a = AffineAugment()
    t1, g1 = a.sample(5, 1234)
    t2, g2 = a.sample(5, 1234)
    t3, _ = a.sample(5, 1235)
    assert np.array_equal(t1, t2) and np.array_equal(g1, g2) and not np.array_equal(t1, t3)
    sc = np.sqrt(np.abs(np.linalg.det(t1[:, :, :2])))  # |det| = 1/scale^2
    assert np.all(1 / sc >= 0.8 - 1e-6) and np.all(1 / sc <= 1.3 + 1e-6)
    assert np.all(np.abs(t1[:, :, 2]) <= 0.1 + 1e-6) and np.all((g1 >= 0.5) & (g1 <= 2.0))


def test_epocher_protocol_order_and_guards(monkeypatch):
    """hook call order of one batch (epocher.py:86-116) with the compute pieces stubbed out"""
    from contrastyou.hooks.base import EpocherHook
    from contrastyou.meters import UniversalDice
    from semi_seg.epochers import SemiSupervisedEpocher

    class Net(nn.Module):
        num_classes = 3

        def __init__(self):
            super().__init__()
            self.c = nn.Conv2d(1, 3, 1)

        def forward(self, x):
            return self.c(x)

    class DS:
        class transforms:
            _total_freedom = False

    class Loader:
        dataset = DS()

        def __init__(self, n):
            self.b = {"img": [torch.rand(n, 1, 8, 8)] * 2, "gt": [torch.randint(0, 3, (n, 1, 8, 8))] * 2,
                      "filename": [[str(i) for i in range(n)]] * 2, "partition": [["0"] * n] * 2,
                      "scan_num": [[f"p{i}_00" for i in range(n)]] * 2}

        def __len__(self):
            return 2

        def __iter__(self):
            yield self.b
            yield self.b

    order = []

    class H(EpocherHook):
        def before_batch_update(self, **kw):
            order.append("before_batch")

        def before_forward_pass(self, **kw):
            order.append("before_fwd")

        def after_forward_pass(self, **kw):
            order.append("after_fwd")

        def before_regularization(self, **kw):
            order.append("before_reg")

        def _call_implementation(self, *, seed, unlabeled_tf_logits, unlabeled_logits_tf, label_group,
                                 partition_group, affine_transformer, **kw):
            order.append("call")
            assert unlabeled_tf_logits.shape == unlabeled_logits_tf.shape
            assert len(label_group) == len(partition_group) == unlabeled_tf_logits.shape[0]
            return unlabeled_tf_logits.mean() * 0

        def after_regularization(self, **kw):
            order.append("after_reg")

        def after_batch_update(self, **kw):
            order.append("after_batch")

    net = Net()
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    crit = lambda p, t: -(t * torch.log(p + 1e-8)).sum(1).mean()  # noqa: E731  (no from_logits -> generic path)
    ep = SemiSupervisedEpocher(model=net, optimizer=opt, labeled_loader=Loader(2), unlabeled_loader=Loader(3),
                               sup_criterion=crit, num_batches=2, device="cpu", two_stage=True,
                               scaler=torch.amp.GradScaler("cpu", enabled=False), accumulate_iter=1)
    with pytest.raises(RuntimeError):
        ep.run()
    ep.init()
    ep._affine_transformer = lambda x, *, mode, seed: x  # geometry kernels are GPU-only
    monkeypatch.setattr(UniversalDice, "add_logits",
                        lambda self, lg, tg, group_name=None: self.add(lg.argmax(1), tg.squeeze(1),
                                                                       group_name=group_name))
    w0 = net.c.weight.detach().clone()
    with ep.register_hook(H(name="h")):
        ep.run()
    one = ["before_batch", "before_fwd", "after_fwd", "before_reg", "call", "after_reg", "after_batch"]
    assert order == one * 2
    assert not torch.equal(w0, net.c.weight)
    m = ep.get_metric()
    assert set(m["semi"]) == {"lr", "sup_loss", "sup_dice", "reg_loss"} and "h" not in m or m.get("h") == {}


def test_warmup_scheduler_matches_reference_sequence(golden_dir):
    """GradualWarmupScheduler (product) against the lr sequence the reference class produced"""
    import numpy as np
    from contrastyou.optim import GradualWarmupScheduler
    g = np.load(golden_dir / "next_rows.npz")
    par = [torch.nn.Parameter(torch.zeros(1))]
    opt = torch.optim.SGD(par, lr=1e-6)
    cos = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=30, eta_min=1e-7)
    sch = GradualWarmupScheduler(opt, 300, total_epoch=10, after_scheduler=cos)
    lrs = []
    for e in range(40):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
        if e == 20:  # checkpoint round trip in the middle of the cosine phase
            state, opt_state = sch.state_dict(), opt.state_dict()
            opt = torch.optim.SGD(par, lr=1e-6)
            cos = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=30, eta_min=1e-7)
            sch = GradualWarmupScheduler(opt, 300, total_epoch=10, after_scheduler=cos)
            opt.load_state_dict(opt_state)  # the order Trainer.load_state_dict uses
            sch.load_state_dict(state)
    assert np.allclose(lrs, g["lr_seq"], rtol=1e-12, atol=0)
    with pytest.raises(ValueError):
        GradualWarmupScheduler(opt, 1.0, total_epoch=10)


def test_trainer_checkpoint_schema_and_resume(tmp_path):
    """Trainer: optimizer from config (+ hook param group), reference checkpoint schema, safe
    (weights_only) resume of epoch counters, scheduler, optimizer lr and storage"""
    from contrastyou.arch import UNet
    from contrastyou.losses.kl import KL_div
    from semi_seg.trainers import trainer_zoo
    cfg = {"Optim": {"name": "RAdam", "lr": 1e-6, "weight_decay": 1e-5},
           "Scheduler": {"multiplier": 300, "warmup_max": 10}, "Trainer": {"name": "semi"}}

    def make():
        tr = trainer_zoo["semi"](model=UNet(input_dim=1, num_classes=4, max_channel=128), labeled_loader=[],
                                 unlabeled_loader=[], val_loader=[], test_loader=[], criterion=KL_div(),
                                 save_dir=str(tmp_path), max_epoch=30, num_batches=2, device="cpu", disable_bn=False,
                                 two_stage=True, config=cfg, enable_scale=True)
        tr.init()
        return tr

    tr = make()
    with pytest.raises(RuntimeError):
        tr.init()
    with pytest.raises(RuntimeError):
        with tr.register_hook():
            pass
    sd = tr.state_dict()
    assert list(sd) == ["module_state", "buffer_state", "other_state"]
    assert set(sd["buffer_state"]) == {"_save_dir", "_max_epoch", "_num_batches", "config", "_cur_epoch",
                                       "_start_epoch", "_best_score"}
    assert {"_optimizer", "_scheduler", "scaler", "_storage"} <= set(sd["other_state"])
    assert "_model._Conv1.conv.0.weight" in sd["module_state"]
    tr._cur_epoch, tr._best_score = 7, 0.5
    for _ in range(7):
        tr._optimizer.step() if False else None
        tr._scheduler.step()
    tr._storage.put("tra/semi/sup_loss", {"mean": 1.0}, epoch=7)
    with torch.no_grad():
        tr._model._Deconv_1x1.bias.fill_(0.25)
    tr.save_to(save_name="last.pth")
    tr2 = make()
    tr2.resume_from_path(str(tmp_path))
    assert tr2._cur_epoch == 7 and tr2._best_score == 0.5 and tr2._scheduler.last_epoch == 7
    assert abs(tr2._optimizer.param_groups[0]["lr"] - tr._optimizer.param_groups[0]["lr"]) < 1e-15
    assert tr2._storage.get("tra/semi/sup_loss", 7) == {"mean": 1.0}
    assert float(tr2._model._Deconv_1x1.bias[0]) == 0.25
    assert (tmp_path / "config.yaml").exists() and (tmp_path / "config_1.yaml").exists()
    with pytest.raises(NotImplementedError):
        trainer_zoo["mixup"]()


def test_feature_tap_tail_equals_the_slice_of_the_concatenation():
    """SingleFeatureExtractor.tail(rows) == feature()[-rows:], without touching earlier recordings
    when the boundary falls between two recordings"""
    import torch
    from torch import nn
    from contrastyou.arch.utils import SingleFeatureExtractor

    class Net(nn.Module):
        arch_elements = ("blk",)

        def __init__(self):
            super().__init__()
            self.blk = nn.Linear(3, 2)

        def get_module(self, name):
            return getattr(self, name)

        def forward(self, x):
            return self.blk(x)

    net = Net()
    with SingleFeatureExtractor(net, "blk") as tap:
        tap.set_enable(True)
        a, b = net(torch.randn(2, 3)), net(torch.randn(4, 3))
        full = tap.feature()
        assert torch.equal(tap.tail(4), full[-4:]) and tap.tail(4) is b
        assert torch.equal(tap.tail(3), full[-3:]) and torch.equal(tap.tail(6), full)
        tap.tail(4).sum().backward()
        assert a.grad_fn is not None and net.blk.weight.grad is not None


def test_dp_early_rule_and_bucket_boundaries():
    """data parallel: the early start of gradient buckets is decided by a rule (gradient volume, world size), and when
    it is on no bucket mixes parameters of different "gradients final" marks (one bucket over the whole
    U-Net could never start early)"""
    from contrastyou.arch.unet import UNet
    from contrastyou.optim.fused_radam import FlatParams, FusedRAdam
    from cyhip import ops
    assert not ops.dp_early_rule(int(34.5e6), 8)      # the U-Net's gradients at 8 ranks: 0.21 ms exposed -> off
    assert ops.dp_early_rule(int(80e6), 8)
    assert not ops.dp_early_rule(int(1e9), 1)
    net = UNet(input_dim=1, num_classes=4, max_channel=512)
    opt = FusedRAdam(net.parameters(), lr=1e-3, data_parallel=False)
    f = FlatParams(list(net.parameters()))
    was = ops.DP_EARLY
    try:
        ops.DP_EARLY = False
        assert len(opt._buckets(f)) == 1
        ops.DP_EARLY = True
        cuts = opt._buckets(f)
        tags = [{p.__dict__.get("_cy_ready_tag") for p in f.params[i:j]} for _, _, i, j in cuts]
        assert all(len(t) == 1 for t in tags), tags
        assert [next(iter(t)) for t in tags] == ["decoder", "conv5", "conv4", None]
        assert sum(b - a for a, b, _, _ in cuts) == f.numel
    finally:
        ops.DP_EARLY = was


def test_nearest_source_index_is_atens():
    """the superpixel hook's label lookup = F.interpolate(mode="nearest") at the sampled positions, for every size
    (integer (dst * in) // out is a different index at e.g. 224 -> 48)"""
    import torch.nn.functional as F
    from semi_seg.hooks.infonce import nearest_source_index
    diff = 0
    for n_in, n_out in ((224, 48), (224, 46), (224, 92), (224, 184), (256, 82), (72, 6), (224, 20), (48, 6), (224, 224)):
        src = torch.arange(n_in, dtype=torch.float32).view(1, 1, n_in, 1).expand(1, 1, n_in, 4).contiguous()
        want = F.interpolate(src, size=(n_out, 4), mode="nearest")[0, 0, :, 0].long().tolist()
        got = [nearest_source_index(d, n_in, n_out) for d in range(n_out)]
        assert got == want, (n_in, n_out)
        diff += sum(g != (d * n_in) // n_out for d, g in enumerate(got))
    assert diff > 0  # (the sizes above include ones where the integer formula is wrong)
