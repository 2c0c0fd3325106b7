"""Integration of the section-8 "next" rows with the epocher/trainer layer on the GPU:
dense InfoNCE + discrete-MI hooks inside a SemiSupervisedEpocher step (losses re-derived by the
oracle from the tapped feature maps), the pre-training epocher (config C5 in small) against the
oracle's encoder-only step, optimizer semantics for parameters without gradients, and two epochs of
SemiTrainer with the warm-up schedule, checkpoint and resume."""
import random

import pytest
import torch

from test_gpu_hooks_dice import Loader, blob_batch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def test_dense_infonce_and_mi_hooks_in_a_semi_step():
    from contrastyou.arch import UNet
    from contrastyou.hooks.base import CombineTrainerHook, EpocherHook, TrainerHook
    from contrastyou.losses.kl import KL_div
    from contrastyou.optim import RAdam
    from oracle import losses as ol
    from oracle import next_rows as onr
    from oracle import unet as ou
    from semi_seg.augment import AffineAugment
    from semi_seg.epochers import SemiSupervisedEpocher
    from semi_seg.hooks import (create_discrete_mi_consistency_hook, create_iid_segmentation_hook,
                                create_infonce_hooks)

    g = torch.Generator().manual_seed(2)
    n, hw, K = 3, 48, 4
    sd0 = ou.init_state_dict(1, K, 256, seed=3)  # Up_conv2 has 16 channels at max_channel=256
    lab, unl = blob_batch(n, hw, K, g), blob_batch(n, hw, K, g)
    unl["img"] = [unl["img"][0], torch.rand(n, 1, hw, hw, generator=g)]
    model = UNet(input_dim=1, num_classes=K, max_channel=256, momentum=0.01)
    model.load_state_dict(sd0)
    model.to(DEV)
    type(TrainerHook).names.clear()
    dense = create_infonce_hooks(model=model, feature_names=["Conv5", "Up_conv2"], weights=[1.0, 0.5],
                                 contrast_ons=["partition", "self"], spatial_size=[1, 7], data_name="acdc")
    mi = create_discrete_mi_consistency_hook(model=model, feature_names=["Conv5", "Up_conv2"], mi_weights=[0.1, 0.05],
                                             dense_paddings=None, consistency_weight=1.0)
    iid = create_iid_segmentation_hook(weight=0.2, mi_lambda=1.5)
    hook = CombineTrainerHook(dense, mi, iid).to(DEV)
    before = {k: v.detach().clone() for k, v in hook.state_dict()["module_state"].items()}
    opt = RAdam([{"params": list(model.parameters())}, {"params": list(hook.parameters())}], lr=1e-3)
    ep = SemiSupervisedEpocher(model=model, optimizer=opt, labeled_loader=Loader([lab]), unlabeled_loader=Loader([unl]),
                               sup_criterion=KL_div(), num_batches=1, cur_epoch=0, device=DEV, two_stage=True,
                               disable_bn=False, scaler=torch.amp.GradScaler("cuda", enabled=False), accumulate_iter=1)
    ep.init()
    tap = {}
    dense_hook, mi_dec_hook = dense._hooks[1], mi._hooks[0]._hooks[1]
    mi_enc_hook = mi._hooks[0]._hooks[0]

    class Spy(EpocherHook):
        def _call_implementation(self, *, seed, unlabeled_tf_logits, unlabeled_logits_tf, **kw):
            tap["seed"] = seed
            tap["tf_logits"], tap["logits_tf"] = unlabeled_tf_logits.detach(), unlabeled_logits_tf.detach()
            tap["up2"] = dense_hook._extractor.feature()[-2 * n:].detach().float().cpu()
            tap["conv5"] = mi_enc_hook._extractor.feature()[-2 * n:].detach().float().cpu()
            return torch.zeros((), device=DEV)

    random.seed(11)
    with ep.register_hook(hook(), Spy(name="spy")):
        ep.run()
    torch.cuda.synchronize()
    stats = ep.get_metric()

    theta = torch.from_numpy(AffineAugment().sample(n, tap["seed"])[0])
    # ---- dense InfoNCE (infonce.py:251-279): affine -> projector -> 5 points/image -> SupCon, own class each
    f_u, f_utf = torch.chunk(tap["up2"], 2, 0)
    psd = {k: before[f"_hooks.0._hooks.1._projector.{k}"].cpu() for k in dense_hook._projector.state_dict()}
    z = ol.dense_projection_head(psd, torch.cat([ol.affine_nearest(f_u, theta), f_utf], 0), (7, 7))
    z1, z2 = torch.chunk(z, 2, 0)
    s1, s2 = onr.region_extractor(z1, tap["seed"]), onr.region_extractor(z2, tap["seed"])
    want = ol.supcon_loss(s1, s2, target=list(range(s1.shape[0]))).item()
    got = stats["infonce/Up_conv2/self"]["loss"]
    assert abs(got - want) < 2e-4 * abs(want), (got, want)
    # ---- discrete MI on Up_conv2 (discretemi.py:93-106): 5 sub-heads, IIDSegmentationLoss(padding 0), mean
    csd = [{k: before[f"_hooks.1._hooks.0._hooks.1._projector._headers.{i}.{k}"].cpu() for k in ("0.weight", "0.bias")}
           for i in range(5)]
    probs = onr.dense_cluster_head(csd, torch.cat([ol.affine_nearest(f_u, theta), f_utf], 0))
    want = sum(onr.iid_segmentation_loss(*torch.chunk(p, 2, 0)) for p in probs).item() / 5
    got = stats["discreteMI/up_conv2"]["mi"]
    assert abs(got - want) < 2e-4 * abs(want), (got, want)
    # ---- discrete MI on Conv5 (IIDLoss on pooled vectors)
    c_u, c_utf = torch.chunk(tap["conv5"], 2, 0)
    esd = [{k: before[f"_hooks.1._hooks.0._hooks.0._projector._headers.{i}.{k}"].cpu() for k in ("2.weight", "2.bias")}
           for i in range(5)]
    probs = onr.cluster_head(esd, torch.cat([ol.affine_nearest(c_u, theta), c_utf], 0))
    want = sum(onr.iid_loss(*torch.chunk(p, 2, 0))[0] for p in probs).item() / 5
    got = stats["discreteMI/conv5"]["mi"]  # ~1e-5: three samples carry almost no information (f32 cancellation)
    assert abs(got - want) < 2e-4 * abs(want) + 2e-7, (got, want)
    # ---- IIC on the outputs (midl.py:49-54)
    want = onr.iid_segmentation_loss(tap["tf_logits"].float().cpu().softmax(1), tap["logits_tf"].float().cpu().softmax(1),
                                     lamda=1.5).item()
    got = stats["midl_hook"]["mi"]
    assert abs(got - want) < 2e-4 * abs(want), (got, want)
    # ---- gradients reached every learnable hook parameter through the fused kernels (the flat gradient
    # buffer is intact after step(); RAdam's first, un-normalised steps are too small to test on weights)
    flat = opt._flat[1]
    for p, a, b in zip(flat.params, flat.offsets[:-1], flat.offsets[1:]):
        assert p.__dict__.get("_cy_touched") and bool(flat.grad[a:b].any()), tuple(p.shape)
    assert len(flat.params) == len([k for k in before if before[k].is_floating_point()])


def test_pretrain_encoder_epocher_matches_oracle_and_skips_untouched_parameters():
    """config C5 in small: both views through model(until="Conv5"), InfoNCE on pooled Conv5 features;
    the decoder receives no gradient and must not move (torch.optim.RAdam skips grad-less params)"""
    from contrastyou.arch import UNet
    from contrastyou.hooks.base import TrainerHook
    from contrastyou.losses.kl import KL_div
    from contrastyou.optim import RAdam
    from oracle import losses as ol
    from oracle import unet as ou
    from semi_seg.augment import AffineAugment
    from semi_seg.epochers import PretrainEncoderEpocher
    from semi_seg.hooks import create_infonce_hooks

    class FreeLoader(Loader):
        class dataset:
            class transforms:
                _total_freedom = True

    g = torch.Generator().manual_seed(5)
    n, hw = 6, 32
    sd0 = ou.init_state_dict(1, 4, 128, seed=9)
    psd0 = ol.init_projector_sd(128, 256, 256, seed=10)
    batch = blob_batch(n, hw, 4, g)
    batch["img"] = [batch["img"][0], torch.rand(n, 1, hw, hw, generator=g)]
    model = UNet(input_dim=1, num_classes=4, max_channel=128, momentum=0.01)
    model.load_state_dict(sd0)
    model.to(DEV)
    type(TrainerHook).names.clear()
    hook = create_infonce_hooks(model=model, feature_names="Conv5", weights=1.0, contrast_ons="self", spatial_size=1,
                                data_name="acdc")
    hook._hooks[0]._projector.load_state_dict(psd0)
    hook.to(DEV)
    lr, wd = 1e-3, 1e-2
    opt = RAdam([{"params": list(model.parameters())}, {"params": list(hook.parameters())}], lr=lr, weight_decay=wd)
    ep = PretrainEncoderEpocher(model=model, optimizer=opt, labeled_loader=FreeLoader([batch]),
                                unlabeled_loader=FreeLoader([batch]), sup_criterion=KL_div(), num_batches=1, cur_epoch=0,
                                device=DEV, two_stage=False, disable_bn=False, chain_dataloader=[batch],
                                inference_until="Conv5", scaler=torch.amp.GradScaler("cuda", enabled=False),
                                accumulate_iter=1)
    ep.init()
    assert "sup_loss" not in dict(ep.meters.statistics())["semi"]
    random.seed(4)
    with ep.register_hook(hook()):
        ep.run()
    torch.cuda.synchronize()
    got = ep.get_metric()["semi"]["reg_loss"]

    random.seed(4)
    seed = random.randint(0, int(1e7))
    theta_np, gam_np = AffineAugment().sample(n, seed)
    theta, gam = torch.from_numpy(theta_np), torch.from_numpy(gam_np)
    sd = ou.clone_state_dict(sd0, requires_grad=True)
    psd = {k: v.clone().requires_grad_(True) for k, v in psd0.items()}
    img_tf = ol.affine_nearest(batch["img"][1], theta, gam)
    feats = ou.unet_forward(sd, torch.cat([batch["img"][0], img_tf], 0), training=True, momentum=0.01, until="Conv5")
    f_u, f_utf = torch.chunk(feats, 2, 0)
    z1, z2 = torch.chunk(ol.projection_head(psd, torch.cat([ol.affine_nearest(f_u, theta), f_utf], 0)), 2, 0)
    loss = ol.supcon_loss(z1, z2, target=list(range(n)))
    assert abs(got - loss.item()) < 1e-4 * abs(loss.item()), (got, loss.item())
    loss.backward()
    names = [k for k, v in sd.items() if v.requires_grad]
    oopt = torch.optim.RAdam([{"params": [sd[k] for k in names]}, {"params": list(psd.values())}], lr=lr,
                             weight_decay=wd)
    for k in names:  # what autograd leaves behind for unused parameters
        if sd[k].grad is not None and not sd[k].grad.any():
            sd[k].grad = None
    oopt.step()
    pm = dict(model.named_parameters())
    enc = [k for k in names if k.startswith("_Conv")]
    dec = [k for k in names if not k.startswith("_Conv")]
    assert max(rel(pm[k], sd[k]) for k in enc) < 2e-4
    for k in dec:  # untouched: bit-identical to the initial weights (no weight decay applied)
        assert torch.equal(pm[k].detach().cpu(), sd0[k]), k
    assert max(rel(p, psd[k]) for k, p in hook._hooks[0]._projector.named_parameters()) < 2e-4
    # BN running statistics of the encoder advanced, the decoder's did not
    bufs = dict(model.named_buffers())
    assert int(bufs["_Conv5.conv.1.num_batches_tracked"]) == 1 and int(bufs["_Up_conv5.conv.1.num_batches_tracked"]) == 0


def test_semi_trainer_two_epochs_schedule_checkpoint_resume(tmp_path):
    from contrastyou.arch import UNet
    from contrastyou.hooks.base import TrainerHook
    from contrastyou.losses.kl import KL_div
    from semi_seg.hooks import create_infonce_hooks
    from semi_seg.trainers import trainer_zoo

    g = torch.Generator().manual_seed(8)
    n, hw = 4, 32
    lab = [blob_batch(n, hw, 4, g) for _ in range(2)]
    unl = [blob_batch(n, hw, 4, g) for _ in range(2)]
    val = [{k: v[0] for k, v in blob_batch(n, hw, 4, g, views=1).items()} for _ in range(2)]  # single view
    cfg = {"Optim": {"name": "RAdam", "lr": 1e-5, "weight_decay": 1e-5},
           "Scheduler": {"multiplier": 100, "warmup_max": 2}, "Trainer": {"name": "semi"}}

    def make(save_dir):
        type(TrainerHook).names.clear()
        model = UNet(input_dim=1, num_classes=4, max_channel=128, momentum=0.1)
        torch.manual_seed(0)
        tr = trainer_zoo["semi"](model=model, labeled_loader=Loader(lab), unlabeled_loader=Loader(unl),
                                 val_loader=Loader(val), test_loader=Loader(val), criterion=KL_div(),
                                 save_dir=str(save_dir), max_epoch=3, num_batches=2, device=DEV, disable_bn=False,
                                 two_stage=True, config=cfg, enable_scale=True)
        hook = create_infonce_hooks(model=model, feature_names="Conv5", weights=1.0, contrast_ons="partition",
                                    spatial_size=1, data_name="acdc")
        return tr, hook

    tr, hook = make(tmp_path)
    with tr.register_hook(hook):
        tr.init()
        assert len(tr._optimizer.param_groups) == 2
        tr._max_epoch = 2
        tr.start_training()
    assert (tmp_path / "last.pth").exists() and (tmp_path / ".success").exists()
    assert tr._cur_epoch == 2
    lr_after = tr._optimizer.param_groups[0]["lr"]
    assert abs(lr_after - 1e-5 * 100) < 1e-12  # two warm-up epochs of two: base * multiplier
    rows = (tmp_path / "storage.csv").read_text().strip().splitlines()
    assert len(rows) == 3 and "tra/semi/sup_loss," in rows[0] and "val/eval/dice/DSC_mean" in rows[0]

    # main.py:51-58: a later run seeds its model from a trainer checkpoint
    from contrastyou.utils import extract_model_state_dict
    fresh = UNet(input_dim=1, num_classes=4, max_channel=128)
    fresh.load_state_dict(extract_model_state_dict(str(tmp_path / "last.pth")), strict=True)
    assert all(torch.equal(v.cpu(), tr._model.state_dict()[k].cpu()) for k, v in fresh.state_dict().items())

    tr2, hook2 = make(tmp_path)
    with tr2.register_hook(hook2):
        tr2.init()
        tr2.resume_from_path(str(tmp_path))
        assert tr2._cur_epoch == 2 and tr2._scheduler.last_epoch == 2
        assert abs(tr2._optimizer.param_groups[0]["lr"] - lr_after) < 1e-15
        a, b = tr.state_dict()["module_state"], tr2.state_dict()["module_state"]
        assert all(torch.equal(a[k].cpu(), b[k].cpu()) for k in a)
        tr2._max_epoch = 3  # the checkpoint restored max_epoch = 2 (a persisted Buffer, as in the reference)
        tr2.start_training()  # runs epoch 3
    assert tr2._cur_epoch == 3
    assert len((tmp_path / "storage.csv").read_text().strip().splitlines()) == 4


@pytest.mark.parametrize("accumulate_iter", [1, 2])
def test_hip_graph_replay_of_the_two_passes_equals_eager_steps(accumulate_iter):
    """cyhip.graphed: step 1 probes gradient flow eagerly, step 2 captures, steps 3-4 replay; weights,
    BN statistics and meters after four steps must equal four eager steps from the same state.
    accumulate_iter=2: the capture happens INSIDE a gradient-accumulation window (the first micro-batch's
    gradients are live in .grad while warm-up and capture run) and must not disturb it."""
    from contrastyou.amp import BF16Scaler
    from contrastyou.arch import UNet
    from contrastyou.hooks.base import TrainerHook
    from contrastyou.losses.kl import KL_div
    from contrastyou.optim import RAdam
    from cyhip import graphed
    from oracle import unet as ou
    from semi_seg.epochers import SemiSupervisedEpocher
    from semi_seg.hooks import create_consistency_hook, create_infonce_hooks
    from contrastyou.hooks.base import CombineTrainerHook

    g = torch.Generator().manual_seed(21)
    n, hw, steps = 4, 32, 4
    sd0 = ou.init_state_dict(1, 4, 128, seed=14)
    lab = [blob_batch(n, hw, 4, g) for _ in range(steps)]
    unl = [blob_batch(n, hw, 4, g) for _ in range(steps)]

    def run(use_graph: bool):
        graphed.GRAPH_STEP = use_graph
        type(TrainerHook).names.clear()
        model = UNet(input_dim=1, num_classes=4, max_channel=128, momentum=0.1)
        model.load_state_dict(sd0)
        model.to(DEV)
        torch.manual_seed(3)
        hook = CombineTrainerHook(
            create_infonce_hooks(model=model, feature_names="Conv5", weights=1.0, contrast_ons="partition",
                                 spatial_size=1, data_name="acdc"),
            create_consistency_hook(weight=0.5)).to(DEV)
        opt = RAdam([{"params": list(model.parameters())}, {"params": list(hook.parameters())}], lr=3e-3,
                    weight_decay=1e-4)
        ep = SemiSupervisedEpocher(model=model, optimizer=opt, labeled_loader=Loader(lab), unlabeled_loader=Loader(unl),
                                   sup_criterion=KL_div(), num_batches=steps, cur_epoch=0, device=DEV, two_stage=True,
                                   disable_bn=False, scaler=BF16Scaler(), accumulate_iter=accumulate_iter)
        ep.init()
        random.seed(9)
        with ep.register_hook(hook()):
            ep.run()
        torch.cuda.synchronize()
        replayed = any(isinstance(v, graphed.GraphedTwoPass) for v in model.__dict__.get("_cy_graphed", {}).values())
        return {k: v.detach().float().cpu() for k, v in model.state_dict().items()}, ep.get_metric(), replayed

    default = graphed.GRAPH_STEP
    try:
        sd_e, m_e, rep_e = run(False)
        sd_g, m_g, rep_g = run(True)
    finally:
        graphed.GRAPH_STEP = default
    assert rep_g and not rep_e, "the second run must have captured and replayed the passes"
    for k in sd_e:  # same kernels in the same order on the same data: equal to the last bit
        assert torch.equal(sd_e[k], sd_g[k]), k
    assert m_e["semi"]["sup_loss"] == m_g["semi"]["sup_loss"] and m_e["semi"]["reg_loss"] == m_g["semi"]["reg_loss"]
