"""SURVEY section-8 leftovers closed in round 3, on the GPU: `region_extractor` against the reference's own
outputs (tests/golden/round3.npz), SelfPacedSupConLoss with an ignore mask, SelfPacedINFONCEHook (gamma schedule
through two epochs of hooks inside a SemiSupervisedEpocher step) and SuperPixelInfoNCEHook (labels = superpixel ids
under the sampled positions) against the oracle."""
import random

import numpy as np
import pytest
import torch

from test_gpu_hooks_dice import Loader, blob_batch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def close(a, b, rel, what):
    a = a.detach().double().cpu()
    b = torch.as_tensor(b).double()
    err = (a - b).abs().max().item()
    assert err <= rel * (b.abs().max().item() + 1e-30), f"{what}: {err:.3e}"


def test_region_extractor_matches_reference(golden_dir):
    """the HIP row gather behind `region_extractor` picks the vectors the REFERENCE picks (infonce.py:31-46)"""
    from semi_seg.hooks import region_extractor
    g = np.load(golden_dir / "round3.npz")
    n = 0
    for key in g.files:
        if not (key.startswith("re_") and key.endswith("_out")):
            continue
        tag = key[:-4]
        seed = int(tag.split("_s")[-1])
        feat = torch.from_numpy(g[tag + "_feat"]).to(DEV).requires_grad_(True)
        got = region_extractor(feat, point_nums=5, seed=seed)
        assert np.array_equal(got.detach().cpu().numpy(), g[key]), tag
        got.sum().backward()  # adjoint: ones at the sampled positions (5 distinct per image)
        assert feat.grad.sum().item() == got.numel()
        n += 1
    assert n == 6


def test_self_paced_supcon_ignore_mask(golden_dir):
    """mask == 1 positives, mask == 0 negatives, anything else neither (contrastive.py:117-121)"""
    from contrastyou.losses import SelfPacedSupConLoss
    g = np.load(golden_dir / "round3.npz")
    z1, z2, mask = (torch.from_numpy(g[k]).to(DEV) for k in ("spm_z1", "spm_z2", "spm_mask"))
    for mode, gamma in (("hard", 2.5), ("soft", 4.0)):
        a, b = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
        crit = SelfPacedSupConLoss(temperature=0.07, weight_update=mode)
        crit.set_gamma(gamma)
        l = crit(a, b, mask=mask)
        l.backward()
        want = float(g[f"spm_{mode}_loss"])
        assert abs(l.item() - want) < 2e-5 * abs(want), (mode, l.item(), want)
        assert abs(crit.downgrade_ratio - float(g[f"spm_{mode}_ratio"])) < 1e-6
        close(a.grad, g[f"spm_{mode}_dz1"], 2e-4, f"{mode} dz1")
        close(b.grad, g[f"spm_{mode}_dz2"], 2e-4, f"{mode} dz2")


def _semi_epocher(model, hook_module, lab, unl, epoch):
    from contrastyou.losses.kl import KL_div
    from contrastyou.optim import RAdam
    from semi_seg.epochers import SemiSupervisedEpocher
    opt = RAdam([{"params": list(model.parameters())}, {"params": list(hook_module.parameters())}], lr=1e-3)
    ep = SemiSupervisedEpocher(model=model, optimizer=opt, labeled_loader=Loader([lab]), unlabeled_loader=Loader([unl]),
                               sup_criterion=KL_div(), num_batches=1, cur_epoch=epoch, device=DEV, two_stage=True,
                               disable_bn=False, scaler=torch.amp.GradScaler("cuda", enabled=False), accumulate_iter=1)
    ep.init()
    return ep


def test_self_paced_infonce_hook_gamma_schedule_and_loss():
    """SelfPacedINFONCEHook (infonce.py:146-178): epoch e's hook carries gamma(e) of PScheduler, the loss is
    SelfPacedSupConLoss of the projected Conv5 features (oracle), meters sp_weight / age_param are filled"""
    from contrastyou.arch import UNet
    from contrastyou.hooks.base import EpocherHook, TrainerHook
    from oracle import losses as ol
    from oracle import next_rows as onr
    from oracle import unet as ou
    from semi_seg.augment import AffineAugment
    from semi_seg.hooks import PScheduler, create_sp_infonce_hooks

    g = torch.Generator().manual_seed(4)
    n, hw, K = 4, 32, 4
    sd0 = ou.init_state_dict(1, K, 128, seed=5)
    lab, unl = blob_batch(n, hw, K, g), blob_batch(n, hw, K, g)
    unl["img"] = [unl["img"][0], torch.rand(n, 1, hw, hw, generator=g)]
    model = UNet(input_dim=1, num_classes=K, max_channel=128, momentum=0.01)
    model.load_state_dict(sd0)
    model.to(DEV)
    type(TrainerHook).names.clear()
    trainer_hook = create_sp_infonce_hooks(model=model, feature_names="Conv5", weights=0.7, contrast_ons="partition",
                                           data_name="acdc", begin_values=6.0, end_values=2.0, mode="soft", p=0.5,
                                           max_epoch=4, correct_grad=False).to(DEV)
    sp_hook = trainer_hook._hooks[0]
    ref_sched = PScheduler(max_epoch=4, begin_value=6.0, end_value=2.0, p=0.5)
    for epoch in range(2):
        psd = {k: v.detach().cpu().clone() for k, v in sp_hook._projector.state_dict().items()}
        ep = _semi_epocher(model, trainer_hook, lab, unl, epoch)
        tap = {}

        class Spy(EpocherHook):
            def _call_implementation(self, *, seed, partition_group, label_group, **kw):
                tap["seed"], tap["partition"], tap["group"] = seed, list(partition_group), list(label_group)
                tap["conv5"] = sp_hook._extractor.feature()[-2 * n:].detach().float().cpu()
                return torch.zeros((), device=DEV)

        random.seed(21 + epoch)
        with ep.register_hook(trainer_hook(), Spy(name=f"spy{epoch}")):
            ep.run()
        torch.cuda.synchronize()
        stats = ep.get_metric()
        gamma = float(ref_sched.value)
        ref_sched.step()
        key = "spinfoce/Conv5/partition"
        assert abs(stats[key]["age_param"] - gamma) < 1e-9, (epoch, stats[key], gamma)
        theta = torch.from_numpy(AffineAugment().sample(n, tap["seed"])[0])
        f_u, f_utf = torch.chunk(tap["conv5"], 2, 0)
        z = ol.projection_head(psd, torch.cat([ol.affine_nearest(f_u, theta), f_utf], 0))
        z1, z2 = torch.chunk(z, 2, 0)
        target = ol.get_label("partition", "acdc", tap["partition"], tap["group"])
        want, ratio = onr.self_paced_supcon(z1, z2, list(target), gamma=gamma, weight_update="soft")
        assert abs(stats[key]["loss"] - want.item()) < 3e-4 * abs(want.item()), (epoch, stats[key], want.item())
        assert abs(stats[key]["sp_weight"] - ratio) < 1e-4, (epoch, stats[key], ratio)


def test_superpixel_infonce_hook_labels_and_loss():
    """SuperPixelInfoNCEHook (infonce.py:180-194,308-340) inside PretrainDecoderEpocher -- the epocher that hands
    `batch_data` to its hooks (epochers/pretrain.py:40-56): the classes of the sampled vectors are the superpixel
    ids under the sampled positions of the transformed, nearest-resized superpixel map"""
    from contrastyou.arch import UNet
    from contrastyou.hooks.base import TrainerHook
    from contrastyou.losses.kl import KL_div
    from contrastyou.optim import RAdam
    from oracle import losses as ol
    from oracle import next_rows as onr
    from oracle import unet as ou
    from semi_seg.augment import AffineAugment
    from semi_seg.epochers import PretrainDecoderEpocher
    from semi_seg.hooks import create_superpixel_hooks

    g = torch.Generator().manual_seed(8)
    n, hw, K = 3, 48, 4
    sd0 = ou.init_state_dict(1, K, 256, seed=3)
    batch = blob_batch(n, hw, K, g)
    batch["img"] = [batch["img"][0], torch.rand(n, 1, hw, hw, generator=g)]
    # a superpixel map: 4 x 4 blocks of 12 x 12 pixels, ids 0..15 (stored as id / 255 like the reference's loader)
    ids = (torch.arange(hw)[:, None] // 12) * 4 + torch.arange(hw)[None, :] // 12
    batch["superpixel"] = [(ids.float() / 255.0).view(1, 1, hw, hw).repeat(n, 1, 1, 1)]
    model = UNet(input_dim=1, num_classes=K, max_channel=256, momentum=0.01)
    model.load_state_dict(sd0)
    model.to(DEV)
    type(TrainerHook).names.clear()
    trainer_hook = create_superpixel_hooks(model=model, feature_names="Up_conv2", weights=1.0, spatial_size=6,
                                           data_name="acdc").to(DEV)
    sp = trainer_hook._hooks[0]
    psd = {k: v.detach().cpu().clone() for k, v in sp._projector.state_dict().items()}
    opt = RAdam([{"params": list(model.parameters())}, {"params": list(trainer_hook.parameters())}], lr=1e-3)
    ep = PretrainDecoderEpocher(model=model, optimizer=opt, labeled_loader=Loader([batch]), unlabeled_loader=Loader([batch]),
                                sup_criterion=KL_div(), num_batches=1, cur_epoch=0, device=DEV, two_stage=False,
                                disable_bn=False, chain_dataloader=[batch], inference_until="Up_conv2",
                                scaler=torch.amp.GradScaler("cuda", enabled=False), accumulate_iter=1)
    ep.init()
    random.seed(3)
    with ep.register_hook(trainer_hook()):
        ep.run()
    torch.cuda.synchronize()
    got = ep.get_metric()["infonce/Up_conv2/superpixel"]["loss"]

    random.seed(3)
    seed = random.randint(0, int(1e7))
    theta_np, gam_np = AffineAugment().sample(n, seed)
    theta, gam = torch.from_numpy(theta_np), torch.from_numpy(gam_np)
    img_tf = ol.affine_nearest(batch["img"][1], theta, gam)
    feats = ou.unet_forward(sd0, torch.cat([batch["img"][0], img_tf], 0), training=True, momentum=0.01, until="Up_conv2")
    f_u, f_utf = torch.chunk(feats, 2, 0)
    z = ol.dense_projection_head(psd, torch.cat([ol.affine_nearest(f_u, theta), f_utf], 0), (6, 6))
    z1, z2 = torch.chunk(z, 2, 0)
    s1, s2 = onr.region_extractor(z1, seed), onr.region_extractor(z2, seed)
    mask = (batch["superpixel"][0] * 255.0).type(torch.uint8).float()
    pooled = torch.nn.functional.interpolate(ol.affine_nearest(mask, theta), size=(6, 6), mode="nearest")
    labels = onr.region_extractor(pooled, seed).squeeze().type(torch.uint8).tolist()
    assert len(set(labels)) > 1
    want = ol.supcon_loss(s1, s2, target=labels).item()
    assert abs(got - want) < 3e-4 * abs(want), (got, want)
