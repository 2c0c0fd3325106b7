"""The oracle (CPU restatement) against vectors produced by the reference itself
(tests/golden/gen_goldens.py imported /root/reference in the build container)."""
import numpy as np
import torch

from oracle import losses as ol
from oracle import unet as ou

T = torch.from_numpy


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach() if isinstance(a, torch.Tensor) else torch.as_tensor(a)
    b = b.detach() if isinstance(b, torch.Tensor) else torch.as_tensor(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a.double() - b.double()).abs().max().item()
    ref = b.double().abs().max().item()
    assert err <= atol + rtol * ref, f"max err {err:.3e} vs ref scale {ref:.3e}"


def test_unet_forward_backward_matches_reference(golden_dir):
    g = np.load(golden_dir / "unet_small.npz")
    sd = ou.init_state_dict(1, 4, 128, seed=11)
    chk = float(sum(v.double().sum() for v in sd.values() if v.is_floating_point()))
    assert abs(chk - float(g["sd_checksum"][0])) < 1e-6, "seeded init drifted from the fixture's"
    sd = ou.clone_state_dict(sd, requires_grad=True)
    feats = {}
    logits = ou.unet_forward(sd, T(g["x"]), training=True, momentum=0.01, feats=feats)
    close(logits, g["logits"])
    for name, f in feats.items():
        close(f, g[f"feat_{name}"])
    loss = ol.sup_loss(logits, T(g["target"]))
    close(loss, g["loss"], rtol=1e-6)
    loss.backward()
    for k, v in sd.items():
        if v.requires_grad:
            close(v.grad, g[f"grad_{k}"], rtol=2e-4, atol=1e-7)
    for k in sd:
        if "running" in k or "num_batches" in k:
            close(sd[k].float(), T(g[f"buf_{k}"]).float(), rtol=1e-6)
    with torch.no_grad():
        close(ou.unet_forward(sd, T(g["x"]), training=False), g["logits_eval"], rtol=1e-5, atol=1e-5)
        close(ou.unet_forward(sd, T(g["x"]), training=False, until="Conv5"), g["conv5_until_eval"],
              rtol=1e-5, atol=1e-5)


def test_projection_heads_match_reference(golden_dir):
    g = np.load(golden_dir / "heads_losses.npz")
    psd = {k: v.requires_grad_(True) for k, v in ol.init_projector_sd(128, 256, 256, seed=3).items()}
    feat = T(g["proj_feat"]).requires_grad_(True)
    z = ol.projection_head(psd, feat)
    close(z, g["proj_z"])
    (z * torch.linspace(-1, 1, z.numel()).view_as(z)).sum().backward()
    close(feat.grad, g["proj_dfeat"], rtol=1e-4)
    for k, v in psd.items():
        close(v.grad, g[f"proj_grad_{k}"], rtol=1e-4)
    dsd = ol.init_dense_projector_sd(16, 32, 32, seed=4)
    close(ol.dense_projection_head(dsd, T(g["dense_feat"]), (4, 4)), g["dense_z"])


def test_supcon_matches_reference(golden_dir):
    g = np.load(golden_dir / "heads_losses.npz")
    for tag, target in (("simclr", None), ("partition", [0, 1, 2, 0, 1, 2, 0, 1]),
                        ("patient", [0, 1, 2, 3, 3, 4, 5, 6])):
        z1 = T(g["sc_z1"]).requires_grad_(True)
        z2 = T(g["sc_z2"]).requires_grad_(True)
        loss, S, E, pos, neg = ol.supcon_loss(z1, z2, target=target, return_all=True)
        close(loss, g[f"sc_{tag}_loss"], rtol=1e-6)
        loss.backward()
        close(z1.grad, g[f"sc_{tag}_dz1"], rtol=1e-4)
        close(z2.grad, g[f"sc_{tag}_dz2"], rtol=1e-4)
        close(S, g[f"sc_{tag}_sim_logits"], rtol=1e-5, atol=1e-5)
        close(E, g[f"sc_{tag}_sim_exp"], rtol=1e-5)
        close(pos, g[f"sc_{tag}_pos"])
        close(neg, g[f"sc_{tag}_neg"])
    z1 = T(g["sc_z1"]).requires_grad_(True)
    loss = ol.supcon_loss(z1, T(g["sc_z2"]), mask=T(g["sc_mask"]))
    close(loss, g["sc_mask_loss"], rtol=1e-6)
    loss.backward()
    close(z1.grad, g["sc_mask_dz1"], rtol=1e-4)


def test_kl_and_dice_match_reference(golden_dir):
    g = np.load(golden_dir / "heads_losses.npz")
    lg = T(g["kl_logits"]).requires_grad_(True)
    loss = ol.sup_loss(lg, T(g["kl_target"]))
    close(loss, g["kl_loss"], rtol=1e-6)
    loss.backward()
    close(lg.grad, g["kl_dlogits"], rtol=1e-5)
    preds = [T(p) for p in g["dice_preds"]]
    tgts = [T(p) for p in g["dice_targets"]]
    groups = [list(x) for x in g["dice_groups"]]
    summ = ol.dice_summary(preds, tgts, groups, 4, [1, 2, 3])
    for k, v in zip(g["dice_keys"], g["dice_vals"]):
        assert abs(summ[str(k)] - float(v)) < 1e-6, (k, summ[str(k)], v)


def test_label_generators():
    assert ol.get_label("partition", "acdc", ["1", "0", "2", "1"], ["p1_00"] * 4) == [1, 0, 2, 1]
    assert ol.get_label("patient", "acdc", ["0"] * 3, ["patient003_01", "patient001_00", "patient003_00"]) == [1, 0, 1]
    assert ol.get_label("cycle", "acdc", ["0"] * 2, ["patient003_01", "patient001_00"]) == [1, 0]
    assert ol.get_label("self", "acdc", ["0"] * 3, ["a_0"] * 3) == [0, 1, 2]


# ------------------------------------------------------------------ section-8 "next" rows
def test_dense_projection_head_grads_match_reference(golden_dir):
    g = np.load(golden_dir / "next_rows.npz")
    dsd = {k: v.requires_grad_(True) for k, v in ol.init_dense_projector_sd(16, 32, 32, seed=4).items()}
    feat = T(g["dp_feat"]).requires_grad_(True)
    z = ol.dense_projection_head(dsd, feat, (4, 4))
    close(z, g["dp_z"])
    (z * torch.linspace(-1, 1, z.numel()).view_as(z)).sum().backward()
    close(feat.grad, g["dp_dfeat"], rtol=1e-4)
    for k, v in dsd.items():
        close(v.grad, g[f"dp_grad_{k}"], rtol=1e-4)


def test_cluster_heads_and_mi_losses_match_reference(golden_dir):
    from oracle import next_rows as onr
    g = np.load(golden_dir / "next_rows.npz")
    for dense, tag, fn in ((False, "ch", onr.cluster_head), (True, "dch", onr.dense_cluster_head)):
        sds = onr.init_cluster_sds(16, 6, 3, dense, seed=9)
        for i, pr in enumerate(fn(sds, T(g[f"{tag}_feat"]))):
            close(pr, g[f"{tag}_prob{i}"])
    a, b = T(g["iid_a"]).requires_grad_(True), T(g["iid_b"]).requires_grad_(True)
    l, l0, pij = onr.iid_loss(a, b, lamb=1.5)
    close(l, g["iid_loss"]), close(l0, g["iid_loss_nolamb"]), close(pij, g["iid_joint"])
    l.backward()
    close(a.grad, g["iid_da"], rtol=1e-4), close(b.grad, g["iid_db"], rtol=1e-4)
    for pad in (0, 1, 2):
        for sym in (False, True):
            pa, pb = T(g["seg_a"]).requires_grad_(True), T(g["seg_b"]).requires_grad_(True)
            l = onr.iid_segmentation_loss(pa, pb, lamda=1.2, padding=pad, symmetric=sym)
            t = f"seg_p{pad}_s{int(sym)}"
            close(l, g[f"{t}_loss"], rtol=1e-5)
            l.backward()
            close(pa.grad, g[f"{t}_da"], rtol=2e-4, atol=1e-9), close(pb.grad, g[f"{t}_db"], rtol=2e-4, atol=1e-9)


def test_gn_block_bilinear_and_lr_law_match_reference(golden_dir):
    from oracle import next_rows as onr
    g = np.load(golden_dir / "next_rows.npz")
    bsd = {k: v.requires_grad_(True) for k, v in onr.init_gn_block(16, 32, seed=13).items()}
    x = T(g["gn_x"]).requires_grad_(True)
    y = onr.gn_silu_block(x, bsd["proj.weight"], bsd["proj.bias"], bsd["norm.weight"], bsd["norm.bias"])
    close(y, g["gn_y"])
    (y * torch.linspace(-1, 1, y.numel()).view_as(y)).sum().backward()
    close(x.grad, g["gn_dx"], rtol=1e-4)
    for k, v in bsd.items():
        close(v.grad, g[f"gn_grad_{k}"], rtol=1e-4)
    for hw in ((7, 7), (12, 10), (48, 40)):
        close(onr.bilinear_resize(T(g["bl_img"]), hw), g[f"bl_{hw[0]}x{hw[1]}"], rtol=1e-6)
    lrs = onr.warmup_cosine_lrs(1e-6, 300, 10, 40)
    assert np.allclose(lrs, g["lr_seq"], rtol=1e-12, atol=0)


def test_region_points_follow_numpy_choice_stream():
    """a8 has no importable reference (semi_seg.hooks.infonce needs tensorboard): pinned by the
    documented law instead -- seeded numpy stream, rows then columns, without replacement"""
    from oracle import next_rows as onr
    import random
    pts = onr.region_points(4, 20, 20, seed=123)
    state = np.random.get_state()
    np.random.seed(123)
    exp = []
    for _ in range(4):
        hs = np.random.choice(range(20), 5, replace=False)
        ws = np.random.choice(range(20), 5, replace=False)
        exp.append(list(zip(hs.tolist(), ws.tolist())))
    np.random.set_state(state)
    assert pts == exp
    assert all(len({p[0] for p in im}) == 5 and len({p[1] for p in im}) == 5 for im in pts)


# ---------------------------------------------------------------- round-2 rows (tests/golden/round2.npz)
def _sub_sds(g, tag, n_sub):
    """per-sub-head state dicts (keys = the reference Sequential's own indices) from the saved reference state"""
    pre = f"{tag}_sd__headers."
    sds = [dict() for _ in range(n_sub)]
    for k in g.files:
        if k.startswith(pre):
            i, rest = k[len(pre):].split(".", 1)
            sds[int(i)][rest] = torch.from_numpy(g[k])
    return sds


def test_round2_cluster_heads_redundancy_selfpaced(golden_dir):
    from oracle import next_rows as onr
    g = np.load(golden_dir / "round2.npz")
    for dense, tag0 in ((False, "ch"), (True, "dch")):
        for head_type, normalize in (("mlp", False), ("mlp", True), ("linear", True)):
            tag = f"{tag0}_{head_type}_{int(normalize)}"
            probs = onr.cluster_head_general(_sub_sds(g, tag, 3), torch.from_numpy(g[f"{tag}_feat"]), dense=dense,
                                             head_type=head_type, normalize=normalize)
            for i, p in enumerate(probs):
                assert torch.allclose(p, torch.from_numpy(g[f"{tag}_prob{i}"]), rtol=1e-5, atol=1e-6), (tag, i)
    xa, xb = torch.from_numpy(g["rr_a"]), torch.from_numpy(g["rr_b"])
    for sym in (False, True):
        for alpha in (0.0, 0.4, 1.0):
            t = f"rr_s{int(sym)}_a{int(alpha * 10)}"
            pa, pb = xa.clone().requires_grad_(True), xb.clone().requires_grad_(True)
            l = onr.redundancy_criterion(pa, pb, alpha=alpha, lamda=1.3, symmetric=sym)
            l.backward()
            assert abs(l.item() - float(g[f"{t}_loss"])) < 1e-5 * abs(float(g[f"{t}_loss"])), t
            assert torch.allclose(pa.grad, torch.from_numpy(g[f"{t}_da"]), rtol=1e-4, atol=1e-7), t
    z1, z2 = torch.from_numpy(g["sp_z1"]), torch.from_numpy(g["sp_z2"])
    target = g["sp_target"].tolist()
    for mode in ("hard", "soft"):
        for gamma in (1e10, 3.0, 1.5):
            for cg in (False, True):
                t = f"sp_{mode}_g{gamma:g}_c{int(cg)}"
                a, b = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
                l, ratio = onr.self_paced_supcon(a, b, target, gamma=gamma, weight_update=mode, correct_grad=cg)
                l.backward()
                assert abs(l.item() - float(g[f"{t}_loss"])) < 1e-5 * abs(float(g[f"{t}_loss"])), t
                assert abs(ratio - float(g[f"{t}_ratio"])) < 1e-6, t
                assert torch.allclose(a.grad, torch.from_numpy(g[f"{t}_dz1"]), rtol=1e-4, atol=1e-6), t
    # the one numeric identity the reference's own files hold (contrastive.py:241-248): gamma -> inf == SupConLoss1
    assert abs(float(g["sp_soft_g1e+10_c0_loss"]) - float(g["sp_supcon_loss"])) < 1e-6


# ---------------------------------------------------------------- round-3 rows (tests/golden/round3.npz)
def test_region_extractor_matches_reference(golden_dir):
    """`region_extractor` of the REFERENCE (semi_seg/hooks/infonce.py:31-46, imported by gen_goldens.py --only
    round3) on 3 seeds x 2 grid sizes: the oracle's restatement picks the same vectors (the row
    was pinned to the numpy-stream law only)"""
    from oracle import next_rows as onr
    g = np.load(golden_dir / "round3.npz")
    n = 0
    for key in g.files:
        if not key.endswith("_out") or not key.startswith("re_"):
            continue
        tag = key[:-4]
        seed = int(tag.split("_s")[-1])
        feat = torch.from_numpy(g[tag + "_feat"])
        got = onr.region_extractor(feat, seed)
        assert np.array_equal(got.numpy(), g[key]), tag
        n += 1
    assert n == 6
    assert int(g["re_state_restored"][0]) == 1


def test_pscheduler_matches_reference(golden_dir):
    """gamma law of SelfPacedINFONCEHook (infonce.py:58-80): the product's PScheduler against the reference's
    sequences"""
    import sys
    sys.path.insert(0, str(golden_dir.parents[1] / "contrast-you_amd"))
    from semi_seg.hooks.infonce import PScheduler
    g = np.load(golden_dir / "round3.npz")
    for tag in ("a", "b", "c"):
        max_epoch, begin, end, p = g[f"ps_{tag}_cfg"]
        sch = PScheduler(max_epoch=int(max_epoch), begin_value=begin, end_value=end, p=p)
        vals = []
        for _ in range(int(max_epoch) + 1):
            vals.append(float(sch.value))
            sch.step()
        assert np.array_equal(np.array(vals), g[f"ps_{tag}"]), tag


def test_self_paced_supcon_with_ignore_mask_matches_reference(golden_dir):
    """mask values other than 0 / 1 are neither positives nor negatives (contrastive.py:117-121)"""
    from oracle import next_rows as onr
    g = np.load(golden_dir / "round3.npz")
    z1, z2, mask = (torch.from_numpy(g[k]) for k in ("spm_z1", "spm_z2", "spm_mask"))
    assert (mask == -1).any()
    for mode, gamma in (("hard", 2.5), ("soft", 4.0)):
        a, b = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
        l, ratio = onr.self_paced_supcon(a, b, gamma=gamma, weight_update=mode, mask=mask)
        l.backward()
        close(l, g[f"spm_{mode}_loss"], rtol=2e-5)
        assert abs(ratio - float(g[f"spm_{mode}_ratio"])) < 1e-6
        close(a.grad, g[f"spm_{mode}_dz1"], rtol=1e-4, atol=1e-6)
        close(b.grad, g[f"spm_{mode}_dz2"], rtol=1e-4, atol=1e-6)


def test_round4_option_gaps_oracle_vs_reference(golden_dir):
    """SupConLoss1(exclude_other_pos=True) and the adaptive_max pooling of both projection heads: the oracle's
    restatements against the reference's own outputs and gradients (tests/golden/round4.npz)"""
    import numpy as np
    from oracle import losses as ol
    g = np.load(golden_dir / "round4.npz")
    z1 = torch.from_numpy(g["x_z1"]).requires_grad_(True)
    z2 = torch.from_numpy(g["x_z2"]).requires_grad_(True)
    loss = ol.supcon_loss_exclude_pos(z1, z2, target=g["x_target"].tolist())
    loss.backward()
    assert abs(loss.item() - float(g["x_loss"])) < 1e-6
    assert np.allclose(z1.grad.numpy(), g["x_dz1"], atol=1e-6) and np.allclose(z2.grad.numpy(), g["x_dz2"], atol=1e-6)
    z1.grad = z2.grad = None
    loss = ol.supcon_loss_exclude_pos(z1, z2, mask=torch.from_numpy(g["x_mask"]))
    loss.backward()
    assert abs(loss.item() - float(g["x_mask_loss"])) < 1e-6
    assert np.allclose(z1.grad.numpy(), g["x_mask_dz1"], atol=1e-6)
    psd = ol.init_projector_sd(64, 128, 96, seed=5)
    z = ol.projection_head(psd, torch.from_numpy(g["pm_feat"]), pool="adaptive_max")
    assert np.allclose(z.numpy(), g["pm_z"], atol=1e-6)
    dsd = ol.init_dense_projector_sd(16, 32, 24, seed=6)
    dz = ol.dense_projection_head(dsd, torch.from_numpy(g["dm_feat"]), (4, 4), pool="adaptive_max")
    assert np.allclose(dz.numpy(), g["dm_z"], atol=1e-6)
