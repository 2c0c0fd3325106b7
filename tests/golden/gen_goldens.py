#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own leaf modules on CPU.

Runs only in the build container (needs /root/reference, which never travels to the GPU box):

    python tests/golden/gen_goldens.py

It copies nothing from the reference into the repo: /root/reference is copied to a scratch dir
under /tmp (its packages mkdir at import), no-arithmetic import stubs (loguru, termcolor,
segmentation_models_pytorch, medpy, torch_optimizer) are written next to it, the reference modules are imported
from there, fed seeded inputs/weights produced by oracle.* initialisers, and only numeric
inputs/outputs are saved.
"""
import os
import shutil
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[2]
OUT = Path(__file__).resolve().parent
REF = Path("/root/reference")

STUBS = {
    "loguru/__init__.py": (
        "class _L:\n"
        "    def __getattr__(self, k):\n"
        "        return lambda *a, **k2: self\n"
        "    def catch(self, *a, **k):\n"
        "        return lambda fn: fn\n"
        "logger = _L()\n"),
    "termcolor/__init__.py": "def colored(s, *a, **k):\n    return str(s)\n",
    "segmentation_models_pytorch/__init__.py": (
        "import torch.nn as nn\n"
        "class Unet(nn.Module):\n"
        "    def __init__(self, *a, **k):\n"
        "        super().__init__()\n"),
    "torch_optimizer/__init__.py": "",  # `from torch_optimizer import *` in contrastyou/optim/__init__.py:2
    "medpy/__init__.py": "",
    "medpy/metric/__init__.py": "def assd(*a, **k):\n    raise NotImplementedError\n",
    "medpy/metric/binary.py": "def __surface_distances(*a, **k):\n    raise NotImplementedError\n",
    "skimage/__init__.py": "",  # semi_seg/epochers/helper.py:8 (image dumps only)
    "skimage/io.py": "def imsave(*a, **k):\n    raise NotImplementedError\n",
}


def setup_reference():
    scratch = Path(tempfile.mkdtemp(prefix="cy_ref_"))
    shutil.copytree(REF, scratch / "ref", ignore=shutil.ignore_patterns(".git"))
    for rel, text in STUBS.items():
        p = scratch / "stubs" / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(text)
    sys.path.insert(0, str(scratch / "ref"))
    sys.path.insert(0, str(scratch / "stubs"))
    os.chdir(scratch / "ref")
    return scratch


def npy(t):
    return t.detach().cpu().numpy()


def gen_next_rows(scratch):
    """section-8 "next" rows: dense InfoNCE head with grads, cluster heads, discrete-MI losses,
    GroupNorm+SiLU block, bilinear resize and the warm-up/cosine lr law -> next_rows.npz"""
    import types

    from oracle import losses as ol
    from oracle import next_rows as onr
    from contrastyou.losses.kl import Entropy
    from contrastyou.projectors.heads import ClusterHead, DenseClusterHead, DenseProjectionHead

    # discreteMI.py:17 imports semi_seg.hooks.midl (-> tensorboard, absent); it only needs this
    # module attribute (midl.py:13), so seed it before the import
    midl = types.ModuleType("semi_seg.hooks.midl")
    midl.entropy_criterion = Entropy(reduction="none", eps=1e-8)
    hooks_pkg = types.ModuleType("semi_seg.hooks")
    hooks_pkg.midl = midl
    hooks_pkg.__path__ = []
    sys.modules.setdefault("semi_seg.hooks", hooks_pkg)
    sys.modules.setdefault("semi_seg.hooks.midl", midl)
    from contrastyou.losses.discreteMI import IIDLoss, IIDSegmentationLoss
    from contrastyou.arch.unet2 import Block
    from contrastyou.optim.scheduler import GradualWarmupScheduler

    g = torch.Generator().manual_seed(21)
    out = {}

    # dense projection head (a7), fwd + all grads, non-divisible pooling (13 -> 4, overlapping bins)
    dsd = ol.init_dense_projector_sd(16, 32, 32, seed=4)
    dhead = DenseProjectionHead(input_dim=16, hidden_dim=32, output_dim=32, head_type="mlp", normalize=True,
                                spatial_size=(4, 4))
    dhead.load_state_dict(dsd, strict=True)
    feat = torch.randn(3, 16, 13, 13, generator=g).requires_grad_(True)
    z = dhead(feat)
    (z * torch.linspace(-1, 1, z.numel()).view_as(z)).sum().backward()
    out["dp_feat"], out["dp_z"], out["dp_dfeat"] = npy(feat), npy(z), npy(feat.grad)
    for n, p in dhead.named_parameters():
        out[f"dp_grad_{n}"] = npy(p.grad)

    # cluster heads (a17)
    for dense, cls, tag in ((False, ClusterHead, "ch"), (True, DenseClusterHead, "dch")):
        sds = onr.init_cluster_sds(16, 6, 3, dense, seed=9)
        head = cls(input_dim=16, num_clusters=6, num_subheads=3, head_type="linear", T=1, normalize=False)
        head.load_state_dict({f"_headers.{i}.{k}": v for i, sd in enumerate(sds) for k, v in sd.items()},
                             strict=True)
        x = torch.randn(4, 16, 8, 8, generator=g)
        out[f"{tag}_feat"] = npy(x)
        for i, pr in enumerate(head(x)):
            out[f"{tag}_prob{i}"] = npy(pr)

    # IIDLoss on [n,k] simplex pairs
    a = torch.randn(12, 6, generator=g).softmax(1).requires_grad_(True)
    b = torch.randn(12, 6, generator=g).softmax(1).requires_grad_(True)
    l, l0, pij = IIDLoss(lamb=1.5)(a, b)
    l.backward()
    out["iid_a"], out["iid_b"] = npy(a), npy(b)
    out["iid_loss"], out["iid_loss_nolamb"], out["iid_joint"] = npy(l), npy(l0), npy(pij)
    out["iid_da"], out["iid_db"] = npy(a.grad), npy(b.grad)

    # IIDSegmentationLoss, padding 0 / 1 / 2, symmetric and not
    xa = torch.randn(2, 5, 9, 11, generator=g).softmax(1)
    xb = torch.randn(2, 5, 9, 11, generator=g).softmax(1)
    out["seg_a"], out["seg_b"] = npy(xa), npy(xb)
    for pad in (0, 1, 2):
        for sym in (False, True):
            pa, pb = xa.clone().requires_grad_(True), xb.clone().requires_grad_(True)
            crit = IIDSegmentationLoss(lamda=1.2, padding=pad, symmetric=sym)
            l = crit(pa, pb)
            l.backward()
            t = f"seg_p{pad}_s{int(sym)}"
            out[f"{t}_loss"], out[f"{t}_da"], out[f"{t}_db"] = npy(l), npy(pa.grad), npy(pb.grad)

    # GroupNorm + SiLU block (a18)
    bsd = onr.init_gn_block(16, 32, seed=13)
    blk = Block(16, 32, groups=8)
    blk.load_state_dict(bsd, strict=True)
    x = torch.randn(2, 16, 12, 12, generator=g).requires_grad_(True)
    y = blk(x)
    (y * torch.linspace(-1, 1, y.numel()).view_as(y)).sum().backward()
    out["gn_x"], out["gn_y"], out["gn_dx"] = npy(x), npy(y), npy(x.grad)
    for n, p in blk.named_parameters():
        out[f"gn_grad_{n}"] = npy(p.grad)

    # bilinear resize (a19): the call of semi_seg/hooks/cc.py:132
    img = torch.rand(2, 1, 24, 24, generator=g)
    out["bl_img"] = npy(img)
    for hw in ((7, 7), (12, 10), (48, 40)):
        out[f"bl_{hw[0]}x{hw[1]}"] = npy(torch.nn.functional.interpolate(img, size=hw, mode="bilinear"))

    # lr law: config/base.yaml:10-17 (lr 1e-6... multiplier 300, warmup 10) over 40 epochs
    par = [torch.nn.Parameter(torch.zeros(1))]
    opt = torch.optim.SGD(par, lr=1e-6)
    cos = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=40 - 10, eta_min=1e-7)
    sch = GradualWarmupScheduler(opt, 300, total_epoch=10, after_scheduler=cos)
    lrs = []
    for _ in range(40):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    out["lr_seq"] = np.array(lrs, dtype=np.float64)
    np.savez_compressed(OUT / "next_rows.npz", **out)


def gen_round2(scratch):
    """rows finished in round 2: cluster heads with head_type="mlp" / normalize=True, RedundancyCriterion,
    SelfPacedSupConLoss, UNet2 -> round2.npz (inputs, the reference modules' own random state dicts, outputs,
    gradients)"""
    import types

    from contrastyou.losses.kl import Entropy
    from contrastyou.projectors.heads import ClusterHead, DenseClusterHead
    midl = types.ModuleType("semi_seg.hooks.midl")
    midl.entropy_criterion = Entropy(reduction="none", eps=1e-8)
    hooks_pkg = types.ModuleType("semi_seg.hooks")
    hooks_pkg.midl = midl
    hooks_pkg.__path__ = []
    sys.modules.setdefault("semi_seg.hooks", hooks_pkg)
    sys.modules.setdefault("semi_seg.hooks.midl", midl)
    from contrastyou.arch.unet2 import UNet2
    from contrastyou.losses.contrastive import SelfPacedSupConLoss, SupConLoss1
    from contrastyou.losses.redundancy_reduction import RedundancyCriterion

    g = torch.Generator().manual_seed(77)
    out = {}
    torch.manual_seed(123)
    for dense, cls, tag in ((False, ClusterHead, "ch"), (True, DenseClusterHead, "dch")):
        for head_type, normalize in (("mlp", False), ("mlp", True), ("linear", True)):
            t = f"{tag}_{head_type}_{int(normalize)}"
            kw = dict(hidden_dim=24) if dense else {}
            head = cls(input_dim=16, num_clusters=6, num_subheads=3, head_type=head_type, T=1, normalize=normalize, **kw)
            for k, v in head.state_dict().items():
                out[f"{t}_sd_{k}"] = npy(v)
            x = torch.randn(4, 16, 8, 8, generator=g).requires_grad_(True)
            probs = head(x)
            sum((p * torch.linspace(-1, 1, p.numel()).view_as(p)).sum() for p in probs).backward()
            out[f"{t}_feat"], out[f"{t}_dfeat"] = npy(x), npy(x.grad)
            for i, pr in enumerate(probs):
                out[f"{t}_prob{i}"] = npy(pr)
            for n, p in head.named_parameters():
                out[f"{t}_grad_{n}"] = npy(p.grad)

    xa = torch.randn(2, 5, 9, 11, generator=g).softmax(1)
    xb = torch.randn(2, 5, 9, 11, generator=g).softmax(1)
    out["rr_a"], out["rr_b"] = npy(xa), npy(xb)
    for sym in (False, True):
        for alpha in (0.0, 0.4, 1.0):
            pa, pb = xa.clone().requires_grad_(True), xb.clone().requires_grad_(True)
            l = RedundancyCriterion(symmetric=sym, lamda=1.3, alpha=alpha)(pa, pb)
            l.backward()
            t = f"rr_s{int(sym)}_a{int(alpha * 10)}"
            out[f"{t}_loss"], out[f"{t}_da"], out[f"{t}_db"] = npy(l), npy(pa.grad), npy(pb.grad)

    n = 10
    z1 = torch.nn.functional.normalize(torch.randn(n, 32, generator=g), dim=1)
    z2 = torch.nn.functional.normalize(z1 + 0.4 * torch.randn(n, 32, generator=g), dim=1)
    out["sp_z1"], out["sp_z2"] = npy(z1), npy(z2)
    target = [0, 1, 2, 0, 1, 2, 0, 1, 2, 0]
    out["sp_target"] = np.array(target)
    a, b = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
    ref = SupConLoss1()(a, b, target=target)
    out["sp_supcon_loss"] = npy(ref)
    for mode in ("hard", "soft"):
        for gamma in (1e10, 3.0, 1.5):
            for cg in (False, True):
                a, b = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
                crit = SelfPacedSupConLoss(temperature=0.07, weight_update=mode, correct_grad=cg)
                crit.set_gamma(gamma)
                l = crit(a, b, target=target)
                l.backward()
                t = f"sp_{mode}_g{gamma:g}_c{int(cg)}"
                out[f"{t}_loss"], out[f"{t}_dz1"], out[f"{t}_dz2"] = npy(l), npy(a.grad), npy(b.grad)
                out[f"{t}_ratio"] = np.array(crit.downgrade_ratio)

    torch.manual_seed(321)
    net = UNet2(input_dim=1, num_classes=4, dim=8)
    for k, v in net.state_dict().items():
        out[f"u2_sd_{k}"] = npy(v)
    x = torch.rand(2, 1, 32, 32, generator=g)
    y = net(x)
    (y * torch.linspace(-1, 1, y.numel()).view_as(y)).sum().backward()
    out["u2_x"], out["u2_y"] = npy(x), npy(y)
    for name in ("init_conv.weight", "downs.0.0.block1.proj.weight", "downs.1.1.block2.norm.weight",
                 "downs.2.2.fn.fn.to_qkv.weight", "mid_attn.fn.fn.to_out.weight", "ups.0.0.block1.proj.weight",
                 "ups.2.3.weight", "final_conv.1.bias"):
        out[f"u2_grad_{name}"] = npy(dict(net.named_parameters())[name].grad)
    np.savez_compressed(OUT / "round2.npz", **out)


def gen_round3(scratch):
    """rows pinned in round 3: `region_extractor` (semi_seg/hooks/infonce.py:31-46: the seeded numpy draws of the
    dense InfoNCE hook) and `PScheduler` (infonce.py:58-80: the gamma law of SelfPacedINFONCEHook) -> round3.npz.
    semi_seg/hooks/__init__.py pulls every hook (tensorboard, ...): the package is entered as a bare namespace with
    its real path, so that only infonce.py and what it names are executed; `contrastyou.writer` (tensorboard) is
    pre-seeded with a no-op `get_tb_writer`."""
    import types

    writer = types.ModuleType("contrastyou.writer")
    writer.get_tb_writer = lambda *a, **k: None
    writer.SummaryWriter = type("SummaryWriter", (), {})
    sys.modules["contrastyou.writer"] = writer
    hooks_pkg = types.ModuleType("semi_seg.hooks")
    hooks_pkg.__path__ = [str(scratch / "ref" / "semi_seg" / "hooks")]
    sys.modules["semi_seg.hooks"] = hooks_pkg
    from semi_seg.hooks.infonce import PScheduler, region_extractor

    out = {}
    g = torch.Generator().manual_seed(2024)
    for (h, w) in ((14, 14), (20, 12)):
        for seed in (1, 7, 123456):
            feat = torch.nn.functional.normalize(torch.randn(3, 6, h, w, generator=g), dim=1)
            sel = region_extractor(feat, point_nums=5, seed=seed)
            t = f"re_{h}x{w}_s{seed}"
            out[f"{t}_feat"], out[f"{t}_out"] = npy(feat), npy(sel)
    # the global numpy / torch streams are restored by the reference's context manager: pin that too
    np.random.seed(99)
    before = np.random.get_state()[1][:4].copy()
    region_extractor(torch.zeros(1, 2, 8, 8), point_nums=5, seed=3)
    out["re_state_restored"] = np.array([int((np.random.get_state()[1][:4] == before).all())])
    for tag, kw in (("a", dict(max_epoch=20, begin_value=1e6, end_value=1e6, p=0.5)),
                    ("b", dict(max_epoch=10, begin_value=4.0, end_value=0.5, p=0.5)),
                    ("c", dict(max_epoch=7, begin_value=0.0, end_value=3.0, p=2.0))):
        sch = PScheduler(**kw)
        vals = []
        for _ in range(kw["max_epoch"] + 1):
            vals.append(float(sch.value))
            sch.step()
        out[f"ps_{tag}"] = np.array(vals, dtype=np.float64)
        out[f"ps_{tag}_cfg"] = np.array([kw["max_epoch"], kw["begin_value"], kw["end_value"], kw["p"]], dtype=np.float64)
    # SelfPacedSupConLoss with an explicit mask holding values other than 0 / 1 (mask == 0 are the
    # negatives, contrastive.py:120-121; -1 = neither)
    from contrastyou.losses.contrastive import SelfPacedSupConLoss
    n = 8
    z1 = torch.nn.functional.normalize(torch.randn(n, 16, generator=g), dim=1)
    z2 = torch.nn.functional.normalize(z1 + 0.5 * torch.randn(n, 16, generator=g), dim=1)
    lab = torch.tensor([0, 1, 2, 3, 0, 1, 2, 3])
    mask = torch.eq(lab[:, None], lab[None, :]).float()
    ignore = torch.rand(n, n, generator=g) < 0.3
    ignore = (ignore | ignore.t()) & (mask == 0)
    mask[ignore] = -1.0
    out["spm_z1"], out["spm_z2"], out["spm_mask"] = npy(z1), npy(z2), npy(mask)
    for mode, gamma in (("hard", 2.5), ("soft", 4.0)):
        a, b = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
        crit = SelfPacedSupConLoss(temperature=0.07, weight_update=mode)
        crit.set_gamma(gamma)
        l = crit(a, b, mask=mask)
        l.backward()
        t = f"spm_{mode}"
        out[f"{t}_loss"], out[f"{t}_dz1"], out[f"{t}_dz2"] = npy(l), npy(a.grad), npy(b.grad)
        out[f"{t}_ratio"] = np.array(crit.downgrade_ratio)
    np.savez_compressed(OUT / "round3.npz", **out)


def gen_round4(scratch):
    """option gaps closed in round 4 (VERDICT r03 missing #4) -> round4.npz: the reference's own
    SupConLoss1(exclude_other_pos=True) (contrastyou/losses/contrastive.py:87-91) for class targets and an explicit mask,
    ProjectionHead(pool_name="adaptive_max") and DenseProjectionHead(pool_name="adaptive_max")
    (contrastyou/projectors/heads.py:84-85,102, nn.py:16-23), each with outputs and gradients"""
    from oracle import losses as ol
    from contrastyou.losses.contrastive import SupConLoss1
    from contrastyou.projectors.heads import DenseProjectionHead, ProjectionHead

    g = torch.Generator().manual_seed(41)
    out = {}
    n = 10
    z1 = torch.nn.functional.normalize(torch.randn(n, 32, generator=g), dim=1).requires_grad_(True)
    z2 = torch.nn.functional.normalize(torch.randn(n, 32, generator=g), dim=1).requires_grad_(True)
    out["x_z1"], out["x_z2"] = npy(z1), npy(z2)
    crit = SupConLoss1(exclude_other_pos=True)
    target = [0, 1, 2, 0, 1, 2, 0, 1, 3, 3]
    loss = crit(z1, z2, target=target)
    loss.backward()
    out["x_target"] = np.array(target)
    out["x_loss"], out["x_dz1"], out["x_dz2"] = npy(loss), npy(z1.grad), npy(z2.grad)
    msk = (torch.rand(n, n, generator=g) > 0.6).float()
    msk = ((msk + msk.t() + torch.eye(n)) > 0).float()
    z1.grad = z2.grad = None
    loss = crit(z1, z2, mask=msk)
    loss.backward()
    out["x_mask"], out["x_mask_loss"], out["x_mask_dz1"], out["x_mask_dz2"] = npy(msk), npy(loss), npy(z1.grad), npy(z2.grad)

    psd = ol.init_projector_sd(64, 128, 96, seed=5)
    head = ProjectionHead(input_dim=64, hidden_dim=128, output_dim=96, head_type="mlp", normalize=True,
                          pool_name="adaptive_max")
    head.load_state_dict(psd, strict=True)
    feat = torch.randn(5, 64, 6, 7, generator=g).requires_grad_(True)
    z = head(feat)
    (z * torch.linspace(-1, 1, z.numel()).view_as(z)).sum().backward()
    out["pm_feat"], out["pm_z"], out["pm_dfeat"] = npy(feat), npy(z), npy(feat.grad)
    for k, p_ in head.named_parameters():
        out[f"pm_grad_{k}"] = npy(p_.grad)

    dsd = ol.init_dense_projector_sd(16, 32, 24, seed=6)
    dhead = DenseProjectionHead(input_dim=16, hidden_dim=32, output_dim=24, head_type="mlp", normalize=True,
                                pool_name="adaptive_max", spatial_size=(4, 4))
    dhead.load_state_dict(dsd, strict=True)
    dfeat = torch.randn(2, 16, 13, 13, generator=g).requires_grad_(True)
    dz = dhead(dfeat)
    (dz * torch.linspace(-1, 1, dz.numel()).view_as(dz)).sum().backward()
    out["dm_feat"], out["dm_z"], out["dm_dfeat"] = npy(dfeat), npy(dz), npy(dfeat.grad)
    for k, p_ in dhead.named_parameters():
        out[f"dm_grad_{k}"] = npy(p_.grad)
    np.savez_compressed(OUT / "round4.npz", **out)


def gen_round4_unet2_time(scratch):
    """UNet2(with_time_emb=True) (contrastyou/arch/unet2.py:51-58,101-119,161-173,227-246) -> round4_unet2_time.npz:
    the reference's own state dict, outputs and a sample of parameter gradients for an image batch and a time vector"""
    from contrastyou.arch.unet2 import UNet2

    g = torch.Generator().manual_seed(4242)
    out = {}
    torch.manual_seed(987)
    net = UNet2(input_dim=1, num_classes=4, dim=8, with_time_emb=True)
    for k, v in net.state_dict().items():
        out[f"sd_{k}"] = npy(v)
    x = torch.rand(2, 1, 32, 32, generator=g)
    t = torch.tensor([3.0, 11.0])
    y = net(x, t)
    (y * torch.linspace(-1, 1, y.numel()).view_as(y)).sum().backward()
    out["x"], out["time"], out["y"] = npy(x), npy(t), npy(y)
    params = dict(net.named_parameters())
    for name in ("init_conv.weight", "time_mlp.1.weight", "time_mlp.3.bias", "downs.0.0.mlp.1.weight",
                 "downs.0.0.block1.norm.weight", "downs.0.0.block1.norm.bias", "downs.0.0.block1.proj.bias",
                 "downs.2.1.mlp.1.bias", "mid_block1.mlp.1.weight", "mid_block2.block1.proj.weight",
                 "ups.1.1.mlp.1.weight", "ups.2.0.block1.norm.weight", "final_conv.1.bias"):
        out[f"grad_{name}"] = npy(params[name].grad)
    np.savez_compressed(OUT / "round4_unet2_time.npz", **out)


def main():
    sys.path.insert(0, str(REPO))
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "round4t":
        scratch = setup_reference()
        gen_round4_unet2_time(scratch)
        shutil.rmtree(scratch, ignore_errors=True)
        print("wrote round4_unet2_time.npz")
        return
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "round4":
        scratch = setup_reference()
        gen_round4(scratch)
        shutil.rmtree(scratch, ignore_errors=True)
        print("wrote round4.npz")
        return
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "round3":
        scratch = setup_reference()
        gen_round3(scratch)
        shutil.rmtree(scratch, ignore_errors=True)
        print("wrote round3.npz")
        return
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "round2":
        scratch = setup_reference()
        gen_round2(scratch)
        shutil.rmtree(scratch, ignore_errors=True)
        print("wrote round2.npz")
        return
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "next":
        scratch = setup_reference()
        gen_next_rows(scratch)
        shutil.rmtree(scratch, ignore_errors=True)
        print("wrote next_rows.npz")
        return
    from oracle import losses as ol
    from oracle import unet as ou
    scratch = setup_reference()
    from contrastyou.arch.unet import UNet
    from contrastyou.losses.contrastive import SupConLoss1
    from contrastyou.losses.kl import KL_div
    from contrastyou.meters import UniversalDice
    from contrastyou.projectors.heads import DenseProjectionHead, ProjectionHead
    from contrastyou.utils.general import class2one_hot

    torch.manual_seed(0)
    torch.set_num_threads(4)

    # ------------------------------------------------------------------ U-Net
    sd = ou.init_state_dict(input_dim=1, num_classes=4, max_channel=128, seed=11)
    net = UNet(input_dim=1, num_classes=4, max_channel=128, momentum=0.01)
    missing = net.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 1, 32, 32, generator=g)
    tgt = torch.randint(0, 4, (2, 32, 32), generator=g)
    net.train()
    feats = {}
    hooks = [net.get_module(n).register_forward_hook(lambda m, i, o, n=n: feats.__setitem__(n, o))
             for n in net.arch_elements]
    logits = net(x)
    for h in hooks:
        h.remove()
    loss = KL_div()(logits.softmax(1), class2one_hot(tgt, 4))
    loss.backward()
    out = {"x": npy(x), "target": npy(tgt), "logits": npy(logits), "loss": npy(loss),
           "sd_checksum": np.array([float(sum(v.double().sum() for v in sd.values() if v.is_floating_point()))])}
    for n, f in feats.items():
        out[f"feat_{n}"] = npy(f)
    for n, p in net.named_parameters():
        out[f"grad_{n}"] = npy(p.grad)
    for n, b in net.named_buffers():
        out[f"buf_{n}"] = npy(b)
    net.eval()
    with torch.no_grad():
        out["logits_eval"] = npy(net(x))
        out["conv5_until_eval"] = npy(net(x, until="Conv5"))
    np.savez_compressed(OUT / "unet_small.npz", **out)

    # ------------------------------------------------------------------ heads + losses
    out = {}
    psd = ol.init_projector_sd(128, 256, 256, seed=3)
    head = ProjectionHead(input_dim=128, hidden_dim=256, output_dim=256, head_type="mlp", normalize=True)
    head.load_state_dict(psd, strict=True)
    feat = torch.randn(6, 128, 2, 2, generator=g).requires_grad_(True)
    z = head(feat)
    (z * torch.linspace(-1, 1, z.numel()).view_as(z)).sum().backward()
    out["proj_feat"], out["proj_z"], out["proj_dfeat"] = npy(feat), npy(z), npy(feat.grad)
    for n, p in head.named_parameters():
        out[f"proj_grad_{n}"] = npy(p.grad)

    dsd = ol.init_dense_projector_sd(16, 32, 32, seed=4)
    dhead = DenseProjectionHead(input_dim=16, hidden_dim=32, output_dim=32, head_type="mlp", normalize=True,
                                spatial_size=(4, 4))
    dhead.load_state_dict(dsd, strict=True)
    dfeat = torch.randn(2, 16, 8, 8, generator=g)
    out["dense_feat"], out["dense_z"] = npy(dfeat), npy(dhead(dfeat))

    crit = SupConLoss1()
    n = 8
    z1 = torch.nn.functional.normalize(torch.randn(n, 32, generator=g), dim=1).requires_grad_(True)
    z2 = torch.nn.functional.normalize(torch.randn(n, 32, generator=g), dim=1).requires_grad_(True)
    out["sc_z1"], out["sc_z2"] = npy(z1), npy(z2)
    for tag, target in (("simclr", None), ("partition", [0, 1, 2, 0, 1, 2, 0, 1]),
                        ("patient", [0, 1, 2, 3, 3, 4, 5, 6])):
        z1.grad = z2.grad = None
        l = crit(z1, z2, target=target)
        l.backward()
        out[f"sc_{tag}_loss"] = npy(l)
        out[f"sc_{tag}_dz1"], out[f"sc_{tag}_dz2"] = npy(z1.grad), npy(z2.grad)
        out[f"sc_{tag}_sim_exp"], out[f"sc_{tag}_sim_logits"] = npy(crit.sim_exp), npy(crit.sim_logits)
        out[f"sc_{tag}_pos"], out[f"sc_{tag}_neg"] = npy(crit.pos_mask), npy(crit.neg_mask)
    msk = (torch.rand(n, n, generator=g) > 0.5).float()
    msk = ((msk + msk.t() + torch.eye(n)) > 0).float()
    z1.grad = z2.grad = None
    l = crit(z1, z2, mask=msk)
    l.backward()
    out["sc_mask"], out["sc_mask_loss"], out["sc_mask_dz1"] = npy(msk), npy(l), npy(z1.grad)

    lg = torch.randn(2, 4, 16, 16, generator=g).requires_grad_(True)
    tg = torch.randint(0, 4, (2, 16, 16), generator=g)
    l = KL_div()(lg.softmax(1), class2one_hot(tg, 4))
    l.backward()
    out["kl_logits"], out["kl_target"], out["kl_loss"], out["kl_dlogits"] = npy(lg), npy(tg), npy(l), npy(lg.grad)

    meter = UniversalDice(4, report_axis=[1, 2, 3])
    preds, tgts, groups = [], [], []
    for b in range(3):
        pr = torch.randint(0, 4, (4, 16, 16), generator=g)
        tt = torch.randint(0, 4, (4, 16, 16), generator=g)
        gp = [f"patient{(b * 4 + i) // 3:03d}_00" for i in range(4)]
        meter.add(pr, tt, group_name=gp)
        preds.append(npy(pr)), tgts.append(npy(tt)), groups.append(gp)
    summ = meter.summary()
    out["dice_preds"], out["dice_targets"] = np.stack(preds), np.stack(tgts)
    out["dice_groups"] = np.array(groups)
    out["dice_keys"] = np.array(sorted(summ.keys()))
    out["dice_vals"] = np.array([summ[k] for k in sorted(summ.keys())], dtype=np.float64)
    np.savez_compressed(OUT / "heads_losses.npz", **out)

    gen_next_rows(scratch)
    gen_round2(scratch)
    gen_round3(scratch)
    gen_round4(scratch)
    gen_round4_unet2_time(scratch)
    shutil.rmtree(scratch, ignore_errors=True)
    print("wrote", sorted(p.name for p in OUT.glob("*.npz")))


if __name__ == "__main__":
    main()
