"""Data-parallel step on real kernels: two ranks (both on cuda:0, `gloo` collectives -- a 1-GPU box cannot host two
RCCL ranks) each run one SemiSupervisedEpocher + InfoNCE step on their OWN batch through FusedRAdam's bucketed
asynchronous gradient all-reduce.  Checked: (i) replicas that start from different seeds hold bit-identical
parameters after the step (state broadcast + identical reduced gradients); (ii) those parameters equal a
single-process run over the two batches with gradient accumulation (accumulate_iter=2: the same gradient mean, each
batch with its own BatchNorm statistics, which is what per-rank BN means) within f32 re-association."""
import os
import random
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parents[1]


def _setup_paths():
    for p in (REPO, REPO / "contrast-you_amd"):
        if str(p) not in sys.path:
            sys.path.insert(0, str(p))


def _batches(rank_seed, n, hw):
    from tests.test_gpu_hooks_dice import blob_batch
    g = torch.Generator().manual_seed(rank_seed)
    return blob_batch(n, hw, 4, g), blob_batch(n, hw, 4, g)


def _build(model_seed, lr=1e-3):
    from contrastyou.arch import UNet
    from contrastyou.hooks.base import TrainerHook
    from contrastyou.optim import RAdam
    from oracle import unet as ou
    from semi_seg.hooks import create_infonce_hooks
    model = UNet(input_dim=1, num_classes=4, max_channel=128, momentum=0.01)
    model.load_state_dict(ou.init_state_dict(1, 4, 128, seed=model_seed))
    model.to("cuda:0")
    type(TrainerHook).names.clear()
    torch.manual_seed(model_seed)
    hook = create_infonce_hooks(model=model, feature_names="Conv5", weights=1.0, contrast_ons="partition",
                                spatial_size=1, data_name="acdc").to("cuda:0")
    opt = RAdam([{"params": list(model.parameters())}, {"params": list(hook.parameters())}], lr=lr, weight_decay=1e-5)
    return model, hook, opt


def _run(model, hook, opt, lab_batches, unl_batches, accumulate_iter):
    from contrastyou.losses.kl import KL_div
    from semi_seg.epochers import SemiSupervisedEpocher
    from tests.test_gpu_hooks_dice import Loader
    ep = SemiSupervisedEpocher(model=model, optimizer=opt, labeled_loader=Loader(lab_batches),
                               unlabeled_loader=Loader(unl_batches), sup_criterion=KL_div(), num_batches=len(lab_batches),
                               cur_epoch=0, device="cuda:0", two_stage=True, disable_bn=False,
                               scaler=torch.amp.GradScaler("cuda", enabled=False), accumulate_iter=accumulate_iter)
    ep.init()
    with ep.register_hook(hook()):
        ep.run()
    torch.cuda.synchronize()


def _flat(model, hook):
    return torch.cat([p.detach().reshape(-1).float().cpu() for p in list(model.parameters()) + list(hook.parameters())])


def _worker(rank, world, port, q):
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      CY_GRAPH_STEP="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        model, hook, opt = _build(model_seed=100 + rank)  # DIFFERENT initial weights per rank
        assert opt._dp
        lab, unl = _batches(500 + rank, 3, 32)
        random.seed(42)
        for _ in range(rank):
            random.randint(0, int(1e7))  # rank r uses the r-th affine seed of the single-process run
        _run(model, hook, opt, [lab], [unl], accumulate_iter=1)
        q.put((rank, "ok", _flat(model, hook).numpy()))  # (by value: a tensor would travel as a shared-memory handle)
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__)), None))
    finally:
        dist.destroy_process_group()


def test_two_rank_step_equals_accumulated_single_process_step():
    _setup_paths()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = {}
    for _ in procs:
        r, msg, flat = q.get(timeout=600)
        assert msg == "ok", f"rank {r}: {msg}"
        results[r] = torch.from_numpy(flat)
    for p in procs:
        p.join(timeout=120)
    assert torch.equal(results[0], results[1]), "replicas diverged"
    # single process: rank 0's initial weights (the broadcast source), both rank batches, accumulate_iter = 2
    from cyhip import graphed
    was = graphed.GRAPH_STEP
    graphed.GRAPH_STEP = False
    try:
        model, hook, opt = _build(model_seed=100)
        (lab0, unl0), (lab1, unl1) = _batches(500, 3, 32), _batches(501, 3, 32)
        random.seed(42)
        _run(model, hook, opt, [lab0, lab1], [unl0, unl1], accumulate_iter=2)
    finally:
        graphed.GRAPH_STEP = was
    single = _flat(model, hook)
    err = (single - results[0]).abs().max().item()
    assert err < 2e-5 * single.abs().max().item() + 1e-7, err
