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

import numpy as np
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


def _run(model, hook, opt, lab_batches, unl_batches, accumulate_iter, bf16=False):
    from contrastyou.amp import BF16Scaler
    from contrastyou.losses.kl import KL_div
    from semi_seg.epochers import SemiSupervisedEpocher
    from tests.test_gpu_hooks_dice import Loader
    scaler = BF16Scaler() if bf16 else torch.amp.GradScaler("cuda", enabled=False)
    ep = SemiSupervisedEpocher(model=model, optimizer=opt, labeled_loader=Loader(lab_batches),
                               unlabeled_loader=Loader(unl_batches), sup_criterion=KL_div(), num_batches=len(lab_batches),
                               cur_epoch=0, device="cuda:0", two_stage=True, disable_bn=False,
                               scaler=scaler, accumulate_iter=accumulate_iter)
    ep.init()
    with ep.register_hook(hook()):
        ep.run()
    torch.cuda.synchronize()


def _flat(model, hook):
    return torch.cat([p.detach().reshape(-1).float().cpu() for p in list(model.parameters()) + list(hook.parameters())])


def _seed_stream(n):
    """the affine seeds the single-process run draws for n batches (random.seed(42), one randint per batch)"""
    random.seed(42)
    return [random.randint(0, int(1e7)) for _ in range(n)]


def _patch_randint(seeds):
    """the epocher draws its per-batch seed with random.randint: hand it this rank's share of the single-process
    stream instead"""
    it = iter(seeds)
    random.randint = lambda a, b: next(it)  # noqa: E731  (worker process only)


def _worker(rank, world, port, q, mode):
    _setup_paths()
    graph = mode in ("graph", "bf16_buckets")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      CY_GRAPH_STEP="1" if graph else "0",
                      # the opt-in early start of the marked buckets (cyhip.ops.grad_ready_mark) in one of the modes
                      CY_DP_EARLY="1" if mode == "bf16_buckets" else "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        steps = 3 if graph else 1  # graph mode: eager probe step, captured step, replayed step
        if mode == "bf16_buckets":
            from contrastyou.optim.fused_radam import FusedRAdam
            FusedRAdam.BUCKET_ELEMS = 1 << 14  # several buckets in flight inside step()
        model, hook, opt = _build(model_seed=100 + rank)  # DIFFERENT initial weights per rank
        assert opt._dp
        labs, unls = zip(*[_batches(500 + 10 * k + rank, 3, 32) for k in range(steps)])
        _patch_randint(_seed_stream(world * steps)[rank::world])
        _run(model, hook, opt, list(labs), list(unls), accumulate_iter=1, bf16=(mode == "bf16_buckets"))
        if mode == "bf16_buckets":
            assert sum(len(opt._buckets(f)) for f in opt._flat) > 4, "the test wants several buckets in flight"
            # small buckets: some hold only decoder / Conv5 / Conv4 parameters and start on those blocks' marks,
            # before the backward pass (here: the replayed graph) has ended
            assert getattr(opt, "early_buckets", 0) >= steps, "no bucket was reduced ahead of the backward pass's end"
        q.put((rank, "ok", _flat(model, hook).numpy()))  # (by value: a tensor would travel as a shared-memory handle)
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__)), None))
    finally:
        dist.destroy_process_group()


def _spawn(target, args_of_rank, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=args_of_rank(r, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in procs:
        r, msg, payload = q.get(timeout=900)
        assert msg == "ok", f"rank {r}: {msg}"
        results[r] = payload
    for p in procs:
        p.join(timeout=120)
    return results


@pytest.mark.parametrize("mode", ["eager", "graph", "bf16_buckets"])
def test_two_rank_step_equals_accumulated_single_process_step(mode):
    """eager: one step, graph replay off (round 2's case).  graph: three steps with HIP-graph replay ON in the ranks
    -- bucketed asynchronous all-reduce + side-stream joins + two replayed graphs, the combination bench.py --gpus N
    runs.  bf16_buckets: the default bf16 / BF16Scaler path with FusedRAdam.BUCKET_ELEMS small, so
    that step() really pipelines several in-flight buckets against the RAdam launches."""
    _setup_paths()
    port = 29600 + (os.getpid() % 2000) + {"eager": 0, "graph": 1, "bf16_buckets": 2}[mode]
    results = _spawn(_worker, lambda r, q: (r, 2, port, q, mode))
    r0, r1 = torch.from_numpy(results[0]), torch.from_numpy(results[1])
    assert torch.equal(r0, r1), "replicas diverged"
    # single process: rank 0's initial weights (the broadcast source), the ranks' batches interleaved, accumulate_iter = 2
    from cyhip import graphed
    steps = 1 if mode == "eager" else 3
    was = graphed.GRAPH_STEP
    graphed.GRAPH_STEP = False
    try:
        model, hook, opt = _build(model_seed=100)
        labs, unls = [], []
        for k in range(steps):
            for rank in range(2):
                lab, unl = _batches(500 + 10 * k + rank, 3, 32)
                labs.append(lab)
                unls.append(unl)
        random.seed(42)
        _run(model, hook, opt, labs, unls, accumulate_iter=2, bf16=(mode == "bf16_buckets"))
    finally:
        graphed.GRAPH_STEP = was
    single = _flat(model, hook)
    err = (single - r0).abs().max().item()
    # f32: re-association of the two gradient halves only; bf16: the same kernels on the same batches -- also only
    # the order in which the two halves meet in f32
    assert err < 2e-5 * single.abs().max().item() + 1e-7, (mode, err)


def _pretrain_worker(rank, world, port, q):
    """one PretrainEncoderEpocher step (C5 composition) with INFONCEHook(global_negatives=True): every rank projects
    its own batch, all-gathers the embeddings and evaluates the full SupCon matrix"""
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      CY_GRAPH_STEP="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from contrastyou.arch import UNet
        from contrastyou.hooks.base import TrainerHook
        from contrastyou.losses.kl import KL_div
        from contrastyou.optim import RAdam
        from oracle import unet as ou
        from semi_seg.epochers import PretrainEncoderEpocher
        from semi_seg.hooks import create_infonce_hooks
        from tests.test_gpu_hooks_dice import Loader, blob_batch

        class FreeLoader(Loader):
            class dataset:
                class transforms:
                    _total_freedom = True

        g = torch.Generator().manual_seed(900 + rank)
        n, hw = 6, 32
        batch = blob_batch(n, hw, 4, g)
        batch["img"] = [batch["img"][0], torch.rand(n, 1, hw, hw, generator=g)]
        model = UNet(input_dim=1, num_classes=4, max_channel=128, momentum=0.01)
        model.load_state_dict(ou.init_state_dict(1, 4, 128, seed=9 + rank))
        model.to("cuda:0")
        type(TrainerHook).names.clear()
        torch.manual_seed(rank)
        hook = create_infonce_hooks(model=model, feature_names="Conv5", weights=1.0, contrast_ons="self", spatial_size=1,
                                    data_name="acdc", global_negatives=True).to("cuda:0")
        opt = RAdam([{"params": list(model.parameters())}, {"params": list(hook.parameters())}], lr=1e-3)
        ep = PretrainEncoderEpocher(model=model, optimizer=opt, labeled_loader=FreeLoader([batch]),
                                    unlabeled_loader=FreeLoader([batch]), sup_criterion=KL_div(), num_batches=1,
                                    cur_epoch=0, device="cuda:0", two_stage=False, disable_bn=False,
                                    chain_dataloader=[batch], inference_until="Conv5",
                                    scaler=torch.amp.GradScaler("cuda", enabled=False), accumulate_iter=1)
        ep.init()
        random.seed(4)
        with ep.register_hook(hook()):
            ep.run()
        torch.cuda.synchronize()
        loss = float(ep.get_metric()["infonce/Conv5/self"]["loss"])
        q.put((rank, "ok", (loss, _flat(model, hook).numpy())))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__)), None))
    finally:
        dist.destroy_process_group()


def test_two_rank_pretrain_step_with_global_negatives():
    """C5 composition at world size 2 on the GPU: the embedding all-gather inside a real epocher
    step.  Every rank evaluates the same 2*(6+6)-row matrix, so the metered loss is the same number on both (scaled by
    world_size on each); the replicas stay bit-identical; the loss is that of 12 samples, i.e. above the 6-sample one"""
    _setup_paths()
    import math
    port = 29700 + (os.getpid() % 2000)
    results = _spawn(_pretrain_worker, lambda r, q: (r, 2, port, q))
    (l0, f0), (l1, f1) = results[0], results[1]
    assert math.isfinite(l0) and abs(l0 - l1) < 1e-6 * abs(l0), (l0, l1)
    assert np.array_equal(f0, f1), "replicas diverged"
    assert l0 / 2 > math.log(2 * 6 - 1) * 0.5, l0  # (world_size x a 24-row SupCon: ~ log(23) at random init)
