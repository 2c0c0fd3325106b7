"""world_size-2 `gloo` tests (CPU) of the data-parallel plumbing: flat-buffer gradient all-reduce
inside FusedRAdam and the embedding all-gather with local-only autograd."""
import os
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = Path(__file__).resolve().parents[1]


def _worker(rank: int, world: int, port: int, q):
    sys.path.insert(0, str(REPO / "contrast-you_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from contrastyou.optim.fused_radam import FusedRAdam
        from cyhip import parallel
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
        head = torch.nn.Linear(3, 2)
        opt = FusedRAdam([{"params": list(net.parameters())}, {"params": list(head.parameters())}], lr=1e-3)
        assert opt._dp, "data-parallel mode must switch on by itself when world_size > 1"
        opt.zero_grad()
        # parameters now live in one flat buffer per group
        p0 = next(net.parameters())
        assert p0.data_ptr() == opt._flat[0].data.data_ptr()
        x = torch.randn(4, 5, generator=torch.Generator().manual_seed(100 + rank))
        loss = head(net(x)).pow(2).mean()
        loss.backward()
        local = [f.grad.clone() for f in opt._flat]
        opt.all_reduce_grads()
        gathered = [[torch.empty_like(g) for _ in range(world)] for g in local]
        for g, buf in zip(local, gathered):
            dist.all_gather(buf, g)
        for f, buf in zip(opt._flat, gathered):
            assert torch.allclose(f.grad, torch.stack(buf).mean(0), atol=1e-7)
        # embedding all-gather: values from every rank, gradient only into the local rows
        z = torch.full((3, 4), float(rank + 1), requires_grad=True)
        zg = parallel.gather_cat(z)
        assert zg.shape == (3 * world, 4)
        assert torch.equal(zg[:3], torch.full((3, 4), 1.0)) and torch.equal(zg[3:], torch.full((3, 4), 2.0))
        (zg * torch.arange(zg.numel()).view_as(zg).float()).sum().backward()
        expect = torch.arange(zg.numel()).view_as(zg).float()[3 * rank: 3 * rank + 3]
        assert torch.equal(z.grad, expect)
        # replicas that were initialised differently (per-rank seeds, a checkpoint resumed on one rank only)
        # are made identical when the flat buffers are built: rank 0's parameters and moments win
        torch.manual_seed(1234 + rank)
        net2 = torch.nn.Linear(6, 4)
        mine = torch.cat([p.detach().reshape(-1).clone() for p in net2.parameters()])
        opt2 = FusedRAdam(net2.parameters(), lr=1e-3)
        opt2.zero_grad()
        flat = opt2._flat[0].data
        both = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        assert torch.equal(both[0], both[1]), "parameters differ across ranks after the broadcast"
        assert torch.equal(flat, mine) == (rank == 0)
        # InfoNCE with global negatives (BASELINE config 5): every rank evaluates the loss on the embeddings of
        # ALL ranks (gather_cat: values from everywhere, autograd into the local rows), scaled by world_size;
        # the gradient MEAN the optimizer takes over ranks must equal the gradient of the single-process loss
        # on the union batch (semi_seg/hooks/infonce.py uses exactly this composition)
        sys.path.insert(0, str(REPO))
        from oracle.losses import supcon_loss
        torch.manual_seed(5)
        enc = torch.nn.Linear(6, 8).double()
        gx = torch.Generator().manual_seed(77)
        xs = [torch.randn(4, 6, generator=gx, dtype=torch.float64) for _ in range(2 * world)]  # (view1, view2) per rank
        norm = torch.nn.functional.normalize
        z1 = norm(enc(xs[2 * rank]), dim=1)
        z2 = norm(enc(xs[2 * rank + 1]), dim=1)
        tgt = parallel.gather_labels([(rank * 4 + i) % 3 for i in range(4)])
        loss = supcon_loss(parallel.gather_cat(z1), parallel.gather_cat(z2), target=tgt) * world
        enc.zero_grad()
        loss.backward()
        gdp = [p.grad.clone() for p in enc.parameters()]
        for t_ in gdp:
            dist.all_reduce(t_)
            t_.div_(world)
        enc.zero_grad()
        z1a = norm(enc(torch.cat([xs[2 * r] for r in range(world)])), dim=1)
        z2a = norm(enc(torch.cat([xs[2 * r + 1] for r in range(world)])), dim=1)
        single = supcon_loss(z1a, z2a, target=[(r * 4 + i) % 3 for r in range(world) for i in range(4)])
        single.backward()
        assert abs(loss.item() / world - single.item()) < 1e-12
        for a_, p_ in zip(gdp, enc.parameters()):
            assert torch.allclose(a_, p_.grad, rtol=1e-10, atol=1e-12), (a_ - p_.grad).abs().max()
        labels = parallel.gather_labels([f"r{rank}_{i}" for i in range(2)])
        assert labels == ["r0_0", "r0_1", "r1_0", "r1_1"]
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))
    finally:
        dist.destroy_process_group()


def test_gloo_world2_grad_allreduce_and_embedding_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r, msg in results:
        assert msg == "ok", f"rank {r}: {msg}"
