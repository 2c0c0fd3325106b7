#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: 2D slices/s of the ACDC U-Net + InfoNCE training step.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "C2"): SemiSupervisedEpocher + INFONCEHook@Conv5
(contrast_on=partition, weight 1), two-stage forward, 16 labeled + 16 unlabeled synthetic
1x224x224 slices per rank, 4 classes, UNet(max_channel=512, momentum=0.01), bf16 compute,
RAdam.  One step = forward of 16 + 32 slices, supervised KL loss, InfoNCE loss, backward,
optimizer step (+ gradient all-reduce over RCCL when N > 1).  Inputs are resident in HBM.
`value` = (n_l + n_unl) * world / step_time  (dataset slices consumed per second).

Extra JSON objects (see DESIGN.md "Measurement"):
  roofline     -- dominant conv kernel family: algorithmic FLOP / HIP-event time of its launches,
                  collected in a separate single-stream instrumented pass of the same step AFTER the
                  timed region
  cpu_baseline -- the oracle's CPU restatement of the same step (kind "port") on a bounded sample
"""
from __future__ import annotations

import argparse
import json
import re
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO / "contrast-you_amd"))
sys.path.insert(0, str(REPO))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_BPS = 8.0e12      # HBM3E peak (same table)


class _Transforms:
    _total_freedom = False


class _Dataset:
    transforms = _Transforms()


class SyntheticLoader:
    """infinite iterator over ONE device-resident batch with the reference's collated schema
    (contrastyou/data/dataset/base.py:139-164): img/gt are [view1, view2] lists."""

    dataset = _Dataset()

    def __init__(self, n: int, hw: int, num_classes: int, device, seed: int, tag: str):
        g = torch.Generator().manual_seed(seed)
        img1, img2 = torch.rand(n, 1, hw, hw, generator=g), torch.rand(n, 1, hw, hw, generator=g)
        gt = torch.randint(0, num_classes, (n, 1, hw, hw), generator=g)
        self.batch = {
            "img": [img1.to(device), img2.to(device)], "gt": [gt.to(device), gt.to(device)],
            "filename": [[f"{tag}_{i:04d}" for i in range(n)]] * 2,
            "partition": [[str(i % 3) for i in range(n)]] * 2,
            "scan_num": [[f"patient{i // 3:03d}_{i % 2:02d}" for i in range(n)]] * 2,
        }
        self._n = n

    def __len__(self):
        return 1 << 30

    before_batch = None  # instrumented pass: called before a batch is handed out (= at the start of a step)

    def __iter__(self):
        while True:
            if self.before_batch is not None:
                self.before_batch()
            yield self.batch


def build_step(device, rank: int, n_l: int, n_unl: int, hw: int, max_channel: int, bf16: bool = True,
               num_classes: int = 4, fp16: bool = False):
    from contrastyou.amp import BF16Scaler
    from contrastyou.arch import UNet
    from contrastyou.hooks.base import TrainerHook
    from contrastyou.losses.kl import KL_div
    from contrastyou.optim import RAdam
    from semi_seg.hooks import create_infonce_hooks

    torch.manual_seed(10)
    model = UNet(input_dim=1, num_classes=num_classes, max_channel=max_channel, momentum=0.01).to(device)
    type(TrainerHook).names.clear()  # allow re-building hooks inside one process
    hook = create_infonce_hooks(model=model, feature_names="Conv5", weights=1.0, contrast_ons="partition",
                                spatial_size=1, data_name="acdc").to(device)
    optimizer = RAdam([{"params": list(model.parameters())}, {"params": list(hook.parameters())}],
                      lr=3e-5, weight_decay=1e-5)
    labeled = SyntheticLoader(n_l, hw, num_classes, device, 1234 + rank, "lab")
    unlabeled = SyntheticLoader(n_unl, hw, num_classes, device, 4321 + rank, "unl")
    if fp16:  # the reference's own AMP mode: fp16 autocast + torch GradScaler (contrastyou/amp/amp.py:13-45)
        scaler = torch.amp.GradScaler("cuda", enabled=True)
    else:
        scaler = BF16Scaler() if bf16 else torch.amp.GradScaler("cuda", enabled=False)
    return dict(model=model, hook=hook, optimizer=optimizer, labeled=labeled, unlabeled=unlabeled,
                criterion=KL_div(), scaler=scaler)


def build_step_c5(device, rank: int, n_unl: int, hw: int, max_channel: int):
    """BASELINE config 5: encoder pre-training, `until="Conv5"`, n_unl slices x 2 views per rank, InfoNCE with
    the embeddings of every rank as negatives (all-gather over RCCL when world > 1), decoder frozen
    (main.py:93-100: switch_grad(False, start=until, include_start=False))"""
    from contrastyou.amp import BF16Scaler
    from contrastyou.arch import UNet
    from contrastyou.hooks.base import TrainerHook
    from contrastyou.losses.kl import KL_div
    from contrastyou.optim import RAdam
    from semi_seg.hooks import create_infonce_hooks

    torch.manual_seed(10)
    model = UNet(input_dim=1, num_classes=4, max_channel=max_channel, momentum=0.01).to(device)
    for name in model.decoder_names:
        model.get_module(name).requires_grad_(False)
    type(TrainerHook).names.clear()
    hook = create_infonce_hooks(model=model, feature_names="Conv5", weights=1.0, contrast_ons="partition",
                                spatial_size=1, data_name="acdc", global_negatives=True).to(device)
    params = [p for p in model.parameters() if p.requires_grad]
    optimizer = RAdam([{"params": params}, {"params": list(hook.parameters())}], lr=3e-5, weight_decay=1e-5)
    chain = SyntheticLoader(n_unl, hw, 4, device, 4321 + rank, "unl")
    return dict(model=model, hook=hook, optimizer=optimizer, labeled=chain, unlabeled=chain, chain=chain,
                criterion=KL_div(), scaler=BF16Scaler(), workload="c5")


def run_epoch(ctx, device, num_batches: int, epoch: int):
    if ctx.get("workload") == "c5":
        from semi_seg.epochers import PretrainDecoderEpocher
        ep = PretrainDecoderEpocher(model=ctx["model"], optimizer=ctx["optimizer"], labeled_loader=ctx["labeled"],
                                    unlabeled_loader=ctx["unlabeled"], sup_criterion=ctx["criterion"],
                                    num_batches=num_batches, cur_epoch=epoch, device=device, two_stage=False,
                                    disable_bn=False, chain_dataloader=ctx["chain"], inference_until="Conv5",
                                    scaler=ctx["scaler"], accumulate_iter=1)
        ep.init()
        with ep.register_hook(ctx["hook"]()):
            ep.run()
        return ep
    from semi_seg.epochers import SemiSupervisedEpocher
    ep = SemiSupervisedEpocher(model=ctx["model"], optimizer=ctx["optimizer"], labeled_loader=ctx["labeled"],
                               unlabeled_loader=ctx["unlabeled"], sup_criterion=ctx["criterion"],
                               num_batches=num_batches, cur_epoch=epoch, device=device, two_stage=True,
                               disable_bn=False, scaler=ctx["scaler"], accumulate_iter=1)
    ep.init()
    with ep.register_hook(ctx["hook"]()):
        ep.run()
    return ep


def kernel_roofline(ctx, device, steps: int = 3):
    """instrumented pass: HIP events around every conv3x3 launch (on the launch stream).  The timed
    region overlaps weight-gradient kernels with the rest of the backward pass on a second stream;
    a kernel's roofline fraction is a property of the kernel, so this pass runs the same step with
    every launch on ONE stream (an overlapped kernel's begin-to-end time includes its neighbour's
    share of the CUs).  The committed rocprofv3 stats use the same setting
    (CY_ASYNC_WGRAD=0 CY_TWO_STREAM=0)."""
    from cyhip import _lib, ops
    ops.PROFILE = []
    was = (ops.ASYNC_WGRAD, ops.TWO_STREAM)
    ops.ASYNC_WGRAD = ops.TWO_STREAM = False
    # An eager step is host-bound (the host needs longer to enqueue ~380 launches than the GPU to run them), and an
    # event pair around a launch the GPU is waiting for brackets the host's enqueue latency, not the kernel: every
    # step therefore starts with a one-lane spin kernel that holds the stream while the host enqueues the step
    # behind it, so that the events see back-to-back GPU execution (they then agree with rocprofv3's durations).
    loader = ctx["labeled"]
    if os.environ.get("CY_BENCH_SPIN", "1") != "0" and ctx.get("workload") != "c5":  # (the c5 step is GPU-bound as it is)
        loader.before_batch = lambda: _lib.call("cy_debug_spin", 12000, ops._stream())
    settle = 3  # leading steps not counted: the single-stream mode's first steps allocate (hipMalloc synchronises)
    try:
        run_epoch(ctx, device, settle + steps, 99)
        torch.cuda.synchronize()
    finally:
        ops.ASYNC_WGRAD, ops.TWO_STREAM = was
        loader.before_batch = None
    rec, ops.PROFILE = ops.PROFILE, None
    assert len(rec) % (settle + steps) == 0, "every step makes the same launches"
    rec = rec[len(rec) // (settle + steps) * settle:]
    if os.environ.get("CY_BENCH_DUMP_EVENTS"):  # per-launch table of the counted steps, in launch order
        for kind, flops, e0, e1, *rest in rec[:len(rec) // steps]:
            print(f"[bench] {kind:18s} {flops / 1e9:9.2f} GFLOP {e0.elapsed_time(e1) * 1e3:8.1f} us", file=sys.stderr)
    fam, fam_bytes = {}, {}
    for kind, flops, e0, e1, *rest in rec:
        f = fam.setdefault(kind, [0.0, 0.0, 0])
        f[0] += flops
        f[1] += e0.elapsed_time(e1) * 1e-3
        f[2] += 1
        if rest:
            fam_bytes[kind] = fam_bytes.get(kind, 0.0) + rest[0]
    if not fam:
        return None, {}
    dom = max(fam, key=lambda k: fam[k][1])
    fl, sec, cnt = fam[dom]
    ach = fl / sec / 1e12
    # HBM bytes per launch of the same family from the two PMC passes (FETCH_SIZE, WRITE_SIZE) of this
    # command, summarised by tools/traffic_summary.py into profiles/ (a profiler cannot wrap itself)
    def latest(suffix):
        # profiles/rNN_<workload>_<suffix>.json of the newest round that has one for THIS workload; a PMC pass of
        # another workload's geometry says nothing about this one (None then)
        wl = ctx.get("workload", "c2")
        best = None
        for f in (REPO / "profiles").glob(f"r*_{wl}_{suffix}.json"):
            m = re.match(r"r(\d+)_", f.name)
            if m and (best is None or int(m.group(1)) > best[0]):
                best = (int(m.group(1)), f)
        return (json.load(open(best[1])), best[1].name) if best else ({}, None)

    (traffic_all, traffic_src), (mfma_all, mfma_src) = latest("traffic"), latest("mfma")
    traffic = traffic_all.get(dom, {}).get("hbm_bytes_per_launch")
    avg_s = sec / cnt
    roof = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
            "algorithmic_bytes_per_launch": round(fam_bytes.get(dom, 0.0) / cnt) if fam_bytes.get(dom) else None,
            "launches_per_step": cnt // steps, "avg_launch_ms": round(avg_s * 1e3, 4),
            # north_star: "rocprof HBM GB/s and MFMA-busy reported against MI355X peak" (PMC passes of the same
            # command, profiles/rNN_traffic.json and rNN_mfma.json; durations from this run's HIP events)
            "hbm_gbps": None if traffic is None else round(traffic / avg_s / 1e9, 1),
            "hbm_frac": None if traffic is None else round(traffic / avg_s / PEAK_HBM_BPS, 4),
            "mfma_busy": mfma_all.get(dom, {}).get("mfma_busy"),
            "profile_source": [x for x in (traffic_src, mfma_src) if x] or None}
    detail = {}
    for k, v in fam.items():
        d = {"tflops": round(v[0] / v[1] / 1e12, 2), "ms_per_step": round(v[1] / steps * 1e3, 3),
             "launches_per_step": v[2] // steps, "tflop_per_step": round(v[0] / steps / 1e12, 4),
             "algorithmic_bytes_per_launch": round(fam_bytes[k] / v[2]) if fam_bytes.get(k) else None,
             "hbm_bytes_per_launch": traffic_all.get(k, {}).get("hbm_bytes_per_launch"),
             "mfma_busy": mfma_all.get(k, {}).get("mfma_busy")}
        if d["hbm_bytes_per_launch"]:
            d["hbm_gbps"] = round(d["hbm_bytes_per_launch"] / (v[1] / v[2]) / 1e9, 1)
        detail[k] = d
    return roof, detail


def cpu_baseline(hw: int, max_channel: int):
    """oracle step (CPU restatement, kind=port) on a bounded sample: 2 labeled + 2 unlabeled slices"""
    from oracle import losses as ol
    from oracle import step as ostep
    from oracle import unet as ou
    n = 2
    try:
        threads = len(os.sched_getaffinity(0))
    except AttributeError:
        threads = os.cpu_count() or 1
    threads = max(1, min(threads, 16))  # a 1-GPU box gives this job a 16-core share
    torch.set_num_threads(threads)
    sd = ou.clone_state_dict(ou.init_state_dict(1, 4, max_channel, seed=1), requires_grad=True)
    psd = {k: v.requires_grad_(True) for k, v in ol.init_projector_sd(max_channel, 256, 256, seed=2).items()}
    b = ostep.synthetic_batch(n, n, hw, 4)
    theta = torch.stack([ol.make_theta(1.1, 20.0, 0.05, -0.05, False, i % 2 == 0) for i in range(n)])
    params = [v for v in list(sd.values()) + list(psd.values()) if v.requires_grad]
    opt = torch.optim.RAdam(params, lr=3e-5, weight_decay=1e-5)
    labels = ol.get_label("partition", "acdc", b["partition"], b["scan"])

    def one():
        opt.zero_grad()
        out = ostep.semi_step(sd, psd, labeled_image=b["labeled_image"], labeled_target=b["labeled_target"],
                              unlabeled_image=b["unlabeled_image"],
                              unlabeled_image_tf=ol.affine_nearest(b["unlabeled_image_cf"], theta), theta=theta,
                              labels=labels, momentum=0.01)
        out["total"].backward()
        opt.step()

    one()
    t0, k = time.perf_counter(), 0
    while k < 3 or (time.perf_counter() - t0 < 10.0 and k < 12):
        one()
        k += 1
    dt = (time.perf_counter() - t0) / k
    return {"value": round(2 * n / dt, 3), "unit": "slices/s", "cores": threads, "kind": "port",
            "sample": f"{k} steps of the oracle CPU step at {n}+{n} slices {hw}x{hw}, max_channel={max_channel}, "
                      f"f32, torch {torch.__version__} CPU ops"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=15)
    ap.add_argument("--n-labeled", type=int, default=16)
    ap.add_argument("--n-unlabeled", type=int, default=16)
    ap.add_argument("--hw", type=int, default=224)
    ap.add_argument("--max-channel", type=int, default=512)
    ap.add_argument("--workload", default="c2", choices=["c2", "c4", "c5"],
                    help="c2 (default, the BASELINE metric's config): semi-supervised step; c4: the same step at 8 "
                         "classes, 256 x 256, fp16 autocast + GradScaler; c5: encoder pre-training, 256 slices x 2 "
                         "views per GPU, until=Conv5, global negatives")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # (rehearsal on a one-GPU box: CY_BENCH_BACKEND=gloo puts every rank on the same card -- RCCL refuses two ranks
    #  on one device -- and runs the whole N > 1 code path, collectives included)
    backend = os.environ.get("CY_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    # proof of the collective world the timed steps run in: what torch.distributed itself reports, and
    # an all-reduce of ones over the group the gradient buckets use -- N ranks give N
    dp_proof = None
    if world > 1:
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)
        dp_proof = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                    "allreduce_of_ones": int(round(ones.item()))}

    def note(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:7.1f}s] {msg}", file=sys.stderr, flush=True)

    t_start = time.perf_counter()
    if a.workload == "c5":
        n_c5 = a.n_unlabeled if a.n_unlabeled != 16 else 256
        ctx = build_step_c5(device, rank, n_c5, a.hw, a.max_channel)
        a.n_labeled, a.n_unlabeled = 0, n_c5
        a.no_cpu_baseline = True
    elif a.workload == "c4":
        a.hw = 256 if a.hw == 224 else a.hw
        ctx = build_step(device, rank, a.n_labeled, a.n_unlabeled, a.hw, a.max_channel, num_classes=8, fp16=True)
        a.no_cpu_baseline = True
    else:
        ctx = build_step(device, rank, a.n_labeled, a.n_unlabeled, a.hw, a.max_channel)
    ctx["workload"] = a.workload  # (the PMC summaries under profiles/ are keyed by it)
    note("model / hooks / optimizer built")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_epoch(ctx, device, max(a.warmup, 1), 0)
    barrier()
    note(f"warm-up of {max(a.warmup, 1)} steps done")
    t0 = time.perf_counter()
    ep = run_epoch(ctx, device, a.steps, 1)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    metrics = ep.get_metric()
    note(f"timed {a.steps} steps: {dt / a.steps * 1e3:.2f} ms/step")

    roof, detail = (None, {})
    if not a.no_roofline:
        # EVERY rank runs the instrumented steps: they contain the step's collectives (gradient all-reduce, the
        # embedding all-gather of c5) -- rank 0 alone would wait for partners that have moved on to the barrier
        roof, detail = kernel_roofline(ctx, device)
        note(f"instrumented pass done: {detail}")
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a.hw, a.max_channel)
        note(f"cpu baseline done: {cpu}")
    if world > 1:
        dist.barrier()

    if rank == 0:
        per_step = dt / a.steps
        slices = (a.n_labeled + a.n_unlabeled) * world
        passes = (a.n_labeled + 2 * a.n_unlabeled)
        # algorithmic conv FLOPs actually executed per step (forward of all passes, backward only
        # where a loss reaches: the decoder of the unlabeled pass gets no gradient in this config),
        # summed over the instrumented launches; falls back to the SURVEY formula without them
        flops_step = sum(v["tflop_per_step"] for v in detail.values()) * 1e12 if detail else (
            passes * 75.04e9 * (a.hw / 224.0) ** 2 if (a.max_channel == 512 and a.workload == "c2") else None)
        line = {
            "metric": "2D slices/sec on ACDC U-Net+InfoNCE step", "value": round(slices / per_step, 2),
            "unit": "slices/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(per_step * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16" if a.workload == "c4" else "bf16", "data": "synthetic",
            "config": {"workload": (f"C5: encoder pre-training (PretrainDecoderEpocher, until=Conv5), {a.n_unlabeled} "
                                    f"slices x 2 views 1x{a.hw}x{a.hw} per GPU, InfoNCE over {2 * a.n_unlabeled * world} "
                                    f"embeddings (all-gather), UNet max_channel={a.max_channel}, decoder frozen, RAdam")
                       if a.workload == "c5" else
                       ((f"C4: SemiSupervisedEpocher + InfoNCE@Conv5 (partition), two-stage, fp16 autocast + GradScaler, "
                         f"{a.n_labeled} labeled + {a.n_unlabeled} unlabeled 1x{a.hw}x{a.hw} per GPU, 8 classes, "
                         f"UNet max_channel={a.max_channel}, RAdam") if a.workload == "c4" else
                        ("C2: ACDC SemiSupervisedEpocher + InfoNCE@Conv5 (partition), two-stage, "
                         f"{a.n_labeled} labeled + {a.n_unlabeled} unlabeled 1x{a.hw}x{a.hw} per GPU, "
                         f"4 classes, UNet max_channel={a.max_channel}, RAdam")),
                       "global_batch": slices, "network_passes_per_step_per_gpu": passes,
                       "parallelism": f"dp{world}",
                       # data parallel: gradient buckets per step whose all-reduce started on a "gradients final"
                       # mark inside the backward pass instead of at its end (contrastyou/optim/fused_radam.py)
                       "dp_early_buckets_per_step": (round(getattr(ctx["optimizer"], "early_buckets", 0) / max(
                           1, getattr(ctx["optimizer"], "dp_steps", 1)), 2) if world > 1 else None),
                       "dp": None if dp_proof is None else dict(
                           dp_proof,
                           grad_bytes_reduced_per_step=round(getattr(ctx["optimizer"], "dp_bytes", 0) / max(
                               1, getattr(ctx["optimizer"], "dp_steps", 1))),
                           buckets_per_step=round(getattr(ctx["optimizer"], "dp_buckets", 0) / max(
                               1, getattr(ctx["optimizer"], "dp_steps", 1)), 2),
                           early_start=bool(__import__("cyhip.ops", fromlist=["ops"]).DP_EARLY)),
                       "step_tflops_per_gpu": None if flops_step is None else round(flops_step / 1e12, 3),
                       "achieved_step_tflops_per_gpu": None if flops_step is None else round(
                           flops_step / per_step / 1e12, 1)},
            "roofline": roof, "cpu_baseline": cpu,
            "kernels": detail,
            "losses": {g: {k: v for k, v in d.items() if not isinstance(v, dict)} for g, d in metrics.items()},
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
