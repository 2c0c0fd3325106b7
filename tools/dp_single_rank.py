#!/usr/bin/env python3
"""One process, one GPU, a ONE-rank RCCL group: the data-parallel code path of the C2 step (bucketed all-reduce through
RCCL's own stream, optionally started early on the "gradients final" marks) without a second process sharing the
card.  What a step costs with CY_DP_EARLY=0 / 1, same box:   python tools/dp_single_rank.py"""
import os
import subprocess
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]


def child():
    import torch
    import torch.distributed as dist
    sys.path.insert(0, str(REPO))
    sys.path.insert(0, str(REPO / "contrast-you_amd"))
    import bench
    from cyhip import ops
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{os.environ.get('CY_PORT', '29611')}", world_size=1, rank=0, device_id=dev)
    ctx = bench.build_step(dev, 0, 16, 16, 224, 512)
    ctx["optimizer"]._dp = True
    if os.environ.get("CY_BUCKET_ELEMS"):
        type(ctx["optimizer"]).BUCKET_ELEMS = int(os.environ["CY_BUCKET_ELEMS"])
    ops.marks_wanted = True
    bench.run_epoch(ctx, dev, 15, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.run_epoch(ctx, dev, 40, 1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 40
    opt = ctx["optimizer"]
    print(f"  CY_DP_EARLY={os.environ.get('CY_DP_EARLY')}: {dt * 1e3:.3f} ms/step, early buckets per step "
          f"{getattr(opt, 'early_buckets', 0) / max(1, getattr(opt, 'dp_steps', 1)):.1f}")
    if os.environ.get("CY_DP_EARLY") == "1" and not os.environ.get("CY_BUCKET_ELEMS"):
        # (default bucket size: the buckets are cut at the mark boundaries, so the marked ones do start early)
        assert getattr(opt, "early_buckets", 0) > 0, "CY_DP_EARLY=1 started no bucket early at the default bucket size"
    dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
    else:
        port = [29610]
        for early in ("0", "1", "0", "1"):
            port[0] += 1
            subprocess.run([sys.executable, __file__, "child"],
                           env=dict(os.environ, CY_DP_EARLY=early, CY_PORT=str(port[0])), check=True)
