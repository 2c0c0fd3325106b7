#!/usr/bin/env python3
"""Which torch (ATen) operators run on CUDA tensors inside a C2 step, with their Python call sites: the step's own
kernels go through the C ABI, so everything listed here is glue that could be folded away.
    python tools/torch_ops_in_step.py"""
import collections
import os
import sys
import traceback
from pathlib import Path

import torch
from torch.utils._python_dispatch import TorchDispatchMode

os.environ.setdefault("CY_GRAPH_STEP", "0")
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "contrast-you_amd"))
import bench  # noqa: E402

SKIP = ("aten::view", "aten::_unsafe_view", "aten::detach", "aten::alias", "aten::permute", "aten::select", "aten::slice",
        "aten::empty", "aten::as_strided", "aten::t", "aten::transpose", "aten::expand", "aten::reshape", "aten::split",
        "aten::unsqueeze", "aten::squeeze", "aten::is_", "aten::size", "aten::stride", "aten::_has", "aten::lift",
        "aten::record_stream", "aten::set_", "aten::resize_", "aten::chunk", "aten::unbind", "aten::narrow", "aten::sym_",
        "aten::is_pinned", "aten::_pin_memory", "aten::equal", "aten::new_empty")
counts = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func._schema.name
        flat = [a for a in list(args) + list((kwargs or {}).values()) if isinstance(a, torch.Tensor)]
        flat += [b for a in args if isinstance(a, (list, tuple)) for b in a if isinstance(b, torch.Tensor)]
        if any(t.is_cuda for t in flat) and not name.startswith(SKIP):
            st = [f for f in traceback.extract_stack() if "/repo/" in f.filename and "tools/" not in f.filename]
            site = f"{Path(st[-1].filename).name}:{st[-1].lineno} {st[-1].name}" if st else "?"
            counts[(name, site)] += 1
        return func(*args, **(kwargs or {}))


dev = torch.device("cuda", 0)
ctx = bench.build_step(dev, 0, 16, 16, 224, 512)
bench.run_epoch(ctx, dev, 4, 0)
torch.cuda.synchronize()
with Log():
    bench.run_epoch(ctx, dev, 2, 1)
torch.cuda.synchronize()
for (n, s), c in sorted(counts.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f"{c / 2:5.1f}/step  {n:30s} {s}")
