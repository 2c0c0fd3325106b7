"""Build libcontrastyou_hip.so (gfx950 only) from csrc/*.hip with hipcc.

    python contrast-you_amd/build.py [--force] [--jobs N]

hipcc cross-compiles without a GPU.  Objects go to contrast-you_amd/build/, the shared
library to contrast-you_amd/lib/ (both git-ignored; the .so travels with gpurun snapshots).
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OBJ = HERE / "build"
LIB = HERE / "lib" / "libcontrastyou_hip.so"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-unused-but-set-variable"] + os.environ.get("CY_HIPCC_EXTRA", "").split()


# per-file flags (kept empty unless a file needs one; CY_HIPCC_EXTRA_<STEM> adds to a single file for experiments)
PER_FILE = {}


def _file_flags(src: Path):
    return PER_FILE.get(src.stem, []) + os.environ.get("CY_HIPCC_EXTRA_" + src.stem.upper(), "").split()


def _deps_mtime() -> float:
    hdrs = list(CSRC.glob("*.h")) + [HERE.parent / "include" / "contrastyou_hip.h"]
    return max(p.stat().st_mtime for p in hdrs)


def _compile(src: Path, force: bool) -> Path:
    obj = OBJ / (src.stem + ".o")
    if not force and obj.exists() and obj.stat().st_mtime > max(src.stat().st_mtime, _deps_mtime()):
        return obj
    cmd = [HIPCC, *FLAGS, *_file_flags(src), "-c", str(src), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
    return obj


def build(force: bool = False, jobs: int = 4) -> Path:
    OBJ.mkdir(exist_ok=True)
    LIB.parent.mkdir(exist_ok=True)
    srcs = sorted(CSRC.glob("*.hip"))
    if not srcs:
        raise RuntimeError("no .hip sources found")
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    newest = max(o.stat().st_mtime for o in objs)
    if force or not LIB.exists() or LIB.stat().st_mtime < newest:
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", *map(str, objs), "-o", str(LIB)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=4)
    a = ap.parse_args()
    print(build(a.force, a.jobs))
    sys.exit(0)
