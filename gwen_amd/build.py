"""Build recipe for libgwen_hip.so (hipcc, gfx950 only, in-tree).

``python -m gwen_amd.build`` or ``__graft_entry__.build()``.  hipcc cross-compiles without a GPU.
-ffp-contract=off: products are rounded before they are added unless the source says fmaf(), which
is what makes K2 bit-identical to a sequential CPU scatter-add ("Numerics").
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgwen_hip.so")
SOURCES = ["api.hip", "prep.hip", "propagate.hip", "linear.hip", "layer.hip", "chain.hip", "forward.hip", "grad.hip",
           "interact.hip", "interact_rows.hip", "interact_bwd.hip", "small.hip", "tiles.hip", "cluster.hip", "wide.hip", "hash.hip", "backward.hip", "loss.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest() -> str:
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for name in sorted(os.listdir(CSRC)) + ["../../include/gwen_hip.h"]:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode()); h.update(f.read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every .hip translation unit to an object and link the shared library."""
    stamp = LIB + ".stamp"
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    objs = []
    for src, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
        if verbose and out:
            print(out.decode())
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout.decode()}")
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB


def build_variant(name: str, defines, sources=None, verbose: bool = False) -> str:
    """An EXPERIMENTAL build beside the product library: ``gwen_amd/variants/libgwen_hip.<name>.so`` with extra
    ``-D`` flags on the listed sources (default: all).  Select it with ``GWEN_HIP_LIB=<path>`` (gwen_amd/_lib.py).
    The product library and its stamp are never touched, so an ablated or diagnostic build ("results are wrong
    by construction") can not leak into tests or bench.py."""
    build()                                              # the unchanged objects come from the product build
    vdir = os.path.join(HERE, "variants")
    odir = os.path.join(vdir, name + ".o")
    os.makedirs(odir, exist_ok=True)
    hipcc = _hipcc()
    changed = list(sources) if sources else list(SOURCES)
    procs, objs = [], []
    for src in SOURCES:
        if src in changed:
            obj = os.path.join(odir, src.replace(".hip", ".o"))
            cmd = [hipcc, *FLAGS, *defines, "-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        else:
            obj = os.path.join(HERE, "build", src.replace(".hip", ".o"))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src} ({' '.join(defines)}):\n{out.decode()}")
    lib = os.path.join(vdir, f"libgwen_hip.{name}.so")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout.decode()}")
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:       # python -m gwen_amd.build --variant NAME [--src a.hip,b.hip] -DX=1 ...
        i = sys.argv.index("--variant")
        srcs = sys.argv[sys.argv.index("--src") + 1].split(",") if "--src" in sys.argv else None
        extra, args = [], sys.argv[1:]
        for k, a in enumerate(args):                      # -D / -f / -m flags, and the value of every "-mllvm"
            if a.startswith(("-D", "-f", "-m")) or (k > 0 and args[k - 1] == "-mllvm"):
                extra.append(a)
        print(build_variant(sys.argv[i + 1], extra, srcs, verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
