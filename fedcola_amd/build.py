"""Build libfedcola_hip.so (gfx950) in-tree with hipcc.  `python -m fedcola_amd.build [--force]`."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libfedcola_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]
HEADERS = glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "fedcola_hip.h")]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(s) > t for s in glob.glob(os.path.join(CSRC, "*")) + HEADERS)


def build(force=False, verbose=True, probes=False):
    """probes=True: the tools build (-DFC_PROBES -> libfedcola_hip_probes.so) that carries the measurement aids (kernel-family
    ablation, GEMM phase knobs).  The product library is built without them."""
    if probes:
        return _build(True, verbose, os.path.join(HERE, "libfedcola_hip_probes.so"), os.path.join(HERE, "build", "probes"), ["-DFC_PROBES"])
    if not force and not needs_build():
        return OUT
    return _build(force, verbose, OUT, os.path.join(HERE, "build"), [])


def _build(force, verbose, OUT, objdir, extra):
    objs, procs = [], []
    os.makedirs(objdir, exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in HEADERS)
    for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_t):
            continue
        cmd = [HIPCC] + FLAGS + extra + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, probes="--probes" in sys.argv))
