"""
Build libnegf_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

``python -m gaunegf_amd.build`` or ``gaunegf_amd.build.build()``.  hipcc
cross-compiles gfx950 code objects without a GPU, so this also runs on a CPU-only
box; objects go to gaunegf_amd/csrc/_build/, the library to gaunegf_amd/lib/.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(CSRC, "_build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libnegf_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "negf.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def build(force=False, verbose=False, extra_flags=()):
    """Compile every .hip translation unit for gfx950 and link the shared library.
    NEGF_EXTRA_HIPCC_FLAGS adds flags for diagnostic builds (e.g. -DRS_STAMPS=1: the chain kernel's phase stamps);
    use it with force=True / --force."""
    extra_flags = tuple(extra_flags) + tuple(os.environ.get("NEGF_EXTRA_HIPCC_FLAGS", "").split())
    os.makedirs(BUILD, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    hdr_time = _deps_mtime()
    jobs = []
    objs = []
    for src in _sources():
        obj = os.path.join(BUILD, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        stale = force or not os.path.exists(obj) or \
            os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time)
        if stale:
            jobs.append([hipcc, *FLAGS, *extra_flags, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        return cmd, r

    failed = False
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for cmd, r in ex.map(run, jobs):
            if r.returncode != 0:
                failed = True
                sys.stderr.write(f"FAILED: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}\n")
            elif verbose and r.stderr.strip():
                sys.stderr.write(r.stderr)
    if failed:
        raise RuntimeError("hipcc failed; see messages above")
    if jobs or force or not os.path.exists(LIB):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs]
        cmd, r = run(cmd)
        if r.returncode != 0:
            raise RuntimeError(f"link failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
