"""Build librepconc_hip.so (gfx950) in-tree with hipcc.  No torch involved: the library is a
plain C-ABI shared object (include/repconc_hip.h).

    python -m repconc_amd.build [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "librepconc_hip.so")
SOURCES = ["pq_distance.hip", "pq_assign_mfma.hip", "sinkhorn.hip", "sinkhorn_f64.hip", "pq_misc.hip", "kmeans.hip", "adc_search.hip", "ivf_lists.hip", "ivf_search.hip", "index.hip", "comm.hip"]
# -ffp-contract=off: the fp32 distance arithmetic must round every sub/mul/add separately
# (bit parity with the torch-CPU oracle); fused multiply-adds are written explicitly where wanted.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-mllvm", "-amdgpu-mfma-vgpr-form", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, "rc_common.h"), os.path.join(CSRC, "adc_common.h"), os.path.join(CSRC, "ivfs_screen16.h"),
               os.path.join(os.path.dirname(HERE), "include", "repconc_hip.h")]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc, *FLAGS, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-8000:]))
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for warn in ex.map(run, jobs):
            if warn and verbose:
                print(warn[-4000:], file=sys.stderr)
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
