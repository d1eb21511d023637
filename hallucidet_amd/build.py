"""Build libhallucidet_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The library is the C-ABI drop-in boundary declared in include/hallucidet_hip.h.
Objects are cached per source file under hallucidet_amd/lib/obj and rebuilt only
when the source (or a header) is newer.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libhallucidet_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wno-unused-result"]
# sources built a second time with -DHD_STORE_F32 (fp32 activation storage, entry points suffixed _f32: csrc/hd_common.h)
TWICE = ("elementwise.hip", "roi_align.hip", "fcos.hip")


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "hallucidet_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def source_digest():
    """sha256 (12 hex digits) over the HIP sources and headers the library is built from.  The profiling tools stamp it into the PMC
    summaries they write (profiles/*_conv_mfma_util.json, *_conv_traffic.json) and bench.py compares it with the tree it runs on:
    counters collected on another build are reported as stale.  (The GPU boxes have no .git, a commit id cannot be read there.)"""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files.append(os.path.join(HERE, "..", "include", "hallucidet_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def build(verbose=False, force=False, trace=False):
    """trace=True builds lib/libhallucidet_hip_trace.so with -DHD_CONV_TRACE (per-block timeline stamps in the conv kernels,
    tools/conv_trace.py); the product library never carries them."""
    objdir = os.path.join(LIBDIR, "obj_trace" if trace else "obj")
    lib_path = os.path.join(LIBDIR, "libhallucidet_hip_trace.so") if trace else LIB
    flags = FLAGS + (["-DHD_CONV_TRACE"] if trace else []) + os.environ.get("HD_EXTRA_FLAGS", "").split()      # HD_EXTRA_FLAGS: experiment builds
    os.makedirs(objdir, exist_ok=True)
    hm = _headers_mtime()
    jobs = []
    objs = []
    for src in _sources():
        variants = [("", [])] + ([("_f32", ["-DHD_STORE_F32"])] if os.path.basename(src) in TWICE else [])
        for suffix, extra in variants:
            obj = os.path.join(objdir, os.path.basename(src)[:-4] + suffix + ".o")
            objs.append(obj)
            if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hm):
                jobs.append((src, obj, extra))

    def cc(job):
        src, obj, extra = job
        cmd = [HIPCC] + flags + extra + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        return r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for warn in ex.map(cc, jobs):
                if verbose and warn:
                    print(warn)
    if jobs or not os.path.exists(lib_path):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return lib_path


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv, trace="--trace" in sys.argv))
