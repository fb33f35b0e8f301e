"""Build libgrx.so (the C-ABI engine) for gfx950 with hipcc, in-tree.

    python -m gunrock_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting gunrock_amd/libgrx.so is
git-ignored but travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libgrx.so")
OBJ = os.path.join(HERE, "_obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

SOURCES = ["grx_api.hip", "grx_bfs.hip", "grx_sssp.hip", "grx_pr.hip", "grx_transpose.hip", "grx_dist_sssp.hip", "grx_host.cpp"]
# The block-asynchronous relaxation for road-like graphs (grx_block.hip, 1000 lines; opt-in GRX_BLOCK=1, measured no better than the
# level-synchronous kernels: DESIGN.md 3.7) is NOT part of the default library since round 6.  `python -m gunrock_amd.build
# --with-block` builds gunrock_amd/libgrx_block.so (all of the above + that file, -DGRX_WITH_BLOCK); its tests run against it with
# GRX_TEST_BLOCK=1 (tests/test_block_variant.py).
BLOCK_SOURCES = SOURCES + ["grx_block.hip"]
FLAGS = ["-std=c++17", "-O3", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics",
         "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-I" + os.path.join(ROOT, "include")]


def source_sha():
    """Hash of everything that decides what the engine's kernels are: gunrock_amd/csrc/*, the C-ABI header and the
    compiler flags.  Committed rocprofv3 PMC numbers carry it; bench.py attaches them only while it still matches."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "grx.h"), "rb").read())
    h.update(" ".join(f for f in FLAGS if not f.startswith("-I")).encode())
    return h.hexdigest()[:16]


def _deps():
    out = [os.path.join(ROOT, "include", "grx.h")]
    for f in os.listdir(CSRC):
        out.append(os.path.join(CSRC, f))
    return out


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


def _compile(src):
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ, src + ".o")
    newest = max(os.path.getmtime(d) for d in _deps() if d.endswith((".hpp", ".h")) or d == path)
    if os.path.exists(obj) and os.path.getmtime(obj) >= newest:
        return obj
    cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build_variant(name, defines, verbose=False, sources=None):
    """A second build of the same sources with extra -D flags, as gunrock_amd/libgrx_<name>.so (tuning aids only:
    e.g. `timers` = -DGRX_MID_TIMERS, the per-phase clocks of the multi-level body).  Loaded through GRX_LIB_PATH."""
    lib = os.path.join(HERE, "libgrx_%s.so" % name)
    obj_dir = os.path.join(HERE, "_obj_" + name)
    if os.path.exists(lib) and all(os.path.getmtime(d) <= os.path.getmtime(lib) for d in _deps()):
        return lib
    os.makedirs(obj_dir, exist_ok=True)

    def one(src):
        path = os.path.join(CSRC, src)
        obj = os.path.join(obj_dir, src + ".o")
        cmd = [HIPCC] + FLAGS + ["-D" + d for d in defines] + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        return obj

    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(one, sources or SOURCES))
    r = subprocess.run([HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib] + objs + ["-lpthread", "-ldl"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built", lib)
    return lib


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs + ["-lpthread", "-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    if "--timers" in sys.argv:
        build_variant("timers", ["GRX_MID_TIMERS"], verbose=True)
    elif "--with-block" in sys.argv:
        build_variant("block", ["GRX_WITH_BLOCK"], verbose=True, sources=BLOCK_SOURCES)
    elif "--fine-timers" in sys.argv:  # every sub-phase of the many-levels body behind a full wait (tools/mid_phases.py)
        build_variant("fine", ["GRX_MID_TIMERS=2"], verbose=True)
    else:
        build(force="--force" in sys.argv, verbose=True)
