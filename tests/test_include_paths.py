"""Every header under include/gunrock/ compiles when it is the ONLY include of a translation unit (host and gfx950
device pass, syntax only), and every hot-path include path of the reference resolves here -- a user TU that includes
e.g. <gunrock/framework/operators/filter/predicated.hxx> or <gunrock/cuda/atomic_functions.hxx> directly must build as it
does against the reference (SURVEY 2.1 file list; VERDICT r3 missing #5).  CPU only (hipcc cross-compiles)."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

from conftest import ROOT

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
INC = os.path.join(ROOT, "include")

# include paths of the reference's hot-path files -- relative to include/
REFERENCE_HOT_PATH_HEADERS = """
gunrock/algorithms/algorithms.hxx gunrock/algorithms/bfs.hxx gunrock/algorithms/sssp.hxx gunrock/algorithms/pr.hxx
gunrock/algorithms/search/binary_search.hxx
gunrock/framework/framework.hxx gunrock/framework/enactor.hxx gunrock/framework/problem.hxx gunrock/framework/benchmark.hxx
gunrock/framework/frontier/frontier.hxx gunrock/framework/frontier/vector_frontier.hxx gunrock/framework/frontier/configs.hxx
gunrock/framework/operators/operators.hxx gunrock/framework/operators/configs.hxx
gunrock/framework/operators/advance/advance.hxx gunrock/framework/operators/advance/helpers.hxx
gunrock/framework/operators/advance/thread_mapped.hxx gunrock/framework/operators/advance/block_mapped.hxx
gunrock/framework/operators/advance/bucketing.hxx gunrock/framework/operators/advance/merge_path.hxx
gunrock/framework/operators/advance/merge_path_v2.hxx
gunrock/framework/operators/filter/filter.hxx gunrock/framework/operators/filter/predicated.hxx
gunrock/framework/operators/filter/remove.hxx gunrock/framework/operators/filter/bypass.hxx
gunrock/framework/operators/filter/compact.hxx
gunrock/framework/operators/uniquify/uniquify.hxx gunrock/framework/operators/uniquify/unique.hxx
gunrock/framework/operators/uniquify/unique_copy.hxx
gunrock/framework/operators/for/for.hxx gunrock/framework/operators/batch/batch.hxx
gunrock/graph/graph.hxx gunrock/graph/build.hxx gunrock/graph/detail/build.hxx gunrock/graph/detail/base.hxx
gunrock/graph/csr.hxx gunrock/graph/csc.hxx gunrock/graph/properties.hxx gunrock/graph/vertex_pair.hxx
gunrock/formats/formats.hxx gunrock/formats/csr.hxx gunrock/formats/csc.hxx gunrock/formats/coo.hxx
gunrock/io/matrix_market.hxx gunrock/io/parameters.hxx gunrock/io/sample.hxx
gunrock/cuda/cuda.hxx gunrock/cuda/context.hxx gunrock/cuda/launch_box.hxx gunrock/cuda/detail/launch_box.hxx
gunrock/cuda/detail/launch_kernels.hxx gunrock/cuda/atomic_functions.hxx gunrock/cuda/global.hxx gunrock/cuda/sm.hxx
gunrock/cuda/device.hxx gunrock/cuda/function.hxx gunrock/cuda/intrinsics.hxx gunrock/cuda/stream_management.hxx
gunrock/cuda/event_management.hxx
gunrock/container/vector.hxx gunrock/memory.hxx gunrock/error.hxx
gunrock/util/math.hxx gunrock/util/load_store.hxx gunrock/util/type_limits.hxx gunrock/util/timer.hxx
gunrock/util/compare.hxx gunrock/util/print.hxx gunrock/util/filepath.hxx gunrock/util/performance.hxx
""".split()


def _all_headers():
    out = []
    for d, _, files in os.walk(os.path.join(INC, "gunrock")):
        for f in files:
            if f.endswith((".hxx", ".h")):
                out.append(os.path.relpath(os.path.join(d, f), INC))
    return sorted(out)


def _compile_alone(header, tmp):
    src = os.path.join(tmp, header.replace("/", "__") + ".cu")
    with open(src, "w") as f:
        f.write("#include <%s>\nint main() { return 0; }\n" % header)
    r = subprocess.run([HIPCC, "-std=c++17", "--offload-arch=gfx950", "-I" + INC, "-x", "hip", "-fsyntax-only",
                        "-Wno-unused-command-line-argument", src], capture_output=True, text=True)
    return header, r.returncode, r.stderr[-2000:]


def test_reference_hot_path_include_paths_exist():
    missing = [h for h in REFERENCE_HOT_PATH_HEADERS if not os.path.isfile(os.path.join(INC, h))]
    assert not missing, missing
    if os.path.isdir("/root/reference/include/gunrock"):  # this container only: the list names real reference files
        ghosts = [h for h in REFERENCE_HOT_PATH_HEADERS if not os.path.isfile(os.path.join("/root/reference/include", h))]
        assert not ghosts, ghosts


def test_every_header_compiles_alone(tmp_path):
    headers = _all_headers()
    assert len(headers) >= len(REFERENCE_HOT_PATH_HEADERS)
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4)) as ex:
        results = list(ex.map(lambda h: _compile_alone(h, str(tmp_path)), headers))
    bad = [(h, err) for h, rc, err in results if rc != 0]
    assert not bad, "\n\n".join("%s:\n%s" % b for b in bad)
