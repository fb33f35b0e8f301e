"""CLI drivers (bin/bfs, bin/sssp, bin/pr): the reference's command lines and
stdout contract.  CPU part: they exist / print help / reject unknown options
without touching a GPU.  GPU part: --validate reports zero errors for every
load-balance / filter combination, on the engine path, on the generic operator
path (--generic_operators and the header-only builds), and for the REFERENCE's
own driver and algorithm sources compiled against our headers (tests/dropin)."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

BIN = os.path.join(ROOT, "bin")
DROPIN = os.path.join(ROOT, "tests", "dropin", "_build")
CHES = os.path.join(GOLDEN, "chesapeake.mtx")
CHES_HEAD = "0 2 2 2 2 2 1 1 2 2 1 1 1 2 2 2 2 2 2 2 2 1 1 2 2 2 2 2 2 2 2 2 2 1 1 2 1 2 1"


def run(cmd, check=True):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if check:
        assert r.returncode == 0, (cmd, r.stdout[-2000:], r.stderr[-2000:])
    return r


@pytest.fixture(scope="session")
def binaries():
    subprocess.check_call(["make", "-C", ROOT, "-j4", "all", "header_only"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/include/gunrock"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "dropin"), "-j3"],
                              stdout=subprocess.DEVNULL)
    return BIN


def write_mtx(path, ro, ci, w=None, symmetric=False):
    src = np.repeat(np.arange(len(ro) - 1), np.diff(ro))
    field = "pattern" if w is None else "real"
    with open(path, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate %s general\n%d %d %d\n" % (field, len(ro) - 1, len(ro) - 1, len(ci)))
        if w is None:
            np.savetxt(f, np.stack([src + 1, ci + 1], 1), fmt="%d %d")
        else:
            for s, d, x in zip(src + 1, ci + 1, w):
                f.write("%d %d %.9g\n" % (s, d, x))


def test_drivers_build_and_help(binaries):
    for name in ("bfs", "sssp", "pr", "bfs_generic", "sssp_generic", "pr_generic"):
        assert os.path.exists(os.path.join(BIN, name))
    r = run([os.path.join(BIN, "bfs"), "--help"])
    for flag in ("--market", "--src", "--num_runs", "--validate", "--advance_load_balance",
                 "--filter_algorithm", "--enable_filter", "--enable_uniquify", "--export_metrics",
                 "--json_dir", "--json_file", "--tag"):
        assert flag in r.stdout
    r = run([os.path.join(BIN, "pr"), "--help"])
    assert "--src" not in r.stdout and "--validate" not in r.stdout  # registered per algorithm
    r = run([os.path.join(BIN, "bfs")], check=False)  # no --market => help, exit 0
    assert r.returncode == 0 and "--market" in r.stdout
    r = run([os.path.join(BIN, "bfs"), "--nonsense", "1"], check=False)
    assert r.returncode != 0


def test_host_utilities_batch_and_random(binaries):
    """operators::batch::execute and generate::random need no GPU: every job runs once, an
    exception surfaces after the batch drained, random fills are in range and reproducible."""
    r = run([os.path.join(BIN, "test_host_utils")], check=False)
    assert r.returncode == 0 and "ALL CHECKS PASSED" in r.stdout, r.stdout[-2000:] + r.stderr[-1000:]


@pytest.mark.skipif(not os.path.isdir("/root/reference/include/gunrock"), reason="needs the reference tree")
def test_reference_sources_compile_against_our_headers(binaries):
    for name in ("ref_bfs", "ref_sssp", "ref_pr", "refalg_bfs", "refalg_sssp", "refalg_pr",
                 "refalg_kcore", "refalg_hits", "refalg_ppr", "refalg_spmv", "refalg_bc", "refalg_color"):
        assert os.path.exists(os.path.join(DROPIN, name)), name


@pytest.mark.gpu
def test_readme_commands_on_chesapeake(binaries):
    # reference README.md:97-111: bfs --market chesapeake.mtx --src 0 ; tuned: merge_path + compact filter
    r = run([os.path.join(BIN, "bfs"), "--market", CHES, "--src", "0", "--validate"])
    assert "Source : 0" in r.stdout
    assert "GPU distances[:39] = " + CHES_HEAD in r.stdout.replace(" \n", "\n")
    assert "CPU Distances[:39] = " + CHES_HEAD in r.stdout.replace(" \n", "\n")
    assert re.search(r"GPU Elapsed Time : [0-9.e+-]+ \(ms\)", r.stdout)
    assert "Number of errors : 0" in r.stdout
    r = run([os.path.join(BIN, "bfs"), "-m", CHES, "-s", "0", "--advance_load_balance", "merge_path",
             "--enable_filter", "--filter_algorithm", "compact", "--validate"])
    assert "Number of errors : 0" in r.stdout
    r = run([os.path.join(BIN, "sssp"), "--market=" + CHES, "--src=5", "--validate"])
    assert "Number of errors : 0" in r.stdout
    r = run([os.path.join(BIN, "pr"), "--market", CHES, "-n", "2"])
    assert "GPU rank[:39] = " in r.stdout and "GPU Elapsed Time" in r.stdout


@pytest.mark.gpu
def test_every_operator_combination_validates(binaries, gr, tmp_path):
    _, c = gr.generate("rmat", 20000, 300000, seed=21)
    rng = np.random.default_rng(2)
    w = (rng.integers(1, 64, c.number_of_nonzeros) / 8.0).astype(np.float32)
    unweighted = str(tmp_path / "g.mtx")
    weighted = str(tmp_path / "gw.mtx")
    write_mtx(unweighted, c.row_offsets, c.column_indices)
    write_mtx(weighted, c.row_offsets, c.column_indices, w)
    src = str(int(np.argmax(np.diff(c.row_offsets))))
    lbs = ["thread_mapped", "warp_mapped", "block_mapped", "merge_path", "merge_path_v2", "bucketing"]
    filters = [[], ["--enable_filter"], ["--enable_filter", "--filter_algorithm", "compact"],
               ["--enable_filter", "--filter_algorithm", "remove"], ["--enable_filter", "--filter_algorithm", "bypass"]]
    r = run([os.path.join(BIN, "bfs"), "--market", unweighted, "--src", src, "--validate",
             "--advance_direction", "optimized"])
    assert "Number of errors : 0" in r.stdout
    for exe, extra in (("bfs", []), ("bfs", ["--generic_operators"]), ("bfs_generic", [])):
        for lb in lbs:
            for flt in filters:
                r = run([os.path.join(BIN, exe), "--market", unweighted, "--src", src, "--validate",
                         "--advance_load_balance", lb] + flt + extra)
                assert "Number of errors : 0" in r.stdout, (exe, lb, flt, r.stdout[-500:])
    for exe, extra in (("sssp", []), ("sssp", ["--generic_operators"]), ("sssp_generic", []),
                       ("sssp_generic", ["--enable_uniquify"])):
        for lb in lbs:
            r = run([os.path.join(BIN, exe), "--market", weighted, "--src", src, "--validate",
                     "--advance_load_balance", lb] + extra)
            assert "Number of errors : 0" in r.stdout, (exe, lb, r.stdout[-500:])
    # work_stealing is unsupported, as upstream: exception, non-zero exit
    r = run([os.path.join(BIN, "bfs_generic"), "--market", unweighted, "--src", src,
             "--advance_load_balance", "work_stealing"], check=False)
    assert r.returncode != 0
    # PR: engine (pull) vs generic (push on the operators) agree
    def ranks(out):
        line = [l for l in out.splitlines() if l.startswith("GPU rank[:")][0]
        return np.array(line.split("=")[1].split(), dtype=np.float64)
    a = ranks(run([os.path.join(BIN, "pr"), "--market", weighted]).stdout)
    b = ranks(run([os.path.join(BIN, "pr_generic"), "--market", weighted]).stdout)
    assert np.abs(a - b).max() < 1e-6


@pytest.mark.gpu
def test_operator_level_semantics(binaries):
    """tests/cpp/test_operators.cu: frontier API, advance slot semantics for every load
    balance, filter stability, uniquify, parallel_for, bucketing's histogram choice."""
    r = run([os.path.join(BIN, "test_operators")], check=False)
    assert "ALL CHECKS PASSED" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("CHECK ") >= 25


@pytest.mark.gpu
def test_engine_handle_cache_sees_in_place_edits(binaries):
    """tests/cpp/test_engine_cache.cu: the C++ bridge's cached graph handles under in-place edits of the (non-owning) CSR
    arrays -- full validation by default, identity + engine::invalidate on request."""
    r = run([os.path.join(BIN, "test_engine_cache")], check=False)
    assert "ALL CHECKS PASSED" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("CHECK ") >= 8


@pytest.mark.gpu
def test_export_metrics_json(binaries, tmp_path):
    import json
    run([os.path.join(BIN, "bfs"), "--market", CHES, "--src", "0,5", "--export_metrics", "--json_dir", str(tmp_path),
         "--json_file", "m.json", "--tag", "a,b"])
    j = json.load(open(tmp_path / "m.json"))
    assert j["primitive"] == "bfs" and j["srcs"] == [0, 5] and j["tags"] == ["a", "b"]
    assert j["num-vertices"] == 39 and j["num-edges"] == 340
    assert j["edges-visited"] == [340, 340] and len(j["mteps"]) == 2
    assert j["json-schema"] == "2022-10-28"


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(DROPIN, "refalg_bfs")), reason="drop-in binaries not built")
def test_reference_sources_run_on_our_framework(gr, tmp_path):
    """The reference's driver AND algorithm headers (its own device lambdas),
    compiled unmodified against our headers, validate against its own CPU oracle."""
    _, c = gr.generate("rmat", 5000, 60000, seed=4)
    g = str(tmp_path / "g.mtx")
    write_mtx(g, c.row_offsets, c.column_indices)
    src = str(int(np.argmax(np.diff(c.row_offsets))))
    for exe in ("ref_bfs", "refalg_bfs", "ref_sssp", "refalg_sssp"):
        for mtx, s in ((CHES, "0"), (g, src)):
            for lb in ("thread_mapped", "block_mapped", "merge_path"):
                r = run([os.path.join(DROPIN, exe), "--market", mtx, "--src", s, "--validate",
                         "--advance_load_balance", lb])
                assert "Number of errors : 0" in r.stdout, (exe, mtx, lb, r.stdout[-400:])
    r = run([os.path.join(DROPIN, "refalg_bfs"), "--market", CHES, "--src", "0", "--validate",
             "--advance_load_balance", "merge_path", "--enable_filter", "--filter_algorithm", "compact"])
    assert "Number of errors : 0" in r.stdout  # throws upstream (filter/compact.hxx:21-24); works here
    def ranks(out):
        line = [l for l in out.splitlines() if l.startswith("GPU rank[:")][0]
        return np.array(line.split("=")[1].split(), dtype=np.float64)
    a = ranks(run([os.path.join(DROPIN, "refalg_pr"), "--market", CHES]).stdout)
    b = ranks(run([os.path.join(BIN, "pr"), "--market", CHES]).stdout)
    assert np.abs(a - b).max() < 1e-6


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(DROPIN, "refalg_kcore")), reason="drop-in binaries not built")
def test_other_reference_algorithms_run_on_our_operators(gr, tmp_path):
    """the reference's OWN k-core, PPR, SpMV, HITS, BC and colouring (headers and
    drivers, unmodified) on our enactor / frontier / advance / filter / parallel_for / batch;
    those with a CPU oracle upstream validate against it."""
    _, c = gr.generate("rmat_sym", 3000, 20000, seed=9)
    g = str(tmp_path / "g.mtx")
    write_mtx(g, c.row_offsets, c.column_indices)
    # these four drivers take the file as argv[1] and always validate against their CPU oracle
    for exe in ("refalg_kcore", "refalg_ppr", "refalg_spmv", "refalg_color"):
        for mtx in (CHES, g):
            r = run([os.path.join(DROPIN, exe), mtx], check=False)
            assert r.returncode == 0, (exe, mtx, r.stdout[-600:], r.stderr[-600:])
            assert "Number of errors : 0" in r.stdout, (exe, mtx, r.stdout[-600:])
    for exe in ("refalg_hits", "refalg_bc"):
        r = run([os.path.join(DROPIN, exe), "--market", CHES], check=False)
        assert r.returncode == 0 and "GPU Elapsed Time" in r.stdout, (exe, r.stdout[-600:], r.stderr[-600:])
