#!/usr/bin/env python
"""bench.py -- MTEPS of the BFS / SSSP / PageRank hot path on the BASELINE.json workloads.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--only bfs,bfs_forward,sssp,pr] [--no-cpu-baseline]

ONE JSON line.  Top level = BASELINE.json configs[1]: BFS on the soc-LiveJournal1 stand-in (C2'),
direction-optimising; a "step" is one full pass of the hot path -- problem reset + the enact loop
(frontier seed -> convergence) over the graph already resident in HBM; value = traversed edges /
wall time of K steps (synchronize on both sides, max over ranks), in MTEPS
(= edges_visited / (elapsed_ms * 1000), include/gunrock/util/performance.hxx:225-229 of the reference).

At N = 1 the same line carries the other single-GPU configurations of the metric ("BFS + SSSP", and
PageRank), each measured in this run with its own roofline and CPU baselines:
  bfs_forward  configs[1] as written: merge-path advance + compact filter, advance_direction = forward
  sssp         configs[2]: road_usa stand-in (C3'), unit weights (what the reference loader makes of the
               pattern file) and the U{1..1000} weighted variant (near-far / delta-stepping schedule)
  pr           configs[3]: kron_g500-logn21 stand-in (C4') at full size, alpha 0.85, tol 1e-6

roofline objects: algorithmic bytes (SURVEY.md 8d / BASELINE.md) of the launches of the dominant kernel
divided by their HIP-event time on the engine's stream (GRX_FLAG_PROFILE), against 8 TB/s HBM.
`traffic` = FETCH_SIZE + WRITE_SIZE per launch from the committed rocprofv3 --pmc passes of this
command, attached only while the engine sources still hash to what those passes profiled
(profiles/history/r2_bench_pmc.json: source_sha), else null.
cpu_baseline = the oracle (port of the reference's CPU path), 1 core, bounded sample;
cpu_baseline_ncore = the reference's operator loop on all host cores (oracle/oracle_omp.c).
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[1]: soc-LiveJournal1 stand-in (SURVEY.md 8d C2')
    "lj": dict(file="soc-LiveJournal1", kind="rmat", V=4_847_571, entries=68_993_773, a=0.57, b=0.19, c=0.19,
               name="soc-LiveJournal1 stand-in C2': R-MAT(0.57,0.19,0.19,0.05) 4,847,571 V / 68,993,773 E, "
                    "src = max out-degree vertex"),
    # configs[3]: kron_g500-logn21 stand-in C4'
    "kron": dict(file="kron_g500-logn21", kind="rmat_sym", V=1 << 21, entries=91_042_010, a=0.57, b=0.19, c=0.19,
                 name="kron_g500-logn21 stand-in C4': symmetric R-MAT(0.57,0.19,0.19,0.05) 2^21 V / 91,042,010 "
                      "entries (~182 M edges)"),
    # configs[2]: road_usa stand-in C3'
    "road": dict(file="road_usa", kind="road", V=4894 * 4894, entries=0, a=0.602, b=0.0, c=0.0,
                 name="road_usa stand-in C3': 4894x4894 lattice, edges kept with p=0.602 (~57.7 M directed edges), "
                      "src = centre vertex"),
    # configs[4]: soc-twitter-2010 stand-in C5' (N = 8)
    "twitter": dict(file="soc-twitter-2010", kind="rmat_sym", V=21_297_772, entries=265_025_809, a=0.57, b=0.19, c=0.19,
                    name="soc-twitter-2010 stand-in C5': symmetric R-MAT(0.57,0.19,0.19,0.05) 21,297,772 V / "
                         "265,025,809 entries (~530 M edges)"),
    # round 5: the same V / E as C2' with the DEPTH of the published graph -- R-MAT core + a long-tailed
    # periphery (gunrock_amd/csrc/grx_host.cpp, kind 3): 14 levels from the hub where C2' has 7
    "deep": dict(kind="rmat_deep", V=4_847_571, entries=68_993_773, a=0.57, b=0.19, c=0.19,
                 name="deep scale-free stand-in: R-MAT(0.57,0.19,0.19,0.05) core on 9/10 of 4,847,571 V + a periphery whose "
                      "population falls by 0.3 per hop (68,993,773 E; 14 levels from the max out-degree vertex)"),
    "small": dict(kind="rmat", V=1 << 18, entries=4_000_000, a=0.57, b=0.19, c=0.19,
                  name="R-MAT 262,144 V / 4,000,000 E (smoke size)"),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def find_real(key, data_dir):
    """Path of the published graph of workload `key` under data_dir ("if the real .mtx files are supplied
    on the measurement box, use them"): <dir>/<name>.mtx or <dir>/<name>/<name>.mtx as the SuiteSparse archives
    unpack (datasets/*/Makefile of the reference), else None."""
    name = WORKLOADS[key].get("file")
    if not data_dir or not name:
        return None
    for cand in (os.path.join(data_dir, name + ".mtx"), os.path.join(data_dir, name, name + ".mtx")):
        if os.path.isfile(cand):
            return cand
    return None


def load_workload(gr, key, data_dir=None, weighted=False, seed=42):
    """-> (properties, csr, source, info).  The published graph through the engine's Matrix-Market loader
    (grx_host_csr_load_mtx: byte-equal to io/matrix_market.hxx:99-254 + formats/csr.hxx:81-140) when its file is
    present under data_dir, else the seeded stand-in of SURVEY 8d.  Source: the max out-degree vertex (the road
    stand-in: its centre vertex).  weighted: U{1..1000} integer weights -- generated with the stand-in, drawn (seeded)
    onto the real topology when the published file is a pattern matrix."""
    wl = WORKLOADS[key]
    path = find_real(key, data_dir)
    if path:
        props, coo = gr.matrix_market_t().load(path)
        csr = gr.csr_t().from_coo(coo)
        info = {"data": "real", "file": path, "name": "%s (%s)" % (wl["file"], os.path.basename(path))}
        if weighted and not props.weighted:
            csr.nonzero_values = pair_hash_weights(csr, seed)
            csr._device = None
            props.weighted = True
            info["data"] = "real topology, synthetic U{1..1000} weights"
        src = int(np.argmax(np.diff(csr.row_offsets)))
        return props, csr, src, info
    c_par = 1.0 if (weighted and wl["kind"] == "road") else wl["c"]
    props, csr = gr.generate(wl["kind"], wl["V"], wl["entries"], wl["a"], wl["b"], c_par, seed=seed)
    if wl["kind"] == "road":
        side = int(round(wl["V"] ** 0.5))
        src = (side // 2) * side + side // 2
    else:
        src = int(np.argmax(np.diff(csr.row_offsets)))
    return props, csr, src, {"data": "synthetic", "file": None, "name": wl["name"]}


def source_sha():
    """Hash of the engine sources, the C-ABI header and the build flags: ties committed PMC traffic numbers to the
    kernels they were taken from."""
    from gunrock_amd.build import source_sha as f
    return f()


def load_pmc():
    """The newest committed profiles/r*_bench_pmc.json whose source_sha is the current one, else None."""
    import glob
    sha = source_sha()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_pmc.json")), reverse=True):
        try:
            pmc = json.load(open(path))
        except Exception:
            continue
        if pmc.get("source_sha") == sha:
            pmc["_file"] = os.path.relpath(path, ROOT)
            return pmc
    return None


def roof(levels, bytes_of, kernel, note=None):
    ms = sum(l["advance_ms"] for l in levels)
    byt = sum(bytes_of(l) for l in levels)
    ach = byt / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    r = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "kernel": kernel,
         "launches_per_step": len(levels), "alg_bytes_per_step": int(byt),
         "kernel_ms_per_step": round(ms, 4), "avg_launch_us": round(ms * 1e3 / max(1, len(levels)), 2)}
    if note:
        r["alg_bytes_model"] = note
    return r


def attach_traffic(r, pmc, cls, per_step=False):
    """per_step: the PMC class is summed over one whole search (many levels per launch make the launch count
    meaningless there); it is divided by this run's launch count to stay comparable with `achieved`."""
    k = (pmc or {}).get("classes", {}).get(cls)
    if not k or "fetch_bytes_per_launch" not in k:
        r["traffic_note"] = ("no PMC passes committed for the current engine sources (no profiles/r*_bench_pmc.json "
                             "carries this source_sha)")
        return
    tr = k["fetch_bytes_per_launch"] + k.get("write_bytes_per_launch", 0.0)
    if per_step:
        r["traffic_per_step"] = int(tr)
        tr /= max(1, r["launches_per_step"])
    r["traffic"] = int(tr)
    r["traffic_committed"] = True  # from the committed counter passes of THESE sources, not from this run (see traffic_source)
    if k.get("SQ_LDS_IDX_ACTIVE_per_launch"):
        # LDS bank conflicts of the class's kernels (north_star: "LDS bank conflicts on filter" -- the filter is fused into these)
        r["lds_bank_conflict_ratio"] = round(k.get("SQ_LDS_BANK_CONFLICT_per_launch", 0.0) / k["SQ_LDS_IDX_ACTIVE_per_launch"], 3)
    r["traffic_source"] = ("%s class '%s': FETCH_SIZE + WRITE_SIZE per launch, separate --pmc passes of this command on "
                           "these sources, NOT re-measured in this run (raw counters; gfx950 tallies wide coalesced "
                           "reads at 1/2)" % (pmc.get("_file", "profiles/"), cls))
    if "duration_us_per_launch" in k:
        r["rocprof_avg_launch_us"] = round(k["duration_us_per_launch"], 2)


def best_profile(run, profile_of, tries=3):
    best = None
    for _ in range(tries):
        run()
        prof = profile_of()
        t = sum(l["advance_ms"] for l in prof)
        if best is None or t < best[0]:
            best = (t, prof)
    return best[1]


def timed(run, sync, steps, warmup):
    """warm-up steps, then EXACTLY `steps` steps between two synchronisations.  The collector is held off for the timed
    region: a generation-2 pass over the graph generator's arrays takes tens of milliseconds, and a step takes 0.2."""
    import gc
    for _ in range(warmup):
        run()
    sync()
    was = gc.isenabled()
    gc.disable()
    try:
        t1 = time.perf_counter()
        for _ in range(steps):
            run()
        sync()
        return (time.perf_counter() - t1) * 1e3 / max(1, steps)
    finally:
        if was:
            gc.enable()


def pair_hash_weights(csr, seed=42):
    """U{1..1000} integer weights, one per UNORDERED vertex pair (w(u, v) == w(v, u) on a symmetric graph): a hash of
    (min, max, seed).  Drawn onto pattern topologies (the published LJ / kron / road / twitter files are all pattern
    matrices; the reference loader gives them 1.0 everywhere, io/matrix_market.hxx:170-171) for the weighted SSSP sections."""
    u = np.repeat(np.arange(csr.number_of_rows, dtype=np.uint64), np.diff(csr.row_offsets).astype(np.int64))
    v = csr.column_indices.astype(np.uint64)
    h = np.minimum(u, v) * np.uint64(2654435761) + np.maximum(u, v) * np.uint64(2246822519) + np.uint64(seed)
    h ^= h >> np.uint64(13)
    h *= np.uint64(0x9E3779B1)
    h ^= h >> np.uint64(17)
    return (1 + (h % np.uint64(1000))).astype(np.float32)


def slim(r, keys=("frac", "achieved", "peak", "unit", "bound", "traffic", "traffic_committed", "lds_bank_conflict_ratio",
                  "avg_launch_us", "launches_per_step",
                  "alg_bytes_per_step", "kernel_ms_per_step", "fat_levels_frac", "rocprof_avg_launch_us")):
    return None if r is None else {k: r[k] for k in keys if k in r}


def cpu_slim(c):
    return None if c is None else {k: c[k] for k in ("value", "unit", "cores", "kind", "matches_gpu") if k in c}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="lj", choices=sorted(WORKLOADS),
                    help="graph of the top-level BFS line (default: BASELINE configs[1])")
    ap.add_argument("--only", default="bfs,bfs_do,multi,bfs_deep,sssp,sssp_lj,sssp_kron,pr,pr_lj,c5",
                    help="comma list of the sections to run at N = 1 (bfs = the top-level forward search, always run)")
    ap.add_argument("--data-dir", default=os.environ.get("GRX_DATA_DIR", ""),
                    help="directory holding the published graphs (soc-LiveJournal1.mtx, road_usa.mtx, "
                         "kron_g500-logn21.mtx, soc-twitter-2010.mtx, flat or one folder each): every section whose "
                         "file is present runs on it (data: real) instead of its seeded stand-in")
    ap.add_argument("--lb", default="merge_path")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--detail", default=os.environ.get("GRX_BENCH_DETAIL", ""),
                    help="file that receives the FULL objects of every section (per-level arrays, kernel names, samples); "
                         "default gpurun_out/bench_detail.json when that directory exists.  stdout carries ONE compact line")
    ap.add_argument("--topdown-only", action="store_true", help="(kept for old command lines: the top level IS forward now)")
    args = ap.parse_args()
    only = set(x for x in args.only.split(",") if x)
    if "bfs_forward" in only:  # round-3 name of what is now the top level
        only.add("bfs_do")

    import torch
    import gunrock_amd as gr

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        return bench_multi_gpu(args, gr, torch, rank, local_rank, world)

    dev = "cuda:%d" % local_rank
    torch.cuda.set_device(local_rank)
    ctx = gr.multi_context_t(local_rank)
    pmc = load_pmc()
    cpu_on = not args.no_cpu_baseline
    O = None
    ncore = 0
    if cpu_on:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O  # checker / timed CPU baseline only
        ncore = O.omp_threads()

    def sync():
        torch.cuda.synchronize()
        ctx.synchronize()

    env = dict(gr=gr, torch=torch, ctx=ctx, dev=dev, sync=sync, pmc=pmc, cpu_on=cpu_on, O=O, ncore=ncore, args=args)
    lb = getattr(gr, args.lb)
    INF = np.iinfo(np.int32).max

    # ------------------------------------------------------------------ BFS on the configs[1] graph
    t0 = time.time()
    props, csr, src, info = load_workload(gr, args.workload, args.data_dir)
    G = gr.build_graph(props, csr, ctx, device=dev)
    V, E = G.get_number_of_vertices(), G.get_number_of_edges()
    dist_t = torch.empty(V, dtype=torch.int32, device=dev)
    t_setup = time.time() - t0

    def bfs_opts(direction, flags=0):
        return gr.options_t(advance_load_balance=lb, enable_filter=True, filter_algorithm=gr.compact,
                            advance_direction=direction, engine_flags=flags)

    # ONE-SHOT cost: the first search on a fresh graph handle builds what the handle caches for its direction (forward: bin
    # table + E-entry bin array; direction-optimising: symmetry check or transpose, no-in-edges bitmap, two-neighbour
    # array) and then searches.  Untimed for `value` (like the CSR build), reported per direction.
    first_call_ms = {}
    do_dirs = (gr.forward, gr.optimized) if "bfs_do" in only else (gr.forward,)
    for direction0 in do_dirs:
        G0 = gr.build_graph(props, csr, ctx, device=dev) if direction0 != gr.forward else G
        sync()
        t1 = time.perf_counter()
        gr.bfs(G0, src, dist_t, None, ctx, bfs_opts(direction0))
        sync()
        first_call_ms[direction0] = (time.perf_counter() - t1) * 1e3
        if G0 is not G:
            gr.bfs(G, src, dist_t, None, ctx, bfs_opts(direction0))  # the handle the timed steps use
            del G0

    def bfs_section(direction):
        o = bfs_opts(direction, gr.FLAG_ASYNC_RETURN)
        ms_step = timed(lambda: gr.bfs(G, src, dist_t, None, ctx, o), sync, args.steps, args.warmup)
        st = gr.run_stats(ctx)
        rep = [round(timed(lambda: gr.bfs(G, src, dist_t, None, ctx, o), sync, args.steps, 0), 4) for _ in range(2)]
        return ms_step, st, rep

    td_bytes = lambda l: 12 * l["frontier_size"] + 12 * l["edges"]
    # bottom-up kernel: visited r/w + next-frontier bitmap (3 V/8), two in-offsets per open vertex (8), column
    # index + frontier-bitmap word per probed in-edge (8), label + two out-offsets per discovered vertex (12)
    bu_bytes = lambda l, nxt: 3 * (V // 8) + 8 * l["bu_open"] + 8 * l["bu_probes"] + 12 * nxt

    def bfs_profile(direction):
        po = bfs_opts(direction, gr.FLAG_PROFILE)
        return best_profile(lambda: gr.bfs(G, src, dist_t, None, ctx, po), lambda: gr.level_profile(ctx))

    # --- top level: configs[1] AS WRITTEN -- forward merge-path advance + compact filter
    ms_per_step, st, rep_ms = bfs_section(gr.forward)
    edges_rank = st["edges_visited"]
    mteps = edges_rank / (ms_per_step * 1e3)
    bfs_gpu_depths = dist_t.cpu().numpy().copy()
    prof = bfs_profile(gr.forward)
    roofline = roof(prof, td_bytes, "forward BFS level kernels: advance_block on thin levels; bfs_scatter2_kernel + "
                    "bfs_sweep2_kernel (grx_bin.hpp) on fat ones", "12 B per frontier slot + 12 B per traversed edge (SURVEY 8d)")
    attach_traffic(roofline, pmc, "topdown_fat")
    fat = sorted(prof, key=lambda l: -l["edges"])[:2]
    roofline["fat_levels_avg_launch_us"] = round(sum(l["advance_ms"] for l in fat) * 1e3 / max(1, len(fat)), 2)
    roofline["fat_levels_frac"] = round(sum(td_bytes(l) for l in fat) / max(1e-9, sum(l["advance_ms"] for l in fat) * 1e-3)
                                        / 1e9 / HBM_PEAK_GBS, 4)
    roofline["levels"] = [[l["frontier_size"], l["edges"], int(l["bottom_up"]), round(l["advance_ms"], 4),
                           round(l["other_ms"], 4)] for l in prof]
    roofline["levels_columns"] = "frontier vertices, out-edges, mode (2 = binned), level kernel(s) ms, head kernel ms"

    # --- configs[1] on the LITERAL kernel its text names: every level on the load-balanced claim-per-edge advance (advance_block,
    # the engine's merge-path body) + fused compact filter, no binned levels, no many-levels body (GRX_FLAG_LB_STRICT)
    strict_o = bfs_opts(gr.forward, gr.FLAG_ASYNC_RETURN | gr.FLAG_LB_STRICT)
    strict_ms = timed(lambda: gr.bfs(G, src, dist_t, None, ctx, strict_o), sync, args.steps, 2)
    strict_same = bool(np.array_equal(dist_t.cpu().numpy(), bfs_gpu_depths))
    gr.bfs(G, src, dist_t, None, ctx, bfs_opts(gr.forward))  # (the handle's hints back on the default schedule)

    # --- direction-optimising search on the same graph (engine extension)
    do = None
    if "bfs_do" in only:
        ms_d, st_d, rep_d = bfs_section(gr.optimized)
        same = bool(np.array_equal(dist_t.cpu().numpy(), bfs_gpu_depths))
        prof_do = bfs_profile(gr.optimized)
        sizes = [l["frontier_size"] for l in prof_do] + [0]
        bu = [dict(l, nxt=sizes[i + 1]) for i, l in enumerate(prof_do) if l["bottom_up"] == 1]
        td = [l for l in prof_do if l["bottom_up"] != 1]
        r_bu = roof(bu, lambda l: bu_bytes(l, l["nxt"]), "bfs_level_kernel (bottom-up launches)",
                    "3 V/8 + 8 open + 8 probes + 12 found (DESIGN.md 5)")
        t_bu, t_td = sum(l["advance_ms"] for l in bu), sum(l["advance_ms"] for l in td)
        r_bu["share_of_step_kernel_time"] = round(t_bu / max(t_bu + t_td, 1e-9), 3)
        if bu:
            attach_traffic(r_bu, pmc, "bottom_up")
        r_bu["all_levels"] = [[l["frontier_size"], l["edges"], int(l["bottom_up"]), l["bu_open"], l["bu_probes"],
                               round(l["advance_ms"], 4), round(l["other_ms"], 4)] for l in prof_do]
        # what a direction-optimising search really reads: out-edges of the top-down levels + in-edges probed bottom-up
        scanned = int(sum(l["edges"] for l in td) + sum(l["bu_probes"] for l in bu))
        do = {"config": "same graph / source, advance_direction = optimized (Beamer switch decided per level on the device)",
              "ms_per_step": round(ms_d, 4), "mteps": round(st_d["edges_visited"] / (ms_d * 1e3), 1),
              "mteps_convention": "Graph500 / reference edges_visited: sum of out-degrees of the reached vertices",
              "edges_actually_scanned": scanned, "mteps_edges_actually_scanned": round(scanned / (ms_d * 1e3), 1),
              "ms_per_step_repeated": rep_d, "steps": args.steps, "search_depth": st_d["search_depth"],
              "enact_ms_last": round(st_d["elapsed_ms"], 4), "equal_to_forward_depths": same,
              "first_call_ms": round(first_call_ms[gr.optimized], 3), "one_shot_ms": round(first_call_ms[gr.optimized], 3),
              "roofline": r_bu}

    # --- other sources: 16 vertices of the giant component, each searched ONCE, on a handle whose launch-group hint was
    # trained by a different source (reference: io/parameters.hxx:192-223 -- several / random sources)
    multi = None
    if "multi" in only:
        multi = bench_multi_source(env, G, csr, src, dist_t, bfs_gpu_depths, bfs_opts, do is not None)

    # --- SSSP and PageRank on the same graph (north_star: BFS / SSSP / PR on soc-LiveJournal1 and kron_g500-logn21)
    tag = args.workload
    extra = {}
    if tag != "kron" and ("sssp_" + tag) in only:
        extra["sssp_" + tag] = bench_sssp_on(env, tag, props, csr, src, info, G_unit=G)
    if tag != "kron" and ("pr_" + tag) in only:
        extra["pr_" + tag] = bench_pr_on(env, tag, props, csr, info)

    cpu = cpu_n = None
    if cpu_on:
        g = O.Csr(csr.row_offsets, csr.column_indices, csr.nonzero_values)
        t_cpu, runs = 0.0, 0
        while t_cpu < 8e3 and runs < 64:
            d_cpu, ms = O.bfs(g, src)
            t_cpu += ms
            runs += 1
        cpu = {"value": round(runs * edges_rank / (t_cpu * 1e3), 2), "unit": "MTEPS", "cores": 1, "kind": "port",
               "sample": "%d full BFS runs of the same workload/source with oracle/oracle.c orc_bfs (priority-queue "
                         "search of examples/algorithms/bfs/bfs_cpu.hxx), %.1f s" % (runs, t_cpu / 1e3),
               "matches_gpu": bool(np.array_equal(d_cpu, bfs_gpu_depths))}
        t_n, runs_n, ev_n = 0.0, 0, 0
        while t_n < 3e3 and runs_n < 64:
            d_n, ms, ev = O.bfs_omp(g, src)
            t_n += ms
            runs_n += 1
            ev_n += ev
        cpu_n = {"value": round(ev_n / (t_n * 1e3), 2), "unit": "MTEPS", "cores": ncore, "kind": "port",
                 "sample": "%d full level-synchronous BFS runs on %d host threads (oracle/oracle_omp.c orc_bfs_omp: the "
                           "reference's advance with atomicMin, bfs.hxx:105-146, per level), %.1f s"
                           % (runs_n, ncore, t_n / 1e3),
                 "matches_gpu": bool(np.array_equal(d_n, bfs_gpu_depths))}
        del g
    del G, dist_t

    detail = {"metric": "MTEPS (million traversed edges/sec) BFS", "value": round(mteps, 1), "unit": "MTEPS",
              "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": info["data"],
              "config": {"workload": "BFS on " + info["name"] + " -- BASELINE.json configs[1] as written: merge-path "
                                     "advance + compact filter, advance_direction = forward",
                         "data_file": info["file"], "n_vertices": V, "n_edges": E, "source": src,
                         "advance_load_balance": args.lb, "filter": "compact (fused into the advance)",
                         "strict_merge_path_ms": round(strict_ms, 4), "strict_merge_path_mteps": round(edges_rank / (strict_ms * 1e3), 1),
                         "strict_merge_path_note": "same search, GRX_FLAG_LB_STRICT: every level on advance_block (merge-path balance, "
                                                   "claim per edge); equal depths: %s" % strict_same,
                         "advance_direction": "forward", "completion": "GRX_FLAG_ASYNC_RETURN, K steps bracketed by syncs",
                         "parallelism": "single GPU (the single-GPU engine; the partitioned path is used for N > 1 only)",
                         "edges_visited_per_step": edges_rank, "search_depth": st["search_depth"],
                         "enact_ms_last": round(st["elapsed_ms"], 4), "ms_per_step_repeated": rep_ms,
                         "launch_groups_last": int(st.get("aux", 0)),
                         "schedule": "launch groups and binned-level kernels follow the previous search on the handle (one "
                                     "repeated source, like the reference's driver); multi_source: other sources + cold",
                         "first_call_ms": round(first_call_ms[gr.forward], 3),
                         "one_shot_ms": round(first_call_ms[gr.forward], 3),
                         "one_shot_note": "fresh graph handle: per-graph preprocessing + one search",
                         "setup_s": round(t_setup, 1), "engine_source_sha": source_sha()},
              "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_ncore": cpu_n,
              "bfs_direction_optimized": do, "multi_source": multi}
    detail.update(extra)
    del csr

    # ------------------------------------------------------------------ the other graphs
    if "sssp" in only:
        detail["sssp_road"] = bench_sssp_on(env, "road", None, None, None, None)
    if "pr" in only or "sssp_kron" in only:
        props_k, csr_k, src_k, info_k = load_workload(gr, "kron", args.data_dir)
        if "pr" in only:
            detail["pr_kron"] = bench_pr_on(env, "kron", props_k, csr_k, info_k)
        if "sssp_kron" in only:
            detail["sssp_kron"] = bench_sssp_on(env, "kron", props_k, csr_k, src_k, info_k)
        del csr_k
    if "c5" in only:
        detail["c5_single_gpu"] = bench_c5(gr, torch, ctx, dev, sync, cpu_on, args)
    if "bfs_deep" in only:
        detail["bfs_deep"] = bench_bfs_deep(env)

    # ------------------------------------------------------------------ output: the full objects to a file, ONE compact line to stdout
    path = args.detail or (os.path.join(ROOT, "gpurun_out", "bench_detail.json")
                           if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else "")
    if path:
        try:
            with open(path, "w") as f:
                json.dump(detail, f)
        except OSError:
            path = ""
    line = {k: detail[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                   "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfg = dict(detail["config"])
    sec = {}
    if do:
        sec["bfs_do"] = {"ms": do["ms_per_step"], "mteps": do["mteps"], "mteps_scanned": do["mteps_edges_actually_scanned"],
                         "frac": do["roofline"]["frac"], "first_call_ms": do["first_call_ms"], "eq_fwd": do["equal_to_forward_depths"]}
    if multi:
        sec["multi_source"] = {k: multi[k] for k in ("n_sources", "forward_mteps", "forward_vs_single_source",
                                                      "forward_cold_mteps", "do_mteps", "violations") if k in multi}
    for name, it in detail.items():
        if name.startswith("sssp_") and it:
            for lab in ("unit_weights", "weighted_1_1000"):
                x = it.get(lab)
                if x:
                    sec["%s_%s" % (name, "unit" if lab == "unit_weights" else "w")] = {
                        "ms": x["ms_per_step"], "mteps": x["mteps"], "frac": x["roofline"]["frac"],
                        "cpu1_mteps": (x["cpu_baseline"] or {}).get("value"), "eq_cpu": (x["cpu_baseline"] or {}).get("matches_gpu"),
                        "viol": x.get("property_check_violations")}
                    if x.get("block_async"):
                        sec["%s_%s" % (name, "unit" if lab == "unit_weights" else "w")]["supersteps"] = x["block_async"]["supersteps"]
        if name.startswith("pr_") and it:
            sec[name] = {"ms": it["ms_per_step"], "iters": it["iterations"], "ms_per_iter": it["ms_per_iteration"],
                         "mteps": it["mteps"], "frac": it["roofline"]["frac"], "first_call_ms": it["first_call_ms"],
                         "d_f64": it.get("max_abs_diff_to_float64_same_iterations"),
                         "cpu1_mteps": (it["cpu_baseline"] or {}).get("value")}
    c5 = detail.get("c5_single_gpu")
    if c5:
        sec["c5_1gpu"] = {"fwd_ms": c5["forward"]["ms_per_step"], "fwd_mteps": c5["forward"]["mteps"],
                          "fwd_frac": (c5["forward"]["roofline"] or {}).get("frac"),
                          "do_ms": c5["direction_optimized"]["ms_per_step"], "do_mteps": c5["direction_optimized"]["mteps"],
                          "viol": c5.get("property_check_violations")}
    bd = detail.get("bfs_deep")
    if bd:
        sec["bfs_deep"] = {"levels": bd["forward"]["search_depth"], "fwd_ms": bd["forward"]["ms_per_step"],
                           "fwd_mteps": bd["forward"]["mteps"], "fwd_frac": bd["forward"]["roofline"]["frac"],
                           "do_ms": bd["direction_optimized"]["ms_per_step"], "do_mteps": bd["direction_optimized"]["mteps"],
                           "us_per_thin_level": bd["forward"]["thin_levels_us_per_level"],
                           "eq_cpu": (bd.get("cpu_baseline") or {}).get("matches_gpu")}
    cfg["sections"] = sec
    cfg["sections_note"] = ("ms = ms per step (search / run), frac = roofline fraction of the section's dominant kernels, "
                            "cpu1 = 1-core oracle on the same workload, eq_cpu = GPU result == oracle's, viol = oracle "
                            "fixed-point violations; full objects: " + (os.path.relpath(path, ROOT) if path else "--detail FILE"))
    line["config"] = cfg
    if multi:
        line["cold_unknown_source_mteps"] = multi.get("forward_cold_mteps")  # 16 other sources, no launch history used at all
    r = slim(roofline)
    r["kernel"] = "forward BFS level kernels (advance_block; bfs_scatter2 + bfs_sweep2 on fat levels)"
    r["traffic_class"] = "topdown_fat (per fat level)" if roofline.get("traffic") else None
    line["roofline"] = r
    line["cpu_baseline"] = None if cpu is None else dict(cpu_slim(cpu), sample=cpu["sample"][:120])
    line["cpu_baseline_ncore"] = cpu_slim(cpu_n)
    print(json.dumps(line))


def bench_multi_source(env, G, csr, src, dist_t, depths, bfs_opts, with_do):
    """Sources other than the one the handle's launch-group hint was trained on: 16 seeded vertices of
    the main source's component with out-edges, each searched ONCE in random order, every search timed on its own between
    two synchronisations; harmonic-mean rate = sum of edges / sum of times.  `forward_cold`: the same with GRX_BIN_HINT=0
    (every launch group carries the scatter / sweep kernels -- no dependence on history at all)."""
    gr, ctx, sync, O = env["gr"], env["ctx"], env["sync"], env["O"]
    INF = np.iinfo(np.int32).max
    deg = np.diff(csr.row_offsets)
    cand = np.nonzero((depths != INF) & (deg > 0))[0]
    rng = np.random.default_rng(7)
    srcs = [int(x) for x in rng.choice(cand, size=min(16, len(cand)), replace=False)]
    res = {"n_sources": len(srcs), "sources": srcs,
           "protocol": "hint trained on the bench source, then each source once, random order, per-search wall time"}

    def sweep(direction, cold=False):
        if cold:
            os.environ["GRX_BIN_HINT"] = "0"
        try:
            o = bfs_opts(direction, gr.FLAG_ASYNC_RETURN)
            gr.bfs(G, src, dist_t, None, ctx, o)  # (re)train on the bench source
            sync()
            tot_e, tot_t, per = 0, 0.0, []
            for s in srcs:
                t1 = time.perf_counter()
                gr.bfs(G, s, dist_t, None, ctx, o)
                sync()
                dt = time.perf_counter() - t1
                e = gr.run_stats(ctx)["edges_visited"]
                tot_e += e
                tot_t += dt
                per.append(round(e / (dt * 1e9), 1))
            return tot_e / (tot_t * 1e6), per
        finally:
            if cold:
                os.environ.pop("GRX_BIN_HINT", None)

    single = timed(lambda: gr.bfs(G, src, dist_t, None, ctx, bfs_opts(gr.forward, gr.FLAG_ASYNC_RETURN)), sync, 5, 1)
    e_single = gr.run_stats(ctx)["edges_visited"]
    f, per_f = sweep(gr.forward)
    fc, _ = sweep(gr.forward, cold=True)
    res.update({"forward_mteps": round(f, 1), "forward_per_source_gteps": per_f, "forward_cold_mteps": round(fc, 1),
                "forward_single_source_mteps_sync_each": None})
    # the single-source figure measured the same way (one search per sync) for a like-for-like ratio
    t1 = time.perf_counter()
    for _ in range(5):
        gr.bfs(G, src, dist_t, None, ctx, bfs_opts(gr.forward, gr.FLAG_ASYNC_RETURN))
        sync()
    like = e_single * 5 / ((time.perf_counter() - t1) * 1e6)
    res["forward_single_source_mteps_sync_each"] = round(like, 1)
    res["forward_vs_single_source"] = round(f / like, 3)
    if with_do:
        d, per_d = sweep(gr.optimized)
        res.update({"do_mteps": round(d, 1), "do_per_source_gteps": per_d})
    if O is not None:
        g = O.Csr(csr.row_offsets, csr.column_indices, csr.nonzero_values)
        viol = 0
        for s in srcs[:4]:
            gr.bfs(G, s, dist_t, None, ctx, bfs_opts(gr.forward))
            viol += int(O.check_bfs(g, s, dist_t.cpu().numpy()))
        res["violations"] = viol
        res["violations_note"] = "oracle fixed-point check of the first 4 sources (forward)"
    del single
    return res


def bench_sssp_on(env, tag, props, csr, src, info, G_unit=None):
    """SSSP on workload `tag`: unit weights -- what the reference loader makes of the published pattern file
    (io/matrix_market.hxx:170-171) -- and a U{1..1000} weighted variant.  road: generated per variant (the weighted
    lattice comes from the generator); R-MAT graphs: weights drawn per unordered pair onto the topology."""
    gr, torch, ctx, dev, sync, pmc, cpu_on, O, ncore, args = (env[k] for k in (
        "gr", "torch", "ctx", "dev", "sync", "pmc", "cpu_on", "O", "ncore", "args"))
    res = {}
    for label, weighted in (("unit_weights", False), ("weighted_1_1000", True)):
        t0 = time.time()
        if tag == "road":
            props_v, csr_v, src_v, info_v = load_workload(gr, "road", args.data_dir, weighted=weighted)
        else:
            props_v, csr_v, src_v, info_v = props, csr, src, info
            if weighted:
                import copy
                csr_v = copy.copy(csr)
                csr_v.nonzero_values = pair_hash_weights(csr)
                csr_v._device = None
                props_v = copy.copy(props)
                props_v.weighted = True
        res["workload"] = "SSSP on " + info_v["name"]
        G = G_unit if (G_unit is not None and not weighted) else gr.build_graph(props_v, csr_v, ctx, device=dev)
        V, E = G.get_number_of_vertices(), G.get_number_of_edges()
        d = torch.empty(V, dtype=torch.float32, device=dev)
        t_setup = time.time() - t0
        o = gr.options_t(advance_load_balance=gr.merge_path)
        sync()
        t1 = time.perf_counter()
        gr.sssp(G, src_v, d, None, ctx, o)
        sync()
        first_ms = (time.perf_counter() - t1) * 1e3
        road = tag == "road"
        steps = (3 if weighted else 5) if road else 10
        ms_step = timed(lambda: gr.sssp(G, src_v, d, None, ctx, o), sync, steps, 1)
        st = gr.run_stats(ctx)
        bs = gr.block_stats(ctx)  # road-like graphs: block-asynchronous relaxation (grx_block.hip)
        po = gr.options_t(advance_load_balance=gr.merge_path, engine_flags=gr.FLAG_PROFILE)
        prof = best_profile(lambda: gr.sssp(G, src_v, d, None, ctx, po), lambda: gr.level_profile(ctx), tries=1 if road else 2)
        # near-far: records with bottom_up == 2 only pull a bucket out of the far pile; unit weights run on the BFS engine,
        # whose records carry the body (2 = binned there) -- all of them are advance launches
        mean_deg = E / max(1, V)
        near_far = weighted and mean_deg < 6  # (grx_sssp.hip: dense graphs run the plain label-correcting schedule)
        adv = [l for l in prof if not (near_far and l["bottom_up"] == 2)]
        per_edge = 16 if weighted else 12  # weighted: + 4 B weight; all-equal weights are never read
        r = roof(adv, lambda l: 12 * l["frontier_size"] + per_edge * l["edges"],
                 ("SSSP relaxation kernels (near-far schedule, advance_block<sssp_nf_policy>)" if near_far else
                  "SSSP relaxation kernels: fat levels as binned relaxation (sssp_rscatter_kernel + sssp_rsweep_kernel, "
                  "grx_relax.hpp), the others advance_block<sssp_policy>") if weighted else
                 "BFS engine level kernels (all weights equal: depths -> k-fold sums of w)",
                 "12 B per frontier slot + %d B per relaxed edge (SURVEY 8d)" % per_edge)
        if weighted and not near_far:
            fat = [l for l in prof if l["bottom_up"] == 2]
            r["binned_levels"] = len(fat)
            if fat:
                r["binned_levels_frac"] = round(sum(12 * l["frontier_size"] + per_edge * l["edges"] for l in fat) /
                                                (sum(l["advance_ms"] for l in fat) * 1e-3) / 8.0e12, 4)
                r["binned_levels_edges_relaxed"] = int(sum(l["edges"] for l in fat))
        if road:
            attach_traffic(r, pmc, "sssp_" + label, per_step=True)
        elif weighted and tag == "lj":  # (tools/profile_r4.sh ssspd: the same graph, weights and source)
            attach_traffic(r, pmc, "sssp_weighted_dense", per_step=True)
        r["head_kernel_ms_per_step"] = round(sum(l["other_ms"] for l in prof), 3)
        if near_far:
            r["bucket_pull_launches"] = len(prof) - len(adv)
            r["bucket_pull_ms_per_step"] = round(sum(l["advance_ms"] for l in prof if l["bottom_up"] == 2), 3)
        if not road:
            r["levels"] = [[l["frontier_size"], l["edges"], int(l["bottom_up"]), round(l["advance_ms"], 4),
                            round(l["other_ms"], 4)] for l in prof]
        blocky = bs["supersteps"] > 0
        item = {"schedule": (("all weights equal: BFS engine + one pass depths -> distances (grx_sssp.hip)" if not weighted
                              else ("near-far (delta-stepping)" if mean_deg < 6 else "label-correcting levels, the fat ones as binned relaxation (frontier "
                                    "Bellman-Ford, the reference's schedule)")) +
                             ("; road-like graph: block-asynchronous relaxation (grx_block.hip), supersteps between blocks "
                              "of %d vertices inside global label buckets" % bs["block_vertices"] if blocky else "")),
                "data": info_v["data"] if not (weighted and tag != "road") else "synthetic U{1..1000} weights on " + info_v["data"] + " topology",
                "data_file": info_v["file"], "n_vertices": V, "n_edges": E, "source": src_v, "steps": steps,
                "ms_per_step": round(ms_step, 4), "mteps": round(st["edges_visited"] / (ms_step * 1e3), 1),
                "mteps_note": "edges RELAXED (re-relaxations included) / time; mteps_useful_edges divides the out-edges of "
                              "the reached vertices, each counted once",
                "edges_relaxed_per_step": st["edges_visited"], "iterations": st["search_depth"],
                "first_call_ms": round(first_ms, 3), "setup_s": round(t_setup, 1), "roofline": r,
                "cpu_baseline": None, "cpu_baseline_ncore": None}
        if blocky:
            item["block_async"] = dict(bs, us_per_superstep=round(ms_step * 1e3 / max(1, bs["supersteps"]), 2),
                                       relaxations_per_useful_edge=None)
        mine = d.cpu().numpy()
        reached = mine < np.float32(3.0e38)
        useful = int(np.diff(csr_v.row_offsets)[reached].sum())
        if blocky:
            item["block_async"]["relaxations_per_useful_edge"] = round(bs["edges_relaxed"] / max(1, useful), 3)
        item["useful_edges_per_step"] = useful
        item["mteps_useful_edges"] = round(useful / (ms_step * 1e3), 1)
        if cpu_on:
            g = O.Csr(csr_v.row_offsets, csr_v.column_indices, csr_v.nonzero_values)
            item["property_check_violations"] = int(O.check_sssp(g, src_v, mine))
            d1, ms1, ev1, fin1 = O.sssp_budget(g, src_v, 12e3)
            item["cpu_baseline"] = {
                "value": round(ev1 / (ms1 * 1e3), 2), "unit": "MTEPS", "cores": 1, "kind": "port",
                "sample": "oracle/oracle.c orc_sssp (priority-queue Dijkstra of examples/algorithms/sssp/sssp_cpu.hxx) "
                          "from the same source, %s after %.1f s: %d edges scanned"
                          % ("finished" if fin1 else "stopped", ms1 / 1e3, ev1),
                "matches_gpu": bool(np.array_equal(d1, mine)) if fin1 else None}
            dn, msn, evn, itn, finn = O.sssp_omp(g, src_v, budget_ms=10e3 if road else 5e3)
            item["cpu_baseline_ncore"] = {
                "value": round(evn / (msn * 1e3), 2), "unit": "MTEPS", "cores": ncore, "kind": "port",
                "sample": "oracle/oracle_omp.c orc_sssp_omp (the reference's frontier relaxation, sssp.hxx:116-151, on "
                          "%d host threads), %s after %.1f s: %d iterations, %d edges relaxed"
                          % (ncore, "finished" if finn else "stopped", msn / 1e3, itn, evn),
                "matches_gpu": bool(np.array_equal(dn, mine)) if finn else None}
            del g
        res[label] = item
        if G is not G_unit:
            del G
        del d
    return res


def bench_pr_on(env, tag, props, csr, info):
    gr, torch, ctx, dev, sync, pmc, cpu_on, O, ncore, args = (env[k] for k in (
        "gr", "torch", "ctx", "dev", "sync", "pmc", "cpu_on", "O", "ncore", "args"))
    t0 = time.time()
    pattern = bool(np.all(csr.nonzero_values == 1.0))
    G = gr.build_graph(props, csr, ctx, device=dev)  # a FRESH handle: the first call below is the one-shot cost
    V, E = G.get_number_of_vertices(), G.get_number_of_edges()
    p = torch.empty(V, dtype=torch.float32, device=dev)
    result = gr.pr_result_t(p)
    par = gr.pr_param_t(0.85, 1e-6)
    sync()
    t1 = time.perf_counter()
    gr.pr_run(G, par, result, ctx)  # first call: builds the pull layout (transpose, partitions, XCD-blocked copy) and runs
    sync()
    first_ms = (time.perf_counter() - t1) * 1e3
    t_setup = time.time() - t0
    steps = 5
    ms_step = timed(lambda: gr.pr_run(G, par, result, ctx), sync, steps, 1)
    iters = result.iterations
    ppar = gr.pr_param_t(0.85, 1e-6, gr.options_t(engine_flags=gr.FLAG_PROFILE))
    prof = best_profile(lambda: gr.pr_run(G, ppar, result, ctx), lambda: gr.level_profile(ctx))
    # pattern graph: column index + gathered x per edge; offsets, p, x, iweights per vertex; else + the weight stream
    per_iter = 8 * E + 16 * V if pattern else 12 * E + 20 * V
    r = roof(prof, lambda l: per_iter, "pr pull iteration (pr_pull[_xcd]_kernel + long-row pieces + pr_combine_kernel)",
             "8 E + 16 V per iteration on a pattern graph (weights all 1.0 are not read); SURVEY 8d" if pattern else
             "12 E + 20 V per iteration (SURVEY 8d)")
    if tag == "kron":
        attach_traffic(r, pmc, "pr_pull")
    r["prepare_scalar_ms_per_step"] = round(sum(l["other_ms"] for l in prof), 4)
    r["ms_per_iteration_pull"] = round(sum(l["advance_ms"] for l in prof) / max(1, len(prof)), 4)
    item = {"workload": "PageRank on " + info["name"], "data": info["data"], "data_file": info["file"], "alpha": 0.85,
            "tol": 1e-6, "n_vertices": V, "n_edges": E, "steps": steps, "ms_per_step": round(ms_step, 4), "iterations": iters,
            "ms_per_iteration": round(ms_step / max(1, iters), 4), "mteps": round(E * iters / (ms_step * 1e3), 1),
            "first_call_ms": round(first_ms, 2), "one_shot_ms": round(first_ms, 2),
            "one_shot_note": "fresh graph handle: transpose + pull partitions (+ XCD-blocked layout) + one full run",
            "setup_s": round(t_setup, 1), "roofline": r, "cpu_baseline": None, "cpu_baseline_ncore": None}
    if cpu_on:
        g = O.Csr(csr.row_offsets, csr.column_indices, csr.nonzero_values)
        mine = p.cpu().numpy()
        delta, err, _ = O.pr_f64_trace(g, max(iters + 1, 8), [mine], pattern=pattern)
        item["max_abs_diff_to_float64_same_iterations"] = float(err[0][iters - 1])
        item["float64_iterations"] = O.pr_iterations_from_trace(delta)
        _, it1, ms1 = O.pr_f32(g, max_iterations=3)
        item["cpu_baseline"] = {
            "value": round(E * it1 / (ms1 * 1e3), 2), "unit": "MTEPS", "cores": 1, "kind": "port",
            "sample": "oracle/oracle.c orc_pr_f32 (the reference's push iteration, pr.hxx:107-152, fp32), first %d "
                      "iterations, %.1f s" % (it1, ms1 / 1e3)}
        _, msn = O.pr_omp(g, iterations=5, pattern=pattern)
        item["cpu_baseline_ncore"] = {
            "value": round(E * 5 / (msn * 1e3), 2), "unit": "MTEPS", "cores": ncore, "kind": "port",
            "sample": "oracle/oracle_omp.c orc_pr_omp (the same recurrence as a pull over the transpose on %d host "
                      "threads), 5 iterations, %.1f s" % (ncore, msn / 1e3)}
        del g
    del G, p
    return item


def bench_c5(gr, torch, ctx, dev, sync, cpu_on, args):
    """BASELINE.json configs[4]'s graph (soc-twitter-2010 / its stand-in C5', 21.3 M V / 530 M E) on ONE MI355X: the
    same-graph N = 1 point the 8-GPU figure of `--gpus 8` is to be divided by (strong scaling), with its own parity
    check (the oracle's exact fixed-point characterisation), roofline and CPU baselines."""
    if cpu_on:
        import oracle_lib as O
        ncore = O.omp_threads()
    t0 = time.time()
    props, csr, src, info = load_workload(gr, "twitter", args.data_dir)
    G = gr.build_graph(props, csr, ctx, device=dev)
    V, E = G.get_number_of_vertices(), G.get_number_of_edges()
    d = torch.empty(V, dtype=torch.int32, device=dev)
    t_setup = time.time() - t0
    lb = getattr(gr, args.lb)
    steps = max(3, min(args.steps, 5))
    item = {"workload": "BFS on " + info["name"] + " (BASELINE.json configs[4] graph, single GPU)", "data": info["data"],
            "data_file": info["file"], "n_vertices": V, "n_edges": E, "source": src, "steps": steps,
            "setup_s": None}
    depths = {}
    for label, direction in (("direction_optimized", gr.optimized), ("forward", gr.forward)):
        o = gr.options_t(advance_load_balance=lb, enable_filter=True, filter_algorithm=gr.compact,
                         advance_direction=direction, engine_flags=gr.FLAG_ASYNC_RETURN)
        t1 = time.time()
        gr.bfs(G, src, d, None, ctx, o)  # first call: per-graph preprocessing (symmetry check / transpose, bins), untimed
        sync()
        first = time.time() - t1
        ms = timed(lambda: gr.bfs(G, src, d, None, ctx, o), sync, steps, 1)
        st = gr.run_stats(ctx)
        depths[label] = d.cpu().numpy().copy()
        po = gr.options_t(advance_load_balance=lb, enable_filter=True, filter_algorithm=gr.compact,
                          advance_direction=direction, engine_flags=gr.FLAG_PROFILE)
        prof = best_profile(lambda: gr.bfs(G, src, d, None, ctx, po), lambda: gr.level_profile(ctx), tries=2)
        sizes = [l["frontier_size"] for l in prof] + [0]
        bu = [dict(l, nxt=sizes[i + 1]) for i, l in enumerate(prof) if l["bottom_up"] == 1]
        td = [l for l in prof if l["bottom_up"] != 1]
        r_td = roof(td, lambda l: 12 * l["frontier_size"] + 12 * l["edges"], "top-down level kernels (advance_block / "
                    "binned scatter + sweep)", "12 B per frontier slot + 12 B per traversed edge (SURVEY 8d)") if td else None
        r_bu = roof(bu, lambda l: 3 * (V // 8) + 8 * l["bu_open"] + 8 * l["bu_probes"] + 12 * l["nxt"],
                    "bfs_level_kernel (bottom-up launches)", "3 V/8 + 8 open + 8 probes + 12 found") if bu else None
        item[label] = {"ms_per_step": round(ms, 4), "mteps": round(st["edges_visited"] / (ms * 1e3), 1),
                       "edges_visited_per_step": st["edges_visited"], "search_depth": st["search_depth"],
                       "enact_ms_last": round(st["elapsed_ms"], 4),
                       "first_call_s_incl_graph_preprocessing": round(first, 3),
                       "roofline": r_bu if (r_bu and (not r_td or r_bu["kernel_ms_per_step"] >= r_td["kernel_ms_per_step"]))
                       else r_td,
                       "roofline_other_direction": r_td if (r_bu and r_td and r_bu["kernel_ms_per_step"] >=
                                                            r_td["kernel_ms_per_step"]) else (r_bu if r_td else None),
                       "levels": [[l["frontier_size"], l["edges"], int(l["bottom_up"]), round(l["advance_ms"], 4),
                                   round(l["other_ms"], 4)] for l in prof]}
    item["forward_equals_direction_optimized"] = bool(np.array_equal(depths["forward"], depths["direction_optimized"]))
    item["setup_s"] = round(t_setup, 1)
    if cpu_on:
        g = O.Csr(csr.row_offsets, csr.column_indices, csr.nonzero_values)
        item["property_check_violations"] = int(O.check_bfs(g, src, depths["direction_optimized"]))
        d_n, ms_n, ev_n = O.bfs_omp(g, src)
        item["cpu_baseline_ncore"] = {
            "value": round(ev_n / (ms_n * 1e3), 2), "unit": "MTEPS", "cores": ncore, "kind": "port",
            "sample": "1 full level-synchronous BFS on %d host threads (oracle/oracle_omp.c orc_bfs_omp), %.1f s"
                      % (ncore, ms_n / 1e3),
            "matches_gpu": bool(np.array_equal(d_n, depths["direction_optimized"]))}
        d_q, ms_q, ev_q = O.bfs_queue(g, src)
        item["cpu_baseline"] = {
            "value": round(ev_q / (ms_q * 1e3), 2), "unit": "MTEPS", "cores": 1, "kind": "port",
            "sample": "1 full BFS of the same workload/source on one core (oracle/oracle.c orc_bfs_queue: FIFO "
                      "restatement of examples/algorithms/bfs/bfs_cpu.hxx, identical depths), %.1f s" % (ms_q / 1e3),
            "matches_gpu": bool(np.array_equal(d_q, depths["direction_optimized"]))}
    del G, d
    return item


def bench_bfs_deep(env):
    """Round 5: BFS on a graph of the C2' size with the DEPTH of the published soc-LiveJournal1 (14 levels
    from the hub instead of 7; WORKLOADS["deep"]), forward and direction-optimising, each level's kernel + head time -- what
    a search pays per THIN level is what the 7-level stand-in hides.  Depths compared with the oracle's, array for array."""
    gr, torch, ctx, dev, sync, cpu_on, O, args = (env[k] for k in ("gr", "torch", "ctx", "dev", "sync", "cpu_on", "O", "args"))
    t0 = time.time()
    props, csr, src, info = load_workload(gr, "deep", None)
    G = gr.build_graph(props, csr, ctx, device=dev)
    V, E = G.get_number_of_vertices(), G.get_number_of_edges()
    d = torch.empty(V, dtype=torch.int32, device=dev)
    lb = getattr(gr, args.lb)
    item = {"workload": "BFS on " + info["name"], "data": info["data"], "n_vertices": V, "n_edges": E, "source": src,
            "steps": args.steps}
    depths = {}
    for label, direction in (("forward", gr.forward), ("direction_optimized", gr.optimized)):
        o = gr.options_t(advance_load_balance=lb, enable_filter=True, filter_algorithm=gr.compact,
                         advance_direction=direction, engine_flags=gr.FLAG_ASYNC_RETURN)
        sync()
        t1 = time.perf_counter()
        gr.bfs(G, src, d, None, ctx, o)  # first call on the handle: per-graph preprocessing of this direction
        sync()
        first = (time.perf_counter() - t1) * 1e3
        ms = timed(lambda: gr.bfs(G, src, d, None, ctx, o), sync, args.steps, args.warmup)
        st = gr.run_stats(ctx)
        depths[label] = d.cpu().numpy().copy()
        po = gr.options_t(advance_load_balance=lb, enable_filter=True, filter_algorithm=gr.compact,
                          advance_direction=direction, engine_flags=gr.FLAG_PROFILE)
        prof = best_profile(lambda: gr.bfs(G, src, d, None, ctx, po), lambda: gr.level_profile(ctx))
        td = [l for l in prof if l["bottom_up"] != 1]
        r_td = roof(td, lambda l: 12 * l["frontier_size"] + 12 * l["edges"], "top-down level kernels (advance_block / "
                    "binned scatter + sweep / many levels per launch)", "12 B per frontier slot + 12 B per traversed edge")
        thin = [l for l in prof if l["edges"] < (1 << 20)]
        item[label] = {"ms_per_step": round(ms, 4), "mteps": round(st["edges_visited"] / (ms * 1e3), 1),
                       "edges_visited_per_step": st["edges_visited"], "search_depth": st["search_depth"],
                       "enact_ms_last": round(st["elapsed_ms"], 4), "first_call_ms": round(first, 3),
                       "launch_groups_last": int(st.get("aux", 0)), "roofline": r_td,
                       # (profiled search: one launch group per level, no level merged into a head or a multi-level launch)
                       "thin_levels_us_per_level": round(sum(l["advance_ms"] + l["other_ms"] for l in thin) * 1e3 / max(1, len(thin)), 2),
                       "levels": [[l["frontier_size"], l["edges"], int(l["bottom_up"]), round(l["advance_ms"], 4),
                                   round(l["other_ms"], 4)] for l in prof]}
    item["forward_equals_direction_optimized"] = bool(np.array_equal(depths["forward"], depths["direction_optimized"]))
    item["setup_s"] = round(time.time() - t0, 1)
    if cpu_on:
        g = O.Csr(csr.row_offsets, csr.column_indices, csr.nonzero_values)
        d_q, ms_q, ev_q = O.bfs_queue(g, src)
        item["cpu_baseline"] = {
            "value": round(ev_q / (ms_q * 1e3), 2), "unit": "MTEPS", "cores": 1, "kind": "port",
            "sample": "1 full BFS of the same workload/source on one core (oracle/oracle.c orc_bfs_queue), %.1f s" % (ms_q / 1e3),
            "matches_gpu": bool(np.array_equal(d_q, depths["forward"]) and np.array_equal(d_q, depths["direction_optimized"]))}
    del G, d
    return item


def bench_multi_gpu(args, gr, torch, rank, local_rank, world):
    """N > 1: ONE graph, vertex-range partitioned over the ranks (gunrock_amd/distributed.py).
    N = 8: BASELINE.json configs[4], the soc-twitter-2010 stand-in C5'.  N = 2, 4: N x the single-GPU C2' size
    (weak scaling: per-GPU work fixed; 8 x C2' = 38.8 M V / 552 M E is within 4 % of C5' in edges)."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # GRX_BENCH_BACKEND=gloo lets the N > 1 path be exercised with several ranks sharing one GPU (tests);
    # the driver's multi-GPU runs use nccl (= RCCL), one rank per GPU
    backend = os.environ.get("GRX_BENCH_BACKEND", "nccl")
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    dev = "cuda:%d" % local_rank
    from gunrock_amd import distributed as D
    name = os.environ.get("GRX_BENCH_MULTI_WORKLOAD", "twitter" if world == 8 else args.workload)
    wl = WORKLOADS[name]
    if wl["kind"] not in ("rmat", "rmat_sym"):
        raise SystemExit("multi-GPU bench supports the R-MAT workloads")
    t0 = time.time()
    scale = 1 if name == "twitter" else world
    V = wl["V"] * scale
    entries = wl["entries"] * scale
    bounds = D.vertex_bounds(V, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    props, mine = gr.generate_rows(wl["kind"], V, entries, lo, hi, wl["a"], wl["b"], wl["c"], seed=42)
    topdown_only = args.topdown_only
    mine_in = None
    if wl["kind"] == "rmat" and not topdown_only:  # directed: the bottom-up step needs the in-rows too
        _, mine_in = gr.generate_rows(wl["kind"], V, entries, lo, hi, wl["a"], wl["b"], wl["c"], seed=42, in_rows=True)
    deg = np.diff(mine.row_offsets)
    cdev = dev if dist.get_backend() == "nccl" else "cpu"
    best = torch.tensor([int(deg.max())], dtype=torch.int64, device=cdev)
    dist.all_reduce(best, op=dist.ReduceOp.MAX)
    cand = int(np.argmax(deg)) if int(deg.max()) == int(best.item()) else V
    srct = torch.tensor([cand], dtype=torch.int64, device=cdev)
    dist.all_reduce(srct, op=dist.ReduceOp.MIN)
    src = int(srct.item())
    E = int(mine.number_of_nonzeros)
    et = torch.tensor([E], dtype=torch.int64, device=cdev)
    dist.all_reduce(et)
    overlap = os.environ.get("GRX_BENCH_OVERLAP", "0") == "1"
    eng = D.GrxEngine(props, mine, rank, world, dev, int(et.item()), in_rows=mine_in, overlap=overlap)
    dist_t = eng.new_labels()
    optimized = not topdown_only

    def agree(ok):
        """every rank must take the same branch (a level group carries collectives): MIN over the ranks of a flag"""
        t = torch.tensor([1 if ok else 0], dtype=torch.int64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    # Transport.  Default on RCCL: the collectives of a level group are issued by libgrx itself over its own
    # communicator (one C call per batch of levels, one HIP-graph launch per level once recorded) -- AFTER a self-test:
    # one search through torch.distributed (eager), the same search through the library transport, and the two must
    # agree on depth and on the global traversed-edge count on every rank.  Anything else (librccl not loadable,
    # communicator creation fails, results differ) falls back to the torch.distributed transport on all ranks alike.
    # GRX_DIST_RCCL=0 skips the attempt.
    transport_note = "torch.distributed transport (backend %s)" % backend
    want_lib = os.environ.get("GRX_DIST_RCCL", "1") == "1" and backend == "nccl" and not overlap
    if want_lib:
        ref_st = D.bfs(eng, dist, src, dist_t, optimized=optimized)
        ref_e = torch.tensor([ref_st["edges_visited"]], dtype=torch.int64, device=cdev)
        dist.all_reduce(ref_e)
        ok, why = True, ""
        try:
            eng.enable_library_transport(dist)
        except Exception as e:  # noqa: BLE001
            ok, why = False, "communicator setup failed: %s" % str(e)[:200]
        if agree(ok):
            try:
                st_l = D.bfs(eng, dist, src, dist_t, optimized=optimized)
                got_e = torch.tensor([st_l["edges_visited"]], dtype=torch.int64, device=cdev)
                dist.all_reduce(got_e)
                ok = st_l["search_depth"] == ref_st["search_depth"] and int(got_e.item()) == int(ref_e.item())
                why = "" if ok else "self-test search differs from the torch.distributed search"
            except Exception as e:  # noqa: BLE001
                ok, why = False, "self-test search failed: %s" % str(e)[:200]
            ok = agree(ok)
        else:
            ok = False
        if ok:
            transport_note = "in-library RCCL transport (self-test passed)"
        else:
            eng.disable_library_transport()
            transport_note = "torch.distributed transport (in-library RCCL transport rejected%s)" % (": " + why if why else " on another rank")
    t_setup = time.time() - t0

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    for _ in range(2):  # part of the setup: the first search records the level group as a HIP graph, the second replays it
        st = D.bfs(eng, dist, src, dist_t, optimized=optimized)
    for _ in range(args.warmup):
        st = D.bfs(eng, dist, src, dist_t, optimized=optimized)
    barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        st = D.bfs(eng, dist, src, dist_t, optimized=optimized)
    barrier()
    elapsed = time.perf_counter() - t1
    tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
    ee = torch.tensor([st["edges_visited"], E], dtype=torch.int64, device=cdev)
    dist.all_reduce(ee, op=dist.ReduceOp.SUM)
    edges_total, e_total = int(ee[0].item()), int(ee[1].item())
    ms_per_step = elapsed * 1e3 / args.steps
    mteps = edges_total / (ms_per_step * 1e3)
    captured = eng.group_captured()
    cap_t = torch.tensor([1 if captured else 0], dtype=torch.int64, device=cdev)
    dist.all_reduce(cap_t, op=dist.ReduceOp.MIN)
    # per-level breakdown (untimed, eager, one extra search): kernels before the exchange | bitmap all-to-all | kernels
    # after it + statistics all-reduce, per rank; rank 0 reports the max and the mean over the ranks
    levels_breakdown = None
    try:
        recs = D.profile_bfs(eng, dist, src, dist_t, optimized=optimized)
        n_g = torch.tensor([len(recs)], dtype=torch.int64, device=cdev)
        dist.all_reduce(n_g, op=dist.ReduceOp.MIN)
        n_g = int(n_g.item())
        arr = torch.tensor([[r["pre_ms"], r["exchange_ms"], r["post_ms"]] for r in recs[:n_g]], dtype=torch.float64,
                           device=cdev).reshape(n_g, 3)
        mx, sm = arr.clone(), arr.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        levels_breakdown = {"columns": "per level group: kernels before the exchange ms, exchange ms, kernels after + "
                                       "all-reduce ms (max over ranks | mean over ranks)",
                            "max_over_ranks": [[round(float(x), 4) for x in row] for row in mx.tolist()],
                            "mean_over_ranks": [[round(float(x) / world, 4) for x in row] for row in sm.tolist()]}
        # PREDICTION of the step from a model (profiles/dist_comm_model.json: small-message latency of the two collectives of a
        # level group + the bitmap bytes over one xGMI link) and the kernel times just measured: sum over the level groups of
        # max-over-ranks kernel time + all-to-all + all-reduce.  Printed beside the measured ms_per_step so that the record
        # of a multi-GPU run checks the model.
        try:
            cm = json.load(open(os.path.join(ROOT, "profiles", "dist_comm_model.json")))
            per_pair = eng.S // 8
            t_x = cm["alpha_all_to_all_us"] * 1e-3 + per_pair / (cm["link_GBs"] * cm["link_efficiency"] * 1e9) * 1e3
            t_r = cm["alpha_all_reduce_us"] * 1e-3
            kern = [row[0] + row[2] for row in mx.tolist()]
            levels_breakdown["predicted_ms"] = round(sum(kern) + len(kern) * (t_x * eng.parts + t_r), 4)
            levels_breakdown["predicted_from"] = ("sum over %d level groups of (max-over-ranks kernels %.3f ms total) + %d x "
                                                  "(all-to-all %.4f ms x %d + all-reduce %.4f ms); model: profiles/dist_comm_model.json"
                                                  % (len(kern), sum(kern), len(kern), t_x, eng.parts, t_r))
        except Exception as e:  # noqa: BLE001
            levels_breakdown["predicted_ms"] = None
            levels_breakdown["predicted_from"] = "model not available: %s" % str(e)[:120]
    except Exception as e:  # noqa: BLE001  (diagnostics only)
        levels_breakdown = {"error": str(e)[:200]}
    check = None
    if os.environ.get("GRX_BENCH_CHECK", "0") == "1":
        # parity of the partitioned search (test harness; outside the timed region): the owned label slices are
        # gathered on rank 0 over a gloo group and checked with the oracle's exact fixed-point characterisation
        # against the WHOLE graph, which only rank 0 generates
        gg = dist.new_group(backend="gloo")
        mine_labels = dist_t.cpu().numpy()[:hi - lo].copy()
        parts = [None] * world if rank == 0 else None
        dist.gather_object((lo, hi, mine_labels), parts, dst=0, group=gg)
        if rank == 0:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            _, full = gr.generate(wl["kind"], V, entries, wl["a"], wl["b"], wl["c"], seed=42)
            g = O.Csr(full.row_offsets, full.column_indices, full.nonzero_values)
            labels = np.full(V, -1, np.int32)
            for plo, phi, part in parts:
                labels[plo:phi] = part
            reached = labels != np.iinfo(np.int32).max
            check = {"property_check_violations": int(O.check_bfs(g, src, labels)),
                     "edges_match_reached_out_degrees": bool(int(np.diff(g.row_offsets)[reached].sum()) == edges_total),
                     "vertices_reached": int(reached.sum())}
        dist.barrier(group=gg)
    if rank == 0:
        if name == "twitter":
            wname = "BFS on " + wl["name"] + " (BASELINE.json configs[4]), src = max out-degree vertex"
        else:
            wname = ("BFS on %d x the single-GPU C2' size: R-MAT(0.57,0.19,0.19,0.05) %d V / %d E, src = max out-degree "
                     "vertex (weak scaling)" % (world, V, e_total))
        print(json.dumps({
            "metric": "MTEPS (million traversed edges/sec) BFS", "value": round(mteps, 1), "unit": "MTEPS",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            # N = 8 runs ONE fixed graph (BASELINE configs[4]; its single-GPU point is c5_1gpu of the N = 1 line): strong
            # scaling.  Other N run N x the single-GPU graph: weak scaling.
            "higher_is_better": True, "scaling": "strong" if name == "twitter" else "weak", "vs_baseline": None, "dtype": "int32",
            "data": "synthetic", "predicted_ms": (levels_breakdown or {}).get("predicted_ms"),
            "config": {"workload": wname, "n_vertices": V, "n_edges": e_total, "source": src,
                       "per_gpu_edges": e_total // world,
                       "parallelism": "vertex-range partition over %d GPUs, labels sharded (V/P per rank); per level "
                                      "one all-to-all of fixed-size bitmaps (%d B per pair) + a 4-word all-reduce over "
                                      "RCCL; device-side direction choice; %s%s"
                                      % (world, eng.S // 8, eng.transport_description(),
                                         "; top-down halves overlapped" if overlap else ""),
                       "advance_direction": "forward (top-down)" if topdown_only else "optimized (Beamer, decided on "
                                            "the device from all-reduced statistics)",
                       "edges_visited_per_step": edges_total, "search_depth": st["search_depth"],
                       "setup_s": round(t_setup, 1), "backend": backend, "transport": transport_note,
                       "rccl_ranks_seen": int(dist.get_world_size()),
                       "n1_path": "python bench.py --gpus 1 runs the single-GPU engine (no partition, no collective); "
                                  "this line is the partitioned path",
                       "like_for_like_n1": "the N = 1 line's TOP LEVEL is the forward search (BASELINE configs[1] as written); "
                                           "this line's search is direction-optimising unless --topdown-only: its single-GPU "
                                           "counterpart is config.sections.bfs_do of the N = 1 line (C2' stand-in) and "
                                           "c5_1gpu.do_mteps (C5' stand-in) -- take scaling efficiency against those",
                       "level_group_captured_as_hip_graph_on_all_ranks": bool(cap_t.item()),
                       "per_level_breakdown": levels_breakdown, "parity_check": check},
            "roofline": None, "cpu_baseline": None}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
