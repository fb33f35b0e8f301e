"""rocprofv3 numbers of the engine's dominant kernels, per launch / per search, for bench.py's `traffic` fields.

    python tools/pmc_json.py <prof dir> [<prof dir> ...] > profiles/rNN_bench_pmc.json

Each <prof dir> is the output of tools/profile.sh for ONE command (kt/ = --kernel-trace --stats pass, pmc*/ = one
--pmc group per pass).  Dispatches are grouped into SEARCHES (a search starts at a *_init_kernel dispatch) and the
kernels of interest are classified by name and by their position inside the search:

  bottom_up            bfs_level_kernel launches 1..3 of a direction-optimising search (after a bfs_reset_kernel);
                       per LAUNCH
  topdown_fat          launches 1 and 2 of a forward search, bfs_level_bin_kernel (+ the bfs_sweep_kernel / bfs_claim_kernel of
                       the same level when the level ran binned): per LEVEL (= what bench.py's forward profile calls a launch)
  sssp_unit_weights    all level launches of a unit-weight search (round 4: bfs_level_bin_kernel -- the search runs on the BFS
                       engine; before: sssp_level_kernel); per SEARCH (with many levels per launch the
  sssp_weighted_1_1000 all sssp_nf_level_kernel launches           launch count is not a unit of work)
  sssp_weighted_dense  round 4: weighted search on a dense graph -- sssp_level_kernel + sssp_rscatter_kernel + sssp_rsweep_kernel
                       launches (binned relaxation of the fat levels); per SEARCH
  pr_pull              pr_pull_xcd_kernel / pr_pull_kernel + pr_long* + pr_combine_kernel launches that did work
                       (launches queued past convergence exit at once and are dropped: < 10 % of the largest value);
                       per ITERATION

Only the LAST search of a kind in a process is used (warm caches, layouts built).  Every counter group comes from
its own rocprofv3 pass; durations come from the kernel-trace pass.  FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 a
wide coalesced read is tallied at half its bytes and a random 4-byte gather at one 64-B sector
(MI355X_MICROARCH.md, HBM section; tools/calibrate_fetch.py) -- values here are the raw counters x 1024.
`source_sha` = hash of gunrock_amd/csrc at the time of profiling: bench.py attaches these numbers only while the
sources still hash to it."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_sha():
    sys.path.insert(0, ROOT)
    from gunrock_amd.build import source_sha as f
    return f()


def searches_of(rows):
    """rows: [(id, kernel name)] in launch order -> list of searches {kind, do, kernels: [(id, name)]}"""
    out, cur, reset_seen = [], None, False
    for did, name in rows:
        if "bfs_reset_kernel" in name:  # (GRX_SEED_IN_RESET=0)
            reset_seen = True
        kind = None
        if "bfs_reset_seed_kernel" in name:  # problem.reset() + seed of a direction-optimising search, one launch
            kind = "bfs"
            reset_seen = True
        elif "bfs_init_kernel" in name or "bfs_fwd_reset_seed_kernel" in name or "bfs_fwd_start_kernel" in name:  # (reset + seed [+ level 0] of a forward search in one launch)
            kind = "bfs"
        elif "sssp_init_kernel" in name:
            kind = "sssp"
        elif "pr_init_kernel" in name:
            kind = "pr"
        if kind:
            cur = {"kind": kind, "do": reset_seen, "kernels": []}
            out.append(cur)
            reset_seen = False
        elif cur is not None:
            cur["kernels"].append((did, name))
    return out


def classify(rows):
    """-> {dispatch id: (class, unit index)}: values of one unit are summed, units are averaged"""
    cls = {}
    ss = searches_of(rows)
    last = {}
    for s in ss:
        key = (s["kind"], s["do"], any("sssp_nf_level" in n for _, n in s["kernels"]),
               any("sssp_rscatter" in n for _, n in s["kernels"]))
        last[key] = s
    for (kind, do, nf, rb), s in last.items():
        if kind == "bfs" and do:
            lv = [d for d, n in s["kernels"] if "bfs_level_kernel" in n]
            for pos, d in enumerate(lv):
                if pos in (1, 2, 3):
                    cls[d] = ("bottom_up", pos)
        elif kind == "bfs":
            # a level group of a forward run: level kernel (a no-op on a binned level when the second scatter is in use),
            # scatter kernel (second version), claim / sweep kernel
            lv = [d for d, n in s["kernels"] if "bfs_level_bin_kernel" in n]
            if len(lv) > 64 or any("sssp_depth_to_dist" in n for _, n in s["kernels"]):
                # a unit-weight SSSP: since round 4 it IS this forward BFS (grx_sssp.hip, all weights equal; the pass that turns
                # depths into distances follows the search) -- or a high-diameter BFS (road stand-in), the same kernels: every
                # level launch of the search (many levels per launch on such graphs), per SEARCH
                for d in lv:
                    cls[d] = ("sssp_unit_weights", 0)
                continue
            sc = [d for d, n in s["kernels"] if "bfs_scatter2_kernel" in n]
            cl = [d for d, n in s["kernels"] if "bfs_claim_kernel" in n or "bfs_sweep" in n]
            if sc and len(sc) == len(cl) and len(sc) < len(lv) + 2 and len(sc) <= 4:
                # round 5, exact schedule of a repeated source (grx_graph::bin_exact; profiled searches too): scatter + sweep are
                # launched in the fat groups ONLY, and those groups carry no level kernel -- every pair is one fat level
                for pos, (d_sc, d_cl) in enumerate(zip(sc, cl)):
                    cls[d_sc] = ("topdown_fat", pos)
                    cls[d_cl] = ("topdown_fat", pos)
                continue
            for pos in (1, 2):
                for group in (lv, sc, cl):
                    if pos < len(group):
                        cls[group[pos]] = ("topdown_fat", pos)
        elif kind == "sssp":
            # rb: a weighted search on a dense graph, fat levels as binned relaxation (grx_relax.hpp): scatter + sweep kernels
            name = "sssp_weighted_1_1000" if nf else ("sssp_weighted_dense" if rb else "sssp_unit_weights")
            for d, n in s["kernels"]:
                if any(k in n for k in ("sssp_nf_level_kernel", "sssp_level_kernel", "advance_kernel", "sssp_rscatter_kernel",
                                        "sssp_rsweep_kernel")):
                    cls[d] = (name, 0)
        elif kind == "pr":
            it = -1
            for d, n in s["kernels"]:
                if "pr_scalar_kernel" in n:  # one per iteration (pr_prepare_kernel runs once on the XCD-blocked path)
                    it += 1
                if any(k in n for k in ("pr_pull", "pr_long", "pr_combine")):
                    cls[d] = ("pr_pull", it)
    return cls


result = {"source_sha": source_sha(),
          "units": "bytes = FETCH_SIZE / WRITE_SIZE x 1024, raw (gfx950 tallies wide coalesced reads at 1/2, random 4-byte "
                   "gathers at one 64-B sector each); durations from the kernel-trace pass; `per_launch` = per unit of the "
                   "class (see tools/pmc_json.py: launch, level, search or iteration)",
          "sources": [], "classes": {}}
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))  # class -> counter -> unit -> sum

for out in sys.argv[1:]:
    result["sources"].append(os.path.basename(out.rstrip("/")))
    kt = glob.glob(os.path.join(out, "kt", "**", "p_kernel_trace.csv"), recursive=True)
    if kt:
        rows = sorted(csv.DictReader(open(kt[0])), key=lambda r: int(r["Start_Timestamp"]))
        cls = classify([(i, r["Kernel_Name"]) for i, r in enumerate(rows)])
        for i, r in enumerate(rows):
            if i in cls:
                c, u = cls[i]
                acc[c]["duration_us"][u] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                acc[c]["dispatches"][u] += 1
    for f in sorted(glob.glob(os.path.join(out, "pmc*", "**", "p_counter_collection.csv"), recursive=True)):
        per = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            per.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], {}])[1][r["Counter_Name"]] = float(r["Counter_Value"])
        order = sorted(per)
        cls = classify([(d, per[d][0]) for d in order])
        for d in order:
            if d in cls:
                c, u = cls[d]
                for name, v in per[d][1].items():
                    acc[c][name][u] += v

for c, counters in acc.items():
    o = {}
    units = counters.get("duration_us", {})
    keep = set(units)
    if c == "pr_pull" and units:  # iterations queued past convergence did nothing
        top = max(units.values())
        keep = {u for u, v in units.items() if v >= 0.1 * top}
    o["units_seen"] = len(keep)
    for name, per_unit in counters.items():
        vals = [v for u, v in per_unit.items() if u in keep] if name in ("duration_us", "dispatches") else None
        if vals is None:
            # a counter pass may see the same units (positions are stable run to run)
            vals = [v for u, v in per_unit.items() if u in keep or not keep]
            if c == "pr_pull" and vals:
                top = max(vals)
                vals = [v for v in vals if v >= 0.1 * top]
        if vals:
            o[name + "_per_launch"] = sum(vals) / len(vals)
    if "FETCH_SIZE_per_launch" in o:
        o["fetch_bytes_per_launch"] = o["FETCH_SIZE_per_launch"] * 1024.0
    if "WRITE_SIZE_per_launch" in o:
        o["write_bytes_per_launch"] = o["WRITE_SIZE_per_launch"] * 1024.0
    if "duration_us_per_launch" in o:
        o["duration_us_per_launch"] = o["duration_us_per_launch"]
    result["classes"][c] = o
print(json.dumps(result, indent=1))
