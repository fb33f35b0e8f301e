"""Per-launch rocprofv3 numbers of the BFS level kernel, split by what the launch did.

    python tools/pmc_json.py <prof dir of tools/profile.sh> [bottom-up level positions, default 1,2,3] > profiles/rNN_bench_pmc.json

The single per-level kernel (bfs_level_kernel) runs a level top-down or bottom-up as the head
kernel decided, so its dispatches are classified by POSITION inside a search: a search starts at
a bfs_init_kernel dispatch; searches whose reset was a bfs_reset_kernel (labels + bitmaps) are
direction-optimising runs, whose level dispatches at the given 0-based positions ran bottom-up
(bench.py's `all_levels` shows the same flags); searches without them are top-down-only runs,
whose positions 1 and 2 are the two fat top-down levels of the LJ stand-in.

Every counter group comes from its own rocprofv3 pass (--kernel-trace --pmc <group>); durations
come from the --kernel-trace --stats pass.  FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 a wide
coalesced read is tallied at half its bytes and a random 4-byte gather at one 64-B sector
(MI355X_MICROARCH.md, HBM section; tools/calibrate_fetch.py on this box) -- the values here
are the raw counters x 1024, uncorrected."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
bu_pos = set(int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,3").split(","))


def classify(rows):
    """rows: (dispatch id, kernel name) in launch order -> {dispatch id: class}"""
    cls, cur = {}, None
    searches = []
    reset_seen = False
    for did, name in rows:
        if "bfs_reset_kernel" in name:
            reset_seen = True
        elif "bfs_init_kernel" in name:
            cur = {"convert": reset_seen, "levels": []}
            searches.append(cur)
            reset_seen = False
        elif cur is not None and "bfs_level_kernel" in name:
            cur["levels"].append(did)
    for s in searches:
        for pos, did in enumerate(s["levels"]):
            if s["convert"]:
                cls[did] = "bottom_up" if pos in bu_pos else ("do_topdown_first" if pos == 0 else "do_other")
            else:
                cls[did] = "topdown_fat" if pos in (1, 2) else "td_other"
    return cls, searches


result = {"source": "tools/profile.sh passes in " + os.path.basename(out.rstrip("/")),
          "units": "bytes = FETCH_SIZE / WRITE_SIZE x 1024, raw (gfx950 tallies wide coalesced reads at 1/2, "
                   "random 4-byte gathers at one 64-B sector each); durations from the kernel-trace pass",
          "bottom_up_positions": sorted(bu_pos), "classes": {}}
acc = collections.defaultdict(lambda: collections.defaultdict(list))

kt = glob.glob(os.path.join(out, "kt", "**", "p_kernel_trace.csv"), recursive=True)
if kt:
    rows = sorted(csv.DictReader(open(kt[0])), key=lambda r: int(r["Start_Timestamp"]))
    ids = [(i, r["Kernel_Name"]) for i, r in enumerate(rows)]
    cls, searches = classify(ids)
    for i, r in enumerate(rows):
        if i in cls:
            acc[cls[i]]["duration_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    result["searches_direction_optimising"] = sum(1 for s in searches if s["convert"])
    result["searches_topdown_only"] = sum(1 for s in searches if not s["convert"])

for f in sorted(glob.glob(os.path.join(out, "pmc*", "**", "p_counter_collection.csv"), recursive=True)):
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        per.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], {}])[1][r["Counter_Name"]] = float(r["Counter_Value"])
    order = sorted(per)
    cls, _ = classify([(d, per[d][0]) for d in order])
    for d in order:
        if d in cls:
            for c, v in per[d][1].items():
                acc[cls[d]][c].append(v)

for k, counters in acc.items():
    o = {"launches_seen": len(counters.get("duration_us", []))}
    for c, vals in counters.items():
        o[c + "_per_launch"] = sum(vals) / max(1, len(vals))
    if "FETCH_SIZE_per_launch" in o:
        o["fetch_bytes_per_launch"] = o["FETCH_SIZE_per_launch"] * 1024.0
    if "WRITE_SIZE_per_launch" in o:
        o["write_bytes_per_launch"] = o["WRITE_SIZE_per_launch"] * 1024.0
    result["classes"][k] = o
print(json.dumps(result, indent=1))
