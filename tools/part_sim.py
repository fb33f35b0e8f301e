"""P ranks of the partitioned BFS in ONE process on ONE GPU, stepped in lockstep, the exchange done with device copies:
what each rank's kernels cost per level when nothing else runs on the device (ranks sharing a GPU through separate
processes overlap each other's kernels, so their event times say little).  Prints, per level group, the max and the sum
over the ranks of the kernel time in front of the exchange (head, level kernels) and behind it (post, statistics), and
the totals beside the single-GPU engine's search on the same graph.

    python tools/part_sim.py [lj|twitter|kron|small] [P ...]        e.g.  python tools/part_sim.py twitter 1 8
    env: PART_SIM_DIR=forward|optimized|both (default both), PART_SIM_JSON=<file> appends one JSON record per run

This is a MEASUREMENT AID (and the input of bench.py's predicted_ms at N > 1), not the product path: a real run has one
process per GPU and RCCL between them (gunrock_amd/distributed.py)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gunrock_amd as gr  # noqa: E402
from gunrock_amd import distributed as D  # noqa: E402
from bench import WORKLOADS  # noqa: E402


def build_engines(wl, P, dev="cuda:0", want_in=True):
    V, entries = wl["V"], wl["entries"]
    bounds = D.vertex_bounds(V, P)
    engs, e_tot, best = [], 0, (-1, 0)
    rows = []
    for r in range(P):
        lo, hi = int(bounds[r]), int(bounds[r + 1])
        props, mine = gr.generate_rows(wl["kind"], V, entries, lo, hi, wl["a"], wl["b"], wl["c"], seed=42)
        mine_in = None
        if wl["kind"] == "rmat" and want_in:
            _, mine_in = gr.generate_rows(wl["kind"], V, entries, lo, hi, wl["a"], wl["b"], wl["c"], seed=42, in_rows=True)
        deg = np.diff(mine.row_offsets)
        if int(deg.max()) > best[0]:
            best = (int(deg.max()), int(np.argmax(deg)))
        e_tot += int(mine.number_of_nonzeros)
        rows.append((props, mine, mine_in))
    for r in range(P):
        props, mine, mine_in = rows[r]
        engs.append(D.GrxEngine(props, mine, r, P, dev, e_tot, in_rows=mine_in))
        rows[r] = None
    return engs, best[1], e_tot, V


def lockstep_search(engs, src, labels, optimized, timed=True):
    """one search, level groups in lockstep; -> (per level [(pre_ms per rank), (post_ms per rank)], stats per rank)"""
    P = len(engs)
    sw = engs[0].slice_words
    for e, d in zip(engs, labels):
        with torch.cuda.stream(e.stream):
            e.begin(src, d, optimized)
    torch.cuda.synchronize()

    def allreduce():
        tot = sum(e.stats_local for e in engs)
        for e in engs:
            e.stats_global.copy_(tot)
        torch.cuda.synchronize()

    allreduce()
    levels = []
    for _ in range(200):
        pre, post = [], []
        for e in engs:
            with torch.cuda.stream(e.stream):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                e.pre(0)
                b.record()
            b.synchronize()
            pre.append(a.elapsed_time(b))
        if P > 1:
            for r, e in enumerate(engs):  # recv_r[j] <- send_j[r]
                for j, o in enumerate(engs):
                    e.recv[j * sw:(j + 1) * sw].copy_(o.send[r * sw:(r + 1) * sw])
            torch.cuda.synchronize()
        for e in engs:
            with torch.cuda.stream(e.stream):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                e.post()
                b.record()
            b.synchronize()
            post.append(a.elapsed_time(b))
        torch.cuda.synchronize()
        allreduce()
        levels.append((pre, post))
        if all(e.poll()[0] for e in engs):
            break
    stats = []
    for e in engs:
        with torch.cuda.stream(e.stream):
            stats.append(e.end())
    return levels, stats


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "lj"
    Ps = [int(x) for x in sys.argv[2:]] or [1, 8]
    wl = WORKLOADS[name]
    dirs = {"forward": [False], "optimized": [True], "both": [False, True]}[os.environ.get("PART_SIM_DIR", "both")]
    out = os.environ.get("PART_SIM_JSON")
    for P in Ps:
        t0 = time.time()
        engs, src, e_tot, V = build_engines(wl, P)
        labels = [e.new_labels() for e in engs]
        print("== %s  P = %d  (V %d, E %d, src %d; setup %.1f s)" % (name, P, V, e_tot, src, time.time() - t0), flush=True)
        for optimized in dirs:
            lockstep_search(engs, src, labels, optimized)  # first search on the handles: per-graph builds
            best = None
            for _ in range(int(os.environ.get("PART_SIM_REPS", "3"))):
                levels, stats = lockstep_search(engs, src, labels, optimized)
                tot = sum(max(p) + max(q) for p, q in levels)
                if best is None or tot < best[0]:
                    best = (tot, levels, stats)
            tot, levels, stats = best
            edges = sum(s["edges_visited"] for s in stats)
            depth = stats[0]["search_depth"]
            rec = {"workload": name, "P": P, "direction": "optimized" if optimized else "forward", "search_depth": depth,
                   "edges_visited": edges, "kernel_ms_max_over_ranks": round(tot, 4),
                   "levels_max_ms": [[round(max(p), 4), round(max(q), 4)] for p, q in levels],
                   "levels_mean_ms": [[round(sum(p) / P, 4), round(sum(q) / P, 4)] for p, q in levels]}
            if P == 1:
                # the same search through the whole-search call (one rank: the single-GPU engine itself)
                e = engs[0]
                ts = []
                for _ in range(5):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    st = D.bfs(e, None, src, labels[0], optimized=optimized)
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t1) * 1e3)
                rec["whole_search_call_ms"] = round(sorted(ts)[len(ts) // 2], 4)
                rec["whole_search_enact_ms"] = round(st["elapsed_ms"], 4)
            print("  %-9s depth %2d edges %d | kernels, max over ranks: %.3f ms%s" %
                  (rec["direction"], depth, edges, tot,
                   (" | whole-search call %.3f ms (enact %.3f)" % (rec["whole_search_call_ms"], rec["whole_search_enact_ms"])) if P == 1 else ""))
            print("    per level group [pre, post] max ms : " + " ".join("[%.3f %.3f]" % (max(p), max(q)) for p, q in levels))
            if P > 1:
                print("    per level group [pre, post] mean ms: " + " ".join("[%.3f %.3f]" % (sum(p) / P, sum(q) / P) for p, q in levels))
            if out:
                with open(out, "a") as f:
                    f.write(json.dumps(rec) + "\n")
        del engs, labels
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
