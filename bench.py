#!/usr/bin/env python
"""bench.py -- MTEPS of the BFS hot path on the BASELINE.json configs[1] workload.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload lj|kron|road|small]

A "step" is one full pass of the hot path: problem reset + the enact loop
(frontier seed -> convergence) of one BFS from the fixed source over the graph
already resident in HBM.  value = traversed edges of all ranks / wall time of K
steps (barrier + synchronize on both sides, max over ranks), in MTEPS
(= edges_visited / (elapsed_ms * 1000), include/gunrock/util/performance.hxx:225-229
of the reference).

Extra objects on the JSON line:
  roofline      advance kernel: algorithmic bytes (12*|F| + 12*m_F per
                launch, summed over the launches of one BFS) / HIP-event time of
                those launches on the engine's stream, against 8 TB/s HBM.
  cpu_baseline  the oracle (port of the reference's priority-queue CPU path)
                timed on this box's host, rank 0, N=1 only, bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[1]: soc-LiveJournal1 stand-in (SURVEY.md 8d C2')
    "lj": dict(kind="rmat", V=4_847_571, entries=68_993_773, a=0.57, b=0.19, c=0.19,
               name="BFS soc-LiveJournal1 stand-in: R-MAT(0.57,0.19,0.19,0.05) 4,847,571 V / 68,993,773 E, "
                    "src = max out-degree vertex"),
    "kron": dict(kind="rmat_sym", V=1 << 21, entries=91_042_010, a=0.57, b=0.19, c=0.19,
                 name="BFS kron_g500-logn21 stand-in"),
    "road": dict(kind="road", V=4894 * 4894, entries=0, a=0.602, b=0.0, c=0.0,
                 name="BFS road_usa stand-in: 4894x4894 lattice p=0.602"),
    "small": dict(kind="rmat", V=1 << 18, entries=4_000_000, a=0.57, b=0.19, c=0.19,
                  name="BFS R-MAT 262,144 V / 4,000,000 E (smoke size)"),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="lj", choices=sorted(WORKLOADS))
    ap.add_argument("--lb", default="merge_path")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--topdown-only", action="store_true",
                    help="advance_direction=forward: every level runs the top-down advance kernel")
    args = ap.parse_args()

    import torch
    import gunrock_amd as gr

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist_on = world > 1
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # GRX_BENCH_BACKEND=gloo lets the N > 1 path be exercised with several ranks sharing
        # one GPU (tests); the driver's multi-GPU runs use nccl (= RCCL), one rank per GPU
        backend = os.environ.get("GRX_BENCH_BACKEND", "nccl")
        local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = "cuda:%d" % local_rank
    torch.cuda.set_device(local_rank)

    wl = WORKLOADS[args.workload]
    t0 = time.time()
    if dist_on:
        # ---- N > 1: ONE graph, N times the single-GPU size (weak scaling), vertex-range
        # partitioned; per level an RCCL all-to-all of the non-owned discoveries
        # (gunrock_amd/distributed.py).  At N = 8 this is BASELINE.json configs[4] scale
        # (soc-twitter-2010: 21 M V / 530 M E) -- here 38.8 M V / 552 M E.
        from gunrock_amd import distributed as D
        if wl["kind"] not in ("rmat", "rmat_sym"):
            raise SystemExit("multi-GPU bench supports the R-MAT workloads")
        V = wl["V"] * world
        entries = wl["entries"] * world
        bounds = D.vertex_bounds(V, world)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        props, mine = gr.generate_rows(wl["kind"], V, entries, lo, hi, wl["a"], wl["b"], wl["c"], seed=42)
        topdown_only = args.topdown_only
        mine_in = None
        if wl["kind"] == "rmat" and not topdown_only:  # directed: the bottom-up step needs the in-rows too
            _, mine_in = gr.generate_rows(wl["kind"], V, entries, lo, hi, wl["a"], wl["b"], wl["c"], seed=42,
                                          in_rows=True)
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
        dist_t = torch.empty(V, dtype=torch.int32, device=dev)
        t_setup = time.time() - t0

        def barrier():
            dist.barrier()
            torch.cuda.synchronize()

        for _ in range(args.warmup):
            st = D.bfs(eng, dist, src, dist_t, optimized=not topdown_only)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            st = D.bfs(eng, dist, src, dist_t, optimized=not topdown_only)
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
        if rank == 0:
            print(json.dumps({
                "metric": "MTEPS (million traversed edges/sec) BFS", "value": round(mteps, 1), "unit": "MTEPS",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
                "data": "synthetic",
                "config": {"workload": "BFS R-MAT(0.57,0.19,0.19,0.05), %d x the single-GPU size: %d V / %d E, "
                                       "src = max out-degree vertex" % (world, V, e_total),
                           "n_vertices": V, "n_edges": e_total, "source": src,
                           "parallelism": "vertex-range partition over %d GPUs; per level one RCCL "
                                          "all_to_all_single of fixed-size bitmaps (%d B per pair) + a 4-word "
                                          "all_reduce; device-side direction choice; host polls once per batch "
                                          "of levels%s" % (world, eng.S // 8,
                                                           "; top-down halves overlapped" if overlap else ""),
                           "advance_direction": "forward (top-down)" if topdown_only else "optimized (Beamer, "
                                                "decided on the device from all-reduced statistics)",
                           "edges_visited_per_step": edges_total, "search_depth": st["search_depth"],
                           "setup_s": round(t_setup, 1)},
                "roofline": None, "cpu_baseline": None}))
        dist.barrier()
        dist.destroy_process_group()
        return

    props, csr = gr.generate(wl["kind"], wl["V"], wl["entries"], wl["a"], wl["b"], wl["c"], seed=42)
    deg = np.diff(csr.row_offsets)
    src = int(np.argmax(deg))
    ctx = gr.multi_context_t(local_rank)
    G = gr.build_graph(props, csr, ctx, device=dev)
    V, E = G.get_number_of_vertices(), G.get_number_of_edges()
    dist_t = torch.empty(V, dtype=torch.int32, device=dev)
    t_setup = time.time() - t0

    lb = getattr(gr, args.lb)
    direction = gr.forward if args.topdown_only else gr.optimized
    opts = gr.options_t(advance_load_balance=lb, enable_filter=True, filter_algorithm=gr.compact,
                        advance_direction=direction)

    def barrier():
        torch.cuda.synchronize()
        ctx.synchronize()

    for _ in range(args.warmup):
        gr.bfs(G, src, dist_t, None, ctx, opts)
    barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        gr.bfs(G, src, dist_t, None, ctx, opts)
    barrier()
    elapsed = time.perf_counter() - t1
    st = gr.run_stats(ctx)
    edges_rank = st["edges_visited"]
    enact_ms = st["elapsed_ms"]
    edges_total = edges_rank
    ms_per_step = elapsed * 1e3 / args.steps
    mteps = edges_total / (ms_per_step * 1e3)

    out = None
    if rank == 0:
        # ---- rooflines, HIP events on the engine stream (GRX_FLAG_PROFILE records per-level
        # kernel times with events on ctx's stream and syncs after every level)
        def profile(direction_):
            popts = gr.options_t(advance_load_balance=lb, enable_filter=True, filter_algorithm=gr.compact,
                                 advance_direction=direction_, engine_flags=gr.FLAG_PROFILE)
            best_ = None
            for _ in range(3):
                gr.bfs(G, src, dist_t, None, ctx, popts)
                prof_ = gr.level_profile(ctx)
                t_ = sum(l["advance_ms"] for l in prof_)
                if best_ is None or t_ < best_[0]:
                    best_ = (t_, prof_)
            return best_[1]

        # HBM traffic of the same kernels from the committed rocprofv3 PMC passes of this round
        # (FETCH_SIZE / WRITE_SIZE, separate --pmc runs; tools/profile.sh, profiles/r1_bench_pmc.json)
        pmc = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r1_bench_pmc.json")))
        except Exception:
            pass

        def roof(levels, bytes_of, kernel):
            ms = sum(l["advance_ms"] for l in levels)
            byt = sum(bytes_of(l) for l in levels)
            ach = byt / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "kernel": kernel,
                    "launches_per_step": len(levels), "alg_bytes_per_step": int(byt),
                    "kernel_ms_per_step": round(ms, 4),
                    "avg_launch_us": round(ms * 1e3 / max(1, len(levels)), 2)}

        # top-down advance kernel: BASELINE.md / SURVEY 8d, 12 B per frontier slot + 12 B per edge
        td_bytes = lambda l: 12 * l["frontier_size"] + 12 * l["edges"]
        # bottom-up kernel: bytes it must touch -- visited r/w + next-frontier bitmap (3 * V/8),
        # two in-offsets per open vertex (8), column index + frontier-bitmap word per probed
        # in-edge (8), label + two out-offsets per discovered vertex (12)
        bu_bytes = lambda l, nxt: 3 * (V // 8) + 8 * l["bu_open"] + 8 * l["bu_probes"] + 12 * nxt
        prof_td = profile(gr.forward)
        roofline_td = roof(prof_td, td_bytes, "bfs_level_kernel (top-down advance, forward-only run)")
        def attach_traffic(r, cls):
            # per-launch HBM-side bytes of the same launches from the committed rocprofv3 PMC passes
            k = (pmc or {}).get("classes", {}).get(cls)
            if not k or args.workload != "lj" or "fetch_bytes_per_launch" not in k:
                return
            r["traffic"] = int(k["fetch_bytes_per_launch"] + k.get("write_bytes_per_launch", 0.0))
            r["traffic_source"] = ("profiles/r1_bench_pmc.json class '%s': FETCH_SIZE + WRITE_SIZE per launch, separate "
                                   "--pmc passes of this command (raw counters; gfx950 tallies coalesced reads at 1/2)" % cls)
            if "duration_us_per_launch" in k:
                r["rocprof_avg_launch_us"] = round(k["duration_us_per_launch"], 2)

        attach_traffic(roofline_td, "topdown_fat")
        if roofline_td.get("traffic") is not None:
            roofline_td["traffic_scope"] = "average of the two fat top-down levels (the launches that carry 99 % of the bytes)"
            # the PMC class covers the two fat levels only: quote the event time of the same two launches
            fat = sorted(prof_td, key=lambda l: -l["edges"])[:2]
            roofline_td["fat_levels_avg_launch_us"] = round(sum(l["advance_ms"] for l in fat) * 1e3 / max(1, len(fat)), 2)
        roofline_td["levels"] = [[l["frontier_size"], l["edges"], round(l["advance_ms"], 4), round(l["other_ms"], 4)]
                                 for l in prof_td]
        if args.topdown_only:
            roofline, roofline_other = roofline_td, None
        else:
            prof_do = profile(gr.optimized)
            sizes = [l["frontier_size"] for l in prof_do] + [0]
            bu = [dict(l, nxt=sizes[i + 1]) for i, l in enumerate(prof_do) if l["bottom_up"]]
            td = [l for l in prof_do if not l["bottom_up"]]
            t_bu = sum(l["advance_ms"] for l in bu)
            t_td = sum(l["advance_ms"] for l in td)
            r_bu = roof(bu, lambda l: bu_bytes(l, l["nxt"]), "bfs_level_kernel (bottom-up launches)")
            r_bu["levels"] = [[l["frontier_size"], l["edges"], l["bu_open"], l["bu_probes"], round(l["advance_ms"], 4),
                               round(l["other_ms"], 4)] for l in bu]
            r_bu["share_of_step_kernel_time"] = round(t_bu / max(t_bu + t_td, 1e-9), 3)
            if bu:
                attach_traffic(r_bu, "bottom_up")
            roofline, roofline_other = (r_bu, roofline_td) if t_bu >= t_td or not td else (roofline_td, r_bu)
            roofline["all_levels"] = [[l["frontier_size"], l["edges"], int(l["bottom_up"]), round(l["advance_ms"], 4),
                                       round(l["other_ms"], 4)] for l in prof_do]
        # ---- CPU baseline: the oracle (port of the reference's PQ CPU path), bounded sample
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            g = O.Csr(csr.row_offsets, csr.column_indices, csr.nonzero_values)
            t_cpu, runs, ev_cpu, budget = 0.0, 0, 0, 12.0
            while t_cpu < budget * 1e3 and runs < 64:
                d_cpu, ms = O.bfs(g, src)
                t_cpu += ms
                runs += 1
                ev_cpu += edges_rank
            ok = bool(np.array_equal(d_cpu, dist_t.cpu().numpy()))
            cpu = {"value": round(ev_cpu / (t_cpu * 1e3), 2), "unit": "MTEPS", "cores": 1, "kind": "port",
                   "sample": "%d full BFS runs of the same workload/source with oracle/oracle.c orc_bfs "
                             "(priority-queue search of examples/algorithms/bfs/bfs_cpu.hxx), %.1f s"
                             % (runs, t_cpu / 1e3),
                   "matches_gpu": ok}
        out = {"metric": "MTEPS (million traversed edges/sec) BFS", "value": round(mteps, 1), "unit": "MTEPS",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "int32", "data": "synthetic",
               "config": {"workload": wl["name"], "n_vertices": V, "n_edges": E, "source": src,
                          "advance_load_balance": args.lb, "filter": "compact (fused into advance)",
                          "advance_direction": "forward" if args.topdown_only else "optimized",
                          "kernel_launch_groups_per_step": int(st["aux"]),
                          "parallelism": "single GPU",
                          "edges_visited_per_step": edges_rank, "search_depth": st["search_depth"],
                          "enact_ms_last": round(enact_ms, 4), "setup_s": round(t_setup, 1)},
               "roofline": roofline, "roofline_topdown_advance": roofline_other, "cpu_baseline": cpu}
        print(json.dumps(out))
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
