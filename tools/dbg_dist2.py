"""Debug/driver: partitioned BFS with W ranks sharing cuda:0 (gloo carries the exchange).
    python tools/dbg_dist2.py [world]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch, torch.distributed as dist
    import gunrock_amd as gr
    from gunrock_amd import distributed as D
    import oracle_lib as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    V, E = 20000, 160000
    for kind, seed in (("rmat", 5), ("rmat_sym", 9)):
        props, full = gr.generate(kind, V, E, seed=seed)
        g = O.Csr(full.row_offsets, full.column_indices, full.nonzero_values)
        b = D.vertex_bounds(V, world); lo, hi = int(b[rank]), int(b[rank + 1])
        _, mine = gr.generate_rows(kind, V, E, lo, hi, seed=seed)
        mine_in = None
        if kind == "rmat":
            _, mine_in = gr.generate_rows(kind, V, E, lo, hi, seed=seed, in_rows=True)
        src = int(np.argmax(np.diff(full.row_offsets)))
        want, _, ev = O.bfs_queue(g, src)
        for overlap in (False, True):
            eng = D.GrxEngine(props, mine, rank, world, "cuda:0", int(full.number_of_nonzeros), in_rows=mine_in, overlap=overlap)
            d = torch.empty(V, dtype=torch.int32, device="cuda:0")
            for s_, optimized in ((src, True), (src, False), (0, True), (V - 1, True)):
                want, _, ev = O.bfs_queue(g, s_)
                st = D.bfs(eng, dist, s_, d, optimized=optimized)
                got = d.cpu().numpy()[lo:hi]
                bad = np.flatnonzero(got != want[lo:hi])
                print("rank", rank, kind, "src", s_, "overlap", overlap, "opt", optimized, "mismatch", len(bad), st, flush=True)
                if len(bad):
                    print("   reached", int((got != 2**31 - 1).sum()), "of", int((want[lo:hi] != 2**31 - 1).sum()), flush=True)
                    print("   bad", (bad[:6] + lo).tolist(), got[bad[:6]].tolist(), want[lo:hi][bad[:6]].tolist(), flush=True)
            del eng
    dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    mp.spawn(worker, args=(world, 29533), nprocs=world, join=True)
