import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_distributed as T
import torch.multiprocessing as mp
if __name__ == "__main__":
    tmp = tempfile.mkdtemp()
    mp.spawn(T._worker, args=(2, T.free_port(), True, tmp), nprocs=2, join=True)
    per = [np.load(os.path.join(tmp, "r%d.npy" % r), allow_pickle=True)[0] for r in range(2)]
    for key in per[0]:
        out = []
        for r in range(2):
            part, lo, hi, e_r, depth, modes = per[r][key]
            out.append((int((part != 2**31 - 1).sum()), e_r, depth))
        print(key, out)
