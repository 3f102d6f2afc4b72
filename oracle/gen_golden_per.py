"""Golden vectors for the prioritised replay (BaseClass/replay_buffer.py:57-223), made by EXECUTING the reference's
SumTree / ReplayTree from a scratch copy (oracle/ref_harness.py).  TEST INFRASTRUCTURE.

For each capacity (a power of two and two others: the flat-array SumTree visits its leaves in a ROTATED order when
the capacity is not a power of two) the script pushes entries with known errors, re-prioritises some with
batch_update, then samples: the uniform draws `random.uniform(a, b)` makes are recorded, so a restatement can be fed
the same numbers.  python oracle/gen_golden_per.py -> tests/golden/per.npz
"""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_harness import RefSession  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden")


def main():
    s = RefSession()
    try:
        import importlib
        rb = importlib.import_module("BaseClass.replay_buffer")
        out = {}
        rng = np.random.default_rng(20240917)
        for ci, (cap, n_push, batch) in enumerate([(64, 64, 16), (100, 137, 32), (1000, 700, 64), (5, 5, 4)]):
            tree = rb.ReplayTree(cap)
            errs = rng.uniform(0.0, 3.0, n_push)
            errs[rng.integers(0, n_push, max(1, n_push // 7))] = 0.0            # fresh transitions: error 0
            for k in range(n_push):
                tree.push(k, torch.tensor(errs[k]))                             # data = push counter
            prio_after_push = tree.tree.tree[-cap:].copy()
            # re-prioritise a few leaves (tree indices as sample() returns them)
            upd_data = rng.choice(min(n_push, cap), size=max(1, min(n_push, cap) // 5), replace=False)
            upd_tree_idx = upd_data + cap - 1
            upd_err = rng.uniform(0.0, 2.5, len(upd_data))
            tree.batch_update(list(upd_tree_idx), upd_err.copy())
            prio = tree.tree.tree[-cap:].copy()
            draws = []
            real_uniform = random.uniform

            def rec_uniform(a, b):
                v = real_uniform(a, b)
                draws.append(v)
                return v

            random.uniform = rec_uniform
            random.seed(100 + ci)
            rounds = []
            try:
                for r in range(3):
                    draws.clear()
                    b, idxs, w = tree.sample(batch)
                    rounds.append((np.array(draws), np.array(idxs), np.array([int(x) for x in b]), np.array(w), tree.beta))
            finally:
                random.uniform = real_uniform
            pre = f"c{ci}_"
            out[pre + "capacity"] = cap
            out[pre + "n_push"] = n_push
            out[pre + "batch"] = batch
            out[pre + "errors"] = errs
            out[pre + "prio_after_push"] = prio_after_push
            out[pre + "upd_data"] = upd_data
            out[pre + "upd_err"] = upd_err
            out[pre + "prio"] = prio
            out[pre + "n_entries"] = tree.tree.n_entries
            out[pre + "total_int"] = tree.tree.total()
            out[pre + "total"] = tree.tree.tree[0]
            for r, (d, ti, data, w, beta) in enumerate(rounds):
                out[pre + f"r{r}_draws"] = d
                out[pre + f"r{r}_tree_idx"] = ti
                out[pre + f"r{r}_data"] = data
                out[pre + f"r{r}_weights"] = w
                out[pre + f"r{r}_beta"] = beta
        out["alpha"], out["epsilon"], out["clip"], out["beta0"], out["beta_inc"] = 0.6, 0.01, 1.0, 0.4, 0.001
        np.savez_compressed(os.path.join(OUT, "per.npz"), **out)
        print("wrote per.npz:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in list(out.items())[:12]})
    finally:
        s.close()


if __name__ == "__main__":
    main()
