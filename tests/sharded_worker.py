"""Rank body of test_fill_sharded_on_the_gpu's two-GPU case (started by torch.distributed.run): every rank
computes its shard on its own GPU, the result must be the oracle's on every rank."""
import importlib
import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    dist.init_process_group("gloo")  # control plane only: slab names, barrier
    pkg = importlib.import_module("distancetransform-depthcompletion_amd")
    from oracle import oracle as O

    rng = np.random.default_rng(77)
    x = np.where(rng.random((5, 96, 320)) < 0.05, rng.uniform(1, 80, (5, 96, 320)), 0).astype(np.float32)
    out = pkg.fill_sharded(x)
    ref = O.fill_batch(x)
    assert np.array_equal(out["depth"], ref[0]) and np.array_equal(out["dt"], ref[1]) and np.array_equal(out["index"], ref[2])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
