"""Golden index streams of the reference's samplers — runs ONLY in the authoring container (needs /root/reference).
The three sampler files are loaded by path (the package's __init__ pulls torchvision in, which is absent); their
outputs for fixed seeds / group layouts are stored as data in tests/golden/reference_samplers.json."""
import importlib.util
import json
import os
import sys

import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/maskrcnn_benchmark/data/samplers"


def load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    dist_mod, grp_mod, it_mod = load("distributed"), load("grouped_batch_sampler"), load("iteration_based_batch_sampler")
    out = {"distributed": [], "grouped": [], "iteration": []}
    for n, world, epoch in [(10, 1, 0), (23, 4, 7), (100, 8, 12345), (5, 2, 3)]:
        per_rank = []
        for rank in range(world):
            s = dist_mod.DistributedSampler(list(range(n)), num_replicas=world, rank=rank, shuffle=True)
            s.set_epoch(epoch)
            per_rank.append(list(s))
        out["distributed"].append({"n": n, "world": world, "epoch": epoch, "indices": per_rank})
    g = torch.Generator().manual_seed(0)
    for n, bs, drop in [(17, 2, False), (40, 3, True), (9, 4, False)]:
        groups = torch.randint(0, 2, (n,), generator=g).tolist()
        base = dist_mod.DistributedSampler(list(range(n)), num_replicas=1, rank=0, shuffle=True)
        base.set_epoch(5)
        b = grp_mod.GroupedBatchSampler(base, groups, bs, drop_uneven=drop)
        out["grouped"].append({"n": n, "batch_size": bs, "drop_uneven": drop, "groups": groups, "epoch": 5,
                               "batches": [list(x) for x in b]})
    for n, bs, iters, start in [(7, 2, 11, 0), (12, 4, 9, 3)]:
        base = dist_mod.DistributedSampler(list(range(n)), num_replicas=1, rank=0, shuffle=True)
        bsamp = torch.utils.data.sampler.BatchSampler(base, bs, drop_last=False)
        it = it_mod.IterationBasedBatchSampler(bsamp, iters, start)
        out["iteration"].append({"n": n, "batch_size": bs, "num_iterations": iters, "start_iter": start,
                                 "batches": [list(x) for x in it]})
    with open(os.path.join(HERE, "reference_samplers.json"), "w") as f:
        json.dump(out, f)
    print({k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
