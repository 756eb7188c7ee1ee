#!/usr/bin/env python3
"""One analysis over all GPUs of a node, one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/multi_gpu.py

Every rank generates, evaluates and reduces its own row shard; the accumulator record and the Gram
sums of the second order cross the GPUs in ONE RCCL all-reduce; every rank ends with the full result.
`--host-exchange --shared-device` rehearses the same flow on a single GPU (records cross the host).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import varsens_amd as vs  # noqa: E402
from varsens_amd import _native, dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=6)
    ap.add_argument("--n", type=int, default=1 << 16)
    ap.add_argument("--host-exchange", action="store_true")
    ap.add_argument("--shared-device", action="store_true")
    ap.add_argument("--out", default=None, help="rank 0 saves the result arrays here (.npz)")
    ap.add_argument("--rows-per-pass", type=int, default=0,
                    help="stream every rank's shard through HBM in row ranges of this many draws")
    args = ap.parse_args()

    group = dist.HostGroup()
    ctx = _native.Context(dist.device_for_rank(group.local_rank, args.shared_device))
    comm = dist.HostExchange(ctx, group) if args.host_exchange else dist.RcclComm(ctx, group)

    a = np.array([0, 0.5, 3, 9, 99, 99], dtype=float)[:args.k] if args.k <= 6 else np.linspace(0, 9, args.k)
    shard = dist.shard_rows(args.n, group.world, group.rank)
    # shard=: only this rank's slice of the seed-1 shuffle goes to its GPU
    sample = vs.Sample(args.k, args.n, vs.scale.Identity(), verbose=False, context=ctx, shard=shard)
    if args.rows_per_pass:
        v = vs.Varsens(vs.objectives.GFunction([a, a[::-1]]), sample=sample, verbose=False, comm=comm, shard=shard,
                       rows_per_pass=args.rows_per_pass)
    else:
        objective = vs.Objective(args.k, args.n, sample, vs.objectives.GFunction([a, a[::-1]]), verbose=False,
                                 shard=shard)
        v = vs.Varsens(objective, verbose=False, comm=comm)
    if group.rank == 0:
        print("ranks=%d rows/rank=%d sens[:,0]=%s" % (group.world, shard[1], np.array2string(v.sens[:, 0], precision=6)))
        if args.out:
            np.savez(args.out, E_2=v.E_2, var_y=v.var_y, sens=v.sens, sens_t=v.sens_t, sens_2=v.sens_2, sens_2n=v.sens_2n)
    comm.close()
    ctx.close()
    group.close()


if __name__ == "__main__":
    main()
