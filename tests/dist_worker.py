#!/usr/bin/env python3
"""One rank of kmerdb_amd.distributed.parsefile_distributed, for the multi-process tests (a fresh process per rank).

    python tests/dist_worker.py RANK WORLD PORT PATH K RWN CANON BLOCK_BYTES OUT_DIR [BACKEND] [ENGINE_OPTS_JSON]

All ranks use device 0 (one-GPU box), so the collectives run on gloo; on a multi-GPU node the same function runs on
nccl (= RCCL) with one device per rank.  Rank 0 saves counts.npy and its metadata; every rank writes rankN.json."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    path, k, rwn, canon, block, out_dir = sys.argv[4], int(sys.argv[5]), sys.argv[6] == "1", sys.argv[7] == "1", int(sys.argv[8]), sys.argv[9]
    backend = sys.argv[10] if len(sys.argv) > 10 else "gloo"
    opts = json.loads(sys.argv[11]) if len(sys.argv) > 11 else {}
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import numpy as np
    import torch
    import torch.distributed as dist
    from kmerdb_amd import distributed
    device = 0 if backend == "gloo" else rank
    torch.cuda.set_device(device)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    res = {"rank": rank}
    try:
        counts, meta, nullomers = distributed.parsefile_distributed(path, k, replace_with_none=rwn, canonicalize=canon, device=device,
                                                                    block_bytes=block, engine_opts=opts)
        if rank == 0:
            if k <= 13:
                np.save(os.path.join(out_dir, "counts.npy"), counts)
            else:
                nz = np.flatnonzero(counts)
                np.savez(os.path.join(out_dir, "counts_sparse.npz"), ids=nz.astype(np.uint64), cnt=counts[nz])
            res["meta"] = meta
            res["nullomers"] = int(len(nullomers))
        else:
            res["none"] = counts is None and meta is None and nullomers is None
    except Exception as e:  # noqa: BLE001 - reported to the parent
        res["error"] = type(e).__name__ + ": " + str(e)[:300]
    json.dump(res, open(os.path.join(out_dir, f"rank{rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
