"""BASELINE config 5 (developer tool; bench.py keeps the weak-scaling contract): a fixed set of
5000 synthetic 368x368 images (COCO val2017 is not available offline) sharded contiguously
across the ranks (parallel.shard_range), processed in batches of 32 per rank, ONE RCCL
all_gather of the result records per batch.  STRONG scaling: total work is fixed.

  python tools/bench_config5.py                      # 1 GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
         --master-port 29511 tools/bench_config5.py

Images are regenerated per batch from the image index (seeded), so every rank sees exactly the
shard it owns whatever the world size; rank 0 prints one JSON line."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "pytorch_realtime_multi-person_pose_estimation_amd"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=5000)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dtype", choices=("fp32", "bf16", "bf16x3"), default="fp32")
    args = ap.parse_args()
    pkg = importlib.import_module(PKG)
    par = importlib.import_module(PKG + ".parallel")
    synth = importlib.import_module(PKG + ".synth")
    dec = importlib.import_module(PKG + ".decode")
    pipeline = importlib.import_module(PKG + ".pipeline")
    rank, local_rank, world = par.init_from_env("nccl")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    model = pkg.get_model('vgg19')
    model.load_state_dict(synth.he_init_state_dict(model, seed=0))
    model = model.cuda().float().eval()
    model.set_compute_dtype(args.dtype)
    est = pipeline.PoseEstimator(model)
    lo, hi = par.shard_range(args.images, rank, world)
    B = args.batch
    # a small pool of distinct device batches, cycled (generating 5000 scenes on the host would
    # time numpy, not the path); the image index -> pool slot mapping is rank independent
    pool = []
    for s in range(4):
        g = torch.Generator().manual_seed(1000 + s)
        x = (torch.rand(B, 3, 368, 368, generator=g) - 0.5).to(dev)
        heat, paf, _ = synth.make_batch(B, 368, 368, seed=2000 + s)
        pool.append((x, (torch.from_numpy(heat).to(dev), torch.from_numpy(paf).to(dev))))
    est(pool[0][0], pool[0][1])       # capacity growth + plan/weights warm-up (untimed)
    sched = par.batch_schedule(args.images, rank, world, B)   # every rank runs the same number of collectives
    nb_max = len(sched)
    humans = 0
    torch.cuda.synchronize()
    par.barrier(dev)
    t0 = time.perf_counter()
    for b in range(nb_max):
        i0, n_valid = sched[b]
        x, scene = pool[((i0 // B) if n_valid else 0) % len(pool)]
        bufs = est.enqueue(x, scene)
        rec = bufs.result.view(bufs.n, bufs.words)
        if world > 1:
            host = par.gather_records(rec, world).cpu()
        else:
            host = dec.fetch(bufs)
        if rank == 0:
            humans += int(np.asarray(host).reshape(-1, bufs.words)[:, dec.RES_HEADER + 1].sum())
    torch.cuda.synchronize()
    par.barrier(dev)
    elapsed = par.max_over_ranks(time.perf_counter() - t0, dev)
    if rank == 0:
        print(json.dumps({"metric": "config 5: fixed 5000-image set, images/s (strong scaling)",
                          "value": round(args.images / elapsed, 2), "unit": "images/s", "n_gpus": world,
                          "images": args.images, "batch_per_rank": B, "batches_per_rank": nb_max,
                          "seconds": round(elapsed, 3), "dtype": args.dtype, "scaling": "strong",
                          "note": "the last batch of a shard is padded to the batch size (the padded images are "
                                  "computed and gathered, not counted)", "humans_seen_rank0_gather": humans}))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
