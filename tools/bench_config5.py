"""BASELINE configs[4] (developer tool; bench.py keeps the weak-scaling contract): the evaluation flow of
evaluate/coco_eval.py:245-283 over a 5000-entry image index (COCO val2017 is not available offline, so
the entries are synthetic 368x368 uint8 BGR images), sharded contiguously across the ranks
(parallel.shard_range), in batches of 32 per rank, ONE RCCL all_gather of the result records per batch.
STRONG scaling: total work is fixed.

  python tools/bench_config5.py                      # 1 GPU
  python tools/bench_config5.py --gpus 8             # re-executes itself through torch.distributed.run, 8 ranks
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
         --master-port 29511 tools/bench_config5.py --gpus 8

Every index entry is a distinct image: entry i's pixels are a hash of (i, pixel) evaluated on the device
(generating 5000 images with numpy would time numpy), so every rank sees exactly the shard it owns
whatever the world size.  Per batch, like preprocess.run_eval_batched: uint8 images -> ONE
rtpose_preprocess_u8_batch launch (resize / pad / normalise into the plan's input) -> forward ->
scene blend (random weights give junk maps, see bench.py) -> decode -> gather.  Rank 0 prints one JSON line."""
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


def synth_images(index, hw, dev):
    """uint8 [len(index), hw, hw, 3]: entry i = low bits of a multiplicative hash of (i, pixel), on the device."""
    p = torch.arange(hw * hw * 3, device=dev, dtype=torch.int64)[None, :]
    i = torch.as_tensor(index, device=dev, dtype=torch.int64)[:, None]
    v = ((p * 2654435761 + (i + 1) * 40503 * 65537) >> 11) & 255
    return v.to(torch.uint8).reshape(len(index), hw, hw, 3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=5000)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dtype", choices=("fp32", "bf16", "bf16x3"), default="fp32")
    ap.add_argument("--gpus", type=int, default=0, help="ranks (0 = whatever the launcher's WORLD_SIZE says, else 1)")
    args = ap.parse_args()
    par = importlib.import_module(PKG + ".parallel")
    par.require_devices(max(args.gpus, 1), "bench_config5.py")
    if args.gpus > 1 and not par.launched_by_torchrun():
        raise SystemExit(par.relaunch_under_torchrun(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    if args.gpus and par.env_world()[2] != args.gpus:
        raise SystemExit("bench_config5.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, par.env_world()[2]))
    pkg = importlib.import_module(PKG)
    synth = importlib.import_module(PKG + ".synth")
    dec = importlib.import_module(PKG + ".decode")
    pre = importlib.import_module(PKG + ".preprocess")
    capi = pkg._capi
    lib = capi.lib
    import ctypes as C
    rank, local_rank, world = par.init_from_env("nccl")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    model = pkg.get_model('vgg19')
    model.load_state_dict(synth.he_init_state_dict(model, seed=0))
    model = model.cuda().float().eval()
    model.set_compute_dtype(args.dtype)
    B, HW = args.batch, 368
    index = list(range(args.images))                       # the evaluation set: 5000 entries
    sched = par.batch_schedule(len(index), rank, world, B)   # every rank runs the same number of collectives
    # scenes blended over the net output (decoder input = scene + 1e-3 * maps): a pool of 16 batches, entry i
    # uses scene (i % pool size) - rank independent
    pool = []
    for s in range(16):
        heat, paf, _ = synth.make_batch(B, HW, HW, seed=2000 + s)
        pool.append((torch.from_numpy(heat).to(dev), torch.from_numpy(paf).to(dev)))
    plan = model.plan_for_shape(B, HW, HW, dev)
    pipeline = importlib.import_module(PKG + ".pipeline")
    side = pipeline.SideDecoder(None)       # decode + record D2H (+ gather) of batch k beside the forward of batch k + 1
    stream = capi.current_stream()
    post = (lambda blk: par.gather_records(blk, world)) if world > 1 else None

    def enqueue_batch(i0, n_valid, slot):
        idx = [index[min(i0 + k, len(index) - 1)] for k in range(B)] if n_valid else [index[0]] * B
        imgs = synth_images(idx, HW, dev)
        pre.preprocess_into_plan(plan, [imgs.data_ptr() + k * HW * HW * 3 for k in range(B)], [(HW, HW)] * B, HW, 0, stream)
        capi.check(lib.rtpose_net_set_keep_intermediates(plan.handle, 0))
        side.guarded(plan, lambda: capi.check(lib.rtpose_net_forward_prepared(plan.handle, stream),
                                              "rtpose_net_forward_prepared"))
        pbase, lpaf, _, h, w = model.output_view(plan, 0)
        hbase, lheat, _, _, _ = model.output_view(plan, 1)
        sh, sp = pool[(i0 // B) % len(pool)]
        capi.check(lib.rtpose_layout_axpby(hbase, C.byref(lheat), capi.ptr(sh), 19, B, h, w, 1e-3, 1.0, stream))
        capi.check(lib.rtpose_layout_axpby(pbase, C.byref(lpaf), capi.ptr(sp), 38, B, h, w, 1e-3, 1.0, stream))
        side.decode(slot, (hbase, lheat, pbase, lpaf, h, w), B, dev, 64, 64, post, plan=plan, model=model)

    def records(slot):
        bufs, host = side.wait(slot)
        return host.reshape(-1, bufs.words)

    enqueue_batch(0, B, 0)              # plan / weights / RCCL warm-up (untimed)
    records(0)
    humans = flags = 0

    def count(host):
        nonlocal humans, flags
        if rank == 0:
            humans += int(host[:, dec.RES_HEADER + 1].sum())
            flags |= int(np.bitwise_or.reduce(host[:, dec.RES_HEADER + 2]))

    torch.cuda.synchronize()
    par.barrier(dev)
    t0 = time.perf_counter()
    prev = None
    for j, (i0, n_valid) in enumerate(sched):      # the host reads batch j - 1's records after it has queued batch j
        enqueue_batch(i0, n_valid, j & 1)
        if prev is not None:
            count(records(prev))
        prev = j & 1
    if prev is not None:
        count(records(prev))
    torch.cuda.synchronize()
    par.barrier(dev)
    elapsed = par.max_over_ranks(time.perf_counter() - t0, dev)
    if rank == 0:
        print(json.dumps({"metric": "config 5: fixed 5000-image set, images/s (strong scaling)",
                          "value": round(args.images / elapsed, 2), "unit": "images/s", "n_gpus": world,
                          "ranks_seen": torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1,
                          "images": args.images, "batch_per_rank": B, "batches_per_rank": len(sched),
                          "seconds": round(elapsed, 3), "dtype": args.dtype, "scaling": "strong",
                          "index": "%d distinct synthetic uint8 images (device hash of (entry, pixel)), contiguous "
                                   "shards; image prep by one rtpose_preprocess_u8_batch launch per batch" % args.images,
                          "note": "the last batch of a shard is padded to the batch size (the padded images are "
                                  "computed and gathered, not counted)",
                          "pipeline": "decoder + record D2H (+ gather) of batch k on a second stream under the forward "
                                      "of batch k + 1", "humans_seen_rank0_gather": humans,
                          "overflow_flags": flags}))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
