"""Multi-GPU layer: one process per GPU, images sharded, ONE collective.

Images are independent once the reference's process-global decoder state
(lib/pafprocess/pafprocess.cpp:12-13) is gone, so the path shards with no
data-path collective: rank r owns the contiguous image range
[r*N/G, (r+1)*N/G).  The only exchange is the gather of the fixed-capacity
result records (a few KB per image) — ``torch.distributed`` all_gather, which is
RCCL over xGMI on the GPU box (backend "nccl") and gloo in the CPU tests.  The
reference's own multi-GPU mechanism (nn.DataParallel, demo/picture_demo.py:47)
is not reused.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1 process = 1 GPU)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def launched_by_torchrun():
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ and "MASTER_ADDR" in os.environ


def init_from_env(backend=None, always=False):
    """Join the process group the torchrun environment describes.  A single process needs none; with
    ``always`` a world of ONE launched by torchrun still creates it, so that the RCCL library load, the
    device binding and the collective itself are exercised on a 1-GPU box (tests/test_runtime_gpu.py)."""
    rank, local_rank, world = env_world()
    if (world > 1 or (always and launched_by_torchrun())) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(n_items, rank, world):
    """Contiguous, balanced [lo, hi) of rank's items (first n_items % world ranks get one more)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_records(local, world=None, group=None, force=False):
    """all_gather of equally shaped int32 record blocks [n_local, words] -> [world*n_local, words]
    in rank order on every rank (one fused collective per batch, never per image).  A world of one returns its
    block as is, unless ``force`` asks for the collective anyway (needs an initialised process group)."""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (force and dist.is_initialized()):
        return local
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out


def max_over_ranks(value, device):
    """max of a python float over ranks (bench timing contract)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier(device=None):
    if dist.is_initialized() and dist.get_world_size() > 1:
        if device is not None and device.type == "cuda":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()


def batch_schedule(n_items, rank, world, batch):
    """Batches of a rank's contiguous shard for a run in which EVERY rank issues the same number of
    collectives (the gather is one all_gather per batch): list of (first_item, n_valid) of length
    ceil(ceil(n_items / world) / batch); trailing entries of a short shard have n_valid == 0 (the
    rank still computes a padded batch and takes part in the gather)."""
    lo, hi = shard_range(n_items, rank, world)
    per_rank = -(-n_items // world)
    n_batches = -(-per_rank // batch)
    out = []
    for b in range(n_batches):
        i0 = lo + b * batch
        out.append((i0, max(0, min(batch, hi - i0))))
    return out


def all_gather_floats(value, device):
    """[value of rank 0, ..., value of rank G-1] on every rank (per-rank timings of the bench line)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = torch.empty(dist.get_world_size(), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, t)
    return [float(v) for v in out.cpu()]


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def require_devices(n, who):
    """Fail fast - before any rank is spawned, and again in every rank - when this node shows fewer HIP devices than
    the run was asked to use: N ranks on fewer devices would either fault in hipSetDevice or, worse, share one GPU
    and be reported as N."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < int(n):
        raise SystemExit("%s: %d GPU(s) requested but this node shows %d HIP device(s) "
                         "(HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?)" % (who, int(n), have))
    return have


def relaunch_under_torchrun(script, argv, nproc, python=None):
    """Run ``script argv`` as ``nproc`` ranks of ONE node - ``python -m torch.distributed.run --nnodes=1
    --nproc-per-node nproc --master-addr 127.0.0.1 --master-port <free> script argv`` - with this process's
    stdout / stderr, and return its exit code.  A tool asked for N > 1 GPUs calls this when it was started plain, so
    that `--gpus N` can never silently measure one GPU (the reference's own multi-GPU mechanism, nn.DataParallel in
    demo/picture_demo.py:47, needs no launcher either)."""
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC only on these hosts (RCCL needs it)
    cmd = [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(nproc)),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), script] + list(argv)
    return subprocess.call(cmd, env=env)
