"""Things around the kernels that only a GPU box can prove:

* RCCL is reachable from this code at all: `torch.distributed.run --nproc-per-node 1` with backend "nccl" (= RCCL on
  ROCm), the all_gather_into_tensor of parallel.gather_records forced on the REAL int32 device record block the
  decoder wrote (library load, device binding, dtype / shape acceptance), and bench.py under torchrun against the
  plain run.  A 1-GPU lease cannot give a scaling curve: multi-GPU stays "unmeasured on hardware" (DESIGN.md §6).
* the torch-free C++ host (examples/c_host.cpp, built by __graft_entry__.build()) drives the whole path through the
  C ABI with hipMalloc'ed arenas - incl. the Winograd plan with its workspace-resident hand-over scratch."""
import json
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_NAME = "pytorch_realtime_multi-person_pose_estimation_amd"

_WORKER = r'''
import importlib, os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
PKG = "pytorch_realtime_multi-person_pose_estimation_amd"
pkg = importlib.import_module(PKG)
par = importlib.import_module(PKG + ".parallel")
dec = importlib.import_module(PKG + ".decode")
synth = importlib.import_module(PKG + ".synth")
rank, local_rank, world = par.init_from_env("nccl", always=True)
assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
dev = torch.device("cuda", local_rank)
heat, paf, _ = synth.make_batch(4, 184, 184, seed=5)
cfg = dec.make_cfg(None, 32, 64)
bufs = dec.DecodeBuffers(cfg, 4, dev)
hd, pd = torch.from_numpy(heat).to(dev), torch.from_numpy(paf).to(dev)
lay = pkg._capi.Layout
dec.decode_enqueue(pkg._capi.ptr(hd), lay.dense(19, 23, 23), pkg._capi.ptr(pd), lay.dense(38, 23, 23), 4, 23, 23, bufs)
local = bufs.result.view(4, bufs.words)                      # the int32 device record block, as bench.py gathers it
out = par.gather_records(local, world, force=True)           # RCCL all_gather_into_tensor of a world of one
assert out.data_ptr() != local.data_ptr() and out.shape == local.shape and out.dtype == torch.int32
torch.cuda.synchronize()
assert torch.equal(out, local)
t = torch.tensor([3.5], dtype=torch.float64, device=dev)
assert par.max_over_ranks(3.5, dev) == 3.5
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier(device_ids=[local_rank])
rec = dec.parse_image(out.cpu().numpy()[0])
assert rec["n_peaks"] > 0 and rec["parts"].shape[0] > 0
dist.destroy_process_group()
print("RCCL_OK peaks", rec["n_peaks"], "humans", rec["parts"].shape[0])
'''


def _torchrun(args, timeout):
    port = 29700 + (os.getpid() % 200)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                           "--master-addr", "127.0.0.1", "--master-port", str(port)] + args, cwd=ROOT, env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_rccl_all_gather_of_the_device_record_block(tmp_path, cuda):
    script = tmp_path / "rccl_worker.py"
    script.write_text(_WORKER)
    p = _torchrun([str(script), ROOT], 600)
    assert p.returncode == 0 and "RCCL_OK" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])


_TWO_RANKS_ONE_GPU = r"""
import importlib, os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
PKG = "pytorch_realtime_multi-person_pose_estimation_amd"
pkg = importlib.import_module(PKG)
par = importlib.import_module(PKG + ".parallel")
dec = importlib.import_module(PKG + ".decode")
synth = importlib.import_module(PKG + ".synth")
pipeline = importlib.import_module(PKG + ".pipeline")
rank, local_rank, world = par.init_from_env("gloo")          # NOT nccl: RCCL refuses two ranks on one device
assert world == 2 and dist.get_backend() == "gloo"
dev = torch.device("cuda", 0)                                # both ranks on THE one GPU of the box
torch.cuda.set_device(dev)
B, S, STEPS = 16, 368, 6

def build():
    m = pkg.get_model('vgg19')
    m.load_state_dict(synth.he_init_state_dict(m, seed=0))
    return m.cuda().float().eval()

def shard(r):
    g = torch.Generator().manual_seed(50 + r)
    x = (torch.rand(B, 3, S, S, generator=g) - 0.5).to(dev)
    h, p, _ = synth.make_batch(B, S, S, seed=200 + r)
    return x, (torch.from_numpy(h).to(dev), torch.from_numpy(p).to(dev))

model = build()
est = pipeline.PoseEstimator(model)
x, scene = shard(rank)
est(x, scene)                                                # capacities settle, weights packed
dist.barrier()
blocks = []
for _ in range(STEPS):                                       # both processes drive the device at the same time
    bufs = est.enqueue(x, scene)
    local = dec.fetch(bufs)                                  # records -> host (pinned D2H + stream sync)
    out = par.gather_records(torch.from_numpy(np.ascontiguousarray(local)), world)   # gloo all_gather of host blocks
    blocks.append(out.numpy().copy())
status = model.device_status(model.plan_for(x))
assert status == 0, "device error word %d with a second process on the GPU" % status
def content(block):       # what a record SAYS (the slack of its fixed-capacity tables is never written)
    out = []
    for r in block:
        d = dec.parse_image(r)
        out.append((d["peaks"].view(np.uint32).tobytes(), d["parts"].tobytes(), d["score"].view(np.uint32).tobytes(),
                    d["flags"]))
    return out
said = content(blocks[0])
for b in blocks[1:]:
    assert content(b) == said, "a step's records changed while the other process was running"
dist.barrier()
if rank == 0:       # alone on the GPU now: both shards serially through ONE process - must be the gathered bits
    ref = []
    for r in range(world):
        xr, sr = shard(r)
        ref.append(dec.fetch(est.enqueue(xr, sr)).copy())
    ref = np.concatenate(ref)
    assert ref.shape == blocks[0].shape and content(ref) == said, "records differ from the one-process run"
    people = sum(dec.parse_image(r)["parts"].shape[0] for r in ref)
    assert people > 2 * B
    print("TWO_RANKS_ONE_GPU_OK images", ref.shape[0], "people", people)
dist.barrier()
dist.destroy_process_group()
"""


def test_two_processes_share_the_one_gpu(tmp_path, cuda):
    """What a 1-GPU lease CAN show of the multi-process path: two ranks (torch.distributed.run --nproc-per-node 2,
    gloo - RCCL refuses two ranks on one device, so RCCL itself stays at world 1 until an 8-GPU node exists) both on
    cuda:0, each with its own plan, weight arena, decode buffers and persistent-kernel grids (the 7x7 launches hand
    split tiles from block to block through device flags: their bounded waits must survive a second process's
    kernels on the same CUs), stepping its own 16-image shard concurrently; the records gathered through the host
    equal, bit for bit, the two shards run serially in one process, and the device error word stays 0.  This is NOT a
    scaling measurement: multi-GPU throughput stays unmeasured on hardware."""
    script = tmp_path / "two_ranks.py"
    script.write_text(_TWO_RANKS_ONE_GPU)
    port = 29400 + (os.getpid() % 200)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "two_ranks_one_gpu.txt"), "w") as f:
            f.write("rc %d\n--- stdout ---\n%s\n--- stderr ---\n%s\n" % (p.returncode, p.stdout[-4000:], p.stderr[-12000:]))
    except OSError:
        pass
    assert p.returncode == 0 and "TWO_RANKS_ONE_GPU_OK" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])


def test_memory_that_is_not_of_the_current_device_is_refused(cuda):
    """Every entry point that takes device memory asks the runtime which device owns it (hipPointerGetAttributes, once
    per new pointer) and refuses host memory - and, on a multi-GPU node, memory of another device than the current one
    (a DataParallel replica, a mis-set LOCAL_RANK) - instead of launching on it.  A 1-GPU box can show the host-memory
    half; the other-device half is the same comparison with another owner."""
    import ctypes as C
    import importlib
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module(PKG_NAME)
    capi = pkg._capi
    lib = capi.lib
    dec = importlib.import_module(PKG_NAME + ".decode")
    synth = importlib.import_module(PKG_NAME + ".synth")
    s = capi.current_stream()
    # plan arenas in pinned host memory
    net = C.c_void_p()
    capi.check(lib.rtpose_net_create(1, 64, 64, C.byref(net)))
    ws_b, wt_b = lib.rtpose_net_workspace_bytes(net), lib.rtpose_net_weight_bytes(net)
    host_ws = torch.empty(ws_b + 256, dtype=torch.uint8).pin_memory()
    dev_ws = torch.empty(ws_b + 256, dtype=torch.uint8, device="cuda")
    dev_wt = torch.zeros(wt_b + 256, dtype=torch.uint8, device="cuda")
    al = lambda t: (t.data_ptr() + 255) // 256 * 256
    rc = lib.rtpose_net_bind(net, al(host_ws), ws_b, al(dev_wt), wt_b, 1, s)
    assert rc != 0 and b"not device memory" in lib.rtpose_last_error(), lib.rtpose_last_error()
    capi.check(lib.rtpose_net_bind(net, al(dev_ws), ws_b, al(dev_wt), wt_b, 1, s))
    # the input of a forward in host memory
    x_host = torch.zeros(1, 3, 64, 64).pin_memory()
    rc = lib.rtpose_net_forward(net, x_host.data_ptr(), s)
    assert rc != 0 and b"input tensor" in lib.rtpose_last_error(), lib.rtpose_last_error()
    x_dev = torch.zeros(1, 3, 64, 64, device="cuda")
    # (weights never loaded: the forward only has to be ACCEPTED for launch - finalize reads the estimates of zeros)
    assert lib.rtpose_net_forward(net, x_dev.data_ptr(), s) == 0, lib.rtpose_last_error()
    torch.cuda.synchronize()
    lib.rtpose_net_destroy(net)
    # decoder: maps in host memory
    heat, paf, _ = synth.make_batch(1, 184, 184, seed=5)
    cfg = dec.make_cfg(None, 32, 64)
    bufs = dec.DecodeBuffers(cfg, 1, torch.device("cuda", 0))
    hh, ph = torch.from_numpy(heat).pin_memory(), torch.from_numpy(paf).pin_memory()
    lay = capi.Layout
    with pytest.raises(capi.RtposeError, match="heat-map tensor"):
        dec.decode_enqueue(hh.data_ptr(), lay.dense(19, 23, 23), ph.data_ptr(), lay.dense(38, 23, 23), 1, 23, 23, bufs)
    hd, pd = hh.cuda(), ph.cuda()
    dec.decode_enqueue(capi.ptr(hd), lay.dense(19, 23, 23), capi.ptr(pd), lay.dense(38, 23, 23), 1, 23, 23, bufs)
    assert dec.parse_image(dec.fetch(bufs)[0])["n_peaks"] > 0


_MASKED = r"""
import ctypes as C, importlib, sys
import numpy as np
import torch
sys.path.insert(0, sys.argv[1])
PKG = "pytorch_realtime_multi-person_pose_estimation_amd"
pkg = importlib.import_module(PKG)
synth = importlib.import_module(PKG + ".synth")
lib = pkg._capi.lib
m = pkg.get_model('vgg19')
m.load_state_dict(synth.he_init_state_dict(m, seed=0))
m = m.cuda().float().eval()
x = (torch.rand(12, 3, 368, 368, generator=torch.Generator().manual_seed(5)) - 0.5).cuda()
out = {}
with torch.no_grad():
    for persist in (1, 0):
        plan = m.plan_for(x)
        pkg._capi.check(lib.rtpose_net_set_persistent7(plan.handle, persist))
        assert lib.rtpose_net_persistent7(plan.handle) == persist
        (paf, heat), _ = m(x)
        torch.cuda.synchronize()
        status = m.device_status(plan)
        out[persist] = (paf.cpu().numpy(), heat.cpu().numpy(), status)
assert out[0][2] == 0, "error word %d with one block per tile" % out[0][2]
np.savez(sys.argv[2], paf1=out[1][0], heat1=out[1][1], status1=out[1][2], paf0=out[0][0], heat0=out[0][1])
print("MASKED_OK cus", torch.cuda.get_device_properties(0).multi_processor_count, "status", out[1][2])
"""


def test_seven_by_seven_launches_without_split_tiles_and_under_a_cu_mask(tmp_path, cuda):
    """rtpose_net_set_persistent7(plan, 0): every 7x7 launch runs one block per tile instead of persistent blocks that hand
    split tiles over through device flags - for CU-masked / shared devices, where the dispatch order and residency the
    hand-over assumes are not given.  Both forms give the same bits (the sums run in the order of an unsplit tile), on
    the whole device and - in a child process - under HSA_CU_MASK (64 of the 256 CUs): there the split-tile form either
    works (error word 0: same bits again) or reports its timed-out hand-over in the error word, never silently wrong;
    the one-block-per-tile form must always be clean.  12 x 368 x 368: 2 x 12 x 12 strips of 46 x 46 = 288 tiles per
    grouped launch on 256 CUs - not whole rounds, so the default plan does split tiles."""
    script = tmp_path / "masked.py"
    script.write_text(_MASKED)
    res = {}
    for tag, extra in (("full", {}), ("masked", {"HSA_CU_MASK": "0:0-63"})):
        env = dict(os.environ, **extra)
        out = tmp_path / ("%s.npz" % tag)
        p = subprocess.run([sys.executable, str(script), ROOT, str(out)], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=900)
        assert p.returncode == 0 and "MASKED_OK" in p.stdout, (tag, p.stdout[-2000:], p.stderr[-3000:])
        import numpy as np
        res[tag] = dict(np.load(str(out)))
    import numpy as np
    full, masked = res["full"], res["masked"]
    assert int(full["status1"]) == 0
    assert np.array_equal(full["paf1"], full["paf0"]) and np.array_equal(full["heat1"], full["heat0"])
    assert np.array_equal(masked["paf0"], full["paf0"]) and np.array_equal(masked["heat0"], full["heat0"])
    if int(masked["status1"]) == 0:
        assert np.array_equal(masked["paf1"], full["paf0"]) and np.array_equal(masked["heat1"], full["heat0"])


def test_decoder_of_one_batch_under_the_forward_of_the_next(cuda):
    """PoseEstimator.submit / collect: the decoder and the record D2H of batch k run on a second stream while the forward of
    batch k + 1 is already on the compute stream; that forward waits for the decoder's last read of the maps only where
    it first writes their buffer (rtpose_net_set_output_guard; fp32 plans - a bf16 plan waits in front of its whole launch
    list since round 5, see the soak test below).  Three different batches through one 16-image plan,
    interleaved and repeated, two tickets in flight: every collected record block SAYS what the serial path
    (enqueue + fetch, one stream) says for the same batch - peaks, people, float scores, bit for bit."""
    import importlib
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module(PKG_NAME)
    dec = importlib.import_module(PKG_NAME + ".decode")
    synth = importlib.import_module(PKG_NAME + ".synth")
    pipeline = importlib.import_module(PKG_NAME + ".pipeline")
    dev = torch.device("cuda", 0)
    m = pkg.get_model('vgg19')
    m.load_state_dict(synth.he_init_state_dict(m, seed=0))
    m = m.cuda().float().eval()
    B, S = 16, 368

    def batch(r):
        g = torch.Generator().manual_seed(300 + r)
        h, p, _ = synth.make_batch(B, S, S, seed=400 + r)
        return (torch.rand(B, 3, S, S, generator=g) - 0.5).to(dev), (torch.from_numpy(h).to(dev), torch.from_numpy(p).to(dev))

    def content(block):
        out = []
        for r in block:
            d = dec.parse_image(r)
            out.append((d["peaks"].view(np.uint32).tobytes(), d["parts"].tobytes(), d["score"].view(np.uint32).tobytes(), d["flags"]))
        return out
    data = [batch(r) for r in range(3)]
    est = pipeline.PoseEstimator(m)
    want = []
    for x, scene in data:
        est(x, scene)                                            # capacities settle
        want.append(content(dec.fetch(est.enqueue(x, scene)).copy()))
    assert sum(len(dec.parse_image(r)["parts"]) for r in dec.fetch(est.enqueue(*data[0]))) > B
    order = [0, 1, 2, 2, 1, 0, 0, 1, 2, 1]
    prev, got = None, []
    for r in order:
        t = est.submit(*data[r])
        if prev is not None:
            got.append(content(est.collect(prev[0])[1].reshape(B, -1)))
        prev = (t, r)
    got.append(content(est.collect(prev[0])[1].reshape(B, -1)))
    torch.cuda.synchronize()
    for k, r in enumerate(order):
        assert got[k] == want[r], "step %d (batch %d): the overlapped path's records differ from the serial path's" % (k, r)
    # the same under bf16 (the decoder's share of a step is larger there, the forward 2.6x shorter)
    m.set_compute_dtype('bf16')
    try:
        est(data[0][0], data[0][1])
        want_b = [content(dec.fetch(est.enqueue(x, scene)).copy()) for x, scene in data]
        prev, got = None, []
        for r in order:
            t = est.submit(*data[r])
            if prev is not None:
                got.append(content(est.collect(prev[0])[1].reshape(B, -1)))
            prev = (t, r)
        got.append(content(est.collect(prev[0])[1].reshape(B, -1)))
        for k, r in enumerate(order):
            assert got[k] == want_b[r], "bf16 step %d (batch %d)" % (k, r)
    finally:
        m.set_compute_dtype('fp32')


def test_decoder_beside_the_next_forward_gives_the_serial_records_every_time(cuda, monkeypatch):
    """Soak of the overlapped pipeline in its default, most exposed form - the forward of batch k + 1 waits for the decoder
    of batch k only where it first rewrites the maps' buffer, so the decoder runs BESIDE the next forward's first launches - 600 bf16 and 200 fp32 steps of 16 images that carry 8 people each (49-64 candidate pairs per limb: the high
    lanes of the scoring wave are busy in every image), every record block compared with the serial path's.
    Round 5 found one limb score a sample off in ~1 % of such bf16 batches; round 6 traced it to the packed-fp32 VALU
    instructions clang's SLP vectoriser had put into the scoring loop (wrong values in lanes 48..63 when the wave shares a CU
    with the bf16 plan's kernels: 342 of 48,000 decodes on the box where the same source built without them gave 0 of
    240,000 - DESIGN.md 3.3, profiles/r06_decoder_beside_forward.txt).  The decoder is now built without them
    (tests/test_capi_cpu.py checks the object); this test keeps watch."""
    import importlib
    import numpy as np
    import torch
    monkeypatch.delenv("RTPOSE_GUARD_WHOLE_FORWARD", raising=False)   # the library default is the exposed form; the variables
    monkeypatch.setenv("RTPOSE_GUARD_FINE", "1")                      # are read when a plan's guard position is first decided
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module(PKG_NAME)
    dec = importlib.import_module(PKG_NAME + ".decode")
    synth = importlib.import_module(PKG_NAME + ".synth")
    pipeline = importlib.import_module(PKG_NAME + ".pipeline")
    dev = torch.device("cuda", 0)
    m = pkg.get_model('vgg19')
    m.load_state_dict(synth.he_init_state_dict(m, seed=0))
    m = m.cuda().float().eval()
    B, S = 16, 368
    data = []
    for r in range(3):
        g = torch.Generator().manual_seed(300 + r)
        rng = np.random.default_rng(400 + r)
        hp = [synth.render(synth.random_people(rng, 8, S, S, drop_prob=0.02), S, S, noise=0.02, rng=rng) for _ in range(B)]
        data.append(((torch.rand(B, 3, S, S, generator=g) - 0.5).to(dev),
                     (torch.from_numpy(np.stack([h for h, _ in hp])).to(dev), torch.from_numpy(np.stack([p for _, p in hp])).to(dev))))
    order = [0, 1, 2, 2, 1, 0, 0, 1, 2, 1]
    est = pipeline.PoseEstimator(m)
    try:
        for dt, steps in (('bf16', 600), ('fp32', 200)):
            m.set_compute_dtype(dt)
            for x, scene in data:
                est(x, scene)
            want = [dec.fetch(est.enqueue(x, scene)).copy() for x, scene in data]
            masks = [dec.result_mask(w) for w in want]
            assert np.mean([w[:, dec.RES_PART_COUNT:dec.RES_PART_COUNT + 18].mean() for w in want]) >= 6.5
            bad = []
            prev = None
            for k in range(steps + 1):
                r = order[k % len(order)] if k < steps else None
                t = est.submit(*data[r]) if r is not None else None
                if prev is not None:
                    got = est.collect(prev[0])[1].reshape(B, -1)
                    w, mk = want[prev[1]], masks[prev[1]]
                    if got.shape != w.shape or not np.array_equal(got[mk], w[mk]):
                        bad.append((dt, k - 1, [b for b in range(B) if not np.array_equal(got[b][mk[b]], w[b][mk[b]])]))
                prev = (t, r)
            assert not bad, bad
    finally:
        m.set_compute_dtype('fp32')


def _bench_line(p):
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.returncode == 0 and lines, (p.stdout[-2000:], p.stderr[-3000:])
    return json.loads(lines[-1])


def test_bench_under_torchrun_matches_the_plain_run(cuda):
    """`bench.py --gpus 1` launched the way the driver launches N > 1 (torch.distributed.run, RCCL process group,
    the record gather as a collective) prints the metric of the plain run: same workload, same value within 5 %
    (the collective of a world of one costs a few microseconds per step; measured ratio printed)."""
    flags = ["--gpus", "1", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-traffic"]
    plain = subprocess.run([sys.executable, "bench.py"] + flags, cwd=ROOT, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=900)
    a = _bench_line(plain)
    b = _bench_line(_torchrun(["bench.py"] + flags, 900))
    assert a["metric"] == b["metric"] and a["n_gpus"] == b["n_gpus"] == 1
    assert "RCCL" in b["config"]["parallelism"] and "RCCL" not in a["config"]["parallelism"]
    ratio = b["value"] / a["value"]
    print("bench.py under torchrun / plain: %.1f / %.1f img/s = %.3f" % (b["value"], a["value"], ratio))
    # the lower bound is the claim (the launcher and the one-rank collective cost nothing); the upper one only catches a
    # broken plain run - the first process on a cold box has measured 6 % slower than the second
    assert 0.95 <= ratio <= 1.10


def test_c_host_runs_the_whole_path_without_python(cuda, tmp_path):
    """examples/c_host: rtpose_net_create_opts -> hipMalloc arenas -> bind -> load 92 convs -> finalize -> forward
    -> scene blend -> decode -> D2H, for the default fp32 plan (guarded per-layer Winograd forms) at a batch whose 7x7
    launches run persistent blocks with split tiles (11 images), for the direct-kernel plan and for the bf16 plan; exit
    code 0, a sane rate, people decoded, no table overflow and a clean device status."""
    import importlib
    import numpy as np
    exe = os.path.join(ROOT, "examples", "c_host")
    if not os.path.exists(exe):
        pytest.skip("examples/c_host not built (python -c 'import __graft_entry__ as g; g.build()')")
    sys.path.insert(0, ROOT)
    synth = importlib.import_module(PKG_NAME + ".synth")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, PKG_NAME, "lib") + ":" + env.get("LD_LIBRARY_PATH", "")
    for args in (["11", "0", "default"], ["8", "0", "direct"], ["8", "1", "default"]):
        n = int(args[0])
        heat_s, paf_s, _ = synth.make_batch(n, 368, 368, seed=100)       # the decoder input of bench.py: scene + 1e-3 maps
        scene = tmp_path / ("scene%d.bin" % n)
        with open(scene, "wb") as f:
            f.write(np.ascontiguousarray(heat_s, np.float32).tobytes())
            f.write(np.ascontiguousarray(paf_s, np.float32).tobytes())
        p = subprocess.run([exe] + args + [str(scene)], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-3000:]
        m = re.search(r"([0-9.]+) images/s", p.stdout)
        assert m and float(m.group(1)) > 50.0, p.stdout[-2000:]
        assert "device status 0" in p.stdout and "overflow flags 0" in p.stdout, p.stdout[-2000:]
        h = re.search(r"(\d+) humans", p.stdout)
        assert h and int(h.group(1)) >= n, p.stdout[-2000:]
        print(p.stdout.strip().splitlines()[-2])


def test_c_host_decodes_people_and_its_records_equal_the_python_paths(cuda, tmp_path):
    """The C path of demo/picture_demo.py:57-61 (get_outputs -> paf_to_pose_cpp over lib/pafprocess/pafprocess.h:53-59)
    exercised for CONTENT: examples/c_host generates hash-seeded He-scale weights and images, blends the scene this test
    wrote over the maps and decodes; the Python path (get_model + PoseEstimator) on the same weights, images and
    scene must give the same records - people > 0, equal digest, equal raw words wherever a record defines them."""
    import importlib
    import numpy as np
    import torch
    exe = os.path.join(ROOT, "examples", "c_host")
    if not os.path.exists(exe):
        pytest.skip("examples/c_host not built (python -c 'import __graft_entry__ as g; g.build()')")
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module(PKG_NAME)
    synth = importlib.import_module(PKG_NAME + ".synth")
    pipeline = importlib.import_module(PKG_NAME + ".pipeline")
    dec = importlib.import_module(PKG_NAME + ".decode")
    n = 4
    heat_s, paf_s, _ = synth.make_batch(n, 368, 368, seed=100)
    scene, dump = tmp_path / "scene.bin", tmp_path / "records.bin"
    with open(scene, "wb") as f:
        f.write(np.ascontiguousarray(heat_s, np.float32).tobytes())
        f.write(np.ascontiguousarray(paf_s, np.float32).tobytes())
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, PKG_NAME, "lib") + ":" + env.get("LD_LIBRARY_PATH", "")
    p = subprocess.run([exe, str(n), "0", "default", str(scene), str(dump)], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:]
    m = re.search(r"(\d+) peaks, (\d+) humans, overflow flags (\d+)", p.stdout)
    d = re.search(r"record digest ([0-9a-f]{16})", p.stdout)
    assert m and d, p.stdout[-2000:]
    peaks, humans, flags = (int(v) for v in m.groups())
    assert peaks >= 18 * n // 2 and humans >= n and flags == 0, p.stdout[-2000:]
    assert "device status 0" in p.stdout

    model = pkg.get_model('vgg19')
    model.load_state_dict(synth.hashed_state_dict(model))
    model = model.cuda().float().eval()
    est = pipeline.PoseEstimator(model, max_peaks_per_part=256, max_humans=256)
    x = torch.from_numpy(synth.hashed_input(n)).to(cuda)
    bufs = est.enqueue(x, (torch.from_numpy(heat_s).to(cuda), torch.from_numpy(paf_s).to(cuda)), 1e-3)
    recs = dec.fetch(bufs).copy()
    got = np.fromfile(dump, dtype=np.int32).reshape(n, -1)
    assert got.shape == recs.shape
    assert int(recs[:, 0].sum()) == peaks and int(recs[:, 1].sum()) == humans
    assert "%016x" % synth.record_digest(recs) == d.group(1) == "%016x" % synth.record_digest(got)
    for i in range(n):
        a, b = dec.parse_image(recs[i]), dec.parse_image(got[i])
        assert np.array_equal(a["peaks"], b["peaks"]) and np.array_equal(a["parts"], b["parts"])
        assert np.array_equal(a["score"], b["score"]) and len(a["parts"]) >= 1
    print(p.stdout.strip().splitlines()[-3])
