"""Data-parallel path with the ENGINE in the loop: two fresh processes share the one GPU of the
test box, each runs KHarmonicTrainer(process_group=...) on its half of the baselines, gloo carries the
collectives (RCCL refuses two ranks on one device), and after two Adam iterations the replicated
parameters must equal a single-process run over the global batch.  Plus the RCCL communicator of the C ABI
at world size 1 (symbols, stream enqueue, engine attachment)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from oracle import lshm_oracle as O
from tests.util import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
B, K, BPB, STEPS = 8, 4, 2, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs():
    ocfg = O.StepConfig(K=K, bpb=BPB, batch_size=B // BPB)
    params, M = O.make_params(ocfg)
    x, uv = O.closed_form_inputs(B, 4)
    return params, M, x, uv


FAKE_RCCL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_rccl", "libfake_rccl.so")


def _rank_main(rank, world, port, q, lbfgs, engine_comm=False, fail_rank=None):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if engine_comm:
        # the engine-attached collective path (lshm_engine_set_comm), RCCL's entry points bound from the
        # host-shared-memory stand-in: RCCL itself refuses two ranks on one device
        os.environ.update(LSHM_DP_ENGINE="1", LSHM_RCCL_LIB=FAKE_RCCL)
        if fail_rank == rank:
            os.environ["LSHM_RCCL_LIB"] = "/nonexistent/librccl.so"  # this rank cannot build a communicator
    import torch.distributed as dist
    from lshm_amd import KHarmonicTrainer, TrainConfig
    from lshm_amd import dist as D
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    params, M, x, uv = _inputs()
    b0, b1 = D.shard_baselines(B // BPB, rank, world)
    sl = slice(b0 * BPB, b1 * BPB)
    tr = KHarmonicTrainer(TrainConfig(Kc=K), batch=(b1 - b0) * BPB, batch_per_bline=BPB, default_batch=b1 - b0,
                          device="cuda:0", process_group=dist.group.WORLD)
    if engine_comm and fail_rank is None:
        assert tr._comm is not None and tr.lib.lshm_engine_comm_early_bucket(tr._h) == 1
    else:  # the default (and the all-ranks fallback when one rank has no communicator): torch.distributed after the closure
        assert tr.world == world and tr._comm is None
    tr.load_state_dicts(params["net"], params["netT"], params["netF"], {"M": M})
    tr.new_minibatch(x[sl].cuda(), uv[sl].cuda())
    if lbfgs:
        opt = tr.make_lbfgs()
        tr.step_lbfgs(opt)
    else:
        for _ in range(STEPS):
            tr.step()
    torch.cuda.synchronize()
    terms = tr.read_terms()
    if engine_comm and fail_rank is None and not lbfgs:
        # the last closure sent the netT / netF gradients as the early bucket
        tr.closure_only()
        assert tr.lib.lshm_engine_last_flags(tr._h) & 1
        torch.cuda.synchronize()
    q.put((rank, tr.params.cpu().numpy(), [terms[k] for k in ("loss0", "loss1", "loss2", "loss3", "kdist", "aug", "sim", "rica",
                                                              "total", "nonfinite")],
           [t.cpu().numpy() for t in tr.y]))
    dist.barrier()
    dist.destroy_process_group()


def _run_two_ranks(lbfgs, engine_comm=False, fail_rank=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q, lbfgs, engine_comm, fail_rank)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=540) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return got


def _single_process(lbfgs):
    from lshm_amd import KHarmonicTrainer, TrainConfig
    params, M, x, uv = _inputs()
    tr = KHarmonicTrainer(TrainConfig(Kc=K), batch=B, batch_per_bline=BPB, default_batch=B // BPB, device=DEV)
    tr.load_state_dicts(params["net"], params["netT"], params["netF"], {"M": M})
    tr.new_minibatch(x.to(DEV), uv.to(DEV))
    if lbfgs:
        tr.step_lbfgs(tr.make_lbfgs())
    else:
        for _ in range(STEPS):
            tr.step()
    torch.cuda.synchronize()
    return tr


@pytest.mark.timeout(900)
@pytest.mark.parametrize("lbfgs,path", [(False, "torch"), (True, "torch"), (False, "engine"), (True, "engine"),
                                        (False, "engine-one-rank-fails")],  # (the fallback decision is made at construction: one optimiser suffices)
                         ids=["adam-torch", "lbfgs-torch", "adam-engine", "lbfgs-engine", "adam-engine-one-rank-fails"])
def test_two_ranks_on_one_gpu_equal_the_global_batch(lbfgs, path):
    """engine -> all-reduce -> optimiser on two ranks == one process on the global batch: replicated parameters
    identical across ranks (bitwise) and equal to the global-batch parameters to 2e-5; the logged terms are the
    global ones on every rank; each rank's multipliers are its slice of the global ones.
    path "torch": the collectives go through torch.distributed after the closure (gloo here, RCCL on a node).
    path "engine": through lshm_engine_set_comm -- the early netT / netF bucket on its own stream beside the 2-D
    backward, the closing group, the loss terms of the gradient-free closures -- with the RCCL entry points
    bound from tests/fake_rccl (host shared memory).  path "engine-one-rank-fails": rank 1 cannot build its
    communicator; the ranks must agree to fall back together instead of issuing mismatched collectives."""
    if path != "torch" and not os.path.exists(FAKE_RCCL):
        pytest.fail("tests/fake_rccl/libfake_rccl.so is not built (make testlibs)")
    got = _run_two_ranks(lbfgs, engine_comm=path != "torch", fail_rank=1 if path == "engine-one-rank-fails" else None)
    ref = _single_process(lbfgs)
    p0, p1 = torch.from_numpy(got[0][1]), torch.from_numpy(got[1][1])
    assert torch.equal(p0, p1)
    assert rel_err(p0, ref.params.cpu()) < 2e-5
    t = ref.read_terms()
    want = [t[k] for k in ("loss0", "loss1", "loss2", "loss3", "kdist", "aug", "sim", "rica", "total")]
    for r in range(2):
        for a, b in zip(got[r][2][:9], want):
            assert abs(a - b) <= 2e-5 * abs(b) + 1e-9
        assert got[r][2][9] == 0.0
    n = ref.x.numel()
    for k in range(3):
        yk = ref.y[k].cpu().view(B, -1)
        for r in range(2):
            mine = torch.from_numpy(got[r][3][k]).view(B // 2, -1)
            assert rel_err(mine, yk[r * B // 2:(r + 1) * B // 2]) < 2e-4, (k, r)
    assert n == 2 * got[0][3][0].size


def test_rccl_communicator_world_one():
    """lshm_comm_* binds RCCL at run time; at world size 1 an all-reduce is the identity, on the caller's stream,
    for the float arena and the double tail in one group."""
    from lshm_amd import _lib
    from lshm_amd.dist import Communicator
    lib = _lib.load()
    if not lib.lshm_comm_available():
        pytest.skip("no RCCL in this process")
    c = Communicator(None, torch.device("cuda:0"))
    assert (c.rank, c.world) == (0, 1) and lib.lshm_comm_world(c.handle) == 1
    a = torch.randn(100003, device=DEV)
    t = torch.randn(10, device=DEV, dtype=torch.float64)
    a0, t0 = a.clone(), t.clone()
    c.allreduce_flat(a, t)
    c.allreduce_flat(None, t)
    torch.cuda.synchronize()
    assert torch.equal(a, a0) and torch.equal(t, t0)
    c.close()


def test_engine_with_attached_communicator_world_one():
    """lshm_engine_set_comm: the closure all-reduces inside the call (early bucket for netT / netF on its own
    stream, the rest after the last weight gradient).  At world size 1 the trajectory must be bit-identical to
    the engine without a communicator -- the extra streams and events may not disturb any dependency."""
    from lshm_amd import KHarmonicTrainer, TrainConfig, _lib
    from lshm_amd.dist import Communicator
    lib = _lib.load()
    if not lib.lshm_comm_available():
        pytest.skip("no RCCL in this process")
    params, M, x, uv = _inputs()
    outs = []
    for attach in (False, True):
        tr = KHarmonicTrainer(TrainConfig(Kc=K), batch=B, batch_per_bline=BPB, default_batch=B // BPB, device=DEV)
        comm = None
        if attach:
            comm = Communicator(None, torch.device("cuda:0"))
            _lib.check(lib.lshm_engine_set_comm(tr._h, comm.handle))
        tr.load_state_dicts(params["net"], params["netT"], params["netF"], {"M": M})
        tr.new_minibatch(x.to(DEV), uv.to(DEV))
        for _ in range(3):
            tr.step()
        tr.lbfgs_closure()
        with torch.no_grad():
            tr.lbfgs_closure()
        torch.cuda.synchronize()
        outs.append((tr.params.clone(), tr.grads.clone(), tr.terms[:10].clone()))
        if attach:
            _lib.check(lib.lshm_engine_set_comm(tr._h, None))
            comm.close()
        del tr
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    # a communicator of another world size is refused
    tr = KHarmonicTrainer(TrainConfig(Kc=K), batch=B, batch_per_bline=BPB, default_batch=B // BPB, device=DEV)
    tr._sc.world = 2
    import ctypes as C
    h = C.c_void_p()
    _lib.check(lib.lshm_engine_create(C.byref(tr._sc), C.byref(h)))
    comm = Communicator(None, torch.device("cuda:0"))
    assert lib.lshm_engine_set_comm(h, comm.handle) != 0
    lib.lshm_engine_destroy(h)
    comm.close()


def test_captured_closure_with_communicator_sends_no_early_bucket():
    """The early bucket forks a stream off the forked weight-gradient stream; ending a capture of that topology
    crashes hipStreamEndCapture (ROCm 7.2, seen in round 2 with the same nesting).  A captured call therefore
    keeps every collective on the capturing stream: eager closures report the early bucket, a captured one does
    not, and its replay gives the eager gradients."""
    from lshm_amd import KHarmonicTrainer, TrainConfig, _lib
    from lshm_amd.dist import Communicator
    lib = _lib.load()
    if not lib.lshm_comm_available():
        pytest.skip("no RCCL in this process")
    params, M, x, uv = _inputs()
    tr = KHarmonicTrainer(TrainConfig(Kc=K), batch=B, batch_per_bline=BPB, default_batch=B // BPB, device=DEV)
    comm = Communicator(None, torch.device("cuda:0"))
    _lib.check(lib.lshm_engine_set_comm(tr._h, comm.handle))
    assert lib.lshm_engine_comm_early_bucket(tr._h) == 1
    tr.load_state_dicts(params["net"], params["netT"], params["netF"], {"M": M})
    tr.new_minibatch(x.to(DEV), uv.to(DEV))
    tr.closure_only()
    assert lib.lshm_engine_last_flags(tr._h) & _lib.ENGINE_USED_EARLY_BUCKET
    torch.cuda.synchronize()
    eager = tr.grads.clone()
    # the agreed switch turns it off for eager calls too
    _lib.check(lib.lshm_engine_set_early_bucket(tr._h, 0))
    tr.closure_only()
    assert not (lib.lshm_engine_last_flags(tr._h) & _lib.ENGINE_USED_EARLY_BUCKET)
    torch.cuda.synchronize()
    assert torch.equal(tr.grads, eager)
    _lib.check(lib.lshm_engine_set_early_bucket(tr._h, 1))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        tr.closure_only()
    assert not (lib.lshm_engine_last_flags(tr._h) & _lib.ENGINE_USED_EARLY_BUCKET)
    tr.grads.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(tr.grads, eager)
    _lib.check(lib.lshm_engine_set_comm(tr._h, None))
    comm.close()


@pytest.mark.timeout(900)
def test_bench_line_at_world_two_carries_what_the_collective_measured():
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank; here both ranks on the one GPU
    over gloo): the line's `dp` object is measured THROUGH the process group -- ranks seen by an all-reduce of 1, the gathered
    devices, the data-parallel path taken, one all-reduce of the gradient arena -- not read from the environment."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LSHM_SHARE_GPU0="1", LSHM_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "16",
           "--no-cpu-baseline", "--no-roofline", "--no-reuse-mode", "--no-lbfgs", "--no-rica", "--no-extra-modes"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=800, cwd=root, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 32
    d = j["dp"]
    assert d["ranks_seen"] == 2 and len(d["devices"]) == 2 and sorted(x["rank"] for x in d["devices"]) == [0, 1]
    assert all(x["shared_with_other_ranks"] for x in d["devices"])
    assert d["path"] in ("torch", "engine") and d["backend"] == "gloo"
    assert d["allreduce_us"] > 0 and d["allreduce_bytes"] == 4 * 1725716
