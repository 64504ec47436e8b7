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


def _rank_main(rank, world, port, q, lbfgs):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
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
    assert tr.world == world and tr._comm is None  # gloo: the collectives go through torch.distributed
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
    q.put((rank, tr.params.cpu().numpy(), [terms[k] for k in ("loss0", "loss1", "loss2", "loss3", "kdist", "aug", "sim", "rica",
                                                              "total", "nonfinite")],
           [t.cpu().numpy() for t in tr.y]))
    dist.barrier()
    dist.destroy_process_group()


def _run_two_ranks(lbfgs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q, lbfgs)) for r in range(2)]
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
@pytest.mark.parametrize("lbfgs", [False, True], ids=["adam", "lbfgs"])
def test_two_ranks_on_one_gpu_equal_the_global_batch(lbfgs):
    """engine -> all-reduce -> optimiser on two ranks == one process on the global batch: replicated parameters
    identical across ranks (bitwise) and equal to the global-batch parameters to 2e-5; the logged terms are the
    global ones on every rank; each rank's multipliers are its slice of the global ones."""
    got = _run_two_ranks(lbfgs)
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
