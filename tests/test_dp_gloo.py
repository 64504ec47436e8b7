"""World-size-2 rehearsal of the data-parallel path on CPU (gloo): the sharding / normalisation
rules of lshm_amd.dist + the two collectives reproduce the single-process global-batch loss and
gradients.  The per-rank arithmetic is the oracle here (the HIP engine needs a GPU; its own
world-scaling is covered by tests/test_gpu_step.py::test_world2_shares_sum_to_global)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from oracle import lshm_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, B, bpb, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from lshm_amd import dist as D
    r, w, _, pg = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    cfg = O.StepConfig(K=4, bpb=bpb, batch_size=B // bpb // world)
    params, M = O.make_params(cfg)
    x, uv = O.closed_form_inputs(B, 4)
    b0, b1 = D.shard_baselines(B // bpb, rank, world)
    xs, uvs = x[b0 * bpb:b1 * bpb], uv[b0 * bpb:b1 * bpb]
    n_local = xs.numel()
    y = [0.01 * O.closed_form((x.numel(),), f"y{k}", 1.0, 0.123 + 0.1 * k) for k in range(3)]
    ys = [t.view(B, -1)[b0 * bpb:b1 * bpb].reshape(-1) for t in y]
    leaves = O.flat_leaves(params, M)
    for t in leaves:
        t.requires_grad_(True)
    total, terms = O.closure_losses(params, M, xs, uvs, ys, cfg)
    grads = torch.autograd.grad(total, leaves)
    # this rank's share of the global means: equal shards => 1/world of every local mean
    flat = torch.cat([g.reshape(-1) for g in grads]) / world
    tv = torch.tensor([float(t) for t in terms] + [float(total)], dtype=torch.float64) / world
    D.allreduce_closure(flat, tv, pg)
    # centroid partial sums
    Z = 0.8 * O.closed_form((B, 256), "dp:Z", 1.0, 0.4142) + 0.3
    num, den = O.khm_offline_partials(Z[b0 * bpb:b1 * bpb], M.detach(), cfg.p)
    Mnew = D.allreduce_centroid_partials(num, den, pg)
    if rank == 0:
        q.put((flat.numpy(), tv.numpy(), Mnew.numpy(), n_local))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_global_batch():
    B, bpb, world = 8, 2, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, B, bpb, q)) for r in range(world)]
    for p in procs:
        p.start()
    flat, tv, Mnew, n_local = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process global batch
    cfg = O.StepConfig(K=4, bpb=bpb, batch_size=B // bpb)
    params, M = O.make_params(cfg)
    x, uv = O.closed_form_inputs(B, 4)
    assert n_local * world == x.numel()
    y = [0.01 * O.closed_form((x.numel(),), f"y{k}", 1.0, 0.123 + 0.1 * k) for k in range(3)]
    leaves = O.flat_leaves(params, M)
    for t in leaves:
        t.requires_grad_(True)
    total, terms = O.closure_losses(params, M, x, uv, y, cfg)
    grads = torch.autograd.grad(total, leaves)
    ref = torch.cat([g.reshape(-1) for g in grads])
    got = torch.from_numpy(flat)
    assert ((got - ref).norm() / ref.norm()).item() < 2e-5
    tref = [float(t) for t in terms] + [float(total)]
    for a, b in zip(tv, tref):
        assert abs(a - b) <= 2e-6 * abs(b) + 1e-9
    Z = 0.8 * O.closed_form((B, 256), "dp:Z", 1.0, 0.4142) + 0.3
    assert torch.allclose(torch.from_numpy(Mnew), O.khm_offline_update(Z, M.detach(), cfg.p), rtol=1e-10)


def test_shard_baselines():
    from lshm_amd.dist import shard_baselines
    assert [shard_baselines(32, r, 8) for r in (0, 7)] == [(0, 4), (28, 32)]
    with pytest.raises(ValueError):
        shard_baselines(10, 0, 4)
