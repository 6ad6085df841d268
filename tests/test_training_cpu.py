"""Training step, CPU side: the oracle's autograd is pinned against the REFERENCE's (tests/golden/train_step.npz, written by
tests/golden/make_golden_train.py from the unmodified reference), and the host-side training logic (replay buffer, masks,
weight blob round trip, gradient all-reduce over gloo) is exercised without a GPU."""
import os
import socket

import numpy as np
import pytest
import torch

from helmnet_amd.engine import pack_weights, unpack_weights, weight_names, weight_shapes
from helmnet_amd.phantoms import ring_sos_batch
from oracle import helmnet_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def g_train():
    with np.load(os.path.join(GOLDEN, "train_step.npz")) as z:
        return {k: z[k] for k in z.files}


def _rel(a, b):
    a, b = torch.as_tensor(a).float(), torch.as_tensor(b).float()
    return float((a - b).abs().max() / b.abs().max())


@pytest.fixture(scope="module")
def oracle_run(weights, g_train):
    torch.set_num_threads(8)
    n, b = 96, 2
    sos = torch.from_numpy(ring_sos_batch(n, b, seed=21))
    k_sq = (1.0 / sos) ** 2
    src = O.point_source_map(n, [82, 48], 10.0).repeat(b, 1, 1, 1)
    t = O.SpectralTables(n, 8, 2, 1.0)
    w = {k: v.clone().requires_grad_(True) for k, v in weights.items()}
    wf, res, st = (torch.from_numpy(g_train[k]).clone().requires_grad_(True) for k in ("wf0", "res0", "st0"))
    loss, wfs, ress, sts = O.training_loss(wf, res, st, k_sq, src, w, t, 10)
    loss.backward()
    return {"loss": float(loss), "w": w, "wf": wf, "res": res, "st": st, "wfs": wfs, "ress": ress, "sts": sts,
            "ctx": (k_sq, src, t)}


def test_oracle_training_loss_and_gradients_match_the_reference(oracle_run, g_train):
    r = oracle_run
    assert abs(r["loss"] - float(g_train["loss"])) <= 1e-5 * float(g_train["loss"])
    assert _rel(r["wfs"][-1].detach(), g_train["wf_T"]) <= 1e-4
    assert _rel(r["ress"][-1].detach(), g_train["res_T"]) <= 1e-4
    assert _rel(r["sts"][-1].detach(), g_train["st_T"]) <= 1e-4
    want = unpack_weights(g_train["grad"], 4)
    errs = {k: _rel(r["w"][k].grad, want[k]) for k in want}
    errs.update(wf0=_rel(r["wf"].grad, g_train["grad_wf0"]), res0=_rel(r["res"].grad, g_train["grad_res0"]), st0=_rel(r["st"].grad, g_train["grad_st0"]))
    bad = {k: v for k, v in errs.items() if v > 1e-3}
    assert not bad, bad
    blob = torch.cat([r["w"][k].grad.reshape(-1) for k in weight_names(4)])
    assert _rel(blob, g_train["grad"]) <= 2e-4


def test_oracle_adam_matches_the_reference_optimiser(oracle_run, weights, g_train):
    """Three steps of clip_grad_value_ + Adam(0.9, 0.95) driven by the oracle's own gradients end where the reference ended."""
    k_sq, src, t = oracle_run["ctx"]
    lr, b1, b2, eps, wd, clip = (float(v) for v in g_train["adam_hparams"])
    names = weight_names(4)
    shapes = weight_shapes(4)
    flat = torch.from_numpy(pack_weights(weights))
    wf0, res0, st0 = (torch.from_numpy(g_train[k]) for k in ("wf0", "res0", "st0"))
    p = torch.nn.Parameter(flat.clone())
    opt = torch.optim.Adam([p], lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd)
    losses = []
    for _ in range(3):
        w, pos = {}, 0
        for k in names:
            cnt = int(np.prod(shapes[k]))
            w[k] = p[pos:pos + cnt].reshape(weights[k].shape)
            pos += cnt
        opt.zero_grad()
        loss, *_ = O.training_loss(wf0, res0, st0, k_sq, src, w, t, 10)
        loss.backward()
        torch.nn.utils.clip_grad_value_([p], clip)
        opt.step()
        losses.append(float(loss))
    assert np.allclose(losses, g_train["adam_losses"], rtol=2e-3)
    want = torch.from_numpy(g_train["adam_weights"])
    err = (p.detach() - want).abs()
    assert float(err.quantile(0.99)) <= 0.05 * lr and float(err.max()) <= 1.5 * lr


def test_weight_blob_round_trip_and_trainable_mask(weights):
    from helmnet_amd.training import trainable_mask
    blob = pack_weights(weights)
    back = unpack_weights(blob, 4)
    assert all(np.array_equal(back[k].reshape(weights[k].shape), weights[k].numpy()) for k in weights)
    assert list(weight_shapes(4)) == weight_names(4) and sum(int(np.prod(s)) for s in weight_shapes(4).values()) == 48160
    m = trainable_mask(4, "prelu")
    assert m.dtype == np.uint8 and m.size == 48160 and m.all()
    m2 = trainable_mask(4, "relu")
    assert m2.sum() == 48160 - 14          # inc + 4 x (conv_signal, conv_state) + 5 decoders: 14 slope slots
    pos = 0
    for name, shape in weight_shapes(4).items():
        cnt = int(np.prod(shape))
        assert (m2[pos:pos + cnt] == (0 if name.endswith("double_conv.1.weight") else 1)).all()
        pos += cnt


def test_replay_buffer_follows_the_reference_semantics():
    """replaybuffer.py:20-47: fixed capacity, append by index, sample without replacement, stacked fields, capacity clamp -- on the
    pre-allocated [capacity, ...] field tensors, with the batched write the training step uses; the indices drawn under a seed are the
    ones ``np.random.choice(capacity, batch, replace=False)`` gives the reference."""
    from helmnet_amd.training import Experience, ReplayBuffer
    rb = ReplayBuffer(6)
    assert len(rb) == 6 and rb.buffer == [None] * 6
    with pytest.raises(ValueError):
        np.random.seed(0); rb.sample(2)          # nothing written yet
    for i in range(6):
        rb.append(Experience(torch.full((2, 4, 4), float(i)), torch.full((2, 21), float(i)), torch.ones(1, 4, 4), torch.zeros(2, 4, 4),
                             torch.ones(2, 4, 4), 10 * i), i)
    np.random.seed(1)
    wf, h, k, r, s, its, idx = rb.sample(4)
    assert wf.shape == (4, 2, 4, 4) and h.shape == (4, 2, 21) and k.shape == (4, 1, 4, 4) and s.shape == (4, 2, 4, 4)
    assert len(set(idx.tolist())) == 4 and [int(wf[j, 0, 0, 0]) for j in range(4)] == idx.tolist() and list(its) == [10 * j for j in idx]
    np.random.seed(1)
    assert idx.tolist() == np.random.choice(6, 4, replace=False).tolist()      # the reference's draw
    assert rb.sample(99)[0].shape[0] == 6
    rb.append(Experience(torch.zeros(2, 4, 4), torch.zeros(2, 21), torch.zeros(1, 4, 4), torch.zeros(2, 4, 4), torch.zeros(2, 4, 4), 7), 2)
    assert rb.buffer[2].iteration == 7 and float(rb.buffer[2].wavefield.abs().max()) == 0.0
    # batched write (what training_step does with its advanced / fresh experiences): rows land in their slots, the others are untouched
    rb.write(np.array([5, 0]), torch.full((2, 2, 4, 4), 9.0), torch.full((2, 2, 21), 9.0), torch.ones(2, 1, 4, 4), torch.zeros(2, 2, 4, 4),
             torch.ones(2, 2, 4, 4), [31, 32])
    assert [b.iteration for b in rb.buffer] == [32, 10, 7, 30, 40, 31]
    assert float(rb.buffer[5].wavefield[0, 0, 0]) == 9.0 and float(rb.buffer[1].wavefield[0, 0, 0]) == 1.0
    rb.write(np.array([], dtype=np.int64), *(torch.zeros(0, 2, 4, 4), torch.zeros(0, 2, 21), torch.zeros(0, 1, 4, 4), torch.zeros(0, 2, 4, 4), torch.zeros(0, 2, 4, 4)), [])
    # partial update (the training step's write-back): the advanced fields of all sampled slots with the index tensor ``sample`` left behind,
    # then the rejected ones re-initialised (zero wavefield / state, new k_sq / residual / source); untouched fields and slots keep their contents
    np.random.seed(3)
    wf, h, k, r, s, its, idx = rb.sample(3)
    assert rb.last_sample_index.tolist() == idx.tolist()
    k_before = [rb.buffer[i].k_sq.clone() for i in range(6)]
    rb.update(idx, [101, 102, 103], index_device=rb.last_sample_index, wavefield=wf + 50, hidden_state=h + 50, residual=r + 50)
    for j, i in enumerate(idx):
        e = rb.buffer[i]
        assert e.iteration == 101 + j and torch.equal(e.wavefield, wf[j] + 50) and torch.equal(e.residual, r[j] + 50) and torch.equal(e.k_sq, k_before[i])
    others = [i for i in range(6) if i not in idx.tolist()]
    rb.update(idx[1:2], [0], zero=("wavefield", "hidden_state"), k_sq=torch.full((1, 1, 4, 4), 4.0), residual=torch.full((1, 2, 4, 4), -1.0), source=torch.ones(1, 2, 4, 4))
    e = rb.buffer[idx[1]]
    assert e.iteration == 0 and float(e.wavefield.abs().max()) == 0 and float(e.hidden_state.abs().max()) == 0 and float(e.k_sq.min()) == 4.0 and float(e.residual.max()) == -1.0
    assert all(torch.equal(rb.buffer[i].k_sq, k_before[i]) for i in others)
    rb2 = ReplayBuffer(2)
    rb2.append(Experience(torch.zeros(2, 4, 4), torch.zeros(2, 21), torch.zeros(1, 4, 4), torch.zeros(2, 4, 4), torch.zeros(2, 4, 4), 0), 0)
    with pytest.raises(ValueError):
        rb2.update([1], [5], wavefield=torch.zeros(1, 2, 4, 4))      # slot 1 was never written


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ddp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helmnet_amd.training import allreduce_gradients
        g = torch.arange(8, dtype=torch.float32) * (rank + 1)
        allreduce_gradients(g)
        q.put((rank, g.tolist()))
    finally:
        dist.destroy_process_group()


def _ddp_sync_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helmnet_amd.training import allreduce_gradients, allreduce_mean_scalar, broadcast_from_rank0
        torch.manual_seed(100 + rank)                      # every rank initialises differently, as freshly constructed modules do
        w, m, v = torch.randn(50), torch.rand(50), torch.rand(50)
        broadcast_from_rank0(w, m, v)
        holder = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
        sched = torch.optim.lr_scheduler.ReduceLROnPlateau(holder, mode="min", factor=0.5, patience=1, min_lr=1e-6)
        lrs = []
        for epoch in range(6):
            for step in range(2):                          # two data-parallel steps: rank-local gradients, averaged, plain SGD on the blob
                g = torch.randn(50) * (rank + 1)
                allreduce_gradients(g)
                w -= holder.param_groups[0]["lr"] * g
            # rank-local epoch losses that would trip the plateau detector in DIFFERENT epochs on the two ranks
            local = [1.0, 0.9, 0.95, 0.97, 0.5, 0.6][epoch] if rank == 0 else [1.0, 1.1, 1.2, 0.2, 0.3, 0.4][epoch]
            sched.step(allreduce_mean_scalar(local))
            lrs.append(holder.param_groups[0]["lr"])
        q.put((rank, w.tolist(), m.tolist(), lrs))
    finally:
        dist.destroy_process_group()


def test_data_parallel_replicas_stay_identical_gloo():
    """the replicas start from rank 0's weights / moments and step their learning-rate schedulers on the SAME (rank-mean) epoch
    loss -- with rank-local losses the two schedulers below would halve the rate in different epochs (world size 2, gloo on the CPU)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_ddp_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = {r[0]: r[1:] for r in (q.get(timeout=120) for _ in ps)}
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert got[0][0] == got[1][0] and got[0][1] == got[1][1]      # weights after 12 steps, broadcast moments: bit-identical
    assert got[0][2] == got[1][2] and got[0][2][-1] < 1e-3        # same learning-rate history, and the plateau logic did act
    from helmnet_amd.training import allreduce_mean_scalar, broadcast_from_rank0
    t = torch.ones(2)
    broadcast_from_rank0(t)                                       # no process group: identity
    assert allreduce_mean_scalar(0.25) == 0.25 and torch.equal(t, torch.ones(2))


def test_gradient_allreduce_averages_over_ranks_gloo():
    """The data-parallel step's only collective: the flat gradient averaged over the ranks (world size 2, gloo on CPU)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    want = [1.5 * i for i in range(8)]
    assert got[0] == want and got[1] == want
    # without a process group it is the identity
    from helmnet_amd.training import allreduce_gradients
    g = torch.ones(3)
    assert torch.equal(allreduce_gradients(g), torch.ones(3))
