"""world_size-2 CPU (gloo) test of the data-parallel path: flat gradient bucket + one all-reduce."""
import importlib
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ddp = importlib.import_module("3dvlp_amd.ddp")
    torch.manual_seed(rank)  # replicas start different on purpose
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    unused = torch.nn.Linear(2, 2)  # never used in forward: must still be reduced (as zeros)
    model.add_module("unused", unused)
    ddp.broadcast_parameters(model)
    bucket = ddp.FlatGradBucket(model)
    lo, hi = ddp.shard_range(8, rank, world)
    torch.manual_seed(123)
    x = torch.randn(8, 6)
    bucket.zero()
    model[2](model[1](model[0](x[lo:hi]))).pow(2).mean().backward()
    bucket.collect()
    bucket.all_reduce()
    # single-process reference: gradient of the mean over BOTH shards with the same (broadcast) weights
    import copy
    full = copy.deepcopy(model)
    for p in full.parameters():
        p.grad = None
    full[2](full[1](full[0](x))).pow(2).mean().backward()
    ref = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in full.parameters()])
    assert model.unused.weight.grad is None  # no gradient produced: the optimiser skips it, like the reference's
    n_unused = sum(p.numel() for p in model.unused.parameters())
    ret[rank] = (model[0].weight.detach().clone(), bucket.flat.clone(), bucket.flat[-n_unused:].clone(), ref)
    dist.barrier()
    dist.destroy_process_group()


def test_flat_bucket_allreduce_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    (w0, g0, u0, r0), (w1, g1, u1, r1) = ret[0], ret[1]
    assert torch.equal(w0, w1)          # parameter broadcast
    assert torch.allclose(g0, g1)       # identical averaged gradients on both ranks
    assert (u0 == 0).all() and (u1 == 0).all()
    assert torch.allclose(g0, r0, atol=1e-6) and torch.allclose(g1, r1, atol=1e-6)  # == full-batch gradient


def test_shard_range_covers_batch():
    ddp = importlib.import_module("3dvlp_amd.ddp")
    spans = [ddp.shard_range(64, r, 8) for r in range(8)]
    assert spans[0] == (0, 8) and spans[-1] == (56, 64)
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    import pytest
    with pytest.raises(ValueError):
        ddp.shard_range(10, 0, 4)


def _worker8(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    ddp = importlib.import_module("3dvlp_amd.ddp")
    torch.manual_seed(100 + rank)
    model = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.BatchNorm1d(8), torch.nn.ReLU(), torch.nn.Linear(8, 3))
    with torch.no_grad():
        model[1].running_mean.fill_(float(rank))      # buffers differ per rank before the broadcast
        model[1].num_batches_tracked.fill_(rank)
    ddp.broadcast_parameters(model)                     # two packed broadcasts (fp32 tensors, the int64 counter), not one per tensor
    bucket = ddp.FlatGradBucket(model)
    lo, hi = ddp.shard_range(32, rank, world)
    torch.manual_seed(7)
    x = torch.randn(32, 6)
    bucket.zero()
    model(x[lo:hi]).pow(2).mean().backward()
    bucket.collect()
    local = bucket.flat.clone()
    bucket.all_reduce()
    ret[rank] = (lo, hi, torch.cat([p.detach().reshape(-1) for p in model.parameters()]), float(model[1].running_mean[0]) if False else 0.0,
                 int(model[1].num_batches_tracked), local, bucket.flat.clone())
    dist.barrier()
    dist.destroy_process_group()


def test_flat_bucket_allreduce_world8():
    """Eight ranks on the CPU (gloo): contiguous scene shards that tile the global batch, ONE packed parameter broadcast
    (parameters and buffers of every rank equal rank 0's afterwards), and the averaged flat gradient equal on all ranks and
    equal to the mean of the ranks' local gradients."""
    world = 8
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker8, args=(world, _free_port(), ret), nprocs=world, join=True)
    spans = [(ret[r][0], ret[r][1]) for r in range(world)]
    assert spans == [(4 * r, 4 * r + 4) for r in range(world)]
    for r in range(1, world):
        assert torch.equal(ret[r][2], ret[0][2])                 # parameters after the broadcast
        assert ret[r][4] == 1                                    # rank 0's counter (0) + this step's forward
        assert torch.allclose(ret[r][6], ret[0][6])             # identical averaged gradients everywhere
    mean_local = torch.stack([ret[r][5] for r in range(world)]).mean(0)
    assert torch.allclose(ret[0][6], mean_local, atol=1e-6)


def _worker_buckets(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    ddp = importlib.import_module("3dvlp_amd.ddp")
    torch.manual_seed(3)
    model = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.ReLU(), torch.nn.Linear(8, 8), torch.nn.ReLU(), torch.nn.Linear(8, 3))
    bucket = ddp.FlatGradBucket(model)
    lo, hi = ddp.shard_range(4 * world, rank, world)
    torch.manual_seed(11)
    x = torch.randn(4 * world, 6)
    tail = list(model[0].parameters())                              # the layers whose backward finishes LAST (SA1 / SA2 there)
    head = [p for p in model.parameters() if all(p is not q for q in tail)]
    rt, rh = bucket.param_range(tail), bucket.param_range(head)
    assert rt == (0, 6 * 8 + 8) and rh == (rt[1], bucket.flat.numel())
    assert bucket.param_range([tail[0], head[-1]]) is None          # not one contiguous run

    def grads():
        bucket.zero()
        model(x[lo:hi]).pow(2).mean().backward()
        bucket.collect()
    grads()
    bucket.all_reduce()
    one = bucket.flat.clone()
    grads()
    pending = bucket.all_reduce_range(*rh)      # issued while the "tail" is still being produced in the real step
    bucket.all_reduce_range(*rt).wait()
    pending.wait()
    two = bucket.flat.clone()
    ret[rank] = (one, two)
    dist.barrier()
    dist.destroy_process_group()


def _two_bucket_case(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_buckets, args=(world, _free_port(), ret), nprocs=world, join=True)
    for r in range(world):
        one, two = ret[r]
        # the two overlapped pieces give the single all-reduce's averages (a ring all-reduce adds the ranks' terms in an order
        # that depends on where an element sits in its message: equal up to the order of eight float additions)
        assert torch.allclose(one, two, rtol=1e-6, atol=1e-9)
        assert torch.equal(two, ret[0][1]) and torch.equal(one, ret[0][0])   # and every rank holds the same buffer


def test_two_bucket_overlapped_allreduce_equals_single_world2():
    """the head parameters' slice of the flat gradient buffer is reduced while the tail's gradients are still
    being computed, the tail's slice afterwards — same averaged buffer as ONE all-reduce, on every rank."""
    _two_bucket_case(2)


def test_two_bucket_overlapped_allreduce_equals_single_world8():
    _two_bucket_case(8)


def test_collect_subsets_equal_one_collect():
    """FlatGradBucket.collect_subset over two disjoint parameter sets (the step driver's split backward: each stream copies the
    gradients IT completed) leaves the same flat buffer, .grad views and touched flags as one collect()."""
    ddp = importlib.import_module("3dvlp_amd.ddp")
    torch.manual_seed(0)

    def make():
        torch.manual_seed(0)
        m = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
        m.add_module("unused", torch.nn.Linear(2, 2))
        return m

    x = torch.randn(8, 6)
    res = []
    for split in (False, True):
        model = make()
        bucket = ddp.FlatGradBucket(model)
        bucket.zero()
        model[2](model[1](model[0](x))).pow(2).mean().backward()
        if split:
            head = list(model[2].parameters()) + list(model.unused.parameters())
            tail = [p for p in model.parameters() if all(p is not q for q in head)]
            bucket.collect_subset(head)
            bucket.collect_subset(tail)
        else:
            bucket.collect()
        views_ok = all(p.grad is None or p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views))
        res.append((bucket.flat.clone(), list(bucket.touched), views_ok))
    assert torch.equal(res[0][0], res[1][0])
    assert res[0][1] == res[1][1] and res[0][1].count(False) == 2  # the unused layer's weight and bias
    assert res[0][2] and res[1][2]
