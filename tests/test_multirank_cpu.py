"""world_size-2 `gloo` test of the N>1 path on CPU.

Graph-parallel sharding needs no data-path collective (graphs are independent); the only
exchanges are the optional gather of per-shard outputs to one rank (what DataParallel.gather does,
ogbg-code/tg/data_parallel.py:62) and the max-over-ranks timing reduction of bench.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dagnn_amd import collate_sharded, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _graph_row(g):
    return [float(g.num_nodes), float(g.edge_index.shape[1])]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        graphs = synth.code2_graphs(7, 24, 30)   # every rank sees the same loader batch ...
        shards = collate_sharded(graphs, world)  # ... and takes its contiguous, node-balanced shard
        first = sum(s.num_graphs for s in shards[:rank])
        mine = shards[rank]
        # stand-in for the per-shard forward: one output row per graph, a pure function of that graph
        rows = torch.tensor([_graph_row(g) for g in graphs[first:first + mine.num_graphs]])
        sizes = [torch.zeros(1, dtype=torch.long) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([rows.shape[0]]))
        sizes = [int(s) for s in sizes]
        m = max(sizes)
        pad = torch.zeros(m, 2)
        pad[:rows.shape[0]] = rows
        bufs = [torch.zeros(m, 2) for _ in range(world)]
        dist.all_gather(bufs, pad)               # ragged shards travel through padded buffers
        full = torch.cat([b[:s] for b, s in zip(bufs, sizes)], 0)
        assert torch.equal(full, torch.tensor([_graph_row(g) for g in graphs]))  # == single-device order
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # bench.py's timing reduction
        assert float(t) == float(world)
        dist.barrier()
        out[rank] = sum(sizes)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_sharding_and_gather_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: 24, 1: 24}


def _train_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dagnn_amd.train import GradBucket
        torch.manual_seed(0)  # identical replicas
        net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
        bucket = GradBucket(net.parameters())
        opt = torch.optim.Adam(bucket.params, lr=1e-2)
        g = torch.Generator().manual_seed(100 + rank)  # every rank its own shard
        x, y = torch.randn(8, 6, generator=g), torch.randint(0, 3, (8,), generator=g)
        for _ in range(3):
            bucket.zero()
            loss = torch.nn.functional.cross_entropy(net(x), y)
            loss.backward()
            local = bucket.flat.clone()
            bucket.all_reduce_mean()
            gathered = [torch.zeros_like(local) for _ in range(world)]
            dist.all_gather(gathered, local)
            assert torch.allclose(bucket.flat, sum(gathered) / world, atol=1e-7)
            assert all(p.grad.data_ptr() >= bucket.flat.data_ptr() for p in bucket.params)  # still views
            opt.step()
        w = torch.cat([p.detach().flatten() for p in net.parameters()])
        ws = [torch.zeros_like(w) for _ in range(world)]
        dist.all_gather(ws, w)
        assert torch.equal(ws[0], ws[1])  # replicas stay in lock-step
        out[rank] = float(loss)
    finally:
        dist.destroy_process_group()


def _uneven_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dagnn_amd.train import GradBucket
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
        g = torch.Generator().manual_seed(5)
        x, y = torch.randn(11, 6, generator=g), torch.randint(0, 3, (11,), generator=g)   # the global batch, everywhere
        cut = 4                                                                         # shards of 4 and 7 rows
        mine = slice(0, cut) if rank == 0 else slice(cut, 11)
        bucket = GradBucket(net.parameters())
        bucket.zero()
        torch.nn.functional.cross_entropy(net(x[mine]), y[mine]).backward()            # mean over the LOCAL shard
        bucket.all_reduce_mean(local_count=x[mine].shape[0])
        got = bucket.flat.clone()
        ref_net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
        ref_net.load_state_dict(net.state_dict())
        torch.nn.functional.cross_entropy(ref_net(x), y).backward()                    # mean over the GLOBAL batch
        ref = torch.cat([p.grad.flatten() for p in ref_net.parameters()])
        assert torch.allclose(got, ref, atol=1e-6), float((got - ref).abs().max())
        # the unweighted mean of the two local means is NOT the global mean here
        naive = GradBucket(ref_net.parameters())
        out[rank] = float((got - ref).abs().max())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_uneven_shards_give_the_global_mean_gradient_gloo():
    """Shards of 4 and 7 rows: the bucket's count-weighted all-reduce equals the single-process gradient of the mean
    loss over all 11 rows (the reference's loss, main_pyg.py:55-60); the node-balanced Collater split makes uneven
    shards the normal case (tg/dataloader.py:17-27)."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_uneven_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert len(out) == 2 and max(out.values()) < 1e-6


@pytest.mark.timeout(180)
def test_two_rank_gradient_bucket_all_reduce_gloo():
    """The training exchange of the N>1 path (one flat-bucket all-reduce, bench.py's training leg) on gloo."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_train_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert len(out) == 2


# ---------------------------------------------------------------- the reference's caller: DataParallel(list[Batch])
class _RowSum(torch.nn.Module):
    """Stand-in module: one output row per graph, differentiable in its parameter."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor([0.5, -0.25]))

    def forward(self, G):
        n = torch.bincount(G.batch, minlength=G.num_graphs).float()
        e = torch.bincount(G.batch[G.edge_index[0]], minlength=G.num_graphs).float()
        return torch.stack([n, e], dim=1) @ self.w


def test_data_parallel_shim_single_process_semantics():
    """`tg/data_parallel.py:41-50`: empty list -> warning and None; one device -> `module(data_list[0])` only;
    `.module` and the `module.`-prefixed state dict of the reference's checkpoints."""
    from dagnn_amd import DataParallel
    graphs = synth.code2_graphs(3, 6, 20)
    shards = collate_sharded(graphs, 2)
    dp = DataParallel(_RowSum(), device_ids=[])
    with pytest.warns(UserWarning):
        assert dp([]) is None
    assert torch.equal(dp(shards), dp.module(shards[0]))
    assert list(dp.state_dict().keys()) == ["module.w"]
    assert dp.src_device.type == "cpu"
    # k device ids in ONE process (`:52-62`: scatter, one replica per element, gather): every element runs, the
    # outputs come back concatenated in list order - nothing is silently dropped
    dpk = DataParallel(_RowSum(), device_ids=[0, 1])
    dpk.src_device = torch.device("cpu")
    full = dpk.module(synth.GraphBatch.from_data_list(graphs))
    assert torch.equal(dpk(shards), full)
    lists = DataParallel._gather([[torch.ones(2, 3), torch.zeros(2, 1)], [torch.ones(1, 3), torch.zeros(1, 1)]])
    assert [tuple(t.shape) for t in lists] == [(3, 3), (3, 1)]


class _Checked(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.ones(1))
        self.checks = 0

    def forward(self, G):
        return self.w * float(G.num_graphs)

    def check(self):
        self.checks += 1


def test_data_parallel_eval_forward_checks_the_module():
    """In eval mode the wrapper asks the module for a blocking device-side error check behind every forward
    (`main_pyg.py:91-124` consumes the outputs right away; the last batch of a loop has no next forward)."""
    from dagnn_amd import DataParallel
    shards = collate_sharded(synth.code2_graphs(3, 4, 12), 1)
    dp = DataParallel(_Checked(), device_ids=[])
    dp.train()
    dp(shards)
    assert dp.module.checks == 0
    dp.eval()
    dp(shards)
    dp(shards)
    assert dp.module.checks == 2


def _dp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dagnn_amd import DataParallel
        graphs = synth.code2_graphs(11, 9, 25)
        shards = collate_sharded(graphs, world)            # every rank builds the same list, as the reference's one loader
        dp = DataParallel(_RowSum(), device_ids=[])
        y = dp(shards)                                      # this rank's element
        assert y.shape[0] == shards[rank].num_graphs
        y.mean().backward()                                 # local mean loss
        dp.reduce_gradients(local_count=shards[rank].num_graphs)
        full = _RowSum()
        full(synth.GraphBatch.from_data_list(graphs)).mean().backward()   # mean over the global batch
        assert torch.allclose(dp.module.w.grad, full.w.grad, atol=1e-6)
        dp.zero_grad()
        assert float(dp.module.w.grad.abs().max()) == 0.0 and dp.module.w.grad.data_ptr() == dp._bucket.flat.data_ptr()
        # the reference's loop calls optimizer.zero_grad() (`main_pyg.py:50`), which sets the gradients to None: the
        # next backward() allocates them OUTSIDE the bucket.  reduce_gradients must notice, or step 2 reduces a stale
        # bucket and the replicas drift apart silently
        opt = torch.optim.SGD(dp.parameters(), lr=0.1)
        ref = _RowSum()
        ref_opt = torch.optim.SGD(ref.parameters(), lr=0.1)
        for _ in range(3):
            opt.zero_grad()        # set_to_none=True is the default
            assert dp.module.w.grad is None
            dp(shards).mean().backward()
            assert dp.module.w.grad.data_ptr() != dp._bucket.flat.data_ptr()
            dp.reduce_gradients(local_count=shards[rank].num_graphs)
            assert dp.module.w.grad.data_ptr() == dp._bucket.flat.data_ptr()
            opt.step()
            ref_opt.zero_grad()
            ref(synth.GraphBatch.from_data_list(graphs)).mean().backward()
            ref_opt.step()
            assert torch.allclose(dp.module.w.detach(), ref.w.detach(), atol=1e-6)
        with pytest.raises(ValueError):
            dp(shards + shards)
        assert dp(shards[:1]) is None if rank == 1 else dp(shards[:1]) is not None   # a list shorter than the world
        out[rank] = shards[rank].num_graphs
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_data_parallel_shim_gloo():
    """k processes, one `data_list` each: rank r runs element r, and `reduce_gradients` gives every rank the
    gradient of the mean loss over the global batch although the node-balanced shards differ in size."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dp_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert len(out) == 2 and sum(out.values()) == 9


# ---------------------------------------------------------------- two buckets: the heads' exchange leaves mid-backward
class _TwoPart(torch.nn.Module):
    """A 'core' whose output feeds two 'heads' (the shape of DAGNN + its vocabulary heads, dagnn.py:106-112)."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(5)
        self.core = torch.nn.Linear(2, 3)
        self.heads = torch.nn.ModuleList([torch.nn.Linear(3, 4), torch.nn.Linear(3, 4)])

    def forward(self, G):
        n = torch.bincount(G.batch, minlength=G.num_graphs).float()
        e = torch.bincount(G.batch[G.edge_index[0]], minlength=G.num_graphs).float()
        z = torch.tanh(self.core(torch.stack([n, e], dim=1) / 50.0))
        return [h(z) for h in self.heads]


def _overlap_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dagnn_amd.train import GradBucket, OverlappedGradReducer
        graphs = synth.code2_graphs(13, 11, 25)
        shards = collate_sharded(graphs, world)             # uneven, node-balanced shards
        mine = shards[rank]
        loss_of = lambda m, G: sum(p.pow(2).mean() for p in m(G)) / 2   # noqa: E731
        a, b, full = _TwoPart(), _TwoPart(), _TwoPart()
        red = OverlappedGradReducer(a.parameters(), early=a.heads.parameters())
        single = GradBucket(b.parameters())
        launched_mid_backward = []
        orig = red.early.launch
        red.early.launch = lambda *args, **kw: (launched_mid_backward.append(a.core.weight.grad.abs().sum().item() == 0.0),
                                                orig(*args, **kw))[1]
        for step in range(2):
            red.zero(local_count=mine.num_graphs)
            loss_of(a, mine).backward()
            red.finish()
            single.zero()
            loss_of(b, mine).backward()
            single.all_reduce_mean(local_count=mine.num_graphs)
            full.zero_grad()
            loss_of(full, synth.GraphBatch.from_data_list(graphs)).backward()
            for (k, pa), pb, pf in zip(a.named_parameters(), b.parameters(), full.parameters()):
                assert torch.equal(pa.grad, pb.grad), k                       # two buckets == one bucket, bit for bit
                assert torch.allclose(pa.grad, pf.grad, atol=1e-6), k         # == the single-process global-mean gradient
        # the heads' exchange left from the hook, before the core's gradients existed
        assert launched_mid_backward and all(launched_mid_backward)
        # the reference's step clips the GLOBAL gradient norm between backward() and step() (main_pyg.py:63-64, --clip 0.25):
        # behind both buckets' exchange, == torch's clip_grad_norm_ on the single-process global-mean gradient; both ranks
        # end with the same gradients, and a norm under the threshold leaves them alone
        for clip in (1e-3, 1e3):
            red.zero(local_count=mine.num_graphs)
            loss_of(a, mine).backward()
            norm = red.finish(clip=clip)
            full.zero_grad()
            loss_of(full, synth.GraphBatch.from_data_list(graphs)).backward()
            ref_norm = torch.nn.utils.clip_grad_norm_(full.parameters(), clip)
            assert abs(float(norm) - float(ref_norm)) <= 1e-5 * float(ref_norm)
            assert (float(ref_norm) > clip) == (clip < 1.0)
            for (k, pa), pf in zip(a.named_parameters(), full.parameters()):
                assert torch.allclose(pa.grad, pf.grad, rtol=1e-5, atol=1e-9), k
            flat = torch.cat([pa.grad.reshape(-1) for pa in a.parameters()])
            both = [torch.zeros_like(flat) for _ in range(world)]
            dist.all_gather(both, flat)
            assert torch.equal(both[0], both[1])
        out[rank] = mine.num_graphs
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_overlapped_two_bucket_reduce_equals_single_bucket_gloo():
    """`OverlappedGradReducer`: the heads' bucket is all-reduced asynchronously from a post-accumulate hook in the
    middle of `backward()`, the core's bucket afterwards; on uneven shards the result equals the single-bucket
    count-weighted exchange bit for bit and the single-process global-mean gradient to rounding."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_overlap_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert len(out) == 2 and sum(out.values()) == 11


def _reserve_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dagnn_amd import engine
        got = []
        assert engine.RESERVED_CUS == -1
        engine._COLLECTIVE.update(registered=False, group=None)
        os.environ.pop("NCCL_MAX_NCHANNELS", None)
        got.append(engine.reserved_cus_info(True)[0])          # default group, world 2, no pin: RCCL's upper bound
        os.environ["NCCL_MAX_NCHANNELS"] = "12"
        got.append(engine.reserved_cus_info(True)[0])          # pinned: 12 channels -> 16 CUs (whole CUs per XCD)
        assert "NCCL_MAX_NCHANNELS=12" in engine.reserved_cus_info(True)[1]
        got.append(engine.reserved_cus_info(False)[0])         # inference passes never reserve
        groups = [dist.new_group([r]) for r in range(world)]   # every rank creates every group
        engine.register_collective(groups[rank])               # the exchange runs on a one-rank group: nothing to leave room for
        got.append(engine.reserved_cus_info(True)[0])
        engine.register_collective(None)                       # the default group again
        got.append(engine.reserved_cus_info(True)[0])
        os.environ.pop("NCCL_MAX_NCHANNELS", None)
        engine._COLLECTIVE.update(registered=False, group=None)
        dist.barrier()
        out[rank] = got
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_reserved_cus_follow_the_exchange_group_and_the_channel_pin_gloo():
    """`engine.reserved_cus`: CUs a training pass leaves to the gradient collective = the channel count RCCL may use
    (`NCCL_MAX_NCHANNELS` when pinned, else its upper bound of 64), on the group the exchange really runs on
    (`engine.register_collective`, called by `OverlappedGradReducer`), never for an inference pass."""
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_reserve_worker, args=(world, port, out), nprocs=world, join=True)
        for r in range(world):
            assert list(out[r]) == [64, 16, 0, 0, 16]
