"""Two ranks on one GPU (gloo rendezvous; RCCL refuses two ranks on one device): the real DAGNN training step on
the reference's node-balanced shards of one batch (`tg/dataloader.py:17-27` -> uneven graph counts) reduces to the
single-process gradient of the mean loss over the global batch (`main_pyg.py:55-60`)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dagnn_amd import collate_sharded, synth
        from dagnn_amd.train import GradBucket
        from tests import helpers as Hh
        dev = torch.device("cuda:0")
        meta = dict(H=64, n_attr=300, V=24, S=3, w_seed=41,
                    ctor=dict(w_edge_attr=True, num_layers=2, bidirectional=True, agg="attn_h", out_wx=False,
                              out_pool_all=False, out_pool="max", dropout=0.0))
        graphs = synth.code2_graphs(17, 14, 45)
        for g in graphs:
            g.x[:, 1] %= 300
        shards = collate_sharded(graphs, world)
        sizes = [s.num_graphs for s in shards]
        assert len(shards) == world and sizes[0] != sizes[1]      # node-balanced, hence uneven in graphs
        first = sum(sizes[:rank])
        y_all = torch.randint(0, 24, (len(graphs), 3), generator=torch.Generator().manual_seed(3))

        def step(model, G, y, bucket):
            model.train()
            bucket.zero()
            pred = model(G.to(dev))
            loss = sum(torch.nn.functional.cross_entropy(p, y[:, s].to(dev)) for s, p in enumerate(pred)) / len(pred)
            loss.backward()

        model = Hh.code2_model(meta).to(dev)
        bucket = GradBucket(model.parameters())
        step(model, shards[rank], y_all[first:first + sizes[rank]], bucket)
        bucket.all_reduce_mean(local_count=sizes[rank])
        got = bucket.flat.clone()
        ref_model = Hh.code2_model(meta).to(dev)
        ref_bucket = GradBucket(ref_model.parameters())
        step(ref_model, synth.GraphBatch.from_data_list(graphs), y_all, ref_bucket)   # the whole batch, one process
        ref = ref_bucket.flat
        torch.cuda.synchronize()
        worst = 0.0
        off = 0
        for p in bucket.params:
            a, b = got[off:off + p.numel()], ref[off:off + p.numel()]
            scale = float(b.abs().max())
            if scale > 1e-7:
                worst = max(worst, float((a - b).abs().max()) / scale)
            off += p.numel()
        out[rank] = (sizes, worst)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_uneven_shards_match_full_batch_gradients():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert len(out) == 2
    for sizes, worst in out.values():
        assert sizes[0] != sizes[1] and worst < 1e-4, (sizes, worst)
