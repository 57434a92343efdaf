"""CPU-only, world_size = 2 over gloo: the data-parallel path of trainer.py (document sharding, flat-gradient
all-reduce spans, 1/world scaling) with the per-shard gradients supplied by the CPU oracle (the HIP kernels need a
GPU, the exchange logic does not)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import restatement as R
from tests import helpers as H

D, HEADS, FF, NL, WINDOW, B, L = 32, 4, 16, 1, 4, 6, 12


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_batch():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, L, D, generator=g)
    y = (torch.rand(B, L, generator=g) < 0.3).float()
    return {'src_tokens': x, 'src_lengths': torch.full((B,), L), 'tgt_tokens': y, 'src_tokens2': None, 'id': torch.arange(B),
            'domain': None}


def _oracle_grads(model, batch):
    p = {k: v.detach().clone().double().requires_grad_(True) for k, v in model.state_dict().items()}
    scores = R.transformer_scores(batch['src_tokens'].double(), batch['src_lengths'], p, HEADS, R.pyramidal_radii(NL, WINDOW))
    loss = R.tagger_loss(scores, batch['src_lengths'], batch['tgt_tokens'].double(), 'FocalLoss')
    gs = torch.autograd.grad(loss, list(p.values()), allow_unused=True)
    return loss.item(), {k: (g if g is not None else torch.zeros_like(v)) for (k, v), g in zip(p.items(), gs)}


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from multimodaltopicsegmentation_amd.taggers import Transformer_segmenter
    from multimodaltopicsegmentation_amd.trainer import NativeTrainer, shard_batch
    model = Transformer_segmenter(2, D, FF, num_layers=NL, nheads=HEADS, loss_fn='FocalLoss', window_size=WINDOW, compute_dtype='fp32',
                                  max_position_embedding=64, seed=3)
    tr = NativeTrainer(model, lr=1e-3)
    assert tr.world == world
    shard = shard_batch(_make_batch(), rank, world)
    assert shard['src_tokens'].shape[0] == B // world and shard['id'].tolist() == list(range(rank, B, world))
    _, grads = _oracle_grads(model, shard)
    views = model.grad_views()
    for k, g in grads.items():
        views[k].copy_(g.float())
    tr._last_L = L
    spans = tr._reduce_spans()
    # rows of the position table that no batch of this length can touch are excluded from the exchange
    assert spans[0][0] == 2 * D and spans[0][1] == (L + 2) * D and spans[1][1] == model.flat.numel()
    tr.allreduce_grads()
    torch.save({k: v.clone() for k, v in views.items()}, os.path.join(out_dir, f'g{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_allreduce_matches_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g0 = torch.load(os.path.join(tmp_path, 'g0.pt'))
    g1 = torch.load(os.path.join(tmp_path, 'g1.pt'))
    from multimodaltopicsegmentation_amd.taggers import Transformer_segmenter
    model = Transformer_segmenter(2, D, FF, num_layers=NL, nheads=HEADS, loss_fn='FocalLoss', window_size=WINDOW, compute_dtype='fp32',
                                  max_position_embedding=64, seed=3)
    _, full = _oracle_grads(model, _make_batch())
    for k in full:
        assert torch.equal(g0[k], g1[k]), k                                       # both ranks hold the same sum
        # equal-length shards: mean of shard means == global mean (SURVEY.md §8e caveat for ragged batches)
        np.testing.assert_allclose((g0[k] / world).numpy(), full[k].float().numpy(), rtol=2e-4, atol=2e-7, err_msg=k)


def test_shard_batch_partitions_documents():
    from multimodaltopicsegmentation_amd.trainer import shard_batch
    b = _make_batch()
    parts = [shard_batch(b, r, 3) for r in range(3)]
    ids = sorted(i for p in parts for i in p['id'].tolist())
    assert ids == list(range(B))
    assert shard_batch(b, 0, 1) is b
