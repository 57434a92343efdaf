"""The REAL N > 1 step path of trainer.NativeTrainer on the GPU (train_fit.py:284-296 Trainer(gpus=N)): two processes,
both on cuda:0, talking over gloo (a one-GPU box has no second device for RCCL; the exchange logic is backend-agnostic and the
kernels, hooks and stream waits are the ones an 8-GPU run executes).  Each rank runs `NativeTrainer.step` -- gradient-ready hooks ->
asynchronous all-reduce per span -> stream-level waits -> fused Adam -- on its document shard; afterwards

  * both ranks hold bit-identical parameters, and
  * they equal the parameters of a ONE-process run of the same steps on the global batch

for the restricted-window transformer (per-projection release of the q/k/v gradient) and for late fusion (second encoder and its
announcements on a side stream, models/CRF.py:420-425), with equal-length shards and, token-weighted, with ragged ones.
"""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu

B, L, STEPS = 8, 48, 2


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(kind):
    if kind == 'transformer':
        from multimodaltopicsegmentation_amd.taggers import Transformer_segmenter
        return Transformer_segmenter(2, 64, 32, num_layers=2, nheads=4, loss_fn='FocalLoss', window_size=6, compute_dtype='fp32',
                                     max_position_embedding=128, seed=11)
    from multimodaltopicsegmentation_amd.rnn_taggers import BiLSTMLateFusion
    return BiLSTMLateFusion(2, [40, 24], 32, num_layers=2, loss_fn='FocalLoss', compute_dtype='fp32', seed=11)


def _batches(kind, ragged):
    g = torch.Generator().manual_seed(77)
    out = []
    for _ in range(STEPS):
        lengths = torch.full((B,), L, dtype=torch.int64)
        if ragged:
            lengths = torch.randint(3, L + 1, (B,), generator=g)
            lengths[0] = L                                   # every rank collates to the same padded length
            lengths[1] = L
        d1, d2 = (64, None) if kind == 'transformer' else (40, 24)
        x = torch.randn(B, L, d1, generator=g)
        y = (torch.rand(B, L, generator=g) < 0.2).float()
        x2 = torch.randn(B, L, d2, generator=g) if d2 else None
        for b, n in enumerate(lengths.tolist()):
            x[b, n:] = 0.0
            y[b, n:] = -1.0
            if x2 is not None:
                x2[b, n:] = 0.0
        out.append({'src_tokens': x, 'src_lengths': lengths, 'tgt_tokens': y, 'src_tokens2': x2, 'id': torch.arange(B), 'domain': None})
    return out


def _to_dev(batch):
    return {k: (v.cuda() if isinstance(v, torch.Tensor) and k != 'src_lengths' else v) for k, v in batch.items()}


def _run(kind, ragged, rank, world):
    from multimodaltopicsegmentation_amd.trainer import NativeTrainer, shard_batch
    model = _build(kind).to('cuda')
    tr = NativeTrainer(model, lr=1e-3, optimizer='Adam', token_weighted=ragged)
    losses = []
    for batch in _batches(kind, ragged):
        losses.append(float(tr.step(_to_dev(shard_batch(batch, rank, world)))))
    torch.cuda.synchronize()
    return model.flat.detach().cpu().clone(), losses, tr


def _worker(rank, world, port, out_dir, kind, ragged, schedule='allreduce'):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), MTS_DP_SCHEDULE=schedule)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    flat, losses, tr = _run(kind, ragged, rank, world)
    assert tr.world == world and tr.exchange_schedule == schedule
    # the overlapped path ran: hooks installed, nothing left pending
    assert getattr(tr.model, 'grad_hooks_cover_all', False) and tr._pending == [] and tr.model._grad_hook is not None
    torch.save({'flat': flat, 'losses': losses}, os.path.join(out_dir, f'r{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize('kind,ragged', [('transformer', False), ('latefusion', False), ('transformer', True), ('latefusion', True)])
def test_two_rank_native_step_equals_the_single_process_step(tmp_path, kind, ragged):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), kind, ragged), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, 'r0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'r1.pt'))
    assert torch.equal(r0['flat'], r1['flat'])                    # both ranks applied the same exchanged gradient
    single, losses, _ = _run(kind, ragged, 0, 1)
    init = _build(kind).flat.detach().clone()
    moved = (single - init).abs()
    assert float(moved.max()) > 1e-3                              # the steps did something (lr 1e-3, Adam: ~lr per step)
    # Adam's update is lr * m / (sqrt(v) + 1e-7): where |g| ~ 1e-7 the quotient amplifies fp32 summation-order noise of the
    # two-shard sum against the one-pass sum, so the bar is absolute: a small fraction of one step's movement (lr = 1e-3)
    diff = (r0['flat'] - single).abs()
    assert float(diff.max()) <= 2e-5, (float(diff.max()), int(diff.argmax()))
    assert float(diff.mean()) <= 1e-7, float(diff.mean())
    # the first step's local losses average (weighted by valid sentences when ragged) to the global batch's loss
    b0 = _batches(kind, ragged)[0]
    n = [int(b0['src_lengths'][r::world].sum()) for r in range(world)]
    w = [v / sum(n) for v in n] if ragged else [0.5, 0.5]
    assert abs(w[0] * r0['losses'][0] + w[1] * r1['losses'][0] - losses[0]) <= 2e-6 * max(1.0, abs(losses[0]))


@pytest.mark.timeout(600)
@pytest.mark.parametrize('kind', ['transformer', 'latefusion'])
def test_reduce_scatter_all_gather_schedule_in_the_overlapped_step(tmp_path, kind):
    """MTS_DP_SCHEDULE=rs_ag (trainer.NativeTrainer exchange_schedule): every span announced by the backward is reduce-scattered and
    the reduced shards gathered, asynchronously, the optimizer ordered behind them.  At world 2 each element is one addition either way:
    the parameters after two steps must equal the all-reduce schedule's bit for bit."""
    import torch.multiprocessing as mp
    world, flats = 2, {}
    for sched in ('allreduce', 'rs_ag'):
        out = tmp_path / sched
        out.mkdir()
        mp.spawn(_worker, args=(world, _free_port(), str(out), kind, False, sched), nprocs=world, join=True)
        r0, r1 = torch.load(out / 'r0.pt'), torch.load(out / 'r1.pt')
        assert torch.equal(r0['flat'], r1['flat'])
        flats[sched] = r0['flat']
    assert torch.equal(flats['allreduce'], flats['rs_ag'])
