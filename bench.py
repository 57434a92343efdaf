#!/usr/bin/env python3
"""Benchmark of the tagger hot path on MI355X (contract: see the task brief / DESIGN.md §Measurement).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = forward + backward + Adam(eps 1e-7) of the restricted-window transformer tagger (BASELINE.json
configs[1]: d=1792 early-fused embeddings, 8 heads, ff=256, one-sided window 15, 1 layer, focal loss, bf16) on one
synthetic batch of 64 documents x 256 sentences per GPU, inputs resident in HBM.  Weak scaling: every rank gets its
own 64 documents; gradients are all-reduced over RCCL.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')    # before HIP initialises: see multimodaltopicsegmentation_amd/_lib.py (two-stream late fusion next to RCCL)

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense bf16 peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
PMC_TRAFFIC_FILE = 'r04_pmc_traffic.json'   # written by tools/pmc_traffic.py from two rocprofv3 --pmc passes, stamped with csrc_sha()


def fwd_bwd_mflop_per_sentence(D, ff, radius, n_layers, n_out=1):
    """MAC = 2 FLOP, fwd+bwd = 3 x fwd."""
    macs = 0
    for li in range(n_layers):
        r = radius * (n_layers - li)
        macs += 3 * D * D + D * D + 2 * (2 * r + 1) * D + 2 * D * ff
    macs += n_out * D
    return macs * 2 * 3 / 1e6


def synthetic_batch(B, L, D, rank, device, D2=None, ragged=False):
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn(B, L, D, generator=g)
    y = (torch.rand(B, L, generator=g) < 0.05).float()
    y[:, -1] = 0.0                                             # utils/load_datasets_precomputed.py:172
    lengths = torch.full((B,), L, dtype=torch.int64)
    if ragged:                                                 # len ~ U{L/4 .. L}, zero-padded, targets padded with -1
        lengths = torch.randint(L // 4, L + 1, (B,), generator=g)
        for b, n in enumerate(lengths.tolist()):
            x[b, n:] = 0.0
            y[b, n:] = -1.0
            y[b, n - 1] = 0.0
    batch = {'src_tokens': x.to(device), 'src_lengths': lengths, 'tgt_tokens': y.to(device),
             'src_tokens2': None, 'id': torch.arange(B), 'domain': None}
    if D2:
        batch['src_tokens2'] = torch.randn(B, L, D2, generator=g).to(device)
    return batch


def csrc_sha():
    """Hash of the kernel sources in the tree: ties a measurement artifact (profiles/*_pmc_traffic.json) to the build it was taken
    on -- .git does not travel to the GPU box, the sources do."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'multimodaltopicsegmentation_amd', 'csrc')
    for fn in sorted(os.listdir(d)):
        if fn.endswith(('.hip', '.h')):
            h.update(fn.encode())
            h.update(open(os.path.join(d, fn), 'rb').read())
    return h.hexdigest()[:16]


def cpu_baseline(args, D, ff, heads, window, n_layers):
    """The CPU oracle (a port of the reference's arithmetic, oracle/restatement.py, with the block-wise matmul form of the band
    attention) timed on this box's host cores on a bounded sample of the same workload: fwd + bwd + Adam, fp32.  The thread count
    is the one that maximises the port's throughput on a short probe (an 8-32 document batch does not scale to 128 threads)."""
    from oracle import restatement as R
    from tests import helpers as H
    radii = R.pyramidal_radii(n_layers, window)
    g = torch.Generator().manual_seed(99)

    def make(Bc):
        p = H.seeded_params(H.band_param_shapes(D, ff, n_layers, 1, max_pos=args.seq + 2), 7, torch.float32, True)
        opt = torch.optim.Adam(list(p.values()), lr=1e-3, eps=1e-7)
        x = torch.randn(Bc, args.seq, D, generator=g)
        y = (torch.rand(Bc, args.seq, generator=g) < 0.05).float()
        lengths = torch.full((Bc,), args.seq)

        def step():
            t0 = time.perf_counter()
            opt.zero_grad()
            R.tagger_loss(R.transformer_scores(x, lengths, p, heads, radii, attention=R.band_attention_blocked), lengths, y, 'FocalLoss').backward()
            opt.step()
            return time.perf_counter() - t0
        return step

    ncpu = os.cpu_count() or 1
    probe = make(min(8, args.cpu_docs))
    probe()                                                    # warm-up (allocator, thread pool)
    # (all of a 128-core / 256-thread host is never the answer for a 32-document batch: one step took 34 s there; probing it would
    # cost more than the whole measurement)
    cands = sorted({t for t in (8, 16, 32, 64) if t <= ncpu} or {ncpu})
    timing = {}
    for t in cands:
        torch.set_num_threads(t)
        probe()
        timing[t] = probe()
    cores = min(timing, key=timing.get)
    torch.set_num_threads(cores)
    step = make(args.cpu_docs)
    step()
    best = min(step(), step())
    sample = (f'{args.cpu_docs} docs x {args.seq} sentences x {D}-d, same model, fp32 fwd+bwd+Adam, best of 2 after 1 warm-up '
              f'({best:.2f} s/step); threads picked from {cands} by a probe on {min(8, args.cpu_docs)} docs '
              f'({", ".join(f"{t}: {v:.2f} s" for t, v in timing.items())})')
    try:
        rv = json.load(open(os.path.join(ROOT, 'profiles', 'r04_cpu_ref_vs_port.json')))
        sample += (f'; port vs the reference itself, measured in the build container ({rv["sample"]}, {rv["threads"]} threads, '
                   f'tools/cpu_ref_vs_port.py): reference {rv["reference_sentences_per_s"]:.0f} sentences/s, this port '
                   f'{rv["port_blocked_sentences_per_s"]:.0f} = {rv["port_blocked_vs_reference"]:.2f}x the reference')
    except (OSError, ValueError, KeyError):
        sample += '; port-vs-reference ratio: profiles/r04_cpu_ref_vs_port.json missing'
    return {'value': args.cpu_docs * args.seq / best, 'unit': 'sentences/s', 'cores': cores, 'kind': 'port', 'sample': sample}


def infer_latency(args, model, batch, wl, world, rank):
    """Latency of one predict_step-sized call: forward + greedy decode (Viterbi for the CRF head) + the device-to-host copy of the
    boundary lists, i.e. what predict.py / test_step wait for per document.  Wall-clock per call (host launch overhead included:
    at batch 1 the path is launch-bound), median / p95 over --steps calls after --warmup."""
    model.eval()
    if getattr(args, 'graph', False):
        model.inference_graphs = True            # Transformer_segmenter: replay forward + decode as one hipGraph per (B, L) shape
    x, lengths = batch['src_tokens'], batch['src_lengths']
    x2 = batch.get('src_tokens2')
    call = (lambda: model(x, x2, lengths)) if x2 is not None else (lambda: model(x, lengths))
    for _ in range(max(args.warmup, 2)):
        call()
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.steps):
        t0 = time.perf_counter()
        call()                                   # returns python lists: the call itself synchronises
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med, p95 = ts[len(ts) // 2], ts[min(len(ts) - 1, int(0.95 * len(ts)))]
    n_sent = int(lengths.sum())
    if rank == 0:
        print(json.dumps({'metric': 'inference latency per call (forward + decode + D2H)', 'value': 1e3 * med, 'unit': 'ms', 'p95_ms': 1e3 * p95,
                          'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'higher_is_better': False, 'dtype': args.dtype,
                          'data': 'synthetic', 'sentences_per_s': n_sent / med,
                          'config': {'workload': f'inference: {wl}, {args.docs} document(s) x {args.seq} sentences per call' + (', hipGraph replay' if getattr(args, 'graph', False) and hasattr(model, 'inference_graphs') else '')}}))


def other_configs(device, steps=50, warmup=10):
    """The other BASELINE.json configurations (and the fp32 parity mode an unconfigured drop-in user runs) timed in the same
    process right after the headline run, so that the driver's record witnesses them too: K (= 50) steps of fwd + bwd + Adam after W
    warm-up steps, inputs resident in HBM, one synchronise on either side.  Never part of `value` / `roofline`."""
    from multimodaltopicsegmentation_amd.rnn_taggers import BiLSTM, BiLSTMLateFusion, BiRnnCrf
    from multimodaltopicsegmentation_amd.taggers import Transformer_segmenter
    from multimodaltopicsegmentation_amd.trainer import NativeTrainer
    runs = [
        ('configs[2] BiLSTM 2x256 focal, 64x256x1792, bf16', lambda: BiLSTM(2, 1792, 256, num_layers=2, loss_fn='FocalLoss', compute_dtype='bf16', seed=1234), 256, 1792, None, 34.61),
        ('configs[2] BiLSTM 2x256 + CRF NLL, 64x256x1792, bf16', lambda: BiRnnCrf(2, 1792, 256, num_layers=2, compute_dtype='bf16', seed=1234), 256, 1792, None, 34.62),
        ('configs[4] per-GPU workload: late fusion 1024+768, 64x512, bf16', lambda: BiLSTMLateFusion(2, [1024, 768], 256, num_layers=2, loss_fn='FocalLoss', compute_dtype='bf16', seed=1234), 512, 1024, 768, 47.19),
        ('configs[2] BiLSTM 2x256 focal in fp32 parity mode (the drop-in classes\' default dtype)', lambda: BiLSTM(2, 1792, 256, num_layers=2, loss_fn='FocalLoss', compute_dtype='fp32', seed=1234), 256, 1792, None, 34.61),
        ('configs[1] in fp32 parity mode (the drop-in classes\' default dtype)', lambda: Transformer_segmenter(2, 1792, 256, num_layers=1, nheads=8, loss_fn='FocalLoss', window_size=30, compute_dtype='fp32', seed=1234), 256, 1792, None, fwd_bwd_mflop_per_sentence(1792, 256, 15, 1)),
    ]
    out = {}
    for label, make, seq, D, D2, mflop in runs:
        model = make().to(device)
        trainer = NativeTrainer(model, lr=1e-3, optimizer='Adam')
        batch = synthetic_batch(64, seq, D, 0, device, D2)
        for _ in range(warmup):
            trainer.step(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = trainer.step(batch)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        sps = 64 * seq * steps / dt
        out[label] = {'ms_per_step': 1e3 * dt / steps, 'sentences_per_s': sps, 'steps': steps, 'warmup': warmup,
                      'model_tflops': sps * mflop / 1e6, 'final_loss': float(loss)}
        del model, trainer, batch
    return out


def h2d_leg(trainer, docs, seq, D, D2, device, steps, warmup, wire='fp32', source='pinned'):
    """The step with HOST batches in the loop (what a drop-in user who feeds the tagger from a DataLoader gets): three distinct
    synthetic batches cycled through prefetch.DevicePrefetcher (depth 2: copies on a side stream, two batches ahead).
    source = 'pinned': the batches sit in pinned host memory (DataLoader(pin_memory=True) / a collater that writes into pinned
    buffers) -> the leg measures how well the PCIe transfer hides under the step; 'pageable': the prefetcher's producer thread also
    stages every batch into its pinned ring (one host memcpy of the batch per step).  Never part of `value`."""
    from multimodaltopicsegmentation_amd.prefetch import DevicePrefetcher
    if source == 'collater':
        return h2d_collater_leg(trainer, docs, seq, D, D2, device, steps, warmup, wire)
    host = []
    for i in range(3):
        b = synthetic_batch(docs, seq, D, 100 + i, 'cpu', D2)
        if wire == 'bf16':
            # the embeddings are HELD in bf16 on the host (a dataset converted once when it is loaded): converting 117 MB per step on
            # the host inside the loop costs 14 ms per batch (measured: profiles/r03 h2d), seven steps' worth
            b = {k: (v.to(torch.bfloat16) if isinstance(v, torch.Tensor) and k in ('src_tokens', 'src_tokens2') else v) for k, v in b.items()}
        if source == 'pinned':
            b = {k: (v.pin_memory() if isinstance(v, torch.Tensor) and k in ('src_tokens', 'src_tokens2', 'tgt_tokens') else v) for k, v in b.items()}
        host.append(b)
    n_total = warmup + steps

    def cycle():
        for i in range(n_total):
            yield host[i % 3]
    pf = DevicePrefetcher(cycle(), device, depth=2)
    t0 = None
    for i, batch in enumerate(pf):
        if i == warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        trainer.step(batch)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per_batch = sum(v.numel() * (2 if (wire == 'bf16' and k != 'tgt_tokens') else 4) for k, v in host[0].items()
                    if isinstance(v, torch.Tensor) and k in ('src_tokens', 'src_tokens2', 'tgt_tokens'))
    return {'ms_per_step': 1e3 * dt / steps, 'sentences_per_s': docs * seq * steps / dt, 'steps': steps, 'warmup': warmup,
            'wire_dtype': wire + (' (the host batches already hold bf16 embeddings)' if wire == 'bf16' else ''), 'host_memory': source, 'MB_per_batch': per_batch / 1e6,
            'pcie_GBps_sustained': per_batch * steps / dt / 1e9,
            'note': 'host batches -> DevicePrefetcher (side-stream H2D two batches ahead) -> step; NOT the headline value'}


def h2d_collater_leg(trainer, docs, seq, D, D2, device, steps, warmup, wire):
    """documents -> the PRODUCT's collater -> DevicePrefetcher -> step: a synthetic corpus of 4 x `docs` documents held on the host,
    `AudioPortionDataset(pin_memory=True, wire_dtype=wire)` pads every batch into its pinned ring (mts_collate_pad, one threaded pass;
    wire = 'bf16': the corpus is converted once, when the dataset is built), the prefetcher sends the collater's own buffers.  The
    collation runs inside the timed loop, on the prefetcher's producer thread.  Never part of `value`."""
    from multimodaltopicsegmentation_amd.encoder_dataset import AudioPortionDataset
    from multimodaltopicsegmentation_amd.prefetch import DevicePrefetcher
    g = torch.Generator().manual_seed(4321)
    n_docs = 4 * docs
    lines = [(torch.randn(seq, D, generator=g), (torch.rand(seq, generator=g) < 0.05).float().tolist(), f'doc{i}') for i in range(n_docs)]
    second = [(torch.randn(seq, D2, generator=g), None, f'doc{i}') for i in range(n_docs)] if D2 else None
    threads = min(16, os.cpu_count() or 8)
    ds = AudioPortionDataset(lines, {'0': 0, '1': 1}, CRF=False, truncate=False, second_input=second, pin_memory=True, wire_dtype=wire,
                             pin_slots=4, collate_threads=threads)
    n_total = warmup + steps
    t_collate = [0.0, 0]

    def batches():
        for i in range(n_total):
            idx = [(i * docs + j) % n_docs for j in range(docs)]
            t0 = time.perf_counter()
            b = ds.collater(ds.__getitems__(idx))         # the batched fetch a torch DataLoader makes (encoder_dataset.py: indices -> fast path)
            t_collate[0] += time.perf_counter() - t0
            t_collate[1] += 1
            yield b
    pf = DevicePrefetcher(batches(), device, depth=2)
    t0 = None
    for i, batch in enumerate(pf):
        if i == warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        trainer.step(batch)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per_batch = docs * seq * ((D + (D2 or 0)) * (2 if wire == 'bf16' else 4) + 4)
    return {'ms_per_step': 1e3 * dt / steps, 'sentences_per_s': docs * seq * steps / dt, 'steps': steps, 'warmup': warmup,
            'wire_dtype': wire + (' (the corpus is held in bf16 on the host)' if wire == 'bf16' else ''), 'host_memory': 'the collater\'s pinned ring',
            'MB_per_batch': per_batch / 1e6, 'pcie_GBps_sustained': per_batch * steps / dt / 1e9,
            'collater_ms_per_batch': 1e3 * t_collate[0] / max(1, t_collate[1]), 'collate_threads': threads,
            'note': 'documents -> AudioPortionDataset(pin_memory=True).collater -> DevicePrefetcher -> step; NOT the headline value'}


def self_launch(n, rehearsal):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start N fresh ranks (one process per GPU) under
    torch.distributed.run on 127.0.0.1 with the same arguments and return their exit code.  Rank 0's JSON line goes straight to the
    inherited stdout.  This parent never initialises the GPU (torch.cuda.device_count() only counts devices), so the children are
    ordinary child processes of a GPU-free launcher."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if rehearsal:
        if n > 6:                                  # a GPU box allows 6 processes on its card at once
            print(f'bench.py: MTS_BENCH_REHEARSAL runs every rank on cuda:0; at most 6 ranks, got --gpus {n}', file=sys.stderr)
            return 2
    elif have < n:
        print(f'bench.py: --gpus {n} requested but {have} GPU(s) are visible on this node', file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--graph', action='store_true', help='--infer: replay the transformer forward + decode as a hipGraph')
    ap.add_argument('--gpus', type=int, default=1)
    # defaults: 200 timed steps (0.4 s) after 25 warm-up steps.  A 20-step region right after the warm-up's synchronise runs 4-5 % slower than the
    # same build over 200+ steps (2.04 against 1.95 ms on one box: the first steps after an idle queue), i.e. it measures the start-up, not the step
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=25)
    ap.add_argument('--docs', type=int, default=64, help='documents per GPU')
    ap.add_argument('--seq', type=int, default=256, help='sentences per document')
    ap.add_argument('--dim', type=int, default=1792)
    ap.add_argument('--arch', default='transformer', choices=['transformer', 'bilstm', 'bilstm_crf', 'latefusion'])
    ap.add_argument('--layers', type=int, default=None)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--cpu-docs', type=int, default=32)
    ap.add_argument('--sustained-steps', type=int, default=1000, help='extra timed leg after the K-step region (0 = skip); reported under "extra"')
    ap.add_argument('--h2d', default=None, choices=['pinned', 'pageable', 'collater'], help='variant line: host batches through prefetch.DevicePrefetcher '
                    'in the loop (PCIe-inclusive step); the batches live in pinned or pageable host memory')
    ap.add_argument('--h2d-wire', default='fp32', choices=['fp32', 'bf16'], help='--h2d: dtype of the embeddings on the wire')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the 50-step lines of the other BASELINE configurations under "extra"')
    ap.add_argument('--no-kernel-timer', action='store_true')
    ap.add_argument('--timer-every', type=int, default=5, help='bracket the dominant kernel with HIP events in every n-th step of the timed region')
    ap.add_argument('--ragged', action='store_true', help='ragged lengths U{L/4..L} (value counts valid sentences only)')
    ap.add_argument('--infer', action='store_true', help='inference latency instead of the training step: model(x, lengths) -> scores + boundary lists '
                    '(predict_step, lightning_model.py:678-683; the reference decodes test documents with batch_size=1, train_fit.py:154); '
                    'typical use: --infer --docs 1 --seq 2437 (the longest RadioNews document)')
    ap.add_argument('--no-pack', action='store_true', help='with --ragged: keep the padded rows in the encoder (A/B of the packed training path)')
    args = ap.parse_args()

    import torch.distributed as dist
    # MTS_BENCH_REHEARSAL=1: every rank on cuda:0 over gloo -- exercises the N>1 code path (sharding, gradient-ready hooks,
    # async exchange, max-over-ranks timing) on a one-GPU box; the numbers it prints mean nothing
    rehearsal = os.environ.get('MTS_BENCH_REHEARSAL') == '1'
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` by itself (the reference's entry is a plain Trainer(gpus=N), train_fit.py:284-296): this
        # process never touches the GPU; it starts N fresh ranks under torch.distributed.run and exits with their code
        raise SystemExit(self_launch(args.gpus, rehearsal))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}, or without torch.distributed.run')
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if rehearsal:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=device)      # backend "nccl" is RCCL on ROCm
    # MTS_BENCH_SINGLE_RANK_DP=1: a process group of ONE rank over RCCL and the trainer's overlapped exchange path forced on -- what
    # the N > 1 step costs on this GPU besides the wire time (hook order, per-projection weight gradients, collective launches)
    single_rank_dp = world == 1 and os.environ.get('MTS_BENCH_SINGLE_RANK_DP') == '1'
    if single_rank_dp:
        import socket
        s_ = socket.socket()
        s_.bind(('127.0.0.1', 0))
        port_ = s_.getsockname()[1]
        s_.close()
        dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port_}', rank=0, world_size=1, device_id=device)

    from multimodaltopicsegmentation_amd import ops
    from multimodaltopicsegmentation_amd.rnn_taggers import BiLSTM, BiLSTMLateFusion, BiRnnCrf
    from multimodaltopicsegmentation_amd.taggers import Transformer_segmenter
    from multimodaltopicsegmentation_amd.trainer import NativeTrainer

    # A/B hook: MTS_OPTIONS="key=value,..." -> mts_set_option(key, value) before anything runs (tuning switches of include/mts.h)
    for kv in filter(None, os.environ.get('MTS_OPTIONS', '').split(',')):
        from multimodaltopicsegmentation_amd import _lib as _L
        k, v = kv.split('=')
        _L.check(_L.lib.mts_set_option(k.encode(), int(v)))

    torch.manual_seed(1234)
    D, heads, ff, window = args.dim, 8, 256, 30                    # window 30 = one-sided radius 15 ("win=15")
    D2 = None
    if args.arch == 'transformer':
        n_layers = args.layers or 1
        model = Transformer_segmenter(2, D, ff, num_layers=n_layers, nheads=heads, loss_fn='FocalLoss', window_size=window,
                                      compute_dtype=args.dtype, seed=1234)
        mflop = fwd_bwd_mflop_per_sentence(D, ff, window // 2, n_layers)
        wl = f'restricted-window transformer tagger d={D} heads={heads} ff={ff} radius={window // 2} layers={n_layers} focal loss'
    elif args.arch == 'bilstm':
        n_layers = args.layers or 2
        model = BiLSTM(2, D, 256, num_layers=n_layers, loss_fn='FocalLoss', compute_dtype=args.dtype, seed=1234)
        mflop = 34.61 if (D == 1792 and n_layers == 2) else None
        wl = f'BiLSTM tagger d={D} H=256 layers={n_layers} focal loss'
    elif args.arch == 'bilstm_crf':
        n_layers = args.layers or 2
        model = BiRnnCrf(2, D, 256, num_layers=n_layers, compute_dtype=args.dtype, seed=1234)
        mflop = 34.62 if (D == 1792 and n_layers == 2) else None
        wl = f'BiLSTM + CRF head d={D} H=256 layers={n_layers} CRF NLL'
    else:
        n_layers = args.layers or 2
        D, D2 = 1024, 768
        model = BiLSTMLateFusion(2, [D, D2], 256, num_layers=n_layers, loss_fn='FocalLoss', compute_dtype=args.dtype, seed=1234)
        mflop = 47.19 if n_layers == 2 else None
        wl = f'late fusion (concat) two BiLSTMs {D}+{D2} H=256 layers={n_layers} focal loss'
    # which BASELINE.json configuration this run is (per-GPU workload); anything else is labelled as a variant
    std = args.docs == 64 and args.dim == 1792 and not args.ragged
    if args.arch == 'transformer' and std and args.seq == 256 and n_layers == 1:
        cfg_label = 'BASELINE configs[1]' if world == 1 else f'BASELINE configs[3] ({world} of 8 GPUs)' if world < 8 else 'BASELINE configs[3]'
    elif args.arch in ('bilstm', 'bilstm_crf') and std and args.seq == 256 and n_layers == 2:
        cfg_label = 'BASELINE configs[2] (' + ('CRF NLL head' if args.arch == 'bilstm_crf' else 'focal-loss head') + ')'
    elif args.arch == 'latefusion' and args.docs == 64 and args.seq == 512 and n_layers == 2:
        cfg_label = 'BASELINE configs[4] per-GPU workload' + (f' ({world} GPU{"s" if world > 1 else ""})')
    else:
        cfg_label = 'variant (not a BASELINE.json configuration)'
    model = model.to(device)
    trainer = NativeTrainer(model, lr=1e-3, optimizer='Adam', always_hook=single_rank_dp)
    if single_rank_dp:
        cfg_label = 'variant (data-parallel step path forced on in a one-rank RCCL group): ' + cfg_label
    batch = synthetic_batch(args.docs, args.seq, D, rank, device, D2, ragged=args.ragged)
    if args.no_pack and hasattr(model, 'pack_rows'):
        model.pack_rows = False

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if args.infer:
        return infer_latency(args, model, batch, wl, world, rank)
    if args.h2d:
        for _ in range(2):
            trainer.step(batch)                          # allocate workspaces, prime the planner
        leg = h2d_leg(trainer, args.docs, args.seq, D, D2, device, args.steps, max(args.warmup, 3), args.h2d_wire, args.h2d)
        if rank == 0:
            print(json.dumps({'metric': 'sentences/sec (fwd+bwd), host batches in the loop', 'value': world * leg['sentences_per_s'], 'unit': 'sentences/s',
                              'n_gpus': world, 'steps': args.steps, 'warmup': max(args.warmup, 3), 'ms_per_step': leg['ms_per_step'],
                              'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
                              'config': {'workload': f'variant (PCIe-inclusive: {args.h2d} host batches, {args.h2d_wire} on the wire) of {cfg_label}: {wl}, '
                                                     f'{args.docs} docs x {args.seq} sentences per GPU'}, 'h2d': leg}))
        return

    # Warm-up with every GEMM / band-attention launch bracketed by HIP events: finds the dominant kernel symbol and fills the
    # per-kernel table.  An event pair costs a few microseconds of stream time (14 timed launches = ~5 % of this step), so the
    # timed region itself brackets only the launches of that dominant symbol (+ the band-attention forward, the HBM headline).
    wtimer = None if args.no_kernel_timer else ops.KernelTimer()
    for i in range(args.warmup):
        ops.TIMER = wtimer if i > 0 else None        # the very first step only primes caches (which kernel the cost model picks per shape)
        trainer.step(batch)
    sync()
    ops.TIMER = None
    wsum = wtimer.summary() if wtimer is not None else {}
    wsym = {}
    for tag, (n, ms) in wsum.items():
        if tag[0] == 'gemm':
            wsym[(tag[1], tag[7])] = wsym.get((tag[1], tag[7]), 0.0) + ms
    dom_key = max(wsym.items(), key=lambda kv: kv[1])[0] if wsym else None
    # (fewer than two warm-up steps: nothing was measured yet, so bracket everything as before)
    timer = None if args.no_kernel_timer else ops.KernelTimer(
        only=(lambda tag: tag[0] == 'gemm' and (tag[1], tag[7]) == dom_key) if dom_key is not None else None)
    # ... and only in every `--timer-every`-th step of the region (default 5): measured on MI355X, three bracketed launches per step cost
    # 0.08-0.16 ms of a 1.95 ms step (each event is a timestamp packet behind a queue barrier) -- 2.05 ms with every step bracketed against
    # 1.95 ms with none, same build, same box.  `launches_timed` in the JSON says how many launches the average is over.
    t0 = time.perf_counter()
    for i in range(args.steps):
        ops.TIMER = timer if (timer is not None and i % max(1, args.timer_every) == 0) else None
        loss = trainer.step(batch)
    sync()
    elapsed = time.perf_counter() - t0
    ops.TIMER = None
    loss_val = float(loss)
    ksum = timer.summary() if timer is not None else {}
    if rank == 0 and os.environ.get('MTS_BENCH_DETAIL'):
        for tag, (n, ms) in sorted(wsum.items(), key=lambda kv: -kv[1][1]):
            extra = ''
            if tag[0] == 'gemm':
                extra = f' {2.0 * tag[4] * tag[5] * tag[6] * n / (ms * 1e-3) / 1e12:7.1f} TFLOP/s'
            print(f'[detail] {str(tag):60s} n={n:4d} avg={1e3 * ms / n:9.1f} us{extra}', file=sys.stderr)
    if world > 1:
        t = torch.tensor([elapsed], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    # sustained figure: the K-step value above is taken on a chip that has been busy for a fraction of a second; under sustained
    # load the clock settles lower (DVFS), so the same build is timed again over >= 1000 steps (no per-kernel events)
    sustained = None
    if args.sustained_steps > 0:
        sync()
        t1 = time.perf_counter()
        for _ in range(args.sustained_steps):
            trainer.step(batch)
        sync()
        sus = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([sus], device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            sus = float(t)
        sustained = sus

    if rank == 0:
        sentences = world * int(batch['src_lengths'].sum()) * args.steps      # valid sentences (= docs x seq unless --ragged)
        value = sentences / elapsed
        out = {
            'metric': 'sentences/sec (fwd+bwd)', 'value': value, 'unit': 'sentences/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'{cfg_label}: {wl}, {args.docs} docs x {args.seq} sentences per GPU, '
                                   f'fwd+bwd+Adam(eps 1e-7), inputs resident in HBM' + (', ragged lengths U{L/4..L}' + (' (padded rows kept)' if args.no_pack else ' (packed)') if args.ragged else ''),
                       'global_batch_docs': world * args.docs, 'sentences_per_doc': args.seq, 'parallelism': f'dp{world} (document-sharded, ' + ('gloo ' if rehearsal else 'RCCL ') + ('reduce-scatter + all-gather' if trainer.exchange_schedule == 'rs_ag' else 'all-reduce') + (', every rank on cuda:0: REHEARSAL, numbers mean nothing)' if rehearsal else ')')},
            'final_loss': loss_val,
        }
        if sustained is not None:
            out['extra'] = {'sustained_steps': args.sustained_steps, 'sustained_ms_per_step': 1e3 * sustained / args.sustained_steps,
                            'sustained_value': world * int(batch['src_lengths'].sum()) * args.sustained_steps / sustained,
                            'note': 'same build, timed again over sustained_steps right after the K-step region (clock settled under load)'}
        if mflop:
            out['model_tflops'] = value * mflop / 1e6
            out['model_mfma_frac'] = out['model_tflops'] / (MFMA_BF16_PEAK_TFLOPS * world)
        # roofline of the dominant kernel (by total time).  One GEMM symbol = one (layout, output dtype, tile) instantiation;
        # the timed region of a weight-gradient call also holds its split-K reduce kernel (named in the label).
        sym = {(0, 128): 'gemm_bf16_kernel<NT,bf16,GLDS> (128x128)', (1, 128): 'gemm_bf16_kernel<NN,bf16,GLDS> (128x128)',
               (2, 128): 'gemm_bf16_kernel<TN,f32,GLDS> (128x128)',
               (0, 224): 'gemm_bf16_224p_kernel<NT,bf16> (256x224, four waves, LDS-DMA, persistent)', (1, 224): 'gemm_bf16_224_kernel<NN,bf16> (256x224)',
               (2, 224): 'gemm_bf16_224t_kernel<TN,f32> (256x224, four waves, LDS-DMA, 32-deep units)',
               (1, 225): 'gemm_bf16_224n_kernel<NN,bf16> (256x224, four waves, LDS-DMA, k-strided B)',
               (0, 226): 'gemm_bf16_224d_kernel<NT,bf16> (256x224, four waves, LDS-DMA, one tile per workgroup, residual through the LDS)',
               (0, 256): 'gemm_bf16_256_kernel<NT,bf16> (256x256)', (1, 256): 'gemm_bf16_256_kernel<NN,bf16> (256x256)',
               (2, 256): 'gemm_bf16_256_kernel<TN,f32> (256x256)'}
        rocprof_names = {(0, 128): ['void gemm_bf16_kernel<0, bool _Accum, bool, E>(GemmArgs)'], (1, 128): ['void gemm_bf16_kernel<1, bool _Accum, bool, E>(GemmArgs)'],
                         (2, 128): ['void gemm_bf16_kernel<2, float, true>(GemmArgs)'],
                         (0, 224): ['gemm_bf16_224p_kernel(GemmArgs)'], (0, 226): ['gemm_bf16_224d_kernel(GemmArgs)'], (1, 224): ['_Z20gemm_bf16_224_kernelILi1EDF16bLb0ELb1EEv8GemmArgs'],
                         (2, 224): ['void gemm_bf16_224t_kernel<false>(GemmArgs, int, void const*, void const*, float*)', 'gemm_bf16_224t_kernel(GemmArgs, int)'],
                         (1, 225): ['gemm_bf16_224n_kernel(GemmArgs)'],
                         (2, 256): ['void gemm_bf16_256_kernel<2, float, true>(GemmArgs)'],
                         (0, 256): ['_Z20gemm_bf16_256_kernelILi0EDF16bLb0EEv8GemmArgs'], (1, 256): ['_Z20gemm_bf16_256_kernelILi1EDF16bLb0EEv8GemmArgs']}
        def aggregate(summary):
            per_sym, other = {}, {}
            for tag, (n, ms) in summary.items():
                if tag[0] == 'gemm':
                    _, layout, a_dt, c_dt, M, N, K, tile = tag
                    d = per_sym.setdefault((layout, tile), {'launches': 0, 'ms': 0.0, 'flop': 0.0})
                    d['launches'] += n
                    d['ms'] += ms
                    d['flop'] += n * 2.0 * M * N * K
                else:
                    other[tag[0]] = {'launches': n, 'avg_us': 1e3 * ms / n}
            return per_sym, other

        per_sym, other = aggregate(ksum)             # timed region: the dominant GEMM symbol and the band-attention forward only
        wper_sym, wother = aggregate(wsum if wsum else ksum)   # warm-up steps: every GEMM / band launch
        if per_sym:
            key, dom = max(per_sym.items(), key=lambda kv: kv[1]['ms'])
            ach = dom['flop'] / (dom['ms'] * 1e-3) / 1e12
            # HBM-side bytes per launch from the committed PMC passes (profiles/: FETCH_SIZE x2 + WRITE_SIZE, see tools/pmc_traffic.py)
            # the JSON is stamped with the hash of the kernel sources it was measured on; a different build gets traffic = null
            traffic, traffic_note = None, None
            try:
                pmc = json.load(open(os.path.join(ROOT, 'profiles', PMC_TRAFFIC_FILE)))
                if pmc.get('_csrc_sha') != csrc_sha():
                    traffic_note = f'{PMC_TRAFFIC_FILE} was taken on kernel sources {pmc.get("_csrc_sha")}, this build is {csrc_sha()}: not quoted'
                else:
                    traffic = sum(pmc[k]['bytes_per_launch'] for k in rocprof_names.get(key, []) if k in pmc) or None
            except (OSError, ValueError, KeyError) as e:
                traffic_note = f'{PMC_TRAFFIC_FILE}: {type(e).__name__}'
            out['roofline'] = {'bound': 'mfma', 'kernel': sym.get(key, str(key)), 'achieved': ach, 'peak': MFMA_BF16_PEAK_TFLOPS,
                               'unit': 'TFLOP/s', 'frac': ach / MFMA_BF16_PEAK_TFLOPS, 'traffic': traffic,
                               'launches_timed': dom['launches'], 'avg_launch_us': 1e3 * dom['ms'] / dom['launches'],
                               'algorithmic_gflop_per_launch': dom['flop'] / dom['launches'] / 1e9}
            if traffic_note:
                out['roofline']['traffic_note'] = traffic_note
            out['kernels_note'] = 'per-launch averages from the warm-up steps (every GEMM / band launch bracketed by HIP events; the bracket of a split-K weight gradient ends before its reduce launch: mts_gemm_set_mid_hook); the roofline entry is from the timed region'
            out['kernels'] = {sym.get(k, str(k)): {'launches': d['launches'], 'avg_us': 1e3 * d['ms'] / d['launches'],
                                                    'tflops': d['flop'] / (d['ms'] * 1e-3) / 1e12} for k, d in wper_sym.items()}
            # the HBM-bound headline kernel: band attention, 14 336 algorithmic bytes per sentence (bf16 q,k,v in, ctx out)
            if 'band_fwd' not in other and 'band_fwd' in wother:
                other['band_fwd'] = wother['band_fwd']          # (bracketed in the warm-up steps only: every bracket in the timed region costs the step ~16 us)
            if 'band_fwd' in other:
                by = int(batch['src_lengths'].sum()) * 4 * D * (2 if args.dtype == 'bf16' else 4)
                gbs = by / (other['band_fwd']['avg_us'] * 1e-6) / 1e9
                out['kernels']['band_attn_fwd'] = {**other['band_fwd'], 'algorithmic_GBps': gbs, 'hbm_frac': gbs / HBM_PEAK_GBS}
            if 'band_bwd' in wother:
                out['kernels']['band_attn_bwd (one pass)'] = wother['band_bwd']
        if world == 1 and not single_rank_dp and not args.no_other_configs and cfg_label == 'BASELINE configs[1]':
            out.setdefault('extra', {})['other_configs'] = other_configs(device)
        if world == 1 and not single_rank_dp and not args.no_other_configs and cfg_label == 'BASELINE configs[1]':
            out['extra']['h2d'] = {f'{src}, {w} on the wire': h2d_leg(trainer, args.docs, args.seq, D, D2, device, 60, 15, w, src)
                                   for src, w in (('pinned', 'fp32'), ('pageable', 'fp32'), ('pinned', 'bf16'), ('collater', 'fp32'), ('collater', 'bf16'))}
        if world == 1 and not args.no_cpu_baseline and args.arch == 'transformer':
            out['cpu_baseline'] = cpu_baseline(args, D, ff, heads, window, n_layers)
        print(json.dumps(out))
    if world > 1 or single_rank_dp:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
