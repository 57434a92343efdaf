#!/usr/bin/env python3
"""Build container only (needs /root/reference): time the REFERENCE's own CPU path against this repo's CPU port
(oracle/restatement.py, the thing bench.py's cpu_baseline runs on the GPU box) on the same sample, same threads:

    python tools/cpu_ref_vs_port.py [--docs 8] [--threads 8]  ->  profiles/r04_cpu_ref_vs_port.json

BASELINE configs[1] model (d=1792, 8 heads, ff 256, window 30 = radius 15, 1 layer, focal loss), fp32, one optimizer step =
forward + backward + Adam(eps 1e-7); best of 3 after one warm-up.  bench.py quotes the resulting ratio in cpu_baseline.sample so
that the port's number on the GPU box can be tied to what the reference itself would do there (the reference never ships)."""
import argparse
import json
import os
import sys
import time
import types

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get('MTS_REFERENCE', '/root/reference')


def import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    m = types.ModuleType('models.longformer_noffn')
    m.LongformerLayer = type('LongformerLayer', (nn.Module,), {})
    sys.modules['models.longformer_noffn'] = m
    pl = types.ModuleType('pytorch_lightning')

    class _LM(nn.Module):
        def log(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass
    pl.LightningModule = _LM
    sys.modules['pytorch_lightning'] = pl
    sys.modules['segeval'] = types.ModuleType('segeval')
    from models.lightning_model import TextSegmenter
    return TextSegmenter


def best_of(fn, n=3):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--docs', type=int, default=8)
    ap.add_argument('--seq', type=int, default=256)
    ap.add_argument('--threads', type=int, default=8)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    from oracle import restatement as R
    from tests import helpers as H
    D, heads, ff, window, NL = 1792, 8, 256, 30, 1
    g = torch.Generator().manual_seed(99)
    x = torch.randn(args.docs, args.seq, D, generator=g)
    y = (torch.rand(args.docs, args.seq, generator=g) < 0.05).float()
    lengths = torch.full((args.docs,), args.seq)

    TextSegmenter = import_reference()
    ts = TextSegmenter(2, D, ff, num_layers=NL, architecture='Transformer', loss_fn='FocalLoss', nheads=heads, attention_window=window)
    ts.model.device = 'cpu'
    ts.model.eval()
    opt = torch.optim.Adam(ts.model.parameters(), lr=1e-3, eps=1e-7)

    def ref_step():
        opt.zero_grad()
        ts.model.loss(x, lengths, y).backward()
        opt.step()
    t_ref = best_of(ref_step)

    out = {'sample': f'{args.docs} docs x {args.seq} sentences x {D}-d', 'threads': args.threads, 'reference_s_per_step': t_ref,
           'reference_sentences_per_s': args.docs * args.seq / t_ref}
    radii = R.pyramidal_radii(NL, window)
    for name, attn in (('port_shifted', None), ('port_blocked', R.band_attention_blocked)):
        p = H.seeded_params(H.band_param_shapes(D, ff, NL, 1, max_pos=args.seq + 2), 7, torch.float32, True)
        popt = torch.optim.Adam(list(p.values()), lr=1e-3, eps=1e-7)

        def port_step():
            popt.zero_grad()
            R.tagger_loss(R.transformer_scores(x, lengths, p, heads, radii, attention=attn), lengths, y, 'FocalLoss').backward()
            popt.step()
        t = best_of(port_step)
        out[name + '_s_per_step'] = t
        out[name + '_sentences_per_s'] = args.docs * args.seq / t
        out[name + '_vs_reference'] = t_ref / t          # > 1: the port is FASTER than the reference
    out['cpu'] = next((l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')), '?')
    print(json.dumps(out, indent=1))
    with open(os.path.join(ROOT, 'profiles', 'r04_cpu_ref_vs_port.json'), 'w') as f:
        json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
