#!/usr/bin/env python3
"""Phase timing of tests/test_gpu_parity_r2.py::test_late_fusion_1024_768_long_documents (which part is slow on a given box?)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import restatement as R
from multimodaltopicsegmentation_amd import BiLSTMLateFusion, _lib as L
if len(sys.argv) > 1: L.check(L.lib.mts_set_option(b'lstm_parts', int(sys.argv[1])))
if len(sys.argv) > 2: torch.set_num_threads(int(sys.argv[2]))
DEV = 'cuda'
B, Lq, D1, D2, Hd, NL = 20, 512, 1024, 768, 256, 2
m = BiLSTMLateFusion(2, [D1, D2], Hd, num_layers=NL, loss_fn='FocalLoss', compute_dtype='bf16', seed=5).to(DEV)
g = torch.Generator().manual_seed(78)
lengths = torch.randint(100, Lq + 1, (B,), generator=g)
lengths[2], lengths[9], lengths[19] = Lq, 1, Lq
x1, x2 = torch.randn(B, Lq, D1, generator=g), torch.randn(B, Lq, D2, generator=g)
y = torch.full((B, Lq), -1.0)
for b, n in enumerate(lengths.tolist()):
    x1[b, n:] = 0.0; x2[b, n:] = 0.0
    y[b, :n] = (torch.rand(n, generator=g) < 0.2).float()
t0 = time.time()
for rep in range(3):
    loss = m.loss(x1.to(DEV), x2.to(DEV), lengths, y.to(DEV)); loss.backward(); torch.cuda.synchronize()
    print(f'product loss+backward rep {rep}: {time.time() - t0:.2f} s, loss {loss.item():.5f}', flush=True); t0 = time.time()
sc, tags = m(x1.to(DEV), x2.to(DEV), lengths); torch.cuda.synchronize()
print(f'product forward+decode: {time.time() - t0:.2f} s', flush=True); t0 = time.time()
p = {k: v.detach().cpu().float().requires_grad_(True) for k, v in m.state_dict().items()}
ref = R.late_fusion_scores(x1, x2, lengths, p, NL, batched=True)
print(f'oracle forward: {time.time() - t0:.2f} s (threads {torch.get_num_threads()})', flush=True); t0 = time.time()
R.tagger_loss(ref, lengths, y, 'FocalLoss').backward()
print(f'oracle backward: {time.time() - t0:.2f} s', flush=True)
