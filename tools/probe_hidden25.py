import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, traceback
from multimodaltopicsegmentation_amd import TextSegmenter
g = torch.Generator().manual_seed(0)
x = torch.randn(3, 20, 64, generator=g).cuda(); lengths = torch.tensor([20, 11, 5]); y = (torch.rand(3, 20, generator=g) < .3).float().cuda()
for arch, kw in (('Transformer', dict(nheads=4, attention_window=8)), ('BiLSTM', {}), ('biLSTMCRF', {})):
    for dt in ('fp32', 'bf16'):
        try:
            ts = TextSegmenter(2, 64, 25, num_layers=1, architecture=arch, loss_fn='FocalLoss', compute_dtype=dt, **kw).cuda()
            loss = ts.model.loss(x, lengths, y if arch != 'biLSTMCRF' else y.long())
            loss.backward()
            print(arch, dt, 'ok', float(loss))
        except Exception as e:
            print(arch, dt, 'FAIL', type(e).__name__, str(e)[:150])
