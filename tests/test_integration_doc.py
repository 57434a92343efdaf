"""INTEGRATION.md's reference-side ctypes stub is EXECUTED here, so the document cannot drift from include/mts.h again
(round 1 shipped stubs with 11 of 14 and 15 of 17 arguments).  CPU: the block runs against the built library and every
``argtypes`` list it declares has the header's arity.  GPU: band_attention() / focal_loss() from the block against the oracle."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_namespace():
    from multimodaltopicsegmentation_amd import _lib as L
    md = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    m = re.search(r'```python\n(# models/mts_ffi\.py.*?)```', md, flags=re.S)
    assert m, 'INTEGRATION.md lost its mts_ffi.py block'
    code = m.group(1).replace("C.CDLL('libmts_hip.so')", f'C.CDLL({L.LIB_PATH!r})')
    ns = {}
    exec(compile(code, 'INTEGRATION.md:mts_ffi.py', 'exec'), ns)
    return ns, code


def test_integration_stub_declares_the_header_arity():
    from tests.test_abi import _declared
    ns, code = _stub_namespace()
    decl = _declared()
    bound = re.findall(r'lib\.(mts_\w+)\.argtypes', code)
    assert {'mts_band_attn_fwd', 'mts_tagger_loss'} <= set(bound)
    for name in bound:
        assert len(getattr(ns['lib'], name).argtypes) == decl[name], f'{name}: doc stub has {len(getattr(ns["lib"], name).argtypes)} args, mts.h {decl[name]}'
    # and every CALL in the block passes exactly that many arguments
    for name in bound:
        for call in re.finditer(r'lib\.' + name + r'\((.*?)\)\)?\n', code, flags=re.S):
            args = call.group(1)
            depth, n = 0, 1
            for ch in args:
                depth += ch in '([' 
                depth -= ch in ')]'
                n += (ch == ',' and depth == 0)
            if 'argtypes' not in call.group(0):
                assert n == decl[name], (name, n, decl[name], args)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_integration_stub_runs_against_the_oracle(dtype):
    from oracle import restatement as R
    ns, _ = _stub_namespace()
    B, L, D, heads, radius = 3, 50, 64, 2, 15
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(B * L, 3 * D, generator=g)
    lengths = torch.tensor([50, 17, 3], dtype=torch.int32)
    ctx, probs = ns['band_attention'](qkv.to('cuda', dtype), lengths.cuda(), B, L, D, heads, radius)
    q, k, v = (qkv.to(dtype).double()[:, i * D:(i + 1) * D].view(B, L, heads, D // heads) for i in range(3))
    ref = R.band_attention(q, k, v, lengths.long(), radius).reshape(B * L, D)
    got = ctx.double().cpu()
    for b, n in enumerate(lengths.tolist()):
        sl = slice(b * L, b * L + n)
        assert float((got[sl] - ref[sl]).abs().max()) < (2e-5 if dtype == torch.float32 else 3e-2)
    assert probs.shape == (B * L, heads * 32)
    scores = torch.randn(B, L, 1, generator=g)
    tg = torch.full((B, L), -1.0)
    for b, n in enumerate(lengths.tolist()):
        tg[b, :n] = (torch.rand(n, generator=g) < 0.3).float()
    loss, dsc = ns['focal_loss'](scores.cuda(), tg.cuda(), lengths.cuda(), 0.9, 2.0)
    s = scores.double().requires_grad_(True)
    ref_loss = R.tagger_loss(s, lengths.long(), tg.double(), 'FocalLoss')
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 2e-6
    assert float((dsc.cpu().double() - s.grad).abs().max()) < 1e-7
