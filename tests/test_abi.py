"""CPU-only: the C-ABI library loads, exports every symbol include/mts.h declares, and the ctypes signatures in
multimodaltopicsegmentation_amd/_lib.py have the same arity as the header (no compute calls here)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'mts.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    out = {}
    for m in re.finditer(r'\b(?:int|size_t|const char\*)\s+(mts_\w+)\s*\(([^;{]*?)\)\s*;', src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ('', 'void') else len([a for a in args.split(',')])
        out[m.group(1)] = n
    return out


def test_library_exports_every_declared_symbol():
    from multimodaltopicsegmentation_amd import _lib as L
    decl = _declared()
    assert len(decl) >= 28
    for name, nargs in decl.items():
        assert hasattr(L.lib, name), f'{name} declared in mts.h but not exported'
        assert name in L.SIGNATURES, f'{name} has no ctypes signature'
        assert len(L.SIGNATURES[name][1]) == nargs, f'{name}: header has {nargs} args, ctypes binding {len(L.SIGNATURES[name][1])}'
    assert set(L.SIGNATURES) <= set(decl), set(L.SIGNATURES) - set(decl)
    assert L.lib.mts_version().decode().endswith('gfx950')


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'multimodaltopicsegmentation_amd')
    for fn in os.listdir(pkg):
        if fn.endswith('.py'):
            txt = open(os.path.join(pkg, fn)).read()
            assert 'oracle' not in txt.replace('the oracle', ''), fn


def test_error_codes_map_to_reference_exceptions():
    import pytest
    from multimodaltopicsegmentation_amd import _lib as L
    # argument validation happens before any device work, so this is safe without a GPU
    rc = L.lib.mts_tagger_loss(None, 9, 1, 1, 1, 1, None, None, None, 0.9, 2.0, None, None, None, 0, None, 0)
    assert rc == 1
    with pytest.raises(ValueError):
        L.check(rc)
    assert L.lib.mts_band_slots(15) == 32 and L.lib.mts_band_slots(30) == 64 and L.lib.mts_band_slots(16) == 64
