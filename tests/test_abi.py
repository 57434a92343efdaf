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


def test_tuning_options_are_per_thread_and_the_planner_honours_them():
    """include/mts.h "Threads": mts_set_option from one host thread never changes the plan of a GEMM another thread issues.  Two
    threads ask for different `gemm_tile` values at the same time and each gets, from the planner mts_gemm itself uses
    (mts_gemm_plan: pure host code), the plan IT asked for; a third thread that set nothing gets the cost model's choice."""
    import ctypes as C
    import threading
    from multimodaltopicsegmentation_amd import _lib as L
    M, N, K = 16384, 5376, 1792                     # the forward Q|K|V projection of BASELINE configs[1]: N = 24 x 224 and 21 x 256
    barrier = threading.Barrier(3)
    got, errs = {}, []

    def plan():
        t, s = C.c_int(-1), C.c_int(-1)
        assert L.lib.mts_gemm_plan(L.BF16, L.BF16, L.NT, M, N, K, L.EPI_BIAS, 0, C.byref(t), C.byref(s)) == 0
        return t.value, s.value

    def worker(name, tile):
        try:
            if tile is not None:
                assert L.lib.mts_set_option(b'gemm_tile', tile) == 0
            barrier.wait(timeout=30)                # all three threads have set their option before anyone plans
            seen = {plan() for _ in range(200)}
            barrier.wait(timeout=30)
            got[name] = seen
        except Exception as e:  # noqa: BLE001
            errs.append((name, repr(e)))

    ts = [threading.Thread(target=worker, args=a) for a in (('t128', 128), ('t256', 256), ('default', None))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
    assert not errs, errs
    assert got['t128'] == {(128, 1)} and got['t256'] == {(256, 1)}, got
    assert got['default'] == {(224, 1)}, got                                    # the cost model's own choice for this shape
    assert plan() == (224, 1)                                                   # and this (main) thread never saw their options
    # split-K planning needs a workspace: the q/k/v weight gradient (fp32 C, K = all sentences)
    t, s = C.c_int(0), C.c_int(0)
    assert L.lib.mts_gemm_plan(L.BF16, L.F32, L.TN, 5376, 1792, 16384, 0, 16 * 5376 * 1792 * 4, C.byref(t), C.byref(s)) == 0
    assert t.value == 224 and s.value >= 2
    assert L.lib.mts_gemm_plan(L.BF16, L.F32, L.TN, 5376, 1792, 16384, 0, 0, C.byref(t), C.byref(s)) == 0 and s.value == 1
    assert L.lib.mts_gemm_plan(L.BF16, L.BF16, 7, M, N, K, 0, 0, None, None) == 1                # bad layout -> MTS_ERR_INVALID
