"""CU-pair / CU-quad LSTM recurrence kernels (H = 256; bf16 and, for the quad form, fp32) on awkward batches: a last group with a single document, documents of
length 1 and of full length, odd and even longest lengths per group, and repeated launches (the hand-off between the two
workgroups of a pair is tagged data polled through L2: results must be bitwise the same run after run)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'

from oracle import restatement as R  # noqa: E402


def _rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _check(got, ref, rtol, atol, msg):
    err = (got.detach().float().cpu().double() - ref.detach().double()).abs()
    bad = err > atol + rtol * ref.detach().double().abs()
    assert not bad.any(), f'{msg}: {int(bad.sum())}/{bad.numel()} off, max err {float(err.max()):.3e}'


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float32], ids=['bf16', 'fp32'])
@pytest.mark.parametrize('B,Lq,lengths', [
    (33, 37, None),                                                       # three groups, the last holds one document
    (17, 8, [8] * 17),                                                    # every document full length, even count of steps
    (16, 9, [9, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1]),           # one long document among length-1 ones, odd steps
    (3, 64, [64, 63, 2]),
])
def test_pair_kernels_against_oracle_and_across_launches(B, Lq, lengths, dt):
    from multimodaltopicsegmentation_amd import ops
    H = 256
    bf = dt == torch.bfloat16
    if lengths is None:
        g = torch.Generator().manual_seed(B)
        lengths = torch.randint(1, Lq + 1, (B,), generator=g).tolist()
        lengths[0], lengths[16], lengths[32] = Lq, 7, 5                   # group maxima: even / odd / odd
    N = B * Lq
    xproj = _rnd(N, 8 * H, seed=1).to(dt)
    w_hh = _rnd(2, 4 * H, H, seed=2, scale=1 / math.sqrt(H))
    b_hh = _rnd(2, 4 * H, seed=3, scale=0.1)
    dout = _rnd(N, 2 * H, seed=4).to(dt)
    len_t = torch.tensor(lengths)
    li32 = len_t.to(torch.int32).to(DEV)
    xd, wd, bd, dd = xproj.to(DEV), w_hh.to(DEV), b_hh.to(DEV), dout.to(DEV)

    def run():
        out = torch.full((N, 2 * H), float('nan'), dtype=dt, device=DEV)
        gates = torch.empty(N, 8 * H, dtype=dt, device=DEV)
        cells = torch.empty(N, 2 * H, device=DEV)
        ops.lstm_fwd(xd, wd, bd, li32, B, Lq, H, 2, out, gates, cells)
        dxp = torch.full((N, 8 * H), float('nan'), dtype=dt, device=DEV)
        dwhh = torch.empty(2, 4 * H, H, device=DEV)
        ops.lstm_bwd(wd, li32, out, gates, cells, dd, B, Lq, H, 2, dxp, dwhh)
        torch.cuda.synchronize()
        return out, dxp, dwhh

    out, dxp, dwhh = run()
    xp = xproj.double().view(B, Lq, 8 * H).requires_grad_(True)
    whh = w_hh.double().requires_grad_(True)
    eye, zero = torch.eye(4 * H, dtype=torch.float64), torch.zeros(4 * H, dtype=torch.float64)
    ref = torch.cat([R.lstm_direction(xp[:, :, d * 4 * H:(d + 1) * 4 * H], len_t, eye, whh[d], zero, b_hh[d].double(), rev)
                     for d, rev in ((0, False), (1, True))], dim=2)
    _check(out, ref.view(N, 2 * H), *((2e-2, 2e-2) if bf else (1e-5, 1e-5)), 'out')
    o3, d3 = out.view(B, Lq, -1), dxp.view(B, Lq, -1)
    for b, n in enumerate(lengths):
        if n < Lq:
            assert float(o3[b, n:].abs().max()) == 0.0 and float(d3[b, n:].abs().max()) == 0.0   # padded rows exactly zero
    ref.backward(dout.double().view(B, Lq, 2 * H))
    _check(dxp, xp.grad.view(N, 8 * H), *((5e-2, 3e-2) if bf else (1e-4, 2e-5)), 'dxproj')
    _check(dwhh, whh.grad, *((5e-2, 0.15 * max(1.0, math.sqrt(sum(lengths) / 100.0))) if bf else (1e-4, 1e-4)), 'dw_hh')
    for _ in range(10):
        o2, x2, w2 = run()
        assert torch.equal(o2, out) and torch.equal(x2, dxp) and torch.equal(w2, dwhh)


def _pair_run(ops, B, Lq, lengths, seed=0, dt=torch.bfloat16):
    H = 256
    N = B * Lq
    xd = _rnd(N, 8 * H, seed=seed + 1).to(dt).to(DEV)
    wd = _rnd(2, 4 * H, H, seed=seed + 2, scale=1 / math.sqrt(H)).to(DEV)
    bd = _rnd(2, 4 * H, seed=seed + 3, scale=0.1).to(DEV)
    dd = _rnd(N, 2 * H, seed=seed + 4).to(dt).to(DEV)
    li32 = torch.tensor(lengths, dtype=torch.int32, device=DEV)
    out = torch.full((N, 2 * H), float('nan'), dtype=dt, device=DEV)
    gates = torch.empty(N, 8 * H, dtype=dt, device=DEV)
    cells = torch.empty(N, 2 * H, device=DEV)
    ops.lstm_fwd(xd, wd, bd, li32, B, Lq, H, 2, out, gates, cells)
    dxp = torch.full((N, 8 * H), float('nan'), dtype=dt, device=DEV)
    dwhh = torch.empty(2, 4 * H, H, device=DEV)
    ops.lstm_bwd(wd, li32, out, gates, cells, dd, B, Lq, H, 2, dxp, dwhh)
    torch.cuda.synchronize()
    return out, dxp, dwhh


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float32], ids=['bf16', 'fp32'])
def test_pair_path_splits_large_batches_into_resident_launches(dt):
    """A pair needs both of its workgroups resident (one per CU): a launch covers at most 64 pairs = 512 documents x 2
    directions, larger batches run as consecutive launches over document ranges.  B = 1100 (three launches: 512 + 512 + 76)
    must equal, bit for bit, the same documents run in batches that fit one launch; and forcing 8 pairs per launch on a
    100-document batch (13 launches) must equal the single-launch result."""
    from multimodaltopicsegmentation_amd import ops, _lib as L
    B, Lq = 1100, 6
    g = torch.Generator().manual_seed(5)
    lengths = torch.randint(1, Lq + 1, (B,), generator=g).tolist()
    lengths[0] = lengths[511] = lengths[512] = lengths[1099] = Lq
    out, dxp, dwhh = _pair_run(ops, B, Lq, lengths, dt=dt)
    assert not torch.isnan(out.float()).any() and not torch.isnan(dxp.float()).any()
    L.check_async()
    H = 256
    # the same rows through launches of <= 512 documents each: slice the inputs exactly as _pair_run builds them
    N = B * Lq
    xd = _rnd(N, 8 * H, seed=1).to(dt).to(DEV)
    wd = _rnd(2, 4 * H, H, seed=2, scale=1 / math.sqrt(H)).to(DEV)
    bd = _rnd(2, 4 * H, seed=3, scale=0.1).to(DEV)
    for b0, b1 in ((0, 400), (400, 800), (800, 1100)):
        n = (b1 - b0) * Lq
        o = torch.empty(n, 2 * H, dtype=dt, device=DEV)
        gt = torch.empty(n, 8 * H, dtype=dt, device=DEV)
        c = torch.empty(n, 2 * H, device=DEV)
        ops.lstm_fwd(xd[b0 * Lq:b1 * Lq].contiguous(), wd, bd, torch.tensor(lengths[b0:b1], dtype=torch.int32, device=DEV), b1 - b0, Lq, H, 2, o, gt, c)
        assert torch.equal(o, out[b0 * Lq:b1 * Lq]), (b0, b1)
    try:
        ref = _pair_run(ops, 100, 9, [9] * 50 + [3] * 50, seed=7, dt=dt)
        L.lib.mts_set_option(b'lstm_pair_max_pairs', 8)
        got = _pair_run(ops, 100, 9, [9] * 50 + [3] * 50, seed=7, dt=dt)
        for a, r in zip(got, ref):
            assert torch.equal(a, r)
    finally:
        L.lib.mts_set_option(b'lstm_pair_max_pairs', 64)


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float32], ids=['bf16', 'fp32'])
def test_pair_timeout_is_reported_by_the_next_call(dt):
    """A partner poll that gives up used to be silent (status word set on the device, never read).  With the re-poll budget
    forced to 0 every hand-off that is not already there times out: the launch completes (bounded spins), and the NEXT
    mts_lstm_* call -- or mts_async_status() -- returns MTS_ERR_TIMEOUT once; after that the library is usable again."""
    from multimodaltopicsegmentation_amd import ops, _lib as L
    try:
        L.lib.mts_set_option(b'lstm_pair_spin_limit', 0)
        _pair_run(ops, 32, 40, [40] * 32, dt=dt)               # poisoned launches; the backward call may already report the forward's timeout
        raised = False
    except L.MtsError as e:
        raised = 'timed out' in str(e)
    finally:
        L.lib.mts_set_option(b'lstm_pair_spin_limit', -1)
    torch.cuda.synchronize()
    if not raised:
        with pytest.raises(L.MtsError, match='timed out'):
            L.check_async()
    L.lib.mts_async_status()                                   # drain whatever the second poisoned launch left
    L.check_async()                                            # clean again
    out, dxp, _ = _pair_run(ops, 32, 40, [40] * 32, dt=dt)
    assert not torch.isnan(out.float()).any() and not torch.isnan(dxp.float()).any()
    L.check_async()


def test_two_concurrent_pair_grids_on_two_streams():
    """Late fusion runs two recurrences at once on two HIP streams (rnn_taggers.BiLSTMLateFusion._fwd): two 128-workgroup
    grids together still fit the 256 CUs.  B = 512 per stream = the per-launch maximum; results equal the serial run."""
    from multimodaltopicsegmentation_amd import ops, _lib as L
    B, Lq = 512, 12
    lengths = [Lq] * B
    ref_a = _pair_run(ops, B, Lq, lengths, seed=11)
    ref_b = _pair_run(ops, B, Lq, lengths, seed=23)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    # _pair_run synchronises at its end, so drive the two streams by hand: enqueue both forwards, then wait
    H, dt, N = 256, torch.bfloat16, B * Lq
    bufs = []
    for seed in (11, 23):
        bufs.append(dict(x=_rnd(N, 8 * H, seed=seed + 1).to(dt).to(DEV), w=_rnd(2, 4 * H, H, seed=seed + 2, scale=1 / math.sqrt(H)).to(DEV),
                         b=_rnd(2, 4 * H, seed=seed + 3, scale=0.1).to(DEV), out=torch.empty(N, 2 * H, dtype=dt, device=DEV),
                         gates=torch.empty(N, 8 * H, dtype=dt, device=DEV), cells=torch.empty(N, 2 * H, device=DEV)))
    li32 = torch.tensor(lengths, dtype=torch.int32, device=DEV)
    torch.cuda.synchronize()
    for st, bf in zip((s1, s2), bufs):
        with torch.cuda.stream(st):
            ops.lstm_fwd(bf['x'], bf['w'], bf['b'], li32, B, Lq, H, 2, bf['out'], bf['gates'], bf['cells'])
    torch.cuda.synchronize()
    L.check_async()
    assert torch.equal(bufs[0]['out'], ref_a[0]) and torch.equal(bufs[1]['out'], ref_b[0])


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float32], ids=['bf16', 'fp32'])
@pytest.mark.parametrize('H', [256, 32, 64], ids=['quad', 'generic', 'mfma_or_generic'])
def test_backward_in_two_calls_equals_the_single_call(H, dt):
    """mts_lstm_bwd_recurrence + mts_lstm_bwd_whh (the second on ANOTHER stream, behind an event, as the recurrent taggers issue them) against
    mts_lstm_bwd: dxproj and dW_hh bit for bit, on every recurrence path (CU-quad at H = 256; the kernels that build h_{t-1} themselves below)."""
    from multimodaltopicsegmentation_amd import ops
    B, Lq = 19, 23
    g = torch.Generator().manual_seed(5)
    lengths = torch.randint(1, Lq + 1, (B,), generator=g)
    lengths[0] = Lq
    N = B * Lq
    xd = _rnd(N, 8 * H, seed=1).to(dt).to(DEV)
    wd = _rnd(2, 4 * H, H, seed=2, scale=1 / math.sqrt(H)).to(DEV)
    bd = _rnd(2, 4 * H, seed=3, scale=0.1).to(DEV)
    dd = _rnd(N, 2 * H, seed=4).to(dt).to(DEV)
    li32 = lengths.to(torch.int32).to(DEV)
    out = torch.empty(N, 2 * H, dtype=dt, device=DEV)
    gates = torch.empty(N, 8 * H, dtype=dt, device=DEV)
    cells = torch.empty(N, 2 * H, device=DEV)
    ops.lstm_fwd(xd, wd, bd, li32, B, Lq, H, 2, out, gates, cells)
    dxp1 = torch.full((N, 8 * H), float('nan'), dtype=dt, device=DEV)
    dw1 = torch.full((2, 4 * H, H), float('nan'), device=DEV)
    ops.lstm_bwd(wd, li32, out, gates, cells, dd, B, Lq, H, 2, dxp1, dw1)
    dxp2 = torch.full((N, 8 * H), float('nan'), dtype=dt, device=DEV)
    dw2 = torch.full((2, 4 * H, H), float('nan'), device=DEV)
    ws = ops.lstm_workspace(dt, B, Lq, H, 2, DEV, tag='test_split')
    ops.lstm_bwd_recurrence(wd, li32, out, gates, cells, dd, B, Lq, H, 2, dxp2, ws)
    ev = torch.cuda.Event()
    ev.record()
    side = torch.cuda.Stream()
    side.wait_event(ev)
    with torch.cuda.stream(side):
        ops.lstm_bwd_whh(li32, out, dxp2, B, Lq, H, 2, dw2, ws)
    torch.cuda.synchronize()
    view = torch.int16 if dt == torch.bfloat16 else torch.int32
    assert torch.equal(dxp1.view(view), dxp2.view(view))
    assert torch.equal(dw1.view(torch.int32), dw2.view(torch.int32)) and not torch.isnan(dw2).any()


def test_backward_on_another_host_thread_follows_the_forwards_recurrence_form():
    """The tuning switches are per host thread and torch's autograd runs backward nodes on a thread of its own: a forward under
    mts_set_option("lstm_parts", 2) (CU-pair form: its own layout of the saved gates / cells) followed by mts_lstm_bwd from a thread whose
    switches are the defaults (CU quad) must still read the state in the form it was written -- the library records the form per saved
    state.  Same bits as the backward issued from the forward's thread."""
    import threading
    from multimodaltopicsegmentation_amd import _lib as L, ops
    B, Lq, H, dt = 18, 21, 256, torch.bfloat16
    lengths = torch.randint(1, Lq + 1, (B,), generator=torch.Generator().manual_seed(9))
    lengths[0] = Lq
    N = B * Lq
    xd = _rnd(N, 8 * H, seed=1).to(dt).to(DEV)
    wd = _rnd(2, 4 * H, H, seed=2, scale=1 / math.sqrt(H)).to(DEV)
    bd = _rnd(2, 4 * H, seed=3, scale=0.1).to(DEV)
    dd = _rnd(N, 2 * H, seed=4).to(dt).to(DEV)
    li32 = lengths.to(torch.int32).to(DEV)
    out = torch.empty(N, 2 * H, dtype=dt, device=DEV)
    gates = torch.empty(N, 8 * H, dtype=dt, device=DEV)
    cells = torch.empty(N, 2 * H, device=DEV)
    res = {}

    def bwd(tag):
        dxp = torch.full((N, 8 * H), float('nan'), dtype=dt, device=DEV)
        dw = torch.empty(2, 4 * H, H, device=DEV)
        ops.lstm_bwd(wd, li32, out, gates, cells, dd, B, Lq, H, 2, dxp, dw)
        torch.cuda.synchronize()
        res[tag] = (dxp, dw)
    try:
        L.check(L.lib.mts_set_option(b'lstm_parts', 2))
        ops.lstm_fwd(xd, wd, bd, li32, B, Lq, H, 2, out, gates, cells)
        bwd('same thread')
        th = threading.Thread(target=bwd, args=('other thread',))
        th.start()
        th.join()
    finally:
        L.check(L.lib.mts_set_option(b'lstm_parts', 4))
    assert torch.equal(res['same thread'][0].view(torch.int16), res['other thread'][0].view(torch.int16))
    assert torch.equal(res['same thread'][1].view(torch.int32), res['other thread'][1].view(torch.int32))
    assert not torch.isnan(res['other thread'][0].float()).any()
