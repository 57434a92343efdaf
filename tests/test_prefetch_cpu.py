"""prefetch.DevicePrefetcher host logic on the CPU device (no streams): order, contents, wire dtype, error propagation, early exit."""
import pytest
import torch


def _batches(n, B=3, L=5, D=8):
    g = torch.Generator().manual_seed(3)
    return [{'id': torch.arange(B) + 10 * i, 'src_tokens': torch.randn(B, L, D, generator=g), 'src_lengths': torch.tensor([5, 3, 1]),
             'tgt_tokens': (torch.rand(B, L, generator=g) < .3).float(), 'src_tokens2': torch.randn(B, L, 4, generator=g) if i % 2 else None,
             'domain': None} for i in range(n)]


@pytest.mark.parametrize('depth', [1, 2, 4])
def test_order_and_contents(depth):
    from multimodaltopicsegmentation_amd import DevicePrefetcher
    src = _batches(7)
    got = list(DevicePrefetcher(iter(src), 'cpu', depth=depth))
    assert len(got) == 7
    for a, b in zip(src, got):
        assert torch.equal(a['id'], b['id']) and torch.equal(a['src_tokens'], b['src_tokens']) and torch.equal(a['tgt_tokens'], b['tgt_tokens'])
        assert (a['src_tokens2'] is None) == (b['src_tokens2'] is None) and b['src_lengths'] is a['src_lengths'] and b['domain'] is None
        if a['src_tokens2'] is not None:
            assert torch.equal(a['src_tokens2'], b['src_tokens2'])


def test_bf16_wire_rounds_the_embeddings_only():
    from multimodaltopicsegmentation_amd import DevicePrefetcher
    src = _batches(2)
    got = list(DevicePrefetcher(src, 'cpu', wire_dtype='bf16'))
    for a, b in zip(src, got):
        assert b['src_tokens'].dtype == torch.bfloat16 and torch.equal(b['src_tokens'], a['src_tokens'].to(torch.bfloat16))
        assert b['tgt_tokens'].dtype == torch.float32 and torch.equal(a['tgt_tokens'], b['tgt_tokens'])
    assert src[0]['src_tokens'].dtype == torch.float32                      # the caller's batch is not modified


def test_producer_errors_reach_the_consumer_and_early_exit_stops_the_thread():
    import threading
    from multimodaltopicsegmentation_amd import DevicePrefetcher

    def bad():
        yield _batches(1)[0]
        raise RuntimeError('collater failed')
    it = iter(DevicePrefetcher(bad(), 'cpu'))
    next(it)
    with pytest.raises(RuntimeError, match='collater failed'):
        next(it)
    with pytest.raises(TypeError):
        list(DevicePrefetcher([1, 2], 'cpu'))
    with pytest.raises(ValueError):
        DevicePrefetcher([], 'cpu', wire_dtype='fp8')
    n0 = sum(t.name == 'mts-prefetch' and t.is_alive() for t in threading.enumerate())
    for i, _ in enumerate(DevicePrefetcher(iter(_batches(50)), 'cpu', depth=2)):
        if i == 2:
            break
    import time
    time.sleep(0.3)
    assert sum(t.name == 'mts-prefetch' and t.is_alive() for t in threading.enumerate()) <= n0
