"""Checkpoint I/O without a GPU: reference-shaped state_dict keys / shapes (padded storage is invisible), HF's dead tensors are
ignored on load, and the Lightning-style ``load_from_checkpoint`` entry point (train_fit.py:349-371) round-trips."""
import torch


def test_load_from_checkpoint_roundtrip_with_dead_hf_keys(tmp_path):
    from multimodaltopicsegmentation_amd import TextSegmenter
    kw = dict(tagset_size=2, embedding_dim=64, hidden_dim=25, num_layers=2, architecture='Transformer', loss_fn='FocalLoss', nheads=4,
              attention_window=8)
    a = TextSegmenter(**kw)
    sd = {k: v.clone() for k, v in a.state_dict().items()}
    assert sd['model.model.model.encoder.layer.0.intermediate.dense.weight'].shape == (25, 64)        # the reference's shape
    assert sd['model.model.model.encoder.layer.1.output.dense.weight'].shape == (64, 25)
    # what a reference checkpoint carries on top: HF's word embeddings, global-attention projections, pooler, buffers
    sd['model.model.model.embeddings.word_embeddings.weight'] = torch.zeros(8, 64)
    sd['model.model.model.encoder.layer.0.attention.self.query_global.weight'] = torch.zeros(64, 64)
    sd['model.model.model.pooler.dense.weight'] = torch.zeros(64, 64)
    sd['model.model.model.embeddings.position_ids'] = torch.arange(10).unsqueeze(0)
    path = tmp_path / 'epoch=3-val_loss=0.12-threshold=0.40.ckpt'                                       # train_fit.py:337-338 parses this name
    torch.save({'state_dict': sd, 'hyper_parameters': kw}, path)
    b = TextSegmenter.load_from_checkpoint(str(path))
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka
    c = TextSegmenter.load_from_checkpoint(str(path), threshold=0.4)                                   # keyword overrides, as the reference passes them
    assert c.model.th == 0.4


def test_recurrent_checkpoint_shapes_hidden_25():
    from multimodaltopicsegmentation_amd import TextSegmenter
    a = TextSegmenter(2, [50, 26], 25, num_layers=2, architecture='BiLSTMLateFusion', loss_fn='BinaryCrossEntropy')
    sd = a.state_dict()
    assert sd['model.model1.rnn.weight_ih_l0'].shape == (100, 50) and sd['model.model2.rnn.weight_ih_l0_reverse'].shape == (100, 26)
    assert sd['model.model1.rnn.weight_ih_l1'].shape == (100, 50) and sd['model.model1.rnn.weight_hh_l1'].shape == (100, 25)
    assert sd['model.model2.rnn.bias_ih_l0'].shape == (100,) and sd['model.classification.weight'].shape == (1, 100)
    assert torch.all(sd['model.model1.rnn.bias_ih_l0'][25:50] == 1) and float(sd['model.model1.rnn.bias_ih_l0'].sum()) == 25   # forget-gate bias init
    b = TextSegmenter(2, [50, 26], 25, num_layers=2, architecture='BiLSTMLateFusion', loss_fn='BinaryCrossEntropy')
    b.load_state_dict(sd)
    for k, v in b.state_dict().items():
        assert torch.equal(v, sd[k]), k
    # the padded units behind the real ones are exactly zero in storage
    p = dict(b.named_parameters())['model.model1.rnn.weight_hh_l0']
    assert p.shape == (128, 32) and float(p.detach().view(4, 32, 32)[:, 25:, :].abs().sum()) == 0.0
