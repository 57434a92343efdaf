"""``TextSegmenter`` -- the drop-in boundary (reference: models/lightning_model.py:178-781).

Same constructor signature, the same ``training_step / validation_step / test_step / predict_step /
configure_optimizers`` methods over the same batch dict (EncoderDataset.py:91-152), the same error behaviour
(ValueError for unknown architectures :250, NotImplementedError for search_threshold in test :570 and
switch='bias' :235).  ``pytorch_lightning`` is optional: when it is importable the class derives from
``pl.LightningModule`` so train_fit.py:300-335 / predict.py:228-311 drive it unchanged; otherwise it is a plain
``nn.Module`` with no-op ``log`` / ``log_dict`` that the native trainer (trainer.py) drives.
"""
import numpy as np
import torch
import torch.nn as nn

from . import metrics
from .rnn_taggers import BiLSTM, BiLSTMLateFusion, BiRnnCrf
from .taggers import Transformer_segmenter

try:  # pragma: no cover - not installed in the build image
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # noqa: BLE001
    class _Base(nn.Module):
        def log(self, *a, **k):
            self._last_logged = getattr(self, '_last_logged', {})
            if a:
                self._last_logged[a[0]] = a[1] if len(a) > 1 else None

        def log_dict(self, d, *a, **k):
            self._last_logged = getattr(self, '_last_logged', {})
            self._last_logged.update(d)

        @classmethod
        def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict=True, **kwargs):
            """Lightning's entry point as train_fit.py:349-371 / predict.py:228-256 use it: a ``.ckpt`` is a pickled dict with
            ``state_dict`` (keys ``model.<tagger key>``) and, if the module saved them, ``hyper_parameters``; keyword arguments
            override the latter.  HF's dead tensors in a reference checkpoint are dropped by the tagger's load hook."""
            ckpt = torch.load(checkpoint_path, map_location=map_location or 'cpu', weights_only=False)
            hp = dict(ckpt.get('hyper_parameters', {}) or {}) if isinstance(ckpt, dict) else {}
            hp.update(kwargs)
            obj = cls(**hp)
            obj.load_state_dict(ckpt['state_dict'] if isinstance(ckpt, dict) and 'state_dict' in ckpt else ckpt, strict=strict)
            return obj

# architectures of the reference that are outside the hot path
_OUT_OF_SCOPE = ('SimpleBiLSTM', 'MLP', 'Transformer-CRF', 'RecurrentLongT5', 'BiLSTMRestrictedMHA', 'SwitchBiLSTM', 'SheikhBiLSTM')


class TextSegmenter(_Base):
    def __init__(self, tagset_size, embedding_dim, hidden_dim, num_layers=1, batch_first=True, LSTM=True, bidirectional=True,
                 architecture='biLSTMCRF', lr=0.01, dropout_in=0.0, dropout_out=0.0, optimizer='SGD', positional_encoding=True,
                 nheads=8, end_boundary=False, threshold=None, search_threshold=False, metric='Pk', cosine_loss=False,
                 zero_baseline=False, loss_fn='CrossEntropy', no_validation=False, all_results=False, all_scores=False, alpha=0.9,
                 gamma=2, attention_window=120, switch='dense', compute_dtype=None, ksplit=False):
        super().__init__()
        self.validation = not no_validation
        # ksplit=True (extension, default off): an early-fusion model takes the batch's two embedding tensors ('src_tokens' = text,
        # 'src_tokens2' = audio, as AudioPortionDataset(second_input=...) collates them) as ONE input of width D1 + D2 without the
        # host-side concat of utils/load_datasets_precomputed.py:158-161 ever being made (datasets.load_dataset_from_precomputed(
        # ..., split_modalities=True) keeps the two directories apart)
        self.ksplit = bool(ksplit)
        self.cos = cosine_loss
        self.double_input = False
        self.domain = False
        self.delete_last_target = False
        if architecture == 'biLSTMCRF':
            self.cos = False
            self.model = BiRnnCrf(tagset_size, embedding_dim, hidden_dim, num_layers=num_layers, bidirectional=bidirectional,
                                  dropout_in=dropout_in, dropout_out=dropout_out, batch_first=batch_first, LSTM=LSTM,
                                  architecture='rnn', compute_dtype=compute_dtype)
        elif architecture == 'BiLSTM':
            self.model = BiLSTM(tagset_size, embedding_dim, hidden_dim, num_layers=num_layers, bidirectional=bidirectional,
                                dropout_in=dropout_in, dropout_out=dropout_out, batch_first=batch_first, LSTM=LSTM, loss_fn=loss_fn,
                                threshold=threshold, alpha=alpha, gamma=gamma, compute_dtype=compute_dtype)
        elif architecture == 'Transformer':
            self.model = Transformer_segmenter(tagset_size, embedding_dim, hidden_dim, num_layers=num_layers, dropout_in=dropout_in,
                                               dropout_out=dropout_out, batch_first=batch_first, loss_fn=loss_fn,
                                               positional_encoding=positional_encoding, nheads=nheads, threshold=threshold,
                                               alpha=alpha, gamma=gamma, window_size=attention_window, compute_dtype=compute_dtype)
        elif architecture == 'BiLSTMLateFusion':
            self.model = BiLSTMLateFusion(tagset_size, embedding_dim, hidden_dim, num_layers=num_layers, bidirectional=bidirectional,
                                          dropout_in=dropout_in, dropout_out=dropout_out, batch_first=batch_first, LSTM=LSTM,
                                          loss_fn=loss_fn, threshold=threshold, alpha=alpha, gamma=gamma, compute_dtype=compute_dtype)
            self.double_input = True
        elif architecture in _OUT_OF_SCOPE:
            if architecture == 'SwitchBiLSTM' and switch == 'bias':
                raise NotImplementedError()                                  # lightning_model.py:235
            raise NotImplementedError(f"architecture '{architecture}' exists in the reference but is outside the accelerated "
                                      'hot path (SURVEY.md §2/§8f)')
        else:
            raise ValueError('No other architectures implemented yet')       # lightning_model.py:250
        self.learning_rate = lr
        self.optimizer = optimizer
        self.eb = end_boundary
        self.threshold = threshold
        self.s_th = search_threshold
        self.metric = metric
        self.best_th, self.losses, self.targets = [], [], []
        self.zero_base = zero_baseline
        self.all = bool(all_results)
        if self.all:
            self.results = []
        self.all_scores = bool(all_scores)
        if self.all_scores:
            self.scores = []

    def forward(self, x):
        return self.model(x)

    def _sentence(self, batch):
        """The model input of a batch: 'src_tokens', or the (text, audio) pair when ksplit is on and the batch carries both."""
        if self.ksplit and not self.double_input and batch.get('src_tokens2') is not None:
            return (batch['src_tokens'], batch['src_tokens2'])
        return batch['src_tokens']

    # ---- lightning_model.py:273-309 -------------------------------------------------------------------
    def training_step(self, batch, batch_idx):
        sentence, target, lengths = self._sentence(batch), batch['tgt_tokens'], batch['src_lengths']
        segments = batch['src_segments'] if self.cos else None
        self.best_th, self.losses, self.targets = [], [], []
        if self.double_input:
            sentence2 = batch['src_tokens2']
            try:
                loss = self.model.loss(sentence, sentence2, lengths, target, segments=segments)
            except TypeError:
                loss = self.model.loss(sentence, sentence2, lengths, target)
        else:
            try:
                loss = self.model.loss(sentence, lengths, target, segments=segments)
            except TypeError:
                loss = self.model.loss(sentence, lengths, target)
        self.log('training_loss', loss, on_step=True, on_epoch=True, prog_bar=True, logger=True)
        return loss

    # ---- lightning_model.py:320-351 -------------------------------------------------------------------
    def validation_step(self, batch, batch_idx):
        sentence, target, lengths = self._sentence(batch), batch['tgt_tokens'], batch['src_lengths']
        if self.s_th:
            if self.double_input:
                scores, tags = self.model(sentence, batch['src_tokens2'], lengths)
            else:
                scores, tags = self.model(sentence, lengths)
            for index, score in enumerate(scores):
                self.losses.append(score[:lengths[index]].detach().cpu().numpy())
                self.targets.append(target[index][:lengths[index]].detach().cpu().numpy())
            return None
        with torch.no_grad():
            if self.double_input:
                loss = self.model.loss(sentence, batch['src_tokens2'], lengths, target)
            else:
                loss = self.model.loss(sentence, lengths, target)
        self.log_dict({'val_loss': loss, 'threshold': 0.5})
        return loss

    # ---- lightning_model.py:558-676 -------------------------------------------------------------------
    def test_step(self, batch, batch_idx):
        sentence, target, lengths = self._sentence(batch), batch['tgt_tokens'], batch['src_lengths']
        if self.s_th:
            raise NotImplementedError()                                      # lightning_model.py:570
        score = None
        if self.zero_base:
            threshold = 0.4
            tags = [np.zeros(int(n)) for n in lengths]
        else:
            threshold = self.threshold if self.threshold is not None else .4
            if not threshold:
                threshold = 0.5
            self.model.th = threshold
            if self.double_input:
                score, tags = self.model(sentence, batch['src_tokens2'], lengths)
            else:
                score, tags = self.model(sentence, lengths)
        b_like = self.metric.lower() in ('b', 'scaiano')
        sums = {'p': 0.0, 'r': 0.0, 'f1': 0.0, 'b': 0.0, 'pk': 0.0, 'wd': 0.0}
        target = target.clone() if self.eb else target
        for i, tag in enumerate(tags):
            tag = list(tag)
            if self.eb:
                tag[-1] = 0
                target[i][-1] = 0
            tgt = target[i][:int(lengths[i])].detach().cpu().numpy()
            if self.metric.lower() == 'b':
                p, r, f1, b = metrics.B_measure(tag, tgt)
                sums['p'] += p; sums['r'] += r; sums['f1'] += f1; sums['b'] += b
            elif self.metric.lower() == 'scaiano':
                p, r, f1 = metrics.WinPR(tag, tgt)
                sums['p'] += p; sums['r'] += r; sums['f1'] += f1
            else:
                sums['pk'] += float(metrics.compute_Pk(np.array(tag), tgt))
                sums['f1'] += metrics.f1_boundary(tgt.astype(int), np.array(tag).astype(int))
                try:
                    sums['wd'] += float(metrics.compute_window_diff(np.array(tag), tgt))
                except AssertionError:
                    sums['wd'] += float(metrics.compute_Pk(np.array(tag), tgt))
        n = len(target)
        if b_like:
            results = {'b_precision': sums['p'] / n, 'b_recall': sums['r'] / n, 'b_f1': sums['f1'] / n, 'threshold': threshold}
            results['test_loss'] = sums['b'] / n if self.metric.lower() == 'b' else results.pop('b_f1')
        else:
            results = {'Pk_loss': sums['pk'] / n, 'F1_loss': sums['f1'] / n, 'WD_loss': sums['wd'] / n, 'threshold': threshold}
            key = {'F1': 'F1_loss', 'WD': 'WD_loss'}.get(self.metric, 'Pk_loss')
            results['test_loss'] = results.pop(key)
        if self.all:
            self.results.append(results)
        if self.all_scores and score is not None:
            self.scores.extend([s.detach().cpu().numpy() for s in score])
        self.log_dict(results, on_epoch=True, prog_bar=True)
        return results

    # ---- lightning_model.py:678-683 -------------------------------------------------------------------
    def predict_step(self, batch, batch_idx):
        if getattr(self, 'double_input', False) and batch.get('src_tokens2') is not None:
            # the reference calls model(sentence, lengths) here and raises TypeError for late-fusion models; serve them instead
            score, tags = self.model(batch['src_tokens'], batch['src_tokens2'], batch['src_lengths'])
        else:
            score, tags = self.model(self._sentence(batch), batch['src_lengths'])
        return tags

    # ---- lightning_model.py:759-781 -------------------------------------------------------------------
    def configure_optimizers(self):
        if self.optimizer == 'SGD':
            optimizer = torch.optim.SGD(self.parameters(), lr=self.learning_rate, weight_decay=1e-4, momentum=0.9)
        else:
            optimizer = torch.optim.Adam(self.parameters(), eps=1e-7, lr=self.learning_rate)
        mode = 'min' if (self.metric.lower() in ('pk', 'wd') or not self.s_th) else 'max'
        monitor = 'val_loss' if self.validation else 'training_loss'
        scheduler = {'scheduler': torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode, factor=.8, patience=10),
                     'monitor': monitor}
        return {'optimizer': optimizer, 'lr_scheduler': scheduler}
