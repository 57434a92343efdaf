"""MI355X-native sentence-boundary tagger: drop-in for the tagger hot path of Ighina/MultimodalTopicSegmentation.

Importing this package loads libmts_hip.so (hand-written HIP kernels for gfx950) and fails loudly if it is missing;
there is no CPU or PyTorch fallback for the tagger arithmetic.
"""
from . import _lib  # noqa: F401  (raises ImportError when the HIP library is not built)
from .encoder_dataset import AudioPortionDataset, AudioPortionDatasetInference  # noqa: F401
from .datasets import load_dataset_for_inference, load_dataset_from_precomputed, second_input_of  # noqa: F401
from .lightning_model import TextSegmenter  # noqa: F401
from .prefetch import DevicePrefetcher  # noqa: F401
from .rnn_taggers import BiLSTM, BiLSTMLateFusion, BiRnnCrf  # noqa: F401
from .taggers import RestrictedTransformerEncoderLayer, Transformer_segmenter  # noqa: F401

__all__ = ['TextSegmenter', 'Transformer_segmenter', 'BiLSTM', 'BiLSTMLateFusion', 'BiRnnCrf', 'AudioPortionDataset',
           'AudioPortionDatasetInference', 'RestrictedTransformerEncoderLayer', 'DevicePrefetcher']
