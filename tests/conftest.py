import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # The CPU oracle is many small batched matmuls: on a GPU box torch sees every hardware thread of the host (256) while the job
    # owns a share of about 16 cores -- oversubscribed, one oracle evaluation went from seconds to minutes (a 197 s test call on one
    # box of the pool).  Cap the thread pool at the share.
    import torch
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when no device is present, e.g. when a plain `pytest tests/`
    # is run in the CPU container.  On the GPU box the product path itself fails loudly if the HIP
    # library is missing.
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
