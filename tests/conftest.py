import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


# GPU tests written after the round's GPU budget was spent have never run on hardware.  They are skipped unless MVE_RUN_PENDING=1
# so that an unexpected device fault in unverified code cannot take the verified suite down with it (`pytest -x`, or a fault that
# kills the process).  First thing next round: MVE_RUN_PENDING=1 python -m pytest tests -m gpu (tools/gpu_pending.sh runs them one file per process); then drop the mark.
pending_first_gpu_run = pytest.mark.skipif(os.environ.get('MVE_RUN_PENDING') != '1',
                                           reason='written after the round-1 GPU budget was exhausted: not yet run on an MI355X '
                                                  '(set MVE_RUN_PENDING=1 to run)')


@pytest.fixture(scope='session')
def lib():
    """The C-ABI library, built in-tree if needed (hipcc cross-compiles without a GPU)."""
    from mvedit_amd import build
    build.build()
    from mvedit_amd import _lib
    return _lib
