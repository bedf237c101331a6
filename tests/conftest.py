import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The shared libraries are git-ignored build products: (re)build them in-tree when they are missing or older than
    # their sources (hipcc cross-compiles gfx950 without a GPU; a few tens of seconds the first time).
    from bifromq_amd import _lib
    _lib.build()
    from oracle import oracle as O
    O.build()
