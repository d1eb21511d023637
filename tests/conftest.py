import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


@pytest.fixture
def pinned_tiles(dev):
    """The convolution tile model evaluated at ONE batch size for every launch of the test (hd_conv_nominal_batch(8): rounds 2-5's rule).
    Since round 6 the model sees each launch's own batch, so a batched launch may run other tiles than the same images launched alone;
    tiles split K differently and agree to fp16 rounding only.  Tests whose statement is "the orchestration / the tile -> block mapping
    adds nothing of its own: bit-identical" pin the tiles and keep the bitwise assertion; the per-launch rule's tolerance is asserted where
    the test says so."""
    from hallucidet_amd import _abi
    lib = _abi.load()
    lib.hd_conv_nominal_batch(8)
    try:
        yield lib
    finally:
        lib.hd_conv_nominal_batch(0)
