import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "relative-entropy-coding_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "both_suites: host-only test that runs in the CPU suite (-m 'not gpu') AND, as a "
                                       "second item marked gpu, in the driver's -m gpu run on the GPU box")


@pytest.fixture
def suite(request):
    """'cpu' or 'gpubox' -- which of the two items of a both_suites test this is."""
    return getattr(request, "param", "cpu")


def pytest_generate_tests(metafunc):
    # The golden-vector tests of the wire format, the C-ABI symbol check and the importance-sampler tests need no GPU, but
    # the driver only records what `-m gpu` ran on the GPU box: give each of them a second, gpu-marked item.
    if metafunc.definition.get_closest_marker("both_suites") and "suite" in metafunc.fixturenames:
        metafunc.parametrize("suite", ["cpu", pytest.param("gpubox", marks=pytest.mark.gpu)], indirect=True)


def golden_files(kind=None):
    out = []
    for f in sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))):
        g = np.load(f)
        if kind is None or ("kind" in g.files and str(g["kind"]) == kind):
            out.append(f)
    return out


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def engine():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import irec
    return irec.get_engine()
