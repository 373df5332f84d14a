import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    # a native crash (SIGSEGV / SIGABRT) leaves its Python stacks in a file of its own: the tail of a long pytest log gets
    # cut by the harness that runs the GPU suite
    import faulthandler
    out_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        config._gsx_fault_file = open(os.path.join(out_dir, "faulthandler.log"), "w")
        faulthandler.enable(file=config._gsx_fault_file, all_threads=True)
    except OSError:
        faulthandler.enable(all_threads=True)
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test")


@pytest.fixture(scope="session")
def oracle32():
    from oracle.oracle import Oracle
    import numpy as np
    return Oracle(np.float32)


@pytest.fixture(scope="session")
def oracle64():
    from oracle.oracle import Oracle
    import numpy as np
    return Oracle(np.float64)
