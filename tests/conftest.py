import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(params=[1, 0], ids=["bf16x6", "f32mfma"])
def gemm_core(request):
    """Run a whole-model GPU test on both arithmetic cores of the GEMM (dvae_gemm_x6): six bf16 MFMAs on operands
    split exactly into three bf16 pieces (the default), and v_mfma_f32_32x32x2_f32."""
    from dvae_amd._lib import load
    lib = load()
    lib.dvae_gemm_x6(request.param)
    yield request.param
    lib.dvae_gemm_x6(int(os.environ.get("DVAE_GEMM_X6", "1")))
