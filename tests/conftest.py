import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# the licensed MANO files / hamer mean parameters are not here: the tests opt in to the synthetic
# stand-ins (hands_amd.mano.build_mano_asset raises without this, like the reference without $MANO_DIR)
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (run on the MI355X box)")
    # The oracle runs ATen's CPU kernels, whose summation order depends on the thread count: the vertices of a
    # handoccnet_light forward move by 2-5e-7 m between 1 and 4+ threads (hands_light: 1e-7).  The golden fixtures were generated
    # by the reference with 8 threads (tests/golden/_ref_shims.py): every test uses the same count, on any box.
    import torch
    torch.set_num_threads(min(8, os.cpu_count() or 1))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def recipe_model():
    """hands_amd.HandsLight with the deterministic weight recipe (CPU parameters)."""
    import hands_amd
    m = hands_amd.HandsLight()
    hands_amd.apply_recipe(m)
    m.eval()
    return m


@pytest.fixture(scope="session")
def recipe_sd(recipe_model):
    return {k: v.detach().cpu().clone() for k, v in recipe_model.state_dict().items()}
