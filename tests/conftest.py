import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """SCV_TEST_ORDER=reverse | shuffle:<seed>: run the collected tests in another order (round 6: the suite had always run in file order, in
    which an early DEVICE-mode call bound the engine to torch's stream and hid a race of the split-N scratch: profiles/r06_split_scratch_race.log)."""
    order = os.environ.get("SCV_TEST_ORDER", "")
    if order == "reverse":
        items.reverse()
    elif order.startswith("shuffle:"):
        import random
        random.Random(int(order.split(":", 1)[1])).shuffle(items)


@pytest.fixture(scope="session")
def golden():
    import json
    from tests.golden import gen
    with open(os.path.join(REPO, "tests", "golden", "golden_o1.json")) as f:
        g = json.load(f)
    for case in g["cases"]:
        gen.materialise(case)          # generator-backed cases: samples are rebuilt, not stored
    return g


@pytest.fixture(scope="session")
def oracle_engine():
    from tests._adapters import OracleEngine
    return OracleEngine()


@pytest.fixture(scope="session")
def hip_engine():
    """The product engine.  On the GPU box a missing library must FAIL, never skip or fall back."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked test running without a GPU"
    from o1_inference_scaling_laws_amd.engine import Engine
    eng = Engine(timing=True)
    yield eng
    eng.close()
