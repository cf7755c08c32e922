import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
DATA = os.path.join(ROOT, "tests", "golden", "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "survey_8c.json")) as f:
        return json.load(f)


def load_problem(name, golden):
    """Problem dict of one of the golden problems, read with the test-only numpy reader."""
    from sba_text import read_problem
    p = golden["problems"][name]
    return read_problem(os.path.join(DATA, p["cams"]), os.path.join(DATA, p["pts"]))


@pytest.fixture(scope="session")
def problems(golden):
    return {name: load_problem(name, golden) for name in golden["problems"]}
