import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.dog_oracle import Oracle, build
    build()
    return Oracle()


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "dog_cases.npz")
    z = np.load(path, allow_pickle=False)
    cases = []
    for name in z["names"]:
        name = str(name)
        tw, wh, ww, darker, g1, g2, fill, l = (int(v) for v in z[name + "/params"])
        cases.append(dict(name=name, frame=z[name + "/frame"], tw=tw, ws=(wh, ww), darker=bool(darker),
                          guess=(g1, g2), fill=fill, l=l, ij=tuple(int(v) for v in z[name + "/ij"]),
                          resp=z[name + "/resp"]))
    return cases
