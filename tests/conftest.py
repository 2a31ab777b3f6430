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
    from oracle import rm_oracle
    rm_oracle.lib()
    return rm_oracle


@pytest.fixture()
def default_setup():
    """(material, Raymarching script, props dict) for the sample scene at a given size."""
    from oracle import rm_oracle as O
    from raymarching_amd import Material, Raymarching, scenes

    made = []

    def make(w, h, camera=None):
        mat = Material()
        rm = Raymarching(camera or scenes.default_camera(w, h), mat, scenes.default_transforms())
        rm.bind()
        mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
        made.append(mat)
        return mat, rm, O.props_from_material(mat)

    yield make
    for m in made:
        m.close()
