"""Model compiler: the committed blobs are what the compiler produces from the reference's robot assets."""
import ctypes
import os

import numpy as np
import pytest

from kbot_joystick_amd.spec import compiler, constants as K, layout as L

REF = "/root/reference/robot"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference assets are only present in the build container")
@pytest.mark.parametrize("name", ["kbot", "kbot-headless"])
def test_committed_blob_matches_compiler(name):
    m = compiler.compile_model(os.path.join(REF, name))
    assert compiler.model_to_bytes(m) == compiler.model_to_bytes(compiler.load_model(name))


def test_blob_topology_and_tables(model):
    assert (model.nbody, model.nq, model.nv, model.nu) == (24, 27, 26, 20)
    assert list(model.body_parent)[:4] == [0, 0, 1, 2] and model.body_parent[8] == 2 and model.body_parent[23] == 2
    assert np.allclose(list(model.joint_bias), K.JOINT_BIASES)
    assert list(model.cap_body) == [7, 7, 12, 12]
    # metadata.json gains
    assert model.kp[0] == 150.0 and abs(model.kd[0] - 24.722) < 1e-6 and model.tau_limit[4] == np.float32(11.9)
    # joint classes: robstride_04 hip pitch
    assert abs(model.dof_armature[6] - 0.04) < 1e-9 and abs(model.dof_frictionloss[6] - 0.2) < 1e-7 and model.act_range[0][1] == 120.0
    # collision class: solref (0.02, 0.8), solimp (0.98, 0.99, 0.1), friction 0.64
    assert np.allclose(list(model.contact_solref), [0.02, 0.8]) and np.allclose(list(model.contact_solimp)[:3], [0.98, 0.99, 0.1])
    assert abs(model.contact_mu - 0.64) < 1e-7


def test_blob_roundtrip(model):
    b = compiler.model_to_bytes(model)
    assert len(b) == ctypes.sizeof(L.Model)
    m2 = compiler.model_from_bytes(b)
    assert compiler.model_to_bytes(m2) == b
    with pytest.raises(ValueError):
        compiler.model_from_bytes(b[:-1])
