"""C-ABI checks that need no GPU: libkbj.so loads, exports every symbol include/kbj.h declares, struct sizes match the
Python mirrors, and the library refuses to run without a HIP device (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from kbot_joystick_amd.spec import compiler, layout as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from kbot_joystick_amd.host import binding
    if not os.path.exists(binding.LIB_PATH):
        binding.build_library()
    return binding.load_library()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "kbj.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(kbj_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    from kbot_joystick_amd.host import binding
    assert declared == set(binding.SIGNATURES), declared ^ set(binding.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_sizes(lib):
    assert lib.kbj_sizeof_model() == ctypes.sizeof(L.Model)
    assert lib.kbj_sizeof_config() == ctypes.sizeof(L.Config)
    from kbot_joystick_amd.host import binding
    assert lib.kbj_sizeof_traj() == ctypes.sizeof(binding.Traj) and lib.kbj_sizeof_carry() == ctypes.sizeof(binding.Carry)


def test_param_counts_match_survey(lib):
    # H=256 -> 1,077,800 + 1,172,737; H=128 -> 276,776 + 324,225
    for H, a, c in ((256, 1077800, 1172737), (128, 276776, 324225)):
        cfg = L.default_config(hidden_size=H)
        assert lib.kbj_actor_param_count(ctypes.byref(cfg)) == a
        assert lib.kbj_param_count(ctypes.byref(cfg)) == a + c
        assert L.param_count(H) == (a, c)


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from kbot_joystick_amd.host import binding
    with pytest.raises(binding.KbjError, match="no HIP device"):
        binding.Context(compiler.load_model("kbot-headless"), L.default_config(num_envs=4, batch_size=4))


def test_bad_blob_rejected(lib):
    h = ctypes.c_void_p()
    cfg = L.default_config(num_envs=4, batch_size=4)
    assert lib.kbj_create(ctypes.byref(h), b"\0" * 10, 10, ctypes.byref(cfg), 0, None) != 0
    assert b"wrong size" in lib.kbj_last_error(None)
    m = compiler.load_model("kbot-headless")
    m.body_parent[5] = 1          # break the topology the kernels are specialised on
    blob = ctypes.string_at(ctypes.addressof(m), ctypes.sizeof(m))
    assert lib.kbj_create(ctypes.byref(h), blob, len(blob), ctypes.byref(cfg), 0, None) != 0
    assert b"topology" in lib.kbj_last_error(None)


def test_task_config_guards():
    from kbot_joystick_amd.host.task import HumanoidWalkingTaskConfig, launch_config
    d = HumanoidWalkingTaskConfig().to_kbj(4096)          # dataclass defaults enable the mirror losses (train.py:115-122)
    assert (d.actor_mirror_loss_scale, abs(d.critic_mirror_loss_scale - 0.01) < 1e-9) == (1.0, True)
    z = launch_config().to_kbj(4096)                      # the launch block switches them off (train.py:1771-1772)
    assert (z.actor_mirror_loss_scale, z.critic_mirror_loss_scale) == (0.0, 0.0)
    with pytest.raises(ValueError):
        launch_config(batch_size=500).to_kbj(4096)
    c = launch_config().to_kbj(4096)
    assert (c.rollout_len, c.substeps, c.hidden_size, c.batch_size, c.num_passes) == (100, 5, 256, 512, 3)
    assert abs(c.lpf_alpha - 0.02 / (0.02 + 1 / (2 * 3.141592653589793 * 10))) < 1e-7


def test_cosine_decay_schedule():
    """optax.cosine_decay_schedule restated on the host (train.py:1067-1072)."""
    from kbot_joystick_amd.host.task import cosine_decay_lr, launch_config
    c = launch_config(use_lr_decay=True, lr_decay_steps=1000, lr_final_multiplier=0.01, learning_rate=5e-4)
    assert abs(cosine_decay_lr(c, 0) - 5e-4) < 1e-12
    assert abs(cosine_decay_lr(c, 500) - 5e-4 * (0.99 * 0.5 + 0.01)) < 1e-12
    assert abs(cosine_decay_lr(c, 1000) - 5e-6) < 1e-12 and abs(cosine_decay_lr(c, 5000) - 5e-6) < 1e-12
    c.to_kbj(4096)                                               # adamw + schedule (train.py:1076-1077)
    z = launch_config(use_lr_decay=True, adam_weight_decay=0.0, reproduce_reference_lr_sign=True).to_kbj(4096)   # scale_by_adam + scale_by_schedule (train.py:1074-1075): served behind the opt-in
    assert z.weight_decay == 0.0


def test_check_config_refuses_operands_of_two_gib():
    """The GEMM / recurrence tiles are fetched with 32-bit byte offsets (kbj_gemm.h load_tile, kbj_lstm_seq.h SeqTile::load): a stash
    array of >= 2 GiB would be read as zeros beyond the limit. The guard's arithmetic, on the host: T x B x max(4 H, 476) x 4 bytes and
    N x max(4 H, 476) x 4 bytes must stay below 2^31."""
    from kbot_joystick_amd.host import binding
    ok = lambda **kw: binding.check_config(L.default_config(**kw))
    assert ok(num_envs=8192, batch_size=512, rollout_len=100, hidden_size=256) == ""
    assert ok(num_envs=65536, batch_size=512, rollout_len=100, hidden_size=256) == ""
    # 1024 x 512 x 1024 floats = 2^31 bytes exactly: refused; one step shorter: served
    assert "2 GiB" in ok(num_envs=8192, batch_size=512, rollout_len=1024, hidden_size=256)
    assert ok(num_envs=8192, batch_size=512, rollout_len=1023, hidden_size=256) == ""
    # hidden 64: the 476-float critic rows are the wider operand (T x B x 476 x 4)
    assert "2 GiB" in ok(num_envs=8192, batch_size=4096, rollout_len=276, hidden_size=64)
    assert ok(num_envs=8192, batch_size=4096, rollout_len=275, hidden_size=64) == ""
    # one control step's rows: N x 1024 x 4 bytes
    assert "num_envs" in ok(num_envs=524288, batch_size=512, rollout_len=10, hidden_size=256)
    assert ok(num_envs=64, batch_size=64, hidden_size=320) == "" and ok(num_envs=64, batch_size=64, hidden_size=512) == ""   # the wide schedule
    assert "hidden_size" in ok(num_envs=64, batch_size=64, hidden_size=513)
