"""kbot-joystick_amd — MI355X-native rollout + PPO hot path for the K-Bot joystick task.

Sub-packages: `spec` (model compiler + data layouts), `host` (ctypes binding to the C ABI in
include/kbj.h and the ksim-shaped Task facade), `csrc` (HIP kernels, built by __graft_entry__.build()).
"""
__version__ = "0.1.0"
