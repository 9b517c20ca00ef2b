"""Device buffers of the hot path, owned by torch, handed to libkbj.so as raw pointers."""
from __future__ import annotations

import torch

from ..spec import layout as L
from . import binding as B


class CarryBuffers:
    """Model carry (train.py:1049-1055, 1526-1543): actor/critic LSTM (h, c) per layer + low-pass filter state, and the
    same again for the mirror branches of the aux losses when they are enabled."""

    def __init__(self, N: int, H: int, depth: int, device, mirror: bool = False):
        self.actor_hc = torch.zeros(depth, 2, N, H, device=device)
        self.critic_hc = torch.zeros(depth, 2, N, H, device=device)
        self.lpf = torch.zeros(N, L.NU, device=device)
        self.mirror = mirror
        if mirror:
            self.actor_mirror_hc = torch.zeros(depth, 2, N, H, device=device)
            self.critic_mirror_hc = torch.zeros(depth, 2, N, H, device=device)
            self.lpf_mirror = torch.zeros(N, L.NU, device=device)
            self.c = B.Carry(self.actor_hc.data_ptr(), self.critic_hc.data_ptr(), self.lpf.data_ptr(), self.actor_mirror_hc.data_ptr(),
                             self.critic_mirror_hc.data_ptr(), self.lpf_mirror.data_ptr())
        else:
            self.c = B.Carry(self.actor_hc.data_ptr(), self.critic_hc.data_ptr(), self.lpf.data_ptr(), None, None, None)

    def zero_(self):
        self.actor_hc.zero_(); self.critic_hc.zero_(); self.lpf.zero_()
        if self.mirror:
            self.actor_mirror_hc.zero_(); self.critic_mirror_hc.zero_(); self.lpf_mirror.zero_()


class TrajBuffers:
    """One rollout's worth of trajectory arrays ([T(+1)][N][dim], time-major)."""

    def __init__(self, T: int, N: int, H: int, depth: int, device, mirror: bool = False, reward_comps: bool = False,
                 ld_actor: int = L.LD_ACTOR, ld_critic: int = L.LD_CRITIC, record_state: bool = False):
        """ld_actor / ld_critic: row strides of the context's observation rows (layout.obs_widths(kbj_config): 68 / 476 plus user columns)."""
        self.T, self.N = T, N
        z = lambda *s: torch.zeros(*s, device=device)
        self.actor_obs = z(T + 1, N, ld_actor)
        self.critic_obs = z(T + 1, N, ld_critic)
        self.aux = z(T + 1, N, L.AUX["SIZE"])
        self.action = z(T, N, L.NU)
        self.logp = z(T, N)
        self.value = z(T, N)
        self.reward = z(T, N)
        self.carry0_actor_hc = z(depth, 2, N, H)
        self.carry0_critic_hc = z(depth, 2, N, H)
        self.carry0_lpf = z(N, L.NU)
        self.adv = z(T, N)
        self.target = z(T, N)
        self.comps = z(T, N, 12) if reward_comps else None   # unscaled reward terms (train.py:1224-1256 order)
        # per-step state record (kbj_model.h KBJ_QSTATE_*): what host/trajectory.Trajectory turns into ksim's qpos / qvel / xpos / xquat
        self.qstate = z(T, N, L.QSTATE["SIZE"]) if record_state else None
        mptr = [None, None, None]
        if mirror:
            self.carry0_actor_mirror_hc = z(depth, 2, N, H)
            self.carry0_critic_mirror_hc = z(depth, 2, N, H)
            self.carry0_lpf_mirror = z(N, L.NU)
            mptr = [self.carry0_actor_mirror_hc.data_ptr(), self.carry0_critic_mirror_hc.data_ptr(), self.carry0_lpf_mirror.data_ptr()]
        self.c = B.Traj(T, N, self.actor_obs.data_ptr(), self.critic_obs.data_ptr(), self.aux.data_ptr(), self.action.data_ptr(),
                        self.logp.data_ptr(), self.value.data_ptr(), self.reward.data_ptr(), self.carry0_actor_hc.data_ptr(),
                        self.carry0_critic_hc.data_ptr(), self.carry0_lpf.data_ptr(), *mptr, self.comps.data_ptr() if reward_comps else None,
                        self.qstate.data_ptr() if record_state else None)

    @property
    def done(self) -> torch.Tensor:
        return self.aux[: self.T, :, L.AUX["DONE"]]
