"""`run_mode=view` of the reference (`README.md:66-70`: "visualize training your model without using kscale-mujoco-viewer",
`train.py:1783-1784` render_track_body_id / render_length_seconds): roll the current policy out deterministically on a few envs of their
own context (the same calls `task.validate()` makes) and record the generalised positions per control step. There is no MuJoCo and no
window system on a training node, so the recording is written as

  * `<stem>.npz`  - qpos [F][K][27], body positions [F][K][nbody][3] and quaternions, foot capsule end points, command, reward, done
  * `<stem>.html` - a self-contained page (no dependencies) that plays the recording: side and front orthographic views that track
                    `render_track_body_id`, the skeleton (body -> parent links), the foot capsules, the ground / "sine" terrain profile,
                    a HUD with time, command and reward

The kinematics below are the host-side restatement of the model blob's body tree (`include/kbj_model.h`), checked against the oracle's
`xpos` / `xquat` in `tests/test_host_cpu.py`; nothing here is on the hot path."""
from __future__ import annotations

import json
import math
from typing import Optional

import numpy as np


def _qmul(a, b):
    aw, ax, ay, az = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bw, bx, by, bz = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], axis=-1)


def _qrot(q, v):
    """v rotated by the unit quaternion q (w, x, y, z); broadcasts over leading axes."""
    w, u = q[..., :1], q[..., 1:]
    t = 2.0 * np.cross(u, v)
    return v + w * t + np.cross(u, t)


def forward_kinematics(model, qpos: np.ndarray):
    """World positions [..., nbody, 3] and orientations [..., nbody, 4] of every body for generalised positions qpos [..., nq]
    (free joint of the base first: position, quaternion w x y z; then one hinge angle per dof in dof order)."""
    qpos = np.asarray(qpos, np.float64)
    lead = qpos.shape[:-1]
    nb = int(model.nbody)
    xpos = np.zeros(lead + (nb, 3))
    xquat = np.zeros(lead + (nb, 4))
    xquat[..., 0, 0] = 1.0
    for b in range(1, nb):
        p = int(model.body_parent[b])
        bp = np.array(model.body_pos[b][:], np.float64)
        bq = np.array(model.body_quat[b][:], np.float64)
        num, adr = int(model.body_dofnum[b]), int(model.body_dofadr[b])
        if num == 6:       # free joint: qpos holds the world pose
            q = qpos[..., 3:7]
            xpos[..., b, :] = qpos[..., 0:3]
            xquat[..., b, :] = q / np.linalg.norm(q, axis=-1, keepdims=True)
            continue
        pos = xpos[..., p, :] + _qrot(xquat[..., p, :], bp)
        quat = _qmul(xquat[..., p, :], np.broadcast_to(bq, lead + (4,)))
        if num == 1:       # hinge about jnt_axis through the body origin; qpos index = dof index + 1 (the free joint has 7 positions, 6 dofs)
            ang = qpos[..., adr + 1]
            ax = np.array(model.jnt_axis[b][:], np.float64)
            jq = np.concatenate([np.cos(0.5 * ang)[..., None], np.sin(0.5 * ang)[..., None] * ax], axis=-1)
            quat = _qmul(quat, jq)
        xpos[..., b, :] = pos
        xquat[..., b, :] = quat / np.linalg.norm(quat, axis=-1, keepdims=True)
    return xpos, xquat


def capsule_segments(model, xpos, xquat, ep: Optional[np.ndarray] = None):
    """End points [..., ncap, 2, 3] and radii [..., ncap] of the foot collision capsules; `ep` [..., EP_SIZE] are the envs' randomised
    parameter records (kbj_env_get_state) - without them the model's nominal capsules are drawn."""
    from ..spec import layout as L
    nc = int(model.ncap)
    seg = np.zeros(xpos.shape[:-2] + (nc, 2, 3))
    rad = np.zeros(xpos.shape[:-2] + (nc,))
    for c in range(nc):
        b = int(model.cap_body[c])
        if ep is not None:
            cp = ep[..., L.EP["CAP_POS"] + 3 * c: L.EP["CAP_POS"] + 3 * c + 3].astype(np.float64)
            hl = ep[..., L.EP["CAP_HALF"] + c].astype(np.float64)[..., None]
            rad[..., c] = ep[..., L.EP["CAP_RAD"] + c]
        else:
            cp = np.array(model.cap_pos[c][:], np.float64)
            hl = float(model.cap_halflen[c])
            rad[..., c] = float(model.cap_radius[c])
        ax = np.array(model.cap_axis[c][:], np.float64)
        centre = xpos[..., b, :] + _qrot(xquat[..., b, :], np.broadcast_to(cp, xpos.shape[:-2] + (3,)))
        d = _qrot(xquat[..., b, :], np.broadcast_to(ax, xpos.shape[:-2] + (3,))) * hl
        seg[..., c, 0, :], seg[..., c, 1, :] = centre - d, centre + d
    return seg, rad


_PAGE = """<!doctype html><html><head><meta charset="utf-8"><title>__TITLE__</title>
<style>body{background:#111;color:#ddd;font:13px monospace;margin:8px}canvas{background:#1b1b1f;border:1px solid #333}button,select{font:13px monospace}</style></head>
<body><div>__TITLE__ &nbsp; <button id="pp">pause</button> env <select id="env"></select> speed <select id="sp"><option>0.25</option><option>0.5</option><option selected>1</option><option>2</option></select>
<input id="sl" type="range" min="0" max="0" value="0" style="width:40%"></div>
<canvas id="c" width="1200" height="520"></canvas><pre id="hud"></pre>
<script>
const R = __DATA__;
const cv = document.getElementById('c'), g = cv.getContext('2d'), hud = document.getElementById('hud');
const sel = document.getElementById('env'), sl = document.getElementById('sl'), pp = document.getElementById('pp'), sp = document.getElementById('sp');
for (let k = 0; k < R.K; k++) { const o = document.createElement('option'); o.text = k; sel.add(o); }
sl.max = R.F - 1;
let f = 0, playing = true, acc = 0, last = null;
pp.onclick = () => { playing = !playing; pp.textContent = playing ? 'pause' : 'play'; };
sl.oninput = () => { f = +sl.value; };
function ground(x, y) { return R.terrain_amp * Math.sin(2 * Math.PI * x / R.terrain_wavelength) * Math.sin(2 * Math.PI * y / R.terrain_wavelength); }
function panel(x0, w, h, k, a, b, label) {            // orthographic view on world axes (a, 2), depth axis b, tracking the chosen body
  const P = R.xpos[f][k], T = P[R.track], s = 230, cx = x0 + w / 2, cy = h * 0.72;
  const X = p => cx + (p[a] - T[a]) * s, Y = p => cy - (p[2] - 0.35) * s;
  g.save(); g.beginPath(); g.rect(x0, 0, w, h); g.clip();
  g.strokeStyle = '#2a2a30'; g.lineWidth = 1;
  const lo = Math.floor(T[a] - w / s), hi = Math.ceil(T[a] + w / s);
  for (let m = lo; m <= hi; m += 0.5) { g.beginPath(); g.moveTo(cx + (m - T[a]) * s, 0); g.lineTo(cx + (m - T[a]) * s, h); g.stroke(); }
  g.strokeStyle = '#6a6'; g.lineWidth = 2; g.beginPath();
  for (let i = 0; i <= 120; i++) { const u = T[a] - w / (2 * s) + i * w / (120 * s); const q = [0, 0, 0]; q[a] = u; q[b] = T[b];
    const z = R.terrain_amp ? ground(q[0], q[1]) : 0; const px = cx + (u - T[a]) * s, py = cy - (z - 0.35) * s; if (i) g.lineTo(px, py); else g.moveTo(px, py); }
  g.stroke();
  g.lineCap = 'round';
  for (let c = 0; c < R.ncap; c++) { const S = R.caps[f][k][c]; g.strokeStyle = 'rgba(230,160,60,0.55)'; g.lineWidth = Math.max(2, 2 * R.cap_radius[k][c] * s);
    g.beginPath(); g.moveTo(X(S[0]), Y(S[0])); g.lineTo(X(S[1]), Y(S[1])); g.stroke(); }
  g.lineWidth = 3;
  for (let i = 2; i < R.nbody; i++) { const p = R.parent[i]; if (p < 1) continue; g.strokeStyle = R.side[i] > 0 ? '#6cf' : (R.side[i] < 0 ? '#f7a' : '#ddd');
    g.beginPath(); g.moveTo(X(P[p]), Y(P[p])); g.lineTo(X(P[i]), Y(P[i])); g.stroke(); }
  g.fillStyle = '#fff'; for (let i = 1; i < R.nbody; i++) { g.beginPath(); g.arc(X(P[i]), Y(P[i]), 2.5, 0, 6.3); g.fill(); }
  g.fillStyle = '#888'; g.fillText(label, x0 + 8, 14); g.restore();
}
function draw() {
  const k = sel.selectedIndex < 0 ? 0 : sel.selectedIndex;
  g.clearRect(0, 0, cv.width, cv.height);
  panel(0, 800, 520, k, 0, 1, 'side view (x, z)'); panel(810, 390, 520, k, 1, 0, 'front view (y, z)');
  const c = R.cmd[f][k];
  hud.textContent = 't = ' + (f * R.dt).toFixed(2) + ' s   frame ' + f + '/' + (R.F - 1) + '   command vx ' + c[0].toFixed(2) + ' vy ' + c[1].toFixed(2) + ' wz ' + c[2].toFixed(2) +
    '   base z ' + R.xpos[f][k][R.track][2].toFixed(3) + '   reward ' + (f < R.reward.length ? R.reward[f][k].toFixed(3) : '-') + (f < R.done.length && R.done[f][k] ? '   [episode end: ' + (R.done[f][k] < 0 ? 'failure' : 'time limit') + ']' : '');
  sl.value = f;
}
function tick(ts) { if (last === null) last = ts; if (playing) { acc += (ts - last) / 1000 * (+sp.value); while (acc >= R.dt) { acc -= R.dt; f = (f + 1) % R.F; } } last = ts; draw(); requestAnimationFrame(tick); }
requestAnimationFrame(tick);
</script></body></html>
"""


class Recording:
    """What `HumanoidWalkingTask.view()` returns: numpy arrays of one deterministic rollout (F = T + 1 frames, K envs)."""

    def __init__(self, model, qpos, ep, cmd, reward, done, dt, track_body, terrain=(0.0, 1.0)):
        self.qpos = np.asarray(qpos, np.float32)                  # [F][K][nq]
        self.xpos, self.xquat = forward_kinematics(model, self.qpos)
        self.caps, self.cap_radius = capsule_segments(model, self.xpos, self.xquat, None if ep is None else np.asarray(ep)[None])
        self.cmd, self.reward, self.done = np.asarray(cmd, np.float32), np.asarray(reward, np.float32), np.asarray(done, np.float32)
        self.dt, self.track_body = float(dt), int(track_body)
        self.terrain_amp, self.terrain_wavelength = float(terrain[0]), float(terrain[1]) or 1.0
        self.parent = [int(model.body_parent[b]) for b in range(int(model.nbody))]
        # colour by side of the body's rest position relative to the base (left +y / right -y)
        rest, _ = forward_kinematics(model, np.array(model.qpos0[:int(model.nq)], np.float64))
        self.side = [0 if abs(rest[b, 1] - rest[1, 1]) < 0.02 else (1 if rest[b, 1] > rest[1, 1] else -1) for b in range(int(model.nbody))]

    def save_npz(self, path: str):
        np.savez_compressed(path, qpos=self.qpos, xpos=self.xpos.astype(np.float32), xquat=self.xquat.astype(np.float32), caps=self.caps.astype(np.float32),
                            cap_radius=self.cap_radius.astype(np.float32), cmd=self.cmd, reward=self.reward, done=self.done, dt=self.dt,
                            parent=np.array(self.parent, np.int32))

    def save_html(self, path: str, title: str = "kbot-joystick: policy rollout"):
        r3 = lambda a: np.round(np.asarray(a, np.float64), 4).tolist()
        F, K = self.qpos.shape[:2]
        track = self.track_body if 0 < self.track_body < len(self.parent) else 1     # body 0 is the world: track the base instead
        rad = np.broadcast_to(self.cap_radius[0] if self.cap_radius.ndim == 3 else self.cap_radius, (K, self.caps.shape[2]))
        data = dict(F=F, K=K, nbody=len(self.parent), ncap=int(self.caps.shape[2]), dt=self.dt, track=track, parent=self.parent, side=self.side,
                    xpos=r3(self.xpos), caps=r3(self.caps), cap_radius=r3(rad), cmd=r3(self.cmd[..., :3]), reward=r3(self.reward), done=r3(self.done),
                    terrain_amp=self.terrain_amp, terrain_wavelength=self.terrain_wavelength)
        page = _PAGE.replace("__TITLE__", title).replace("__DATA__", json.dumps(data, separators=(",", ":")))
        with open(path, "w") as fh:
            fh.write(page)
