"""Task constants of the K-Bot joystick task, in neural-network output order.

Values restate the tables of the reference task definition (train.py:22-70: neutral pose in
degrees and joint limits in radians); the order is the MJCF joint order (left leg, right leg,
right arm, left arm), which is also the actuator order (robot.mjcf <actuator> block).
"""
from __future__ import annotations

import math

JOINT_NAMES = (
    "dof_left_hip_pitch_04", "dof_left_hip_roll_03", "dof_left_hip_yaw_03", "dof_left_knee_04", "dof_left_ankle_02",
    "dof_right_hip_pitch_04", "dof_right_hip_roll_03", "dof_right_hip_yaw_03", "dof_right_knee_04", "dof_right_ankle_02",
    "dof_right_shoulder_pitch_03", "dof_right_shoulder_roll_03", "dof_right_shoulder_yaw_02", "dof_right_elbow_02",
    "dof_right_wrist_00",
    "dof_left_shoulder_pitch_03", "dof_left_shoulder_roll_03", "dof_left_shoulder_yaw_02", "dof_left_elbow_02",
    "dof_left_wrist_00",
)

# neutral pose, degrees (train.py:24-45)
_BIAS_DEG = (20, 0, 0, 50, -30, -20, 0, 0, -50, 30, 0, -10, 0, 90, 0, 0, 10, 0, -90, 0)
JOINT_BIASES = tuple(math.radians(float(d)) for d in _BIAS_DEG)

# observation-normalisation limits, radians (train.py:47-68; ankles are deliberately tighter than the MJCF range)
JOINT_LIMITS = (
    (-1.047198, 2.216568), (-0.20944, 2.268928), (-1.570796, 1.570796), (0.0, 2.70526), (-1.134464, 0.261799),
    (-2.216568, 1.047198), (-2.268928, 0.20944), (-1.570796, 1.570796), (-2.70526, 0.0), (-0.261799, 1.134464),
    (-3.490658, 1.047198), (-1.658063, 0.436332), (-1.671886, 1.671886), (0.0, 2.478368), (-1.37881, 1.37881),
    (-1.047198, 3.490658), (-0.436332, 1.658063), (-1.671886, 1.671886), (-2.478368, 0.0), (-1.37881, 1.37881),
)

# names the task wiring looks up (train.py:1120-1123, 1186-1188, 1193, 1179-1182)
BASE_BODY = "base"
FOOT_LEFT_BODY = "LFootBushing_GPF_1517_12"
FOOT_RIGHT_BODY = "RFootBushing_GPF_1517_12"
IMU_SITE = "imu_site"
FOOT_SITES = ("left_foot", "right_foot")
COLLISION_CAPSULES = (
    "LFootBushing_GPF_1517_12_collision_capsule_0",
    "LFootBushing_GPF_1517_12_collision_capsule_1",
    "RFootBushing_GPF_1517_12_collision_capsule_0",
    "RFootBushing_GPF_1517_12_collision_capsule_1",
)

# command vector names (convert.py:48-65 deployment contract)
COMMAND_NAMES = (
    "xvel", "yvel", "yawrate", "baseheight", "baseroll", "basepitch",
    "rshoulderpitch", "rshoulderroll", "rshoulderyaw", "relbowpitch", "rwristroll",
    "lshoulderpitch", "lshoulderroll", "lshoulderyaw", "lelbowpitch", "lwristroll",
)

# argument order of the deployed step function and what it returns (convert.py:84-119): the first five are concatenated into the
# 65-float actor row exactly as run_actor packs it, `carry` is the flat (depth, 2, H) LSTM carry followed by the 20 low-pass floats
STEP_FN_INPUTS = ("joint_angles", "joint_angular_velocities", "projected_gravity", "gyroscope", "command", "carry")

REWARD_NAMES = (
    "linvel", "angvel", "roll_pitch", "base_height", "arm_pos", "single_contact", "no_contact_p",
    "feet_airtime", "feet_orient", "com_distance", "base_accel", "torque",
)
# train.py:1225-1256
REWARD_SCALES = (0.2, 0.1, 0.2, 0.2, 0.2, 0.1, 0.1, 1.5, 0.1, 0.05, 0.1, 0.1)
