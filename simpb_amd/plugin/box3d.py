"""Index constants of the 11-d undecoded box state and of the decoded box / quality vectors
(core/box3d.py:1-3 of the reference defines the same names; the layout itself is fixed by the
checkpoints: x, y, z, log w, log l, log h, sin yaw, cos yaw, vx, vy, vz)."""
_STATE = ("X", "Y", "Z", "W", "L", "H", "SIN_YAW", "COS_YAW", "VX", "VY", "VZ")
globals().update({name: i for i, name in enumerate(_STATE)})
CNS, YNS = 0, 1   # quality vector: centerness, yawness
YAW = 6           # decoded box: yaw replaces (sin, cos)
__all__ = list(_STATE) + ["CNS", "YNS", "YAW"]
