"""Synthetic inputs and procedural weights shared by the golden generator, the tests and bench.py.

Nothing here reads /root/reference: the scene follows SURVEY.md §8(d) (nuScenes-like 6-camera
ring, 900 anchors in a 55 m disc, streams 0.5 s apart) and the weights are a deterministic
function of the state_dict key, so both sides of a parity check can rebuild them without a
weight fixture.
"""
import math
import re
import zlib

import numpy as np
import torch

CAM_YAW_DEG = (0.0, -55.0, 55.0, 180.0, 110.0, -110.0)
CAM_FOCAL = (557.0, 557.0, 557.0, 356.0, 557.0, 557.0)
MEAN_LOG_WLH = (math.log(1.9), math.log(4.6), math.log(1.7))


def _rng(key, seed=0):
    return np.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0xFFFFFFFF)


def randn(key, shape, seed=0):
    """float32 standard normal, a pure function of (key, shape, seed)."""
    return _rng(key, seed).standard_normal(tuple(shape)).astype(np.float32)


def camera_rig(image_wh=(704, 256), height=1.5, forward_offset=0.5):
    """projection_mat f32[6,4,4] = K.E for the ring described in SURVEY.md §8(d).

    Ego frame: x forward, y left, z up. Camera frame: x right, y down, z forward.
    """
    w, h = image_wh
    sx = w / 704.0
    mats = []
    for yaw_deg, focal in zip(CAM_YAW_DEG, CAM_FOCAL):
        psi = math.radians(yaw_deg)
        fwd = np.array([math.cos(psi), math.sin(psi), 0.0])
        right = np.array([math.sin(psi), -math.cos(psi), 0.0])
        down = np.array([0.0, 0.0, -1.0])
        rot = np.stack([right, down, fwd])
        pos = np.array([0.0, 0.0, height]) + forward_offset * fwd
        ext = np.eye(4)
        ext[:3, :3] = rot
        ext[:3, 3] = -rot @ pos
        intr = np.eye(4)
        intr[0, 0] = intr[1, 1] = focal * sx
        intr[0, 2] = w / 2.0
        intr[1, 2] = h / 2.0
        mats.append(intr @ ext)
    return np.stack(mats).astype(np.float32)


def anchors(num_anchor=900, radius=55.0, seed=0):
    """[num_anchor, 11] = x,y,z,log w,log l,log h,sin,cos,vx,vy,vz (core/box3d.py:1 order)."""
    r = _rng("synth.anchors", seed)
    rad = radius * np.sqrt(r.uniform(0.02, 1.0, num_anchor))
    ang = r.uniform(-math.pi, math.pi, num_anchor)
    out = np.zeros((num_anchor, 11), np.float32)
    out[:, 0] = rad * np.cos(ang)
    out[:, 1] = rad * np.sin(ang)
    out[:, 2] = r.uniform(-2.0, 1.0, num_anchor)
    out[:, 3:6] = np.asarray(MEAN_LOG_WLH, np.float32) + 0.2 * r.standard_normal((num_anchor, 3))
    yaw = r.uniform(-math.pi, math.pi, num_anchor)
    out[:, 6] = np.sin(yaw)
    out[:, 7] = np.cos(yaw)
    return out


def ego_pose(t, speed=5.0, yaw_rate=0.05):
    """T_global f64[4,4] at time t: ego moving along a gentle arc."""
    th = yaw_rate * t
    pose = np.eye(4)
    pose[0, 0], pose[0, 1] = math.cos(th), -math.sin(th)
    pose[1, 0], pose[1, 1] = math.sin(th), math.cos(th)
    pose[0, 3] = speed * t
    pose[1, 3] = 0.3 * speed * t * th
    return pose


def frame_metas(bs, frame_idx, image_wh=(704, 256), device="cpu", dt=0.5, stream_offset=100.0, jump=None):
    """metas dict of one frame (keys of the test pipeline, config :349-358): stream b is a
    different time origin so streams are not identical copies. jump=(b, frame, seconds) inserts
    a time gap in stream b from that frame on (exercises the bank's max_time_interval mask)."""
    proj = torch.from_numpy(camera_rig(image_wh))[None].repeat(bs, 1, 1, 1)
    if bs > 1:  # small per-stream perturbation of the intrinsics so batch items differ
        for b in range(bs):
            proj[b, :, 0, :] *= 1.0 + 0.01 * b
    wh = torch.tensor([float(image_wh[0]), float(image_wh[1])]).view(1, 1, 2).repeat(bs, 6, 1)
    ts, img_metas = [], []
    for b in range(bs):
        t = stream_offset * b + dt * frame_idx
        if jump is not None and b == jump[0] and frame_idx >= jump[1]:
            t += jump[2]
        pose = ego_pose(t)
        ts.append(t)
        img_metas.append(
            dict(
                T_global=pose,
                T_global_inv=np.linalg.inv(pose),
                timestamp=t,
                aug_config=dict(resize=0.44, crop=(0, 140, 704, 396)),
            )
        )
    return dict(
        projection_mat=proj.to(device),
        image_wh=wh.to(device),
        timestamp=torch.tensor(ts, dtype=torch.float64).to(device),
        img_metas=img_metas,
    )


def level_shapes(image_wh=(704, 256), strides=(4, 8, 16, 32)):
    w, h = image_wh
    return [(h // s, w // s) for s in strides]


def feature_maps_nchw(bs, frame_idx, image_wh=(704, 256), channels=256, num_cams=6, seed=0, scale=1.0):
    """list of 4 tensors [bs, 6, C, H, W] (what FPN + reshape hands to feature_maps_format)."""
    out = []
    for lvl, (h, w) in enumerate(level_shapes(image_wh)):
        x = randn(f"synth.feat.f{frame_idx}.l{lvl}", (bs, num_cams, channels, h, w), seed)
        out.append(torch.from_numpy(x * scale))
    return out


def images(bs, frame_idx, image_wh=(704, 256), seed=0):
    w, h = image_wh
    return torch.from_numpy(randn(f"synth.img.f{frame_idx}", (bs, 6, 3, h, w), seed))


_DAMPED = re.compile(r"layers\.\d+\.layers\.10\.(weight|bias)$")  # last Linear of the refine MLPs


def procedural_tensor(key, shape, seed=0):
    """Deterministic value for one state_dict entry. Scales keep activations O(1) through the
    6-layer decoder and damp the refinement heads so anchors do not drift."""
    shape = tuple(shape)
    leaf = key.rsplit(".", 1)[-1]
    z = randn(key, shape, seed) if len(shape) else randn(key, (1,), seed)[0]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, np.int64)
    if leaf == "running_mean":
        return 0.05 * z
    if leaf == "running_var":
        return 1.0 + 0.1 * np.abs(z)
    if leaf == "anchor" and len(shape) == 2 and shape[1] == 11:
        return anchors(shape[0], seed=seed)
    if leaf == "instance_feature":
        return 0.5 * z
    if leaf == "scale":  # mmcv Scale
        return 1.0 + 0.1 * z
    if leaf == "in_proj_weight":
        return z / math.sqrt(shape[1])
    if leaf in ("in_proj_bias", "bias"):
        val = 0.05 * z
        if key.endswith("sampling_offsets.bias"):
            val = 2.0 * z  # offsets are in feature-map pixels (group_attn.py:191-196)
        if _DAMPED.search(key):
            val = 0.02 * val
        return val
    if leaf == "weight":
        if len(shape) == 1:  # LayerNorm / BatchNorm gain
            return 1.0 + 0.1 * z
        fan_in = int(np.prod(shape[1:]))
        gain = math.sqrt(2.0) if len(shape) == 4 else 1.0
        val = z * (gain / math.sqrt(fan_in))
        if key.endswith("sampling_offsets.weight"):
            val = 0.5 * val
        if _DAMPED.search(key):
            val = 0.02 * val
        return val
    return 0.1 * z


def load_procedural(module, seed=0, skip=("fix_scale",)):
    """Overwrite every parameter/buffer of `module` with procedural_tensor(key, shape)."""
    sd = module.state_dict()
    new = {}
    for key, ref in sd.items():
        if key.rsplit(".", 1)[-1] in skip:
            continue
        val = procedural_tensor(key, ref.shape, seed)
        new[key] = torch.as_tensor(np.asarray(val)).to(dtype=ref.dtype).reshape(ref.shape)
    module.load_state_dict(new, strict=False)
    return module


# ----------------------------------------------------------------------------- trace sketches
FULL_NUMEL = 16384
SKETCH_COLS = 4


def sketch(t):
    """Compact signature of a traced tensor for the golden fixtures: small tensors are kept whole,
    large float tensors [.., D] are reduced per row with a fixed random projection [D, 4]."""
    t = torch.as_tensor(t).detach().cpu()
    if t.dtype == torch.bool:
        t = t.to(torch.uint8)
    if not t.is_floating_point() or t.numel() <= FULL_NUMEL or t.dim() < 2:
        return t.numpy()
    d = t.shape[-1]
    proj = torch.from_numpy(randn(f"sketch.{d}", (d, SKETCH_COLS))) / math.sqrt(d)
    return (t.reshape(-1, d).double() @ proj.double()).float().numpy()


class Trace:
    """Ordered (name, sketch) records; the same call sites exist in the golden generator (hooks
    on the reference modules), in oracle/ and in the HIP head, so traces line up by name."""

    def __init__(self):
        self.items = {}
        self._count = {}

    def add(self, name, value):
        if value is None:
            return
        k = self._count.get(name, 0)
        self._count[name] = k + 1
        key = f"{name}#{k}"
        if isinstance(value, (list, tuple)) and not torch.is_tensor(value):
            value = torch.as_tensor(np.asarray(value))
        self.items[key] = sketch(value)

    def as_npz_dict(self, prefix=""):
        return {prefix + k: v for k, v in self.items.items()}
