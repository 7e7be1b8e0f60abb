"""Detection records -> nuScenes submission entries, on the host.

Restates datasets/nuscenes_dataset.py:504-586 (`_format_bbox`) with its helpers
`output_to_nusc_box` (:824-874) and `lidar_nusc_box_to_global` (:877-899) of the reference without
nuscenes-devkit / pyquaternion (absent here): quaternions are (w, x, y, z) numpy arrays. The class
ranges are those of nuscenes-devkit's `detection_cvpr_2019` config [third-party, restated: parity
unpinned -- the reference has no fixture for this step and the devkit is not in the container].
SURVEY.md §8(f) item 2."""
import json
import math

import numpy as np

DEFAULT_ATTRIBUTE = {  # nuscenes_dataset.py:26-37
    "car": "vehicle.parked", "pedestrian": "pedestrian.moving", "trailer": "vehicle.parked", "truck": "vehicle.parked",
    "bus": "vehicle.moving", "motorcycle": "cycle.without_rider", "construction_vehicle": "vehicle.parked",
    "bicycle": "cycle.without_rider", "barrier": "", "traffic_cone": "",
}
CLASS_RANGE = {  # detection_cvpr_2019
    "car": 50, "truck": 50, "bus": 50, "trailer": 50, "construction_vehicle": 50, "pedestrian": 40, "motorcycle": 40,
    "bicycle": 40, "traffic_cone": 30, "barrier": 30,
}


def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw])


def quat_rotmat(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def yaw_quat(yaw):
    return np.array([math.cos(yaw / 2), 0.0, 0.0, math.sin(yaw / 2)])


def format_sample(det, info, classes, tracking=False, threshold=None):
    """One sample's result dict (decoder.py:230-251 keys boxes_3d / scores_3d / labels_3d /
    cls_scores / instance_ids) -> list of nuScenes annotation dicts. `info` holds sample token,
    lidar2ego_{rotation,translation}, ego2global_{rotation,translation} (quaternions w,x,y,z)."""
    box3d = np.asarray(det["boxes_3d"], np.float64)
    scores = np.asarray(det["scores_3d"], np.float64)
    labels = np.asarray(det["labels_3d"]).astype(np.int64)
    ids = np.asarray(det["instance_ids"]).astype(np.int64) if "instance_ids" in det else None
    if threshold is not None:  # :830-838
        mask = (np.asarray(det["cls_scores"]) if "cls_scores" in det else scores) >= threshold
        box3d, scores, labels = box3d[mask], scores[mask], labels[mask]
        ids = ids[mask] if ids is not None else None
    l2e_q, l2e_t = np.asarray(info["lidar2ego_rotation"], np.float64), np.asarray(info["lidar2ego_translation"], np.float64)
    e2g_q, e2g_t = np.asarray(info["ego2global_rotation"], np.float64), np.asarray(info["ego2global_translation"], np.float64)
    r_l2e, r_e2g = quat_rotmat(l2e_q), quat_rotmat(e2g_q)
    annos = []
    for i in range(len(box3d)):
        name = classes[labels[i]]
        center = box3d[i, :3].copy()
        wlh = box3d[i, [4, 3, 5]]  # nus_box_dims = dims[:, [1, 0, 2]] (:849)
        quat = yaw_quat(box3d[i, 6])
        vel = np.array([box3d[i, 7], box3d[i, 8], 0.0])
        # lidar -> ego (:887-888)
        center, quat, vel = r_l2e @ center + l2e_t, quat_mul(l2e_q, quat), r_l2e @ vel
        if np.linalg.norm(center[:2]) > CLASS_RANGE[name]:  # :890-894
            continue
        # ego -> global (:896-897)
        center, quat, vel = r_e2g @ center + e2g_t, quat_mul(e2g_q, quat), r_e2g @ vel
        if tracking and name in ("barrier", "traffic_cone", "construction_vehicle"):
            continue
        if math.hypot(vel[0], vel[1]) > 0.2:  # :526-549
            if name in ("car", "construction_vehicle", "bus", "truck", "trailer"):
                attr = "vehicle.moving"
            elif name in ("bicycle", "motorcycle"):
                attr = "cycle.with_rider"
            else:
                attr = DEFAULT_ATTRIBUTE[name]
        else:
            attr = "pedestrian.standing" if name == "pedestrian" else ("vehicle.stopped" if name == "bus" else DEFAULT_ATTRIBUTE[name])
        anno = dict(sample_token=info["token"], translation=center.tolist(), size=wlh.tolist(), rotation=quat.tolist(),
                    velocity=vel[:2].tolist())
        if tracking:
            anno.update(tracking_name=name, tracking_score=float(scores[i]), tracking_id=str(int(ids[i])))
        else:
            anno.update(detection_name=name, detection_score=float(scores[i]), attribute_name=attr)
        annos.append(anno)
    return annos


def write_submission(results, infos, classes, path, modality=None, tracking=False, threshold=None):
    """results: list of per-sample dicts (or {'img_bbox': dict}); infos: matching list of sample infos."""
    out = {}
    for res, info in zip(results, infos):
        det = res.get("img_bbox", res)
        out[info["token"]] = format_sample(det, info, classes, tracking, threshold)
    sub = {"meta": modality or dict(use_lidar=False, use_camera=True, use_radar=False, use_map=False, use_external=False),
           "results": out}
    with open(path, "w") as f:
        json.dump(sub, f)
    return path
