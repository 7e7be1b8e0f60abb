"""CPU: nuScenes submission formatter (simpb_amd/results.py) known-answer checks. The reference's
formatter needs nuscenes-devkit/pyquaternion, which are not in the container: parity unpinned."""
import math

import numpy as np

from simpb_amd import results
from simpb_amd.configs import CLASS_NAMES


def _info(yaw_e2g=0.0, t=(0, 0, 0)):
    return dict(token="tok", lidar2ego_rotation=[1, 0, 0, 0], lidar2ego_translation=[0, 0, 0],
                ego2global_rotation=results.yaw_quat(yaw_e2g).tolist(), ego2global_translation=list(t))


def test_identity_and_rotation():
    det = dict(boxes_3d=np.array([[10.0, 0, 0, 2, 4, 1.5, 0.0, 3.0, 0.0, 0.0], [60.0, 0, 0, 2, 4, 1.5, 0, 0, 0, 0]]),
               scores_3d=np.array([0.9, 0.8]), labels_3d=np.array([0, 0]), instance_ids=np.array([5, 6]))
    a = results.format_sample(det, _info(), CLASS_NAMES)
    assert len(a) == 1  # the box at 60 m is outside the 50 m car range
    assert a[0]["translation"] == [10.0, 0.0, 0.0] and a[0]["size"] == [4.0, 2.0, 1.5]
    assert a[0]["attribute_name"] == "vehicle.moving" and a[0]["detection_name"] == "car"
    b = results.format_sample(det, _info(math.pi / 2, (100, 200, 0)), CLASS_NAMES)[0]
    assert np.allclose(b["translation"], [100.0, 210.0, 0.0]) and np.allclose(b["velocity"], [0.0, 3.0], atol=1e-12)
    assert np.allclose(b["rotation"], results.yaw_quat(math.pi / 2))


def test_tracking_and_attributes():
    det = dict(boxes_3d=np.array([[5.0, 0, 0, 1, 1, 2, 0, 0.0, 0.0, 0], [6.0, 0, 0, 1, 1, 1, 0, 0, 0, 0]]),
               scores_3d=np.array([0.5, 0.4]), labels_3d=np.array([8, 9]), cls_scores=np.array([0.6, 0.1]),
               instance_ids=np.array([11, 12]))
    d = results.format_sample(det, _info(), CLASS_NAMES)
    assert [x["attribute_name"] for x in d] == ["pedestrian.standing", ""]
    t = results.format_sample(det, _info(), CLASS_NAMES, tracking=True, threshold=0.2)
    assert len(t) == 1 and t[0]["tracking_id"] == "11" and t[0]["tracking_name"] == "pedestrian"
