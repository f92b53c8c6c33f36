"""The prediction files between the reference's `predict` and `eval` steps, and the evaluation's per-model loop.

    writer   dpc/run/predict_to.py:331-336   pickle of {"points": [views, N, 3] float32, "camera_pose": [views, 4]}
                                              -> <save_pred_dir>/<model>_pc.pkl  ("camera_pose" only with predict_pose)
    reader   dpc/run/eval_chamfer_to.py:95-130  per model: load the pickle (optional "num_points" truncates each view's
                                              cloud), rotate by the reference rotation when the shape was learnt without
                                              supervision, Chamfer distance to the ground-truth cloud in both directions

Host I/O plus dpc.render.point_cloud_distance (the GPU nearest-point kernel) and quaternion_rotate: files written by the
reference load here and files written here load there.
"""
import pickle

import numpy as np
import torch


def save_predictions(path, points, camera_pose=None, num_points=None):
    """Write `<model>_pc.pkl` exactly as predict_to.py does (pickle.HIGHEST_PROTOCOL, numpy arrays)."""
    as_np = lambda x: x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
    save_dict = {"points": as_np(points)}
    if camera_pose is not None:
        save_dict["camera_pose"] = as_np(camera_pose)
    if num_points is not None:
        save_dict["num_points"] = as_np(num_points)
    with open(path, "wb") as handle:
        pickle.dump(save_dict, handle, protocol=pickle.HIGHEST_PROTOCOL)


def load_predictions(path):
    """(points [views,N,3], camera_pose [views,4] | None, num_points [views] | None) of a `<model>_pc.pkl`."""
    with open(path, "rb") as handle:
        data = pickle.load(handle)
    points = np.squeeze(data["points"])
    if points.ndim == 2:  # a single view
        points = points[None]
    nums = np.squeeze(data["num_points"]).reshape(-1) if "num_points" in data else None
    return points, data.get("camera_pose"), nums


def chamfer_of_predictions(points, gt_points, reference_rotation=None, num_points=None, device=None):
    """The body of eval_chamfer_to.py's model loop (:108-130): per view (pred -> gt, gt -> pred) mean nearest distances,
    [views, 2] float64.  reference_rotation: a quaternion [1,4] applied to every predicted cloud first (eval_unsup)."""
    from . import point_cloud_distance, quaternion_rotate

    device = torch.device("cuda") if device is None else device
    gt = torch.from_numpy(np.ascontiguousarray(gt_points)).to(device)
    out = np.zeros((points.shape[0], 2), dtype=np.float64)
    for i in range(points.shape[0]):
        pred = points[i]
        if num_points is not None:
            pred = pred[0:int(num_points[i])]
        p = torch.from_numpy(np.ascontiguousarray(pred)).to(device)
        if reference_rotation is not None:
            rot = torch.from_numpy(np.ascontiguousarray(reference_rotation)).to(device)
            p = quaternion_rotate(p.unsqueeze(0), rot).squeeze(0).to(p.dtype)
        pred_to_gt = point_cloud_distance(p, gt)[1]
        gt_to_pred = point_cloud_distance(gt, p)[1]
        assert not bool(torch.isnan(pred_to_gt).any())   # the reference asserts the same
        out[i, 0], out[i, 1] = float(pred_to_gt.double().mean()), float(gt_to_pred.double().mean())
    return out
