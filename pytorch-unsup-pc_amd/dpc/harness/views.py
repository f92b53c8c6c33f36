"""View sampler of the training step: which views of which objects a step trains on, and the tensors it selects.

Restates ModelBase.preprocess of the reference (dpc/models/model_base_to.py:23-109) for the harness: host numpy RNG with
the reference's call order (one np.random.choice per object, in batch order), so a seeded run picks the same views; the
selection itself is an index_select on whatever device the raw tensors live on.  Pinned by fixture F12
(tests/golden/f12_view_sampler.npz, produced by the reference).

Not restated: `camera_quaternion` (quaternion_from_campos over util/euler.py), which only the pose-supervised branch and
the pose evaluation read -- outside SURVEY.md section 8.
"""
import numpy as np
import torch


def pool_single_view(cfg, tensor, view_idx):
    """Rows of the `view_idx`-th selected view of every object (model_base_to.py:7-9)."""
    return tensor[torch.arange(cfg.batch_size, device=tensor.device) * cfg.step_size + view_idx]


def camera_from_blender(their):
    """4x4 extrinsic in the reference's convention from a Blender one (dpc/util/camera.py:15-37): rows 0 and 2 swapped,
    columns permuted (x, y, z) -> (-x, z, y) with the signs of camera.py."""
    their = np.asarray(their, dtype=np.float32)
    our = np.zeros((4, 4), dtype=np.float32)
    our[0, :3] = (-their[2, 0], their[2, 2], their[2, 1])
    our[1, :3] = (their[1, 0], -their[1, 2], -their[1, 1])
    our[2, :3] = (-their[0, 0], their[0, 2], their[0, 1])
    our[0, 3], our[1, 3], our[2, 3], our[3, 3] = their[2, 3], their[1, 3], their[0, 3], their[3, 3]
    return our


def sample_view_indices(cfg, step_size, num_views, random_views=True, all_num_views=None):
    """[(object, view)] rows, object-major, and the 0/1 validity of each (model_base_to.py:36-62).
    variable_num_views: object n has all_num_views[n, 0] real views; missing ones are padded with view 0, marked 0."""
    max_views = num_views if cfg.num_views_to_use == -1 else cfg.num_views_to_use
    rows, valid = [], []
    for n in range(cfg.batch_size):
        ok = np.ones(step_size, dtype=np.float32)
        if cfg.variable_num_views:
            have = int(all_num_views[n, 0])
            ids = np.random.choice(have, min(step_size, have), replace=False)
            if have < step_size:
                ids = np.concatenate((ids, np.zeros(step_size - have, dtype=ids.dtype)))
                ok[have:] = 0.0
        elif random_views:
            ids = np.random.choice(max_views, step_size, replace=False)
        else:
            ids = np.arange(0, step_size).astype(np.int64)
        rows.append(np.stack((np.full(step_size, n, dtype=np.int64), ids.astype(np.int64)), axis=-1))
        valid.append(ok)
    return np.concatenate(rows, axis=0), np.concatenate(valid, axis=0)


def sample_views(cfg, raw_inputs, step_size, random_views=True):
    """The reference's `inputs` dict: images [B*V,...], masks, valid_samples, images_1 (first selected view of every
    object), depths if cfg.saved_depth, matrices if cfg.saved_camera (model_base_to.py:64-104)."""
    num_views = raw_inputs["image"].shape[1]
    counts = raw_inputs["num_views"].cpu().numpy() if cfg.variable_num_views else None
    idx, valid = sample_view_indices(cfg, step_size, num_views, random_views, counts)
    pick = lambda data: data[torch.from_numpy(idx[:, 0]).to(data.device), torch.from_numpy(idx[:, 1]).to(data.device)]
    inputs = {"valid_samples": torch.from_numpy(valid).to(raw_inputs["image"].device),
              "masks": pick(raw_inputs["mask"]), "images": pick(raw_inputs["image"])}
    if cfg.saved_depth:
        inputs["depths"] = pick(raw_inputs["depth"])
    inputs["images_1"] = pool_single_view(cfg, inputs["images"], 0)
    if cfg.saved_camera and "extrinsic" in raw_inputs:
        extr = pick(torch.as_tensor(raw_inputs["extrinsic"])).cpu().numpy()
        inputs["matrices"] = torch.from_numpy(np.stack([camera_from_blender(m) for m in extr]).astype(np.float64)).to(
            raw_inputs["image"].device)   # float64 like the reference's np.zeros(extr.shape)
    return inputs
