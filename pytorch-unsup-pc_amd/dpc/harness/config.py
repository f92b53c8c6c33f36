"""The live configuration of the reference's chair_unsupervised experiment, as a plain attribute dict.

Values: dpc/resources/default_config.yaml (lines 20-124, 175) overridden by experiments/chair_unsupervised/config.yaml
(vox_size 64, 21-tap Gaussian, sigma_rel 3.0 -> 0.2, 8000 points, keep-probability 0.07 -> 1, predicted pose with 4
candidates and a student weighted 20).  Only the keys the renderer and the harness read are listed.
"""


class AttrDict(dict):
    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key) from None

    __setattr__ = dict.__setitem__


def chair_unsupervised(**overrides):
    cfg = AttrDict(
        # networks
        z_dim=1024, fc_dim=1024, f_dim=16, input_shape=[128, 128, 3], pc_num_points=8000, pc_unit_cube=True,
        predict_pose=True, pose_predict_num_candidates=4, pose_candidates_num_layers=3, pose_predictor_student=True,
        pose_predictor_student_loss_weight=20.0, pose_student_align_loss=False,
        predict_translation=False, predict_translation_tanh=True, predict_translation_scaling_factor=0.15,
        pc_learn_occupancy_scaling=True, pc_occupancy_scaling_maximum=1.0, learn_focal_length=False, focal_length_mean=2.0, focal_length_range=1.0,
        # batch
        batch_size=8, step_size=4, variable_num_views=False, num_views_to_use=-1, saved_camera=True, saved_depth=False,
        # renderer
        vox_size=64, vox_size_z=-1, pc_gauss_kernel_size=21, pc_relative_sigma=3.0, pc_relative_sigma_end=0.2,
        camera_distance=2.0, focal_length=1.875, drc_logsum_clip_val=1e-5, max_depth=10.0,
        pc_fast=True, pose_quaternion=True, pc_separable_gauss_filter=True, drc_logsum=True, drc_tf_cumulative=True,
        ptn_max_projection=False, pc_rgb=False,
        # schedules, loss, optimiser
        pc_point_dropout=0.07, pc_point_dropout_scheduled=True, pc_point_dropout_exponential_schedule=False,
        pc_point_dropout_start_step=0.0, pc_point_dropout_end_step=1.0, max_number_of_steps=600000,
        proj_weight=1.0, drc_weight=0.0, proj_depth_weight=0.0, weight_decay=0.001, learning_rate=1e-4)
    cfg.update(overrides)
    return cfg
