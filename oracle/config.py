"""Oracle-side mirror of the hot-path keys of the reference's config singleton.

TEST INFRASTRUCTURE.  Keys, defaults and JSON precedence follow
/root/reference/src/core/config_default.py:44-129 (only the keys that
src/models/{eye_net,refine_net,common,eve}.py and src/losses read).
"""
import json


class OracleConfig(object):
    # data / geometry (config_default.py:44-51)
    assumed_frame_rate = 10
    max_sequence_len = 30
    eyes_size = [128, 128]
    screen_size = [128, 72]
    actual_screen_size = [1920, 1080]
    load_screen_content = False
    # training (config_default.py:70-95)
    batch_size = 16
    weight_decay = 0.001
    base_learning_rate = 0.0005
    num_warmup_epochs = 0.0           # config_default.py:87-90: LR warm-up / decay (core/training.py:382-418)
    lr_decay_strategy = 'none'
    lr_decay_factor = 0.5
    lr_decay_epoch_interval = 0.5
    do_gradient_clipping = True
    gradient_clip_by = 'norm'
    gradient_clip_amount = 5.0
    # EyeNet (config_default.py:98-108)
    eye_net_load_pretrained = False
    eye_net_frozen = False
    eye_net_use_rnn = True
    eye_net_rnn_type = 'GRU'
    eye_net_rnn_num_cells = 1
    eye_net_rnn_num_features = 128
    eye_net_static_num_features = 128
    eye_net_use_head_pose_input = True
    loss_coeff_PoG_cm_initial = 0.0
    loss_coeff_g_ang_initial = 1.0
    loss_coeff_pupil_size = 1.0
    # RefineNet (config_default.py:111-126)
    refine_net_enabled = False
    refine_net_load_pretrained = False
    refine_net_do_offset_augmentation = True
    refine_net_offset_augmentation_sigma = 3.0
    refine_net_use_skip_connections = True
    refine_net_use_rnn = True
    refine_net_rnn_type = 'CGRU'
    refine_net_rnn_num_cells = 1
    refine_net_num_features = 64
    loss_coeff_heatmap_ce_initial = 0.0
    loss_coeff_heatmap_ce_final = 1.0
    loss_coeff_heatmap_mse_final = 0.0
    loss_coeff_PoG_cm_final = 0.001
    # heat-maps (config_default.py:129-133)
    gaze_heatmap_size = [128, 72]
    gaze_heatmap_sigma_initial = 10.0
    gaze_heatmap_sigma_history = 3.0
    gaze_heatmap_sigma_final = 5.0
    gaze_history_map_decay_per_ms = 0.999

    def __init__(self, json_path=None, **overrides):
        if json_path is not None:
            with open(json_path, 'r') as f:
                for k, v in json.load(f).items():
                    if hasattr(type(self), k):   # non-hot-path keys are ignored here
                        setattr(self, k, v)
        for k, v in overrides.items():
            if not hasattr(type(self), k):
                raise ValueError('Unknown configuration key: ' + k)
            setattr(self, k, v)

    @property
    def learning_rate(self):  # config_default.py:81-83
        return self.batch_size * self.base_learning_rate
