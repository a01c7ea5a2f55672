"""Configuration seen by the drop-in modules.

Inside the reference code base the modules must honour the process-wide singleton
`core.DefaultConfig()` exactly like the originals do (/root/reference/src/models/eye_net.py:30,
refine_net.py:29; keys at src/core/config_default.py:44-129).  `get_config()` therefore returns that
singleton whenever the reference's `core` package is importable/imported, and otherwise a stand-alone
mirror with the same hot-path keys, defaults and `import_json` / `import_dict` / `override` methods
(src/core/config_default.py:168-199), so the same modules run outside the reference (bench, tests).
"""
import json
import sys


class HotPathConfig(object):
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

    @property
    def learning_rate(self):          # config_default.py:81-83
        return self.batch_size * self.base_learning_rate

    def import_dict(self, dictionary, strict=False):
        """Keys outside the hot path (data loading, logging, ...) are accepted and ignored unless
        strict; known keys are type-checked like config_default.py:188-194."""
        for key, value in dictionary.items():
            if not hasattr(type(self), key):
                if strict:
                    raise ValueError('Unknown configuration key: ' + key)
                continue
            cur = getattr(self, key)
            if type(cur) is float and type(value) is int:
                value = float(value)
            elif type(cur) is not type(value):
                raise TypeError('Config key %s expects %s, got %s' % (key, type(cur), type(value)))
            setattr(self, key, value)

    def import_json(self, json_path, strict=False):
        with open(json_path, 'r') as f:
            self.import_dict(json.load(f), strict=strict)

    def override(self, key, value):
        if not hasattr(type(self), key):
            raise ValueError('Unknown configuration key: ' + key)
        setattr(self, key, value)


_standalone = None


def get_config():
    """The reference's singleton if its `core` package is loaded, else the stand-alone mirror."""
    core = sys.modules.get('core')
    if core is not None and hasattr(core, 'DefaultConfig'):
        return core.DefaultConfig()
    global _standalone
    if _standalone is None:
        _standalone = HotPathConfig()
    return _standalone


def reset_standalone_config():
    global _standalone
    _standalone = HotPathConfig()
    return _standalone
