"""Run configuration for the MI355X-native Faster R-CNN path.

Drop-in for the reference's `faster_rcnn/config.py` (config.py:5-133): the same plain
attribute bag, same names, same defaults, no methods, picklable under the module path
`faster_rcnn.config.Config` so a `config.pickle` written by either side loads on the other
(RADNet.py:724, train.py:180-181).  tests/test_config.py compares `Config().__dict__`
against the attribute dump of the reference's own class (tests/golden/config_attrs.json).
"""
import copy

# name -> default.  Grouped as in the reference; values are the checked-in ones
# (config.py:11-133), not the commented-out alternatives.
_DEFAULTS = (
    ("verbose", True),
    # backbone
    ("network", "resnet50"),            # 'resnet50' | 'vgg16'
    ("base_net_trainable", False),      # train.py: whole base frozen
    ("base_net_cont_trainable", True),  # cont_train.py: stages 3-4 unfrozen
    ("base_net_weights", "imagenet"),
    # augmentation switches (host data feed; not on the device path)
    ("use_horizontal_flips", True), ("use_vertical_flips", True), ("use_90_rotations", True),
    ("use_rotations", True), ("use_shear", True), ("use_brightness", True), ("use_noise", True),
    ("use_img_type", False),
    ("img_types", ["enhanced_topo_grey", "topo_grey"]),
    # tiling
    ("tile_size", 2000), ("tile_overlap", 400), ("tile_bbox_clip_threshold", 0.75),
    ("max_n_tiles_train", 1), ("max_n_tiles_val", 1), ("include_full_img", False),
    # anchors: 4 scales x 3 ratios = 12 per location
    ("anchor_box_scales", [64, 128, 256, 512]),
    ("anchor_box_ratios", [[1.0, 1.0], [1.0, 2.0], [2.0, 1.0]]),
    ("img_size", 600),                  # short side after resize
    ("n_rois", 20),                     # RoIs per classifier-head call
    ("rpn_stride", 16),
    ("balanced_classes", True),
    ("std_scaling", 4.0),
    ("classifier_regr_std", [8.0, 8.0, 4.0, 4.0]),
    ("rpn_min_overlap", 0.3), ("rpn_max_overlap", 0.7),
    ("classifier_min_overlap", 0.1), ("classifier_max_overlap", 0.5),
    ("class_mapping", {"boat": 0, "human": 1, "other": 2, "animal": 3, "circle": 4, "wheel": 5, "bg": 6}),
)


class Config:

    def __init__(self):
        for key, value in _DEFAULTS:
            setattr(self, key, copy.deepcopy(value))
        self.model_path = "faster_rcnn_" + self.network
