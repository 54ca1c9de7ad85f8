"""VGG16 backbone protocol of the reference (base_models/vgg16.py) for the MI355X-native engine.

    get_img_output_length   vgg16.py:18-23    stride-16 feature map: L // 16
    preprocess              vgg16.py:25-27    keras 'caffe' preprocess_input
    nn_base                 vgg16.py:29-65    keras.applications VGG16 cut at block5_conv3
    classifier_layer        vgg16.py:67-124   RoI crop-resize 7x7 + Flatten + fc1/fc2 (4096, ReLU, Dropout .5) + 2 dense

As in resnet50.py of this package the functions return spec objects; the layers run in libradnet_hip.so
(radnet_hip.engine_vgg.VGG16Engine).
"""
from .resnet50 import preprocess  # noqa: F401  (both backbones use keras' 'caffe' mode: BGR minus the ImageNet means)

FINE_TUNING_CUT = 7        # vgg16.py:16 (input + block1 + block2 always frozen)
N_FEATURES = 512
POOLING_REGIONS = 7


def get_img_output_length(width, height):
    return width // 16, height // 16


class BaseSpec:
    def __init__(self, trainable, weights):
        self.network = "vgg16"
        self.trainable = trainable
        self.weights = weights
        self.n_features = N_FEATURES


class ClassifierSpec:
    def __init__(self, base, n_rois, nb_classes):
        self.base, self.n_rois, self.nb_classes = base, n_rois, nb_classes
        self.pooling_regions = POOLING_REGIONS


def nn_base(input_tensor=None, trainable=False, weights='imagenet'):
    return BaseSpec(trainable, weights)


def classifier_layer(input_layer, input_rois, n_rois, nb_classes=4):
    spec = ClassifierSpec(input_layer, n_rois, nb_classes)
    return [("dense_class_%d" % nb_classes, spec), ("dense_regress_%d" % nb_classes, spec)]
