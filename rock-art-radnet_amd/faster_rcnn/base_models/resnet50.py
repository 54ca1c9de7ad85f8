"""ResNet50 backbone protocol of the reference (base_models/resnet50.py) for the MI355X-native engine.

The reference's functions build Keras tensors; here they return small spec objects that
`faster_rcnn.models` binds to the HIP layer program (radnet_hip.engine).  The arithmetic of the layers they
describe runs in libradnet_hip.so.

    get_img_output_length   resnet50.py:19-35    stride-16 feature-map size
    preprocess              resnet50.py:37-39    keras 'caffe' preprocess_input on an RGB float batch
    nn_base                 resnet50.py:150-228  conv1 + stages 2-4 (frozen BN everywhere)
    classifier_layer        resnet50.py:231-281  RoI crop-resize 14x14 + stage 5 + avgpool + 2 dense
"""
import numpy as np

FINE_TUNING_CUT = 38        # resnet50.py:15 (conv1 + stage 2 always frozen)
WEIGHT_PATH = 'faster_rcnn/base_models/resnet50_weights_tf_dim_ordering_tf_kernels_notop.h5'
N_FEATURES = 1024
POOLING_REGIONS = 14


def get_img_output_length(width, height):
    def out_len(n):
        n += 6                                  # ZeroPadding2D((3, 3))
        for k in (7, 3, 1, 1):                  # conv1, maxpool, stage-3 and stage-4 strided 1x1
            n = (n - k + 2) // 2
        return n
    return out_len(width), out_len(height)


def preprocess(img):
    """keras.applications.resnet50.preprocess_input ('caffe' mode): RGB -> BGR, subtract the ImageNet BGR means,
    no scaling.  img: float array (..., 3) in RGB order; returns a new float32 array."""
    x = np.asarray(img, dtype=np.float32)[..., ::-1].copy()
    x[..., 0] -= 103.939
    x[..., 1] -= 116.779
    x[..., 2] -= 123.68
    return x


class BaseSpec:
    def __init__(self, trainable, weights):
        self.network = "resnet50"
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
