"""MI355X-native drop-in for the reference's `faster_rcnn` package (hot path only).

    config.Config                       same attribute bag               (config.py)
    rpn.rpn_to_roi / calc_iou / ...     HIP-backed proposal + labelling  (rpn.py)
    utils.calc_region_props / iou ...   HIP-backed anchor targets        (utils.py)
    RADNet.RADNet / load_radnet         construct / predict surface      (RADNet.py)
    base_models.resnet50 / vgg16        layer programs + feature sizes   (base_models/*.py)
    losses                              loss factories (device kernels)  (losses.py)

Everything numerical runs in libradnet_hip.so (hand-written gfx950 kernels) through the
C ABI declared in include/radnet_hip.h; there is no CPU fallback.
"""
