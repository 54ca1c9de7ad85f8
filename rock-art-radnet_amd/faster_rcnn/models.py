"""Keras-like model objects over the HIP engine: the seam the reference's drivers use
(train.py:192-256,288,291,393,494,513,574; cont_train.py:155,164; RADNet.py:124,552,752-770).

    model_rpn.predict(X) / predict_on_batch / train_on_batch / test_on_batch
    model_classifier.train_on_batch([X, rois], [Y1, Y2]) / test_on_batch
    model_detector.predict([F, ROIs])
    model_all.save_weights(path) / load_weights(path, by_name=True)

All arrays in and out are host NumPy, calls are synchronous, the models share one set of weights (one engine),
and each of model_rpn / model_classifier owns its own Adam state (train.py:236-252).  The reference recomputes the
frozen base three times per iteration on the same image (train.py:288,291,393); here the base features are cached
per input array, which is bit-identical because the base is frozen in train.py mode.
"""
import os

import numpy as np


class Adam:
    """Stand-in for keras.optimizers.Adam(lr=...) in `compile` (Keras-2 defaults beta=(.9,.999), eps=1e-7)."""

    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.lr, self.beta_1, self.beta_2, self.epsilon = lr, beta_1, beta_2, epsilon


class _Shared:
    def __init__(self, C, device_index=0, weights=None, lr=5e-5, bce_mode=0, workload=None):
        from radnet_hip import synth
        from radnet_hip import make_engine
        self.C = C
        self.eng = make_engine(C, device_index=device_index, bce_mode=bce_mode, lr=lr, workload=workload)
        if weights is None:
            gen = synth.synthetic_weights_vgg16 if C.network == "vgg16" else synth.synthetic_weights
            weights = gen(seed=3, n_anchors=self.eng.A, n_classes=self.eng.nc)
        self.W = weights
        self.eng.set_weights(self.W)
        self._x_key = None
        self._bp = None
        self._f_host = None

    def base(self, X):
        """Base features for a preprocessed (1,H,W,3) fp32 batch; cached while the same array content is fed again."""
        X = np.asarray(X, dtype=np.float32)
        key = (X.shape, X.__array_interface__["data"][0], float(X.ravel()[::4099].sum()))
        if key != self._x_key:
            bp = self.eng.upload_preprocessed(X)
            self.eng.base_forward(bp)
            self._x_key, self._bp, self._f_host = key, bp, None
        return self._bp

    def base_from_u8(self, img_dev, slot=0):
        """Base features for a uint8 BGR HWC image that is ALREADY on the device at network size (RADNet's device-resident
        tile path): preprocess kernel + base forward, nothing crosses PCIe.  slot: buffer set (two tiles in flight)."""
        eng = self.eng
        H, W = int(img_dev.shape[0]), int(img_dev.shape[1])
        bp = eng._plan_base(1, H, W, slot)
        eng.ctx.call("radnet_preprocess_bgr", img_dev, H, W, 4, bp["x"])
        eng.base_forward(bp)
        self._x_key, self._bp, self._f_host = None, bp, None
        return bp

    def sync_weights_from_engine(self):
        self.W.update(self.eng.get_weights())


class _ModelBase:
    def __init__(self, shared):
        self._s = shared
        self.optimizer = None
        self.loss = None

    def compile(self, optimizer=None, loss=None, metrics=None, **kw):
        self.optimizer, self.loss = optimizer, loss
        lr = getattr(optimizer, "lr", None)
        if isinstance(lr, (int, float)):
            self._lr = float(lr)

    def _use_lr(self):
        if hasattr(self, "_lr"):
            self._s.eng.lr = self._lr

    def save_weights(self, path):
        """model_all.save_weights (train.py:574).  '*.h5' / '*.hdf5' -> a Keras-2 HDF5 weights file (layer groups, weight_names
        attributes: faster_rcnn/keras_h5.py, no h5py needed); anything else -> '<path>.npz' with keys '<layer>/<param>'."""
        self._s.sync_weights_from_engine()
        path = str(path)
        if path.endswith((".h5", ".hdf5")):
            from . import keras_h5
            keras_h5.write_keras_weights(path, self._s.W)
            return
        flat = {"%s/%s" % (n, k): v for n, d in self._s.W.items() for k, v in d.items()}
        np.savez(path if path.endswith(".npz") else path + ".npz", **flat)

    def load_weights(self, path, by_name=True):
        """model.load_weights(path, by_name=True) (RADNet.py:754,769; cont_train.py:155,164): layers are matched by NAME, layers
        of the file this model does not have are skipped, layers of the model the file lacks keep their weights.  Keras HDF5
        files (recognised by their signature, whatever the suffix) and this package's .npz files."""
        path = str(path)
        if not os.path.exists(path) and os.path.exists(path + ".npz"):
            path += ".npz"
        with open(path, "rb") as f:
            magic = f.read(8)
        from . import keras_h5
        if magic == keras_h5.SIGNATURE:
            loaded = keras_h5.load_weights_by_name(path, known_layers=set(self._s.W))
            for n, d in loaded.items():
                for k, v in d.items():
                    if k in self._s.W[n] and tuple(np.shape(self._s.W[n][k])) != tuple(v.shape):
                        raise ValueError("load_weights: layer %r %s has shape %s in the file, the model expects %s"
                                         % (n, k, v.shape, np.shape(self._s.W[n][k])))
                self._s.W[n].update(d)
        else:
            z = np.load(path, allow_pickle=False)
            for key in z.files:
                n, k = key.rsplit("/", 1)
                self._s.W.setdefault(n, {})[k] = z[key]
        self._s.eng.set_weights(self._s.W)
        self._s._x_key = None


class RPNModel(_ModelBase):
    """Model(img_input, rpn[:2]) (train.py:209) or Model(img_input, [cls, regr, F]) (RADNet.py:752-753)."""

    def __init__(self, shared, with_features=False):
        super().__init__(shared)
        self.with_features = with_features

    def _forward(self, X):
        eng = self._s.eng
        bp = self._s.base(X)
        rp = eng.rpn_forward(bp)
        return bp, rp

    def predict(self, X, **kw):
        eng = self._s.eng
        bp, rp = self._forward(X)
        A, fh, fw = eng.A, rp["fh"], rp["fw"]
        pred = rp["pred"].cpu().numpy()
        out = [pred[:, :A].reshape(1, fh, fw, A).copy(), pred[:, A:5 * A].reshape(1, fh, fw, 4 * A).copy()]
        if self.with_features:
            F = bp["F"].cpu().numpy()
            self._s._f_host = F
            out.append(F)
        return out

    predict_on_batch = predict

    def propose_device(self, img_dev, overlap_thresh=0.7, max_boxes=300):
        """Device-resident twin of predict() + rpn.rpn_to_roi() for RADNet's tile path: uint8 BGR image on the device ->
        proposals (n,4) int64 x1,y1,x2,y2 in feature-map units on the host (a few KB) and the plans that hold the feature
        map on the device.  Same kernels as the NumPy-facing calls, so the same proposals."""
        return self.propose_finish(self.propose_launch(img_dev, overlap_thresh, max_boxes))

    def propose_launch(self, img_dev, overlap_thresh=0.7, max_boxes=300, slot=0):
        """First half of propose_device: everything enqueued on the current lane, nothing read back."""
        eng = self._s.eng
        bp = self._s.base_from_u8(img_dev, slot)
        rp = eng.rpn_forward(bp)
        R, Rn = eng.proposals(rp, overlap_thresh=overlap_thresh, max_boxes=max_boxes)
        return R, Rn, bp

    @staticmethod
    def propose_finish(handle):
        R, Rn, bp = handle
        n = int(Rn.cpu()[0])
        if n <= 0:
            raise ValueError("rpn_to_roi: no valid box")        # the reference's failed tuple-unpack (rpn.py:164-172)
        return R[:n].cpu().numpy(), bp

    def _targets(self, Y, rp):
        import torch
        y_cls = torch.from_numpy(np.ascontiguousarray(Y[0], dtype=np.float32).reshape(rp["M"], -1)).cuda()
        y_regr = torch.from_numpy(np.ascontiguousarray(Y[1], dtype=np.float32).reshape(rp["M"], -1)).cuda()
        return y_cls, y_regr

    def train_on_batch(self, X, Y):
        eng = self._s.eng
        self._use_lr()
        bp, rp = self._forward(X)
        y_cls, y_regr = self._targets(Y, rp)
        eng.set_accumulate(rp["bwd"], False)
        eng.rpn_backward(rp, y_cls, y_regr)
        eng.adam(eng.rpn_arena)
        l = eng.rpn_losses.cpu().numpy()
        return [float(l[0] + l[1]), float(l[0]), float(l[1])]

    def test_on_batch(self, X, Y):
        eng = self._s.eng
        bp, rp = self._forward(X)
        y_cls, y_regr = self._targets(Y, rp)
        eng.ctx.call("radnet_rpn_loss", rp["pred"], 64, y_cls, y_regr, rp["M"], eng.A, eng.bce_mode, rp["dz"], 64, eng.rpn_losses, eng.loss_scratch)
        l = eng.rpn_losses.cpu().numpy()
        return [float(l[0] + l[1]), float(l[0]), float(l[1])]


class ClassifierModel(_ModelBase):
    """Model([img_input, roi_input], classifier) (train.py:210): base + RoI crop-resize + stage 5 + dense heads."""

    def _prepare(self, inputs, targets=None, training=False):
        import torch
        eng = self._s.eng
        X, rois = inputs
        bp = self._s.base(X)
        rois = np.asarray(rois, dtype=np.float32).reshape(-1, 4)
        hp = eng._plan_head(rois.shape[0], bp["fh"], bp["fw"], bp["F"])
        hp["rois"].copy_(torch.from_numpy(rois))
        if targets is not None:
            hp["y1"].copy_(torch.from_numpy(np.ascontiguousarray(targets[0], dtype=np.float32).reshape(rois.shape[0], -1)))
            hp["y2"].copy_(torch.from_numpy(np.ascontiguousarray(targets[1], dtype=np.float32).reshape(rois.shape[0], -1)))
        eng.head_forward(hp, training=training)
        return hp

    def predict(self, inputs, **kw):
        hp = self._prepare(inputs)
        return [hp["pcls"].cpu().numpy()[None], hp["pregr"].cpu().numpy()[None]]

    def _losses(self):
        l = self._s.eng.det_losses.cpu().numpy()
        return [float(l[0] + l[1]), float(l[0]), float(l[1]), float(l[2])]

    def train_on_batch(self, inputs, targets):
        eng = self._s.eng
        self._use_lr()
        hp = self._prepare(inputs, targets, training=True)
        eng.set_accumulate(hp["bwd"], False)
        eng.head_backward(hp, accumulate=False)
        eng.adam(eng.head_arena)
        eng.refresh_head_shift()
        return self._losses()

    def test_on_batch(self, inputs, targets):
        eng = self._s.eng
        hp = self._prepare(inputs, targets)
        eng.ctx.call("radnet_det_loss", hp["pcls"], hp["pregr"], hp["y1"], hp["y2"], hp["R"], eng.nc, eng.nreg, hp["dz"], eng.det_losses)
        return self._losses()


class DetectorModel(_ModelBase):
    """Model([feature_map_input, roi_input], detector_layers) (RADNet.py:761-770)."""
    accepts_any_roi_count = True        # RADNet.apply_spatial_pyramid_pooling then sends all RoIs of a tile at once

    def _features(self, F):
        import torch
        s = self._s
        if s._f_host is not None and F is s._f_host:         # the very array model_rpn.predict returned: still on the device
            return s._bp["F"], s._bp["fh"], s._bp["fw"]
        Fd = getattr(self, "_fd", None)
        F = np.ascontiguousarray(F, dtype=np.float32)
        if Fd is None or tuple(Fd.shape) != F.shape:
            self._fd = Fd = torch.empty(F.shape, dtype=torch.float32, device="cuda")
        Fd.copy_(torch.from_numpy(F))
        return Fd, F.shape[1], F.shape[2]

    def predict_launch(self, inputs):
        """First half of predict: the classifier pass enqueued on the current lane, nothing read back."""
        import torch
        eng = self._s.eng
        F, rois = inputs
        if isinstance(F, dict) and "F" in F:                 # a base plan from RPNModel.propose_device: features on the device
            Fd, fh, fw = F["F"], F["fh"], F["fw"]
        else:
            Fd, fh, fw = self._features(F)
        rois = np.asarray(rois, dtype=np.float32).reshape(-1, 4)
        hp = eng._plan_head(rois.shape[0], fh, fw, Fd, training=False)
        hp["rois"].copy_(torch.from_numpy(rois))
        eng.head_forward(hp)
        return hp

    @staticmethod
    def predict_finish(hp):
        return [hp["pcls"].cpu().numpy()[None], hp["pregr"].cpu().numpy()[None]]

    def predict(self, inputs, **kw):
        return self.predict_finish(self.predict_launch(inputs))


class AllModel(_ModelBase):
    """Model([img, rois], rpn[:2] + classifier) (train.py:211): exists to save / load every weight."""


def build_models(C, device_index=0, weights=None, lr=5e-5, bce_mode=None, workload=None):
    """The four model objects of train.py:199-211 / RADNet.py:748-770 over one shared engine.
    Returns (model_rpn [2 outputs], model_classifier, model_all, model_rpn_predict [3 outputs], model_detector)."""
    from . import losses
    s = _Shared(C, device_index, weights, lr, losses.BCE_MODE if bce_mode is None else bce_mode, workload)
    return RPNModel(s), ClassifierModel(s), AllModel(s), RPNModel(s, with_features=True), DetectorModel(s)
