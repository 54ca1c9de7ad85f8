"""Loss factories with the reference's names (losses.py).  On the training path the losses are fused with their
gradients inside libradnet_hip.so (radnet_rpn_loss / radnet_det_loss); the callables returned here evaluate the
same kernels on host arrays so a driver can use them stand-alone, and carry the metadata `Model.compile` reads.

    rpn_loss_regr(num_anchors)    losses.py:16-44   masked smooth-L1 / sum(1e-4 + mask)
    rpn_loss_cls(num_anchors)     losses.py:47-66   K.binary_crossentropy as called there (see BCE_MODE)
    class_loss_regr(num_classes)  losses.py:69-90
    class_loss_cls                losses.py:93-95   mean categorical cross-entropy over the RoIs
"""
import numpy as np

lambda_rpn_regr = 1.0
lambda_rpn_class = 1.0
lambda_cls_regr = 1.0
lambda_cls_class = 1.0
epsilon = 1e-4

# losses.py:64 calls K.binary_crossentropy(y_pred, y_true).  Under the Keras 2.x API the reference is written
# against, the signature is (target, output): the *prediction* is taken as the target and the 0/1 label is clipped
# to [1e-7, 1-1e-7] and turned into a logit.  0 reproduces that (what the reference executes); 1 is the textbook
# BCE(label, clip(prediction)) of the Keras-1 argument order.
BCE_MODE = 0


def _rpn_eval(y_true_cls, y_true_regr, p_cls, p_regr, A):
    import torch
    from radnet_hip import runtime as rt
    ctx = rt.default_context()
    m = int(np.prod(p_cls.shape[:-1]))
    pred = np.zeros((m, 5 * A), np.float32)
    pred[:, :A] = np.asarray(p_cls, np.float32).reshape(m, A)
    pred[:, A:] = np.asarray(p_regr, np.float32).reshape(m, 4 * A)
    dz = torch.zeros(m, 5 * A, device="cuda")
    out = torch.zeros(2, device="cuda")
    scratch = torch.zeros(8, dtype=torch.float64, device="cuda")
    ctx.call("radnet_rpn_loss", rt.to_dev(pred), 5 * A, rt.to_dev(y_true_cls, np.float32), rt.to_dev(y_true_regr, np.float32), m, A, BCE_MODE,
             dz, 5 * A, out, scratch)
    return out.cpu().numpy()


def rpn_loss_regr(num_anchors):
    def rpn_loss_regr_fixed_num(y_true, y_pred):
        A = num_anchors
        y_cls = np.zeros(tuple(y_true.shape[:-1]) + (2 * A,), np.float32)
        p_cls = np.full(tuple(y_true.shape[:-1]) + (A,), 0.5, np.float32)
        return float(lambda_rpn_regr * _rpn_eval(y_cls, y_true, p_cls, y_pred, A)[1])
    rpn_loss_regr_fixed_num.kind, rpn_loss_regr_fixed_num.num_anchors = "rpn_regr", num_anchors
    return rpn_loss_regr_fixed_num


def rpn_loss_cls(num_anchors):
    def rpn_loss_cls_fixed_num(y_true, y_pred):
        A = num_anchors
        y_regr = np.zeros(tuple(y_true.shape[:-1]) + (8 * A,), np.float32)
        p_regr = np.zeros(tuple(y_true.shape[:-1]) + (4 * A,), np.float32)
        return float(lambda_rpn_class * _rpn_eval(y_true, y_regr, y_pred, p_regr, A)[0])
    rpn_loss_cls_fixed_num.kind, rpn_loss_cls_fixed_num.num_anchors = "rpn_cls", num_anchors
    return rpn_loss_cls_fixed_num


def _det_eval(y1, y2, p_cls, p_regr):
    import torch
    from radnet_hip import runtime as rt
    ctx = rt.default_context()
    r, nc = p_cls.shape[-2], p_cls.shape[-1]
    nreg = p_regr.shape[-1]
    dz = torch.zeros(r, nc + nreg, device="cuda")
    out = torch.zeros(3, device="cuda")
    ctx.call("radnet_det_loss", rt.to_dev(p_cls.reshape(r, nc), np.float32), rt.to_dev(p_regr.reshape(r, nreg), np.float32),
             rt.to_dev(y1.reshape(r, nc), np.float32), rt.to_dev(y2.reshape(r, 2 * nreg), np.float32), r, nc, nreg, dz, out)
    return out.cpu().numpy()


def class_loss_regr(num_classes):
    def class_loss_regr_fixed_num(y_true, y_pred):
        r, nreg = y_pred.shape[-2], y_pred.shape[-1]
        nc = num_classes + 1
        p = np.full((r, nc), 1.0 / nc, np.float32)
        y1 = np.zeros((r, nc), np.float32); y1[:, -1] = 1
        return float(lambda_cls_regr * _det_eval(y1, np.asarray(y_true), p, np.asarray(y_pred))[1])
    class_loss_regr_fixed_num.kind, class_loss_regr_fixed_num.num_classes = "det_regr", num_classes
    return class_loss_regr_fixed_num


def class_loss_cls(y_true, y_pred):
    r, nc = y_pred.shape[-2], y_pred.shape[-1]
    nreg = 4 * (nc - 1)
    return float(lambda_cls_class * _det_eval(np.asarray(y_true), np.zeros((r, 2 * nreg), np.float32), np.asarray(y_pred), np.zeros((r, nreg), np.float32))[0])


class_loss_cls.kind = "det_cls"
