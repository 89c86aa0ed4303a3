"""ORACLE (test infrastructure, never shipped, never measured as the product).

fp32 CPU restatement of the two networks whose arithmetic the reference delegates to TensorRT engines.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.

The reference's only open description of the network arithmetic is the PyTorch source under
``/root/reference/conversion`` - each function below cites the lines it follows.  Pinning: ``tests/golden/*.npz`` were
produced by importing those very ``nn.Module`` definitions in the build container (``tests/golden/make_golden.py``)
and ``tests/test_oracle_nets.py`` checks this restatement against them.  The recogniser (``model_irse.py``) imports
with stock PyTorch -> fully pinned.  The detector modules import ``torchvision`` (absent in this image); its goldens were
generated with an in-memory stand-in for ``torchvision.models._utils.IntermediateLayerGetter`` only (all arithmetic
still comes from the reference's ``net.py`` / ``retinaface_trim.py``) -> stated as "pinned with a torchvision shim" in
DESIGN.md.

Everything is plain ``torch.nn.functional`` on float32 CPU tensors taking a ``state_dict``-like mapping of numpy arrays.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm2d default, never overridden in the reference


def _t(sd, name):
    return torch.from_numpy(np.ascontiguousarray(sd[name], dtype=np.float32))


def _bn(x, sd, p):
    return F.batch_norm(x, _t(sd, p + ".running_mean"), _t(sd, p + ".running_var"), _t(sd, p + ".weight"), _t(sd, p + ".bias"),
                        False, 0.0, BN_EPS)


# ----------------------------------------------------------------------------------------------------------------------
# RetinaFace mobilenet0.25, trimmed
# ----------------------------------------------------------------------------------------------------------------------
def _conv_bn(x, sd, p, stride=1, relu=True):
    """``net.py:9-20`` conv_bn / conv_bn_no_relu: 3x3 pad 1 no bias + BN (+ReLU)."""
    x = _bn(F.conv2d(x, _t(sd, p + ".0.weight"), None, stride, 1), sd, p + ".1")
    return F.relu(x) if relu else x


def _conv_dw(x, sd, p, stride):
    """``net.py:29-38`` conv_dw: depthwise 3x3 + BN + ReLU, pointwise 1x1 + BN + ReLU."""
    w = _t(sd, p + ".0.weight")
    x = F.relu(_bn(F.conv2d(x, w, None, stride, 1, 1, w.shape[0]), sd, p + ".1"))
    return F.relu(_bn(F.conv2d(x, _t(sd, p + ".3.weight")), sd, p + ".4"))


_STAGES = dict(
    stage1=[("bn", 2), ("dw", 1), ("dw", 2), ("dw", 1), ("dw", 2), ("dw", 1)],
    stage2=[("dw", 2)] + [("dw", 1)] * 5,
    stage3=[("dw", 2), ("dw", 1)],
)


def _ssh(x, sd, p):
    """``net.py:55-66`` SSH.forward.  NB the 7x7 branch consumes conv5X5_1 (post-ReLU), ``net.py:58-62``."""
    c3 = _conv_bn(x, sd, p + ".conv3X3", relu=False)
    c5_1 = _conv_bn(x, sd, p + ".conv5X5_1")
    c5 = _conv_bn(c5_1, sd, p + ".conv5X5_2", relu=False)
    c7_2 = _conv_bn(c5_1, sd, p + ".conv7X7_2")
    c7 = _conv_bn(c7_2, sd, p + ".conv7x7_3", relu=False)
    return F.relu(torch.cat([c3, c5, c7], dim=1))


def retinaface_forward(sd, x, return_features=False):
    """``retinaface_trim.py:107-127`` in phase 'test'.

    x: float32 [B,3,H,W], planar BGR, mean-subtracted (binding ``input_det``).  Returns (loc [B,A,4], conf [B,A,2]).
    """
    with torch.no_grad():
        x = torch.as_tensor(x, dtype=torch.float32)
        feats = []
        for st in ("stage1", "stage2", "stage3"):  # body = IntermediateLayerGetter(return_layers stage1..3), config.py:17
            for i, (kind, s) in enumerate(_STAGES[st]):
                p = "body.%s.%d" % (st, i)
                x = _conv_bn(x, sd, p, s) if kind == "bn" else _conv_dw(x, sd, p, s)
            feats.append(x)
        # FPN, net.py:81-98
        o1 = F.relu(_bn(F.conv2d(feats[0], _t(sd, "fpn.output1.0.weight")), sd, "fpn.output1.1"))
        o2 = F.relu(_bn(F.conv2d(feats[1], _t(sd, "fpn.output2.0.weight")), sd, "fpn.output2.1"))
        o3 = F.relu(_bn(F.conv2d(feats[2], _t(sd, "fpn.output3.0.weight")), sd, "fpn.output3.1"))
        o2 = _conv_bn(o2 + F.interpolate(o3, size=[o2.size(2), o2.size(3)], mode="nearest"), sd, "fpn.merge2")
        o1 = _conv_bn(o1 + F.interpolate(o2, size=[o1.size(2), o1.size(3)], mode="nearest"), sd, "fpn.merge1")
        f = [_ssh(o1, sd, "ssh1"), _ssh(o2, sd, "ssh2"), _ssh(o3, sd, "ssh3")]
        locs, confs, ldms = [], [], []
        for i, t in enumerate(f):  # heads: retinaface_trim.py:14-35, permute to NHWC then view(-1, 4|2)
            b = F.conv2d(t, _t(sd, "BboxHead.%d.conv1x1.weight" % i), _t(sd, "BboxHead.%d.conv1x1.bias" % i))
            c = F.conv2d(t, _t(sd, "ClassHead.%d.conv1x1.weight" % i), _t(sd, "ClassHead.%d.conv1x1.bias" % i))
            locs.append(b.permute(0, 2, 3, 1).contiguous().view(b.shape[0], -1, 4))
            confs.append(c.permute(0, 2, 3, 1).contiguous().view(c.shape[0], -1, 2))
            if "LandmarkHead.%d.conv1x1.weight" % i in sd:  # full model, retinaface.py:37-46 (optional mode, D1)
                l = F.conv2d(t, _t(sd, "LandmarkHead.%d.conv1x1.weight" % i), _t(sd, "LandmarkHead.%d.conv1x1.bias" % i))
                ldms.append(l.permute(0, 2, 3, 1).contiguous().view(l.shape[0], -1, 10))
        loc = torch.cat(locs, 1)
        conf = F.softmax(torch.cat(confs, 1), dim=-1)
        out = (loc.numpy(), conf.numpy())
        if ldms:
            out = out + (torch.cat(ldms, 1).numpy(),)
        if return_features:
            out = out + ([t.numpy() for t in feats + [o1, o2, o3] + f],)
        return out


# ----------------------------------------------------------------------------------------------------------------------
# ArcFace IR-50 / IR-SE-50
# ----------------------------------------------------------------------------------------------------------------------
def _units(sd):
    i = 0
    while "body.%d.res_layer.1.weight" % i in sd:
        i += 1
    return i


def ir_unit(x, sd, p, stride):
    """``model_irse.py:48-90`` bottleneck_IR / bottleneck_IR_SE (SE tail iff the fc1 weight exists)."""
    if p + ".shortcut_layer.0.weight" in sd:
        sc = _bn(F.conv2d(x, _t(sd, p + ".shortcut_layer.0.weight"), None, stride), sd, p + ".shortcut_layer.1")
    else:
        sc = F.max_pool2d(x, 1, stride)  # MaxPool2d(1, stride): pure subsampling
    r = _bn(x, sd, p + ".res_layer.0")  # BN *before* the zero-padded conv
    r = F.conv2d(r, _t(sd, p + ".res_layer.1.weight"), None, 1, 1)
    r = F.prelu(r, _t(sd, p + ".res_layer.2.weight"))
    r = F.conv2d(r, _t(sd, p + ".res_layer.3.weight"), None, stride, 1)
    r = _bn(r, sd, p + ".res_layer.4")
    if p + ".res_layer.5.fc1.weight" in sd:  # SEModule, model_irse.py:22-45
        s = F.adaptive_avg_pool2d(r, 1)
        s = F.relu(F.conv2d(s, _t(sd, p + ".res_layer.5.fc1.weight")))
        s = torch.sigmoid(F.conv2d(s, _t(sd, p + ".res_layer.5.fc2.weight")))
        r = r * s
    return r + sc


def ir_strides(n_units):
    from collections import OrderedDict  # noqa: F401

    cfg = {24: (3, 4, 14, 3), 49: (3, 13, 30, 3), 50: (3, 8, 36, 3)}[n_units]
    st = []
    for n in cfg:
        st += [2] + [1] * (n - 1)
    return st


def arcface_forward(sd, x, return_blocks=False):
    """``model_irse.py:166-173`` Backbone.forward (eval).  x float32 [B,3,112,112] planar RGB in [-1,1] -> [B,512]."""
    with torch.no_grad():
        x = torch.as_tensor(x, dtype=torch.float32)
        x = F.conv2d(x, _t(sd, "input_layer.0.weight"), None, 1, 1)
        x = F.prelu(_bn(x, sd, "input_layer.1"), _t(sd, "input_layer.2.weight"))
        blocks = [x.numpy().copy()] if return_blocks else None
        n = _units(sd)
        for i, s in enumerate(ir_strides(n)):
            x = ir_unit(x, sd, "body.%d" % i, s)
            if return_blocks:
                blocks.append(x.numpy().copy())
        x = _bn(x, sd, "output_layer.0")  # Dropout is identity in eval
        x = x.reshape(x.size(0), -1)  # Flatten over NCHW: index c*49 + h*7 + w (model_irse.py:11-13)
        x = F.linear(x, _t(sd, "output_layer.3.weight"), _t(sd, "output_layer.3.bias"))
        pre = F.batch_norm(x, _t(sd, "output_layer.4.running_mean"), _t(sd, "output_layer.4.running_var"),
                           _t(sd, "output_layer.4.weight"), _t(sd, "output_layer.4.bias"), False, 0.0, BN_EPS)
        out = F.normalize(pre, p=2.0, dim=1)  # x / max(||x||, 1e-12), model_irse.py:171
        if return_blocks:
            return out.numpy(), blocks, pre.numpy()
        return out.numpy()
