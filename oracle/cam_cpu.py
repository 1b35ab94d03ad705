"""CPU restatement (plain PyTorch, fp32) of the WaveCAM ResNet-50 CAM inference path.  TEST INFRASTRUCTURE.

BASELINE config 5 names a "ResNet-38d CAM path" that the reference does not contain; its conv-only
relative in the reference is WaveCAM-TMM2023/net/resnet50_cam.py::CAM (:112-126) over net/resnet50.py (FixedBatchNorm :11-14,
Bottleneck :17-57, ResNet :60-100).  Functional style over a flat parameter dict keyed by the reference's `state_dict` names
(prefix `resnet50.`; the reference registers the same modules a second and third time as stage1..4 / backbone - the duplicates carry
no information).  Pinned against tests/golden/cam_*.npz (oracle/make_golden.py::case_cam imports the reference itself).
Citations are into /root/reference/WaveCAM-TMM2023/."""
import torch
import torch.nn.functional as F

LAYERS = (3, 4, 6, 3)                 # resnet50.py:104
PLANES = (64, 128, 256, 512)


def _bn(x, P, pre):
    """FixedBatchNorm (resnet50.py:11-14): always the running statistics."""
    return F.batch_norm(x, P[pre + "running_mean"], P[pre + "running_var"], P[pre + "weight"], P[pre + "bias"], False, 0.0, 1e-5)


def bottleneck(x, P, pre, stride, dilation, downsample):
    """resnet50.py:35-57."""
    out = F.relu(_bn(F.conv2d(x, P[pre + "conv1.weight"]), P, pre + "bn1."))
    out = F.relu(_bn(F.conv2d(out, P[pre + "conv2.weight"], None, stride, dilation, dilation), P, pre + "bn2."))
    out = _bn(F.conv2d(out, P[pre + "conv3.weight"]), P, pre + "bn3.")
    res = x
    if downsample:
        res = _bn(F.conv2d(x, P[pre + "downsample.0.weight"], None, stride), P, pre + "downsample.1.")
    return F.relu(out + res)


def features(x, P, strides=(2, 2, 2, 1), dilations=(1, 1, 1, 1), pre="resnet50."):
    """Net.stage1..stage4 (resnet50_cam.py:13-22, 33-37) = ResNet.forward up to layer4 (resnet50.py:84-93)."""
    x = F.relu(_bn(F.conv2d(x, P[pre + "conv1.weight"], None, strides[0], 3), P, pre + "bn1."))
    x = F.max_pool2d(x, 3, 2, 1)
    inplanes = 64
    for li, (planes, blocks) in enumerate(zip(PLANES, LAYERS)):
        stride = 1 if li == 0 else strides[li]
        for b in range(blocks):
            first = b == 0
            ds = first and (stride != 1 or inplanes != planes * 4)
            # resnet50.py:76-80: the first block of a layer is built with dilation 1, the others with the layer's dilation
            x = bottleneck(x, P, f"{pre}layer{li + 1}.{b}.", stride if first else 1, 1 if first else dilations[li], ds)
            inplanes = planes * 4
    return x


def cam_forward(x, P, separate=False):
    """CAM.forward (resnet50_cam.py:117-126): x = [image, horizontally flipped image]; class activation maps of the pair,
    ReLU, sum of the first with the un-flipped second."""
    f = features(x, P)
    c = F.conv2d(f, P["classifier.weight"])
    if separate:
        return c
    c = F.relu(c)
    return c[0] + c[1].flip(-1)


def cam_template(n_classes=20):
    """name -> zero tensor of the reference's shapes (the `resnet50.` copy of the parameters + classifier)."""
    t = {}

    def bn(pre, c):
        t[pre + "weight"] = torch.zeros(c); t[pre + "bias"] = torch.zeros(c)
        t[pre + "running_mean"] = torch.zeros(c); t[pre + "running_var"] = torch.ones(c)
        t[pre + "num_batches_tracked"] = torch.zeros((), dtype=torch.long)

    pre = "resnet50."
    t[pre + "conv1.weight"] = torch.zeros(64, 3, 7, 7)
    bn(pre + "bn1.", 64)
    inplanes = 64
    for li, (planes, blocks) in enumerate(zip(PLANES, LAYERS)):
        for b in range(blocks):
            q = f"{pre}layer{li + 1}.{b}."
            t[q + "conv1.weight"] = torch.zeros(planes, inplanes, 1, 1); bn(q + "bn1.", planes)
            t[q + "conv2.weight"] = torch.zeros(planes, planes, 3, 3); bn(q + "bn2.", planes)
            t[q + "conv3.weight"] = torch.zeros(planes * 4, planes, 1, 1); bn(q + "bn3.", planes * 4)
            if b == 0:
                t[q + "downsample.0.weight"] = torch.zeros(planes * 4, inplanes, 1, 1); bn(q + "downsample.1.", planes * 4)
            inplanes = planes * 4
    t["classifier.weight"] = torch.zeros(n_classes, 2048, 1, 1)
    return t
