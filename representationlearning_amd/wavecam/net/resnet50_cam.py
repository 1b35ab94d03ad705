"""Class-activation-map networks of the WaveCAM code-drop (reference: WaveCAM-TMM2023/net/resnet50_cam.py:9-126): `Net` (the
classifier used for training the CAMs) and `CAM` (inference: CAMs of an image and its horizontal flip, summed).  Same module tree -
including the reference's habit of registering the backbone modules three times (resnet50.*, stage1..4.*, backbone.*) - hence the
same `state_dict` keys."""
import torch
import torch.nn as nn

from ... import nnf
from . import resnet50


class _Stage1(nn.Sequential):
    """nn.Sequential(conv1, bn1, relu, maxpool, layer1) of the reference (:13-14): the first three run as one fused op."""

    def forward(self, x):
        conv1, bn1, _, maxpool, layer1 = self
        if torch.is_autocast_enabled():
            x = x.to(torch.get_autocast_dtype("cuda"))
        x = x.contiguous(memory_format=torch.channels_last)
        return layer1(maxpool(nnf.conv_bn_act(x, conv1, bn1, nnf.ACT_RELU)))


class Net(nn.Module):
    def __init__(self, stride=16, n_classes=20, pretrained=False, weight_path=None):
        super().__init__()
        if stride == 16:
            self.resnet50 = resnet50.resnet50(pretrained=pretrained, weight_path=weight_path, strides=(2, 2, 2, 1))
        else:
            self.resnet50 = resnet50.resnet50(pretrained=pretrained, weight_path=weight_path, strides=(2, 2, 1, 1), dilations=(1, 1, 2, 2))
        r = self.resnet50
        self.stage1 = _Stage1(r.conv1, r.bn1, r.relu, r.maxpool, r.layer1)
        self.stage2 = nn.Sequential(r.layer2)
        self.stage3 = nn.Sequential(r.layer3)
        self.stage4 = nn.Sequential(r.layer4)
        self.n_classes = n_classes
        self.classifier = nn.Conv2d(2048, n_classes, 1, bias=False)
        self.bg = nn.Conv2d(2048, n_classes, 1, bias=False)
        self.backbone = nn.ModuleList([self.stage1, self.stage2, self.stage3, self.stage4])
        self.newly_added = nn.ModuleList([self.classifier])

    def features(self, x):
        return self.stage4(self.stage3(self.stage2(self.stage1(x))))

    def forward(self, x):
        f = self.features(x)
        pooled = f.float().mean((2, 3), keepdim=True)                     # torchutils.gap2d(x, keepdims=True)
        return nnf.conv_bias(pooled.to(f.dtype).contiguous(memory_format=torch.channels_last), self.classifier).float().view(-1, self.n_classes)

    def train(self, mode=True):
        super().train(mode)
        for p in list(self.resnet50.conv1.parameters()) + list(self.resnet50.bn1.parameters()):
            p.requires_grad = False

    def trainable_parameters(self):
        return (list(self.backbone.parameters()), list(self.newly_added.parameters()))


class CAM(Net):
    def forward(self, x, separate=False):
        x = nnf.conv_bias(self.features(x), self.classifier).float()
        if separate:
            return x
        x = torch.relu(x)
        return x[0] + x[1].flip(-1)
