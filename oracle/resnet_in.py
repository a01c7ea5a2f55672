"""Oracle: ResNet-18 trunk with InstanceNorm, as constructed by the reference.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates the published
algorithm of ``torchvision==0.6.1`` ``torchvision/models/resnet.py``
(``ResNet``, ``BasicBlock``; pinned by /root/reference/requirements.txt:10,
constructed at /root/reference/src/models/eye_net.py:48-50 as
``ResNet(block=BasicBlock, layers=[2,2,2,2], num_classes=F,
norm_layer=nn.InstanceNorm2d)`` and run at eye_net.py:106).

The dependency is not vendored under /root/reference and not installed in
this image, and the reference holds no test for it: the trunk arithmetic is
"parity unpinned" by the reference.  What the algorithm is:

  stem    conv 7x7 / stride 2 / pad 3, no bias -> norm -> ReLU -> max-pool 3x3/2 pad 1
  stages  (64, 128, 256, 512) x 2 BasicBlocks; the first block of stages 2-4
          has stride 2 and a ``conv1x1/stride 2 (no bias) -> norm`` down-sample
  block   out = ReLU(norm(conv3x3(ReLU(norm(conv3x3_stride(x))))) + identity)
  head    adaptive avg-pool to 1x1 -> flatten -> Linear(512, num_classes)
  init    conv: Kaiming-normal, fan_out, relu gain; norm layers have no
          parameters here (InstanceNorm2d defaults: affine=False,
          track_running_stats=False, eps=1e-5, biased variance)

Attribute names equal torchvision's so that ``state_dict()`` keys equal the
reference's (conv1, layer{1..4}.{0,1}.conv{1,2}, layer{2..4}.0.downsample.0, fc).
"""
import torch
from torch import nn


def _conv3x3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, norm_layer=None):
        super().__init__()
        if norm_layer is None:
            norm_layer = nn.BatchNorm2d
        self.conv1 = _conv3x3(inplanes, planes, stride)
        self.bn1 = norm_layer(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv3x3(planes, planes)
        self.bn2 = norm_layer(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class ResNet(nn.Module):
    def __init__(self, block=BasicBlock, layers=(2, 2, 2, 2), num_classes=1000,
                 norm_layer=None):
        super().__init__()
        if norm_layer is None:
            norm_layer = nn.BatchNorm2d
        self._norm_layer = norm_layer
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        norm_layer = self._norm_layer
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1,
                          stride=stride, bias=False),
                norm_layer(planes * block.expansion),
            )
        layers = [block(self.inplanes, planes, stride, downsample, norm_layer)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, norm_layer=norm_layer))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = torch.flatten(self.avgpool(x), 1)
        return self.fc(x)
