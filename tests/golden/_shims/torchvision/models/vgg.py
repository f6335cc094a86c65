"""VGG19 'E' feature stack with torchvision's module layout (features[0..36]); random init."""
from torch import nn

_CFG_E = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M",
          512, 512, 512, 512, "M"]


class _VGG(nn.Module):
    def __init__(self):
        super().__init__()
        layers, c = [], 3
        for v in _CFG_E:
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(c, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                c = v
        self.features = nn.Sequential(*layers)


def vgg19(weights=None, **kwargs):
    return _VGG()
