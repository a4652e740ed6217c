"""Host-side weight initialisers, same behaviour as the reference's model/unet2d/init_weights.py:5-64
(class-name matching on 'Conv' / 'Linear' / 'BatchNorm').  One-off host work, no kernels involved."""
from torch.nn import init

_CONV_LINEAR = {
    "normal": lambda w: init.normal_(w, 0.0, 0.02),
    "xavier": lambda w: init.xavier_normal_(w, gain=1),
    "kaiming": lambda w: init.kaiming_normal_(w, a=0, mode="fan_in"),
    "orthogonal": lambda w: init.orthogonal_(w, gain=1),
}


def _make(kind):
    def fn(m):
        name = m.__class__.__name__
        if name.find("Conv") != -1 or name.find("Linear") != -1:
            _CONV_LINEAR[kind](m.weight.data)
        elif name.find("BatchNorm") != -1:
            init.normal_(m.weight.data, 1.0, 0.02)
            init.constant_(m.bias.data, 0.0)
    return fn


weights_init_normal = _make("normal")
weights_init_xavier = _make("xavier")
weights_init_kaiming = _make("kaiming")
weights_init_orthogonal = _make("orthogonal")


def init_weights(net, init_type="normal"):
    if init_type not in _CONV_LINEAR:
        raise NotImplementedError("initialization method [%s] is not implemented" % init_type)
    net.apply(_make(init_type))
