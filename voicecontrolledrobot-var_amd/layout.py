"""Parameter arena layout = state_dict() registration order of the reference Kuka
VARPretextNet (models/pretext/arm_pretext_model.py:39-56), PyTorch shapes."""
PARAM_SPECS = [
    ("imgBranch.0.weight", (32, 3, 3, 3)), ("imgBranch.0.bias", (32,)),
    ("imgBranch.2.weight", (32, 32, 3, 3)), ("imgBranch.2.bias", (32,)),
    ("imgBranch.4.weight", (64, 32, 3, 3)), ("imgBranch.4.bias", (64,)),
    ("imgBranch.6.weight", (64, 64, 3, 3)), ("imgBranch.6.bias", (64,)),
    ("imgBranch.8.weight", (64, 64, 3, 3)), ("imgBranch.8.bias", (64,)),
    ("soundCNN.0.weight", (32, 1, 5, 40)), ("soundCNN.0.bias", (32,)),
    ("soundCNN.2.weight", (32, 32, 3, 1)), ("soundCNN.2.bias", (32,)),
    ("soundCNN.4.weight", (32, 32, 3, 1)), ("soundCNN.4.bias", (32,)),
    ("soundCNN.6.weight", (32, 32, 3, 1)), ("soundCNN.6.bias", (32,)),
    ("imgTriplet.0.weight", (128, 576)), ("imgTriplet.0.bias", (128,)),
    ("imgTriplet.2.weight", (3, 128)), ("imgTriplet.2.bias", (3,)),
    ("soundTriplet.0.weight", (128, 160)), ("soundTriplet.0.bias", (128,)),
    ("soundTriplet.2.weight", (3, 128)), ("soundTriplet.2.bias", (3,)),
]


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


PARAM_OFFSETS = []
_o = 0
for _name, _shape in PARAM_SPECS:
    PARAM_OFFSETS.append(_o)
    _o += _numel(_shape)
N_PARAMS = _o
assert N_PARAMS == 213478
