import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name))
    return {k: torch.from_numpy(d[k]) for k in d.files}


def sub(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def rel(a, b):
    """max-norm relative error max|a-b| / max|b| (SURVEY.md section 7 'Parity at 1e-5')."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    s = b.abs().max().item() if b.numel() else 0.0
    d = (a - b).abs().max().item() if b.numel() else 0.0
    return d / s if s > 0 else d


def is_zero_grad(name):
    """Last conv bias of every MlpBlock_Real: analytically zero gradient (GraphNorm removes the
    mean), pure rounding noise in fp32."""
    return name.endswith('convs.2.bias')
