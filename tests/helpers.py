import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# model configs of the golden cases (must match oracle/gen_golden.py CASES)
from oracle.gen_golden import CASES  # noqa: E402  (pure data + helpers, no reference import)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def golden_state_dict(g):
    return {k[len("param."):]: torch.from_numpy(v.copy()) for k, v in g.items()
            if k.startswith("param.")}


def rel_err(a, b):
    """SURVEY.md §8d parity metric: max|a-b| / max(|b|, tiny) per tensor."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-12))
