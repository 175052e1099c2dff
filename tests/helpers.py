"""Shared helpers for the test-suite (fixture loading, error metrics)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


def load_meta():
    with open(os.path.join(GOLDEN, "meta.json")) as f:
        return json.load(f)


def sub_sd(blob, prefix):
    """Tensors stored under ``prefix`` (e.g. 'conv0/sd/') as a torch state dict."""
    return {k[len(prefix):]: torch.from_numpy(np.asarray(v)) for k, v in blob.items()
            if k.startswith(prefix)}


def rms(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float(((a - b) ** 2).mean().sqrt())


def max_abs(a, b):
    return float((torch.as_tensor(a, dtype=torch.float64) - torch.as_tensor(b, dtype=torch.float64)).abs().max())
