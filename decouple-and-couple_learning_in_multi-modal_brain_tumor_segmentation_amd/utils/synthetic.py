"""Deterministic, generator-defined weights and BraTS-shaped synthetic inputs.

The reference ships neither weights nor its ``data/`` package (SURVEY.md F1, §2 "Data
pipeline"), and 67 MB of weights cannot be committed as fixtures.  Every party that needs
the *same* numbers (the imported reference in ``oracle/make_golden.py``, the CPU oracle,
the HIP product path, ``bench.py``) therefore regenerates them from this counter-based
generator: value i of tensor ``name`` is a pure function of (crc32(name), i), computed in
uint64 numpy arithmetic (splitmix64), so it is bit-identical on every box.

Also provides the synthetic sample generator standing in for ``data.ClsWiseBraTS128``
(imported at train_no_amp.py:20, absent from the reference): x [4,D,H,W] float32,
target int64 in {0..3}, edge code int64 in {0,1,2,4,5,6,7,8}.  The edge encoding is
inferred from tools.get_edge_separate_loss (utils/tools.py:174-218): E1={1,5,6,7},
E2={2,5,6,8}, E4={4,5,7,8} => 5 = all three, 6 = 1&2, 7 = 1&4, 8 = 2&4.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Iterable, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform01(name: str, n: int, salt: int = 0) -> np.ndarray:
    """n float64 values in [0,1), exactly representable in float32 (24-bit mantissa)."""
    seed = np.uint64(zlib.crc32(name.encode()) + (salt << 32))
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + seed * np.uint64(0x2545F4914F6CDD1D)
    bits = _splitmix64(ctr) >> np.uint64(40)
    return bits.astype(np.float64) / float(1 << 24)


def det_tensor(name: str, shape: Sequence[int], scale: float, offset: float = 0.0, salt: int = 0) -> torch.Tensor:
    n = int(np.prod(shape))
    u = uniform01(name, n, salt) * 2.0 - 1.0
    return torch.from_numpy((u * scale + offset).astype(np.float32)).reshape(tuple(shape))


def det_state_dict(shapes: Iterable[Tuple[str, Tuple[int, ...], bool]], salt: int = 0) -> Dict[str, torch.Tensor]:
    """Deterministic weights for a list of (name, shape, is_buffer) (buffers are skipped:
    the four ``pe`` tables are formula-defined).  Magnitudes follow PyTorch's default
    inits: U(+-1/sqrt(fan_in)) for conv/linear weights and biases, LayerNorm weight
    1 +- 0.1, tokens +- 0.04."""
    shapes = list(shapes)
    fan_in = {}
    for name, shape, _ in shapes:
        if name.endswith(".weight") and len(shape) >= 2:
            fi = int(np.prod(shape[1:]))
            if "DeUp" in name and name.endswith("conv2.weight"):   # ConvTranspose3d: [Cin,Cout,k,k,k]
                fi = int(shape[1] * np.prod(shape[2:]))
            fan_in[name[:-7]] = fi
    out = {}
    for name, shape, is_buf in shapes:
        if is_buf:
            continue
        if "token" in name:
            out[name] = det_tensor(name, shape, 0.04, salt=salt)
        elif ".norm" in name:
            out[name] = det_tensor(name, shape, 0.1, 1.0 if name.endswith("weight") else 0.0, salt=salt)
        else:
            base = name.rsplit(".", 1)[0]
            out[name] = det_tensor(name, shape, 1.0 / math.sqrt(fan_in[base]), salt=salt)
    return out


def _dilate(m: torch.Tensor) -> torch.Tensor:
    return F.max_pool3d(m[None, None].float(), 3, 1, 1)[0, 0] > 0


def edge_codes(target: torch.Tensor) -> torch.Tensor:
    """Edge-code volume from a label volume [D,H,W]: band_k = dilate(R_k) & ~erode(R_k)
    for the three sub-regions R_1={t==1}, R_2={t==2}, R_4={t==3}; code by membership."""
    bands = []
    for k in (1, 2, 3):
        r = target == k
        er = ~_dilate(~r)
        bands.append(_dilate(r) & ~er)
    b1, b2, b4 = bands
    code = torch.zeros_like(target)
    code[b1] = 1
    code[b2] = 2
    code[b4] = 4
    code[b1 & b2] = 6
    code[b1 & b4] = 7
    code[b2 & b4] = 8
    code[b1 & b2 & b4] = 5
    return code


def synthetic_sample(index: int, size: Sequence[int] = (128, 128, 128), seed: int = 1000):
    """One BraTS-shaped sample: (x float32 [4,D,H,W], target int64 [D,H,W], edge int64 [D,H,W]).
    x is unit-variance noise plus a class-dependent offset; the target is three nested
    ellipsoids (background >> tumour, as in BraTS, which the CE class weights depend on)."""
    d, h, w = size
    name = "sample%d_%d" % (seed, index)
    u = uniform01(name + "_c", 8)
    cz, cy, cx = (0.35 + 0.3 * u[0]) * d, (0.35 + 0.3 * u[1]) * h, (0.35 + 0.3 * u[2]) * w
    rad = (0.22 + 0.1 * u[3]) * min(d, h, w)
    zz, yy, xx = torch.meshgrid(torch.arange(d), torch.arange(h), torch.arange(w), indexing="ij")
    rr = torch.sqrt(((zz - cz) / 1.0) ** 2 + ((yy - cy) / 1.2) ** 2 + ((xx - cx) / 0.9) ** 2)
    target = torch.zeros(size, dtype=torch.int64)
    target[rr < rad] = 2            # oedema shell
    target[rr < 0.62 * rad] = 1     # core
    target[rr < 0.35 * rad] = 3     # enhancing
    n = 4 * d * h * w
    x = torch.from_numpy(((uniform01(name + "_x", n) * 2.0 - 1.0) * math.sqrt(3.0)).astype(np.float32)).reshape(4, d, h, w)
    x = x + 0.5 * target.float()[None] * torch.tensor([1.0, -1.0, 0.5, 0.25]).reshape(4, 1, 1, 1)
    return x, target, edge_codes(target)


def synthetic_batch(indices: Sequence[int], size=(128, 128, 128), seed: int = 1000):
    xs, ts, es = zip(*(synthetic_sample(i, size, seed) for i in indices))
    return torch.stack(xs), torch.stack(ts), torch.stack(es)
