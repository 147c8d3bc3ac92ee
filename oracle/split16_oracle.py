"""TEST INFRASTRUCTURE (not on the product path): numpy restatement of the operand-splitting arithmetic of
csrc/split16.h / csrc/convsplit.hip, used by tests/ to pin the error bounds the kernels' header comments claim.

A dot product of fp32 vectors is evaluated from 16-bit planes exactly as the kernels do: each operand is (for fp16: scaled by
the power of two derived from its absolute maximum, then) split into planes by round-to-nearest-even, the plane products that
the format keeps are formed exactly and summed; accumulation here is float64, so what is measured is the REPRESENTATION +
dropped-term error of a format, the part that differs from an fp32 FMA chain.
"""
import numpy as np


def _rn_bf16(x):
    """float32 -> nearest-even bfloat16, returned as float32."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def _rn_f16(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def scale_from_absmax(amax):
    """csrc/split16.h: scale_from_absmax -- power of two c with |x| c < 2^14 for |x| <= amax."""
    amax = np.float32(amax)
    e = int((amax.view(np.uint32) >> 23) & 0xFF)
    if e == 0 or e == 255:
        return np.float32(1.0)
    se = min(max(127 + 14 - (e - 126), 1), 254)
    return np.uint32(se << 23).view(np.float32)


def split(x, fmt):
    """Planes of x (list of float32 arrays) and the scale applied before splitting."""
    x = np.asarray(x, np.float32)
    if fmt == "f16x3":
        c = scale_from_absmax(np.abs(x).max() if x.size else 0.0)
        rn, ns = _rn_f16, 2
    else:
        c = np.float32(1.0)
        rn, ns = _rn_bf16, (3 if fmt == "bf16x6" else 2)
    r = (x * c).astype(np.float32)
    planes = []
    for _ in range(ns):
        p = rn(r)
        planes.append(p)
        r = (r - p).astype(np.float32)          # exact in fp32
    return planes, c


def dot(a, b, fmt):
    """sum_k a[k] b[k] from the kept plane products (pa + pb < number of planes), float64 accumulation."""
    pa, ca = split(a, fmt)
    pb, cb = split(b, fmt)
    ns = len(pa)
    acc = 0.0
    for s in range(ns - 1, -1, -1):
        for i in range(s + 1):
            acc += float(np.dot(pa[i].astype(np.float64), pb[s - i].astype(np.float64)))
    return acc / (float(ca) * float(cb))
