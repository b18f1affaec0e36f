"""numpy restatements of the int8 engine's activation schemes, shared by the CPU and GPU tests."""
import numpy as np


def dynamic_uint8_product(A, W):
    """numpy restatement of ONNX DynamicQuantizeLinear (per row of A) + MatMulInteger against W as int8 per output row,
    scaled once: the arithmetic of a dynamically quantised ONNX MatMul.  Returns (y, q, zp, s)."""
    A = np.asarray(A, np.float32)
    W = np.asarray(W, np.float32)
    mw = np.abs(W).max(axis=1)
    invw = np.where(mw > 0, np.float32(127.0) / np.where(mw > 0, mw, 1), 0).astype(np.float32)
    qw = np.clip(np.rint(W * invw[:, None]), -127, 127).astype(np.int64)
    sw = (mw / np.float32(127.0)).astype(np.float32)
    lo = np.minimum(A.min(axis=1), 0).astype(np.float32)
    hi = np.maximum(A.max(axis=1), 0).astype(np.float32)
    s = ((hi - lo) / np.float32(255.0)).astype(np.float32)
    ok = s > 0
    ss = np.where(ok, s, 1).astype(np.float32)
    zp = np.where(ok, np.rint(np.clip((-lo / ss).astype(np.float32), 0, 255)), 0).astype(np.int64)
    q = np.where(ok[:, None], np.clip(np.rint((A / ss[:, None]).astype(np.float32)).astype(np.int64) + zp[:, None], 0, 255), 0)
    isum = (q - zp[:, None]) @ qw.T
    y = isum.astype(np.float32) * (np.where(ok, s, 0).astype(np.float32)[:, None] * sw[None, :]).astype(np.float32)
    return y.astype(np.float32), q, zp, s
