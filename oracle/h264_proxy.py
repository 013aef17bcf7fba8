"""numpy restatement of the H.264-style transform-coding proxy (TEST INFRASTRUCTURE, see oracle/__init__.py).

PARITY UNPINNED with respect to the reference: augmentation/video.py:20-119 shells out to libx264 / libx265 through PyAV; neither
the codec source nor a PyAV build is available offline, so there is nothing to pin this against.  This file
DEFINES the proxy that videoseal_amd/csrc/h264_proxy.hip implements (integer arithmetic -> the HIP kernel must match bit for bit):

  clamp -> uint8 by truncation (video.py:38-40) -> [yuv420] integer BT.601 limited-range YCbCr, 2x2 rounded-mean chroma ->
  per 4x4 block: residual vs a flat 128 prediction -> H.264 forward core transform -> quantisation at QP (MF table, intra dead zone
  2^qbits/3) -> de-quantisation (V table) -> inverse core transform ((x+32)>>6) -> +128, clip -> nearest chroma up-sampling,
  integer YCbCr -> RGB -> /255.   QP = clamp(crf, 0, 51); chroma QP from the standard's table.  No prediction, no deblocking.
"""
import numpy as np

MF = np.array([[13107, 5243, 8066], [11916, 4660, 7490], [10082, 4194, 6554], [9362, 3647, 5825], [8192, 3355, 5243], [7282, 2893, 4559]], dtype=np.int64)
V = np.array([[10, 16, 13], [11, 18, 14], [13, 20, 16], [14, 23, 18], [16, 25, 20], [18, 29, 23]], dtype=np.int64)
QPC = [29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39]
CF = np.array([[1, 1, 1, 1], [2, 1, -1, -2], [1, -1, -1, 1], [1, -2, 2, -1]], dtype=np.int64)
_a, _c = np.arange(4)[:, None] & 1, np.arange(4)[None, :] & 1
CLS = np.where((_a == 0) & (_c == 0), 0, np.where((_a == 1) & (_c == 1), 1, 2))


def _inv_1d(w, axis):
    w0, w1, w2, w3 = [np.take(w, i, axis=axis) for i in range(4)]
    e0, e1, e2, e3 = w0 + w2, w0 - w2, (w1 >> 1) - w3, w1 + (w3 >> 1)
    return np.stack([e0 + e3, e1 + e2, e1 - e2, e0 - e3], axis=axis)


def block_tq(plane: np.ndarray, qp: int) -> np.ndarray:
    """plane: uint8 [..., h, w] with h, w multiples of 4 -> reconstructed uint8 plane."""
    h, w = plane.shape[-2:]
    x = plane.astype(np.int64).reshape(plane.shape[:-2] + (h // 4, 4, w // 4, 4)).swapaxes(-3, -2) - 128     # [..., by, bx, 4, 4]
    W = CF @ x @ CF.T
    qm, qd = qp % 6, qp // 6
    qbits = 15 + qd
    f = (1 << qbits) // 3
    z = (np.abs(W) * MF[qm][CLS] + f) >> qbits
    dq = np.sign(W) * ((z * V[qm][CLS]) << qd)
    r = _inv_1d(_inv_1d(dq, -1), -2)
    out = np.clip(((r + 32) >> 6) + 128, 0, 255)
    return out.swapaxes(-3, -2).reshape(plane.shape).astype(np.uint8)


def roundtrip(frames: np.ndarray, crf: int, rgb_mode: bool = False) -> np.ndarray:
    """frames float32 [F, 3, H, W] -> float32 [F, 3, H, W]."""
    qp = int(min(max(int(crf), 0), 51))
    F, _, H, Wd = frames.shape
    H8, W8 = (H + 7) // 8 * 8, (Wd + 7) // 8 * 8
    u = (np.clip(frames, 0.0, 1.0).astype(np.float32) * np.float32(255.0)).astype(np.int64)          # truncation
    u = np.pad(u, ((0, 0), (0, 0), (0, H8 - H), (0, W8 - Wd)), mode="edge")
    R, G, B = u[:, 0], u[:, 1], u[:, 2]
    if rgb_mode:
        rec = np.stack([block_tq(p.astype(np.uint8), qp) for p in (R, G, B)], 1).astype(np.int64)
    else:
        Y = ((66 * R + 129 * G + 25 * B + 128) >> 8) + 16
        cb = ((-38 * R - 74 * G + 112 * B + 128) >> 8) + 128
        cr = ((112 * R - 94 * G - 18 * B + 128) >> 8) + 128
        pool = lambda p: (p.reshape(F, H8 // 2, 2, W8 // 2, 2).sum(axis=(2, 4)) + 2) >> 2
        qpc = qp if qp < 30 else QPC[qp - 30]
        Yr = block_tq(Y.astype(np.uint8), qp).astype(np.int64)
        Cb = block_tq(pool(cb).astype(np.uint8), qpc).astype(np.int64).repeat(2, axis=1).repeat(2, axis=2)
        Cr = block_tq(pool(cr).astype(np.uint8), qpc).astype(np.int64).repeat(2, axis=1).repeat(2, axis=2)
        C, D, E = Yr - 16, Cb - 128, Cr - 128
        rec = np.stack([np.clip((298 * C + 409 * E + 128) >> 8, 0, 255), np.clip((298 * C - 100 * D - 208 * E + 128) >> 8, 0, 255),
                        np.clip((298 * C + 516 * D + 128) >> 8, 0, 255)], 1)
    return (rec[:, :, :H, :Wd].astype(np.float32) / np.float32(255.0)).astype(np.float32)
