""".flo reader / writer and the end-point-error definition of the reference
(utils/flow_utils.py:23-32 load, :35-65 write_flow, :145-148 EPE), numpy only (the reference needs cv2)."""
import numpy as np

FLO_MAGIC = np.float32(202021.25)


def write_flow(filename, uv):
    """uv: [H, W, 2] float array (u, v) -> Middlebury .flo (magic, width, height, interleaved u,v rows)."""
    uv = np.asarray(uv, dtype=np.float32)
    assert uv.ndim == 3 and uv.shape[2] == 2
    h, w = uv.shape[:2]
    with open(filename, 'wb') as f:
        np.array([FLO_MAGIC], np.float32).tofile(f)
        np.array([w, h], np.int32).tofile(f)
        uv.tofile(f)


def read_flow(filename):
    with open(filename, 'rb') as f:
        magic = np.fromfile(f, np.float32, count=1)
        if magic.size != 1 or magic[0] != FLO_MAGIC:
            raise ValueError('Magic number incorrect. Invalid .flo file')
        w, h = np.fromfile(f, np.int32, count=2)
        data = np.fromfile(f, np.float32, count=2 * int(w) * int(h))
    return data.reshape(int(h), int(w), 2)


def epe(pred, gt):
    """Mean end-point error between two [H, W, 2] flows of the same size."""
    pred, gt = np.asarray(pred, np.float64), np.asarray(gt, np.float64)
    return float(np.sqrt(((pred[..., :2] - gt[..., :2]) ** 2).sum(-1)).mean())
