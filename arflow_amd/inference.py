#!/usr/bin/env python3
"""Two- / three-frame inference CLI with the contract the reference's README documents
(README.md:32-50:  inference.py -m <ckpt> -s 384 640 -i img1 img2 [img0 img1 img2]); the fork's own
inference.py only drives its probabilistic models.

    python -m arflow_amd.inference -s 384 640 -i a.png b.png [-m ckpt.pth.tar] [-o flow.flo] [--arch pwclite|pwclite_uflow|uflow]

Runs on the GPU through the HIP kernels.  Without -m the network is seeded-random (no checkpoint ships
with the reference: .MISSING_LARGE_BLOBS), which still exercises the whole path.
"""
import argparse

import numpy as np
import torch

from .config import AttrDict
from .flow_io import write_flow
from .models import get_model


def load_image(path, size):
    from PIL import Image
    img = Image.open(path).convert('RGB')
    if size is not None:
        img = img.resize((size[1], size[0]), Image.BILINEAR)
    return torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1)


def build_model(kind, n_frames, ckpt=None, seed=0):
    if kind == 'pwclite':
        cfg = AttrDict(type='pwclite', upsample=True, n_frames=n_frames, reduce_dense=True)
    elif kind == 'pwclite_uflow':
        cfg = AttrDict(type='pwclite_uflow', level_dropout=0.0, feature_norm=True, align_corners=True, warp_pad='zeros',
                       n_frames=2, reduce_dense=False)
    elif kind == 'uflow':
        cfg = AttrDict(type='uflow', level_dropout=0.0, feature_norm=True)
    else:
        raise NotImplementedError(kind)
    model = get_model(cfg)
    if ckpt:
        # same container as utils/torch_utils.py:39-51: {'epoch':..., 'state_dict':...}; tensors only
        blob = torch.load(ckpt, map_location='cpu', weights_only=True)
        state = blob.get('state_dict', blob)
        model.load_state_dict({k.replace('module.', '', 1): v for k, v in state.items()})
    else:
        torch.manual_seed(seed)
        model.init_weights()
    return model.eval()


@torch.no_grad()
def infer(model, frames, device):
    """frames: list of [3,H,W] tensors in [0,1] -> forward flow of the (middle) reference frame, [H,W,2]."""
    x = torch.cat(frames, 0).unsqueeze(0).to(device)
    res = model(x, with_bk=False) if len(frames) == 2 else model(x)
    return res['flows_fw'][0][0].permute(1, 2, 0).float().cpu().numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('-m', '--model', default=None, help='checkpoint (.pth.tar with state_dict)')
    ap.add_argument('-s', '--test_shape', default=[384, 640], type=int, nargs=2)
    ap.add_argument('-i', '--img_list', nargs='+', required=True)
    ap.add_argument('-o', '--output', default=None, help='write the flow as .flo')
    ap.add_argument('--arch', default='pwclite', choices=['pwclite', 'pwclite_uflow', 'uflow'])
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit('arflow_amd.inference needs a GPU: the correlation / warp kernels have no CPU fallback')
    device = torch.device('cuda')
    frames = [load_image(p, args.test_shape) for p in args.img_list]
    model = build_model(args.arch, len(frames), args.model).to(device)
    flow = infer(model, frames, device)
    print('flow %s  mean |u|=%.4f |v|=%.4f' % (flow.shape, np.abs(flow[..., 0]).mean(), np.abs(flow[..., 1]).mean()))
    if args.output:
        write_flow(args.output, flow)


if __name__ == '__main__':
    main()
