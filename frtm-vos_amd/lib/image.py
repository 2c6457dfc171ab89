"""Image I/O and the affine warp front-end (names of the reference's lib/image.py:17-59).
I/O is outside the hot path; warp_affine dispatches to the HIP kernel (csrc/image_ops.hip) --
the reference dispatches to OpenCV on CPU and to its NPP extension on CUDA."""
import ctypes

import numpy as np
import torch

from .. import _hip as H

davis_palette = np.repeat(np.arange(256, dtype=np.uint8)[:, None], 3, 1)
davis_palette[:22] = [[0, 0, 0], [128, 0, 0], [0, 128, 0], [128, 128, 0], [0, 0, 128], [128, 0, 128], [0, 128, 128],
                      [128, 128, 128], [64, 0, 0], [191, 0, 0], [64, 128, 0], [191, 128, 0], [64, 0, 128], [191, 0, 128],
                      [64, 128, 128], [191, 128, 128], [0, 64, 0], [128, 64, 0], [0, 191, 0], [128, 191, 0], [0, 64, 128],
                      [128, 64, 128]]

_MODES = {'nearest': 0, 'bilinear': 1, 'bicubic': 2}


def imread(filename):
    from PIL import Image
    im = np.atleast_3d(np.array(Image.open(filename))).transpose(2, 0, 1)
    return torch.from_numpy(np.ascontiguousarray(im))


def imwrite(filename, im):
    from PIL import Image
    assert im.dim() < 4 or im.shape[0] == 1
    Image.fromarray(im.detach().cpu().reshape(-1, *im.shape[-2:]).permute(1, 2, 0).numpy()).save(filename)


def imwrite_indexed(filename, im, color_palette=None):
    from PIL import Image
    assert im.dim() < 4 or im.shape[0] == 1
    pal = davis_palette if color_palette is None else color_palette
    out = Image.fromarray(im.detach().cpu().squeeze().numpy().astype(np.uint8), 'P')
    out.putpalette(pal.ravel())
    out.save(filename)


def warp_affine(src, Hm, size, mode='bicubic'):
    """src: (C,H,W) or (H,W) tensor on the GPU, float32 or uint8 (the two types the reference's extension takes, lib/_npp/nppig.cpp:94-104;
    other types are warped as float32); Hm: 3x3 (or 2x3) FORWARD transform source -> destination, pixel centres at integer coordinates;
    size: (H,W) of the result; taps outside the source read 0 (reference lib/image.py:38-59).  Returns the source's type."""
    assert src.dim() < 4 or src.shape[0] == 1
    H.require_gpu(src, 'warp_affine')
    no_cdim = src.dim() == 2
    u8 = src.dtype == torch.uint8
    s = src.reshape(-1, *src.shape[-2:])
    s = s.contiguous() if u8 else s.float().contiguous()
    dst = torch.empty(s.shape[0], int(size[0]), int(size[1]), device=s.device, dtype=torch.uint8 if u8 else torch.float32)
    m = (ctypes.c_float * 6)(*[float(v) for v in np.asarray(Hm, dtype=np.float32)[:2].ravel()])
    H.call('frtm_warp_affine_u8' if u8 else 'frtm_warp_affine', s.data_ptr(), s.shape[0], s.shape[1], s.shape[2], dst.data_ptr(),
           dst.shape[1], dst.shape[2], m, _MODES[mode])
    return dst.squeeze(0) if no_cdim else dst
