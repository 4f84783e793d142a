"""Bitstream container + image padding helpers (reference contract: lvae/utils/coding.py:26-91)."""
import math
import struct

import numpy as np


def pack_byte_strings(list_of_strings):
    """'B' count, then count x 'I' lengths, then the concatenated payloads (native byte order, coding.py:26-47)."""
    lengths = [len(s) for s in list_of_strings]
    head = struct.pack('B', len(lengths)) + struct.pack(f'{len(lengths)}I', *lengths)
    return head + b''.join(list_of_strings)


def unpack_byte_string(string):
    """Inverse of pack_byte_strings (coding.py:50-70); asserts that the lengths add up."""
    num = struct.unpack('B', string[:1])[0]
    lengths = struct.unpack(f'{num}I', string[1:1 + 4 * num])
    body = string[1 + 4 * num:]
    assert sum(lengths) == len(body), f'{sum(lengths)=} should equal to {len(body)=}'
    out, o = [], 0
    for n in lengths:
        out.append(body[o:o + n])
        o += n
    return out


def pad_divisible_by(img, div=64):
    """Replicate-pad a PIL image on the right/bottom so both sides are multiples of `div` (coding.py:73-91)."""
    from PIL import Image
    h, w = img.height, img.width
    if h % div == 0 and w % div == 0:
        return img
    ht, wt = div * math.ceil(h / div), div * math.ceil(w / div)
    a = np.asarray(img)
    pw = ((0, ht - h), (0, wt - w)) + (((0, 0),) if a.ndim == 3 else ())
    return Image.fromarray(np.pad(a, pw, mode='edge'))


def pil_to_tensor01(img):
    """torchvision.transforms.functional.to_tensor for uint8 PIL images: HWC uint8 -> CHW float32 / 255."""
    import torch
    a = np.asarray(img)
    if a.ndim == 2:
        a = a[:, :, None]
    t = torch.from_numpy(np.array(a, copy=True)).permute(2, 0, 1).contiguous()
    return t.to(dtype=torch.float32).div(255)
