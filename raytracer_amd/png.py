"""Minimal PNG reader (8-bit greyscale / RGB / palette / RGBA, non-interlaced) with zlib only: enough for the reference's texture
files (GLUtils.cpp:148-209 loads them through SDL_image and converts to 8-bit RGBA)."""
import struct
import zlib

import numpy as np


def read_png_rgba(path: str) -> np.ndarray:
    """Returns [height, width, 4] uint8, row 0 = top of the image."""
    data = open(path, "rb").read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG file: %s" % path)
    pos, idat, palette, trns = 8, [], None, None
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if kind == b"IHDR":
            width, height, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
        elif kind == b"PLTE":
            palette = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif kind == b"tRNS":
            trns = np.frombuffer(body, np.uint8)
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
    if depth != 8 or interlace != 0:
        raise ValueError("only 8-bit non-interlaced PNG files are supported")
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(height, 1 + width * channels)
    out = np.zeros((height, width * channels), np.uint8)
    bpp = channels
    prev = np.zeros(width * channels, np.int32)
    for r in range(height):
        f, line = int(raw[r, 0]), raw[r, 1:].astype(np.int32)
        if f == 0:
            cur = line
        elif f == 2:
            cur = (line + prev) & 255
        elif f == 1:
            cur = line.reshape(-1, bpp).cumsum(0).reshape(-1) & 255      # every channel accumulates along the row
        else:
            cur = np.zeros_like(line)
            for i in range(len(line)):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if f == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (line[i] + pred) & 255
        out[r] = cur
        prev = cur
    px = out.reshape(height, width, channels)
    rgba = np.full((height, width, 4), 255, np.uint8)
    if ctype == 6:
        rgba = px.copy()
    elif ctype == 2:
        rgba[..., :3] = px
    elif ctype == 0:
        rgba[..., :3] = px
    elif ctype == 4:
        rgba[..., :3] = px[..., :1]; rgba[..., 3] = px[..., 1]
    else:
        rgba[..., :3] = palette[px[..., 0]]
        if trns is not None:
            a = np.full(256, 255, np.uint8); a[:len(trns)] = trns
            rgba[..., 3] = a[px[..., 0]]
    return rgba
