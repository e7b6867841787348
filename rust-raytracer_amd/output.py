"""Writing the rendered image: what `img.save("output.png")` does for the reference (src/main.rs:559).

The renderer returns H x W x 3 u8, row 0 = top (render.rs:119, 156-159); this module stores it as an 8-bit
RGB PNG (or a binary PPM) without any imaging library: a PNG is a signature, an IHDR chunk, one zlib stream of
the rows, each prefixed by filter type 0, and an IEND chunk.
"""
import struct
import zlib

import numpy as np


def _chunk(tag, data):
    body = tag + data
    return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)


def png_bytes(rgb8):
    img = np.ascontiguousarray(rgb8, dtype=np.uint8)
    if img.ndim != 3 or img.shape[2] != 3:
        raise ValueError(f"expected H x W x 3 u8, got {img.shape}")
    h, w, _ = img.shape
    rows = np.concatenate([np.zeros((h, 1), dtype=np.uint8), img.reshape(h, w * 3)], axis=1)  # filter type 0 per row
    ihdr = struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)  # 8 bits, colour type 2 (RGB), deflate, no interlace
    return b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", zlib.compress(rows.tobytes(), 6)) + _chunk(b"IEND", b"")


def save_png(path, rgb8):
    with open(path, "wb") as f:
        f.write(png_bytes(rgb8))


def save_ppm(path, rgb8):
    img = np.ascontiguousarray(rgb8, dtype=np.uint8)
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(img.tobytes())


def read_png_rgb8(data):
    """Decoder for the subset png_bytes writes (8-bit RGB, filter 0 or Sub/Up rows, no interlace): used by the tests."""
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG")
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        (n,) = struct.unpack(">I", data[pos:pos + 4])
        tag, body = data[pos + 4:pos + 8], data[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        if zlib.crc32(tag + body) & 0xFFFFFFFF != crc:
            raise ValueError("bad chunk checksum")
        if tag == b"IHDR":
            w, h, depth, colour, _, _, interlace = struct.unpack(">IIBBBBB", body)
            if (depth, colour, interlace) != (8, 2, 0):
                raise ValueError("only 8-bit non-interlaced RGB")
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + w * 3)
    out = np.zeros((h, w * 3), dtype=np.uint8)
    for y in range(h):
        f, row = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if f == 0:
            out[y] = row
        elif f == 2:
            out[y] = (row + (out[y - 1] if y else 0)) & 255
        elif f == 1:
            acc = row.copy()
            for x in range(3, w * 3):
                acc[x] = (acc[x] + acc[x - 3]) & 255
            out[y] = acc
        else:
            raise ValueError("unsupported filter")
    return out.reshape(h, w, 3)
