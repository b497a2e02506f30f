"""Writers of the file formats a DynaFrame data directory holds (test helpers): uncompressed BMP and the
cv::FileStorage YAML of the calibration (layout of R/Result.yml, numbers supplied by the caller)."""
import struct

import numpy as np


def write_bmp(path, img, bits=8, top_down=False, palette=None):
    """img: uint8 [H, W].  bits 8: paletted (palette: 256 grey levels, default identity; pixel values are palette
    indices); bits 24: B = G = R = img."""
    h, w = img.shape
    if bits == 8:
        pal = np.arange(256, dtype=np.uint8) if palette is None else np.asarray(palette, dtype=np.uint8)
        pal_bytes = b"".join(bytes([p, p, p, 0]) for p in pal)
        row = (w + 3) // 4 * 4
        body = np.zeros((h, row), dtype=np.uint8)
        body[:, :w] = img
    else:
        pal_bytes = b""
        row = (3 * w + 3) // 4 * 4
        body = np.zeros((h, row), dtype=np.uint8)
        body[:, : 3 * w] = np.repeat(img, 3, axis=1)
    if not top_down:
        body = body[::-1]
    off = 14 + 40 + len(pal_bytes)
    with open(path, "wb") as f:
        f.write(b"BM" + struct.pack("<IHHI", off + body.size, 0, 0, off))
        f.write(struct.pack("<IiiHHIIiiII", 40, w, -h if top_down else h, 1, bits, 0, body.size, 2835, 2835,
                            256 if bits == 8 else 0, 0))
        f.write(pal_bytes)
        f.write(body.tobytes())


def _fmt(v):
    s = "%.16e" % v                       # cv::FileStorage style: 1.2138714552009253e+003, integers as "0." / "1."
    if v == int(v) and abs(v) < 10:
        return "%d." % int(v)
    m, e = s.split("e")
    return "%se%s%03d" % (m, e[0], int(e[1:]))


def write_calibration_yaml(path, cam, pro, rot, trans):
    def block(name, rows, cols, data):
        vals = [_fmt(float(x)) for x in data]
        lines, cur = [], "   data: ["
        for i, v in enumerate(vals):
            piece = " " + v + ("," if i + 1 < len(vals) else " ]")
            if len(cur) + len(piece) > 72:
                lines.append(cur)
                cur = "      "
            cur += piece
        lines.append(cur)
        return "%s: !!opencv-matrix\n   rows: %d\n   cols: %d\n   dt: d\n%s\n" % (name, rows, cols, "\n".join(lines))
    with open(path, "w") as f:
        f.write("%YAML:1.0\n")
        f.write(block("CamMat", 3, 3, cam) + block("ProMat", 3, 3, pro) + block("R", 3, 3, rot) + block("T", 3, 1, trans))
