"""function.tiffio — a small TIFF reader for the scenes the reference opens with `libtiff.TIFF.open(f).read_image()`
(function/function.py:33-42).  libtiff is not importable in this image, so the reader is written against the TIFF 6.0
/ BigTIFF layouts directly: numpy only, host side, one-off per run (the hot path starts after the scene is in HBM).

Supported: classic TIFF and BigTIFF, both byte orders, strips and tiles, chunky and planar samples, 8/16/32/64-bit
unsigned / signed / float samples, compression none (1), LZW (5), Deflate (8, 32946), PackBits (32773), horizontal
predictor (2).  `read_image` returns `[H, W]` for one sample per pixel and `[H, W, C]` otherwise — libtiff's shapes.
Not supported (raises): JPEG / CCITT compressions, sub-byte samples, floating-point predictor (3).
"""
import struct
import zlib

import numpy as np

_TYPES = {1: ('B', 1), 2: ('c', 1), 3: ('H', 2), 4: ('I', 4), 5: ('II', 8), 6: ('b', 1), 7: ('B', 1), 8: ('h', 2),
          9: ('i', 4), 10: ('ii', 8), 11: ('f', 4), 12: ('d', 8), 13: ('I', 4), 16: ('Q', 8), 17: ('q', 8), 18: ('Q', 8)}


class TiffError(ValueError):
    pass


def _lzw_decode(data):
    """TIFF-flavoured LZW (MSB-first codes, 9..12 bits, ClearCode 256, EOI 257, early change)."""
    out = bytearray()
    table = [bytes([i]) for i in range(256)] + [b'', b'']
    nbits, bitbuf, bitcnt, prev = 9, 0, 0, None
    for byte in data:
        bitbuf = (bitbuf << 8) | byte
        bitcnt += 8
        while bitcnt >= nbits:
            code = (bitbuf >> (bitcnt - nbits)) & ((1 << nbits) - 1)
            bitcnt -= nbits
            if code == 256:
                table = table[:258]
                nbits, prev = 9, None
                continue
            if code == 257:
                return bytes(out)
            if prev is None:
                entry = table[code]
            elif code < len(table):
                entry = table[code]
                table.append(prev + entry[:1])
            elif code == len(table):
                entry = prev + prev[:1]
                table.append(entry)
            else:
                raise TiffError('corrupt LZW stream')
            out += entry
            prev = entry
            if len(table) >= (1 << nbits) - 1 and nbits < 12:
                nbits += 1
    return bytes(out)


def _packbits_decode(data):
    out = bytearray()
    i, n = 0, len(data)
    while i < n:
        c = data[i]
        i += 1
        if c < 128:
            out += data[i:i + c + 1]
            i += c + 1
        elif c > 128:
            out += data[i:i + 1] * (257 - c)
            i += 1
    return bytes(out)


def _decompress(buf, compression):
    if compression == 1:
        return buf
    if compression == 5:
        return _lzw_decode(buf)
    if compression in (8, 32946):
        return zlib.decompress(buf)
    if compression == 32773:
        return _packbits_decode(buf)
    raise TiffError('TIFF compression %d is not supported' % compression)


def _read_ifd(f, bo, big, offset):
    f.seek(offset)
    n = struct.unpack(bo + ('Q' if big else 'H'), f.read(8 if big else 2))[0]
    entry = 20 if big else 12
    raw = f.read(n * entry)
    tags = {}
    for i in range(n):
        e = raw[i * entry:(i + 1) * entry]
        tag, typ = struct.unpack(bo + 'HH', e[:4])
        count = struct.unpack(bo + ('Q' if big else 'I'), e[4:12 if big else 8])[0]
        val = e[12:] if big else e[8:]
        if typ not in _TYPES:
            continue
        fmt, size = _TYPES[typ]
        nbytes = size * count
        if nbytes > len(val):
            pos = struct.unpack(bo + ('Q' if big else 'I'), val)[0]
            here = f.tell()
            f.seek(pos)
            val = f.read(nbytes)
            f.seek(here)
        else:
            val = val[:nbytes]
        if typ == 2:
            tags[tag] = val.rstrip(b'\x00').decode('latin-1')
        elif typ in (5, 10):
            v = struct.unpack(bo + fmt[0] * (2 * count), val)
            tags[tag] = tuple(v[2 * k] / v[2 * k + 1] if v[2 * k + 1] else 0.0 for k in range(count))
        else:
            tags[tag] = struct.unpack(bo + fmt * count, val)
    return tags


def read_image(path):
    """First image of a TIFF file as a numpy array, `[H, W]` or `[H, W, C]`."""
    with open(path, 'rb') as f:
        head = f.read(16)
        if head[:2] == b'II':
            bo = '<'
        elif head[:2] == b'MM':
            bo = '>'
        else:
            raise TiffError('%s is not a TIFF file' % path)
        magic = struct.unpack(bo + 'H', head[2:4])[0]
        if magic == 42:
            big, first = False, struct.unpack(bo + 'I', head[4:8])[0]
        elif magic == 43:
            big, first = True, struct.unpack(bo + 'Q', head[8:16])[0]
        else:
            raise TiffError('%s: bad TIFF magic %d' % (path, magic))
        t = _read_ifd(f, bo, big, first)
        W, H = int(t[256][0]), int(t[257][0])
        spp = int(t.get(277, (1,))[0])
        bits = t.get(258, (1,) * spp)
        if len(set(bits)) != 1 or bits[0] not in (8, 16, 32, 64):
            raise TiffError('unsupported BitsPerSample %s' % (bits,))
        fmt = int(t.get(339, (1,))[0])
        kind = {1: 'u', 2: 'i', 3: 'f', 4: 'u'}.get(fmt)
        if kind is None or (kind == 'f' and bits[0] < 32):
            raise TiffError('unsupported SampleFormat %d / %d bits' % (fmt, bits[0]))
        dt = np.dtype(bo + kind + str(bits[0] // 8))
        compression = int(t.get(259, (1,))[0])
        predictor = int(t.get(317, (1,))[0])
        if predictor not in (1, 2):
            raise TiffError('unsupported TIFF predictor %d' % predictor)
        planar = int(t.get(284, (1,))[0])
        planes = spp if planar == 2 else 1
        chans = 1 if planar == 2 else spp
        out = np.zeros((planes, H, W, chans), dtype=dt.newbyteorder('='))

        def chunk(offset, nbytes, h, w):
            f.seek(offset)
            raw = _decompress(f.read(nbytes), compression)
            a = np.frombuffer(raw, dtype=dt, count=h * w * chans).reshape(h, w, chans)
            if predictor == 2:
                a = np.cumsum(a.astype(dt.newbyteorder('=')), axis=1, dtype=dt.newbyteorder('='))
            return a

        if 322 in t:                                             # tiles
            tw, th = int(t[322][0]), int(t[323][0])
            offs, cnts = t[324], t[325]
            across, down = (W + tw - 1) // tw, (H + th - 1) // th
            for p in range(planes):
                for ty in range(down):
                    for tx in range(across):
                        i = (p * down + ty) * across + tx
                        a = chunk(offs[i], cnts[i], th, tw)
                        hh, ww = min(th, H - ty * th), min(tw, W - tx * tw)
                        out[p, ty * th:ty * th + hh, tx * tw:tx * tw + ww] = a[:hh, :ww]
        else:                                                    # strips
            rps = int(t.get(278, (H,))[0])
            rps = min(rps, H)
            offs, cnts = t[273], t[279]
            per_plane = (H + rps - 1) // rps
            for p in range(planes):
                for s in range(per_plane):
                    i = p * per_plane + s
                    hh = min(rps, H - s * rps)
                    out[p, s * rps:s * rps + hh] = chunk(offs[i], cnts[i], hh, W)
    img = out[0] if planar != 2 else np.concatenate([out[p] for p in range(planes)], axis=2)
    return img[:, :, 0] if img.shape[2] == 1 else img


def write_image(path, array, rows_per_strip=64, compress=False, tile=None):
    """Uncompressed (or Deflate) little-endian TIFF with chunky samples, in strips or (tile=(th, tw), multiples of 16)
    tiles — the writer the tests use to produce `ms4.tif` / `pan.tif` files any TIFF reader opens."""
    a = np.ascontiguousarray(array)
    if a.ndim == 2:
        a = a[:, :, None]
    H, W, C = a.shape
    kind = {'u': 1, 'i': 2, 'f': 3}[a.dtype.kind]
    a = a.astype(a.dtype.newbyteorder('<'))
    if tile is not None:
        th, tw = tile
        strips = []
        for ty in range(0, H, th):
            for tx in range(0, W, tw):
                t = np.zeros((th, tw, C), dtype=a.dtype)
                blk = a[ty:ty + th, tx:tx + tw]
                t[:blk.shape[0], :blk.shape[1]] = blk
                strips.append(t.tobytes())
    else:
        strips = [a[r:r + rows_per_strip].tobytes() for r in range(0, H, rows_per_strip)]
    if compress:
        strips = [zlib.compress(s) for s in strips]
    n = len(strips)
    entries = []
    extra = bytearray()

    def add(tag, typ, values):
        fmt, size = _TYPES[typ]
        data = struct.pack('<' + fmt * len(values), *values)
        entries.append((tag, typ, len(values), data))

    add(256, 4, [W]); add(257, 4, [H]); add(258, 3, [a.dtype.itemsize * 8] * C); add(259, 3, [8 if compress else 1])
    add(262, 3, [1]); add(277, 3, [C]); add(284, 3, [1]); add(339, 3, [kind] * C)
    OFF, CNT = (324, 325) if tile is not None else (273, 279)
    if tile is not None:
        add(322, 4, [tile[1]]); add(323, 4, [tile[0]])
    else:
        add(278, 4, [rows_per_strip])
    add(OFF, 4, [0] * n); add(CNT, 4, [len(s) for s in strips])
    entries.sort(key=lambda e: e[0])
    ifd_off = 8
    ifd_len = 2 + 12 * len(entries) + 4
    extra_off = ifd_off + ifd_len
    # lay out out-of-line values, then the strips
    blobs, pos = {}, extra_off
    for tag, typ, cnt, data in entries:
        if len(data) > 4:
            blobs[tag] = pos
            pos += len(data) + (len(data) & 1)
    strip_off = []
    for s in strips:
        strip_off.append(pos)
        pos += len(s)
    with open(path, 'wb') as f:
        f.write(b'II' + struct.pack('<HI', 42, ifd_off))
        f.write(struct.pack('<H', len(entries)))
        for tag, typ, cnt, data in entries:
            if tag == OFF:
                data = struct.pack('<' + 'I' * n, *strip_off)
            f.write(struct.pack('<HHI', tag, typ, cnt))
            f.write(struct.pack('<I', blobs[tag]) if len(data) > 4 else data.ljust(4, b'\x00'))
        f.write(struct.pack('<I', 0))
        for tag, typ, cnt, data in entries:
            if len(data) > 4:
                if tag == OFF:
                    data = struct.pack('<' + 'I' * n, *strip_off)
                f.write(data + (b'\x00' if len(data) & 1 else b''))
        for s in strips:
            f.write(s)
