"""TEST INFRASTRUCTURE ONLY — CPU restatement of baseline-JPEG decoding as the reference obtains it.

The reference decodes every image with ``PIL.Image.open(...).convert('RGB')``
(torchvision ``CocoDetection._load_image``, used by oadp/oake/base.py:53), i.e. Pillow -> libjpeg-turbo
with its defaults: accurate integer IDCT (``jpeg_idct_islow``, jidctint.c), "fancy" (triangle-filter)
chroma upsampling (jdsample.c ``h2v1_fancy_upsample`` / ``h2v2_fancy_upsample``) and the fixed-point
YCbCr -> RGB tables of jdcolor.c.  libjpeg-turbo is a third-party dependency that is not vendored in
/root/reference; this file restates its published algorithm in numpy (slow: pure-Python Huffman loop,
small images only) and is pinned BIT-EXACTLY against the Pillow 12.2 / libjpeg-turbo build in this
image by tests/test_jpeg.py.  Scope: 8-bit baseline sequential DCT (SOF0/SOF1 Huffman), 1 or 3
components, sampling factors 1 or 2, restart intervals; progressive frames (SOF2, jdphuff.c) behind
``progressive_ok``.  Arithmetic coding / CMYK / 12-bit: NotImplementedError.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import struct

import numpy as np

ZIGZAG = np.array([
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,
    7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
    39, 46, 53, 60, 61, 54, 47, 55, 62, 63], dtype=np.int64)  # zigzag index -> natural (row-major) index


class Component:
    def __init__(self, cid: int, h: int, v: int, tq: int) -> None:
        self.cid, self.h, self.v, self.tq = cid, h, v, tq
        self.td = self.ta = 0
        self.pred = 0


class HuffTable:
    """jdhuff.c jpeg_make_d_derived_tbl: canonical codes from BITS / HUFFVAL."""

    def __init__(self, bits: list[int], vals: list[int]) -> None:
        self.lookup: dict[tuple[int, int], int] = {}
        code = 0
        k = 0
        for length in range(1, 17):
            for _ in range(bits[length - 1]):
                self.lookup[(length, code)] = vals[k]
                k += 1
                code += 1
            code <<= 1


class BitReader:
    def __init__(self, data: bytes, pos: int) -> None:
        self.data, self.pos = data, pos
        self.acc = 0
        self.n = 0

    def _fill(self) -> None:
        b = self.data[self.pos] if self.pos < len(self.data) else 0
        if b == 0xFF:
            nxt = self.data[self.pos + 1] if self.pos + 1 < len(self.data) else 0xD9
            if nxt == 0:
                self.pos += 2  # stuffed zero
            else:
                b = 0  # marker: feed zeros (libjpeg does the same after a premature marker)
        else:
            self.pos += 1
        self.acc = (self.acc << 8) | b
        self.n += 8

    def bit(self) -> int:
        if self.n == 0:
            self._fill()
        self.n -= 1
        return (self.acc >> self.n) & 1

    def bits(self, k: int) -> int:
        v = 0
        for _ in range(k):
            v = (v << 1) | self.bit()
        return v

    def decode(self, t: HuffTable) -> int:
        code = 0
        for length in range(1, 17):
            code = (code << 1) | self.bit()
            s = t.lookup.get((length, code))
            if s is not None:
                return s
        raise ValueError('bad Huffman code')

    def restart(self) -> None:
        """Byte-align and consume the RSTn marker."""
        self.n = 0
        self.acc = 0
        while not (self.data[self.pos] == 0xFF and 0xD0 <= self.data[self.pos + 1] <= 0xD7):
            self.pos += 1
        self.pos += 2


def _extend(v: int, s: int) -> int:
    return v if v >= (1 << (s - 1)) else v - (1 << s) + 1


def _decode_scan_baseline(br, comps, planes, dc, ac, mcux, mcuy, ri):
    count = 0
    for my in range(mcuy):
        for mx in range(mcux):
            if ri and count and count % ri == 0:
                br.restart()
                for c in comps:
                    c.pred = 0
            count += 1
            for ci, c in enumerate(comps):
                for by in range(c.v):
                    for bx in range(c.h):
                        blk = planes[ci][my * c.v + by, mx * c.h + bx]
                        s = br.decode(dc[c.td])
                        diff = _extend(br.bits(s), s) if s else 0
                        c.pred += diff
                        blk[0] = c.pred
                        k = 1
                        while k < 64:
                            rs = br.decode(ac[c.ta])
                            r, s = rs >> 4, rs & 15
                            if s == 0:
                                if r == 15:
                                    k += 16
                                    continue
                                break
                            k += r
                            blk[ZIGZAG[k]] = _extend(br.bits(s), s)
                            k += 1


def _decode_scan_progressive(br, scan, comps, planes, dc, ac, width, height, hmax, vmax, ri, ss, se, ah, al):
    """One scan of a progressive frame (ITU T.81 G.1.2 / jdphuff.c): DC first / refine (may be
    interleaved), AC first / refine (always a single component, over that component's own blocks)."""
    for c, _ in scan:
        c.pred = 0
    eobrun = 0
    if len(scan) > 1:
        mcux = -(-width // (8 * hmax))
        mcuy = -(-height // (8 * vmax))
        units = [(mx, my) for my in range(mcuy) for mx in range(mcux)]
    else:
        c = scan[0][0]
        bw = -(-(-(-width * c.h // hmax)) // 8)   # ceil(ceil(W h / hmax) / 8): the component's own blocks
        bh = -(-(-(-height * c.v // vmax)) // 8)
        units = [(bx, by) for by in range(bh) for bx in range(bw)]
    count = 0
    for ux, uy in units:
        if ri and count and count % ri == 0:
            br.restart()
            for c, _ in scan:
                c.pred = 0
            eobrun = 0
        count += 1
        if len(scan) > 1:
            blocks = [(c, planes[ci][uy * c.v + by, ux * c.h + bx]) for c, ci in scan
                      for by in range(c.v) for bx in range(c.h)]
        else:
            c, ci = scan[0]
            blocks = [(c, planes[ci][uy, ux])]
        for c, blk in blocks:
            if ss == 0:
                if ah == 0:
                    s = br.decode(dc[c.td])
                    c.pred += _extend(br.bits(s), s) if s else 0
                    blk[0] = c.pred * (1 << al)
                elif br.bit():
                    blk[0] |= 1 << al
                continue
            if ah == 0:  # AC first
                if eobrun:
                    eobrun -= 1
                    continue
                k = ss
                while k <= se:
                    rs = br.decode(ac[c.ta])
                    r, s = rs >> 4, rs & 15
                    if s == 0:
                        if r == 15:
                            k += 16
                            continue
                        eobrun = (1 << r) - 1 + (br.bits(r) if r else 0)
                        break
                    k += r
                    blk[ZIGZAG[k]] = _extend(br.bits(s), s) * (1 << al)
                    k += 1
                continue
            # AC refinement
            p1, m1 = 1 << al, -1 << al
            k = ss
            if eobrun == 0:
                while k <= se:
                    rs = br.decode(ac[c.ta])
                    r, s = rs >> 4, rs & 15
                    val = 0
                    if s:
                        val = p1 if br.bit() else m1
                    elif r != 15:
                        eobrun = (1 << r) + (br.bits(r) if r else 0)
                        break
                    while k <= se:
                        z = ZIGZAG[k]
                        if blk[z] != 0:
                            if br.bit() and (blk[z] & p1) == 0:
                                blk[z] += p1 if blk[z] >= 0 else m1
                        else:
                            if r == 0:
                                if val:
                                    blk[z] = val
                                k += 1
                                break
                            r -= 1
                        k += 1
            if eobrun > 0:
                while k <= se:
                    z = ZIGZAG[k]
                    if blk[z] != 0 and br.bit() and (blk[z] & p1) == 0:
                        blk[z] += p1 if blk[z] >= 0 else m1
                    k += 1
                eobrun -= 1


def parse(data: bytes, progressive_ok: bool = False):
    """-> (height, width, components, qtables (natural order), coefficient planes per component
    [blocks_y, blocks_x, 64] int32 in natural order (MCU-padded)).  Progressive frames (SOF2) only with
    ``progressive_ok`` (the device decoder's host half implements them; see tests/test_jpeg.py)."""
    assert data[0:2] == b'\xff\xd8', 'not a JPEG'
    pos = 2
    qt: dict[int, np.ndarray] = {}
    dc: dict[int, HuffTable] = {}
    ac: dict[int, HuffTable] = {}
    comps: list[Component] = []
    planes = None
    height = width = 0
    ri = 0
    progressive = False
    while True:
        while pos < len(data) and data[pos] != 0xFF:
            pos += 1
        while pos < len(data) and data[pos] == 0xFF:
            pos += 1
        if pos >= len(data):
            break
        m = data[pos]
        pos += 1
        if m == 0xD9:
            break
        if m in (0x01,) or 0xD0 <= m <= 0xD7:
            continue
        (seg,) = struct.unpack('>H', data[pos:pos + 2])
        body = data[pos + 2:pos + seg]
        if m == 0xDB:
            i = 0
            while i < len(body):
                pq, tq = body[i] >> 4, body[i] & 15
                i += 1
                if pq:
                    vals = np.frombuffer(body[i:i + 128], dtype='>u2').astype(np.int64)
                    i += 128
                else:
                    vals = np.frombuffer(body[i:i + 64], dtype=np.uint8).astype(np.int64)
                    i += 64
                nat = np.zeros(64, np.int64)
                nat[ZIGZAG] = vals
                qt[tq] = nat
        elif m in (0xC0, 0xC1, 0xC2):
            if m == 0xC2:
                if not progressive_ok:
                    raise NotImplementedError('SOF2: progressive JPEG')
                progressive = True
            assert body[0] == 8, 'only 8-bit samples'
            height, width = struct.unpack('>HH', body[1:5])
            for c in range(body[5]):
                cid, hv, tq = body[6 + 3 * c:9 + 3 * c]
                comps.append(Component(cid, hv >> 4, hv & 15, tq))
            if len(comps) not in (1, 3):
                raise NotImplementedError('only grayscale and YCbCr')
            if len(comps) == 1:
                comps[0].h = comps[0].v = 1  # a single-component scan is never interleaved
            hmax = max(c.h for c in comps)
            vmax = max(c.v for c in comps)
            mcux = -(-width // (8 * hmax))
            mcuy = -(-height // (8 * vmax))
            planes = [np.zeros((mcuy * c.v, mcux * c.h, 64), np.int32) for c in comps]
        elif m in (0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
            raise NotImplementedError(f'SOF marker {m:#x}: only Huffman sequential / progressive DCT')
        elif m == 0xC4:
            i = 0
            while i < len(body):
                tc, th = body[i] >> 4, body[i] & 15
                bits = list(body[i + 1:i + 17])
                n = sum(bits)
                vals = list(body[i + 17:i + 17 + n])
                (ac if tc else dc)[th] = HuffTable(bits, vals)
                i += 17 + n
        elif m == 0xDD:
            (ri,) = struct.unpack('>H', body[0:2])
        elif m == 0xDA:
            ns = body[0]
            scan = []
            for k in range(ns):
                cs, tt = body[1 + 2 * k], body[2 + 2 * k]
                ci = next(i for i, c in enumerate(comps) if c.cid == cs)
                comps[ci].td, comps[ci].ta = tt >> 4, tt & 15
                scan.append((comps[ci], ci))
            ss, se, aa = body[1 + 2 * ns], body[2 + 2 * ns], body[3 + 2 * ns]
            br = BitReader(data, pos + seg)
            if not progressive:
                assert ns == len(comps), 'non-interleaved sequential scans are not supported'
                for c in comps:
                    c.pred = 0
                _decode_scan_baseline(br, comps, planes, dc, ac, mcux, mcuy, ri)
                return height, width, comps, qt, planes
            _decode_scan_progressive(br, scan, comps, planes, dc, ac, width, height, hmax, vmax, ri, ss, se,
                                     aa >> 4, aa & 15)
            pos = br.pos  # (the bit reader stops in front of the next marker)
            continue
        pos += seg
    if planes is None or not progressive:
        raise ValueError('no scan found')
    return height, width, comps, qt, planes


# ---- jidctint.c jpeg_idct_islow ----------------------------------------------------------------
CONST_BITS, PASS1_BITS = 13, 2
F_0_298, F_0_390, F_0_541, F_0_765 = 2446, 3196, 4433, 6270
F_0_899, F_1_175, F_1_501, F_1_847 = 7373, 9633, 12299, 15137
F_1_961, F_2_053, F_2_562, F_3_072 = 16069, 16819, 20995, 25172


def _descale(x: np.ndarray, n: int) -> np.ndarray:
    return (x + (1 << (n - 1))) >> n


def _idct_1d(d: list[np.ndarray], shift: int) -> list[np.ndarray]:
    z2, z3 = d[2], d[6]
    z1 = (z2 + z3) * F_0_541
    tmp2 = z1 + z3 * (-F_1_847)
    tmp3 = z1 + z2 * F_0_765
    z2, z3 = d[0], d[4]
    tmp0 = (z2 + z3) << CONST_BITS
    tmp1 = (z2 - z3) << CONST_BITS
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = d[7], d[5], d[3], d[1]
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * F_1_175
    tmp0, tmp1, tmp2, tmp3 = tmp0 * F_0_298, tmp1 * F_2_053, tmp2 * F_3_072, tmp3 * F_1_501
    z1, z2, z3, z4 = z1 * (-F_0_899), z2 * (-F_2_562), z3 * (-F_1_961), z4 * (-F_0_390)
    z3 = z3 + z5
    z4 = z4 + z5
    tmp0, tmp1, tmp2, tmp3 = tmp0 + z1 + z3, tmp1 + z2 + z4, tmp2 + z2 + z3, tmp3 + z1 + z4
    return [_descale(tmp10 + tmp3, shift), _descale(tmp11 + tmp2, shift), _descale(tmp12 + tmp1, shift),
            _descale(tmp13 + tmp0, shift), _descale(tmp13 - tmp0, shift), _descale(tmp12 - tmp1, shift),
            _descale(tmp11 - tmp2, shift), _descale(tmp10 - tmp3, shift)]


def _range_limit(x: np.ndarray) -> np.ndarray:
    """jdmaster.c prepare_range_limit_table, post-IDCT half, indexed with (x & RANGE_MASK)."""
    t = x & 1023
    out = np.where(t < 128, t + 128, np.where(t < 512, 255, np.where(t < 896, 0, t - 896)))
    return out.astype(np.uint8)


def idct_islow(coef: np.ndarray, q: np.ndarray) -> np.ndarray:
    """coef [..., 64] int (natural order), q [64] -> samples [..., 8, 8] uint8."""
    d = (coef.astype(np.int64) * q).reshape(coef.shape[:-1] + (8, 8))
    cols = _idct_1d([d[..., r, :] for r in range(8)], CONST_BITS - PASS1_BITS)  # pass 1: down columns
    ws = np.stack(cols, axis=-2)                                                 # [..., row, col]
    rows = _idct_1d([ws[..., :, c] for c in range(8)], CONST_BITS + PASS1_BITS + 3)  # pass 2: along rows
    return _range_limit(np.stack(rows, axis=-1))


# ---- jdsample.c fancy upsampling ----------------------------------------------------------------
def _h2_fancy(row3: np.ndarray, add_even: int, add_odd: int, shift: int, edge_mul: int) -> np.ndarray:
    """row3 [rows, n] already weighted vertically (or plain samples); horizontal triangle filter."""
    n = row3.shape[1]
    out = np.empty((row3.shape[0], 2 * n), np.int64)
    if n == 1:
        out[:, 0] = (row3[:, 0] * edge_mul + add_even) >> shift
        out[:, 1] = (row3[:, 0] * edge_mul + add_odd) >> shift
        return out
    left = np.concatenate([row3[:, :1], row3[:, :-1]], axis=1)
    right = np.concatenate([row3[:, 1:], row3[:, -1:]], axis=1)
    out[:, 0::2] = (row3 * 3 + left + add_even) >> shift
    out[:, 1::2] = (row3 * 3 + right + add_odd) >> shift
    out[:, 0] = (row3[:, 0] * edge_mul + add_even) >> shift
    out[:, -1] = (row3[:, -1] * edge_mul + add_odd) >> shift
    return out


def upsample_h2v1(p: np.ndarray) -> np.ndarray:
    """h2v1_fancy_upsample: out[2i] = (3 in[i] + in[i-1] + 1) >> 2, out[2i+1] = (3 in[i] + in[i+1] + 2) >> 2,
    first / last output columns copy the edge sample."""
    p = p.astype(np.int64)
    out = _h2_fancy(p, 1, 2, 2, 4)
    out[:, 0] = p[:, 0]
    out[:, -1] = p[:, -1]
    return out.astype(np.uint8)


def upsample_h2v2(p: np.ndarray) -> np.ndarray:
    """h2v2_fancy_upsample: vertical 3:1 blend with the nearer neighbour row (edge rows replicated by
    the main controller's context rows), then the horizontal triangle with roundings 8 / 7."""
    p = p.astype(np.int64)
    up = np.concatenate([p[:1], p[:-1]], axis=0)
    dn = np.concatenate([p[1:], p[-1:]], axis=0)
    out = np.empty((2 * p.shape[0], 2 * p.shape[1]), np.int64)
    out[0::2] = _h2_fancy(p * 3 + up, 8, 7, 4, 4)
    out[1::2] = _h2_fancy(p * 3 + dn, 8, 7, 4, 4)
    return out.astype(np.uint8)


def upsample_h1v2(p: np.ndarray) -> np.ndarray:
    """h1v2_fancy_upsample (libjpeg-turbo): out_upper = (3 in + above + 1) >> 2, out_lower = (3 in + below + 2) >> 2."""
    p = p.astype(np.int64)
    up = np.concatenate([p[:1], p[:-1]], axis=0)
    dn = np.concatenate([p[1:], p[-1:]], axis=0)
    out = np.empty((2 * p.shape[0], p.shape[1]), np.int64)
    out[0::2] = (p * 3 + up + 1) >> 2
    out[1::2] = (p * 3 + dn + 2) >> 2
    return out.astype(np.uint8)


# ---- jdcolor.c ycc_rgb_convert -------------------------------------------------------------------
def ycc_to_rgb(y: np.ndarray, cb: np.ndarray, cr: np.ndarray) -> np.ndarray:
    y = y.astype(np.int64)
    xb = cb.astype(np.int64) - 128
    xr = cr.astype(np.int64) - 128
    r = y + ((91881 * xr + 32768) >> 16)
    b = y + ((116130 * xb + 32768) >> 16)
    g = y + ((-22554 * xb + 32768 - 46802 * xr) >> 16)
    return np.clip(np.stack([r, g, b], axis=-1), 0, 255).astype(np.uint8)


def decode(data: bytes, progressive_ok: bool = False) -> np.ndarray:
    """-> uint8 [H, W, 3], equal to np.asarray(PIL.Image.open(...).convert('RGB'))."""
    height, width, comps, qt, planes = parse(data, progressive_ok)
    hmax = max(c.h for c in comps)
    vmax = max(c.v for c in comps)
    full = []
    for c, coef in zip(comps, planes):
        px = idct_islow(coef, qt[c.tq])                       # [by, bx, 8, 8]
        plane = px.transpose(0, 2, 1, 3).reshape(coef.shape[0] * 8, coef.shape[1] * 8)
        dw = -(-width * c.h // hmax)                          # compptr->downsampled_width / _height
        dh = -(-height * c.v // vmax)
        plane = plane[:dh, :dw]
        fh, fv = hmax // c.h, vmax // c.v
        if (fh, fv) == (1, 1):
            pass
        elif fh == 2 and fv in (1, 2) and dw <= 2:
            # jdsample.c jinit_upsampler: the fancy filters need downsampled_width > 2; narrower components
            # (images up to 4 pixels wide) get h2v1_upsample / h2v2_upsample = plain replication
            plane = np.repeat(np.repeat(plane, fv, axis=0), 2, axis=1)
        elif (fh, fv) == (2, 1):
            plane = upsample_h2v1(plane)
        elif (fh, fv) == (2, 2):
            plane = upsample_h2v2(plane)
        elif (fh, fv) == (1, 2):
            plane = upsample_h1v2(plane)
        else:
            raise NotImplementedError(f'sampling ratio {fh}x{fv}')
        full.append(plane[:height, :width])
    if len(full) == 1:
        return np.repeat(full[0][..., None], 3, axis=-1)
    return ycc_to_rgb(*full)
