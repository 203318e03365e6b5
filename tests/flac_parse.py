"""A small FLAC frame parser / decoder written from the format specification (RFC 9639), used
only by the tests: it is independent of the oracle's writer and of the kernels, so frame bytes that
it decodes back to the input samples (with both CRCs matching) are valid FLAC frames."""
import numpy as np

FIXED_COEFS = [[], [1], [2, -1], [3, -3, 1], [4, -6, 4, -1]]
BLOCK_SIZES = {1: 192, 2: 576, 3: 1152, 4: 2304, 5: 4608, 8: 256, 9: 512, 10: 1024, 11: 2048, 12: 4096,
               13: 8192, 14: 16384, 15: 32768}
SAMPLE_RATES = {1: 88200, 2: 176400, 3: 192000, 4: 8000, 5: 16000, 6: 22050, 7: 24000, 8: 32000, 9: 44100,
                10: 48000, 11: 96000}
SAMPLE_SIZES = {1: 8, 2: 12, 4: 16, 5: 20, 6: 24, 7: 32}


def crc8(data):
    crc = 0
    for b in data:
        crc ^= b
        for _ in range(8):
            crc = ((crc << 1) ^ 0x07) & 0xFF if crc & 0x80 else (crc << 1) & 0xFF
    return crc


def crc16(data):
    crc = 0
    for b in data:
        crc ^= b << 8
        for _ in range(8):
            crc = ((crc << 1) ^ 0x8005) & 0xFFFF if crc & 0x8000 else (crc << 1) & 0xFFFF
    return crc


class Bits:
    def __init__(self, data, pos=0):
        self.bits = np.unpackbits(np.frombuffer(bytes(data), np.uint8))
        self.ones = np.flatnonzero(self.bits)
        self.pos = pos

    def u(self, n):
        v = 0
        for b in self.bits[self.pos:self.pos + n]:
            v = (v << 1) | int(b)
        assert self.pos + n <= len(self.bits), "ran past the end of the frame"
        self.pos += n
        return v

    def s(self, n):
        v = self.u(n)
        return v - (1 << n) if n and v >> (n - 1) else v

    def unary(self):
        i = int(np.searchsorted(self.ones, self.pos))
        assert i < len(self.ones), "unterminated unary code"
        q = int(self.ones[i]) - self.pos
        self.pos += q + 1
        return q


def _residual(br, n, order):
    method = br.u(2)
    assert method in (0, 1)
    pbits, esc = (4, 15) if method == 0 else (5, 31)
    porder = br.u(4)
    nparts = 1 << porder
    assert n % nparts == 0
    out = np.zeros(n, np.int64)
    t = order
    for q in range(nparts):
        p = br.u(pbits)
        count = n // nparts - (order if q == 0 else 0)
        if p == esc:
            raw = br.u(5)
            for _ in range(count):
                out[t] = br.s(raw)
                t += 1
        else:
            for _ in range(count):
                u = (br.unary() << p) | br.u(p)
                out[t] = (u >> 1) ^ -(u & 1)
                t += 1
    return out


def _subframe(br, n, bps):
    assert br.u(1) == 0
    typ = br.u(6)
    assert br.u(1) == 0, "wasted bits are not produced by this encoder"
    if typ == 0:
        return np.full(n, br.s(bps), np.int64), "constant"
    if typ == 1:
        return np.array([br.s(bps) for _ in range(n)], np.int64), "verbatim"
    if 8 <= typ <= 12:
        order = typ - 8
        coefs, shift, kind = FIXED_COEFS[order], 0, "fixed"
        warm = [br.s(bps) for _ in range(order)]
    else:
        assert typ >= 32
        order = typ - 31
        warm = [br.s(bps) for _ in range(order)]
        prec = br.u(4) + 1
        assert prec != 16
        shift = br.s(5)
        assert shift >= 0
        coefs, kind = [br.s(prec) for _ in range(order)], "lpc"
    out = _residual(br, n, order)
    out[:order] = warm
    for t in range(order, n):
        pred = sum(int(c) * int(out[t - 1 - j]) for j, c in enumerate(coefs))
        out[t] += pred >> shift
    return out, kind


def parse_frame(data, stream_bps=None, stream_rate=None):
    """-> dict(header fields, channels = int64 [nch, n] after undoing the stereo decorrelation,
    kinds, length = bytes consumed).  Asserts the sync code, reserved bits and both CRCs."""
    data = bytes(data)
    br = Bits(data)
    assert br.u(14) == 0x3FFE, "sync code"
    assert br.u(1) == 0
    variable = br.u(1)
    bs_tag, sr_tag, ch_tag, ss_tag = br.u(4), br.u(4), br.u(4), br.u(3)
    assert br.u(1) == 0
    first = br.u(8)
    if first < 0x80:
        number = first
    else:
        nbytes = len(bin(first)[2:].split("0")[0])   # number of leading one bits
        number = first & ((1 << (7 - nbytes)) - 1)
        for _ in range(nbytes - 1):
            b = br.u(8)
            assert b >> 6 == 2
            number = (number << 6) | (b & 0x3F)
    if bs_tag == 6:
        n = br.u(8) + 1
    elif bs_tag == 7:
        n = br.u(16) + 1
    else:
        n = BLOCK_SIZES[bs_tag]
    if sr_tag == 12:
        rate = br.u(8) * 1000
    elif sr_tag == 13:
        rate = br.u(16)
    elif sr_tag == 14:
        rate = br.u(16) * 10
    else:
        rate = SAMPLE_RATES.get(sr_tag, stream_rate)
    hdr_len = br.pos // 8
    assert br.u(8) == crc8(data[:hdr_len]), "header CRC-8"
    bps = SAMPLE_SIZES.get(ss_tag, stream_bps)
    assert bps is not None
    nch = ch_tag + 1 if ch_tag < 8 else 2
    assert ch_tag <= 10
    subs, kinds = [], []
    for c in range(nch):
        side = (ch_tag == 8 and c == 1) or (ch_tag == 9 and c == 0) or (ch_tag == 10 and c == 1)
        x, kind = _subframe(br, n, bps + (1 if side else 0))
        subs.append(x)
        kinds.append(kind)
    br.pos = (br.pos + 7) // 8 * 8
    body_len = br.pos // 8
    assert br.u(16) == crc16(data[:body_len]), "frame CRC-16"
    if ch_tag == 8:
        subs[1] = subs[0] - subs[1]
    elif ch_tag == 9:
        subs[0] = subs[0] + subs[1]
    elif ch_tag == 10:
        mid, side = subs
        mid = (mid << 1) | (side & 1)
        subs = [(mid + side) >> 1, (mid - side) >> 1]
    return {"variable": variable, "number": number, "block_size": n, "sample_rate": rate, "bps": bps,
            "channel_tag": ch_tag, "channels": np.stack(subs), "kinds": kinds, "length": br.pos // 8}
