"""A small JPEG *encoder* for the tests of app/jpeg_reader.cpp: baseline and progressive Huffman streams with
arbitrary sampling factors, restart intervals, custom Huffman tables, JFIF / Adobe markers.  It only has to
produce valid ITU T.81 streams; what a decoder must make of them is decided by the reference's own decoder
(oracle/_ref), not by this file."""
import struct

import numpy as np
from scipy.fft import dctn

ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35,
          42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


class Bits:
    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, value, nbits):
        if nbits == 0:
            return
        self.acc = (self.acc << nbits) | (value & ((1 << nbits) - 1))
        self.n += nbits
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 255
            self.out.append(b)
            if b == 255:
                self.out.append(0)
            self.n -= 8

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)  # pad with ones


def canonical(lengths):
    """symbol -> length  =>  (counts[16], symbols in code order, symbol -> (code, length))"""
    order = sorted(lengths, key=lambda s: (lengths[s], s))
    counts = [0] * 16
    codes, code, prev = {}, 0, 0
    for s in order:
        ln = lengths[s]
        code <<= ln - prev
        prev = ln
        codes[s] = (code, ln)
        code += 1
        counts[ln - 1] += 1
    assert all(c < (1 << ln) - 1 or ln == 0 for c, ln in codes.values()), "all-ones code"
    return counts, order, codes


def table_for(symbols, style):
    """a valid prefix code over `symbols` (style 0: one length; style 1: a few short codes, the rest long)"""
    symbols = sorted(set(symbols))
    n = len(symbols)
    if style == 0 or n < 6:
        ln = max(2, int(np.ceil(np.log2(n + 1))))
        return {s: ln for s in symbols}
    lengths = {}
    for i, s in enumerate(symbols):
        lengths[s] = 3 if i < 3 else (5 if i < 7 else (12 if n - 7 < 120 else 13))
    kraft = sum(2.0 ** -v for v in lengths.values())
    assert kraft < 1.0
    return lengths


def magnitude(v):
    a = abs(int(v))
    s = a.bit_length()
    return s, (v if v >= 0 else v + (1 << s) - 1)


class Encoder:
    def __init__(self, planes, samp, qtabs, tq, progressive=False, restart=0, style=0, markers=b"", ids=None, script=None):
        """planes: list of 2-D uint8 arrays at FULL resolution (one per component); samp: [(h, v)] per component"""
        self.H, self.W = planes[0].shape
        self.samp, self.tq, self.qtabs = samp, tq, qtabs
        self.progressive, self.restart, self.style, self.markers = progressive, restart, style, markers
        self.ids = ids or list(range(1, len(planes) + 1))
        self.script = script
        hmax, vmax = max(h for h, _ in samp), max(v for _, v in samp)
        self.hmax, self.vmax = hmax, vmax
        self.mcu_x = -(-self.W // (8 * hmax))
        self.mcu_y = -(-self.H // (8 * vmax))
        self.coefs = []
        for p, (h, v) in zip(planes, samp):
            fx, fy = hmax // h, vmax // v
            cw, chh = -(-self.W * h // hmax), -(-self.H * v // vmax)
            pad = np.pad(p.astype(np.float64), ((0, chh * fy - self.H), (0, cw * fx - self.W)), mode="edge")
            sub = pad.reshape(chh, fy, cw, fx).mean(axis=(1, 3))
            w2, h2 = self.mcu_x * h * 8, self.mcu_y * v * 8
            sub = np.pad(sub, ((0, h2 - chh), (0, w2 - cw)), mode="edge") - 128.0
            blocks = sub.reshape(h2 // 8, 8, w2 // 8, 8).transpose(0, 2, 1, 3)
            d = dctn(blocks, axes=(2, 3), norm="ortho")
            q = np.asarray(qtabs[tq[len(self.coefs)]], np.float64).reshape(8, 8)
            self.coefs.append(np.rint(d / q).astype(np.int32).reshape(h2 // 8, w2 // 8, 64)[:, :, ZIGZAG])  # zigzag order
            self.dims = getattr(self, "dims", []) + [(cw, chh)]

    # ---- marker segments
    def seg(self, m, payload):
        return b"\xff" + bytes([m]) + struct.pack(">H", len(payload) + 2) + payload

    def header(self):
        out = b"\xff\xd8" + self.markers
        for t, q in enumerate(self.qtabs):
            zz = [int(np.asarray(q).reshape(64)[ZIGZAG[i]]) for i in range(64)]
            if max(zz) > 255:
                out += self.seg(0xDB, bytes([0x10 | t]) + b"".join(struct.pack(">H", v) for v in zz))
            else:
                out += self.seg(0xDB, bytes([t]) + bytes(zz))
        sof = struct.pack(">BHHB", 8, self.H, self.W, len(self.samp))
        for i, (h, v) in enumerate(self.samp):
            sof += bytes([self.ids[i], (h << 4) | v, self.tq[i]])
        out += self.seg(0xC2 if self.progressive else 0xC0, sof)
        if self.restart:
            out += self.seg(0xDD, struct.pack(">H", self.restart))
        return out

    def dht(self, cls, idx, lengths):
        counts, order, codes = canonical(lengths)
        return self.seg(0xC4, bytes([(cls << 4) | idx]) + bytes(counts) + bytes(order)), codes

    def sos(self, comps, ss, se, ah, al, td_ta):
        p = bytes([len(comps)])
        for c in comps:
            p += bytes([self.ids[c], td_ta[c]])
        return self.seg(0xDA, p + bytes([ss, se, (ah << 4) | al]))

    # ---- iteration orders
    def units(self, comps):
        """restart units: lists of (component, by, bx)"""
        if len(comps) == 1:
            c = comps[0]
            cw, chh = self.dims[c]
            for by in range(-(-chh // 8)):
                for bx in range(-(-cw // 8)):
                    yield [(c, by, bx)]
        else:
            for my in range(self.mcu_y):
                for mx in range(self.mcu_x):
                    u = []
                    for c in comps:
                        h, v = self.samp[c]
                        for y in range(v):
                            for x in range(h):
                                u.append((c, my * v + y, mx * h + x))
                    yield u

    # ---- sequential
    def encode_baseline(self, interleaved=True):
        out = self.header()
        groups = [list(range(len(self.samp)))] if interleaved else [[c] for c in range(len(self.samp))]
        for comps in groups:
            # symbols first (table per scan), then bits
            dc_syms, ac_syms = set(), set()
            events = []
            pred = {c: 0 for c in comps}
            n_units = 0
            for unit in self.units(comps):
                if self.restart and n_units and n_units % self.restart == 0:
                    events.append(("rst",))
                    pred = {c: 0 for c in comps}
                n_units += 1
                for c, by, bx in unit:
                    z = self.coefs[c][by, bx]
                    s, bits = magnitude(int(z[0]) - pred[c])
                    pred[c] = int(z[0])
                    events.append(("dc", s, bits))
                    dc_syms.add(s)
                    run = 0
                    last = max([k for k in range(1, 64) if z[k] != 0], default=0)
                    for k in range(1, last + 1):
                        if z[k] == 0:
                            run += 1
                            continue
                        while run > 15:
                            events.append(("ac", 0xF0, 0, 0)); ac_syms.add(0xF0); run -= 16
                        s, bits = magnitude(int(z[k]))
                        events.append(("ac", (run << 4) | s, s, bits)); ac_syms.add((run << 4) | s)
                        run = 0
                    if last < 63:
                        events.append(("ac", 0x00, 0, 0)); ac_syms.add(0x00)
            seg_dc, dcc = self.dht(0, 0, table_for(dc_syms, self.style))
            seg_ac, acc = self.dht(1, 0, table_for(ac_syms, self.style))
            out += seg_dc + seg_ac + self.sos(comps, 0, 63, 0, 0, {c: 0x00 for c in comps})
            out += self.emit(events, dcc, acc)
        return out + b"\xff\xd9"

    def emit(self, events, dcc, acc):
        out, b, rst = b"", Bits(), 0
        for e in events:
            if e[0] == "rst":
                b.flush(); out += bytes(b.out) + bytes([0xFF, 0xD0 + (rst & 7)]); rst += 1; b = Bits()
            elif e[0] == "dc":
                code, ln = dcc[e[1]]; b.put(code, ln); b.put(e[2], e[1])
            elif e[0] == "ac":
                code, ln = acc[e[1]]; b.put(code, ln); b.put(e[3], e[2])
            else:  # raw bits
                b.put(e[1], e[2])
        b.flush()
        return out + bytes(b.out)

    # ---- progressive: script = list of (components, ss, se, ah, al)
    def encode_progressive(self):
        out = self.header()
        n = len(self.samp)
        script = self.script or ([(list(range(n)), 0, 0, 0, 1)] + [([c], 1, 5, 0, 2) for c in range(n)] +
                                 [([c], 6, 63, 0, 2) for c in range(n)] + [([c], 1, 63, 2, 1) for c in range(n)] +
                                 [(list(range(n)), 0, 0, 1, 0)] + [([c], 1, 63, 1, 0) for c in range(n)])
        for comps, ss, se, ah, al in script:
            events, syms = [], set()
            pred = {c: 0 for c in comps}
            state = {"eobrun": 0, "be": []}

            def flush_eobrun():
                if state["eobrun"]:
                    nb = state["eobrun"].bit_length() - 1
                    events.append(("ac", nb << 4, nb, state["eobrun"] & ((1 << nb) - 1))); syms.add(nb << 4)
                    state["eobrun"] = 0
                    for bit in state["be"]:
                        events.append(("raw", bit, 1))
                    state["be"] = []

            n_units = 0
            for unit in self.units(comps):
                if self.restart and n_units and n_units % self.restart == 0:
                    flush_eobrun()
                    events.append(("rst",))
                    pred = {c: 0 for c in comps}
                n_units += 1
                for c, by, bx in unit:
                    z = self.coefs[c][by, bx]
                    if ss == 0:
                        if ah == 0:
                            v = int(z[0]) >> al
                            s, bits = magnitude(v - pred[c]); pred[c] = v
                            events.append(("dc", s, bits)); syms.add(s)
                        else:
                            events.append(("raw", (int(z[0]) >> al) & 1, 1))
                        continue
                    absv = [abs(int(z[k])) >> al for k in range(64)]
                    if ah == 0:
                        run = 0
                        for k in range(ss, se + 1):
                            if absv[k] == 0:
                                run += 1
                                continue
                            flush_eobrun()
                            while run > 15:
                                events.append(("ac", 0xF0, 0, 0)); syms.add(0xF0); run -= 16
                            s = absv[k].bit_length()
                            bits = absv[k] if z[k] >= 0 else (absv[k] ^ ((1 << s) - 1))
                            events.append(("ac", (run << 4) | s, s, bits)); syms.add((run << 4) | s)
                            run = 0
                        if run > 0:
                            state["eobrun"] += 1
                            if state["eobrun"] == 0x7FFF:
                                flush_eobrun()
                    else:
                        eob = max([k for k in range(ss, se + 1) if absv[k] == 1], default=-1)
                        run, br = 0, []
                        for k in range(ss, se + 1):
                            t = absv[k]
                            if t == 0:
                                run += 1
                                continue
                            while run > 15 and k <= eob:
                                flush_eobrun()
                                events.append(("ac", 0xF0, 0, 0)); syms.add(0xF0); run -= 16
                                for bit in br:
                                    events.append(("raw", bit, 1))
                                br = []
                            if t > 1:
                                br.append(t & 1)
                                continue
                            flush_eobrun()
                            events.append(("ac", (run << 4) | 1, 1, 1 if z[k] >= 0 else 0)); syms.add((run << 4) | 1)
                            for bit in br:
                                events.append(("raw", bit, 1))
                            br, run = [], 0
                        if run > 0 or br:
                            state["eobrun"] += 1
                            state["be"] += br
                            if state["eobrun"] == 0x7FFF or len(state["be"]) > 900:
                                flush_eobrun()
            flush_eobrun()
            if ss == 0 and ah == 0:
                seg, codes = self.dht(0, 0, table_for(syms, self.style))
                out += seg + self.sos(comps, ss, se, ah, al, {c: 0x00 for c in comps}) + self.emit(events, codes, {})
            elif ss == 0:
                out += self.sos(comps, ss, se, ah, al, {c: 0x00 for c in comps}) + self.emit(events, {}, {})
            else:
                seg, codes = self.dht(1, 0, table_for(syms | {0}, self.style))
                out += seg + self.sos(comps, ss, se, ah, al, {c: 0x00 for c in comps}) + self.emit(events, {}, codes)
        return out + b"\xff\xd9"


def quant_table(scale):
    base = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51,
                     87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101,
                     72, 92, 95, 98, 112, 100, 103, 99])
    return np.clip(np.rint(base * scale), 1, 65535).astype(int)
