"""Codestream rewriter for tests: takes a codestream whose packets carry SOP and EPH markers (so that packets
and their headers can be found without decoding them) and re-arranges it into the container variants the
vector factory does not produce itself: several tile-parts per tile, TLM / PLT, packed packet headers in
PPM or PPT segments, COC / QCC / RGN / POC segments in the main or the tile-part headers.

The image content never changes: every variant must decode to the same frame as the stream it was made from
(T.800 Annex A); the plans of the host parser and of the oracle's parser must agree on each.  Test tooling only."""
import struct

SOC, SIZ, COD, COC, TLM, PLT, QCD, QCC, RGN, POC, PPM, PPT, SOT, SOP, EPH, SOD, EOC = (
    0xFF4F, 0xFF51, 0xFF52, 0xFF53, 0xFF55, 0xFF58, 0xFF5C, 0xFF5D, 0xFF5E, 0xFF5F, 0xFF60, 0xFF61, 0xFF90, 0xFF91,
    0xFF92, 0xFF93, 0xFFD9)


def seg(code, payload):
    return struct.pack(">HH", code, len(payload) + 2) + bytes(payload)


class Stream:
    """main: list of (code, payload); tiles: {isot: {"hdr": [(code, payload)], "packets": [(sop, header, body)]}}"""

    def __init__(self, cs):
        cs = bytes(cs)
        assert cs[:2] == b"\xff\x4f"
        pos, self.main, self.tiles, self.order = 2, [], {}, []
        while True:
            code, = struct.unpack_from(">H", cs, pos)
            if code == SOT or code == EOC:
                break
            ln, = struct.unpack_from(">H", cs, pos + 2)
            self.main.append((code, cs[pos + 4:pos + 2 + ln]))
            pos += 2 + ln
        while True:
            code, = struct.unpack_from(">H", cs, pos)
            if code == EOC:
                break
            assert code == SOT, hex(code)
            isot, psot, tpsot, tnsot = struct.unpack_from(">HIBB", cs, pos + 4)
            end = pos + psot if psot else len(cs) - 2
            p, hdr = pos + 12, []
            while struct.unpack_from(">H", cs, p)[0] != SOD:
                c, ln = struct.unpack_from(">HH", cs, p)
                hdr.append((c, cs[p + 4:p + 2 + ln]))
                p += 2 + ln
            t = self.tiles.setdefault(isot, {"hdr": [], "packets": []})
            if isot not in self.order:
                self.order.append(isot)
            t["hdr"] += [h for h in hdr if h[0] not in (PLT,)]
            t["packets"] += self._packets(cs[p + 2:end])
            pos = end
        siz = dict(self.main)[SIZ]
        self.ncomp, = struct.unpack_from(">H", siz, 34)

    @staticmethod
    def _packets(body):
        out, pos = [], 0
        while pos < len(body):
            assert body[pos:pos + 4] == b"\xff\x91\x00\x04", "packets need SOP markers"
            e = body.index(b"\xff\x92", pos + 6)           # a packet header cannot hold FF 92 (bit stuffing)
            nxt = body.find(b"\xff\x91\x00\x04", e + 2)    # a packet body cannot hold FF 91 (bytes after FF are < 0x90)
            nxt = len(body) if nxt < 0 else nxt
            out.append((body[pos:pos + 6], body[pos + 6:e + 2], body[e + 2:nxt]))
            pos = nxt
        return out

    # ---- writers ----
    def build(self, parts_of=None, tlm=False, plt=False, packed=None, extra_tile_hdr=None, interleave=False):
        """parts_of(tile, npackets) -> list of packet counts per tile-part; packed: None | 'ppm' | 'ppt';
        extra_tile_hdr: {(isot, tpsot): [(code, payload)]}; interleave: tile-parts of different tiles alternate"""
        main = [m for m in self.main if m[0] not in (TLM, PPM)]
        tparts = []                                        # (isot, tpsot, ntp, hdr segments, packets)
        for isot in self.order:
            t = self.tiles[isot]
            counts = parts_of(isot, len(t["packets"])) if parts_of else [len(t["packets"])]
            assert sum(counts) == len(t["packets"])
            at = 0
            for k, n in enumerate(counts):
                hdr = list(t["hdr"]) if k == 0 else []
                hdr += (extra_tile_hdr or {}).get((isot, k), [])
                tparts.append([isot, k, len(counts), hdr, t["packets"][at:at + n]])
                at += n
        if interleave:
            tparts.sort(key=lambda tp: (tp[1], tp[0]))
        blobs, ppm_data = [], b""
        for isot, k, ntp, hdr, packets in tparts:
            hdr = list(hdr)
            heads = b"".join(p[1] for p in packets)
            if packed == "ppt" and k == 0:
                allheads = b"".join(p[1] for tp in tparts if tp[0] == isot for p in tp[4])
                for z, off in enumerate(range(0, max(len(allheads), 1), 60000)):
                    hdr.append((PPT, bytes([z]) + allheads[off:off + 60000]))
            if packed == "ppm":
                ppm_data += struct.pack(">I", len(heads)) + heads
            if plt:
                lens, z = b"", 0
                for p in packets:
                    n = len(p[0]) + len(p[2]) + (0 if packed else len(p[1]))
                    groups = [n & 0x7F]
                    n >>= 7
                    while n:
                        groups.insert(0, 0x80 | (n & 0x7F))
                        n >>= 7
                    if len(lens) + len(groups) > 60000:            # a marker segment holds 65 535 bytes: the list goes on in the next
                        hdr.append((PLT, bytes([z & 255]) + lens))
                        lens, z = b"", z + 1
                    lens += bytes(groups)
                hdr.append((PLT, bytes([z & 255]) + lens))
            body = b"".join(p[0] + (b"" if packed else p[1]) + p[2] for p in packets)
            h = b"".join(seg(c, pl) for c, pl in hdr)
            psot = 12 + len(h) + 2 + len(body)
            blobs.append((isot, struct.pack(">HHHIBB", SOT, 10, isot, psot, k, ntp) + h + struct.pack(">H", SOD) + body))
        if packed == "ppm":
            for z, off in enumerate(range(0, max(len(ppm_data), 1), 60000)):
                main.append((PPM, bytes([z]) + ppm_data[off:off + 60000]))
        if tlm:
            main.append((TLM, b"\x00\x60" + b"".join(struct.pack(">HI", isot, len(b)) for isot, b in blobs)))
        return struct.pack(">H", SOC) + b"".join(seg(c, pl) for c, pl in main) + b"".join(b for _, b in blobs) + struct.pack(">H", EOC)

    # ---- header edits (return new payload lists; the Stream itself is changed in place) ----
    def add_coc_qcc(self, comp, in_tile=None):
        """COC + QCC for `comp` repeating the COD / QCD values: same decode, different header path"""
        cod, qcd = dict(self.main)[COD], dict(self.main)[QCD]
        coc = bytes([comp, cod[0] & 1]) + cod[5:]
        qcc = bytes([comp]) + qcd
        target = self.main if in_tile is None else self.tiles[in_tile]["hdr"]
        target += [(COC, coc), (QCC, qcc)]

    def add_poc_split(self, at_res):
        """two progression volumes (resolutions < at_res, then the rest) in the COD's own order: for a one-layer
        LRCP / RLCP stream the packets stay where they are"""
        cod = dict(self.main)[COD]
        order, layers = cod[1], struct.unpack_from(">H", cod, 2)[0]
        ent = struct.pack(">BBHBBB", 0, 0, layers, at_res, self.ncomp, order) + \
              struct.pack(">BBHBBB", at_res, 0, layers, 33, self.ncomp, order)
        self.main.append((POC, ent))

    def add_rgn(self, comp, shift, in_tile=None):
        target = self.main if in_tile is None else self.tiles[in_tile]["hdr"]
        target.append((RGN, bytes([comp, 0, shift])))


def variants(cs, ht):
    """(name, codestream) pairs made from one SOP+EPH codestream.  HT streams may carry PPT / tile-part COC, QCC
    only when Ccap15 bit 11 (heterogeneous) is set, RGN only with bit 12: the caller encodes with
    cap_extra_bits=0x1800 for those."""
    out = []
    s = Stream(cs)
    three = lambda isot, n: [n // 3, n // 3, n - 2 * (n // 3)] if n >= 3 else [n]
    ones = lambda isot, n: [1] * min(n, 31) + ([n - 31] if n > 31 else [])
    out.append(("same", s.build()))
    out.append(("plt", s.build(plt=True)))
    out.append(("tp3_tlm_plt", s.build(parts_of=three, tlm=True, plt=True)))
    out.append(("tp_each_packet", s.build(parts_of=ones, plt=True)))
    out.append(("tp3_interleaved", s.build(parts_of=three, interleave=True, tlm=True)))
    out.append(("ppm", s.build(packed="ppm")))
    out.append(("ppm_tp3", s.build(parts_of=three, packed="ppm", plt=True)))
    out.append(("ppt", s.build(packed="ppt")))
    out.append(("ppt_tp3", s.build(parts_of=three, packed="ppt")))
    if s.ncomp > 1:
        s2 = Stream(cs); s2.add_coc_qcc(1)
        out.append(("coc_qcc_main", s2.build()))
        s3 = Stream(cs); s3.add_coc_qcc(s.ncomp - 1, in_tile=s3.order[-1])
        out.append(("coc_qcc_tile", s3.build(parts_of=three)))
    cod = dict(s.main)[COD]
    if cod[1] in (0, 1) and struct.unpack_from(">H", cod, 2)[0] == 1 and cod[5] >= 2:
        s4 = Stream(cs); s4.add_poc_split(1)
        out.append(("poc_split", s4.build()))
    s5 = Stream(cs); s5.add_rgn(0, 3)
    out.append(("rgn_main", s5.build()))
    s6 = Stream(cs); s6.add_rgn(s.ncomp - 1, 2, in_tile=s6.order[0])
    out.append(("rgn_tile", s6.build()))
    del ht
    return out
