"""Minimal OLE2 / BIFF8 (.xls) cell reader — harness-only helper.

`pandas.read_excel` needs `xlrd`, which this image lacks. The golden-vector generator only
needs the text / number cells of one sheet of `data/shp_jasenkunnat_2020.xls` (healthcare
district -> municipalities), so this reads exactly that: OLE2 header -> FAT chain -> 'Workbook'
stream -> BOUNDSHEET / SST / LABELSST / NUMBER / RK / MULRK records.
"""
import struct

FREESECT = 0xFFFFFFFF
ENDOFCHAIN = 0xFFFFFFFE


def _read_ole_stream(data, name):
    assert data[:8] == b'\xd0\xcf\x11\xe0\xa1\xb1\x1a\xe1', 'not an OLE2 file'
    sec_shift, mini_shift = struct.unpack_from('<HH', data, 30)
    sec_size = 1 << sec_shift
    n_fat, dir_start, _, mini_cutoff, minifat_start, n_minifat, difat_start, n_difat = \
        struct.unpack_from('<IIIIIIII', data, 44)

    def sector(n):
        off = 512 + n * sec_size
        return data[off:off + sec_size]

    difat = list(struct.unpack_from('<109I', data, 76))
    nxt = difat_start
    for _ in range(n_difat):
        sec = sector(nxt)
        vals = struct.unpack('<%dI' % (sec_size // 4), sec)
        difat.extend(vals[:-1])
        nxt = vals[-1]
    fat = []
    for s in difat:
        if s == FREESECT or s == ENDOFCHAIN:
            continue
        fat.extend(struct.unpack('<%dI' % (sec_size // 4), sector(s)))

    def chain(start):
        out = bytearray()
        n = start
        seen = 0
        while n != ENDOFCHAIN and n != FREESECT:
            out += sector(n)
            n = fat[n]
            seen += 1
            assert seen < 1 << 24
        return bytes(out)

    directory = chain(dir_start)
    entries = []
    for off in range(0, len(directory), 128):
        e = directory[off:off + 128]
        nlen = struct.unpack_from('<H', e, 64)[0]
        if nlen < 2:
            continue
        ename = e[:nlen - 2].decode('utf-16-le')
        etype = e[66]
        start, size = struct.unpack_from('<II', e, 116)
        entries.append((ename, etype, start, size))
    root = [e for e in entries if e[1] == 5][0]
    for ename, etype, start, size in entries:
        if ename == name and etype == 2:
            if size < mini_cutoff:
                # stream lives in the mini stream
                mini_stream = chain(root[2])
                minifat = struct.unpack('<%dI' % (len(chain(minifat_start)) // 4), chain(minifat_start))
                msz = 1 << mini_shift
                out = bytearray()
                n = start
                while n != ENDOFCHAIN:
                    out += mini_stream[n * msz:(n + 1) * msz]
                    n = minifat[n]
                return bytes(out[:size])
            return chain(start)[:size]
    raise KeyError(name)


def _records(stream, pos=0):
    while pos + 4 <= len(stream):
        rtype, rlen = struct.unpack_from('<HH', stream, pos)
        yield pos, rtype, stream[pos + 4:pos + 4 + rlen]
        pos += 4 + rlen


def _rk(v):
    mult100 = v & 1
    is_int = v & 2
    if is_int:
        val = struct.unpack('<i', struct.pack('<I', v))[0] >> 2
        val = float(val)
    else:
        val = struct.unpack('<d', b'\x00\x00\x00\x00' + struct.pack('<I', v & 0xFFFFFFFC))[0]
    return val / 100 if mult100 else val


class _SSTReader:
    """Shared-string table spanning SST + CONTINUE records (strings may split across records,
    with a fresh compression flag byte at each continuation)."""

    def __init__(self, chunks):
        self.chunks = chunks
        self.ci = 0
        self.pos = 0

    def _need(self):
        while self.pos >= len(self.chunks[self.ci]):
            self.ci += 1
            self.pos = 0

    def read(self, n):
        self._need()
        out = self.chunks[self.ci][self.pos:self.pos + n]
        assert len(out) == n
        self.pos += n
        return out

    def read_chars(self, nchars, wide):
        out = ''
        while nchars:
            if self.pos >= len(self.chunks[self.ci]):
                self.ci += 1
                self.pos = 0
                wide = self.chunks[self.ci][0] & 1
                self.pos = 1
            avail = len(self.chunks[self.ci]) - self.pos
            per = 2 if wide else 1
            take = min(nchars, avail // per)
            raw = self.chunks[self.ci][self.pos:self.pos + take * per]
            out += raw.decode('utf-16-le') if wide else raw.decode('latin-1')
            self.pos += take * per
            nchars -= take
        return out

    def skip(self, n):
        while n:
            if self.pos >= len(self.chunks[self.ci]):
                self.ci += 1
                self.pos = 0
            take = min(n, len(self.chunks[self.ci]) - self.pos)
            self.pos += take
            n -= take

    def string(self):
        nchars, = struct.unpack('<H', self.read(2))
        flags = self.read(1)[0]
        rt = 0
        ext = 0
        if flags & 8:
            rt, = struct.unpack('<H', self.read(2))
        if flags & 4:
            ext, = struct.unpack('<I', self.read(4))
        s = self.read_chars(nchars, flags & 1)
        self.skip(4 * rt + ext)
        return s


def read_sheet_cells(path, sheet_name):
    """Return {(row, col): value} for text and numeric cells of one worksheet."""
    with open(path, 'rb') as f:
        data = f.read()
    wb = _read_ole_stream(data, 'Workbook')

    sheets = {}
    sst = []
    recs = list(_records(wb))
    i = 0
    while i < len(recs):
        pos, rtype, body = recs[i]
        if rtype == 0x0085:  # BOUNDSHEET
            off, = struct.unpack_from('<I', body, 0)
            nlen = body[6]
            wide = body[7] & 1
            nm = body[8:8 + nlen * (2 if wide else 1)]
            sheets[nm.decode('utf-16-le') if wide else nm.decode('latin-1')] = off
        elif rtype == 0x00FC:  # SST
            chunks = [body[8:]]
            total, unique = struct.unpack_from('<II', body, 0)
            j = i + 1
            while j < len(recs) and recs[j][1] == 0x003C:
                chunks.append(recs[j][2])
                j += 1
            rd = _SSTReader(chunks)
            for _ in range(unique):
                sst.append(rd.string())
            i = j - 1
        elif rtype == 0x000A:  # EOF of globals
            break
        i += 1

    cells = {}
    start = sheets[sheet_name]
    for pos, rtype, body in _records(wb, start):
        if rtype == 0x000A:
            break
        if rtype == 0x00FD:  # LABELSST
            r, c, _, idx = struct.unpack_from('<HHHI', body, 0)
            cells[(r, c)] = sst[idx]
        elif rtype == 0x0203:  # NUMBER
            r, c, _, v = struct.unpack_from('<HHHd', body, 0)
            cells[(r, c)] = v
        elif rtype == 0x027E:  # RK
            r, c, _, v = struct.unpack_from('<HHHI', body, 0)
            cells[(r, c)] = _rk(v)
        elif rtype == 0x00BD:  # MULRK
            r, c0 = struct.unpack_from('<HH', body, 0)
            n = (len(body) - 6) // 6
            for k in range(n):
                v, = struct.unpack_from('<I', body, 4 + k * 6 + 2)
                cells[(r, c0 + k)] = _rk(v)
        elif rtype == 0x0204:  # LABEL (BIFF8 inline string)
            r, c, _, n = struct.unpack_from('<HHHH', body, 0)
            wide = body[8] & 1
            raw = body[9:9 + n * (2 if wide else 1)]
            cells[(r, c)] = raw.decode('utf-16-le') if wide else raw.decode('latin-1')
    return cells
