"""Reader for the GDX files PIPS-IPM++ takes its block-structured LPs from.

The reference reads one GDX file per block through the GAMS GDX library (`gmspips_reader`, Drivers/gams/gmspips/
gmspips_reader.cpp:30-60 -> `readBlock`, gmspipsio.c:1357-2033).  The files are "jacobian" dumps of GAMS' CONVERTD: sets
`i` (rows) and `j` (columns), `jobj`/`iobj`/`objcoef`, the equation records `e` and variable records `x` (level, marginal,
lower, upper, scale - the scale field carries the stage annotation) and the 2-dimensional parameter `A` (the Jacobian).

This module restates the two layers that are needed, nothing else of the GDX library:
  * `GdxFile`       - container format, version 7, uncompressed: header with the section offsets, symbol table, per-symbol
                      record stream (first-changed-dimension byte, delta-coded keys, special-value coded doubles);
  * `read_block`    - the block extraction rules of `readBlock`: which variables / rows belong to block k, the row classes
                      (A | C | BL | DL), bounds with 0/1 indicators, objective row elimination.
`write_gdx` produces files in the same container format; the tests use it for round trips.
"""
import struct

import numpy as np

MARK_BOI = 19510624
# special-value codes of the record stream: undefined, NA, +inf, -inf, EPS, 0, 1, -1, 1/2, 2, "a double follows"
SV_UNDEF, SV_NA, SV_PINF, SV_MINF = 1.0e300, 2.0e300, 3.0e300, 4.0e300
_CODED = {0: SV_UNDEF, 1: SV_NA, 2: SV_PINF, 3: SV_MINF, 4: 0.0, 5: 0.0, 6: 1.0, 7: -1.0, 8: 0.5, 9: 2.0}   # EPS reads as 0
SET, PARAMETER, VARIABLE, EQUATION = 0, 1, 2, 3
_NVALS = {SET: 1, PARAMETER: 1, VARIABLE: 5, EQUATION: 5}
LEVEL, MARGINAL, LOWER, UPPER, SCALE = range(5)


class GdxError(Exception):
    pass


class _Cursor:
    def __init__(self, data, pos=0):
        self.d, self.p = data, pos

    def take(self, fmt):
        v = struct.unpack_from("<" + fmt, self.d, self.p)
        self.p += struct.calcsize("<" + fmt)
        return v[0] if len(v) == 1 else v

    def string(self):
        n = self.d[self.p]
        s = self.d[self.p + 1:self.p + 1 + n].decode("latin-1")
        self.p += 1 + n
        return s

    def expect(self, marker):
        got = self.string()
        if got != marker:
            raise GdxError(f"expected section marker {marker!r} at offset {self.p}, found {got!r}")


class GdxSymbol:
    def __init__(self, name, pos, dim, typ, user_info, count, text):
        self.name, self.pos, self.dim, self.type, self.user_info, self.count, self.text = name, pos, dim, typ, user_info, count, text
        self.keys = None     # (count, dim) int32, raw UEL numbers
        self.values = None   # (count, nvals) float64


class GdxFile:
    """Uncompressed GDX version 7 file: symbols with raw integer keys and values; `uels` holds the label table (may be empty:
    the per-block files written by the reference's splitter carry the label count in the scalar `numUel` instead)."""

    def __init__(self, path):
        with open(path, "rb") as f:
            self.data = f.read()
        try:
            self._read_header()
        except (struct.error, IndexError, UnicodeDecodeError, OverflowError) as e:
            raise GdxError(f"damaged or truncated GDX file: {e}") from None

    def _read_header(self):
        c = _Cursor(self.data)
        # stream signature: sizes and byte order probes of word / integer / double
        if c.take("B") != 2 or c.take("H") != 0x1234 or c.take("B") != 4 or c.take("I") != 0x12345678 or c.take("B") != 8:
            raise GdxError("not a little-endian GDX stream")
        c.take("d")
        if c.take("B") != 123 or c.string() != "GAMSGDX":
            raise GdxError("GDX header not found")
        self.version, compressed = c.take("i"), c.take("i")
        if self.version != 7 or compressed != 0:
            raise GdxError(f"only uncompressed version 7 files are supported (version {self.version}, compression {compressed})")
        self.audit, self.producer = c.string(), c.string()
        if c.take("i") != MARK_BOI:
            raise GdxError("index marker not found")
        sym_pos, uel_pos, sett_pos, acro_pos, next_pos, doms_pos = c.take("6q")
        self.symbols, self._order = {}, []
        c = _Cursor(self.data, sym_pos)
        c.expect("_SYMB_")
        for _ in range(c.take("i")):
            name = c.string()
            pos, dim, typ, user_info, count, _nerr, _has_text = c.take("q"), c.take("i"), c.take("B"), c.take("i"), c.take("i"), c.take("i"), c.take("B")
            text = c.string()
            if c.take("B") != 0:
                raise GdxError(f"symbol {name}: compressed records are not supported")
            if c.take("B") != 0:          # domain-controlled: one symbol number per dimension
                c.take(f"{dim}i")
            for _ in range(c.take("i")):   # comment lines
                c.string()
            self.symbols[name.lower()] = GdxSymbol(name, pos, dim, typ, user_info, count, text)
            self._order.append(name)
        c.expect("_SYMB_")
        c = _Cursor(self.data, uel_pos)
        c.expect("_UEL_")
        self.uels = [c.string() for _ in range(c.take("i"))]
        c.expect("_UEL_")

    def names(self):
        return list(self._order)

    def symbol(self, name):
        s = self.symbols.get(name.lower())
        if s is None:
            raise GdxError(f"symbol {name!r} not in file")
        if s.keys is None:
            try:
                self._read_records(s)
            except (struct.error, IndexError, KeyError, OverflowError) as e:
                raise GdxError(f"symbol {s.name}: damaged or truncated records: {e}") from None
        return s

    def has(self, name):
        return name.lower() in self.symbols

    def _read_records(self, s):
        if s.type not in _NVALS:
            raise GdxError(f"symbol {s.name}: type {s.type} not supported")
        c = _Cursor(self.data, s.pos)
        c.expect("_DATA_")
        dim = c.take("B")
        c.take("i")                      # record count as known when the section was started (-1: see the symbol table)
        if dim != s.dim:
            raise GdxError(f"symbol {s.name}: dimension mismatch")
        lo, fmt = [], []
        for _ in range(dim):
            mn, mx = c.take("i"), c.take("i")
            lo.append(mn)
            span = mx - mn
            fmt.append("B" if span <= 255 else ("H" if span <= 65535 else "i"))
        nv = _NVALS[s.type]
        keys, vals = [], []
        key = [0] * dim
        while True:
            b = c.take("B")
            if b == 255:
                break
            if dim:
                if b > dim:                       # only the last index moved, by a small step
                    key[-1] += b - dim
                else:                             # indices b..dim follow
                    for d in range(b - 1, dim):
                        key[d] = lo[d] + c.take(fmt[d])
            rec = []
            for _ in range(nv):
                code = c.take("B")
                rec.append(c.take("d") if code == 10 else _CODED[code])
            keys.append(tuple(key))
            vals.append(rec)
        s.keys = np.array(keys, dtype=np.int32).reshape(len(keys), dim)
        s.values = np.array(vals, dtype=np.float64).reshape(len(vals), nv)
        if s.count >= 0 and s.count != len(keys):
            raise GdxError(f"symbol {s.name}: {len(keys)} records read, symbol table says {s.count}")


# ------------------------------------------------------------------------------------------------------------------------
# writer (same container format; used to build test files)
# ------------------------------------------------------------------------------------------------------------------------
def _pstr(s):
    b = s.encode("latin-1")
    return bytes([len(b)]) + b


def _coded(v):
    for code, val in ((2, SV_PINF), (3, SV_MINF), (0, SV_UNDEF), (1, SV_NA), (5, 0.0), (6, 1.0), (7, -1.0), (8, 0.5), (9, 2.0)):
        if v == val:
            return bytes([code])
    return bytes([10]) + struct.pack("<d", v)


def write_gdx(path, symbols, uels=(), producer="pips_ipmpp_amd"):
    """symbols: list of (name, type, text, keys (n, dim) ints ascending, values (n, nvals))."""
    out = bytearray()
    out += struct.pack("<BHBIBd", 2, 0x1234, 4, 0x12345678, 8, 3.141592653589793)
    out += bytes([123]) + _pstr("GAMSGDX") + struct.pack("<ii", 7, 0) + _pstr("GDX Library (restated writer)") + _pstr(producer)
    out += struct.pack("<i", MARK_BOI)
    index_at = len(out)
    out += bytes(48)
    table = []
    for name, typ, text, keys, vals in symbols:
        keys = np.asarray(keys, dtype=np.int64)
        vals = np.asarray(vals, dtype=np.float64)
        n = len(vals)
        dim = keys.shape[1] if keys.ndim == 2 else 0
        keys = keys.reshape(n, dim)
        vals = vals.reshape(n, _NVALS[typ])
        table.append((name, len(out), dim, typ, n, text))
        out += _pstr("_DATA_") + bytes([dim]) + struct.pack("<i", -1)
        lo = [int(keys[:, d].min()) if n else 0 for d in range(dim)]
        hi = [int(keys[:, d].max()) if n else 0 for d in range(dim)]
        fmt = []
        for d in range(dim):
            out += struct.pack("<ii", lo[d], hi[d])
            span = hi[d] - lo[d]
            fmt.append("B" if span <= 255 else ("H" if span <= 65535 else "i"))
        prev = None
        for r in range(n):
            k = [int(x) for x in keys[r]]
            if dim == 0:
                out += bytes([1])
            else:
                first = 0 if prev is None else next(d for d in range(dim) if k[d] != prev[d])
                step = k[-1] - prev[-1] if prev is not None else 0
                if prev is not None and first == dim - 1 and 0 < step and dim + step < 255:
                    out += bytes([dim + step])
                else:
                    out += bytes([first + 1])
                    for d in range(first, dim):
                        out += struct.pack("<" + fmt[d], k[d] - lo[d])
            for v in vals[r]:
                out += _coded(float(v))
            prev = k
        out += bytes([255])
    sym_pos = len(out)
    out += _pstr("_SYMB_") + struct.pack("<i", len(table))
    for name, pos, dim, typ, n, text in table:
        out += _pstr(name) + struct.pack("<qiBiiiB", pos, dim, typ, 0, n, 0, 0) + _pstr(text) + bytes([0, 0]) + struct.pack("<i", 0)
    out += _pstr("_SYMB_")
    sett_pos = len(out)
    out += _pstr("_SETT_") + struct.pack("<i", 0) + _pstr("_SETT_")
    uel_pos = len(out)
    out += _pstr("_UEL_") + struct.pack("<i", len(uels))
    for u in uels:
        out += _pstr(u)
    out += _pstr("_UEL_")
    acro_pos = len(out)
    out += _pstr("_ACRO_") + struct.pack("<i", 0) + _pstr("_ACRO_")
    doms_pos = len(out)
    out += _pstr("_DOMS_") + struct.pack("<i", 0) + _pstr("_DOMS_") + struct.pack("<i", -1) + _pstr("_DOMS_")
    out[index_at:index_at + 48] = struct.pack("<6q", sym_pos, uel_pos, sett_pos, acro_pos, sym_pos, doms_pos)
    with open(path, "wb") as f:
        f.write(out)


# ------------------------------------------------------------------------------------------------------------------------
# block extraction (readBlock, gmspipsio.c:1357-2033)
# ------------------------------------------------------------------------------------------------------------------------
def _csr(rows, n_rows, n_cols):
    """rows: list (per row) of [(col, val)] in file order -> dict in the callback convention."""
    rowptr, colidx, val = [0], [], []
    for r in rows:
        for cidx, v in r:
            colidx.append(int(cidx))
            val.append(float(v))
        rowptr.append(len(colidx))
    return dict(rows=n_rows, cols=n_cols, rowptr=rowptr, colidx=colidx, val=val)


def read_block(path, num_blocks, act_block, offset=1):
    """See `_read_block`; anything a damaged file can trigger surfaces as GdxError."""
    try:
        return _read_block(path, num_blocks, act_block, offset)
    except (KeyError, IndexError, ValueError, OverflowError, struct.error) as e:
        raise GdxError(f"{path}: inconsistent jacobian data ({type(e).__name__}: {e})") from None


def _read_block(path, num_blocks, act_block, offset=1):
    """Block `act_block` (0 = the linking variables and their rows) of a `num_blocks`-block problem from one jacobian GDX file.

    Follows `readBlock` (gmspipsio.c:1357-2033): a variable belongs to block `stage - offset` (stage = the scale field of
    `x`), an equation likewise, equations of stage `num_blocks + offset` are linking rows; rows with both bounds finite are
    equalities (A/B, linking: BL), rows with one infinite bound inequalities (C/D, linking: DL), free rows and the objective
    row are dropped; the objective row gives c = -direction * coefficient / (coefficient of the objective variable).
    Returns a dict with the reference's block fields (GMSPIPSBlockData_t, gmspipsio.h:5-58): n0, ni, c, xlow/ixlow/xupp/ixupp,
    b, clow/iclow/cupp/icupp, bL, dlow/idlow/dupp/idupp and the CSR matrices A, B, C, D, BL, DL (None where the reference
    leaves the pointer NULL)."""
    g = GdxFile(path)
    direction = int(g.symbol("objcoef").values[0, LEVEL])
    if direction not in (-1, 1):
        raise GdxError("objcoef must be 1 (min) or -1 (max)")
    obj_var = int(g.symbol("jobj").keys[0, 0])
    jset, iset, x, e, A = (g.symbol(n) for n in ("j", "i", "x", "e", "A"))
    col_of = {int(k): n for n, k in enumerate(jset.keys[:, 0])}      # label number -> position in j
    row_of = {int(k): m for m, k in enumerate(iset.keys[:, 0])}
    gdx_n, gdx_m = len(col_of), len(row_of)
    # ---- variables: 0 = not ours, 1.. = position in [block-0 variables | block variables]
    var_perm = np.zeros(gdx_n, dtype=np.int64)
    n0 = ni = 0
    for k, v in zip(x.keys[:, 0], x.values):
        if k == obj_var:
            continue
        blk = int(v[SCALE]) - offset
        if blk == 0:
            var_perm[col_of[int(k)]] = 1
            n0 += 1
        elif blk == act_block:
            var_perm[col_of[int(k)]] = 2
            ni += 1
    c0 = ci = 0
    for jn in range(gdx_n):
        if var_perm[jn] == 1:
            c0 += 1
            var_perm[jn] = c0
        elif var_perm[jn] == 2:
            ci += 1
            var_perm[jn] = n0 + ci
    if act_block == 0:
        ni = n0
    out = dict(numBlocks=num_blocks, blockID=act_block, n0=n0, ni=ni)
    xlow, xupp, ixlow, ixupp = np.zeros(ni), np.zeros(ni), np.zeros(ni, dtype=np.int16), np.zeros(ni, dtype=np.int16)
    n = 0
    for k, v in zip(x.keys[:, 0], x.values):
        if k == obj_var or int(v[SCALE]) - offset != act_block:
            continue
        if v[LOWER] != SV_MINF:
            xlow[n], ixlow[n] = v[LOWER], 1
        if v[UPPER] != SV_PINF:
            xupp[n], ixupp[n] = v[UPPER], 1
        n += 1
    # ---- objective row: the one row the objective variable appears in
    obj_row, obj_coef, obj_terms = 0, 0.0, []
    cur = []
    for (ri, cj), v in zip(A.keys, A.values[:, LEVEL]):
        if cj == obj_var:
            if obj_row:
                raise GdxError("objective variable used in more than one row")
            obj_row, obj_coef = int(ri), float(v)
    if obj_row:
        cur = [(int(cj), float(v)) for (ri, cj), v in zip(A.keys, A.values[:, LEVEL]) if ri == obj_row and cj != obj_var]
        obj_terms = cur
    c = np.zeros(ni)
    for cj, v in obj_terms:
        p = var_perm[col_of[cj]]
        if p == 0 or (p <= n0 and act_block > 0):
            continue
        c[p - (1 if p <= n0 else n0 + 1)] = direction * (-v / obj_coef)
    # ---- equations: 1 = A (own equality), 2 = C (own inequality), 3 = BL, 4 = DL
    etype = np.zeros(gdx_m, dtype=np.int64)
    b, clow, cupp, iclow, icupp, bL, dlow, dupp, idlow, idupp = ([] for _ in range(10))
    for k, v in zip(e.keys[:, 0], e.values):
        lo_inf, up_inf = v[LOWER] == SV_MINF, v[UPPER] == SV_PINF
        if (lo_inf and up_inf) or k == obj_row:
            continue
        blk = int(v[SCALE]) - offset
        if blk != act_block and blk != num_blocks:
            continue
        link = blk == num_blocks and blk != act_block
        ineq = lo_inf or up_inf
        etype[row_of[int(k)]] = (4 if ineq else 3) if link else (2 if ineq else 1)
        lows, upps, ilows, iupps, rhs = (dlow, dupp, idlow, idupp, bL) if link else (clow, cupp, iclow, icupp, b)
        if lo_inf:
            lows.append(0.0); ilows.append(0); upps.append(float(v[UPPER])); iupps.append(1)
        elif up_inf:
            lows.append(float(v[LOWER])); ilows.append(1); upps.append(0.0); iupps.append(0)
        else:
            rhs.append(float(v[LOWER]))
    m_of = {t: int((etype == t).sum()) for t in (1, 2, 3, 4)}
    # position of every row inside its class, in the order of the set i
    pos_in_class, cnt = np.zeros(gdx_m, dtype=np.int64), {1: 0, 2: 0, 3: 0, 4: 0}
    for m in range(gdx_m):
        if etype[m]:
            pos_in_class[m] = cnt[etype[m]]
            cnt[etype[m]] += 1
    mats = {name: [[] for _ in range(m_of[t])] for name, t in (("A", 1), ("B", 1), ("C", 2), ("D", 2), ("BL", 3), ("DL", 4))}
    for (ri, cj), v in zip(A.keys, A.values[:, LEVEL]):
        if ri == obj_row:
            continue
        row, col = row_of[int(ri)], col_of[int(cj)]
        t, p = etype[row], var_perm[col]
        if t == 0 and (p == 0 or p <= n0):
            continue
        if t > 2 and (p == 0 or (p <= n0 and act_block != 0)):
            continue
        if t == 0 or p == 0:
            raise GdxError(f"unexpected matrix coefficient: row label {ri} and column label {cj} belong to different blocks")
        r = pos_in_class[row]
        if p <= n0:
            name = {1: "A", 2: "C", 3: "BL", 4: "DL"}[t]
            mats[name][r].append((p - 1, v))
        else:
            name = {1: "B", 2: "D", 3: "BL", 4: "DL"}[t]
            mats[name][r].append((p - n0 - 1, v))
    nloc = n0 if act_block == 0 else ni
    out.update(c=c, xlow=xlow, xupp=xupp, ixlow=ixlow, ixupp=ixupp, b=np.array(b), clow=np.array(clow), cupp=np.array(cupp),
               iclow=np.array(iclow, dtype=np.int16), icupp=np.array(icupp, dtype=np.int16), bL=np.array(bL), dlow=np.array(dlow),
               dupp=np.array(dupp), idlow=np.array(idlow, dtype=np.int16), idupp=np.array(idupp, dtype=np.int16),
               mA=m_of[1], mC=m_of[2], mBL=m_of[3], mDL=m_of[4])
    out["A"] = _csr(mats["A"], m_of[1], n0) if m_of[1] else None
    out["B"] = _csr(mats["B"], m_of[1], ni) if m_of[1] and act_block != 0 else None
    out["C"] = _csr(mats["C"], m_of[2], n0) if m_of[2] else None
    out["D"] = _csr(mats["D"], m_of[2], ni) if m_of[2] and act_block != 0 else None
    out["BL"] = _csr(mats["BL"], m_of[3], nloc) if m_of[3] else None
    out["DL"] = _csr(mats["DL"], m_of[4], nloc) if m_of[4] else None
    return out


def read_problem(stem, num_blocks, offset=1):
    """All blocks of an instance the way `gmspips_reader` opens them: `<stem><k>.gdx`, k = 0..num_blocks-1."""
    return [read_block(f"{stem}{k}.gdx", num_blocks, k, offset) for k in range(num_blocks)]
