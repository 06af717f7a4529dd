"""tests/pyhevc.py -- TEST-ONLY second decoder: H.265 Main profile I / P / B pictures in plain Python + numpy, written from the
standard's text independently of oracle/ (C) and of kvazzup_amd/csrc (HIP + host C++).  It exists so that the checker is
not the only reading of the syntax layer: CABAC context selection (residual_coding, split / skip / part_mode / cbf ...),
merge and AMVP candidate derivation, cu_qp_delta and the QpY predictor, intra mode derivation, the transform tree, SAO
syntax, deblocking decisions.  tests/test_python_decoder.py decodes the checker's and the generator's streams with it and
compares reconstructed pictures with oracle/hevc_dec.c bit for bit.

Scope: 8-bit 4:2:0, CTB 16..64, one slice per picture, I, P and B slices (bi-prediction, output in POC order), tile rows and columns, WPP, every CU size and
partitioning, transform trees, several reference pictures (short-term RPS), TMVP, cu_qp_delta, sign data hiding, transform
skip, deblocking with offsets, SAO, scaling lists (default / SPS / PPS), cu_transquant_bypass, explicit weighted prediction.  No PCM, long-term pictures.
Normative tables are typed here per syntax element (initValue: Tables 9-5 .. 9-37); rangeTabLps and the state transition
tables, the transform matrices and the interpolation filters are passed in by the caller (tests take them from the KAT-checked
oracle tables: tests/test_oracle_kat.py), so that this file holds logic rather than 400 more typed constants."""
import numpy as np

# ----------------------------------------------------------------------------------------------- bit reader
class Bits:
    def __init__(self, data):
        self.d = data
        self.pos = 0

    def u(self, n):
        v = 0
        for _ in range(n):
            byte = self.d[self.pos >> 3] if (self.pos >> 3) < len(self.d) else 0
            v = (v << 1) | ((byte >> (7 - (self.pos & 7))) & 1)
            self.pos += 1
        return v

    def ue(self):
        z = 0
        while self.u(1) == 0:
            z += 1
            if z > 32:
                raise ValueError("ue")
        return (1 << z) - 1 + (self.u(z) if z else 0)

    def se(self):
        k = self.ue()
        return (k + 1) >> 1 if k & 1 else -(k >> 1)

    def align(self):
        self.pos = (self.pos + 7) & ~7


def unescape(nal):
    out = bytearray()
    z = 0
    for b in nal:
        if z >= 2 and b == 3:
            z = 0
            continue
        out.append(b)
        z = z + 1 if b == 0 else 0
    return bytes(out)


def split_nals(stream):
    """Annex B byte stream -> list of NAL units (without start codes)"""
    s = bytes(stream)
    idx = []
    i = 0
    while i + 3 <= len(s):
        if s[i] == 0 and s[i + 1] == 0 and s[i + 2] == 1:
            idx.append(i + 3)
            i += 3
        else:
            i += 1
    out = []
    for k, a in enumerate(idx):
        b = idx[k + 1] - 3 if k + 1 < len(idx) else len(s)
        while b > a and s[b - 1] == 0:
            b -= 1
        out.append(s[a:b])
    return out


# ----------------------------------------------------------------------------------------------- parameter sets
def parse_ptl(r, max_sub):
    r.u(8)
    r.u(32)
    r.u(4)
    r.u(43)
    r.u(1)
    r.u(8)
    prof = [0] * max_sub
    lev = [0] * max_sub
    for i in range(max_sub):
        prof[i] = r.u(1)
        lev[i] = r.u(1)
    if max_sub > 0:
        for i in range(max_sub, 8):
            r.u(2)
    for i in range(max_sub):
        if prof[i]:
            r.u(88)
        if lev[i]:
            r.u(8)


RPS_FORMS = {"inter_sps": 0, "inter_slice": 0, "explicit_sps": 0, "explicit_slice": 0, "named": 0}      # (for tests: how the streams wrote their sets)


def parse_st_rps(r, idx, num, sets):
    """7.3.7 / 7.4.8: returns list of (delta_poc, used) sorted: negatives closest first, then positives closest first"""
    inter = r.u(1) if idx != 0 else 0
    RPS_FORMS["inter_sps" if inter and idx != num else "inter_slice" if inter else "explicit_sps" if idx != num else "explicit_slice"] += 1
    if inter:
        delta_idx = r.ue() + 1 if idx == num else 1
        ref = sets[idx - delta_idx]
        sign = r.u(1)
        absd = r.ue() + 1
        drps = (1 - 2 * sign) * absd
        used_f, use_d = [], []
        for j in range(len(ref) + 1):
            uf = r.u(1)
            ud = 1
            if not uf:
                ud = r.u(1)
            used_f.append(uf)
            use_d.append(ud)
        neg_ref = [e for e in ref if e[0] < 0]
        pos_ref = [e for e in ref if e[0] > 0]
        nneg = len(neg_ref)
        neg, pos = [], []
        # 7-61: negatives
        for j in range(len(pos_ref) - 1, -1, -1):
            d = pos_ref[j][0] + drps
            if d < 0 and use_d[nneg + j]:
                neg.append((d, used_f[nneg + j]))
        if drps < 0 and use_d[len(ref)]:
            neg.append((drps, used_f[len(ref)]))
        for j in range(nneg):
            d = neg_ref[j][0] + drps
            if d < 0 and use_d[j]:
                neg.append((d, used_f[j]))
        # 7-62: positives
        for j in range(nneg - 1, -1, -1):
            d = neg_ref[j][0] + drps
            if d > 0 and use_d[j]:
                pos.append((d, used_f[j]))
        if drps > 0 and use_d[len(ref)]:
            pos.append((drps, used_f[len(ref)]))
        for j in range(len(pos_ref)):
            d = pos_ref[j][0] + drps
            if d > 0 and use_d[nneg + j]:
                pos.append((d, used_f[nneg + j]))
        return neg + pos
    nneg = r.ue()
    npos = r.ue()
    out = []
    d = 0
    for _ in range(nneg):
        d -= r.ue() + 1
        out.append((d, r.u(1)))
    d = 0
    for _ in range(npos):
        d += r.ue() + 1
        out.append((d, r.u(1)))
    return out


# ---------------------------------------------------------------------------------------------------------------- scaling lists (7.3.4, 7.4.5)
# Table 7-6 as the symmetric 8x8 matrices it lists in diagonal scan order (intra: matrixId 0..2, inter: 3..5)
DEFAULT_INTRA_8x8 = [[16, 16, 16, 16, 17, 18, 21, 24], [16, 16, 16, 16, 17, 19, 22, 25], [16, 16, 17, 18, 20, 22, 25, 29], [16, 16, 18, 21, 24, 27, 31, 36],
                     [17, 17, 20, 24, 30, 35, 41, 47], [18, 19, 22, 27, 35, 44, 54, 65], [21, 22, 25, 31, 41, 54, 70, 88], [24, 25, 29, 36, 47, 65, 88, 115]]
DEFAULT_INTER_8x8 = [[16, 16, 16, 16, 17, 18, 20, 24], [16, 16, 16, 17, 18, 20, 24, 25], [16, 16, 17, 18, 20, 24, 25, 28], [16, 17, 18, 20, 24, 25, 28, 33],
                     [17, 18, 20, 24, 25, 28, 33, 41], [18, 20, 24, 25, 28, 33, 41, 54], [20, 24, 25, 28, 33, 41, 54, 71], [24, 25, 28, 33, 41, 54, 71, 91]]


def _diag(n):
    """up-right diagonal scan of an n x n block (6.5.3): list of (x, y)"""
    out, x, y = [], 0, 0
    while len(out) < n * n:
        while y >= 0:
            if x < n and y < n:
                out.append((x, y))
            y -= 1
            x += 1
        y, x = x, 0
    return out


def default_list(size_id, matrix_id):
    """(ScalingList entries in diagonal scan order, dc) of Tables 7-5 / 7-6"""
    if size_id == 0:
        return [16] * 16, 16
    inter = matrix_id >= (1 if size_id == 3 else 3)
    mat = DEFAULT_INTER_8x8 if inter else DEFAULT_INTRA_8x8
    return [mat[y][x] for (x, y) in _diag(8)], 16


def default_scaling_lists():
    return {(s, m): default_list(s, m) for s in range(4) for m in range(2 if s == 3 else 6)}


def parse_scaling_list_data(r, lists):
    for s in range(4):
        for m in range(2 if s == 3 else 6):
            if not r.u(1):                                   # scaling_list_pred_mode_flag
                delta = r.ue()
                if delta > m:
                    raise ValueError("scaling_list_pred_matrix_id_delta")
                lists[(s, m)] = default_list(s, m) if delta == 0 else lists[(s, m - delta)]
            else:
                nxt, dc = 8, 16
                if s >= 2:
                    dc = r.se() + 8
                    nxt = dc
                coefs = []
                for _ in range(16 if s == 0 else 64):
                    nxt = (nxt + r.se() + 256) % 256
                    coefs.append(nxt)
                lists[(s, m)] = (coefs, dc)


def scaling_factor(lists, size_id, matrix_id):
    """m[y, x] of 8.6.4.2 for an (4 << size_id)-square transform block (7.4.5)"""
    n = 4 << size_id
    coefs, dc = lists[(size_id, matrix_id)]
    m = np.zeros((n, n), np.int64)
    if size_id == 0:
        for i, (x, y) in enumerate(_diag(4)):
            m[y, x] = coefs[i]
        return m
    rep = n >> 3
    for i, (x, y) in enumerate(_diag(8)):
        m[y * rep:(y + 1) * rep, x * rep:(x + 1) * rep] = coefs[i]
    if size_id >= 2:
        m[0, 0] = dc
    return m


def parse_sps(rbsp):
    r = Bits(rbsp)
    r.u(16)
    s = {}
    r.u(4)
    max_sub = r.u(3)
    r.u(1)
    parse_ptl(r, max_sub)
    s["id"] = r.ue()
    if r.ue() != 1:
        raise ValueError("chroma format")
    s["w"] = r.ue()
    s["h"] = r.ue()
    s["crop"] = (0, 0, 0, 0)
    if r.u(1):
        s["crop"] = (2 * r.ue(), 2 * r.ue(), 2 * r.ue(), 2 * r.ue())      # left, right, top, bottom in luma samples
    if r.ue() or r.ue():
        raise ValueError("bit depth")
    s["poc_bits"] = r.ue() + 4
    present = r.u(1)
    for _ in range(0 if present else max_sub, max_sub + 1):
        r.ue()
        s["reorder"] = r.ue()                # sps_max_num_reorder_pics of the highest sub-layer: how many pictures may wait for their turn (C.5.2)
        r.ue()
    s["min_cb"] = r.ue() + 3
    s["ctb"] = s["min_cb"] + r.ue()
    s["min_tb"] = r.ue() + 2
    s["max_tb"] = s["min_tb"] + r.ue()
    s["th_inter"] = r.ue()
    s["th_intra"] = r.ue()
    s["scaling"] = None                      # ScalingList per (sizeId, matrixId) when scaling_list_enabled_flag
    if r.u(1):
        s["scaling"] = default_scaling_lists()
        if r.u(1):                           # sps_scaling_list_data_present_flag
            parse_scaling_list_data(r, s["scaling"])
    s["amp"] = r.u(1)
    s["sao"] = r.u(1)
    s["pcm"] = None
    if r.u(1):                               # pcm_enabled_flag: sample bit depths, the coding block sizes that may be PCM, pcm_loop_filter_disabled_flag
        dl, dc = r.u(4) + 1, r.u(4) + 1
        lo = r.ue() + 3
        s["pcm"] = {"depth": (dl, dc), "min": lo, "max": lo + r.ue(), "no_filter": r.u(1)}
    n = r.ue()
    s["rps"] = []
    for i in range(n):
        s["rps"].append(parse_st_rps(r, i, n, s["rps"]))
    s["lt"] = None                           # long_term_ref_pics_present_flag: the SPS's candidates (POC LSBs, used_by_curr_pic_lt_sps_flag)
    if r.u(1):
        s["lt"] = [(r.u(s["poc_bits"]), r.u(1)) for _ in range(r.ue())]
    s["tmvp"] = r.u(1)
    s["strong"] = r.u(1)
    return s


def parse_pps(rbsp):
    r = Bits(rbsp)
    r.u(16)
    p = {"id": r.ue(), "sps": r.ue()}
    p["dep"] = r.u(1)                        # dependent_slice_segments_enabled_flag
    p["output_flag"] = r.u(1)
    p["extra_bits"] = r.u(3)
    p["sign_hiding"] = r.u(1)
    p["cabac_init_present"] = r.u(1)
    p["nref_default"] = r.ue() + 1
    p["nref1_default"] = r.ue() + 1
    p["init_qp"] = 26 + r.se()
    p["cip"] = r.u(1)                        # constrained_intra_pred_flag
    p["tskip"] = r.u(1)
    p["cu_qp_delta"] = r.u(1)
    p["qg_depth"] = r.ue() if p["cu_qp_delta"] else 0
    p["cb_off"] = r.se()
    p["cr_off"] = r.se()
    p["slice_chroma_off"] = r.u(1)
    p["weighted_pred"] = r.u(1)
    p["weighted_bipred"] = r.u(1)
    p["tq_bypass"] = r.u(1)
    p["tiles"] = r.u(1)
    p["wpp"] = r.u(1)
    p["tile_rows"] = p["tile_cols"] = 1
    p["uniform"] = 1
    p["row_heights"] = p["col_widths"] = []
    p["lf_tiles"] = 1
    if p["tiles"]:
        p["tile_cols"] = r.ue() + 1
        p["tile_rows"] = r.ue() + 1
        p["uniform"] = r.u(1)
        if not p["uniform"]:
            p["col_widths"] = [r.ue() + 1 for _ in range(p["tile_cols"] - 1)]
            p["row_heights"] = [r.ue() + 1 for _ in range(p["tile_rows"] - 1)]
        p["lf_tiles"] = r.u(1)
    p["lf_slices"] = r.u(1)
    p["dbk_override"] = 0
    p["dbk_disabled"] = 0
    p["beta"] = p["tc"] = 0
    if r.u(1):
        p["dbk_override"] = r.u(1)
        p["dbk_disabled"] = r.u(1)
        if not p["dbk_disabled"]:
            p["beta"] = 2 * r.se()
            p["tc"] = 2 * r.se()
    p["scaling"] = None
    if r.u(1):                               # pps_scaling_list_data_present_flag
        p["scaling"] = default_scaling_lists()
        parse_scaling_list_data(r, p["scaling"])
    p["lists_mod"] = r.u(1)
    p["par_mrg"] = r.ue() + 2
    p["sh_ext"] = r.u(1)
    return p


# ----------------------------------------------------------------------------------------------- CABAC
# initValue per syntax element and initType (0: I, 1: P, 2: B) -- H.265 Tables 9-5 .. 9-37
INIT = {
    "sao_merge": ([153], [153], [153]),
    "sao_type": ([200], [185], [160]),
    "split_cu": ([139, 141, 157], [107, 139, 126], [107, 139, 126]),
    "skip": (None, [197, 185, 201], [197, 185, 201]),
    "pred_mode": (None, [149], [134]),
    "part_mode": ([184], [154, 139, 154, 154], [154, 139, 154, 154]),
    "prev_intra": ([184], [154], [183]),
    "chroma_mode": ([63], [152], [152]),
    "rqt_root": (None, [79], [79]),
    "merge_flag": (None, [110], [154]),
    "merge_idx": (None, [122], [137]),
    "ref_idx": (None, [153, 153], [153, 153]),
    "inter_pred_idc": (None, [95, 79, 63, 31, 31], [95, 79, 63, 31, 31]),
    "mvp": (None, [168], [168]),
    "split_tf": ([153, 138, 138], [124, 138, 94], [224, 167, 122]),
    "cbf_luma": ([111, 141], [153, 111], [153, 111]),
    "cbf_chroma": ([94, 138, 182, 154], [149, 107, 167, 154], [149, 92, 167, 154]),
    "mvd_gt0": (None, [140], [169]),
    "mvd_gt1": (None, [198], [198]),
    "qp_delta": ([154, 154], [154, 154], [154, 154]),
    "ts_flag": ([139, 139], [139, 139], [139, 139]),
    "tq_bypass": ([154], [154], [154]),
    "last_x": ([110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63],
               [125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108],
               [125, 110, 124, 110, 95, 94, 125, 111, 111, 79, 125, 126, 111, 111, 79, 108, 123, 93]),
    "csbf": ([91, 171, 134, 141], [121, 140, 61, 154], [121, 140, 61, 154]),
    "sig": ([111, 111, 125, 110, 110, 94, 124, 108, 124, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125,
             140, 139, 182, 182, 152, 136, 152, 136, 153, 136, 139, 111, 136, 139, 111],
            [155, 154, 139, 153, 139, 123, 123, 63, 153, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154,
             170, 153, 123, 123, 107, 121, 107, 121, 167, 151, 183, 140, 151, 183, 140],
            [170, 154, 139, 153, 139, 123, 123, 63, 124, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154,
             170, 153, 138, 138, 122, 121, 122, 121, 167, 151, 183, 140, 151, 183, 140]),
    "gt1": ([140, 92, 137, 138, 140, 152, 138, 139, 153, 74, 149, 92, 139, 107, 122, 152, 140, 179, 166, 182, 140, 227, 122, 197],
            [154, 196, 196, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 137, 169, 194, 166, 167, 154, 167, 137, 182],
            [154, 196, 167, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 122, 169, 208, 166, 167, 154, 152, 167, 182]),
    "gt2": ([138, 153, 136, 167, 152, 152], [107, 167, 91, 122, 107, 167], [107, 167, 91, 107, 107, 167]),
}
INIT["last_y"] = INIT["last_x"]


class Cabac:
    def __init__(self, tabs, data, init_type, slice_qp):
        self.lps = tabs["range_lps"]
        self.next_lps = tabs["trans_lps"]
        self.next_mps = tabs["trans_mps"]
        self.ctx = {}
        self.init_type = init_type
        self.qp = slice_qp
        self.init_contexts()
        self.start(data)

    def init_contexts(self):
        q = min(max(self.qp, 0), 51)
        self.ctx = {}
        for name, tab in INIT.items():
            vals = tab[self.init_type]
            if vals is None:
                continue
            st = []
            for v in vals:
                slope = (v >> 4) * 5 - 45
                off = ((v & 15) << 3) - 16
                pre = min(max(((slope * q) >> 4) + off, 1), 126)
                mps = 1 if pre > 63 else 0
                st.append([(pre - 64) if mps else (63 - pre), mps])
            self.ctx[name] = st

    def start(self, data):
        self.d = data
        self.pos = 0
        self.range = 510
        self.offset = self.bits(9)

    def bits(self, n):
        v = 0
        for _ in range(n):
            byte = self.d[self.pos >> 3] if (self.pos >> 3) < len(self.d) else 0
            v = (v << 1) | ((byte >> (7 - (self.pos & 7))) & 1)
            self.pos += 1
        return v

    def save(self):
        return {k: [list(e) for e in v] for k, v in self.ctx.items()}

    def load(self, saved):
        self.ctx = {k: [list(e) for e in v] for k, v in saved.items()}

    def bin(self, name, idx=0):
        c = self.ctx[name][idx]
        lps = int(self.lps[c[0]][(self.range >> 6) & 3])
        self.range -= lps
        if self.offset >= self.range:
            b = 1 - c[1]
            self.offset -= self.range
            self.range = lps
            if c[0] == 0:
                c[1] = 1 - c[1]
            c[0] = int(self.next_lps[c[0]])
        else:
            b = c[1]
            c[0] = int(self.next_mps[c[0]])
        while self.range < 256:
            self.range <<= 1
            self.offset = (self.offset << 1) | self.bits(1)
        return b

    def bypass(self):
        self.offset = (self.offset << 1) | self.bits(1)
        if self.offset >= self.range:
            self.offset -= self.range
            return 1
        return 0

    def bypass_bits(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | self.bypass()
        return v

    def terminate(self):
        self.range -= 2
        if self.offset >= self.range:
            return 1
        while self.range < 256:
            self.range <<= 1
            self.offset = (self.offset << 1) | self.bits(1)
        return 0

    def end_substream(self):
        """after a terminating bin equal to 1 (9.3.2.5): the arithmetic decoder has read up to and including the bit the encoder's
        flush wrote last -- the '1' that doubles as alignment_bit_equal_to_one -- so the next substream starts at the next byte
        boundary; returns that byte offset"""
        return (self.pos + 7) >> 3


# ----------------------------------------------------------------------------------------------- scans
def diag_scan(n):
    out = []
    x = y = 0
    stop = False
    while not stop:
        while y >= 0:
            if x < n and y < n:
                out.append((x, y))
                if len(out) == n * n:
                    stop = True
                    break
            y -= 1
            x += 1
        y = x
        x = 0
    return out


def scan_order(n, scan_idx):
    if scan_idx == 0:
        return diag_scan(n)
    if scan_idx == 1:
        return [(i % n, i // n) for i in range(n * n)]
    return [(i // n, i % n) for i in range(n * n)]


CTX_MAP_4x4 = [0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8]
LEVEL_SCALE = [40, 45, 51, 57, 64, 72]
QPC_TABLE = {30: 29, 31: 30, 32: 31, 33: 32, 34: 33, 35: 33, 36: 34, 37: 34, 38: 35, 39: 35, 40: 36, 41: 36, 42: 37}


def chroma_qp(qpi):
    qpi = min(max(qpi, 0), 57)
    if qpi < 30:
        return qpi
    if qpi >= 43:
        return qpi - 6
    return QPC_TABLE[qpi]


PART_2Nx2N, PART_2NxN, PART_Nx2N, PART_NxN, PART_2NxnU, PART_2NxnD, PART_nLx2N, PART_nRx2N = range(8)


class Picture:
    def __init__(self, w, h):
        self.w, self.h = w, h
        self.planes = [np.zeros((h, w), np.int32), np.zeros((h // 2, w // 2), np.int32), np.zeros((h // 2, w // 2), np.int32)]
        self.poc = 0
        b4w, b4h = (w + 3) // 4, (h + 3) // 4
        self.mv = np.zeros((b4h, b4w, 2, 2), np.int32)          # [list][x, y]
        self.ref_idx = np.full((b4h, b4w, 2), -1, np.int32)     # per list; -1: the list is not used (both: intra / not inter)
        self.ref_poc = np.zeros((b4h, b4w, 2), np.int32)        # POC of the reference picture (for deblocking and TMVP)
        self.ref_lt = np.zeros((b4h, b4w, 2), np.int8)          # ... was a long-term reference picture when this picture was decoded (LongTermRefPic of 8.5.3.2.9)
        self.is_ref = True
        self.is_lt = False                                      # marked "used for long-term reference"


class Decoder:
    def __init__(self, tabs):
        """tabs: dict with range_lps [64][4], trans_lps [64], trans_mps [64], dct [32][32], dst [4][4], luma_filter [4][8], chroma_filter [8][4],
        beta [52], tc [54], intra_angle [35], inv_angle [35]"""
        self.t = tabs
        self.sps = {}
        self.pps = {}
        self.dpb = []
        self.cur = None         # the picture whose slice segments are still arriving (its SliceDecoder)
        self.prev_poc_tid0 = 0
        self.out = []
        self.cvs = 0            # coded video sequences started so far
        self.after_eos = False  # an end of sequence NAL unit came: the next picture starts a coded video sequence
        self.skip_rasl = False  # NoRaslOutputFlag of the last IRAP picture (8.1.3): its RASL pictures are dropped
        self.skipping = False   # inside a dropped picture: its further slice segments go the same way
        self.waiting = []       # decoded pictures that wait for their turn in output order (C.5.2.2 "needed for output")
        self.trace = None       # optional list: ("cu", x, y, log2, pred_mode, part), ("tu", cidx, x, y, log2, [levels...]) ...

    # ------------------------------------------------------------------------------------------- NAL level
    def decode(self, stream):
        """feeds the NAL units of `stream`; self.out collects the pictures in OUTPUT order (C.5.2: a picture leaves when more pictures wait than may overtake it, when a
        new coded video sequence begins, or with flush())"""
        for nal in split_nals(stream):
            self.decode_nal(nal)
        return self.out                                            # (what has had its turn so far; flush() at the end of the stream)

    def bump(self):
        """C.5.2.4: the waiting picture with the smallest picture order count is output"""
        k = min(range(len(self.waiting)), key=lambda i: self.waiting[i]["poc"])
        self.out.append(self.waiting.pop(k))

    def flush(self):
        while self.waiting:
            self.bump()
        return self.out

    def decode_nal(self, nal):
        t = (nal[0] >> 1) & 63
        if t == 33:
            s = parse_sps(unescape(nal))
            self.sps[s["id"]] = s
        elif t == 34:
            p = parse_pps(unescape(nal))
            self.pps[p["id"]] = p
        elif t in (36, 37):
            self.after_eos = True
            self.flush()                                           # (the sequence is over: nothing waits for pictures that will not come)
        elif t <= 21 and (t <= 9 or t >= 16):
            self.tid = (nal[1] & 7) - 1
            self.decode_slice(nal, t)

    def decode_slice(self, nal, nal_type):
        rbsp = unescape(nal)
        r = Bits(rbsp)
        r.u(16)
        irap = 16 <= nal_type <= 23
        idr = nal_type in (19, 20)
        first = r.u(1)
        if irap:
            no_prior = r.u(1)                                      # no_output_of_prior_pics_flag
        pps = self.pps[r.ue()]
        sps = self.sps[pps["sps"]]
        dependent, address = 0, 0
        if first:
            # 8.1.3: IDR and BLA pictures, and a CRA picture that opens the stream or follows an end of sequence NAL unit, start a coded video sequence
            # (NoRaslOutputFlag = 1); the RASL pictures that belong to such a picture refer to pictures that do not exist -- dropped
            self.no_rasl_out = idr or nal_type in (16, 17, 18) or (irap and (self.cvs == 0 or self.after_eos))
            if irap:
                self.skip_rasl = self.no_rasl_out
                # C.5.2.2: an IDR or BLA picture with no_output_of_prior_pics_flag empties the buffer WITHOUT output of what still waits; otherwise what waits goes first
                self.drop_prior = bool(no_prior) and nal_type != 21 and self.cvs > 0 and not self.after_eos
            self.skipping = (nal_type in (8, 9) and self.skip_rasl) or (self.cvs == 0 and not irap)
        if self.skipping:
            return
        if not first:                                              # 7.3.6.1: a further slice segment of the picture under way
            if pps["dep"]:
                dependent = r.u(1)
            wc, hc = -(-sps["w"] >> sps["ctb"]), -(-sps["h"] >> sps["ctb"])
            address = r.u(max(1, (wc * hc - 1).bit_length()))
            if self.cur is None or pps["tile_rows"] > 1 or pps["tile_cols"] > 1:
                raise ValueError("slice segment without its picture, or several slices in a picture with tiles")
        if dependent:
            sh = dict(self.cur.sh)                                 # everything but the entry points is the slice's first segment's
            rps, poc = None, sh["poc"]
        else:
            sh, rps, poc = self.slice_header_body(r, sps, pps, nal_type, idr)
        self.finish_header_and_decode(r, rbsp, nal, nal_type, idr, first, dependent, address, sps, pps, sh, rps, poc)

    def slice_header_body(self, r, sps, pps, nal_type, idr):
        r.u(pps["extra_bits"])
        slice_type = r.ue()          # 0 B, 1 P, 2 I
        shown = r.u(1) if pps["output_flag"] else 1
        sh = {"type": slice_type, "intra": slice_type == 2, "b": slice_type == 0, "shown": shown}
        self.last_sh = sh                                          # (for tests that look at what a stream carries)
        rps = []
        poc = 0
        if not idr:
            lsb = r.u(sps["poc_bits"])
            mx = 1 << sps["poc_bits"]
            prev = self.prev_poc_tid0
            plsb, pmsb = prev & (mx - 1), prev - (prev & (mx - 1))
            if lsb < plsb and plsb - lsb >= mx // 2:
                msb = pmsb + mx
            elif lsb > plsb and lsb - plsb > mx // 2:
                msb = pmsb - mx
            else:
                msb = pmsb
            if 16 <= nal_type <= 23 and self.no_rasl_out:
                msb = 0
            poc = msb + lsb
            if not r.u(1):
                rps = parse_st_rps(r, len(sps["rps"]), len(sps["rps"]), sps["rps"])
            else:
                n = len(sps["rps"])
                bits = (n - 1).bit_length() if n > 1 else 0
                rps = sps["rps"][r.u(bits) if bits else 0]
                RPS_FORMS["named"] += 1
        sh["lt"] = []                                              # (POC LSBs, used by the current picture, delta_poc_msb_present_flag, DeltaPocMsbCycleLt)
        if sps["lt"] is not None and not idr:
            n_sps = r.ue() if sps["lt"] else 0
            n_pics = r.ue()
            cycle = 0
            for i in range(n_sps + n_pics):
                if i < n_sps:
                    k = r.u((len(sps["lt"]) - 1).bit_length()) if len(sps["lt"]) > 1 else 0
                    lsb, used = sps["lt"][k]
                else:
                    lsb, used = r.u(sps["poc_bits"]), r.u(1)
                present = r.u(1)
                delta = r.ue() if present else 0
                cycle = delta if i in (0, n_sps) else cycle + delta   # (7-52)
                sh["lt"].append((lsb, used, present, cycle))
        sh["tmvp"] = r.u(1) if (sps["tmvp"] and not idr) else 0
        sh["sao_luma"] = sh["sao_chroma"] = 0
        if sps["sao"]:
            sh["sao_luma"] = r.u(1)
            sh["sao_chroma"] = r.u(1)
        sh["nref"] = sh["nref1"] = 0
        sh["cabac_init"] = 0
        sh["col_idx"] = 0
        sh["col_l0"] = 1
        sh["mvd_l1_zero"] = 0
        sh["max_merge"] = 5
        if slice_type != 2:                                        # 7.3.6.1
            sh["nref"] = pps["nref_default"]
            sh["nref1"] = pps["nref1_default"] if sh["b"] else 0
            if r.u(1):
                sh["nref"] = r.ue() + 1
                if sh["b"]:
                    sh["nref1"] = r.ue() + 1
            sh["list_entry"] = [None, None]
            total = sum(1 for _, used in rps if used) + sum(1 for e in sh["lt"] if e[1])      # NumPicTotalCurr
            if pps["lists_mod"] and total > 1:                         # ref_pic_lists_modification() (7.3.6.2)
                bits = (total - 1).bit_length()
                for X in range(2 if sh["b"] else 1):
                    if r.u(1):
                        sh["list_entry"][X] = [r.u(bits) for _ in range(sh["nref1"] if X else sh["nref"])]
            if sh["b"]:
                sh["mvd_l1_zero"] = r.u(1)
            if pps["cabac_init_present"]:
                sh["cabac_init"] = r.u(1)
            if sh["tmvp"]:
                if sh["b"]:
                    sh["col_l0"] = r.u(1)
                if (sh["nref"] if sh["col_l0"] else sh["nref1"]) > 1:
                    sh["col_idx"] = r.ue()
            sh["wp"] = None
            if (pps["weighted_bipred"] if sh["b"] else pps["weighted_pred"]):
                # pred_weight_table() (7.3.6.3) and the variables 7.4.7.3 derives: sh["wp"][X][i] = ((w, o) luma, (w, o) Cb, (w, o) Cr)
                ld = r.ue()
                cd = ld + r.se()
                wp = []
                for X in range(2 if sh["b"] else 1):
                    n = sh["nref1"] if X else sh["nref"]
                    lf = [r.u(1) for _ in range(n)]
                    cf = [r.u(1) for _ in range(n)]
                    ent = []
                    for i in range(n):
                        lw, lo = 1 << ld, 0
                        if lf[i]:
                            lw += r.se()
                            lo = r.se()
                        ch = []
                        for j in range(2):
                            cw, co = 1 << cd, 0
                            if cf[i]:
                                cw += r.se()
                                co = min(max(128 + r.se() - ((128 * cw) >> cd), -128), 127)
                            ch.append((cw, co))
                        ent.append(((lw, lo), ch[0], ch[1]))
                    wp.append(ent)
                sh["wp"] = (ld, cd, wp)
            sh["max_merge"] = 5 - r.ue()
        sh["qp"] = pps["init_qp"] + r.se()
        sh["cb_off"] = sh["cr_off"] = 0
        if pps["slice_chroma_off"]:
            sh["cb_off"] = r.se()
            sh["cr_off"] = r.se()
        sh["dbk_disabled"] = pps["dbk_disabled"]
        sh["beta"], sh["tc"] = pps["beta"], pps["tc"]
        if pps["dbk_override"] and r.u(1):
            sh["dbk_disabled"] = r.u(1)
            if not sh["dbk_disabled"]:
                sh["beta"] = 2 * r.se()
                sh["tc"] = 2 * r.se()
        sh["lf_across"] = pps["lf_slices"]                         # slice_loop_filter_across_slices_enabled_flag: the PPS's value unless sent
        if pps["lf_slices"] and (sh["sao_luma"] or sh["sao_chroma"] or not sh["dbk_disabled"]):
            sh["lf_across"] = r.u(1)
        return sh, rps, poc

    def finish_header_and_decode(self, r, rbsp, nal, nal_type, idr, first, dependent, address, sps, pps, sh, rps, poc):
        entry = []
        if pps["tiles"] or pps["wpp"]:
            n = r.ue()
            if n:
                ln = r.ue() + 1
                entry = [r.u(ln) + 1 for _ in range(n)]
        if pps["sh_ext"]:
            for _ in range(r.ue()):
                r.u(8)
        r.u(1)
        r.align()
        data = rbsp[r.pos >> 3:]
        # entry points count bytes of the NAL unit payload including emulation prevention bytes: map them into the rbsp
        starts = [0]
        if entry:
            # entry points count bytes of the NAL unit including emulation prevention bytes (7.4.7.1): map raw offsets to rbsp offsets
            raw = bytes(nal)
            rbsp_before = []          # rbsp_before[i] = number of rbsp bytes in raw[:i]
            z = k = 0
            for b_ in raw:
                rbsp_before.append(k)
                if z >= 2 and b_ == 3:
                    z = 0
                    continue
                k += 1
                z = z + 1 if b_ == 0 else 0
            rbsp_before.append(k)
            hdr = r.pos >> 3
            raw0 = max(i for i in range(len(raw) + 1) if rbsp_before[i] == hdr)     # (past an emulation prevention byte that sits right before the data)
            acc = raw0
            for e in entry:
                acc += e
                starts.append(rbsp_before[acc] - hdr)
        if not first:
            # a further segment of the picture under way (one tile): an independent slice brings its own header (its SliceQpY), the picture's state goes on
            sl = self.cur
            if not dependent:
                for k in ("type", "nref", "nref1", "tmvp", "sao_luma", "sao_chroma", "cabac_init", "max_merge", "dbk_disabled", "beta", "tc", "cb_off", "cr_off", "col_idx", "col_l0", "mvd_l1_zero"):
                    if sh[k] != sl.sh[k]:
                        raise ValueError("slices of one picture with different " + k)
                sh["poc"] = poc
                sl.sh = sh
            if sl.run_segment(data, starts, address, dependent):
                self.picture_done(sl, sps, nal_type)
            return
        if self.cur is not None:
            raise ValueError("a picture's slice segments did not complete")
        # ---- reference picture set (8.3.2) and list (8.3.4)
        self.pre_refs = list(self.dpb)                                 # (what the DPB held when this picture arrived: the sources of stand-ins, missing_ref)
        for p in self.dpb:
            p.is_ref = False
            p.fresh = False
        if self.no_rasl_out and 16 <= nal_type <= 23:
            self.dpb = []                                              # 8.3.2: nothing that came before is a reference picture any more
            rps = [(d, 0) for d, _ in rps]                             # (what a CRA or BLA picture's set names is for its RASL pictures)
        before = [poc + d for d, used in rps if d < 0 and used]
        after = [poc + d for d, used in rps if d > 0 and used]
        keep = [poc + d for d, _ in rps]
        # 8.3.2: the long-term entries first, among all pictures still in the buffer -- by the POC's LSBs, or by the whole POC when delta_poc_msb_present_flag is set;
        # what they name is a long-term reference picture from now on; the short-term entries name pictures among the rest
        mx = 1 << sps["poc_bits"]
        lt_curr = []
        for p in self.dpb:
            p.keep_lt = False
        for lsb, used, present, cycle in sh.get("lt", []):
            full = poc - cycle * mx - (poc & (mx - 1)) + lsb
            hit = [p for p in self.dpb if (p.poc == full if present else (p.poc & (mx - 1)) == lsb)]
            for p in hit:
                p.is_lt = True
                p.keep_lt = True
            if used:
                if not hit and not sh["intra"]:
                    at = full
                    if not present:
                        at = poc - (poc & (mx - 1)) + lsb
                        if at >= poc:
                            at -= mx
                    hit = [("missing", at)]                            # (its stand-in is made once the set has been applied: it copies a picture the set keeps)
                lt_curr.append(hit[0] if hit else None)
        for p in self.dpb:
            p.is_ref = p.keep_lt or (not p.is_lt and p.poc in keep)
        self.dpb = [p for p in self.dpb if p.is_ref]
        lt_curr = [self.missing_ref(sps, e[1], True) if isinstance(e, tuple) else e for e in lt_curr]
        def by_poc(q):
            for p in self.dpb:
                if p.poc == q and not p.is_lt:
                    return p
            return self.missing_ref(sps, q, False)
        # 8.3.4: list 0 starts with the pictures before the current one, list 1 with the ones after it, the long-term ones close both; short lists repeat
        c0 = [] if sh["intra"] else [by_poc(q) for q in before + after] + lt_curr
        c1 = [] if sh["intra"] else [by_poc(q) for q in after + before] + lt_curr
        le = sh.get("list_entry") or [None, None]                    # (a modified list: entries of the temporary list in the slice's order)
        refs = [[c0[le[0][i] if le[0] else i % len(c0)] for i in range(sh["nref"])] if sh["nref"] else [],
                [c1[le[1][i] if le[1] else i % len(c1)] for i in range(sh["nref1"])] if sh["nref1"] else []]
        if self.no_rasl_out and 16 <= nal_type <= 23:
            self.cvs += 1
            if self.drop_prior:
                self.waiting = []
            self.flush()
        else:
            while len(self.waiting) > sps["reorder"]:              # C.5.2.2: room for the current picture
                self.bump()
        self.after_eos = False
        pic = Picture(sps["w"], sps["h"])
        pic.poc = poc
        sh["poc"] = poc
        sl = SliceDecoder(self, sps, pps, sh, pic, refs, data, starts)
        if pps["tile_rows"] > 1 or pps["tile_cols"] > 1:
            sl.run()                                               # tiles: one slice per picture
            sl.loop_filters()
            self.picture_done(sl, sps, nal_type, filtered=True)
        else:
            self.cur = sl
            if sl.run_segment(data, starts, 0, 0):
                self.picture_done(sl, sps, nal_type)

    def missing_ref(self, sps, poc, is_lt):
        """a reference picture that never arrived (its access unit was lost): a stand-in with its picture order count joins the DPB, without motion, never output --
        a copy of the reference picture nearest in output order among those the DPB held when the current picture arrived (of two equally near the earlier one), mid-grey
        when there is none.  This project's concealment
        rule (oracle/hevc_dec.c missing_ref); not the standard's business"""
        pic = Picture(sps["w"], sps["h"])
        cands = [p for p in self.pre_refs if not getattr(p, "fresh", False)]
        if cands:
            src = min(cands, key=lambda p: (abs(p.poc - poc), p.poc))
            for dst, pl in zip(pic.planes, src.planes):
                dst[:] = pl
        else:
            for pl in pic.planes:
                pl[:] = 128
        pic.poc, pic.is_ref, pic.is_lt, pic.keep_lt, pic.fresh = poc, True, is_lt, False, True
        self.dpb.append(pic)
        self.concealed = getattr(self, "concealed", 0) + 1
        return pic

    def picture_done(self, sl, sps, nal_type, filtered=False):
        if not filtered:
            sl.loop_filters()
        self.cur = None
        pic, poc = sl.pic, sl.sh["poc"]
        self.dpb.append(pic)
        if self.tid == 0 and not (nal_type <= 14 and (nal_type & 1) == 0) and not (6 <= nal_type <= 9):
            self.prev_poc_tid0 = poc           # prevTid0Pic (8.3.1): TemporalId 0; RASL / RADL / sub-layer non-reference pictures excluded
        cl, cr_, ct, cb = sps["crop"]
        y = pic.planes[0][ct:sps["h"] - cb, cl:sps["w"] - cr_]
        u = pic.planes[1][ct // 2:(sps["h"] - cb) // 2, cl // 2:(sps["w"] - cr_) // 2]
        v = pic.planes[2][ct // 2:(sps["h"] - cb) // 2, cl // 2:(sps["w"] - cr_) // 2]
        if not sl.sh["shown"]:
            return                                                     # pic_output_flag = 0: decoded, kept as a reference, never handed out
        self.waiting.append({"poc": poc, "cvs": self.cvs, "i420": np.concatenate([y.reshape(-1), u.reshape(-1), v.reshape(-1)]).astype(np.uint8),
                         "width": y.shape[1], "height": y.shape[0]})
        while len(self.waiting) > sps["reorder"]:                  # C.5.2.3: more pictures wait than may overtake one
            self.bump()


class SliceDecoder:
    def __init__(self, dec, sps, pps, sh, pic, refs, data, starts):
        self.dec, self.t, self.sps, self.pps, self.sh, self.pic, self.refs, self.data, self.starts = dec, dec.t, sps, pps, sh, pic, refs, data, starts
        self.w, self.h = sps["w"], sps["h"]
        self.ctb_log2 = sps["ctb"]
        self.ctb = 1 << self.ctb_log2
        self.wc = (self.w + self.ctb - 1) >> self.ctb_log2
        self.hc = (self.h + self.ctb - 1) >> self.ctb_log2
        m = 1 << sps["min_cb"]
        self.mcw, self.mch = (self.w + m - 1) // m, (self.h + m - 1) // m
        self.cu_skip = np.zeros((self.mch, self.mcw), np.int8)
        self.cu_depth = np.zeros((self.mch, self.mcw), np.int8)
        self.cu_pred = np.full((self.mch, self.mcw), -1, np.int8)          # -1 not decoded, 0 inter, 1 intra
        b4w, b4h = (self.w + 3) // 4, (self.h + 3) // 4
        self.decoded4 = np.zeros((b4h, b4w), np.int8)                       # 4x4 blocks whose samples are reconstructed (intra availability)
        self.intra_mode = np.full((b4h, b4w), 1, np.int8)                   # luma intra prediction mode per 4x4 (DC when not intra)
        self.qp_y = np.zeros((b4h, b4w), np.int32)
        self.tu_nz = np.zeros((b4h, b4w), np.int8)                          # luma transform block has non-zero coefficients
        self.bypass = np.zeros((b4h, b4w), np.int8)                         # cu_transquant_bypass_flag: the loop filters leave these samples alone
        self.cu_bypass = 0
        # the active scaling lists (7.4.5): the PPS's when it carries any, else the SPS's (the default ones without data); None: flat 16
        self.scaling = None if sps["scaling"] is None else (pps["scaling"] if pps["scaling"] is not None else sps["scaling"])
        self.edge_v = np.zeros((b4h, b4w), np.int8)                         # 1: PU edge, 2: TU edge on the left side of this 4x4
        self.edge_h = np.zeros((b4h, b4w), np.int8)
        self.sao = {}
        self.c = None
        self.ctb_slice = [-1] * (self.wc * self.hc)                         # SliceAddrRs of every coding tree block decoded so far
        self.slice_addr, self.ctbs_done, self.wpp_saved, self.ds_saved = 0, 0, None, None
        self.ctb_order = [0] * (self.wc * self.hc)                          # decoding order of the coding tree blocks
        self.ctb_lf = [1] * (self.wc * self.hc)                             # slice_loop_filter_across_slices_enabled_flag of each block's slice
        self.n_ctb = 0
        # tile rows (6.5.1)
        tr = pps["tile_rows"]
        if pps["uniform"]:
            bd = [(i * self.hc) // tr for i in range(tr + 1)]
        else:
            bd = [0]
            for hgt in pps["row_heights"]:
                bd.append(bd[-1] + hgt)
            bd.append(self.hc)
        self.tile_bd = bd
        self.tile_of_row = [max(i for i in range(tr) if bd[i] <= cy) for cy in range(self.hc)]
        # tile columns
        tcn = pps["tile_cols"]
        if pps["uniform"]:
            cbd = [(i * self.wc) // tcn for i in range(tcn + 1)]
        else:
            cbd = [0]
            for wid in pps["col_widths"]:
                cbd.append(cbd[-1] + wid)
            cbd.append(self.wc)
        self.col_bd = cbd
        self.tile_of_col = [max(i for i in range(tcn) if cbd[i] <= cx) for cx in range(self.wc)]
        self.log2_qg = self.ctb_log2 - pps["qg_depth"]

    # ------------------------------------------------------------------------------------------- availability (6.4.1)
    def zaddr(self, x, y):
        ctb = (y >> self.ctb_log2) * self.wc + (x >> self.ctb_log2)
        xi, yi = (x & (self.ctb - 1)) >> 2, (y & (self.ctb - 1)) >> 2
        z = 0
        for b in range(self.ctb_log2 - 2):
            z |= ((xi >> b) & 1) << (2 * b) | ((yi >> b) & 1) << (2 * b + 1)
        return (ctb << (2 * (self.ctb_log2 - 2))) | z

    def avail(self, xc, yc, xn, yn):
        if xn < 0 or yn < 0 or xn >= self.w or yn >= self.h:
            return False
        if self.tile_of_row[yn >> self.ctb_log2] != self.tile_of_row[yc >> self.ctb_log2] or self.tile_of_col[xn >> self.ctb_log2] != self.tile_of_col[xc >> self.ctb_log2]:
            return False
        if self.ctb_slice[(yn >> self.ctb_log2) * self.wc + (xn >> self.ctb_log2)] != self.ctb_slice[(yc >> self.ctb_log2) * self.wc + (xc >> self.ctb_log2)]:
            return False                                                    # another slice (free slices, one tile; run() leaves every entry at -1)
        return self.zaddr(xn, yn) <= self.zaddr(xc, yc) and self.cu_pred[yn >> self.sps["min_cb"], xn >> self.sps["min_cb"]] >= 0

    # ------------------------------------------------------------------------------------------- slice data (7.3.8.1)
    def run(self):
        sh, pps = self.sh, self.pps
        init_type = 0 if sh["intra"] else ((1 if sh["cabac_init"] else 2) if sh["b"] else (2 if sh["cabac_init"] else 1))      # 9.3.2.2
        self.c = Cabac(self.t, self.data, init_type, sh["qp"])
        c = self.c
        sub = 0
        done = False
        ntr, ntc = pps["tile_rows"], pps["tile_cols"]
        # 6.5.1: tile after tile, the CTBs of a tile in raster order
        for tr in range(ntr):
            for tc in range(ntc):
                x0, x1 = self.col_bd[tc], self.col_bd[tc + 1]
                saved = None
                for cy in range(self.tile_bd[tr], self.tile_bd[tr + 1]):
                    tile_start = cy == self.tile_bd[tr]
                    for cx in range(x0, x1):
                        first = tr == 0 and tc == 0 and tile_start and cx == x0
                        if cx == x0 and not first:
                            if tile_start:
                                c.init_contexts()
                            elif pps["wpp"]:
                                if x1 - x0 >= 2 and saved is not None:
                                    c.load(saved)
                                else:
                                    c.init_contexts()
                        if cx == x0 and (tile_start or pps["wpp"]):
                            self.last_qp = sh["qp"]       # 8.6.1: the first quantisation group of a slice / tile / CTB row (WPP) predicts from SliceQpY
                        self.ctu(cx, cy)
                        end = c.terminate()
                        if pps["wpp"] and cx == x0 + 1:
                            saved = c.save()
                        last_in_row = cx == x1 - 1
                        tile_ends = cy + 1 == self.tile_bd[tr + 1]
                        row_ends_sub = last_in_row and (pps["wpp"] or tile_ends)
                        if end:
                            if not (tr == ntr - 1 and tc == ntc - 1 and tile_ends and last_in_row):
                                raise ValueError("early end of slice")
                            done = True
                            break
                        if row_ends_sub:
                            if not c.terminate():
                                raise ValueError("end_of_subset_one_bit")
                            sub += 1
                            c.start(self.data[self.starts[sub]:] if sub < len(self.starts) else self.data[c.end_substream():])
                    if done:
                        break
                    if pps["wpp"] and x1 - x0 < 2:
                        saved = None
                if done:
                    break
            if done:
                break

    def loop_filters(self):
        if not self.sh["dbk_disabled"]:
            self.deblock()
        if self.sh["sao_luma"] or self.sh["sao_chroma"]:
            self.apply_sao()

    # ------------------------------------------------------------------------------------------- slice data of ONE segment of a one-tile picture
    def run_segment(self, data, starts, address, dependent):
        """the coding tree blocks from `address` on until end_of_slice_segment_flag; True when the picture is complete.  9.3.1: an independent slice initialises the
        context variables (with its SliceQpY); under WPP a CTB row starts from the states behind the second block of the row above when that block is AVAILABLE
        (6.4.1: same slice), else -- a dependent segment that begins there -- from the states the previous segment ended with, else afresh; a dependent segment
        that begins anywhere else goes on with the previous segment's states.  8.6.1: the QP predictor starts again with a slice and with a CTB row under WPP."""
        sh, pps, wc = self.sh, self.pps, self.wc
        total = wc * self.hc
        init_type = 0 if sh["intra"] else ((1 if sh["cabac_init"] else 2) if sh["b"] else (2 if sh["cabac_init"] else 1))      # 9.3.2.2
        if self.c is None:
            self.c = Cabac(self.t, data, init_type, sh["qp"])
        c = self.c
        c.init_type, c.qp = init_type, sh["qp"]
        c.start(data)
        if not dependent:
            self.slice_addr = address
        a, sub, first = address, 0, True
        while True:
            if a >= total:
                raise ValueError("slice segment runs past the picture")
            cx, cy = a % wc, a // wc
            self.ctb_slice[a] = self.slice_addr
            if first and not dependent:
                c.init_contexts()
                self.last_qp = sh["qp"]
            elif pps["wpp"] and cx == 0:
                if cy > 0 and wc >= 2 and self.ctb_slice[a - wc + 1] == self.slice_addr:
                    c.load(self.wpp_saved)
                elif first and dependent and wc >= 2:
                    c.load(self.ds_saved)
                else:
                    c.init_contexts()
                self.last_qp = sh["qp"]
            elif first and dependent:
                c.load(self.ds_saved)
            first = False
            self.ctu(cx, cy)
            end = c.terminate()
            if pps["wpp"] and cx == 1:
                self.wpp_saved = c.save()
            a += 1
            self.ctbs_done += 1
            if end:
                self.ds_saved = c.save()
                break
            if pps["wpp"] and a % wc == 0:
                if not c.terminate():
                    raise ValueError("end_of_subset_one_bit")
                sub += 1
                c.start(data[starts[sub]:] if sub < len(starts) else data[c.end_substream():])
        return self.ctbs_done >= total

    def lf_ok(self, xq, yq, xp, yp):
        """may an in-loop filter working on luma location q use the sample at p?  7.4.3.3.1: not across a tile boundary when loop_filter_across_tiles_enabled_flag
        is 0; 7.4.7.1: slice_loop_filter_across_slices_enabled_flag = 0 closes the left and upper boundary of the slice that carries it -- of two slices, the boundary
        between them belongs to the one decoded later"""
        cq = (yq >> self.ctb_log2) * self.wc + (xq >> self.ctb_log2)
        cp = (yp >> self.ctb_log2) * self.wc + (xp >> self.ctb_log2)
        if cq == cp:
            return True
        tq = (self.tile_of_row[yq >> self.ctb_log2], self.tile_of_col[xq >> self.ctb_log2])
        tp = (self.tile_of_row[yp >> self.ctb_log2], self.tile_of_col[xp >> self.ctb_log2])
        if tq != tp and not self.pps["lf_tiles"]:
            return False
        if self.ctb_slice[cq] != self.ctb_slice[cp]:
            later = cq if self.ctb_order[cq] > self.ctb_order[cp] else cp
            if not self.ctb_lf[later]:
                return False
        return True

    def ctu(self, cx, cy):
        x0, y0 = cx << self.ctb_log2, cy << self.ctb_log2
        a = cy * self.wc + cx
        self.ctb_order[a] = self.n_ctb
        self.n_ctb += 1
        self.ctb_lf[a] = self.sh["lf_across"]
        if self.sh["sao_luma"] or self.sh["sao_chroma"]:
            self.parse_sao(cx, cy)
        self.quadtree(x0, y0, self.ctb_log2, 0)

    # ------------------------------------------------------------------------------------------- SAO syntax (7.3.8.3)
    def parse_sao(self, cx, cy):
        c = self.c
        a = cy * self.wc + cx
        left = cx > 0 and self.tile_of_col[cx - 1] == self.tile_of_col[cx] and self.ctb_slice[a - 1] == self.ctb_slice[a]
        up = cy > 0 and self.tile_of_row[cy - 1] == self.tile_of_row[cy] and self.ctb_slice[a - self.wc] == self.ctb_slice[a]
        if left and c.bin("sao_merge"):
            self.sao[(cx, cy)] = self.sao[(cx - 1, cy)]
            return
        if up and c.bin("sao_merge"):
            self.sao[(cx, cy)] = self.sao[(cx, cy - 1)]
            return
        p = {"type": [0, 0, 0], "off": [[0] * 4 for _ in range(3)], "band": [0, 0, 0], "eo": [0, 0, 0]}
        for ci in range(3):
            if (ci == 0 and not self.sh["sao_luma"]) or (ci > 0 and not self.sh["sao_chroma"]):
                continue
            if ci < 2:
                t = 0
                if c.bin("sao_type"):
                    t = 2 if c.bypass() else 1
                p["type"][ci] = t
            else:
                p["type"][2] = p["type"][1]
            if not p["type"][ci]:
                continue
            mag = []
            for _ in range(4):
                a = 0
                while a < 7 and c.bypass():
                    a += 1
                mag.append(a)
            if p["type"][ci] == 1:
                for i in range(4):
                    if mag[i] and c.bypass():
                        mag[i] = -mag[i]
                p["band"][ci] = c.bypass_bits(5)
            else:
                mag[2], mag[3] = -mag[2], -mag[3]
                if ci == 0:
                    p["eo"][0] = c.bypass_bits(2)
                elif ci == 1:
                    p["eo"][1] = c.bypass_bits(2)
                else:
                    p["eo"][2] = p["eo"][1]
            p["off"][ci] = mag
        self.sao[(cx, cy)] = p

    # ------------------------------------------------------------------------------------------- coding quadtree (7.3.8.4)
    def quadtree(self, x0, y0, log2, depth):
        c, sps = self.c, self.sps
        n = 1 << log2
        if x0 + n <= self.w and y0 + n <= self.h and log2 > sps["min_cb"]:
            m = sps["min_cb"]
            inc = 0
            if self.avail(x0, y0, x0 - 1, y0) and self.cu_depth[y0 >> m, (x0 - 1) >> m] > depth:
                inc += 1
            if self.avail(x0, y0, x0, y0 - 1) and self.cu_depth[(y0 - 1) >> m, x0 >> m] > depth:
                inc += 1
            split = c.bin("split_cu", inc)
        else:
            split = 1 if log2 > sps["min_cb"] else 0
        if self.pps["cu_qp_delta"] and log2 >= self.log2_qg:
            self.qp_coded = False
            self.qp_delta_val = 0
            self.qg_x, self.qg_y = x0, y0
            self.qp_prev = self.last_qp                   # QpY of the last coding unit of the previous quantisation group
        if split:
            h = n >> 1
            for k in range(4):
                x1, y1 = x0 + (k & 1) * h, y0 + (k >> 1) * h
                if x1 < self.w and y1 < self.h:
                    self.quadtree(x1, y1, log2 - 1, depth + 1)
        else:
            self.coding_unit(x0, y0, log2, depth)

    # ------------------------------------------------------------------------------------------- QpY (8.6.1)
    def qp_pred(self, xcb, ycb):
        if not self.pps["cu_qp_delta"]:
            return self.sh["qp"]
        xq, yq = self.qg_x, self.qg_y
        prev = self.qp_prev
        ctb_of = lambda x, y: (y >> self.ctb_log2) * self.wc + (x >> self.ctb_log2)
        a = prev
        if self.avail(xcb, ycb, xq - 1, yq) and ctb_of(xq - 1, yq) == ctb_of(xcb, ycb):
            a = int(self.qp_y[yq >> 2, (xq - 1) >> 2])
        b = prev
        if self.avail(xcb, ycb, xq, yq - 1) and ctb_of(xq, yq - 1) == ctb_of(xcb, ycb):
            b = int(self.qp_y[(yq - 1) >> 2, xq >> 2])
        return (a + b + 1) >> 1

    # ------------------------------------------------------------------------------------------- coding unit (7.3.8.5)
    def coding_unit(self, x0, y0, log2, depth):
        c, sps, sh = self.c, self.sps, self.sh
        n = 1 << log2
        m = sps["min_cb"]
        my0, my1, mx0, mx1 = y0 >> m, min((y0 + n + (1 << m) - 1) >> m, self.mch), x0 >> m, min((x0 + n + (1 << m) - 1) >> m, self.mcw)
        self.cu_qp_pred = self.qp_pred(x0, y0)
        skip = 0
        self.cu_bypass = c.bin("tq_bypass") if self.pps["tq_bypass"] else 0
        if self.cu_bypass:
            self.bypass[y0 >> 2:(y0 + n) >> 2, x0 >> 2:(x0 + n) >> 2] = 1
        if not sh["intra"]:
            inc = 0
            if self.avail(x0, y0, x0 - 1, y0) and self.cu_skip[y0 >> m, (x0 - 1) >> m]:
                inc += 1
            if self.avail(x0, y0, x0, y0 - 1) and self.cu_skip[(y0 - 1) >> m, x0 >> m]:
                inc += 1
            skip = c.bin("skip", inc)
        self.cu_depth[my0:my1, mx0:mx1] = depth
        self.cu_skip[my0:my1, mx0:mx1] = skip
        intra = 1 if sh["intra"] else 0
        part = PART_2Nx2N
        # QpY of the CU is known once cu_qp_delta has been parsed (or not): samples are reconstructed per transform unit below,
        # so the CU-wide record is written before the transform tree is walked with the value known so far and fixed up afterwards
        self.cur_qp = (self.cu_qp_pred + (self.qp_delta_val if self.pps["cu_qp_delta"] else 0) + 52) % 52
        if skip:
            self.cu_pred[my0:my1, mx0:mx1] = 0
            midx = self.merge_idx()
            self.pred_unit(x0, y0, n, x0, y0, n, n, 0, part, True, midx)
            self.set_qp(x0, y0, n, self.cur_qp)
            self.mark_edges(x0, y0, n, n, 2)              # (8.7.2.3: a coding block's edge is a transform block edge, residual or not)
            self.decoded4[y0 >> 2:(y0 + n) >> 2, x0 >> 2:(x0 + n) >> 2] = 1
            self.last_qp = self.cur_qp
            return
        if not sh["intra"]:
            intra = c.bin("pred_mode")
        self.cu_pred[my0:my1, mx0:mx1] = intra
        if not intra or log2 == m:
            part = self.part_mode(intra, log2)
        self.set_qp(x0, y0, n, self.cur_qp)
        self.mark_edges(x0, y0, n, n, 2)
        self.cu_intra = intra
        self.cu_part = part
        pcm = sps["pcm"]
        if intra and part == PART_2Nx2N and pcm and pcm["min"] <= log2 <= pcm["max"] and c.terminate():
            # pcm_flag (7.3.8.5) = 1: the arithmetic codeword ends here; zero bits up to the byte boundary, pcm_sample() (7.3.8.7), and the arithmetic decoder starts
            # again behind the samples with the context variables as they are (9.3.2.5).  8.4.4.1: the samples ARE the reconstruction, shifted up to 8 bits.
            off = c.end_substream()
            rb = Bits(c.d[off:])
            for ci in range(3):
                mm = n >> (1 if ci else 0)
                depth = pcm["depth"][1 if ci else 0]
                xs, ys = x0 >> (1 if ci else 0), y0 >> (1 if ci else 0)
                for y in range(mm):
                    for x in range(mm):
                        v = rb.u(depth) << (8 - depth)
                        if ys + y < self.pic.planes[ci].shape[0] and xs + x < self.pic.planes[ci].shape[1]:
                            self.pic.planes[ci][ys + y, xs + x] = v
            assert rb.pos % 8 == 0
            c.start(c.d[off + rb.pos // 8:])
            self.intra_mode[y0 >> 2:(y0 + n) >> 2, x0 >> 2:(x0 + n) >> 2] = 1      # DC for the neighbours' candidate lists (8.4.2)
            self.pic.ref_idx[y0 >> 2:(y0 + n) >> 2, x0 >> 2:(x0 + n) >> 2, :] = -1
            if pcm["no_filter"]:
                self.bypass[y0 >> 2:(y0 + n) >> 2, x0 >> 2:(x0 + n) >> 2] = 1       # pcm_loop_filter_disabled_flag: deblocking and SAO leave the samples alone
            self.decoded4[y0 >> 2:(y0 + n) >> 2, x0 >> 2:(x0 + n) >> 2] = 1
            self.last_qp = self.cur_qp
            return
        if intra:
            nparts = 4 if part == PART_NxN else 1
            pn = n >> 1 if part == PART_NxN else n
            prev = [c.bin("prev_intra") for _ in range(nparts)]
            modes = []
            for k in range(nparts):
                xp, yp = x0 + (k & 1) * pn, y0 + (k >> 1) * pn
                cand = self.mpm(xp, yp)
                if prev[k]:
                    i = 0
                    if c.bypass():
                        i = 2 if c.bypass() else 1
                    mode = cand[i]
                else:
                    mode = c.bypass_bits(5)
                    for cm in sorted(cand):
                        if mode >= cm:
                            mode += 1
                self.intra_mode[yp >> 2:(yp + pn) >> 2, xp >> 2:(xp + pn) >> 2] = mode
                modes.append(mode)
            cm = 4
            if c.bin("chroma_mode"):
                cm = c.bypass_bits(2)
            if cm == 4:
                self.chroma_pred_mode = modes[0]
            else:
                self.chroma_pred_mode = [0, 26, 10, 1][cm]
                if self.chroma_pred_mode == modes[0]:
                    self.chroma_pred_mode = 34
            self.luma_modes = modes
            self.pic.ref_idx[y0 >> 2:(y0 + n) >> 2, x0 >> 2:(x0 + n) >> 2, :] = -1
            rqt = 1
        else:
            self.inter_cu(x0, y0, log2, part)
            rqt = 1
            if not (part == PART_2Nx2N and self.last_merge):
                rqt = c.bin("rqt_root")
        if rqt:
            intra_split = 1 if (intra and part == PART_NxN) else 0
            self.max_depth = sps["th_intra"] + intra_split if intra else sps["th_inter"]
            self.inter_split = (sps["th_inter"] == 0 and not intra and part != PART_2Nx2N)
            self.transform_tree(x0, y0, x0, y0, log2, 0, 0, 1, 1, intra_split, log2)
        self.set_qp(x0, y0, n, self.cur_qp)
        self.decoded4[y0 >> 2:(y0 + n) >> 2, x0 >> 2:(x0 + n) >> 2] = 1
        self.last_qp = self.cur_qp

    def set_qp(self, x0, y0, n, qp):
        self.qp_y[y0 >> 2:(y0 + n + 3) >> 2, x0 >> 2:(x0 + n + 3) >> 2] = qp

    def mark_edges(self, x, y, w, h, kind):
        """left and top edge of a block: kind 1 = prediction / coding block edge, 2 = transform block edge"""
        if x & 7 == 0:
            a = self.edge_v[y >> 2:(y + h) >> 2, x >> 2]
            np.maximum(a, kind, out=a)
        if y & 7 == 0:
            a = self.edge_h[y >> 2, x >> 2:(x + w) >> 2]
            np.maximum(a, kind, out=a)

    def part_mode(self, intra, log2):
        c, sps = self.c, self.sps
        if intra:
            return PART_2Nx2N if c.bin("part_mode", 0) else PART_NxN
        if c.bin("part_mode", 0):
            return PART_2Nx2N
        if log2 == sps["min_cb"]:
            if log2 == 3:
                return PART_2NxN if c.bin("part_mode", 1) else PART_Nx2N
            if c.bin("part_mode", 1):
                return PART_2NxN
            return PART_Nx2N if c.bin("part_mode", 2) else PART_NxN
        if not sps["amp"]:
            return PART_2NxN if c.bin("part_mode", 1) else PART_Nx2N
        if c.bin("part_mode", 1):
            if c.bin("part_mode", 3):
                return PART_2NxN
            return PART_2NxnD if c.bypass() else PART_2NxnU
        if c.bin("part_mode", 3):
            return PART_Nx2N
        return PART_nRx2N if c.bypass() else PART_nLx2N

    def mpm(self, x, y):
        """8.4.2: candModeList"""
        def cand(xn, yn, above):
            if not self.avail(x, y, xn, yn):
                return 1
            if self.cu_pred[yn >> self.sps["min_cb"], xn >> self.sps["min_cb"]] != 1:
                return 1
            if above and (yn >> self.ctb_log2) != (y >> self.ctb_log2):
                return 1
            return int(self.intra_mode[yn >> 2, xn >> 2])
        a, b = cand(x - 1, y, False), cand(x, y - 1, True)
        if a == b:
            if a < 2:
                return [0, 1, 26]
            return [a, 2 + ((a + 29) % 32), 2 + ((a - 2 + 1) % 32)]
        third = 0 if (a != 0 and b != 0) else (1 if (a != 1 and b != 1) else 26)
        return [a, b, third]

    # ------------------------------------------------------------------------------------------- inter prediction units (7.3.8.6)
    def merge_idx(self):
        c = self.c
        mx = self.sh["max_merge"]
        if mx <= 1:
            return 0
        i = 0
        if c.bin("merge_idx"):
            i = 1
            while i < mx - 1 and c.bypass():
                i += 1
        return i

    def inter_cu(self, x0, y0, log2, part):
        n = 1 << log2
        h, q = n >> 1, n >> 2
        shapes = {PART_2Nx2N: [(0, 0, n, n)], PART_2NxN: [(0, 0, n, h), (0, h, n, h)], PART_Nx2N: [(0, 0, h, n), (h, 0, h, n)],
                  PART_NxN: [(0, 0, h, h), (h, 0, h, h), (0, h, h, h), (h, h, h, h)],
                  PART_2NxnU: [(0, 0, n, q), (0, q, n, n - q)], PART_2NxnD: [(0, 0, n, n - q), (0, n - q, n, q)],
                  PART_nLx2N: [(0, 0, q, n), (q, 0, n - q, n)], PART_nRx2N: [(0, 0, n - q, n), (n - q, 0, q, n)]}[part]
        for idx, (dx, dy, pw, ph) in enumerate(shapes):
            merge = self.c.bin("merge_flag")
            midx = self.merge_idx() if merge else 0
            self.pred_unit(x0, y0, n, x0 + dx, y0 + dy, pw, ph, idx, part, merge, midx)
            self.last_merge = merge
            self.mark_edges(x0 + dx, y0 + dy, pw, ph, 1)

    def pred_unit(self, xcb, ycb, ncb, xpb, ypb, pw, ph, part_idx, part, merge, midx):
        """motion of one prediction block as (mv0x, mv0y, ref0, mv1x, mv1y, ref1), a reference index of -1 = the list is not used"""
        c, sh = self.c, self.sh
        if merge:
            m = self.merge_candidates(xcb, ycb, ncb, xpb, ypb, pw, ph, part_idx, part)[midx]
            if m[2] >= 0 and m[5] >= 0 and pw + ph == 12:          # 8.5.3.2.2: no bi-prediction of 8x4 / 4x8 blocks
                m = (m[0], m[1], m[2], 0, 0, -1)
        else:
            idc = 0                                                 # 0: list 0, 1: list 1, 2: both (9.3.4.2)
            if sh["b"]:
                depth = int(self.cu_depth[ycb >> self.sps["min_cb"], xcb >> self.sps["min_cb"]])
                if pw + ph != 12 and c.bin("inter_pred_idc", depth):
                    idc = 2
                else:
                    idc = c.bin("inter_pred_idc", 4)

            def ref_index(n):
                ref = 0
                if n > 1 and c.bin("ref_idx", 0):
                    ref = 1
                    if n > 2 and c.bin("ref_idx", 1):
                        ref = 2
                        while ref < n - 1 and c.bypass():
                            ref += 1
                return ref

            def mvd():
                gx, gy = c.bin("mvd_gt0"), c.bin("mvd_gt0")
                g1x = c.bin("mvd_gt1") if gx else 0
                g1y = c.bin("mvd_gt1") if gy else 0

                def rest(g0, g1):
                    if not g0:
                        return 0
                    a = 1
                    if g1:
                        k = 1
                        v = 0
                        while c.bypass():                      # EG1 prefix
                            v += 1 << k
                            k += 1
                        v += c.bypass_bits(k)
                        a = v + 2
                    return -a if c.bypass() else a
                return rest(gx, g1x), rest(gy, g1y)
            wrap = lambda v: ((v + 32768) & 65535) - 32768
            m = [0, 0, -1, 0, 0, -1]
            for X in (0, 1):
                if idc == 1 - X:                                   # PRED_L1 skips list 0, PRED_L0 skips list 1
                    continue
                ref = ref_index(sh["nref1"] if X else sh["nref"])
                dx, dy = (0, 0) if (X == 1 and sh["mvd_l1_zero"] and idc == 2) else mvd()
                mvp = c.bin("mvp")
                px, py = self.amvp_candidates(xcb, ycb, ncb, xpb, ypb, pw, ph, part_idx, X, ref)[mvp]
                m[3 * X:3 * X + 3] = [wrap(px + dx), wrap(py + dy), ref]
            m = tuple(m)
        pic = self.pic
        ys, xs = slice(ypb >> 2, (ypb + ph) >> 2), slice(xpb >> 2, (xpb + pw) >> 2)
        for X in (0, 1):
            ref = m[3 * X + 2]
            pic.mv[ys, xs, X] = (m[3 * X], m[3 * X + 1]) if ref >= 0 else (0, 0)
            pic.ref_idx[ys, xs, X] = ref
            pic.ref_poc[ys, xs, X] = self.refs[X][ref].poc if ref >= 0 else 0
            pic.ref_lt[ys, xs, X] = 1 if ref >= 0 and self.refs[X][ref].is_lt else 0
        preds = [self.motion_compensate(xpb, ypb, pw, ph, m[3 * X], m[3 * X + 1], self.refs[X][m[3 * X + 2]]) for X in (0, 1) if m[3 * X + 2] >= 0]
        wp = self.sh.get("wp")
        for ci in range(3):
            sx = 1 if ci else 0
            if wp:                                                 # 8.5.3.3.4.3: explicit weights on the 14-bit predictions (shift1 = 6 at 8 bits)
                log2wd = (wp[1] if ci else wp[0]) + 6
                wo = [wp[2][X][m[3 * X + 2]][ci] for X in (0, 1) if m[3 * X + 2] >= 0]
                if len(preds) == 2:
                    v = (preds[0][ci].astype(np.int64) * wo[0][0] + preds[1][ci].astype(np.int64) * wo[1][0] + ((wo[0][1] + wo[1][1] + 1) << log2wd)) >> (log2wd + 1)
                else:
                    v = ((preds[0][ci].astype(np.int64) * wo[0][0] + (1 << (log2wd - 1))) >> log2wd) + wo[0][1]
            elif len(preds) == 2:                                    # 8.5.3.3.4.2: the rounded mean of the two 14-bit predictions
                v = (preds[0][ci] + preds[1][ci] + 64) >> 7
            else:
                v = (preds[0][ci] + 32) >> 6
            self.pic.planes[ci][ypb >> sx:(ypb + ph) >> sx, xpb >> sx:(xpb + pw) >> sx] = np.clip(v, 0, 255)

    # 6.4.2 prediction block availability
    def pb_avail(self, xcb, ycb, ncb, xpb, ypb, pw, ph, part_idx, xn, yn):
        same = xcb <= xn < xcb + ncb and ycb <= yn < ycb + ncb
        if not same:
            a = self.avail(xpb, ypb, xn, yn)
        else:
            a = not (pw << 1 == ncb and ph << 1 == ncb and part_idx == 1 and ycb + ph <= yn and xcb + pw > xn)
            # (a neighbour inside the same coding block that has not been given motion yet: only the NxN case above can name one)
        if a and self.cu_pred[yn >> self.sps["min_cb"], xn >> self.sps["min_cb"]] == 1:
            a = False
        return a

    def motion(self, x, y):
        """(mv0x, mv0y, ref0, mv1x, mv1y, ref1) of the 4x4 block at (x, y); the vector of an unused list reads as zero"""
        mv, ri = self.pic.mv[y >> 2, x >> 2], self.pic.ref_idx[y >> 2, x >> 2]
        return (int(mv[0, 0]), int(mv[0, 1]), int(ri[0]), int(mv[1, 0]), int(mv[1, 1]), int(ri[1]))

    def temporal(self, xpb, ypb, pw, ph, X, ref_idx):
        """8.5.3.2.8: temporal candidate of list X for reference index ref_idx; returns (mvx, mvy) or None"""
        sh = self.sh
        if not sh["tmvp"]:
            return None
        col = self.refs[0 if (not sh["b"] or sh["col_l0"]) else 1][sh["col_idx"]]
        cur_diff = sh["poc"] - self.refs[X][ref_idx].poc
        # NoBackwardPredFlag: no picture of either list comes after the current one in output order
        no_backward = all(p.poc <= sh["poc"] for lst in self.refs for p in lst)
        for k, (x, y) in enumerate([(xpb + pw, ypb + ph), (xpb + (pw >> 1), ypb + (ph >> 1))]):
            if k == 0 and not ((ypb >> self.ctb_log2) == (y >> self.ctb_log2) and y < self.h and x < self.w):
                continue
            x, y = (x >> 4) << 4, (y >> 4) << 4
            if x >= self.w or y >= self.h:
                continue
            ri = col.ref_idx[y >> 2, x >> 2]
            if ri[0] < 0 and ri[1] < 0:
                continue
            if ri[0] < 0:
                L = 1
            elif ri[1] < 0:
                L = 0
            else:
                L = X if no_backward else sh["col_l0"]            # (8.5.3.2.9: "mvCol ... set equal to mvLNCol ... with N being the value of collocated_from_l0_flag")
            mvx, mvy = int(col.mv[y >> 2, x >> 2, L, 0]), int(col.mv[y >> 2, x >> 2, L, 1])
            cur_lt = self.refs[X][ref_idx].is_lt
            if bool(col.ref_lt[y >> 2, x >> 2, L]) != cur_lt:
                continue                                           # 8.5.3.2.9: one reference picture long-term, the other not: nothing from this block
            col_diff = col.poc - int(col.ref_poc[y >> 2, x >> 2, L])
            if not cur_lt and col_diff != cur_diff and col_diff != 0:
                mvx, mvy = self.scale(mvx, mvy, col_diff, cur_diff)
            return (mvx, mvy)
        return None

    @staticmethod
    def scale(mvx, mvy, td, tb):
        td = min(max(td, -128), 127)
        tb = min(max(tb, -128), 127)
        tx = int((16384 + (abs(td) >> 1)) / td)                 # division with truncation towards zero
        dsf = min(max((tb * tx + 32) >> 6, -4096), 4095)

        def one(v):
            p = dsf * v
            s = -1 if p < 0 else 1
            return min(max(s * ((abs(p) + 127) >> 8), -32768), 32767)
        return one(mvx), one(mvy)

    def merge_candidates(self, xcb, ycb, ncb, xpb, ypb, pw, ph, part_idx, part):
        lvl = self.pps["par_mrg"]
        sh = self.sh
        if lvl > 2 and ncb == 8:
            xpb, ypb, pw, ph, part_idx, part = xcb, ycb, ncb, ncb, 0, PART_2Nx2N
        par = lambda xn, yn: (xpb >> lvl) == (xn >> lvl) and (ypb >> lvl) == (yn >> lvl)
        pos = {"A1": (xpb - 1, ypb + ph - 1), "B1": (xpb + pw - 1, ypb - 1), "B0": (xpb + pw, ypb - 1), "A0": (xpb - 1, ypb + ph), "B2": (xpb - 1, ypb - 1)}
        av = {}
        for k, (xn, yn) in pos.items():
            av[k] = self.pb_avail(xcb, ycb, ncb, xpb, ypb, pw, ph, part_idx, xn, yn) and not par(xn, yn)
        if part_idx == 1 and part in (PART_Nx2N, PART_nLx2N, PART_nRx2N):
            av["A1"] = False
        if part_idx == 1 and part in (PART_2NxN, PART_2NxnU, PART_2NxnD):
            av["B1"] = False
        mo = {k: self.motion(*pos[k]) if av[k] else None for k in pos}
        out = []
        if av["A1"]:
            out.append(mo["A1"])
        b1_in = av["B1"] and not (av["A1"] and mo["A1"] == mo["B1"])
        if b1_in:
            out.append(mo["B1"])
        b0_in = av["B0"] and not (av["B1"] and mo["B1"] == mo["B0"])
        if b0_in:
            out.append(mo["B0"])
        a0_in = av["A0"] and not (av["A1"] and mo["A1"] == mo["A0"])
        if a0_in:
            out.append(mo["A0"])
        b2_in = av["B2"] and not (av["A1"] and mo["A1"] == mo["B2"]) and not (av["B1"] and mo["B1"] == mo["B2"]) and \
            (int(av["A1"]) + int(b1_in) + int(b0_in) + int(a0_in) != 4)
        if b2_in:
            out.append(mo["B2"])
        mx = sh["max_merge"]
        out = out[:mx]
        if len(out) < mx:
            t0 = self.temporal(xpb, ypb, pw, ph, 0, 0)
            t1 = self.temporal(xpb, ypb, pw, ph, 1, 0) if sh["b"] else None
            if t0 is not None or t1 is not None:
                out.append(((t0 or (0, 0))[0], (t0 or (0, 0))[1], 0 if t0 is not None else -1, (t1 or (0, 0))[0], (t1 or (0, 0))[1], 0 if t1 is not None else -1))
        if sh["b"] and 1 < len(out) < mx:
            # 8.5.3.2.4: pairs (l0Cand, l1Cand) in the order of Table 8-7
            orig = len(out)
            pairs = [(0, 1), (1, 0), (0, 2), (2, 0), (1, 2), (2, 1), (0, 3), (3, 0), (1, 3), (3, 1), (2, 3), (3, 2)]
            for i0, i1 in pairs[:orig * (orig - 1)]:
                if len(out) == mx:
                    break
                a, b = out[i0], out[i1]
                if a[2] < 0 or b[5] < 0:
                    continue
                if self.refs[0][a[2]].poc == self.refs[1][b[5]].poc and (a[0], a[1]) == (b[3], b[4]):
                    continue
                out.append((a[0], a[1], a[2], b[3], b[4], b[5]))
        z = 0
        nz = min(sh["nref"], sh["nref1"]) if sh["b"] else sh["nref"]
        while len(out) < mx:
            r = z if z < nz else 0
            out.append((0, 0, r, 0, 0, r if sh["b"] else -1))
            z += 1
        return out

    def amvp_candidates(self, xcb, ycb, ncb, xpb, ypb, pw, ph, part_idx, X, ref_idx):
        target = self.refs[X][ref_idx].poc
        cur = self.sh["poc"]
        a_pos = [(xpb - 1, ypb + ph), (xpb - 1, ypb + ph - 1)]
        b_pos = [(xpb + pw, ypb - 1), (xpb + pw - 1, ypb - 1), (xpb - 1, ypb - 1)]
        av_a = [self.pb_avail(xcb, ycb, ncb, xpb, ypb, pw, ph, part_idx, x, y) for x, y in a_pos]
        av_b = [self.pb_avail(xcb, ycb, ncb, xpb, ypb, pw, ph, part_idx, x, y) for x, y in b_pos]
        is_scaled = av_a[0] or av_a[1]

        target_lt = self.refs[X][ref_idx].is_lt

        def vectors(pos):
            """the neighbour's vectors as (mvx, mvy, POC of the picture they point into, that picture is a long-term one), list X's first"""
            mo = self.motion(*pos)
            return [(mo[3 * L], mo[3 * L + 1], self.refs[L][mo[3 * L + 2]].poc, self.refs[L][mo[3 * L + 2]].is_lt) for L in (X, 1 - X) if mo[3 * L + 2] >= 0]

        def same_picture(pos):
            for mx, my, poc, _ in vectors(pos):
                if poc == target:
                    return (mx, my)
            return None

        def any_picture(pos):
            # 8.5.3.2.7 step 7: a vector into another picture counts when that picture and the target are both long-term (taken as it is) or both short-term (scaled)
            for mx, my, poc, lt in vectors(pos):
                if lt == target_lt:
                    return (mx, my) if (poc == target or lt) else self.scale(mx, my, cur - poc, cur - target)
            return None
        a = b = None
        for k in range(2):
            if av_a[k] and a is None:
                a = same_picture(a_pos[k])
        for k in range(2):
            if av_a[k] and a is None:
                a = any_picture(a_pos[k])
        for k in range(3):
            if av_b[k] and b is None:
                b = same_picture(b_pos[k])
        if not is_scaled and b is not None and a is None:
            a = b
        if not is_scaled:
            b = None
            for k in range(3):
                if av_b[k] and b is None:
                    b = any_picture(b_pos[k])
        out = []
        if a is not None:
            out.append(a)
        if b is not None and not (a is not None and a == b):
            out.append(b)
        if len(out) < 2:
            t = self.temporal(xpb, ypb, pw, ph, X, ref_idx)
            if t is not None:
                out.append(t)
        while len(out) < 2:
            out.append((0, 0))
        return out[:2]

    # ------------------------------------------------------------------------------------------- 8.5.3.3 sample interpolation
    def motion_compensate(self, x0, y0, pw, ph, mvx, mvy, ref):
        """the three planes' prediction sample arrays at 14-bit precision (8.5.3.3.3), before the weighted sample prediction"""
        lf, cf = self.t["luma_filter"], self.t["chroma_filter"]

        def fetch(plane, xs, ys):
            hh, ww = plane.shape
            return plane[np.clip(ys, 0, hh - 1)[:, None], np.clip(xs, 0, ww - 1)[None, :]]

        def interp(plane, xi, yi, fx, fy, w, h, filt, taps):
            half = taps // 2 - 1
            xs = np.arange(xi - half, xi + w + taps - 1 - half)
            ys = np.arange(yi - half, yi + h + taps - 1 - half)
            win = fetch(plane, xs, ys).astype(np.int64)
            if fx == 0 and fy == 0:
                return win[half:half + h, half:half + w] << 6
            if fy == 0:
                rows = win[half:half + h]
                return sum(int(filt[fx][k]) * rows[:, k:k + w] for k in range(taps))
            if fx == 0:
                cols = win[:, half:half + w]
                return sum(int(filt[fy][k]) * cols[k:k + h] for k in range(taps))
            tmp = sum(int(filt[fx][k]) * win[:, k:k + w] for k in range(taps))
            return sum(int(filt[fy][k]) * tmp[k:k + h] for k in range(taps)) >> 6
        out = [interp(ref.planes[0], x0 + (mvx >> 2), y0 + (mvy >> 2), mvx & 3, mvy & 3, pw, ph, lf, 8)]
        for ci in (1, 2):
            out.append(interp(ref.planes[ci], (x0 >> 1) + (mvx >> 3), (y0 >> 1) + (mvy >> 3), mvx & 7, mvy & 7, pw >> 1, ph >> 1, cf, 4))
        return out

    # ------------------------------------------------------------------------------------------- transform tree (7.3.8.8)
    def transform_tree(self, x0, y0, xb, yb, log2, depth, blk, pcb, pcr, intra_split, cu_log2):
        c, sps = self.c, self.sps
        if log2 <= sps["max_tb"] and log2 > sps["min_tb"] and depth < self.max_depth and not (intra_split and depth == 0):
            split = c.bin("split_tf", 5 - log2)
        else:
            split = 1 if (log2 > sps["max_tb"] or (intra_split and depth == 0) or (self.inter_split and depth == 0)) else 0
        cb = cr = 0
        if log2 > 2:
            if pcb:
                cb = c.bin("cbf_chroma", depth)
            if pcr:
                cr = c.bin("cbf_chroma", depth)
        else:
            cb, cr = pcb, pcr
        if split:
            h = 1 << (log2 - 1)
            for k in range(4):
                self.transform_tree(x0 + (k & 1) * h, y0 + (k >> 1) * h, x0, y0, log2 - 1, depth + 1, k, cb, cr, intra_split, cu_log2)
            return
        luma = 1
        if self.cu_intra or depth != 0 or cb or cr:
            luma = c.bin("cbf_luma", 1 if depth == 0 else 0)
        self.transform_unit(x0, y0, xb, yb, log2, depth, blk, luma, cb, cr)

    def transform_unit(self, x0, y0, xb, yb, log2, depth, blk, cbf_l, cbf_cb, cbf_cr):
        c, pps = self.c, self.pps
        n = 1 << log2
        self.mark_edges(x0, y0, n, n, 2)
        chroma_here = log2 > 2 or blk == 3
        any_c = (cbf_cb or cbf_cr)
        # 7.3.8.10: chroma cbfs count for the 4x4 luma blocks of an 8x8 parent as the parent's (cbfChroma of the transform unit syntax)
        if cbf_l or any_c:
            if pps["cu_qp_delta"] and not self.qp_coded:
                a = 0
                if c.bin("qp_delta", 0):
                    a = 1
                    while a < 5 and c.bin("qp_delta", 1):
                        a += 1
                    if a == 5:
                        k = 0
                        v = 0
                        while c.bypass():
                            v += 1 << k
                            k += 1
                        v += c.bypass_bits(k)
                        a += v
                if a and c.bypass():
                    a = -a
                self.qp_coded = True
                self.qp_delta_val = a
                self.cur_qp = ((self.cu_qp_pred + a + 52) % 52)
        qp = self.cur_qp
        # ---- luma: prediction (intra) then residual
        if self.cu_intra:
            self.intra_predict(0, x0, y0, n, int(self.intra_mode[y0 >> 2, x0 >> 2]))
        if cbf_l:
            scan_idx = 0
            if self.cu_intra and log2 <= 3:
                mode = int(self.intra_mode[y0 >> 2, x0 >> 2])
                scan_idx = 2 if 6 <= mode <= 14 else (1 if 22 <= mode <= 30 else 0)
            res = self.residual(0, log2, scan_idx, qp, self.cu_intra and log2 == 2)
            blk_ = self.pic.planes[0][y0:y0 + n, x0:x0 + n]
            blk_[:] = np.clip(blk_ + res[:blk_.shape[0], :blk_.shape[1]], 0, 255)
            self.tu_nz[y0 >> 2:(y0 + n) >> 2, x0 >> 2:(x0 + n) >> 2] = 1
        self.decoded4[y0 >> 2:(y0 + n) >> 2, x0 >> 2:(x0 + n) >> 2] = 1
        if chroma_here:
            xc, yc, cl2 = (x0 >> 1, y0 >> 1, log2 - 1) if log2 > 2 else (xb >> 1, yb >> 1, 2)
            cn = 1 << cl2
            for ci, cbf, off in ((1, cbf_cb, pps["cb_off"] + self.sh["cb_off"]), (2, cbf_cr, pps["cr_off"] + self.sh["cr_off"])):
                if self.cu_intra:
                    self.intra_predict(ci, xc, yc, cn, self.chroma_pred_mode)
                if cbf:
                    scan_idx = 0
                    if self.cu_intra and cl2 == 2:
                        mode = self.chroma_pred_mode
                        scan_idx = 2 if 6 <= mode <= 14 else (1 if 22 <= mode <= 30 else 0)
                    res = self.residual(ci, cl2, scan_idx, chroma_qp(qp + off), False)
                    b = self.pic.planes[ci][yc:yc + cn, xc:xc + cn]
                    b[:] = np.clip(b + res[:b.shape[0], :b.shape[1]], 0, 255)

    # ------------------------------------------------------------------------------------------- residual coding (7.3.8.11)
    def residual(self, cidx, log2, scan_idx, qp, dst):
        c, pps = self.c, self.pps
        n = 1 << log2
        tskip = 0
        if pps["tskip"] and log2 == 2 and not self.cu_bypass:
            tskip = c.bin("ts_flag", 1 if cidx else 0)
        if cidx == 0:
            off, shift = 3 * (log2 - 2) + ((log2 - 1) >> 2), (log2 + 1) >> 2
        else:
            off, shift = 15, log2 - 2
        pre = []
        for name in ("last_x", "last_y"):
            v = 0
            while v < (log2 << 1) - 1 and c.bin(name, off + (v >> shift)):
                v += 1
            pre.append(v)

        def suffix(p):
            if p > 3:
                nb = (p >> 1) - 1
                return (1 << nb) * (2 + (p & 1)) + c.bypass_bits(nb)
            return p
        lx, ly = suffix(pre[0]), suffix(pre[1])
        if scan_idx == 2:
            lx, ly = ly, lx
        sb_log2 = log2 - 2
        sb_scan = scan_order(1 << sb_log2, scan_idx)
        pos_scan = scan_order(4, scan_idx)
        last_sb = sb_scan.index((lx >> 2, ly >> 2))
        last_pos = pos_scan.index((lx & 3, ly & 3))
        coded = {}
        coef = np.zeros((n, n), np.int64)
        g1_state = 1
        first_sb = True
        for i in range(last_sb, -1, -1):
            xs, ys = sb_scan[i]
            right = coded.get((xs + 1, ys), 0)
            below = coded.get((xs, ys + 1), 0)
            infer_dc = False
            if i < last_sb and i > 0:
                coded[(xs, ys)] = c.bin("csbf", min(right + below, 1) + (2 if cidx else 0))
                infer_dc = True
            else:
                coded[(xs, ys)] = 1
            if not coded[(xs, ys)]:
                continue
            sig = {}
            start = 15
            if i == last_sb:
                sig[last_pos] = 1
                start = last_pos - 1
            for k in range(start, -1, -1):
                xp, yp = pos_scan[k]
                if k > 0 or not infer_dc:
                    xc_, yc_ = (xs << 2) + xp, (ys << 2) + yp
                    if log2 == 2:
                        s = CTX_MAP_4x4[(yc_ << 2) + xc_]
                    elif xc_ + yc_ == 0:
                        s = 0
                    else:
                        pc = right | (below << 1)
                        if pc == 0:
                            s = 2 if xp + yp == 0 else (1 if xp + yp < 3 else 0)
                        elif pc == 1:
                            s = 2 if yp == 0 else (1 if yp == 1 else 0)
                        elif pc == 2:
                            s = 2 if xp == 0 else (1 if xp == 1 else 0)
                        else:
                            s = 2
                        if cidx == 0:
                            if (xs, ys) != (0, 0):
                                s += 3
                            s += (9 if scan_idx == 0 else 15) if log2 == 3 else 21
                        else:
                            s += 9 if log2 == 3 else 12
                    b = c.bin("sig", s if cidx == 0 else 27 + s)
                    sig[k] = b
                    if b:
                        infer_dc = False
                else:
                    sig[k] = 1
            positions = [k for k in range(15, -1, -1) if sig.get(k)]
            if not positions:
                continue
            ctx_set = 2 if (i > 0 and cidx == 0) else 0
            if not first_sb and g1_state == 0:
                ctx_set += 1
            first_sb = False
            g1_state = 1
            absv = [1] * len(positions)
            first_g1 = -1
            for j in range(min(8, len(positions))):
                g = c.bin("gt1", ctx_set * 4 + g1_state + (16 if cidx else 0))
                if g:
                    absv[j] = 2
                    g1_state = 0
                    if first_g1 < 0:
                        first_g1 = j
                elif 0 < g1_state < 3:
                    g1_state += 1
            if first_g1 >= 0 and c.bin("gt2", ctx_set + (4 if cidx else 0)):
                absv[first_g1] = 3
            hidden = pps["sign_hiding"] and not self.cu_bypass and (positions[0] - positions[-1] > 3)
            nsign = len(positions) - 1 if hidden else len(positions)
            signs = [c.bypass() for _ in range(nsign)]
            rice = 0
            total = 0
            for j in range(len(positions)):
                base = (3 if j == first_g1 else 2) if j < 8 else 1
                if absv[j] == base:
                    p = 0
                    while p < 32 and c.bypass():
                        p += 1
                    if p <= 3:
                        rem = (p << rice) + c.bypass_bits(rice)
                    else:
                        rem = (((1 << (p - 3)) + 3 - 1) << rice) + c.bypass_bits(p - 3 + rice)
                    absv[j] = base + rem
                    if absv[j] > 3 * (1 << rice):
                        rice = min(rice + 1, 4)
                total += absv[j]
            if hidden:
                signs.append(total & 1)
            for j, k in enumerate(positions):
                xp, yp = pos_scan[k]
                v = -absv[j] if signs[j] else absv[j]
                coef[(ys << 2) + yp, (xs << 2) + xp] = min(max(v, -32768), 32767)
        if self.dec.trace is not None:
            self.dec.trace.append(("tu", cidx, log2, [int(v) for v in coef.reshape(-1)]))
        if self.cu_bypass:                   # 8.6.2: the residual is the level array itself
            return coef
        # ---- scaling (8.6.4.2: m = 16, or the scaling factors of the active lists) and transformation (8.6.4.2 / 8.6.2)
        bd = log2 + 3
        mfac = 16
        if self.scaling is not None:
            inter = 0 if self.cu_intra else 1
            mfac = scaling_factor(self.scaling, log2 - 2, (inter if log2 == 5 else 3 * inter + cidx))
        d = np.clip((coef * mfac * (LEVEL_SCALE[qp % 6] << (qp // 6)) + (1 << (bd - 1))) >> bd, -32768, 32767)
        if tskip:
            return ((d << 7) + 2048) >> 12
        m = np.array(self.t["dst"], np.int64) if dst else np.array(self.t["dct"], np.int64)[::(32 >> log2), :n][:n]
        e = np.clip((m.T @ d + 64) >> 7, -32768, 32767)          # columns first
        return ((e @ m) + 2048) >> 12

    # ------------------------------------------------------------------------------------------- intra prediction (8.4.4.2)
    def intra_predict(self, cidx, x0, y0, n, mode):
        plane = self.pic.planes[cidx]
        sh = 1 if cidx else 0
        X, Y = x0 << sh, y0 << sh

        def sample(xn, yn):
            """neighbouring sample at component position (xn, yn) or None"""
            lx, ly = xn << sh, yn << sh
            if not self.avail(X, Y, lx, ly):
                return None
            if not self.decoded4[ly >> 2, lx >> 2]:
                return None
            if self.pps["cip"] and self.cu_pred[ly >> self.sps["min_cb"], lx >> self.sps["min_cb"]] != 1:
                return None                                        # 8.4.4.2.2: constrained_intra_pred_flag -- a sample of a block that is not intra-coded is no reference sample
            return int(plane[yn, xn])
        # scan order of 8.4.4.2.2: p[-1][2n - 1] up to p[-1][-1], then p[0][-1] .. p[2n - 1][-1]
        coords = [(x0 - 1, y0 + 2 * n - 1 - i) for i in range(2 * n)] + [(x0 - 1, y0 - 1)] + [(x0 + i, y0 - 1) for i in range(2 * n)]
        # availability is constant over each 4x4 luma block: query once per sample, but cheaply
        vals = [sample(x, y) for x, y in coords]
        if all(v is None for v in vals):
            vals = [128] * len(vals)
        else:
            if vals[0] is None:
                vals[0] = next(v for v in vals if v is not None)
            for i in range(1, len(vals)):
                if vals[i] is None:
                    vals[i] = vals[i - 1]
        left = [vals[2 * n]] + [vals[2 * n - 1 - i] for i in range(2 * n)]        # left[0] = corner, left[1 + i] = p[-1][i]
        top = [vals[2 * n]] + [vals[2 * n + 1 + i] for i in range(2 * n)]
        # 8.4.4.2.3 filtering
        filt = False
        if cidx == 0 and mode != 1 and n != 4:
            dist = min(abs(mode - 26), abs(mode - 10))
            filt = dist > {8: 7, 16: 1, 32: 0}[n]
        if filt:
            if self.sps["strong"] and n == 32 and abs(top[0] + top[2 * n] - 2 * top[n]) < 8 and abs(left[0] + left[2 * n] - 2 * left[n]) < 8:
                c0, tl, ll = top[0], top[2 * n], left[2 * n]
                top = [c0] + [((64 - i) * c0 + i * tl + 32) >> 6 for i in range(1, 2 * n)] + [tl]
                left = [c0] + [((64 - i) * c0 + i * ll + 32) >> 6 for i in range(1, 2 * n)] + [ll]
            else:
                nt = [(left[1] + 2 * top[0] + top[1] + 2) >> 2] + [(top[i - 1] + 2 * top[i] + top[i + 1] + 2) >> 2 for i in range(1, 2 * n)] + [top[2 * n]]
                nl = [nt[0]] + [(left[i - 1] + 2 * left[i] + left[i + 1] + 2) >> 2 for i in range(1, 2 * n)] + [left[2 * n]]
                top, left = nt, nl
        out = np.zeros((n, n), np.int64)
        l2 = n.bit_length() - 1
        if mode == 0:
            for y in range(n):
                for x in range(n):
                    out[y, x] = ((n - 1 - x) * left[1 + y] + (x + 1) * top[n + 1] + (n - 1 - y) * top[1 + x] + (y + 1) * left[n + 1] + n) >> (l2 + 1)
        elif mode == 1:
            dc = (sum(left[1:n + 1]) + sum(top[1:n + 1]) + n) >> (l2 + 1)
            out[:] = dc
            if cidx == 0 and n < 32:
                out[0, 0] = (left[1] + 2 * dc + top[1] + 2) >> 2
                for x in range(1, n):
                    out[0, x] = (top[1 + x] + 3 * dc + 2) >> 2
                for y in range(1, n):
                    out[y, 0] = (left[1 + y] + 3 * dc + 2) >> 2
        else:
            angle = int(self.t["intra_angle"][mode])
            inv = int(self.t["inv_angle"][mode])
            vert = mode >= 18
            main, side = (top, left) if vert else (left, top)
            ref = {}
            for i in range(0, n + 1):
                ref[i] = main[i]
            if angle < 0:
                last = (n * angle) >> 5
                if last < -1:
                    for i in range(-1, last - 1, -1):
                        ref[i] = side[(i * inv + 128) >> 8]
            else:
                for i in range(n + 1, 2 * n + 1):
                    ref[i] = main[i]
            for a in range(n):             # along the direction perpendicular to the main side
                idx, fact = ((a + 1) * angle) >> 5, ((a + 1) * angle) & 31
                for b in range(n):
                    v = ((32 - fact) * ref[b + idx + 1] + fact * ref[b + idx + 2] + 16) >> 5 if fact else ref[b + idx + 1]
                    if vert:
                        out[a, b] = v
                    else:
                        out[b, a] = v
            if angle == 0 and cidx == 0 and n < 32:
                for k in range(n):
                    v = min(max(main[1] + ((side[1 + k] - side[0]) >> 1), 0), 255)
                    if vert:
                        out[k, 0] = v
                    else:
                        out[0, k] = v
        hh, ww = plane[y0:y0 + n, x0:x0 + n].shape
        plane[y0:y0 + n, x0:x0 + n] = out[:hh, :ww]

    # ------------------------------------------------------------------------------------------- deblocking (8.7.2)
    def deblock(self):
        pic = self.pic
        beta_t, tc_t = self.t["beta"], self.t["tc"]
        sh = self.sh
        Y = pic.planes[0]

        def motion_set(x, y):
            """the block's motion as a list of (reference picture, mvx, mvy) -- 8.7.2.4 looks at pictures, not at lists or indices"""
            return [(int(pic.ref_poc[y >> 2, x >> 2, L]), int(pic.mv[y >> 2, x >> 2, L, 0]), int(pic.mv[y >> 2, x >> 2, L, 1]))
                    for L in (0, 1) if pic.ref_idx[y >> 2, x >> 2, L] >= 0]

        def bs_of(xq, yq, xp, yp, tu_edge):
            q, p_ = motion_set(xq, yq), motion_set(xp, yp)
            if not q or not p_:
                return 2
            if tu_edge and (self.tu_nz[yq >> 2, xq >> 2] or self.tu_nz[yp >> 2, xp >> 2]):
                return 1
            if len(q) != len(p_) or sorted(m[0] for m in q) != sorted(m[0] for m in p_):
                return 1                                               # a different number of vectors, or different reference pictures
            far = lambda a, b: abs(a[1] - b[1]) >= 4 or abs(a[2] - b[2]) >= 4
            if len(q) == 1:
                return 1 if far(q[0], p_[0]) else 0
            if q[0][0] != q[1][0]:                                     # two pictures: each vector against the other block's vector into the same picture
                other = p_ if p_[0][0] == q[0][0] else [p_[1], p_[0]]
                return 1 if far(q[0], other[0]) or far(q[1], other[1]) else 0
            straight = far(q[0], p_[0]) or far(q[1], p_[1])            # both vectors into one picture: either pairing may be the close one
            crossed = far(q[0], p_[1]) or far(q[1], p_[0])
            return 1 if straight and crossed else 0
        for vertical in (True, False):
            edges = self.edge_v if vertical else self.edge_h
            bs_map = {}
            for by in range(edges.shape[0]):
                for bx in range(edges.shape[1]):
                    kind = edges[by, bx]
                    x, y = bx << 2, by << 2
                    if not kind or (vertical and (x == 0 or x & 7)) or (not vertical and (y == 0 or y & 7)):
                        continue
                    xp, yp = (x - 1, y) if vertical else (x, y - 1)
                    if not self.lf_ok(x, y, xp, yp):
                        continue                                           # 8.7.2.3 filterEdgeFlag = 0: a closed slice / tile boundary
                    bs = bs_of(x, y, xp, yp, kind == 2)
                    if bs:
                        bs_map[(x, y)] = bs
            # luma
            for (x, y), bs in bs_map.items():
                if x >= self.w or y >= self.h:
                    continue
                xp, yp = (x - 1, y) if vertical else (x, y - 1)
                qpl = (int(self.qp_y[y >> 2, x >> 2]) + int(self.qp_y[yp >> 2, xp >> 2]) + 1) >> 1
                beta = int(beta_t[min(max(qpl + sh["beta"], 0), 51)])
                tc = int(tc_t[min(max(qpl + 2 * (bs - 1) + sh["tc"], 0), 53)])

                def px(k, i):          # sample i of line k: i = -4..-1 are p3..p0, 0..3 are q0..q3
                    return (y + k, x + i) if vertical else (y + i, x + k)
                g = lambda k, i: int(Y[px(k, i)])
                keep_p, keep_q = self.bypass[yp >> 2, xp >> 2], self.bypass[y >> 2, x >> 2]      # cu_transquant_bypass on either side: that side stays as it is

                class _Side:           # writes to a side that is not to be modified are dropped
                    def __setitem__(_s, key, val, Y=Y, keep_p=keep_p, keep_q=keep_q, x=x, y=y, vertical=vertical):
                        on_p = (key[1] < x) if vertical else (key[0] < y)
                        if not (keep_p if on_p else keep_q):
                            Y[key] = val
                YW = _Side()
                dp0 = abs(g(0, -3) - 2 * g(0, -2) + g(0, -1)); dp3 = abs(g(3, -3) - 2 * g(3, -2) + g(3, -1))
                dq0 = abs(g(0, 2) - 2 * g(0, 1) + g(0, 0)); dq3 = abs(g(3, 2) - 2 * g(3, 1) + g(3, 0))
                if dp0 + dq0 + dp3 + dq3 >= beta:
                    continue

                def strong_line(k, dpq):
                    return 2 * dpq < (beta >> 2) and abs(g(k, -4) - g(k, -1)) + abs(g(k, 0) - g(k, 3)) < (beta >> 3) and abs(g(k, -1) - g(k, 0)) < ((5 * tc + 1) >> 1)
                strong = strong_line(0, dp0 + dq0) and strong_line(3, dp3 + dq3)
                dep = dp0 + dp3 < ((beta + (beta >> 1)) >> 3)
                deq = dq0 + dq3 < ((beta + (beta >> 1)) >> 3)
                for k in range(4):
                    p3, p2, p1, p0, q0, q1, q2, q3 = (g(k, i) for i in range(-4, 4))
                    if strong:
                        cl = lambda v, o: min(max(v, o - 2 * tc), o + 2 * tc)
                        YW[px(k, -1)] = cl((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3, p0)
                        YW[px(k, -2)] = cl((p2 + p1 + p0 + q0 + 2) >> 2, p1)
                        YW[px(k, -3)] = cl((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3, p2)
                        YW[px(k, 0)] = cl((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3, q0)
                        YW[px(k, 1)] = cl((p0 + q0 + q1 + q2 + 2) >> 2, q1)
                        YW[px(k, 2)] = cl((p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3, q2)
                    else:
                        d = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4
                        if abs(d) < tc * 10:
                            d = min(max(d, -tc), tc)
                            YW[px(k, -1)] = min(max(p0 + d, 0), 255)
                            YW[px(k, 0)] = min(max(q0 - d, 0), 255)
                            if dep:
                                dp = min(max((((p2 + p0 + 1) >> 1) - p1 + d) >> 1, -(tc >> 1)), tc >> 1)
                                YW[px(k, -2)] = min(max(p1 + dp, 0), 255)
                            if deq:
                                dq = min(max((((q2 + q0 + 1) >> 1) - q1 - d) >> 1, -(tc >> 1)), tc >> 1)
                                YW[px(k, 1)] = min(max(q1 + dq, 0), 255)
            # chroma: edges on the 8-sample chroma grid, boundary strength 2 only
            for (x, y), bs in bs_map.items():
                if bs != 2 or x >= self.w or y >= self.h:
                    continue
                if (vertical and x & 15) or (not vertical and y & 15):
                    continue
                xp, yp = (x - 1, y) if vertical else (x, y - 1)
                for ci, off in ((1, self.pps["cb_off"]), (2, self.pps["cr_off"])):
                    C_ = pic.planes[ci]
                    qpi = ((int(self.qp_y[y >> 2, x >> 2]) + int(self.qp_y[yp >> 2, xp >> 2]) + 1) >> 1) + off
                    qpc = chroma_qp(qpi) if qpi >= 30 else qpi
                    tc = int(tc_t[min(max(qpc + 2 + sh["tc"], 0), 53)])
                    xc, yc = x >> 1, y >> 1
                    for k in range(2):
                        pos = lambda i: (yc + k, xc + i) if vertical else (yc + i, xc + k)
                        p1, p0, q0, q1 = (int(C_[pos(i)]) for i in (-2, -1, 0, 1))
                        d = min(max((((q0 - p0) << 2) + p1 - q1 + 4) >> 3, -tc), tc)
                        if not self.bypass[yp >> 2, xp >> 2]:
                            C_[pos(-1)] = min(max(p0 + d, 0), 255)
                        if not self.bypass[y >> 2, x >> 2]:
                            C_[pos(0)] = min(max(q0 - d, 0), 255)

    # ------------------------------------------------------------------------------------------- SAO (8.7.3)
    def apply_sao(self):
        pic = self.pic
        src = [p.copy() for p in pic.planes]
        for (cx, cy), p in self.sao.items():
            for ci in range(3):
                t = p["type"][ci]
                if not t:
                    continue
                s = self.ctb >> (1 if ci else 0)
                plane, out = src[ci], pic.planes[ci]
                hh, ww = plane.shape
                x0, y0 = cx * s, cy * s
                x1, y1 = min(x0 + s, ww), min(y0 + s, hh)
                off = p["off"][ci]
                if t == 1:
                    table = [0] * 32
                    for k in range(4):
                        table[(p["band"][ci] + k) & 31] = off[k]
                    blk = plane[y0:y1, x0:x1]
                    out[y0:y1, x0:x1] = np.clip(blk + np.array(table)[blk >> 3], 0, 255)
                else:
                    dx, dy = [(1, 0), (0, 1), (1, 1), (-1, 1)][p["eo"][ci]]
                    for y in range(y0, y1):
                        for x in range(x0, x1):
                            xa, ya, xb, yb = x - dx, y - dy, x + dx, y + dy
                            if xa < 0 or ya < 0 or xb < 0 or yb < 0 or xa >= ww or xb >= ww or ya >= hh or yb >= hh:
                                continue
                            shc = 1 if ci else 0
                            if not (self.lf_ok(x << shc, y << shc, xa << shc, ya << shc) and self.lf_ok(x << shc, y << shc, xb << shc, yb << shc)):
                                continue                                   # 8.7.3.2: a neighbour across a closed boundary: edgeIdx 0
                            v = int(plane[y, x])
                            e = 2 + (v > plane[ya, xa]) - (v < plane[ya, xa]) + (v > plane[yb, xb]) - (v < plane[yb, xb])
                            if e in (0, 1, 2):
                                e = 0 if e == 2 else e + 1
                            if e:
                                out[y, x] = min(max(v + off[e - 1], 0), 255)
        # 8.7.3: samples of coding units with cu_transquant_bypass_flag are not modified
        if self.bypass.any():
            for ci in range(3):
                k = 4 >> (1 if ci else 0)
                hh, ww = src[ci].shape
                keep = np.kron(self.bypass, np.ones((k, k), np.int8))[:hh, :ww].astype(bool)
                pic.planes[ci][keep] = src[ci][keep]
