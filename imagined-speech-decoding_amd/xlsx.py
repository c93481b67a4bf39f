"""Minimal ``.xlsx`` reader (zip + XML, standard library only) for images without ``openpyxl``.

The reference reads the competition's answer sheet with ``pd.read_excel(excel_path, header=None)`` and takes
``iloc[3:53, 2 * (i + 1)]`` (src/fast/data/preprocess.py:104, :118-121; notebooks/svm_baseline.ipynb cell 14).
``read_sheet`` returns the first worksheet as a 2-D object array in absolute sheet coordinates (row 0 = Excel
row 1, column 0 = column A, blanks = None) -- the frame ``header=None`` gives -- with numbers as float and text
as str.  ``isd_amd.data`` prefers pandas + openpyxl and falls back to this.
"""
import re
import zipfile
import xml.etree.ElementTree as ET

import numpy as np

_NS = {"m": "http://schemas.openxmlformats.org/spreadsheetml/2006/main",
       "r": "http://schemas.openxmlformats.org/officeDocument/2006/relationships",
       "p": "http://schemas.openxmlformats.org/package/2006/relationships"}
_REF = re.compile(r"([A-Z]+)([0-9]+)$")


def _col_index(letters):
    n = 0
    for ch in letters:
        n = n * 26 + (ord(ch) - 64)
    return n - 1


def _text(node):
    return "".join(t.text or "" for t in node.iter(f"{{{_NS['m']}}}t"))


def read_sheet(path, sheet=0):
    """First (or ``sheet``-th) worksheet of an .xlsx file -> object ndarray [n_rows, n_cols]."""
    with zipfile.ZipFile(path) as z:
        wb = ET.fromstring(z.read("xl/workbook.xml"))
        sheets = wb.find("m:sheets", _NS).findall("m:sheet", _NS)
        rid = sheets[sheet].get(f"{{{_NS['r']}}}id")
        rels = ET.fromstring(z.read("xl/_rels/workbook.xml.rels"))
        target = next(r.get("Target") for r in rels.findall("p:Relationship", _NS) if r.get("Id") == rid)
        target = target.lstrip("/")
        if not target.startswith("xl/"):
            target = "xl/" + target
        shared = []
        if "xl/sharedStrings.xml" in z.namelist():
            sst = ET.fromstring(z.read("xl/sharedStrings.xml"))
            shared = [_text(si) for si in sst.findall("m:si", _NS)]
        ws = ET.fromstring(z.read(target))
    cells = {}
    next_row = 0
    for row in ws.find("m:sheetData", _NS).findall("m:row", _NS):
        r = int(row.get("r")) - 1 if row.get("r") else next_row
        next_row = r + 1
        next_col = 0
        for c in row.findall("m:c", _NS):
            m = _REF.match(c.get("r") or "")
            col = _col_index(m.group(1)) if m else next_col
            next_col = col + 1
            kind, v = c.get("t", "n"), c.find("m:v", _NS)
            if kind == "inlineStr":
                node = c.find("m:is", _NS)
                val = _text(node) if node is not None else None
            elif v is None or v.text is None:
                val = None
            elif kind == "s":
                val = shared[int(v.text)]
            elif kind in ("str", "e"):
                val = v.text
            elif kind == "b":
                val = bool(int(v.text))
            else:
                val = float(v.text)
            if val is not None:
                cells[(r, col)] = val
    n_rows = max((k[0] for k in cells), default=-1) + 1
    n_cols = max((k[1] for k in cells), default=-1) + 1
    grid = np.full((n_rows, n_cols), None, dtype=object)
    for (r, col), val in cells.items():
        grid[r, col] = val
    return grid


def numeric_column(grid, col, row_lo, row_hi):
    """``pd.to_numeric(frame.iloc[row_lo:row_hi, col], errors='coerce').values`` on a ``read_sheet`` grid."""
    out = np.full(max(row_hi - row_lo, 0), np.nan)
    for i, r in enumerate(range(row_lo, row_hi)):
        if r < grid.shape[0] and col < grid.shape[1]:
            v = grid[r, col]
            if isinstance(v, bool):
                out[i] = float(v)
            elif isinstance(v, (int, float)):
                out[i] = float(v)
            elif isinstance(v, str):
                try:
                    out[i] = float(v)
                except ValueError:
                    pass
    return out
