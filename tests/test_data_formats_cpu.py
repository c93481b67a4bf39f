"""CPU: the reference's on-disk formats without h5py / openpyxl in the image.

``isd_amd.h5lite`` (ctypes over libhdf5) and ``isd_amd.xlsx`` are checked against files written / read by real
h5py and pandas + openpyxl, which this image carries only inside a separate interpreter (/opt/conda/bin/python3.9);
those cross-checks are skipped where that interpreter is absent, the self round trips always run when libhdf5 loads.
"""
import os
import subprocess
import zipfile

import numpy as np
import pytest

from isd_amd import data as D
from isd_amd import h5lite, xlsx

OTHER = "/opt/conda/bin/python3.9"
needs_hdf5 = pytest.mark.skipif(not h5lite.available(), reason="no loadable libhdf5")


def _other(code):
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
    r = subprocess.run([OTHER, "-W", "ignore", "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr
    return r.stdout


def _has_other(mods):
    if not os.path.exists(OTHER):
        return False
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
    return subprocess.run([OTHER, "-c", f"import {mods}"], capture_output=True, env=env).returncode == 0


@needs_hdf5
def test_standardized_cache_hdf5_roundtrip_and_layout(tmp_path):
    rng = np.random.default_rng(0)
    per = {sid: (rng.standard_normal((6, 64, 800)).astype(np.float32), rng.integers(0, 5, 6).astype(np.uint8))
           for sid in ("01", "07")}
    path = D.save_standardized(str(tmp_path / "BCIC2020Track3.h5"), per)
    back = D.load_standardized(path)
    assert sorted(back) == ["01", "07"]
    for sid in per:
        assert back[sid][0].dtype == np.float32 and back[sid][1].dtype == np.uint8
        assert np.array_equal(back[sid][0], per[sid][0]) and np.array_equal(back[sid][1], per[sid][1])
    assert list(D.load_standardized(path, subjects=["07"])) == ["07"]
    with h5lite.File(path, "r") as f:                          # preprocess.py:220-223: {SID}/X, {SID}/Y
        assert f.keys() == ["01", "07"] and f["01"].keys() == ["X", "Y"]
        assert f["01/X"].shape == (6, 64, 800) and f["01/X"].dtype == np.float32 and f["01/Y"].dtype == np.uint8
        assert "01/Z" not in f and "02" not in f
        with pytest.raises(KeyError):
            f["02/X"]


@needs_hdf5
def test_split_file_with_gzip_and_attributes(tmp_path):
    rng = np.random.default_rng(1)
    splits = {"train": (rng.standard_normal((9, 64, 800)).astype(np.float32), rng.integers(0, 5, 9)),
              "test": (rng.standard_normal((4, 64, 800)).astype(np.float32), rng.integers(0, 5, 4))}
    path = D.save_splits(str(tmp_path / "splits.h5"), splits)
    back, meta = D.load_splits(path)
    assert sorted(back) == ["test", "train"]
    assert np.array_equal(back["train"][0], splits["train"][0]) and back["test"][1].dtype == np.uint8
    assert int(meta["n_subjects"]) == 15 and int(meta["n_classes"]) == 5 and int(meta["sfreq"]) == 250
    assert meta["classes"] == "['hello', 'help-me', 'stop', 'thank-you', 'yes']"        # str(CLASSES), as the script
    assert meta["electrodes"].startswith("['Fp1', 'Fp2'")


@needs_hdf5
@pytest.mark.skipif(not _has_other("h5py"), reason="no second interpreter with h5py")
def test_h5lite_against_real_h5py(tmp_path):
    # (1) files written the way the reference writes them (h5py) are read correctly
    p1 = str(tmp_path / "ref.h5")
    _other(f'''
import h5py, numpy as np
rng = np.random.default_rng(5)
with h5py.File({p1!r}, "w") as f:
    f.create_dataset("03/X", data=rng.standard_normal((5, 64, 800)).astype(np.float32))
    f.create_dataset("03/Y", data=rng.integers(0, 5, 5).astype(np.uint8))
    f.create_dataset("X_train", data=rng.standard_normal((3, 4, 50)).astype(np.float32), compression="gzip")
    f.create_dataset("Y_train", data=np.array([0, 4, 2], dtype=np.int64), compression="gzip")
    f.attrs["n_subjects"] = 15; f.attrs["classes"] = str(["hello", "yes"]); f.attrs["sfreq"] = 250
    g = f.create_group("epo_test"); g.create_dataset("x", data=rng.standard_normal((50, 64, 795)))
''')
    rng = np.random.default_rng(5)
    x3 = rng.standard_normal((5, 64, 800)).astype(np.float32)
    y3 = rng.integers(0, 5, 5).astype(np.uint8)
    xt = rng.standard_normal((3, 4, 50)).astype(np.float32)
    xe = rng.standard_normal((50, 64, 795))
    got = D.load_standardized(p1, subjects=["03"])
    assert np.array_equal(got["03"][0], x3) and np.array_equal(got["03"][1], y3)
    splits, meta = D.load_splits(p1)
    assert np.array_equal(splits["train"][0], xt) and splits["train"][1].tolist() == [0, 4, 2]
    assert int(meta["sfreq"]) == 250 and meta["classes"] == "['hello', 'yes']"
    with h5lite.File(p1, "r") as f:
        assert "epo_test" in f and f["epo_test"]["x"].dtype == np.float64
        assert np.array_equal(np.array(f["epo_test"]["x"]), xe)
    # (2) files written here are read correctly by h5py
    p2 = D.save_splits(str(tmp_path / "mine.h5"), {"valid": (xt, np.array([1, 2, 3]))})
    D.save_standardized(str(tmp_path / "cache.h5"), {"03": (x3, y3)})
    out = _other(f'''
import h5py, numpy as np
with h5py.File({p2!r}, "r") as f:
    print(sorted(f.keys()), f["X_valid"].dtype, f["X_valid"].shape, f["X_valid"].compression, f["Y_valid"][...].tolist())
    print(f.attrs["n_classes"], f.attrs["sfreq"], f.attrs["classes"], type(f.attrs["classes"]).__name__)
    print(repr(float(np.array(f["X_valid"]).astype(np.float64).sum())))
with h5py.File({str(tmp_path / "cache.h5")!r}, "r") as f:
    print(list(f.keys()), f["03/X"].shape, f["03/X"].dtype, f["03/Y"].dtype, repr(float(np.array(f["03/X"]).astype(np.float64).sum())))
''').splitlines()
    assert out[0] == "['X_valid', 'Y_valid'] float32 (3, 4, 50) gzip [1, 2, 3]"
    assert out[1] == "5 250 ['hello', 'help-me', 'stop', 'thank-you', 'yes'] str"
    assert float(out[2]) == float(xt.astype(np.float64).sum())
    assert out[3].startswith("['03'] (5, 64, 800) float32 uint8") and float(out[3].split()[-1]) == float(x3.astype(np.float64).sum())


def _write_min_xlsx(path, cells, strings):
    """A hand-made workbook: numbers as <v>, text through the shared-string table."""
    rows = {}
    for (r, c), v in cells.items():
        rows.setdefault(r, []).append((c, v))

    def ref(r, c):
        s = ""
        c += 1
        while c:
            c, k = divmod(c - 1, 26)
            s = chr(65 + k) + s
        return f"{s}{r + 1}"
    body = ""
    for r in sorted(rows):
        body += f'<row r="{r + 1}">'
        for c, v in sorted(rows[r]):
            if isinstance(v, str):
                body += f'<c r="{ref(r, c)}" t="s"><v>{strings.index(v)}</v></c>'
            else:
                body += f'<c r="{ref(r, c)}"><v>{v}</v></c>'
        body += "</row>"
    ns = "http://schemas.openxmlformats.org/spreadsheetml/2006/main"
    with zipfile.ZipFile(path, "w") as z:
        z.writestr("[Content_Types].xml", '<?xml version="1.0"?><Types xmlns="http://schemas.openxmlformats.org/package/2006/content-types">'
                   '<Default Extension="rels" ContentType="application/vnd.openxmlformats-package.relationships+xml"/>'
                   '<Default Extension="xml" ContentType="application/xml"/>'
                   '<Override PartName="/xl/workbook.xml" ContentType="application/vnd.openxmlformats-officedocument.spreadsheetml.sheet.main+xml"/>'
                   '<Override PartName="/xl/worksheets/sheet1.xml" ContentType="application/vnd.openxmlformats-officedocument.spreadsheetml.worksheet+xml"/>'
                   '<Override PartName="/xl/sharedStrings.xml" ContentType="application/vnd.openxmlformats-officedocument.spreadsheetml.sharedStrings+xml"/></Types>')
        z.writestr("_rels/.rels", '<?xml version="1.0"?><Relationships xmlns="http://schemas.openxmlformats.org/package/2006/relationships">'
                   '<Relationship Id="rId1" Type="http://schemas.openxmlformats.org/officeDocument/2006/relationships/officeDocument" Target="xl/workbook.xml"/></Relationships>')
        z.writestr("xl/workbook.xml", f'<?xml version="1.0"?><workbook xmlns="{ns}" xmlns:r="http://schemas.openxmlformats.org/officeDocument/2006/relationships">'
                   '<sheets><sheet name="Sheet1" sheetId="1" r:id="rId1"/></sheets></workbook>')
        z.writestr("xl/_rels/workbook.xml.rels", '<?xml version="1.0"?><Relationships xmlns="http://schemas.openxmlformats.org/package/2006/relationships">'
                   '<Relationship Id="rId1" Type="http://schemas.openxmlformats.org/officeDocument/2006/relationships/worksheet" Target="worksheets/sheet1.xml"/>'
                   '<Relationship Id="rId2" Type="http://schemas.openxmlformats.org/officeDocument/2006/relationships/sharedStrings" Target="sharedStrings.xml"/></Relationships>')
        z.writestr("xl/sharedStrings.xml", f'<?xml version="1.0"?><sst xmlns="{ns}">' + "".join(f"<si><t>{s}</t></si>" for s in strings) + "</sst>")
        z.writestr("xl/worksheets/sheet1.xml", f'<?xml version="1.0"?><worksheet xmlns="{ns}"><sheetData>{body}</sheetData></worksheet>')


def _answer_sheet_cells(rng):
    """Layout of Track3_Answer_Sheet_Test.xlsx as the loader addresses it: labels 1..5 in rows 3:53 of columns
    2, 4, ..., 30 (subject i -> column 2 (i + 1)); headers and trial numbers around them."""
    cells, labels = {(0, 0): "Track3"}, {}
    for i in range(15):
        col = 2 * (i + 1)
        cells[(1, col)] = f"Data_Sample{i + 1:02d}"
        cells[(2, col - 1)] = "Trial"
        cells[(2, col)] = "Answer"
        lab = rng.integers(1, 6, 50)
        labels[i] = lab
        for t in range(50):
            cells[(3 + t, col - 1)] = t + 1
            cells[(3 + t, col)] = int(lab[t])
    return cells, labels


def test_xlsx_reader_on_a_hand_made_answer_sheet(tmp_path):
    cells, labels = _answer_sheet_cells(np.random.default_rng(2))
    strings = sorted({v for v in cells.values() if isinstance(v, str)})
    path = str(tmp_path / "answers.xlsx")
    _write_min_xlsx(path, cells, strings)
    grid = xlsx.read_sheet(path)
    assert grid.shape == (53, 31) and grid[0, 0] == "Track3" and grid[1, 2] == "Data_Sample01" and grid[0, 1] is None
    column = D.read_answer_sheet(path)
    for i in range(15):
        assert np.array_equal(column(2 * (i + 1)), labels[i].astype(np.float64))
    assert np.isnan(xlsx.numeric_column(grid, 2, 0, 3)).all()                  # text / blanks coerce to NaN


@pytest.mark.skipif(not _has_other("openpyxl"), reason="no second interpreter with openpyxl")
def test_xlsx_reader_against_openpyxl(tmp_path):
    # a workbook written by openpyxl itself, and openpyxl's own view of both files (the rows pandas' openpyxl
    # engine builds its header=None frame from), agree with the reader
    cells, labels = _answer_sheet_cells(np.random.default_rng(3))
    mine = str(tmp_path / "mine.xlsx")
    _write_min_xlsx(mine, cells, sorted({v for v in cells.values() if isinstance(v, str)}))
    theirs = str(tmp_path / "theirs.xlsx")
    out = _other(f'''
import numpy as np, openpyxl
wb = openpyxl.Workbook(); ws = wb.active
rng = np.random.default_rng(3)
ws.cell(1, 1, "Track3")
for i in range(15):
    col = 2 * (i + 1)
    ws.cell(2, col + 1, "Data_Sample%02d" % (i + 1)); ws.cell(3, col, "Trial"); ws.cell(3, col + 1, "Answer")
    lab = rng.integers(1, 6, 50)
    for t in range(50):
        ws.cell(4 + t, col, t + 1); ws.cell(4 + t, col + 1, int(lab[t]))
wb.save({theirs!r})
for p in ({mine!r}, {theirs!r}):
    rows = list(openpyxl.load_workbook(p).active.iter_rows(values_only=True))
    print((len(rows), len(rows[0])), [int(r[6]) for r in rows[3:53]])
''').splitlines()
    for path, line in zip((mine, theirs), out):
        grid = xlsx.read_sheet(path)
        assert line.startswith(str(grid.shape))
        assert str([int(v) for v in xlsx.numeric_column(grid, 6, 3, 53)]) in line
        assert np.array_equal(xlsx.numeric_column(grid, 6, 3, 53), labels[2].astype(np.float64))


@needs_hdf5
def test_official_test_split_loader(tmp_path):
    # MATLAB v7.3 test files are HDF5 with epo_test/x; labels come from the answer sheet (preprocess.py:96-129)
    rng = np.random.default_rng(4)
    cells, labels = _answer_sheet_cells(rng)
    sheet = str(tmp_path / "Track3_Answer_Sheet_Test.xlsx")
    _write_min_xlsx(sheet, cells, sorted({v for v in cells.values() if isinstance(v, str)}))
    os.makedirs(tmp_path / "Test set")
    xs = {}
    for sid in ("01", "03"):
        xs[sid] = rng.standard_normal((50, 64, 795))
        with h5lite.File(str(tmp_path / "Test set" / f"Data_Sample{sid}.mat"), "w") as f:
            f.create_dataset("epo_test/x", data=xs[sid])
    per = D.load_test_set_per_subject(str(tmp_path), sheet)
    assert sorted(per) == ["01", "03"]
    for sid, i in (("01", 0), ("03", 2)):
        x, y = per[sid]
        assert x.shape == (50, 64, 800) and x.dtype == np.float32 and y.dtype == np.uint8
        assert np.array_equal(x[..., :795], xs[sid].astype(np.float32))
        assert np.array_equal(x[..., 795:], np.repeat(x[..., 794:795], 5, axis=-1))       # edge pad
        assert np.array_equal(y, (labels[i] - 1).astype(np.uint8))                       # 1..5 -> 0..4
    X, Y = D.load_test_set(str(tmp_path), sheet)
    assert X.shape == (100, 64, 800) and Y.shape == (100,)
