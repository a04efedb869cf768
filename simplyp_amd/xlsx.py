"""Minimal .xlsx reader on the standard library (zipfile + xml.etree).

The reference reads its parameter workbook with ``pd.read_excel``
(``Current_Release/v0-2A/simplyP/inputs.py:44-77``, ``:119-145``), which
needs an Excel engine (xlrd/openpyxl).  Neither is a dependency of this
package: an .xlsx file is a zip of XML parts, and the SimplyP template only
uses plain value cells (numbers, shared strings, dates), so this module
recovers them directly and hands pandas a DataFrame shaped like the one
``read_excel(..., index_col=0, usecols="A,C")`` would return.
"""

import datetime as _dt
import re
import zipfile
import xml.etree.ElementTree as ET

import numpy as np
import pandas as pd

_NS = '{http://schemas.openxmlformats.org/spreadsheetml/2006/main}'
_RNS = '{http://schemas.openxmlformats.org/officeDocument/2006/relationships}'
_PKG_RNS = '{http://schemas.openxmlformats.org/package/2006/relationships}'

# Built-in number formats that denote dates/times (ECMA-376 part 1, 18.8.30).
_BUILTIN_DATE_FMTS = set(range(14, 23)) | set(range(27, 37)) | set(range(45, 48)) | set(range(50, 59))


def _col_to_idx(letters):
    n = 0
    for ch in letters:
        n = n * 26 + (ord(ch.upper()) - ord('A') + 1)
    return n - 1


def _parse_usecols(usecols):
    """'A,C' / 'B,E:G' -> sorted list of zero-based column indices."""
    out = []
    for part in usecols.split(','):
        part = part.strip()
        if ':' in part:
            a, b = part.split(':')
            out.extend(range(_col_to_idx(a), _col_to_idx(b) + 1))
        elif part:
            out.append(_col_to_idx(part))
    return sorted(set(out))


class Workbook(object):
    """Read-only view of an .xlsx file: ``sheet_names`` and ``rows(sheet)``."""

    def __init__(self, path):
        self.path = path
        self._zip = zipfile.ZipFile(path)
        self._shared = self._read_shared_strings()
        self._date_styles = self._read_date_styles()
        wb = ET.fromstring(self._zip.read('xl/workbook.xml'))
        rels = ET.fromstring(self._zip.read('xl/_rels/workbook.xml.rels'))
        targets = {r.get('Id'): r.get('Target') for r in rels.iter(_PKG_RNS + 'Relationship')}
        self._sheets = {}
        self.sheet_names = []
        for sh in wb.find(_NS + 'sheets'):
            name = sh.get('name')
            target = targets[sh.get(_RNS + 'id')]
            target = target.lstrip('/')
            if not target.startswith('xl/'):
                target = 'xl/' + target
            self._sheets[name] = target
            self.sheet_names.append(name)
        date1904 = wb.find(_NS + 'workbookPr')
        self._epoch = (_dt.datetime(1904, 1, 1) if date1904 is not None and
                       date1904.get('date1904') in ('1', 'true') else _dt.datetime(1899, 12, 30))

    def _read_shared_strings(self):
        try:
            root = ET.fromstring(self._zip.read('xl/sharedStrings.xml'))
        except KeyError:
            return []
        return [''.join(t.text or '' for t in si.iter(_NS + 't')) for si in root.findall(_NS + 'si')]

    def _read_date_styles(self):
        try:
            root = ET.fromstring(self._zip.read('xl/styles.xml'))
        except KeyError:
            return set()
        custom = {}
        numfmts = root.find(_NS + 'numFmts')
        if numfmts is not None:
            for nf in numfmts:
                custom[int(nf.get('numFmtId'))] = nf.get('formatCode', '')
        date_styles = set()
        xfs = root.find(_NS + 'cellXfs')
        if xfs is not None:
            for i, xf in enumerate(xfs):
                fid = int(xf.get('numFmtId', '0'))
                if fid in _BUILTIN_DATE_FMTS:
                    date_styles.add(i)
                elif fid in custom:
                    code = re.sub(r'"[^"]*"|\[[^\]]*\]|\\.', '', custom[fid])
                    if re.search(r'[dmyhs]', code, re.I):
                        date_styles.add(i)
        return date_styles

    def _cell_value(self, c):
        t = c.get('t')
        if t == 'inlineStr':
            return ''.join(x.text or '' for x in c.iter(_NS + 't'))
        v = c.find(_NS + 'v')
        if v is None or v.text is None:
            return None
        text = v.text
        if t == 's':
            return self._shared[int(text)]
        if t in ('str', 'e'):
            return text
        if t == 'b':
            return text == '1'
        # numeric
        style = c.get('s')
        if style is not None and int(style) in self._date_styles:
            return pd.Timestamp(self._epoch + _dt.timedelta(days=float(text)))
        if re.match(r'^-?\d+$', text):
            return int(text)
        return float(text)

    def rows(self, sheet_name):
        """List of ``{column_index: value}`` dicts, one per sheet row (1-based gaps kept)."""
        root = ET.fromstring(self._zip.read(self._sheets[sheet_name]))
        out = []
        for row in root.find(_NS + 'sheetData'):
            ridx = int(row.get('r')) - 1
            while len(out) < ridx:
                out.append({})
            cells = {}
            for c in row:
                val = self._cell_value(c)
                if val is None:
                    continue
                letters = re.match(r'[A-Z]+', c.get('r')).group(0)
                cells[_col_to_idx(letters)] = val
            out.append(cells)
        return out


def read_excel(path, sheet_name, index_col=None, usecols=None):
    """Stand-in for the ``pd.read_excel`` calls the reference makes.

    First sheet row is the header; ``usecols`` is an Excel letter spec such as
    ``"B,E:G"``; ``index_col`` indexes into the *selected* columns.  Trailing
    all-empty rows are dropped (as pandas does); empty cells become NaN.
    """
    wb = path if isinstance(path, Workbook) else Workbook(path)
    rows = wb.rows(str(sheet_name))
    if not rows:
        return pd.DataFrame()
    width = 1 + max((max(r) for r in rows if r), default=-1)
    cols = _parse_usecols(usecols) if usecols else list(range(width))
    header = [rows[0].get(c) for c in cols]
    header = ['Unnamed: %d' % i if h is None else h for i, h in enumerate(header)]
    body = [[r.get(c, np.nan) for c in cols] for r in rows[1:]]
    while body and all(isinstance(v, float) and np.isnan(v) for v in body[-1]):
        body.pop()
    df = pd.DataFrame(body, columns=header)
    if index_col is not None:
        df = df.set_index(df.columns[index_col])
        df.columns = pd.Index(list(df.columns))      # re-infer the label dtype (e.g. int sub-catchment ids)
    return df
