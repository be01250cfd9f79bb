"""Every ``[REF file:lines]`` citation in DESIGN.md / INTEGRATION.md points at lines the reference has (build
container only: /root/reference is not on the GPU box; nothing of it is read at GPU-test time)."""
import pathlib
import re

import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]
REF = pathlib.Path('/root/reference')
SHORT = {'globals.py': 'oadp/oake/globals.py', 'blocks.py': 'oadp/oake/blocks.py', 'objects.py': 'oadp/oake/objects.py',
         'base.py': 'oadp/oake/base.py'}


@pytest.mark.skipif(not REF.exists(), reason='needs the reference checkout')
def test_reference_citations_resolve():
    checked = 0
    for doc in ('DESIGN.md', 'INTEGRATION.md'):
        for m in re.finditer(r'\[REF ([^\]]+)\]', (ROOT / doc).read_text()):
            body = m.group(1)
            if body == 'path:line':
                continue
            for part in body.split(';'):
                part = part.strip()
                if ':' not in part:
                    continue
                name, spans = part.split(':', 1)
                path = REF / SHORT.get(name, name)
                assert path.is_file(), f'{doc}: [REF {body}]: {path} does not exist'
                n_lines = len(path.read_text().splitlines())
                for span in spans.split(','):
                    span = span.strip()
                    if not re.fullmatch(r'\d+(-\d+)?', span):
                        continue
                    lo, _, hi = span.partition('-')
                    lo, hi = int(lo), int(hi or lo)
                    assert 1 <= lo <= hi <= n_lines, f'{doc}: [REF {body}]: {name} has {n_lines} lines'
                    checked += 1
    assert checked > 30
