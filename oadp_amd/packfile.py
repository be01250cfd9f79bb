"""The packed feature file: one memory-mappable blob per (mode, split) + a JSON index —
``{data_root}/{task_name}.pack`` / ``.pack.json`` (a sharded sweep: ``{task_name}.r<rank>of<world>.pack``) — that
``oadp_amd.dp.PackAccessLayer`` reads with zero-copy views.  The per-image ``.pth`` files stay the reference's contract
[REF oadp/oake/base.py:44,112; oadp/dp/datasets.py:153-161] and the default; the pack holds the same tensors bit for
bit, made either afterwards from the files (``oadp_amd.dp.features.pack``) or directly by the validators
(``writer='pack'``: ``PackWriter`` below — no 354 k small files for a COCO sweep).

Format: tensors raw, C order, 64-byte aligned; index = {magic, bytes, index: key -> [[field, dtype, shape, offset],
...]} with field '' for a bare tensor (globals).  The index is written by rename, after the data it describes has been
flushed: a reader never sees an index that points past the blob, and bytes of the blob beyond ``bytes`` (a writer that
was killed between two index checkpoints) are ignored and overwritten by the next run."""
from __future__ import annotations

import json
import os
import pathlib
import threading
from typing import Any

import torch

MAGIC = 'oake-pack-1'
ALIGN = 64
DTYPES = {torch.float16: 'float16', torch.float32: 'float32', torch.float64: 'float64',
          torch.int64: 'int64', torch.int32: 'int32', torch.uint8: 'uint8', torch.bool: 'bool'}


def fields(value: Any) -> list[tuple[str, torch.Tensor]]:
    """A feature payload is a tensor (globals) or a flat dict of tensors (blocks, objects)."""
    if isinstance(value, torch.Tensor):
        return [('', value)]
    if isinstance(value, dict) and all(isinstance(v, torch.Tensor) for v in value.values()):
        if '' in value:
            raise ValueError('empty field name')
        return list(value.items())
    raise TypeError(f'cannot pack {type(value).__name__}: expected a tensor or a dict of tensors')


def blob_path(output_dir: str | os.PathLike, rank: int = 0, world: int = 1) -> pathlib.Path:
    """``<data_root>/<task_name>`` (the validators' ``output_dir``) -> the blob a rank writes."""
    d = pathlib.Path(output_dir)
    shard = '' if world <= 1 else f'.r{rank}of{world}'
    return d.with_name(d.name + shard + '.pack')


def read_index(blob: pathlib.Path) -> dict:
    meta = json.loads(blob.with_name(blob.name + '.json').read_text())
    if meta.get('magic') != MAGIC:
        raise ValueError(f'{blob}: not an OAKE feature pack')
    return meta


class PackWriter:
    """Append-only writer of one blob, with the ``AsyncWriter`` face (``submit(obj, path)`` / ``drain`` / ``close`` /
    ``bytes``) so that a validator swaps one for the other.  ``path.stem`` is the key (``<image_id:012d>``).  Appends
    are serialised by a lock (the validators call from one thread); the index is checkpointed every ``checkpoint``
    keys and at ``close``.  Opening an existing pack resumes it: ``keys`` lists what it already holds (the datasets
    skip those images, as they skip existing ``.pth`` files), and an orphaned tail beyond the index is truncated."""

    def __init__(self, blob: str | os.PathLike, checkpoint: int = 4096) -> None:
        self.blob = pathlib.Path(blob)
        self.blob.parent.mkdir(parents=True, exist_ok=True)
        self.bytes = 0  # written by THIS run (the throughput counters)
        self._index: dict[str, list] = {}
        self._offset = 0
        if self.blob.exists() and self.blob.with_name(self.blob.name + '.json').exists():
            meta = read_index(self.blob)
            if self.blob.stat().st_size < meta['bytes']:
                raise ValueError(f'{self.blob}: shorter than its index says (truncated pack?)')
            self._index, self._offset = meta['index'], meta['bytes']
        self._f = open(self.blob, 'r+b' if self.blob.exists() else 'w+b')
        self._f.truncate(self._offset)
        self._f.seek(self._offset)
        self._lock = threading.Lock()
        self._checkpoint = max(1, checkpoint)
        self._since = 0

    @property
    def keys(self) -> set[str]:
        return set(self._index)

    def submit(self, obj: Any, path: pathlib.Path) -> None:
        key = pathlib.Path(path).stem
        entry, chunks, offset = [], [], self._offset
        for name, t in fields(obj):
            if t.dtype not in DTYPES:
                raise TypeError(f'{key}: dtype {t.dtype} is not packable')
            pad = -offset % ALIGN
            data = t.detach().contiguous().cpu().numpy().tobytes()
            chunks.append(b'\0' * pad + data)
            offset += pad
            entry.append([name, DTYPES[t.dtype], list(t.shape), offset])
            offset += len(data)
        with self._lock:
            if key in self._index:  # (a DistributedSampler pads by wrap-around: the duplicate is dropped, as the
                return              # resume rule drops it for .pth files)
            # offsets were computed against self._offset read outside the lock: single submitting thread
            self._f.write(b''.join(chunks))
            self.bytes += offset - self._offset
            self._offset = offset
            self._index[key] = entry
            self._since += 1
            if self._since >= self._checkpoint:
                self._write_index()

    def _write_index(self) -> None:
        self._f.flush()
        os.fsync(self._f.fileno())
        meta = self.blob.with_name(self.blob.name + '.json')
        tmp = meta.with_name(meta.name + f'.tmp{os.getpid()}')
        tmp.write_text(json.dumps(dict(magic=MAGIC, bytes=self._offset, index=self._index)))
        os.replace(tmp, meta)
        self._since = 0

    def drain(self) -> None:
        with self._lock:
            self._write_index()

    def close(self) -> None:
        with self._lock:
            if not self._f.closed:
                self._write_index()
                self._f.close()
