"""Reading the OAKE features back during detector training.

The reference's detector pipeline does three ``torch.load`` calls per training sample — one small
pickle each for globals / blocks / objects — through ``todd.datasets.PthAccessLayer``
(oadp/dp/datasets.py:137-214, configs/dp/datasets/ov_coco.py:23-32).  Two access layers with the
same ``Mapping[str, ...]`` face live here:

* ``PthAccessLayer``: the per-image ``{data_root}/{task_name}/{key}.pth`` files exactly as
  ``oadp_amd.oake`` (and the reference) write them;
* ``PackAccessLayer``: one memory-mapped blob per (mode, split) made by ``pack()`` from those files.
  A lookup is an index probe + zero-copy views of the map: no unpickling, no per-sample file open,
  and the page cache is shared by all dataloader workers.  Values are bit-identical to the
  ``.pth`` path (tests/test_dp_features.py).

``LoadCLIPFeatures`` is the pipeline step (same result keys, same filtering) on plain tensors /
arrays; the mmdet registry and ``todd.BBoxesXYXY`` are replaced by the few lines of box arithmetic
they stand for ([RECALL] ``a & b`` = pairwise intersection areas, ``indices(min_wh)`` as in
oadp_amd/oake/objects.py).
"""
from __future__ import annotations

import json
import os
import pathlib
from collections.abc import Iterator, Mapping
from typing import Any

import numpy as np
import torch

from ..oake.objects import indices_min_wh
from ..packfile import ALIGN as _ALIGN, DTYPES as _DTYPES, MAGIC as _MAGIC, fields as _fields
from ..store import Store


class PthAccessLayer(Mapping):
    """``{data_root}/{task_name}/{key}.pth`` -> ``torch.load(..., 'cpu')``."""

    def __init__(self, data_root: str, task_name: str = '', **_: Any) -> None:
        self._dir = pathlib.Path(data_root) / task_name

    def __getitem__(self, key: str) -> Any:
        path = self._dir / f'{key}.pth'
        if not path.exists():
            raise KeyError(key)
        return torch.load(path, map_location='cpu')

    def __iter__(self) -> Iterator[str]:
        return (p.stem for p in sorted(self._dir.glob('*.pth')))

    def __len__(self) -> int:
        return sum(1 for _ in self._dir.glob('*.pth'))


def pack(data_root: str, task_name: str, out: str | None = None) -> pathlib.Path:
    """Gather ``{data_root}/{task_name}/*.pth`` into ``{data_root}/{task_name}.pack`` (+ ``.json``
    index: key -> [[field, dtype, shape, byte offset], ...]; tensors 64-byte aligned, C order).
    The index is written last, by rename, so a reader never sees a half-written pack."""
    src = PthAccessLayer(data_root, task_name)
    blob = pathlib.Path(out) if out is not None else pathlib.Path(data_root) / f'{task_name}.pack'
    index: dict[str, list] = {}
    offset = 0
    tmp = blob.with_name(blob.name + f'.tmp{os.getpid()}')
    with open(tmp, 'wb') as f:
        for key in src:
            entry = []
            for name, t in _fields(src[key]):
                if t.dtype not in _DTYPES:
                    raise TypeError(f'{key}: dtype {t.dtype} is not packable')
                pad = -offset % _ALIGN
                f.write(b'\0' * pad)
                offset += pad
                data = t.detach().contiguous().numpy().tobytes()
                f.write(data)
                entry.append([name, _DTYPES[t.dtype], list(t.shape), offset])
                offset += len(data)
            index[key] = entry
    os.replace(tmp, blob)
    meta = blob.with_name(blob.name + '.json')
    tmp = meta.with_name(meta.name + f'.tmp{os.getpid()}')
    tmp.write_text(json.dumps(dict(magic=_MAGIC, bytes=offset, index=index)))
    os.replace(tmp, meta)
    return blob


class PackAccessLayer(Mapping):
    """Read side of ``pack()``.  Tensors are read-only views of one shared memory map (opened lazily,
    so the object pickles cheaply into dataloader workers); ``copy=True`` hands out private copies
    for callers that write into what they load."""

    def __init__(self, data_root: str, task_name: str = '', copy: bool = False, **_: Any) -> None:
        # one blob (`pack()`, or a one-rank sweep with writer='pack') or the shards of a multi-rank sweep
        # (`<task_name>.r<rank>of<world>.pack`): the index maps a key to (shard, fields)
        root = pathlib.Path(data_root)
        blobs = [p for p in [root / f'{task_name}.pack'] if p.exists()] + sorted(root.glob(f'{task_name}.r*of*.pack'))
        if not blobs:
            raise FileNotFoundError(root / f'{task_name}.pack')
        # one sweep = one world size: shards left over from a sweep with a different rank count (or a stale single
        # blob beside fresh shards) would silently shadow or mix with the fresh features (first index entry wins)
        import re
        base = pathlib.PurePath(task_name).name
        shards = [(p, m) for p in blobs if (m := re.fullmatch(re.escape(base) + r'\.r(\d+)of(\d+)\.pack', p.name))]
        worlds = {int(m.group(2)) for _, m in shards}
        if len(worlds) > 1 or (shards and len(shards) < len(blobs)):
            raise ValueError(f'{root}: feature packs of more than one sweep for task {task_name!r} '
                             f'({", ".join(p.name for p in blobs)}): remove the stale ones')
        if worlds:
            w = next(iter(worlds))
            have = sorted(int(m.group(1)) for _, m in shards)
            if have != list(range(w)):
                import warnings
                warnings.warn(f'{root}: task {task_name!r} has the shards of ranks {have} of {w} - the keys of the '
                              f'missing ranks are absent', stacklevel=2)
        self._blobs = blobs
        self._index: dict[str, tuple[int, list]] = {}
        for i, blob in enumerate(blobs):
            meta = json.loads(blob.with_name(blob.name + '.json').read_text())
            if meta.get('magic') != _MAGIC:
                raise ValueError(f'{blob}: not an OAKE feature pack')
            if blob.stat().st_size < meta['bytes']:  # (longer is fine: a killed writer's tail behind the index)
                raise ValueError(f'{blob}: size does not match its index (truncated pack?)')
            for key, entry in meta['index'].items():
                self._index.setdefault(key, (i, entry))
        self._copy = copy
        self._maps: list[np.memmap | None] = [None] * len(blobs)

    def __getstate__(self) -> dict:
        return dict(self.__dict__, _maps=[None] * len(self._blobs))

    def _view(self, shard: int, dtype: str, shape: list[int], offset: int) -> torch.Tensor:
        if self._maps[shard] is None:
            blob = self._blobs[shard]
            self._maps[shard] = np.memmap(blob, dtype=np.uint8, mode='r') if blob.stat().st_size \
                else np.zeros(0, np.uint8)
        n = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        a = self._maps[shard][offset:offset + n].view(dtype).reshape(shape)
        if self._copy:
            return torch.from_numpy(np.array(a))
        import warnings
        with warnings.catch_warnings():  # read-only map: torch warns that writes are undefined
            warnings.simplefilter('ignore', UserWarning)
            return torch.from_numpy(a)

    def __getitem__(self, key: str) -> Any:
        shard, entry = self._index[key]
        if len(entry) == 1 and entry[0][0] == '':
            return self._view(shard, *entry[0][1:])
        return {name: self._view(shard, dtype, shape, offset) for name, dtype, shape, offset in entry}

    def __iter__(self) -> Iterator[str]:
        return iter(self._index)

    def __len__(self) -> int:
        return len(self._index)


ACCESS_LAYERS = dict(PthAccessLayer=PthAccessLayer, PackAccessLayer=PackAccessLayer)


def build_access_layer(config: Mapping[str, Any], default: Mapping[str, Any]) -> Mapping[str, Any]:
    """``ALR.build(config, default)``: ``default`` supplies the keys ``config`` leaves out."""
    cfg = {**default, **config}
    return ACCESS_LAYERS[cfg.pop('type')](**cfg)


def pairwise_intersection(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """[na, nb] intersection areas of xyxy boxes."""
    lt = torch.maximum(a[:, None, :2], b[None, :, :2])
    rb = torch.minimum(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp_min(0)
    return wh[..., 0] * wh[..., 1]


class LoadCLIPFeatures:
    """Pipeline step of oadp/dp/datasets.py:137-214.  ``results`` in/out keys as the reference:
    reads ``img_info.id``, ``bbox_fields``, optionally ``gt_bboxes`` / ``gt_labels``; writes
    ``clip_global``, ``clip_blocks`` / ``block_bboxes`` / ``block_labels``, ``clip_objects`` /
    ``object_bboxes``.  ``num_all`` is ``Globals.categories.num_all`` there (labels >= num_all are
    pseudo labels and do not mark blocks)."""

    def __init__(self, default: Mapping[str, Any], globals_: Mapping[str, Any] | None = None,
                 blocks: Mapping[str, Any] | None = None, objects: Mapping[str, Any] | None = None,
                 num_all: int = 65) -> None:
        if globals_ is None and blocks is None and objects is None:
            raise ValueError('at least one of globals_, blocks, objects is required')
        default = dict(default)
        if os.environ.get('TRAIN_WITH_VAL_DATASET', '').lower() in ('1', 'true', 'yes', 'on'):
            default['task_name'] = default['task_name'].replace('train', 'val')
        self._num_all = num_all
        self._globals, self._blocks, self._objects = (
            None if c is None else build_access_layer(c, default) for c in (globals_, blocks, objects))
        self._dry_key: str | None = None
        if Store.DRY_RUN:
            layers = [m for m in (self._globals, self._blocks, self._objects) if m is not None]
            self._dry_key = sorted(set.intersection(*(set(m.keys()) for m in layers)))[0]

    def __call__(self, results: dict[str, Any]) -> dict[str, Any]:
        key = self._dry_key if self._dry_key is not None else f'{results["img_info"]["id"]:012d}'
        bbox_fields: list[str] = results['bbox_fields']

        if self._globals is not None:
            results['clip_global'] = self._globals[key].squeeze(0)

        if self._blocks is not None:
            blocks = self._blocks[key]
            block_bboxes = blocks['bboxes']
            if 'gt_bboxes' in results:
                gt_bboxes = np.asarray(results['gt_bboxes'])
                gt_labels = np.asarray(results['gt_labels'])
                real = gt_labels < self._num_all
                gt_bboxes, gt_labels = gt_bboxes[real], gt_labels[real]
                # f16 block boxes against f32 ground truth: torch promotes to f32, as here
                overlap = pairwise_intersection(
                    block_bboxes.float(), torch.as_tensor(gt_bboxes, dtype=torch.float32).reshape(-1, 4)) > 0
                block_ids, gt_ids = torch.where(overlap)
                block_labels = np.zeros((block_bboxes.shape[0], self._num_all), dtype=bool)
                block_labels[block_ids.numpy(), gt_labels[gt_ids.numpy()]] = True
                results['block_labels'] = block_labels
            results['clip_blocks'] = blocks['embeddings']
            results['block_bboxes'] = block_bboxes.float().numpy()
            bbox_fields.append('block_bboxes')

        if self._objects is not None:
            objects = self._objects[key]
            object_bboxes = objects['bboxes']
            keep = indices_min_wh(object_bboxes, (4, 4))
            results['clip_objects'] = objects['embeddings'][keep]
            results['object_bboxes'] = object_bboxes[keep].float().numpy()
            bbox_fields.append('object_bboxes')

        return results


def main() -> None:
    import argparse
    p = argparse.ArgumentParser(description='pack per-image OAKE feature files into one mappable blob')
    p.add_argument('data_root')
    p.add_argument('task_name')
    a = p.parse_args()
    blob = pack(a.data_root, a.task_name)
    print(f'{blob}: {blob.stat().st_size} bytes, {len(PackAccessLayer(a.data_root, a.task_name))} keys')


if __name__ == '__main__':
    main()
