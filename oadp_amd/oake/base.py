"""Shared dataset / run loop of the three OAKE modes.

Reference: oadp/oake/base.py (BaseDataset :28-63, parse_args :66-72, BaseValidator :75-152).
``torchvision.datasets.CocoDetection`` and ``todd.utils.Validator`` are not dependencies: the COCO
annotation JSON is read directly (same ``ids = sorted(image ids)`` order and RGB loading), and the
run loop is ours: iterate the sampler-sharded dataset, skip ``None`` (already extracted), gather
crops of many images into one encoder batch, scatter per-image results to ``.pth`` files.
"""
from __future__ import annotations

import argparse
import json
import collections
import os
import pathlib
import queue
import threading
import time
from abc import ABC, abstractmethod
from typing import Any, Generic, Iterable, Protocol, TypeVar

import PIL.Image
import torch
import torch.distributed
import torch.utils.data
import torch.utils.data.distributed

from ..config import Config, parse_override
from ..store import Store, get_local_rank, get_rank, get_world_size, parse_shard, pin_cpus, shard_device_index
from . import fastsave
from ..packfile import PackWriter, blob_path


class Batch(Protocol):

    @property
    def output(self) -> pathlib.Path:
        ...


T = TypeVar('T', bound=Batch)


class CocoImages:
    """The slice of pycocotools' COCO + torchvision CocoDetection that OAKE uses."""

    def __init__(self, root: str, annFile: str) -> None:
        self.root = root
        with open(annFile) as f:
            data = json.load(f)
        self.imgs = {img['id']: img for img in data['images']}  # annotation-file order
        self.ids = list(sorted(self.imgs.keys()))

    def loadImgs(self, ids: Iterable[int]) -> list[dict]:
        return [self.imgs[i] for i in ids]


class EncodedImage:
    """Stands in for the PIL image on the device-decode path: the host only reads the file and its
    header; Huffman decoding (native threads), IDCT, upsampling and colour conversion happen in the
    process that owns the GPU (``BaseValidator._images_u8`` -> ``oake_decode_jpeg_batch``)."""

    def __init__(self, data: bytes, size: tuple[int, int]) -> None:
        self.data = data
        self.size = size  # (width, height), as PIL.Image.size


def jpeg_size(data: bytes) -> tuple[int, int] | None:
    """(width, height) if the device decoder covers this file, else None (host only, no GPU)."""
    import ctypes as C

    from .. import _lib
    lib = _lib.load()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    h, w = C.c_int(0), C.c_int(0)
    if lib.oake_jpeg_info(buf, len(data), C.byref(h), C.byref(w), None) != _lib.OAKE_OK:
        return None
    return w.value, h.value


class BaseDataset(torch.utils.data.Dataset, ABC, Generic[T]):
    _device_preprocess = False
    _device_decode: bool | str = False

    def __init__(self, root: str, annFile: str, *, auto_fix: bool = False, output_dir: str,
                 transform=None, device_preprocess: bool = False,
                 device_decode: bool | str = False, **kwargs) -> None:
        if kwargs:  # a misspelled option must not vanish silently
            raise TypeError(f'{type(self).__name__}: unknown dataset option(s) {sorted(kwargs)}')
        self.coco = CocoImages(root, annFile)
        self.root = root
        self.ids = self.coco.ids
        self.transform = transform
        self._auto_fix = auto_fix
        # True: workers only decode; crop / antialiased-bicubic resize / normalise run on the GPU
        # (csrc/resample.hip, bit-exact with the PIL path) — needs a HIP device in the main process
        self._device_preprocess = device_preprocess or bool(device_decode)
        if self._device_preprocess:
            from ..clip.preprocess import check_pillow_version
            check_pillow_version()
        # True: the workers only read baseline JPEG files; they are decoded by the process that owns
        # the GPU (csrc/jpeg.hip: Huffman passes on native threads, the rest on the device,
        # bit-identical to PIL); files outside that subset (progressive, CMYK, PNG, ...) take the
        # reference's own PIL decode in the worker.  'strict': raise for those instead.
        self._device_decode = device_decode
        self._output_dir = pathlib.Path(output_dir)
        self._output_dir.mkdir(parents=True, exist_ok=True)
        # keys ("<image_id:012d>") a direct pack writer already holds (BaseValidator(writer='pack')): skipped like
        # images whose .pth file exists
        self._done: set[str] = set()

    @property
    def transforms(self):
        # torchvision's StandardTransform: ``self.transforms.transform(image)``
        class _T:
            transform = staticmethod(self.transform)
        return _T

    def __len__(self) -> int:
        return len(self.ids)

    def _image_path(self, id_: int) -> str:
        # torchvision CocoDetection._load_image: <root>/<file_name>
        return os.path.join(self.root, self.coco.loadImgs([id_])[0]['file_name'])

    def _load_image(self, id_: int) -> PIL.Image.Image | EncodedImage:
        path = self._image_path(id_)
        if self._device_decode:
            data = pathlib.Path(path).read_bytes()
            size = jpeg_size(data)
            if size is not None:
                return EncodedImage(data, size)
            if self._device_decode == 'strict':
                raise ValueError(f'{path}: not a baseline JPEG the device decoder supports')
        return PIL.Image.open(path).convert('RGB')

    def __getitem__(self, index: int) -> T | None:
        # reference oadp/oake/base.py:42-54 (resume by skipping; auto_fix re-verifies the file)
        id_ = self.ids[index]
        output = self._output_dir / f'{id_:012d}.pth'
        if output.stem in self._done:
            return None
        if output.exists():
            if not self._auto_fix:
                return None
            try:
                torch.load(output, 'cpu')
                return None
            except Exception:
                print(f'Fixing {output}', flush=True)
        image = self._load_image(id_)
        return self._preprocess(id_, output, image)

    @abstractmethod
    def _preprocess(self, id_: int, output: pathlib.Path, image: PIL.Image.Image) -> T:
        pass


def image_to_u8(image: PIL.Image.Image | EncodedImage) -> torch.Tensor:
    """RGB PIL image -> uint8 HWC tensor (what the device preprocessing kernels consume); an
    ``EncodedImage`` -> its file bytes as a 1-D uint8 tensor (decoded by ``BaseValidator._images_u8``)."""
    import numpy as np
    if isinstance(image, EncodedImage):
        return torch.frombuffer(bytearray(image.data), dtype=torch.uint8)
    return torch.from_numpy(np.asarray(image.convert('RGB'), dtype=np.uint8).copy())


def parse_args(argv: list[str] | None = None) -> argparse.Namespace:
    parser = argparse.ArgumentParser(description='OAKE feature extraction')
    parser.add_argument('name', type=str)
    parser.add_argument('config', type=Config.load)
    parser.add_argument('--override', nargs='*')
    return parser.parse_args(argv)


def atomic_save(obj: Any, path: pathlib.Path) -> None:
    """torch.save via tmp + rename: same visible contract as oadp/oake/base.py:112, but a killed
    run cannot leave a truncated file (what ``auto_fix`` exists to repair)."""
    tmp = path.with_name(path.name + f'.tmp{os.getpid()}')
    data = fastsave.SAVER.dumps(obj)  # the archive torch.save would write (fastsave.py), or None
    if data is None:
        torch.save(obj, tmp)
    else:
        with open(tmp, 'wb') as f:
            f.write(data)
    os.replace(tmp, path)


class _HostCopy:
    """A device -> host copy in flight: ``get()`` waits for it and returns the host tensor.  The pinned
    staging buffer belongs to THIS copy until then (``_PinnedPool``): however many copies a subclass starts
    per flush, and however deep the look-ahead, none can overwrite a result that has not been taken yet."""

    def __init__(self, tensor: torch.Tensor, event: 'torch.cuda.Event | None' = None,
                 pool: '_PinnedPool | None' = None, slot: int = -1) -> None:
        self._tensor, self._event, self._pool, self._slot = tensor, event, pool, slot

    def get(self) -> torch.Tensor:
        if self._event is not None:
            self._event.synchronize()
            self._event = None
        if self._pool is not None:  # private copy out of the pinned slot, which goes back to the pool
            self._tensor = self._tensor.clone()
            self._pool.release(self._slot)
            self._pool = None
        return self._tensor


class _PinnedPool:
    """Pinned host buffers for asynchronous copies in either direction; a buffer is handed out to one copy
    at a time and returns when that copy's result has been taken (device -> host: ``release``) or when the
    copy has run (host -> device: ``release_after`` an event); a new one is allocated if all are in flight."""

    def __init__(self) -> None:
        self._bufs: list[torch.Tensor] = []
        self._busy: list[bool] = []
        self._events: dict[int, Any] = {}

    @staticmethod
    def _alloc(nbytes: int) -> torch.Tensor:
        return torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)

    def release_after(self, slot: int, event: Any) -> None:
        self._events[slot] = event

    def acquire(self, numel: int, dtype: torch.dtype) -> tuple[int, torch.Tensor]:
        """(slot, pinned 1-D tensor of >= numel elements of dtype).  Buffers are untyped bytes with
        power-of-two capacities, so requests of different types and slowly varying sizes (images, masks,
        embeddings of successive flushes) reuse them instead of re-pinning memory."""
        for slot in [k for k, ev in self._events.items() if ev.query()]:
            del self._events[slot]
            self._busy[slot] = False
        item = torch.empty(0, dtype=dtype).element_size()
        need = max(numel, 1) * item
        free = [i for i, busy in enumerate(self._busy) if not busy]
        fit = [i for i in free if self._bufs[i].numel() >= need]
        if fit:
            i = min(fit, key=lambda k: self._bufs[k].numel())
        else:
            cap = 1 << max(need - 1, 4095).bit_length()
            buf = self._alloc(cap)
            if free:  # too small: replace the largest free one
                i = max(free, key=lambda k: self._bufs[k].numel())
                self._bufs[i] = buf
            else:
                self._bufs.append(buf)
                self._busy.append(False)
                i = len(self._bufs) - 1
        self._busy[i] = True
        return i, self._bufs[i][:need].view(dtype)

    def release(self, slot: int) -> None:
        self._busy[slot] = False


class AsyncWriter:
    """Background ``atomic_save``: the reference writes one small file per
    image synchronously on the thread that also drives the GPU (oadp/oake/base.py:112); at 10^4-10^5
    images/s per GPU that serialises the whole sweep.  ``threads`` workers take (object, path)
    pairs from a bounded queue; ``drain()`` blocks until everything is on disk and re-raises the first
    worker error.  ``threads=0`` degenerates to the synchronous behaviour."""

    def __init__(self, threads: int = 4, depth: int = 4096) -> None:
        self.bytes = 0
        self._threads: list[threading.Thread] = []
        self._error: BaseException | None = None
        self._lock = threading.Lock()
        self._queue: queue.Queue | None = queue.Queue(maxsize=depth) if threads > 0 else None
        for i in range(threads):
            t = threading.Thread(target=self._work, name=f'oake-writer-{i}', daemon=True)
            t.start()
            self._threads.append(t)

    def _save(self, obj: Any, path: pathlib.Path) -> None:
        atomic_save(obj, path)
        size = path.stat().st_size
        with self._lock:
            self.bytes += size

    def _work(self) -> None:
        assert self._queue is not None
        while True:
            item = self._queue.get()
            try:
                if item is None:
                    return
                if self._error is None:
                    self._save(*item)
            except BaseException as e:  # noqa: BLE001 — surfaced by drain()
                with self._lock:
                    self._error = self._error or e
            finally:
                self._queue.task_done()

    def submit(self, obj: Any, path: pathlib.Path) -> None:
        if self._error is not None:
            self.drain()
        if self._queue is None:
            self._save(obj, path)
        else:
            self._queue.put((obj, path))

    def drain(self) -> None:
        if self._queue is not None:
            self._queue.join()
        if self._error is not None:
            e, self._error = self._error, None
            raise e

    def close(self) -> None:
        try:
            self.drain()
        finally:
            if self._queue is not None:
                for _ in self._threads:
                    self._queue.put(None)
                for t in self._threads:
                    t.join()
                self._threads.clear()


class Counters:
    """[images, crops, seconds, bytes] — the only thing exchanged between ranks."""

    def __init__(self) -> None:
        self.images = 0
        self.crops = 0
        self.seconds = 0.0
        self.bytes = 0

    def tensor(self, device) -> torch.Tensor:
        return torch.tensor([self.images, self.crops, self.seconds, self.bytes],
                            dtype=torch.float64, device=device)


def gather_counters(counters: Counters, device) -> list[list[float]]:
    """Rank-0 throughput report: one all_gather of 4 x f64 per rank (RCCL on GPU, gloo on CPU)."""
    if (torch.distributed.is_available() and torch.distributed.is_initialized()
            and torch.distributed.get_backend() == 'gloo'):
        device = 'cpu'
    t = counters.tensor(device)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        out = [torch.zeros_like(t) for _ in range(torch.distributed.get_world_size())]
        torch.distributed.all_gather(out, t)
        return [o.tolist() for o in out]
    return [t.tolist()]


def pick_backend(cuda: bool, gpus_here: int, env=None) -> str:
    """Collective backend for the one counters gather.  RCCL ('nccl') needs a GPU per rank ON THIS NODE:
    torchrun's LOCAL_WORLD_SIZE is the ranks per node (a multi-node run has more ranks in total than any node
    has GPUs, and must still get RCCL); with more ranks than GPUs on a node — several processes per GPU: the host
    side of the sweep (file reads, Huffman decode, .pth writing) scales with processes, docs/history/design_sections_5_6_as_of_round5.md §5.5 — the
    gather goes over gloo.  OAKE_DIST_BACKEND overrides."""
    env = os.environ if env is None else env
    if env.get('OAKE_DIST_BACKEND'):
        return env['OAKE_DIST_BACKEND']
    per_node = int(env.get('LOCAL_WORLD_SIZE') or env.get('WORLD_SIZE') or 1)
    return 'nccl' if cuda and per_node <= gpus_here else 'gloo'


class BaseValidator(ABC, Generic[T]):

    def __init__(self, name: str, model, *, dataloader: Config, log: Config | None = None,
                 batch_size: int = 256, device: torch.device | str | None = None,
                 writer_threads: int = 4, decode_threads: int = 16, prefetch: int = 512,
                 streams: int = 2, host_threads: int = 8, writer: str = 'pth', lookahead: int = 1,
                 **kwargs) -> None:
        if kwargs:  # a misspelled option must not vanish silently
            raise TypeError(f'{type(self).__name__}: unknown option(s) {sorted(kwargs)}')
        self.name = name
        self._model = model
        self._log_interval = (log or {}).get('interval', 50)
        self._batch_size = batch_size
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device()) if Store.CUDA else 'cpu'
        self._device = torch.device(device)
        self.counters = Counters()
        self._writer_threads = writer_threads
        # 'pth' (default): one <image_id:012d>.pth per image, the reference's contract [REF oadp/oake/base.py:44,112].
        # 'pack': the same payloads appended to ONE memory-mappable blob per split (and rank) + a JSON index, read by
        # oadp_amd.dp.PackAccessLayer (oadp_amd/packfile.py) — no per-image files at all
        if writer not in ('pth', 'pack'):
            raise ValueError(f"writer must be 'pth' or 'pack', got {writer!r}")
        self._writer_kind = writer
        self._decode_threads = decode_threads
        self._prefetch = prefetch
        # torch's intra-op pool for the sweep: the host side of a flush is hundreds of tiny tensor ops per
        # image (.half(), clone, cat of a few KB), and with the default — one OpenMP thread per core — each
        # of them wakes the whole pool: on a 256-thread host the blocks sweep ran at 880 images/s against
        # 3 250 with the pool capped (tools/sweep_ranks.py; torchrun sets OMP_NUM_THREADS=1 for the same
        # reason).  0 leaves torch's setting alone.
        self._host_threads = host_threads
        self._writer: AsyncWriter | None = None
        # flushes whose results are still on the GPU, oldest first: `lookahead` of them stay in flight while the host
        # prepares the next (OAKE_LOOKAHEAD overrides)
        self._inflight: collections.deque[tuple[list, Any]] = collections.deque()
        self._lookahead = max(1, int(os.environ.get('OAKE_LOOKAHEAD', lookahead)))
        self._host_pool = _PinnedPool()
        # consecutive flushes alternate over `streams` lanes = (native handle, HIP stream) pairs: the
        # kernels of two independent batches fill each other's start-up and tail (GPU only)
        self._n_lanes = max(1, int(streams)) if self._device.type == 'cuda' and hasattr(
            getattr(model, 'visual', None), 'lane') else 1
        self._lane_streams: list | None = None
        self._flush_no = 0
        self._dataloader = self._build_dataloader(Config(dataloader))

    # -- reference surface ----------------------------------------------------------------------
    def _build_dataloader(self, config: Config) -> torch.utils.data.DataLoader:
        # reference oadp/oake/base.py:78-89
        config = Config(config)
        if Store.DRY_RUN:
            config['num_workers'] = 0
        if config.get('num_workers', 0) > 0 and getattr(config.get('dataset'), '_device_decode', False):
            # device decode hands over file bytes: through DataLoader workers every sample crosses a process
            # boundary first (measured 117 vs 3 135 images/s in blocks mode, profiles/r02_sweep_1gpu.log); the
            # prefetch thread of `_items` reads the files instead
            if get_rank() == 0:
                print(f'[{self.name}] device_decode: num_workers {config["num_workers"]} -> 0 '
                      f'(files are read by a prefetch thread of this process)', flush=True)
            config['num_workers'] = 0
        world, rank = get_world_size(), get_rank()
        # OAKE_SHARD=r/W: the DistributedSampler shard r of W without a process group — array-job style
        # launches (one independent process per GPU or per node, nothing to rendezvous: the path has no
        # data-path collective), and measuring one rank's share of a W-rank sweep on a single GPU
        shard = parse_shard()
        if shard is not None:
            if world > 1:
                raise RuntimeError('OAKE_SHARD and a torch.distributed launch (WORLD_SIZE > 1) are exclusive')
            rank, world = shard
        if world > 1:
            config['sampler'] = torch.utils.data.distributed.DistributedSampler(
                config['dataset'], num_replicas=world, rank=rank, shuffle=False)
        return torch.utils.data.DataLoader(batch_size=None, **config)

    @classmethod
    @abstractmethod
    def _build_model(cls):
        pass

    @abstractmethod
    def _encode(self, batches: list[T]) -> list[Any]:
        """Encode the crops of several images in one pass; one result object per image."""

    def _n_crops(self, batch: T) -> int:
        return 1

    def _images_u8(self, ts: list[torch.Tensor]) -> list[torch.Tensor]:
        """What ``image_to_u8`` produced for every image of a flush -> uint8 HWC images on the device;
        JPEG file bytes (1-D tensors) are decoded there, all of them in one native call."""
        out: list[torch.Tensor | None] = [None] * len(ts)
        enc = [i for i, t in enumerate(ts) if t.dim() == 1]
        if enc:
            decoded = self._model.visual.decode_jpeg_batch([ts[i] for i in enc], self._device,
                                                           threads=self._decode_threads)
            for i, img in zip(enc, decoded):
                if img is None:  # entropy segment the device decoder rejects: PIL has the final word
                    import io
                    img = image_to_u8(PIL.Image.open(io.BytesIO(ts[i].numpy().tobytes()))).to(self._device)
                out[i] = img
        raw = [i for i in range(len(ts)) if out[i] is None]
        if raw and self._device.type == 'cuda':
            # all raw uint8 images of the flush in ONE copy through a pinned slot (each at a 256-byte
            # boundary), instead of one stream-ordered copy from pageable memory per image
            offs, total = [], 0
            for i in raw:
                offs.append(total)
                total += -(-ts[i].numel() // 256) * 256
            slot, buf = self._host_pool.acquire(total, torch.uint8)
            for i, o in zip(raw, offs):
                buf[o:o + ts[i].numel()].view(ts[i].shape).copy_(ts[i])
            dev = buf[:total].to(self._device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._host_pool.release_after(slot, ev)
            for i, o in zip(raw, offs):
                out[i] = dev[o:o + ts[i].numel()].view(ts[i].shape)
        else:
            for i in raw:
                out[i] = ts[i].to(self._device)
        return out

    def _to_host(self, t: torch.Tensor) -> '_HostCopy':
        """Start the device -> host copy of an encoder output without waiting for it (``_encode`` may
        return a closure that calls ``.get()``: see ``_flush``).  The pinned staging buffer is owned by the
        returned copy until ``.get()`` (``_PinnedPool``)."""
        if not t.is_cuda:
            return _HostCopy(t)
        n = t.numel()
        slot, buf = self._host_pool.acquire(n, t.dtype)
        dst = buf[:n].view(t.shape)
        dst.copy_(t, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return _HostCopy(dst, ev, self._host_pool, slot)

    def _to_device(self, t: torch.Tensor) -> torch.Tensor:
        """Host tensor -> device without blocking the host: a copy from pageable memory waits for the
        stream it is queued on (here: for the crops of this flush and whatever that lane still runs), so
        the tensor goes through a pinned slot that is taken back when the copy has run."""
        if self._device.type != 'cuda' or t.is_cuda or t.numel() == 0:
            return t.to(self._device)
        n = t.numel()
        slot, buf = self._host_pool.acquire(n, t.dtype)
        src = buf[:n].view(t.shape)
        src.copy_(t)
        out = src.to(self._device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._host_pool.release_after(slot, ev)
        return out

    def _submit(self, batches: list[T], results: list) -> None:
        assert self._writer is not None
        for batch, result in zip(batches, results):
            self._writer.submit(result, batch.output)
            self.counters.images += 1

    def _finish_inflight(self) -> None:
        while self._inflight:
            batches, finish = self._inflight.popleft()
            self._submit(batches, finish())

    def _flush(self, pending: list[T]) -> None:
        """Encode a flush and hand its results to the writer.  ``_encode`` returns the results, or — on
        the GPU — a closure that produces them once the device has finished: then the flush stays in
        flight while the host decodes and cuts the crops of the next one, and is written out when that
        one has been launched (`lookahead` flushes stay in flight; file order is unchanged)."""
        if not pending:
            return
        if self._n_lanes > 1:
            if self._lane_streams is None:
                self._lane_streams = [torch.cuda.Stream(self._device) for _ in range(self._n_lanes)]
            lane = self._flush_no % self._n_lanes
            self._flush_no += 1
            visual = self._model.visual
            visual.lane = lane
            try:
                with torch.cuda.stream(self._lane_streams[lane]):
                    results = self._encode(pending)
            finally:
                visual.lane = 0
        else:
            results = self._encode(pending)
        if callable(results):
            self._inflight.append((list(pending), results))
            while len(self._inflight) > self._lookahead:
                batches, finish = self._inflight.popleft()
                self._submit(batches, finish())
        else:
            self._finish_inflight()
            self._submit(pending, results)
        pending.clear()

    def run(self) -> Counters:
        t0 = time.perf_counter()
        pending: list[T] = []
        crops = 0
        torch_threads = torch.get_num_threads()
        if 0 < self._host_threads < torch_threads:  # (before the worker threads run their first tensor op)
            torch.set_num_threads(self._host_threads)
        if self._writer_kind == 'pack':
            ds = self._dataloader.dataset
            shard = parse_shard() or (get_rank(), get_world_size())
            self._writer = PackWriter(blob_path(ds._output_dir, *shard))
            ds._done = self._writer.keys  # resume: what the blob already holds
        else:
            self._writer = AsyncWriter(self._writer_threads)
        try:
            self._run_loop(pending, crops)
        finally:
            # (a run that raised leaves flushes in flight: their results belong to THIS run's writer and dataset and
            # must not reach the next run()'s — val then train share the validator class, not the instance, but a
            # caller may well re-run an instance)
            self._inflight.clear()
            writer, self._writer = self._writer, None
            try:
                writer.close()  # every file of this split is on disk (or the error is raised) here
            finally:
                self.counters.bytes += writer.bytes
                if torch.get_num_threads() != torch_threads:
                    torch.set_num_threads(torch_threads)
        if self._device.type == 'cuda':
            torch.cuda.synchronize(self._device)
        self.counters.seconds += time.perf_counter() - t0
        return self.counters

    def _items(self):
        """The dataloader's items, produced one flush ahead by a background thread when there are no
        DataLoader workers: file reads and the per-image index math then overlap the encoder pass the
        main thread is waiting on (both release the GIL for most of their time)."""
        if getattr(self._dataloader, 'num_workers', 0) > 0 or self._prefetch <= 0:
            yield from self._dataloader
            return
        q: queue.Queue = queue.Queue(maxsize=self._prefetch)
        stop = threading.Event()
        end = object()

        def produce() -> None:
            try:
                for item in self._dataloader:
                    while not stop.is_set():
                        try:
                            q.put(item, timeout=0.1)
                            break
                        except queue.Full:
                            continue
                    if stop.is_set():
                        return
                q.put(end)
            except BaseException as e:  # noqa: BLE001 — re-raised in the consumer
                q.put(e)

        t = threading.Thread(target=produce, name='oake-prefetch', daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is end:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            stop.set()
            t.join(timeout=5)

    def _run_loop(self, pending: list[T], crops: int) -> None:
        for i, batch in enumerate(self._items()):
            if batch is None:  # reference _control_run_iter: CONTINUE on None (base.py:96-104)
                continue
            pending.append(batch)
            crops += self._n_crops(batch)
            if crops >= self._batch_size:
                self.counters.crops += crops
                self._flush(pending)
                crops = 0
            if (i + 1) % self._log_interval == 0 and get_rank() == 0:
                print(f'[{self.name}] iter {i + 1}/{len(self._dataloader)} '
                      f'images {self.counters.images} crops {self.counters.crops}', flush=True)
        self.counters.crops += crops
        self._flush(pending)
        self._finish_inflight()

    @classmethod
    def main(cls, argv: list[str] | None = None) -> None:
        # reference oadp/oake/base.py:115-152
        args = parse_args(argv)
        config: Config = args.config
        override = parse_override(args.override)
        if override is not None:
            config.override(override)

        # (OAKE_FORCE_DIST=1: a process group even at world size 1 — the RCCL calls of the multi-rank path executed on a
        # 1-GPU box, tests/test_rccl_n1_gpu.py; needs the launcher's RANK / WORLD_SIZE / MASTER_* variables)
        distributed = get_world_size() > 1 or os.environ.get('OAKE_FORCE_DIST', '') not in ('', '0')
        pin_cpus()  # a rank of a multi-rank node keeps its share of the host's cores (OAKE_CPU_AFFINITY=0: off)
        if Store.CUDA:
            # LOCAL_RANK under a launcher; shard r of OAKE_SHARD=r/W without one
            torch.cuda.set_device(shard_device_index(torch.cuda.device_count()))
        if distributed:
            backend = pick_backend(Store.CUDA, torch.cuda.device_count() if Store.CUDA else 0)
            torch.distributed.init_process_group(backend=backend)

        # the unpinned behaviours of the un-vendored fork (oadp_amd/clip/settings.py), before the model exists
        from ..clip import settings as fork_settings
        fork_settings.configure(**config.pop('fork', {}))

        model, preprocess = cls._build_model()

        train = config.pop('train')
        val = config.pop('val')
        train.dataloader.dataset.transform = preprocess
        val.dataloader.dataset.transform = preprocess

        totals = []
        for split in (val, train):  # val first, then train — as the reference
            validator = cls(args.name, model, **split, **config)
            validator.run()
            totals.append(gather_counters(validator.counters, validator._device))
        if get_rank() == 0:
            for split_name, per_rank in zip(('val', 'train'), totals):
                images = sum(r[0] for r in per_rank)
                crops = sum(r[1] for r in per_rank)
                secs = max(r[2] for r in per_rank)
                shard = parse_shard()
                who = f'shard {shard[0]} of {shard[1]} (OAKE_SHARD)' if shard else f'{len(per_rank)} rank(s)'
                print(f'[{args.name}] {split_name}: {int(images)} images, {int(crops)} crops, '
                      f'{secs:.1f} s, {images / max(secs, 1e-9):.1f} images/s over {who}', flush=True)
        if distributed:
            torch.distributed.destroy_process_group()
