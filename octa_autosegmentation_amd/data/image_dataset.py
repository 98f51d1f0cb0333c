"""Datasets and loaders of the entry points (reference data/image_dataset.py:19-81), without MONAI / torch DataLoader
workers: `get_dataset(config, phase)` builds the file lists exactly as the reference does (recursive glob, natural sort,
optional split files, the shorter lists repeated to the longest) and returns a DeviceLoader whose samples are transformed ON
THE GPU (data/data_transforms.py) and collated into device-resident mini-batches -- the reference's loader workers
spend 0.6 s per sample on the CPU (CSV parse + two matplotlib renders), which would cap training at a few images per
second whatever the convolutions do."""
import os
import queue
import re
import threading
from glob import glob
from math import ceil

import numpy as np
import torch

from ..utils.enums import Phase, Task
from .data_transforms import Compose, get_data_augmentations
from .unalignedZipDataset import UnalignedZipDataset


def natsorted(paths):
    """Natural sort (the reference uses natsort.natsorted): digit runs compare as integers."""
    return sorted(paths, key=lambda s: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", s)])


def _get_transformation(config, phase: str, dtype=torch.float32) -> Compose:
    return Compose(get_data_augmentations(config[phase]["data_augmentation"], config["General"].get("seed", 42), dtype))


def get_post_transformation(config: dict, phase: str) -> dict:
    """`post_processing` of a phase -> {output name: Compose}, applied to predictions / labels before metrics and files."""
    post = dict()
    for k, v in config[phase]["post_processing"].items():
        try:
            post[k] = Compose(get_data_augmentations(v, seed=config["General"].get("seed", 42)))
        except Exception as e:
            print("Error: Your provided data augmentations for prediction are invalid.\n")
            raise e
    return post


class ListDataset:
    def __init__(self, items, transform):
        self.items, self.transform = items, transform

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.transform(self.items[i])

    def get_batch(self, indices):
        """Samples `indices`, transformed; chains with a mini-batch form (the frozen generator of configs/config_ves_seg-S_GAN.yml)
        run it once per mini-batch (data_transforms.Compose.call_batch) -- same random decisions as sample-by-sample."""
        items = [self.items[j] for j in indices]
        t = self.transform
        has_bg = all("background" in it for it in items)
        if len(items) > 1 and hasattr(t, "batchable") and t.batchable(has_background=has_bg):
            return t.call_batch(items)
        return [t(it) for it in items]


class FusedGraphSegBatches:
    """Batch-level form of the segmentation configs' training chain (configs/config_ves_seg-S.yml:28-102) for the common case --
    graph loader without dropout, then ScaleIntensityd, EnsureChannelFirstd, Resized (bilinear), RandFlipd (both axes), RandRotate90d,
    RandRotated (zeros padding), AsDiscreted on the label, CastToTyped: a whole mini-batch is rendered by two rasteriser launch
    sequences (image and label resolution) and augmented by two streaming kernels per tensor (data/gpu_augment.py) instead of a
    dozen small torch ops per sample and key. Takes the decisions of the generic per-sample transforms for the same seed."""

    NAMES = ["LoadGraphAndFilterByRandomRadiusd", "ScaleIntensityd", "EnsureChannelFirstd", "Resized", "RandFlipd", "RandRotate90d",
             "RandRotated", "AsDiscreted", "CastToTyped"]

    @staticmethod
    def applies(aug_config):
        if [d.get("name") for d in aug_config] != FusedGraphSegBatches.NAMES or not torch.cuda.is_available():
            return False
        load, scale, _, resize, flip, rot90, rot, disc, cast = aug_config
        both = lambda d: list(d.get("keys", [])) == ["image", "label"]
        size = resize.get("spatial_size", [0, 1])
        return (both(load) and both(scale) and both(resize) and both(flip) and both(rot90) and both(rot) and both(cast)
                and list(disc.get("keys", [])) == ["label"] and load.get("max_dropout_prob", 0) == 0 and resize.get("mode") == "bilinear"
                and size[0] == size[1] and sorted(flip.get("spatial_axis", [])) == [0, 1] and rot.get("padding_mode") == "zeros"
                and not rot.get("range_y") and not rot.get("range_z") and cast.get("dtype") == "dtype" and rot90.get("max_k", 3) == 3)

    def __init__(self, aug_config, seed, dtype):
        from .gpu_augment import GpuSegAugmentation
        self.load = aug_config[0]
        self.aug = GpuSegAugmentation(aug_config, seed=seed)
        self.dtype = dtype
        # Without dropout the raster of a graph file at a given resolution / radius window is the same in every epoch: it is rendered
        # once and kept in HBM (uint8: 0.09 MB per 304^2 image + 1.48 MB per 1216^2 label = 0.8 GB for the reference's 500 training
        # pairs, of 288 GB) -- from the second epoch on a mini-batch costs the augmentation kernels only. OCTA_RASTER_CACHE_GB bounds
        # it (default 16; 0 disables). The reference re-renders every sample with matplotlib in every epoch (0.6 s per sample).
        self.cache = {}
        self.cache_bytes = 0
        self.cache_cap = int(float(os.environ.get("OCTA_RASTER_CACHE_GB", "16")) * (1 << 30))

    def _render(self, items, i, key):
        """uint8 [B, h, w]: rasters of the mini-batch's graphs for key i, from the cache or from ONE launch sequence over the misses."""
        from ..vessel_graph_generation import tree2img
        from .data_transforms import load_graph_cached
        res, lo = self.load["image_resolutions"][i], float(self.load["min_radius"][i])
        graphs = [load_graph_cached(it[key]) for it in items]
        keys = [(it[key], len(e), int(res[0]), int(res[1]), lo, int(self.load.get("MIP_axis", 2))) for it, (e, _) in zip(items, graphs)]
        miss = [j for j, k in enumerate(keys) if k not in self.cache]
        fresh = {}
        if miss:
            off = np.concatenate(([0], np.cumsum([len(graphs[j][0]) for j in miss]))).astype(np.int64)
            d_edges = torch.cat([graphs[j][1] for j in miss], dim=0) if len(miss) > 1 else graphs[miss[0]][1]
            imgs = tree2img.rasterize_edges_device(d_edges, off, res, self.load.get("MIP_axis", 2), min_radius=lo, max_radius=1.0)
            for n, j in enumerate(miss):
                fresh[j] = imgs[n]
                nbytes = imgs[n].numel()
                if self.cache_bytes + nbytes <= self.cache_cap:
                    self.cache[keys[j]] = imgs[n].clone()      # own storage: the batch tensor is freed with the mini-batch
                    self.cache_bytes += nbytes
            if len(miss) == len(items):
                return imgs
        return torch.stack([fresh[j] if j in fresh else self.cache[k] for j, k in enumerate(keys)])

    def __call__(self, items):
        """items: list of {image: csv path, label: csv path, *_path}. -> collated batch dict on the device."""
        from .data_transforms import advance_python_random, load_graph_cached
        out = {k: [it[k] for it in items] for k in items[0] if k.endswith("_path")}
        rendered = {key: self._render(items, i, key) for i, key in enumerate(("image", "label"))}
        draws = 0                          # random() draws of tree2img.py:62,78 over the mini-batch: per sample one for the dropout probability
        for it in items:                   # (first key) and one per in-range edge of either key; only their total moves the stream
            for i, key in enumerate(("image", "label")):
                e, _ = load_graph_cached(it[key])
                lo = float(self.load["min_radius"][i])
                draws += int(np.count_nonzero((e[:, 6] >= lo) & (e[:, 6] <= 1.0))) + (1 if i == 0 else 0)
        advance_python_random(draws)
        mb = self.aug(rendered["image"], rendered["label"])
        out["image"], out["label"] = mb["image"].to(self.dtype), mb["label"].to(self.dtype)
        return out


def collate(samples):
    out = {}
    for k in samples[0]:
        vals = [s[k] for s in samples]
        out[k] = torch.stack(vals) if torch.is_tensor(vals[0]) else vals
    return out


class DeviceLoader:
    """Iterable over collated, device-resident mini-batches. shuffle: a fresh torch.randperm per epoch (the global CPU
    generator, seeded by train.py like the reference's set_determinism). num_workers > 0: ONE producer thread prepares
    the next batches on its own HIP stream while the current one trains (hand-over through an event + record_stream);
    num_workers == 0: batches are prepared inline."""

    def __init__(self, dataset, batch_size=1, shuffle=False, num_workers=0, prefetch=2):
        self.dataset, self.batch_size, self.shuffle = dataset, int(batch_size), shuffle
        self.num_workers, self.prefetch = int(num_workers), prefetch
        self.shard = (0, 1)          # (rank, world): data-parallel ranks take every world-th batch of the SAME permutation
        self.fused = None            # batch-level transform chain (FusedGraphSegBatches) replacing the per-sample one
        self.perm_generator = None   # data-parallel runs: the permutation's own generator (same on every rank); else one seeded from torch's global stream at the first epoch

    def reseed_augmentations(self, seed):
        """Give the random transforms of this loader their own streams (data-parallel ranks: seed + rank, so rank r's k-th batch
        does not get the flips / rotations of every other rank's k-th batch; only the batch permutation is shared)."""
        from .data_transforms import Randomizable
        for t in getattr(getattr(self.dataset, "transform", None), "transforms", []):
            if isinstance(t, Randomizable):
                t.set_random_state(seed=seed)
        if self.fused is not None:
            self.fused.aug.R_flip, self.fused.aug.R_rot90, self.fused.aug.R_rot = (np.random.RandomState(seed) for _ in range(3))

    def __len__(self):
        return ceil(ceil(len(self.dataset) / self.batch_size) / self.shard[1])

    def _batches(self):
        n = len(self.dataset)
        if self.shuffle and self.perm_generator is None:
            # as torch's DataLoader does: ONE draw from the global generator seeds a private one, so that later permutations (taken by the
            # producer thread, possibly while the main thread uses the global stream) do not interleave with it
            self.perm_generator = torch.Generator().manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
        order = torch.randperm(n, generator=self.perm_generator).tolist() if self.shuffle else list(range(n))     # all ranks share this stream, hence the order
        rank, world = self.shard
        starts = list(range(0, n, self.batch_size))
        if world > 1:                                                              # same number of steps on every rank
            starts = (starts + starts[: (-len(starts)) % world])[rank::world]
        for i in starts:
            if self.fused is not None:
                yield self.fused([self.dataset.items[j] for j in order[i:i + self.batch_size]])
            else:
                idx = order[i:i + self.batch_size]
                yield collate(self.dataset.get_batch(idx) if hasattr(self.dataset, "get_batch") else [self.dataset[j] for j in idx])

    _END = object()

    def _start_producer(self):
        """ONE producer thread for the life of the loader: it prepares epoch after epoch on its own HIP stream, at most `prefetch`
        mini-batches ahead, so the first batches of the next epoch are ready while the training loop still drains the current one
        and does its end-of-epoch work (the reference's DataLoader restarts its workers' prefetch at every epoch). Every random
        decision of the loader is taken in this thread, in the same order as an inline loader would take it."""
        self._q = queue.Queue(maxsize=self.prefetch)
        self._stop = threading.Event()
        self._go = threading.Semaphore(0)        # one permit per epoch the consumer has asked for (chains that share global random streams)
        dev = torch.cuda.current_device()
        stream = torch.cuda.Stream()
        q, stop, go = self._q, self._stop, self._go
        # Preparing epoch N + 1 while the consumer still validates / checkpoints after epoch N is only safe when every random decision of
        # the loader comes from PRIVATE streams: the fused batch chain (own RandomStates + the private permutation generator). The
        # per-sample chains draw from python's `random`, numpy's global stream and torch's global generator, which the main thread uses
        # too (GAN image pool, dropout): there the producer waits for the next __iter__ (round 4). This gate orders the streams
        # BETWEEN epochs only: within an epoch the producer still runs up to `prefetch` mini-batches ahead of the consumer on those global
        # streams, exactly as the reference's DataLoader workers run ahead of its training loop -- a main thread that draws from the
        # global generators while an epoch is being served interleaves with the loader (there as here).
        lookahead = self.fused is not None

        def put(item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    pass
            return False

        def produce():
            try:
                torch.cuda.set_device(dev)
                with torch.cuda.stream(stream):
                    while not stop.is_set():
                        if not lookahead:
                            while not go.acquire(timeout=0.1):
                                if stop.is_set():
                                    return
                        for b in self._batches():
                            ev = torch.cuda.Event()
                            ev.record(stream)
                            if not put((b, ev)):
                                return
                        if not put(DeviceLoader._END):
                            return
            except BaseException as e:  # noqa: BLE001 -- re-raised by the consumer
                put(e)

        self._thread = threading.Thread(target=produce, name="octa-loader", daemon=True)
        self._thread.start()

    def close(self):
        """Stop the producer thread (idempotent). Mini-batches it prepared ahead are dropped."""
        th = getattr(self, "_thread", None)
        if th is not None:
            self._stop.set()
            th.join()
            self._thread = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def __iter__(self):
        if self.num_workers <= 0 or not torch.cuda.is_available():
            yield from self._batches()
            return
        if getattr(self, "_thread", None) is None:
            self._start_producer()
        self._go.release()
        complete = False
        try:
            while True:
                item = self._q.get()
                if item is DeviceLoader._END:
                    complete = True
                    break
                if isinstance(item, BaseException):
                    self._thread = None
                    raise RuntimeError("the loader thread failed") from item
                b, ev = item
                cur = torch.cuda.current_stream()
                cur.wait_event(ev)
                for v in b.values():
                    if torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(cur)
                yield b
        finally:
            if not complete:          # the consumer left mid-epoch (break / exception): the next __iter__ must start a fresh epoch
                self.close()


def get_dataset(config: dict, phase: str, batch_size=None, num_workers=None) -> DeviceLoader:
    task = config["General"]["task"]
    amp_train = phase == Phase.TRAIN and bool(config["General"].get("amp"))
    # the reference hands fp16 tensors to its fp16 autocast (image_dataset.py:46); the MI355X path computes in bf16
    transform = _get_transformation(config, phase, dtype=torch.bfloat16 if amp_train else torch.float32)
    data = dict()
    for key, val in config[phase]["data"].items():
        paths = natsorted(glob(val["files"], recursive=True))
        assert len(paths) > 0, f"Error: Your provided file path {val['files']} for {key} does not match any files!"
        if "split" in val:
            assert os.path.isfile(val["split"]), f"Error: Your provided split file path {val['split']} for {key} does not exist."
            with open(val["split"], "r") as f:
                indices = [int(line.rstrip()) for line in f.readlines()]
            assert max(indices) < len(paths), (f"Error: Your provided split file for {key} does not seem to match your dataset! The index "
                                               f"{max(indices)} was requested but the dataset only contains {len(paths)} files.")
            paths = np.array(paths)[indices].tolist()
            assert len(paths) > 0, "Error: Your provided split file does not reference any file!"
        data[key] = paths
        data[key + "_path"] = paths

    def zipped():
        max_length = max(len(v) for v in data.values())
        cols = {k: np.resize(np.array(v), max_length).tolist() for k, v in data.items()}
        return ListDataset([dict(zip(cols, t)) for t in zip(*cols.values())], transform)

    if task == Task.VESSEL_SEGMENTATION:
        data_set = zipped()
    elif task == Task.GAN_VESSEL_SEGMENTATION:
        data_set = zipped() if phase == Phase.VALIDATION else UnalignedZipDataset(data, transform, phase)
    else:
        raise NotImplementedError(f"task {task} is outside the MI355X hot path")
    loader = DeviceLoader(data_set, batch_size=batch_size or config[phase].get("batch_size") or 1, shuffle=phase != Phase.TEST,
                          num_workers=1 if num_workers is None else num_workers)
    # the permutation stream is the loader's own (seeded from General.seed and the phase): loaders prepare batches from their own
    # threads, ahead of the training loop, and must not race for torch's global generator
    loader.perm_generator = torch.Generator().manual_seed(int(config["General"].get("seed", 42)) + {"Train": 0, "Validation": 1, "Test": 2}.get(getattr(phase, "value", phase), 3))
    aug_config = config[phase]["data_augmentation"]
    if (isinstance(data_set, ListDataset) and FusedGraphSegBatches.applies(aug_config) and not config["General"].get("generic_loader")
            and all("blackdict" not in it for it in data_set.items)):
        loader.fused = FusedGraphSegBatches(aug_config, config["General"].get("seed", 42),
                                            torch.bfloat16 if amp_train else torch.float32)
    return loader
