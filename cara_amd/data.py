"""VTAB-1k input pipeline, MI355X-first.

The reference (``/root/reference/image_classification/vtab.py:36-107``) decodes every image again in
every epoch with 4 DataLoader workers: ``ImageFilelist`` over ``impath label`` lines, ``Resize((224,224),
bicubic)`` + ``ToTensor`` + ImageNet ``Normalize``, ``DataLoader(batch 64, shuffle, drop_last)`` for
training and ``DataLoader(batch 256)`` for evaluation.  At 5 000+ images/s per GPU that feed is the
bottleneck, and a VTAB-1k task is 1 000 training images: 150 MB as resized uint8 pixels.  So the whole split is
decoded ONCE (same arithmetic: PIL bicubic resize of the RGB image) into one device-resident uint8 tensor, and an epoch
is an index permutation plus ``index_select`` and the /255 + per-channel normalise of the drawn batch on the GPU.  Under
data parallelism every rank holds the split and takes the rank-strided part of the same epoch-seeded
permutation (``dist.epoch_shard``), which is ``drop_last`` per rank like the reference loader.

No torchvision here (not installed): the three transforms are restated on PIL + torch and checked
against an independent numpy computation in tests/test_data.py.
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import dist as D

# vtab.py:9-31
DATASET_NAMES = ("cifar", "caltech101", "dtd", "oxford_flowers102", "oxford_iiit_pet", "svhn", "sun397",
                 "patch_camelyon", "eurosat", "resisc45", "diabetic_retinopathy", "clevr_count", "clevr_dist",
                 "dmlab", "kitti", "dsprites_loc", "dsprites_ori", "smallnorb_azi", "smallnorb_ele")
CLASSES_NUM = (100, 102, 47, 102, 37, 10, 397, 2, 10, 45, 5, 8, 6, 6, 4, 16, 16, 18, 9)
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def get_classes_num(dataset_name: str) -> int:
    """vtab.py:32-34 (KeyError on an unknown name, like the reference's dict lookup)."""
    return dict(zip(DATASET_NAMES, CLASSES_NUM))[dataset_name]


def read_filelist(flist: str) -> List[Tuple[str, int]]:
    """vtab.py:40-50: one ``impath label`` pair per line, split on whitespace (a line with any other
    number of fields raises ValueError, as the reference's tuple unpacking does)."""
    out = []
    with open(flist, "r") as fh:
        for line in fh.readlines():
            impath, imlabel = line.strip().split()
            out.append((impath, int(imlabel)))
    return out


def decode_image_u8(path: str, size: int = 224) -> torch.Tensor:
    """vtab.py:36-37 + the Resize of :91: RGB -> bicubic resize to size x size (PIL semantics, as torchvision applies
    to PIL images).  uint8 [3,size,size]."""
    from PIL import Image
    with Image.open(path) as im:
        im = im.convert("RGB").resize((size, size), Image.BICUBIC)
        a = np.asarray(im, dtype=np.uint8)
    return torch.from_numpy(a.copy()).permute(2, 0, 1).contiguous()


def normalize_u8(x: torch.Tensor) -> torch.Tensor:
    """ToTensor + Normalize of vtab.py:92-94 on uint8 [..., 3, H, W] (any device): /255, (x - mean) / std, fp32.
    The same fp32 operations in the same order on the CPU and on the GPU (equal to within one ulp: the GPU's fp32
    division is not correctly rounded in every case)."""
    mean = torch.tensor(IMAGENET_MEAN, device=x.device).view(3, 1, 1)
    std = torch.tensor(IMAGENET_STD, device=x.device).view(3, 1, 1)
    return x.to(torch.float32).div_(255.0).sub_(mean).div_(std)


def decode_image(path: str, size: int = 224) -> torch.Tensor:
    """The whole transform of vtab.py:91-94 for one file: fp32 [3,size,size]."""
    return normalize_u8(decode_image_u8(path, size))


class ResidentSplit:
    """One file list decoded once and kept on ``device`` as the resized uint8 pixels ``pixels`` [N,3,size,size]
    (150 KB per image: the 73k-image dsprites test split is 11 GB, not 44) plus ``labels`` int64 [N]; a batch is
    normalised to fp32 when it is drawn.  The split is decoded and uploaded in chunks, so the host never holds more
    than one chunk.  ``len()`` and ``[i]`` behave like the reference's ``ImageFilelist``."""

    def __init__(self, root: str, flist: str, device="cuda", size: int = 224, workers: int = 8, chunk: int = 2048):
        self.root, self.imlist = root, read_filelist(flist)
        paths = [os.path.join(root, p) for p, _ in self.imlist]
        self.pixels = torch.empty(len(paths), 3, size, size, dtype=torch.uint8, device=device)
        for c0 in range(0, len(paths), chunk):
            part = paths[c0:c0 + chunk]
            if workers > 1 and len(part) > 1:
                with ThreadPoolExecutor(max_workers=workers) as ex:   # PIL releases the GIL while decoding/resizing
                    decoded = list(ex.map(lambda p: decode_image_u8(p, size), part))
            else:
                decoded = [decode_image_u8(p, size) for p in part]
            self.pixels[c0:c0 + len(part)].copy_(torch.stack(decoded))
        self.labels = torch.tensor([l for _, l in self.imlist], dtype=torch.int64, device=device)

    @property
    def images(self) -> torch.Tensor:
        """The whole split normalised, fp32 [N,3,size,size] (materialised: for small splits and tests)."""
        return normalize_u8(self.pixels)

    def __len__(self) -> int:
        return len(self.imlist)

    def __getitem__(self, i: int):
        return normalize_u8(self.pixels[i]), int(self.labels[i])

    # ---- loaders -----------------------------------------------------------------------------------
    def train_batches(self, batch_size: int = 64, seed: int = 0, rank: Optional[int] = None,
                      world: Optional[int] = None) -> Callable[[int], Iterator[Tuple[torch.Tensor, torch.Tensor]]]:
        """``f(epoch)`` -> iterator of (images, labels) on the device: shuffle + drop_last per rank
        (vtab.py:84-88), rank-strided shard of one epoch-seeded permutation under data parallelism.
        The shape ``recipe.fit`` expects for ``train_batches``."""
        rank = D.get_rank() if rank is None else rank
        world = D.world_size() if world is None else world

        def epoch_iter(epoch: int):
            for idx in D.epoch_shard(len(self), epoch, rank, world, batch_size, seed):
                idx = idx.to(self.pixels.device)
                yield normalize_u8(self.pixels.index_select(0, idx)), self.labels.index_select(0, idx)
        return epoch_iter

    def eval_batches(self, batch_size: int = 256) -> Callable[[], Iterator[Tuple[torch.Tensor, torch.Tensor]]]:
        """In file order, last batch partial (vtab.py:96-100: shuffle False, no drop_last)."""
        def it():
            for i in range(0, len(self), batch_size):
                yield normalize_u8(self.pixels[i:i + batch_size]), self.labels[i:i + batch_size]
        return it


def get_data(name: str, evaluate: bool = True, batch_size: int = 64, root: Optional[str] = None, device="cuda",
             seed: int = 0, workers: int = 8):
    """Drop-in for ``vtab.get_data`` (vtab.py:88-107): the same split files -- ``train800val200.txt`` /
    ``test.txt`` when ``evaluate`` else ``train800.txt`` / ``val200.txt`` -- under ``./data/vtab-1k/<name>``.
    Returns (train_batches, test_batches) in the form ``recipe.fit`` takes instead of two DataLoaders."""
    root = root if root is not None else "./data/vtab-1k/" + name
    tr, te = ("train800val200.txt", "test.txt") if evaluate else ("train800.txt", "val200.txt")
    train = ResidentSplit(root, os.path.join(root, tr), device=device, workers=workers)
    test = ResidentSplit(root, os.path.join(root, te), device=device, workers=workers)
    return train.train_batches(batch_size, seed=seed), test.eval_batches(256)
