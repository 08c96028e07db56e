"""VTAB-1k input pipeline (cara_amd/data.py) against the reference's loader semantics
(/root/reference/image_classification/vtab.py:36-107).  CPU only: this is data plumbing, not the HIP path."""
import os

import numpy as np
import pytest
import torch

from cara_amd import data as D


def _make_split(tmp_path, n=23, classes=5, seed=0):
    from PIL import Image
    rng = np.random.default_rng(seed)
    root = tmp_path / "vtab-1k" / "cifar"
    (root / "images").mkdir(parents=True)
    lines = []
    for i in range(n):
        h, w = int(rng.integers(20, 60)), int(rng.integers(20, 60))
        mode = "RGB" if i % 3 else "L"                      # grey files must come out as 3 channels (convert('RGB'))
        arr = rng.integers(0, 256, size=(h, w, 3) if mode == "RGB" else (h, w), dtype=np.uint8)
        rel = f"images/{i:03d}.png"
        Image.fromarray(arr, mode=mode).save(root / rel)
        lines.append(f"{rel} {int(rng.integers(0, classes))}")
    for name in ("train800val200.txt", "test.txt", "train800.txt", "val200.txt"):
        (root / name).write_text("\n".join(lines) + "\n")
    return str(root), lines


def test_classes_table():
    assert D.get_classes_num("cifar") == 100 and D.get_classes_num("sun397") == 397   # vtab.py:30-34
    assert len(D.DATASET_NAMES) == len(D.CLASSES_NUM) == 19
    with pytest.raises(KeyError):
        D.get_classes_num("imagenet")


def test_filelist_reader(tmp_path):
    f = tmp_path / "l.txt"
    f.write_text("a/b.png 3\nc.png   17\n")
    assert D.read_filelist(str(f)) == [("a/b.png", 3), ("c.png", 17)]
    f.write_text("only_one_field\n")
    with pytest.raises(ValueError):                         # the reference's tuple unpacking (vtab.py:47)
        D.read_filelist(str(f))


def test_decode_matches_independent_transform(tmp_path):
    from PIL import Image
    root, lines = _make_split(tmp_path, n=4)
    for line in lines:
        rel, _ = line.split()
        got = D.decode_image(os.path.join(root, rel))
        im = Image.open(os.path.join(root, rel)).convert("RGB").resize((224, 224), resample=3)   # interpolation=3: bicubic
        ref = (np.asarray(im, dtype=np.float64) / 255.0 - np.array(D.IMAGENET_MEAN)) / np.array(D.IMAGENET_STD)
        assert got.shape == (3, 224, 224) and got.dtype == torch.float32
        np.testing.assert_allclose(got.permute(1, 2, 0).numpy(), ref, rtol=0, atol=2e-6)


def test_resident_split_and_loaders(tmp_path):
    root, lines = _make_split(tmp_path, n=23)
    split = D.ResidentSplit(root, os.path.join(root, "train800.txt"), device="cpu", workers=4)
    assert len(split) == 23 and split.images.shape == (23, 3, 224, 224)
    assert split.labels.tolist() == [int(l.split()[1]) for l in lines]
    img0, lab0 = split[0]
    assert torch.equal(img0, D.decode_image(os.path.join(root, lines[0].split()[0]))) and lab0 == int(lines[0].split()[1])
    # training: shuffle + drop_last, a different permutation every epoch, the same for a given (seed, epoch)
    tb = split.train_batches(batch_size=4, seed=7, rank=0, world=1)
    e0 = [y for _, y in tb(0)]
    assert len(e0) == 23 // 4 and all(len(y) == 4 for y in e0)
    xs0 = torch.cat([x for x, _ in tb(0)])
    assert torch.equal(xs0, torch.cat([x for x, _ in tb(0)])) and not torch.equal(xs0, torch.cat([x for x, _ in tb(1)]))
    # data parallel: the two ranks' shards of one epoch are disjoint, equally long, and come from one permutation
    seen = []
    for r in range(2):
        tbr = split.train_batches(batch_size=4, seed=7, rank=r, world=2)
        bs = list(tbr(3))
        assert len(bs) == (23 // 2) // 4
        for x, _ in bs:
            for img in x:
                hits = [i for i in range(23) if torch.equal(img, split.images[i])]
                assert len(hits) == 1
                seen.append(hits[0])
    assert len(seen) == len(set(seen)) == 2 * 2 * 4
    # evaluation: file order, partial last batch
    eb = list(split.eval_batches(batch_size=10)())
    assert [len(y) for _, y in eb] == [10, 10, 3]
    assert torch.equal(torch.cat([y for _, y in eb]), split.labels)


def test_get_data_split_names(tmp_path):
    root, _ = _make_split(tmp_path, n=9)
    tr, te = D.get_data("cifar", evaluate=True, batch_size=4, root=root, device="cpu", workers=1)
    assert len(list(tr(0))) == 2 and sum(len(y) for _, y in te()) == 9
    tr2, te2 = D.get_data("cifar", evaluate=False, batch_size=4, root=root, device="cpu", workers=1)
    assert len(list(tr2(0))) == 2 and sum(len(y) for _, y in te2()) == 9
    with pytest.raises(FileNotFoundError):
        D.get_data("dtd", root=str(tmp_path / "nope"), device="cpu")
