"""Data formats on the input side of the hot path.

Readers for the two benchmarks' on-disk layouts, mirroring ``FairFedMedDataset`` / ``FedChexMimicDataset``
(utils/data_utils.py:559-790) and the loader surface the trainer consumes
(Dassl/dassl/data/data_manager.py:20-59, 104-133, 435-502):

    <root>/fairfedmed/all/<file>.npz            slo_fundus u8 [H,W] | oct_bscans u8 [128,H,W], glaucoma, race, gender, ...
    <root>/fairfedmed/meta_site{k}_{attr}_{train|test}.csv      column 'filename'
    <root>/fedchexmimic/meta_{chexpert|mimic}_{attr}_{train|test}.csv   filename, disease_label, <attr>_label columns

``__getitem__`` returns what the reference returns: (float32 [C,H,W] raw 0..255, label int64, attrs int64 [n_attr]).
``raw(item)`` returns the same sample in its TRANSPORT form: uint8, before the float conversion and the channel
repeat (1 channel for SLO fundus / chest X-ray), 12x fewer bytes over PCIe; the engine expands it on the GPU
(``ffm_expand_u8``), bit-identically, because uint8 -> float32 is exact.

scikit-image is not in this image: ``resize_image`` restates ``skimage.transform.resize`` (order 1, mode 'reflect',
no anti-aliasing when enlarging, clip to the input range) on scipy.ndimage.zoom - parity for the resize branch is
unpinned; the branches without a resize - both readers, FairFedMedDataset and FedChexMimicDataset, with their
count_by_attribute - are pinned against the imported reference (tests/golden/make_golden.py -> dataset.json).
"""
from __future__ import annotations

import csv
import os
from types import SimpleNamespace as NS
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

CLASSNAMES = {"FairFedMed": ["NOT Glaucoma", "Glaucoma"],                        # datasets/FairFedMed.py:47-48
              "FedChexMimic": ["NOT Pleural Effusion", "Pleural Effusion"]}      # datasets/FedChexMimic.py:47-48
DATASET_DIRS = {"FairFedMed": "fairfedmed", "FedChexMimic": "fedchexmimic"}
_FILTERED = {"gender", "maritalstatus", "hispanic", "language", "ethnicity", "race"}   # utils/data_utils.py:582


def resize_image(img: np.ndarray, shape) -> np.ndarray:
    """skimage.transform.resize(img, shape) for a 2-D float image (defaults: order 1, mode='reflect', clip=True,
    anti-aliasing Gaussian with sigma = (in/out - 1) / 2 when an axis shrinks), restated on scipy.ndimage because
    scikit-image is not installable here; pinned by tests/test_data_cpu.py against values worked out by hand from the
    published algorithm and against a loop-level implementation of it that shares no code with this one."""
    from scipy import ndimage as ndi
    img = np.asarray(img)
    out_dtype = img.dtype if img.dtype.kind == "f" else np.float64
    src = img.astype(np.float64)
    factors = [o / i for o, i in zip(shape, img.shape)]
    if any(f < 1 for f in factors):                       # anti-aliasing filter when shrinking (skimage default)
        sigma = [max(0.0, (1 / f - 1) / 2) for f in factors]
        src = ndi.gaussian_filter(src, sigma, mode="mirror")
    out = ndi.zoom(src, factors, order=1, mode="mirror", grid_mode=True)
    out = np.clip(out, img.min(), img.max())
    return out.astype(out_dtype)


def _read_csv(path: str) -> Dict[str, list]:
    with open(path, newline="") as f:
        rows = list(csv.DictReader(f))
    if not rows or "filename" not in rows[0]:
        raise AssertionError("filename must be included in the head")
    return {k: [r[k] for r in rows] for k in rows[0]}


class FairFedMedDataset:
    """utils/data_utils.py:559-726 for the modalities the FairLoRA scripts use."""

    def __init__(self, base_path, site, attribute_type=None, attributes=None, modality_type=None, resolution=224,
                 depth=3, train=True, transform=None):
        self.task = "cls"
        self.base_path, self.data_path = base_path, os.path.join(base_path, "all")
        self.modality_type, self.attribute_type, self.attributes = modality_type, attribute_type, attributes
        split = "train" if train else "test"
        files = _read_csv(os.path.join(base_path, f"meta_site{site}_{attribute_type}_{split}.csv"))["filename"]
        if modality_type is None:
            raise AssertionError("modality_type")
        if modality_type not in ("oct_bscans", "oct_bscans_3d", "slo_fundus"):
            raise NotImplementedError(modality_type)
        key = "oct_bscans" if modality_type.startswith("oct") else "slo_fundus"
        self.data_files, self.data_attrs = [], []
        for x in files:
            with np.load(os.path.join(self.data_path, x), allow_pickle=True) as raw:
                attr = raw[attribute_type].item()
                if attribute_type in _FILTERED and not attr > -1:      # -1 = unknown: dropped (:586)
                    continue
                if len(raw[key]) > 0:
                    self.data_files.append(x)
                    self.data_attrs.append(attr)
        self.depth, self.resolution, self.transform = depth, resolution, transform

    def __len__(self):
        return len(self.data_files)

    def _load(self, item):
        return np.load(os.path.join(self.data_path, self.data_files[item]), allow_pickle=True)

    def _meta(self, raw):
        label = int(float(raw["glaucoma"].item()))
        attrs = [int(raw[k]) for k in self.attributes] if (self.attribute_type is not None and self.attributes) else []
        return label, attrs

    def raw(self, item):
        """(image in transport form [C1,H,W], channel repeat, label, attrs); uint8 when the file holds uint8 and no
        resize is needed, else float32."""
        with self._load(item) as raw:
            label, attrs = self._meta(raw)
            if self.modality_type == "slo_fundus":
                img = np.transpose(raw["slo_fundus"])[None, :, :]
                rep = self.depth if self.depth > 1 else 1
            elif self.modality_type == "oct_bscans":
                img, rep = raw["oct_bscans"][::4], 1                   # 128 -> 32 B-scans (:639)
            else:                                                      # oct_bscans_3d: [1, D, H, W] volume
                img, rep = raw["oct_bscans"][None], 1
        if self.modality_type != "oct_bscans_3d" and img.shape[1] != self.resolution:
            img = np.stack([resize_image(s.astype(np.float32), (self.resolution, self.resolution)) for s in img])
        if img.dtype != np.uint8:
            img = img.astype(np.float32) if self.modality_type != "oct_bscans_3d" else img.astype(int).astype(np.float32)
        return np.ascontiguousarray(img), rep, label, attrs

    def __getitem__(self, item):
        img, rep, label, attrs = self.raw(item)
        data = img.astype(np.float32)
        if rep > 1:
            data = np.repeat(data, rep, axis=0)
        if self.transform is not None and self.modality_type != "slo_fundus":
            data = self.transform(data)
        return data, torch.tensor(label).long(), torch.tensor(attrs)

    def count_by_attribute(self, attr: str) -> List[int]:
        """Dassl/dassl/data/data_manager.py:443-460: samples per group id 0..max."""
        vals = []
        for item in range(len(self)):
            with self._load(item) as raw:
                vals.append(int(raw[attr]))
        return [vals.count(g) for g in range(max(vals) + 1)]


class FedChexMimicDataset:
    """utils/data_utils.py:729-790: gray-scale chest X-rays (jpg) listed in a csv with label columns."""

    def __init__(self, base_path, site, attribute_type, attributes, modality_type=None, resolution=224, depth=3,
                 train=True, transform=None):
        self.task = "cls"
        self.base_path = base_path
        if site == 1:
            name, self.data_path = "chexpert", base_path
        elif site == 2:
            name, self.data_path = "mimic", os.path.join(base_path, "files_336p")
        else:
            raise NotImplementedError(site)
        self.modality_type, self.attribute_type, self.attributes = modality_type, attribute_type, attributes
        cols = _read_csv(os.path.join(base_path, f"meta_{name}_{attribute_type}_{'train' if train else 'test'}.csv"))
        self.data_files = cols["filename"]
        self.data_attrs = [int(v) for v in cols[attribute_type + "_label"]]
        self.disease_labels = [int(v) for v in cols["disease_label"]]
        self.data_attributes = [[int(v) for v in cols[k + "_label"]] for k in attributes]
        self.depth, self.resolution, self.transform = depth, resolution, transform

    def __len__(self):
        return len(self.data_files)

    def raw(self, item):
        from PIL import Image
        img = np.array(Image.open(os.path.join(self.data_path, self.data_files[item])).convert("L"))[None, :, :]
        if img.shape[1] != self.resolution:
            img = np.stack([resize_image(s.astype(np.float32), (self.resolution, self.resolution)) for s in img])
        attrs = [a[item] for a in self.data_attributes]
        return np.ascontiguousarray(img), (self.depth if self.depth > 1 else 1), self.disease_labels[item], attrs

    def __getitem__(self, item):
        img, rep, label, attrs = self.raw(item)
        data = img.astype(np.float32)
        if rep > 1:
            data = np.repeat(data, rep, axis=0)
        return data, torch.tensor(label).long(), torch.tensor(attrs)

    def count_by_attribute(self, attr: str) -> List[int]:
        """Dassl/dassl/data/data_manager.py:462-473."""
        vals = self.data_attributes[list(self.attributes).index(attr)]
        return [vals.count(g) for g in range(max(vals) + 1)]


class FedLoader:
    """The DataLoader the trainer iterates (build_data_loader, data_manager.py:20-59): random order and drop_last for
    training, sequential for testing, batches in the dict contract {"img", "label", "attrs"}.

    transport="uint8" ships the sample's transport form when it is uint8 (``img`` is then uint8 [B,C1,H,W] and the
    engine expands / repeats the channels on the GPU); "float32" ships what the reference ships.  Batches are
    assembled in pinned memory so that the trainer's ``.to(device, non_blocking=True)`` is an asynchronous copy."""

    def __init__(self, dataset, batch_size: int, train: bool, seed: int = 0, transport: str = "float32",
                 pin_memory: Optional[bool] = None):
        assert transport in ("float32", "uint8")
        self.dataset, self.batch_size, self.train, self.transport = dataset, batch_size, train, transport
        self.drop_last = train and len(dataset) >= batch_size
        self.gen = np.random.default_rng(seed)
        self.pin = torch.cuda.is_available() if pin_memory is None else pin_memory
        assert len(self) > 0

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _collate(self, idx: Sequence[int]) -> Dict[str, torch.Tensor]:
        imgs, labels, attrs = [], [], []
        for i in idx:
            img, rep, label, a = self.dataset.raw(int(i))
            if self.transport == "float32" or img.dtype != np.uint8:
                img = img.astype(np.float32)
                if rep > 1:
                    img = np.repeat(img, rep, axis=0)
            imgs.append(img)
            labels.append(label)
            attrs.append(a)
        out = {"img": torch.from_numpy(np.stack(imgs)), "label": torch.tensor(labels, dtype=torch.int64),
               "attrs": torch.tensor(attrs, dtype=torch.int64).reshape(len(idx), -1)}
        if self.pin:
            out = {k: v.pin_memory() for k, v in out.items()}
        return out

    def __iter__(self):
        n = len(self.dataset)
        order = self.gen.permutation(n) if self.train else np.arange(n)
        for b in range(len(self)):
            yield self._collate(order[b * self.batch_size:(b + 1) * self.batch_size])


class FedData:
    """What ``GLP_OT_SVLoRA(cfg, data=...)`` needs from the reference's DataManager: per-client loaders, class names."""

    def __init__(self, cfg, transport: str = "float32"):
        name = cfg.DATASET.NAME
        if name not in DATASET_DIRS:
            raise NotImplementedError(name)
        root = os.path.join(os.path.abspath(os.path.expanduser(cfg.DATASET.ROOT)), DATASET_DIRS[name])
        cls = FairFedMedDataset if name == "FairFedMed" else FedChexMimicDataset
        kw = dict(attribute_type=cfg.DATASET.ATTRIBUTE_TYPE, attributes=list(cfg.DATASET.ATTRIBUTES),
                  modality_type=getattr(cfg.DATASET, "MODALITY_TYPE", None), resolution=cfg.INPUT.SIZE[0], depth=3)
        self.fed_train_loader_x_dict, self.fed_test_loader_x_dict = {}, {}
        seed = getattr(cfg, "SEED", 1)
        for net_id in range(cfg.DATASET.USERS):
            tr = cls(root, net_id + 1, train=True, **kw)
            te = cls(root, net_id + 1, train=False, **kw)
            self.fed_train_loader_x_dict[net_id] = FedLoader(tr, cfg.DATALOADER.TRAIN_X.BATCH_SIZE, True,
                                                             seed=seed * 1000 + net_id, transport=transport)
            self.fed_test_loader_x_dict[net_id] = FedLoader(te, cfg.TEST.BATCH_SIZE, False, transport=transport)
        self.classnames = list(CLASSNAMES[name])
        self.dataset = NS(classnames=self.classnames)
        self.num_classes = len(self.classnames)
        self.lab2cname = {i: n for i, n in enumerate(self.classnames)}


# ------------------------------------------------------------------------------------------------ synthetic files --
def write_synthetic_fairfedmed(root: str, sites: int = 2, n_train: int = 12, n_test: int = 6, size: int = 224,
                               seed: int = 0, modality: str = "slo_fundus", attribute_type: str = "race",
                               unknown_every: int = 0) -> str:
    """A FairFedMed tree of random uint8 samples (tests, input-path benchmark).  Returns <root>/fairfedmed."""
    g = np.random.Generator(np.random.Philox(key=[0xDA7A, seed & 0xFFFFFFFF]))
    base = os.path.join(root, DATASET_DIRS["FairFedMed"])
    os.makedirs(os.path.join(base, "all"), exist_ok=True)
    k = 0
    for site in range(1, sites + 1):
        for split, n in (("train", n_train), ("test", n_test)):
            names = []
            for _ in range(n):
                name = f"data_{k:05d}.npz"
                arrs = {"glaucoma": np.array(int(g.integers(0, 2))), "race": np.array(int(g.integers(0, 3))),
                        "gender": np.array(int(g.integers(0, 2))), "ethnicity": np.array(int(g.integers(0, 2))),
                        "language": np.array(int(g.integers(0, 3)))}
                if unknown_every and k % unknown_every == unknown_every - 1:
                    arrs[attribute_type] = np.array(-1)
                if modality == "slo_fundus":
                    arrs["slo_fundus"] = g.integers(0, 256, size=(size, size), dtype=np.uint8)
                    arrs["oct_bscans"] = np.zeros((0,), np.uint8)
                else:
                    arrs["oct_bscans"] = g.integers(0, 256, size=(128, size, size), dtype=np.uint8)
                    arrs["slo_fundus"] = np.zeros((0,), np.uint8)
                np.savez(os.path.join(base, "all", name), **arrs)
                names.append(name)
                k += 1
            with open(os.path.join(base, f"meta_site{site}_{attribute_type}_{split}.csv"), "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["filename"])
                w.writerows([[x] for x in names])
    return base


def write_synthetic_fedchexmimic(root: str, n_train: int = 12, n_test: int = 6, size: int = 224, seed: int = 0,
                                 attribute_type: str = "gender") -> str:
    """A FedChexMimic tree (utils/data_utils.py:729-753): site 1 "chexpert" with its images under the base path, site 2
    "mimic" under <base>/files_336p; csv columns filename, disease_label, gender_label, race_label.  The images alternate
    between 8-bit gray PNG, RGB JPEG and RGB PNG, so that the reader's ``convert('L')`` sees all three.  Returns
    <root>/fedchexmimic."""
    from PIL import Image
    g = np.random.Generator(np.random.Philox(key=[0xC4E5, seed & 0xFFFFFFFF]))
    base = os.path.join(root, DATASET_DIRS["FedChexMimic"])
    k = 0
    for name, sub in (("chexpert", "."), ("mimic", "files_336p")):
        for split, n in (("train", n_train), ("test", n_test)):
            rows = []
            for _ in range(n):
                kind = k % 3
                rel = os.path.join("imgs" if name == "chexpert" else "p10", f"view_{k:05d}." + ("jpg" if kind == 1 else "png"))
                path = os.path.join(base, sub, rel)
                os.makedirs(os.path.dirname(path), exist_ok=True)
                if kind == 0:
                    Image.fromarray(g.integers(0, 256, size=(size, size), dtype=np.uint8), "L").save(path)
                else:
                    # smooth RGB content (a JPEG of white noise is all quantisation error)
                    low = g.integers(0, 256, size=(size // 4 + 1, size // 4 + 1, 3), dtype=np.uint8)
                    arr = np.kron(low, np.ones((4, 4, 1), np.uint8))[:size, :size]
                    Image.fromarray(arr, "RGB").save(path, **({"quality": 90} if kind == 1 else {}))
                rows.append([rel, int(g.integers(0, 2)), int(g.integers(0, 2)), int(g.integers(0, 3))])
                k += 1
            with open(os.path.join(base, f"meta_{name}_{attribute_type}_{split}.csv"), "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["filename", "disease_label", "gender_label", "race_label"])
                w.writerows(rows)
    return base
