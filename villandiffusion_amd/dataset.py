"""Backdoor triggers/targets and the poisoned dataset, MI355X-native (D1-D6 of SURVEY.md §8a).

Same surface as the reference's ``dataset.Backdoor`` / ``dataset.DatasetLoader`` (dataset.py:42-968): constants,
``set_poison(...).prepare_dataset(mode=...)``, ``get_dataloader()``, batch keys
``image, pixel_values, pixel_values_trigger, trigger, target, label, is_clean``.

What is different underneath: the whole uint8 dataset (60 000 x 32x32x3 = 184 MB for CIFAR10) lives in HBM, the
FIXED/FLEX/EXTEND poison split is an index partition with per-sample flags, and a batch is produced by ONE kernel
(``vd_poison_batch``: uint8 -> ToTensor -> util.normalize -> random h-flip -> trigger stamping / target select)
instead of 8 CPU DataLoader workers (dataset.py:467).  Image-file triggers/targets are decoded once with PIL.
"""
from __future__ import annotations

import os
import pickle
from typing import Dict, Iterator, List, Optional, Tuple, Union

import numpy as np
import torch

from . import ops

DEFAULT_VMIN, DEFAULT_VMAX = float(-1.0), float(1.0)
_ASSET_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def normalize(x, vmin_in=None, vmax_in=None, vmin_out=0, vmax_out=1, eps: float = 1e-5):
    """util.py:119-147 (host version, used for the one-off trigger/target tensors)."""
    if vmax_out is None and vmin_out is None:
        return x
    lo = x.min() if vmin_in is None else vmin_in
    hi = x.max() if vmax_in is None else vmax_in
    if vmax_out is None:
        vmax_out = hi
    if vmin_out is None:
        vmin_out = lo
    return ((x - lo) / (hi - lo + eps)) * (vmax_out - vmin_out) + vmin_out


class Backdoor:
    CHANNEL_LAST, CHANNEL_FIRST = -1, -3
    GREY_BG_RATIO = 0.3
    STOP_SIGN_IMG, CAT_IMG, GLASSES_IMG = "static/stop_sign_wo_bg.png", "static/cat_wo_bg.png", "static/glasses.png"
    TARGET_FA, TARGET_TG, TARGET_BOX, TARGET_SHIFT = "SHOE", "NOSHIFT", "CORNER", "SHIFT"
    TARGET_HAT, TARGET_FEDORA_HAT, TARGET_CAT = "BWHAT", "HAT", "CAT"
    TRIGGER_GAP_X = TRIGGER_GAP_Y = 2
    TRIGGER_NONE, TRIGGER_FA, TRIGGER_FA_EZ, TRIGGER_MNIST, TRIGGER_MNIST_EZ = "NONE", "FASHION", "FASHION_EZ", "MNIST", "MNIST_EZ"
    TRIGGER_SM_BOX, TRIGGER_XSM_BOX, TRIGGER_XXSM_BOX, TRIGGER_XXXSM_BOX, TRIGGER_BIG_BOX = \
        "SM_BOX", "XSM_BOX", "XXSM_BOX", "XXXSM_BOX", "BIG_BOX"
    TRIGGER_BIG_BOX_MED, TRIGGER_SM_BOX_MED, TRIGGER_XSM_BOX_MED, TRIGGER_XXSM_BOX_MED, TRIGGER_XXXSM_BOX_MED = \
        "BOX_18", "BOX_14", "BOX_11", "BOX_8", "BOX_4"
    TRIGGER_GLASSES = "GLASSES"
    TRIGGER_BIG_STOP_SIGN, TRIGGER_SM_STOP_SIGN, TRIGGER_XSM_STOP_SIGN, TRIGGER_XXSM_STOP_SIGN, TRIGGER_XXXSM_STOP_SIGN = \
        "STOP_SIGN_18", "STOP_SIGN_14", "STOP_SIGN_11", "STOP_SIGN_8", "STOP_SIGN_4"

    _WHITE = {"SM_BOX": 14, "XSM_BOX": 11, "XXSM_BOX": 8, "XXXSM_BOX": 4, "BIG_BOX": 18}
    _GREY = {"BOX_18": 18, "BOX_14": 14, "BOX_11": 11, "BOX_8": 8, "BOX_4": 4}
    _STOP = {"STOP_SIGN_18": 18, "STOP_SIGN_14": 14, "STOP_SIGN_11": 11, "STOP_SIGN_8": 8, "STOP_SIGN_4": 4}
    _IMG_TARGETS = {"BWHAT": "static/hat.png", "HAT": "static/fedora-hat.png", "CAT": "static/cat_wo_bg.png"}

    def __init__(self, root: Optional[str] = None):
        self._root = root

    def _asset(self, rel: str) -> str:
        for base in (os.getcwd(), _ASSET_ROOT):
            p = os.path.join(base, rel)
            if os.path.exists(p):
                return p
        raise FileNotFoundError(rel)

    def _load_rgb(self, rel: str, size, channel: int) -> torch.Tensor:
        """convert('RGB'|'L') -> Resize (PIL bilinear; int = short side) -> ToTensor   (dataset.py:689-703)."""
        from PIL import Image
        img = Image.open(self._asset(rel)).convert("RGB" if channel == 3 else "L")
        return _pil_to_tensor01(_pil_resize(img, size))

    @staticmethod
    def _bg2grey(t, vmin, vmax):
        thres = (vmax - vmin) * Backdoor.GREY_BG_RATIO + vmin
        t = t.clone()
        t[t <= thres] = thres
        return t

    @staticmethod
    def _bg2black(t, vmin, vmax):                                      # dataset.py:712-715
        thres = (vmax - vmin) * Backdoor.GREY_BG_RATIO + vmin
        t = t.clone()
        t[t <= thres] = vmin
        return t

    # trigger id -> (torchvision dataset name, train item, roll dx, roll dy)   dataset.py:791-812
    _IDX_TRIGGERS = {"FASHION": ("FashionMNIST", 0, 0, 2), "FASHION_EZ": ("FashionMNIST", 144, 0, 4),
                     "MNIST": ("MNIST", 3, 10, 3), "MNIST_EZ": ("MNIST", 6, 10, 3)}

    def _idx_item(self, ds: str, item: int, size, channel: int) -> torch.Tensor:
        """Train image `item` of torchvision's MNIST / FashionMNIST (an 'L' PIL image) through the reference's transform
        (dataset.py:689-703): Grayscale(1) | convert('RGB') -> Resize(size) -> ToTensor.  The reference lets torchvision download
        the files (download=True); there is no network here, so they must already sit where torchvision would have put them:
        <root>/<ds>/raw/train-images-idx3-ubyte[.gz]."""
        from PIL import Image
        imgs = read_idx_images(locate_idx(self._root, ds, "train-images-idx3-ubyte"))
        img = Image.fromarray(imgs[item], mode="L").convert("RGB" if channel == 3 else "L")
        return _pil_to_tensor01(_pil_resize(img, size))

    @staticmethod
    def _box(n, channel, image_size, vmin, val):
        t = torch.full((channel, image_size, image_size), float(vmin))
        g = Backdoor.TRIGGER_GAP_X
        t[:, -(n + g):-g, -(n + g):-g] = val          # dataset.py:768-788: n px box, 2 px from bottom/right
        return t

    def _img_trigger(self, rel, image_size, channel, trigger_sz, vmin, vmax, x=None, y=None):
        """dataset.py:733-761."""
        resid = image_size - trigger_sz
        l = t = int(resid / 2)
        r, b = resid - l, resid - t
        if x is not None:
            l, r = (x, resid - x) if x > 0 else (resid + x, -x)
        if y is not None:
            t, b = (y, resid - y) if y > 0 else (resid + y, -y)
        img = normalize(self._load_rgb(rel, trigger_sz, channel), 0.0, 1.0, vmin, vmax)
        out = torch.nn.functional.pad(img, (l, r, t, b), value=float(vmin))
        out[out >= 0.999] = vmin
        return out

    def get_trigger(self, type: str, channel: int, image_size: int, vmin=DEFAULT_VMIN, vmax=DEFAULT_VMAX) -> torch.Tensor:
        if type in self._WHITE:
            return self._box(self._WHITE[type], channel, image_size, vmin, vmax)
        if type in self._GREY:
            return self._box(self._GREY[type], channel, image_size, vmin, (vmin + vmax) / 2)
        if type in self._STOP:
            return self._img_trigger(self.STOP_SIGN_IMG, image_size, channel, self._STOP[type], vmin, vmax, x=-2, y=-2)
        if type == self.TRIGGER_GLASSES:
            return self._img_trigger(self.GLASSES_IMG, image_size, channel, int(image_size * 0.625), vmin, vmax)
        if type == self.TRIGGER_NONE:
            return torch.full((channel, image_size, image_size), float(vmin))
        if type in self._IDX_TRIGGERS:                                 # dataset.py:791-812
            ds, item, dx, dy = self._IDX_TRIGGERS[type]
            img = normalize(self._idx_item(ds, item, image_size, channel), 0.0, 1.0, vmin, vmax)
            img = self._bg2black(img, vmin, vmax)
            return torch.roll(img, shifts=(0, dy, dx), dims=(0, 1, 2))
        raise ValueError(f"Trigger type {type} isn't found")

    def get_target(self, type: str, trigger: torch.Tensor = None, dx: int = -5, dy: int = -3, vmin=DEFAULT_VMIN,
                   vmax=DEFAULT_VMAX) -> torch.Tensor:
        channel, image_size = trigger.shape[-3], trigger.shape[-1]
        if type == self.TARGET_TG:
            return self._bg2grey(trigger, vmin, vmax)
        if type == self.TARGET_SHIFT:
            return self._bg2grey(torch.roll(trigger, shifts=(dy, dx), dims=(-2, -1)), vmin, vmax)
        if type == self.TARGET_BOX:
            t = torch.full((channel, image_size, image_size), float(vmin))
            t[:, :10, :10] = (vmin + vmax) / 2
            return self._bg2grey(t, vmin, vmax)
        if type in self._IMG_TARGETS:
            img = normalize(self._load_rgb(self._IMG_TARGETS[type], (image_size, image_size), channel), 0.0, 1.0, vmin, vmax)
            return self._bg2grey(img, vmin, vmax)
        if type == self.TARGET_FA:                                     # dataset.py:947-951: FashionMNIST train item 0
            img = normalize(self._idx_item("FashionMNIST", 0, image_size, channel), 0.0, 1.0, vmin, vmax)
            return self._bg2grey(img, vmin, vmax)
        raise NotImplementedError(f"Target type {type} isn't found")


# ------------------------------------------------------------------------------------------------------ PIL restatement of the transform
def _pil_resize(img, size):
    """torchvision.transforms.Resize on a PIL image [UPSTREAM]: bilinear `Image.resize`; an int sizes the SHORT side
    (long side int(size * long / short)), a pair is (h, w)."""
    from PIL import Image
    if isinstance(size, int):
        w, h = img.size
        ow, oh = (size, int(size * h / w)) if w <= h else (int(size * w / h), size)
    else:
        oh, ow = size
    if (ow, oh) == img.size:
        return img
    return img.resize((ow, oh), Image.BILINEAR)


def _pil_to_tensor01(img) -> torch.Tensor:
    """transforms.ToTensor: uint8 HWC -> float CHW / 255."""
    t = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy())
    t = t.permute(2, 0, 1) if t.dim() == 3 else t[None]
    return t.float() / 255.0


def to_training_size(img, size: int, channel: int) -> np.ndarray:
    """The deterministic head of the reference's dataset transform (dataset.py:160-176) for ONE image, kept in uint8:
    Grayscale(1) / convert('RGB') -> Resize([S, S]).  ToTensor + util.normalize + the random flip run in `vd_poison_batch`;
    resizing once at load time is the same function of the pixels as resizing on every fetch."""
    from PIL import Image
    if not isinstance(img, Image.Image):
        a = np.asarray(img, dtype=np.uint8)
        img = Image.fromarray(a[..., 0] if a.ndim == 3 and a.shape[-1] == 1 else a)
    img = _pil_resize(img.convert("RGB" if channel == 3 else "L"), (size, size))
    a = np.asarray(img, dtype=np.uint8)
    return a[..., None] if a.ndim == 2 else a


# ------------------------------------------------------------------------------------------------------ data sources
IMG_EXT = (".png", ".jpg", ".jpeg", ".bmp", ".webp", ".ppm", ".pgm", ".tif", ".tiff")


def read_idx_images(path: str) -> np.ndarray:
    """An idx3-ubyte file (MNIST family), optionally gzip-compressed -> uint8 [N, H, W]."""
    import gzip
    import struct
    with (gzip.open(path, "rb") if path.endswith(".gz") else open(path, "rb")) as f:
        raw = f.read()
    magic, n, h, w = struct.unpack(">IIII", raw[:16])
    if magic != 2051 or len(raw) != 16 + n * h * w:
        raise ValueError(f"{path}: not an idx3-ubyte image file (magic {magic}, {len(raw)} bytes for {n}x{h}x{w})")
    return np.frombuffer(raw, dtype=np.uint8, offset=16).reshape(n, h, w)


def read_idx_labels(path: str) -> np.ndarray:
    import gzip
    import struct
    with (gzip.open(path, "rb") if path.endswith(".gz") else open(path, "rb")) as f:
        raw = f.read()
    magic, n = struct.unpack(">II", raw[:8])
    if magic != 2049 or len(raw) != 8 + n:
        raise ValueError(f"{path}: not an idx1-ubyte label file")
    return np.frombuffer(raw, dtype=np.uint8, offset=8).astype(np.int64)


def locate_idx(root: Optional[str], ds: str, stem: str) -> str:
    roots = [r for r in (root, "datasets", ".") if r]
    for r in roots:
        for sub in (os.path.join(ds, "raw"), ds, ds.lower(), ""):
            for ext in ("", ".gz"):
                p = os.path.join(r, sub, stem + ext)
                if os.path.exists(p):
                    return p
    raise FileNotFoundError(
        f"{ds}: {stem}[.gz] not found under {roots} (torchvision layout <root>/{ds}/raw/). The reference downloads it "
        f"(dataset.py:791-812, 947-951); there is no network here: place the idx files locally.")


def _load_mnist(root: Optional[str]) -> Tuple[np.ndarray, np.ndarray]:
    """train + test (70 000), the reference's 'train+test' split (dataset.py:111-114), from local idx files."""
    xs = [read_idx_images(locate_idx(root, "MNIST", f"{p}-images-idx3-ubyte")) for p in ("train", "t10k")]
    ys = [read_idx_labels(locate_idx(root, "MNIST", f"{p}-labels-idx1-ubyte")) for p in ("train", "t10k")]
    return np.concatenate(xs)[..., None], np.concatenate(ys)


FOLDER_NAMES = {"CELEBA-HQ": ("celeba_hq_256", "celeba_hq", "CelebA-HQ", "CELEBA-HQ"), "CELEBA": ("celeba", "celebA", "CelebA", "CELEBA")}


def locate_image_source(name: str, root: Optional[str]) -> str:
    """Local stand-in for the reference's `load_dataset("datasets/celeba_hq_256", split='train')` / `load_dataset("student/celebA")`
    (dataset.py:118-122): a directory of image files (any nesting, sorted by path like HF's imagefolder builder) or an .npz / .npy
    holding a uint8 [N, H, W, C] array under `images`."""
    roots = [r for r in (root, "datasets", ".") if r]
    for r in roots:
        for sub in FOLDER_NAMES[name]:
            for cand in (os.path.join(r, sub), os.path.join(r, sub + ".npz"), os.path.join(r, sub + ".npy")):
                if os.path.isdir(cand) or os.path.isfile(cand):
                    return cand
    raise FileNotFoundError(
        f"{name}: no local copy found (looked for {FOLDER_NAMES[name]} as an image folder / .npz / .npy under {roots}). The "
        f"reference reads it from disk or the hub (dataset.py:118-122); there is no network here.")


def _load_image_source(src: str, size: int, channel: int, workers: int = 0) -> np.ndarray:
    """All images of `src` at the training size, uint8 [N, S, S, C] (decode + convert + resize on a thread pool: PIL releases the GIL)."""
    if os.path.isfile(src):
        arr = np.load(src)
        arr = arr["images"] if hasattr(arr, "files") else arr
        arr = np.asarray(arr, dtype=np.uint8)
        if arr.ndim == 3:
            arr = arr[..., None]
        if arr.shape[1] == size and arr.shape[2] == size and arr.shape[3] == channel:
            return np.ascontiguousarray(arr)
        return np.stack([to_training_size(a, size, channel) for a in arr])
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    files = sorted(os.path.join(d, f) for d, _, fs in os.walk(src) for f in fs if f.lower().endswith(IMG_EXT))
    if not files:
        raise FileNotFoundError(f"{src}: no image files ({', '.join(IMG_EXT)})")
    out = np.empty((len(files), size, size, channel), dtype=np.uint8)

    def one(i):
        with Image.open(files[i]) as im:
            out[i] = to_training_size(im, size, channel)

    with ThreadPoolExecutor(max_workers=workers or min(32, (os.cpu_count() or 8))) as ex:
        list(ex.map(one, range(len(files))))
    return out


def synthetic_images(n: int = 60000, size: int = 32, channel: int = 3, seed: int = 0) -> np.ndarray:
    """SURVEY §8d synthetic inputs: i.i.d. uniform uint8, numpy.random.default_rng(seed)."""
    return np.random.default_rng(seed).integers(0, 256, size=(n, size, size, channel), dtype=np.uint8)


def _load_cifar10(root: str) -> Tuple[np.ndarray, np.ndarray]:
    """train+test (60 000) in the reference's order (dataset.py:113-117), from the standard python pickles or an npz."""
    for cand in (root, os.path.join(root, "cifar10"), "datasets", "."):
        npz = os.path.join(cand, "cifar10.npz")
        if os.path.exists(npz):
            d = np.load(npz)
            return d["images"].astype(np.uint8), d["labels"].astype(np.int64)
        py = os.path.join(cand, "cifar-10-batches-py")
        if os.path.isdir(py):
            xs, ys = [], []
            for f in [f"data_batch_{i}" for i in range(1, 6)] + ["test_batch"]:
                with open(os.path.join(py, f), "rb") as fh:
                    d = pickle.load(fh, encoding="bytes")
                xs.append(np.asarray(d[b"data"], dtype=np.uint8).reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1))
                ys.append(np.asarray(d[b"labels"], dtype=np.int64))
            return np.concatenate(xs), np.concatenate(ys)
    raise FileNotFoundError(
        "CIFAR10 not found locally (no network here): put cifar-10-batches-py/ or cifar10.npz under the dataset root, "
        "or use --dataset SYNTHETIC-CIFAR10")


class DatasetLoader:
    MODE_FIXED, MODE_FLEX, MODE_NONE, MODE_EXTEND = "FIXED", "FLEX", "NONE", "EXTEND"
    MNIST, CIFAR10, CELEBA, LSUN_CHURCH, LSUN_BEDROOM, CELEBA_HQ = "MNIST", "CIFAR10", "CELEBA", "LSUN-CHURCH", "LSUN-BEDROOM", "CELEBA-HQ"
    CELEBA_HQ_LATENT_PR05, CELEBA_HQ_LATENT = "CELEBA-HQ-LATENT_PR05", "CELEBA-HQ-LATENT"
    SYNTHETIC_CIFAR10 = "SYNTHETIC-CIFAR10"
    SYNTHETIC_CELEBA_HQ = "SYNTHETIC-CELEBA-HQ"            # 256x256 stand-in for BASELINE config #4 (no dataset download here)
    INPAINT_BOX, INPAINT_LINE = "INPAINT_BOX", "INPAINT_LINE"
    TRAIN, TEST = "train", "test"
    PIXEL_VALUES, PIXEL_VALUES_TRIGGER, TRIGGER, TARGET = "pixel_values", "pixel_values_trigger", "trigger", "target"
    IS_CLEAN, R_trigger_only, IMAGE, LABEL = "is_clean", "R_trigger_only", "image", "label"

    def __init__(self, name: str, label: int = None, root: str = None, channel: int = None, image_size: int = None,
                 vmin: Union[int, float] = DEFAULT_VMIN, vmax: Union[int, float] = DEFAULT_VMAX, batch_size: int = 512,
                 shuffle: bool = True, seed: int = 0, device=None, images: Optional[np.ndarray] = None,
                 labels: Optional[np.ndarray] = None):
        self._root, self._name = root, name
        self._latent = None
        self._label = None if label is None else (list(label) if isinstance(label, (list, tuple)) else [label])
        self._vmin, self._vmax = float(vmin), float(vmax)
        self._batch_size, self._shuffle, self._seed = batch_size, shuffle, seed
        if device is not None:
            self._dev = torch.device(device)
        else:
            self._dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        if images is not None:
            self._images, self._labels = images, labels
        elif name == self.SYNTHETIC_CIFAR10:
            self._images, self._labels = synthetic_images(), None
        elif name == self.SYNTHETIC_CELEBA_HQ:
            self._images, self._labels = synthetic_images(n=int(os.environ.get("VILLAN_SYNTHETIC_N", 2048)), size=256), None
        elif name == self.CIFAR10:
            self._images, self._labels = _load_cifar10(root or "datasets")
        elif name == self.MNIST:
            self._images, self._labels = _load_mnist(root)
            channel, image_size = channel or 1, image_size or 32             # dataset.py:131-149
        elif name in (self.CELEBA_HQ, self.CELEBA):
            channel, image_size = channel or 3, image_size or (256 if name == self.CELEBA_HQ else 64)
            self._images, self._labels = _load_image_source(locate_image_source(name, root), image_size, channel), None
        elif name == self.CELEBA_HQ_LATENT:
            # precomputed VQ-VAE latents (dataset.py:125-126, 440-442): <root>/celeba_hq_256_latents in LatentDataset format;
            # trigger / target stay IMAGE-space tensors (256x256) -- the pipeline encodes them when it needs latents
            self._latent = LatentDataset(os.path.join(root or "datasets", "celeba_hq_256_latents"))
            self._images, self._labels = None, None
            channel, image_size = channel or 3, image_size or 256
        elif name == self.CELEBA_HQ_LATENT_PR05:
            raise NotImplementedError(f"{name}: the reference itself cannot load it (dataset.py:124 calls an un-imported load_from_disk)")
        else:                                                              # LSUN-* included: dataset.py:109-128 has no loader for them
            raise NotImplementedError(f"Undefined dataset: {name}")
        self._channel = channel if channel is not None else self._images.shape[-1]
        self._image_size = image_size if image_size is not None else self._images.shape[1]
        if self._images is not None and (self._images.shape[1] != self._image_size or self._images.shape[2] != self._image_size
                                         or self._images.shape[-1] != self._channel):
            # dataset.py:160-176: channel conversion + Resize([S, S]) (PIL bilinear), applied once here instead of per fetch
            self._images = np.stack([to_training_size(a, self._image_size, self._channel) for a in self._images])
        self._backdoor = Backdoor(root=root)
        self._trigger = self._target = None
        self._trigger_type = self._target_type = None
        self._clean_rate, self._poison_rate, self._ext_poison_rate = 1.0, 0.0, 0.0
        self._rng = np.random.default_rng(seed)
        self._index = self._flags = None
        self._dev_images = None
        self._R_trigger_only = False
        self.random_flip = True            # dataset.py:165-168: RandomHorizontalFlip is always on

    # ---- reference surface ----
    def set_poison(self, trigger_type: str, target_type: str, target_dx: int = -5, target_dy: int = -3, clean_rate: float = 1.0,
                   poison_rate: float = 0.2, ext_poison_rate: float = 0.0) -> "DatasetLoader":
        if self._root is None:
            raise ValueError("Attribute 'root' is None")
        self._clean_rate, self._poison_rate, self._ext_poison_rate = clean_rate, poison_rate, ext_poison_rate
        self._trigger_type, self._target_type = trigger_type, target_type
        self._trigger = self._backdoor.get_trigger(trigger_type, self._channel, self._image_size, self._vmin, self._vmax)
        self._target = self._backdoor.get_target(target_type, self._trigger, dx=target_dx, dy=target_dy, vmin=self._vmin, vmax=self._vmax)
        return self

    def _subset(self, n_total: int, rate: float) -> np.ndarray:
        """HF train_test_split(test_size=int(n*rate)).test restated as a seeded random subset (the reference's split is
        unseeded, dataset.py:216-231: only the sizes are pinned)."""
        rate = float(rate)
        if rate == 0.0:
            return np.zeros(0, dtype=np.int64)
        if rate == 1.0:
            return np.arange(n_total, dtype=np.int64)
        if rate > 1.0:
            mul, mod = int(rate // 1), float(rate - int(rate // 1))
            parts = [np.arange(n_total, dtype=np.int64) for _ in range(mul)]
            if mod > 0:
                parts.append(self._subset(n_total, mod))
            return np.concatenate(parts)
        return self._rng.permutation(n_total)[: int(n_total * rate)].astype(np.int64)

    @classmethod
    def check_available(cls, name: str, root: Optional[str]) -> None:
        """Raise (FileNotFoundError / NotImplementedError) if `name` cannot be loaded from local files -- the driver calls this
        before it creates the run directory."""
        if name in (cls.SYNTHETIC_CIFAR10, cls.SYNTHETIC_CELEBA_HQ, cls.CELEBA_HQ_LATENT):
            return
        if name == cls.CIFAR10:
            for cand in (root or "datasets", os.path.join(root or "datasets", "cifar10"), "datasets", "."):
                if os.path.exists(os.path.join(cand, "cifar10.npz")) or os.path.isdir(os.path.join(cand, "cifar-10-batches-py")):
                    return
            raise FileNotFoundError("CIFAR10 not found locally (no network here): put cifar-10-batches-py/ or cifar10.npz under the "
                                    "dataset root, or use --dataset SYNTHETIC-CIFAR10")
        if name == cls.MNIST:
            locate_idx(root, "MNIST", "train-images-idx3-ubyte")
            return
        if name in FOLDER_NAMES:
            locate_image_source(name, root)
            return
        raise NotImplementedError(f"Undefined dataset: {name}")

    def prepare_dataset(self, mode: str = "FIXED", R_trigger_only: bool = False, ext_R_trigger_only: bool = False,
                        R_gaussian_aug: float = 0.0) -> "DatasetLoader":
        # R_gaussian_aug: stored and never read by the reference either (dataset.py:419-422 is its only use)
        self._R_trigger_only = bool(R_trigger_only)
        if self._latent is not None:
            return self._prepare_latent()
        base = np.arange(len(self._images), dtype=np.int64)
        if self._label is not None:
            if self._labels is None:
                raise ValueError("label filter requested but the dataset has no labels")
            base = base[np.isin(self._labels, self._label)]
        n = len(base)
        pr = float(self._poison_rate)
        POISON, RTO = 1, 4
        if mode == self.MODE_FIXED:                                    # dataset.py:215-260
            if pr < 0 or pr > 1:
                raise ValueError(f"In {self.MODE_FIXED}, poison rate should <= 1.0 and >= 0.0")
            k = int(n * pr)
            perm = self._rng.permutation(n) if 0.0 < pr < 1.0 else np.arange(n)
            if pr == 1.0:
                k = n
            idx = np.concatenate([perm[: n - k], perm[n - k:]])
            flags = np.concatenate([np.zeros(n - k, np.uint8), np.full(k, POISON | (RTO if R_trigger_only else 0), np.uint8)])
        elif mode == self.MODE_FLEX:                                   # dataset.py:288-334
            c, p = self._subset(n, self._clean_rate), self._subset(n, pr)
            idx = np.concatenate([c, p])
            flags = np.concatenate([np.zeros(len(c), np.uint8), np.full(len(p), POISON | (RTO if R_trigger_only else 0), np.uint8)])
        elif mode == self.MODE_EXTEND:                                 # dataset.py:336-417
            er = float(self._ext_poison_rate)
            perm = self._rng.permutation(n) if 0.0 < er < 1.0 else np.arange(n)
            ke = n if er == 1.0 else int(n * er)
            c, e, p = perm[: n - ke], perm[n - ke:], self._subset(n, pr)
            idx = np.concatenate([c, e, p])
            flags = np.concatenate([np.zeros(len(c), np.uint8),
                                    np.full(len(e), POISON | (RTO if ext_R_trigger_only else 0), np.uint8),
                                    np.full(len(p), POISON | (RTO if R_trigger_only else 0), np.uint8)])
        elif mode == self.MODE_NONE:
            idx, flags = np.arange(n), np.zeros(n, np.uint8)
        else:
            raise NotImplementedError(f"Argument mode: {mode} isn't defined")
        self._index, self._flags = base[idx], flags
        self._dev_images = None                                # device copies (images, partition, trigger / target) are rebuilt on first use
        return self

    def _prepare_latent(self) -> "DatasetLoader":
        """dataset.py:440-442: the latent dataset poisons BY INDEX (first int(len * poison_rate) items), whatever the mode."""
        lds = self._latent.set_poison(target_key=self._target_type, poison_key=self._trigger_type, raw=LatentDataset.RAW_LATENTS_FILE_NAME,
                                      poison_rate=self._poison_rate, use_latent=True)
        lds.set_use_names(target=self.TARGET, poison=self.PIXEL_VALUES, raw=self.IMAGE)
        n = len(lds)
        if n == 0:
            raise FileNotFoundError(f"no latents under {lds._root}/raw: create them with make_latent_dataset.py")
        k = int(n * float(self._poison_rate))
        self._lat_raw = torch.stack([lds.get_data_latent_by_idx(LatentDataset.RAW_LATENTS_FILE_NAME, i) for i in range(n)]).float()
        self._lat_poison = (torch.stack([lds.get_data_latent_by_idx(self._trigger_type, i) for i in range(k)]).float()
                            if k > 0 else self._lat_raw[:0])
        self._lat_target = lds.get_target_latent().float()
        self._index = np.arange(n, dtype=np.int64)
        self._flags = (np.arange(n) < k).astype(np.uint8)
        self._lat_dev = None
        return self

    def _make_latent_batch(self, sample_ids: torch.Tensor, full: bool) -> Dict[str, torch.Tensor]:
        if self._lat_dev is None:                       # latents resident on the device: a batch is an index gather
            self._lat_dev = (self._lat_raw.to(self._dev), self._lat_poison.to(self._dev), self._lat_target.to(self._dev),
                             torch.from_numpy(self._flags.astype(bool)).to(self._dev))
        raw, poi, tgt, fl = self._lat_dev
        idx = self._h2d(sample_ids.long())
        is_p = fl[idx]
        r = raw[idx]
        m = is_p[:, None, None, None]
        pv = torch.zeros_like(r)
        if poi.shape[0] > 0:
            pv = torch.where(m, poi[idx.clamp(max=poi.shape[0] - 1)], pv)
        batch = {self.PIXEL_VALUES: pv.contiguous(), self.TARGET: torch.where(m, tgt[None].expand_as(r), r).contiguous(), self.IMAGE: r}
        if full:
            batch[self.IS_CLEAN] = ~is_p
            batch[self.LABEL] = torch.full((len(idx),), -1.0, device=self._dev)
        return batch

    # ---- properties ----
    @property
    def trigger(self):
        return self._trigger

    @property
    def target(self):
        return self._target

    @property
    def image_size(self):
        return self._image_size

    @property
    def channel(self):
        return self._channel

    @property
    def name(self):
        return self._name

    @property
    def root(self):
        return self._root

    @property
    def batch_size(self):
        return self._batch_size

    @property
    def num_batch(self):
        return int(np.ceil(len(self) / self._batch_size))

    def __len__(self):
        return len(self._index) if self._index is not None else len(self._images)

    def get_mask(self, trigger: torch.Tensor) -> torch.Tensor:
        return torch.where(trigger > self._vmin, 0, 1)

    def get_poisoned(self, imgs: torch.Tensor) -> torch.Tensor:
        trig = self._trigger.to(imgs.device)
        m = self.get_mask(trig)
        return m * imgs + (1 - m) * trig

    def get_inpainted_by_type(self, imgs: torch.Tensor, inpaint_type: str) -> torch.Tensor:
        half = imgs.shape[-1] // 2
        mask = torch.ones_like(imgs[0])
        if inpaint_type == self.INPAINT_LINE:
            mask[..., 0:2 * half, half - half // 10: half + half // 20] = 0
        elif inpaint_type == self.INPAINT_BOX:
            lo, hi = half - half // 3, half + half // 3
            mask[..., lo:hi, lo:hi] = 0
        else:
            raise NotImplementedError(f"inpaint: {inpaint_type} is not implemented")
        return mask * imgs + (1 - mask) * torch.full_like(imgs, float(imgs.min()))

    # ---- GPU batch production ----
    def _ensure_device(self):
        if self._dev_images is None:
            self._dev_images = torch.from_numpy(self._images).to(self._dev)
            self._dev_trigger = self._trigger.to(self._dev).contiguous()
            self._dev_target = self._target.to(self._dev).contiguous()
            # the partition itself is resident too: a batch addressed by DEVICE positions needs no host -> device copy at all
            self._dev_index = torch.from_numpy(np.ascontiguousarray(self._index, dtype=np.int64)).to(self._dev)
            self._dev_flags = torch.from_numpy(np.ascontiguousarray(self._flags, dtype=np.uint8)).to(self._dev)
            self._any_rto = bool((self._flags & 4).any())
            self._flip_gen = torch.Generator(device=self._dev).manual_seed(int(self._seed) + 7919) if self._dev.type == "cuda" else None

    def loader_state(self) -> Dict:
        """What a resumed run needs to draw the same random flips: the device flip generator's state (None before the first device batch)."""
        g = getattr(self, "_flip_gen", None)
        return {"flip_gen": None if g is None else g.get_state().cpu()}

    def load_loader_state(self, st: Optional[Dict]):
        if not st or st.get("flip_gen") is None or self._latent is not None:
            return
        self._ensure_device()
        if self._flip_gen is not None:
            self._flip_gen.set_state(st["flip_gen"].cpu())

    def _h2d(self, t: torch.Tensor) -> torch.Tensor:
        """Small per-batch host arrays (indices, flags) go up from PINNED memory without blocking: a pageable copy makes the host wait for
        everything already queued on the stream, i.e. one full drain per training step (measured: 0.5 ms of idle GPU at every step start)."""
        if self._dev.type == "cuda" and os.environ.get("VILLAN_PAGEABLE_H2D", "0") != "1":      # (1: the blocking copy, for A/B measurements)
            return t.contiguous().pin_memory().to(self._dev, non_blocking=True)
        return t.to(self._dev)

    def make_batch(self, sample_ids: torch.Tensor, flip_bits: Optional[torch.Tensor] = None, full: bool = True) -> Dict[str, torch.Tensor]:
        """Batch dict for positions `sample_ids` of the prepared (partitioned) dataset."""
        if self._latent is not None:
            return self._make_latent_batch(sample_ids, full)
        self._ensure_device()
        if sample_ids.device.type == "cuda" and not full and not self._any_rto:
            return self._make_batch_resident(sample_ids, flip_bits)
        pos = sample_ids.cpu().numpy()
        ds_idx = self._h2d(torch.from_numpy(self._index[pos]))
        flags = torch.from_numpy(self._flags[pos].copy())
        B = len(pos)
        if flip_bits is None:
            flip_bits = (torch.rand(B) < 0.5) if self.random_flip else torch.zeros(B, dtype=torch.bool)
        rto = bool((flags & 4).any())
        if rto and not bool(((flags & 1) == 0).logical_or((flags & 4) != 0).all()):
            raise NotImplementedError("mixed R_trigger_only flags inside one batch")
        kflags = self._h2d(((flags & 1) | (flip_bits.to(torch.uint8) << 1)).to(torch.uint8))
        C, S = self._channel, self._image_size
        pv, tg, im = (torch.empty((B, C, S, S), device=self._dev, dtype=torch.float32) for _ in range(3))
        ops.poison_batch(self._dev_images, kflags, self._dev_trigger, self._dev_target, pv, tg, im, self._vmin, self._vmax,
                         R_trigger_only=rto, idx=ds_idx.contiguous())
        batch = {self.PIXEL_VALUES: pv, self.TARGET: tg, self.IMAGE: im}
        if full:
            is_p = self._h2d((flags & 1).bool())
            trig = self._dev_trigger[None].expand(B, C, S, S)
            batch[self.TRIGGER] = trig
            batch[self.PIXEL_VALUES_TRIGGER] = torch.where(is_p[:, None, None, None], trig, torch.zeros((), device=self._dev))
            lab = torch.full((B,), -1.0) if self._labels is None else torch.from_numpy(self._labels[self._index[pos]]).float()
            batch[self.LABEL] = self._h2d(lab)
            batch[self.IS_CLEAN] = ~is_p
        return batch

    def _make_batch_resident(self, pos: torch.Tensor, flip_bits: Optional[torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Training batch (full=False) for DEVICE-resident positions: dataset indices, poison flags and flip bits are gathered / drawn on the
        GPU, so a steady-state step issues no host -> device copy (each one costs ~50-100 us of idle GPU in the runtime's copy path)."""
        B = len(pos)
        pos = pos.long()
        ds_idx = self._dev_index.index_select(0, pos)
        fl = self._dev_flags.index_select(0, pos)
        if flip_bits is None:
            flip = (torch.rand(B, device=self._dev, generator=self._flip_gen) < 0.5) if self.random_flip else None
        else:
            flip = flip_bits if flip_bits.device.type == "cuda" else self._h2d(flip_bits)
        kflags = fl & 1
        if flip is not None:
            kflags = kflags | (flip.to(torch.uint8) << 1)
        C, S = self._channel, self._image_size
        pv, tg, im = (torch.empty((B, C, S, S), device=self._dev, dtype=torch.float32) for _ in range(3))
        ops.poison_batch(self._dev_images, kflags.contiguous(), self._dev_trigger, self._dev_target, pv, tg, im, self._vmin, self._vmax,
                         R_trigger_only=False, idx=ds_idx)
        return {self.PIXEL_VALUES: pv, self.TARGET: tg, self.IMAGE: im}

    def get_dataloader(self, batch_size: int = None, shuffle: bool = None, num_workers: int = None, collate_fn=None,
                       rank: int = 0, world: int = 1, epoch: int = 0, full: bool = True):
        bs = batch_size or self._batch_size
        shuffle = self._shuffle if shuffle is None else shuffle
        return _Loader(self, bs, shuffle, rank, world, epoch, full)

    def get_dataset(self):
        return self


class _Loader:
    def __init__(self, dsl: DatasetLoader, bs: int, shuffle: bool, rank: int, world: int, epoch: int, full: bool):
        from .trainer import shard_indices
        self.dsl, self.bs, self.full = dsl, bs, full
        self.ids = shard_indices(len(dsl), epoch, rank, world, seed=dsl._seed, shuffle=shuffle)
        # training batches (full=False) of the pixel datasets are addressed by device positions: uploaded once per epoch, not per batch
        self._ids_dev = self.ids.to(dsl._dev) if (not full and dsl._latent is None and dsl._dev.type == "cuda") else None

    def __len__(self):
        return (len(self.ids) + self.bs - 1) // self.bs

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        ids = self.ids if self._ids_dev is None else self._ids_dev
        for s in range(0, len(ids), self.bs):
            yield self.dsl.make_batch(ids[s:s + self.bs], full=self.full)


class LatentDataset(torch.utils.data.Dataset):
    """Precomputed VQ-VAE latents on disk, the reference's format (dataset.py:1037-1371):

        <root>/target..pt            dict  {target key -> latent}          (note the DOUBLE dot: add_ext() joins with '.' and
        <root>/<data type>/<idx>..pt one latent tensor per sample           DATA_EXT already starts with one, dataset.py:1038,1065)

    ``set_poison(target_key, poison_key, raw, poison_rate)``: item i is poisoned iff ``i < int(len * poison_rate)``
    (dataset.py:1352-1359): target = the target latent, poison = the stored poisoned latent; clean items: target = raw
    latent, poison = zeros.  With ``use_latent=False`` items are decoded through the VQ-VAE set by ``set_vae``."""
    DATA_EXT: str = ".pt"
    TARGET_LATENTS_FILE_NAME: str = "target"
    POISON_LATENTS_FILE_NAME: str = "poison"
    RAW_LATENTS_FILE_NAME: str = "raw"

    def __init__(self, ds_root: str):
        os.makedirs(ds_root, exist_ok=True)
        self._root = ds_root
        self._target = self._poison = self._raw = None
        self._target_name = self._poison_name = self._raw_name = None
        self._poison_rate, self._len, self._vae, self._use_latent = None, None, None, True

    def set_vae(self, vae):
        self._vae = vae
        return self

    @staticmethod
    def add_ext(p: str) -> str:
        return f"{p}.{LatentDataset.DATA_EXT}"

    # ---- raw file access ----
    @staticmethod
    def read_ext(file: str):
        try:
            return torch.load(LatentDataset.add_ext(file))
        except (FileNotFoundError, OSError):
            return None

    @staticmethod
    def save_ext(val, file: str) -> None:
        torch.save(val, LatentDataset.add_ext(file))

    @property
    def targe_latents_path(self) -> str:      # (sic) reference spelling, dataset.py:1070
        p = os.path.join(self._root, LatentDataset.TARGET_LATENTS_FILE_NAME)
        if not os.path.exists(LatentDataset.add_ext(p)):
            LatentDataset.save_ext({}, p)
        return p

    def _dir(self, data_type: str) -> str:
        p = os.path.join(self._root, data_type)
        os.makedirs(p, exist_ok=True)
        return p

    def _encode(self, x: torch.Tensor) -> torch.Tensor:
        if self._vae is None:
            raise ValueError("Please provide encoder first")
        with torch.no_grad():
            return self._vae.encode(x.to(self._vae.device)).latents.detach().cpu()

    def _decode(self, z: torch.Tensor) -> torch.Tensor:
        if self._vae is None:
            raise ValueError("Please provide encoder first")
        with torch.no_grad():
            return self._vae.decode(z.to(self._vae.device)).sample.clone().detach().cpu()

    # ---- targets: one dict file ----
    def update_target_latent_by_key(self, key: str, val: torch.Tensor):
        res = LatentDataset.read_ext(self.targe_latents_path) or {}
        res[key] = val
        LatentDataset.save_ext(res, self.targe_latents_path)

    def update_target_by_key(self, key: str, val: torch.Tensor):
        self.update_target_latent_by_key(key, self._encode(val.unsqueeze(0)).squeeze(0))

    def get_target_latent_by_key(self, key: str) -> torch.Tensor:
        return LatentDataset.read_ext(self.targe_latents_path)[key]

    def get_target_by_key(self, key: str) -> torch.Tensor:
        return self._decode(self.get_target_latent_by_key(key).unsqueeze(0)).squeeze(0)

    # ---- per-sample latents: one file per index ----
    def update_data_latent_by_idx(self, data_type: str, idx: int, val: torch.Tensor):
        LatentDataset.save_ext(val, os.path.join(self._dir(data_type), f"{idx}"))

    def update_data_latents_by_idxs(self, data_type: str, idxs, vals):
        if isinstance(idxs, int):
            idxs, vals = [idxs], (vals.unsqueeze(0) if isinstance(vals, torch.Tensor) else vals)
        elif isinstance(vals, list):
            vals = torch.stack(vals)
        for idx, val in zip(idxs, vals):
            self.update_data_latent_by_idx(data_type, idx, val.clone())

    def update_data_by_idxs(self, data_type: str, idxs, vals):
        if isinstance(idxs, int):
            idxs, vals = [idxs], vals.unsqueeze(0)
        elif isinstance(vals, list):
            vals = torch.stack(vals)
        self.update_data_latents_by_idxs(data_type, list(idxs), self._encode(vals))

    def get_data_latent_by_idx(self, data_type: str, idx: int) -> torch.Tensor:
        return LatentDataset.read_ext(os.path.join(self._dir(data_type), f"{idx}"))

    def get_data_by_idx(self, data_type: str, idx: int) -> torch.Tensor:
        return self._decode(self.get_data_latent_by_idx(data_type, idx).unsqueeze(0)).squeeze(0)

    # ---- poisoning view ----
    def set_poison(self, target_key: str, poison_key: str, raw: str, poison_rate: float, use_latent: bool = True):
        import glob
        self._target, self._poison, self._raw = target_key, poison_key, raw
        self._poison_rate, self._use_latent = poison_rate, use_latent
        self._len = len(glob.glob(os.path.join(self._dir(raw), f"*{LatentDataset.DATA_EXT}")))
        return self

    def set_use_names(self, target: str, poison: str, raw: str):
        self._target_name, self._poison_name, self._raw_name = target, poison, raw
        return self

    def get_target_latent(self) -> torch.Tensor:
        if self._target is None:
            raise ValueError("Please set up the target first")
        return self.get_target_latent_by_key(self._target)

    def __len__(self):
        return self._len

    def __getitem__(self, i: int):
        i = i % len(self)
        get = self.get_data_latent_by_idx if self._use_latent else self.get_data_by_idx
        raw = get(self._raw, i)
        if i < int(self._len * self._poison_rate):
            tgt, poi = self.get_target_latent(), get(self._poison, i)       # reference returns the LATENT target in both modes
        else:
            tgt, poi = raw, torch.zeros_like(raw)
        return {self._target_name: tgt, self._poison_name: poi, self._raw_name: raw}
