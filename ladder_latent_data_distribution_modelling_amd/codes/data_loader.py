"""Input side of the training path (reference: codes/data_loader.py:7-61, codes/models.py:26-44,346-386).

The reference downloads MNIST through tf.keras and reads CelebA from TFRecords; this build has no
network, so:
  * MNIST / Fashion-MNIST are read from a keras-format `mnist.npz` / `fashion-mnist.npz`
    (x_train, y_train, x_test, y_test) found in config['data_path'] or ~/.keras/datasets;
  * CelebA is read from `celebA_{train,val,test}.tfrecords` (tf.Example with one bytes feature 'X' =
    H*W*C raw uint8, models.py:354-371) by a small TFRecord/protobuf reader that needs no TensorFlow;
  * if the files are absent a seeded SYNTHETIC data set of the same shape is used and
    `DataGenerator.synthetic` is True (benchmarks and plumbing tests).
"""
import os
import struct

import numpy as np


def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _fields(buf):
    pos, n = 0, len(buf)
    while pos < n:
        tag, pos = _varint(buf, pos)
        wt = tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v, pos = buf[pos:pos + 4], pos + 4
        elif wt == 1:
            v, pos = buf[pos:pos + 8], pos + 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield tag >> 3, v


def parse_example_bytes(record, key=b"X"):
    """tf.Example -> bytes of the bytes_list feature `key` (Example.features=1, Features.feature=1 (map entry:
    key=1,value=2), Feature.bytes_list=1, BytesList.value=1)."""
    for f, feats in _fields(record):
        if f != 1:
            continue
        for f2, entry in _fields(feats):
            if f2 != 1:
                continue
            k = v = None
            for f3, x in _fields(entry):
                if f3 == 1:
                    k = bytes(x)
                elif f3 == 2:
                    v = x
            if k == key and v is not None:
                for f4, bl in _fields(v):
                    if f4 == 1:
                        for f5, raw in _fields(bl):
                            if f5 == 1:
                                return bytes(raw)
    raise KeyError("feature %r not found" % key)


def tfrecord_iterator(path, rank=0, world=1):
    """Yield the payload of each record of a TFRecord file (length:u64, crc:u32, data, crc:u32).  With world > 1 only the records
    whose index i satisfies i % world == rank are read; the others are seeked over (a data-parallel rank never touches the bytes of
    another rank's shard)."""
    with open(path, "rb") as f:
        i = 0
        while True:
            head = f.read(12)
            if len(head) < 12:
                return
            (n,) = struct.unpack("<Q", head[:8])
            if i % world == rank:
                data = f.read(n)
                f.seek(4, 1)
                yield data
            else:
                f.seek(n + 4, 1)
            i += 1


def tfrecord_count(path):
    """Number of records, from the framing only (payloads are seeked over)."""
    n_rec = 0
    with open(path, "rb") as f:
        while True:
            head = f.read(12)
            if len(head) < 12:
                return n_rec
            (n,) = struct.unpack("<Q", head[:8])
            f.seek(n + 4, 1)
            n_rec += 1


def write_tfrecord(path, images_uint8):
    """Writer used by the tests (same framing; CRCs are written as zeros, readers here ignore them)."""
    def vint(n):
        out = bytearray()
        while True:
            b = n & 0x7F
            n >>= 7
            out.append(b | (0x80 if n else 0))
            if not n:
                return bytes(out)

    def ld(field, payload):
        return vint((field << 3) | 2) + vint(len(payload)) + payload

    with open(path, "wb") as f:
        for img in images_uint8:
            raw = np.ascontiguousarray(img, np.uint8).tobytes()
            feature = ld(1, ld(1, raw))
            entry = ld(1, b"X") + ld(2, feature)
            ex = ld(1, ld(1, entry))
            f.write(struct.pack("<Q", len(ex)) + b"\0\0\0\0" + ex + b"\0\0\0\0")


class BatchIterator:
    """Epoch-seeded shuffling minibatch iterator with drop_remainder (models.py:33-40)."""

    def __init__(self, images, batch_size, seed=0, shuffle=True):
        self.images, self.bs = images, int(batch_size)
        if self.images.shape[0] < self.bs:
            raise ValueError("data set of %d samples is smaller than one minibatch of %d (drop_remainder leaves no batch)"
                             % (self.images.shape[0], self.bs))
        self.rng = np.random.default_rng(seed)
        self.shuffle = shuffle
        self._order, self._pos = None, 0

    def next(self):
        n = self.images.shape[0]
        if self._order is None or self._pos + self.bs > n:
            self._order = self.rng.permutation(n) if self.shuffle else np.arange(n)
            self._pos = 0
        idx = self._order[self._pos:self._pos + self.bs]
        self._pos += self.bs
        return np.ascontiguousarray(self.images[idx], dtype=np.float32)


class DeviceBatchIterator:
    """BatchIterator over a data set that is RESIDENT IN HBM (MI355X: 288 GB; the CelebA train split is 8.8 GB as uint8).
    Same epoch-seeded permutation and drop_remainder as BatchIterator; `next()` returns a float32 device tensor assembled by
    `ladder_gather_rows` (index gather + uint8 -> float * 1/255 in one pass), so the hot loop does no host work and no
    host->device copy.  `images`: numpy / torch array [n, ...] of uint8 (scaled by 1/255, models.py:361,370) or float32."""

    def __init__(self, images, batch_size, seed=0, shuffle=True, device="cuda:0"):
        import torch
        from .. import _lib as L
        self._torch, self._L = torch, L
        if isinstance(images, torch.Tensor):
            t = images
        else:
            a = np.ascontiguousarray(images)
            if a.dtype != np.uint8:
                a = a.astype(np.float32, copy=False)
            t = torch.as_tensor(a)
        self.data = t.to(device).contiguous()                 # one upload; callers cache and pass the device tensor back in
        if self.data.dtype not in (torch.uint8, torch.float32):
            raise TypeError("device-resident data must be uint8 or float32, got %s" % self.data.dtype)
        self.is_u8 = self.data.dtype == torch.uint8
        self.shape = tuple(self.data.shape[1:])
        self.D = int(np.prod(self.shape))
        self.bs = int(batch_size)
        if self.data.shape[0] < self.bs:
            raise ValueError("data set of %d samples is smaller than one minibatch of %d (drop_remainder leaves no batch)"
                             % (self.data.shape[0], self.bs))
        self.rng = np.random.default_rng(seed)
        self.shuffle = shuffle
        self._order, self._pos = None, 0

    def next(self):
        torch, L = self._torch, self._L
        n = self.data.shape[0]
        if self._order is None or self._pos + self.bs > n:
            order = self.rng.permutation(n) if self.shuffle else np.arange(n)
            self._order = torch.as_tensor(order.astype(np.int64)).to(self.data.device)      # one small upload per epoch
            self._pos = 0
        idx = self._order[self._pos:self._pos + self.bs]
        self._pos += self.bs
        out = torch.empty((self.bs,) + self.shape, dtype=torch.float32, device=self.data.device)
        L.call("ladder_gather_rows", self.data.data_ptr(), 1 if self.is_u8 else 0, idx.data_ptr(), out.data_ptr(), self.bs, self.D,
               (1.0 / 255.0) if self.is_u8 else 1.0, torch.cuda.current_stream(self.data.device).cuda_stream)
        return out


class DataGenerator:
    def __init__(self, config, sess=None):
        self.config, self.sess = config, sess
        self.synthetic = False
        exp = config["exp_name"]
        if exp in ("mnist_digit", "mnist_fashion"):
            self.load_MNIST_dataset("digit" if exp == "mnist_digit" else "fashion")
        elif exp == "celeba":
            self.n_train, self.n_val = 180000, 20000          # data_loader.py:15-17
            self._celeba, self._celeba_u8 = {}, {}
        else:
            raise ValueError("unknown exp_name %r" % exp)

    # ---------------------------------------------------------------- MNIST
    def _find_mnist(self, choice):
        names = ["mnist.npz"] if choice == "digit" else ["fashion-mnist.npz", "fashion_mnist.npz"]
        dirs = [self.config.get("data_path", ""), os.path.expanduser("~/.keras/datasets")]
        for d in dirs:
            for n in names:
                p = os.path.join(d, n) if d else ""
                if p and os.path.isfile(p):
                    return p
        return None

    def load_MNIST_dataset(self, choice):
        path = self._find_mnist(choice)
        bs = int(self.config["batch_size"])
        if path is not None:
            d = np.load(path)
            x_train, y_train, x_test, y_test = d["x_train"], d["y_train"], d["x_test"], d["y_test"]
        else:
            self.synthetic = True
            n_tr, n_te = int(self.config.get("synthetic_n_train", 2048)), int(self.config.get("synthetic_n_val", 512))
            n_tr, n_te = max(n_tr, 2 * bs), max(n_te, 10 * bs)
            rng = np.random.default_rng(0)
            x_train = (rng.random((n_tr, 28, 28)) * 255).astype(np.uint8)
            x_test = (rng.random((n_te, 28, 28)) * 255).astype(np.uint8)
            y_train = rng.integers(0, 10, n_tr).astype(np.uint8)
            y_test = (np.arange(n_te) % 10).astype(np.uint8)
        x_train, x_test = x_train / 255.0, x_test / 255.0
        self.n_train, self.n_val = x_train.shape[0], x_test.shape[0]
        self.train_set = dict(attrib=y_train, image=np.expand_dims(x_train, -1).astype(np.float32))
        self.val_set = dict(attrib=y_test, image=np.expand_dims(x_test, -1).astype(np.float32))
        # class-balanced test batch (data_loader.py:35-56)
        table = {64: (7, 7, 7, 7, 6, 6, 6, 6, 6, 6), 128: (13,) * 8 + (12, 12), 256: (26,) * 6 + (25,) * 4,
                 512: (51,) * 8 + (52, 52)}
        if bs not in table:
            raise ValueError("MNIST batch_size must be one of 64/128/256/512 (reference data_loader.py:37-44), got %d" % bs)
        number = table[bs]
        xs = np.zeros((bs, 28, 28), np.float32)
        ys = np.zeros((bs,), np.uint8)
        count, idx = [0] * 10, 0
        while sum(count) < bs:
            c = int(y_test[idx])
            if count[c] < number[c]:
                slot = sum(number[:c]) + count[c]
                xs[slot], ys[slot] = x_test[idx], c
                count[c] += 1
            idx += 1
        self.test_set = dict(attrib=ys, image=np.expand_dims(xs, -1))
        if choice == "fashion":
            self.class_name = ("top", "trousers", "pullover", "dress", "coat", "sandal", "shirt", "sneaker", "bag", "ankle boot")

    # ---------------------------------------------------------------- CelebA
    def celeba_images_u8(self, split, limit=None, rank=0, world=1):
        """uint8 [n,H,W,C] exactly as stored in celebA_<split>.tfrecords (tf.Example, bytes feature 'X', models.py:354-371);
        seeded synthetic uint8 images when the file is absent.  rank / world select the data-parallel shard (records rank::world):
        the array is allocated once at its final size and only this rank's records are parsed (the full 180 000-image split is
        8.8 GB; a per-image list + np.stack, or a full parse on every rank of an 8-GPU node, would multiply that on the host)."""
        key = (split, limit, rank, world)
        if key in self._celeba_u8:
            return self._celeba_u8[key]
        H, W, Cc = int(self.config["dim_input_x"]), int(self.config["dim_input_y"]), int(self.config["dim_input_channel"])
        path = os.path.join(self.config.get("data_path", ""), "celebA_%s.tfrecords" % split)
        if os.path.isfile(path):
            total = tfrecord_count(path)
            n = len(range(rank, total, world))
            if limit:
                n = min(n, limit)
            arr = np.empty((n, H, W, Cc), np.uint8)
            for i, rec in enumerate(tfrecord_iterator(path, rank, world)):
                if i >= n:
                    break
                arr[i] = np.frombuffer(parse_example_bytes(rec), np.uint8).reshape(H, W, Cc)
        else:
            self.synthetic = True
            bs = int(self.config["batch_size"])
            n = int(self.config.get("synthetic_n_train", 4 * bs)) if split == "train" else 2 * bs
            rng = np.random.default_rng({"train": 0, "val": 1, "test": 2}[split])
            arr = rng.integers(0, 256, (n, H, W, Cc), dtype=np.uint8)
            if split == "train":
                self.n_train = n
            elif split == "val":
                self.n_val = n
            arr = arr[rank::world]
            if limit:
                arr = arr[:limit]
        self._celeba_u8[key] = arr
        return arr

    def celeba_images(self, split, limit=None):
        """float32 [n,H,W,C] in [0,1] = uint8 * 1/255 (models.py:361,370).  The trainers use the uint8 form (HBM-resident,
        normalised on the device); this float view is for small slices (test batch, demos)."""
        key = (split, limit)
        if key not in self._celeba:
            self._celeba[key] = self.celeba_images_u8(split, limit).astype(np.float32) * np.float32(1.0 / 255)
        return self._celeba[key]
