"""Input pipeline of the training step.

The reference feeds its graph synchronously: `minibatches` (model/utils/general.py:15-35) yields Python lists, the
step pads them (`pad_batch_images` utils/image.py:46-62, `pad_batch_formulas` utils/text.py:116-139) and
`sess.run` copies them in, all on the training thread (img2seq.py:160-167).  Here the same host functions run on a
background thread into PINNED staging buffers and cross PCIe on a dedicated copy stream, `depth` batches ahead of
the compute stream; the step receives device tensors plus host-side lengths (so it never synchronises to count
tokens).  `ShardedBuckets` gives every data-parallel rank batches of ONE image shape per step (equal shapes on all
ranks, the reference's `bucket` grouping of data_generator.py:84-122 lifted to the global batch).
"""
import queue
import threading

import numpy as np
import torch

from .model.utils.general import minibatches
from .model.utils.image import pad_batch_images
from .model.utils.text import pad_batch_formulas


class Batch(object):
    """One staged batch: img u8 [B,H,W,1] and formula i32 [B,T] on the device, lengths (numpy, host)."""

    def wait(self, stream=None):
        """Make `stream` (default: the current stream) wait for this batch's host-to-device copies."""
        if self.ready is not None:
            st = stream or torch.cuda.current_stream(self.img.device)
            st.wait_event(self.ready)
            # the tensors were allocated on the copy stream: tell the caching allocator that `st` uses them too, or the
            # block could be handed back to the copy stream (and overwritten by a later upload) while kernels of this
            # step -- conv1's backward re-reads the image -- are still running
            self.img.record_stream(st)
            self.formula.record_stream(st)
        return self


class Prefetcher(object):
    """Iterate over `dataset` in minibatches of `batch_size`, padding + uploading `depth` batches ahead."""

    def __init__(self, dataset, batch_size, id_pad, id_end, device="cuda:0", depth=2, batches=None):
        self.dataset, self.batch_size, self.id_pad, self.id_end = dataset, batch_size, id_pad, id_end
        self.device = torch.device(device)
        self.depth = max(1, int(depth))
        self.batches = batches                      # optional iterable of (imgs, formulas) lists (e.g. ShardedBuckets)
        self.cuda = self.device.type == "cuda"
        self.copy_stream = torch.cuda.Stream(self.device) if self.cuda else None

    def _source(self):
        return self.batches if self.batches is not None else minibatches(self.dataset, self.batch_size)

    def _pinned_slot(self, nbytes_img, nbytes_f):
        """A ring of depth + 2 pinned staging slots, grown on demand and REUSED: a fresh pin_memory() per batch costs a
        hipHostMalloc + page locking of 4 MB every step (measured: 32 ms per step with it, against a 10.8 ms training step)."""
        if not hasattr(self, "_ring"):
            self._ring, self._ring_i = [], 0
        n = self.depth + 2
        if len(self._ring) < n:
            self._ring.append({"img": None, "f": None, "ev": None})
            slot = self._ring[-1]
        else:
            slot = self._ring[self._ring_i % n]
            self._ring_i += 1
            if slot["ev"] is not None:
                slot["ev"].synchronize()               # its previous asynchronous copies have read the buffers
        if slot["img"] is None or slot["img"].numel() < nbytes_img:
            slot["img"] = torch.empty(nbytes_img, dtype=torch.uint8).pin_memory()
        if slot["f"] is None or slot["f"].numel() < nbytes_f:
            slot["f"] = torch.empty(max(nbytes_f, 4096), dtype=torch.uint8).pin_memory()
        return slot

    def _stage(self, imgs, forms):
        img = pad_batch_images(imgs)
        f, l = pad_batch_formulas(forms, self.id_pad, self.id_end)
        f = np.ascontiguousarray(f, dtype=np.int32)
        b = Batch()
        b.lengths, b.n_tokens, b.size, b.ready = np.asarray(l), int(np.sum(l)), len(imgs), None
        if self.cuda:
            slot = self._pinned_slot(img.nbytes, f.nbytes)
            pi = slot["img"][:img.nbytes].view(torch.uint8).reshape(img.shape)
            pf = slot["f"][:f.nbytes].view(torch.int32).reshape(f.shape)
            pi.numpy()[...] = img
            pf.numpy()[...] = f
            with torch.cuda.stream(self.copy_stream):
                b.img = pi.to(self.device, non_blocking=True)
                b.formula = pf.to(self.device, non_blocking=True)
                b.ready = torch.cuda.Event()
                b.ready.record(self.copy_stream)
            slot["ev"] = b.ready
        else:
            b.img, b.formula = torch.from_numpy(img), torch.from_numpy(f)
        return b

    def __iter__(self):
        q = queue.Queue(maxsize=self.depth)
        stop = threading.Event()
        END = object()

        def work():
            try:
                if self.cuda:
                    torch.cuda.set_device(self.device)
                for imgs, forms in self._source():
                    if stop.is_set():
                        return
                    q.put(self._stage(imgs, forms))
                q.put(END)
            except BaseException as e:              # surface loader errors on the training thread
                q.put(e)

        th = threading.Thread(target=work, name="lxo-prefetch", daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is END:
                    break
                if isinstance(item, BaseException):
                    raise item
                yield item.wait() if self.cuda else item
        finally:
            stop.set()
            while th.is_alive():                    # unblock a producer stuck on a full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    th.join(timeout=0.05)


class ShardedBuckets(object):
    """Shape-bucketed batches for `world` data-parallel ranks.

    Examples are grouped by image shape (first-seen order, like DataGenerator.bucket, data_generator.py:84-122);
    every group is cut into GLOBAL batches of world * batch_size examples and rank r takes the r-th slice, so at
    each step all ranks run the same (H, W) and -- except for the ragged tail of a bucket -- the same batch size.
    A tail smaller than `world` examples is dropped on every rank (a rank with no data would desynchronise the
    collectives); `dropped` counts them."""

    def __init__(self, dataset, batch_size, world=1, rank=0, n_steps=None):
        self.dataset, self.batch_size, self.world, self.rank = dataset, int(batch_size), int(world), int(rank)
        self.dropped = 0
        self._n_steps = n_steps            # known from an earlier pass over the same set (len() then costs nothing)

    def _groups(self):
        groups, order = {}, []
        for img, form in self.dataset:
            key = tuple(np.asarray(img).shape)
            if key not in groups:
                groups[key] = []
                order.append(key)
            groups[key].append((img, form))
        return groups, order

    def __len__(self):
        """Optimisation steps one pass takes: sum over shape buckets of ceil(n_bucket / (batch_size * world)), minus the
        dropped tails -- NOT ceil(N / (batch_size * world)): the LR schedule and the global batch counter are scaled by
        this number (train.py, Img2SeqModel._run_train).  One pass over the dataset that keeps only a COUNT per image shape,
        never the images (the reference's DataGenerator.__len__ also iterates, data_generator.py:206-215)."""
        if getattr(self, "_n_steps", None) is None:
            counts, order = {}, []
            for img, _ in self.dataset:
                key = tuple(np.asarray(img).shape)
                if key not in counts:
                    counts[key] = 0
                    order.append(key)
                counts[key] += 1
            gb = self.batch_size * self.world
            self._n_steps = sum(counts[k] // gb + (1 if (counts[k] % gb) >= self.world else 0) for k in order)
        return self._n_steps

    def __iter__(self):
        groups, order = self._groups()
        self.dropped = 0
        gb = self.batch_size * self.world
        for key in order:
            items = groups[key]
            for i in range(0, len(items), gb):
                chunk = items[i:i + gb]
                if len(chunk) < self.world:
                    self.dropped += len(chunk)
                    continue
                base, rem = divmod(len(chunk), self.world)          # balanced: the first `rem` ranks take one more
                lo = self.rank * base + min(self.rank, rem)
                mine = chunk[lo:lo + base + (1 if self.rank < rem else 0)]
                yield [m[0] for m in mine], [m[1] for m in mine]
