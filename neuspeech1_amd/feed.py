"""On-GPU data feed: raw recordings -> the conv GEMM's fp16 operand without a host-side float pass.

The reference produces a batch on the host (utils/reader.py:253-303 `_get_list_data`, :496-516 `padding_sample` /
`pad_sample_ch`; utils/data_utils.py:191-193 collator): np.load (float64) -> channel slice -> zero-pad channels ->
crop / zero-pad time -> float32 tensor (B, ch, 6000), which the device then casts to fp16 inside the first conv.  At the
rates of BASELINE configs[1] that is ~0.64 GB of float64 per batch touched three times by host cores.  Here the host only
moves BYTES: the kept channel rows of each .npy file are read straight into a pinned staging block (a thread pool of
pread calls, no numpy temporaries), copied to HBM on a copy stream, and ONE kernel (ns_feed_pack) applies every reader /
collator rule while it transposes into the (B, T+2, Cp) fp16 layout the conv GEMM reads.  Bit-identical to
collator -> ns_signal_pack (tests/test_feed_gpu.py).

Round 6: an OPT-IN cache of the kept channel rows in a narrower type (`SignalFeed(cache_dir=..., cache_dtype="f16" |
"f32")`, finetune.py / evaluation.py --feed_cache_dir / --feed_cache_dtype).  The reference's recordings are float64; what reaches the
first conv is round16(round32(x)) (collator: torch float32; autocast: fp16), which is exactly what ns_feed_pack computes from the float64
bytes.  The cache stores those roundings once per (file, channel slice) -- numpy's astype is round-to-nearest-even like the device's
conversions, so a cached batch is BIT-IDENTICAL to the float64 one (tests/test_feed_gpu.py) -- and from then on a rank reads and stages a
quarter (f16) or half (f32) of the bytes: at eight ranks x ~2 100 samples/s the float64 files are ~180 GB/s through host DRAM and eight
PCIe links, the f16 cache 45 GB/s.  f16 cannot reproduce the optional fp32 copy of the batch (keep_x32): such feeds cache f32.

Host objects:
  RawSignal     what utils.reader.CustomDataset(raw_signals=True) returns instead of the padded array
  plan_read     header-only planning of one recording (CPU-testable)
  SignalFeed    staging slots + copy stream; load(raws) -> PackedSignal
  PackedSignal  stands in for the (B, ch, T) fp32 tensor at MegWhisperEngine.forward / train_step
"""
import ast
import os
import struct
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np
import torch

from .lib import GPU_CAPTURE_LOCK, NS_FEED_F16, NS_FEED_F32, NS_FEED_F64

_DTYPES = {"<f8": (NS_FEED_F64, 8), "<f4": (NS_FEED_F32, 4), "<f2": (NS_FEED_F16, 2)}
_ALIGN = 256
ITEM_BYTES = 32     # sizeof(ns_feed_item)


@dataclass(frozen=True)
class RawSignal:
    """A recording the reader has not loaded: the file and the channel slice [row0:row1] it would take
    (utils/reader.py:282-290); `ch` = modal_ch, channels the file cannot supply are zero channels."""
    path: str
    row0: int
    row1: int
    ch: int


def npy_layout(path: str) -> Tuple[int, tuple, str, bool]:
    """(data offset, shape, descr, fortran_order) from the .npy header alone (format: magic, version, little-endian
    header length, python dict literal)."""
    with open(path, "rb") as f:
        magic = f.read(8)
        if len(magic) < 8 or magic[:6] != b"\x93NUMPY":
            raise ValueError(f"{path}: not a .npy file")
        major = magic[6]
        if major == 1:
            hlen, off = struct.unpack("<H", f.read(2))[0], 10
        elif major in (2, 3):
            hlen, off = struct.unpack("<I", f.read(4))[0], 12
        else:
            raise ValueError(f"{path}: unsupported .npy version {major}")
        hdr = ast.literal_eval(f.read(hlen).decode("utf8" if major == 3 else "latin1"))
    return off + hlen, tuple(hdr["shape"]), str(hdr["descr"]), bool(hdr["fortran_order"])


@dataclass
class ReadPlan:
    """How one recording reaches the staging block.  kind: 'span' one contiguous byte range (rows x n elements);
    'rows' one range per channel row (recording longer than 2T: only the first T samples of each row are read);
    'array' host fallback (non-float / big-endian / Fortran-order files: numpy loads it, already sliced)."""
    kind: str
    path: str
    rows: int
    n: int
    ld: int
    dtype: int
    itemsize: int
    nbytes: int
    offset: int = 0           # file offset of the first kept element
    row_stride: int = 0       # file bytes between channel rows ('rows' only)
    array: Optional[np.ndarray] = None


def plan_read(raw: RawSignal, T: int) -> ReadPlan:
    off, shape, descr, fortran = npy_layout(raw.path)
    assert len(shape) == 2, f"sample.shape is {shape}"                      # reader.py:510
    rows_file, n = shape
    r0, r1 = min(raw.row0, rows_file), min(raw.row1, rows_file)
    rows = max(0, r1 - r0)
    assert rows <= raw.ch, f"sample shape {(rows, T)} != {(raw.ch, T)}"     # reader.py:503-505
    if descr in _DTYPES and not fortran:
        code, isz = _DTYPES[descr]
        if n <= 2 * T:
            return ReadPlan("span", raw.path, rows, n, n, code, isz, rows * n * isz, off + r0 * n * isz)
        return ReadPlan("rows", raw.path, rows, T, T, code, isz, rows * T * isz, off + r0 * n * isz, n * isz)
    arr = np.load(raw.path)[r0:r1, :T]
    if arr.dtype.kind == "f" and arr.dtype.itemsize in (2, 4, 8):
        arr = np.ascontiguousarray(arr.astype(arr.dtype.newbyteorder("=")))
    else:       # the collator's own conversion: torch.tensor(x, dtype=float32)
        arr = np.ascontiguousarray(arr.astype(np.float32))
    code, isz = _DTYPES["<f" + str(arr.dtype.itemsize)]
    return ReadPlan("array", raw.path, rows, arr.shape[1], arr.shape[1], code, isz, arr.nbytes, array=arr)


# ---------------------------------------------------------------------------------------------------- narrow-type cache of the kept rows
_CACHE_NP = {"f16": np.float16, "f32": np.float32}
_NPY_HDR = 128      # cache files: .npy version 1 with the header padded to 128 bytes (data 64-B aligned)


def cache_path(cache_dir: str, raw: RawSignal, T: int, dtype: str) -> str:
    """One cache file per (source file identity, channel slice, crop rule, type).  The source's size and mtime are part of the name: a
    rewritten recording gets a new cache file instead of serving stale rows."""
    import hashlib
    st = os.stat(raw.path)
    key = f"{os.path.abspath(raw.path)}|{st.st_size}|{st.st_mtime_ns}|{raw.row0}|{raw.row1}|{T}|{dtype}"
    return os.path.join(cache_dir, hashlib.sha256(key.encode()).hexdigest()[:32] + f".{dtype}.npy")


def build_cache_file(plan: ReadPlan, dst: str, dtype: str):
    """The rows a plan would stage, rounded as collator + autocast round them (float64 -> float32 [-> float16], round-to-nearest-even at
    each step: what ns_feed_pack does on the device), written as a C-order .npy next to its final name and renamed into place (readers
    never see a partial file; two ranks racing write the same bytes)."""
    buf = bytearray(plan.nbytes)
    _fill(plan, memoryview(buf))
    src = np.frombuffer(buf, dtype={NS_FEED_F64: "<f8", NS_FEED_F32: "<f4", NS_FEED_F16: "<f2"}[plan.dtype]).reshape(plan.rows, plan.n)
    out = src.astype(np.float32)            # the collator's torch.tensor(x, dtype=float32)
    if dtype == "f16":
        out = out.astype(np.float16)        # autocast's cast in front of the first conv
    hdr = ("{'descr': '%s', 'fortran_order': False, 'shape': (%d, %d), }" % ("<f2" if dtype == "f16" else "<f4", plan.rows, plan.n)).encode("latin1")
    pad = _NPY_HDR - 10 - len(hdr) - 1
    assert pad >= 0
    head = b"\x93NUMPY\x01\x00" + struct.pack("<H", _NPY_HDR - 10) + hdr + b" " * pad + b"\n"
    tmp = f"{dst}.tmp.{os.getpid()}.{id(buf)}"
    with open(tmp, "wb") as f:
        f.write(head)
        f.write(np.ascontiguousarray(out).tobytes())
    os.replace(tmp, dst)


def plan_cached(raw: RawSignal, T: int, cache_dir: str, dtype: str) -> ReadPlan:
    """plan_read through the cache: the plan of the cached rows (one contiguous span of f16 / f32), building the file on first touch.
    Recordings the plain plan would hand to numpy ('array': exotic dtypes / byte orders) and empty slices are not cached."""
    cp = cache_path(cache_dir, raw, T, dtype)
    if not os.path.exists(cp):
        plan = plan_read(raw, T)
        if plan.kind == "array" or plan.nbytes == 0 or (plan.dtype == NS_FEED_F16) or (plan.dtype == NS_FEED_F32 and dtype == "f32"):
            return plan                      # nothing narrower to store
        os.makedirs(cache_dir, exist_ok=True)
        build_cache_file(plan, cp, dtype)
    off, shape, descr, fortran = npy_layout(cp)
    code, isz = _DTYPES[descr]
    rows, n = shape
    return ReadPlan("span", cp, rows, n, n, code, isz, rows * n * isz, off)


def _pread_into(fd: int, view: memoryview, offset: int):
    done = 0
    while done < len(view):
        k = os.preadv(fd, [view[done:]], offset + done)
        if k <= 0:
            raise IOError("short read")
        done += k


def _fill(plan: ReadPlan, dst: memoryview):
    """Bytes of one recording into its staging range (runs in a pool thread; pread drops the GIL)."""
    if plan.nbytes == 0:
        return
    if plan.kind == "array":
        dst[:] = memoryview(plan.array).cast("B")
        return
    fd = os.open(plan.path, os.O_RDONLY)
    try:
        if plan.kind == "span":
            _pread_into(fd, dst, plan.offset)
        else:
            rb = plan.n * plan.itemsize
            for r in range(plan.rows):
                _pread_into(fd, dst[r * rb:(r + 1) * rb], plan.offset + r * plan.row_stride)
    finally:
        os.close(fd)


def layout_batch(plans: List[ReadPlan]) -> Tuple[List[int], int]:
    """Aligned staging offsets of each recording and the bytes used."""
    offs, cur = [], 0
    for p in plans:
        offs.append(cur)
        cur += (p.nbytes + _ALIGN - 1) // _ALIGN * _ALIGN
    return offs, cur


def item_table(plans: List[ReadPlan], offs: List[int], base_ptr: int) -> np.ndarray:
    """ns_feed_item records (include/neuspeech_hip.h) for a staging block that starts at device address base_ptr."""
    t = np.zeros(len(plans), dtype=np.dtype([("src", "<u8"), ("ld", "<i8"), ("rows", "<i4"), ("n", "<i4"),
                                             ("dtype", "<i4"), ("reserved", "<i4")]))
    assert t.dtype.itemsize == ITEM_BYTES
    for i, (p, o) in enumerate(zip(plans, offs)):
        t[i] = (base_ptr + o, p.ld, p.rows, p.n, p.dtype, 0)
    return t


class PackedSignal:
    """(B, T+2, Cp) fp16 batch already in the conv GEMM's operand layout; accepted wherever the engine takes the
    (B, ch, T) fp32 tensor.  `release()` (or dropping the last reference) hands the slot back to the feed: call it after
    the step that reads the batch has been enqueued."""

    def __init__(self, feed, slot, B, ch, T):
        self.feed, self.slot, self.B, self.ch, self.T = feed, slot, B, ch, T
        self.xin = slot.xin[:B]
        self.x32 = slot.x32[:B] if slot.x32 is not None else None
        self._acquired = False

    @property
    def shape(self):
        return (self.B, self.ch, self.T)

    @property
    def device(self):
        return self.xin.device

    def to(self, *a, **k):      # call sites move `input_features` to the model's device: already there
        return self

    def contiguous(self):
        return self

    def acquire(self):
        """make the CURRENT stream wait for the copy-stream work that fills this batch (idempotent)"""
        if not self._acquired:
            torch.cuda.current_stream().wait_event(self.slot.ready)
            self._acquired = True
        return self

    def release(self):
        if self.slot is not None:
            self.slot.free.record(torch.cuda.current_stream())
            self.slot.busy = False
            self.feed._tick += 1
            self.slot.released_at = self.feed._tick
            self.slot = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class _Slot:
    def __init__(self):
        self.pinned = self.staging = self.items_host = self.items_dev = self.xin = self.x32 = None
        self.h2d_done = torch.cuda.Event()
        self.ready = torch.cuda.Event()
        self.free = torch.cuda.Event()
        self.busy = False
        self.used_once = False
        self.released_at = 0


class SignalFeed:
    """Staging slots + copy stream.  load() returns as soon as the file bytes are in pinned memory and the H2D copy +
    ns_feed_pack are enqueued on the copy stream; the consumer's stream waits on the batch's event (acquire)."""
    MAX_SLOTS = 8
    MIN_SLOTS = 3

    def __init__(self, device, ch: int, T: int, Cp: int, threads: int = 8, keep_x32: bool = False, cache_dir: Optional[str] = None,
                 cache_dtype: str = "f16"):
        self.device, self.ch, self.T, self.Cp, self.keep_x32 = torch.device(device), ch, T, Cp, keep_x32
        assert self.device.type == "cuda", "SignalFeed drives the HIP path; there is no host fallback"
        if cache_dir:
            if cache_dtype not in _CACHE_NP:
                raise ValueError(f"feed cache type {cache_dtype!r}: 'f16' or 'f32'")
            if keep_x32 and cache_dtype == "f16":
                cache_dtype = "f32"         # the fp32 copy of the batch is round32(x): an f16 cache could not reproduce it
        self.cache_dir, self.cache_dtype = (cache_dir or None), cache_dtype
        self.cache_built = 0
        self.stream = torch.cuda.Stream(self.device)
        self.pool = ThreadPoolExecutor(max_workers=max(1, threads))
        self.slots: List[_Slot] = []
        self.bytes_staged = 0
        self._tick = 0
        self._loader = None

    def _slot(self) -> _Slot:
        # least recently released first: a slot that was just handed back still has its reader in flight, and the copy
        # stream would sit behind that step (first-fit reuse serialised copy and compute: 52 ms per step instead of 39)
        free = [s for s in self.slots if not s.busy]
        if len(free) >= 2 or (free and len(self.slots) >= self.MIN_SLOTS):
            s = min(free, key=lambda q: q.released_at)
            return s
        if len(self.slots) >= self.MAX_SLOTS:
            raise RuntimeError("SignalFeed: every slot is held by an unreleased PackedSignal")
        s = _Slot()
        self.slots.append(s)
        return s

    def _fit(self, s: _Slot, B: int, nbytes: int):
        """grow the slot's buffers (rare: first batches only).  A grown buffer replaces one the GPU may still read, so
        the device is drained first."""
        grow_stage = s.pinned is None or s.pinned.numel() < nbytes
        grow_b = s.xin is None or s.xin.shape[0] < B
        if not (grow_stage or grow_b):
            return
        torch.cuda.synchronize(self.device)
        with torch.cuda.stream(self.stream):
            if grow_stage:
                cap = max(int(nbytes * 1.5), 1 << 20)      # few regrows: each one drains the device and re-pins
                s.pinned = torch.empty(cap, dtype=torch.uint8, pin_memory=True)
                s.staging = torch.empty(cap, dtype=torch.uint8, device=self.device)
            if grow_b:
                s.items_host = torch.empty(B * ITEM_BYTES, dtype=torch.uint8, pin_memory=True)
                s.items_dev = torch.empty(B * ITEM_BYTES, dtype=torch.uint8, device=self.device)
                s.xin = torch.empty(B, self.T + 2, self.Cp, dtype=torch.float16, device=self.device)
                s.x32 = torch.empty(B, self.ch, self.T, dtype=torch.float32, device=self.device) if self.keep_x32 else None
        torch.cuda.synchronize(self.device)

    def load(self, raws: List[RawSignal]) -> PackedSignal:
        from . import ops
        B = len(raws)
        assert B > 0
        for r in raws:
            assert r.ch == self.ch, f"recording planned for {r.ch} channels, feed built for {self.ch}"
        if self.cache_dir:
            plans = list(self.pool.map(lambda r: plan_cached(r, self.T, self.cache_dir, self.cache_dtype), raws))
        else:
            plans = list(self.pool.map(lambda r: plan_read(r, self.T), raws))
        offs, used = layout_batch(plans)
        s = self._slot()
        with GPU_CAPTURE_LOCK:              # never synchronize / allocate while another thread has a graph capture open
            self._fit(s, B, max(used, 1))
        if s.used_once:
            s.h2d_done.synchronize()        # the pinned block is about to be overwritten by host threads
        host = memoryview(s.pinned.numpy())
        list(self.pool.map(lambda po: _fill(po[0], host[po[1]:po[1] + po[0].nbytes]), zip(plans, offs)))
        tab = item_table(plans, offs, s.staging.data_ptr())
        s.items_host[:B * ITEM_BYTES].copy_(torch.from_numpy(tab.view(np.uint8).reshape(-1)))
        with GPU_CAPTURE_LOCK, torch.cuda.stream(self.stream):
            if s.used_once:
                self.stream.wait_event(s.free)      # the step that read this slot's previous batch has finished
            if used:
                s.staging[:used].copy_(s.pinned[:used], non_blocking=True)
            s.items_dev[:B * ITEM_BYTES].copy_(s.items_host[:B * ITEM_BYTES], non_blocking=True)
            s.h2d_done.record(self.stream)
            ops.feed_pack(s.items_dev, B, self.ch, self.T, self.Cp, s.xin, s.x32)
            s.ready.record(self.stream)
        s.busy, s.used_once = True, True
        self.bytes_staged += used
        return PackedSignal(self, s, B, self.ch, self.T)

    def submit(self, raws: List[RawSignal]):
        """load() on the feed's own loader thread (one at a time, in order): the file reads then overlap the caller's
        kernel launches instead of preceding them.  Returns a Future of PackedSignal."""
        if self._loader is None:
            self._loader = ThreadPoolExecutor(max_workers=1, thread_name_prefix="ns-feed")
        return self._loader.submit(self._load_on_thread, raws)

    def _load_on_thread(self, raws):
        torch.cuda.set_device(self.device)      # the current device is per thread
        return self.load(raws)

    def close(self):
        if self._loader is not None:
            self._loader.shutdown(wait=True)
        self.pool.shutdown(wait=True)
