"""Data ingest on either side of the forward path.

The reference reads ONE ``.mat`` file per sample with ``scipy.io.loadmat`` and extracts the pilots
with a boolean mask on the host (reference src/data/dataset.py:95-190, src/utils.py:68-110); once
the model runs at > 10^4 frames/s that is the end-to-end bottleneck.  Here a folder of ``.mat`` files
(same file-name convention, same ``H`` variable) is packed ONCE into a contiguous ``.npz`` and batches
are cut from it; the non-zero-pilot gather and the LS-baseline metric run on the GPU
(``aft_pilot_gather_f32`` / ``aft_ls_mse_db_f32``).  Batches come out in exactly the format the
reference's ``DataLoader`` yields: ``(pilots cfloat[B,Ps,Pt], H cfloat[B,S,T], meta 6-tuple)`` with
``meta = (file_no, snr, ds, dop, n: float32 [B,1] each, [tuple of B channel-type strings])``.
"""
from __future__ import annotations

import logging
import os
import re
from pathlib import Path
from typing import Dict, Iterator, List, Optional, Tuple, Union

import numpy as np
import torch

_NAME = re.compile(r"(\d+)_SNR-(\d+)_DS-(\d+)_DOP-(\d+)_N-(\d+)_([A-Z\-]+)\.mat")


def parse_filename(file_name: str) -> Tuple[float, float, float, float, float, str]:
    """``{no}_SNR-{snr}_DS-{ds}_DOP-{dop}_N-{n}_{type}.mat`` -> numbers + channel type
    (reference src/utils.py:68-110; same ``re.match`` semantics, same error text)."""
    m = _NAME.match(file_name)
    if not m:
        raise ValueError("Cannot extract file information.")
    return (float(int(m.group(1))), float(int(m.group(2))), float(int(m.group(3))), float(int(m.group(4))),
            float(int(m.group(5))), m.group(6))


def pack_mat_folder(folder: Union[str, Path], out_path: Optional[Union[str, Path]] = None) -> Dict[str, np.ndarray]:
    """Read every ``*.mat`` of ``folder`` (variable ``H`` [S,T,>=2] complex: [:,:,0] ideal channel,
    [:,:,1] LS estimate at the pilot positions and zero elsewhere, optional [:,:,2] full LS estimate,
    reference dataset.py:3-21, utils.py:283-288) into contiguous arrays; optionally save as ``.npz``."""
    import scipy.io as sio
    folder = Path(folder)
    if not folder.exists():
        raise FileNotFoundError(f"Data directory not found: {folder}")
    files = sorted(folder.glob("*.mat"), key=lambda p: p.name)   # the reference's glob order is unspecified (B13)
    if not files:
        raise ValueError(f"No .mat files found in {folder}")
    ideal, sparse, full, meta, ctype = [], [], [], [], []
    for f in files:
        mat = sio.loadmat(f)
        if "H" not in mat or mat["H"].shape[-1] < 2:
            raise ValueError(f"Error processing file {f}: Invalid .mat file format: missing required data")
        H = mat["H"]
        ideal.append(H[:, :, 0].astype(np.complex64))
        sparse.append(H[:, :, 1].astype(np.complex64))
        if H.shape[-1] > 2:
            full.append(H[:, :, 2].astype(np.complex64))
        *nums, ct = parse_filename(f.name)
        meta.append(nums)
        ctype.append(ct)
    # loadmat hands back Fortran-ordered planes and np.stack keeps that order: force row-major
    packed = {"h_ideal": np.ascontiguousarray(np.stack(ideal)), "h_ls_sparse": np.ascontiguousarray(np.stack(sparse)),
              "meta": np.asarray(meta, dtype=np.float32), "channel_type": np.asarray(ctype)}
    if len(full) == len(files):
        packed["h_ls_full"] = np.ascontiguousarray(np.stack(full))
    if out_path is not None:
        np.savez(out_path, **packed)
    return packed


def extract_pilots_host(h_ls_sparse: np.ndarray, pilot_size: Tuple[int, int]) -> np.ndarray:
    """CPU path of the pilot gather (used when the tensors live on the CPU)."""
    B = h_ls_sparse.shape[0]
    expected = pilot_size[0] * pilot_size[1]
    out = np.empty((B, pilot_size[0], pilot_size[1]), np.complex64)
    for b in range(B):
        nz = h_ls_sparse[b][h_ls_sparse[b] != 0]
        if nz.size != expected:
            raise ValueError(f"Expected {expected} pilot values, got {nz.size} (frame {b})")
        out[b] = nz.reshape(pilot_size)
    return out


class PackedLoader:
    """Iterates a packed folder in file order (``shuffle=False`` as in the reference's test loaders,
    dataset.py:254-260) and yields reference-format batches.  On a HIP device the pilots are gathered
    by the GPU kernel and both pilots and targets stay on the device.  There the reference's "Expected 24 pilot values,
    got 25" ValueError (dataset.py:128-132) is raised ONE BATCH LATE (the counts come back asynchronously): when the
    next batch is requested, when the iterator is exhausted, or when it is CLOSED (``it.close()``,
    ``contextlib.closing(iter(loader))``).  A consumer that simply leaves the loop (``break``, or an exception of its own,
    which propagates unchanged) abandons the sweep with one check outstanding: that check is still settled when the
    iterator is finalised, a failure is LOGGED as an error (never an "Exception ignored in generator" on stderr that nobody
    reads) and kept on the loader, where ``raise_pending()`` -- or leaving a ``with loader:`` block -- raises it."""

    def __init__(self, packed: Union[str, Path, Dict[str, np.ndarray]], pilot_size: Tuple[int, int], batch_size: int,
                 device: Union[str, torch.device] = "cpu", pin_memory: bool = True,
                 max_pinned_bytes: int = 8 << 30) -> None:
        if not isinstance(packed, dict):
            z = np.load(packed, allow_pickle=False)
            packed = {k: z[k] for k in z.files}
        self.p = packed
        self.pilot_size = tuple(pilot_size)
        self.batch_size = int(batch_size)
        self.device = torch.device(device)
        self.n = packed["h_ideal"].shape[0]
        # HIP device: the two grids a batch needs are pinned once, so that every batch is two asynchronous copies and the
        # host never waits for the device inside the sweep (a pageable .to(device) is a synchronisation point: with it
        # the "sync-free" evaluation sweep of evaluation.py ran no faster than the reference's .item()-per-batch loop)
        # (the grids are held ONCE: the pinned copies replace the pageable arrays in self.p; packs above
        # ``max_pinned_bytes`` -- page-locked memory is a scarce host resource -- stay pageable and every batch goes
        # through a small pinned ring instead)
        self._pinned = None
        self._stage_ring = None
        if self.device.type == "cuda" and pin_memory:
            grids = [np.ascontiguousarray(packed[k]) for k in ("h_ideal", "h_ls_sparse")]
            if sum(g.nbytes for g in grids) <= max_pinned_bytes:
                self._pinned = tuple(torch.from_numpy(g).pin_memory() for g in grids)
                self.p = dict(packed)
                self.p["h_ideal"], self.p["h_ls_sparse"] = (t.numpy() for t in self._pinned)   # views of the pinned memory
            else:
                shape = (self.batch_size, *grids[0].shape[1:])
                dt_i, dt_s = (torch.from_numpy(self.p[k][:0]).dtype for k in ("h_ideal", "h_ls_sparse"))   # the grids' own dtypes
                self._stage_ring = [(torch.empty(shape, dtype=dt_i).pin_memory(),
                                     torch.empty(shape, dtype=dt_s).pin_memory(), None) for _ in range(3)]

    def __len__(self) -> int:
        return (self.n + self.batch_size - 1) // self.batch_size

    # -- deferred pilot-count errors of abandoned sweeps ---------------------------------------------------------------
    def raise_pending(self) -> None:
        """Raise (once) the count error of a sweep that was abandoned before its last check could be delivered."""
        err, self._pending_error = getattr(self, "_pending_error", None), None
        if err is not None:
            raise err

    def __enter__(self) -> "PackedLoader":
        return self

    def __exit__(self, exc_type, exc, tb) -> bool:
        if exc_type is None:
            self.raise_pending()
        return False

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor, tuple]]:
        if self.device.type != "cuda":
            return self._batches()
        return _Sweep(self, self._batches())

    def _batches(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor, tuple]]:
        if self.device.type != "cuda":
            for lo in range(0, self.n, self.batch_size):
                hi = min(lo + self.batch_size, self.n)
                ideal = torch.from_numpy(self.p["h_ideal"][lo:hi])
                pilots = torch.from_numpy(extract_pilots_host(self.p["h_ls_sparse"][lo:hi], self.pilot_size))
                yield pilots, ideal, self._meta(lo, hi)
            return
        # HIP device: the grids of batch k+1 are copied on a side stream while the consumer works on batch k (3.4 MB per
        # 128-frame batch is ~0.15 ms of PCIe time: on the compute stream it would sit in front of every forward); the
        # consumer's stream waits for the copy's event on the device, the host never does.  Only COPIES run on the side
        # stream: the gather kernel is launched on the consumer's stream -- a kernel running beside the forward's
        # persistent-grid launches takes workgroup slots from them and costs more than its own 10 us.  Pilot counts come
        # back through a small pinned ring and are checked one batch late (the reference raises at the offending sample).
        from .hip_ops import check_pilot_counts, pilot_gather
        expected = self.pilot_size[0] * self.pilot_size[1]
        side = torch.cuda.Stream(device=self.device)
        ring = [torch.empty(self.batch_size, dtype=torch.int32).pin_memory() for _ in range(3)]

        ring_turn = [0]

        def copy(lo):
            hi = min(lo + self.batch_size, self.n)
            with torch.cuda.stream(side):
                if self._pinned is not None:
                    ideal = self._pinned[0][lo:hi].to(self.device, non_blocking=True)
                    sparse = self._pinned[1][lo:hi].to(self.device, non_blocking=True)
                elif self._stage_ring is not None:     # large pack: pageable -> pinned ring slot -> device, still async
                    slot = ring_turn[0] % len(self._stage_ring)
                    ring_turn[0] += 1
                    h_i, h_s, busy = self._stage_ring[slot]
                    if busy is not None:
                        busy.synchronize()             # the copy that last read this slot (three batches ago)
                    h_i[:hi - lo].copy_(torch.from_numpy(self.p["h_ideal"][lo:hi]))
                    h_s[:hi - lo].copy_(torch.from_numpy(self.p["h_ls_sparse"][lo:hi]))
                    ideal = h_i[:hi - lo].to(self.device, non_blocking=True)
                    sparse = h_s[:hi - lo].to(self.device, non_blocking=True)
                    busy = torch.cuda.Event()
                    busy.record(side)
                    self._stage_ring[slot] = (h_i, h_s, busy)
                else:
                    ideal = torch.from_numpy(self.p["h_ideal"][lo:hi]).to(self.device)
                    sparse = torch.from_numpy(self.p["h_ls_sparse"][lo:hi]).to(self.device)
                ev = torch.cuda.Event()
                ev.record(side)
            return ideal, sparse, ev, lo, hi

        def settle(b):
            b[0].synchronize()   # recorded behind that batch's gather: long done when the NEXT batch has been consumed
            check_pilot_counts(b[1], expected, b[2])

        starts = list(range(0, self.n, self.batch_size))
        nxt = copy(starts[0]) if starts else None
        prev = None
        try:
            for k in range(len(starts)):
                ideal, sparse, ev, lo, hi = nxt
                nxt = copy(starts[k + 1]) if k + 1 < len(starts) else None
                main = torch.cuda.current_stream(self.device)
                main.wait_event(ev)
                ideal.record_stream(main)
                sparse.record_stream(main)
                pilots, counts = pilot_gather(sparse, self.pilot_size, return_counts=True)
                host_counts = ring[k % 3][:hi - lo]
                host_counts.copy_(counts, non_blocking=True)
                done = torch.cuda.Event()
                done.record(main)
                if prev is not None:
                    pending, prev = prev, None
                    settle(pending)
                prev = (done, host_counts, lo)
                yield pilots, ideal, self._meta(lo, hi)
        except GeneratorExit:
            # the consumer stopped early: the last yielded batch's count check is settled before the generator goes away.  A
            # failure is handed to the loader, not raised here -- raised inside a generator that the garbage collector is
            # finalising it would only be printed; _Sweep.close() raises it, _Sweep.__del__ logs it
            if prev is not None:
                pending, prev = prev, None
                try:
                    settle(pending)
                except ValueError as err:
                    self._pending_error = err
            raise
        # normal exhaustion: the count check of the last batch.  (An exception thrown by the consumer's loop body is NOT
        # intercepted: it propagates unchanged, and the pending check is dropped with the failed sweep.)
        if prev is not None:
            settle(prev)

    def _meta(self, lo: int, hi: int) -> tuple:
        m = torch.from_numpy(self.p["meta"][lo:hi])
        return (m[:, 0:1], m[:, 1:2], m[:, 2:3], m[:, 3:4], m[:, 4:5],
                [tuple(str(c) for c in self.p["channel_type"][lo:hi])])


class _Sweep:
    """One pass over a PackedLoader on the HIP device: the batch generator plus what happens to the count check that is
    still outstanding when the consumer stops early (PackedLoader docstring)."""

    def __init__(self, loader: "PackedLoader", gen) -> None:
        self._loader, self._gen = loader, gen

    def __iter__(self) -> "_Sweep":
        return self

    def __next__(self):
        return next(self._gen)

    def close(self) -> None:
        """Settle the outstanding check and RAISE its error (explicit close: ``it.close()`` / ``contextlib.closing``)."""
        self._gen.close()
        self._loader.raise_pending()

    def __del__(self) -> None:
        try:
            self._gen.close()
        except Exception:      # nothing may escape a finaliser
            return
        err = getattr(self._loader, "_pending_error", None)
        if err is not None:
            logging.getLogger(__name__).error(
                "PackedLoader: %s -- found while settling an abandoned sweep; loader.raise_pending() (or `with loader:`) raises it", err)


def ls_mse_db_per_frame(h_ls_full: torch.Tensor, h_ideal: torch.Tensor) -> torch.Tensor:
    """10 log10(mean |LS - ideal|^2) per frame (reference utils.py:248-261)."""
    if h_ls_full.device.type == "cuda":
        from .hip_ops import ls_mse_db
        return ls_mse_db(h_ls_full, h_ideal.to(h_ls_full.device))
    d = (h_ls_full - h_ideal).abs().double() ** 2
    return (10.0 * torch.log10(d.mean(dim=(1, 2)))).float()


def get_ls_mse_per_folder(folders_dir: Union[str, Path], device: Union[str, torch.device] = "cpu") -> Dict[int, float]:
    """{int(val): mean over files of the per-file LS MSE in dB} for sub-folders named ``prefix_val``,
    sorted by val -- the reference's get_ls_mse_per_folder (utils.py:264-303), with the per-file
    reductions done in one kernel per folder."""
    out: Dict[int, float] = {}
    for folder in sorted(os.listdir(folders_dir), key=lambda s: int(s.split("_")[1])):
        packed = pack_mat_folder(os.path.join(folders_dir, folder))
        if "h_ls_full" not in packed:
            raise ValueError(f"{folder}: 'H' has no [:,:,2] LS-estimate slice")
        ls = torch.from_numpy(packed["h_ls_full"]).to(device)
        ideal = torch.from_numpy(packed["h_ideal"]).to(device)
        out[int(folder.split("_")[1])] = float(ls_mse_db_per_frame(ls, ideal).double().mean())
    return out
