"""Persistence of the in-situ tile pins (``cfgpp_unet_tuning``) across processes.

The first forward at a batch size times every implicit-GEMM launch of the plan in
place (about 40 extra forwards and one host sync) and pins the
fastest tile per launch.  Every candidate gives bit-identical results, so the
pins only decide speed - and they are a property of (model, latent size, batch
rows, GPU model, kernel build), not of the process.  ``PinCache`` keeps them in
``$CFGPP_TUNE_CACHE`` (default ``~/.cache/cfgpp_amd``; ``0`` or empty = off) as
``tune_<model>_<HxW>_r<rows>_<device>_<build>.json``: only the first process on a
box pays for the tuning passes, later ones import the pins before their first
forward and the C ABI's "never synchronises" holds from the first call on.
``CFGPP_AUTOTUNE=0`` keeps its meaning (heuristic tiles, nothing pinned, nothing
read or written).
"""
from __future__ import annotations

import hashlib
import json
import os
import re
import tempfile
from typing import Callable, List, Optional

_BUILD_ID = {}


def build_id(lib_path: str) -> str:
    """short content digest of the kernel library: pins of another build are never replayed"""
    if lib_path not in _BUILD_ID:
        h = hashlib.sha1()
        try:
            with open(lib_path, "rb") as f:
                for chunk in iter(lambda: f.read(1 << 20), b""):
                    h.update(chunk)
            _BUILD_ID[lib_path] = h.hexdigest()[:12]
        except OSError:
            _BUILD_ID[lib_path] = "nolib"
    return _BUILD_ID[lib_path]


def skewed_run() -> bool:
    """timings of this process are not representative: a profiler is attached (rocprofv3 / rocprof export these into the child)"""
    return any(k.startswith(("ROCP_", "ROCPROFILER_", "ROCPROF_")) or k == "HSA_TOOLS_LIB" for k in os.environ)


def cache_dir() -> Optional[str]:
    d = os.environ.get("CFGPP_TUNE_CACHE", os.path.join(os.path.expanduser("~"), ".cache", "cfgpp_amd"))
    if d in ("", "0") or os.environ.get("CFGPP_AUTOTUNE", "1") == "0":
        return None
    return d


class PinCache:
    """``export_fn(rows) -> list[int]`` / ``import_fn(hints, rows)`` are the engine's two halves of ``cfgpp_unet_tuning``."""

    def __init__(self, model: str, latent_hw, device_name: str, build: str,
                 export_fn: Callable[[int], List[int]], import_fn: Callable[[List[int], int], None], directory: Optional[str] = None,
                 knobs: Optional[Callable[[], tuple]] = None):
        self.dir = cache_dir() if directory is None else directory
        self.tag = "_".join(re.sub(r"[^A-Za-z0-9.]+", "-", str(x)) for x in (model, f"{latent_hw[0]}x{latent_hw[1]}"))
        self.dev = re.sub(r"[^A-Za-z0-9.]+", "-", device_name)
        self.build = build
        self._export, self._import = export_fn, import_fn
        self._done = set()          # rows whose pins are on disk (or were read from it, or must not be written)
        self._tries = {}            # rows -> forwards after which the engine still had no pins to export
        self._knobs = knobs         # () -> tuple of the tuner's switches (candidate mask, big tiles, ...); pins are only persisted
        self._knobs_default = knobs() if knobs is not None else None      # while it still equals its value at construction

    def path(self, rows: int) -> str:
        return os.path.join(self.dir, f"tune_{self.tag}_r{rows}_{self.dev}_{self.build}.json")

    def load(self, rows: int) -> bool:
        """before the first forward at `rows`: install the pins a previous process left (True when it did)"""
        if self.dir is None or rows in self._done:
            return False
        try:
            with open(self.path(rows)) as f:
                rec = json.load(f)
            hints = [int(h) for h in rec["hints"]]
            if rec.get("rows") != rows or not hints:
                return False
            self._import(hints, rows)
        except Exception:  # noqa: BLE001  (missing / stale / unreadable file, slot count of another plan: tune as usual)
            return False
        self._done.add(rows)
        return True

    def mark_imported(self, rows: int):
        """pins for `rows` were installed explicitly (bench.py: rank 0's pins broadcast to every rank): neither read the disk
        over them nor write them back"""
        self._done.add(rows)

    def save(self, rows: int) -> bool:
        """after the first forward at `rows`: write the pins the tuner chose (atomic rename; races between ranks are harmless -
        every writer holds valid pins).  Nothing is written from a skewed run: under a profiler (rocprofv3 sets ROCP_* /
        ROCPROFILER_* variables in the child), or with a non-default tuner state (`knobs`)."""
        if self.dir is None or rows in self._done:
            return False
        if skewed_run() or (self._knobs is not None and self._knobs() != self._knobs_default):
            self._done.add(rows)    # definitive: this process never persists pins for this batch
            return False
        try:
            hints = self._export(rows)
        except Exception:  # noqa: BLE001  (not tuned yet: the first forward ran inside a stream capture, autotune off through the API)
            hints = None
        if not hints:
            self._tries[rows] = self._tries.get(rows, 0) + 1
            if self._tries[rows] >= 3:      # "not tuned" after three forwards is definitive
                self._done.add(rows)
            return False
        try:
            os.makedirs(self.dir, exist_ok=True)
            fd, tmp = tempfile.mkstemp(dir=self.dir, prefix=".tune_", suffix=".tmp")
            with os.fdopen(fd, "w") as f:
                json.dump({"rows": rows, "hints": hints, "model": self.tag, "device": self.dev, "build": self.build}, f)
            os.replace(tmp, self.path(rows))
            self._done.add(rows)
            return True
        except Exception:  # noqa: BLE001  (read-only home, ...)
            self._done.add(rows)    # one write attempt per batch size and process
            return False
