"""Tile-pin persistence (cfgpp_amd/tune_cache.py) with a stand-in for the engine's two halves of cfgpp_unet_tuning: the first
"process" tunes and writes, the second imports before its first forward; another build / device / batch never replays them."""
import json
import os

from cfgpp_amd.tune_cache import PinCache, cache_dir


class FakeUNet:
    """export raises until a forward 'tuned' the batch (as cfgpp_unet_tuning returns -3), import records what it was given"""
    def __init__(self, n_slots=5):
        self.n, self.tuned, self.imported = n_slots, {}, {}

    def forward(self, rows):
        if rows not in self.imported and rows not in self.tuned:
            self.tuned[rows] = [(7 * i + rows) % 27 for i in range(self.n)]      # "tuning passes"
            return "tuned"
        return "replayed"

    def export(self, rows):
        if rows in self.imported:
            return self.imported[rows]
        if rows not in self.tuned:
            raise RuntimeError("batch has not been tuned")
        return self.tuned[rows]

    def imp(self, hints, rows):
        if len(hints) != self.n:
            raise RuntimeError("slot count of another plan")
        self.imported[rows] = list(hints)


def _cache(u, d, build="b1", dev="AMD Instinct MI355X", model="sd15"):
    return PinCache(model, (64, 64), dev, build, u.export, u.imp, directory=str(d))


def test_second_process_imports_the_first_ones_pins(tmp_path):
    u1 = FakeUNet()
    c1 = _cache(u1, tmp_path)
    assert not c1.load(16)                       # nothing on disk yet
    assert u1.forward(16) == "tuned"
    assert c1.save(16) and not c1.save(16)       # written once
    files = os.listdir(tmp_path)
    assert len(files) == 1 and files[0].startswith("tune_sd15_64x64_r16_AMD-Instinct-MI355X_b1")
    rec = json.load(open(tmp_path / files[0]))
    assert rec["rows"] == 16 and rec["hints"] == u1.tuned[16]
    u2 = FakeUNet()
    c2 = _cache(u2, tmp_path)
    assert c2.load(16) and u2.imported[16] == u1.tuned[16]
    assert u2.forward(16) == "replayed"          # no tuning passes in the second process
    assert not c2.save(16)                       # and nothing rewritten


def test_other_build_device_batch_or_plan_is_not_replayed(tmp_path):
    u1 = FakeUNet()
    c1 = _cache(u1, tmp_path)
    u1.forward(16)
    c1.save(16)
    for kw in (dict(build="b2"), dict(dev="AMD Instinct MI300X"), dict(model="sdxl")):
        u = FakeUNet()
        assert not _cache(u, tmp_path, **kw).load(16) and not u.imported
    u = FakeUNet()
    assert not _cache(u, tmp_path).load(4) and not u.imported            # another batch
    u = FakeUNet(n_slots=9)
    assert not _cache(u, tmp_path).load(16) and not u.imported           # another plan (slot count): import refuses, tune as usual


def test_untuned_batch_and_unwritable_directory_are_not_errors(tmp_path):
    u = FakeUNet()
    c = _cache(u, tmp_path)
    assert not c.save(8)                         # export fails (autotune off / not tuned): nothing written, no exception
    assert os.listdir(tmp_path) == []
    u.forward(8)
    ro = PinCache("sd15", (64, 64), "x", "b", u.export, u.imp, directory="/proc/cfgpp-not-writable")
    assert not ro.save(8)


def test_cache_switches(monkeypatch, tmp_path):
    monkeypatch.setenv("CFGPP_TUNE_CACHE", str(tmp_path))
    monkeypatch.delenv("CFGPP_AUTOTUNE", raising=False)
    assert cache_dir() == str(tmp_path)
    monkeypatch.setenv("CFGPP_AUTOTUNE", "0")    # heuristic tiles: nothing read, nothing written
    assert cache_dir() is None
    monkeypatch.delenv("CFGPP_AUTOTUNE")
    monkeypatch.setenv("CFGPP_TUNE_CACHE", "0")
    assert cache_dir() is None
    u = FakeUNet()
    off = PinCache("sd15", (64, 64), "x", "b", u.export, u.imp)
    u.forward(2)
    assert not off.save(2) and not off.load(2)


def test_skewed_runs_do_not_persist_pins(tmp_path, monkeypatch):
    """a profiled process (rocprofv3 exports ROCP_* into its child) or one with non-default tuner switches keeps its pins to itself"""
    u = FakeUNet()
    monkeypatch.setenv("ROCP_TOOL_LIB", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    c = _cache(u, tmp_path)
    u.forward(16)
    assert not c.save(16) and os.listdir(tmp_path) == []
    monkeypatch.delenv("ROCP_TOOL_LIB")
    state = [0xf1ffffff]
    u2 = FakeUNet()
    c2 = PinCache("sd15", (64, 64), "x", "b", u2.export, u2.imp, directory=str(tmp_path), knobs=lambda: (state[0],))
    u2.forward(16)
    state[0] = 0xffffffff                        # cfgpp_igemm_set_tune_mask(...) after the engine was built
    assert not c2.save(16) and os.listdir(tmp_path) == []
    state[0] = 0xf1ffffff
    u2.forward(4)
    assert c2.save(4) and len(os.listdir(tmp_path)) == 1


def test_late_tuning_is_still_persisted_and_explicit_import_wins(tmp_path):
    u = FakeUNet()
    c = _cache(u, tmp_path)
    assert not c.save(16)                        # first forward did not tune (e.g. stream capture): not definitive yet
    u.forward(16)
    assert c.save(16)                            # the second one did
    # another rank: pins arrive by broadcast, the stale file on its disk must not replace them
    u2 = FakeUNet()
    c2 = _cache(u2, tmp_path)
    u2.imp([1, 2, 3, 4, 5], 16)
    c2.mark_imported(16)
    assert not c2.load(16) and u2.imported[16] == [1, 2, 3, 4, 5] and not c2.save(16)
