#!/bin/bash
set -u
OUT=gpurun_out/r02_mf16; mkdir -p $OUT
timeout 52 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -6 | tee $OUT/mf16_vs_default_real_sd15.txt
import torch
from cfgpp_amd.hip_engine import HipEngine
from cfgpp_amd import _lib
eng = HipEngine("sd15", max_batch=8)
g = torch.Generator().manual_seed(1)
uc = (torch.randn(1, 77, 768, generator=g) * 0.5).half().cuda(); c = (torch.randn(8, 77, 768, generator=g) * 0.5).half().cuda()
eng.set_context(uc, c)
z = torch.randn(8, 4, 64, 64, generator=g).cuda()
a = torch.cat(eng.predict(z, 481.0)).float().clone()
_lib.load().cfgpp_igemm_set_mf16(4)
b = torch.cat(eng.predict(z, 481.0)).float().clone()
_lib.load().cfgpp_igemm_set_mf16(3)
c3 = torch.cat(eng.predict(z, 481.0)).float().clone()
rel = lambda x, y: float((x - y).norm() / y.norm())
print("real sd15 rows=16 @64x64: rel-L2(mf16=4 vs default)", rel(b, a), " rel-L2(mf16=3 vs default)", rel(c3, a), " mf16=3 == mf16=4:", bool(torch.equal(b, c3)), " finite", bool(torch.isfinite(b).all()), " |eps| mean", float(a.abs().mean()))
PY
