"""MI355X-native CFG++ sampling hot path."""
import os as _os

# Kernel arguments in device memory instead of host-coherent memory: the command processor then reads them over the local
# fabric when it dispatches a launch.  A forward is 300 (SD1.5) .. 1200 (SDXL) dependent launches of 10-130 us, so the
# per-launch saving is visible end to end: same box, 10 back-to-back forwards, SD1.5 16 rows 20.84 -> 20.26 ms, SDXL 4 rows
# 40.21 -> 38.54 ms (profiles/r03/ab/kernarg_placement.txt).  The HIP runtime reads the variable when it initialises (lazily,
# at the first HIP call), so importing this package before the first torch.cuda call is early enough; an explicit setting
# in the environment wins.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
