"""vtc_amd -- MI355X-native implementation of the DL++ VTC step (danieldager/VTC `src.pipeline.vtc`).

Layout:
  csrc/           hand-written gfx950 kernels + the C ABI (include/vtc_amd.h) -> libvtc_amd.so
  lib.py          ctypes binding (fails loudly when the library is missing; no CPU fallback)
  engine.py       VtcEngine: one vtc_model handle per GPU
  segma_compat/   the segma import surface the reference host code uses, backed by the engine
  core/, utils.py host-side mirror of src/core/{intervals,metadata}.py and src/utils.py
  pipeline/vtc.py the `python -m src.pipeline.vtc`-compatible CLI
  launcher.py     single-node multi-GPU launcher (replaces the SLURM array)
"""

__version__ = "0.1.0"
