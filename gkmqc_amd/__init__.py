"""gkmqc_amd -- MI355X-native gkm kernel-matrix path behind gkmQC's gkmkern_pylib C ABI.

Only what the hot path needs lives here:
  csrc/      host C (boundary, FASTA, weights, logger) + HIP kernels (gfx950)
  bin/       the built drop-in `gkmkern_pylib.so`
  gkmsvm.py  host-side mirror of the reference's scripts/gkmsvm.py caller
  device.py  ctypes binding of the device-layer C ABI (include/gkm_hip.h)
  synth.py   deterministic synthetic FASTA generator
"""
__version__ = "0.1.0"
