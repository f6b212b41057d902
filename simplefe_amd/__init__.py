"""simplefe_amd -- MI355X (gfx950) implementation of simpleFE's libdsp sample-stream hot path.

  simplefe_amd.lib    ctypes binding of the in-tree HIP extension libsfe_dsp.so (C ABI:
                      include/sfe_dsp.h); raises if the extension is not built
  simplefe_amd.api    blkconv / resample / decimate (the reference's Python projection) and
                      the device-resident Fir / Rs bulk path
  simplefe_amd.synth  deterministic synthetic I/Q streams and filter prototypes
  simplefe_amd.build  hipcc build of the extension
"""
__version__ = "0.1"
