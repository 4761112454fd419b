"""salmon_amd — MI355X-native `salmon quant` hot path (HIP kernels behind a C ABI).

`from salmon_amd import api` gives the host-side mirror of the reference's seams; the compute
lives in libsalmon_hip.so (build: `python -m salmon_amd.build`).
"""
__version__ = "0.1.0"
