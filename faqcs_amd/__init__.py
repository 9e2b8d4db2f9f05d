"""faqcs_amd -- MI355X-native implementation of the FaQCs per-read hot path.

csrc/       hand-written HIP kernels (gfx950) + the C ABI of include/faqcs_mi.h  -> libfaqcs_mi.so
_capi.py    ctypes view of the C ABI            engine.py  HipEngine (the trim() seam)
options.py  FaQCs flag surface + adapter table  driver.py  FaQCs process contract (FASTQ in, trimmed FASTQ + stats out)
report.py   byte-exact QC.stats.txt / --debug tables       parallel.py  shard + all-reduce of the counter block
"""
__version__ = "0.1.0"
